"""Oracle-backed slab engine (CPU): the same engine interface as housescan_amd.sharded.HipSlabEngine, built
from the stage functions of oracle/kinfu_oracle.c.  Test infrastructure only -- it lets the world_size-2 gloo
test run the production orchestration (SlabOrchestrator) without a GPU."""
import numpy as np
import torch

KEY_NONE = 0x7FFFFFFF


class OracleSlabEngine:
    def __init__(self, O, cfg, z0, z1, halo):
        self.O, self.cfg = O, cfg
        X, Y, Z = cfg.vol
        self.z0, self.z1 = z0, z1
        self.zs0, self.zs1 = max(0, z0 - halo), min(Z, z1 + halo)
        self.vol = np.zeros((self.zs1 - self.zs0, Y, X, 2), np.int16)
        self.init = np.eye(4, dtype=np.float32)
        self.init[:3, :3] = np.array(cfg.init_R, np.float32).reshape(3, 3)
        self.init[:3, 3] = np.array(cfg.init_t, np.float32)
        self.reset()

    def reset(self):
        self.vol[:] = 0
        self.pose = self.init.copy()
        self.frame = 0
        self.lost = False

    def frame_index(self):
        return self.frame

    def frame_begin(self, depth):
        O, cfg = self.O, self.cfg
        lv = [O.bilateral(cfg, depth)]
        lv.append(O.pyrdown(lv[0]))
        lv.append(O.pyrdown(lv[1]))
        self.vcur = [O.vmap(cfg, lv[l], l) for l in range(3)]
        self.ncur = [O.nmap(v) for v in self.vcur]
        self.scaled = O.scale_depth(cfg, depth)
        if self.frame == 0:
            O.integrate(cfg, self.vol, self.scaled, self.pose, zs0=self.zs0)
            tm = [O.transform_maps(self.vcur[l], self.ncur[l], self.pose) for l in range(3)]
            self.vmod = [t[0] for t in tm]
            self.nmod = [t[1] for t in tm]
        else:
            self.prev = self.pose.copy()
            self.est = self.pose.copy()
            self.lost = False

    def icp_accumulate(self, level, r0, r1):
        if self.lost or r0 == r1:
            return torch.zeros(27, dtype=torch.float64)
        sums, _ = self.O.icp_accumulate(self.cfg, level, self.vcur[level], self.ncur[level], self.vmod[level],
                                        self.nmod[level], self.est, self.prev, r0, r1)
        return torch.from_numpy(sums.copy())

    def icp_update(self, sums):
        if self.lost:
            return
        x, ok = self.O.icp_solve(sums.numpy())
        if not ok:
            self.lost = True
        else:
            self.est = self.O.pose_update(self.est, x)

    def integrate(self):
        if not self.lost:
            self.O.integrate(self.cfg, self.vol, self.scaled, self.est, zs0=self.zs0)

    def raycast_local(self):
        vm, nm, keys, _ = self.O.raycast(self.cfg, self.vol, self.est, zs0=self.zs0, zo0=self.z0, zo1=self.z1)
        self._vm, self._nm, self._keys = vm, nm, keys
        return torch.from_numpy(keys.reshape(-1).copy())

    def raycast_resolve(self, keys_min):
        km = keys_min.numpy().reshape(self._keys.shape)
        mine = (self._keys == km) & (km != KEY_NONE) & ((km & 1) == 0)
        maps = np.concatenate([self._vm, self._nm], axis=0).view(np.int32)
        return torch.from_numpy(np.where(mine[None], maps, 0).astype(np.int32).reshape(-1))

    def frame_end(self, keys_min, bits):
        if self.frame == 0:
            self.frame = 1
            return self.pose.copy(), False
        if self.lost:
            self.reset()
            return self.pose.copy(), False
        H, W = self.cfg.H, self.cfg.W
        km = keys_min.numpy().reshape(H, W)
        hit = (km != KEY_NONE) & ((km & 1) == 0)
        maps = bits.numpy().reshape(6, H, W).view(np.float32)
        maps = np.where(hit[None], maps, np.float32(np.nan)).astype(np.float32)
        self.vmod = [np.ascontiguousarray(maps[:3])]
        self.nmod = [np.ascontiguousarray(maps[3:])]
        for l in (1, 2):
            self.vmod.append(self.O.resize_vmap(self.vmod[l - 1]))
            self.nmod.append(self.O.resize_nmap(self.nmod[l - 1]))
        self.pose = self.est.copy()
        self.frame += 1
        return self.pose.copy(), True
