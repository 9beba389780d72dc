"""Host mirror of the room-stitching chain (include/hshouse.h; SURVEY.md 8f-2, BASELINE configs[0]).

`House` keeps the reference's vocabulary -- rooms, planes (walls), corners, connected walls -- and its function
names (camelCase aliases of the reference's own names are provided next to the snake_case methods), so a script
against HouseScan's `Main.hs` reads the same here:

    loadRoom -> rotateKinfuRoom -> autoAlignFloor -> removeCeiling -> suggestPoints -> fitCuboidToRoom
      -> connectWalls -> optimizeRoomPositions -> exportAllRoomXfFiles / roomProjectionToString

All numerics run in libhskinfu.so (housescan_amd/csrc/house.cpp); this file is ctypes plumbing only.
"""
import ctypes as C
import os

import numpy as np

from . import _lib

AXIS_X, AXIS_Y, AXIS_Z = 0, 1, 2
OPPOSITE, SAME = 0, 1
FIT_AS_NAMED, FIT_AS_PASSED = 0, 1
FIT_FROM_CENTER_FIRST, FIT_FROM_CENTER, FIT_ORDERED = 0, 1, 2
DEFAULT_WALL_THICKNESS = 0.1        # Main.hs:1080
DEFAULT_SUGGESTION_CUTOFF = 1.2     # Main.hs:1084


class HouseError(RuntimeError):
    pass


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


def _f64(a):
    return np.ascontiguousarray(a, np.float64)


def _ck0(rc, what):
    if rc != 0:
        msg = _lib.load().hsh_last_error(None)
        raise HouseError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


# ---- house-less numerics ------------------------------------------------------------------------------------
def plane_corner(eq1, eq2, eq3):
    """planeCorner (Main.hs:1413-1430): intersection of three planes (n, d) with n.x = d, or None."""
    eq = _f32([eq1, eq2, eq3]).reshape(12)
    out = np.zeros(3, np.float32)
    found = C.c_int()
    _ck0(_lib.load().hsh_plane_corner(eq.ctypes.data, out.ctypes.data, C.byref(found)), "hsh_plane_corner")
    return out if found.value else None


def fit_plane(points):
    """fitPlane (Main.hs:1436-1450): total-least-squares plane (n, d) through >= 3 points."""
    pts = _f32(points).reshape(-1, 3)
    out = np.zeros(4, np.float32)
    _ck0(_lib.load().hsh_fit_plane(pts.ctypes.data, len(pts), out.ctypes.data), "hsh_fit_plane")
    return out


def rotation_between(n1, n2):
    """rotationBetweenPlaneEqs (Main.hs:1553-1560): right-multiplicative 3x3 R with n1 R = n2."""
    out = np.zeros((3, 3), np.float32)
    a, b = _f32(n1), _f32(n2)   # keep the converted arrays alive across the call
    _ck0(_lib.load().hsh_rotation_between(a.ctypes.data, b.ctypes.data, out.ctypes.data), "hsh_rotation_between")
    return out


def cuboid_from_params(params):
    """cuboidFromParams (FitCuboidBFGS.hs:98-112): [x,y,z, a,b,c, q0,q1,q2,q3] -> 8 corners."""
    p = _f64(params).reshape(10)
    out = np.zeros((8, 3))
    _ck0(_lib.load().hsh_cuboid_from_params(p.ctypes.data, out.ctypes.data), "hsh_cuboid_from_params")
    return out


def guess_dims(corners):
    c = _f64(corners).reshape(24)
    out = np.zeros(3)
    _ck0(_lib.load().hsh_guess_dims(c.ctypes.data, out.ctypes.data), "hsh_guess_dims")
    return out


def errfun(corners, params, closest=False):
    c, p = _f64(corners).reshape(24), _f64(params).reshape(10)
    e = C.c_double()
    _ck0(_lib.load().hsh_errfun(c.ctypes.data, p.ctypes.data, int(closest), C.byref(e)), "hsh_errfun")
    return e.value


def fit_cuboid(corners, mode=FIT_FROM_CENTER_FIRST, arg_order=FIT_AS_PASSED):
    """fitCuboidFromCenterFirst / fitCuboidFromCenter / fitCuboid (FitCuboidBFGS.hs:172-233) -> (params, steps, err)."""
    c = _f64(corners).reshape(24)
    p = np.zeros(10)
    steps, err = C.c_int(), C.c_double()
    _ck0(_lib.load().hsh_fit_cuboid(c.ctypes.data, mode, arg_order, p.ctypes.data, C.byref(steps), C.byref(err)), "hsh_fit_cuboid")
    return p, steps.value, err.value


def nm_minimize(f, start, steps, eps=1e-8, maxit=2000):
    """GSL nmsimplex2 under hmatrix-gsl's `minimize` loop -> (x, f(x), iterations)."""
    n = len(start)
    cb = _lib.HSH_OBJECTIVE(lambda x, k, _u: float(f(np.ctypeslib.as_array(x, (k,)).copy())))
    s, st, out = _f64(start), _f64(steps), np.zeros(n)
    fv, it = C.c_double(), C.c_int()
    _ck0(_lib.load().hsh_nm_minimize(cb, None, n, s.ctypes.data, st.ctypes.data, eps, maxit, out.ctypes.data, C.byref(fv), C.byref(it)),
         "hsh_nm_minimize")
    return out, fv.value, it.value


def lstsq_distances(dist_map):
    """lstSqDistances (TranslationOptimizer.hs:36-42): {(a, b): d} -> ({node: position}, rmse) or None."""
    keys = list(dist_map)
    a = np.array([k[0] for k in keys], np.uint32)
    b = np.array([k[1] for k in keys], np.uint32)
    d = np.array([dist_map[k] for k in keys], np.float64)
    cap = 2 * len(keys)
    nodes, pos = np.zeros(cap, np.uint32), np.zeros(cap)
    n, rmse = C.c_int(), C.c_double()
    rc = _lib.load().hsh_lstsq_distances(a.ctypes.data, b.ctypes.data, d.ctypes.data, len(keys), nodes.ctypes.data, pos.ctypes.data, cap,
                                         C.byref(n), C.byref(rmse))
    if rc == -4:
        return None
    _ck0(rc, "hsh_lstsq_distances")
    return {int(nodes[i]): float(pos[i]) for i in range(n.value)}, rmse.value


def group_connected_components(edges_data):
    """groupConnectedComponents (GroupConnectedComponents.hs:16-32): [((i, j), data)] -> list of such lists."""
    if not edges_data:
        return []
    a = np.array([e[0][0] for e in edges_data], np.uint32)
    b = np.array([e[0][1] for e in edges_data], np.uint32)
    comp = np.zeros(len(edges_data), np.int32)
    n = C.c_int()
    _ck0(_lib.load().hsh_group_connected_components(a.ctypes.data, b.ctypes.data, len(edges_data), comp.ctypes.data, C.byref(n)),
         "hsh_group_connected_components")
    return [[e for e, c in zip(edges_data, comp) if c == k] for k in range(n.value)]


def show_float(v):
    buf = C.create_string_buffer(64)
    _ck0(_lib.load().hsh_show_float(C.c_float(v), buf, 64), "hsh_show_float")
    return buf.value.decode()


def read_pcd_xyz(path):
    lib = _lib.load()
    n = C.c_size_t()
    _ck0(lib.hsh_read_pcd_xyz(os.fsencode(path), None, 0, C.byref(n)), "hsh_read_pcd_xyz")
    out = np.zeros((n.value, 3), np.float32)
    _ck0(lib.hsh_read_pcd_xyz(os.fsencode(path), out.ctypes.data, n.value, C.byref(n)), "hsh_read_pcd_xyz")
    return out


def write_ply_points(path, xyz):
    pts = _f32(xyz).reshape(-1, 3)
    _ck0(_lib.load().hsh_write_ply_points(os.fsencode(path), pts.ctypes.data, len(pts)), "hsh_write_ply_points")


def read_ply_points(path):
    lib = _lib.load()
    n = C.c_size_t()
    _ck0(lib.hsh_read_ply_points(os.fsencode(path), None, 0, C.byref(n)), "hsh_read_ply_points")
    out = np.zeros((n.value, 3), np.float32)
    _ck0(lib.hsh_read_ply_points(os.fsencode(path), out.ctypes.data, n.value, C.byref(n)), "hsh_read_ply_points")
    return out


# ---- the house ------------------------------------------------------------------------------------------------
class House:
    """The rooms of one building plus the wall connections between them (Main.hs `sRooms`, `sConnectedWalls`)."""

    def __init__(self):
        self._lib = _lib.load()
        self._h = self._lib.hsh_create()
        if not self._h:
            raise HouseError("hsh_create failed")

    def close(self):
        if self._h:
            self._lib.hsh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc != 0:
            raise HouseError(f"{what} failed ({rc}): {self._lib.hsh_last_error(self._h).decode()}")

    # -- rooms
    def load_room(self, directory):
        """loadRoom (Main.hs:1738-1762) -> room id"""
        rid = C.c_uint32()
        self._ck(self._lib.hsh_load_room(self._h, os.fsencode(directory), C.byref(rid)), "hsh_load_room")
        return rid.value

    def add_room(self, name, cloud, planes_abcd, hulls):
        """loadRoom from memory: planes in PCL file form (ax+by+cz+d=0), one hull polygon per plane."""
        pts = _f32(cloud).reshape(-1, 3)
        eq = _f32(planes_abcd).reshape(-1, 4)
        offs = np.zeros(len(eq) + 1, np.int32)
        for k, hk in enumerate(hulls):
            offs[k + 1] = offs[k] + len(hk)
        hull = _f32(np.concatenate([_f32(x).reshape(-1, 3) for x in hulls])) if len(hulls) else np.zeros((0, 3), np.float32)
        rid = C.c_uint32()
        self._ck(self._lib.hsh_add_room(self._h, os.fsencode(name), pts.ctypes.data, len(pts), eq.ctypes.data, len(eq), hull.ctypes.data,
                                        offs.ctypes.data, C.byref(rid)), "hsh_add_room")
        return rid.value

    def room_ids(self):
        n = C.c_int()
        self._ck(self._lib.hsh_room_ids(self._h, None, 0, C.byref(n)), "hsh_room_ids")
        ids = np.zeros(max(1, n.value), np.uint32)
        self._ck(self._lib.hsh_room_ids(self._h, ids.ctypes.data, len(ids), C.byref(n)), "hsh_room_ids")
        return [int(i) for i in ids[:n.value]]

    def room_planes(self, room):
        """-> (plane ids, [k,4] (nx, ny, nz, d) with n.x = d)"""
        n = C.c_int()
        self._ck(self._lib.hsh_room_planes(self._h, room, None, None, 0, C.byref(n)), "hsh_room_planes")
        ids, eq = np.zeros(max(1, n.value), np.uint32), np.zeros((max(1, n.value), 4), np.float32)
        self._ck(self._lib.hsh_room_planes(self._h, room, ids.ctypes.data, eq.ctypes.data, len(ids), C.byref(n)), "hsh_room_planes")
        return [int(i) for i in ids[:n.value]], eq[:n.value]

    def plane_bounds(self, plane):
        n = C.c_int()
        self._ck(self._lib.hsh_plane_bounds(self._h, plane, None, 0, C.byref(n)), "hsh_plane_bounds")
        out = np.zeros((max(1, n.value), 3), np.float32)
        self._ck(self._lib.hsh_plane_bounds(self._h, plane, out.ctypes.data, len(out), C.byref(n)), "hsh_plane_bounds")
        return out[:n.value]

    def room_corners(self, room, suggested=False):
        n = C.c_int()
        self._ck(self._lib.hsh_room_corners(self._h, room, int(suggested), None, None, 0, C.byref(n)), "hsh_room_corners")
        ids, xyz = np.zeros(max(1, n.value), np.uint32), np.zeros((max(1, n.value), 3), np.float32)
        self._ck(self._lib.hsh_room_corners(self._h, room, int(suggested), ids.ctypes.data, xyz.ctypes.data, len(ids), C.byref(n)),
                 "hsh_room_corners")
        return [int(i) for i in ids[:n.value]], xyz[:n.value]

    def room_cloud(self, room):
        n = C.c_size_t()
        self._ck(self._lib.hsh_room_cloud(self._h, room, None, 0, C.byref(n)), "hsh_room_cloud")
        out = np.zeros((max(1, n.value), 3), np.float32)
        self._ck(self._lib.hsh_room_cloud(self._h, room, out.ctypes.data, len(out), C.byref(n)), "hsh_room_cloud")
        return out[:n.value]

    def room_mean(self, room):
        m = np.zeros(3, np.float32)
        self._ck(self._lib.hsh_room_means(self._h, room, m.ctypes.data, None), "hsh_room_means")
        return m

    def corner_mean(self, room):
        m = np.zeros(3, np.float32)
        self._ck(self._lib.hsh_room_means(self._h, room, None, m.ctypes.data), "hsh_room_means")
        return m

    def set_room_corners(self, room, corners):
        c = _f32(corners).reshape(-1, 3)
        self._ck(self._lib.hsh_set_room_corners(self._h, room, c.ctypes.data, len(c)), "hsh_set_room_corners")

    def accept_corner_suggestion(self, room, suggestion_id):
        self._ck(self._lib.hsh_accept_corner_suggestion(self._h, room, suggestion_id), "hsh_accept_corner_suggestion")

    # -- rigid edits
    def translate_room(self, room, off):
        v = _f32(off)
        self._ck(self._lib.hsh_translate_room(self._h, room, v.ctypes.data), "hsh_translate_room")

    def rotate_room(self, room, rot_right):
        r = _f32(rot_right).reshape(9)
        self._ck(self._lib.hsh_rotate_room(self._h, room, r.ctypes.data), "hsh_rotate_room")

    def rotate_kinfu_room(self, room):
        self._ck(self._lib.hsh_rotate_kinfu_room(self._h, room), "hsh_rotate_kinfu_room")

    def room_auto_align_axis(self, room, axis):
        v = _f32(axis)
        self._ck(self._lib.hsh_room_auto_align_axis(self._h, room, v.ctypes.data), "hsh_room_auto_align_axis")

    def auto_align_floor(self, room):
        self._ck(self._lib.hsh_auto_align_floor(self._h, room), "hsh_auto_align_floor")

    def remove_ceiling(self, room):
        self._ck(self._lib.hsh_remove_ceiling(self._h, room), "hsh_remove_ceiling")

    # -- corners, cuboid
    def suggest_points(self, room, cutoff_factor=DEFAULT_SUGGESTION_CUTOFF):
        """suggestPoints (Main.hs:1522-1538) -> (number suggested, adopted as the room's corners?)"""
        n, adopted = C.c_int(), C.c_int()
        self._ck(self._lib.hsh_suggest_points(self._h, room, C.c_float(cutoff_factor), C.byref(n), C.byref(adopted)), "hsh_suggest_points")
        return n.value, bool(adopted.value)

    def fit_cuboid_to_room(self, room, arg_order=FIT_AS_PASSED):
        """fitCuboidToRoom (Main.hs:1814-1847) -> (params[10], steps, rmse)"""
        steps, rmse = C.c_int(), C.c_double()
        p = np.zeros(10)
        self._ck(self._lib.hsh_fit_cuboid_to_room(self._h, room, arg_order, C.byref(steps), C.byref(rmse), p.ctypes.data_as(C.POINTER(C.c_double))),
                 "hsh_fit_cuboid_to_room")
        return p, steps.value, rmse.value

    # -- walls
    def connect_walls(self, plane1, plane2, relation=OPPOSITE, thickness=DEFAULT_WALL_THICKNESS):
        """connectWalls (Main.hs:2019-2052) -> True when a new connection was recorded"""
        c = C.c_int()
        self._ck(self._lib.hsh_connect_walls(self._h, plane1, plane2, relation, C.c_float(thickness), C.byref(c)), "hsh_connect_walls")
        return bool(c.value)

    def disconnect_walls(self, plane1, plane2):
        self._ck(self._lib.hsh_disconnect_walls(self._h, plane1, plane2), "hsh_disconnect_walls")

    def connected_walls(self):
        n = C.c_int()
        self._ck(self._lib.hsh_connected_walls(self._h, None, None, None, None, None, 0, C.byref(n)), "hsh_connected_walls")
        k = max(1, n.value)
        ax, rel, th = np.zeros(k, np.int32), np.zeros(k, np.int32), np.zeros(k, np.float32)
        p1, p2 = np.zeros(k, np.uint32), np.zeros(k, np.uint32)
        self._ck(self._lib.hsh_connected_walls(self._h, ax.ctypes.data, rel.ctypes.data, th.ctypes.data, p1.ctypes.data, p2.ctypes.data, k,
                                               C.byref(n)), "hsh_connected_walls")
        return [(int(ax[i]), int(rel[i]), float(th[i]), int(p1[i]), int(p2[i])) for i in range(n.value)]

    def optimize_room_positions(self):
        """optimizeRoomPositions (Main.hs:2074-2162) -> worst component RMSE per axis (NaN: axis untouched)"""
        r = np.zeros(3)
        self._ck(self._lib.hsh_optimize_room_positions(self._h, r.ctypes.data_as(C.POINTER(C.c_double))), "hsh_optimize_room_positions")
        return r

    # -- export
    def room_projection(self, room):
        """Row-major left-multiplicative 4x4: how the room was moved versus its file (Main.hs:2271-2284)."""
        m = np.zeros(16, np.float32)
        self._ck(self._lib.hsh_room_projection(self._h, room, m.ctypes.data_as(C.POINTER(C.c_float))), "hsh_room_projection")
        return m.reshape(4, 4)

    def room_projection_to_string(self, room):
        buf = C.create_string_buffer(1024)
        self._ck(self._lib.hsh_room_projection_string(self._h, room, 0, buf, 1024), "hsh_room_projection_string")
        return buf.value.decode()

    def room_projection_to_xf_format(self, room):
        buf = C.create_string_buffer(1024)
        self._ck(self._lib.hsh_room_projection_string(self._h, room, 1, buf, 1024), "hsh_room_projection_string")
        return buf.value.decode()

    def export_all_room_xf_files(self, directory="xf"):
        self._ck(self._lib.hsh_export_all_room_xf_files(self._h, os.fsencode(directory)), "hsh_export_all_room_xf_files")

    # the reference's own names
    loadRoom = load_room
    rotateKinfuRoom = rotate_kinfu_room
    autoAlignFloor = auto_align_floor
    roomAutoAlignAxis = room_auto_align_axis
    removeCeiling = remove_ceiling
    translateRoom = translate_room
    rotateRoom = rotate_room
    suggestPoints = suggest_points
    acceptCornerSuggestion = accept_corner_suggestion
    fitCuboidToRoom = fit_cuboid_to_room
    connectWalls = connect_walls
    disconnectWalls = disconnect_walls
    optimizeRoomPositions = optimize_room_positions
    roomProjectionToString = room_projection_to_string
    roomProjectionToXfFormat = room_projection_to_xf_format
    exportAllRoomXfFiles = export_all_room_xf_files


planeCorner = plane_corner
fitPlane = fit_plane
rotationBetweenPlaneEqs = rotation_between
cuboidFromParams = cuboid_from_params
guessDims = guess_dims
fitCuboidFromCenterFirst = fit_cuboid
lstSqDistances = lstsq_distances
groupConnectedComponents = group_connected_components
