"""housescan_amd -- MI355X-native KinectFusion core behind a C ABI (include/hskinfu.h).

Importing the package loads libhskinfu.so; it fails loudly if the HIP library has not been built.
"""
from . import _lib

_lib.load()

from .kinfu import (GROUP_DIRECT, GROUP_FORCE_RCCL, GROUP_ICP_ALLREDUCE, GROUP_PROFILE, KinfuError, KinfuGroup, KinfuTracker,  # noqa: E402
                    bilateral_tables, default_config, synth_depth, synth_noisy_frames, synth_pose, synth_room_depth, synth_room_extents,
                    synth_room_pose, synth_sensor_depth, synth_sensor_frames)

from .products import DepthStreamReader, DepthStreamWriter  # noqa: E402  (recorded depth streams: the HSKD container)

__all__ = ["DepthStreamReader", "DepthStreamWriter", "KinfuError", "KinfuTracker", "KinfuGroup", "GROUP_FORCE_RCCL", "GROUP_ICP_ALLREDUCE", "GROUP_DIRECT", "GROUP_PROFILE", "default_config",
           "synth_depth", "synth_noisy_frames", "synth_pose", "bilateral_tables", "synth_room_depth", "synth_room_extents", "synth_room_pose", "synth_sensor_depth", "synth_sensor_frames"]
