"""ctypes binding of libhskinfu.so (the C ABI declared in include/hskinfu.h).

The library is the product; there is NO fallback: if the shared object is missing the import fails loudly, and
`hsk_create` fails with HSK_ERR_NOGPU when no HIP device is present.
"""
import ctypes as C
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhskinfu.so")

HSK_LEVELS = 3
HSK_NSTAGES = 4
HSK_KEY_NONE = 0x7FFFFFFF


class HskConfig(C.Structure):
    """Mirror of `hsk_config` (include/hskinfu.h)."""

    _fields_ = [
        ("vol_x", C.c_int), ("vol_y", C.c_int), ("vol_z", C.c_int),
        ("vol_size_m", C.c_float * 3),
        ("trunc_dist_m", C.c_float),
        ("width", C.c_int), ("height", C.c_int),
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
        ("icp_iters", C.c_int * HSK_LEVELS),
        ("icp_dist_thresh_m", C.c_float),
        ("icp_angle_thresh_sin", C.c_float),
        ("integrate_move_thresh", C.c_float),
        ("init_pose", C.c_float * 16),
        ("device_id", C.c_int),
        ("own_z0", C.c_int), ("own_z1", C.c_int), ("halo", C.c_int),
        ("use_graph", C.c_int),
    ]


# every symbol include/hskinfu.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_F = C.POINTER(C.c_float)
_D = C.POINTER(C.c_double)
_I = C.POINTER(C.c_int)
SYMBOLS = {
    "hsk_build_id": (C.c_char_p, []),
    "hsk_default_config": (None, [C.POINTER(HskConfig), C.c_int]),
    "hsk_create": (C.c_int, [C.POINTER(HskConfig), C.POINTER(_P)]),
    "hsk_destroy": (None, [_P]),
    "hsk_reset": (C.c_int, [_P]),
    "hsk_last_error": (C.c_char_p, [_P]),
    "hsk_process_frame": (C.c_int, [_P, _P, C.c_int, C.c_int, _F, _I]),
    "hsk_process_frame_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, _F, _I]),
    "hsk_submit_frame_dev": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "hsk_submit_frame": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "hsk_wait_frame": (C.c_int, [_P, _F, _I]),
    "hsk_integrate": (C.c_int, [_P, _P, C.c_int, C.c_int, _F]),
    "hsk_raycast": (C.c_int, [_P, _F, _P, _P, _P]),
    "hsk_preprocess": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "hsk_icp_accumulate": (C.c_int, [_P, C.c_int, _F, C.c_int, C.c_int, _D]),
    "hsk_icp_solve": (C.c_int, [_D, _F, _I]),
    "hsk_count_updates": (C.c_int, [_P, _P, C.c_int, C.c_int, _F, C.POINTER(C.c_uint64)]),
    "hsk_download_tsdf": (C.c_int, [_P, _P]),
    "hsk_upload_tsdf": (C.c_int, [_P, _P]),
    "hsk_flush_weights": (C.c_int, [_P]),
    "hsk_prepare_readout": (C.c_int, [_P, C.c_size_t]),
    "hsk_stored_planes": (C.c_int, [_P, _I, _I]),
    "hsk_get_pose": (C.c_int, [_P, _F]),
    "hsk_set_pose": (C.c_int, [_P, _F]),
    "hsk_download_map": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "hsk_upload_map": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "hsk_download_depth_level": (C.c_int, [_P, C.c_int, _P]),
    "hsk_download_scaled_depth": (C.c_int, [_P, _P]),
    "hsk_extract_cloud": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "hsk_extract_mesh": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "hsk_extract_mesh_cubes": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "hsk_mgpu_frame_begin": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "hsk_mgpu_prefetch": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "hsk_mgpu_frame_front": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "hsk_mgpu_icp_accumulate": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P]),
    "hsk_mgpu_icp_update": (C.c_int, [_P, _P]),
    "hsk_mgpu_icp_replicated": (C.c_int, [_P]),
    "hsk_mgpu_integrate": (C.c_int, [_P]),
    "hsk_mgpu_raycast_local": (C.c_int, [_P, _P]),
    "hsk_mgpu_raycast_resolve": (C.c_int, [_P, _P, _P]),
    "hsk_mgpu_raycast_push": (C.c_int, [_P, _P, C.POINTER(_P), C.c_int]),
    "hsk_mgpu_frame_end": (C.c_int, [_P, _P, _P, _F, _I]),
    "hsk_mgpu_frame_index": (C.c_int, [_P]),
    "hsk_mgpu_frame_end_async": (C.c_int, [_P, _P, _P]),
    "hsk_mgpu_restart_pending": (C.c_int, [_P]),
    "hsk_stream": (_P, [_P]),
    "hsk_set_stream": (C.c_int, [_P, _P]),
    "hsk_synchronize": (C.c_int, [_P]),
    "hsk_set_profiling": (C.c_int, [_P, C.c_int]),
    "hsk_stage_ms": (C.c_int, [_P, _D, C.POINTER(C.c_uint64), C.c_int]),
    "hsk_icp_level_ms": (C.c_int, [_P, _D]),
    "hsk_submit_host_us": (C.c_int, [_P, _D, C.POINTER(C.c_uint64), C.c_int]),
    "hsk_integrate_queue_entries": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "hsk_integrate_light_entries": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "hsk_integrate_coarse_counts": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "hsk_bilateral_tables": (C.c_int, [_F, _F]),
    "hsk_selftest_exact_ops": (C.c_int, [C.c_int, C.POINTER(C.c_uint64)]),
    "hsk_group_create": (C.c_int, [C.POINTER(HskConfig), C.c_int, _I, C.c_int, C.POINTER(_P)]),
    "hsk_group_unique_id": (C.c_int, [_P]),
    "hsk_group_create_rank": (C.c_int, [C.POINTER(HskConfig), C.c_int, C.c_int, _P, C.c_int, C.POINTER(_P)]),
    "hsk_group_destroy": (None, [_P]),
    "hsk_group_last_error": (C.c_char_p, [_P]),
    "hsk_group_reset": (C.c_int, [_P]),
    "hsk_group_process_frame": (C.c_int, [_P, _P, C.c_int, C.c_int, _F, _I]),
    "hsk_group_submit_frame": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "hsk_group_submit_frame_dev": (C.c_int, [_P, C.POINTER(_P), C.c_int, C.c_int]),
    "hsk_group_wait_frame": (C.c_int, [_P, _F, _I]),
    "hsk_group_exchange_ms": (C.c_int, [_P, _D, _D, C.POINTER(C.c_ulonglong)]),
    "hsk_group_n_slabs": (C.c_int, [_P]),
    "hsk_group_ranks_seen": (C.c_int, [_P, _I]),
    "hsk_group_slab": (_P, [_P, C.c_int]),
    "hsk_group_download_tsdf": (C.c_int, [_P, _P]),
    "hsk_synth_pose": (C.c_int, [C.c_int, _F]),
    "hsk_synth_render": (C.c_int, [_F, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, _P]),
    "hsk_synth_room_extents": (C.c_int, [C.c_int, _F]),
    "hsk_synth_room_pose": (C.c_int, [C.c_int, C.c_int, C.c_int, _F]),
    "hsk_synth_room_render": (C.c_int, [C.c_int, _F, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, _P]),
    "hsk_synth_render_sensor": (C.c_int, [C.c_int, _F, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_uint64, C.c_float, C.c_float, C.c_int, _P, _D]),
    "hsk_write_pcd_xyz": (C.c_int, [C.c_char_p, _P, C.c_size_t]),
    "hsk_write_ply_mesh": (C.c_int, [C.c_char_p, _P, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "hsk_weld_triangles": (C.c_int, [_P, C.c_size_t, _P, C.c_size_t, C.POINTER(C.c_size_t), _P]),
    "hsk_voxel_downsample": (C.c_int, [_P, C.c_size_t, C.c_float, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "hsk_detect_planes": (C.c_int, [_P, C.c_size_t, C.c_float, C.c_float, C.c_int, C.c_int, _P, _P, _I]),
    "hsk_plane_hull": (C.c_int, [_P, C.c_size_t, _P, C.c_int, _F, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "hsk_write_planes_txt": (C.c_int, [C.c_char_p, _P, C.c_int]),
    "hsk_write_xf": (C.c_int, [C.c_char_p, _F]),
    "hsk_read_xf": (C.c_int, [C.c_char_p, _F]),
    "hsk_transform_cloud": (C.c_int, [_P, C.c_size_t, _F, _P]),
    "hsk_stream_create": (_P, [C.c_char_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]),
    "hsk_stream_open": (_P, [C.c_char_p, _I, _I, _I, _F]),
    "hsk_stream_write": (C.c_int, [_P, _P]),
    "hsk_stream_read": (C.c_int, [_P, C.c_int, _P]),
    "hsk_stream_close": (C.c_int, [_P]),
    "hsk_stream_info": (C.c_int, [_P, _I, _I, _I, _F]),
    "hsk_track_stream": (C.c_int, [_P, _P, C.c_int, C.c_int, _F, _I]),
}

# every symbol include/hshouse.h declares (host-side room stitching)
_U32 = C.POINTER(C.c_uint32)
_SZ = C.POINTER(C.c_size_t)
HSH_OBJECTIVE = C.CFUNCTYPE(C.c_double, _D, C.c_int, _P)
HOUSE_SYMBOLS = {
    "hsh_create": (_P, []),
    "hsh_destroy": (None, [_P]),
    "hsh_last_error": (C.c_char_p, [_P]),
    "hsh_load_room": (C.c_int, [_P, C.c_char_p, _U32]),
    "hsh_add_room": (C.c_int, [_P, C.c_char_p, _P, C.c_size_t, _P, C.c_int, _P, _P, _U32]),
    "hsh_room_ids": (C.c_int, [_P, _P, C.c_int, _I]),
    "hsh_room_planes": (C.c_int, [_P, C.c_uint32, _P, _P, C.c_int, _I]),
    "hsh_plane_bounds": (C.c_int, [_P, C.c_uint32, _P, C.c_int, _I]),
    "hsh_room_corners": (C.c_int, [_P, C.c_uint32, C.c_int, _P, _P, C.c_int, _I]),
    "hsh_room_cloud": (C.c_int, [_P, C.c_uint32, _P, C.c_size_t, _SZ]),
    "hsh_room_means": (C.c_int, [_P, C.c_uint32, _P, _P]),
    "hsh_set_room_corners": (C.c_int, [_P, C.c_uint32, _P, C.c_int]),
    "hsh_accept_corner_suggestion": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "hsh_translate_room": (C.c_int, [_P, C.c_uint32, _P]),
    "hsh_rotate_room": (C.c_int, [_P, C.c_uint32, _P]),
    "hsh_rotate_kinfu_room": (C.c_int, [_P, C.c_uint32]),
    "hsh_room_auto_align_axis": (C.c_int, [_P, C.c_uint32, _P]),
    "hsh_auto_align_floor": (C.c_int, [_P, C.c_uint32]),
    "hsh_remove_ceiling": (C.c_int, [_P, C.c_uint32]),
    "hsh_suggest_points": (C.c_int, [_P, C.c_uint32, C.c_float, _I, _I]),
    "hsh_fit_cuboid_to_room": (C.c_int, [_P, C.c_uint32, C.c_int, _I, _D, _D]),
    "hsh_connect_walls": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_int, C.c_float, _I]),
    "hsh_disconnect_walls": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "hsh_connected_walls": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, _I]),
    "hsh_optimize_room_positions": (C.c_int, [_P, _D]),
    "hsh_room_projection": (C.c_int, [_P, C.c_uint32, _F]),
    "hsh_room_projection_string": (C.c_int, [_P, C.c_uint32, C.c_int, C.c_char_p, C.c_size_t]),
    "hsh_export_all_room_xf_files": (C.c_int, [_P, C.c_char_p]),
    "hsh_plane_corner": (C.c_int, [_P, _P, _I]),
    "hsh_fit_plane": (C.c_int, [_P, C.c_int, _P]),
    "hsh_rotation_between": (C.c_int, [_P, _P, _P]),
    "hsh_cuboid_from_params": (C.c_int, [_P, _P]),
    "hsh_guess_dims": (C.c_int, [_P, _P]),
    "hsh_errfun": (C.c_int, [_P, _P, C.c_int, _D]),
    "hsh_fit_cuboid": (C.c_int, [_P, C.c_int, C.c_int, _P, _I, _D]),
    "hsh_nm_minimize": (C.c_int, [HSH_OBJECTIVE, _P, C.c_int, _P, _P, C.c_double, C.c_int, _P, _D, _I]),
    "hsh_lstsq_distances": (C.c_int, [_P, _P, _P, C.c_int, _P, _P, C.c_int, _I, _D]),
    "hsh_group_connected_components": (C.c_int, [_P, _P, C.c_int, _P, _I]),
    "hsh_show_float": (C.c_int, [C.c_float, C.c_char_p, C.c_size_t]),
    "hsh_read_pcd_xyz": (C.c_int, [C.c_char_p, _P, C.c_size_t, _SZ]),
    "hsh_read_planes_txt": (C.c_int, [C.c_char_p, _P, C.c_int, _I]),
    "hsh_write_ply_points": (C.c_int, [C.c_char_p, _P, C.c_size_t]),
    "hsh_read_ply_points": (C.c_int, [C.c_char_p, _P, C.c_size_t, _SZ]),
}

_lib = None


def _share_torch_hip_runtime():
    """One HIP runtime per process.  The torch wheel ships its own libamdhip64.so.7 (ROCm 7.0) beside the system's
    (ROCm 7.2, the one libhskinfu.so names in its RUNPATH); the loader keeps whichever copy comes first for BOTH, and
    torch cannot initialise on top of the system copy ("No HIP GPUs are available").  So when torch is installed but not
    imported yet, its copy is loaded first; without torch the system runtime is used."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            return  # fall back to the system runtime; a later `import torch` in this process may then fail
        # ... and one RCCL: hsk_group_* loads RCCL at run time; beside torch's HIP runtime it must be torch's copy
        rccl = os.path.join(os.path.dirname(spec.origin), "lib", "librccl.so")
        if os.path.exists(rccl):
            os.environ.setdefault("HSK_RCCL_PATH", rccl)


def load():
    """Load libhskinfu.so and bind every symbol; raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    _share_torch_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C housescan_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in list(SYMBOLS.items()) + list(HOUSE_SYMBOLS.items()):
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
