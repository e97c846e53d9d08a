"""ctypes binding of libssm_hip.so (include/ssm_hip.h).  Loading fails loudly: there is no CPU fallback."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libssm_hip.so")


class Camera(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("cx", "cy", "fx", "fy", "scale")]


class Config(C.Structure):
    _fields_ = [
        ("width", C.c_int), ("height", C.c_int),
        ("orb_features", C.c_int), ("orb_scale", C.c_float), ("orb_levels", C.c_int),
        ("orb_iniThFAST", C.c_int), ("orb_minThFAST", C.c_int),
        ("knn_match_ratio", C.c_double), ("tracker_ref_frames", C.c_int),
        ("mapper_resolution", C.c_double), ("mapper_max_distance", C.c_double),
        ("camera", Camera), ("max_batch", C.c_int), ("voxel_capacity_log2", C.c_int),
        ("brief_pattern", C.c_void_p),
        ("voxel_max_capacity_log2", C.c_int), ("sgbm_form", C.c_int), ("sgbm_streams", C.c_int), ("stereo_batch", C.c_int),
    ]


class FramesDev(C.Structure):
    _fields_ = [("bgr", C.c_void_p), ("depth", C.c_void_p), ("sem_bgr", C.c_void_p), ("pose", C.c_void_p),
                ("n", C.c_int), ("continue_sequence", C.c_int), ("stages", C.c_int)]


class SeqOutDev(C.Structure):
    _fields_ = [("kps", C.c_void_p), ("desc", C.c_void_p), ("pos3d", C.c_void_p), ("nkp", C.c_void_p),
                ("matches", C.c_void_p), ("nmatch", C.c_void_p), ("npoints", C.c_void_p),
                ("cap", C.c_int), ("R", C.c_int)]


class SgbmParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("minDisparity", "numberOfDisparities", "SADWindowSize", "P1", "P2", "disp12MaxDiff", "preFilterCap",
                                         "uniquenessRatio", "speckleWindowSize", "speckleRange")]


class VoParams(C.Structure):
    _fields_ = [("f", C.c_double), ("cu", C.c_double), ("cv", C.c_double), ("base", C.c_double), ("inlier_threshold", C.c_double),
                ("reweighting", C.c_int32), ("pad", C.c_int32)]


class StereoFramesDev(C.Structure):
    _fields_ = [("left", C.c_void_p), ("right", C.c_void_p), ("n", C.c_int), ("w", C.c_int), ("h", C.c_int), ("continue_sequence", C.c_int),
                ("stages", C.c_int), ("max_corners", C.c_int), ("sgbm", SgbmParams)] + \
               [(n, C.c_double) for n in ("baseline", "cu", "cv", "f", "roix", "roiy", "roiz", "scale")] + \
               [("vo", VoParams), ("ransac_iters", C.c_int), ("rand_stream", C.c_void_p)]


class StereoOutDev(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("quad", "nquad", "corners", "ncorners", "disp", "depth", "tr", "inliers", "vo_result", "rand_draws_used")] + \
               [("max_corners", C.c_int)]


class TrackerParams(C.Structure):
    _fields_ = [("max_lost_frame", C.c_int32), ("ref_frames", C.c_int32), ("pnp_min_inliers", C.c_int32), ("use_device", C.c_int32), ("first_pose", C.c_double * 16), ("own_stream", C.c_int32), ("blocks", C.c_int32)]


# every symbol include/ssm_hip.h declares: name -> (restype, argtypes)
_P, _I, _D, _F, _U64, _SZ = C.c_void_p, C.c_int, C.c_double, C.c_float, C.c_uint64, C.c_size_t
SYMBOLS = {
    "ssm_config_default": (None, [C.POINTER(Config)]),
    "ssm_create": (_I, [_I, C.POINTER(Config), C.POINTER(_P)]),
    "ssm_destroy": (None, [_P]),
    "ssm_last_error": (C.c_char_p, [_P]),
    "ssm_version": (C.c_char_p, []),
    "ssm_orb_capacity": (_I, [_P]),
    "ssm_sync": (_I, [_P]),
    "ssm_stream": (_P, [_P]),
    "ssm_orb_extract": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, C.POINTER(_I)]),
    "ssm_orb_extract_async": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, C.POINTER(_I)]),
    "ssm_match_async": (_I, [_P, _P, _I, _P, _I, _D, _P, _I, C.POINTER(_I)]),
    "ssm_wait": (_I, [_P]),
    "ssm_match_refs": (_I, [_P, _P, _P, _I, _P, _I, _D, _P, _P, _P]),
    "ssm_match_refs_async": (_I, [_P, _P, _P, _I, _P, _I, _D, _P, _P, _P]),
    "ssm_hamming_knn2": (_I, [_P, _P, _I, _P, _I, _P, _P]),
    "ssm_match": (_I, [_P, _P, _I, _P, _I, _D, _P, _I, C.POINTER(_I)]),
    "ssm_moving_mask": (_I, [_P, _P, _I, _I, _I, _P]),
    "ssm_backproject": (_I, [_P, _P, _P, _P, _I, _I, C.POINTER(Camera), _P, _D, _P, _I, C.POINTER(_I)]),
    "ssm_voxel_filter": (_I, [_P, _P, _I, _F, _P, _I, C.POINTER(_I)]),
    "ssm_backproject_dev": (_I, [_P, _P, _P, _P, _I, _I, C.POINTER(Camera), _D, C.POINTER(_P)]),
    "ssm_cloud_size": (_I, [_P]),
    "ssm_cloud_fetch": (_I, [_P, _P, _P, _P, _I, C.POINTER(_I)]),
    "ssm_cloud_free": (None, [_P, _P]),
    "ssm_viewer_map_update": (_I, [_P, _I, _P, _P, _I, _F, C.POINTER(_I)]),
    "ssm_viewer_map_release": (_I, [_P, _I]),
    "ssm_viewer_map_fetch": (_I, [_P, _P, _I, C.POINTER(_I)]),
    "ssm_map_clear": (_I, [_P]),
    "ssm_map_insert": (_I, [_P, _P, _I]),
    "ssm_map_size": (_I, [_P, C.POINTER(_I)]),
    "ssm_map_stats": (_I, [_P, C.POINTER(C.c_int64)]),
    "ssm_map_export": (_I, [_P, _P, _I, C.POINTER(_I)]),
    "ssm_map_export_table": (_I, [_P, _P, _I, C.POINTER(_I)]),
    "ssm_map_merge_table": (_I, [_P, _P, _I]),
    "ssm_map_export_table_dev": (_I, [_P, _P, _I, C.POINTER(_I)]),
    "ssm_map_merge_table_dev": (_I, [_P, _P, _I]),
    "ssm_comm_get_unique_id": (_I, [_P]),
    "ssm_comm_init_rank": (_I, [_P, _I, _I, _P]),
    "ssm_comm_finalize": (_I, [_P]),
    "ssm_comm_rank": (_I, [_P]),
    "ssm_comm_size": (_I, [_P]),
    "ssm_voxel_allgather": (_I, [_P, _P]),
    "ssm_seq_process": (_I, [_P, C.POINTER(FramesDev), C.POINTER(SeqOutDev)]),
    "ssm_stereo_seq_process": (_I, [_P, C.POINTER(StereoFramesDev), C.POINTER(StereoOutDev)]),
    "ssm_stereo_batch": (_I, [_P]),
    "ssm_tracker_params_default": (None, [C.POINTER(TrackerParams)]),
    "ssm_tracker_create": (_I, [_P, C.POINTER(TrackerParams), C.POINTER(_P)]),
    "ssm_tracker_destroy": (None, [_P]),
    "ssm_tracker_reset": (_I, [_P]),
    "ssm_tracker_run": (_I, [_P, C.POINTER(SeqOutDev), _I, _P, _P]),
    "ssm_tracker_last_error": (C.c_char_p, [_P]),
    "ssm_tracker_stats": (_I, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "ssm_tracker_work": (_I, [_P, C.POINTER(C.c_int64 * 4)]),
    "ssm_quad_track": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _I, C.POINTER(_I)]),
    "ssm_gftt": (_I, [_P, _P, _I, _I, _I, _I, _D, _D, _P, _I, C.POINTER(_I)]),
    "ssm_lk_track": (_I, [_P, _P, _P, _I, _I, _I, _P, _I, _P, _P, _P, _I, _D, _D]),
    "ssm_window_match": (_I, [_P, _P, _P, _I, _P, _P, _I, _I, _I, _F, _P]),
    "ssm_sgbm_params_default": (None, [_P]),
    "ssm_sgbm": (_I, [_P, _P, _P, _I, _I, _I, _P, _I, _P]),
    "ssm_stereo_depth": (_I, [_P, _P, _P, _I, _I, _I, _P] + [_D] * 8 + [_P, _P]),
    "ssm_vo_estimate": (_I, [_P, _P, _I, _P, _P, _I, _P, _P, _I, _P, _P]),
    "ssm_pnp_solve": (_I, [_P, _P, _P, _I, _P, _I, _P, _P, _P, _P]),
    "ssm_segnet_num_layers": (_I, []),
    "ssm_segnet_layer_shape": (_I, [_I, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "ssm_segnet_set_layer": (_I, [_P, _I, _P, _P, _P]),
    "ssm_segnet_forward": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "ssm_segnet_forward_dev": (_I, [_P, _P, _I, _P, _P, _I]),
    "ssm_segnet_logits": (_I, [_P, _P]),
    "ssm_segnet_debug_op": (_I, [_P, _I, _I, _P, _I, _I, _P, _P]),
    "ssm_set_profiling": (_I, [_P, _I]),
    "ssm_get_stage_times": (_I, [_P, _P, _P, _P, _I, C.POINTER(_I)]),
    "ssm_dev_alloc": (_I, [_P, _SZ, C.POINTER(_P)]),
    "ssm_dev_free": (_I, [_P, _P]),
    "ssm_dev_mem_info": (_I, [_P, C.POINTER(_SZ), C.POINTER(_SZ)]),
    "ssm_memcpy_h2d": (_I, [_P, _P, _P, _SZ]),
    "ssm_memcpy_d2h": (_I, [_P, _P, _P, _SZ]),
    "ssm_memcpy_h2d_async": (_I, [_P, _P, _P, _SZ]),
    "ssm_memcpy_d2h_async": (_I, [_P, _P, _P, _SZ]),
    "ssm_host_alloc": (_I, [_SZ, C.POINTER(_P)]),
    "ssm_host_free": (_I, [_P]),
    "ssm_synth_frames_dev": (_I, [_P, _U64, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
}

_lib = None


def load():
    """Load libssm_hip.so or raise.  Never falls back to a CPU implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). semantic_slam_mapping_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
