"""ctypes binding of librgbl_frontend.so (the C ABI declared in include/rgbl_frontend.h).

There is exactly one product library: the hipcc build for gfx950.  `load()` fails loudly when it is
missing or when no HIP device is usable — there is no CPU fallback.  (`bind(path)` is also used by the
test-suite to bind the CPU SIMT emulation of the same sources; the product never calls it with that.)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librgbl_frontend.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])

RGBL_OK = 0
ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_CAPACITY, ERR_OVERFLOW, ERR_EMPTY, ERR_COMM = -1, -2, -3, -4, -5, -6, -7
COMM_ID_BYTES = 128


class RgblError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("rgbl error %d: %s" % (code, msg))
        self.code = code


class ExtractorCfg(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scale_factor", C.c_float), ("nlevels", C.c_int),
                ("ini_th_fast", C.c_int), ("min_th_fast", C.c_int), ("width", C.c_int),
                ("height", C.c_int), ("max_batch", C.c_int)]


class DepthCfg(C.Structure):
    _fields_ = [("proj", C.c_float * 12), ("min_dist", C.c_float), ("max_dist", C.c_float),
                ("mbf", C.c_float), ("method", C.c_int), ("kernel_w", C.c_int), ("kernel_h", C.c_int),
                ("kernel", C.c_uint8 * 81), ("avg_kernel_size", C.c_int), ("nn_search_radius", C.c_float),
                ("width", C.c_int), ("height", C.c_int), ("max_points", C.c_int),
                ("max_keypoints", C.c_int), ("max_batch", C.c_int)]


class KeyframeView(C.Structure):
    _fields_ = [("n", C.c_int), ("desc", C.c_void_p), ("kp_xy", C.c_void_p), ("kp_octave", C.c_void_p),
                ("kp_angle", C.c_void_p), ("uright", C.c_void_p), ("has_mappoint", C.c_void_p),
                ("n_nodes", C.c_int), ("node_id", C.c_void_p), ("node_off", C.c_void_p),
                ("node_feat", C.c_void_p), ("device", C.c_void_p)]


class TriangulationParams(C.Structure):
    _fields_ = [("F12", C.c_float * 9), ("epipole", C.c_float * 2), ("scale_factors2", C.c_void_p),
                ("level_sigma2_2", C.c_void_p), ("n_levels", C.c_int), ("only_stereo", C.c_int),
                ("coarse", C.c_int), ("check_orientation", C.c_int)]


class ProjectionInput(C.Structure):
    """rgbl_projection_input (ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono) as flat arrays)."""
    _fields_ = [("n1", C.c_int), ("valid1", C.c_void_p), ("world_pos1", C.c_void_p), ("mp_desc1", C.c_void_p),
                ("mp_observed1", C.c_void_p), ("octave1", C.c_void_p), ("angle1", C.c_void_p),
                ("n2", C.c_int), ("kp2_xy", C.c_void_p), ("kp2_octave", C.c_void_p), ("kp2_angle", C.c_void_p),
                ("uright2", C.c_void_p), ("desc2", C.c_void_p), ("grid", C.c_float * 6),
                ("Tcw_q", C.c_float * 4), ("Tcw_t", C.c_float * 3), ("Tlw_q", C.c_float * 4), ("Tlw_t", C.c_float * 3),
                ("K", C.c_float * 4), ("mb", C.c_float), ("mbf", C.c_float), ("scale_factors", C.c_void_p),
                ("n_levels", C.c_int), ("th", C.c_float), ("mono", C.c_int), ("check_orientation", C.c_int),
                ("device2", C.c_void_p)]


class KeyFrameProjectionInput(C.Structure):
    """rgbl_keyframe_projection_input (ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist))."""
    _fields_ = [("n1", C.c_int), ("valid1", C.c_void_p), ("world_pos1", C.c_void_p), ("mp_desc1", C.c_void_p),
                ("level1", C.c_void_p), ("angle1", C.c_void_p),
                ("n2", C.c_int), ("kp2_xy", C.c_void_p), ("kp2_octave", C.c_void_p), ("kp2_angle", C.c_void_p),
                ("desc2", C.c_void_p), ("occupied2", C.c_void_p), ("grid", C.c_float * 6),
                ("Tcw_q", C.c_float * 4), ("Tcw_t", C.c_float * 3), ("K", C.c_float * 4), ("scale_factors", C.c_void_p),
                ("n_levels", C.c_int), ("th", C.c_float), ("orb_dist", C.c_int), ("check_orientation", C.c_int),
                ("device2", C.c_void_p)]


class FuseInput(C.Structure):
    """rgbl_fuse_input (the per-point search of ORBmatcher::Fuse(pKF, vpMapPoints, th))."""
    _fields_ = [("n1", C.c_int), ("valid1", C.c_void_p), ("world_pos1", C.c_void_p), ("mp_desc1", C.c_void_p),
                ("level1", C.c_void_p), ("n2", C.c_int), ("kp2_xy", C.c_void_p), ("kp2_octave", C.c_void_p),
                ("uright2", C.c_void_p), ("desc2", C.c_void_p), ("grid", C.c_float * 6), ("Tcw_q", C.c_float * 4),
                ("Tcw_t", C.c_float * 3), ("K", C.c_float * 4), ("bf", C.c_float), ("scale_factors", C.c_void_p),
                ("inv_level_sigma2", C.c_void_p), ("n_levels", C.c_int), ("th", C.c_float), ("device2", C.c_void_p)]


class ProjectSearchInput(C.Structure):
    """rgbl_project_search_input (the per-point search of Fuse(pKF, Scw, ...) and SearchBySim3)."""
    _fields_ = [("n1", C.c_int), ("valid1", C.c_void_p), ("cam_pos1", C.c_void_p), ("mp_desc1", C.c_void_p),
                ("level1", C.c_void_p), ("n2", C.c_int), ("kp2_xy", C.c_void_p), ("kp2_octave", C.c_void_p),
                ("desc2", C.c_void_p), ("grid", C.c_float * 6), ("K", C.c_float * 4), ("scale_factors", C.c_void_p),
                ("n_levels", C.c_int), ("th", C.c_float), ("proj_form", C.c_int), ("max_dist", C.c_int),
                ("device2", C.c_void_p)]


class LocalPointsInput(C.Structure):
    """rgbl_local_points_input (ORBmatcher::SearchByProjection(F, vpMapPoints, th, ...) as flat arrays)."""
    _fields_ = [("n1", C.c_int), ("valid1", C.c_void_p), ("proj1", C.c_void_p), ("level1", C.c_void_p),
                ("view_cos1", C.c_void_p), ("mp_desc1", C.c_void_p), ("mp_observed1", C.c_void_p),
                ("n2", C.c_int), ("kp2_xy", C.c_void_p), ("kp2_octave", C.c_void_p), ("uright2", C.c_void_p),
                ("desc2", C.c_void_p), ("blocked2", C.c_void_p), ("grid", C.c_float * 6), ("scale_factors", C.c_void_p),
                ("n_levels", C.c_int), ("th", C.c_float), ("nnratio", C.c_float), ("device2", C.c_void_p)]


class InitializationInput(C.Structure):
    """rgbl_initialization_input (ORBmatcher::SearchForInitialization as flat arrays)."""
    _fields_ = [("n1", C.c_int), ("kp1_octave", C.c_void_p), ("kp1_angle", C.c_void_p), ("desc1", C.c_void_p),
                ("n2", C.c_int), ("kp2_xy", C.c_void_p), ("kp2_octave", C.c_void_p), ("kp2_angle", C.c_void_p),
                ("desc2", C.c_void_p), ("grid", C.c_float * 6), ("window_size", C.c_int), ("nnratio", C.c_float),
                ("check_orientation", C.c_int)]


# name -> (restype, argtypes); every symbol of include/rgbl_frontend.h
_V, _I, _F, _Z = C.c_void_p, C.c_int, C.c_float, C.c_size_t
SYMBOLS = {
    "rgbl_last_error": (C.c_char_p, []),
    "rgbl_backend": (C.c_char_p, []),
    "rgbl_device_count": (_I, []),
    "rgbl_extractor_create": (_I, [C.POINTER(ExtractorCfg), _I, C.POINTER(_V)]),
    "rgbl_extractor_destroy": (None, [_V]),
    "rgbl_extractor_tables": (_I, [_V, _V, _V, _V, _V, _V, _V]),
    "rgbl_extractor_max_keypoints": (_I, [_V]),
    "rgbl_extract": (_I, [_V, _V, _I, _I, _I, _I, _I, _V, _V, _I, C.POINTER(_I), C.POINTER(_I)]),
    "rgbl_extract_begin": (_I, [_V, _V, _I, _I, _I, _I, _I]),
    "rgbl_extract_cancel": (_I, [_V]),
    "rgbl_extract_batch": (_I, [_V, _V, _I, _I, _I, _I, _Z, _I, _I, _V, _V, _I, _V, _V]),
    "rgbl_extract_batch_device": (_I, [_V, _V, _I, _I, _I, _I, _Z, _I, _I, _V, _V, _I, _V, _V]),
    "rgbl_extractor_sync": (_I, [_V]),
    "rgbl_extractor_level_size": (_I, [_V, _I, C.POINTER(_I), C.POINTER(_I)]),
    "rgbl_extractor_get_level": (_I, [_V, _I, _I, _I, _I, _V, _I]),
    "rgbl_extractor_get_candidates": (_I, [_V, _I, _I, _V, _I, C.POINTER(_I)]),
    "rgbl_cvt_gray_batch_device": (_I, [_V, _V, _I, _I, _I, _I, _I, _I, _Z, _V, _I, _Z]),
    "rgbl_extract_color": (_I, [_V, _V, _I, _I, _I, _I, _I, _I, _I, _V, _V, _I, C.POINTER(_I), C.POINTER(_I), _V, _I]),
    "rgbl_stereo_matches": (_I, [_V, _V, _V, _V, _I, _V, _V, _I, _F, _F, _V, _V]),
    "rgbl_stereo_matches_batch_device": (_I, [_V, _V, _I, _V, _V, _V, _V, _V, _V, _I, _F, _F, _V, _V]),
    "rgbl_extractor_debug_stamps": (_I, [_V, _V, _I]),
    "rgbl_selftest_wrappers": (_I, [_I, _I, C.c_uint]),
    "rgbl_extractor_set_stream": (_I, [_V, _V]),
    "rgbl_extractor_profile": (_I, [_V, _I]),
    "rgbl_extractor_profile_read": (_I, [_V, _V, _V, _V, _I]),
    "rgbl_extractor_profile_samples": (_I, [_V, _I, _V, _I]),
    "rgbl_depth_create": (_I, [C.POINTER(DepthCfg), _I, C.POINTER(_V)]),
    "rgbl_depth_destroy": (None, [_V]),
    "rgbl_projection_matrix": (None, [_V, _V, _V]),
    "rgbl_structuring_element": (_I, [_I, _I, _I, _V]),
    "rgbl_depth_compute_xyzi": (_I, [_V, _V, _I, _I, _I, _V, _V, _I, _V, _V, _V, _V]),
    "rgbl_depth_project_xyzi_batch_device": (_I, [_V, _V, _I, _I, _Z, _I, _I, _V]),
    "rgbl_depth_compute": (_I, [_V, _V, _I, _I, _I, _I, _V, _V, _I, _V, _V, _V, _V]),
    "rgbl_depth_prefetch": (_I, [_V, _V, _I, _I, _I, _I]),
    "rgbl_depth_prefetch_xyzi": (_I, [_V, _V, _I, _I, _I]),
    "rgbl_depth_prefetch_cancel": (_I, [_V]),
    "rgbl_depth_batch_device": (_I, [_V, _V, _I, _I, _I, _Z, _I, _I, _V, _V, _I, _V, _V, _V, _V]),
    "rgbl_depth_project_batch_device": (_I, [_V, _V, _I, _I, _I, _Z, _I, _I, _V]),
    "rgbl_depth_gather_batch_device": (_I, [_V, _I, _I, _I, _V, _V, _I, _V, _V, _V]),
    "rgbl_depth_sync": (_I, [_V]),
    "rgbl_depth_stream": (_V, [_V]),
    "rgbl_undistort_points": (_I, [_V, _V, _I, _V, _V, _I, _V]),
    "rgbl_undistort_keypoints_batch_device": (_I, [_V, _V, _V, _I, _I, _V, _V, _I, _V]),
    "rgbl_extractor_stream": (_V, [_V]),
    "rgbl_extractor_aux_stream": (_V, [_V]),
    "rgbl_matcher_stream": (_V, [_V]),
    "rgbl_stream_wait": (_I, [_V, _V]),
    "rgbl_pack_records_device": (_I, [_V, _V, _V, _V, _V, _V, _I, _I, C.c_longlong, C.c_longlong, _V, _V, _V]),
    "rgbl_comm_available": (_I, []),
    "rgbl_comm_unique_id": (_I, [_V]),
    "rgbl_comm_create": (_I, [_V, _I, _I, _I, C.POINTER(_V)]),
    "rgbl_comm_destroy": (None, [_V]),
    "rgbl_comm_info": (_I, [_V, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "rgbl_gather_create": (_I, [_V, _I, _I, _I, _I, _V, C.POINTER(_V)]),
    "rgbl_gather_destroy": (None, [_V]),
    "rgbl_gather_stream": (_V, [_V]),
    "rgbl_gather_set_loopback": (_I, [_V, _I]),
    "rgbl_gather_pack": (_I, [_V, _I, _V, _V, _V, _V, _V, _V, _I, _V]),
    "rgbl_gather_exchange": (_I, [_V, _I]),
    "rgbl_gather_sync": (_I, [_V]),
    "rgbl_gather_result": (_I, [_V, _I, C.POINTER(_V), C.POINTER(_V), C.POINTER(C.c_longlong)]),
    "rgbl_gather_copy_result": (_I, [_V, _I, _V, C.c_longlong]),
    "rgbl_event_create": (_I, [C.POINTER(_V)]),
    "rgbl_event_destroy": (None, [_V]),
    "rgbl_event_record": (_I, [_V, _V]),
    "rgbl_event_wait": (_I, [_V, _V]),
    "rgbl_stream_create": (_I, [C.POINTER(_V), _I]),
    "rgbl_stream_create_on": (_I, [_I, C.POINTER(_V), _I]),
    "rgbl_stream_destroy": (None, [_V]),
    "rgbl_depth_set_stream": (_I, [_V, _V]),
    "rgbl_depth_set_sparse": (_I, [_V, _I]),
    "rgbl_depth_profile": (_I, [_V, _I]),
    "rgbl_depth_profile_read": (_I, [_V, _V, _V, _V, _I]),
    "rgbl_depth_profile_samples": (_I, [_V, _I, _V, _I]),
    "rgbl_matcher_create": (_I, [_I, C.POINTER(_V)]),
    "rgbl_matcher_destroy": (None, [_V]),
    "rgbl_matcher_sync": (_I, [_V]),
    "rgbl_matcher_set_stream": (_I, [_V, _V]),
    "rgbl_matcher_profile": (_I, [_V, _I]),
    "rgbl_matcher_profile_read": (_I, [_V, _V, _V, _V, _I]),
    "rgbl_matcher_profile_samples": (_I, [_V, _I, _V, _I]),
    "rgbl_descriptor_distance": (_I, [_V, _V]),
    "rgbl_matcher_acquire": (_I, [_I, C.POINTER(_V)]),
    "rgbl_matcher_release": (None, [_V]),
    "rgbl_matcher_pool_size": (_I, []),
    "rgbl_matcher_pool_clear": (_I, []),
    "rgbl_hamming_bf": (_I, [_V, _V, _I, _V, _I, _V, _V, _V]),
    "rgbl_stereo_fisheye_matches": (_I, [_V, _V, _I, _I, _V, _I, _I, _V, _V, _V]),
    "rgbl_hamming_bf_batch_device": (_I, [_V, _V, _V, _I, _V, _V, _I, _V, _V, _V]),
    "rgbl_search_triangulation": (_I, [_V, C.POINTER(KeyframeView), C.POINTER(KeyframeView),
                                       C.POINTER(TriangulationParams), _V, C.POINTER(_I)]),
    "rgbl_search_by_bow": (_I, [_V, _V, _V, _F, _I, _V, C.POINTER(_I)]),
    "rgbl_search_by_bow_rig": (_I, [_V, _V, _V, _I, _F, _I, _V, C.POINTER(_I)]),
    "rgbl_search_by_bow_keyframes": (_I, [_V, _V, _V, _F, _I, _V, C.POINTER(_I)]),
    "rgbl_search_by_projection": (_I, [_V, _V, _V, C.POINTER(_I)]),
    "rgbl_search_local_points": (_I, [_V, _V, _V, C.POINTER(_I)]),
    "rgbl_search_for_initialization": (_I, [_V, _V, _V, _V, C.POINTER(_I)]),
    "rgbl_fuse_search": (_I, [_V, _V, _V, _V]),
    "rgbl_project_search": (_I, [_V, _V, _V, _V]),
    "rgbl_search_by_projection_sim3": (_I, [_V, _V, _V, _V, C.POINTER(_I)]),
    "rgbl_distinctive_descriptors": (_I, [_V, _V, _V, _I, _V]),
    "rgbl_search_by_projection_keyframe": (_I, [_V, _V, _V, C.POINTER(_I)]),
    "rgbl_vocabulary_load_text": (_I, [C.c_char_p, _I, C.POINTER(_V)]),
    "rgbl_vocabulary_create": (_I, [_I, _I, _V, _V, _V, _V, _V, _I, C.POINTER(_V)]),
    "rgbl_vocabulary_destroy": (None, [_V]),
    "rgbl_vocabulary_info": (_I, [_V, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "rgbl_bow_transform": (_I, [_V, _V, _I, _I, _V, _V, _I, C.POINTER(_I), _V, _V, _V, _I, C.POINTER(_I)]),
    "rgbl_bow_descend_batch_device": (_I, [_V, _V, _V, _V, _I, _I, _I, _V, _V, _V]),
    "rgbl_fundamental": (None, [_V, _V, _V, _V, _V]),
    "rgbl_device_frame_create": (_I, [_I, _I, C.POINTER(_V)]),
    "rgbl_device_frame_destroy": (None, [_V]),
    "rgbl_device_frame_upload": (_I, [_V, _I, _V, _V, _V, _V]),
    "rgbl_device_frame_capture": (_I, [_V, _V, _I, _I, _V, _V, _V, _I]),
    "rgbl_device_frame_set_feature_vector": (_I, [_V, _I, _V, _V]),
    "rgbl_device_frame_size": (_I, [_V]),
    "rgbl_device_frame_set_grid": (_I, [_V, _V]),
    "rgbl_device_frame_download": (_I, [_V, _V, _V, _V, _V]),
    "rgbl_bow_transform_frame": (_I, [_V, _V, _I, _V, _V, _I, C.POINTER(_I), _V, _V, _V, _I, C.POINTER(_I)]),
}


def bind(path):
    """dlopen `path` and attach the prototypes of every symbol in include/rgbl_frontend.h."""
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    return lib


_lib = None


def load():
    """The product library. Raises if it has not been built (run __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RgblError(ERR_NO_DEVICE, "%s not built; run `python -c 'import __graft_entry__ as g; g.build()'` "
                                           "(there is no CPU fallback)" % LIB_PATH)
        _lib = bind(LIB_PATH)
    return _lib


def check(lib, rc):
    if rc != RGBL_OK:
        raise RgblError(rc, lib.rgbl_last_error().decode("utf-8", "replace"))
    return rc


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def read_profile(lib, fn, handle):
    names = (C.c_char_p * 32)()
    ms = (C.c_double * 32)()
    cnt = (C.c_long * 32)()
    n = fn(handle, C.cast(names, C.c_void_p), C.cast(ms, C.c_void_p), C.cast(cnt, C.c_void_p), 32)
    return {names[i].decode(): (ms[i], cnt[i]) for i in range(min(n, 32))}


def read_profile_samples(lib, fn_read, fn_samples, handle):
    """{kernel: [every launch's duration in ms, launch order]} since profiling was switched on."""
    names = (C.c_char_p * 32)()
    n = fn_read(handle, C.cast(names, C.c_void_p), None, None, 32)
    out = {}
    for i in range(min(n, 32)):
        k = fn_samples(handle, i, None, 0)
        buf = (C.c_float * max(k, 1))()
        k = fn_samples(handle, i, C.cast(buf, C.c_void_p), k)
        out[names[i].decode()] = [float(buf[j]) for j in range(k)]
    return out
