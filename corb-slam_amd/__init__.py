"""corb_slam_amd -- thin ctypes harness over libcorb_accel.so (the MI355X-native CORB-SLAM hot path).

The product is the C-ABI library (include/corb_accel.h) plus the C++ adapter in host/; this module only
drives it from tests/, bench.py and __graft_entry__.py.  Class / method names mirror the reference
(ORBextractor::operator(), ORBmatcher::SearchByBoW / SearchForTriangulation,
Optimizer::GlobalBundleAdjustemnt).  There is NO fallback: if the HIP library is missing or no gfx950
device is visible, every call raises CorbError.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcorb_accel.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
EDGE_DTYPE = np.dtype([("pose", "<i4"), ("point", "<i4"), ("u", "<f4"), ("v", "<f4"),
                       ("ur", "<f4"), ("inv_sigma2", "<f4")])

EXPORTS = [
    "corb_last_error", "corb_device_count", "corb_version", "corb_abi_version", "corb_warmup", "corb_release_scratch", "corb_pinned_alloc", "corb_pinned_free",
    "corb_orb_create", "corb_orb_destroy", "corb_orb_extract", "corb_orb_tables", "corb_orb_pyramid_level",
    "corb_orb_upload", "corb_orb_run", "corb_orb_sync", "corb_orb_fetch", "corb_orb_fetch_candidates",
    "corb_orb_device_image", "corb_orb_upload_batch", "corb_orb_capacity", "corb_orb_fetch_batch", "corb_stereo_upload_batch", "corb_stereo_fetch_matches_batch", "corb_orb_profile", "corb_orb_profile_read",
    "corb_stereo_create", "corb_stereo_destroy", "corb_stereo_orb", "corb_stereo_upload", "corb_stereo_run",
    "corb_stereo_sync", "corb_stereo_fetch_matches", "corb_stereo_frame_layout", "corb_stereo_frames", "corb_track_search_reloc", "corb_search_by_sim3_store", "corb_mp_store_set_counters", "corb_mp_store_get_counters", "corb_mp_store_set_scratch", "corb_mp_store_get_scratch", "corb_mp_store_replace",
    "corb_descriptor_distance", "corb_search_by_bow", "corb_search_for_triangulation", "corb_ba_solve", "corb_ba_solve_ex", "corb_ba_solve_staged",
    "corb_search_by_projection_map", "corb_search_by_projection_frame", "corb_pose_optimization_batch",
    "corb_search_by_projection_reloc", "corb_search_by_projection_scw", "corb_search_for_initialization", "corb_fuse", "corb_search_by_sim3", "corb_distinctive_descriptors", "corb_rebase_map", "corb_optimize_sim3", "corb_optimize_essential_graph",
    "corb_kf_store_create", "corb_kf_store_destroy", "corb_kf_store_record_bytes", "corb_kf_store_put_from_stereo", "corb_kf_store_put_host", "corb_kf_store_set_bow",
    "corb_kf_store_set_flags", "corb_kf_store_get", "corb_search_by_bow_slots", "corb_search_for_triangulation_slots",
    "corb_comm_unique_id", "corb_comm_create", "corb_comm_destroy", "corb_map_push",
    "corb_kf_store_set_meta", "corb_kf_store_get_meta", "corb_kf_store_set_map_points", "corb_kf_store_get_map_points",
    "corb_mp_store_create", "corb_mp_store_destroy", "corb_mp_store_record_bytes", "corb_mp_store_put_host", "corb_mp_store_get",
    "corb_comm_create_local", "corb_comm_rank", "corb_comm_world", "corb_map_push_ex", "corb_map_push_plan", "corb_map_push_messages", "corb_comm_test_rccl_exchange", "corb_map_push_setup", "corb_map_push_begin", "corb_map_push_wait", "corb_rebase_map_store", "corb_ba_solve_store", "corb_local_ba_store", "corb_fuse_store", "corb_search_by_projection_scw_store", "corb_ba_solve_devflat", "corb_kf_store_put_batch", "corb_spd_solve",
    "corb_mp_store_build_index", "corb_kf_store_count", "corb_track_search_last_frame", "corb_track_pose_optimization", "corb_track_search_local_points", "corb_kf_store_put_frame",
]


class CorbError(RuntimeError):
    pass


class OrbConfig(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("max_images", C.c_int32), ("device", C.c_int32)]


class StereoConfig(C.Structure):
    _fields_ = [("orb", OrbConfig), ("max_frames", C.c_int32), ("fx", C.c_float), ("bf", C.c_float)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("total_ms", C.c_double), ("launches", C.c_int64)]


class _FeatVec(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("node_id", C.c_void_p), ("offset", C.c_void_p), ("idx", C.c_void_p)]


class _BowSide(C.Structure):
    _fields_ = [("desc", C.c_void_p), ("angle", C.c_void_p), ("valid", C.c_void_p), ("n", C.c_int32), ("fv", _FeatVec)]


class _TriSide(C.Structure):
    _fields_ = [("desc", C.c_void_p), ("kp", C.c_void_p), ("u_right", C.c_void_p), ("has_mappoint", C.c_void_p),
                ("n", C.c_int32), ("fv", _FeatVec)]


MP_DTYPE = np.dtype([("world", "<f4", 3), ("normal", "<f4", 3), ("min_distance", "<f4"), ("max_distance", "<f4"), ("angle", "<f4"),
                     ("valid", "u1"), ("pad", "u1", 3)])


class _KeyFrameView(C.Structure):
    _fields_ = [("keys_un", C.c_void_p), ("u_right", C.c_void_p), ("desc", C.c_void_p), ("n", C.c_int32),
                ("min_x", C.c_float), ("min_y", C.c_float), ("max_x", C.c_float), ("max_y", C.c_float),
                ("scale", C.c_void_p), ("inv_level_sigma2", C.c_void_p), ("nlevels", C.c_int32), ("log_scale_factor", C.c_float),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("bf", C.c_float)]


class _Sim3Problem(C.Structure):
    _fields_ = [("n", C.c_int32), ("p1c", C.c_void_p), ("p2c", C.c_void_p), ("obs1", C.c_void_p), ("obs2", C.c_void_p),
                ("inv_sigma2_1", C.c_void_p), ("inv_sigma2_2", C.c_void_p)] + [(k, C.c_float) for k in ("fx1", "fy1", "cx1", "cy1", "fx2", "fy2", "cx2", "cy2")]


class _PoseOptFrame(C.Structure):
    _fields_ = [("Tcw", C.c_void_p), ("n_obs", C.c_int32), ("points", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p),
                ("u_right", C.c_void_p), ("inv_sigma2", C.c_void_p),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("bf", C.c_float)]


class _BAProblem(C.Structure):
    _fields_ = [("n_poses", C.c_int32), ("n_points", C.c_int32), ("n_edges", C.c_int32),
                ("poses", C.c_void_p), ("pose_fixed", C.c_void_p), ("points", C.c_void_p),
                ("point_fixed", C.c_void_p), ("edges", C.c_void_p),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("bf", C.c_float), ("intr", C.c_void_p)]


class StereoFrameLayout(C.Structure):
    _fields_ = [("capacity", C.c_int32), ("frame_bytes", C.c_int32), ("off_kp_left", C.c_int32), ("off_kp_right", C.c_int32),
                ("off_desc_left", C.c_int32), ("off_desc_right", C.c_int32), ("off_u_right", C.c_int32), ("off_depth", C.c_int32)]


class StereoFrameTiming(C.Structure):
    _fields_ = [("ms_upload", C.c_float), ("ms_kernels", C.c_float), ("ms_download", C.c_float)]


class _BAResult(C.Structure):
    _fields_ = [("poses", C.c_void_p), ("points", C.c_void_p), ("chi2", C.c_void_p), ("lam", C.c_void_p),
                ("iters_done", C.c_int32), ("trials_total", C.c_int32),
                ("ms_total", C.c_double), ("ms_build", C.c_double), ("ms_schur", C.c_double),
                ("ms_solve", C.c_double), ("ms_update", C.c_double), ("solver_used", C.c_int32), ("pcg_iterations", C.c_int32),
                ("free_poses", C.c_int32), ("free_points", C.c_int32), ("active_edges", C.c_int32), ("nnz_blocks", C.c_int64), ("schur_pairs", C.c_int64),
                ("pc_block", C.c_int32), ("pc_levels", C.c_int32),
                ("pcg_residual_max", C.c_double), ("pcg_residual_last", C.c_double), ("grad_inf", C.c_double), ("pcg_refined_trials", C.c_int32), ("reserved0", C.c_int32)]


KF_META_DTYPE = np.dtype([("id", "<u8"), ("client_id", "<i4"), ("flags", "<u4"), ("fx", "<f4"), ("fy", "<f4"), ("cx", "<f4"), ("cy", "<f4"), ("bf", "<f4"),
                          ("nlevels", "<i4"), ("Tcw", "<f4", 16), ("TcwGBA", "<f4", 16), ("ba_global_for_kf", "<u8"), ("inv_level_sigma2", "<f4", 16)])
MP_RECORD_DTYPE = np.dtype([("id", "<u8"), ("ref_kf_id", "<u8"), ("descriptor", "u1", 32), ("client_id", "<i4"), ("n_obs", "<i4"), ("flags", "<u4"), ("world_pos", "<f4", 3),
                            ("normal", "<f4", 3), ("min_distance", "<f4"), ("max_distance", "<f4"), ("pos_gba", "<f4", 3), ("ba_global_for_kf", "<u8")], align=True)
assert KF_META_DTYPE.itemsize == 240 and MP_RECORD_DTYPE.itemsize == 112 and MP_RECORD_DTYPE.fields["descriptor"][1] == 16
MP_COUNTERS_DTYPE = np.dtype([("n_visible", "<i4"), ("n_found", "<i4"), ("replaced_by", "<u8")])
MP_SCRATCH_DTYPE = np.dtype([("first_kf_id", "<i8"), ("first_frame", "<i8"), ("track_reference_for_frame", "<u8"), ("last_frame_seen", "<u8"), ("ba_local_for_kf", "<u8"),
                             ("fuse_candidate_for_kf", "<u8"), ("loop_point_for_kf", "<u8"), ("corrected_by_kf", "<u8"), ("corrected_reference", "<u8"),
                             ("track_proj_x", "<f4"), ("track_proj_y", "<f4"), ("track_proj_xr", "<f4"), ("track_view_cos", "<f4"), ("track_scale_level", "<i4"),
                             ("n_obs_weight", "<i4"), ("track_in_view", "u1"), ("pad", "u1", 7)])      # CorbMapPointScratch (104 bytes)
PUSH_HEADER_DTYPE = np.dtype([("status", "<i4"), ("n_kf", "<i4"), ("n_mp", "<i4"), ("kf_record_bytes", "<i4"), ("mp_record_bytes", "<i4")])
NO_MAP_POINT = 0xFFFFFFFFFFFFFFFF
KF_BAD, KF_FIXED, MP_BAD, MP_FIXED = 1, 2, 1, 2


class _MapPush(C.Structure):
    _fields_ = [("kf", C.c_void_p), ("kf_slots", C.c_void_p), ("n_kf", C.c_int32), ("mp", C.c_void_p), ("mp_slots", C.c_void_p), ("n_mp", C.c_int32),
                ("kf_dst_first", C.c_void_p), ("mp_dst_first", C.c_void_p), ("kf_recv_counts", C.c_void_p), ("mp_recv_counts", C.c_void_p)]


TRACKED_DTYPE = np.dtype([("proj_x", "<f4"), ("proj_y", "<f4"), ("proj_xr", "<f4"), ("view_cos", "<f4"), ("level", "<i4"),
                          ("valid", "u1"), ("claims", "u1"), ("pad", "u1", 2)])
LAST_DTYPE = np.dtype([("world", "<f4", 3), ("angle", "<f4"), ("octave", "<i4"), ("valid", "u1"), ("claims", "u1"), ("pad", "u1", 2)])


class _FrameView(C.Structure):
    _fields_ = [("keys_un", C.c_void_p), ("u_right", C.c_void_p), ("desc", C.c_void_p), ("n", C.c_int32), ("claimed", C.c_void_p),
                ("min_x", C.c_float), ("min_y", C.c_float), ("max_x", C.c_float), ("max_y", C.c_float), ("scale", C.c_void_p), ("nlevels", C.c_int32)]


class BAStage(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("robust", C.c_int32), ("chi2_mono", C.c_float), ("chi2_stereo", C.c_float),
                ("check_depth", C.c_int32), ("recompute_inactive", C.c_int32), ("allow_reactivate", C.c_int32),
                ("reset_estimates", C.c_int32), ("float_compare", C.c_int32), ("huber_mono", C.c_float), ("huber_stereo", C.c_float)]


# Optimizer::LocalBundleAdjustment (Optimizer.cc:487-838) and Optimizer::PoseOptimization (272-485) as stage lists
_HM, _HS = float(np.float32(np.sqrt(5.991))), float(np.float32(np.sqrt(7.815)))
# (the final "Check inlier observations" pass of LocalBundleAdjustment tests EVERY edge -- also those set to level 1 after the first round -- with
# its last computed chi2 and a fresh depth: allow_reactivate = 1 on the second stage; Optimizer.cc:763-790)
LOCAL_BA_STAGES = [(5, 1, 5.991, 7.815, 1, 0, 0, 0, 0, _HM, _HS), (10, 0, 5.991, 7.815, 1, 0, 1, 0, 0, _HM, _HS)]
POSE_OPT_STAGES = [(10, 1, 5.991, 7.815, 0, 1, 1, 1, 1, _HM, _HS)] * 3 + [(10, 0, 5.991, 7.815, 0, 1, 1, 1, 1, _HM, _HS)]


class BAOptions(C.Structure):
    _fields_ = [("solver", C.c_int32), ("pcg_tol", C.c_double), ("pcg_max_iter", C.c_int32), ("pc_block", C.c_int32), ("pc_multilevel", C.c_int32), ("scale_factor", C.c_float)]


_lib = None


ABI_VERSION = 6          # = CORB_ABI_VERSION of the header the ctypes structures below mirror


def load():
    """dlopen libcorb_accel.so; raises CorbError if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CorbError("libcorb_accel.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                        "there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    if not hasattr(L, "corb_abi_version") or L.corb_abi_version() != ABI_VERSION:       # include/corb_accel.h: CORB_ABI_VERSION -- the structs carry no size fields
        raise CorbError("libcorb_accel.so was built from another corb_accel.h (struct layout version %s, this harness mirrors %d): rebuild it"
                        % (L.corb_abi_version() if hasattr(L, "corb_abi_version") else "< 5", ABI_VERSION))
    L.corb_last_error.restype = C.c_char_p
    L.corb_stereo_orb.restype = C.c_void_p
    L.corb_stereo_orb.argtypes = [C.c_void_p]
    L.corb_orb_create.argtypes = [C.POINTER(OrbConfig), C.POINTER(C.c_void_p)]
    L.corb_orb_destroy.argtypes = [C.c_void_p]
    L.corb_orb_destroy.restype = None
    L.corb_orb_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    L.corb_orb_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 6
    L.corb_orb_pyramid_level.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.corb_orb_upload.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.corb_orb_run.argtypes = [C.c_void_p, C.c_int]
    L.corb_orb_sync.argtypes = [C.c_void_p]
    L.corb_orb_fetch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    L.corb_orb_fetch_candidates.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    L.corb_orb_upload_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.corb_orb_capacity.argtypes = [C.c_void_p]
    L.corb_orb_fetch_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.corb_stereo_upload_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.corb_stereo_frame_layout.argtypes = [C.c_void_p, C.POINTER(StereoFrameLayout)]
    L.corb_stereo_frames.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.corb_stereo_fetch_matches_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.corb_orb_device_image.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.corb_orb_profile.argtypes = [C.c_void_p, C.c_int]
    L.corb_orb_profile_read.argtypes = [C.c_void_p, C.POINTER(KernelTime), C.c_int, C.POINTER(C.c_int)]
    L.corb_stereo_create.argtypes = [C.POINTER(StereoConfig), C.POINTER(C.c_void_p)]
    L.corb_stereo_destroy.argtypes = [C.c_void_p]
    L.corb_stereo_destroy.restype = None
    L.corb_stereo_upload.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.corb_stereo_run.argtypes = [C.c_void_p, C.c_int]
    L.corb_stereo_sync.argtypes = [C.c_void_p]
    L.corb_stereo_fetch_matches.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.corb_descriptor_distance.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.corb_search_by_bow.argtypes = [C.c_int, C.POINTER(_BowSide), C.POINTER(_BowSide), C.c_float, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_int]
    L.corb_search_for_triangulation.argtypes = [C.POINTER(_TriSide), C.POINTER(_TriSide), C.c_void_p, C.c_float, C.c_float,
                                                C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_int]
    L.corb_search_by_projection_map.argtypes = [C.POINTER(_FrameView), C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.POINTER(C.c_int), C.c_int]
    L.corb_search_by_projection_frame.argtypes = [C.POINTER(_FrameView), C.c_void_p, C.c_void_p] + [C.c_float] * 6 + [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_int]
    L.corb_ba_solve.argtypes = [C.POINTER(_BAProblem), C.c_int, C.c_int, C.c_void_p, C.POINTER(_BAResult), C.c_int]
    L.corb_ba_solve_ex.argtypes = [C.POINTER(_BAProblem), C.c_int, C.c_int, C.c_void_p, C.POINTER(_BAResult), C.c_int, C.POINTER(BAOptions)]
    L.corb_ba_solve_devflat.argtypes = [C.POINTER(_BAProblem), C.c_int, C.c_int, C.POINTER(_BAResult), C.c_int, C.POINTER(BAOptions)]
    L.corb_spd_solve.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_int]
    L.corb_ba_solve_staged.argtypes = [C.POINTER(_BAProblem), C.POINTER(BAStage), C.c_int, C.c_void_p, C.POINTER(_BAResult), C.c_void_p, C.c_int, C.POINTER(BAOptions)]
    L.corb_search_by_projection_reloc.restype = C.c_int
    L.corb_search_by_projection_reloc.argtypes = [C.POINTER(_KeyFrameView), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_int]
    L.corb_search_for_initialization.restype = C.c_int
    L.corb_search_for_initialization.argtypes = [C.POINTER(_FrameView), C.POINTER(_FrameView), C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_int]
    L.corb_search_by_projection_scw.restype = C.c_int
    L.corb_search_by_projection_scw.argtypes = [C.POINTER(_KeyFrameView), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.POINTER(C.c_int), C.c_int]
    L.corb_fuse.restype = C.c_int
    L.corb_fuse.argtypes = [C.POINTER(_KeyFrameView), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_int]
    L.corb_search_by_sim3.restype = C.c_int
    L.corb_search_by_sim3.argtypes = [C.POINTER(_KeyFrameView), C.POINTER(_KeyFrameView)] + [C.c_void_p] * 6 + [C.c_float, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.POINTER(C.c_int), C.c_int]
    L.corb_distinctive_descriptors.restype = C.c_int
    L.corb_distinctive_descriptors.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.corb_rebase_map.restype = C.c_int
    L.corb_rebase_map.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
    L.corb_optimize_essential_graph.restype = C.c_int
    L.corb_optimize_essential_graph.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_int]
    L.corb_optimize_sim3.restype = C.c_int
    L.corb_optimize_sim3.argtypes = [C.POINTER(_Sim3Problem), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.corb_pose_optimization_batch.restype = C.c_int
    L.corb_pose_optimization_batch.argtypes = [C.POINTER(_PoseOptFrame), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.corb_kf_store_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.corb_kf_store_destroy.argtypes = [C.c_void_p]; L.corb_kf_store_destroy.restype = None
    L.corb_kf_store_record_bytes.argtypes = [C.c_void_p]
    L.corb_kf_store_put_from_stereo.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_uint64]
    L.corb_kf_store_put_host.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64]
    L.corb_kf_store_set_bow.argtypes = [C.c_void_p, C.c_int, C.POINTER(_FeatVec)]
    L.corb_kf_store_set_flags.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.corb_kf_store_get.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
    L.corb_search_by_bow_slots.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
    L.corb_search_for_triangulation_slots.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                      C.c_void_p, C.POINTER(C.c_int)]
    L.corb_comm_unique_id.argtypes = [C.c_void_p]
    L.corb_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.corb_comm_destroy.argtypes = [C.c_void_p]; L.corb_comm_destroy.restype = None
    L.corb_map_push.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.corb_kf_store_set_meta.argtypes = [C.c_void_p, C.c_int, C.c_void_p]; L.corb_kf_store_get_meta.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.corb_kf_store_set_map_points.argtypes = [C.c_void_p, C.c_int, C.c_void_p]; L.corb_kf_store_get_map_points.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.corb_kf_store_put_batch.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 7
    L.corb_mp_store_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.corb_mp_store_destroy.argtypes = [C.c_void_p]; L.corb_mp_store_destroy.restype = None
    L.corb_mp_store_record_bytes.argtypes = [C.c_void_p]
    L.corb_mp_store_put_host.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.corb_mp_store_get.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.corb_comm_create_local.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    L.corb_comm_rank.argtypes = [C.c_void_p]; L.corb_comm_world.argtypes = [C.c_void_p]
    L.corb_map_push_ex.argtypes = [C.c_void_p, C.POINTER(_MapPush), C.c_int]
    L.corb_map_push_plan.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    L.corb_rebase_map_store.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.corb_ba_solve_store.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.POINTER(_BAResult), C.POINTER(BAOptions)]
    L.corb_fuse_store.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(TrackCamera), C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    L.corb_local_ba_store.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(BAStage), C.c_int, C.c_float, C.c_int, C.c_void_p,
                                      C.POINTER(_BAResult), C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(BAOptions)]
    _lib = L
    return L


def _chk(rc, what):
    if rc != 0:
        raise CorbError("%s failed (%d): %s" % (what, rc, load().corb_last_error().decode(errors="replace")))


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def device_count():
    return load().corb_device_count()


class ORBextractor:
    """Mirror of ORB_SLAM2::ORBextractor (corbslam_client/include/ORBextractor.h:45-114)."""

    def __init__(self, nfeatures=2000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7,
                 width=1241, height=376, max_images=1, device=0, _handle=None):
        self.L = load()
        self.nlevels, self.width, self.height, self.max_images = nlevels, width, height, max_images
        self._owned = _handle is None
        if _handle is None:
            cfg = OrbConfig(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, width, height, max_images, device)
            h = C.c_void_p()
            _chk(self.L.corb_orb_create(C.byref(cfg), C.byref(h)), "corb_orb_create")
            self.h = h
        else:
            self.h = C.c_void_p(_handle)
        self.cap = self._out_cap(nfeatures)

    def _out_cap(self, nfeatures):
        return nfeatures + 16 * self.nlevels + 256

    def close(self):
        if self._owned and self.h:
            self.L.corb_orb_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # getters (ORBextractor.h:62-82)
    def tables(self):
        n = self.nlevels
        sc, isc, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        quota = np.zeros(n, np.int32); umax = np.zeros(16, np.int32)
        _chk(self.L.corb_orb_tables(self.h, _p(sc), _p(isc), _p(s2), _p(is2), _p(quota), _p(umax)), "corb_orb_tables")
        return dict(scale=sc, inv_scale=isc, sigma2=s2, inv_sigma2=is2, quota=quota, umax=umax)

    def GetScaleFactors(self):
        return self.tables()["scale"]

    def GetInverseScaleFactors(self):
        return self.tables()["inv_scale"]

    def __call__(self, image):
        """operator()(image, mask, keypoints, descriptors): returns (keypoints[KP_DTYPE], descriptors[n,32])."""
        image = np.ascontiguousarray(image, np.uint8)
        kps = np.zeros(self.cap, KP_DTYPE); desc = np.zeros((self.cap, 32), np.uint8); n = C.c_int()
        if image.size == 0:
            _chk(self.L.corb_orb_extract(self.h, None, 0, 0, 0, _p(kps), _p(desc), self.cap, C.byref(n)), "corb_orb_extract")
        else:
            h, w = image.shape
            _chk(self.L.corb_orb_extract(self.h, _p(image), w, h, w, _p(kps), _p(desc), self.cap, C.byref(n)), "corb_orb_extract")
        return kps[: n.value].copy(), desc[: n.value].copy()

    # batched path
    def upload(self, slot, image):
        image = np.ascontiguousarray(image, np.uint8)
        assert image.shape == (self.height, self.width)
        _chk(self.L.corb_orb_upload(self.h, slot, _p(image), self.width), "corb_orb_upload")
        self._keep = image

    def run(self, n_images):
        _chk(self.L.corb_orb_run(self.h, n_images), "corb_orb_run")

    def sync(self):
        _chk(self.L.corb_orb_sync(self.h), "corb_orb_sync")

    def fetch(self, slot):
        kps = np.zeros(self.cap, KP_DTYPE); desc = np.zeros((self.cap, 32), np.uint8); n = C.c_int()
        _chk(self.L.corb_orb_fetch(self.h, slot, _p(kps), _p(desc), self.cap, C.byref(n)), "corb_orb_fetch")
        return kps[: n.value].copy(), desc[: n.value].copy()

    def pyramid_level(self, slot, level, blurred=False):
        w, h = C.c_int(), C.c_int()
        _chk(self.L.corb_orb_pyramid_level(self.h, slot, level, int(blurred), None, 0, C.byref(w), C.byref(h)), "pyramid_level")
        out = np.zeros((h.value, w.value), np.uint8)
        _chk(self.L.corb_orb_pyramid_level(self.h, slot, level, int(blurred), _p(out), out.size, C.byref(w), C.byref(h)), "pyramid_level")
        return out

    def candidates(self, slot, level):
        cap = 1 << 20
        out = np.zeros(cap, KP_DTYPE); n = C.c_int()
        _chk(self.L.corb_orb_fetch_candidates(self.h, slot, level, _p(out), cap, C.byref(n)), "fetch_candidates")
        return out[: n.value].copy()

    def profile(self, enable=True):
        _chk(self.L.corb_orb_profile(self.h, int(enable)), "corb_orb_profile")

    def profile_read(self):
        arr = (KernelTime * 64)(); n = C.c_int()
        _chk(self.L.corb_orb_profile_read(self.h, arr, 64, C.byref(n)), "corb_orb_profile_read")
        return {arr[i].name.decode(): (arr[i].total_ms, arr[i].launches) for i in range(n.value)}


_pinned_keep = []


def warmup(device=0):
    """corb_warmup: create the two workspace lanes (stream, events, pinned block) now instead of inside the first optimisation of the process."""
    _chk(load().corb_warmup(int(device)), "corb_warmup")


def release_scratch(device=0):
    """corb_release_scratch: the workspace arenas and the host-array staging of `device` back to the runtime; returns the bytes freed."""
    n = C.c_uint64(0)
    _chk(load().corb_release_scratch(int(device), C.byref(n)), "corb_release_scratch")
    return int(n.value)


def pinned_empty(shape, dtype):
    """numpy array in page-locked host memory (hipHostMalloc): copies to / from it are DMA transfers and asynchronous on the handle's stream.
    The allocation lives until the process exits."""
    L = load()                                              # (through the library: the HIP runtime IT is linked with must own the allocation)
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    ptr = C.c_void_p()
    L.corb_pinned_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
    _chk(L.corb_pinned_alloc(C.c_size_t(max(nbytes, 1)), C.byref(ptr)), "corb_pinned_alloc")
    buf = (C.c_uint8 * max(nbytes, 1)).from_address(ptr.value)
    _pinned_keep.append(buf)
    return np.frombuffer(buf, np.uint8, nbytes).view(dtype).reshape(shape)


class StereoFrontend:
    """Frame::Frame(stereo) hot path (corbslam_client/src/Frame.cc:61-117): left/right extraction +
    ComputeStereoMatches for a batch of frames."""

    def __init__(self, nfeatures=2000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7,
                 width=1241, height=376, max_frames=1, fx=718.856, bf=386.1448, device=0):
        self.L = load()
        cfg = StereoConfig(OrbConfig(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, width, height, 0, device),
                           max_frames, fx, bf)
        h = C.c_void_p()
        _chk(self.L.corb_stereo_create(C.byref(cfg), C.byref(h)), "corb_stereo_create")
        self.h = h
        self.max_frames = max_frames
        self.orb = ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, width, height, 2 * max_frames,
                                device, _handle=self.L.corb_stereo_orb(h))
        self._keep = []

    def close(self):
        if self.h:
            self.L.corb_stereo_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, frame, left, right):
        left = np.ascontiguousarray(left, np.uint8); right = np.ascontiguousarray(right, np.uint8)
        _chk(self.L.corb_stereo_upload(self.h, frame, _p(left), _p(right), left.shape[1]), "corb_stereo_upload")
        self._keep.append((left, right))

    def run(self, n_frames):
        _chk(self.L.corb_stereo_run(self.h, n_frames), "corb_stereo_run")

    def sync(self):
        _chk(self.L.corb_stereo_sync(self.h), "corb_stereo_sync")
        self._keep = []

    def upload_batch(self, first, packed):
        """packed: uint8 array [n_frames][2][height][width] (left, right), C-contiguous; pinned memory makes it a DMA."""
        assert packed.flags["C_CONTIGUOUS"] and packed.dtype == np.uint8
        _chk(self.L.corb_stereo_upload_batch(self.h, first, packed.shape[0], _p(packed)), "corb_stereo_upload_batch")

    def fetch_batch(self, first, n, out=None):
        """All results of frames first .. first+n-1 in one set of copies.  Returns dict of strided arrays:
        kp [2n][cap], desc [2n][cap][32], counts [2n], u_right / depth [n][cap], n_matched [n]  (image 2f = left, 2f+1 = right)."""
        cap = self.L.corb_orb_capacity(self.orb.h)
        if out is None:
            out = dict(kp=np.zeros((2 * n, cap), KP_DTYPE), desc=np.zeros((2 * n, cap, 32), np.uint8), counts=np.zeros(2 * n, np.int32),
                       u_right=np.zeros((n, cap), np.float32), depth=np.zeros((n, cap), np.float32), n_matched=np.zeros(n, np.int32))
        _chk(self.L.corb_orb_fetch_batch(self.orb.h, 2 * first, 2 * n, _p(out["kp"]), _p(out["desc"]), _p(out["counts"])), "corb_orb_fetch_batch")
        _chk(self.L.corb_stereo_fetch_matches_batch(self.h, first, n, _p(out["u_right"]), _p(out["depth"]), _p(out["n_matched"])), "corb_stereo_fetch_matches_batch")
        return out

    def frame_layout(self):
        lay = StereoFrameLayout()
        _chk(self.L.corb_stereo_frame_layout(self.h, C.byref(lay)), "corb_stereo_frame_layout")
        return lay

    def frames(self, packed, result=None, timing=None):
        """corb_stereo_frames: Frame::Frame(stereo) for the n frames of `packed` ([n][2][height][width] uint8) in ONE call -- one transfer each way, one
        synchronisation.  result: uint8 array of n * frame_layout().frame_bytes bytes (page-locked: pinned_empty); timing: a StereoFrameTiming to fill.
        Returns the result block (parse with unpack_frame)."""
        assert packed.flags["C_CONTIGUOUS"] and packed.dtype == np.uint8
        n = packed.shape[0]
        if result is None:
            result = np.zeros(n * self.frame_layout().frame_bytes, np.uint8)
        _chk(self.L.corb_stereo_frames(self.h, n, _p(packed), _p(result), C.byref(timing) if timing is not None else None), "corb_stereo_frames")
        return result

    def unpack_frame(self, result, f=0):
        """views into frame f's block of a corb_stereo_frames result"""
        lay = self.frame_layout()
        b = result[f * lay.frame_bytes: (f + 1) * lay.frame_bytes]
        nl, nr, nm, status = (int(x) for x in b[:16].view(np.int32))
        kp = lambda off, n: b[off: off + n * KP_DTYPE.itemsize].view(KP_DTYPE)
        return dict(kl=kp(lay.off_kp_left, nl), kr=kp(lay.off_kp_right, nr), dl=b[lay.off_desc_left: lay.off_desc_left + 32 * nl].reshape(nl, 32),
                    dr=b[lay.off_desc_right: lay.off_desc_right + 32 * nr].reshape(nr, 32), u_right=b[lay.off_u_right: lay.off_u_right + 4 * nl].view(np.float32),
                    depth=b[lay.off_depth: lay.off_depth + 4 * nl].view(np.float32), n_matched=nm, status=status)

    def fetch(self, frame):
        kl, dl = self.orb.fetch(2 * frame)
        kr, dr = self.orb.fetch(2 * frame + 1)
        ur = np.zeros(self.orb.cap, np.float32); dp = np.zeros(self.orb.cap, np.float32)
        n, nm = C.c_int(), C.c_int()
        _chk(self.L.corb_stereo_fetch_matches(self.h, frame, _p(ur), _p(dp), self.orb.cap, C.byref(n), C.byref(nm)), "fetch_matches")
        return dict(kl=kl, dl=dl, kr=kr, dr=dr, u_right=ur[: n.value].copy(), depth=dp[: n.value].copy(), n_matched=nm.value)


def _fv(node_id, offset, idx, keep):
    node_id = np.ascontiguousarray(node_id, np.uint32); offset = np.ascontiguousarray(offset, np.int32)
    idx = np.ascontiguousarray(idx, np.uint32)
    keep += [node_id, offset, idx]
    return _FeatVec(len(node_id), _p(node_id), _p(offset), _p(idx))


class ORBmatcher:
    """Mirror of ORB_SLAM2::ORBmatcher (corbslam_client/include/ORBmatcher.h:41-107), flat-array form."""
    TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30

    def __init__(self, nnratio=0.6, checkOri=True, device=0):
        self.nnratio, self.checkOri, self.device = float(nnratio), bool(checkOri), device
        self.L = load()

    @staticmethod
    def DescriptorDistance(a, b, device=0):
        a = np.ascontiguousarray(a, np.uint8).reshape(-1, 32); b = np.ascontiguousarray(b, np.uint8).reshape(-1, 32)
        out = np.zeros(len(a), np.int32)
        _chk(load().corb_descriptor_distance(_p(a), _p(b), len(a), _p(out), device), "corb_descriptor_distance")
        return out

    def _bow(self, variant, desc1, angle1, valid1, fv1, desc2, angle2, valid2, fv2):
        keep = []
        d1 = np.ascontiguousarray(desc1, np.uint8); d2 = np.ascontiguousarray(desc2, np.uint8)
        a1 = np.ascontiguousarray(angle1, np.float32); a2 = np.ascontiguousarray(angle2, np.float32)
        v1 = np.ascontiguousarray(valid1, np.uint8); v2 = np.ascontiguousarray(valid2, np.uint8)
        A = _BowSide(_p(d1), _p(a1), _p(v1), len(d1), _fv(*fv1, keep))
        B = _BowSide(_p(d2), _p(a2), _p(v2), len(d2), _fv(*fv2, keep))
        nslots = len(d2) if variant == 0 else len(d1)
        match = np.zeros(max(nslots, 1), np.int32); n = C.c_int()
        _chk(self.L.corb_search_by_bow(variant, C.byref(A), C.byref(B), self.nnratio, int(self.checkOri), _p(match),
                                       C.byref(n), self.device), "corb_search_by_bow")
        return match[:nslots], n.value

    def SearchByBoW(self, kf, frame):
        """SearchByBoW(KeyFrame*,Frame&) / SearchByBoWInServer: kf, frame = dict(desc, angle, valid, fv)."""
        return self._bow(0, kf["desc"], kf["angle"], kf["valid"], kf["fv"], frame["desc"], frame["angle"],
                         frame.get("valid", np.ones(len(frame["desc"]), np.uint8)), frame["fv"])

    SearchByBoWInServer = SearchByBoW

    def SearchByBoW_KFKF(self, kf1, kf2):
        """SearchByBoW(KeyFrame*,KeyFrame*)"""
        return self._bow(1, kf1["desc"], kf1["angle"], kf1["valid"], kf1["fv"], kf2["desc"], kf2["angle"], kf2["valid"], kf2["fv"])

    @staticmethod
    def _frame_view(fr, keep):
        k = np.ascontiguousarray(fr["keys_un"], KP_DTYPE); ur = np.ascontiguousarray(fr["u_right"], np.float32)
        d = np.ascontiguousarray(fr["desc"], np.uint8); cl = np.ascontiguousarray(fr["claimed"], np.uint8); sc = np.ascontiguousarray(fr["scale"], np.float32)
        keep += [k, ur, d, cl, sc]
        return _FrameView(_p(k), _p(ur), _p(d), len(k), _p(cl), fr["min_x"], fr["min_y"], fr["max_x"], fr["max_y"], _p(sc), len(sc))

    def SearchByProjection(self, frame, mps, mp_desc, th):
        """SearchByProjection(Frame&, const vector<MapPoint*>&, th): mps = TRACKED_DTYPE records (isInFrustum outputs)."""
        keep = []; fv = self._frame_view(frame, keep)
        mps = np.ascontiguousarray(mps, TRACKED_DTYPE); mp_desc = np.ascontiguousarray(mp_desc, np.uint8)
        match = np.zeros(max(len(frame["keys_un"]), 1), np.int32); n = C.c_int()
        _chk(self.L.corb_search_by_projection_map(C.byref(fv), _p(mps), _p(mp_desc), len(mps), float(th), self.nnratio, _p(match), C.byref(n), self.device),
             "corb_search_by_projection_map")
        return match[: len(frame["keys_un"])].copy(), n.value

    def SearchByProjection_Frame(self, cur, Tcw, Tlw, fx, fy, cx, cy, bf, mb, last, last_desc, th, bMono):
        """SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono)"""
        keep = []; fv = self._frame_view(cur, keep)
        Tcw = np.ascontiguousarray(Tcw, np.float32).reshape(16); Tlw = np.ascontiguousarray(Tlw, np.float32).reshape(16)
        last = np.ascontiguousarray(last, LAST_DTYPE); last_desc = np.ascontiguousarray(last_desc, np.uint8)
        match = np.zeros(max(len(cur["keys_un"]), 1), np.int32); n = C.c_int()
        _chk(self.L.corb_search_by_projection_frame(C.byref(fv), _p(Tcw), _p(Tlw), fx, fy, cx, cy, bf, mb, _p(last), _p(last_desc), len(last),
                                                    float(th), int(bMono), int(self.checkOri), _p(match), C.byref(n), self.device),
             "corb_search_by_projection_frame")
        return match[: len(cur["keys_un"])].copy(), n.value

    @staticmethod
    def _kf_view(kf, keep):
        k = np.ascontiguousarray(kf["keys_un"], KP_DTYPE); ur = np.ascontiguousarray(kf["u_right"], np.float32)
        d = np.ascontiguousarray(kf["desc"], np.uint8); sc = np.ascontiguousarray(kf["scale"], np.float32); s2 = np.ascontiguousarray(kf["inv_level_sigma2"], np.float32)
        keep += [k, ur, d, sc, s2]
        return _KeyFrameView(_p(k), _p(ur), _p(d), len(k), kf["min_x"], kf["min_y"], kf["max_x"], kf["max_y"], _p(sc), _p(s2), len(sc),
                             kf["log_scale_factor"], kf["fx"], kf["fy"], kf["cx"], kf["cy"], kf["bf"])

    def SearchByProjection_Reloc(self, cur, claimed, Tcw, pts, desc, th, ORBdist):
        """SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, th, ORBdist)"""
        keep = []; kv = self._kf_view(cur, keep)
        claimed = np.ascontiguousarray(claimed, np.uint8); Tcw = np.ascontiguousarray(Tcw, np.float32).reshape(16)
        pts = np.ascontiguousarray(pts, MP_DTYPE); desc = np.ascontiguousarray(desc, np.uint8)
        match = np.zeros(max(len(cur["keys_un"]), 1), np.int32); n = C.c_int()
        _chk(self.L.corb_search_by_projection_reloc(C.byref(kv), _p(claimed), _p(Tcw), _p(pts), _p(desc), len(pts), float(th), int(ORBdist), int(self.checkOri),
                                                    _p(match), C.byref(n), self.device), "corb_search_by_projection_reloc")
        return match[: len(cur["keys_un"])].copy(), n.value

    def SearchForInitialization(self, f1, f2, prev_matched, window_size=100):
        """ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (ORBmatcher.cc:540-655).  Returns (vnMatches12, vbPrevMatched after the call, nmatches)."""
        keep = []
        def fv(fr):
            fr = dict(fr); fr.setdefault("claimed", np.zeros(len(fr["keys_un"]), np.uint8)); fr.setdefault("u_right", -np.ones(len(fr["keys_un"]), np.float32))
            return self._frame_view(fr, keep)
        v1, v2 = fv(f1), fv(f2)
        pm = np.array(prev_matched, np.float32, copy=True).reshape(-1, 2); assert len(pm) == len(f1["keys_un"])
        m = np.zeros(max(len(pm), 1), np.int32); n = C.c_int()
        _chk(self.L.corb_search_for_initialization(C.byref(v1), C.byref(v2), _p(pm), int(window_size), float(self.nnratio), int(self.checkOri), _p(m), C.byref(n), self.device),
             "corb_search_for_initialization")
        return m[: len(pm)].copy(), pm, n.value

    def SearchByProjection_Scw(self, kf, claimed, Scw, pts, desc, th):
        """SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, vector<MapPoint*> &vpMatched, int th) (ORBmatcher.cc:425-538).
        claimed[idx] = vpMatched[idx] is set on entry (None: nothing set).  Returns (match per keyframe feature: point index or -1, nmatches)."""
        keep = []; kv = self._kf_view(kf, keep)
        nk = len(kf["keys_un"])
        claimed = np.zeros(nk, np.uint8) if claimed is None else np.ascontiguousarray(claimed, np.uint8)
        Scw = np.ascontiguousarray(Scw, np.float32).reshape(16)
        pts = np.ascontiguousarray(pts, MP_DTYPE); desc = np.ascontiguousarray(desc, np.uint8)
        match = np.zeros(max(nk, 1), np.int32); n = C.c_int()
        _chk(self.L.corb_search_by_projection_scw(C.byref(kv), _p(claimed), _p(Scw), _p(pts), _p(desc), len(pts), float(th), _p(match), C.byref(n), self.device),
             "corb_search_by_projection_scw")
        return match[:nk].copy(), n.value

    def Fuse(self, kf, T, Ow, pts, desc, th, sim3=False):
        """Fuse(KeyFrame*, vpMapPoints, th) (sim3=False, T = Tcw, Ow = camera centre) / Fuse(KeyFrame*, Scw, vpPoints, th, vpReplacePoint) (sim3=True).
        Returns (best feature per point or -1, best distance, nFused)."""
        keep = []; kv = self._kf_view(kf, keep)
        T = np.ascontiguousarray(T, np.float32).reshape(16); Ow = np.ascontiguousarray(Ow if Ow is not None else np.zeros(3), np.float32)
        pts = np.ascontiguousarray(pts, MP_DTYPE); desc = np.ascontiguousarray(desc, np.uint8)
        bi = np.zeros(max(len(pts), 1), np.int32); bd = np.zeros(max(len(pts), 1), np.int32); n = C.c_int()
        _chk(self.L.corb_fuse(C.byref(kv), _p(T), _p(Ow), int(sim3), _p(pts), _p(desc), len(pts), float(th), _p(bi), _p(bd), C.byref(n), self.device), "corb_fuse")
        return bi[: len(pts)].copy(), bd[: len(pts)].copy(), n.value

    def SearchBySim3(self, kf1, kf2, T1w, T2w, pts1, desc1, pts2, desc2, s12, R12, t12, th):
        keep = []; k1 = self._kf_view(kf1, keep); k2 = self._kf_view(kf2, keep)
        T1w = np.ascontiguousarray(T1w, np.float32).reshape(16); T2w = np.ascontiguousarray(T2w, np.float32).reshape(16)
        pts1 = np.ascontiguousarray(pts1, MP_DTYPE); pts2 = np.ascontiguousarray(pts2, MP_DTYPE)
        desc1 = np.ascontiguousarray(desc1, np.uint8); desc2 = np.ascontiguousarray(desc2, np.uint8)
        R12 = np.ascontiguousarray(R12, np.float32).reshape(9); t12 = np.ascontiguousarray(t12, np.float32).reshape(3)
        m = np.zeros(max(len(pts1), 1), np.int32); n = C.c_int()
        _chk(self.L.corb_search_by_sim3(C.byref(k1), C.byref(k2), _p(T1w), _p(T2w), _p(pts1), _p(desc1), _p(pts2), _p(desc2), float(np.float32(s12)),
                                        _p(R12), _p(t12), float(th), _p(m), C.byref(n), self.device), "corb_search_by_sim3")
        return m[: len(pts1)].copy(), n.value

    def SearchForTriangulation(self, kf1, kf2, F12, ex, ey, scale2, sigma2_2, bOnlyStereo):
        keep = []
        def side(k):
            d = np.ascontiguousarray(k["desc"], np.uint8); kp = np.ascontiguousarray(k["kp"], KP_DTYPE)
            ur = np.ascontiguousarray(k["u_right"], np.float32); mp = np.ascontiguousarray(k["has_mp"], np.uint8)
            keep.extend([d, kp, ur, mp])
            return _TriSide(_p(d), _p(kp), _p(ur), _p(mp), len(d), _fv(*k["fv"], keep))
        A, B = side(kf1), side(kf2)
        F = np.ascontiguousarray(F12, np.float32).reshape(9)
        sc = np.ascontiguousarray(scale2, np.float32); sg = np.ascontiguousarray(sigma2_2, np.float32)
        pairs = np.zeros((max(len(kf1["desc"]), 1), 2), np.int32); n = C.c_int()
        _chk(self.L.corb_search_for_triangulation(C.byref(A), C.byref(B), _p(F), float(np.float32(ex)), float(np.float32(ey)),
                                                  _p(sc), _p(sg), len(sc), int(bOnlyStereo), int(self.checkOri), _p(pairs),
                                                  C.byref(n), self.device), "corb_search_for_triangulation")
        return pairs[: n.value].copy(), n.value


def spd_solve(A, b, device=0):
    """corb_spd_solve: x with A x = b for a symmetric positive definite A (hand-written blocked Cholesky, csrc/dense_chol.hip); returns (x, info)"""
    A = np.ascontiguousarray(A, np.float64); b = np.ascontiguousarray(b, np.float64); x = np.zeros(len(b), np.float64); info = C.c_int(0)
    _chk(load().corb_spd_solve(_p(A), len(b), _p(b), _p(x), C.byref(info), device), "corb_spd_solve")
    return x, info.value


def ComputeDistinctiveDescriptors(desc, offset, device=0):
    """MapPoint::ComputeDistinctiveDescriptors for a batch: returns the adopted row (relative) per map point."""
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32); offset = np.ascontiguousarray(offset, np.int32)
    best = np.zeros(max(len(offset) - 1, 1), np.int32)
    _chk(load().corb_distinctive_descriptors(_p(desc), _p(offset), len(offset) - 1, _p(best), device), "corb_distinctive_descriptors")
    return best[: len(offset) - 1].copy()


def RebaseMap(To2n, poses, points, device=0):
    """MapFusion::insertServerMapToGlobleMap arithmetic: returns (Tcw * To2n per keyframe, Rwc (p - tcw) per map point)."""
    T = np.ascontiguousarray(To2n, np.float32).reshape(16)
    P = np.array(poses, np.float32).reshape(-1, 16).copy(); X = np.array(points, np.float32).reshape(-1, 3).copy()
    _chk(load().corb_rebase_map(_p(T), _p(P), len(P), _p(X), len(X), device), "corb_rebase_map")
    return P.reshape(-1, 4, 4), X


class Optimizer:
    """Mirror of ORB_SLAM2::Optimizer::GlobalBundleAdjustemnt (sic, corbslam_client/include/Optimizer.h:45)."""

    @staticmethod
    def GlobalBundleAdjustemnt(poses, pose_fixed, points, point_fixed, edges, fx, fy, cx, cy, bf,
                               nIterations=5, bRobust=True, device=0, solver=0, pcg_tol=0.0, pcg_max_iter=0, pc_block=0, intr=None, devflat=False, pc_multilevel=0):
        poses = np.ascontiguousarray(poses, np.float32).reshape(-1, 16)
        points = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
        pose_fixed = np.ascontiguousarray(pose_fixed, np.uint8); point_fixed = np.ascontiguousarray(point_fixed, np.uint8)
        edges = np.ascontiguousarray(edges, EDGE_DTYPE)
        if intr is not None:
            intr = np.ascontiguousarray(intr, np.float32).reshape(len(poses), 5)   # per-keyframe fx, fy, cx, cy, bf (pKF->fx ... pKF->mbf)
        prob = _BAProblem(len(poses), len(points), len(edges), _p(poses), _p(pose_fixed), _p(points), _p(point_fixed),
                          _p(edges), fx, fy, cx, cy, bf, _p(intr) if intr is not None else None)
        oposes = np.zeros_like(poses); opoints = np.zeros_like(points)
        chi2 = np.zeros(nIterations + 1, np.float64); lam = np.zeros(max(nIterations, 1), np.float64)
        res = _BAResult(_p(oposes), _p(opoints), _p(chi2), _p(lam), 0, 0, 0, 0, 0, 0, 0, 0, 0)
        opt = BAOptions(solver, pcg_tol, pcg_max_iter, pc_block, pc_multilevel)
        if devflat:          # the graph flattening on the device (the path of corb_ba_solve_store)
            _chk(load().corb_ba_solve_devflat(C.byref(prob), nIterations, int(bRobust), C.byref(res), device, C.byref(opt)), "corb_ba_solve_devflat")
        else:
            _chk(load().corb_ba_solve_ex(C.byref(prob), nIterations, int(bRobust), None, C.byref(res), device, C.byref(opt)), "corb_ba_solve_ex")
        return dict(poses=oposes.reshape(-1, 4, 4), points=opoints, chi2=chi2[: res.iters_done + 1],
                    lam=lam[: res.iters_done], iters_done=res.iters_done, trials=res.trials_total,
                    solver=res.solver_used, pcg_iterations=res.pcg_iterations,
                    structure=dict(free_poses=res.free_poses, free_points=res.free_points, active_edges=res.active_edges, nnz_blocks=res.nnz_blocks,
                                   schur_pairs=res.schur_pairs, pc_block=res.pc_block, pc_levels=res.pc_levels),
                    certificate=dict(pcg_residual_max=res.pcg_residual_max, pcg_residual_last=res.pcg_residual_last, grad_inf=res.grad_inf, pcg_refined_trials=res.pcg_refined_trials),
                    ms=dict(total=res.ms_total, build=res.ms_build, schur=res.ms_schur, solve=res.ms_solve, update=res.ms_update))


    @staticmethod
    def _staged(stages, poses, pose_fixed, points, point_fixed, edges, fx, fy, cx, cy, bf, device=0, solver=0, intr=None, stop=None):
        poses = np.ascontiguousarray(poses, np.float32).reshape(-1, 16)
        points = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
        pose_fixed = np.ascontiguousarray(pose_fixed, np.uint8); point_fixed = np.ascontiguousarray(point_fixed, np.uint8)
        edges = np.ascontiguousarray(edges, EDGE_DTYPE)
        if intr is not None:
            intr = np.ascontiguousarray(intr, np.float32).reshape(len(poses), 5)   # per-keyframe fx, fy, cx, cy, bf (pKF->fx ... pKF->mbf)
        prob = _BAProblem(len(poses), len(points), len(edges), _p(poses), _p(pose_fixed), _p(points), _p(point_fixed),
                          _p(edges), fx, fy, cx, cy, bf, _p(intr) if intr is not None else None)
        oposes = np.zeros_like(poses); opoints = np.zeros_like(points)
        res = _BAResult(_p(oposes), _p(opoints), None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0)
        st = (BAStage * len(stages))(*[BAStage(*s) for s in stages])
        outl = np.zeros(max(len(edges), 1), np.uint8)
        opt = BAOptions(solver, 0.0, 0, 0)
        # pbStopFlag (tests): "before" = raised before the call; "after_first_stage" = the flag aliases the result's iteration counter, which turns
        # non-zero exactly when the first optimize() returns (a deterministic stand-in for LocalMapping::InterruptBA during the first round)
        one = C.c_int(1)
        sp = None if stop is None else (C.cast(C.byref(one), C.c_void_p) if stop == "before" else C.cast(C.byref(res, _BAResult.iters_done.offset), C.c_void_p))
        _chk(load().corb_ba_solve_staged(C.byref(prob), st, len(stages), sp, C.byref(res), _p(outl), device, C.byref(opt)), "corb_ba_solve_staged")
        return dict(poses=oposes.reshape(-1, 4, 4), points=opoints, outlier=outl[: len(edges)].copy(), iters_done=res.iters_done,
                    trials=res.trials_total, ms_total=res.ms_total, device_route=bool(res.reserved0))

    @staticmethod
    def LocalBundleAdjustment(*args, **kw):
        """Optimizer::LocalBundleAdjustment: local keyframes free, fixed keyframes fixed; returns poses, points and the
        observations to erase (outlier[i] = 1 -> pKFi->EraseMapPointMatch / pMP->EraseObservation)."""
        return Optimizer._staged(LOCAL_BA_STAGES, *args, **kw)

    @staticmethod
    def PoseOptimization(Tcw, points, obs, inv_sigma2, fx, fy, cx, cy, bf, device=0, solver=0):
        """Optimizer::PoseOptimization(Frame*): one free pose, fixed map points; obs = (u, v, uRight) per matched point.
        Returns (Tcw, mvbOutlier, nInitialCorrespondences - nBad).  solver 0/3: fused single-workgroup kernel, 1: general path."""
        n = len(points)
        edges = np.zeros(n, EDGE_DTYPE)
        edges["pose"] = 0; edges["point"] = np.arange(n); edges["u"] = obs[:, 0]; edges["v"] = obs[:, 1]; edges["ur"] = obs[:, 2]
        edges["inv_sigma2"] = inv_sigma2
        r = Optimizer._staged(POSE_OPT_STAGES, np.asarray(Tcw, np.float32).reshape(1, 16), np.zeros(1, np.uint8), points, np.ones(n, np.uint8),
                              edges, fx, fy, cx, cy, bf, device=device, solver=solver)
        return r["poses"][0], r["outlier"].astype(bool), int(n - r["outlier"].sum())

    @staticmethod
    def OptimizeEssentialGraph(g, iterations=20, bFixScale=False, device=0):
        """Optimizer::OptimizeEssentialGraph on a flattened graph (dict like synth.essential_graph)."""
        S = np.ascontiguousarray(g["S"], np.float64).copy()
        fixed = np.ascontiguousarray(g["fixed"], np.uint8); vi = np.ascontiguousarray(g["vi"], np.int32); vj = np.ascontiguousarray(g["vj"], np.int32)
        meas = np.ascontiguousarray(g["meas"], np.float64)
        Tiw = np.zeros((len(S), 16), np.float32); pts = np.ascontiguousarray(g["points"], np.float32).copy(); ref = np.ascontiguousarray(g["ref"], np.int32)
        chi2 = np.zeros(iterations + 1, np.float64); it = C.c_int32()
        _chk(load().corb_optimize_essential_graph(len(S), _p(S), _p(fixed), len(vi), _p(vi), _p(vj), _p(meas), iterations, int(bFixScale), _p(Tiw), len(pts),
                                                  _p(ref), _p(pts), _p(chi2), C.byref(it), device), "corb_optimize_essential_graph")
        return dict(S=S, chi2=chi2[: it.value + 1], iters_done=it.value, Tiw=Tiw.reshape(-1, 4, 4), points=pts)

    @staticmethod
    def OptimizeSim3(problems, th2=10.0, bFixScale=False, device=0):
        """Optimizer::OptimizeSim3 for a list of loop-closure candidates (dicts like synth.sim3_problem).  Returns a list of
        dict(R, t, s, removed, n_in, iters_done)."""
        L = load(); n = len(problems)
        arr = (_Sim3Problem * n)(); keep = []; rem = []
        R = np.zeros((n, 9), np.float64); t = np.zeros((n, 3), np.float64); s = np.zeros(n, np.float64)
        for f, q in enumerate(problems):
            a = [np.ascontiguousarray(q[k], np.float32) for k in ("p1c", "p2c", "obs1", "obs2", "inv_sigma2_1", "inv_sigma2_2")]
            keep.append(a); rem.append(np.zeros(max(len(a[0]), 1), np.uint8))
            K = [float(np.float32(q[k])) for k in ("fx1", "fy1", "cx1", "cy1", "fx2", "fy2", "cx2", "cy2")]
            arr[f] = _Sim3Problem(len(a[0]), *[_p(x) for x in a], *K)
            R[f] = np.asarray(q["R12"], np.float64).reshape(9); t[f] = np.asarray(q["t12"], np.float64).reshape(3); s[f] = q["s12"]
        rptr = (C.c_void_p * n)(*[r.ctypes.data for r in rem])
        nin = np.zeros(n, np.int32); its = np.zeros(n, np.int32)
        _chk(L.corb_optimize_sim3(arr, n, _p(R), _p(t), _p(s), float(np.float32(th2)), int(bFixScale), C.cast(rptr, C.c_void_p), _p(nin), _p(its), device), "corb_optimize_sim3")
        return [dict(R=R[f].reshape(3, 3).copy(), t=t[f].copy(), s=float(s[f]), removed=rem[f][: len(keep[f][0])].copy(), n_in=int(nin[f]), iters_done=int(its[f])) for f in range(n)]

    @staticmethod
    def PoseOptimizationBatch(frames, fx, fy, cx, cy, bf, device=0):
        """corb_pose_optimization_batch: frames = list of (Tcw, points, obs, inv_sigma2).  Returns a list of
        (Tcw, mvbOutlier, n_inliers) -- one workgroup per frame, one launch for the whole batch."""
        L = load()
        n = len(frames)
        arr = (_PoseOptFrame * n)()
        keep = []
        outl = []
        for f, (Tcw, points, obs, inv_sigma2) in enumerate(frames):
            T = np.ascontiguousarray(Tcw, np.float32).reshape(16)
            P = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
            u = np.ascontiguousarray(obs[:, 0], np.float32); v = np.ascontiguousarray(obs[:, 1], np.float32); ur = np.ascontiguousarray(obs[:, 2], np.float32)
            w = np.ascontiguousarray(inv_sigma2, np.float32)
            o = np.zeros(max(len(P), 1), np.uint8)
            keep.append((T, P, u, v, ur, w)); outl.append(o)
            arr[f] = _PoseOptFrame(_p(T), len(P), _p(P), _p(u), _p(v), _p(ur), _p(w), fx, fy, cx, cy, bf)
        Tout = np.zeros((n, 16), np.float32)
        ninl = np.zeros(n, np.int32)
        optr = (C.c_void_p * n)(*[o.ctypes.data for o in outl])
        _chk(L.corb_pose_optimization_batch(arr, n, _p(Tout), C.cast(optr, C.c_void_p), _p(ninl), device), "corb_pose_optimization_batch")
        return [(Tout[f].reshape(4, 4).copy(), outl[f][: len(keep[f][1])].astype(bool), int(ninl[f])) for f in range(n)]



class TrackCamera(C.Structure):
    """CorbTrackCamera: Frame intrinsics, image bounds and mvScaleFactors"""
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("bf", C.c_float), ("mb", C.c_float),
                ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float), ("max_y", C.c_float), ("nlevels", C.c_int32), ("scale", C.c_float * 16)]

    @classmethod
    def make(cls, fx, fy, cx, cy, bf, mb, min_x, max_x, min_y, max_y, scale):
        c = cls(); c.fx, c.fy, c.cx, c.cy, c.bf, c.mb = fx, fy, cx, cy, bf, mb
        c.min_x, c.max_x, c.min_y, c.max_y = min_x, max_x, min_y, max_y
        c.nlevels = len(scale)
        for i, v in enumerate(scale):
            c.scale[i] = float(v)
        return c


class KeyFrameStore:
    """Device-resident keyframe store (corb_kf_store_*): one fixed-size SoA record per keyframe in HBM -- what the reference serialises per KeyFrame for
    the client -> server push (corbslam_client/include/KeyFrame.h:59-87).  Slots are filled device-to-device from a StereoFrontend, matched without
    uploads (SearchByBoW / SearchForTriangulation on slots) and pushed to the server rank over RCCL (map_push)."""

    def __init__(self, capacity, max_features, device=0):
        self.h = C.c_void_p(); self.capacity = capacity; self.F = max_features; self.device = device
        _chk(load().corb_kf_store_create(device, capacity, max_features, C.byref(self.h)), "corb_kf_store_create")

    def close(self):
        if self.h:
            load().corb_kf_store_destroy(self.h); self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def record_bytes(self):
        return load().corb_kf_store_record_bytes(self.h)

    def put_from_stereo(self, slot, sf, frame, keyframe_id=0):
        _chk(load().corb_kf_store_put_from_stereo(self.h, slot, sf.h, frame, keyframe_id), "corb_kf_store_put_from_stereo")

    def put(self, slot, kp, desc, u_right=None, depth=None, keyframe_id=0):
        kp = np.ascontiguousarray(kp, KP_DTYPE); desc = np.ascontiguousarray(desc, np.uint8)
        ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32); dp = None if depth is None else np.ascontiguousarray(depth, np.float32)
        _chk(load().corb_kf_store_put_host(self.h, slot, _p(kp), _p(desc), _p(ur), _p(dp), len(kp), keyframe_id), "corb_kf_store_put_host")

    def put_frame(self, slot, kp, desc, u_right, depth, meta):
        """corb_kf_store_put_frame: features and header (KF_META_DTYPE record) in one upload"""
        kp = np.ascontiguousarray(kp, KP_DTYPE); desc = np.ascontiguousarray(desc, np.uint8); m = np.ascontiguousarray(meta, KF_META_DTYPE)
        ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32); dp = None if depth is None else np.ascontiguousarray(depth, np.float32)
        _chk(load().corb_kf_store_put_frame(self.h, slot, _p(kp), _p(desc), _p(ur), _p(dp), len(kp), _p(m)), "corb_kf_store_put_frame")

    def set_bow(self, slot, fv):
        node, off, idx = (np.ascontiguousarray(fv[0], np.uint32), np.ascontiguousarray(fv[1], np.int32), np.ascontiguousarray(fv[2], np.uint32))
        c = _FeatVec(len(node), _p(node), _p(off), _p(idx))
        _chk(load().corb_kf_store_set_bow(self.h, slot, C.byref(c)), "corb_kf_store_set_bow")

    def set_flags(self, slot, flags):
        f = None if flags is None else np.ascontiguousarray(flags, np.uint8)
        _chk(load().corb_kf_store_set_flags(self.h, slot, _p(f)), "corb_kf_store_set_flags")

    def get(self, slot):
        F = self.F
        kp = np.zeros(F, KP_DTYPE); desc = np.zeros((F, 32), np.uint8); ur = np.zeros(F, np.float32); dp = np.zeros(F, np.float32); fl = np.zeros(F, np.uint8)
        node = np.zeros(F, np.uint32); off = np.zeros(F + 1, np.int32); idx = np.zeros(F, np.uint32)
        n = C.c_int(0); kid = C.c_uint64(0); nn = C.c_int32(0)
        _chk(load().corb_kf_store_get(self.h, slot, _p(kp), _p(desc), _p(ur), _p(dp), _p(fl), F, C.byref(n), C.byref(kid), _p(node), _p(off), _p(idx), C.byref(nn)), "corb_kf_store_get")
        m = n.value; k = nn.value
        return dict(kp=kp[:m], desc=desc[:m], u_right=ur[:m], depth=dp[:m], flags=fl[:m], id=kid.value, fv=(node[:k].copy(), off[:k + 1].copy(), idx[:off[k]].copy() if k else idx[:0]))

    def put_batch(self, first, meta, feat_offset, kp, desc=None, u_right=None, depth=None, mp_id=None):
        """corb_kf_store_put_batch: n whole keyframes (meta[n] of KF_META_DTYPE, CSR feature arrays) in one upload and one kernel"""
        m = np.ascontiguousarray(meta, KF_META_DTYPE); off = np.ascontiguousarray(feat_offset, np.int32); k = np.ascontiguousarray(kp, KP_DTYPE)
        d = None if desc is None else np.ascontiguousarray(desc, np.uint8); u = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
        dp = None if depth is None else np.ascontiguousarray(depth, np.float32); ids = None if mp_id is None else np.ascontiguousarray(mp_id, np.uint64)
        _chk(load().corb_kf_store_put_batch(self.h, first, len(m), _p(m), _p(off), _p(k), _p(d), _p(u), _p(dp), _p(ids)), "corb_kf_store_put_batch")

    # ---- tracking-thread calls on records (corb_track_*): the current / last Frame are slots of this store, the map is a MapPointStore ----
    def TrackSearchLastFrame(self, cur_slot, last_slot, mp_store, Tcw, Tlw, cam, th, mono=False, nnratio=0.9, check_orientation=True, want_match=True):
        """ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (ORBmatcher.cc:1470-1614): writes CurrentFrame.mvpMapPoints into the record"""
        n = self._n_features(cur_slot)
        m = np.full(n, -1, np.int32) if want_match else None; cnt = C.c_int(0)
        a = np.ascontiguousarray(Tcw, np.float32).reshape(16); b = np.ascontiguousarray(Tlw, np.float32).reshape(16)
        _chk(load().corb_track_search_last_frame(self.h, int(cur_slot), int(last_slot), mp_store.h, _p(a), _p(b), C.byref(cam), C.c_float(th), int(bool(mono)),
                                                 C.c_float(nnratio), int(bool(check_orientation)), _p(m), C.byref(cnt)), "corb_track_search_last_frame")
        return m, cnt.value

    def TrackPoseOptimization(self, slot, mp_store, cam, Tcw, discard_outliers=False, want_outliers=True):
        """Optimizer::PoseOptimization(Frame*) (Optimizer.cc:272-485) on the record: returns (Tcw, mvbOutlier, inliers); pose and outlier flags stay in the record"""
        n = self._n_features(slot)
        a = np.ascontiguousarray(Tcw, np.float32).reshape(16); out = np.zeros(16, np.float32); fl = np.zeros(n, np.uint8) if want_outliers else None; inl = C.c_int32(0)
        _chk(load().corb_track_pose_optimization(self.h, int(slot), mp_store.h, C.byref(cam), _p(a), _p(out), int(bool(discard_outliers)), _p(fl), C.byref(inl)), "corb_track_pose_optimization")
        return out.reshape(4, 4), (fl.astype(bool) if want_outliers else None), inl.value

    def TrackSearchLocalPoints(self, slot, mp_store, local_ids, cam, Tcw, log_scale_factor, th=1.0, nnratio=0.8, want_match=True, want_tracked=False):
        """Tracking::SearchLocalPoints (Tracking.cc:1168-1216) on the record: isInFrustum on the device, SearchByProjection(Frame&, vpMapPoints, th); returns
        (match into local_ids per feature, matches, points in view[, TRACKED array])"""
        n = self._n_features(slot)
        ids = np.ascontiguousarray(local_ids, np.uint64)
        m = np.full(n, -1, np.int32) if want_match else None; tr = np.zeros(len(ids), TRACKED_DTYPE) if want_tracked else None
        cnt = C.c_int(0); inv = C.c_int(0)
        a = np.ascontiguousarray(Tcw, np.float32).reshape(16)
        _chk(load().corb_track_search_local_points(self.h, int(slot), mp_store.h, _p(ids), len(ids), C.byref(cam), _p(a), C.c_float(log_scale_factor), C.c_float(th),
                                                   C.c_float(nnratio), _p(m), _p(tr), C.byref(cnt), C.byref(inv)), "corb_track_search_local_points")
        return (m, cnt.value, inv.value, tr) if want_tracked else (m, cnt.value, inv.value)

    def SearchByProjectionScw(self, slot, mp_store, mp_slots, cam, Scw, log_scale_factor, matched_ids, th=10):
        """ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) (ORBmatcher.cc:425-538) on records (corb_search_by_projection_scw_store): pKF = this store's `slot`,
        vpPoints = mp_slots of mp_store, vpMatched = matched_ids (uint64 per feature, NO_MAP_POINT = NULL).  Returns (vpMatched after the call as ids, match into mp_slots or -1, nmatches)."""
        n = self._n_features(slot)
        ms = np.ascontiguousarray(mp_slots, np.int32); S = np.ascontiguousarray(Scw, np.float32).reshape(16)
        ids = np.array(matched_ids, np.uint64, copy=True); assert len(ids) == n
        m = np.full(max(n, 1), -1, np.int32); cnt = C.c_int(0)
        L = load(); L.corb_search_by_projection_scw_store.restype = C.c_int
        L.corb_search_by_projection_scw_store.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        _chk(L.corb_search_by_projection_scw_store(self.h, int(slot), mp_store.h, _p(ms), len(ms), C.byref(cam), _p(S), C.c_float(log_scale_factor), C.c_float(th),
                                                   _p(ids), _p(m), C.byref(cnt)), "corb_search_by_projection_scw_store")
        return ids, m[:n], cnt.value

    def Fuse(self, slot, mp_store, mp_slots, cam, Tcw, log_scale_factor, th=3.0, apply=False):
        """ORBmatcher::Fuse(pKF, vpMapPoints, th) on records (corb_fuse_store): pKF = this store's `slot`, vpMapPoints = mp_slots of mp_store.
        Returns (best_idx, best_dist, n_fused, action): action 1 = entered a feature without MapPoint (apply: records updated), 2 = Replace pending."""
        ms = np.ascontiguousarray(mp_slots, np.int32); T = np.ascontiguousarray(Tcw, np.float32).reshape(16)
        bi = np.full(max(len(ms), 1), -1, np.int32); bd = np.full(max(len(ms), 1), 256, np.int32); act = np.zeros(max(len(ms), 1), np.uint8); n = C.c_int(0)
        _chk(load().corb_fuse_store(self.h, int(slot), mp_store.h, _p(ms), len(ms), C.byref(cam), _p(T), C.c_float(log_scale_factor), C.c_float(th), int(bool(apply)),
                                    _p(bi), _p(bd), _p(act), C.byref(n)), "corb_fuse_store")
        return bi[: len(ms)], bd[: len(ms)], n.value, act[: len(ms)]

    def TrackSearchReloc(self, cur_slot, kf_store, kf_slot, mp_store, cam, Tcw, log_scale_factor, th=10.0, orb_dist=100, check_orientation=True):
        """ORBmatcher::SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) on records (corb_track_search_reloc): the frame = this store's cur_slot,
        pKF = kf_slot of kf_store, sAlreadyFound = the MapPoints the frame holds.  Returns (match per frame feature = pKF feature or -1, matches); the ids go into the record."""
        n = self._n_features(cur_slot)
        m = np.full(n, -1, np.int32); cnt = C.c_int(0); a = np.ascontiguousarray(Tcw, np.float32).reshape(16)
        L = load(); L.corb_track_search_reloc.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _chk(L.corb_track_search_reloc(self.h, int(cur_slot), kf_store.h, int(kf_slot), mp_store.h, C.byref(cam), _p(a), C.c_float(log_scale_factor), C.c_float(th),
                                       int(orb_dist), int(bool(check_orientation)), _p(m), C.byref(cnt)), "corb_track_search_reloc")
        return m, cnt.value

    def SearchBySim3(self, slot1, slot2, mp_store, cam, log_scale_factor, T1w, T2w, s12, R12, t12, th=7.5, matched12_ids=None):
        """ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) on records (corb_search_by_sim3_store).  Returns (match12 = KF2 feature per KF1 feature or -1,
        the MapPoint ids those features hold, matches found)."""
        n1 = self._n_features(slot1)
        m = np.full(n1, -1, np.int32); ids = np.full(n1, NO_MAP_POINT, np.uint64); cnt = C.c_int(0)
        a = np.ascontiguousarray(T1w, np.float32).reshape(16); b = np.ascontiguousarray(T2w, np.float32).reshape(16)
        R = np.ascontiguousarray(R12, np.float32).reshape(9); t = np.ascontiguousarray(t12, np.float32).reshape(3)
        mi = None if matched12_ids is None else np.ascontiguousarray(matched12_ids, np.uint64)
        L = load(); L.corb_search_by_sim3_store.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                                            C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        _chk(L.corb_search_by_sim3_store(self.h, int(slot1), int(slot2), mp_store.h, C.byref(cam), C.c_float(log_scale_factor), _p(a), _p(b), _p(mi) if mi is not None else None,
                                         C.c_float(s12), _p(R), _p(t), C.c_float(th), _p(m), _p(ids), C.byref(cnt)), "corb_search_by_sim3_store")
        return m, ids, cnt.value

    def _n_features(self, slot):
        n = load().corb_kf_store_count(self.h, int(slot))
        if n < 0:
            raise RuntimeError("slot %d holds no frame with a host-known feature count" % slot)
        return n

    def _n_any(self, slot):
        """feature count of a slot: the host's mirror, or (a slot filled from a device-side count) the record's"""
        n = load().corb_kf_store_count(self.h, int(slot))
        return n if n >= 0 else len(self.get(slot)["kp"])

    def set_meta(self, slot, **kw):
        """pose, intrinsics, ids, flags of the keyframe (KeyFrame.h:65-79); unspecified fields keep the record's values"""
        m = self.get_meta(slot)
        for k, v in kw.items():
            m[k] = np.asarray(v).reshape(m[k].shape) if m[k].shape else v
        _chk(load().corb_kf_store_set_meta(self.h, slot, _p(m)), "corb_kf_store_set_meta")

    def set_meta_raw(self, slot, meta):
        """corb_kf_store_set_meta with a complete KF_META_DTYPE record (no read-modify-write)"""
        m = np.ascontiguousarray(meta, KF_META_DTYPE)
        _chk(load().corb_kf_store_set_meta(self.h, slot, _p(m)), "corb_kf_store_set_meta")

    def get_meta(self, slot):
        m = np.zeros((), KF_META_DTYPE)
        _chk(load().corb_kf_store_get_meta(self.h, slot, _p(m)), "corb_kf_store_get_meta")
        return m

    def set_map_points(self, slot, mp_id):
        a = np.ascontiguousarray(mp_id, np.uint64)
        _chk(load().corb_kf_store_set_map_points(self.h, slot, _p(a)), "corb_kf_store_set_map_points")

    def get_map_points(self, slot):
        a = np.zeros(self.F, np.uint64)
        _chk(load().corb_kf_store_get_map_points(self.h, slot, _p(a), self.F), "corb_kf_store_get_map_points")
        return a[: self._n_any(slot)].copy()

    def SearchByBoW(self, slot_a, other, slot_b, nnratio=0.6, checkOri=True, variant=0):
        na = self._n_any(slot_a); nb = other._n_any(slot_b)
        out = np.full(max(nb if variant == 0 else na, 1), -1, np.int32); n = C.c_int(0)
        _chk(load().corb_search_by_bow_slots(variant, self.h, slot_a, other.h, slot_b, nnratio, int(checkOri), _p(out), C.byref(n)), "corb_search_by_bow_slots")
        return out[: (nb if variant == 0 else na)], n.value

    def SearchForTriangulation(self, slot_a, other, slot_b, F12, ex, ey, scale2, sigma2_2, bOnlyStereo, checkOri=True):
        na = self._n_any(slot_a)
        F12 = np.ascontiguousarray(F12, np.float32); sc = np.ascontiguousarray(scale2, np.float32); sg = np.ascontiguousarray(sigma2_2, np.float32)
        pairs = np.zeros((max(na, 1), 2), np.int32); n = C.c_int(0)
        _chk(load().corb_search_for_triangulation_slots(self.h, slot_a, other.h, slot_b, _p(F12), ex, ey, _p(sc), _p(sg), len(sc), int(bOnlyStereo), int(checkOri), _p(pairs), C.byref(n)),
             "corb_search_for_triangulation_slots")
        return pairs[: n.value].copy(), n.value


class MapPointStore:
    """Device-resident map-point store (corb_mp_store_*): one fixed-size record per MapPoint -- header (MapPoint.h:52-72) + observation list."""

    def __init__(self, capacity, max_obs=32, device=0):
        self.h = C.c_void_p(); self.capacity = capacity; self.O = max_obs; self.device = device
        _chk(load().corb_mp_store_create(device, capacity, max_obs, C.byref(self.h)), "corb_mp_store_create")

    def close(self):
        if self.h:
            load().corb_mp_store_destroy(self.h); self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def record_bytes(self):
        return load().corb_mp_store_record_bytes(self.h)

    def put(self, first, records, obs_offset, obs_kf_id, obs_idx):
        rec = np.ascontiguousarray(records, MP_RECORD_DTYPE); off = np.ascontiguousarray(obs_offset, np.int32)
        okf = np.ascontiguousarray(obs_kf_id, np.uint64); oi = np.ascontiguousarray(obs_idx, np.uint32)
        _chk(load().corb_mp_store_put_host(self.h, first, len(rec), _p(rec), _p(off), _p(okf), _p(oi)), "corb_mp_store_put_host")

    def get(self, first, n):
        rec = np.zeros(n, MP_RECORD_DTYPE); okf = np.zeros((n, self.O), np.uint64); oi = np.zeros((n, self.O), np.uint32)
        _chk(load().corb_mp_store_get(self.h, first, n, _p(rec), _p(okf), _p(oi)), "corb_mp_store_get")
        return rec, okf, oi

    def set_counters(self, first, visible, found, replaced_by=None):
        """mnVisible / mnFound / mpReplaced (0 = none, else id + 1) of the records first .. (the header's spare 16 bytes)"""
        n = len(visible); a = np.zeros(n, MP_COUNTERS_DTYPE); a["n_visible"] = visible; a["n_found"] = found
        if replaced_by is not None: a["replaced_by"] = replaced_by
        L = load(); L.corb_mp_store_set_counters.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _chk(L.corb_mp_store_set_counters(self.h, int(first), n, _p(a)), "corb_mp_store_set_counters")

    def get_counters(self, first, n):
        a = np.zeros(n, MP_COUNTERS_DTYPE)
        L = load(); L.corb_mp_store_get_counters.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _chk(L.corb_mp_store_get_counters(self.h, int(first), int(n), _p(a)), "corb_mp_store_get_counters")
        return a

    def set_scratch(self, first, scratch):
        """CorbMapPointScratch of the records first ..: the tracking / local-mapping / loop-closing fields of MapPoint's serialised state the header does not carry"""
        a = np.ascontiguousarray(scratch, MP_SCRATCH_DTYPE)
        L = load(); L.corb_mp_store_set_scratch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]; L.corb_mp_store_set_scratch.restype = C.c_int
        _chk(L.corb_mp_store_set_scratch(self.h, int(first), len(a), _p(a)), "corb_mp_store_set_scratch")

    def get_scratch(self, first, n):
        a = np.zeros(n, MP_SCRATCH_DTYPE)
        L = load(); L.corb_mp_store_get_scratch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]; L.corb_mp_store_get_scratch.restype = C.c_int
        _chk(L.corb_mp_store_get_scratch(self.h, int(first), int(n), _p(a)), "corb_mp_store_get_scratch")
        return a

    def Replace(self, slot_this, slot_into, kf_store, kf_first, kf_n):
        """MapPoint::Replace(pMP) on records (corb_mp_store_replace): returns 0 (done) or 1 (the same point)"""
        st = C.c_int(0)
        L = load(); L.corb_mp_store_replace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _chk(L.corb_mp_store_replace(self.h, int(slot_this), int(slot_into), kf_store.h, int(kf_first), int(kf_n), C.byref(st)), "corb_mp_store_replace")
        return st.value

    def build_index(self, first, n):
        """corb_mp_store_build_index: the mnId -> slot table the tracking calls on records look map points up in"""
        _chk(load().corb_mp_store_build_index(self.h, int(first), int(n)), "corb_mp_store_build_index")


def map_push_plan(headers, root, kf_capacity, mp_capacity, kf_dst_first, mp_dst_first=None):
    """corb_map_push_plan: (verdict code, failing rank) of a push from what the ranks contributed to the header all-gather.  Pure host arithmetic."""
    h = np.ascontiguousarray(headers, PUSH_HEADER_DTYPE)
    kd = None if kf_dst_first is None else np.ascontiguousarray(kf_dst_first, np.int32); md = None if mp_dst_first is None else np.ascontiguousarray(mp_dst_first, np.int32)
    who = C.c_int(-1)
    rc = load().corb_map_push_plan(len(h), root, _p(h), kf_capacity, mp_capacity, _p(kd), _p(md), C.byref(who))
    return rc, who.value


PUSH_MSG_DTYPE = np.dtype([("peer", "<i4"), ("kind", "<i4"), ("first_record", "<i4"), ("n_records", "<i4"), ("bytes", "<i8")])


def map_push_messages(headers, rank, root, kf_dst_first=None, mp_dst_first=None):
    """corb_map_push_messages: (sends, recvs) rank `rank` posts for a push with these gathered headers.  Pure host arithmetic."""
    h = np.ascontiguousarray(headers, PUSH_HEADER_DTYPE); W = len(h)
    kd = None if kf_dst_first is None else np.ascontiguousarray(kf_dst_first, np.int32); md = None if mp_dst_first is None else np.ascontiguousarray(mp_dst_first, np.int32)
    sends = np.zeros(2, PUSH_MSG_DTYPE); recvs = np.zeros(2 * W, PUSH_MSG_DTYPE); ns = C.c_int(0); nr = C.c_int(0)
    L = load(); L.corb_map_push_messages.restype = C.c_int
    _chk(L.corb_map_push_messages(W, int(rank), int(root), _p(h), _p(kd), _p(md), _p(sends), C.byref(ns), _p(recvs), C.byref(nr)), "corb_map_push_messages")
    return sends[: ns.value].copy(), recvs[: nr.value].copy()


class _RcclFns(C.Structure):
    _fields_ = [("group_start", C.CFUNCTYPE(C.c_int)), ("group_end", C.CFUNCTYPE(C.c_int)),
                ("send", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p)),
                ("recv", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p))]


def rccl_exchange_with_fake(sends, recvs, fail_at=None):
    """corb_comm_test_rccl_exchange: the RCCL branch's posting loop on a recording fake of ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd (no GPU, no librccl).
    Returns (return code, call log); fail_at = index of the send / recv call (0-based, sends first) that returns an error."""
    log = []; n = [0]
    def gs(): log.append(("group_start",)); return 0
    def ge(): log.append(("group_end",)); return 0
    def snd(buf, count, dt, peer, comm, stream):
        log.append(("send", int(peer), int(count), int(dt))); n[0] += 1; return 1 if fail_at is not None and n[0] - 1 == fail_at else 0
    def rcv(buf, count, dt, peer, comm, stream):
        log.append(("recv", int(peer), int(count), int(dt))); n[0] += 1; return 1 if fail_at is not None and n[0] - 1 == fail_at else 0
    f = _RcclFns(); keep = (_RcclFns._fields_[0][1](gs), _RcclFns._fields_[1][1](ge), _RcclFns._fields_[2][1](snd), _RcclFns._fields_[3][1](rcv))
    f.group_start, f.group_end, f.send, f.recv = keep
    sm = np.ascontiguousarray(sends, PUSH_MSG_DTYPE); rm = np.ascontiguousarray(recvs, PUSH_MSG_DTYPE)
    L = load(); L.corb_comm_test_rccl_exchange.restype = C.c_int
    rc = L.corb_comm_test_rccl_exchange(C.byref(f), _p(sm) if len(sm) else None, len(sm), _p(rm) if len(rm) else None, len(rm))
    return rc, log


def RebaseMapStore(To2n, kf, kf_slots, mp=None, mp_slots=()):
    """MapFusion::insertServerMapToGlobleMap on store records (in place on the device)"""
    T = np.ascontiguousarray(To2n, np.float32).reshape(16); ks = np.ascontiguousarray(kf_slots, np.int32); ms = np.ascontiguousarray(mp_slots, np.int32)
    _chk(load().corb_rebase_map_store(_p(T), kf.h if kf is not None else None, _p(ks) if len(ks) else None, len(ks), mp.h if mp is not None else None, _p(ms) if len(ms) else None, len(ms)),
         "corb_rebase_map_store")


def GlobalBundleAdjustemntStore(kf, kf_slots, mp, mp_slots, nIterations=10, bRobust=False, nLoopKF=0, solver=0, pcg_tol=0.0, pcg_max_iter=0, pc_block=0, fetch=True, pc_multilevel=0, scale_factor=0.0):
    """Optimizer::GlobalBundleAdjustemnt on store records (corb_ba_solve_store): graph built on the device, estimates written back into the records
    (scale_factor > 0 with nLoopKF == 0: UpdateNormalAndDepth on the records as well)"""
    ks = np.ascontiguousarray(kf_slots, np.int32); ms = np.ascontiguousarray(mp_slots, np.int32)
    oposes = np.zeros((len(ks), 16), np.float32) if fetch else None; opoints = np.zeros((len(ms), 3), np.float32) if fetch else None
    chi2 = np.zeros(nIterations + 1, np.float64); lam = np.zeros(max(nIterations, 1), np.float64)
    res = _BAResult(_p(oposes), _p(opoints), _p(chi2), _p(lam), 0, 0, 0, 0, 0, 0, 0, 0, 0)
    opt = BAOptions(solver, pcg_tol, pcg_max_iter, pc_block, pc_multilevel, scale_factor)
    _chk(load().corb_ba_solve_store(kf.h, _p(ks), len(ks), mp.h, _p(ms), len(ms), nIterations, int(bRobust), None, nLoopKF, C.byref(res), C.byref(opt)), "corb_ba_solve_store")
    return dict(poses=None if oposes is None else oposes.reshape(-1, 4, 4), points=opoints, chi2=chi2[: res.iters_done + 1], lam=lam[: res.iters_done], iters_done=res.iters_done,
                trials=res.trials_total, solver=res.solver_used, pcg_iterations=res.pcg_iterations,
                structure=dict(free_poses=res.free_poses, free_points=res.free_points, active_edges=res.active_edges, nnz_blocks=res.nnz_blocks, schur_pairs=res.schur_pairs, pc_block=res.pc_block, pc_levels=res.pc_levels),
                certificate=dict(pcg_residual_max=res.pcg_residual_max, pcg_residual_last=res.pcg_residual_last, grad_inf=res.grad_inf, pcg_refined_trials=res.pcg_refined_trials),
                ms=dict(total=res.ms_total, build=res.ms_build, schur=res.ms_schur, solve=res.ms_solve, update=res.ms_update))


def LocalBundleAdjustmentStore(kf, kf_slots, n_local, mp, mp_slots, scale_factor=1.2, apply_erase=True, stages=None, stop=None, solver=0):
    """Optimizer::LocalBundleAdjustment on store records (corb_local_ba_store): kf_slots = lLocalKeyFrames (the first n_local) + lFixedCameras, mp_slots =
    lLocalMapPoints.  The records receive the poses / positions, vToErase (apply_erase) and UpdateNormalAndDepth; returns the estimates and the erased
    observations as (index into kf_slots, index into mp_slots) rows."""
    ks = np.ascontiguousarray(kf_slots, np.int32); ms = np.ascontiguousarray(mp_slots, np.int32)
    stages = LOCAL_BA_STAGES if stages is None else stages
    oposes = np.zeros((len(ks), 16), np.float32); opoints = np.zeros((len(ms), 3), np.float32)
    res = _BAResult(_p(oposes), _p(opoints), None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0)
    st = (BAStage * len(stages))(*[BAStage(*s) for s in stages])
    cap = max(1, len(ms) * max(1, len(ks)))
    pairs = np.zeros((cap, 2), np.int32); ne = C.c_int(0)
    one = C.c_int(1)
    sp = None if stop is None else (C.cast(C.byref(one), C.c_void_p) if stop == "before" else C.cast(C.byref(res, _BAResult.iters_done.offset), C.c_void_p))
    opt = BAOptions(solver, 0.0, 0, 0, 0)
    _chk(load().corb_local_ba_store(kf.h, _p(ks), int(n_local), len(ks), mp.h, _p(ms), len(ms), st, len(stages), C.c_float(scale_factor), int(bool(apply_erase)), sp,
                                    C.byref(res), _p(pairs), cap, C.byref(ne), C.byref(opt)), "corb_local_ba_store")
    return dict(poses=oposes.reshape(-1, 4, 4), points=opoints, erase=pairs[: ne.value].copy(), iters_done=res.iters_done, trials=res.trials_total, ms_total=res.ms_total,
                structure=dict(free_poses=res.free_poses, free_points=res.free_points, active_edges=res.active_edges, nnz_blocks=res.nnz_blocks, schur_pairs=res.schur_pairs),
                device_route=bool(res.reserved0))


class Comm:
    """RCCL communicator of the client / server ranks (corb_comm_*): rank 0 creates the 128-byte id, every rank gets it by the job's own means."""

    @staticmethod
    def unique_id():
        b = (C.c_char * 128)()
        _chk(load().corb_comm_unique_id(b), "corb_comm_unique_id")
        return bytes(b)

    def __init__(self, unique_id, rank, world, device=0, _handle=None):
        self.h = C.c_void_p(); self.rank = rank; self.world = world
        if _handle is not None:
            self.h = C.c_void_p(_handle); return
        buf = (C.c_char * 128).from_buffer_copy(unique_id)
        _chk(load().corb_comm_create(buf, rank, world, device, C.byref(self.h)), "corb_comm_create")

    @staticmethod
    def local(world, devices=None):
        """corb_comm_create_local: `world` communicators over the in-process transport (one host thread per rank)"""
        arr = (C.c_void_p * world)()
        dv = None if devices is None else np.ascontiguousarray(devices, np.int32)
        _chk(load().corb_comm_create_local(world, _p(dv), arr), "corb_comm_create_local")
        return [Comm(None, r, world, _handle=arr[r]) for r in range(world)]

    def close(self):
        if self.h:
            load().corb_comm_destroy(self.h); self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def map_push(self, store, slots, root=0, dst_first=None):
        """corb_map_push: every rank sends the records of `slots` to `root`, which files rank r's records from slot dst_first[r] on; returns the per-rank counts on the root"""
        sl = np.ascontiguousarray(slots, np.int32)
        df = None if dst_first is None else np.ascontiguousarray(dst_first, np.int32)
        cnt = np.zeros(self.world, np.int32)
        _chk(load().corb_map_push(self.h, store.h, _p(sl) if len(sl) else None, len(sl), root, _p(df), _p(cnt)), "corb_map_push")
        return cnt if self.rank == root else None

    def map_push_ex(self, kf, kf_slots, mp=None, mp_slots=(), root=0, kf_dst_first=None, mp_dst_first=None):
        """corb_map_push_ex: keyframe AND map-point records to the root; returns (kf counts, mp counts) on the root"""
        ks = np.ascontiguousarray(kf_slots, np.int32); ms = np.ascontiguousarray(mp_slots, np.int32)
        kd = None if kf_dst_first is None else np.ascontiguousarray(kf_dst_first, np.int32); md = None if mp_dst_first is None else np.ascontiguousarray(mp_dst_first, np.int32)
        kc = np.zeros(self.world, np.int32); mc = np.zeros(self.world, np.int32)
        p = _MapPush(kf.h if kf is not None else None, _p(ks) if len(ks) else None, len(ks), mp.h if mp is not None else None, _p(ms) if len(ms) else None, len(ms),
                     _p(kd), _p(md), _p(kc), _p(mc))
        _chk(load().corb_map_push_ex(self.h, C.byref(p), root), "corb_map_push_ex")
        return (kc, mc) if self.rank == root else None

    def map_push_setup(self, root, kf=None, mp=None, kf_dst_first=None, mp_dst_first=None):
        """corb_map_push_setup (collective, once): the root's layout to every rank, for the asynchronous pushes"""
        kd = None if kf_dst_first is None else np.ascontiguousarray(kf_dst_first, np.int32); md = None if mp_dst_first is None else np.ascontiguousarray(mp_dst_first, np.int32)
        L = load(); L.corb_map_push_setup.restype = C.c_int; L.corb_map_push_setup.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _chk(L.corb_map_push_setup(self.h, int(root), kf.h if kf is not None else None, mp.h if mp is not None else None, _p(kd), _p(md)), "corb_map_push_setup")

    def map_push_begin(self, kf, kf_slots, mp=None, mp_slots=(), root=0):
        """corb_map_push_begin: one header all-gather, records enqueued; returns at once (map_push_wait completes it and returns the counts on the root)"""
        ks = np.ascontiguousarray(kf_slots, np.int32); ms = np.ascontiguousarray(mp_slots, np.int32)
        kc = np.zeros(self.world, np.int32); mc = np.zeros(self.world, np.int32)
        p = _MapPush(kf.h if kf is not None else None, _p(ks) if len(ks) else None, len(ks), mp.h if mp is not None else None, _p(ms) if len(ms) else None, len(ms),
                     None, None, _p(kc), _p(mc))
        self._flight = (ks, ms, kc, mc, p, root)                # (the arrays the C side keeps pointers to until the wait)
        L = load(); L.corb_map_push_begin.restype = C.c_int; L.corb_map_push_begin.argtypes = [C.c_void_p, C.POINTER(_MapPush), C.c_int]
        _chk(L.corb_map_push_begin(self.h, C.byref(p), int(root)), "corb_map_push_begin")

    def map_push_wait(self):
        L = load(); L.corb_map_push_wait.restype = C.c_int; L.corb_map_push_wait.argtypes = [C.c_void_p]
        _chk(L.corb_map_push_wait(self.h), "corb_map_push_wait")
        fl = getattr(self, "_flight", None); self._flight = None
        return (fl[2], fl[3]) if fl is not None and self.rank == fl[5] else None
