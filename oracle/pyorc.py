"""ctypes view of the CPU oracle (oracle/liborc.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see oracle/orc.h).  The product package never does.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28
EDGE_DTYPE = np.dtype([("pose", "<i4"), ("point", "<i4"), ("u", "<f4"), ("v", "<f4"),
                       ("ur", "<f4"), ("inv_sigma2", "<f4")])
assert EDGE_DTYPE.itemsize == 24


def build(force=False):
    so = os.path.join(_HERE, "liborc.so")
    srcs = [os.path.join(_HERE, f) for f in ("orc_orb.c", "orc_match.c", "orc_ba.c", "orc_proj.c", "orc.h", "brief_pattern.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs if os.path.exists(s)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def _ptr(a, ty=C.c_void_p):
    return a.ctypes.data_as(ty) if a is not None else None


class _StereoParams(C.Structure):
    _fields_ = [("bf", C.c_float), ("mb", C.c_float), ("nlevels", C.c_int),
                ("scale", C.c_void_p), ("inv_scale", C.c_void_p)]


class _FeatVec(C.Structure):
    _fields_ = [("n_nodes", C.c_int), ("node_id", C.c_void_p), ("offset", C.c_void_p), ("idx", C.c_void_p)]


class _TriParams(C.Structure):
    _fields_ = [("F12", C.c_float * 9), ("ex", C.c_float), ("ey", C.c_float), ("nlevels", C.c_int),
                ("scale2", C.c_void_p), ("sigma2_2", C.c_void_p)]


class _BAProblem(C.Structure):
    _fields_ = [("n_poses", C.c_int), ("n_points", C.c_int), ("n_edges", C.c_int),
                ("poses", C.c_void_p), ("pose_fixed", C.c_void_p), ("points", C.c_void_p),
                ("point_fixed", C.c_void_p), ("edges", C.c_void_p),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("bf", C.c_float), ("intr", C.c_void_p)]


TRACKED_DTYPE = np.dtype([("proj_x", "<f4"), ("proj_y", "<f4"), ("proj_xr", "<f4"), ("view_cos", "<f4"), ("level", "<i4"),
                          ("valid", "u1"), ("claims", "u1"), ("pad", "u1", 2)])
LAST_DTYPE = np.dtype([("world", "<f4", 3), ("angle", "<f4"), ("octave", "<i4"), ("valid", "u1"), ("claims", "u1"), ("pad", "u1", 2)])
assert TRACKED_DTYPE.itemsize == 24 and LAST_DTYPE.itemsize == 24
MP_DTYPE = np.dtype([("world", "<f4", 3), ("normal", "<f4", 3), ("min_distance", "<f4"), ("max_distance", "<f4"), ("angle", "<f4"),
                     ("valid", "u1"), ("pad", "u1", 3)])
assert MP_DTYPE.itemsize == 40


class _Sim3Problem(C.Structure):
    _fields_ = [("n", C.c_int), ("p1c", C.c_void_p), ("p2c", C.c_void_p), ("obs1", C.c_void_p), ("obs2", C.c_void_p),
                ("inv_sigma2_1", C.c_void_p), ("inv_sigma2_2", C.c_void_p)] + [(k, C.c_float) for k in ("fx1", "fy1", "cx1", "cy1", "fx2", "fy2", "cx2", "cy2")]


class _KeyFrameView(C.Structure):
    _fields_ = [("keys_un", C.c_void_p), ("u_right", C.c_void_p), ("desc", C.c_void_p), ("n", C.c_int32),
                ("min_x", C.c_float), ("min_y", C.c_float), ("max_x", C.c_float), ("max_y", C.c_float),
                ("scale", C.c_void_p), ("inv_level_sigma2", C.c_void_p), ("nlevels", C.c_int32), ("log_scale_factor", C.c_float),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("bf", C.c_float)]


class _FrameView(C.Structure):
    _fields_ = [("keys_un", C.c_void_p), ("u_right", C.c_void_p), ("desc", C.c_void_p), ("n", C.c_int), ("claimed", C.c_void_p),
                ("min_x", C.c_float), ("min_y", C.c_float), ("max_x", C.c_float), ("max_y", C.c_float), ("scale", C.c_void_p), ("nlevels", C.c_int)]


class _BAStage(C.Structure):
    _fields_ = [("iterations", C.c_int), ("robust", C.c_int), ("chi2_mono", C.c_float), ("chi2_stereo", C.c_float),
                ("check_depth", C.c_int), ("recompute_inactive", C.c_int), ("allow_reactivate", C.c_int),
                ("reset_estimates", C.c_int), ("float_compare", C.c_int), ("huber_mono", C.c_float), ("huber_stereo", C.c_float)]


_HM, _HS = float(np.float32(np.sqrt(5.991))), float(np.float32(np.sqrt(7.815)))
LOCAL_BA_STAGES = [(5, 1, 5.991, 7.815, 1, 0, 0, 0, 0, _HM, _HS), (10, 0, 5.991, 7.815, 1, 0, 1, 0, 0, _HM, _HS)]   # final test on every edge (Optimizer.cc:763-790)
POSE_OPT_STAGES = [(10, 1, 5.991, 7.815, 0, 1, 1, 1, 1, _HM, _HS)] * 3 + [(10, 0, 5.991, 7.815, 0, 1, 1, 1, 1, _HM, _HS)]


class _BAResult(C.Structure):
    _fields_ = [("poses", C.c_void_p), ("points", C.c_void_p), ("chi2", C.c_void_p), ("lam", C.c_void_p),
                ("iters_done", C.c_int), ("trials_total", C.c_int)]


_lib = None


_default_native = False


def use_native(on=True):
    """make the -O3 -march=native build (the reference's own flags) the default library: for the timing legs of bench.py"""
    global _default_native
    _default_native = bool(on)


def lib(native=False):
    global _lib
    if native or _default_native:
        build()
        L = C.CDLL(os.path.join(_HERE, "liborc_native.so"))
        _proto(L)
        return L
    if _lib is None:
        _lib = C.CDLL(build())
        _proto(_lib)
    return _lib


def _proto(L):
    L.orc_orb_create.restype = C.c_void_p
    L.orc_orb_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
    L.orc_orb_destroy.argtypes = [C.c_void_p]
    L.orc_orb_extract.restype = C.c_int
    L.orc_orb_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.orc_orb_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 6
    L.orc_orb_level_dims.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_orb_level_data.restype = C.c_void_p
    L.orc_orb_level_data.argtypes = [C.c_void_p, C.c_int]
    L.orc_orb_blur_data.restype = C.c_void_p
    L.orc_orb_blur_data.argtypes = [C.c_void_p, C.c_int]
    L.orc_orb_level_candidates.restype = C.c_int
    L.orc_orb_level_candidates.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.orc_orb_level_count.restype = C.c_int
    L.orc_orb_level_count.argtypes = [C.c_void_p, C.c_int]
    L.orc_resize_linear_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.orc_gaussian_blur7_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.orc_fast9_16.restype = C.c_int
    L.orc_fast9_16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.orc_fast_atan2.restype = C.c_float
    L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
    L.orc_sincosf.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.orc_ic_angle.restype = C.c_float
    L.orc_ic_angle.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.orc_distribute_octree.restype = C.c_int
    L.orc_distribute_octree.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.orc_descriptor_distance.restype = C.c_int
    L.orc_descriptor_distance.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_stereo_match.restype = C.c_int
    L.orc_stereo_match.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                   C.POINTER(_StereoParams), C.c_void_p, C.c_void_p]
    L.orc_search_by_bow.restype = C.c_int
    L.orc_search_by_bow.argtypes = [C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(_FeatVec),
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(_FeatVec),
                                    C.c_float, C.c_int, C.c_void_p]
    L.orc_search_for_triangulation.restype = C.c_int
    L.orc_search_for_triangulation.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(_FeatVec),
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(_FeatVec),
                                               C.POINTER(_TriParams), C.c_int, C.c_int, C.c_void_p]
    L.orc_search_by_projection_map.restype = C.c_int
    L.orc_search_by_projection_map.argtypes = [C.POINTER(_FrameView), C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p]
    L.orc_optimize_essential_graph.restype = C.c_int
    L.orc_optimize_essential_graph.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_essential_graph_apply.restype = None
    L.orc_essential_graph_apply.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.orc_optimize_sim3.restype = C.c_int
    L.orc_optimize_sim3.argtypes = [C.POINTER(_Sim3Problem), C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_distinctive_descriptors.restype = None
    L.orc_distinctive_descriptors.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.orc_mappoint_replace.restype = C.c_int
    L.orc_mappoint_replace.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_rebase_map.restype = None
    L.orc_rebase_map.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.orc_search_by_projection_reloc.restype = C.c_int
    L.orc_search_by_projection_reloc.argtypes = [C.POINTER(_KeyFrameView), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p]
    L.orc_search_for_initialization.restype = C.c_int
    L.orc_search_for_initialization.argtypes = [C.POINTER(_FrameView), C.POINTER(_FrameView), C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
    L.orc_search_by_projection_scw.restype = C.c_int
    L.orc_search_by_projection_scw.argtypes = [C.POINTER(_KeyFrameView), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p]
    L.orc_fuse.restype = C.c_int
    L.orc_fuse.argtypes = [C.POINTER(_KeyFrameView), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    L.orc_search_by_sim3.restype = C.c_int
    L.orc_search_by_sim3.argtypes = [C.POINTER(_KeyFrameView), C.POINTER(_KeyFrameView)] + [C.c_void_p] * 6 + [C.c_float, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    L.orc_search_by_projection_frame.restype = C.c_int
    L.orc_search_by_projection_frame.argtypes = [C.POINTER(_FrameView), C.c_void_p, C.c_void_p] + [C.c_float] * 6 + [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p]
    L.orc_ba_solve_staged.restype = C.c_int
    L.orc_ba_solve_staged.argtypes = [C.POINTER(_BAProblem), C.POINTER(_BAStage), C.c_int, C.c_void_p, C.POINTER(_BAResult), C.c_void_p]
    L.orc_ba_solve.restype = C.c_int
    L.orc_ba_solve.argtypes = [C.POINTER(_BAProblem), C.c_int, C.c_int, C.c_void_p, C.POINTER(_BAResult)]


class FeatVec:
    """Flat DBoW2::FeatureVector: ascending node ids, CSR offsets, feature indices."""

    def __init__(self, node_id, offset, idx):
        self.node_id = np.ascontiguousarray(node_id, dtype=np.uint32)
        self.offset = np.ascontiguousarray(offset, dtype=np.int32)
        self.idx = np.ascontiguousarray(idx, dtype=np.uint32)
        assert len(self.offset) == len(self.node_id) + 1

    def c(self):
        return _FeatVec(len(self.node_id), _ptr(self.node_id), _ptr(self.offset), _ptr(self.idx))


class Extractor:
    def __init__(self, nfeatures=2000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7, native=False):
        self.L = lib(native)
        self.nlevels = nlevels
        self.nfeatures = nfeatures
        self.h = self.L.orc_orb_create(nfeatures, scale_factor, nlevels, ini_th, min_th)
        assert self.h

    def __del__(self):
        try:
            self.L.orc_orb_destroy(self.h)
        except Exception:
            pass

    def tables(self):
        n = self.nlevels
        sc, isc, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        quota = np.zeros(n, np.int32)
        umax = np.zeros(16, np.int32)
        self.L.orc_orb_tables(self.h, _ptr(sc), _ptr(isc), _ptr(s2), _ptr(is2), _ptr(quota), _ptr(umax))
        return dict(scale=sc, inv_scale=isc, sigma2=s2, inv_sigma2=is2, quota=quota, umax=umax)

    def extract(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape if img.size else (0, 0)
        cap = self.nfeatures + 64 * self.nlevels + 1024
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = self.L.orc_orb_extract(self.h, _ptr(img), w, h, w, _ptr(kps), _ptr(desc), cap)
        assert n >= 0
        return kps[:n].copy(), desc[:n].copy()

    def level(self, l):
        w, h = C.c_int(), C.c_int()
        self.L.orc_orb_level_dims(self.h, l, C.byref(w), C.byref(h))
        p = self.L.orc_orb_level_data(self.h, l)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (h.value, w.value)).copy()

    def blurred(self, l):
        w, h = C.c_int(), C.c_int()
        self.L.orc_orb_level_dims(self.h, l, C.byref(w), C.byref(h))
        p = self.L.orc_orb_blur_data(self.h, l)
        if not p:
            return None
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (h.value, w.value)).copy()

    def candidates(self, l):
        n = self.L.orc_orb_level_candidates(self.h, l, None, 0)
        out = np.zeros(max(n, 1), KP_DTYPE)
        self.L.orc_orb_level_candidates(self.h, l, _ptr(out), n)
        return out[:n]

    def level_count(self, l):
        return self.L.orc_orb_level_count(self.h, l)


def resize_linear(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_linear_u8(_ptr(src), src.shape[1], src.shape[0], src.shape[1], _ptr(dst), dw, dh, dw)
    return dst


def gaussian_blur7(src):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros_like(src)
    lib().orc_gaussian_blur7_u8(_ptr(src), src.shape[1], src.shape[0], src.shape[1], _ptr(dst), src.shape[1])
    return dst


def fast(img, threshold, nms=True):
    img = np.ascontiguousarray(img, np.uint8)
    cap = img.size
    out = np.zeros(max(cap, 1), KP_DTYPE)
    n = lib().orc_fast9_16(_ptr(img), img.shape[1], img.shape[0], img.shape[1], threshold, int(nms), _ptr(out), cap)
    return out[:n].copy()


def fast_atan2(y, x):
    return lib().orc_fast_atan2(float(y), float(x))


def sincosf(x):
    s, c = C.c_float(), C.c_float()
    lib().orc_sincosf(float(x), C.byref(s), C.byref(c))
    return s.value, c.value


def ic_angle(img, cx, cy):
    img = np.ascontiguousarray(img, np.uint8)
    return lib().orc_ic_angle(_ptr(img), img.shape[1], cx, cy)


def distribute_octree(kps, minX, maxX, minY, maxY, N):
    kps = np.ascontiguousarray(kps, KP_DTYPE)
    out = np.zeros(len(kps) + N + 8, KP_DTYPE)
    n = lib().orc_distribute_octree(_ptr(kps), len(kps), minX, maxX, minY, maxY, N, _ptr(out), len(out))
    return out[:n].copy()


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().orc_descriptor_distance(_ptr(a), _ptr(b))


def stereo_match(exl, exr, kl, dl, kr, dr, bf, fx, scale, inv_scale):
    """Frame::ComputeStereoMatches with mb := bf/fx (float)."""
    kl = np.ascontiguousarray(kl, KP_DTYPE); kr = np.ascontiguousarray(kr, KP_DTYPE)
    dl = np.ascontiguousarray(dl, np.uint8); dr = np.ascontiguousarray(dr, np.uint8)
    scale = np.ascontiguousarray(scale, np.float32); inv_scale = np.ascontiguousarray(inv_scale, np.float32)
    mb = np.float32(bf) / np.float32(fx)
    p = _StereoParams(float(np.float32(bf)), float(mb), len(scale), _ptr(scale), _ptr(inv_scale))
    ur = np.zeros(len(kl), np.float32); depth = np.zeros(len(kl), np.float32)
    n = exl.L.orc_stereo_match(exl.h, exr.h, _ptr(kl), _ptr(dl), len(kl), _ptr(kr), _ptr(dr), len(kr),
                               C.byref(p), _ptr(ur), _ptr(depth))
    return ur, depth, n


def search_by_bow(variant, desc1, angle1, valid1, fv1, desc2, angle2, valid2, fv2, nnratio, check_ori):
    desc1 = np.ascontiguousarray(desc1, np.uint8); desc2 = np.ascontiguousarray(desc2, np.uint8)
    angle1 = np.ascontiguousarray(angle1, np.float32); angle2 = np.ascontiguousarray(angle2, np.float32)
    valid1 = np.ascontiguousarray(valid1, np.uint8); valid2 = np.ascontiguousarray(valid2, np.uint8)
    n1, n2 = len(desc1), len(desc2)
    out = np.zeros(max(n2 if variant == 0 else n1, 1), np.int32)
    c1, c2 = fv1.c(), fv2.c()
    n = lib().orc_search_by_bow(variant, _ptr(desc1), _ptr(angle1), _ptr(valid1), n1, C.byref(c1),
                                _ptr(desc2), _ptr(angle2), _ptr(valid2), n2, C.byref(c2),
                                float(nnratio), int(check_ori), _ptr(out))
    return out[: (n2 if variant == 0 else n1)], n


def search_for_triangulation(desc1, kp1, ur1, mp1, fv1, desc2, kp2, ur2, mp2, fv2, F12, ex, ey, scale2, sigma2_2,
                             only_stereo, check_ori):
    desc1 = np.ascontiguousarray(desc1, np.uint8); desc2 = np.ascontiguousarray(desc2, np.uint8)
    kp1 = np.ascontiguousarray(kp1, KP_DTYPE); kp2 = np.ascontiguousarray(kp2, KP_DTYPE)
    ur1 = np.ascontiguousarray(ur1, np.float32); ur2 = np.ascontiguousarray(ur2, np.float32)
    mp1 = np.ascontiguousarray(mp1, np.uint8); mp2 = np.ascontiguousarray(mp2, np.uint8)
    scale2 = np.ascontiguousarray(scale2, np.float32); sigma2_2 = np.ascontiguousarray(sigma2_2, np.float32)
    p = _TriParams()
    F = np.asarray(F12, np.float32).reshape(9)
    for i in range(9):
        p.F12[i] = float(F[i])
    p.ex, p.ey, p.nlevels = float(np.float32(ex)), float(np.float32(ey)), len(scale2)
    p.scale2, p.sigma2_2 = _ptr(scale2), _ptr(sigma2_2)
    out = np.zeros((max(len(kp1), 1), 2), np.int32)
    c1, c2 = fv1.c(), fv2.c()
    n = lib().orc_search_for_triangulation(_ptr(desc1), _ptr(kp1), _ptr(ur1), _ptr(mp1), len(kp1), C.byref(c1),
                                           _ptr(desc2), _ptr(kp2), _ptr(ur2), _ptr(mp2), len(kp2), C.byref(c2),
                                           C.byref(p), int(only_stereo), int(check_ori), _ptr(out))
    return out[:n].copy(), n


def ba_set_solver(solver, native=False):
    """reduced-system solver of the BA oracle: 0 auto (dense up to 256 free poses, block-sparse LDL^T above), 1 dense, 2 block-sparse"""
    L = lib(native)
    L.orc_ba_set_solver.restype = None; L.orc_ba_set_solver.argtypes = [C.c_int]
    L.orc_ba_set_solver(int(solver))


def ba_solve(poses, pose_fixed, points, point_fixed, edges, fx, fy, cx, cy, bf, iters=10, robust=False, native=False, intr=None):
    poses = np.ascontiguousarray(poses, np.float32).reshape(-1, 16)
    points = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
    pose_fixed = np.ascontiguousarray(pose_fixed, np.uint8)
    point_fixed = np.ascontiguousarray(point_fixed, np.uint8)
    edges = np.ascontiguousarray(edges, EDGE_DTYPE)
    if intr is not None:
        intr = np.ascontiguousarray(intr, np.float32).reshape(len(poses), 5)       # per-keyframe fx, fy, cx, cy, bf
    prob = _BAProblem(len(poses), len(points), len(edges), _ptr(poses), _ptr(pose_fixed), _ptr(points),
                      _ptr(point_fixed), _ptr(edges), fx, fy, cx, cy, bf, _ptr(intr) if intr is not None else None)
    oposes = np.zeros_like(poses); opoints = np.zeros_like(points)
    chi2 = np.zeros(iters + 1, np.float64); lam = np.zeros(max(iters, 1), np.float64)
    res = _BAResult(_ptr(oposes), _ptr(opoints), _ptr(chi2), _ptr(lam), 0, 0)
    rc = lib(native).orc_ba_solve(C.byref(prob), iters, int(robust), None, C.byref(res))
    if rc != 0:
        raise RuntimeError("orc_ba_solve failed rc=%d" % rc)
    return dict(poses=oposes.reshape(-1, 4, 4), points=opoints, chi2=chi2[: res.iters_done + 1],
                lam=lam[: res.iters_done], iters_done=res.iters_done, trials=res.trials_total)


def ba_solve_staged(poses, pose_fixed, points, point_fixed, edges, fx, fy, cx, cy, bf, stages, native=False, intr=None, stop=None):
    poses = np.ascontiguousarray(poses, np.float32).reshape(-1, 16)
    points = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
    pose_fixed = np.ascontiguousarray(pose_fixed, np.uint8); point_fixed = np.ascontiguousarray(point_fixed, np.uint8)
    edges = np.ascontiguousarray(edges, EDGE_DTYPE)
    if intr is not None:
        intr = np.ascontiguousarray(intr, np.float32).reshape(len(poses), 5)       # per-keyframe fx, fy, cx, cy, bf
    prob = _BAProblem(len(poses), len(points), len(edges), _ptr(poses), _ptr(pose_fixed), _ptr(points),
                      _ptr(point_fixed), _ptr(edges), fx, fy, cx, cy, bf, _ptr(intr) if intr is not None else None)
    oposes = np.zeros_like(poses); opoints = np.zeros_like(points)
    res = _BAResult(_ptr(oposes), _ptr(opoints), None, None, 0, 0)
    st = (_BAStage * len(stages))(*[_BAStage(*s) for s in stages])
    outl = np.zeros(max(len(edges), 1), np.uint8)
    # pbStopFlag for tests: "before" = raised before the call; "after_first_stage" = the flag aliases the result's iteration counter, which turns
    # non-zero exactly when the first optimize() returns (deterministic stand-in for LocalMapping::InterruptBA during the first round)
    one = C.c_int(1)
    sp = None if stop is None else (C.cast(C.byref(one), C.c_void_p) if stop == "before" else C.cast(C.byref(res, _BAResult.iters_done.offset), C.c_void_p))
    rc = lib(native).orc_ba_solve_staged(C.byref(prob), st, len(stages), sp, C.byref(res), _ptr(outl))
    if rc != 0:
        raise RuntimeError("orc_ba_solve_staged failed rc=%d" % rc)
    return dict(poses=oposes.reshape(-1, 4, 4), points=opoints, outlier=outl[: len(edges)].copy(), iters_done=res.iters_done, trials=res.trials_total)


def _frame_view(fr, keep):
    k = np.ascontiguousarray(fr["keys_un"], KP_DTYPE); ur = np.ascontiguousarray(fr["u_right"], np.float32)
    d = np.ascontiguousarray(fr["desc"], np.uint8); cl = np.ascontiguousarray(fr["claimed"], np.uint8); sc = np.ascontiguousarray(fr["scale"], np.float32)
    keep += [k, ur, d, cl, sc]
    return _FrameView(_ptr(k), _ptr(ur), _ptr(d), len(k), _ptr(cl), fr["min_x"], fr["min_y"], fr["max_x"], fr["max_y"], _ptr(sc), len(sc))


def search_by_projection_map(frame, mps, mp_desc, th, nnratio):
    keep = []; fv = _frame_view(frame, keep)
    mps = np.ascontiguousarray(mps, TRACKED_DTYPE); mp_desc = np.ascontiguousarray(mp_desc, np.uint8)
    match = np.zeros(max(len(frame["keys_un"]), 1), np.int32)
    n = lib().orc_search_by_projection_map(C.byref(fv), _ptr(mps), _ptr(mp_desc), len(mps), float(th), float(nnratio), _ptr(match))
    return match[: len(frame["keys_un"])].copy(), n


def is_in_frustum(Tcw, world, normal, min_distance, max_distance, fx, fy, cx, cy, bf, min_x, max_x, min_y, max_y, log_scale_factor, nlevels, cos_limit=0.5):
    """bool Frame::isInFrustum(MapPoint*, viewingCosLimit) for an array of MapPoints (corbslam_client/src/Frame.cc:270-329; MapPoint::PredictScale
    corbslam_client/src/MapPoint.cc:500-514; mOw = -mRcw.t()*mtcw, Frame.cc UpdatePoseMatrices) in numpy with the reference's arithmetic: float operands,
    cv::gemm / cv::norm / Mat::dot accumulate in double and round once, libm log on a float DEFINED as (float)log((double)x) like orc_proj.c:217.
    Returns a TRACKED_DTYPE array (valid = mbTrackInView; claims left 0)."""
    f = np.float32
    T = np.asarray(Tcw, f).reshape(4, 4); P = np.asarray(world, f).reshape(-1, 3); Pn = np.asarray(normal, f).reshape(-1, 3)
    R = T[:3, :3].astype(np.float64); t = T[:3, 3].astype(np.float64)
    Pd = P.astype(np.float64)
    Pc = np.stack([((R[i, 0] * Pd[:, 0] + R[i, 1] * Pd[:, 1]) + R[i, 2] * Pd[:, 2]) + t[i] for i in range(3)], 1).astype(f)
    Ow = np.array([-((R[0, i] * t[0] + R[1, i] * t[1]) + R[2, i] * t[2]) for i in range(3)]).astype(f)
    out = np.zeros(len(P), TRACKED_DTYPE)
    with np.errstate(divide="ignore", invalid="ignore"):
        ok = Pc[:, 2] > f(0)
        invz = f(1) / Pc[:, 2]
        u = (f(fx) * Pc[:, 0]) * invz + f(cx); v = (f(fy) * Pc[:, 1]) * invz + f(cy)
        ok &= ~((u < f(min_x)) | (u > f(max_x)) | (v < f(min_y)) | (v > f(max_y)))
        maxD = f(1.2) * np.asarray(max_distance, f); minD = f(0.8) * np.asarray(min_distance, f)
        PO = P - Ow
        POd = PO.astype(np.float64)
        dist = np.sqrt((POd[:, 0] * POd[:, 0] + POd[:, 1] * POd[:, 1]) + POd[:, 2] * POd[:, 2]).astype(f)
        ok &= ~((dist < minD) | (dist > maxD))
        Pnd = Pn.astype(np.float64)
        dot = (POd[:, 0] * Pnd[:, 0] + POd[:, 1] * Pnd[:, 1]) + POd[:, 2] * Pnd[:, 2]
        view = (dot / dist.astype(np.float64)).astype(f)
        ok &= ~(view < f(cos_limit))
        ratio = np.asarray(max_distance, f) / dist
        lg = np.log(ratio.astype(np.float64)).astype(f)
        lvl = np.ceil(lg / f(log_scale_factor))
    lvl = np.where(np.isfinite(lvl), lvl, 0).astype(np.int64)
    lvl = np.clip(lvl, 0, nlevels - 1)
    out["valid"] = ok
    out["proj_x"] = np.where(ok, u, 0); out["proj_y"] = np.where(ok, v, 0); out["proj_xr"] = np.where(ok, u - f(bf) * invz, 0)
    out["view_cos"] = np.where(ok, view, 0); out["level"] = np.where(ok, lvl, 0)
    return out


def search_by_projection_frame(cur, Tcw, Tlw, fx, fy, cx, cy, bf, mb, last, last_desc, th, mono, check_ori):
    keep = []; fv = _frame_view(cur, keep)
    Tcw = np.ascontiguousarray(Tcw, np.float32).reshape(16); Tlw = np.ascontiguousarray(Tlw, np.float32).reshape(16)
    last = np.ascontiguousarray(last, LAST_DTYPE); last_desc = np.ascontiguousarray(last_desc, np.uint8)
    match = np.zeros(max(len(cur["keys_un"]), 1), np.int32)
    n = lib().orc_search_by_projection_frame(C.byref(fv), _ptr(Tcw), _ptr(Tlw), fx, fy, cx, cy, bf, mb, _ptr(last), _ptr(last_desc), len(last),
                                             float(th), int(mono), int(check_ori), _ptr(match))
    return match[: len(cur["keys_un"])].copy(), n


def _kf_view(kf, keep):
    k = np.ascontiguousarray(kf["keys_un"], KP_DTYPE); ur = np.ascontiguousarray(kf["u_right"], np.float32)
    d = np.ascontiguousarray(kf["desc"], np.uint8); sc = np.ascontiguousarray(kf["scale"], np.float32); s2 = np.ascontiguousarray(kf["inv_level_sigma2"], np.float32)
    keep += [k, ur, d, sc, s2]
    return _KeyFrameView(_ptr(k), _ptr(ur), _ptr(d), len(k), kf["min_x"], kf["min_y"], kf["max_x"], kf["max_y"], _ptr(sc), _ptr(s2), len(sc),
                         kf["log_scale_factor"], kf["fx"], kf["fy"], kf["cx"], kf["cy"], kf["bf"])


def search_by_projection_reloc(cur, claimed, Tcw, pts, desc, th, orb_dist, check_ori):
    keep = []; kv = _kf_view(cur, keep)
    claimed = np.ascontiguousarray(claimed, np.uint8); Tcw = np.ascontiguousarray(Tcw, np.float32).reshape(16)
    pts = np.ascontiguousarray(pts, MP_DTYPE); desc = np.ascontiguousarray(desc, np.uint8)
    match = np.zeros(max(len(cur["keys_un"]), 1), np.int32)
    n = lib().orc_search_by_projection_reloc(C.byref(kv), _ptr(claimed), _ptr(Tcw), _ptr(pts), _ptr(desc), len(pts), float(th), int(orb_dist), int(check_ori), _ptr(match))
    return match[: len(cur["keys_un"])].copy(), n


def search_for_initialization(f1, f2, prev_matched, window_size, nnratio=0.9, check_ori=True):
    """ORBmatcher::SearchForInitialization (ORBmatcher.cc:540-655): f1 / f2 = frame dicts (keys_un, desc, ...); prev_matched n1 x 2 float32.
    Returns (vnMatches12, vbPrevMatched after the call, nmatches)."""
    keep = []
    def fv(fr):
        fr = dict(fr); fr.setdefault("claimed", np.zeros(len(fr["keys_un"]), np.uint8)); fr.setdefault("u_right", -np.ones(len(fr["keys_un"]), np.float32))
        return _frame_view(fr, keep)
    v1, v2 = fv(f1), fv(f2)
    pm = np.array(prev_matched, np.float32, copy=True).reshape(-1, 2); assert len(pm) == len(f1["keys_un"])
    m = np.zeros(max(len(pm), 1), np.int32)
    n = lib().orc_search_for_initialization(C.byref(v1), C.byref(v2), _ptr(pm), int(window_size), float(nnratio), int(check_ori), _ptr(m))
    return m[: len(pm)].copy(), pm, n


def search_by_projection_scw(kf, claimed, Scw, pts, desc, th):
    """ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) (ORBmatcher.cc:425-538): match[idx] = point index written into vpMatched[idx] or -1"""
    keep = []; kv = _kf_view(kf, keep)
    claimed = np.ascontiguousarray(claimed, np.uint8); Scw = np.ascontiguousarray(Scw, np.float32).reshape(16)
    pts = np.ascontiguousarray(pts, MP_DTYPE); desc = np.ascontiguousarray(desc, np.uint8)
    match = np.zeros(max(len(kf["keys_un"]), 1), np.int32)
    n = lib().orc_search_by_projection_scw(C.byref(kv), _ptr(claimed), _ptr(Scw), _ptr(pts), _ptr(desc), len(pts), float(th), _ptr(match))
    return match[: len(kf["keys_un"])].copy(), n


def fuse(kf, T, Ow, sim3, pts, desc, th):
    keep = []; kv = _kf_view(kf, keep)
    T = np.ascontiguousarray(T, np.float32).reshape(16); Ow = np.ascontiguousarray(Ow if Ow is not None else np.zeros(3), np.float32)
    pts = np.ascontiguousarray(pts, MP_DTYPE); desc = np.ascontiguousarray(desc, np.uint8)
    bi = np.zeros(max(len(pts), 1), np.int32); bd = np.zeros(max(len(pts), 1), np.int32)
    n = lib().orc_fuse(C.byref(kv), _ptr(T), _ptr(Ow), int(sim3), _ptr(pts), _ptr(desc), len(pts), float(th), _ptr(bi), _ptr(bd))
    return bi[: len(pts)].copy(), bd[: len(pts)].copy(), n


def search_by_sim3(kf1, kf2, T1w, T2w, pts1, desc1, pts2, desc2, s12, R12, t12, th):
    keep = []; k1 = _kf_view(kf1, keep); k2 = _kf_view(kf2, keep)
    T1w = np.ascontiguousarray(T1w, np.float32).reshape(16); T2w = np.ascontiguousarray(T2w, np.float32).reshape(16)
    pts1 = np.ascontiguousarray(pts1, MP_DTYPE); pts2 = np.ascontiguousarray(pts2, MP_DTYPE)
    desc1 = np.ascontiguousarray(desc1, np.uint8); desc2 = np.ascontiguousarray(desc2, np.uint8)
    R12 = np.ascontiguousarray(R12, np.float32).reshape(9); t12 = np.ascontiguousarray(t12, np.float32).reshape(3)
    m = np.zeros(max(len(pts1), 1), np.int32)
    n = lib().orc_search_by_sim3(C.byref(k1), C.byref(k2), _ptr(T1w), _ptr(T2w), _ptr(pts1), _ptr(desc1), _ptr(pts2), _ptr(desc2),
                                 float(np.float32(s12)), _ptr(R12), _ptr(t12), float(th), _ptr(m))
    return m[: len(pts1)].copy(), n


def distinctive_descriptors(desc, offset):
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32); offset = np.ascontiguousarray(offset, np.int32)
    best = np.zeros(max(len(offset) - 1, 1), np.int32)
    lib().orc_distinctive_descriptors(_ptr(desc), _ptr(offset), len(offset) - 1, _ptr(best))
    return best[: len(offset) - 1].copy()


def mappoint_replace(id_this, id_into, obs_this, obs_into, cap_into, counters_this=(0, 0), counters_into=(0, 0)):
    """MapPoint::Replace (MapPoint.cc:277-316) on flat lists: obs_* = [(keyframe id, feature index)] ascending in the keyframe id; counters = (mnVisible, mnFound).
    Returns (status, pMP's list after, action per observation of this: 1 moved / 2 erased in its keyframe, pMP's counters after)."""
    kt = np.array([a for a, _ in obs_this], np.uint64); it = np.array([b for _, b in obs_this], np.uint32)
    ki = np.zeros(max(cap_into, len(obs_into), 1), np.uint64); ii = np.zeros(len(ki), np.uint32)
    ki[: len(obs_into)] = [a for a, _ in obs_into]; ii[: len(obs_into)] = [b for _, b in obs_into]
    n = np.array([len(obs_into)], np.int32); act = np.zeros(max(len(kt), 1), np.uint8)
    ct = np.array(counters_this, np.int32); ci = np.array(counters_into, np.int32)
    st = lib().orc_mappoint_replace(int(id_this), int(id_into), _ptr(kt), _ptr(it), len(kt), _ptr(ki), _ptr(ii), _ptr(n), int(cap_into), _ptr(act), _ptr(ct), _ptr(ci))
    return st, [(int(a), int(b)) for a, b in zip(ki[: n[0]], ii[: n[0]])], act[: len(kt)].copy(), (int(ci[0]), int(ci[1]))


def rebase_map(To2n, poses, points):
    T = np.ascontiguousarray(To2n, np.float32).reshape(16)
    P = np.array(poses, np.float32).reshape(-1, 16).copy(); X = np.array(points, np.float32).reshape(-1, 3).copy()
    lib().orc_rebase_map(_ptr(T), _ptr(P), len(P), _ptr(X), len(X))
    return P.reshape(-1, 4, 4), X


def optimize_sim3(q, th2=10.0, fix_scale=False):
    """q: dict from synth.sim3_problem.  Returns dict(R, t, s, removed, n_in, iters_done, trials)."""
    a = {k: np.ascontiguousarray(q[k], np.float32) for k in ("p1c", "p2c", "obs1", "obs2", "inv_sigma2_1", "inv_sigma2_2")}
    K = [float(np.float32(q[k])) for k in ("fx1", "fy1", "cx1", "cy1", "fx2", "fy2", "cx2", "cy2")]
    prob = _Sim3Problem(len(a["p1c"]), _ptr(a["p1c"]), _ptr(a["p2c"]), _ptr(a["obs1"]), _ptr(a["obs2"]), _ptr(a["inv_sigma2_1"]), _ptr(a["inv_sigma2_2"]), *K)
    R = np.ascontiguousarray(q["R12"], np.float64).reshape(9).copy(); t = np.ascontiguousarray(q["t12"], np.float64).reshape(3).copy(); s = np.array([q["s12"]], np.float64)
    removed = np.zeros(max(len(a["p1c"]), 1), np.uint8); it = C.c_int(); tr = C.c_int()
    n_in = lib().orc_optimize_sim3(C.byref(prob), _ptr(R), _ptr(t), _ptr(s), float(np.float32(th2)), int(fix_scale), _ptr(removed), C.byref(it), C.byref(tr))
    return dict(R=R.reshape(3, 3), t=t, s=float(s[0]), removed=removed[: len(a["p1c"])].copy(), n_in=n_in, iters_done=it.value, trials=tr.value)


def optimize_essential_graph(g, iters=20, fix_scale=False):
    """g: dict from synth.essential_graph.  Returns dict(S, chi2, iters_done, trials, Tiw, points)."""
    S = np.ascontiguousarray(g["S"], np.float64).copy(); S0 = S.copy()
    fixed = np.ascontiguousarray(g["fixed"], np.uint8); vi = np.ascontiguousarray(g["vi"], np.int32); vj = np.ascontiguousarray(g["vj"], np.int32)
    meas = np.ascontiguousarray(g["meas"], np.float64)
    chi2 = np.zeros(iters + 1, np.float64); it = C.c_int(); tr = C.c_int()
    rc = lib().orc_optimize_essential_graph(len(S), _ptr(S), _ptr(fixed), len(vi), _ptr(vi), _ptr(vj), _ptr(meas), iters, int(fix_scale), _ptr(chi2), C.byref(it), C.byref(tr))
    if rc != 0:
        raise RuntimeError("orc_optimize_essential_graph rc=%d" % rc)
    Tiw = np.zeros((len(S), 16), np.float32); pts = np.ascontiguousarray(g["points"], np.float32).copy(); ref = np.ascontiguousarray(g["ref"], np.int32)
    lib().orc_essential_graph_apply(len(S), _ptr(S0), _ptr(S), _ptr(Tiw), len(pts), _ptr(ref), _ptr(pts))
    return dict(S=S, chi2=chi2[: it.value + 1], iters_done=it.value, trials=tr.value, Tiw=Tiw.reshape(-1, 4, 4), points=pts)


# ---- Optimizer::LocalBundleAdjustment on map objects (C/src/Optimizer.cc:487-838) ----
# The map as plain Python objects, the way the reference holds it:
#   keyframe  dict(id, T (4x4 f32), fixed, bad, keys (KP_DTYPE), ur (f32), mp (per feature: map-point id or None), intr (fx, fy, cx, cy, bf), nlevels, inv_level_sigma2)
#   map point dict(id, pos (3 f32), fixed, bad, nObs (2 per stereo observation, 1 per monocular one: MapPoint::AddObservation), obs (dict keyframe id -> feature index; iterated in ascending id = std::map<LightKeyFrame, size_t>), ref, normal, min_distance, max_distance)
# local_kfs = lLocalKeyFrames, fixed_kfs = lFixedCameras, mps = lLocalMapPoints (the caller selects them, :493-546).  Everything after that follows the
# reference: vertices (:552-622), edges per point in mObservations order (:626-700), optimize(5) / classification / optimize(10) (ba_solve_staged with
# LOCAL_BA_STAGES), vToErase monocular edges first, then stereo (:766-807), SetPose (:811-820), SetWorldPos + UpdateNormalAndDepth (:823-835).
def map_point_erase_observation(mp, kf, kfs_by_id):
    """MapPoint::EraseObservation (C/src/MapPoint.cc:192-217) + SetBadFlag (:255-269)"""
    if kf["id"] not in mp["obs"]:
        return
    idx = mp["obs"].pop(kf["id"])
    mp["nObs"] -= 2 if kf["ur"][idx] >= 0 else 1
    if mp["ref"] == kf["id"] and mp["obs"]:
        mp["ref"] = min(mp["obs"])
    if mp["nObs"] <= 2:
        mp["bad"] = True
        for kid, i in mp["obs"].items():
            if kid in kfs_by_id:
                kfs_by_id[kid]["mp"][i] = None
        mp["obs"] = {}


def map_point_update_normal_and_depth(mp, kfs_by_id, scale_factor):
    """MapPoint::UpdateNormalAndDepth (C/src/MapPoint.cc:424-472); mvScaleFactor as ORBextractor.cc:418-424 builds it"""
    if mp["bad"] or not mp["obs"] or mp["ref"] not in kfs_by_id:
        return
    ref = kfs_by_id[mp["ref"]]
    if mp["ref"] not in mp["obs"] or mp["obs"][mp["ref"]] >= len(ref["keys"]):
        return
    pos = np.asarray(mp["pos"], np.float32)

    def center(T):
        T = np.asarray(T, np.float32)
        return -(T[:3, :3].T @ T[:3, 3]).astype(np.float32)
    normal = np.zeros(3, np.float32); n = 0
    for kid in sorted(mp["obs"]):
        if kid not in kfs_by_id:
            continue
        v = (pos - center(kfs_by_id[kid]["T"])).astype(np.float32)
        normal = (normal + (v.astype(np.float64) * (1.0 / np.sqrt((v.astype(np.float64) ** 2).sum()))).astype(np.float32)).astype(np.float32); n += 1
    pc = (pos - center(ref["T"])).astype(np.float32)
    dist = np.float32(np.sqrt((pc.astype(np.float64) ** 2).sum()))
    sc = [np.float32(1.0)]
    for _ in range(1, ref["nlevels"]):
        sc.append(np.float32(sc[-1] * np.float32(scale_factor)))
    level = int(ref["keys"]["octave"][mp["obs"][mp["ref"]]])
    mp["max_distance"] = np.float32(dist * sc[level]); mp["min_distance"] = np.float32(mp["max_distance"] / sc[ref["nlevels"] - 1])
    mp["normal"] = (normal.astype(np.float64) * (1.0 / n)).astype(np.float32)


def local_bundle_adjustment(local_kfs, fixed_kfs, mps, scale_factor=1.2, apply_erase=True, stop=None):
    kfs = list(local_kfs) + list(fixed_kfs)
    by_id = {k["id"]: k for k in kfs}
    vidx = {k["id"]: i for i, k in enumerate(kfs)}
    poses = np.stack([np.asarray(k["T"], np.float32).reshape(16) for k in kfs])
    pose_fixed = np.array([1 if (i >= len(local_kfs) or k["id"] == 1 or k["fixed"] or k["bad"]) else 0 for i, k in enumerate(kfs)], np.uint8)
    intr = np.array([k["intr"] for k in kfs], np.float32)
    points = np.array([m["pos"] for m in mps], np.float32).reshape(-1, 3)
    point_fixed = np.array([1 if m["fixed"] else 0 for m in mps], np.uint8)
    E, who = [], []
    entry_bad = [bool(m["bad"]) for m in mps]                              # (lLocalMapPoints holds no bad point, :518)
    for j, m in enumerate(mps):
        if m["bad"]:
            continue
        for kid in sorted(m["obs"]):
            k = by_id.get(kid)
            if k is None or k["bad"]:
                continue
            f = m["obs"][kid]; kp = k["keys"][f]
            E.append((vidx[kid], j, kp["x"], kp["y"], k["ur"][f], k["inv_level_sigma2"][int(kp["octave"])]))     # e->setInformation(Eye * pKFi->mvInvLevelSigma2[kpUn.octave]) (:655, :684)
            who.append((kid, j))
    edges = np.zeros(len(E), EDGE_DTYPE)
    for i, e in enumerate(E):
        edges[i] = e
    if stop == "before":
        return dict(erase=[], edges=edges)
    r = ba_solve_staged(poses, pose_fixed, points, point_fixed, edges, 0, 0, 0, 0, 0, LOCAL_BA_STAGES, intr=intr, stop=stop)
    out = np.nonzero(r["outlier"])[0]
    mono = [i for i in out if edges["ur"][i] < 0]; stereo = [i for i in out if edges["ur"][i] >= 0]
    erase = [who[i] for i in mono + stereo]
    if apply_erase:
        for kid, j in erase:
            k = by_id[kid]; m = mps[j]
            if kid in m["obs"]:
                k["mp"][m["obs"][kid]] = None                              # pKFi->EraseMapPointMatch(pMPi)
            else:
                for f, q in enumerate(k["mp"]):                            # (the point went bad through an earlier erasure; its matches are gone already)
                    if q == m["id"]:
                        k["mp"][f] = None
            map_point_erase_observation(m, k, by_id)
    for i, k in enumerate(local_kfs):
        if not k["fixed"] and not k["bad"]:
            k["T"] = r["poses"][i].astype(np.float32).reshape(4, 4)
    for j, m in enumerate(mps):
        if not m["fixed"] and not entry_bad[j]:
            m["pos"] = r["points"][j].astype(np.float32)
    for m in mps:
        if not m["fixed"]:
            map_point_update_normal_and_depth(m, by_id, scale_factor)
    return dict(erase=[(vidx[kid], j) for kid, j in erase], edges=edges, poses=r["poses"], points=r["points"], outlier=r["outlier"])
