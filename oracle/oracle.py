"""ctypes binding of oracle/_build/liboracle.so — TEST INFRASTRUCTURE.

The oracle is the CPU restatement of the reference's scan-to-map registration path
(see oracle/sf_oracle.h for what each function follows, file:line).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; it is
the checker, never the thing measured or shipped.  PARITY UNPINNED upstream (no reference
fixtures exist); pinned here by oracle/_ref (geo_lib.hpp), scipy/numpy cross-checks and
analytic known answers.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_REF_PATH = os.path.join(_HERE, "_ref", "libsfref.so")


def build(force=False):
    """Compile liboracle.so (and _ref/libsfref.so when /root/reference exists)."""
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.isdir("/root/reference") and not os.path.exists(_REF_PATH)):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))


class Result(C.Structure):
    _fields_ = [("T", C.c_double * 16), ("error", C.c_double), ("fitness", C.c_double),
                ("iterations", C.c_int), ("converged", C.c_int), ("n_corr", C.c_int),
                ("n_research", C.c_int)]

    def as_dict(self):
        return dict(T=np.array(self.T, dtype=np.float64).reshape(4, 4), error=self.error,
                    fitness=self.fitness, iterations=self.iterations,
                    converged=bool(self.converged), n_corr=self.n_corr,
                    n_research=self.n_research)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_kdtree_f_build.restype = C.c_void_p
        _lib.orc_kdtree_d_build.restype = C.c_void_p
        _lib.orc_sfilter_new.restype = C.c_void_p
        _lib.orc_sfilter_zscore.restype = C.c_float
        _lib.orc_compass_to_yaw.restype = C.c_float
        _lib.orc_closest_altitude.restype = C.c_float
    return _lib


def ref_lib():
    """oracle/_ref/libsfref.so: the reference's own geo_lib.hpp compiled here; None if absent."""
    global _ref
    if _ref is None and os.path.exists(_REF_PATH):
        _ref = C.CDLL(_REF_PATH)
    return _ref


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ---------------------------------------------------------------- NN
class KdTreeF:
    def __init__(self, xyz, leaf=15):
        self.xyz = _f32(xyz).reshape(-1, 3)
        self.h = C.c_void_p(lib().orc_kdtree_f_build(_p(self.xyz), C.c_int(len(self.xyz)), C.c_int(leaf)))

    def nn(self, q):
        q = _f32(q).reshape(-1, 3)
        idx = np.empty(len(q), np.int32)
        d2 = np.empty(len(q), np.float32)
        lib().orc_kdtree_f_nn(self.h, _p(q), C.c_int(len(q)), _p(idx), _p(d2))
        return idx, d2

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_kdtree_f_free(self.h)
            self.h = None


class KdTreeD:
    def __init__(self, xyz, leaf=15):
        self.xyz = _f64(xyz).reshape(-1, 3)
        self.h = C.c_void_p(lib().orc_kdtree_d_build(_p(self.xyz), C.c_int(len(self.xyz)), C.c_int(leaf)))

    def nn(self, q):
        q = _f64(q).reshape(-1, 3)
        idx = np.empty(len(q), np.int32)
        d2 = np.empty(len(q), np.float64)
        lib().orc_kdtree_d_nn(self.h, _p(q), C.c_int(len(q)), _p(idx), _p(d2))
        return idx, d2

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_kdtree_d_free(self.h)
            self.h = None


def bruteforce_nn(tgt, q):
    tgt = _f32(tgt).reshape(-1, 3)
    q = _f32(q).reshape(-1, 3)
    idx = np.empty(len(q), np.int32)
    d2 = np.empty(len(q), np.float32)
    lib().orc_bruteforce_nn_f(_p(tgt), C.c_int(len(tgt)), _p(q), C.c_int(len(q)), _p(idx), _p(d2))
    return idx, d2


# ---------------------------------------------------------------- crops
def uniform_subsample(xyz, step):
    xyz = _f32(xyz).reshape(-1, 3)
    out = np.empty_like(xyz)
    k = lib().orc_uniform_subsample(_p(xyz), C.c_int(len(xyz)), C.c_int(step), _p(out))
    return out[:k].copy()


def crop_radius(xyz, center, radius):
    xyz = _f32(xyz).reshape(-1, 3)
    out = np.empty_like(xyz)
    idx = np.empty(len(xyz), np.int32)
    c = _f32(center)
    k = lib().orc_crop_radius(_p(xyz), C.c_int(len(xyz)), _p(c), C.c_double(radius), _p(out), _p(idx))
    return out[:k].copy(), idx[:k].copy()


def remove_floor(xyz):
    xyz = _f32(xyz).reshape(-1, 3)
    out = np.empty_like(xyz)
    k = lib().orc_remove_floor(_p(xyz), C.c_int(len(xyz)), _p(out))
    return out[:k].copy()


def crop_aabb(xyz, lo, hi):
    xyz = _f32(xyz).reshape(-1, 3)
    out = np.empty_like(xyz)
    idx = np.empty(len(xyz), np.int32)
    lo, hi = _f64(lo), _f64(hi)
    k = lib().orc_crop_aabb(_p(xyz), C.c_int(len(xyz)), _p(lo), _p(hi), _p(out), _p(idx))
    return out[:k].copy(), idx[:k].copy()


def crop_obb(xyz, center, R, extent):
    xyz = _f32(xyz).reshape(-1, 3)
    out = np.empty_like(xyz)
    idx = np.empty(len(xyz), np.int32)
    c, R, e = _f64(center), _f64(R).reshape(3, 3), _f64(extent)
    k = lib().orc_crop_obb(_p(xyz), C.c_int(len(xyz)), _p(c), _p(R), _p(e), _p(out), _p(idx))
    return out[:k].copy(), idx[:k].copy()


# ---------------------------------------------------------------- voxel grids
def voxel_pcl(xyz, leaf=0.1):
    """-> (centroids f32 [k,3], per-point voxel index int32 [n], per-voxel index int32 [k], status)"""
    xyz = _f32(xyz).reshape(-1, 3)
    out = np.empty_like(xyz)
    vidx = np.empty(len(xyz), np.int32)
    ovox = np.empty(len(xyz), np.int32)
    k = lib().orc_voxel_pcl(_p(xyz), C.c_int(len(xyz)), C.c_float(leaf), _p(out), _p(vidx), _p(ovox))
    if k < 0:
        return out.copy(), vidx, ovox[:0].copy(), k
    return out[:k].copy(), vidx, ovox[:k].copy(), 0


def voxel_pcl64(xyz, leaf=0.1):
    """int64-index variant of voxel_pcl (never overflows) -> (centroids, per-point index int64, per-voxel index int64)"""
    xyz = _f32(xyz).reshape(-1, 3)
    out = np.empty_like(xyz)
    vidx = np.empty(len(xyz), np.int64)
    ovox = np.empty(len(xyz), np.int64)
    k = lib().orc_voxel_pcl64(_p(xyz), C.c_int(len(xyz)), C.c_float(leaf), _p(out), _p(vidx), _p(ovox))
    return out[:k].copy(), vidx, ovox[:k].copy()


def voxel_o3d(xyz, voxel=0.1):
    """-> (means f64 [k,3], per-point ijk int32 [n,3], per-voxel ijk [k,3], status)"""
    xyz = _f64(xyz).reshape(-1, 3)
    out = np.empty_like(xyz)
    ijk = np.empty((len(xyz), 3), np.int32)
    oijk = np.empty((len(xyz), 3), np.int32)
    k = lib().orc_voxel_o3d(_p(xyz), C.c_int(len(xyz)), C.c_double(voxel), _p(out), _p(ijk), _p(oijk))
    if k < 0:
        return out[:0].copy(), ijk, oijk[:0].copy(), k
    return out[:k].copy(), ijk, oijk[:k].copy(), 0


# ---------------------------------------------------------------- linear algebra
def svd3(A, dtype=np.float64):
    A = np.ascontiguousarray(A, dtype=dtype).reshape(3, 3)
    U = np.empty((3, 3), dtype)
    S = np.empty(3, dtype)
    V = np.empty((3, 3), dtype)
    fn = lib().orc_svd3_d if dtype == np.float64 else lib().orc_svd3_f
    fn(_p(A), _p(U), _p(S), _p(V))
    return U, S, V


def kabsch(src, tgt, precise=True):
    src, tgt = _f32(src).reshape(-1, 3), _f32(tgt).reshape(-1, 3)
    T = np.empty(16, np.float64)
    lib().orc_kabsch(_p(src), _p(tgt), C.c_int(len(src)), C.c_int(int(precise)), _p(T))
    return T.reshape(4, 4)


# ---------------------------------------------------------------- ICP
def icp_ref_cpp(src, tgt, init=None, max_corr_dist=0.5, num_iters=10, accept_err=0.05,
                eps=1e-5, precise=False):
    src, tgt = _f32(src).reshape(-1, 3), _f32(tgt).reshape(-1, 3)
    init = _f32(np.eye(4) if init is None else init).reshape(16)
    r = Result()
    lib().orc_icp_ref_cpp(_p(src), C.c_int(len(src)), _p(tgt), C.c_int(len(tgt)), _p(init),
                          C.c_float(max_corr_dist), C.c_int(num_iters), C.c_float(accept_err),
                          C.c_float(eps), C.c_int(int(precise)), C.byref(r))
    return r.as_dict()


def icp_o3d_p2p(src, tgt, init=None, max_dist=0.5, max_iter=30):
    src, tgt = _f32(src).reshape(-1, 3), _f32(tgt).reshape(-1, 3)
    init = _f64(np.eye(4) if init is None else init).reshape(16)
    r = Result()
    lib().orc_icp_o3d_p2p(_p(src), C.c_int(len(src)), _p(tgt), C.c_int(len(tgt)), _p(init),
                          C.c_double(max_dist), C.c_int(max_iter), C.byref(r))
    return r.as_dict()


def icp_p2plane(src, tgt, normals, init=None, max_dist=0.5, num_iters=20):
    src, tgt, normals = _f32(src).reshape(-1, 3), _f32(tgt).reshape(-1, 3), _f32(normals).reshape(-1, 3)
    init = _f64(np.eye(4) if init is None else init).reshape(16)
    r = Result()
    lib().orc_icp_p2plane(_p(src), C.c_int(len(src)), _p(tgt), _p(normals), C.c_int(len(tgt)),
                          _p(init), C.c_double(max_dist), C.c_int(num_iters), C.byref(r))
    return r.as_dict()


def normals_radius(xyz, radius):
    xyz = _f32(xyz).reshape(-1, 3)
    nrm = np.empty_like(xyz)
    cnt = np.empty(len(xyz), np.int32)
    lib().orc_normals_radius(_p(xyz), C.c_int(len(xyz)), C.c_double(radius), _p(nrm), _p(cnt))
    return nrm, cnt


def normals_radius_cov(xyz, radius):
    xyz = _f32(xyz).reshape(-1, 3)
    nrm = np.empty_like(xyz)
    cnt = np.empty(len(xyz), np.int32)
    cov = np.empty((len(xyz), 6), np.float64)
    lib().orc_normals_radius_cov(_p(xyz), C.c_int(len(xyz)), C.c_double(radius), _p(nrm), _p(cnt), _p(cov))
    return nrm, cnt, cov


# ---------------------------------------------------------------- fusion
def ll_to_utm(lat, lon):
    n, e = C.c_double(), C.c_double()
    lib().orc_ll_to_utm(C.c_double(lat), C.c_double(lon), C.byref(n), C.byref(e))
    return n.value, e.value


def ref_ll_to_utm(lat, lon):
    n, e = C.c_double(), C.c_double()
    ref_lib().sfref_ll_to_utm(C.c_double(lat), C.c_double(lon), C.byref(n), C.byref(e))
    return n.value, e.value


def utm_from_latlon(lat, lon):
    e, n = C.c_double(), C.c_double()
    lib().orc_utm_from_latlon(C.c_double(lat), C.c_double(lon), C.byref(e), C.byref(n))
    return e.value, n.value


def quat_to_pose(q_wxyz, t):
    q, t = _f64(q_wxyz), _f64(t)
    T = np.empty(16, np.float32)
    lib().orc_quat_to_pose(_p(q), _p(t), _p(T))
    return T.reshape(4, 4)


def mat4f_inverse(A):
    A = _f32(A).reshape(16)
    out = np.empty(16, np.float32)
    lib().orc_mat4f_inverse(_p(A), _p(out))
    return out.reshape(4, 4)


def mat4f_mul(A, B):
    A, B = _f32(A).reshape(16), _f32(B).reshape(16)
    out = np.empty(16, np.float32)
    lib().orc_mat4f_mul(_p(A), _p(B), _p(out))
    return out.reshape(4, 4)


def odom_prediction(map_T_sensor, odom_T_prev, odom_T_cur):
    a, b, c = (_f32(x).reshape(16) for x in (map_T_sensor, odom_T_prev, odom_T_cur))
    out = np.empty(16, np.float32)
    lib().orc_odom_prediction(_p(a), _p(b), _p(c), _p(out))
    return out.reshape(4, 4)


def compass_to_yaw(deg):
    return float(lib().orc_compass_to_yaw(C.c_double(deg)))


def closest_altitude(table, lat, lon):
    table = _f64(table).reshape(-1, 3)
    return float(lib().orc_closest_altitude(_p(table), C.c_int(len(table)), C.c_double(lat), C.c_double(lon)))


def gps_pose(map_T_global, yaw, lat, lon, table_alt):
    M = _f64(map_T_global).reshape(16)
    out = np.empty(16, np.float32)
    lib().orc_gps_pose(_p(M), C.c_float(yaw), C.c_double(lat), C.c_double(lon), C.c_float(table_alt), _p(out))
    return out.reshape(4, 4)


def pose_gains(gps_cov, odom_cov, fixed=False):
    g, o = _f64(gps_cov).reshape(9), _f64(odom_cov).reshape(36)
    a, b = C.c_float(), C.c_float()
    lib().orc_pose_gains(_p(g), _p(o), C.c_int(int(fixed)), C.byref(a), C.byref(b))
    return a.value, b.value  # (odom_gain, gps_gain)


def blend(g_odom, T_odom, g_gps, T_gps):
    a, b = _f32(T_odom).reshape(16), _f32(T_gps).reshape(16)
    out = np.empty(16, np.float32)
    lib().orc_blend(C.c_float(g_odom), _p(a), C.c_float(g_gps), _p(b), _p(out))
    return out.reshape(4, 4)


def map_T_global(latlonalt, yaw):
    l, y = _f64(latlonalt).reshape(-1, 3), _f32(yaw)
    out = np.empty(16, np.float64)
    lib().orc_map_T_global(_p(l), _p(y), C.c_int(len(l)), _p(out))
    return out.reshape(4, 4)


class StochasticFilter:
    def __init__(self, queue_size=10, z_threshold=1.0):
        self.q = queue_size
        self.h = C.c_void_p(lib().orc_sfilter_new(C.c_int(queue_size), C.c_float(z_threshold)))

    def weights(self):
        w = np.empty(self.q, np.float32)
        lib().orc_sfilter_weights(self.h, _p(w))
        return w

    def add_pose(self, pose):
        p = _f32(pose).reshape(16)
        lib().orc_sfilter_add_pose(self.h, _p(p))

    def zscore(self, prev, cur):
        a, b = _f32(prev).reshape(16), _f32(cur).reshape(16)
        return float(lib().orc_sfilter_zscore(self.h, _p(a), _p(b)))

    def apply(self, prev, cur):
        a, b = _f32(prev).reshape(16), _f32(cur).reshape(16)
        out = np.empty(16, np.float32)
        lib().orc_sfilter_apply(self.h, _p(a), _p(b), _p(out))
        return out.reshape(4, 4)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_sfilter_free(self.h)
            self.h = None


def bf_sequence(rng, step):
    seq = np.empty(4096, np.float32)
    k = lib().orc_bf_sequence(C.c_float(rng), C.c_float(step), _p(seq), C.c_int(4096))
    return seq[:k].copy()


class BfParams(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("x_step", "y_step", "z_step", "yaw_step", "x_range", "y_range", "z_range",
                                         "yaw_range", "threshold")]


def bf_align(src, tgt, prev_T, x_step=0.1, y_step=0.1, z_step=0.05, yaw_step=np.pi / 18.0, x_range=1.5, y_range=1.5,
             z_range=0.1, yaw_range=np.pi / 6.0, threshold=0.1):
    """-> dict(found, best_T, best_score, index, n_candidates, scores, prev_T_after)"""
    src, tgt = _f32(src).reshape(-1, 3), _f32(tgt).reshape(-1, 3)
    prev = _f32(prev_T).reshape(16).copy()
    prm = BfParams(x_step, y_step, z_step, yaw_step, x_range, y_range, z_range, yaw_range, threshold)
    best = np.empty(16, np.float32)
    score, idx, ncand = C.c_float(), C.c_int(), C.c_int()
    scores = np.full(1 << 16, np.nan, np.float32)
    found = lib().orc_bf_align(_p(src), C.c_int(len(src)), _p(tgt), C.c_int(len(tgt)), _p(prev), C.byref(prm), _p(best),
                               C.byref(score), C.byref(idx), C.byref(ncand), _p(scores))
    return dict(found=bool(found), best_T=best.reshape(4, 4), best_score=score.value, index=idx.value,
                n_candidates=ncand.value, scores=scores[:ncand.value].copy(), prev_T_after=prev.reshape(4, 4))


# ------------------------------------------------------------------ Python twin: optimize_global_map_pose
def map_builder_py(odom_positions, gps_imu_rows, max_poses=50, max_translation=0.5):
    """map_T_global as the reference's Python MapBuilder computes it
    (localization_python/localization_python/optimize_global_map_pose.py:21-32 leading poses under 0.5 m, :34-47 rows ->
    pose[3:7] and UTM, :66-99 mean angles / mean position -> from_euler('xyz') -> inverse), restated with scipy's own
    Rotation (the reference's dependency, importable here) and the oracle's C restatement of utm.from_latlon.
    Raises ValueError like the reference when a row has anything but 6 columns' worth of angles."""
    from scipy.spatial.transform import Rotation
    odom = np.atleast_2d(np.asarray(odom_positions, dtype=np.float64))
    rows = np.atleast_2d(np.asarray(gps_imu_rows, dtype=np.float64))
    count = 0
    for p in odom:
        if np.linalg.norm(p) < max_translation:
            count += 1
        else:
            break
    rpy = [r[3:7] for r in rows]
    t = [np.array([*utm_from_latlon(r[0], r[1])[:2], r[2]]) for r in rows]
    n = min(count, len(rpy), max_poses)
    G = np.eye(4)
    G[:3, :3] = Rotation.from_euler('xyz', np.mean(rpy[:n], axis=0)).as_matrix()
    G[:3, 3] = np.mean(t[:n], axis=0)
    return np.linalg.inv(G), n
