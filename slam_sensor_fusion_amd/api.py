"""ctypes binding of libslamfusion.so (include/slamfusion.h) — the product's host side.

There is no CPU fallback: if the HIP library is missing or no device is present every
entry point raises.  Nothing here imports or calls anything under oracle/.
"""
import atexit
import ctypes as C
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libslamfusion.so")

SF_ICP_REF_CPP, SF_ICP_O3D_P2P, SF_ICP_P2PLANE = 0, 1, 2
SF_VOXEL_PCL, SF_VOXEL_O3D, SF_VOXEL_PCL64 = 0, 1, 2
SF_FLAG_VOXEL_OVERFLOW = 1
SF_ICP_FLAG_FEW_CORR, SF_ICP_FLAG_SINGULAR, SF_ICP_FLAG_SHARD_STALE, SF_ICP_FLAG_BARRIER_TIMEOUT = 1, 2, 4, 8
PROF_NN, PROF_REDUCE, PROF_COLLECTIVE, PROF_SOLVE, PROF_SHARD_BUILD = 0, 1, 2, 3, 4
SF_ERR_COMM = -6
MODES = {"ref_cpp": SF_ICP_REF_CPP, "o3d_p2p": SF_ICP_O3D_P2P, "p2plane": SF_ICP_P2PLANE}


class SlamFusionError(RuntimeError):
    pass


class IcpResult(C.Structure):
    """sf_icp_result (ICPResult of icp_point_to_point.h:28-39 plus diagnostics)."""
    _fields_ = [("T", C.c_float * 16), ("error", C.c_float), ("iterations", C.c_int32),
                ("converged", C.c_int32), ("n_corr", C.c_int32), ("n_research", C.c_int32),
                ("flags", C.c_int32), ("fitness", C.c_double), ("rmse", C.c_double),
                ("T64", C.c_double * 16)]

    def as_dict(self):
        return dict(T=np.array(self.T, dtype=np.float32).reshape(4, 4),
                    T64=np.array(self.T64, dtype=np.float64).reshape(4, 4),
                    error=float(self.error), iterations=int(self.iterations),
                    converged=bool(self.converged), n_corr=int(self.n_corr),
                    n_research=int(self.n_research), flags=int(self.flags),
                    fitness=float(self.fitness), rmse=float(self.rmse))


_lib = None
_live = weakref.WeakSet()   # every wrapper object, closed in dependency order at exit


def _close_all():
    for kind in ("Node", "Comm", "Icp", "BruteForceAlignment", "Map", "Cloud", "Context"):
        for obj in [o for o in list(_live) if type(o).__name__ == kind]:
            try:
                obj.close()
            except Exception:
                pass


atexit.register(_close_all)


def load_library():
    """Load libslamfusion.so; fails loudly (no fallback) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SlamFusionError(
            "libslamfusion.so not found at %s — build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
            "There is no CPU fallback." % LIB_PATH)
    # This image carries two HIP runtimes (ROCm 7.2 under /opt/rocm, and the 7.0 one bundled
    # in the torch wheel).  A process must use exactly one: import torch FIRST so that the
    # library's libamdhip64.so.7 dependency resolves to the runtime torch already loaded
    # (the other order leaves torch with "No HIP GPUs are available").  torch is plumbing
    # only (streams, RCCL); nothing in the data path goes through it.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    lib.sf_last_error.restype = C.c_char_p
    lib.sf_ctx_stream.restype = C.c_void_p
    lib.sf_cloud_device_ptr.restype = C.c_void_p
    lib.sf_icp_exchange_ptr.restype = C.c_void_p
    lib.sf_sfilter_create.restype = C.c_void_p
    lib.sf_node_icp.restype = C.c_void_p
    lib.sf_node_bf.restype = C.c_void_p
    lib.sf_fusion_compass_to_yaw.restype = C.c_float
    lib.sf_fusion_closest_altitude.restype = C.c_float
    lib.sf_sfilter_pose_zscore.restype = C.c_float
    _lib = lib
    return lib


def voxel_merge_min_points(n=-1):
    """Map size from which Cloud.voxel_merge merges instead of re-filtering (returns the previous value; n < 0 only reads it)."""
    return int(load_library().sf_cloud_voxel_merge_min_points(C.c_int64(n)))


def kernel_source_hash():
    """sha256[:16] of the sources the dominant kernel (k_nn_red / k_ref_nn) is compiled from.  profiles/*_traffic.json
    carry the hash of the build their counters were taken on; bench.py only quotes them while it still matches."""
    import hashlib
    h = hashlib.sha256()
    for name in ("sf_icp.hip", "sf_nn.hpp", "sf_tile.hpp", "sf_order.hpp", "sf_common.hpp"):
        with open(os.path.join(_HERE, "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _check(rc):
    if rc != 0:
        raise SlamFusionError("libslamfusion error %d: %s" % (rc, load_library().sf_last_error().decode()))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _pc2_layout(msg):
    """(buffer, width, height, point_step, row_step, (off_x, off_y, off_z), datatype, is_bigendian) of a PointCloud2-like message"""
    offs, types = {"x": 0, "y": 4, "z": 8}, {"x": 7, "y": 7, "z": 7}
    for f in getattr(msg, "fields", []) or []:
        if f.name in offs:
            offs[f.name] = int(f.offset)
            types[f.name] = int(getattr(f, "datatype", 7))
    if len(set(types.values())) != 1:
        raise SlamFusionError("PointCloud2 x/y/z fields have different datatypes: %r" % (types,))
    buf = np.frombuffer(msg.data, dtype=np.uint8)
    return (buf, int(msg.width), int(msg.height), int(msg.point_step), int(getattr(msg, "row_step", 0) or 0), (offs["x"], offs["y"], offs["z"]), types["x"],
            int(bool(getattr(msg, "is_bigendian", False))))


class Context:
    """sf_ctx: one device + one HIP stream (optionally a caller-owned stream)."""

    def __init__(self, device=0, stream=None):
        self.lib = load_library()
        self.h = C.c_void_p()
        _check(self.lib.sf_ctx_create(C.c_int(device), C.c_void_p(stream), C.byref(self.h)))
        _live.add(self)

    def synchronize(self):
        _check(self.lib.sf_ctx_synchronize(self.h))

    @property
    def stream(self):
        return self.lib.sf_ctx_stream(self.h)

    def device_name(self):
        buf = C.create_string_buffer(256)
        _check(self.lib.sf_ctx_device_name(self.h, buf, C.c_int(256)))
        return buf.value.decode()

    def close(self):
        if self.h:
            self.lib.sf_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Cloud:
    """sf_cloud: device point set with the reference's preprocessing operations."""

    def __init__(self, ctx, xyz=None):
        self.ctx, self.lib = ctx, ctx.lib
        self.h = C.c_void_p()
        _check(self.lib.sf_cloud_create(ctx.h, C.byref(self.h)))
        _live.add(self)
        if xyz is not None:
            self.upload(xyz)

    def upload(self, xyz):
        xyz = _f32(xyz).reshape(-1, 3)
        _check(self.lib.sf_cloud_upload(self.h, _p(xyz), C.c_int64(len(xyz))))
        return self

    def upload_async_ptr(self, host_ptr, n):
        """n x 3 float32 at a raw host address, enqueue only (pinned memory: asynchronous, stream-ordered)"""
        _check(self.lib.sf_cloud_upload_async(self.h, C.c_void_p(host_ptr), C.c_int64(n)))
        return self

    def device_ptr(self):
        return self.lib.sf_cloud_device_ptr(self.h)

    def from_pointcloud2(self, msg):
        """PointCloud2-like message (width, height, point_step, row_step, is_bigendian, fields or x/y/z float32 at
        0/4/8, data).  The buffer length, the x/y/z datatype (FLOAT32 = 7 or FLOAT64 = 8, all three alike) and the
        endianness are checked here and again behind the C ABI."""
        buf, width, height, point_step, row_step, offs, dtype, big = _pc2_layout(msg)
        _check(self.lib.sf_cloud_from_pointcloud2_msg(self.h, _p(buf), C.c_int64(buf.size), C.c_int64(width), C.c_int64(height), C.c_int(point_step), C.c_int64(row_step),
                                                      C.c_int(offs[0]), C.c_int(offs[1]), C.c_int(offs[2]), C.c_int(dtype), C.c_int(big)))
        return self

    def load_pcd(self, path):
        _check(self.lib.sf_cloud_load_pcd(self.h, str(path).encode()))
        return self

    def save_pcd(self, path):
        _check(self.lib.sf_cloud_save_pcd(self.h, str(path).encode()))

    def from_device(self, ptr, n):
        _check(self.lib.sf_cloud_from_device(self.h, C.c_void_p(ptr), C.c_int64(n)))
        return self

    def __len__(self):
        n = C.c_int64()
        _check(self.lib.sf_cloud_size(self.h, C.byref(n)))
        return n.value

    def download(self):
        n = len(self)
        out = np.empty((n, 3), np.float32)
        _check(self.lib.sf_cloud_download(self.h, _p(out), C.c_int64(n), None))
        return out

    def copy(self):
        c = Cloud(self.ctx)
        _check(self.lib.sf_cloud_copy(c.h, self.h))
        return c

    def copy_from(self, other):
        """device-to-device copy into this cloud's own (persistent) buffer"""
        _check(self.lib.sf_cloud_copy(self.h, other.h))
        return self

    def subsample(self, step):                       # applyUniformSubsample
        _check(self.lib.sf_cloud_subsample(self.h, C.c_int(step)))
        return self

    def crop_radius(self, center, radius, sorted=False):  # cropPointCloudThroughRadius
        c = _f32(center)
        _check(self.lib.sf_cloud_crop_radius(self.h, _p(c), C.c_double(radius), C.c_int(int(sorted))))
        return self

    def remove_floor(self):                          # removeFloor
        _check(self.lib.sf_cloud_remove_floor(self.h))
        return self

    def crop_aabb(self, lo, hi):                     # readFilterPtcRegionPoints
        lo, hi = _f64(lo), _f64(hi)
        _check(self.lib.sf_cloud_crop_aabb(self.h, _p(lo), _p(hi)))
        return self

    def crop_obb(self, center, R, extent):           # OrientedBoundingBox crop
        c, R, e = _f64(center), _f64(R).reshape(3, 3), _f64(extent)
        _check(self.lib.sf_cloud_crop_obb(self.h, _p(c), _p(R), _p(e)))
        return self

    def append(self, other):                         # *map_cloud += *cloud
        _check(self.lib.sf_cloud_append(self.h, other.h))
        return self

    def transform(self, T):                          # applyTransformation
        T = _f32(T).reshape(16)
        _check(self.lib.sf_cloud_transform(self.h, _p(T)))
        return self

    def last_indices(self):
        n = C.c_int64()
        cap = 1 << 20
        while True:
            out = np.empty(cap, np.int32)
            rc = self.lib.sf_cloud_last_indices(self.h, _p(out), C.c_int64(cap), C.byref(n))
            if rc == 0:
                return out[:n.value].copy()
            if n.value > cap:
                cap = n.value
                continue
            _check(rc)

    def voxel_downsample(self, leaf=0.1, flavour="pcl"):
        flags = C.c_int(0)
        fl = {"pcl": SF_VOXEL_PCL, "o3d": SF_VOXEL_O3D, "pcl64": SF_VOXEL_PCL64}[flavour]
        _check(self.lib.sf_cloud_voxel_downsample(self.h, C.c_double(leaf), C.c_int(fl), C.byref(flags)))
        return flags.value

    def voxel_merge(self, pending, leaf=0.1):
        """append(pending) + voxel_downsample(leaf, "pcl") as a merge when this cloud is already voxel-filtered -> (status flags, merged?)"""
        st, mg = C.c_int(), C.c_int()
        _check(self.lib.sf_cloud_voxel_merge(self.h, pending.h, C.c_double(leaf), C.byref(st), C.byref(mg)))
        return st.value, bool(mg.value)

    def _i32(self, fn):
        n = C.c_int64()
        fn(self.h, None, C.c_int64(0), C.byref(n))
        out = np.empty(max(n.value, 1), np.int32)
        _check(fn(self.h, _p(out), C.c_int64(len(out)), C.byref(n)))
        return out[:n.value]

    def _i64(self, fn):
        n = C.c_int64()
        fn(self.h, None, C.c_int64(0), C.byref(n))
        out = np.empty(max(n.value, 1), np.int64)
        _check(fn(self.h, _p(out), C.c_int64(len(out)), C.byref(n)))
        return out[:n.value]

    def voxel_point_ids64(self):                     # after voxel_downsample(flavour="pcl64")
        return self._i64(self.lib.sf_cloud_voxel_point_ids64)

    def voxel_out_ids64(self):
        return self._i64(self.lib.sf_cloud_voxel_out_ids64)

    def voxel_point_ids(self):
        return self._i32(self.lib.sf_cloud_voxel_point_ids)

    def voxel_out_ids(self):
        return self._i32(self.lib.sf_cloud_voxel_out_ids)

    def voxel_out_means_f64(self):
        n = C.c_int64()
        self.lib.sf_cloud_voxel_out_means_f64(self.h, None, C.c_int64(0), C.byref(n))
        out = np.empty((max(n.value, 1), 3), np.float64)
        _check(self.lib.sf_cloud_voxel_out_means_f64(self.h, _p(out), C.c_int64(len(out)), C.byref(n)))
        return out[:n.value]

    def close(self):
        if self.h:
            self.lib.sf_cloud_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Map:
    """sf_map: whole-map uniform-grid NN index (replaces the per-crop FLANN kd-tree)."""

    def __init__(self, ctx, cloud=None, cell=0.0):
        self.ctx, self.lib = ctx, ctx.lib
        self.h = C.c_void_p()
        _check(self.lib.sf_map_create(ctx.h, C.byref(self.h)))
        _live.add(self)
        if cloud is not None:
            self.build(cloud, cell)

    def build(self, cloud, cell=0.0):
        if not isinstance(cloud, Cloud):
            cloud = Cloud(self.ctx, cloud)
        _check(self.lib.sf_map_build(self.h, cloud.h, C.c_float(cell)))
        return self

    def set_origin_lattice(self, cells):
        """sf_map_set_origin_lattice: the grid origin snapped down to a multiple of `cells` cells (0 = off), from the next build on."""
        _check(self.lib.sf_map_set_origin_lattice(self.h, C.c_int(int(cells))))
        return self

    def patch(self, cloud):
        """sf_map_patch: the index after the last `cloud.voxel_merge(...)`, merged from the old one when it can be; -> True if it was."""
        done = C.c_int(0)
        _check(self.lib.sf_map_patch(self.h, cloud.h, C.byref(done)))
        self.last_patch = done.value                          # 1 merged, SF_PATCH_* (<= 0): built, and why (include/slamfusion.h)
        return done.value > 0

    def index(self):
        """The index as it lies on the device (parity tests): dict(pts4 [n, 4] float32 with the point id bit-cast into column 3,
        cell_start [cells + 1] uint32, org, inv_h, gap_eps)."""
        n, nc, org, inv_h, eps = C.c_int64(), C.c_int64(), (C.c_float * 3)(), C.c_float(), C.c_float()
        _check(self.lib.sf_map_index_info(self.h, C.byref(n), C.byref(nc), org, C.byref(inv_h), C.byref(eps)))
        pts4 = np.empty((max(n.value, 1), 4), np.float32)
        cs = np.empty(nc.value + 1, np.uint32)
        _check(self.lib.sf_map_download_index(self.h, pts4.ctypes.data_as(C.c_void_p), C.c_int64(len(pts4)), cs.ctypes.data_as(C.c_void_p), C.c_int64(len(cs))))
        return dict(pts4=pts4[:n.value], cell_start=cs, org=np.array(list(org), np.float32), inv_h=inv_h.value, gap_eps=eps.value)

    def __len__(self):
        n = C.c_int64()
        _check(self.lib.sf_map_size(self.h, C.byref(n)))
        return n.value

    def cell_size(self):
        cell = C.c_float()
        dims = (C.c_int32 * 3)()
        _check(self.lib.sf_map_cell_size(self.h, C.byref(cell), dims))
        return cell.value, tuple(dims)

    def window_none(self):
        _check(self.lib.sf_map_window_none(self.h))

    def window_sphere(self, center, radius):
        c = _f32(center)
        _check(self.lib.sf_map_window_sphere(self.h, _p(c), C.c_double(radius)))

    def window_obb(self, center, R, extent):
        c, R, e = _f64(center), _f64(R).reshape(3, 3), _f64(extent)
        _check(self.lib.sf_map_window_obb(self.h, _p(c), _p(R), _p(e)))

    def window_count(self):
        n = C.c_int64()
        _check(self.lib.sf_map_window_count(self.h, C.byref(n)))
        return n.value

    def estimate_normals(self, radius, covariance=False):
        _check(self.lib.sf_map_estimate_normals_cov(self.h, C.c_float(radius), C.c_int(int(covariance))))

    def download_covariances(self):
        """[n, 6] float64: xx xy xz yy yz zz of each point's neighbourhood (original point order)."""
        n = len(self)
        cov = np.empty((n, 6), np.float64)
        _check(self.lib.sf_map_download_covariances(self.h, _p(cov), C.c_int64(n), None))
        return cov

    def set_normals(self, normals):
        nrm = _f32(normals).reshape(-1, 3)
        _check(self.lib.sf_map_set_normals(self.h, _p(nrm), C.c_int64(len(nrm))))

    def download_normals(self):
        n = len(self)
        nrm = np.empty((n, 3), np.float32)
        cnt = np.empty(n, np.int32)
        _check(self.lib.sf_map_download_normals(self.h, _p(nrm), _p(cnt), C.c_int64(n), None))
        return nrm, cnt

    def nn(self, queries, max_d2=np.inf):
        q = _f32(queries).reshape(-1, 3)
        idx = np.empty(len(q), np.int32)
        d2 = np.empty(len(q), np.float32)
        _check(self.lib.sf_map_nn(self.h, _p(q), C.c_int64(len(q)), C.c_float(min(max_d2, 3.0e38)), _p(idx), _p(d2)))
        return idx, d2

    def close(self):
        if self.h:
            self.lib.sf_map_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Icp:
    """sf_icp: mirrors ICPPointToPoint (icp_point_to_point.h:41-136) setter for setter."""

    def __init__(self, ctx, max_correspondence_dist=0.5, num_iterations=10,
                 acceptable_mean_error=0.05, transformation_epsilon=1e-5):
        self.ctx, self.lib = ctx, ctx.lib
        self.h = C.c_void_p()
        self._map = None
        self.batch = 1
        _check(self.lib.sf_icp_create(ctx.h, C.c_float(max_correspondence_dist), C.c_int(num_iterations),
                                      C.c_float(acceptable_mean_error), C.c_float(transformation_epsilon),
                                      C.byref(self.h)))
        _live.add(self)

    def set_max_correspondence_dist(self, v):
        _check(self.lib.sf_icp_set_max_correspondence_dist(self.h, C.c_float(v)))

    def set_num_iterations(self, v):
        _check(self.lib.sf_icp_set_num_iterations(self.h, C.c_int(v)))

    def set_transformation_epsilon(self, v):
        _check(self.lib.sf_icp_set_transformation_epsilon(self.h, C.c_float(v)))

    def set_acceptable_mean_error(self, v):
        _check(self.lib.sf_icp_set_acceptable_mean_error(self.h, C.c_float(v)))

    def set_debug_mode(self, on):
        _check(self.lib.sf_icp_set_debug_mode(self.h, C.c_int(int(on))))

    def set_initial_transformation(self, T):
        T = np.asarray(T)
        if T.dtype == np.float32:
            T = _f32(T).reshape(16)
            _check(self.lib.sf_icp_set_initial_transformation(self.h, _p(T)))
        else:
            T = _f64(T).reshape(16)
            _check(self.lib.sf_icp_set_initial_transformation_f64(self.h, _p(T)))

    def set_source(self, xyz):
        if isinstance(xyz, Cloud):
            _check(self.lib.sf_icp_set_source_cloud(self.h, xyz.h))
        else:
            xyz = _f32(xyz).reshape(-1, 3)
            _check(self.lib.sf_icp_set_source(self.h, _p(xyz), C.c_int64(len(xyz))))
        self.batch = 1

    def set_source_batch(self, xyz):
        xyz = _f32(xyz)
        assert xyz.ndim == 3 and xyz.shape[2] == 3
        _check(self.lib.sf_icp_set_source_batch(self.h, _p(xyz), C.c_int64(xyz.shape[1]), C.c_int(xyz.shape[0])))
        self.batch = xyz.shape[0]

    def set_source_batch_host_ptr(self, ptr, n_per_scan, batch):
        """batch x n_per_scan x 3 float32 at a raw HOST address (e.g. pinned memory: the copy is then asynchronous and
        stream-ordered -- the buffer must stay untouched until the stream has passed it)"""
        _check(self.lib.sf_icp_set_source_batch(self.h, C.c_void_p(ptr), C.c_int64(n_per_scan), C.c_int(batch)))
        self.batch = batch

    def set_source_batch_device(self, ptr, n_per_scan, batch):
        _check(self.lib.sf_icp_set_source_batch_device(self.h, C.c_void_p(ptr), C.c_int64(n_per_scan), C.c_int(batch)))
        self.batch = batch

    def set_initial_batch(self, inits=None):
        if inits is None:
            _check(self.lib.sf_icp_set_initial_batch_f64(self.h, None))
        else:
            inits = _f64(inits).reshape(self.batch, 16)
            _check(self.lib.sf_icp_set_initial_batch_f64(self.h, _p(inits)))

    def set_target(self, target):
        if isinstance(target, Map):
            self._map = target
            _check(self.lib.sf_icp_set_target_map(self.h, target.h))
        else:
            xyz = _f32(target).reshape(-1, 3)
            _check(self.lib.sf_icp_set_target(self.h, _p(xyz), C.c_int64(len(xyz))))

    def use_graph(self, on=True):
        _check(self.lib.sf_icp_use_graph(self.h, C.c_int(int(on))))

    def graph_counts(self):
        a, b = C.c_int64(), C.c_int64()
        _check(self.lib.sf_icp_graph_counts(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_source_scan(self, raw_cloud, stride, center, radius):
        """sf_icp_set_source_scan: subsample + radius crop + set_source in one pass, the count stays on the device"""
        c = _f32(center)
        _check(self.lib.sf_icp_set_source_scan(self.h, raw_cloud.h, C.c_int(int(stride)), _p(c), C.c_double(radius)))
        self.batch = 1

    def source_count(self):
        n = C.c_int64()
        _check(self.lib.sf_icp_source_count(self.h, C.byref(n)))
        return n.value

    def set_fused(self, on=True):
        _check(self.lib.sf_icp_set_fused(self.h, C.c_int(int(on))))

    def fused_count(self):
        a = C.c_int64()
        _check(self.lib.sf_icp_fused_count(self.h, C.byref(a)))
        return a.value

    def fused_redone(self):
        n = C.c_int64()
        _check(self.lib.sf_icp_fused_redone(self.h, C.byref(n)))
        return n.value

    def test_inject_barrier_timeout(self):
        _check(self.lib.sf_icp_test_inject_barrier_timeout(self.h))

    def set_nn_reuse(self, on=True):
        _check(self.lib.sf_icp_set_nn_reuse(self.h, C.c_int(int(on))))

    def set_freeze(self, on="auto"):
        """Frozen pairs (sf_icp_set_freeze): P2PLANE launch list of wide scans, see include/slamfusion.h.
        False / "off", "auto" (default: batches of at least 0.7 M queries), True / "always"."""
        code = {False: 0, "off": 0, "auto": 1, True: 2, "always": 2}[on]
        _check(self.lib.sf_icp_set_freeze(self.h, C.c_int(code)))

    def set_wide_scan_points(self, points):
        """Scans above this many points take the launch list with two queries per lane and may freeze (sf_icp_set_wide_scan_points;
        default 131 072).  Call before set_source*."""
        _check(self.lib.sf_icp_set_wide_scan_points(self.h, C.c_int64(int(points))))

    def set_pipeline(self, on=True):
        """Overlap of consecutive align_batch_async calls on unchanged inputs (sf_icp_set_pipeline, default on)."""
        _check(self.lib.sf_icp_set_pipeline(self.h, C.c_int(int(bool(on)))))

    def set_tile_search(self, on="auto"):
        """Tile search (sf_icp_set_tile_search): the searching launches of large batches served out of LDS tile by tile, see
        include/slamfusion.h.  False / "off", "auto" (default: batches of at least 2 M queries), True / "always"."""
        code = {False: 0, "off": 0, "auto": 1, True: 2, "always": 2}[on]
        _check(self.lib.sf_icp_set_tile_search(self.h, C.c_int(code)))

    def tile_info(self):
        """Of the last alignment (sf_icp_tile_info); the counters are filled while profiling is on."""
        a = (C.c_int64 * 12)()
        _check(self.lib.sf_icp_tile_info(self.h, a))
        return {"on": bool(a[0]), "cells_per_tile": [a[1], a[2], a[3]], "tiles": [a[4], a[5], a[6]], "searched": a[7], "from_lds": a[8], "left_region": a[9],
                "beyond_ring_1": a[10], "tiles_too_dense": a[11]}

    def set_freeze_params(self, guard_scale=8.0, guard_min=2.0e-5, guard_max=3.0e-4, max_tries=3, from_launch=5):
        _check(self.lib.sf_icp_set_freeze_params(self.h, C.c_float(guard_scale), C.c_float(guard_min), C.c_float(guard_max), C.c_int(max_tries), C.c_int(from_launch)))

    def freeze_stats(self):
        """Of the last batched alignment, summed over its scans (sf_icp_freeze_stats)."""
        a = (C.c_int64 * 5)()
        _check(self.lib.sf_icp_freeze_stats(self.h, a))
        return {"froze": a[0], "thawed": a[1], "failed": a[2], "active_queries": a[3], "frozen_at_end": a[4]}

    def set_query_order(self, order="auto"):
        """'auto' | 'as_given' | 'cell' (SF_ORDER_*): the order a scan's points are walked in."""
        _check(self.lib.sf_icp_set_query_order(self.h, C.c_int({"auto": 0, "as_given": 1, "cell": 2}[order])))

    def align(self, mode="ref_cpp"):
        r = IcpResult()
        _check(self.lib.sf_icp_align(self.h, C.c_int(MODES[mode]), C.byref(r)))
        return r.as_dict()

    def align_batch(self, mode="p2plane"):
        arr = (IcpResult * self.batch)()
        _check(self.lib.sf_icp_align_batch(self.h, C.c_int(MODES[mode]), arr))
        self._prev_batch, self._last_batch = getattr(self, "_last_batch", None), self.batch
        return [r.as_dict() for r in arr]

    def align_batch_async(self, mode="p2plane"):
        _check(self.lib.sf_icp_align_batch_async(self.h, C.c_int(MODES[mode])))
        # how many scans the latest / the previous enqueued alignment registered (a source set since belongs to the next one)
        self._prev_batch, self._last_batch = getattr(self, "_last_batch", None), self.batch

    def fetch_results(self, raw=False):
        """Results of the LATEST enqueued alignment, as it was enqueued."""
        arr = (IcpResult * int(getattr(self, "_last_batch", None) or self.batch))()
        _check(self.lib.sf_icp_fetch_results(self.h, arr))
        return arr if raw else [r.as_dict() for r in arr]

    def fetch_previous(self, batch=None, raw=False):
        """Results of the alignment enqueued before the latest one (both asynchronous, no fetch between them): waits for that
        one only; raw: the ctypes array as it came."""
        arr = (IcpResult * int(batch or getattr(self, "_prev_batch", None) or self.batch))()
        _check(self.lib.sf_icp_fetch_previous(self.h, arr))
        return arr if raw else [r.as_dict() for r in arr]

    # multi-GPU stepping
    def set_shard(self, x_lo, x_hi):
        _check(self.lib.sf_icp_set_shard(self.h, C.c_float(x_lo), C.c_float(x_hi)))

    def set_exchange_buffer(self, ptr, nbytes):
        _check(self.lib.sf_icp_set_exchange_buffer(self.h, C.c_void_p(ptr), C.c_int64(nbytes)))

    def exchange_ptr(self):
        n = C.c_int64()
        p = self.lib.sf_icp_exchange_ptr(self.h, C.byref(n))
        return p, n.value

    def set_shard_margin(self, margin_m):
        _check(self.lib.sf_icp_set_shard_margin(self.h, C.c_float(margin_m)))

    def step_begin(self, mode, first):
        _check(self.lib.sf_icp_step_begin(self.h, C.c_int(MODES[mode]), C.c_int(int(first))))

    def step_end(self, mode, last):
        _check(self.lib.sf_icp_step_end(self.h, C.c_int(MODES[mode]), C.c_int(int(last))))

    def align_sharded(self, mode, comm):
        """The whole sharded alignment from the C side (RCCL all-reduce per iteration on the context's stream)."""
        arr = (IcpResult * self.batch)()
        resumes = C.c_int()
        _check(self.lib.sf_icp_align_sharded(self.h, C.c_int(MODES[mode]), comm.h, arr, C.byref(resumes)))
        self.resumes = resumes.value
        return [r.as_dict() for r in arr]

    def align_sharded_async(self, mode, comm, first=1):
        _check(self.lib.sf_icp_align_sharded_async(self.h, C.c_int(MODES[mode]), comm.h, C.c_int(first)))

    def owned_counts(self):
        out = np.empty(self.batch, np.int64)
        _check(self.lib.sf_icp_owned_counts(self.h, _p(out), C.c_int(self.batch)))
        return out

    def profile_enable(self, on=True):
        _check(self.lib.sf_icp_profile_enable(self.h, C.c_int(int(on))))

    def profile_read(self):
        n, ms = C.c_int64(), C.c_double()
        _check(self.lib.sf_icp_profile_read(self.h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def profile_launches(self):
        """-> (ms[k], searched_queries[k], searched_waves[k]) of every profiled NN launch, in launch order"""
        n = C.c_int64()
        _check(self.lib.sf_icp_profile_read_launches(self.h, None, None, None, C.c_int64(0), C.byref(n)))
        ms = np.empty(max(n.value, 1), np.float32)
        q = np.empty(max(n.value, 1), np.uint32)
        w = np.empty(max(n.value, 1), np.uint32)
        _check(self.lib.sf_icp_profile_read_launches(self.h, _p(ms), _p(q), _p(w), C.c_int64(len(ms)), C.byref(n)))
        return ms[:n.value], q[:n.value], w[:n.value]

    def profile_phases(self, kind):
        """Durations [ms] of one phase of the sharded path while profiling (PROF_REDUCE / COLLECTIVE / SOLVE / SHARD_BUILD)."""
        n = C.c_int64()
        _check(self.lib.sf_icp_profile_read_phases(self.h, C.c_int(kind), None, C.c_int64(0), C.byref(n)))
        ms = np.empty(max(n.value, 1), np.float32)
        _check(self.lib.sf_icp_profile_read_phases(self.h, C.c_int(kind), _p(ms), C.c_int64(len(ms)), C.byref(n)))
        return ms[:n.value]

    def close(self):
        if self.h:
            self.lib.sf_icp_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def align_group(members, mode):
    """Several slabs of one map held by this process on one device (sf_icp_align_group) -> (results, resumes)."""
    n = len(members)
    arr = (IcpResult * members[0].batch)()
    hs = (C.c_void_p * n)(*[m.h for m in members])
    resumes = C.c_int()
    _check(load_library().sf_icp_align_group(hs, C.c_int(n), C.c_int(MODES[mode]), arr, C.byref(resumes)))
    return [r.as_dict() for r in arr], resumes.value


def shard_route(scans, inits, edges, margin):
    """Slab range [lo, hi] of every scan (sf_shard_route)."""
    scans = _f32(scans)
    assert scans.ndim == 3 and scans.shape[2] == 3
    B = scans.shape[0]
    T = None if inits is None else _f64(inits).reshape(B, 16)
    e = _f64(edges)
    lo, hi = np.empty(B, np.int32), np.empty(B, np.int32)
    _check(load_library().sf_shard_route(_p(scans), C.c_int64(scans.shape[1]), C.c_int(B), _p(T) if T is not None else None, _p(e), C.c_int(len(e) - 1),
                                         C.c_double(margin), _p(lo), _p(hi)))
    return lo, hi


def rccl_library_path():
    """The RCCL the process must share: the one inside the torch wheel when torch is importable (its HIP runtime is
    the one this process uses, see load_library), else the system's."""
    try:
        import torch
        p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if os.path.exists(p):
            return p
    except ImportError:
        pass
    return None


class Comm:
    """sf_comm: the communicator the C side all-reduces the normal-equation records on, bound to a context's stream.

    Comm(ctx, nranks, rank, unique_id)        RCCL, created from a 128-byte unique id
    Comm.p2p(ctx, nranks, rank, max_count)    the hand-written P2P transport (hipIpc store-and-flag, fixed rank-order sum):
                                              then .handle() -> bytes, .connect([handles in rank order]) -- or
                                              .rendezvous(name) for ranks of one node without a launcher
    """

    @staticmethod
    def unique_id():
        lib = load_library()
        path = rccl_library_path()
        _check(lib.sf_comm_load_rccl(path.encode() if path else None))
        buf = (C.c_char * 128)()
        _check(lib.sf_comm_unique_id(buf))
        return bytes(buf)

    def __init__(self, ctx, nranks, rank, unique_id=None, _p2p_max_count=None):
        self.ctx, self.lib = ctx, ctx.lib
        self.h = C.c_void_p()
        if _p2p_max_count is not None:
            _check(self.lib.sf_comm_p2p_create(ctx.h, C.c_int(nranks), C.c_int(rank), C.c_int64(int(_p2p_max_count)), C.byref(self.h)))
            self.kind = "p2p"
        else:
            path = rccl_library_path()
            _check(self.lib.sf_comm_load_rccl(path.encode() if path else None))
            assert len(unique_id) == 128
            _check(self.lib.sf_comm_create(ctx.h, C.c_int(nranks), C.c_int(rank), (C.c_char * 128).from_buffer_copy(unique_id), C.byref(self.h)))
            self.kind = "rccl"
        self.nranks, self.rank = nranks, rank
        _live.add(self)

    @classmethod
    def p2p(cls, ctx, nranks, rank, max_count):
        return cls(ctx, nranks, rank, _p2p_max_count=max_count)

    def handle(self):
        buf = (C.c_char * 128)()
        _check(self.lib.sf_comm_p2p_handle(self.h, buf))
        return bytes(buf)

    def connect(self, handles):
        blob = b"".join(handles)
        assert len(blob) == 128 * self.nranks
        _check(self.lib.sf_comm_p2p_connect(self.h, (C.c_char * len(blob)).from_buffer_copy(blob)))

    def rendezvous(self, name, timeout_s=60.0):
        _check(self.lib.sf_comm_p2p_rendezvous(self.h, name.encode(), C.c_double(timeout_s)))

    def set_timeout(self, seconds):
        _check(self.lib.sf_comm_set_timeout(self.h, C.c_double(seconds)))

    def abort(self):
        _check(self.lib.sf_comm_abort(self.h))

    def status(self):
        """Raises SlamFusionError (SF_ERR_COMM) once a collective of this communicator timed out or was aborted."""
        _check(self.lib.sf_comm_status(self.h))

    def allreduce_f64(self, ptr, count):
        _check(self.lib.sf_comm_allreduce_f64(self.h, C.c_void_p(ptr), C.c_int64(count)))

    def close(self):
        if self.h:
            self.lib.sf_comm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BruteForceAlignment:
    """sf_bf: mirrors BruteForceAlignment (brute_force_alignment.h:22-112) setter for setter."""

    def __init__(self, ctx):
        self.ctx, self.lib = ctx, ctx.lib
        self.h = C.c_void_p()
        self._map = None
        _check(self.lib.sf_bf_create(ctx.h, C.byref(self.h)))
        _live.add(self)

    def setXYZStep(self, x, y, z):
        _check(self.lib.sf_bf_set_xyz_step(self.h, C.c_float(x), C.c_float(y), C.c_float(z)))

    def setXYZRange(self, x, y, z):
        _check(self.lib.sf_bf_set_xyz_range(self.h, C.c_float(x), C.c_float(y), C.c_float(z)))

    def setRotationStep(self, v):
        _check(self.lib.sf_bf_set_rotation_step(self.h, C.c_float(v)))

    def setRotationRange(self, v):
        _check(self.lib.sf_bf_set_rotation_range(self.h, C.c_float(v)))

    def setMeanErrorThreshold(self, v):
        _check(self.lib.sf_bf_set_mean_error_threshold(self.h, C.c_float(v)))

    def setInitialGuess(self, T):
        T = _f32(T).reshape(16)
        _check(self.lib.sf_bf_set_initial_guess(self.h, _p(T)))

    def setSourceCloud(self, cloud):
        if isinstance(cloud, Cloud):
            _check(self.lib.sf_bf_set_source_cloud(self.h, cloud.h))
        else:
            xyz = _f32(cloud).reshape(-1, 3)
            _check(self.lib.sf_bf_set_source(self.h, _p(xyz), C.c_int64(len(xyz))))

    def setTargetCloud(self, target):
        if isinstance(target, Map):
            self._map = target
            _check(self.lib.sf_bf_set_target_map(self.h, target.h))
        else:
            xyz = _f32(target).reshape(-1, 3)
            _check(self.lib.sf_bf_set_target(self.h, _p(xyz), C.c_int64(len(xyz))))

    def resetFirstAlignment(self, value):
        _check(self.lib.sf_bf_reset_first_alignment(self.h, C.c_int(int(value))))

    def alignClouds(self):
        found = C.c_int()
        _check(self.lib.sf_bf_align_clouds(self.h, C.byref(found)))
        return bool(found.value)

    def firstAlignmentCompleted(self):
        return bool(self.lib.sf_bf_first_alignment_completed(self.h))

    def getBestTransformation(self):
        T = np.empty(16, np.float32)
        _check(self.lib.sf_bf_get_best_transformation(self.h, _p(T)))
        return T.reshape(4, 4)

    def last_result(self):
        idx, score, ncand = C.c_int32(), C.c_float(), C.c_int32()
        _check(self.lib.sf_bf_last_result(self.h, C.byref(idx), C.byref(score), C.byref(ncand), None, C.c_int64(0)))
        scores = np.empty(max(ncand.value, 1), np.float32)
        _check(self.lib.sf_bf_last_result(self.h, None, None, None, _p(scores), C.c_int64(len(scores))))
        return dict(index=idx.value, score=score.value, n_candidates=ncand.value, scores=scores[:ncand.value])

    def close(self):
        if self.h:
            self.lib.sf_bf_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NodeParams(C.Structure):
    """sf_node_params"""
    _fields_ = [("ref_frame_distance", C.c_float), ("cloud_crop_radius", C.c_float), ("index_cell", C.c_float), ("pcl_crop_order", C.c_int32),
                ("icp_mode", C.c_int32), ("map_is_downsampled", C.c_int32)]


class GpsFix(C.Structure):
    """sf_gps_fix"""
    _fields_ = [("latitude", C.c_double), ("longitude", C.c_double), ("altitude", C.c_double), ("position_covariance", C.c_double * 9)]


class Odom(C.Structure):
    """sf_odom"""
    _fields_ = [("q_wxyz", C.c_double * 4), ("t", C.c_double * 3), ("covariance", C.c_double * 36)]


class NodeOutput(C.Structure):
    """sf_node_output"""
    _fields_ = [("status", C.c_int32), ("recropped", C.c_int32), ("coarse_ran", C.c_int32), ("pad_", C.c_int32), ("n_scan", C.c_int64),
                ("map_T_sensor", C.c_float * 16), ("prior", C.c_float * 16), ("odom_pose", C.c_float * 16), ("gps_pose", C.c_float * 16),
                ("odometry_gain", C.c_float), ("gps_compass_gain", C.c_float), ("icp", IcpResult), ("coarse_icp", IcpResult)]


SF_NODE_OK, SF_NODE_GATED_ALTITUDE, SF_NODE_FIRST_MESSAGE, SF_NODE_COARSE_FAILED = 0, 1, 2, 3
SF_NODE_POSE_MAP_T_SENSOR, SF_NODE_POSE_MAP_T_REF, SF_NODE_POSE_ODOM_PREVIOUS = 0, 1, 2


class _BorrowedIcp(Icp):
    """The icp_ of a Node: same methods, the handle belongs to the node."""

    def __init__(self, ctx, handle):
        self.ctx, self.lib = ctx, ctx.lib
        self.h = C.c_void_p(handle)
        self._map = None
        self.batch = 1

    def close(self):
        self.h = C.c_void_p()


class _BorrowedBf(BruteForceAlignment):
    """The brute_force_alignment_ of a Node."""

    def __init__(self, ctx, handle):
        self.ctx, self.lib = ctx, ctx.lib
        self.h = C.c_void_p(handle)
        self._map = None

    def close(self):
        self.h = C.c_void_p()


class Node:
    """sf_node: LocalizationNode's per-scan callback (localization_node.cpp:263-344) as one library call."""

    def __init__(self, ctx, map_points, map_T_global, altitude_table=None, **params):
        self.ctx, self.lib = ctx, ctx.lib
        self.h = C.c_void_p()
        prm = NodeParams()
        self.lib.sf_node_default_params(C.byref(prm))
        for k, v in params.items():
            if k == "icp_mode":
                v = MODES[v] if isinstance(v, str) else v
            setattr(prm, k, type(getattr(prm, k))(v))
        self.params = prm
        xyz = _f32(map_points).reshape(-1, 3)
        M = _f64(map_T_global).reshape(16)
        tab = _f64(np.zeros((0, 3)) if altitude_table is None else altitude_table).reshape(-1, 3)
        _check(self.lib.sf_node_create(ctx.h, _p(xyz), C.c_int64(len(xyz)), _p(M), _p(tab), C.c_int(len(tab)), C.byref(prm), C.byref(self.h)))
        self.icp = _BorrowedIcp(ctx, self.lib.sf_node_icp(self.h))
        self.bf = _BorrowedBf(ctx, self.lib.sf_node_bf(self.h))
        self._gps, self._odom, self._out = GpsFix(), Odom(), NodeOutput()
        _live.add(self)

    def compass(self, compass_deg):
        _check(self.lib.sf_node_compass(self.h, C.c_double(compass_deg)))

    def _messages(self, gps, odom):
        g, o = self._gps, self._odom
        g.latitude, g.longitude, g.altitude = gps["latitude"], gps["longitude"], gps["altitude"]
        for dst, src, n in ((g.position_covariance, gps["position_covariance"], 9), (o.q_wxyz, odom["q_wxyz"], 4), (o.t, odom["t"], 3), (o.covariance, odom["covariance"], 36)):
            a = np.ascontiguousarray(src, dtype=np.float64)           # one block copy instead of a Python-level loop over the elements
            if a.size != n:
                raise SlamFusionError("message field has %d values, %d expected" % (a.size, n))
            C.memmove(dst, a.ctypes.data, 8 * n)
        return g, o

    def callback(self, scan, gps, odom):
        """scan: float32 [n, 3] or a PointCloud2-like message (data, width, height, point_step, row_step, offsets);
        returns the sf_node_output structure (valid until the next call)."""
        g, o = self._messages(gps, odom)
        if hasattr(scan, "point_step"):
            buf, width, height, point_step, row_step, offs, dtype, big = _pc2_layout(scan)
            _check(self.lib.sf_node_callback_pointcloud2(self.h, _p(buf), C.c_int64(buf.size), C.c_int64(width), C.c_int64(height), C.c_int(point_step), C.c_int64(row_step),
                                                         C.c_int(offs[0]), C.c_int(offs[1]), C.c_int(offs[2]), C.c_int(dtype), C.c_int(big), C.byref(g), C.byref(o), C.byref(self._out)))
        else:
            xyz = _f32(scan).reshape(-1, 3)
            _check(self.lib.sf_node_callback_xyz(self.h, _p(xyz), C.c_int64(len(xyz)), C.byref(g), C.byref(o), C.byref(self._out)))
        return self._out

    def get_pose(self, which=SF_NODE_POSE_MAP_T_SENSOR):
        T = np.empty(16, np.float32)
        _check(self.lib.sf_node_get_pose(self.h, C.c_int(which), _p(T)))
        return T.reshape(4, 4)

    def set_pose(self, which, T):
        T = _f32(T).reshape(16)
        _check(self.lib.sf_node_set_pose(self.h, C.c_int(which), _p(T)))

    def coarse_alignment_complete(self):
        return bool(self.lib.sf_node_coarse_alignment_complete(self.h))

    def set_coarse_alignment_complete(self, v=True):
        _check(self.lib.sf_node_set_coarse_alignment_complete(self.h, C.c_int(int(v))))

    def close(self):
        if self.h:
            self.icp.close()
            self.bf.close()
            self.lib.sf_node_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pcd_read(path):
    lib = load_library()
    ptr, n = C.POINTER(C.c_float)(), C.c_int64()
    _check(lib.sf_pcd_read(str(path).encode(), C.byref(ptr), C.byref(n)))
    out = np.ctypeslib.as_array(ptr, shape=(max(n.value, 1) * 3,))[:n.value * 3].reshape(-1, 3).copy()
    lib.sf_free(ptr)
    return out


def pcd_write_binary(path, xyz):
    xyz = _f32(xyz).reshape(-1, 3)
    _check(load_library().sf_pcd_write_binary(str(path).encode(), _p(xyz), C.c_int64(len(xyz))))


class MapDataSaver:
    """sf_recorder: the file writers of mapping/src/map_data_save_node.cpp (tiles + the two text logs)."""

    def __init__(self, map_data_path):
        self.lib = load_library()
        self.lib.sf_recorder_compass_yaw.restype = C.c_double
        self.h = C.c_void_p()
        _check(self.lib.sf_recorder_create(str(map_data_path).encode(), C.byref(self.h)))
        self.current_compass_yaw = 0.0

    def compassCallback(self, hdg_deg):
        self.current_compass_yaw = float(self.lib.sf_recorder_compass_yaw(C.c_double(hdg_deg)))

    def mappingCallback(self, xyz, odom_xyz, lat, lon, alt):
        xyz, o = _f32(xyz).reshape(-1, 3), _f64(odom_xyz).reshape(3)
        _check(self.lib.sf_recorder_add(self.h, _p(xyz), C.c_int64(len(xyz)), _p(o), C.c_double(lat), C.c_double(lon), C.c_double(alt),
                                        C.c_double(self.current_compass_yaw)))

    def onShutdown(self):
        _check(self.lib.sf_recorder_shutdown(self.h))

    def __del__(self):
        try:
            if self.h:
                self.lib.sf_recorder_destroy(self.h)
                self.h = None
        except Exception:
            pass


class GlobalMapFramesManager:
    """sf_frames: GlobalMapFramesManager of localization/src/global_map_frames_manager.cpp."""

    def __init__(self, data_folder, map_name="map", num_poses_max=50):
        self.lib = load_library()
        self.lib.sf_frames_create.restype = C.c_void_p
        self.lib.sf_frames_get_closest_altitude.restype = C.c_float
        self.h = C.c_void_p(self.lib.sf_frames_create(str(data_folder).encode(), str(map_name).encode(), C.c_int64(num_poses_max)))

    def getMapCloud(self, ctx, voxel_size=0.1):
        cloud = Cloud(ctx)
        cached = C.c_int()
        _check(self.lib.sf_frames_get_map_cloud(self.h, cloud.h, C.c_float(voxel_size), C.byref(cached)))
        self.loaded_cached = bool(cached.value)
        return cloud

    def getMapTGlobal(self):
        T = np.empty(16, np.float64)
        _check(self.lib.sf_frames_get_map_T_global(self.h, _p(T)))
        return T.reshape(4, 4)

    def getClosestAltitude(self, lat, lon):
        return float(self.lib.sf_frames_get_closest_altitude(self.h, C.c_double(lat), C.c_double(lon)))

    def altitude_table(self):
        rows = C.c_int64()
        _check(self.lib.sf_frames_altitude_table(self.h, None, C.c_int64(0), C.byref(rows)))
        tab = np.empty((max(rows.value, 1), 3), np.float64)
        _check(self.lib.sf_frames_altitude_table(self.h, _p(tab), C.c_int64(len(tab)), C.byref(rows)))
        return tab[:rows.value]

    def __del__(self):
        try:
            if self.h:
                self.lib.sf_frames_destroy(self.h)
                self.h = None
        except Exception:
            pass


# ------------------------------------------------------------------ pose fusion (host C++)
def quat_to_pose(q_wxyz, t):
    q, t = _f64(q_wxyz), _f64(t)
    T = np.empty(16, np.float32)
    load_library().sf_fusion_quat_to_pose(_p(q), _p(t), _p(T))
    return T.reshape(4, 4)


def odom_prediction(map_T_sensor, odom_T_prev, odom_T_cur):
    a, b, c = (_f32(x).reshape(16) for x in (map_T_sensor, odom_T_prev, odom_T_cur))
    out = np.empty(16, np.float32)
    load_library().sf_fusion_odom_prediction(_p(a), _p(b), _p(c), _p(out))
    return out.reshape(4, 4)


def compass_to_yaw(deg):
    return float(load_library().sf_fusion_compass_to_yaw(C.c_double(deg)))


def ll_to_utm(lat, lon):
    n, e = C.c_double(), C.c_double()
    load_library().sf_fusion_ll_to_utm(C.c_double(lat), C.c_double(lon), C.byref(n), C.byref(e))
    return n.value, e.value


def closest_altitude(table, lat, lon):
    table = _f64(table).reshape(-1, 3)
    return float(load_library().sf_fusion_closest_altitude(_p(table), C.c_int(len(table)), C.c_double(lat), C.c_double(lon)))


def gps_pose(map_T_global, yaw, lat, lon, table_alt):
    M = _f64(map_T_global).reshape(16)
    out = np.empty(16, np.float32)
    load_library().sf_fusion_gps_pose(_p(M), C.c_float(yaw), C.c_double(lat), C.c_double(lon), C.c_float(table_alt), _p(out))
    return out.reshape(4, 4)


def pose_gains(gps_cov, odom_cov, fixed=False):
    g, o = _f64(gps_cov).reshape(9), _f64(odom_cov).reshape(36)
    a, b = C.c_float(), C.c_float()
    load_library().sf_fusion_pose_gains(_p(g), _p(o), C.c_int(int(fixed)), C.byref(a), C.byref(b))
    return a.value, b.value


def blend(g_odom, T_odom, g_gps, T_gps):
    a, b = _f32(T_odom).reshape(16), _f32(T_gps).reshape(16)
    out = np.empty(16, np.float32)
    load_library().sf_fusion_blend(C.c_float(g_odom), _p(a), C.c_float(g_gps), _p(b), _p(out))
    return out.reshape(4, 4)


def map_T_global(latlonalt, yaw):
    l, y = _f64(latlonalt).reshape(-1, 3), _f32(yaw)
    out = np.empty(16, np.float64)
    load_library().sf_fusion_map_T_global(_p(l), _p(y), C.c_int(len(l)), _p(out))
    return out.reshape(4, 4)


def mat4f_inverse(A):
    A = _f32(A).reshape(16)
    out = np.empty(16, np.float32)
    load_library().sf_fusion_mat4f_inverse(_p(A), _p(out))
    return out.reshape(4, 4)


def mat4f_mul(A, B):
    A, B = _f32(A).reshape(16), _f32(B).reshape(16)
    out = np.empty(16, np.float32)
    load_library().sf_fusion_mat4f_mul(_p(A), _p(B), _p(out))
    return out.reshape(4, 4)


class Ekf:
    """sf_ekf: error-state EKF pose prior with IMU pre-integration (extension f-4; the reference has none)."""

    def __init__(self):
        self.lib = load_library()
        self.h = C.c_void_p()
        _check(self.lib.sf_ekf_create(C.byref(self.h)))

    def reset(self, T, v=None, P_diag=None):
        T = _f64(T).reshape(16)
        v = None if v is None else _f64(v).reshape(3)
        P = None if P_diag is None else _f64(P_diag).reshape(9)
        _check(self.lib.sf_ekf_reset(self.h, _p(T), _p(v) if v is not None else None, _p(P) if P is not None else None))

    def set_noise(self, gyro_sigma, accel_sigma, gravity=None):
        g = None if gravity is None else _f64(gravity).reshape(3)
        _check(self.lib.sf_ekf_set_noise(self.h, C.c_double(gyro_sigma), C.c_double(accel_sigma), _p(g) if g is not None else None))

    def set_bias(self, gyro_bias=None, accel_bias=None, gyro_bias_var=None, accel_bias_var=None):
        args = [None if a is None else _f64(a).reshape(3) for a in (gyro_bias, accel_bias, gyro_bias_var, accel_bias_var)]
        _check(self.lib.sf_ekf_set_bias(self.h, *[_p(a) if a is not None else None for a in args]))

    def set_bias_noise(self, gyro_bias_walk, accel_bias_walk):
        _check(self.lib.sf_ekf_set_bias_noise(self.h, C.c_double(gyro_bias_walk), C.c_double(accel_bias_walk)))

    def full_state(self):
        """-> (gyro bias, accelerometer bias, 15x15 covariance over dp, dv, dtheta, dbg, dba)"""
        bg, ba, P = np.empty(3), np.empty(3), np.empty(225)
        _check(self.lib.sf_ekf_get_full(self.h, _p(bg), _p(ba), _p(P)))
        return bg, ba, P.reshape(15, 15)

    def predict_imu(self, gyro, accel, dt):
        gyro, accel = _f64(gyro).reshape(-1, 3), _f64(accel).reshape(-1, 3)
        assert len(gyro) == len(accel)
        _check(self.lib.sf_ekf_predict_imu(self.h, _p(gyro), _p(accel), C.c_int64(len(gyro)), C.c_double(dt)))

    def predict_odometry(self, T_prev, T_cur, cov_pos=None, cov_rot=None):
        a, b = _f64(T_prev).reshape(16), _f64(T_cur).reshape(16)
        cp = None if cov_pos is None else _f64(cov_pos).reshape(3)
        cr = None if cov_rot is None else _f64(cov_rot).reshape(3)
        _check(self.lib.sf_ekf_predict_odometry(self.h, _p(a), _p(b), _p(cp) if cp is not None else None, _p(cr) if cr is not None else None))

    def update_position(self, z, cov):
        z, cov = _f64(z).reshape(3), _f64(cov).reshape(9)
        _check(self.lib.sf_ekf_update_position(self.h, _p(z), _p(cov)))

    def update_yaw(self, yaw, var):
        _check(self.lib.sf_ekf_update_yaw(self.h, C.c_double(yaw), C.c_double(var)))

    def update_pose(self, T, cov_pos, cov_rot):
        T, cp, cr = _f64(T).reshape(16), _f64(cov_pos).reshape(3), _f64(cov_rot).reshape(3)
        _check(self.lib.sf_ekf_update_pose(self.h, _p(T), _p(cp), _p(cr)))

    def state(self):
        T, v, P = np.empty(16), np.empty(3), np.empty(81)
        _check(self.lib.sf_ekf_get(self.h, _p(T), _p(v), _p(P)))
        return T.reshape(4, 4), v, P.reshape(9, 9)

    def close(self):
        if self.h:
            self.lib.sf_ekf_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class StochasticFilter:
    """sf_sfilter: StochasticFilter of localization/src/stochastic_filter.cpp."""

    def __init__(self, queue_size=10, n_std_dev_threshold=1.0):
        self.lib = load_library()
        self.q = queue_size
        self.h = C.c_void_p(self.lib.sf_sfilter_create(C.c_int(queue_size), C.c_float(n_std_dev_threshold)))
        if not self.h:
            raise SlamFusionError("sf_sfilter_create failed")

    def setMaximumLinearVelocity(self, v):
        self.lib.sf_sfilter_set_maximum_linear_velocity(self.h, C.c_float(v))

    def weights(self):
        w = np.empty(self.q, np.float32)
        self.lib.sf_sfilter_weights(self.h, _p(w))
        return w

    def addPoseToQueue(self, pose):
        p = _f32(pose).reshape(16)
        self.lib.sf_sfilter_add_pose_to_queue(self.h, _p(p))

    def computePoseZScore(self, prev, cur):
        a, b = _f32(prev).reshape(16), _f32(cur).reshape(16)
        return float(self.lib.sf_sfilter_pose_zscore(self.h, _p(a), _p(b)))

    def applyGaussianFilterToCurrentPose(self, prev, cur):
        a, b = _f32(prev).reshape(16), _f32(cur).reshape(16)
        out = np.empty(16, np.float32)
        self.lib.sf_sfilter_apply_gaussian_filter(self.h, _p(a), _p(b), _p(out))
        return out.reshape(4, 4)

    def __del__(self):
        try:
            if self.h:
                self.lib.sf_sfilter_destroy(self.h)
                self.h = None
        except Exception:
            pass
