"""First end-to-end GPU check: NN / voxel / ICP parity against the oracle + rough timing."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slam_sensor_fusion_amd import api, synth
from oracle import oracle as orc

ctx = api.Context(0)
print("device:", ctx.device_name())
M = int(os.environ.get("M", 100_000)); N = int(os.environ.get("N", 10_000))
raw = synth.make_map(M)
# voxel (pcl)
c = api.Cloud(ctx, raw)
t0 = time.time(); fl = c.voxel_downsample(0.1, "pcl"); ctx.synchronize(); t1 = time.time()
ds_gpu = c.download(); pid = c.voxel_point_ids(); oid = c.voxel_out_ids()
ds, vidx, ovox, st = orc.voxel_pcl(raw, 0.1)
print("voxel pcl: gpu", len(ds_gpu), "oracle", len(ds), "flags", fl, "time %.3f" % (t1 - t0))
print("  point ids equal:", np.array_equal(pid, vidx), " out ids equal:", np.array_equal(oid, ovox), " centroids bit-equal:", np.array_equal(ds_gpu, ds))
# voxel o3d
c2 = api.Cloud(ctx, raw); c2.voxel_downsample(0.1, "o3d")
m64 = c2.voxel_out_means_f64(); ijk_p = c2.voxel_point_ids().reshape(-1, 3); ijk_o = c2.voxel_out_ids().reshape(-1, 3)
om, oijk, ooijk, st2 = orc.voxel_o3d(raw.astype(np.float64), 0.1)
print("voxel o3d: gpu", len(m64), "oracle", len(om), " ijk equal:", np.array_equal(ijk_p, oijk), np.array_equal(ijk_o, ooijk), " means bit-equal:", np.array_equal(m64, om))

scan, sidx = synth.make_scan(ds, N)
mp = api.Map(ctx, api.Cloud(ctx, ds), 0.25)
print("map", len(mp), "cell", mp.cell_size())
# NN parity
q = (scan.astype(np.float64) @ synth.t_true()[:3, :3].T + synth.t_true()[:3, 3]).astype(np.float32)
qq = np.concatenate([q, scan, scan + np.float32(3.0), np.array([[1e6, 0, 0], [np.nan, 0, 0]], np.float32)])
gi, gd = mp.nn(qq)
tree = orc.KdTreeF(ds); oi, od = tree.nn(qq)
same = gi == oi
print("NN: idx equal %.6f ; mismatches with equal d2: %d of %d ; d2 bit-equal: %s" % (same.mean(), int((gd[~same] == od[~same]).sum()), int((~same).sum()), np.array_equal(gd[np.isfinite(od)], od[np.isfinite(od)])))
gi2, gd2 = mp.nn(qq, 0.5)
print("NN thr 0.5: rejected gpu %d oracle %d" % ((gi2 < 0).sum(), (od >= 0.5).sum()), np.array_equal(gi2 >= 0, od < 0.5))

# ICP
for mode in ("ref_cpp", "o3d_p2p"):
    icp = api.Icp(ctx, 0.5, 10 if mode == "ref_cpp" else 30, 0.05, 1e-5)
    icp.set_target(mp); icp.set_source(scan); icp.set_initial_transformation(np.eye(4, dtype=np.float32))
    t0 = time.time(); r = icp.align(mode); t1 = time.time()
    if mode == "ref_cpp":
        o32 = orc.icp_ref_cpp(scan, ds, precise=False); o64 = orc.icp_ref_cpp(scan, ds, precise=True)
    else:
        o64 = orc.icp_o3d_p2p(scan, ds, max_iter=30); o32 = o64
    print(mode, "gpu: it %d err %.6g conv %s ncorr %d nres %d | oracle64: it %d err %.6g conv %s ncorr %d" % (r['iterations'], r['error'], r['converged'], r['n_corr'], r['n_research'], o64['iterations'], o64['error'], o64['converged'], o64['n_corr']))
    print("   pose err vs f64 oracle:", synth.pose_error(r['T64'], o64['T']), " vs f32 oracle:", synth.pose_error(r['T64'], o32['T']), " vs truth:", synth.pose_error(r['T64'], synth.t_true()), "time %.4f" % (t1 - t0))
# p2plane
t0 = time.time(); mp.estimate_normals(0.25); t1 = time.time()
gn, gc = mp.download_normals()
on, oc = orc.normals_radius(ds, 0.25)
dots = np.abs((gn * on).sum(1))
print("normals: time %.3f counts equal %s ; |dot| min %.9f ; frac<1-1e-6: %.2e" % (t1 - t0, np.array_equal(gc, oc), dots.min(), (dots < 1 - 1e-6).mean()))
icp = api.Icp(ctx, 0.5, 20, 0.05, 1e-5); icp.set_target(mp); icp.set_source(scan)
t0 = time.time(); r = icp.align("p2plane"); t1 = time.time()
o = orc.icp_p2plane(scan, ds, gn, num_iters=20)
print("p2plane gpu: it %d rmse %.6g fit %.4f | oracle: it %d rmse %.6g fit %.4f" % (r['iterations'], r['rmse'], r['fitness'], o['iterations'], o['error'], o['fitness']))
print("   pose err vs oracle:", synth.pose_error(r['T64'], o['T']), " vs truth:", synth.pose_error(r['T64'], synth.t_true()), "time %.4f" % (t1 - t0))
# timing
for mode, iters in (("p2plane", 20), ("o3d_p2p", 30), ("ref_cpp", 10)):
    icp = api.Icp(ctx, 0.5, iters, 0.05, 1e-5); icp.set_target(mp); icp.set_source(scan)
    for g in (False, True):
        icp.use_graph(g)
        icp.align(mode)
        t0 = time.time()
        for _ in range(20): icp.align(mode)
        print("timing %s graph=%s: %.3f ms/scan" % (mode, g, (time.time() - t0) / 20 * 1e3))
