#!/usr/bin/env python3
"""Generates tests/golden/*.npz.

The reference ships NO fixtures, golden vectors or recorded scans for this path
(SURVEY.md §4), and neither of its implementations can be built or imported here (PCL /
Eigen / rclcpp / open3d / utm absent) — PARITY UNPINNED upstream.  These vectors are
therefore produced by oracle/ (the CPU restatement) after it has been cross-checked by
tests/test_oracle_*.py against scipy.spatial.cKDTree, numpy.linalg.svd, scipy Rotation,
analytic known answers and oracle/_ref (the reference's own geo_lib.hpp compiled here).
The UTM block is produced by oracle/_ref itself, i.e. by reference code.
They freeze the oracle (CPU tests) and are what the HIP path is compared with on the GPU
box, where /root/reference and the oracle's _ref build recipe do not exist.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from slam_sensor_fusion_amd import synth  # noqa: E402


def main():
    orc.build()
    raw = synth.make_map(4000, seed=11)
    raw[7] = [np.nan, 0.0, 0.0]                    # one non-finite point: PCL skips it
    ds, vidx, ovox, st = orc.voxel_pcl(raw, 0.1)
    fin = raw[np.isfinite(raw).all(1)]
    o3d_means, ijk, oijk, st2 = orc.voxel_o3d(fin.astype(np.float64), 0.1)
    scan, sidx = synth.make_scan(ds, 600, scan_id=5)
    tree = orc.KdTreeF(ds)
    q = np.concatenate([scan, scan + np.float32(0.37), np.array([[50.0, 0, 0]], np.float32)])
    nn_idx, nn_d2 = tree.nn(q)
    nrm, cnt = orc.normals_radius(ds, 0.3)
    r_ref32 = orc.icp_ref_cpp(scan, ds, precise=False)
    r_ref64 = orc.icp_ref_cpp(scan, ds, precise=True)
    r_o3d = orc.icp_o3d_p2p(scan, ds, max_iter=30)
    r_pl = orc.icp_p2plane(scan, ds, nrm, num_iters=20)
    crop_r, crop_ri = orc.crop_radius(ds, [0.3, -0.2, 0.1], 0.8)
    crop_a, crop_ai = orc.crop_aabb(ds, [0, -0.5, 0], [1.0, 0.5, 0.5])
    Rb = 0.8 * synth.rpy_to_R(0.01, -0.02, 0.3) + 0.2 * np.eye(3)   # blended, non-orthonormal
    crop_o, crop_oi = orc.crop_obb(ds, [0.1, 0.2, 0.0], Rb, [1.5, 0.8, 0.8])
    np.savez_compressed(os.path.join(HERE, "registration_small.npz"),
                        raw=raw, map=ds, vox_point_ids=vidx, vox_out_ids=ovox,
                        o3d_means=o3d_means, o3d_point_ijk=ijk, o3d_out_ijk=oijk,
                        scan=scan, nn_queries=q, nn_idx=nn_idx, nn_d2=nn_d2,
                        normals=nrm, normal_counts=cnt,
                        ref32_T=r_ref32["T"], ref32_meta=np.array([r_ref32["iterations"], r_ref32["converged"], r_ref32["n_corr"], r_ref32["n_research"]]),
                        ref32_error=r_ref32["error"],
                        ref64_T=r_ref64["T"], ref64_meta=np.array([r_ref64["iterations"], r_ref64["converged"], r_ref64["n_corr"], r_ref64["n_research"]]),
                        ref64_error=r_ref64["error"],
                        o3d_T=r_o3d["T"], o3d_meta=np.array([r_o3d["iterations"], r_o3d["converged"], r_o3d["n_corr"]]), o3d_rmse=r_o3d["error"], o3d_fitness=r_o3d["fitness"],
                        pl_T=r_pl["T"], pl_meta=np.array([r_pl["iterations"], r_pl["n_corr"]]), pl_rmse=r_pl["error"],
                        crop_radius_idx=crop_ri, crop_aabb_idx=crop_ai, crop_obb_idx=crop_oi, obb_R=Rb)

    # ---- round-2 additions: int64 voxel grid past PCL's overflow, covariances of the normal estimation, brute-force scores
    rng2 = np.random.Generator(np.random.PCG64(12))
    far = np.concatenate([rng2.uniform(0, 1, (300, 3)), rng2.uniform(0, 1, (300, 3)) + [2500.0, 1800.0, 900.0]]).astype(np.float32)
    far_ds, far_pid, far_oid = orc.voxel_pcl64(far, 0.1)
    assert orc.voxel_pcl(far, 0.1)[3] == -1
    _, _, cov6 = orc.normals_radius_cov(ds, float(np.float32(0.3)))
    bf_T = synth.make_T((0.2, -0.1, 0.05), (0, 0, 10.0))
    bf_scan, _ = synth.make_scan(ds, 300, scan_id=9, T=bf_T)
    bf_prev = synth.make_T((0.01, 0.0, 0.0), (0, 0, 0.5)).astype(np.float32)
    bf_prm = dict(x_step=0.1, y_step=0.1, z_step=0.05, yaw_step=np.pi / 18.0, x_range=0.3, y_range=0.3, z_range=0.1, yaw_range=np.pi / 6.0)
    bf = orc.bf_align(bf_scan, ds, bf_prev, threshold=1e-9, **bf_prm)
    np.savez_compressed(os.path.join(HERE, "extensions_small.npz"), far=far, far_ds=far_ds, far_point_ids=far_pid, far_out_ids=far_oid,
                        cov6=cov6, bf_scan=bf_scan, bf_prev=bf_prev, bf_scores=bf["scores"], bf_best_T=bf["best_T"], bf_index=np.array([bf["index"], bf["n_candidates"]]))

    # ---- fusion vectors
    ll = np.array([[-22.9068, -43.1729], [48.8566, 2.3522], [59.9, 10.7], [0.0, 0.0], [-33.86, 151.21], [35.68, 139.69], [64.1, -21.9]])
    utm_ref = np.array([orc.ref_ll_to_utm(a, b) for a, b in ll]) if orc.ref_lib() is not None else np.array([orc.ll_to_utm(a, b) for a, b in ll])
    rng = np.random.Generator(np.random.PCG64(77))
    poses = []
    f = orc.StochasticFilter(4, 3.0)
    zs, outs = [], []
    prev = np.eye(4, dtype=np.float32)
    for k in range(9):
        T = synth.make_T((0.1 * k, 0.01 * k, 0.0), (0, 0, 0.5 * k)).astype(np.float32)
        if k == 7:
            T[0, 3] += 1.5                         # outlier that must trip the 3-sigma gate
        f.add_pose(T)
        zs.append(f.zscore(prev, T))
        outs.append(f.apply(prev, T))
        poses.append(T)
        prev = T
    mtg = orc.map_T_global(np.array([[-22.9068, -43.1729, 12.0], [-22.9069, -43.1728, 12.5], [-22.9067, -43.1730, 11.5]]), np.array([0.3, 0.31, 0.29], np.float32))
    gps = orc.gps_pose(mtg, orc.compass_to_yaw(75.0), -22.90685, -43.17295, 12.2)
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    np.savez_compressed(os.path.join(HERE, "fusion.npz"), latlon=ll, utm_ref=utm_ref,
                        utm_from_reference_build=np.array(orc.ref_lib() is not None),
                        filter_weights=f.weights(), filter_poses=np.array(poses), filter_z=np.array(zs, np.float32), filter_out=np.array(outs),
                        map_T_global=mtg, gps_pose=gps, quat=q, quat_pose=orc.quat_to_pose(q, [1.0, 2.0, 3.0]),
                        gains=np.array(orc.pose_gains(np.diag([0.25, 0.25, 0.25]), np.diag([1e-4] * 6))),
                        bf_x=orc.bf_sequence(1.5, 0.1), bf_z=orc.bf_sequence(0.1, 0.05), bf_yaw=orc.bf_sequence(np.pi / 6.0, np.pi / 18.0))
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
