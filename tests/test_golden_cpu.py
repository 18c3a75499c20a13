"""The oracle against the committed golden vectors (tests/golden/make_golden.py): freezes
the checker so a later edit of oracle/ cannot silently move the target."""
import numpy as np

from conftest import load_golden


def test_oracle_reproduces_golden_registration(orc, synth):
    g = load_golden("registration_small.npz")
    ds, vidx, ovox, st = orc.voxel_pcl(g["raw"], 0.1)
    assert np.array_equal(ds, g["map"]) and np.array_equal(vidx, g["vox_point_ids"]) and np.array_equal(ovox, g["vox_out_ids"])
    fin = g["raw"][np.isfinite(g["raw"]).all(1)]
    means, ijk, oijk, _ = orc.voxel_o3d(fin.astype(np.float64), 0.1)
    assert np.array_equal(means, g["o3d_means"]) and np.array_equal(ijk, g["o3d_point_ijk"]) and np.array_equal(oijk, g["o3d_out_ijk"])
    idx, d2 = orc.KdTreeF(ds).nn(g["nn_queries"])
    assert np.array_equal(idx, g["nn_idx"]) and np.array_equal(d2, g["nn_d2"])
    nrm, cnt = orc.normals_radius(ds, 0.3)
    assert np.array_equal(cnt, g["normal_counts"]) and np.allclose(nrm, g["normals"], atol=1e-6)
    r = orc.icp_ref_cpp(g["scan"], ds, precise=False)
    assert np.allclose(r["T"], g["ref32_T"], atol=1e-6) and list(g["ref32_meta"]) == [r["iterations"], r["converged"], r["n_corr"], r["n_research"]]
    r = orc.icp_ref_cpp(g["scan"], ds, precise=True)
    assert np.allclose(r["T"], g["ref64_T"], atol=1e-12)
    r = orc.icp_o3d_p2p(g["scan"], ds, max_iter=30)
    assert np.allclose(r["T"], g["o3d_T"], atol=1e-12) and r["iterations"] == g["o3d_meta"][0]
    r = orc.icp_p2plane(g["scan"], ds, g["normals"], num_iters=20)
    assert np.allclose(r["T"], g["pl_T"], atol=1e-12)
    assert np.array_equal(orc.crop_radius(ds, [0.3, -0.2, 0.1], 0.8)[1], g["crop_radius_idx"])
    assert np.array_equal(orc.crop_aabb(ds, [0, -0.5, 0], [1.0, 0.5, 0.5])[1], g["crop_aabb_idx"])
    assert np.array_equal(orc.crop_obb(ds, [0.1, 0.2, 0.0], g["obb_R"], [1.5, 0.8, 0.8])[1], g["crop_obb_idx"])
    # the golden registration itself recovers the generating transform
    dt, dr = synth.pose_error(g["o3d_T"], synth.t_true())
    assert dt < 5e-3 and dr < 2e-3


def test_oracle_reproduces_golden_extensions(orc, synth):
    g, e = load_golden("registration_small.npz"), load_golden("extensions_small.npz")
    ds, pid, oid = orc.voxel_pcl64(e["far"], 0.1)
    assert np.array_equal(ds, e["far_ds"]) and np.array_equal(pid, e["far_point_ids"]) and np.array_equal(oid, e["far_out_ids"]) and oid.max() > 2**31
    _, _, cov = orc.normals_radius_cov(g["map"], float(np.float32(0.3)))
    assert np.allclose(cov, e["cov6"], rtol=1e-12, atol=1e-18)
    prm = dict(x_step=0.1, y_step=0.1, z_step=0.05, yaw_step=np.pi / 18.0, x_range=0.3, y_range=0.3, z_range=0.1, yaw_range=np.pi / 6.0)
    bf = orc.bf_align(e["bf_scan"], g["map"], e["bf_prev"], threshold=1e-9, **prm)
    assert np.array_equal(bf["scores"], e["bf_scores"]) and np.array_equal(bf["best_T"], e["bf_best_T"]) and [bf["index"], bf["n_candidates"]] == list(e["bf_index"])
