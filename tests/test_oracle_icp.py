"""Pins the oracle's ICP drivers against INDEPENDENT restatements written with
scipy.spatial.cKDTree + numpy.linalg.svd (no shared code with oracle/), and against
analytic known answers.  Upstream parity is unpinned (no reference fixtures)."""
import numpy as np
import pytest
from scipy.spatial import cKDTree


def np_kabsch(src, tgt):
    cs, ct = src.mean(0), tgt.mean(0)
    H = (src - cs).T @ (tgt - ct)
    U, S, Vt = np.linalg.svd(H)
    V = Vt.T
    R = V @ U.T
    if np.linalg.det(R) < 0:
        V[:, 2] *= -1
        R = V @ U.T
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = ct - R @ cs
    return T


def apply(T, p):
    return p @ T[:3, :3].T + T[:3, 3]


def np_icp_ref_cpp(src, tgt, init, max_corr, iters, accept, eps):
    """icp_point_to_point.cpp:185-254 in float64 with cKDTree."""
    tree = cKDTree(tgt)

    def correspond(x):
        d, i = tree.query(x)
        keep = d ** 2 < max_corr                       # squared vs un-squared quirk (:70)
        return x[keep], tgt[i[keep]]
    x, y = correspond(apply(init, src))
    if len(x) < 10:
        return init, 1e6, 0, False
    T, last, taken = init.copy(), np.finfo(np.float32).max, 0
    for _ in range(iters):
        err = np.linalg.norm(x - y, axis=1).mean()
        if err < accept:
            last = err
            break
        if abs(last - err) < eps:
            x, y = correspond(x)
        step = np_kabsch(x, y)
        T = step @ T
        x = apply(step, x)
        last = err
        taken += 1
    return T, last, taken, last < accept


def np_icp_o3d(src, tgt, init, max_dist, max_iter):
    """open3d RegistrationICP, point-to-point, relative criteria 1e-6 (localization_node.py:233-237)."""
    tree = cKDTree(tgt)

    def evaluate(p):
        d, i = tree.query(p)
        keep = d ** 2 < max_dist ** 2
        n = keep.sum()
        return keep, i, (n / len(p), np.sqrt((d[keep] ** 2).sum() / n) if n else 0.0)
    T = init.copy()
    p = apply(T, src)
    keep, i, (fit, rmse) = evaluate(p)
    it = 0
    for _ in range(max_iter):
        upd = np_kabsch(p[keep], tgt[i[keep]])
        T = upd @ T
        p = apply(upd, p)
        it += 1
        pf, pr = fit, rmse
        keep, i, (fit, rmse) = evaluate(p)
        if abs(pf - fit) < 1e-6 and abs(pr - rmse) < 1e-6:
            break
    return T, it, fit, rmse


def test_svd3_against_numpy(orc):
    rng = np.random.default_rng(0)
    for k in range(50):
        A = rng.normal(size=(3, 3))
        if k % 10 == 0:
            A[:, 2] = A[:, 0] * 0.5                        # rank 2
        U, S, V = orc.svd3(A)
        assert np.allclose(U @ np.diag(S) @ V.T, A, atol=1e-12)
        assert np.allclose(S, np.linalg.svd(A, compute_uv=False), atol=1e-12)
        assert np.allclose(U.T @ U, np.eye(3), atol=1e-10) and np.allclose(V.T @ V, np.eye(3), atol=1e-12)
    Uf, Sf, Vf = orc.svd3(rng.normal(size=(3, 3)).astype(np.float32), np.float32)
    assert Sf[0] >= Sf[1] >= Sf[2]


def test_kabsch_recovers_exact_rigid_transform_in_one_step(orc, synth):
    rng = np.random.default_rng(1)
    src = rng.uniform(-5, 5, (500, 3)).astype(np.float32)
    T = synth.make_T((0.3, -0.2, 0.1), (2.0, -3.0, 10.0))
    tgt = apply(T, src.astype(np.float64)).astype(np.float32)
    for precise, tol in ((True, 2e-6), (False, 2e-5)):
        Tk = orc.kabsch(src, tgt, precise)
        dt, dr = synth.pose_error(Tk, T)
        assert dt < tol and dr < tol
    assert np.allclose(orc.kabsch(src, tgt, True), np_kabsch(src.astype(np.float64), tgt.astype(np.float64)), atol=1e-9)
    # reflection case: a mirrored cloud must still give a proper rotation (det +1, cpp:145-149)
    mirrored = tgt.copy()
    mirrored[:, 2] *= -1
    R = orc.kabsch(src, mirrored, True)[:3, :3]
    assert abs(np.linalg.det(R) - 1) < 1e-9


def test_ref_cpp_matches_independent_numpy_restatement(orc, small_world, synth):
    m, scan = small_world["map"], small_world["scan"]
    cases = [dict(max_corr=0.5, iters=10, accept=0.05, eps=1e-5),       # localization_node.cpp:24-28
             dict(max_corr=0.5, iters=10, accept=0.001, eps=1e-5),      # never accepts: runs all iterations
             dict(max_corr=0.5, iters=15, accept=0.001, eps=5e-3),      # forces lazy re-searches
             dict(max_corr=5.0, iters=80, accept=0.4, eps=1e-2)]        # "strong" fallback, :226-229
    for c in cases:
        Tn, en, itn, cn = np_icp_ref_cpp(scan.astype(np.float64), m.astype(np.float64), np.eye(4), c["max_corr"], c["iters"], c["accept"], c["eps"])
        r = orc.icp_ref_cpp(scan, m, None, c["max_corr"], c["iters"], c["accept"], c["eps"], precise=True)
        assert r["iterations"] == itn and r["converged"] == cn
        dt, dr = synth.pose_error(r["T"], Tn)
        assert dt < 1e-9 and dr < 1e-9
        assert abs(r["error"] - en) < 1e-9
        r32 = orc.icp_ref_cpp(scan, m, None, c["max_corr"], c["iters"], c["accept"], c["eps"], precise=False)
        dt, dr = synth.pose_error(r32["T"], Tn)
        assert dt < 5e-4 and dr < 5e-5                                 # the reference's own float32 spread
    r = orc.icp_ref_cpp(scan, m, None, 0.5, 15, 0.001, 5e-3, precise=True)
    assert r["n_research"] >= 1


def test_ref_cpp_too_few_correspondences_returns_initial(orc, small_world):
    m = small_world["map"]
    far = small_world["scan"][:50] + np.float32(500.0)
    init = np.eye(4, dtype=np.float32)
    init[0, 3] = 0.25
    r = orc.icp_ref_cpp(far, m, init)
    assert r["iterations"] == 0 and not r["converged"] and r["error"] == 1e6   # ICPResult defaults, h:28-39
    assert np.array_equal(r["T"], init.astype(np.float64))


def test_o3d_p2p_matches_independent_numpy_restatement(orc, small_world, synth):
    m, scan = small_world["map"], small_world["scan"][:4000]
    for init in (np.eye(4), synth.make_T((0.05, 0.02, -0.01), (0.1, 0.0, 0.3))):
        Tn, itn, fit, rmse = np_icp_o3d(scan.astype(np.float64), m.astype(np.float64), init, 0.5, 30)
        r = orc.icp_o3d_p2p(scan, m, init, 0.5, 30)
        assert r["iterations"] == itn
        dt, dr = synth.pose_error(r["T"], Tn)
        assert dt < 1e-9 and dr < 1e-9
        assert abs(r["fitness"] - fit) < 1e-12 and abs(r["error"] - rmse) < 1e-9
    # the registration recovers the generating transform to the noise floor
    dt, dr = synth.pose_error(r["T"], synth.t_true())
    assert dt < 2e-3 and dr < 2e-4


def test_p2plane_recovers_planar_scene(orc, synth):
    rng = np.random.default_rng(4)
    # three orthogonal noisy planes: every degree of freedom is constrained
    a = rng.uniform(-4, 4, (9000, 2))
    planes = np.concatenate([np.c_[a[:3000], np.zeros(3000)], np.c_[a[3000:6000, 0], np.zeros(3000), a[3000:6000, 1]], np.c_[np.zeros(3000), a[6000:]]])
    tgt = planes.astype(np.float32)
    nrm, cnt = orc.normals_radius(tgt, 0.4)
    T = synth.make_T((0.05, -0.04, 0.03), (0.5, -0.4, 0.8))
    src = apply(np.linalg.inv(T), tgt[::3].astype(np.float64)).astype(np.float32)
    r = orc.icp_p2plane(src, tgt, nrm, None, 0.5, 20)
    dt, dr = synth.pose_error(r["T"], T)
    assert r["iterations"] == 20 and dt < 1e-4 and dr < 1e-4
