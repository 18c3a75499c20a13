"""BruteForceAlignment (localization/src/brute_force_alignment.cpp:65-136):
the oracle against an independent numpy/cKDTree restatement, then the HIP path against the
oracle — chosen candidate, transformation (bit-identical) and scores."""
import numpy as np
import pytest
from scipy.spatial import cKDTree


def np_bf(src, tgt, prev, prm):
    """Independent float64 restatement: nesting order x, y, z, yaw; mean squared NN distance;
    first candidate under the threshold returns at once."""
    def seq(rng, step):
        out = []
        i = 0
        while i < np.float32(rng) / (np.float32(2) * np.float32(step)) + 1:
            out += [-i * step, i * step]
            i += 1
        return out
    tree = cKDTree(tgt)
    xs, ys, zs, ws = seq(prm["x_range"], prm["x_step"]), seq(prm["y_range"], prm["y_step"]), seq(prm["z_range"], prm["z_step"]), seq(prm["yaw_range"], prm["yaw_step"])
    best, best_idx, k = np.inf, -1, 0
    for x in xs:
        for y in ys:
            for z in zs:
                for w in ws:
                    L = np.eye(4)
                    L[:2, :2] = [[np.cos(w), -np.sin(w)], [np.sin(w), np.cos(w)]]
                    L[:3, 3] = [x, y, z]
                    T = prev @ L
                    d, _ = tree.query(src @ T[:3, :3].T + T[:3, 3])
                    score = (d ** 2).mean()
                    if score < best:
                        best, best_idx = score, k
                    if score < prm["threshold"]:
                        return True, k, T, score, len(xs) * len(ys) * len(zs) * len(ws)
                    k += 1
    return False, best_idx, None, best, k


PRM = dict(x_step=0.1, y_step=0.1, z_step=0.05, yaw_step=np.pi / 18.0, x_range=0.5, y_range=0.5, z_range=0.1, yaw_range=np.pi / 6.0)


@pytest.fixture(scope="module")
def bf_world(orc, synth, small_world):
    m = small_world["map"]
    T = synth.make_T((0.2, -0.1, 0.05), (0, 0, 10.0))
    scan, _ = synth.make_scan(m, 1500, scan_id=3, T=T)
    prev = synth.make_T((0.01, 0.0, 0.0), (0, 0, 0.5)).astype(np.float32)
    return dict(map=m, scan=scan, T=T, prev=prev)


def test_oracle_bf_matches_independent_numpy(orc, synth, bf_world):
    m, scan, prev = bf_world["map"], bf_world["scan"], bf_world["prev"]
    for thr in (0.001, 1e-9):                                           # hit with early exit / exhaustive miss
        r = orc.bf_align(scan, m, prev, threshold=thr, **PRM)
        found, k, T, score, ncand = np_bf(scan.astype(np.float64), m.astype(np.float64), prev.astype(np.float64), dict(PRM, threshold=thr))
        assert r["n_candidates"] == ncand == 8 * 8 * 4 * 6      # i < range / (2 step) + 1 -> 4 steps, each tried with both signs
        assert r["found"] == found and r["index"] == k
        assert abs(r["best_score"] - score) < 1e-6
        if found:
            assert np.allclose(r["best_T"], T, atol=1e-6)
            assert np.isfinite(r["scores"]).sum() == k + 1                # early exit: later candidates never scored
            assert np.array_equal(r["prev_T_after"], prev)               # previous untouched on a hit
            dt, dr = synth.pose_error(r["best_T"], bf_world["T"])
            assert dt < 0.03 and dr < 0.02                              # the grid point next to the true offset
        else:
            assert np.isfinite(r["scores"]).all()
            assert np.array_equal(r["prev_T_after"], r["best_T"])       # :123 previous <- best


def test_reference_candidate_grid_size(orc):
    # localization_node.cpp:39-43
    n = len(orc.bf_sequence(1.5, 0.1)) ** 2 * len(orc.bf_sequence(0.1, 0.05)) * len(orc.bf_sequence(np.pi / 6.0, np.pi / 18.0))
    assert n == 7776


@pytest.mark.gpu
def test_gpu_bf_matches_oracle(api, ctx, orc, synth, bf_world):
    m, scan, prev = bf_world["map"], bf_world["scan"], bf_world["prev"]
    mp = api.Map(ctx, api.Cloud(ctx, m), 0.25)
    for thr in (0.001, 1e-9):
        bf = api.BruteForceAlignment(ctx)
        bf.setMeanErrorThreshold(thr)
        bf.setXYZStep(PRM["x_step"], PRM["y_step"], PRM["z_step"])
        bf.setXYZRange(PRM["x_range"], PRM["y_range"], PRM["z_range"])
        bf.setRotationStep(PRM["yaw_step"])
        bf.setRotationRange(PRM["yaw_range"])
        bf.setInitialGuess(np.eye(4, dtype=np.float32))                  # trace == 4: accepted ...
        bf.setInitialGuess(prev)                                         # ... still identity -> replaced
        bf.setInitialGuess(np.eye(4, dtype=np.float32) * 2)              # trace != 4 now -> ignored (cpp:44-51)
        bf.setSourceCloud(scan)
        bf.setTargetCloud(mp)
        assert not bf.firstAlignmentCompleted()
        found = bf.alignClouds()
        o = orc.bf_align(scan, m, prev, threshold=thr, **PRM)
        res = bf.last_result()
        assert found == o["found"] and res["index"] == o["index"] and res["n_candidates"] == o["n_candidates"]
        assert np.array_equal(bf.getBestTransformation(), o["best_T"])   # float32 matrix, bit-identical
        done = np.isfinite(o["scores"])
        # the device evaluates a whole x slice per launch; everything the reference scored must agree
        assert np.array_equal(res["scores"][done], o["scores"][done])   # the reference's serial float32 sum and division, bit for bit
        assert bf.firstAlignmentCompleted() == o["found"]
        if not found:                                                    # second call continues from the best pose (:123)
            o2 = orc.bf_align(scan, m, o["prev_T_after"], threshold=thr, **PRM)
            bf.alignClouds()
            assert bf.last_result()["index"] == o2["index"]
            assert np.array_equal(bf.getBestTransformation(), o2["best_T"])


@pytest.mark.gpu
def test_gpu_bf_full_reference_grid(api, ctx, synth, small_world):
    """The node's own configuration (7 776 candidates, localization_node.cpp:39-43) on the
    device; the early exit must land on the grid point nearest the true offset."""
    m = small_world["map"]
    T = synth.make_T((0.3, -0.2, 0.0), (0, 0, -10.0))
    scan, _ = synth.make_scan(m, 4000, scan_id=9, T=T)
    bf = api.BruteForceAlignment(ctx)
    bf.setMeanErrorThreshold(0.001)                                      # tighter than the node's 0.1 so the true cell is found
    bf.setXYZStep(0.1, 0.1, 0.05)
    bf.setXYZRange(1.5, 1.5, 0.1)
    bf.setRotationStep(np.pi / 18.0)
    bf.setRotationRange(np.pi / 6.0)
    bf.setSourceCloud(scan)
    bf.setTargetCloud(m)
    assert bf.alignClouds()
    res = bf.last_result()
    assert res["n_candidates"] == 7776
    dt, dr = synth.pose_error(bf.getBestTransformation(), T)
    assert dt < 0.03 and dr < 0.02
