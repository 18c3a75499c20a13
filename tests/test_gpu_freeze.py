"""Frozen pairs (sf_icp_set_freeze, k_nn_red_fz / k_reduce_solve_fz in sf_icp.hip): once a wide scan's pairs are certified to stay
as they are, the P2PLANE normal equations are evaluated from 96 moments of those pairs -- a polynomial in the pose -- instead
of a pass over the scan; queries close to a change stay active and are evaluated launch by launch.  No reference
counterpart (the reference searches every point in every iteration, localization/src/icp_point_to_point.cpp:64-69).

Checked here: the result equals the launch-by-launch evaluation (same pairs: n_corr and iterations equal; float64 sums in
another order: poses within 1e-11) and the oracle (1e-9, as every P2PLANE parity test); bitwise equal from run to run, under
graph replay and whatever else is in the batch; and the paths around the happy one -- a guard so small that every scan
thaws at once, a guard so large that the active lists overflow, an in-between guard with thousands of active queries,
other first launches -- all give that same result."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_SCAN = 140_000          # above 131 072: two queries per lane, the launch list (never the single-launch kernels)


@pytest.fixture(scope="module")
def world(api, ctx, orc, synth):
    raw = synth.make_map(400_000)
    ds = orc.voxel_pcl(raw, 0.1)[0]
    mp = api.Map(ctx, api.Cloud(ctx, ds), 0.25)
    mp.estimate_normals(0.25)
    scans = np.stack([synth.make_scan(ds, N_SCAN, scan_id=40 + k)[0] for k in range(3)])
    inits = np.stack([np.eye(4), synth.make_T((0.04, -0.03, 0.02), (0.2, -0.1, 0.3)), synth.make_T((-0.05, 0.05, 0.0), (0.0, 0.3, -0.4))])
    return dict(map=ds, mp=mp, scans=scans, inits=inits)


def run(api, ctx, world, freeze, graph=False, params=None, scans=None, inits=None, iters=20):
    icp = api.Icp(ctx, 0.5, iters, 0.05, 1e-5)
    icp.set_target(world["mp"])
    icp.use_graph(graph)
    icp.set_query_order("cell")      # (auto orders a batch by cell and a single scan not: another summation order)
    if freeze is not None:
        icp.set_freeze(freeze)       # True = always; the default ("auto") freezes batches of at least 0.7 M queries only
    if params:
        icp.set_freeze_params(**params)
    icp.set_source_batch(world["scans"] if scans is None else scans)
    icp.set_initial_batch(world["inits"] if inits is None else inits)
    res = icp.align_batch("p2plane")
    stats = icp.freeze_stats()
    icp.close()
    return res, stats


def same_result(a, b, tol=1e-11):
    for x, y in zip(a, b):
        assert x["iterations"] == y["iterations"] and x["n_corr"] == y["n_corr"] and x["flags"] == y["flags"] and x["converged"] == y["converged"]
        assert np.abs(x["T64"] - y["T64"]).max() < tol, np.abs(x["T64"] - y["T64"]).max()
        assert abs(x["rmse"] - y["rmse"]) < 1e-11 and x["fitness"] == y["fitness"]


def bitwise(a, b):
    for x, y in zip(a, b):
        assert np.array_equal(x["T64"], y["T64"]) and x["rmse"] == y["rmse"] and x["n_corr"] == y["n_corr"]


def test_frozen_pairs_equal_the_launch_by_launch_evaluation(api, ctx, synth, world):
    off, s_off = run(api, ctx, world, False)
    on, s_on = run(api, ctx, world, True)
    assert s_off == {"froze": 0, "thawed": 0, "failed": 0, "active_queries": 0, "frozen_at_end": 0}
    assert s_on["froze"] >= 3 and s_on["frozen_at_end"] == 3 and s_on["failed"] == 0          # every scan froze and stayed frozen
    assert 0 <= s_on["active_queries"] < 0.05 * 3 * N_SCAN
    same_result(on, off)
    for r in on:
        assert r["iterations"] == 20 and r["fitness"] > 0.99
    dt, dr = synth.pose_error(on[0]["T64"], synth.t_true())
    assert dt < 5e-4 and dr < 5e-5
    again, _ = run(api, ctx, world, True)
    bitwise(on, again)                                                                        # run to run
    auto, s_auto = run(api, ctx, world, None)
    assert s_auto == s_off                                                                    # 0.42 M queries: a frozen launch would cost more than it saves
    bitwise(auto, off)
    replay, _ = run(api, ctx, world, True, graph=True)
    bitwise(on, replay)                                                                       # graph replay == plain launches
    single, _ = run(api, ctx, world, True, scans=world["scans"][1:2], inits=world["inits"][1:2])
    bitwise(on[1:2], single)                                                                  # whatever else is in the batch


def test_frozen_pairs_against_the_oracle(api, ctx, orc, synth, world):
    normals, _ = world["mp"].download_normals()
    on, stats = run(api, ctx, world, True, scans=world["scans"][1:2], inits=world["inits"][1:2])
    assert stats["frozen_at_end"] == 1
    o = orc.icp_p2plane(world["scans"][1], world["map"], normals, world["inits"][1], 0.5, 20)
    assert on[0]["iterations"] == o["iterations"] == 20
    dt, dr = synth.pose_error(on[0]["T64"], o["T"])
    assert dt < 1e-9 and dr < 1e-9, (dt, dr)


@pytest.mark.parametrize("name,params,expect", [
    ("thaws_at_once", dict(guard_scale=0.0, guard_min=1e-12, guard_max=1e-12), lambda s: s["thawed"] >= 3 and s["frozen_at_end"] == 0),
    ("lists_overflow", dict(guard_scale=0.0, guard_min=0.2, guard_max=0.2), lambda s: s["failed"] >= 3 and s["froze"] == 0),
    ("many_active", dict(guard_scale=0.0, guard_min=1e-3, guard_max=1e-3), lambda s: s["froze"] >= 3 and s["active_queries"] > 2000),
    ("early", dict(from_launch=4, guard_max=1e-3), lambda s: s["froze"] >= 3),
    ("late", dict(from_launch=12), lambda s: s["froze"] >= 3 and s["frozen_at_end"] == 3),
    ("one_try", dict(max_tries=1, guard_scale=0.0, guard_min=1e-12, guard_max=1e-12), lambda s: s["thawed"] == 3),
])
def test_paths_around_the_frozen_one_give_the_same_result(api, ctx, world, name, params, expect):
    off, _ = run(api, ctx, world, False)
    on, stats = run(api, ctx, world, True, params=params)
    assert expect(stats), (name, stats)
    same_result(on, off)
    replay, s2 = run(api, ctx, world, True, graph=True, params=params)
    assert s2 == stats
    bitwise(on, replay)


def test_outliers_and_a_scan_leaving_the_acceptance_radius(api, ctx, synth, world):
    """A third of the scan has no map point within the acceptance radius (rejected pairs, frozen as such), a band sits right
    at it (active: their status can change), and one scan is empty of finite points."""
    rng = np.random.default_rng(5)
    scans = world["scans"].copy()
    n = scans.shape[1]
    scans[0, : n // 3] += np.array([0.0, 0.0, 30.0], dtype=np.float32)          # far above the map
    scans[1, : n // 10, 2] += rng.uniform(0.2, 0.4, n // 10).astype(np.float32)    # around the 0.3 m acceptance radius below
    scans[2, ::5] = np.nan
    icp_args = dict(scans=scans, inits=world["inits"])

    def go(freeze):
        icp = api.Icp(ctx, 0.3, 20, 0.05, 1e-5)
        icp.set_target(world["mp"])
        icp.set_freeze(freeze)
        icp.set_source_batch(icp_args["scans"])
        icp.set_initial_batch(icp_args["inits"])
        r = icp.align_batch("p2plane")
        s = icp.freeze_stats()
        icp.close()
        return r, s
    off, _ = go(False)
    on, stats = go(True)
    assert stats["froze"] >= 2
    same_result(on, off, tol=1e-10)


def test_rows_that_overflow_next_to_rows_that_do_not(api, ctx, synth, world):
    """A third of one scan carries five times the noise: its neighbours are far and its runner-up bounds tight, so with a guard
    of a few millimetres a quarter and more of those queries stay active -- the slab rows that hold them overflow their
    lists while the rows after them do not.  The freeze of that scan must be voided as a whole (a partial list would run into
    the next scan's) and the other scans of the batch, which freeze, must come out as without.  The guard at which exactly
    that happens depends on the data: a few are tried, every one of them must give the launch-by-launch result, and at
    least one must show the mixed outcome."""
    rng = np.random.default_rng(11)
    scans = world["scans"].copy()
    sel = np.nonzero(scans[1, :, 0] < np.quantile(scans[1, :, 0], 0.3))[0]        # cell order: a contiguous stretch of slab rows
    scans[1, sel] += rng.normal(0.0, 0.05, (len(sel), 3)).astype(np.float32)
    off, _ = run(api, ctx, world, False, scans=scans)
    mixed = []
    for guard in (1.0e-3, 1.5e-3, 2.0e-3, 2.5e-3, 3.0e-3):
        on, stats = run(api, ctx, world, True, scans=scans, params=dict(guard_scale=0.0, guard_min=guard, guard_max=guard))
        same_result(on, off, tol=1e-10)
        if stats["failed"] >= 1 and stats["froze"] >= 2:
            mixed.append((guard, stats))
    assert mixed, "no guard gave a scan whose lists overflow next to scans that freeze"


def test_a_prior_that_is_not_quite_rigid(api, ctx, synth, world):
    """sf_icp_set_initial_transformation takes float32 (orthonormal to 6e-8 only) and the node's prior is an element-wise blend
    of two poses (localization_node.cpp:329): the frozen evaluation must hold for any initial pose, not only rigid ones."""
    f32 = np.stack([T.astype(np.float32).astype(np.float64) for T in world["inits"]])
    blend = np.stack([0.8 * T + 0.2 * synth.make_T((0.02, 0.01, -0.01), (0.3, 0.2, -0.5)) @ T for T in world["inits"]])
    for inits in (f32, blend):
        off, _ = run(api, ctx, world, False, inits=inits)
        on, stats = run(api, ctx, world, True, inits=inits)
        assert stats["froze"] >= 3
        same_result(on, off)


def test_the_schedule_learnt_from_the_last_alignment_changes_no_result(api, ctx, synth, world):
    """sf_icp_fetch_results moves the launch at which the NEXT alignment starts asking for a freeze to where this one's scans first
    froze (freeze_learn_schedule): priors 0.08 m / 0.5 degrees off converge a few launches later, so the second alignment of the same object runs
    another launch list -- with the same pairs and, to summation rounding, the same poses as the first and as no freezing at all;
    sf_icp_set_freeze_params pins the launch and switches the learning off."""
    rng = np.random.default_rng(21)
    inits = np.stack([synth.make_T(rng.normal(0, 0.08, 3), rng.normal(0, 0.5, 3)) @ T for T in world["inits"]])
    off, _ = run(api, ctx, world, False, inits=inits, iters=25)
    icp = api.Icp(ctx, 0.5, 25, 0.05, 1e-5)
    icp.set_target(world["mp"])
    icp.set_query_order("cell")
    icp.set_freeze(True)
    icp.set_source_batch(world["scans"])
    icp.set_initial_batch(inits)
    first = icp.align_batch("p2plane")
    s1 = icp.freeze_stats()
    second = icp.align_batch("p2plane")
    s2 = icp.freeze_stats()
    third = icp.align_batch("p2plane")
    assert s1["froze"] >= 3 and s2["froze"] >= 3
    same_result(first, off, tol=1e-10)
    same_result(second, off, tol=1e-10)
    bitwise(second, third)                    # the schedule has settled
    icp.set_freeze_params(from_launch=5)      # explicit: the default launch again, and it stays
    fourth = icp.align_batch("p2plane")
    bitwise(fourth, first)
    fifth = icp.align_batch("p2plane")
    bitwise(fifth, first)
    icp.close()


def test_a_lower_wide_scan_limit_lets_smaller_scans_freeze(api, ctx, synth, world):
    """A full 64-ring scan has at most 130 048 returns: below the 131 072 points the single-launch kernels take.  With an explicit
    limit of 131 072 such scans keep one query per lane and never freeze; with a lower limit (sf_icp_set_wide_scan_points) they
    run wide and freeze: the same pairs, poses equal to float64 rounding.  Without a call the library decides: wide when the
    batch has more rows than any single-launch kernel keeps resident (three scans of 100 000 points: 1 173 rows), not wide for
    one scan alone (391 rows: the single launch takes it)."""
    scans = world["scans"][:, :100_000].copy()

    def go(limit, freeze, batch=3):
        icp = api.Icp(ctx, 0.5, 20, 0.05, 1e-5)
        icp.set_target(world["mp"])
        icp.set_query_order("cell")
        icp.set_freeze(freeze)
        if limit:
            icp.set_wide_scan_points(limit)
        icp.set_source_batch(scans[:batch])
        icp.set_initial_batch(world["inits"][:batch])
        r = icp.align_batch("p2plane")
        s = icp.freeze_stats()
        fused = icp.fused_count()
        icp.close()
        return r, s, fused
    plain, s0, _ = go(131072, True)
    assert s0["froze"] == 0                                   # one query per lane, nothing to freeze
    wide, s1, _ = go(65536, True)
    assert s1["froze"] >= 3 and s1["frozen_at_end"] == 3
    same_result(wide, plain, tol=1e-10)
    auto, s2, f2 = go(None, True)                             # the library's rule: this batch is beyond every single launch -> wide
    assert s2["froze"] >= 3 and f2 == 0
    bitwise(auto, wide)
    one_auto, s3, f3 = go(None, True, batch=1)                # one scan alone: a single launch, one query per lane
    one_plain, _, f4 = go(131072, True, batch=1)
    assert s3["froze"] == 0 and f3 == f4
    bitwise(one_auto, one_plain)
    with pytest.raises(api.SlamFusionError):
        api.Icp(ctx, 0.5, 20, 0.05, 1e-5).set_wide_scan_points(200_000)


def test_a_map_a_kilometre_from_the_origin(api, ctx, orc, synth, world):
    """The moments are taken in un-centred map coordinates: sum r^2 and J^T r are differences of O(N |y|^2) terms, so digits
    go like |y|^2.  Measured with the map 1 km out (1.1 km from the origin): poses still agree with the launch-by-launch
    evaluation to 4e-12 m, the reported rmse to 3e-8 -- pinned here one order above."""
    off = np.array([1000.0, -500.0, 0.0])
    dsm = (world["map"] + off.astype(np.float32)).astype(np.float32)
    mp = api.Map(ctx, api.Cloud(ctx, dsm), 0.25)
    mp.estimate_normals(0.25)
    w = dict(world, mp=mp)
    inits = np.stack([synth.make_T(off, (0, 0, 0)) @ T for T in world["inits"]])
    off_, _ = run(api, ctx, w, False, inits=inits)
    on_, stats = run(api, ctx, w, True, inits=inits)
    assert stats["froze"] >= 3
    for a, b in zip(on_, off_):
        assert a["iterations"] == b["iterations"] and a["n_corr"] == b["n_corr"]
        assert np.abs(a["T64"] - b["T64"]).max() < 5e-11 and abs(a["rmse"] - b["rmse"]) < 5e-7


@pytest.mark.parametrize("margin,want_resume", [(1.0, False), (0.02, True)])
def test_sharded_ranks_freeze_their_own_queries(api, ctx, synth, world, margin, want_resume):
    """Three x-slabs of the map held by three sf_icp objects stepping in lockstep (sf_icp_align_group: the C side's sharded loop
    with a fixed-order device sum as the collective).  Every rank freezes ITS OWN owned queries -- ownership (x inside the
    slab) is part of what must not change within the guard -- and contributes the same record as before; a scan that moves
    beyond the margin goes stale, is rebuilt, resumed and freezes again."""
    from slam_sensor_fusion_amd import sharded
    ds = world["map"]
    edges = sharded.slab_edges(ds[:, 0], 3)
    off, _ = run(api, ctx, world, False)
    members = []
    for r in range(3):
        keep = sharded.slab_select(ds, edges, r, halo=0.5 + 0.25 + 0.25)
        mp = api.Map(ctx, api.Cloud(ctx, ds[keep]), 0.25)
        mp.estimate_normals(0.25)
        icp = api.Icp(ctx, 0.5, 20, 0.05, 1e-5)
        icp.set_target(mp)
        icp.set_freeze(True)
        icp.set_source_batch(world["scans"])
        icp.set_initial_batch(world["inits"])
        icp.set_shard(float(max(edges[r], -1e30)), float(min(edges[r + 1], 1e30)))
        icp.set_shard_margin(margin)
        members.append(icp)
    res, resumes = api.align_group(members, "p2plane")
    assert (resumes > 0) == want_resume, resumes
    stats = [m.freeze_stats() for m in members]
    assert all(s["froze"] >= 1 for s in stats), stats
    for a, b in zip(res, off):
        assert a["iterations"] == b["iterations"] == 20 and a["n_corr"] == b["n_corr"] and a["flags"] == 0
        dt, dr = synth.pose_error(a["T64"], b["T64"])
        assert dt < 1e-9 and dr < 1e-10, (dt, dr)
    again, _ = api.align_group(members, "p2plane")
    bitwise(res, again)
    for m in members:
        m.set_freeze(False)
    plain, _ = api.align_group(members, "p2plane")
    same_result(res, plain, tol=1e-10)
    for m in members:
        m.close()
