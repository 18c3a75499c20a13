"""The C-side sharded loop (sf_icp_align_sharded) with MORE THAN ONE RANK on a one-GPU box: 2 and 4 fresh child
processes on device 0, each holding one x-slab (+ halo) of the map, all-reducing the normal-equation records through
the hand-written P2P transport (sf_comm_p2p_*: hipIpc exchange regions, store-and-flag, fixed rank-order sum --
SURVEY.md §8e option ii; RCCL refuses two ranks on one device).  No reference counterpart: the reference is one CPU
process (localization/src/main.cpp:18).

Checked: results equal the unsharded alignment (1e-9 m, 1e-10 rad; iteration and correspondence counts equal), bitwise
equal across ranks and from run to run, the stale -> resume path, routed sub-group communicators, and that a rank which
dies or aborts makes every other rank fail with SF_ERR_COMM and a non-zero exit instead of waiting in a collective.
The children are started with subprocess from a parent that only waits for them (never an exec of a process that has
touched the GPU); at most 4 children + this process use the device at a time."""
import os
import subprocess
import sys
import uuid

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

MAX_DIST, NORMAL_RADIUS, CELL = 0.5, 0.25, 0.25
WORKER = os.path.join(ROOT, "tests", "p2p_rank_worker.py")


def run_ranks(tmp_path, world, case, ds, scans, inits, expect_rc=None, timeout=240):
    d = str(tmp_path)
    np.save(os.path.join(d, "map.npy"), ds)
    np.save(os.path.join(d, "scans.npy"), scans)
    np.save(os.path.join(d, "inits.npy"), inits)
    with open(os.path.join(d, "tag.txt"), "w") as f:
        f.write(uuid.uuid4().hex[:12])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, WORKER, d, str(r), str(world), case], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    logs, rcs = [], []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
            o += "\n[killed by the test after %d s]" % timeout
        logs.append(o)
        rcs.append(p.returncode)
    want = [0] * world if expect_rc is None else expect_rc
    assert rcs == want, "exit codes %r, expected %r\n%s" % (rcs, want, "\n".join("--- rank %d ---\n%s" % (r, l[-3000:]) for r, l in enumerate(logs)))
    return [dict(np.load(os.path.join(d, "out_rank%d.npz" % r))) for r in range(world) if os.path.exists(os.path.join(d, "out_rank%d.npz" % r))]


def unsharded(api, ctx, ds, scans, inits, mode, iters):
    mp = api.Map(ctx, api.Cloud(ctx, ds), CELL)
    mp.estimate_normals(NORMAL_RADIUS)
    icp = api.Icp(ctx, MAX_DIST, iters, 0.05, 1e-5)
    icp.set_target(mp)
    icp.set_source_batch(scans)
    icp.set_initial_batch(inits)
    return icp.align_batch(mode)


@pytest.mark.parametrize("world", [2, 4])
def test_c_side_sharded_loop_with_p2p_ranks_on_one_gpu(api, ctx, synth, small_world, tmp_path, world):
    ds = small_world["map"]
    scans = np.stack([synth.make_scan(ds, 6000, scan_id=s)[0] for s in range(3)])
    # scan 1 starts 0.45 m off along x: crosses a 0.2 m margin (stale -> rebuilt -> resumed), not a 1 m one
    inits = np.stack([np.eye(4), synth.make_T((0.45, 0.0, 0.0), (0, 0, 0)), synth.make_T((0.0, 0.1, 0.0), (0, 0, 0.05))])
    outs = run_ranks(tmp_path, world, "parity", ds, scans, inits)
    assert len(outs) == world
    for name, mode, iters in (("p2plane_m1", "p2plane", 20), ("p2plane_m02", "p2plane", 20), ("o3d", "o3d_p2p", 30)):
        ref = unsharded(api, ctx, ds, scans, inits, mode, iters)
        for rep in range(2):
            k = "%s_r%d_" % (name, rep)
            for r in range(1, world):                                          # bitwise equal across ranks ...
                assert np.array_equal(outs[r][k + "T"], outs[0][k + "T"]), (name, rep, r)
                assert np.array_equal(outs[r][k + "n_corr"], outs[0][k + "n_corr"])
            assert np.array_equal(outs[0][k + "T"], outs[0]["%s_r0_T" % name])  # ... and from run to run
            assert (outs[0][k + "flags"] == 0).all()
            for b in range(len(scans)):
                assert outs[0][k + "iterations"][b] == ref[b]["iterations"] and outs[0][k + "n_corr"][b] == ref[b]["n_corr"], (name, b)
                assert outs[0][k + "converged"][b] == int(ref[b]["converged"])
                dt, dr = synth.pose_error(outs[0][k + "T"][b], ref[b]["T64"])
                assert dt < 1e-9 and dr < 1e-10, (name, rep, b, dt, dr)
            owned = np.stack([o[k + "owned"] for o in outs])
            resumes = int(outs[0][k + "resumes"])
            assert all(int(o[k + "resumes"]) == resumes for o in outs)          # the same decisions on every rank
            if name == "p2plane_m02":
                assert resumes > 0
            else:
                assert resumes == 0
                assert (owned.sum(0) >= scans.shape[1]).all()                   # every query is somebody's candidate
                assert (owned.max(0) < scans.shape[1]).all()                    # and no rank walks a whole scan


def test_ranks_freeze_their_own_queries_over_p2p(api, ctx, orc, synth, tmp_path):
    """Two processes, wide scans (140 k points: two queries per lane), sf_icp_set_freeze(always): each rank's reduce kernel folds
    ITS frozen pairs' moments into the record it publishes, the gather kernel keeps the freeze state; result == the
    unsharded launch-by-launch evaluation (1e-9), bitwise equal across ranks and runs, and every rank did freeze."""
    world = 2
    raw = synth.make_map(400_000)
    ds = orc.voxel_pcl(raw, 0.1)[0]
    scans = np.stack([synth.make_scan(ds, 140_000, scan_id=60 + k)[0] for k in range(2)])
    inits = np.stack([np.eye(4), synth.make_T((0.03, -0.02, 0.01), (0.1, 0.0, 0.2))])
    outs = run_ranks(tmp_path, world, "frozen", ds, scans, inits)
    assert len(outs) == world
    mp = api.Map(ctx, api.Cloud(ctx, ds), CELL)
    mp.estimate_normals(NORMAL_RADIUS)
    icp = api.Icp(ctx, MAX_DIST, 20, 0.05, 1e-5)
    icp.set_target(mp)
    icp.set_freeze(False)
    icp.set_source_batch(scans)
    icp.set_initial_batch(inits)
    ref = icp.align_batch("p2plane")
    for r in range(world):
        assert int(outs[r]["frozen_froze"]) >= 2 and int(outs[r]["frozen_at_end"]) == 2, r
        assert np.array_equal(outs[r]["frozen_r0_T"], outs[0]["frozen_r0_T"]) and np.array_equal(outs[r]["frozen_r1_T"], outs[0]["frozen_r0_T"])
    for b in range(2):
        assert outs[0]["frozen_r0_iterations"][b] == ref[b]["iterations"] == 20 and outs[0]["frozen_r0_n_corr"][b] == ref[b]["n_corr"]
        dt, dr = synth.pose_error(outs[0]["frozen_r0_T"][b], ref[b]["T64"])
        assert dt < 1e-9 and dr < 1e-10, (b, dt, dr)


def test_many_scans_in_flight_with_four_ranks_on_one_gpu(api, ctx, synth, small_world, tmp_path):
    """320 scans per rank: three peers' gather kernels (one wave per scan, spinning until every rank has published) put a
    wave on every compute unit of the device while this rank's publish kernel still has to be placed.  With the publish
    kernel on 1024 threads at 122 VGPRs -- a whole unit's register file per workgroup -- it never was (every rank timed out;
    found with bench.py's default batch on 4 ranks); it runs on 256 threads."""
    ds = small_world["map"]
    world, B = 4, 320
    scans = np.stack([synth.make_scan(ds, 1500, scan_id=100 + s)[0] for s in range(B)])
    inits = np.stack([np.eye(4)] * B)
    outs = run_ranks(tmp_path, world, "crowd", ds, scans, inits, timeout=200)
    assert len(outs) == world
    for r in range(1, world):
        assert np.array_equal(outs[r]["crowd_T"], outs[0]["crowd_T"])
    assert (outs[0]["crowd_iterations"] == 8).all() and (outs[0]["crowd_flags"] == 0).all()
    ref = unsharded(api, ctx, ds, scans[:4], inits[:4], "p2plane", 8)
    for b in range(4):
        dt, dr = synth.pose_error(outs[0]["crowd_T"][b], ref[b]["T64"])
        assert dt < 1e-9 and dr < 1e-10 and outs[0]["crowd_n_corr"][b] == ref[b]["n_corr"]


def test_routed_subgroups_over_p2p(api, ctx, orc, synth, tmp_path):
    """Four slabs of a 20 m map; scans inside one slab (no collective), straddling two and three slabs (communicators
    of just those ranks) and over the whole map (all four)."""
    world = 4
    raw = synth.make_map(400_000)
    ds = orc.voxel_pcl(raw, 0.1)[0]
    from slam_sensor_fusion_amd import sharded
    edges = sharded.slab_edges(ds[:, 0], world)
    x = ds[:, 0]

    def scan_in(x0, x1, sid):
        return synth.make_scan(ds[(x >= x0) & (x < x1)], 5000, scan_id=sid)[0]
    bands = [(-10.0, edges[1] - 1.6), (edges[1] - 1.2, edges[2] - 1.6), (edges[1] - 1.5, edges[3] - 1.6), (edges[3] + 1.6, 10.0), (-10.0, 10.0), (edges[2] - 1.0, edges[2] + 1.0)]
    scans = np.stack([scan_in(a, b, 10 + i) for i, (a, b) in enumerate(bands)])
    inits = np.stack([np.eye(4)] * len(scans))
    lo, hi = api.shard_route(scans, inits, edges, 1.0)
    assert [(int(a), int(b)) for a, b in zip(lo, hi)] == [(0, 0), (0, 1), (0, 2), (3, 3), (0, 3), (1, 2)]
    outs = run_ranks(tmp_path, world, "routed", ds, scans, inits)
    ref = unsharded(api, ctx, ds, scans, inits, "p2plane", 20)
    seen = {}
    for r, o in enumerate(outs):
        want_ids = [b for b in range(len(scans)) if lo[b] <= r <= hi[b]]
        assert list(o["ids"]) == want_ids, (r, list(o["ids"]))                 # a rank only ever sees the scans routed to it
        assert int(o["n_comms"]) == len({(int(lo[b]), int(hi[b])) for b in want_ids if lo[b] != hi[b]})
        for j, b in enumerate(o["ids"]):
            seen.setdefault(int(b), []).append((r, o["routed_T"][j], int(o["routed_iterations"][j]), int(o["routed_n_corr"][j])))
    assert sorted(seen) == list(range(len(scans)))
    for b, lst in seen.items():
        for r, T, it, nc in lst:
            assert np.array_equal(T, lst[0][1])                                 # members of a group agree bit for bit
            assert it == ref[b]["iterations"] and nc == ref[b]["n_corr"], (b, r)
            dt, dr = synth.pose_error(T, ref[b]["T64"])
            assert dt < 1e-9 and dr < 1e-10, (b, r, dt, dr)


@pytest.mark.parametrize("case", ["peer_dies", "peer_aborts"])
def test_a_lost_rank_fails_every_rank_instead_of_hanging(synth, small_world, tmp_path, case):
    """Rank 2 of 3 leaves (silently / after sf_comm_abort) between two alignments: ranks 0 and 1 get SF_ERR_COMM -- through
    the 3 s time limit of the flag wait, or at once through the abort word -- and exit non-zero; the communicator stays
    refused afterwards."""
    ds = small_world["map"]
    scans = np.stack([synth.make_scan(ds, 4000, scan_id=s)[0] for s in range(2)])
    inits = np.stack([np.eye(4)] * 2)
    run_ranks(tmp_path, 3, case, ds, scans, inits, expect_rc=[3, 3, 7 if case == "peer_aborts" else 0], timeout=120)
    for r in range(2):
        f = np.load(os.path.join(str(tmp_path), "fail_rank%d.npz" % r))
        assert int(f["ok"]) == 1
        if case == "peer_aborts":
            assert float(f["took"]) < 3.0                                       # the abort word, not the time limit
