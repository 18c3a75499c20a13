"""The N > 1 host path on CPU: two gloo ranks, the map split into x-slabs with halo by
slam_sensor_fusion_amd.sharded, every rank accumulating only the queries it owns, the
30-double record all-reduced once per iteration, identical solve on both ranks
(SURVEY.md §8e).  The device step is replaced by a numpy stand-in with the same
step_begin / step_end contract (NN through the oracle's kd-tree — test infrastructure), so
what is exercised is the product's sharding logic and collective flow; the result must
equal the unsharded oracle registration."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

MAX_DIST, NORMAL_RADIUS, ITERS = 0.5, 0.25, 8


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class NumpyShardStep:
    """Stand-in for sf_icp on one rank: same step contract, numpy arithmetic."""

    def __init__(self, orc, map_pts, normals, scan, x_lo, x_hi, xchg):
        self.tree = orc.KdTreeD(map_pts.astype(np.float64))
        self.map, self.nrm, self.scan = map_pts.astype(np.float64), normals.astype(np.float64), scan.astype(np.float64)
        self.lo, self.hi, self.xchg = x_lo, x_hi, xchg
        self.T = np.eye(4)
        self.iterations = 0

    def fetch_results(self):
        return [dict(flags=0, iterations=self.iterations)]

    def step_begin(self, mode, first):
        if first == 1:
            self.T = np.eye(4)
            self.iterations = 0
        s = self.scan @ self.T[:3, :3].T + self.T[:3, 3]
        own = (s[:, 0].astype(np.float32) >= np.float32(self.lo)) & (s[:, 0].astype(np.float32) < np.float32(self.hi))
        idx, d2 = self.tree.nn(s[own])
        ok = (idx >= 0) & (d2 < MAX_DIST ** 2)
        s, t, n = s[own][ok], self.map[idx[ok]], self.nrm[idx[ok]]
        r = ((s - t) * n).sum(1)
        J = np.c_[np.cross(s, n), n]
        rec = np.zeros(32)
        rec[0] = len(s)
        rec[1] = (r * r).sum()
        JtJ = J.T @ J
        rec[2:23] = JtJ[np.triu_indices(6)]
        rec[23:29] = J.T @ r
        rec[29] = d2[ok].sum()
        self.xchg.copy_(torch.from_numpy(rec))

    def step_end(self, mode, last):
        rec = self.xchg.numpy()
        A = np.zeros((6, 6))
        A[np.triu_indices(6)] = rec[2:23]
        A = A + A.T - np.diag(np.diag(A))
        x = np.linalg.solve(A, -rec[23:29])
        ca, sa, cb, sb, cg, sg = np.cos(x[0]), np.sin(x[0]), np.cos(x[1]), np.sin(x[1]), np.cos(x[2]), np.sin(x[2])
        upd = np.eye(4)
        upd[:3, :3] = [[cg * cb, cg * sb * sa - sg * ca, cg * sb * ca + sg * sa],
                       [sg * cb, sg * sb * sa + cg * ca, sg * sb * ca - cg * sa],
                       [-sb, cb * sa, cb * ca]]
        upd[:3, 3] = x[3:]
        self.T = upd @ self.T
        self.iterations += 1
        self.n_corr = int(rec[0])


def worker_pipelined(rank, world, port, out_dir):
    """Two scans as two pipelined parts (sharded.PipelinedShardedIcp): collectives interleave A, B, A, B ..."""
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    from slam_sensor_fusion_amd import sharded, synth
    raw = synth.make_map(60_000)
    ds = orc.voxel_pcl(raw, 0.1)[0]
    edges = sharded.slab_edges(ds[:, 0], world)
    keep = sharded.slab_select(ds, edges, rank, halo=MAX_DIST + NORMAL_RADIUS + 0.25)
    local = ds[keep]
    normals, _ = orc.normals_radius(local, NORMAL_RADIUS)
    parts, steps = [], []
    for k in range(2):
        scan = synth.make_scan(ds, 2000, scan_id=10 + k)[0]
        xchg = torch.zeros(32, dtype=torch.float64)
        st = NumpyShardStep(orc, local, normals, scan, edges[rank], edges[rank + 1], xchg)
        steps.append(st)
        parts.append((st, (lambda x=xchg: dist.all_reduce(x))))
    drv = sharded.PipelinedShardedIcp(parts, "p2plane", ITERS)
    res = drv.align()
    assert len(res) == 2 and all(r["iterations"] == ITERS for r in res) and drv.resumes == 0
    for k, st in enumerate(steps):
        np.save(os.path.join(out_dir, "P%d_T_%d.npy" % (k, rank)), st.T)
    dist.destroy_process_group()


def test_two_rank_pipelined_parts_equal_unsharded(orc, synth, tmp_path):
    world = 2
    mp.spawn(worker_pipelined, args=(world, free_port(), str(tmp_path)), nprocs=world, join=True)
    raw = synth.make_map(60_000)
    ds = orc.voxel_pcl(raw, 0.1)[0]
    normals, _ = orc.normals_radius(ds, NORMAL_RADIUS)
    for k in range(2):
        T0, T1 = np.load(tmp_path / ("P%d_T_0.npy" % k)), np.load(tmp_path / ("P%d_T_1.npy" % k))
        assert np.array_equal(T0, T1)
        scan = synth.make_scan(ds, 2000, scan_id=10 + k)[0]
        ref = orc.icp_p2plane(scan, ds, normals, None, MAX_DIST, ITERS)
        dt, dr = synth.pose_error(T0, ref["T"])
        assert dt < 1e-9 and dr < 1e-9


def worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    from slam_sensor_fusion_amd import sharded, synth
    raw = synth.make_map(60_000)
    ds = orc.voxel_pcl(raw, 0.1)[0]
    scan = synth.make_scan(ds, 3000)[0]
    edges = sharded.slab_edges(ds[:, 0], world)
    keep = sharded.slab_select(ds, edges, rank, halo=MAX_DIST + NORMAL_RADIUS + 0.25)
    local = ds[keep]
    normals, _ = orc.normals_radius(local, NORMAL_RADIUS)
    xchg = torch.zeros(32, dtype=torch.float64)
    step = NumpyShardStep(orc, local, normals, scan, edges[rank], edges[rank + 1], xchg)
    drv = sharded.ShardedIcp(step, "p2plane", ITERS, lambda: dist.all_reduce(xchg))
    assert drv.n_steps() == ITERS
    res = drv.align()
    assert res[0]["iterations"] == ITERS and drv.resumes == 0
    np.save(os.path.join(out_dir, "T_%d.npy" % rank), step.T)
    np.save(os.path.join(out_dir, "n_%d.npy" % rank), np.array([step.n_corr, len(local), step.iterations]))
    dist.destroy_process_group()


def test_two_rank_sharded_registration_equals_unsharded(orc, synth, tmp_path):
    world = 2
    mp.spawn(worker, args=(world, free_port(), str(tmp_path)), nprocs=world, join=True)
    T0, T1 = np.load(tmp_path / "T_0.npy"), np.load(tmp_path / "T_1.npy")
    n0, n1 = np.load(tmp_path / "n_0.npy"), np.load(tmp_path / "n_1.npy")
    assert np.array_equal(T0, T1)                                     # identical solve on every rank
    raw = synth.make_map(60_000)
    ds = orc.voxel_pcl(raw, 0.1)[0]
    scan = synth.make_scan(ds, 3000)[0]
    assert n0[1] < len(ds) and n1[1] < len(ds) and n0[1] + n1[1] > len(ds)   # slabs + halo overlap
    normals, _ = orc.normals_radius(ds, NORMAL_RADIUS)
    ref = orc.icp_p2plane(scan, ds, normals, None, MAX_DIST, ITERS)
    assert n0[0] == ref["n_corr"] and n0[2] == ITERS                  # every query owned by exactly one rank
    dt, dr = synth.pose_error(T0, ref["T"])
    assert dt < 1e-9 and dr < 1e-9


def test_slab_partition_properties():
    from slam_sensor_fusion_amd import sharded
    rng = np.random.default_rng(0)
    pts = rng.uniform(-50, 50, (100_000, 3)).astype(np.float32)
    for world in (1, 2, 4, 8):
        edges = sharded.slab_edges(pts[:, 0], world)
        assert len(edges) == world + 1 and np.isinf(edges[0]) and np.isinf(edges[-1])
        core = [((pts[:, 0] >= edges[r]) & (pts[:, 0] < edges[r + 1])).sum() for r in range(world)]
        assert sum(core) == len(pts) and max(core) - min(core) <= 2   # equal-count, disjoint, complete
        for r in range(world):
            sel = sharded.slab_select(pts, edges, r, halo=1.0)
            x = pts[sel, 0]
            assert len(sel) >= core[r] and x.min() >= edges[r] - 1.0 and x.max() < edges[r + 1] + 1.0
    drv = sharded.ShardedIcp(None, "o3d_p2p", 30, None)
    assert drv.n_steps() == 31                                        # final evaluation after the last update


# ------------------------------------------------------------------ routing: collectives only where a scan spans slabs
class NumpyRoutedIcp:
    """Stand-in for api.Icp behind sharded.RoutedRegistration on one rank: a batch of scans, local (unsharded) or
    sharded over a gloo sub-group.  Counts the collectives it issues."""

    def __init__(self, orc, map_pts, normals, log):
        self.orc, self.map, self.nrm, self.log = orc, map_pts, normals, log
        self.lo = self.hi = None

    def set_shard(self, lo, hi):
        self.lo, self.hi = lo, hi

    def set_source_batch(self, scans):
        self.scans = scans
        self.inits = [np.eye(4)] * len(scans)

    def set_initial_batch(self, inits):
        self.inits = [np.eye(4)] * len(self.scans) if inits is None else list(inits)

    def align_batch_async(self, mode):
        self.results = []
        for s, T0 in zip(self.scans, self.inits):
            r = self.orc.icp_p2plane(s, self.map, self.nrm, T0, MAX_DIST, ITERS)
            self.results.append(dict(T64=r["T"], iterations=r["iterations"], n_corr=r["n_corr"], flags=0))
        self.log.append(("local", len(self.scans)))

    def fetch_results(self):
        return self.results

    def align_sharded(self, mode, group):
        xchg = torch.zeros(32 * len(self.scans), dtype=torch.float64)
        steps = [NumpyShardStep(self.orc, self.map, self.nrm, s, self.lo, self.hi, xchg[32 * k:32 * (k + 1)]) for k, s in enumerate(self.scans)]
        for k, st in enumerate(steps):
            st.step_begin(mode, 1)
            st.T = np.array(self.inits[k], dtype=np.float64)
        for it in range(ITERS):
            for st in steps:
                st.step_begin(mode, 0)
            dist.all_reduce(xchg, group=group)                 # one collective per iteration for the whole group, on the sub-group only
            self.log.append(("allreduce", dist.get_process_group_ranks(group)))
            for st in steps:
                st.step_end(mode, it == ITERS - 1)
        return [dict(T64=st.T, iterations=st.iterations, n_corr=st.n_corr, flags=0) for st in steps]


def routed_world(orc, synth, world):
    """Map, slab edges and a batch of scans: inside one slab, straddling two, over the whole map, and one that starts in
    one slab and is moved across an edge by its initial pose."""
    raw = synth.make_map(60_000)
    ds = orc.voxel_pcl(raw, 0.1)[0]
    from slam_sensor_fusion_amd import sharded
    edges = sharded.slab_edges(ds[:, 0], world)
    rng = np.random.default_rng(5)

    def scan_from(mask, n=1500, T=None, seed=0):
        pts = ds[mask]
        return synth.make_scan(pts, n, scan_id=50 + seed, T=T)[0]

    x = ds[:, 0]
    inner = [0.5 * (max(edges[r], x.min()) + min(edges[r + 1], x.max())) for r in range(world)]
    scans, inits = [], []
    r_last = world - 1
    scans.append(scan_from(np.abs(x - inner[r_last]) < 0.4, seed=1))                     # 0: inside the last slab
    inits.append(np.eye(4))
    e = edges[1]
    scans.append(scan_from(np.abs(x - e) < 0.6, seed=2))                                  # 1: straddles slabs 0 | 1
    inits.append(np.eye(4))
    scans.append(scan_from(np.ones(len(ds), bool), n=2500, seed=3))                       # 2: the whole map
    inits.append(np.eye(4))
    scans.append(scan_from(np.abs(x - inner[0]) < 0.4, seed=4))                           # 3: inside slab 0
    inits.append(np.eye(4))
    T_move = synth.make_T((0.8, 0.0, 0.0), (0.0, 0.0, 0.0))                               # 4: lies wholly inside slab 0 as given ...
    scans.append(scan_from(np.abs(x - e) < 0.35, T=T_move, seed=5))
    inits.append(synth.make_T((0.78, 0.01, 0.0), (0.0, 0.0, 0.0)))                        # ... and is moved across the edge by its prior
    assert scans[-1][:, 0].max() < e - 0.3
    n = min(len(s) for s in scans)
    scans = np.stack([s[:n] for s in scans])
    return ds, edges, scans, np.stack(inits)


def worker_routed(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    from slam_sensor_fusion_amd import api, sharded, synth
    ds, edges, scans, inits = routed_world(orc, synth, world)
    keep = sharded.slab_select(ds, edges, rank, halo=MAX_DIST + NORMAL_RADIUS + 0.25)
    # every rank's normals come from the full map here, so that a local registration can be compared bit for bit in
    # its record sums with the unsharded one (on the GPU the halo makes the slab's own normals identical in the core)
    normals_full, _ = orc.normals_radius(ds, NORMAL_RADIUS)
    local_map, local_nrm = ds[keep], normals_full[keep]
    groups_by_range = {rg: dist.new_group(list(range(rg[0], rg[1] + 1))) for rg in sharded.contiguous_ranges(world)}   # same order on every rank
    log = []
    reg = sharded.RoutedRegistration(rank, world, edges, api.shard_route,
                                     make_local=lambda: NumpyRoutedIcp(orc, local_map, local_nrm, log),
                                     make_sharded=lambda lo, hi: (NumpyRoutedIcp(orc, local_map, local_nrm, log), groups_by_range[(lo, hi)]),
                                     margin=0.3)
    groups = reg.set_source_batch(scans, inits)
    res = reg.align("p2plane")
    np.save(os.path.join(out_dir, "groups_%d.npy" % rank), np.array([[a, e, b] for (a, e), ids in groups.items() for b in ids]))
    for b, r in res.items():
        np.save(os.path.join(out_dir, "T_%d_%d.npy" % (b, rank)), r["T64"])
        np.save(os.path.join(out_dir, "n_%d_%d.npy" % (b, rank)), np.array([r["n_corr"], r["iterations"]]))
    np.save(os.path.join(out_dir, "coll_%d.npy" % rank), np.array([len([1 for kind, _ in log if kind == "allreduce"]), len([1 for kind, _ in log if kind == "local"])]))
    # a rank never issues a collective on a group it is not part of
    for kind, who in log:
        assert kind != "allreduce" or rank in who
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_routed_registration_only_spanning_scans_use_collectives(orc, synth, tmp_path, world):
    mp.spawn(worker_routed, args=(world, free_port(), str(tmp_path)), nprocs=world, join=True)
    ds, edges, scans, inits = routed_world(orc, synth, world)
    normals, _ = orc.normals_radius(ds, NORMAL_RADIUS)
    plan = np.load(tmp_path / "groups_0.npy")
    for r in range(1, world):
        assert np.array_equal(plan, np.load(tmp_path / ("groups_%d.npy" % r)))        # every rank computes the same routing
    ranges = {int(b): (int(a), int(e)) for a, e, b in plan}
    assert ranges[0] == (world - 1, world - 1) and ranges[3] == (0, 0)                 # single-slab scans: one rank, no collective
    assert ranges[1] == (0, 1) and ranges[2] == (0, world - 1) and ranges[4] == (0, 1)
    for b in range(len(scans)):
        a, e = ranges[b]
        ref = orc.icp_p2plane(scans[b], ds, normals, inits[b], MAX_DIST, ITERS)
        Ts = []
        for r in range(world):
            f = tmp_path / ("T_%d_%d.npy" % (b, r))
            assert f.exists() == (a <= r <= e), (b, r)                                 # only the ranks of the scan's range take part
            if f.exists():
                Ts.append(np.load(f))
                n = np.load(tmp_path / ("n_%d_%d.npy" % (b, r)))
                assert n[0] == ref["n_corr"] and n[1] == ITERS                         # every query owned exactly once
        for T in Ts[1:]:
            assert np.array_equal(T, Ts[0])                                            # identical solve on every participating rank
        dt, dr = synth.pose_error(Ts[0], ref["T"])
        assert dt < 1e-9 and dr < 1e-9, (b, dt, dr)
    # collectives: rank r joins ITERS all-reduces per multi-slab group it belongs to, none for its local scans
    multi = sorted(set(rg for rg in ranges.values() if rg[0] != rg[1]))
    for r in range(world):
        coll = np.load(tmp_path / ("coll_%d.npy" % r))
        assert coll[0] == ITERS * sum(1 for a, e in multi if a <= r <= e)
        assert coll[1] == len(set(rg for rg in ranges.values() if rg == (r, r)))


# ------------------------------------------------------------------ a scan must not leave the slabs it was routed to
class _StubIcp:
    """icp-like whose 'registration' returns a prescribed pose (what check_reach looks at)."""

    def __init__(self, T):
        self.T, self.shard = T, None

    def set_shard(self, lo, hi):
        self.shard = (lo, hi)

    def set_source_batch(self, scans):
        self.n = len(scans)

    def set_initial_batch(self, inits):
        pass

    def align_batch_async(self, mode):
        pass

    def fetch_results(self):
        return [dict(T64=self.T, flags=0)] * self.n

    def align_sharded(self, mode, comm):
        return self.fetch_results()


def test_routed_registration_raises_when_a_scan_leaves_its_slabs():
    """an initial error larger than the routing margin.  Scan 0 lies inside slab 0, scan 1 spans slabs 0..1 of
    three; a registration that ends 2 m further along +x leaves the range the routed ranks cover and must raise on them
    instead of returning a pose computed without the next slab's points; the end ranks of a group own to +-infinity."""
    from slam_sensor_fusion_amd import api, sharded
    edges = np.array([-np.inf, 0.0, 5.0, np.inf])
    rng = np.random.default_rng(0)
    inside = np.c_[rng.uniform(-4.0, -2.0, 200), rng.uniform(-1, 1, 200), rng.uniform(-1, 1, 200)].astype(np.float32)
    spanning = np.c_[rng.uniform(-2.0, 2.0, 200), rng.uniform(-1, 1, 200), rng.uniform(-1, 1, 200)].astype(np.float32)
    scans = np.stack([inside, spanning])
    for shift, ok in ((0.5, True), (2.7, False), (4.0, False)):
        T = np.eye(4)
        T[0, 3] = shift
        made = []

        def make(T=T):
            made.append(_StubIcp(T))
            return made[-1]
        for rank in (0, 1):
            reg = sharded.RoutedRegistration(rank, 3, edges, api.shard_route, make_local=make, make_sharded=lambda lo, hi: (make(), None), margin=1.0, slack=0.5)
            plan = reg.set_source_batch(scans, None)
            assert plan == {(0, 0): [0], (0, 1): [1]}
            if rank == 0:
                assert made[-1].shard == (-1e30, 0.0)          # first rank of the group owns down to -inf ...
            else:
                assert made[-1].shard == (0.0, 1e30)           # ... the last one up to +inf, although slab 1 ends at x = 5
            if ok or (shift == 2.7 and rank == 1):             # 2.7: only scan 0 (rank 0's) ends outside: x up to 0.7 >= 0 + slack
                assert sorted(reg.align("p2plane")) == ([0, 1] if rank == 0 else [1])
            else:
                with pytest.raises(sharded.ScanLeftItsSlabs):  # 4.0: scan 1 ends at x up to 6 >= 5 + slack on both of its ranks
                    reg.align("p2plane")
