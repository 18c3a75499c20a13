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
