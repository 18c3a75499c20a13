"""Multi-GPU host driver: the map tile-sharded along x with a halo, one rank per GPU, the
17-/30-scalar normal-equation record all-reduced once per ICP iteration.

Every rank holds the whole scan batch (<= 2.4 MB per scan) and only accumulates the
queries whose TRANSFORMED x falls inside its core slab [x_lo, x_hi); the slab carries a
halo of (max correspondence distance + normal radius) so an owned query's true nearest
neighbour — and that neighbour's normal — are always local.  After the all-reduce every
rank runs the identical solve, so no broadcast of the pose is needed.
"""
import numpy as np


def slab_edges(map_x, world):
    """Equal-count slab boundaries along x: world+1 edges, first/last infinite."""
    if world <= 1:
        return np.array([-np.inf, np.inf])
    qs = np.quantile(np.asarray(map_x, dtype=np.float64), np.arange(1, world) / world)
    return np.concatenate([[-np.inf], qs, [np.inf]])


def slab_select(map_xyz, edges, rank, halo):
    """Indices of the map points rank `rank` must hold: core slab widened by the halo."""
    x = map_xyz[:, 0]
    lo, hi = edges[rank] - halo, edges[rank + 1] + halo
    return np.nonzero((x >= lo) & (x < hi))[0]


SF_ICP_FLAG_SHARD_STALE = 4


class ShardedIcp:
    """Drives sf_icp_step_begin / all-reduce / sf_icp_step_end for one rank.

    `allreduce()` is injected: RCCL through torch.distributed on the GPU, gloo in the CPU
    tests (where `icp` is a numpy stand-in with the same step API).  At the start of an
    alignment every rank compacts and cell-orders the queries it may own (slab + margin); a scan
    that moves close to that margin stops with SF_ICP_FLAG_SHARD_STALE — the same decision on
    every rank, they all hold the same pose — and `align()` rebuilds and resumes it.
    """

    def __init__(self, icp, mode, num_iterations, allreduce):
        self.icp, self.mode, self.iters, self.allreduce = icp, mode, num_iterations, allreduce
        self.resumes = 0

    def n_steps(self):
        # O3D_P2P evaluates once more after the last update (registration_icp's final search)
        return self.iters + 1 if self.mode == "o3d_p2p" else self.iters

    def run_pass(self, first):
        """One pass over all iterations, enqueue only (first: 1 = start, 2 = resume stale scans)."""
        steps = self.n_steps()
        for k in range(steps):
            self.icp.step_begin(self.mode, first=(first if k == 0 else 0))
            self.allreduce()
            self.icp.step_end(self.mode, last=(k == steps - 1))

    def align(self):
        """Blocking alignment of the batch; returns the per-scan results."""
        first = 1
        for _ in range(self.n_steps() + 1):          # every resume completes at least one iteration
            self.run_pass(first)
            results = self.icp.fetch_results()
            if not any(r["flags"] & SF_ICP_FLAG_SHARD_STALE for r in results):
                return results
            self.resumes += 1
            first = 2
        raise RuntimeError("sharded alignment did not finish")


class PipelinedShardedIcp:
    """Several parts of a batch, each an sf_icp on its own stream (with its own exchange buffer).

    Per iteration the parts are stepped one after the other, so the all-reduce of one part is in
    flight while the next part searches: the collective is latency-bound (a few KB), and this hides
    that latency behind kernels instead of leaving the GPU idle.  Every rank issues the collectives in
    the same order (part 0, part 1, ... per iteration).  Stale scans (see ShardedIcp) are resumed
    for all parts together; a part without stale scans just has nothing left to do.
    """

    def __init__(self, parts, mode, num_iterations):
        """parts: list of (icp, allreduce)."""
        self.parts, self.mode, self.iters = parts, mode, num_iterations
        self.resumes = 0

    def n_steps(self):
        return self.iters + 1 if self.mode == "o3d_p2p" else self.iters

    def run_pass(self, first):
        steps = self.n_steps()
        for k in range(steps):
            for i, (icp, allreduce) in enumerate(self.parts):
                icp.step_begin(self.mode, first=(first[i] if k == 0 else 0))
                allreduce()
            for icp, _ in self.parts:
                icp.step_end(self.mode, last=(k == steps - 1))

    def align(self):
        first = [1] * len(self.parts)
        for _ in range(self.n_steps() + 1):
            self.run_pass(first)
            results = [icp.fetch_results() for icp, _ in self.parts]
            stale = [any(r["flags"] & SF_ICP_FLAG_SHARD_STALE for r in res) for res in results]
            if not any(stale):
                return [r for res in results for r in res]
            self.resumes += 1
            first = [2 if s else 0 for s in stale]
        raise RuntimeError("sharded alignment did not finish")
