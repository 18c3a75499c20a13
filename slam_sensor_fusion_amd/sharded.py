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


# ------------------------------------------------------------------ routing: collectives only where a scan spans slabs
def contiguous_ranges(world):
    """Every rank range [lo, hi] with lo < hi, in the order every rank must create / use communicators in."""
    return [(lo, hi) for lo in range(world) for hi in range(lo + 1, world)]


def plan_groups(lo, hi):
    """Scans grouped by the slab range they touch: {(lo, hi): [scan ids]}, ranges sorted.  Scans without a finite
    point (hi < lo) are in no group."""
    groups = {}
    for b, (a, e) in enumerate(zip(lo, hi)):
        if e >= a:
            groups.setdefault((int(a), int(e)), []).append(b)
    return dict(sorted(groups.items()))


class ScanLeftItsSlabs(RuntimeError):
    """A registration moved a scan out of the x-range its routed ranks cover: the result may lack correspondences that
    only another rank's slab holds.  Re-register with a larger routing margin."""


def box_x_range(box, T):
    """x-extent of the axis-aligned box (lo[3], hi[3]) under the affine map T (extremes are at the corners)."""
    lo, hi = box
    xs = [T[0, 0] * (hi[0] if c & 1 else lo[0]) + T[0, 1] * (hi[1] if c & 2 else lo[1]) + T[0, 2] * (hi[2] if c & 4 else lo[2]) + T[0, 3] for c in range(8)]
    return min(xs), max(xs)


class RoutedRegistration:
    """One rank's side of registering a batch of scans against a map sharded into x-slabs (north_star: "RCCL
    all-reduce of the normal equations only when the submap spans tiles").

    Every scan is routed to the slabs its bounding box (under its initial pose, widened by `margin`) can reach
    (`route`, = api.shard_route / sf_shard_route).  A scan inside ONE slab is registered by that slab's rank alone
    with the unsharded fast path -- no collective, no other rank takes part (the rank's map carries a halo, so every
    neighbour within the correspondence distance is local).  Scans that span slabs lo..hi are registered by exactly
    those ranks with the sharded path, all-reducing on a communicator of just those ranks.  Ranks walk the groups in
    the same sorted order, so communicators shared between ranks see their collectives in the same order.

    Inside a group the first rank owns down to -inf and the last up to +inf, so no query is ever without an owner.
    What the group's maps cover is [edges[lo] - halo, edges[hi + 1] + halo): a scan whose registration ends with points
    outside the group's x-range widened by `slack` (= halo - correspondence distance: how far a query may leave the
    range and still find every neighbour within the correspondence distance in the end rank's halo) raises
    ScanLeftItsSlabs on every rank of the group -- the same decision everywhere, they hold the same pose -- instead of
    returning a result that silently lacks the correspondences of a slab nobody asked.

    make_local()            -> icp-like (set_source_batch, set_initial_batch, align_batch_async, fetch_results)
    make_sharded(lo, hi)    -> (icp-like with set_shard / set_source_batch / set_initial_batch / align_sharded(mode, comm), comm)
    """

    def __init__(self, rank, world, edges, route, make_local, make_sharded, margin=1.0, slack=0.0):
        self.rank, self.world, self.edges, self.route = rank, world, np.asarray(edges, dtype=np.float64), route
        self.make_local, self.make_sharded, self.margin, self.slack = make_local, make_sharded, margin, slack
        self.groups, self.mine, self.resumes, self.boxes = {}, {}, 0, {}

    def set_source_batch(self, scans, inits=None):
        """scans [B, n, 3] (every rank is handed the same batch), inits [B, 4, 4] or None."""
        scans = np.asarray(scans, dtype=np.float32)
        lo, hi = self.route(scans, inits, self.edges, self.margin)
        self.groups = plan_groups(lo, hi)
        self.mine, self.boxes = {}, {}
        for (a, e), ids in self.groups.items():
            if not (a <= self.rank <= e):
                continue                                      # this rank never sees these scans again
            if a == e:
                icp, comm = self.make_local(), None
            else:
                icp, comm = self.make_sharded(a, e)
                # the group's end ranks own everything beyond the group's range: a query is never unowned
                icp.set_shard(-1e30 if self.rank == a else float(self.edges[self.rank]), 1e30 if self.rank == e else float(self.edges[self.rank + 1]))
            icp.set_source_batch(scans[ids])
            icp.set_initial_batch(None if inits is None else np.asarray(inits, dtype=np.float64)[ids])
            self.mine[(a, e)] = (icp, comm, ids)
            for b in ids:
                pts = scans[b][np.isfinite(scans[b]).all(1)].astype(np.float64)
                # a scan without a finite point: an empty box at the origin, as k_scan_boxes_final gives it on the device
                # (every rank must take the same path into the collective; check_reach skips such a scan)
                self.boxes[b] = (pts.min(0), pts.max(0)) if len(pts) else None
        return self.groups

    def check_reach(self, a, e, ids, results):
        """Every scan of group (a, e) must end inside the x-range its ranks cover (see the class docstring)."""
        x_lo, x_hi = self.edges[a] - self.slack, self.edges[e + 1] + self.slack
        for b, r in zip(ids, results):
            if self.boxes[b] is None:
                continue
            x0, x1 = box_x_range(self.boxes[b], np.asarray(r["T64"], dtype=np.float64).reshape(4, 4))
            if x0 < x_lo or x1 >= x_hi:
                raise ScanLeftItsSlabs("scan %d was routed to slabs %d..%d (x in [%g, %g) with slack) but its registration ends at x in [%g, %g]: "
                                       "re-register with a routing margin above %g m" % (b, a, e, x_lo, x_hi, x0, x1, self.margin))

    def align(self, mode):
        """-> {scan id: result} for the scans this rank took part in."""
        out, self.resumes = {}, 0
        local = [(k, v) for k, v in self.mine.items() if k[0] == k[1]]
        for _, (icp, _, _) in local:
            icp.align_batch_async(mode)                       # no collective: runs while this rank waits in the groups below
        for (a, e), (icp, comm, ids) in self.mine.items():
            if a == e:
                continue
            res = icp.align_sharded(mode, comm)
            self.resumes += int(getattr(icp, "resumes", 0))
            self.check_reach(a, e, ids, res)
            out.update(dict(zip(ids, res)))
        for (a, e), (icp, _, ids) in local:
            res = icp.fetch_results()
            self.check_reach(a, e, ids, res)
            out.update(dict(zip(ids, res)))
        return out
