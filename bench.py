#!/usr/bin/env python3
"""bench.py — scan-to-map registration throughput on MI355X.

Metric (BASELINE.json): scans/s and ms/ICP-iteration, 200k-point scans vs a 10M-point map
(uniform-random synthetic).  A "step" = one batch of `--batch` scans pushed through the
whole ICP with scans and map already resident in HBM.  `--mode`:
  p2plane  (default; the north_star's hot path) 20 Gauss-Newton point-to-plane iterations, NN every iteration
  o3d_p2p  the Python reference's registration_icp (localization_node.py:233-237): 30 point-to-point iterations
  ref_cpp  the C++ reference's ICPPointToPoint::calculateAlignment (icp_point_to_point.cpp:185-254) with the node's
           parameters 0.5 / 10 / 0.05 / 1e-5 (localization_node.cpp:24-28): lazy re-search, float32 point updates

What the p2plane line quotes: `value` with the library's defaults -- exact NN result in every iteration, neighbour reuse on (a
query whose neighbour provably cannot have changed skips its search) and frozen pairs on (once a scan's pairs are certified to
stay, their sums are evaluated from moments instead of streaming the scan; same pairs, float64 sums equal to rounding), and with the
steps of the timed loop -- enqueued back to back without a host synchronisation, as the contract's loop does -- overlapping on the library's
two internal lanes (sf_icp_set_pipeline: a step's last, nearly idle launches run under the next step's first).  Measured
beside it in the same run: `value_no_pipeline` (one lane), `value_no_freeze` (frozen pairs off), `value_no_reuse` (every query searches in every iteration),
`value_32_in_flight`, `value_upload_inclusive`, `single_scan_latency_ms`, `value_stream_config4`.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

N > 1: the map is tile-sharded along x (equal-count slabs + halo).  Default (`--scan-kind whole`, BASELINE's
synthetic scans: points drawn from the WHOLE map) every scan spans every tile, so every rank owns 1/N of each
scan's queries and the 30-double normal-equation records are all-reduced once per ICP iteration; the whole
iteration loop incl. the collective is enqueued from the C side (sf_icp_align_sharded).  `--collective`:
  auto   (default) create the hand-written P2P transport (hipIpc store-and-flag, fixed rank-order sum) AND the RCCL
         communicator, verify each with a known sum, time both on the record size and use the faster one that works on
         every rank; if neither does, torch.distributed stepping from Python
  p2p / c / torch   that transport only (p2p / c fall back to torch stepping if they cannot be created on every rank)
The N > 1 line carries per-rank min / max of owned queries, NN / reduce / collective / solve time per iteration and the
owned-query build (`ranks`), the collective's measured latency (`collective`), a per-rank roofline block from the
compulsory-traffic model, and -- weak scaling -- `value_strong`, the same run with `--batch` scans in all.  `--scan-kind local` draws every scan
from a 10 m neighbourhood (what a sensor sees; the reference crops to 10 m, localization_node.cpp:296): scans are
routed to the tiles they touch, one-tile scans are registered by one rank alone with no collective, spanning scans
all-reduce on a communicator of just their ranks (sharded.RoutedRegistration).
`--scaling weak` (default): `--batch` scans in flight PER GPU (N x batch in all); `strong`: `--batch` in all.
`value` = all scans of the common batch / wall time (max over ranks).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

A_NN_P2PLANE = 754.0   # algorithmic bytes per query-iteration, point-to-plane
A_NN_P2P = 742.0
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
POSE_TOL_M, POSE_TOL_RAD = 1e-4, 1e-5          # north_star: pose within 1e-4 m / 1e-5 rad of the reference CPU path
MODE_DEFAULT_ITERS = {"p2plane": 20, "o3d_p2p": 30, "ref_cpp": 10}
N_SIMD, CLOCK_GHZ = 1024, 2.4                   # 256 CUs x 4 SIMDs; MI355X_MICROARCH.md max clock


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--map-points", type=int, default=10_000_000)
    ap.add_argument("--scan-points", type=int, default=200_000)
    ap.add_argument("--iters", type=int, default=0, help="0 = the mode's default (p2plane 20, o3d_p2p 30, ref_cpp 10)")
    ap.add_argument("--batch", type=int, default=64, help="scans registered concurrently per step (per GPU with --scaling weak); rounds 1 and 2 up to the "
                                                          "end of round 2 ran 32: that figure stays in the line as value_32_in_flight")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--scan-kind", default="whole", choices=["whole", "local"],
                    help="whole: BASELINE's scans (points drawn from the whole map); local: a 10 m neighbourhood per scan, routed to the tiles it touches")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the sharded path + collective even with one rank (rehearses the RCCL plumbing on one GPU)")
    ap.add_argument("--collective", default="auto", choices=["auto", "p2p", "c", "torch"],
                    help="auto: P2P and RCCL both created, verified and timed, the faster one used; p2p: the hand-written hipIpc store-and-flag transport; "
                         "c: RCCL from the C side; torch: step_begin / dist.all_reduce / step_end from Python")
    ap.add_argument("--mode", default="p2plane", choices=["p2plane", "o3d_p2p", "ref_cpp"])
    ap.add_argument("--cell", type=float, default=0.25)
    ap.add_argument("--query-order", default="auto", choices=["auto", "as_given", "cell"], help="sf_icp_set_query_order")
    ap.add_argument("--no-nn-reuse", action="store_true", help="sf_icp_set_nn_reuse(0): search every query in every iteration")
    ap.add_argument("--no-freeze", action="store_true", help="sf_icp_set_freeze(0): every launch of an alignment streams every query (rounds 1-3 up to here)")
    ap.add_argument("--deliver", default="all", choices=["all", "last"],
                    help="all: every step's results are fetched inside the timed region (sf_icp_fetch_previous: the step before, while this one runs); last: only the last step's, after it")
    ap.add_argument("--freeze-params", default="", help="experiment: guard_scale,guard_min,guard_max,max_tries,from_launch for sf_icp_set_freeze_params (default: the library's)")
    ap.add_argument("--force-freeze", action="store_true", help="sf_icp_set_freeze(2): frozen pairs for batches below the automatic threshold (0.7 M queries) too")
    ap.add_argument("--tile", action="store_true", help="sf_icp_set_tile_search(always): the searching launches served out of LDS tile by tile (sf_tile.hpp; measured "
                                                        "slower than the walk through the global grid index, which stays the default)")
    ap.add_argument("--no-pipeline", action="store_true", help="sf_icp_set_pipeline(0): consecutive steps do not overlap (every alignment on the context's stream)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extra legs (no-reuse throughput, upload-inclusive rate, single-scan latency)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, default); gloo to rehearse the N>1 path with several ranks on ONE GPU (rendezvous and barriers only: the data "
                         "path is --collective p2p, which works between processes on one device, or torch)")
    return ap.parse_args()


def local_scan(synth, map_sorted_x, center, n_points, scan_id):
    """A scan a sensor at `center` sees: n points of the map within 10 m (localization_node.cpp:296), noise and T_true as make_scan."""
    lo, hi = np.searchsorted(map_sorted_x[:, 0], [center[0] - 10.0, center[0] + 10.0])
    near = map_sorted_x[lo:hi]
    near = near[((near[:, :2] - center[:2]) ** 2).sum(1) < 100.0]
    return synth.make_scan(near, n_points, scan_id=scan_id)[0]


def main():
    args = parse()
    import torch
    from slam_sensor_fusion_amd import api, synth, sharded

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback); torch.cuda.is_available() is False")
    device = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device)
    iters = args.iters or MODE_DEFAULT_ITERS[args.mode]
    dist = None
    sharded_run = world > 1 or args.force_dist
    if sharded_run and args.mode == "ref_cpp":
        raise SystemExit("the sharded path registers with p2plane / o3d_p2p (sf_icp_step_begin); ref_cpp runs unsharded")
    if sharded_run:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")
            if args.collective in ("auto", "c"):
                args.collective = "p2p"              # RCCL refuses several ranks on one device; P2P works between processes on one device
    ddev = "cuda" if (dist is not None and args.dist_backend == "nccl") else "cpu"   # where torch.distributed's own tensors live

    def agree(ok):
        """True iff `ok` holds on every rank (any failure must send every rank down the same branch)."""
        if dist is None:
            return bool(ok)
        flag = torch.tensor([1 if ok else 0], device=ddev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    def gather_stats(obj):
        """[obj of rank 0, ..., obj of rank N-1] on every rank (small python objects)."""
        if dist is None or world == 1:
            return [obj]
        out_ = [None] * world
        dist.all_gather_object(out_, obj)
        return out_

    stream = torch.cuda.Stream()
    ctx = api.Context(device, stream.cuda_stream)

    # ---------------- setup (untimed): map build on the device, scans resident in HBM
    t_setup = time.time()
    raw = synth.make_map(args.map_points)
    cloud = api.Cloud(ctx, raw)
    del raw
    cloud.voxel_downsample(0.1, "pcl")            # a4: map voxel grid, leaf 0.1 m
    map_ds = cloud.download()
    n_map = len(map_ds)
    max_dist = 0.5
    normal_radius = 0.25
    edges = sharded.slab_edges(map_ds[:, 0], world)
    if world > 1:
        keep = sharded.slab_select(map_ds, edges, rank, halo=max_dist + normal_radius + args.cell)
        cloud = api.Cloud(ctx, map_ds[keep])
    mp = api.Map(ctx, cloud, args.cell)
    if args.mode == "p2plane":
        mp.estimate_normals(normal_radius)
    B = args.batch * world if args.scaling == "weak" else args.batch

    def make_one(b, msx=None, centers=None):
        if args.scan_kind == "whole":
            return synth.make_scan(map_ds, args.scan_points, scan_id=b)[0]
        return local_scan(synth, msx, centers[b], args.scan_points, b)

    msx = centers = None
    if args.scan_kind == "local":
        msx = map_ds[np.argsort(map_ds[:, 0], kind="stable")]
        rng = np.random.Generator(np.random.PCG64(synth.SCAN_SEED - 1))
        L = float(np.sqrt(args.map_points / synth.DENSITY))
        centers = np.c_[rng.uniform(-L / 2 + 10, L / 2 - 10, B), rng.uniform(-L / 2 + 10, L / 2 - 10, B), np.zeros(B)]
    if world > 1 and B % world == 0:
        # every scan is generated ONCE: rank r makes scans r*B/N .. (r+1)*B/N - 1, an all-gather hands them round
        per = B // world
        mine_ = [make_one(b, msx, centers) for b in range(rank * per, (rank + 1) * per)]
        n_min = torch.tensor([min(len(s_) for s_ in mine_)], device=ddev)
        dist.all_reduce(n_min, op=dist.ReduceOp.MIN)
        n_min = int(n_min.item())
        part = torch.from_numpy(np.stack([s_[:n_min] for s_ in mine_])).to(ddev)
        full = torch.empty((B,) + tuple(part.shape[1:]), dtype=part.dtype, device=ddev)
        dist.all_gather_into_tensor(full, part)
        scans = full.cpu().numpy()
        del part, full, mine_
    else:
        scans = [make_one(b, msx, centers) for b in range(B)]
        n_min = min(len(s_) for s_ in scans)
        scans = np.stack([s_[:n_min] for s_ in scans])
    del msx
    n_scan = scans.shape[1]

    def new_icp(context=ctx):
        icp = api.Icp(context, max_dist, iters, 0.05, 1e-5)
        icp.set_target(mp)
        icp.use_graph(not args.no_graph)
        icp.set_query_order(args.query_order)
        icp.set_nn_reuse(not args.no_nn_reuse)
        icp.set_freeze(False if args.no_freeze else (True if args.force_freeze else "auto"))
        icp.set_tile_search("always" if args.tile else False)
        icp.set_pipeline(not args.no_pipeline)
        if args.freeze_params:
            gs, gmin, gmax, tries, frm = args.freeze_params.split(",")
            icp.set_freeze_params(float(gs), float(gmin), float(gmax), int(tries), int(frm))
        return icp

    # ---------------- the registration driver of this rank
    routed, comm_kind, icp = None, "none", None
    coll_info = None
    my_scans = list(range(B))                     # scan ids this rank takes part in
    if not sharded_run:
        icp = new_icp()
        icp.set_source_batch(scans)
        icp.set_initial_batch(None)
    else:
        import uuid
        job = [uuid.uuid4().hex[:10] if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(job, src=0)  # names the shared-memory rendezvous objects of this run
        job = job[0]
        init_timeout = float(os.environ.get("SF_COMM_INIT_TIMEOUT_S", "180"))
        rec_count = 32 * B                          # doubles in one all-reduce of a whole batch's records

        def make_p2p(lo, hi):
            """P2P communicator of the ranks lo..hi: hipIpc handles meet in a POSIX shared-memory object (ranks of one node);
            the rendezvous has its own time limit, nothing is left blocked when a rank does not arrive."""
            c = api.Comm.p2p(ctx, hi - lo + 1, rank - lo, rec_count)
            try:
                c.rendezvous("sfb_%s_%d_%d" % (job, lo, hi), init_timeout)
            except Exception:
                c.close()
                raise
            if os.environ.get("SF_COMM_TIMEOUT_S"):    # how long a collective waits for a peer before every rank is told to stop (default 20 s)
                c.set_timeout(float(os.environ["SF_COMM_TIMEOUT_S"]))
            return c

        def make_rccl(lo, hi):
            """C-side RCCL communicator of the ranks lo..hi (every member calls this in the same order)."""
            ident = [None]
            if rank == lo:
                try:                               # whatever happens here, the broadcast below must still take place
                    ident = [api.Comm.unique_id()]
                except Exception as e:             # noqa: BLE001
                    print("rank %d: ncclGetUniqueId failed: %s" % (rank, e), file=sys.stderr)
            dist.broadcast_object_list(ident, src=lo, group=groups[(lo, hi)] if (lo, hi) in groups else None)
            if ident[0] is None:
                raise RuntimeError("no RCCL unique id for ranks %d..%d" % (lo, hi))
            # ncclCommInitRank blocks until every member has called it.  Should it never return (a rank lost, a bootstrap
            # interface that cannot be reached) a thread stays inside RCCL for good: this process must not carry on with
            # collectives on the same device next to it, so the run ends here, non-zero, with the reason.
            import threading
            box = {}

            def work():
                try:
                    box["comm"] = api.Comm(ctx, hi - lo + 1, rank - lo, ident[0])
                except Exception as e:             # noqa: BLE001
                    box["err"] = e
            th = threading.Thread(target=work, daemon=True)
            th.start()
            th.join(init_timeout)
            if th.is_alive():
                print("bench.py: rank %d: ncclCommInitRank of ranks %d..%d did not return within %.0f s -- giving up" % (rank, lo, hi, init_timeout), file=sys.stderr, flush=True)
                os._exit(4)
            if "err" in box:
                raise box["err"]
            return box["comm"]

        makers = {"p2p": make_p2p, "c": make_rccl}
        groups = {}
        if args.scan_kind == "local" and world > 1:   # torch sub-groups: rendezvous for the RCCL ids (and the torch fallback's collectives)
            groups = {rg: dist.new_group(list(range(rg[0], rg[1] + 1))) for rg in sharded.contiguous_ranges(world) if rg != (0, world - 1)}

        def try_comm(kind, lo, hi):
            """-> (ok, communicator of `kind` for ranks lo..hi or None on non-members); ok is the same on every rank of the WORLD
            (every rank calls this, members or not)."""
            c, member = None, lo <= rank <= hi
            if member:
                try:
                    c = makers[kind](lo, hi)
                except Exception as e:               # noqa: BLE001 -- any failure: every rank must take the same branch
                    print("rank %d: %s communicator of ranks %d..%d failed: %s" % (rank, kind, lo, hi, e), file=sys.stderr)
            if not agree(c is not None or not member):
                if c is not None:
                    c.close()
                return False, None
            return True, c

        def verify_and_time(c):
            """A known sum through the communicator (rank r contributes r + 1 + i/1024) and its latency on the record size:
            -> microseconds per all-reduce as the slowest rank's stream saw it, or None if the sum is wrong anywhere."""
            i_ = torch.arange(rec_count, dtype=torch.float64, device="cuda")
            with torch.cuda.stream(stream):
                x = (rank + 1) + i_ / 1024.0
                c.allreduce_f64(x.data_ptr(), rec_count)
            stream.synchronize()
            want = world * (world + 1) / 2.0 + world * i_ / 1024.0
            good = True
            try:
                c.status()
            except api.SlamFusionError:
                good = False
            if not agree(good and bool(torch.equal(x, want))):
                return None
            reps = 100
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if dist is not None:
                dist.barrier()
            with torch.cuda.stream(stream):
                e0.record(stream)
                for _ in range(reps):
                    c.allreduce_f64(x.data_ptr(), rec_count)
                e1.record(stream)
            stream.synchronize()
            t = torch.tensor([e0.elapsed_time(e1) / reps * 1e3], dtype=torch.float64, device=ddev)
            if dist is not None:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        want_kinds = {"auto": ["p2p", "c"], "p2p": ["p2p"], "c": ["c"], "torch": []}[args.collective]
        world_comms, coll_us = {}, {}
        for kind in want_kinds:
            ok_, c = try_comm(kind, 0, world - 1)
            if not ok_:
                continue
            us = verify_and_time(c)
            if us is None:
                print("rank %d: the %s communicator did not sum correctly; not used" % (rank, kind), file=sys.stderr)
                c.close()
                continue
            world_comms[kind], coll_us[kind] = c, us
        chosen = min(world_comms, key=lambda k: coll_us[k]) if world_comms else None
        for kind, c in world_comms.items():
            if kind != chosen:
                c.close()
        names = {"p2p": "hand-written P2P (hipIpc store-and-flag, fixed rank-order sum) from the C side (sf_icp_align_sharded)",
                 "c": "RCCL from the C side (sf_icp_align_sharded)", None: "torch.distributed all_reduce per iteration (Python stepping)"}
        comm_kind = names[chosen]
        coll_info = {"kind": chosen or "torch", "allreduce_us_measured": coll_us, "bytes_per_allreduce": rec_count * 8,
                     "note": "one all-reduce of the batch's 32-double records per ICP iteration; allreduce_us_measured: 100 back-to-back all-reduces of that "
                             "size on the stream, slowest rank, measured before the run for every transport that came up and summed a known pattern correctly"}

        class TorchSharded:
            """Fallback: the stepping API + dist.all_reduce from Python (round-1 path), same interface as api.Icp.align_sharded."""

            def __init__(self, icp_, group):
                self.icp, self.group, self.resumes = icp_, group, 0

            def __getattr__(self, name):
                return getattr(self.icp, name)

            def set_source_batch(self, s):
                self.icp.set_source_batch(s)
                self.xb = torch.zeros(self.icp.batch * 32, dtype=torch.float64, device="cuda")
                self.icp.set_exchange_buffer(self.xb.data_ptr(), self.xb.numel() * 8)

            def align_sharded(self, mode, comm_):
                def allreduce():
                    with torch.cuda.stream(stream):
                        if ddev == "cuda":
                            dist.all_reduce(self.xb, group=self.group)
                        else:                        # gloo rehearsal on one GPU: through the host
                            stream.synchronize()
                            h = self.xb.cpu()
                            dist.all_reduce(h, group=self.group)
                            self.xb.copy_(h)
                drv = sharded.ShardedIcp(self.icp, mode, iters, allreduce)
                res = drv.align()
                self.resumes = drv.resumes
                return res

        comms = {(0, world - 1): world_comms.get(chosen)}
        slack = normal_radius + args.cell              # = halo - correspondence distance (sharded.RoutedRegistration)
        if chosen and args.scan_kind == "local" and world > 1:
            # sub-group communicators: planned first, created in the same sorted order by every rank, all inside the same
            # agreed try / fall-back as the world communicator -- one failure anywhere and EVERY rank steps through torch
            lo_, hi_ = api.shard_route(scans, None, edges, 1.0)
            ok_all = True
            for (a, e) in sharded.plan_groups(lo_, hi_):
                if a != e and (a, e) not in comms and ok_all:
                    ok_all, comms[(a, e)] = try_comm(chosen, a, e)
            if not ok_all:
                for c in comms.values():
                    if c is not None:
                        c.close()
                comms, chosen = {}, None
                comm_kind, coll_info["kind"] = names[None], "torch"

        def make_sharded(lo, hi):
            if chosen:
                return new_icp(), comms[(lo, hi)]
            return TorchSharded(new_icp(), groups.get((lo, hi))), None

        ctx_local = api.Context(device, torch.cuda.Stream().cuda_stream) if args.scan_kind == "local" else ctx
        if world == 1:                              # --force-dist: one rank, the sharded path on the whole batch (routing would call it local)
            class OneRank:
                groups, resumes = {(0, 0): list(range(B))}, 0

                def __init__(self, scans_):
                    self.icp, self.comm = make_sharded(0, 0)
                    self.icp.set_shard(-1e30, 1e30)
                    self.icp.set_source_batch(scans_)
                    self.icp.set_initial_batch(None)
                    self.mine = {(0, 0): (self.icp, self.comm, list(range(len(scans_))))}

                def align(self, mode):
                    res = self.icp.align_sharded(mode, self.comm)
                    self.resumes = int(getattr(self.icp, "resumes", 0))
                    return dict(enumerate(res))

            def make_routed(scans_):
                return OneRank(scans_)
        else:
            def make_routed(scans_):
                r_ = sharded.RoutedRegistration(rank, world, edges, api.shard_route, make_local=lambda: new_icp(ctx_local), make_sharded=make_sharded, margin=1.0,
                                                slack=slack)
                r_.set_source_batch(scans_, None)
                return r_
        routed = make_routed(scans)
        my_scans = sorted(b_ for (a, e), ids in routed.groups.items() if a <= rank <= e for b_ in ids)
    setup_s = time.time() - t_setup

    results_box = {}
    # Every step's results reach the host inside the timed region (--deliver all, the default): the streaming loop of
    # include/slamfusion.h -- enqueue this step's alignment, then fetch the PREVIOUS step's results (sf_icp_fetch_previous waits for
    # that alignment only, this one goes on running on the other lane); with one lane (--no-pipeline) every step is fetched before
    # the next is enqueued.  --deliver last: only the last step's results are fetched, after the timed region (rounds 1-3).
    delivered = []
    lanes_on = routed is None and not args.no_pipeline
    deliver_all = routed is None and args.deliver == "all"
    enq = {"n": 0}

    def step(deliver=False):
        if routed is None:
            icp.align_batch_async(args.mode)
            enq["n"] += 1
            if deliver_all and deliver:
                if not lanes_on:
                    delivered.append(icp.fetch_results(raw=True))
                elif enq["n"] > 1:
                    delivered.append(icp.fetch_previous(raw=True))
        else:
            results_box["r"] = routed.align(args.mode)    # blocking per group: resumes are decided on fetched states

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(True)
    barrier()
    del delivered[:]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed_local = elapsed
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=ddev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if routed is None:
        results = dict(enumerate(icp.fetch_results()))
        # what was delivered inside the timed region (with the lanes: every step but the last, whose fetch is the line above, plus the
        # last warm-up step's): the same inputs every step, so every delivered pose must be the last step's, bit for bit
        delivered_ok = all(np.array_equal(np.array(arr[b].T64), results[b]["T64"].ravel()) and arr[b].iterations == results[b]["iterations"]
                           for arr in delivered for b in range(B))
        delivery = {"mode": args.deliver if deliver_all else "last", "results_fetched_in_timed_region": len(delivered) * B,
                    "how": ("sf_icp_fetch_previous after every enqueue (two lanes)" if lanes_on else "sf_icp_fetch_results after every enqueue (one lane)") if deliver_all else None,
                    "all_equal_to_last": bool(delivered_ok)}
        if not delivered_ok:
            print("bench: a delivered result differs from the last step's", file=sys.stderr)
    else:
        results = results_box["r"]
        delivery = {"mode": "all", "how": "sf_icp_align_sharded fetches every step (blocking)"}

    # ---------------- correctness of what was timed: the registrations against the generating transform
    terr = max(synth.pose_error(r["T64"], synth.t_true())[0] for r in results.values()) if results else 0.0
    rerr = max(synth.pose_error(r["T64"], synth.t_true())[1] for r in results.values()) if results else 0.0
    truth_bar = (0.05, 5e-3) if args.mode == "ref_cpp" else (5e-3, 5e-4)   # ref_cpp stops at its 5 cm mean-error rule
    ok_truth = all(r["iterations"] == iters or args.mode != "p2plane" for r in results.values()) and terr < truth_bar[0] and rerr < truth_bar[1]
    if routed is None:
        ok_truth = ok_truth and delivered_ok   # (a delivered result that is not the last step's, bit for bit, fails the run)

    # ---------------- the dominant kernel (fused transform + NN + accumulate; ref_cpp: the search kernel), HIP events per launch
    prof = None
    if routed is None:
        icp.use_graph(False)
        icp.profile_enable(True)
        prof_steps = max(2, min(4, args.steps))
        for _ in range(prof_steps):
            step()
        torch.cuda.synchronize()
        ms, sq, sw = icp.profile_launches()
        fz_stats = icp.freeze_stats()
        icp.profile_enable(False)
        icp.use_graph(not args.no_graph)
        per = len(ms) // prof_steps
        ms, sq, sw = (a[:per * prof_steps].reshape(prof_steps, per) for a in (ms, sq, sw))
        q_launch = n_scan * B
        waves = q_launch / 64.0
        frac_search = sq.mean(0) / q_launch
        searching = frac_search > 0.5 if args.mode != "ref_cpp" else ms.mean(0) > 5e-3
        # frozen pairs (p2plane): launches from FZ_FROM on run k_nn_red_fz -- one freeze launch per scan, then launches that only
        # touch the active queries.  Labelled from the library's own counts: every scan froze once, at the first chance, and stayed so
        FZ_FROM = 5
        kind = np.where(searching, "searching", "verifying").astype(object)
        fz_clean = fz_stats["froze"] == B and fz_stats["frozen_at_end"] == B and fz_stats["failed"] == 0 and fz_stats["thawed"] == 0
        if fz_stats["froze"] > 0 and per > FZ_FROM:
            kind[FZ_FROM:] = "frozen" if fz_clean else "mixed (some scans frozen)"
            if fz_clean:
                kind[FZ_FROM] = "freeze"
        prof = dict(per=per, ms=ms.mean(0), frac_search=frac_search, wave_frac=sw.mean(0) / waves, searching=searching, q_launch=q_launch, kind=kind,
                    fz=fz_stats, fz_from=FZ_FROM if fz_stats["froze"] > 0 else per)

    # ---------------- N > 1: what every rank spent where (HIP events around every phase of the sharded loop, on the stream)
    rank_stats, roof_dist = None, None
    if routed is not None:
        cache_b = 32 if args.mode == "p2plane" else 20
        mine_stat = {"rank": rank, "step_ms": elapsed_local / args.steps * 1e3, "map_points": len(mp), "scans": len(my_scans), "resumes": int(routed.resumes)}
        grp = max(routed.mine.items(), key=lambda kv: len(kv[1][2])) if routed.mine else None
        bases = [getattr(v[0], "icp", v[0]) for v in routed.mine.values()]       # TorchSharded wraps its api.Icp
        for b_ in bases:
            b_.profile_enable(True)
        prof_steps = 2
        for _ in range(prof_steps):
            step()                                                                # every rank: the same collectives in the same order
        torch.cuda.synchronize()
        if grp is not None:
            (a_, e_), (gicp, _, gids) = grp
            base = getattr(gicp, "icp", gicp)
            ms, sq, _sw = base.profile_launches()
            per = max(len(ms) // prof_steps, 1)
            ms = ms[:per * prof_steps].reshape(prof_steps, per).mean(0)
            sq = sq[:per * prof_steps].reshape(prof_steps, per).mean(0)
            is_sharded = (a_ != e_) or world == 1
            owned = int(base.owned_counts().sum()) if is_sharded else n_scan * len(gids)
            ph = {name: base.profile_phases(kind) for name, kind in (("reduce", api.PROF_REDUCE), ("collective", api.PROF_COLLECTIVE), ("solve", api.PROF_SOLVE),
                                                                     ("shard_build", api.PROF_SHARD_BUILD))}
            searching = sq / max(owned, 1) > 0.5
            map_b = len(mp) * (32 if args.mode == "p2plane" else 16) + 4.0 * np.prod(mp.cell_size()[1])
            comp = np.where(searching, owned * (12.0 + (cache_b if not args.no_nn_reuse else 0)) + map_b, owned * (12.0 + cache_b))
            # frozen pairs (a rank freezes its own owned queries): from the launch after the freeze launch only the active queries are touched
            fzs = base.freeze_stats() if args.mode == "p2plane" else {"froze": 0, "frozen_at_end": 0, "failed": 0, "thawed": 0, "active_queries": 0}
            streamed = len(ms)
            if fzs["froze"] > 0 and fzs["frozen_at_end"] == len(gids) and fzs["failed"] == 0 and fzs["thawed"] == 0 and len(ms) > 6:
                streamed = 6
                comp[streamed:] = fzs["active_queries"] * (12.0 + cache_b)
            mine_stat.update({
                "frozen_scans": int(fzs["frozen_at_end"]), "active_queries": int(fzs["active_queries"]), "launches_streaming_the_owned_queries": int(streamed),
                "group": [int(a_), int(e_)], "group_scans": len(gids), "owned_queries_per_launch": owned,
                "nn_us_per_launch": [round(float(v) * 1e3, 1) for v in ms], "nn_us_mean": float(ms.mean() * 1e3),
                "queries_searching_frac_per_launch": [round(float(v) / max(owned, 1), 4) for v in sq],
                "reduce_us_mean": float(ph["reduce"].mean() * 1e3) if len(ph["reduce"]) else None,
                "collective_us_mean": float(ph["collective"].mean() * 1e3) if len(ph["collective"]) else None,
                "collective_us_max": float(ph["collective"].max() * 1e3) if len(ph["collective"]) else None,
                "solve_us_mean": float(ph["solve"].mean() * 1e3) if len(ph["solve"]) else None,
                "shard_build_ms_mean": float(ph["shard_build"].mean()) if len(ph["shard_build"]) else None,
                "compulsory_bytes_per_launch": float(comp[:streamed].mean()),
                "compulsory_gbs": float(comp[:streamed].mean() / max(ms[:streamed].mean() * 1e-3, 1e-12) / 1e9),   # (over the launches that stream the owned queries)
            })
        for b_ in bases:
            b_.profile_enable(False)
        all_stats = gather_stats(mine_stat)
        if rank == 0:
            def mm(key):
                v = [st[key] for st in all_stats if st.get(key) is not None]
                return {"min": min(v), "max": max(v), "mean": float(np.mean(v))} if v else None
            rank_stats = {k: mm(k) for k in ("step_ms", "map_points", "scans", "owned_queries_per_launch", "nn_us_mean", "reduce_us_mean", "collective_us_mean",
                                             "collective_us_max", "solve_us_mean", "shard_build_ms_mean", "resumes", "frozen_scans", "active_queries")}
            rank_stats["per_rank"] = all_stats
            rank_stats["note"] = ("HIP events on each rank's stream around every phase of the sharded loop of its largest group, 2 profiled steps after the timed "
                                  "region; collective_us includes waiting for the slowest peer; nn_us_per_launch / queries_searching_frac_per_launch: the k_nn_red "
                                  "launches of one alignment in order")
            gbs = mm("compulsory_gbs")
            if gbs is not None:
                roof_dist = {"bound": "hbm", "achieved": gbs["mean"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs["mean"] / HBM_PEAK_GBS,
                             "frac_min_rank": gbs["min"] / HBM_PEAK_GBS, "frac_max_rank": gbs["max"] / HBM_PEAK_GBS, "traffic": None,
                             "achieved_from": "compulsory-traffic model per rank (owned queries x (12 B + neighbour cache) + the rank's map lines and cell table once per "
                                              "searching launch) / that rank's mean k_nn_red launch time over the launches that stream its owned queries "
                                              "(ranks.launches_streaming_the_owned_queries: with the frozen pairs the first six); mean over ranks",
                             "kernel": "k_nn_red<SHARD>", "avg_launch_ms": mm("nn_us_mean")["mean"] * 1e-3,
                             "queries_per_launch": mm("owned_queries_per_launch")}

    # ---------------- N > 1, weak scaling: the same machine on `--batch` scans in ALL (strong scaling) for the record
    strong = None
    if routed is not None and world > 1 and args.scaling == "weak" and not args.no_extras:
        keep_routed = routed
        routed = make_routed(scans[:args.batch])
        step()
        barrier()
        k_s = max(3, min(10, args.steps))
        t1 = time.perf_counter()
        for _ in range(k_s):
            step()
        barrier()
        ts = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=ddev)
        dist.all_reduce(ts, op=dist.ReduceOp.MAX)
        strong = {"value": args.batch * k_s / float(ts.item()), "scans_in_flight": args.batch, "ms_per_step": float(ts.item()) / k_s * 1e3}
        routed = keep_routed

    # ---------------- N > 1: the fallback SURVEY §8e names -- REPLICAS: every rank holds the whole map (330 MB: it fits) and registers
    # its own scans unsharded, no collective at all.  Reported beside the sharded metric, never as it: it says what the node does when
    # the map fits one GPU, and bounds from above what sharding can reach.
    replicas = None
    if routed is not None and world > 1 and args.scan_kind == "whole" and not args.no_extras:
        full_cloud = api.Cloud(ctx, map_ds)
        full_map = api.Map(ctx, full_cloud, args.cell)
        if args.mode == "p2plane":
            full_map.estimate_normals(normal_radius)
        per = B // world if args.scaling == "weak" else max(B // world, 1)
        mine_ = scans[rank * per:(rank + 1) * per] if per * world <= B else scans[:per]
        rep = api.Icp(ctx, max_dist, iters, 0.05, 1e-5)
        rep.set_target(full_map)
        rep.use_graph(not args.no_graph)
        rep.set_query_order(args.query_order)
        rep.set_nn_reuse(not args.no_nn_reuse)
        rep.set_source_batch(mine_)
        rep.set_initial_batch(None)
        rep.align_batch_async(args.mode)
        barrier()
        k_r = max(3, min(10, args.steps))
        t1 = time.perf_counter()
        for _ in range(k_r):
            rep.align_batch_async(args.mode)
        barrier()
        tr = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=ddev)
        dist.all_reduce(tr, op=dist.ReduceOp.MAX)
        rr = rep.fetch_results()
        terr_r = max(synth.pose_error(r_["T64"], synth.t_true())[0] for r_ in rr)
        replicas = {"value": per * world * k_r / float(tr.item()), "unit": "scans/s", "scans_in_flight_per_gpu": per, "ms_per_step": float(tr.item()) / k_r * 1e3,
                    "max_translation_err_vs_truth_m": terr_r,
                    "what": "replicas, NOT the sharded metric: every rank holds the whole map and registers its own scans unsharded, no collective (SURVEY 8e fallback)"}
        rep.close()

    # ---------------- untimed extra legs: no-reuse throughput, upload-inclusive rate, single-scan latency
    extras = {}
    if routed is None and not args.no_extras:
        k = max(3, min(10, args.steps))
        if args.mode != "ref_cpp" and not args.no_nn_reuse:
            icp.set_nn_reuse(False)
            step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(k):
                step()
            torch.cuda.synchronize()
            extras["value_no_reuse"] = B * k / (time.perf_counter() - t1)
            icp.set_nn_reuse(True)
        if args.mode == "p2plane" and not args.no_nn_reuse and not args.no_freeze:
            icp.set_freeze(False)
            step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(k):
                step()
            torch.cuda.synchronize()
            extras["value_no_freeze"] = B * k / (time.perf_counter() - t1)
            icp.set_freeze("auto")
        # consecutive steps on one lane (sf_icp_set_pipeline(0)): every alignment on the context's stream, as up to round 3
        icp.set_pipeline(False)
        step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(k):
            step()
        torch.cuda.synchronize()
        extras["value_no_pipeline"] = B * k / (time.perf_counter() - t1)
        icp.set_pipeline(True)
        if args.mode != "ref_cpp" and not args.tile:
            # the searching launches served out of LDS tile by tile (sf_tile.hpp, opt-in): the measurement round 4 was asked for
            icp.set_tile_search("always")
            step()
            torch.cuda.synchronize()
            if icp.tile_info()["on"]:
                t1 = time.perf_counter()
                for _ in range(k):
                    step()
                torch.cuda.synchronize()
                extras["value_tile_search"] = B * k / (time.perf_counter() - t1)
            icp.set_tile_search(False)
        # upload-inclusive: every step uploads its batch from pinned host memory.  Double buffering: the raw H2D copy of
        # batch k+1 runs on a copy stream into one of two staging buffers while batch k is registered on the compute
        # stream, which picks the staged batch up on the device (sf_icp_set_source_batch_device) once its copy event has
        # fired.  (Two alignments side by side on two streams only fight for the caches: 4.8 ms per step against 3.6 ms.)
        pinned = torch.from_numpy(scans).pin_memory()
        copy_stream = torch.cuda.Stream()
        copy_ctx = api.Context(device, copy_stream.cuda_stream)
        staging = [api.Cloud(copy_ctx) for _ in range(2)]
        copied = [torch.cuda.Event(), torch.cuda.Event()]
        consumed = [torch.cuda.Event(), torch.cuda.Event()]
        up = new_icp()
        up.set_source_batch(scans)
        up.set_initial_batch(None)
        up.align_batch_async(args.mode)
        torch.cuda.synchronize()

        def upload(j):
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(consumed[j % 2])           # the staging buffer has been read by the SoA split two steps ago
                staging[j % 2].upload_async_ptr(pinned.data_ptr(), B * n_scan)    # hipMemcpyAsync from pinned memory on the copy stream
                copied[j % 2].record(copy_stream)

        for j in range(2):
            consumed[j].record(stream)
        t1 = time.perf_counter()
        upload(0)
        for s_ in range(k):
            if s_ + 1 < k:
                upload(s_ + 1)
            stream.wait_event(copied[s_ % 2])
            up.set_source_batch_device(staging[s_ % 2].device_ptr(), n_scan, B)
            consumed[s_ % 2].record(stream)
            up.align_batch_async(args.mode)
        torch.cuda.synchronize()
        extras["value_upload_inclusive"] = B * k / (time.perf_counter() - t1)
        extras["upload_bytes_per_step"] = int(scans.nbytes)
        up.close()
        # the same through the library alone -- the streaming loop of include/slamfusion.h: set the next batch from (pinned) host
        # memory, enqueue its alignment, fetch the previous batch's results.  With an alignment in flight the upload and the
        # conversion take the other source set on the next lane's stream (sf_icp.hip, SrcScope): they run beside the alignment,
        # and every result reaches the host.  Two host buffers take turns (what a sensor driver's double buffer does).
        if not args.no_pipeline:
            pinned2 = [pinned, torch.from_numpy(scans.copy()).pin_memory()]
            st = new_icp()
            st.set_source_batch_host_ptr(pinned2[0].data_ptr(), n_scan, B)
            st.set_initial_batch(None)
            first = st.align_batch(args.mode)
            t1 = time.perf_counter()
            got = []
            for s_ in range(k):
                st.set_source_batch_host_ptr(pinned2[s_ % 2].data_ptr(), n_scan, B)
                st.align_batch_async(args.mode)
                if s_ > 0:
                    got.append(st.fetch_previous(raw=True))
            got.append(st.fetch_results(raw=True))
            extras["value_stream_upload_inclusive"] = B * k / (time.perf_counter() - t1)
            extras["stream_results_delivered"] = len(got) * B
            extras["stream_results_equal"] = bool(all(np.array_equal(np.array(a[b].T64), first[b]["T64"].ravel()) for a in got for b in range(B)))
            st.close()
        for c_ in staging:
            c_.close()
        copy_ctx.close()
        del pinned
        if B > 32:                                   # continuity with the earlier rounds' 32 scans in flight
            i32 = new_icp()
            i32.set_source_batch(scans[:32])
            i32.set_initial_batch(None)
            i32.align_batch_async(args.mode)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(k):
                i32.align_batch_async(args.mode)
            torch.cuda.synchronize()
            extras["value_32_in_flight"] = 32 * k / (time.perf_counter() - t1)
            i32.close()
        lat = new_icp()
        lat.set_source(scans[0])
        lat.align(args.mode)
        tl = time.perf_counter()
        for _ in range(10):
            lat.align(args.mode)
        extras["single_scan_latency_ms"] = (time.perf_counter() - tl) / 10 * 1e3
        lat.close()

    # ---------------- BASELINE config 4 beside the headline: the sequential stream (per-scan path, not a batch)
    # 1000 scans of 20 k points along a 136 m corridor, the first 20 m mapped; 15-state EKF fed 100 Hz IMU samples
    # (sf_ekf_*), per-scan registration (O3D_P2P to convergence, single launch), every registered scan appended on the
    # device and the map re-voxelised + re-indexed every 10 scans -- tests/test_gpu_config4_stream.py is the same
    # run with its oracle checks.  Wall clock around every callback incl. the Python -> C calls; scans pre-generated.
    if routed is None and not args.no_extras and args.mode == "p2plane":
        from scipy.spatial.transform import Rotation
        from slam_sensor_fusion_amd.localization_flow import ImuEkfMappingFlow
        n_stream, pts_stream = 1000, 20_000
        sctx = api.Context(device)
        wc = api.Cloud(sctx, synth.make_corridor(136.0, 28.0))
        wc.voxel_downsample(0.1, "pcl")
        corridor = wc.download()
        corridor = corridor[np.argsort(corridor[:, 0], kind="stable")]
        kc = api.Cloud(sctx, corridor[corridor[:, 0] < 20.0])
        kc.voxel_downsample(0.1, "pcl")
        known = kc.download()
        lla0 = np.array([[-22.9068, -43.1729, 12.0]])
        flow = ImuEkfMappingFlow(sctx, known, api.map_T_global(lla0, np.zeros(1, np.float32)), altitude_table=lla0, grow_every=10)
        flow.coarse_alignment_complete_ = True
        stm = synth.make_stream(n_stream)
        gyro, accel, imu_dt = synth.make_imu(n_stream)
        rng = np.random.default_rng(synth.STREAM_SEED)
        msgs = []
        for k_ in range(n_stream):
            truth, odomT = stm["truth"][k_], stm["odom"][k_]
            lo_, hi_ = np.searchsorted(corridor[:, 0], [truth[0, 3] - 12.0, truth[0, 3] + 12.0])
            pick = corridor[lo_ + rng.choice(hi_ - lo_, pts_stream, replace=False)].astype(np.float64) + rng.normal(0, 0.01, (pts_stream, 3))
            Ti = np.linalg.inv(truth)
            q_ = Rotation.from_matrix(odomT[:3, :3]).as_quat()
            msgs.append(((pick @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32),
                         dict(latitude=-22.9068, longitude=-43.1729, altitude=12.0, position_covariance=stm["gps_cov"].ravel(), map_xyz=stm["gps_xyz"][k_]),
                         dict(q_wxyz=[q_[3], q_[0], q_[1], q_[2]], t=odomT[:3, 3], covariance=stm["odom_cov"].ravel()),
                         None if k_ == 0 else dict(gyro=gyro[k_ - 1], accel=accel[k_ - 1], dt=imu_dt)))
        n0 = len(known)
        cb_ms, grow_ms, errs_ = [], [], []
        t_all = time.perf_counter()
        for k_, (scan_, gps_, odom_, imu_) in enumerate(msgs):
            flow.compassCallback(90.0 - np.degrees(stm["compass"][k_]))
            g0_ = flow.growths_
            t1 = time.perf_counter()
            out_ = flow.localizationCallback(scan_, gps_, odom_, imu=imu_)
            sctx.synchronize()
            dt_ = (time.perf_counter() - t1) * 1e3
            if k_ == 0:
                flow.map_T_sensor_ = stm["truth"][0].astype(np.float32)
                flow.map_T_ref_ = stm["truth"][0].astype(np.float32)
                continue
            (grow_ms if flow.growths_ != g0_ else cb_ms).append(dt_)
            errs_.append(synth.pose_error(out_, stm["truth"][k_])[0])
        wall = time.perf_counter() - t_all
        extras["value_stream_config4"] = {
            "value": (n_stream - 1) / wall, "unit": "scans/s", "scans": n_stream, "scan_points": pts_stream,
            "callback_ms_median": float(np.median(cb_ms)), "callback_ms_p99": float(np.quantile(cb_ms + grow_ms, 0.99)),
            "growth_callback_ms_mean": float(np.mean(grow_ms)), "growth_steps": int(flow.growths_), "growth_steps_index_patched": int(flow.patches_),
            "map_points_start_end": [int(n0), int(len(flow.map_full_))],
            "drift_translation_m": {"median": float(np.median(errs_)), "last_100_median": float(np.median(errs_[-100:])), "max": float(np.max(errs_))},
            "what": "sequential 1000-scan stream: 15-state EKF with 100 Hz IMU pre-integration as the prior, O3D_P2P registration per scan (single launch), registered scans "
                    "appended on the device, voxel grid 0.1 m + grid index every 10 scans (re-filtered and rebuilt on the hand-written radix sort at this map size; from 4 M map points on the filter is a merge and the index is carried over, sf_cloud_voxel_merge / sf_map_patch: tools/growth_bench.py); host clock around every callback; reference budget 100 ms per scan"}
        flow = None
        sctx.synchronize()

        # a sensor-shaped workload beside the metric's uniform-random one (tools/city_bench.py is the full version, profiles/r04_city_bench*):
        # ring scans of a synthetic city against its surface map, every scan from its own pose with its own prior error
        if args.mode == "p2plane":
            c_ext, c_rings, c_scans = 120.0, 128, 16
            boxes = synth.make_city(c_ext, 30)
            ccloud = api.Cloud(sctx, synth.sample_city(boxes, c_ext, 2_500_000))
            ccloud.voxel_downsample(0.1, "pcl")
            cmap = api.Map(sctx, ccloud, args.cell)
            cmap.estimate_normals(normal_radius)
            crng = np.random.default_rng(77)
            truths_, cscans = [], []
            while len(cscans) < c_scans:
                xy = crng.uniform(-8.0, 8.0, 2)
                T_ = synth.make_T((xy[0], xy[1], 1.8), (0.0, 0.0, crng.uniform(0, 360)))
                s_ = synth.raycast_scan(boxes, T_, rings=c_rings, max_range=55.0, seed=synth.CITY_SEED + 10 + len(cscans) + len(truths_))
                truths_.append(T_)
                if len(s_) >= 0.4 * c_rings * 2032:
                    cscans.append((T_, s_))
            n_c = min(len(s_) for _, s_ in cscans)
            cs = np.stack([s_[crng.choice(len(s_), n_c, replace=False)] for _, s_ in cscans])
            cin = np.stack([T_ @ synth.make_T(crng.normal(0, 0.06, 3), crng.normal(0, 0.3, 3)) for T_, _ in cscans])
            cicp = api.Icp(sctx, max_dist, iters, 0.05, 1e-5)
            cicp.set_target(cmap)
            cicp.use_graph(not args.no_graph)
            cicp.set_source_batch(cs)
            cicp.set_initial_batch(cin)
            cres = cicp.align_batch("p2plane")            # (the freeze schedule is learnt from this one)
            cicp.align_batch("p2plane")
            t1 = time.perf_counter()
            for _ in range(5):
                cicp.align_batch_async("p2plane")
            sctx.synchronize()
            v_on = c_scans * 5 / (time.perf_counter() - t1)
            fs_ = cicp.freeze_stats()
            cicp.set_freeze(False)
            cicp.align_batch("p2plane")
            t1 = time.perf_counter()
            for _ in range(5):
                cicp.align_batch_async("p2plane")
            sctx.synchronize()
            v_off = c_scans * 5 / (time.perf_counter() - t1)
            extras["value_city"] = {
                "value": v_on, "value_no_freeze": v_off, "unit": "scans/s", "scans_in_flight": c_scans, "points_per_scan": int(n_c), "map_points": int(len(cmap)),
                "froze": int(fs_["froze"]), "thawed": int(fs_["thawed"]), "voided": int(fs_["failed"]), "active_share": fs_["active_queries"] / float(n_c * c_scans),
                "median_translation_err_vs_truth_m": float(np.median([synth.pose_error(r_["T64"], T_)[0] for r_, (T_, _) in zip(cres, cscans)])),
                "what": "ring scans (%d x 2032 rays, 55 m range) of a %d m synthetic city vs its surface map (voxel 0.1 m), %d in flight, every scan from its own pose with a "
                        "0.06 m / 0.3 deg prior error (1 sigma per axis), 20 point-to-plane iterations; scans this size freeze (above 131 072 points), a 64-ring scan would not -- "
                        "profiles/r04_city_bench*.jsonl hold 64 in flight against the 8.3 M-point city, both sensors, and a 0.3 m / 1.5 deg prior" % (c_rings, int(c_ext), c_scans)}
            cicp.close()

    # ---------------- roofline of the dominant kernel
    # achieved = bytes the memory system moves per launch / launch duration: from the rocprofv3 PMC passes of this same
    # command committed under profiles/ (request counts x request sizes, the gfx950 FETCH_SIZE x2 correction made explicit)
    # when the configuration matches, else from the compulsory-traffic model below.  The SURVEY §8d convention
    # (754 B per query-iteration whether or not a search runs) is reported beside it as an "effective" figure: the
    # implementation proves most searches unnecessary, so that figure is a speed-up measure, not a roofline fraction.
    a_nn = A_NN_P2PLANE if args.mode == "p2plane" else A_NN_P2P
    roof = None
    if prof is not None:
        sel_k = np.arange(prof["per"]) < prof["fz_from"]        # the launches of the dominant kernel itself (k_nn_red; from fz_from on: k_nn_red_fz)
        nn_ms = float(prof["ms"][sel_k].mean())
        cache_b = 32 if args.mode == "p2plane" else 20          # neighbour cache entry: (neighbour, index) + (normal, E) / + E alone
        map_b = len(mp) * (32 if args.mode == "p2plane" else 16) + 4.0 * np.prod(mp.cell_size()[1])   # every point (+ normal) line and the cell table once
        sel_s, sel_v = prof["searching"], ~prof["searching"]
        if args.mode == "ref_cpp":
            comp_search = prof["q_launch"] * (12 + 4) + map_b
            comp_verify = 0.0
        else:
            comp_search = prof["q_launch"] * (12 + (cache_b if not args.no_nn_reuse else 0)) + map_b
            comp_verify = prof["q_launch"] * (12 + cache_b)
        comp = np.where(sel_s, comp_search, comp_verify)
        traffic, traffic_src, valu, traffic_note = None, None, None, None
        src_hash = api.kernel_source_hash()

        def load_traffic(name):
            """profiles/<name> when it was taken on THIS build of the kernel and on this configuration, else None (+ why)."""
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", name)))
                c = tj["config"]
            except (OSError, KeyError, ValueError):
                return None, "profiles/%s not found" % name
            if tj.get("source_hash") != src_hash:
                return None, "profiles/%s was measured on another build of the kernel (source hash %s, now %s)" % (name, tj.get("source_hash"), src_hash)
            if not (world == 1 and c["batch"] == B and c["scan_points"] == n_scan and c["map_points"] == args.map_points and c["iters"] == iters and c["mode"] == args.mode):
                return None, "profiles/%s is for another configuration" % name
            return tj, None
        tj, traffic_note = load_traffic("r04_traffic.json" if not args.no_nn_reuse else "r04_search_traffic.json")
        if tj is not None and tj["config"].get("nn_reuse", True) == (not args.no_nn_reuse):
            traffic = tj["traffic_bytes_per_launch"]
            traffic_src = "profiles/%s (rocprofv3 --pmc TCC_EA0_RDREQ/WRREQ by request size, separate passes; source hash %s matches this build)" % (
                "r04_traffic.json" if not args.no_nn_reuse else "r04_search_traffic.json", src_hash)
            valu = tj.get("valu")
        bytes_launch = traffic if traffic is not None else float(comp[sel_k].mean())
        achieved = bytes_launch / (nn_ms * 1e-3) / 1e9 if nn_ms > 0 else 0.0

        def phase(sel):
            if not sel.any():
                return None
            return {"launches_per_alignment": int(sel.sum()), "avg_launch_us": float(prof["ms"][sel].mean() * 1e3),
                    "queries_searching_frac": float(prof["frac_search"][sel].mean()), "waves_searching_frac": float(prof["wave_frac"][sel].mean()),
                    "compulsory_bytes_per_launch": float(comp[sel].mean()),
                    "compulsory_gbs": float(comp[sel].mean() / (prof["ms"][sel].mean() * 1e-3) / 1e9),
                    "compulsory_frac_of_hbm_peak": float(comp[sel].mean() / (prof["ms"][sel].mean() * 1e-3) / 1e9 / HBM_PEAK_GBS)}
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                "achieved_from": "PMC traffic" if traffic is not None else
                                 "compulsory-traffic model (every touched map line once + query / neighbour-cache streams) -- no counter file applies: %s" % traffic_note,
                "kernel_source_hash": src_hash,
                # the SURVEY §8(d) figure under its own name: 742 / 754 algorithmic bytes per query-iteration x queries per launch / average launch
                # duration / peak.  It exceeds 1 because most searches are proven unnecessary and the rest are pruned and shared in L2
                "sec8d_frac": prof["q_launch"] * a_nn / (nn_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "kernel": "k_ref_nn" if args.mode == "ref_cpp" else "k_nn_red", "avg_launch_ms": nn_ms, "launches_per_alignment": int(sel_k.sum()),
                "launches_per_alignment_all_kernels": int(prof["per"]),
                "queries_per_launch": prof["q_launch"],
                "phases": {"searching": phase(sel_s & sel_k), "verifying": phase(sel_v & sel_k)},
                "per_launch_us": [round(float(v) * 1e3, 1) for v in prof["ms"]],
                "per_launch_kind": [str(v) for v in prof["kind"]],
                "per_launch_queries_searching_frac": [round(float(v), 4) for v in prof["frac_search"]],
                "effective_algorithmic": {"bytes_per_query": a_nn, "bytes_per_launch": prof["q_launch"] * a_nn,
                                          "gbs": prof["q_launch"] * a_nn / (nn_ms * 1e-3) / 1e9,
                                          "x_hbm_peak": prof["q_launch"] * a_nn / (nn_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                          "note": "SURVEY §8d convention: 754 B charged per query-iteration whether or not a search runs; > 1 x peak means searches were proven unnecessary or served from L2, it is not a roofline fraction"}}
        if valu is not None:
            # what binds a searching launch: vector-instruction issue (profiles/: SQ_INSTS_VALU per wave, SQ_ACTIVE_INST_VALU utilisation)
            roof["valu_issue"] = valu
        # the searching phase under its own name: counters of the same command with `--no-nn-reuse` (every query searches in every
        # launch), taken on this build of the kernel; and the verifying phase from this run's own launch times and its compulsory stream
        sj, _why = load_traffic("r04_search_traffic.json") if not args.no_nn_reuse else (None, None)
        if sj is not None and sj.get("avg_launch_ns_kernel_trace"):
            roof["searching_frac"] = {"frac": sj["frac_of_hbm_peak_kernel_trace"], "traffic_bytes_per_launch": sj["traffic_bytes_per_launch"],
                                      "avg_launch_us": sj["avg_launch_ns_kernel_trace"] * 1e-3, "valu_busy_frac": (sj.get("valu") or {}).get("valu_busy_frac"),
                                      "source": "profiles/r04_search_traffic.json: PMC traffic / rocprofv3 kernel-trace duration of launches in which every query searches"}
        sel_vk = sel_v & sel_k
        if sel_vk.any() and args.mode == "p2plane":   # (the other modes converge early: their later launches return at once)
            roof["verifying_frac"] = float(comp[sel_vk].mean() / (prof["ms"][sel_vk].mean() * 1e-3) / 1e9 / HBM_PEAK_GBS)
        if prof["fz"]["froze"] > 0:
            # the launches after the dominant kernel's: k_nn_red_fz.  The freeze launch streams what a verifying launch streams (44 B per
            # query) and forms 96 moment sums per pair instead of 30; a frozen launch touches the active queries only -- its time is the
            # latency of one search, not traffic: no roofline fraction is claimed for it
            fzl = np.arange(prof["per"]) >= prof["fz_from"]
            is_freeze = np.array([k_ == "freeze" for k_ in prof["kind"]])
            roof["frozen_pairs"] = {
                "kernel": "k_nn_red_fz", "launches_per_alignment": int(fzl.sum()), "sum_us_per_alignment": float(prof["ms"][fzl].sum() * 1e3),
                "freeze_launch_us": float(prof["ms"][is_freeze].mean() * 1e3) if is_freeze.any() else None,
                "freeze_launch_frac_of_hbm_peak": float(comp_verify / (prof["ms"][is_freeze].mean() * 1e-3) / 1e9 / HBM_PEAK_GBS) if is_freeze.any() else None,
                "frozen_launch_us": float(prof["ms"][fzl & ~is_freeze].mean() * 1e3) if (fzl & ~is_freeze).any() else None,
                "scans": B, **prof["fz"], "active_queries_frac": prof["fz"]["active_queries"] / float(prof["q_launch"]),
                "what": "from launch %d on the P2PLANE sums of the pairs certified to stay are evaluated from 96 moments (a polynomial in the pose, float64), "
                        "the active queries launch by launch; same pairs, sums equal to rounding (tests/test_gpu_freeze.py); --no-freeze / value_no_freeze: without" % (prof["fz_from"] + 1)}

    scans_total = B * args.steps
    value = scans_total / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    nn_per_scan = iters + (1 if args.mode == "o3d_p2p" else 0)
    out = {
        "metric": "scans_per_s_200k_scan_vs_10M_map",
        "value": value,
        "unit": "scans/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "ms_per_icp_iter": ms_per_step / nn_per_scan / B,
        "ms_per_icp_iter_batch": ms_per_step / nn_per_scan,
        "single_scan_latency_ms": extras.get("single_scan_latency_ms"),
        "value_no_reuse": extras.get("value_no_reuse"),
        "value_no_freeze": extras.get("value_no_freeze"),
        "value_no_pipeline": extras.get("value_no_pipeline"),
        "value_tile_search": extras.get("value_tile_search"),
        "value_city": extras.get("value_city"),
        "value_upload_inclusive": extras.get("value_upload_inclusive"),
        "value_stream_upload_inclusive": extras.get("value_stream_upload_inclusive"),
        "stream_results": {"delivered": extras.get("stream_results_delivered"), "all_equal": extras.get("stream_results_equal")},
        "value_32_in_flight": extras.get("value_32_in_flight"),
        "value_stream_config4": extras.get("value_stream_config4"),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32 points / f64 accumulation" if args.mode != "ref_cpp" else "f32 points and updates (reference arithmetic) / f64 accumulation",
        "data": "synthetic",
        "config": {"workload": "%dk-pt scans (%s) vs %.1fM-pt map (voxel 0.1 m -> %d pts), %d %s ICP iters, %s"
                               % (n_scan // 1000, "drawn from the whole map" if args.scan_kind == "whole" else "10 m neighbourhoods", args.map_points / 1e6, n_map, iters,
                                  args.mode, "lazy re-search (reference rule)" if args.mode == "ref_cpp" else
                                  ("exact NN result every iter; neighbour reuse %s" % ("off: every query searches in every iteration" if args.no_nn_reuse else
                                                                                     "on: a query whose neighbour provably cannot have changed skips its search (bit-identical results)" +
                                                                                     ("" if (args.no_freeze or args.mode != "p2plane") else
                                                                                      "; once a scan's pairs are certified to stay, their sums come from 96 moments (float64, equal to rounding)")))),
                   "mode": args.mode, "scans_in_flight": B, "scans_in_flight_per_gpu": B // world if args.scaling == "weak" else B, "cell_m": args.cell,
                   "max_corr_dist_m": max_dist, "nn_reuse": not args.no_nn_reuse, "frozen_pairs": bool((prof is not None and prof["fz"]["froze"] > 0) or (rank_stats is not None and (rank_stats.get("frozen_scans") or {}).get("max", 0) > 0)),
                   "scan_kind": args.scan_kind,
                   "parallelism": ("map sharded into %d x-slabs (+halo); %s; collective: %s"
                                   % (world, "every scan spans every slab: each rank owns 1/%d of every scan's queries, all-reduce of the normal-equation records once per ICP iteration" % world
                                      if args.scan_kind == "whole" else "scans routed to the slabs they touch, one-slab scans registered by one rank without a collective", comm_kind))
                   if sharded_run else "single GPU",
                   "hip_graph": (not args.no_graph) and not sharded_run,
                   "collective": (coll_info or {}).get("kind"),
                   "shard_resumes": (routed.resumes if routed is not None else 0),
                   "routing_groups": ({"%d-%d" % k: len(v) for k, v in routed.groups.items()} if routed is not None else None)},
        "delivery": delivery,
        "parity": {"max_translation_err_vs_truth_m": terr, "max_rotation_err_vs_truth_rad": rerr, "ok": bool(ok_truth)},
        "roofline": roof if roof is not None else roof_dist,
        "setup_s": setup_s,
    }
    if sharded_run:
        out["ranks"] = rank_stats
        out["collective"] = coll_info
        out["value_strong"] = strong
        out["value_replicas"] = replicas

    # ---------------- CPU baseline: the oracle (port of the reference path), 1 thread, rank 0, N=1 -- and the parity gate
    if rank == 0 and world == 1 and not sharded_run and not args.no_cpu_baseline:
        from oracle import oracle as orc
        tb = time.perf_counter()
        tree = orc.KdTreeD(map_ds.astype(np.float64))
        build_s = time.perf_counter() - tb
        del tree
        tc = time.perf_counter()
        if args.mode == "p2plane":
            normals, _ = mp.download_normals()
            tc = time.perf_counter()
            r = orc.icp_p2plane(scans[0], map_ds, normals, max_dist=max_dist, num_iters=iters)
            what = "oracle/icp.c orc_icp_p2plane (kd-tree leaf 15, float64)"
        elif args.mode == "o3d_p2p":
            r = orc.icp_o3d_p2p(scans[0], map_ds, max_dist=max_dist, max_iter=iters)
            what = "oracle/icp.c orc_icp_o3d_p2p (kd-tree leaf 15, float64; restates Open3D registration_icp)"
        else:
            r = orc.icp_ref_cpp(scans[0], map_ds, None, max_dist, iters, 0.05, 1e-5, precise=True)
            what = "oracle/icp.c orc_icp_ref_cpp (restates icp_point_to_point.cpp:185-254; kd-tree leaf 15)"
        cpu_s = max(time.perf_counter() - tc - build_s, 1e-9)
        dt, dr = synth.pose_error(results[0]["T64"], r["T"])
        out["cpu_baseline"] = {"value": 1.0 / cpu_s, "unit": "scans/s", "cores": 1, "kind": "port",
                               "sample": "1 scan (%d pts), %d ICP iterations of %s, kd-tree build (%.1f s) excluded like the GPU index build; host has %d cores"
                                         % (n_scan, r["iterations"], what, build_s, os.cpu_count()),
                               "gpu_vs_oracle_translation_m": dt, "gpu_vs_oracle_rotation_rad": dr,
                               "gpu_vs_oracle_iterations": [int(results[0]["iterations"]), int(r["iterations"])]}
        out["parity"]["gpu_vs_oracle_ok"] = bool(dt <= POSE_TOL_M and dr <= POSE_TOL_RAD and results[0]["iterations"] == r["iterations"])
        out["parity"]["ok"] = bool(out["parity"]["ok"] and out["parity"]["gpu_vs_oracle_ok"])
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0 and not out["parity"]["ok"]:
        print("bench.py: PARITY FAILED: %s" % json.dumps(out["parity"]), file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
