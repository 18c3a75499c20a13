#!/usr/bin/env python3
"""bench.py — scan-to-map registration throughput on MI355X.

Metric (BASELINE.json): scans/s and ms/ICP-iteration, 200k-point scans vs a 10M-point map
(uniform-random synthetic), 20 point-to-plane ICP iterations per scan,
exact NN every iteration.  A "step" = one batch of `--batch` scans pushed through the whole
ICP (20 x [fused transform+NN+accumulate kernel, reduce+solve kernel]) with scans and map
already resident in HBM.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

N > 1: the map is tile-sharded along x (equal-count slabs + halo); `--batch` scans are in flight
PER GPU (N x 32 in all), every rank holds the whole scan batch, accumulates only the queries
that fall in its slab (1/N of every scan), and the 30-double normal-equation records are
all-reduced over RCCL once per ICP iteration; `--pipeline 2` steps the batch as two
halves on two streams so that one half's all-reduce (latency-bound) overlaps the other half's
search.  Work per GPU is fixed as N grows => "scaling": "weak"; `value` = all scans of all ranks'
common batch / wall time.
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--map-points", type=int, default=10_000_000)
    ap.add_argument("--scan-points", type=int, default=200_000)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32, help="scans registered concurrently per step, per GPU")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="sharded runs: step the batch as this many parts on separate streams so that one part's all-reduce overlaps the "
                         "next part's search (sharded.PipelinedShardedIcp).  With one rank 2 parts cost 17 %% (smaller kernels, twice the "
                         "host calls); whether it pays with 8 ranks depends on the RCCL latency, to be decided on a multi-GPU measurement")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the sharded stepping path + collective even with one rank (rehearses the RCCL plumbing on one GPU)")
    ap.add_argument("--mode", default="p2plane", choices=["p2plane", "o3d_p2p"])
    ap.add_argument("--cell", type=float, default=0.25)
    ap.add_argument("--query-order", default="auto", choices=["auto", "as_given", "cell"], help="sf_icp_set_query_order")
    ap.add_argument("--no-nn-reuse", action="store_true", help="sf_icp_set_nn_reuse(0): search every query in every iteration")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--cpu-baseline-iters", type=int, default=20)
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, default); gloo only to rehearse the N>1 path with several ranks on ONE GPU")
    return ap.parse_args()


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
    dist = None
    sharded_run = world > 1 or args.force_dist
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
    mp.estimate_normals(normal_radius)
    B = args.batch * world                        # weak scaling: --batch scans in flight per GPU
    scans = np.stack([synth.make_scan(map_ds, args.scan_points, scan_id=rank * 0 + b)[0] for b in range(B)])
    n_scan = scans.shape[1]
    parts = max(1, min(args.pipeline, B)) if sharded_run else 1
    while B % parts:
        parts -= 1
    streams = [stream] + [torch.cuda.Stream() for _ in range(parts - 1)]
    ctxs = [ctx] + [api.Context(device, st.cuda_stream) for st in streams[1:]]
    icps, xbufs = [], []
    for h in range(parts):
        part = api.Icp(ctxs[h], max_dist, args.iters, 0.05, 1e-5)
        part.set_target(mp)                                # the map index is shared (read-only) by the parts
        part.set_source_batch(scans[h * (B // parts):(h + 1) * (B // parts)])
        part.set_initial_batch(None)
        part.use_graph(not args.no_graph)
        part.set_query_order(args.query_order)
        part.set_nn_reuse(not args.no_nn_reuse)
        icps.append(part)
    icp = icps[0]
    drv = None
    if sharded_run:
        lo, hi = float(edges[rank]), float(edges[rank + 1])
        pairs = []
        for h, part in enumerate(icps):
            part.set_shard(max(lo, -1e30), min(hi, 1e30))   # finite bounds keep the sharded code path even for one rank
            xb = torch.zeros((B // parts) * 32, dtype=torch.float64, device="cuda")
            part.set_exchange_buffer(xb.data_ptr(), xb.numel() * 8)
            xbufs.append(xb)

            def allreduce(xb=xb, st=streams[h]):
                with torch.cuda.stream(st):
                    dist.all_reduce(xb)
            pairs.append((part, allreduce))
        drv = sharded.PipelinedShardedIcp(pairs, args.mode, args.iters)
    setup_s = time.time() - t_setup

    def step():
        if drv is None:
            icp.align_batch_async(args.mode)
        else:
            drv.align()       # blocking: the driver checks for scans that must be resumed (sharded.py)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    results = [r for part in icps for r in part.fetch_results()]

    # ---------------- correctness of what was timed: every scan must recover T_true
    terr = max(synth.pose_error(r["T64"], synth.t_true())[0] for r in results)
    rerr = max(synth.pose_error(r["T64"], synth.t_true())[1] for r in results)
    ok = all(r["iterations"] == args.iters or args.mode != "p2plane" for r in results) and terr < 5e-3

    # ---------------- roofline of the dominant kernel (fused transform+NN+accumulate), HIP events
    icp.use_graph(False)
    icp.profile_enable(True)
    prof_steps = max(2, min(5, args.steps))
    for _ in range(prof_steps):
        step()
    torch.cuda.synchronize()
    n_launch, ms_total = icp.profile_read()
    icp.profile_enable(False)
    a_nn = A_NN_P2PLANE if args.mode == "p2plane" else A_NN_P2P
    nn_ms = ms_total / max(n_launch, 1)
    queries_per_launch = n_scan * (B // parts) / (world if world > 1 else 1)   # the profiled part's launches
    achieved_gbs = queries_per_launch * a_nn / (nn_ms * 1e-3) / 1e9 if nn_ms > 0 else 0.0

    # ---------------- single-scan latency (one scan in flight, graph replay), outside the timed region
    single_ms = None
    if world == 1 and not sharded_run:
        lat = api.Icp(ctx, max_dist, args.iters, 0.05, 1e-5)
        lat.set_target(mp)
        lat.set_source(scans[0])
        lat.use_graph(not args.no_graph)
        lat.set_query_order(args.query_order)
        lat.set_nn_reuse(not args.no_nn_reuse)
        lat.align(args.mode)
        tl = time.perf_counter()
        for _ in range(10):
            lat.align(args.mode)
        single_ms = (time.perf_counter() - tl) / 10 * 1e3

    # ---------------- HBM-side traffic of the dominant kernel: PMC counters cannot be read from inside this
    # process, so the figure comes from the rocprofv3 --pmc passes of this same command committed under
    # profiles/ (request counts x request sizes, i.e. with the gfx950 FETCH_SIZE x2 correction made
    # explicit); reported only when the configuration matches the profiled one, otherwise null.
    traffic, traffic_src = None, None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
        c = tj["config"]
        if (world == 1 and c["batch"] == B and c["scan_points"] == n_scan and c["map_points"] == args.map_points
                and c["iters"] == args.iters and c["mode"] == args.mode):
            traffic = tj["traffic_bytes_per_launch"]
            traffic_src = "profiles/r01_traffic.json (rocprofv3 --pmc TCC_EA0_RDREQ/WRREQ by request size, separate passes)"
    except (OSError, KeyError, ValueError):
        pass

    scans_total = B * args.steps
    value = scans_total / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    nn_per_scan = args.iters + (1 if args.mode == "o3d_p2p" else 0)
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
        "single_scan_latency_ms": single_ms,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32 points / f64 accumulation",
        "data": "synthetic",
        "config": {"workload": "%dk-pt scans vs %.1fM-pt map (voxel 0.1 m -> %d pts), %d %s ICP iters, NN every iter"
                               % (n_scan // 1000, args.map_points / 1e6, n_map, args.iters, args.mode),
                   "scans_in_flight": B, "scans_in_flight_per_gpu": args.batch, "cell_m": args.cell, "max_corr_dist_m": max_dist,
                   "parallelism": ("map sharded into %d x-slabs (+halo), every rank owns 1/%d of each scan's queries, RCCL all-reduce of the "
                                   "normal-equation records once per ICP iteration" % (world, world)) if sharded_run else "single GPU",
                   "hip_graph": (not args.no_graph) and not sharded_run,
                   "shard_resumes": (drv.resumes if drv is not None else 0), "pipeline_parts": parts},
        "parity": {"max_translation_err_vs_truth_m": terr, "max_rotation_err_vs_truth_rad": rerr, "ok": bool(ok)},
        "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "bytes per launch",
                     "traffic_source": traffic_src, "algorithmic_bytes_per_launch": queries_per_launch * a_nn,
                     "kernel": "k_nn_red", "avg_launch_ms": nn_ms, "launches_timed": n_launch,
                     "algorithmic_bytes_per_query": a_nn, "queries_per_launch": queries_per_launch},
        "setup_s": setup_s,
    }

    # ---------------- CPU baseline: the oracle (port of the reference path), 1 thread, rank 0, N=1
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc
        normals, _ = mp.download_normals()
        tb = time.perf_counter()
        tree = orc.KdTreeD(map_ds.astype(np.float64))
        build_s = time.perf_counter() - tb
        del tree
        it = args.cpu_baseline_iters
        tc = time.perf_counter()
        if args.mode == "p2plane":
            r = orc.icp_p2plane(scans[0], map_ds, normals, max_dist=max_dist, num_iters=it)
        else:
            r = orc.icp_o3d_p2p(scans[0], map_ds, max_dist=max_dist, max_iter=it)
        cpu_s = time.perf_counter() - tc - build_s
        cpu_s = max(cpu_s, 1e-9)
        cpu_scans_per_s = 1.0 / (cpu_s * args.iters / max(r["iterations"], 1))
        dt, dr = synth.pose_error(results[0]["T64"], r["T"])
        out["cpu_baseline"] = {"value": cpu_scans_per_s, "unit": "scans/s", "cores": 1, "kind": "port",
                               "sample": "1 scan (%d pts) x %d ICP iterations of oracle/icp.c (kd-tree leaf 15, float64), "
                                         "kd-tree build (%.1f s) excluded like the GPU index build; host has %d cores"
                                         % (n_scan, r["iterations"], build_s, os.cpu_count()),
                               "gpu_vs_oracle_translation_m": dt, "gpu_vs_oracle_rotation_rad": dr}
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
