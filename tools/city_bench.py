#!/usr/bin/env python3
"""The bench's registration step on a SENSOR-shaped workload (not BASELINE's metric config: an extra measurement): ring scans
(rings x 2032 azimuths, ray cast) of a synthetic city registered against its voxel-filtered surface map, `--batch` scans in
flight, 20 point-to-plane iterations, each scan from its own pose with its own prior error.  Surfaces make the occupied
map cells dense and the neighbour structure anisotropic, and a scan converges from a per-scan prior instead of the metric
config's common 0.1 m offset -- what the reference's callback sees (a cropped ring scan and a blended prior,
localization_node.cpp:292-305,329-337).

One JSON line per (rings, prior error) case: scans/s with the library's defaults, with the frozen pairs off, with the neighbour
reuse off; per-launch times, the share of queries that search per launch, and sf_icp_freeze_stats (froze / thawed / voided /
active share).  Frozen pairs need wide scans (include/slamfusion.h, sf_icp_set_wide_scan_points): above 131 072 points, or above
65 536 in a batch that no single launch could take -- a 64-ring scan (<= 130 048 returns) in a batch of 64 qualifies.
   python tools/city_bench.py [--map-points 10000000] [--batch 64] [--rings 64 128] [--prior 0.06:0.3 0.3:1.5]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_sensor_fusion_amd import api, synth  # noqa: E402


def timed(icp, ctx, mode, steps, batch):
    icp.align_batch(mode)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        icp.align_batch_async(mode)
    ctx.synchronize()
    return batch * steps / (time.perf_counter() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--map-points", type=int, default=10_000_000)
    ap.add_argument("--extent", type=float, default=240.0)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rings", type=int, nargs="+", default=[64, 128])
    ap.add_argument("--prior", nargs="+", default=["0.06:0.3", "0.3:1.5"], help="1-sigma prior error per axis, metres:degrees")
    ap.add_argument("--wide-from", type=int, default=0, help="sf_icp_set_wide_scan_points (0: the library's own rule -- above 131 072 points, or above 65 536 in a batch no single launch could take)")
    args = ap.parse_args()
    mode = "p2plane"
    ctx = api.Context(0)
    boxes = synth.make_city(args.extent, int(120 * (args.extent / 240.0) ** 2))
    raw = synth.sample_city(boxes, args.extent, args.map_points)
    cloud = api.Cloud(ctx, raw)
    del raw
    cloud.voxel_downsample(0.1, "pcl")
    n_map = len(cloud)
    mp = api.Map(ctx, cloud, 0.25)
    mp.estimate_normals(0.25)
    cell, dims = mp.cell_size()
    for rings in args.rings:
        rng = np.random.default_rng(77)
        truths, scans = [], []
        while len(scans) < args.batch:
            xy = rng.uniform(-12.0, 12.0, 2)
            T = synth.make_T((xy[0], xy[1], 1.8), (0.0, 0.0, rng.uniform(0, 360)))
            s = synth.raycast_scan(boxes, T, rings=rings, seed=synth.CITY_SEED + 10 + len(scans))
            if len(s) < 0.45 * rings * 2032:
                continue
            truths.append(T)
            scans.append(s)
        n = min(len(s) for s in scans)
        scans = np.stack([s[rng.choice(len(s), n, replace=False)] for s in scans])     # a common length, uniformly thinned
        for prior in args.prior:
            sig_t, sig_r = (float(v) for v in prior.split(":"))
            prng = np.random.default_rng(78)
            inits = np.stack([T @ synth.make_T(prng.normal(0, sig_t, 3), prng.normal(0, sig_r, 3)) for T in truths])
            icp = api.Icp(ctx, 0.5, args.iters, 0.05, 1e-5)
            icp.set_target(mp)
            icp.use_graph(True)
            if args.wide_from > 0:
                icp.set_wide_scan_points(args.wide_from)
            icp.set_source_batch(scans)
            icp.set_initial_batch(inits)
            out = {}
            res = icp.align_batch(mode)
            errs = [synth.pose_error(r["T64"], T) for r, T in zip(res, truths)]
            out["max_translation_err_m"], out["max_rotation_err_rad"] = max(e[0] for e in errs), max(e[1] for e in errs)
            out["median_translation_err_m"] = float(np.median([e[0] for e in errs]))
            out["scans_per_s"] = timed(icp, ctx, mode, args.steps, args.batch)
            fs = icp.freeze_stats()
            out["freeze_stats"] = dict(fs, active_share=fs["active_queries"] / float(n * args.batch), scans=args.batch, wide_from=args.wide_from or "library rule")
            icp.use_graph(False)
            icp.profile_enable(True)
            icp.align_batch_async(mode)
            ctx.synchronize()
            ms, sq, _ = icp.profile_launches()
            icp.profile_enable(False)
            icp.use_graph(True)
            out["per_launch_us"] = [round(float(v) * 1e3, 1) for v in ms]
            out["per_launch_queries_searching_frac"] = [round(float(v) / (n * args.batch), 4) for v in sq]
            icp.set_freeze(False)
            off = icp.align_batch(mode)
            out["scans_per_s_no_freeze"] = timed(icp, ctx, mode, args.steps, args.batch)
            out["freeze_vs_no_freeze_max_pose_diff_m"] = max(synth.pose_error(a["T64"], b["T64"])[0] for a, b in zip(res, off))
            icp.set_freeze("auto")
            icp.set_nn_reuse(False)
            out["scans_per_s_no_reuse"] = timed(icp, ctx, mode, max(2, args.steps // 2), args.batch)
            icp.close()
            print(json.dumps(dict(workload="ring scans (%d x 2032 rays) vs a %.0f m synthetic city, %d samples -> %d map points (voxel 0.1 m), %d scans in flight x %d points, "
                                           "%d %s iterations, prior error %.2f m / %.1f deg (1 sigma per axis)" % (rings, args.extent, args.map_points, n_map, args.batch, n, args.iters,
                                                                                                                  mode, sig_t, sig_r),
                                  rings=rings, prior_sigma_m=sig_t, prior_sigma_deg=sig_r, points_per_scan=int(n), cell_m=cell, grid=list(dims), **out)), flush=True)


if __name__ == "__main__":
    main()
