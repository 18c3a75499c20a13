#!/usr/bin/env python3
"""The bench's registration step on a SURFACE-structured workload (not BASELINE's metric config: an extra measurement):
ring scans (64 rings x 2032 azimuths, ray cast) of a synthetic city registered against its voxel-filtered map, 32 scans in
flight, 20 point-to-plane iterations, each scan from its own pose with a 0.1 m / 0.5 degree prior error.  Surfaces make
the map cells dense where they are occupied and the neighbour structure anisotropic -- the case the uniform-random map
of the metric config does not cover.  Prints one JSON line: scans/s with and without neighbour reuse, per-launch times.
   python tools/city_bench.py [--map-points 10000000] [--batch 32] [--steps 10]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_sensor_fusion_amd import api, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--map-points", type=int, default=10_000_000)
    ap.add_argument("--extent", type=float, default=240.0)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--mode", default="p2plane", choices=["p2plane", "o3d_p2p", "ref_cpp"])
    args = ap.parse_args()
    ctx = api.Context(0)
    boxes = synth.make_city(args.extent, int(120 * (args.extent / 240.0) ** 2))
    raw = synth.sample_city(boxes, args.extent, args.map_points)
    cloud = api.Cloud(ctx, raw)
    del raw
    cloud.voxel_downsample(0.1, "pcl")
    n_map = len(cloud)
    mp = api.Map(ctx, cloud, 0.25)
    if args.mode == "p2plane":
        mp.estimate_normals(0.25)
    rng = np.random.default_rng(77)
    truths, scans = [], []
    while len(scans) < args.batch:
        xy = rng.uniform(-12.0, 12.0, 2)
        T = synth.make_T((xy[0], xy[1], 1.8), (0.0, 0.0, rng.uniform(0, 360)))
        s = synth.raycast_scan(boxes, T, seed=synth.CITY_SEED + 10 + len(scans))
        if len(s) < 60000:
            continue
        truths.append(T)
        scans.append(s)
    n = min(len(s) for s in scans)
    scans = np.stack([s[rng.choice(len(s), n, replace=False)] for s in scans])     # a common length, uniformly thinned
    inits = np.stack([T @ synth.make_T(rng.normal(0, 0.06, 3), rng.normal(0, 0.3, 3)) for T in truths])
    icp = api.Icp(ctx, 0.5, args.iters, 0.05, 1e-5)
    icp.set_target(mp)
    icp.use_graph(True)
    icp.set_source_batch(scans)
    icp.set_initial_batch(inits)
    out = {}
    for reuse in (True, False):
        if args.mode == "ref_cpp" and not reuse:
            continue
        icp.set_nn_reuse(reuse)
        res = icp.align_batch(args.mode)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            icp.align_batch_async(args.mode)
        ctx.synchronize()
        out["scans_per_s" if reuse else "scans_per_s_no_reuse"] = args.batch * args.steps / (time.perf_counter() - t0)
        if reuse:
            errs = [synth.pose_error(r["T64"], T) for r, T in zip(res, truths)]
            out["max_translation_err_m"], out["max_rotation_err_rad"] = max(e[0] for e in errs), max(e[1] for e in errs)
            out["iterations"] = sorted(set(r["iterations"] for r in res))
            icp.use_graph(False)
            icp.profile_enable(True)
            icp.align_batch_async(args.mode)
            ctx.synchronize()
            ms, sq, sw = icp.profile_launches()
            icp.profile_enable(False)
            icp.use_graph(True)
            out["per_launch_us"] = [round(float(v) * 1e3, 1) for v in ms]
            out["per_launch_queries_searching_frac"] = [round(float(v) / (n * args.batch), 3) for v in sq]
    cell, dims = mp.cell_size()
    print(json.dumps(dict(workload="ring scans (64 x 2032 rays) vs a %.0f m synthetic city, %d samples -> %d map points (voxel 0.1 m), %d scans in flight x %d points, %d %s iterations, "
                                   "prior error 0.06 m / 0.3 deg (1 sigma per axis)" % (args.extent, args.map_points, n_map, args.batch, n, args.iters, args.mode),
                          cell_m=cell, grid=list(dims), **out)))


if __name__ == "__main__":
    main()
