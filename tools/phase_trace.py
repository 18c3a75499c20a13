#!/usr/bin/env python3
"""Where a searching wave of k_nn_red spends its lifetime (diagnostic build -DSF_PHASE_TRACE copied over the library):
   tools/phase_trace.py [--iters 1|20] [--batch 64] [--reuse]
prints, per phase, the share of the summed wave lifetimes (s_memtime ticks) and the task / trip counters per wave."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--map-points", type=int, default=10_000_000)
ap.add_argument("--scan-points", type=int, default=200_000)
ap.add_argument("--cell", type=float, default=0.25)
ap.add_argument("--reuse", action="store_true")
ap.add_argument("--tile", action="store_true", help="phases of k_tile_search instead (first wave of every workgroup)")
args = ap.parse_args()
import torch  # noqa: F401
from slam_sensor_fusion_amd import api, synth

ctx = api.Context(0)
raw = synth.make_map(args.map_points)
cloud = api.Cloud(ctx, raw)
cloud.voxel_downsample(0.1, "pcl")
ds = cloud.download()
mp = api.Map(ctx, cloud, args.cell)
mp.estimate_normals(0.25)
scans = [synth.make_scan(ds, args.scan_points, scan_id=b)[0] for b in range(args.batch)]
n = min(len(s) for s in scans)
scans = np.stack([s[:n] for s in scans])
icp = api.Icp(ctx, 0.5, args.iters, 0.05, 1e-5)
icp.set_target(mp)
icp.use_graph(False)
icp.set_nn_reuse(args.reuse)
icp.set_freeze(False)
icp.set_tile_search("always" if args.tile else False)
icp.set_source_batch(scans)
icp.set_initial_batch(None)
lib = api.load_library()
out = (C.c_ulonglong * 16)()
icp.align_batch("p2plane")
lib.sf_icp_phase_trace(out)  # warm-up discarded
icp.align_batch("p2plane")
if args.tile:
    lib.sf_icp_tile_trace(out)
    icp.align_batch("p2plane")
    assert lib.sf_icp_tile_trace(out) == 0
    v = np.array(list(out), dtype=np.float64)
    wg = v[7]
    tot = v[:4].sum()
    print("k_tile_search, iters %d batch %d reuse %s: %d workgroups with queries, %.0f ticks per workgroup" % (args.iters, args.batch, args.reuse, wg, tot / wg))
    for i, nm in enumerate(["0 segment tables + poses", "1 staging the tile", "2 queries (load, transform, search, store)", "3 last barrier"]):
        print("  %-48s %6.1f %%  %8.0f ticks/workgroup" % (nm, 100 * v[i] / tot, v[i] / wg))
    sys.exit(0)
assert lib.sf_icp_phase_trace(out) == 0
v = np.array(list(out), dtype=np.float64)
names = ["0 load+transform+certificate", "1 prologue (box, geometry, row bounds back)", "2 own-cell trip", "3 masks, LDS, task queue", "4 task rounds",
         "5 epilogue (winner fetch, ring test, ring>=2)", "6 normal gather + cache write", "7 pair terms + wave reductions", "8 barrier + record store"]
waves = v[14]
tot = v[:9].sum()
print("iters %d batch %d reuse %s: %d waves, %.0f ticks per wave" % (args.iters, args.batch, args.reuse, waves, tot / waves))
for i, nm in enumerate(names):
    print("  %-48s %6.1f %%  %8.0f ticks/wave" % (nm, 100 * v[i] / tot, v[i] / waves))
print("  per wave: tasks %.1f, rounds %.2f, wave-level trips in rounds %.2f, lane trips %.1f, candidates in tasks %.1f" % (v[11] / waves, v[9] / waves, v[10] / waves, v[12] / waves, v[13] / waves))
