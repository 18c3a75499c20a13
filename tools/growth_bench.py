#!/usr/bin/env python3
"""A growth step of the map at size (`*map_cloud += *cloud` + VoxelGrid + setTargetPointCloud: global_map_frames_manager.cpp:131,
142-146, icp_point_to_point.cpp:49-55): sf_cloud_voxel_merge followed by sf_map_build, against the same merge followed by
sf_map_patch.  The map is a voxel-filtered uniform volume of --map-points raw points; every step adds --scans registered
scans of --scan-points points, half of them re-observing the map and half beyond its +x face.  One JSON line.
   python tools/growth_bench.py [--map-points 20000000] [--steps 12]"""
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
    ap.add_argument("--map-points", type=int, default=20_000_000)
    ap.add_argument("--scans", type=int, default=10)
    ap.add_argument("--scan-points", type=int, default=20_000)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--cell", type=float, default=0.25)
    ap.add_argument("--merge-min", type=int, default=-1, help="sf_cloud_voxel_merge_min_points (map size from which the filter merges; -1: the library's default)")
    args = ap.parse_args()
    ctx = api.Context(0)
    if args.merge_min >= 0:
        api.voxel_merge_min_points(args.merge_min)
    raw = synth.make_map(args.map_points)
    L = float(raw[:, 0].max())
    out = {}
    for how in ("build", "patch"):
        rng = np.random.default_rng(3)
        cloud = api.Cloud(ctx, raw)
        cloud.voxel_downsample(0.1, "pcl")
        n0 = len(cloud)
        mp = api.Map(ctx, cloud, args.cell)
        ds = None
        t_merge, t_index, patched, n_merged = [], [], 0, 0
        for k in range(args.steps + 2):
            m = args.scans * args.scan_points
            seen = (np.stack([rng.uniform(L - 6.0, L - 1.0, m // 2), rng.uniform(-L + 1, L - 1, m // 2), rng.uniform(-4.5, 4.5, m // 2)], 1)).astype(np.float32)
            new = (np.stack([rng.uniform(L - 1.0, L + 0.05 * (k + 1), m - m // 2), rng.uniform(-L + 1, L - 1, m - m // 2), rng.uniform(-4.5, 4.5, m - m // 2)], 1)).astype(np.float32)
            pending = api.Cloud(ctx, np.concatenate([seen, new]))
            ctx.synchronize()
            t0 = time.perf_counter()
            _, merged = cloud.voxel_merge(pending, 0.1)
            ctx.synchronize()
            t1 = time.perf_counter()
            if how == "build":
                mp.build(cloud, args.cell)
            else:
                patched += int(mp.patch(cloud))
            ctx.synchronize()
            t2 = time.perf_counter()
            n_merged += int(merged)
            if k >= 2:
                t_merge.append((t1 - t0) * 1e3)
                t_index.append((t2 - t1) * 1e3)
        out[how] = dict(merge_ms_median=float(np.median(t_merge)), index_ms_median=float(np.median(t_index)), step_ms_median=float(np.median(np.add(t_merge, t_index))),
                        step_ms_max=float(np.max(np.add(t_merge, t_index))), merged_steps=n_merged, patched_steps=patched, map_points_start_end=[int(n0), int(len(cloud))])
        if how == "patch":                                            # the index after the last step equals a build of the same cloud
            a, b = mp.index(), api.Map(ctx, cloud, args.cell).index()
            out["patched_index_equals_build"] = bool(all(np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)) for k in ("pts4", "cell_start")))
        cell, dims = mp.cell_size()
        out["cell_m"], out["grid"] = cell, list(dims)
        del mp, cloud
    print(json.dumps(dict(workload="growth step: %d scans x %d points merged into a voxel-filtered map (leaf 0.1 m), index cell %.2f m; host clock around the calls, %d steps"
                                   % (args.scans, args.scan_points, args.cell, args.steps), **out)))


if __name__ == "__main__":
    main()
