#!/usr/bin/env python3
"""Host time of every call of the streaming loop (set source from pinned memory, enqueue, fetch previous): where a step's time goes.
   python tools/probes/stream_timing.py [steps]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from slam_sensor_fusion_amd import api
from slam_sensor_fusion_amd import synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ctx = api.Context(0)
raw = synth.make_map(3_000_000)
cloud = api.Cloud(ctx, raw); cloud.voxel_downsample(0.1, "pcl"); ds = cloud.download()
mp = api.Map(ctx, cloud, 0.25); mp.estimate_normals(0.25)
B, n = 32, 200_000
scans = np.stack([synth.make_scan(ds, n, scan_id=b)[0][:n] for b in range(B)])
pin = [torch.from_numpy(scans).pin_memory(), torch.from_numpy(scans.copy()).pin_memory()]
icp = api.Icp(ctx, 0.5, 20, 0.05, 1e-5); icp.set_target(mp); icp.use_graph(True)
icp.set_source_batch_host_ptr(pin[0].data_ptr(), n, B); icp.set_initial_batch(None); icp.align_batch("p2plane")
for rep in range(2):
    t0 = time.perf_counter(); rows = []
    for s in range(steps):
        a = time.perf_counter(); icp.set_source_batch_host_ptr(pin[s % 2].data_ptr(), n, B)
        b = time.perf_counter(); icp.align_batch_async("p2plane")
        c = time.perf_counter()
        if s > 0: icp.fetch_previous(raw=True)
        d = time.perf_counter(); rows.append((b - a, c - b, d - c))
    icp.fetch_results(raw=True)
    tot = time.perf_counter() - t0
    print("rep", rep, "ms/step %.3f" % (tot / steps * 1e3), "scans/s %.0f" % (B * steps / tot))
    for r in rows: print("   set_source %.3f  enqueue %.3f  fetch_previous %.3f ms" % tuple(x * 1e3 for x in r))
# the same without a new source per step
t0 = time.perf_counter()
for s in range(steps):
    icp.align_batch_async("p2plane")
    if s > 0: icp.fetch_previous(raw=True)
icp.fetch_results(raw=True)
tot = time.perf_counter() - t0
print("same source: ms/step %.3f" % (tot / steps * 1e3))
