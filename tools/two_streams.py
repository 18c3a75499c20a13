"""Throughput of several batches side by side: NS contexts (streams) each registering B 200 k-point scans against ONE 10 M-point
map, steps issued alternately (DESIGN.md §3, round-3 ladder).  gpurun -- python tools/two_streams.py"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_sensor_fusion_amd import api, synth
torch.cuda.set_device(0)
streams = [torch.cuda.Stream() for _ in range(4)]
ctxs = [api.Context(0, s.cuda_stream) for s in streams]
ctx = ctxs[0]
raw = synth.make_map(10_000_000)
cloud = api.Cloud(ctx, raw)
cloud.voxel_downsample(0.1, "pcl")
map_ds = cloud.download()
mp = api.Map(ctx, cloud, 0.25)
mp.estimate_normals(0.25)
scans_all = np.stack([synth.make_scan(map_ds, 200_000, scan_id=b)[0] for b in range(128)])
ctx.synchronize()
for B, NS in ((64, 1), (64, 2), (32, 2), (32, 4), (128, 1)):
    icps = []
    for s in range(NS):
        icp = api.Icp(ctxs[s], 0.5, 20, 0.05, 1e-5)
        icp.set_target(mp)
        icp.use_graph(True)
        icp.set_source_batch(scans_all[(s * B) % 128:(s * B) % 128 + B] if B < 128 else scans_all)
        icp.set_initial_batch(None)
        icps.append(icp)
    for _ in range(2):
        for icp in icps:
            icp.align_batch_async("p2plane")
    torch.cuda.synchronize()
    K = 8
    t0 = time.perf_counter()
    for _ in range(K):
        for icp in icps:
            icp.align_batch_async("p2plane")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = icps[-1].fetch_results()
    err = max(synth.pose_error(r["T64"], synth.t_true())[0] for r in res)
    print("B %d x %d streams: %.1f scans/s, %.3f ms per batch of %d, max err %.2e" % (B, NS, B * NS * K / dt, dt / K / NS * 1e3, B, err), flush=True)
    for icp in icps:
        icp.close()
