#!/usr/bin/env python3
"""profiles/<round>_traffic.json from the request-size PMC summaries (tools/profile_round.sh).
usage: make_traffic_json.py profiles/r01 <scans in flight>"""
import json
import sys

prefix = sys.argv[1]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
rd = json.load(open(prefix + "_rdsz_pmc.json"))
wr = json.load(open(prefix + "_wrsz_pmc.json"))
key = max((k for k in rd if k.startswith("k_nn_red")), key=lambda k: rd[k]["TCC_EA0_RDREQ_sum"]["avg"])   # the batched launches
r, w = rd[key], wr[key]
n32, n64, n128, nall = (r["TCC_EA0_RDREQ_%s" % s]["avg"] for s in ("32B_sum", "64B_sum", "128B_sum", "sum"))
other = nall - n32 - n64 - n128
read_b = 32 * n32 + 64 * n64 + 128 * n128 + 64 * max(other, 0.0)
w64, wall = w["TCC_EA0_WRREQ_64B_sum"]["avg"], w["TCC_EA0_WRREQ_sum"]["avg"]
write_b = 64 * w64 + 32 * max(wall - w64, 0.0)
q = 200000 * batch
out = {
    "kernel": key.split(" grid=")[0],
    "config": {"scan_points": 200000, "map_points": 10000000, "batch": batch, "iters": 20, "mode": "p2plane"},
    "queries_per_launch": q,
    "read_requests_per_launch": {"32B": n32, "64B": n64, "128B": n128, "all": nall},
    "read_bytes_per_launch": read_b,
    "write_bytes_per_launch": write_b,
    "traffic_bytes_per_launch": read_b + write_b,
    "traffic_bytes_per_query": (read_b + write_b) / q,
    "method": "rocprofv3 --pmc TCC_EA0_RDREQ_{sum,32B,64B,128B}_sum and TCC_EA0_WRREQ_{sum,64B}_sum in separate passes (no tracing flags) on "
              "`python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph`; bytes = sum(requests x request size), averaged over the "
              "batched k_nn_red launches. Practically all read requests are 128-byte requests, so FETCH_SIZE (requests x 64 B) under-reports this "
              "kernel by the factor 2 that MI355X_MICROARCH.md (HBM section) gives for gfx950; WRITE_SIZE is exact. These are the memory-side "
              "requests of the L2 (Infinity-Cache hits included).",
    "source": [prefix + s for s in ("_rdsz_pmc.json", "_wrsz_pmc.json", "_fetch_pmc.json", "_write_pmc.json")],
}
json.dump(out, open(prefix + "_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
