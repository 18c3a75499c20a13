#!/usr/bin/env python3
"""Short view of a bench.py JSON line:  tools/show_bench.py gpurun_out/b.json"""
import json
import sys

t = open(sys.argv[1]).read().strip().splitlines()
j = json.loads(t[-1])
print("value %.0f %s  (%d GPU, %.3f ms/step)  no_freeze %s  no_reuse %s  32-in-flight %s  upload-incl %s  single %s ms" % (
    j["value"], j["unit"], j["n_gpus"], j["ms_per_step"], j.get("value_no_freeze"), j.get("value_no_reuse"), j.get("value_32_in_flight"), j.get("value_upload_inclusive"),
    j.get("single_scan_latency_ms")))
print("parity", j["parity"])
r = j.get("roofline") or {}
print("roofline frac %s achieved %s from %s" % (r.get("frac"), r.get("achieved"), r.get("achieved_from")))
for k in ("sec8d_frac", "searching_frac", "verifying_frac"):
    if k in r:
        print("  ", k, r[k])
if "per_launch_us" in r:
    print("   per_launch_us", r["per_launch_us"])
    print("   searching frac", r["per_launch_queries_searching_frac"])
if r.get("frozen_pairs"):
    f = r["frozen_pairs"]
    print("   frozen pairs: freeze launch %.0f us, frozen launch %.1f us, %d of %d scans frozen at the end, %d active queries (%.2f %%), voided %d, thawed %d" % (
        f["freeze_launch_us"] or 0, f["frozen_launch_us"] or 0, f["frozen_at_end"], f["scans"], f["active_queries"], 100 * f["active_queries_frac"], f["failed"], f["thawed"]))
if j.get("ranks"):
    for k, v in j["ranks"].items():
        if k not in ("per_rank", "note"):
            print("  ", k, v)
    print("   collective", j.get("collective"))
    print("   value_strong", j.get("value_strong"))
    print("   value_replicas", j.get("value_replicas"))
if "cpu_baseline" in j:
    print("cpu", j["cpu_baseline"]["value"], j["cpu_baseline"]["sample"][:80])
for k in ("value_stream_config4",):
    if k in j:
        print(k, j[k])
