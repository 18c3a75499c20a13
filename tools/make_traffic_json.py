#!/usr/bin/env python3
"""profiles/<prefix>_traffic.json from the PMC summaries of tools/profile_round.sh: memory-side traffic of the dominant
kernel by request size, and its vector-issue figures.
usage: make_traffic_json.py profiles/r02 <scans in flight> <mode> <iters> <reuse|noreuse>"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_sensor_fusion_amd.api import kernel_source_hash  # noqa: E402

prefix = sys.argv[1]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
mode = sys.argv[3] if len(sys.argv) > 3 else "p2plane"
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
reuse = (sys.argv[5] if len(sys.argv) > 5 else "reuse") == "reuse"
kname = "k_ref_nn" if mode == "ref_cpp" else "k_nn_red"
rd = json.load(open(prefix + "_rdsz_pmc.json"))
wr = json.load(open(prefix + "_wrsz_pmc.json"))
mine = lambda k: k.startswith(kname) and not k.startswith("k_nn_red_fz")   # (k_nn_red_fz: the launches after the freeze, reported apart)
top = max(rd[k]["TCC_EA0_RDREQ_sum"]["avg"] for k in rd if mine(k))
keys = sorted(k for k in rd if mine(k) and rd[k]["TCC_EA0_RDREQ_sum"]["avg"] >= 0.1 * top)   # the batched launches of every variant of the kernel
key = keys[0]                                                                                   # (one or two queries per lane: same kernel, two instantiations)


def wavg(tab, counter):
    """average per launch over all variants, weighted by their launches"""
    n = sum(tab[k][counter]["dispatches"] for k in keys)
    return sum(tab[k][counter]["avg"] * tab[k][counter]["dispatches"] for k in keys) / n


n32, n64, n128, nall = (wavg(rd, "TCC_EA0_RDREQ_%s" % s) for s in ("32B_sum", "64B_sum", "128B_sum", "sum"))
other = nall - n32 - n64 - n128
read_b = 32 * n32 + 64 * n64 + 128 * n128 + 64 * max(other, 0.0)
w64, wall = wavg(wr, "TCC_EA0_WRREQ_64B_sum"), wavg(wr, "TCC_EA0_WRREQ_sum")
write_b = 64 * w64 + 32 * max(wall - w64, 0.0)
q = 200000 * batch
valu = None
try:
    sq_all = json.load(open(prefix + "_sq_pmc.json"))
    sq2_all = json.load(open(prefix + "_sq2_pmc.json"))
    sq = {c: {"avg": wavg(sq_all, c)} for c in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS")}
    sq2 = {c: {"avg": wavg(sq2_all, c)} for c in ("GRBM_GUI_ACTIVE", "SQ_ACTIVE_INST_VALU")}
    waves = sq["SQ_WAVES"]["avg"]
    cycles = sq2["GRBM_GUI_ACTIVE"]["avg"] / 8.0                       # rocprofv3 sums the 8 XCDs
    n_simd = 1024
    util = sq2["SQ_ACTIVE_INST_VALU"]["avg"] * 4.0 / (n_simd * cycles)  # SQ_ACTIVE_INST_* count quad-cycles
    valu = {"valu_instructions_per_wave": sq["SQ_INSTS_VALU"]["avg"] / waves, "salu_instructions_per_wave": sq["SQ_INSTS_SALU"]["avg"] / waves,
            "vmem_read_instructions_per_wave": sq["SQ_INSTS_VMEM_RD"]["avg"] / waves, "lds_instructions_per_wave": sq["SQ_INSTS_LDS"]["avg"] / waves,
            "waves_per_launch": waves, "kernel_cycles": cycles,
            "valu_busy_frac": util,
            "note": "valu_busy_frac = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): the share of the SIMDs' cycles spent issuing "
                    "vector instructions, averaged over the launches of an alignment -- the roofline that binds a searching launch"}
except (OSError, KeyError):
    pass
avg_ns = None
try:
    tr = json.load(open(prefix + "_final_kernel_trace_by_grid.json"))
    tks = [k for k in tr if mine(k) and tr[k]["avg_ns"] * tr[k]["launches"] >= 0.02 * max(tr[j]["avg_ns"] * tr[j]["launches"] for j in tr if mine(j))]
    avg_ns = sum(tr[k]["avg_ns"] * tr[k]["launches"] for k in tks) / sum(tr[k]["launches"] for k in tks)
except (OSError, KeyError, ValueError):
    pass
def side_kernel(prefix_name):
    """traffic and kernel-trace duration of another kernel of the same runs (None if it did not run)"""
    try:
        fk = max((k for k in rd if k.startswith(prefix_name)), key=lambda k: rd[k]["TCC_EA0_RDREQ_sum"]["avg"])
        fr, fw = rd[fk], wr[fk]
        f_read = sum(sz * fr["TCC_EA0_RDREQ_%s" % nm]["avg"] for sz, nm in ((32, "32B_sum"), (64, "64B_sum"), (128, "128B_sum")))
        f_write = 64 * fw["TCC_EA0_WRREQ_64B_sum"]["avg"] + 32 * max(fw["TCC_EA0_WRREQ_sum"]["avg"] - fw["TCC_EA0_WRREQ_64B_sum"]["avg"], 0.0)
        ft = max((k for k in tr if k.startswith(prefix_name)), key=lambda k: tr[k]["launches"] * tr[k]["avg_ns"])
        return {"kernel": fk.split(" grid=")[0], "traffic_bytes_per_launch_avg": f_read + f_write, "avg_launch_ns_kernel_trace": tr[ft]["avg_ns"],
                "min_launch_ns": tr[ft]["min_ns"], "max_launch_ns": tr[ft]["max_ns"], "launches_traced": tr[ft]["launches"]}
    except (ValueError, KeyError, NameError):
        return None


# the frozen pairs: the launch in which scans freeze (k_nn_red_fz, full grid) and the launches after it (k_nn_red_fz_few)
fz = {"freeze_launch": side_kernel("k_nn_red_fz<"), "frozen_launches": side_kernel("k_nn_red_fz_few")}
if fz["freeze_launch"] is None and fz["frozen_launches"] is None:
    fz = None
out = {
    "kernel": " + ".join(k.split(" grid=")[0] for k in keys),
    "launches_profiled": {k.split(" grid=")[0]: rd[k]["TCC_EA0_RDREQ_sum"]["dispatches"] for k in keys},
    "frozen_pairs_kernel": fz,
    "avg_launch_ns_kernel_trace": avg_ns,
    "frac_of_hbm_peak_kernel_trace": ((read_b + write_b) / (avg_ns * 1e-9) / 8e12) if avg_ns else None,
    "config": {"scan_points": 200000, "map_points": 10000000, "batch": batch, "iters": iters, "mode": mode, "nn_reuse": reuse},
    "source_hash": kernel_source_hash(),
    "queries_per_launch": q,
    "read_requests_per_launch": {"32B": n32, "64B": n64, "128B": n128, "all": nall},
    "read_bytes_per_launch": read_b,
    "write_bytes_per_launch": write_b,
    "traffic_bytes_per_launch": read_b + write_b,
    "traffic_bytes_per_query": (read_b + write_b) / q,
    "valu": valu,
    "method": "rocprofv3 --pmc TCC_EA0_RDREQ_{sum,32B,64B,128B}_sum and TCC_EA0_WRREQ_{sum,64B}_sum in separate passes (no tracing flags) on "
              "`python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-extras [...]`; bytes = sum(requests x request size), averaged over the "
              "batched launches of the kernel. Practically all read requests are 128-byte requests, so FETCH_SIZE (requests x 64 B) under-reports this "
              "kernel by the factor 2 that MI355X_MICROARCH.md (HBM section) gives for gfx950; WRITE_SIZE is exact. These are the memory-side "
              "requests of the L2 (Infinity-Cache hits included).",
    "source": [prefix + s for s in ("_rdsz_pmc.json", "_wrsz_pmc.json", "_fetch_pmc.json", "_write_pmc.json", "_sq_pmc.json", "_sq2_pmc.json")],
}
json.dump(out, open(prefix + "_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
