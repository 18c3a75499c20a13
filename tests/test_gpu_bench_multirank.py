"""bench.py's N > 1 path, rehearsed with two ranks on ONE GPU exactly as the driver launches it (torch.distributed.run), with
gloo for rendezvous / barriers and the P2P transport for the records (RCCL refuses two ranks on one device): the line must carry
the per-rank phase timings, the collective's measured latency, a roofline block, the strong-scaling value, and pass its own
parity gate.  A small workload: this checks the wiring the 8-GPU run depends on, not speed."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("scan_kind", ["whole", "local"])
def test_bench_two_ranks_on_one_gpu(scan_kind):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4", "--map-points", "2000000", "--scan-points", "140000",
           "--dist-backend", "gloo", "--scan-kind", scan_kind]
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{") and '"metric"' in l][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["parity"]["ok"] and d["value"] > 0 and d["scaling"] == "weak"
    assert d["collective"]["kind"] == "p2p" and d["collective"]["allreduce_us_measured"]["p2p"] > 0
    assert d["config"]["collective"] == "p2p" and "P2P" in d["config"]["parallelism"]
    r = d["ranks"]
    assert len(r["per_rank"]) == 2 and r["step_ms"]["max"] >= r["step_ms"]["min"] > 0
    for st in r["per_rank"]:
        assert st["scans"] >= 1 and st["map_points"] > 0
    if scan_kind == "whole":                                           # every scan spans both slabs: one group, both ranks own about half of every scan
        assert d["config"]["routing_groups"] == {"0-1": 8}
        assert r["owned_queries_per_launch"]["min"] > 0.3 * 8 * 140000 / 2 and r["collective_us_mean"]["mean"] > 0 and r["nn_us_mean"]["mean"] > 0
        assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
        assert d["value_strong"]["value"] > 0 and d["value_strong"]["scans_in_flight"] == 4
        assert d["value_replicas"]["value"] > 0 and d["value_replicas"]["scans_in_flight_per_gpu"] == 4 and d["value_replicas"]["max_translation_err_vs_truth_m"] < 5e-3
    else:                                                              # 10 m scans in a 44 m map: some inside one slab, some across the edge
        assert sum(d["config"]["routing_groups"].values()) == 8
