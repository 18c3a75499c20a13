"""The multi-GPU paths on MORE THAN ONE DEVICE: skipped on a one-GPU box (every test here needs at least two), run as they
stand the day two devices are visible.  What they exercise that the one-device rehearsals (tests/test_gpu_p2p_ranks.py,
tests/test_gpu_bench_multirank.py) cannot: peer mapping of the uncached exchange region across devices
(sf_comm_p2p_rendezvous: hipIpcOpenMemHandle of another device's allocation), store-and-flag visibility over xGMI, and RCCL
with one rank per device.  No reference counterpart (one CPU process, localization/src/main.cpp:18)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from test_gpu_bench_multirank import free_port
from test_gpu_p2p_ranks import run_ranks, unsharded

pytestmark = pytest.mark.gpu


def n_devices():
    import torch
    return torch.cuda.device_count()            # (counting devices does not initialise the GPU)


needs_two = pytest.mark.skipif(n_devices() < 2, reason="needs at least two GPUs")


@needs_two
def test_p2p_ranks_one_per_device(api, ctx, synth, small_world, tmp_path, monkeypatch):
    world = min(n_devices(), 4)
    monkeypatch.setenv("SF_TEST_RANK_PER_DEVICE", "1")
    ds = small_world["map"]
    scans = np.stack([synth.make_scan(ds, 6000, scan_id=s)[0] for s in range(3)])
    inits = np.stack([np.eye(4), synth.make_T((0.45, 0.0, 0.0), (0, 0, 0)), synth.make_T((0.0, 0.1, 0.0), (0, 0, 0.05))])
    outs = run_ranks(tmp_path, world, "parity", ds, scans, inits)
    assert len(outs) == world
    for name, mode, iters in (("p2plane_m1", "p2plane", 20), ("o3d", "o3d_p2p", 30)):
        ref = unsharded(api, ctx, ds, scans, inits, mode, iters)
        k = "%s_r0_" % name
        for r in range(1, world):
            assert np.array_equal(outs[r][k + "T"], outs[0][k + "T"])               # bitwise equal across devices
        assert np.array_equal(outs[0]["%s_r1_T" % name], outs[0][k + "T"])           # and from run to run
        for b in range(len(scans)):
            assert outs[0][k + "iterations"][b] == ref[b]["iterations"] and outs[0][k + "n_corr"][b] == ref[b]["n_corr"]
            dt, dr = synth.pose_error(outs[0][k + "T"][b], ref[b]["T64"])
            assert dt < 1e-9 and dr < 1e-10, (name, b, dt, dr)


@needs_two
@pytest.mark.parametrize("collective", ["c", "p2p"])
def test_bench_two_ranks_on_two_devices(collective):
    """bench.py as the driver launches it for N = 2, RCCL (`c`) and the hand-written transport (`p2p`) over xGMI."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4", "--map-points", "2000000", "--scan-points", "140000",
           "--collective", collective]
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{") and '"metric"' in l][-1])
    assert d["n_gpus"] == 2 and d["parity"]["ok"] and d["value"] > 0
    assert d["collective"]["kind"] == collective, d["collective"]
