"""pytest configuration: `gpu` marks tests that need an MI355X and call the HIP path through
the C ABI; everything else runs on CPU (oracle vs golden vectors, host logic, ABI surface)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (runs the HIP kernels through libslamfusion.so)")


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def synth():
    from slam_sensor_fusion_amd import synth as s
    return s


@pytest.fixture(scope="session")
def api():
    from slam_sensor_fusion_amd import api as a
    return a


@pytest.fixture(scope="session")
def ctx(api):
    """HIP context on cuda:0 — fails loudly (no fallback) if the library or the GPU is missing."""
    return api.Context(0)


@pytest.fixture(scope="session")
def small_world(orc, synth):
    """Config-1-sized synthetic world: 100k-pt raw map -> voxel 0.1 (oracle) + 10k-pt scan."""
    raw = synth.make_map(100_000)
    ds, vidx, ovox, st = orc.voxel_pcl(raw, 0.1)
    scan, idx = synth.make_scan(ds, 10_000)
    return dict(raw=raw, map=ds, scan=scan, scan_idx=idx)


def load_golden(name):
    return np.load(os.path.join(ROOT, "tests", "golden", name), allow_pickle=False)
