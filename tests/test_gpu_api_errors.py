"""Error behaviour of the C ABI: infrastructure / call-order faults are negative
status codes with a message; algorithmic "no lock" is reported like the reference, not as an error."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_call_order_and_argument_errors(api, ctx, small_world):
    icp = api.Icp(ctx, 0.5, 10, 0.05, 1e-5)
    with pytest.raises(api.SlamFusionError, match="no source"):
        icp.align("ref_cpp")
    icp.set_source(small_world["scan"][:100])
    with pytest.raises(api.SlamFusionError, match="no target"):
        icp.align("ref_cpp")
    mp = api.Map(ctx, api.Cloud(ctx, small_world["map"]), 0.25)
    icp.set_target(mp)
    with pytest.raises(api.SlamFusionError, match="normals"):
        icp.align("p2plane")
    with pytest.raises(api.SlamFusionError):
        api.Map(ctx).nn(np.zeros((1, 3), np.float32))                  # map not built
    with pytest.raises(api.SlamFusionError):
        mp.set_normals(np.zeros((5, 3), np.float32))                   # wrong size
    with pytest.raises(api.SlamFusionError):
        api.Cloud(ctx, small_world["scan"]).voxel_downsample(-1.0, "pcl")
    with pytest.raises(api.SlamFusionError):
        api.Map(ctx, api.Cloud(ctx, small_world["map"]), 1e-4)         # explicit cell far too small for the extent
    with pytest.raises(api.SlamFusionError):
        api.Context(99)                                                # no such device
    icp.set_source_batch(np.stack([small_world["scan"][:100]] * 2))
    with pytest.raises(api.SlamFusionError, match="single source scan"):
        icp.align("ref_cpp")
    assert len(icp.align_batch("ref_cpp")) == 2


def test_empty_and_tiny_inputs(api, ctx, small_world):
    mp = api.Map(ctx, api.Cloud(ctx, small_world["map"]), 0.25)
    mp.estimate_normals(0.25)
    icp = api.Icp(ctx, 0.5, 10, 0.05, 1e-5)
    icp.set_target(mp)
    icp.set_source(np.zeros((0, 3), np.float32))
    r = icp.align("ref_cpp")
    assert r["iterations"] == 0 and not r["converged"] and r["error"] == np.float32(1e6)   # < 10 correspondences
    assert icp.align("o3d_p2p")["n_corr"] == 0
    r = icp.align("p2plane")
    assert r["flags"] & api.SF_ICP_FLAG_SINGULAR and r["iterations"] == 0
    icp.set_source(small_world["scan"][:9])                            # 9 < 10: the reference aborts
    assert icp.align("ref_cpp")["flags"] & api.SF_ICP_FLAG_FEW_CORR
    icp.set_source(small_world["scan"][:10])
    assert icp.align("ref_cpp")["n_corr"] == 10
    empty_map = api.Map(ctx, api.Cloud(ctx, np.zeros((0, 3), np.float32)))
    icp2 = api.Icp(ctx, 0.5, 10, 0.05, 1e-5)
    icp2.set_target(empty_map)
    icp2.set_source(small_world["scan"][:100])
    assert icp2.align("ref_cpp")["iterations"] == 0
    nan_scan = small_world["scan"][:500].copy()
    nan_scan[::7] = np.nan                                             # non-finite source points never match
    icp.set_source(nan_scan)
    r = icp.align("p2plane")
    assert r["n_corr"] == np.isfinite(nan_scan).all(1).sum() and np.isfinite(r["T64"]).all()


def test_sharded_icp_needs_the_step_api(api, ctx, small_world):
    mp = api.Map(ctx, api.Cloud(ctx, small_world["map"]), 0.25)
    icp = api.Icp(ctx, 0.5, 10, 0.05, 1e-5)
    icp.set_target(mp)
    icp.set_source(small_world["scan"][:500])
    icp.set_shard(-1.0, 1.0)
    with pytest.raises(api.SlamFusionError, match="sharded"):
        icp.align("o3d_p2p")
    with pytest.raises(api.SlamFusionError):
        icp.step_begin("o3d_p2p", 3)
    with pytest.raises(api.SlamFusionError):
        icp.set_shard_margin(0.0)
