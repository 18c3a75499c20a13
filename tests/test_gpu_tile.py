"""Tile search (sf_tile.hpp, sf_icp_set_tile_search): the searching launches served out of LDS tile by tile must form the
same pairs as the search through the global grid index (icp_point_to_point.cpp:64-69 / localization_node.py:233-237 are
exact 1-NN searches either way).  The queries are sorted by tile instead of by cell bucket, so the float64 sums differ by
summation order only: iterations and correspondence counts equal, poses within 1e-9."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world(api, ctx, synth):
    raw = synth.make_map(600_000)
    cloud = api.Cloud(ctx, raw)
    cloud.voxel_downsample(0.1, "pcl")
    ds = cloud.download()
    mp = api.Map(ctx, cloud, 0.25)
    mp.estimate_normals(0.25)
    B, N = 5, 140_000                      # above 131 072 points per scan: the launch list of wide scans
    scans = np.stack([synth.make_scan(ds, N, scan_id=b)[0][:N] for b in range(B)])
    return dict(map=mp, ds=ds, scans=scans)


def run(api, ctx, world, mode, tile, reuse, iters=10, inits=None, scans=None):
    icp = api.Icp(ctx, 0.5, iters, 0.05, 1e-5)
    icp.set_target(world["map"])
    icp.use_graph(False)
    icp.set_nn_reuse(reuse)
    icp.set_freeze(False)
    icp.set_tile_search("always" if tile else False)
    icp.profile_enable(True)
    icp.set_source_batch(world["scans"] if scans is None else scans)
    icp.set_initial_batch(inits)
    res = icp.align_batch(mode)
    return res, icp.tile_info()


def same(synth, a, b, tol_t=1e-9, tol_r=1e-9):
    for x, y in zip(a, b):
        assert x["iterations"] == y["iterations"] and x["n_corr"] == y["n_corr"] and x["converged"] == y["converged"]
        dt, dr = synth.pose_error(x["T64"], y["T64"])
        assert dt < tol_t and dr < tol_r, (dt, dr)


@pytest.mark.parametrize("mode", ["p2plane", "o3d_p2p"])
@pytest.mark.parametrize("reuse", [False, True])
def test_tile_search_forms_the_same_pairs(api, ctx, synth, world, mode, reuse):
    r0, i0 = run(api, ctx, world, mode, False, reuse)
    r1, i1 = run(api, ctx, world, mode, True, reuse)
    assert not i0["on"] and i1["on"]
    assert i1["searched"] > 0 and i1["from_lds"] + i1["left_region"] + i1["beyond_ring_1"] == i1["searched"]
    assert i1["from_lds"] > 0.99 * i1["searched"]          # 0.1 m of motion: inside the two-cell halo
    same(synth, r0, r1)


def test_queries_that_leave_the_staged_region_walk_the_global_index(api, ctx, synth, world):
    # an index of 0.1 m cells: the staged halo tolerates 0.1 m of motion since the queries were binned, the scans start 0.11 m off
    cloud = api.Cloud(ctx, world["ds"])
    mp = api.Map(ctx, cloud, 0.1)
    mp.estimate_normals(0.25)
    w = dict(map=mp, scans=world["scans"])
    r0, _ = run(api, ctx, w, "p2plane", False, True, iters=8)
    r1, i1 = run(api, ctx, w, "p2plane", True, True, iters=8)
    assert i1["on"] and i1["left_region"] > 0 and i1["from_lds"] > 0
    same(synth, r0, r1)


def test_scan_points_off_the_map_and_non_finite(api, ctx, synth, world):
    scans = world["scans"].copy()
    scans[0, :500] += np.float32(300.0)        # far outside the map's box: no neighbour
    scans[1, 100:110] = np.nan
    scans[2, :2000] *= np.float32(1.02)        # the rim: some beyond the acceptance radius, ring 2 and beyond for others
    r0, _ = run(api, ctx, world, "p2plane", False, True, scans=scans)
    r1, i1 = run(api, ctx, world, "p2plane", True, True, scans=scans)
    same(synth, r0, r1)
    r0, _ = run(api, ctx, world, "o3d_p2p", False, False, scans=scans)
    r1, i1 = run(api, ctx, world, "o3d_p2p", True, False, scans=scans)
    same(synth, r0, r1)


def test_a_map_too_dense_for_its_tiles_falls_back_tile_by_tile(api, ctx, synth):
    # half the map's points packed into one cubic metre: the tiles there exceed what a workgroup stages
    rng = np.random.Generator(np.random.PCG64(5))
    raw = synth.make_map(200_000)
    clump = (rng.random((150_000, 3), dtype=np.float32) - 0.5) * np.float32(1.0) + np.float32([3.0, -2.0, 0.5])
    pts = np.concatenate([raw, clump]).astype(np.float32)
    cloud = api.Cloud(ctx, pts)
    mp = api.Map(ctx, cloud, 0.25)
    mp.estimate_normals(0.25)
    B, N = 3, 135_000
    scans = np.stack([synth.make_scan(pts, N, scan_id=b)[0][:N] for b in range(B)])
    w = dict(map=mp, scans=scans)
    r0, _ = run(api, ctx, w, "p2plane", False, True, iters=6)
    r1, i1 = run(api, ctx, w, "p2plane", True, True, iters=6)
    assert i1["on"] and i1["tiles_too_dense"] > 0
    same(synth, r0, r1)
