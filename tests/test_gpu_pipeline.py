"""Consecutive sf_icp_align_batch_async calls overlap on two internal lanes (sf_icp_set_pipeline, include/slamfusion.h): the
results must be those of the same alignment run alone -- same kernels, same data -- whatever is enqueued between the calls:
nothing (the alignments run side by side), new initial poses, a new source, a rebuilt target.  No reference counterpart (the
reference registers one scan per callback on one thread, localization_node.cpp:337, main.cpp:18)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world(api, ctx, synth):
    raw = synth.make_map(300_000)
    cloud = api.Cloud(ctx, raw)
    cloud.voxel_downsample(0.1, "pcl")
    ds = cloud.download()
    mp = api.Map(ctx, cloud, 0.25)
    mp.estimate_normals(0.25)
    scans = np.stack([synth.make_scan(ds, 140_000, scan_id=70 + b)[0][:140_000] for b in range(4)])
    return dict(map=mp, ds=ds, scans=scans)


def make(api, ctx, world, pipeline, graph, iters=12):
    icp = api.Icp(ctx, 0.5, iters, 0.05, 1e-5)
    icp.set_target(world["map"])
    icp.use_graph(graph)
    icp.set_pipeline(pipeline)
    icp.set_freeze(True)
    icp.set_source_batch(world["scans"])
    icp.set_initial_batch(None)
    return icp


def bitwise(a, b):
    for x, y in zip(a, b):
        assert np.array_equal(x["T64"], y["T64"]) and x["iterations"] == y["iterations"] and x["n_corr"] == y["n_corr"] and x["rmse"] == y["rmse"]


@pytest.mark.parametrize("graph", [False, True])
def test_back_to_back_alignments_equal_one_alone(api, ctx, synth, world, graph):
    ref = make(api, ctx, world, False, graph).align_batch("p2plane")
    icp = make(api, ctx, world, True, graph)
    for _ in range(5):
        icp.align_batch_async("p2plane")
    bitwise(icp.fetch_results(), ref)
    bitwise(icp.align_batch("p2plane"), ref)            # and again, on the other lane
    for mode in ("o3d_p2p", "ref_cpp"):
        want = make(api, ctx, world, False, graph).align_batch(mode)
        for _ in range(3):
            icp.align_batch_async(mode)
        bitwise(icp.fetch_results(), want)


def test_inputs_changed_between_calls_are_seen(api, ctx, synth, world):
    icp = make(api, ctx, world, True, True)
    alone = make(api, ctx, world, False, True)
    inits = np.stack([synth.make_T((0.03 * k, -0.02, 0.01), (0.1 * k, 0.0, 0.2)) for k in range(4)])
    icp.align_batch_async("p2plane")
    icp.set_initial_batch(inits)                          # new priors while the first alignment is still in flight
    icp.align_batch_async("p2plane")
    alone.set_initial_batch(inits)
    bitwise(icp.fetch_results(), alone.align_batch("p2plane"))
    other = world["scans"][::-1].copy()
    icp.align_batch_async("p2plane")
    icp.set_source_batch(other)                           # a new source: uploaded behind the alignment in flight, seen by the next
    icp.set_initial_batch(inits)
    icp.align_batch_async("p2plane")
    alone.set_source_batch(other)
    alone.set_initial_batch(inits)
    bitwise(icp.fetch_results(), alone.align_batch("p2plane"))


def test_a_target_rebuilt_between_calls(api, ctx, synth, world):
    cloud = api.Cloud(ctx, world["ds"])
    mp = api.Map(ctx, cloud, 0.25)
    mp.estimate_normals(0.25)
    w = dict(world, map=mp)
    icp = make(api, ctx, w, True, True)
    icp.align_batch_async("p2plane")
    half = world["ds"][world["ds"][:, 0] > -2.0]          # the map shrinks while the alignment is in flight
    mp.build(api.Cloud(ctx, half), 0.25)
    mp.estimate_normals(0.25)
    icp.align_batch_async("p2plane")
    got = icp.fetch_results()
    alone = make(api, ctx, w, False, True)
    bitwise(got, alone.align_batch("p2plane"))
