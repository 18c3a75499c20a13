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


def test_random_call_sequences_equal_the_same_calls_on_one_lane(api, ctx, synth, small_world):
    """Seeded fuzz: the same random sequence of calls -- asynchronous alignments in all three modes, new priors, new sources, another
    iteration count, fetches -- on an object with the lanes and on one without; every fetched result must be bit-identical.
    Small scans through the launch list (sf_icp_set_fused(0)) keep it quick; SF_FUZZ_TRIALS / SF_FUZZ_SEED override the defaults."""
    import os
    m = small_world["map"]
    mp = api.Map(ctx, api.Cloud(ctx, m), 0.25)
    mp.estimate_normals(0.25)
    rng = np.random.default_rng(int(os.environ.get("SF_FUZZ_SEED", "5")))
    pool = [np.stack([synth.make_scan(m, 6000, scan_id=300 + 3 * k + b)[0][:6000] for b in range(3)]) for k in range(3)]
    objs = []
    for pipe in (True, False):
        icp = api.Icp(ctx, 0.5, 8, 0.05, 1e-5)
        icp.set_target(mp)
        icp.set_fused(False)
        icp.use_graph(True)
        icp.set_pipeline(pipe)
        icp.set_source_batch(pool[0])
        icp.set_initial_batch(None)
        objs.append(icp)
    pending = False
    for t in range(int(os.environ.get("SF_FUZZ_TRIALS", "60"))):
        op = rng.choice(["align", "align", "align", "inits", "source", "iters", "fetch"])
        if op == "align":
            mode = str(rng.choice(["p2plane", "o3d_p2p", "ref_cpp"]))
            for icp in objs:
                icp.align_batch_async(mode)
            pending = True
        elif op == "inits":
            inits = np.stack([synth.make_T(rng.normal(0, 0.02, 3), rng.normal(0, 0.2, 3)) for _ in range(3)])
            for icp in objs:
                icp.set_initial_batch(inits)
        elif op == "source":
            k = int(rng.integers(0, 3))
            for icp in objs:
                icp.set_source_batch(pool[k])
                icp.set_initial_batch(None)
            pending = False
        elif op == "iters":
            it = int(rng.integers(1, 12))
            for icp in objs:
                icp.set_num_iterations(it)
        elif pending:
            bitwise(objs[0].fetch_results(), objs[1].fetch_results())
    if pending:
        bitwise(objs[0].fetch_results(), objs[1].fetch_results())


def test_fetch_previous_delivers_every_result_of_a_stream(api, ctx, synth, world):
    """The streaming loop of include/slamfusion.h (sf_icp_fetch_previous): enqueue the next alignment, fetch the one before it.
    Every result is that of the same alignment run alone, bit for bit; new priors between the calls are seen by the next
    alignment only."""
    inits = [None] + [np.stack([synth.make_T((0.02 * k * s, -0.01 * s, 0.01), (0.1 * k, 0.05 * s, 0.2)) for k in range(4)]) for s in (1, 2, 3)]
    alone = make(api, ctx, world, False, True)
    want = []
    for T in inits:
        alone.set_initial_batch(T)
        want.append(alone.align_batch("p2plane"))
    icp = make(api, ctx, world, True, True)
    got = []
    for i, T in enumerate(inits):
        icp.set_initial_batch(T)
        icp.align_batch_async("p2plane")
        if i > 0:
            got.append(icp.fetch_previous())
    got.append(icp.fetch_results())
    assert len(got) == len(want)
    for g, w in zip(got, want):
        bitwise(g, w)
    # nothing left: the previous one was fetched, and after a fetch the next alignment takes the buffers at hand
    with pytest.raises(api.SlamFusionError):
        icp.fetch_previous()
    icp.align_batch_async("p2plane")
    with pytest.raises(api.SlamFusionError):
        icp.fetch_previous()
    bitwise(icp.fetch_results(), want[-1])


def test_fetch_previous_across_modes_and_batch_sizes(api, ctx, synth, world):
    """The earlier alignment is described by what it was enqueued with: another mode, another number of scans."""
    icp = make(api, ctx, world, True, False)
    alone = make(api, ctx, world, False, False)
    icp.align_batch_async("o3d_p2p")                      # four scans, point-to-point
    two = world["scans"][:2].copy()
    icp.set_source_batch(two)
    icp.set_initial_batch(None)
    icp.align_batch_async("p2plane")                      # two scans, point-to-plane, beside it
    first = icp.fetch_previous()
    bitwise(first, alone.align_batch("o3d_p2p"))
    alone.set_source_batch(two)
    alone.set_initial_batch(None)
    bitwise(icp.fetch_results(), alone.align_batch("p2plane"))
    # pipeline off: alignments share one set of buffers, there is never a previous one to fetch
    off = make(api, ctx, world, False, False)
    off.align_batch_async("p2plane")
    off.align_batch_async("p2plane")
    with pytest.raises(api.SlamFusionError):
        off.fetch_previous()
    off.fetch_results()


def test_a_stream_of_new_batches_equals_each_alone(api, ctx, synth, world):
    """The whole streaming loop: a NEW source batch and new priors every step, set while the previous step's alignment is in
    flight (the upload takes the other source set on the next lane's stream, sf_icp.hip SrcScope), the previous step's results
    fetched while this step runs.  Every step's results are those of the same batch registered alone, bit for bit -- also when a
    step sets its source twice, changes the number of scans, or fetches in between."""
    rng = np.random.default_rng(11)
    pool = [world["scans"], world["scans"][::-1].copy(), world["scans"][[1, 3, 0, 2]].copy(), world["scans"][:2].copy(), world["scans"][1:4].copy()]
    plan = [0, 1, 2, 3, 1, 4, 0, 2, 2, 1]
    steps = []
    for i, k in enumerate(plan):
        B = len(pool[k])
        inits = np.stack([synth.make_T(rng.normal(0, 0.02, 3), rng.normal(0, 0.2, 3)) for _ in range(B)])
        steps.append((k, inits))
    alone = make(api, ctx, world, False, True)
    want = []
    for k, inits in steps:
        alone.set_source_batch(pool[k])
        alone.set_initial_batch(inits)
        want.append(alone.align_batch("p2plane"))
    for graph in (True, False):
        icp = make(api, ctx, world, True, graph)
        got, prev_batch = [], None
        for i, (k, inits) in enumerate(steps):
            if i == 4:
                icp.set_source_batch(pool[0])             # a source that is replaced before any alignment reads it
            icp.set_source_batch(pool[k])
            icp.set_initial_batch(inits)
            if i == 6:                                    # a fetch between the source and its alignment: the latest alignment, as it was enqueued
                got.append(icp.fetch_results())
                prev_batch = None
            icp.align_batch_async("p2plane")
            if prev_batch is not None:
                got.append(icp.fetch_previous())
            prev_batch = len(pool[k])
        got.append(icp.fetch_results())
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert len(g) == len(w)
            bitwise(g, w)


def test_priors_set_before_the_next_source_are_kept(api, ctx, synth, world):
    """The setters are order independent: priors set BEFORE a source of the same number of scans stay, also when that source goes
    into the other source set because an alignment is in flight (whose batch size the set at hand may not share)."""
    inits = np.stack([synth.make_T((0.03 * k, -0.02, 0.01), (0.1 * k, 0.0, 0.2)) for k in range(4)])
    other = world["scans"][::-1].copy()
    alone = make(api, ctx, world, False, True)
    alone.set_source_batch(other)
    alone.set_initial_batch(inits)
    want = alone.align_batch("p2plane")
    icp = make(api, ctx, world, True, True)
    icp.align_batch_async("p2plane")
    icp.set_initial_batch(inits)
    icp.set_source_batch(other)                           # ahead of the alignment in flight: the first use of the second source set
    icp.align_batch_async("p2plane")
    icp.fetch_previous()
    bitwise(icp.fetch_results(), want)


def test_random_streams_deliver_every_result(api, ctx, synth, small_world):
    """Seeded fuzz of the streaming loop: random sources (three batches of different sizes, set once or twice, before or after the
    priors), priors, modes and iteration counts; one object streams -- enqueue, then sf_icp_fetch_previous, now and then a plain
    fetch in between -- the other registers every step alone.  EVERY step's results must agree bit for bit.  Launch list only
    (sf_icp_set_fused(0)); SF_FUZZ_TRIALS / SF_FUZZ_SEED override the defaults."""
    import os
    m = small_world["map"]
    mp = api.Map(ctx, api.Cloud(ctx, m), 0.25)
    mp.estimate_normals(0.25)
    rng = np.random.default_rng(int(os.environ.get("SF_FUZZ_SEED", "9")))
    pool = [np.stack([synth.make_scan(m, n, scan_id=400 + 5 * k + b)[0][:n] for b in range(nb)]) for k, (n, nb) in enumerate(((6000, 3), (4000, 2), (7000, 4)))]
    objs = []
    for pipe in (True, False):
        icp = api.Icp(ctx, 0.5, 8, 0.05, 1e-5)
        icp.set_target(mp)
        icp.set_fused(False)
        icp.use_graph(True)
        icp.set_pipeline(pipe)
        icp.set_source_batch(pool[0])
        icp.set_initial_batch(None)
        objs.append(icp)
    stream, alone = objs
    cur = 0
    want, got = [], []
    in_flight = 0        # alignments of `stream` enqueued and not fetched (0, 1 or 2)
    for t in range(int(os.environ.get("SF_FUZZ_TRIALS", "80"))):
        op = rng.choice(["align", "align", "align", "source", "inits", "iters", "fetch"])
        if op == "source":
            for _ in range(int(rng.integers(1, 3))):      # now and then a source that is replaced before it is read
                cur = int(rng.integers(0, 3))
                for icp in objs:
                    icp.set_source_batch(pool[cur])
                    icp.set_initial_batch(None)
        elif op == "inits":
            inits = np.stack([synth.make_T(rng.normal(0, 0.02, 3), rng.normal(0, 0.2, 3)) for _ in range(len(pool[cur]))])
            for icp in objs:
                icp.set_initial_batch(inits)
        elif op == "iters":
            it = int(rng.integers(1, 12))
            for icp in objs:
                icp.set_num_iterations(it)
        elif op == "fetch":
            if in_flight == 1:
                got.append(stream.fetch_results())
                in_flight = 0
        else:
            mode = str(rng.choice(["p2plane", "o3d_p2p", "ref_cpp"]))
            want.append(alone.align_batch(mode))
            if in_flight == 2:                            # (two in flight at most: the older one first)
                raise AssertionError("test bookkeeping")
            stream.align_batch_async(mode)
            in_flight += 1
            if in_flight == 2:
                got.append(stream.fetch_previous())
                in_flight = 1
    if in_flight == 1:
        got.append(stream.fetch_results())
    assert len(got) == len(want) and len(want) > 10
    for g, w in zip(got, want):
        assert len(g) == len(w)
        bitwise(g, w)
