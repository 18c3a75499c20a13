"""One rank of the multi-process P2P tests (tests/test_gpu_p2p_ranks.py starts `world` of these as fresh child
processes on device 0).  Every rank indexes its x-slab (+ halo) of the map the parent saved, creates the P2P
communicators through the shared-memory rendezvous and runs the C-side sharded loop (sf_icp_align_sharded);
results go to <dir>/out_rank<r>.npz, failures to a non-zero exit code.

argv: <dir> <rank> <world> <case>
  case "parity"  spanning scans: p2plane at margin 1.0 and 0.2 (stale -> resume), o3d_p2p; every alignment twice
  case "routed"  scans routed to the slabs they touch: one-slab scans alone, sub-group communicators for the rest
  case "frozen"  wide scans, frozen pairs forced on, over the P2P collective
  case "crowd"   hundreds of scans in flight per rank (co-residency of the ranks' small kernels on one device)
  case "peer_dies"  the last rank leaves after the first alignment; the others must get SF_ERR_COMM, not wait
  case "peer_aborts" the last rank calls sf_comm_abort instead of its second alignment
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MAX_DIST, NORMAL_RADIUS, CELL = 0.5, 0.25, 0.25


def pack(results):
    return dict(T=np.stack([r["T64"] for r in results]), iterations=np.array([r["iterations"] for r in results]),
                n_corr=np.array([r["n_corr"] for r in results]), flags=np.array([r["flags"] for r in results]),
                converged=np.array([int(r["converged"]) for r in results]))


def main():
    d, rank, world, case = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    from slam_sensor_fusion_amd import api, sharded
    ds = np.load(os.path.join(d, "map.npy"))
    scans = np.load(os.path.join(d, "scans.npy"))
    inits = np.load(os.path.join(d, "inits.npy"))
    tag = open(os.path.join(d, "tag.txt")).read().strip()
    # SF_TEST_RANK_PER_DEVICE: one rank per device (tests/test_gpu_multi_device.py); default: every rank on device 0
    ctx = api.Context(rank if os.environ.get("SF_TEST_RANK_PER_DEVICE") == "1" else 0)
    edges = sharded.slab_edges(ds[:, 0], world)
    keep = sharded.slab_select(ds, edges, rank, halo=MAX_DIST + NORMAL_RADIUS + CELL)
    mp = api.Map(ctx, api.Cloud(ctx, ds[keep]), CELL)
    mp.estimate_normals(NORMAL_RADIUS)
    out = {}

    def comm_for(lo, hi, capacity):
        c = api.Comm.p2p(ctx, hi - lo + 1, rank - lo, capacity)
        c.rendezvous("sf_%s_%d_%d" % (tag, lo, hi), 90.0)
        return c

    def new_icp(iters, margin=None):
        icp = api.Icp(ctx, MAX_DIST, iters, 0.05, 1e-5)
        icp.set_target(mp)
        if margin is not None:
            icp.set_shard_margin(margin)
        return icp

    def my_slab(icp):
        icp.set_shard(float(max(edges[rank], -1e30)), float(min(edges[rank + 1], 1e30)))

    if case == "parity":
        comm = comm_for(0, world - 1, 32 * len(scans))
        for name, mode, iters, margin in (("p2plane_m1", "p2plane", 20, 1.0), ("p2plane_m02", "p2plane", 20, 0.2), ("o3d", "o3d_p2p", 30, 1.0)):
            icp = new_icp(iters, margin)
            my_slab(icp)
            icp.set_source_batch(scans)
            icp.set_initial_batch(inits)
            for rep in range(2):
                res = icp.align_sharded(mode, comm)
                for k, v in pack(res).items():
                    out["%s_r%d_%s" % (name, rep, k)] = v
                out["%s_r%d_resumes" % (name, rep)] = np.array(icp.resumes)
                out["%s_r%d_owned" % (name, rep)] = icp.owned_counts()
            icp.close()
        comm.status()
    elif case == "frozen":
        # wide scans (two queries per lane) with the frozen pairs forced on: every rank freezes its own owned queries, the
        # records still meet in the same collective
        comm = comm_for(0, world - 1, 32 * len(scans))
        icp = new_icp(20, 1.0)
        icp.set_freeze(True)
        my_slab(icp)
        icp.set_source_batch(scans)
        icp.set_initial_batch(inits)
        for rep in range(2):
            res = icp.align_sharded("p2plane", comm)
            for k, v in pack(res).items():
                out["frozen_r%d_%s" % (rep, k)] = v
        st = icp.freeze_stats()
        out["frozen_froze"] = np.array(st["froze"])
        out["frozen_at_end"] = np.array(st["frozen_at_end"])
        icp.close()
    elif case == "crowd":
        # many scans in flight: (world - 1) x scans waves of the peers' gather kernels spin on this device while this rank's
        # publish kernel has to be placed
        comm = comm_for(0, world - 1, 32 * len(scans))
        comm.set_timeout(30.0)
        icp = new_icp(8, 1.0)
        my_slab(icp)
        icp.set_source_batch(scans)
        icp.set_initial_batch(inits)
        res = icp.align_sharded("p2plane", comm)
        for k, v in pack(res).items():
            out["crowd_" + k] = v
        icp.close()
    elif case == "routed":
        comms = {}

        def make_sharded(lo, hi):
            if (lo, hi) not in comms:
                comms[(lo, hi)] = comm_for(lo, hi, 32 * len(scans))
            return new_icp(20), comms[(lo, hi)]

        reg = sharded.RoutedRegistration(rank, world, edges, api.shard_route, make_local=lambda: new_icp(20), make_sharded=make_sharded, margin=1.0)
        plan = reg.set_source_batch(scans, inits)
        res = reg.align("p2plane")
        ids = sorted(res)
        out["ids"] = np.array(ids, dtype=np.int64)
        for k, v in pack([res[i] for i in ids]).items():
            out["routed_" + k] = v
        out["groups"] = np.array([[a, e, len(v)] for (a, e), v in plan.items()], dtype=np.int64)
        out["n_comms"] = np.array(len(comms))
    elif case in ("peer_dies", "peer_aborts"):
        comm = comm_for(0, world - 1, 32 * len(scans))
        comm.set_timeout(3.0)
        icp = new_icp(20)
        my_slab(icp)
        icp.set_source_batch(scans)
        icp.set_initial_batch(inits)
        res = icp.align_sharded("p2plane", comm)          # everyone takes part: fine
        out["first_iterations"] = np.array([r["iterations"] for r in res])
        np.savez(os.path.join(d, "out_rank%d.npz" % rank), **out)
        if rank == world - 1:
            if case == "peer_aborts":
                comm.abort()
                os._exit(7)
            os._exit(0)                                    # leaves without a word
        t0 = time.time()
        try:
            icp.align_sharded("p2plane", comm)
        except api.SlamFusionError as e:
            took = time.time() - t0
            ok = "error -6" in str(e) and took < 15.0
            print("rank %d: second alignment refused after %.2f s: %s" % (rank, took, e), flush=True)
            np.savez(os.path.join(d, "fail_rank%d.npz" % rank), took=np.array(took), ok=np.array(int(ok)))
            try:
                icp.align_sharded("p2plane", comm)         # the communicator stays poisoned: immediate refusal
                os._exit(1)
            except api.SlamFusionError:
                pass
            os._exit(3 if ok else 1)                       # a rank whose collective failed exits non-zero
        os._exit(1)                                        # the alignment must not succeed without its peer
    else:
        raise SystemExit("unknown case " + case)
    np.savez(os.path.join(d, "out_rank%d.npz" % rank), **out)


if __name__ == "__main__":
    main()
