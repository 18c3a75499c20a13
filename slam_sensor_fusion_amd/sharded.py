"""Multi-GPU host driver: the map tile-sharded along x with a halo, one rank per GPU, the
17-/30-scalar normal-equation record all-reduced once per ICP iteration.

Every rank holds the whole scan batch (<= 2.4 MB per scan) and only accumulates the
queries whose TRANSFORMED x falls inside its core slab [x_lo, x_hi); the slab carries a
halo of (max correspondence distance + normal radius) so an owned query's true nearest
neighbour — and that neighbour's normal — are always local.  After the all-reduce every
rank runs the identical solve, so no broadcast of the pose is needed.
"""
import numpy as np


def slab_edges(map_x, world):
    """Equal-count slab boundaries along x: world+1 edges, first/last infinite."""
    if world <= 1:
        return np.array([-np.inf, np.inf])
    qs = np.quantile(np.asarray(map_x, dtype=np.float64), np.arange(1, world) / world)
    return np.concatenate([[-np.inf], qs, [np.inf]])


def slab_select(map_xyz, edges, rank, halo):
    """Indices of the map points rank `rank` must hold: core slab widened by the halo."""
    x = map_xyz[:, 0]
    lo, hi = edges[rank] - halo, edges[rank + 1] + halo
    return np.nonzero((x >= lo) & (x < hi))[0]


class ShardedIcp:
    """Drives sf_icp_step_begin / all-reduce / sf_icp_step_end for one rank.

    `allreduce(ptr_or_tensor)` is injected: RCCL through torch.distributed on the GPU,
    gloo in the CPU tests (where `icp` is a numpy stand-in with the same step API).
    """

    def __init__(self, icp, mode, num_iterations, allreduce):
        self.icp, self.mode, self.iters, self.allreduce = icp, mode, num_iterations, allreduce

    def n_steps(self):
        # O3D_P2P evaluates once more after the last update (registration_icp's final search)
        return self.iters + 1 if self.mode == "o3d_p2p" else self.iters

    def align_async(self):
        steps = self.n_steps()
        for k in range(steps):
            self.icp.step_begin(self.mode, first=(k == 0))
            self.allreduce()
            self.icp.step_end(self.mode, last=(k == steps - 1))
