"""Multi-GPU MSM: one process per GPU, points partitioned by range, one all-gather.

An MSM is a sum over independent (point, scalar) pairs, so each rank runs the whole
single-GPU pipeline on its range and keeps its per-window partial sums on the device;
the partials (n_windows x 128 B per rank) are exchanged with ONE all_gather
(torch.distributed; backend "nccl" is RCCL over xGMI) and every rank combines them
window by window on its GPU.  Curve addition is not an RCCL reduction operator, hence
all-gather + combine rather than all-reduce.

The window sums of different ranks only add up if every rank cut its scalars into
windows of the same width.  The width is therefore a function of the TOTAL size and the
world size alone (bp_msm_shard_window_bits: the width the library picks for the largest
shard), never of a rank's own shard size.  This module is a thin driver over the C entry
points bp_msm_shard_window_bits / bp_msm_shard_windows_dev /
bp_msm_shard_windows_resident / bp_msm_combine_windows_dev; a host in another language
calls those directly with its own all-gather.
"""
from __future__ import annotations

MAX_WINDOWS = 64
WINDOW_U64 = 16  # one XYZZ point


def shard_range(n: int, rank: int, world: int):
    """[begin, end) of the points rank owns: contiguous ranges, sizes differ by at most one."""
    base, rem = divmod(n, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


class ShardedMsm:
    def __init__(self, ctx, curve, dist=None, device="cuda", local_windows=None, combine=None, n_total=None):
        """n_total: points of the whole MSM over all ranks (fixes the common window width; may also
        be given per call).  local_windows(d_x, d_y, d_inf, d_s, n, window_bits, out_tensor) ->
        n_windows and combine(gathered_tensor, world, n_windows, window_bits) -> xyz default to the
        library's device entry points; tests on CPU inject oracle-backed stand-ins."""
        import torch

        self.torch = torch
        self.ctx, self.curve, self.dist = ctx, curve, dist
        self.collective = dist is not None and dist.is_initialized()
        self.world = dist.get_world_size() if self.collective else 1
        self.rank = dist.get_rank() if self.collective else 0
        self.n_total = n_total
        self.win = torch.zeros(MAX_WINDOWS * WINDOW_U64, dtype=torch.int64, device=device)
        self.gathered = torch.zeros((self.world, MAX_WINDOWS * WINDOW_U64), dtype=torch.int64, device=device)
        self._local = local_windows or self._local_dev
        self._combine = combine or self._combine_dev

    def window_bits(self, n_total=None):
        from .api import shard_window_bits

        n_total = self.n_total if n_total is None else n_total
        if n_total is None:
            raise ValueError("ShardedMsm needs the total number of points (n_total) to fix the window width")
        return shard_window_bits(n_total, self.world)

    def _local_dev(self, d_x, d_y, d_inf, d_s, n, window_bits, out):
        return self.ctx.msm_shard_windows_dev(self.curve, d_x, d_y, d_inf, d_s, n, window_bits, out.data_ptr())

    def _combine_dev(self, gathered, world, nw, cb):
        packed = gathered[:, :nw * WINDOW_U64].contiguous()
        return self.ctx.msm_combine_windows_dev(self.curve, packed.data_ptr(), world, nw, cb)

    def _exchange(self, nw, cb):
        if self.collective:
            self.dist.all_gather_into_tensor(self.gathered.view(-1), self.win)
        else:
            self.gathered[0].copy_(self.win)
        if self.gathered.is_cuda:
            self.torch.cuda.current_stream().synchronize()
        return self._combine(self.gathered, self.world, nw, cb)

    def msm(self, d_x, d_y, d_inf, d_s, n_local, n_total=None):
        """this rank's n_local points (device pointers) of an n_total-point MSM; every rank returns the sum"""
        cb = self.window_bits(n_total)
        nw = self._local(d_x, d_y, d_inf, d_s, n_local, cb, self.win)
        return self._exchange(nw, cb)

    def msm_resident(self, bases, scalars, n_total=None, offset=0, n=None):
        """as msm over this rank's registered range of a sharded base set (api.Bases), host scalars"""
        cb = self.window_bits(n_total)
        nw = bases.shard_windows(scalars, cb, self.win.data_ptr(), offset, n)
        return self._exchange(nw, cb)
