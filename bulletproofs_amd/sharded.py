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


# ---- one inner-product argument over several GPUs ------------------------------------------------

_FR_MODULUS = {
    0: 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141,  # secp256k1
    1: 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001,  # vesta (its scalar field)
}


def _mont_one(curve):
    import numpy as np

    v = (1 << 256) % _FR_MODULUS[curve]
    return np.array([(v >> (64 * i)) & ((1 << 64) - 1) for i in range(4)], dtype=np.uint64)


def dist_gather(dist, device="cuda"):
    """gather(packet) -> (world, len(packet)) uint64 array over a torch.distributed group: one
    all_gather_into_tensor of a few hundred bytes per rank (RCCL with device="cuda", gloo with "cpu")"""
    import numpy as np
    import torch

    world = dist.get_world_size()

    def gather(packet):
        src = torch.from_numpy(np.ascontiguousarray(packet, dtype=np.uint64).view(np.int64)).to(device)
        out = torch.empty(world * src.numel(), dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(out, src)
        return out.cpu().numpy().view(np.uint64).reshape(world, -1)

    return gather


def host_gather(dist, group=None):
    """gather(packet) -> (world, len(packet)) uint64 array over a torch.distributed group whose backend
    takes CPU tensors (gloo): the packets of a sharded argument are results that have just arrived on the
    host, so they are exchanged there, without a trip to the device and back.  With an RCCL-only job, create
    the group with dist.new_group(backend="gloo") and pass it here."""
    import numpy as np
    import torch

    world = dist.get_world_size(group)

    def gather(packet):
        src = torch.from_numpy(np.ascontiguousarray(packet, dtype=np.uint64).view(np.int64))
        parts = [torch.empty_like(src) for _ in range(world)]
        dist.all_gather(parts, src, group=group)
        return torch.stack(parts).numpy().view(np.uint64).reshape(world, -1)

    return gather


class ShardedIpa:
    """The round loop of ipa::prove (src/ipa/mod.rs:33-95) or poly_commit::prove
    (src/poly_commit/mod.rs:69-105) with the vectors spread over `world` GPUs by residue class of the
    index: rank r keeps the elements i = r (mod world) of G, H, a and b.  A fold pairs i with i + n/2,
    which are in the same class as long as the half length is a multiple of world, so every rank runs
    an ordinary session (bp_ipa_*) over its strided slice and nothing but the round's results crosses
    between GPUs: L = sum_r L_r, R = sum_r R_r, c_l = sum_r c_l,r, c_r = sum_r c_r,r (all linear,
    the u terms included) -- 256 bytes per rank and round in one all-gather.  When the local vectors
    reach length one the world-many remaining elements are gathered and every rank finishes the last
    log2(world) rounds on its own.

    The rank-local and the combining halves of a round are separate methods (local_lr /
    combine_lr, local_tail / begin_tail) so that a single process can drive several shards in
    lockstep; round_lr and round_fold put them together around `gather`."""

    def __init__(self, ctx, curve, variant, gs, hs, a, b, rank=0, world=1, gather=None, u_xy=None, sum_points=None,
                 sum_scalars=None):
        """sum_points(curve, (world, 12) results) -> 12 u64 and sum_scalars(curve, (world, 4)) -> 4 u64 default to
        the host sums of libbp_host (bph_points_sum, bph_scalars_sum); tests may inject stand-ins."""
        import numpy as np

        self.sum_points, self.sum_scalars = sum_points, sum_scalars

        self.np = np
        n = len(a)
        if world < 1 or world & (world - 1) or n % world or n & (n - 1):
            raise ValueError("lengths and world size must be powers of two, world <= n")
        if world > 1 and gather is None:
            raise ValueError("more than one rank needs a gather function (see dist_gather)")
        self.ctx, self.curve, self.variant = ctx, curve, variant
        self.rank, self.world, self.gather = rank, world, gather
        self.u_xy = None if u_xy is None else np.ascontiguousarray(u_xy, dtype=np.uint64).reshape(8)
        self.n = n
        self.one = _mont_one(curve)
        sl = slice(rank, None, world)

        def cut(P):
            return None if P is None else tuple(None if v is None else np.ascontiguousarray(v[sl]) for v in P)

        self.tail = None
        self.local = ctx.ipa_begin(curve, variant, cut(gs), cut(hs), np.ascontiguousarray(a[sl]), np.ascontiguousarray(b[sl]))
        if self.u_xy is not None:
            self.local.set_u(self.u_xy)
        # vectors no longer than the world size: the caller gathers at once (round_lr does it by itself)
        self.wants_tail = world > 1 and n == world

    def __len__(self):
        return self.n

    def _gather(self, packet):
        return packet.reshape(1, -1) if self.world == 1 else self.gather(packet)

    # -- a round's results ---------------------------------------------------------------------
    def local_lr(self):
        """this rank's share of the round: L_r | R_r | c_l,r | c_r,r (32 u64)"""
        L, R, cl, cr = self.local.round_lr()
        return self.np.concatenate([L, R, cl, cr])

    def combine_lr(self, packets):
        """sums of the shares, on the host: `world` point additions per cross term (bph_points_sum) and the
        two field sums; no launch and no copy for values that have just arrived from the device"""
        np = self.np
        packets = np.ascontiguousarray(packets, dtype=np.uint64).reshape(self.world, 32)
        if self.world == 1:
            p = packets[0]
            return p[0:12].copy(), p[12:24].copy(), p[24:28].copy(), p[28:32].copy()
        return (self._sum_points(packets[:, 0:12]), self._sum_points(packets[:, 12:24]),
                self._sum_scalars(packets[:, 24:28]), self._sum_scalars(packets[:, 28:32]))

    def _sum_points(self, xyz):
        if self.sum_points is not None:
            return self.sum_points(self.curve, xyz)
        from .host_mirror import points_sum

        return points_sum(self.curve, xyz)

    def _sum_scalars(self, v):
        if self.sum_scalars is not None:
            return self.sum_scalars(self.curve, v)
        from .host_mirror import scalars_sum

        return scalars_sum(self.curve, v)

    def round_lr(self):
        if self.wants_tail:
            self.begin_tail(self._gather(self.local_tail()))
        if self.tail is not None:
            return self.tail.round_lr()
        return self.combine_lr(self._gather(self.local_lr()))

    # -- the fold ------------------------------------------------------------------------------
    def local_fold(self, x, x_inv):
        """folds this rank's session (or the replicated tail); True when the tail has to be started"""
        if self.tail is not None:
            self.tail.round_fold(x, x_inv)
            self.n //= 2
            return False
        self.local.round_fold(x, x_inv)
        self.n //= 2
        return self.world > 1 and self.n == self.world

    def local_tail(self):
        """this rank's last element: G | H | a | b as x, y, inf-flag word per point (26 u64)"""
        np = self.np
        G, H, a, b = self.local.read()
        pk = np.zeros(26, dtype=np.uint64)
        pk[0:4], pk[4:8], pk[8] = G[0][0], G[1][0], G[2][0]
        if H is not None:
            pk[9:13], pk[13:17], pk[17] = H[0][0], H[1][0], H[2][0]
        pk[18:22], pk[22:26] = a[0], b[0]
        return pk

    def begin_tail(self, packets):
        """every rank continues alone over the world-many elements that are left (rank order = index order)"""
        np = self.np
        p = np.ascontiguousarray(packets, dtype=np.uint64).reshape(self.world, 26)
        G = (np.ascontiguousarray(p[:, 0:4]), np.ascontiguousarray(p[:, 4:8]), p[:, 8].astype(np.uint8))
        H = None
        if self.variant == 0:
            H = (np.ascontiguousarray(p[:, 9:13]), np.ascontiguousarray(p[:, 13:17]), p[:, 17].astype(np.uint8))
        self.tail = self.ctx.ipa_begin(self.curve, self.variant, G, H, np.ascontiguousarray(p[:, 18:22]),
                                       np.ascontiguousarray(p[:, 22:26]))
        if self.u_xy is not None:
            self.tail.set_u(self.u_xy)
        self.wants_tail = False
        self.local.free()

    def round_fold(self, x, x_inv):
        if self.local_fold(x, x_inv):
            self.begin_tail(self._gather(self.local_tail()))

    def read(self):
        """the current vectors: the replicated tail's, or (single rank) the session's own"""
        s = self.tail if self.tail is not None else self.local
        if self.tail is None and self.world > 1:
            raise RuntimeError("vectors are spread over the ranks until their length reaches the world size")
        return s.read()

    def free(self):
        for s in (self.tail, self.local):
            if s is not None:
                s.free()
        self.tail = self.local = None


# ---- the fold of deferred checks over several GPUs -----------------------------------------------

def _limbs_to_int(v):
    return sum(int(x) << (64 * i) for i, x in enumerate(v))


def _int_to_limbs(v):
    import numpy as np

    return np.array([(v >> (64 * i)) & ((1 << 64) - 1) for i in range(4)], dtype=np.uint64)


def fold_todos_witness_shard(ctx, curve, challenges, challenge_inverses, weights, rank=0, world=1):
    """This rank's share of fold_todos_witness (src/poly_commit/mod.rs:228-244): the elements
    i = rank (mod world) of sum_m weights[m] * coeffs(h_poly_m), in index order -- the slice a
    ShardedIpa over the folded witness keeps on this rank, so the witness is built where it is used
    and nothing crosses between GPUs.

    coeffs(h_poly)_i = prod_j (bit j of i ? u[k-1-j] : u[k-1-j]^-1) (src/poly_commit/mod.rs:194-219):
    the low log2(world) bits of i are the rank's, so on a residue class the vector is the s-vector of
    the first k - log2(world) challenges times one constant per deferred check, which goes into that
    check's weight.  challenges, challenge_inverses: (m, k, 4); weights: (m, 4) (the powers of alpha);
    all Montgomery limbs.  Returns (2^k / world, 4)."""
    import numpy as np

    ch = np.ascontiguousarray(challenges, dtype=np.uint64)
    chi = np.ascontiguousarray(challenge_inverses, dtype=np.uint64)
    w = np.ascontiguousarray(weights, dtype=np.uint64).reshape(-1, 4)
    if ch.ndim != 3 or ch.shape != chi.shape or ch.shape[2] != 4 or ch.shape[0] != w.shape[0] or ch.shape[0] == 0:
        raise ValueError("challenges must be (m, k, 4) with m >= 1, one weight per deferred check")
    m, k = ch.shape[0], ch.shape[1]
    if world < 1 or world & (world - 1) or world > (1 << k) or not 0 <= rank < world:
        raise ValueError("world must be a power of two no larger than 2^k, 0 <= rank < world")
    t = world.bit_length() - 1
    p = _FR_MODULUS[curve]
    r_inv = pow(1 << 256, -1, p)  # Montgomery limbs hold v * 2^256: a product of two carries one factor too many
    scaled = np.empty_like(w)
    total = 0
    for i in range(m):
        acc = _limbs_to_int(w[i])
        for j in range(t):
            f = ch[i, k - 1 - j] if (rank >> j) & 1 else chi[i, k - 1 - j]
            acc = acc * _limbs_to_int(f) * r_inv % p
        scaled[i] = _int_to_limbs(acc)
        total = (total + acc) % p
    if k == t:  # one element per rank: the s-vector of no challenges is (1)
        return _int_to_limbs(total).reshape(1, 4)
    return ctx.fold_s_vectors(curve, np.ascontiguousarray(ch[:, :k - t]), np.ascontiguousarray(chi[:, :k - t]), scaled)
