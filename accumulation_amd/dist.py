"""Point-sharded MSM across the GPUs of one node (one process per GPU, torch.distributed).

SURVEY.md section 8(e): sum_i s_i*G_i over disjoint index ranges are independent; the total is the EC sum of
the per-rank partials.  Rank r keeps generators [lo_r, hi_r) resident (loaded once per key) and gets only
its slice of the scalars per call.  The only exchange is an all-gather of one fixed-size partial record
per rank (128 B Pallas / 192 B BLS12-381) as raw bytes -- EC addition is not an RCCL reduction op -- after
which every rank folds the records identically.  With backend "nccl" the all-gather runs over RCCL/xGMI.

The reference has no counterpart (single process, CPU only); this is the new component C1 of SURVEY.md section 2.1.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np


def shard_bounds(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced ranges: the first n_total % world ranks get one extra element."""
    q, r = divmod(n_total, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


class HipEngine:
    """Per-rank engine: libamsm.so context + this rank's shard of the committer key."""

    def __init__(self, ctx, ck):
        import torch
        self.ctx, self.ck = ctx, ck
        self.record_bytes = int(ctx._lib.amsm_partial_bytes(ctx._h))
        # (a context of the library's host backend keeps its "device" memory on the host: the records are CPU tensors, the exchange
        # runs over gloo -- tests/test_dist_cpu.py drives the product library this way without a GPU)
        self._dev = "cpu" if int(ctx.device) < 0 else f"cuda:{ctx.device}"
        self._partial = torch.empty(self.record_bytes, dtype=torch.uint8, device=self._dev)

    def partial(self, scalars, mont: bool):
        """scalars: FrVector (this rank's slice).  Returns a uint8 CUDA tensor holding the record."""
        from . import ffi
        ffi.check(self.ctx._lib.amsm_msm_partial_device(self.ctx._h, self.ck._h, 0, scalars.ptr, scalars.n,
                                                        1 if mont else 0, C.c_void_p(self._partial.data_ptr())),
                  "amsm_msm_partial_device")
        return self._partial

    def combine(self, gathered, count: int):
        from . import ffi
        from .engine import _ptr
        out = np.zeros((2 * self.ctx.fq_limbs,), dtype=np.uint64)
        inf = C.c_uint8(0)
        ffi.check(self.ctx._lib.amsm_partials_combine(self.ctx._h, C.c_void_p(gathered.data_ptr()), count, _ptr(out),
                                                      C.byref(inf)), "amsm_partials_combine")
        return out, bool(inf.value)

    def partial_batch(self, vecs, mont: bool):
        """vecs: FrVectors of equal length (this rank's slices of len(vecs) MSMs, pipelined on the device).
        Returns a uint8 CUDA tensor of len(vecs) consecutive records."""
        import torch
        from . import ffi
        k = len(vecs)
        # torch.empty: no fill kernel on torch's stream that could land after the engine's write on its own stream
        out = torch.empty(max(k, 1) * self.record_bytes, dtype=torch.uint8, device=self._dev)
        ptrs = (C.c_void_p * max(k, 1))(*[v.ptr for v in vecs])
        ffi.check(self.ctx._lib.amsm_msm_partial_batch_device(self.ctx._h, self.ck._h, 0, ptrs, k, vecs[0].n if k else 0,
                                                              1 if mont else 0, C.c_void_p(out.data_ptr())),
                  "amsm_msm_partial_batch_device")
        return out[: k * self.record_bytes]

    def combine_batch(self, grouped, n_groups: int, count: int):
        """grouped: n_groups groups of `count` consecutive records.  Returns (n_groups x 2L u64, n_groups bools)."""
        from . import ffi
        from .engine import _ptr
        out = np.zeros((max(n_groups, 1), 2 * self.ctx.fq_limbs), dtype=np.uint64)
        inf = np.zeros((max(n_groups, 1),), dtype=np.uint8)
        ffi.check(self.ctx._lib.amsm_partials_combine_batch(self.ctx._h, C.c_void_p(grouped.data_ptr()), n_groups, count,
                                                            _ptr(out), _ptr(inf)), "amsm_partials_combine_batch")
        return out[:n_groups], inf[:n_groups].astype(bool)


class ShardedMSM:
    """msm(local_scalars) -> the affine result of the WHOLE (all ranks) MSM, identical on every rank."""

    def __init__(self, engine, group=None, force_collective: bool = False):
        """force_collective: run the all-gather even when the group has ONE rank (round 4: a one-GPU box then executes the
        RCCL call an 8-GPU node will -- tests/test_rccl_gpu.py)"""
        import torch.distributed as dist
        self.engine = engine
        self.group = group
        self.force_collective = bool(force_collective) and dist.is_initialized()
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    @staticmethod
    def _settle(t):
        """The engine reads the gathered records on ITS stream: finish torch's (the collective and the regrouping copy
        are ordered on torch's current stream) before handing the buffer over."""
        if t.is_cuda:
            import torch
            torch.cuda.current_stream(t.device).synchronize()

    def _all_gather(self, parts):
        """All ranks' records, rank-major.  RCCL gathers device tensors in place; a CPU-only backend (gloo: the
        2-process tests on one GPU) is fed through host memory."""
        import torch
        import torch.distributed as dist
        if parts.is_cuda and dist.get_backend(self.group) == "gloo":
            host = parts.cpu()
            out = torch.empty(host.numel() * self.world, dtype=torch.uint8)
            dist.all_gather_into_tensor(out, host, group=self.group)
            return out.to(parts.device)
        out = torch.empty(parts.numel() * self.world, dtype=torch.uint8, device=parts.device)
        dist.all_gather_into_tensor(out, parts, group=self.group)
        return out

    def msm(self, local_scalars, mont: bool = True):
        part = self.engine.partial(local_scalars, mont)
        if self.world == 1 and not self.force_collective:
            return self.engine.combine(part, 1)
        gathered = self._all_gather(part)
        self._settle(gathered)
        return self.engine.combine(gathered, self.world)

    def msm_batch(self, local_vecs, mont: bool = True):
        """len(local_vecs) whole-job MSMs with ONE exchange: every rank runs its slices back to back (pipelined on
        the device), the ranks all-gather all their records at once, and every rank folds group j = the ranks'
        records of MSM j.  Returns (k x 2L u64 array, k bools), identical on every rank."""
        import torch
        import torch.distributed as dist
        k = len(local_vecs)
        rec = self.engine.record_bytes
        if hasattr(self.engine, "partial_batch"):
            parts = self.engine.partial_batch(local_vecs, mont)
        else:  # engines without a batched entry point: one record at a time
            parts = torch.cat([self.engine.partial(v, mont).clone() for v in local_vecs]) if k else \
                torch.zeros(0, dtype=torch.uint8)
        if self.world > 1 or self.force_collective:
            gathered = self._all_gather(parts.contiguous())
            # [rank][msm][record] -> [msm][rank][record]
            grouped = gathered.view(self.world, k, rec).permute(1, 0, 2).contiguous().view(-1)
            self._settle(grouped)
        else:
            grouped = parts
        if hasattr(self.engine, "combine_batch"):
            return self.engine.combine_batch(grouped, k, self.world)
        outs, infs = [], []
        for j in range(k):
            o_, i_ = self.engine.combine(grouped[j * self.world * rec:(j + 1) * self.world * rec], self.world)
            outs.append(o_)
            infs.append(i_)
        return np.array(outs, dtype=np.uint64), np.array(infs, dtype=bool)


class ReplicatedMSM:
    """The one-process-per-GPU form of a REPLICATED key (round 6; the one-process form is amsm.h AMSM_BASES_REPLICATE): every
    rank holds the WHOLE committer key and the whole scalar vectors -- what a small key wants, where point-sharding 2^18
    generators eight ways leaves every GPU a 2^15-pair sliver per MSM -- and MSM v of a batch runs, whole, on rank v mod world.
    The data path has no exchange of partial sums at all; because every rank of a torchrun job needs every result, the ranks
    all-gather the finished AFFINE points once per batch (2L + 1 u64 each).  `r1cs_nark_as::prove` issues 2-8 independent
    commitments per round (src/r1cs_nark_as/r1cs_nark/mod.rs:216-218,234-236,251,261; src/r1cs_nark_as/mod.rs:394-410)."""

    def __init__(self, ck, group=None):
        import torch.distributed as dist
        self.ck, self.ctx, self.group = ck, ck.ctx, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def msm_batch(self, vecs, mont: bool = True):
        """len(vecs) MSMs over the key -> (k x 2L u64, k bools), identical on every rank, in call order"""
        import torch
        import torch.distributed as dist
        from .engine import VariableBaseMSM
        k, L2 = len(vecs), 2 * self.ctx.fq_limbs
        mine = [v for i, v in enumerate(vecs) if i % self.world == self.rank]
        slots = -(-k // self.world) if k else 0
        buf = np.zeros((max(slots, 1), L2 + 1), dtype=np.uint64)
        if mine:
            pts, infs = VariableBaseMSM.multi_scalar_mul_batch(self.ck, mine, mont=mont)
            buf[: len(mine), :L2] = pts
            buf[: len(mine), L2] = np.asarray(infs, dtype=np.uint64)
        if self.world == 1:
            allb = buf[None]
        else:
            # (affine results live on the host: gathered as CPU tensors over gloo, or through a device tensor over RCCL)
            t = torch.from_numpy(buf.view(np.int64).copy()).reshape(-1)
            if dist.get_backend(self.group) == "nccl":
                t = t.to(f"cuda:{self.ctx.device}")
            out = torch.empty(self.world * t.numel(), dtype=t.dtype, device=t.device)
            dist.all_gather_into_tensor(out, t, group=self.group)
            allb = out.cpu().numpy().view(np.uint64).reshape((self.world,) + buf.shape)
        pts = np.zeros((k, L2), dtype=np.uint64)
        infs = np.zeros((k,), dtype=bool)
        for v in range(k):
            row = allb[v % self.world][v // self.world]
            pts[v], infs[v] = row[:L2], bool(row[L2])
        return pts, infs


class ShardedCommitterKey:
    """A committer key whose generators are spread over the ranks: rank r holds [lo_r, hi_r) (`shard_bounds`) as an
    ordinary HBM-resident `engine.CommitterKey`, and every vector the schemes commit to is sharded the same way (rank r
    holds elements [lo_r, hi_r)).  `VariableBaseMSM.multi_scalar_mul_batch` and `PedersenCommitment.commit` recognise
    the type and go through `ShardedMSM` -- per-rank partial records, ONE all-gather per batch, identical fold on every
    rank -- so the scheme drivers (hp_as: every O(len) step is elementwise, src/hp_as/mod.rs:278-349,482-512) run
    unchanged on a slice and produce the accumulator instance of the unsharded run, bit for bit; the accumulator's
    witness vectors stay sharded.  `supported_num_elems()` is the GLOBAL length (what the statement absorbs,
    src/hp_as/mod.rs:747); `local_num_elems()` the slice.  Randomised runs need the same rng stream on every rank."""

    def __init__(self, local_key, n_global: int, group=None):
        self.local = local_key
        self.ctx = local_key.ctx
        self.hiding_generator = local_key.hiding_generator
        self.n_global = int(n_global)
        self.sharded = ShardedMSM(HipEngine(self.ctx, local_key), group)
        lo, hi = shard_bounds(self.n_global, self.sharded.rank, self.sharded.world)
        if hi - lo != len(local_key):
            raise ValueError(f"rank {self.sharded.rank}: local key has {len(local_key)} generators, shard is [{lo}, {hi})")
        self.lo, self.hi = lo, hi

    @classmethod
    def from_global(cls, ctx, xy_mont, hiding_generator=None, flags=None, group=None):
        """Every rank passes the same global generator array (n x 2L u64) and keeps only its slice on the device."""
        import torch.distributed as dist
        from . import ffi
        from .engine import CommitterKey
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        lo, hi = shard_bounds(len(xy_mont), rank, world)
        local = CommitterKey.load(ctx, xy_mont[lo:hi], None, ffi.AMSM_BASES_DEFAULT if flags is None else flags,
                                  hiding_generator=hiding_generator)
        return cls(local, len(xy_mont), group)

    def supported_num_elems(self) -> int:
        return self.n_global

    def local_num_elems(self) -> int:
        return self.hi - self.lo

    def __len__(self):
        return self.n_global

    def msm_batch(self, vectors, mont: bool = True):
        """-> (k x 2L u64, k uint8): whole-job MSMs of the ranks' slices (all the same length on this rank)."""
        outs, infs = self.sharded.msm_batch(list(vectors), mont)
        return outs, np.asarray(infs, dtype=np.uint8)

    def commit(self, elems, randomizer=None):
        """PedersenCommitment::commit over the sharded key: msm + randomizer * hiding_generator (added once, on the host)."""
        outs, infs = self.msm_batch([elems], True)
        pt = (outs[0], bool(infs[0]))
        if randomizer is None:
            return pt
        if self.hiding_generator is None:
            raise ValueError("committer key has no hiding generator")
        from . import ffi
        from .engine import _ptr
        from .scalar_field import Fr
        ctx = self.ctx
        xy = np.stack([np.asarray(pt[0], dtype=np.uint64), np.asarray(self.hiding_generator, dtype=np.uint64)])
        infs = np.array([1 if pt[1] else 0, 0], dtype=np.uint8)
        sc = np.stack([Fr(ctx.curve).to_limbs(1), np.ascontiguousarray(randomizer, dtype=np.uint64).reshape(4)])
        out = np.zeros((2 * ctx.fq_limbs,), dtype=np.uint64)
        inf = C.c_uint8(0)
        ffi.check(ctx._lib.amsm_host_lincomb(ctx.curve, _ptr(xy), _ptr(infs), _ptr(sc), 2, _ptr(out), C.byref(inf)),
                  "amsm_host_lincomb")
        return out, bool(inf.value)
