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
        self._partial = torch.empty(self.record_bytes, dtype=torch.uint8, device=f"cuda:{ctx.device}")

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
        out = torch.empty(max(k, 1) * self.record_bytes, dtype=torch.uint8, device=f"cuda:{self.ctx.device}")
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

    def __init__(self, engine, group=None):
        import torch.distributed as dist
        self.engine = engine
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._gathered = None

    @staticmethod
    def _settle(t):
        """The engine reads the gathered records on ITS stream: finish torch's (the collective and the regrouping copy
        are ordered on torch's current stream) before handing the buffer over."""
        if t.is_cuda:
            import torch
            torch.cuda.current_stream(t.device).synchronize()

    def msm(self, local_scalars, mont: bool = True):
        import torch
        import torch.distributed as dist
        part = self.engine.partial(local_scalars, mont)
        if self.world == 1:
            return self.engine.combine(part, 1)
        if self._gathered is None or self._gathered.numel() != part.numel() * self.world:
            self._gathered = torch.empty(part.numel() * self.world, dtype=torch.uint8, device=part.device)
        dist.all_gather_into_tensor(self._gathered, part, group=self.group)
        self._settle(self._gathered)
        return self.engine.combine(self._gathered, self.world)

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
        if self.world > 1:
            gathered = torch.empty(self.world * k * rec, dtype=torch.uint8, device=parts.device)
            dist.all_gather_into_tensor(gathered, parts.contiguous(), group=self.group)
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
