"""Host-side mirror (Python, over the C ABI) of the reference interfaces on the MSM hot path.

Names follow the reference so parity tests read like its own:
  * `VariableBaseMSM.multi_scalar_mul(bases, scalars)`  -- ark_ec::msm (ext), SURVEY.md section 8(a) a1
  * `PedersenCommitment.{setup, commit}` / `CommitterKey` -- ark_poly_commit::trivial_pc (ext), a2;
    call sites src/hp_as/mod.rs:196,377,911
  * `FrVector` = a `Vec<G::ScalarField>` resident in HBM (raw Montgomery memory)

numpy carries the data: scalars are (n, 4) uint64 little-endian limbs; affine points are
(n, 2*limbs) uint64 (x_mont | y_mont) plus an (n,) uint8 infinity mask.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import ffi


class _ArrayPtr(C.c_void_p):
    """Address of a numpy array that keeps the array alive for as long as the pointer object lives."""


def _ptr(a: Optional[np.ndarray]):
    """ndarray -> void* argument.  NOT `a.ctypes.data_as(...)`: numpy builds that pointer through `ctypes.cast`, which
    leaves a reference cycle per call, and with torch's heap loaded the cycle collector those cycles keep triggering
    costs ~50 us per call on average and tens of ms when a full collection hits (measured: 13 of 26 ms of an IPA opening)."""
    if a is None:
        return None
    if not a.flags.c_contiguous:
        raise ValueError("the C ABI takes contiguous arrays")
    p = _ArrayPtr(a.__array_interface__["data"][0])
    p._keep = a
    return p


class Context:
    """One GPU + one HIP stream + workspace (amsm_ctx)."""

    def __init__(self, curve: int = ffi.AMSM_PALLAS, device: int = 0, stream: Optional[int] = None):
        self._lib = ffi.load()
        h = C.c_void_p()
        ffi.check(self._lib.amsm_ctx_create(C.byref(h), curve, device, C.c_void_p(stream) if stream else None),
                  "amsm_ctx_create")
        self._h = h
        self.curve = curve
        self.device = device
        self.fq_limbs = self._lib.amsm_ctx_fq_limbs(h)
        # size-keyed free lists of device buffers: hipMalloc/hipFree synchronise the device, so FrVectors
        # released by the scheme drivers are recycled instead (same idea as a caching tensor allocator)
        self._pool = {}

    def _alloc(self, nbytes: int):
        lst = self._pool.get(nbytes)
        if lst:
            return lst.pop()
        p = C.c_void_p()
        ffi.check(self._lib.amsm_dev_alloc(self._h, nbytes, C.byref(p)), "amsm_dev_alloc")
        return p

    def _release(self, p, nbytes: int):
        if self._h:
            self._pool.setdefault(nbytes, []).append(p)

    def empty_cache(self):
        if self._h:
            self.synchronize()
            for lst in self._pool.values():
                for p in lst:
                    self._lib.amsm_dev_free(self._h, p)
        self._pool = {}

    def close(self):
        if getattr(self, "_h", None):
            self.empty_cache()
            self._lib.amsm_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def is_host(self) -> bool:
        """a context of the library's host backend (amsm_ctx_is_host)"""
        return bool(self._lib.amsm_ctx_is_host(self._h))

    def memory(self) -> dict:
        """Device memory the context holds (amsm_ctx_memory): MSM workspace, live vectors, the allocator's free lists."""
        ws, live, pooled = C.c_size_t(), C.c_size_t(), C.c_size_t()
        ffi.check(self._lib.amsm_ctx_memory(self._h, C.byref(ws), C.byref(live), C.byref(pooled)), "amsm_ctx_memory")
        return {"workspace_bytes": ws.value, "vectors_live_bytes": live.value, "vectors_pooled_bytes": pooled.value}

    def set_table_budget(self, nbytes: int) -> None:
        """bytes ONE key may spend on each of its optional tables (direct-sum table, twin): amsm_ctx_set_table_budget"""
        ffi.check(self._lib.amsm_ctx_set_table_budget(self._h, nbytes), "amsm_ctx_set_table_budget")

    def tables_denied(self) -> int:
        return int(self._lib.amsm_ctx_tables_denied(self._h))

    def two_valued_msms(self) -> int:
        """MSMs of device vectors that took the two-valued form (every scalar 0 or one value v) so far"""
        return int(self._lib.amsm_ctx_two_valued_msms(self._h))

    def pipeline_stats(self) -> dict:
        """MSMs that took the bucket-per-lane pipeline / that fell back to the chunked one (amsm_ctx_pipeline_stats)"""
        a, b = C.c_ulonglong(), C.c_ulonglong()
        ffi.check(self._lib.amsm_ctx_pipeline_stats(self._h, C.byref(a), C.byref(b)), "amsm_ctx_pipeline_stats")
        c, d = C.c_ulonglong(), C.c_ulonglong()
        ffi.check(self._lib.amsm_ctx_pipeline_stats_small(self._h, C.byref(c), C.byref(d)), "amsm_ctx_pipeline_stats_small")
        return {"bucket_per_lane": a.value, "fallbacks": b.value, "bucket_split": c.value, "bucket_split_fallbacks": d.value,
                "direct_sum": int(self._lib.amsm_ctx_direct_sum_msms(self._h)),
                "shared_bucket_sets": int(self._lib.amsm_ctx_shared_bucket_msms(self._h)),
                "unit_scalar_sums": int(self._lib.amsm_ctx_unit_scalar_msms(self._h))}

    def trim(self):
        """Release the MSM workspace and every cached buffer (amsm_ctx_trim); live vectors and keys stay."""
        self.empty_cache()
        ffi.check(self._lib.amsm_ctx_trim(self._h), "amsm_ctx_trim")

    def set_window(self, c_bits: int):
        ffi.check(self._lib.amsm_ctx_set_window(self._h, c_bits), "amsm_ctx_set_window")

    def set_profiling(self, on: bool):
        ffi.check(self._lib.amsm_ctx_set_profiling(self._h, 1 if on else 0), "amsm_ctx_set_profiling")

    def stage_ms(self) -> dict:
        out = {}
        for i in range(self._lib.amsm_stage_count()):
            ms = C.c_float()
            ffi.check(self._lib.amsm_ctx_stage_ms(self._h, i, C.byref(ms)), "amsm_ctx_stage_ms")
            out[self._lib.amsm_stage_name(i).decode()] = ms.value
        return out

    def synchronize(self):
        ffi.check(self._lib.amsm_ctx_synchronize(self._h), "amsm_ctx_synchronize")

    # ---- Fr vectors in HBM ----
    def vector(self, n: int) -> "FrVector":
        return FrVector(self, n)

    def upload(self, limbs: np.ndarray) -> "FrVector":
        limbs = np.ascontiguousarray(limbs, dtype=np.uint64).reshape(-1, 4)
        v = FrVector(self, limbs.shape[0])
        if v.n:
            ffi.check(self._lib.amsm_dev_upload(self._h, v.ptr, _ptr(limbs), limbs.nbytes), "amsm_dev_upload")
        return v

    def fill(self, value_mont: np.ndarray, n: int) -> "FrVector":
        """vec![value; n] on the device."""
        v = FrVector(self, n)
        val = np.ascontiguousarray(value_mont, dtype=np.uint64).reshape(4)
        ffi.check(self._lib.amsm_vec_fill(self._h, _ptr(val), n, v.ptr), "amsm_vec_fill")
        return v

    def host_register(self, arr: np.ndarray) -> None:
        """amsm_host_register: a documented no-op since round 5 (page-locking never gained and sometimes cost: include/amsm.h)."""
        ffi.check(self._lib.amsm_host_register(C.c_void_p(arr.ctypes.data), arr.nbytes), "amsm_host_register")

    def host_unregister(self, arr: np.ndarray) -> None:
        ffi.check(self._lib.amsm_host_unregister(C.c_void_p(arr.ctypes.data)), "amsm_host_unregister")

    def random_vector(self, seed: int, n: int, mont: bool) -> "FrVector":
        v = FrVector(self, n)
        ffi.check(self._lib.amsm_vec_random(self._h, seed, n, 1 if mont else 0, v.ptr), "amsm_vec_random")
        return v


class _ShardContext(Context):
    """Borrowed single-device context of one shard of a MultiContext (amsm_ctx_shard): allocate / fill vectors ON that
    device; destroyed with its parent."""

    def __init__(self, parent: "MultiContext", g: int, handle):
        self._lib = parent._lib
        self._h = handle
        self.curve = parent.curve
        self.device = parent.devices[g]
        self.fq_limbs = parent.fq_limbs
        self._pool = {}
        self._parent = parent

    def close(self):
        if getattr(self, "_h", None):
            self.empty_cache()
            self._h = None  # owned by the parent


class MultiContext(Context):
    """ONE process driving several GPUs (amsm_ctx_create_multi): keys created through it are sharded over the devices,
    the MSM entry points accept them, the partial sums are gathered inside the library (RCCL / peer copies).  The context
    itself is the primary device's; shard(g) gives the per-device contexts."""

    def __init__(self, curve: int = ffi.AMSM_PALLAS, devices=(0,)):
        self._lib = ffi.load()
        self.devices = [int(d) for d in devices]
        ids = (C.c_int * len(self.devices))(*self.devices)
        h = C.c_void_p()
        ffi.check(self._lib.amsm_ctx_create_multi(C.byref(h), curve, ids, len(self.devices)), "amsm_ctx_create_multi")
        self._h = h
        self.curve = curve
        self.device = self.devices[0]
        self.fq_limbs = self._lib.amsm_ctx_fq_limbs(h)
        self._pool = {}
        self._shards = [self] + [_ShardContext(self, g, C.c_void_p(self._lib.amsm_ctx_shard(h, g)))
                                 for g in range(1, len(self.devices))]

    @property
    def num_devices(self) -> int:
        return len(self.devices)

    def set_replicate_below(self, n_generators: int):
        """keys of up to n_generators created from now on are REPLICATED on every device (ffi.AMSM_BASES_REPLICATE) instead of sharded"""
        ffi.check(self._lib.amsm_ctx_set_replicate_below(self._h, n_generators), "amsm_ctx_set_replicate_below")

    @property
    def replicated_msms(self) -> int:
        return int(self._lib.amsm_ctx_replicated_msms(self._h))

    @property
    def collective(self) -> str:
        return self._lib.amsm_ctx_collective(self._h).decode()

    @property
    def collectives(self) -> int:
        """exchanges of partial records so far: one per sharded MSM / commit CALL (amsm_ctx_collectives)"""
        return int(self._lib.amsm_ctx_collectives(self._h))

    def shard(self, g: int) -> Context:
        return self._shards[g]

    def shard_range(self, key: "CommitterKey", g: int):
        lo, hi = C.c_size_t(), C.c_size_t()
        ffi.check(self._lib.amsm_bases_shard_range(key._h, g, C.byref(lo), C.byref(hi)), "amsm_bases_shard_range")
        return lo.value, hi.value

    def msm_batch_sharded(self, key: "CommitterKey", slices, mont: bool):
        """slices[v][g]: FrVector on device g holding shard g's part of vector v -> (points (k, 2L), infinity flags)."""
        k, N = len(slices), self.num_devices
        ptrs = (C.c_void_p * (k * N))(*[slices[v][g].ptr for v in range(k) for g in range(N)])
        out = np.zeros((k, 2 * self.fq_limbs), dtype=np.uint64)
        inf = np.zeros(k, dtype=np.uint8)
        ffi.check(self._lib.amsm_msm_batch_sharded_device(self._h, key._h, ptrs, k, 1 if mont else 0, _ptr(out), _ptr(inf)),
                  "amsm_msm_batch_sharded_device")
        return out, inf

    def close(self):
        if getattr(self, "_h", None):
            for s in self._shards[1:]:
                s.close()
            super().close()


class FrVector:
    """n scalar-field elements (32 B each) in device memory."""

    def __init__(self, ctx: Context, n: int):
        self.ctx = ctx
        self.n = n
        self._nbytes = max(n, 1) * 32
        self.ptr = ctx._alloc(self._nbytes)

    _view_of = None

    def view(self, off: int, n: int) -> "FrVector":
        """Non-owning window [off, off+n) of this vector (keeps the parent alive)."""
        assert 0 <= off and off + n <= self.n
        v = object.__new__(FrVector)
        v.ctx, v.n, v._nbytes = self.ctx, n, 0
        v.ptr = C.c_void_p(self.ptr.value + off * 32)
        v._view_of = self
        return v

    def download(self) -> np.ndarray:
        out = np.empty((self.n, 4), dtype=np.uint64)
        if self.n:
            ffi.check(self.ctx._lib.amsm_dev_download(self.ctx._h, _ptr(out), self.ptr, out.nbytes), "amsm_dev_download")
        return out

    def free(self):
        if self._view_of is not None:
            self.ptr = None
            self._view_of = None
            return
        # stream-ordered reuse: every kernel of this engine runs on the context's stream (or is ordered after
        # it), so a recycled buffer is never overwritten before its last reader has been enqueued
        if self.ptr is not None and self.ctx._h:
            self.ctx._release(self.ptr, self._nbytes)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def __len__(self):
        return self.n


class PointVector:
    """n affine points (x_mont | y_mont, (0,0) = identity) in device memory -- e.g. the folded commitment keys of
    the IPA rounds."""

    _view_of = None

    def __init__(self, ctx: Context, n: int):
        self.ctx, self.n = ctx, n
        self.pb = 16 * ctx.fq_limbs
        self._nbytes = max(n, 1) * self.pb
        self.ptr = ctx._alloc(self._nbytes)

    def view(self, off: int, n: int) -> "PointVector":
        assert 0 <= off and off + n <= self.n
        v = object.__new__(PointVector)
        v.ctx, v.n, v.pb, v._nbytes = self.ctx, n, self.pb, 0
        v.ptr = C.c_void_p(self.ptr.value + off * self.pb)
        v._view_of = self
        return v

    @classmethod
    def of_key(cls, ck: "CommitterKey", n: Optional[int] = None) -> "PointVector":
        """Non-owning view of the first n generators of a key (level 0 of its table)."""
        v = object.__new__(PointVector)
        v.ctx, v.n, v.pb, v._nbytes = ck.ctx, (len(ck) if n is None else n), 16 * ck.ctx.fq_limbs, 0
        v.ptr = C.c_void_p(ck.ctx._lib.amsm_bases_device_ptr(ck._h))
        v._view_of = ck
        return v

    def download(self) -> np.ndarray:
        out = np.empty((self.n, 2 * self.ctx.fq_limbs), dtype=np.uint64)
        if self.n:
            ffi.check(self.ctx._lib.amsm_dev_download(self.ctx._h, _ptr(out), self.ptr, out.nbytes), "amsm_dev_download")
        return out

    def free(self):
        if self._view_of is None and self.ptr is not None and self.ctx._h:
            self.ctx._release(self.ptr, self._nbytes)
        self.ptr = None
        self._view_of = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class CommitterKey:
    """Generators of a Pedersen committer key resident in HBM (amsm_bases).
    Mirrors ark_poly_commit::trivial_pc::CommitterKey{generators, hiding_generator}."""

    def __init__(self, ctx: Context, handle, hiding_generator: Optional[np.ndarray] = None):
        self.ctx = ctx
        self._h = handle
        self.hiding_generator = hiding_generator  # (2*limbs,) uint64 Montgomery affine, host side

    @classmethod
    def load(cls, ctx: Context, xy_mont: np.ndarray, is_inf: Optional[np.ndarray] = None,
             flags: int = ffi.AMSM_BASES_DEFAULT, hiding_generator: Optional[np.ndarray] = None) -> "CommitterKey":
        xy = np.ascontiguousarray(xy_mont, dtype=np.uint64).reshape(-1, 2 * ctx.fq_limbs)
        inf = None if is_inf is None else np.ascontiguousarray(is_inf, dtype=np.uint8)
        h = C.c_void_p()
        ffi.check(ctx._lib.amsm_bases_load(ctx._h, _ptr(xy), _ptr(inf), xy.shape[0], flags, C.byref(h)), "amsm_bases_load")
        return cls(ctx, h, hiding_generator)

    @classmethod
    def from_device(cls, ctx: Context, points: "PointVector", flags: int = ffi.AMSM_BASES_NO_PRECOMPUTE) -> "CommitterKey":
        """Key over a COPY of device-resident points (one-shot bases, e.g. IPA round keys)."""
        h = C.c_void_p()
        ffi.check(ctx._lib.amsm_bases_from_device(ctx._h, points.ptr, points.n, flags, C.byref(h)),
                  "amsm_bases_from_device")
        return cls(ctx, h)

    @classmethod
    def generate(cls, ctx: Context, seed: int, n: int, flags: int = ffi.AMSM_BASES_DEFAULT) -> "CommitterKey":
        h = C.c_void_p()
        ffi.check(ctx._lib.amsm_bases_generate(ctx._h, seed, n, flags, C.byref(h)), "amsm_bases_generate")
        return cls(ctx, h)

    def fold(self, n_half: int, x_limbs: np.ndarray, nbits: int = 255) -> "CommitterKey":
        """New (plain) key of n_half generators: out[i] = self[i] + x * self[n_half + i] -- the key fold
        `key_l += key_r * xi` of the IPA opening (ark_poly_commit::ipa_pc ext, under src/ipa_pc_as/mod.rs:454),
        key to key on the device.  x_limbs: the scalar in Montgomery form."""
        h = C.c_void_p()
        x = np.ascontiguousarray(x_limbs, dtype=np.uint64)
        ffi.check(self.ctx._lib.amsm_bases_fold(self.ctx._h, self._h, n_half, _ptr(x), nbits, C.byref(h)), "amsm_bases_fold")
        return CommitterKey(self.ctx, h)

    def local_num_elems(self) -> int:
        """Generators resident on THIS rank (= supported_num_elems() unless the key is a dist.ShardedCommitterKey)."""
        return self.supported_num_elems()

    def supported_num_elems(self) -> int:
        return int(self.ctx._lib.amsm_bases_len(self._h))

    def __len__(self):
        return self.supported_num_elems()

    @property
    def precomputed(self) -> bool:
        return bool(self.ctx._lib.amsm_bases_precomputed(self._h))

    @property
    def window_bits(self) -> int:
        """Window width of a precomputed key, 0 for a plain one."""
        return int(self.ctx._lib.amsm_bases_window_bits(self._h))

    def memory(self) -> dict:
        """device bytes of the key: its table, the C-ABI copy (if made) and the lazily built 17-bit twin (amsm_bases_memory)"""
        t, a, w = C.c_size_t(), C.c_size_t(), C.c_size_t()
        ffi.check(self.ctx._lib.amsm_bases_memory(self._h, C.byref(t), C.byref(a), C.byref(w)), "amsm_bases_memory")
        return {"table": t.value, "abi_copy": a.value, "twin": w.value}

    @property
    def num_shards(self) -> int:
        """1, or the number of devices the key is sharded over (a key made through a MultiContext)"""
        return int(self.ctx._lib.amsm_bases_num_shards(self._h))

    def tables(self) -> dict:
        """which tables the key holds, one by one, and why an optional one is missing (amsm_bases_tables)"""
        out = (C.c_size_t * 7)()
        ffi.check(self.ctx._lib.amsm_bases_tables(self._h, out), "amsm_bases_tables")
        why = {0: None, 1: "flag", 2: "budget", 3: "allocation failed"}
        return {"window_table": out[0], "direct_sum_table": out[1], "twin": out[2], "abi_copy": out[3], "levels": out[4],
                "direct_sum_table_denied": why[out[5]], "twin_denied": why[out[6]]}

    def prebuild_twin(self) -> None:
        ffi.check(self.ctx._lib.amsm_bases_prebuild_twin(self.ctx._h, self._h), "amsm_bases_prebuild_twin")

    def read(self, off: int = 0, n: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
        n = len(self) - off if n is None else n
        xy = np.empty((n, 2 * self.ctx.fq_limbs), dtype=np.uint64)
        inf = np.empty((n,), dtype=np.uint8)
        ffi.check(self.ctx._lib.amsm_bases_read(self.ctx._h, self._h, off, n, _ptr(xy), _ptr(inf)), "amsm_bases_read")
        return xy, inf

    def free(self):
        if self._h is not None:
            self.ctx._lib.amsm_bases_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class VariableBaseMSM:
    """ark_ec::msm::VariableBaseMSM (ext).  Results are affine (x_mont|y_mont, is_inf)."""

    @staticmethod
    def multi_scalar_mul(bases: CommitterKey, scalars, base_off: int = 0, mont: bool = False
                         ) -> Tuple[np.ndarray, bool]:
        if hasattr(bases, "sharded"):  # dist.ShardedCommitterKey (device-resident slices only)
            assert base_off == 0 and isinstance(scalars, FrVector)
            outs, infs = bases.msm_batch([scalars], mont)
            return outs[0], bool(infs[0])
        ctx = bases.ctx
        out = np.zeros((2 * ctx.fq_limbs,), dtype=np.uint64)
        inf = C.c_uint8(0)
        if isinstance(scalars, FrVector):
            ffi.check(ctx._lib.amsm_msm_device(ctx._h, bases._h, base_off, scalars.ptr, scalars.n, 1 if mont else 0,
                                               _ptr(out), C.byref(inf)), "amsm_msm_device")
        else:
            s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
            ffi.check(ctx._lib.amsm_msm(ctx._h, bases._h, base_off, _ptr(s), s.shape[0], 1 if mont else 0, _ptr(out),
                                        C.byref(inf)), "amsm_msm")
        return out, bool(inf.value)

    @staticmethod
    def multi_scalar_mul_oneshot(ctx, bases_xy: np.ndarray, scalars: np.ndarray, bases_is_inf=None, mont: bool = False
                                 ) -> Tuple[np.ndarray, bool]:
        """`VariableBaseMSM::multi_scalar_mul(&[G], &[BigInt])` literally: host bases AND host scalars belong to this call only
        (amsm_msm_oneshot); min(len(bases), len(scalars)) pairs, as ark-ec takes them."""
        xy = np.ascontiguousarray(bases_xy, dtype=np.uint64).reshape(-1, 2 * ctx.fq_limbs)
        s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        flags = None if bases_is_inf is None else np.ascontiguousarray(bases_is_inf, dtype=np.uint8)
        assert flags is None or flags.shape[0] == xy.shape[0]
        out = np.zeros((2 * ctx.fq_limbs,), dtype=np.uint64)
        inf = C.c_uint8(0)
        ffi.check(ctx._lib.amsm_msm_oneshot(ctx._h, _ptr(xy) if xy.size else None, None if flags is None else _ptr(flags), xy.shape[0],
                                            _ptr(s) if s.size else None, s.shape[0], 1 if mont else 0, _ptr(out), C.byref(inf)),
                  "amsm_msm_oneshot")
        return out, bool(inf.value)

    @staticmethod
    def multi_scalar_mul_batch(bases: CommitterKey, vectors: Sequence[FrVector], mont: bool = True, base_off: int = 0):
        if hasattr(bases, "sharded"):  # dist.ShardedCommitterKey: per-rank partials + one all-gather
            assert base_off == 0
            return bases.msm_batch(vectors, mont)
        ctx = bases.ctx
        k = len(vectors)
        n = vectors[0].n if k else 0
        assert all(v.n == n for v in vectors)
        ptrs = (C.c_void_p * max(k, 1))(*[v.ptr for v in vectors])
        out = np.zeros((k, 2 * ctx.fq_limbs), dtype=np.uint64)
        inf = np.zeros((k,), dtype=np.uint8)
        ffi.check(ctx._lib.amsm_msm_batch_device(ctx._h, bases._h, base_off, ptrs, k, n, 1 if mont else 0, _ptr(out),
                                                 _ptr(inf)), "amsm_msm_batch_device")
        return out, inf


    @staticmethod
    def multi_scalar_mul_batch_host(bases: CommitterKey, vectors: Sequence[np.ndarray], mont: bool = False, base_off: int = 0):
        """len(vectors) MSMs over HOST slices ((n, 4) uint64 each, equal n): amsm_msm_batch -- the upload of vector v + 1
        overlaps MSM v (what back-to-back `commit` calls of a patched ark-poly-commit would go through)."""
        ctx = bases.ctx
        k = len(vectors)
        arrs = [np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4) for v in vectors]
        n = arrs[0].shape[0] if k else 0
        assert all(a.shape[0] == n for a in arrs)
        ptrs = (C.c_void_p * max(k, 1))(*[a.ctypes.data for a in arrs])
        out = np.zeros((k, 2 * ctx.fq_limbs), dtype=np.uint64)
        inf = np.zeros((k,), dtype=np.uint8)
        ffi.check(ctx._lib.amsm_msm_batch(ctx._h, bases._h, base_off, ptrs, k, n, 1 if mont else 0, _ptr(out), _ptr(inf)),
                  "amsm_msm_batch")
        return out, inf

    @staticmethod
    def multi_scalar_mul_multi(bases: CommitterKey, jobs: Sequence[Tuple[int, "FrVector"]], mont: bool = True):
        """Independent MSMs over windows of one key, pipelined on the device: job = (base_off, scalars); MSM j uses
        generators [base_off, base_off + len(scalars)).  Returns (k x 2L u64, k uint8)."""
        ctx = bases.ctx
        k = len(jobs)
        offs = (C.c_size_t * max(k, 1))(*[int(o) for o, _ in jobs])
        ns = (C.c_size_t * max(k, 1))(*[v.n for _, v in jobs])
        ptrs = (C.c_void_p * max(k, 1))(*[v.ptr for _, v in jobs])
        out = np.zeros((k, 2 * ctx.fq_limbs), dtype=np.uint64)
        inf = np.zeros((k,), dtype=np.uint8)
        ffi.check(ctx._lib.amsm_msm_multi_device(ctx._h, bases._h, k, offs, ptrs, ns, 1 if mont else 0, _ptr(out),
                                                 _ptr(inf)), "amsm_msm_multi_device")
        return out, inf


    @staticmethod
    def multi_scalar_mul_grouped(bases: CommitterKey, scalars: "FrVector", group_shift: int, mont: bool = True,
                                 base_off: int = 0):
        """Two MSMs in one pass: result g = sum over the i with ((i >> group_shift) & 1) == g.  -> (2 x 2L u64, 2 uint8)"""
        ctx = bases.ctx
        out = np.zeros((2, 2 * ctx.fq_limbs), dtype=np.uint64)
        inf = np.zeros((2,), dtype=np.uint8)
        ffi.check(ctx._lib.amsm_msm_grouped_device(ctx._h, bases._h, base_off, scalars.ptr, scalars.n, 1 if mont else 0,
                                                   group_shift, _ptr(out), _ptr(inf)), "amsm_msm_grouped_device")
        return out, inf


class PedersenCommitment:
    """ark_poly_commit::trivial_pc::PedersenCommitment (ext): setup / commit."""

    @staticmethod
    def setup(ctx: Context, n: int, seed: int = 0x5EED1001, flags: int = ffi.AMSM_BASES_DEFAULT) -> CommitterKey:
        """n generators + one hiding generator (the (n+1)-th point of the synthetic stream)."""
        ck = CommitterKey.generate(ctx, seed, n + 1, ffi.AMSM_BASES_NO_PRECOMPUTE)
        xy, _ = ck.read(0, n + 1)
        ck.free()
        out = CommitterKey.load(ctx, xy[:n], None, flags, hiding_generator=xy[n].copy())
        return out

    @staticmethod
    def commit_batch(ck: CommitterKey, vectors: Sequence["FrVector"], randomizers: Sequence[Optional[np.ndarray]]):
        """Several independent commitments to device vectors of one length as ONE pipelined MSM batch; the hiding terms
        randomizer * hiding_generator are added on the host.  Same points as len(vectors) calls of commit()."""
        ctx = ck.ctx
        pts, infs = VariableBaseMSM.multi_scalar_mul_batch(ck, list(vectors), mont=True)
        out = []
        one = None
        for i, r in enumerate(randomizers):
            if r is None:
                out.append((pts[i], bool(infs[i])))
                continue
            if ck.hiding_generator is None:
                raise ValueError("committer key has no hiding generator")
            if one is None:
                one = np.zeros(4, dtype=np.uint64)
                ffi.check(ctx._lib.amsm_fr_to_mont(ctx.curve, _ptr(np.array([1, 0, 0, 0], dtype=np.uint64)), 1, _ptr(one)), "to_mont")
            xy = np.stack([np.asarray(pts[i], dtype=np.uint64), np.asarray(ck.hiding_generator, dtype=np.uint64)])
            inf = np.array([1 if infs[i] else 0, 0], dtype=np.uint8)
            sc = np.stack([one, np.ascontiguousarray(r, dtype=np.uint64).reshape(4)])
            o = np.zeros((2 * ctx.fq_limbs,), dtype=np.uint64)
            oinf = C.c_uint8(0)
            ffi.check(ctx._lib.amsm_host_lincomb(ctx.curve, _ptr(xy), _ptr(inf), _ptr(sc), 2, _ptr(o), C.byref(oinf)),
                      "amsm_host_lincomb")
            out.append((o, bool(oinf.value)))
        return out

    @staticmethod
    def commit_batch_host(ck: CommitterKey, elems: Sequence[np.ndarray], randomizers: Optional[Sequence[Optional[np.ndarray]]] = None):
        """commit(ck, elems[v], randomizers[v]) for HOST vectors of Montgomery elements (lengths may differ) in one call:
        amsm_pedersen_commit_batch.  Returns [(point, is_inf), ...]."""
        ctx = ck.ctx
        k = len(elems)
        arrs = [np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4) for v in elems]
        ptrs = (C.c_void_p * max(k, 1))(*[a.ctypes.data for a in arrs])
        ns = (C.c_size_t * max(k, 1))(*[a.shape[0] for a in arrs])
        rptrs, keep = None, []
        if randomizers is not None and any(r is not None for r in randomizers):
            if ck.hiding_generator is None:
                raise ValueError("committer key has no hiding generator")
            keep = [None if r is None else np.ascontiguousarray(r, dtype=np.uint64).reshape(4) for r in randomizers]
            rptrs = (C.c_void_p * max(k, 1))(*[None if r is None else r.ctypes.data for r in keep])
        out = np.zeros((k, 2 * ctx.fq_limbs), dtype=np.uint64)
        inf = np.zeros((k,), dtype=np.uint8)
        # (bound to a local: the pointer must not outlive a temporary copy np.ascontiguousarray may make)
        hg_arr = None if rptrs is None else np.ascontiguousarray(ck.hiding_generator, dtype=np.uint64)
        ffi.check(ctx._lib.amsm_pedersen_commit_batch(ctx._h, ck._h, ptrs, ns, k, rptrs, _ptr(hg_arr), _ptr(out), _ptr(inf)),
                  "amsm_pedersen_commit_batch")
        del hg_arr, keep, arrs
        return [(out[i], bool(inf[i])) for i in range(k)]

    @staticmethod
    def commit(ck: CommitterKey, elems, randomizer: Optional[np.ndarray] = None) -> Tuple[np.ndarray, bool]:
        """commit(ck, &[F] (Montgomery), Option<F>) -> affine point."""
        if hasattr(ck, "sharded"):  # dist.ShardedCommitterKey
            return ck.commit(elems, randomizer)
        ctx = ck.ctx
        r = None if randomizer is None else np.ascontiguousarray(randomizer, dtype=np.uint64)
        hg = None
        if r is not None:
            if ck.hiding_generator is None:
                raise ValueError("committer key has no hiding generator")
            hg = np.ascontiguousarray(ck.hiding_generator, dtype=np.uint64)
        if isinstance(elems, FrVector):
            out = np.zeros((2 * ctx.fq_limbs,), dtype=np.uint64)
            inf = C.c_uint8(0)
            ffi.check(ctx._lib.amsm_pedersen_commit_device(ctx._h, ck._h, elems.ptr, elems.n, _ptr(r), _ptr(hg),
                                                           _ptr(out), C.byref(inf)), "amsm_pedersen_commit_device")
            return out, bool(inf.value)
        e = np.ascontiguousarray(elems, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros((2 * ctx.fq_limbs,), dtype=np.uint64)
        inf = C.c_uint8(0)
        ffi.check(ctx._lib.amsm_pedersen_commit(ctx._h, ck._h, _ptr(e), e.shape[0], _ptr(r), _ptr(hg), _ptr(out),
                                                C.byref(inf)), "amsm_pedersen_commit")
        return out, bool(inf.value)
