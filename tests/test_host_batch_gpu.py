"""Host-slice batches (round 3: amsm_msm_batch, amsm_pedersen_commit_batch): the reference's provers hand `commit` host
vectors back to back (src/hp_as/mod.rs:372-385, :196-214, src/r1cs_nark_as/r1cs_nark/mod.rs:216-218); the library overlaps the
upload of vector v + 1 with the MSM of vector v.  Results must equal the one-at-a-time entry points and the CPU restatement
for every shape the callers produce: equal and ragged lengths, with and without randomizers, empty vectors, more vectors
than pipeline slots, a constant vector in the middle of a 2^20 batch (bucket-per-lane fallback re-run from the staging ring),
vectors longer than one window of the key, both curves."""
import numpy as np
import pytest

from oracle import pyref as o
from tests import helpers as h

pytestmark = pytest.mark.gpu
CURVES = [o.PALLAS, o.BLS12_381_G1]


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("n", [0, 1, 1000, 1 << 14])
def test_msm_batch_host_equals_single_calls(cref, c, n):
    from accumulation_amd import CommitterKey, Context, VariableBaseMSM
    ctx = Context(c.curve_id)
    try:
        ck = CommitterKey.generate(ctx, 3, max(n, 1))
        xy, _ = ck.read()
        vecs = [cref.rng_scalars(20 + j, n) for j in range(7)]  # more than the three pipeline slots, more than the ring
        pts, infs = VariableBaseMSM.multi_scalar_mul_batch_host(ck, vecs)
        for j, v in enumerate(vecs):
            ref, rinf = cref.msm(c.curve_id, xy[:n], v, threads=4)
            assert bool(infs[j]) == bool(rinf) and np.array_equal(pts[j], ref), j
        mont = [cref.fr_to_mont(c.curve_id, v) for v in vecs[:3]]
        pts2, infs2 = VariableBaseMSM.multi_scalar_mul_batch_host(ck, mont, mont=True)
        assert np.array_equal(pts2, pts[:3]) and np.array_equal(infs2, infs[:3])
        assert VariableBaseMSM.multi_scalar_mul_batch_host(ck, [])[0].shape[0] == 0
        ck.free()
    finally:
        ctx.close()


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_pedersen_commit_batch_ragged_with_and_without_randomizers(cref, c):
    from accumulation_amd import Context, PedersenCommitment
    ctx = Context(c.curve_id)
    try:
        n = 3000
        ck = PedersenCommitment.setup(ctx, n, seed=17)
        lens = [n, 1234, 0, n, 1, 2999]
        elems = [cref.fr_to_mont(c.curve_id, cref.rng_scalars(40 + j, ln)) for j, ln in enumerate(lens)]
        rnd = [None, cref.fr_to_mont(c.curve_id, cref.rng_scalars(50, 1))[0], cref.fr_to_mont(c.curve_id, cref.rng_scalars(51, 1))[0],
               None, None, cref.fr_to_mont(c.curve_id, cref.rng_scalars(52, 1))[0]]
        got = PedersenCommitment.commit_batch_host(ck, elems, rnd)
        for j in range(len(lens)):
            exp = PedersenCommitment.commit(ck, elems[j], rnd[j])
            assert bool(got[j][1]) == bool(exp[1]) and np.array_equal(got[j][0], exp[0]), j
        plain = PedersenCommitment.commit_batch_host(ck, elems)
        for j in range(len(lens)):
            exp = PedersenCommitment.commit(ck, elems[j], None)
            assert bool(plain[j][1]) == bool(exp[1]) and np.array_equal(plain[j][0], exp[0]), j
        ck.free()
    finally:
        ctx.close()


@pytest.mark.parametrize("probe", ["1", "0"], ids=["probe", "no_probe"])
def test_batch_at_2p20_with_a_constant_vector_in_the_middle(cref, probe):
    """five host vectors of 2^20 Pallas scalars, the third constant: the bucket-per-lane MSMs overlap the uploads; the constant
    one goes chunked -- picked out by the host-side digit probe, or (AMSM_BPL_PROBE=0) attempted and re-run FROM THE STAGING
    RING after its successors were uploaded"""
    import os
    from accumulation_amd import CommitterKey, Context, VariableBaseMSM
    c = o.PALLAS
    n = 1 << 20
    os.environ["AMSM_BPL_PROBE"] = probe
    os.environ["AMSM_TWO_VALUED"] = "0"  # (with it a slice that looks two-valued sends the call down the device path: next test)
    try:
        ctx = Context(c.curve_id)
    finally:
        del os.environ["AMSM_BPL_PROBE"], os.environ["AMSM_TWO_VALUED"]
    try:
        ck = CommitterKey.generate(ctx, 0x5EED1001, n)
        xy, _ = ck.read()
        vecs = [cref.rng_scalars(70 + j, n) for j in range(5)]
        vecs[2] = np.tile(h.scalars_to_np([o.rng_scalar(71, 0)]), (n, 1))
        before = ctx.pipeline_stats()
        pts, infs = VariableBaseMSM.multi_scalar_mul_batch_host(ck, vecs)
        after = ctx.pipeline_stats()
        took, fell = after["bucket_per_lane"] - before["bucket_per_lane"], after["fallbacks"] - before["fallbacks"]
        assert (took, fell) == ((4, 0) if probe == "1" else (5, 1))
        for j, v in enumerate(vecs):
            ref, rinf = cref.msm(c.curve_id, xy, v, threads=17)
            assert bool(infs[j]) == bool(rinf) and np.array_equal(pts[j], ref), j
        ck.free()
    finally:
        ctx.close()


def ffi_mod():
    from accumulation_amd import ffi
    return ffi


def test_batch_at_2p20_with_a_constant_slice_takes_the_two_valued_form(cref):
    """the same five host slices with the two-valued shortcut on (the default): the constant one is v * (the sum of the
    generators) -- no skew fallback, no twin -- and the four uniform ones run bucket-per-lane behind it"""
    from accumulation_amd import CommitterKey, Context, VariableBaseMSM
    c = o.PALLAS
    n = 1 << 20
    ctx = Context(c.curve_id)
    try:
        ck = CommitterKey.generate(ctx, 0x5EED1001, n)
        xy, _ = ck.read()
        vecs = [cref.rng_scalars(70 + j, n) for j in range(5)]
        vecs[2] = np.tile(h.scalars_to_np([o.rng_scalar(71, 0)]), (n, 1))
        before, tv0 = ctx.pipeline_stats(), ctx.two_valued_msms()
        pts, infs = VariableBaseMSM.multi_scalar_mul_batch_host(ck, vecs)
        after = ctx.pipeline_stats()
        assert (after["bucket_per_lane"] - before["bucket_per_lane"], after["fallbacks"] - before["fallbacks"]) == (4, 0)
        assert ctx.two_valued_msms() - tv0 == 1 and ck.memory()["twin"] == 0
        for j, v in enumerate(vecs):
            ref, rinf = cref.msm(c.curve_id, xy, v, threads=17)
            assert bool(infs[j]) == bool(rinf) and np.array_equal(pts[j], ref), j
        ck.free()
    finally:
        ctx.close()


def test_vectors_longer_than_a_window_of_the_key(cref):
    """2^21-pair host vectors over a 2^21-generator key: each is cut into two 2^20 windows whose uploads interleave"""
    from accumulation_amd import CommitterKey, Context, VariableBaseMSM
    c = o.PALLAS
    n = 1 << 21
    ctx = Context(c.curve_id)
    try:
        ck = CommitterKey.generate(ctx, 0x5EED1001, n)
        xy, _ = ck.read()
        vecs = [cref.rng_scalars(80 + j, n) for j in range(3)]
        pts, infs = VariableBaseMSM.multi_scalar_mul_batch_host(ck, vecs)
        for j, v in enumerate(vecs):
            ref, rinf = cref.msm(c.curve_id, xy, v, threads=17)
            assert bool(infs[j]) == bool(rinf) and np.array_equal(pts[j], ref), j
        ck.free()
    finally:
        ctx.close()


def test_page_locked_slices_same_results_and_register_is_a_no_op(cref):
    """round 5: amsm_host_register / _unregister are documented no-ops (page-locking never gained throughput and cost some:
    include/amsm.h, profiles/r05_host_slices.md) -- memory the CALLER page-locked itself (here: torch's pinned allocator) is still
    taken as it is and gives the same points as pageable slices."""
    import ctypes as C
    import torch
    from accumulation_amd import CommitterKey, Context, VariableBaseMSM
    c = o.PALLAS
    ctx = Context(c.curve_id)
    try:
        n = 1 << 17
        ck = CommitterKey.generate(ctx, 5, n)
        xy, _ = ck.read()
        vecs = [np.ascontiguousarray(cref.rng_scalars(70 + j, n)) for j in range(9)]
        ref, rinf = VariableBaseMSM.multi_scalar_mul_batch_host(ck, vecs)
        lib = ctx._lib
        assert lib.amsm_host_is_pinned(C.c_void_p(vecs[0].ctypes.data)) == 0
        ctx.host_register(vecs[0])  # no-op: nothing is page-locked behind the caller's back
        assert lib.amsm_host_is_pinned(C.c_void_p(vecs[0].ctypes.data)) == 0
        ctx.host_unregister(vecs[0])
        assert lib.amsm_host_register(None, 8) == ffi_mod().AMSM_E_INVALID_ARG and lib.amsm_host_unregister(None) == ffi_mod().AMSM_E_INVALID_ARG
        keep = [torch.from_numpy(v.view(np.int64)).pin_memory() for v in vecs]
        pinned = [t.numpy().view(np.uint64) for t in keep]
        assert all(lib.amsm_host_is_pinned(C.c_void_p(v.ctypes.data)) == 1 for v in pinned)
        pts, infs = VariableBaseMSM.multi_scalar_mul_batch_host(ck, pinned)
        assert np.array_equal(pts, ref) and np.array_equal(infs, rinf)
        one, one_inf = VariableBaseMSM.multi_scalar_mul(ck, pinned[4])  # amsm_msm from a page-locked slice
        assert np.array_equal(one, ref[4]) and bool(one_inf) == bool(rinf[4])
        mixed = VariableBaseMSM.multi_scalar_mul_batch_host(ck, pinned[:3] + [vecs[3]] + pinned[4:6])  # one pageable among them
        assert np.array_equal(mixed[0], ref[:6])
        del pinned, keep
        got, ginf = cref.msm(c.curve_id, xy, vecs[2], threads=8)
        assert np.array_equal(ref[2], got) and bool(rinf[2]) == bool(ginf)
        ck.free()
    finally:
        ctx.close()


def test_skewed_host_slices_at_bucket_split_sizes(cref):
    """round-3 ADVICE: host slices of 2^16 .. 2^17 pairs are probed like device vectors -- a constant slice (the zk provers'
    hiding vectors) goes straight to the chunked pipeline instead of an aborted bucket-split attempt"""
    import os
    from accumulation_amd import CommitterKey, Context, VariableBaseMSM
    c = o.PALLAS
    os.environ["AMSM_TWO_VALUED"] = "0"
    try:
        ctx = Context(c.curve_id)
    finally:
        del os.environ["AMSM_TWO_VALUED"]
    try:
        n = 1 << 16
        ck = CommitterKey.generate(ctx, 6, n)
        xy, _ = ck.read()
        const = np.tile(cref.rng_scalars(80, 1)[0], (n, 1))
        uni = cref.rng_scalars(81, n)
        before = ctx.pipeline_stats()
        pts, infs = VariableBaseMSM.multi_scalar_mul_batch_host(ck, [uni, const, uni])
        after = ctx.pipeline_stats()
        assert after["bucket_split"] - before["bucket_split"] == 2 and after["bucket_split_fallbacks"] == before["bucket_split_fallbacks"]
        for j, v in enumerate((uni, const, uni)):
            ref, rinf = cref.msm(c.curve_id, xy, v, threads=8)
            assert bool(infs[j]) == bool(rinf) and np.array_equal(pts[j], ref), j
        ck.free()
    finally:
        ctx.close()


def test_key_memory_report_and_prebuilt_twin(cref):
    """round-3 ADVICE: the 17-bit twin of a 20-bit key is visible (amsm_bases_memory) and can be built ahead of time"""
    from accumulation_amd import CommitterKey, Context, VariableBaseMSM
    c = o.PALLAS
    ctx = Context(c.curve_id)
    try:
        n = 1 << 20
        ck = CommitterKey.generate(ctx, 7, n)
        m = ck.memory()
        assert ck.window_bits == 20 and m["table"] == 13 * n * 64 and m["twin"] == 0 and m["abi_copy"] == 0
        sc = cref.rng_scalars(90, 1 << 19)
        VariableBaseMSM.multi_scalar_mul(ck, sc)            # half the key: still the 20-bit table
        assert ck.memory()["twin"] == 0
        ck.prebuild_twin()
        assert ck.memory()["twin"] == 16 * n * 64
        xy, _ = ck.read(0, 4099)
        got, inf = VariableBaseMSM.multi_scalar_mul(ck, sc[:4099])  # a short range: over the twin
        ref, rinf = cref.msm(c.curve_id, xy, sc[:4099])
        assert bool(inf) == bool(rinf) and np.array_equal(got, ref)
        small = CommitterKey.generate(ctx, 8, 1 << 12)
        small.prebuild_twin()                               # no twin to build: a no-op
        assert small.memory()["twin"] == 0
        small.free()
        ck.free()
    finally:
        ctx.close()
