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
    try:
        ctx = Context(c.curve_id)
    finally:
        del os.environ["AMSM_BPL_PROBE"]
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
