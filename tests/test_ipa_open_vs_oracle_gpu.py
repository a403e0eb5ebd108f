"""A WHOLE IPA opening at the config sizes against the oracle, every proof element bit for bit (round-3 verdict: the last
hot-path piece whose config-size output no oracle had seen): `InnerProductArgPC::open` (ark-poly-commit ipa_pc, ext; called
at src/ipa_pc_as/mod.rs:454) and `check` (:836) at d + 1 = 2^16 on Pallas (BASELINE config 2) and d + 1 = 2^20 on BLS12-381
G1 (config 3), with and without hiding, with the physical-fold threshold on both sides of its default.

The product runs the rounds its own way -- `amsm_ipa_round_fused`: challenge-product scalars over the ORIGINAL key, a grouped
MSM on the bucket-split / bucket-per-lane pipelines, a few physical key folds through the joint table ladder and the Jacobian
ladders for large keys, the final key as one MSM of the check polynomial's coefficients.  The oracle (oracle/pyref_as.py
`ipa_open` on the array backend oracle/fastref.py over oracle/ark_msm.c) runs the DEFINITION: per round two plain MSMs over
the current key halves and `key_l += x key_r` for every generator, every round.  The Fiat-Shamir challenges and the prover's
random draws are recorded from the product and injected (the sponge is host hashing outside the accelerated path)."""
import os

import numpy as np
import pytest

from oracle import pyref as o
from oracle import pyref_as as oa
from tests import helpers as h
from tests.test_r1cs_nark_gpu import RecordingRng

pytestmark = pytest.mark.gpu


def _open_both_ways(curve, log_n, hiding, fold_above, oracle_threads=None, probe_jump=False):
    from accumulation_amd import Context
    from accumulation_amd.ipa_pc import InnerProductArgPC as IpaPC
    from accumulation_amd.scalar_field import Fr
    from oracle import fastref
    c = curve
    n = 1 << log_n
    if fold_above is not None:
        os.environ["AMSM_IPA_FOLD_ABOVE"] = str(fold_above)
    ctx = Context(c.curve_id)
    try:
        fr = Fr(ctx.curve)
        pp = IpaPC.setup(ctx, n - 1, seed=0x1BA0000 + log_n)
        ck, vk = IpaPC.trim(pp, n - 1)
        assert ck.supported_degree() == n - 1
        poly = ctx.random_vector(0x1BA1000 + log_n, n - 5, mont=True)  # degree d - 5: the padding to d + 1 is exercised
        point = o.rng_scalar(0x1BA2000, log_n) % c.r
        rng = RecordingRng(0x1BA3000 + log_n) if hiding else None
        comm, rand = IpaPC.commit(ck, poly, hiding, rng)
        n_commit_draws = len(rng.draws) if hiding else 0
        log = []
        orig = IpaPC._challenge.__func__

        def recording(cls, fr_, parts):
            v = orig(cls, fr_, parts)
            log.append(v)
            return v
        IpaPC._challenge = classmethod(recording)
        before = ctx.pipeline_stats()
        try:
            proof = IpaPC.open(ck, poly, comm, point, rand, hiding, rng)
        finally:
            IpaPC._challenge = classmethod(orig)
        stats = {k: v - before[k] for k, v in ctx.pipeline_stats().items()}
        challenges = list(log)
        assert len(challenges) == (1 if hiding else 0) + 1 + log_n
        accepted = IpaPC.check(vk, comm, point, proof_value(ctx, fr, poly, point, IpaPC), proof)
        # ---- the oracle's opening ----
        xy, _ = ck.comm_key.read()
        ops = fastref.NumpyOps(c, threads=oracle_threads)
        pt = lambda p: h.np_to_point(c, p[0], p[1])
        hid = None
        if hiding:
            draws = rng.draws[n_commit_draws:]
            assert len(draws) == n + 1
            hid = {"polynomial": ops.mont(draws[:n]), "rand": draws[n], "poly_rand": rand}
        with oa.use_ops(ops):
            ref = oa.ipa_open(c, xy, pt(ck.h), pt(ck.s), poly.download(), pt(comm.comm), point, challenges, hid)
            ref_ok = oa.ipa_check(c, xy, pt(ck.h), pt(ck.s), pt(comm.comm), point, ref["combined_v"], ref, challenges)
        got = {"l_vec": [pt(p) for p in proof.l_vec], "r_vec": [pt(p) for p in proof.r_vec], "final_comm_key": pt(proof.final_comm_key),
               "c": proof.c % c.r, "hiding_comm": None if proof.hiding_comm is None else pt(proof.hiding_comm),
               "rand": None if proof.rand is None else proof.rand % c.r}
        if probe_jump:  # does this key qualify for the jump fold (then the opening above took it: ipa_pc.py open)?
            from tests.test_ipa_jump_gpu import _jump
            stats["jump_rc"] = _jump(ctx, ck.comm_key, log_n, [3 + 2 * r for r in range(log_n - 6)], fr)[0]
        return got, ref, accepted, ref_ok, stats
    finally:
        ctx.close()
        os.environ.pop("AMSM_IPA_FOLD_ABOVE", None)


def proof_value(ctx, fr, poly, point, IpaPC):
    """p(point), by the product's own kernels (the oracle's `combined_v` is compared with it through `check`)"""
    from accumulation_amd import ffi
    from accumulation_amd.engine import _ptr
    z = ctx.vector(poly.n)
    ffi.check(ctx._lib.amsm_vec_powers(ctx._h, _ptr(fr.to_limbs(point)), poly.n, z.ptr), "amsm_vec_powers")
    return IpaPC._inner_product(ctx, fr, poly, z)


def _assert_equal(got, ref, accepted, ref_ok):
    assert accepted, "the product's own check rejects its opening"
    assert ref_ok, "the oracle's check rejects the oracle's opening"
    for j, (a, b) in enumerate(zip(got["l_vec"], ref["l_vec"])):
        assert a == b, f"L_{j}"
    for j, (a, b) in enumerate(zip(got["r_vec"], ref["r_vec"])):
        assert a == b, f"R_{j}"
    assert len(got["l_vec"]) == len(ref["l_vec"]) == len(got["r_vec"]) == len(ref["r_vec"])
    assert got["final_comm_key"] == ref["final_comm_key"], "final key"
    assert got["c"] == ref["c"], "final coefficient"
    assert got["hiding_comm"] == ref["hiding_comm"] and got["rand"] == ref["rand"], "hiding terms"


@pytest.mark.parametrize("hiding", [False, True], ids=["no_zk", "zk"])
@pytest.mark.parametrize("fold_above", [None, 99, 13], ids=["default_schedule", "never_fold_the_key", "fold_down_to_2p13"])
def test_pallas_2p16_opening(hiding, fold_above):
    """BASELINE config 2.  Default at this size: no physical fold (every round a grouped MSM over the original key, the
    bucket-split pipeline); 13: three physical folds first (table ladder over the key's window multiples, then plain)"""
    got, ref, accepted, ref_ok, stats = _open_both_ways(o.PALLAS, 16, hiding, fold_above)
    _assert_equal(got, ref, accepted, ref_ok)


@pytest.mark.parametrize("hiding,fold_above,log_n", [(False, None, 20), (True, 16, 18)], ids=["no_zk_default_schedule", "zk_2p18_fold_down_to_2p16"])
def test_bls12_381_2p20_opening(hiding, fold_above, log_n):
    """BASELINE config 3 (the 384-bit field path).  Default: five physical folds (the first through the 20-bit key's window
    multiples, the top four of which sit off the 20-bit grid -- MsmGeom::n_narrow), then 15 rounds over the 2^15-point key;
    the zk run (at 2^18: the suite's time budget): two folds, then 16 rounds of grouped MSMs over a plain 2^16-point key (8-bit
    windows, the chunked pipeline with 128-bucket sets)"""
    got, ref, accepted, ref_ok, stats = _open_both_ways(o.BLS12_381_G1, log_n, hiding, fold_above)
    _assert_equal(got, ref, accepted, ref_ok)
    assert stats["fallbacks"] == 0


@pytest.mark.parametrize("hiding", [False, True], ids=["no_zk", "zk"])
def test_bls12_381_2p16_opening_with_the_jump_fold(hiding):
    """BLS12-381 at d + 1 = 2^16 with the key never folded physically: ten rounds of grouped MSMs over the 16-bit table, the jump fold
    to 64 generators (round 6: amsm_ipa_jump_fold over the 384-bit field) and six rounds on the host -- the one schedule of this
    curve that takes the jump (the default folds once, and a folded key is plain)"""
    got, ref, accepted, ref_ok, stats = _open_both_ways(o.BLS12_381_G1, 16, hiding, 99, probe_jump=True)
    _assert_equal(got, ref, accepted, ref_ok)
    assert stats["fallbacks"] == 0 and stats["jump_rc"] == 0


def test_pallas_2p20_opening_grouped_msms_stay_on_the_bucket_per_lane_pipeline():
    """never fold: twenty rounds, each ONE grouped MSM of 2^20 pairs over the 20-bit key (two bucket sets on the
    bucket-per-lane pipeline -- round 3 sent these to the 17-bit twin and the chunked pipeline)"""
    got, ref, accepted, ref_ok, stats = _open_both_ways(o.PALLAS, 20, False, 99)
    _assert_equal(got, ref, accepted, ref_ok)
    assert stats["bucket_per_lane"] >= 20 and stats["fallbacks"] == 0
