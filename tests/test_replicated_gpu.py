"""Replicated keys of a multi-device context (amsm.h AMSM_BASES_REPLICATE, round 6): every device holds the WHOLE key and the batch
entry points deal their independent MSMs round-robin to the devices -- MSM v on device v mod n_dev, no partial sums, no
exchange, results in call order (the commit rounds of `r1cs_nark_as::prove`: src/r1cs_nark_as/r1cs_nark/mod.rs:216-218,234-236,
251,261; src/r1cs_nark_as/mod.rs:394-410).  All "devices" are GPU 0 here; results must equal the single-device ones bit for bit."""
import numpy as np
import pytest

from oracle import pyref as o

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("c", [o.PALLAS, o.BLS12_381_G1], ids=lambda c: c.name)
def test_batches_dealt_over_three_devices(cref, c):
    from accumulation_amd import CommitterKey, Context, MultiContext, PedersenCommitment, VariableBaseMSM, ffi
    n = 5000
    one, multi = Context(c.curve_id), MultiContext(c.curve_id, [0, 0, 0])
    try:
        ck1 = CommitterKey.generate(one, 0x4E9, n)
        ckr = CommitterKey.generate(multi, 0x4E9, n, ffi.AMSM_BASES_REPLICATE)
        lib = multi._lib
        assert lib.amsm_bases_replicas(ckr._h) == 3 and lib.amsm_bases_num_shards(ckr._h) == 1 and lib.amsm_bases_replicas(ck1._h) == 1
        assert ckr.memory()["table"] == 3 * ck1.memory()["table"]  # three full copies, reported
        xy, _ = ck1.read()
        assert np.array_equal(ckr.read()[0], xy)
        hv = [cref.rng_frs(c.curve_id, 0x4EA + j, n - 11 * j) for j in range(7)]
        refs = [cref.msm(c.curve_id, xy[: v.shape[0]], v, threads=4) for v in hv]
        # host slices (amsm_pedersen_commit_batch: ragged lengths, hiding terms) and amsm_msm_batch (equal lengths)
        elems = [cref.fr_to_mont(c.curve_id, v) for v in hv]
        pk1 = PedersenCommitment.setup(one, n, seed=0x4EB)
        pkr = PedersenCommitment.setup(multi, n, seed=0x4EB, flags=ffi.AMSM_BASES_REPLICATE)
        assert lib.amsm_bases_replicas(pkr._h) == 3
        rnd = [None, cref.fr_to_mont(c.curve_id, cref.rng_frs(c.curve_id, 50, 1))[0], None, cref.fr_to_mont(c.curve_id, cref.rng_frs(c.curve_id, 51, 1))[0],
               None, None, cref.fr_to_mont(c.curve_id, cref.rng_frs(c.curve_id, 52, 1))[0]]
        p1 = PedersenCommitment.commit_batch_host(pk1, elems, rnd)
        pr = PedersenCommitment.commit_batch_host(pkr, elems, rnd)
        assert all(np.array_equal(a[0], b[0]) and a[1] == b[1] for a, b in zip(p1, pr))
        pk1.free()
        pkr.free()
        before = multi.replicated_msms
        eq = [v[: n - 66] for v in hv]
        got, ginf = VariableBaseMSM.multi_scalar_mul_batch_host(ckr, eq)
        want, winf = VariableBaseMSM.multi_scalar_mul_batch_host(ck1, eq)
        assert np.array_equal(got, want) and np.array_equal(ginf, winf)
        assert multi.replicated_msms - before == 7 - 3  # MSMs 0, 3, 6 stay on the primary
        for k, v in enumerate(eq[:2]):
            r, rinf = cref.msm(c.curve_id, xy[: v.shape[0]], v, threads=4)
            assert np.array_equal(got[k], r) and bool(ginf[k]) == bool(rinf)
        # device vectors on the primary (amsm_msm_batch_device: the form the scheme drivers use)
        dv_r = [multi.upload(cref.fr_to_mont(c.curve_id, v[: n - 66])) for v in hv[:5]]
        dv_1 = [one.upload(cref.fr_to_mont(c.curve_id, v[: n - 66])) for v in hv[:5]]
        got, ginf = VariableBaseMSM.multi_scalar_mul_batch(ckr, dv_r, mont=True)
        want, winf = VariableBaseMSM.multi_scalar_mul_batch(ck1, dv_1, mont=True)
        assert np.array_equal(got, want) and np.array_equal(ginf, winf)
        # a single MSM, a grouped MSM and a base offset use the primary's copy like any single-device key
        a, ai = VariableBaseMSM.multi_scalar_mul(ckr, dv_r[0], mont=True)
        assert np.array_equal(a, want[0]) and bool(ai) == bool(winf[0])
        a, ai = VariableBaseMSM.multi_scalar_mul(ckr, hv[3][:100], base_off=n - 100)
        b, bi = VariableBaseMSM.multi_scalar_mul(ck1, hv[3][:100], base_off=n - 100)
        assert np.array_equal(a, b) and ai == bi
        assert multi._lib.amsm_ctx_collectives(multi._h) == 0  # no exchange of partial records ever happened
        # per-shard slices make no sense for a key without shards: refused, not misread
        import ctypes as C
        ptrs = (C.c_void_p * 6)(*([dv_r[0].ptr.value] * 6))
        out6, inf6 = np.zeros((2, 2 * multi.fq_limbs), dtype=np.uint64), np.zeros((2,), dtype=np.uint8)
        assert lib.amsm_msm_batch_sharded_device(multi._h, ckr._h, ptrs, 2, 1, out6.ctypes.data_as(C.c_void_p),
                                                 inf6.ctypes.data_as(C.c_void_p)) == ffi.AMSM_E_INVALID_ARG
        ckr.free()
        ck1.free()
    finally:
        multi.close()
        one.close()


def test_replicate_below_threshold_and_sharded_above(cref):
    from accumulation_amd import CommitterKey, MultiContext
    c = o.PALLAS
    multi = MultiContext(c.curve_id, [0, 0])
    try:
        multi.set_replicate_below(1 << 10)
        small = CommitterKey.generate(multi, 7, 1 << 10)
        big = CommitterKey.generate(multi, 7, (1 << 10) + 1)
        lib = multi._lib
        assert lib.amsm_bases_replicas(small._h) == 2 and lib.amsm_bases_num_shards(small._h) == 1
        assert lib.amsm_bases_replicas(big._h) == 1 and lib.amsm_bases_num_shards(big._h) == 2
        xy, _ = big.read()
        v = [cref.rng_frs(c.curve_id, 90 + j, 1 << 10) for j in range(4)]
        from accumulation_amd import VariableBaseMSM
        got, ginf = VariableBaseMSM.multi_scalar_mul_batch_host(small, v)
        for k in range(4):
            r, rinf = cref.msm(c.curve_id, xy[: 1 << 10], v[k], threads=4)
            assert np.array_equal(got[k], r) and bool(ginf[k]) == bool(rinf)
        small.free()
        big.free()
    finally:
        multi.close()
