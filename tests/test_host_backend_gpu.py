"""Both backends behind the one C ABI on the GPU box: the HIP path (device 0) and the library's host backend (AMSM_DEVICE_HOST,
accumulation_amd/csrc/api_cpu.inc) return the same bytes for the same calls -- MSMs of every call form over both key kinds and
both curves, the vector kernels, key folds and a whole hp_as accumulation.  (Without a GPU the host backend is checked against
the oracles: tests/host_backend/.)"""
import numpy as np
import pytest

from accumulation_amd import CommitterKey, Context, PedersenCommitment, VariableBaseMSM, ffi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[ffi.AMSM_PALLAS, ffi.AMSM_BLS12_381_G1], ids=["pallas", "bls12_381_g1"])
def pair(request):
    g, h = Context(request.param, device=0), Context(request.param, device=ffi.AMSM_DEVICE_HOST)
    yield g, h
    g.close()
    h.close()


def same(a, b):
    return bool(a[1]) == bool(b[1]) and np.array_equal(np.asarray(a[0]), np.asarray(b[0]))


@pytest.mark.parametrize("n", [1, 2, 31, 33, 1000, 1 << 12, (1 << 14) + 5])
def test_msm_forms_agree(pair, n):
    g, h = pair
    for flags in (ffi.AMSM_BASES_PRECOMPUTE, ffi.AMSM_BASES_NO_PRECOMPUTE):
        kg, kh = CommitterKey.generate(g, 4711, n, flags), CommitterKey.generate(h, 4711, n, flags)
        assert np.array_equal(kg.read()[0], kh.read()[0])
        for mont in (False, True):
            vg, vh = g.random_vector(n + 3, n, mont=mont), h.random_vector(n + 3, n, mont=mont)
            assert np.array_equal(vg.download(), vh.download())
            assert same(VariableBaseMSM.multi_scalar_mul(kg, vg, mont=mont), VariableBaseMSM.multi_scalar_mul(kh, vh, mont=mont))
            assert same(VariableBaseMSM.multi_scalar_mul(kg, vg.download(), mont=mont), VariableBaseMSM.multi_scalar_mul(kh, vh.download(), mont=mont))
            if n >= 4:
                a = VariableBaseMSM.multi_scalar_mul_grouped(kg, vg, 1, mont=mont)
                b = VariableBaseMSM.multi_scalar_mul_grouped(kh, vh, 1, mont=mont)
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
                a = VariableBaseMSM.multi_scalar_mul_multi(kg, [(0, vg), (n // 3, vg)], mont=mont)
                b = VariableBaseMSM.multi_scalar_mul_multi(kh, [(0, vh), (n // 3, vh)], mont=mont)
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        kg.free()
        kh.free()


def test_scalar_range_error_is_the_same(pair):
    g, h = pair
    sc = np.zeros((40, 4), dtype=np.uint64)
    sc[7, 3] = 0xF << 60  # (far above 2^255: the direct sum takes scalars a little beyond it, include/amsm.h)
    for ctx in (g, h):
        ck = CommitterKey.generate(ctx, 1, 40)
        with pytest.raises(ffi.AmsmError) as e:
            VariableBaseMSM.multi_scalar_mul(ck, sc)
        assert e.value.status == ffi.AMSM_E_SCALAR_RANGE
        ck.free()


def test_vector_kernels_and_fold_agree(pair):
    from accumulation_amd.hp_as import combine_vectors, compute_hp, compute_t_vecs
    from accumulation_amd.scalar_field import Fr
    g, h = pair
    n = 3000
    outs = []
    for ctx in (g, h):
        fr = Fr(ctx.curve)
        a = [ctx.random_vector(10 + j, n - 7 * j, mont=True) for j in range(3)]
        b = [ctx.random_vector(20 + j, n - 5 * j, mont=True) for j in range(3)]
        mu = np.stack([fr.to_limbs(x) for x in (1, 0x1234567, (1 << 127) + 3, 99)])
        res = [compute_hp(ctx, a[0], b[0]).download(), combine_vectors(ctx, a, mu[:3], b[1]).download()]
        res += [t.download() for t in compute_t_vecs(ctx, a, b, mu, n, (a[2], b[2]))]
        ck = CommitterKey.generate(ctx, 3, 64, ffi.AMSM_BASES_NO_PRECOMPUTE)
        f = ck.fold(32, fr.to_limbs(0xDEADBEEF12345), 128)
        res.append(f.read()[0])
        outs.append(res)
    for x, y in zip(*outs):
        assert np.array_equal(x, y)


def test_hp_as_accumulation_agrees(pair):
    from accumulation_amd.hp_as import ASForHadamardProducts as AS
    from tests.test_hp_as_scheme_gpu import SchemeRng, generate_inputs
    g, h = pair
    if g.curve != ffi.AMSM_PALLAS:
        pytest.skip("the scheme mirrors' test inputs are Pallas")
    accs = []
    for ctx in (g, h):
        ck = PedersenCommitment.setup(ctx, 300, seed=4242)
        pk, vk, dk = AS.index(ck)
        inputs = generate_inputs(ctx, ck, 3, True)
        rng = SchemeRng(7)
        acc1, _ = AS.prove(pk, inputs[:1], [], rng, None)
        acc2, proof = AS.prove(pk, inputs[1:], [acc1], rng, None)
        assert AS.decide(dk, acc2, None)
        accs.append((acc2, proof))
    (a, pa), (b, pb) = accs
    for x, y in ((a.instance.comm_1, b.instance.comm_1), (a.instance.comm_2, b.instance.comm_2), (a.instance.comm_3, b.instance.comm_3)):
        assert same(x, y)
    assert np.array_equal(a.witness.a_vec.download(), b.witness.a_vec.download())
    assert np.array_equal(a.witness.b_vec.download(), b.witness.b_vec.download())
    for x, y in zip(pa.product_poly_comm.low + pa.product_poly_comm.high, pb.product_poly_comm.low + pb.product_poly_comm.high):
        assert same(x, y)
