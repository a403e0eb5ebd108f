"""The reference's accumulation-scheme test template for ASForTrivialPC (src/trivial_pc_as/mod.rs:634-815: degree 11,
no zk, six scenarios from src/lib.rs:263-461) run on the mirror accumulation_amd/trivial_pc_as.py, plus the pieces
against the big-int oracle: the commitment is the MSM of the coefficients, the witness polynomial is the exact
quotient."""
import numpy as np
import pytest

from oracle import pyref as o
from tests import helpers as h
from tests.test_hp_as_scheme_gpu import SchemeRng

pytestmark = pytest.mark.gpu

NUM_ITERATIONS = 3
DEGREE = 11


@pytest.fixture(scope="module")
def env():
    from accumulation_amd import Context, ffi
    from accumulation_amd.trivial_pc_as import TrivialPC
    ctx = Context(ffi.AMSM_PALLAS)
    pp = TrivialPC.setup(ctx, DEGREE)
    yield ctx, pp
    ctx.close()


def generate_inputs(env, ck, num_inputs, rng):
    """src/trivial_pc_as/mod.rs:700-748: random degree-d polynomials, their commitments, a random point each."""
    from accumulation_amd.scalar_field import Fr
    from accumulation_amd.trivial_pc_as import Input, InputInstance, LabeledPolynomial, TrivialPC
    ctx, _ = env
    fr = Fr(ctx.curve)
    out = []
    for _ in range(num_inputs):
        poly = LabeledPolynomial([rng.field() % fr.r for _ in range(TrivialPC.supported_degree(ck) + 1)])
        comm = TrivialPC.commit(ck, poly)
        point = rng.field() % fr.r
        out.append(Input(InputInstance(comm, point, poly.evaluate(fr, point)), poly))
    return out


def run_template(env, num_inputs_per_iteration, num_iterations=NUM_ITERATIONS):
    from accumulation_amd.trivial_pc_as import ASForTrivialPC as AS, TrivialPC
    ctx, pp = env
    ck, _ = TrivialPC.trim(pp, DEGREE)
    pk, vk, dk = AS.index(pp, DEGREE)
    rng = SchemeRng(777)
    inputs = generate_inputs(env, ck, num_iterations * sum(num_inputs_per_iteration), rng)
    start = 0
    for _ in range(num_iterations):
        old = []
        for k in num_inputs_per_iteration:
            step = inputs[start:start + k]
            start += k
            acc, proof = AS.prove(pk, step, old, None, None)
            assert AS.verify(ctx, vk, [i.instance for i in step], [a.instance for a in old], acc.instance, proof,
                             None), "Verify failed"
            old.append(acc)
        assert AS.decide(dk, old[-1], None), "Decide failed"
    return True


class TestASForTrivialPC:
    def test_single_input_init(self, env):
        assert run_template(env, [1])

    def test_multiple_inputs_init(self, env):
        assert run_template(env, [3])

    def test_simple_accumulation(self, env):
        assert run_template(env, [1, 1])

    def test_multiple_inputs_accumulation(self, env):
        assert run_template(env, [1, 1, 2, 3], num_iterations=2)

    def test_accumulators_only(self, env):
        assert run_template(env, [1, 0, 0, 0])

    def test_no_inputs_init(self, env):
        assert run_template(env, [0], num_iterations=1)


def test_simple_accumulation_reference_iteration_count(env):
    """the reference runs every scenario NUM_ITERATIONS = 50 times (src/lib.rs:273); one scenario at that count"""
    assert run_template(env, [1, 1], num_iterations=50)


def test_pieces_vs_oracle(env):
    from accumulation_amd.scalar_field import Fr
    from accumulation_amd.trivial_pc_as import ASForTrivialPC as AS, LabeledPolynomial, TrivialPC, _poly_div_linear
    ctx, pp = env
    c = o.PALLAS
    fr = Fr(ctx.curve)
    ck, _ = TrivialPC.trim(pp, DEGREE)
    xy, inf = ck.read()
    gens = [h.np_to_point(c, xy[i], inf[i]) for i in range(DEGREE + 1)]
    coeffs = [o.rng_scalar(5, i) % c.r for i in range(DEGREE + 1)]
    comm = TrivialPC.commit(ck, LabeledPolynomial(coeffs))
    assert h.np_to_point(c, comm.elem[0], comm.elem[1]) == o.msm_naive(c, gens, coeffs)
    # quotient: p(X) - p(z) == q(X) (X - z)
    z = o.rng_scalar(6, 0) % c.r
    v = sum(cf * pow(z, i, c.r) for i, cf in enumerate(coeffs)) % c.r
    q = _poly_div_linear(fr, coeffs, v, z)
    prod = [0] * (len(q) + 1)
    for i, qi in enumerate(q):
        prod[i] = (prod[i] - z * qi) % c.r
        prod[i + 1] = (prod[i + 1] + qi) % c.r
    assert prod == [(coeffs[0] - v) % c.r] + coeffs[1:]
    # a tampered proof or accumulator must be rejected
    pk, vk, dk = AS.index(pp, DEGREE)
    rng = SchemeRng(1)
    inp = generate_inputs(env, ck, 2, rng)
    acc, proof = AS.prove(pk, inp, [], None, None)
    insts = [i.instance for i in inp]
    assert AS.verify(ctx, vk, insts, [], acc.instance, proof, None)
    bad = [type(p)(p.witness_commitment, (p.witness_eval + 1) % c.r, p.eval) for p in proof]
    assert not AS.verify(ctx, vk, insts, [], acc.instance, bad, None)
    acc.witness.coeffs[0] = (acc.witness.coeffs[0] + 1) % c.r
    assert not AS.decide(dk, acc, None)


def test_config0_size_degree_1023(cref):
    """BASELINE.json config 0 (examples/scaling-as.rs:62-63,91-104): trivial_pc_as on Pallas at degree 2^10 - 1, one input
    accumulated into two old accumulators -- 3 MSMs of <= 1023 points in prove, one of 1024 in decide -- with the
    commitments and the accumulator's commitment checked against the CPU oracle."""
    from accumulation_amd import Context, ffi
    from accumulation_amd.scalar_field import Fr
    from accumulation_amd.trivial_pc_as import ASForTrivialPC as AS, TrivialPC
    d = (1 << 10) - 1
    ctx = Context(ffi.AMSM_PALLAS)
    c = o.PALLAS
    fr = Fr(ctx.curve)
    pp = TrivialPC.setup(ctx, d)
    ck, _ = TrivialPC.trim(pp, d)
    pk, vk, dk = AS.index(pp, d)
    rng = SchemeRng(4242)
    ins = generate_inputs((ctx, pp), ck, 3, rng)
    xy, _ = ck.read()
    for x in ins:  # the input commitments are the MSMs the CPU oracle computes
        ref, rinf = cref.msm(c.curve_id, xy[: d + 1], h.scalars_to_np([v % c.r for v in x.witness.coeffs]), threads=2)
        assert bool(x.instance.commitment.elem[1]) == rinf and np.array_equal(np.asarray(x.instance.commitment.elem[0]), ref)
    acc_a, pa = AS.prove(pk, [ins[0]], [], None, None)
    acc_b, pb = AS.prove(pk, [ins[1]], [], None, None)
    acc, proof = AS.prove(pk, [ins[2]], [acc_a, acc_b], None, None)   # 1 input + 2 old accumulators
    assert AS.verify(ctx, vk, [ins[2].instance], [acc_a.instance, acc_b.instance], acc.instance, proof, None)
    assert AS.decide(dk, acc, None)
    ref, rinf = cref.msm(c.curve_id, xy[: d + 1], h.scalars_to_np([v % c.r for v in acc.witness.coeffs]), threads=2)
    assert bool(acc.instance.commitment.elem[1]) == rinf and np.array_equal(np.asarray(acc.instance.commitment.elem[0]), ref)
    ctx.close()
