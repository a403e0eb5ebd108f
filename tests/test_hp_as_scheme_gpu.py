"""The reference's own accumulation-scheme test template (src/lib.rs:334-395) and its six scenarios
(:398-459), instantiated for ASForHadamardProducts exactly like src/hp_as/mod.rs:957-1151 (vector_len 11,
zk and no-zk), but running on the GPU path through the C ABI: prove -> verify after every step, decide on the
last accumulator of every iteration.  Inputs are generated like the reference does: a_vec / b_vec are
`vec![rand; len]` CONSTANT vectors (src/hp_as/mod.rs:991-992)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NUM_ITERATIONS = 8  # reference: 50 (src/lib.rs:273); same logic, fewer repetitions to keep the GPU suite short
VECTOR_LEN = 11     # src/hp_as/mod.rs:1057 ff.


class SchemeRng:
    """Deterministic stand-in for ark_std::test_rng(): .field() -> scalar."""

    def __init__(self, seed):
        self.seed, self.i = seed, 0

    def field(self):
        from accumulation_amd.scalar_field import MODULI
        # splitmix-style stream, 254 bits
        x = 0
        for k in range(4):
            z = (self.seed * 0xD1342543DE82EF95 + (4 * self.i + k) * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019) & (2**64 - 1)
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
            z ^= z >> 31
            x |= z << (64 * k)
        self.i += 1
        return x & ((1 << 254) - 1)


@pytest.fixture(scope="module")
def env():
    from accumulation_amd import Context, PedersenCommitment, ffi
    ctx = Context(ffi.AMSM_PALLAS)
    ck = PedersenCommitment.setup(ctx, VECTOR_LEN, seed=4242)
    yield ctx, ck
    ctx.close()


def generate_inputs(ctx, ck, num_inputs, make_zk):
    """src/hp_as/mod.rs:980-1044"""
    from accumulation_amd import PedersenCommitment
    from accumulation_amd.hp_as import Accumulator, InputInstance, InputWitness, InputWitnessRandomness, compute_hp
    from accumulation_amd.scalar_field import Fr
    fr = Fr(ctx.curve)
    rng = SchemeRng(0xC0FFEE)  # the reference re-seeds test_rng() here (Appendix E.2 of SURVEY.md)
    out = []
    for _ in range(num_inputs):
        a = ctx.fill(fr.to_limbs(rng.field()), VECTOR_LEN)
        b = ctx.fill(fr.to_limbs(rng.field()), VECTOR_LEN)
        prod = compute_hp(ctx, a, b)
        rnd = InputWitnessRandomness(rng.field(), rng.field(), rng.field()) if make_zk else None
        lim = (lambda v: fr.to_limbs(v)) if make_zk else (lambda v: None)
        c1 = PedersenCommitment.commit(ck, a, lim(rnd.rand_1) if rnd else None)
        c2 = PedersenCommitment.commit(ck, b, lim(rnd.rand_2) if rnd else None)
        c3 = PedersenCommitment.commit(ck, prod, lim(rnd.rand_3) if rnd else None)
        out.append(Accumulator(InputInstance(c1, c2, c3), InputWitness(a, b, rnd)))
    return out


def run_template(env, num_inputs_per_iteration, make_zk, num_iterations=NUM_ITERATIONS):
    """src/lib.rs:334-395"""
    from accumulation_amd.hp_as import ASForHadamardProducts as AS
    ctx, ck = env
    pk, vk, dk = AS.index(ck)
    total = num_iterations * sum(num_inputs_per_iteration)
    inputs = generate_inputs(ctx, ck, total, make_zk)
    assert len(inputs) == total
    rng = SchemeRng(7) if make_zk else None
    start = 0
    for _ in range(num_iterations):
        old = []
        for k in num_inputs_per_iteration:
            step_inputs = inputs[start:start + k]
            start += k
            acc, proof = AS.prove(pk, step_inputs, old, rng, None)
            ok = AS.verify(ctx, vk, [x.instance for x in step_inputs], [x.instance for x in old], acc.instance, proof, None)
            assert ok, "Verify failed"
            old.append(acc)
        assert old
        assert AS.decide(dk, old[-1], None), "Decide failed"
    return True


@pytest.mark.parametrize("make_zk", [False, True], ids=["no_zk", "zk"])
class TestASForHP:
    def test_single_input_init(self, env, make_zk):          # src/lib.rs:398-405
        assert run_template(env, [1], make_zk)

    def test_multiple_inputs_init(self, env, make_zk):       # :408-415
        assert run_template(env, [3], make_zk)

    def test_simple_accumulation(self, env, make_zk):        # :418-425
        assert run_template(env, [1, 1], make_zk)

    def test_multiple_inputs_accumulation(self, env, make_zk):  # :428-435
        assert run_template(env, [1, 1, 2, 3], make_zk)

    def test_accumulators_only(self, env, make_zk):          # :438-445
        assert run_template(env, [1, 0, 0, 0], make_zk)

    def test_no_inputs_init(self, env, make_zk):             # :448-459 (one iteration)
        assert run_template(env, [0], make_zk, num_iterations=1)


@pytest.mark.parametrize("make_zk", [False, True], ids=["no_zk", "zk"])
def test_simple_accumulation_reference_iteration_count(env, make_zk):
    """the reference runs every scenario NUM_ITERATIONS = 50 times (src/lib.rs:273); one scenario at that count"""
    assert run_template(env, [1, 1], make_zk, num_iterations=50)


def test_error_behaviour(env):
    """prove() error variants of src/hp_as/mod.rs:110-157, 664-673."""
    from accumulation_amd.hp_as import (ASForHadamardProducts as AS, Accumulator, InputInstance, InputWitness,
                                        InputWitnessRandomness, MalformedInput, MissingRng)
    ctx, ck = env
    good = generate_inputs(ctx, ck, 1, False)[0]
    z = np.zeros(4, dtype=np.uint64)
    short = Accumulator(InputInstance.zero(ctx), InputWitness(ctx.fill(z, 5), ctx.fill(z, 5), None))
    with pytest.raises(MalformedInput):
        AS.prove(ck, [good, short], [], None, None)  # unequal lengths
    too_long = Accumulator(InputInstance.zero(ctx), InputWitness(ctx.fill(z, 12), ctx.fill(z, 12), None))
    with pytest.raises(MalformedInput):
        AS.prove(ck, [too_long], [], None, None)  # exceeds the key
    hiding = Accumulator(good.instance, InputWitness(good.witness.a_vec, good.witness.b_vec, InputWitnessRandomness(1, 2, 3)))
    with pytest.raises(MissingRng):
        AS.prove(ck, [hiding], [], None, None)
    # a tampered proof must not verify
    acc, proof = AS.prove(ck, [good], [], None, None)
    assert AS.verify(ctx, 11, [good.instance], [], acc.instance, proof, None)
    bad = Accumulator(InputInstance(acc.instance.comm_2, acc.instance.comm_1, acc.instance.comm_3), acc.witness)
    assert not AS.verify(ctx, 11, [good.instance], [], bad.instance, proof, None)
    assert not AS.decide(ck, bad, None)
