"""The reference's accumulation-scheme test template (src/lib.rs:334-395), six scenarios (:398-459), for
ASForR1CSNark exactly as instantiated at src/r1cs_nark_as/mod.rs:1190-1395: DummyCircuit with num_inputs 5,
num_constraints 10, zk and no-zk; every input is a fresh NARK proof; prove -> verify per step, decide per
iteration.  All SpMVs / vector loops / commitments run on the GPU through the C ABI."""
import pytest

from tests.test_hp_as_scheme_gpu import SchemeRng
from tests.test_r1cs_nark_gpu import dummy_circuit

pytestmark = pytest.mark.gpu

NUM_ITERATIONS = 4  # reference: 50 (src/lib.rs:273)
NUM_INPUTS, NUM_CONSTRAINTS = 5, 10  # src/r1cs_nark_as/mod.rs:1289 ff.


@pytest.fixture(scope="module")
def env():
    from accumulation_amd import Context, ffi
    from accumulation_amd import r1cs_nark as nark
    from accumulation_amd.scalar_field import MODULI
    ctx = Context(ffi.AMSM_PALLAS)
    r = MODULI[ctx.curve]
    A, B, C_, _, _ = dummy_circuit(NUM_INPUTS, NUM_CONSTRAINTS, 2, 3, r)
    ipk = nark.index(ctx, A, B, C_, NUM_INPUTS + 1, NUM_INPUTS + 3, key_seed=31337)
    yield ctx, ipk, r
    ctx.close()


def generate_inputs(env, num_inputs, make_zk, rng):
    """src/r1cs_nark_as/mod.rs:1229-1276"""
    from accumulation_amd import r1cs_nark as nark
    from accumulation_amd.r1cs_nark_as import ASForR1CSNark as AS, Input, InputInstance
    from accumulation_amd.scalar_field import Fr
    from accumulation_amd.sponge import Sha256Sponge
    ctx, ipk, r = env
    fr = Fr(ctx.curve)
    out = []
    for _ in range(num_inputs):
        a, b = rng.field() % r, rng.field() % r
        _, _, _, inst, w = dummy_circuit(NUM_INPUTS, NUM_CONSTRAINTS, a, b, r)
        nark_sponge, _, _ = AS._sponges(Sha256Sponge())
        proof = nark.prove(ipk, inst, ctx.upload(fr.to_limbs_many(w)), make_zk, nark_sponge, rng if make_zk else None)
        out.append(Input(InputInstance(inst, proof.first_msg), proof.second_msg))
    return out


def run_template(env, num_inputs_per_iteration, make_zk, num_iterations=NUM_ITERATIONS):
    from accumulation_amd.r1cs_nark_as import ASForR1CSNark as AS
    ctx, ipk, r = env
    pk, vk, dk = AS.index(ipk)
    rng = SchemeRng(2024)
    total = num_iterations * sum(num_inputs_per_iteration)
    inputs = generate_inputs(env, total, make_zk, rng)
    start = 0
    for _ in range(num_iterations):
        old = []
        for k in num_inputs_per_iteration:
            step = inputs[start:start + k]
            start += k
            acc, proof = AS.prove(pk, step, old, rng if make_zk else None, None)
            assert AS.verify(ctx, vk, [x.instance for x in step], [x.instance for x in old], acc.instance, proof, None), \
                "Verify failed"
            old.append(acc)
        assert AS.decide(dk, old[-1], None), "Decide failed"
    return True


@pytest.mark.parametrize("make_zk", [False, True], ids=["no_zk", "zk"])
class TestASForR1CSNark:
    def test_single_input_init(self, env, make_zk):
        assert run_template(env, [1], make_zk)

    def test_multiple_inputs_init(self, env, make_zk):
        assert run_template(env, [3], make_zk)

    def test_simple_accumulation(self, env, make_zk):
        assert run_template(env, [1, 1], make_zk)

    def test_multiple_inputs_accumulation(self, env, make_zk):
        assert run_template(env, [1, 1, 2, 3], make_zk)

    def test_accumulators_only(self, env, make_zk):
        assert run_template(env, [1, 0, 0, 0], make_zk)

    def test_no_inputs_init(self, env, make_zk):
        assert run_template(env, [0], make_zk, num_iterations=1)


@pytest.mark.parametrize("make_zk", [False, True], ids=["no_zk", "zk"])
def test_simple_accumulation_reference_iteration_count(env, make_zk):
    """the reference runs every scenario NUM_ITERATIONS = 50 times (src/lib.rs:273); one scenario at that count"""
    assert run_template(env, [1, 1], make_zk, num_iterations=50)


def test_tampered_accumulator_rejected(env):
    from accumulation_amd.r1cs_nark_as import ASForR1CSNark as AS, Accumulator, AccumulatorInstance
    ctx, ipk, r = env
    pk, vk, dk = AS.index(ipk)
    rng = SchemeRng(5)
    inputs = generate_inputs(env, 2, False, rng)
    acc, proof = AS.prove(pk, inputs, [], None, None)
    assert AS.verify(ctx, vk, [x.instance for x in inputs], [], acc.instance, proof, None)
    assert AS.decide(dk, acc, None)
    i = acc.instance
    bad_input = list(i.r1cs_input)
    bad_input[1] = (bad_input[1] + 1) % r
    bad = AccumulatorInstance(bad_input, i.comm_a, i.comm_b, i.comm_c, i.hp_instance)
    assert not AS.verify(ctx, vk, [x.instance for x in inputs], [], bad, proof, None)
    assert not AS.decide(dk, Accumulator(bad, acc.witness), None)
    swapped = AccumulatorInstance(i.r1cs_input, i.comm_b, i.comm_a, i.comm_c, i.hp_instance)
    assert not AS.decide(dk, Accumulator(swapped, acc.witness), None)
