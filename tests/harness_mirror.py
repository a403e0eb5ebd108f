"""tools/profile_as.cpp's workloads (the reference's harness, examples/scaling-as.rs:38-138) rebuilt on the PYTHON mirrors: the same
keys, the same synthetic vectors, the same `HarnessRng` stream drawn in the same order, the same sponge -- so that the serialised
new accumulator and proof of `profile_as --dump` (the C++ drivers, the path bench.py times for accumulations/sec) can be compared
byte for byte with the mirrors', which tests/test_config_size_gpu.py / test_ipa_open_vs_oracle_gpu.py hold against the oracle at
the same sizes.  Each function returns (accumulator_bytes, proof_bytes).  Test infrastructure."""
from accumulation_amd import PedersenCommitment, ffi
from accumulation_amd.scalar_field import Fr
from tests.ser_mirror import Ser
from tests.test_hp_as_scheme_gpu import SchemeRng as HarnessRng  # the same splitmix stream as tools/profile_as.cpp's HarnessRng


def make_sponge(name, curve):
    from accumulation_amd.sponge import PoseidonSponge, Sha256Sponge
    return PoseidonSponge(curve) if name == "poseidon" else Sha256Sponge()


def hp_as(ctx, lg, harness_shape, sponge="poseidon", seed=0, constant=False):
    """profile_hp: vector length 2^lg; harness shape = 1 input + the same accumulator twice, zk; n2 = 1 input + 1 accumulator"""
    from accumulation_amd.hp_as import ASForHadamardProducts as AS, Accumulator, InputInstance, InputWitness, InputWitnessRandomness, compute_hp
    fr = Fr(ctx.curve)
    n = 1 << lg
    hr = HarnessRng(0xA11CE ^ seed)
    ck = PedersenCommitment.setup(ctx, n, seed=0x5EED1001)
    pk, vk, dk = AS.index(ck)
    if constant:
        a = ctx.fill(fr.to_limbs(hr.field() % fr.r), n)
        b = ctx.fill(fr.to_limbs(hr.field() % fr.r), n)
    else:
        a, b = ctx.random_vector(100 + seed, n, mont=True), ctx.random_vector(101 + seed, n, mont=True)
    prod = compute_hp(ctx, a, b)
    rnd = InputWitnessRandomness(hr.field() % fr.r, hr.field() % fr.r, hr.field() % fr.r) if harness_shape else None
    commit = PedersenCommitment.commit
    c = [commit(ck, v, None if rnd is None else fr.to_limbs(r)) for v, r in ((a, rnd and rnd.rand_1), (b, rnd and rnd.rand_2), (prod, rnd and rnd.rand_3))]
    inputs = [Accumulator(InputInstance(*c), InputWitness(a, b, rnd))]
    rng = hr if harness_shape else None
    first, _ = AS.prove(pk, inputs, [], rng, make_sponge(sponge, ctx.curve))
    old = [first, first] if harness_shape else [first]
    acc, proof = AS.prove(pk, inputs, old, rng, make_sponge(sponge, ctx.curve))
    assert AS.decide(dk, acc, None)
    s = Ser(ctx)
    return s.hp_accumulator(acc), s.hp_proof(proof)


def r1cs_nark_as(ctx, lg, harness_shape, sponge="poseidon", seed=0):
    """profile_nark_as: the reference's DummyCircuit with 2^lg constraints, 5 public inputs"""
    from accumulation_amd import r1cs_nark as nark
    from accumulation_amd.r1cs_nark_as import ASForR1CSNark as AS, Input, InputInstance
    fr = Fr(ctx.curve)
    n_con, n_inputs = 1 << lg, 5
    n_inst = n_inputs + 1
    hr = HarnessRng(0xB0B ^ seed)
    A = [[(1, n_inst)] for _ in range(n_con - 1)] + [[]]
    B = [[(1, n_inst + 1)] for _ in range(n_con - 1)] + [[]]
    Cm = [[(1, 1)] for _ in range(n_con - 1)] + [[]]
    ipk = nark.index(ctx, A, B, Cm, n_inst, n_inst + 2, key_seed=31337)
    pk, vk, dk = AS.index(ipk)
    rng = hr if harness_shape else None
    a, b = hr.field() % fr.r, hr.field() % fr.r
    inst = [1, a * b % fr.r] + [a] * (n_inputs - 1)
    nark_sponge, _, _ = AS._sponges(make_sponge(sponge, ctx.curve))
    proof = nark.prove(ipk, inst, ctx.upload(fr.to_limbs_many([a, b])), harness_shape, nark_sponge, rng)
    inputs = [Input(InputInstance(inst, proof.first_msg), proof.second_msg)]
    first, _ = AS.prove(pk, inputs, [], rng, make_sponge(sponge, ctx.curve))
    old = [first, first] if harness_shape else [first]
    acc, pr = AS.prove(pk, inputs, old, rng, make_sponge(sponge, ctx.curve))
    assert AS.decide(dk, acc, None)
    s = Ser(ctx)
    return s.nark_as_accumulator(acc), s.nark_as_proof(pr)


def ipa_pc_as(ctx, lg, harness_shape, sponge="poseidon", seed=0):
    """profile_ipa: degree 2^lg - 1 (dl_param_gen / dl_input_gen, examples/scaling-as.rs:199-280)"""
    import ctypes  # noqa: F401
    from accumulation_amd import ipa_pc_as as M
    from accumulation_amd.engine import _ptr
    from accumulation_amd.ipa_pc import InnerProductArgPC as IpaPC
    from accumulation_amd.sponge import PoseidonSponge, Sha256Sponge
    AS = M.AtomicASForInnerProductArgPC
    fr = Fr(ctx.curve)
    degree = (1 << lg) - 1
    hr = HarnessRng(0xD1 ^ seed)
    old_cls = AS.sponge_cls, IpaPC.sponge_cls
    curve = ctx.curve
    AS.sponge_cls = IpaPC.sponge_cls = (lambda: PoseidonSponge(curve)) if sponge == "poseidon" else Sha256Sponge
    try:
        pp = IpaPC.setup(ctx, degree, seed=0x1BA5EED)
        pk, vk, dk = AS.index(pp, degree)
        poly = ctx.random_vector(77 + seed, degree + 1, mont=True)
        comm, rand = IpaPC.commit(pk.ipa_ck, poly, harness_shape, hr)
        point = hr.field() % fr.r
        z = ctx.vector(degree + 1)
        ffi.check(ctx._lib.amsm_vec_powers(ctx._h, _ptr(fr.to_limbs(point)), degree + 1, z.ptr), "amsm_vec_powers")
        value = IpaPC._inner_product(ctx, fr, poly, z)
        proof = IpaPC.open(pk.ipa_ck, poly, comm, point, rand, harness_shape, hr)
        inputs = [M.InputInstance(comm, point, value, proof)]
        rng = hr if harness_shape else None
        first, _ = AS.prove(pk, inputs, [], rng, None)
        old = [first.instance, first.instance] if harness_shape else [first.instance]
        acc, pr = AS.prove(pk, inputs, old, rng, None)
        assert AS.decide(dk, acc, None)
    finally:
        AS.sponge_cls, IpaPC.sponge_cls = old_cls
    s = Ser(ctx)
    return s.ipa_as_instance(acc.instance), s.ipa_as_proof(pr)


def trivial_pc_as(ctx, lg, harness_shape, sponge="poseidon", seed=0):
    """profile_trivial: degree 2^lg - 1 (lh_param_gen / lh_input_gen, examples/scaling-as.rs:145-197); no zk mode"""
    from accumulation_amd import trivial_pc_as as M
    from accumulation_amd.sponge import PoseidonSponge, Sha256Sponge
    AS = M.ASForTrivialPC
    fr = Fr(ctx.curve)
    degree = (1 << lg) - 1
    hr = HarnessRng(0x7121A1 ^ seed)
    pp = M.TrivialPC.setup(ctx, degree, seed=0x7121A1)
    ck, _ = M.TrivialPC.trim(pp, degree)
    pk, vk, dk = AS.index(pp, degree)
    poly = M.LabeledPolynomial([hr.field() % fr.r for _ in range(degree + 1)])
    comm = M.TrivialPC.commit(ck, poly)
    point = hr.field() % fr.r
    inputs = [M.Input(M.InputInstance(comm, point, poly.evaluate(fr, point)), poly)]
    curve = ctx.curve
    mk = (lambda: PoseidonSponge(curve)) if sponge == "poseidon" else Sha256Sponge
    first, _ = AS.prove(pk, inputs, [], None, mk())
    old = [first, first] if harness_shape else [first]
    acc, pr = AS.prove(pk, inputs, old, None, mk())
    assert AS.decide(dk, acc, None)
    s = Ser(ctx)
    return s.trivial_accumulator(acc), s.trivial_proof(pr)


SCHEMES = {"hp_as": hp_as, "r1cs_nark_as": r1cs_nark_as, "ipa_pc_as": ipa_pc_as, "trivial_pc_as": trivial_pc_as}
