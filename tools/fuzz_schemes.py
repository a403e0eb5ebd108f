"""Randomised runs of the accumulation layers against the big-int oracle (oracle/pyref_as.py), beyond the fixed scenarios of
tests/test_as_layers_vs_oracle_gpu.py: hp_as proves with random vector lengths, numbers of inputs / old accumulators, zk on or
off and fresh seeds, and r1cs_nark_as accumulation CHAINS of random shape (new inputs and subsets of the earlier accumulators per
step); every combined instance, witness vector, proof commitment and decide() must match the oracle bit for bit.  Round 6:
ipa_pc_as chains (random degree 2^k - 1, 2-4 inputs then 0-2 more plus the first accumulator, zk on or off: the first prove's
combine step against the oracle, the second verified and decided) and trivial_pc_as chains (random degree, random template shape:
every step verified, the last accumulator decided, and its commitment recomputed by the oracle's naive MSM over the key).
Usage: python tools/fuzz_schemes.py [seconds] [seed] [--host] [--bls12-381]   (--host: the library's host backend, no GPU needed;
--bls12-381: every layer over BLS12-381 G1 / its scalar field instead of Pallas -- the reference's own tests run the layers on Pallas
only, BASELINE config 3 runs ipa_pc_as on BLS12-381)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402

from accumulation_amd import Context, PedersenCommitment, ffi  # noqa: E402
from oracle import pyref as o  # noqa: E402
from tests import helpers as h  # noqa: E402
import tests.test_as_layers_vs_oracle_gpu as T  # noqa: E402
from tests.test_as_layers_vs_oracle_gpu import hp_case  # noqa: E402
from tests.test_hp_as_scheme_gpu import SchemeRng  # noqa: E402
from tests.test_r1cs_nark_gpu import dummy_circuit  # noqa: E402

HOST = "--host" in sys.argv
BLS = "--bls12-381" in sys.argv
argv = [a for a in sys.argv if a not in ("--host", "--bls12-381")]
budget = float(argv[1]) if len(argv) > 1 else 60.0
seed = int(argv[2]) if len(argv) > 2 else 1
rs = np.random.RandomState(seed)
C = o.BLS12_381_G1 if BLS else o.PALLAS
T.C = C  # (the layer helpers read their module's curve when they run)
ctx = Context(C.curve_id, device=ffi.AMSM_DEVICE_HOST if HOST else 0)
t_end = time.time() + budget
n_cases = n_chains = n_steps = n_ipa = n_trivial = 0


def nark_env():
    """the module fixture of the test file, built by hand"""
    from accumulation_amd import r1cs_nark as nark
    A, B, C_, _, _ = dummy_circuit(T.N_IN, T.N_CON, 2, 3, C.r)
    ipk = nark.index(ctx, A, B, C_, T.N_IN + 1, T.N_IN + 3, key_seed=int(rs.randint(1 << 30)))
    xy, _ = ipk.ck.read()
    gens = [h.np_to_point(C, xy[i], 0) for i in range(T.N_CON)]
    H = h.np_to_point(C, ipk.ck.hiding_generator, 0)
    return ctx, ipk, (A, B, C_), gens, H


def nark_chain():
    global n_steps
    from accumulation_amd.r1cs_nark_as import ASForR1CSNark as AS
    env = nark_env()
    make_zk = bool(rs.rand() < 0.5)
    rng = SchemeRng(int(rs.randint(1, 1 << 20)))
    accs, last = [], None
    for step in range(int(rs.randint(1, 5))):
        olds = [accs[i] for i in sorted(rs.choice(len(accs), size=min(len(accs), int(rs.randint(0, 3))), replace=False))] if accs else []
        n_new = int(rs.randint(0, 3))
        ins = T.nark_inputs(env, n_new, make_zk, rng)
        acc, proof, ref, dk = T.nark_as_step(env, ins, olds, make_zk, int(rs.randint(1, 1 << 20)))
        T.assert_nark_acc_equal(acc, proof, ref)
        accs.append(acc)
        last = (acc, ref, dk)
        n_steps += 1
    acc, ref, dk = last
    A, B, C_ = env[2]
    assert T.oa.nark_as_decide(C, A, B, C_, env[3], env[4], ref) and AS.decide(dk, acc, None)


def ipa_chain():
    T.ipa_as_case(ctx, int(rs.choice([3, 7, 15, 31])), int(rs.randint(2, 5)), int(rs.randint(0, 3)), bool(rs.rand() < 0.5),
                  int(rs.randint(1, 1 << 20)))


def trivial_chain():
    from accumulation_amd.trivial_pc_as import ASForTrivialPC as AS, TrivialPC
    import tests.test_trivial_pc_as_scheme_gpu as TT
    degree = int(rs.choice([1, 2, 5, 11, 30]))
    pp = TrivialPC.setup(ctx, degree)
    env = (ctx, pp)
    ck, _ = TrivialPC.trim(pp, degree)
    pk, vk, dk = AS.index(pp, degree)
    rng = SchemeRng(int(rs.randint(1, 1 << 20)))
    shape = [int(rs.randint(0, 4)) for _ in range(int(rs.randint(1, 5)))]
    if shape[0] == 0 and sum(shape) == 0:
        shape[0] = 1
    inputs = TT.generate_inputs(env, ck, sum(shape), rng)
    old, start = [], 0
    for k in shape:
        step = inputs[start:start + k]
        start += k
        acc, proof = AS.prove(pk, step, old, None, None)
        assert AS.verify(ctx, vk, [i.instance for i in step], [a.instance for a in old], acc.instance, proof, None)
        old.append(acc)
    assert AS.decide(dk, old[-1], None)
    xy, inf = ck.read()
    gens = [h.np_to_point(C, xy[i], inf[i]) for i in range(degree + 1)]
    cm = old[-1].instance.commitment
    assert h.np_to_point(C, cm.elem[0], cm.elem[1]) == o.msm_naive(C, gens, [c % C.r for c in old[-1].witness.coeffs])


while time.time() < t_end:
    u = rs.rand()
    if u < 0.25:
        nark_chain()
        n_chains += 1
        continue
    if u < 0.40:
        ipa_chain()
        n_ipa += 1
        continue
    if u < 0.55:
        trivial_chain()
        n_trivial += 1
        continue
    n = int(rs.choice([1, 2, 3, 7, 23, 64, 257]))
    ck = PedersenCommitment.setup(ctx, n, seed=int(rs.randint(1 << 30)))
    xy, _ = ck.read()
    gens = [h.np_to_point(C, xy[i], 0) for i in range(n)]
    H = h.np_to_point(C, ck.hiding_generator, 0)
    for _ in range(3):
        n_in = int(rs.randint(0, 5))
        n_acc = int(rs.randint(0, 4))
        make_zk = bool(rs.rand() < 0.5)
        if n_in + n_acc + (1 if make_zk else 0) > 8:
            continue
        try:
            hp_case(ctx, ck, gens, H, n, make_zk, (n_in, n_acc), seed=int(rs.randint(1000, 1 << 20)))
        except AssertionError:
            print("FAILED case", dict(n=n, n_in=n_in, n_acc=n_acc, make_zk=make_zk, seed=seed, case=n_cases), flush=True)
            raise
        n_cases += 1
    ck.free()
print(f"fuzz_schemes ok: {n_cases} hp_as proves, {n_chains} r1cs_nark_as chains ({n_steps} accumulation steps), {n_ipa} ipa_pc_as chains and "
      f"{n_trivial} trivial_pc_as chains against the oracle on {C.name} in {budget:.0f} s (seed {seed}{', host backend' if HOST else ''})")
