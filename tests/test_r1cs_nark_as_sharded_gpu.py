"""The R1CS NARK and ASForR1CSNark over a constraint-sharded key (BASELINE config 3's layout):
two or three processes share the test box's GPU; each loads the constraints [lo, hi) of the three matrices and the
matching slice of the Pedersen key, the assignment is replicated, and the prover / verifier / accumulation-scheme
drivers run unchanged -- commitments go through dist.ShardedMSM (gloo here, RCCL on a multi-GPU node).  NARK proofs,
accumulator instances and accumulation proofs must equal the unsharded run's bit for bit; verify / decide pass on every
rank; constraint-length witness vectors are the slices."""
import hashlib
import os
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NUM_INPUTS, NUM_CONSTRAINTS = 5, 257
KEY_SEED = 0xB0B


def _pt(p):
    return (np.asarray(p[0], dtype=np.uint64).tolist(), bool(p[1]))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).hexdigest()


def _run(ctx, ck, make_zk, NUM_CONSTRAINTS=NUM_CONSTRAINTS):
    from accumulation_amd import r1cs_nark as nark
    from accumulation_amd.r1cs_nark_as import ASForR1CSNark as AS, Input, InputInstance
    from accumulation_amd.scalar_field import MODULI, Fr
    from accumulation_amd.sponge import Sha256Sponge
    from tests.test_hp_as_scheme_gpu import SchemeRng
    from tests.test_r1cs_nark_gpu import dummy_circuit
    r = MODULI[ctx.curve]
    fr = Fr(ctx.curve)
    A, B, C_, _, _ = dummy_circuit(NUM_INPUTS, NUM_CONSTRAINTS, 2, 3, r)
    ipk = nark.index(ctx, A, B, C_, NUM_INPUTS + 1, NUM_INPUTS + 3, ck=ck)
    assert ipk.index_info.num_constraints == NUM_CONSTRAINTS
    rng = SchemeRng(2024)
    inputs, nark_ok = [], []
    for _ in range(3):
        a, b = rng.field() % r, rng.field() % r
        _, _, _, inst, w = dummy_circuit(NUM_INPUTS, NUM_CONSTRAINTS, a, b, r)
        nark_sponge, _, _ = AS._sponges(Sha256Sponge())
        proof = nark.prove(ipk, inst, ctx.upload(fr.to_limbs_many(w)), make_zk, nark_sponge, rng if make_zk else None)
        nark_sponge, _, _ = AS._sponges(Sha256Sponge())
        nark_ok.append(bool(nark.verify(ipk, inst, proof, nark_sponge)))
        inputs.append(Input(InputInstance(inst, proof.first_msg), proof.second_msg))
    pk, vk, dk = AS.index(ipk)
    zk_rng = rng if make_zk else None
    acc1, proof1 = AS.prove(pk, inputs[:2], [], zk_rng, None)
    ok1 = AS.verify(ctx, vk, [x.instance for x in inputs[:2]], [], acc1.instance, proof1, None)
    acc2, proof2 = AS.prove(pk, inputs[2:], [acc1], zk_rng, None)
    ok2 = AS.verify(ctx, vk, [inputs[2].instance], [acc1.instance], acc2.instance, proof2, None)
    dec = AS.decide(dk, acc2, None)
    i = acc2.instance
    return {"ok": nark_ok + [bool(ok1), bool(ok2), bool(dec)],
            "first_msgs": [[_pt(x.instance.first_round_message.comm_a), _pt(x.instance.first_round_message.comm_b),
                            _pt(x.instance.first_round_message.comm_c)] for x in inputs],
            "instance": [_pt(i.comm_a), _pt(i.comm_b), _pt(i.comm_c), _pt(i.hp_instance.comm_1), _pt(i.hp_instance.comm_2),
                         _pt(i.hp_instance.comm_3)],
            "r1cs_input": [int(x) % r for x in i.r1cs_input],
            "hp_low": [_pt(p) for p in proof2.hp_proof.product_poly_comm.low],
            "blinded": acc2.witness.r1cs_blinded_witness.download().tolist(),
            "hp_a": acc2.witness.hp_witness.a_vec.download()}


def _worker(rank, world, init_file, make_zk, q, NUM_CONSTRAINTS=NUM_CONSTRAINTS, digest=False, device=0):
    import torch.distributed as dist
    from accumulation_amd import CommitterKey, Context, ffi
    from accumulation_amd.dist import ShardedCommitterKey
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    try:
        ctx = Context(ffi.AMSM_PALLAS, device=device)
        tmp = CommitterKey.generate(ctx, KEY_SEED, NUM_CONSTRAINTS + 1, ffi.AMSM_BASES_NO_PRECOMPUTE)
        xy, _ = tmp.read()
        ck = ShardedCommitterKey.from_global(ctx, xy[:NUM_CONSTRAINTS], hiding_generator=xy[NUM_CONSTRAINTS].copy())
        res = _run(ctx, ck, make_zk, NUM_CONSTRAINTS)
        res["range"] = (ck.lo, ck.hi)
        res["hp_a"] = sha(res["hp_a"]) if digest else res["hp_a"].tolist()
        q.put((rank, res))
        ctx.close()
    finally:
        dist.destroy_process_group()


def run_sharded_vs_unsharded(make_zk, world, NUM_CONSTRAINTS=NUM_CONSTRAINTS, digest=False, timeout=600, device=0):
    """device: 0 = the GPU; ffi.AMSM_DEVICE_HOST = every rank on the library's host backend (tests/host_backend/: no GPU needed)"""
    import torch.multiprocessing as mp
    from accumulation_amd import CommitterKey, Context, ffi
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    with tempfile.TemporaryDirectory() as d:
        procs = [mpc.Process(target=_worker, args=(r, world, os.path.join(d, "init"), make_zk, q, NUM_CONSTRAINTS, digest, device))
                 for r in range(world)]
        for p in procs:
            p.start()
        got = dict(q.get(timeout=timeout) for _ in range(world))
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    ctx = Context(ffi.AMSM_PALLAS, device=device)
    tmp = CommitterKey.generate(ctx, KEY_SEED, NUM_CONSTRAINTS + 1, ffi.AMSM_BASES_NO_PRECOMPUTE)
    xy, _ = tmp.read()
    ck = CommitterKey.load(ctx, xy[:NUM_CONSTRAINTS], None, ffi.AMSM_BASES_DEFAULT, hiding_generator=xy[NUM_CONSTRAINTS].copy())
    ref = _run(ctx, ck, make_zk, NUM_CONSTRAINTS)
    assert all(ref["ok"])
    cut = (lambda v, lo, hi: sha(v[lo:hi])) if digest else (lambda v, lo, hi: v[lo:hi].tolist())
    covered = 0
    for rank in range(world):
        r = got[rank]
        lo, hi = r["range"]
        assert all(r["ok"]), (rank, r["ok"])
        for key in ("first_msgs", "instance", "r1cs_input", "hp_low", "blinded"):
            assert r[key] == ref[key], (rank, key)
        assert r["hp_a"] == cut(ref["hp_a"], lo, hi), rank
        covered += hi - lo
    assert covered == NUM_CONSTRAINTS
    ctx.close()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("make_zk", [False, True], ids=["no_zk", "zk"])
def test_r1cs_nark_as_sharded_equals_unsharded(built_lib, make_zk, world):
    run_sharded_vs_unsharded(make_zk, world)
