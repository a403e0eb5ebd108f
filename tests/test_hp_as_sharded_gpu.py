"""hp_as over a point-sharded committer key (BASELINE config 4's layout): two processes share
the one GPU of the test box, each holds half of the generators and half of every vector, and the scheme driver runs
unchanged on the slices -- commitments go through dist.ShardedMSM (per-rank partial records + one all-gather; gloo here,
RCCL on a multi-GPU node).  The accumulator instances and proofs must equal the unsharded run's bit for bit, the
accumulator's witness vectors must be the corresponding slices, and verify / decide must pass on every rank."""
import hashlib
import os
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 1000  # global vector / key length (odd split: 500 + 500; 333 + 333 + 334 for world 3)
KEY_SEED = 0xA11CE


def _pt(p):
    return (np.asarray(p[0], dtype=np.uint64).tolist(), bool(p[1]))


def _run(ctx, ck, lo, hi, make_zk, commit, AS, N=N):
    """Two inputs -> accumulator; a third input + that accumulator -> second accumulator; verify both, decide the last."""
    from accumulation_amd.hp_as import Accumulator, InputInstance, InputWitness, InputWitnessRandomness, compute_hp
    from accumulation_amd.scalar_field import Fr
    from tests.test_hp_as_scheme_gpu import SchemeRng
    fr = Fr(ctx.curve)
    rng_in = SchemeRng(0xC0FFEE)
    inputs, keep = [], []
    for t in range(3):
        a_full = ctx.random_vector(900 + 2 * t, N, mont=True)
        b_full = ctx.random_vector(901 + 2 * t, N, mont=True)
        keep += [a_full, b_full]
        a, b = a_full.view(lo, hi - lo), b_full.view(lo, hi - lo)
        prod = compute_hp(ctx, a, b)
        rnd = InputWitnessRandomness(rng_in.field(), rng_in.field(), rng_in.field()) if make_zk else None
        c = [commit(ck, v, fr.to_limbs(r) if make_zk else None)
             for v, r in zip((a, b, prod), (rnd.rand_1, rnd.rand_2, rnd.rand_3) if make_zk else (0, 0, 0))]
        inputs.append(Accumulator(InputInstance(*c), InputWitness(a, b, rnd)))
    pk, vk, dk = AS.index(ck)
    rng = SchemeRng(7) if make_zk else None
    acc1, proof1 = AS.prove(pk, inputs[:2], [], rng, None)
    ok1 = AS.verify(ctx, vk, [x.instance for x in inputs[:2]], [], acc1.instance, proof1, None)
    acc2, proof2 = AS.prove(pk, inputs[2:], [acc1], rng, None)
    ok2 = AS.verify(ctx, vk, [inputs[2].instance], [acc1.instance], acc2.instance, proof2, None)
    dec = AS.decide(dk, acc2, None)
    inst = [_pt(acc2.instance.comm_1), _pt(acc2.instance.comm_2), _pt(acc2.instance.comm_3)]
    low = [_pt(p) for p in proof2.product_poly_comm.low]
    return {"ok": [bool(ok1), bool(ok2), bool(dec)], "instance": inst, "low": low,
            "a": acc2.witness.a_vec.download(), "b": acc2.witness.b_vec.download(),
            "rand": None if acc2.witness.randomness is None else
            [acc2.witness.randomness.rand_1, acc2.witness.randomness.rand_2, acc2.witness.randomness.rand_3]}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).hexdigest()


def _worker(rank, world, init_file, make_zk, q, N=N, digest=False, device=0):
    """digest: the witness slices travel back as SHA-256 (config-size runs: 2^22 elements)"""
    import torch.distributed as dist
    from accumulation_amd import CommitterKey, Context, PedersenCommitment, ffi
    from accumulation_amd.dist import ShardedCommitterKey
    from accumulation_amd.hp_as import ASForHadamardProducts as AS
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    try:
        ctx = Context(ffi.AMSM_PALLAS, device=device)
        tmp = CommitterKey.generate(ctx, KEY_SEED, N + 1, ffi.AMSM_BASES_NO_PRECOMPUTE)
        xy, _ = tmp.read()
        ck = ShardedCommitterKey.from_global(ctx, xy[:N], hiding_generator=xy[N].copy())
        assert ck.supported_num_elems() == N and ck.local_num_elems() == ck.hi - ck.lo
        res = _run(ctx, ck, ck.lo, ck.hi, make_zk, PedersenCommitment.commit, AS, N)
        res["range"] = (ck.lo, ck.hi)
        for k in ("a", "b"):
            res[k] = sha(res[k]) if digest else res[k].tolist()
        # the no-input default accumulator on a sharded key (src/hp_as/mod.rs:685-696): local-length zero vectors
        acc0, proof0 = AS.prove(ck, [], [], None, None)
        res["default_ok"] = bool(AS.verify(ctx, N, [], [], acc0.instance, proof0, None)) and bool(AS.decide(ck, acc0, None)) \
            and acc0.witness.a_vec.n == ck.local_num_elems()
        q.put((rank, res))
        ctx.close()
    finally:
        dist.destroy_process_group()


def run_sharded_vs_unsharded(make_zk, world, N=N, digest=False, timeout=600, device=0):
    """device: 0 = the GPU; ffi.AMSM_DEVICE_HOST = every rank on the library's host backend (tests/host_backend/: no GPU needed)"""
    import torch.multiprocessing as mp
    from accumulation_amd import CommitterKey, Context, PedersenCommitment, ffi
    from accumulation_amd.hp_as import ASForHadamardProducts as AS
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    with tempfile.TemporaryDirectory() as d:
        procs = [mpc.Process(target=_worker, args=(r, world, os.path.join(d, "init"), make_zk, q, N, digest, device)) for r in range(world)]
        for p in procs:
            p.start()
        got = dict(q.get(timeout=timeout) for _ in range(world))
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    ctx = Context(ffi.AMSM_PALLAS, device=device)
    tmp = CommitterKey.generate(ctx, KEY_SEED, N + 1, ffi.AMSM_BASES_NO_PRECOMPUTE)
    xy, _ = tmp.read()
    ck = CommitterKey.load(ctx, xy[:N], None, ffi.AMSM_BASES_DEFAULT, hiding_generator=xy[N].copy())
    ref = _run(ctx, ck, 0, N, make_zk, PedersenCommitment.commit, AS, N)
    assert ref["ok"] == [True, True, True]
    cut = (lambda v, lo, hi: sha(v[lo:hi])) if digest else (lambda v, lo, hi: v[lo:hi].tolist())
    covered = 0
    for rank in range(world):
        r = got[rank]
        lo, hi = r["range"]
        assert r["ok"] == [True, True, True] and r["default_ok"], rank
        assert r["instance"] == ref["instance"], rank          # same accumulator instance, bit for bit
        assert r["low"] == ref["low"], rank                    # same proof
        assert r["rand"] == ref["rand"], rank
        assert r["a"] == cut(ref["a"], lo, hi) and r["b"] == cut(ref["b"], lo, hi), rank  # the witness is the slice
        covered += hi - lo
    assert covered == N
    ctx.close()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("make_zk", [False, True], ids=["no_zk", "zk"])
def test_hp_as_sharded_equals_unsharded(built_lib, make_zk, world):
    run_sharded_vs_unsharded(make_zk, world)
