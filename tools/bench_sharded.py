#!/usr/bin/env python3
"""BASELINE configs 3 and 4 in their multi-GPU layout: hp_as over 2^22-element vectors and r1cs_nark_as over 2^18
constraints with the committer key, the matrices' rows and every constraint-length vector SHARDED over the ranks
(dist.ShardedCommitterKey; strong scaling: the global problem is fixed, rank r holds [lo_r, hi_r)).  The only exchange
per batch of commitments is one all-gather of fixed-size partial records (RCCL with --backend nccl).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_sharded.py
    python tools/bench_sharded.py                       # N = 1
    ... --backend gloo --one-gpu                        # several ranks sharing GPU 0 (functional check only)

Rank 0 prints one JSON object per line; times are the max over ranks."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from accumulation_amd import CommitterKey, Context, ffi  # noqa: E402
from accumulation_amd.dist import ShardedCommitterKey, shard_bounds  # noqa: E402
from accumulation_amd.hp_as import (ASForHadamardProducts as AS, Accumulator, InputInstance, InputWitness,  # noqa: E402
                                    compute_hp)
from accumulation_amd.scalar_field import Fr  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=22, help="hp_as: global vector length 2^log2n")
    ap.add_argument("--log2c", type=int, default=18, help="r1cs_nark_as: 2^log2c constraints")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--one-gpu", action="store_true", help="every rank on GPU 0")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    device = 0 if args.one_gpu else local_rank
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")

    def agree_max(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=f"cuda:{device}" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def emit(**kw):
        if rank == 0:
            print(json.dumps(kw), flush=True)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    ctx = Context(ffi.AMSM_PALLAS, device=device)
    fr = Fr(ctx.curve)

    def sharded_key(n_global, seed):
        lo, hi = shard_bounds(n_global, rank, world)
        tmp = CommitterKey.generate(ctx, seed, 1, ffi.AMSM_BASES_NO_PRECOMPUTE)  # hiding generator: same on all ranks
        hg, _ = tmp.read()
        tmp.free()
        local = CommitterKey.generate(ctx, seed + 1 + rank, hi - lo, ffi.AMSM_BASES_PRECOMPUTE)
        local.hiding_generator = hg[0].copy()
        return ShardedCommitterKey(local, n_global)

    # ---- config 4: hp_as, 1 input + 1 old accumulator, no zk -----------------------------------------------------
    n = 1 << args.log2n
    ck = sharded_key(n, 0x5EED1001)
    m = ck.local_num_elems()
    pk, vk, dk = AS.index(ck)

    def make_input(seed):
        a = ctx.random_vector(seed + 16 * rank, m, mont=True)
        b = ctx.random_vector(seed + 16 * rank + 1, m, mont=True)
        pts, infs = ck.msm_batch([a, b, compute_hp(ctx, a, b)], True)
        return Accumulator(InputInstance(*[(pts[i], bool(infs[i])) for i in range(3)]), InputWitness(a, b, None))

    inp0, inp1 = make_input(100), make_input(200)
    acc0, _ = AS.prove(pk, [inp0], [], None, None)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        acc, proof = AS.prove(pk, [inp1], [acc0], None, None)
    barrier()
    dt = agree_max((time.perf_counter() - t0) / args.reps)
    ok = AS.verify(ctx, vk, [inp1.instance], [acc0.instance], acc.instance, proof, None)
    barrier()
    t0 = time.perf_counter()
    dec = AS.decide(dk, acc, None)
    barrier()
    t_dec = agree_max(time.perf_counter() - t0)
    emit(kind="hp_as_sharded", n_gpus=world, backend=args.backend if world > 1 else None, log2n=args.log2n,
         elements_per_rank=m, accumulations_per_s=1 / dt, prove_ms=dt * 1e3, decide_ms=t_dec * 1e3, verify_ok=bool(ok),
         decide_ok=bool(dec), scaling="strong")
    del inp0, inp1, acc0, acc, ck, pk, dk
    ctx.empty_cache()

    # ---- config 3: r1cs_nark_as, 1 input + 1 old accumulator, no zk ----------------------------------------------
    from accumulation_amd import r1cs_nark as nark
    from accumulation_amd.r1cs_nark_as import ASForR1CSNark as NAS, Input, InputInstance as NarkInstance
    from accumulation_amd.sponge import Sha256Sponge
    nc, n_in = 1 << args.log2c, 5
    n_inst = n_in + 1
    A = [[(1, n_inst)] for _ in range(nc - 1)] + [[]]
    B = [[(1, n_inst + 1)] for _ in range(nc - 1)] + [[]]
    Cm = [[(1, 1)] for _ in range(nc - 1)] + [[]]
    ck = sharded_key(nc, 0x5EED2002)
    ipk = nark.index(ctx, A, B, Cm, n_inst, n_inst + 2, ck=ck)
    pk, vk, dk = NAS.index(ipk)

    def nark_input(a, b):
        inst = [1, a * b % fr.r] + [a] * (n_in - 1)
        proof = nark.prove(ipk, inst, ctx.upload(fr.to_limbs_many([a, b])), False, NAS._sponges(Sha256Sponge())[0], None)
        return Input(NarkInstance(inst, proof.first_msg), proof.second_msg)

    i0, i1 = nark_input(3, 5), nark_input(7, 11)
    acc0, _ = NAS.prove(pk, [i0], [], None, None)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        acc, proof = NAS.prove(pk, [i1], [acc0], None, None)
    barrier()
    dt = agree_max((time.perf_counter() - t0) / args.reps)
    ok = NAS.verify(ctx, vk, [i1.instance], [acc0.instance], acc.instance, proof, None)
    barrier()
    t0 = time.perf_counter()
    dec = NAS.decide(dk, acc, None)
    barrier()
    t_dec = agree_max(time.perf_counter() - t0)
    emit(kind="r1cs_nark_as_sharded", n_gpus=world, backend=args.backend if world > 1 else None, log2_constraints=args.log2c,
         constraints_per_rank=ck.local_num_elems(), accumulations_per_s=1 / dt, prove_ms=dt * 1e3, decide_ms=t_dec * 1e3,
         verify_ok=bool(ok), decide_ok=bool(dec), scaling="strong")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
