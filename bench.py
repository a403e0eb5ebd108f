#!/usr/bin/env python3
"""bench.py -- MSM throughput (point-scalar pairs/sec) at 2^20 Pallas on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: one MSM of 2^log2n pairs per
GPU (scalars and the committer key already resident in HBM), result normalised to affine on the host.
With N > 1 each rank owns a disjoint shard of the key (weak scaling: 2^log2n pairs per GPU, the job is
one N*2^log2n-pair MSM); ranks all-gather their 128-byte partial sums over RCCL and fold them.

Prints ONE JSON line on rank 0 (contract in the task statement): metric/value/unit/..., plus
  "roofline":     the dominant kernel (bucket accumulation) against the HBM roofline, achieved =
                  algorithmic bytes (96 B per pair) / mean kernel time measured
                  with hipEvents on the engine's stream inside the timed region;
  "cpu_baseline": the plain-C ark-ec-style restatement (oracle/ark_msm.c, kind "port") timed on this
                  box's host cores on the same inputs (rank 0, N = 1 only), and used to check the GPU
                  result bit-for-bit;
  "accumulations": accumulations/sec (the metric's second half) of hp_as at 2^22, r1cs_nark_as at 2^18 constraints and
                  ipa_pc_as at d + 1 = 2^16, measured after and outside the timed region (N = 1 only; --no-schemes skips).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SEED_SCALARS = 0x5EED0001
SEED_POINTS = 0x5EED1001


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--log2n", type=int, default=20, help="pairs per GPU = 2^log2n")
    ap.add_argument("--curve", default="pallas", choices=["pallas", "bls12_381_g1"])
    ap.add_argument("--no-precompute", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-schemes", action="store_true", help="skip the accumulations/sec lines (second half of the metric)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --one-gpu: functional check of the N > 1 path with every rank on GPU 0 (numbers meaningless)")
    ap.add_argument("--one-gpu", action="store_true")
    ap.add_argument("--sync", action="store_true", help="one synchronous MSM call per step (no MSMs overlapped)")
    ap.add_argument("--cpu-log2n", type=int, default=None, help="sample size of the CPU baseline (default: log2n)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from accumulation_amd import CommitterKey, Context, VariableBaseMSM, ffi
    from accumulation_amd.engine import _ptr
    import ctypes as C

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    curve_id = ffi.AMSM_PALLAS if args.curve == "pallas" else ffi.AMSM_BLS12_381_G1
    n = 1 << args.log2n
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ctx = Context(curve_id, device=local_rank, stream=stream.cuda_stream)
        flags = ffi.AMSM_BASES_NO_PRECOMPUTE if args.no_precompute else ffi.AMSM_BASES_PRECOMPUTE
        t0 = time.time()
        ck = CommitterKey.generate(ctx, SEED_POINTS + rank, n, flags)
        t_key = time.time() - t0
        # four distinct scalar vectors, cycled over the steps (step k uses vector k % 4)
        n_distinct = 4
        vecs = [ctx.random_vector(SEED_SCALARS + 1000 * j + rank, n, mont=False) for j in range(n_distinct)]
        scalars = vecs[0]
        ctx.synchronize()
        sharded = None
        if world > 1:
            from accumulation_amd.dist import HipEngine, ShardedMSM
            sharded = ShardedMSM(HipEngine(ctx, ck))
        last = {}

        def run_steps(k):
            """k steps = k MSMs, issued as one batch call: the ABI keeps several MSMs in flight so the latency-bound
            tail of one and the sort of the next run beside the bucket accumulation of the current one, like the
            prover's back-to-back commits of src/hp_as/mod.rs:354-388.  N > 1: every rank runs its k local MSMs
            the same way -> k partial records -> ONE RCCL all-gather of raw bytes -> identical fold on every rank."""
            if world == 1 and args.sync:
                for i in range(k):
                    last["xy"], last["inf"] = VariableBaseMSM.multi_scalar_mul(ck, vecs[i % n_distinct])
            elif world == 1:
                pts, infs = VariableBaseMSM.multi_scalar_mul_batch(ck, [vecs[i % n_distinct] for i in range(k)],
                                                                   mont=False)
                last["xy"], last["inf"] = pts[0], bool(infs[0])
            else:
                outs, infs = sharded.msm_batch([vecs[i % n_distinct] for i in range(k)], mont=False)
                last["xy"], last["inf"] = outs[0], bool(infs[0])

        def sync_all():
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        run_steps(args.warmup)
        ctx.set_profiling(True)
        sync_all()
        t0 = time.perf_counter()
        run_steps(args.steps)
        sync_all()
        elapsed = time.perf_counter() - t0
        stage_ms = ctx.stage_ms()
        ctx.set_profiling(False)
        # latency of ONE synchronous MSM call (no overlap between consecutive MSMs), for reference
        out = np.zeros((2 * ctx.fq_limbs,), dtype=np.uint64)
        inf = C.c_uint8(0)
        ctx.set_profiling(True)
        t1 = time.perf_counter()
        for _ in range(5):
            ffi.check(ctx._lib.amsm_msm_device(ctx._h, ck._h, 0, scalars.ptr, n, 0, _ptr(out), C.byref(inf)),
                      "amsm_msm_device")
        ms_sync = (time.perf_counter() - t1) / 5 * 1e3
        stage_ms_alone = ctx.stage_ms()  # stages of the last blocking call: nothing else on the GPU
        ctx.set_profiling(False)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    total_pairs = n * world * args.steps
    value = total_pairs / elapsed
    ms_per_step = elapsed / args.steps * 1e3

    result = None
    if rank == 0:
        bytes_per_pair = 32 + 16 * ctx.fq_limbs  # 32 B scalar + affine point (64 B Pallas / 96 B BLS12-381)
        dom = "accum_l0"
        dom_ms = stage_ms.get(dom, 0.0)
        achieved = (n * bytes_per_pair / (dom_ms * 1e-3)) / 1e9 if dom_ms > 0 else 0.0
        result = {
            "metric": "MSM throughput (point-scalar pairs/sec) at 2^20 Pallas",
            "value": value,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": f"2^{args.log2n}-pair {args.curve} MSM per GPU (north_star size; uniform 254-bit scalars, "
                            f"G_i = k_i*G), key resident{'' if args.no_precompute else ' + precomputed window multiples'}",
                "pairs_per_gpu": n,
                "curve": args.curve,
                "precomputed_key": bool(ck.precomputed),
                "parallelism": f"point-sharded x{world}" + (" + one RCCL all-gather of the steps' partial records" if world > 1 else ""),
                "key_setup_s": round(t_key, 3),
                "ms_per_msm_synchronous_call": round(ms_sync, 4),
                "msms_in_flight": 1 if args.sync else 3,
                "seeds": {"scalars": SEED_SCALARS, "points": SEED_POINTS},
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_accum_l0 (bucket accumulation)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic(args),
                "kernel_ms": dom_ms,
                "algorithmic_bytes_per_launch": n * bytes_per_pair,
                # the kernel is integer-VALU bound, not HBM bound: the ceiling that binds is
                # the issue rate of the 10-multiplication mixed addition measured in isolation (tools/fp_bench.hip)
                "alu": alu_roofline(args, ck, n, dom_ms),
                # the same kernel with the GPU to itself (one blocking MSM call after the timed region)
                "kernel_ms_unshared": stage_ms_alone.get(dom, 0.0),
                "alu_unshared": alu_roofline(args, ck, n, stage_ms_alone.get(dom, 0.0)),
            },
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(ctx, ck, scalars, curve_id, args, out.copy(), bool(inf.value),
                                                  last["xy"], last["inf"])
        if world == 1 and not args.no_schemes and args.log2n == 20 and args.curve == "pallas":
            result["accumulations"] = scheme_rates()
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def scheme_rates():
    """accumulations/sec (one `prove` = one accumulation; BASELINE.json's second metric) of the three schemes at the
    sizes of configs 1, 3 and 4, after the timed region and outside it: tools/bench_configs.py's workloads (1 input + 1 old
    accumulator, no zk), each verified and decided.  Never fails the bench line: an error is reported in place."""
    import importlib.util
    out = {}
    try:
        spec = importlib.util.spec_from_file_location("amsm_bench_configs", os.path.join(ROOT, "tools", "bench_configs.py"))
        bc = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bc)
        got = []
        bc.emit = lambda **kw: got.append(kw)
        for name, fn in (("hp_as_2^22", lambda: bc.bench_hp_as(22)), ("r1cs_nark_as_2^18", lambda: bc.bench_r1cs_nark_as(18)),
                         ("ipa_pc_as_2^16", lambda: bc.bench_ipa(16))):
            try:
                fn()
                r = got[-1]
                out[name] = {"accumulations_per_s": round(r["accumulations_per_s"], 2), "prove_ms": round(r["prove_ms"], 3),
                             "decide_ms": round(r["decide_ms"], 3), "verified": bool(r["verify_ok"] and r["decide_ok"])}
            except Exception as e:  # noqa: BLE001
                out[name] = {"error": f"{type(e).__name__}: {e}"}
    except Exception as e:  # noqa: BLE001
        out["error"] = f"{type(e).__name__}: {e}"
    return out


ALU_PEAK_GMADD = {"pallas": 18.7, "bls12_381_g1": 6.95}  # isolated xyzz_madd, all SIMDs busy (tools/fp_bench.hip)


def alu_roofline(args, ck, n, kernel_ms):
    """Mixed additions per second of accumulate L0 against the isolated-ALU ceiling of the same formula."""
    if kernel_ms <= 0 or not ck.precomputed:
        return None
    windows = 16  # both curves have 255-bit scalars: c = 16, W = 16 at 2^20
    if args.log2n != 20:
        return None
    madds = n * windows  # one gathered mixed addition per (pair, window); c = 16, W = 16 at 2^20
    achieved = madds / (kernel_ms * 1e-3) / 1e9
    peak = ALU_PEAK_GMADD[args.curve]
    return {"unit": "G mixed-additions/s", "achieved": achieved, "peak": peak, "frac": achieved / peak,
            "madds_per_launch": madds}


def pmc_traffic(args):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (tools/profile_round.sh -> profiles/*_pmc_accum_l0.json; FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for 16-B-per-lane loads).  Only valid for the default workload; null otherwise."""
    if args.log2n != 20 or args.curve != "pallas" or args.no_precompute:
        return None
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_accum_l0.json")))
    if not files:
        return None
    try:
        return json.load(open(files[-1]))["hbm_traffic_bytes_per_launch"]
    except Exception:
        return None


def cpu_baseline(ctx, ck, scalars, curve_id, args, gpu_out, gpu_inf, batch_out, batch_inf):
    """Time the plain-C ark-ec-style restatement on the host cores, on the same inputs, and check the
    GPU result against it bit-for-bit.  (Only this leg of bench.py touches oracle/.)"""
    from oracle import cref

    cpu_log2n = args.cpu_log2n if args.cpu_log2n is not None else args.log2n
    m = min(1 << cpu_log2n, len(ck))
    xy, _ = ck.read(0, m)
    sc = scalars.download()[:m]
    cores = os.cpu_count() or 1
    n_windows = -(-255 // cref.load().ark_msm_window_bits(m))
    threads = max(1, min(cores, n_windows))
    t0 = time.perf_counter()
    cpu_out, cpu_inf = cref.msm(curve_id, xy, sc, threads=threads)
    t_par = time.perf_counter() - t0
    # single-thread figure (the reference's default features) on a bounded 2^16 sample
    ms = min(m, 1 << 16)
    t0 = time.perf_counter()
    cref.msm(curve_id, xy[:ms], sc[:ms], threads=1)
    t_one = time.perf_counter() - t0
    match = None
    if m == len(ck):
        match = bool(np.array_equal(cpu_out, gpu_out) and cpu_inf == gpu_inf
                     and np.array_equal(cpu_out, batch_out) and cpu_inf == batch_inf)
    return {
        "value": m / t_par,
        "unit": "pairs/s",
        "cores": threads,
        "kind": "port",
        "sample": f"one 2^{cpu_log2n}-pair MSM on the bench inputs, window-parallel on {threads} threads "
                  f"(ark-ec `parallel` semantics; host has {cores} cores); {t_par:.2f} s",
        "single_thread_value": ms / t_one,
        "single_thread_sample": f"one 2^{ms.bit_length() - 1}-pair MSM, 1 thread (reference default features); {t_one:.2f} s",
        "gpu_result_bit_exact_vs_cpu": match,
    }


if __name__ == "__main__":
    sys.exit(main())
