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
  "accumulations": accumulations/sec (the metric's second half) of trivial_pc_as at 2^10, ipa_pc_as at d + 1 = 2^16,
                  r1cs_nark_as at 2^18 constraints and hp_as at 2^22 through the C++ scheme drivers (tools/profile_as.cpp =
                  the reference's harness examples/scaling-as.rs:38-138): the harness's own shape (1 input + the same
                  accumulator twice, MakeZK::Enabled) and the lighter n_all = 2 no-zk shape, each verified, decided and
                  serialised; measured after and outside the timed region (--no-schemes skips).  N > 1: `accumulations_multi_device`,
                  the same harness over one multi-device context of the N GPUs.

    python bench.py --gpus N --single-process [--devices 0,1,...]
runs the N-GPU job from ONE process through the multi-device context of the C ABI (amsm_ctx_create_multi: key sharded over
the devices, scalars resident per shard, partial sums gathered inside the library over RCCL / peer copies) -- the form a
single-process host (the reference's `prove` is one call in one process) would use; the torchrun form above stays the
driver's contract.
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
# (the pool's host driver supports dmabuf IPC only: RCCL's multi-process rendezvous needs this before the HIP runtime starts)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PREHEAT_MSMS = 40  # untimed MSMs before the warm-up steps: the device reaches its steady clocks (see main())
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
    ap.add_argument("--cpu-leg-max-log2", type=int, default=None,
                    help="cap the sizes of the accumulations' CPU legs (tests: the same code path in seconds; every entry names its size)")
    ap.add_argument("--no-bls", action="store_true", help="skip config.pairs_per_s_bls12_381_2p20 (BASELINE config 3's MSM)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --one-gpu: functional check of the N > 1 path with every rank on GPU 0 (numbers meaningless)")
    ap.add_argument("--one-gpu", action="store_true")
    ap.add_argument("--sync", action="store_true", help="one synchronous MSM call per step (no MSMs overlapped)")
    ap.add_argument("--no-preheat", action="store_true", help="no untimed MSMs before the warm-up steps (cold clocks)")
    ap.add_argument("--torch-stream", action="store_true",
                    help="hand the context a torch stream as its main stream (rounds 1-3; 8-10 %% slower: hardware-queue aliasing)")
    ap.add_argument("--strong", action="store_true",
                    help="N > 1: strong scaling -- ONE 2^log2n-pair MSM per step split over the ranks (2^log2n / N pairs each) "
                         "instead of 2^log2n pairs per rank; `scaling` says which")
    ap.add_argument("--cpu-log2n", type=int, default=None, help="sample size of the CPU baseline (default: log2n)")
    ap.add_argument("--single-process", action="store_true",
                    help="drive all --gpus devices from this process through amsm_ctx_create_multi (no torchrun)")
    ap.add_argument("--devices", default=None, help="--single-process: comma-separated device ids (default 0..gpus-1; an id "
                                                    "may repeat: several shards on one GPU, numbers meaningless)")
    args = ap.parse_args()
    if args.single_process:
        return main_single_process(args)

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
        raise SystemExit("bench.py needs a GPU: it times the HIP path (the library's host backend is never chosen implicitly)")
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
    if args.strong and world > 1:  # one 2^log2n-pair MSM per step, point-sharded: the first n % world ranks take one more
        from accumulation_amd.dist import shard_bounds
        lo_s, hi_s = shard_bounds(n, rank, world)
        n = hi_s - lo_s
    rank_info = None
    if world > 1:  # who actually joined: the driver reads this to see that N ranks ran on N distinct devices over RCCL
        rank_info = [None] * world
        dist.all_gather_object(rank_info, {"rank": rank, "device": int(torch.cuda.current_device()),
                                           "pci_bus_id": str(getattr(torch.cuda.get_device_properties(local_rank), "pci_bus_id", "")),
                                           "pairs": n})
    # The context creates its own streams (main, prep, tail -- in that order, so that each gets a hardware queue of its own).
    # Rounds 1-3 handed it a torch stream as its main stream: torch's stream pool had taken hardware queues before the library's
    # prep / tail streams were created, the accumulation shared a queue with one of them, and the SAME batch ran 8-10 % slower
    # than through a context that owns its stream (round 4, same box, 20 MSMs per call: 1.18-1.20 against 1.09 ms per MSM;
    # --torch-stream restores the old behaviour).  torch.cuda.synchronize() in sync_all() covers every stream of the device.
    stream = torch.cuda.Stream() if args.torch_stream else None
    import contextlib
    with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
        ctx = Context(curve_id, device=local_rank, stream=stream.cuda_stream if stream is not None else None)
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
                last["all"] = (np.array(pts), np.array(infs))
            else:
                outs, infs = sharded.msm_batch([vecs[i % n_distinct] for i in range(k)], mont=False)
                last["xy"], last["inf"] = outs[0], bool(infs[0])

        def sync_all():
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        # Clock ramp (round 4, measured: tools/clock_ramp.py in profiles/r04_experiments.md section 9): after ANY idle of the device --
        # 50 ms are enough -- the first ~30 ms of MSMs run ~10 % slower than the steady state (20 MSMs right after idle: 1.18-1.25
        # ms each, the next 20: 1.12; chunks of 5 after idle: 1.48, 1.33, 1.27, 1.24, 1.22), and the driver's `--steps 20 --warmup 5`
        # puts the whole timed region inside that ramp.  A prover commits back to back for seconds: the steady state is the regime
        # the metric is about, so the device is kept busy with PREHEAT untimed MSMs of the same workload before the W warm-up steps
        # (reported as config.preheat_msms; --no-preheat measures from cold clocks).  The timed region is still exactly K steps.
        # value_cold (round 5): the SAME K steps timed the way rounds 1-3 did -- W warm-up steps after an idle
        # device, no preheat -- reported beside `value` in the same line so that rounds stay comparable whatever the protocol
        elapsed_cold = None
        if not args.no_preheat:
            sync_all()
            time.sleep(0.25)  # idle clocks, as a freshly started process finds them
            run_steps(args.warmup)
            sync_all()
            t0 = time.perf_counter()
            run_steps(args.steps)
            sync_all()
            elapsed_cold = time.perf_counter() - t0
            run_steps(PREHEAT_MSMS)
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
        # the PCIe-inclusive form (`amsm_msm`: the scalars start in pageable host memory like a Rust `&[BigInt]`); never `value`
        h_scalars = scalars.download()
        ffi.check(ctx._lib.amsm_msm(ctx._h, ck._h, 0, _ptr(h_scalars), n, 0, _ptr(out), C.byref(inf)), "amsm_msm")
        t1 = time.perf_counter()
        for _ in range(5):
            ffi.check(ctx._lib.amsm_msm(ctx._h, ck._h, 0, _ptr(h_scalars), n, 0, _ptr(out), C.byref(inf)), "amsm_msm")
        ms_host = (time.perf_counter() - t1) / 5 * 1e3
        # ... and the same host slices handed over back to back, the way the reference's provers call `commit`
        # (src/hp_as/mod.rs:372-385): amsm_msm_batch overlaps the upload of vector v + 1 with MSM v
        ms_host_batch = pipe = plain_rate = witness_rate = ms_dev_batch12 = oneshot = None
        if world == 1:
            oneshot = oneshot_line(ctx, ck, h_scalars, n, out.copy(), bool(inf.value))
            h_vecs = [v.download() for v in vecs]
            VariableBaseMSM.multi_scalar_mul_batch_host(ck, [h_vecs[i % n_distinct] for i in range(3)])
            reps = 12
            # like with like: the device-resident batch of the SAME length (pipeline fill and drain weigh more in 12 MSMs than in K)
            t1 = time.perf_counter()
            VariableBaseMSM.multi_scalar_mul_batch(ck, [vecs[i % n_distinct] for i in range(reps)], mont=False)
            ms_dev_batch12 = (time.perf_counter() - t1) / reps * 1e3
            t1 = time.perf_counter()
            hb_pts, hb_inf = VariableBaseMSM.multi_scalar_mul_batch_host(ck, [h_vecs[i % n_distinct] for i in range(reps)])
            ms_host_batch = (time.perf_counter() - t1) / reps * 1e3
            if "all" in last and not (np.array_equal(hb_pts[:n_distinct], last["all"][0][:n_distinct])):
                raise SystemExit("host-slice batch differs from the device-resident batch")
            del h_vecs
            pipe = ctx.pipeline_stats()
            pipe["two_valued"] = ctx.two_valued_msms()  # 0 here: the timed vectors are uniform
            # the TRUE variable-base rate: no precomputed multiples (what an ark-ec `[patch]` that passes its bases per
            # call gets, or pays ~50 ms of table building per new base set to avoid)
            if not args.no_precompute and args.log2n <= 21:
                ck_plain = CommitterKey.generate(ctx, SEED_POINTS + rank, n, ffi.AMSM_BASES_NO_PRECOMPUTE)
                VariableBaseMSM.multi_scalar_mul_batch(ck_plain, [vecs[i % n_distinct] for i in range(3)], mont=False)
                t1 = time.perf_counter()
                pp, pi = VariableBaseMSM.multi_scalar_mul_batch(ck_plain, [vecs[i % n_distinct] for i in range(12)], mont=False)
                plain_rate = 12 * n / (time.perf_counter() - t1)  # (batches of 12, like the host-slice lines above)
                if "all" in last and not np.array_equal(pp[:4], last["all"][0][:4]):
                    raise SystemExit("plain-key MSM differs from the precomputed-key MSM")
                # ... and over a vector that looks like an R1CS witness: 10 % of the scalars are boolean wires (0 or 1).  ark-ec adds
                # the bases of unit scalars directly; here they are summed apart since round 5 (amsm_ctx_unit_scalar_msms) -- before,
                # all of them landed in bucket 1 of the lowest window and this batch ran 2.3x slower than the uniform one
                hw = vecs[0].download().copy()
                pick = np.random.default_rng(SEED_SCALARS).random(n) < 0.1
                bits = np.zeros_like(hw)
                bits[:, 0] = np.random.default_rng(SEED_SCALARS + 1).integers(0, 2, n)
                hw[pick] = bits[pick]
                wv = ctx.upload(hw)
                VariableBaseMSM.multi_scalar_mul_batch(ck_plain, [wv] * 3, mont=False)
                t1 = time.perf_counter()
                VariableBaseMSM.multi_scalar_mul_batch(ck_plain, [wv] * 12, mont=False)
                witness_rate = 12 * n / (time.perf_counter() - t1)
                wv.free()
                ck_plain.free()

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if world > 1 and elapsed_cold is not None:
        t = torch.tensor([elapsed_cold], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed_cold = float(t.item())
    if world > 1:  # what every rank holds, for the line's `ranks` (a bad curve should be diagnosable from the JSON alone)
        mem = ctx.memory()
        mine = {"rank": rank, "key_bytes": key_bytes(ck, ctx, n), "workspace_bytes": mem["workspace_bytes"],
                "host_pool_threads": int(ctx._lib.amsm_host_threads()), "pipeline_stats": ctx.pipeline_stats()}
        held = [None] * world
        dist.all_gather_object(held, mine)
        for r_, h_ in zip(rank_info, held):
            r_.update({k: v for k, v in h_.items() if k != "rank"})
    pairs_per_step = sum(r["pairs"] for r in rank_info) if rank_info else n
    value_cold = None if elapsed_cold is None else pairs_per_step * args.steps / elapsed_cold
    total_pairs = pairs_per_step * args.steps
    value = total_pairs / elapsed
    ms_per_step = elapsed / args.steps * 1e3

    result = None
    if rank == 0:
        bytes_per_pair = 32 + 16 * ctx.fq_limbs  # 32 B scalar + affine point (64 B Pallas / 96 B BLS12-381)
        dom = "accum_l0"
        dom_ms = stage_ms.get(dom, 0.0)
        achieved = (n * bytes_per_pair / (dom_ms * 1e-3)) / 1e9 if dom_ms > 0 else 0.0
        result = {
            "metric": metric_name(args),
            "value": value,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong" if (args.strong and world > 1) else "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": (f"ONE 2^{args.log2n}-pair {args.curve} MSM per step, point-sharded over {world} GPUs"
                             if (args.strong and world > 1) else f"2^{args.log2n}-pair {args.curve} MSM per GPU") +
                            f" (north_star size; scalars uniform in [0, r), "
                            f"G_i = k_i*G), key resident{'' if args.no_precompute else ' + precomputed window multiples'}",
                "pairs_per_gpu": n,
                "curve": args.curve,
                "precomputed_key": bool(ck.precomputed),
                "parallelism": f"point-sharded x{world}" + (" + one RCCL all-gather of the steps' partial records" if world > 1 else ""),
                "key_setup_s": round(t_key, 3),
                "ms_per_msm_synchronous_call": round(ms_sync, 4),
                "ms_per_msm_host_scalars": round(ms_host, 4),
                "pairs_per_s_host_scalars": round(n / (ms_host * 1e-3), 1),
                # host slices, 12 per call (amsm_msm_batch): PCIe-inclusive like the line above, uploads overlapped
                "ms_per_msm_host_scalars_batch": None if ms_host_batch is None else round(ms_host_batch, 4),
                "pairs_per_s_host_scalars_batch": None if ms_host_batch is None else round(n / (ms_host_batch * 1e-3), 1),
                # the device-resident batch of the same twelve MSMs, and the ratio the host slices reach of it
                "pairs_per_s_device_batch_of_12": None if ms_dev_batch12 is None else round(n / (ms_dev_batch12 * 1e-3), 1),
                "host_slices_fraction_of_device_batch_of_12": None if not (ms_dev_batch12 and ms_host_batch) else round(ms_dev_batch12 / ms_host_batch, 3),
                "host_scalars_page_locked": "not measured: amsm_host_register is a documented no-op since round 5 -- page-locked slices never ran "
                                            "faster than pageable ones and ran up to 27 % slower (profiles/r05_host_slices.md)",
                # no precomputed multiples (one copy of the key, a bucket set per window): the variable-base rate
                "pairs_per_s_plain_key": None if plain_rate is None else round(plain_rate, 1),
                "pairs_per_s_plain_key_witness_10pct_booleans": None if witness_rate is None else round(witness_rate, 1),
                # the ark-ec call shape itself: multi_scalar_mul(&[G], &[BigInt]) with BOTH slices in host memory, nothing resident
                # (amsm_msm_oneshot): PCIe-bound -- its rate against 96 B per pair over the H2D rate measured in this run
                "pairs_per_s_oneshot_host_bases": None if oneshot is None else oneshot.get("pairs_per_s"),
                "oneshot_host_bases": oneshot,
                "key_bytes": key_bytes(ck, ctx, n),
                "window_bits": int(ck.window_bits),
                "pipeline": ("bucket-per-lane (k_prep_local_t + k_accum_bpl; skewed scalars re-run chunked)"
                             if ck.window_bits == 20 else "chunked (k_accum_l0 + k_accum_l1)"),
                "window_widths": "9 x 20 + 4 x 19 bits = 256 (MsmGeom::n_narrow)" if ck.window_bits == 20 else None,
                "pipeline_stats": pipe,
                "rccl_ranks": None if rank_info is None else (len(rank_info) if args.backend == "nccl" else 0),
                "ranks": rank_info,
                "msms_in_flight": 1 if args.sync else 3,
                "preheat_msms": 0 if args.no_preheat else PREHEAT_MSMS,
                # the --no-preheat protocol of rounds 1-3 on the same box, same process (whole-job pairs/s; max over ranks)
                "value_cold": value_cold,
                "host_pool_threads": int(ctx._lib.amsm_host_threads()),
                "collective": None if world == 1 else (f"torch.distributed all_gather_into_tensor of {sharded.engine.record_bytes}-byte records, "
                                                        f"backend {args.backend}" + (f" (RCCL {rccl_version()})" if args.backend == "nccl" else "")),
                "seeds": {"scalars": SEED_SCALARS, "points": SEED_POINTS},
            },
            "roofline": {
                "bound": "hbm",
                "kernel": ("k_accum_bpl" if ck.window_bits == 20 else "k_accum_l0") + " (bucket accumulation)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic(args)[0],
                "traffic_source": pmc_traffic(args)[1],
                "kernel_ms": dom_ms,
                "algorithmic_bytes_per_launch": n * bytes_per_pair,
                # the kernel is integer-VALU bound, not HBM bound: the ceiling that binds is
                # the issue rate of the 10-multiplication mixed addition measured in isolation (tools/fp_bench.hip)
                "alu": alu_roofline(args, ck, n, dom_ms),
                # a ceiling that does not come from this repo's own micro-benchmark: the multiplier's issue rate alone --
                # 1024 SIMDs x 2.4 GHz / 5.2 cycles per wave64 v_mad_u64_u32 x 64 lanes / MADs per mixed addition
                "alu_hw": alu_hw_roofline(args, ck, n, dom_ms),
                # the same kernel with the GPU to itself (one blocking MSM call after the timed region)
                "kernel_ms_unshared": stage_ms_alone.get(dom, 0.0),
                "alu_unshared": alu_roofline(args, ck, n, stage_ms_alone.get(dom, 0.0)),
            },
            # elapsed device time per pipeline stage (hipEvent pairs on the stage's stream): inside the timed batch, where
            # the prep chain of MSM k+1 and the tail of MSM k-1 share the GPU with accumulate L0 of MSM k (so the prep chain
            # STRETCHES to about one L0: it is not queue wait), and for one blocking call with the GPU to itself
            "stage_ms": {"in_batch": {k: round(v, 4) for k, v in stage_ms.items()},
                         "blocking_call": {k: round(v, 4) for k, v in stage_ms_alone.items()}},
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(ctx, ck, vecs, curve_id, args, out.copy(), bool(inf.value), last.get("all"))
        if world == 1 and not args.no_bls and args.log2n == 20 and args.curve == "pallas" and not args.no_precompute:
            # BASELINE config 3's MSM (2^20 pairs, BLS12-381 G1: the 384-bit field path), after and outside the timed region
            for v in vecs:
                v.free()
            ck.free()
            ctx.trim()
            result["config"].update(bls12_381_line(args, check=not args.no_cpu_baseline))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if not args.no_schemes and args.log2n == 20 and args.curve == "pallas":
            if world == 1:
                result["accumulations"] = scheme_rates(cpu_leg_max_log2=args.cpu_leg_max_log2)
            else:
                # the N-GPU form of BASELINE configs 2-5 (strong scaling: the same sizes as the N = 1 line's `accumulations`): ONE
                # process over the N devices through amsm_ctx_create_multi, after the ranks' process group is gone (the other
                # ranks have exited; this rank's key stays resident).  Outside the timed region, bounded, never fails the line.
                result["accumulations_multi_device"] = scheme_rates([0] * world if args.one_gpu else list(range(world)))
        print(json.dumps(result), flush=True)
    return 0


# accumulations/sec: (scheme, log2 size, profile_as options, key suffix, CPU leg) -- BASELINE.json's configs 1-5 in order, plus
# the variants the `inputs` note explains.  CPU leg = log2 size of the SAME harness on the library's host backend
# (`--device -1`: product code, never oracle/): the full size where one run fits the leg's time box on the GPU box's host, else
# the largest size that does -- the entry names the size it ran; nothing is extrapolated.
SCHEME_RUNS = (
    ("trivial_pc_as", 10, ["--reps", "5"], "", {"harness": 10, "n2": 10}),                     # config 1 on the GPU context
    ("trivial_pc_as", 10, ["--reps", "5", "--device", "-1"], "_host_backend", None),           # config 1 as it reads: no GPU
    ("ipa_pc_as", 16, ["--reps", "3"], "", {"harness": 16, "n2": 16}),                         # config 2
    ("ipa_pc_as", 20, ["--reps", "2", "--curve", "1"], "_bls12_381", {"harness": 18, "n2": 18}),  # config 3; CPU at 2^18 (2^20: a minute)
    ("r1cs_nark_as", 18, ["--reps", "3"], "", {"harness": 18, "n2": 18}),                      # config 4
    ("r1cs_nark_as", 18, ["--reps", "3", "--uniform"], "_uniform_witness", {"harness": 18, "n2": 18}),
    # N > 1 only (`accumulations_multi_device`): config 4 with its 2^18-generator keys REPLICATED on every device instead of sharded --
    # the round's independent commitments are dealt to the devices whole, no exchange (amsm.h AMSM_BASES_REPLICATE)
    ("r1cs_nark_as", 18, ["--reps", "3", "--uniform", "--replicate-below", "18"], "_uniform_witness_replicated_keys", None),
    ("r1cs_nark_as", 18, ["--reps", "3", "--replicate-below", "18"], "_replicated_keys", None),
    ("hp_as", 22, ["--reps", "3"], "", {"harness": 20, "n2": 22}),                             # config 5; CPU harness-zk at 2^20
    ("hp_as", 22, ["--reps", "3", "--constant"], "_harness_constant_inputs", {"harness": 20, "n2": 20}),
)
# rough cost of a CPU run (seconds on 16 threads of the GPU box's host, measured in round 6): longest first onto the lanes
CPU_LEG_COST = {("hp_as", 22): 12, ("hp_as", 20): 7, ("ipa_pc_as", 18): 7, ("ipa_pc_as", 16): 1.5, ("r1cs_nark_as", 18): 3, ("trivial_pc_as", 10): 0.1}
CPU_LEG_TIMEOUT_S = 70  # per run; the runs go side by side on disjoint cores (cpu_scheme_rates)


def _json_lines(stdout):
    return [json.loads(line[5:]) for line in stdout.splitlines() if line.startswith("JSON ")]


def _shape_key(r):
    return "harness_1in_2acc_zk" if r["shape"].startswith("harness") else "n2_1in_1acc_nozk"


def _scheme_entry(r):
    rt = r["serialize_roundtrip_decides"]  # null when the harness skipped the check (--no-roundtrip)
    return {"accumulations_per_s": round(r["accumulations_per_s"], 2), "prove_ms": round(r["prove_ms"], 3),
            "verify_ms": round(r["verify_ms"], 3), "decide_ms": round(r["decide_ms"], 3),
            "index_ms": round(r["index_ms"], 1), "zk": r["zk"], "sponge": r["sponge"],
            "curve": "bls12_381_g1" if r.get("curve") == 1 else "pallas",
            "accumulator_bytes": r["accumulator_bytes"],
            "verified": bool(r["verified"] and r["decided"]),
            "serialize_roundtrip": "skipped" if rt is None else bool(rt)}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_scheme_rates(exe, out, max_log2=None):
    """The CPU side of accumulations/sec (the reference's harness times prove / verify / decide on the CPU, examples/scaling-as.rs:94-122):
    every GPU entry of `out` gets a `cpu` sub-object from the SAME profile_as command on the library's host backend (--device -1).  The
    runs are separate processes with AMSM_HOST_THREADS helpers each, one after the other (see below), each bounded by
    CPU_LEG_TIMEOUT_S.  One repetition, no warm-up (--cold: measured within 3 % of a warmed-up repetition when the run has the box
    to itself)."""
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    # one process per (entry, shape), ONE AT A TIME: the host backend's MSMs share the memory system -- four runs of 16 threads side
    # by side on a 2 x 64-core box ran the hp_as 2^22 prove 3.4x slower than the same run alone (6.2-7.4 s against 1.85 s; fourteen
    # side by side: 4-15x; profiles/r06_experiments.md section 3) -- a CPU figure measured under contention would flatter the GPU
    jobs = [(scheme, lg, extra, tag, (cpu[shape] if max_log2 is None else min(cpu[shape], max_log2), shape))
            for scheme, lg, extra, tag, cpu in SCHEME_RUNS if cpu is not None for shape in ("harness", "n2")]
    jobs.sort(key=lambda j: -CPU_LEG_COST.get((j[0], j[4][0]), 5) * (1.5 if j[4][1] == "harness" else 1.0))
    cores = os.cpu_count() or 1
    threads = max(1, min(16, cores))  # caller + helpers per run
    side_by_side = 1
    out["cpu"] = {"backend": "libamsm.so host backend (AMSM_DEVICE_HOST: window-parallel signed-digit Pippenger, vector loops on the host pool; "
                             "product code -- not oracle/, not ark-ec)", "threads_per_run": threads, "runs_side_by_side": side_by_side,
                  "host_cores": cores, "cpu_model": cpu_model(), "repetitions": 1,
                  "note": "`cpu.log2_size` names the size the CPU run used when the full size does not fit the time box; rates are never extrapolated"}

    def run(job):
        scheme, lg, extra, tag, (cpu_lg, shapes) = job
        args = [a for a in extra if a not in ("--device", "-1")]
        args = [a for i, a in enumerate(args) if not (a == "--reps" or (i > 0 and args[i - 1] == "--reps"))]
        cmd = [exe, scheme, str(cpu_lg), str(cpu_lg), "--sponge", "poseidon", "--device", "-1", "--reps", "1", "--cold", "--no-roundtrip",
               "--shape", shapes, *args]
        t0 = time.perf_counter()
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=CPU_LEG_TIMEOUT_S,
                               env=dict(os.environ, AMSM_HOST_THREADS=str(threads - 1)))
            if p.returncode != 0:
                raise RuntimeError(p.stderr[-300:])
            return job, _json_lines(p.stdout), time.perf_counter() - t0, None
        except Exception as e:  # noqa: BLE001
            return job, [], time.perf_counter() - t0, f"{type(e).__name__}: {str(e)[:200]}"

    with ThreadPoolExecutor(max_workers=side_by_side) as ex:
        results = list(ex.map(run, jobs))
    for (scheme, lg, extra, tag, (cpu_lg, shapes)), lines, wall, err in results:
        for shape in ("harness_1in_2acc_zk", "n2_1in_1acc_nozk"):
            key = f"{scheme}_2^{lg}_{shape}{tag}"
            if key not in out or not isinstance(out[key], dict):
                continue
            r = next((x for x in lines if _shape_key(x) == shape), None)
            if r is None:
                if shape.startswith(shapes):
                    out[key]["cpu"] = {"error": err or "no result line", "log2_size": cpu_lg}
                continue
            e = {"log2_size": cpu_lg, "full_size": cpu_lg == lg, "accumulations_per_s": round(r["accumulations_per_s"], 4),
                 "prove_ms": round(r["prove_ms"], 2), "verify_ms": round(r["verify_ms"], 2), "decide_ms": round(r["decide_ms"], 2),
                 "index_ms": round(r["index_ms"], 1), "verified": bool(r["verified"] and r["decided"]), "threads": threads,
                 "run_wall_s": round(wall, 1)}
            if cpu_lg == lg:
                e["gpu_over_cpu_prove"] = round(r["prove_ms"] / out[key]["prove_ms"], 1)
            out[key]["cpu"] = e


def scheme_rates(devices=None, cpu_leg_max_log2=None):
    """accumulations/sec (one `prove` = one accumulation; BASELINE.json's second metric) through the C++ scheme drivers:
    tools/profile_as.cpp, the reference's harness (examples/scaling-as.rs:38-138), at the sizes of BASELINE.json's configs --
    the harness's shape (1 input + the same accumulator twice, zk) and the n_all = 2 no-zk shape -- after the timed region
    and outside it.  Never fails the bench line: an error is reported in place.
    devices (N > 1): the same harness over ONE multi-device context (`profile_as --devices a,b,..`: keys sharded over the devices,
    one exchange of partial records per commit round / grouped MSM / IPA round inside the library) -- Poseidon lines only, each run
    bounded; the first run that does not come back ends the leg."""
    import subprocess
    multi = ["--devices", ",".join(str(d) for d in devices)] if devices else []
    out = {"driver": "C++ (include/amsm_*.hpp) via tools/profile_as.cpp" + (" --devices " + multi[1] if multi else ""),
           "sponge": "poseidon (ark-sponge PoseidonSponge<Fq> as the reference's harness instantiates it, examples/scaling-as.rs; "
                     "parameters restated as recalled: unpinned).  `sha256_standin_*` keys repeat the run on the cheaper "
                     "SHA-256 stand-in sponge: NOT the reference's transcript, shown for the sponge's share only"}
    try:
        exe = os.path.join(ROOT, "build", "profile_as")
        src = os.path.join(ROOT, "tools", "profile_as.cpp")
        libdir = os.path.join(ROOT, "accumulation_amd")
        lib = os.path.join(libdir, "libamsm.so")
        if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(lib)):
            os.makedirs(os.path.dirname(exe), exist_ok=True)
            subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), src, "-o", exe, "-L", libdir,
                                   "-l:libamsm.so", f"-Wl,-rpath,{libdir}", "-Wl,--allow-shlib-undefined"])
        out["inputs"] = ("hp_as: uniform random vectors; `hp_as_..._harness_constant_inputs` repeats it with the reference harness's "
                         "own inputs, vec![rand; n] (src/hp_as/mod.rs:189-190, :991-992): every commitment of such a vector takes the "
                         "two-valued form (v * sum of generators, amsm_ctx_two_valued_msms) instead of a windowed MSM.  r1cs_nark_as: the "
                         "reference's DummyCircuit (src/r1cs_nark_as/mod.rs:1159-1188), whose A z / B z / C z are one value per row: "
                         "two-valued as well; `r1cs_nark_as_..._uniform_witness` repeats it over a circuit whose rows are w_i * w_i = v_i "
                         "with a random witness (A z, B z, C z uniform: the windowed pipelines; not the reference's harness).  The "
                         "`value` of this bench line is measured on uniform random scalars only")
        gave_up = False
        for scheme, lg, extra, tag, _cpu in SCHEME_RUNS:
            if multi and ("--device" in extra or "--curve" in extra):
                continue  # the host backend has no devices to spread over; BASELINE config 3 names ONE GPU
            if not multi and "--replicate-below" in extra:
                continue  # (one device: nothing to replicate on)
            for sponge in ("poseidon", "sha256"):
                if (tag or multi) and sponge == "sha256":
                    continue
                if gave_up:
                    out[f"{scheme}_2^{lg}{tag}"] = {"error": "skipped: an earlier multi-device run did not come back"}
                    continue
                try:
                    # (the stand-in runs report a prove time only: no need to serialise 268 MB of hp_as witness for them)
                    more = (["--no-roundtrip"] if sponge == "sha256" else []) + multi
                    try:
                        p = subprocess.run([exe, scheme, str(lg), str(lg), "--sponge", sponge, *extra, *more], capture_output=True,
                                           text=True, timeout=240 if multi else 600)
                    except subprocess.TimeoutExpired:
                        gave_up = bool(multi)
                        raise
                    if p.returncode != 0:
                        raise RuntimeError(p.stderr[-300:])
                    for r in _json_lines(p.stdout):
                        shape = _shape_key(r)
                        if sponge == "sha256":  # the stand-in: prove time only, never the reported rate
                            out.setdefault("sha256_standin_prove_ms", {})[f"{scheme}_2^{lg}_{shape}{tag}"] = round(r["prove_ms"], 3)
                            continue
                        out[f"{scheme}_2^{lg}_{shape}{tag}"] = _scheme_entry(r)
                except Exception as e:  # noqa: BLE001
                    out[f"{scheme}_2^{lg}{tag}" + ("" if sponge == "poseidon" else "_sha256")] = {"error": f"{type(e).__name__}: {e}"}
        if not multi:
            cpu_scheme_rates(exe, out, cpu_leg_max_log2)
    except Exception as e:  # noqa: BLE001
        out["error"] = f"{type(e).__name__}: {e}"
    return out


def main_single_process(args) -> int:
    """--gpus N --single-process: ONE process, N devices, through amsm_ctx_create_multi.  Weak scaling like the torchrun
    form: 2^log2n pairs per device, the job is one N * 2^log2n-pair MSM per step; every shard's scalars are resident on its
    device; per step the devices' 128-byte partial sums are gathered inside the library."""
    import torch  # noqa: F401  (same HIP runtime initialisation as the torchrun form)

    from accumulation_amd import CommitterKey, MultiContext, ffi

    devices = [int(d) for d in args.devices.split(",")] if args.devices else list(range(args.gpus))
    N = len(devices)
    curve_id = ffi.AMSM_PALLAS if args.curve == "pallas" else ffi.AMSM_BLS12_381_G1
    n = 1 << args.log2n
    ctx = MultiContext(curve_id, devices)
    flags = ffi.AMSM_BASES_NO_PRECOMPUTE if args.no_precompute else ffi.AMSM_BASES_PRECOMPUTE
    t0 = time.time()
    ck = CommitterKey.generate(ctx, SEED_POINTS, n * N, flags)
    t_key = time.time() - t0
    n_distinct = 4
    # vector j, shard g: its own slice of the seed's stream would need an offset; independent streams per (j, g) do as well
    slices = [[ctx.shard(g).random_vector(SEED_SCALARS + 1000 * j + g, ctx.shard_range(ck, g)[1] - ctx.shard_range(ck, g)[0],
                                          mont=False) for g in range(N)] for j in range(n_distinct)]
    ctx.synchronize()

    def run_steps(k):
        return ctx.msm_batch_sharded(ck, [slices[i % n_distinct] for i in range(k)], mont=False)

    run_steps(args.warmup)
    ctx.synchronize()
    t0 = time.perf_counter()
    pts, infs = run_steps(args.steps)
    ctx.synchronize()
    elapsed = time.perf_counter() - t0
    result = {
        "metric": metric_name(args), "value": n * N * args.steps / elapsed,
        "unit": "pairs/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"one {N} x 2^{args.log2n}-pair {args.curve} MSM per step, key sharded over {N} devices of ONE "
                               f"process (amsm_ctx_create_multi), scalars resident per shard",
                   "pairs_per_gpu": n, "curve": args.curve, "devices": devices, "collective": ctx.collective,
                   "parallelism": f"point-sharded x{N}, single process, one host thread per device",
                   "key_setup_s": round(t_key, 3), "seeds": {"scalars": SEED_SCALARS, "points": SEED_POINTS}},
    }
    if not args.no_cpu_baseline and n * N <= (1 << 21):  # bit-exact check of the first step against the CPU restatement
        from oracle import cref
        xy, _ = ck.read(0, n * N)
        sc = np.concatenate([slices[0][g].download() for g in range(N)])
        ref, ref_inf = cref.msm(curve_id, xy, sc, threads=min(os.cpu_count() or 1, 16))
        result["cpu_baseline"] = {"gpu_result_bit_exact_vs_cpu": bool(np.array_equal(ref, pts[0]) and bool(ref_inf) == bool(infs[0])),
                                  "kind": "port", "sample": "step 0 of the timed batch, not timed"}
    print(json.dumps(result), flush=True)
    ctx.close()
    return 0


def oneshot_line(ctx, ck, h_scalars, n, ref_xy, ref_inf):
    """amsm_msm_oneshot on the bench inputs: generators (read back from the key: C-ABI Montgomery limbs, as a Rust `&[G]` holds them)
    and scalars both in pageable host memory, five blocking calls; the result must equal the resident-key MSM's.  Beside it the
    link: one blocking upload of the same 96 B per pair, so that the line says what fraction of the PCIe bound the call reaches."""
    import ctypes as C

    from accumulation_amd import VariableBaseMSM
    from accumulation_amd.engine import _ptr
    try:
        xy, inf = ck.read(0, n)
        assert not inf.any()
        got, ginf = VariableBaseMSM.multi_scalar_mul_oneshot(ctx, xy, h_scalars)  # (sizes the one-shot buffer)
        if not (np.array_equal(got, ref_xy) and bool(ginf) == bool(ref_inf)):
            return {"error": "one-shot MSM differs from the resident-key MSM"}
        t = []
        for _ in range(5):
            t0 = time.perf_counter()
            VariableBaseMSM.multi_scalar_mul_oneshot(ctx, xy, h_scalars)
            t.append(time.perf_counter() - t0)
        ms = sorted(t)[len(t) // 2] * 1e3
        d = C.c_void_p()
        nbytes = xy.nbytes + h_scalars.nbytes
        both = np.concatenate([xy.reshape(-1), h_scalars.reshape(-1)])
        ffi_lib = ctx._lib
        if ffi_lib.amsm_dev_alloc(ctx._h, nbytes, C.byref(d)) != 0:
            return {"pairs_per_s": round(n / (ms * 1e-3), 1), "ms_per_call": round(ms, 4)}
        ffi_lib.amsm_dev_upload(ctx._h, d, _ptr(both), nbytes)
        t = []
        for _ in range(5):
            t0 = time.perf_counter()
            ffi_lib.amsm_dev_upload(ctx._h, d, _ptr(both), nbytes)
            t.append(time.perf_counter() - t0)
        ffi_lib.amsm_dev_free(ctx._h, d)
        up_ms = sorted(t)[len(t) // 2] * 1e3
        return {"pairs_per_s": round(n / (ms * 1e-3), 1), "ms_per_call": round(ms, 4), "bytes_per_pair": nbytes // n,
                "h2d_ms_same_bytes": round(up_ms, 4), "h2d_GB_per_s": round(nbytes / (up_ms * 1e-3) / 1e9, 2),
                "fraction_of_h2d_bound": round(up_ms / ms, 3), "ranges": -(-n // (1 << 19)),
                "note": "pageable host slices; key = none (plain generators imported per call, no precomputed multiples)"}
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}


def bls12_381_line(args, check=True):
    """config.pairs_per_s_bls12_381_2p20: the same batch protocol as `value` (K steps = one amsm_msm_batch_device call after
    preheat + W warm-up steps) on a 2^20-pair BLS12-381 G1 MSM, scalars uniform in [0, r) (0.45 of them above 2^254: the recoding's
    carry reaches a 13th window), every MSM of the batch compared bit for bit with the CPU restatement.  Never fails the line."""
    import ctypes as C

    import torch

    from accumulation_amd import CommitterKey, Context, VariableBaseMSM, ffi
    from accumulation_amd.engine import _ptr
    out = {}
    try:
        n, n_distinct = 1 << 20, 4
        ctx = Context(ffi.AMSM_BLS12_381_G1, device=0)
        t0 = time.time()
        ck = CommitterKey.generate(ctx, SEED_POINTS + 1, n, ffi.AMSM_BASES_PRECOMPUTE)
        t_key = time.time() - t0
        vecs = [ctx.random_vector(SEED_SCALARS + 1000 * j + 7, n, mont=False) for j in range(n_distinct)]

        def run(k):
            return VariableBaseMSM.multi_scalar_mul_batch(ck, [vecs[i % n_distinct] for i in range(k)], mont=False)

        run(16)
        run(args.warmup)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pts, infs = run(args.steps)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        o1 = np.zeros((2 * ctx.fq_limbs,), dtype=np.uint64)
        i1 = C.c_uint8(0)
        t0 = time.perf_counter()
        for _ in range(5):
            ffi.check(ctx._lib.amsm_msm_device(ctx._h, ck._h, 0, vecs[0].ptr, n, 0, _ptr(o1), C.byref(i1)), "amsm_msm_device")
        ms_sync = (time.perf_counter() - t0) / 5 * 1e3
        hv = [v.download() for v in vecs]
        out = {"pairs_per_s_bls12_381_2p20": round(n * args.steps / el, 1), "ms_per_msm_bls12_381_2p20": round(el / args.steps * 1e3, 4),
               "bls12_381_2p20": {"ms_per_msm_synchronous_call": round(ms_sync, 4), "key_setup_s": round(t_key, 3),
                                  "window_bits": int(ck.window_bits), "key_bytes": ck.memory()["table"],
                                  "scalars": "uniform in [0, r)", "scalars_above_2^254": round(float((hv[0][:, 3] >> np.uint64(62)).astype(bool).mean()), 4),
                                  "pipeline_stats": ctx.pipeline_stats(), "steps": args.steps,
                                  "seeds": {"scalars": SEED_SCALARS + 7, "points": SEED_POINTS + 1}}}
        if check:
            from oracle import cref
            xy, _ = ck.read(0, n)
            threads = max(1, min(os.cpu_count() or 1, -(-255 // cref.load().ark_msm_window_bits(n))))
            t0 = time.perf_counter()
            refs = [cref.msm(ffi.AMSM_BLS12_381_G1, xy, h, threads=threads) for h in hv]
            t_cpu = time.perf_counter() - t0
            ok = all(np.array_equal(pts[k], refs[k % n_distinct][0]) and bool(infs[k]) == bool(refs[k % n_distinct][1]) for k in range(len(pts)))
            ok = ok and np.array_equal(o1, refs[0][0])
            out["bls12_381_2p20"].update({"gpu_result_bit_exact_vs_cpu": bool(ok), "timed_batch_msms_checked": len(pts),
                                          "cpu_pairs_per_s": round(n_distinct * n / t_cpu, 1), "cpu_threads": threads, "cpu_kind": "port"})
        for v in vecs:
            v.free()
        ck.free()
        ctx.close()
    except Exception as e:  # noqa: BLE001
        out["bls12_381_2p20_error"] = f"{type(e).__name__}: {e}"
    return out


def metric_name(args):
    """BASELINE.json's metric for the default workload; other sizes / the other curve say so in the name."""
    curve = {"pallas": "Pallas", "bls12_381_g1": "BLS12-381 G1"}.get(args.curve, args.curve)
    return f"MSM throughput (point-scalar pairs/sec) at 2^{args.log2n} {curve}"


ALU_PEAK_GMADD = {"pallas": 18.7, "bls12_381_g1": 6.95}  # isolated xyzz_madd, all SIMDs busy (tools/fp_bench.hip)


def alu_roofline(args, ck, n, kernel_ms):
    """Mixed additions per second of accumulate L0 against the isolated-ALU ceiling of the same formula."""
    if kernel_ms <= 0 or not ck.precomputed:
        return None
    c = ck.window_bits
    if not c:
        return None
    # one gathered mixed addition per non-zero c-bit digit: both scalar fields have 255 bits, so ceil(255 / c) windows
    # hold digits (16 at c = 16, 15 at c = 17); the recoding's carry into one more window is never set for Pallas
    # (r < 2^254 + 2^126) and set for the 0.448 of BLS12-381 scalars that exceed 2^254
    windows = -(-255 // c)
    carry = 0.448 if (args.curve != "pallas" and windows * c == 255) else 0.0
    madds = int(n * (windows + carry))
    achieved = madds / (kernel_ms * 1e-3) / 1e9
    peak = ALU_PEAK_GMADD[args.curve]
    return {"unit": "G mixed-additions/s", "achieved": achieved, "peak": peak, "frac": achieved / peak,
            "madds_per_launch": madds, "window_bits": c}


MADS_PER_MADD = {"pallas": 7 * 126 + 2 * 90 + 207, "bls12_381_g1": 7 * 392 + 2 * 301 + 588}  # fpu.h: mul / sqr / fused pair


def alu_hw_roofline(args, ck, n, kernel_ms):
    a = alu_roofline(args, ck, n, kernel_ms)
    if a is None:
        return None
    peak = 1024 * 2.4e9 / 5.2 * 64 / MADS_PER_MADD[args.curve] / 1e9
    return {"unit": "G mixed-additions/s", "achieved": a["achieved"], "peak": round(peak, 2), "frac": a["achieved"] / peak,
            "mads_per_mixed_addition": MADS_PER_MADD[args.curve],
            "basis": "v_mad_u64_u32 issue only: 5.2 cycles per wave64 per SIMD (tools/ubench_valu.hip), 1024 SIMDs, 2.4 GHz nominal"}


def rccl_version():
    try:
        import torch
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception as e:  # noqa: BLE001
        return f"unknown ({type(e).__name__})"


def key_bytes(ck, ctx, n):
    """HBM held by the committer key, as the library reports it (amsm_bases_memory): the table's W levels of affine points
    and the 17-bit twin of a 20-bit key -- built only when a range below a quarter of the key or a skewed vector needs it:
    0 here unless the run above did"""
    c = int(ck.window_bits)
    levels = (255 // c + 1) if (ck.precomputed and c) else 1
    m = ck.memory()
    t = ck.tables()
    return {"table": m["table"], "levels": levels, "twin_17_bit_table": m["twin"], "window_table": t["window_table"],
            "direct_sum_table": t["direct_sum_table"], "direct_sum_table_denied": t["direct_sum_table_denied"],
            "twin_denied": t["twin_denied"], "tables_denied_on_context": ctx.tables_denied()}


def source_hash():
    """SHA-256 over the kernel sources the library is built from (the PMC summaries carry it: a stale one is flagged)"""
    import glob
    import hashlib
    hh = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "accumulation_amd", "csrc", "*"))):
        if os.path.isfile(f):
            hh.update(os.path.basename(f).encode())
            hh.update(open(f, "rb").read())
    return hh.hexdigest()[:16]


def pmc_traffic(args):
    """(HBM bytes per launch, provenance) of the dominant kernel from the committed rocprofv3 PMC passes
    (tools/profile_round.sh -> profiles/*_pmc_accum_l0.json; FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for 16-B-per-lane loads).  Only valid for the default workload; null otherwise."""
    if args.log2n != 20 or args.curve != "pallas" or args.no_precompute:
        return None, None
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_accum_*.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        same = d.get("source_hash") == source_hash()
        return d["hbm_traffic_bytes_per_launch"], os.path.relpath(files[-1], ROOT) + \
            " (committed rocprofv3 --pmc passes of this workload; not re-measured in this run; kernel sources " + \
            ("UNCHANGED since it was collected)" if same else "CHANGED since it was collected: stale)")
    except Exception:
        return None, None


def cpu_baseline(ctx, ck, vecs, curve_id, args, gpu_out, gpu_inf, batch):
    """Time the plain-C ark-ec-style restatement on the host cores, on the same inputs, and check the GPU results against
    it bit-for-bit: the blocking call, the host-scalar call and EVERY MSM of the timed batch (step k used vector k % 4: four
    CPU MSMs cover them all).  (Only this leg of bench.py touches oracle/.)"""
    from oracle import cref

    cpu_log2n = args.cpu_log2n if args.cpu_log2n is not None else args.log2n
    m = min(1 << cpu_log2n, len(ck))
    xy, _ = ck.read(0, m)
    scalars = vecs[0]
    sc = scalars.download()[:m]
    cores = os.cpu_count() or 1
    n_windows = -(-255 // cref.load().ark_msm_window_bits(m))
    threads = max(1, min(cores, n_windows))
    t0 = time.perf_counter()
    cpu_out, cpu_inf = cref.msm(curve_id, xy, sc, threads=threads)
    t_par = time.perf_counter() - t0
    n_par = 1
    match = None
    checked = 0
    if m == len(ck):
        match = bool(np.array_equal(cpu_out, gpu_out) and cpu_inf == gpu_inf)
        if batch is not None:
            refs = [(cpu_out, cpu_inf)]
            for v in vecs[1:]:  # the other vectors of the timed batch: timed as well, the sample is all of them
                host_v = v.download()[:m]
                t0 = time.perf_counter()
                refs.append(cref.msm(curve_id, xy, host_v, threads=threads))
                t_par += time.perf_counter() - t0
                n_par += 1
            pts, infs = batch
            for k in range(len(pts)):
                r_out, r_inf = refs[k % len(vecs)]
                match = match and bool(np.array_equal(pts[k], r_out) and bool(infs[k]) == bool(r_inf))
                checked += 1
    # single-thread figure (the reference's default features) on a bounded sample: 2^16-pair MSMs for ~6 s
    ms = min(m, 1 << 16)
    t_one, n_one = 0.0, 0
    while t_one < 6.0 and n_one < 64:
        lo = (n_one * ms) % max(m - ms + 1, 1)
        t0 = time.perf_counter()
        cref.msm(curve_id, xy[lo:lo + ms], sc[lo:lo + ms], threads=1)
        t_one += time.perf_counter() - t0
        n_one += 1
    return {
        "value": n_par * m / t_par,
        "unit": "pairs/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{n_par} x 2^{cpu_log2n}-pair MSM on the bench inputs (the distinct scalar vectors of the timed batch), window-parallel on "
                  f"{threads} threads (ark-ec `parallel` semantics; host has {cores} cores); {t_par:.2f} s",
        "single_thread_value": n_one * ms / t_one,
        "single_thread_sample": f"{n_one} x 2^{ms.bit_length() - 1}-pair MSM, 1 thread (reference default features); {t_one:.2f} s",
        "gpu_result_bit_exact_vs_cpu": match,
        "timed_batch_msms_checked": checked,
    }


if __name__ == "__main__":
    sys.exit(main())
