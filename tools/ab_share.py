"""A/B of the shared bucket set (AMSM_SHARE_BUCKETS, round 5): MSMs of 2^lg pairs over the 20-bit key as ranges over ONE bucket set
against independent ranges.  One fresh PROCESS per measurement, alternating (a second context inside one process gets other hardware
queues and runs a third slower: tools/second_context_streams.py), same box.  Prints M pairs/s in batches and ms per blocking call.
    python tools/ab_share.py [lg=22] [reps=12] [rounds=3]      Not a test."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from accumulation_amd import CommitterKey, Context, VariableBaseMSM, ffi
    lg, reps = int(sys.argv[2]), int(sys.argv[3])
    n = 1 << lg
    ctx = Context(ffi.AMSM_PALLAS)
    ck = CommitterKey.generate(ctx, 0x5EED1001, n)
    vecs = [ctx.random_vector(10 + j, n, mont=True) for j in range(2)]
    for _ in range(2):
        VariableBaseMSM.multi_scalar_mul_batch(ck, [vecs[i % 2] for i in range(reps)], mont=True)
    best = 1e9
    for _ in range(3):
        ctx.synchronize()
        t = time.perf_counter()
        out, _ = VariableBaseMSM.multi_scalar_mul_batch(ck, [vecs[i % 2] for i in range(reps)], mont=True)
        best = min(best, (time.perf_counter() - t) / reps)
    t = time.perf_counter()
    for i in range(4):
        VariableBaseMSM.multi_scalar_mul(ck, vecs[i % 2], mont=True)
    db = (time.perf_counter() - t) / 4
    import hashlib
    print(f"share={os.environ.get('AMSM_SHARE_BUCKETS', '1')} 2^{lg}: batch {n/best/1e6:.0f} M pairs/s ({best*1e3:.3f} ms per MSM), blocking "
          f"{db*1e3:.3f} ms, shared sets {ctx.pipeline_stats()['shared_bucket_sets']}, result {hashlib.sha256(out[0].tobytes()).hexdigest()[:12]}", flush=True)
    sys.exit(0)
lg = sys.argv[1] if len(sys.argv) > 1 else "22"
reps = sys.argv[2] if len(sys.argv) > 2 else "12"
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
for r in range(rounds):
    for share in ("1", "0"):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "child", lg, reps], env=dict(os.environ, AMSM_SHARE_BUCKETS=share),
                           capture_output=True, text=True)
        print(f"round {r}", (p.stdout.strip().splitlines() or [p.stderr[-300:]])[-1], flush=True)
