"""Randomised differential test of round 3's accumulation pipelines against the CPU restatement (longer than pytest wants to be):
  * keys of 2^20 (sometimes 2^21) generators, precomputed for 20-bit windows: random (base_off, n) ranges around the limits of the
    bucket-per-lane pipeline (2^19, 2^20], scalar vectors from uniform to constant through every mixture in between (a random
    share of equal values, few distinct values, small ranges, top-of-field values, sparse), device vectors, batches, host slices;
  * grouped MSMs and (AMSM_BPS=2 contexts) plain MSMs of 2^16 .. 2^17 pairs through the bucket-split pipeline, both curves;
  * contexts with the skew probe on and off (the overflow re-run on its own);
  * round 6 -- LONG MSMs over ONE bucket set (amsm_ctx_shared_bucket_msms: more pairs than the 2^20-pair window of a 20-bit key):
    keys of 2^21 generators, (2^20, 2^21] pairs, vectors whose RANGES differ -- uniform / skewed / witness-like (a share of boolean
    wires) per range, so that a skewed range sends the whole MSM back to independent ranges --, device vectors, batches and host
    slices, single-device keys and keys sharded over two "devices" (2^22 generators: every shard runs its own shared set).
Usage: python tools/fuzz_pipelines.py [seconds] [seed] [--long-share P]   (P: share of iterations spent on the long MSMs, default 0.2)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402

from accumulation_amd import CommitterKey, Context, MultiContext, VariableBaseMSM  # noqa: E402
from oracle import cref, pyref as o  # noqa: E402
from tests import helpers as h  # noqa: E402

LONG_SHARE = 0.2
if "--long-share" in sys.argv:
    i = sys.argv.index("--long-share")
    LONG_SHARE = float(sys.argv[i + 1])
    del sys.argv[i:i + 2]
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rs = np.random.RandomState(seed)
t_end = time.time() + budget
THREADS = min(os.cpu_count() or 1, 20)


def ctx_with(curve_id, **env):
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return Context(curve_id)
    finally:
        for k in env:
            del os.environ[k]


def scalars(c, n):
    base = cref.rng_scalars(int(rs.randint(1 << 30)), n)
    kind = rs.choice(["uniform", "uniform", "share_equal", "few", "small_range", "top", "sparse", "constant", "one_window",
                      "two_valued_with_exceptions"])
    if kind == "share_equal":
        share = float(rs.choice([1e-5, 1e-4, 1e-3, 0.01, 0.1, 0.5, 0.9]))
        base[rs.rand(n) < share] = base[0]
    elif kind == "few":
        k = int(rs.choice([1, 2, 3, 16, 200, 5000]))
        base = base[rs.randint(0, k, size=n)]
    elif kind == "small_range":
        bits = int(rs.choice([1, 8, 16, 19, 20, 21, 40, 64]))
        out = np.zeros_like(base)
        out[:, 0] = base[:, 0] & np.uint64((1 << bits) - 1)
        base = out
    elif kind == "top":
        sp = h.scalars_to_np([c.r - 1, c.r - 2, 1 << 254, (1 << 254) - 1, 1, 0, (1 << 128) - 1, (1 << 240) - 1, 1 << 240])
        m = rs.rand(n) < float(rs.choice([0.001, 0.05, 1.0]))
        base[m] = sp[rs.randint(0, len(sp), size=int(m.sum()))]
    elif kind == "sparse":
        base[rs.rand(n) >= float(rs.choice([0.01, 0.2]))] = 0
    elif kind == "constant":
        base[:] = base[0]
    elif kind == "two_valued_with_exceptions":  # 0 / v with 0 .. 12 other values anywhere (<= 8: the two-valued form's exceptions)
        others = base[1:13].copy()
        v = base[0].copy()
        base[:] = v
        base[rs.rand(n) < float(rs.choice([0.0, 0.3, 0.999]))] = 0
        k = int(rs.randint(0, 13))
        if k:
            pos = rs.randint(0, n, size=k)
            if rs.rand() < 0.3:
                pos[: min(k, 3)] = [0, 1, n - 1][: min(k, 3)]
            base[pos] = others[:k]
    elif kind == "one_window":
        w = int(rs.randint(0, 12))
        limb, sh = (20 * w) // 64, (20 * w) % 64
        width = min(20, 64 - sh)
        base[:, limb] = (base[:, limb] & ~np.uint64(((1 << width) - 1) << sh)) | np.uint64((int(rs.randint(1, 1 << width)) & ((1 << width) - 1)) << sh)
    return base, kind


def ranged_scalars(c, n):
    """a vector of more than 2^20 scalars whose 2^20-pair ranges are drawn apart: uniform, one of the skewed kinds of scalars(), or
    witness-like (uniform values with a share of boolean wires)"""
    parts, kinds = [], []
    for lo in range(0, n, 1 << 20):
        m = min(1 << 20, n - lo)
        u = rs.rand()
        if u < 0.55:
            v, kind = cref.rng_frs(c.curve_id, int(rs.randint(1 << 30)), m), "uniform"
        elif u < 0.75:
            v = cref.rng_frs(c.curve_id, int(rs.randint(1 << 30)), m)
            share = float(rs.choice([0.01, 0.1, 0.5, 0.9]))
            pick = rs.rand(m) < share
            bits = np.zeros_like(v)
            bits[:, 0] = rs.randint(0, 2, size=m)
            v[pick] = bits[pick]
            kind = f"witness_{share}"
        else:
            v, kind = scalars(c, m)
        parts.append(v)
        kinds.append(kind)
    return np.concatenate(parts), "+".join(kinds)


def check(c, xy, sc, got, inf, off, what):
    n = min(len(sc), len(xy) - off)
    ref, rinf = cref.msm(c.curve_id, xy[off:off + n], sc[:n], threads=THREADS)
    if bool(inf) != bool(rinf) or not np.array_equal(got, ref):
        print("MISMATCH", c.name, what, "off", off, "n", n, "seed", seed, flush=True)
        sys.exit(1)


n_cases = n_big = n_small = n_long = 0
stats = {}
while time.time() < t_end:
    c = o.PALLAS if rs.rand() < 0.75 else o.BLS12_381_G1
    if rs.rand() < LONG_SHARE:  # ---- more pairs than the key's window: ONE bucket set for all ranges (round 5's Share; round 6's soak)
        sharded = rs.rand() < 0.3
        probe = int(rs.rand() < 0.7)
        os.environ["AMSM_BPL_PROBE"] = str(probe)
        try:
            ctx = MultiContext(c.curve_id, [0, 0]) if sharded else Context(c.curve_id)
        finally:
            del os.environ["AMSM_BPL_PROBE"]
        kn = (1 << 22) if sharded else (1 << 21)
        ck = CommitterKey.generate(ctx, int(rs.randint(1 << 30)), kn)
        xy, _ = ck.read()
        for _ in range(int(rs.randint(2, 4))):
            n = kn if (sharded or rs.rand() < 0.5) else int(rs.randint((1 << 20) + 1, kn + 1))
            off = 0 if sharded else int(rs.randint(0, kn - n + 1))
            mode = rs.choice(["device", "host", "batch"]) if not sharded else rs.choice(["device", "host"])
            if mode in ("device", "host"):
                sc, kind = ranged_scalars(c, n)
                arg = ctx.upload(sc) if mode == "device" else sc
                got, inf = VariableBaseMSM.multi_scalar_mul(ck, arg, base_off=off)
                check(c, xy, sc, got, inf, off, f"long {mode} {kind} probe={probe} sharded={sharded}")
                n_cases += 1
                n_long += 1
            else:
                vecs = [ranged_scalars(c, n) for _ in range(int(rs.randint(2, 4)))]
                pts, infs = VariableBaseMSM.multi_scalar_mul_batch(ck, [ctx.upload(v) for v, _ in vecs], mont=False, base_off=off)
                for j, (v, kind) in enumerate(vecs):
                    check(c, xy, v, pts[j], infs[j], off, f"long batch[{j}] {kind} probe={probe}")
                    n_cases += 1
                    n_long += 1
        for g in range(2 if sharded else 1):
            st = (ctx.shard(g) if sharded else ctx).pipeline_stats()
            for k, v in st.items():
                stats[k] = stats.get(k, 0) + v
        ck.free()
        ctx.close()
    elif rs.rand() < 0.6:  # ---- 2^20 / 2^21-generator key, bucket-per-lane range and around it
        probe = int(rs.rand() < 0.6)
        ctx = ctx_with(c.curve_id, AMSM_BPL_PROBE=probe)
        kn = (1 << 20) if rs.rand() < 0.8 else (1 << 21)
        ck = CommitterKey.generate(ctx, int(rs.randint(1 << 30)), kn)
        xy, _ = ck.read()
        for _ in range(int(rs.randint(2, 5))):
            n = int(rs.choice([1 << 20, (1 << 19) + 1, 1 << 19, (1 << 20) - int(rs.randint(1, 5000)), int(rs.randint(1 << 19, (1 << 20) + 1)),
                               int(rs.randint(1, 1 << 19)), kn]))
            n = min(n, kn)
            off = int(rs.randint(0, kn - n + 1))
            mode = rs.choice(["device", "host", "batch", "host_batch"])
            if mode in ("device", "host"):
                sc, kind = scalars(c, n)
                arg = ctx.upload(sc) if mode == "device" else sc
                got, inf = VariableBaseMSM.multi_scalar_mul(ck, arg, base_off=off)
                check(c, xy, sc, got, inf, off, f"{mode} {kind} probe={probe}")
                n_cases += 1
            else:
                vecs = [scalars(c, n) for _ in range(int(rs.randint(2, 6)))]
                if mode == "batch":
                    pts, infs = VariableBaseMSM.multi_scalar_mul_batch(ck, [ctx.upload(v) for v, _ in vecs], mont=False, base_off=off)
                else:
                    pts, infs = VariableBaseMSM.multi_scalar_mul_batch_host(ck, [v for v, _ in vecs], base_off=off)
                for j, (v, kind) in enumerate(vecs):
                    check(c, xy, v, pts[j], infs[j], off, f"{mode}[{j}] {kind} probe={probe}")
                    n_cases += 1
            n_big += 1
        for k, v in ctx.pipeline_stats().items():
            stats[k] = stats.get(k, 0) + v
        ck.free()
        ctx.close()
    else:  # ---- small MSMs: bucket-split pipeline (grouped by default, everything with AMSM_BPS=2)
        bps = int(rs.choice([1, 2]))
        probe = int(rs.rand() < 0.5)
        ctx = ctx_with(c.curve_id, AMSM_BPS=bps, AMSM_BPL_PROBE=probe)
        kn = int(rs.choice([1 << 16, 1 << 17, (1 << 16) + 777, 100000]))
        ck = CommitterKey.generate(ctx, int(rs.randint(1 << 30)), kn)
        xy, _ = ck.read()
        for _ in range(int(rs.randint(2, 6))):
            n = int(rs.choice([kn, 1 << 16, int(rs.randint(1 << 15, kn + 1))]))
            n = min(n, kn)
            off = int(rs.randint(0, kn - n + 1))
            sc, kind = scalars(c, n)
            if rs.rand() < 0.5:
                got, inf = VariableBaseMSM.multi_scalar_mul(ck, ctx.upload(sc), base_off=off)
                check(c, xy, sc, got, inf, off, f"small {kind} bps={bps} probe={probe}")
            else:
                shift = int(rs.randint(0, 16))
                pts, infs = VariableBaseMSM.multi_scalar_mul_grouped(ck, ctx.upload(sc), shift, mont=False, base_off=off)
                for g in (0, 1):
                    sel = sc.copy()
                    sel[((np.arange(n) >> shift) & 1) != g] = 0
                    check(c, xy, sel, pts[g], infs[g], off, f"grouped[{g}] shift={shift} {kind} bps={bps}")
            n_cases += 1
            n_small += 1
        for k, v in ctx.pipeline_stats().items():
            stats[k] = stats.get(k, 0) + v
        ck.free()
        ctx.close()
print(f"fuzz_pipelines: {n_cases} MSMs checked ({n_big} calls on 2^20+ keys, {n_small} small, {n_long} LONG MSMs of > 2^20 pairs), all bit-exact; seed {seed}; pipelines {stats}", flush=True)
