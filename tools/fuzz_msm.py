"""Randomised differential test of the MSM entry points against the CPU oracle (longer than the pytest suite wants to be):
random sizes, window overrides, key kinds, base offsets, scalar distributions (uniform / few distinct / sparse / boolean-heavy witness / top of
the field), single / batch / multi / grouped calls, both curves.  Usage: python tools/fuzz_msm.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402

from accumulation_amd import CommitterKey, Context, VariableBaseMSM  # noqa: E402
from oracle import cref, pyref as o  # noqa: E402
from tests import helpers as h  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rs = np.random.RandomState(seed)
t_end = time.time() + budget
n_cases = 0
ctxs = {c.name: Context(c.curve_id) for c in (o.PALLAS, o.BLS12_381_G1)}
BIG = 1 << int(os.environ.get("FUZZ_BIG_LOG2", "18"))  # a few cases per minute at sizes where skewed scalars make heavy prep partitions and keys fold in batches
pools = {c.name: cref.rng_points(c.curve_id, 1000 + seed, BIG) for c in (o.PALLAS, o.BLS12_381_G1)}
n_big = n_fold = n_adv = n_bpl = 0
# round 4: share of cases that exercise the bucket-per-lane pipeline's new geometries -- plain keys of 2^16 .. 2^20 pairs (14- /
# 15- / 16-bit windows whose widths add up to 256 bits, the top window split over two sets), ranges and grouped MSMs over a
# 20-bit precomputed key (FUZZ_BIG_LOG2=20) -- on mostly uniform scalars with edge values sprinkled in
BPL_FRACTION = float(os.environ.get("FUZZ_BPL_FRACTION", "0.0"))
PALLAS_FRACTION = float(os.environ.get("FUZZ_PALLAS_FRACTION", "0.7"))  # the rest of the cases run on BLS12-381
# round 4: share of cases on the direct sum (precomputed keys of up to 2^15 generators, automatic window: DESIGN.md 4.2g) -- every
# call form, grouped MSMs at lengths that are / are not multiples of 512, every scalar distribution, adversarial points in the key
DIRECT_FRACTION = float(os.environ.get("FUZZ_DIRECT_FRACTION", "0.0"))
n_direct = 0
# points with extreme coordinates in the device's internal Montgomery radix (tests/golden/adversarial_points.json) and their
# negatives: mixed into a third of the keys, with repetitions, so that equal / opposite / edge-valued operands meet in buckets
import json  # noqa: E402
_fix = json.load(open(os.path.join(ROOT, "tests", "golden", "adversarial_points.json")))
adv = {}
for _c in (o.PALLAS, o.BLS12_381_G1):
    _pts = [(int(x, 16), int(y, 16)) for v in _fix["curves"][_c.name].values() for x, y in v]
    _pts += [(P[0], (-P[1]) % _c.p) for P in _pts]
    adv[_c.name] = h.points_to_np(_c, _pts)[0]


def scalars(c, n, kind):
    base = cref.rng_scalars(int(rs.randint(1 << 30)), n)
    if kind == "uniform":
        return base
    if kind == "few":
        k = min(n, int(rs.randint(1, 5)))
        return base[rs.randint(0, k, size=n)]
    if kind == "sparse":
        out = np.zeros_like(base)
        m = rs.rand(n) < 0.05
        out[m] = base[m]
        return out
    if kind == "witness":  # uniform values with a share of boolean wires (round 5: their unit scalars are summed apart)
        frac = float(rs.choice([0.01, 0.1, 0.5, 0.9]))
        m = rs.rand(n) < frac
        out = base.copy()
        bits = np.zeros_like(base)
        bits[:, 0] = rs.randint(0, 2, size=n) if rs.rand() < 0.7 else 1
        out[m] = bits[m]
        return out
    if kind == "top":
        sp = h.scalars_to_np([c.r - 1, c.r - 2, 1 << 254, (1 << 254) - 1, 1, 0, (1 << 128) - 1])
        return sp[rs.randint(0, len(sp), size=n)]
    raise ValueError(kind)


while time.time() < t_end:
    c = o.PALLAS if rs.rand() < PALLAS_FRACTION else o.BLS12_381_G1
    ctx = ctxs[c.name]
    roll = rs.rand()
    if rs.rand() < BPL_FRACTION:
        n_key = int(rs.randint((1 << 16) + 1, BIG + 1)) if rs.rand() < 0.7 else BIG
        flags = int(rs.choice([1, 2, 2]))
        xy = pools[c.name][:n_key]
        ck = CommitterKey.load(ctx, xy, None, flags)
        off = int(rs.randint(0, n_key // 2)) if rs.rand() < 0.3 else 0
        n = n_key - off if rs.rand() < 0.5 else int(rs.randint((n_key - off) // 4 + 1, n_key - off + 1))
        sc = scalars(c, n, "witness") if rs.rand() < 0.3 else cref.rng_scalars(int(rs.randint(1 << 30)), n)
        k = int(rs.randint(0, 40))
        if k:
            edge = h.scalars_to_np([c.r - 1, c.r - 2, (1 << 254) - 1 if c.r > (1 << 254) else (1 << 253), 1, 0, 2, (1 << 128) - 1,
                                    (c.r - 1) // 2, (c.r + 1) // 2, 1 << 240, (1 << 240) - 1, 1 << 224, (1 << 15) + 1, 1 << 19])
            sc[rs.randint(0, n, size=k)] = edge[rs.randint(0, len(edge), size=k)]
        tag = ("bpl", c.name, n_key, flags, off, n)
        if rs.rand() < 0.7:
            out, inf = VariableBaseMSM.multi_scalar_mul(ck, ctx.upload(sc), base_off=off)
            ref, rinf = cref.msm(c.curve_id, xy[off:off + n], sc, threads=8)
            assert bool(inf) == bool(rinf) and np.array_equal(out, ref), tag
        else:
            shift = int(rs.randint(0, 20))
            outs, infs = VariableBaseMSM.multi_scalar_mul_grouped(ck, ctx.upload(sc), shift, mont=False, base_off=off)
            cls = (np.arange(n) >> shift) & 1
            for g in (0, 1):
                ref, rinf = cref.msm(c.curve_id, xy[off:off + n][cls == g], sc[cls == g], threads=8) if (cls == g).any() else (None, True)
                assert bool(infs[g]) == bool(rinf) and (rinf or np.array_equal(outs[g], ref)), (tag, shift, g)
        ck.free()
        n_bpl += 1
        n_cases += 1
        continue
    if rs.rand() < DIRECT_FRACTION:
        n_key = int(rs.choice([1, 2, 3, 17, 255, 256, 511, 512, 513, 1000, 1024, 4097, 8192, 20000, 32768])) if rs.rand() < 0.7 \
            else int(rs.randint(1, (1 << 15) + 1))
        xy = pools[c.name][:n_key]
        if rs.rand() < 0.35:
            xy = xy.copy()
            k = int(rs.randint(1, min(n_key, 24) + 1))
            xy[rs.randint(0, n_key, size=k)] = adv[c.name][rs.randint(0, len(adv[c.name]), size=k)]
            n_adv += 1
        ck = CommitterKey.load(ctx, xy, None, 1)
        kind = str(rs.choice(["uniform", "few", "sparse", "top", "witness"]))
        mode = str(rs.choice(["single", "batch", "multi", "grouped", "grouped"]))
        off = int(rs.randint(0, n_key)) if rs.rand() < 0.3 else 0
        n = int(rs.randint(1, n_key - off + 1))
        tag = ("direct", c.name, n_key, kind, mode, off, n)
        before = ctx.pipeline_stats()["direct_sum"]
        expect = None
        if mode == "single":
            sc = scalars(c, n, kind)
            out, inf = VariableBaseMSM.multi_scalar_mul(ck, ctx.upload(sc) if rs.rand() < 0.5 else sc, base_off=off)
            ref, rinf = cref.msm(c.curve_id, xy[off:off + n], sc, threads=4)
            assert inf == rinf and np.array_equal(out, ref), tag
            expect = 1 if n < (1 << 14) else None  # (from 2^14 pairs the two-valued shortcut may take a "few" / "top" vector)
        elif mode == "batch":
            vs = [scalars(c, n, kind) for _ in range(int(rs.randint(1, 6)))]
            outs, infs = VariableBaseMSM.multi_scalar_mul_batch(ck, [ctx.upload(v) for v in vs], mont=False, base_off=off)
            for j, v in enumerate(vs):
                ref, rinf = cref.msm(c.curve_id, xy[off:off + n], v, threads=4)
                assert bool(infs[j]) == rinf and np.array_equal(outs[j], ref), (tag, j)
            expect = len(vs) if n < (1 << 14) else None
        elif mode == "multi":
            jobs = []
            for _ in range(int(rs.randint(1, 5))):
                o2 = int(rs.randint(0, n_key))
                n2 = int(rs.randint(1, n_key - o2 + 1))
                jobs.append((o2, scalars(c, n2, kind)))
            outs, infs = VariableBaseMSM.multi_scalar_mul_multi(ck, [(o2, ctx.upload(v)) for o2, v in jobs], mont=False)
            for j, (o2, v) in enumerate(jobs):
                ref, rinf = cref.msm(c.curve_id, xy[o2:o2 + len(v)], v, threads=4)
                assert bool(infs[j]) == rinf and np.array_equal(outs[j], ref), (tag, j)
        else:
            if rs.rand() < 0.7 and n_key - off >= 512:
                n = 512 * int(rs.randint(1, (n_key - off) // 512 + 1))  # lengths most shifts divide
            sc = scalars(c, n, kind)
            shift = int(rs.randint(0, 15))
            outs, infs = VariableBaseMSM.multi_scalar_mul_grouped(ck, ctx.upload(sc), shift, mont=False, base_off=off)
            idx = np.arange(n)
            for g in (0, 1):
                m = sc.copy()
                m[((idx >> shift) & 1) != g] = 0
                ref, rinf = cref.msm(c.curve_id, xy[off:off + n], m, threads=4)
                assert bool(infs[g]) == rinf and np.array_equal(outs[g], ref), (tag, shift, g)
            expect = 1 if n % (2 << shift) == 0 else 0
        if expect is not None:
            assert ctx.pipeline_stats()["direct_sum"] - before == expect, (tag, expect)
        ck.free()
        n_direct += 1
        n_cases += 1
        continue
    if roll < 0.02:
        # large and skewed: constant / few-valued / mostly-constant scalars over >= 2^17 points (heavy prep partitions,
        # batched affine conversion of the precomputed levels)
        n = int(rs.randint(1 << 17, BIG + 1))
        xy = pools[c.name][:n]
        ck = CommitterKey.load(ctx, xy, None, int(rs.choice([1, 2])))
        sc = scalars(c, n, str(rs.choice(["few", "uniform", "top"])))
        if rs.rand() < 0.5:
            step = int(rs.randint(2, 50))  # a uniform fraction mixed in
            sc[::step] = cref.rng_scalars(int(rs.randint(1 << 30)), n)[::step]
        out, inf = VariableBaseMSM.multi_scalar_mul(ck, sc)
        ref, rinf = cref.msm(c.curve_id, xy, sc, threads=8)
        assert inf == rinf and np.array_equal(out, ref), ("big", c.name, n)
        ck.free()
        n_big += 1
        n_cases += 1
        continue
    if roll < 0.05:
        # key fold l + x r (NAF ladder; batched conversion above 2^17 points) through MSM linearity:
        # msm(fold(key), s) == msm(key, [s ; x s])
        from accumulation_amd.scalar_field import Fr
        fr = Fr(ctx.curve)
        half = int(rs.choice([1, 2, 7, 64, 1000, 5000, (1 << 17) - 1, 1 << 17]))
        if rs.rand() < 0.3:
            half = int(rs.randint(1, BIG // 2 + 1))
        xy = pools[c.name][: 2 * half]
        ck = CommitterKey.load(ctx, xy, None, int(rs.choice([1, 2])))
        x = int(o.rng_scalar(int(rs.randint(1 << 30)), 0)) % c.r
        nbits = 255
        if rs.rand() < 0.6:
            x %= 1 << 128
            nbits = 128
        if rs.rand() < 0.1:
            x = int(rs.choice([0, 1, 2, 3])) if nbits == 128 else c.r - int(rs.randint(1, 4))
        folded = ck.fold(half, fr.to_limbs(x), nbits)
        base = [int(o.rng_scalar(int(rs.randint(1 << 30)), i)) % c.r for i in range(16)]
        s_half = (base * (half // 16 + 1))[:half]
        got, ginf = VariableBaseMSM.multi_scalar_mul(folded, ctx.upload(h.scalars_to_np(s_half)), mont=False)
        s_full = s_half + [(v * x) % c.r for v in s_half]
        sc_full = h.scalars_to_np(s_full)
        if 2 * half <= 40000:  # small enough for the CPU oracle: pins the identity's right-hand side too
            exp, einf = cref.msm(c.curve_id, xy, sc_full, threads=4)
        else:
            exp, einf = VariableBaseMSM.multi_scalar_mul(ck, ctx.upload(sc_full), mont=False)
        assert bool(ginf) == bool(einf) and np.array_equal(got, exp), ("fold", c.name, half, hex(x), nbits)
        folded.free()
        ck.free()
        n_fold += 1
        n_cases += 1
        continue
    n_key = int(rs.choice([1, 2, 3, 17, 255, 256, 257, 1000, 4097, 20000, 40000]))
    n_key = min(n_key, 40000)
    xy = pools[c.name][:n_key]
    if rs.rand() < 0.35:
        xy = xy.copy()
        k = int(rs.randint(1, min(n_key, 24) + 1))
        xy[rs.randint(0, n_key, size=k)] = adv[c.name][rs.randint(0, len(adv[c.name]), size=k)]
        n_adv += 1
    w = int(rs.choice([0, 0, 0, 2, 3, 5, 8, 10, 13, 15, 16, 17, 19]))
    flags = int(rs.choice([1, 2]))
    ctx.set_window(w)
    ck = CommitterKey.load(ctx, xy, None, flags)
    kind = str(rs.choice(["uniform", "few", "sparse", "top", "witness"]))
    mode = str(rs.choice(["single", "batch", "multi", "grouped"]))
    off = int(rs.randint(0, n_key)) if rs.rand() < 0.3 else 0
    n = int(rs.randint(1, n_key - off + 1))
    tag = (c.name, n_key, w, flags, kind, mode, off, n)
    if mode == "single":
        sc = scalars(c, n, kind)
        out, inf = VariableBaseMSM.multi_scalar_mul(ck, sc, base_off=off)
        ref, rinf = cref.msm(c.curve_id, xy[off:off + n], sc, threads=4)
        assert inf == rinf and np.array_equal(out, ref), tag
    elif mode == "batch":
        vs = [scalars(c, n, kind) for _ in range(int(rs.randint(1, 6)))]
        outs, infs = VariableBaseMSM.multi_scalar_mul_batch(ck, [ctx.upload(v) for v in vs], mont=False, base_off=off)
        for j, v in enumerate(vs):
            ref, rinf = cref.msm(c.curve_id, xy[off:off + n], v, threads=4)
            assert bool(infs[j]) == rinf and np.array_equal(outs[j], ref), (tag, j)
    elif mode == "multi":
        jobs = []
        for _ in range(int(rs.randint(1, 5))):
            o2 = int(rs.randint(0, n_key))
            n2 = int(rs.randint(1, n_key - o2 + 1))
            jobs.append((o2, scalars(c, n2, kind)))
        outs, infs = VariableBaseMSM.multi_scalar_mul_multi(ck, [(o2, ctx.upload(v)) for o2, v in jobs], mont=False)
        for j, (o2, v) in enumerate(jobs):
            ref, rinf = cref.msm(c.curve_id, xy[o2:o2 + len(v)], v, threads=4)
            assert bool(infs[j]) == rinf and np.array_equal(outs[j], ref), (tag, j)
    else:
        sc = scalars(c, n, kind)
        shift = int(rs.randint(0, 12))
        outs, infs = VariableBaseMSM.multi_scalar_mul_grouped(ck, ctx.upload(sc), shift, mont=False, base_off=off)
        idx = np.arange(n)
        for g in (0, 1):
            m = sc.copy()
            m[((idx >> shift) & 1) != g] = 0
            ref, rinf = cref.msm(c.curve_id, xy[off:off + n], m, threads=4)
            assert bool(infs[g]) == rinf and np.array_equal(outs[g], ref), (tag, g)
    ck.free()
    ctx.set_window(0)
    n_cases += 1
print(f"fuzz ok: {n_cases} cases ({n_big} large skewed, {n_fold} key folds, {n_adv} keys with adversarial points, {n_bpl} bucket-per-lane geometries, {n_direct} direct sums) "
      f"in {budget:.0f} s (seed {seed}); pipeline stats: " + ", ".join(f"{k} {v.pipeline_stats()}" for k, v in ctxs.items()))
