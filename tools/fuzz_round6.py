"""Randomised runs of the round-6 entry points against the CPU oracle (test infrastructure: oracle/ is the checker here), beyond
the fixed cases of tests/test_oneshot_gpu.py, test_ipa_jump_gpu.py and test_replicated_gpu.py:
  * amsm_msm_oneshot -- random lengths either side of the 2^19-pair range boundary, min(len) both ways, identity bases by flag and
    by (0, 0), adversarial points, every scalar distribution, canonical / Montgomery scalars, one- and three-shard contexts;
  * amsm_ipa_jump_fold -- random key lengths 2^7 .. 2^13, every j that leaves a multiple of 64 generators, 128-bit / full-width /
    tiny challenges, against the library's own physical folds (k_points_fold: another code path) and, for a few outputs, the
    big-integer oracle's naive sums;
  * replicated keys -- batches of random size and ragged lengths over a 2- or 3-device context (all GPU 0) against the C oracle,
    host slices and device vectors, with and without hiding terms.
Usage: python tools/fuzz_round6.py [seconds] [seed] [--host]   (--host: the library's host backend, no GPU needed; the
multi-device cases need devices and are skipped there)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402

from accumulation_amd import CommitterKey, Context, MultiContext, PedersenCommitment, VariableBaseMSM, ffi  # noqa: E402
from accumulation_amd.engine import _ptr  # noqa: E402
from accumulation_amd.scalar_field import Fr  # noqa: E402
from oracle import cref, pyref as o  # noqa: E402
from tests import helpers as h  # noqa: E402

HOST = "--host" in sys.argv
argv = [a for a in sys.argv if a != "--host"]
budget = float(argv[1]) if len(argv) > 1 else 60.0
seed = int(argv[2]) if len(argv) > 2 else 1
rs = np.random.RandomState(seed)
t_end = time.time() + budget
CURVES = (o.PALLAS, o.BLS12_381_G1)
DEV = ffi.AMSM_DEVICE_HOST if HOST else 0
ctxs = {c.name: Context(c.curve_id, device=DEV) for c in CURVES}
BIG = 1 << int(os.environ.get("FUZZ_BIG_LOG2", "20" if not HOST else "14"))
pools = {c.name: cref.rng_points(c.curve_id, 2000 + seed, BIG, threads=8) for c in CURVES}
_fix = json.load(open(os.path.join(ROOT, "tests", "golden", "adversarial_points.json")))
adv = {}
for _c in CURVES:
    _pts = [(int(x, 16), int(y, 16)) for v in _fix["curves"][_c.name].values() for x, y in v]
    _pts += [(P[0], (-P[1]) % _c.p) for P in _pts]
    adv[_c.name] = h.points_to_np(_c, _pts)[0]
n_one = n_jump = n_rep = n_big = 0


def scalars(c, n, kind):
    base = cref.rng_frs(c.curve_id, int(rs.randint(1 << 30)), n)
    if kind == "uniform" or n == 0:
        return base
    if kind == "few":
        return base[rs.randint(0, min(n, int(rs.randint(1, 5))), size=n)]
    if kind == "sparse":
        out = np.zeros_like(base)
        m = rs.rand(n) < 0.05
        out[m] = base[m]
        return out
    if kind == "witness":
        m = rs.rand(n) < float(rs.choice([0.1, 0.5, 0.9]))
        bits = np.zeros_like(base)
        bits[:, 0] = rs.randint(0, 2, size=n)
        out = base.copy()
        out[m] = bits[m]
        return out
    sp = h.scalars_to_np([c.r - 1, c.r - 2, (c.r - 1) // 2, (c.r + 1) // 2, 1, 0, (1 << 128) - 1, 1 << 240])
    return sp[rs.randint(0, len(sp), size=n)]


def oneshot_case(c):
    global n_one, n_big
    big = rs.rand() < 0.04 and not HOST
    if big:
        nb = int(rs.randint((1 << 19) - 2, BIG + 1))
        n_big += 1
    else:
        nb = int(rs.choice([0, 1, 2, 3, 63, 64, 65, 255, 1000, 4097, 20000, 70001])) if rs.rand() < 0.6 else int(rs.randint(1, 1 << 17))
    nb = min(nb, BIG)
    ns = nb if rs.rand() < 0.6 else int(rs.randint(0, nb + 50))
    xy = pools[c.name][:nb].copy()
    inf = None
    keep = np.ones(nb, dtype=bool)
    if nb and rs.rand() < 0.4:
        k = int(rs.randint(1, min(nb, 20) + 1))
        idx = rs.randint(0, nb, size=k)
        if rs.rand() < 0.5:
            inf = np.zeros(nb, dtype=np.uint8)
            inf[idx] = 1
        else:
            xy[idx] = 0  # (0, 0): the identity as ark-ec serialises it
        keep[idx] = False
    if nb and rs.rand() < 0.3:
        k = int(rs.randint(1, min(nb, 24) + 1))
        idx = rs.randint(0, nb, size=k)
        idx = idx[keep[idx]]
        xy[idx] = adv[c.name][rs.randint(0, len(adv[c.name]), size=len(idx))]
    kind = str(rs.choice(["uniform", "uniform", "few", "sparse", "witness", "top"]))
    sc = scalars(c, ns, kind)
    mont = bool(rs.rand() < 0.3)
    k = min(nb, ns)
    kk = keep[:k]
    tag = ("oneshot", c.name, nb, ns, kind, mont, inf is not None)
    if kk.any():
        ref, rinf = cref.msm(c.curve_id, xy[:k][kk], sc[:k][kk], threads=8)
    else:
        ref, rinf = np.zeros(2 * ctxs[c.name].fq_limbs, dtype=np.uint64), True
    ctx = ctxs[c.name]
    multi = None
    if not HOST and not big and rs.rand() < 0.15:
        multi = ctx = MultiContext(c.curve_id, [0] * int(rs.randint(2, 4)))
    try:
        L = 2 * ctx.fq_limbs
        got, ginf = VariableBaseMSM.multi_scalar_mul_oneshot(ctx, xy.reshape(-1, L), (cref.fr_to_mont(c.curve_id, sc) if mont and ns else sc).reshape(-1, 4),
                                                             inf, mont)
        assert bool(ginf) == bool(rinf) and (rinf or np.array_equal(got, ref)), tag
    finally:
        if multi is not None:
            multi.close()
    n_one += 1


def jump_case(c):
    global n_jump
    ctx = ctxs[c.name]
    fr = Fr(ctx.curve)
    log_key = int(rs.randint(7, 14)) if not HOST else int(rs.randint(3, 9))
    n = 1 << log_key
    j_max = log_key - 6 if not HOST else log_key - 1
    j = int(rs.randint(1, j_max + 1))
    m0 = n >> j
    xy = pools[c.name][int(rs.randint(0, BIG - n + 1)):][:n]
    ck = CommitterKey.load(ctx, xy, None, ffi.AMSM_BASES_PRECOMPUTE | ffi.AMSM_BASES_NO_DIRECT_TABLE)
    u = rs.rand()
    xs = []
    for r in range(j):
        x = int(o.rng_scalar(int(rs.randint(1 << 30)), r)) % c.r
        if u < 0.6:
            x %= 1 << 128
        elif u < 0.7:
            x = int(rs.randint(1, 4))
        elif u < 0.8:
            x = c.r - int(rs.randint(1, 4))
        xs.append(x or 1)
    out = np.zeros((m0, 2 * ctx.fq_limbs), dtype=np.uint64)
    oinf = np.zeros((m0,), dtype=np.uint8)
    xi = fr.to_limbs_many(xs)
    rc = ctx._lib.amsm_ipa_jump_fold(ctx._h, ck._h, log_key, _ptr(xi), j, _ptr(out), _ptr(oinf))
    tag = ("jump", c.name, log_key, j, [hex(x) for x in xs])
    assert rc == ffi.AMSM_OK, (tag, rc)
    key, half = ck, n // 2
    for x in xs:
        nxt = key.fold(half, fr.to_limbs(x), 255)
        if key is not ck:
            key.free()
        key, half = nxt, half // 2
    fxy, finf = key.read()
    key.free()
    assert np.array_equal(out, fxy) and np.array_equal(oinf, finf), tag
    if (1 << j) <= 16:
        gens = lambda k: [h.np_to_point(c, xy[t * m0 + k], 0) for t in range(1 << j)]  # noqa: E731
        S = []
        for t in range(1 << j):
            s = 1
            for r in range(j):
                if (t >> (j - 1 - r)) & 1:
                    s = s * xs[r] % c.r
            S.append(s)
        k = int(rs.randint(0, m0))
        assert h.np_to_point(c, out[k], oinf[k]) == o.msm_naive(c, gens(k), S), (tag, k)
    ck.free()
    n_jump += 1


def replicated_case(c):
    global n_rep
    n_dev = int(rs.randint(2, 4))
    n = int(rs.choice([1, 2, 100, 1000, 5000, 20000])) if rs.rand() < 0.6 else int(rs.randint(1, 1 << 15))
    multi = MultiContext(c.curve_id, [0] * n_dev)
    try:
        kseed = int(rs.randint(1 << 30))
        pk = PedersenCommitment.setup(multi, n, seed=kseed, flags=ffi.AMSM_BASES_REPLICATE)
        assert multi._lib.amsm_bases_replicas(pk._h) == n_dev
        xy, _ = pk.read()
        H = pk.hiding_generator
        k = int(rs.randint(1, 10))
        kind = str(rs.choice(["uniform", "uniform", "few", "witness", "top"]))
        hv = [scalars(c, int(rs.randint(1, n + 1)) if rs.rand() < 0.5 else n, kind) for _ in range(k)]
        rnd_i = [int(o.rng_scalar(int(rs.randint(1 << 30)), 0)) % c.r if rs.rand() < 0.4 else None for _ in range(k)]
        rnd = [None if r is None else cref.fr_to_mont(c.curve_id, h.scalars_to_np([r]))[0] for r in rnd_i]
        got = PedersenCommitment.commit_batch_host(pk, [cref.fr_to_mont(c.curve_id, v) for v in hv], rnd)
        tag = ("replicated", c.name, n_dev, n, k, kind)
        for i, v in enumerate(hv):
            bases, sc = xy[: v.shape[0]], v
            if rnd_i[i] is not None:
                bases = np.concatenate([bases, np.asarray(H, dtype=np.uint64).reshape(1, -1)])
                sc = np.concatenate([v, h.scalars_to_np([rnd_i[i]])])
            ref, rinf = cref.msm(c.curve_id, bases, sc, threads=4)
            assert bool(got[i][1]) == bool(rinf) and (rinf or np.array_equal(got[i][0], ref)), (tag, i)
        # equal lengths through amsm_msm_batch_device (device vectors on the primary)
        m = int(rs.randint(1, n + 1))
        eq = [scalars(c, m, kind) for _ in range(k)]
        dv = [multi.upload(cref.fr_to_mont(c.curve_id, v)) for v in eq]
        outs, infs = VariableBaseMSM.multi_scalar_mul_batch(pk, dv, mont=True)
        for i, v in enumerate(eq):
            ref, rinf = cref.msm(c.curve_id, xy[:m], v, threads=4)
            assert bool(infs[i]) == bool(rinf) and (rinf or np.array_equal(outs[i], ref)), (tag, "device", i)
        for v in dv:
            v.free()
        pk.free()
    finally:
        multi.close()
    n_rep += 1


while time.time() < t_end:
    c = CURVES[0] if rs.rand() < 0.65 else CURVES[1]
    u = rs.rand()
    if u < 0.5:
        oneshot_case(c)
    elif u < 0.8 or HOST:
        jump_case(c)
    else:
        replicated_case(c)
print(f"fuzz_round6 ok: {n_one} one-shot MSMs ({n_big} above 2^19 pairs), {n_jump} jump folds, {n_rep} replicated-key batches against the oracle "
      f"in {budget:.0f} s (seed {seed}{', host backend' if HOST else ''})")
