"""MSMs longer than the 2^20-pair window of the 20-bit key over ONE bucket set (round 5; csrc/api_types.h: struct Share,
k_accum_bpl<ACC>): range 1 writes the bucket table, ranges 2 .. k add to it, one reduction and one fold per MSM.  Against the C
restatement at 2^22 (BASELINE.json config 5), 2^21 and ragged lengths (a short last range stays an MSM of its own), in batches
(consecutive long MSMs alternate between the context's two tables), mixed with one-window MSMs, and with a range whose digits are
skewed where the probe does not look (the whole MSM falls back to independent ranges: same point)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import pyref as o

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = o.PALLAS


@pytest.fixture(scope="module")
def env(cref):
    from accumulation_amd import CommitterKey, Context
    ctx = Context(C.curve_id)
    n = (1 << 22) + 777
    ck = CommitterKey.generate(ctx, 0x5EED5005, n)
    assert ck.precomputed and ck.window_bits == 20
    xy, _ = ck.read()
    yield ctx, ck, xy
    ck.free()
    ctx.close()


def oracle(cref, xy, vec, n, off=0, mont=False):
    sc = vec.download()[:n]
    if mont:
        sc = cref.fr_from_mont(C.curve_id, sc)
    return cref.msm(C.curve_id, xy[off:off + n], sc, threads=8)


@pytest.mark.parametrize("n,shared", [(1 << 22, 1), (1 << 21, 1), ((1 << 21) + 5, 1), ((1 << 22) + 777, 1), (3 << 20, 1),
                                      ((1 << 20) + (1 << 19), 1), ((1 << 21) - 1, 1), ((1 << 20) + (1 << 17), 0)],
                         ids=["2p22", "2p21", "2p21_plus_5_short_tail", "2p22_plus_777", "3_ranges", "one_and_a_half",
                              "2p21_minus_1", "short_second_range_not_shared"])
def test_long_msm_over_one_bucket_set_vs_c_oracle(env, cref, n, shared):
    from accumulation_amd import VariableBaseMSM
    ctx, ck, xy = env
    v = ctx.random_vector(0x5EED5100 + (n & 0xFFFF), n, mont=False)
    before = ctx.pipeline_stats()
    got, ginf = VariableBaseMSM.multi_scalar_mul(ck, v, mont=False)
    after = ctx.pipeline_stats()
    ref, rinf = oracle(cref, xy, v, n)
    assert ginf == rinf and np.array_equal(got, ref)
    assert after["shared_bucket_sets"] - before["shared_bucket_sets"] == shared
    assert after["fallbacks"] == before["fallbacks"]


def test_batches_alternate_the_two_tables_and_mix_with_short_msms(env, cref):
    """five long MSMs and three one-window MSMs in ONE call (more long MSMs than tables, more ranges than slots), base offsets"""
    from accumulation_amd import VariableBaseMSM
    ctx, ck, xy = env
    n = 1 << 22
    big = [ctx.random_vector(0x5EED5200 + j, n, mont=True) for j in range(2)]
    small = ctx.random_vector(0x5EED5210, 1 << 20, mont=True)
    jobs = [(0, big[0]), (5, small), (0, big[1]), (777, big[0]), (0, small), (0, big[1]), (0, big[0])]
    before = ctx.pipeline_stats()["shared_bucket_sets"]
    got, ginf = VariableBaseMSM.multi_scalar_mul_multi(ck, jobs, mont=True)
    assert ctx.pipeline_stats()["shared_bucket_sets"] - before == 5
    cache = {}
    for k, (off, vec) in enumerate(jobs):
        key = (off, id(vec))
        if key not in cache:
            cache[key] = oracle(cref, xy, vec, min(vec.n, xy.shape[0] - off), off, mont=True)
        ref, rinf = cache[key]
        assert bool(ginf[k]) == rinf and np.array_equal(got[k], ref), k
    assert np.array_equal(got[0], got[6]) and np.array_equal(got[2], got[5])


def test_sharded_partial_record_and_commit_forms(env, cref):
    """the partial-record form (the one-process-per-GPU exchange) and a hiding commitment take the same path"""
    from accumulation_amd import PedersenCommitment
    ctx, ck, xy = env
    n = 1 << 22
    v = ctx.random_vector(0x5EED5300, n, mont=True)
    got = PedersenCommitment.commit(ck, v, None)
    ref, rinf = oracle(cref, xy, v, n, mont=True)
    assert bool(got[1]) == rinf and np.array_equal(got[0], ref)


CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from accumulation_amd import CommitterKey, Context, VariableBaseMSM, ffi
from oracle import cref
cref.build(); cref.load()
ctx = Context(ffi.AMSM_PALLAS)
n = 1 << 22
ck = CommitterKey.generate(ctx, 0x5EED5005, n)
xy, _ = ck.read()
rows = ctx.random_vector(0x5EED5400, n, mont=False).download()
which = int(sys.argv[1])
rows[which << 20:(which + 1) << 20] = rows[12345]      # one range of 2^20 equal scalars: a whole window per bucket
v = ctx.upload(rows)
got, ginf = VariableBaseMSM.multi_scalar_mul(ck, v, mont=False)
st = ctx.pipeline_stats()
ref, rinf = cref.msm(ffi.AMSM_PALLAS, xy, rows, threads=8)
assert ginf == rinf and np.array_equal(got, ref), "wrong point"
print("OK", st["shared_bucket_sets"], st["fallbacks"])
"""


@pytest.mark.parametrize("which", [0, 3], ids=["first_range_skewed", "last_range_skewed"])  # (a middle range: tools/fuzz_pipelines.py --long-share)
def test_one_skewed_range_sends_the_whole_msm_back(which):
    """AMSM_BPL_PROBE=0 / AMSM_TWO_VALUED=0: nothing looks at the scalars beforehand, so the skewed range is found by its prep's
    overflow flag -- in the first range (which would have written the table), a middle one, the last (which carries the tail)"""
    env = dict(os.environ, AMSM_BPL_PROBE="0", AMSM_TWO_VALUED="0")
    p = subprocess.run([sys.executable, "-c", CHILD % ROOT, str(which)], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    tag, shared, fallbacks = p.stdout.split()[-3:]
    assert tag == "OK" and int(shared) == 1 and int(fallbacks) >= 1


def test_switch_off_gives_the_same_point():
    """AMSM_SHARE_BUCKETS=0 (independent ranges, round 4's path) against the default, in fresh processes"""
    child = r"""
import sys, numpy as np, hashlib
sys.path.insert(0, %r)
from accumulation_amd import CommitterKey, Context, VariableBaseMSM, ffi
ctx = Context(ffi.AMSM_PALLAS)
n = (1 << 22) + 99
ck = CommitterKey.generate(ctx, 0x5EED5005, n)
v = ctx.random_vector(7, n, mont=True)
got, ginf = VariableBaseMSM.multi_scalar_mul(ck, v, mont=True)
print("R", hashlib.sha256(got.tobytes()).hexdigest(), int(ginf), ctx.pipeline_stats()["shared_bucket_sets"])
""" % ROOT
    outs = []
    for flag in ("1", "0"):
        p = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, timeout=900, env=dict(os.environ, AMSM_SHARE_BUCKETS=flag))
        assert p.returncode == 0, p.stderr[-3000:]
        outs.append(p.stdout.split()[-3:])
    assert outs[0][:2] == outs[1][:2] and outs[0][2] == "1" and outs[1][2] == "0"
