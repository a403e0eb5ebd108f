"""Several contexts of ONE process driving the same GPU at once, each from its own thread (what a multi-device context's workers
do, and what a host with one context per prover thread does): every MSM must still equal the CPU restatement.  Round 6 found a race
this way -- a slot's ticket counters zeroed by a hipMemset that the context's non-blocking streams did not wait for, visible only when
many contexts made their first reductions side by side (profiles/r06_experiments.md section 1b); this test keeps that class of bug
from hiding behind single-context suites."""
import threading

import numpy as np
import pytest

from oracle import pyref as o

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("c", [o.PALLAS, o.BLS12_381_G1], ids=lambda c: c.name)
def test_eight_contexts_eight_threads_fresh_slots(cref, c):
    from accumulation_amd import CommitterKey, Context, VariableBaseMSM, ffi
    n_ctx, sizes = 8, [(1 << 16), 5000, (1 << 14) + 3, 1 << 12]
    n = max(sizes)
    xy = cref.rng_points(c.curve_id, 0xC0C, n, threads=16)
    vecs = [cref.rng_frs(c.curve_id, 0xC10 + j, n) for j in range(3)]
    want = {}
    for sz in sizes:
        for j in range(3):
            want[(sz, j)] = cref.msm(c.curve_id, xy[:sz], vecs[j][:sz])
    errors = []

    def worker(t):
        try:
            # a FRESH context per thread: its slots allocate (and zero) their buffers while the other threads are already computing
            ctx = Context(c.curve_id)
            keys = {}
            for flags in (ffi.AMSM_BASES_PRECOMPUTE | ffi.AMSM_BASES_NO_DIRECT_TABLE, ffi.AMSM_BASES_NO_PRECOMPUTE):
                keys[flags] = CommitterKey.load(ctx, xy, None, flags)
            for rep in range(2):
                for sz in sizes:
                    for flags, ck in keys.items():
                        # batches (three slots in flight) and blocking calls, host slices
                        pts, infs = VariableBaseMSM.multi_scalar_mul_batch_host(ck, [v[:sz] for v in vecs])
                        for j in range(3):
                            ref, rinf = want[(sz, j)]
                            if not (np.array_equal(pts[j], ref) and bool(infs[j]) == bool(rinf)):
                                errors.append((t, rep, sz, flags, j, "batch"))
                        got, ginf = VariableBaseMSM.multi_scalar_mul(ck, vecs[t % 3][:sz])
                        ref, rinf = want[(sz, t % 3)]
                        if not (np.array_equal(got, ref) and bool(ginf) == bool(rinf)):
                            errors.append((t, rep, sz, flags, "blocking"))
            for ck in keys.values():
                ck.free()
            ctx.close()
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_ctx)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:5]
