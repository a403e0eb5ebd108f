"""CPU test of the N > 1 path: world_size 2 over gloo.  The per-rank MSM is played by the CPU oracle
(tests may use oracle/; the product engine is accumulation_amd.dist.HipEngine, which needs a GPU), so what
is covered here is the host logic: shard bounds, the fixed-size partial record, the all-gather of raw
bytes and the identical fold on every rank."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from accumulation_amd.dist import ShardedMSM, shard_bounds
from oracle import pyref as o
from tests import helpers as h


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 8, 9, 1000, (1 << 20) + 3):
        for world in (1, 2, 3, 8):
            ranges = [shard_bounds(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            for (a, b), (c, d) in zip(ranges, ranges[1:]):
                assert b == c and a <= b
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1


class OracleEngine:
    """Stand-in for HipEngine in the CPU test: same record layout (XYZZ, 4*L u32 little-endian)."""

    def __init__(self, curve, xy_shard):
        from oracle import cref
        self.c, self.cref, self.xy = curve, cref, xy_shard
        self.record_bytes = 4 * 2 * curve.limbs * 4

    def partial(self, scalars, mont):
        sc = self.cref.fr_from_mont(self.c.curve_id, scalars) if mont else scalars
        out, inf = self.cref.msm(self.c.curve_id, self.xy, sc, threads=2)
        L2 = 2 * self.c.limbs  # u32 limbs per coordinate
        rec = np.zeros(4 * L2, dtype=np.uint32)
        if not inf:
            rec[: 2 * L2] = out.view(np.uint32)
            one = np.array(o.int_to_limbs(self.c.R % self.c.p, self.c.limbs), dtype=np.uint64).view(np.uint32)
            rec[2 * L2: 3 * L2] = one
            rec[3 * L2:] = one
        return torch.from_numpy(rec.view(np.uint8).copy())

    def combine(self, gathered, count):
        L2 = 2 * self.c.limbs
        recs = gathered.numpy().view(np.uint32).reshape(count, 4 * L2)
        acc = None
        for r in recs:
            if not r[2 * L2: 3 * L2].any():
                continue
            xy = r[: 2 * L2].copy().view(np.uint64)
            acc = o.add(self.c, acc, h.np_to_point(self.c, xy, 0))
        w, inf = o.point_to_mont_limbs(self.c, acc)
        return np.array(w, dtype=np.uint64), bool(inf)


def _worker(rank, world, init_file, n, q):
    from oracle import cref
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    try:
        c = o.PALLAS
        xy = cref.rng_points(c.curve_id, 0x5EED1001, n)
        sc = cref.rng_scalars(0x5EED0001, n)
        lo, hi = shard_bounds(n, rank, world)
        eng = OracleEngine(c, xy[lo:hi])
        sm = ShardedMSM(eng)
        out, inf = sm.msm(sc[lo:hi], mont=False)
        # batched form: 3 MSMs (the second with the scalars reversed), one all-gather of all the records
        sc2 = cref.rng_scalars(0x5EED0002, n)
        outs, infs = sm.msm_batch([sc[lo:hi], sc2[lo:hi], sc[lo:hi]], mont=False)
        q.put((rank, out.tolist(), inf, outs.tolist(), [bool(x) for x in infs]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [1, 333])
def test_sharded_msm_world2_gloo(n, cref):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with tempfile.TemporaryDirectory() as d:
        init_file = os.path.join(d, "init")
        procs = [ctx.Process(target=_worker, args=(r, world, init_file, n, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = [q.get(timeout=240) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    c = o.PALLAS
    ref, rinf = cref.msm(c.curve_id, cref.rng_points(c.curve_id, 0x5EED1001, n), cref.rng_scalars(0x5EED0001, n))
    ref2, rinf2 = cref.msm(c.curve_id, cref.rng_points(c.curve_id, 0x5EED1001, n), cref.rng_scalars(0x5EED0002, n))
    for rank, out, inf, outs, infs in res:
        assert inf == rinf and out == ref.tolist(), rank
        assert infs == [rinf, rinf2, rinf] and outs == [ref.tolist(), ref2.tolist(), ref.tolist()], rank


# ---- the PRODUCT engine on the library's host backend (round 5): the same two-rank exchange, no stand-in ----------------------------
def _worker_host_backend(rank, world, init_file, n, q, curve="pallas"):
    from accumulation_amd import CommitterKey, Context, ffi
    from accumulation_amd.dist import HipEngine
    from oracle import cref
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    try:
        c = o.CURVES[curve]
        xy = cref.rng_points(c.curve_id, 0x5EED1001, n)
        sc, sc2 = cref.rng_scalars(0x5EED0001, n), cref.rng_scalars(0x5EED0002, n)
        lo, hi = shard_bounds(n, rank, world)
        ctx = Context(c.curve_id, device=ffi.AMSM_DEVICE_HOST)
        ck = CommitterKey.load(ctx, xy[lo:hi], None, ffi.AMSM_BASES_DEFAULT)
        sm = ShardedMSM(HipEngine(ctx, ck))
        a, b = ctx.upload(sc[lo:hi]), ctx.upload(sc2[lo:hi])
        out, inf = sm.msm(a, mont=False)
        outs, infs = sm.msm_batch([a, b, a], mont=False)
        q.put((rank, out.tolist(), bool(inf), np.asarray(outs).tolist(), [bool(x) for x in infs]))
        ck.free()
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,curve", [(2, "pallas"), (5000, "pallas"), (1500, "bls12_381_g1")])
def test_sharded_msm_world2_gloo_on_the_host_backend(n, curve, cref, built_lib):
    """accumulation_amd.dist.HipEngine over a host-backend context (libamsm.so itself: amsm_msm_partial[_batch]_device,
    amsm_partials_combine[_batch]) on two gloo ranks without a GPU -- the N > 1 torchrun form of the product, not of a stand-in"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with tempfile.TemporaryDirectory() as d:
        init_file = os.path.join(d, "init")
        procs = [ctx.Process(target=_worker_host_backend, args=(r, world, init_file, n, q, curve)) for r in range(world)]
        for p in procs:
            p.start()
        res = [q.get(timeout=300) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    c = o.CURVES[curve]  # (BLS12-381: 192-byte partial records through the same all-gather)
    pts = cref.rng_points(c.curve_id, 0x5EED1001, n)
    ref, rinf = cref.msm(c.curve_id, pts, cref.rng_scalars(0x5EED0001, n))
    ref2, rinf2 = cref.msm(c.curve_id, pts, cref.rng_scalars(0x5EED0002, n))
    for rank, out, inf, outs, infs in res:
        assert inf == rinf and out == ref.tolist(), rank
        assert infs == [rinf, rinf2, rinf] and outs == [ref.tolist(), ref2.tolist(), ref.tolist()], rank


# ---- REPLICATED keys, the one-process-per-GPU form (round 6): whole MSMs dealt to the ranks, one all-gather of affine results --------
def _worker_replicated(rank, world, init_file, n, q, curve="pallas"):
    from accumulation_amd import CommitterKey, Context, ffi
    from accumulation_amd.dist import ReplicatedMSM
    from oracle import cref
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    try:
        c = o.CURVES[curve]
        xy = cref.rng_points(c.curve_id, 0x5EED1001, n)
        ctx = Context(c.curve_id, device=ffi.AMSM_DEVICE_HOST)
        ck = CommitterKey.load(ctx, xy, None, ffi.AMSM_BASES_DEFAULT)  # the WHOLE key on every rank
        vecs = [ctx.upload(cref.rng_frs(c.curve_id, 0x5EED0100 + j, n)) for j in range(5)]
        outs, infs = ReplicatedMSM(ck).msm_batch(vecs, mont=False)
        none, _ = ReplicatedMSM(ck).msm_batch([], mont=False)
        q.put((rank, np.asarray(outs).tolist(), [bool(x) for x in infs], len(none)))
        ck.free()
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,curve", [(2, "pallas"), (3, "pallas"), (2, "bls12_381_g1")])
def test_replicated_msm_over_gloo_on_the_host_backend(world, curve, cref, built_lib):
    """five MSMs over a replicated key on two / three gloo ranks of the library's host backend: rank r computes MSMs r, r + world,
    .. whole, one all-gather of the affine results, every rank ends with all five in call order -- equal to the CPU restatement"""
    n = 3000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with tempfile.TemporaryDirectory() as d:
        init_file = os.path.join(d, "init")
        procs = [ctx.Process(target=_worker_replicated, args=(r, world, init_file, n, q, curve)) for r in range(world)]
        for p in procs:
            p.start()
        res = [q.get(timeout=300) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    c = o.CURVES[curve]
    pts = cref.rng_points(c.curve_id, 0x5EED1001, n)
    refs = [cref.msm(c.curve_id, pts, cref.rng_frs(c.curve_id, 0x5EED0100 + j, n)) for j in range(5)]
    for rank, outs, infs, n_none in res:
        assert n_none == 0
        assert infs == [bool(r[1]) for r in refs] and outs == [r[0].tolist() for r in refs], rank
