"""RCCL executes ONCE on the one-GPU test box, both ways the N > 1 path reaches it (north_star: "an
RCCL reduce of partial bucket sums over xGMI"):

 * inside libamsm.so: AMSM_COLLECTIVE=rccl with a single-device list makes amsm_ctx_create_multi build a ONE-rank
   communicator (dlopen'd librccl, ncclCommInitAll) and its keys sharded keys with one shard, so every MSM over them runs
   msm_sharded -- partial record, ncclAllGather of raw bytes, host fold -- exactly the code an 8-GPU context runs;
 * one process per GPU (the driver's torchrun contract): torch.distributed backend "nccl" (= RCCL on ROCm), world_size 1,
   ShardedMSM(force_collective=True) so the all-gather is issued although there is nothing to gather.

Both in FRESH child processes (a process that has initialised the GPU is never re-executed), results against the CPU
restatement.  What this cannot show is xGMI traffic or a multi-rank rendezvous -- that stays the driver's 8-GPU run -- but
the binding (symbol names, the ncclUint8 constant, stream / buffer conventions) no longer meets the hardware for the
first time there."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD_LIB = r"""
import json, sys
import numpy as np
sys.path.insert(0, {root!r})
from accumulation_amd import CommitterKey, Context, MultiContext, PedersenCommitment, VariableBaseMSM, ffi
n = {n}
mc = MultiContext(ffi.AMSM_PALLAS, devices=(0,))
out = {{"collective": mc.collective, "devices": mc.num_devices}}
ck = CommitterKey.generate(mc, 0x5EED7001, n)
out["shards"] = int(mc._lib.amsm_bases_num_shards(ck._h))
vecs = [mc.random_vector(0x700 + j, n, mont=False) for j in range(3)]
c0 = mc.collectives
pts, infs = VariableBaseMSM.multi_scalar_mul_batch(ck, vecs, mont=False)
one, one_inf = VariableBaseMSM.multi_scalar_mul(ck, vecs[1], mont=False)
out["collectives"] = mc.collectives - c0
out["pts"] = [p.tolist() for p in pts] + [one.tolist()]
out["infs"] = [bool(x) for x in infs] + [bool(one_inf)]
# the same MSMs on an ordinary context (no collective at all)
sc = Context(ffi.AMSM_PALLAS)
ck1 = CommitterKey.generate(sc, 0x5EED7001, n)
v1 = [sc.random_vector(0x700 + j, n, mont=False) for j in range(3)]
p1, i1 = VariableBaseMSM.multi_scalar_mul_batch(ck1, v1, mont=False)
out["single"] = [p.tolist() for p in p1]
out["scalars"] = [v.download().tolist() for v in v1] if n <= 4096 else None
xy, _ = ck1.read()
out["xy"] = xy.tolist() if n <= 4096 else None
print("RESULT " + json.dumps(out))
"""

CHILD_TORCH = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, {root!r})
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "{port}")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
from accumulation_amd import CommitterKey, Context, VariableBaseMSM, ffi
from accumulation_amd.dist import HipEngine, ShardedMSM
n = {n}
ctx = Context(ffi.AMSM_PALLAS)
ck = CommitterKey.generate(ctx, 0x5EED7002, n)
vecs = [ctx.random_vector(0x710 + j, n, mont=False) for j in range(3)]
sm = ShardedMSM(HipEngine(ctx, ck), force_collective=True)
out = {{"backend": dist.get_backend(), "world": dist.get_world_size()}}
pts, infs = sm.msm_batch(vecs, mont=False)
one, one_inf = sm.msm(vecs[2], mont=False)
ref, rinf = VariableBaseMSM.multi_scalar_mul_batch(ck, vecs, mont=False)
out["pts"] = [np.asarray(p).tolist() for p in pts] + [np.asarray(one).tolist()]
out["infs"] = [bool(x) for x in infs] + [bool(one_inf)]
out["ref"] = [p.tolist() for p in ref] + [ref[2].tolist()]
out["rinf"] = [bool(x) for x in rinf] + [bool(rinf[2])]
dist.barrier()
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
"""


def _run_child(code, env_extra, timeout=600):
    env = dict(os.environ)
    env.update(env_extra)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-2000:] + "\n" + p.stderr[-4000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
    assert line, p.stdout[-2000:] + "\n" + p.stderr[-2000:]
    return json.loads(line[-1][len("RESULT "):])


def test_one_rank_rccl_communicator_inside_the_library(cref):
    """amsm_ctx_create_multi + AMSM_COLLECTIVE=rccl on device list (0,): collective == "rccl", keys have one shard, one
    ncclAllGather per sharded call, results equal the ordinary context's and the CPU restatement's"""
    from accumulation_amd import ffi
    n = 4096
    r = _run_child(CHILD_LIB.format(root=ROOT, n=n), {"AMSM_COLLECTIVE": "rccl"})
    assert r["collective"] == "rccl" and r["devices"] == 1 and r["shards"] == 1
    assert r["collectives"] == 2  # one exchange per CALL: the batch of three, the single MSM
    xy = np.array(r["xy"], dtype=np.uint64)
    for j in range(3):
        sc = np.array(r["scalars"][j], dtype=np.uint64)
        ref, rinf = cref.msm(ffi.AMSM_PALLAS, xy, sc)
        assert r["infs"][j] == bool(rinf) and r["pts"][j] == ref.tolist(), j
        assert r["single"][j] == ref.tolist()
    assert r["pts"][3] == r["pts"][1] and r["infs"][3] == r["infs"][1]


def test_without_the_request_a_single_device_list_builds_no_communicator():
    r = _run_child(CHILD_LIB.format(root=ROOT, n=8192), {"AMSM_COLLECTIVE": ""})
    assert r["collective"] == "none" and r["shards"] == 1 and r["collectives"] == 0
    assert r["pts"][:3] == r["single"]


def test_torch_nccl_backend_world_size_one_all_gather():
    """backend "nccl" IS RCCL on ROCm: the ShardedMSM all-gather of the partial records, forced at world_size 1"""
    r = _run_child(CHILD_TORCH.format(root=ROOT, n=1 << 14, port=29500 + (os.getpid() % 2000)), {})
    assert r["backend"] == "nccl" and r["world"] == 1
    assert r["pts"] == r["ref"] and r["infs"] == r["rinf"]
