"""bench.py's contract on a GPU box: the default N = 1 line carries every field the driver reads (plus `roofline`,
`cpu_baseline` and `accumulations`), and the N > 1 branch -- launched exactly like the driver launches it, but with both
ranks on the one GPU of the test box and gloo instead of RCCL -- runs through its barrier / max-over-ranks timing / sharded
MSM path and prints one line with n_gpus = 2."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline"]


def _line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_default_line_n1(built_lib):
    # (--cpu-leg-max-log2 14: the CPU legs of `accumulations` at <= 2^14 -- the same code path in seconds instead of a minute)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "12", "--warmup", "3", "--cpu-leg-max-log2", "14"], capture_output=True,
                       text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    for k in KEYS + ["cpu_baseline", "accumulations"]:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 12 and d["warmup"] == 3 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["metric"].startswith("MSM throughput") and d["unit"] == "pairs/s" and d["value"] > 1e8
    assert set(["bound", "achieved", "peak", "unit", "frac", "traffic"]) <= set(d["roofline"])
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["gpu_result_bit_exact_vs_cpu"] is True
    assert d["cpu_baseline"]["timed_batch_msms_checked"] == 12  # every MSM of the timed batch against the CPU result
    acc = {k: v for k, v in d["accumulations"].items() if isinstance(v, dict) and "accumulations_per_s" in v}
    # 4 schemes x 2 shapes + hp_as again with the reference harness's constant inputs (the two-valued form)
    # (incl. the two uniform-witness r1cs_nark_as lines, trivial_pc_as on the host backend and -- round 6 -- config 3: ipa_pc_as at 2^20 on BLS12-381)
    assert len(acc) == 16 and all(v.get("verified") for v in acc.values()), d["accumulations"]
    assert acc["ipa_pc_as_2^20_n2_1in_1acc_nozk_bls12_381"]["curve"] == "bls12_381_g1"
    # the CPU side of every entry: the same harness on the library's host backend, size and threads named, no silent extrapolation
    cpu = d["accumulations"]["cpu"]
    assert cpu["threads_per_run"] >= 1 and cpu["cpu_model"] and cpu["host_cores"] >= 1
    for k, v in acc.items():
        if k.endswith("_host_backend"):
            assert "cpu" not in v
            continue
        c = v["cpu"]
        assert c["verified"] is True and c["accumulations_per_s"] > 0 and c["log2_size"] >= 10, (k, c)
        assert ("gpu_over_cpu_prove" in c) == bool(c.get("full_size")), (k, c)
        assert c["log2_size"] <= 14 and c["full_size"] == (c["log2_size"] == int(k.split("^")[1].split("_")[0])), (k, c)
    for k in ("trivial_pc_as_2^10_harness_1in_2acc_zk", "ipa_pc_as_2^16_n2_1in_1acc_nozk", "r1cs_nark_as_2^18_n2_1in_1acc_nozk",
              "hp_as_2^22_n2_1in_1acc_nozk", "ipa_pc_as_2^20_n2_1in_1acc_nozk_bls12_381"):
        assert acc[k]["cpu"].get("verified") is True, (k, acc[k]["cpu"])
    # BASELINE config 3's MSM beside the headline: BLS12-381 G1 at 2^20, scalars uniform in [0, r), bit-exact
    cfg = d["config"]
    assert cfg["pairs_per_s_bls12_381_2p20"] > 1e8 and cfg["bls12_381_2p20"]["gpu_result_bit_exact_vs_cpu"] is True
    # the ark-ec call shape (host bases + host scalars per call): PCIe-bound, and the line says how close to the link it gets
    assert cfg["pairs_per_s_oneshot_host_bases"] > 5e7 and 0.2 < cfg["oneshot_host_bases"]["fraction_of_h2d_bound"] <= 1.05, cfg["oneshot_host_bases"]
    assert cfg["bls12_381_2p20"]["timed_batch_msms_checked"] == 12 and 0.40 < cfg["bls12_381_2p20"]["scalars_above_2^254"] < 0.50
    assert sum(1 for k in acc if k.endswith("_harness_constant_inputs")) == 2
    assert all(v["sponge"] == "poseidon" for v in acc.values())  # the reference's sponge, not the SHA-256 stand-in
    assert len(d["accumulations"]["sha256_standin_prove_ms"]) == 8
    assert any(k.endswith("harness_1in_2acc_zk") for k in acc)
    assert "workload" in d["config"] and d["config"]["ms_per_msm_host_scalars"] > d["config"]["ms_per_msm_synchronous_call"] * 0.9
    assert d["roofline"]["traffic_source"] and set(d["stage_ms"]) == {"in_batch", "blocking_call"}
    assert "prep_chain" in d["stage_ms"]["in_batch"]


def test_single_process_two_shards_on_one_gpu(built_lib):
    """--single-process: the N-device job from one process through amsm_ctx_create_multi (both shards on GPU 0 here)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-process", "--devices", "0,0",
                        "--log2n", "16", "--steps", "4", "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    for k in KEYS[:-1]:
        assert k in d, k
    assert d["n_gpus"] == 2 and d["config"]["collective"] == "peer-copy" and d["value"] > 1e6
    assert d["cpu_baseline"]["gpu_result_bit_exact_vs_cpu"] is True


def test_two_rank_branch_on_one_gpu(built_lib):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6",
                        "--warmup", "2", "--backend", "gloo", "--one-gpu"], capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    for k in KEYS:
        assert k in d, k
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["value"] > 1e7
    assert "point-sharded x2" in d["config"]["parallelism"]


def _torchrun(nproc, port, script, *args, timeout=1500):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
                           "127.0.0.1", "--master-port", str(port), script, *args], capture_output=True, text=True, timeout=timeout,
                          cwd=ROOT, env=env)


@pytest.mark.parametrize("strong", [False, True], ids=["weak", "strong"])
def test_eight_rank_branch_on_one_gpu(built_lib, strong):
    """The driver's 8-GPU command line with all eight ranks on the one GPU of this box (gloo; numbers meaningless): rendezvous, the
    rank table, max-over-ranks timing, eight 832 MiB keys + eight workspaces + eight host pools side by side -- what the first real
    8-rank run would otherwise meet for the first time."""
    args = ["--gpus", "8", "--steps", "4", "--warmup", "1", "--backend", "gloo", "--one-gpu"] + (["--strong", "--no-schemes"] if strong else [])
    r = _torchrun(8, 29561 + int(strong), os.path.join(ROOT, "bench.py"), *args)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    d = _line(r.stdout)
    for k in KEYS:
        assert k in d, k
    cfg = d["config"]
    assert d["n_gpus"] == 8 and d["steps"] == 4 and d["scaling"] == ("strong" if strong else "weak")
    assert len(cfg["ranks"]) == 8 and sorted(x["rank"] for x in cfg["ranks"]) == list(range(8))
    per_rank = (1 << 20) // 8 if strong else 1 << 20
    assert all(x["pairs"] == per_rank for x in cfg["ranks"])
    assert d["value"] * d["ms_per_step"] * 1e-3 == pytest.approx(8 * per_rank, rel=1e-6)  # whole-job pairs / max-over-ranks time
    for x in cfg["ranks"]:  # what a bad curve would be diagnosed from
        assert x["key_bytes"]["table"] > 0 and x["workspace_bytes"] > 0 and "pipeline_stats" in x
        assert 0 <= x["host_pool_threads"] <= 7
    import multiprocessing
    assert cfg["ranks"][0]["host_pool_threads"] <= max(0, multiprocessing.cpu_count() // 8 - 1) or cfg["ranks"][0]["host_pool_threads"] == 0
    assert "gloo" in cfg["collective"] and "128-byte records" in cfg["collective"]
    assert cfg["value_cold"] and cfg["value_cold"] > 0
    if not strong:  # the N-GPU form of the `accumulations` half: the C++ harness over ONE context of eight shards (all on GPU 0 here)
        acc = d["accumulations_multi_device"]
        assert "--devices 0,0,0,0,0,0,0,0" in acc["driver"] and "error" not in acc
        for key in ("r1cs_nark_as_2^18_n2_1in_1acc_nozk_uniform_witness_replicated_keys", "r1cs_nark_as_2^18_harness_1in_2acc_zk_replicated_keys",
                    "trivial_pc_as_2^10_n2_1in_1acc_nozk", "ipa_pc_as_2^16_harness_1in_2acc_zk", "ipa_pc_as_2^16_n2_1in_1acc_nozk",
                    "r1cs_nark_as_2^18_harness_1in_2acc_zk", "r1cs_nark_as_2^18_n2_1in_1acc_nozk", "hp_as_2^22_harness_1in_2acc_zk",
                    "hp_as_2^22_n2_1in_1acc_nozk"):
            assert acc[key]["verified"] is True and acc[key]["accumulations_per_s"] > 0, (key, acc[key])
            assert acc[key]["serialize_roundtrip"] is True, key


def test_sharded_schemes_eight_ranks_on_one_gpu(built_lib):
    """BASELINE configs 4 / 5 in their 8-way layout (tools/bench_sharded.py): hp_as over 2^22 elements and r1cs_nark_as over 2^18
    constraints split over eight ranks that share GPU 0, records exchanged over gloo; both accumulations verify and decide"""
    r = _torchrun(8, 29571, os.path.join(ROOT, "tools", "bench_sharded.py"), "--backend", "gloo", "--one-gpu", "--reps", "2")
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    kinds = {x["kind"]: x for x in lines}
    assert set(kinds) == {"hp_as_sharded", "r1cs_nark_as_sharded"}
    for x in lines:
        assert x["n_gpus"] == 8 and x["verify_ok"] and x["decide_ok"] and x["scaling"] == "strong"
    assert kinds["hp_as_sharded"]["elements_per_rank"] == (1 << 22) // 8
    assert kinds["r1cs_nark_as_sharded"]["constraints_per_rank"] == (1 << 18) // 8
