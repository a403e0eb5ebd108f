"""bench.py's CPU side of accumulations/sec (cpu_scheme_rates): the SAME C++ harness on the library's host backend, at small sizes
here -- every scheme / shape entry receives a `cpu` sub-object that verified and decided, names its size and thread count, and
never involves oracle/ (the host backend is product code)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_cpu_leg_fills_every_entry(built_lib, monkeypatch):
    import bench

    exe = os.path.join(ROOT, "build", "profile_as")
    libdir = os.path.join(ROOT, "accumulation_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "profile_as.cpp"),
                           "-o", exe, "-L", libdir, "-l:libamsm.so", f"-Wl,-rpath,{libdir}", "-Wl,--allow-shlib-undefined"])
    runs = (("trivial_pc_as", 10, ["--reps", "5"], "", {"harness": 6, "n2": 6}),
            ("trivial_pc_as", 10, ["--reps", "5", "--device", "-1"], "_host_backend", None),
            ("ipa_pc_as", 16, ["--reps", "3"], "", {"harness": 7, "n2": 7}),
            ("ipa_pc_as", 20, ["--reps", "2", "--curve", "1"], "_bls12_381", {"harness": 6, "n2": 6}),
            ("r1cs_nark_as", 18, ["--reps", "3", "--uniform"], "_uniform_witness", {"harness": 8, "n2": 8}),
            ("hp_as", 9, ["--reps", "3"], "", {"harness": 8, "n2": 9}))
    monkeypatch.setattr(bench, "SCHEME_RUNS", runs)
    out = {}
    for scheme, lg, _, tag, _cpu in runs:
        for shape in ("harness_1in_2acc_zk", "n2_1in_1acc_nozk"):
            out[f"{scheme}_2^{lg}_{shape}{tag}"] = {"prove_ms": 1.0}
    bench.cpu_scheme_rates(exe, out)
    assert out["cpu"]["threads_per_run"] >= 1 and out["cpu"]["cpu_model"]
    assert "cpu" not in out["trivial_pc_as_2^10_n2_1in_1acc_nozk_host_backend"]  # that entry IS the host backend
    for key in ("trivial_pc_as_2^10_harness_1in_2acc_zk", "trivial_pc_as_2^10_n2_1in_1acc_nozk", "ipa_pc_as_2^16_harness_1in_2acc_zk",
                "ipa_pc_as_2^16_n2_1in_1acc_nozk", "ipa_pc_as_2^20_n2_1in_1acc_nozk_bls12_381",
                "r1cs_nark_as_2^18_n2_1in_1acc_nozk_uniform_witness", "r1cs_nark_as_2^18_harness_1in_2acc_zk_uniform_witness",
                "hp_as_2^9_n2_1in_1acc_nozk"):
        c = out[key]["cpu"]
        assert c["verified"] is True and c["accumulations_per_s"] > 0 and c["threads"] == out["cpu"]["threads_per_run"], (key, c)
        assert c["full_size"] == (key == "hp_as_2^9_n2_1in_1acc_nozk"), key
        assert ("gpu_over_cpu_prove" in c) == c["full_size"]  # a ratio only where both sides ran the same size
    assert out["hp_as_2^9_harness_1in_2acc_zk"]["cpu"]["log2_size"] == 8 and not out["hp_as_2^9_harness_1in_2acc_zk"]["cpu"]["full_size"]
    assert out["hp_as_2^9_harness_1in_2acc_zk"]["cpu"]["verified"] and out["ipa_pc_as_2^20_harness_1in_2acc_zk_bls12_381"]["cpu"]["log2_size"] == 6
