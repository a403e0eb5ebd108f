"""GLV set-up and scalar split of the key folds (accumulation_amd/csrc/host_glv.h) on the host, no GPU: lambda / beta pair up on
the generator, the lattice vectors are short, and [k] P == [k1] P + [k2] phi(P) through the digit masks the kernel consumes,
for random and edge-case scalars on both curves (tests/cpp_host/glv_check.cpp against the host group law).  The device side
(k_points_fold with the split) is compared with the big-int oracle in tests/test_ipa_gpu.py."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_glv_split_matches_plain_scalar_multiplication():
    out = os.path.join(ROOT, "build", "glv_check")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["hipcc", "-std=c++17", "-O2", "--offload-host-only", "--offload-arch=gfx950", "-x", "hip", "-w",
                           "-I", os.path.join(ROOT, "accumulation_amd", "csrc"), os.path.join(ROOT, "tests", "cpp_host", "glv_check.cpp"),
                           "-o", out])
    res = subprocess.run([out], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "OK" in res.stdout
