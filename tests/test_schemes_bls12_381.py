"""The four accumulation layers over BLS12-381 G1 and its scalar field, each against the big-integer oracle: a short run of
tools/fuzz_schemes.py --bls12-381 (hp_as proves, r1cs_nark_as / ipa_pc_as / trivial_pc_as chains of random shape).  The reference's
own tests instantiate the layers over Pallas only (`type G = ark_pallas::Affine`: src/hp_as/mod.rs:1047, src/r1cs_nark_as/mod.rs:1279,
src/ipa_pc_as/mod.rs:1007, src/trivial_pc_as/mod.rs:756) and BASELINE config 3 runs ipa_pc_as on BLS12-381: nothing in the
drivers is tied to one curve, and this keeps it that way.  Once on the GPU, once on the library's host backend in the CPU suite."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(seconds, seed, *flags):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_schemes.py"), str(seconds), str(seed), "--bls12-381", *flags],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "fuzz_schemes ok" in r.stdout and "on bls12_381_g1" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
    return r.stdout


@pytest.mark.gpu
def test_layers_over_bls12_381_on_the_gpu(built_lib):
    _run(6, 31)


def test_layers_over_bls12_381_on_the_host_backend(built_lib):
    _run(8, 32, "--host")
