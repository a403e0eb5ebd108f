"""The reference's six-scenario test template at ITS iteration count -- NUM_ITERATIONS = 50 (src/lib.rs:273,398-459) -- on the four C++
scheme drivers (include/amsm_{hp_as,r1cs_nark_as,ipa_pc_as,trivial_pc_as}.hpp), with and without zk, on the library's host backend
(no GPU needed).  The GPU suite runs the same programs at 2-3 iterations; here every prove / verify of 50 iterations x
{1; 3; 1,1; 1,1,2,3; 1,0,0,0} inputs must verify and every iteration's last accumulator must decide."""
import os
import subprocess

import pytest

from tests import test_cpp_hp_as, test_cpp_ipa_pc_as, test_cpp_r1cs_nark_as, test_cpp_trivial_pc_as

NAMES = ["single_input_init", "multiple_inputs_init", "simple_accumulation", "multiple_inputs_accumulation", "accumulators_only"]


@pytest.mark.parametrize("mod", [test_cpp_hp_as, test_cpp_r1cs_nark_as, test_cpp_ipa_pc_as, test_cpp_trivial_pc_as],
                         ids=["hp_as", "r1cs_nark_as", "ipa_pc_as", "trivial_pc_as"])
def test_six_scenarios_at_fifty_iterations_on_the_host_backend(built_lib, mod):
    mod.build()
    out = subprocess.run([mod.EXE], capture_output=True, text=True, timeout=1500,
                         env=dict(os.environ, AMSM_CHECK_DEVICE="-1", AMSM_CHECK_ITERATIONS="50"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [ln.split() for ln in out.stdout.splitlines() if ln.startswith("scenario")]
    assert ["done"] in [ln.split() for ln in out.stdout.splitlines()]
    zk_modes = 1 if mod is test_cpp_trivial_pc_as else 2  # (the scheme has no zk mode: src/trivial_pc_as/mod.rs:314)
    fifty = [ln for ln in lines if ln[-1] == "50" and "ok" in ln]
    assert len(fifty) == len(NAMES) * zk_modes, out.stdout[-1500:]
    assert {ln[1] for ln in fifty} == set(NAMES)
    once = [ln for ln in lines if ln[1] == "no_inputs_init"]
    assert len(once) == zk_modes and all(ln[-1] == "1" for ln in once)  # src/lib.rs:451-458: one iteration
