"""A short run of tools/fuzz_round6.py on the library's host backend in the CPU suite: amsm_msm_oneshot (random lengths, min(len),
identity bases by flag and by (0, 0), adversarial points, every scalar distribution) and amsm_ipa_jump_fold (random key lengths,
every j, 128-bit / full-width / tiny challenges, against physical folds and the big-integer oracle) -- the GPU soak of the same
script is profiles/r06_fuzz.txt."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_round6_fuzz_short_on_the_host_backend(built_lib):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_round6.py"), "8", "7", "--host"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and "fuzz_round6 ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
