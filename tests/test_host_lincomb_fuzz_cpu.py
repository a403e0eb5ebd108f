"""A short run of tools/fuzz_host_lincomb.py in the CPU suite: amsm_host_lincomb_batch (GLV halves, signed digits, terms shared over
the host pool, fixed-base tables that get built, hit and go stale over recurring points) against the big-int oracle, both curves.
profiles/r05_fuzz.txt holds the long runs."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_lincomb_fuzz_short(built_lib):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_host_lincomb.py"), "--seconds", "10", "--seed", "77"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "fuzz_host_lincomb ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
