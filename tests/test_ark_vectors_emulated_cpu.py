"""Keeps the consumers of the pin one command away from working: tools/ark_vectors/emulate.py writes the
generator's file layout from THIS REPO'S OWN oracle into a scratch directory (it pins nothing and says so in every file), and the
consuming tests must be green on it -- so that the day someone runs the real Rust generator, only the numbers can be new."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_consumers_pass_on_emulated_vectors(built_lib, tmp_path):
    d = str(tmp_path / "ark_emulated")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "ark_vectors", "emulate.py"), d], stdout=subprocess.DEVNULL)
    for f in os.listdir(d):
        assert "NOT arkworks" in json.load(open(os.path.join(d, f)))["generator"]  # never mistaken for the pin
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_ark_vectors_cpu.py"), "-q", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, ARK_VECTORS_DIR=d), timeout=900)
    assert r.returncode == 0 and " passed" in r.stdout and "skipped" not in r.stdout, r.stdout[-1500:] + r.stderr[-500:]
