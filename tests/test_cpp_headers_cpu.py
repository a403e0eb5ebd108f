"""The C++ side above the C ABI (include/*.hpp, the test drivers in tests/cpp, tools/profile_as.cpp) compiles without a GPU:
syntax-only with -Wall -Werror, both with the default test sponge and with the Poseidon sponge selected.  The GPU tests build
and run the same sources; this keeps a header break from hiding until then."""
import glob
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = sorted(glob.glob(os.path.join(ROOT, "tests", "cpp", "*.cpp"))) + [os.path.join(ROOT, "tools", "profile_as.cpp")]


@pytest.mark.parametrize("src", SOURCES, ids=lambda p: os.path.basename(p))
@pytest.mark.parametrize("defs", [[], ["-DAMSM_TEST_POSEIDON"]], ids=["sha256_sponge", "poseidon_sponge"])
def test_compiles(src, defs):
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", *defs, "-I", os.path.join(ROOT, "include"), src])
