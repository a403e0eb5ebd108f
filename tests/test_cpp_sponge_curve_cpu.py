"""tests/cpp/sponge_curve_check.cpp on the library's host backend: the drivers' default-argument Poseidon sponge is the sponge over
the CONTEXT's curve (BLS12-381: the 381-bit base field), a pristine sponge of another curve is re-created, a used one is refused."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_argument_sponge_follows_the_context_curve(built_lib):
    exe = os.path.join(ROOT, "build", "sponge_curve_check")
    libdir = os.path.join(ROOT, "accumulation_amd")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "sponge_curve_check.cpp"), "-o", exe + f".{os.getpid()}", "-L", libdir,
                           "-l:libamsm.so", f"-Wl,-rpath,{libdir}", "-Wl,--allow-shlib-undefined"])
    os.replace(exe + f".{os.getpid()}", exe)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=dict(os.environ, AMSM_CHECK_DEVICE="-1"))
    assert r.returncode == 0 and "SPONGE_CURVE_OK" in r.stdout, r.stdout[-1000:] + r.stderr[-1000:]
