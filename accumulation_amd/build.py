"""Build libamsm.so in-tree with hipcc for gfx950 (one object per translation unit, in parallel).

    python -m accumulation_amd.build [--force]

hipcc cross-compiles without a GPU.  The objects land in build/obj/, the library next to this file so
that it travels with the repo snapshot to the GPU box (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(ROOT, "build", "obj")
LIB = os.path.join(HERE, "libamsm.so")
ARCH = "gfx950"
UNITS = ["api.hip", "kern_pallas.hip", "kern_bls12_381.hip", "kern_fr.hip"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-result", "-Wno-pass-failed",
         "-Rpass-analysis=kernel-resource-usage"]


def _deps():
    out = [os.path.join(ROOT, "include", "amsm.h")]
    for f in os.listdir(CSRC):
        out.append(os.path.join(CSRC, f))
    return out


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _unit_deps(unit: str, all_deps):
    """The files a unit really includes (the depfile hipcc wrote on its last compile: -MD), else everything."""
    dep = os.path.join(OBJ, unit.replace(".hip", ".d"))
    if not os.path.exists(dep):
        return all_deps
    try:
        words = open(dep).read().replace("\\\n", " ").split()
    except OSError:
        return all_deps
    files = [w for w in words[1:] if not w.endswith(":") and (w.startswith(ROOT) or not os.path.isabs(w))]
    files = [f if os.path.isabs(f) else os.path.join(ROOT, f) for f in files]
    return [f for f in files if os.path.exists(f)] or all_deps


def _compile(unit: str) -> str:
    src = os.path.join(CSRC, unit)
    obj = os.path.join(OBJ, unit.replace(".hip", ".o"))
    log = obj + ".log"
    extra = os.environ.get("AMSM_EXTRA_FLAGS", "").split()  # (sanitizer / debug builds of the host side)
    cmd = ["hipcc", *FLAGS, *extra, "-MD", "-MF", obj[:-2] + ".d", "-c", src, "-o", obj]
    with open(log, "w") as lf:
        rc = subprocess.call(cmd, stdout=lf, stderr=subprocess.STDOUT)
    if rc != 0:
        sys.stderr.write(open(log).read()[-6000:])
        raise RuntimeError(f"hipcc failed on {unit} (log: {log})")
    return obj


def build_lib(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    deps = _deps()
    todo = [u for u in UNITS if force or _stale(os.path.join(OBJ, u.replace(".hip", ".o")), _unit_deps(u, deps))]
    if todo and verbose:
        print(f"[accumulation_amd.build] hipcc --offload-arch={ARCH}: {', '.join(todo)}", flush=True)
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(_compile, todo))
    objs = [os.path.join(OBJ, u.replace(".hip", ".o")) for u in UNITS]
    if todo or not os.path.exists(LIB):
        subprocess.check_call(["hipcc", "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv))
