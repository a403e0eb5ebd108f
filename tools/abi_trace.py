#!/usr/bin/env python3
"""Where does the HOST time of a scheme driver go?  Generates an LD_PRELOAD shim from include/amsm.h that wraps every exported
entry point with two clock reads and appends (name, start, end) to a log, runs a command under it and prints, for the LAST
`--window-ms` of the run (the last timed prove of tools/profile_as), the time inside each entry point and the time between calls
(the driver's own C++: transcript bookkeeping, allocation, serialisation).  No change to the product library.

    python tools/abi_trace.py [--window-ms 12] -- build/profile_as r1cs_nark_as 18 18 --shape harness --sponge poseidon --reps 3
"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def prototypes():
    src = open(os.path.join(ROOT, "include", "amsm.h")).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r"^\s*#.*$", " ", src, flags=re.M)
    out = []
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(amsm_\w+)\s*\(([^()]*)\)\s*;", src):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "typedef" in ret or "struct" in ret and "*" not in ret:
            continue
        names = []
        if params and params != "void":
            for p in params.split(","):
                names.append(re.findall(r"[A-Za-z_]\w*", p)[-1])
        out.append((ret, name, params or "void", names))
    return out


def generate(path):
    protos = prototypes()
    with open(path, "w") as f:
        f.write('#define _GNU_SOURCE\n#include <dlfcn.h>\n#include <stdint.h>\n#include <stddef.h>\n#include <stdio.h>\n#include <stdlib.h>\n#include <time.h>\n'
                '#include "amsm.h"\n'
                'static FILE* lg;\nstatic double now(void){struct timespec t;clock_gettime(CLOCK_MONOTONIC,&t);return t.tv_sec*1e3+t.tv_nsec*1e-6;}\n'
                'static void rec(const char* n,double a,double b){if(!lg){const char* p=getenv("ABI_TRACE_LOG");lg=fopen(p?p:"/tmp/abi_trace.log","w");}'
                'fprintf(lg,"%s %.4f %.4f\\n",n,a,b);}\n'
                '__attribute__((destructor)) static void fin(void){if(lg)fclose(lg);}\n')
        for ret, name, params, names in protos:
            call = f"fn({', '.join(names)})"
            f.write(f"{ret} {name}({params}){{static {ret}(*fn)({params});if(!fn)fn=({ret}(*)({params}))dlsym(RTLD_NEXT,\"{name}\");double a=now();")
            if ret == "void":
                f.write(f"{call};rec(\"{name}\",a,now());}}\n")
            else:
                f.write(f"{ret} r={call};rec(\"{name}\",a,now());return r;}}\n")
    return len(protos)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--window-ms", type=float, default=0.0, help="summarise only the last W ms before the last call (0: everything)")
    ap.add_argument("--after", default="", help="summarise from the LAST call of this entry point on (e.g. amsm_vec_random)")
    ap.add_argument("--mark", default="amsm_ctx_is_host", help="summarise the span between the last two calls of this entry point")
    ap.add_argument("--timeline", type=int, default=0, help="also print the first N calls of the span (start, duration, gap before)")
    ap.add_argument("--min-us", type=float, default=0.0, help="timeline: only calls at least this long")
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    c, so, log = (os.path.join(ROOT, "build", x) for x in ("abi_trace.c", "libabi_trace.so", "abi_trace.log"))
    n = generate(c)
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"), c, "-o", so, "-ldl"])
    p = subprocess.run(cmd, env=dict(os.environ, LD_PRELOAD=so, ABI_TRACE_LOG=log), capture_output=True, text=True)
    sys.stdout.write(p.stdout[-1500:])
    if p.returncode != 0:
        sys.stderr.write(p.stderr[-2000:])
        return p.returncode
    calls = [(ln.split()[0], float(ln.split()[1]), float(ln.split()[2])) for ln in open(log) if ln.strip()]
    calls.sort(key=lambda x: x[1])
    print(f"\n{n} entry points wrapped, {len(calls)} calls logged")
    sel = calls
    if a.after:
        idx = [i for i, c_ in enumerate(calls) if c_[0] == a.after]
        sel = calls[idx[-1]:] if idx else calls
    if a.mark:  # the span between the last two calls of the marker entry point (tools/profile_as.cpp: the last timed prove)
        idx = [i for i, c_ in enumerate(calls) if c_[0] == a.mark]
        if len(idx) >= 2:
            sel = calls[idx[-2] + 1:idx[-1]]
    if a.window_ms:
        end = sel[-1][2]
        sel = [c_ for c_ in sel if c_[1] >= end - a.window_ms]
    span = sel[-1][2] - sel[0][1]
    inside, count, between = {}, {}, 0.0
    gaps = []
    prev_end = sel[0][1]
    for name, s, e in sel:  # (calls from helper threads may nest in time: clamp)
        inside[name] = inside.get(name, 0.0) + (e - s)
        count[name] = count.get(name, 0) + 1
        if s > prev_end:
            between += s - prev_end
            gaps.append((s - prev_end, name))
        prev_end = max(prev_end, e)
    print(f"window {span:.3f} ms: inside the library {sum(inside.values()):.3f} ms, between calls (driver C++) {between:.3f} ms")
    for k in sorted(inside, key=inside.get, reverse=True)[:25]:
        print(f"  {k:44s} {count[k]:5d} x {inside[k] / count[k] * 1e3:9.1f} us = {inside[k]:8.3f} ms")
    prev = sel[0][1]
    shown = 0
    for name, s_, e_ in sel:
        if shown >= a.timeline:
            break
        if 1e3 * (e_ - s_) < a.min_us:
            prev = max(prev, e_)
            continue
        shown += 1
        print(f"    {s_ - sel[0][1]:9.3f} ms  {name:40s} {1e3 * (e_ - s_):9.1f} us   gap {1e3 * max(0.0, s_ - prev):8.1f} us")
        prev = max(prev, e_)
    print("  largest gaps between calls (ms, before which call):", ", ".join(f"{g:.3f} {nm}" for g, nm in sorted(gaps, reverse=True)[:8]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
