"""Instruction mix of one kernel in a hipcc -S listing:  python tools/isa_mix.py file.s kernel_substring"""
import collections
import sys

lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = [i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.split(":")[0].endswith(l.split(":")[0]) and ":" in l][0]
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
ins = [l.strip().split()[0] for l in lines[start + 1:end] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
c = collections.Counter(ins)
print(lines[start].split(":")[0], "static instructions:", len(ins))
for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
    print(f"  {k:28s} {v}")
