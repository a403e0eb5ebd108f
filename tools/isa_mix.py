"""Instruction mix of one kernel in a hipcc -S listing:  python tools/isa_mix.py file.s kernel_substring [top_n]
(the whole function body up to its .Lfunc_end label, not only up to the first s_endpgm)"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = [i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l][0]
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
ins = [l.strip().split()[0] for l in lines[start + 1:end] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
c = collections.Counter(ins)
print(lines[start].split(":")[0], "static instructions:", len(ins))
groups = collections.Counter()
for k, v in c.items():
    if k.startswith("v_mad_u64") or k.startswith("v_mad_i64"):
        groups["mad64 (half rate)"] += v
    elif re.match(r"v_(lshlrev|lshrrev|ashrrev)_b64|v_lshl_add_u64|v_add_co|v_addc|v_sub_co|v_subb|v_add3|v_and_or|v_lshl_or|v_lshl_add|v_add_lshl|v_alignbit|v_bfe|v_mul_lo|v_mul_hi|v_xad", k):
        groups["other half-rate / 3-operand VALU"] += v
    elif k.startswith("v_cndmask") or k.startswith("v_cmp"):
        groups["select / compare"] += v
    elif k.startswith("v_"):
        groups["full-rate VALU"] += v
    elif k.startswith(("global_", "buffer_", "flat_", "ds_", "scratch_")):
        groups["memory / LDS"] += v
    else:
        groups["scalar / control"] += v
for k, v in groups.most_common():
    print(f"  [{k}] {v}")
for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
    print(f"  {k:28s} {v}")
