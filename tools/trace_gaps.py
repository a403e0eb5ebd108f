"""Cut a rocprofv3 --kernel-trace --memory-copy-trace timeline of tools/host_slices_trace.py into its phases (idle gaps > 50 ms)
and attribute each phase: when every accumulation started / ended, what ran in between, the uploads.
    python tools/trace_gaps.py gpurun_out/r5a/hs/runc/544   (prefix of the _kernel_trace.csv / _memory_copy_trace.csv pair)"""
import csv, sys, re
pre = sys.argv[1]
K = [r for r in csv.DictReader(open(pre + "_kernel_trace.csv"))]
M = [r for r in csv.DictReader(open(pre + "_memory_copy_trace.csv"))]
ev = []
for r in K:
    name = re.sub(r"^void amsm::", "", r["Kernel_Name"]).split("<")[0].split("(")[0]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", name, r["Stream_Id"]))
for r in M:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r["Direction"].replace("MEMORY_COPY_", ""), r["Stream_Id"]))
ev.sort()
phases, cur = [], []
for e in ev:
    if cur and e[0] - max(x[1] for x in cur) > 50_000_000:
        phases.append(cur); cur = []
    cur.append(e)
phases.append(cur)
for pi, ph in enumerate(phases):
    acc = [e for e in ph if e[3] == "k_accum_bpl"]
    if len(acc) < 6:
        continue
    t0, t1 = ph[0][0], max(e[1] for e in ph)
    cps = [e for e in ph if e[2] == "C" and e[3] == "HOST_TO_DEVICE" and e[1] - e[0] > 100_000]
    print(f"\n== phase {pi}: {len(acc)} accumulations, span {(t1-t0)/1e6:.3f} ms, {len(cps)} large uploads")
    print(f"   first event -> first accum start: {(acc[0][0]-t0)/1e6:.3f} ms; last accum end -> phase end: {(t1-acc[-1][1])/1e6:.3f} ms")
    durs = [(e[1]-e[0])/1e6 for e in acc]
    gaps = [(acc[i+1][0]-acc[i][1])/1e6 for i in range(len(acc)-1)]
    print("   accum dur ms:", " ".join(f"{d:.3f}" for d in durs), f"| sum {sum(durs):.3f}")
    print("   gaps ms     :", " ".join(f"{g:.3f}" for g in gaps), f"| sum {sum(gaps):.3f}")
    if cps:
        print("   uploads (start rel ms, dur ms):", " ".join(f"{(c[0]-t0)/1e6:.2f}/{(c[1]-c[0])/1e6:.2f}" for c in cps))
        print("   accum starts rel ms:", " ".join(f"{(a[0]-t0)/1e6:.2f}" for a in acc))
    # time by kernel name inside the phase
    tot = {}
    for e in ph:
        tot[e[3]] = tot.get(e[3], 0) + (e[1]-e[0])/1e6
    print("   busy ms by name:", ", ".join(f"{k} {v:.2f}" for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:9]))
