#!/bin/bash
# rocprofv3 evidence for the paths besides the headline MSM (run through gpurun from the repo root): the scalar-field vector
# kernels (kernel-trace stats + FETCH/WRITE PMC passes), the BLS12-381 MSM, one ipa_pc_as prove.  The program is always the
# direct child of rocprofv3; counters are collected in their own runs.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profile_extra
mkdir -p $OUT/summary
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_vec -- python3 $R/tools/bench_configs.py --vec > $OUT/trace_vec.log 2>&1
cp $(find $OUT/trace_vec -name "*kernel_stats.csv" | head -1) $OUT/summary/vec_kernel_stats.csv
grep '"kind": "vec"' $OUT/trace_vec.log > $OUT/summary/vec_bench_under_rocprof.jsonl
for p in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $p --kernel-include-regex "k_vec_combine|k_hp_t_vecs|k_vec_hadamard" --output-format csv -d $OUT/pmc_vec_$p -- python3 $R/tools/bench_configs.py --vec > $OUT/pmc_vec_$p.log 2>&1
  cp $(find $OUT/pmc_vec_$p -name "*counter_collection.csv" | head -1) $OUT/pmc_vec_$p.csv
done
python3 - "$OUT" <<'PY'
import csv, json, sys, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    try:
        rows = list(csv.DictReader(open(f"{out}/pmc_vec_{ctr}.csv")))
    except Exception as e:
        res["error"][ctr] = str(e)
        continue
    acc = collections.defaultdict(list)
    for r in rows:
        if r.get("Counter_Name") == ctr:
            name = r["Kernel_Name"].split("(")[0].replace("void amsm::", "")
            acc[name].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[k][ctr + "_raw_mean"] = sum(v) / len(v)
        res[k][ctr + "_launches"] = len(v)
json.dump(res, open(f"{out}/summary/pmc_vec_kernels.json", "w"), indent=1)
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_bls -- python3 $R/bench.py --curve bls12_381_g1 --steps 20 --warmup 4 --no-cpu-baseline --no-schemes > $OUT/trace_bls.log 2>&1
cp $(find $OUT/trace_bls -name "*kernel_stats.csv" | head -1) $OUT/summary/bench_bls12_381_kernel_stats.csv
grep "^{\"metric\"" $OUT/trace_bls.log | tail -1 > $OUT/summary/bench_bls12_381.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_ipa -- $R/build/profile_as ipa_pc_as 16 16 --reps 5 --shape harness > $OUT/trace_ipa.log 2>&1
cp $(find $OUT/trace_ipa -name "*kernel_stats.csv" | head -1) $OUT/summary/profile_as_ipa_pc_as_2p16_kernel_stats.csv
grep "^JSON" $OUT/trace_ipa.log > $OUT/summary/profile_as_ipa_pc_as_2p16.jsonl
ls -la $OUT/summary
