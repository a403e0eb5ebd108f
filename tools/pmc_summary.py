"""Summarise the rocprofv3 PMC passes of tools/profile_round.sh into profiles/<round>_pmc_accum_l0.json.

    python tools/pmc_summary.py gpurun_out/profile_round profiles/r01_pmc_accum_l0.json

Every counter is averaged over the launches of k_accum_l0 found in the pass that collected it.  HBM traffic:
FETCH_SIZE / WRITE_SIZE are in KiB.  MI355X_MICROARCH.md (HBM section) says FETCH_SIZE under-counts wide coalesced
streaming reads 2x and that other patterns must be calibrated; tools/calib_gather.hip calibrated THIS kernel's
pattern (random 64-byte records fetched by LDS-DMA dwordx4, 4 lanes per record): FETCH_SIZE is exact (factor 1.00)
for it, 2.00 for the streaming reads.  The kernel's fetches are >90 % gather, so the reported traffic uses factor
1.00 for the gather share and 2.00 for the (known) streaming share: the sorted entry words, 4 B per entry.
"""
import collections
import csv
import glob
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_accum_l0" in r["Kernel_Name"] or "k_accum_bpl" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
mean = {k: sum(v) / len(v) for k, v in sorted(acc.items())}
# windows that hold digits: 15 at c = 17 (the default for 2^20 since round 2), 16 at c = 16 (AMSM_WINDOW=16); argv[3] overrides
n, windows = 1 << 20, (int(sys.argv[3]) if len(sys.argv) > 3 else 13)
entries = n * windows
stream_bytes = entries * 4  # entry words, read once (dwordx4 per 4 entries)
fetch = mean.get("FETCH_SIZE", 0.0) * 1024.0
write = mean.get("WRITE_SIZE", 0.0) * 1024.0
# fetch = gather/1.0 + stream/2.0  ->  true bytes = (fetch - stream/2) * 1.0 + stream
true_fetch = (fetch - stream_bytes / 2.0) + stream_bytes if fetch else 0.0
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_hash  # noqa: E402

out = {
    "kernel": "k_accum_bpl" if windows == 13 else "k_accum_l0",
    "source_hash": source_hash(),
    "workload": f"2^20 Pallas, precomputed key ({windows} entries per scalar), bench.py --sync",
    "launches_averaged": {k: len(v) for k, v in sorted(acc.items())},
    "counters_mean_per_launch": mean,
    "fetch_bytes_counter_x1": fetch,
    "fetch_bytes_counter_x2": 2.0 * fetch,
    "hbm_traffic_bytes_per_launch": true_fetch + write,
    "correction": "FETCH_SIZE KiB x1.00 for the random 64-B-record LDS-DMA gather (calibrated, tools/calib_gather.hip) and "
                  "x2.00 for the streamed entry words (4 B each) (MI355X_MICROARCH.md HBM section); WRITE_SIZE exact",
    "algorithmic_bytes_per_launch": n * 96,
    "gather_bytes_if_every_point_missed_cache": entries * 64,
}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: out[k] for k in ("fetch_bytes_counter_x1", "hbm_traffic_bytes_per_launch")}))
