"""Summarise rocprofv3 counter_collection CSVs (one row per kernel: per-dispatch means in millions).

    python scripts/pmc_summary.py gpurun_out/pmc_TAG [kernel-substring]
"""
import collections
import csv
import glob
import sys

root = sys.argv[1]
needle = sys.argv[2] if len(sys.argv) > 2 else "resample"
for path in sorted(glob.glob(f"{root}/*_counter_collection.csv")):
    sums = collections.defaultdict(lambda: collections.defaultdict(float))
    counts = collections.defaultdict(lambda: collections.defaultdict(int))
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"]
        if needle not in name:
            continue
        key = (name[:90], row["Grid_Size"])
        sums[key][row["Counter_Name"]] += float(row["Counter_Value"])
        counts[key][row["Counter_Name"]] += 1
    for key, counters in sums.items():
        print(path.split("/")[-1], key)
        for counter, total in sorted(counters.items()):
            print(f"    {counter:28s} {total / counts[key][counter] / 1e6:12.3f} M  (n={counts[key][counter]})")
