"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel (per-wave averages for k_step)."""
import collections, csv, glob, sys
for d in sys.argv[1:]:
    for f in sorted(glob.glob(d + "/*/*_counter_collection.csv")):
        acc = collections.defaultdict(list); waves = 256
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("k_step"):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"])); waves = int(r["Grid_Size"]) // 64
        print(d, {k: round(sum(v) / len(v) / waves, 1) for k, v in acc.items()})
