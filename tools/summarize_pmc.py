"""Summarise rocprofv3 --pmc counter_collection CSVs for the GAE kernels (per-dispatch averages)."""
import csv, glob, sys, collections
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        if "gae_rtg" not in name:
            continue
        out[name.split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in out.items():
    for c, vals in d.items():
        print(f"{k},{c},dispatches={len(vals)},avg={sum(vals)/len(vals):.1f},min={min(vals):.1f},max={max(vals):.1f}")
