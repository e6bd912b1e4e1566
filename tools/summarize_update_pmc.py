"""Per-dispatch averages of rocprofv3 --pmc counter_collection CSVs for the update kernels: config,kernel,counter,dispatches,avg,min,max"""
import csv, glob, sys, collections
out = collections.defaultdict(lambda: collections.defaultdict(list))
want = ("ppo_update", "icm_", "mat_update", "clip_adam", "grad_sqnorm")
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        if not any(w in name for w in want):
            continue
        out[name.split("(")[0].replace("void ", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(out):
    for c, vals in sorted(out[k].items()):
        print(f"{sys.argv[2]},{k},{c},dispatches={len(vals)},avg={sum(vals)/len(vals):.1f},min={min(vals):.1f},max={max(vals):.1f}")
