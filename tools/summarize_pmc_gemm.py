"""Per-launch counter table from a tools/pmc_gemm.sh output directory.  Usage: python tools/summarize_pmc_gemm.py <dir> <kernel substring>"""
import collections, csv, glob, sys
base, pat = sys.argv[1], sys.argv[2]
res = collections.defaultdict(dict)
names = {}
for p in sorted(glob.glob(base + "/p*/*/*_counter_collection.csv")):
    byd = collections.OrderedDict()
    for r in csv.DictReader(open(p)):
        if pat not in r["Kernel_Name"]:
            continue
        byd.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
        names[int(r["Dispatch_Id"])] = r["Kernel_Name"][:60]
    for i, (d, c) in enumerate(sorted(byd.items())):
        res[i].update(c)
kt = sorted(glob.glob(base + "/p*/*/*_kernel_trace.csv"))[-1]
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt)) if pat in r["Kernel_Name"]]
keys = sorted({k for c in res.values() for k in c})
print("| counter | " + " | ".join(f"launch {i}" for i in sorted(res)) + " |")
print("|---|" + "---|" * len(res))
print("| duration (us, last pass) | " + " | ".join(f"{durs[i]:.0f}" for i in sorted(res)) + " |")
for k in keys:
    print(f"| {k} | " + " | ".join(f"{res[i].get(k, float('nan')):.4g}" for i in sorted(res)) + " |")
