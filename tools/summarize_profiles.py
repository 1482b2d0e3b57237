"""gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) -> profiles/<tag>_kernel_stats.csv, _summary.md, _traffic.json"""
import collections, csv, glob, json, os, shutil, sys
tag = sys.argv[1]
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
st = glob.glob(f"{src}/stats/*/*kernel_stats.csv")[0]
shutil.copy(st, f"profiles/{tag}_kernel_stats.csv")
rows = list(csv.DictReader(open(st)))
line = [l for l in open(f"{src}/stats.log") if l.startswith('{"metric"')][0]
bench = json.loads(line)
steps = bench["steps"] + max(bench["warmup"], 2)
def counters(d, name):
    f = glob.glob(f"{src}/{d}/*/*counter_collection.csv")[0]
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            vals[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return vals
fe, wr = counters("fetch", "FETCH_SIZE"), counters("write", "WRITE_SIZE")
KEY = "gemm_bs_kernelIDF16bLi128ELb0ELi5ELb1E"       # bf16, BN 128, no stats, flags BIAS|GELU_DUAL, plain A = stage-1 fc1
k = [n for n in fe if KEY in n][0]
fetch_kb, write_kb = sum(fe[k]) / len(fe[k]), sum(wr[k]) / len(wr[k])
avg_ns = [float(r["AverageNs"]) for r in rows if KEY in r["Name"]][0]
traffic = {
    "kernel": k, "launches_sampled": len(fe[k]),
    "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
    "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE counts 64 B per 128-B request for 16-B/lane streaming reads -> x2; WRITE_SIZE exact; separate --pmc passes",
    "hbm_bytes_per_launch": 2 * fetch_kb * 1024 + write_kb * 1024,
    "rocprof_avg_launch_ms": avg_ns / 1e6, "bench_line": bench,
}
json.dump(traffic, open(f"profiles/{tag}_traffic.json", "w"), indent=1)
with open(f"profiles/{tag}_summary.md", "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats of `python bench.py --steps {bench['steps']} --warmup {bench['warmup']}` ({tag})\n\n")
    f.write(f"{bench['value']} img/s, {bench['ms_per_step']} ms/step; totals over the {steps} traced steps.\n\n| kernel | calls | ms/step | avg us | % |\n|---|---|---|---|---|\n")
    for r in rows[:22]:
        f.write(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6/steps:.2f} | {float(r['AverageNs'])/1e3:.1f} | {r['Percentage']} |\n")
    f.write(f"\nRoofline kernel (stage-1 fc1, `{KEY}`): rocprof average {avg_ns/1e3:.1f} us; HBM traffic per launch "
            f"{traffic['hbm_bytes_per_launch']/1e6:.0f} MB (FETCH_SIZE {fetch_kb:.0f} KB x2 + WRITE_SIZE {write_kb:.0f} KB) vs 1811 MB algorithmic.\n")
print(json.dumps(traffic)[:300])
