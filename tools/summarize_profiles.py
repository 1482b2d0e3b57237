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
KEYS = ("gemm_nt3_kernelILi257E", "gemm_nt3_kernel<257>")
KEY = "gemm_nt3_kernel<257>"       # pipelined bf16 NT kernel, flags BIAS|GELU: every linear fc1; the stage-1 launches
                                   # (M=524288, N=768, K=192) are the longest of them, selected by duration below
def dispatches(path, name=None):
    """(duration_ns, counter_value) of every dispatch of KEY in a rocprofv3 CSV (kernel trace or counter collection)."""
    out = []
    for r in csv.DictReader(open(path)):
        if any(k_ in r["Kernel_Name"] for k_ in KEYS) and (name is None or r["Counter_Name"] == name):
            out.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"]), float(r.get("Counter_Value", 0) or 0)))
    return out
def stage1(ds):
    mx = max(d for d, _ in ds)
    return [(d, v) for d, v in ds if d >= 0.8 * mx]
fe = stage1(dispatches(glob.glob(f"{src}/fetch/*/*counter_collection.csv")[0], "FETCH_SIZE"))
wr = stage1(dispatches(glob.glob(f"{src}/write/*/*counter_collection.csv")[0], "WRITE_SIZE"))
kt = stage1(dispatches(glob.glob(f"{src}/stats/*/*kernel_trace.csv")[0]))
k = KEY
fetch_kb, write_kb = sum(v for _, v in fe) / len(fe), sum(v for _, v in wr) / len(wr)
avg_ns = sum(d for d, _ in kt) / len(kt)
traffic = {
    "kernel": k + " (stage-1 fc1 launches, selected by duration)", "launches_sampled": len(fe),
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
    f.write(f"\nRoofline kernel (stage-1 fc1 = the longest launches of `{KEY}`): rocprof kernel-trace average {avg_ns/1e3:.1f} us; HBM traffic per launch "
            f"{traffic['hbm_bytes_per_launch']/1e6:.0f} MB (FETCH_SIZE {fetch_kb:.0f} KB x2 + WRITE_SIZE {write_kb:.0f} KB) vs 1007 MB algorithmic.\n")
print(json.dumps(traffic)[:300])
