"""gpurun_out/pmc_<tag>/{a,b,c,d} (tools/pmc_wmsa.sh) -> profiles/<tag>_wmsa_pmc.md: per-launch averages of the counters for the
inference and training builds of wmsa_hg_kernel at the bench shape."""
import collections, csv, glob, sys
tag = sys.argv[1]
src = f"gpurun_out/pmc_{tag}"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(f"{src}/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "wmsa_hg_kernel" not in r["Kernel_Name"]:
            continue
        form = "training" if "<true" in r["Kernel_Name"] else "inference"
        agg[form][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[form].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
names = sorted({n for f in agg.values() for n in f})
with open(f"profiles/{tag}_wmsa_pmc.md", "w") as out:
    out.write(f"# PMC counters of wmsa_hg_kernel at the bench shape (T = 524,288, C = 192; tools/mb_wmsa.py 8 under rocprofv3 --pmc, {tag})\n\n"
              "Per-launch averages, summed over the chip (SQ_* cycle counters count quad-cycles per wave or per SIMD as documented in\n"
              "MI355X_MICROARCH.md; SQ_VALU_MFMA_BUSY_CYCLES counts cycles).  Launch time under the profiler: "
              + ", ".join(f"{k} {sum(v) / len(v) / 1e3:.0f} us" for k, v in dur.items()) + ".\n\n| counter | inference | training |\n|---|---|---|\n")
    for n in names:
        vals = []
        for form in ("inference", "training"):
            v = agg[form].get(n)
            vals.append(f"{sum(v) / len(v):.4g}" if v else "-")
        out.write(f"| {n} | {vals[0]} | {vals[1]} |\n")
    def g(form, n):
        v = agg[form].get(n)
        return sum(v) / len(v) if v else float("nan")
    out.write("\nDerived (inference | training):\n\n")
    for form in ("inference", "training"):
        wc = g(form, "SQ_WAVE_CYCLES")
        out.write(f"* {form}: VALU-active share of wave cycles {g(form, 'SQ_ACTIVE_INST_VALU') / wc:.3f}, LDS {g(form, 'SQ_ACTIVE_INST_LDS') / wc:.3f}, "
                  f"VMEM {g(form, 'SQ_ACTIVE_INST_VMEM') / wc:.3f}, waiting (s_waitcnt / barrier) {g(form, 'SQ_WAIT_ANY') / wc:.3f}, issue-stalled "
                  f"{g(form, 'SQ_WAIT_INST_ANY') / wc:.3f}; VALU instructions per MFMA {g(form, 'SQ_INSTS_VALU') / g(form, 'SQ_INSTS_MFMA'):.2f}; "
                  f"MFMA busy {g(form, 'SQ_VALU_MFMA_BUSY_CYCLES'):.4g} cycles of which co-executing with VALU {g(form, 'SQ_VALU_MFMA_COEXEC_CYCLES'):.4g}; "
                  f"LDS bank-conflict cycles / LDS active cycles {g(form, 'SQ_LDS_BANK_CONFLICT') / g(form, 'SQ_LDS_IDX_ACTIVE'):.3f}\n")
print(open(f"profiles/{tag}_wmsa_pmc.md").read())
