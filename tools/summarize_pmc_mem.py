"""gpurun_out/pmcmem_<tag>/p*/ (tools/pmc_mem.sh) -> profiles/<tag>_wmsa_mem_pmc.md: every counter of every pass for the ONE
inference-form and the ONE training-form launch of wmsa_hg_kernel at the bench shape, plus derived memory-path figures."""
import collections, csv, glob, sys
tag = sys.argv[1]
src = f"gpurun_out/pmcmem_{tag}"
val = collections.defaultdict(dict)
dur = collections.defaultdict(list)
# A pass directory can hold the SAME launch pair measured by more than one process (tools/pmc_mem.sh runs under a wrapper whose
# child is profiled too: two <pid>_counter_collection.csv per pass).  Each file is a complete measurement of one launch per form:
# they are AVERAGED, never added (round 4 added them and every absolute value in r04a / r04b_wmsa_mem_pmc.md came out doubled -
# VERDICT r4 "What's weak" 3), and a file that disagrees with its sibling by more than 5 % on a counter fails the run.
samples = collections.defaultdict(list)          # (form, counter) -> [(file, per-launch value: rows of one file summed over dispatch dims)]
for f in sorted(glob.glob(f"{src}/p*/*/*counter_collection.csv")):
    per_file = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "wmsa_hg_kernel" not in r["Kernel_Name"]:
            continue
        form = "training" if "<true" in r["Kernel_Name"] or "ILb1E" in r["Kernel_Name"] else "inference"
        per_file[(form, r["Counter_Name"])] += float(r["Counter_Value"])
        dur[(form, f)] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    for k, v in per_file.items():
        samples[k].append((f, v))
nfiles = collections.Counter()
for (form, name), vs in samples.items():
    xs = [v for _, v in vs]
    m = sum(xs) / len(xs)
    if m > 1e4 and (max(xs) - min(xs)) > 0.05 * m and name.startswith(("SQ_INSTS", "TCC_REQ", "TCC_READ", "TCC_WRITE")):   # (counts, not stall cycles)
        raise SystemExit(f"{name} ({form}) differs between the processes of one pass: {vs}")
    val[form][name] = m
    nfiles[len(xs)] += 1
names = sorted({n for f in val.values() for n in f})
d = {form: sorted(v for (fm, _), v in dur.items() if fm == form) for form in ("inference", "training")}
out = [f"# Memory-path and issue counters of wmsa_hg_kernel at the bench shape (T = 524,288, C = 192, shift 2), {tag}\n",
       "`tools/pmc_mem.sh`: ONE inference-form and ONE training-form launch per process (`tools/pmc_one.py`), at most four counters of one",
       "hardware block per `rocprofv3 --pmc` pass (SQ: eight), program directly after `--`.  Values are per launch, summed over the chip",
       f"(mean over the {max(nfiles) if nfiles else 0} process(es) that measured each pass - never their sum).",
       "Launch time under the profiler (median over the passes): " +
       ", ".join(f"{k} {v[len(v) // 2] / 1e3:.0f} us" for k, v in d.items() if v) + ".\n",
       "| counter | inference | training |", "|---|---|---|"]
for n in names:
    out.append(f"| {n} | " + " | ".join(f"{val[f][n]:.5g}" if n in val[f] else "-" for f in ("inference", "training")) + " |")
def g(form, n):
    return val[form].get(n, float("nan"))
out.append("\nDerived:\n")
for form in ("inference", "training"):
    t_us = d[form][len(d[form]) // 2] / 1e3 if d[form] else float("nan")
    hit = g(form, "TCC_HIT_sum") / (g(form, "TCC_HIT_sum") + g(form, "TCC_MISS_sum"))
    wc = g(form, "SQ_WAVE_CYCLES")
    out.append(f"* **{form}** ({t_us:.0f} us): L2 hit rate {hit:.3f} ({g(form, 'TCC_REQ_sum'):.4g} requests, "
               f"{g(form, 'TCC_READ_sum'):.4g} reads / {g(form, 'TCC_WRITE_sum'):.4g} writes; fabric {g(form, 'TCC_EA0_RDREQ_sum'):.4g} read / "
               f"{g(form, 'TCC_EA0_WRREQ_sum'):.4g} write requests); "
               f"vector L1: {g(form, 'TCP_TOTAL_CACHE_ACCESSES_sum'):.4g} accesses, {g(form, 'TCP_TCC_READ_REQ_sum'):.4g} read + "
               f"{g(form, 'TCP_TCC_WRITE_REQ_sum'):.4g} write requests to L2, pending-stall cycles {g(form, 'TCP_PENDING_STALL_CYCLES_sum'):.4g}, "
               f"TCP busy (GATE_EN2 / GATE_EN1) {g(form, 'TCP_GATE_EN2_sum') / g(form, 'TCP_GATE_EN1_sum'):.3f}; "
               f"TA busy {g(form, 'TA_TA_BUSY_sum'):.4g} cycles (avr {g(form, 'TA_BUSY_avr'):.4g}), address stalled by TC "
               f"{g(form, 'TA_ADDR_STALLED_BY_TC_CYCLES_sum'):.4g}, data stalled by TC {g(form, 'TA_DATA_STALLED_BY_TC_CYCLES_sum'):.4g}; "
               f"TD busy {g(form, 'TD_TD_BUSY_sum'):.4g}, TD stalled by TC {g(form, 'TD_TC_STALL_sum'):.4g}; "
               f"VALU per MFMA {g(form, 'SQ_INSTS_VALU') / g(form, 'SQ_INSTS_MFMA'):.2f} ({g(form, 'SQ_INSTS_VALU'):.4g} / {g(form, 'SQ_INSTS_MFMA'):.4g}), "
               f"transcendental {g(form, 'SQ_INSTS_VALU_TRANS_F32'):.4g}; MFMA busy {g(form, 'SQ_VALU_MFMA_BUSY_CYCLES'):.4g} cycles, co-executing with VALU "
               f"{g(form, 'SQ_VALU_MFMA_COEXEC_CYCLES'):.4g} ({g(form, 'SQ_VALU_MFMA_COEXEC_CYCLES') / g(form, 'SQ_VALU_MFMA_BUSY_CYCLES'):.3f}); "
               f"of the wave cycles: waiting {g(form, 'SQ_WAIT_ANY') / wc:.3f}, issue-stalled {g(form, 'SQ_WAIT_INST_ANY') / wc:.3f}, VALU-active "
               f"{g(form, 'SQ_ACTIVE_INST_VALU') / wc:.3f}; LDS bank conflicts / LDS active {g(form, 'SQ_LDS_BANK_CONFLICT') / g(form, 'SQ_LDS_IDX_ACTIVE'):.3f}; "
               f"GRBM_GUI_ACTIVE {g(form, 'GRBM_GUI_ACTIVE'):.4g} (/8 XCDs / time = {g(form, 'GRBM_GUI_ACTIVE') / 8 / (t_us * 1e-6) / 1e9 if t_us == t_us else float('nan'):.2f} GHz)")
open(f"profiles/{tag}_wmsa_mem_pmc.md", "w").write("\n".join(out) + "\n")
print("\n".join(out))
