"""Instruction-class census of wmsa_hg_kernel from the compiler's assembly.

  python tools/valu_census.py [--save] [--out profiles/r04_wmsa_valu_census.md]

Compiles csrc/wmsa_hg.hip with -DSODT_HG_MARK -save-temps (phase markers as assembly comments, no instruction), walks the
kernel's text in file order and attributes every instruction to the last marker seen.  The head-step phases appear three times
(steps 0..2 are unrolled); masked / unmasked softmax bodies and the exact (cold) form are separate rows.  Static counts: the
dynamic count of a row is static x (its executions per window pair), given in the table header."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "small-object-detection-transformers_amd", "csrc", "wmsa_hg.hip")
save = "--save" in sys.argv
out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
tmp = tempfile.mkdtemp()
subprocess.run(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast", "-Wno-unused-value", "-Wno-inline-asm",
                "-DSODT_HG_MARK", "-save-temps", "-c", SRC, "-o", os.path.join(tmp, "w.o")], cwd=tmp, check=True, capture_output=True)
asm = open(os.path.join(tmp, "wmsa_hg-hip-amdgcn-amd-amdhsa-gfx950.s")).read().splitlines()
sym = "_ZN12_GLOBAL__N_114wmsa_hg_kernelILb%dELb0EEEv5WArgs" % (1 if save else 0)
start = next(i for i, l in enumerate(asm) if l.startswith(sym + ":"))
end = next(i for i in range(start, len(asm)) if asm[i].strip().startswith("s_endpgm"))

def klass(op):
    if op.startswith("v_mfma"): return "MFMA"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt")): return "VALU transcendental"
    if op.startswith("v_cvt_pk_bf16"): return "VALU cvt_pk_bf16"
    if op.startswith("v_cvt"): return "VALU other cvt"
    if op.startswith(("v_permlane", "v_readlane", "v_readfirstlane", "v_writelane", "v_mov_b32_dpp", "v_add_f32_dpp", "v_max_f32_dpp")) or "dpp" in op: return "VALU cross-lane"
    if op.startswith(("v_pk_",)): return "VALU packed f32"
    if op.startswith(("v_fma", "v_fmac", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mac", "v_mad_f32")): return "VALU f32 arithmetic"
    if op.startswith(("v_max", "v_min", "v_med3")): return "VALU max/min"
    if op.startswith(("v_and", "v_or", "v_xor", "v_lshl", "v_lshr", "v_ashr", "v_bfe", "v_bfi", "v_perm", "v_alignbit", "v_lshl_or", "v_and_or", "v_bitop")): return "VALU bit ops (unpack, masks)"
    if op.startswith(("v_cmp", "v_cndmask")): return "VALU compare/select"
    if op.startswith(("v_mov", "v_accvgpr", "v_swap")): return "VALU moves"
    if op.startswith(("v_add_u32", "v_add_co", "v_addc", "v_sub_u32", "v_mul_lo", "v_mul_hi", "v_mad_u", "v_mad_i", "v_add3", "v_sub_co", "v_mul_u32", "v_lshl_add", "v_add_lshl", "v_mad_co")): return "VALU integer / address"
    if op.startswith("v_"): return "VALU other"
    if op.startswith("ds_"): return "LDS"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")): return "VMEM"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_barrier"): return "s_barrier"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("s_"): return "SALU / branch"
    return "other"

cnt = collections.OrderedDict()
cur = "entry"
for l in asm[start:end]:
    t = l.strip()
    m = re.match(r"; HGMARK (.*)", t)
    if m:
        cur = m.group(1)
        continue
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        continue
    op = t.split()[0]
    cnt.setdefault(cur, collections.Counter())[klass(op)] += 1
classes = ["MFMA", "VALU f32 arithmetic", "VALU transcendental", "VALU cvt_pk_bf16", "VALU bit ops (unpack, masks)", "VALU max/min", "VALU cross-lane",
           "VALU compare/select", "VALU moves", "VALU integer / address", "VALU packed f32", "VALU other cvt", "VALU other", "LDS", "VMEM", "SALU / branch", "s_waitcnt", "s_barrier", "s_nop"]
lines = [f"# Instruction census of `wmsa_hg_kernel<{'save' if save else 'inference'}>` from the assembly (tools/valu_census.py)\n",
         "Static instruction counts per phase marker, file order (one wave).  Per window pair a wave executes: pair-setup, LN1, O->tile, projection,",
         "staging, epilogue once; QKV, QKV-post, step-end three times (the three rows with that name are the unrolled steps 0..2); ONE of",
         "`softmax+PV` / `softmax+PV masked` per step (masked: windows on the last row / column of a shifted launch); `softmax-exact(cold)`",
         "only after a failed range guard.\n",
         "| phase | " + " | ".join(c.replace("VALU ", "") for c in classes) + " | VALU total |", "|---|" + "---|" * (len(classes) + 1)]
tot = collections.Counter()
for ph, c in cnt.items():
    valu = sum(v for k, v in c.items() if k.startswith("VALU"))
    lines.append(f"| {ph} | " + " | ".join(str(c.get(k, 0)) for k in classes) + f" | {valu} |")
print("\n".join(lines))
# hot-path estimate: everything except the cold / masked variants and entry/exit
hot = collections.Counter()
for ph, c in cnt.items():
    if "cold" in ph or "masked" in ph or ph in ("entry", "exit"):
        continue
    hot.update(c)
hv = sum(v for k, v in hot.items() if k.startswith("VALU"))
summary = (f"\nHot path (unmasked windows, no fallback), per wave and window pair: {hv} VALU instructions, {hot['MFMA']} MFMA "
           f"=> **{hv / hot['MFMA']:.2f} VALU per MFMA**; by class: " + ", ".join(f"{k.replace('VALU ', '')} {v}" for k, v in hot.most_common() if k.startswith("VALU")) + ".")
print(summary)
if out:
    open(out, "w").write("\n".join(lines) + "\n" + summary + "\n")
