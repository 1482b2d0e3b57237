"""Instruction census of the largest inner loop of a kernel in a hipcc -S listing.  usage: python tools/asm_loop_census.py file.s kernel_substring"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
pat = sys.argv[2]
kern = [k for k in re.split(r"\n(?=_Z\w+:)", s) if k.startswith("_Z") and pat in k.split(":")[0] and "v_mfma" in k]
k = kern[int(sys.argv[3]) if len(sys.argv) > 3 else 0]
lines = k.split("\n")
print(lines[0])
labels = {l.split(":")[0]: i for i, l in enumerate(lines) if re.match(r"\.LBB\d+_\d+:", l)}
best = None
for i, l in enumerate(lines):
    m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        span = (labels[m.group(1)], i)
        n = sum(1 for x in lines[span[0]:span[1]] if "v_mfma" in x)
        if best is None or n > best[0]:
            best = (n, span)
n, (i0, i1) = best
body = lines[i0:i1 + 1]
ins = [l.strip().split()[0] for l in body if l.strip() and not l.strip().startswith((".", ";")) and not l.strip().endswith(":")]
c = Counter(ins)
valu = sum(v for k_, v in c.items() if k_.startswith("v_") and not k_.startswith("v_mfma"))
print(f"loop {lines[i0].split(':')[0]}: {len(ins)} instructions, {c.get('v_mfma_f32_16x16x32_bf16', 0) + c.get('v_mfma_f32_32x32x16_bf16', 0)} MFMA, {valu} VALU, "
      f"{sum(v for k_, v in c.items() if k_.startswith('ds_'))} DS, {sum(v for k_, v in c.items() if k_.startswith('s_nop'))} s_nop")
for k_, v in c.most_common(24):
    print(f"  {v:4d} {k_}")
