"""Kernel table of the Model(sr=True) step from a rocprofv3 --kernel-trace database (rocpd .db) of tools/sr_step.py.
Usage: python tools/sr_profile_table.py <results.db> [n_steps_incl_warmup=5]"""
import collections, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows = list(db.execute("select name, start, end, grid_x from kernels order by start"))
i0 = next(i for i, r in enumerate(rows) if "bilinear" in r[0])      # first SR kernel: the sr=True model's launches follow
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e, gx in rows[i0:]:
    k = re.sub(r"^void ", "", n)
    k = re.sub(r"\(anonymous namespace\)::", "", k)
    k = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", k)[:100]
    agg[k][0] += 1
    agg[k][1] += (e - s) / 1e6
tot = sum(v[1] for v in agg.values()) / steps
print(f"Kernel time per step: {tot:.1f} ms.\n\n| ms / step | launches / step | kernel |\n|---:|---:|---|")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"| {v[1] / steps:.2f} | {v[0] / steps:.1f} | `{k}` |")
