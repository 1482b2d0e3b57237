"""Static checks on the generated gfx950 assembly of the pipelined GEMM (csrc/gemm3.hip), no GPU needed (hipcc cross-compiles).

The kernel fetches its epilogue operand with inline-asm `global_load_dwordx4` issued a K-step ahead and retired by a counted
`s_waitcnt vmcnt(7)`.  hipcc believes an inline-asm output is valid the moment the statement ends, so two things make such a load
unsafe: (1) register spills around it (the destination is stored to scratch before the data lands: DESIGN.md section 4.1a), and (2)
a destination the compiler can prove unused (its registers are re-allocated while the load is in flight and the arriving data
clobbers whatever lives there - seen in the thin-N instantiation before its loops were cut to the live column groups).  Both are
visible in the assembly: no `scratch_` access in any gemm_nt3_kernel instantiation, and between the first prefetch load and the
counted wait no instruction may name a register that an earlier prefetch load is still writing."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "small-object-detection-transformers_amd", "csrc", "gemm3.hip")


@pytest.fixture(scope="module")
def gemm3_asm(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    out = str(tmp_path_factory.mktemp("asm") / "gemm3.s")
    cmd = [hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast", "-Wno-unused-value", "-Wno-inline-asm",
           "-S", "--cuda-device-only", "-o", out, SRC]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(out).read()


def _regs(text):
    regs = set()
    for a, b in re.findall(r"v\[(\d+):(\d+)\]", text):
        regs.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", text):
        regs.add(int(a))
    return regs


def test_pipelined_nt_gemm_has_no_spills_and_no_touched_inflight_destinations(gemm3_asm):
    kernels = list(re.finditer(r"^(_ZN[^:\n]*gemm_nt3_kernelILi(\d+)ELb([01])ELi(\d)E[^:\n]*):.*?s_endpgm", gemm3_asm, re.S | re.M))
    assert len(kernels) >= 12
    checked = 0
    for m in kernels:
        body = m.group(0)
        name = f"gemm_nt3_kernel<{m.group(2)}, {m.group(3)}, {m.group(4)}>"
        assert "scratch_" not in body, f"{name} spills registers"
        k = body.split("\n")
        loads = [i for i, l in enumerate(k) if re.search(r"global_load_dwordx4 v\[", l)]
        if not loads:
            continue
        waits = [i for i, l in enumerate(k) if "vmcnt(7)" in l and i > loads[-1]]
        assert waits, name
        live = set()
        for i in range(loads[0], waits[0]):
            l = k[i].strip()
            if not l or l[0] in ";.":
                continue
            mm = re.match(r"global_load_dwordx4 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\]", l)
            if mm:
                assert not (set(range(int(mm.group(3)), int(mm.group(4)) + 1)) & live), f"{name}: address in an in-flight destination: {l}"
                live.update(range(int(mm.group(1)), int(mm.group(2)) + 1))
                continue
            parts = l.split(None, 1)
            if len(parts) == 2:
                hit = _regs(parts[1]) & live
                assert not hit, f"{name}: `{l}` touches v{sorted(hit)} while a prefetch load is still writing it"
        checked += 1
    assert checked >= 6          # the RESID / DGELU / DRELU instantiations, full and thin


def test_fused_block_kernel_spill_budget(tmp_path):
    """csrc/wmsa_hg.hip reads LDS with inline-asm `ds_read` retired by counted `lgkmcnt` waits: a build that spills around them
    computes garbage (36 spilled registers once broke the multi-iteration save form).  The shipped state is 0
    scratch instructions in the inference instantiations and <= 4 (cold exact-softmax fallback) in the training ones."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    out = str(tmp_path / "wmsa_hg.s")
    src = os.path.join(ROOT, "small-object-detection-transformers_amd", "csrc", "wmsa_hg.hip")
    r = subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast", "-Wno-unused-value",
                        "-Wno-inline-asm", "-S", "--cuda-device-only", "-o", out, src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = open(out).read()
    ks = list(re.finditer(r"^(_Z[^:\n]*wmsa_hg_kernelILb([01])ELb([01])E[^:\n]*):.*?s_endpgm", asm, re.S | re.M))
    assert len(ks) == 4
    for m in ks:
        n = len(re.findall(r"scratch_", m.group(0)))
        limit = 4 if m.group(2) == "1" else 0            # <SAVE, STAMP>: training builds may spill in the cold path only
        assert n <= limit, f"wmsa_hg_kernel<{m.group(2)}, {m.group(3)}>: {n} scratch instructions (limit {limit})"
