"""Static checks on the generated gfx950 assembly of every kernel that hides asynchronous loads from hipcc, no GPU needed (hipcc
cross-compiles).

csrc/gemm3.hip, wmsa_hg.hip, wmsa_block.hip (through wmsa_common.h) and mlp.hip read LDS - and gemm3.hip also global memory - with
INLINE-ASM loads that are retired by hand-counted `s_waitcnt lgkmcnt(N)` / `vmcnt(N)`.  hipcc believes an inline-asm output is valid
the moment the statement ends, so three things make such a load unsafe (cdna_hip_programming.md section 5.7):
  (1) a register spill or reload between the load and its wait (the destination is stored to scratch before the data lands),
  (2) any other instruction that reads or writes the destination before the wait (a compiler copy, or a destination the compiler could
      prove dead and re-allocated: the hang of round 4's first thin-N GEMM instantiation),
  (3) an address operand that is itself an in-flight destination.
All three are visible in the assembly.  The checker walks each kernel's instruction stream in program order, keeps the ordered list of
in-flight inline-asm loads per counter (LDS loads retire in order on lgkmcnt; loads, stores and LDS-DMA retire in order on vmcnt -
scalar loads, which also count on lgkmcnt and return out of order, only make the hardware wait LONGER than the model), retires them
at every `s_waitcnt`, and fails on any instruction - spill stores and reloads included - that names an in-flight destination."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "small-object-detection-transformers_amd", "csrc")


def _hipcc():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    assert os.path.exists(hipcc), "hipcc is required: these checks guard kernels that would otherwise hang or corrupt memory on the GPU"
    return hipcc


_ASM_CACHE = {}


def asm_of(name, tmp_path_factory):
    if name not in _ASM_CACHE:
        out = str(tmp_path_factory.mktemp("asm") / (name + ".s"))
        cmd = [_hipcc(), "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast", "-Wno-unused-value", "-Wno-inline-asm",
               "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, name + ".hip")]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        _ASM_CACHE[name] = open(out).read()
    return _ASM_CACHE[name]


def kernels_of(asm, pattern):
    """[(mangled name, body)] of the kernels whose name contains `pattern`"""
    out = []
    for m in re.finditer(r"^(_Z[^:\s]*):[^\n]*\n(.*?)\n\s*s_endpgm", asm, re.S | re.M):
        if pattern in m.group(1):
            out.append((m.group(1), m.group(2)))
    return out


def _regset(text):
    regs = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        regs.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", text):
        regs.add(int(a))
    return regs


_VMEM = re.compile(r"^(global_|buffer_|flat_|scratch_)(load|store|atomic)")
_DS = re.compile(r"^ds_")


def check_async_loads(name, body):
    """Walk one kernel.  Returns the number of inline-asm asynchronous loads seen."""
    in_asm = False
    lds, vm = [], []            # in-flight operations in issue order: None (compiler-visible) or the set of asm destination registers
    seen = 0
    for raw in body.split("\n"):
        l = raw.strip()
        if l.startswith(";;#ASMSTART") or l.startswith(";#ASMSTART"):
            in_asm = True
            continue
        if l.startswith(";;#ASMEND") or l.startswith(";#ASMEND"):
            in_asm = False
            continue
        if not l or l[0] in ";." or l.endswith(":"):
            continue
        l = l.split(";")[0].strip()
        if not l or l.endswith(":"):
            continue
        op = l.split(None, 1)[0]
        if op in ("s_branch", "s_setpc_b64", "s_endpgm"):
            # the next instruction is reached only through a label from blocks the linear walk does not follow: the walk restarts there
            # with nothing in flight (a hazard that spans such an edge is not seen - the kernels keep load .. wait in one block)
            lds.clear()
            vm.clear()
            continue
        args = l.split(None, 1)[1] if " " in l else ""
        live = set().union(*[d for d in lds + vm if d]) if (lds or vm) else set()
        if op == "s_waitcnt":
            for cname, q in (("lgkmcnt", lds), ("vmcnt", vm)):
                mm = re.search(cname + r"\((\d+)\)", args)
                if mm:
                    n = int(mm.group(1))
                    del q[:max(0, len(q) - n)]
            continue
        is_load = in_asm and (re.match(r"ds_read|ds_load", op) or re.match(r"(global|buffer)_load_(dword|ushort|ubyte|short)", op)) \
            and " lds" not in l and not op.endswith("_lds") and "_lds_" not in op
        if is_load:
            dst = _regset(args.split(",")[0])
            src = _regset(",".join(args.split(",")[1:]))
            assert not (src & live), f"{name}: `{l}` takes its address from an in-flight destination"
            q, other = (lds, vm) if _DS.match(op) else (vm, lds)
            other_live = set().union(*[d for d in other if d]) if other else set()
            assert not (dst & other_live), f"{name}: `{l}` overwrites a destination that a load of the other counter is still writing"
            # the same queue returns in order: a later load may re-use the destination of an earlier one whose value is dead (the later
            # data lands last); the earlier entry simply stops owning those registers
            for k, d in enumerate(q):
                if d:
                    q[k] = d - dst
            q.append(dst)
            seen += 1
            continue
        hit = _regset(args) & live
        assert not hit, f"{name}: `{l}` names v{sorted(hit)} while an inline-asm load is still writing it"
        if _DS.match(op):
            lds.append(None)
        elif _VMEM.match(op):
            vm.append(None)
    return seen


CASES = [
    # file, kernel-name substring, minimum number of instantiations, each must contain inline-asm loads
    ("gemm3", "gemm_nt3_kernel", 12),
    ("gemm3", "gemm_tn3_kernel", 4),
    ("wmsa_hg", "wmsa_hg_kernel", 4),
    ("wmsa_block", "wmsa_block_kernel", 3),
    ("mlp", "mlp_fwd_kernel", 4),
]


@pytest.mark.parametrize("src,pattern,nmin", CASES)
def test_no_instruction_touches_an_inflight_inline_asm_destination(tmp_path_factory, src, pattern, nmin):
    ks = kernels_of(asm_of(src, tmp_path_factory), pattern)
    assert len(ks) >= nmin, f"{src}.hip: {len(ks)} kernels match {pattern}"
    total = 0
    for name, body in ks:
        total += check_async_loads(f"{src}.hip {name}", body)
    assert total > 0, f"{src}.hip {pattern}: no inline-asm loads found - the checker no longer sees the idiom it guards"


# spill budgets: kernels whose asynchronous reads sit in hot loops must not spill there
SPILL_LIMITS = [
    ("gemm3", "gemm_nt3_kernel", lambda n: 0),
    ("gemm3", "gemm_tn3_kernel", lambda n: 0),
    ("mlp", "mlp_fwd_kernel", lambda n: 0),
    # <SAVE, STAMP>: the training builds of the fused block may spill in the cold exact-softmax fallback only (checked above: never
    # while a read is in flight)
    ("wmsa_hg", "wmsa_hg_kernel", lambda n: 4 if "ILb1E" in n.split("wmsa_hg_kernel")[1][:6] else 0),
    # the direct 3x3 kernels keep 18 weight fragments + a tile prefetch in registers (202-254 VGPRs): a spill lands in their row loops
    ("conv3", "conv3_c64_kernel", lambda n: 0),
    ("conv3", "conv3_c64_wgrad_kernel", lambda n: 0),
    ("conv3", "conv3_n8_", lambda n: 0),
]


@pytest.mark.parametrize("src,pattern,limit", SPILL_LIMITS)
def test_spill_budget(tmp_path_factory, src, pattern, limit):
    ks = kernels_of(asm_of(src, tmp_path_factory), pattern)
    assert ks
    for name, body in ks:
        n = len(re.findall(r"\bscratch_", body))
        assert n <= limit(name), f"{src}.hip {name}: {n} scratch instructions (limit {limit(name)})"
