"""CPU-side checks: the C-ABI library loads and exports every symbol include/sodt_hip.h declares, the ctypes
records match the header, the Model boundary mirrors the reference's parameter names, and host logic."""
import copy
import ctypes
import importlib
import os
import pickle
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "sodt_hip.h")).read()
    return sorted(set(re.findall(r"^(?:int|long|const char\*)\s+(sodt_\w+)\s*\(", src, flags=re.M)))


def test_library_exports_every_declared_symbol(pkg):
    L = pkg._lib
    lib = L.load()
    declared = header_symbols()
    assert len(declared) >= 24
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/sodt_hip.h but not exported by libsodt_hip.so"
    bound = set(L.exported_symbols())
    assert set(declared) == bound, (sorted(set(declared) - bound), sorted(bound - set(declared)))
    assert lib.sodt_version().startswith(b"sodt_hip")


def test_ctypes_record_layout(pkg):
    L = pkg._lib
    assert ctypes.sizeof(L.Seg) == 40 and ctypes.sizeof(L.ASpec) == 40 * 9 + 16
    assert ctypes.sizeof(L.PrepDesc) == 48
    assert L.GemmArgs.a.offset == 0 and L.GemmArgs.W.offset == ctypes.sizeof(L.ASpec)
    assert L.GemmTnArgs.x.offset == 16


def test_missing_library_fails_loudly(pkg, monkeypatch):
    L = pkg._lib
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", "/nonexistent/libsodt_hip.so")
    with pytest.raises(ImportError):
        L.load()


@pytest.mark.parametrize("var", ["SODT_LIB_PATH", "SODT_HG_DBG", "SODT_WMSA_ONE_WAVE"])
def test_bench_refuses_diagnostic_environment(var):
    """a timed number must not come from an ablated kernel or a swapped library."""
    import subprocess
    import sys
    env = dict(os.environ, **{var: "1"})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert var in r.stderr and "refusing" in r.stderr
    assert r.stdout.strip() == ""          # no JSON line


def test_library_reads_no_environment_switch():
    """The ablation paths of the fused kernel are compile-time macros of the tools/exp A/B build, not run-time switches."""
    csrc = os.path.join(ROOT, "small-object-detection-transformers_amd", "csrc")
    for f in os.listdir(csrc):
        assert "getenv" not in open(os.path.join(csrc, f)).read(), f


def test_version_reports_library_override(pkg, ops, monkeypatch):
    assert pkg._lib.overrides() == {} or "SODT_LIB_PATH" in os.environ
    monkeypatch.setenv("SODT_HG_DBG", "3")
    assert pkg._lib.overrides().get("SODT_HG_DBG") == "3"
    assert "SODT_HG_DBG=3" in ops.version()


@pytest.fixture(scope="module")
def M(pkg):
    return importlib.import_module("small-object-detection-transformers_amd.model")


def test_model_mirrors_reference_state_dict(M):
    from oracle import ref_torch as R
    for cfg in ("model.yaml", "SRyolo_MF.yaml"):
        m = M.Model(cfg, input_mode="RGB+IR", ch_steam=3, ch=128, nc=8)
        assert sum(p.numel() for p in m.parameters()) == 22007851
        sd = m.state_dict()
        assert len(sd) == 273
        spec = R.state_dict_spec(512, 8)
        for k, shape in spec.items():
            assert tuple(sd[k].shape) == tuple(shape), k
        extra = set(sd) - set(spec)
        assert all(("relative_position_index" in k or "attn_mask" in k or "num_batches_tracked" in k) for k in extra), extra
        det = m.detect[-1]
        assert det.nl == 1 and det.na == 3 and det.no == 13 and float(m.stride[0]) == 4.0
        assert torch.allclose(det.anchors.view(-1), torch.tensor([10, 13, 16, 30, 33, 23.]) / 4)
        assert m.yaml["nc"] == 8 and hasattr(m, "yaml_file")
        assert m.image_encoder.stage1[1].attn_mask.shape == (256, 64, 64)   # kept for checkpoint parity
        assert m.image_encoder.stage3[0].attn.relative_position_bias_table.shape == (3969, 12)
        bn = m.detect[0].bn
        assert bn.eps == 1e-3 and bn.momentum == 0.03


def test_model_has_no_cpu_path_and_is_copyable(M):
    m = M.Model("model.yaml", input_mode="RGB+IR", ch_steam=3, ch=128, nc=8)
    x = torch.zeros(1, 3, 64, 64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(x, x, "RGB+IR")
    with pytest.raises(RuntimeError):
        m.detect[0](x)                     # containers never compute in torch
    with pytest.raises(NotImplementedError):
        m(x, x, "RGB")
    m2 = copy.deepcopy(m)                  # ModelEMA
    assert m2._engine is None and len(m2.state_dict()) == 273
    m3 = pickle.loads(pickle.dumps(m))     # checkpoints pickle the module object
    assert torch.equal(m3.detect[8].m[0].bias, m.detect[8].m[0].bias)
    # sr=True builds the reference's DeepLab(4, c1, c2) parameter tree (model.py:109-117); its 82 tensors follow the detector's
    ms = M.Model("model.yaml", input_mode="RGB+IR", ch_steam=3, ch=128, nc=8, sr=True)
    keys = list(ms.state_dict())
    assert len(keys) == 273 + 82 and keys[273] == "model_up.sr_decoder.conv1.weight" and (ms.l1, ms.l2) == (4, 8)
    with pytest.raises(RuntimeError, match="inside the MI355X engine"):
        ms.model_up(x, x)
    with pytest.raises(NotImplementedError):
        M.Model("model.yaml", input_mode="RGB+IR", sr=True, factor=4)


def test_fuse_matches_reference_formula(M):
    m = M.Model("model.yaml", input_mode="RGB+IR", ch_steam=3, ch=128, nc=8)
    c = m.detect[3].m[0].cv2
    with torch.no_grad():
        c.bn.running_mean.uniform_(-0.5, 0.5); c.bn.running_var.uniform_(0.5, 2); c.bn.weight.uniform_(0.5, 1.5); c.bn.bias.uniform_(-1, 1)
    x = torch.randn(2, c.conv.in_channels, 6, 6)
    with torch.no_grad():
        z = torch.nn.functional.conv2d(x, c.conv.weight, None, padding=1)
        want = torch.nn.functional.batch_norm(z, c.bn.running_mean, c.bn.running_var, c.bn.weight, c.bn.bias, False, 0.0, 1e-3)
    m.fuse()
    assert not hasattr(c, "bn") and c.conv.bias is not None
    with torch.no_grad():
        got = torch.nn.functional.conv2d(x, c.conv.weight, c.conv.bias, padding=1)
    assert float((got - want).abs().max()) < 1e-4


def test_parse_model_rejects_out_of_scope_modules(M):
    cfg = dict(nc=8, depth_multiple=0.33, width_multiple=0.5, anchors=[[10, 13, 16, 30, 33, 23]],
               backbone=[[-1, 1, "MF", [3]]], head=[])
    with pytest.raises(NotImplementedError):
        M.Model(cfg, input_mode="RGB+IR")


def test_host_helpers(ops):
    assert ops.tn_splits(524288, 768, 192) == 85 and ops.tn_splits(100, 768, 192) == 1
    assert ops.tn_splits(32768, 3072, 768) == 10
    assert ops.tn_splits(524288, 192, 768, True) == 85 and ops.tn_splits(131072, 384, 384, True) == 64
    assert ops.tn_splits(512, 192, 768, True) == ops.tn_splits(512, 192, 768)      # short M keeps the f32-path rule
    s = ops.SegSpec(torch.zeros(4, 96), 32, 64)
    assert (s.ld, s.klen, s.coff) == (96, 32, 64)


def test_sr_direct_conv_bias_operand(pkg):
    """advisor r5: the direct 64 -> <= 8 convolution reads 8 bias floats; at cout == 8 there is no zero-padded copy (np_ == cout) and
    the parameter itself must be passed (a None there silently dropped the bias), at cout < 8 the padded copy, without a bias None."""
    import importlib
    import types
    sr = importlib.import_module("small-object-detection-transformers_amd.sr")
    b8, pad = torch.arange(8.0), torch.zeros(8)
    assert sr.SRBranch._bias8(types.SimpleNamespace(bias=b8, bias_pad=None)) is b8
    assert sr.SRBranch._bias8(types.SimpleNamespace(bias=torch.arange(4.0), bias_pad=pad)) is pad
    assert sr.SRBranch._bias8(types.SimpleNamespace(bias=None, bias_pad=None)) is None


def test_engine_names_the_input_sizes_it_refuses(M):
    """Window padding (round 6): the geometry helper must refuse, by name, the single-window stages whose rows 64-token tiles do not
    cover; it accepts 8 / 16 / 32-token windows and leaves larger grids to the padded path."""
    import importlib
    import types
    E = importlib.import_module("small-object-detection-transformers_amd.engine")
    tab = lambda ws: types.SimpleNamespace(relative_position_bias_table=torch.zeros((2 * ws - 1) ** 2, 12), window_size=(ws, ws))
    geo = E.Engine._block_geo
    assert geo(None, types.SimpleNamespace(window_size=32, shift_size=0, attn=tab(16)), 16, 16) == (16, 0)     # clamped to one window
    assert geo(None, types.SimpleNamespace(window_size=32, shift_size=0, attn=tab(32)), 40, 40) == (32, 0)     # padded by the caller
    with pytest.raises(NotImplementedError, match="ONE 20x20 window"):
        geo(None, types.SimpleNamespace(window_size=32, shift_size=0, attn=tab(20)), 20, 20)
