"""SPP (basics/models/common.py:129-140) on the HIP kernels, token-major: cv1 (1x1 Conv+BN+SiLU) -> MaxPool 5 / 9 / 13
(stride 1, same padding) -> Concat -> cv2 (1x1 Conv+BN+SiLU), forward and hand-written backward.

The three pools are ONE kernel applied in cascade (``sodt_maxpool5_fwd``: 9 = 5 o 5, 13 = 5 o 5 o 5, csrc/pool.hip); the
concat never exists: cv2 reads [a1 | m5 | m9 | m13] as four K-segments of one GEMM, and its input gradient is written
as one [M][4 c_] tensor whose slices the pool backward accumulates through in place.  BatchNorm is the batch-statistics
form of the head's Conv (f64 column sums in the GEMM epilogue, engine.Engine._conv_fwd).

Ties: forward values equal nn.MaxPool2d(9 / 13) exactly.  The backward routes each gradient through the chain of per-stage
first-in-scan-order argmax positions of the 5x5 pools; when the 9x9 / 13x13 window holds several equal maxima this can
pick a different (equally valid) subgradient than torch's single-window first-in-row-major choice.  With distinct
maxima (the tested case, tests/golden/spp.pt) the two agree; a map with exact ties (constant regions, coarse bf16
quantisation after SiLU) gets the same total gradient mass per window placed on another tied element.

model.yaml's head has no SPP (it appears in the SuperYOLO CNN configs this fork can no longer parse, SURVEY.md appendix
C).  A head yaml WITH an SPP row runs through ``Model(cfg)``: the engine's head walk (engine.Engine._check_head /
_spp_fwd / _spp_bwd) places it as a graph node with plan-owned buffers (tests/test_head_graph_gpu.py, pinned against the
reference's own parse_model + forward_once for that head).  This module is the same computation as a STANDALONE operator
(own buffers, direct calls) - what tests/test_spp_gpu.py pins against the reference's ``common.SPP`` class.
"""
from __future__ import annotations

import torch

from . import _lib as L
from . import ops
from .ops import SegSpec

EPS, MOM = 1e-3, 0.03          # initialize_weights, torch_utils.py:150-152


class SPPOp:
    def __init__(self, c1: int, c2: int, B: int, H: int, W: int, dt: torch.dtype, dev):
        self.c1, self.c2, self.c_ = c1, c2, c1 // 2
        self.B, self.H, self.W, self.dt, self.dev = B, H, W, dt, dev
        M, c_ = B * H * W, self.c_
        z = lambda *s, d=dt: torch.zeros(*s, device=dev, dtype=d)
        self.z1, self.cat, self.z2, self.out = z(M, c_), z(M, 4 * c_), z(M, c2), z(M, c2)
        self.arg = [torch.zeros(M, c_, device=dev, dtype=torch.uint8) for _ in range(3)]
        self.st1, self.st2 = z(L.STATS_REPL, 2, c_, d=torch.float64), z(L.STATS_REPL, 2, c2, d=torch.float64)
        self.mr1, self.mr2 = z(2, c_, d=torch.float32), z(2, c2, d=torch.float32)
        self.w = {}

    def _prep(self, p):
        """run-dtype GEMM layouts of the two 1x1 conv weights: [N][K] and [K][N]"""
        for k in ("cv1", "cv2"):
            w32 = p[k + ".conv.weight"].reshape(p[k + ".conv.weight"].shape[0], -1).contiguous().float()
            N, K = w32.shape
            wt32 = torch.empty(K, N, device=self.dev, dtype=torch.float32)
            ops.transpose_f32(w32, wt32, N, K)
            w, wt = torch.empty(N, K, device=self.dev, dtype=self.dt), torch.empty(K, N, device=self.dev, dtype=self.dt)
            ops.cast(w32, w, N * K)
            ops.cast(wt32, wt, N * K)
            self.w[k], self.w[k + "T"] = w, wt

    def _conv_bn_silu(self, p, name, segs, K, Cout, z, stats, mr, y, ldy, training):
        M = self.B * self.H * self.W
        if training:
            ops.zero_(stats)
            ops.gemm_nt(segs, self.w[name], z, M, Cout, K, stats=stats)
            ops.bn_finalize(stats, mr, p[name + ".bn.running_mean"], p[name + ".bn.running_var"], M, Cout, EPS, MOM)
        else:
            ops.gemm_nt(segs, self.w[name], z, M, Cout, K)
            ops.bn_finalize(None, mr, p[name + ".bn.running_mean"], p[name + ".bn.running_var"], M, Cout, EPS, MOM)
        ops.bn_silu_fwd(z, mr, p[name + ".bn.weight"], p[name + ".bn.bias"], y, ldy, M, Cout)

    def forward(self, p: dict, x: torch.Tensor, training: bool = True) -> torch.Tensor:
        """p: cv1/cv2 .conv.weight, .bn.{weight,bias,running_mean,running_var} (f32, device); x [B*H*W][c1] run dtype."""
        B, H, W, c_ = self.B, self.H, self.W, self.c_
        self._prep(p)
        self.x = x
        self._conv_bn_silu(p, "cv1", [SegSpec(x)], self.c1, c_, self.z1, self.st1, self.mr1, self.cat, 4 * c_, training)   # a1 = cat[:, :c_]
        for i in range(3):                                   # m5, m9, m13 into the next slices of the concat buffer
            ops.maxpool5_fwd(self.cat, self.cat, self.arg[i] if training else None, B, H, W, c_, ldx=4 * c_, ldy=4 * c_,
                             x_off=i * c_, y_off=(i + 1) * c_)
        self._conv_bn_silu(p, "cv2", [SegSpec(self.cat)], 4 * c_, self.c2, self.z2, self.st2, self.mr2, self.out, self.c2, training)
        return self.out

    def _bn_bwd(self, p, name, dy, z, mr, Cout, g):
        M = self.B * self.H * self.W
        red = torch.zeros(2, Cout, device=self.dev, dtype=torch.float64)
        dz = torch.empty(M, Cout, device=self.dev, dtype=self.dt)
        ops.bn_silu_bwd_reduce(dy, dy.shape[-1], z, mr, p[name + ".bn.weight"], p[name + ".bn.bias"], red, M, Cout)
        ops.bn_silu_bwd_apply(dy, dy.shape[-1], z, mr, p[name + ".bn.weight"], p[name + ".bn.bias"], red, dz,
                              g[name + ".bn.weight"], g[name + ".bn.bias"], M, Cout)
        return dz

    def backward(self, p: dict, dout: torch.Tensor, g: dict) -> torch.Tensor:
        """g: f32 gradient accumulators with the parameter names (zeroed by the caller); returns dx [M][c1]."""
        B, H, W, c_, M = self.B, self.H, self.W, self.c_, self.B * self.H * self.W
        dz2 = self._bn_bwd(p, "cv2", dout, self.z2, self.mr2, self.c2, g)
        ops.gemm_tn(dz2, [SegSpec(self.cat)], g["cv2.conv.weight"].view(self.c2, 4 * c_), M, self.c2, 4 * c_)
        dcat = torch.empty(M, 4 * c_, device=self.dev, dtype=self.dt)
        ops.gemm_nt([SegSpec(dz2)], self.w["cv2T"], dcat, M, 4 * c_, self.c2)
        for i in (2, 1, 0):                                  # d m13 -> m9 -> m5 -> a1, accumulated into the lower slice
            ops.maxpool5_bwd(dcat, self.arg[i], dcat, B, H, W, c_, lddy=4 * c_, lddx=4 * c_, dy_off=(i + 1) * c_, dx_off=i * c_,
                             accumulate=True)
        da1 = dcat[:, :c_].contiguous()
        dz1 = self._bn_bwd(p, "cv1", da1, self.z1, self.mr1, c_, g)
        ops.gemm_tn(dz1, [SegSpec(self.x)], g["cv1.conv.weight"].view(c_, self.c1), M, c_, self.c1)
        dx = torch.empty(M, self.c1, device=self.dev, dtype=self.dt)
        ops.gemm_nt([SegSpec(dz1)], self.w["cv1T"], dx, M, self.c1, c_)
        return dx
