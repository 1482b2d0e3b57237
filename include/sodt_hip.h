/* C ABI of libsodt_hip.so -- the MI355X (gfx950) kernels beneath the reference's
 * Python nn.Module boundary (Model / ImageEncoderViT / C3 / Detect).
 *
 * The reference (Bissmella/Small-object-detection-transformers) has no FFI: every op
 * on its hot path is a torch.nn call.  Each entry point below therefore cites the
 * reference lines whose ATen op sequence it replaces (paths relative to the
 * reference root).  Conventions: raw device pointers + explicit sizes, caller-owned
 * memory (workspaces included), a hipStream_t argument, int return (0 = ok,
 * SODT_EINVAL = shape/alignment the kernel does not support -- nothing launched),
 * no allocation / synchronisation / exceptions inside; safe to capture in a hipGraph.
 *
 * dtype: 0 = float32 (parity path, exact-f32 MFMA), 1 = bfloat16 (throughput path,
 * f32 accumulate).  Activations are token-major ("NHWC"): row = (b*H + y)*W + x,
 * channels contiguous.  Parameters handed to the kernels are the "prepared" copies
 * produced by sodt_prep_weights (cast to the run dtype, GEMM layouts [N][K]).
 */
#ifndef SODT_HIP_H
#define SODT_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* sodt_stream_t;   /* == hipStream_t */

#define SODT_F32 0
#define SODT_BF16 1
#define SODT_MAX_SEG 9

/* One K-segment of a GEMM's left operand: a (possibly spatially mapped) view of a
 * token-major tensor.  For output row m = (b, y, x) on the (Ho, Wo) grid the source
 * pixel is ((y*mul + dy) >> shr, (x*mul + dx) >> shr) on the (Hi, Wi) grid; rows
 * that fall outside read as zeros.  This one mechanism expresses the 2x2 conv of the
 * enhanced Swin MLP (backbone_vit.py:892-905), the 3x3 Bottleneck conv
 * (common.py:61), PatchMerging's 2x2 gather (backbone_vit.py:848-855),
 * nn.Upsample(x2, nearest) + Concat (models/model.yaml:66-72), channel concat of the
 * necks (backbone_vit.py:239,268) and every transposed (backward) form of them. */
typedef struct {
  const void* p;   /* base pointer, already offset to the first channel used */
  int ld;          /* row stride in elements */
  int klen;        /* channels taken from this segment (multiple of 16 bytes) */
  int dy, dx;      /* tap offset */
  int mul, shr;    /* stride multiply / upsample shift */
  int Hi, Wi;      /* source grid */
} sodt_seg;

typedef struct {
  sodt_seg s[SODT_MAX_SEG];
  int nseg;
  int spatial;     /* 0: source row == output row for every segment */
  int Ho, Wo;      /* output grid (rows = B*Ho*Wo) when spatial */
} sodt_aspec;

/* epilogue flags of sodt_gemm_nt */
#define SODT_EPI_BIAS 1          /* + bias[n] (f32) */
#define SODT_EPI_RESID 2         /* + R[m % rmod or m][n] */
#define SODT_EPI_GELU_DUAL 4     /* C = pre-activation, C2 = GELU(erf) of it */
#define SODT_EPI_DGELU 8         /* value *= gelu'(aux[m][n]) */
#define SODT_EPI_STATS 16        /* stats[r][0][n] += sum_m v, stats[r][1][n] += sum_m v*v (f64 atomics); the buffer holds
                                  * SODT_STATS_REPL replicas r of [2][N] (workgroups spread over them: same-address atomics
                                  * serialise in L2); sodt_bn_finalize sums the replicas */
#define SODT_STATS_REPL 16
#define SODT_EPI_AFFINE_SILU 32  /* v = silu(v*scale[n] + shift[n])  (fused / eval-mode Conv) */
#define SODT_EPI_DETECT 64       /* store f32 to (B, na, HW, no): Detect's view+permute */
#define SODT_EPI_OUT_F32 128
#define SODT_EPI_GELU 256        /* C = GELU(erf) of the value (after bias): fc1 when only the activation is kept */
#define SODT_EPI_DGELU_RC 512    /* two-phase K (bf16 pipelined kernel only): the first K/2 columns of A / W recompute the
                                  * pre-activation h = A1 W1^T (+ bias), the second K/2 give dh_act = A2 W2^T;
                                  * C = dh_act * gelu'(h).  Backward of Linear-GELU-Linear without saving h
                                  * (backbone_vit.py:886-904): A = [xn | dy], W = [fc1.weight | fc2.weight^T] */     /* C is float regardless of dtype */
#define SODT_EPI_RELU 2048       /* v = max(v, 0) after the bias: the ReLUs of the super-resolution branch (sr_decoder_noBN_noD.py:30-38,
                                  * edsr.py:44) */
#define SODT_EPI_DRELU 4096      /* v = aux[m][n] > 0 ? v : 0 - the gradient through a ReLU whose OUTPUT is aux (applied before RESID) */
/* (1024 was SODT_EPI_LNBWD, rounds 4-5: the LayerNorm backward as a GEMM epilogue - built, pinned, slower than the two launches it
 * replaced (profiles/r04_lnfold_ab.md) and removed in round 6; the bit is not reused) */

typedef struct {
  sodt_aspec a;                 /* A [M][K] as K-segments */
  const void* W; int ldw;       /* W [N][K], run dtype */
  void* C; int ldc;
  void* C2; int ldc2;
  const float* bias;
  const void* R; int ldr; int rmod;
  const void* aux; int ldaux;
  double* stats;                /* [2][N] */
  const float* scale; const float* shift;
  int M, N, K, flags;
  int oscatter, omul, ody, odx, OH, OW;   /* optional output-row scatter (PatchMerging backward) */
  int det_na, det_no, det_hw;
} sodt_gemm_args;

/* C = epilogue(A @ W^T): every nn.Linear / 1x1 / 2x2 / 3x3 Conv2d forward and every
 * input-gradient GEMM of the path (backbone_vit.py:968,990,886-904,857,213,268-270;
 * common.py:43,49; model.py:53). */
int sodt_gemm_nt(const sodt_gemm_args* g, int dtype, sodt_stream_t st);
/* test hook: 1 forces the K-loop tile kernel even where the A-stationary kernel applies, 0 = automatic */
int sodt_gemm_set_variant(int force_tiled);

typedef struct {
  const void* dY; int ldy;      /* [M][N] run dtype; ldy >= N rounded up to 16 bytes (pad columns must be zero) */
  sodt_aspec x;                 /* X [M][K] as K-segments (same row mapping as the forward A) */
  float* dW; int lddw;          /* [N][K] f32, accumulated with atomics (zero it first) */
  float* dbias;                 /* [N] f32 or NULL: += column sums of dY */
  int M, N, K;
  int kperm_c, kperm_t;         /* if kperm_t > 1: column k = tap*C + ci is stored at ci*T + tap (torch conv layout) */
  int splits;                   /* M is cut into this many slices (grid.y) */
  float* partial;               /* optional f32 scratch of >= splits*N*K floats (16-byte aligned): the bf16 pipelined kernel then
                                 * writes one partial tile per slice with plain stores and a second kernel adds their sum to dW,
                                 * instead of splits*N*K contended atomics.  NULL / too small: atomics */
  long partial_floats;
} sodt_gemm_tn_args;

/* dW += dY^T @ X (+ dbias): every weight gradient of the path (autograd of the
 * reference ops listed above). */
int sodt_gemm_tn(const sodt_gemm_tn_args* g, int dtype, sodt_stream_t st);

/* LayerNorm over the last dim, eps 1e-5 (backbone_vit.py:1090,1128,858).  y and stats
 * (mean, rstd per row, f32 [M][2]) are written; C % (16 bytes) == 0. */
int sodt_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats,
                       int M, int C, int dtype, sodt_stream_t st);
/* dx = [dres +] LN'(dy); dgamma/dbeta (f32) accumulated with atomics. */
int sodt_layernorm_bwd(const void* dy, const void* x, const float* stats, const float* gamma,
                       const void* dres, void* dx, float* dgamma, float* dbeta,
                       int M, int C, int dtype, sodt_stream_t st);

/* Window multi-head self-attention core on a natural-order QKV tensor
 * (backbone_vit.py:968-989 + :1094-1124): cyclic shift, window partition, q*scale@k^T,
 * relative-position bias, -100 shift mask, softmax, @v, un-partition, un-shift are all
 * index arithmetic inside the kernel.  qkv [B*H*W][3C], out [B*H*W][C], lse f32
 * [B*H*W][heads], bias_t f32 [heads][(2ws-1)^2] (transposed table). */
int sodt_window_attn_fwd(const void* qkv, const float* bias_t, void* out, float* lse,
                         int B, int H, int W, int C, int heads, int ws, int shift,
                         int dtype, sodt_stream_t st);
/* dqkv from dout (recomputes P); dbias_t accumulated with atomics; dq_acc is an f32
 * scratch of B*H*W*(C + heads) floats needed only when ws*ws > 64; its first B*H*W*C
 * floats must be zero on entry and are left zero on exit. */
int sodt_window_attn_bwd(const void* qkv, const float* bias_t, const void* out, const void* dout,
                         const float* lse, void* dqkv, float* dbias_t, float* dq_acc,
                         int B, int H, int W, int C, int heads, int ws, int shift,
                         int dtype, sodt_stream_t st);

/* ---- super-resolution auxiliary branch (deeplabedsr.py:35-73): data movement between its convolutions, token-major [B][H][W][C] ----
 * sodt_bilinear_up2_fwd/bwd: F.interpolate(x, size = (2H, 2W), mode = 'bilinear', align_corners = True) (sr_decoder_noBN_noD.py:37) and
 *   its adjoint (dx = sum over the output pixels a source pixel contributes to; written, not accumulated).  y / dy may be a
 *   column slice of a wider buffer (row stride ldy / lddy elements, pointer already at the slice); relu_out != NULL: x was the
 *   output of a ReLU ([B*H*W][C]) and dx is zeroed where it was not positive.
 * sodt_pixel_shuffle2: nn.PixelShuffle(2) (edsr.py:23): out[b][2y+i][2x+j][c] = in[b][y][x][4c + 2i + j], in [B][H][W][4C] ->
 *   out [B][2H][2W][C]; inverse != 0 runs the adjoint (= inverse permutation) out -> in.
 * sodt_add_rows: dst[m][dcol .. dcol + C) += src[m][scol .. scol + C) (gradient accumulation where a feature has two consumers); C, the
 *   leading dimensions and the column offsets are multiples of 16 bytes and both pointers 16-byte aligned - except ONE f32 row (M == 1)
 *   of fewer than 64 elements, which may have any width (the bias gradient of the 3-channel closing convolution).
 * sodt_nchw_f32_from_rows / sodt_rows_from_nchw_f32: y (B, C, H, W) f32 <-> token-major rows [B*H*W][ld] run dtype (C <= ld, the
 *   pad columns are written as zeros): the branch's output / its incoming gradient at the model boundary. */
int sodt_bilinear_up2_fwd(const void* x, void* y, int ldy, int B, int H, int W, int C, int dtype, sodt_stream_t st);
int sodt_bilinear_up2_bwd(const void* dy, int lddy, void* dx, const void* relu_out, int B, int H, int W, int C, int dtype,
                          sodt_stream_t st);
int sodt_pixel_shuffle2(const void* in, void* out, int B, int H, int W, int C, int inverse, int dtype, sodt_stream_t st);
int sodt_add_rows(void* dst, int ldd, int dcol, const void* src, int lds, int scol, long M, int C, int dtype, sodt_stream_t st);
int sodt_nchw_f32_from_rows(const void* rows, int ld, float* y, int B, int C, int H, int W, int dtype, sodt_stream_t st);
int sodt_rows_from_nchw_f32(const float* y, void* rows, int ld, int B, int C, int H, int W, int dtype, sodt_stream_t st);

/* ---- EDSR's closing convolution (edsr.py:81-84: conv(n_feats = 64, num_channels, 3) = nn.Conv2d(64, ch, 3, padding = 1), edsr.py:9-12),
 * forward and both gradients, on token-major rows (csrc/conv3.hip).  cout <= 8 output channels; bf16 only (SODT_EINVAL otherwise: the
 * float32 path runs the convolution as nine K-segments of sodt_gemm_nt / sodt_gemm_tn).  Each launch moves its tensors once.
 *   fwd:   y [B*H*W][8] = x [B*H*W][64] (*) w + bias; w [8][9*64] = [n][tap*64 + c] (tap = 3 ky + kx; rows >= cout zero), bias f32[8] or NULL.
 *          y_nchw != NULL: the result goes to (B, cout, H, W) float32 instead (the branch's output at the model boundary,
 *          deeplabedsr.py:73), unrounded; y is then unused and may be NULL.
 *   dgrad: dx [B*H*W][64] = dy (*)^T wT;  wT [64][9*8] = [c][tap*8 + n] (columns >= cout zero).  dy [B*H*W][8] bf16 (columns >= cout zero),
 *          or dy_nchw != NULL: (B, cout <= 4, H, W) float32, rounded to bf16 as it is read.
 *   wgrad: dw [cout][64][3][3] (the torch layout) += dy^T x(taps), db [cout] += column sums of dy (db may be NULL); dy / dy_nchw as above;
 *          scratch: sodt_conv3x3_c64n8_wgrad_scratch_bytes() bytes of f32 (per-workgroup partials, summed in a fixed order by a second launch). */
int sodt_conv3x3_c64n8_fwd(const void* x, const void* w, const float* bias, void* y, float* y_nchw, int B, int H, int W, int cout, int dtype,
                           sodt_stream_t st);
int sodt_conv3x3_c64n8_dgrad(const void* dy, const float* dy_nchw, const void* wT, void* dx, int B, int H, int W, int cout, int dtype,
                             sodt_stream_t st);
int sodt_conv3x3_c64n8_wgrad(const void* dy, const float* dy_nchw, const void* x, float* dw, float* db, float* scratch, int B, int H, int W,
                             int cout, int dtype, sodt_stream_t st);
long sodt_conv3x3_c64n8_wgrad_scratch_bytes(void);

/* ---- the 3x3 convolutions of EDSR at 64 input channels and 64 output channels per launch (edsr.py:34-53 ResBlock: conv -> ReLU -> conv,
 * + x; :64-70 head and body-closing convolutions; :14-24 Upsampler: conv(64 -> 256) + nn.PixelShuffle(2) as four launches), token-major
 * rows, bf16 only (csrc/conv3.hip): the input tile crosses the CU once, the weights live in registers.
 *   sodt_conv3x3_c64_fwd: y [B*H*W][64] = epilogue(x [B*H*W][64] (*) w); w [64][9*64] = [n][tap*64 + c].  flip != 0 mirrors the taps
 *     (tap' = 8 - tap): with w = the transposed weights [c][tap*64 + n] and x = dy this is the input gradient.  flags: any of
 *     SODT_EPI_BIAS (bias f32), SODT_EPI_RELU, and ONE of SODT_EPI_DRELU (aux [B*H*W][64]: keep where aux > 0) / SODT_EPI_RESID (resid
 *     [B*H*W][64]; may be y itself), applied in that order as in sodt_gemm_nt; anything else is SODT_EINVAL.
 *   sodt_conv3x3_c64_wgrad: dw [64][64][3][3] (torch layout) += dy^T x(taps), db [64] += column sums of dy (or NULL); scratch:
 *     sodt_conv3x3_c64_wgrad_scratch_bytes() bytes (per-workgroup partials, summed in a fixed order by a second launch).
 *   geo (NULL = identity): H x W is the grid the kernel walks.  Output channel n uses weight / bias (and dw / db) row
 *     w_row_stride * n + w_row_off; the input pixel (y, x) is read at (in_mul y + in_i, in_mul x + in_j) of an (in_mul H) x (in_mul W)
 *     tensor; the output pixel - and the epilogue operand, and dy of the weight gradient - at (out_mul y + out_i, out_mul x + out_j) of
 *     an (out_mul H) x (out_mul W) tensor; in_mul, out_mul in {1, 2}.  PixelShuffle(2) of a 64 -> 256 convolution = four launches with
 *     (w_row_stride, w_row_off, out_mul, out_i, out_j) = (4, 2 i + j, 2, i, j); its gradients read the fine gradient through the same map. */
typedef struct {
  int w_row_stride, w_row_off;
  int in_mul, in_i, in_j;
  int out_mul, out_i, out_j;
} sodt_conv3_geo;
int sodt_conv3x3_c64_fwd(const void* x, const void* w, const float* bias, const void* resid, const void* aux, void* y, int B, int H, int W,
                         int flags, int flip, const sodt_conv3_geo* geo, int dtype, sodt_stream_t st);
int sodt_conv3x3_c64_wgrad(const void* dy, const void* x, float* dw, float* db, float* scratch, int B, int H, int W, const sodt_conv3_geo* geo,
                           int dtype, sodt_stream_t st);
long sodt_conv3x3_c64_wgrad_scratch_bytes(void);

/* ---- fused W-MSA / SW-MSA half of a Swin block (csrc/wmsa_block.hip) -------------------------------------------
 * x_mid = x + Proj(WindowAttention(LN1(x))) and xn2 = LN2(x_mid) in ONE launch: SwinTransformerBlock.forward
 * backbone_vit.py:1084-1128 up to the MLP, with WindowAttention.forward :961-992, window_partition / unpartition
 * :619-672 and both torch.roll :1096,1118 as index arithmetic.  Built for the stage-1 geometry of model.yaml
 * (backbone_vit.py:117-133): C == 192, heads == 12 (head_dim 16), ws == 8, shift in [0, 8); H, W multiples of 8.
 * Any other shape returns SODT_EINVAL (callers fall back to sodt_layernorm_fwd + sodt_gemm_nt + sodt_window_attn_fwd).
 *
 * sodt_wmsa_pack: the block's raw f32 parameters -> the kernel's streaming order (per head: Wq/Wk/Wv rows and the
 *   Wproj column slice as MFMA fragments, the head's relative-position table x log2 e, q/k/v bias; then proj bias and
 *   the two LayerNorms), cast to the run dtype.  rpb_table is relative_position_bias_table as stored: ((2ws-1)^2, heads).
 *   wpk must hold sodt_wmsa_pack_bytes() bytes, 16-byte aligned.  Re-run whenever the parameters change.
 * sodt_wmsa_block_fwd: x, xm, xn2 [B*H*W][C] run dtype, natural token order.  Training passes the save-for-backward
 *   outputs (all or none): xn1 = LN1(x) [M][C]; st1 / st2 = (mean, rstd) of LN1 / LN2, f32 [M][2]; ao = attention
 *   output before the projection [M][C]; qkvw = q|k|v in window-major order [window][head][3][64][16] run dtype and
 *   lsew = log-sum-exp [window][head][64] f32, window = (b*(H/8) + wy)*(W/8) + wx in the shifted frame, token =
 *   window-local row-major - the operands of sodt_window_attn_bwd_wm.  Pass xn1 == NULL for inference.
 *   bf16: qkvw may be NULL (and the four-waves-per-window kernel never writes it): its backward, sodt_wmsa_block_bwd,
 *   recomputes q / k / v from xn1 and the pack; a non-NULL qkvw with bf16 returns SODT_EINVAL.  f32 (the parity path)
 *   requires qkvw. */
long sodt_wmsa_pack_bytes(int C, int heads, int ws, int dtype);
int sodt_wmsa_pack(const float* qkv_w, const float* qkv_b, const float* proj_w, const float* proj_b,
                   const float* rpb_table, const float* n1_w, const float* n1_b, const float* n2_w,
                   const float* n2_b, void* wpk, int C, int heads, int ws, int dtype, sodt_stream_t st);
int sodt_wmsa_block_fwd(const void* x, const void* wpk, void* xm, void* xn2, float* st1, float* st2,
                        void* xn1, void* qkvw, float* lsew, void* ao,
                        int B, int H, int W, int C, int heads, int ws, int shift, int dtype, sodt_stream_t st);
/* diagnostic hook: per-phase shader-cycle sums of the instrumented bf16 build (see csrc/wmsa_block.hip) */
int sodt_debug_wmsa_stamps(long long* out_host_256x8, int enable);
/* the same for the four-waves-per-window bf16 kernel (csrc/wmsa_hg.hip): 12 phases per workgroup */
int sodt_debug_wmsa_hg_stamps(long long* out_host_512x12, int enable);
/* sodt_window_attn_bwd on the window-major qkvw / lsew of sodt_wmsa_block_fwd (8x8 windows, head_dim 16);
 * dout and dqkv keep the natural layouts [M][C] / [M][3C]. */
int sodt_window_attn_bwd_wm(const void* qkvw, const float* bias_t, const void* dout, const float* lsew,
                            void* dqkv, float* dbias_t, int B, int H, int W, int C, int heads, int ws,
                            int shift, int dtype, sodt_stream_t st);
/* First stage of the fused block's backward, bf16 (backbone_vit.py:968 + the autograd of :971-989): d(attention output) dout
 * [M][C] -> dqkv [M][3C] and the relative-position-bias gradient dbias_t [heads][(2ws-1)^2] (accumulated), with q / k / v
 * RECOMPUTED per (window, head) from the block's saved LayerNorm-1 output xn1 [M][C] and its parameter pack wpk
 * (sodt_wmsa_pack) - the same MFMA product the forward ran - instead of read back from HBM; lsew as written by
 * sodt_wmsa_block_fwd.  C == 192, heads == 12, ws == 8, dtype == SODT_BF16 only; anything else returns SODT_EINVAL.  The
 * dgrad / wgrad GEMMs and the LayerNorm backward of the block stay separate launches (sodt_gemm_nt / _tn, sodt_layernorm_bwd). */
int sodt_wmsa_block_bwd(const void* xn1, const void* wpk, const float* bias_t, const void* dout, const float* lsew,
                        void* dqkv, float* dbias_t, int B, int H, int W, int C, int heads, int ws, int shift,
                        int dtype, sodt_stream_t st);

/* ---- fused linear MLP of a Swin block (csrc/mlp.hip) -----------------------------------------------------------
 * out = resid + fc2(GELU(fc1(xn))): Mlp.forward's linear branch backbone_vit.py:884-890 (fc1, exact-erf GELU, fc2; the
 * dropouts are identities at p = 0) together with the block's residual add `x + drop_path(mlp(norm2(x)))` (:1128), ONE launch.
 * xn, resid, out [M][C] run dtype (resid nullable: then out = fc2(GELU(fc1(xn)))); w1 = fc1.weight [4C][C], w2 = fc2.weight
 * [C][4C] in the run dtype (the prepared [N][K] copies sodt_gemm_nt takes), b1 [4C] / b2 [C] f32.
 * hact (nullable): GELU(fc1(xn) + b1) [M][4C] run dtype is also written - the operand of fc2's weight gradient; without it the
 * 4C-wide hidden activation never leaves the CU (inference form).
 * bf16, C == 192: the fused kernel (token tile in LDS, W1 / W2 streamed in 32-column slabs by LDS-DMA, GELU(h) chained from the
 * first product's accumulators into the second product's operand).  f32 (the parity path) and every other width run the
 * two-launch chain sodt_gemm_nt(GELU) -> sodt_gemm_nt(bias + residual) through hact, which is then required. */
int sodt_mlp_fwd(const void* xn, const void* w1, const float* b1, const void* w2, const float* b2, const void* resid,
                 void* out, void* hact, long M, int C, int dtype, sodt_stream_t st);

/* ---- 2x2-conv MLP of the shifted Swin blocks with fc1 folded into the convolution (csrc/convmlp.hip), bf16 -------------------
 * Mlp.forward's convolutional branch, backbone_vit.py:892-905: fc1 (C -> C), F.pad right / bottom, conv1 (2x2), GELU, fc2.  conv1 o fc1 is
 * one linear map per tap, so the 2x2 convolution runs directly on the LayerNorm output with composed weights and bias; the tokens of
 * the last column / row (where the padded fc1 output is zero including its bias) get a correction.  Exact algebra, bf16 rounding
 * differs (fc1's output is never rounded).  f32 masters in their torch layouts: fc1_w [C][C], conv_w [C][C][2][2] (tap = kh * 2 + kw).
 * sodt_convmlp_compose: weff bf16 [C][4C] (column tap * C + ci: the W operand of sodt_gemm_nt over the four tap segments), weffT bf16
 *   [C][4C] (row ci, column tap * C + co: the W operand of the input-gradient GEMM over the negated taps), beff f32 [C] = conv bias +
 *   sum_taps Wc_tap b1, vtap f32 [4][C] = Wc_tap b1.  Run once per forward (the masters change every step).
 * sodt_convmlp_border_fix: cp (pre-activation) and ca (GELU of it) [B*H*W][C] as written by the GEMM with beff: on the tokens with
 *   x == W - 1 / y == H - 1 the taps outside the image lose their vtap again and ca is recomputed.
 * sodt_convmlp_border_sums: bs f32 [3][C] (zeroed by the caller) += sums of dc [B*H*W][C] over the last-column tokens, the last-row
 *   tokens and the corner tokens.
 * sodt_convmlp_decompose: parameter gradients by the chain rule from dweff f32 [C][4C] (= dc^T x(taps), sodt_gemm_tn without kperm),
 *   colsum f32 [C] (its dbias) and bs: g_conv_w [C][C][2][2], g_conv_b, g_fc1_w, g_fc1_b are ADDED to.  C <= 384.
 * Reproducibility: sodt_convmlp_border_sums and the fc1.weight part of sodt_convmlp_decompose sum their per-workgroup partials with
 *   f32 atomicAdd (four k-quarters per tile / one partial per 8-token chunk), so g_fc1_w, g_fc1_b and g_conv_w of a folded MLP agree
 *   run to run only to f32 summation order (~1e-7 relative) - unlike the weight gradients of sodt_gemm_tn and the direct 3x3 kernels,
 *   which reduce their partial tiles in a fixed order and are bitwise reproducible. */
int sodt_convmlp_compose(const float* fc1_w, const float* fc1_b, const float* conv_w, const float* conv_b, void* weff, void* weffT,
                         float* beff, float* vtap, int C, int dtype, sodt_stream_t st);
int sodt_convmlp_border_fix(void* cp, void* ca, const float* vtap, int B, int H, int W, int C, int dtype, sodt_stream_t st);
int sodt_convmlp_border_sums(const void* dc, float* bs, int B, int H, int W, int C, int dtype, sodt_stream_t st);
int sodt_convmlp_decompose(const float* dweff, const float* colsum, const float* bs, const float* fc1_w, const float* fc1_b,
                           const float* conv_w, float* g_conv_w, float* g_conv_b, float* g_fc1_w, float* g_fc1_b, int C, sodt_stream_t st);

/* Front end (backbone_vit.py:195-210): channel split, 4x Conv2d(1->48,k4,s4) (R with
 * padding 1), pairwise cross-channel attention (R<-G, G<-B, B<-IR, IR<-G; 12 heads x 4,
 * scale 1/2, window ca_ws, no projections) + residual + LayerNorm(48), concatenated to
 * [B*t*t][192].  rgb (B,3,S,S) f32 / ir plane pointer with its batch stride, f32 in
 * [0,1].  params f32: w[4][48][16], b[4][48], gamma[4][48], beta[4][48].  ca_ws == 1
 * is the shipped configuration (backbone_vit.py:438).
 * The backward accumulates into dw/db/dgamma/dbeta.  With a workspace of sodt_frontend_bwd_workspace_bytes() (f32, need
 * not be zeroed) the per-workgroup partial sums are reduced by a second small kernel; ws == NULL falls back to direct
 * atomics (same result up to summation order, ~0.15 ms slower at 8 x 1024^2). */
int sodt_frontend_fwd(const float* rgb, const float* ir, long ir_bstride, const float* w, const float* b,
                      const float* gamma, const float* beta, void* out, int B, int S, int ca_ws,
                      int dtype, sodt_stream_t st);
long sodt_frontend_bwd_workspace_bytes(int B, int S);
int sodt_frontend_bwd(const float* rgb, const float* ir, long ir_bstride, const float* w, const float* b,
                      const float* gamma, const float* beta, const void* dout,
                      float* dw, float* db, float* dgamma, float* dbeta, int B, int S, int ca_ws,
                      float* ws, long ws_bytes, int dtype, sodt_stream_t st);

/* General cross-channel attention (window ca_ws in 1..8, optional cyclic shift; backbone_vit.py:469-561, :589-616) as
 * separate stages: e = 4x Conv2d(1->48,k4,s4) embeddings, f32 [B*t*t][192] (caller's workspace); then per pair
 * softmax((q k^T + mask)/2) v + residual + LayerNorm(48).  The backward accumulates d e with atomics (zero it first);
 * dw/db/dgamma/dbeta accumulate.  ca_ws == 1 callers should use the fused sodt_frontend_* instead. */
int sodt_patch_embed4_fwd(const float* rgb, const float* ir, long ir_bstride, const float* w, const float* b,
                          float* e, int B, int S, sodt_stream_t st);
int sodt_patch_embed4_bwd(const float* rgb, const float* ir, long ir_bstride, const float* de, float* dw, float* db,
                          int B, int S, sodt_stream_t st);
int sodt_cross_attn_ln_fwd(const float* e, const float* gamma, const float* beta, void* out, int B, int S, int ws,
                           int shift, int dtype, sodt_stream_t st);
int sodt_cross_attn_ln_bwd(const float* e, const float* gamma, const void* dout, float* de, float* dgamma,
                           float* dbeta, int B, int S, int ws, int shift, int dtype, sodt_stream_t st);

/* BatchNorm2d (eps 1e-3, momentum 0.03) + SiLU of the head's Conv (common.py:38-50,
 * torch_utils.py:150-152), token-major.  stats f64 [SODT_STATS_REPL][2][C] from SODT_EPI_STATS.
 * finalize: mean/rstd (f32 [2][C]) + running-stat update (unbiased var); stats == NULL
 * takes mean/rstd from the running statistics (eval).
 * sodt_col_stats: the same statistics from a STORED convolution output z [M][ldz] (stats[r][0][c] += sum_m z, stats[r][1][c] += sum_m z^2,
 * f64, one replica r per workgroup) - for the convolutions that run on the direct 3x3 kernels (csrc/conv3.hip), which have no
 * statistics epilogue. */
int sodt_col_stats(const void* z, int ldz, double* stats, long M, int C, int dtype, sodt_stream_t st);
int sodt_bn_finalize(const double* stats, float* mean_rstd, float* running_mean, float* running_var,
                     long count, int C, float eps, float momentum, sodt_stream_t st);
/* scale = gamma*rstd, shift = beta - mean*scale: the fused / eval-mode form (torch_utils.py:182-203) */
int sodt_bn_affine(const float* mean_rstd, const float* gamma, const float* beta, float* scale, float* shift,
                   int C, sodt_stream_t st);
int sodt_bn_silu_fwd(const void* z, const float* mean_rstd, const float* gamma, const float* beta,
                     void* y, int ldy, long M, int C, int dtype, sodt_stream_t st);
/* pass 1: red[0][c] += sum g, red[1][c] += sum g*xhat with g = dy * silu'(a);  pass 2: dz */
int sodt_bn_silu_bwd_reduce(const void* dy, int lddy, const void* z, const float* mean_rstd,
                            const float* gamma, const float* beta, double* red, long M, int C,
                            int dtype, sodt_stream_t st);
int sodt_bn_silu_bwd_apply(const void* dy, int lddy, const void* z, const float* mean_rstd,
                           const float* gamma, const float* beta, const double* red, void* dz,
                           float* dgamma, float* dbeta, long M, int C, int dtype, sodt_stream_t st);

/* dst[(b,y,x)][coff : coff+C] = src[(b, y>>shr, x>>shr)][0:C]  (nn.Upsample nearest + Concat slot) */
int sodt_copy_rows(const void* src, int lds_, void* dst, int ldd, int B, int Ho, int Wo, int shr, int C,
                   int dtype, sodt_stream_t st);
/* dsrc[(b,y,x)][c] (+)= sum_{a,b<2^shr} d[(b, (y<<shr)+a, (x<<shr)+b)][c]  (their backward) */
int sodt_gather_sum_rows(const void* d, int ldd, void* dsrc, int lds_, int B, int Hs, int Ws, int shr, int C,
                         int accumulate, int dtype, sodt_stream_t st);

/* nn.MaxPool2d(5, stride 1, padding 2), token-major, the unit of SPP (common.py:129-140: its 5 / 9 / 13 pools are this pool
 * applied 1 / 2 / 3 times).  argmax (nullable in inference): one byte per (token, channel), window slot ky*5+kx of the
 * first maximum in scan order of THAT 5x5 window (PyTorch's rule for a 5x5 pool; for the cascaded 9 / 13 pools ties inside
 * the composed window may resolve to a different, equally valid, tied element than torch's single 9x9 / 13x13 scan).
 * bwd: dx (+)= gather of dy through argmax. */
int sodt_maxpool5_fwd(const void* x, int ldx, void* y, int ldy, unsigned char* argmax, int B, int H, int W, int C,
                      int dtype, sodt_stream_t st);
int sodt_maxpool5_bwd(const void* dy, int lddy, const unsigned char* argmax, void* dx, int lddx, int accumulate,
                      int B, int H, int W, int C, int dtype, sodt_stream_t st);

/* Detect: dpred f32 (B, na, HW, no) -> dz run dtype [B*HW][ldz] (cols >= na*no zeroed)  (model.py:55 backward) */
int sodt_detect_unpermute(const float* dpred, void* dz, int ldz, int B, int HW, int na, int no,
                          int dtype, sodt_stream_t st);
/* Detect eval decode (model.py:61-64): raw f32 (B,na,ny,nx,no) -> z f32 (B, na*ny*nx, no) */
int sodt_detect_decode(const float* raw, const float* anchor_grid, float* z, int B, int na, int ny, int nx,
                       int no, float stride, sodt_stream_t st);

/* ---- non_max_suppression (general.py:425-512), one image per call, eval only ---------------------------
 * z: decoded rows (N, 5+nc) f32 [cx cy w h obj cls...] of one image (Detect eval output, model.py:64).
 * sodt_nms_candidates: general.py:433,446-476 - rows with obj > conf; multi_label (forced off when nc == 1):
 *   one candidate per class with obj*cls > conf, else the best class; class_allow (nc bytes, device, nullable)
 *   is the `classes` filter (general.py:479-480).  Writes *count (device int, may exceed cap - then the call must
 *   be repeated with a larger cap) and one 64-bit key per candidate, (~score_bits << 32) | (row*nc + class).
 * sodt_nms_select: sorts the keys (descending score, ties in the reference's candidate order), keeps the top
 *   30000 (general.py:489-490), runs class-offset NMS at iou_thres (torchvision.ops.nms semantics,
 *   general.py:493-496), cuts at 300 (general.py:497-498), merge-NMS + redundancy filter when
 *   1 < n_total < 3000 (general.py:499-506).  out: (<=300, 6) f32 [x1 y1 x2 y2 conf cls]; out_index: the
 *   candidate id row*nc + class of each output row; *out_count: rows written.  ws: scratch of at least
 *   sodt_nms_workspace_bytes(n_total).  Nothing synchronises; the caller reads *count / *out_count. */
int sodt_nms_candidates(const float* z, int N, int nc, float conf_thres, int multi_label,
                        const unsigned char* class_allow, unsigned long long* keys, int cap, int* count,
                        sodt_stream_t st);
int sodt_nms_workspace_bytes(long n_total, size_t* bytes);
int sodt_nms_select(const float* z, int nc, const unsigned long long* keys, long n_total, float iou_thres,
                    int agnostic, void* ws, size_t ws_bytes, float* out, int* out_index, int* out_count,
                    sodt_stream_t st);

/* Batched parameter preparation: out = cast(permute3(in)) for a device-resident table. */
typedef struct {
  const float* src; void* dst;
  int d0, d1, d2;       /* source viewed as [d0][d1][d2] */
  int p0, p1, p2;       /* destination dims order: dst[i_p0][i_p1][i_p2] */
  int dst_ld;           /* elements per destination "row" (>= product of the trailing two dims) */
  int inner_ld;         /* elements between consecutive i_p1 (0: the extent of dims[p2], i.e. dense) - > extent when dims[p2] is zero-padded */
} sodt_prep_desc;
int sodt_prep_weights(const sodt_prep_desc* table_dev, int n, int max_elems, int dtype, sodt_stream_t st);

/* Fused optimizer step over the flat f32 parameter / gradient / momentum / EMA buffers (csrc/optim.hip): what the reference
 * does with torch.optim.SGD(momentum, nesterov) over the weight-decay groups of basics/optimizer.py:35-49 (Train.py:145-150,
 * :448-450), ModelEMA.update (basics/utils/torch_utils.py:291-301) and the cast of the masters to the run dtype, in one
 * streaming launch.  All buffers hold n_elems f32 (16-byte aligned, n_elems % 4 == 0); every 4-element chunk belongs to
 * one parameter and group_of_chunk[chunk] (device bytes, nullable = all group 0) picks its hyper-parameters
 * lr / momentum / weight_decay[group] (host arrays of ngroups <= 4); 255 = not trained (cast / EMA only).
 *   d = g*grad_scale + wd*p;  m = momentum*m + d;  u = nesterov ? d + momentum*m : m;  p -= lr*u
 *   ema = ema*ema_decay + (1-ema_decay)*p (ema nullable);  p_cast = (cast_dtype) p (p_cast nullable). */
int sodt_sgd_ema_step(float* p, const float* g, float* mom, float* ema, void* p_cast, int cast_dtype,
                      const unsigned char* group_of_chunk, long n_elems, int ngroups, const float* lr,
                      const float* momentum, const float* weight_decay, int nesterov, float grad_scale,
                      float ema_decay, sodt_stream_t st);

/* Input pre-processing of the training / evaluation loop (csrc/preprocess.hip): `imgs.to(device).float() / 255.0` followed by
 * `F.interpolate(image, size=[i // down_factor ...], mode='bilinear', align_corners=True)` (Train.py:364-374; test.py:124-129
 * is the factor-1 case) for the RGB and the IR batch in one launch.  rgb / ir: uint8 (B, c, Hin, Win) contiguous (device);
 * out_rgb / out_ir: f32 (B, c, Hout, Wout), the planes sodt_frontend_fwd reads.  Hout <= Hin, Wout <= Win; Hout == Hin is
 * the plain u8 -> f32 / 255.  c_ir may be 0 (ir, out_ir then unused). */
int sodt_preprocess_u8(const unsigned char* rgb, const unsigned char* ir, float* out_rgb, float* out_ir, int B, int c_rgb,
                       int c_ir, int Hin, int Win, int Hout, int Wout, sodt_stream_t st);

/* ComputeLoss.__call__ + build_targets (basics/utils/loss.py:116-224) with bbox_iou(CIoU) (basics/utils/general.py:347-389)
 * for the single detection layer of models/model.yaml: loss values and d(loss * batch) / d pred in one call.
 * pred f32 (B, na, ny, nx, 5+nc) contiguous; targets f32 (nt, 6) = (image, class, x, y, w, h) normalised (device);
 * anchors f32 (na, 2) in grid units (Detect.anchors[0]); hyper-parameters as in models/hyp.scratch.yaml (box, cls,
 * cls_pw, obj, obj_pw, anchor_t) and model.gr.  dpred f32 like pred; out4 = (loss * B, lbox, lobj, lcls) as loss.py:163.
 * Duplicate cells take the objectness target of the LAST matching candidate in the reference's order (CPU index_put).
 * ws: scratch of sodt_yolo_loss_workspace_bytes(B*na*ny*nx, nt, nc) bytes.  nc <= 32, na <= 8. */
int sodt_yolo_loss_workspace_bytes(long ncells, int nt, int nc, size_t* bytes);
int sodt_yolo_loss(const float* pred, const float* targets, int nt, const float* anchors, int B, int na, int ny, int nx,
                   int nc, float h_box, float h_cls, float cls_pw, float h_obj, float obj_pw, float anchor_t, float gr,
                   void* ws, size_t ws_bytes, float* dpred, float* out4, sodt_stream_t st);

/* hipMemsetAsync(p, 0, bytes) on the stream (statistics / gradient accumulators) */
int sodt_memset_zero(void* p, long bytes, sodt_stream_t st);

/* bias table (L, heads) f32 -> (heads, L) f32, and the reverse accumulate for its gradient
 * (accumulate: 0 = store, 1 = add, 2 = add and clear the source) */
int sodt_transpose_f32(const float* src, float* dst, int rows, int cols, int accumulate, sodt_stream_t st);

/* out[r][c] (f32) += sum_b d[b][r][c]: gradient of the batch-broadcast pos_embed add (backbone_vit.py:215-217) */
int sodt_batch_sum(const void* d, float* out, int B, long RC, int dtype, sodt_stream_t st);

/* f32 <-> run dtype elementwise cast */
int sodt_cast(const void* src, void* dst, long n, int src_dtype, int dst_dtype, sodt_stream_t st);

const char* sodt_version(void);

#ifdef __cplusplus
}
#endif
#endif
