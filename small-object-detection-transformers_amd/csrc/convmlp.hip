// The 2x2-conv MLP of the shifted Swin blocks (backbone_vit.py:892-905) with fc1 FOLDED INTO the convolution, bf16 path.
//
//   reference:   u = fc1(x) = x W1^T + b1          [M][C] -> [M][C]
//                c = conv1(pad(u, right / bottom))   c[t] = bc + sum_taps Wc_tap u[t + tap]     (a tap outside the image reads 0)
//                out = fc2(GELU(c))
//
// conv1 o fc1 is one linear map per tap:   c[t] = bc + sum_{valid taps} ( Wc_tap W1 ) x[t + tap] + Wc_tap b1
// so the forward runs the 2x2 convolution directly on x with the composed weights Weff_tap = Wc_tap W1 (192 x 192 per tap, 28 MFLOP per
// block and step) and the bias beff = bc + sum_taps Wc_tap b1; only tokens on the right column / bottom row of an image - where the
// padded u is zero INCLUDING its bias - need a correction (the missing taps' Wc_tap b1 comes off again, 0.8 % of the tokens at 256^2).
// What it removes per block and step (stage 1 of B=8 @1024^2): the fc1 GEMM and its [M][C] output (0.11 ms, 2 x 201 MB), and in the
// backward the d(x) = du W1 GEMM (0.08 ms) and the dW1 = du^T x weight-gradient GEMM (0.13 ms): the input gradient is ONE 2x2-tap GEMM
// with Weff^T, and the parameter gradients follow from d(Weff) = dc^T x(taps) by the chain rule on 192 x 192 matrices:
//     dWc_tap = dWeff_tap W1^T + dv_tap b1^T        dW1 = sum_taps Wc_tap^T dWeff_tap        db1 = sum_taps Wc_tap^T dv_tap
//     dv_tap  = sum over the tokens where the tap is inside the image of dc = colsum(dc) - (border sums)
// Exact algebra; the bf16 rounding differs from the reference's (u is never rounded to bf16, Weff is).  The f32 parity path keeps
// the three-GEMM form.
//
// Layouts: fc1.weight W1 [C][C] (row m, column ci), conv1.weight Wc [C][C][2][2] (co, m, kh, kw; tap = kh * 2 + kw = (dy, dx)), both the
// f32 masters.  weff bf16 [C][4 C]: column tap * C + ci (the K order of the tap segments); weffT bf16 [C][4 C]: row ci, column
// tap * C + co (the W operand of the input-gradient GEMM over the negated taps).
#include "common.h"
#include "../../include/sodt_hip.h"

namespace {

// ---- 32 x 32 output tile of  acc[i][j] = sum_k A(i, k) B(k, j)  by one 256-thread workgroup (f32, K a multiple of 32): 32-deep k chunks
//      through LDS, thread (ty, tx) = (tid / 16, tid % 16) owns outputs (2 ty .. + 1, 2 tx .. + 1).  la(r, k) / lb(k, c): global loads of one
//      element of the A / B tile (r, c in 0..31 relative to the tile, k absolute); the callers choose which index runs along the lanes.
template <typename LA, typename LB>
__device__ __forceinline__ void tile32(float (&acc)[2][2], int K, float (*As)[33], float (*Bs)[33], LA la, LB lb, bool a_k_fast, bool b_k_fast) {
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  for (int k0 = 0; k0 < K; k0 += 32) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + 256 * e, hi = idx >> 5, lo = idx & 31;
      if (a_k_fast) As[hi][lo] = la(hi, k0 + lo); else As[lo][hi] = la(lo, k0 + hi);      // As[row][k]
      if (b_k_fast) Bs[lo][hi] = lb(k0 + lo, hi); else Bs[hi][lo] = lb(k0 + hi, lo);      // Bs[k][col]
    }
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
      const float a0 = As[2 * ty][k], a1 = As[2 * ty + 1][k], b0 = Bs[k][2 * tx], b1 = Bs[k][2 * tx + 1];
      acc[0][0] = fmaf(a0, b0, acc[0][0]); acc[0][1] = fmaf(a0, b1, acc[0][1]);
      acc[1][0] = fmaf(a1, b0, acc[1][0]); acc[1][1] = fmaf(a1, b1, acc[1][1]);
    }
    __syncthreads();
  }
}

// weff[co][t C + ci] = sum_m Wc[co][m][t] W1[m][ci] (and its transposed copy): workgroup (ci tile, co tile, tap); the last grid row
// (blockIdx.y == C / 32) computes vtap / beff instead
__global__ __launch_bounds__(256) void convmlp_compose_kernel(const float* __restrict__ W1, const float* __restrict__ b1,
                                                              const float* __restrict__ Wc, const float* __restrict__ bc,
                                                              bf16* __restrict__ weff, bf16* __restrict__ weffT,
                                                              float* __restrict__ beff, float* __restrict__ vtap, int C) {
  __shared__ float As[32][33], Bs[32][33];
  const int tid = threadIdx.x, t = blockIdx.z;
  if ((int)blockIdx.y == C / 32) {               // vtap[t][co] = Wc_tap b1 for 32 output channels; beff by the tap-0 workgroups
    const int co = blockIdx.x * 32 + (tid >> 3), part = tid & 7;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int m = part; m < C; m += 8) {
      const float4 w = *(const float4*)(Wc + ((long)co * C + m) * 4);
      const float b = b1[m];
      a[0] = fmaf(w.x, b, a[0]); a[1] = fmaf(w.y, b, a[1]); a[2] = fmaf(w.z, b, a[2]); a[3] = fmaf(w.w, b, a[3]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] += __shfl_xor(a[j], 1); a[j] += __shfl_xor(a[j], 2); a[j] += __shfl_xor(a[j], 4); }
    if (part == 0 && t == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) vtap[j * C + co] = a[j];
      beff[co] = bc[co] + a[0] + a[1] + a[2] + a[3];
    }
    return;
  }
  const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
  float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  tile32(acc, C, As, Bs,
         [&](int r, int k) { return Wc[((long)(co0 + r) * C + k) * 4 + t]; },
         [&](int k, int c) { return W1[(long)k * C + ci0 + c]; }, true, false);
  const int ty = tid >> 4, tx = tid & 15;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      weff[(long)(co0 + 2 * ty + i) * 4 * C + t * C + ci0 + 2 * tx + j] = (bf16)acc[i][j];
      As[2 * tx + j][2 * ty + i] = acc[i][j];      // transposed through LDS: [ci][co]
    }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int idx = tid + 256 * e, r = idx >> 5, c = idx & 31;
    weffT[(long)(ci0 + r) * 4 * C + t * C + co0 + c] = (bf16)As[r][c];
  }
}

// tokens on the right column (taps dx = 1 fall outside) and the bottom row (taps dy = 1) of every image: the pre-activation loses the
// missing taps' Wc_tap b1 and the activation is recomputed.  One thread per (border token, 8-channel chunk).
__global__ __launch_bounds__(256) void convmlp_border_fix_kernel(bf16* __restrict__ cp, bf16* __restrict__ ca, const float* __restrict__ vtap,
                                                                 int B, int H, int W, int C) {
  const int nb = H + W - 1, cpr = C / 8;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * nb * cpr) return;
  const int c8 = (int)(i % cpr) * 8;
  const int k = (int)((i / cpr) % nb), b = (int)(i / ((long)cpr * nb));
  const int y = k < W ? H - 1 : k - W, x = k < W ? k : W - 1;         // the bottom row, then the right column above it
  const bool bot = y == H - 1, rgt = x == W - 1;
  const long row = ((long)b * H + y) * W + x;
  float v[8];
  unpack<bf16>(*(const uint4*)(cp + row * C + c8), v);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float s = 0.f;
    if (rgt) s += vtap[1 * C + c8 + j];
    if (bot) s += vtap[2 * C + c8 + j];
    if (rgt || bot) s += vtap[3 * C + c8 + j];
    v[j] -= s;
  }
  *(uint4*)(cp + row * C + c8) = pack<bf16>(v);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = gelu_t<bf16>(v[j]);
  *(uint4*)(ca + row * C + c8) = pack<bf16>(v);
}

// bs[0][c] = sum of dc over the right-column tokens, bs[1][c] over the bottom-row tokens, bs[2][c] over the corner tokens (f32, zeroed
// by the caller).  One workgroup per (8-token chunk of a line, image, line kind): 8-channel chunks across the lanes, tokens down the
// waves, one LDS reduction, then one atomic per channel and workgroup.
__global__ __launch_bounds__(256) void convmlp_border_sums_kernel(const bf16* __restrict__ dc, float* __restrict__ bs, int B, int H, int W, int C) {
  __shared__ float red[8][392];                              // C <= 384 (+ 8: rows on different banks)
  const int b = blockIdx.y, kind = blockIdx.z;               // 0: right column (x = W - 1), 1: bottom row (y = H - 1)
  const int len = kind == 0 ? H : W;
  const int t = blockIdx.x * 8 + (threadIdx.x >> 5);         // 8 tokens per workgroup, 32 lanes (x 8 channels, twice for C = 384) each
  const int cpr = C / 8;
  const bool ok = t < len;
  const int y = kind == 0 ? t : H - 1, x = kind == 0 ? W - 1 : t;
  for (int ch = threadIdx.x & 31; ch < cpr; ch += 32) {
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (ok) unpack<bf16>(*(const uint4*)(dc + (((long)b * H + y) * W + x) * C + ch * 8), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.x >> 5][ch * 8 + j] = v[j];
    if (ok && kind == 0 && t == H - 1)
#pragma unroll
      for (int j = 0; j < 8; ++j) atomicAdd(bs + 2 * C + ch * 8 + j, v[j]);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) a += red[r][c];
    atomicAdd(bs + kind * C + c, a);
  }
}

// parameter gradients from d(Weff) [C][4 C] (f32), the column sums of dc and the border sums (all of THIS backward call); the results
// are ADDED to the gradient buffers (torch layouts).  One launch; workgroup ranges: [0, 4 n) conv1.weight tiles (m tile, co tile, tap),
// [4 n, 8 n) fc1.weight tiles (ci tile, m tile, quarter of the (co, tap) contraction: f32 atomics), then one workgroup for fc1.bias and
// conv1.bias; n = (C / 32)^2
__global__ __launch_bounds__(256) void convmlp_decompose_kernel(const float* __restrict__ dweff, const float* __restrict__ colsum,
                                                                const float* __restrict__ bs, const float* __restrict__ W1,
                                                                const float* __restrict__ b1, const float* __restrict__ Wc,
                                                                float* __restrict__ gWc, float* __restrict__ gbc, float* __restrict__ gW1,
                                                                float* __restrict__ gb1, int C) {
  __shared__ float As[32][33], Bs[32][33];
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const int nt = C / 32, n = nt * nt;
  // dv[t][co] = colsum[co] - (sum of dc over the tokens whose tap t is outside): t = 1 right column, t = 2 bottom row, t = 3 either
  auto dv = [&](int t, int co) {
    const float r = bs[co], bt = bs[C + co], cn = bs[2 * C + co];
    return colsum[co] - (t == 0 ? 0.f : t == 1 ? r : t == 2 ? bt : r + bt - cn);
  };
  int bid = blockIdx.x;
  if (bid < 4 * n) {                               // gWc[co][m][t] += sum_ci dweff[co][t C + ci] W1[m][ci] + dv[t][co] b1[m]
    const int t = bid & 3, tl = bid >> 2;
    const int m0 = (tl % nt) * 32, co0 = (tl / nt) * 32;
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    tile32(acc, C, As, Bs,
           [&](int r, int k) { return dweff[(long)(co0 + r) * 4 * C + t * C + k]; },
           [&](int k, int c) { return W1[(long)(m0 + c) * C + k]; }, true, true);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        gWc[((long)(co0 + 2 * ty + i) * C + m0 + 2 * tx + j) * 4 + t] += acc[i][j] + dv(t, co0 + 2 * ty + i) * b1[m0 + 2 * tx + j];
    return;
  }
  bid -= 4 * n;
  if (bid < 4 * n) {                               // gW1[m][ci] += sum_(co, t) Wc[co][m][t] dweff[co][t C + ci]: k = 4 co + t, a quarter of the co's
    const int q = bid & 3, tl = bid >> 2;
    const int ci0 = (tl % nt) * 32, m0 = (tl / nt) * 32;
    const int kq = q * C;                          // k range [q C, (q + 1) C)
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    tile32(acc, C, As, Bs,
           [&](int r, int k) { const int kk = kq + k; return Wc[((long)(kk >> 2) * C + m0 + r) * 4 + (kk & 3)]; },
           [&](int k, int c) { const int kk = kq + k; return dweff[(long)(kk >> 2) * 4 * C + (kk & 3) * C + ci0 + c]; }, true, false);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) atomicAdd(gW1 + (long)(m0 + 2 * ty + i) * C + ci0 + 2 * tx + j, acc[i][j]);
    return;
  }
  for (int m = tid; m < C; m += 256) {             // gb1[m] += sum_t sum_co Wc[co][m][t] dv[t][co];  gbc += colsum
    float acc = 0.f;
    for (int co = 0; co < C; ++co) {
      const float4 wc = *(const float4*)(Wc + ((long)co * C + m) * 4);
      acc += wc.x * dv(0, co) + wc.y * dv(1, co) + wc.z * dv(2, co) + wc.w * dv(3, co);
    }
    gb1[m] += acc;
    gbc[m] += colsum[m];                           // conv1.bias: beff carries bc with coefficient 1
  }
}

inline int nblk(long n) { return (int)((n + 255) / 256); }

}  // namespace

extern "C" int sodt_convmlp_compose(const float* fc1_w, const float* fc1_b, const float* conv_w, const float* conv_b, void* weff,
                                    void* weffT, float* beff, float* vtap, int C, int dtype, sodt_stream_t st) {
  if (!fc1_w || !fc1_b || !conv_w || !conv_b || !weff || !weffT || !beff || !vtap || C <= 0 || (C % 32) || dtype != SODT_BF16 ||
      (((uintptr_t)conv_w) & 15))
    return SODT_EINVAL;
  hipLaunchKernelGGL(convmlp_compose_kernel, dim3(C / 32, C / 32 + 1, 4), dim3(256), 0, (hipStream_t)st, fc1_w, fc1_b, conv_w, conv_b,
                     (bf16*)weff, (bf16*)weffT, beff, vtap, C);
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

extern "C" int sodt_convmlp_border_fix(void* cp, void* ca, const float* vtap, int B, int H, int W, int C, int dtype, sodt_stream_t st) {
  if (!cp || !ca || !vtap || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 8) || dtype != SODT_BF16 ||
      ((((uintptr_t)cp) | ((uintptr_t)ca)) & 15))
    return SODT_EINVAL;
  hipLaunchKernelGGL(convmlp_border_fix_kernel, dim3(nblk((long)B * (H + W - 1) * (C / 8))), dim3(256), 0, (hipStream_t)st, (bf16*)cp,
                     (bf16*)ca, vtap, B, H, W, C);
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

extern "C" int sodt_convmlp_border_sums(const void* dc, float* bs, int B, int H, int W, int C, int dtype, sodt_stream_t st) {
  if (!dc || !bs || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 8) || C > 384 || dtype != SODT_BF16 || B > 65535 || (((uintptr_t)dc) & 15))
    return SODT_EINVAL;
  const int len = H > W ? H : W;
  hipLaunchKernelGGL(convmlp_border_sums_kernel, dim3((len + 7) / 8, B, 2), dim3(256), 0, (hipStream_t)st, (const bf16*)dc, bs, B, H, W, C);
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

extern "C" int sodt_convmlp_decompose(const float* dweff, const float* colsum, const float* bs, const float* fc1_w, const float* fc1_b,
                                      const float* conv_w, float* g_conv_w, float* g_conv_b, float* g_fc1_w, float* g_fc1_b, int C,
                                      sodt_stream_t st) {
  if (!dweff || !colsum || !bs || !fc1_w || !fc1_b || !conv_w || !g_conv_w || !g_conv_b || !g_fc1_w || !g_fc1_b || C <= 0 || (C % 32) ||
      ((((uintptr_t)conv_w) | ((uintptr_t)g_conv_w)) & 15))
    return SODT_EINVAL;
  const int n = (C / 32) * (C / 32);
  hipLaunchKernelGGL(convmlp_decompose_kernel, dim3(8 * n + 1), dim3(256), 0, (hipStream_t)st, dweff, colsum, bs, fc1_w, fc1_b, conv_w,
                     g_conv_w, g_conv_b, g_fc1_w, g_fc1_b, C);
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}
