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

// ---- 64 x 64 output tile of  acc[i][j] = sum_k A(i, k) B(k, j)  by one 256-thread workgroup (f32, K a multiple of 16): 16-deep k chunks
//      through LDS, thread (ty, tx) = (tid / 16, tid % 16) owns the 4 x 4 outputs (ty + 16 i, tx + 16 j).  la(r, k) / lb(k, c): global loads
//      of one element of the A / B tile (r, c in 0..63 relative to the tile, k absolute); *_k_fast says which index runs along the lanes.
constexpr int CT = 64, CK = 16;
constexpr int CONVMLP_MAXC = 384;                 // widest folded MLP (stage 2); the decompose kernel's bias tail holds 4 C floats in LDS
template <typename LA, typename LB>
__device__ __forceinline__ void tile64(float (&acc)[4][4], int K, float (*As)[CT + 1], float (*Bs)[CT + 1], LA la, LB lb, bool a_k_fast, bool b_k_fast) {
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  // the next chunk's eight global loads are in flight while the current one is multiplied (one workgroup per CU: nothing else hides
  // their latency - the first version, without the prefetch, took 225 us for the C = 384 decomposition)
  float pa[4], pb[4];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + 256 * e;                         // 1024 elements per operand tile
      pa[e] = a_k_fast ? la(idx >> 4, k0 + (idx & 15)) : la(idx & 63, k0 + (idx >> 6));
      pb[e] = b_k_fast ? lb(k0 + (idx & 15), idx >> 4) : lb(k0 + (idx >> 6), idx & 63);
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += CK) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + 256 * e;
      if (a_k_fast) As[idx & 15][idx >> 4] = pa[e]; else As[idx >> 6][idx & 63] = pa[e];      // As[k][row]
      if (b_k_fast) Bs[idx & 15][idx >> 4] = pb[e]; else Bs[idx >> 6][idx & 63] = pb[e];      // Bs[k][col]
    }
    __syncthreads();
    if (k0 + CK < K) fetch(k0 + CK);
#pragma unroll
    for (int k = 0; k < CK; ++k) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = As[k][ty + 16 * i]; b[i] = Bs[k][tx + 16 * i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
}

// weff[co][t C + ci] = sum_m Wc[co][m][t] W1[m][ci] (and its transposed copy): workgroup (ci tile, co tile, tap); the last grid row
// (blockIdx.y == C / 64) computes vtap / beff instead
__global__ __launch_bounds__(256) void convmlp_compose_kernel(const float* __restrict__ W1, const float* __restrict__ b1,
                                                              const float* __restrict__ Wc, const float* __restrict__ bc,
                                                              bf16* __restrict__ weff, bf16* __restrict__ weffT,
                                                              float* __restrict__ beff, float* __restrict__ vtap, int C) {
  __shared__ float As[CT][CT + 1], Bs[CK][CT + 1];          // (As doubles as the 64 x 64 transposition buffer of the output tile)
  const int tid = threadIdx.x, t = blockIdx.z;
  if ((int)blockIdx.y == C / CT) {               // vtap[t][co] = Wc_tap b1 for 32 output channels per workgroup; beff with them
    if (t != 0 || (int)blockIdx.x * 32 >= C) return;
    const int co = blockIdx.x * 32 + (tid >> 3), part = tid & 7;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int m = part; m < C; m += 8) {
      const float4 w = *(const float4*)(Wc + ((long)co * C + m) * 4);
      const float b = b1[m];
      a[0] = fmaf(w.x, b, a[0]); a[1] = fmaf(w.y, b, a[1]); a[2] = fmaf(w.z, b, a[2]); a[3] = fmaf(w.w, b, a[3]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] += __shfl_xor(a[j], 1); a[j] += __shfl_xor(a[j], 2); a[j] += __shfl_xor(a[j], 4); }
    if (part == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) vtap[j * C + co] = a[j];
      beff[co] = bc[co] + a[0] + a[1] + a[2] + a[3];
    }
    return;
  }
  if ((int)blockIdx.x >= C / CT) return;        // (the grid is C / 32 wide for the vtap row)
  const int ci0 = blockIdx.x * CT, co0 = blockIdx.y * CT;
  float acc[4][4] = {};
  tile64(acc, C, (float (*)[CT + 1])As, Bs,
         [&](int r, int k) { return Wc[((long)(co0 + r) * C + k) * 4 + t]; },
         [&](int k, int c) { return W1[(long)k * C + ci0 + c]; }, true, false);
  const int ty = tid >> 4, tx = tid & 15;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      weff[(long)(co0 + ty + 16 * i) * 4 * C + t * C + ci0 + tx + 16 * j] = (bf16)acc[i][j];
      As[tx + 16 * j][ty + 16 * i] = acc[i][j];      // transposed through LDS: [ci][co]
    }
  __syncthreads();
  for (int idx = tid; idx < CT * CT; idx += 256) {
    const int r = idx >> 6, c = idx & 63;
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
// [4 n, 8 n) fc1.weight tiles (ci tile, m tile, quarter of the (co, tap) contraction: f32 atomics), then C / 32 workgroups for fc1.bias
// and conv1.bias; n = (C / 64)^2
__global__ __launch_bounds__(256) void convmlp_decompose_kernel(const float* __restrict__ dweff, const float* __restrict__ colsum,
                                                                const float* __restrict__ bs, const float* __restrict__ W1,
                                                                const float* __restrict__ b1, const float* __restrict__ Wc,
                                                                float* __restrict__ gWc, float* __restrict__ gbc, float* __restrict__ gW1,
                                                                float* __restrict__ gb1, int C) {
  __shared__ float As[CK][CT + 1], Bs[CK][CT + 1];
  __shared__ float dvs[4 * CONVMLP_MAXC];          // dv[t][co] of the bias tail (its own array: 4 C floats do not fit As at C = 384)
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const int nt = C / CT, n = nt * nt;
  // dv[t][co] = colsum[co] - (sum of dc over the tokens whose tap t is outside): t = 1 right column, t = 2 bottom row, t = 3 either
  auto dv = [&](int t, int co) {
    const float r = bs[co], bt = bs[C + co], cn = bs[2 * C + co];
    return colsum[co] - (t == 0 ? 0.f : t == 1 ? r : t == 2 ? bt : r + bt - cn);
  };
  int bid = blockIdx.x;
  if (bid < 4 * n) {                               // gWc[co][m][t] += sum_ci dweff[co][t C + ci] W1[m][ci] + dv[t][co] b1[m]
    const int t = bid & 3, tl = bid >> 2;
    const int m0 = (tl % nt) * CT, co0 = (tl / nt) * CT;
    float acc[4][4] = {};
    tile64(acc, C, As, Bs,
           [&](int r, int k) { return dweff[(long)(co0 + r) * 4 * C + t * C + k]; },
           [&](int k, int c) { return W1[(long)(m0 + c) * C + k]; }, true, true);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        gWc[((long)(co0 + ty + 16 * i) * C + m0 + tx + 16 * j) * 4 + t] += acc[i][j] + dv(t, co0 + ty + 16 * i) * b1[m0 + tx + 16 * j];
    return;
  }
  bid -= 4 * n;
  if (bid < 4 * n) {                               // gW1[m][ci] += sum_(co, t) Wc[co][m][t] dweff[co][t C + ci]: k = 4 co + t, a quarter of the co's
    const int q = bid & 3, tl = bid >> 2;
    const int ci0 = (tl % nt) * CT, m0 = (tl / nt) * CT;
    const int kq = q * C;                          // k range [q C, (q + 1) C)
    float acc[4][4] = {};
    tile64(acc, C, As, Bs,
           [&](int r, int k) { const int kk = kq + k; return Wc[((long)(kk >> 2) * C + m0 + r) * 4 + (kk & 3)]; },
           [&](int k, int c) { const int kk = kq + k; return dweff[(long)(kk >> 2) * 4 * C + (kk & 3) * C + ci0 + c]; }, true, false);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) atomicAdd(gW1 + (long)(m0 + ty + 16 * i) * C + ci0 + tx + 16 * j, acc[i][j]);
    return;
  }
  // ---- the last C / 32 workgroups: gb1[m] += sum_t sum_co Wc[co][m][t] dv[t][co] and gbc += colsum, 32 values of m each, eight lanes
  //      per m over the co's (one workgroup walking all of it serially was the longest pole of the launch: 40 of its 51 us at C = 192)
  bid -= 4 * n;
  for (int i = tid; i < 4 * C; i += 256) dvs[i] = dv(i / C, i % C);
  __syncthreads();
  const int m = bid * 32 + (tid >> 3), part = tid & 7;
  float acc = 0.f;
  for (int co = part; co < C; co += 8) {
    const float4 wc = *(const float4*)(Wc + ((long)co * C + m) * 4);
    acc += wc.x * dvs[co] + wc.y * dvs[C + co] + wc.z * dvs[2 * C + co] + wc.w * dvs[3 * C + co];
  }
  acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4);
  if (part == 0) {
    gb1[m] += acc;
    gbc[m] += colsum[m];                           // conv1.bias: beff carries bc with coefficient 1
  }
}

inline int nblk(long n) { return (int)((n + 255) / 256); }

}  // namespace

extern "C" int sodt_convmlp_compose(const float* fc1_w, const float* fc1_b, const float* conv_w, const float* conv_b, void* weff,
                                    void* weffT, float* beff, float* vtap, int C, int dtype, sodt_stream_t st) {
  if (!fc1_w || !fc1_b || !conv_w || !conv_b || !weff || !weffT || !beff || !vtap || C <= 0 || (C % 64) || dtype != SODT_BF16 ||
      (((uintptr_t)conv_w) & 15))
    return SODT_EINVAL;
  hipLaunchKernelGGL(convmlp_compose_kernel, dim3(C / 32, C / 64 + 1, 4), dim3(256), 0, (hipStream_t)st, fc1_w, fc1_b, conv_w, conv_b,
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
  if (!dweff || !colsum || !bs || !fc1_w || !fc1_b || !conv_w || !g_conv_w || !g_conv_b || !g_fc1_w || !g_fc1_b || C <= 0 || (C % 64) ||
      C > CONVMLP_MAXC || ((((uintptr_t)conv_w) | ((uintptr_t)g_conv_w)) & 15))
    return SODT_EINVAL;
  const int n = (C / 64) * (C / 64);
  hipLaunchKernelGGL(convmlp_decompose_kernel, dim3(8 * n + C / 32), dim3(256), 0, (hipStream_t)st, dweff, colsum, bs, fc1_w, fc1_b, conv_w,
                     g_conv_w, g_conv_b, g_fc1_w, g_fc1_b, C);
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}
