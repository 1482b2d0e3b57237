// LayerNorm forward / backward over token-major rows (HBM-bound; one wave per row,
// 16-byte loads, wave-64 shuffle reductions, two-pass variance from registers).
// Reference ops: nn.LayerNorm(C), eps 1e-5 (backbone_vit.py:1048,1054,837).
#include "common.h"
#include "../../include/sodt_hip.h"

namespace {


template <typename T, int MAXCH>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ stats, int M, int C) {
  constexpr int KPL = TT<T>::KPL;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nch = C / KPL;
  const float invC = 1.0f / (float)C;
  for (long row = (long)blockIdx.x * 4 + wid; row < M; row += (long)gridDim.x * 4) {
    float v[MAXCH][KPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
      const int ch = lane + 64 * i;
      if (ch < nch) {
        unpack<T>(*(const uint4*)(x + row * C + ch * KPL), v[i]);
#pragma unroll
        for (int j = 0; j < KPL; ++j) s += v[i][j];
      }
    }
    const float mean = wave_sum(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
      const int ch = lane + 64 * i;
      if (ch < nch) {
#pragma unroll
        for (int j = 0; j < KPL; ++j) { const float d = v[i][j] - mean; q += d * d; }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) * invC + 1e-5f);
    if (lane == 0 && stats) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
      const int ch = lane + 64 * i;
      if (ch < nch) {
        float o[KPL];
#pragma unroll
        for (int j = 0; j < KPL; ++j) o[j] = (v[i][j] - mean) * rstd * gamma[ch * KPL + j] + beta[ch * KPL + j];
        *(uint4*)(y + row * C + ch * KPL) = pack<T>(o);
      }
    }
  }
}

template <typename T, int MAXCH>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ stats, const float* __restrict__ gamma,
                                                     const T* __restrict__ dres, T* __restrict__ dx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     int M, int C) {
  constexpr int KPL = TT<T>::KPL;
  __shared__ float red[4][64 * MAXCH * KPL + 1];   // per-wave partials of one parameter vector at a time
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nch = C / KPL;
  const float invC = 1.0f / (float)C;
  float ag[MAXCH][KPL], ab[MAXCH][KPL], gm[MAXCH][KPL];
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int ch = lane + 64 * i;
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
      ag[i][j] = 0.f; ab[i][j] = 0.f;
      gm[i][j] = (ch < nch) ? gamma[ch * KPL + j] : 0.f;
    }
  }
  for (long row = (long)blockIdx.x * 4 + wid; row < M; row += (long)gridDim.x * 4) {
    const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
    float xh[MAXCH][KPL], g[MAXCH][KPL];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
      const int ch = lane + 64 * i;
      if (ch < nch) {
        float a[KPL], d[KPL];
        unpack<T>(*(const uint4*)(x + row * C + ch * KPL), a);
        unpack<T>(*(const uint4*)(dy + row * C + ch * KPL), d);
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
          xh[i][j] = (a[j] - mean) * rstd;
          g[i][j] = d[j] * gm[i][j];
          c1 += g[i][j]; c2 += g[i][j] * xh[i][j];
          ag[i][j] += d[j] * xh[i][j]; ab[i][j] += d[j];
        }
      }
    }
    c1 = wave_sum(c1) * invC; c2 = wave_sum(c2) * invC;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
      const int ch = lane + 64 * i;
      if (ch < nch) {
        float o[KPL];
#pragma unroll
        for (int j = 0; j < KPL; ++j) o[j] = rstd * (g[i][j] - c1 - xh[i][j] * c2);
        if (dres) {
          float r[KPL];
          unpack<T>(*(const uint4*)(dres + row * C + ch * KPL), r);
#pragma unroll
          for (int j = 0; j < KPL; ++j) o[j] += r[j];
        }
        *(uint4*)(dx + row * C + ch * KPL) = pack<T>(o);
      }
    }
  }
  // block-level reduce of dgamma / dbeta partials, then one atomic per column
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MAXCH; ++i)
#pragma unroll
      for (int j = 0; j < KPL; ++j)
        red[wid][(lane + 64 * i) * KPL + j] = pass == 0 ? ag[i][j] : ab[i][j];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      const float t = red[0][c] + red[1][c] + red[2][c] + red[3][c];
      atomicAdd((pass == 0 ? dgamma : dbeta) + c, t);
    }
  }
}

// Rows of at most 32 chunks (C = 192 in bf16 is 24): one row per 32-lane half-wave, two rows per wave -- twice
// the lanes carry data compared with the one-row-per-wave kernels above.
__device__ __forceinline__ float half_sum(float v) {
  v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); v += __shfl_xor(v, 16);
  return v;
}

template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_half_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, T* __restrict__ y,
                                                          float* __restrict__ stats, int M, int C) {
  constexpr int KPL = TT<T>::KPL;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, l32 = lane & 31, sub = lane >> 5;
  const int nch = C / KPL;
  const bool act = l32 < nch;
  const float invC = 1.0f / (float)C;
  float gm[KPL], bt[KPL];
#pragma unroll
  for (int j = 0; j < KPL; ++j) { gm[j] = act ? gamma[l32 * KPL + j] : 0.f; bt[j] = act ? beta[l32 * KPL + j] : 0.f; }
  for (long row = ((long)blockIdx.x * 4 + wid) * 2 + sub; row < M; row += (long)gridDim.x * 8) {
    float v[KPL];
#pragma unroll
    for (int j = 0; j < KPL; ++j) v[j] = 0.f;
    if (act) unpack<T>(*(const uint4*)(x + row * C + l32 * KPL), v);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < KPL; ++j) s += v[j];
    const float mean = half_sum(s) * invC;
    float q = 0.f;
    if (act) {
#pragma unroll
      for (int j = 0; j < KPL; ++j) { const float d = v[j] - mean; q += d * d; }
    }
    const float rstd = rsqrtf(half_sum(q) * invC + 1e-5f);
    if (l32 == 0 && stats) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
    if (act) {
      float o[KPL];
#pragma unroll
      for (int j = 0; j < KPL; ++j) o[j] = (v[j] - mean) * rstd * gm[j] + bt[j];
      *(uint4*)(y + row * C + l32 * KPL) = pack<T>(o);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_half_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                          const float* __restrict__ stats, const float* __restrict__ gamma,
                                                          const T* __restrict__ dres, T* __restrict__ dx,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                          int M, int C) {
  constexpr int KPL = TT<T>::KPL;
  __shared__ float red[8][32 * KPL + 1];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, l32 = lane & 31, sub = lane >> 5;
  const int nch = C / KPL;
  const bool act = l32 < nch;
  const float invC = 1.0f / (float)C;
  float ag[KPL], ab[KPL], gm[KPL];
#pragma unroll
  for (int j = 0; j < KPL; ++j) { ag[j] = 0.f; ab[j] = 0.f; gm[j] = act ? gamma[l32 * KPL + j] : 0.f; }
  for (long row = ((long)blockIdx.x * 4 + wid) * 2 + sub; row < M; row += (long)gridDim.x * 8) {
    const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
    float xh[KPL], g[KPL];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int j = 0; j < KPL; ++j) { xh[j] = 0.f; g[j] = 0.f; }
    if (act) {
      float a[KPL], d[KPL];
      unpack<T>(*(const uint4*)(x + row * C + l32 * KPL), a);
      unpack<T>(*(const uint4*)(dy + row * C + l32 * KPL), d);
#pragma unroll
      for (int j = 0; j < KPL; ++j) {
        xh[j] = (a[j] - mean) * rstd;
        g[j] = d[j] * gm[j];
        c1 += g[j]; c2 += g[j] * xh[j];
        ag[j] += d[j] * xh[j]; ab[j] += d[j];
      }
    }
    c1 = half_sum(c1) * invC; c2 = half_sum(c2) * invC;
    if (act) {
      float o[KPL];
#pragma unroll
      for (int j = 0; j < KPL; ++j) o[j] = rstd * (g[j] - c1 - xh[j] * c2);
      if (dres) {
        float r[KPL];
        unpack<T>(*(const uint4*)(dres + row * C + l32 * KPL), r);
#pragma unroll
        for (int j = 0; j < KPL; ++j) o[j] += r[j];
      }
      *(uint4*)(dx + row * C + l32 * KPL) = pack<T>(o);
    }
  }
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KPL; ++j) red[wid * 2 + sub][l32 * KPL + j] = pass == 0 ? ag[j] : ab[j];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) t += red[r][c];
      atomicAdd((pass == 0 ? dgamma : dbeta) + c, t);
    }
  }
}

}  // namespace

extern "C" int sodt_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats,
                                  int M, int C, int dtype, sodt_stream_t st) {
  const int kpl = dtype == SODT_BF16 ? 8 : 4;
  if (M <= 0 || C <= 0 || (C % kpl) || C > 256 * kpl || !x || !y || !gamma || !beta) return SODT_EINVAL;
  long blocks = ((long)M + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  if (C / kpl <= 32) {
    long hb = ((long)M + 7) / 8;
    if (hb > 4096) hb = 4096;
    if (dtype == SODT_BF16) hipLaunchKernelGGL(ln_fwd_half_kernel<bf16>, dim3((unsigned)hb), dim3(256), 0, (hipStream_t)st, (const bf16*)x, gamma, beta, (bf16*)y, stats, M, C);
    else hipLaunchKernelGGL(ln_fwd_half_kernel<float>, dim3((unsigned)hb), dim3(256), 0, (hipStream_t)st, (const float*)x, gamma, beta, (float*)y, stats, M, C);
    return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
  }
  const int nc = (C / kpl + 63) / 64;
#define LNF(TY, NC) hipLaunchKernelGGL((ln_fwd_kernel<TY, NC>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)st, \
                       (const TY*)x, gamma, beta, (TY*)y, stats, M, C)
  if (dtype == SODT_BF16) { if (nc == 1) LNF(bf16, 1); else if (nc == 2) LNF(bf16, 2); else LNF(bf16, 4); }
  else { if (nc == 1) LNF(float, 1); else if (nc == 2) LNF(float, 2); else LNF(float, 4); }
#undef LNF
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

extern "C" int sodt_layernorm_bwd(const void* dy, const void* x, const float* stats, const float* gamma,
                                  const void* dres, void* dx, float* dgamma, float* dbeta,
                                  int M, int C, int dtype, sodt_stream_t st) {
  const int kpl = dtype == SODT_BF16 ? 8 : 4;
  if (M <= 0 || C <= 0 || (C % kpl) || C > 256 * kpl || !x || !dy || !dx || !stats || !gamma || !dgamma || !dbeta)
    return SODT_EINVAL;
  long blocks = ((long)M + 3) / 4;
  if (blocks > 1024) blocks = 1024;
  if (C / kpl <= 32) {
    long hb = ((long)M + 7) / 8;
    if (hb > 1024) hb = 1024;
    if (dtype == SODT_BF16) hipLaunchKernelGGL(ln_bwd_half_kernel<bf16>, dim3((unsigned)hb), dim3(256), 0, (hipStream_t)st, (const bf16*)dy, (const bf16*)x, stats, gamma, (const bf16*)dres, (bf16*)dx, dgamma, dbeta, M, C);
    else hipLaunchKernelGGL(ln_bwd_half_kernel<float>, dim3((unsigned)hb), dim3(256), 0, (hipStream_t)st, (const float*)dy, (const float*)x, stats, gamma, (const float*)dres, (float*)dx, dgamma, dbeta, M, C);
    return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
  }
  const int nc = (C / kpl + 63) / 64;
#define LNB(TY, NC) hipLaunchKernelGGL((ln_bwd_kernel<TY, NC>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)st, \
                       (const TY*)dy, (const TY*)x, stats, gamma, (const TY*)dres, (TY*)dx, dgamma, dbeta, M, C)
  if (dtype == SODT_BF16) { if (nc == 1) LNB(bf16, 1); else if (nc == 2) LNB(bf16, 2); else LNB(bf16, 4); }
  else { if (nc == 1) LNB(float, 1); else if (nc == 2) LNB(float, 2); else LNB(float, 4); }
#undef LNB
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}
