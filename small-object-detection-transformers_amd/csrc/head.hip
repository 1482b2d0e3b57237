// HBM-bound pieces of the YOLOv5 head and small data-movement kernels (gfx950):
//   BatchNorm2d(eps 1e-3, momentum 0.03) + SiLU of common.py:38-50 (train-mode batch
//   statistics come from the GEMM epilogue's f64 column sums), nn.Upsample(nearest) /
//   Concat slot copies (models/model.yaml:66-72), Detect's permutes and eval decode
//   (model.py:55-64), parameter preparation.  All token-major, 16-byte accesses.
#include "common.h"
#include "../../include/sodt_hip.h"

namespace {

__global__ void bn_finalize_kernel(const double* __restrict__ stats, float* __restrict__ mr, float* __restrict__ rmean,
                                   float* __restrict__ rvar, long count, int C, float eps, float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (stats) {
    double s1 = 0.0, s2 = 0.0;
    for (int r = 0; r < SODT_STATS_REPL; ++r) { s1 += stats[(size_t)r * 2 * C + c]; s2 += stats[(size_t)r * 2 * C + C + c]; }
    const double mean = s1 / (double)count;
    double var = s2 / (double)count - mean * mean;
    if (var < 0.0) var = 0.0;
    mr[c] = (float)mean;
    mr[C + c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) {
      const double unb = count > 1 ? var * (double)count / (double)(count - 1) : var;
      rmean[c] = (1.0f - momentum) * rmean[c] + momentum * (float)mean;
      rvar[c] = (1.0f - momentum) * rvar[c] + momentum * (float)unb;
    }
  } else {   // eval: running statistics
    mr[c] = rmean[c];
    mr[C + c] = rsqrtf(rvar[c] + eps);
  }
}

__global__ void bn_affine_kernel(const float* __restrict__ mr, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, float* __restrict__ scale, float* __restrict__ shift, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float s = gamma[c] * mr[C + c];
  scale[c] = s;
  shift[c] = beta[c] - mr[c] * s;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_silu_fwd_kernel(const T* __restrict__ z, const float* __restrict__ mr,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         T* __restrict__ y, int ldy, long M, int C) {
  constexpr int KPL = TT<T>::KPL;
  const int CH = C / KPL;
  const long total = M * CH;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long m = i / CH; const int c0 = (int)(i - m * CH) * KPL;
    float v[KPL];
    unpack<T>(*(const uint4*)(z + m * C + c0), v);
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
      const int c = c0 + j;
      const float a = (v[j] - mr[c]) * mr[C + c] * gamma[c] + beta[c];
      v[j] = a * sigmoid_f(a);
    }
    *(uint4*)(y + m * ldy + c0) = pack<T>(v);
  }
}

// per-channel  stats[r][0][c] += sum_m z,  stats[r][1][c] += sum_m z^2  (the BatchNorm batch statistics of a convolution whose kernel has no
// statistics epilogue: csrc/conv3.hip); replica r = blockIdx.x % SODT_STATS_REPL as in the GEMM epilogue
template <typename T>
__global__ __launch_bounds__(256) void col_stats_kernel(const T* __restrict__ z, int ldz, double* __restrict__ stats, long M, int C) {
  constexpr int KPL = TT<T>::KPL;
  __shared__ float sred[2][256 * KPL];
  const int CH = C / KPL;
  const int rpp = 256 / CH;                  // rows per pass
  const int tid = threadIdx.x;
  const int rl = tid / CH, ch = tid - rl * CH;
  const bool act = rl < rpp;
  const int c0 = ch * KPL;
  float a0[KPL], a1[KPL];
#pragma unroll
  for (int j = 0; j < KPL; ++j) { a0[j] = 0.f; a1[j] = 0.f; }
  if (act) {
    const long stride = (long)gridDim.x * rpp;
    for (long m = (long)blockIdx.x * rpp + rl; m < M; m += 4 * stride) {
      uint4 vq[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long mm = m + u * stride;
        vq[u] = mm < M ? *(const uint4*)(z + mm * ldz + c0) : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float v[KPL];
        unpack<T>(vq[u], v);
#pragma unroll
        for (int j = 0; j < KPL; ++j) { a0[j] += v[j]; a1[j] = fmaf(v[j], v[j], a1[j]); }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < KPL; ++j) { sred[0][tid * KPL + j] = a0[j]; sred[1][tid * KPL + j] = a1[j]; }
  __syncthreads();
  double* st = stats + (size_t)(blockIdx.x % SODT_STATS_REPL) * 2 * C;
  for (int c = tid; c < C; c += 256) {
    const int cc = c / KPL, j = c - cc * KPL;
    double s0 = 0.0, s1 = 0.0;
    for (int r = 0; r < rpp; ++r) { s0 += sred[0][(r * CH + cc) * KPL + j]; s1 += sred[1][(r * CH + cc) * KPL + j]; }
    atomicAdd(st + c, s0);
    atomicAdd(st + C + c, s1);
  }
}

// per-channel  red[0][c] += sum_m g,  red[1][c] += sum_m g * xhat,   g = dy * silu'(a)
template <typename T>
__global__ __launch_bounds__(256) void bn_silu_bwd_reduce_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ z,
                                                                const float* __restrict__ mr, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, double* __restrict__ red,
                                                                long M, int C) {
  constexpr int KPL = TT<T>::KPL;
  __shared__ float sred[2][256 * KPL];
  const int CH = C / KPL;
  const int rpp = 256 / CH;                  // rows per pass
  const int tid = threadIdx.x;
  const int rl = tid / CH, ch = tid - rl * CH;
  const bool act = rl < rpp;
  const int c0 = ch * KPL;
  float mu[KPL], rs[KPL], ga[KPL], be[KPL], a0[KPL], a1[KPL];
#pragma unroll
  for (int j = 0; j < KPL; ++j) {
    const int c = act ? c0 + j : 0;
    mu[j] = mr[c]; rs[j] = mr[C + c]; ga[j] = gamma[c]; be[j] = beta[c]; a0[j] = 0.f; a1[j] = 0.f;
  }
  if (act) {
    // four rows per trip, their eight 16-byte loads issued before the first is used: with one row per trip a thread has 32 bytes in
    // flight and the launch runs at half the HBM rate (2.6 TB/s at 512 K rows x 64 channels)
    const long stride = (long)gridDim.x * rpp;
    for (long m = (long)blockIdx.x * rpp + rl; m < M; m += 4 * stride) {
      uint4 dq[4], vq[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long mm = m + u * stride;
        const bool ok = mm < M;
        dq[u] = ok ? *(const uint4*)(dy + mm * lddy + c0) : make_uint4(0u, 0u, 0u, 0u);
        vq[u] = ok ? *(const uint4*)(z + mm * C + c0) : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float d[KPL], v[KPL];
        unpack<T>(dq[u], d);
        unpack<T>(vq[u], v);
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
          const float xh = (v[j] - mu[j]) * rs[j];
          const float a = xh * ga[j] + be[j];
          const float sg = sigmoid_f(a);
          const float gg = d[j] * sg * (1.0f + a * (1.0f - sg));       // (a zero-filled row: d = 0 -> gg = 0)
          a0[j] += gg; a1[j] += gg * xh;
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < KPL; ++j) { sred[0][tid * KPL + j] = a0[j]; sred[1][tid * KPL + j] = a1[j]; }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    const int cc = c / KPL, j = c - cc * KPL;
    double s0 = 0.0, s1 = 0.0;
    for (int r = 0; r < rpp; ++r) { s0 += sred[0][(r * CH + cc) * KPL + j]; s1 += sred[1][(r * CH + cc) * KPL + j]; }
    atomicAdd(red + c, s0);
    atomicAdd(red + C + c, s1);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_silu_bwd_apply_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ z,
                                                               const float* __restrict__ mr, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, const double* __restrict__ red,
                                                               T* __restrict__ dz, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, long M, int C) {
  constexpr int KPL = TT<T>::KPL;
  const int CH = C / KPL;
  const long total = M * CH;
  const double invM = 1.0 / (double)M;
  if (blockIdx.x == 0) {
    for (int c = threadIdx.x; c < C; c += 256) { dbeta[c] += (float)red[c]; dgamma[c] += (float)red[C + c]; }
  }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long m = i / CH; const int c0 = (int)(i - m * CH) * KPL;
    float d[KPL], v[KPL], o[KPL];
    unpack<T>(*(const uint4*)(dy + m * lddy + c0), d);
    unpack<T>(*(const uint4*)(z + m * C + c0), v);
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
      const int c = c0 + j;
      const float rs = mr[C + c], ga = gamma[c];
      const float xh = (v[j] - mr[c]) * rs;
      const float a = xh * ga + beta[c];
      const float sg = sigmoid_f(a);
      const float gg = d[j] * sg * (1.0f + a * (1.0f - sg));
      const float mg = (float)(red[c] * invM), mgx = (float)(red[C + c] * invM);
      o[j] = ga * rs * (gg - mg - xh * mgx);
    }
    *(uint4*)(dz + m * C + c0) = pack<T>(o);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void copy_rows_kernel(const T* __restrict__ src, int lds_, T* __restrict__ dst, int ldd,
                                                       int B, int Ho, int Wo, int shr, int C) {
  constexpr int KPL = TT<T>::KPL;
  const int CH = C / KPL;
  const long total = (long)B * Ho * Wo * CH;
  const int Hi = Ho >> shr, Wi = Wo >> shr;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long m = i / CH; const int c0 = (int)(i - m * CH) * KPL;
    const int b = (int)(m / ((long)Ho * Wo)); const int rem = (int)(m - (long)b * Ho * Wo);
    const int y = rem / Wo, x = rem - y * Wo;
    const long sm = ((long)b * Hi + (y >> shr)) * Wi + (x >> shr);
    *(uint4*)(dst + m * ldd + c0) = *(const uint4*)(src + sm * lds_ + c0);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gather_sum_rows_kernel(const T* __restrict__ d, int ldd, T* __restrict__ dsrc, int lds_,
                                                             int B, int Hs, int Ws, int shr, int C, int accumulate) {
  constexpr int KPL = TT<T>::KPL;
  const int CH = C / KPL;
  const long total = (long)B * Hs * Ws * CH;
  const int f = 1 << shr;
  const int Hd = Hs << shr, Wd = Ws << shr;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long m = i / CH; const int c0 = (int)(i - m * CH) * KPL;
    const int b = (int)(m / ((long)Hs * Ws)); const int rem = (int)(m - (long)b * Hs * Ws);
    const int y = rem / Ws, x = rem - y * Ws;
    float acc[KPL];
#pragma unroll
    for (int j = 0; j < KPL; ++j) acc[j] = 0.f;
    if (accumulate) unpack<T>(*(const uint4*)(dsrc + m * lds_ + c0), acc);
    for (int a = 0; a < f; ++a)
      for (int bb = 0; bb < f; ++bb) {
        const long dm = ((long)b * Hd + (y << shr) + a) * Wd + (x << shr) + bb;
        float v[KPL];
        unpack<T>(*(const uint4*)(d + dm * ldd + c0), v);
#pragma unroll
        for (int j = 0; j < KPL; ++j) acc[j] += v[j];
      }
    *(uint4*)(dsrc + m * lds_ + c0) = pack<T>(acc);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void detect_unpermute_kernel(const float* __restrict__ dpred, T* __restrict__ dz, int ldz,
                                                              int B, int HW, int na, int no) {
  const long total = (long)B * HW * ldz;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long m = i / ldz; const int n = (int)(i - m * ldz);
    float v = 0.f;
    if (n < na * no) {
      const long b = m / HW, p = m - b * HW;
      const int a = n / no, o = n - a * no;
      v = dpred[((b * na + a) * HW + p) * no + o];
    }
    dz[i] = from_f<T>(v);
  }
}

__global__ __launch_bounds__(256) void detect_decode_kernel(const float* __restrict__ raw, const float* __restrict__ anchor_grid,
                                                           float* __restrict__ zout, int B, int na, int ny, int nx, int no,
                                                           float stride) {
  const long total = (long)B * na * ny * nx * no;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int o = (int)(i % no); long r = i / no;
    const int x = (int)(r % nx); r /= nx;
    const int y = (int)(r % ny); r /= ny;
    const int a = (int)(r % na);
    const float s = sigmoid_f(raw[i]);
    float v = s;
    if (o == 0) v = (s * 2.f - 0.5f + (float)x) * stride;
    else if (o == 1) v = (s * 2.f - 0.5f + (float)y) * stride;
    else if (o == 2) v = (s * 2.f) * (s * 2.f) * anchor_grid[a * 2];
    else if (o == 3) v = (s * 2.f) * (s * 2.f) * anchor_grid[a * 2 + 1];
    zout[i] = v;   // (B, na, ny, nx, no) flat == (B, na*ny*nx, no)
  }
}

template <typename T>
__global__ __launch_bounds__(256) void prep_kernel(const sodt_prep_desc* __restrict__ tab, int n) {
  const sodt_prep_desc d = tab[blockIdx.y];
  const int dims[3] = {d.d0, d.d1, d.d2};
  const int sstr[3] = {d.d1 * d.d2, d.d2, 1};
  const int e0 = dims[d.p0], e1 = dims[d.p1], e2 = dims[d.p2];
  const long total = (long)e0 * e1 * e2;
  T* dst = (T*)d.dst;
  if (d.d2 == 1 && d.p0 == 1 && d.p2 == 0) {
    // plain 2-D transpose dst[k][n] = src[n][k] (every wT of an nn.Linear / 1x1 Conv2d): 32 x 32 tiles through LDS so that
    // both the f32 reads and the run-dtype writes are coalesced (the generic loop below reads with stride K)
    __shared__ float tile[32][33];
    const int N = d.d0, K = d.d1;
    const int tn = (N + 31) >> 5, tk = (K + 31) >> 5;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int tIdx = blockIdx.x; tIdx < tn * tk; tIdx += gridDim.x) {
      const int n0 = (tIdx / tk) << 5, k0 = (tIdx % tk) << 5;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + ty + 8 * j, k = k0 + tx;
        tile[ty + 8 * j][tx] = (n < N && k < K) ? d.src[(long)n * K + k] : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + ty + 8 * j, n = n0 + tx;
        if (k < K && n < N) dst[(long)k * d.dst_ld + n] = from_f<T>(tile[tx][ty + 8 * j]);
      }
      __syncthreads();
    }
    return;
  }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int i2 = (int)(i % e2); const long r = i / e2;
    const int i1 = (int)(r % e1); const int i0 = (int)(r / e1);
    const long s = (long)i0 * sstr[d.p0] + (long)i1 * sstr[d.p1] + (long)i2 * sstr[d.p2];
    dst[(long)i0 * d.dst_ld + (long)i1 * (d.inner_ld > 0 ? d.inner_ld : e2) + i2] = from_f<T>(d.src[s]);
  }
}

__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows,
                                                           int cols, int accumulate) {
  const long total = (long)rows * cols;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int r = (int)(i / cols), c = (int)(i - (long)r * cols);
    if (accumulate) dst[(long)c * rows + r] += src[i];
    else dst[(long)c * rows + r] = src[i];
    if (accumulate == 2) const_cast<float*>(src)[i] = 0.f;
  }
}

template <typename S, typename D>
__global__ __launch_bounds__(256) void cast_kernel(const S* __restrict__ s, D* __restrict__ d, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) d[i] = from_f<D>(to_f(s[i]));
}

// out[r][c] (f32) += sum_b d[b][r][c]   (pos_embed gradient: backbone_vit.py:215-217 backward)
template <typename T>
__global__ __launch_bounds__(256) void batch_sum_kernel(const T* __restrict__ d, float* __restrict__ out, int B, long RC) {
  constexpr int KPL = TT<T>::KPL;
  if ((RC % KPL) == 0 && ((uintptr_t)d & 15) == 0 && ((uintptr_t)out & 15) == 0) {     // 16-byte chunks (the engine's case)
    const long nch = RC / KPL;
    for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < nch; c += (long)gridDim.x * 256) {
      float s[KPL];
#pragma unroll
      for (int j = 0; j < KPL; ++j) s[j] = 0.f;
      for (int b = 0; b < B; ++b) {
        float f[KPL];
        unpack<T>(*(const uint4*)(d + (long)b * RC + c * KPL), f);
#pragma unroll
        for (int j = 0; j < KPL; ++j) s[j] += f[j];
      }
#pragma unroll
      for (int j = 0; j < KPL; j += 4) {
        float4 o = *(float4*)(out + c * KPL + j);
        o.x += s[j]; o.y += s[j + 1]; o.z += s[j + 2]; o.w += s[j + 3];
        *(float4*)(out + c * KPL + j) = o;
      }
    }
    return;
  }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < RC; i += (long)gridDim.x * 256) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += to_f(d[(long)b * RC + i]);
    out[i] += s;
  }
}

inline unsigned nblocks(long work, int cap = 4096) {
  long b = (work + 255) / 256;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (unsigned)b;
}

}  // namespace

extern "C" int sodt_bn_finalize(const double* stats, float* mean_rstd, float* running_mean, float* running_var,
                                long count, int C, float eps, float momentum, sodt_stream_t st) {
  if (!mean_rstd || C <= 0 || (!stats && (!running_mean || !running_var))) return SODT_EINVAL;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)st, stats, mean_rstd,
                     running_mean, running_var, count, C, eps, momentum);
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

extern "C" int sodt_bn_affine(const float* mean_rstd, const float* gamma, const float* beta, float* scale, float* shift,
                              int C, sodt_stream_t st) {
  if (!mean_rstd || !gamma || !beta || !scale || !shift || C <= 0) return SODT_EINVAL;
  hipLaunchKernelGGL(bn_affine_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)st, mean_rstd, gamma, beta, scale, shift, C);
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

extern "C" int sodt_bn_silu_fwd(const void* z, const float* mean_rstd, const float* gamma, const float* beta,
                                void* y, int ldy, long M, int C, int dtype, sodt_stream_t st) {
  const int kpl = dtype == SODT_BF16 ? 8 : 4;
  if (!z || !y || M <= 0 || C <= 0 || (C % kpl) || (ldy % kpl)) return SODT_EINVAL;
  const unsigned gr = nblocks(M * (C / kpl));
  if (dtype == SODT_BF16) hipLaunchKernelGGL(bn_silu_fwd_kernel<bf16>, dim3(gr), dim3(256), 0, (hipStream_t)st, (const bf16*)z, mean_rstd, gamma, beta, (bf16*)y, ldy, M, C);
  else if (dtype == SODT_F32) hipLaunchKernelGGL(bn_silu_fwd_kernel<float>, dim3(gr), dim3(256), 0, (hipStream_t)st, (const float*)z, mean_rstd, gamma, beta, (float*)y, ldy, M, C);
  else return SODT_EINVAL;
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

extern "C" int sodt_bn_silu_bwd_reduce(const void* dy, int lddy, const void* z, const float* mean_rstd,
                                       const float* gamma, const float* beta, double* red, long M, int C,
                                       int dtype, sodt_stream_t st) {
  const int kpl = dtype == SODT_BF16 ? 8 : 4;
  if (!dy || !z || !red || M <= 0 || C <= 0 || (C % kpl) || (lddy % kpl) || C / kpl > 256) return SODT_EINVAL;
  const int rpp = 256 / (C / kpl);
  long gr = (M + rpp - 1) / rpp; if (gr > 1024) gr = 1024;
  if (dtype == SODT_BF16) hipLaunchKernelGGL(bn_silu_bwd_reduce_kernel<bf16>, dim3((unsigned)gr), dim3(256), 0, (hipStream_t)st, (const bf16*)dy, lddy, (const bf16*)z, mean_rstd, gamma, beta, red, M, C);
  else if (dtype == SODT_F32) hipLaunchKernelGGL(bn_silu_bwd_reduce_kernel<float>, dim3((unsigned)gr), dim3(256), 0, (hipStream_t)st, (const float*)dy, lddy, (const float*)z, mean_rstd, gamma, beta, red, M, C);
  else return SODT_EINVAL;
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

extern "C" int sodt_col_stats(const void* z, int ldz, double* stats, long M, int C, int dtype, sodt_stream_t st) {
  const int kpl = dtype == SODT_BF16 ? 8 : 4;
  if (!z || !stats || M <= 0 || C <= 0 || (C % kpl) || (ldz % kpl) || C / kpl > 256 || (((uintptr_t)z) & 15)) return SODT_EINVAL;
  const int rpp = 256 / (C / kpl);
  long gr = (M + 4L * rpp - 1) / (4L * rpp); if (gr > 1024) gr = 1024;
  if (dtype == SODT_BF16) hipLaunchKernelGGL(col_stats_kernel<bf16>, dim3((unsigned)gr), dim3(256), 0, (hipStream_t)st, (const bf16*)z, ldz, stats, M, C);
  else if (dtype == SODT_F32) hipLaunchKernelGGL(col_stats_kernel<float>, dim3((unsigned)gr), dim3(256), 0, (hipStream_t)st, (const float*)z, ldz, stats, M, C);
  else return SODT_EINVAL;
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

extern "C" int sodt_bn_silu_bwd_apply(const void* dy, int lddy, const void* z, const float* mean_rstd,
                                      const float* gamma, const float* beta, const double* red, void* dz,
                                      float* dgamma, float* dbeta, long M, int C, int dtype, sodt_stream_t st) {
  const int kpl = dtype == SODT_BF16 ? 8 : 4;
  if (!dy || !z || !red || !dz || !dgamma || !dbeta || M <= 0 || C <= 0 || (C % kpl) || (lddy % kpl)) return SODT_EINVAL;
  const unsigned gr = nblocks(M * (C / kpl));
  if (dtype == SODT_BF16) hipLaunchKernelGGL(bn_silu_bwd_apply_kernel<bf16>, dim3(gr), dim3(256), 0, (hipStream_t)st, (const bf16*)dy, lddy, (const bf16*)z, mean_rstd, gamma, beta, red, (bf16*)dz, dgamma, dbeta, M, C);
  else if (dtype == SODT_F32) hipLaunchKernelGGL(bn_silu_bwd_apply_kernel<float>, dim3(gr), dim3(256), 0, (hipStream_t)st, (const float*)dy, lddy, (const float*)z, mean_rstd, gamma, beta, red, (float*)dz, dgamma, dbeta, M, C);
  else return SODT_EINVAL;
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

extern "C" int sodt_copy_rows(const void* src, int lds_, void* dst, int ldd, int B, int Ho, int Wo, int shr, int C,
                              int dtype, sodt_stream_t st) {
  const int kpl = dtype == SODT_BF16 ? 8 : 4;
  if (!src || !dst || B <= 0 || Ho <= 0 || Wo <= 0 || shr < 0 || C <= 0 || (C % kpl) || (lds_ % kpl) || (ldd % kpl)) return SODT_EINVAL;
  if ((Ho & ((1 << shr) - 1)) || (Wo & ((1 << shr) - 1))) return SODT_EINVAL;
  const unsigned gr = nblocks((long)B * Ho * Wo * (C / kpl));
  if (dtype == SODT_BF16) hipLaunchKernelGGL(copy_rows_kernel<bf16>, dim3(gr), dim3(256), 0, (hipStream_t)st, (const bf16*)src, lds_, (bf16*)dst, ldd, B, Ho, Wo, shr, C);
  else if (dtype == SODT_F32) hipLaunchKernelGGL(copy_rows_kernel<float>, dim3(gr), dim3(256), 0, (hipStream_t)st, (const float*)src, lds_, (float*)dst, ldd, B, Ho, Wo, shr, C);
  else return SODT_EINVAL;
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

extern "C" int sodt_gather_sum_rows(const void* d, int ldd, void* dsrc, int lds_, int B, int Hs, int Ws, int shr, int C,
                                    int accumulate, int dtype, sodt_stream_t st) {
  const int kpl = dtype == SODT_BF16 ? 8 : 4;
  if (!d || !dsrc || B <= 0 || Hs <= 0 || Ws <= 0 || shr < 0 || C <= 0 || (C % kpl) || (lds_ % kpl) || (ldd % kpl)) return SODT_EINVAL;
  const unsigned gr = nblocks((long)B * Hs * Ws * (C / kpl));
  if (dtype == SODT_BF16) hipLaunchKernelGGL(gather_sum_rows_kernel<bf16>, dim3(gr), dim3(256), 0, (hipStream_t)st, (const bf16*)d, ldd, (bf16*)dsrc, lds_, B, Hs, Ws, shr, C, accumulate);
  else if (dtype == SODT_F32) hipLaunchKernelGGL(gather_sum_rows_kernel<float>, dim3(gr), dim3(256), 0, (hipStream_t)st, (const float*)d, ldd, (float*)dsrc, lds_, B, Hs, Ws, shr, C, accumulate);
  else return SODT_EINVAL;
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

extern "C" int sodt_detect_unpermute(const float* dpred, void* dz, int ldz, int B, int HW, int na, int no,
                                     int dtype, sodt_stream_t st) {
  if (!dpred || !dz || B <= 0 || HW <= 0 || na * no > ldz) return SODT_EINVAL;
  const unsigned gr = nblocks((long)B * HW * ldz);
  if (dtype == SODT_BF16) hipLaunchKernelGGL(detect_unpermute_kernel<bf16>, dim3(gr), dim3(256), 0, (hipStream_t)st, dpred, (bf16*)dz, ldz, B, HW, na, no);
  else if (dtype == SODT_F32) hipLaunchKernelGGL(detect_unpermute_kernel<float>, dim3(gr), dim3(256), 0, (hipStream_t)st, dpred, (float*)dz, ldz, B, HW, na, no);
  else return SODT_EINVAL;
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

extern "C" int sodt_detect_decode(const float* raw, const float* anchor_grid, float* z, int B, int na, int ny, int nx,
                                  int no, float stride, sodt_stream_t st) {
  if (!raw || !anchor_grid || !z || B <= 0 || na <= 0 || ny <= 0 || nx <= 0 || no < 5) return SODT_EINVAL;
  hipLaunchKernelGGL(detect_decode_kernel, dim3(nblocks((long)B * na * ny * nx * no)), dim3(256), 0, (hipStream_t)st,
                     raw, anchor_grid, z, B, na, ny, nx, no, stride);
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

extern "C" int sodt_prep_weights(const sodt_prep_desc* table_dev, int n, int max_elems, int dtype, sodt_stream_t st) {
  if (!table_dev || n <= 0 || max_elems <= 0) return SODT_EINVAL;
  unsigned bx = nblocks(max_elems / 4, 512);   // the largest weights (768 x 3072) are 2304 transpose tiles: 64 blocks walked 36 tiles each, one latency chain per tile
  if (dtype == SODT_BF16) hipLaunchKernelGGL(prep_kernel<bf16>, dim3(bx, n), dim3(256), 0, (hipStream_t)st, table_dev, n);
  else if (dtype == SODT_F32) hipLaunchKernelGGL(prep_kernel<float>, dim3(bx, n), dim3(256), 0, (hipStream_t)st, table_dev, n);
  else return SODT_EINVAL;
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

extern "C" int sodt_transpose_f32(const float* src, float* dst, int rows, int cols, int accumulate, sodt_stream_t st) {
  if (!src || !dst || rows <= 0 || cols <= 0) return SODT_EINVAL;
  hipLaunchKernelGGL(transpose_f32_kernel, dim3(nblocks((long)rows * cols)), dim3(256), 0, (hipStream_t)st, src, dst, rows, cols, accumulate);
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

extern "C" int sodt_cast(const void* src, void* dst, long n, int src_dtype, int dst_dtype, sodt_stream_t st) {
  if (!src || !dst || n <= 0) return SODT_EINVAL;
  const unsigned gr = nblocks(n);
  if (src_dtype == SODT_F32 && dst_dtype == SODT_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16>), dim3(gr), dim3(256), 0, (hipStream_t)st, (const float*)src, (bf16*)dst, n);
  else if (src_dtype == SODT_BF16 && dst_dtype == SODT_F32) hipLaunchKernelGGL((cast_kernel<bf16, float>), dim3(gr), dim3(256), 0, (hipStream_t)st, (const bf16*)src, (float*)dst, n);
  else if (src_dtype == SODT_F32 && dst_dtype == SODT_F32) hipLaunchKernelGGL((cast_kernel<float, float>), dim3(gr), dim3(256), 0, (hipStream_t)st, (const float*)src, (float*)dst, n);
  else return SODT_EINVAL;
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

extern "C" int sodt_batch_sum(const void* d, float* out, int B, long RC, int dtype, sodt_stream_t st) {
  if (!d || !out || B <= 0 || RC <= 0) return SODT_EINVAL;
  const unsigned gr = nblocks(RC / 4);
  if (dtype == SODT_BF16) hipLaunchKernelGGL(batch_sum_kernel<bf16>, dim3(gr), dim3(256), 0, (hipStream_t)st, (const bf16*)d, out, B, RC);
  else if (dtype == SODT_F32) hipLaunchKernelGGL(batch_sum_kernel<float>, dim3(gr), dim3(256), 0, (hipStream_t)st, (const float*)d, out, B, RC);
  else return SODT_EINVAL;
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

extern "C" int sodt_memset_zero(void* p, long bytes, sodt_stream_t st) {
  if (!p || bytes <= 0) return SODT_EINVAL;
  return hipMemsetAsync(p, 0, (size_t)bytes, (hipStream_t)st) == hipSuccess ? SODT_OK : SODT_EINVAL;
}

extern "C" const char* sodt_version(void) { return "sodt_hip 0.1 (gfx950)"; }
