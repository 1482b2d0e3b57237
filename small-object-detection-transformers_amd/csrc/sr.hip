// Data movement of the super-resolution auxiliary branch (basics/models/deeplabedsr.py:35-73 = Decoder, sr_decoder_noBN_noD.py:6-45,
// + EDSR, edsr.py:55-102) between its convolutions, which run as K-segment GEMMs (csrc/gemm.hip): bilinear x2 resize with
// align_corners=True and its adjoint, PixelShuffle(2) and its adjoint, gradient accumulation, and the (B, C, H, W) f32 <->
// token-major conversion of the branch's output at the model boundary.  All HBM-bound, 16-byte accesses, grid-stride.
#include "common.h"
#include "../../include/sodt_hip.h"

namespace {

inline unsigned nblk(long n) {
  long b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

// source coordinate of output index o (align_corners=True): o * (n_in - 1) / (n_out - 1)
__device__ __forceinline__ void src_coord(int o, int n_in, int n_out, int& i0, int& i1, float& w1) {
  const float f = n_out > 1 ? (float)o * (float)(n_in - 1) / (float)(n_out - 1) : 0.f;
  i0 = (int)f;
  if (i0 > n_in - 1) i0 = n_in - 1;
  i1 = i0 + 1 < n_in ? i0 + 1 : n_in - 1;
  w1 = f - (float)i0;
}

template <typename T>
__global__ __launch_bounds__(256) void bilinear_up2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int ldy, int B, int H, int W, int C) {
  constexpr int KPL = TT<T>::KPL;
  const int CH = C / KPL, Ho = 2 * H, Wo = 2 * W;
  const long total = (long)B * Ho * Wo * CH;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long m = i / CH; const int c0 = (int)(i - m * CH) * KPL;
    const int b = (int)(m / ((long)Ho * Wo)); const int rem = (int)(m - (long)b * Ho * Wo);
    const int oy = rem / Wo, ox = rem - oy * Wo;
    int y0, y1, x0, x1; float wy, wx;
    src_coord(oy, H, Ho, y0, y1, wy);
    src_coord(ox, W, Wo, x0, x1, wx);
    const T* base = x + (long)b * H * W * C + c0;
    float a[KPL], bb[KPL], c[KPL], d[KPL], o[KPL];
    unpack<T>(*(const uint4*)(base + ((long)y0 * W + x0) * C), a);
    unpack<T>(*(const uint4*)(base + ((long)y0 * W + x1) * C), bb);
    unpack<T>(*(const uint4*)(base + ((long)y1 * W + x0) * C), c);
    unpack<T>(*(const uint4*)(base + ((long)y1 * W + x1) * C), d);
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
      const float top = a[j] + wx * (bb[j] - a[j]), bot = c[j] + wx * (d[j] - c[j]);
      o[j] = top + wy * (bot - top);
    }
    *(uint4*)(y + m * ldy + c0) = pack<T>(o);
  }
}

// adjoint: source pixel (iy, ix) collects from the output pixels whose interpolation stencil contains it.  With the x2 +
// align_corners map f(o) = o (n - 1) / (2n - 1) a source index i is the floor of outputs o in [o_lo, o_hi] and the "+ 1"
// neighbour of the outputs whose floor is i - 1: at most four outputs per axis (<= 16 per pixel), found by scanning
// o in [2i - 2, 2i + 2].
template <typename T>
__global__ __launch_bounds__(256) void bilinear_up2_bwd_kernel(const T* __restrict__ dy, int lddy, T* __restrict__ dx, const T* __restrict__ relu_out,
                                                              int B, int H, int W, int C) {
  constexpr int KPL = TT<T>::KPL;
  const int CH = C / KPL, Ho = 2 * H, Wo = 2 * W;
  const long total = (long)B * H * W * CH;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long m = i / CH; const int c0 = (int)(i - m * CH) * KPL;
    const int b = (int)(m / ((long)H * W)); const int rem = (int)(m - (long)b * H * W);
    const int iy = rem / W, ix = rem - iy * W;
    float acc[KPL];
#pragma unroll
    for (int j = 0; j < KPL; ++j) acc[j] = 0.f;
    const T* base = dy + (long)b * Ho * Wo * lddy + c0;
    for (int oy = 2 * iy - 2; oy <= 2 * iy + 2; ++oy) {
      if (oy < 0 || oy >= Ho) continue;
      int y0, y1; float wy;
      src_coord(oy, H, Ho, y0, y1, wy);
      const float cy = (y0 == iy ? 1.f - wy : 0.f) + (y1 == iy ? wy : 0.f);
      if (cy == 0.f) continue;
      for (int ox = 2 * ix - 2; ox <= 2 * ix + 2; ++ox) {
        if (ox < 0 || ox >= Wo) continue;
        int x0, x1; float wx;
        src_coord(ox, W, Wo, x0, x1, wx);
        const float cx = (x0 == ix ? 1.f - wx : 0.f) + (x1 == ix ? wx : 0.f);
        if (cx == 0.f) continue;
        float v[KPL];
        unpack<T>(*(const uint4*)(base + ((long)oy * Wo + ox) * lddy), v);
        const float wgt = cy * cx;
#pragma unroll
        for (int j = 0; j < KPL; ++j) acc[j] = fmaf(wgt, v[j], acc[j]);
      }
    }
    if (relu_out) {                       // x was the output of a ReLU: its gradient passes where that output was positive
      float r[KPL];
      unpack<T>(*(const uint4*)(relu_out + m * C + c0), r);
#pragma unroll
      for (int j = 0; j < KPL; ++j) acc[j] = r[j] > 0.f ? acc[j] : 0.f;
    }
    *(uint4*)(dx + m * C + c0) = pack<T>(acc);
  }
}

// PixelShuffle(2): out[b][2y + i][2x + j][c] = in[b][y][x][4c + 2i + j].  One thread = one 16-byte chunk of the SMALL-grid tensor
// (KPL channels 4c + s = KPL / 4 output channels for each of the four sub-pixels): the wide side moves as 16-byte vectors, the
// shuffled side as KPL / 4-element pieces that are contiguous ACROSS the threads of a wave (thread q + 1 holds the next output
// channels of the same four pixels), so both sides are whole cache lines per wave.  (The first cut gathered 2-byte elements 8 bytes
// apart on the small-grid side: 20 ms of the 252 ms SR step at B=4 @ 2048^2, profiles/r04_sr_step_kernel_stats.md.)
template <typename T>
__global__ __launch_bounds__(256) void pixel_shuffle2_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int H, int W, int C,
                                                            int inverse) {
  constexpr int KPL = TT<T>::KPL, OC = KPL / 4;          // output channels per thread and sub-pixel: 2 (bf16) or 1 (f32)
  const int CH = 4 * C / KPL, Wo = 2 * W;
  const long total = (long)B * H * W * CH;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long m = i / CH; const int q = (int)(i - m * CH);
    const int b = (int)(m / ((long)H * W)); const int rem = (int)(m - (long)b * H * W);
    const int y = rem / W, x = rem - y * W;
    const long o00 = ((long)(b * 2 * H + 2 * y) * Wo + 2 * x) * C + q * OC;       // in the (B, 2H, 2W, C) tensor
    T v[KPL];
    if (!inverse) {
      *(uint4*)v = *(const uint4*)(in + m * 4 * C + q * KPL);
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        T* dst = out + o00 + ((long)(s2 >> 1) * Wo + (s2 & 1)) * C;
#pragma unroll
        for (int cc = 0; cc < OC; ++cc) dst[cc] = v[4 * cc + s2];
      }
    } else {                                             // `in` is the large side here
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        const T* src = in + o00 + ((long)(s2 >> 1) * Wo + (s2 & 1)) * C;
#pragma unroll
        for (int cc = 0; cc < OC; ++cc) v[4 * cc + s2] = src[cc];
      }
      *(uint4*)(out + m * 4 * C + q * KPL) = *(const uint4*)v;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void add_rows_kernel(T* __restrict__ dst, int ldd, int dcol, const T* __restrict__ src, int lds_, int scol,
                                                      long M, int C) {
  constexpr int KPL = TT<T>::KPL;
  const int CH = C / KPL;
  const long total = M * CH;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long m = i / CH; const int c0 = (int)(i - m * CH) * KPL;
    float a[KPL], b[KPL];
    unpack<T>(*(const uint4*)(dst + m * ldd + dcol + c0), a);
    unpack<T>(*(const uint4*)(src + m * lds_ + scol + c0), b);
#pragma unroll
    for (int j = 0; j < KPL; ++j) a[j] += b[j];
    *(uint4*)(dst + m * ldd + dcol + c0) = pack<T>(a);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void nchw_from_rows_kernel(const T* __restrict__ rows, int ld, float* __restrict__ y, int B, int C, int HW) {
  const long total = (long)B * C * HW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long p = i % HW; const long bc = i / HW;
    const int c = (int)(bc % C); const long b = bc / C;
    y[i] = to_f(rows[(b * HW + p) * ld + c]);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void rows_from_nchw_kernel(const float* __restrict__ y, T* __restrict__ rows, int ld, int B, int C, int HW) {
  const long total = (long)B * HW * ld;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % ld); const long m = i / ld;
    const long b = m / HW, p = m - b * HW;
    rows[i] = from_f<T>(c < C ? y[(b * C + c) * HW + p] : 0.f);
  }
}

__global__ void add_small_f32_kernel(float* dst, const float* src, int n) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] += src[threadIdx.x];
}

}  // namespace

// the kernels move 16-byte chunks (uint4): pointers - including column-slice pointers formed by the callers - must be 16-byte aligned
static bool al16(const void* a, const void* b = nullptr, const void* c = nullptr) {
  return ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c)) & 15) == 0;
}
static bool sr_ok(const void* a, const void* b, int B, int H, int W, int C, int dtype) {
  const int kpl = dtype == SODT_BF16 ? 8 : 4;
  return a && b && al16(a, b) && B > 0 && H > 0 && W > 0 && C > 0 && (C % kpl) == 0 && (long)B * H * W * 4 < (1L << 31);
}

extern "C" int sodt_bilinear_up2_fwd(const void* x, void* y, int ldy, int B, int H, int W, int C, int dtype, sodt_stream_t st) {
  if (!sr_ok(x, y, B, H, W, C, dtype) || ldy < C || (ldy % (dtype == SODT_BF16 ? 8 : 4))) return SODT_EINVAL;
  const long n = (long)B * 4 * H * W * (C / (dtype == SODT_BF16 ? 8 : 4));
  if (dtype == SODT_BF16) hipLaunchKernelGGL(bilinear_up2_fwd_kernel<bf16>, dim3(nblk(n)), dim3(256), 0, (hipStream_t)st, (const bf16*)x, (bf16*)y, ldy, B, H, W, C);
  else if (dtype == SODT_F32) hipLaunchKernelGGL(bilinear_up2_fwd_kernel<float>, dim3(nblk(n)), dim3(256), 0, (hipStream_t)st, (const float*)x, (float*)y, ldy, B, H, W, C);
  else return SODT_EINVAL;
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

extern "C" int sodt_bilinear_up2_bwd(const void* dy, int lddy, void* dx, const void* relu_out, int B, int H, int W, int C, int dtype,
                                     sodt_stream_t st) {
  if (!sr_ok(dy, dx, B, H, W, C, dtype) || !al16(relu_out) || lddy < C || (lddy % (dtype == SODT_BF16 ? 8 : 4))) return SODT_EINVAL;
  const long n = (long)B * H * W * (C / (dtype == SODT_BF16 ? 8 : 4));
  if (dtype == SODT_BF16) hipLaunchKernelGGL(bilinear_up2_bwd_kernel<bf16>, dim3(nblk(n)), dim3(256), 0, (hipStream_t)st, (const bf16*)dy, lddy, (bf16*)dx, (const bf16*)relu_out, B, H, W, C);
  else if (dtype == SODT_F32) hipLaunchKernelGGL(bilinear_up2_bwd_kernel<float>, dim3(nblk(n)), dim3(256), 0, (hipStream_t)st, (const float*)dy, lddy, (float*)dx, (const float*)relu_out, B, H, W, C);
  else return SODT_EINVAL;
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

extern "C" int sodt_pixel_shuffle2(const void* in, void* out, int B, int H, int W, int C, int inverse, int dtype, sodt_stream_t st) {
  if (!sr_ok(in, out, B, H, W, C, dtype)) return SODT_EINVAL;
  const long n = (long)B * H * W * (4 * C / (dtype == SODT_BF16 ? 8 : 4));
  if (dtype == SODT_BF16) hipLaunchKernelGGL(pixel_shuffle2_kernel<bf16>, dim3(nblk(n)), dim3(256), 0, (hipStream_t)st, (const bf16*)in, (bf16*)out, B, H, W, C, inverse);
  else if (dtype == SODT_F32) hipLaunchKernelGGL(pixel_shuffle2_kernel<float>, dim3(nblk(n)), dim3(256), 0, (hipStream_t)st, (const float*)in, (float*)out, B, H, W, C, inverse);
  else return SODT_EINVAL;
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

extern "C" int sodt_add_rows(void* dst, int ldd, int dcol, const void* src, int lds_, int scol, long M, int C, int dtype, sodt_stream_t st) {
  const int kpl = dtype == SODT_BF16 ? 8 : 4;
  if (dst && src && dtype == SODT_F32 && M == 1 && C > 0 && C < 64 && (C % kpl)) {
    // one short f32 row below the 16-byte granule (the bias gradient of a 3-channel convolution): element-wise
    hipLaunchKernelGGL(add_small_f32_kernel, dim3(1), dim3(64), 0, (hipStream_t)st, (float*)dst + dcol, (const float*)src + scol, C);
    return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
  }
  if (!dst || !src || !al16(dst, src) || M <= 0 || C <= 0 || (C % kpl) || (ldd % kpl) || (lds_ % kpl) || (dcol % kpl) || (scol % kpl)) return SODT_EINVAL;
  const long n = M * (C / kpl);
  if (dtype == SODT_BF16) hipLaunchKernelGGL(add_rows_kernel<bf16>, dim3(nblk(n)), dim3(256), 0, (hipStream_t)st, (bf16*)dst, ldd, dcol, (const bf16*)src, lds_, scol, M, C);
  else if (dtype == SODT_F32) hipLaunchKernelGGL(add_rows_kernel<float>, dim3(nblk(n)), dim3(256), 0, (hipStream_t)st, (float*)dst, ldd, dcol, (const float*)src, lds_, scol, M, C);
  else return SODT_EINVAL;
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

extern "C" int sodt_nchw_f32_from_rows(const void* rows, int ld, float* y, int B, int C, int H, int W, int dtype, sodt_stream_t st) {
  if (!rows || !y || B <= 0 || C <= 0 || C > ld || H <= 0 || W <= 0) return SODT_EINVAL;
  const long n = (long)B * C * H * W;
  if (dtype == SODT_BF16) hipLaunchKernelGGL(nchw_from_rows_kernel<bf16>, dim3(nblk(n)), dim3(256), 0, (hipStream_t)st, (const bf16*)rows, ld, y, B, C, H * W);
  else if (dtype == SODT_F32) hipLaunchKernelGGL(nchw_from_rows_kernel<float>, dim3(nblk(n)), dim3(256), 0, (hipStream_t)st, (const float*)rows, ld, y, B, C, H * W);
  else return SODT_EINVAL;
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

extern "C" int sodt_rows_from_nchw_f32(const float* y, void* rows, int ld, int B, int C, int H, int W, int dtype, sodt_stream_t st) {
  if (!rows || !y || B <= 0 || C <= 0 || C > ld || H <= 0 || W <= 0) return SODT_EINVAL;
  const long n = (long)B * H * W * ld;
  if (dtype == SODT_BF16) hipLaunchKernelGGL(rows_from_nchw_kernel<bf16>, dim3(nblk(n)), dim3(256), 0, (hipStream_t)st, y, (bf16*)rows, ld, B, C, H * W);
  else if (dtype == SODT_F32) hipLaunchKernelGGL(rows_from_nchw_kernel<float>, dim3(nblk(n)), dim3(256), 0, (hipStream_t)st, y, (float*)rows, ld, B, C, H * W);
  else return SODT_EINVAL;
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}
