// nn.MaxPool2d(kernel 5, stride 1, padding 2) on token-major ("NHWC") activations, forward and backward: the building block
// of SPP (basics/models/common.py:129-140), whose 5 / 9 / 13 pools are this pool applied once, twice and three times
// (max over a (2r+1)^2 window of a max over a 5x5 window = max over the (2(r+2)+1)^2 window; -inf padding composes).
// One lane = one 16-byte channel chunk of one output token: 25 coalesced chunk reads (rows of the map are L2 / L1 resident),
// running max per channel with the window position of the FIRST maximum (PyTorch's scan order, which its backward routes
// the gradient to) kept as one byte per channel.  Backward is a gather: an input position sums the gradients of the <= 25
// outputs whose recorded argmax points at it - no atomics, deterministic.  HBM-bound, small (the SPP runs on the stride-16
// map: B x (S/16)^2 tokens).
#include "common.h"
#include "../../include/sodt_hip.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void maxpool5_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy,
                                                          unsigned char* __restrict__ arg, int B, int H, int W, int C) {
  constexpr int KPL = TT<T>::KPL;
  const int cpr = C / KPL;
  const long total = (long)B * H * W * cpr;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int cc = (int)(i % cpr); long r = i / cpr;
    const int xx = (int)(r % W); r /= W;
    const int yy = (int)(r % H); const int b = (int)(r / H);
    float best[KPL]; int bi[KPL];
#pragma unroll
    for (int j = 0; j < KPL; ++j) { best[j] = -INFINITY; bi[j] = 12; }
    for (int dy = -2; dy <= 2; ++dy) {
      const int sy = yy + dy;
      if (sy < 0 || sy >= H) continue;
      for (int dx = -2; dx <= 2; ++dx) {
        const int sx = xx + dx;
        if (sx < 0 || sx >= W) continue;
        float f[KPL];
        unpack<T>(*(const uint4*)(x + ((long)(b * H + sy) * W + sx) * ldx + cc * KPL), f);
        const int k = (dy + 2) * 5 + dx + 2;
#pragma unroll
        for (int j = 0; j < KPL; ++j)
          if (f[j] > best[j] || (f[j] != f[j] && best[j] == best[j])) { best[j] = f[j]; bi[j] = k; }     // first maximum; NaN propagates
      }
    }
    const long o = (long)(b * H + yy) * W + xx;
    *(uint4*)(y + o * ldy + cc * KPL) = pack<T>(best);
    if (arg) {
#pragma unroll
      for (int j = 0; j < KPL; ++j) arg[o * C + cc * KPL + j] = (unsigned char)bi[j];
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool5_bwd_kernel(const T* __restrict__ dy_, int lddy, const unsigned char* __restrict__ arg,
                                                          T* __restrict__ dx_, int lddx, int accumulate, int B, int H, int W, int C) {
  constexpr int KPL = TT<T>::KPL;
  const int cpr = C / KPL;
  const long total = (long)B * H * W * cpr;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int cc = (int)(i % cpr); long r = i / cpr;
    const int xx = (int)(r % W); r /= W;
    const int yy = (int)(r % H); const int b = (int)(r / H);
    float acc[KPL];
#pragma unroll
    for (int j = 0; j < KPL; ++j) acc[j] = 0.f;
    for (int dy = -2; dy <= 2; ++dy) {                       // output (yy + dy, xx + dx) sees this input at window slot (2 - dy, 2 - dx)
      const int oy = yy + dy;
      if (oy < 0 || oy >= H) continue;
      for (int dx = -2; dx <= 2; ++dx) {
        const int ox = xx + dx;
        if (ox < 0 || ox >= W) continue;
        const long o = (long)(b * H + oy) * W + ox;
        const int want = (2 - dy) * 5 + (2 - dx);
        float g[KPL];
        unpack<T>(*(const uint4*)(dy_ + o * lddy + cc * KPL), g);
        const unsigned char* ap = arg + o * C + cc * KPL;
#pragma unroll
        for (int j = 0; j < KPL; ++j) acc[j] += ap[j] == want ? g[j] : 0.f;
      }
    }
    T* dst = dx_ + ((long)(b * H + yy) * W + xx) * lddx + cc * KPL;
    if (accumulate) {
      float f[KPL];
      unpack<T>(*(const uint4*)dst, f);
#pragma unroll
      for (int j = 0; j < KPL; ++j) acc[j] += f[j];
    }
    *(uint4*)dst = pack<T>(acc);
  }
}

inline unsigned pool_blocks(long work) { long b = (work + 255) / 256; return (unsigned)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }

}  // namespace

extern "C" int sodt_maxpool5_fwd(const void* x, int ldx, void* y, int ldy, unsigned char* argmax, int B, int H, int W, int C,
                                 int dtype, sodt_stream_t st) {
  const int kpl = dtype == SODT_BF16 ? 8 : 4;
  if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % kpl) || (ldx % kpl) || (ldy % kpl)) return SODT_EINVAL;
  const unsigned gr = pool_blocks((long)B * H * W * (C / kpl));
  if (dtype == SODT_BF16) hipLaunchKernelGGL(maxpool5_fwd_kernel<bf16>, dim3(gr), dim3(256), 0, (hipStream_t)st, (const bf16*)x, ldx, (bf16*)y, ldy, argmax, B, H, W, C);
  else if (dtype == SODT_F32) hipLaunchKernelGGL(maxpool5_fwd_kernel<float>, dim3(gr), dim3(256), 0, (hipStream_t)st, (const float*)x, ldx, (float*)y, ldy, argmax, B, H, W, C);
  else return SODT_EINVAL;
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

extern "C" int sodt_maxpool5_bwd(const void* dy, int lddy, const unsigned char* argmax, void* dx, int lddx, int accumulate,
                                 int B, int H, int W, int C, int dtype, sodt_stream_t st) {
  const int kpl = dtype == SODT_BF16 ? 8 : 4;
  if (!dy || !dx || !argmax || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % kpl) || (lddx % kpl) || (lddy % kpl)) return SODT_EINVAL;
  const unsigned gr = pool_blocks((long)B * H * W * (C / kpl));
  if (dtype == SODT_BF16) hipLaunchKernelGGL(maxpool5_bwd_kernel<bf16>, dim3(gr), dim3(256), 0, (hipStream_t)st, (const bf16*)dy, lddy, argmax, (bf16*)dx, lddx, accumulate, B, H, W, C);
  else if (dtype == SODT_F32) hipLaunchKernelGGL(maxpool5_bwd_kernel<float>, dim3(gr), dim3(256), 0, (hipStream_t)st, (const float*)dy, lddy, argmax, (float*)dx, lddx, accumulate, B, H, W, C);
  else return SODT_EINVAL;
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}
