// Input pre-processing of the training / evaluation loop on the device:
//
//     image = imgs.to(device).float() / 255.0                                                    Train.py:364-365, test.py:124-129
//     imgs  = F.interpolate(image, size=[i // down_factor ...], mode='bilinear', align_corners=True)   Train.py:371-374
//
// for the RGB and the IR batch in ONE launch: uint8 planes in, f32 planes out (the layout the front-end kernel reads).
// Bilinear with align_corners=True: src = dst * (in - 1) / (out - 1), i0 = floor(src), l1 = src - i0, i1 = min(i0 + 1, in - 1),
// value = (1-ly)((1-lx) p00 + lx p01) + ly((1-lx) p10 + lx p11).  ATen evaluates src in f32 (scale = (in-1)/(out-1) rounded,
// then a product that its AVX2 / AVX-512 builds may or may not contract with the subtraction): its own l1 moves by up to an
// ulp of the COORDINATE between machines (6e-5 at 1024 pixels).  Here i0 and l1 come from the exact integer quotient and
// remainder of dst * (in - 1) by (out - 1) - one rounding, in the final division - so the result is the correctly rounded
// evaluation of the same formula: within 1e-6 of the float64 evaluation, and inside ATen-f32's own coordinate noise.
// out == in is the identity: plain u8 -> f32 / 255.
//
// HBM-bound byte work (6 MB in, 6.3 MB out per 1024^2 image pair at down_factor 2): one thread produces four consecutive
// output pixels of one plane (one 16-byte store); its 2 x (up to 10) source bytes are two rows of one cache line region,
// read through L1 by neighbouring lanes.
#include "common.h"
#include "../../include/sodt_hip.h"

namespace {

struct PreArgs {
  const unsigned char* src[2]; float* dst[2];
  int planes[2];              // B * channels of the RGB / IR tensor
  int Hin, Win, Hout, Wout;
};

__global__ __launch_bounds__(256) void preprocess_u8_kernel(const PreArgs a) {
  const int wq = (a.Wout + 3) >> 2;                       // 4-pixel groups per output row
  const long per_plane = (long)a.Hout * wq;
  const long total = per_plane * (a.planes[0] + a.planes[1]);
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long pl = idx / per_plane;
    const long rem = idx - pl * per_plane;
    const int oy = (int)(rem / wq), ox0 = (int)(rem - (long)oy * wq) * 4;
    const int which = pl >= a.planes[0];
    if (which) pl -= a.planes[0];
    const unsigned char* sp = a.src[which] + pl * (long)a.Hin * a.Win;
    float* dp = a.dst[which] + pl * (long)a.Hout * a.Wout + (long)oy * a.Wout + ox0;
    const int dy = a.Hout > 1 ? a.Hout - 1 : 1, ny = a.Hout > 1 ? oy * (a.Hin - 1) : 0;      // src = ny / dy exactly
    const int y0 = ny / dy;
    const float ly = (float)(ny - y0 * dy) / (float)dy;
    const int y1 = y0 + 1 < a.Hin ? y0 + 1 : a.Hin - 1;
    const unsigned char* r0 = sp + (long)y0 * a.Win;
    const unsigned char* r1 = sp + (long)y1 * a.Win;
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ox = ox0 + i < a.Wout ? ox0 + i : a.Wout - 1;
      const int dx = a.Wout > 1 ? a.Wout - 1 : 1, nx = a.Wout > 1 ? ox * (a.Win - 1) : 0;
      const int x0 = nx / dx;
      const float lx = (float)(nx - x0 * dx) / (float)dx;
      const int x1 = x0 + 1 < a.Win ? x0 + 1 : a.Win - 1;
      const float p00 = (float)r0[x0] / 255.0f, p01 = (float)r0[x1] / 255.0f;
      const float p10 = (float)r1[x0] / 255.0f, p11 = (float)r1[x1] / 255.0f;
      o[i] = (1.f - ly) * ((1.f - lx) * p00 + lx * p01) + ly * ((1.f - lx) * p10 + lx * p11);
    }
    if (ox0 + 3 < a.Wout && (((uintptr_t)dp) & 15) == 0) {
      *(float4*)dp = make_float4(o[0], o[1], o[2], o[3]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (ox0 + i < a.Wout) dp[i] = o[i];
    }
  }
}

}  // namespace

extern "C" int sodt_preprocess_u8(const unsigned char* rgb, const unsigned char* ir, float* out_rgb, float* out_ir, int B,
                                  int c_rgb, int c_ir, int Hin, int Win, int Hout, int Wout, sodt_stream_t st) {
  if (!rgb || !out_rgb || B <= 0 || c_rgb <= 0 || c_ir < 0 || (c_ir > 0 && (!ir || !out_ir))) return SODT_EINVAL;
  if (Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0 || Hout > Hin || Wout > Win) return SODT_EINVAL;
  PreArgs a;
  a.src[0] = rgb; a.src[1] = ir; a.dst[0] = out_rgb; a.dst[1] = out_ir;
  a.planes[0] = B * c_rgb; a.planes[1] = B * c_ir;
  a.Hin = Hin; a.Win = Win; a.Hout = Hout; a.Wout = Wout;
  if ((long)Hin * Hout >= (1L << 31) || (long)Win * Wout >= (1L << 31)) return SODT_EINVAL;
  const long total = (long)Hout * ((Wout + 3) / 4) * (a.planes[0] + a.planes[1]);
  long blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(preprocess_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)st, a);
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}
