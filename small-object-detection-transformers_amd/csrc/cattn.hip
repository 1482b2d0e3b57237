// General cross-channel attention of the front end: CAttentionBlock / CAttention with window_size >= 1 and an
// optional cyclic shift (backbone_vit.py:469-561, :589-616).  The shipped model hard-codes window_size = 1, which
// sodt_frontend_fwd/bwd fuse into one kernel; this file is the general form the code expresses:
//
//   e_p = Conv2d(1->48, k4, s4)(plane p)                 sodt_patch_embed4_fwd        (R: padding 1)
//   for pairs (q, kv) = (R,G), (G,B), (B,IR), (IR,G):
//     per window (ws x ws tokens, after roll(-shift)), per head (12 x 4 dims, NO projections):
//       attn = softmax((q k^T + mask) / sqrt(4)),  mask = 0 / -100 added BEFORE the scale (:601-608)
//       o = attn v ;  x_q = LayerNorm48(e_q + o)           sodt_cross_attn_ln_fwd / _bwd
//
// head_dim = 4 is far below any MFMA shape, so this is VALU work: one thread per (token, pair), looping over
// the <= 64 keys of its window and the 12 heads.  e is kept in f32 (workspace of B*t*t*192 floats).
#include "common.h"
#include "../../include/sodt_hip.h"

namespace {

constexpr int CE = 48, NH = 12, HD4 = 4;

struct CaGeo { int B, S, t, ws, shift; long ir_bstride; };

__device__ __forceinline__ void patch16(const float* __restrict__ rgb, const float* __restrict__ ir, const CaGeo& g, int p,
                                        int b, int y, int x, float* pt) {
  const int S = g.S;
  const float* base = (p == 3) ? (ir + (long)b * g.ir_bstride) : (rgb + ((long)b * 3 + p) * S * S);
  const int o = (p == 0) ? -1 : 0;      // channel_embed_r has padding (1,1): backbone_vit.py:69-74,:751
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int yy = 4 * y + o + i, xx = 4 * x + o + j;
      pt[i * 4 + j] = (yy >= 0 && xx >= 0 && yy < S && xx < S) ? base[(long)yy * S + xx] : 0.f;
    }
}

__global__ __launch_bounds__(256) void patch_embed4_fwd_kernel(const float* __restrict__ rgb, const float* __restrict__ ir,
                                                              const float* __restrict__ w, const float* __restrict__ bias,
                                                              float* __restrict__ e, const CaGeo g, long ntok) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;    // one thread per (token, plane)
  if (i >= ntok * 4) return;
  const long tok = i >> 2; const int p = (int)(i & 3);
  const int tt = g.t * g.t;
  const int b = (int)(tok / tt); const int rem = (int)(tok - (long)b * tt);
  const int y = rem / g.t, x = rem - y * g.t;
  float pt[16];
  patch16(rgb, ir, g, p, b, y, x, pt);
  const float* wp = w + p * CE * 16;
  for (int j = 0; j < CE; ++j) {
    float a = bias[p * CE + j];
#pragma unroll
    for (int k = 0; k < 16; ++k) a = fmaf(wp[j * 16 + k], pt[k], a);
    e[tok * 192 + p * CE + j] = a;
  }
}

__global__ __launch_bounds__(256) void patch_embed4_bwd_kernel(const float* __restrict__ rgb, const float* __restrict__ ir,
                                                              const float* __restrict__ de, float* __restrict__ dw,
                                                              float* __restrict__ db, const CaGeo g, long ntok) {
  // one thread per (plane, out channel j, tap k) accumulates over a strided token range; 3072 + 192 outputs
  const int oid = blockIdx.x * 256 + threadIdx.x;       // 0 .. 4*48*17-1  (tap 16 = bias)
  if (oid >= 4 * CE * 17) return;
  const int p = oid / (CE * 17), r = oid - p * CE * 17, j = r / 17, k = r - j * 17;
  const int tt = g.t * g.t;
  float acc = 0.f;
  for (long tok = blockIdx.y; tok < ntok; tok += gridDim.y) {
    const float d = de[tok * 192 + p * CE + j];
    if (k == 16) { acc += d; continue; }
    const int b = (int)(tok / tt); const int rem = (int)(tok - (long)b * tt);
    const int y = rem / g.t, x = rem - y * g.t;
    const int o = (p == 0) ? -1 : 0;
    const int yy = 4 * y + o + (k >> 2), xx = 4 * x + o + (k & 3);
    const float* base = (p == 3) ? (ir + (long)b * g.ir_bstride) : (rgb + ((long)b * 3 + p) * g.S * g.S);
    const float v = (yy >= 0 && xx >= 0 && yy < g.S && xx < g.S) ? base[(long)yy * g.S + xx] : 0.f;
    acc = fmaf(d, v, acc);
  }
  if (k == 16) atomicAdd(db + p * CE + j, acc);
  else atomicAdd(dw + (p * CE + j) * 16 + k, acc);
}

__device__ __forceinline__ int pair_kv(int q) { return q == 0 ? 1 : (q == 1 ? 2 : (q == 2 ? 3 : 1)); }   // :508-521

// window geometry of token (b, y, x) under roll(-shift): window origin in shifted coordinates + mask region id
__device__ __forceinline__ void ca_window(const CaGeo& g, int y, int x, int& ys0, int& xs0, int& rid) {
  int ys = y - g.shift, xs = x - g.shift;               // shifted[ys] = x[(ys + shift) % H]
  if (ys < 0) ys += g.t;
  if (xs < 0) xs += g.t;
  ys0 = ys / g.ws * g.ws; xs0 = xs / g.ws * g.ws;
  rid = 0;
  if (g.shift > 0) {
    const int ry = ys < g.t - g.ws ? 0 : (ys < g.t - g.shift ? 1 : 2);
    const int rx = xs < g.t - g.ws ? 0 : (xs < g.t - g.shift ? 1 : 2);
    rid = ry * 3 + rx;
  }
}
__device__ __forceinline__ void ca_key(const CaGeo& g, int ys0, int xs0, int n, int& ky, int& kx, int& krid) {
  const int iy = n / g.ws, ix = n - iy * g.ws;
  const int ys = ys0 + iy, xs = xs0 + ix;
  ky = ys + g.shift; if (ky >= g.t) ky -= g.t;
  kx = xs + g.shift; if (kx >= g.t) kx -= g.t;
  krid = 0;
  if (g.shift > 0) {
    const int ry = ys < g.t - g.ws ? 0 : (ys < g.t - g.shift ? 1 : 2);
    const int rx = xs < g.t - g.ws ? 0 : (xs < g.t - g.shift ? 1 : 2);
    krid = ry * 3 + rx;
  }
}

template <typename T>
__global__ __launch_bounds__(128) void cross_attn_ln_fwd_kernel(const float* __restrict__ e, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, T* __restrict__ out,
                                                               const CaGeo g, long ntok) {
  const long i = (long)blockIdx.x * 128 + threadIdx.x;
  if (i >= ntok * 4) return;
  const long tok = i >> 2; const int p = (int)(i & 3), kvp = pair_kv(p);
  const int tt = g.t * g.t, N = g.ws * g.ws;
  const int b = (int)(tok / tt); const int rem = (int)(tok - (long)b * tt);
  const int y = rem / g.t, x = rem - y * g.t;
  int ys0, xs0, qrid;
  ca_window(g, y, x, ys0, xs0, qrid);
  const float* qrow = e + tok * 192 + p * CE;
  float s[CE];
#pragma unroll 1
  for (int h = 0; h < NH; ++h) {
    const float4 q = *(const float4*)(qrow + h * HD4);
    float mx = -1e30f, den = 0.f;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int n = 0; n < N; ++n) {       // online softmax over the window's keys
      int ky, kx, krid;
      ca_key(g, ys0, xs0, n, ky, kx, krid);
      const float4 k = *(const float4*)(e + ((long)(b * g.t + ky) * g.t + kx) * 192 + kvp * CE + h * HD4);
      float sc = q.x * k.x + q.y * k.y + q.z * k.z + q.w * k.w;
      if (qrid != krid) sc += -100.0f;
      sc *= 0.5f;                        // / sqrt(4), after the mask
      const float mn = fmaxf(mx, sc), a = __expf(mx - mn), pr = __expf(sc - mn);
      den = den * a + pr;
      o.x = o.x * a + pr * k.x; o.y = o.y * a + pr * k.y; o.z = o.z * a + pr * k.z; o.w = o.w * a + pr * k.w;
      mx = mn;
    }
    const float inv = 1.0f / den;
    s[h * 4] = q.x + o.x * inv; s[h * 4 + 1] = q.y + o.y * inv; s[h * 4 + 2] = q.z + o.z * inv; s[h * 4 + 3] = q.w + o.w * inv;
  }
  float mean = 0.f;
#pragma unroll
  for (int j = 0; j < CE; ++j) mean += s[j];
  mean *= (1.0f / CE);
  float var = 0.f;
#pragma unroll
  for (int j = 0; j < CE; ++j) { const float d = s[j] - mean; var += d * d; }
  const float rstd = rsqrtf(var * (1.0f / CE) + 1e-5f);
#pragma unroll
  for (int j = 0; j < CE; ++j)
    out[tok * 192 + p * CE + j] = from_f<T>((s[j] - mean) * rstd * gamma[p * CE + j] + beta[p * CE + j]);
}

template <typename T>
__global__ __launch_bounds__(128) void cross_attn_ln_bwd_kernel(const float* __restrict__ e, const float* __restrict__ gamma,
                                                               const T* __restrict__ dout, float* __restrict__ de,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               const CaGeo g, long ntok) {
  const long i = (long)blockIdx.x * 128 + threadIdx.x;
  const bool live = i < ntok * 4;
  const long tok = live ? (i >> 2) : 0; const int p = (int)(i & 3), kvp = pair_kv(p);
  const int tt = g.t * g.t, N = g.ws * g.ws;
  const int b = (int)(tok / tt); const int rem = (int)(tok - (long)b * tt);
  const int y = rem / g.t, x = rem - y * g.t;
  int ys0, xs0, qrid;
  ca_window(g, y, x, ys0, xs0, qrid);
  const float* qrow = e + tok * 192 + p * CE;
  // ---- recompute the forward for this (token, pair): attention output per head (needs max / denominator)
  float s[CE], mxh[NH], denh[NH];
#pragma unroll 1
  for (int h = 0; h < NH; ++h) {
    const float4 q = *(const float4*)(qrow + h * HD4);
    float mx = -1e30f, den = 0.f;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int n = 0; n < N; ++n) {
      int ky, kx, krid;
      ca_key(g, ys0, xs0, n, ky, kx, krid);
      const float4 k = *(const float4*)(e + ((long)(b * g.t + ky) * g.t + kx) * 192 + kvp * CE + h * HD4);
      float sc = q.x * k.x + q.y * k.y + q.z * k.z + q.w * k.w;
      if (qrid != krid) sc += -100.0f;
      sc *= 0.5f;
      const float mn = fmaxf(mx, sc), a = __expf(mx - mn), pr = __expf(sc - mn);
      den = den * a + pr;
      o.x = o.x * a + pr * k.x; o.y = o.y * a + pr * k.y; o.z = o.z * a + pr * k.z; o.w = o.w * a + pr * k.w;
      mx = mn;
    }
    const float inv = 1.0f / den;
    mxh[h] = mx; denh[h] = den;
    s[h * 4] = q.x + o.x * inv; s[h * 4 + 1] = q.y + o.y * inv; s[h * 4 + 2] = q.z + o.z * inv; s[h * 4 + 3] = q.w + o.w * inv;
  }
  // ---- LayerNorm backward -> ds (gradient of e_q + o)
  float mean = 0.f;
#pragma unroll
  for (int j = 0; j < CE; ++j) mean += s[j];
  mean *= (1.0f / CE);
  float var = 0.f;
#pragma unroll
  for (int j = 0; j < CE; ++j) { s[j] -= mean; var += s[j] * s[j]; }
  const float rstd = rsqrtf(var * (1.0f / CE) + 1e-5f);
  float c1 = 0.f, c2 = 0.f;
  float dy[CE];
#pragma unroll
  for (int j = 0; j < CE; ++j) {
    dy[j] = live ? to_f(dout[tok * 192 + p * CE + j]) : 0.f;
    s[j] *= rstd;                                  // xhat
    const float gg = dy[j] * gamma[p * CE + j];
    c1 += gg; c2 += gg * s[j];
  }
  c1 *= (1.0f / CE); c2 *= (1.0f / CE);
  // per-pair parameter gradients: reduce over the lanes of this wave that share the pair (lane & 3)
#pragma unroll
  for (int j = 0; j < CE; ++j) {
    float a = dy[j] * s[j], c = dy[j];
    a += __shfl_xor(a, 4); a += __shfl_xor(a, 8); a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
    c += __shfl_xor(c, 4); c += __shfl_xor(c, 8); c += __shfl_xor(c, 16); c += __shfl_xor(c, 32);
    if ((threadIdx.x & 63) < 4) { atomicAdd(dgamma + p * CE + j, a); atomicAdd(dbeta + p * CE + j, c); }
    dy[j] = rstd * (dy[j] * gamma[p * CE + j] - c1 - s[j] * c2);        // ds
  }
  if (!live) return;
  // ---- residual path: d e_q += ds ; attention path per head
#pragma unroll 1
  for (int h = 0; h < NH; ++h) {
    const float4 q = *(const float4*)(qrow + h * HD4);
    const float4 dO = make_float4(dy[h * 4], dy[h * 4 + 1], dy[h * 4 + 2], dy[h * 4 + 3]);
    const float inv = 1.0f / denh[h];
    // delta = sum_j p_j (dO . v_j)
    float delta = 0.f;
    for (int n = 0; n < N; ++n) {
      int ky, kx, krid;
      ca_key(g, ys0, xs0, n, ky, kx, krid);
      const float4 k = *(const float4*)(e + ((long)(b * g.t + ky) * g.t + kx) * 192 + kvp * CE + h * HD4);
      float sc = q.x * k.x + q.y * k.y + q.z * k.z + q.w * k.w;
      if (qrid != krid) sc += -100.0f;
      sc *= 0.5f;
      const float pr = __expf(sc - mxh[h]) * inv;
      delta += pr * (dO.x * k.x + dO.y * k.y + dO.z * k.z + dO.w * k.w);
    }
    float4 dq = dO;                                 // residual e_q + o
    for (int n = 0; n < N; ++n) {
      int ky, kx, krid;
      ca_key(g, ys0, xs0, n, ky, kx, krid);
      const long krow = ((long)(b * g.t + ky) * g.t + kx) * 192 + kvp * CE + h * HD4;
      const float4 k = *(const float4*)(e + krow);
      float sc = q.x * k.x + q.y * k.y + q.z * k.z + q.w * k.w;
      if (qrid != krid) sc += -100.0f;
      sc *= 0.5f;
      const float pr = __expf(sc - mxh[h]) * inv;
      const float dp = dO.x * k.x + dO.y * k.y + dO.z * k.z + dO.w * k.w;
      const float dsc = pr * (dp - delta) * 0.5f;   // through the 1/sqrt(4) scale
      dq.x += dsc * k.x; dq.y += dsc * k.y; dq.z += dsc * k.z; dq.w += dsc * k.w;
      // k = v = e_kv[j]:  d e_kv[j] += p_j dO (value path) + dsc * q (key path)
      atomicAdd(de + krow, pr * dO.x + dsc * q.x);
      atomicAdd(de + krow + 1, pr * dO.y + dsc * q.y);
      atomicAdd(de + krow + 2, pr * dO.z + dsc * q.z);
      atomicAdd(de + krow + 3, pr * dO.w + dsc * q.w);
    }
    float* dqp = de + tok * 192 + p * CE + h * HD4;
    atomicAdd(dqp, dq.x); atomicAdd(dqp + 1, dq.y); atomicAdd(dqp + 2, dq.z); atomicAdd(dqp + 3, dq.w);
  }
}

bool ca_geo(CaGeo& g, int B, int S, int ws, int shift, long ir_bstride) {
  if (B <= 0 || S <= 0 || (S % 4) || ws < 1 || ws > 8 || shift < 0 || shift >= (ws > 1 ? ws : 1)) return false;
  g.B = B; g.S = S; g.t = S / 4; g.ws = ws; g.shift = shift; g.ir_bstride = ir_bstride;
  return (g.t % ws) == 0;
}

}  // namespace

extern "C" int sodt_patch_embed4_fwd(const float* rgb, const float* ir, long ir_bstride, const float* w, const float* b,
                                     float* e, int B, int S, sodt_stream_t st) {
  CaGeo g;
  if (!rgb || !ir || !w || !b || !e || !ca_geo(g, B, S, 1, 0, ir_bstride)) return SODT_EINVAL;
  const long ntok = (long)B * g.t * g.t;
  hipLaunchKernelGGL(patch_embed4_fwd_kernel, dim3((unsigned)((ntok * 4 + 255) / 256)), dim3(256), 0, (hipStream_t)st, rgb, ir, w, b, e, g, ntok);
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

extern "C" int sodt_patch_embed4_bwd(const float* rgb, const float* ir, long ir_bstride, const float* de, float* dw, float* db,
                                     int B, int S, sodt_stream_t st) {
  CaGeo g;
  if (!rgb || !ir || !de || !dw || !db || !ca_geo(g, B, S, 1, 0, ir_bstride)) return SODT_EINVAL;
  const long ntok = (long)B * g.t * g.t;
  const unsigned gy = (unsigned)(ntok < 512 ? ntok : 512);
  hipLaunchKernelGGL(patch_embed4_bwd_kernel, dim3((4 * CE * 17 + 255) / 256, gy), dim3(256), 0, (hipStream_t)st, rgb, ir, de, dw, db, g, ntok);
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

extern "C" int sodt_cross_attn_ln_fwd(const float* e, const float* gamma, const float* beta, void* out, int B, int S, int ws,
                                      int shift, int dtype, sodt_stream_t st) {
  CaGeo g;
  if (!e || !gamma || !beta || !out || !ca_geo(g, B, S, ws, shift, 0)) return SODT_EINVAL;
  const long ntok = (long)B * g.t * g.t;
  const unsigned gr = (unsigned)((ntok * 4 + 127) / 128);
  if (dtype == SODT_BF16) hipLaunchKernelGGL(cross_attn_ln_fwd_kernel<bf16>, dim3(gr), dim3(128), 0, (hipStream_t)st, e, gamma, beta, (bf16*)out, g, ntok);
  else if (dtype == SODT_F32) hipLaunchKernelGGL(cross_attn_ln_fwd_kernel<float>, dim3(gr), dim3(128), 0, (hipStream_t)st, e, gamma, beta, (float*)out, g, ntok);
  else return SODT_EINVAL;
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

extern "C" int sodt_cross_attn_ln_bwd(const float* e, const float* gamma, const void* dout, float* de, float* dgamma,
                                      float* dbeta, int B, int S, int ws, int shift, int dtype, sodt_stream_t st) {
  CaGeo g;
  if (!e || !gamma || !dout || !de || !dgamma || !dbeta || !ca_geo(g, B, S, ws, shift, 0)) return SODT_EINVAL;
  const long ntok = (long)B * g.t * g.t;
  const unsigned gr = (unsigned)((ntok * 4 + 127) / 128);
  if (dtype == SODT_BF16) hipLaunchKernelGGL(cross_attn_ln_bwd_kernel<bf16>, dim3(gr), dim3(128), 0, (hipStream_t)st, e, gamma, (const bf16*)dout, de, dgamma, dbeta, g, ntok);
  else if (dtype == SODT_F32) hipLaunchKernelGGL(cross_attn_ln_bwd_kernel<float>, dim3(gr), dim3(128), 0, (hipStream_t)st, e, gamma, (const float*)dout, de, dgamma, dbeta, g, ntok);
  else return SODT_EINVAL;
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}
