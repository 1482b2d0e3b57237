// Pieces shared by the GEMM translation units (gemm.hip, gemm3.hip): the spatial row map of a K-segment
// and the fused epilogue of one 16-byte output chunk.
#pragma once
#include "common.h"
#include "../../include/sodt_hip.h"

namespace {

struct RowGeo { int b, y, x; bool ok; };

__device__ __forceinline__ long seg_src_row(const sodt_seg& s, const RowGeo& r, int spatial, long m) {
  if (!r.ok) return -1;
  if (!spatial) return m;
  const int yy = r.y * s.mul + s.dy, xx = r.x * s.mul + s.dx;
  if (yy < 0 || xx < 0) return -1;
  const int yi = yy >> s.shr, xi = xx >> s.shr;
  if (yi >= s.Hi || xi >= s.Wi) return -1;
  return ((long)r.b * s.Hi + yi) * s.Wi + xi;
}

// fused epilogue of one 16-byte output chunk: v[KPL] = accumulators of row m, columns n .. n+KPL-1
// CF >= 0: the epilogue flag set is a compile-time constant (the handful of combinations the model uses get
// branch-free instantiations); CF < 0: flags are read at run time (generic fallback).
template <typename T, int CF = -1>
__device__ __forceinline__ void epi_chunk(const sodt_gemm_args& g, const int rtflags, const long m, const int n,
                                          float (&v)[TT<T>::KPL], const int hw) {
  constexpr int KPL = TT<T>::KPL;
  const int flags = CF >= 0 ? CF : rtflags;
  const bool out32 = (flags & SODT_EPI_OUT_F32) != 0;
  const bool full = (n + KPL <= g.N);
  if (flags & SODT_EPI_BIAS) {
    if (full) {
#pragma unroll
      for (int j = 0; j < KPL; j += 4) {
        const float4 bb = *(const float4*)(g.bias + n + j);
        v[j] += bb.x; v[j + 1] += bb.y; v[j + 2] += bb.z; v[j + 3] += bb.w;
      }
    } else {
      for (int j = 0; j < KPL; ++j) if (n + j < g.N) v[j] += g.bias[n + j];
    }
  }
#ifndef SODT_EXP_NO_ACT          // (A/B builds of tools/exp/ab_build.sh only: what the activation's VALU work costs)
  if (flags & SODT_EPI_GELU) {
#pragma unroll
    for (int j = 0; j < KPL; ++j) v[j] = gelu_t<T>(v[j]);
  }
#endif
  if (flags & SODT_EPI_RELU) {
#pragma unroll
    for (int j = 0; j < KPL; ++j) v[j] = fmaxf(v[j], 0.f);
  }
  if (flags & SODT_EPI_AFFINE_SILU) {
    float sc[KPL], sh[KPL];
    if (full) {
#pragma unroll
      for (int j = 0; j < KPL; j += 4) {
        const float4 a4 = *(const float4*)(g.scale + n + j);
        const float4 b4 = *(const float4*)(g.shift + n + j);
        sc[j] = a4.x; sc[j + 1] = a4.y; sc[j + 2] = a4.z; sc[j + 3] = a4.w;
        sh[j] = b4.x; sh[j + 1] = b4.y; sh[j + 2] = b4.z; sh[j + 3] = b4.w;
      }
    } else {
      for (int j = 0; j < KPL; ++j) { sc[j] = (n + j < g.N) ? g.scale[n + j] : 0.f; sh[j] = (n + j < g.N) ? g.shift[n + j] : 0.f; }
    }
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
      const float a = v[j] * sc[j] + sh[j];
      v[j] = a * sigmoid_f(a);
    }
  }
  if (flags & SODT_EPI_DGELU) {
    float x[KPL];
    const T* ap = (const T*)g.aux + m * g.ldaux + n;
    if (full) { unpack<T>(*(const uint4*)ap, x); }
    else { for (int j = 0; j < KPL; ++j) x[j] = (n + j < g.N) ? to_f(ap[j]) : 0.f; }
#pragma unroll
    for (int j = 0; j < KPL; ++j) v[j] *= dgelu_t<T>(x[j]);
  }
  if (flags & SODT_EPI_DRELU) {            // gradient through a ReLU whose OUTPUT is aux: passes where the output was positive
    float x[KPL];
    const T* ap = (const T*)g.aux + m * g.ldaux + n;
    if (full) { unpack<T>(*(const uint4*)ap, x); }
    else { for (int j = 0; j < KPL; ++j) x[j] = (n + j < g.N) ? to_f(ap[j]) : 0.f; }
#pragma unroll
    for (int j = 0; j < KPL; ++j) v[j] = x[j] > 0.f ? v[j] : 0.f;
  }
  if (flags & SODT_EPI_RESID) {
    const long rr = g.rmod > 0 ? (m % g.rmod) : m;
    float x[KPL];
    const T* rp = (const T*)g.R + rr * g.ldr + n;
    if (full) { unpack<T>(*(const uint4*)rp, x); }
    else { for (int j = 0; j < KPL; ++j) x[j] = (n + j < g.N) ? to_f(rp[j]) : 0.f; }
#pragma unroll
    for (int j = 0; j < KPL; ++j) v[j] += x[j];
  }
  long orow = m;
  if (CF < 0 && g.oscatter) {
    const int b = (int)(m / hw), rem = (int)(m - (long)b * hw);
    const int y = rem / g.a.Wo, x = rem - y * g.a.Wo;
    orow = ((long)b * g.OH + y * g.omul + g.ody) * g.OW + x * g.omul + g.odx;
  }
  if (out32) {
    float* cp = (float*)g.C + orow * g.ldc + n;
    for (int j = 0; j < KPL; ++j) if (full || n + j < g.N) cp[j] = v[j];
  } else {
    T* cp = (T*)g.C + orow * g.ldc + n;
#ifdef SODT_EXP_NO_STORE         // (A/B builds only: what the output stores cost - one store per 256 rows keeps the work alive)
    if (full && (m & 255) == 0) *(uint4*)cp = pack<T>(v);
    else if (full) asm volatile("" :: "v"(pack<T>(v).x));
#else
    if (full) *(uint4*)cp = pack<T>(v);
#endif
    else for (int j = 0; j < KPL; ++j) if (n + j < g.N) cp[j] = from_f<T>(v[j]);
    if (flags & SODT_EPI_GELU_DUAL) {
      float a[KPL];
#pragma unroll
      for (int j = 0; j < KPL; ++j) a[j] = gelu_t<T>(v[j]);
      T* c2 = (T*)g.C2 + orow * g.ldc2 + n;
      if (full) *(uint4*)c2 = pack<T>(a);
      else for (int j = 0; j < KPL; ++j) if (n + j < g.N) c2[j] = from_f<T>(a[j]);
    }
  }
}

}  // namespace
