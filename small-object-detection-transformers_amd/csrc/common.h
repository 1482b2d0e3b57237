// Shared device helpers for the gfx950 (CDNA4, wave64) kernels.
//
// Activation dtype T is either float (the 1e-3 parity path, exact-f32 MFMA
// v_mfma_f32_16x16x4_f32) or __bf16 (the throughput path, v_mfma_f32_16x16x32_bf16).
// All MFMA operands are fetched as ONE 16-byte vector per lane whose elements are
// consecutive along the contraction index k:
//     lane l holds  A[row = l & 15][k = KPL*(l >> 4) + j],  j = 0 .. KPL-1
//     lane l holds  B[k = KPL*(l >> 4) + j][col = l & 15]
// with KPL = 8 (bf16) or 4 (f32).  For f32 the four elements are fed to four
// 16x16x4 MFMAs (element j -> MFMA j); because A and B use the same k permutation
// the sum over the 16 k's of one call is exact.  The accumulator layout is the
// dtype-independent 16x16 C/D map: acc[r] <-> row = 4*(l >> 4) + r, col = l & 15.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define SODT_OK 0
#define SODT_EINVAL 1

template <typename T> struct TT;
template <> struct TT<float> {
  static constexpr int KPL = 4;     // elements per 16-byte chunk
  static constexpr int MMA_K = 16;  // contraction depth of one mma16() call
  static constexpr int SZ = 4;
};
template <> struct TT<bf16> {
  static constexpr int KPL = 8;
  static constexpr int MMA_K = 32;
  static constexpr int SZ = 2;
};

__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float v) { return (bf16)v; }  // v_cvt_pk_bf16_f32 (RNE, NaN kept)

// one 16x16 tile update: acc += A(16 x MMA_K) * B(MMA_K x 16)
template <typename T> __device__ __forceinline__ void mma16(f32x4& acc, const uint4& a, const uint4& b);
template <> __device__ __forceinline__ void mma16<bf16>(f32x4& acc, const uint4& a, const uint4& b) {
  union { uint4 u; bf16x8 v; } ua, ub;
  ua.u = a; ub.u = b;
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v, ub.v, acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma16<float>(f32x4& acc, const uint4& a, const uint4& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}

// first update of a tile: C is the inline constant 0 (no v_mov zero-initialisation of the accumulator)
template <typename T> __device__ __forceinline__ f32x4 mma16z(const uint4& a, const uint4& b);
template <> __device__ __forceinline__ f32x4 mma16z<bf16>(const uint4& a, const uint4& b) {
  union { uint4 u; bf16x8 v; } ua, ub;
  ua.u = a; ub.u = b;
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v, ub.v, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 mma16z<float>(const uint4& a, const uint4& b) {
  f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  return acc;
}

// 16-byte chunk <-> KPL floats
template <typename T> __device__ __forceinline__ void unpack(const uint4& u, float* f);
template <> __device__ __forceinline__ void unpack<float>(const uint4& u, float* f) {
  f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y); f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
}
template <> __device__ __forceinline__ void unpack<bf16>(const uint4& u, float* f) {
  f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
  f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
  f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
  f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}
// two floats -> one dword of packed bf16 (RNE): a single v_cvt_pk_bf16_f32 (scalar casts + shift/or cost four VALU ops)
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f32x2_;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
  const f32x2_ f = {lo, hi};
  union { bf16x2_ v; uint32_t u; } r;
  r.v = __builtin_convertvector(f, bf16x2_);
  return r.u;
}
template <typename T> __device__ __forceinline__ uint4 pack(const float* f);
template <> __device__ __forceinline__ uint4 pack<float>(const float* f) {
  return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
}
template <> __device__ __forceinline__ uint4 pack<bf16>(const float* f) {
  return make_uint4(pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7]));
}

// erf(x/sqrt2) by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far inside the 1e-3 logit gate) and the
// Gaussian exp(-x^2/2) it shares with gelu'(x): ~14 VALU ops instead of libm erff's ~50.
__device__ __forceinline__ void erf_gauss(float x, float& erfv, float& gauss) {
  const float ax = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  gauss = __expf(-ax * ax);
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  erfv = copysignf(1.0f - p * t * gauss, x);
}
__device__ __forceinline__ float gelu_f(float x) {
  float e, g;
  erf_gauss(x, e, g);
  return 0.5f * x * (1.0f + e);
}
__device__ __forceinline__ float dgelu_f(float x) {
  float e, g;
  erf_gauss(x, e, g);
  return 0.5f * (1.0f + e) + x * 0.3989422804014327f * g;
}
// bf16 throughput path: erf(x / sqrt 2) as an odd degree-15 polynomial in the clamped argument (near-minimax fit of
// erf(z)/z in z^2 on [0, 2.7]; |err| <= 4.2e-5 inside, 1.4e-4 on the clamped tail - far below bf16's 3.9e-3) - no
// rcp / exp on the quarter-rate transcendental unit, and pure FMA chains that hipcc packs into v_pk_fma_f32.
// The f32 parity path keeps the 1.5e-7 form above.
__device__ __forceinline__ float erf_poly(float x) {
  const float z = fminf(fmaxf(x * 0.70710678118654752f, -2.7f), 2.7f);
  const float u = z * z;
  float p = -8.474542596559331e-07f;
  p = fmaf(p, u, 2.943508661701344e-05f);
  p = fmaf(p, u, -0.00045067412429489195f);
  p = fmaf(p, u, 0.004084492567926645f);
  p = fmaf(p, u, -0.024983685463666916f);
  p = fmaf(p, u, 0.11123709380626678f);
  p = fmaf(p, u, -0.37559017539024353f);
  p = fmaf(p, u, 1.1283488273620605f);
  return p * z;
}
template <typename T> __device__ __forceinline__ float gelu_t(float x);
template <> __device__ __forceinline__ float gelu_t<float>(float x) { return gelu_f(x); }
template <> __device__ __forceinline__ float gelu_t<bf16>(float x) {
  const float h = 0.5f * x;
  return fmaf(h, erf_poly(x), h);
}
template <typename T> __device__ __forceinline__ float dgelu_t(float x);
template <> __device__ __forceinline__ float dgelu_t<float>(float x) { return dgelu_f(x); }
// gelu'(x) - 1/2 = erf(x / sqrt 2) / 2 + x phi(x) is odd: one degree-15 odd minimax polynomial of the argument clamped to
// [-4, 4] (|err| <= 2.8e-4 evaluated in f32; gelu'(+-4) = 0.5 +- 0.4995) - 11 FMA-pipe operations instead of the erf
// polynomial plus an exp2 (the epilogues that multiply by gelu' were VALU-bound: -0.33 ms of 1.7 on the stage-1 launches)
template <> __device__ __forceinline__ float dgelu_t<bf16>(float x) {
  const float z = fminf(fmaxf(x, -4.0f), 4.0f);
  const float u = z * z;
  float p = -1.642114889e-08f;
  p = fmaf(p, u, 1.213881774e-06f);
  p = fmaf(p, u, -3.846123582e-05f);
  p = fmaf(p, u, 6.876639673e-04f);
  p = fmaf(p, u, -7.687550504e-03f);
  p = fmaf(p, u, 5.591514707e-02f);
  p = fmaf(p, u, -2.620302439e-01f);
  p = fmaf(p, u, 7.967218161e-01f);
  return fmaf(p, z, 0.5f);
}
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// reductions inside a 16-lane group (lanes sharing l >> 4): four DPP steps on the VALU (quad_perm xor 1, xor 2,
// row_half_mirror, row_mirror) instead of ds_bpermute shuffles through the LDS pipe; every lane ends with the result
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float group16_max(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v)); v = fmaxf(v, dpp_f<0x4E>(v));
  v = fmaxf(v, dpp_f<0x141>(v)); v = fmaxf(v, dpp_f<0x140>(v));
  return v;
}
__device__ __forceinline__ float group16_sum(float v) {
  v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); v += dpp_f<0x141>(v); v += dpp_f<0x140>(v);
  return v;
}
// sum over the 8 consecutive lanes l & ~7 .. l | 7 (quad xor 1, xor 2, then the mirror inside each half row); every lane ends
// with the result
__device__ __forceinline__ float group8_sum(float v) {
  v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); v += dpp_f<0x141>(v);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
  return v;
}

// hand-over through LDS between lanes of ONE wave: a wave's LDS operations execute in order, so only the compiler has
// to be told (no s_barrier)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// XCD-aware, bijective block remap (8 XCDs, blocks dealt round-robin): logical tiles
// that are adjacent run on one XCD and share its L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
