// Shared layout, operand and addressing helpers of the fused W-MSA block kernels (wmsa_block.hip: one wave per window,
// f32 parity path; wmsa_hg.hip: four waves per window, bf16 throughput path).  See wmsa_block.hip for the layouts.
#pragma once
#include "common.h"
#include "wmsa_pack.h"
#include "../../include/sodt_hip.h"
#include <type_traits>

// launch arguments of both kernels (external linkage: wmsa_block.hip dispatches bf16 launches to wmsa_hg.hip)
struct WArgs {
  const unsigned char* x; const unsigned char* wpk;
  unsigned char* xm; unsigned char* xn2; float* st1; float* st2;
  unsigned char* xn1; unsigned char* qkvw; float* lsew; unsigned char* ao;
  int B, H, W, shift, nwy, nwx, nwin;
};
int wmsa_hg_launch(const WArgs& a, bool save, hipStream_t st);     // wmsa_hg.hip

namespace {

constexpr int WC = 192, WHD = 16, WHEADS = 12, WWS = 8;
#define WMSA_LOG2E 1.4426950408889634f

typedef __attribute__((ext_vector_type(4))) short s16x4_;

template <typename T> struct WL {
  static constexpr int E = TT<T>::SZ, KPL = TT<T>::KPL, KU = TT<T>::MMA_K;
  static constexpr int KSTEPS = WC / KU;                 // 6 (bf16) / 12 (f32) MFMA k-steps over the channels
  static constexpr int ROWB = WC * E;                    // bytes of a token row: 384 / 768
  static constexpr int NCH = ROWB / 16;                  // 16-byte chunks per row: 24 / 48
  static constexpr int CHL = NCH / 4;                    // chunks per lane and token in the (g, t) mapping: 6 / 12
  static constexpr int K16B = 4 * E;                     // bytes of a k16 operand per lane: 8 / 16
  static constexpr int WFRAG = KSTEPS * 1024;            // Wq_h (or Wk_h, Wv_h) in fragment order: 6 KB / 12 KB
  static constexpr int WQ_OFF = 0, WK_OFF = WFRAG, WV_OFF = 2 * WFRAG;
  static constexpr int BIAS_OFF = 3 * WFRAG;
  static constexpr int BIASB = 4 * 15 * 16 * E;          // four shifted copies of the head's [15][16] table (x log2 e)
  static constexpr int BQKV_OFF = BIAS_OFF + BIASB;      // bq[16] bk[16] bv[16] f32, padded to 256 bytes
  static constexpr int HEAD_BYTES = BQKV_OFF + 256;      // 20608 / 40960
  // output projection: out^T = Wproj O^T contracts over all 192 attention channels AFTER the heads (O^T of the 12 heads
  // stays packed in registers), so Wproj streams through the same stage ring as three more stages of four 16-row strips
  // each: strip n, k-step kp = one 1 KB fragment (lane (g, t): row 16 n + t, 16 bytes of k)
  static constexpr int KP = WC / KU;                     // k-steps of the projection: 6 (head pairs) / 12 (heads)
  static constexpr int PROJ_BYTES = 4 * KP * 1024;       // 24576 / 49152
  static constexpr int STAGE = PROJ_BYTES > HEAD_BYTES ? PROJ_BYTES : HEAD_BYTES;
  static constexpr int NSTG = WHEADS + 3;                // stages per block: 12 heads + 3 projection stages
  static constexpr int TAIL_OFF = NSTG * STAGE;          // bproj[192], g1, b1, g2, b2: f32
  // bf16 only: the weights again as ONE contiguous 288 KB stream for the four-waves-per-window kernel (wmsa_hg.hip), four
  // 72 KB stages it copies into LDS by DMA: stages 0..2 = Wq | Wk | Wv fragments of heads 4 s .. 4 s + 3 (18 KB per head,
  // the first 18 KB of the per-head stages above), stage 3 = Wproj in NATURAL k order (its projection reads the attention
  // output from an LDS tile, 8 consecutive channels per lane): strip n, k-step kp = one 1 KB fragment, lane (g, t): row
  // 16 n + t, columns 32 kp + 8 g + j
  static constexpr int HGW_OFF = TAIL_OFF + 5 * WC * 4;
  static constexpr int HGW_BYTES = E == 2 ? WHEADS * 3 * WFRAG : 0;       // 221184 / 0
  static constexpr int PROJ2_OFF = HGW_OFF + HGW_BYTES;
  static constexpr int PROJ2_BYTES = E == 2 ? WHEADS * KP * 1024 : 0;     // 73728 / 0
  static constexpr int PACK_BYTES = PROJ2_OFF + PROJ2_BYTES;
  static constexpr int XNB = 64 * ROWB;                  // one wave's LN1 tile: 24 KB / 48 KB
  static constexpr int VPB = 64 * WHD * E;               // v transposition patch: 2 KB / 4 KB
};
static_assert(WL<bf16>::STAGE == 24576 && WL<float>::STAGE == 49152, "stage layout");
static_assert(WL<bf16>::STAGE == wmsa_pack_bf16::STAGE && WL<bf16>::WFRAG == wmsa_pack_bf16::WFRAG && WL<bf16>::BQKV_OFF == wmsa_pack_bf16::BQKV_OFF &&
              WL<bf16>::HGW_OFF == wmsa_pack_bf16::HGW_OFF,
              "wmsa_pack.h mirrors the bf16 pack layout");


// ---- k16 operands: 16 contraction elements per MFMA, lane (g, t) holds k = 4 g + j, j = 0..3
template <typename T> struct K16;
template <> struct K16<bf16> { typedef uint2 type; };
template <> struct K16<float> { typedef uint4 type; };

template <typename T> __device__ __forceinline__ typename K16<T>::type pk16(const f32x4& v);
template <> __device__ __forceinline__ uint2 pk16<bf16>(const f32x4& v) { return make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3])); }
template <> __device__ __forceinline__ uint4 pk16<float>(const f32x4& v) {
  return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
}
__device__ __forceinline__ void mmak16(f32x4& acc, const uint2& a, const uint2& b) {      // v_mfma_f32_16x16x16_bf16
  union { uint2 u; s16x4_ v; } ua, ub;
  ua.u = a; ub.u = b;
  acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ua.v, ub.v, acc, 0, 0, 0);
}
__device__ __forceinline__ void mmak16(f32x4& acc, const uint4& a, const uint4& b) { mma16<float>(acc, a, b); }

// ---- hand-scheduled LDS reads.  One wave per SIMD: nothing hides an LDS round trip but the wave's own MFMAs, and hipcc
// places `s_waitcnt lgkmcnt(0)` right behind every read it can see.  The fragment reads of the hot loops are therefore
// inline asm (invisible to the compiler's wait insertion), issued one k-step ahead, retired by counted waits whose
// operand list ties the consuming MFMAs behind them.  LDS returns in order, so a compiler-generated wait can only
// over-wait these reads, never miss them.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_;
template <int OFF> __device__ __forceinline__ u32x4_ lds_rd128a(unsigned addr) {
  u32x4_ v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF> __device__ __forceinline__ u32x2_ lds_rd64a(unsigned addr) {
  u32x2_ v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF> __device__ __forceinline__ unsigned lds_rd32a(unsigned addr) {
  unsigned v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <typename T> struct KR;                     // register image of a k16 operand / four table entries
template <> struct KR<bf16> {
  typedef u32x2_ type;
  template <int OFF> static __device__ __forceinline__ type rd(unsigned a) { return lds_rd64a<OFF>(a); }
  static __device__ __forceinline__ uint2 op(const type& v) { return make_uint2(v.x, v.y); }
  static __device__ __forceinline__ type reg(const uint2& v) { return type{v.x, v.y}; }
  static __device__ __forceinline__ f32x4 f4(const type& v) {
    return f32x4{__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u)};
  }
};
template <> struct KR<float> {
  typedef u32x4_ type;
  template <int OFF> static __device__ __forceinline__ type rd(unsigned a) { return lds_rd128a<OFF>(a); }
  static __device__ __forceinline__ uint4 op(const type& v) { return make_uint4(v.x, v.y, v.z, v.w); }
  static __device__ __forceinline__ type reg(const uint4& v) { return type{v.x, v.y, v.z, v.w}; }
  static __device__ __forceinline__ f32x4 f4(const type& v) {
    return f32x4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
  }
};
__device__ __forceinline__ uint4 u4(const u32x4_& v) { return make_uint4(v.x, v.y, v.z, v.w); }
#define LDS_DEP(x) asm volatile("" : "+v"(x))
#define LAUNDER(p) asm volatile("" : "+v"(p))
#define LDS_WAIT(N) asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory")
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

// reductions over the four 16-lane rows of the wave (lanes t, t + 16, t + 32, t + 48: one token / query held by the four
// g groups): v_permlane32_swap + v_permlane16_swap on the VALU instead of two ds_bpermute round trips through the LDS
// pipe (whose latency nothing hides at one wave per SIMD).  Every lane ends with the result.
__device__ __forceinline__ float rows_sum(float v) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float rows_max(float v) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float s = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// token row (natural order) of window-local token n of window (b, wy, wx) after the cyclic shift (backbone_vit.py:1096)
__device__ __forceinline__ int wtoken(const WArgs& a, int b, int wy, int wx, int n) {
  int y = wy * WWS + (n >> 3) + a.shift, x = wx * WWS + (n & 7) + a.shift;
  if (y >= a.H) y -= a.H;
  if (x >= a.W) x -= a.W;
  return (b * a.H + y) * a.W + x;
}
// mask region of a token in the shifted frame (backbone_vit.py:1061-1072)
__device__ __forceinline__ int wrid(const WArgs& a, int wy, int wx, int n) {
  const int ys = wy * WWS + (n >> 3), xs = wx * WWS + (n & 7);
  const int ry = ys < a.H - WWS ? 0 : (ys < a.H - a.shift ? 1 : 2);
  const int rx = xs < a.W - WWS ? 0 : (xs < a.W - a.shift ? 1 : 2);
  return ry * 3 + rx;
}

}  // namespace
