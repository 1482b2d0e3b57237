// Window multi-head self-attention core (W-MSA / SW-MSA) for gfx950.
//
// Replaces, for one Swin block, the reference's roll -> window_partition -> per-head
// split -> q*scale @ k^T -> + relative_position_bias_table[index] -> + attn_mask(0/-100)
// -> softmax -> @ v -> head merge -> window_unpartition -> roll back
// (backbone_vit.py:1094-1124, :968-989).  The QKV tensor stays in natural token order
// [B*H*W][3C]; shift, partition, mask region and relative index are index arithmetic.
// The N x N score matrix is never written to memory.
//
// One workgroup = NW waves = NW heads of one (window, 64-query tile); keys/values are
// walked in 64-key tiles with an online softmax (one tile when the window is 8x8,
// sixteen when it is 32x32).  Q K^T and P V run on MFMA 16x16 tiles through the
// dtype-generic mma16<T>() of common.h; P goes through LDS to become an A operand and
// V is transposed while it is staged so that both MFMA operands are k-contiguous.
//
// Backward recomputes P from the saved log-sum-exp (flash style): per (q tile, kv
// tile) it forms S, dP = dO V^T, dS = P o (dP - delta), then dV += P^T dO,
// dK += dS^T Q, dQ += dS K.  The relative-position-bias gradient is accumulated in a
// small LDS table per head ((2R-1) x (2ws-1) <= 225 entries, R = 64/ws rows per tile)
// and flushed with a few global atomics.
#include "common.h"
#include "wmsa_pack.h"
#include "../../include/sodt_hip.h"
#include <type_traits>

namespace {

struct AttnGeo {
  int B, H, W, C, heads, ws, shift;
  int nwy, nwx, N, nqt;   // windows per column/row, tokens per window, 64-token tiles per window
};

__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32
#define SODT_LOG2E 1.4426950408889634f

// token row + mask region of window-local token n of window (b, wy, wx)
__device__ __forceinline__ void win_token(const AttnGeo& g, int b, int wy, int wx, int n, int& row, int& rid,
                                          int& iy, int& ix) {
  iy = n / g.ws; ix = n - iy * g.ws;
  const int ys = wy * g.ws + iy, xs = wx * g.ws + ix;
  int y = ys + g.shift, x = xs + g.shift;
  if (y >= g.H) y -= g.H;
  if (x >= g.W) x -= g.W;
  row = (b * g.H + y) * g.W + x;
  rid = 0;
  if (g.shift > 0) {
    const int ry = ys < g.H - g.ws ? 0 : (ys < g.H - g.shift ? 1 : 2);
    const int rx = xs < g.W - g.ws ? 0 : (xs < g.W - g.shift ? 1 : 2);
    rid = ry * 3 + rx;
  }
}

template <typename T, int HD>
struct Lay {
  static constexpr int E = TT<T>::SZ;
  static constexpr int KPL = TT<T>::KPL;
  static constexpr int QROW = HD * E + 16;     // [64 tokens][HD] tile row (bytes)
  static constexpr int TROW = 64 * E + 16;     // [HD][64 tokens] and [64][64] tile row (bytes)
  static constexpr int QTILE = 64 * QROW;
  static constexpr int TTILE = HD * TROW;
  static constexpr int STILE = 64 * TROW;
  static constexpr int DCH = HD / KPL;         // 16-byte chunks per head row
  static constexpr int KBQ = (HD + TT<T>::MMA_K - 1) / TT<T>::MMA_K;   // mma steps over d
  static constexpr int KBT = 64 / TT<T>::MMA_K;                        // mma steps over 64 tokens
};

// MFMA operand fragment from a [rows][K] tile: lane reads 16 bytes at (row0 + (l&15), k = kb*MMA_K + KPL*(l>>4))
template <typename T>
__device__ __forceinline__ uint4 frag(const unsigned char* tile, int rowbytes, int row0, int kb, int klimit, int lane) {
  const int k = kb * TT<T>::MMA_K + TT<T>::KPL * (lane >> 4);
  if (k >= klimit) return make_uint4(0, 0, 0, 0);
  return *(const uint4*)(tile + (row0 + (lane & 15)) * rowbytes + k * TT<T>::SZ);
}

// MFMA operand fragment whose contraction index runs along the LDS ROWS of a [K rows][cols] tile
// (lane reads column col0 + (l&15), rows k0 + KPL*(l>>4) .. +KPL-1): bf16 through the transposing
// ds_read_b64_tr_b16 (two reads of 4 rows), f32 through four strided ds_read_b32.  EXEC must be full.
template <typename T> __device__ __forceinline__ uint4 fragT(const unsigned char* tile, int rowbytes, int k0, int col0, int lane);
template <> __device__ __forceinline__ uint4 fragT<bf16>(const unsigned char* tile, int rowbytes, int k0, int col0, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const unsigned char* a = tile + (k0 + 8 * g + q) * rowbytes + (col0 + 4 * p) * 2;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  union { s16x4 v; uint2 u; } lo, hi;
  lo.v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
  hi.v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * rowbytes));
  return make_uint4(lo.u.x, lo.u.y, hi.u.x, hi.u.y);
}
template <> __device__ __forceinline__ uint4 fragT<float>(const unsigned char* tile, int rowbytes, int k0, int col0, int lane) {
  const unsigned char* a = tile + (k0 + 4 * (lane >> 4)) * rowbytes + (col0 + (lane & 15)) * 4;
  uint4 r;
  r.x = *(const uint32_t*)(a);
  r.y = *(const uint32_t*)(a + rowbytes);
  r.z = *(const uint32_t*)(a + 2 * rowbytes);
  r.w = *(const uint32_t*)(a + 3 * rowbytes);
  return r;
}

template <typename T> __device__ __forceinline__ void st_elem(unsigned char* p, float v) { *(T*)p = from_f<T>(v); }

// store 4 consecutive elements (accumulator rows 4g..4g+3 of one column) into a transposed tile
template <typename T> __device__ __forceinline__ void st4(unsigned char* p, const f32x4& v);
template <> __device__ __forceinline__ void st4<float>(unsigned char* p, const f32x4& v) {
  *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void st4<bf16>(unsigned char* p, const f32x4& v) {
  *(uint2*)p = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
}

// ---------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------
template <typename T, int HD, int NW>
__global__ __launch_bounds__(NW * 64) void attn_fwd_kernel(const T* __restrict__ qkv, const float* __restrict__ bias_t,
                                                          T* __restrict__ out, float* __restrict__ lse, const AttnGeo g) {
  using L = Lay<T, HD>;
  constexpr int E = L::E, KPL = L::KPL;
  constexpr int NT = NW * 64;
  __shared__ __attribute__((aligned(16))) unsigned char sQ[NW * L::QTILE];
  __shared__ __attribute__((aligned(16))) unsigned char sK[NW * L::QTILE];
  __shared__ __attribute__((aligned(16))) unsigned char sV[NW * L::QTILE];
  __shared__ __attribute__((aligned(16))) unsigned char sP[NW * L::STILE];
  __shared__ int sTokQ[64], sTokK[64];
  __shared__ short sGeoQ[64][4], sGeoK[64][4];   // iy, ix, rid, -
  __shared__ float sBias[NW][228];               // bias rows reachable from this (q tile, kv tile) pair

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int HG = g.heads / NW;
  int bid = blockIdx.x;
  const int hg = bid % HG; bid /= HG;
  const int qt = bid % g.nqt; bid /= g.nqt;
  const int wx = bid % g.nwx; bid /= g.nwx;
  const int wy = bid % g.nwy; const int b = bid / g.nwy;
  const int head = hg * NW + w;
  const int C3 = 3 * g.C;
  const int L2 = 2 * g.ws - 1;

  if (tid < 64) {
    int row, rid, iy, ix;
    win_token(g, b, wy, wx, qt * 64 + tid, row, rid, iy, ix);
    sTokQ[tid] = row; sGeoQ[tid][0] = (short)iy; sGeoQ[tid][1] = (short)ix; sGeoQ[tid][2] = (short)rid;
  }
  __syncthreads();
  // ---- Q tile: 64 rows x NW*HD
  constexpr int CPR = NW * L::DCH;
  for (int idx = tid; idx < 64 * CPR; idx += NT) {
    const int r = idx / CPR, cc = idx - r * CPR;
    const int h = cc / L::DCH, dc = cc - h * L::DCH;
    const uint4 v = *(const uint4*)(qkv + (long)sTokQ[r] * C3 + (hg * NW) * HD + cc * KPL);
    *(uint4*)(sQ + (h * 64 + r) * L::QROW + dc * 16) = v;
  }

  const int fr = lane & 15, fg = lane >> 4;
  const float scale = rsqrtf((float)HD);
  float m_run[4][4], l_run[4][4];
  f32x4 o[4][HD / 16];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { m_run[i][r] = -1e30f; l_run[i][r] = 0.f; }
#pragma unroll
    for (int d = 0; d < HD / 16; ++d) o[i][d] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const unsigned char* myQ = sQ + w * L::QTILE;
  const unsigned char* myK = sK + w * L::QTILE;
  const unsigned char* myV = sV + w * L::QTILE;
  unsigned char* myP = sP + w * L::STILE;
  const float* bt = bias_t + (long)head * L2 * L2;
  const int R = g.ws >= 64 ? 1 : 64 / g.ws;     // window rows per 64-token tile
  const int LT = (2 * R - 1) * L2;

  for (int kt = 0; kt < g.nqt; ++kt) {
    if (tid < 64) {
      int row, rid, iy, ix;
      win_token(g, b, wy, wx, kt * 64 + tid, row, rid, iy, ix);
      sTokK[tid] = row; sGeoK[tid][0] = (short)iy; sGeoK[tid][1] = (short)ix; sGeoK[tid][2] = (short)rid;
    }
    const int dyoff = (qt - kt) * R;
    for (int i = lane; i < LT; i += 64) {
      const int a = i / L2, c = i - a * L2;
      const int gy = a - (R - 1) + dyoff + g.ws - 1;
      sBias[w][i] = (gy >= 0 && gy < L2) ? bt[gy * L2 + c] : 0.f;
    }
    __syncthreads();
    for (int idx = tid; idx < 64 * CPR; idx += NT) {
      const int r = idx / CPR, cc = idx - r * CPR;
      const int h = cc / L::DCH, dc = cc - h * L::DCH;
      const T* src = qkv + (long)sTokK[r] * C3 + (hg * NW) * HD + cc * KPL;
      const uint4 kv = *(const uint4*)(src + g.C);
      const uint4 vv = *(const uint4*)(src + 2 * g.C);
      *(uint4*)(sK + (h * 64 + r) * L::QROW + dc * 16) = kv;
      *(uint4*)(sV + (h * 64 + r) * L::QROW + dc * 16) = vv;
    }
    __syncthreads();

    // ---- S = Q K^T (per 16-row strip), bias, mask, online softmax, P -> LDS
#pragma unroll
    for (int ms = 0; ms < 4; ++ms) {
      f32x4 s[4];
#pragma unroll
      for (int ns = 0; ns < 4; ++ns) s[ns] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < L::KBQ; ++kb) {
        const uint4 fa = frag<T>(myQ, L::QROW, ms * 16, kb, HD, lane);
#pragma unroll
        for (int ns = 0; ns < 4; ++ns) {
          const uint4 fb = frag<T>(myK, L::QROW, ns * 16, kb, HD, lane);
          mma16<T>(s[ns], fa, fb);
        }
      }
      int kiy[4], kix[4], krid[4];
#pragma unroll
      for (int ns = 0; ns < 4; ++ns) {
        kiy[ns] = sGeoK[ns * 16 + fr][0]; kix[ns] = sGeoK[ns * 16 + fr][1]; krid[ns] = sGeoK[ns * 16 + fr][2];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qn = ms * 16 + fg * 4 + r;
        const int qiy = sGeoQ[qn][0], qix = sGeoQ[qn][1], qrid = sGeoQ[qn][2];
        float mx = -1e30f;
#pragma unroll
        for (int ns = 0; ns < 4; ++ns) {
          float v = s[ns][r] * scale + sBias[w][(qiy - kiy[ns] - dyoff + R - 1) * L2 + (qix - kix[ns] + g.ws - 1)];
          if (qrid != krid[ns]) v += -100.0f;
          s[ns][r] = v;
          mx = fmaxf(mx, v);
        }
        mx = group16_max(mx);
        const float mnew = fmaxf(m_run[ms][r], mx);
        const float alpha = fast_exp(m_run[ms][r] - mnew);
        float rs = 0.f;
#pragma unroll
        for (int ns = 0; ns < 4; ++ns) { const float p = fast_exp(s[ns][r] - mnew); s[ns][r] = p; rs += p; }
        rs = group16_sum(rs);
        l_run[ms][r] = l_run[ms][r] * alpha + rs;
        m_run[ms][r] = mnew;
#pragma unroll
        for (int d = 0; d < HD / 16; ++d) o[ms][d][r] *= alpha;
#pragma unroll
        for (int ns = 0; ns < 4; ++ns) st_elem<T>(myP + qn * L::TROW + (ns * 16 + fr) * E, s[ns][r]);
      }
    }
    __syncthreads();
    // ---- O += P V
#pragma unroll
    for (int kb = 0; kb < L::KBT; ++kb) {
      uint4 fb[HD / 16];
#pragma unroll
      for (int d = 0; d < HD / 16; ++d) fb[d] = fragT<T>(myV, L::QROW, kb * TT<T>::MMA_K, d * 16, lane);
#pragma unroll
      for (int ms = 0; ms < 4; ++ms) {
        const uint4 fa = frag<T>(myP, L::TROW, ms * 16, kb, 64, lane);
#pragma unroll
        for (int d = 0; d < HD / 16; ++d) mma16<T>(o[ms][d], fa, fb[d]);
      }
    }
    __syncthreads();
  }

  // ---- normalise, write lse, stage O through this head's Q tile, coalesced store
  unsigned char* myO = sQ + w * L::QTILE;
#pragma unroll
  for (int ms = 0; ms < 4; ++ms)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qn = ms * 16 + fg * 4 + r;
      const float inv = 1.0f / l_run[ms][r];
      if (fr == 0 && lse) lse[(long)sTokQ[qn] * g.heads + head] = m_run[ms][r] + __logf(l_run[ms][r]);
#pragma unroll
      for (int d = 0; d < HD / 16; ++d) st_elem<T>(myO + qn * L::QROW + (d * 16 + fr) * E, o[ms][d][r] * inv);
    }
  __syncthreads();
  for (int idx = tid; idx < 64 * CPR; idx += NT) {
    const int r = idx / CPR, cc = idx - r * CPR;
    const int h = cc / L::DCH, dc = cc - h * L::DCH;
    *(uint4*)(out + (long)sTokQ[r] * g.C + (hg * NW) * HD + cc * KPL) = *(const uint4*)(sQ + (h * 64 + r) * L::QROW + dc * 16);
  }
}

// P V operand of the register-resident P strips: V rows (keys) in strip-pair order (see attn_fwd_mt_kernel)
template <typename T> __device__ __forceinline__ uint4 fragTp_fwd(const unsigned char* tile, int rowbytes, int kbq, int col0, int lane);
template <> __device__ __forceinline__ uint4 fragTp_fwd<bf16>(const unsigned char* tile, int rowbytes, int kbq, int col0, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const unsigned char* a = tile + (32 * kbq + 4 * g + q) * rowbytes + (col0 + 4 * p) * 2;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  union { s16x4 v; uint2 u; } lo, hi;
  lo.v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
  hi.v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 16 * rowbytes));
  return make_uint4(lo.u.x, lo.u.y, hi.u.x, hi.u.y);
}
template <> __device__ __forceinline__ uint4 fragTp_fwd<float>(const unsigned char* tile, int rowbytes, int kbq, int col0, int lane) {
  return fragT<float>(tile, rowbytes, 16 * kbq, col0, lane);
}

// ---------------------------------------------------------------------------------
// forward, 8x8 windows (one 64-token tile per window: stages 1 and 2).  Persistent workgroups walk the
// windows of one head group; the next window's Q/K/V chunks are prefetched into registers.  Scores are
// computed TRANSPOSED (S^T = K Q^T: keys on accumulator rows, queries on lanes) so that a query's softmax
// is 16 in-lane values + two cross-group shuffles, and P leaves the accumulators directly as the A operand of O += P V
// (strip pairs for bf16; V read with the transposing LDS load in the same order): no P tile in LDS.  Bias values
// (x log2 e) sit in 28 registers (strip differences).  bf16, four heads per workgroup: 61 KB of LDS and <= 256 registers, two
// workgroups per CU.
// ---------------------------------------------------------------------------------
template <typename T, int HD, int NW>
__global__ __launch_bounds__(NW * 64, (NW == 4 && sizeof(T) == 2) ? 2 : 1) void attn_fwd_fast_kernel(const T* __restrict__ qkv, const float* __restrict__ bias_t,
                                                               T* __restrict__ out, float* __restrict__ lse, const AttnGeo g,
                                                               int nwin_total) {
  using L = Lay<T, HD>;
  constexpr int E = L::E, KPL = L::KPL, NT = NW * 64, CPR = NW * L::DCH, NPF = L::DCH, MK = TT<T>::MMA_K;
  constexpr int SPK = MK / 16;
  __shared__ __attribute__((aligned(16))) unsigned char sQ[NW * L::QTILE], sK[NW * L::QTILE], sV[NW * L::QTILE];
  __shared__ float sL[NW][64];
  __shared__ int sTok[64];
  __shared__ short sRid[64];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int hg = blockIdx.y, head = hg * NW + w;
  const int C3 = 3 * g.C, L2 = 2 * g.ws - 1;
  const float scale2 = rsqrtf((float)HD) * SODT_LOG2E;
  const float* bt = bias_t + (long)head * L2 * L2;
  unsigned char* myQ = sQ + w * L::QTILE; unsigned char* myK = sK + w * L::QTILE;
  unsigned char* myV = sV + w * L::QTILE;

  // 8x8 windows: a 16-token strip is two window rows, so the table entry of (key, query) depends on the strips only
  // through ms - ks: 7 x 4 bias registers.  bias7[ms - ks + 3][r]: key = ks*16 + 4*fg + r (accumulator row), query = ms*16 + fr
  float bias7[7][4];
#pragma unroll
  for (int d = 0; d < 7; ++d) {
    const int ks = d < 3 ? 3 - d : 0, ms = ks + d - 3;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kn = ks * 16 + fg * 4 + r, qn = ms * 16 + fr;
      const int kiy = kn / g.ws, kix = kn - kiy * g.ws, qiy = qn / g.ws, qix = qn - qiy * g.ws;
      bias7[d][r] = bt[(qiy - kiy + g.ws - 1) * L2 + (qix - kix + g.ws - 1)] * SODT_LOG2E;
    }
  }

  // prefetch slots as individual registers (arrays / structs were left in scratch by hipcc, serialising the prefetch)
  uint4 pq0, pq1, pq2, pq3, pk0, pk1, pk2, pk3, pv0, pv1, pv2, pv3;
#define FWD_ISSUE_ONE(i, ITEM)                                                          \
  if constexpr (NPF > i) {                                                              \
    int t_ = (ITEM);                                                                    \
    const int wx_ = t_ % g.nwx; t_ /= g.nwx;                                            \
    const int wy_ = t_ % g.nwy; const int b_ = t_ / g.nwy;                              \
    const int idx = tid + i * NT;                                                       \
    const int r = idx / CPR, cc = idx - r * CPR;                                        \
    int row, rid, iy, ix;                                                               \
    win_token(g, b_, wy_, wx_, r, row, rid, iy, ix);                                    \
    const T* src = qkv + (long)row * C3 + (hg * NW) * HD + cc * KPL;                    \
    pq##i = *(const uint4*)(src); pk##i = *(const uint4*)(src + g.C); pv##i = *(const uint4*)(src + 2 * g.C); \
  }
#define FWD_ISSUE(ITEM) { FWD_ISSUE_ONE(0, ITEM) FWD_ISSUE_ONE(1, ITEM) FWD_ISSUE_ONE(2, ITEM) FWD_ISSUE_ONE(3, ITEM) }
#define FWD_STORE_ONE(i)                                                                \
  if constexpr (NPF > i) {                                                              \
    const int idx = tid + i * NT;                                                       \
    const int r = idx / CPR, cc = idx - r * CPR;                                        \
    const int h = cc / L::DCH, dc = cc - h * L::DCH;                                    \
    const int off = (h * 64 + r) * L::QROW + dc * 16;                                   \
    *(uint4*)(sQ + off) = pq##i; *(uint4*)(sK + off) = pk##i; *(uint4*)(sV + off) = pv##i; \
  }
  FWD_ISSUE((int)blockIdx.x < nwin_total ? (int)blockIdx.x : 0)

  for (int item = blockIdx.x; item < nwin_total; item += gridDim.x) {
    int t = item;
    const int wx = t % g.nwx; t /= g.nwx;
    const int wy = t % g.nwy; const int b = t / g.nwy;
    const bool msk = g.shift > 0 && (wy == g.nwy - 1 || wx == g.nwx - 1);
    __syncthreads();                                   // previous window fully consumed
    if (tid < 64) {
      int row, rid, iy, ix;
      win_token(g, b, wy, wx, tid, row, rid, iy, ix);
      sTok[tid] = row; sRid[tid] = (short)rid;
    }
    FWD_STORE_ONE(0) FWD_STORE_ONE(1) FWD_STORE_ONE(2) FWD_STORE_ONE(3)
    __syncthreads();
    FWD_ISSUE(item + (int)gridDim.x < nwin_total ? item + (int)gridDim.x : item)

    // the window body is compiled twice (see attn_bwd_fast2_kernel): only the last window row / column of a shifted
    // block mixes mask regions, and a run-time test per element is if-converted into compares + selects for everyone
    auto body = [&](auto MSK_) {
    constexpr bool MSK = decltype(MSK_)::value;
    // ---- S^T = K Q^T
    f32x4 s[4][4];
#pragma unroll
    for (int kb = 0; kb < L::KBQ; ++kb) {
      uint4 fk[4], fq[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { fk[i] = frag<T>(myK, L::QROW, i * 16, kb, HD, lane); fq[i] = frag<T>(myQ, L::QROW, i * 16, kb, HD, lane); }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int ms = 0; ms < 4; ++ms) {
          if (kb == 0) s[ks][ms] = mma16z<T>(fk[ks], fq[ms]); else mma16<T>(s[ks][ms], fk[ks], fq[ms]);
        }
    }
    // V^T-side operands of O += P V (rows = keys in strip-pair order), shared by the four query strips
    uint4 fb[4 / SPK][HD / 16];
#pragma unroll
    for (int kbq = 0; kbq < 4 / SPK; ++kbq)
#pragma unroll
      for (int d = 0; d < HD / 16; ++d) fb[kbq][d] = fragTp_fwd<T>(myV, L::QROW, kbq, d * 16, lane);
    f32x4 o[4][HD / 16];
    // ---- softmax over keys for query (ms, fr): 16 in-lane values, then across the four 16-lane groups; P leaves the
    //      accumulators as the A operand of O[ms] += P V (no P tile in LDS)
#pragma unroll
    for (int ms = 0; ms < 4; ++ms) {
      const int qrid = MSK ? (int)sRid[ms * 16 + fr] : 0;
      float mx = -1e30f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = fmaf(s[ks][ms][r], scale2, bias7[ms - ks + 3][r]);
          if constexpr (MSK) { if (qrid != (int)sRid[ks * 16 + fg * 4 + r]) v += -100.0f * SODT_LOG2E; }
          s[ks][ms][r] = v;
          mx = fmaxf(mx, v);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      float sum = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float p = fast_exp2(s[ks][ms][r] - mx); s[ks][ms][r] = p; sum += p; }
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      if (fg == 0) {
        const int qn = ms * 16 + fr;
        sL[w][qn] = __builtin_amdgcn_rcpf(sum);
        if (lse) lse[(long)sTok[qn] * g.heads + head] = mx * (1.0f / SODT_LOG2E) + __logf(sum);
      }
#pragma unroll
      for (int kbq = 0; kbq < 4 / SPK; ++kbq) {
        uint4 ap;
        if constexpr (std::is_same<T, bf16>::value) {
          ap = make_uint4(pack2bf(s[2 * kbq][ms][0], s[2 * kbq][ms][1]), pack2bf(s[2 * kbq][ms][2], s[2 * kbq][ms][3]),
                          pack2bf(s[2 * kbq + 1][ms][0], s[2 * kbq + 1][ms][1]), pack2bf(s[2 * kbq + 1][ms][2], s[2 * kbq + 1][ms][3]));
        } else {
          ap = make_uint4(__float_as_uint(s[kbq][ms][0]), __float_as_uint(s[kbq][ms][1]), __float_as_uint(s[kbq][ms][2]), __float_as_uint(s[kbq][ms][3]));
        }
#pragma unroll
        for (int d = 0; d < HD / 16; ++d) {
          if (kbq == 0) o[ms][d] = mma16z<T>(ap, fb[kbq][d]); else mma16<T>(o[ms][d], ap, fb[kbq][d]);
        }
      }
    }
    __syncthreads();                                   // sL (1 / sum of every query) is complete
    // ---- normalise, stage through this head's Q tile, coalesced store
#pragma unroll
    for (int ms = 0; ms < 4; ++ms)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qn = ms * 16 + fg * 4 + r;
        const float inv = sL[w][qn];
#pragma unroll
        for (int d = 0; d < HD / 16; ++d) st_elem<T>(myQ + qn * L::QROW + (d * 16 + fr) * E, o[ms][d][r] * inv);
      }
    };
    if (msk) body(std::true_type{}); else body(std::false_type{});
    __syncthreads();
    for (int idx = tid; idx < 64 * CPR; idx += NT) {
      const int r = idx / CPR, cc = idx - r * CPR;
      const int h = cc / L::DCH, dc = cc - h * L::DCH;
      *(uint4*)(out + (long)sTok[r] * g.C + (hg * NW) * HD + cc * KPL) = *(const uint4*)(sQ + (h * 64 + r) * L::QROW + dc * 16);
    }
  }
}

// ---------------------------------------------------------------------------------
// forward, windows of more than 64 tokens.  One workgroup = four waves = four 64-query tiles of ONE (window, head):
// the K / V tiles are staged once per workgroup and shared (a quarter of the L2 traffic of one tile per workgroup), the
// next K / V tile is prefetched into registers.  Scores are computed transposed (S^T = K Q^T: keys on the accumulator
// rows, a query per lane column), so the online softmax of a query is in-lane work plus two cross-group shuffles and
// P leaves the accumulators directly as the A operand of O += P V (strip pairs for bf16; V read with the transposing
// LDS load in the same order).  No P tile in LDS.
// ---------------------------------------------------------------------------------
template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_fwd_mt_kernel(const T* __restrict__ qkv, const float* __restrict__ bias_t,
                                                         T* __restrict__ out, float* __restrict__ lse, const AttnGeo g) {
  using L = Lay<T, HD>;
  constexpr int E = L::E, KPL = L::KPL, MK = TT<T>::MMA_K, SPK = MK / 16;
  constexpr int NPF = 64 * L::DCH / 256;         // K (and V) chunks per thread per tile
  static_assert(NPF >= 1 && NPF <= 4, "prefetch slots");
  __shared__ __attribute__((aligned(16))) unsigned char sQ[4 * L::QTILE];     // one query tile per wave
  __shared__ __attribute__((aligned(16))) unsigned char sK[L::QTILE], sV[L::QTILE];
  __shared__ int sTokQ[4][64], sTokK[64];
  __shared__ short sGeoQ[4][64][4], sGeoK[64][4];
  __shared__ float sBias[4][228];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  int bid = blockIdx.x;
  const int head = bid % g.heads; bid /= g.heads;
  const int qg = bid % (g.nqt / 4); bid /= (g.nqt / 4);
  const int wx = bid % g.nwx; bid /= g.nwx;
  const int wy = bid % g.nwy; const int b = bid / g.nwy;
  const int qt = qg * 4 + w;
  const int C3 = 3 * g.C, L2 = 2 * g.ws - 1;
  const int R = g.ws >= 64 ? 1 : 64 / g.ws;
  const int LT = (2 * R - 1) * L2;
  const float scale2 = rsqrtf((float)HD) * SODT_LOG2E;
  const float* bt = bias_t + (long)head * L2 * L2;
  const bool msk = g.shift > 0 && (wy == g.nwy - 1 || wx == g.nwx - 1);
  unsigned char* myQ = sQ + w * L::QTILE;

  {
    int row, rid, iy, ix;
    win_token(g, b, wy, wx, qt * 64 + lane, row, rid, iy, ix);
    sTokQ[w][lane] = row; sGeoQ[w][lane][0] = (short)iy; sGeoQ[w][lane][1] = (short)ix; sGeoQ[w][lane][2] = (short)rid;
  }
  __syncthreads();
  for (int idx = lane; idx < 64 * L::DCH; idx += 64) {           // this wave's own query tile
    const int r = idx / L::DCH, dc = idx - r * L::DCH;
    *(uint4*)(myQ + r * L::QROW + dc * 16) = *(const uint4*)(qkv + (long)sTokQ[w][r] * C3 + head * HD + dc * KPL);
  }
  // this lane's query column of every strip: geometry for bias / mask
  int qiy[4], qix[4], qrid[4];
#pragma unroll
  for (int ms = 0; ms < 4; ++ms) {
    qiy[ms] = sGeoQ[w][ms * 16 + fr][0]; qix[ms] = sGeoQ[w][ms * 16 + fr][1]; qrid[ms] = sGeoQ[w][ms * 16 + fr][2];
  }
  float m_run[4], l_run[4];
  f32x4 o[4][HD / 16];
#pragma unroll
  for (int ms = 0; ms < 4; ++ms) {
    m_run[ms] = -1e30f; l_run[ms] = 0.f;
#pragma unroll
    for (int d = 0; d < HD / 16; ++d) o[ms][d] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  uint4 pk0, pk1, pk2, pk3, pv0, pv1, pv2, pv3;
#define FM_ISSUE_ONE(i, KT_)                                                            \
  if constexpr (NPF > i) {                                                              \
    const int idx = tid + i * 256;                                                      \
    const int r = idx / L::DCH, dc = idx - r * L::DCH;                                  \
    int row, rid, iy, ix;                                                               \
    win_token(g, b, wy, wx, (KT_) * 64 + r, row, rid, iy, ix);                          \
    const T* src = qkv + (long)row * C3 + head * HD + dc * KPL;                         \
    pk##i = *(const uint4*)(src + g.C); pv##i = *(const uint4*)(src + 2 * g.C);         \
  }
#define FM_ISSUE(KT_) { FM_ISSUE_ONE(0, KT_) FM_ISSUE_ONE(1, KT_) FM_ISSUE_ONE(2, KT_) FM_ISSUE_ONE(3, KT_) }
#define FM_STORE_ONE(i)                                                                 \
  if constexpr (NPF > i) {                                                              \
    const int idx = tid + i * 256;                                                      \
    const int r = idx / L::DCH, dc = idx - r * L::DCH;                                  \
    *(uint4*)(sK + r * L::QROW + dc * 16) = pk##i; *(uint4*)(sV + r * L::QROW + dc * 16) = pv##i; \
  }
  FM_ISSUE(0)

  for (int kt = 0; kt < g.nqt; ++kt) {
    __syncthreads();                                   // every wave is done with the previous K / V tile
    if (tid < 64) {
      int row, rid, iy, ix;
      win_token(g, b, wy, wx, kt * 64 + tid, row, rid, iy, ix);
      sGeoK[tid][0] = (short)iy; sGeoK[tid][1] = (short)ix; sGeoK[tid][2] = (short)rid;
    }
    FM_STORE_ONE(0) FM_STORE_ONE(1) FM_STORE_ONE(2) FM_STORE_ONE(3)
    const int dyoff = (qt - kt) * R;
    for (int i = lane; i < LT; i += 64) {
      const int a = i / L2, c = i - a * L2;
      const int gy = a - (R - 1) + dyoff + g.ws - 1;
      sBias[w][i] = (gy >= 0 && gy < L2) ? bt[gy * L2 + c] * SODT_LOG2E : 0.f;
    }
    __syncthreads();
    { const int nk_ = kt + 1 < g.nqt ? kt + 1 : kt; FM_ISSUE(nk_) }

#pragma unroll
    for (int ms = 0; ms < 4; ++ms) {
      // S^T strip: keys on rows (4 fg + r of key strip ks), this lane's query = ms*16 + fr
      f32x4 s[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) s[ks] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < L::KBQ; ++kb) {
        const uint4 fq = frag<T>(myQ, L::QROW, ms * 16, kb, HD, lane);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) mma16<T>(s[ks], frag<T>(sK, L::QROW, ks * 16, kb, HD, lane), fq);
      }
      float mx = -1e30f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kn = ks * 16 + fg * 4 + r;
          const int kiy = sGeoK[kn][0], kix = sGeoK[kn][1];
          float v = fmaf(s[ks][r], scale2, sBias[w][(qiy[ms] - kiy - dyoff + R - 1) * L2 + (qix[ms] - kix + g.ws - 1)]);
          if (msk && qrid[ms] != (int)sGeoK[kn][2]) v += -100.0f * SODT_LOG2E;
          s[ks][r] = v;
          mx = fmaxf(mx, v);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float mnew = fmaxf(m_run[ms], mx);
      const float alpha = fast_exp2(m_run[ms] - mnew);
      float sum = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float p = fast_exp2(s[ks][r] - mnew); s[ks][r] = p; sum += p; }
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      l_run[ms] = l_run[ms] * alpha + sum;
      m_run[ms] = mnew;
      // rescale O rows (row 4 fg + r of strip ms <-> query column 4 fg + r of this strip, held by lane 4 fg + r)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ar = __shfl(alpha, 4 * fg + r);
#pragma unroll
        for (int d = 0; d < HD / 16; ++d) o[ms][d][r] *= ar;
      }
      // O[ms] += P V: P strips straight from the accumulators (A operand: row = query fr, k-slots = keys)
#pragma unroll
      for (int kbq = 0; kbq < 4 / SPK; ++kbq) {
        uint4 ap;
        if constexpr (std::is_same<T, bf16>::value) {
          ap = make_uint4(pack2bf(s[2 * kbq][0], s[2 * kbq][1]), pack2bf(s[2 * kbq][2], s[2 * kbq][3]),
                          pack2bf(s[2 * kbq + 1][0], s[2 * kbq + 1][1]), pack2bf(s[2 * kbq + 1][2], s[2 * kbq + 1][3]));
        } else {
          ap = make_uint4(__float_as_uint(s[kbq][0]), __float_as_uint(s[kbq][1]), __float_as_uint(s[kbq][2]), __float_as_uint(s[kbq][3]));
        }
#pragma unroll
        for (int d = 0; d < HD / 16; ++d) mma16<T>(o[ms][d], ap, fragTp_fwd<T>(sV, L::QROW, kbq, d * 16, lane));
      }
    }
  }
  // ---- normalise, lse, stage O through this wave's query tile, coalesced store
#pragma unroll
  for (int ms = 0; ms < 4; ++ms) {
    const float inv = __builtin_amdgcn_rcpf(l_run[ms]);
    if (fg == 0 && lse) lse[(long)sTokQ[w][ms * 16 + fr] * g.heads + head] = m_run[ms] * (1.0f / SODT_LOG2E) + __logf(l_run[ms]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float ir = __shfl(inv, 4 * fg + r);
#pragma unroll
      for (int d = 0; d < HD / 16; ++d)
        st_elem<T>(myQ + (ms * 16 + fg * 4 + r) * L::QROW + (d * 16 + fr) * E, o[ms][d][r] * ir);
    }
  }
  for (int idx = lane; idx < 64 * L::DCH; idx += 64) {
    const int r = idx / L::DCH, dc = idx - r * L::DCH;
    *(uint4*)(out + (long)sTokQ[w][r] * g.C + head * HD + dc * KPL) = *(const uint4*)(myQ + r * L::QROW + dc * 16);
  }
#undef FM_ISSUE_ONE
#undef FM_ISSUE
#undef FM_STORE_ONE
}

// ---------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------
// delta[tok][head] = sum_d dO * O   (needed only when a window has more than one key tile).  One 16-byte chunk of
// both tensors per lane, reduced over the hd / KPL lanes of a head with DPP-free xor shuffles: every byte is read once.
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(const T* __restrict__ o, const T* __restrict__ d_o,
                                                        float* __restrict__ delta, long M, int C, int heads) {
  constexpr int KPL = TT<T>::KPL;
  const int cpr = C / KPL;                 // chunks per token row
  const int cph = cpr / heads;             // chunks (= lanes) per head: a power of two <= 16
  const long total = M * cpr;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (total + 63) / 64 * 64; i += (long)gridDim.x * 256) {
    float s = 0.f;
    if (i < total) {
      float a[KPL], b[KPL];
      unpack<T>(*(const uint4*)(o + i * KPL), a);
      unpack<T>(*(const uint4*)(d_o + i * KPL), b);
#pragma unroll
      for (int j = 0; j < KPL; ++j) s = fmaf(a[j], b[j], s);
    }
    for (int w = 1; w < cph; w <<= 1) s += __shfl_xor(s, w);
    if (i < total && (i % cph) == 0) {
      const long tok = i / cpr; const int h = (int)(i - tok * cpr) / cph;
      delta[tok * heads + h] = s;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_dq_finish_kernel(float* __restrict__ acc, T* __restrict__ dqkv,
                                                            long M, int C) {
  const long total = M * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long tok = i / C; const int c = (int)(i - tok * C);
    dqkv[tok * 3 * C + c] = from_f<T>(acc[i]);
    acc[i] = 0.f;   // the accumulator is left zeroed for the next call
  }
}

template <typename T, int HD, int NW, bool FAST>
__global__ __launch_bounds__(NW * 64) void attn_bwd_kernel(const T* __restrict__ qkv, const float* __restrict__ bias_t,
                                                          const T* __restrict__ d_out, const float* __restrict__ lse,
                                                          const float* __restrict__ delta, T* __restrict__ dqkv,
                                                          float* __restrict__ dbias_t, float* __restrict__ dq_acc,
                                                          const AttnGeo g, int nwin_total, int fulldb) {
  using L = Lay<T, HD>;
  constexpr int E = L::E, KPL = L::KPL;
  constexpr int NT = NW * 64;
  constexpr int LTMAX = 225;
  constexpr int MK = TT<T>::MMA_K;
  extern __shared__ __attribute__((aligned(16))) float dbfull[];   // [NW][(2ws-1)^2] when fulldb (multi-tile windows)
  // only row-major [token][d] tiles are staged; every "transposed" MFMA operand is read with fragT
  __shared__ __attribute__((aligned(16))) unsigned char sQ[NW * L::QTILE], sK[NW * L::QTILE], sV[NW * L::QTILE],
      sDO[NW * L::QTILE];
  __shared__ __attribute__((aligned(16))) unsigned char sPT[NW * L::STILE], sDST[NW * L::STILE];   // [key][q]
  __shared__ float sDB[NW][LTMAX + 3];
  __shared__ float sBias[NW][LTMAX + 3];
  __shared__ float sLse[NW][64], sDelta[NW][64];
  __shared__ int sTokQ[64], sTokK[64];
  __shared__ short sGeoQ[64][4], sGeoK[64][4];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int hg = blockIdx.y;
  const int head = hg * NW + w;
  const int C3 = 3 * g.C;
  const int L2 = 2 * g.ws - 1;
  const int R = g.ws >= 64 ? 1 : 64 / g.ws;      // window rows per 64-token tile
  const int LT = (2 * R - 1) * L2;
  const bool single = FAST || g.nqt == 1;
  const float scale = rsqrtf((float)HD);
  const float* bt = bias_t + (long)head * L2 * L2;
  constexpr int CPR = NW * L::DCH;

  for (int i = tid; i < NW * (LTMAX + 3); i += NT) (&sDB[0][0])[i] = 0.f;
  if (fulldb)
    for (int i = tid; i < NW * L2 * L2; i += NT) dbfull[i] = 0.f;

  unsigned char* myQ = sQ + w * L::QTILE; unsigned char* myK = sK + w * L::QTILE;
  unsigned char* myV = sV + w * L::QTILE; unsigned char* myDO = sDO + w * L::QTILE;
  unsigned char* myPT = sPT + w * L::STILE; unsigned char* myDST = sDST + w * L::STILE;

  // relative-position-bias gradient: dS summed in registers over every window this wave visits (the
  // (q, key) -> table-entry map is the same for all of them), reduced through LDS only when it changes
  f32x4 dbacc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) dbacc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nitems = nwin_total * g.nqt;
  // Single-tile windows (8x8: stages 1 and 2): the next window's Q/K/V/dO chunks and lse are prefetched into
  // registers while the current window is computed, so no global-memory latency is exposed inside the loop.
  constexpr bool pf = FAST;                       // FAST <=> one 64-token tile per window and 4*DCH <= 16 registers
  constexpr int NPF = FAST ? L::DCH : 1;          // chunks per thread and tensor (64*CPR / NT)
  uint4 pq0, pq1, pq2, pq3, pk0, pk1, pk2, pk3, pv0, pv1, pv2, pv3, pd0, pd1, pd2, pd3;
  float plse = 0.f;
#define BWD_ISSUE_ONE(i, ITEM)                                                          \
  if constexpr (FAST && NPF > i) {                                                      \
    int t_ = (ITEM);                                                                    \
    const int wx_ = t_ % g.nwx; t_ /= g.nwx;                                            \
    const int wy_ = t_ % g.nwy; const int b_ = t_ / g.nwy;                              \
    const int idx = tid + i * NT;                                                       \
    const int r = idx / CPR, cc = idx - r * CPR;                                        \
    int row, rid, iy, ix;                                                               \
    win_token(g, b_, wy_, wx_, r, row, rid, iy, ix);                                    \
    const T* src = qkv + (long)row * C3 + (hg * NW) * HD + cc * KPL;                    \
    pq##i = *(const uint4*)(src); pk##i = *(const uint4*)(src + g.C); pv##i = *(const uint4*)(src + 2 * g.C); \
    pd##i = *(const uint4*)(d_out + (long)row * g.C + (hg * NW) * HD + cc * KPL);       \
  }
#define BWD_ISSUE(ITEM) {                                                               \
    BWD_ISSUE_ONE(0, ITEM) BWD_ISSUE_ONE(1, ITEM) BWD_ISSUE_ONE(2, ITEM) BWD_ISSUE_ONE(3, ITEM) \
    int t2_ = (ITEM);                                                                   \
    const int wx2_ = t2_ % g.nwx; t2_ /= g.nwx;                                         \
    const int wy2_ = t2_ % g.nwy; const int b2_ = t2_ / g.nwy;                          \
    int row2, rid2, iy2, ix2;                                                           \
    win_token(g, b2_, wy2_, wx2_, lane, row2, rid2, iy2, ix2);                          \
    plse = lse[(long)row2 * g.heads + head];                                            \
  }
#define BWD_STORE_ONE(i)                                                                \
  if constexpr (FAST && NPF > i) {                                                      \
    const int idx = tid + i * NT;                                                       \
    const int r = idx / CPR, cc = idx - r * CPR;                                        \
    const int h = cc / L::DCH, dc = cc - h * L::DCH;                                    \
    const int off = (h * 64 + r) * L::QROW + dc * 16;                                   \
    *(uint4*)(sQ + off) = pq##i; *(uint4*)(sK + off) = pk##i; *(uint4*)(sV + off) = pv##i; *(uint4*)(sDO + off) = pd##i; \
  }
  // per-lane bias values (x log2 e) of its 64 (q, key) positions: identical for every window of this head
  float bias2[4][4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) bias2[a][c][r] = 0.f;
  if constexpr (pf) {
#pragma unroll
    for (int ms = 0; ms < 4; ++ms)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qn = ms * 16 + fg * 4 + r;
        const int qiy = qn / g.ws, qix = qn - qiy * g.ws;
#pragma unroll
        for (int ns = 0; ns < 4; ++ns) {
          const int kn = ns * 16 + fr;
          const int kiy_ = kn / g.ws, kix_ = kn - kiy_ * g.ws;
          bias2[ms][ns][r] = bt[(qiy - kiy_ + g.ws - 1) * L2 + (qix - kix_ + g.ws - 1)] * SODT_LOG2E;
        }
      }
    BWD_ISSUE((int)blockIdx.x < nitems ? (int)blockIdx.x : 0)
  }
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    int t = item;
    const int kt = t % g.nqt; t /= g.nqt;
    const int wx = t % g.nwx; t /= g.nwx;
    const int wy = t % g.nwy; const int b = t / g.nwy;

    __syncthreads();
    if (tid < 64) {
      int row, rid, iy, ix;
      win_token(g, b, wy, wx, kt * 64 + tid, row, rid, iy, ix);
      sTokK[tid] = row; sGeoK[tid][0] = (short)iy; sGeoK[tid][1] = (short)ix; sGeoK[tid][2] = (short)rid;
    }
    if constexpr (pf) {
      BWD_STORE_ONE(0) BWD_STORE_ONE(1) BWD_STORE_ONE(2) BWD_STORE_ONE(3)
      sLse[w][lane] = plse * SODT_LOG2E;
    }
    __syncthreads();
    if constexpr (pf) {
      {   // unconditional (clamped) so that the prefetch registers stay registers
        const int nxt = item + (int)gridDim.x < nitems ? item + (int)gridDim.x : item;
        BWD_ISSUE(nxt)
      }
    } else {
      for (int idx = tid; idx < 64 * CPR; idx += NT) {
        const int r = idx / CPR, cc = idx - r * CPR;
        const int h = cc / L::DCH, dc = cc - h * L::DCH;
        const T* src = qkv + (long)sTokK[r] * C3 + (hg * NW) * HD + cc * KPL;
        *(uint4*)(sK + (h * 64 + r) * L::QROW + dc * 16) = *(const uint4*)(src + g.C);
        *(uint4*)(sV + (h * 64 + r) * L::QROW + dc * 16) = *(const uint4*)(src + 2 * g.C);
      }
    }
    f32x4 dk[4][HD / 16], dv[4][HD / 16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int d = 0; d < HD / 16; ++d) { dk[i][d] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[i][d] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    for (int qt = 0; qt < g.nqt; ++qt) {
      __syncthreads();
      if (tid < 64) {
        int row, rid, iy, ix;
        win_token(g, b, wy, wx, qt * 64 + tid, row, rid, iy, ix);
        sTokQ[tid] = row; sGeoQ[tid][0] = (short)iy; sGeoQ[tid][1] = (short)ix; sGeoQ[tid][2] = (short)rid;
      }
      __syncthreads();
      const int dyoff = (qt - kt) * R;
      if constexpr (!pf) {
        for (int i = lane; i < LT; i += 64) {
          const int a = i / L2, c = i - a * L2;
          const int gy = a - (R - 1) + dyoff + g.ws - 1;
          sBias[w][i] = (gy >= 0 && gy < L2) ? bt[gy * L2 + c] : 0.f;
        }
        const long tq = sTokQ[lane];
        sLse[w][lane] = lse[tq * g.heads + head] * SODT_LOG2E;
        sDelta[w][lane] = single ? 0.f : delta[tq * g.heads + head];
        for (int idx = tid; idx < 64 * CPR; idx += NT) {
          const int r = idx / CPR, cc = idx - r * CPR;
          const int h = cc / L::DCH, dc = cc - h * L::DCH;
          *(uint4*)(sQ + (h * 64 + r) * L::QROW + dc * 16) = *(const uint4*)(qkv + (long)sTokQ[r] * C3 + (hg * NW) * HD + cc * KPL);
          *(uint4*)(sDO + (h * 64 + r) * L::QROW + dc * 16) = *(const uint4*)(d_out + (long)sTokQ[r] * g.C + (hg * NW) * HD + cc * KPL);
        }
      }
      __syncthreads();

      // ---- phase A: per 16-query strip  S, dP -> P, dS ; P^T and dS^T to LDS (packed 4-q stores)
      int kiy[4], kix[4], krid[4];
#pragma unroll
      for (int ns = 0; ns < 4; ++ns) {
        kiy[ns] = sGeoK[ns * 16 + fr][0]; kix[ns] = sGeoK[ns * 16 + fr][1]; krid[ns] = sGeoK[ns * 16 + fr][2];
      }
      // only windows in the last window row / column see more than one mask region (backbone_vit.py:1061-1072)
      const bool msk = g.shift > 0 && (wy == g.nwy - 1 || wx == g.nwx - 1);
      const float scale2 = scale * SODT_LOG2E;
#pragma unroll
      for (int ms = 0; ms < 4; ++ms) {
        f32x4 s[4], dp[4];
#pragma unroll
        for (int kb = 0; kb < L::KBQ; ++kb) {
          const uint4 fq = frag<T>(myQ, L::QROW, ms * 16, kb, HD, lane);
          const uint4 fo = frag<T>(myDO, L::QROW, ms * 16, kb, HD, lane);
#pragma unroll
          for (int ns = 0; ns < 4; ++ns) {
            const uint4 fk = frag<T>(myK, L::QROW, ns * 16, kb, HD, lane);
            const uint4 fv = frag<T>(myV, L::QROW, ns * 16, kb, HD, lane);
            if (kb == 0) { s[ns] = mma16z<T>(fq, fk); dp[ns] = mma16z<T>(fo, fv); }
            else { mma16<T>(s[ns], fq, fk); mma16<T>(dp[ns], fo, fv); }
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qn = ms * 16 + fg * 4 + r;
          const float lq = sLse[w][qn];
          int qiy = 0, qix = 0, qrid = 0;
          if (!pf || msk) { qiy = sGeoQ[qn][0]; qix = sGeoQ[qn][1]; qrid = sGeoQ[qn][2]; }
          (void)qiy; (void)qix;
          float dl = 0.f;
#pragma unroll
          for (int ns = 0; ns < 4; ++ns) {
            float bv;
            if constexpr (pf) bv = bias2[ms][ns][r];
            else bv = sBias[w][(qiy - kiy[ns] - dyoff + R - 1) * L2 + (qix - kix[ns] + g.ws - 1)] * SODT_LOG2E;
            float v = fmaf(s[ns][r], scale2, bv);
            if (msk && qrid != krid[ns]) v += -100.0f * SODT_LOG2E;
            const float p = fast_exp2(v - lq);
            s[ns][r] = p;
            dl = fmaf(p, dp[ns][r], dl);
          }
          if (single) dl = group16_sum(dl);
          else dl = sDelta[w][qn];
#pragma unroll
          for (int ns = 0; ns < 4; ++ns) {
            const float ds = s[ns][r] * (dp[ns][r] - dl);
            dp[ns][r] = ds;
            dbacc[ms][ns][r] += ds;
          }
        }
#pragma unroll
        for (int ns = 0; ns < 4; ++ns) {
          st4<T>(myPT + (ns * 16 + fr) * L::TROW + (ms * 16 + fg * 4) * E, s[ns]);
          st4<T>(myDST + (ns * 16 + fr) * L::TROW + (ms * 16 + fg * 4) * E, dp[ns]);
        }
      }
      __syncthreads();

      // ---- phase B: dV += P^T dO ; dK += dS^T Q ; dQ = dS K   (contraction over LDS rows -> fragT)
      f32x4 dq[4][HD / 16];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int d = 0; d < HD / 16; ++d) dq[i][d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < L::KBT; ++kb) {
        uint4 fdo[HD / 16], fqq[HD / 16], fkk[HD / 16];
#pragma unroll
        for (int d = 0; d < HD / 16; ++d) {
          fdo[d] = fragT<T>(myDO, L::QROW, kb * MK, d * 16, lane);
          fqq[d] = fragT<T>(myQ, L::QROW, kb * MK, d * 16, lane);
          fkk[d] = fragT<T>(myK, L::QROW, kb * MK, d * 16, lane);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const uint4 fp = frag<T>(myPT, L::TROW, ks * 16, kb, 64, lane);
          const uint4 fs = frag<T>(myDST, L::TROW, ks * 16, kb, 64, lane);
          const uint4 fst = fragT<T>(myDST, L::TROW, kb * MK, ks * 16, lane);    // dS[q = ks strip][key block kb]
#pragma unroll
          for (int d = 0; d < HD / 16; ++d) {
            mma16<T>(dv[ks][d], fp, fdo[d]);
            mma16<T>(dk[ks][d], fs, fqq[d]);
            mma16<T>(dq[ks][d], fst, fkk[d]);
          }
        }
      }
      __syncthreads();   // every wave is done reading sQ / sDO / sPT / sDST of this pair
      if (single) {
#pragma unroll
        for (int ms = 0; ms < 4; ++ms)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int d = 0; d < HD / 16; ++d)
              st_elem<T>(myQ + (ms * 16 + fg * 4 + r) * L::QROW + (d * 16 + fr) * E, dq[ms][d][r] * scale);
        __syncthreads();
        for (int idx = tid; idx < 64 * CPR; idx += NT) {
          const int r = idx / CPR, cc = idx - r * CPR;
          const int h = cc / L::DCH, dc = cc - h * L::DCH;
          *(uint4*)(dqkv + (long)sTokQ[r] * C3 + (hg * NW) * HD + cc * KPL) = *(const uint4*)(sQ + (h * 64 + r) * L::QROW + dc * 16);
        }
      } else {
#pragma unroll
        for (int ms = 0; ms < 4; ++ms)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const long tq = sTokQ[ms * 16 + fg * 4 + r];
#pragma unroll
            for (int d = 0; d < HD / 16; ++d)
              atomicAdd(dq_acc + tq * g.C + head * HD + d * 16 + fr, dq[ms][d][r] * scale);
          }
        // the (q tile, kv tile) offset changes with every pair: reduce the register sums into the LDS table now
#pragma unroll
        for (int ms = 0; ms < 4; ++ms)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int qn = ms * 16 + fg * 4 + r;
            const int qiy = sGeoQ[qn][0], qix = sGeoQ[qn][1];
#pragma unroll
            for (int ns = 0; ns < 4; ++ns) {
              if (fulldb) atomicAdd(&dbfull[w * L2 * L2 + (qiy - kiy[ns] + g.ws - 1) * L2 + (qix - kix[ns] + g.ws - 1)], dbacc[ms][ns][r]);
              else atomicAdd(&sDB[w][(qiy - kiy[ns] - dyoff + R - 1) * L2 + (qix - kix[ns] + g.ws - 1)], dbacc[ms][ns][r]);
              dbacc[ms][ns][r] = 0.f;
            }
          }
        if (!fulldb) {
          __syncthreads();
          for (int i = lane; i < LT; i += 64) {
            const int a = i / L2, c = i - a * L2;
            const int gy = a - (R - 1) + dyoff + g.ws - 1;
            const float v = sDB[w][i];
            if (gy >= 0 && gy < L2 && v != 0.f) atomicAdd(dbias_t + (long)head * L2 * L2 + gy * L2 + c, v);
            sDB[w][i] = 0.f;
          }
        }
      }
    }
    // ---- dK, dV of this kv tile -> staged through sK / sV -> dqkv[:, C:2C], [:, 2C:3C]
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int d = 0; d < HD / 16; ++d) {
          st_elem<T>(myK + (ks * 16 + fg * 4 + r) * L::QROW + (d * 16 + fr) * E, dk[ks][d][r] * scale);
          st_elem<T>(myV + (ks * 16 + fg * 4 + r) * L::QROW + (d * 16 + fr) * E, dv[ks][d][r]);
        }
    __syncthreads();
    for (int idx = tid; idx < 64 * CPR; idx += NT) {
      const int r = idx / CPR, cc = idx - r * CPR;
      const int h = cc / L::DCH, dc = cc - h * L::DCH;
      T* dst = dqkv + (long)sTokK[r] * C3 + (hg * NW) * HD + cc * KPL;
      *(uint4*)(dst + g.C) = *(const uint4*)(sK + (h * 64 + r) * L::QROW + dc * 16);
      *(uint4*)(dst + 2 * g.C) = *(const uint4*)(sV + (h * 64 + r) * L::QROW + dc * 16);
    }
  }
  if (!single && fulldb) {
    __syncthreads();
    for (int i = lane; i < L2 * L2; i += 64) {
      const float v = dbfull[w * L2 * L2 + i];
      if (v != 0.f) atomicAdd(dbias_t + (long)head * L2 * L2 + i, v);
    }
  }
  if (single) {
    // one reduction for all the windows this wave handled: registers -> LDS table -> global atomics
    // (window-local geometry is identical for every window, so the last sGeoQ / sGeoK are valid)
    __syncthreads();
    if (nitems > (int)blockIdx.x) {
      int kiy[4], kix[4];
#pragma unroll
      for (int ns = 0; ns < 4; ++ns) { kiy[ns] = sGeoK[ns * 16 + fr][0]; kix[ns] = sGeoK[ns * 16 + fr][1]; }
#pragma unroll
      for (int ms = 0; ms < 4; ++ms)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qn = ms * 16 + fg * 4 + r;
          const int qiy = sGeoQ[qn][0], qix = sGeoQ[qn][1];
#pragma unroll
          for (int ns = 0; ns < 4; ++ns)
            atomicAdd(&sDB[w][(qiy - kiy[ns] + R - 1) * L2 + (qix - kix[ns] + g.ws - 1)], dbacc[ms][ns][r]);
        }
    }
    __syncthreads();
    for (int i = lane; i < LT; i += 64) {
      const float v = sDB[w][i];
      if (v != 0.f) atomicAdd(dbias_t + (long)head * L2 * L2 + i, v);
    }
  }
}

// ---------------------------------------------------------------------------------
// backward, 8x8 windows (one 64-token tile per window), register-resident P / dS.
//
// S = Q K^T and dP = dO V^T leave the accumulators with four consecutive QUERIES per lane for one key, which is
// exactly the A-operand layout of the query contractions dV += P^T dO and dK += dS^T Q: two 16-query strips packed
// to bf16 form one 32-deep MFMA operand (the B operands dO / Q are read with the transposing LDS load from rows in
// the same strip-pair order; f32: one strip = one operand).  Only dQ = dS K contracts over keys: dS goes through a
// 2.5 KiB per-wave [key][16 q] patch, one strip at a time.  No 64 x 64 P^T / dS^T tiles: LDS per wave drops from
// 30 KiB to 15 KiB (hd 16) and the phase barriers disappear, so twice as many waves are resident per CU.
// Everything else (persistent walk over the windows of a head group, register prefetch of the next window, bias
// values and bias-gradient sums in registers, staged coalesced stores) is as in attn_bwd_kernel<FAST>.
// ---------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ uint4 fragTp(const unsigned char* tile, int rowbytes, int kbq, int col0, int lane);
template <> __device__ __forceinline__ uint4 fragTp<bf16>(const unsigned char* tile, int rowbytes, int kbq, int col0, int lane) {
  // k-slots j < 4 <-> row 32 kbq + 4 g + j (even strip), j >= 4 <-> row 32 kbq + 16 + 4 g + (j - 4) (odd strip)
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const unsigned char* a = tile + (32 * kbq + 4 * g + q) * rowbytes + (col0 + 4 * p) * 2;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  union { s16x4 v; uint2 u; } lo, hi;
  lo.v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
  hi.v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 16 * rowbytes));
  return make_uint4(lo.u.x, lo.u.y, hi.u.x, hi.u.y);
}
template <> __device__ __forceinline__ uint4 fragTp<float>(const unsigned char* tile, int rowbytes, int kbq, int col0, int lane) {
  return fragT<float>(tile, rowbytes, 16 * kbq, col0, lane);
}

// WM: q / k / v and the log-sum-exp come in the WINDOW-MAJOR layout the fused forward (wmsa_block.hip) saves:
// qkv = [window][head][q|k|v][64 tokens][HD], lse = [window][head][64] - 2 KB contiguous per tensor, window and head.
// RC (bf16, HD 16, NW 4; implies window-major lse): q / k / v are not read at all - `qkv` is the block's saved LayerNorm-1
// output xn1 [M][C] (token-major) and `wpk` its parameter pack (sodt_wmsa_pack): each wave recomputes q^T, k^T, v^T of its head
// for the window by MFMA (72 per window and head) from the window's xn1 tile in LDS and the head's weight fragments streamed
// from L2, exactly as the fused forward computed them, and writes them into the Q / K / V tiles the body reads.  The fused
// forward then saves no q / k / v (604 of its 1,644 MB per stage-1 launch); this kernel reads 201 MB (one pass over xn1 per
// head group: three workgroups walk the same windows, the second and third read is served by L2) instead of 604 MB.
template <typename T, int HD, int NW, bool WM = false, bool RC = false>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void attn_bwd_fast2_kernel(const T* __restrict__ qkv, const float* __restrict__ bias_t,
                                                                const T* __restrict__ d_out, const float* __restrict__ lse,
                                                                T* __restrict__ dqkv, float* __restrict__ dbias_t,
                                                                const AttnGeo g, int nwin_total, const unsigned char* __restrict__ wpk = nullptr) {
  static_assert(!RC || (WM && std::is_same<T, bf16>::value && HD == 16 && NW == 4), "RC: bf16, head dim 16, four heads per workgroup");
  using L = Lay<T, HD>;
  constexpr int E = L::E, KPL = L::KPL, NT = NW * 64, CPR = NW * L::DCH, NPF = L::DCH, MK = TT<T>::MMA_K;
  constexpr int SPK = MK / 16;                  // 16-query strips per MFMA k-block: 2 (bf16) or 1 (f32)
  constexpr int DSROW = 16 * E + 16;            // [key][16 q] patch row (bytes)
  constexpr int LTMAX = 225;
  constexpr bool LATE_PF = NW == 4;             // four-head workgroups run two per CU (see the launch bounds)
  __shared__ __attribute__((aligned(16))) unsigned char sQ[NW * L::QTILE], sK[NW * L::QTILE], sV[NW * L::QTILE],
      sDO[NW * L::QTILE];
  // dS patches | dQ staging (Q stays live to the end).  RC: the window's xn1 tile ([64][C] bf16, 16-byte chunks XOR-swizzled
  // with row & 7) lives in the same bytes while q / k / v are recomputed, before the first dS patch is written
  __shared__ __attribute__((aligned(16))) unsigned char sDSQ[NW * 64 * DSROW + NW * L::QTILE];
  unsigned char* const sDS = sDSQ;
  unsigned char* const sDQ = sDSQ + NW * 64 * DSROW;
  static_assert(!RC || NW * 64 * DSROW + NW * L::QTILE >= 64 * 192 * 2, "xn1 tile fits the dS / dQ area");
  __shared__ float sDB[NW][LTMAX + 3];       // bias values x log2 e during the walk, bias-gradient table at the end
  __shared__ float sLse[NW][64];
  __shared__ int sTokQ[64];
  __shared__ short sGeoQ[64][4];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  // Workgroup -> (window walker bx of gxd, head group hg).  RC: a 1-D grid of 3 gxd workgroups, gxd % 8 == 0; the three head groups
  // of a walker read the SAME xn1 rows, so they sit on one XCD (linear id % 8) at consecutive positions and the second and third
  // read hit that XCD's L2 (with the (gx, 3) grid they sat 170 ids apart on other XCDs: pmc r04, 819 MB fetched for 427)
  int bx, gxd, hg;
  if constexpr (RC) {
    const int id = blockIdx.x, k = id >> 3;
    hg = k % 3; bx = (k / 3) * 8 + (id & 7); gxd = gridDim.x / 3;
  } else { bx = blockIdx.x; gxd = gridDim.x; hg = blockIdx.y; }
  const int head = hg * NW + w;
  const int C3 = 3 * g.C, L2 = 2 * g.ws - 1;
  const int R = 64 / g.ws, LT = (2 * R - 1) * L2;
  const float scale = rsqrtf((float)HD), scale2 = scale * SODT_LOG2E;
  // RC: the Q tile holds q_s = scale2 x q, so the logits need no factor, dK = dS^T q_s / log2 e, and dQ (the gradient of the
  // UNSCALED q, what the dWqkv / dxn1 GEMMs expect) = scale x dS K as always
  const float sc_s = RC ? 1.0f : scale2, sc_dk = RC ? (1.0f / SODT_LOG2E) : scale;
  const float* bt = bias_t + (long)head * L2 * L2;
  unsigned char* myQ = sQ + w * L::QTILE; unsigned char* myK = sK + w * L::QTILE;
  unsigned char* myV = sV + w * L::QTILE; unsigned char* myDO = sDO + w * L::QTILE;
  unsigned char* myDS = sDS + w * 64 * DSROW;
  unsigned char* myDQ = sDQ + w * L::QTILE;

  // 8x8 windows: a 16-token strip is two window rows, so the table entry of (q, key) depends on the strips only
  // through ms - ns:  qy - ky = 2 (ms - ns) + ((4 fg + r) >> 3) - (fr >> 3),  qx - kx = ((4 fg + r) & 7) - (fr & 7).
  // Bias values and bias-gradient sums are therefore kept per strip DIFFERENCE (7 x 4 registers each, not 64).
  f32x4 dbc[7];
#pragma unroll
  for (int dc = 0; dc < 7; ++dc) dbc[dc] = f32x4{0.f, 0.f, 0.f, 0.f};
  // bias values (x log2 e) of this head in LDS; lane (fg, fr) / accumulator row r reads entry (dy + 7) * 15 + dx + 7 with
  // dy = 2 (ms - ns) + ((4 fg + r) >> 3) - (fr >> 3), dx = ((4 fg + r) & 7) - (fr & 7): one base address per r, the strip
  // difference is an immediate offset of 120 bytes per step
  for (int i = lane; i < LT; i += 64) sDB[w][i] = bt[i] * SODT_LOG2E;
  uint32_t bAddr[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int dy0 = -6 + ((4 * fg + r) >> 3) - (fr >> 3), dx = ((4 * fg + r) & 7) - (fr & 7);   // strip difference -3
    bAddr[r] = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&sDB[w][(dy0 + 7) * L2 + dx + 7];
  }
  uint4 pq0, pq1, pq2, pq3, pk0, pk1, pk2, pk3, pv0, pv1, pv2, pv3, pd0, pd1, pd2, pd3;
  uint4 px0, px1, px2, px3, px4, px5;        // RC: this thread's six 16-byte chunks of the next window's xn1 tile
  float plse = 0.f;
  constexpr int XCH = 24;                    // RC: 16-byte chunks per xn1 row (C = 192 bf16)
  // RC: thread (r = tid >> 3, c8 = tid & 7) fetches chunks c8, c8 + 8, c8 + 16 of tile rows r and r + 32: eight lanes read 128
  // contiguous bytes of a token row (whole cache lines), the chunk step is an immediate, and the only per-thread index
  // state is two token rows per window (six independent (row, chunk) pairs per thread cost ~40 loop-invariant address
  // registers and put the kernel over its 256-register budget)
#define B2_ISSUE_XALL(ITEM)                                                             \
  {                                                                                     \
    int t_ = (ITEM);                                                                    \
    const int wx_ = t_ % g.nwx; t_ /= g.nwx;                                            \
    const int wy_ = t_ % g.nwy; const int b_ = t_ / g.nwy;                              \
    int tz_ = tid; asm volatile("" : "+v"(tz_));      /* (not loop-invariant for the compiler: re-derived, not spilled) */ \
    const int r_ = tz_ >> 3;                                                            \
    int ya = wy_ * 8 + (r_ >> 3) + g.shift, xa = wx_ * 8 + (r_ & 7) + g.shift;          \
    int yb = ya + 4;                                                                    \
    if (ya >= g.H) ya -= g.H;                                                           \
    if (yb >= g.H) yb -= g.H;                                                           \
    if (xa >= g.W) xa -= g.W;                                                           \
    const T* sa_ = qkv + ((long)(b_ * g.H + ya) * g.W + xa) * g.C + (tz_ & 7) * KPL;    \
    const T* sb_ = qkv + ((long)(b_ * g.H + yb) * g.W + xa) * g.C + (tz_ & 7) * KPL;    \
    px0 = *(const uint4*)(sa_); px1 = *(const uint4*)(sa_ + 8 * KPL); px2 = *(const uint4*)(sa_ + 16 * KPL); \
    px3 = *(const uint4*)(sb_); px4 = *(const uint4*)(sb_ + 8 * KPL); px5 = *(const uint4*)(sb_ + 16 * KPL); \
  }
#define B2_STORE_XALL()                                                                 \
  {                                                                                     \
    int tz_ = tid; asm volatile("" : "+v"(tz_));                                        \
    unsigned char* d_ = sDSQ + (tz_ >> 3) * (XCH * 16) + ((((tz_ & 7) ^ ((tz_ >> 3) & 7))) << 4);   \
    *(uint4*)(d_) = px0; *(uint4*)(d_ + 128) = px1; *(uint4*)(d_ + 256) = px2;          \
    d_ += 32 * (XCH * 16);                                                              \
    *(uint4*)(d_) = px3; *(uint4*)(d_ + 128) = px4; *(uint4*)(d_ + 256) = px5;          \
  }
#define B2_ISSUE_ONE(i, ITEM)                                                           \
  if constexpr (NPF > i) {                                                              \
    int t_ = (ITEM);                                                                    \
    const int wx_ = t_ % g.nwx; t_ /= g.nwx;                                            \
    const int wy_ = t_ % g.nwy; const int b_ = t_ / g.nwy;                              \
    const int idx = tid + i * NT;                                                       \
    const int r = idx / CPR, cc = idx - r * CPR;                                        \
    int row, rid, iy, ix;                                                               \
    win_token(g, b_, wy_, wx_, r, row, rid, iy, ix);                                    \
    if constexpr (RC) {                                                                 \
    } else if constexpr (WM) {                                                          \
      const int h_ = cc / L::DCH, dc_ = cc - h_ * L::DCH;                               \
      const T* src = qkv + (((long)(ITEM) * g.heads + hg * NW + h_) * 3 * 64 + r) * HD + dc_ * KPL; \
      pq##i = *(const uint4*)(src); pk##i = *(const uint4*)(src + 64 * HD); pv##i = *(const uint4*)(src + 2 * 64 * HD); \
    } else {                                                                            \
      const T* src = qkv + (long)row * C3 + (hg * NW) * HD + cc * KPL;                  \
      pq##i = *(const uint4*)(src); pk##i = *(const uint4*)(src + g.C); pv##i = *(const uint4*)(src + 2 * g.C); \
    }                                                                                   \
    pd##i = *(const uint4*)(d_out + (long)row * g.C + (hg * NW) * HD + cc * KPL);       \
  }
#define B2_ISSUE(ITEM) {                                                                \
    B2_ISSUE_ONE(0, ITEM) B2_ISSUE_ONE(1, ITEM) B2_ISSUE_ONE(2, ITEM) B2_ISSUE_ONE(3, ITEM) \
    if constexpr (RC) B2_ISSUE_XALL(ITEM)                                               \
    if constexpr (WM) {                                                                 \
      plse = lse[((long)(ITEM) * g.heads + head) * 64 + lane];                          \
    } else {                                                                            \
      int t2_ = (ITEM);                                                                 \
      const int wx2_ = t2_ % g.nwx; t2_ /= g.nwx;                                       \
      const int wy2_ = t2_ % g.nwy; const int b2_ = t2_ / g.nwy;                        \
      int row2, rid2, iy2, ix2;                                                         \
      win_token(g, b2_, wy2_, wx2_, lane, row2, rid2, iy2, ix2);                        \
      plse = lse[(long)row2 * g.heads + head];                                          \
    }                                                                                   \
  }
#define B2_STORE_ONE(i)                                                                 \
  if constexpr (NPF > i) {                                                              \
    const int idx = tid + i * NT;                                                       \
    const int r = idx / CPR, cc = idx - r * CPR;                                        \
    const int h = cc / L::DCH, dc = cc - h * L::DCH;                                    \
    const int off = (h * 64 + r) * L::QROW + dc * 16;                                   \
    if constexpr (!RC) { *(uint4*)(sQ + off) = pq##i; *(uint4*)(sK + off) = pk##i; *(uint4*)(sV + off) = pv##i; } \
    *(uint4*)(sDO + off) = pd##i;                                                       \
  }
  B2_ISSUE(bx < nwin_total ? bx : 0)

  for (int item = bx; item < nwin_total; item += gxd) {
    int t = item;
    const int wx = t % g.nwx; t /= g.nwx;
    const int wy = t % g.nwy; const int b = t / g.nwy;
    const bool msk = g.shift > 0 && (wy == g.nwy - 1 || wx == g.nwx - 1);
    __syncthreads();                                   // previous window's staged outputs are out
    if (tid < 64) {
      int row, rid, iy, ix;
      win_token(g, b, wy, wx, tid, row, rid, iy, ix);
      sTokQ[tid] = row; sGeoQ[tid][0] = (short)iy; sGeoQ[tid][1] = (short)ix; sGeoQ[tid][2] = (short)rid;
    }
    B2_STORE_ONE(0) B2_STORE_ONE(1) B2_STORE_ONE(2) B2_STORE_ONE(3)
    if constexpr (RC) B2_STORE_XALL()
    sLse[w][lane] = plse * SODT_LOG2E;
    __syncthreads();
    if constexpr (RC) {
      // ---- q^T, k^T, v^T of this wave's head (channel rows 4 fg + r, token columns 16 ms + fr) = W_h xn^T + b: the forward's
      // own product (backbone_vit.py:968), weights as A fragments straight from the pack (L2-resident: every workgroup of
      // the head group re-reads the same 72 KB per window), xn^T fragments from the tile
      // (lane-derived addresses from a laundered lane id: as loop invariants they - and the three bias vectors - would be held
      //  across the register-heaviest part of the kernel, i.e. spilled)
      int lz = lane; asm volatile("" : "+v"(lz));
      const int fr_ = lz & 15, fg_ = lz >> 4;
      // The forward's OWN operands: its weight stream, where Wq and the q bias carry hd^-1/2 x log2 e (q_s = that scale x q), so
      // that P = exp2(q_s k + bias - lse) here reproduces the P whose log-sum-exp the forward saved (recomputing an unscaled q
      // from the unscaled weight copy rounds differently: the logits move by up to 0.05 and the rows of P no longer sum to 1)
      const unsigned char* wh = wpk + wmsa_pack_bf16::HGW_OFF + (size_t)head * 3 * wmsa_pack_bf16::WFRAG;
      const float* bqkv = (const float*)(wpk + (size_t)head * wmsa_pack_bf16::STAGE + wmsa_pack_bf16::BQKV_OFF);
      const f32x4 bq4 = *(const f32x4*)(bqkv + 48 + 4 * fg_), bk4 = *(const f32x4*)(bqkv + 16 + 4 * fg_), bv4 = *(const f32x4*)(bqkv + 32 + 4 * fg_);
      f32x4 aq[4], ak[4], av[4];
#pragma unroll
      for (int ms = 0; ms < 4; ++ms) { aq[ms] = bq4; ak[ms] = bk4; av[ms] = bv4; }
      const unsigned char* xl = sDSQ + fr_ * (XCH * 16);
#pragma unroll
      for (int kk = 0; kk < 6; ++kk) {
        const uint4 wq = *(const uint4*)(wh + kk * 1024 + lz * 16);
        const uint4 wk = *(const uint4*)(wh + wmsa_pack_bf16::WFRAG + kk * 1024 + lz * 16);
        const uint4 wv = *(const uint4*)(wh + 2 * wmsa_pack_bf16::WFRAG + kk * 1024 + lz * 16);
#pragma unroll
        for (int ms = 0; ms < 4; ++ms) {
          const uint4 xf = *(const uint4*)(xl + ms * 16 * (XCH * 16) + (((4 * kk + fg_) ^ (fr_ & 7)) << 4));
          mma16<bf16>(aq[ms], wq, xf); mma16<bf16>(ak[ms], wk, xf); mma16<bf16>(av[ms], wv, xf);
        }
      }
      // -> the [64 tokens][HD] tiles of this head (this wave's own: no barrier needed before its reads)
#pragma unroll
      for (int ms = 0; ms < 4; ++ms) {
        const int off = (ms * 16 + fr_) * L::QROW + 8 * fg_;
        *(uint2*)(myQ + off) = make_uint2(pack2bf(aq[ms][0], aq[ms][1]), pack2bf(aq[ms][2], aq[ms][3]));
        *(uint2*)(myK + off) = make_uint2(pack2bf(ak[ms][0], ak[ms][1]), pack2bf(ak[ms][2], ak[ms][3]));
        *(uint2*)(myV + off) = make_uint2(pack2bf(av[ms][0], av[ms][1]), pack2bf(av[ms][2], av[ms][3]));
      }
      __syncthreads();                                 // every wave is done with the xn1 tile: the dS patches may overwrite it
    }
    if constexpr (!LATE_PF) {
      const int nxt = item + gxd < nwin_total ? item + gxd : item;
      B2_ISSUE(nxt)
    }

    f32x4 dk[4][HD / 16], dv[4][HD / 16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int d = 0; d < HD / 16; ++d) { dk[i][d] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[i][d] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    // the window's strips, compiled twice: only windows in the last window row / column of a shifted block see more than
    // one mask region (backbone_vit.py:1061-1072); a run-time test inside the softmax is if-converted by hipcc into
    // 64 compares + 64 selects + 64 ors per window for everyone
    auto strips = [&](auto MSK_) {
    constexpr bool MSK = decltype(MSK_)::value;
    int krid[4];
#pragma unroll
    for (int ns = 0; ns < 4; ++ns) krid[ns] = MSK ? (int)sGeoQ[ns * 16 + fr][2] : 0;
    // the K fragments of S = Q K^T do not depend on the strip: read once per window (16 registers; the four V fragments would be the
    // next 16, which the kernel does not have).  Same box, interleaved: 0.512-0.520 / 0.482-0.483 -> 0.488-0.497 / 0.459-0.461 ms.
    uint4 fkh[4];
    if constexpr (L::KBQ == 1) {
#pragma unroll
      for (int ns = 0; ns < 4; ++ns) fkh[ns] = frag<T>(myK, L::QROW, ns * 16, 0, HD, lane);
    }
#pragma unroll
    for (int kbq = 0; kbq < 4 / SPK; ++kbq) {
      // B operands of the query contractions for this k-block (rows = the strip pair, transposed read)
      uint4 fdo[HD / 16], fqq[HD / 16];
#pragma unroll
      for (int d = 0; d < HD / 16; ++d) {
        fdo[d] = fragTp<T>(myDO, L::QROW, kbq, d * 16, lane);
        fqq[d] = fragTp<T>(myQ, L::QROW, kbq, d * 16, lane);
      }
#pragma unroll
      for (int hh = 0; hh < SPK; ++hh) {
        const int ms = kbq * SPK + hh;
        f32x4 s[4], dp[4];
#pragma unroll
        for (int kb = 0; kb < L::KBQ; ++kb) {
          const uint4 fq = frag<T>(myQ, L::QROW, ms * 16, kb, HD, lane);
          const uint4 fo = frag<T>(myDO, L::QROW, ms * 16, kb, HD, lane);
#pragma unroll
          for (int ns = 0; ns < 4; ++ns) {
            const uint4 fk = (L::KBQ == 1) ? fkh[ns] : frag<T>(myK, L::QROW, ns * 16, kb, HD, lane);
            const uint4 fv = frag<T>(myV, L::QROW, ns * 16, kb, HD, lane);
            if (kb == 0) { s[ns] = mma16z<T>(fq, fk); dp[ns] = mma16z<T>(fo, fv); }
            else { mma16<T>(s[ns], fq, fk); mma16<T>(dp[ns], fo, fv); }
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qn = ms * 16 + fg * 4 + r;
          const float lq = sLse[w][qn];
          const int qrid = MSK ? (int)sGeoQ[qn][2] : 0;
          float dl = 0.f;
#pragma unroll
          for (int ns = 0; ns < 4; ++ns) {
            float v = fmaf(s[ns][r], sc_s, *(const float*)((const __attribute__((address_space(3))) char*)(uintptr_t)bAddr[r] + 120 * (ms - ns + 3)));
            if constexpr (MSK) { if (qrid != krid[ns]) v += -100.0f * SODT_LOG2E; }
            const float p = fast_exp2(v - lq);
            s[ns][r] = p;
            dl = fmaf(p, dp[ns][r], dl);
          }
          dl = group16_sum(dl);
#pragma unroll
          for (int ns = 0; ns < 4; ++ns) {
            const float ds = s[ns][r] * (dp[ns][r] - dl);
            dp[ns][r] = ds;
            dbc[ms - ns + 3][r] += ds;
          }
        }
        // this strip as the A operand of the query contractions: a bf16 strip is one 16-deep half of the 32-deep operand (k-slots
        // 0-3 even strip, 4-7 odd strip) - the K = 16 MFMA on that half and the matching half of the B fragment (round 5; before:
        // the 32-deep MFMA with zeros in the other half, 160 v_mov 0 per window and head).  Twice the MFMAs of a packed pair
        // (still < 2 % of the kernel), no operand registers held across strips.  dS strip -> [key][16 q] patch for dQ.
#pragma unroll
        for (int ns = 0; ns < 4; ++ns) {
          uint4 ap, ads;
          if constexpr (std::is_same<T, bf16>::value) {
            const uint32_t p01 = pack2bf(s[ns][0], s[ns][1]), p23 = pack2bf(s[ns][2], s[ns][3]);
            const uint32_t d01 = pack2bf(dp[ns][0], dp[ns][1]), d23 = pack2bf(dp[ns][2], dp[ns][3]);
            ap = hh == 0 ? make_uint4(p01, p23, 0u, 0u) : make_uint4(0u, 0u, p01, p23);
            ads = hh == 0 ? make_uint4(d01, d23, 0u, 0u) : make_uint4(0u, 0u, d01, d23);
            *(uint2*)(myDS + (ns * 16 + fr) * DSROW + 8 * fg) = make_uint2(d01, d23);
          } else {
            ap = make_uint4(__float_as_uint(s[ns][0]), __float_as_uint(s[ns][1]), __float_as_uint(s[ns][2]), __float_as_uint(s[ns][3]));
            ads = make_uint4(__float_as_uint(dp[ns][0]), __float_as_uint(dp[ns][1]), __float_as_uint(dp[ns][2]), __float_as_uint(dp[ns][3]));
            *(float4*)(myDS + (ns * 16 + fr) * DSROW + 16 * fg) = make_float4(dp[ns][0], dp[ns][1], dp[ns][2], dp[ns][3]);
          }
#pragma unroll
          for (int d = 0; d < HD / 16; ++d) {
            if constexpr (std::is_same<T, bf16>::value) {
              // the strip fills one 16-deep half of the operand: the K = 16 MFMA on that half (no zero registers, no v_mov 0)
              typedef __attribute__((ext_vector_type(4))) short s16x4_;
              union { uint2 u; s16x4_ v; } a1, a2, b1, b2;
              a1.u = hh == 0 ? make_uint2(ap.x, ap.y) : make_uint2(ap.z, ap.w);
              a2.u = hh == 0 ? make_uint2(ads.x, ads.y) : make_uint2(ads.z, ads.w);
              b1.u = hh == 0 ? make_uint2(fdo[d].x, fdo[d].y) : make_uint2(fdo[d].z, fdo[d].w);
              b2.u = hh == 0 ? make_uint2(fqq[d].x, fqq[d].y) : make_uint2(fqq[d].z, fqq[d].w);
              // (operands swapped: the accumulators hold dV^T / dK^T - four consecutive CHANNELS of one key per lane - so that the
              //  staging below is one 8-byte LDS write per tile instead of four 2-byte ones)
              dv[ns][d] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(b1.v, a1.v, dv[ns][d], 0, 0, 0);
              dk[ns][d] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(b2.v, a2.v, dk[ns][d], 0, 0, 0);
            } else {
              mma16<T>(dv[ns][d], ap, fdo[d]);
              mma16<T>(dk[ns][d], ads, fqq[d]);
            }
          }
        }
        // dQ strip ms = dS K: A = dS[q = fr][keys] read transposed from the patch (same wave: LDS ops are in order),
        // B = K read transposed from its tile; finished strips go straight to the dQ staging tile
        f32x4 dq[HD / 16];
#pragma unroll
        for (int kb = 0; kb < L::KBT; ++kb) {
          const uint4 fst = fragT<T>(myDS, DSROW, kb * MK, 0, lane);
#pragma unroll
          for (int d = 0; d < HD / 16; ++d) {
            const uint4 fkk = fragT<T>(myK, L::QROW, kb * MK, d * 16, lane);
            if constexpr (std::is_same<T, bf16>::value) {      // dQ^T = K^T dS^T: four consecutive channels of one query per lane
              if (kb == 0) dq[d] = mma16z<T>(fkk, fst); else mma16<T>(dq[d], fkk, fst);
            } else {
              if (kb == 0) dq[d] = mma16z<T>(fst, fkk); else mma16<T>(dq[d], fst, fkk);
            }
          }
        }
        if constexpr (std::is_same<T, bf16>::value) {
#pragma unroll
          for (int d = 0; d < HD / 16; ++d)
            *(uint2*)(myDQ + (ms * 16 + fr) * L::QROW + (d * 16 + 4 * fg) * E) =
                make_uint2(pack2bf(dq[d][0] * scale, dq[d][1] * scale), pack2bf(dq[d][2] * scale, dq[d][3] * scale));
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int d = 0; d < HD / 16; ++d)
              st_elem<T>(myDQ + (ms * 16 + fg * 4 + r) * L::QROW + (d * 16 + fr) * E, dq[d][r] * scale);
        }
      }
    }
    };
    if (msk) strips(std::true_type{}); else strips(std::false_type{});
    if constexpr (LATE_PF) {   // next window's Q / K / V / dO chunks + lse: issued after the strips so the 33 prefetch
        // registers are not live across the register-heaviest part of the kernel (234 VGPRs: two workgroups per CU, no
        // AGPR copies); they land under the staging / store phase and the other workgroup's compute
      const int nxt = item + gxd < nwin_total ? item + gxd : item;
      B2_ISSUE(nxt)
    }
    // ---- stage dQ / dK / dV through this head's own Q / K / V tiles, then coalesced stores
    if constexpr (std::is_same<T, bf16>::value) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int d = 0; d < HD / 16; ++d) {
          const int off = (i * 16 + fr) * L::QROW + (d * 16 + 4 * fg) * E;        // (transposed accumulators: key fr, channels 4 fg ..)
          *(uint2*)(myK + off) = make_uint2(pack2bf(dk[i][d][0] * sc_dk, dk[i][d][1] * sc_dk), pack2bf(dk[i][d][2] * sc_dk, dk[i][d][3] * sc_dk));
          *(uint2*)(myV + off) = make_uint2(pack2bf(dv[i][d][0], dv[i][d][1]), pack2bf(dv[i][d][2], dv[i][d][3]));
        }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int d = 0; d < HD / 16; ++d) {
            const int off = (i * 16 + fg * 4 + r) * L::QROW + (d * 16 + fr) * E;
            st_elem<T>(myK + off, dk[i][d][r] * sc_dk);
            st_elem<T>(myV + off, dv[i][d][r]);
          }
    }
    __syncthreads();
    for (int idx = tid; idx < 64 * CPR; idx += NT) {
      const int r = idx / CPR, cc = idx - r * CPR;
      const int h = cc / L::DCH, dc = cc - h * L::DCH;
      T* dst = dqkv + (long)sTokQ[r] * C3 + (hg * NW) * HD + cc * KPL;
      const int off = (h * 64 + r) * L::QROW + dc * 16;
      *(uint4*)(dst) = *(const uint4*)(sDQ + off);
      *(uint4*)(dst + g.C) = *(const uint4*)(sK + off);
      *(uint4*)(dst + 2 * g.C) = *(const uint4*)(sV + off);
    }
  }
  // ---- bias gradient: one reduction for all the windows this wave handled (window-local geometry is the same
  //      for every window): registers -> LDS table -> global atomics
  __syncthreads();
  for (int i = tid; i < NW * (LTMAX + 3); i += NT) (&sDB[0][0])[i] = 0.f;
  __syncthreads();
  if (nwin_total > bx) {
#pragma unroll
    for (int dc = 0; dc < 7; ++dc)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int dy = 2 * (dc - 3) + ((4 * fg + r) >> 3) - (fr >> 3), dx = ((4 * fg + r) & 7) - (fr & 7);
        if (dy > -8 && dy < 8) atomicAdd(&sDB[w][(dy + 7) * L2 + dx + 7], dbc[dc][r]);
      }
  }
  __syncthreads();
  for (int i = lane; i < LT; i += 64) {
    const float v = sDB[w][i];
    if (v != 0.f) atomicAdd(dbias_t + (long)head * L2 * L2 + i, v);
  }
#undef B2_ISSUE_ONE
#undef B2_ISSUE
#undef B2_STORE_ONE
#undef B2_ISSUE_XALL
#undef B2_STORE_XALL
}

// ---------------------------------------------------------------------------------
// backward, windows of more than 64 tokens (16x16, 32x32, 64x64), register-resident P / dS.
// One workgroup = NW waves = NW heads of one (window, 64-key tile); the 64-query tiles of the window are walked with
// the kv tile's dK / dV in registers.  Same operand scheme as attn_bwd_fast2_kernel: P and dS leave the S / dP
// accumulators as (strip pairs of) A operands for dV += P^T dO and dK += dS^T Q; dS goes through a 3 KiB per-wave
// [key][16 q] patch one strip at a time for dQ = dS K, whose finished strips are added to the f32 dQ accumulator
// straight from the accumulators.  No P^T / dS^T tiles and no full bias-gradient table in LDS: 44 KiB per
// workgroup at hd 64 (was 59 + 16), three workgroups per CU.  delta = rowsum(dO o O) comes from attn_delta_kernel.
// ---------------------------------------------------------------------------------
template <typename T, int HD, int NW>
__global__ __launch_bounds__(NW * 64) void attn_bwd_mt_kernel(const T* __restrict__ qkv, const float* __restrict__ bias_t,
                                                             const T* __restrict__ d_out, const float* __restrict__ lse,
                                                             const float* __restrict__ delta, T* __restrict__ dqkv,
                                                             float* __restrict__ dbias_t, float* __restrict__ dq_acc,
                                                             const AttnGeo g, int nwin_total) {
  using L = Lay<T, HD>;
  constexpr int E = L::E, KPL = L::KPL, NT = NW * 64, CPR = NW * L::DCH, MK = TT<T>::MMA_K;
  constexpr int SPK = MK / 16;
  constexpr int DSROW = 16 * E + 16;
  constexpr int LTMAX = 225;
  __shared__ __attribute__((aligned(16))) unsigned char sQ[NW * L::QTILE], sK[NW * L::QTILE], sV[NW * L::QTILE],
      sDO[NW * L::QTILE];
  __shared__ __attribute__((aligned(16))) unsigned char sDS[NW * 64 * DSROW];
  __shared__ float sDB[NW][LTMAX + 3];
  __shared__ float sBias[NW][LTMAX + 3];
  __shared__ float sLse[NW][64], sDelta[NW][64];
  __shared__ int sTokQ[64], sTokK[64];
  __shared__ short sGeoQ[64][4], sGeoK[64][4];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int hg = blockIdx.y, head = hg * NW + w;
  const int C3 = 3 * g.C, L2 = 2 * g.ws - 1;
  const int R = g.ws >= 64 ? 1 : 64 / g.ws;
  const int LT = (2 * R - 1) * L2;
  const float scale = rsqrtf((float)HD), scale2 = scale * SODT_LOG2E;
  const float* bt = bias_t + (long)head * L2 * L2;
  unsigned char* myQ = sQ + w * L::QTILE; unsigned char* myK = sK + w * L::QTILE;
  unsigned char* myV = sV + w * L::QTILE; unsigned char* myDO = sDO + w * L::QTILE;
  unsigned char* myDS = sDS + w * 64 * DSROW;
  for (int i = tid; i < NW * (LTMAX + 3); i += NT) (&sDB[0][0])[i] = 0.f;

  // the next query tile (Q, dO rows + lse, delta) is fetched into registers under the current pair's MFMAs; named scalars:
  // hipcc leaves uint4 arrays used like this in scratch
  constexpr int NPF = 64 * CPR / NT;            // chunks per thread and tensor (= DCH)
  constexpr bool PF = NPF <= 8;                 // f32 at hd 64 (16 chunks) loads each tile where it is used instead
  uint4 pq0, pq1, pq2, pq3, pq4, pq5, pq6, pq7, pd0, pd1, pd2, pd3, pd4, pd5, pd6, pd7;
  float plse = 0.f, pdel = 0.f;
#define MT_ISSUE_ONE(i, B_, WY_, WX_, QT_)                                              \
  if constexpr (PF && NPF > i) {                                                        \
    const int idx = tid + i * NT;                                                       \
    const int r = idx / CPR, cc = idx - r * CPR;                                        \
    int row, rid, iy, ix;                                                               \
    win_token(g, B_, WY_, WX_, (QT_) * 64 + r, row, rid, iy, ix);                       \
    pq##i = *(const uint4*)(qkv + (long)row * C3 + (hg * NW) * HD + cc * KPL);          \
    pd##i = *(const uint4*)(d_out + (long)row * g.C + (hg * NW) * HD + cc * KPL);       \
  }
#define MT_ISSUE(B_, WY_, WX_, QT_) {                                                   \
    MT_ISSUE_ONE(0, B_, WY_, WX_, QT_) MT_ISSUE_ONE(1, B_, WY_, WX_, QT_) MT_ISSUE_ONE(2, B_, WY_, WX_, QT_) \
    MT_ISSUE_ONE(3, B_, WY_, WX_, QT_) MT_ISSUE_ONE(4, B_, WY_, WX_, QT_) MT_ISSUE_ONE(5, B_, WY_, WX_, QT_) \
    MT_ISSUE_ONE(6, B_, WY_, WX_, QT_) MT_ISSUE_ONE(7, B_, WY_, WX_, QT_)               \
    int row2, rid2, iy2, ix2;                                                           \
    win_token(g, B_, WY_, WX_, (QT_) * 64 + lane, row2, rid2, iy2, ix2);                \
    plse = lse[(long)row2 * g.heads + head]; pdel = delta[(long)row2 * g.heads + head]; \
  }
#define MT_STORE_ONE(i)                                                                 \
  if constexpr (PF && NPF > i) {                                                              \
    const int idx = tid + i * NT;                                                       \
    const int r = idx / CPR, cc = idx - r * CPR;                                        \
    const int h = cc / L::DCH, dc = cc - h * L::DCH;                                    \
    const int off = (h * 64 + r) * L::QROW + dc * 16;                                   \
    *(uint4*)(sQ + off) = pq##i; *(uint4*)(sDO + off) = pd##i;                          \
  }
  const int nitems = nwin_total * g.nqt;
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    int t = item;
    const int kt = t % g.nqt; t /= g.nqt;
    const int wx = t % g.nwx; t /= g.nwx;
    const int wy = t % g.nwy; const int b = t / g.nwy;
    const bool msk = g.shift > 0 && (wy == g.nwy - 1 || wx == g.nwx - 1);
    __syncthreads();
    if (tid < 64) {
      int row, rid, iy, ix;
      win_token(g, b, wy, wx, kt * 64 + tid, row, rid, iy, ix);
      sTokK[tid] = row; sGeoK[tid][0] = (short)iy; sGeoK[tid][1] = (short)ix; sGeoK[tid][2] = (short)rid;
    }
    __syncthreads();
    for (int idx = tid; idx < 64 * CPR; idx += NT) {
      const int r = idx / CPR, cc = idx - r * CPR;
      const int h = cc / L::DCH, dc = cc - h * L::DCH;
      const T* src = qkv + (long)sTokK[r] * C3 + (hg * NW) * HD + cc * KPL;
      *(uint4*)(sK + (h * 64 + r) * L::QROW + dc * 16) = *(const uint4*)(src + g.C);
      *(uint4*)(sV + (h * 64 + r) * L::QROW + dc * 16) = *(const uint4*)(src + 2 * g.C);
    }
    f32x4 dk[4][HD / 16], dv[4][HD / 16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int d = 0; d < HD / 16; ++d) { dk[i][d] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[i][d] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    MT_ISSUE(b, wy, wx, 0)

    for (int qt = 0; qt < g.nqt; ++qt) {
      __syncthreads();
      if (tid < 64) {
        int row, rid, iy, ix;
        win_token(g, b, wy, wx, qt * 64 + tid, row, rid, iy, ix);
        sTokQ[tid] = row; sGeoQ[tid][0] = (short)iy; sGeoQ[tid][1] = (short)ix; sGeoQ[tid][2] = (short)rid;
      }
      __syncthreads();
      const int dyoff = (qt - kt) * R;
      for (int i = lane; i < LT; i += 64) {
        const int a = i / L2, c = i - a * L2;
        const int gy = a - (R - 1) + dyoff + g.ws - 1;
        sBias[w][i] = (gy >= 0 && gy < L2) ? bt[gy * L2 + c] * SODT_LOG2E : 0.f;
      }
      if constexpr (!PF) {
        for (int idx = tid; idx < 64 * CPR; idx += NT) {
          const int r = idx / CPR, cc = idx - r * CPR;
          const int h = cc / L::DCH, dc = cc - h * L::DCH;
          *(uint4*)(sQ + (h * 64 + r) * L::QROW + dc * 16) = *(const uint4*)(qkv + (long)sTokQ[r] * C3 + (hg * NW) * HD + cc * KPL);
          *(uint4*)(sDO + (h * 64 + r) * L::QROW + dc * 16) = *(const uint4*)(d_out + (long)sTokQ[r] * g.C + (hg * NW) * HD + cc * KPL);
        }
      }
      sLse[w][lane] = plse * SODT_LOG2E;
      sDelta[w][lane] = pdel;
      MT_STORE_ONE(0) MT_STORE_ONE(1) MT_STORE_ONE(2) MT_STORE_ONE(3) MT_STORE_ONE(4) MT_STORE_ONE(5) MT_STORE_ONE(6) MT_STORE_ONE(7)
      __syncthreads();
      {   // unconditional (clamped) so that the prefetch registers stay registers
        const int nq_ = qt + 1 < g.nqt ? qt + 1 : qt;
        MT_ISSUE(b, wy, wx, nq_)
      }

      int kiy[4], kix[4], krid[4];
#pragma unroll
      for (int ns = 0; ns < 4; ++ns) {
        kiy[ns] = sGeoK[ns * 16 + fr][0]; kix[ns] = sGeoK[ns * 16 + fr][1]; krid[ns] = sGeoK[ns * 16 + fr][2];
      }
      f32x4 dbacc[4][4];
#pragma unroll
      for (int kbq = 0; kbq < 4 / SPK; ++kbq) {
        uint4 Ap[4], Ads[4];
#pragma unroll
        for (int hh = 0; hh < SPK; ++hh) {
          const int ms = kbq * SPK + hh;
          f32x4 s[4], dp[4];
#pragma unroll
          for (int kb = 0; kb < L::KBQ; ++kb) {
            const uint4 fq = frag<T>(myQ, L::QROW, ms * 16, kb, HD, lane);
            const uint4 fo = frag<T>(myDO, L::QROW, ms * 16, kb, HD, lane);
#pragma unroll
            for (int ns = 0; ns < 4; ++ns) {
              const uint4 fk = frag<T>(myK, L::QROW, ns * 16, kb, HD, lane);
              const uint4 fv = frag<T>(myV, L::QROW, ns * 16, kb, HD, lane);
              if (kb == 0) { s[ns] = mma16z<T>(fq, fk); dp[ns] = mma16z<T>(fo, fv); }
              else { mma16<T>(s[ns], fq, fk); mma16<T>(dp[ns], fo, fv); }
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int qn = ms * 16 + fg * 4 + r;
            const float lq = sLse[w][qn], dl = sDelta[w][qn];
            const int qiy = sGeoQ[qn][0], qix = sGeoQ[qn][1], qrid = sGeoQ[qn][2];
#pragma unroll
            for (int ns = 0; ns < 4; ++ns) {
              float v = fmaf(s[ns][r], scale2, sBias[w][(qiy - kiy[ns] - dyoff + R - 1) * L2 + (qix - kix[ns] + g.ws - 1)]);
              if (msk && qrid != krid[ns]) v += -100.0f * SODT_LOG2E;
              const float p = fast_exp2(v - lq);
              s[ns][r] = p;
              const float ds = p * (dp[ns][r] - dl);
              dp[ns][r] = ds;
            }
          }
#pragma unroll
          for (int ns = 0; ns < 4; ++ns) {
            dbacc[ms][ns] = dp[ns];
            if constexpr (std::is_same<T, bf16>::value) {
              const uint32_t p01 = pack2bf(s[ns][0], s[ns][1]), p23 = pack2bf(s[ns][2], s[ns][3]);
              const uint32_t d01 = pack2bf(dp[ns][0], dp[ns][1]), d23 = pack2bf(dp[ns][2], dp[ns][3]);
              if (hh == 0) { Ap[ns].x = p01; Ap[ns].y = p23; Ads[ns].x = d01; Ads[ns].y = d23; }
              else { Ap[ns].z = p01; Ap[ns].w = p23; Ads[ns].z = d01; Ads[ns].w = d23; }
              *(uint2*)(myDS + (ns * 16 + fr) * DSROW + 8 * fg) = make_uint2(d01, d23);
            } else {
              Ap[ns] = make_uint4(__float_as_uint(s[ns][0]), __float_as_uint(s[ns][1]), __float_as_uint(s[ns][2]), __float_as_uint(s[ns][3]));
              Ads[ns] = make_uint4(__float_as_uint(dp[ns][0]), __float_as_uint(dp[ns][1]), __float_as_uint(dp[ns][2]), __float_as_uint(dp[ns][3]));
              *(float4*)(myDS + (ns * 16 + fr) * DSROW + 16 * fg) = make_float4(dp[ns][0], dp[ns][1], dp[ns][2], dp[ns][3]);
            }
          }
          // dQ strip ms = dS K, added to the f32 accumulator of the window straight from the MFMA accumulators
          f32x4 dq[HD / 16];
#pragma unroll
          for (int kb = 0; kb < L::KBT; ++kb) {
            const uint4 fst = fragT<T>(myDS, DSROW, kb * MK, 0, lane);
#pragma unroll
            for (int d = 0; d < HD / 16; ++d) {
              const uint4 fkk = fragT<T>(myK, L::QROW, kb * MK, d * 16, lane);
              if (kb == 0) dq[d] = mma16z<T>(fst, fkk); else mma16<T>(dq[d], fst, fkk);
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const long tq = sTokQ[ms * 16 + fg * 4 + r];
#pragma unroll
            for (int d = 0; d < HD / 16; ++d)
              atomicAdd(dq_acc + tq * g.C + head * HD + d * 16 + fr, dq[d][r] * scale);
          }
        }
        uint4 fdo[HD / 16], fqq[HD / 16];
#pragma unroll
        for (int d = 0; d < HD / 16; ++d) {
          fdo[d] = fragTp<T>(myDO, L::QROW, kbq, d * 16, lane);
          fqq[d] = fragTp<T>(myQ, L::QROW, kbq, d * 16, lane);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int d = 0; d < HD / 16; ++d) {
            mma16<T>(dv[ks][d], Ap[ks], fdo[d]);
            mma16<T>(dk[ks][d], Ads[ks], fqq[d]);
          }
      }
      // bias gradient of this (q tile, kv tile) pair: registers -> per-wave LDS table -> global atomics
#pragma unroll
      for (int ms = 0; ms < 4; ++ms)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qn = ms * 16 + fg * 4 + r;
          const int qiy = sGeoQ[qn][0], qix = sGeoQ[qn][1];
#pragma unroll
          for (int ns = 0; ns < 4; ++ns)
            atomicAdd(&sDB[w][(qiy - kiy[ns] - dyoff + R - 1) * L2 + (qix - kix[ns] + g.ws - 1)], dbacc[ms][ns][r]);
        }
      __syncthreads();
      for (int i = lane; i < LT; i += 64) {
        const int a = i / L2, c = i - a * L2;
        const int gy = a - (R - 1) + dyoff + g.ws - 1;
        const float v = sDB[w][i];
        if (gy >= 0 && gy < L2 && v != 0.f) atomicAdd(dbias_t + (long)head * L2 * L2 + gy * L2 + c, v);
        sDB[w][i] = 0.f;
      }
    }
    // ---- dK, dV of this kv tile -> staged through this head's K / V tiles -> dqkv[:, C:2C], [:, 2C:3C]
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int d = 0; d < HD / 16; ++d) {
          st_elem<T>(myK + (ks * 16 + fg * 4 + r) * L::QROW + (d * 16 + fr) * E, dk[ks][d][r] * scale);
          st_elem<T>(myV + (ks * 16 + fg * 4 + r) * L::QROW + (d * 16 + fr) * E, dv[ks][d][r]);
        }
    __syncthreads();
    for (int idx = tid; idx < 64 * CPR; idx += NT) {
      const int r = idx / CPR, cc = idx - r * CPR;
      const int h = cc / L::DCH, dc = cc - h * L::DCH;
      T* dst = dqkv + (long)sTokK[r] * C3 + (hg * NW) * HD + cc * KPL;
      *(uint4*)(dst + g.C) = *(const uint4*)(sK + (h * 64 + r) * L::QROW + dc * 16);
      *(uint4*)(dst + 2 * g.C) = *(const uint4*)(sV + (h * 64 + r) * L::QROW + dc * 16);
    }
  }
}
#undef MT_ISSUE_ONE
#undef MT_ISSUE
#undef MT_STORE_ONE

// ---------------------------------------------------------------------------------
// backward, unshifted windows of more than 64 tokens (stage 3: 32x32 windows, head_dim 64), two passes without global
// dQ atomics.  attn_bwd_mt_kernel walks the query tiles per key tile and adds every dQ strip into an f32 accumulator
// with global atomics: 64 atomic instructions per (query tile, key tile) pair, 4 x 10^8 atomic lane operations at the
// bench shape - they, not the MFMAs, were its 3 ms.  Here
//   attn_bwd_dkv_kernel  one workgroup = four waves = four 32-key tiles of ONE (window, head); the 64-query tiles
//                        (Q, dO, lse, delta) are staged once per workgroup (double buffered, next tile prefetched in
//                        registers) and shared; K / V fragments and dK / dV stay in registers.  S = Q K^T and
//                        dP = dO V^T come out with queries on the accumulator rows, so P and dS leave the accumulators
//                        as the A operands of dV += P^T dO and dK += dS^T Q (as in attn_bwd_fast2_kernel).
//   attn_bwd_dq_kernel   one workgroup = four waves = four 32-query tiles of ONE (window, head), the 64-key K / V tiles
//                        staged once per workgroup and shared (as attn_fwd_mt_kernel); Q / dO fragments and dQ in
//                        registers (32 queries per wave keep the kernel under 256 registers with the K / V prefetch
//                        live: a spilled prefetch waits for its own load at the top of every iteration).  S^T = K Q^T, dP^T = V dO^T:
//                        keys on the accumulator rows, so dS leaves them as the A operand of dQ += dS K.  dQ is written
//                        once, in the run dtype (no f32 accumulator, no finish kernel).  The bias gradient of the head
//                        is summed in one (2ws-1)^2 LDS table per workgroup.
// P is recomputed in both passes (7 instead of 5 tile products); both are VALU/MFMA balanced and free of atomics in the
// pair loop.  The relative-position bias of a 16 x 16 tile is four consecutive table entries per lane (a strip of 16
// tokens lies in one window row): the head's whole (2ws-1)^2 table x log2 e sits in LDS (15.9 KB at ws 32, loaded once per
// workgroup) and a lane reads its four values at one tile-uniform base + its own constant offset - no per-element
// gather or index arithmetic, and no global load in the pair loop (a per-pair staging of the touched rows in shifted
// copies cost an exposed L2 round trip per key / query tile: 0.51 -> 0.40 ms, 0.94 -> 0.78 ms, forward 0.38 -> 0.32 ms).
// ---------------------------------------------------------------------------------
// value of the lane n positions up / down the 16-lane row, 0 beyond the row's ends (DPP row_shl / row_shr, bound_ctrl)
template <int N> __device__ __forceinline__ float dpp_row_shl(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + N, 0xf, 0xf, true));
}
template <int N> __device__ __forceinline__ float dpp_row_shr(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + N, 0xf, 0xf, true));
}

// window-local token n of the unshifted window (b, wy, wx) -> token row
__device__ __forceinline__ int win_row0(const AttnGeo& g, int b, int wy, int wx, int n) {
  const int iy = n / g.ws, ix = n - iy * g.ws;
  return (b * g.H + wy * g.ws + iy) * g.W + wx * g.ws + ix;
}

template <typename T, int HD>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 2 : 1) void attn_bwd_dkv_kernel(
    const T* __restrict__ qkv, const float* __restrict__ bias_t, const T* __restrict__ d_out, const float* __restrict__ lse,
    const float* __restrict__ delta, T* __restrict__ dqkv, const AttnGeo g) {
  using L = Lay<T, HD>;
  constexpr int E = L::E, KPL = L::KPL, MK = TT<T>::MMA_K, SPK = MK / 16, KBQ = L::KBQ, DB = HD / 16;
  constexpr int NPF = 64 * L::DCH / 256;           // 16-byte chunks per thread, tensor and query tile
  static_assert(HD % MK == 0 && NPF >= 1 && NPF <= 4, "head_dim");
  __shared__ __attribute__((aligned(16))) unsigned char sQ[2][L::QTILE], sDO[2][L::QTILE];
  __shared__ __attribute__((aligned(16))) float sLse[2][64], sDel[2][64];
  __shared__ float sTab[63 * 63 + 3];              // the head's relative-position table x log2 e (ws <= 32), loaded once

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int nkg = g.N / 128;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int kvg = bid % nkg; bid /= nkg;
  const int head = bid % g.heads; bid /= g.heads;
  const int wx = bid % g.nwx; bid /= g.nwx;
  const int wy = bid % g.nwy; const int b = bid / g.nwy;
  const int C3 = 3 * g.C, L2 = 2 * g.ws - 1;
  const float scale = rsqrtf((float)HD), scale2 = scale * SODT_LOG2E;
  const float* bt = bias_t + (long)head * L2 * L2;
  const int kb0 = (kvg * 4 + w) * 32;              // this wave's 32 keys (window-local)
  for (int i = tid; i < L2 * L2; i += 256) sTab[i] = bt[i] * SODT_LOG2E;
  const float* tabl = &sTab[4 * fg - fr];          // + row * L2 + (qx0 - kx0 + ws - 1) + r: queries 4 fg + r, key fr

  // K / V fragments of the wave's keys: B operands of S = Q K^T and dP = dO V^T (lane: key fr of strip ns, 16 bytes of d)
  uint4 kf[2][KBQ], vf[2][KBQ];
#pragma unroll
  for (int ns = 0; ns < 2; ++ns) {
    const T* src = qkv + (long)win_row0(g, b, wy, wx, kb0 + ns * 16 + fr) * C3 + head * HD + KPL * fg;
#pragma unroll
    for (int kb = 0; kb < KBQ; ++kb) {
      kf[ns][kb] = *(const uint4*)(src + g.C + kb * MK);
      vf[ns][kb] = *(const uint4*)(src + 2 * g.C + kb * MK);
    }
  }
  f32x4 dk[2][DB], dv[2][DB];
#pragma unroll
  for (int ns = 0; ns < 2; ++ns)
#pragma unroll
    for (int d = 0; d < DB; ++d) { dk[ns][d] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[ns][d] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  // next query tile in registers (named scalars: hipcc leaves uint4 arrays used like this in scratch)
  uint4 pq0, pq1, pq2, pq3, pd0, pd1, pd2, pd3;
  float pls = 0.f;
#define KV_ISSUE_ONE(i, QT_)                                                            \
  if constexpr (NPF > i) {                                                              \
    const int idx = tid + i * 256;                                                      \
    const int r = idx / L::DCH, dc = idx - r * L::DCH;                                  \
    const long row = win_row0(g, b, wy, wx, (QT_) * 64 + r);                            \
    pq##i = *(const uint4*)(qkv + row * C3 + head * HD + dc * KPL);                     \
    pd##i = *(const uint4*)(d_out + row * g.C + head * HD + dc * KPL);                  \
  }
#define KV_ISSUE(QT_) {                                                                 \
    KV_ISSUE_ONE(0, QT_) KV_ISSUE_ONE(1, QT_) KV_ISSUE_ONE(2, QT_) KV_ISSUE_ONE(3, QT_) \
    if (tid < 128) {                                                                    \
      const long row = win_row0(g, b, wy, wx, (QT_) * 64 + (tid & 63));                 \
      pls = tid < 64 ? lse[row * g.heads + head] * SODT_LOG2E : delta[row * g.heads + head]; \
    }                                                                                   \
  }
#define KV_STORE_ONE(i, BUF_)                                                           \
  if constexpr (NPF > i) {                                                              \
    const int idx = tid + i * 256;                                                      \
    const int r = idx / L::DCH, dc = idx - r * L::DCH;                                  \
    *(uint4*)(sQ[BUF_] + r * L::QROW + dc * 16) = pq##i;                                \
    *(uint4*)(sDO[BUF_] + r * L::QROW + dc * 16) = pd##i;                               \
  }
#define KV_STORE(BUF_) {                                                                \
    KV_STORE_ONE(0, BUF_) KV_STORE_ONE(1, BUF_) KV_STORE_ONE(2, BUF_) KV_STORE_ONE(3, BUF_) \
    if (tid < 64) sLse[BUF_][tid] = pls; else if (tid < 128) sDel[BUF_][tid - 64] = pls; \
  }
  KV_ISSUE(0)
  KV_STORE(0)

  for (int qt = 0; qt < g.nqt; ++qt) {
    const int cur = qt & 1;
    __syncthreads();            // tile qt is staged; every wave is done with the other buffer
    { const int nq_ = qt + 1 < g.nqt ? qt + 1 : qt; KV_ISSUE(nq_) }
    const unsigned char* cQ = sQ[cur]; const unsigned char* cDO = sDO[cur];

#pragma unroll
    for (int kbq = 0; kbq < 4 / SPK; ++kbq) {
      uint4 Ap[2], Ads[2];
#pragma unroll
      for (int hh = 0; hh < SPK; ++hh) {
        const int ms = kbq * SPK + hh;
        f32x4 s[2], dp[2];
#pragma unroll
        for (int kb = 0; kb < KBQ; ++kb) {
          const uint4 fq = frag<T>(cQ, L::QROW, ms * 16, kb, HD, lane);
          const uint4 fo = frag<T>(cDO, L::QROW, ms * 16, kb, HD, lane);
#pragma unroll
          for (int ns = 0; ns < 2; ++ns) {
            if (kb == 0) { s[ns] = mma16z<T>(fq, kf[ns][0]); dp[ns] = mma16z<T>(fo, vf[ns][0]); }
            else { mma16<T>(s[ns], fq, kf[ns][kb]); mma16<T>(dp[ns], fo, vf[ns][kb]); }
          }
        }
        // rows 4 fg + r of the strip: queries; column fr: key
        const float4 lq = *(const float4*)&sLse[cur][ms * 16 + 4 * fg];
        const float4 dl = *(const float4*)&sDel[cur][ms * 16 + 4 * fg];
        const float lqa[4] = {lq.x, lq.y, lq.z, lq.w}, dla[4] = {dl.x, dl.y, dl.z, dl.w};
        const int qn0 = qt * 64 + ms * 16;
        const int qy = qn0 / g.ws, qx0 = qn0 - qy * g.ws;
#pragma unroll
        for (int ns = 0; ns < 2; ++ns) {
          const int kn0 = kb0 + ns * 16;
          const int ky = kn0 / g.ws, kx0 = kn0 - ky * g.ws;
          const float* tb = tabl + (qy - ky + g.ws - 1) * L2 + (qx0 - kx0 + g.ws - 1);   // entries tb[0..3] <-> queries 4 fg .. 4 fg + 3
          const float ba[4] = {tb[0], tb[1], tb[2], tb[3]};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = fast_exp2(fmaf(s[ns][r], scale2, ba[r]) - lqa[r]);
            s[ns][r] = p;
            dp[ns][r] = p * (dp[ns][r] - dla[r]);
          }
          if constexpr (std::is_same<T, bf16>::value) {
            const uint32_t p01 = pack2bf(s[ns][0], s[ns][1]), p23 = pack2bf(s[ns][2], s[ns][3]);
            const uint32_t d01 = pack2bf(dp[ns][0], dp[ns][1]), d23 = pack2bf(dp[ns][2], dp[ns][3]);
            if (hh == 0) { Ap[ns].x = p01; Ap[ns].y = p23; Ads[ns].x = d01; Ads[ns].y = d23; }
            else { Ap[ns].z = p01; Ap[ns].w = p23; Ads[ns].z = d01; Ads[ns].w = d23; }
          } else {
            Ap[ns] = make_uint4(__float_as_uint(s[ns][0]), __float_as_uint(s[ns][1]), __float_as_uint(s[ns][2]), __float_as_uint(s[ns][3]));
            Ads[ns] = make_uint4(__float_as_uint(dp[ns][0]), __float_as_uint(dp[ns][1]), __float_as_uint(dp[ns][2]), __float_as_uint(dp[ns][3]));
          }
        }
      }
      // dV += P^T dO, dK += dS^T Q over the queries of this strip (pair)
#pragma unroll
      for (int d = 0; d < DB; ++d) {
        const uint4 fdo = fragTp<T>(cDO, L::QROW, kbq, d * 16, lane);
        const uint4 fqq = fragTp<T>(cQ, L::QROW, kbq, d * 16, lane);
#pragma unroll
        for (int ns = 0; ns < 2; ++ns) {
          mma16<T>(dv[ns][d], Ap[ns], fdo);
          mma16<T>(dk[ns][d], Ads[ns], fqq);
        }
      }
    }
    KV_STORE(cur ^ 1)
  }
#undef KV_ISSUE_ONE
#undef KV_ISSUE
#undef KV_STORE_ONE
#undef KV_STORE
  // ---- dK (rows 0..31) and dV (rows 32..63) of the wave's keys staged through one of the four tiles, coalesced store
  __syncthreads();
  unsigned char* st = w < 2 ? sQ[w] : sDO[w - 2];
#pragma unroll
  for (int ns = 0; ns < 2; ++ns)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int d = 0; d < DB; ++d) {
        st_elem<T>(st + (ns * 16 + fg * 4 + r) * L::QROW + (d * 16 + fr) * E, dk[ns][d][r] * scale);
        st_elem<T>(st + (32 + ns * 16 + fg * 4 + r) * L::QROW + (d * 16 + fr) * E, dv[ns][d][r]);
      }
  wave_sync();
  for (int idx = lane; idx < 64 * L::DCH; idx += 64) {
    const int r = idx / L::DCH, dc = idx - r * L::DCH;
    T* dst = dqkv + (long)win_row0(g, b, wy, wx, kb0 + (r & 31)) * C3 + (r < 32 ? g.C : 2 * g.C) + head * HD + dc * KPL;
    *(uint4*)dst = *(const uint4*)(st + r * L::QROW + dc * 16);
  }
}

template <typename T, int HD>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 2 : 1) void attn_bwd_dq_kernel(
    const T* __restrict__ qkv, const float* __restrict__ bias_t, const T* __restrict__ d_out, const float* __restrict__ lse,
    const float* __restrict__ delta, T* __restrict__ dqkv, float* __restrict__ dbias_t, const AttnGeo g) {
  using L = Lay<T, HD>;
  constexpr int E = L::E, KPL = L::KPL, MK = TT<T>::MMA_K, SPK = MK / 16, KBQ = L::KBQ, DB = HD / 16;
  constexpr int NPF = 64 * L::DCH / 256;
  static_assert(HD % MK == 0 && NPF >= 1 && NPF <= 4, "head_dim");
  __shared__ __attribute__((aligned(16))) unsigned char sK[2][L::QTILE], sV[2][L::QTILE];
  __shared__ float sTab[63 * 63 + 3];              // the head's relative-position table x log2 e (ws <= 32), loaded once
  __shared__ float sDB[63 * 63 + 3];               // bias gradient of the head (ws <= 32)
  __shared__ float sScr[4][192];                   // per wave: four 48-float rows for the cross-row diagonal sums

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int nqg = g.N / 128;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int qg = bid % nqg; bid /= nqg;
  const int head = bid % g.heads; bid /= g.heads;
  const int wx = bid % g.nwx; bid /= g.nwx;
  const int wy = bid % g.nwy; const int b = bid / g.nwy;
  const int q0 = (qg * 4 + w) * 32;                // this wave's 32 queries (window-local)
  const int C3 = 3 * g.C, L2 = 2 * g.ws - 1;
  const float scale = rsqrtf((float)HD), scale2 = scale * SODT_LOG2E;
  const float* bt = bias_t + (long)head * L2 * L2;
  for (int i = tid; i < L2 * L2; i += 256) { sDB[i] = 0.f; sTab[i] = bt[i] * SODT_LOG2E; }
  for (int i = tid; i < 4 * 192; i += 256) (&sScr[0][0])[i] = 0.f;
  float* myScr = sScr[w];
  const float* tabl = &sTab[fr - 4 * fg - 3];      // + row * L2 + (qx0 - kx0 + ws - 1) + (3 - r): query fr, keys 4 fg + r

  // Q / dO fragments of the wave's queries: B operands of S^T = K Q^T and dP^T = V dO^T (lane: query fr of strip ms)
  uint4 fq[2][KBQ], fo[2][KBQ];
  float lq[2], dl[2];
#pragma unroll
  for (int ms = 0; ms < 2; ++ms) {
    const long row = win_row0(g, b, wy, wx, q0 + ms * 16 + fr);
#pragma unroll
    for (int kb = 0; kb < KBQ; ++kb) {
      fq[ms][kb] = *(const uint4*)(qkv + row * C3 + head * HD + kb * MK + KPL * fg);
      fo[ms][kb] = *(const uint4*)(d_out + row * g.C + head * HD + kb * MK + KPL * fg);
    }
    lq[ms] = lse[row * g.heads + head] * SODT_LOG2E;
    dl[ms] = delta[row * g.heads + head];
  }
  f32x4 dq[2][DB];
#pragma unroll
  for (int ms = 0; ms < 2; ++ms)
#pragma unroll
    for (int d = 0; d < DB; ++d) dq[ms][d] = f32x4{0.f, 0.f, 0.f, 0.f};

  uint4 pk0, pk1, pk2, pk3, pv0, pv1, pv2, pv3;
#define DQ_ISSUE_ONE(i, KT_)                                                            \
  if constexpr (NPF > i) {                                                              \
    const int idx = tid + i * 256;                                                      \
    const int r = idx / L::DCH, dc = idx - r * L::DCH;                                  \
    const T* src = qkv + (long)win_row0(g, b, wy, wx, (KT_) * 64 + r) * C3 + head * HD + dc * KPL; \
    pk##i = *(const uint4*)(src + g.C); pv##i = *(const uint4*)(src + 2 * g.C);         \
  }
#define DQ_ISSUE(KT_) { DQ_ISSUE_ONE(0, KT_) DQ_ISSUE_ONE(1, KT_) DQ_ISSUE_ONE(2, KT_) DQ_ISSUE_ONE(3, KT_) }
#define DQ_STORE_ONE(i, BUF_)                                                           \
  if constexpr (NPF > i) {                                                              \
    const int idx = tid + i * 256;                                                      \
    const int r = idx / L::DCH, dc = idx - r * L::DCH;                                  \
    *(uint4*)(sK[BUF_] + r * L::QROW + dc * 16) = pk##i; *(uint4*)(sV[BUF_] + r * L::QROW + dc * 16) = pv##i; \
  }
#define DQ_STORE(BUF_) { DQ_STORE_ONE(0, BUF_) DQ_STORE_ONE(1, BUF_) DQ_STORE_ONE(2, BUF_) DQ_STORE_ONE(3, BUF_) }
  DQ_ISSUE(0)
  DQ_STORE(0)

  for (int kt = 0; kt < g.nqt; ++kt) {
    const int cur = kt & 1;
    __syncthreads();
    { const int nk_ = kt + 1 < g.nqt ? kt + 1 : kt; DQ_ISSUE(nk_) }
    const unsigned char* cK = sK[cur]; const unsigned char* cV = sV[cur];

#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
      const int qn0 = q0 + ms * 16;
      const int qy = qn0 / g.ws, qx0 = qn0 - qy * g.ws;
#pragma unroll
      for (int kbq = 0; kbq < 4 / SPK; ++kbq) {
        uint4 ap;
#pragma unroll
        for (int hh = 0; hh < SPK; ++hh) {
          // rows 4 fg + r of key strip ks: keys; column fr: this lane's query of strip ms
          const int ks = kbq * SPK + hh;
          f32x4 s, dp;
#pragma unroll
          for (int kb = 0; kb < KBQ; ++kb) {
            const uint4 fk = frag<T>(cK, L::QROW, ks * 16, kb, HD, lane);
            const uint4 fv = frag<T>(cV, L::QROW, ks * 16, kb, HD, lane);
            if (kb == 0) { s = mma16z<T>(fk, fq[ms][0]); dp = mma16z<T>(fv, fo[ms][0]); }
            else { mma16<T>(s, fk, fq[ms][kb]); mma16<T>(dp, fv, fo[ms][kb]); }
          }
          const int kn0 = kt * 64 + ks * 16;
          const int ky = kn0 / g.ws, kx0 = kn0 - ky * g.ws;
          const float* tb = tabl + (qy - ky + g.ws - 1) * L2 + (qx0 - kx0 + g.ws - 1);   // key 4 fg + r <-> tb[3 - r]
          const float ba[4] = {tb[3], tb[2], tb[1], tb[0]};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = fast_exp2(fmaf(s[r], scale2, ba[r]) - lq[ms]);
            dp[r] = p * (dp[r] - dl[ms]);
          }
          // bias gradient: dS[key 4 fg + r][query fr] belongs to dx = (qx0 + fr) - (kx0 + 4 fg + r).  LDS float atomics
          // cost ~3 cycles per active lane, so the four keys of a lane are first summed along the diagonals of the tile
          // inside the 16-lane row (key + 1 <-> query + 1 <-> next lane): lane fr ends with the diagonal
          // dx0 + fr, dx0 = qx0 - kx0 - 4 fg; the six elements that fall off the low end are the diagonals
          // dx0 - 3 .. dx0 - 1, collected in lanes 0..2 of `lo` (76 instead of 256 values per tile) ...
          // ... and the four rows (diagonals x - 4 fg of row fg) are then lined up through a 4 x 48-float per-wave scratch
          // (slot x + 16; never-written slots stay zero) so that ONE atomic instruction with 31 active lanes covers the tile.
          {
            const float u = dp[0] + dpp_row_shl<1>(dp[1]) + dpp_row_shl<2>(dp[2]) + dpp_row_shl<3>(dp[3]);
            const float lo = dp[3] + dpp_row_shr<1>(dp[2]) + dpp_row_shr<2>(dp[1]);
            myScr[fg * 48 + 16 + fr] = u;
            if (fr < 3) myScr[fg * 48 + 13 + fr] = lo;
            wave_sync();
            if (lane < 31) {
              const float v = myScr[lane + 1] + myScr[48 + lane + 5] + myScr[96 + lane + 9] + myScr[144 + lane + 13];
              atomicAdd(&sDB[(qy - ky + g.ws - 1) * L2 + (qx0 - kx0 + g.ws - 1) + lane - 15], v);
            }
            wave_sync();
          }
          // dS strips leave the accumulators as the A operand of dQ += dS K (row = query fr, k-slots = keys)
          if constexpr (std::is_same<T, bf16>::value) {
            const uint32_t d01 = pack2bf(dp[0], dp[1]), d23 = pack2bf(dp[2], dp[3]);
            if (hh == 0) { ap.x = d01; ap.y = d23; } else { ap.z = d01; ap.w = d23; }
          } else {
            ap = make_uint4(__float_as_uint(dp[0]), __float_as_uint(dp[1]), __float_as_uint(dp[2]), __float_as_uint(dp[3]));
          }
        }
#pragma unroll
        for (int d = 0; d < DB; ++d) mma16<T>(dq[ms][d], ap, fragTp_fwd<T>(cK, L::QROW, kbq, d * 16, lane));
      }
    }
    DQ_STORE(cur ^ 1)
  }
#undef DQ_ISSUE_ONE
#undef DQ_ISSUE
#undef DQ_STORE_ONE
#undef DQ_STORE
  // ---- bias gradient of the head; dQ staged through one of the four tiles, coalesced store
  __syncthreads();
  for (int i = tid; i < L2 * L2; i += 256) {
    const float v = sDB[i];
    if (v != 0.f) atomicAdd(dbias_t + (long)head * L2 * L2 + i, v);
  }
  unsigned char* st = sK[w >> 1] + (w & 1) * 32 * L::QROW;
#pragma unroll
  for (int ms = 0; ms < 2; ++ms)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int d = 0; d < DB; ++d)
        st_elem<T>(st + (ms * 16 + fg * 4 + r) * L::QROW + (d * 16 + fr) * E, dq[ms][d][r] * scale);
  wave_sync();
  for (int idx = lane; idx < 32 * L::DCH; idx += 64) {
    const int r = idx / L::DCH, dc = idx - r * L::DCH;
    *(uint4*)(dqkv + (long)win_row0(g, b, wy, wx, q0 + r) * C3 + head * HD + dc * KPL) = *(const uint4*)(st + r * L::QROW + dc * 16);
  }
}

// ---------------------------------------------------------------------------------
// forward, unshifted windows of more than 64 tokens, second generation (same machinery as attn_bwd_dq_kernel): one
// workgroup = four waves = four 64-query tiles of ONE (window, head); Q fragments live in registers, the 64-key K / V
// tiles are double buffered in LDS and shared, the bias of a 16 x 16 tile is four table entries at a tile-uniform base
// (whole table in LDS; attn_fwd_mt_kernel gathers one table entry per element through three LDS reads and ~10
// integer operations: its softmax, not its MFMAs, set its 0.44 ms), online-softmax statistics cross the four 16-lane
// rows with v_permlane swaps instead of ds_bpermute.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float rows4_sum(float v) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float rows4_max(float v) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float s = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

template <typename T, int HD>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 2 : 1) void attn_fwd_mt2_kernel(
    const T* __restrict__ qkv, const float* __restrict__ bias_t, T* __restrict__ out, float* __restrict__ lse, const AttnGeo g) {
  using L = Lay<T, HD>;
  constexpr int E = L::E, KPL = L::KPL, MK = TT<T>::MMA_K, SPK = MK / 16, KBQ = L::KBQ, DB = HD / 16;
  constexpr int NPF = 64 * L::DCH / 256;
  static_assert(HD % MK == 0 && NPF >= 1 && NPF <= 4, "head_dim");
  __shared__ __attribute__((aligned(16))) unsigned char sK[2][L::QTILE], sV[2][L::QTILE];
  __shared__ float sTab[63 * 63 + 3];              // the head's relative-position table x log2 e (ws <= 32), loaded once

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int nqg = g.nqt / 4;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int qg = bid % nqg; bid /= nqg;
  const int head = bid % g.heads; bid /= g.heads;
  const int wx = bid % g.nwx; bid /= g.nwx;
  const int wy = bid % g.nwy; const int b = bid / g.nwy;
  const int qt = qg * 4 + w;
  const int C3 = 3 * g.C, L2 = 2 * g.ws - 1;
  const float scale2 = rsqrtf((float)HD) * SODT_LOG2E;
  const float* bt = bias_t + (long)head * L2 * L2;
  for (int i = tid; i < L2 * L2; i += 256) sTab[i] = bt[i] * SODT_LOG2E;
  const float* tabl = &sTab[fr - 4 * fg - 3];      // + row * L2 + (qx0 - kx0 + ws - 1) + (3 - r): query fr, keys 4 fg + r

  uint4 fq[4][KBQ];                                  // B operands of S^T = K Q^T (lane: query fr of strip ms)
#pragma unroll
  for (int ms = 0; ms < 4; ++ms) {
    const long row = win_row0(g, b, wy, wx, qt * 64 + ms * 16 + fr);
#pragma unroll
    for (int kb = 0; kb < KBQ; ++kb) fq[ms][kb] = *(const uint4*)(qkv + row * C3 + head * HD + kb * MK + KPL * fg);
  }
  float m_run[4], l_run[4];
  f32x4 o[4][DB];
#pragma unroll
  for (int ms = 0; ms < 4; ++ms) {
    m_run[ms] = -1e30f; l_run[ms] = 0.f;
#pragma unroll
    for (int d = 0; d < DB; ++d) o[ms][d] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  uint4 pk0, pk1, pk2, pk3, pv0, pv1, pv2, pv3;
#define F2_ISSUE_ONE(i, KT_)                                                            \
  if constexpr (NPF > i) {                                                              \
    const int idx = tid + i * 256;                                                      \
    const int r = idx / L::DCH, dc = idx - r * L::DCH;                                  \
    const T* src = qkv + (long)win_row0(g, b, wy, wx, (KT_) * 64 + r) * C3 + head * HD + dc * KPL; \
    pk##i = *(const uint4*)(src + g.C); pv##i = *(const uint4*)(src + 2 * g.C);         \
  }
#define F2_ISSUE(KT_) { F2_ISSUE_ONE(0, KT_) F2_ISSUE_ONE(1, KT_) F2_ISSUE_ONE(2, KT_) F2_ISSUE_ONE(3, KT_) }
#define F2_STORE_ONE(i, BUF_)                                                           \
  if constexpr (NPF > i) {                                                              \
    const int idx = tid + i * 256;                                                      \
    const int r = idx / L::DCH, dc = idx - r * L::DCH;                                  \
    *(uint4*)(sK[BUF_] + r * L::QROW + dc * 16) = pk##i; *(uint4*)(sV[BUF_] + r * L::QROW + dc * 16) = pv##i; \
  }
#define F2_STORE(BUF_) { F2_STORE_ONE(0, BUF_) F2_STORE_ONE(1, BUF_) F2_STORE_ONE(2, BUF_) F2_STORE_ONE(3, BUF_) }
  F2_ISSUE(0)
  F2_STORE(0)

  for (int kt = 0; kt < g.nqt; ++kt) {
    const int cur = kt & 1;
    __syncthreads();
    { const int nk_ = kt + 1 < g.nqt ? kt + 1 : kt; F2_ISSUE(nk_) }
    const unsigned char* cK = sK[cur]; const unsigned char* cV = sV[cur];

#pragma unroll
    for (int ms = 0; ms < 4; ++ms) {
      // S^T strip: keys on rows (4 fg + r of key strip ks), this lane's query = 16 ms + fr
      f32x4 s[4];
#pragma unroll
      for (int kb = 0; kb < KBQ; ++kb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const uint4 fk = frag<T>(cK, L::QROW, ks * 16, kb, HD, lane);
          if (kb == 0) s[ks] = mma16z<T>(fk, fq[ms][0]); else mma16<T>(s[ks], fk, fq[ms][kb]);
        }
      const int qn0 = qt * 64 + ms * 16;
      const int qy = qn0 / g.ws, qx0 = qn0 - qy * g.ws;
      float mx = -1e30f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int kn0 = kt * 64 + ks * 16;
        const int ky = kn0 / g.ws, kx0 = kn0 - ky * g.ws;
        const float* tb = tabl + (qy - ky + g.ws - 1) * L2 + (qx0 - kx0 + g.ws - 1);     // key 4 fg + r <-> tb[3 - r]
        const float ba[4] = {tb[3], tb[2], tb[1], tb[0]};
#pragma unroll
        for (int r = 0; r < 4; ++r) { s[ks][r] = fmaf(s[ks][r], scale2, ba[r]); mx = fmaxf(mx, s[ks][r]); }
      }
      mx = rows4_max(mx);
      const float mnew = fmaxf(m_run[ms], mx);
      const float alpha = fast_exp2(m_run[ms] - mnew);
      float sum = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float p = fast_exp2(s[ks][r] - mnew); s[ks][r] = p; sum += p; }
      sum = rows4_sum(sum);
      l_run[ms] = l_run[ms] * alpha + sum;
      m_run[ms] = mnew;
      // rescale O rows (row 4 fg + r of strip ms <-> query column 4 fg + r, held by lane 4 fg + r of every row)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ar = __shfl(alpha, 4 * fg + r);
#pragma unroll
        for (int d = 0; d < DB; ++d) o[ms][d][r] *= ar;
      }
      // O[ms] += P V: P strips straight from the accumulators (A operand: row = query fr, k-slots = keys)
#pragma unroll
      for (int kbq = 0; kbq < 4 / SPK; ++kbq) {
        uint4 ap;
        if constexpr (std::is_same<T, bf16>::value) {
          ap = make_uint4(pack2bf(s[2 * kbq][0], s[2 * kbq][1]), pack2bf(s[2 * kbq][2], s[2 * kbq][3]),
                          pack2bf(s[2 * kbq + 1][0], s[2 * kbq + 1][1]), pack2bf(s[2 * kbq + 1][2], s[2 * kbq + 1][3]));
        } else {
          ap = make_uint4(__float_as_uint(s[kbq][0]), __float_as_uint(s[kbq][1]), __float_as_uint(s[kbq][2]), __float_as_uint(s[kbq][3]));
        }
#pragma unroll
        for (int d = 0; d < DB; ++d) mma16<T>(o[ms][d], ap, fragTp_fwd<T>(cV, L::QROW, kbq, d * 16, lane));
      }
    }
    F2_STORE(cur ^ 1)
  }
#undef F2_ISSUE_ONE
#undef F2_ISSUE
#undef F2_STORE_ONE
#undef F2_STORE
  // ---- normalise, lse, stage O through one of the four tiles, coalesced store
  __syncthreads();
  unsigned char* st = w < 2 ? sK[w] : sV[w - 2];
#pragma unroll
  for (int ms = 0; ms < 4; ++ms) {
    const float inv = __builtin_amdgcn_rcpf(l_run[ms]);
    if (fg == 0 && lse)
      lse[(long)win_row0(g, b, wy, wx, qt * 64 + ms * 16 + fr) * g.heads + head] = m_run[ms] * (1.0f / SODT_LOG2E) + __logf(l_run[ms]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float ir = __shfl(inv, 4 * fg + r);
#pragma unroll
      for (int d = 0; d < DB; ++d)
        st_elem<T>(st + (ms * 16 + fg * 4 + r) * L::QROW + (d * 16 + fr) * E, o[ms][d][r] * ir);
    }
  }
  wave_sync();
  for (int idx = lane; idx < 64 * L::DCH; idx += 64) {
    const int r = idx / L::DCH, dc = idx - r * L::DCH;
    *(uint4*)(out + (long)win_row0(g, b, wy, wx, qt * 64 + r) * g.C + head * HD + dc * KPL) = *(const uint4*)(st + r * L::QROW + dc * 16);
  }
}

bool make_geo(AttnGeo& g, int B, int H, int W, int C, int heads, int ws, int shift) {
  if (B <= 0 || H <= 0 || W <= 0 || ws <= 0 || (H % ws) || (W % ws) || heads <= 0 || (C % heads)) return false;
  if ((ws * ws) % 64) return false;
  if (ws < 8 || (64 % ws && ws < 64) ) return false;           // a 64-token tile must cover whole window rows
  if (ws > 64) return false;
  if (shift < 0 || shift >= ws) return false;
  g.B = B; g.H = H; g.W = W; g.C = C; g.heads = heads; g.ws = ws; g.shift = shift;
  g.nwy = H / ws; g.nwx = W / ws; g.N = ws * ws; g.nqt = g.N / 64;
  return true;
}

template <typename T, int HD, int NW>
int launch_fwd(const void* qkv, const float* bias_t, void* out, float* lse, const AttnGeo& g, hipStream_t st) {
  if (g.heads % NW) return SODT_EINVAL;
  if constexpr (3 * Lay<T, HD>::DCH <= 12) {
    if (g.nqt == 1) {
      const int nwin = g.B * g.nwy * g.nwx;
      const int gx = nwin < 512 ? nwin : 512;
      hipLaunchKernelGGL((attn_fwd_fast_kernel<T, HD, NW>), dim3(gx, g.heads / NW), dim3(NW * 64), 0, st,
                         (const T*)qkv, bias_t, (T*)out, lse, g, nwin);
      return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
    }
  }
  if constexpr (HD % TT<T>::MMA_K == 0 && Lay<T, HD>::DCH >= 4 && Lay<T, HD>::DCH <= 16) {
    if (g.nqt > 1 && (g.nqt % 4) == 0 && g.shift == 0 && g.ws <= 32) {
      const long nb = (long)g.B * g.nwy * g.nwx * (g.nqt / 4) * g.heads;
      hipLaunchKernelGGL((attn_fwd_mt2_kernel<T, HD>), dim3((unsigned)nb), dim3(256), 0, st,
                         (const T*)qkv, bias_t, (T*)out, lse, g);
      return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
    }
  }
  if constexpr (Lay<T, HD>::DCH >= 4 && Lay<T, HD>::DCH <= 16) {
    if (g.nqt > 1 && (g.nqt % 4) == 0) {
      const long nb = (long)g.B * g.nwy * g.nwx * (g.nqt / 4) * g.heads;
      hipLaunchKernelGGL((attn_fwd_mt_kernel<T, HD>), dim3((unsigned)nb), dim3(256), 0, st,
                         (const T*)qkv, bias_t, (T*)out, lse, g);
      return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
    }
  }
  const long blocks = (long)g.B * g.nwy * g.nwx * g.nqt * (g.heads / NW);
  hipLaunchKernelGGL((attn_fwd_kernel<T, HD, NW>), dim3((unsigned)blocks), dim3(NW * 64), 0, st,
                     (const T*)qkv, bias_t, (T*)out, lse, g);
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

// Grid of the persistent single-tile backward kernels: every workgroup ends with one flush of its bias-gradient table
// (225 global atomics per wave onto the 2,700 floats of the 12 heads).  With 1024 x (heads / NW) workgroups that was
// 2.7 M same-line device-scope atomics per launch - 0.2 ms, 40 % of the stage-1 launch (the loops themselves ran six
// rounds of 0.05 ms).  Launch only as many workgroups as are resident at once and let them walk more windows.
int bwd_persistent_grid(int nwin, int ngroups, int NW) {
  const int resident = 256 * (NW == 4 ? 2 : (NW == 2 ? 2 : 4));     // CUs x workgroups per CU (launch bounds / LDS)
  int gx = resident / (ngroups > 0 ? ngroups : 1);
  if (gx < 1) gx = 1;
  return nwin < gx ? nwin : gx;
}

template <typename T, int HD, int NW>
int launch_bwd_wm(const void* qkvw, const float* bias_t, const void* dout, const float* lsew, void* dqkv,
                  float* dbias_t, const AttnGeo& g, hipStream_t st) {
  if (g.heads % NW || g.nqt != 1 || g.ws != 8) return SODT_EINVAL;
  const int nwin = g.B * g.nwy * g.nwx;
  const int gx = bwd_persistent_grid(nwin, g.heads / NW, NW);
  hipLaunchKernelGGL((attn_bwd_fast2_kernel<T, HD, NW, true>), dim3(gx, g.heads / NW), dim3(NW * 64), 0, st,
                     (const T*)qkvw, bias_t, (const T*)dout, lsew, (T*)dqkv, dbias_t, g, nwin);
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

// the same walk with q / k / v recomputed from the block's saved LN1 output (RC)
int launch_bwd_rc(const void* xn1, const unsigned char* wpk, const float* bias_t, const void* dout, const float* lsew, void* dqkv,
                  float* dbias_t, const AttnGeo& g, hipStream_t st) {
  if (g.heads != 12 || g.C != 192 || g.nqt != 1 || g.ws != 8) return SODT_EINVAL;
  const int nwin = g.B * g.nwy * g.nwx;
  int gx = bwd_persistent_grid(nwin, g.heads / 4, 4);
  gx = gx >= 8 ? gx / 8 * 8 : 8;                      // walkers in whole rounds of the 8 XCDs (see the kernel's id map); walkers
  //                                                    beyond the window count only join the final (all-zero) bias-gradient flush
  hipLaunchKernelGGL((attn_bwd_fast2_kernel<bf16, 16, 4, true, true>), dim3(3 * gx), dim3(256), 0, st,
                     (const bf16*)xn1, bias_t, (const bf16*)dout, lsew, (bf16*)dqkv, dbias_t, g, nwin, wpk);
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

template <typename T, int HD, int NW>
int launch_bwd(const void* qkv, const float* bias_t, const void* out, const void* dout, const float* lse, void* dqkv,
               float* dbias_t, float* scratch, const AttnGeo& g, hipStream_t st) {
  if (g.heads % NW) return SODT_EINVAL;
  const long M = (long)g.B * g.H * g.W;
  const int nwin = g.B * g.nwy * g.nwx;
  float* dq_acc = nullptr; float* delta = nullptr;
  if (g.nqt > 1) {
    if (!scratch || !out) return SODT_EINVAL;
    dq_acc = scratch; delta = scratch + M * g.C;
    hipLaunchKernelGGL((attn_delta_kernel<T>), dim3(1024), dim3(256), 0, st, (const T*)out, (const T*)dout, delta, M, g.C, g.heads);
  }
  const int nitems = nwin * g.nqt;
  int gx = nitems < 1024 ? nitems : 1024;
  constexpr bool PFOK = (4 * Lay<T, HD>::DCH <= 16);
  if (PFOK && g.nqt == 1 && g.ws == 8) {
    gx = bwd_persistent_grid(nwin, g.heads / NW, NW);
    if constexpr (PFOK) {
      hipLaunchKernelGGL((attn_bwd_fast2_kernel<T, HD, NW>), dim3(gx, g.heads / NW), dim3(NW * 64), 0, st,
                         (const T*)qkv, bias_t, (const T*)dout, lse, (T*)dqkv, dbias_t, g, nwin);
    }
  } else if (PFOK && g.nqt == 1) {
    hipLaunchKernelGGL((attn_bwd_kernel<T, HD, NW, PFOK>), dim3(gx, g.heads / NW), dim3(NW * 64), 0, st,
                       (const T*)qkv, bias_t, (const T*)dout, lse, delta, (T*)dqkv, dbias_t, dq_acc, g, nwin, 0);
  } else {
    if constexpr (HD % TT<T>::MMA_K == 0 && Lay<T, HD>::DCH >= 4 && Lay<T, HD>::DCH <= 16) {
      if (g.nqt > 1 && g.shift == 0 && g.ws <= 32 && (g.nqt % 4) == 0 && (g.N % 128) == 0) {   // two passes, no dQ atomics
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, HD>), dim3((unsigned)(nwin * g.heads * (g.N / 128))), dim3(256), 0, st,
                           (const T*)qkv, bias_t, (const T*)dout, lse, delta, (T*)dqkv, g);
        hipLaunchKernelGGL((attn_bwd_dq_kernel<T, HD>), dim3((unsigned)(nwin * g.heads * (g.N / 128))), dim3(256), 0, st,
                           (const T*)qkv, bias_t, (const T*)dout, lse, delta, (T*)dqkv, dbias_t, g);
        return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
      }
    }
    if (g.nqt > 1) {
      hipLaunchKernelGGL((attn_bwd_mt_kernel<T, HD, NW>), dim3(gx, g.heads / NW), dim3(NW * 64), 0, st,
                         (const T*)qkv, bias_t, (const T*)dout, lse, delta, (T*)dqkv, dbias_t, dq_acc, g, nwin);
      hipLaunchKernelGGL((attn_dq_finish_kernel<T>), dim3(1024), dim3(256), 0, st, dq_acc, (T*)dqkv, M, g.C);
      return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
    }
    // multi-tile windows: keep the whole (2ws-1)^2 bias-gradient table of each head in LDS when it fits
    using L = Lay<T, HD>;
    const int L2 = 2 * g.ws - 1;
    const int static_lds = NW * (4 * L::QTILE + 2 * L::STILE) + 2 * NW * 228 * 4 + 2 * NW * 64 * 4 + 2048;
    const int dyn = NW * L2 * L2 * 4;
    int fulldb = (g.nqt > 1 && static_lds + dyn <= 160 * 1024) ? 1 : 0;
    if (fulldb) {
      static int attr_bytes = 0;
      if (dyn > attr_bytes) {
        if (hipFuncSetAttribute((const void*)attn_bwd_kernel<T, HD, NW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn) != hipSuccess) {
          (void)hipGetLastError();
          fulldb = 0;
        } else attr_bytes = dyn;
      }
      if (gx > 128) gx = 128;      // few, long-lived workgroups: one table flush each
    }
    hipLaunchKernelGGL((attn_bwd_kernel<T, HD, NW, false>), dim3(gx, g.heads / NW), dim3(NW * 64), fulldb ? dyn : 0, st,
                       (const T*)qkv, bias_t, (const T*)dout, lse, delta, (T*)dqkv, dbias_t, dq_acc, g, nwin, fulldb);
  }
  if (g.nqt > 1)
    hipLaunchKernelGGL((attn_dq_finish_kernel<T>), dim3(1024), dim3(256), 0, st, dq_acc, (T*)dqkv, M, g.C);
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

}  // namespace

extern "C" int sodt_window_attn_fwd(const void* qkv, const float* bias_t, void* out, float* lse,
                                    int B, int H, int W, int C, int heads, int ws, int shift,
                                    int dtype, sodt_stream_t st_) {
  AttnGeo g;
  if (!qkv || !bias_t || !out || !make_geo(g, B, H, W, C, heads, ws, shift)) return SODT_EINVAL;
  hipStream_t st = (hipStream_t)st_;
  const int hd = C / heads;
  if (dtype == SODT_BF16) {
    if (hd == 16) return launch_fwd<bf16, 16, 4>(qkv, bias_t, out, lse, g, st);
    if (hd == 32) return launch_fwd<bf16, 32, 4>(qkv, bias_t, out, lse, g, st);
    if (hd == 64) return launch_fwd<bf16, 64, 2>(qkv, bias_t, out, lse, g, st);
  } else if (dtype == SODT_F32) {
    if (hd == 16) return launch_fwd<float, 16, 4>(qkv, bias_t, out, lse, g, st);
    if (hd == 32) return launch_fwd<float, 32, 2>(qkv, bias_t, out, lse, g, st);
    if (hd == 64) return launch_fwd<float, 64, 2>(qkv, bias_t, out, lse, g, st);
  }
  return SODT_EINVAL;
}

extern "C" int sodt_window_attn_bwd(const void* qkv, const float* bias_t, const void* out, const void* dout,
                                    const float* lse, void* dqkv, float* dbias_t, float* dq_acc,
                                    int B, int H, int W, int C, int heads, int ws, int shift,
                                    int dtype, sodt_stream_t st_) {
  AttnGeo g;
  if (!qkv || !bias_t || !dout || !lse || !dqkv || !dbias_t || !make_geo(g, B, H, W, C, heads, ws, shift)) return SODT_EINVAL;
  hipStream_t st = (hipStream_t)st_;
  const int hd = C / heads;
  if (dtype == SODT_BF16) {
    if (hd == 16) return launch_bwd<bf16, 16, 4>(qkv, bias_t, out, dout, lse, dqkv, dbias_t, dq_acc, g, st);
    if (hd == 32) return launch_bwd<bf16, 32, 2>(qkv, bias_t, out, dout, lse, dqkv, dbias_t, dq_acc, g, st);
    if (hd == 64) return launch_bwd<bf16, 64, 1>(qkv, bias_t, out, dout, lse, dqkv, dbias_t, dq_acc, g, st);
  } else if (dtype == SODT_F32) {
    if (hd == 16) return launch_bwd<float, 16, 2>(qkv, bias_t, out, dout, lse, dqkv, dbias_t, dq_acc, g, st);
    if (hd == 32) return launch_bwd<float, 32, 2>(qkv, bias_t, out, dout, lse, dqkv, dbias_t, dq_acc, g, st);
    if (hd == 64) return launch_bwd<float, 64, 1>(qkv, bias_t, out, dout, lse, dqkv, dbias_t, dq_acc, g, st);
  }
  return SODT_EINVAL;
}

/* the same backward on the window-major q / k / v and log-sum-exp the fused forward saves (wmsa_block.hip) */
extern "C" int sodt_window_attn_bwd_wm(const void* qkvw, const float* bias_t, const void* dout, const float* lsew,
                                       void* dqkv, float* dbias_t, int B, int H, int W, int C, int heads, int ws,
                                       int shift, int dtype, sodt_stream_t st_) {
  AttnGeo g;
  if (!qkvw || !bias_t || !dout || !lsew || !dqkv || !dbias_t || !make_geo(g, B, H, W, C, heads, ws, shift)) return SODT_EINVAL;
  hipStream_t st = (hipStream_t)st_;
  const int hd = C / heads;
  if (hd != 16) return SODT_EINVAL;
  if (dtype == SODT_BF16) return launch_bwd_wm<bf16, 16, 4>(qkvw, bias_t, dout, lsew, dqkv, dbias_t, g, st);
  if (dtype == SODT_F32) return launch_bwd_wm<float, 16, 2>(qkvw, bias_t, dout, lsew, dqkv, dbias_t, g, st);
  return SODT_EINVAL;
}

/* First stage of the fused W-MSA block's backward (bf16, C = 192, 12 heads, 8x8 windows): d(attention output) -> dqkv and the
 * relative-position-bias gradient, with q / k / v RECOMPUTED per (window, head) from the block's saved LayerNorm-1 output
 * `xn1` [M][C] and its parameter pack `wpk` (sodt_wmsa_pack) instead of read back from HBM: the forward (sodt_wmsa_block_fwd
 * with qkvw = NULL) then saves no q / k / v.  backbone_vit.py:968 (the qkv Linear) + :971-989 (its autograd). */
extern "C" int sodt_wmsa_block_bwd(const void* xn1, const void* wpk, const float* bias_t, const void* dout, const float* lsew,
                                   void* dqkv, float* dbias_t, int B, int H, int W, int C, int heads, int ws, int shift,
                                   int dtype, sodt_stream_t st_) {
  AttnGeo g;
  if (!xn1 || !wpk || !bias_t || !dout || !lsew || !dqkv || !dbias_t || !make_geo(g, B, H, W, C, heads, ws, shift)) return SODT_EINVAL;
  if (dtype != SODT_BF16) return SODT_EINVAL;            // the f32 parity path keeps its saved q / k / v (sodt_window_attn_bwd_wm)
  return launch_bwd_rc(xn1, (const unsigned char*)wpk, bias_t, dout, lsew, dqkv, dbias_t, g, (hipStream_t)st_);
}
