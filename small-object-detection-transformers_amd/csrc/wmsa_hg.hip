// Fused W-MSA / SW-MSA half of a Swin block for gfx950, bf16 throughput path: FOUR WAVES PER WINDOW.
//
//     x_mid = x + Proj( WindowAttention( LN1(x) ) )          backbone_vit.py:1088-1126, :961-992
//     xn2   = LN2(x_mid)                                     backbone_vit.py:1128 (the MLP's input)
//
// Same contract, parameter pack and saved-tensor layouts as wmsa_block.hip (which stays the f32 parity path); what
// changes is the work decomposition.  wmsa_block.hip gives one wave a whole window: a 24 KB LN1 tile per wave caps the
// workgroup at four waves = ONE wave per SIMD with a 512-register budget, so MFMA, softmax VALU and memory phases of a
// window run back to back and a lone wave issues one VALU instruction per 4 cycles.  Here a workgroup is EIGHT waves
// (two per SIMD, <= 256 registers each) working on TWO windows:
//
//   * the four waves of a window share its LN1 tile in LDS ([64][192] bf16, XOR-swizzled 16-byte chunks) and split the
//     HEADS: at step i = 0, 1, 2 wave j owns head 4 i + j end to end (q^T, k^T, v by MFMA from the shared tile, S^T = K Q^T,
//     relative-position bias from the strip-difference table, -100 shift mask, softmax in registers, O^T = V^T P^T) - the
//     operand chaining of wmsa_block.hip unchanged, nothing of a head ever leaves the wave's registers;
//   * token-major phases (LN1 prologue, residual + LN2 epilogue) split the TOKENS: wave j owns rows 16 j .. 16 j + 15;
//   * the output projection splits the OUTPUT CHANNELS: O^T of the 12 heads meets in the (dead) LN1 tile, wave j computes
//     out^T rows 48 j .. 48 j + 47 against it with the same three-fragment k-loop as the QKV phase, and the result goes
//     back through the tile for the token-major epilogue;
//   * weights: the Wq / Wk / Wv fragments of the four heads of a step (72 KB) sit in ONE LDS buffer shared by both
//     windows, refilled by LDS-DMA (global_load_lds_dwordx4) right after the step's QKV phase - the softmax / PV part of
//     the step hides the copy; Wproj (72 KB, natural k order) goes through the same buffer.  Relative-position tables
//     and q/k/v biases of all 12 heads (26 KB) stay resident.
//   * ~9 workgroup barriers per window pair; every global store drains in the background (counted vmcnt).
//
// LDS: 2 x 24 KB tiles + 72 KB weights + 25.5 KB tables + 3.75 KB vectors (+ 8 KB v patches when saving) = 157.25 KB.
#include "wmsa_common.h"

namespace {

constexpr int HG_TILE = 64 * 384;                       // one window's [64][192] bf16 tile
constexpr int HG_WBUF_OFF = 2 * HG_TILE;                // 49152
constexpr int HG_HEADW = 18432;                         // Wq | Wk | Wv fragments of one head (3 x 6 KB) = 3 Wproj strips
constexpr int HG_WBUF = 4 * HG_HEADW;                   // 73728
constexpr int HG_TAB_OFF = HG_WBUF_OFF + HG_WBUF;       // 122880
constexpr int HG_TABH = 1920 + 256;                     // table (4 shifted copies) + q/k/v bias of one head
constexpr int HG_LNV_OFF = HG_TAB_OFF + WHEADS * HG_TABH;   // 148992: bproj | g1 | b1 | g2 | b2 (f32)
constexpr int HG_VP_OFF = HG_LNV_OFF + 5 * WC * 4;      // 152832: per-wave 1 KB v transposition patch (SAVE)
constexpr int HG_LDS_INF = HG_VP_OFF, HG_LDS_SAVE = HG_VP_OFF + 8 * 1024;
static_assert(HG_LDS_SAVE <= 160 * 1024, "LDS budget");
static_assert(WL<bf16>::STAGE == 24576 && WL<bf16>::BIAS_OFF == HG_HEADW && WL<bf16>::BQKV_OFF == HG_HEADW + 1920, "pack layout");

typedef __attribute__((ext_vector_type(2))) float f32x2_;
// a - b on two f32 lanes in one VALU slot (hipcc scalarises a vector subtraction whose results feed v_exp_f32)
__device__ __forceinline__ f32x2_ pk_sub(const f32x2_& a, const f32x2_& b) {
  f32x2_ r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// STAMP: diagnostic build (sodt_debug_wmsa_hg_stamps): wave 0 of every workgroup sums shader cycles per phase
__device__ long long g_hg_stamps[512][12];     // rows 0..255: wave 0 (window A, older), 256..511: wave 4 (window B, same SIMD)
__device__ __forceinline__ long long hg_now() {
  long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define HG_STAMP(i) do { if constexpr (STAMP) { const long long now_ = hg_now(); acc_st[i] += now_ - last_st; last_st = now_; } } while (0)

template <bool SAVE, bool STAMP = false>
__global__ __launch_bounds__(512, 2) void wmsa_hg_kernel(const WArgs a) {
  typedef bf16 T;
  using L = WL<bf16>;
  constexpr int E = 2, KPL = 8, ROWB = 384;
  typedef uint2 k16_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  lds_u8* const sm3 = (lds_u8*)smem;
  const unsigned smem0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), ww = w >> 2, j = w & 3;
  const int t = lane & 15, g = lane >> 4;
  const unsigned tile = (unsigned)(ww * HG_TILE);
  // fragment addressing into the window's tile (see wmsa_block.hip: chunk c of row r sits at c ^ (r & 7))
  const unsigned gx3 = (unsigned)((g ^ (t & 3)) << 4), swb = (unsigned)(((t >> 2) & 1) * 64);
  const unsigned xrow = tile + (unsigned)(t * ROWB) + gx3;
  const unsigned xfE = smem0 + xrow + swb, xfO = smem0 + xrow - swb;
  const unsigned wb16 = smem0 + HG_WBUF_OFF + (unsigned)(j * HG_HEADW) + (unsigned)(lane * 16);
  const float scale2 = 0.25f * WMSA_LOG2E;
  // bias-table addressing of this lane (wmsa_block.hip): four consecutive entries at one aligned address, strip difference 0
  const int j0 = 7 - (t & 7) + 4 * (g & 1), jv = j0 & 3;
  const int bias_lane_off = (((jv * 15 + (t >> 3) - (g >> 1) + 7) * 16) + (j0 - jv)) * E;

  // ---- weights: LDS-DMA of 72 one-KB pieces, nine per wave.  stage 0..2: Wq|Wk|Wv fragments of heads 4 s .. 4 s + 3,
  // stage 3: Wproj in natural k order (WL::HGW_OFF)
  const int rot = (int)((blockIdx.x >> 3) * 7 + (blockIdx.x & 7) * 3) % 72;     // (blocks b, b + 8, ... share an XCD: distinct rotations)
  // pieces q0 .. q1 - 1 of this wave's nine: the issue of a 1 KB piece stalls the wave for 100+ cycles while the CU's address
  // path works through the burst, so the nine are spread over the VALU work of the phase that hides the copy
  auto dma_part = [&](int stage, int q0, int q1) {
#ifdef SODT_HG_ABLATE_DMA      // timing-only A/B build (tools/exp/ab_build.sh): never defined in the library build
    return;
#endif
    const unsigned char* gsrc = a.wpk + L::HGW_OFF + (size_t)stage * HG_WBUF;     // the stages are one contiguous stream
#pragma unroll
    for (int q = q0; q < q1; ++q) {
      // (the piece order is rotated per workgroup: every CU streams the same 72 KB, and walking it in the same order at the
      //  same time piles the requests of an XCD's 32 CUs onto one L2 channel after the other)
      int p = w + 8 * q + rot;
      p = p >= 72 ? p - 72 : p;
      const unsigned off = (unsigned)p << 10;
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                   :: "v"((unsigned)(lane * 16) + off), "s"(gsrc), "s"(smem0 + HG_WBUF_OFF + off) : "memory", "m0");
    }
  };
  auto dma_w = [&](int stage) { dma_part(stage, 0, 9); };
#define HG_VMWAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")

  dma_w(0);
  // resident: tables + q/k/v bias of the 12 heads, projection bias, LayerNorm vectors
  for (int i = tid; i < WHEADS * (HG_TABH / 16); i += 512) {
    const int h = i / (HG_TABH / 16), c = i % (HG_TABH / 16);
    ((uint4*)(smem + HG_TAB_OFF + h * HG_TABH))[c] = ((const uint4*)(a.wpk + (size_t)h * L::STAGE + L::BIAS_OFF))[c];
  }
  for (int i = tid; i < 5 * WC / 4; i += 512) ((float4*)(smem + HG_LNV_OFF))[i] = ((const float4*)(a.wpk + L::TAIL_OFF))[i];

  const int npairs = (a.nwin + 1) / 2;
  // token-major phases: this lane's row (token 16 j + t of the window) and its six chunks 4 i + g
  auto row_of = [&](int pair) {
    int item = 2 * pair + ww;
    if (item >= a.nwin) item = a.nwin - 1;
    const int wx_ = item % a.nwx; item /= a.nwx;
    const int wy_ = item % a.nwy; const int b_ = item / a.nwy;
    return (unsigned)wtoken(a, b_, wy_, wx_, 16 * j + t);
  };
  uint4 xc[6], xnext[6];      // x of this wave's 16 tokens: the current pair's (LN1 input AND residual) and the next pair's
  if ((int)blockIdx.x < npairs) {
    const unsigned ro = row_of(blockIdx.x) * (unsigned)ROWB + (unsigned)(g * 16);
#pragma unroll
    for (int i = 0; i < 6; ++i) xc[i] = *(const uint4*)(a.x + (ro + 64u * i));
  }
  // Per-lane addresses of the token-major phases, the tile hand-overs and the output rows are re-derived from a LAUNDERED lane
  // id where they are used: as loop invariants they would stay live across the head steps - hipcc spills them - while
  // re-deriving costs a few VALU instructions per window pair.
  // token-major phases: this lane's LDS row (token 16 j + t of the window), chunk 4 i + g: even i at +64 b, odd i at -64 b (+ 64 i)
#define HG_TOKEN_PTRS()                                                                                   \
  int ll_ = lane; LAUNDER(ll_);                                                                           \
  const int tl = ll_ & 15, gl = ll_ >> 4;                                                                 \
  const unsigned trow_ = tile + (unsigned)((16 * j + tl) * ROWB) + (unsigned)((gl ^ (tl & 3)) << 4);      \
  const unsigned swb_ = (unsigned)(((tl >> 2) & 1) * 64);                                                 \
  lds_u8* const p_rE = sm3 + trow_ + swb_; lds_u8* const p_rO = sm3 + trow_ - swb_;                       \
  lds_u8* const p_ln = sm3 + HG_LNV_OFF + gl * KPL * 4;                                                   \
  const unsigned myrow = (unsigned)wtoken(a, b, wy, wx, 16 * j + tl);                                     \
  const unsigned myoff = myrow * (unsigned)ROWB + (unsigned)(gl * 16)
  // bias tables of this wave's heads 4 step + j: one base per lane, the step is an immediate offset
  const unsigned tbl = smem0 + HG_TAB_OFF + (unsigned)(j * HG_TABH);
  const unsigned bb3 = tbl + (unsigned)bias_lane_off - (unsigned)(3 * 2 * 16 * E);
  const unsigned sbg = tbl + 1920u + (unsigned)(16 * g), sbt = tbl + 1920u + (unsigned)(4 * t);
  __syncthreads();
  long long acc_st[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, last_st = 0;
  if constexpr (STAMP) last_st = hg_now();

  for (int it = blockIdx.x; it < npairs; it += gridDim.x) {
    int item = 2 * it + ww;
    const bool valid = item < a.nwin;
    if (!valid) item = a.nwin - 1;
    int tq = item;
    const int wx = tq % a.nwx; tq /= a.nwx;
    const int wy = tq % a.nwy; const int b = tq / a.nwy;
    const bool msk = a.shift > 0 && (wy == a.nwy - 1 || wx == a.nwx - 1);
    const unsigned whoff = (unsigned)item * WHEADS;
    unsigned diffm[4] = {0u, 0u, 0u, 0u};
    if (msk) {
      int kr[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) kr[i] = wrid(a, wy, wx, 16 * (i >> 2) + 4 * g + (i & 3));
#pragma unroll
      for (int ms = 0; ms < 4; ++ms) {
        const int qr = wrid(a, wy, wx, 16 * ms + t);
#pragma unroll
        for (int i = 0; i < 16; ++i) diffm[ms] |= (qr != kr[i] ? 1u : 0u) << i;
      }
    }

    // ================= prologue: LN1 of this wave's 16 tokens -> tile rows 16 j .. 16 j + 15
    {
      HG_TOKEN_PTRS();
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        float f[KPL];
        unpack<T>(xc[i], f);
#pragma unroll
        for (int k = 0; k < KPL; ++k) s += f[k];
      }
      s = rows_sum(s);
      const float mu = s * (1.0f / WC);
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        float f[KPL];
        unpack<T>(xc[i], f);
#pragma unroll
        for (int k = 0; k < KPL; ++k) { const float d = f[k] - mu; q = fmaf(d, d, q); }
      }
      q = rows_sum(q);
      const float rstd = rsqrtf(q * (1.0f / WC) + 1e-5f);
      if (SAVE && valid && gl == 0) *(float2*)((unsigned char*)a.st1 + myrow * 8u) = make_float2(mu, rstd);
      wave_sync();                                       // this wave's epilogue reads of the same rows are done
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        float ga[KPL], be[KPL], f[KPL];
#pragma unroll
        for (int k = 0; k < KPL; k += 4) {
          *(f32x4*)(ga + k) = *(const __attribute__((address_space(3))) f32x4*)(p_ln + (WC + 4 * i * KPL + k) * 4);
          *(f32x4*)(be + k) = *(const __attribute__((address_space(3))) f32x4*)(p_ln + (2 * WC + 4 * i * KPL + k) * 4);
        }
        unpack<T>(xc[i], f);
#pragma unroll
        for (int k = 0; k < KPL; ++k) f[k] = fmaf((f[k] - mu) * rstd, ga[k], be[k]);
        const uint4 y = pack<T>(f);
        *(__attribute__((address_space(3))) u32x4_*)(((i & 1) ? p_rO : p_rE) + 64 * i) = u32x4_{y.x, y.y, y.z, y.w};
        if (SAVE && valid) *(uint4*)(a.xn1 + (myoff + 64u * i)) = y;
      }
    }
    // W(0) was requested before the previous pair's epilogue stores (or at kernel start): everything older than the
    // youngest stores has landed.  The stores themselves stay in flight.
    HG_STAMP(0);
    // (inference: the 12 x_mid / xn2 stores of the previous pair; training: + its LN2 statistics and this pair's LN1 statistics
    //  and xn1 rows, 20 in all - a wave of a clamped tail window issued fewer and waits for everything)
    if (!SAVE) HG_VMWAIT(12);
    else if (valid) HG_VMWAIT(20);
    else HG_VMWAIT(0);
    __syncthreads();                                     // B1: tiles complete, W(0) visible
    HG_STAMP(1);

    k16_t poall[3][4];
    static_for<0, 3>([&](auto i_) {
      constexpr int step = decltype(i_)::value;
      const int h = 4 * step + j;
      constexpr int TBO = step * 4 * HG_TABH;            // table / bias of head 4 step + j relative to the lane bases
      typedef typename KR<T>::type kreg_t;
      // ---- q^T, k^T (channel rows, token columns) and v (token rows, channel columns) of head h
      u32x4_ bqr = lds_rd128a<TBO>(sbg), bkr = lds_rd128a<TBO + 64>(sbg);
      unsigned bvr = lds_rd32a<TBO + 128>(sbt);
      // fragments are single-buffered: the reads of k-step kk + 1 are issued right after the MFMAs of k-step kk (which
      // have taken their operands) and land while those execute; the SIMD's second wave fills what is left of the gap
      u32x4_ wf[3], xf[4];
      auto issue_k = [&](auto kk_) {
        constexpr int kk = decltype(kk_)::value;
        wf[0] = lds_rd128a<0 + kk * 1024>(wb16);
        wf[1] = lds_rd128a<6144 + kk * 1024>(wb16);
        wf[2] = lds_rd128a<12288 + kk * 1024>(wb16);
        static_for<0, 4>([&](auto ms_) {
          constexpr int ms = decltype(ms_)::value;
          xf[ms] = lds_rd128a<ms * 16 * ROWB + 64 * kk>((kk & 1) ? xfO : xfE);
        });
      };
      issue_k(std::integral_constant<int, 0>{});
      k16_t pq[4], pqs[4], pkk[4], pv[4];
      f32x4 qT[4], kT[4], vv[4];
      static_for<0, 6>([&](auto kk_) {
        constexpr int kk = decltype(kk_)::value;
        LDS_WAIT(0);
        LDS_DEP(wf[0]); LDS_DEP(wf[1]); LDS_DEP(wf[2]);
        LDS_DEP(xf[0]); LDS_DEP(xf[1]); LDS_DEP(xf[2]); LDS_DEP(xf[3]);
        if constexpr (kk == 0) {
          LDS_DEP(bqr); LDS_DEP(bkr); LDS_DEP(bvr);
          const f32x4 bqv = KR<float>::f4(bqr), bkv = KR<float>::f4(bkr);
          const float bvs = __uint_as_float(bvr);
#pragma unroll
          for (int ms = 0; ms < 4; ++ms) { qT[ms] = bqv; kT[ms] = bkv; vv[ms] = f32x4{bvs, bvs, bvs, bvs}; }
        }
        const uint4 wq = u4(wf[0]), wk = u4(wf[1]), wv = u4(wf[2]);
#pragma unroll
        for (int ms = 0; ms < 4; ++ms) {
          const uint4 x4 = u4(xf[ms]);
          mma16<T>(qT[ms], wq, x4);
          mma16<T>(kT[ms], wk, x4);
          mma16<T>(vv[ms], x4, wv);
        }
        if constexpr (kk + 1 < 6) issue_k(std::integral_constant<int, kk + 1>{});
      });
      // the head's bias-table entries (7 x 4 per lane): requested now, used after the hand-over barrier
      kreg_t biar[7];
      static_for<0, 7>([&](auto d_) { constexpr int d = decltype(d_)::value; biar[d] = KR<T>::template rd<TBO + d * 2 * 16 * E>(bb3); });
#pragma unroll
      for (int ms = 0; ms < 4; ++ms) {
        if (SAVE) pq[ms] = pk16<T>(qT[ms]);
        pqs[ms] = pk16<T>(qT[ms] * scale2);              // hd^-1/2 x log2 e folded into q: S^T leaves the MFMA ready for exp2
        pkk[ms] = pk16<T>(kT[ms]); pv[ms] = pk16<T>(vv[ms]);
      }
      HG_STAMP(2);
      __syncthreads();                                   // B2/4/6: everyone is done with the weight buffer (step 2: and the LN1 tile)
      HG_STAMP(3);
      dma_part(step + 1, 0, 3);                          // next heads' weights / Wproj land under the softmax (pieces 3..8: inside it)
      if (SAVE && valid) {
        unsigned char* qb = a.qkvw + (size_t)(whoff + h) * (3 * 64 * WHD * E);     // uniform
        const unsigned lo = (unsigned)(t * (WHD * E) + g * 8);
#pragma unroll
        for (int ms = 0; ms < 4; ++ms) {
          *(k16_t*)(qb + (lo + (unsigned)(16 * ms * WHD * E))) = pq[ms];
          *(k16_t*)(qb + (lo + (unsigned)(64 * WHD * E + 16 * ms * WHD * E))) = pkk[ms];
        }
        // v -> [token][16] through a 1 KB per-wave patch, two token strips at a time (the accumulator holds four tokens of
        // ONE channel per lane)
        lds_u8* const vp = sm3 + HG_VP_OFF + w * 1024;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          wave_sync();
#pragma unroll
          for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              *(__attribute__((address_space(3))) T*)(vp + ((16 * m2 + 4 * g + r) * WHD + t) * E) = from_f<T>(vv[2 * half + m2][r]);
          wave_sync();
          *(uint4*)(qb + (unsigned)(2 * 64 * WHD * E + half * 1024 + lane * 16)) =
              u4((u32x4_)*(const __attribute__((address_space(3))) u32x4_*)(vp + lane * 16));
        }
      }

      HG_STAMP(4);
      auto body = [&](auto MSK_) {
        constexpr bool MSK = decltype(MSK_)::value;
        // ---- S^T = K Q^T: row = key 16 ks + 4 g + r, column = query 16 ms + t; the bias is the accumulator's initial value
        f32x4 bia[7];
        LDS_WAIT(0);
#pragma unroll
        for (int d = 0; d < 7; ++d) { LDS_DEP(biar[d]); bia[d] = KR<T>::f4(biar[d]); }
        k16_t pp[4][4];                                  // P^T strips, packed: [ks][ms]
        float inv[4];
#pragma unroll
        for (int ms = 0; ms < 4; ++ms) {
          f32x4 s[4];
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) { s[ks] = bia[ms - ks + 3]; mmak16(s[ks], pkk[ks], pqs[ms]); }
          float mx = -1e30f;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if constexpr (MSK) { if ((diffm[ms] >> (4 * ks + r)) & 1u) s[ks][r] += -100.0f * WMSA_LOG2E; }
              mx = fmaxf(mx, s[ks][r]);
            }
          mx = rows_max(mx);
          // (vector forms: hipcc turns the subtraction and the running sums into v_pk_add_f32, two elements per VALU slot -
          //  the SIMD issues one wave64 f32 instruction per 4 cycles whatever the number of waves)
          f32x4 sum4 = f32x4{0.f, 0.f, 0.f, 0.f};
          const f32x2_ mx2 = f32x2_{mx, mx};
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const f32x2_ d0 = pk_sub(f32x2_{s[ks][0], s[ks][1]}, mx2), d1 = pk_sub(f32x2_{s[ks][2], s[ks][3]}, mx2);
            s[ks] = f32x4{__builtin_amdgcn_exp2f(d0[0]), __builtin_amdgcn_exp2f(d0[1]), __builtin_amdgcn_exp2f(d1[0]), __builtin_amdgcn_exp2f(d1[1])};
            sum4 += s[ks];
            pp[ks][ms] = pk16<T>(s[ks]);
          }
          float sum = (sum4[0] + sum4[1]) + (sum4[2] + sum4[3]);
          sum = rows_sum(sum);
          inv[ms] = __builtin_amdgcn_rcpf(sum);
          if (ms < 3) dma_part(step + 1, 3 + 2 * ms, 5 + 2 * ms);
          if (SAVE && valid && g == 0)
            (a.lsew + (size_t)(whoff + h) * 64)[16 * ms + t] = mx * (1.0f / WMSA_LOG2E) + __logf(sum);
        }
        // ---- O^T = V^T P^T: row = channel 4 g + r, column = query
#pragma unroll
        for (int ms = 0; ms < 4; ++ms) {
          f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kp = 0; kp < 2; ++kp)
            mma16<bf16>(o, make_uint4(pv[2 * kp].x, pv[2 * kp].y, pv[2 * kp + 1].x, pv[2 * kp + 1].y),
                        make_uint4(pp[2 * kp][ms].x, pp[2 * kp][ms].y, pp[2 * kp + 1][ms].x, pp[2 * kp + 1][ms].y));
          o *= inv[ms];
          poall[step][ms] = pk16<T>(o);
        }
      };
      if (msk) body(std::true_type{}); else body(std::false_type{});
      HG_STAMP(5);

      if constexpr (step < 2) {
        HG_VMWAIT(0);                                    // this wave's pieces of the next weights have landed
        __syncthreads();                                 // B3/5
        HG_STAMP(6);
      }
    });

    // ================= O^T of this wave's three heads -> the (dead) LN1 tile, now the attention-output tile [64][192]
    // (every wave passed B6 after its last QKV phase: nobody reads LN1 rows any more)
    {
      // head 4 step + j, channels 4 g .. 4 g + 3 of token (ms, t): chunk 2 h + (g >> 1) = 8 step + (2 j + (g >> 1)), stored
      // at chunk ^ (t & 7): the step is a +128-byte immediate
      int ll_ = lane; LAUNDER(ll_);
      const int tl = ll_ & 15, gl = ll_ >> 4;
      lds_u8* const p_ao = sm3 + tile + tl * ROWB + (((2 * j + (gl >> 1)) ^ (tl & 7)) << 4) + 8 * (gl & 1);
#pragma unroll
      for (int step = 0; step < 3; ++step)
#pragma unroll
        for (int ms = 0; ms < 4; ++ms)
          *(__attribute__((address_space(3))) u32x2_*)(p_ao + ms * 16 * ROWB + 128 * step) = u32x2_{poall[step][ms].x, poall[step][ms].y};
    }
    HG_VMWAIT(0);                                        // Wproj has landed
    __syncthreads();                                     // B7: attention-output tiles complete, Wproj visible
    HG_STAMP(7);

    if (SAVE && valid) {                                 // attention output, natural token order (operand of the dWproj GEMM)
      HG_TOKEN_PTRS();
#pragma unroll
      for (int i = 0; i < 6; ++i)
        *(uint4*)(a.ao + (myoff + 64u * i)) = u4((u32x4_)*(const __attribute__((address_space(3))) u32x4_*)(((i & 1) ? p_rO : p_rE) + 64 * i));
    }
    // ================= output projection: out^T rows 48 j .. 48 j + 47 (three 16-row strips) x 64 tokens, K = 192
    f32x4 oT[3][4];
    {
      u32x4_ wf[3], xf[4];
      auto issue_k = [&](auto kk_) {
        constexpr int kk = decltype(kk_)::value;
        wf[0] = lds_rd128a<0 + kk * 1024>(wb16);
        wf[1] = lds_rd128a<6144 + kk * 1024>(wb16);
        wf[2] = lds_rd128a<12288 + kk * 1024>(wb16);
        static_for<0, 4>([&](auto ms_) {
          constexpr int ms = decltype(ms_)::value;
          xf[ms] = lds_rd128a<ms * 16 * ROWB + 64 * kk>((kk & 1) ? xfO : xfE);
        });
      };
      // projection bias of this lane's channels 16 (3 j + nl) + 4 g .. + 3: the accumulators' initial value
      int llp = lane; LAUNDER(llp);
      const unsigned pbb = smem0 + HG_LNV_OFF + (unsigned)((48 * j + 4 * (llp >> 4)) * 4);
      u32x4_ pbr[3];
      pbr[0] = lds_rd128a<0>(pbb); pbr[1] = lds_rd128a<64>(pbb); pbr[2] = lds_rd128a<128>(pbb);
      issue_k(std::integral_constant<int, 0>{});
      static_for<0, 6>([&](auto kk_) {
        constexpr int kk = decltype(kk_)::value;
        LDS_WAIT(0);
        LDS_DEP(wf[0]); LDS_DEP(wf[1]); LDS_DEP(wf[2]);
        LDS_DEP(xf[0]); LDS_DEP(xf[1]); LDS_DEP(xf[2]); LDS_DEP(xf[3]);
        if constexpr (kk == 0) {
          LDS_DEP(pbr[0]); LDS_DEP(pbr[1]); LDS_DEP(pbr[2]);
#pragma unroll
          for (int nl = 0; nl < 3; ++nl)
#pragma unroll
            for (int ms = 0; ms < 4; ++ms) oT[nl][ms] = KR<float>::f4(pbr[nl]);
        }
#pragma unroll
        for (int nl = 0; nl < 3; ++nl) {
          const uint4 wa = u4(wf[nl]);
#pragma unroll
          for (int ms = 0; ms < 4; ++ms) mma16<T>(oT[nl][ms], wa, u4(xf[ms]));
        }
        if constexpr (kk + 1 < 6) issue_k(std::integral_constant<int, kk + 1>{});
      });
    }
    HG_STAMP(8);
    __syncthreads();                                     // B8: everyone is done with the attention-output tile and Wproj
    HG_STAMP(9);

    // the first weights and the x rows of the next pair: requested before the output stores (a load queued behind a store
    // waits for the store's acknowledgement), consumed by the next prologue
    const bool more = it + (int)gridDim.x < npairs;
    {
      const int nx = it + (int)gridDim.x < npairs ? it + (int)gridDim.x : it;
      const unsigned ro = row_of(nx) * (unsigned)ROWB + (unsigned)(g * 16);
#pragma unroll
      for (int i = 0; i < 6; ++i) xnext[i] = *(const uint4*)(a.x + (ro + 64u * i));
    }
    // out^T (+ bias) -> tile, run dtype: the rounding a separate projection launch applies to its output
    {
      int ll_ = lane; LAUNDER(ll_);
      const int tl = ll_ & 15, gl = ll_ >> 4;
      lds_u8* const p_t = sm3 + tile + tl * ROWB + 8 * (gl & 1);
#pragma unroll
      for (int nl = 0; nl < 3; ++nl) {
        if (more) dma_part(0, 3 * nl, 3 * nl + 3);
        const int cw = 2 * (3 * j + nl) + (gl >> 1);
        lds_u8* const p_o = p_t + ((cw ^ (tl & 7)) << 4);
#pragma unroll
        for (int ms = 0; ms < 4; ++ms) {
          const k16_t v = pk16<T>(oT[nl][ms]);
          *(__attribute__((address_space(3))) u32x2_*)(p_o + ms * 16 * ROWB) = u32x2_{v.x, v.y};
        }
      }
    }
    __syncthreads();                                     // B9: output tiles complete
    HG_STAMP(10);

    // ================= epilogue: x_mid = x + (out + bproj), xn2 = LN2(x_mid) for this wave's 16 tokens
    {
      HG_TOKEN_PTRS();
      float v[6][KPL];
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        float f[KPL], o[KPL];
        unpack<T>(xc[i], f);                             // residual: the x this wave loaded for LN1, still in registers
        unpack<T>(u4((u32x4_)*(const __attribute__((address_space(3))) u32x4_*)(((i & 1) ? p_rO : p_rE) + 64 * i)), o);
        // x_mid is stored in bf16: LN2 normalises the ROUNDED value, as a separate LayerNorm launch reading x_mid would
#pragma unroll
        for (int k = 0; k < KPL; ++k) { v[i][k] = to_f(from_f<T>(f[k] + o[k])); s += v[i][k]; }
      }
      s = rows_sum(s);
      const float mu = s * (1.0f / WC);
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int k = 0; k < KPL; ++k) { const float d = v[i][k] - mu; q = fmaf(d, d, q); }
      q = rows_sum(q);
      const float rs = rsqrtf(q * (1.0f / WC) + 1e-5f);
#ifdef SODT_HG_ABLATE_STORES   // timing-only A/B build (tools/exp/ab_build.sh): never defined in the library build
      if (false) {
#else
      if (valid) {
#endif
        if (SAVE && gl == 0) *(float2*)((unsigned char*)a.st2 + myrow * 8u) = make_float2(mu, rs);
#pragma unroll
        for (int i = 0; i < 6; ++i) *(uint4*)(a.xm + (myoff + 64u * i)) = pack<T>(v[i]);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          float f[KPL], ga[KPL], be[KPL];
#pragma unroll
          for (int k = 0; k < KPL; k += 4) {
            *(f32x4*)(ga + k) = *(const __attribute__((address_space(3))) f32x4*)(p_ln + (3 * WC + 4 * i * KPL + k) * 4);
            *(f32x4*)(be + k) = *(const __attribute__((address_space(3))) f32x4*)(p_ln + (4 * WC + 4 * i * KPL + k) * 4);
          }
#pragma unroll
          for (int k = 0; k < KPL; ++k) f[k] = fmaf((v[i][k] - mu) * rs, ga[k], be[k]);
          *(uint4*)(a.xn2 + (myoff + 64u * i)) = pack<T>(f);
        }
      } else {
        HG_VMWAIT(0);                                    // (no stores were issued: the counted wait of the next prologue must not run short)
      }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) xc[i] = xnext[i];
    HG_STAMP(11);
  }
  HG_VMWAIT(0);
  if constexpr (STAMP) {
    if ((tid == 0 || tid == 256) && blockIdx.x < 256)
      for (int i = 0; i < 12; ++i) g_hg_stamps[blockIdx.x + (tid ? 256 : 0)][i] = acc_st[i];
  }
}

bool g_hg_stamp_enable = false;

template <bool SAVE, bool STAMP = false>
int hg_launch(const WArgs& a, hipStream_t st) {
  constexpr int LDS = SAVE ? HG_LDS_SAVE : HG_LDS_INF;
  static bool attr_set = false;
  auto kern = wmsa_hg_kernel<SAVE, STAMP>;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
      (void)hipGetLastError();
      return SODT_EINVAL;
    }
    attr_set = true;
  }
  const int npairs = (a.nwin + 1) / 2;
  const int grid = npairs < 256 ? npairs : 256;          // one workgroup per CU, persistent over the window pairs
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, st, a);
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

}  // namespace

int wmsa_hg_launch(const WArgs& a, bool save, hipStream_t st) {
  if (g_hg_stamp_enable) return save ? hg_launch<true, true>(a, st) : hg_launch<false, true>(a, st);
  return save ? hg_launch<true>(a, st) : hg_launch<false>(a, st);
}

/* diagnostic hook (tools/mb_wmsa.py --hg-stamps): enable != 0 makes the following bf16 launches run the instrumented build; out
 * (host, 512 x 12 long long, nullable: rows 0..255 wave 0, 256..511 wave 4) receives the per-phase shader-cycle sums of wave 0 of each workgroup of the last such
 * launch: [LN1, B1 wait, QKV, B2/4/6 wait, dma issue + saves, softmax + PV, B3/5 wait, O^T -> tile + B7, projection, B8 wait,
 * residual loads + staging + B9, epilogue] */
extern "C" int sodt_debug_wmsa_hg_stamps(long long* out, int enable) {
  g_hg_stamp_enable = enable != 0;
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hg_stamps), sizeof(long long) * 512 * 12) != hipSuccess) return SODT_EINVAL;
  return SODT_OK;
}
