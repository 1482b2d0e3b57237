// Pipelined bf16 NT GEMM for the long-contraction layers (K >= 384: stage-2/3 linears, the 2x2-conv MLPs,
// PatchMerging, the dX GEMMs of the wide layers):   C[M][N] = epilogue( A[M][K] @ W[N][K]^T ).
//
// Structure (gfx950, one workgroup of 8 waves per CU, persistent over output tiles):
//   * 256(m) x 192(n) output tile, K-step 64 (128-byte rows); waves 4(m) x 2(n), 64 x 96 per wave.
//   * Operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR staging): A (the activation
//     rows, HBM-streamed, gathered through the K-segment row map) three K-steps deep, W (L2-resident) two
//     deep.  The rings run across tile boundaries, so the next tile's first K-steps land while the current
//     tile's epilogue stores.
//   * One raw s_barrier per K-step; DMA completion is a counted s_waitcnt vmcnt(4) (the newest A stage stays
//     in flight across the barrier).  Fragment reads are inline-asm ds_read_b128: hipcc waits vmcnt(0) before
//     any LDS read it can see while an LDS-DMA is outstanding, which would serialise the pipeline.
//   * LDS images are XOR-swizzled on the 16-byte chunk (source-side swizzle of the DMA address + the same
//     involution on the read address), conflict-free for both fragment shapes.
//   * The MFMA operands are swapped (W rows on the accumulator rows, permuted inside each 32-row group), so
//     each lane ends with 8 consecutive output columns of one row: the fused epilogue (gemm_epi.h) runs from
//     registers with 16-byte stores, no LDS staging.
// Roofline: MFMA-bound shapes (AI = 2NK/(2(K_in + N)) >= 300 flop/B for K >= 768); LDS traffic per K-step
// 160 KB reads + 56 KB DMA writes per CU against 1536 MFMA cycles per SIMD.
#include "gemm_epi.h"

// scheduling experiments (tools/exp/ab_gemm3.sh): SODT_EXP_PRIO 1 = s_setprio(1) around every MFMA cluster (keeps hipcc from
// moving MFMAs across the raw barriers: cdna_hip_programming.md T5); 2 = static priority for the younger half of the workgroup
#ifndef SODT_EXP_PRIO
#define SODT_EXP_PRIO 0
#endif
#if SODT_EXP_PRIO == 1
#define T3_PRIO_HI() __builtin_amdgcn_s_setprio(1)
#define T3_PRIO_LO() __builtin_amdgcn_s_setprio(0)
#else
#define T3_PRIO_HI() do {} while (0)
#define T3_PRIO_LO() do {} while (0)
#endif

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int T3_BM = 256, T3_BN = 192, T3_BK = 64;
constexpr int T3_AST = T3_BM * 128;                 // 32 KiB per A stage
constexpr int T3_BST = T3_BN * 128;                 // 24 KiB per W stage
constexpr int T3_BOFF = 3 * T3_AST;                 // W ring after the A ring
constexpr int T3_SEGOFF = T3_BOFF + 2 * T3_BST;     // segment table (48 B per entry)
constexpr int T3_BIASOFF = T3_SEGOFF + 448;         // f32 bias[N], N <= 3072 (read with inline-asm LDS loads in the epilogue:
constexpr int T3_MAXBIAS = 3072;                    //  an ordinary global load there would drain the DMA queue every tile)
constexpr int T3_LDS = T3_BIASOFF + T3_MAXBIAS * 4;

// DMA source of rows outside the image / beyond M.  8 KiB: such a row's source pointer advances by one K-step (128 B) per stage like
// every other row's, up to the end of its K-segment (klen <= T3_MAXKLEN elements)
__device__ uint4 g_zero16[512];
constexpr int T3_MAXKLEN = (8192 - 256) / 2;

__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(lds_void*)p; }

template <int OFF> __device__ __forceinline__ u32x4 lds_rd128(uint32_t addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ uint2 lds_rd64(uint32_t addr) {
  uint2 v;
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
  u32x2 t;
  asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"(addr));
  v.x = t.x; v.y = t.y;
  return v;
}
#define T3_LGKM0() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

__device__ __forceinline__ void mma_sw(f32x4& acc, const u32x4& w, const u32x4& a) {
  union { u32x4 u; bf16x8 v; } uw, ua;
  uw.u = w; ua.u = a;
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(uw.v, ua.v, acc, 0, 0, 0);
}

struct Seg3 { const unsigned char* p; int ld, klen, dy, dx, mul, shr, Hi, Wi; };

__device__ __forceinline__ Seg3 load_seg3(uint32_t table, int i) {
  const uint32_t a = table + 48 * i;
  const uint2 p = lds_rd64(a);
  const u32x4 q0 = lds_rd128<16>(a), q1 = lds_rd128<32>(a);
  T3_LGKM0();
  Seg3 s;
  s.p = (const unsigned char*)(((uint64_t)p.y << 32) | p.x);
  s.ld = (int)q0.x; s.klen = (int)q0.y; s.dy = (int)q0.z; s.dx = (int)q0.w;
  s.mul = (int)q1.x; s.shr = (int)q1.y; s.Hi = (int)q1.z; s.Wi = (int)q1.w;
  return s;
}

// the same entry for the whole wave (i uniform): kept in SGPRs - the NT K-loop compares against klen every stage
__device__ __forceinline__ Seg3 load_seg3_uniform(uint32_t table, int i) {
  const Seg3 v = load_seg3(table, i);
  auto sc = [](int x) { return (int)__builtin_amdgcn_readfirstlane(x); };
  Seg3 s;
  const uint64_t pv = (uint64_t)(uintptr_t)v.p;
  s.p = (const unsigned char*)(((uint64_t)(unsigned)sc((int)(pv >> 32)) << 32) | (unsigned)sc((int)pv));
  s.ld = sc(v.ld); s.klen = sc(v.klen); s.dy = sc(v.dy); s.dx = sc(v.dx);
  s.mul = sc(v.mul); s.shr = sc(v.shr); s.Hi = sc(v.Hi); s.Wi = sc(v.Wi);
  return s;
}

__device__ __forceinline__ long seg3_row(const Seg3& s, bool ok, int b, int y, int x, int spatial, long m) {
  if (!ok) return -1;
  if (!spatial) return m;
  const int yy = y * s.mul + s.dy, xx = x * s.mul + s.dx;
  if (yy < 0 || xx < 0) return -1;
  const int yi = yy >> s.shr, xi = xx >> s.shr;
  if (yi >= s.Hi || xi >= s.Wi) return -1;
  return ((long)b * s.Hi + yi) * s.Wi + xi;
}

// OSC: the output rows are scattered (PatchMerging backward: row m of the half-resolution grid -> one of the four
// stride-2 positions of the full grid); only with CF == 0
// NV < 3 ("thin" instantiations, N <= 32 NV, one column tile): only the waves wc = 0 hold real columns, in their first NV 32-column
// groups - the other fragment reads and MFMAs do not exist in the instantiation (a run-time skip cost 31-397 spilled registers).
template <int CF, bool OSC = false, int NV = 3>
__global__ __launch_bounds__(512) void gemm_nt3_kernel(const sodt_gemm_args g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: LDS-DMA destinations stay in SGPRs)
  const int wr = wid >> 1, wc = wid & 1;
  const int fi = lane & 15, fg = lane >> 4;
  const uint32_t lbase = lds_addr(dsm);
#if SODT_EXP_PRIO == 2
  if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);
#endif

  if (tid == 0) {                  // own padded copy of the segment table (16-byte aligned fields)
#pragma unroll
    for (int j = 0; j < SODT_MAX_SEG; ++j) {
      unsigned char* e = dsm + T3_SEGOFF + 48 * j;
      *(uint64_t*)(e) = (uint64_t)(uintptr_t)g.a.s[j].p;
      *(int4*)(e + 16) = make_int4(g.a.s[j].ld, g.a.s[j].klen, g.a.s[j].dy, g.a.s[j].dx);
      *(int4*)(e + 32) = make_int4(g.a.s[j].mul, g.a.s[j].shr, g.a.s[j].Hi, g.a.s[j].Wi);
    }
  }
  if (CF & SODT_EPI_BIAS)
    for (int i = tid; i < g.N; i += 512) *(float*)(dsm + T3_BIASOFF + 4 * i) = g.bias[i];
  __syncthreads();
  const uint32_t segtab = lbase + T3_SEGOFF;

  const int ntn = (g.N + T3_BN - 1) / T3_BN;          // N % 8 == 0; the last column tile may be partial (rows >= N of W read as zeros,
  const int ntm = (g.M + T3_BM - 1) / T3_BM;          //  their output chunks not stored)
  const int ntiles = ntm * ntn;
  const int G = gridDim.x;
  const int lw = xcd_remap(blockIdx.x, G);
  const int nt_my = lw < ntiles ? (ntiles - lw + G - 1) / G : 0;
  const int nk = g.K / T3_BK;
  const int total = nt_my * nk;
  if (total == 0) return;
  const int hw = g.a.Ho * g.a.Wo;
  const int spatial = g.a.spatial;
  const unsigned char* zero = (const unsigned char*)g_zero16;

  // ---- A issue state: 4 rows per thread and stage (wave-instruction q covers tile rows 32 wid + 8 q .. + 7).  The K-loop
  //      carries one source POINTER per row (re-derived only when the tile or the K-segment changes) and advances it by one
  //      K-step per stage: one 64-bit VALU add per DMA instruction (the row * ld + offset form cost ~10, and the two waves of
  //      a SIMD run their address arithmetic and their MFMAs in the same phases: pmc r04, VALU 3.3 per MFMA on a plain GEMM)
  const int arow0 = 32 * wid + (lane >> 3);                    // + 8 q
  const int acb0 = ((lane & 7) ^ ((lane >> 4) & 3)) << 4;      // swizzled source chunk (bytes), q even
  const int acb1 = ((lane & 7) ^ (4 + ((lane >> 4) & 3))) << 4;   // q odd
  int a_ord = 0, a_kt = 0, a_seg = 0, a_off = 0;
  Seg3 aseg = load_seg3_uniform(segtab, 0);
  int gb[4], gy[4], gx[4]; bool gok[4];
  const unsigned char* ap[4];
  // "same-grid taps" (a 3x3 / 2x2 convolution over ONE tensor: every K-segment is the same [rows][ld] view on the output's own
  // grid, only (dy, dx) differ): the source row of tap (dy, dx) is the output row + dy * Wo + dx, so a segment change is a uniform
  // pointer offset plus two bounds tests per row - not a table read (LDS round trip) and a row computation per row.  With 64-channel
  // taps the segment changes EVERY K-step: the generic form cost ~4,800 cycles per K-step against ~3,800 for a plain A stream.
  bool taps_fast = spatial != 0;
  for (int j = 0; j < g.a.nseg; ++j)
    taps_fast = taps_fast && g.a.s[j].p == g.a.s[0].p && g.a.s[j].ld == g.a.s[0].ld && g.a.s[j].klen == g.a.s[0].klen &&
                g.a.s[j].mul == 1 && g.a.s[j].shr == 0 && g.a.s[j].Hi == g.a.Ho && g.a.s[j].Wi == g.a.Wo;
  const unsigned char* acen = nullptr;         // taps_fast: source address of this lane's row 0 of the tile at tap (0, 0), k = 0
  auto a_ptrs_fast = [&](int dy, int dx) {
    const long toff = ((long)(dy * g.a.Wo + dx) * aseg.ld) << 1;
    const long qstep = ((long)8 * aseg.ld) << 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool ok = gok[q] && (unsigned)(gy[q] + dy) < (unsigned)g.a.Ho && (unsigned)(gx[q] + dx) < (unsigned)g.a.Wo;
      ap[q] = (ok ? acen + q * qstep + toff : zero) + ((q & 1) ? acb1 : acb0);
    }
  };
  auto a_ptrs = [&]() {
    const int t = lw + a_ord * G;
    const long m0 = (long)(t / ntn) * T3_BM;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long r = seg3_row(aseg, gok[q], gb[q], gy[q], gx[q], spatial, m0 + arow0 + 8 * q);
      ap[q] = (r >= 0 ? aseg.p + ((r * aseg.ld) << 1) : zero) + ((q & 1) ? acb1 : acb0);
    }
  };
  auto a_tile_geo = [&]() {
    const int t = lw + a_ord * G;
    const long m0 = (long)(t / ntn) * T3_BM;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long m = m0 + arow0 + 8 * q;
      gok[q] = m < g.M;
      gb[q] = 0; gy[q] = 0; gx[q] = 0;
      if (spatial && gok[q]) {
        const int b = (int)(m / hw), rem = (int)(m - (long)b * hw);
        gb[q] = b; gy[q] = rem / g.a.Wo; gx[q] = rem - gy[q] * g.a.Wo;
      }
    }
    if (taps_fast) {
      acen = aseg.p + (((m0 + arow0) * (long)aseg.ld) << 1);
      a_ptrs_fast(g.a.s[0].dy, g.a.s[0].dx);
    } else a_ptrs();
  };
  a_tile_geo();
  auto issue_a = [&](int slot) {
    const uint32_t dst = T3_AST * slot + wid * 4096;
    if (a_ord < nt_my) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        __builtin_amdgcn_global_load_lds((glb_void*)ap[q], (lds_void*)(dsm + dst + q * 1024), 16, 0, 0);
        ap[q] += 2 * T3_BK;
      }
      a_off += T3_BK; ++a_kt;
      if (a_kt == nk) {
        ++a_ord; a_kt = 0; a_seg = 0; a_off = 0;
        if (a_ord < nt_my) { aseg = load_seg3_uniform(segtab, 0); a_tile_geo(); }
      } else if (a_off >= aseg.klen) {
        a_off = 0; ++a_seg;
        if (taps_fast) a_ptrs_fast(g.a.s[a_seg].dy, g.a.s[a_seg].dx);        // (kernel-argument scalar loads)
        else { aseg = load_seg3_uniform(segtab, a_seg); a_ptrs(); }
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        __builtin_amdgcn_global_load_lds((glb_void*)zero, (lds_void*)(dsm + dst + q * 1024), 16, 0, 0);
    }
  };

  // ---- W issue state: 3 wave-instructions per wave and stage (rows 8 (3 wid + q) + (lane >> 3))
  int b_ord = 0, b_kt = 0;
  const unsigned char* wp[3];
  auto b_ptrs = [&]() {
    const int t = lw + b_ord * G;
    const int n0 = (t % ntn) * T3_BN;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int row = 8 * (3 * wid + q) + (lane >> 3);
      const int f = ((row >> 1) & 1) | (((row >> 3) & 3) << 1);
      wp[q] = (n0 + row < g.N ? (const unsigned char*)g.W + (((long)(n0 + row) * g.ldw) << 1) : zero) + (((lane & 7) ^ f) << 4);
    }
  };
  b_ptrs();
  auto issue_b = [&](int slot) {
    const uint32_t dst = T3_BOFF + T3_BST * slot + wid * 3072;
    if (b_ord < nt_my) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        __builtin_amdgcn_global_load_lds((glb_void*)wp[q], (lds_void*)(dsm + dst + q * 1024), 16, 0, 0);
        wp[q] += 2 * T3_BK;
      }
      if (++b_kt == nk) {
        b_kt = 0; ++b_ord;
        if (b_ord < nt_my) b_ptrs();
      }
    } else {
#pragma unroll
      for (int q = 0; q < 3; ++q)
        __builtin_amdgcn_global_load_lds((glb_void*)zero, (lds_void*)(dsm + dst + q * 1024), 16, 0, 0);
    }
  };

  // ---- fragment read addresses (lane part; + slot base, + immediates for the sub-tiles)
  const int fA = (fi >> 1) & 7;
  const int fW = ((fi >> 1) & 1) | (((fi >> 2) & 3) << 1);
  const uint32_t aRd0 = lbase + (wr * 64 + fi) * 128 + ((fg ^ fA) << 4);
  const uint32_t aRd1 = lbase + (wr * 64 + fi) * 128 + (((4 + fg) ^ fA) << 4);
  const int wr_row = wc * 96 + 8 * (fi >> 2) + (fi & 3);
  const uint32_t wRd0 = lbase + T3_BOFF + wr_row * 128 + ((fg ^ fW) << 4);
  const uint32_t wRd1 = lbase + T3_BOFF + wr_row * 128 + (((4 + fg) ^ fW) << 4);

  f32x4 acc[4][3][2];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int t = 0; t < 3; ++t) { acc[u][t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[u][t][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  // SODT_EPI_STATS (thin instantiations: the head's BatchNorm convolutions with N <= 64, common.py:38-50): per-lane column sums of v
  // and v^2 over every tile of this persistent workgroup in f32 (<= 2,048 rows per workgroup at one tile per CU round), reduced over
  // the 16 rows of a lane group at the end, ONE f64 atomic per column, statistic and wave into the replica buffer (gemm.hip's
  // weight-stationary kernel does the same per workgroup; per tile it cost 1.4-2.7x: round-1 notes)
  constexpr bool STATS = (CF & SODT_EPI_STATS) != 0;
  static_assert(!STATS || NV < 3, "column statistics: thin instantiations only");
  float st1[STATS ? NV : 1][8], st2[STATS ? NV : 1][8];
#pragma unroll
  for (int t = 0; t < (STATS ? NV : 1); ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) { st1[t][j] = 0.f; st2[t][j] = 0.f; }

  // ---- prologue: A(0), W(0), A(1)
  issue_a(0);
  issue_b(0);
  issue_a(1);
  int a_slot = 0, b_slot = 0, c_ord = 0, c_kt = 0;
  // epilogue operand (residual / GELU input) of the tile being finished: fetched with inline-asm loads at the START of
  // the tile's last K-step, ahead of that step's DMA issues, so it lands under the MFMAs and is waited for with a
  // counted vmcnt(7); a compiler-visible load in the epilogue would wait vmcnt(0) and drain the DMA queue every tile
  constexpr bool PRE = (CF & (SODT_EPI_RESID | SODT_EPI_DGELU | SODT_EPI_DRELU)) != 0;
  static_assert(!((CF & SODT_EPI_RESID) && (CF & (SODT_EPI_DGELU | SODT_EPI_DRELU))), "one prefetched epilogue operand");
  // SODT_EPI_DGELU_RC: at the middle of the K range the accumulators hold the recomputed pre-activation; gelu'(h) is
  // parked (bf16) in the same registers the prefetched operand would use and the accumulators restart for dh_act
  constexpr bool RC = (CF & SODT_EPI_DGELU_RC) != 0;
  // (thin instantiations: t < NV only - an inline-asm load whose result the compiler can prove unused gets its destination
  //  registers re-allocated while the load is still in flight)
  u32x4 pre[4][3];
  for (int s = 0; s < total; ++s) {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // everything but the newest A stage has landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (PRE && c_kt == nk - 1) {
      const int t_ = lw + c_ord * G;
      const long m0 = (long)(t_ / ntn) * T3_BM;
      const int n0 = (t_ % ntn) * T3_BN;
      const bf16* base = (CF & SODT_EPI_RESID) ? (const bf16*)g.R : (const bf16*)g.aux;
      const int ldp = (CF & SODT_EPI_RESID) ? g.ldr : g.ldaux;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        long m = m0 + wr * 64 + 16 * u + fi;
        if (m >= g.M) m = g.M - 1;
#pragma unroll
        for (int t = 0; t < NV; ++t) {
          int n = n0 + wc * 96 + 32 * t + 8 * fg;
          if (n >= g.N) n = 0;                                  // (partial column tile: any in-bounds address, the chunk is not stored)
          const bf16* ptr = base + m * ldp + n;
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pre[u][t]) : "v"(ptr) : "memory");
        }
      }
    }
    issue_b(b_slot ^ 1);
    issue_a(a_slot == 0 ? 2 : a_slot - 1);                // (s + 2) % 3
    const uint32_t ao = a_slot * T3_AST, bo = b_slot * T3_BST;
#define T3_KB(ARD, WRD)                                                              \
    {                                                                                \
      u32x4 fa0 = lds_rd128<0>(ARD + ao), fa1 = lds_rd128<2048>(ARD + ao);            \
      u32x4 fa2 = lds_rd128<4096>(ARD + ao), fa3 = lds_rd128<6144>(ARD + ao);         \
      u32x4 fw0 = lds_rd128<0>(WRD + bo), fw1 = lds_rd128<512>(WRD + bo);             \
      u32x4 fw2 = lds_rd128<4096>(WRD + bo), fw3 = lds_rd128<4608>(WRD + bo);         \
      u32x4 fw4 = lds_rd128<8192>(WRD + bo), fw5 = lds_rd128<8704>(WRD + bo);         \
      T3_LGKM0();                                                                    \
      T3_PRIO_HI();                                                                  \
      mma_sw(acc[0][0][0], fw0, fa0); mma_sw(acc[0][0][1], fw1, fa0);                \
      mma_sw(acc[0][1][0], fw2, fa0); mma_sw(acc[0][1][1], fw3, fa0);                \
      mma_sw(acc[0][2][0], fw4, fa0); mma_sw(acc[0][2][1], fw5, fa0);                \
      mma_sw(acc[1][0][0], fw0, fa1); mma_sw(acc[1][0][1], fw1, fa1);                \
      mma_sw(acc[1][1][0], fw2, fa1); mma_sw(acc[1][1][1], fw3, fa1);                \
      mma_sw(acc[1][2][0], fw4, fa1); mma_sw(acc[1][2][1], fw5, fa1);                \
      mma_sw(acc[2][0][0], fw0, fa2); mma_sw(acc[2][0][1], fw1, fa2);                \
      mma_sw(acc[2][1][0], fw2, fa2); mma_sw(acc[2][1][1], fw3, fa2);                \
      mma_sw(acc[2][2][0], fw4, fa2); mma_sw(acc[2][2][1], fw5, fa2);                \
      mma_sw(acc[3][0][0], fw0, fa3); mma_sw(acc[3][0][1], fw1, fa3);                \
      mma_sw(acc[3][1][0], fw2, fa3); mma_sw(acc[3][1][1], fw3, fa3);                \
      mma_sw(acc[3][2][0], fw4, fa3); mma_sw(acc[3][2][1], fw5, fa3);                \
      T3_PRIO_LO();                                                                  \
    }
#define T3_KBP(ARD, WRD)                                                             \
    {                                                                                \
      u32x4 fa0 = lds_rd128<0>(ARD + ao), fa1 = lds_rd128<2048>(ARD + ao);            \
      u32x4 fa2 = lds_rd128<4096>(ARD + ao), fa3 = lds_rd128<6144>(ARD + ao);         \
      u32x4 fw0 = lds_rd128<0>(WRD + bo), fw1 = lds_rd128<512>(WRD + bo);             \
      u32x4 fw2 = fw0, fw3 = fw1;                                                    \
      if constexpr (NV > 1) { fw2 = lds_rd128<4096>(WRD + bo); fw3 = lds_rd128<4608>(WRD + bo); } \
      T3_LGKM0();                                                                    \
      mma_sw(acc[0][0][0], fw0, fa0); mma_sw(acc[0][0][1], fw1, fa0);                \
      mma_sw(acc[1][0][0], fw0, fa1); mma_sw(acc[1][0][1], fw1, fa1);                \
      mma_sw(acc[2][0][0], fw0, fa2); mma_sw(acc[2][0][1], fw1, fa2);                \
      mma_sw(acc[3][0][0], fw0, fa3); mma_sw(acc[3][0][1], fw1, fa3);                \
      if constexpr (NV > 1) {                                                        \
        mma_sw(acc[0][1][0], fw2, fa0); mma_sw(acc[0][1][1], fw3, fa0);              \
        mma_sw(acc[1][1][0], fw2, fa1); mma_sw(acc[1][1][1], fw3, fa1);              \
        mma_sw(acc[2][1][0], fw2, fa2); mma_sw(acc[2][1][1], fw3, fa2);              \
        mma_sw(acc[3][1][0], fw2, fa3); mma_sw(acc[3][1][1], fw3, fa3);              \
      }                                                                              \
    }
    if constexpr (NV == 3) {
      T3_KB(aRd0, wRd0)
      if (PRE || RC) __builtin_amdgcn_sched_barrier(0);   // keep the second half's fragment reads behind the first half's MFMAs:
      T3_KB(aRd1, wRd1)                             //  frees ~40 VGPRs for the prefetched epilogue operand (no spills)
    } else if (wc == 0) {
      T3_KBP(aRd0, wRd0)
      __builtin_amdgcn_sched_barrier(0);
      T3_KBP(aRd1, wRd1)
    }
#undef T3_KB
#undef T3_KBP
    a_slot = a_slot == 2 ? 0 : a_slot + 1;
    b_slot ^= 1;
    if (RC && c_kt + 1 == (nk >> 1)) {
      const int t_ = lw + c_ord * G;
      const int n0 = (t_ % ntn) * T3_BN;
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const int n = n0 + wc * 96 + 32 * t + 8 * fg;
          const u32x4 b0 = lds_rd128<0>(lbase + T3_BIASOFF + 4 * n), b1 = lds_rd128<16>(lbase + T3_BIASOFF + 4 * n);
          T3_LGKM0();
          float h[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) { h[r] = acc[u][t][0][r]; h[4 + r] = acc[u][t][1][r]; }
          h[0] += __uint_as_float(b0.x); h[1] += __uint_as_float(b0.y); h[2] += __uint_as_float(b0.z); h[3] += __uint_as_float(b0.w);
          h[4] += __uint_as_float(b1.x); h[5] += __uint_as_float(b1.y); h[6] += __uint_as_float(b1.z); h[7] += __uint_as_float(b1.w);
#pragma unroll
          for (int j = 0; j < 8; ++j) h[j] = dgelu_t<bf16>(h[j]);
          const uint4 pk = pack<bf16>(h);
          pre[u][t].x = pk.x; pre[u][t].y = pk.y; pre[u][t].z = pk.z; pre[u][t].w = pk.w;
          acc[u][t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[u][t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    if (++c_kt == nk) {
      // ---- epilogue of this tile straight from the accumulators: lane (fg, fi) holds, per (u, t), the 8 columns
      //      n0 + wc*96 + 32 t + 8 fg .. + 7 of row m0 + wr*64 + 16 u + fi
      const int t_ = lw + c_ord * G;
      const long m0 = (long)(t_ / ntn) * T3_BM;
      const int n0 = (t_ % ntn) * T3_BN;
      if (PRE) { asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long m = m0 + wr * 64 + 16 * u + fi;
#pragma unroll
        for (int t = 0; t < NV; ++t) {
          float v[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) { v[r] = acc[u][t][0][r]; v[4 + r] = acc[u][t][1][r]; }
          const int n = n0 + wc * 96 + 32 * t + 8 * fg;
          if (PRE || RC) {
            float x[8];
            uint4 pu; pu.x = pre[u][t].x; pu.y = pre[u][t].y; pu.z = pre[u][t].z; pu.w = pre[u][t].w;
            unpack<bf16>(pu, x);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              if (RC) v[j] *= x[j];                               // parked gelu'(h)
              else if (CF & SODT_EPI_DGELU) v[j] *= dgelu_t<bf16>(x[j]);     // same order as epi_chunk: (bias, dgelu | drelu, resid)
              else if (CF & SODT_EPI_DRELU) v[j] = x[j] > 0.f ? v[j] : 0.f;
              else v[j] += x[j];
            }
          }
          if ((CF & SODT_EPI_BIAS) && !RC) {
            const u32x4 b0 = lds_rd128<0>(lbase + T3_BIASOFF + 4 * n), b1 = lds_rd128<16>(lbase + T3_BIASOFF + 4 * n);
            T3_LGKM0();
            v[0] += __uint_as_float(b0.x); v[1] += __uint_as_float(b0.y); v[2] += __uint_as_float(b0.z); v[3] += __uint_as_float(b0.w);
            v[4] += __uint_as_float(b1.x); v[5] += __uint_as_float(b1.y); v[6] += __uint_as_float(b1.z); v[7] += __uint_as_float(b1.w);
          }
          constexpr int CF2 = CF & ~(SODT_EPI_BIAS | SODT_EPI_RESID | SODT_EPI_DGELU | SODT_EPI_DGELU_RC | SODT_EPI_DRELU | SODT_EPI_STATS);
          if constexpr (STATS) {
            if (m < g.M && n < g.N) {
#pragma unroll
              for (int j = 0; j < 8; ++j) { st1[t][j] += v[j]; st2[t][j] = fmaf(v[j], v[j], st2[t][j]); }
            }
          }
          if (m < g.M && n < g.N) {
            if constexpr (OSC) epi_chunk<bf16, -1>(g, CF2, m, n, v, hw);      // (generic path: knows the row scatter)
            else epi_chunk<bf16, CF2>(g, CF2, m, n, v, hw);
          }
          acc[u][t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[u][t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      c_kt = 0; ++c_ord;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the dummy tail DMAs must land before the LDS is released
  if constexpr (STATS) {
    if (wc == 0) {
#pragma unroll
      for (int t = 0; t < NV; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float a = group16_sum(st1[t][j]), b = group16_sum(st2[t][j]);
          const int n = 32 * t + 8 * fg + j;                 // (one column tile: n0 = 0)
          if (fi == 0 && n < g.N) {
            double* st = g.stats + (size_t)((blockIdx.x * 4 + wr) % SODT_STATS_REPL) * 2 * g.N;
            atomicAdd(st + n, (double)a);
            atomicAdd(st + g.N + n, (double)b);
          }
        }
    }
  }
}

template <int CF, bool OSC = false, int NV = 3>
int launch_nt3(const sodt_gemm_args* g, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gemm_nt3_kernel<CF, OSC, NV>, hipFuncAttributeMaxDynamicSharedMemorySize, T3_LDS) != hipSuccess) {
      (void)hipGetLastError();
      return SODT_EINVAL;
    }
    attr_set = true;
  }
  const long ntiles = (long)((g->M + T3_BM - 1) / T3_BM) * ((g->N + T3_BN - 1) / T3_BN);
  const int grid = (int)(ntiles < 256 ? ntiles : 256);
  hipLaunchKernelGGL((gemm_nt3_kernel<CF, OSC, NV>), dim3(grid), dim3(512), T3_LDS, st, *g);
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}


// ---------------------------------------------------------------------------------
// Pipelined bf16 TN GEMM (weight gradients):  dW[N][K] += dY[M][N]^T @ X[M][K]  (+ dbias = column sums of dY).
// Same machinery as the NT kernel above: LDS-DMA ring (5 stages of 32 rows, 4 in flight), one raw barrier per
// stage, counted vmcnt, inline-asm fragment reads.  Both operands are contracted along their LDS ROWS, so the
// fragments come from ds_read_b64_tr_b16 on row-major [m][col] images, XOR-swizzled on the 32-byte unit.
// One workgroup owns a 256 x 192 tile of dW for one M-slice; the 256-wide side is whichever of N / K pads less
// (SWAP: K on the 256 side), so the C = 192-multiple layer widths tile with little or no waste.  dbias is an
// extra MFMA against a vector of ones (no LDS column walk).  Partial tiles are combined with f32 atomics.
// ---------------------------------------------------------------------------------
constexpr int N5_ROWS = 32, N5_NST = 5;
constexpr int N5_PST = N5_ROWS * 512;               // P image: 256 bf16 per row
constexpr int N5_QST = N5_ROWS * 384;               // Q image: 192 bf16 per row
constexpr int N5_STAGE = N5_PST + N5_QST;           // 28 KiB
constexpr int N5_SEGOFF = N5_NST * N5_STAGE;
constexpr int N5_LDS = N5_SEGOFF + 48 * SODT_MAX_SEG;

template <int OFF> __device__ __forceinline__ uint2 lds_rd_tr(uint32_t addr) {
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
  u32x2 t;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(t) : "v"(addr), "n"(OFF));
  uint2 v; v.x = t.x; v.y = t.y;
  return v;
}
template <int HI> __device__ __forceinline__ u32x4 tn3_frag(uint32_t addr) {
  const uint2 lo = lds_rd_tr<0>(addr), hi = lds_rd_tr<HI>(addr);
  u32x4 r; r.x = lo.x; r.y = lo.y; r.z = hi.x; r.w = hi.y;
  return r;
}

struct XDesc { Seg3 sg; int off; int b, y, x; bool colok; const unsigned char* cur; };   // one X-side DMA instruction

template <bool SWAP, bool SPATIAL>
__global__ __launch_bounds__(512) void gemm_tn3_kernel(const sodt_gemm_tn_args g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int fr = lane & 15, fg = lane >> 4;
  const uint32_t lbase = lds_addr(dsm);
#if SODT_EXP_PRIO == 2
  if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);
#endif
  if (tid == 0) {
#pragma unroll
    for (int j = 0; j < SODT_MAX_SEG; ++j) {
      unsigned char* e = dsm + N5_SEGOFF + 48 * j;
      *(uint64_t*)(e) = (uint64_t)(uintptr_t)g.x.s[j].p;
      *(int4*)(e + 16) = make_int4(g.x.s[j].ld, g.x.s[j].klen, g.x.s[j].dy, g.x.s[j].dx);
      *(int4*)(e + 32) = make_int4(g.x.s[j].mul, g.x.s[j].shr, g.x.s[j].Hi, g.x.s[j].Wi);
    }
  }
  __syncthreads();
  const uint32_t segtab = lbase + N5_SEGOFF;

  const int Pdim = SWAP ? g.K : g.N, Qdim = SWAP ? g.N : g.K;
  const int ntp = (Pdim + 255) / 256, ntq = (Qdim + 191) / 192;
  const int ntiles = ntp * ntq;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int split = lid / ntiles, tile = lid - split * ntiles;
  const int tq = tile % ntq, tp = tile / ntq;
  const int P0 = tp * 256, Q0 = tq * 192;
  const long rows_per = (((g.M + g.splits - 1) / g.splits) + N5_ROWS - 1) / N5_ROWS * N5_ROWS;
  const long mbeg = (long)split * rows_per;
  const long mend = (mbeg + rows_per < g.M) ? (mbeg + rows_per) : g.M;
  if (mbeg >= mend) return;
  const int nrows = (int)(mend - mbeg);
  const int nsteps = (nrows + N5_ROWS - 1) / N5_ROWS;
  const unsigned char* zero = (const unsigned char*)&g_zero16;
  const int hw = g.x.Ho * g.x.Wo;

  // ---- DMA descriptors.  P image: instruction j = 2 wid + e covers rows 2j, 2j+1 (32 chunks each);
  //      Q image: linear chunk id = 64 j + lane -> row id / 24, chunk id % 24; waves 0-3 issue two, waves 4-7 one.
  int prow[2], pcol[2], qrow[2], qcol[2];
  const int nq = wid < 4 ? 2 : 1;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int row = 2 * (2 * wid + e) + (lane >> 5), c = lane & 31;
    const int sw = (row & 3) | (((row >> 3) & 1) << 2);
    prow[e] = row; pcol[e] = P0 + (((((c >> 1) ^ sw) << 1) | (c & 1)) << 3);
    const int j = wid < 4 ? 2 * wid + e : 8 + (wid - 4);
    const int id = 64 * j + lane;
    const int r2 = id / 24, c2 = id - r2 * 24;
    const int sw2 = ((r2 >> 1) & 1) | (((r2 >> 3) & 1) << 1);
    qrow[e] = r2; qcol[e] = Q0 + (((((c2 >> 1) ^ sw2) << 1) | (c2 & 1)) << 3);
  }
  // dY-side instructions: a running pointer per instruction, advanced by 32 rows per stage.  Everything is kept in
  // named scalars: arrays indexed through the lambdas end up in scratch, whose reloads drain the DMA queue.
  const int ycol0 = SWAP ? qcol[0] : pcol[0], ycol1 = SWAP ? qcol[1] : pcol[1];
  const int yrow0 = SWAP ? qrow[0] : prow[0], yrow1 = SWAP ? qrow[1] : prow[1];
  const bool yok0 = ycol0 < g.N, yok1 = ycol1 < g.N;
  const unsigned char* ycur0 = (const unsigned char*)g.dY + (((mbeg + yrow0) * g.ldy + (yok0 ? ycol0 : 0)) << 1);
  const unsigned char* ycur1 = (const unsigned char*)g.dY + (((mbeg + yrow1) * g.ldy + (yok1 ? ycol1 : 0)) << 1);
  const long ystep = (long)N5_ROWS * g.ldy * 2;
  // X-side instructions: segment + offset of the fixed column; running pointer (plain) or row geometry (spatial)
  const int xrow0 = SWAP ? prow[0] : qrow[0], xrow1 = SWAP ? prow[1] : qrow[1];
  auto x_setup = [&](XDesc& d, int col, int xrow) {
    d.colok = col < g.K;
    int si = 0;
    if (d.colok) {
      for (int j = 0; j + 1 < g.x.nseg; ++j) {
        const Seg3 t = load_seg3(segtab, si);
        if (col >= t.klen) { col -= t.klen; ++si; }
      }
    }
    d.sg = load_seg3(segtab, d.colok ? si : 0);
    d.off = d.colok ? col : 0;
    const long m = mbeg + xrow;
    d.b = 0; d.y = 0; d.x = 0;
    if (SPATIAL) {
      const int b = (int)(m / hw), rem = (int)(m - (long)b * hw);
      d.b = b; d.y = rem / g.x.Wo; d.x = rem - d.y * g.x.Wo;
    }
    d.cur = d.sg.p + ((m * d.sg.ld + d.off) << 1);
  };
  XDesc xd0, xd1;
  x_setup(xd0, SWAP ? pcol[0] : qcol[0], xrow0);
  x_setup(xd1, SWAP ? pcol[1] : qcol[1], xrow1);
  const long xstep0 = (long)N5_ROWS * xd0.sg.ld * 2, xstep1 = (long)N5_ROWS * xd1.sg.ld * 2;
  int rel = 0;         // row base of the next stage to issue, relative to mbeg
  auto x_src = [&](const XDesc& d, int xrow) -> const unsigned char* {
    const unsigned char* r = zero;
    if (d.colok && rel + xrow < nrows) {
      if (SPATIAL) {
        const Seg3& sg = d.sg;
        const int yy = d.y * sg.mul + sg.dy, xx = d.x * sg.mul + sg.dx;
        const int yi = yy >> sg.shr, xi = xx >> sg.shr;
        if (yy >= 0 && xx >= 0 && yi < sg.Hi && xi < sg.Wi) {
          const int sr = (d.b * sg.Hi + yi) * sg.Wi + xi;          // < 2^31 source rows
          r = sg.p + (((long)sr * sg.ld + d.off) << 1);
        }
      } else {
        r = d.cur;
      }
    }
    return r;
  };
  auto x_adv = [&](XDesc& d, long step) {
    if (SPATIAL) {          // advance the row by 32 tokens
      d.x += N5_ROWS;
      while (d.x >= g.x.Wo) { d.x -= g.x.Wo; ++d.y; }
      if (d.y >= g.x.Ho) { d.y -= g.x.Ho; ++d.b; }
    } else {
      d.cur += step;
    }
  };
  auto issue = [&](int slot) {
    const uint32_t pdst = slot * N5_STAGE + wid * 2048;
    const uint32_t qdst = slot * N5_STAGE + N5_PST + (wid < 4 ? wid * 2048 : 8192 + (wid - 4) * 1024);
    const unsigned char* sy0 = (yok0 && rel + yrow0 < nrows) ? ycur0 : zero;
    const unsigned char* sy1 = (yok1 && rel + yrow1 < nrows) ? ycur1 : zero;
    const unsigned char* sx0 = x_src(xd0, xrow0);
    const unsigned char* sx1 = x_src(xd1, xrow1);
    __builtin_amdgcn_global_load_lds((glb_void*)(SWAP ? sx0 : sy0), (lds_void*)(dsm + pdst), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_void*)(SWAP ? sx1 : sy1), (lds_void*)(dsm + pdst + 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_void*)(SWAP ? sy0 : sx0), (lds_void*)(dsm + qdst), 16, 0, 0);
    if (nq == 2) __builtin_amdgcn_global_load_lds((glb_void*)(SWAP ? sy1 : sx1), (lds_void*)(dsm + qdst + 1024), 16, 0, 0);
    rel += N5_ROWS;
    ycur0 += ystep; ycur1 += ystep;
    x_adv(xd0, xstep0); x_adv(xd1, xstep1);
  };

  // ---- fragment read addresses: lane (fg, q = fr >> 2, p = fr & 3) reads 8 bytes of LDS row 8 fg + q (+4)
  const int q_ = fr >> 2, p_ = fr & 3;
  const int frow = 8 * fg + q_;
  const int swP = q_ | ((fg & 1) << 2);
  const int swQ = (q_ >> 1) | ((fg & 1) << 1);
  uint32_t pa[4], qa[6];
#pragma unroll
  for (int i = 0; i < 4; ++i) pa[i] = lbase + frow * 512 + (((wr * 4 + i) ^ swP) << 5) + 8 * p_;
#pragma unroll
  for (int j = 0; j < 6; ++j) qa[j] = lbase + N5_PST + frow * 384 + (((wc * 6 + j) ^ swQ) << 5) + 8 * p_;

  const bool wave_live = (P0 + wr * 64 < Pdim) && (Q0 + wc * 96 < Qdim);
  // dbias: sums over m of the dY operand; done once per dY column by the tiles / waves on the first X tile
  const bool do_bias = g.dbias != nullptr && (SWAP ? (tp == 0 && wr == 0) : (tq == 0 && wc == 0)) && wave_live;
  u32x4 ones; ones.x = ones.y = ones.z = ones.w = 0x3F803F80u;

  f32x4 acc[4][6], bacc[6];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 6; ++j) bacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue(0); issue(1); issue(2); issue(3);
  int slot = 0;
  for (int s = 0; s < nsteps; ++s) {
    if (wid < 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(slot == 0 ? N5_NST - 1 : slot - 1);          // stage s + 4 -> slot (s + 4) % 5
    if (wave_live) {
      const uint32_t so = slot * N5_STAGE;
      u32x4 fp[4], fq[6];
#pragma unroll
      for (int i = 0; i < 4; ++i) fp[i] = tn3_frag<2048>(pa[i] + so);
#pragma unroll
      for (int j = 0; j < 6; ++j) fq[j] = tn3_frag<1536>(qa[j] + so);
      T3_LGKM0();
      T3_PRIO_HI();
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) mma_sw(acc[i][j], fp[i], fq[j]);
      T3_PRIO_LO();
      if (do_bias) {
        if (SWAP) {
#pragma unroll
          for (int j = 0; j < 6; ++j) mma_sw(bacc[j], ones, fq[j]);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) mma_sw(bacc[i], fp[i], ones);
        }
      }
    }
    slot = slot == N5_NST - 1 ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();        // every wave is done with the stage images (the SWAP epilogue reuses them)
  if (!wave_live) return;

  // acc[i][j][r]: P column P0 + wr*64 + 16 i + 4 fg + r, Q column Q0 + wc*96 + 16 j + fr
  // with a scratch: this slice's partial tile goes to part[n][k] (natural k order) with plain stores
  float* part = g.partial ? g.partial + (long)split * g.N * g.K : nullptr;
  if (!SWAP) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int qc = Q0 + wc * 96 + 16 * j + fr;
        if (qc >= Qdim) continue;
        int k = qc;
        if (g.kperm_t > 1) k = (k % g.kperm_c) * g.kperm_t + k / g.kperm_c;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = P0 + wr * 64 + 16 * i + 4 * fg + r;
          if (n < Pdim) {
            if (part) part[(long)n * g.K + qc] = acc[i][j][r];
            else atomicAdd(g.dW + (long)n * g.lddw + k, acc[i][j][r]);
          }
        }
      }
  } else {
    // K runs along the accumulator ROWS here: transpose each 16(k) x 96(n) strip through this wave's private LDS
    // patch so that the 16 lanes of a group add to consecutive k of one dW row (coalesced atomics)
    float* patch = (float*)(dsm + wid * (96 * 17 * 4));        // [n 96][k 16 (+1 pad)] f32 per wave, 6.4 KiB
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) patch[(16 * j + fr) * 17 + 4 * fg + r] = acc[i][j][r];
      __builtin_amdgcn_wave_barrier();
      const int pc = P0 + wr * 64 + 16 * i + fr;               // this lane's k
      int k = pc;
      if (g.kperm_t > 1) k = (k % g.kperm_c) * g.kperm_t + k / g.kperm_c;
      for (int nn = fg; nn < 96; nn += 4) {
        const int n = Q0 + wc * 96 + nn;
        const float v = patch[nn * 17 + fr];
        if (pc < Pdim && n < Qdim) {
          if (part) part[(long)n * g.K + pc] = v;
          else atomicAdd(g.dW + (long)n * g.lddw + k, v);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (do_bias) {
    if (SWAP) {          // bacc[j][*] rows all equal: column fr <-> n
      if (fg == 0) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int n = Q0 + wc * 96 + 16 * j + fr;
          if (n < g.N) atomicAdd(g.dbias + n, bacc[j][0]);
        }
      }
    } else {             // bacc[i][r]: row 4 fg + r <-> n, all columns equal
      if (fr == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int n = P0 + wr * 64 + 16 * i + 4 * fg + r;
            if (n < g.N) atomicAdd(g.dbias + n, bacc[i][r]);
          }
      }
    }
  }
}

// dW[n][perm(k)] += sum over slices of partial[s][n][k]: 64 float4 columns per workgroup, the slices dealt over its
// four waves (independent 16-byte loads, four in flight per thread) and combined through LDS
__global__ __launch_bounds__(256) void tn3_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dW, int N, int K,
                                                        int lddw, int splits, int kperm_c, int kperm_t) {
  __shared__ float4 red[4][64];
  const long nk4 = (long)N * K / 4;
  const long slice = (long)N * K;
  const int c = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + c;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < nk4) {
    const float* p = partial + i * 4;
    int s = sg;
    for (; s + 12 < splits; s += 16) {
      const float4 v0 = *(const float4*)(p + (long)s * slice), v1 = *(const float4*)(p + (long)(s + 4) * slice);
      const float4 v2 = *(const float4*)(p + (long)(s + 8) * slice), v3 = *(const float4*)(p + (long)(s + 12) * slice);
      a.x += (v0.x + v1.x) + (v2.x + v3.x); a.y += (v0.y + v1.y) + (v2.y + v3.y);
      a.z += (v0.z + v1.z) + (v2.z + v3.z); a.w += (v0.w + v1.w) + (v2.w + v3.w);
    }
    for (; s < splits; s += 4) {
      const float4 v = *(const float4*)(p + (long)s * slice);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  red[sg][c] = a;
  __syncthreads();
  if (sg == 0 && i < nk4) {
    const float4 b1 = red[1][c], b2 = red[2][c], b3 = red[3][c];
    a.x += b1.x + b2.x + b3.x; a.y += b1.y + b2.y + b3.y; a.z += b1.z + b2.z + b3.z; a.w += b1.w + b2.w + b3.w;
    const int n = (int)((i * 4) / K), k0 = (int)((i * 4) - (long)n * K);
    float* d = dW + (long)n * lddw;
    if (kperm_t > 1) {
      const float vv[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) { const int k = k0 + j; d[(k % kperm_c) * kperm_t + k / kperm_c] += vv[j]; }
    } else {
      float4 o = *(float4*)(d + k0);
      o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
      *(float4*)(d + k0) = o;
    }
  }
}

template <bool SWAP, bool SPATIAL>
int launch_tn3(const sodt_gemm_tn_args* g, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gemm_tn3_kernel<SWAP, SPATIAL>, hipFuncAttributeMaxDynamicSharedMemorySize, N5_LDS) != hipSuccess) {
      (void)hipGetLastError();
      return SODT_EINVAL;
    }
    attr_set = true;
  }
  const int Pdim = SWAP ? g->K : g->N, Qdim = SWAP ? g->N : g->K;
  const long tiles = (long)((Pdim + 255) / 256) * ((Qdim + 191) / 192);
  sodt_gemm_tn_args a = *g;
  const bool use_partial = a.partial && a.splits > 1 && (long)a.splits * a.N * a.K <= a.partial_floats &&
                           ((uintptr_t)a.partial & 15) == 0 && (a.lddw % 4) == 0 && ((uintptr_t)a.dW & 15) == 0;
  if (!use_partial) a.partial = nullptr;
  hipLaunchKernelGGL((gemm_tn3_kernel<SWAP, SPATIAL>), dim3((unsigned)(tiles * a.splits)), dim3(512), N5_LDS, st, a);
  if (use_partial) {
    // slices past the end of M launch no work and write no partial tile: the reduction reads only the live ones
    const long rows_per = ((((long)a.M + a.splits - 1) / a.splits) + N5_ROWS - 1) / N5_ROWS * N5_ROWS;
    const int live = (int)(((long)a.M + rows_per - 1) / rows_per);
    const long nk4 = (long)a.N * a.K / 4;
    const int blocks = (int)((nk4 + 63) / 64);
    hipLaunchKernelGGL(tn3_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)a.partial, a.dW, a.N, a.K, a.lddw,
                       live, a.kperm_c, a.kperm_t);
  }
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

}  // namespace

// eligibility of the pipelined kernel (bf16 only); the caller has validated pointers / alignment
bool sodt_nt3_eligible(const sodt_gemm_args* g) {
  switch (g->flags) {
    case 0: case SODT_EPI_BIAS: case SODT_EPI_RESID: case SODT_EPI_BIAS | SODT_EPI_RESID:
    case SODT_EPI_BIAS | SODT_EPI_GELU_DUAL: case SODT_EPI_DGELU: case SODT_EPI_BIAS | SODT_EPI_GELU: break;
    case SODT_EPI_RELU: case SODT_EPI_BIAS | SODT_EPI_RELU: case SODT_EPI_DRELU: break;      // the SR branch's convolutions (sr.py)
    case SODT_EPI_STATS:                           // the head's BatchNorm convolutions, thin outputs only (one column tile, NV = 2)
      if (g->N > 64 || !g->stats || g->oscatter) return false;
      break;
    case SODT_EPI_BIAS | SODT_EPI_DGELU_RC:
      if (g->K % (2 * T3_BK)) return false;        // both halves whole K-steps
      break;
    default: return false;
  }
  if (g->rmod > 0 || (g->oscatter && (g->flags != 0 || !g->a.spatial))) return false;
  // N: whole 192-column tiles, or (K >= 512) any multiple of 8 with the last tile partial - a narrow output (the 64-channel 3x3
  // convolutions of the SR branch, the 256-wide ones of its tail) is priced by the A stream, which this kernel moves by LDS-DMA
  // three stages ahead; the idle accumulator columns cost matrix cycles that are not the bound there
  const int kmin_partial = g->flags == SODT_EPI_STATS ? 192 : 512;     // (statistics: the alternative is the 128 x 128 K-loop kernel)
  if (g->N % 8 || (g->N % T3_BN && (g->K < kmin_partial || g->K > T3_MAXKLEN)) || g->K % T3_BK || g->K < 192 || g->M < T3_BM) return false;
  if ((g->flags & SODT_EPI_RESID) && (g->ldr % 8)) return false;
  if ((g->flags & (SODT_EPI_DGELU | SODT_EPI_DRELU)) && (g->ldaux % 8)) return false;
  if ((g->flags & SODT_EPI_BIAS) && g->N > T3_MAXBIAS) return false;
  if ((g->ldw % 8) || (g->ldc % 8)) return false;
  for (int i = 0; i < g->a.nseg; ++i)
    if (g->a.s[i].klen % T3_BK || g->a.s[i].klen > T3_MAXKLEN) return false;
  return true;
}

int sodt_nt3_launch(const sodt_gemm_args* g, hipStream_t st) {
  if (g->N <= 64 && !g->oscatter) {          // thin output (the 64-channel 3x3 convolutions of the SR branch and the head): NV = 2
    switch (g->flags) {
      case 0: return launch_nt3<0, false, 2>(g, st);
      case SODT_EPI_BIAS: return launch_nt3<SODT_EPI_BIAS, false, 2>(g, st);
      case SODT_EPI_BIAS | SODT_EPI_RESID: return launch_nt3<SODT_EPI_BIAS | SODT_EPI_RESID, false, 2>(g, st);
      case SODT_EPI_RESID: return launch_nt3<SODT_EPI_RESID, false, 2>(g, st);
      case SODT_EPI_BIAS | SODT_EPI_RELU: return launch_nt3<SODT_EPI_BIAS | SODT_EPI_RELU, false, 2>(g, st);
      case SODT_EPI_DRELU: return launch_nt3<SODT_EPI_DRELU, false, 2>(g, st);
      case SODT_EPI_STATS: return launch_nt3<SODT_EPI_STATS, false, 2>(g, st);
      default: break;
    }
  }
  switch (g->flags) {
    case 0: return g->oscatter ? launch_nt3<0, true>(g, st) : launch_nt3<0>(g, st);
    case SODT_EPI_BIAS: return launch_nt3<SODT_EPI_BIAS>(g, st);
    case SODT_EPI_RESID: return launch_nt3<SODT_EPI_RESID>(g, st);
    case SODT_EPI_BIAS | SODT_EPI_RESID: return launch_nt3<SODT_EPI_BIAS | SODT_EPI_RESID>(g, st);
    case SODT_EPI_BIAS | SODT_EPI_GELU_DUAL: return launch_nt3<SODT_EPI_BIAS | SODT_EPI_GELU_DUAL>(g, st);
    case SODT_EPI_DGELU: return launch_nt3<SODT_EPI_DGELU>(g, st);
    case SODT_EPI_BIAS | SODT_EPI_GELU: return launch_nt3<SODT_EPI_BIAS | SODT_EPI_GELU>(g, st);
    case SODT_EPI_RELU: return launch_nt3<SODT_EPI_RELU>(g, st);
    case SODT_EPI_BIAS | SODT_EPI_RELU: return launch_nt3<SODT_EPI_BIAS | SODT_EPI_RELU>(g, st);
    case SODT_EPI_DRELU: return launch_nt3<SODT_EPI_DRELU>(g, st);
    case SODT_EPI_BIAS | SODT_EPI_DGELU_RC: return launch_nt3<SODT_EPI_BIAS | SODT_EPI_DGELU_RC>(g, st);
    default: return SODT_EINVAL;
  }
}

// pipelined TN: K on the 256-wide side when that pads less (ties keep N there); mirrored by ops.tn_splits
bool sodt_tn3_swap(int N, int K) {
  const long a = (long)((N + 255) / 256) * 256 * ((K + 191) / 192) * 192;
  const long b = (long)((K + 255) / 256) * 256 * ((N + 191) / 192) * 192;
  return b < a;
}

int sodt_tn3_launch(const sodt_gemm_tn_args* g, hipStream_t st) {
  const bool sw = sodt_tn3_swap(g->N, g->K);
  if (g->x.spatial) return sw ? launch_tn3<true, true>(g, st) : launch_tn3<false, true>(g, st);
  return sw ? launch_tn3<true, false>(g, st) : launch_tn3<false, false>(g, st);
}
