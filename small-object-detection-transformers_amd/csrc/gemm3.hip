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

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int T3_BM = 256, T3_BN = 192, T3_BK = 64;
constexpr int T3_AST = T3_BM * 128;                 // 32 KiB per A stage
constexpr int T3_BST = T3_BN * 128;                 // 24 KiB per W stage
constexpr int T3_BOFF = 3 * T3_AST;                 // W ring after the A ring
constexpr int T3_SEGOFF = T3_BOFF + 2 * T3_BST;     // segment table (48 B per entry)
constexpr int T3_LDS = T3_SEGOFF + 48 * SODT_MAX_SEG;

__device__ uint4 g_zero16;                          // DMA source of rows outside the image / beyond M

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

__device__ __forceinline__ long seg3_row(const Seg3& s, bool ok, int b, int y, int x, int spatial, long m) {
  if (!ok) return -1;
  if (!spatial) return m;
  const int yy = y * s.mul + s.dy, xx = x * s.mul + s.dx;
  if (yy < 0 || xx < 0) return -1;
  const int yi = yy >> s.shr, xi = xx >> s.shr;
  if (yi >= s.Hi || xi >= s.Wi) return -1;
  return ((long)b * s.Hi + yi) * s.Wi + xi;
}

template <int CF>
__global__ __launch_bounds__(512) void gemm_nt3_kernel(const sodt_gemm_args g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int fi = lane & 15, fg = lane >> 4;
  const uint32_t lbase = lds_addr(dsm);

  if (tid == 0) {                  // own padded copy of the segment table (16-byte aligned fields)
#pragma unroll
    for (int j = 0; j < SODT_MAX_SEG; ++j) {
      unsigned char* e = dsm + T3_SEGOFF + 48 * j;
      *(uint64_t*)(e) = (uint64_t)(uintptr_t)g.a.s[j].p;
      *(int4*)(e + 16) = make_int4(g.a.s[j].ld, g.a.s[j].klen, g.a.s[j].dy, g.a.s[j].dx);
      *(int4*)(e + 32) = make_int4(g.a.s[j].mul, g.a.s[j].shr, g.a.s[j].Hi, g.a.s[j].Wi);
    }
  }
  __syncthreads();
  const uint32_t segtab = lbase + T3_SEGOFF;

  const int ntn = g.N / T3_BN;
  const int ntm = (g.M + T3_BM - 1) / T3_BM;
  const int ntiles = ntm * ntn;
  const int G = gridDim.x;
  const int lw = xcd_remap(blockIdx.x, G);
  const int nt_my = lw < ntiles ? (ntiles - lw + G - 1) / G : 0;
  const int nk = g.K / T3_BK;
  const int total = nt_my * nk;
  if (total == 0) return;
  const int hw = g.a.Ho * g.a.Wo;
  const int spatial = g.a.spatial;
  const unsigned char* zero = (const unsigned char*)&g_zero16;

  // ---- A issue state: 4 rows per thread and stage (wave-instruction q covers tile rows 32 wid + 8 q .. + 7)
  const int arow0 = 32 * wid + (lane >> 3);                    // + 8 q
  const int acb0 = ((lane & 7) ^ ((lane >> 4) & 3)) << 4;      // swizzled source chunk (bytes), q even
  const int acb1 = ((lane & 7) ^ (4 + ((lane >> 4) & 3))) << 4;   // q odd
  int a_ord = 0, a_kt = 0, a_seg = 0, a_off = 0;
  Seg3 aseg = load_seg3(segtab, 0);
  int gb[4], gy[4], gx[4]; bool gok[4]; long srow[4];
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
      srow[q] = seg3_row(aseg, gok[q], gb[q], gy[q], gx[q], spatial, m);
    }
  };
  auto a_rows = [&]() {
    const int t = lw + a_ord * G;
    const long m0 = (long)(t / ntn) * T3_BM;
#pragma unroll
    for (int q = 0; q < 4; ++q) srow[q] = seg3_row(aseg, gok[q], gb[q], gy[q], gx[q], spatial, m0 + arow0 + 8 * q);
  };
  a_tile_geo();
  auto issue_a = [&](int slot) {
    const uint32_t dst = T3_AST * slot + wid * 4096;
    if (a_ord < nt_my) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned char* src = zero;
        if (srow[q] >= 0) src = aseg.p + ((srow[q] * aseg.ld + a_off) << 1) + ((q & 1) ? acb1 : acb0);
        __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(dsm + dst + q * 1024), 16, 0, 0);
      }
      a_off += T3_BK; ++a_kt;
      if (a_kt == nk) {
        ++a_ord; a_kt = 0; a_seg = 0; a_off = 0;
        if (a_ord < nt_my) { aseg = load_seg3(segtab, 0); a_tile_geo(); }
      } else if (a_off >= aseg.klen) {
        a_off = 0; ++a_seg;
        aseg = load_seg3(segtab, a_seg);
        a_rows();
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        __builtin_amdgcn_global_load_lds((glb_void*)zero, (lds_void*)(dsm + dst + q * 1024), 16, 0, 0);
    }
  };

  // ---- W issue state: 3 wave-instructions per wave and stage (rows 8 (3 wid + q) + (lane >> 3))
  int b_ord = 0, b_kt = 0;
  int wrow[3], wcb[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int row = 8 * (3 * wid + q) + (lane >> 3);
    const int f = ((row >> 1) & 1) | (((row >> 3) & 3) << 1);
    wrow[q] = row; wcb[q] = ((lane & 7) ^ f) << 4;
  }
  auto issue_b = [&](int slot) {
    const uint32_t dst = T3_BOFF + T3_BST * slot + wid * 3072;
    if (b_ord < nt_my) {
      const int t = lw + b_ord * G;
      const int n0 = (t % ntn) * T3_BN;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const unsigned char* src = (const unsigned char*)g.W + (((long)(n0 + wrow[q]) * g.ldw + (long)b_kt * T3_BK) << 1) + wcb[q];
        __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(dsm + dst + q * 1024), 16, 0, 0);
      }
      if (++b_kt == nk) { b_kt = 0; ++b_ord; }
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

  // ---- prologue: A(0), W(0), A(1)
  issue_a(0);
  issue_b(0);
  issue_a(1);
  int a_slot = 0, b_slot = 0, c_ord = 0, c_kt = 0;
  for (int s = 0; s < total; ++s) {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // everything but the newest A stage has landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
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
    }
    T3_KB(aRd0, wRd0)
    T3_KB(aRd1, wRd1)
#undef T3_KB
    a_slot = a_slot == 2 ? 0 : a_slot + 1;
    b_slot ^= 1;
    if (++c_kt == nk) {
      // ---- epilogue of this tile straight from the accumulators: lane (fg, fi) holds, per (u, t), the 8 columns
      //      n0 + wc*96 + 32 t + 8 fg .. + 7 of row m0 + wr*64 + 16 u + fi
      const int t_ = lw + c_ord * G;
      const long m0 = (long)(t_ / ntn) * T3_BM;
      const int n0 = (t_ % ntn) * T3_BN;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long m = m0 + wr * 64 + 16 * u + fi;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          float v[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) { v[r] = acc[u][t][0][r]; v[4 + r] = acc[u][t][1][r]; }
          if (m < g.M) epi_chunk<bf16, CF>(g, CF, m, n0 + wc * 96 + 32 * t + 8 * fg, v, hw);
          acc[u][t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[u][t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      c_kt = 0; ++c_ord;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the dummy tail DMAs must land before the LDS is released
}

template <int CF>
int launch_nt3(const sodt_gemm_args* g, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gemm_nt3_kernel<CF>, hipFuncAttributeMaxDynamicSharedMemorySize, T3_LDS) != hipSuccess) {
      (void)hipGetLastError();
      return SODT_EINVAL;
    }
    attr_set = true;
  }
  const long ntiles = (long)((g->M + T3_BM - 1) / T3_BM) * (g->N / T3_BN);
  const int grid = (int)(ntiles < 256 ? ntiles : 256);
  hipLaunchKernelGGL((gemm_nt3_kernel<CF>), dim3(grid), dim3(512), T3_LDS, st, *g);
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

}  // namespace

// eligibility of the pipelined kernel (bf16 only); the caller has validated pointers / alignment
bool sodt_nt3_eligible(const sodt_gemm_args* g) {
  const int ok_flags = SODT_EPI_BIAS | SODT_EPI_RESID | SODT_EPI_GELU_DUAL | SODT_EPI_DGELU;
  if (g->flags & ~ok_flags) return false;
  switch (g->flags) {
    case 0: case SODT_EPI_BIAS: case SODT_EPI_RESID: case SODT_EPI_BIAS | SODT_EPI_RESID:
    case SODT_EPI_BIAS | SODT_EPI_GELU_DUAL: case SODT_EPI_DGELU: break;
    default: return false;
  }
  if (g->oscatter || g->rmod > 0) return false;
  if (g->N % T3_BN || g->K % T3_BK || g->K < 384 || g->M < T3_BM) return false;
  if ((g->ldw % 8) || (g->ldc % 8)) return false;
  for (int i = 0; i < g->a.nseg; ++i)
    if (g->a.s[i].klen % T3_BK) return false;
  return true;
}

int sodt_nt3_launch(const sodt_gemm_args* g, hipStream_t st) {
  switch (g->flags) {
    case 0: return launch_nt3<0>(g, st);
    case SODT_EPI_BIAS: return launch_nt3<SODT_EPI_BIAS>(g, st);
    case SODT_EPI_RESID: return launch_nt3<SODT_EPI_RESID>(g, st);
    case SODT_EPI_BIAS | SODT_EPI_RESID: return launch_nt3<SODT_EPI_BIAS | SODT_EPI_RESID>(g, st);
    case SODT_EPI_BIAS | SODT_EPI_GELU_DUAL: return launch_nt3<SODT_EPI_BIAS | SODT_EPI_GELU_DUAL>(g, st);
    case SODT_EPI_DGELU: return launch_nt3<SODT_EPI_DGELU>(g, st);
    default: return SODT_EINVAL;
  }
}
