// MFMA GEMMs of the hot path (gfx950).
//
//   sodt_gemm_nt : C[M][N] = epilogue( A[M][K] @ W[N][K]^T )       forward + input grads
//   sodt_gemm_tn : dW[N][K] += dY[M][N]^T @ X[M][K]  (+ dbias)      weight grads
//
// A / X are described as K-segments with a spatial row map (include/sodt_hip.h), so
// 1x1 / 2x2 / 3x3 convs, PatchMerging, upsample+concat and their transposes are all
// this one kernel: implicit GEMM, nothing is materialised.
//
// NT kernel: 128x128 output tile, 4 waves (2x2, 64x64 each = 4x4 MFMA 16x16 tiles),
// K-step of 128 bytes per row (64 bf16 / 32 f32), LDS double buffer with an
// XOR-swizzled 16-byte-chunk layout (chunk ^= row & 7), register-staged prefetch of
// the next K-step under the MFMAs of the current one, accumulators staged through LDS
// (padded rows) so that the epilogue reads/writes whole 16-byte row chunks.
// Roofline: MFMA for K >= 384; at K = 192 (stage 1) it is HBM-bound.
#include "gemm_epi.h"

bool sodt_nt3_eligible(const sodt_gemm_args* g);            // gemm3.hip: pipelined bf16 kernel for K >= 384
int sodt_nt3_launch(const sodt_gemm_args* g, hipStream_t st);
int sodt_tn3_launch(const sodt_gemm_tn_args* g, hipStream_t st);   // gemm3.hip: pipelined bf16 weight-gradient kernel

namespace {

int g_force_tiled = 0;   // test hook: sodt_gemm_set_variant(1) forces the K-loop / 128x128 TN kernels, (2) the A-stationary NT kernel
int g_variant = 0;

constexpr int BM = 128, BN = 128, ROWB = 128;          // ROWB: bytes per LDS row per K-step
constexpr int STAGE_BYTES = BM * ROWB;                 // 16 KiB per operand per stage
constexpr int EPI_LD = 132;                            // padded f32 row of the staged accumulator tile
constexpr int NT_LDS = BM * EPI_LD * 4;                // 67584 B >= 4 * STAGE_BYTES


// static-index select so the kernarg segment table stays in SGPRs
__device__ __forceinline__ sodt_seg pick_seg(const sodt_aspec& a, int i) {
  sodt_seg s = a.s[0];
#pragma unroll
  for (int j = 1; j < SODT_MAX_SEG; ++j)
    if (i == j) s = a.s[j];
  return s;
}


template <typename T, int CF>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const sodt_gemm_args g) {
  constexpr int KPL = TT<T>::KPL;
  constexpr int BK = 8 * KPL;                 // elements per K-step
  constexpr int SZ = TT<T>::SZ;
  __shared__ __attribute__((aligned(16))) unsigned char smem[NT_LDS];
  unsigned char* sA = smem;                   // [2][STAGE_BYTES]
  unsigned char* sB = smem + 2 * STAGE_BYTES; // [2][STAGE_BYTES]
  __shared__ sodt_seg sSeg[SODT_MAX_SEG];     // segment table (dynamic indexing of kernel arguments would go to scratch)

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) {
#pragma unroll
    for (int j = 0; j < SODT_MAX_SEG; ++j) sSeg[j] = g.a.s[j];
  }
  __syncthreads();
  const int wr = wid >> 1, wc = wid & 1;
  const int ntn = (g.N + BN - 1) / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = tile % ntn, tm = tile / ntn;
  const long m0 = (long)tm * BM;
  const int n0 = tn * BN;

  // ---- loader state: 4 rows x one 16-byte chunk per thread and operand
  const int lrow = tid >> 3, lch = tid & 7;
  RowGeo geo[4];
  const int hw = g.a.Ho * g.a.Wo;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long m = m0 + lrow + 32 * i;
    geo[i].ok = m < g.M;
    geo[i].b = 0; geo[i].y = 0; geo[i].x = 0;
    if (g.a.spatial && geo[i].ok) {
      const int b = (int)(m / hw), rem = (int)(m - (long)b * hw);
      geo[i].b = b; geo[i].y = rem / g.a.Wo; geo[i].x = rem - geo[i].y * g.a.Wo;
    }
  }
  int a_seg = 0, a_off = lch * KPL;
  sodt_seg seg = g.a.s[0];
  long srow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) srow[i] = seg_src_row(seg, geo[i], g.a.spatial, m0 + lrow + 32 * i);

  uint4 ra[4], rb[4];
  const int nk = (g.K + BK - 1) / BK;

  auto load_regs = [&](int kt) {
    while (a_seg < g.a.nseg && a_off >= seg.klen) {
      a_off -= seg.klen;
      ++a_seg;
      if (a_seg < g.a.nseg) {
        seg = sSeg[a_seg];
#pragma unroll
        for (int i = 0; i < 4; ++i) srow[i] = seg_src_row(seg, geo[i], g.a.spatial, m0 + lrow + 32 * i);
      }
    }
    const bool kin = a_seg < g.a.nseg;
    const T* ap = (const T*)seg.p;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = make_uint4(0, 0, 0, 0);
      if (kin && srow[i] >= 0) ra[i] = *(const uint4*)(ap + srow[i] * seg.ld + a_off);
    }
    a_off += BK;
    const int k = kt * BK + lch * KPL;
    const T* wp = (const T*)g.W;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = n0 + lrow + 32 * i;
      rb[i] = make_uint4(0, 0, 0, 0);
      if (n < g.N && k < g.K) rb[i] = *(const uint4*)(wp + (long)n * g.ldw + k);
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = lrow + 32 * i;
      const int off = r * ROWB + ((lch ^ (r & 7)) << 4);
      *(uint4*)(sA + buf * STAGE_BYTES + off) = ra[i];
      *(uint4*)(sB + buf * STAGE_BYTES + off) = rb[i];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  load_regs(0);
  store_lds(0);
  __syncthreads();
  const int fr = lane & 15, fg = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_regs(kt + 1);
    const unsigned char* a_s = sA + buf * STAGE_BYTES;
    const unsigned char* b_s = sB + buf * STAGE_BYTES;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      uint4 fa[4], fb[4];
      const int ch = kb * 4 + fg;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = wr * 64 + i * 16 + fr;
        fa[i] = *(const uint4*)(a_s + r * ROWB + ((ch ^ (r & 7)) << 4));
        const int c = wc * 64 + i * 16 + fr;
        fb[i] = *(const uint4*)(b_s + c * ROWB + ((ch ^ (c & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) mma16<T>(acc[i][j], fa[i], fb[j]);
    }
    if (kt + 1 < nk) store_lds(buf ^ 1);
    __syncthreads();
  }

  // ---- stage accumulators (f32) through LDS: rows padded to EPI_LD floats
  float* sC = (float*)smem;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        sC[(wr * 64 + i * 16 + fg * 4 + r) * EPI_LD + wc * 64 + j * 16 + fr] = acc[i][j][r];
  __syncthreads();

  const int flags = CF >= 0 ? CF : g.flags;
  if (flags & SODT_EPI_STATS) {
    if (tid < BN && n0 + tid < g.N) {
      double s = 0.0, s2 = 0.0;
      const int rmax = (int)((g.M - m0) < BM ? (g.M - m0) : BM);
      for (int r = 0; r < rmax; ++r) {
        const float v = sC[r * EPI_LD + tid];
        s += v; s2 += (double)v * v;
      }
      double* st = g.stats + (size_t)(blockIdx.x % SODT_STATS_REPL) * 2 * g.N;
      atomicAdd(st + n0 + tid, s);
      atomicAdd(st + g.N + n0 + tid, s2);
    }
  }

  if (flags & SODT_EPI_DETECT) {
    // Detect.forward's view(bs,na,no,ny,nx).permute(0,1,3,4,2): out[((b*na+a)*HW+p)*no+o] = z[b*HW+p][a*no+o] + bias
    float* out = (float*)g.C;
    for (int idx = tid; idx < BM * BN; idx += 256) {
      const int r = idx >> 7, c = idx & 127;
      const long m = m0 + r; const int n = n0 + c;
      if (m < g.M && n < g.N) {
        float v = sC[r * EPI_LD + c];
        if (flags & SODT_EPI_BIAS) v += g.bias[n];
        const long b = m / g.det_hw, p = m - b * g.det_hw;
        const int a = n / g.det_no, o = n - a * g.det_no;
        out[((b * g.det_na + a) * g.det_hw + p) * g.det_no + o] = v;
      }
    }
    return;
  }

  constexpr int CPR = BN / KPL;   // chunks per row
  for (int idx = tid; idx < BM * CPR; idx += 256) {
    const int r = idx / CPR, c = (idx - r * CPR) * KPL;
    const long m = m0 + r; const int n = n0 + c;
    if (m >= g.M || n >= g.N) continue;
    float v[KPL];
#pragma unroll
    for (int j = 0; j < KPL; j += 4) {
      const float4 t = *(const float4*)(sC + r * EPI_LD + c + j);
      v[j] = t.x; v[j + 1] = t.y; v[j + 2] = t.z; v[j + 3] = t.w;
    }
    epi_chunk<T, CF>(g, flags, m, n, v, hw);
  }
}

// ---------------------------------------------------------------------------------
// A-stationary variant for short contractions (K * sizeof(T) <= 768 bytes: every C = 192 GEMM of
// stage 1, C = 384 in bf16, the head's 1x1 convs).  One workgroup keeps its 64 rows of A -- the whole
// K extent -- in LDS and walks ALL column tiles of the output: A is read from HBM exactly once, the
// small weight matrix streams from L2 with a register-staged prefetch of the next tile under the
// MFMAs and the epilogue of the current one, and the workgroup lives long enough to keep stores and
// loads in flight back to back.  These GEMMs are HBM-bound (AI = N*K/(N+K) < 200 flop/B), so the
// target is the output-store rate, not the MFMA rate.
// ---------------------------------------------------------------------------------
constexpr int AS_BM = 64;

template <typename T, int BN_>
__global__ __launch_bounds__(256, 2) void gemm_as_kernel(const sodt_gemm_args g) {
  constexpr int KPL = TT<T>::KPL;
  constexpr int MK = TT<T>::MMA_K;
  constexpr int NSUB = BN_ / 32;               // 16-col subtiles per wave (2 waves across N)
  constexpr int ELD = BN_ + 4;                 // padded f32 row of the staged accumulator tile
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  const int KB = g.K * (int)sizeof(T);         // bytes per row, multiple of 128
  const int CPRK = KB >> 4;                    // 16-byte chunks per row
  unsigned char* sA = dsm;                     // [64][KB]   swizzled
  unsigned char* sW = dsm + AS_BM * KB;        // [BN][KB]   swizzled; aliased by the f32 staging tile
  float* sC = (float*)sW;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int fr = lane & 15, fg = lane >> 4;
  const long m0 = (long)blockIdx.x * AS_BM;
  const int hw = g.a.Ho * g.a.Wo;
  const int flags = g.flags;

  // ---- A tile: every 16-byte chunk finds its K-segment and (spatially mapped) source row
  for (int id = tid; id < AS_BM * CPRK; id += 256) {
    const int r = id / CPRK, c = id - r * CPRK;
    const long m = m0 + r;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (m < g.M) {
      int kcol = c * KPL, si = 0;
      sodt_seg sg = g.a.s[0];
#pragma unroll
      for (int j = 1; j < SODT_MAX_SEG; ++j) {
        if (j < g.a.nseg && si == j - 1 && kcol >= sg.klen) { kcol -= sg.klen; si = j; sg = g.a.s[j]; }
      }
      RowGeo geo; geo.ok = true; geo.b = 0; geo.y = 0; geo.x = 0;
      if (g.a.spatial) {
        const int b = (int)(m / hw), rem = (int)(m - (long)b * hw);
        geo.b = b; geo.y = rem / g.a.Wo; geo.x = rem - geo.y * g.a.Wo;
      }
      const long sr = seg_src_row(sg, geo, g.a.spatial, m);
      if (sr >= 0) v = *(const uint4*)((const T*)sg.p + sr * sg.ld + kcol);
    }
    *(uint4*)(sA + r * KB + (((c & ~7) | ((c ^ r) & 7)) << 4)) = v;
  }

  // ---- weight-tile prefetch registers (BN_ x KB bytes over 256 threads)
  constexpr int WREGS = (BN_ * (BN_ == 128 ? 384 : 768) / 16 + 255) / 256;   // BN 128 <-> KB <= 384, BN 64 <-> KB <= 768
  uint4 wreg[WREGS];
  const int wchunks = BN_ * CPRK;
  auto load_w = [&](int n0) {
#pragma unroll
    for (int i = 0; i < WREGS; ++i) {
      const int id = tid + i * 256;
      wreg[i] = make_uint4(0, 0, 0, 0);
      if (id < wchunks) {
        const int r = id / CPRK, c = id - r * CPRK;
        if (n0 + r < g.N) wreg[i] = *(const uint4*)((const T*)g.W + (long)(n0 + r) * g.ldw + c * KPL);
      }
    }
  };
  auto store_w = [&]() {
#pragma unroll
    for (int i = 0; i < WREGS; ++i) {
      const int id = tid + i * 256;
      if (id < wchunks) {
        const int r = id / CPRK, c = id - r * CPRK;
        *(uint4*)(sW + r * KB + (((c & ~7) | ((c ^ r) & 7)) << 4)) = wreg[i];
      }
    }
  };

  const int ntiles = (g.N + BN_ - 1) / BN_;
  const int nkb = g.K / MK;
  load_w(0);
  for (int jt = 0; jt < ntiles; ++jt) {
    const int n0 = jt * BN_;
    store_w();
    __syncthreads();                                   // A (first pass) and W(jt) visible
    if (jt + 1 < ntiles) load_w(n0 + BN_);   // in flight under the MFMAs + epilogue
    f32x4 acc[2][NSUB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NSUB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kb = 0; kb < nkb; ++kb) {
      const int ch = kb * 4 + fg;
      uint4 fa[2], fb[NSUB];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = wr * 32 + i * 16 + fr;
        fa[i] = *(const uint4*)(sA + r * KB + (((ch & ~7) | ((ch ^ r) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < NSUB; ++j) {
        const int c = wc * (BN_ / 2) + j * 16 + fr;
        fb[j] = *(const uint4*)(sW + c * KB + (((ch & ~7) | ((ch ^ c) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NSUB; ++j) mma16<T>(acc[i][j], fa[i], fb[j]);
    }
    __syncthreads();                                   // every wave is done with sW -> reuse it for staging
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NSUB; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          sC[(wr * 32 + i * 16 + fg * 4 + r) * ELD + wc * (BN_ / 2) + j * 16 + fr] = acc[i][j][r];
    __syncthreads();
    if (flags & SODT_EPI_STATS) {
      if (tid < BN_ && n0 + tid < g.N) {
        double s1 = 0.0, s2 = 0.0;
        const int rmax = (int)((g.M - m0) < AS_BM ? (g.M - m0) : AS_BM);
        for (int r = 0; r < rmax; ++r) {
          const float v = sC[r * ELD + tid];
          s1 += v; s2 += (double)v * v;
        }
        atomicAdd(g.stats + n0 + tid, s1);
        atomicAdd(g.stats + g.N + n0 + tid, s2);
      }
    }
    constexpr int CPR = BN_ / KPL;
    for (int idx = tid; idx < AS_BM * CPR; idx += 256) {
      const int r = idx / CPR, c = (idx - r * CPR) * KPL;
      const long m = m0 + r; const int n = n0 + c;
      if (m >= g.M || n >= g.N) continue;
      float v[KPL];
#pragma unroll
      for (int j = 0; j < KPL; j += 4) {
        const float4 t = *(const float4*)(sC + r * ELD + c + j);
        v[j] = t.x; v[j + 1] = t.y; v[j + 2] = t.z; v[j + 3] = t.w;
      }
      epi_chunk<T>(g, flags, m, n, v, hw);
    }
    __syncthreads();                                   // staging consumed before the next W tile lands
  }
}

template <typename T, int BN_>
int launch_as(const sodt_gemm_args* g, hipStream_t st) {
  const int KB = g->K * (int)sizeof(T);
  const int wbytes = BN_ * KB, sbytes = AS_BM * (BN_ + 4) * 4;
  const int lds = AS_BM * KB + (wbytes > sbytes ? wbytes : sbytes);
  static int max_set = 0;
  if (lds > max_set) {
    if (hipFuncSetAttribute((const void*)gemm_as_kernel<T, BN_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) != hipSuccess)
      return SODT_EINVAL;
    max_set = 160 * 1024;
  }
  const long blocks = ((long)g->M + AS_BM - 1) / AS_BM;
  hipLaunchKernelGGL((gemm_as_kernel<T, BN_>), dim3((unsigned)blocks), dim3(256), lds, st, *g);
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

// ---------------------------------------------------------------------------------
// Persistent weight-stationary variant (the workhorse for K * sizeof(T) <= 384 bytes, i.e. every
// C = 192 GEMM of stage 1 in bf16).  A workgroup owns ONE 128-column tile of the output for its whole
// life: the W tile (128 x K) sits in LDS, 64-row blocks of A stream through a single LDS buffer with a
// register prefetch of the next block (issued before the MFMAs of the current one), and the epilogue
// runs straight out of the accumulators.  MFMA operands are swapped (A-operand = W rows, B-operand =
// activation rows) and the W rows are fed in a permuted order so that each lane ends up with 8
// consecutive output columns of one row -> one 16-byte store per lane, 64 contiguous bytes per row
// and wave instruction, no LDS staging and one barrier pair per row block.  Workgroups that walk the
// same row blocks (one per column tile) are placed on one XCD so that A is fetched from HBM once.
// ---------------------------------------------------------------------------------
// CF: compile-time epilogue flags (or -1); SIMPLE: A is one plain row-major tensor (no segments / spatial map)
template <typename T, int BN_, bool STATS, int CF, bool SIMPLE>
__global__ __launch_bounds__(256, 2) void gemm_bs_kernel(const sodt_gemm_args g, const int teams_per_xcd) {
  constexpr int KPL = TT<T>::KPL;
  constexpr int MK = TT<T>::MMA_K;
  constexpr int NP = BN_ / 64;                 // 32-column "pairs" per wave (2 waves across N)
  constexpr int AREGS = (AS_BM * 384 / 16 + 255) / 256;   // 6 (KB <= 384)
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  __shared__ sodt_seg sSeg[SODT_MAX_SEG];
  const int KB = g.K * (int)sizeof(T);
  const int CPRK = KB >> 4;
  unsigned char* sW = dsm;                     // [BN][KB] swizzled, resident
  unsigned char* sA = dsm + BN_ * KB;          // [64][KB] swizzled, one row block

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int fr = lane & 15, fg = lane >> 4;
  const int hw = g.a.Ho * g.a.Wo;
  const int flags = CF >= 0 ? CF : g.flags;
  const int ntiles = (g.N + BN_ - 1) / BN_;
  // block -> (team, column tile): the ntiles members of a team share an XCD (blockIdx % 8) and walk the same row blocks
  const int xcd = blockIdx.x & 7, ix = blockIdx.x >> 3;
  const int team_local = ix / ntiles, jt = ix - team_local * ntiles;
  if (team_local >= teams_per_xcd) return;
  const int team = team_local * 8 + xcd, nteams = teams_per_xcd * 8;
  const int n0 = jt * BN_;
  const long nrb = ((long)g.M + AS_BM - 1) / AS_BM;

  if (tid == 0) {
#pragma unroll
    for (int j = 0; j < SODT_MAX_SEG; ++j) sSeg[j] = g.a.s[j];
  }
  // ---- resident W tile
  for (int id = tid; id < BN_ * CPRK; id += 256) {
    const int r = id / CPRK, c = id - r * CPRK;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (n0 + r < g.N) v = *(const uint4*)((const T*)g.W + (long)(n0 + r) * g.ldw + c * KPL);
    *(uint4*)(sW + r * KB + (((c & ~7) | ((c ^ r) & 7)) << 4)) = v;
  }

  uint4 areg[AREGS];
  auto load_a = [&](long rb) {
    const long m0 = rb * AS_BM;
#pragma unroll
    for (int i = 0; i < AREGS; ++i) {
      const int id = tid + i * 256;
      areg[i] = make_uint4(0, 0, 0, 0);
      if (id < AS_BM * CPRK) {
        const int r = id / CPRK, c = id - r * CPRK;
        const long m = m0 + r;
        if (SIMPLE) {
          if (m < g.M) areg[i] = *(const uint4*)((const T*)g.a.s[0].p + m * g.a.s[0].ld + c * KPL);
        } else if (m < g.M) {
          int kcol = c * KPL, si = 0;
          while (si + 1 < g.a.nseg && kcol >= sSeg[si].klen) { kcol -= sSeg[si].klen; ++si; }
          const sodt_seg sg = sSeg[si];
          RowGeo geo; geo.ok = true; geo.b = 0; geo.y = 0; geo.x = 0;
          if (g.a.spatial) {
            const int b = (int)(m / hw), rem = (int)(m - (long)b * hw);
            geo.b = b; geo.y = rem / g.a.Wo; geo.x = rem - geo.y * g.a.Wo;
          }
          const long sr = seg_src_row(sg, geo, g.a.spatial, m);
          if (sr >= 0) areg[i] = *(const uint4*)((const T*)sg.p + sr * sg.ld + kcol);
        }
      }
    }
  };
  auto store_a = [&]() {
#pragma unroll
    for (int i = 0; i < AREGS; ++i) {
      const int id = tid + i * 256;
      if (id < AS_BM * CPRK) {
        const int r = id / CPRK, c = id - r * CPRK;
        *(uint4*)(sA + r * KB + (((c & ~7) | ((c ^ r) & 7)) << 4)) = areg[i];
      }
    }
  };

  // per-lane column statistics (SODT_EPI_STATS): lane owns columns n0 + wc*BN/2 + p*32 + 8*fg + 0..7
  float st1[NP][8], st2[NP][8];
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int j = 0; j < 8; ++j) { st1[p][j] = 0.f; st2[p][j] = 0.f; }

  // epilogue operands of this lane's fixed columns: bias in registers for the whole kernel; the residual / gelu'
  // source chunks of the current row block are fetched BEFORE its MFMAs so their latency is never exposed
  constexpr int NCH = 8 / KPL;                         // 16-byte chunks per 8 output columns
  float breg[NP][8];
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int n = n0 + wc * (BN_ / 2) + p * 32 + 8 * fg + j;
      breg[p][j] = ((flags & SODT_EPI_BIAS) && n < g.N) ? g.bias[n] : 0.f;
    }
  uint4 rres[2][NP][NCH], raux[2][NP][NCH];
  const int eflags = flags & ~(SODT_EPI_BIAS | SODT_EPI_RESID | SODT_EPI_DGELU);

  const int nkb = g.K / MK;
  long rb = team;
  __syncthreads();                                     // segment table visible
  if (rb < nrb) load_a(rb);
  for (; rb < nrb; rb += nteams) {
    __syncthreads();                                   // previous block's MFMAs are done with sA
    store_a();
    __syncthreads();
    if (rb + nteams < nrb) load_a(rb + nteams);        // in flight under the MFMAs + epilogue
    if (flags & (SODT_EPI_RESID | SODT_EPI_DGELU)) {
#pragma unroll
      for (int sm = 0; sm < 2; ++sm) {
        const long m = rb * AS_BM + wr * 32 + sm * 16 + fr;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const int n = n0 + wc * (BN_ / 2) + p * 32 + 8 * fg;
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            rres[sm][p][c] = make_uint4(0, 0, 0, 0);
            raux[sm][p][c] = make_uint4(0, 0, 0, 0);
            if (m < g.M && n + c * KPL < g.N) {
              if (flags & SODT_EPI_RESID) {
                const long rr = g.rmod > 0 ? (m % g.rmod) : m;
                rres[sm][p][c] = *(const uint4*)((const T*)g.R + rr * g.ldr + n + c * KPL);
              }
              if (flags & SODT_EPI_DGELU) raux[sm][p][c] = *(const uint4*)((const T*)g.aux + m * g.ldaux + n + c * KPL);
            }
          }
        }
      }
    }
    f32x4 acc[NP][2][2];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int sm = 0; sm < 2; ++sm) acc[p][s2][sm] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kb = 0; kb < nkb; ++kb) {
      const int ch = kb * 4 + fg;
      uint4 fw[NP][2], fx[2];
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int nl = wc * (BN_ / 2) + p * 32 + 8 * (fr >> 2) + 4 * s2 + (fr & 3);   // permuted W row fed as MFMA row fr
          fw[p][s2] = *(const uint4*)(sW + nl * KB + (((ch & ~7) | ((ch ^ nl) & 7)) << 4));
        }
#pragma unroll
      for (int sm = 0; sm < 2; ++sm) {
        const int r = wr * 32 + sm * 16 + fr;
        fx[sm] = *(const uint4*)(sA + r * KB + (((ch & ~7) | ((ch ^ r) & 7)) << 4));
      }
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int sm = 0; sm < 2; ++sm) mma16<T>(acc[p][s2][sm], fw[p][s2], fx[sm]);
    }
    // ---- epilogue from registers: lane (fg, fr) holds out[m = .. + fr][n .. n+7]
    const long m0 = rb * AS_BM;
#pragma unroll
    for (int sm = 0; sm < 2; ++sm) {
      const long m = m0 + wr * 32 + sm * 16 + fr;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int n = n0 + wc * (BN_ / 2) + p * 32 + 8 * fg;
        if (m < g.M && n < g.N) {
          if (STATS) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float a = acc[p][0][sm][j], b = acc[p][1][sm][j];
              st1[p][j] += a; st2[p][j] += a * a; st1[p][4 + j] += b; st2[p][4 + j] += b * b;
            }
          }
          float v8[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) { v8[j] = acc[p][0][sm][j] + breg[p][j]; v8[4 + j] = acc[p][1][sm][j] + breg[p][4 + j]; }
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            float v[TT<T>::KPL];
#pragma unroll
            for (int j = 0; j < KPL; ++j) v[j] = v8[c * KPL + j];
            if (flags & SODT_EPI_DGELU) {
              float x[TT<T>::KPL];
              unpack<T>(raux[sm][p][c], x);
#pragma unroll
              for (int j = 0; j < KPL; ++j) v[j] *= dgelu_t<T>(x[j]);
            }
            if (flags & SODT_EPI_RESID) {
              float x[TT<T>::KPL];
              unpack<T>(rres[sm][p][c], x);
#pragma unroll
              for (int j = 0; j < KPL; ++j) v[j] += x[j];
            }
            if (n + c * KPL < g.N)
              epi_chunk<T, (CF >= 0 ? (CF & ~(SODT_EPI_BIAS | SODT_EPI_RESID | SODT_EPI_DGELU)) : -1)>(g, eflags, m, n + c * KPL, v, hw);
          }
        }
      }
    }
  }
  if (STATS) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float a = group16_sum(st1[p][j]), b = group16_sum(st2[p][j]);
        const int n = n0 + wc * (BN_ / 2) + p * 32 + 8 * fg + j;
        if (fr == 0 && n < g.N) {
          double* st = g.stats + (size_t)((blockIdx.x * 4 + (threadIdx.x >> 6)) % SODT_STATS_REPL) * 2 * g.N;
          atomicAdd(st + n, (double)a);
          atomicAdd(st + g.N + n, (double)b);
        }
      }
  }
}

template <typename T, int BN_, bool STATS, int CF, bool SIMPLE>
int launch_bs(const sodt_gemm_args* g, hipStream_t st) {
  const int KB = g->K * (int)sizeof(T);
  const int lds = (BN_ + AS_BM) * KB;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gemm_bs_kernel<T, BN_, STATS, CF, SIMPLE>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess)
      return SODT_EINVAL;
    attr_set = true;
  }
  const int ntiles = (g->N + BN_ - 1) / BN_;
  const long nrb = ((long)g->M + AS_BM - 1) / AS_BM;
  const int per_cu = lds <= 76 * 1024 ? 2 : 1;
  int teams_per_xcd = (32 * per_cu) / ntiles;          // 32 CUs per XCD
  if (teams_per_xcd < 1) teams_per_xcd = 1;
  const long max_teams = (nrb + 7) / 8;
  if (teams_per_xcd > max_teams) teams_per_xcd = (int)max_teams;
  const int blocks = teams_per_xcd * ntiles * 8;
  hipLaunchKernelGGL((gemm_bs_kernel<T, BN_, STATS, CF, SIMPLE>), dim3(blocks), dim3(256), lds, st, *g, teams_per_xcd);
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

// ---------------------------------------------------------------------------------
// TN: dW[n][k] += sum_m dY[m][n] * X[m][k].  128(n) x 128(k) tile, the M range of this
// split walked in steps of BMS rows (64 bf16 / 32 f32).  Both operands are consumed
// "transposed" (contraction index = LDS row): bf16 through ds_read_b64_tr_b16, f32
// through four strided ds_read_b32.  LDS rows are padded by 16 bytes.
// ---------------------------------------------------------------------------------
template <typename T> struct TNGeo;
template <> struct TNGeo<bf16> { static constexpr int BMS = 64, LDSROW = 128 * 2 + 16; };
template <> struct TNGeo<float> { static constexpr int BMS = 32, LDSROW = 128 * 4 + 16; };

template <typename T>
__device__ __forceinline__ uint4 tn_frag(const unsigned char* tile, int kb, int col0, int lane);
template <>
__device__ __forceinline__ uint4 tn_frag<bf16>(const unsigned char* tile, int kb, int col0, int lane) {
  // k (= LDS row) = kb*32 + 8*(lane>>4) + j ; column = col0 + (lane & 15)
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const unsigned char* a = tile + (kb * 32 + 8 * g + q) * TNGeo<bf16>::LDSROW + (col0 + 4 * p) * 2;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  union { s16x4 v; uint2 u; } lo, hi;
  lo.v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
  hi.v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * TNGeo<bf16>::LDSROW));
  return make_uint4(lo.u.x, lo.u.y, hi.u.x, hi.u.y);
}
template <>
__device__ __forceinline__ uint4 tn_frag<float>(const unsigned char* tile, int kb, int col0, int lane) {
  const int g = lane >> 4, c = col0 + (lane & 15);
  const unsigned char* a = tile + (kb * 16 + 4 * g) * TNGeo<float>::LDSROW + c * 4;
  uint4 r;
  r.x = *(const uint32_t*)(a);
  r.y = *(const uint32_t*)(a + TNGeo<float>::LDSROW);
  r.z = *(const uint32_t*)(a + 2 * TNGeo<float>::LDSROW);
  r.w = *(const uint32_t*)(a + 3 * TNGeo<float>::LDSROW);
  return r;
}

template <typename T>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const sodt_gemm_tn_args g) {
  constexpr int KPL = TT<T>::KPL;
  constexpr int BMS = TNGeo<T>::BMS;
  constexpr int LDSROW = TNGeo<T>::LDSROW;
  constexpr int TILE = BMS * LDSROW;
  constexpr int CPR = 128 / KPL;          // 16-byte chunks per tile row
  constexpr int RPT = 256 / CPR;          // rows covered per pass by the 256 threads
  constexpr int NLD = BMS / RPT;          // = 4 chunks per thread and operand
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILE];
  unsigned char* sY = smem;               // [2][TILE]
  unsigned char* sX = smem + 2 * TILE;    // [2][TILE]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int ntk = (g.K + 127) / 128;
  // all tiles of one M-slice are adjacent logical blocks -> one XCD: the slice's dY / X rows are fetched
  // from HBM once and re-read by the other tiles out of that XCD's L2
  const int ntiles_ = ntk * ((g.N + 127) / 128);
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int split = lid / ntiles_, tile_ = lid - split * ntiles_;
  const int tk = tile_ % ntk, tn = tile_ / ntk;
  const int n0 = tn * 128, k0 = tk * 128;
  const long rows_per = (((g.M + g.splits - 1) / g.splits) + BMS - 1) / BMS * BMS;
  const long mbeg = (long)split * rows_per;
  const long mend = (mbeg + rows_per < g.M) ? (mbeg + rows_per) : g.M;
  if (mbeg >= mend) return;

  // wave-level skip of sub-tiles entirely outside N / K (N = 192, 576 ... tails)
  const bool wave_live = (n0 + wr * 64 < g.N) && (k0 + wc * 64 < g.K);

  // loader: this thread's chunk column is fixed for the whole kernel
  const int lrow = tid / CPR, lch = tid - lrow * CPR;
  const int ycol = n0 + lch * KPL;
  const bool yok = ycol < g.N;
  const int xcol = k0 + lch * KPL;
  // locate the X segment of this column
  int xs_i = -1, xs_off = 0;
  {
    int base = 0;
#pragma unroll
    for (int j = 0; j < SODT_MAX_SEG; ++j) {
      if (j < g.x.nseg) {
        if (xs_i < 0 && xcol < base + g.x.s[j].klen && xcol < g.K) { xs_i = j; xs_off = xcol - base; }
        base += g.x.s[j].klen;
      }
    }
  }
  const sodt_seg seg = pick_seg(g.x, xs_i < 0 ? 0 : xs_i);
  const int hw = g.x.Ho * g.x.Wo;

  uint4 ry[NLD], rx[NLD];
  auto load_regs = [&](long mb) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const long m = mb + lrow + RPT * i;
      ry[i] = make_uint4(0, 0, 0, 0);
      rx[i] = make_uint4(0, 0, 0, 0);
      if (m < mend) {
        if (yok) ry[i] = *(const uint4*)((const T*)g.dY + m * g.ldy + ycol);
        if (xs_i >= 0) {
          RowGeo r; r.ok = true; r.b = 0; r.y = 0; r.x = 0;
          if (g.x.spatial) {
            const int b = (int)(m / hw), rem = (int)(m - (long)b * hw);
            r.b = b; r.y = rem / g.x.Wo; r.x = rem - r.y * g.x.Wo;
          }
          const long sr = seg_src_row(seg, r, g.x.spatial, m);
          if (sr >= 0) rx[i] = *(const uint4*)((const T*)seg.p + sr * seg.ld + xs_off);
        }
      }
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int off = (lrow + RPT * i) * LDSROW + lch * 16;
      *(uint4*)(sY + buf * TILE + off) = ry[i];
      *(uint4*)(sX + buf * TILE + off) = rx[i];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const bool do_bias = (g.dbias != nullptr) && (tk == 0) && tid < 128 && (n0 + tid < g.N);

  const int nsteps = (int)((mend - mbeg + BMS - 1) / BMS);
  load_regs(mbeg);
  store_lds(0);
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const int buf = s & 1;
    if (s + 1 < nsteps) load_regs(mbeg + (long)(s + 1) * BMS);
    const unsigned char* ty = sY + buf * TILE;
    const unsigned char* tx = sX + buf * TILE;
    if (wave_live) {
#pragma unroll
      for (int kb = 0; kb < BMS / TT<T>::MMA_K; ++kb) {
        uint4 fa[4], fb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          fa[i] = tn_frag<T>(ty, kb, wr * 64 + i * 16, lane);
          fb[i] = tn_frag<T>(tx, kb, wc * 64 + i * 16, lane);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) mma16<T>(acc[i][j], fa[i], fb[j]);
      }
    }
    if (do_bias) {
      for (int r = 0; r < BMS; ++r) bsum += to_f(*(const T*)(ty + r * LDSROW + tid * sizeof(T)));
    }
    if (s + 1 < nsteps) store_lds(buf ^ 1);
    __syncthreads();
  }

  if (do_bias) atomicAdd(g.dbias + n0 + tid, bsum);
  if (!wave_live) return;
  const int fr = lane & 15, fg = lane >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + wc * 64 + j * 16 + fr;
      if (k >= g.K) continue;
      int kk = k;
      if (g.kperm_t > 1) kk = (k % g.kperm_c) * g.kperm_t + k / g.kperm_c;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wr * 64 + i * 16 + fg * 4 + r;
        if (n < g.N) atomicAdd(g.dW + (long)n * g.lddw + kk, acc[i][j][r]);
      }
    }
}

// ---------------------------------------------------------------------------------
// TN, large tile: 256(n) x 192(k) per workgroup of 8 waves (4 x 2, 64 x 96 per wave = 4 x 6 MFMA tiles).
// The weight-gradient GEMMs are bound by how often dY and X are re-read (once per column / row tile of
// dW): with K = 192 (every C = 192 layer) X is read once per n-tile and dY exactly once, half the bytes
// of the 128 x 128 kernel above.  Same transposed-operand reads, same double-buffered register prefetch.
// ---------------------------------------------------------------------------------
template <typename T> struct TN2Geo;
template <> struct TN2Geo<bf16> { static constexpr int BMS = 64, YROW = 256 * 2 + 16, XROW = 192 * 2 + 16; };
template <> struct TN2Geo<float> { static constexpr int BMS = 32, YROW = 256 * 4 + 16, XROW = 192 * 4 + 16; };

template <typename T>
__device__ __forceinline__ uint4 tn2_frag(const unsigned char* tile, int rowbytes, int kb, int col0, int lane);
template <>
__device__ __forceinline__ uint4 tn2_frag<bf16>(const unsigned char* tile, int rowbytes, int kb, int col0, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const unsigned char* a = tile + (kb * 32 + 8 * g + q) * rowbytes + (col0 + 4 * p) * 2;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  union { s16x4 v; uint2 u; } lo, hi;
  lo.v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
  hi.v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * rowbytes));
  return make_uint4(lo.u.x, lo.u.y, hi.u.x, hi.u.y);
}
template <>
__device__ __forceinline__ uint4 tn2_frag<float>(const unsigned char* tile, int rowbytes, int kb, int col0, int lane) {
  const unsigned char* a = tile + (kb * 16 + 4 * (lane >> 4)) * rowbytes + (col0 + (lane & 15)) * 4;
  uint4 r;
  r.x = *(const uint32_t*)(a);
  r.y = *(const uint32_t*)(a + rowbytes);
  r.z = *(const uint32_t*)(a + 2 * rowbytes);
  r.w = *(const uint32_t*)(a + 3 * rowbytes);
  return r;
}

template <typename T>
__global__ __launch_bounds__(512, 2) void gemm_tn2_kernel(const sodt_gemm_tn_args g) {
  constexpr int KPL = TT<T>::KPL;
  constexpr int BMS = TN2Geo<T>::BMS, YROW = TN2Geo<T>::YROW, XROW = TN2Geo<T>::XROW;
  constexpr int YT = BMS * YROW, XT = BMS * XROW;
  constexpr int YCPR = 256 / KPL, XCPR = 192 / KPL;             // 16-byte chunks per tile row
  constexpr int NY = BMS * YCPR / 512, NX = BMS * XCPR / 512;   // chunks per thread: 4 and 3
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  __shared__ sodt_seg sSeg[SODT_MAX_SEG];
  unsigned char* sY = dsm;               // [2][YT]
  unsigned char* sX = dsm + 2 * YT;      // [2][XT]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;                        // 4 waves over n, 2 over k
  if (tid == 0) {
#pragma unroll
    for (int j = 0; j < SODT_MAX_SEG; ++j) sSeg[j] = g.x.s[j];
  }
  const int ntk = (g.K + 191) / 192;
  const int ntiles_ = ntk * ((g.N + 255) / 256);
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int split = lid / ntiles_, tile_ = lid - split * ntiles_;
  const int tk = tile_ % ntk, tn = tile_ / ntk;
  const int n0 = tn * 256, k0 = tk * 192;
  const long rows_per = (((g.M + g.splits - 1) / g.splits) + BMS - 1) / BMS * BMS;
  const long mbeg = (long)split * rows_per;
  const long mend = (mbeg + rows_per < g.M) ? (mbeg + rows_per) : g.M;
  __syncthreads();
  if (mbeg >= mend) return;
  const bool wave_live = (n0 + wr * 64 < g.N) && (k0 + wc * 96 < g.K);
  const int hw = g.x.Ho * g.x.Wo;

  // fixed per-thread chunk columns
  int yrow[NY], ych[NY], xrow[NX], xoff[NX], xsi[NX];
#pragma unroll
  for (int i = 0; i < NY; ++i) { const int id = tid + i * 512; yrow[i] = id / YCPR; ych[i] = id - yrow[i] * YCPR; }
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int id = tid + i * 512;
    xrow[i] = id / XCPR;
    const int ch = id - xrow[i] * XCPR;
    int kcol = k0 + ch * KPL, si = 0;
    xsi[i] = -1; xoff[i] = ch;       // xoff keeps the chunk index in its low bits until resolved below
    if (kcol < g.K) {
      while (si + 1 < g.x.nseg && kcol >= sSeg[si].klen) { kcol -= sSeg[si].klen; ++si; }
      xsi[i] = si; 
    }
    xoff[i] = (ch << 16) | (kcol & 0xffff);
  }

  uint4 ry[NY], rx[NX];
  auto load_regs = [&](long mb) {
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      const long m = mb + yrow[i];
      const int col = n0 + ych[i] * KPL;
      ry[i] = make_uint4(0, 0, 0, 0);
      if (m < mend && col < g.N) ry[i] = *(const uint4*)((const T*)g.dY + m * g.ldy + col);
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const long m = mb + xrow[i];
      rx[i] = make_uint4(0, 0, 0, 0);
      if (m < mend && xsi[i] >= 0) {
        const sodt_seg sg = sSeg[xsi[i]];
        RowGeo r; r.ok = true; r.b = 0; r.y = 0; r.x = 0;
        if (g.x.spatial) {
          const int b = (int)(m / hw), rem = (int)(m - (long)b * hw);
          r.b = b; r.y = rem / g.x.Wo; r.x = rem - r.y * g.x.Wo;
        }
        const long sr = seg_src_row(sg, r, g.x.spatial, m);
        if (sr >= 0) rx[i] = *(const uint4*)((const T*)sg.p + sr * sg.ld + (xoff[i] & 0xffff));
      }
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NY; ++i) *(uint4*)(sY + buf * YT + yrow[i] * YROW + ych[i] * 16) = ry[i];
#pragma unroll
    for (int i = 0; i < NX; ++i) *(uint4*)(sX + buf * XT + xrow[i] * XROW + (xoff[i] >> 16) * 16) = rx[i];
  };

  f32x4 acc[4][6];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const bool do_bias = (g.dbias != nullptr) && (tk == 0) && tid < 256 && (n0 + tid < g.N);

  const int nsteps = (int)((mend - mbeg + BMS - 1) / BMS);
  load_regs(mbeg);
  store_lds(0);
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const int buf = s & 1;
    if (s + 1 < nsteps) load_regs(mbeg + (long)(s + 1) * BMS);
    const unsigned char* ty = sY + buf * YT;
    const unsigned char* tx = sX + buf * XT;
    if (wave_live) {
#pragma unroll
      for (int kb = 0; kb < BMS / TT<T>::MMA_K; ++kb) {
        uint4 fa[4], fb[6];
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = tn2_frag<T>(ty, YROW, kb, wr * 64 + i * 16, lane);
#pragma unroll
        for (int j = 0; j < 6; ++j) fb[j] = tn2_frag<T>(tx, XROW, kb, wc * 96 + j * 16, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 6; ++j) mma16<T>(acc[i][j], fa[i], fb[j]);
      }
    }
    if (do_bias) {
      for (int r = 0; r < BMS; ++r) bsum += to_f(*(const T*)(ty + r * YROW + tid * sizeof(T)));
    }
    if (s + 1 < nsteps) store_lds(buf ^ 1);
    __syncthreads();
  }
  if (do_bias) atomicAdd(g.dbias + n0 + tid, bsum);
  if (!wave_live) return;
  const int fr = lane & 15, fg = lane >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int k = k0 + wc * 96 + j * 16 + fr;
      if (k >= g.K) continue;
      int kk = k;
      if (g.kperm_t > 1) kk = (k % g.kperm_c) * g.kperm_t + k / g.kperm_c;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wr * 64 + i * 16 + fg * 4 + r;
        if (n < g.N) atomicAdd(g.dW + (long)n * g.lddw + kk, acc[i][j][r]);
      }
    }
}

template <typename T>
int launch_tn2(const sodt_gemm_tn_args* g, hipStream_t st) {
  constexpr int lds = 2 * TN2Geo<T>::BMS * (TN2Geo<T>::YROW + TN2Geo<T>::XROW);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gemm_tn2_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess)
      return SODT_EINVAL;
    attr_set = true;
  }
  const int tiles = ((g->N + 255) / 256) * ((g->K + 191) / 192);
  hipLaunchKernelGGL((gemm_tn2_kernel<T>), dim3(tiles * g->splits), dim3(512), lds, st, *g);
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

bool aspec_ok(const sodt_aspec& a, int K, int kpl) {
  if (a.nseg < 1 || a.nseg > SODT_MAX_SEG) return false;
  long tot = 0;
  for (int i = 0; i < a.nseg; ++i) {
    const sodt_seg& s = a.s[i];
    if (s.klen <= 0 || (s.klen % kpl) || (s.ld % kpl) || (((uintptr_t)s.p) & 15)) return false;
    if (a.spatial && (s.Hi <= 0 || s.Wi <= 0 || s.mul < 1 || s.shr < 0)) return false;
    tot += s.klen;
  }
  if (a.spatial && (a.Ho <= 0 || a.Wo <= 0)) return false;
  return tot == K;
}

}  // namespace

// branch-free instantiations for the flag sets the model's plain-tensor GEMMs use; everything else is generic
template <typename T>
int dispatch_bs(const sodt_gemm_args* g, bool simple, hipStream_t st) {
  if (simple) {
    switch (g->flags) {
      case 0: return launch_bs<T, 128, false, 0, true>(g, st);
      case SODT_EPI_BIAS: return launch_bs<T, 128, false, SODT_EPI_BIAS, true>(g, st);
      case SODT_EPI_RESID: return launch_bs<T, 128, false, SODT_EPI_RESID, true>(g, st);
      case SODT_EPI_BIAS | SODT_EPI_RESID: return launch_bs<T, 128, false, SODT_EPI_BIAS | SODT_EPI_RESID, true>(g, st);
      case SODT_EPI_BIAS | SODT_EPI_GELU_DUAL: return launch_bs<T, 128, false, SODT_EPI_BIAS | SODT_EPI_GELU_DUAL, true>(g, st);
      case SODT_EPI_DGELU: return launch_bs<T, 128, false, SODT_EPI_DGELU, true>(g, st);
      default: break;
    }
  }
  return launch_bs<T, 128, false, -1, false>(g, st);
}

extern "C" int sodt_gemm_nt(const sodt_gemm_args* g, int dtype, sodt_stream_t st) {
  if (!g || g->M <= 0 || g->N <= 0 || g->K <= 0) return SODT_EINVAL;
  const int kpl = dtype == SODT_BF16 ? 8 : 4;
  if (!aspec_ok(g->a, g->K, kpl)) return SODT_EINVAL;
  if ((g->ldw % kpl) || (((uintptr_t)g->W) & 15) || !g->C) return SODT_EINVAL;
  if (!(g->flags & SODT_EPI_DETECT)) {
    const int okpl = (g->flags & SODT_EPI_OUT_F32) ? 1 : kpl;
    if ((g->ldc % okpl) || (((uintptr_t)g->C) & 15)) return SODT_EINVAL;
  } else if (g->det_na * g->det_no != g->N || g->det_hw <= 0) return SODT_EINVAL;
  if ((g->flags & SODT_EPI_BIAS) && (!g->bias || (((uintptr_t)g->bias) & 15))) return SODT_EINVAL;
  if ((g->flags & SODT_EPI_RESID) && (!g->R || (g->ldr % kpl))) return SODT_EINVAL;
  if ((g->flags & SODT_EPI_DGELU) && (!g->aux || (g->ldaux % kpl))) return SODT_EINVAL;
  if ((g->flags & SODT_EPI_DRELU) && (!g->aux || (g->ldaux % kpl) || (((uintptr_t)g->aux) & 15) || (g->flags & SODT_EPI_DGELU))) return SODT_EINVAL;
  if ((g->flags & SODT_EPI_GELU_DUAL) && (!g->C2 || (g->ldc2 % kpl))) return SODT_EINVAL;
  if ((g->flags & SODT_EPI_STATS) && !g->stats) return SODT_EINVAL;
  if ((g->flags & SODT_EPI_AFFINE_SILU) && (!g->scale || !g->shift || ((((uintptr_t)g->scale) | ((uintptr_t)g->shift)) & 15))) return SODT_EINVAL;
  if (g->oscatter && !g->a.spatial) return SODT_EINVAL;
  if ((g->flags & SODT_EPI_DGELU_RC) && !(dtype == SODT_BF16 && sodt_nt3_eligible(g))) return SODT_EINVAL;
  if (dtype == SODT_BF16 && (g_variant == 0 || g_variant == 3) && sodt_nt3_eligible(g)) return sodt_nt3_launch(g, (hipStream_t)st);
  // short contraction -> A-stationary kernel (row bytes a multiple of 128 so the XOR swizzle stays in-row)
  {
    const int es = dtype == SODT_BF16 ? 2 : 4;
    const int KB = g->K * es;
    bool seg_ok = true;
    for (int i = 0; i < g->a.nseg; ++i) seg_ok = seg_ok && ((g->a.s[i].klen * es) % 16 == 0);
    if (!(g->flags & SODT_EPI_DETECT) && (KB % 128) == 0 && KB <= 384 && seg_ok && !g_force_tiled) {
      hipStream_t s_ = (hipStream_t)st;
      const int kplv = dtype == SODT_BF16 ? 8 : 4;
      // (that kernel adds the residual before the generic epilogue: a ReLU / ReLU mask must come first, so not with both)
      const bool relu_resid = (g->flags & (SODT_EPI_RELU | SODT_EPI_DRELU)) && (g->flags & SODT_EPI_RESID);
      const bool bs_ok = (g->N % kplv) == 0 && (g->flags & SODT_EPI_OUT_F32) == 0 && g_variant != 2 && !relu_resid;
      if (bs_ok) {
        const bool stt = (g->flags & SODT_EPI_STATS) != 0;
        const bool simple = g->a.nseg == 1 && !g->a.spatial && !g->oscatter;
        if (dtype == SODT_BF16) return stt ? launch_bs<bf16, 128, true, -1, false>(g, s_) : dispatch_bs<bf16>(g, simple, s_);
        if (dtype == SODT_F32) return stt ? launch_bs<float, 128, true, -1, false>(g, s_) : dispatch_bs<float>(g, simple, s_);
        return SODT_EINVAL;
      }
      if (dtype == SODT_BF16) return KB <= 384 ? launch_as<bf16, 128>(g, s_) : launch_as<bf16, 64>(g, s_);
      if (dtype == SODT_F32) return KB <= 384 ? launch_as<float, 128>(g, s_) : launch_as<float, 64>(g, s_);
      return SODT_EINVAL;
    }
  }
  if (dtype == SODT_BF16 && (g_variant == 0 || g_variant == 3) && sodt_nt3_eligible(g)) return sodt_nt3_launch(g, (hipStream_t)st);
  const long tiles = ((long)(g->M + BM - 1) / BM) * ((g->N + BN - 1) / BN);
  if (tiles > 0x7fffffffL) return SODT_EINVAL;
  dim3 grid((unsigned)tiles), block(256);
  const int cf = g->oscatter ? -1 : g->flags;
#define NT_CASE(TY, F) case F: hipLaunchKernelGGL((gemm_nt_kernel<TY, F>), grid, block, 0, (hipStream_t)st, *g); break
#define NT_SWITCH(TY)                                                                                   \
  switch (cf) {                                                                                         \
    NT_CASE(TY, 0); NT_CASE(TY, SODT_EPI_BIAS); NT_CASE(TY, SODT_EPI_RESID);                            \
    NT_CASE(TY, (SODT_EPI_BIAS | SODT_EPI_RESID)); NT_CASE(TY, (SODT_EPI_BIAS | SODT_EPI_GELU_DUAL));   \
    NT_CASE(TY, SODT_EPI_DGELU);                                                                        \
    default: hipLaunchKernelGGL((gemm_nt_kernel<TY, -1>), grid, block, 0, (hipStream_t)st, *g); break;  \
  }
  if (dtype == SODT_BF16) { NT_SWITCH(bf16) }
  else if (dtype == SODT_F32) { NT_SWITCH(float) }
  else return SODT_EINVAL;
#undef NT_SWITCH
#undef NT_CASE
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

extern "C" int sodt_gemm_set_variant(int force_tiled) {
  g_force_tiled = force_tiled == 1;
  g_variant = force_tiled;
  return SODT_OK;
}

extern "C" int sodt_gemm_tn(const sodt_gemm_tn_args* g, int dtype, sodt_stream_t st) {
  if (!g || g->M <= 0 || g->N <= 0 || g->K <= 0 || g->splits < 1 || !g->dW || !g->dY) return SODT_EINVAL;
  const int kpl = dtype == SODT_BF16 ? 8 : 4;
  if (!aspec_ok(g->x, g->K, kpl)) return SODT_EINVAL;
  if ((g->ldy % kpl) || g->ldy < (g->N + kpl - 1) / kpl * kpl || (((uintptr_t)g->dY) & 15)) return SODT_EINVAL;
  if (g->kperm_t > 1 && (g->kperm_c <= 0 || g->kperm_c * g->kperm_t != g->K)) return SODT_EINVAL;
  if (dtype == SODT_BF16 && g_variant == 0 && (g->N % 8) == 0 && (g->K % 8) == 0 && g->M >= 1024)
    return sodt_tn3_launch(g, (hipStream_t)st);
  if (!g_force_tiled && (g->N <= 192 || g->K <= 192)) {   // short side <= 192: the 256 x 192 tile reads each operand (almost) once
    if (dtype == SODT_BF16) return launch_tn2<bf16>(g, (hipStream_t)st);
    if (dtype == SODT_F32) return launch_tn2<float>(g, (hipStream_t)st);
    return SODT_EINVAL;
  }
  const int tiles = ((g->N + 127) / 128) * ((g->K + 127) / 128);
  dim3 grid(tiles * g->splits), block(256);
  if (dtype == SODT_BF16) {
    hipLaunchKernelGGL(gemm_tn_kernel<bf16>, grid, block, 0, (hipStream_t)st, *g);
  } else if (dtype == SODT_F32) {
    hipLaunchKernelGGL(gemm_tn_kernel<float>, grid, block, 0, (hipStream_t)st, *g);
  } else return SODT_EINVAL;
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}
