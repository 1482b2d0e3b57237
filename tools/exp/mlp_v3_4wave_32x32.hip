// EXPERIMENT (round 5, not built into the library): the fused linear MLP as FOUR waves per CU (one per SIMD, 512 registers), 64 tokens per
// wave, 32x32x16 MFMAs, fragment reads two groups ahead.  Correct on the first run (tests/test_mlp_fused_gpu.py: 30 / 30) and SLOWER than
// the shipped 8-wave / 16x16x32 kernel of csrc/mlp.hip: 0.480 / 0.532 ms (inference / training form) against 0.430 / 0.490 ms, same box
// (gpurun_out/r05_mb_mlp5.log).  Stamps: 3,523 cycles per step = 1,029 for the six LDS-DMA issues + 2,369 for the 537 instructions of the
// step body (4.4 cycles per instruction: the per-wave issue limit); in the training form the DMA issues take 294 and the body 3,055 -
// every vector-memory instruction (DMA piece, 16-byte store) stalls its wave ~170 cycles while the CU's memory path drains, and with
// one wave per SIMD nothing else issues meanwhile.  Kept for the record: profiles/r05_mlp_stamps.md, DESIGN.md section 4.4.
// Fused linear MLP of a Swin block, bf16 (backbone_vit.py:884-890 with the block's residual add of :1128):
//
//     out = resid + fc2( GELU( fc1(xn) ) )          xn, resid, out [M][C];  fc1.weight [4C][C], fc2.weight [C][4C]
//
// ONE launch; the 4C-wide hidden activation never goes to HBM unless the caller asks for it (training: GELU(h) is the operand
// of fc2's weight gradient).  The two-GEMM chain it replaces wrote the hidden tensor once and read it back once in the forward
// alone (805 MB each way per stage-1 block at B=8 @1024^2).
//
// Structure (gfx950, one persistent workgroup of FOUR waves per CU - one wave per SIMD, 512 registers):
//   * a workgroup owns a tile of 256 token rows, wave w the rows 64 w .. 64 w + 63; the tile's xn rows sit in LDS for the whole
//     tile (3 K-blocks of [256][64 channels], 96 KB, filled by LDS-DMA, XOR-swizzled on the 16-byte chunk like gemm3.hip's A image;
//     every wave reads only its own rows, so the next tile's rows are requested by the wave itself right after its last fc1 read).
//   * the hidden dimension is walked in 24 slabs of 32 columns.  Per slab W1 rows [32][192] and W2 columns [192][32] arrive by
//     LDS-DMA in a two-stage ring (12 KB + 12 KB per stage; one raw s_barrier and one counted s_waitcnt vmcnt per step - the next
//     step's DMAs and this step's GELU(h) stores stay in flight across the barrier);
//         h^T (32 x 64 tokens)  = W1slab . xn^T        24 MFMA 32x32x16 per wave, bias b1 as the accumulator's initial value
//         g = GELU(h)                                  in registers: a lane of a TRANSPOSED 32x32 product holds 16 hidden columns of
//                                                      one token, and with W1's rows fed to the MFMA in a permuted order they are the
//                                                      two k-contiguous 8-column B operands of the next product (and two 16-byte
//                                                      stores of GELU(h))
//         out^T (192 x 64 tokens) += W2slab . g^T      24 MFMA per wave; W2's rows permuted so that a lane ends with 16 consecutive
//                                                      output channels of a token
//   * out^T starts as b2; the residual rows are loaded two steps before the end of the tile (they land under fc2 of the last slabs) and
//     added in the epilogue, which stores from registers, 32 bytes per lane and channel group.
// Why this shape (round 5, profiles/r05_mlp_stamps.md, r05_mlp_ablation.md): the 8-wave / 16x16x32 first version was bound by
// instruction ISSUE - a SIMD issues about one vector instruction per 4 cycles whichever of its waves it comes from, an MFMA
// holds the issue port for 8 cycles whatever its size, and the two waves' issue cycles simply added up (3.1-3.5 K per step against
// 1,536 matrix-pipe cycles).  32x32x16 MFMAs do the same work in half the instructions, 64 tokens per wave halve the weight-fragment
// reads, and one wave per SIMD has nobody to lose arbitration to.
// Fragment reads are inline-asm ds_read_b128 two MFMA groups ahead of their use, behind counted lgkmcnt waits (hipcc drains
// vmcnt(0) before any LDS read it can see while an LDS-DMA is outstanding).  LDS: 96 KB xn + 2 x 24 KB weights + 3.75 KB biases.
//
// Roofline: 16 T C^2 flops over 3 T C 2 B (inference) = 309 GFLOP / 604 MB at stage 1 of B=8 @1024^2: MFMA-bound by AI (512 flop/B);
// with the GELU(h) store (training) 1.41 GB: HBM-priced (0.28 ms at 5 TB/s against 0.12 ms of matrix time).
#include "common.h"
#include "../../include/sodt_hip.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int ML_C = 192, ML_H = 768;
constexpr int ML_BM = 256, ML_HS = 32, ML_NSLAB = ML_H / ML_HS;       // 24 slabs
constexpr int ML_XKB = ML_BM * 128;                                    // one K-block of the xn tile: 32 KiB
constexpr int ML_XN = 0;
constexpr int ML_WR = 3 * ML_XKB;                                      // weight ring
constexpr int ML_W1B = 3 * ML_HS * 128;                                // W1 slab: 3 K-blocks x [32 rows][128 B] = 12 KiB
constexpr int ML_W2B = ML_C * 64;                                      // W2 slab: [192 rows][64 B] = 12 KiB
constexpr int ML_WST = ML_W1B + ML_W2B;                                // 24 KiB per stage
constexpr int ML_B1 = ML_WR + 2 * ML_WST;                              // f32 b1[768]
constexpr int ML_B2 = ML_B1 + ML_H * 4;                                // f32 b2[192]
constexpr int ML_LDS = ML_B2 + ML_C * 4;

__device__ uint4 g_mlp_zero[8];                                        // DMA source of the rows beyond M

__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(lds_void*)p; }
template <int OFF> __device__ __forceinline__ u32x4 lds_rd128(uint32_t addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
#define ML_LGKM(N) do { asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define ML_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")

__device__ __forceinline__ f32x4 mma(const u32x4& w, const u32x4& a, const f32x4& c) {
  union { u32x4 u; bf16x8 v; } uw, ua;
  uw.u = w; ua.u = a;
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(uw.v, ua.v, c, 0, 0, 0);
}

// GELU of the fused kernel: x Phi(x) with Phi(x) ~ 1 / (1 + 2^(xc (A0 + A1 xc^2 + A2 xc^4))), xc = x clamped to +-8 - the logistic
// ("tanh") form with a quartic term, its three constants fitted to the exact erf GELU (nn.GELU() default, backbone_vit.py:872):
// max |error| 2.5e-5 over the whole line (the odd degree-15 erf polynomial of the GEMM epilogues: 1.4e-4; one bf16 ulp at 1: 3.9e-3).
// 7 FMA-pipe instructions + v_exp_f32 + v_rcp_f32 per element instead of 13: the kernel is bound by instruction ISSUE (per step a
// SIMD's two waves needed 2 x (880 VALU + 384 MFMA + 300 LDS) issue cycles against 1,536 matrix-pipe cycles, stamps in
// profiles/r05_mlp_stamps.md), and the activation was more than half of it.  (A two-constant fit, 2.7e-4, is two instructions
// shorter but its error is a smooth bias, not rounding noise: it moved the p95 of the model's gradient-error ratio against the
// reference's own autocast from 1.36 to 1.51 - tests/test_bf16_parity_gpu.py - and was dropped.)  |x| > 8: 2^(+-40) -> -0 or x.
__device__ __forceinline__ float gelu_sig(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -8.0f, 8.0f);
  const float u = xc * xc;
  const float t = fmaf(fmaf(u, 0.001014263f, -0.10677572f), u, -2.3011212f);
  const float e = __builtin_amdgcn_exp2f(xc * t);
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

struct MlpArgs {
  const bf16* xn; const bf16* w1; const float* b1; const bf16* w2; const float* b2; const bf16* resid;
  bf16* out; bf16* hact; long M;
};

#ifdef ML_STAMPS     // diagnostic build only (tools/exp/ab_build.sh mlp -DML_STAMPS stamps): shader-cycle sums per wave
__device__ unsigned long long g_mlp_stamps[256 * 8 * 8];
#define ML_T(V) unsigned long long V; do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(V) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define ML_ACC(I, A, B) st_sum[I] += (B) - (A)
#else
#define ML_T(V) do {} while (0)
#define ML_ACC(I, A, B) do {} while (0)
#endif

template <bool B> struct BoolC { static constexpr bool value = B; };
template <int I> struct IntC { static constexpr int value = I; };
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int N> __device__ __forceinline__ void lgkm_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ f32x16 mma32(const u32x4& w, const u32x4& a, const f32x16& c) {
  union { u32x4 u; bf16x8 v; } uw, ua;
  uw.u = w; ua.u = a;
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(uw.v, ua.v, c, 0, 0, 0);
}

// MFMA 32x32x16 operand maps (cdna_hip_programming.md section 3): A lane l = row (l & 31), k = 8 (l >> 5) + j; B the same with the
// column on the lane; D register e of lane l = row (e & 3) + 8 (e >> 2) + 4 (l >> 5), column l & 31.
// fc1: A row i is hidden column pi(i) of the slab, pi(8 q + 4 hh + r) = 16 (q >> 1) + 8 hh + 4 (q & 1) + r: register e of lane half h then
//      holds hidden column 16 (e >> 3) + 8 h + (e & 7) - registers 8 k2 .. 8 k2 + 7, packed, ARE the B operand of fc2's k-step k2.
// fc2: A row i is output channel sg(i) of the 32-channel group, sg(8 q + 4 hh + r) = 16 hh + 4 q + r: register e of lane half h holds
//      channel 16 h + e.
//
// Schedule of one tile (26 barrier-separated steps k; W1(j) / W2(j) = the slab-j pieces of fc1.weight / fc2.weight):
//     step 0        fc1(0)                                   -> hA                      then out^T = b2
//     step s + 1    fc2(s - 1) | GELU(s) | fc1(s + 1)        s = 0 .. 23 (fc2 from s = 1, fc1 up to s = 22)
//     step 25       fc2(23)                                  then + resid and the output stores
// Inside a step the matrix work (fc2 of the PREVIOUS slab, fc1 of the NEXT slab) does not depend on the vector work (GELU of this
// slab), so the wave's own stream keeps the matrix pipe and the VALU busy together.  h and g alternate between two register sets
// (A: even slabs, B: odd slabs).  Step k reads ring stage k & 1 (W1(k), W2(k - 2)) and requests W1(k + 1), W2(k - 1) into the other
// stage right after its barrier; every wave issues exactly six DMAs per step (from a zero buffer where a piece does not exist), so the
// counted waits are the same for every wave.
// SAVE: GELU(h) is also written to hact [M][768] (the training form); RES: a residual operand exists
template <bool SAVE, bool RES>
__global__ __launch_bounds__(256) void mlp_fwd_kernel(const MlpArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, fh = lane >> 5;
  const uint32_t lbase = lds_addr(dsm);

  for (int i = tid; i < ML_H; i += 256) *(float*)(dsm + ML_B1 + 4 * i) = g.b1[i];
  for (int i = tid; i < ML_C; i += 256) *(float*)(dsm + ML_B2 + 4 * i) = g.b2[i];
  __syncthreads();

  const int ntiles = (int)((g.M + ML_BM - 1) / ML_BM);
  const int G = gridDim.x;
  const int lw = xcd_remap(blockIdx.x, G);
  const int nt_my = lw < ntiles ? (ntiles - lw + G - 1) / G : 0;
  if (nt_my == 0) return;
  const unsigned char* zero = (const unsigned char*)g_mlp_zero + ((lane & 7) << 4);

  // ---- DMA of a tile's xn rows: wave w, piece (kb, q) = rows 64 w + 8 q + (lane >> 3), 128 B of K-block kb per row; the LDS
  //      image is lane-linear, the swizzle (chunk ^ ((row >> 1) & 7)) sits on the SOURCE chunk
  auto issue_xn = [&](int tile) {
    const long m0 = (long)tile * ML_BM;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int row = 64 * wid + 8 * q + (lane >> 3);
      const int cs = (lane & 7) ^ (4 * (q & 1) + ((lane >> 4) & 3));
      const long m = m0 + row;
      const unsigned char* src = m < g.M ? (const unsigned char*)(g.xn + m * ML_C) + (cs << 4) : zero;
      const int kstep = m < g.M ? 128 : 0;
#pragma unroll
      for (int kb = 0; kb < 3; ++kb)
        __builtin_amdgcn_global_load_lds((glb_void*)(src + kb * kstep), (lds_void*)(dsm + ML_XN + kb * ML_XKB + (64 * wid + 8 * q) * 128), 16, 0, 0);
    }
  };
  // ---- the six DMAs of a step into ring stage st.  W1(s1): rows 8 w .. 8 w + 7 of the slab, one 1-KiB piece per K-block, source
  //      chunk ^ ((row >> 1) & 7);  W2(s2): three pieces of 16 rows x 64 B (rows 48 w + 16 q + (lane >> 2)), source chunk ^ ((row >> 2) & 3).
  //      A slab index < 0 = the piece does not exist (zeros)
  const int w1row = 8 * wid + (lane >> 3);
  const int w1cs = (lane & 7) ^ (((lane >> 4) & 3) | ((wid & 1) << 2));            // (row >> 1) & 7 = 4 (w & 1) + (lane >> 4)
  const unsigned char* w1src = (const unsigned char*)(g.w1 + (long)w1row * ML_C) + (w1cs << 4);
  const int w2row = 48 * wid + (lane >> 2);                                         // + 16 q
  auto issue_w = [&](int s1, int s2, int st) {
    {
      const unsigned char* p = s1 >= 0 ? w1src + (long)s1 * (ML_HS * ML_C * 2) : zero;
      const int kstep = s1 >= 0 ? 128 : 0;
#pragma unroll
      for (int kb = 0; kb < 3; ++kb)
        __builtin_amdgcn_global_load_lds((glb_void*)(p + kb * kstep), (lds_void*)(dsm + ML_WR + st * ML_WST + kb * (ML_HS * 128) + wid * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int row = w2row + 16 * q;
      const int cs = (lane & 3) ^ ((row >> 2) & 3);
      const unsigned char* p = s2 >= 0 ? (const unsigned char*)(g.w2 + (long)row * ML_H + s2 * ML_HS) + (cs << 4) : zero;
      __builtin_amdgcn_global_load_lds((glb_void*)p, (lds_void*)(dsm + ML_WR + st * ML_WST + ML_W1B + (48 * wid + 16 * q) * 64), 16, 0, 0);
    }
  };

  // ---- fragment read addresses (lane parts).  The chunk a lane reads is (2 kk + fh) ^ swizzle = (2 kk) ^ (fh ^ swizzle): one address
  //      register per kk (an XOR is not an immediate offset)
  const int pi_r = 16 * (fr >> 4) + 8 * ((fr >> 2) & 1) + 4 * ((fr >> 3) & 1) + (fr & 3);      // fc1: hidden column of A row fr
  const int sg_r = 16 * ((fr >> 2) & 1) + 4 * (fr >> 3) + (fr & 3);                            // fc2: output channel of A row fr
  uint32_t xa[4], xa2[4], w1a[4], w2a[2];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    xa[kk] = lbase + ML_XN + (64 * wid + fr) * 128 + ((((2 * kk) ^ fh) ^ ((fr >> 1) & 7)) << 4);
    xa2[kk] = xa[kk] + 32768;                                                      // (K-block 2: beyond the 16-bit offset field)
    w1a[kk] = lbase + ML_WR + pi_r * 128 + ((((2 * kk) ^ fh) ^ ((pi_r >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int k2 = 0; k2 < 2; ++k2) w2a[k2] = lbase + ML_WR + ML_W1B + sg_r * 64 + ((((2 * k2) ^ fh) ^ ((sg_r >> 2) & 3)) << 4);
  const uint32_t b1a = lbase + ML_B1 + 32 * fh;          // b1 of registers 0-7 at + 128 s, of registers 8-15 at + 128 s + 64
  const uint32_t b2a = lbase + ML_B2 + 64 * fh;          // b2 of the 16 registers of group p at + 128 p

  f32x16 out[6][2];                 // [32-channel group p][token tile u]: channel 32 p + 16 fh + e of token 64 w + 32 u + fr
  f32x16 hA[2], hB[2];              // h^T of the even / odd slabs: [token tile u], hidden column 32 s + 16 (e >> 3) + 8 fh + (e & 7)
  u32x4 gA[2][2], gB[2][2];         // GELU(h) of the even / odd slabs, packed: the B operands of fc2, [k-step k2][token tile u]

#ifdef ML_STAMPS
  unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  issue_xn(lw);
  issue_w(0, -1, 0);
  for (int ord = 0; ord < nt_my; ++ord) {
    const int tile = lw + ord * G;
    const long m0 = (long)tile * ML_BM;
    const bool has_next = ord + 1 < nt_my;
    const bool partial = m0 + ML_BM > g.M;                 // only the last tile: its stores may be skipped, so waits assume none
    const long tok0 = m0 + 64 * wid + fr;                   // token of u = 0; u = 1: + 32
    const bool ok0 = tok0 < g.M, ok1 = tok0 + 32 < g.M;
    const long tk0 = ok0 ? tok0 : g.M - 1, tk1 = ok1 ? tok0 + 32 : g.M - 1;          // clamped: loads stay in bounds
    // GELU(h) rows of this wave: a wave-uniform base (scalar registers) + one 32-bit lane offset
    unsigned char* hbase = SAVE ? (unsigned char*)(g.hact + (m0 + 64 * wid) * ML_H) : nullptr;
    const uint32_t hoff = (uint32_t)(fr * ML_H + 8 * fh) * 2;

    // One step's work.  FC2: out^T += W2(s - 1) . gin;  GELU: gout = GELU(hin) (slab s; stored when SAVE);  FC1: hout = W1(s + 1) . xn^T
    // + b1.  ST = the ring stage read (compile-time).  The work is a list of MFMA groups (two MFMAs sharing a weight fragment: the two
    // token tiles): fc2 groups (p, k2) read one fragment, fc1 groups kk = (K-block, 16-channel step) read three.  The reads of group
    // G + 2 are issued before the MFMAs of group G; the GELU of one accumulator pair (two in a 12-group step) follows each group.
    auto step = [&](auto fc2_c, auto gelu_c, auto fc1_c, auto st_c, const int s, f32x16 (&hin)[2], f32x16 (&hout)[2],
                    u32x4 (&gin)[2][2], u32x4 (&gout)[2][2]) {
      constexpr bool FC2 = decltype(fc2_c)::value, GELU = decltype(gelu_c)::value, FC1 = decltype(fc1_c)::value;
      constexpr int SO = decltype(st_c)::value * ML_WST;
      constexpr int N2 = FC2 ? 12 : 0, NG = N2 + (FC1 ? 12 : 0);
      u32x4 fw[3], fx0[3], fx1[3];            // rotating fragment sets
      u32x4 bb[4];                            // b1 of slab s + 1 (16 values per lane)
      // GELU of accumulator pair P (u = P >> 3, registers 2 (P & 7), + 1) -> dword (P & 3) of gout[(P & 7) >> 2][u]
      auto pair = [&](auto pc) {
        constexpr int P = decltype(pc)::value, U = P >> 3, E = 2 * (P & 7), K2 = (P & 7) >> 2, WD = P & 3;
#ifdef ML_ABL_NOGELU          // (timing ablations of tools/exp/ab_build.sh builds only)
        const uint32_t w = pack2bf(hin[U][E], hin[U][E + 1]);
#else
        const uint32_t w = pack2bf(gelu_sig(hin[U][E]), gelu_sig(hin[U][E + 1]));
#endif
        if constexpr (WD == 0) gout[K2][U].x = w; else if constexpr (WD == 1) gout[K2][U].y = w; else if constexpr (WD == 2) gout[K2][U].z = w; else gout[K2][U].w = w;
        if constexpr (SAVE && WD == 3) {
          if (U == 0 ? ok0 : ok1)
            *(uint4*)(hbase + (hoff + (uint32_t)(U * 32 * ML_H * 2 + s * ML_HS * 2 + K2 * 32))) =
                make_uint4(gout[K2][U].x, gout[K2][U].y, gout[K2][U].z, gout[K2][U].w);
        }
      };
      auto vslot = [&](auto ic) {
        constexpr int I = decltype(ic)::value;
        if constexpr (GELU) {
          if constexpr (NG == 24) { if constexpr (I < 16) pair(IntC<I>{}); }
          else if constexpr (I < 8) { pair(IntC<2 * I>{}); pair(IntC<2 * I + 1>{}); }
        }
      };
      // number of LDS reads of group I (0 beyond the list)
      auto nreads = [](int I) constexpr { return I >= NG ? 0 : (I < N2 ? 1 : (I == N2 ? 7 : 3)); };
      auto rd = [&](auto ic) {
        constexpr int I = decltype(ic)::value, R = I % 3;
        if constexpr (I < NG) {
          if constexpr (I < N2) {
            constexpr int P = I >> 1, K2 = I & 1;
            fw[R] = lds_rd128<SO + P * 2048>(w2a[K2]);
          } else {
            constexpr int KK = I - N2, KB = KK >> 2, KQ = KK & 3;
            if constexpr (KK == 0) {
              bb[0] = lds_rd128<0>(b1a + 128 * (s + 1)); bb[1] = lds_rd128<16>(b1a + 128 * (s + 1));
              bb[2] = lds_rd128<64>(b1a + 128 * (s + 1)); bb[3] = lds_rd128<80>(b1a + 128 * (s + 1));
            }
            fw[R] = lds_rd128<SO + KB * 4096>(w1a[KQ]);
            if constexpr (KB < 2) { fx0[R] = lds_rd128<KB * 32768>(xa[KQ]); fx1[R] = lds_rd128<KB * 32768 + 4096>(xa[KQ]); }
            else { fx0[R] = lds_rd128<32768>(xa2[KQ]); fx1[R] = lds_rd128<32768 + 4096>(xa2[KQ]); }
          }
        }
      };
      auto mm = [&](auto ic) {
        constexpr int I = decltype(ic)::value, R = I % 3;
        if constexpr (I < N2) {
          constexpr int P = I >> 1, K2 = I & 1;
          out[P][0] = mma32(fw[R], gin[K2][0], out[P][0]);
          out[P][1] = mma32(fw[R], gin[K2][1], out[P][1]);
        } else if constexpr (I == N2) {
          f32x16 z;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            z[4 * q + 0] = __uint_as_float(bb[q].x); z[4 * q + 1] = __uint_as_float(bb[q].y);
            z[4 * q + 2] = __uint_as_float(bb[q].z); z[4 * q + 3] = __uint_as_float(bb[q].w);
          }
          hout[0] = mma32(fw[R], fx0[R], z);
          hout[1] = mma32(fw[R], fx1[R], z);
        } else {
          hout[0] = mma32(fw[R], fx0[R], hout[0]);
          hout[1] = mma32(fw[R], fx1[R], hout[1]);
        }
      };
      rd(IntC<0>{});
      rd(IntC<1>{});
      auto run = [&](auto self, auto ic) -> void {
        constexpr int I = decltype(ic)::value;
        rd(IntC<I + 2>{});
        lgkm_wait<nreads(I + 1) + nreads(I + 2)>();
        mm(IntC<I>{});
        vslot(IntC<I>{});
        if constexpr (I + 1 < NG) self(self, IntC<I + 1>{});
      };
      run(run, IntC<0>{});
    };
    // top of a step: the DMAs requested one step ago have landed for every wave
#define ML_TOP() do { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
    constexpr BoolC<true> Y{};
    constexpr BoolC<false> N{};
    // in flight behind the DMAs a step waits for: this wave's four GELU(h) stores of the previous step (training form, complete tiles)
#define ML_WAIT_STEP() do { if (SAVE && !partial) ML_VM(4); else ML_VM(0); } while (0)

    // ---- step 0: fc1(0).  In flight behind the awaited DMAs: after the first tile the previous tile's 24 output stores (that tile
    //      was complete: only the last tile can be partial)
    if (ord == 0) ML_VM(0); else ML_VM(24);
    ML_TOP();
    issue_w(1, -1, 1);
    step(N, N, Y, IntC<0>{}, -1, hA, hA, gA, gA);
    // out^T = b2
#pragma unroll
    for (int p = 0; p < 6; ++p) {
      const u32x4 c0 = lds_rd128<0>(b2a + 128 * p), c1 = lds_rd128<16>(b2a + 128 * p);
      const u32x4 c2 = lds_rd128<32>(b2a + 128 * p), c3 = lds_rd128<48>(b2a + 128 * p);
      lgkm_wait<0>();
      const float bz[16] = {__uint_as_float(c0.x), __uint_as_float(c0.y), __uint_as_float(c0.z), __uint_as_float(c0.w),
                            __uint_as_float(c1.x), __uint_as_float(c1.y), __uint_as_float(c1.z), __uint_as_float(c1.w),
                            __uint_as_float(c2.x), __uint_as_float(c2.y), __uint_as_float(c2.z), __uint_as_float(c2.w),
                            __uint_as_float(c3.x), __uint_as_float(c3.y), __uint_as_float(c3.z), __uint_as_float(c3.w)};
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 16; ++e) out[p][u][e] = bz[e];
    }
    // ---- step 1 (slab 0): GELU(0) | fc1(1)
    ML_VM(0);
    ML_TOP();
    issue_w(2, 0, 0);
    step(N, Y, Y, IntC<1>{}, 0, hA, hB, gA, gA);
    // ---- steps 2 .. 23 (slabs 1 .. 22), two per iteration: the register sets and the ring stage alternate
#pragma unroll 1
    for (int s = 1; s < ML_NSLAB - 2; s += 2) {
      ML_T(t0);
      ML_WAIT_STEP();
      ML_T(t1);
      ML_TOP();
      ML_T(t2);
      issue_w(s + 2, s, 1);
      ML_T(t3);
      step(Y, Y, Y, IntC<0>{}, s, hB, hA, gA, gB);
      ML_T(t4);
      ML_ACC(0, t0, t1); ML_ACC(1, t1, t2); ML_ACC(2, t2, t3); ML_ACC(3, t3, t4);
      ML_WAIT_STEP();
      ML_T(t5);
      ML_TOP();
      ML_T(t6);
      issue_w(s + 3 < ML_NSLAB ? s + 3 : -1, s + 1, 0);
      ML_T(t7);
      step(Y, Y, Y, IntC<1>{}, s + 1, hA, hB, gB, gA);
      ML_T(t8);
      ML_ACC(0, t4, t5); ML_ACC(1, t5, t6); ML_ACC(2, t6, t7); ML_ACC(3, t7, t8);
    }
    // the wave's xn rows are dead after fc1(23): request the next tile's
    if (has_next) issue_xn(tile + G);
    // ---- step 24 (slab 23): fc2(22) | GELU(23)
    if (has_next) { if (SAVE) ML_VM(28); else ML_VM(24); }
    else ML_WAIT_STEP();
    ML_TOP();
    issue_w(-1, ML_NSLAB - 1, 1);
    // residual rows of the tile (plain loads, clamped rows - never skipped): they land under the last two steps, when the h / fragment
    // registers of the fc1 part are free; hipcc waits for them where the epilogue adds them
    uint4 pre[6][2][2];
    if (RES) {
#pragma unroll
      for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          pre[p][0][c] = *(const uint4*)(g.resid + tk0 * ML_C + 32 * p + 16 * fh + 8 * c);
          pre[p][1][c] = *(const uint4*)(g.resid + tk1 * ML_C + 32 * p + 16 * fh + 8 * c);
        }
    }
    step(Y, Y, N, IntC<0>{}, ML_NSLAB - 1, hB, hB, gA, gB);
    // ---- step 25: fc2(23).  Behind the DMAs of step 24: the 24 residual loads (and the four GELU(h) stores)
    if (RES) { if (SAVE && !partial) ML_VM(28); else ML_VM(24); }
    else ML_WAIT_STEP();
    ML_TOP();
    issue_w(has_next ? 0 : -1, -1, 0);
    step(Y, N, N, IntC<1>{}, ML_NSLAB, hA, hA, gB, gB);
#undef ML_TOP
#undef ML_WAIT_STEP
    // ---- the tile's output rows, from registers: 16 consecutive channels per lane and group
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = out[p][u][e];
        if (RES) {
          float x[16];
          unpack<bf16>(pre[p][u][0], x); unpack<bf16>(pre[p][u][1], x + 8);
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] += x[e];
        }
        const long tok = tok0 + 32 * u;
        if (tok < g.M) {
          *(uint4*)(g.out + tok * ML_C + 32 * p + 16 * fh) = pack<bf16>(v);
          *(uint4*)(g.out + tok * ML_C + 32 * p + 16 * fh + 8) = pack<bf16>(v + 8);
        }
      }
  }
  ML_VM(0);
#ifdef ML_STAMPS
  if (lane == 0)
    for (int i = 0; i < 8; ++i) g_mlp_stamps[(blockIdx.x * 8 + wid) * 8 + i] = st_sum[i];
#endif
}

template <bool SAVE, bool RES> int launch_mlp(const MlpArgs& a, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)mlp_fwd_kernel<SAVE, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, ML_LDS) != hipSuccess) {
      (void)hipGetLastError();
      return SODT_EINVAL;
    }
    attr_set = true;
  }
  const long ntiles = (a.M + ML_BM - 1) / ML_BM;
  const int grid = (int)(ntiles < 256 ? ntiles : 256);
  hipLaunchKernelGGL((mlp_fwd_kernel<SAVE, RES>), dim3(grid), dim3(256), ML_LDS, st, a);
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

}  // namespace

extern "C" int sodt_mlp_fwd(const void* xn, const void* w1, const float* b1, const void* w2, const float* b2, const void* resid,
                            void* out, void* hact, long M, int C, int dtype, sodt_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  if (!xn || !w1 || !b1 || !w2 || !b2 || !out || M <= 0 || C <= 0 || (C % 8) != 0) return SODT_EINVAL;
  if (dtype == SODT_F32 || C != ML_C) {
    // the parity path (and widths the fused kernel is not built for): the two-GEMM chain through the caller's hidden buffer
    if (!hact) return SODT_EINVAL;
    const int H4 = 4 * C;
    sodt_gemm_args g1 = {};
    g1.a.nseg = 1; g1.a.spatial = 0; g1.a.Ho = 1; g1.a.Wo = 1;
    g1.a.s[0].p = xn; g1.a.s[0].ld = C; g1.a.s[0].klen = C; g1.a.s[0].mul = 1;
    g1.W = w1; g1.ldw = C; g1.C = hact; g1.ldc = H4; g1.bias = b1;
    g1.M = (int)M; g1.N = H4; g1.K = C; g1.flags = SODT_EPI_BIAS | SODT_EPI_GELU;
    int rc = sodt_gemm_nt(&g1, dtype, st_);
    if (rc != SODT_OK) return rc;
    sodt_gemm_args g2 = {};
    g2.a.nseg = 1; g2.a.spatial = 0; g2.a.Ho = 1; g2.a.Wo = 1;
    g2.a.s[0].p = hact; g2.a.s[0].ld = H4; g2.a.s[0].klen = H4; g2.a.s[0].mul = 1;
    g2.W = w2; g2.ldw = H4; g2.C = out; g2.ldc = C; g2.bias = b2;
    g2.R = resid; g2.ldr = C;
    g2.M = (int)M; g2.N = C; g2.K = H4; g2.flags = SODT_EPI_BIAS | (resid ? SODT_EPI_RESID : 0);
    return sodt_gemm_nt(&g2, dtype, st_);
  }
  if (dtype != SODT_BF16) return SODT_EINVAL;
  if ((((uintptr_t)xn | (uintptr_t)w1 | (uintptr_t)w2 | (uintptr_t)out | (uintptr_t)resid | (uintptr_t)hact) & 15) ||
      (((uintptr_t)b1 | (uintptr_t)b2) & 3))
    return SODT_EINVAL;
  MlpArgs a;
  a.xn = (const bf16*)xn; a.w1 = (const bf16*)w1; a.b1 = b1; a.w2 = (const bf16*)w2; a.b2 = b2; a.resid = (const bf16*)resid;
  a.out = (bf16*)out; a.hact = (bf16*)hact; a.M = M;
  if (hact) return resid ? launch_mlp<true, true>(a, st) : launch_mlp<true, false>(a, st);
  return resid ? launch_mlp<false, true>(a, st) : launch_mlp<false, false>(a, st);
}

#ifdef ML_STAMPS
extern "C" int sodt_debug_mlp_stamps(unsigned long long* host_256x8x8) {
  return hipMemcpyFromSymbol(host_256x8x8, HIP_SYMBOL(g_mlp_stamps), sizeof(unsigned long long) * 256 * 8 * 8) == hipSuccess ? SODT_OK : SODT_EINVAL;
}
#endif
