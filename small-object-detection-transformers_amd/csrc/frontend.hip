// Front end of ImageEncoderViT.forward (backbone_vit.py:195-210), fused:
//   get_channels                      (:810-820)   R,G,B planes of x_rgb + IR plane
//   4 x Conv2d(1->48, k4, s4) + bias  (:69-98)     R with padding 1, G/B/IR padding 0
//   CAttentionBlock                   (:469-561)   pairs (R<-G),(G<-B),(B<-IR),(IR<-G),
//                                                  12 heads x 4, softmax(q k^T / 2) v,
//                                                  residual, LayerNorm(48)
//   torch.cat(dim=-1)                 (:210)       -> [B*t*t][192] token-major
// With the shipped window_size = 1 (:438) every window holds one token, softmax over
// one key is exactly 1, and the block reduces to x_q = LN_q(e_q + e_kv): that is the
// ca_ws == 1 path below.  HBM-bound: reads 4*S*S f32, writes t*t*192 activations.
//
// One workgroup = 64 consecutive tokens; wave p computes plane p's 48-channel embedding
// for its token (weights are wave-uniform -> scalar loads), the four embeddings meet in
// LDS, wave q then forms pair q's sum + LayerNorm, and the 64x192 output block (contiguous
// in memory) is written with 16-byte stores.  Backward recomputes the embeddings from
// the image instead of saving them and reduces the parameter gradients over a
// grid-stride loop before a final burst of atomics.
#include "common.h"
#include "../../include/sodt_hip.h"

namespace {

constexpr int CE = 48;      // channels per plane embedding
constexpr int EP = CE + 1;  // padded LDS row (floats)
constexpr int FE_PART = 4 * CE * 19;   // partial sums of one backward workgroup: per plane/pair dW[48][16] | dgamma | dbeta | sum d(pair sum)
constexpr int FE_BWD_GRID = 512;       // two resident workgroups per CU (66 KB of LDS each)

struct FeGeo { int B, S, t; long ir_bstride; };

// 4x4 patch of plane p for token (b, y, x): R (p == 0) is shifted by -1 with zero padding
__device__ __forceinline__ void load_patch(const float* __restrict__ rgb, const float* __restrict__ ir, const FeGeo& g,
                                           int p, int b, int y, int x, float* pt) {
  const int S = g.S;
  if (p == 0) {
    const float* base = rgb + (long)b * 3 * S * S;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int yy = 4 * y - 1 + i;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int xx = 4 * x - 1 + j;
        pt[i * 4 + j] = (yy >= 0 && xx >= 0 && yy < S && xx < S) ? base[(long)yy * S + xx] : 0.f;
      }
    }
  } else {
    const float* base = (p == 3) ? (ir + (long)b * g.ir_bstride) : (rgb + ((long)b * 3 + p) * S * S);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 v = *(const float4*)(base + (long)(4 * y + i) * S + 4 * x);
      pt[i * 4] = v.x; pt[i * 4 + 1] = v.y; pt[i * 4 + 2] = v.z; pt[i * 4 + 3] = v.w;
    }
  }
}

// pair q: query plane / key-value plane  (backbone_vit.py:508-521)
__device__ __forceinline__ int pair_q(int q) { return q; }
__device__ __forceinline__ int pair_kv(int q) { return q == 0 ? 1 : (q == 1 ? 2 : (q == 2 ? 3 : 1)); }

template <typename T>
__global__ __launch_bounds__(256) void frontend_fwd_kernel(const float* __restrict__ rgb, const float* __restrict__ ir,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          T* __restrict__ out, const FeGeo g, long ntok) {
  __shared__ float sE[4][64][EP];
  __shared__ __attribute__((aligned(16))) T sO[64 * 192];
  float (*sP)[64][17] = (float (*)[64][17])sO;     // patches [4][64][17]: dead before the first write of the output tile
  static_assert(4 * 64 * 17 * 4 <= 64 * 192 * sizeof(T), "patch tile aliases the output tile");
  const int tid = threadIdx.x, lane = tid & 63;
  const int p = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t16 = lane & 15, g4 = lane >> 4;
  const long tok0 = (long)blockIdx.x * 64;
  const long tok = tok0 + lane;
  const bool live = tok < ntok;
  const int tt = g.t * g.t;
  int b = 0, y = 0, x = 0;
  if (live) { b = (int)(tok / tt); const int rem = (int)(tok - (long)b * tt); y = rem / g.t; x = rem - y * g.t; }
  float pt[16];
  if (live) load_patch(rgb, ir, g, p, b, y, x, pt);
  else {
#pragma unroll
    for (int i = 0; i < 16; ++i) pt[i] = 0.f;
  }
  const float* wp = w + p * CE * 16;
  const float* bp = bias + p * CE;
  // E^T[ch][tok] = W[ch][tap] P^T[tap][tok] + bias as 3 x 4 tiles of exact-f32 MFMA 16x16x4 (the scalar-weight FMA loop
  // waited for one s_load round trip per output channel)
#pragma unroll
  for (int k = 0; k < 16; ++k) sP[p][lane][k] = pt[k];
  wave_sync();
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    float pb[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) pb[kk] = sP[p][16 * n + t16][4 * kk + g4];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      f32x4 e;
#pragma unroll
      for (int r = 0; r < 4; ++r) e[r] = bp[16 * s + 4 * g4 + r];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) e = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[(16 * s + t16) * 16 + 4 * kk + g4], pb[kk], e, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) sE[p][16 * n + t16][16 * s + 4 * g4 + r] = e[r];
    }
  }
  __syncthreads();
  // pair p: LN(e_q + e_kv)
  const int qa = pair_q(p), qb = pair_kv(p);
  float s[CE];
  float mean = 0.f;
#pragma unroll
  for (int j = 0; j < CE; ++j) { s[j] = sE[qa][lane][j] + sE[qb][lane][j]; mean += s[j]; }
  mean *= (1.0f / CE);
  float var = 0.f;
#pragma unroll
  for (int j = 0; j < CE; ++j) { const float d = s[j] - mean; var += d * d; }
  const float rstd = rsqrtf(var * (1.0f / CE) + 1e-5f);
  const float* gp = gamma + p * CE;
  const float* bt = beta + p * CE;
#pragma unroll
  for (int j = 0; j < CE; ++j) sO[lane * 192 + p * CE + j] = from_f<T>((s[j] - mean) * rstd * gp[j] + bt[j]);
  __syncthreads();
  constexpr int KPL = TT<T>::KPL;
  const long nvalid = (ntok - tok0) < 64 ? (ntok - tok0) : 64;
  for (int idx = tid; idx < 64 * 192 / KPL; idx += 256) {
    if ((long)idx * KPL < nvalid * 192) *(uint4*)(out + tok0 * 192 + (long)idx * KPL) = *(const uint4*)(sO + idx * KPL);
  }
}

// element e of a 16-byte chunk as f32
template <typename T> __device__ __forceinline__ float chunk_elem(const uint4& u, int e);
template <> __device__ __forceinline__ float chunk_elem<float>(const uint4& u, int e) {
  return __uint_as_float(e == 0 ? u.x : (e == 1 ? u.y : (e == 2 ? u.z : u.w)));
}
template <> __device__ __forceinline__ float chunk_elem<bf16>(const uint4& u, int e) {
  const uint32_t wd = (e >> 1) == 0 ? u.x : ((e >> 1) == 1 ? u.y : ((e >> 1) == 2 ? u.z : u.w));
  return __uint_as_float((e & 1) ? (wd & 0xffff0000u) : (wd << 16));
}

// v[m] = this lane's value for channel class m (m = 0..3).  Returns, in the lanes of row r (lanes 16r .. 16r + 15), the
// sum of v[r] over the lane's column of four rows: v_permlane32_swap pairs classes (0,2) and (1,3), v_permlane16_swap
// then pairs the two results -- three swaps and three adds instead of four accumulators.
__device__ __forceinline__ float fold4(const float* v) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[0]), __float_as_uint(v[2]), false, false);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[1]), __float_as_uint(v[3]), false, false);
  const float sa = __uint_as_float(a[0]) + __uint_as_float(a[1]);   // rows 0,1: class 0 ; rows 2,3: class 2
  const float sb = __uint_as_float(b[0]) + __uint_as_float(b[1]);   // rows 0,1: class 1 ; rows 2,3: class 3
  auto c = __builtin_amdgcn_permlane16_swap(__float_as_uint(sa), __float_as_uint(sb), false, false);
  return __uint_as_float(c[0]) + __uint_as_float(c[1]);             // row r: class r
}

// dW tiles of one plane over a 64-token block: A[ch][tok] = sum of NQ pair-sum gradients ([64][EP] f32 in LDS),
// B = patch taps
template <int NQ>
__device__ __forceinline__ void dw_tiles(const float* d0, const float* d1, const float* d2, const float* pp, int lane,
                                         f32x4* adw) {
  const int t16 = lane & 15, g4 = lane >> 4;
#pragma unroll 4
  for (int kk = 0; kk < 16; ++kk) {
    const int tk = 4 * kk + g4;
    const float bv = pp[tk * 17 + t16];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      float av = d0[tk * EP + 16 * s + t16];
      if constexpr (NQ > 1) av += d1[tk * EP + 16 * s + t16];
      if constexpr (NQ > 2) av += d2[tk * EP + 16 * s + t16];
      adw[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, adw[s], 0, 0, 0);
    }
  }
}

// Backward.  Per 64-token block wave p (lane = token) recomputes plane p's embedding, the four meet in LDS, wave p does
// pair p's LayerNorm backward and overwrites its plane's LDS slot with d(pair sum).  Reductions over tokens never
// leave the register file until the end of the kernel:
//   dgamma, dbeta, db        channels j, j+12, j+24, j+36 fold into one register through permlane swaps (row r of the
//                            wave keeps channel j + 12 r): 12 + 12 running sums per lane, reduced across the 16 lanes
//                            of a row once at the very end;
//   dW[48][16] of plane p    = sum_tok dE[tok][:]^T patch[tok][:]  -> 3 x 16 exact-f32 MFMA 16x16x4 per block, operands
//                            read straight from LDS (A = sum of the pair gradients feeding plane p).
// 66 KB of LDS and <= 256 registers: two workgroups per CU.  dout rows are read as 16-byte chunks (lane = token row),
// the next block's patch is loaded under the MFMA phase, and the embeddings themselves are MFMA tiles too (conv weights
// loop-invariant in registers as the A operand).
template <typename T, bool WS>
__global__ __launch_bounds__(256, 2) void frontend_bwd_kernel(const float* __restrict__ rgb, const float* __restrict__ ir,
                                                             const float* __restrict__ w, const float* __restrict__ bias,
                                                             const float* __restrict__ gamma, const T* __restrict__ dout,
                                                             float* __restrict__ dw, float* __restrict__ db,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             float* __restrict__ ws, const FeGeo g, long ntok, int nblk) {
  __shared__ float sE[4][64][EP];    // embeddings, then d(pair sum) of pair p in slot p
  __shared__ float sP[4][64][17];    // patches
  const int tid = threadIdx.x, lane = tid & 63;
  const int p = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tt = g.t * g.t;
  const float* wp = w + p * CE * 16;
  const float* bp = bias + p * CE;
  const float* gp = gamma + p * CE;
  const int t16 = lane & 15, g4 = lane >> 4;
  constexpr int KPL = TT<T>::KPL, SPR = CE / KPL;   // 16-byte chunks per dout row slice

  f32x4 adw[3];                      // dW tile s: rows 16s + 4*g4 + r, column (tap) t16
#pragma unroll
  for (int s = 0; s < 3; ++s) adw[s] = f32x4{0.f, 0.f, 0.f, 0.f};
  float adg[12], adbt[12], ads[12];  // dgamma / dbeta / sum of d(pair sum) of channel j + 12 * g4, this lane's share of the tokens (fold4)
#pragma unroll
  for (int j = 0; j < 12; ++j) { adg[j] = 0.f; adbt[j] = 0.f; ads[j] = 0.f; }

  auto fetch = [&](int blk, float* pt) {
    const long tok = (long)blk * 64 + lane;
    if (blk < nblk && tok < ntok) {
      const int b = (int)(tok / tt);
      const int rem = (int)(tok - (long)b * tt);
      const int y = rem / g.t, x = rem - y * g.t;
      load_patch(rgb, ir, g, p, b, y, x, pt);
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) pt[i] = 0.f;
    }
  };
  float pt[16];
  fetch(blockIdx.x, pt);
  float wa[3][4];                    // A operand of the embedding MFMAs: W[16s + t16][4kk + g4]
  f32x4 eb[3];                       // bias of rows 16s + 4*g4 + r
#pragma unroll
  for (int s = 0; s < 3; ++s) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) wa[s][kk] = wp[(16 * s + t16) * 16 + 4 * kk + g4];
#pragma unroll
    for (int r = 0; r < 4; ++r) eb[s][r] = bp[16 * s + 4 * g4 + r];
  }

  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const long tok = (long)blk * 64 + lane;
    const bool live = tok < ntok;
    uint4 raw[SPR];                  // this token's dout slice for pair p, in flight under the embedding FMAs
#pragma unroll
    for (int c = 0; c < SPR; ++c)
      raw[c] = live ? *(const uint4*)(dout + tok * 192 + p * CE + c * KPL) : make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();   // previous iteration's readers are done
#pragma unroll
    for (int k = 0; k < 16; ++k) sP[p][lane][k] = pt[k];
    wave_sync();                 // sP[p] is written and read by this wave only
    // E^T[ch][tok] = W[ch][tap] P^T[tap][tok] + bias: 3 x 4 tiles of exact-f32 MFMA 16x16x4, W and bias loop-invariant
    // in registers (the VALU stays free for the other workgroup's LayerNorm phase)
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      float pb[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) pb[kk] = sP[p][16 * n + t16][4 * kk + g4];
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        f32x4 e = eb[s];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) e = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s][kk], pb[kk], e, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) sE[p][16 * n + t16][16 * s + 4 * g4 + r] = e[r];
      }
    }
    __syncthreads();
    // LayerNorm backward of pair p for this token
    const int qa = pair_q(p), qb = pair_kv(p);
    float xh[CE];
    float mean = 0.f;
#pragma unroll
    for (int j0 = 0; j0 < CE; j0 += 12) {   // (12 channels at a time: caps the registers holding LDS results in flight)
#pragma unroll
      for (int j = j0; j < j0 + 12; ++j) { xh[j] = sE[qa][lane][j] + sE[qb][lane][j]; mean += xh[j]; }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();   // every embedding has been read: slot p may take d(pair sum) now
    mean *= (1.0f / CE);
    float var = 0.f;
#pragma unroll
    for (int j = 0; j < CE; ++j) { xh[j] -= mean; var += xh[j] * xh[j]; }
    const float rstd = rsqrtf(var * (1.0f / CE) + 1e-5f);
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int j = 0; j < 12; ++j) {   // channels j, j + 12, j + 24, j + 36 fold into one register per sum
      float u[4], d[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int jj = j + 12 * m;
        d[m] = chunk_elem<T>(raw[jj / KPL], jj % KPL);
        xh[jj] *= rstd;
        const float gg = d[m] * gp[jj];
        c1 += gg; c2 += gg * xh[jj];
        u[m] = d[m] * xh[jj];
      }
      adg[j] += fold4(u);
      adbt[j] += fold4(d);
      __builtin_amdgcn_sched_barrier(0);
    }
    c1 *= (1.0f / CE); c2 *= (1.0f / CE);
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      float d[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int jj = j + 12 * m;
        d[m] = rstd * (chunk_elem<T>(raw[jj / KPL], jj % KPL) * gp[jj] - c1 - xh[jj] * c2);
        sE[p][lane][jj] = d[m];
      }
      ads[j] += fold4(d);
    }
    fetch(blk + gridDim.x, pt);      // next block's patch: in flight under the MFMA phase
    __syncthreads();
    // dW of plane p: A[ch][tok] = sum of the pair-sum gradients plane p feeds
    //   R: pair0 ; G: pair0 (kv) + pair1 (q) + pair3 (kv) ; B: pair1 (kv) + pair2 (q) ; IR: pair2 (kv) + pair3 (q)
    if (p == 0) dw_tiles<1>(&sE[0][0][0], nullptr, nullptr, &sP[0][0][0], lane, adw);
    else if (p == 1) dw_tiles<3>(&sE[0][0][0], &sE[1][0][0], &sE[3][0][0], &sP[1][0][0], lane, adw);
    else if (p == 2) dw_tiles<2>(&sE[1][0][0], &sE[2][0][0], nullptr, &sP[2][0][0], lane, adw);
    else dw_tiles<2>(&sE[2][0][0], &sE[3][0][0], nullptr, &sP[3][0][0], lane, adw);
  }
  // WS: this workgroup's partial sums go to its row of the workspace (plain stores; frontend_bwd_reduce_kernel finishes).
  // Without a workspace every wave adds into the gradients directly: ~2M same-line device-scope atomics, which cost
  // about as much as the whole loop above at the bench shape.
  float* wsp = WS ? ws + (long)blockIdx.x * FE_PART + p * (FE_PART / 4) : nullptr;
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = (16 * s + 4 * g4 + r) * 16 + t16;
      if constexpr (WS) wsp[o] = adw[s][r];
      else atomicAdd(dw + p * CE * 16 + o, adw[s][r]);
    }
  // one reduction over the 16 lanes of each row per wave
#pragma unroll
  for (int j = 0; j < 12; ++j) {
    float a = adg[j], c = adbt[j], d = ads[j];
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) { a += __shfl_xor(a, m); c += __shfl_xor(c, m); d += __shfl_xor(d, m); }
    if (t16 == 0) {
      const int ch = j + 12 * g4;
      if constexpr (WS) {
        wsp[CE * 16 + ch] = a; wsp[CE * 17 + ch] = c; wsp[CE * 18 + ch] = d;
      } else {
        atomicAdd(dgamma + p * CE + ch, a);
        atomicAdd(dbeta + p * CE + ch, c);
        atomicAdd(db + pair_q(p) * CE + ch, d);    // pair p's sum gradient reaches its query plane's bias ...
        atomicAdd(db + pair_kv(p) * CE + ch, d);   // ... and its key/value plane's
      }
    }
  }
}

// second stage of the workspace path: column sums of the [nwg][FE_PART] partials, RG rows per workgroup, then one atomic
// per output element and row group
constexpr int FE_RG = 32;
__global__ __launch_bounds__(64) void frontend_bwd_reduce_kernel(const float* __restrict__ ws, int nwg, float* __restrict__ dw,
                                                                float* __restrict__ db, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta) {
  const int col = blockIdx.x * 64 + threadIdx.x;
  const int r0 = blockIdx.y * FE_RG, r1 = (r0 + FE_RG) < nwg ? (r0 + FE_RG) : nwg;
  float a = 0.f;
  for (int r = r0; r < r1; ++r) a += ws[(long)r * FE_PART + col];
  const int p = col / (FE_PART / 4), o = col - p * (FE_PART / 4);
  if (o < CE * 16) atomicAdd(dw + p * CE * 16 + o, a);
  else if (o < CE * 17) atomicAdd(dgamma + p * CE + o - CE * 16, a);
  else if (o < CE * 18) atomicAdd(dbeta + p * CE + o - CE * 17, a);
  else {
    atomicAdd(db + pair_q(p) * CE + o - CE * 18, a);
    atomicAdd(db + pair_kv(p) * CE + o - CE * 18, a);
  }
}

}  // namespace

extern "C" int sodt_frontend_fwd(const float* rgb, const float* ir, long ir_bstride, const float* w, const float* b,
                                 const float* gamma, const float* beta, void* out, int B, int S, int ca_ws,
                                 int dtype, sodt_stream_t st) {
  if (!rgb || !ir || !w || !b || !gamma || !beta || !out || B <= 0 || S <= 0 || (S % 4)) return SODT_EINVAL;
  if (ca_ws != 1) return SODT_EINVAL;   // general window path: sodt_cross_channel_attn_* (cattn.hip)
  FeGeo g{B, S, S / 4, ir_bstride};
  const long ntok = (long)B * g.t * g.t;
  const unsigned blocks = (unsigned)((ntok + 63) / 64);
  if (dtype == SODT_BF16)
    hipLaunchKernelGGL(frontend_fwd_kernel<bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)st, rgb, ir, w, b, gamma, beta, (bf16*)out, g, ntok);
  else if (dtype == SODT_F32)
    hipLaunchKernelGGL(frontend_fwd_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)st, rgb, ir, w, b, gamma, beta, (float*)out, g, ntok);
  else return SODT_EINVAL;
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

extern "C" long sodt_frontend_bwd_workspace_bytes(int B, int S) {
  (void)B; (void)S;
  return (long)FE_BWD_GRID * FE_PART * (long)sizeof(float);
}

extern "C" int sodt_frontend_bwd(const float* rgb, const float* ir, long ir_bstride, const float* w, const float* b,
                                 const float* gamma, const float* beta, const void* dout,
                                 float* dw, float* db, float* dgamma, float* dbeta, int B, int S, int ca_ws,
                                 float* ws, long ws_bytes, int dtype, sodt_stream_t st) {
  (void)beta;
  if (!rgb || !ir || !w || !b || !gamma || !dout || !dw || !db || !dgamma || !dbeta || B <= 0 || S <= 0 || (S % 4)) return SODT_EINVAL;
  if (ca_ws != 1) return SODT_EINVAL;
  if (dtype != SODT_BF16 && dtype != SODT_F32) return SODT_EINVAL;
  FeGeo g{B, S, S / 4, ir_bstride};
  const long ntok = (long)B * g.t * g.t;
  const int nblk = (int)((ntok + 63) / 64);
  // the two-stage reduction pays once there are enough workgroups for the direct atomics to queue up; without a
  // workspace fewer, longer-lived workgroups keep the final burst of atomics short
  const bool two_stage = ws && ws_bytes >= sodt_frontend_bwd_workspace_bytes(B, S) && nblk > 64;
  const int cap = two_stage ? FE_BWD_GRID : FE_BWD_GRID / 2;
  const unsigned blocks = (unsigned)(nblk < cap ? nblk : cap);
  hipStream_t s = (hipStream_t)st;
#define FE_BWD_LAUNCH(T_, WS_) hipLaunchKernelGGL((frontend_bwd_kernel<T_, WS_>), dim3(blocks), dim3(256), 0, s, rgb, ir, w, b, gamma, \
                                                  (const T_*)dout, dw, db, dgamma, dbeta, ws, g, ntok, nblk)
  if (dtype == SODT_BF16) { if (two_stage) FE_BWD_LAUNCH(bf16, true); else FE_BWD_LAUNCH(bf16, false); }
  else { if (two_stage) FE_BWD_LAUNCH(float, true); else FE_BWD_LAUNCH(float, false); }
#undef FE_BWD_LAUNCH
  if (two_stage)
    hipLaunchKernelGGL(frontend_bwd_reduce_kernel, dim3(FE_PART / 64, (blocks + FE_RG - 1) / FE_RG), dim3(64), 0, s, ws, (int)blocks,
                       dw, db, dgamma, dbeta);
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}
