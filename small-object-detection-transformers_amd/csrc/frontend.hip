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
  const int tid = threadIdx.x, lane = tid & 63;
  const int p = __builtin_amdgcn_readfirstlane(tid >> 6);
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
  for (int j = 0; j < CE; ++j) {
    float a = bp[j];
#pragma unroll
    for (int k = 0; k < 16; ++k) a = fmaf(wp[j * 16 + k], pt[k], a);
    sE[p][lane][j] = a;
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

template <typename T>
__global__ __launch_bounds__(256) void frontend_bwd_kernel(const float* __restrict__ rgb, const float* __restrict__ ir,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          const float* __restrict__ gamma, const T* __restrict__ dout,
                                                          float* __restrict__ dw, float* __restrict__ db,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                          const FeGeo g, long ntok, int nblk) {
  __shared__ float sE[4][64][EP];    // embeddings, later d(embedding)
  __shared__ float sDS[4][64][EP];   // d(pair sum)
  __shared__ float sP[4][64][17];    // patches
  const int tid = threadIdx.x, lane = tid & 63;
  const int p = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tt = g.t * g.t;
  const float* wp = w + p * CE * 16;
  const float* bp = bias + p * CE;
  const float* gp = gamma + p * CE;
  // persistent partial sums
  const int tap = lane & 15, jg = lane >> 4;        // dW: this lane owns (j = jg*12 .. +11, tap)
  float adw[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) adw[i] = 0.f;
  float adb = 0.f;                                   // db: lane j < 48 owns bias j of plane p
  float adg = 0.f, adbt = 0.f;                       // dgamma/dbeta: lane j < 48 owns channel j of pair p

  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const long tok = (long)blk * 64 + lane;
    const bool live = tok < ntok;
    int b = 0, y = 0, x = 0;
    if (live) { b = (int)(tok / tt); const int rem = (int)(tok - (long)b * tt); y = rem / g.t; x = rem - y * g.t; }
    float pt[16];
    if (live) load_patch(rgb, ir, g, p, b, y, x, pt);
    else {
#pragma unroll
      for (int i = 0; i < 16; ++i) pt[i] = 0.f;
    }
    __syncthreads();   // previous iteration's readers are done
#pragma unroll
    for (int k = 0; k < 16; ++k) sP[p][lane][k] = pt[k];
    for (int j = 0; j < CE; ++j) {
      float a = bp[j];
#pragma unroll
      for (int k = 0; k < 16; ++k) a = fmaf(wp[j * 16 + k], pt[k], a);
      sE[p][lane][j] = a;
    }
    __syncthreads();
    // LayerNorm backward of pair p for this token
    {
      const int qa = pair_q(p), qb = pair_kv(p);
      float xh[CE];
      float mean = 0.f;
#pragma unroll
      for (int j = 0; j < CE; ++j) { xh[j] = sE[qa][lane][j] + sE[qb][lane][j]; mean += xh[j]; }
      mean *= (1.0f / CE);
      float var = 0.f;
#pragma unroll
      for (int j = 0; j < CE; ++j) { xh[j] -= mean; var += xh[j] * xh[j]; }
      const float rstd = rsqrtf(var * (1.0f / CE) + 1e-5f);
      float c1 = 0.f, c2 = 0.f;
      float dy[CE];
#pragma unroll
      for (int j = 0; j < CE; ++j) {
        dy[j] = live ? to_f(dout[tok * 192 + p * CE + j]) : 0.f;
        xh[j] *= rstd;
        const float gg = dy[j] * gp[j];
        c1 += gg; c2 += gg * xh[j];
      }
      c1 *= (1.0f / CE); c2 *= (1.0f / CE);
#pragma unroll
      for (int j = 0; j < CE; ++j) sDS[p][lane][j] = rstd * (dy[j] * gp[j] - c1 - xh[j] * c2);
      // dgamma / dbeta: reduce over the 64 tokens of this wave, lane j keeps channel j
#pragma unroll
      for (int j = 0; j < CE; ++j) {
        const float a = wave_sum(dy[j] * xh[j]);
        const float c = wave_sum(dy[j]);
        if (lane == j) { adg += a; adbt += c; }
      }
    }
    __syncthreads();
    // d(embedding of plane p) = sum of the pair-sum gradients it feeds:
    //   R: pair0 ; G: pair0 (kv) + pair1 (q) + pair3 (kv) ; B: pair1 (kv) + pair2 (q) ; IR: pair2 (kv) + pair3 (q)
    for (int j = 0; j < CE; ++j) {
      float d;
      if (p == 0) d = sDS[0][lane][j];
      else if (p == 1) d = sDS[0][lane][j] + sDS[1][lane][j] + sDS[3][lane][j];
      else if (p == 2) d = sDS[1][lane][j] + sDS[2][lane][j];
      else d = sDS[2][lane][j] + sDS[3][lane][j];
      sE[p][lane][j] = d;   // sE[p] is only read by its own wave from here on
    }
    __syncthreads();
    for (int tkn = 0; tkn < 64; ++tkn) {
      const float pv = sP[p][tkn][tap];
#pragma unroll
      for (int i = 0; i < 12; ++i) adw[i] = fmaf(sE[p][tkn][jg * 12 + i], pv, adw[i]);
    }
    if (lane < CE) {
      float a = 0.f;
      for (int tkn = 0; tkn < 64; ++tkn) a += sE[p][tkn][lane];
      adb += a;
    }
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) atomicAdd(dw + (p * CE + jg * 12 + i) * 16 + tap, adw[i]);
  if (lane < CE) {
    atomicAdd(db + p * CE + lane, adb);
    atomicAdd(dgamma + p * CE + lane, adg);
    atomicAdd(dbeta + p * CE + lane, adbt);
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

extern "C" int sodt_frontend_bwd(const float* rgb, const float* ir, long ir_bstride, const float* w, const float* b,
                                 const float* gamma, const float* beta, const void* dout,
                                 float* dw, float* db, float* dgamma, float* dbeta, int B, int S, int ca_ws,
                                 int dtype, sodt_stream_t st) {
  (void)beta;
  if (!rgb || !ir || !w || !b || !gamma || !dout || !dw || !db || !dgamma || !dbeta || B <= 0 || S <= 0 || (S % 4)) return SODT_EINVAL;
  if (ca_ws != 1) return SODT_EINVAL;
  FeGeo g{B, S, S / 4, ir_bstride};
  const long ntok = (long)B * g.t * g.t;
  const int nblk = (int)((ntok + 63) / 64);
  const unsigned blocks = (unsigned)(nblk < 1024 ? nblk : 1024);
  if (dtype == SODT_BF16)
    hipLaunchKernelGGL(frontend_bwd_kernel<bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)st, rgb, ir, w, b, gamma, (const bf16*)dout, dw, db, dgamma, dbeta, g, ntok, nblk);
  else if (dtype == SODT_F32)
    hipLaunchKernelGGL(frontend_bwd_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)st, rgb, ir, w, b, gamma, (const float*)dout, dw, db, dgamma, dbeta, g, ntok, nblk);
  else return SODT_EINVAL;
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}
