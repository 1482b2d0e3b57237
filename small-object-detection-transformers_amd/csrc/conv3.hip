// Direct 3x3 convolutions (stride 1, padding 1) on token-major [B][H][W][64] bf16 rows for the 64-input-channel layers of EDSR
// (edsr.py) - and the one such layer of the detection head (common.py:98-115, the stride-4 C3's Bottleneck).  Two families:
//   * sodt_conv3x3_c64n8_*  - the closing convolution, 64 -> at most 8 channels (this comment);
//   * sodt_conv3x3_c64_*    - 64 -> 64 channels per launch: the body's ResBlocks and, through sodt_conv3_geo, the upsampler stages
//                             conv(64 -> 256) + PixelShuffle(2) as four plane launches (second half of the file).
//
// The closing convolution of EDSR (edsr.py:81-84 `conv(n_feats, num_channels, 3)` = nn.Conv2d(64, ch, 3, padding=1), edsr.py:9-12):
// 64 -> at most 8 channels on the x8 grid - 67 M pixels at BASELINE config 5.  As a nine-segment GEMM it re-read its input nine
// times through the CU memory path for 9 KFLOP per pixel (12.6 / 14.4 / 17.7 ms forward / input gradient / weight gradient); here
// each launch moves its tensors once (HBM bound: 144 B per pixel).
//
//   * forward:  a workgroup walks 16 x 32-pixel tiles; the 18 x 34 halo tile of the input ([pixel][64] bf16, 16-byte chunks XOR-swizzled
//               with the tile column) sits in LDS, the next tile's chunks are in flight in registers meanwhile.  y^T = W x^T per
//               16-pixel strip: the weights are the MFMA A operand (18 fragments, resident in registers for the whole launch; rows 8-15
//               of the 16-row tile are zero), the strip's pixels the B operand (one ds_read_b128 per tap and channel half).
//   * dgrad:    dx^T = Wt dy^T with k = (tap, n): the 16-byte row of a pixel of dy is exactly one k-chunk.  The 16 x 64 result of a
//               strip goes through a per-wave 2 KB patch so that the stores are whole 1 KB runs.
//   * wgrad:    dW[n][tap][c] = sum over pixels dy[p][n] x[p + tap][c]: pixels are the contraction index, both operands come from
//               LDS through the transposing read (ds_read_b64_tr_b16), 36 + 1 accumulator tiles per wave (the "+ 1" against a
//               vector of ones is the bias gradient); one partial per workgroup, summed by a second launch (deterministic).
// The forward can store the (B, cout, H, W) float32 output of the branch directly, the gradients can read the float32 gradient planes
// (deeplabedsr.py:73).  bf16 only (the float32 parity path keeps the K-segment GEMM).
#include "common.h"
#include "../../include/sodt_hip.h"

namespace {

constexpr int TW = 32, CPX = TW + 2;      // tile width in pixels, + halo
constexpr int GRID = 512;                 // persistent: two workgroups per CU

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

struct Tiles {
  int B, H, W, th, tiles_x, tiles_y;
  long ntiles, band;
};

inline Tiles make_tiles(int B, int H, int W, int th) {
  Tiles t;
  t.B = B; t.H = H; t.W = W; t.th = th;
  t.tiles_x = (W + TW - 1) / TW; t.tiles_y = (H + th - 1) / th;
  t.ntiles = (long)B * t.tiles_x * t.tiles_y;
  t.band = (t.ntiles + 7) / 8;
  return t;
}

// Workgroup id -> its k-th tile.  The eight XCDs take the workgroups round-robin, so XCD x = id % 8 walks its own contiguous band of
// tiles, 64 neighbours at a time (row-major: the halo columns two neighbours share are then read from that XCD's L2).
__device__ __forceinline__ long tile_of(const Tiles& t, int k) {
  const int id = blockIdx.x, xcd = id & 7, slot = id >> 3, nslots = gridDim.x >> 3;
  const long local = slot + (long)k * nslots;
  const long tile = xcd * t.band + local;
  return (local < t.band && tile < t.ntiles) ? tile : -1;
}
__device__ __forceinline__ void tile_origin(const Tiles& t, long tile, int& b, int& y0, int& x0) {
  const int tx = (int)(tile % t.tiles_x); tile /= t.tiles_x;
  const int ty = (int)(tile % t.tiles_y); b = (int)(tile / t.tiles_y);
  y0 = ty * t.th; x0 = tx * TW;
}

__device__ __forceinline__ uint4 lds_rd16(const unsigned char* p) { return *(const uint4*)p; }

// ------------------------------------------------------------------------------------------------------------------ forward
constexpr int F_TH = 16, F_PIX = (F_TH + 2) * CPX, F_CH = F_PIX * 8, F_R = (F_CH + 255) / 256;   // 612 pixels, 4896 chunks, 20 rounds

__global__ __launch_bounds__(256, 2) void conv3_n8_fwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w, const float* __restrict__ bias,
                                                             bf16* __restrict__ y, const Tiles t, float* __restrict__ ynchw, const int cout) {
  __shared__ __attribute__((aligned(16))) unsigned char tile[F_PIX * 128];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, fr = lane & 15, fg = lane >> 4;
  // weights: A fragment ks = (tap, channel half): row n = fr (rows >= 8: zero), k = 32 ks + 8 fg ..
  uint4 wf[18];
#pragma unroll
  for (int ks = 0; ks < 18; ++ks)
    wf[ks] = fr < 8 ? *(const uint4*)(w + fr * 576 + 32 * ks + 8 * fg) : make_uint4(0u, 0u, 0u, 0u);
  f32x4 binit = {0.f, 0.f, 0.f, 0.f};
  if (bias && fg < 2) binit = *(const f32x4*)(bias + 4 * fg);
  // per-lane read offsets of tap column kx and channel half hk (the strip's own pixel and the tap row are added per strip)
  uint32_t off[3][2];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int hk = 0; hk < 2; ++hk) off[kx][hk] = (fr + kx) * 128 + ((((hk << 2) | fg) ^ ((fr + kx) & 7)) << 4);

  uint4 pf[F_R];
  auto issue = [&](long tl) {
    int b, y0, x0;
    tile_origin(t, tl, b, y0, x0);
    const char* xb = (const char*)x + (((long)b * t.H + (y0 - 1)) * t.W + (x0 - 1)) * 128;    // (tile origin: uniform; only in-image pixels are read)
    int tz = tid; asm volatile("" : "+v"(tz));       // (not loop-invariant for the compiler: the per-round coordinates are re-derived, not held)
#pragma unroll
    for (int r = 0; r < F_R; ++r) {
      const int i = tz + 256 * r, pix = i >> 3, ch = i & 7;
      const int py = pix / CPX, px = pix - py * CPX;
      const int gy = y0 - 1 + py, gx = x0 - 1 + px;
      const bool ok = i < F_CH && (unsigned)gy < (unsigned)t.H && (unsigned)gx < (unsigned)t.W;
      pf[r] = ok ? *(const uint4*)(xb + (unsigned)((py * t.W + px) * 128 + ch * 16)) : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto stash = [&]() {
    int tz = tid; asm volatile("" : "+v"(tz));       // (not loop-invariant for the compiler: the per-round coordinates are re-derived, not held)
#pragma unroll
    for (int r = 0; r < F_R; ++r) {
      const int i = tz + 256 * r, pix = i >> 3, ch = i & 7;
      const int py = pix / CPX, px = pix - py * CPX;
      if (i < F_CH) *(uint4*)(tile + ((pix << 3) + (ch ^ (px & 7))) * 16) = pf[r];
    }
  };
  long cur = tile_of(t, 0);
  if (cur >= 0) issue(cur);
  for (int k = 0; cur >= 0; ++k) {
    const long nxt = tile_of(t, k + 1);
    __syncthreads();                                  // the previous tile's strips are done with the LDS image
    stash();
    __syncthreads();
    if (nxt >= 0) issue(nxt);
    int b, y0, x0;
    tile_origin(t, cur, b, y0, x0);
#pragma unroll 2
    for (int s = 0; s < 8; ++s) {
      const int sid = wv * 8 + s, oy = sid >> 1, hx = (sid & 1) * 16;
      const unsigned char* sb = tile + (oy * CPX + hx) * 128;
      f32x4 acc = binit;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int hk = 0; hk < 2; ++hk) {
            const uint4 xb = lds_rd16(sb + ky * (CPX * 128) + off[kx][hk]);
            mma16<bf16>(acc, wf[(ky * 3 + kx) * 2 + hk], xb);
          }
      const int gy = y0 + oy, gx = x0 + hx + fr;
      if (ynchw) {
        // the branch's output at the model boundary, (B, cout, H, W) float32 (deeplabedsr.py:73), straight from the accumulators:
        // sixteen consecutive pixels of one channel plane per store
        if (fg < 2 && gy < t.H && gx < t.W) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (4 * fg + r < cout) ynchw[(((long)b * cout + 4 * fg + r) * t.H + gy) * t.W + gx] = acc[r];
        }
      } else if (fg < 2 && gy < t.H && gx < t.W)
        *(uint2*)(y + (((long)b * t.H + gy) * t.W + gx) * 8 + 4 * fg) = make_uint2(pack2bf(acc[0], acc[1]), pack2bf(acc[2], acc[3]));
    }
    cur = nxt;
  }
}

// ------------------------------------------------------------------------------------------------------------------ input gradient
constexpr int D_TH = 16, D_PIX = (D_TH + 2) * CPX, D_R = (D_PIX + 255) / 256;     // 612 one-chunk pixels, 3 rounds

// dy of the closing convolution either as rows [M][8] bf16 or as the (B, cout <= 4, H, W) float32 gradient of the branch's output:
// four plane loads per pixel, rounded to bf16 on the way into the LDS tile (what sodt_rows_from_nchw_f32 did in a launch of its own)
__device__ __forceinline__ uint4 nchw4_load(const float* __restrict__ p, long plane, int cout, bool ok) {
  float f[4] = {0.f, 0.f, 0.f, 0.f};
  if (ok) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (c < cout) f[c] = p[c * plane];
  }
  return make_uint4(pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), 0u, 0u);
}

__global__ __launch_bounds__(256, 2) void conv3_n8_dgrad_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ wT, bf16* __restrict__ dx,
                                                               const Tiles t, const float* __restrict__ dynchw, const int cout) {
  __shared__ __attribute__((aligned(16))) unsigned char tile[D_PIX * 16];
  __shared__ __attribute__((aligned(16))) unsigned char patch[4][16 * 128];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, fr = lane & 15, fg = lane >> 4;
  // Wt: A fragment (ct, ks): row c = 16 ct + fr, k = 32 ks + 8 fg .. = tap (4 ks + fg), n 0 .. 7; taps 9 .. 11 are padding
  uint4 wf[4][3];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
      wf[ct][ks] = (4 * ks + fg) < 9 ? *(const uint4*)(wT + (16 * ct + fr) * 72 + 32 * ks + 8 * fg) : make_uint4(0u, 0u, 0u, 0u);
  // dx[p] = sum over taps dy[p - tap] w[tap]: this lane's tap of k-step ks, as a byte offset from the strip pixel (tile coordinates: + 1)
  int toff[3];
  bool tok[3];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) {
    const int tp = 4 * ks + fg;
    tok[ks] = tp < 9;
    const int ddy = tok[ks] ? tp / 3 - 1 : 0, ddx = tok[ks] ? tp % 3 - 1 : 0;
    toff[ks] = ((1 - ddy) * CPX + (1 - ddx) + fr) * 16;
  }
  unsigned char* const mypatch = patch[wv];
  uint4 pf[D_R];
  auto issue = [&](long tl) {
    int b, y0, x0;
    tile_origin(t, tl, b, y0, x0);
    const char* gb = (const char*)dy + (((long)b * t.H + (y0 - 1)) * t.W + (x0 - 1)) * 16;
    int tz = tid; asm volatile("" : "+v"(tz));
#pragma unroll
    for (int r = 0; r < D_R; ++r) {
      const int pix = tz + 256 * r;
      const int py = pix / CPX, px = pix - py * CPX;
      const int gy = y0 - 1 + py, gx = x0 - 1 + px;
      const bool ok = pix < D_PIX && (unsigned)gy < (unsigned)t.H && (unsigned)gx < (unsigned)t.W;
      if (dynchw) pf[r] = nchw4_load(dynchw + ((long)b * cout * t.H + gy) * t.W + gx, (long)t.H * t.W, cout, ok);
      else pf[r] = ok ? *(const uint4*)(gb + (unsigned)((py * t.W + px) * 16)) : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  long cur = tile_of(t, 0);
  if (cur >= 0) issue(cur);
  for (int k = 0; cur >= 0; ++k) {
    const long nxt = tile_of(t, k + 1);
    __syncthreads();
    int tz = tid; asm volatile("" : "+v"(tz));
#pragma unroll
    for (int r = 0; r < D_R; ++r) {
      const int pix = tz + 256 * r;
      if (pix < D_PIX) *(uint4*)(tile + pix * 16) = pf[r];
    }
    __syncthreads();
    if (nxt >= 0) issue(nxt);
    int b, y0, x0;
    tile_origin(t, cur, b, y0, x0);
    for (int s = 0; s < 8; ++s) {
      const int sid = wv * 8 + s, oy = sid >> 1, hx = (sid & 1) * 16;
      const unsigned char* sb = tile + (oy * CPX + hx) * 16;
      f32x4 acc[4];
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        uint4 g = lds_rd16(sb + toff[ks]);
        if (!tok[ks]) g = make_uint4(0u, 0u, 0u, 0u);       // (a padding tap times a zero weight must not meet an Inf / NaN)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          if (ks == 0) acc[ct] = mma16z<bf16>(wf[ct][0], g);
          else mma16<bf16>(acc[ct], wf[ct][ks], g);
        }
      }
      // acc[ct][r] = dx[pixel fr][c = 16 ct + 4 fg + r] -> patch [pixel][128 B]: chunk (2 ct + (fg >> 1)) ^ (pixel & 7), its 8-byte
      // halves swapped for pixels 8 .. 15 (16 lanes of a ds_write_b64 group then cover 16 different 8-byte slots)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
        *(uint2*)(mypatch + fr * 128 + (((2 * ct + (fg >> 1)) ^ (fr & 7)) << 4) + (((fg & 1) ^ (fr >> 3)) << 3)) =
            make_uint2(pack2bf(acc[ct][0], acc[ct][1]), pack2bf(acc[ct][2], acc[ct][3]));
      const int gy = y0 + oy;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int pix = (lane >> 3) + 8 * j, lc = lane & 7;
        uint4 v = lds_rd16(mypatch + pix * 128 + ((lc ^ (pix & 7)) << 4));
        if (j == 1) v = make_uint4(v.z, v.w, v.x, v.y);
        const int gx = x0 + hx + pix;
        if (gy < t.H && gx < t.W) *(uint4*)(dx + (((long)b * t.H + gy) * t.W + gx) * 64 + lc * 8) = v;
      }
    }
    cur = nxt;
  }
}

// ------------------------------------------------------------------------------------------------------------------ weight gradient
constexpr int W_TH = 8, W_PIX = (W_TH + 2) * CPX, W_CH = W_PIX * 8, W_R = (W_CH + 255) / 256;     // 340 pixels, 2720 chunks, 11 rounds
constexpr int W_PART = 8 * 577;           // one workgroup's partial: [n][tap * 64 + c | bias]

// transposed fragment: lane (fr, fg) <- rows (pixels) p0 + 4 fg + j (j < 4) and p0 + 16 + 4 fg + (j - 4) of column fr of a 16-column
// block; `a` is this lane's own row-segment address (row p0 + 4 g + q, columns 4 p ..), `hi` the byte distance of 16 rows
__device__ __forceinline__ uint4 tr_frag(const unsigned char* a, int hi) {
  union { s16x4 v; uint2 u; } lo, up;
  lo.v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
  up.v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + hi));
  return make_uint4(lo.u.x, lo.u.y, up.u.x, up.u.y);
}

__global__ __launch_bounds__(256, 2) void conv3_n8_wgrad_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x, float* __restrict__ part,
                                                               const Tiles t, const float* __restrict__ dynchw, const int cout) {
  __shared__ __attribute__((aligned(16))) unsigned char tile[W_PIX * 128];          // x halo tile (swizzled chunks); the reduction table at the end
  __shared__ __attribute__((aligned(16))) unsigned char gt[W_TH * TW * 32];         // dy tile, rows padded to 16 columns (8 .. 15 zero)
  static_assert(W_PIX * 128 >= W_PART * 4, "reduction table fits the tile");
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, fr = lane & 15, fg = lane >> 4;
  const int q = (lane & 15) >> 2, p = lane & 3;
  for (int i = tid; i < W_TH * TW; i += 256) *(uint4*)(gt + i * 32 + 16) = make_uint4(0u, 0u, 0u, 0u);
  f32x4 acc[9][4], accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[tp][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  const uint4 ones = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
  uint4 pf[W_R], pg;
  auto issue = [&](long tl) {
    int b, y0, x0;
    tile_origin(t, tl, b, y0, x0);
    const char* xb = (const char*)x + (((long)b * t.H + (y0 - 1)) * t.W + (x0 - 1)) * 128;
    const char* gb = (const char*)dy + (((long)b * t.H + y0) * t.W + x0) * 16;
    int tz = tid; asm volatile("" : "+v"(tz));
#pragma unroll
    for (int r = 0; r < W_R; ++r) {
      const int i = tz + 256 * r, pix = i >> 3, ch = i & 7;
      const int py = pix / CPX, px = pix - py * CPX;
      const int gy = y0 - 1 + py, gx = x0 - 1 + px;
      const bool ok = i < W_CH && (unsigned)gy < (unsigned)t.H && (unsigned)gx < (unsigned)t.W;
      pf[r] = ok ? *(const uint4*)(xb + (unsigned)((py * t.W + px) * 128 + ch * 16)) : make_uint4(0u, 0u, 0u, 0u);
    }
    const int gy = y0 + (tid >> 5), gx = x0 + (tid & 31);
    if (dynchw) pg = nchw4_load(dynchw + ((long)b * cout * t.H + gy) * t.W + gx, (long)t.H * t.W, cout, gy < t.H && gx < t.W);
    else pg = (gy < t.H && gx < t.W) ? *(const uint4*)(gb + (unsigned)(((tid >> 5) * t.W + (tid & 31)) * 16)) : make_uint4(0u, 0u, 0u, 0u);
  };
  long cur = tile_of(t, 0);
  if (cur >= 0) issue(cur);
  for (int k = 0; cur >= 0; ++k) {
    const long nxt = tile_of(t, k + 1);
    __syncthreads();
    int tz = tid; asm volatile("" : "+v"(tz));
#pragma unroll
    for (int r = 0; r < W_R; ++r) {
      const int i = tz + 256 * r, pix = i >> 3, ch = i & 7;
      const int py = pix / CPX, px = pix - py * CPX;
      if (i < W_CH) *(uint4*)(tile + ((pix << 3) + (ch ^ (px & 7))) * 16) = pf[r];
    }
    *(uint4*)(gt + tid * 32) = pg;
    __syncthreads();
    if (nxt >= 0) issue(nxt);
#pragma unroll 1
    for (int s = 0; s < 2; ++s) {
      const int oy = wv * 2 + s;
      // A = dy^T: rows n (lane fr), k-slots = the strip's 32 pixels in the transposed-read order
      const uint4 ga = tr_frag(gt + (oy * TW + 4 * fg + q) * 32 + 8 * p, 16 * 32);
      mma16<bf16>(accb, ga, ones);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int cx = kx + 4 * fg + q;                      // tile column of this lane's row segment (+ 16 for the upper half: same & 7)
        const unsigned char* rb = tile + (oy * CPX + cx) * 128 + (p & 1) * 8;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          const int co = ((2 * ct + (p >> 1)) ^ (cx & 7)) << 4;
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const uint4 xb = tr_frag(rb + ky * (CPX * 128) + co, 16 * 128);
            mma16<bf16>(acc[ky * 3 + kx][ct], ga, xb);
          }
        }
      }
    }
    cur = nxt;
  }
  // ---- workgroup partial: acc[tap][ct][r] = dW[n = 4 fg + r][tap][c = 16 ct + fr] (lanes fg < 2); the four waves add theirs to the LDS
  //      table one after the other (a fixed order: two runs give the same bits)
  __syncthreads();
  float* red = (float*)tile;
  for (int i = tid; i < W_PART; i += 256) red[i] = 0.f;
  for (int turn = 0; turn < 4; ++turn) {
    __syncthreads();
    if (turn == wv && fg < 2) {
#pragma unroll
      for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[(4 * fg + r) * 577 + tp * 64 + 16 * ct + fr] += acc[tp][ct][r];
      if (fr == 0)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(4 * fg + r) * 577 + 576] += accb[r];
    }
  }
  __syncthreads();
  float* dst = part + (long)blockIdx.x * W_PART;
  for (int i = tid; i < W_PART; i += 256) dst[i] = red[i];
}

// dw [cout][64][3][3] += sum of the partials (k = tap * 64 + c -> the torch layout), db[cout] += column 576
__global__ __launch_bounds__(256) void conv3_n8_wgrad_reduce_kernel(const float* __restrict__ part, int nparts, float* __restrict__ dw, float* __restrict__ db,
                                                                   int cout) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= cout * 577) return;
  const int n = i / 577, k = i - n * 577;
  float s = 0.f;
  for (int j = 0; j < nparts; ++j) s += part[(long)j * W_PART + n * 577 + k];
  if (k < 576) dw[(n * 64 + (k & 63)) * 9 + (k >> 6)] += s;
  else if (db) db[n] += s;
}

// ================================================================================================== 64 -> 64 channels (EDSR's body)
// The residual blocks (edsr.py:34-53: conv -> ReLU -> conv, + x) and the head / closing convolutions of the body are 3x3 at 64 -> 64
// channels.  8 x 32-pixel tiles (10 x 34 halo tile in LDS), one 8-wave workgroup per CU: wave (w, h) owns the OUTPUT channels 16 w ..
// 16 w + 15 (its 18 weight fragments resident in registers) of the 16-pixel strip h of all eight tile rows, so no weight ever goes
// through LDS and the input crosses the CU once.  flip: the input gradient is the same correlation with the taps mirrored (tap' = 8 - tap) and the transposed weights.
// Epilogue in the GEMM's order: + bias, ReLU, ReLU mask (aux > 0), + residual.
// The halo chunks of the next tile are PREDICATED loads (`ok ? load : 0` in issue(): out-of-image chunks are zeros), issued once per
// tile ahead of the row walk.  What is unconditional is the EPILOGUE-operand fetch of the FULL instantiation (tiles inside the image;
// tiles that stick out take the second, predicated instantiation): inside the row loop hipcc then counts vmcnt exactly, so the operands
// fetched four rows ahead stay in flight across the rows - with a predicated load inside that loop it falls back to vmcnt(0) before
// every row and the wave eats a memory latency sixteen times per tile.
// Geometry maps (sodt_conv3_geo, include/sodt_hip.h): the input pixel (y, x) of the H x W grid the kernel walks may live at
// (im y + ii, im x + ij) of an (im H) x (im W) tensor, the output pixel likewise (om, oi, oj), and output channel n may take weight /
// bias row wrs n + wro.  With (4, 2 i + j) and om = 2 a 64 -> 256 convolution + nn.PixelShuffle(2) (edsr.py:19-24) is four launches that
// write the shuffled tensor directly; with im = 2 the four planes of the fine gradient are read in place (no inverse shuffle).
struct C64Args {
  const bf16* x; const bf16* w; const float* bias; const bf16* resid; const bf16* aux; bf16* y;
  int flags, flip;
  int wrs, wro, im, ii, ij, om, oi, oj;
};

constexpr int C_TH = 8, C_PIX = (C_TH + 2) * CPX, C_CH = C_PIX * 8, C_R = (C_CH + 511) / 512;   // 340 pixels, 2720 chunks, 6 rounds of 512 threads

// One tile: wave (w, h) owns output channels 16 w .. 16 w + 15 of the 16-pixel strip h of every tile row and slides down the ten halo
// rows.  The six fragments of halo row r (three tap columns x two channel halves) are read ONCE and feed the three output rows r, r - 1,
// r - 2 (tap rows 0, 1, 2), whose accumulators roll through acc[row % 3]: a third of the LDS reads of a row-by-row walk (which left LDS
// and the matrix pipe each ~50 % busy, one after the other).  The next row's fragments are read behind this row's MFMAs
// (sched_group_barrier; hipcc would sink every read next to its MFMA and pay an LDS latency per pair).
template <bool FULL, bool HAS_E>
__device__ __forceinline__ void c64_rows(const C64Args& a, const Tiles& t, const unsigned char* tile, const uint4 (&wf)[18], const uint32_t (&off)[3][2],
                                         const f32x4 binit, int b, int y0, int x0, int n0, int h, int fr) {
  const bool relu = a.flags & SODT_EPI_RELU, drelu = HAS_E && (a.flags & SODT_EPI_DRELU), res = HAS_E && (a.flags & SODT_EPI_RESID);
  const bf16* eop = drelu ? a.aux : a.resid;         // (at most one of the two: checked by the entry point)
  const int oW = a.om * t.W;                          // output row pitch in pixels
  const long tb = (((long)b * a.om * t.H + a.om * y0 + a.oi) * oW + a.om * (x0 + 16 * h) + a.oj) * 64 + n0;
  const unsigned ofr = (unsigned)(fr * a.om * 64), orow = (unsigned)(a.om * oW * 64);
  const bool colok = FULL || x0 + 16 * h + fr < t.W;
  uint2 re[4];
  auto ldrow = [&](int row, uint2& q) {
    if (!HAS_E) return;
    // FULL: no predicated access anywhere in the loop (hipcc then counts vmcnt exactly and the loads stay in flight across rows)
    const bool ok = FULL || (colok && y0 + row < t.H);
    q = ok ? *(const uint2*)(eop + tb + (row * orow + ofr)) : make_uint2(0u, 0u);
  };
#pragma unroll
  for (int jr = 0; jr < 4; ++jr) ldrow(jr, re[jr]);
  f32x4 acc[3];
  uint4 fq[2][6];                                     // fragments of two consecutive halo rows
  auto rd = [&](int r, uint4 (&f)[6]) {
    const unsigned char* sb = tile + (r * CPX + 16 * h) * 128;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) { f[2 * kx] = lds_rd16(sb + off[kx][0]); f[2 * kx + 1] = lds_rd16(sb + off[kx][1]); }
  };
  rd(0, fq[0]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int r = 0; r < C_TH + 2; ++r) {
    if (r + 1 < C_TH + 2) rd(r + 1, fq[(r + 1) & 1]);
    int nmm = 0;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int hk = 0; hk < 2; ++hk)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int oy = r - ky;
          if (oy >= 0 && oy < C_TH) {
            if (ky == 0 && kx == 0 && hk == 0) acc[oy % 3] = binit;
            mma16<bf16>(acc[oy % 3], wf[(ky * 3 + kx) * 2 + hk], fq[r & 1][2 * kx + hk]);
            ++nmm;
          }
        }
    if (r + 1 < C_TH + 2) {                            // pin: one LDS read behind every third (second, first) MFMA of the row
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        if (nmm == 18) __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        else if (nmm == 12) __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        else __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);                 // (the next row's MFMAs must not be pulled up next to "their" reads)
    if (r >= 2) {
      const int oy = r - 2, jr = oy & 3;
      if (FULL || (colok && y0 + oy < t.H)) {
        f32x4 v = acc[oy % 3];
        if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        if (drelu) {
          const uint2 q = re[jr];
          // bf16 > 0  <=>  sign clear and not zero
          if (!((q.x & 0x7fffu) && !(q.x & 0x8000u))) v[0] = 0.f;
          if (!((q.x & 0x7fff0000u) && !(q.x & 0x80000000u))) v[1] = 0.f;
          if (!((q.y & 0x7fffu) && !(q.y & 0x8000u))) v[2] = 0.f;
          if (!((q.y & 0x7fff0000u) && !(q.y & 0x80000000u))) v[3] = 0.f;
        }
        if (res) {
          const uint2 q = re[jr];
          v[0] += __uint_as_float(q.x << 16); v[1] += __uint_as_float(q.x & 0xffff0000u);
          v[2] += __uint_as_float(q.y << 16); v[3] += __uint_as_float(q.y & 0xffff0000u);
        }
        *(uint2*)(a.y + tb + (oy * orow + ofr)) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
      }
      if (oy + 4 < C_TH) ldrow(oy + 4, re[jr]);
    }
  }
}

__global__ __launch_bounds__(512, 2) void conv3_c64_kernel(const C64Args a, const Tiles t) {
  __shared__ __attribute__((aligned(16))) unsigned char tile[C_PIX * 128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wv = wave & 3, h = wave >> 2, fr = lane & 15, fg = lane >> 4;
  uint4 wf[18];
#pragma unroll
  for (int ks = 0; ks < 18; ++ks) {
    const int tp = ks >> 1, col = (a.flip ? 8 - tp : tp) * 64 + (ks & 1) * 32 + 8 * fg;
    wf[ks] = *(const uint4*)(a.w + (long)(a.wrs * (16 * wv + fr) + a.wro) * 576 + col);
  }
  const int n0 = 16 * wv + 4 * fg;
  f32x4 binit = {0.f, 0.f, 0.f, 0.f};
  if (a.flags & SODT_EPI_BIAS) {
#pragma unroll
    for (int r = 0; r < 4; ++r) binit[r] = a.bias[a.wrs * (n0 + r) + a.wro];
  }
  uint32_t off[3][2];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int hk = 0; hk < 2; ++hk) off[kx][hk] = (fr + kx) * 128 + ((((hk << 2) | fg) ^ ((fr + kx) & 7)) << 4);
  const int iW = a.im * t.W;

  uint4 pf[C_R];
  auto issue = [&](long tl) {
    int b, y0, x0;
    tile_origin(t, tl, b, y0, x0);
    const char* xb = (const char*)a.x + (((long)b * a.im * t.H + a.im * (y0 - 1) + a.ii) * iW + a.im * (x0 - 1) + a.ij) * 128;
    int tz = tid; asm volatile("" : "+v"(tz));
#pragma unroll
    for (int r = 0; r < C_R; ++r) {
      const int i = tz + 512 * r, pix = i >> 3, ch = i & 7;
      const int py = pix / CPX, px = pix - py * CPX;
      const int gy = y0 - 1 + py, gx = x0 - 1 + px;
      const bool ok = i < C_CH && (unsigned)gy < (unsigned)t.H && (unsigned)gx < (unsigned)t.W;
      pf[r] = ok ? *(const uint4*)(xb + (unsigned)((py * iW + px) * a.im * 128 + ch * 16)) : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto stash = [&]() {
    int tz = tid; asm volatile("" : "+v"(tz));
#pragma unroll
    for (int r = 0; r < C_R; ++r) {
      const int i = tz + 512 * r, pix = i >> 3, ch = i & 7;
      const int py = pix / CPX, px = pix - py * CPX;
      if (i < C_CH) *(uint4*)(tile + ((pix << 3) + (ch ^ (px & 7))) * 16) = pf[r];
    }
  };
  long cur = tile_of(t, 0);
  if (cur >= 0) issue(cur);
  for (int k = 0; cur >= 0; ++k) {
    const long nxt = tile_of(t, k + 1);
    __syncthreads();
    stash();
    __syncthreads();
    if (nxt >= 0) issue(nxt);
    int b, y0, x0;
    tile_origin(t, cur, b, y0, x0);
    const bool full = y0 + C_TH <= t.H && x0 + TW <= t.W;
    if (a.flags & (SODT_EPI_DRELU | SODT_EPI_RESID)) {
      if (full) c64_rows<true, true>(a, t, tile, wf, off, binit, b, y0, x0, n0, h, fr);
      else c64_rows<false, true>(a, t, tile, wf, off, binit, b, y0, x0, n0, h, fr);
    } else {
      if (full) c64_rows<true, false>(a, t, tile, wf, off, binit, b, y0, x0, n0, h, fr);
      else c64_rows<false, false>(a, t, tile, wf, off, binit, b, y0, x0, n0, h, fr);
    }
    cur = nxt;
  }
}

// weight gradient, 64 -> 64: dW[n][tap][c] = sum_p dy[p][n] x[p + tap][c], pixels as the MFMA contraction index, both operands from LDS
// through the transposing read.  Eight waves: wave (w, e) keeps the accumulator tiles of output channels 16 w .. 16 w + 15 x input
// channels 32 e .. 32 e + 31 for all nine taps (18 tiles + 1 against a vector of ones: the bias gradient) and SLIDES down the ten halo
// rows of x like the forward kernel: the six x fragments of halo row r (three tap columns x two channel tiles) meet the dy fragments
// of rows r, r - 1, r - 2 (tap rows 0, 1, 2; a rolling window), so an x fragment is read once per tile, not once per tap row.
// The next row's fragments are read behind this row's MFMAs (sched_group_barrier / sched_barrier as in c64_rows).  One partial
// [64][577] per workgroup, summed by the reduce launch.
constexpr int G_TH = 8, G_XCH = (G_TH + 2) * CPX * 8, G_XR = (G_XCH + 511) / 512, G_YCH = G_TH * TW * 8, G_YR = G_YCH / 512;   // 2720 / 6, 2048 / 4
constexpr int G_PART = 64 * 577;

__global__ __launch_bounds__(512, 2) void conv3_c64_wgrad_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x, float* __restrict__ part,
                                                                const Tiles t, const int dm, const int di, const int dj) {
  __shared__ __attribute__((aligned(16))) unsigned char tile[(G_TH + 2) * CPX * 128];     // x halo tile
  __shared__ __attribute__((aligned(16))) unsigned char gt[G_TH * TW * 128];              // dy tile (chunks swizzled with the pixel's column)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wv = wave & 3, ce = wave >> 2, fr = lane & 15, fg = lane >> 4;
  f32x4 acc[9][2], accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) acc[tp][c2] = f32x4{0.f, 0.f, 0.f, 0.f};
  const uint4 ones = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
  uint4 pf[G_XR], pg[G_YR];
  auto issue = [&](long tl) {
    int b, y0, x0;
    tile_origin(t, tl, b, y0, x0);
    const char* xb = (const char*)x + (((long)b * t.H + (y0 - 1)) * t.W + (x0 - 1)) * 128;
    const int dW = dm * t.W;                            // dy pixel (y, x) lives at (dm y + di, dm x + dj) of a (dm H) x (dm W) tensor
    const char* gb = (const char*)dy + (((long)b * dm * t.H + dm * y0 + di) * dW + dm * x0 + dj) * 128;
    int tz = tid; asm volatile("" : "+v"(tz));
#pragma unroll
    for (int r = 0; r < G_XR; ++r) {
      const int i = tz + 512 * r, pix = i >> 3, ch = i & 7;
      const int py = pix / CPX, px = pix - py * CPX;
      const int gy = y0 - 1 + py, gx = x0 - 1 + px;
      const bool ok = i < G_XCH && (unsigned)gy < (unsigned)t.H && (unsigned)gx < (unsigned)t.W;
      pf[r] = ok ? *(const uint4*)(xb + (unsigned)((py * t.W + px) * 128 + ch * 16)) : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int r = 0; r < G_YR; ++r) {
      const int i = tz + 512 * r, pix = i >> 3, ch = i & 7;
      const int py = pix >> 5, px = pix & 31;
      const bool ok = y0 + py < t.H && x0 + px < t.W;
      pg[r] = ok ? *(const uint4*)(gb + (unsigned)((py * dW + px) * dm * 128 + ch * 16)) : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  long cur = tile_of(t, 0);
  if (cur >= 0) issue(cur);
  for (int k = 0; cur >= 0; ++k) {
    const long nxt = tile_of(t, k + 1);
    __syncthreads();
    {
      int tz = tid; asm volatile("" : "+v"(tz));
#pragma unroll
      for (int r = 0; r < G_XR; ++r) {
        const int i = tz + 512 * r, pix = i >> 3, ch = i & 7;
        const int py = pix / CPX, px = pix - py * CPX;
        if (i < G_XCH) *(uint4*)(tile + ((pix << 3) + (ch ^ (px & 7))) * 16) = pf[r];
      }
#pragma unroll
      for (int r = 0; r < G_YR; ++r) {
        const int i = tz + 512 * r, pix = i >> 3, ch = i & 7;
        *(uint4*)(gt + ((pix << 3) + (ch ^ (pix & 7))) * 16) = pg[r];
      }
    }
    __syncthreads();
    if (nxt >= 0) issue(nxt);
    int lz = lane; asm volatile("" : "+v"(lz));          // (lane-derived offsets re-derived per tile: as loop invariants they are spilled)
    const int p = lz & 3, ox = (lz >> 4) * 4 + ((lz & 15) >> 2);   // this lane's row segment: strip pixel ox (and ox + 16: same & 7)
    uint4 af[4];                                          // dy fragments of rows r - 2 .. r + 1 (the next one in flight): rows n (lane fr), k-slots = 32 pixels
    uint4 bq[2][6];                                       // x fragments of two consecutive halo rows: [tap column][channel tile]
    auto rdA = [&](int oy, uint4& f) {
      f = tr_frag(gt + (oy * TW + ox) * 128 + (((2 * wv + (p >> 1)) ^ (ox & 7)) << 4) + (p & 1) * 8, 16 * 128);
    };
    auto rdB = [&](int r, uint4 (&f)[6]) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int cx = kx + ox;
        const unsigned char* rb = tile + (r * CPX + cx) * 128 + (p & 1) * 8;
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) f[2 * kx + c2] = tr_frag(rb + (((2 * (2 * ce + c2) + (p >> 1)) ^ (cx & 7)) << 4), 16 * 128);
      }
    };
    rdA(0, af[0]);
    rdB(0, bq[0]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < G_TH + 2; ++r) {
      if (r + 1 < G_TH + 2) rdB(r + 1, bq[(r + 1) & 1]);
      if (r + 1 < G_TH) rdA(r + 1, af[(r + 1) & 3]);
      int nm = 0;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int oy = r - ky;
        if (oy >= 0 && oy < G_TH) {
          if (ky == 0) { mma16<bf16>(accb, af[oy & 3], ones); ++nm; }
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) { mma16<bf16>(acc[ky * 3 + kx][c2], af[oy & 3], bq[r & 1][2 * kx + c2]); ++nm; }
        }
      }
      const int nr = 24 * (r + 1 < G_TH + 2) + 2 * (r + 1 < G_TH);
#pragma unroll
      for (int q = 0; q < 26; ++q)
        if (q < nm && q < nr) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    cur = nxt;
  }
  // acc[tap][c2][r] = dW[n = 16 wv + 4 fg + r][tap][c = 16 (2 ce + c2) + fr]: every (wave, lane, r) owns its own entries of the partial
  float* dst = part + (long)blockIdx.x * G_PART;
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(16 * wv + 4 * fg + r) * 577 + tp * 64 + 16 * (2 * ce + c2) + fr] = acc[tp][c2][r];
  if (fr == 0 && ce == 0)
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[(16 * wv + 4 * fg + r) * 577 + 576] = accb[r];
}

// dw += the sum of the partials in a fixed order: a block owns 64 consecutive entries, its four waves take every fourth partial (256-byte
// row reads), LDS adds the four sums
__global__ __launch_bounds__(256) void conv3_c64_wgrad_reduce_kernel(const float* __restrict__ part, int nparts, float* __restrict__ dw, float* __restrict__ db,
                                                                   int wrs, int wro) {
  __shared__ float sm[4][64];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (i < G_PART) {
    int j = g;
    for (; j + 12 < nparts; j += 16) {
      const float a0 = part[(long)j * G_PART + i], a1 = part[(long)(j + 4) * G_PART + i], a2 = part[(long)(j + 8) * G_PART + i],
                  a3 = part[(long)(j + 12) * G_PART + i];
      s += (a0 + a1) + (a2 + a3);
    }
    for (; j < nparts; j += 4) s += part[(long)j * G_PART + i];
  }
  sm[g][lane] = s;
  __syncthreads();
  if (g == 0 && i < G_PART) {
    s = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
    const int n = wrs * (i / 577) + wro, k = i % 577;
    if (k < 576) dw[((long)n * 64 + (k & 63)) * 9 + (k >> 6)] += s;
    else if (db) db[n] += s;
  }
}

inline bool geo_ok(const sodt_conv3_geo* g) {
  return !g || (g->w_row_stride >= 1 && g->w_row_off >= 0 && g->w_row_off < g->w_row_stride && g->in_mul >= 1 && g->out_mul >= 1 &&
                g->in_i >= 0 && g->in_i < g->in_mul && g->in_j >= 0 && g->in_j < g->in_mul && g->out_i >= 0 && g->out_i < g->out_mul &&
                g->out_j >= 0 && g->out_j < g->out_mul && g->in_mul <= 2 && g->out_mul <= 2);
}

inline bool ok_common(const void* a, const void* b, const void* c, int B, int H, int W, int dtype) {
  return dtype == SODT_BF16 && a && b && c && !((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c)) & 15) && B > 0 && H > 0 && W > 0 &&
         (long)B * H * W < (1L << 31);
}

}  // namespace

extern "C" {

int sodt_conv3x3_c64n8_fwd(const void* x, const void* w, const float* bias, void* y, float* y_nchw, int B, int H, int W, int cout, int dtype,
                           hipStream_t st) {
  if (!ok_common(x, w, y_nchw ? (const void*)y_nchw : y, B, H, W, dtype) || (((uintptr_t)bias) & 15) || cout < 1 || cout > 8) return SODT_EINVAL;
  const Tiles t = make_tiles(B, H, W, F_TH);
  hipLaunchKernelGGL(conv3_n8_fwd_kernel, dim3(GRID), dim3(256), 0, st, (const bf16*)x, (const bf16*)w, bias, (bf16*)y, t, y_nchw, cout);
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

int sodt_conv3x3_c64n8_dgrad(const void* dy, const float* dy_nchw, const void* wT, void* dx, int B, int H, int W, int cout, int dtype, hipStream_t st) {
  if (!ok_common(dy_nchw ? (const void*)dy_nchw : dy, wT, dx, B, H, W, dtype) || cout < 1 || cout > 8 || (dy_nchw && cout > 4)) return SODT_EINVAL;
  const Tiles t = make_tiles(B, H, W, D_TH);
  hipLaunchKernelGGL(conv3_n8_dgrad_kernel, dim3(GRID), dim3(256), 0, st, (const bf16*)dy, (const bf16*)wT, (bf16*)dx, t, dy_nchw, cout);
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

long sodt_conv3x3_c64n8_wgrad_scratch_bytes(void) { return (long)GRID * W_PART * 4; }

int sodt_conv3x3_c64n8_wgrad(const void* dy, const float* dy_nchw, const void* x, float* dw, float* db, float* scratch, int B, int H, int W, int cout,
                             int dtype, hipStream_t st) {
  if (!ok_common(dy_nchw ? (const void*)dy_nchw : dy, x, scratch, B, H, W, dtype) || !dw || cout < 1 || cout > 8 || (dy_nchw && cout > 4))
    return SODT_EINVAL;
  const Tiles t = make_tiles(B, H, W, W_TH);
  hipLaunchKernelGGL(conv3_n8_wgrad_kernel, dim3(GRID), dim3(256), 0, st, (const bf16*)dy, (const bf16*)x, scratch, t, dy_nchw, cout);
  hipLaunchKernelGGL(conv3_n8_wgrad_reduce_kernel, dim3((cout * 577 + 255) / 256), dim3(256), 0, st, (const float*)scratch, GRID, dw, db, cout);
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

int sodt_conv3x3_c64_fwd(const void* x, const void* w, const float* bias, const void* resid, const void* aux, void* y, int B, int H, int W,
                         int flags, int flip, const sodt_conv3_geo* geo, int dtype, hipStream_t st) {
  if (!ok_common(x, w, y, B, H, W, dtype) || ((((uintptr_t)bias) | ((uintptr_t)resid) | ((uintptr_t)aux)) & 15) || !geo_ok(geo)) return SODT_EINVAL;
  if (flags & ~(SODT_EPI_BIAS | SODT_EPI_RELU | SODT_EPI_DRELU | SODT_EPI_RESID)) return SODT_EINVAL;
  if (((flags & SODT_EPI_BIAS) && !bias) || ((flags & SODT_EPI_RESID) && !resid) || ((flags & SODT_EPI_DRELU) && !aux)) return SODT_EINVAL;
  if ((flags & SODT_EPI_RESID) && (flags & SODT_EPI_DRELU)) return SODT_EINVAL;       // one epilogue operand per launch (no layer of the branch has both)
  C64Args a;
  a.x = (const bf16*)x; a.w = (const bf16*)w; a.bias = bias; a.resid = (const bf16*)resid; a.aux = (const bf16*)aux; a.y = (bf16*)y;
  a.flags = flags; a.flip = flip ? 1 : 0;
  a.wrs = geo ? geo->w_row_stride : 1; a.wro = geo ? geo->w_row_off : 0;
  a.im = geo ? geo->in_mul : 1; a.ii = geo ? geo->in_i : 0; a.ij = geo ? geo->in_j : 0;
  a.om = geo ? geo->out_mul : 1; a.oi = geo ? geo->out_i : 0; a.oj = geo ? geo->out_j : 0;
  if ((long)B * H * W * a.im * a.im >= (1L << 31) || (long)B * H * W * a.om * a.om >= (1L << 31)) return SODT_EINVAL;
  const Tiles t = make_tiles(B, H, W, C_TH);
  const int grid = (int)(t.ntiles < GRID / 2 ? ((t.ntiles + 7) / 8) * 8 : GRID / 2);       // one 8-wave workgroup per CU
  hipLaunchKernelGGL(conv3_c64_kernel, dim3(grid), dim3(512), 0, st, a, t);
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

long sodt_conv3x3_c64_wgrad_scratch_bytes(void) { return (long)(GRID / 2) * G_PART * 4; }

int sodt_conv3x3_c64_wgrad(const void* dy, const void* x, float* dw, float* db, float* scratch, int B, int H, int W, const sodt_conv3_geo* geo,
                           int dtype, hipStream_t st) {
  if (!ok_common(dy, x, scratch, B, H, W, dtype) || !dw || !geo_ok(geo)) return SODT_EINVAL;
  if (geo && geo->in_mul != 1) return SODT_EINVAL;                       // (x is always on the walked grid; dy uses the out_* map)
  const int dm = geo ? geo->out_mul : 1;
  if ((long)B * H * W * dm * dm >= (1L << 31)) return SODT_EINVAL;
  const Tiles t = make_tiles(B, H, W, G_TH);
  const int grid = (int)(t.ntiles < GRID / 2 ? ((t.ntiles + 7) / 8) * 8 : GRID / 2);       // one 8-wave workgroup per CU
  hipLaunchKernelGGL(conv3_c64_wgrad_kernel, dim3(grid), dim3(512), 0, st, (const bf16*)dy, (const bf16*)x, scratch, t, dm, geo ? geo->out_i : 0,
                     geo ? geo->out_j : 0);
  hipLaunchKernelGGL(conv3_c64_wgrad_reduce_kernel, dim3((G_PART + 63) / 64), dim3(256), 0, st, (const float*)scratch, grid, dw, db,
                     geo ? geo->w_row_stride : 1, geo ? geo->w_row_off : 0);
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}

}  // extern "C"
