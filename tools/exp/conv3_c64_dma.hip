// EXPERIMENT (round 5), not built into the library: sodt_conv3x3_c64_fwd's kernel with the halo tiles brought by LDS-DMA into three tile
// buffers, two tiles ahead, one raw barrier per tile and a hand-counted vmcnt at the hand-over.  It drops into csrc/conv3.hip after
// conv3_c64_kernel (it uses that file's C64Args, Tiles, tile_of, tile_origin, c64_rows, c3_zero16 and constants) and was dispatched for
// the launches without an epilogue operand.  Correct (tests/test_conv3_gpu.py, tests/test_sr_gpu.py, tests/test_model_sr_gpu.py: 39
// passed) and measured at 4 x 512 x 512 pixels (tools/mb_conv3.py, same box): 86 us against 94 us without an epilogue operand, 122
// against 118 us with one; SR branch alone 53.7 -> 52.7 ms.  Not shipped: the launch is at the machine's mixed read / write rate either
// way, and 1 ms of a 127 ms step does not pay for a second hand-over protocol whose safety rests on an exact count of the compiler's
// vector-memory instructions per tile.  Two things learnt on the way are recorded in DESIGN.md section 5: __syncthreads() with DMA
// writes to LDS in flight is vmcnt(0) (its release fence), and hipcc tracks only a handful of LDS-DMA instructions with alias
// information - with the six DMAs of a tile unrolled it protected every following LDS read with vmcnt(0).

// ---- the same kernel with the halo tiles brought by LDS-DMA, three tile buffers, two tiles ahead -------------------------------------
// The register prefetch above holds ONE tile in flight per CU and hands it over through two barriers and a stash phase; per tile the
// loads have about one tile's compute time (~3 us) to arrive.  Here the next two tiles are in flight (global_load_lds, 16 bytes per
// lane, the source address carries the swizzle and out-of-image chunks read a zero buffer), there is no stash and one barrier per tile.
// vmcnt is counted by hand at the hand-over: the DMA of tile k was followed by the DMA of tile k + 1 (six instructions per wave, dummies
// included) and by the epilogue loads / stores of tile k - 1, whose number is exact on full tiles - any smaller count is safe.
// The three buffers are three distinct __shared__ arrays and the loop is unrolled by three: hipcc then knows that the reads of one
// buffer do not alias the DMA into another (with a selected pointer it puts vmcnt(0) in front of the first read; edge tiles take
// that path).
constexpr int C_PIECES = (C_CH + 63) / 64, C_BUF = C_PIECES * 1024, C_DR = (C_PIECES + 7) / 8;     // 43 pieces of 1 KB, 6 per wave
typedef __attribute__((address_space(1))) void glb_void;
typedef __attribute__((address_space(3))) void lds_void;

template <bool HAS_E>
__global__ __launch_bounds__(512, 2) void conv3_c64_dma_kernel(const C64Args a, const Tiles t) {
  __shared__ __attribute__((aligned(16))) unsigned char tile0[C_BUF];
  __shared__ __attribute__((aligned(16))) unsigned char tile1[C_BUF];
  __shared__ __attribute__((aligned(16))) unsigned char tile2[C_BUF];
  __shared__ __attribute__((aligned(16))) unsigned char dump[8 * 1024];
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
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the weight / bias loads above are not part of the hand-over counts

  // six DMA instructions per wave and tile, always (a piece past the tile, or no tile at all, reads zeros into the dump area)
  auto dma = [&](long tl, unsigned char* buf) {
    int b = 0, y0 = 0, x0 = 0;
    if (tl >= 0) tile_origin(t, tl, b, y0, x0);
    const char* xb = (const char*)a.x + (((long)b * a.im * t.H + a.im * (y0 - 1) + a.ii) * iW + a.im * (x0 - 1) + a.ij) * 128;
    int lz = lane; asm volatile("" : "+v"(lz));
    // (a rolled loop: hipcc tracks a handful of LDS-DMA instructions with alias information; with 6 x 5 static copies it runs out of
    //  slots and protects every LDS read that follows with vmcnt(0))
#pragma unroll 1
    for (int r = 0; r < C_DR; ++r) {
      const int piece = wave + 8 * r;
      const int i = piece * 64 + lz, pix = i >> 3, ph = i & 7;
      const int py = pix / CPX, px = pix - py * CPX;
      const int gy = y0 - 1 + py, gx = x0 - 1 + px;
      const bool ok = tl >= 0 && i < C_CH && (unsigned)gy < (unsigned)t.H && (unsigned)gx < (unsigned)t.W;
      const char* src = ok ? xb + (unsigned)((py * iW + px) * a.im * 128 + ((ph ^ (px & 7)) << 4)) : (const char*)c3_zero16;
      unsigned char* dst = piece < C_PIECES ? buf + piece * 1024 : dump + wave * 1024;
      __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)dst, 16, 0, 0);
    }
  };
  // hand-over + one tile: wait for this wave's pieces of the tile in `cur` (the 6 DMA instructions of the next tile and the `pend`
  // epilogue operations of the previous tile may stay in flight), barrier, start the DMA of the tile after next into `nxtbuf` (the
  // buffer the previous tile was read from), compute.  Returns the number of epilogue operations this tile issued (0: unknown).
  auto step = [&](long tl, long tl2, const unsigned char* cur, unsigned char* nxtbuf, int pend, auto STATIC_) -> int {
    constexpr bool STATIC = decltype(STATIC_)::value;
    // (a raw s_barrier: __syncthreads() carries a release fence, and with DMA writes to LDS in flight the fence is vmcnt(0).  What the
    //  barrier orders here needs none: this wave's pieces have landed - the counted wait - and every wave's reads of the buffer about to
    //  be overwritten were consumed by its MFMAs before it arrived.)
    if (pend == 16) asm volatile("s_waitcnt vmcnt(22)\n\ts_barrier" ::: "memory");
    else if (pend == 8) asm volatile("s_waitcnt vmcnt(14)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
    dma(tl2, nxtbuf);
    int b, y0, x0;
    tile_origin(t, tl, b, y0, x0);
    const bool full = y0 + C_TH <= t.H && x0 + TW <= t.W;
    if (STATIC && full) { c64_rows<true, HAS_E>(a, t, cur, wf, off, binit, b, y0, x0, n0, h, fr); return HAS_E ? 16 : 8; }
    c64_rows<false, HAS_E>(a, t, cur, wf, off, binit, b, y0, x0, n0, h, fr);
    return 0;
  };
  long t0 = tile_of(t, 0), t1 = tile_of(t, 1);
  dma(t0, tile0);
  dma(t1, tile1);
  int pend = 0;
  for (int k = 0; ; k += 3) {
    const long ta = tile_of(t, k), tb = tile_of(t, k + 1), tc = tile_of(t, k + 2);
    if (ta < 0) break;
    pend = step(ta, tc, tile0, tile2, pend, std::true_type{});
    if (tb < 0) break;
    pend = step(tb, tile_of(t, k + 3), tile1, tile0, pend, std::true_type{});
    if (tc < 0) break;
    pend = step(tc, tile_of(t, k + 4), tile2, tile1, pend, std::true_type{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // nothing of this workgroup's DMA may land after it has gone
}

