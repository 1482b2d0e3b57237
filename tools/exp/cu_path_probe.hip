// Memory-only skeleton of the fused block kernel's TRAINING launch (round 6): what do its bytes cost when nothing is computed?
// Per window pair (128 tokens) a workgroup of wmsa_hg.hip moves 288 KB of weights L2 -> LDS by LDS-DMA, reads 48 KB of x from HBM
// and stores 4 x 48 KB (x_mid, xn2, xn1, ao).  This probe issues exactly those requests (same instruction forms, same 8 rows x 128 B
// store shape, 16 pairs per workgroup at B = 8 @ 1024^2), selectable by a bit mask, and times each combination:
//   hipcc --offload-arch=gfx950 -O3 tools/exp/cu_path_probe.hip -o tools/exp/cu_path_probe && tools/exp/cu_path_probe
// If "all" ~ weights + loads + stores the CU's memory path serialises them (DESIGN 4.1d's reading); if "all" ~ max(...) they overlap
// and the kernel's 0.32 ms is something else.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ROWB = 384, PAIR_ROWS = 128, WBYTES = 288 * 1024, LDSB = 72 * 1024;

__global__ __launch_bounds__(512, 2) void probe(const unsigned char* __restrict__ x, const unsigned char* __restrict__ wts,
                                                unsigned char* __restrict__ o0, unsigned char* __restrict__ o1, unsigned char* __restrict__ o2,
                                                unsigned char* __restrict__ o3, unsigned* sink, int npairs, int mask, int nout) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned smem0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned acc = 0;
  for (int it = blockIdx.x; it < npairs; it += gridDim.x) {
    // token rows of this pair: two 8 x 8 windows of a 256 x 256 token image = 16 runs of 8 consecutive rows; thread -> (row, chunk)
    const int r8 = lane >> 3, c8 = lane & 7;
    uint4 v[6];
    unsigned rowoff[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int win = 2 * it + (w >> 2), wy = (win >> 5) & 31, wx = win & 31, img = win >> 10;
      const int trow = (w & 3) * 16 + hh * 8 + r8;                      // token of the window: window row trow >> 3
      const long row = ((long)img * 256 + wy * 8 + (trow >> 3)) * 256 + wx * 8 + (trow & 7);
      rowoff[hh] = (unsigned)(row * ROWB + c8 * 16);
    }
    if (mask & 2) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int i = 0; i < 3; ++i) v[3 * hh + i] = *(const uint4*)(x + rowoff[hh] + 128u * i);
    } else {
#pragma unroll
      for (int i = 0; i < 6; ++i) v[i] = make_uint4(it, i, tid, 7);
    }
    if (mask & 1) {
      // 288 one-KB pieces, 36 per wave, into the (wrapped) 72 KB buffer
#pragma unroll
      for (int q = 0; q < 36; ++q) {
        const unsigned p = (unsigned)(w + 8 * q);
        const unsigned off = p << 10;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     :: "v"((unsigned)(lane * 16) + off), "s"(wts), "s"(smem0 + (off % LDSB)) : "memory", "m0");
      }
    }
    if (mask & 2) {
#pragma unroll
      for (int i = 0; i < 6; ++i) acc += v[i].x ^ v[i].w;
    }
    if (mask & 4) {
      unsigned char* outs[4] = {o0, o1, o2, o3};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (t >= nout) break;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
          for (int i = 0; i < 3; ++i) *(uint4*)(outs[t] + rowoff[hh] + 128u * i) = v[3 * hh + i];
      }
    }
    if (mask & 1) { asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); __syncthreads(); acc += smem[(tid * 16) % LDSB]; }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 0x1234567u) sink[blockIdx.x] = acc;
}

int main() {
  const long T = 8L * 256 * 256;
  unsigned char *x, *wts, *o[4]; unsigned* sink;
  hipMalloc(&x, T * ROWB); hipMalloc(&wts, WBYTES); hipMalloc(&sink, 4096);
  for (int i = 0; i < 4; ++i) hipMalloc(&o[i], T * ROWB);
  hipMemset(x, 1, T * ROWB); hipMemset(wts, 2, WBYTES);
  float* big; hipMalloc(&big, 1L << 30);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB + 80 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int npairs = (int)(T / PAIR_ROWS);
  struct { const char* name; int mask, nout; } cases[] = {
    {"weights (288 KB / pair, L2 -> LDS)", 1, 0}, {"x loads (48 KB / pair)", 2, 0}, {"stores, 4 tensors (192 KB / pair)", 4, 4},
    {"stores, 2 tensors (inference form)", 4, 2}, {"loads + 4 stores", 6, 4}, {"weights + loads", 3, 0}, {"weights + 4 stores", 5, 4},
    {"ALL: training form (weights + loads + 4 stores)", 7, 4}, {"ALL: inference form (weights + loads + 2 stores)", 7, 2},
    {"ALL minus xn1: weights + loads + 3 stores", 7, 3}};
  for (auto& c : cases) {
    float best = 1e9f, sum = 0.f;
    for (int r = 0; r < 5; ++r) {
      hipMemsetAsync(big, r, 1L << 30);                 // flush L2 / Infinity Cache between repetitions
      hipEventRecord(e0);
      hipLaunchKernelGGL(probe, dim3(256), dim3(512), LDSB + 80 * 1024, 0, x, wts, o[0], o[1], o[2], o[3], sink, npairs, c.mask, c.nout);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (r) { sum += ms; best = ms < best ? ms : best; }
    }
    const double hbm = ((c.mask & 2) ? T * ROWB : 0) + ((c.mask & 4) ? (double)c.nout * T * ROWB : 0);
    const double l2 = (c.mask & 1) ? (double)npairs * WBYTES : 0;
    printf("%-52s mean %.3f ms (best %.3f)   HBM %.0f MB -> %.2f TB/s   L2->LDS %.0f MB -> %.2f TB/s\n", c.name, sum / 4, best,
           hbm / 1e6, hbm / (sum / 4) / 1e9, l2 / 1e6, l2 / (sum / 4) / 1e9);
  }
  return 0;
}
