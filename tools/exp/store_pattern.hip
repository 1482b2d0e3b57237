// Experiment: HBM write rate of the NT-GEMM epilogue's store pattern (per wave instruction: 16 rows x 64 contiguous
// bytes, rows N*2 bytes apart) against full 128-byte-line and fully contiguous patterns.  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/store_pattern tools/exp/store_pattern.hip && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// C is [M][N] bf16 (2 bytes).  One workgroup of 512 threads = 8 waves (4 x 2) writes a 256 x 192 tile, 8 tiles per WG.
// pattern 0: epilogue layout (lane (fg, fi): row 16u + fi, 16 B at column 32t + 8fg), 12 stores per lane
// pattern 1: same bytes, but every wave instruction writes whole rows of its 64 x 96 patch: lanes run along the row
//            (12 chunks of 16 B per 192-byte patch row -> 5.33 rows per instruction)
// pattern 2: the WORKGROUP writes whole 384-byte tile rows: 24 chunks per row, lanes contiguous over the tile
__global__ __launch_bounds__(512) void k(uint4* C, int M, int N, int pattern, int ntn) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1, fi = lane & 15, fg = lane >> 4;
  const int ntm = M / 256, ntiles = ntm * ntn;
  const uint4 v = make_uint4(tid, lane, wid, 7);
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const long m0 = (long)(t / ntn) * 256; const int n0 = (t % ntn) * 192;
    if (pattern == 0) {
      for (int u = 0; u < 4; ++u)
        for (int tt = 0; tt < 3; ++tt) {
          const long m = m0 + wr * 64 + 16 * u + fi; const int n = n0 + wc * 96 + 32 * tt + 8 * fg;
          C[(m * N + n) >> 3] = v;
        }
    } else if (pattern == 1) {
      for (int i = 0; i < 12; ++i) {
        const int id = i * 64 + lane;            // chunk id in the wave's 64 x 96 patch (12 chunks per row)
        const int r = id / 12, c = id - r * 12;
        const long m = m0 + wr * 64 + r; const int n = n0 + wc * 96 + c * 8;
        C[(m * N + n) >> 3] = v;
      }
    } else {
      for (int i = 0; i < 12; ++i) {
        const int id = i * 512 + tid;            // chunk id in the 256 x 192 tile (24 chunks per row)
        const int r = id / 24, c = id - r * 24;
        C[((m0 + r) * N + n0 + c * 8) >> 3] = v;
      }
    }
  }
}

int main() {
  const int M = 524288;
  for (int N : {192, 768, 1536}) {
    uint4* C; hipMalloc(&C, (size_t)M * N * 2);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int p = 0; p < 3; ++p) {
      k<<<256, 512>>>(C, M, N, p, N / 192);
      hipEventRecord(a);
      for (int i = 0; i < 10; ++i) k<<<256, 512>>>(C, M, N, p, N / 192);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
      printf("N=%4d pattern %d: %8.1f us  %7.0f GB/s\n", N, p, ms * 1e3, (double)M * N * 2 / ms / 1e6);
    }
    hipFree(C);
  }
  return 0;
}
