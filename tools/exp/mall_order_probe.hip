// Does the consumer's walk DIRECTION matter when it reads a tensor the previous kernel has just written?  (Infinity Cache, 256 MiB:
// a producer that writes S MB ascending leaves its LAST ~256 MB resident; a consumer that walks ascending starts with the evicted head
// and pushes the resident tail out before it gets there, a consumer that walks descending starts on resident lines.)
//   hipcc --offload-arch=gfx950 -O3 tools/exp/mall_order_probe.hip -o tools/exp/mall_order_probe && tools/exp/mall_order_probe
// producer: y = f(x) streaming (read S, write S); consumer: reads y (and optionally writes z) in 64 KB chunks, persistent 1024 workgroups,
// chunk order ascending or descending.  Reported: consumer time and its effective read rate, cold (a 1 GiB fill in between) / after the
// producer ascending / after the producer descending-matched.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int CHUNK = 65536;                      // bytes per chunk (256 threads x 16 B x 16)
__global__ __launch_bounds__(256) void producer(const uint4* __restrict__ x, uint4* __restrict__ y, long nchunks, int desc) {
  for (long c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const long cc = desc ? nchunks - 1 - c : c;
    const uint4* s = x + cc * (CHUNK / 16);
    uint4* d = y + cc * (CHUNK / 16);
#pragma unroll
    for (int i = 0; i < 16; ++i) { uint4 v = s[threadIdx.x + 256 * i]; v.x += 1u; d[threadIdx.x + 256 * i] = v; }
  }
}
__global__ __launch_bounds__(256) void consumer(const uint4* __restrict__ y, uint4* __restrict__ z, unsigned* __restrict__ sink, long nchunks, int desc, int wr) {
  unsigned acc = 0;
  for (long c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const long cc = desc ? nchunks - 1 - c : c;
    const uint4* s = y + cc * (CHUNK / 16);
    uint4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = s[threadIdx.x + 256 * i];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += v[i].x ^ v[i].w;
    if (wr) {
      uint4* d = z + cc * (CHUNK / 16);
#pragma unroll
      for (int i = 0; i < 16; ++i) d[threadIdx.x + 256 * i] = v[i];
    }
  }
  if (acc == 0x12345u) sink[blockIdx.x] = acc;
}
__global__ void fill(uint4* p, long n) { for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) p[i] = make_uint4(1, 2, 3, 4); }

int main() {
  const long GiB = 1L << 30;
  uint4 *x, *y, *z, *big; unsigned* sink;
  hipMalloc(&x, GiB); hipMalloc(&y, GiB); hipMalloc(&z, GiB); hipMalloc(&big, GiB); hipMalloc(&sink, 4096 * 4);
  fill<<<2048, 256>>>(x, GiB / 16); fill<<<2048, 256>>>(y, GiB / 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (long mb : {64L, 128L, 201L, 402L, 805L}) {
    const long bytes = mb * 1000000L / CHUNK * CHUNK, nchunks = bytes / CHUNK;
    for (int wr = 0; wr < 2; ++wr) {
      float t[4] = {0, 0, 0, 0};
      const int reps = 5;
      for (int mode = 0; mode < 4; ++mode) {          // 0 cold asc, 1 producer asc -> consumer asc, 2 producer asc -> consumer desc, 3 producer desc -> consumer asc
        for (int r = 0; r < reps; ++r) {
          if (mode == 0) { producer<<<1024, 256>>>(x, y, nchunks, 0); fill<<<2048, 256>>>(big, GiB / 16); }
          else producer<<<1024, 256>>>(x, y, nchunks, mode == 3);
          hipEventRecord(e0);
          consumer<<<1024, 256>>>(y, z, sink, nchunks, mode == 2, wr);
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1); t[mode] += ms / reps;
        }
      }
      const double gb = bytes * (1 + wr) / 1e9;
      printf("%4ld MB, consumer %s: cold %.3f ms (%.2f TB/s) | after producer, same direction %.3f ms (%.2f) | opposite direction %.3f ms (%.2f) | both descending/ascending mix %.3f ms (%.2f)\n",
             mb, wr ? "read+write" : "read only ", t[0], gb / t[0], t[1], gb / t[1], t[2], gb / t[2], t[3], gb / t[3]);
    }
  }
  return 0;
}
