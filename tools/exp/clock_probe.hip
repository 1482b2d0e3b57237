// Calibrates s_memtime / s_memrealtime against wall time and a dependent VALU chain (4 cycles per v_fma on one wave).
// hipcc --offload-arch=gfx950 -O3 tools/exp/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(long long* out, int n) {
  const long long t0 = __builtin_readcyclecounter();
  const long long r0 = __builtin_amdgcn_s_memrealtime();
  float a = threadIdx.x * 1e-9f, b = 1.0000001f;
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 64; ++k) a = __builtin_fmaf(a, b, 1e-9f);
  }
  const long long t1 = __builtin_readcyclecounter();
  const long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[blockIdx.x * 4] = t1 - t0; out[blockIdx.x * 4 + 1] = r1 - r0; out[blockIdx.x * 4 + 2] = (long long)a; }
}
int main() {
  long long* d; hipMalloc(&d, 4096 * 32);
  for (int blocks : {1, 1024}) {
    const int n = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), 0, 0, d, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[4]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    const double fmas = (double)n * 64;
    printf("blocks %d: wall %.3f ms; s_memtime delta %lld (%.1f MHz); s_memrealtime delta %lld (%.1f MHz); %.2f ns per dependent fma\n",
           blocks, ms, h[0], h[0] / (ms * 1e3), h[1], h[1] / (ms * 1e3), ms * 1e6 / fmas);
  }
  return 0;
}
