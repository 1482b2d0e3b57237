// wall-clock rate of v_mfma_f32_16x16x16_bf16 vs v_mfma_f32_16x16x32_bf16 (one wave per SIMD, 8 independent accumulators)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
template <int K32> __global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  s16x4 a4, b4;
  bf16x8 a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(0.001f * (float)((threadIdx.x + i) & 15)); b8[i] = (__bf16)(0.01f * (float)(i)); }
  for (int i = 0; i < 4; ++i) { a4[i] = (short)(0x3c00 + threadIdx.x); b4[i] = (short)(0x3b00 + i); }
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (K32) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  const int iters = 40000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    long long h[256]; float ms;
    hipEventRecord(e0); hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, cyc, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("16x16x16 bf16: %.2f memtime ticks per MFMA, wall %.3f ms -> %.1f ns per MFMA per SIMD, %.0f TF/s\n", (double)h[7] / (8.0 * iters), ms,
           ms * 1e6 / (8.0 * iters), 256.0 * 4 * 8 * iters * 8192.0 / (ms * 1e-3) / 1e12);
    hipEventRecord(e0); hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, cyc, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("16x16x32 bf16: %.2f memtime ticks per MFMA, wall %.3f ms -> %.1f ns per MFMA per SIMD, %.0f TF/s\n", (double)h[7] / (8.0 * iters), ms,
           ms * 1e6 / (8.0 * iters), 256.0 * 4 * 8 * iters * 16384.0 / (ms * 1e-3) / 1e12);
  }
  return 0;
}
