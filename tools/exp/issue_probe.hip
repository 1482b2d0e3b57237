// Vector-instruction issue rates of one SIMD on gfx950, measured (round 6): how many cycles per wave64 instruction does a SIMD need
// with ONE wave and with TWO waves resident, for the instruction classes the fused block kernel is made of, and what happens when one
// wave of a SIMD streams MFMAs while its partner streams VALU work (the pairing the dephased schedule of wmsa_hg.hip creates).
//   hipcc --offload-arch=gfx950 -O3 tools/exp/issue_probe.hip -o /tmp/issue_probe && /tmp/issue_probe
// One workgroup per CU (256 CUs), W waves per workgroup: waves w and w + 4 share SIMD w & 3.  Every wave runs `iters` trips of an
// unrolled body of 64 independent instructions (8 accumulators round robin) and stamps s_memtime around it; wave 0 and wave 4 report.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

enum { OP_FMA = 0, OP_EXP = 1, OP_CVT = 2, OP_PKFMA = 3, OP_MFMA = 4, OP_NONE = 5, OP_AND = 6 };

template <int OP> __device__ __forceinline__ void body(float (&a)[8], f32x4 (&acc)[8], float b, float c, bf16x8 fa, bf16x8 fb) {
#pragma unroll
  for (int r = 0; r < 8; ++r) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if constexpr (OP == OP_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      if constexpr (OP == OP_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if constexpr (OP == OP_AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if constexpr (OP == OP_PKFMA) {
        typedef __attribute__((ext_vector_type(2))) float f2;
        f2 v = {a[i], a[(i + 1) & 7]};
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(f2{b, b}), "v"(f2{c, c}));
        a[i] = v[0];
      }
      if constexpr (OP == OP_MFMA) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[i], 0, 0, 0);
    }
  }
}

// waves 0..3 run OPA, waves 4..7 (when present) run OPB
template <int OPA, int OPB> __global__ __launch_bounds__(512) void k(float* out, long long* cyc, int itersA, int itersB) {
  float a[8];
  f32x4 acc[8];
  bf16x8 fa, fb;
  for (int i = 0; i < 8; ++i) { a[i] = 1e-3f * (float)(threadIdx.x + i); acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; fa[i] = (__bf16)(0.001f * (float)((threadIdx.x + i) & 15)); fb[i] = (__bf16)(0.01f * (float)i); }
  const float b = 0.999f, c = 1e-6f;
  const int w = threadIdx.x >> 6;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  if (w < 4) { for (int it = 0; it < itersA; ++it) body<OPA>(a, acc, b, c, fa, fb); }
  else { for (int it = 0; it < itersB; ++it) body<OPB>(a, acc, b, c, fa, fb); }
  asm volatile("s_nop 0" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += a[i] + acc[i][0] + acc[i][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
}

static float* out; static long long* cyc;
template <int OPA, int OPB> void run(const char* name, int waves, int iters, int mulB = 1, int mulA = 1) {
  long long h[8 * 256];
  hipLaunchKernelGGL((k<OPA, OPB>), dim3(256), dim3(64 * waves), 0, 0, out, cyc, 10, 10);
  hipDeviceSynchronize();
  hipMemset(cyc, 0, 256 * 8 * 8);
  hipLaunchKernelGGL((k<OPA, OPB>), dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters * mulA, iters * mulB);
  hipDeviceSynchronize();
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double a0 = 0, a4 = 0;
  for (int bl = 0; bl < 256; ++bl) { a0 += h[bl * 8]; a4 += h[bl * 8 + 4]; }
  const double n = 64.0 * iters;
  printf("%-46s waves/CU %d: wave 0: %6.2f cycles per instruction", name, waves, a0 / 256 / (n * mulA));
  if (waves > 4) printf("   wave 4: %6.2f", a4 / 256 / (n * mulB));
  printf("\n");
}

int main() {
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  const int it = 4000;
  run<OP_FMA, OP_NONE>("v_fma_f32 alone", 4, it);
  run<OP_FMA, OP_FMA>("v_fma_f32 + v_fma_f32 (two waves per SIMD)", 8, it);
  run<OP_AND, OP_AND>("v_and_b32 + v_and_b32", 8, it);
  run<OP_EXP, OP_NONE>("v_exp_f32 alone", 4, it);
  run<OP_EXP, OP_EXP>("v_exp_f32 + v_exp_f32", 8, it);
  run<OP_CVT, OP_NONE>("v_cvt_pk_bf16_f32 alone", 4, it);
  run<OP_CVT, OP_CVT>("v_cvt_pk_bf16_f32 + same", 8, it);
  run<OP_PKFMA, OP_NONE>("v_pk_fma_f32 alone", 4, it);
  run<OP_PKFMA, OP_PKFMA>("v_pk_fma_f32 + same", 8, it);
  run<OP_MFMA, OP_NONE>("mfma 16x16x32 bf16 alone", 4, it);
  run<OP_MFMA, OP_MFMA>("mfma + mfma", 8, it);
  run<OP_MFMA, OP_FMA>("wave 0 mfma | wave 4 v_fma_f32 (x4 trips)", 8, it, 4);
  run<OP_MFMA, OP_EXP>("wave 0 mfma | wave 4 v_exp_f32 (x2 trips)", 8, it, 2);
  run<OP_FMA, OP_MFMA>("wave 0 v_fma_f32 (x4 trips) | wave 4 mfma", 8, it, 1, 4);
  run<OP_FMA, OP_EXP>("wave 0 v_fma_f32 (x2 trips) | wave 4 v_exp_f32", 8, it, 1, 2);
  return 0;
}
