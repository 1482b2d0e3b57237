// Fused optimizer step for the training loop around the hot path: one streaming pass
// over the engine's flat f32 buffers does what the reference spreads over
//   torch.optim.SGD(momentum, nesterov=True) with the two weight-decay groups of basics/optimizer.py:35-49
//     (Train.py:145-150, :448-450: scaler.step(optimizer)),
//   ModelEMA.update  (basics/utils/torch_utils.py:291-301: a Python loop over 273 state_dict tensors), and
//   the cast of the updated f32 masters to the run dtype the GEMM kernels read
// into a single launch: p, g, momentum and EMA are read once and written once, 16 bytes per lane.
//
//   d   = g * grad_scale + wd[group] * p              (torch.optim.SGD: weight decay added to the gradient)
//   m'  = momentum[group] * m + d                     (dampening 0; the first step's "buf = d" is m = 0)
//   u   = nesterov ? d + momentum[group] * m' : m'
//   p'  = p - lr[group] * u
//   e'  = e * ema_decay + (1 - ema_decay) * p'        (skipped when ema == NULL)
//   p16 = (run dtype) p'                              (skipped when p_cast == NULL)
//
// Parameters are padded to multiples of four elements in the flat layout (engine.py), so a 16-byte chunk belongs to one
// parameter and `group_of_chunk` (one byte per chunk) selects its hyper-parameter group.  HBM-bound: 4 x 4 B read +
// 3 x 4 B + 2 B written per element.
#include "common.h"
#include "../../include/sodt_hip.h"

namespace {

struct OptHyp { float lr[4], momentum[4], wd[4]; float grad_scale, ema_decay; int nesterov; };

template <typename TC>
__global__ __launch_bounds__(256) void sgd_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                     float* __restrict__ e, TC* __restrict__ pc,
                                                     const unsigned char* __restrict__ group, long nchunk, const OptHyp h) {
  for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < nchunk; c += (long)gridDim.x * 256) {
    const int gi = group ? (int)group[c] : 0;
    const float lr = h.lr[gi & 3], mu = h.momentum[gi & 3], wd = h.wd[gi & 3];
    float4 pv = ((const float4*)p)[c];
    if (gi < 4) {       // 255 marks padding / frozen parameters: cast only
      const float4 gv = ((const float4*)g)[c];
      float4 mv = ((const float4*)m)[c];
      float pa[4] = {pv.x, pv.y, pv.z, pv.w}, ga[4] = {gv.x, gv.y, gv.z, gv.w}, ma[4] = {mv.x, mv.y, mv.z, mv.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = fmaf(wd, pa[j], ga[j] * h.grad_scale);
        ma[j] = fmaf(mu, ma[j], d);
        const float u = h.nesterov ? fmaf(mu, ma[j], d) : ma[j];
        pa[j] = fmaf(-lr, u, pa[j]);
      }
      pv = make_float4(pa[0], pa[1], pa[2], pa[3]);
      ((float4*)p)[c] = pv;
      ((float4*)m)[c] = make_float4(ma[0], ma[1], ma[2], ma[3]);
    }
    if (e) {
      float4 ev = ((const float4*)e)[c];
      const float a = h.ema_decay, b1 = 1.0f - h.ema_decay;
      ev.x = fmaf(ev.x, a, b1 * pv.x); ev.y = fmaf(ev.y, a, b1 * pv.y); ev.z = fmaf(ev.z, a, b1 * pv.z); ev.w = fmaf(ev.w, a, b1 * pv.w);
      ((float4*)e)[c] = ev;
    }
    if (pc) {
      if constexpr (sizeof(TC) == 2) ((uint2*)pc)[c] = make_uint2(pack2bf(pv.x, pv.y), pack2bf(pv.z, pv.w));
      else ((float4*)pc)[c] = pv;
    }
  }
}

}  // namespace

extern "C" int sodt_sgd_ema_step(float* p, const float* g, float* mom, float* ema, void* p_cast, int cast_dtype,
                                 const unsigned char* group_of_chunk, long n_elems, int ngroups, const float* lr,
                                 const float* momentum, const float* weight_decay, int nesterov, float grad_scale,
                                 float ema_decay, sodt_stream_t st) {
  if (!p || !g || !mom || n_elems <= 0 || (n_elems & 3) || ngroups < 1 || ngroups > 4 || !lr || !momentum || !weight_decay)
    return SODT_EINVAL;
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)mom | (uintptr_t)ema | (uintptr_t)p_cast) & 15) return SODT_EINVAL;
  OptHyp h;
  for (int i = 0; i < 4; ++i) {
    const int j = i < ngroups ? i : 0;
    h.lr[i] = lr[j]; h.momentum[i] = momentum[j]; h.wd[i] = weight_decay[j];
  }
  h.grad_scale = grad_scale; h.ema_decay = ema_decay; h.nesterov = nesterov;
  const long nchunk = n_elems >> 2;
  long nb = (nchunk + 255) / 256;
  if (nb > 8192) nb = 8192;
  hipStream_t s = (hipStream_t)st;
  if (p_cast && cast_dtype == SODT_BF16)
    hipLaunchKernelGGL(sgd_ema_kernel<bf16>, dim3((unsigned)nb), dim3(256), 0, s, p, g, mom, ema, (bf16*)p_cast, group_of_chunk, nchunk, h);
  else if (!p_cast || cast_dtype == SODT_F32)
    hipLaunchKernelGGL(sgd_ema_kernel<float>, dim3((unsigned)nb), dim3(256), 0, s, p, g, mom, ema, (float*)p_cast, group_of_chunk, nchunk, h);
  else
    return SODT_EINVAL;
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}
