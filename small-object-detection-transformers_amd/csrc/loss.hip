// YOLOv5 ComputeLoss + build_targets of the reference's training step (basics/utils/loss.py:116-224) with the CIoU of
// basics/utils/general.py:347-389, for the single detection layer of models/model.yaml, as device kernels: loss values AND
// the gradient with respect to the head output in one pass of four small launches, no host synchronisation, no
// autograd graph (the reference builds ~60 tiny ATen kernels and a .item()-free but allocation-heavy index_put chain).
//
//   candidates  one thread per (offset o, anchor a, target i) in the reference's order o*(na*nt) + a*nt + i:
//               anchor-ratio test (:190-194), neighbour-cell test (:200-206), cell (:216-220), CIoU of the decoded
//               prediction against the target box with its gradient by forward-mode differentiation (4 partials carried
//               through general.py:353-385; alpha is a constant, as under torch.no_grad), class BCE and its gradient;
//               box / class loss sums and the match count by double atomics; the LAST candidate of a cell wins the
//               objectness target (index_put semantics of `tobj[b, a, gj, gi] = ...` on the CPU reference: atomicMax of
//               the candidate's order key).
//   obj_dense   every cell: objectness target from the winning candidate's IoU, BCE and its gradient; writes the whole
//               dpred row (zeros except channel 4).
//   scatter     box and class gradients of every valid candidate added (atomics: duplicates accumulate, as autograd's
//               index backward does), scaled by 1/n known only now.
//   finalize    (loss * batch, lbox, lobj, lcls) exactly as loss.py:157-163.
// Latency-bound (a few hundred candidates), except obj_dense which streams pred / dpred once: 2 x B*na*ny*nx*no*4 bytes.
#include "common.h"
#include "../../include/sodt_hip.h"

namespace {

constexpr int LOSS_MAX_NC = 32;

struct LossArgs {
  const float* pred; const float* targets; const float* anchors;
  float* dpred; float* out;
  int* winner; float* rec; double* sums;       // rec: [ncand][6 + nc]: cell, valid, iou, g0..g3, gcls[nc]; sums: box, cls, obj, n
  int B, na, ny, nx, nc, no, nt;
  float h_box, h_cls, cls_pw, h_obj, obj_pw, anchor_t, gr;
};

struct D4 { float v, d[4]; };
__device__ __forceinline__ D4 dc(float c) { return D4{c, {0.f, 0.f, 0.f, 0.f}}; }
__device__ __forceinline__ D4 operator+(const D4& a, const D4& b) { D4 r; r.v = a.v + b.v; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
__device__ __forceinline__ D4 operator-(const D4& a, const D4& b) { D4 r; r.v = a.v - b.v; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
__device__ __forceinline__ D4 operator*(const D4& a, const D4& b) { D4 r; r.v = a.v * b.v; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
__device__ __forceinline__ D4 operator/(const D4& a, const D4& b) {
  D4 r; const float ib = 1.0f / b.v; r.v = a.v * ib;
  for (int i = 0; i < 4; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib;
  return r;
}
__device__ __forceinline__ D4 scl(const D4& a, float s) { D4 r; r.v = a.v * s; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * s; return r; }
__device__ __forceinline__ D4 addc(const D4& a, float c) { D4 r = a; r.v += c; return r; }
__device__ __forceinline__ D4 dmin(const D4& a, const D4& b) { return a.v <= b.v ? a : b; }      // torch.min gradient: the smaller operand
__device__ __forceinline__ D4 dmax(const D4& a, const D4& b) { return a.v >= b.v ? a : b; }
__device__ __forceinline__ D4 clamp0(const D4& a) { return a.v > 0.f ? a : dc(0.f); }
__device__ __forceinline__ D4 datan(const D4& a) { D4 r; r.v = atanf(a.v); const float k = 1.0f / (1.0f + a.v * a.v); for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * k; return r; }
__device__ __forceinline__ float softplus(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(128) void loss_candidates_kernel(const LossArgs a) {
  const int c = blockIdx.x * 128 + threadIdx.x;
  const int ncand = 5 * a.na * a.nt;
  if (c >= ncand) return;
  const int RS = 7 + a.nc;
  float* rec = a.rec + (long)c * RS;
  rec[1] = 0.f;
  const int o = c / (a.na * a.nt), an = (c / a.nt) % a.na, i = c % a.nt;
  const float* t = a.targets + 6 * i;
  const float gx = t[2] * (float)a.nx, gy = t[3] * (float)a.ny, gw = t[4] * (float)a.nx, gh = t[5] * (float)a.ny;   // loss.py:187
  const float aw = a.anchors[2 * an], ah = a.anchors[2 * an + 1];
  const float rw = gw / aw, rh = gh / ah;
  if (!(fmaxf(fmaxf(rw, 1.0f / rw), fmaxf(rh, 1.0f / rh)) < a.anchor_t)) return;                                       // :192-194
  float ox = 0.f, oy = 0.f;
  if (o == 1) { if (!(fmodf(gx, 1.0f) < 0.5f && gx > 1.0f)) return; ox = 0.5f; }                                        // :203-206
  else if (o == 2) { if (!(fmodf(gy, 1.0f) < 0.5f && gy > 1.0f)) return; oy = 0.5f; }
  else if (o == 3) { const float q = (float)a.nx - gx; if (!(fmodf(q, 1.0f) < 0.5f && q > 1.0f)) return; ox = -0.5f; }
  else if (o == 4) { const float q = (float)a.ny - gy; if (!(fmodf(q, 1.0f) < 0.5f && q > 1.0f)) return; oy = -0.5f; }
  const int b = (int)t[0], cls = (int)t[1];
  const int gix = (int)(gx - ox), giy = (int)(gy - oy);                                                               // .long(): truncation
  const int gi = min(max(gix, 0), a.nx - 1), gj = min(max(giy, 0), a.ny - 1);
  if (b < 0 || b >= a.B) return;
  const int cell = ((b * a.na + an) * a.ny + gj) * a.nx + gi;
  const float* ps = a.pred + (long)cell * a.no;
  // ---- decoded prediction (loss.py:130-132) and CIoU (general.py:353-385) as dual numbers in ps[0..3]
  D4 bx, by, bw, bh;
  {
    const float s0 = sigm(ps[0]), s1 = sigm(ps[1]), s2 = sigm(ps[2]), s3 = sigm(ps[3]);
    bx = dc(s0 * 2.f - 0.5f); bx.d[0] = 2.f * s0 * (1.f - s0);
    by = dc(s1 * 2.f - 0.5f); by.d[1] = 2.f * s1 * (1.f - s1);
    bw = dc(4.f * s2 * s2 * aw); bw.d[2] = 8.f * s2 * aw * s2 * (1.f - s2);
    bh = dc(4.f * s3 * s3 * ah); bh.d[3] = 8.f * s3 * ah * s3 * (1.f - s3);
  }
  const float tx = gx - (float)gi, ty = gy - (float)gj;      // tbox (:221) uses the cell index AFTER the in-place clamp of :219
  const float eps = 1e-7f;
  const D4 b1x1 = bx - scl(bw, 0.5f), b1x2 = bx + scl(bw, 0.5f), b1y1 = by - scl(bh, 0.5f), b1y2 = by + scl(bh, 0.5f);
  const D4 b2x1 = dc(tx - gw / 2), b2x2 = dc(tx + gw / 2), b2y1 = dc(ty - gh / 2), b2y2 = dc(ty + gh / 2);
  const D4 inter = clamp0(dmin(b1x2, b2x2) - dmax(b1x1, b2x1)) * clamp0(dmin(b1y2, b2y2) - dmax(b1y1, b2y1));
  const D4 w1 = b1x2 - b1x1, h1 = addc(b1y2 - b1y1, eps);
  const D4 w2 = b2x2 - b2x1, h2 = addc(b2y2 - b2y1, eps);
  const D4 uni = addc(w1 * h1 + w2 * h2 - inter, eps);
  const D4 iou = inter / uni;
  const D4 cw = dmax(b1x2, b2x2) - dmin(b1x1, b2x1), ch = dmax(b1y2, b2y2) - dmin(b1y1, b2y1);
  const D4 c2 = addc(cw * cw + ch * ch, eps);
  const D4 dxs = b2x1 + b2x2 - b1x1 - b1x2, dys = b2y1 + b2y2 - b1y1 - b1y2;
  const D4 rho2 = scl(dxs * dxs + dys * dys, 0.25f);
  const D4 da = datan(w2 / h2) - datan(w1 / h1);
  const D4 v = scl(da * da, 0.40528473456935109f);                                                                   // 4 / pi^2
  const float alpha = v.v / (v.v - iou.v + (1.0f + eps));                                                             // no_grad
  const D4 ciou = iou - (rho2 / c2 + scl(v, alpha));
  rec[0] = __int_as_float(cell);
  rec[1] = 1.f;
  rec[2] = ciou.v;
  for (int k = 0; k < 4; ++k) rec[3 + k] = -ciou.d[k];                 // d(1 - iou) / d ps[k]
  double cls_sum = 0.0;
  if (a.nc > 1) {
    for (int k = 0; k < a.nc; ++k) {                                   // BCEWithLogits(pos_weight) (:142-144)
      const float x = ps[5 + k], tk = k == cls ? 1.f : 0.f, s = sigm(x);
      cls_sum += (double)(a.cls_pw * tk * softplus(-x) + (1.f - tk) * softplus(x));
      rec[7 + k] = -a.cls_pw * tk * (1.f - s) + (1.f - tk) * s;
    }
  }
  atomicAdd(&a.sums[0], (double)(1.0f - ciou.v));
  atomicAdd(&a.sums[1], cls_sum);
  atomicAdd(&a.sums[3], 1.0);
  atomicMax(&a.winner[cell], c);
}

__global__ __launch_bounds__(256) void loss_obj_dense_kernel(const LossArgs a, long ncells) {
  const int RS = 7 + a.nc;
  const float gsc = a.h_obj * 4.0f * (float)a.B / (float)ncells;         // d(loss * bs) / d BCEobj-mean, balance[0] = 4 (loss.py:110)
  double part = 0.0;
  for (long cell = (long)blockIdx.x * 256 + threadIdx.x; cell < ncells; cell += (long)gridDim.x * 256) {
    const float x = a.pred[cell * a.no + 4];
    const int w = a.winner[cell];
    float tobj = 0.f;
    if (w >= 0) tobj = (1.0f - a.gr) + a.gr * fmaxf(a.rec[(long)w * RS + 2], 0.f);      // :137
    const float s = sigm(x);
    part += (double)(a.obj_pw * tobj * softplus(-x) + (1.f - tobj) * softplus(x));
    float* d = a.dpred + cell * a.no;
    for (int k = 0; k < a.no; ++k) d[k] = 0.f;
    d[4] = (-a.obj_pw * tobj * (1.f - s) + (1.f - tobj) * s) * gsc;
  }
  __shared__ double red[256];
  red[threadIdx.x] = part;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(&a.sums[2], red[0]);
}

__global__ __launch_bounds__(128) void loss_scatter_kernel(const LossArgs a) {
  const int c = blockIdx.x * 128 + threadIdx.x;
  if (c >= 5 * a.na * a.nt) return;
  const int RS = 7 + a.nc;
  const float* rec = a.rec + (long)c * RS;
  if (rec[1] == 0.f) return;
  const double n = a.sums[3];
  const float sb = a.h_box * (float)a.B / (float)n, sc = a.h_cls * (float)a.B / (float)(n * a.nc);
  float* d = a.dpred + (long)__float_as_int(rec[0]) * a.no;
  for (int k = 0; k < 4; ++k) atomicAdd(d + k, rec[3 + k] * sb);
  if (a.nc > 1)
    for (int k = 0; k < a.nc; ++k) atomicAdd(d + 5 + k, rec[7 + k] * sc);
}

__global__ void loss_finalize_kernel(const LossArgs a, long ncells) {
  const double n = a.sums[3];
  const float lbox = n > 0 ? (float)(a.sums[0] / n) * a.h_box : 0.f;
  const float lcls = (n > 0 && a.nc > 1) ? (float)(a.sums[1] / (n * a.nc)) * a.h_cls : 0.f;
  const float lobj = (float)(a.sums[2] / (double)ncells) * 4.0f * a.h_obj;
  a.out[0] = (lbox + lobj + lcls) * (float)a.B;
  a.out[1] = lbox; a.out[2] = lobj; a.out[3] = lcls;
}

size_t loss_ws_bytes(long ncells, int nt, int nc) {
  const size_t win = ((size_t)ncells * 4 + 255) & ~(size_t)255;
  const size_t rec = (((size_t)5 * 8 * (nt > 0 ? nt : 1) * (7 + nc)) * 4 + 255) & ~(size_t)255;      // na <= 8
  return win + rec + 256;
}

}  // namespace

extern "C" int sodt_yolo_loss_workspace_bytes(long ncells, int nt, int nc, size_t* bytes) {
  if (!bytes || ncells <= 0 || nt < 0 || nc < 1 || nc > LOSS_MAX_NC) return SODT_EINVAL;
  *bytes = loss_ws_bytes(ncells, nt, nc);
  return SODT_OK;
}

extern "C" int sodt_yolo_loss(const float* pred, const float* targets, int nt, const float* anchors, int B, int na, int ny, int nx,
                              int nc, float h_box, float h_cls, float cls_pw, float h_obj, float obj_pw, float anchor_t, float gr,
                              void* ws, size_t ws_bytes, float* dpred, float* out4, sodt_stream_t st_) {
  if (!pred || !anchors || !ws || !dpred || !out4 || B <= 0 || na <= 0 || na > 8 || ny <= 0 || nx <= 0 || nc < 1 || nc > LOSS_MAX_NC ||
      nt < 0 || (nt > 0 && !targets))
    return SODT_EINVAL;
  const long ncells = (long)B * na * ny * nx;
  if (ws_bytes < loss_ws_bytes(ncells, nt, nc) || (long)5 * na * nt > (1L << 30)) return SODT_EINVAL;
  hipStream_t st = (hipStream_t)st_;
  LossArgs a;
  a.pred = pred; a.targets = targets; a.anchors = anchors; a.dpred = dpred; a.out = out4;
  unsigned char* w = (unsigned char*)ws;
  const size_t win = ((size_t)ncells * 4 + 255) & ~(size_t)255;
  a.winner = (int*)w;
  a.sums = (double*)(w + win);
  a.rec = (float*)(w + win + 256);
  a.B = B; a.na = na; a.ny = ny; a.nx = nx; a.nc = nc; a.no = nc + 5; a.nt = nt;
  a.h_box = h_box; a.h_cls = h_cls; a.cls_pw = cls_pw; a.h_obj = h_obj; a.obj_pw = obj_pw; a.anchor_t = anchor_t; a.gr = gr;
  if (hipMemsetAsync(w, 0xff, win, st) != hipSuccess) return SODT_EINVAL;          // winner = -1
  if (hipMemsetAsync(w + win, 0, 256, st) != hipSuccess) return SODT_EINVAL;       // sums = 0
  const int ncand = 5 * na * nt;
  if (ncand > 0) hipLaunchKernelGGL(loss_candidates_kernel, dim3((ncand + 127) / 128), dim3(128), 0, st, a);
  long nb = (ncells + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(loss_obj_dense_kernel, dim3((unsigned)nb), dim3(256), 0, st, a, ncells);
  if (ncand > 0) hipLaunchKernelGGL(loss_scatter_kernel, dim3((ncand + 127) / 128), dim3(128), 0, st, a);
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1), 0, st, a, ncells);
  return hipGetLastError() == hipSuccess ? SODT_OK : SODT_EINVAL;
}
