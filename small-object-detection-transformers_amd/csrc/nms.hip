// Eval-side post-processing on the GPU: non_max_suppression of general.py:425-512 (candidate selection,
// class-offset batched NMS, max_det cut, merge-NMS with the redundancy filter) for one image per call.
//
// Bit-exactness: every comparison the reference makes (obj > conf, obj*cls > conf, IoU > thr) is evaluated with
// the same f32 operations in the same order - no FMA contraction in this file - so the kept candidate indices are
// identical to the oracle's; only the merged box coordinates (an f32 matmul in the reference) carry a tolerance.
//
// Pipeline (all on the caller's stream):
//   nms_candidates : one thread per decoded row; emits a 64-bit sort key per passing (box, class) pair,
//                    key = (~score_bits << 32) | (box * nc + class)  -> ascending key order == descending score,
//                    ties in the reference's row-major candidate order (== a stable sort of the reference's list)
//   rocprim radix sort of the keys; the first min(n, 30000) survive (general.py:489-490)
//   nms_gather     : rebuild [x1 y1 x2 y2 score cls] and the class-offset boxes from the sorted keys
//   nms_mask       : 64x64-tiled upper-triangular IoU > thr bit matrix
//   nms_reduce     : one workgroup walks the matrix 64 rows at a time (in-register resolve of the diagonal word,
//                    then a parallel OR of the kept rows), stops at max_det
//   nms_merge      : one wave per kept detection: weighted mean of all boxes with IoU > thr (general.py:500-506)
//   nms_compact    : drop non-redundant detections, write (count, rows)
#pragma clang fp contract(off)
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include "common.h"
#include "../../include/sodt_hip.h"

namespace {

constexpr int MAX_NMS = 30000;   // general.py:438
constexpr int MAX_DET = 300;     // general.py:437
constexpr float MAX_WH = 4096.f; // general.py:436

__device__ __forceinline__ uint64_t make_key(float score, uint32_t id) {
  return ((uint64_t)(~__float_as_uint(score)) << 32) | id;   // score > 0 here, so its bit pattern is monotonic
}

__global__ __launch_bounds__(256) void nms_candidates_kernel(const float* __restrict__ z, int N, int nc, float conf,
                                                            int multi, const unsigned char* __restrict__ allow,
                                                            uint64_t* __restrict__ keys, int cap, int* __restrict__ count) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const float* r = z + (long)i * (nc + 5);
  const float obj = r[4];
  if (!(obj > conf)) return;
  if (multi) {
    for (int j = 0; j < nc; ++j) {
      const float s = r[5 + j] * obj;
      if (s > conf && (!allow || allow[j])) {
        const int slot = atomicAdd(count, 1);
        if (slot < cap) keys[slot] = make_key(s, (uint32_t)(i * nc + j));
      }
    }
  } else {
    float best = r[5] * obj; int bj = 0;
    for (int j = 1; j < nc; ++j) {
      const float s = r[5 + j] * obj;
      if (s > best) { best = s; bj = j; }
    }
    if (best > conf && (!allow || allow[bj])) {
      const int slot = atomicAdd(count, 1);
      if (slot < cap) keys[slot] = make_key(best, (uint32_t)(i * nc + bj));
    }
  }
}

__global__ __launch_bounds__(256) void nms_gather_kernel(const float* __restrict__ z, const uint64_t* __restrict__ keys, int n,
                                                        int nc, int agnostic, float* __restrict__ det, float4* __restrict__ boxes) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= n) return;
  const uint32_t id = (uint32_t)(keys[s] & 0xffffffffu);
  const int i = id / nc, j = id - i * nc;
  const float* r = z + (long)i * (nc + 5);
  const float x = r[0], y = r[1], hw = r[2] / 2, hh = r[3] / 2;
  const float x1 = x - hw, y1 = y - hh, x2 = x + hw, y2 = y + hh;
  const float score = r[5 + j] * r[4];
  float* d = det + (long)s * 6;
  d[0] = x1; d[1] = y1; d[2] = x2; d[3] = y2; d[4] = score; d[5] = (float)j;
  const float c = agnostic ? 0.f : (float)j * MAX_WH;
  boxes[s] = make_float4(x1 + c, y1 + c, x2 + c, y2 + c);
}

__device__ __forceinline__ bool iou_gt(const float4 a, const float4 b, float thr) {
  const float aa = (a.z - a.x) * (a.w - a.y), ab = (b.z - b.x) * (b.w - b.y);
  const float w = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.f);
  const float h = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.f);
  const float inter = w * h;
  return inter / (aa + ab - inter) > thr;
}

// mask[i * W + cb] bit t  <=>  IoU(box i, box cb*64+t) > thr, only for cb*64+t > i
__global__ __launch_bounds__(64) void nms_mask_kernel(const float4* __restrict__ boxes, int n, int W, float thr,
                                                     uint64_t* __restrict__ mask) {
  const int rb = blockIdx.y, cb = blockIdx.x;
  if (cb < rb) return;
  __shared__ float4 sb[64];
  const int t = threadIdx.x;
  const int cj = cb * 64 + t;
  sb[t] = cj < n ? boxes[cj] : make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  const int i = rb * 64 + t;
  if (i >= n) return;
  const float4 a = boxes[i];
  const int ncol = min(64, n - cb * 64);
  uint64_t bits = 0;
  for (int q = (cb == rb ? t + 1 : 0); q < ncol; ++q)
    if (iou_gt(a, sb[q], thr)) bits |= 1ull << q;
  mask[(long)i * W + cb] = bits;
}

__global__ __launch_bounds__(512) void nms_reduce_kernel(const uint64_t* __restrict__ mask, int n, int W, int max_det,
                                                        int* __restrict__ keep, int* __restrict__ nkeep) {
  __shared__ uint64_t remv[512];
  __shared__ int s_kept[64];
  __shared__ int s_nk, s_total;
  const int tid = threadIdx.x;
  remv[tid] = 0;
  if (tid == 0) s_total = 0;
  __syncthreads();
  for (int kb = 0; kb < W; ++kb) {
    if (tid < 64) {
      const int row = kb * 64 + tid;
      const uint64_t diag = row < n ? mask[(long)row * W + kb] : 0;
      uint64_t rem = remv[kb];
      const int valid = min(64, n - kb * 64);
      uint64_t kept = 0;
      for (int r = 0; r < valid; ++r) {
        const uint64_t d = __shfl(diag, r);
        if (!((rem >> r) & 1)) { kept |= 1ull << r; rem |= d; }
      }
      if (tid == 0) {
        int total = s_total, nk = 0;
        for (int r = 0; r < valid && total < max_det; ++r)
          if ((kept >> r) & 1) { keep[total++] = kb * 64 + r; s_kept[nk++] = r; }
        s_total = total; s_nk = nk;
      }
    }
    __syncthreads();
    if (s_total >= max_det) break;
    const int nk = s_nk;
    if (tid > kb && tid < W) {
      uint64_t acc = 0;
      for (int q = 0; q < nk; ++q) acc |= mask[(long)(kb * 64 + s_kept[q]) * W + tid];
      remv[tid] |= acc;
    }
    __syncthreads();
  }
  if (tid == 0) *nkeep = s_total;
}

// general.py:500-506: boxes(i) = sum_j w_ij * box_j / sum_j w_ij, w_ij = [IoU(i, j) > thr] * score_j; redundant: > 1 member
__global__ __launch_bounds__(64) void nms_merge_kernel(const float* __restrict__ det, const float4* __restrict__ boxes, int n,
                                                      const int* __restrict__ keep, const int* __restrict__ nkeep, float thr,
                                                      int merge, float* __restrict__ rows, int* __restrict__ valid) {
  const int k = blockIdx.x;
  if (k >= *nkeep) return;
  const int i = keep[k], lane = threadIdx.x;
  float* o = rows + (long)k * 6;
  const float* di = det + (long)i * 6;
  if (!merge) {
    if (lane < 6) o[lane] = di[lane];
    if (lane == 0) valid[k] = 1;
    return;
  }
  const float4 a = boxes[i];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, ws = 0.f; int cnt = 0;
  for (int j = lane; j < n; j += 64) {
    if (iou_gt(a, boxes[j], thr)) {
      const float* dj = det + (long)j * 6;
      const float w = dj[4];
      s0 += w * dj[0]; s1 += w * dj[1]; s2 += w * dj[2]; s3 += w * dj[3]; ws += w; ++cnt;
    }
  }
  s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3); ws = wave_sum(ws);
  cnt += __shfl_xor(cnt, 1); cnt += __shfl_xor(cnt, 2); cnt += __shfl_xor(cnt, 4);
  cnt += __shfl_xor(cnt, 8); cnt += __shfl_xor(cnt, 16); cnt += __shfl_xor(cnt, 32);
  if (lane == 0) {
    o[0] = s0 / ws; o[1] = s1 / ws; o[2] = s2 / ws; o[3] = s3 / ws; o[4] = di[4]; o[5] = di[5];
    valid[k] = cnt > 1;
  }
}

__global__ __launch_bounds__(64) void nms_compact_kernel(const float* __restrict__ rows, const int* __restrict__ valid,
                                                        const int* __restrict__ keep, const uint64_t* __restrict__ keys,
                                                        const int* __restrict__ nkeep, float* __restrict__ out,
                                                        int* __restrict__ out_index, int* __restrict__ out_count) {
  const int nk = *nkeep, lane = threadIdx.x;
  int base = 0;
  for (int k0 = 0; k0 < nk; k0 += 64) {
    const int k = k0 + lane;
    const bool v = k < nk && valid[k];
    const uint64_t b = __ballot(v);
    const int pos = base + __popcll(b & ((1ull << lane) - 1));
    if (v) {
      for (int c = 0; c < 6; ++c) out[(long)pos * 6 + c] = rows[(long)k * 6 + c];
      out_index[pos] = (int)(keys[keep[k]] & 0xffffffffu);
    }
    base += __popcll(b);
  }
  if (lane == 0) *out_count = base;
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct ws_layout {
  size_t keys_sorted, sort_tmp, sort_tmp_bytes, det, boxes, mask, keep, nkeep, rows, valid, total;
};

int layout(long n_total, ws_layout& L) {
  if (n_total < 0) return SODT_EINVAL;
  const long n = n_total < MAX_NMS ? n_total : MAX_NMS;
  const long W = (n + 63) / 64;
  size_t tb = 0;
  if (n_total > 0 && rocprim::radix_sort_keys<rocprim::default_config, const uint64_t*, uint64_t*>(
          nullptr, tb, nullptr, nullptr, (size_t)n_total, 0, 64, 0) != hipSuccess)
    return SODT_EINVAL;
  size_t o = 0;
  L.keys_sorted = o; o += align256((size_t)n_total * 8);
  L.sort_tmp = o; L.sort_tmp_bytes = tb; o += align256(tb);
  L.det = o; o += align256((size_t)n * 24);
  L.boxes = o; o += align256((size_t)n * 16);
  L.mask = o; o += align256((size_t)n * W * 8);
  L.keep = o; o += align256(MAX_DET * 4);
  L.nkeep = o; o += 256;
  L.rows = o; o += align256(MAX_DET * 24);
  L.valid = o; o += align256(MAX_DET * 4);
  L.total = o;
  return SODT_OK;
}

}  // namespace

extern "C" int sodt_nms_candidates(const float* z, int N, int nc, float conf_thres, int multi_label,
                                   const unsigned char* class_allow, unsigned long long* keys, int cap, int* count,
                                   hipStream_t stream) {
  if (!z || !keys || !count || N <= 0 || nc <= 0 || cap <= 0 || (long)N * nc > 0x7fffffffL) return SODT_EINVAL;
  if (hipMemsetAsync(count, 0, sizeof(int), stream) != hipSuccess) return SODT_EINVAL;
  nms_candidates_kernel<<<(N + 255) / 256, 256, 0, stream>>>(z, N, nc, conf_thres, multi_label && nc > 1, class_allow,
                                                             (uint64_t*)keys, cap, count);
  return SODT_OK;
}

extern "C" int sodt_nms_workspace_bytes(long n_total, size_t* bytes) {
  ws_layout L;
  if (!bytes || layout(n_total, L) != SODT_OK) return SODT_EINVAL;
  *bytes = L.total;
  return SODT_OK;
}

extern "C" int sodt_nms_select(const float* z, int nc, const unsigned long long* keys, long n_total, float iou_thres,
                               int agnostic, void* ws, size_t ws_bytes, float* out, int* out_index, int* out_count,
                               hipStream_t stream) {
  ws_layout L;
  if (!z || !keys || !ws || !out || !out_index || !out_count || nc <= 0 || n_total <= 0) return SODT_EINVAL;
  if (layout(n_total, L) != SODT_OK || ws_bytes < L.total) return SODT_EINVAL;
  char* base = (char*)ws;
  uint64_t* ks = (uint64_t*)(base + L.keys_sorted);
  size_t tb = L.sort_tmp_bytes;
  if (rocprim::radix_sort_keys(base + L.sort_tmp, tb, (const uint64_t*)keys, ks, (size_t)n_total, 0, 64, stream) != hipSuccess)
    return SODT_EINVAL;
  const int n = (int)(n_total < MAX_NMS ? n_total : MAX_NMS);
  const int W = (n + 63) / 64;
  float* det = (float*)(base + L.det);
  float4* boxes = (float4*)(base + L.boxes);
  uint64_t* mask = (uint64_t*)(base + L.mask);
  int* keep = (int*)(base + L.keep);
  int* nkeep = (int*)(base + L.nkeep);
  float* rows = (float*)(base + L.rows);
  int* valid = (int*)(base + L.valid);
  nms_gather_kernel<<<(n + 255) / 256, 256, 0, stream>>>(z, ks, n, nc, agnostic, det, boxes);
  nms_mask_kernel<<<dim3(W, W), 64, 0, stream>>>(boxes, n, W, iou_thres, mask);
  nms_reduce_kernel<<<1, 512, 0, stream>>>(mask, n, W, MAX_DET, keep, nkeep);
  const int merge = n_total > 1 && n_total < 3000;   // general.py:500 tests the pre-truncation count
  nms_merge_kernel<<<MAX_DET, 64, 0, stream>>>(det, boxes, n, keep, nkeep, iou_thres, merge, rows, valid);
  nms_compact_kernel<<<1, 64, 0, stream>>>(rows, valid, keep, ks, nkeep, out, out_index, out_count);
  return SODT_OK;
}
