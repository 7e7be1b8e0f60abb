// Adaptive query allocation (3D anchors -> per-camera 2D query slots) and the 2D<->3D index
// moves built on it (gfx950). Follows DynamicQueryAllocation.projection_allocation
// (/root/reference/projects/mmdet3d_plugin/models/allocation.py:27-144) but produces INDEX LISTS
// (q2a: slot -> anchor, a2q: (anchor, cam) -> slot) instead of dense one-hot matrices, so the
// matmuls against those matrices (simpb_head.py:438, aggregation.py:32-35) become row gathers
// and a <=num_cams-term weighted mean. All byte/index work: HBM-bound, no MFMA.
#include <hip/hip_runtime.h>
#include "../../include/simpb_hip.h"
#include "store_fence.h"

extern "C" int simpb_check_launch(void);

namespace {

constexpr int kMaxStaticCams = 8;   // cameras of the static-capacity path (simpb_alloc_static)

// ---- step 1: one thread per (batch, anchor, cam): 9 projected points -> flag, 2D ref, depth
__global__ void alloc_project_kernel(unsigned char* __restrict__ flag, float* __restrict__ sel_xy,
                                     float* __restrict__ depth, const float* __restrict__ anchor,
                                     const float* __restrict__ proj, int bs, int A, int cams, float img_w, float img_h,
                                     float lim_w, float lim_l, float lim_h, int* __restrict__ a2q_fill) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= bs * A * cams) return;
  const int cam = idx % cams;
  const int a = (idx / cams) % A;
  const int b = idx / (cams * A);
  const float* an = anchor + ((size_t)b * A + a) * 11;
  const float* P = proj + ((size_t)b * cams + cam) * 16;
  const float cx = an[0], cy = an[1], cz = an[2];
  const float sw = fminf(expf(an[3]), lim_w), sl = fminf(expf(an[4]), lim_l), sh = fminf(expf(an[5]), lim_h);
  const float sn = an[6], cs = an[7];

  float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
  bool corner_valid = false;
  float ctr_x = 0.f, ctr_y = 0.f, ctr_d = 0.f;
  for (int k = 0; k < 9; ++k) {
    float px, py, pz;
    if (k < 8) {  // corner k = (i, j, l) bits, MSB first (np.unravel_index order, allocation.py:43-44)
      const float ox = (((k >> 2) & 1) - 0.5f) * sw, oy = (((k >> 1) & 1) - 0.5f) * sl, oz = ((k & 1) - 0.5f) * sh;
      px = cs * ox - sn * oy + cx;
      py = sn * ox + cs * oy + cy;
      pz = oz + cz;
    } else {
      px = cx; py = cy; pz = cz;
    }
    const float u = P[0] * px + P[1] * py + P[2] * pz + P[3];
    const float v = P[4] * px + P[5] * py + P[6] * pz + P[7];
    const float d = P[8] * px + P[9] * py + P[10] * pz + P[11];
    const float dc = fmaxf(d, 1e-5f);
    const float x = u / dc, y = v / dc;
    const bool inside = 0.f < x && x < img_w && 0.f < y && y < img_h;
    if (k < 8) {
      corner_valid = corner_valid || (d > 0.f && inside);
      xmin = fminf(xmin, x); xmax = fmaxf(xmax, x);
      ymin = fminf(ymin, y); ymax = fmaxf(ymax, y);
    } else {
      ctr_x = x; ctr_y = y; ctr_d = d;
    }
  }
  const bool center_valid = 0.f < ctr_x && ctr_x < img_w && 0.f < ctr_y && ctr_y < img_h;  // no depth test (:67-68)
  float sx = (fminf(fmaxf(xmin, 0.f), img_w) + fminf(fmaxf(xmax, 0.f), img_w)) / 2.f;
  float sy = (fminf(fmaxf(ymin, 0.f), img_h) + fminf(fmaxf(ymax, 0.f), img_h)) / 2.f;
  if (center_valid) { sx = ctr_x; sy = ctr_y; }
  const size_t o = ((size_t)b * cams + cam) * A + a;
  flag[o] = center_valid ? 2 : (corner_valid ? 1 : 0);
  sel_xy[2 * o] = sx;
  sel_xy[2 * o + 1] = sy;
  depth[o] = ctr_d;
  if (a2q_fill) a2q_fill[idx] = -1;   // the (anchor, cam) -> slot table starts empty (same index space): no fill launch
}

// ---- step 2: one workgroup per (batch, cam): stable compaction of the flagged anchors
__global__ __launch_bounds__(256) void alloc_compact_kernel(int* __restrict__ count, int* __restrict__ order,
                                                            const unsigned char* __restrict__ flag, int A) {
  __shared__ int s_wave[4];
  __shared__ int s_base;
  const int bc = blockIdx.x;
  const unsigned char* f = flag + (size_t)bc * A;
  int* ord = order + (size_t)bc * A;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  for (int a0 = 0; a0 < A; a0 += 256) {
    const int a = a0 + threadIdx.x;
    const bool on = a < A && f[a] != 0;
    const unsigned long long m = __ballot(on);
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; ++w) off += s_wave[w];
    if (on) ord[off + __popcll(m & ((1ull << lane) - 1ull))] = a;
    __syncthreads();
    if (threadIdx.x == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) count[bc] = s_base;
}

// ---- step 3: one thread per (batch, slot): fill the slot tables; pads get q2a = -1 and zeros
__global__ void alloc_scatter_kernel(float* __restrict__ ref_pts2d, float* __restrict__ ref_depth2d,
                                     int* __restrict__ q2a, int* __restrict__ is_center, int* __restrict__ a2q,
                                     int* __restrict__ query_cam, const int* __restrict__ group_start,
                                     const int* __restrict__ count, const int* __restrict__ order,
                                     const unsigned char* __restrict__ flag, const float* __restrict__ sel_xy,
                                     const float* __restrict__ depth, int bs, int A, int cams, int N2, float img_w,
                                     float img_h, int* __restrict__ group_start_out, int* __restrict__ overflow_out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= bs * N2) return;
  const int slot = idx % N2, b = idx / N2;
  int cam = 0, g_lo = 0, g_end = 0;
  int gs[kMaxStaticCams];
  bool over = false;
  if (group_start_out) {
    // static capacity: every thread derives the group table from the counts itself (alloc_group_start_kernel's
    // arithmetic: prefix sums of the max-over-batch counts clipped to the capacity N2); thread 0 publishes it below
    int acc = 0;
    bool found = false;
#pragma unroll
    for (int c = 0; c < kMaxStaticCams; ++c) {
      gs[c] = acc;
      if (c < cams) {
        int m = 0;
        for (int bb = 0; bb < bs; ++bb) m = max(m, count[bb * cams + c]);
        const int lo = acc;
        acc += m;
        if (acc > N2) { over = true; acc = N2; }
        if (!found && (slot < acc || c == cams - 1)) { cam = c; g_lo = lo; found = true; }
        gs[c] = acc;
      }
    }
    g_end = acc;
  } else {
    while (cam + 1 < cams && slot >= group_start[cam + 1]) ++cam;
    g_lo = group_start[cam];
    g_end = group_start[cams];
  }
  const int rank = slot - g_lo;
  const bool in_set = slot < g_end;  // capacity slots past the last group belong to no camera
  const int bc = b * cams + cam;
  float x = 0.f, y = 0.f, d = 0.f;
  int a = -1, ctr = 0;
  if (in_set && rank < count[bc]) {
    a = order[(size_t)bc * A + rank];
    const size_t o = (size_t)bc * A + a;
    x = sel_xy[2 * o] / img_w;
    y = sel_xy[2 * o + 1] / img_h;
    d = fabsf(depth[o]);
    ctr = flag[o] == 2;
  }
  simpb::pin(x); simpb::pin(y); simpb::pin(d); simpb::pin(a); simpb::pin(ctr);
  simpb::loads_retired();  // store_fence.h: every table entry read, then the stores
  if (group_start_out && idx == 0) {
    group_start_out[0] = 0;
#pragma unroll
    for (int c = 0; c < kMaxStaticCams; ++c)
      if (c < cams) group_start_out[c + 1] = gs[c];
    overflow_out[0] = over ? 1 : 0;
  }
  if (b == 0) query_cam[slot] = in_set ? cam : -1;
  if (a >= 0) a2q[((size_t)b * A + a) * cams + cam] = slot;
  ref_pts2d[2 * (size_t)idx] = x;
  ref_pts2d[2 * (size_t)idx + 1] = y;
  ref_depth2d[idx] = d;
  q2a[idx] = a;
  is_center[idx] = ctr;
}

// ---- step 3, independent streams ("ragged" batch; SURVEY.md §8e: keep per-sample counts in the native path). The batch
// holds bs independent camera streams, so nothing is padded to the max over the batch (allocation.py:91-99 would): the 2D
// set is ONE flat slot array, stream-major then camera-major, group g = b * cams + cam holding count[g] slots; a slot's
// q2a is the flat anchor index b * A + a, query_cam its group, and slots past the last group are capacity slots. Every
// stream keeps exactly the set a batch of one would give it (at most `per_stream` slots; more sets the overflow flag and is
// clipped). One thread per flat slot; the group table (prefix sums of the counts) is built per workgroup in LDS.
constexpr int kMaxRaggedGroups = 96;
__global__ __launch_bounds__(256) void alloc_scatter_ragged_kernel(
    float* __restrict__ ref_pts2d, float* __restrict__ ref_depth2d, int* __restrict__ q2a, int* __restrict__ is_center,
    int* __restrict__ a2q, int* __restrict__ query_cam, const int* __restrict__ count, const int* __restrict__ order,
    const unsigned char* __restrict__ flag, const float* __restrict__ sel_xy, const float* __restrict__ depth, int bs, int A,
    int cams, int per_stream, float img_w, float img_h, int* __restrict__ group_start_out, int* __restrict__ overflow_out) {
  __shared__ int s_start[kMaxRaggedGroups + 1];
  __shared__ int s_over;
  const int G = bs * cams;
  if (threadIdx.x == 0) {
    int acc = 0, over = 0;
    for (int b = 0; b < bs; ++b) {
      int in_stream = 0;
      for (int c = 0; c < cams; ++c) {
        s_start[b * cams + c] = acc + in_stream;
        in_stream += count[b * cams + c];
        if (in_stream > per_stream) { over = 1; in_stream = per_stream; }
      }
      acc += in_stream;
    }
    s_start[G] = acc;
    s_over = over;
  }
  __syncthreads();
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = bs * per_stream;
  const int live = s_start[G];
  int g = -1;
  if (slot < live) {   // the last group whose start is <= slot (groups may be empty: equal starts)
    int lo = 0, hi = G - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (s_start[mid] <= slot) lo = mid; else hi = mid - 1;
    }
    g = lo;
  }
  float x = 0.f, y = 0.f, d = 0.f;
  int a = -1, ctr = 0, b = 0;
  if (g >= 0) {
    const int rank = slot - s_start[g];   // < s_start[g + 1] - s_start[g] <= count[g]
    b = g / cams;
    a = order[(size_t)g * A + rank];
    const size_t o = (size_t)g * A + a;
    x = sel_xy[2 * o] / img_w;
    y = sel_xy[2 * o + 1] / img_h;
    d = fabsf(depth[o]);
    ctr = flag[o] == 2;
  }
  simpb::pin(x); simpb::pin(y); simpb::pin(d); simpb::pin(a); simpb::pin(ctr);
  simpb::loads_retired();  // store_fence.h
  if (blockIdx.x == 0 && threadIdx.x <= G) group_start_out[threadIdx.x] = s_start[threadIdx.x];
  if (slot == 0) overflow_out[0] = s_over;
  if (slot >= total) return;
  query_cam[slot] = g;
  if (a >= 0) a2q[((size_t)b * A + a) * cams + (g - b * cams)] = slot;
  ref_pts2d[2 * (size_t)slot] = x;
  ref_pts2d[2 * (size_t)slot + 1] = y;
  ref_depth2d[slot] = d;
  q2a[slot] = a >= 0 ? b * A + a : -1;
  is_center[slot] = ctr;
}

// ---- device-side group table (allocation.py:91-99 without the .tolist()): group_start[c+1] =
// sum over c' <= c of max over batch of count[b, c']; overflow[0] = 1 if the set exceeds capacity
__global__ void alloc_group_start_kernel(int* __restrict__ group_start, int* __restrict__ overflow,
                                         const int* __restrict__ count, int bs, int cams, int capacity) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int acc = 0;
  group_start[0] = 0;
  bool over = false;
  for (int c = 0; c < cams; ++c) {
    int m = 0;
    for (int b = 0; b < bs; ++b) m = max(m, count[b * cams + c]);
    acc += m;
    if (acc > capacity) { over = true; acc = capacity; }
    group_start[c + 1] = acc;
  }
  overflow[0] = over ? 1 : 0;
}

__global__ void fill_int_kernel(int* __restrict__ p, int v, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// ---- row gather: out[b, s, :] = src[b, q2a[b, s], :] (zeros for q2a < 0)   (simpb_head.py:438)
__global__ void gather_rows_kernel(float* __restrict__ out, const float* __restrict__ src,
                                   const int* __restrict__ q2a, int A, int N2, int C4) {
  const int s = blockIdx.x, b = blockIdx.y;
  const int a = q2a[(size_t)b * N2 + s];
  float4* o = reinterpret_cast<float4*>(out) + ((size_t)b * N2 + s) * C4;
  const float4* r = reinterpret_cast<const float4*>(src) + ((size_t)b * A + max(a, 0)) * C4;
  for (int c = threadIdx.x; c < C4; c += blockDim.x) o[c] = a >= 0 ? r[c] : make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---- 2D -> 3D weighted mean (aggregation.py:30-35): for anchor a, over its <= cams slots s,
//   out_q = q3d + sum alpha[s]*q2d[s] / clamp(sum alpha[s], 1e-5)   (same for the pos stream)
// alpha: given per slot, or (hidden != NULL) computed here as ReWeight.alpha (aggregation.py:23-24):
// sigmoid(hidden[s] . w_alpha + b_alpha), one wave-wide dot per slot -- every slot belongs to exactly one anchor, so
// nothing is computed twice and the separate row-dot launch goes away. One wave per anchor.
constexpr int kMaxAggCams = 8;
__global__ __launch_bounds__(64) void aggregate_kernel(float* __restrict__ out_q, float* __restrict__ out_pos,
                                                       const float* __restrict__ q3d, const float* __restrict__ pos3d,
                                                       const float* __restrict__ q2d, const float* __restrict__ pos2d,
                                                       const float* __restrict__ alpha, const int* __restrict__ a2q, int A,
                                                       int cams, int N2, int C4, const float* __restrict__ hidden, int ldh,
                                                       int kh, const float* __restrict__ w_alpha,
                                                       const float* __restrict__ b_alpha) {
  const int a = blockIdx.x, b = blockIdx.y;
  const int* slots = a2q + ((size_t)b * A + a) * cams;
  float wgt[kMaxAggCams];
  int slot[kMaxAggCams];
  float div = 0.f;
#pragma unroll
  for (int k = 0; k < kMaxAggCams; ++k) {
    slot[k] = k < cams ? slots[k] : -1;
    wgt[k] = 0.f;
  }
#pragma unroll
  for (int k = 0; k < kMaxAggCams; ++k) {
    const int s = slot[k];   // wave-uniform
    if (s < 0) continue;
    float w;
    if (hidden) {
      const float* h = hidden + ((size_t)b * N2 + s) * ldh;
      float d = 0.f;
      for (int c = threadIdx.x * 4; c < kh; c += 256) {
        const float4 xv = *reinterpret_cast<const float4*>(h + c);
        const float4 wv = *reinterpret_cast<const float4*>(w_alpha + c);
        d += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;   // rowdot_sigmoid_kernel's arithmetic (csrc/rowops.hip)
      }
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) d += __shfl_xor(d, m);
      w = 1.f / (1.f + expf(-(d + (b_alpha ? b_alpha[0] : 0.f))));
    } else {
      w = alpha[(size_t)b * N2 + s];
    }
    wgt[k] = w;
    div += w;
  }
  div = fmaxf(div, 1e-5f);
  const size_t row = ((size_t)b * A + a) * C4;
  for (int c = threadIdx.x; c < C4; c += blockDim.x) {
    float4 sq = make_float4(0.f, 0.f, 0.f, 0.f), sp = sq;
#pragma unroll
    for (int k = 0; k < kMaxAggCams; ++k) {
      const int s = slot[k];
      if (s < 0) continue;
      const float w = wgt[k];
      const float4 vq = reinterpret_cast<const float4*>(q2d)[((size_t)b * N2 + s) * C4 + c];
      const float4 vp = reinterpret_cast<const float4*>(pos2d)[((size_t)b * N2 + s) * C4 + c];
      sq.x += w * vq.x; sq.y += w * vq.y; sq.z += w * vq.z; sq.w += w * vq.w;
      sp.x += w * vp.x; sp.y += w * vp.y; sp.z += w * vp.z; sp.w += w * vp.w;
    }
    const float4 bq = reinterpret_cast<const float4*>(q3d)[row + c];
    const float4 bp = reinterpret_cast<const float4*>(pos3d)[row + c];
    float4 rq = make_float4(bq.x + sq.x / div, bq.y + sq.y / div, bq.z + sq.z / div, bq.w + sq.w / div);
    float4 rp = make_float4(bp.x + sp.x / div, bp.y + sp.y / div, bp.z + sp.z / div, bp.w + sp.w / div);
    simpb::pin(rq.x); simpb::pin(rq.y); simpb::pin(rq.z); simpb::pin(rq.w);
    simpb::pin(rp.x); simpb::pin(rp.y); simpb::pin(rp.z); simpb::pin(rp.w);
    simpb::loads_retired();  // store_fence.h
    reinterpret_cast<float4*>(out_q)[row + c] = rq;
    reinterpret_cast<float4*>(out_pos)[row + c] = rp;
    simpb::loads_retired();  // (also a store fence: the next channel slice loads with nothing in flight)
  }
}

inline int status() { return simpb_check_launch(); }
inline void clear_stale() { (void)hipGetLastError(); }  // errors left by the caller's earlier runtime calls

}  // namespace

extern "C" int simpb_alloc_project(unsigned char* flag, float* sel_xy, float* depth, const float* anchor,
                                   const float* projection_mat, int batch_size, int num_anchors, int num_cams,
                                   float img_w, float img_h, float limit_w, float limit_l, float limit_h,
                                   void* stream) {
  if (!flag || !sel_xy || !depth || !anchor || !projection_mat || batch_size <= 0 || num_anchors <= 0 || num_cams <= 0)
    return SIMPB_EINVAL;
  const int n = batch_size * num_anchors * num_cams;
  clear_stale();
  hipLaunchKernelGGL(alloc_project_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), flag,
                     sel_xy, depth, anchor, projection_mat, batch_size, num_anchors, num_cams, img_w, img_h, limit_w,
                     limit_l, limit_h, static_cast<int*>(nullptr));
  return status();
}

extern "C" int simpb_alloc_compact(int* count, int* order, const unsigned char* flag, int batch_size, int num_anchors,
                                   int num_cams, void* stream) {
  if (!count || !order || !flag || batch_size <= 0 || num_anchors <= 0 || num_cams <= 0) return SIMPB_EINVAL;
  clear_stale();
  hipLaunchKernelGGL(alloc_compact_kernel, dim3(batch_size * num_cams), dim3(256), 0, static_cast<hipStream_t>(stream),
                     count, order, flag, num_anchors);
  return status();
}

extern "C" int simpb_alloc_group_start(int* group_start, int* overflow, const int* count, int batch_size, int num_cams,
                                       int capacity, void* stream) {
  if (!group_start || !overflow || !count || batch_size <= 0 || num_cams <= 0 || capacity <= 0) return SIMPB_EINVAL;
  clear_stale();
  hipLaunchKernelGGL(alloc_group_start_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), group_start,
                     overflow, count, batch_size, num_cams, capacity);
  return status();
}

extern "C" int simpb_alloc_scatter(float* ref_pts2d, float* ref_depth2d, int* q2a, int* is_center, int* a2q,
                                   int* query_cam, const int* group_start, const int* count, const int* order,
                                   const unsigned char* flag, const float* sel_xy, const float* depth, int batch_size,
                                   int num_anchors, int num_cams, int num_query, float img_w, float img_h,
                                   void* stream) {
  if (!a2q || !group_start || !count || !order || !flag || !sel_xy || !depth || batch_size <= 0 || num_anchors <= 0 ||
      num_cams <= 0 || num_query < 0)
    return SIMPB_EINVAL;
  (void)hipGetLastError();  // drop a stale error left by earlier runtime calls of the caller
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t na = (size_t)batch_size * num_anchors * num_cams;
  hipLaunchKernelGGL(fill_int_kernel, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, s, a2q, -1, na);
  if (num_query > 0) {
    if (!ref_pts2d || !ref_depth2d || !q2a || !is_center || !query_cam) return SIMPB_EINVAL;
    const int n = batch_size * num_query;
    hipLaunchKernelGGL(alloc_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, s, ref_pts2d, ref_depth2d, q2a,
                       is_center, a2q, query_cam, group_start, count, order, flag, sel_xy, depth, batch_size,
                       num_anchors, num_cams, num_query, img_w, img_h, static_cast<int*>(nullptr), static_cast<int*>(nullptr));
  }
  return status();
}

extern "C" int simpb_gather_rows(float* out, const float* src, const int* q2a, int batch_size, int num_anchors,
                                 int num_query, int channels, void* stream) {
  if (!out || !src || !q2a || batch_size <= 0 || num_anchors <= 0 || num_query <= 0 || channels <= 0 ||
      channels % 4 != 0 || batch_size > 65535)
    return SIMPB_EINVAL;
  clear_stale();
  hipLaunchKernelGGL(gather_rows_kernel, dim3(num_query, batch_size), dim3(64), 0, static_cast<hipStream_t>(stream), out,
                     src, q2a, num_anchors, num_query, channels / 4);
  return status();
}

extern "C" int simpb_aggregate_2d_to_3d_alpha(float* out_q, float* out_pos, const float* q3d, const float* pos3d,
                                              const float* q2d, const float* pos2d, const float* alpha, const int* a2q,
                                              const float* hidden, int ld_hidden, int hidden_dim, const float* w_alpha,
                                              const float* b_alpha, int batch_size, int num_anchors, int num_cams,
                                              int num_query, int channels, void* stream) {
  if (!out_q || !out_pos || !q3d || !pos3d || !q2d || !pos2d || !a2q || batch_size <= 0 || num_anchors <= 0 ||
      num_cams <= 0 || num_cams > kMaxAggCams || num_query <= 0 || channels <= 0 || channels % 4 != 0 || batch_size > 65535)
    return SIMPB_EINVAL;
  if (hidden) {
    if (!w_alpha || hidden_dim <= 0 || hidden_dim % 4 || ld_hidden < hidden_dim || (ld_hidden & 3) ||
        (reinterpret_cast<size_t>(hidden) & 15) || (reinterpret_cast<size_t>(w_alpha) & 15))
      return SIMPB_EINVAL;
  } else if (!alpha) {
    return SIMPB_EINVAL;
  }
  clear_stale();
  hipLaunchKernelGGL(aggregate_kernel, dim3(num_anchors, batch_size), dim3(64), 0, static_cast<hipStream_t>(stream),
                     out_q, out_pos, q3d, pos3d, q2d, pos2d, alpha, a2q, num_anchors, num_cams, num_query, channels / 4,
                     hidden, ld_hidden, hidden_dim, w_alpha, b_alpha);
  return status();
}

extern "C" int simpb_aggregate_2d_to_3d(float* out_q, float* out_pos, const float* q3d, const float* pos3d,
                                        const float* q2d, const float* pos2d, const float* alpha, const int* a2q,
                                        int batch_size, int num_anchors, int num_cams, int num_query, int channels,
                                        void* stream) {
  if (!alpha) return SIMPB_EINVAL;
  return simpb_aggregate_2d_to_3d_alpha(out_q, out_pos, q3d, pos3d, q2d, pos2d, alpha, a2q, nullptr, 0, 0, nullptr, nullptr,
                                        batch_size, num_anchors, num_cams, num_query, channels, stream);
}

extern "C" int simpb_alloc_static(unsigned char* flag, float* sel_xy, float* depth, int* count, int* order, int* group_start,
                                  int* overflow, float* ref_pts2d, float* ref_depth2d, int* q2a, int* is_center, int* a2q,
                                  int* query_cam, const float* anchor, const float* projection_mat, int batch_size,
                                  int num_anchors, int num_cams, int capacity, float img_w, float img_h, float limit_w,
                                  float limit_l, float limit_h, void* stream) {
  if (!flag || !sel_xy || !depth || !count || !order || !group_start || !overflow || !ref_pts2d || !ref_depth2d || !q2a ||
      !is_center || !a2q || !query_cam || !anchor || !projection_mat || batch_size <= 0 || num_anchors <= 0 ||
      num_cams <= 0 || num_cams > kMaxStaticCams || capacity <= 0)
    return SIMPB_EINVAL;
  clear_stale();
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int n = batch_size * num_anchors * num_cams;
  hipLaunchKernelGGL(alloc_project_kernel, dim3((n + 255) / 256), dim3(256), 0, s, flag, sel_xy, depth, anchor, projection_mat,
                     batch_size, num_anchors, num_cams, img_w, img_h, limit_w, limit_l, limit_h, a2q);
  hipLaunchKernelGGL(alloc_compact_kernel, dim3(batch_size * num_cams), dim3(256), 0, s, count, order, flag, num_anchors);
  const int ns = batch_size * capacity;
  hipLaunchKernelGGL(alloc_scatter_kernel, dim3((ns + 255) / 256), dim3(256), 0, s, ref_pts2d, ref_depth2d, q2a, is_center, a2q,
                     query_cam, static_cast<const int*>(nullptr), count, order, flag, sel_xy, depth, batch_size, num_anchors,
                     num_cams, capacity, img_w, img_h, group_start, overflow);
  return status();
}

extern "C" int simpb_alloc_ragged(unsigned char* flag, float* sel_xy, float* depth, int* count, int* order, int* group_start,
                                  int* overflow, float* ref_pts2d, float* ref_depth2d, int* q2a, int* is_center, int* a2q,
                                  int* query_cam, const float* anchor, const float* projection_mat, int batch_size,
                                  int num_anchors, int num_cams, int per_stream, float img_w, float img_h, float limit_w,
                                  float limit_l, float limit_h, void* stream) {
  if (!flag || !sel_xy || !depth || !count || !order || !group_start || !overflow || !ref_pts2d || !ref_depth2d || !q2a ||
      !is_center || !a2q || !query_cam || !anchor || !projection_mat || batch_size <= 0 || num_anchors <= 0 ||
      num_cams <= 0 || batch_size * num_cams > kMaxRaggedGroups || per_stream <= 0 ||
      (long long)batch_size * per_stream > 0x7fffffffll / 4 || (long long)batch_size * num_anchors > 0x7fffffffll / 4)
    return SIMPB_EINVAL;
  clear_stale();
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int n = batch_size * num_anchors * num_cams;
  hipLaunchKernelGGL(alloc_project_kernel, dim3((n + 255) / 256), dim3(256), 0, s, flag, sel_xy, depth, anchor, projection_mat,
                     batch_size, num_anchors, num_cams, img_w, img_h, limit_w, limit_l, limit_h, a2q);
  hipLaunchKernelGGL(alloc_compact_kernel, dim3(batch_size * num_cams), dim3(256), 0, s, count, order, flag, num_anchors);
  const int ns = batch_size * per_stream;
  hipLaunchKernelGGL(alloc_scatter_ragged_kernel, dim3((ns + 255) / 256), dim3(256), 0, s, ref_pts2d, ref_depth2d, q2a,
                     is_center, a2q, query_cam, count, order, flag, sel_xy, depth, batch_size, num_anchors, num_cams, per_stream,
                     img_w, img_h, group_start, overflow);
  return status();
}
