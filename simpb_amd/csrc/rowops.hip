// Small row-wise kernels of the decoder that replace runs of 5-12 elementwise PyTorch launches each
// (gfx950). None of them is bandwidth- or FLOP-relevant; what they remove is launch count on the
// decoder's critical path (each such launch is 2-5 us of dependent latency in the replayed frame).
#include <hip/hip_runtime.h>
#include "../../include/simpb_hip.h"
#include "store_fence.h"

extern "C" int simpb_check_launch(void);

namespace {

// ---- SparseBox3DKeyPointsGenerator.anchor_projection
// (/root/reference/projects/mmdet3d_plugin/models/detection3d/blocks.py:248-280), one thread per
// anchor: centre moved by -vel * dt then through [R | t], size kept, velocity rotated, and the
// yaw pair rotated AS WRITTEN THERE (:271-278): the 2x2 block is applied to [cos, sin] and the
// result is stored back into the [sin, cos] slots in that order.
__global__ __launch_bounds__(256) void anchor_projection_kernel(float* __restrict__ out, const float* __restrict__ anchor,
                                                                const float* __restrict__ T, const float* __restrict__ dt,
                                                                int bs, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= bs * n) return;
  const int b = i / n;
  float a[11], m[12];
  {
    const float* ap = anchor + (size_t)i * 11;
    const float* mp = T + (size_t)b * 16;
#pragma unroll
    for (int k = 0; k < 11; ++k) a[k] = ap[k];
#pragma unroll
    for (int k = 0; k < 12; ++k) m[k] = mp[k];
  }
  const float t = dt ? dt[b] : 0.f;
  simpb::loads_retired();  // store_fence.h: every consumer behind a full vmcnt(0)
  const float vx = a[8], vy = a[9], vz = a[10];
  float cx = a[0], cy = a[1], cz = a[2];
  if (dt) { cx -= vx * t; cy -= vy * t; cz -= vz * t; }
  // matmul(T[:3,:3], c) + T[:3,3]: products summed left to right like the batched matmul
  float r[11];
  r[0] = m[0] * cx + m[1] * cy + m[2] * cz + m[3];
  r[1] = m[4] * cx + m[5] * cy + m[6] * cz + m[7];
  r[2] = m[8] * cx + m[9] * cy + m[10] * cz + m[11];
  r[3] = a[3]; r[4] = a[4]; r[5] = a[5];
  const float s = a[6], c = a[7];
  r[6] = m[0] * c + m[1] * s;
  r[7] = m[4] * c + m[5] * s;
  r[8] = m[0] * vx + m[1] * vy + m[2] * vz;
  r[9] = m[4] * vx + m[5] * vy + m[6] * vz;
  r[10] = m[8] * vx + m[9] * vy + m[10] * vz;
#pragma unroll
  for (int k = 0; k < 11; ++k) simpb::pin(r[k]);
  simpb::loads_retired();  // store_fence.h: results in registers, loads retired, then the stores
  float* o = out + (size_t)i * 11;
#pragma unroll
  for (int k = 0; k < 11; ++k) o[k] = r[k];
}

// ---- operands of the grouped multi-scale deformable attention from the fused projection
// (group_attn.py:181-201): one thread per (query slot, head, level*point) element, coalesced; the
// softmax over the L*P (= 16 shipped) logits of a head is a 16-lane reduction. raw row = [offsets
// heads*L*P*2 | logits heads*L*P]; location = reference point + offset / (W_l, H_l).
template <int LP>
__global__ __launch_bounds__(256) void msda_prep_kernel(float* __restrict__ loc, float* __restrict__ attn,
                                                        const float* __restrict__ raw, int ldraw,
                                                        const float* __restrict__ ref, int ldref,
                                                        const long long* __restrict__ spatial_shapes, int rows,
                                                        int heads, int P, const int* __restrict__ m_live) {
  const int per_row = heads * LP;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int q = (int)(i / per_row), e = (int)(i - (long long)q * per_row);  // e = head * LP + level * P + point
  const bool in = q < rows;
  const int live = m_live ? *m_live : rows;
  const bool alive = in && q < live;
  float lg = -INFINITY, ox = 0.f, oy = 0.f, rx = 0.f, ry = 0.f, wl = 1.f, hl = 1.f;
  if (alive) {
    const float* r = raw + (size_t)q * ldraw;
    const float2 o = *reinterpret_cast<const float2*>(r + 2 * e);
    ox = o.x; oy = o.y;
    lg = r[2 * per_row + e];
    rx = ref[(size_t)q * ldref]; ry = ref[(size_t)q * ldref + 1];
    const int l = (e % LP) / P;
    hl = (float)spatial_shapes[2 * l]; wl = (float)spatial_shapes[2 * l + 1];
  }
  float mx = lg;
#pragma unroll
  for (int m = LP / 2; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
  const float ex = alive ? expf(lg - mx) : 0.f;
  float sum = ex;
#pragma unroll
  for (int m = LP / 2; m >= 1; m >>= 1) sum += __shfl_xor(sum, m);
  if (!in) return;
  const size_t o = (size_t)q * per_row + e;
  reinterpret_cast<float2*>(loc)[o] = alive ? make_float2(rx + ox / wl, ry + oy / hl) : make_float2(0.f, 0.f);
  attn[o] = alive ? ex / sum : 0.f;
}

// ---- ReWeight.alpha (aggregation.py:23-24): one wave per row
__global__ __launch_bounds__(256) void rowdot_sigmoid_kernel(float* __restrict__ out, const float* __restrict__ x, int ldx,
                                                             const float* __restrict__ w, const float* __restrict__ b,
                                                             int rows, int k, const int* __restrict__ m_live) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int live = m_live ? *m_live : rows;
  if (row >= live) {
    if (lane == 0) out[row] = 0.f;
    return;
  }
  float s = 0.f;
  for (int c = lane * 4; c < k; c += 256) {
    const float4 xv = *reinterpret_cast<const float4*>(x + (size_t)row * ldx + c);
    const float4 wv = *reinterpret_cast<const float4*>(w + c);
    s += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
  if (lane == 0) out[row] = 1.f / (1.f + expf(-(s + (b ? b[0] : 0.f))));
}

// ---- top-k of one score row per workgroup (n <= 2048), sorted descending, ties by ascending index:
// the whole row is bitonic-sorted in LDS as 64-bit keys (order-preserving bits of the float << 32 |
// ~index). Replaces torch.topk(sorted=True) (a radix-select + a radix sort launch, 40-45 us on 900
// scores) behind InstanceBank.update / cache / update_instance_id (instance_bank.py:13-20,137,159,191)
// and SparseBox3DDecoder (decoder.py:145-167).
template <int CAP>
__global__ __launch_bounds__(CAP / 2) void topk_rows_kernel(float* __restrict__ val_out, int* __restrict__ idx_out,
                                                          const float* __restrict__ scores, int n, int k) {
  __shared__ unsigned long long key[CAP];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* row = scores + (size_t)b * n;
  for (int i = tid; i < CAP; i += CAP / 2) {
    unsigned long long kv = 0ull;  // padding sorts last
    if (i < n) {
      unsigned u = __float_as_uint(row[i]);
      u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // monotone map float -> uint
      kv = ((unsigned long long)u << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)i);
    }
    key[i] = kv;
  }
  __syncthreads();
  for (int size = 2; size <= CAP; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const int lo = 2 * tid - (tid & (stride - 1));  // index of the lower element of this thread's pair
      const int hi = lo + stride;
      const bool desc = (lo & size) == 0;             // first half of each 2*size block descending
      const unsigned long long a = key[lo], c = key[hi];
      if ((a < c) == desc) { key[lo] = c; key[hi] = a; }
      __syncthreads();
    }
  }
  for (int i = tid; i < k; i += CAP / 2) {
    const unsigned long long kv = key[i];
    unsigned u = (unsigned)(kv >> 32);
    u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    val_out[(size_t)b * k + i] = __uint_as_float(u);
    idx_out[(size_t)b * k + i] = (int)(0xFFFFFFFFu - (unsigned)(kv & 0xFFFFFFFFull));
  }
}

}  // namespace

extern "C" int simpb_topk_rows(float* values, int* indices, const float* scores, int batch_size, int n, int k,
                               void* stream) {
  if (!values || !indices || !scores || batch_size <= 0 || n <= 0 || k <= 0 || k > n || n > 2048) return SIMPB_EINVAL;
  (void)hipGetLastError();
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n <= 512)
    hipLaunchKernelGGL(topk_rows_kernel<512>, dim3(batch_size), dim3(256), 0, s, values, indices, scores, n, k);
  else if (n <= 1024)
    hipLaunchKernelGGL(topk_rows_kernel<1024>, dim3(batch_size), dim3(512), 0, s, values, indices, scores, n, k);
  else
    hipLaunchKernelGGL(topk_rows_kernel<2048>, dim3(batch_size), dim3(1024), 0, s, values, indices, scores, n, k);
  return simpb_check_launch();
}

extern "C" int simpb_rowdot_sigmoid(float* out, const float* x, int ldx, const float* w, const float* b, int num_rows,
                                    int k, const int* m_live, void* stream) {
  if (!out || !x || !w || num_rows <= 0 || k <= 0 || (k & 3) || ldx < k || (ldx & 3)) return SIMPB_EINVAL;
  if ((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(w)) & 15) return SIMPB_EINVAL;
  (void)hipGetLastError();
  hipLaunchKernelGGL(rowdot_sigmoid_kernel, dim3((num_rows + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), out, x,
                     ldx, w, b, num_rows, k, m_live);
  return simpb_check_launch();
}

extern "C" int simpb_anchor_projection(float* out, const float* anchor, const float* T_src2dst, const float* time_interval,
                                       int batch_size, int num_anchors, void* stream) {
  if (!out || !anchor || !T_src2dst || batch_size <= 0 || num_anchors <= 0) return SIMPB_EINVAL;
  (void)hipGetLastError();
  const int total = batch_size * num_anchors;
  hipLaunchKernelGGL(anchor_projection_kernel, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                     out, anchor, T_src2dst, time_interval, batch_size, num_anchors);
  return simpb_check_launch();
}

extern "C" int simpb_msda_prep(float* sampling_loc, float* attn_weight, const float* raw, int ldraw, const float* ref,
                               int ldref, const long long* spatial_shapes, int num_rows, int num_heads, int num_levels,
                               int num_points, const int* m_live, void* stream) {
  if (!sampling_loc || !attn_weight || !raw || !ref || !spatial_shapes || num_rows <= 0 || num_heads <= 0 ||
      num_levels <= 0 || num_points <= 0 || ldraw < num_heads * num_levels * num_points * 3 || ldref < 2)
    return SIMPB_EINVAL;
  const int lp = num_levels * num_points;
  if ((lp != 16 && lp != 32 && lp != 8) || (ldraw & 1) || (reinterpret_cast<size_t>(raw) & 7) ||
      (reinterpret_cast<size_t>(sampling_loc) & 7))
    return SIMPB_EINVAL;  // the softmax is a power-of-two lane-group reduction
  (void)hipGetLastError();
  const long long total = (long long)num_rows * num_heads * lp;
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (lp == 16)
    hipLaunchKernelGGL(msda_prep_kernel<16>, grid, block, 0, s, sampling_loc, attn_weight, raw, ldraw, ref, ldref,
                       spatial_shapes, num_rows, num_heads, num_points, m_live);
  else if (lp == 32)
    hipLaunchKernelGGL(msda_prep_kernel<32>, grid, block, 0, s, sampling_loc, attn_weight, raw, ldraw, ref, ldref,
                       spatial_shapes, num_rows, num_heads, num_points, m_live);
  else
    hipLaunchKernelGGL(msda_prep_kernel<8>, grid, block, 0, s, sampling_loc, attn_weight, raw, ldraw, ref, ldref,
                       spatial_shapes, num_rows, num_heads, num_points, m_live);
  return simpb_check_launch();
}
