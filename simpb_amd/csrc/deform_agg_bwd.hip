// 3D deformable feature aggregation, backward (gfx950): gradients w.r.t. the feature tokens, the
// sampling locations and the weights. Semantics follow deformable_aggregation_grad_kernel +
// bilinear_sampling_grad
// (/root/reference/projects/mmdet3d_plugin/ops/src/deformable_aggregation_cuda.cu:62-126,190-262).
//
// The reference spends 4 atomics per tap on the features, 2 on the location and 1 on the weight, per
// (anchor, point, camera, level, CHANNEL). Here the mapping of the forward kernel is kept (workgroup
// per (batch, anchor), wave per level, valid samples found by ballot), so the weight gradient is a
// half-wave register reduction + one plain store, the location gradient a wave reduction + an LDS
// meeting of the 4 waves + one plain store, and only the feature gradient -- a true scatter across
// anchors -- uses float atomics, shaped as whole 256-byte row segments (one dword per lane over 64
// consecutive channels), the shape the memory-side atomic units take at full rate.
// grad_weights and grad_sampling_location are therefore deterministic; grad_mc_ms_feat carries the
// usual last-bit atomic-order noise, as in the reference.
#include <hip/hip_runtime.h>
#include "../../include/simpb_hip.h"

extern "C" int simpb_check_launch(void);

namespace {

constexpr int kWaves = 4;
constexpr int kThreads = kWaves * 64;
constexpr int kMaxPK = 128;
constexpr int kNJ = 4;  // channels per lane: c = lane + 64*j

__global__ __launch_bounds__(kThreads) void daf_bwd_rows(
    float* __restrict__ g_feat, float* __restrict__ g_loc, float* __restrict__ g_w, const float* __restrict__ feat,
    const int* __restrict__ spatial_shape, const int* __restrict__ scale_start, const float* __restrict__ loc,
    const float* __restrict__ weights, const float* __restrict__ g_out, int num_cams, int num_feat, int C, int L, int A,
    int P, int G) {
  __shared__ float s_gl[kWaves][kMaxPK][2];
  const int a = blockIdx.x, b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int PK = P * num_cams;
  const size_t row = (size_t)b * A + a;
  const int nj = C / 64, gd = C / G;

  // this workgroup owns g_w[row] and g_loc[row]: clear them (samples outside the image get zero)
  float* gw_row = g_w + row * PK * L * G;
  for (int i = threadIdx.x; i < PK * L * G; i += kThreads) gw_row[i] = 0.f;
  for (int i = threadIdx.x; i < kWaves * kMaxPK * 2; i += kThreads) (&s_gl[0][0][0])[i] = 0.f;
  __syncthreads();

  const float2* loc2 = reinterpret_cast<const float2*>(loc) + row * PK;
  float2 l0 = make_float2(-1.f, -1.f), l1 = make_float2(-1.f, -1.f);
  if (lane < PK) l0 = loc2[lane];
  if (lane + 64 < PK) l1 = loc2[lane + 64];
  const unsigned long long m0 = __ballot(l0.x > 0.f && l0.x < 1.f && l0.y > 0.f && l0.y < 1.f);
  const unsigned long long m1 = __ballot(l1.x > 0.f && l1.x < 1.f && l1.y > 0.f && l1.y < 1.f);

  float go[kNJ];
  int grp[kNJ];
#pragma unroll
  for (int j = 0; j < kNJ; ++j) {
    const int c = lane + 64 * j;
    go[j] = j < nj ? g_out[row * C + c] : 0.f;
    grp[j] = j < nj ? c / gd : 0;
  }
  const float* featb = feat + (size_t)b * num_feat * C;
  float* gfeatb = g_feat + (size_t)b * num_feat * C;
  const float* wrow = weights + row * PK * L * G;

  for (int lvl = wave; lvl < L; lvl += kWaves) {
    unsigned long long ma = m0, mb = m1;
    while (ma | mb) {
      int i0;
      if (ma) { i0 = __builtin_ctzll(ma); ma &= ma - 1; } else { i0 = 64 + __builtin_ctzll(mb); mb &= mb - 1; }
      const float lx = i0 < 64 ? __shfl(l0.x, i0) : __shfl(l1.x, i0 - 64);
      const float ly = i0 < 64 ? __shfl(l0.y, i0) : __shfl(l1.y, i0 - 64);
      const int cam = i0 % num_cams;
      const int cs = cam * L + lvl;
      const int H = spatial_shape[2 * cs], W = spatial_shape[2 * cs + 1];
      const size_t base = (size_t)scale_start[cs] * C;
      const float h_im = (float)((double)(ly * (float)H) - 0.5);
      const float w_im = (float)((double)(lx * (float)W) - 0.5);
      const float hf = floorf(h_im), wf = floorf(w_im);
      const int h0 = (int)hf, w0 = (int)wf;
      const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
      const bool y0 = h0 >= 0, y1 = h0 + 1 <= H - 1, x0 = w0 >= 0, x1 = w0 + 1 <= W - 1;
      const bool t1 = y0 && x0, t2 = y0 && x1, t3 = y1 && x0, t4 = y1 && x1;
      const size_t p1 = base + (size_t)(h0 * W + w0) * C, p2 = p1 + C, p3 = p1 + (size_t)W * C, p4 = p3 + C;
      const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
      float gx = 0.f, gy = 0.f;
      float gwp[kNJ];
#pragma unroll
      for (int j = 0; j < kNJ; ++j) {
        gwp[j] = 0.f;
        if (j < nj) {
          const int c = lane + 64 * j;
          const float v1 = t1 ? featb[p1 + c] : 0.f, v2 = t2 ? featb[p2 + c] : 0.f;
          const float v3 = t3 ? featb[p3 + c] : 0.f, v4 = t4 ? featb[p4 + c] : 0.f;
          const float wgt = wrow[((size_t)i0 * L + lvl) * G + grp[j]];
          const float top = go[j] * wgt;  // cu:238
          if (t1) atomicAdd(gfeatb + p1 + c, w1 * top);
          if (t2) atomicAdd(gfeatb + p2 + c, w2 * top);
          if (t3) atomicAdd(gfeatb + p3 + c, w3 * top);
          if (t4) atomicAdd(gfeatb + p4 + c, w4 * top);
          gwp[j] = go[j] * (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);           // cu:122-123
          gx += top * (-hh * v1 + hh * v2 - lh * v3 + lh * v4);                // grad_w_weight (cu:92-118)
          gy += top * (-hw * v1 - lw * v2 + hw * v3 + lw * v4);                // grad_h_weight
        }
      }
      // weight gradient: channels of one group sit in whole 32-lane halves (gd % 32 == 0)
#pragma unroll
      for (int j = 0; j < kNJ; ++j) {
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) gwp[j] += __shfl_xor(gwp[j], m);
      }
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) { gx += __shfl_xor(gx, m); gy += __shfl_xor(gy, m); }
      float other[kNJ];
#pragma unroll
      for (int j = 0; j < kNJ; ++j) other[j] = __shfl_xor(gwp[j], 32);
      if (lane == 0) {  // each (sample, level, group) is visited exactly once: plain stores, fixed order
        float* gw = gw_row + ((size_t)i0 * L + lvl) * G;
        for (int g = 0; g < G; ++g) {
          float acc = 0.f;
#pragma unroll
          for (int j = 0; j < kNJ; ++j) {
            if (j < nj) {
              if ((64 * j) / gd == g) acc += gwp[j];
              if ((64 * j + 32) / gd == g) acc += other[j];
            }
          }
          gw[g] = acc;
        }
      }
      if (lane == 0) {
        s_gl[wave][i0][0] += (float)W * gx;   // cu:124
        s_gl[wave][i0][1] += (float)H * gy;   // cu:125
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < PK * 2; i += kThreads) {
    const int idx = i >> 1, xy = i & 1;
    g_loc[row * PK * 2 + i] = ((s_gl[0][idx][xy] + s_gl[1][idx][xy]) + (s_gl[2][idx][xy] + s_gl[3][idx][xy]));
  }
}

}  // namespace

extern "C" int simpb_deformable_aggregation_backward(
    float* grad_mc_ms_feat, float* grad_sampling_location, float* grad_weights, const float* mc_ms_feat,
    const int* spatial_shape, const int* scale_start_index, const float* sample_location, const float* weights,
    const float* grad_output, int batch_size, int num_cams, int num_feat, int num_embeds, int num_scale,
    int num_anchors, int num_pts, int num_groups, void* stream) {
  if (!grad_mc_ms_feat || !grad_sampling_location || !grad_weights || !mc_ms_feat || !spatial_shape ||
      !scale_start_index || !sample_location || !weights || !grad_output)
    return SIMPB_EINVAL;
  if (batch_size <= 0 || num_cams <= 0 || num_feat <= 0 || num_embeds <= 0 || num_scale <= 0 || num_anchors <= 0 ||
      num_pts <= 0 || num_groups <= 0 || num_embeds % num_groups != 0 || batch_size > 65535)
    return SIMPB_EINVAL;
  const int gd = num_embeds / num_groups;
  if (num_embeds % 64 != 0 || num_embeds > 256 || gd % 32 != 0 || num_pts * num_cams > kMaxPK) return SIMPB_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  (void)hipGetLastError();
  if (hipMemsetAsync(grad_mc_ms_feat, 0, (size_t)batch_size * num_feat * num_embeds * sizeof(float), s) != hipSuccess)
    return SIMPB_ELAUNCH;
  hipLaunchKernelGGL(daf_bwd_rows, dim3(num_anchors, batch_size), dim3(kThreads), 0, s, grad_mc_ms_feat,
                     grad_sampling_location, grad_weights, mc_ms_feat, spatial_shape, scale_start_index, sample_location,
                     weights, grad_output, num_cams, num_feat, num_embeds, num_scale, num_anchors, num_pts, num_groups);
  return simpb_check_launch();
}
