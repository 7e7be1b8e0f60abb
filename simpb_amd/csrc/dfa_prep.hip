// Producers of the deformable-aggregation operands (gfx950): sampling locations and weights written
// directly in the layouts the aggregation kernel reads.
//
//  * dfa_points: SparseBox3DKeyPointsGenerator.forward
//    (/root/reference/projects/mmdet3d_plugin/models/detection3d/blocks.py:181-222) +
//    DeformableFeatureAggregation.project_points (models/blocks.py:198-213) + the permute at
//    blocks.py:124-131. The reference does this with a batched [bs*cams*A*P] x (4x4 @ 4x1) matmul
//    (measured 1 020 us through the vendor batched GEMM) and a dozen elementwise kernels.
//  * dfa_weights: the softmax over (cam, lvl, pt) per group of blocks.py:177-187 and the permute to
//    [a, pt, cam, lvl, group] of blocks.py:132-143, with the camera term added on the fly:
//    weights_fc(feature + cam_embed) = weights_fc(feature) + cam_embed @ W^T.
#include <hip/hip_runtime.h>
#include "../../include/simpb_hip.h"
#include "store_fence.h"

extern "C" int simpb_check_launch(void);

namespace {

// thread per (b, a, p): computes the key point once, projects it into every camera
__global__ void dfa_points_kernel(float* __restrict__ loc, float* __restrict__ key_points,
                                  const float* __restrict__ anchor, const float* __restrict__ learn,
                                  const float* __restrict__ fix_scale, const float* __restrict__ proj,
                                  const float* __restrict__ image_wh, int bs, int A, int num_fix, int num_learn,
                                  int cams) {
  const int P = num_fix + num_learn;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= bs * A * P) return;
  const int p = idx % P, a = (idx / P) % A, b = idx / (P * A);
  // store_fence.h discipline: every consumer of loaded data sits behind a full vmcnt(0), and no store is in flight
  // while a counted wait releases consumers (this kernel, scheduled freely, stored camera c's pair while camera c+1's
  // matrix rows were on their way behind counted waits, and faulted in lanes 48-63 beside a busy second queue)
  const float* an = anchor + ((size_t)b * A + a) * 11;
  float av[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) av[k] = an[k];
  float f0 = 0.f, f1 = 0.f, f2 = 0.f;
  if (p < num_fix) {
    f0 = fix_scale[p * 3 + 0]; f1 = fix_scale[p * 3 + 1]; f2 = fix_scale[p * 3 + 2];
  } else {
    const float* l = learn + (((size_t)b * A + a) * num_learn + (p - num_fix)) * 3;
    f0 = l[0]; f1 = l[1]; f2 = l[2];
  }
  simpb::loads_retired();
  const float sw = expf(av[3]), sl = expf(av[4]), sh = expf(av[5]);
  float kx, ky, kz;
  if (p < num_fix) {
    kx = f0 * sw; ky = f1 * sl; kz = f2 * sh;
  } else {
    kx = (1.f / (1.f + expf(-f0)) - 0.5f) * sw;
    ky = (1.f / (1.f + expf(-f1)) - 0.5f) * sl;
    kz = (1.f / (1.f + expf(-f2)) - 0.5f) * sh;
  }
  const float sn = av[6], cs = av[7];
  float px = cs * kx - sn * ky + av[0];
  float py = sn * kx + cs * ky + av[1];
  float pz = kz + av[2];
  if (key_points) {
    simpb::pin(px); simpb::pin(py); simpb::pin(pz);
    float* kp = key_points + (size_t)idx * 3;
    kp[0] = px; kp[1] = py; kp[2] = pz;
  }
  for (int c = 0; c < cams; ++c) {
    const float* M = proj + ((size_t)b * cams + c) * 16;
    const float* wh = image_wh + ((size_t)b * cams + c) * 2;
    float m[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) m[k] = M[k];
    const float w0 = wh[0], w1 = wh[1];
    simpb::loads_retired();  // also retires the previous camera's stores
    const float u = m[0] * px + m[1] * py + m[2] * pz + m[3];
    const float v = m[4] * px + m[5] * py + m[6] * pz + m[7];
    const float d = fmaxf(m[8] * px + m[9] * py + m[10] * pz + m[11], 1e-5f);
    float r0 = u / d / w0, r1 = v / d / w1;
    simpb::pin(r0); simpb::pin(r1);
    float* o = loc + ((size_t)idx * cams + c) * 2;
    o[0] = r0;
    o[1] = r1;
  }
}

// workgroup per (b, a); thread t -> (group g = t % G, slice) ; softmax over the cams*L*P entries of a group
__global__ __launch_bounds__(256) void dfa_weights_kernel(float* __restrict__ w_out,
                                                          const float* __restrict__ feat_logits,
                                                          const float* __restrict__ cam_logits, int A, int cams,
                                                          int L, int P, int G) {
  extern __shared__ float s_val[];  // [cams*L*P][G]
  __shared__ float s_red[256];
  const int a = blockIdx.x, b = blockIdx.y;
  const int LPG = L * P * G;
  const int n = cams * L * P;  // softmax length per group
  const float* fl = feat_logits + ((size_t)b * A + a) * LPG;
  const float* cl = cam_logits + (size_t)b * cams * LPG;
  const int tid = threadIdx.x;
  const int g = tid % G, slice = tid / G, slices = blockDim.x / G;
  // pass 1: logits into LDS, running max per (thread)
  float m = -INFINITY;
  for (int e = slice; e < n; e += slices) {  // e = (cam, lvl, pt) flattened cam-major
    const int cam = e / (L * P), lp = e - cam * (L * P);
    const float v = fl[lp * G + g] + cl[(size_t)cam * LPG + lp * G + g];
    s_val[e * G + g] = v;
    m = fmaxf(m, v);
  }
  s_red[tid] = m;
  __syncthreads();
  for (int s = slices / 2; s >= 1; s >>= 1) {
    if (slice < s) s_red[tid] = fmaxf(s_red[tid], s_red[tid + s * G]);
    __syncthreads();
  }
  m = s_red[g];
  __syncthreads();
  float sum = 0.f;
  for (int e = slice; e < n; e += slices) {
    const float v = expf(s_val[e * G + g] - m);
    s_val[e * G + g] = v;
    sum += v;
  }
  s_red[tid] = sum;
  __syncthreads();
  for (int s = slices / 2; s >= 1; s >>= 1) {
    if (slice < s) s_red[tid] += s_red[tid + s * G];
    __syncthreads();
  }
  const float inv = 1.f / s_red[g];
  // write [a, pt, cam, lvl, g]
  float* wo = w_out + ((size_t)b * A + a) * (size_t)n * G;
  for (int e = slice; e < n; e += slices) {
    const int cam = e / (L * P), lp = e - cam * (L * P);
    const int lvl = lp / P, pt = lp - lvl * P;
    wo[(((size_t)pt * cams + cam) * L + lvl) * G + g] = s_val[e * G + g] * inv;
  }
}

}  // namespace

extern "C" int simpb_dfa_points(float* loc, float* key_points, const float* anchor, const float* learnable,
                                const float* fix_scale, const float* projection_mat, const float* image_wh,
                                int batch_size, int num_anchors, int num_fix, int num_learn, int num_cams,
                                void* stream) {
  if (!loc || !anchor || !fix_scale || !projection_mat || !image_wh || batch_size <= 0 || num_anchors <= 0 ||
      num_fix < 0 || num_learn < 0 || num_fix + num_learn <= 0 || num_cams <= 0 || (num_learn > 0 && !learnable))
    return SIMPB_EINVAL;
  (void)hipGetLastError();
  const int n = batch_size * num_anchors * (num_fix + num_learn);
  hipLaunchKernelGGL(dfa_points_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), loc,
                     key_points, anchor, learnable, fix_scale, projection_mat, image_wh, batch_size, num_anchors, num_fix,
                     num_learn, num_cams);
  return simpb_check_launch();
}

extern "C" int simpb_dfa_weights(float* weights, const float* feat_logits, const float* cam_logits, int batch_size,
                                 int num_anchors, int num_cams, int num_levels, int num_pts, int num_groups,
                                 void* stream) {
  if (!weights || !feat_logits || !cam_logits || batch_size <= 0 || num_anchors <= 0 || num_cams <= 0 ||
      num_levels <= 0 || num_pts <= 0 || num_groups <= 0 || batch_size > 65535)
    return SIMPB_EINVAL;
  // blockDim = G * slices with slices a power of two (tree reduction), at most 256 threads
  int slices = 1;
  while (slices * 2 * num_groups <= 256) slices *= 2;
  if (num_groups > 256) return SIMPB_EINVAL;
  const size_t lds = (size_t)num_cams * num_levels * num_pts * num_groups * sizeof(float);
  if (lds > 60 * 1024) return SIMPB_EINVAL;
  (void)hipGetLastError();
  hipLaunchKernelGGL(dfa_weights_kernel, dim3(num_anchors, batch_size), dim3(num_groups * slices), lds,
                     static_cast<hipStream_t>(stream), weights, feat_logits, cam_logits, num_anchors, num_cams,
                     num_levels, num_pts, num_groups);
  return simpb_check_launch();
}
