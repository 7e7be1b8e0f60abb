// Camera-grouped multi-scale deformable attention, backward (gfx950): gradients w.r.t. value,
// sampling locations and attention weights -- what mmcv's ms_deform_attn_backward computes for each
// camera group in the reference's loop (/root/reference/projects/mmdet3d_plugin/models/
// group_attn.py:227-235, through MultiScaleDeformableAttnFunction.backward) [mmcv-memory: the
// analytic derivative of bilinear sampling with zero padding, align_corners=False].
//
// Same mapping as the forward (workgroup per (batch, query), wave per level, lane = 4 channels of one
// head): the attention-weight and location gradients of a (query, head, level, point) are visited
// exactly once, so they are 8-lane register reductions + plain stores; only grad_value is a scatter
// and uses float atomics.
#include <hip/hip_runtime.h>
#include "../../include/simpb_hip.h"

extern "C" int simpb_check_launch(void);

namespace {

constexpr int kWaves = 4;
constexpr int kThreads = kWaves * 64;

__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void atomic_add4(float* p, const float4& v, float s) {
  atomicAdd(p + 0, v.x * s); atomicAdd(p + 1, v.y * s); atomicAdd(p + 2, v.z * s); atomicAdd(p + 3, v.w * s);
}

__global__ __launch_bounds__(kThreads) void msda_grouped_bwd(
    float* __restrict__ g_value, float* __restrict__ g_loc, float* __restrict__ g_attn, const float* __restrict__ value,
    const long long* __restrict__ spatial_shapes, const long long* __restrict__ level_start,
    const float* __restrict__ loc, const float* __restrict__ attn, const int* __restrict__ query_cam,
    const float* __restrict__ g_out, int num_cams, int num_value, int heads, int ch, int L, int P, int nq) {
  const int q = blockIdx.x, b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int HC = heads * ch;            // == 256 on this path: 64 lanes x 4 channels
  const int lanes_per_head = ch / 4;    // power of two (checked on the host)
  const size_t qrow = (size_t)b * nq + q;
  const int cam = query_cam[q];
  const int coff = lane * 4;
  const int head = coff / ch;
  float* gl = g_loc + ((qrow * heads + head) * L) * P * 2;
  float* ga = g_attn + ((qrow * heads + head) * L) * P;
  const bool head_lead = (lane % lanes_per_head) == 0;
  if (cam < 0) {  // capacity slot outside every camera group: zero gradients
    for (int lvl = wave; lvl < L; lvl += kWaves)
      if (head_lead)
        for (int pt = 0; pt < P; ++pt) { ga[lvl * P + pt] = 0.f; gl[(lvl * P + pt) * 2] = 0.f; gl[(lvl * P + pt) * 2 + 1] = 0.f; }
    return;
  }
  const size_t cam_off = ((size_t)b * num_cams + min(cam, num_cams - 1)) * num_value * HC;
  const float* vcam = value + cam_off;
  float* gvcam = g_value + cam_off;
  const float4 go = ld4(g_out + qrow * HC + coff);
  const float2* locq = reinterpret_cast<const float2*>(loc) + (qrow * heads + head) * L * P;
  const float* attq = attn + (qrow * heads + head) * L * P;

  for (int lvl = wave; lvl < L; lvl += kWaves) {
    const int H = (int)spatial_shapes[2 * lvl], W = (int)spatial_shapes[2 * lvl + 1];
    const size_t lbase = (size_t)level_start[lvl] * HC;
    for (int pt = 0; pt < P; ++pt) {
      const float2 l = locq[lvl * P + pt];
      const float aw = attq[lvl * P + pt];
      const float h_im = l.y * (float)H - 0.5f, w_im = l.x * (float)W - 0.5f;
      const float hf = floorf(h_im), wf = floorf(w_im);
      const int h0 = (int)fminf(fmaxf(hf, -2.f), (float)H), w0 = (int)fminf(fmaxf(wf, -2.f), (float)W);
      const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
      const bool y0 = h0 >= 0 && h0 <= H - 1, y1 = h0 + 1 >= 0 && h0 + 1 <= H - 1;
      const bool x0 = w0 >= 0 && w0 <= W - 1, x1 = w0 + 1 >= 0 && w0 + 1 <= W - 1;
      const bool t1 = y0 && x0, t2 = y0 && x1, t3 = y1 && x0, t4 = y1 && x1;
      const size_t p1 = lbase + (size_t)(h0 * W + w0) * HC + coff, p2 = p1 + HC, p3 = p1 + (size_t)W * HC, p4 = p3 + HC;
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 v1 = t1 ? ld4(vcam + p1) : z, v2 = t2 ? ld4(vcam + p2) : z;
      const float4 v3 = t3 ? ld4(vcam + p3) : z, v4 = t4 ? ld4(vcam + p4) : z;
      if (t1) atomic_add4(gvcam + p1, go, aw * hh * hw);
      if (t2) atomic_add4(gvcam + p2, go, aw * hh * lw);
      if (t3) atomic_add4(gvcam + p3, go, aw * lh * hw);
      if (t4) atomic_add4(gvcam + p4, go, aw * lh * lw);
      const float d1 = dot4(go, v1), d2 = dot4(go, v2), d3 = dot4(go, v3), d4 = dot4(go, v4);
      float ga_p = hh * hw * d1 + hh * lw * d2 + lh * hw * d3 + lh * lw * d4;
      float gx_p = aw * (-hh * d1 + hh * d2 - lh * d3 + lh * d4) * (float)W;
      float gy_p = aw * (-hw * d1 - lw * d2 + hw * d3 + lw * d4) * (float)H;
      for (int m = lanes_per_head / 2; m >= 1; m >>= 1) {
        ga_p += __shfl_xor(ga_p, m); gx_p += __shfl_xor(gx_p, m); gy_p += __shfl_xor(gy_p, m);
      }
      if (head_lead) {
        ga[lvl * P + pt] = ga_p;
        gl[(lvl * P + pt) * 2] = gx_p;
        gl[(lvl * P + pt) * 2 + 1] = gy_p;
      }
    }
  }
}

}  // namespace

extern "C" int simpb_ms_deform_attn_grouped_backward(
    float* grad_value, float* grad_sampling_loc, float* grad_attn_weight, const float* value,
    const long long* spatial_shapes, const long long* level_start, const float* sampling_loc, const float* attn_weight,
    const int* query_cam, const float* grad_output, int batch_size, int num_cams, int num_value, int num_heads,
    int channels, int num_levels, int num_points, int num_query, void* stream) {
  if (!grad_value || !grad_sampling_loc || !grad_attn_weight || !value || !spatial_shapes || !level_start ||
      !sampling_loc || !attn_weight || !query_cam || !grad_output)
    return SIMPB_EINVAL;
  if (batch_size <= 0 || num_cams <= 0 || num_value <= 0 || num_heads <= 0 || channels <= 0 || num_levels <= 0 ||
      num_points <= 0 || num_query <= 0 || batch_size > 65535)
    return SIMPB_EINVAL;
  const int lph = channels / 4;
  if (channels % 4 != 0 || num_heads * channels != 256 || (lph & (lph - 1)) != 0) return SIMPB_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  (void)hipGetLastError();
  if (hipMemsetAsync(grad_value, 0, (size_t)batch_size * num_cams * num_value * num_heads * channels * sizeof(float), s) !=
      hipSuccess)
    return SIMPB_ELAUNCH;
  hipLaunchKernelGGL(msda_grouped_bwd, dim3(num_query, batch_size), dim3(kThreads), 0, s, grad_value, grad_sampling_loc,
                     grad_attn_weight, value, spatial_shapes, level_start, sampling_loc, attn_weight, query_cam,
                     grad_output, num_cams, num_value, num_heads, channels, num_levels, num_points, num_query);
  return simpb_check_launch();
}
