// Camera-grouped multi-scale deformable attention, forward (gfx950).
//
// Replaces the per-camera Python loop over mmcv's ms_deform_attn_forward
// (/root/reference/projects/mmdet3d_plugin/models/group_attn.py:222-235) with ONE launch: query
// slot q samples the value map of camera query_cam[q]. Per (query, head):
//   out[q, head, :] = sum over (lvl, pt) of attn[q,head,lvl,pt] * bilinear(value[cam, lvl, :, head, :], loc)
// Sampling rule = mmcv's CUDA op = grid_sample(bilinear, padding zeros, align_corners=False):
// pixel = loc*size - 0.5, each tap zero outside the map.
//
// Mapping: one workgroup of 4 waves per (batch, query); wave w takes levels w, w+4, ...; a lane
// owns 4 consecutive channels of one head, so one wave-instruction fetches, for every head, the
// 128-byte head slice of that head's own tap (8 heads x 128 B at the shipped 8x32 layout).
// All points of a level are issued before any is consumed (4 points x 4 taps in flight per lane).
#include <hip/hip_runtime.h>
#include "../../include/simpb_hip.h"

extern "C" int simpb_check_launch(void);
extern "C" int simpb_timing_begin(int kernel_id, void* stream);
extern "C" void simpb_timing_end(int slot, void* stream);

namespace {

constexpr int kWaves = 4;
constexpr int kThreads = kWaves * 64;
constexpr int kMaxPts = 8;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int PTS>
__device__ __forceinline__ void level_points(float4& acc, const float* __restrict__ vbase, int H, int W, int HC,
                                             int coff, const float2* __restrict__ locp,
                                             const float* __restrict__ attp) {
  float4 v[PTS][4];
  float tw[PTS][4];
#pragma unroll
  for (int pt = 0; pt < PTS; ++pt) {
    const float2 l = locp[pt];
    const float aw = attp[pt];
    const float h_im = l.y * (float)H - 0.5f;
    const float w_im = l.x * (float)W - 0.5f;
    const float hf = floorf(h_im), wf = floorf(w_im);
    // far-away locations: keep the int conversion defined, every tap ends up invalid anyway
    const int h0 = (int)fminf(fmaxf(hf, -2.f), (float)H);
    const int w0 = (int)fminf(fmaxf(wf, -2.f), (float)W);
    const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
    const bool y0 = h0 >= 0 && h0 <= H - 1, y1 = h0 + 1 >= 0 && h0 + 1 <= H - 1;
    const bool x0 = w0 >= 0 && w0 <= W - 1, x1 = w0 + 1 >= 0 && w0 + 1 <= W - 1;
    const int yc0 = min(max(h0, 0), H - 1), yc1 = min(max(h0 + 1, 0), H - 1);
    const int xc0 = min(max(w0, 0), W - 1), xc1 = min(max(w0 + 1, 0), W - 1);
    tw[pt][0] = (y0 && x0) ? aw * hh * hw : 0.f;
    tw[pt][1] = (y0 && x1) ? aw * hh * lw : 0.f;
    tw[pt][2] = (y1 && x0) ? aw * lh * hw : 0.f;
    tw[pt][3] = (y1 && x1) ? aw * lh * lw : 0.f;
    v[pt][0] = ld4(vbase + (size_t)(yc0 * W + xc0) * HC + coff);
    v[pt][1] = ld4(vbase + (size_t)(yc0 * W + xc1) * HC + coff);
    v[pt][2] = ld4(vbase + (size_t)(yc1 * W + xc0) * HC + coff);
    v[pt][3] = ld4(vbase + (size_t)(yc1 * W + xc1) * HC + coff);
  }
#pragma unroll
  for (int pt = 0; pt < PTS; ++pt)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      acc.x += tw[pt][k] * v[pt][k].x;
      acc.y += tw[pt][k] * v[pt][k].y;
      acc.z += tw[pt][k] * v[pt][k].z;
      acc.w += tw[pt][k] * v[pt][k].w;
    }
}

template <int PTS>
__global__ __launch_bounds__(kThreads) void msda_grouped_fwd(
    float* __restrict__ out, const float* __restrict__ value, const long long* __restrict__ spatial_shapes,
    const long long* __restrict__ level_start, const float* __restrict__ loc, const float* __restrict__ attn,
    const int* __restrict__ query_cam, int num_cams, int num_value, int heads, int ch, int L, int P, int nq) {
  __shared__ float4 s_red[kWaves][64];
  const int q = blockIdx.x, b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int HC = heads * ch;
  int cam = query_cam[q];
  if (cam < 0) {  // capacity slot outside every camera group: defined output, no sampling
    for (int c = threadIdx.x; c < HC; c += kThreads) out[((size_t)b * nq + q) * HC + c] = 0.f;
    return;
  }
  cam = min(cam, num_cams - 1);
  const float* vcam = value + ((size_t)b * num_cams + cam) * num_value * HC;
  const size_t qrow = (size_t)b * nq + q;

  for (int cbase = 0; cbase < HC; cbase += 256) {  // one pass at the shipped 8 heads x 32 channels
    const int coff = cbase + lane * 4;
    const bool active = coff < HC;
    const int ld_off = active ? coff : 0;
    const int head = ld_off / ch;
    const float2* locq = reinterpret_cast<const float2*>(loc) + (qrow * heads + head) * L * P;
    const float* attq = attn + (qrow * heads + head) * L * P;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int lvl = wave; lvl < L; lvl += kWaves) {
      const int H = (int)spatial_shapes[2 * lvl], W = (int)spatial_shapes[2 * lvl + 1];
      const float* vbase = vcam + (size_t)level_start[lvl] * HC;
      if (PTS > 0) {
        level_points<(PTS > 0 ? PTS : 1)>(acc, vbase, H, W, HC, ld_off, locq + lvl * P, attq + lvl * P);
      } else {
        for (int pt = 0; pt < P; ++pt) level_points<1>(acc, vbase, H, W, HC, ld_off, locq + lvl * P + pt, attq + lvl * P + pt);
      }
    }
    if (cbase) __syncthreads();
    s_red[wave][lane] = acc;
    __syncthreads();
    const int c = cbase + threadIdx.x;
    if (c < HC) {
      const float* r = reinterpret_cast<const float*>(s_red);
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) s += r[w * 256 + threadIdx.x];
      out[qrow * HC + c] = s;
    }
  }
}

}  // namespace

extern "C" int simpb_ms_deform_attn_grouped_forward(
    float* output, const float* value, const long long* spatial_shapes, const long long* level_start,
    const float* sampling_loc, const float* attn_weight, const int* query_cam, int batch_size, int num_cams,
    int num_value, int num_heads, int channels, int num_levels, int num_points, int num_query, void* stream) {
  if (!output || !value || !spatial_shapes || !level_start || !sampling_loc || !attn_weight || !query_cam)
    return SIMPB_EINVAL;
  if (batch_size <= 0 || num_cams <= 0 || num_value <= 0 || num_heads <= 0 || channels <= 0 || num_levels <= 0 ||
      num_points <= 0 || num_query <= 0 || batch_size > 65535 || channels % 4 != 0)
    return SIMPB_EINVAL;
  (void)hipGetLastError();  // drop a stale error left by earlier runtime calls of the caller
  hipStream_t s = static_cast<hipStream_t>(stream);
  dim3 grid(num_query, batch_size), block(kThreads);
  const int tslot = simpb_timing_begin(SIMPB_KERNEL_MSDA, stream);
#define SIMPB_MSDA_LAUNCH(PTS)                                                                                   \
  hipLaunchKernelGGL(msda_grouped_fwd<PTS>, grid, block, 0, s, output, value, spatial_shapes, level_start,       \
                     sampling_loc, attn_weight, query_cam, num_cams, num_value, num_heads, channels, num_levels, \
                     num_points, num_query)
  if (num_points == 4) SIMPB_MSDA_LAUNCH(4);
  else if (num_points == 8) SIMPB_MSDA_LAUNCH(8);
  else SIMPB_MSDA_LAUNCH(0);
#undef SIMPB_MSDA_LAUNCH
  simpb_timing_end(tslot, stream);
  return simpb_check_launch();
}
