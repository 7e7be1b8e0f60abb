// 3D deformable feature aggregation, forward (gfx950).
//
// Semantics follow deformable_aggregation_kernel
// (/root/reference/projects/mmdet3d_plugin/ops/src/deformable_aggregation_cuda.cu:129-187):
//   out[b,a,c] = sum over (p, cam, lvl) of w[b,a,p,cam,lvl,c/(C/G)] * bilinear(feat, loc[b,a,p,cam])
// with a sample dropped unless 0 < x < 1 and 0 < y < 1 (:169-171), pixel = loc*size - 0.5
// (:180-181, the 0.5 is a double literal there) and each tap zero outside the map (:35-53).
//
// Mapping (not the reference's one-thread-per-(a,p,cam,lvl,c) + atomicAdd): one workgroup of 4
// waves per (batch, anchor). Every wave tests all P*cam locations itself (two ballots, no LDS,
// no barrier) and walks the valid ones with scalar bit scans; wave w takes levels w, w+4, ...
// so a wave's map geometry is scalar. One lane owns 4 consecutive channels: a tap is one
// 16-byte load per lane = one coalesced 1-KiB row per wave-instruction at C = 256. Sums stay in
// registers; the 4 waves meet once in LDS and the row is written once -> deterministic.
#include <hip/hip_runtime.h>
#include "store_fence.h"
#include "../../include/simpb_hip.h"

extern "C" int simpb_check_launch(void);
extern "C" int simpb_timing_begin(int kernel_id, void* stream);
extern "C" void simpb_timing_end(int slot, void* stream);

namespace {

constexpr int kWaves = 4;
constexpr int kThreads = kWaves * 64;

struct Tap4 {
  float4 v00, v01, v10, v11;
  float w00, w01, w10, w11;
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// Issue the 4 tap loads of one sample. Taps outside the map are redirected to a clamped (valid)
// address and their weight is zeroed, so the loads carry no branch and can all be in flight.
__device__ __forceinline__ void issue_taps(Tap4& t, const float* __restrict__ base, int H, int W, int C,
                                           float lx, float ly, int coff) {
  const float h_im = (float)((double)(ly * (float)H) - 0.5);
  const float w_im = (float)((double)(lx * (float)W) - 0.5);
  const float hf = floorf(h_im), wf = floorf(w_im);
  const int h0 = (int)hf, w0 = (int)wf;
  const float lh = h_im - hf, lw = w_im - wf;
  const float hh = 1.f - lh, hw = 1.f - lw;
  const bool y0 = h0 >= 0, y1 = h0 + 1 <= H - 1, x0 = w0 >= 0, x1 = w0 + 1 <= W - 1;
  const int yc0 = max(h0, 0), yc1 = min(h0 + 1, H - 1), xc0 = max(w0, 0), xc1 = min(w0 + 1, W - 1);
  t.w00 = (y0 && x0) ? hh * hw : 0.f;
  t.w01 = (y0 && x1) ? hh * lw : 0.f;
  t.w10 = (y1 && x0) ? lh * hw : 0.f;
  t.w11 = (y1 && x1) ? lh * lw : 0.f;
  t.v00 = ld4(base + (size_t)(yc0 * W + xc0) * C + coff);
  t.v01 = ld4(base + (size_t)(yc0 * W + xc1) * C + coff);
  t.v10 = ld4(base + (size_t)(yc1 * W + xc0) * C + coff);
  t.v11 = ld4(base + (size_t)(yc1 * W + xc1) * C + coff);
}

__device__ __forceinline__ void accumulate(float4& acc, const Tap4& t, float wgt) {
  acc.x += wgt * (t.w00 * t.v00.x + t.w01 * t.v01.x + t.w10 * t.v10.x + t.w11 * t.v11.x);
  acc.y += wgt * (t.w00 * t.v00.y + t.w01 * t.v01.y + t.w10 * t.v10.y + t.w11 * t.v11.y);
  acc.z += wgt * (t.w00 * t.v00.z + t.w01 * t.v01.z + t.w10 * t.v10.z + t.w11 * t.v11.z);
  acc.w += wgt * (t.w00 * t.v00.w + t.w01 * t.v01.w + t.w10 * t.v10.w + t.w11 * t.v11.w);
}

// Fast path: C % 4 == 0, C <= 256, (C/G) % 4 == 0, P*cams <= 128.
__global__ __launch_bounds__(kThreads) void daf_fwd_rows(
    float* __restrict__ out, const float* __restrict__ feat, const int* __restrict__ spatial_shape,
    const int* __restrict__ scale_start, const float* __restrict__ loc, const float* __restrict__ weights,
    int num_cams, int num_feat, int C, int L, int A, int P, int G) {
  __shared__ float4 s_red[kWaves][64];
  const int a = blockIdx.x, b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int PK = P * num_cams;
  const size_t row = (size_t)b * A + a;

  const float2* loc2 = reinterpret_cast<const float2*>(loc) + row * PK;
  float2 l0 = make_float2(-1.f, -1.f), l1 = make_float2(-1.f, -1.f);
  if (lane < PK) l0 = loc2[lane];
  if (lane + 64 < PK) l1 = loc2[lane + 64];
  const unsigned long long m0 = __ballot(l0.x > 0.f && l0.x < 1.f && l0.y > 0.f && l0.y < 1.f);
  const unsigned long long m1 = __ballot(l1.x > 0.f && l1.x < 1.f && l1.y > 0.f && l1.y < 1.f);

  const int coff = lane * 4;
  const bool active = coff < C;
  const int ld_off = active ? coff : 0;
  const int g = ld_off / (C / G);
  const float* featb = feat + (size_t)b * num_feat * C;
  const float* wrow = weights + row * PK * L * G + g;

  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int lvl = wave; lvl < L; lvl += kWaves) {
    unsigned long long ma = m0, mb = m1;
    while (ma | mb) {
      // pop up to two valid samples and keep their 8 row loads in flight together
      int i0, i1 = -1;
      if (ma) { i0 = __builtin_ctzll(ma); ma &= ma - 1; } else { i0 = 64 + __builtin_ctzll(mb); mb &= mb - 1; }
      if (ma) { i1 = __builtin_ctzll(ma); ma &= ma - 1; } else if (mb) { i1 = 64 + __builtin_ctzll(mb); mb &= mb - 1; }

      const float x0 = i0 < 64 ? __shfl(l0.x, i0) : __shfl(l1.x, i0 - 64);
      const float y0 = i0 < 64 ? __shfl(l0.y, i0) : __shfl(l1.y, i0 - 64);
      const int cam0 = i0 % num_cams;
      const int cs0 = cam0 * L + lvl;
      Tap4 t0, t1;
      issue_taps(t0, featb + (size_t)scale_start[cs0] * C, spatial_shape[2 * cs0], spatial_shape[2 * cs0 + 1], C,
                 x0, y0, ld_off);
      const float wg0 = wrow[((size_t)i0 * L + lvl) * G];
      float wg1 = 0.f;
      if (i1 >= 0) {
        const float x1 = i1 < 64 ? __shfl(l0.x, i1) : __shfl(l1.x, i1 - 64);
        const float y1 = i1 < 64 ? __shfl(l0.y, i1) : __shfl(l1.y, i1 - 64);
        const int cam1 = i1 % num_cams;
        const int cs1 = cam1 * L + lvl;
        issue_taps(t1, featb + (size_t)scale_start[cs1] * C, spatial_shape[2 * cs1], spatial_shape[2 * cs1 + 1], C,
                   x1, y1, ld_off);
        wg1 = wrow[((size_t)i1 * L + lvl) * G];
      }
      accumulate(acc, t0, wg0);
      if (i1 >= 0) accumulate(acc, t1, wg1);
    }
  }
  s_red[wave][lane] = acc;
  __syncthreads();
  // 256 threads, one channel each: thread t sums component (t & 3) of lane (t >> 2) over waves
  const int c = threadIdx.x;
  if (c < C) {
    const float* r = reinterpret_cast<const float*>(s_red);
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) s += r[w * 256 + c];
    out[row * C + c] = s;
  }
}

// Generic path for any (C, G, P*cams): one thread per channel, plain loops.
__global__ void daf_fwd_generic(float* __restrict__ out, const float* __restrict__ feat,
                                const int* __restrict__ spatial_shape, const int* __restrict__ scale_start,
                                const float* __restrict__ loc, const float* __restrict__ weights, int num_cams,
                                int num_feat, int C, int L, int A, int P, int G) {
  const int a = blockIdx.x, b = blockIdx.y;
  const size_t row = (size_t)b * A + a;
  const float* featb = feat + (size_t)b * num_feat * C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const int g = c / (C / G);
    float acc = 0.f;
    for (int p = 0; p < P; ++p)
      for (int cam = 0; cam < num_cams; ++cam) {
        const size_t li = (row * P + p) * num_cams + cam;
        const float lx = loc[2 * li], ly = loc[2 * li + 1];
        if (!(lx > 0.f && lx < 1.f && ly > 0.f && ly < 1.f)) continue;
        for (int lvl = 0; lvl < L; ++lvl) {
          const int cs = cam * L + lvl;
          const int H = spatial_shape[2 * cs], W = spatial_shape[2 * cs + 1];
          const float* base = featb + (size_t)scale_start[cs] * C + c;
          const float h_im = (float)((double)(ly * (float)H) - 0.5);
          const float w_im = (float)((double)(lx * (float)W) - 0.5);
          const float hf = floorf(h_im), wf = floorf(w_im);
          const int h0 = (int)hf, w0 = (int)wf;
          const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
          float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
          if (h0 >= 0 && w0 >= 0) v1 = base[(size_t)(h0 * W + w0) * C];
          if (h0 >= 0 && w0 + 1 <= W - 1) v2 = base[(size_t)(h0 * W + w0 + 1) * C];
          if (h0 + 1 <= H - 1 && w0 >= 0) v3 = base[(size_t)((h0 + 1) * W + w0) * C];
          if (h0 + 1 <= H - 1 && w0 + 1 <= W - 1) v4 = base[(size_t)((h0 + 1) * W + w0 + 1) * C];
          acc += weights[(li * L + lvl) * G + g] * (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4);
        }
      }
    simpb::pin(acc);
    simpb::loads_retired();  // store_fence.h (channel loop: the next channel's loads start with nothing in flight)
    out[row * C + c] = acc;
    simpb::loads_retired();
  }
}

}  // namespace

extern "C" int simpb_deformable_aggregation_forward(
    float* output, const float* mc_ms_feat, const int* spatial_shape, const int* scale_start_index,
    const float* sample_location, const float* weights, int batch_size, int num_cams, int num_feat, int num_embeds,
    int num_scale, int num_anchors, int num_pts, int num_groups, void* stream) {
  if (!output || !mc_ms_feat || !spatial_shape || !scale_start_index || !sample_location || !weights)
    return SIMPB_EINVAL;
  if (batch_size <= 0 || num_cams <= 0 || num_feat <= 0 || num_embeds <= 0 || num_scale <= 0 || num_anchors <= 0 ||
      num_pts <= 0 || num_groups <= 0 || num_embeds % num_groups != 0 || batch_size > 65535)
    return SIMPB_EINVAL;
  (void)hipGetLastError();  // drop a stale error left by earlier runtime calls of the caller
  hipStream_t s = static_cast<hipStream_t>(stream);
  dim3 grid(num_anchors, batch_size);
  const int gd = num_embeds / num_groups;
  const int tslot = simpb_timing_begin(SIMPB_KERNEL_DAF, stream);
  const bool fast = num_embeds % 4 == 0 && num_embeds <= 256 && gd % 4 == 0 && num_pts * num_cams <= 128;
  if (fast) {
    hipLaunchKernelGGL(daf_fwd_rows, grid, dim3(kThreads), 0, s, output, mc_ms_feat, spatial_shape, scale_start_index,
                       sample_location, weights, num_cams, num_feat, num_embeds, num_scale, num_anchors, num_pts,
                       num_groups);
  } else {
    const int threads = num_embeds >= 256 ? 256 : ((num_embeds + 63) / 64) * 64;
    hipLaunchKernelGGL(daf_fwd_generic, grid, dim3(threads), 0, s, output, mc_ms_feat, spatial_shape,
                       scale_start_index, sample_location, weights, num_cams, num_feat, num_embeds, num_scale,
                       num_anchors, num_pts, num_groups);
  }
  simpb_timing_end(tslot, stream);
  return simpb_check_launch();
}

// 2: simpb_mlp_chain gained the post stage (8 chains per launch); 3: simpb_bank_cache takes the hold flags, the 3D
// record carries the int64 track id in two lanes (15 columns); 5: simpb_gemm_job gained the LayerNorm prologue fields,
// new entry points simpb_dfa_fused_forward / simpb_aggregate_2d_to_3d_alpha
extern "C" int simpb_abi_version(void) { return 7; }

// ---- optional per-launch HIP-event timing (bench.py's roofline leg). Events are recorded on the
// launch stream immediately around the kernel launch, inside the same C call, so the interval
// holds the kernel and not the host's time to get from one Python statement to the next.
namespace {
struct TimingSlot { hipEvent_t start, stop; int kernel_id; };
TimingSlot* g_slots = nullptr;
int g_capacity = 0, g_used = 0;
}  // namespace

extern "C" int simpb_timing_enable(int capacity) {
  if (capacity < 0) return SIMPB_EINVAL;
  for (int i = 0; i < g_capacity; ++i) { hipEventDestroy(g_slots[i].start); hipEventDestroy(g_slots[i].stop); }
  delete[] g_slots;
  g_slots = nullptr; g_capacity = 0; g_used = 0;
  if (capacity == 0) return SIMPB_OK;
  g_slots = new TimingSlot[capacity];
  for (int i = 0; i < capacity; ++i) {
    if (hipEventCreate(&g_slots[i].start) != hipSuccess || hipEventCreate(&g_slots[i].stop) != hipSuccess) return SIMPB_ELAUNCH;
    g_slots[i].kernel_id = -1;
  }
  g_capacity = capacity;
  return SIMPB_OK;
}

extern "C" int simpb_timing_begin(int kernel_id, void* stream) {  // returns slot or -1
  if (g_used >= g_capacity) return -1;
  const int i = g_used++;
  g_slots[i].kernel_id = kernel_id;
  hipEventRecord(g_slots[i].start, static_cast<hipStream_t>(stream));
  return i;
}

extern "C" void simpb_timing_end(int slot, void* stream) {
  if (slot >= 0 && slot < g_capacity) hipEventRecord(g_slots[slot].stop, static_cast<hipStream_t>(stream));
}

extern "C" int simpb_timing_read(int kernel_id, float* ms_out, int max_n) {
  if (!ms_out || max_n < 0) return -1;
  int n = 0;
  for (int i = 0; i < g_used && n < max_n; ++i) {
    if (g_slots[i].kernel_id != kernel_id) continue;
    float ms = 0.f;
    if (hipEventSynchronize(g_slots[i].stop) != hipSuccess) return -1;
    if (hipEventElapsedTime(&ms, g_slots[i].start, g_slots[i].stop) != hipSuccess) return -1;
    ms_out[n++] = ms;
  }
  return n;
}

extern "C" void simpb_timing_reset(void) { g_used = 0; }

// Text of the last HIP error seen by simpb_check_launch() on this thread ("" if none).
static thread_local const char* g_last_error = "";
extern "C" const char* simpb_last_error(void) { return g_last_error; }
extern "C" int simpb_check_launch(void) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return SIMPB_OK;
  g_last_error = hipGetErrorString(e);
  return SIMPB_ELAUNCH;
}
