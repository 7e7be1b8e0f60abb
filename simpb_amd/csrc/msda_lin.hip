// Camera-grouped multi-scale deformable attention WITHOUT value_proj over the camera tokens (gfx950).
//
// QueryGroupMultiScaleDeformableAttention.forward (/root/reference/projects/mmdet3d_plugin/models/group_attn.py:176-243)
// projects every camera token first (value_proj: 89 760 x 256 x 256 per layer, the largest product of the decoder, three
// layers per frame = 35 GFLOP, 276 MB written and read back) and then samples 32-channel head slices of the result. Both
// steps are linear, so they commute: for head h of a query,
//     sum_s a_s * bilinear(W_h x + b_h)(s)  =  W_h . [ sum_s a_s * bilinear(x)(s) ]  +  b_h * [ sum_s a_s * (valid tap weights of s) ]
// (taps outside the map contribute zero to the left side -- the reference pads the PROJECTED map with zeros -- hence the
// second bracket instead of a plain b_h). This kernel computes the two brackets: it samples the raw 256-channel token
// rows per (query, head) and writes agg[q] = [8 heads x 256 | 8 tap-weight sums | pad] (2 176 floats); the projection
// with W_h -- folded with output_proj into ONE [256 x 2 176] matrix on the host (plugin/dense.py: fold_msda_linear) --
// is the small query-sized GEMM that follows (1.2 GFLOP instead of 11.8 + 0.15). Softmax of the attention logits,
// reference point + offset / (W_l, H_l) (group_attn.py:181-201; csrc/rowops.hip msda_prep) ride in the prologue.
//
// Mapping: one workgroup of 4 waves per (batch, query slot); a HALF-wave owns one head (wave w: heads 2w, 2w + 1), a
// lane 8 consecutive channels, so a tap is one 16-byte load per lane and a half-wave fetches one whole 512-byte f16 token
// row (two rows per wave-instruction). Per level all 4 points x 4 taps are requested before any is consumed (16 loads in
// flight per lane); the 8 x 256 sums live in registers, no cross-wave reduction. TOK = float serves callers whose tokens
// are not f16 numbers (1 KiB rows, two 16-byte loads per tap).
// Sampling rule = mmcv's CUDA op = grid_sample(bilinear, zeros, align_corners=False), as csrc/msda.hip.
#include <hip/hip_runtime.h>
#include "../../include/simpb_hip.h"
#include "store_fence.h"

extern "C" int simpb_check_launch(void);
extern "C" int simpb_timing_begin(int kernel_id, void* stream);
extern "C" void simpb_timing_end(int slot, void* stream);

namespace {

constexpr int kHeads = 8, kL = 4, kP = 4, kC = 256, kLP = kL * kP;
constexpr int kThreads = 256;

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

// a tap's 8 channels as loaded (converted when consumed, so that 16 taps in flight cost 64 registers, not 128)
template <class TOK> struct Raw8;
template <> struct Raw8<_Float16> {
  h8 x;
  __device__ __forceinline__ void load(const _Float16* p) { x = *reinterpret_cast<const h8*>(p); }
  __device__ __forceinline__ float at(int i) const { return (float)x[i]; }
};
template <> struct Raw8<float> {
  f4 a, b;
  __device__ __forceinline__ void load(const float* p) { a = *reinterpret_cast<const f4*>(p); b = *reinterpret_cast<const f4*>(p + 4); }
  __device__ __forceinline__ float at(int i) const { return i < 4 ? a[i] : b[i - 4]; }
};

template <class TOK>
__global__ __launch_bounds__(kThreads) void msda_linear_fwd(
    float* __restrict__ agg, int ld_agg, const TOK* __restrict__ tokens, const long long* __restrict__ spatial_shapes,
    const long long* __restrict__ level_start, const float* __restrict__ raw, int ldraw, const float* __restrict__ ref,
    int ldref, const int* __restrict__ query_cam, int num_cams, int num_value, int nq, const int* __restrict__ m_live) {
  __shared__ float s_x[kHeads * kLP], s_y[kHeads * kLP], s_a[kHeads * kLP];
  const int q = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int live = m_live ? min(nq, *m_live) : nq;
  int cam = query_cam[q];
  if (cam < 0 || q >= live) return;   // capacity slot: its row stays as the caller left it (plugin/ops.py: zeros unless the product behind skips it)
  cam = min(cam, num_cams - 1);
  const size_t qrow = (size_t)b * nq + q;

  // ---- prologue (msda_prep's arithmetic): thread t < 128 -> (head t / 16, level (t % 16) / 4, point t % 4)
  if (tid < kHeads * kLP) {
    const float* r = raw + qrow * ldraw;
    const float2 o = *reinterpret_cast<const float2*>(r + 2 * tid);
    const float lg = r[2 * kHeads * kLP + tid];
    const float rx = ref[qrow * ldref], ry = ref[qrow * ldref + 1];
    const int l = (tid % kLP) / kP;
    const float hl = (float)spatial_shapes[2 * l], wl = (float)spatial_shapes[2 * l + 1];
    float mx = lg;
#pragma unroll
    for (int m = kLP / 2; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
    const float ex = expf(lg - mx);
    float sum = ex;
#pragma unroll
    for (int m = kLP / 2; m >= 1; m >>= 1) sum += __shfl_xor(sum, m);
    s_x[tid] = rx + o.x / wl;
    s_y[tid] = ry + o.y / hl;
    s_a[tid] = ex / sum;
  }
  __syncthreads();

  const int head = 2 * wave + (lane >> 5);
  const int coff = (lane & 31) * 8;
  const TOK* tcam = tokens + ((size_t)b * num_cams + cam) * num_value * kC;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  float wsum = 0.f;
  constexpr int PB = sizeof(TOK) == 2 ? kP : kP / 2;   // points per batch of loads: 16 (f16) / 8 (f32) taps in flight
#pragma unroll 1
  for (int lb = 0; lb < kL * (kP / PB); ++lb) {
    const int l = lb / (kP / PB), p0 = (lb % (kP / PB)) * PB;
    const int H = (int)spatial_shapes[2 * l], W = (int)spatial_shapes[2 * l + 1];
    const TOK* base = tcam + (size_t)level_start[l] * kC + coff;
    Raw8<TOK> v[PB][4];
    float tw[PB][4];
#pragma unroll
    for (int p = 0; p < PB; ++p) {
      const int t = head * kLP + l * kP + p0 + p;
      const float lx = s_x[t], ly = s_y[t], aw = s_a[t];
      const float h_im = ly * (float)H - 0.5f;
      const float w_im = lx * (float)W - 0.5f;
      const float hf = floorf(h_im), wf = floorf(w_im);
      const int h0 = (int)fminf(fmaxf(hf, -2.f), (float)H);   // far-away locations: keep the conversion defined
      const int w0 = (int)fminf(fmaxf(wf, -2.f), (float)W);
      const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
      const bool y0 = h0 >= 0 && h0 <= H - 1, y1 = h0 + 1 >= 0 && h0 + 1 <= H - 1;
      const bool x0 = w0 >= 0 && w0 <= W - 1, x1 = w0 + 1 >= 0 && w0 + 1 <= W - 1;
      const int yc0 = min(max(h0, 0), H - 1), yc1 = min(max(h0 + 1, 0), H - 1);
      const int xc0 = min(max(w0, 0), W - 1), xc1 = min(max(w0 + 1, 0), W - 1);
      tw[p][0] = (y0 && x0) ? aw * hh * hw : 0.f;
      tw[p][1] = (y0 && x1) ? aw * hh * lw : 0.f;
      tw[p][2] = (y1 && x0) ? aw * lh * hw : 0.f;
      tw[p][3] = (y1 && x1) ? aw * lh * lw : 0.f;
      v[p][0].load(base + (size_t)(yc0 * W + xc0) * kC);
      v[p][1].load(base + (size_t)(yc0 * W + xc1) * kC);
      v[p][2].load(base + (size_t)(yc1 * W + xc0) * kC);
      v[p][3].load(base + (size_t)(yc1 * W + xc1) * kC);
    }
#pragma unroll
    for (int p = 0; p < PB; ++p)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        wsum += tw[p][k];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += tw[p][k] * v[p][k].at(i);
      }
  }
  float* o = agg + qrow * ld_agg;
  f4 lo = {acc[0], acc[1], acc[2], acc[3]}, hi = {acc[4], acc[5], acc[6], acc[7]};
  *reinterpret_cast<f4*>(o + head * kC + coff) = lo;
  *reinterpret_cast<f4*>(o + head * kC + coff + 4) = hi;
  if ((lane & 31) == 0) o[kHeads * kC + head] = wsum;
  if (tid >= kHeads && tid < 128) o[kHeads * kC + tid] = 0.f;   // pad columns of the 128-wide tail block (the product's chunk)
}

}  // namespace

extern "C" int simpb_msda_linear_forward(float* agg, int ld_agg, const void* tokens, int tokens_are_f16,
                                         const long long* spatial_shapes, const long long* level_start, const float* raw,
                                         int ld_raw, const float* ref, int ld_ref, const int* query_cam, const int* m_live,
                                         int batch_size, int num_cams, int num_value, int num_heads, int channels,
                                         int num_levels, int num_points, int num_query, void* stream) {
  if (!agg || !tokens || !spatial_shapes || !level_start || !raw || !ref || !query_cam) return SIMPB_EINVAL;
  if (batch_size <= 0 || batch_size > 65535 || num_cams <= 0 || num_value <= 0 || num_query <= 0) return SIMPB_EINVAL;
  // compiled for the shipped layout (config :163-190: 8 heads x 32 channels, 4 levels, 4 points)
  if (num_heads != kHeads || channels * num_heads != kC || num_levels != kL || num_points != kP) return SIMPB_EINVAL;
  if (ld_agg < kHeads * kC + 128 || (ld_agg & 3) || (reinterpret_cast<size_t>(agg) & 15) || ld_raw < 3 * kHeads * kLP ||
      (ld_raw & 1) || (reinterpret_cast<size_t>(raw) & 7) || ld_ref < 2 || (reinterpret_cast<size_t>(tokens) & 15))
    return SIMPB_EINVAL;
  (void)hipGetLastError();
  hipStream_t s = static_cast<hipStream_t>(stream);
  dim3 grid(num_query, batch_size), block(kThreads);
  const int tslot = simpb_timing_begin(SIMPB_KERNEL_MSDA, stream);
  if (tokens_are_f16)
    hipLaunchKernelGGL(msda_linear_fwd<_Float16>, grid, block, 0, s, agg, ld_agg, static_cast<const _Float16*>(tokens),
                       spatial_shapes, level_start, raw, ld_raw, ref, ld_ref, query_cam, num_cams, num_value, num_query, m_live);
  else
    hipLaunchKernelGGL(msda_linear_fwd<float>, grid, block, 0, s, agg, ld_agg, static_cast<const float*>(tokens),
                       spatial_shapes, level_start, raw, ld_raw, ref, ld_ref, query_cam, num_cams, num_value, num_query, m_live);
  simpb_timing_end(tslot, stream);
  return simpb_check_launch();
}
