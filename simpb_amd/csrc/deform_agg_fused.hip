// DeformableFeatureAggregation.forward between its Linear layers, ONE launch (gfx950): key points
// (/root/reference/projects/mmdet3d_plugin/models/detection3d/blocks.py:181-222) -> project_points
// (models/blocks.py:198-213) -> the softmax of _get_weights (blocks.py:177-187) -> the aggregation kernel
// (ops/src/deformable_aggregation_cuda.cu:129-187). What csrc/dfa_prep.hip + csrc/deform_agg.hip do in three launches
// through two HBM tensors (loc 0.56 MB, weights 9 MB written and read back per layer), here in one: the workgroup that
// aggregates anchor `a` computes that anchor's 13 x 6 sampling locations in registers and its 312 x 8 softmax weights
// in LDS first. Same arithmetic, statement for statement, as the three kernels (tests compare them).
//
// Mapping as daf_fwd_rows: one workgroup of 4 waves per (batch, anchor), wave w = level w, lane = 4 channels, a tap = one
// coalesced row per wave-instruction, sums in registers, one LDS meeting, one store. FEAT = float reads the decoder's
// fp32 token rows (1 KiB per tap); FEAT = _Float16 reads the f16 copy the FPN's output convolutions leave beside them
// (512 B per tap): the tokens ARE f16 numbers (the backbone runs in fp16, simpb.py:63), widening is exact, so both give
// the same bits at half the gather bytes.
#include <hip/hip_runtime.h>
#include "../../include/simpb_hip.h"
#include "store_fence.h"

extern "C" int simpb_check_launch(void);
extern "C" int simpb_timing_begin(int kernel_id, void* stream);
extern "C" void simpb_timing_end(int slot, void* stream);

namespace {

constexpr int kWaves = 4;
constexpr int kThreads = kWaves * 64;
constexpr int kMaxW = 4096;   // floats of LDS for the softmax weights: cams * L * P * G

struct Tap4 {
  float4 v00, v01, v10, v11;
  float w00, w01, w10, w11;
};

__device__ __forceinline__ float4 ld_row(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld_row(const _Float16* p) {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  const h4 v = *reinterpret_cast<const h4*>(p);
  return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}

template <class FEAT>
__device__ __forceinline__ void issue_taps(Tap4& t, const FEAT* __restrict__ base, int H, int W, int C, float lx, float ly,
                                           int coff) {
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
  t.v00 = ld_row(base + (size_t)(yc0 * W + xc0) * C + coff);
  t.v01 = ld_row(base + (size_t)(yc0 * W + xc1) * C + coff);
  t.v10 = ld_row(base + (size_t)(yc1 * W + xc0) * C + coff);
  t.v11 = ld_row(base + (size_t)(yc1 * W + xc1) * C + coff);
}

__device__ __forceinline__ void accumulate(float4& acc, const Tap4& t, float wgt) {
  acc.x += wgt * (t.w00 * t.v00.x + t.w01 * t.v01.x + t.w10 * t.v10.x + t.w11 * t.v11.x);
  acc.y += wgt * (t.w00 * t.v00.y + t.w01 * t.v01.y + t.w10 * t.v10.y + t.w11 * t.v11.y);
  acc.z += wgt * (t.w00 * t.v00.z + t.w01 * t.v01.z + t.w10 * t.v10.z + t.w11 * t.v11.z);
  acc.w += wgt * (t.w00 * t.v00.w + t.w01 * t.v01.w + t.w10 * t.v10.w + t.w11 * t.v11.w);
}

struct DfaArgs {
  float* out;                 // [bs, A, C]
  const void* feat;           // FEAT [bs, num_feat, C]
  const int* spatial_shape;   // [cams, L, 2]
  const int* scale_start;     // [cams, L]
  const float* anchor;        // [bs, A, 11]
  const float* learn;         // [bs, A, num_learn * 3] raw learnable_fc output (sigmoid applied here)
  const float* fix_scale;     // [num_fix, 3]
  const float* proj;          // [bs, cams, 4, 4]
  const float* image_wh;      // [bs, cams, 2]
  const float* feat_logits;   // [bs, A, L * P * G]       weights_fc(feature + anchor_embed)
  const float* cam_logits;    // [bs, cams, L * P * G]    camera_embed . weights_fc.weight^T
  float* loc_out;             // optional [bs, A, P, cams, 2]      (tests / measurement)
  float* w_out;               // optional [bs, A, P, cams, L, G]   (tests)
  int cams, num_feat, C, L, A, num_fix, num_learn, G;
};

// operands of one sampling location (key point p in camera cam), fetched before anything is computed from them
struct PointOps {
  float f0, f1, f2;   // fix_scale row or raw learnable offsets
  float m[12];        // first three rows of the camera's projection matrix
  float w0, w1;       // image width / height
  bool fixed;
};

template <int CAMS, int NUM_FIX, int NUM_LEARN>
__device__ __forceinline__ void fetch_point(PointOps& o, const DfaArgs& k, size_t row, int b, int i) {
  const int p = i / CAMS, cam = i - p * CAMS;
  o.fixed = p < NUM_FIX;
  const float* f = o.fixed ? k.fix_scale + p * 3 : k.learn + (row * NUM_LEARN + (p - NUM_FIX)) * 3;
  o.f0 = f[0]; o.f1 = f[1]; o.f2 = f[2];
  const float* M = k.proj + ((size_t)b * CAMS + cam) * 16;
#pragma unroll
  for (int j = 0; j < 12; ++j) o.m[j] = M[j];
  const float* wh = k.image_wh + ((size_t)b * CAMS + cam) * 2;
  o.w0 = wh[0]; o.w1 = wh[1];
}

// dfa_points_kernel's arithmetic (csrc/dfa_prep.hip), statement for statement
__device__ __forceinline__ float2 project_point(const PointOps& o, const float* av) {
  const float sw = expf(av[3]), sl = expf(av[4]), sh = expf(av[5]);
  float kx, ky, kz;
  if (o.fixed) {
    kx = o.f0 * sw; ky = o.f1 * sl; kz = o.f2 * sh;
  } else {
    kx = (1.f / (1.f + expf(-o.f0)) - 0.5f) * sw;
    ky = (1.f / (1.f + expf(-o.f1)) - 0.5f) * sl;
    kz = (1.f / (1.f + expf(-o.f2)) - 0.5f) * sh;
  }
  const float sn = av[6], cs = av[7];
  const float px = cs * kx - sn * ky + av[0];
  const float py = sn * kx + cs * ky + av[1];
  const float pz = kz + av[2];
  const float u = o.m[0] * px + o.m[1] * py + o.m[2] * pz + o.m[3];
  const float v = o.m[4] * px + o.m[5] * py + o.m[6] * pz + o.m[7];
  const float d = fmaxf(o.m[8] * px + o.m[9] * py + o.m[10] * pz + o.m[11], 1e-5f);
  return make_float2(u / d / o.w0, v / d / o.w1);
}

// CAMS / L / NUM_FIX / NUM_LEARN / G / C are compile-time: the index arithmetic of the prologue (entry -> (cam, level, point))
// is then shifts and multiplies; with run-time divisors it was ~3 000 instructions per thread and as long as the gather.
template <class FEAT, int CAMS, int L, int NUM_FIX, int NUM_LEARN, int G, int C>
__global__ __launch_bounds__(kThreads) void daf_fused_rows(DfaArgs k) {
  __shared__ float s_w[kMaxW];            // [(p * cams + cam) * L + lvl][G]: exp(logit - max), NOT yet divided by the sum
  __shared__ float4 s_red[kWaves][64];    // softmax reductions first, the waves' partial rows at the end
  __shared__ float2 s_loc[128];           // sampling locations, i = p * cams + cam
  const int a = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int cams = CAMS, P = NUM_FIX + NUM_LEARN, PK = P * cams;
  constexpr int LP = L * P, LPG = LP * G, n = cams * LP, slices = kThreads / G;
  static_assert(PK <= 128 && n * G <= kMaxW && (G & (G - 1)) == 0 && G <= 64 && C <= 256 && (C / G) % 4 == 0,
                "layout");
  const size_t row = (size_t)b * k.A + a;
  const int g_sm = tid % G, slice = tid / G;

  // ---- every global operand of the prologue is requested up front (indices clamped, loads unconditional): the
  // prologue then costs ONE memory round trip, not one per dependent step (all workgroups of the launch are resident at
  // once and walk their phases in step, so prologue latency adds to the launch time in full)
  float av[8];
  {
    const float* an = k.anchor + row * 11;
#pragma unroll
    for (int j = 0; j < 8; ++j) av[j] = an[j];
  }
  PointOps op;   // thread i < PK computes sampling location i = p * cams + cam, once per workgroup
  fetch_point<CAMS, NUM_FIX, NUM_LEARN>(op, k, row, b, min(tid, PK - 1));
  // softmax entries of this thread: group g_sm, (level, point) pairs lp = slice + j * slices, every camera
  constexpr int NJ = (LP + slices - 1) / slices;
  float lg[NJ][CAMS];
  {
    const float* fl = k.feat_logits + row * LPG;
    const float* cl = k.cam_logits + (size_t)b * cams * LPG;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int lp = min(slice + j * slices, LP - 1);
      const float f = fl[lp * G + g_sm];
#pragma unroll
      for (int cam = 0; cam < CAMS; ++cam) lg[j][cam] = f + cl[cam * LPG + lp * G + g_sm];   // (blocks.py:177-179)
    }
  }

  // ---- sampling locations (dfa_points_kernel's arithmetic), published through LDS
  if (tid < PK) s_loc[tid] = project_point(op, av);
  else if (tid < 128) s_loc[tid] = make_float2(-1.f, -1.f);

  // ---- softmax over the cams * L * P entries of each group (dfa_weights_kernel's arithmetic): thread -> (group, slice);
  // lanes of one group sit G apart inside a wave, so a group's reduction is xor shuffles + one LDS meeting. The division
  // by the sum is applied once to the finished channel sums (every channel belongs to one group).
  float inv_sum;
  {
    float* red = reinterpret_cast<float*>(s_red);   // [2][kWaves][64]: max, then sum
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      if (slice + j * slices < LP) {
#pragma unroll
        for (int cam = 0; cam < CAMS; ++cam) m = fmaxf(m, lg[j][cam]);
      }
    for (int s = G; s < 64; s <<= 1) m = fmaxf(m, __shfl_xor(m, s));
    if (lane < G) red[wave * 64 + lane] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[g_sm], red[64 + g_sm]), fmaxf(red[128 + g_sm], red[192 + g_sm]));
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int lp = slice + j * slices;
      if (lp < LP) {
        const int lvl = lp / P, pt = lp - lvl * P;
        float* w = s_w + (pt * cams * L + lvl) * G + g_sm;   // [(pt * cams + cam) * L + lvl][G]
#pragma unroll
        for (int cam = 0; cam < CAMS; ++cam) {
          const float v = expf(lg[j][cam] - m);
          w[cam * L * G] = v;
          sum += v;
        }
      }
    }
    for (int s = G; s < 64; s <<= 1) sum += __shfl_xor(sum, s);
    if (lane < G) red[256 + wave * 64 + lane] = sum;
    __syncthreads();   // s_w and s_loc complete, sums in place
    if (k.w_out) {
      const float inv = 1.f / (red[256 + g_sm] + red[256 + 64 + g_sm] + red[256 + 128 + g_sm] + red[256 + 192 + g_sm]);
      float* wo = k.w_out + row * (size_t)n * G;
      for (int e = slice; e < n; e += slices) wo[e * G + g_sm] = s_w[e * G + g_sm] * inv;   // (same layout)
      simpb::stores_retired();
    }
    const int gch = (lane * 4 < C ? lane * 4 : 0) / (C / G);   // group of this lane's channels in the aggregation below
    inv_sum = 1.f / (red[256 + gch] + red[256 + 64 + gch] + red[256 + 128 + gch] + red[256 + 192 + gch]);
  }
  // lane i (and i + 64) of EVERY wave holds location i
  const float2 l0 = s_loc[lane], l1 = s_loc[lane + 64];
  if (k.loc_out && wave == 0) {
    float2* lo = reinterpret_cast<float2*>(k.loc_out) + row * PK;
    if (lane < PK) lo[lane] = l0;
    if (lane + 64 < PK) lo[lane + 64] = l1;
    simpb::stores_retired();   // (measurement-only output: nothing of it in flight beside the gather's counted waits)
  }
  const unsigned long long m0 = __ballot(l0.x > 0.f && l0.x < 1.f && l0.y > 0.f && l0.y < 1.f);
  const unsigned long long m1 = __ballot(l1.x > 0.f && l1.x < 1.f && l1.y > 0.f && l1.y < 1.f);

  // ---- the aggregation itself (daf_fwd_rows' loop with the weights in LDS)
  const int coff = lane * 4;
  const int ld_off = coff < C ? coff : 0;
  const int g = ld_off / (C / G);
  const FEAT* featb = static_cast<const FEAT*>(k.feat) + (size_t)b * k.num_feat * C;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr int kBatch = 4;   // samples (x 4 taps) requested before any is consumed: a trip of the loop is one memory round trip
  for (int lvl = wave; lvl < L; lvl += kWaves) {
    unsigned long long ma = m0, mb = m1;
    while (ma | mb) {
      int idx[kBatch];
#pragma unroll
      for (int j = 0; j < kBatch; ++j) {
        idx[j] = -1;
        if (ma) { idx[j] = __builtin_ctzll(ma); ma &= ma - 1; }
        else if (mb) { idx[j] = 64 + __builtin_ctzll(mb); mb &= mb - 1; }
      }
      Tap4 t[kBatch];
      float wg[kBatch];
#pragma unroll
      for (int j = 0; j < kBatch; ++j) {
        wg[j] = 0.f;
        if (idx[j] >= 0) {   // wave-uniform
          const int i = idx[j];
          const float x = i < 64 ? __shfl(l0.x, i) : __shfl(l1.x, i - 64);
          const float y = i < 64 ? __shfl(l0.y, i) : __shfl(l1.y, i - 64);
          const int cs = (i % cams) * L + lvl;
          issue_taps(t[j], featb + (size_t)k.scale_start[cs] * C, k.spatial_shape[2 * cs], k.spatial_shape[2 * cs + 1], C, x, y,
                     ld_off);
          wg[j] = s_w[(i * L + lvl) * G + g];
        }
      }
#pragma unroll
      for (int j = 0; j < kBatch; ++j)
        if (idx[j] >= 0) accumulate(acc, t[j], wg[j]);
    }
  }
  __syncthreads();   // every wave has read the sums out of s_red
  acc.x *= inv_sum; acc.y *= inv_sum; acc.z *= inv_sum; acc.w *= inv_sum;
  s_red[wave][lane] = acc;
  __syncthreads();
  if (tid < C) {
    const float* r = reinterpret_cast<const float*>(s_red);
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) s += r[w * 256 + tid];
    k.out[row * C + tid] = s;
  }
}

}  // namespace

extern "C" int simpb_dfa_fused_forward(
    float* output, const void* mc_ms_feat, int feat_is_f16, const int* spatial_shape, const int* scale_start_index,
    const float* anchor, const float* learnable, const float* fix_scale, const float* projection_mat, const float* image_wh,
    const float* feat_logits, const float* cam_logits, float* loc_out, float* weights_out, int batch_size, int num_cams,
    int num_feat, int num_embeds, int num_scale, int num_anchors, int num_fix, int num_learn, int num_groups, void* stream) {
  if (!output || !mc_ms_feat || !spatial_shape || !scale_start_index || !anchor || !fix_scale || !projection_mat ||
      !image_wh || !feat_logits || !cam_logits || (num_learn > 0 && !learnable))
    return SIMPB_EINVAL;
  if (batch_size <= 0 || batch_size > 65535 || num_cams <= 0 || num_feat <= 0 || num_scale <= 0 || num_anchors <= 0 ||
      num_fix < 0 || num_learn < 0 || num_fix + num_learn <= 0 || num_groups <= 0)
    return SIMPB_EINVAL;
  // compiled for the shipped layout only; the three-launch route (simpb_dfa_points / simpb_dfa_weights /
  // simpb_deformable_aggregation_forward) takes any other
  if (num_cams != 6 || num_scale != 4 || num_fix != 7 || num_learn != 6 || num_groups != 8 || num_embeds != 256)
    return SIMPB_EINVAL;
  (void)hipGetLastError();
  DfaArgs k;
  k.out = output; k.feat = mc_ms_feat; k.spatial_shape = spatial_shape; k.scale_start = scale_start_index;
  k.anchor = anchor; k.learn = learnable; k.fix_scale = fix_scale; k.proj = projection_mat; k.image_wh = image_wh;
  k.feat_logits = feat_logits; k.cam_logits = cam_logits; k.loc_out = loc_out; k.w_out = weights_out;
  k.cams = num_cams; k.num_feat = num_feat; k.C = num_embeds; k.L = num_scale; k.A = num_anchors; k.num_fix = num_fix;
  k.num_learn = num_learn; k.G = num_groups;
  hipStream_t s = static_cast<hipStream_t>(stream);
  dim3 grid(num_anchors, batch_size);
  const int tslot = simpb_timing_begin(SIMPB_KERNEL_DAF, stream);
  // the shipped layout (config :221-238: 6 cameras, 4 levels, 7 fixed + 6 learnable key points, 8 groups of 32 channels)
  if (feat_is_f16)
    hipLaunchKernelGGL((daf_fused_rows<_Float16, 6, 4, 7, 6, 8, 256>), grid, dim3(kThreads), 0, s, k);
  else
    hipLaunchKernelGGL((daf_fused_rows<float, 6, 4, 7, 6, 8, 256>), grid, dim3(kThreads), 0, s, k);
  simpb_timing_end(tslot, stream);
  return simpb_check_launch();
}
