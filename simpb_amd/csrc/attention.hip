// Multi-head attention core, exact fp32 on the f32 matrix cores (gfx950 v_mfma_f32_32x32x2_f32),
// flash style (no score matrix in memory), head_dim 64.
//
//   out[b, q, h*64 + d] = sum_k softmax_k(scale * Q[b,q,h,:] . K[b,k,h,:]) V[b,k,h,d]
//
// This is the arithmetic torch.nn.MultiheadAttention performs between its in- and out-projections,
// which is what every attention operator of the reference's decoder bottoms out in
// (/root/reference/projects/mmdet3d_plugin/models/simpb_head.py:298-321 -> mmcv MultiheadAttention,
// models/group_attn.py:25-133, models/aggregation.py:96-99). Optionally camera-grouped: query slot q
// attends only key slots of its own camera group (query_cam / group_start device tables). That is
// the reference's N2 x N2 additive mask with -inf across groups (group_attn.py:104-113) without the
// mask tensor and without the cross-group score work; slots with query_cam < 0 (capacity padding)
// produce zeros, which is what nan_to_num gives the reference's fully-masked rows (:131).
//
// Mapping: one workgroup = 32 queries of one (batch, head); its 4 waves split the key tiles (32 keys
// each) and merge their (max, sum, O) partials through LDS at the end. Orientation follows the
// "key on the register, query on the lane" form: S^T = K.Q^T puts each query's scores of a tile in
// ONE lane pair, so the row max/sum are in-register plus one xor-32 shuffle, and the accumulator
// tile is directly the B operand of O^T += V^T.P^T (k-order of the second product follows the
// accumulator's row map), leaving O^T with the query on the lane again for the rescale.
#include <hip/hip_runtime.h>
#include "../../include/simpb_hip.h"

extern "C" int simpb_check_launch(void);

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kHD = 64;
constexpr int kWaves = 4;  // (8 waves per workgroup, 2 per SIMD, measured no faster: 406 vs 391 us per frame)

__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

template <bool GROUPED>
__global__ __launch_bounds__(kWaves * 64) void attention_f32_kernel(
    float* __restrict__ out, const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const int* __restrict__ query_cam, const int* __restrict__ group_start, int Nq, int Nk, int ldq, int ldk, int ldv,
    int ldo, float scale) {
  __shared__ float s_m[kWaves][64];
  __shared__ float s_l[kWaves][64];
  __shared__ float s_o[kWaves][2][16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qi = lane & 31, half = lane >> 5;
  const int q0 = blockIdx.x * 32, head = blockIdx.y, b = blockIdx.z;
  const int qg = q0 + qi;
  const bool q_ok = qg < Nq;

  int cam_q = 0, kb = 0, ke = Nk;
  int my_lo = 0, my_hi = Nk;   // this lane's query attends keys [my_lo, my_hi): its own camera group (groups are contiguous slot ranges)
  if (GROUPED) {
    cam_q = q_ok ? query_cam[qg] : -1;
    int lo = cam_q >= 0 ? group_start[cam_q] : 0x7fffffff;
    int hi = cam_q >= 0 ? group_start[cam_q + 1] : 0;
    my_lo = lo;
    my_hi = hi;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      lo = min(lo, __shfl_xor(lo, m));
      hi = max(hi, __shfl_xor(hi, m));
    }
    // a tile made only of capacity slots (cam -1) has no group at all: empty key range. (lo stays
    // INT_MAX there; adding the wave offset to it would wrap around.)
    const bool any = hi > lo;
    kb = any ? lo : 0;
    ke = any ? min(hi, Nk) : 0;
  }

  // Q^T operand: lane (query qi, half) holds Q[query][32*half + s], s = 0..31, pre-scaled
  float qreg[32];
  {
    const float* qp = q + ((size_t)b * Nq + (q_ok ? qg : 0)) * ldq + head * kHD + 32 * half;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float4 t = q_ok ? *reinterpret_cast<const float4*>(qp + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
      qreg[4 * j + 0] = t.x * scale; qreg[4 * j + 1] = t.y * scale;
      qreg[4 * j + 2] = t.z * scale; qreg[4 * j + 3] = t.w * scale;
    }
  }

  float m_run = -INFINITY, l_run = 0.f;
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }

  const float* kbase = k + (size_t)b * Nk * ldk + head * kHD;
  const float* vbase = v + (size_t)b * Nk * ldv + head * kHD;

  // K tile of this wave's first key tile; inside the loop the NEXT tile's K rows and the CURRENT tile's
  // V values are requested before the matrix work that does not need them, so their latency hides
  // behind the S^T product and the softmax (nothing else overlaps it at 2 waves per SIMD).
  auto load_k = [&](int kt, float (&kreg)[32]) {
    // rows past the range read the first key of the range instead (finite values): their scores are
    // masked to -inf below, so no select touches the loaded registers before the product needs them
    const int kk = kt + qi;
    const float* kp = kbase + (size_t)(kk < ke ? kk : kb) * ldk + 32 * half;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float4 t = *reinterpret_cast<const float4*>(kp + 4 * j);
      kreg[4 * j + 0] = t.x; kreg[4 * j + 1] = t.y; kreg[4 * j + 2] = t.z; kreg[4 * j + 3] = t.w;
    }
  };
  // V values of a tile: lane (d, half) holds V[key kt + acc_row(s, half)][d], s = 0..15
  auto load_v = [&](int kt, float (&a0)[16], float (&a1)[16]) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int kidx = kt + acc_row(s, half);
      const float* vp = vbase + (size_t)(kidx < ke ? kidx : kb) * ldv + qi;  // past the range: weight 0 below
      a0[s] = vp[0];
      a1[s] = vp[32];
    }
  };
  float kcur[32], knext[32];
  const int kt0 = kb + 32 * wave;
  if (kt0 < ke) load_k(kt0, kcur);
  for (int kt = kt0; kt < ke; kt += 32 * kWaves) {
    float va0[16], va1[16];
    load_v(kt, va0, va1);
    // requested UNCONDITIONALLY (past the last tile load_k clamps every row to the first key of the
    // range and the values are never used): a load behind `if (ktn < ke)` made the compiler wait for
    // every outstanding load -- these and the V values above -- in front of the S^T product.
    // (V of the NEXT tile requested a whole tile ahead was measured slower in both forms -- with K at the top: 35 vs 30 us
    // at 900 x 900, 80 loads in flight against a counter of 63; behind the S^T product with two alternating register sets:
    // 31.3 us, and 31-37 vs 25-31 us for the grouped form at 314 VGPRs. What is exposed per tile is not the V round trip but
    // the ~300 vector instructions of the softmax, which one wave per SIMD cannot overlap with its own matrix work.)
    const int ktn = kt + 32 * kWaves;
    load_k(ktn, knext);
    __builtin_amdgcn_sched_barrier(0);  // the requests stay here, ahead of the matrix work (the scheduler sank them)
    // ---- S^T tile = K_tile . Q^T  (A: lane (key, half) holds K[key][32*half + s])
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 32; ++s) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kcur[s], qreg[s], st, 0, 0, 0);
    // ---- mask + online softmax; this lane: query qi, keys kt + acc_row(r, half)
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kidx = kt + acc_row(r, half);
      bool ok = kidx < ke && q_ok;
      if (GROUPED) ok = ok && kidx >= my_lo && kidx < my_hi;   // (= query_cam[kidx] == cam_q, without 16 loads per tile in front of the softmax)
      st[r] = ok ? st[r] : -INFINITY;
      mt = fmaxf(mt, st[r]);
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32));
    const float m_new = fmaxf(m_run, mt);
    const float m_safe = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = expf(m_run - m_safe);  // m_run = -inf -> 0
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      st[r] = expf(st[r] - m_safe);
      psum += st[r];
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
    // ---- O^T += V^T . P^T  (A: va0/va1; B: st[s])
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va0[s], st[s], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(va1[s], st[s], o1, 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < 32; ++s) kcur[s] = knext[s];
  }

  // ---- merge the 4 waves
  s_m[wave][lane] = m_run;
  s_l[wave][lane] = l_run;
#pragma unroll
  for (int r = 0; r < 16; ++r) { s_o[wave][0][r][lane] = o0[r]; s_o[wave][1][r][lane] = o1[r]; }
  __syncthreads();
  float mx = -INFINITY;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) mx = fmaxf(mx, s_m[w][lane]);
  float f[kWaves];
  float lsum = 0.f;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) {
    f[w] = mx == -INFINITY ? 0.f : expf(s_m[w][lane] - mx);
    lsum += f[w] * (s_l[w][lane] + s_l[w][lane ^ 32]);
  }
  const float inv = lsum > 0.f ? 1.f / lsum : 0.f;  // no admissible key: zeros (group_attn.py:131)
  // wave w finishes d-tile w / (kWaves/2), a share of 32/kWaves accumulator registers
  constexpr int kShare = 32 / kWaves;
  const int t = wave / (kWaves / 2), r0 = kShare * (wave % (kWaves / 2));
  if (q_ok) {
    float* op = out + ((size_t)b * Nq + qg) * ldo + head * kHD + 32 * t;
#pragma unroll
    for (int r = r0; r < r0 + kShare; ++r) {
      float acc = 0.f;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) acc += f[w] * s_o[w][t][r][lane];
      op[acc_row(r, half)] = acc * inv;
    }
  }
}

}  // namespace

extern "C" int simpb_attention_f32(float* out, const float* q, const float* k, const float* v, const int* query_cam,
                                   const int* group_start, int batch_size, int num_heads, int head_dim, int num_query,
                                   int num_key, int ldq, int ldk, int ldv, int ldo, float scale, void* stream) {
  if (!out || !q || !k || !v || batch_size <= 0 || num_heads <= 0 || num_query <= 0 || num_key <= 0) return SIMPB_EINVAL;
  if (head_dim != kHD || ldq < num_heads * kHD || ldk < num_heads * kHD || ldv < num_heads * kHD || ldo < num_heads * kHD)
    return SIMPB_EINVAL;
  if ((ldq | ldk) & 3) return SIMPB_EINVAL;
  if ((reinterpret_cast<size_t>(q) | reinterpret_cast<size_t>(k)) & 15) return SIMPB_EINVAL;
  if ((query_cam == nullptr) != (group_start == nullptr)) return SIMPB_EINVAL;
  if (query_cam && num_key != num_query) return SIMPB_EINVAL;  // grouped form is self-attention over one slot set
  if (batch_size > 65535 || num_heads > 65535) return SIMPB_EINVAL;
  (void)hipGetLastError();
  dim3 grid((num_query + 31) / 32, num_heads, batch_size), block(kWaves * 64);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (query_cam)
    hipLaunchKernelGGL(attention_f32_kernel<true>, grid, block, 0, s, out, q, k, v, query_cam, group_start, num_query,
                       num_key, ldq, ldk, ldv, ldo, scale);
  else
    hipLaunchKernelGGL(attention_f32_kernel<false>, grid, block, 0, s, out, q, k, v, query_cam, group_start, num_query,
                       num_key, ldq, ldk, ldv, ldo, scale);
  return simpb_check_launch();
}
