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


// ---------------------------------------------------------------------------------------------------------------------
// The same attention on the FP16 matrix cores with SPLIT operands (round 4), fp32-grade like csrc/gemm.hip's
// gemm_f16x3_kernel: every fp32 operand x is carried as xh + xl / 2^11 (two halfs, 22 bits), products of halfs are exact
// in the fp32 accumulators, and the partial products are summed in separate accumulators per scale:
//     S = Kh.Qh + (Kh.Ql + Kl.Qh) / 2^11 + Kl.Ql / 2^22           (all four terms: the softmax sits behind it)
//     O = Vh.Ph + (Vh.Pl + Vl.Ph) / 2^11                           (the dropped term is 2^-22 of sum p |v|)
// Why: the exact-fp32 kernel above is matrix-bound per SIMD at one stream -- 900 x 900 x 8 heads is 29 x 8 workgroups,
// one wave per SIMD, 64 v_mfma_f32_32x32x2_f32 of 64 cycles each per key tile = 4 096 cycles of the ~5 300 a tile
// takes. v_mfma_f32_32x32x8_f16 moves 4x the k-depth in half the cycles: 56 instructions of 32 cycles = 1 792.
// Same mapping, same operand registers: a lane's 32 contiguous head dims pair up as k = 4c + i of instruction c on BOTH
// operands (a dot product does not care in which order its terms meet), and the accumulator row map of S^T is again
// the B-operand k map of the second product (rows 8c + 4g + i of lane half g = k index 4g + i of instruction c).
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split4(const float* x, h16x4& hi, h16x4& lo) {
  hi = h16x4{(_Float16)x[0], (_Float16)x[1], (_Float16)x[2], (_Float16)x[3]};
  lo = h16x4{(_Float16)((x[0] - (float)hi[0]) * 2048.f), (_Float16)((x[1] - (float)hi[1]) * 2048.f),
             (_Float16)((x[2] - (float)hi[2]) * 2048.f), (_Float16)((x[3] - (float)hi[3]) * 2048.f)};
}

// PACKED operands: the producer (csrc/gemm.hip, out_fmt = SIMPB_GEMM_OUT_SPLIT_HALFS) already left every element as its two
// halfs in the element's own 32-bit word (hi in the low 16 bits, lo * 2^11 in the high 16 bits): same strides, same loads,
// and four consecutive words become one (hi, lo) operand pair with four byte permutes -- instead of ~5 conversions per
// value repeated by each of the 29 query-tile workgroups that read the same keys.
__device__ __forceinline__ void unpack4(const float* w, h16x4& hi, h16x4& lo) {
  const unsigned w0 = __float_as_uint(w[0]), w1 = __float_as_uint(w[1]), w2 = __float_as_uint(w[2]), w3 = __float_as_uint(w[3]);
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 h = {__builtin_amdgcn_perm(w1, w0, 0x05040100u), __builtin_amdgcn_perm(w3, w2, 0x05040100u)};
  const u32x2 l = {__builtin_amdgcn_perm(w1, w0, 0x07060302u), __builtin_amdgcn_perm(w3, w2, 0x07060302u)};
  hi = __builtin_bit_cast(h16x4, h);
  lo = __builtin_bit_cast(h16x4, l);
}

template <bool GROUPED, bool PACKED = false>
__global__ __launch_bounds__(kWaves * 64) void attention_f16s_kernel(
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
  int my_lo = 0, my_hi = Nk;
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
    const bool any = hi > lo;
    kb = any ? lo : 0;
    ke = any ? min(hi, Nk) : 0;
  }

  // Q^T operand, split once: lane (query qi, half) holds Q[query][32*half + 4c + i] * scale as (hi, lo) of instruction c
  h16x4 qh[8], ql[8];
  {
    const float* qp = q + ((size_t)b * Nq + (q_ok ? qg : 0)) * ldq + head * kHD + 32 * half;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float4 t = q_ok ? *reinterpret_cast<const float4*>(qp + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (PACKED) {   // (the producer folded the softmax scale into the query rows of its weights: a power of two, exact)
        const float x[4] = {t.x, t.y, t.z, t.w};
        unpack4(x, qh[c], ql[c]);
      } else {
        const float x[4] = {t.x * scale, t.y * scale, t.z * scale, t.w * scale};
        split4(x, qh[c], ql[c]);
      }
    }
  }

  float m_run = -INFINITY, l_run = 0.f;
  f32x16 o0, o1, x0, x1;   // O^T d-tiles: leading terms / cross terms (scaled by 2^11)
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; x0[r] = 0.f; x1[r] = 0.f; }

  const float* kbase = k + (size_t)b * Nk * ldk + head * kHD;
  const float* vbase = v + (size_t)b * Nk * ldv + head * kHD;
  auto load_k = [&](int kt, float (&kreg)[32]) {
    const int kk = kt + qi;
    const float* kp = kbase + (size_t)(kk < ke ? kk : kb) * ldk + 32 * half;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float4 t = *reinterpret_cast<const float4*>(kp + 4 * j);
      kreg[4 * j + 0] = t.x; kreg[4 * j + 1] = t.y; kreg[4 * j + 2] = t.z; kreg[4 * j + 3] = t.w;
    }
  };
  auto load_v = [&](int kt, float (&a0)[16], float (&a1)[16]) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int kidx = kt + acc_row(s, half);
      const float* vp = vbase + (size_t)(kidx < ke ? kidx : kb) * ldv + qi;
      a0[s] = vp[0];
      a1[s] = vp[32];
    }
  };
  constexpr float kInv = 1.f / 2048.f;
  float kcur[32], knext[32];
  const int kt0 = kb + 32 * wave;
  if (kt0 < ke) load_k(kt0, kcur);
  for (int kt = kt0; kt < ke; kt += 32 * kWaves) {
    float va0[16], va1[16];
    load_v(kt, va0, va1);
    const int ktn = kt + 32 * kWaves;
    load_k(ktn, knext);   // unconditional (clamped): see the exact kernel
    __builtin_amdgcn_sched_barrier(0);
    // ---- S^T tile = K_tile . Q^T in four split terms
    f32x16 shh, sx, sll;
#pragma unroll
    for (int r = 0; r < 16; ++r) { shh[r] = 0.f; sx[r] = 0.f; sll[r] = 0.f; }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      h16x4 kh, kl;
      if (PACKED) unpack4(&kcur[4 * c], kh, kl); else split4(&kcur[4 * c], kh, kl);
      shh = __builtin_amdgcn_mfma_f32_32x32x8f16(kh, qh[c], shh, 0, 0, 0);
      sx = __builtin_amdgcn_mfma_f32_32x32x8f16(kh, ql[c], sx, 0, 0, 0);
      sx = __builtin_amdgcn_mfma_f32_32x32x8f16(kl, qh[c], sx, 0, 0, 0);
      sll = __builtin_amdgcn_mfma_f32_32x32x8f16(kl, ql[c], sll, 0, 0, 0);
    }
    // ---- mask + online softmax (fp32 vector arithmetic, as the exact kernel)
    float st[16];
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kidx = kt + acc_row(r, half);
      bool ok = kidx < ke && q_ok;
      if (GROUPED) ok = ok && kidx >= my_lo && kidx < my_hi;
      const float sv = shh[r] + (sx[r] + sll[r] * kInv) * kInv;
      st[r] = ok ? sv : -INFINITY;
      mt = fmaxf(mt, st[r]);
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32));
    const float m_new = fmaxf(m_run, mt);
    const float m_safe = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = expf(m_run - m_safe);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      st[r] = expf(st[r] - m_safe);
      psum += st[r];
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; x0[r] *= alpha; x1[r] *= alpha; }
    // ---- O^T += V^T . P^T: instruction c covers keys 8c + 4*half + i = accumulator rows 4c + i of this lane
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      h16x4 ph, pl, vh, vl;
      split4(&st[4 * c], ph, pl);
      if (PACKED) unpack4(&va0[4 * c], vh, vl); else split4(&va0[4 * c], vh, vl);
      o0 = __builtin_amdgcn_mfma_f32_32x32x8f16(vh, ph, o0, 0, 0, 0);
      x0 = __builtin_amdgcn_mfma_f32_32x32x8f16(vh, pl, x0, 0, 0, 0);
      x0 = __builtin_amdgcn_mfma_f32_32x32x8f16(vl, ph, x0, 0, 0, 0);
      if (PACKED) unpack4(&va1[4 * c], vh, vl); else split4(&va1[4 * c], vh, vl);
      o1 = __builtin_amdgcn_mfma_f32_32x32x8f16(vh, ph, o1, 0, 0, 0);
      x1 = __builtin_amdgcn_mfma_f32_32x32x8f16(vh, pl, x1, 0, 0, 0);
      x1 = __builtin_amdgcn_mfma_f32_32x32x8f16(vl, ph, x1, 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < 32; ++s) kcur[s] = knext[s];
  }

  // ---- merge the 4 waves (as the exact kernel)
  s_m[wave][lane] = m_run;
  s_l[wave][lane] = l_run;
#pragma unroll
  for (int r = 0; r < 16; ++r) { s_o[wave][0][r][lane] = o0[r] + x0[r] * kInv; s_o[wave][1][r][lane] = o1[r] + x1[r] * kInv; }
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
  const float inv = lsum > 0.f ? 1.f / lsum : 0.f;
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


// ---------------------------------------------------------------------------------------------------------------------
// Packed-operand attention, second form (what simpb_attention_split_halfs launches): with the matrix work down to 56
// half-rate-free instructions per key tile (1 792 cycles) the ~600 vector instructions of a tile (mask, softmax, the
// split of P, address arithmetic: ~2 600 cycles) are the larger half, and one wave per SIMD runs the two strictly one
// after the other (S -> softmax -> O is a dependent chain inside a tile). So: EIGHT waves per workgroup = two per SIMD,
// whose matrix and vector phases interleave in hardware, under a 256-register budget (K in one register set, re-loaded
// right behind the S^T product that consumed it; no Kl.Ql accumulator: that term, 2^-22 of sum |q||k|, is added into the
// cross-term accumulator pre-scaled instead -- see below); and fewer vector instructions: exp2 on pre-multiplied
// arguments, no masks / clamps on full tiles (workgroup-uniform branch), accumulators rescaled only when some lane's
// running maximum moved. The eight partial results meet in two steps (waves 4-7 hand theirs to waves 0-3 through LDS,
// then the four-way meeting of the kernels above), so LDS stays at 35 KB.
// (launch bounds: at least 8 waves per CU in either form, i.e. at most 256 registers per wave. The four-wave form -- used for the
// camera-grouped launches, whose ~190-key groups are 6 key tiles: nothing for eight waves to split -- then fits two workgroups
// per CU, and the 280 live workgroups of a 1 130-slot 2D set run as ONE round of the chip instead of 256 + 24.)
template <bool GROUPED, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 8 / WAVES) void attention_halfs_kernel(
    float* __restrict__ out, const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const int* __restrict__ query_cam, const int* __restrict__ group_start, int Nq, int Nk, int ldq, int ldk, int ldv,
    int ldo) {
  static_assert(WAVES == 4 || WAVES == 8, "four waves, or eight meeting in two steps");
  __shared__ float s_m[4][64];
  __shared__ float s_l[4][64];
  __shared__ float s_o[4][2][16][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int qi = lane & 31, half = lane >> 5;
  const int q0 = blockIdx.x * 32, head = blockIdx.y, b = blockIdx.z;
  const int qg = q0 + qi;
  const bool q_ok = qg < Nq;

  int kb = 0, ke = Nk;
  int my_lo = 0, my_hi = Nk;
  if (GROUPED) {
    const int cam_q = q_ok ? query_cam[qg] : -1;
    int lo = cam_q >= 0 ? group_start[cam_q] : 0x7fffffff;
    int hi = cam_q >= 0 ? group_start[cam_q + 1] : 0;
    my_lo = lo;
    my_hi = hi;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      lo = min(lo, __shfl_xor(lo, m));
      hi = max(hi, __shfl_xor(hi, m));
    }
    const bool any = hi > lo;
    kb = __builtin_amdgcn_readfirstlane(any ? lo : 0);
    ke = __builtin_amdgcn_readfirstlane(any ? min(hi, Nk) : 0);
  }

  h16x4 qh[8], ql[8];
  {
    const float* qp = q + ((size_t)b * Nq + (q_ok ? qg : 0)) * ldq + head * kHD + 32 * half;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float4 t = q_ok ? *reinterpret_cast<const float4*>(qp + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float x[4] = {t.x, t.y, t.z, t.w};
      unpack4(x, qh[c], ql[c]);
    }
  }

  float m_run = -INFINITY, l_run = 0.f;
  f32x16 o0, o1, x0, x1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; x0[r] = 0.f; x1[r] = 0.f; }

  const float* kbase = k + (size_t)b * Nk * ldk + head * kHD + 32 * half;
  const float* vbase = v + (size_t)b * Nk * ldv + head * kHD + qi;
  constexpr float kInv = 1.f / 2048.f, kLog2e = 1.4426950408889634f;
  // rows past the range read the first key of the range (finite halfs; their probability is 0): clamped loads on the
  // last, partial tile only
  auto load_k = [&](int kt, float (&kreg)[32], bool full) {
    const int kk = kt + qi;
    const float* kp = kbase + (size_t)(full || kk < ke ? kk : kb) * ldk;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float4 t = *reinterpret_cast<const float4*>(kp + 4 * j);
      kreg[4 * j + 0] = t.x; kreg[4 * j + 1] = t.y; kreg[4 * j + 2] = t.z; kreg[4 * j + 3] = t.w;
    }
  };
  float kcur[32];
  const int kt0 = kb + 32 * wave;
  if (kt0 < ke) load_k(kt0, kcur, kt0 + 32 <= ke);
  for (int kt = kt0; kt < ke; kt += 32 * WAVES) {
    const bool full = kt + 32 <= ke;   // wave-uniform
    float va0[16], va1[16];
    if (full) {
      const float* vt = vbase + (size_t)(kt + 4 * half) * ldv;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float* vp = vt + (size_t)((s & 3) + 8 * (s >> 2)) * ldv;
        va0[s] = vp[0];
        va1[s] = vp[32];
      }
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int kidx = kt + acc_row(s, half);
        const float* vp = vbase + (size_t)(kidx < ke ? kidx : kb) * ldv;
        va0[s] = vp[0];
        va1[s] = vp[32];
      }
    }
    // ---- S^T = K_tile . Q^T: leading term / cross terms and the trailing term pre-scaled into one accumulator
    f32x16 shh, sx;
#pragma unroll
    for (int r = 0; r < 16; ++r) { shh[r] = 0.f; sx[r] = 0.f; }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      h16x4 kh, kl;
      unpack4(&kcur[4 * c], kh, kl);
      shh = __builtin_amdgcn_mfma_f32_32x32x8f16(kh, qh[c], shh, 0, 0, 0);
      sx = __builtin_amdgcn_mfma_f32_32x32x8f16(kh, ql[c], sx, 0, 0, 0);
      sx = __builtin_amdgcn_mfma_f32_32x32x8f16(kl, qh[c], sx, 0, 0, 0);
    }
    // (Kl.Ql: 2^-22 of sum |q||k| -- dropped like the trailing term of the second product; the S^T of this kernel is the
    // three-term form csrc/linear_split.hip uses for value_proj, bounded by tests/test_gpu_ops.py against float64)
    const int ktn = kt + 32 * WAVES;
    load_k(ktn, kcur, ktn + 32 <= ke);   // this wave's next K tile into the registers the product above has read
    // ---- mask + online softmax in base 2
    float st[16];
    float mt = -INFINITY;
    if (full && !GROUPED) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        st[r] = (shh[r] + sx[r] * kInv) * kLog2e;
        mt = fmaxf(mt, st[r]);
      }
      if (!q_ok) mt = -INFINITY;
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kidx = kt + acc_row(r, half);
        bool ok = kidx < ke && q_ok;
        if (GROUPED) ok = ok && kidx >= my_lo && kidx < my_hi;
        st[r] = ok ? (shh[r] + sx[r] * kInv) * kLog2e : -INFINITY;
        mt = fmaxf(mt, st[r]);
      }
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32));
    const float m_new = fmaxf(m_run, mt);
    const float m_safe = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_safe);  // m_run = -inf -> 0
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      st[r] = q_ok ? __builtin_amdgcn_exp2f(st[r] - m_safe) : 0.f;
      psum += st[r];
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
    if (__builtin_amdgcn_ballot_w64(alpha != 1.f)) {   // some lane's maximum moved: rescale (wave-uniform branch)
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; x0[r] *= alpha; x1[r] *= alpha; }
    }
    // ---- O^T += V^T . P^T
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      h16x4 ph, pl, vh, vl;
      split4(&st[4 * c], ph, pl);
      unpack4(&va0[4 * c], vh, vl);
      o0 = __builtin_amdgcn_mfma_f32_32x32x8f16(vh, ph, o0, 0, 0, 0);
      x0 = __builtin_amdgcn_mfma_f32_32x32x8f16(vh, pl, x0, 0, 0, 0);
      x0 = __builtin_amdgcn_mfma_f32_32x32x8f16(vl, ph, x0, 0, 0, 0);
      unpack4(&va1[4 * c], vh, vl);
      o1 = __builtin_amdgcn_mfma_f32_32x32x8f16(vh, ph, o1, 0, 0, 0);
      x1 = __builtin_amdgcn_mfma_f32_32x32x8f16(vh, pl, x1, 0, 0, 0);
      x1 = __builtin_amdgcn_mfma_f32_32x32x8f16(vl, ph, x1, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] += x0[r] * kInv; o1[r] += x1[r] * kInv; }

  // ---- eight waves: waves 4..7 hand their partials to waves 0..3 (running maxima are base-2 exponents here)
  if (WAVES == 8) {
    if (wave >= 4) {
      s_m[wave - 4][lane] = m_run;
      s_l[wave - 4][lane] = l_run;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s_o[wave - 4][0][r][lane] = o0[r]; s_o[wave - 4][1][r][lane] = o1[r]; }
    }
    __syncthreads();
    if (wave < 4) {
      const float m_b = s_m[wave][lane], l_b = s_l[wave][lane];
      const float m_new = fmaxf(m_run, m_b);
      const float m_safe = m_new == -INFINITY ? 0.f : m_new;
      const float fa = __builtin_amdgcn_exp2f(m_run - m_safe), fb = __builtin_amdgcn_exp2f(m_b - m_safe);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        o0[r] = o0[r] * fa + s_o[wave][0][r][lane] * fb;
        o1[r] = o1[r] * fa + s_o[wave][1][r][lane] * fb;
      }
      l_run = l_run * fa + l_b * fb;
      m_run = m_new;
    }
    __syncthreads();
  }
  // ---- the four-way meeting (waves 4..7 of the eight-wave form only keep the barrier company)
  if (wave < 4) {
    s_m[wave][lane] = m_run;
    s_l[wave][lane] = l_run;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s_o[wave][0][r][lane] = o0[r]; s_o[wave][1][r][lane] = o1[r]; }
  }
  __syncthreads();
  if (wave >= 4) return;
  float mx = -INFINITY;
#pragma unroll
  for (int w = 0; w < 4; ++w) mx = fmaxf(mx, s_m[w][lane]);
  float f[4];
  float lsum = 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    f[w] = mx == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(s_m[w][lane] - mx);
    lsum += f[w] * (s_l[w][lane] + s_l[w][lane ^ 32]);
  }
  const float inv = lsum > 0.f ? 1.f / lsum : 0.f;
  const int t = wave / 2, r0 = 8 * (wave % 2);
  if (q_ok) {
    float* op = out + ((size_t)b * Nq + qg) * ldo + head * kHD + 32 * t;
#pragma unroll
    for (int r = r0; r < r0 + 8; ++r) {
      float acc = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) acc += f[w] * s_o[w][t][r][lane];
      op[acc_row(r, half)] = acc * inv;
    }
  }
}

}  // namespace

static int attention_launch(int split, float* out, const float* q, const float* k, const float* v, const int* query_cam,
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
  if (split == 2 && query_cam)
    hipLaunchKernelGGL((attention_halfs_kernel<true, 4>), grid, dim3(256), 0, s, out, q, k, v, query_cam, group_start, num_query,
                       num_key, ldq, ldk, ldv, ldo);
  else if (split == 2)
    hipLaunchKernelGGL((attention_halfs_kernel<false, 8>), grid, dim3(512), 0, s, out, q, k, v, query_cam, group_start, num_query,
                       num_key, ldq, ldk, ldv, ldo);
  else if (split && query_cam)
    hipLaunchKernelGGL((attention_f16s_kernel<true>), grid, block, 0, s, out, q, k, v, query_cam, group_start, num_query,
                       num_key, ldq, ldk, ldv, ldo, scale);
  else if (split)
    hipLaunchKernelGGL((attention_f16s_kernel<false>), grid, block, 0, s, out, q, k, v, query_cam, group_start, num_query,
                       num_key, ldq, ldk, ldv, ldo, scale);
  else if (query_cam)
    hipLaunchKernelGGL(attention_f32_kernel<true>, grid, block, 0, s, out, q, k, v, query_cam, group_start, num_query,
                       num_key, ldq, ldk, ldv, ldo, scale);
  else
    hipLaunchKernelGGL(attention_f32_kernel<false>, grid, block, 0, s, out, q, k, v, query_cam, group_start, num_query,
                       num_key, ldq, ldk, ldv, ldo, scale);
  return simpb_check_launch();
}

extern "C" int simpb_attention_f32(float* out, const float* q, const float* k, const float* v, const int* query_cam,
                                   const int* group_start, int batch_size, int num_heads, int head_dim, int num_query,
                                   int num_key, int ldq, int ldk, int ldv, int ldo, float scale, void* stream) {
  return attention_launch(0, out, q, k, v, query_cam, group_start, batch_size, num_heads, head_dim, num_query, num_key, ldq,
                          ldk, ldv, ldo, scale, stream);
}

extern "C" int simpb_attention_f32_split(float* out, const float* q, const float* k, const float* v, const int* query_cam,
                                         const int* group_start, int batch_size, int num_heads, int head_dim, int num_query,
                                         int num_key, int ldq, int ldk, int ldv, int ldo, float scale, void* stream) {
  return attention_launch(1, out, q, k, v, query_cam, group_start, batch_size, num_heads, head_dim, num_query, num_key, ldq,
                          ldk, ldv, ldo, scale, stream);
}

extern "C" int simpb_attention_split_halfs(float* out, const void* q, const void* k, const void* v, const int* query_cam,
                                           const int* group_start, int batch_size, int num_heads, int head_dim, int num_query,
                                           int num_key, int ldq, int ldk, int ldv, int ldo, void* stream) {
  return attention_launch(2, out, static_cast<const float*>(q), static_cast<const float*>(k), static_cast<const float*>(v),
                          query_cam, group_start, batch_size, num_heads, head_dim, num_query, num_key, ldq, ldk, ldv, ldo, 1.f,
                          stream);
}
