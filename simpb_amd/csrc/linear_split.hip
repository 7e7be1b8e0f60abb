// y = x . W^T + b with fp32 accuracy on the FP16 matrix cores (gfx950 v_mfma_f32_32x32x16_f16):
// both operands are split into a leading and a trailing half-precision part,
//   x = xh + xl / 2048,  W = Wh + Wl / 2048   (xh = half(x), xl = half((x - xh) * 2048); same for W, done
//   once on the host; the trailing parts are kept scaled by 2^11 so that they stay normal half-precision
//   numbers -- no dependence on how the matrix pipe treats fp16 subnormals)
//   x . W^T  ~=  xh.Wh^T + (xh.Wl^T + xl.Wh^T) / 2048      (the dropped xl.Wl^T term is ~2^-22 relative)
// with every product exact in the fp32 accumulators (one for the leading term, one for the scaled sum). Three passes on a matrix pipe that is 16x the
// fp32 one = 5.3x the fp32 matrix rate at fp32-grade accuracy (measured error against float64:
// tests/test_gpu_ops.py, same 2e-5 bound as the exact kernel).
//
// Used for value_proj of the camera-grouped deformable cross-attention
// (/root/reference/projects/mmdet3d_plugin/models/group_attn.py:176): 89 760 x 256 x 256 per layer,
// 35 GFLOP per frame over its three layers -- the largest block of FLOPs in the decoder. On the exact
// fp32 matrix cores (csrc/linear.hip) it is compute-bound at 142 us (83 TFLOP/s, 54 % MFMA busy);
// here it is bound by reading x and writing y once (184 MB).
//
// Tiling: one workgroup = 8 waves = 128 rows x 256 columns; a wave owns 64 rows x 64 columns as 2x2
// tiles of 32x32; K in chunks of 32 through LDS. x is split while it is staged. (With 64-row
// workgroups the 256 KB of split weights were re-read from L2 by 1 403 workgroups = 359 MB per call,
// and that, not the matrix pipe, set the time: 123 us.)
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <type_traits>
#include "../../include/simpb_hip.h"
#include "mfma_f16.h"

extern "C" int simpb_check_launch(void);

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using h16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int WM = 2;                       // waves along M: W is re-read from L2 once per 64*WM rows
constexpr int BM = 64 * WM, BN = 256, BK = 32;
constexpr int LDH = BK + 8;  // LDS row stride in halfs (80 B): 16-lane groups of ds_read_b128 hit 16 distinct 4-bank slots
constexpr int kThreads = 256 * WM;

__global__ __launch_bounds__(kThreads) void linear_f16x3_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                                const _Float16* __restrict__ wh, const _Float16* __restrict__ wl,
                                                                const float* __restrict__ bias, int M, int N, int K) {
  __shared__ _Float16 s_xh[BM * LDH], s_xl[BM * LDH];
  __shared__ _Float16 s_wh[BN * LDH], s_wl[BN * LDH];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = (tid >> 6) & 3, wm = tid >> 8;  // wave: column group, wm: 64-row group
  const int r32 = lane & 31, kb = lane >> 5;
  const int row0 = blockIdx.x * BM, col0 = blockIdx.y * BN;

  // staging: x chunk 64 x 32 floats: thread -> (row tid/4, 8 floats at column 8*(tid%4));
  // W chunks 256 x 32 halfs: thread -> (row tid/4 + 64 i, 8 halfs at column 8*(tid%4)), i = 0..3. Loads are
  // unconditional (row / column indices clamped; surplus rows and columns are never stored).
  constexpr int NWL = 4 / WM;  // W rows per thread per chunk
  const int sr = tid >> 2, sc = (tid & 3) * 8;
  const float* xp = x + (size_t)min(row0 + sr, M - 1) * K + sc;
  size_t wofs[NWL];
#pragma unroll
  for (int i = 0; i < NWL; ++i) wofs[i] = (size_t)min(col0 + sr + 64 * WM * i, N - 1) * K + sc;
  f32x4 px[2];
  h16x8 pwh[NWL], pwl[NWL];
  auto fetch = [&](int k0) __attribute__((always_inline)) {
    px[0] = *reinterpret_cast<const f32x4*>(xp + k0);
    px[1] = *reinterpret_cast<const f32x4*>(xp + k0 + 4);
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
      pwh[i] = *reinterpret_cast<const h16x8*>(wh + wofs[i] + k0);
      pwl[i] = *reinterpret_cast<const h16x8*>(wl + wofs[i] + k0);
    }
  };
  auto stash = [&]() __attribute__((always_inline)) {
    h16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = e < 4 ? px[0][e] : px[1][e - 4];
      const _Float16 h = (_Float16)v;
      hi[e] = h;
      lo[e] = (_Float16)((v - (float)h) * 2048.f);
    }
    *reinterpret_cast<h16x8*>(&s_xh[sr * LDH + sc]) = hi;
    *reinterpret_cast<h16x8*>(&s_xl[sr * LDH + sc]) = lo;
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
      *reinterpret_cast<h16x8*>(&s_wh[(sr + 64 * WM * i) * LDH + sc]) = pwh[i];
      *reinterpret_cast<h16x8*>(&s_wl[(sr + 64 * WM * i) * LDH + sc]) = pwl[i];
    }
  };

  f32x16 acc[2][2], acs[2][2];  // leading term / trailing terms (scaled by 2^11)
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[m][n][r] = 0.f; acs[m][n][r] = 0.f; }

  fetch(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();  // everyone is done reading the previous chunk
    stash();
    __syncthreads();
    fetch(k0 + BK < K ? k0 + BK : k0);  // the last iteration re-requests its own chunk (unconditional loads)
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      // lane (r32, kb) holds k = 16*ks + 8*kb .. +7 of its row (A) / column (B)
      h16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        ah[m] = *reinterpret_cast<const h16x8*>(&s_xh[(wm * 64 + m * 32 + r32) * LDH + 16 * ks + 8 * kb]);
        al[m] = *reinterpret_cast<const h16x8*>(&s_xl[(wm * 64 + m * 32 + r32) * LDH + 16 * ks + 8 * kb]);
      }
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        bh[n] = *reinterpret_cast<const h16x8*>(&s_wh[(wave * 64 + n * 32 + r32) * LDH + 16 * ks + 8 * kb]);
        bl[n] = *reinterpret_cast<const h16x8*>(&s_wl[(wave * 64 + n * 32 + r32) * LDH + 16 * ks + 8 * kb]);
      }
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          acs[m][n] = simpb::mfma_32x32x16_f16(al[m], bh[n], acs[m][n]);
          acs[m][n] = simpb::mfma_32x32x16_f16(ah[m], bl[n], acs[m][n]);
          acc[m][n] = simpb::mfma_32x32x16_f16(ah[m], bh[n], acc[m][n]);
        }
    }
  }

  // C/D layout of the 32x32 tile: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int gc = col0 + wave * 64 + n * 32 + r32;
    if (gc >= N) continue;
    const float bv = bias ? bias[gc] : 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gr = row0 + wm * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb;
        if (gr < M) y[(size_t)gr * N + gc] = acc[m][n][r] + acs[m][n][r] * (1.f / 2048.f) + bv;
      }
  }
}


// ---- the same product when x is ALREADY half precision --------------------------------------------------------------------
// The camera tokens value_proj reads are the output of the fp16 backbone widened to fp32 (simpb.py:63 `out_fp32`), so their
// trailing part is exactly zero and  x . W^T = x.Wh^T + (x.Wl^T) / 2048  with the same error as the three-pass form: two
// passes instead of three, half the bytes of x, and no splitting arithmetic. csrc/conv3x3.hip leaves the tokens in both
// forms (`tokens_f16`). Pipeline of csrc/conv3x3.hip's staged kernel: 128 rows x 64 output columns per workgroup (B tile =
// the 64 rows of Wh and the 64 rows of Wl), 4 waves as 2 x 2, each 64 rows x 32 columns with a leading and a trailing
// accumulator; K in chunks of 64 through a double-buffered LDS stage, three register sets, one barrier per chunk.
namespace h2 {
constexpr int BMt = 128, BNo = 64, BKc = 64;
constexpr int LDHc = BKc + 8;
constexpr int kRows = BMt + 2 * BNo;
constexpr int kThreads2 = 256;

struct Args {
  float* y;
  const _Float16 *x, *wh, *wl;
  const float* bias;
  int M, N, K, gx, gy, per_xcd;
};

template <int S>
using IC = std::integral_constant<int, S>;

__global__ __launch_bounds__(kThreads2, 2) void linear_h2_kernel(const Args a) {
  __shared__ __attribute__((aligned(16))) _Float16 s_ab[2 * kRows * LDHc];
  const int tile = (blockIdx.x & 7) * a.per_xcd + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= a.per_xcd || tile >= a.gx * a.gy) return;
  const int tx = tile / a.gy, ty = tile - tx * a.gy;   // column blocks of one row block next to each other (same XCD: x from L2)
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r32 = lane & 31, kb = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int row0 = tx * BMt, col0 = ty * BNo;
  const int K = a.K, nchunks = K / BKc;

  const int sr = tid >> 3, sc = (tid & 7) * 8;
  size_t xrow[4], wrow[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) xrow[i] = (size_t)min(row0 + sr + 32 * i, a.M - 1) * K + sc;
#pragma unroll
  for (int i = 0; i < 2; ++i) wrow[i] = (size_t)min(col0 + sr + 32 * i, a.N - 1) * K + sc;
  h16x8 rx[3][4], rh[3][2], rl[3][2];
  auto fetch = [&](auto set_c, int chunk) __attribute__((always_inline)) {
    constexpr int s = decltype(set_c)::value;
    const int k0 = min(chunk, nchunks - 1) * BKc;
#pragma unroll
    for (int i = 0; i < 4; ++i) rx[s][i] = *reinterpret_cast<const h16x8*>(a.x + xrow[i] + k0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      rh[s][i] = *reinterpret_cast<const h16x8*>(a.wh + wrow[i] + k0);
      rl[s][i] = *reinterpret_cast<const h16x8*>(a.wl + wrow[i] + k0);
    }
  };
  auto stash = [&](auto set_c, int buf) __attribute__((always_inline)) {
    constexpr int s = decltype(set_c)::value;
    _Float16* base = s_ab + buf * kRows * LDHc;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<h16x8*>(&base[(sr + 32 * i) * LDHc + sc]) = rx[s][i];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<h16x8*>(&base[(BMt + sr + 32 * i) * LDHc + sc]) = rh[s][i];
      *reinterpret_cast<h16x8*>(&base[(BMt + BNo + sr + 32 * i) * LDHc + sc]) = rl[s][i];
    }
  };
  f32x16 acc[2], acs[2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[m][r] = 0.f; acs[m][r] = 0.f; }
  auto multiply = [&](int buf) __attribute__((always_inline)) {
    const _Float16* base = s_ab + buf * kRows * LDHc + 8 * kb;
    const _Float16* sa = base + (wm * 64 + r32) * LDHc;
    const _Float16* sh = base + (BMt + wn * 32 + r32) * LDHc;
    const _Float16* sl = base + (BMt + BNo + wn * 32 + r32) * LDHc;
#pragma unroll
    for (int ks = 0; ks < BKc / 16; ++ks) {
      const h16x8 a0 = *reinterpret_cast<const h16x8*>(sa + 16 * ks);
      const h16x8 a1 = *reinterpret_cast<const h16x8*>(sa + 32 * LDHc + 16 * ks);
      const h16x8 bh = *reinterpret_cast<const h16x8*>(sh + 16 * ks);
      const h16x8 bl = *reinterpret_cast<const h16x8*>(sl + 16 * ks);
      acc[0] = simpb::mfma_32x32x16_f16(a0, bh, acc[0]);
      acs[0] = simpb::mfma_32x32x16_f16(a0, bl, acs[0]);
      acc[1] = simpb::mfma_32x32x16_f16(a1, bh, acc[1]);
      acs[1] = simpb::mfma_32x32x16_f16(a1, bl, acs[1]);
    }
  };
  fetch(IC<0>{}, 0);
  fetch(IC<1>{}, 1);
  fetch(IC<2>{}, 2);
  __builtin_amdgcn_sched_barrier(0);
  stash(IC<0>{}, 0);
  __syncthreads();
  auto step = [&](auto cur, auto nxt, int c) __attribute__((always_inline)) {
    fetch(cur, c + 3);
    __builtin_amdgcn_sched_barrier(0);
    stash(nxt, (c + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
    multiply(c & 1);
    __syncthreads();
  };
  for (int c = 0; c < nchunks; c += 3) {
    step(IC<0>{}, IC<1>{}, c);
    if (c + 1 < nchunks) step(IC<1>{}, IC<2>{}, c + 1);
    if (c + 2 < nchunks) step(IC<2>{}, IC<0>{}, c + 2);
  }
  // C/D layout of the 32x32 tile: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5): a store
  // instruction writes two whole 128-byte lines
  const int gc = col0 + wn * 32 + r32;
  if (gc < a.N) {
    const float bv = a.bias ? a.bias[gc] : 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gr = row0 + wm * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb;
        if (gr < a.M) a.y[(size_t)gr * a.N + gc] = acc[m][r] + acs[m][r] * (1.f / 2048.f) + bv;
      }
  }
}
}  // namespace h2

}  // namespace

extern "C" int simpb_linear_f16x3(float* y, const float* x, const void* weight_hi, const void* weight_lo, const float* bias,
                                  int M, int N, int K, void* stream) {
  if (!y || !x || !weight_hi || !weight_lo || M <= 0 || N <= 0 || K <= 0 || K % BK != 0) return SIMPB_EINVAL;
  if ((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(weight_hi) | reinterpret_cast<size_t>(weight_lo)) & 15)
    return SIMPB_EINVAL;
  (void)hipGetLastError();
  dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN);
  if (grid.y > 65535) return SIMPB_EINVAL;
  hipLaunchKernelGGL(linear_f16x3_kernel, grid, dim3(kThreads), 0, static_cast<hipStream_t>(stream), y, x,
                     static_cast<const _Float16*>(weight_hi), static_cast<const _Float16*>(weight_lo), bias, M, N, K);
  return simpb_check_launch();
}

extern "C" int simpb_linear_f16in_split(float* y, const void* x_f16, const void* weight_hi, const void* weight_lo,
                                        const float* bias, int M, int N, int K, void* stream) {
  if (!y || !x_f16 || !weight_hi || !weight_lo || M <= 0 || N <= 0 || K <= 0 || K % h2::BKc != 0) return SIMPB_EINVAL;
  if ((reinterpret_cast<size_t>(x_f16) | reinterpret_cast<size_t>(weight_hi) | reinterpret_cast<size_t>(weight_lo)) & 15)
    return SIMPB_EINVAL;
  (void)hipGetLastError();
  h2::Args a{y, static_cast<const _Float16*>(x_f16), static_cast<const _Float16*>(weight_hi),
             static_cast<const _Float16*>(weight_lo), bias, M, N, K, 0, 0, 0};
  a.gx = (M + h2::BMt - 1) / h2::BMt;
  a.gy = (N + h2::BNo - 1) / h2::BNo;
  const long long total = (long long)a.gx * a.gy;
  if (total > (1ll << 28)) return SIMPB_EINVAL;
  a.per_xcd = (int)((total + 7) / 8);
  hipLaunchKernelGGL(h2::linear_h2_kernel, dim3((unsigned)(a.per_xcd * 8)), dim3(h2::kThreads2), 0,
                     static_cast<hipStream_t>(stream), a);
  return simpb_check_launch();
}
