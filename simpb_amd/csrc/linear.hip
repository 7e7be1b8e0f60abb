// y = x . W^T + b in exact fp32 on the f32 matrix cores (gfx950 v_mfma_f32_32x32x2_f32).
//
// Written for value_proj of the camera-grouped deformable cross-attention
// (/root/reference/projects/mmdet3d_plugin/models/group_attn.py:176): x = every camera token
// [89 760, 256] (704x256) or [359 040, 256] (1408x512), W = [256, 256]: the largest GEMM of the
// decoder (11.8 GFLOP per layer) and tall-skinny, which is where the vendor
// heuristic picks a 256x16 tile (measured 1 020 us = 11.5 TFLOP/s). Also used for the other
// nn.Linear layers of the head whose M is large enough.
//
// Tiling: one workgroup = 4 waves = 64 rows x 256 columns of y, so every row of x is read from
// HBM exactly once per 256 output columns; K is walked in chunks of 32 through LDS (x chunk 64x32,
// W chunk 256x32, rows padded to 36 floats so the 16-lane groups of ds_read_b128 hit 16 distinct
// 4-bank slots). Wave w owns columns [64w, 64w+64): a 2x2 grid of 32x32 MFMA tiles (64 accumulator
// registers). K order inside a chunk is permuted (lane half h takes k = 16h .. 16h+15) so that each
// lane's 16 operand values are contiguous in LDS and arrive as four ds_read_b128; A and B use the
// same permutation, so the sum is unchanged. The next chunk is fetched into registers while the
// current one is multiplied.
#include <hip/hip_runtime.h>
#include "../../include/simpb_hip.h"

extern "C" int simpb_check_launch(void);

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BM = 64, BN = 256, BK = 32, LDK = BK + 4;  // LDS row stride in floats
constexpr int kThreads = 256;

template <bool RELU>
__global__ __launch_bounds__(kThreads) void linear_f32_mfma(float* __restrict__ y, const float* __restrict__ x,
                                                            const float* __restrict__ w,
                                                            const float* __restrict__ bias, int M, int N, int K) {
  __shared__ float s_x[BM * LDK];
  __shared__ float s_w[BN * LDK];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r32 = lane & 31, half = lane >> 5;
  const int row0 = blockIdx.x * BM;
  const int col0 = blockIdx.y * BN;

  // staging: x chunk = 64 rows x 8 float4 -> 2 float4 per thread; W chunk = 256 rows x 8 float4 -> 8 per thread
  float4 px[2], pw[8];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int f = tid + i * kThreads;  // float4 index in the 64x8 chunk
      const int r = f >> 3, c4 = f & 7;
      const int gr = row0 + r;
      px[i] = gr < M ? *reinterpret_cast<const float4*>(x + (size_t)gr * K + k0 + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int f = tid + i * kThreads;
      const int r = f >> 3, c4 = f & 7;
      const int gc = col0 + r;
      pw[i] = gc < N ? *reinterpret_cast<const float4*>(w + (size_t)gc * K + k0 + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int f = tid + i * kThreads;
      *reinterpret_cast<float4*>(&s_x[(f >> 3) * LDK + (f & 7) * 4]) = px[i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int f = tid + i * kThreads;
      *reinterpret_cast<float4*>(&s_w[(f >> 3) * LDK + (f & 7) * 4]) = pw[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  fetch(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();  // everyone is done reading the previous chunk
    stash();
    __syncthreads();
    if (k0 + BK < K) fetch(k0 + BK);
    // operands of this chunk: lane (r32, half) holds k = 16*half + 0..15 of its row / column
    float a[2][16], b[2][16];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(&s_x[(m * 32 + r32) * LDK + half * 16 + q * 4]);
        a[m][q * 4 + 0] = v.x; a[m][q * 4 + 1] = v.y; a[m][q * 4 + 2] = v.z; a[m][q * 4 + 3] = v.w;
      }
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(&s_w[(wave * 64 + n * 32 + r32) * LDK + half * 16 + q * 4]);
        b[n][q * 4 + 0] = v.x; b[n][q * 4 + 1] = v.y; b[n][q * 4 + 2] = v.z; b[n][q * 4 + 3] = v.w;
      }
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][s], b[n][s], acc[m][n], 0, 0, 0);
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
        const int gr = row0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (gr < M) {
          float v = acc[m][n][r] + bv;
          if (RELU) v = fmaxf(v, 0.f);
          y[(size_t)gr * N + gc] = v;
        }
      }
  }
}

}  // namespace

extern "C" int simpb_linear_f32(float* y, const float* x, const float* weight, const float* bias, int M, int N, int K,
                                int relu, void* stream) {
  if (!y || !x || !weight || M <= 0 || N <= 0 || K <= 0 || K % BK != 0) return SIMPB_EINVAL;
  if ((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(weight)) & 15) return SIMPB_EINVAL;
  (void)hipGetLastError();
  dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN);
  if (grid.y > 65535) return SIMPB_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (relu)
    hipLaunchKernelGGL(linear_f32_mfma<true>, grid, dim3(kThreads), 0, s, y, x, weight, bias, M, N, K);
  else
    hipLaunchKernelGGL(linear_f32_mfma<false>, grid, dim3(kThreads), 0, s, y, x, weight, bias, M, N, K);
  return simpb_check_launch();
}
