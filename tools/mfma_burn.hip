// tools/daf_stress.py --co burn:<variant>: a co-runner that does nothing but issue one matrix instruction in a loop on
// register operands (no memory traffic, no LDS), to tell WHICH matrix instructions disturb other kernels' vector arithmetic
// (round 2: v_mfma_f32_32x32x16_f16 does). Built on the GPU box by the tool (hipcc --offload-arch=gfx950), not part of the library.
#include <hip/hip_runtime.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef int i16v __attribute__((ext_vector_type(16)));
typedef int i4v __attribute__((ext_vector_type(4)));

template <int V>
__global__ __launch_bounds__(256) void burn(float* sink, int iters) {
  const float seed = (float)(threadIdx.x & 7) * 0.125f;
  f16v acc = {0};
  f4v acc4 = {0, 0, 0, 0}, q1 = {0, 0, 0, 0}, q2 = {0, 0, 0, 0}, q3 = {0, 0, 0, 0};
  h8 a8, c8;
  h4 a4, c4;
  b8 ab, cb;
  for (int e = 0; e < 8; ++e) { a8[e] = (_Float16)(seed + e); c8[e] = (_Float16)(0.01f * e); ab[e] = (__bf16)(seed + e); cb[e] = (__bf16)(0.01f * e); }
  for (int e = 0; e < 4; ++e) { a4[e] = a8[e]; c4[e] = c8[e]; }
  for (int i = 0; i < iters; ++i) {
    if constexpr (V == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, c8, acc, 0, 0, 0);
    if constexpr (V == 1) acc4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, c8, acc4, 0, 0, 0);
    if constexpr (V == 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, cb, acc, 0, 0, 0);
    if constexpr (V == 3) acc4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, cb, acc4, 0, 0, 0);
    if constexpr (V == 4) acc = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, c4, acc, 0, 0, 0);
    if constexpr (V == 5) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, 0.5f, acc, 0, 0, 0);
    if constexpr (V == 6) acc4 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, c4, acc4, 0, 0, 0);
    // the small-tile instructions of csrc/mlp_chain.hip, four independent accumulators each (throughput, not latency)
    if constexpr (V == 7) {
      acc4 = __builtin_amdgcn_mfma_f32_4x4x1f32(seed, 0.5f, acc4, 0, 0, 0);
      q1 = __builtin_amdgcn_mfma_f32_4x4x1f32(seed, 0.25f, q1, 0, 0, 0);
      q2 = __builtin_amdgcn_mfma_f32_4x4x1f32(seed, 0.125f, q2, 0, 0, 0);
      q3 = __builtin_amdgcn_mfma_f32_4x4x1f32(seed, 0.75f, q3, 0, 0, 0);
    }
    if constexpr (V == 8) {
      acc4 = __builtin_amdgcn_mfma_f32_4x4x4f16(a4, c4, acc4, 0, 0, 0);
      q1 = __builtin_amdgcn_mfma_f32_4x4x4f16(c4, a4, q1, 0, 0, 0);
      q2 = __builtin_amdgcn_mfma_f32_4x4x4f16(a4, a4, q2, 0, 0, 0);
      q3 = __builtin_amdgcn_mfma_f32_4x4x4f16(c4, c4, q3, 0, 0, 0);
    }
    if constexpr (V == 9) {
      acc4 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, 0.5f, acc4, 0, 0, 0);
      q1 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, 0.25f, q1, 0, 0, 0);
      q2 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, 0.125f, q2, 0, 0, 0);
      q3 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, 0.75f, q3, 0, 0, 0);
    }
    if constexpr (V == 10) {
      acc4 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, c4, acc4, 0, 0, 0);
      q1 = __builtin_amdgcn_mfma_f32_16x16x16f16(c4, a4, q1, 0, 0, 0);
      q2 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, a4, q2, 0, 0, 0);
      q3 = __builtin_amdgcn_mfma_f32_16x16x16f16(c4, c4, q3, 0, 0, 0);
    }
  }
  for (int e = 0; e < 4; ++e) acc4[e] += q1[e] + q2[e] + q3[e];
  float s = acc4[0] + acc4[1] + acc4[2] + acc4[3];
  for (int e = 0; e < 16; ++e) s += acc[e];
  if (s == 12345.678f) sink[0] = s;  // keep the loop
}

extern "C" int mfma_burn(int variant, float* sink, int blocks, int iters, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (variant) {
    case 0: hipLaunchKernelGGL(burn<0>, dim3(blocks), dim3(256), 0, s, sink, iters); break;
    case 1: hipLaunchKernelGGL(burn<1>, dim3(blocks), dim3(256), 0, s, sink, iters); break;
    case 2: hipLaunchKernelGGL(burn<2>, dim3(blocks), dim3(256), 0, s, sink, iters); break;
    case 3: hipLaunchKernelGGL(burn<3>, dim3(blocks), dim3(256), 0, s, sink, iters); break;
    case 4: hipLaunchKernelGGL(burn<4>, dim3(blocks), dim3(256), 0, s, sink, iters); break;
    case 5: hipLaunchKernelGGL(burn<5>, dim3(blocks), dim3(256), 0, s, sink, iters); break;
    case 6: hipLaunchKernelGGL(burn<6>, dim3(blocks), dim3(256), 0, s, sink, iters); break;
    case 7: hipLaunchKernelGGL(burn<7>, dim3(blocks), dim3(256), 0, s, sink, iters); break;
    case 8: hipLaunchKernelGGL(burn<8>, dim3(blocks), dim3(256), 0, s, sink, iters); break;
    case 9: hipLaunchKernelGGL(burn<9>, dim3(blocks), dim3(256), 0, s, sink, iters); break;
    case 10: hipLaunchKernelGGL(burn<10>, dim3(blocks), dim3(256), 0, s, sink, iters); break;
    default: return 1;
  }
  return (int)hipGetLastError();
}
