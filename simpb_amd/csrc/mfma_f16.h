// FP16 matrix-core step used by conv1x1.hip, linear_split.hip and the split-fp16 variant of gemm.hip:
// D[32x32] += A[32x16] . B[16x32], fp32 accumulate, lane (r32, kb) holding k = 8*kb .. 8*kb+7 of its row/column.
//
// gfx950 has this as ONE instruction, v_mfma_f32_32x32x16_f16 (twice the K of gfx942's 32x32x8). Round 2 measurement
// (tools/daf_stress.py, profiles/r02_mfma_x16_interference/): while any wave of the chip executes that instruction,
// OTHER kernels' plain vector arithmetic goes wrong in lanes 48-63 of a wave now and then -- daf_fwd_rows beside a loop of
// conv1x1 or linear_split: 90-97 % of its launches return wrong channels 192-255 for ~10 anchors; beside fp32-MFMA
// kernels (linear_f32, hipBLASLt), MIOpen's fp16 3x3 convolution, copies or elementwise kernels: 0 of 1 500. This is
// what the "eager two-stream fault" of round 1 was. A register-only burner (tools/mfma_burn.hip) shows the same for all four
// double-K 16-bit instructions gfx950 adds (f16 / bf16, 32x32x16 / 16x16x32) and for none of the older ones. The same product as TWO v_mfma_f32_32x32x8f16 steps (each lane's
// eight k-values split into its low and high four: both operands use the same lane -> k map, so the products pair up
// and only the summation order changes) does not disturb other waves. SIMPB_MFMA_F16_K16=1 selects the single
// instruction again (for the measurement above, never for the product).
#pragma once
#include <hip/hip_runtime.h>

namespace simpb {

typedef _Float16 h16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

#ifndef SIMPB_MFMA_F16_K16
#define SIMPB_MFMA_F16_K16 0
#endif

template <class A8, class C16>
__device__ __forceinline__ C16 mfma_32x32x16_f16(const A8& a, const A8& b, const C16& c) {
#if SIMPB_MFMA_F16_K16
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#else
  const h16x4_t a0 = {a[0], a[1], a[2], a[3]}, a1 = {a[4], a[5], a[6], a[7]};
  const h16x4_t b0 = {b[0], b[1], b[2], b[3]}, b1 = {b[4], b[5], b[6], b[7]};
  C16 d = __builtin_amdgcn_mfma_f32_32x32x8f16(a0, b0, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x8f16(a1, b1, d, 0, 0, 0);
#endif
}

}  // namespace simpb
