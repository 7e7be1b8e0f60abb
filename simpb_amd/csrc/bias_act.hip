// Convolution epilogue for the folded-BatchNorm backbone (gfx950): y = act(y + bias[c] (+ residual)),
// in place, on channels_last (NHWC) fp16 tensors. After conv-BN folding
// (/root/reference/tools/fuse_conv_bn.py:10-48) every ResNet convolution is followed by a bias add, a
// ReLU and, at the end of a bottleneck, the residual add: three elementwise kernels and three trips
// through HBM where one suffices. Memory-bound: 16 B per lane, fp32 arithmetic, one rounding.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "../../include/simpb_hip.h"

extern "C" int simpb_check_launch(void);

namespace {

__global__ void bias_act_nhwc_f16_kernel(__half* __restrict__ y, const __half* __restrict__ bias,
                                         const __half* __restrict__ res, long long n8, int c8, int relu) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c8) * 8;
    uint4 yv = *reinterpret_cast<const uint4*>(y + i * 8);
    const uint4 bv = *reinterpret_cast<const uint4*>(bias + c);
    uint4 rv = make_uint4(0, 0, 0, 0);
    if (res) rv = *reinterpret_cast<const uint4*>(res + i * 8);
    __half2* y2 = reinterpret_cast<__half2*>(&yv);
    const __half2* b2 = reinterpret_cast<const __half2*>(&bv);
    const __half2* r2 = reinterpret_cast<const __half2*>(&rv);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float2 v = __half22float2(y2[j]);
      const float2 b = __half22float2(b2[j]);
      v.x += b.x; v.y += b.y;
      if (res) { const float2 r = __half22float2(r2[j]); v.x += r.x; v.y += r.y; }
      if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); }
      y2[j] = __float22half2_rn(v);
    }
    *reinterpret_cast<uint4*>(y + i * 8) = yv;
  }
}

}  // namespace

extern "C" int simpb_bias_act_nhwc_f16(void* y, const void* bias, const void* residual, long long num_pixels,
                                       int channels, int relu, void* stream) {
  if (!y || !bias || num_pixels <= 0 || channels <= 0 || channels % 8 != 0) return SIMPB_EINVAL;
  if ((reinterpret_cast<size_t>(y) | reinterpret_cast<size_t>(bias) | reinterpret_cast<size_t>(residual)) & 15)
    return SIMPB_EINVAL;
  (void)hipGetLastError();
  const long long n8 = num_pixels * (channels / 8);
  long long blocks = (n8 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(bias_act_nhwc_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<__half*>(y), static_cast<const __half*>(bias), static_cast<const __half*>(residual), n8,
                     channels / 8, relu);
  return simpb_check_launch();
}
