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

// Stem epilogue: y[n, oy, ox, c] = relu(max over the 3x3 window (stride 2, padding 1) of x + bias[c]) -- mmdet ResNet's
// conv1 -> bn1 (folded) -> relu -> maxpool. Bias and ReLU commute with the max (x -> round(relu(x + b)) is monotone), so the
// window maximum is taken on the raw convolution output and the epilogue applied once per POOLED element: one pass that
// reads the 34.6 MB map and writes 8.65 MB, instead of the in-place bias pass (read + write 34.6 MB) and the pooling kernel
// (read 34.6, write 8.65). Bit-equal to the two-pass route. Thread = one pooled pixel x 8 channels.
__global__ void bias_relu_maxpool_kernel(__half* __restrict__ y, const __half* __restrict__ x, const __half* __restrict__ bias,
                                         int N, int H, int W, int Ho, int Wo, int c8) {
  const long long total = (long long)N * Ho * Wo * c8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c8);
    long long p = i / c8;
    const int ox = (int)(p % Wo);
    p /= Wo;
    const int oy = (int)(p % Ho), n = (int)(p / Ho);
    uint4 t[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {   // taps outside the map repeat the centre (always inside): max is unchanged
      const int iy = 2 * oy - 1 + k / 3, ix = 2 * ox - 1 + k % 3;
      const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
      const int yy = in ? iy : 2 * oy, xx = in ? ix : 2 * ox;
      t[k] = *reinterpret_cast<const uint4*>(x + (((size_t)n * H + yy) * W + xx) * (size_t)(c8 * 8) + cc * 8);
    }
    const uint4 bv = *reinterpret_cast<const uint4*>(bias + cc * 8);
    const __half2* b2 = reinterpret_cast<const __half2*>(&bv);
    uint4 o;
    __half2* o2 = reinterpret_cast<__half2*>(&o);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float2 m = __half22float2(reinterpret_cast<const __half2*>(&t[0])[j]);
#pragma unroll
      for (int k = 1; k < 9; ++k) {
        const float2 v = __half22float2(reinterpret_cast<const __half2*>(&t[k])[j]);
        m.x = fmaxf(m.x, v.x);
        m.y = fmaxf(m.y, v.y);
      }
      const float2 b = __half22float2(b2[j]);
      m.x = fmaxf(m.x + b.x, 0.f);
      m.y = fmaxf(m.y + b.y, 0.f);
      o2[j] = __float22half2_rn(m);
    }
    *reinterpret_cast<uint4*>(y + i * 8) = o;
  }
}

}  // namespace

extern "C" int simpb_bias_relu_maxpool_nhwc_f16(void* y, const void* x, const void* bias, int num_images, int in_h, int in_w,
                                                int channels, void* stream) {
  if (!y || !x || !bias || num_images <= 0 || in_h <= 0 || in_w <= 0 || channels <= 0 || channels % 8 != 0) return SIMPB_EINVAL;
  if ((reinterpret_cast<size_t>(y) | reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(bias)) & 15) return SIMPB_EINVAL;
  (void)hipGetLastError();
  const int ho = (in_h - 1) / 2 + 1, wo = (in_w - 1) / 2 + 1;   // kernel 3, stride 2, padding 1
  const long long total = (long long)num_images * ho * wo * (channels / 8);
  long long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(bias_relu_maxpool_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<__half*>(y), static_cast<const __half*>(x), static_cast<const __half*>(bias), num_images, in_h,
                     in_w, ho, wo, channels / 8);
  return simpb_check_launch();
}

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
