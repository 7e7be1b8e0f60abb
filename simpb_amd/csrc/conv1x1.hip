// 1x1 convolution of the fp16 channels_last backbone with its whole epilogue in one launch:
//   y[p, :] = relu?( x[pixel(p), :] . W^T + bias (+ residual[p, :]) )        fp16 in / out, fp32 accumulate
// 36 of the 53 convolutions of ResNet50 and the four FPN lateral convolutions are 1x1
// (/root/reference/projects/configs/simpb_nus_r50_img_704x256.py:79-99: mmdet ResNet style="pytorch" + FPN).
// After conv-BN folding (tools/fuse_conv_bn.py:10-48) each is a vendor convolution (12-30 us at 6 x 256 x 704)
// followed by an in-place bias / residual / ReLU pass over the activation map (csrc/bias_act.hip, 3-20 us):
// the map is written, read again with the residual and written again. These layers are bound by exactly that
// traffic (0.1-2 GFLOP over 10-100 MB), so here the map is written once. (MIOpen's own fused
// conv+bias+ReLU is 30-500x slower on these shapes: tools/bench_conv_fused.py.)
//
// One workgroup = 4 waves = 128 output pixels x 64 output channels, wave w owns pixels [32w, 32w+32) as two 32x32
// tiles of v_mfma_f32_32x32x16_f16; K in chunks of 64 through LDS (rows of 128 B staged as 16-byte pieces, two
// chunks in flight in registers, unconditional clamped loads). The accumulator tile goes through LDS once so that
// the residual is read and y is written in 16-byte pieces of full 128-byte rows. `stride` (1 or 2) subsamples
// the input pixels (the downsample branches of stages 2-4). Long-K layers (Cin >= 512) are forwarded to the staged
// pipeline of csrc/conv3x3.hip with one tap (double-buffered stage, one barrier per chunk, XCD-ordered tiles), where
// it measured faster (tools/bench_conv1x1.py); this kernel keeps the short ones and the input-side epilogue.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <type_traits>
#include "../../include/simpb_hip.h"
#include "mfma_f16.h"

extern "C" int simpb_check_launch(void);
// csrc/conv3x3.hip: the same convolution through its staged pipeline (TAPS = 1)
extern "C" int simpb_conv_pointwise_staged(void* y, const void* x, const void* weight, const void* bias, const void* residual,
                                           int p_out, int in_h, int in_w, int ho, int wo, int in_channels, int out_channels,
                                           int stride, int relu, int residual_upsample2x, int tiling, void* stream);

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using h16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int BM = 128, BN = 64, BK = 64;
constexpr int LDH = BK + 8;   // halfs per staged row (144 B): conflict-free 16-lane groups for ds_read_b128
constexpr int LDC = BN + 1;   // floats per row of the epilogue tile
constexpr int kThreads = 256;

// pixel index of output pixel p = (n, ho, wo) in a residual map of HALF the size, read with nearest-neighbour 2x
// upsampling (F.interpolate(mode="nearest") of an exact 2x: source = floor(dst / 2)): the FPN top-down path
__device__ __forceinline__ int up2(int p, int Ho, int Wo) {
  const int n = p / (Ho * Wo), rem = p - n * (Ho * Wo);
  const int ho = rem / Wo, wo = rem - ho * Wo;
  return (n * (Ho >> 1) + (ho >> 1)) * (Wo >> 1) + (wo >> 1);
}

__global__ __launch_bounds__(kThreads) void conv1x1_f16_kernel(_Float16* __restrict__ y, const _Float16* __restrict__ x,
                                                               const _Float16* __restrict__ w, const _Float16* __restrict__ bias,
                                                               const _Float16* __restrict__ residual, int P_out, int Cin,
                                                               int Cout, int relu, int stride, int Ho, int Wo, int H, int W,
                                                               int res_up, const _Float16* __restrict__ in_bias) {
  // the epilogue tile reuses the staging memory (33 KB per workgroup instead of 61: four workgroups per CU)
  constexpr int kStageBytes = (BM + BN) * LDH * 2, kTileBytes = BM * LDC * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[kStageBytes > kTileBytes ? kStageBytes : kTileBytes];
  _Float16* s_a = reinterpret_cast<_Float16*>(smem);
  _Float16* s_b = s_a + BM * LDH;
  float* s_c = reinterpret_cast<float*>(smem);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r32 = lane & 31, kb = lane >> 5;
  const int p0 = blockIdx.x * BM, c0 = blockIdx.y * BN;

  // staging: a 16-byte piece = 8 halfs; a chunk row has BK/8 = 8 pieces; A: 128 rows -> 4 pieces per thread,
  // B: 64 rows -> 2 per thread. Two register sets of chunks in flight, unconditional clamped loads.
  constexpr int C8 = BK / 8, RS = kThreads / C8, NA = BM / RS, NB = BN / RS;
  const int sr = tid / C8, sc = (tid % C8) * 8;
  size_t arow[NA], brow[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    int p = min(p0 + sr + RS * i, P_out - 1);
    if (stride != 1) {  // output pixel (n, ho, wo) reads input pixel (n, ho * stride, wo * stride)
      const int n = p / (Ho * Wo), rem = p - n * (Ho * Wo);
      const int ho = rem / Wo, wo = rem - ho * Wo;
      p = (n * H + ho * stride) * W + wo * stride;
    }
    arow[i] = (size_t)p * Cin + sc;
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) brow[i] = (size_t)min(c0 + sr + RS * i, Cout - 1) * Cin + sc;
  const int nchunks = Cin / BK;
  h16x8 pa[2][NA], pb[2][NB], pin[2];
  // in_bias: the INPUT is the raw output of the 3x3 convolution in front (ResNet bottleneck conv2 after BN folding); its
  // epilogue x <- relu(x + in_bias[channel]) is applied while the tile is staged (fp32 add, one rounding to fp16: the
  // numbers the separate bias_act pass wrote), so that pass and its trip through memory disappear
  auto fetch = [&](auto set_c, int chunk) __attribute__((always_inline)) {
    constexpr int set = decltype(set_c)::value;
    const int k0 = min(chunk, nchunks - 1) * BK;
    if (in_bias) pin[set] = *reinterpret_cast<const h16x8*>(in_bias + k0 + sc);
#pragma unroll
    for (int i = 0; i < NA; ++i) pa[set][i] = *reinterpret_cast<const h16x8*>(x + arow[i] + k0);
#pragma unroll
    for (int i = 0; i < NB; ++i) pb[set][i] = *reinterpret_cast<const h16x8*>(w + brow[i] + k0);
  };
  auto stash = [&](auto set_c, int) __attribute__((always_inline)) {
    constexpr int set = decltype(set_c)::value;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      h16x8 v = pa[set][i];
      if (in_bias) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (_Float16)fmaxf((float)v[e] + (float)pin[set][e], 0.f);
      }
      *reinterpret_cast<h16x8*>(&s_a[(sr + RS * i) * LDH + sc]) = v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<h16x8*>(&s_b[(sr + RS * i) * LDH + sc]) = pb[set][i];
  };

  f32x16 acc[2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

  auto multiply = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const h16x8 a = *reinterpret_cast<const h16x8*>(&s_a[(wave * 32 + r32) * LDH + 16 * ks + 8 * kb]);
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const h16x8 b = *reinterpret_cast<const h16x8*>(&s_b[(n * 32 + r32) * LDH + 16 * ks + 8 * kb]);
        acc[n] = simpb::mfma_32x32x16_f16(a, b, acc[n]);
      }
    }
  };

  // the residual pieces this thread will add in the epilogue are requested first, so that their round trip
  // is over by the time the (short: Cin / 64 chunks) K loop ends
  const bool full = c0 + BN <= Cout;
  h16x8 rres[4];
  if (residual && full) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * kThreads;
      int p = min(p0 + (idx >> 3), P_out - 1);
      if (res_up) p = up2(p, Ho, Wo);
      rres[i] = *reinterpret_cast<const h16x8*>(residual + (size_t)p * Cout + c0 + (idx & 7) * 8);
    }
  }

  fetch(std::integral_constant<int, 0>{}, 0);
  fetch(std::integral_constant<int, 1>{}, 1);
  for (int c = 0; c < nchunks; c += 2) {
    __syncthreads();
    stash(std::integral_constant<int, 0>{}, c);
    __syncthreads();
    fetch(std::integral_constant<int, 0>{}, c + 2);
    multiply();
    if (c + 1 < nchunks) {
      __syncthreads();
      stash(std::integral_constant<int, 1>{}, c + 1);
      __syncthreads();
      fetch(std::integral_constant<int, 1>{}, c + 3);
      multiply();
    }
  }

  // accumulators -> LDS (C/D layout: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5))
  __syncthreads();  // every wave is done reading the last staged chunk
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) s_c[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb) * LDC + n * 32 + r32] = acc[n][r];
  __syncthreads();
  // epilogue in 16-byte pieces: 128 rows x 8 pieces -> 4 per thread
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + i * kThreads;
    const int r = idx >> 3, c8 = (idx & 7) * 8;
    const int p = p0 + r;
    if (p >= P_out) continue;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = s_c[r * LDC + c8 + e];
    if (full) {
      const h16x8 bv = *reinterpret_cast<const h16x8*>(bias + c0 + c8);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += (float)bv[e];
      if (residual) {
        const h16x8 rv = rres[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
      }
      h16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (_Float16)(relu ? fmaxf(v[e], 0.f) : v[e]);
      *reinterpret_cast<h16x8*>(y + (size_t)p * Cout + c0 + c8) = o;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = c0 + c8 + e;
        if (c < Cout) {
          float t = v[e] + (float)bias[c];
          if (residual) t += (float)residual[(size_t)(res_up ? up2(p, Ho, Wo) : p) * Cout + c];
          y[(size_t)p * Cout + c] = (_Float16)(relu ? fmaxf(t, 0.f) : t);
        }
      }
    }
  }
}

}  // namespace

extern "C" int simpb_conv1x1_nhwc_f16(void* y, const void* x, const void* weight, const void* bias, const void* residual,
                                      int num_images, int in_h, int in_w, int in_channels, int out_channels, int stride,
                                      int relu, int residual_upsample2x, const void* input_bias, int variant, void* stream) {
  if (variant < 0 || variant > 3 || (input_bias && variant > 1)) return SIMPB_EINVAL;
  if (!y || !x || !weight || !bias || num_images <= 0 || in_h <= 0 || in_w <= 0 || in_channels <= 0 || out_channels <= 0 ||
      (stride != 1 && stride != 2) || in_channels % BK != 0 || out_channels % 8 != 0)  // BK = 64
    return SIMPB_EINVAL;
  if ((reinterpret_cast<size_t>(y) | reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(weight) |
       reinterpret_cast<size_t>(bias) | reinterpret_cast<size_t>(residual) | reinterpret_cast<size_t>(input_bias)) & 15)
    return SIMPB_EINVAL;
  const int ho = (in_h - 1) / stride + 1, wo = (in_w - 1) / stride + 1;
  if (residual_upsample2x && (!residual || (ho & 1) || (wo & 1))) return SIMPB_EINVAL;
  const long long p_out = (long long)num_images * ho * wo;
  if (p_out > (1ll << 30)) return SIMPB_EINVAL;
  (void)hipGetLastError();
  if (variant == 0) {
    // measured (tools/bench_conv1x1.py): the staged pipeline of csrc/conv3x3.hip (XCD-ordered tiles, double-buffered
    // stage, one barrier per chunk) wins where the K loop is long (8 chunks and more), this file's kernel on the short ones;
    // the input-side epilogue exists only here
    variant = (input_bias || in_channels < 512) ? 1 : 2;
  }
  if (variant > 1) {
    if ((long long)num_images * in_h * in_w * in_channels > (1ll << 31) - 1) return SIMPB_EINVAL;
    return simpb_conv_pointwise_staged(y, x, weight, bias, residual, (int)p_out, in_h, in_w, ho, wo, in_channels, out_channels,
                                       stride, relu, residual_upsample2x, variant - 2, stream);
  }
  dim3 grid((unsigned)((p_out + BM - 1) / BM), (out_channels + BN - 1) / BN);
  if (grid.y > 65535) return SIMPB_EINVAL;
  hipLaunchKernelGGL(conv1x1_f16_kernel, grid, dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                     static_cast<_Float16*>(y), static_cast<const _Float16*>(x), static_cast<const _Float16*>(weight),
                     static_cast<const _Float16*>(bias), static_cast<const _Float16*>(residual), (int)p_out, in_channels,
                     out_channels, relu, stride, ho, wo, in_h, in_w, residual_upsample2x ? 1 : 0,
                     static_cast<const _Float16*>(input_bias));
  return simpb_check_launch();
}
