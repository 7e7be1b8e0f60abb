// 3x3 convolution (padding 1, stride 1 or 2) of the fp16 channels_last backbone as an implicit GEMM with its epilogue in
// the launch:   y[n, ho, wo, :] = relu?( sum_{dy,dx} x[n, ho*s + dy - 1, wo*s + dx - 1, :] . W[:, dy, dx, :]^T + bias )
// These are conv2 of every ResNet bottleneck and the four output convolutions of the FPN
// (/root/reference/projects/configs/simpb_nus_r50_img_704x256.py:79-99: mmdet ResNet style="pytorch" + FPN), after conv-BN
// folding (tools/fuse_conv_bn.py:10-48). Round 1 left them to the vendor library. Round 2 found two reasons not to:
//  * MIOpen's fastest choice for several of these shapes is a composable-kernel XDL kernel built on the gfx950 double-K FP16
//    matrix instruction, which makes other kernels' vector arithmetic go wrong while it runs; with those
//    solvers off it falls back to split-K assembly kernels that need a zero-fill launch in front and a bias/ReLU pass behind;
//  * the FPN's output convolutions can write the decoder's fp32 token buffer themselves (`tokens` below), which removes the
//    separate format pass (46 MB read + 92 MB written per frame).
//
// GEMM view: M = output pixels, N = output channels, K = 9 taps x Cin walked tap by tap in chunks of 64 input channels.
// The MFMA contraction does not care WHICH eight k-values a lane brings as long as both operands agree, so lane
// (row r32, half kb) of a wave takes channels [32*kb, 32*kb + 32) of the chunk: its four 16-byte loads are 64 contiguous
// bytes of ITS pixel's row, and the A operand goes global -> registers directly -- no staging of the (nine times re-read)
// activations through LDS, no barrier for them; a tap outside the image is read from the pixel's own centre (always in
// bounds) and zeroed when it is consumed. Weights as PyTorch keeps a channels_last convolution weight: [Cout][3][3][Cin], i.e.
// row-major [Cout][K] in exactly this chunk order. Two ways to share the work among the 4 waves of a workgroup:
//  * MSPLIT (large maps): waves tile M (32*AF pixels x 64 channels each); the 64 x 64 weight chunk is shared through a
//    double-buffered LDS stage (one barrier per chunk), three register sets of A in flight;
//  * KSPLIT (small maps, long K: stages 3-4, coarse FPN levels): all four waves own the SAME 32*AF x 64 tile and take every
//    fourth chunk; both operands straight from global memory, no LDS and no barrier in the loop, one LDS meeting at the end.
//    M = 1 056 pixels (8 x 22 x 6 cameras) still yields 264 workgroups.
// FP16 matrix step = two v_mfma_f32_32x32x8f16 (csrc/mfma_f16.h; never the gfx950 double-K instruction).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <type_traits>
#include "../../include/simpb_hip.h"
#include "mfma_f16.h"

extern "C" int simpb_check_launch(void);

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using h16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int BN = 64, BK = 64;
constexpr int LDH = BK + 8;   // halfs per staged weight row (144 B): conflict-free 16-lane groups for ds_read_b128
constexpr int LDC = BN + 1;   // floats per row of the epilogue tile
constexpr int kThreads = 256;
constexpr int kStageBytes = 2 * BN * LDH * 2;        // two weight chunks
constexpr int kTileBytes = 4 * 32 * LDC * 4;         // one 32 x 64 fp32 tile per wave
constexpr int kSmemBytes = kStageBytes > kTileBytes ? kStageBytes : kTileBytes;

struct ConvArgs {
  _Float16* y;            // f16 [P_out, Cout] or NULL
  float* tok;             // f32 token buffer or NULL (ops/__init__.py:63-92 layout: [bs, cams * tokens_per_cam, Cout])
  _Float16* tok16;        // the same rows in f16 (for value_proj's two-pass product) or NULL
  const _Float16 *x, *w, *bias;
  const _Float16* residual;   // f16 like y (or half-size with res_up: read with nearest 2x upsampling), or NULL
  int res_up;
  int P_out, Cin, Cout, relu, stride, Ho, Wo, H, W;
  int tokens_per_cam, level_start;
  int gx, gy, per_xcd;    // tile grid and tiles per XCD range
};

template <int S>
using IC = std::integral_constant<int, S>;

// one 16-byte piece of an output row: bias, ReLU, one rounding to fp16; to the f16 map or, widened, to the token buffer
__device__ __forceinline__ void emit_piece(const ConvArgs& a, int c0, bool full, int p, int c8, const float (&v)[8]) {
  if (p >= a.P_out) return;
  size_t trow = 0;
  if (a.tok) {   // the FPN's output convolution writes the decoder's token row itself: fp32 values of the fp16 result
    const int hw = a.Ho * a.Wo;
    const int n = p / hw, pix = p - n * hw;
    trow = ((size_t)n * a.tokens_per_cam + a.level_start + pix) * a.Cout;
  }
  size_t rrow = 0;
  if (a.residual) {
    int rp = p;
    if (a.res_up) {   // F.interpolate(mode="nearest") of an exact 2x: source = floor(dst / 2) (the FPN top-down path)
      const int hw = a.Ho * a.Wo;
      const int n = p / hw, rem = p - n * hw;
      const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
      rp = (n * (a.Ho >> 1) + (ho >> 1)) * (a.Wo >> 1) + (wo >> 1);
    }
    rrow = (size_t)rp * a.Cout;
  }
  if (full) {
    const h16x8 bv = *reinterpret_cast<const h16x8*>(a.bias + c0 + c8);
    h16x8 rv = {0, 0, 0, 0, 0, 0, 0, 0};
    if (a.residual) rv = *reinterpret_cast<const h16x8*>(a.residual + rrow + c0 + c8);
    h16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float t = v[e] + (float)bv[e] + (float)rv[e];
      o[e] = (_Float16)(a.relu ? fmaxf(t, 0.f) : t);
    }
    if (a.tok) {
      float* d = a.tok + trow + c0 + c8;
      *reinterpret_cast<float4*>(d) = make_float4((float)o[0], (float)o[1], (float)o[2], (float)o[3]);
      *reinterpret_cast<float4*>(d + 4) = make_float4((float)o[4], (float)o[5], (float)o[6], (float)o[7]);
      if (a.tok16) *reinterpret_cast<h16x8*>(a.tok16 + trow + c0 + c8) = o;
    } else {
      *reinterpret_cast<h16x8*>(a.y + (size_t)p * a.Cout + c0 + c8) = o;
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c0 + c8 + e;
      if (c < a.Cout) {
        float t = v[e] + (float)a.bias[c];
        if (a.residual) t += (float)a.residual[rrow + c];
        const _Float16 o = (_Float16)(a.relu ? fmaxf(t, 0.f) : t);
        if (a.tok) {
          a.tok[trow + c] = (float)o;
          if (a.tok16) a.tok16[trow + c] = o;
        } else {
          a.y[(size_t)p * a.Cout + c] = o;
        }
      }
    }
  }
}

template <int AF, bool KSPLIT>
__global__ __launch_bounds__(kThreads, 2) void conv3x3_f16_kernel(const ConvArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[kSmemBytes];
  _Float16* s_b = reinterpret_cast<_Float16*>(smem);
  float* s_c = reinterpret_cast<float*>(smem);
  constexpr int WROWS = 32 * AF;
  constexpr int BMt = KSPLIT ? WROWS : 4 * WROWS;

  // each XCD (blockIdx % 8) walks a contiguous range of tiles, channel blocks of one pixel block next to each other:
  // neighbouring tiles share their activation rows in that XCD's L2
  const int tile = (blockIdx.x & 7) * a.per_xcd + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= a.per_xcd || tile >= a.gx * a.gy) return;
  const int tx = tile / a.gy, ty = tile - tx * a.gy;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r32 = lane & 31, kb = lane >> 5;
  const int p0 = tx * BMt, c0 = ty * BN;
  const int row0 = KSPLIT ? 0 : wave * WROWS;
  const int Cin = a.Cin, W = a.W, H = a.H;
  const int per_tap = Cin / BK, nchunks = 9 * per_tap;
  const size_t K = (size_t)9 * Cin;
  const _Float16* __restrict__ wgt = a.w;

  // A: this lane's pixel per fragment: centre address (+ its half of the chunk) and the 9-bit in-image mask of its taps
  const _Float16* actr[AF];
  unsigned amask[AF];
#pragma unroll
  for (int f = 0; f < AF; ++f) {
    const int p = min(p0 + row0 + 32 * f + r32, a.P_out - 1);
    const int hw = a.Ho * a.Wo;
    const int n = p / hw, rem = p - n * hw;
    const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
    const int hc = ho * a.stride, wc = wo * a.stride;
    actr[f] = a.x + ((size_t)(n * H + hc) * W + wc) * Cin + 32 * kb;
    unsigned m = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int hi = hc + t / 3 - 1, wi = wc + t % 3 - 1;
      m |= (hi >= 0 && hi < H && wi >= 0 && wi < W) ? (1u << t) : 0u;
    }
    amask[f] = m;
  }
  // B rows: MSPLIT stages 64 rows x 8 pieces with 256 threads (2 pieces each); KSPLIT loads its fragments directly
  const int sr = tid >> 3, sc = (tid & 7) * 8;
  size_t brow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
    brow[i] = KSPLIT ? (size_t)min(c0 + 32 * i + r32, a.Cout - 1) * K + 32 * kb : (size_t)min(c0 + sr + 32 * i, a.Cout - 1) * K + sc;

  h16x8 ar[3][AF][4];
  h16x8 bp[3][2];        // MSPLIT: staged pieces
  h16x8 bq[3][2][4];     // KSPLIT: fragments
  unsigned alive[3];     // bit f: fragment f of the set is inside the image (and the chunk exists)

  auto load_a = [&](auto set_c, int chunk) __attribute__((always_inline)) {
    constexpr int s = decltype(set_c)::value;
    const bool exists = chunk < nchunks;
    const int c = min(chunk, nchunks - 1);
    const int tap = c / per_tap, k0 = (c - tap * per_tap) * BK;
    const int dy = tap / 3, dx = tap - dy * 3;
    const int toff = ((dy - 1) * W + (dx - 1)) * Cin;
    unsigned m = 0;
#pragma unroll
    for (int f = 0; f < AF; ++f) {
      const bool in = ((amask[f] >> tap) & 1u) != 0 && exists;
      m |= (in ? 1u : 0u) << f;
      const _Float16* src = actr[f] + (in ? toff : 0) + k0;
#pragma unroll
      for (int j = 0; j < 4; ++j) ar[s][f][j] = *reinterpret_cast<const h16x8*>(src + 8 * j);
    }
    alive[s] = m;
  };
  auto load_b = [&](auto set_c, int chunk) __attribute__((always_inline)) {
    constexpr int s = decltype(set_c)::value;
    const int c = min(chunk, nchunks - 1);
    const int tap = c / per_tap, k0 = (c - tap * per_tap) * BK;
    const size_t koff = (size_t)tap * Cin + k0;
    if constexpr (KSPLIT) {
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int j = 0; j < 4; ++j) bq[s][n][j] = *reinterpret_cast<const h16x8*>(wgt + brow[n] + koff + 8 * j);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) bp[s][i] = *reinterpret_cast<const h16x8*>(wgt + brow[i] + koff);
    }
  };
  auto stash_b = [&](auto set_c, int buf) __attribute__((always_inline)) {
    constexpr int s = decltype(set_c)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<h16x8*>(&s_b[(buf * BN + sr + 32 * i) * LDH + sc]) = bp[s][i];
  };

  f32x16 acc[AF][2];
#pragma unroll
  for (int f = 0; f < AF; ++f)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[f][n][r] = 0.f;

  auto multiply = [&](auto set_c, int buf) __attribute__((always_inline)) {
    constexpr int s = decltype(set_c)::value;
    const h16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      h16x8 b[2];
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        if constexpr (KSPLIT) b[n] = bq[s][n][j];
        else b[n] = *reinterpret_cast<const h16x8*>(&s_b[(buf * BN + n * 32 + r32) * LDH + 32 * kb + 8 * j]);
      }
#pragma unroll
      for (int f = 0; f < AF; ++f) {
        const h16x8 av = ((alive[s] >> f) & 1u) ? ar[s][f][j] : zero;
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[f][n] = simpb::mfma_32x32x16_f16(av, b[n], acc[f][n]);
      }
    }
  };

  // the loads are pinned where they are written (sched_barrier): left to itself the scheduler sinks them to the end of the
  // matrix work in front of their use, and the wait for them then drains the queue
  if constexpr (KSPLIT) {
    // wave w takes chunks w, w + 4, ...; three register sets in flight; chunks past the end are loaded from the last one
    // and contribute zeros (alive = 0)
    const int mine_n = (nchunks + 3) / 4;
    if constexpr (AF == 1) {
      const int iters = (mine_n + 2) / 3 * 3;
      load_a(IC<0>{}, wave);
      load_b(IC<0>{}, wave);
      load_a(IC<1>{}, wave + 4);
      load_b(IC<1>{}, wave + 4);
      auto step = [&](auto cur, auto nxt2, int i) __attribute__((always_inline)) {
        load_a(nxt2, wave + 4 * (i + 2));
        load_b(nxt2, wave + 4 * (i + 2));
        __builtin_amdgcn_sched_barrier(0);
        multiply(cur, 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      for (int i = 0; i < iters; i += 3) {
        step(IC<0>{}, IC<2>{}, i);
        step(IC<1>{}, IC<0>{}, i + 1);
        step(IC<2>{}, IC<1>{}, i + 2);
      }
    } else {
      // 64 x 64 tile per wave: two register sets (a chunk is 16 matrix steps of cover)
      const int iters = (mine_n + 1) / 2 * 2;
      load_a(IC<0>{}, wave);
      load_b(IC<0>{}, wave);
      auto step = [&](auto cur, auto nxt, int i) __attribute__((always_inline)) {
        load_a(nxt, wave + 4 * (i + 1));
        load_b(nxt, wave + 4 * (i + 1));
        __builtin_amdgcn_sched_barrier(0);
        multiply(cur, 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      for (int i = 0; i < iters; i += 2) {
        step(IC<0>{}, IC<1>{}, i);
        step(IC<1>{}, IC<0>{}, i + 1);
      }
    }
  } else {
    // chunk c: A(c) in register set c % 3 (requested two chunks ago); B(c) in LDS buffer c & 1, written during chunk c - 1
    // from register set c % 3 (requested during chunk c - 3)
    load_b(IC<0>{}, 0);
    load_b(IC<1>{}, 1);
    load_b(IC<2>{}, 2);
    load_a(IC<0>{}, 0);
    load_a(IC<1>{}, 1);
    __builtin_amdgcn_sched_barrier(0);
    stash_b(IC<0>{}, 0);
    __syncthreads();
    auto step = [&](auto cur, auto nxt, auto nxt2, int c) __attribute__((always_inline)) {
      load_b(cur, c + 3);    // set `cur` was written to LDS during chunk c - 1
      load_a(nxt2, c + 2);
      __builtin_amdgcn_sched_barrier(0);
      stash_b(nxt, (c + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      multiply(cur, c & 1);
      __syncthreads();
    };
    for (int c = 0; c < nchunks; c += 3) {   // nchunks = 9 * per_tap: a multiple of 3
      step(IC<0>{}, IC<1>{}, IC<2>{}, c);
      step(IC<1>{}, IC<2>{}, IC<0>{}, c + 1);
      step(IC<2>{}, IC<0>{}, IC<1>{}, c + 2);
    }
  }

  // epilogue, one A fragment at a time: accumulators -> LDS (C/D layout: column = lane & 31,
  // row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)), then 16-byte pieces of full rows: bias, ReLU, one rounding to fp16
  const bool full = c0 + BN <= a.Cout;
  auto emit = [&](int p, int c8, const float (&v)[8]) __attribute__((always_inline)) { emit_piece(a, c0, full, p, c8, v); };
  float* mine = s_c + wave * 32 * LDC;
#pragma unroll
  for (int f = 0; f < AF; ++f) {
    __syncthreads();   // the weight stage (or the previous fragment's tile) is no longer read
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) mine[((r & 3) + 8 * (r >> 2) + 4 * kb) * LDC + n * 32 + r32] = acc[f][n][r];
    __syncthreads();
    if constexpr (KSPLIT) {
      const int r = tid >> 3, c8 = (tid & 7) * 8;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float s = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) s += s_c[(w4 * 32 + r) * LDC + c8 + e];   // fixed order: deterministic
        v[e] = s;
      }
      emit(p0 + 32 * f + r, c8, v);
    } else {
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int r = (lane >> 3) + 8 * pass, c8 = (lane & 7) * 8;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = mine[r * LDC + c8 + e];
        emit(p0 + row0 + 32 * f + r, c8, v);
      }
    }
  }
}

template <int AF, bool KSPLIT>
void launch(ConvArgs& a, hipStream_t stream) {
  constexpr int BMt = KSPLIT ? 32 * AF : 128 * AF;
  a.gx = (a.P_out + BMt - 1) / BMt;
  a.gy = (a.Cout + BN - 1) / BN;
  const long long total = (long long)a.gx * a.gy;
  a.per_xcd = (int)((total + 7) / 8);
  hipLaunchKernelGGL((conv3x3_f16_kernel<AF, KSPLIT>), dim3((unsigned)(a.per_xcd * 8)), dim3(kThreads), 0, stream, a);
}


// ---- large maps: both operands staged through LDS with coalesced loads ------------------------------------------------------
// Reading A straight into the MFMA layout costs a wave-instruction 32 different 128-byte lines (one per pixel row, two lanes
// each), and the texture path takes them a quad of lanes at a time: the direct kernels above top out near 430 TFLOP/s on the
// 64 x 176 maps with the matrix cores half idle. Here 8 consecutive lanes load the 8 pieces of one pixel row (a
// wave-instruction = 8 whole lines), the chunk goes through a double-buffered LDS stage (one barrier per chunk, three register
// sets: loads run two chunks ahead), and fragments are read back with ds_read_b128 (256 B/clk per CU on gfx950, conflict-free
// at a 144-byte row pitch). BMt x BNt output tile, WGM x WGN waves, each (BMt / WGM) x (BNt / WGN). The tile is chosen per shape so
// that the grid fills whole rounds of the chip: 67 584 pixels are 528 tiles of 128 (two per CU and sixteen left over) but 704 of 96.
template <int BMt, int BNt, int WGM, int WGN, int TAPS>
__device__ __forceinline__ void conv_staged_tile(const ConvArgs& a, const int tile) {
  constexpr int NT = 64 * WGM * WGN;
  constexpr int AF = BMt / WGM / 32, NF = BNt / WGN / 32;
  static_assert(AF * WGM * 32 == BMt && NF * WGN * 32 == BNt && NF <= 2, "tile = waves x 32-row / 32-column fragments");
  constexpr int NA = (BMt * 8 + NT - 1) / NT, NB = (BNt * 8 + NT - 1) / NT;   // staged 16-byte pieces per thread and chunk
  constexpr int kRows = BMt + BNt;
  constexpr int LDW = 32 * NF + 1;                                             // floats per row of a wave's epilogue tile
  constexpr int kStage = 2 * kRows * LDH * 2, kTile = WGM * WGN * 32 * LDW * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[kStage > kTile ? kStage : kTile];
  _Float16* s_ab = reinterpret_cast<_Float16*>(smem);
  float* s_c = reinterpret_cast<float*>(smem);

  const int tx = tile / a.gy, ty = tile - tx * a.gy;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r32 = lane & 31, kb = lane >> 5;
  const int wm = wave / WGN, wn = wave - wm * WGN;
  const int p0 = tx * BMt, c0 = ty * BNt;
  const int Cin = a.Cin, W = a.W, H = a.H;
  const int per_tap = Cin / BK, nchunks = TAPS * per_tap;
  const size_t K = (size_t)TAPS * Cin;
  const _Float16* __restrict__ wgt = a.w;

  // piece i of this thread: index tid + NT * i -> (row, 8-half column) of the A or B part of the stage
  const int sc = (tid & 7) * 8;
  const _Float16* actr[NA];   // centre of this thread's staged pixel rows (+ its piece)
  unsigned amask = 0;         // 6 bits per row: which of the 3 input rows / 3 input columns around it are inside the image
  static_assert(NA <= 5, "6 mask bits per staged row in one register");
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int p = min(p0 + ((tid + NT * i) >> 3), a.P_out - 1);
    const int hw = a.Ho * a.Wo;
    const int n = p / hw, rem = p - n * hw;
    const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
    const int hc = ho * a.stride, wc = wo * a.stride;
    actr[i] = a.x + ((size_t)(n * H + hc) * W + wc) * Cin + sc;
    const unsigned rows = (hc > 0 ? 1u : 0u) | 2u | (hc + 1 < H ? 4u : 0u), cols = (wc > 0 ? 1u : 0u) | 2u | (wc + 1 < W ? 4u : 0u);
    amask |= (rows | (cols << 3)) << (6 * i);
  }
  size_t brow[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) brow[i] = (size_t)min(c0 + ((tid + NT * i) >> 3), a.Cout - 1) * K + sc;

  h16x8 ra[3][NA], rb[3][NB];
  unsigned alive[3];

  auto fetch = [&](auto set_c, int chunk) __attribute__((always_inline)) {
    constexpr int s = decltype(set_c)::value;
    const int c = min(chunk, nchunks - 1);
    if constexpr (TAPS == 1) {   // a 1x1 convolution: the centre pixel only, always inside the image
      const int k0 = c * BK;
#pragma unroll
      for (int i = 0; i < NA; ++i) ra[s][i] = *reinterpret_cast<const h16x8*>(actr[i] + k0);
      alive[s] = ~0u;
#pragma unroll
      for (int i = 0; i < NB; ++i) rb[s][i] = *reinterpret_cast<const h16x8*>(wgt + brow[i] + k0);
    } else {
      const int tap = c / per_tap, k0 = (c - tap * per_tap) * BK;
      const int dy = tap / 3, dx = tap - dy * 3;
      const int toff = ((dy - 1) * W + (dx - 1)) * Cin;
      unsigned m = 0;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const bool in = ((amask >> (6 * i + dy)) & (amask >> (6 * i + 3 + dx)) & 1u) != 0;
        m |= (in ? 1u : 0u) << i;
        ra[s][i] = *reinterpret_cast<const h16x8*>(actr[i] + (in ? toff : 0) + k0);
      }
      alive[s] = m;
      const size_t koff = (size_t)tap * Cin + k0;
#pragma unroll
      for (int i = 0; i < NB; ++i) rb[s][i] = *reinterpret_cast<const h16x8*>(wgt + brow[i] + koff);
    }
  };
  auto stash = [&](auto set_c, int buf) __attribute__((always_inline)) {
    constexpr int s = decltype(set_c)::value;
    const h16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    _Float16* base = s_ab + buf * kRows * LDH;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int idx = tid + NT * i;
      if ((BMt * 8) % NT == 0 || idx < BMt * 8)
        *reinterpret_cast<h16x8*>(&base[(idx >> 3) * LDH + sc]) = ((alive[s] >> i) & 1u) ? ra[s][i] : zero;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int idx = tid + NT * i;
      if ((BNt * 8) % NT == 0 || idx < BNt * 8) *reinterpret_cast<h16x8*>(&base[(BMt + (idx >> 3)) * LDH + sc]) = rb[s][i];
    }
  };

  f32x16 acc[AF][NF];
#pragma unroll
  for (int f = 0; f < AF; ++f)
#pragma unroll
    for (int n = 0; n < NF; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[f][n][r] = 0.f;

  auto multiply = [&](int buf) __attribute__((always_inline)) {
    const _Float16* sa = s_ab + buf * kRows * LDH + (wm * AF * 32 + r32) * LDH + 8 * kb;
    const _Float16* sb = s_ab + buf * kRows * LDH + (BMt + wn * NF * 32 + r32) * LDH + 8 * kb;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      h16x8 av[AF], bv[NF];
#pragma unroll
      for (int f = 0; f < AF; ++f) av[f] = *reinterpret_cast<const h16x8*>(sa + f * 32 * LDH + 16 * ks);
#pragma unroll
      for (int n = 0; n < NF; ++n) bv[n] = *reinterpret_cast<const h16x8*>(sb + n * 32 * LDH + 16 * ks);
#pragma unroll
      for (int f = 0; f < AF; ++f)
#pragma unroll
        for (int n = 0; n < NF; ++n) acc[f][n] = simpb::mfma_32x32x16_f16(av[f], bv[n], acc[f][n]);
    }
  };

  // chunk c: LDS buffer c & 1; register set (c + 1) % 3 holds chunk c + 1 (requested during chunk c - 2)
  fetch(IC<0>{}, 0);
  fetch(IC<1>{}, 1);
  fetch(IC<2>{}, 2);
  __builtin_amdgcn_sched_barrier(0);
  stash(IC<0>{}, 0);
  __syncthreads();
  auto step = [&](auto cur, auto nxt, int c) __attribute__((always_inline)) {
    fetch(cur, c + 3);      // set `cur` went to LDS during chunk c - 1
    __builtin_amdgcn_sched_barrier(0);
    stash(nxt, (c + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
    multiply(c & 1);
    __syncthreads();
  };
  for (int c = 0; c < nchunks; c += 3) {   // 9 * per_tap is a multiple of 3; a 1x1 convolution has any count
    step(IC<0>{}, IC<1>{}, c);
    if (TAPS == 9 || c + 1 < nchunks) step(IC<1>{}, IC<2>{}, c + 1);
    if (TAPS == 9 || c + 2 < nchunks) step(IC<2>{}, IC<0>{}, c + 2);
  }

  // epilogue, one A fragment at a time through the wave's own 32 x (32 * NF) fp32 tile: 16-byte pieces of output rows
  float* mine = s_c + wave * 32 * LDW;
  const int cw = c0 + wn * NF * 32;
  const bool full = cw + 32 * NF <= a.Cout;
#pragma unroll
  for (int f = 0; f < AF; ++f) {
    if (f) __syncthreads();
#pragma unroll
    for (int n = 0; n < NF; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) mine[((r & 3) + 8 * (r >> 2) + 4 * kb) * LDW + n * 32 + r32] = acc[f][n][r];
    __syncthreads();
    constexpr int PR = 4 * NF;            // pieces per row
#pragma unroll
    for (int pass = 0; pass < 32 * PR / 64; ++pass) {
      const int idx = lane + 64 * pass;
      const int r = idx / PR, c8 = (idx % PR) * 8;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = mine[r * LDW + c8 + e];
      emit_piece(a, cw, full, p0 + (wm * AF + f) * 32 + r, c8, v);
    }
  }
}

template <int BMt, int BNt, int WGM, int WGN, int TAPS>
__global__ __launch_bounds__(64 * WGM * WGN, WGM * WGN == 3 ? 3 : 2) void conv_staged_kernel(const ConvArgs a) {
  // each XCD (blockIdx % 8) walks a contiguous range of tiles
  const int tile = (blockIdx.x & 7) * a.per_xcd + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= a.per_xcd || tile >= a.gx * a.gy) return;
  conv_staged_tile<BMt, BNt, WGM, WGN, TAPS>(a, tile);
}

// Several convolutions of one tile shape in ONE launch (the FPN's four output convolutions: 240 us as four launches, of which
// the three small levels -- 352, 88 and 22 tiles -- leave most of the chip idle for 97 us; in one launch their tiles fill
// the last round of the large level's 1 408). Problem j owns the tiles [start[j], start[j + 1]) of one XCD-contiguous walk.
constexpr int kMaxConvGroup = 4;
struct ConvGroup {
  ConvArgs a[kMaxConvGroup];
  int start[kMaxConvGroup + 1];
  int n, per_xcd;
};
template <int BMt, int BNt, int WGM, int WGN, int TAPS>
__global__ __launch_bounds__(64 * WGM * WGN, WGM * WGN == 3 ? 3 : 2) void conv_staged_group_kernel(const ConvGroup g) {
  const int gt = (blockIdx.x & 7) * g.per_xcd + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= g.per_xcd || gt >= g.start[g.n]) return;
  int j = 0;
#pragma unroll
  for (int t = 1; t < kMaxConvGroup; ++t)
    if (t < g.n && gt >= g.start[t]) j = t;
  conv_staged_tile<BMt, BNt, WGM, WGN, TAPS>(g.a[j], gt - g.start[j]);
}

template <int BMt, int BNt, int WGM, int WGN, int TAPS>
void launch_staged(ConvArgs& a, hipStream_t stream) {
  a.gx = (a.P_out + BMt - 1) / BMt;
  a.gy = (a.Cout + BNt - 1) / BNt;
  const long long total = (long long)a.gx * a.gy;
  a.per_xcd = (int)((total + 7) / 8);
  hipLaunchKernelGGL((conv_staged_kernel<BMt, BNt, WGM, WGN, TAPS>), dim3((unsigned)(a.per_xcd * 8)), dim3(64 * WGM * WGN), 0,
                     stream, a);
}

}  // namespace

extern "C" int simpb_conv3x3_nhwc_f16(void* y, float* tokens, void* tokens_f16, int tokens_per_cam, int level_start,
                                      const void* x, const void* weight, const void* bias, int num_images, int in_h, int in_w,
                                      int in_channels, int out_channels, int stride, int relu, int variant, void* stream) {
  if ((!y && !tokens) || (y && tokens) || (tokens_f16 && !tokens) || (reinterpret_cast<size_t>(tokens_f16) & 15) || !x || !weight || !bias || num_images <= 0 || in_h <= 0 || in_w <= 0 ||
      in_channels <= 0 || out_channels <= 0 || (stride != 1 && stride != 2) || in_channels % BK != 0 || out_channels % 8 != 0 ||
      variant < 0 || variant > 8)
    return SIMPB_EINVAL;
  if ((reinterpret_cast<size_t>(y) | reinterpret_cast<size_t>(tokens) | reinterpret_cast<size_t>(x) |
       reinterpret_cast<size_t>(weight) | reinterpret_cast<size_t>(bias)) & 15)
    return SIMPB_EINVAL;
  const int ho = (in_h - 1) / stride + 1, wo = (in_w - 1) / stride + 1;   // padding 1, kernel 3
  const long long p_out = (long long)num_images * ho * wo;
  const long long in_elems = (long long)num_images * in_h * in_w * in_channels;
  if (p_out > (1ll << 30) || in_elems > (1ll << 31) - 1) return SIMPB_EINVAL;   // tap offsets are 32-bit
  if (tokens && (tokens_per_cam < ho * wo || level_start < 0 || level_start + ho * wo > tokens_per_cam)) return SIMPB_EINVAL;
  (void)hipGetLastError();
  ConvArgs a{static_cast<_Float16*>(y), tokens, static_cast<_Float16*>(tokens_f16), static_cast<const _Float16*>(x), static_cast<const _Float16*>(weight),
             static_cast<const _Float16*>(bias), nullptr, 0, (int)p_out, in_channels, out_channels, relu, stride, ho, wo, in_h, in_w,
             tokens_per_cam, level_start, 0, 0, 0};
  const long long ny = (out_channels + BN - 1) / BN;
  if (variant == 0) {
    // measured on the ResNet50 / FPN shapes at 6 x 256 x 704 (tools/bench_conv3x3.py): the LDS-staged 96-row tilings while
    // they fill the chip (96 divides the pixel counts of a 6-camera rig and leaves no nearly empty last round of
    // workgroups), then the direct ones, K split inside the workgroup for the smallest maps
    const long long t96 = (p_out + 95) / 96, t128 = (p_out + 127) / 128, t64 = (p_out + 63) / 64;
    if (out_channels >= 128 && t128 * ((out_channels + 127) / 128) >= 2048) variant = 6;   // many rounds: the larger tile wins
    else if (out_channels >= 128 && t96 * ((out_channels + 127) / 128) >= 160) variant = 8;
    else if (t96 * ny >= 256) variant = 7;
    else if (t128 * ny >= 128) variant = 1;
    else if (t64 * ny >= 128) variant = 4;
    else variant = 3;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (variant) {
    case 1: launch<1, false>(a, s); break;   // 128 pixels x 64 channels, weights through LDS
    case 2: launch<2, false>(a, s); break;   // 256 x 64
    case 3: launch<1, true>(a, s); break;    // 32 x 64, K split over the waves
    case 4: launch<2, true>(a, s); break;    // 64 x 64, K split over the waves
    case 5: launch_staged<128, 64, 4, 1, 9>(a, s); break;    // 128 x 64, operands staged through LDS
    case 6: launch_staged<128, 128, 2, 2, 9>(a, s); break;  // 128 x 128
    case 7: launch_staged<96, 64, 3, 1, 9>(a, s); break;     // 96 x 64, three waves: 704 workgroups for 67 584 pixels
    default: launch_staged<96, 128, 1, 4, 9>(a, s); break;   // 96 x 128, waves side by side along the channels
  }
  return simpb_check_launch();
}

// The FPN's output convolutions (3x3, stride 1, Cin -> Cout, bias, no ReLU; mmdet FPN `fpn_convs[i].conv` after
// tools/fuse_conv_bn.py:10-48) of up to four levels in ONE launch, each writing its level's token rows (f32 and, optionally,
// f16) as simpb_conv3x3_nhwc_f16 does with `tokens`: x[j] f16 NHWC [num_images, in_h[j], in_w[j], Cin], weight[j] f16
// [Cout, 3, 3, Cin], bias[j] f16 [Cout], level_start[j] the level's first row inside a camera's tokens_per_cam rows.
extern "C" int simpb_conv3x3_group_tokens_f16(int num_levels, float* tokens, void* tokens_f16, int tokens_per_cam,
                                              const int* level_start, const void* const* x, const void* const* weight,
                                              const void* const* bias, int num_images, const int* in_h, const int* in_w,
                                              int in_channels, int out_channels, int relu, void* stream) {
  if (num_levels < 1 || num_levels > kMaxConvGroup || !tokens || !level_start || !x || !weight || !bias || !in_h || !in_w ||
      num_images <= 0 || in_channels <= 0 || out_channels <= 0 || in_channels % BK != 0 || out_channels % 8 != 0 ||
      ((reinterpret_cast<size_t>(tokens) | reinterpret_cast<size_t>(tokens_f16)) & 15))
    return SIMPB_EINVAL;
  ConvGroup g{};
  long long total = 0;
  for (int j = 0; j < num_levels; ++j) {
    if (!x[j] || !weight[j] || !bias[j] || in_h[j] <= 0 || in_w[j] <= 0 ||
        ((reinterpret_cast<size_t>(x[j]) | reinterpret_cast<size_t>(weight[j]) | reinterpret_cast<size_t>(bias[j])) & 15))
      return SIMPB_EINVAL;
    const long long p_out = (long long)num_images * in_h[j] * in_w[j];
    if (p_out > (1ll << 30) || p_out * in_channels > (1ll << 31) - 1) return SIMPB_EINVAL;   // tap offsets are 32-bit
    if (tokens_per_cam < in_h[j] * in_w[j] || level_start[j] < 0 || level_start[j] + in_h[j] * in_w[j] > tokens_per_cam)
      return SIMPB_EINVAL;
    g.a[j] = ConvArgs{nullptr, tokens, static_cast<_Float16*>(tokens_f16), static_cast<const _Float16*>(x[j]),
                      static_cast<const _Float16*>(weight[j]), static_cast<const _Float16*>(bias[j]), nullptr, 0, (int)p_out,
                      in_channels, out_channels, relu, 1, in_h[j], in_w[j], in_h[j], in_w[j], tokens_per_cam, level_start[j],
                      0, 0, 0};
    g.a[j].gx = (int)((p_out + 95) / 96);
    g.a[j].gy = (out_channels + 127) / 128;
    g.start[j] = (int)total;
    total += (long long)g.a[j].gx * g.a[j].gy;
  }
  if (total > (1 << 24)) return SIMPB_EINVAL;
  for (int j = num_levels; j <= kMaxConvGroup; ++j) g.start[j] = (int)total;
  g.n = num_levels;
  g.per_xcd = (int)((total + 7) / 8);
  (void)hipGetLastError();
  hipLaunchKernelGGL((conv_staged_group_kernel<96, 128, 1, 4, 9>), dim3((unsigned)(g.per_xcd * 8)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), g);
  return simpb_check_launch();
}

// 1x1 convolutions through the same staged pipeline (TAPS = 1): csrc/conv1x1.hip validates and forwards here.
// tiling 0: 128 x 64 per workgroup, 1: 128 x 128.
extern "C" int simpb_conv_pointwise_staged(void* y, const void* x, const void* weight, const void* bias, const void* residual,
                                           int p_out, int in_h, int in_w, int ho, int wo, int in_channels, int out_channels,
                                           int stride, int relu, int residual_upsample2x, int tiling, void* stream) {
  ConvArgs a{static_cast<_Float16*>(y), nullptr, nullptr, static_cast<const _Float16*>(x), static_cast<const _Float16*>(weight),
             static_cast<const _Float16*>(bias), static_cast<const _Float16*>(residual), residual_upsample2x ? 1 : 0,
             p_out, in_channels, out_channels, relu, stride, ho, wo, in_h, in_w, 0, 0, 0, 0, 0};
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (tiling == 0) launch_staged<128, 64, 4, 1, 1>(a, s);
  else launch_staged<128, 128, 2, 2, 1>(a, s);
  return simpb_check_launch();
}
