// Fused small-MLP chains (gfx950): a whole `linear_relu_ln` stack of the reference
// (/root/reference/projects/mmdet3d_plugin/models/blocks.py:32-43) -- [Linear, ReLU]*, LayerNorm,
// ..., optional last Linear + Scale -- in ONE launch, with the activations of a row tile living in
// LDS between layers. Used for the 3D anchor encoder (detection3d/blocks.py:57-74: 4 branches x 4 x
// [Linear, ReLU, LN] = 48 tiny kernels in the reference, 7 times per frame), the 2D sine encoder
// (detection2d/blocks.py:48-63 + utils.py:40-63), the refinement heads (detection3d/blocks.py:123-154,
// detection2d/blocks.py:117-144) and the camera encoder (blocks.py:93-99).
//
// These chains are tiny (<= 1.5k rows x 256 wide): the cost in the reference is launch count, not
// FLOPs, so the kernel is organised for latency, not for the matrix cores: one workgroup = R rows,
// thread t = output column t, weights are read pre-transposed ([in][out], one coalesced 1-KiB
// wave-load per k, L2-resident and shared by all workgroups), the row tile is broadcast from LDS.
// Up to 4 independent chains (e.g. the 4 encoder branches) share a launch through blockIdx.y.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "../../include/simpb_hip.h"

extern "C" int simpb_check_launch(void);

#ifdef SIMPB_CHAIN_STAMPS
// diagnostic build only (tools/chain_stamps.py): s_memtime of workgroup (0, 0), wave 0 at phase boundaries
__device__ unsigned long long g_chain_stamps[128];
#define SIMPB_STAMP(i)                                                                       \
  do {                                                                                       \
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && (i) < 128)                 \
      g_chain_stamps[(i)] = __builtin_amdgcn_s_memtime();                                    \
  } while (0)
extern "C" int simpb_debug_chain_stamps(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_chain_stamps), sizeof(g_chain_stamps));
}
#else
#define SIMPB_STAMP(i) do {} while (0)
#endif

namespace {

constexpr int kMaxDim = 256;

// optional last stage on one output value (see SIMPB_MLP_POST_* in the header)
__device__ __forceinline__ float post_stage(const simpb_mlp_chain& ch, float v, int row, int t) {
  if (ch.post == SIMPB_MLP_POST_REFINE3D) {
    if (ch.div && t >= ch.div_col0) v = v / ch.div[row / ch.div_rows];
    if (t < ch.res_cols) v += ch.res[(size_t)row * ch.ldres + t];
  } else if (ch.post == SIMPB_MLP_POST_REFINE2D) {
    if (t < ch.res_cols) {
      float a = ch.res[(size_t)row * ch.ldres + t];
      a = fminf(fmaxf(a, 0.f), 1.f);
      v += logf(fmaxf(a, 1e-5f) / fmaxf(1.f - a, 1e-5f));
    }
    v = 1.f / (1.f + expf(-v));
  } else if (ch.post == SIMPB_MLP_POST_SIGMOID) {
    v = 1.f / (1.f + expf(-v));
  }
  return v;
}

// Rows >= *m_live are capacity slots of the static 2D query set (no camera, nothing reads them): a workgroup whose rows are
// all dead writes zeros and leaves -- a quarter of the 1 536-slot launches' workgroups at N2 ~ 1 130.
__device__ __forceinline__ bool skip_dead_rows(const simpb_mlp_args& args, const simpb_mlp_chain& ch, int row0, int rows,
                                               int tid, int nthreads) {
  if (!args.m_live || row0 < *args.m_live) return false;
  int width = ch.in_mode == SIMPB_MLP_IN_SINE2D ? 256 : ch.in_dim;
  for (int o = 0; o < ch.n_ops; ++o)
    if (ch.ops[o].type == SIMPB_MLP_LINEAR) width = ch.ops[o].out_dim;
  for (int idx = tid; idx < rows * width; idx += nthreads) {
    const int r = idx / width, t = idx - r * width;
    if (row0 + r < args.num_rows) ch.out[(size_t)(row0 + r) * ch.ldo + t] = 0.f;
  }
  if (ch.in_mode == SIMPB_MLP_IN_ROWS_LN && ch.ln_out)
    for (int idx = tid; idx < rows * ch.in_dim; idx += nthreads) {
      const int r = idx / ch.in_dim, t = idx - r * ch.in_dim;
      if (row0 + r < args.num_rows) ch.ln_out[(size_t)(row0 + r) * ch.ld_ln_out + t] = 0.f;
    }
  return true;
}

template <int R, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void mlp_chain_kernel(simpb_mlp_args args) {
  constexpr int kThreads = WAVES * 64;
  __shared__ float act[2][R][kMaxDim];
  __shared__ float part[WAVES][R][kMaxDim];  // per-wave partial sums of the split-K linear
  const simpb_mlp_chain& ch = args.chain[blockIdx.y];
  const int tid = threadIdx.x;
  const int row0 = blockIdx.x * R;
  if (skip_dead_rows(args, args.chain[blockIdx.y], row0, R, threadIdx.x, WAVES * 64)) return;
  const int N = args.num_rows;

  // ---- input stage
  if (ch.in_mode == SIMPB_MLP_IN_SINE2D) {
    // pos2posemb2d (utils.py:40-63) of a 2-d point: 128 features of y then 128 of x;
    // feature i of an axis = sin or cos (even / odd i) of coord * 2*pi / 10000^(2*(i/2)/128)
    for (int idx = tid; idx < R * 256; idx += kThreads) {
      const int r = idx >> 8, j = idx & 255;
      const int row = row0 + r;
      float v = 0.f;
      if (row < N) {
        const int axis = j < 128 ? 1 : 0, i = j & 127;
        const float coord = ch.x[(size_t)row * ch.ldx + axis] * 6.283185307179586f;
        const float dim_t = powf(10000.f, (float)(2 * (i >> 1)) / 128.f);
        const float p = coord / dim_t;
        v = (i & 1) ? cosf(p) : sinf(p);
      }
      act[0][r][j] = v;
    }
  } else {
    for (int idx = tid; idx < R * ch.in_dim; idx += kThreads) {
      const int r = idx / ch.in_dim, k = idx - r * ch.in_dim;
      const int row = row0 + r;
      float v = 0.f;
      if (row < N) {
        v = ch.x[(size_t)row * ch.ldx + k];
        if (ch.x2) v += ch.x2[(size_t)row * ch.ldx2 + k];
      }
      act[0][r][k] = v;
    }
  }
  __syncthreads();

  int cur = 0;
  int width = ch.in_mode == SIMPB_MLP_IN_SINE2D ? 256 : ch.in_dim;
  for (int o = 0; o < ch.n_ops; ++o) {
    const simpb_mlp_op& op = ch.ops[o];
    if (op.type == SIMPB_MLP_LINEAR) {
      const int K = op.in_dim, D = op.out_dim;
      if ((D & 3) == 0 && K >= 16) {
        // Wide form. What bounds a layer is streaming W^T (up to 256 KiB) from L2 into ONE workgroup,
        // so the reduction is split over the 4 waves and each lane takes 4 adjacent columns: every
        // load is 16 B per lane (a 1-KiB row of W^T per wave-instruction), 4x the bytes in flight of a
        // scalar-column layout. Partial sums meet in LDS.
        const int lane = tid & 63, wave = tid >> 6;
        const int c4 = lane * 4;
        const int kq = (K + WAVES - 1) / WAVES;    // k-range per wave, in rows
        const int kb = wave * kq, ke = min(K, kb + kq);
        float4 acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c4 < D) {
          const float* wt = op.w + c4;
          constexpr int KU = 8;
          float4 wa[KU], wb[KU];
          int k = kb;
          const int kfull = kb + ((ke - kb) / KU) * KU;
          if (kfull > kb) {
#pragma unroll
            for (int j = 0; j < KU; ++j) wa[j] = *reinterpret_cast<const float4*>(wt + (size_t)(k + j) * D);
          }
          for (; k < kfull; k += KU) {
            const bool more = k + KU < kfull;
            if (more) {
#pragma unroll
              for (int j = 0; j < KU; ++j) wb[j] = *reinterpret_cast<const float4*>(wt + (size_t)(k + KU + j) * D);
            }
#pragma unroll
            for (int j = 0; j < KU; ++j) {
#pragma unroll
              for (int r = 0; r < R; ++r) {
                const float a = act[cur][r][k + j];
                acc[r].x = fmaf(a, wa[j].x, acc[r].x);
                acc[r].y = fmaf(a, wa[j].y, acc[r].y);
                acc[r].z = fmaf(a, wa[j].z, acc[r].z);
                acc[r].w = fmaf(a, wa[j].w, acc[r].w);
              }
            }
            if (more) {
#pragma unroll
              for (int j = 0; j < KU; ++j) wa[j] = wb[j];
            }
          }
          for (; k < ke; ++k) {
            const float4 w4 = *reinterpret_cast<const float4*>(wt + (size_t)k * D);
#pragma unroll
            for (int r = 0; r < R; ++r) {
              const float a = act[cur][r][k];
              acc[r].x = fmaf(a, w4.x, acc[r].x);
              acc[r].y = fmaf(a, w4.y, acc[r].y);
              acc[r].z = fmaf(a, w4.z, acc[r].z);
              acc[r].w = fmaf(a, w4.w, acc[r].w);
            }
          }
#pragma unroll
          for (int r = 0; r < R; ++r) *reinterpret_cast<float4*>(&part[wave][r][c4]) = acc[r];
        }
        __syncthreads();
        // thread t = column t: sum the 4 partials (fixed order), bias, activation
        if (tid < D) {
          const float b = op.b ? op.b[tid] : 0.f;
#pragma unroll
          for (int r = 0; r < R; ++r) {
            float v = part[0][r][tid];
#pragma unroll
            for (int w = 1; w < WAVES; ++w) v += part[w][r][tid];  // fixed order: deterministic
            v += b;
            act[cur ^ 1][r][tid] = op.relu ? fmaxf(v, 0.f) : v;
          }
        }
        __syncthreads();
        cur ^= 1;
        width = D;
        continue;
      }
      if (tid < D) {
        float acc[R];
        const float b = op.b ? op.b[tid] : 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = b;
        const float* wt = op.w + tid;  // W^T [K][D]
        for (int k = 0; k < K; ++k) {
          const float w0 = wt[(size_t)k * D];
#pragma unroll
          for (int r = 0; r < R; ++r) acc[r] = fmaf(act[cur][r][k], w0, acc[r]);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) act[cur ^ 1][r][tid] = op.relu ? fmaxf(acc[r], 0.f) : acc[r];
      }
      __syncthreads();
      cur ^= 1;
      width = D;
    } else {  // LayerNorm over `width` (= op.in_dim), eps 1e-5, biased variance (torch.nn.LayerNorm)
      const int D = op.in_dim;
      const int lane = tid & 63, wave = tid >> 6;
      for (int r = wave; r < R; r += kThreads / 64) {
        float v[kMaxDim / 64];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < kMaxDim / 64; ++j) {
          const int e = lane + 64 * j;
          v[j] = e < D ? act[cur][r][e] : 0.f;
          s += v[j];
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
        const float mean = s / (float)D;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < kMaxDim / 64; ++j) {
          const int e = lane + 64 * j;
          const float d = e < D ? v[j] - mean : 0.f;
          q += d * d;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) q += __shfl_xor(q, m);
        const float inv = 1.f / sqrtf(q / (float)D + 1e-5f);
#pragma unroll
        for (int j = 0; j < kMaxDim / 64; ++j) {
          const int e = lane + 64 * j;
          if (e < D) act[cur][r][e] = (v[j] - mean) * inv * op.w[e] + op.b[e];
        }
      }
      __syncthreads();
    }
  }

  // ---- output stage
  for (int idx = tid; idx < R * width; idx += kThreads) {
    const int r = idx / width, t = idx - r * width;
    const int row = row0 + r;
    if (row < N) {
      float v = act[cur][r][t];
      if (ch.out_scale) v *= ch.out_scale[t];
      if (ch.post) v = post_stage(ch, v, row, t);
      ch.out[(size_t)row * ch.ldo + t] = v;
    }
  }
}


// ---- matrix-core variant: 16 rows per workgroup, 8 waves; wave w owns output columns [32w, 32w+32)
// of a Linear layer as two 16x16 tiles of v_mfma_f32_16x16x4_f32 (exact fp32). Operands use the
// ORIGINAL weight layout [out][in] with the k-order permuted inside each 64-wide k-chunk: load j of
// lane quarter kq is the float4 at k = 16*j + 4*kq, so the four lanes of a weight row read 64
// CONTIGUOUS bytes per instruction (16 half-lines per wave-load; with k = 16*kq + 4*j it was four
// scattered 16-byte pieces per row = 32 lines per wave-load). A and B use the same permutation, so
// the sum is unchanged.
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kMR = 16, kMW = 8;

// compile-time loop (register-set indices must be constants in the front end: csrc/gemm.hip)
template <int D, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (D < N) {
    f(std::integral_constant<int, D>{});
    static_for<D + 1, N>(f);
  }
}

template <int CTRL>
__device__ __forceinline__ float dpp_add(float x) {
  const int y = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true);
  return x + __builtin_bit_cast(float, y);
}
// sum over the 32 lanes of a half wave (lanes 0-31 / 32-63), result in every lane of the half
__device__ __forceinline__ float half_wave_sum(float x) {
  x = dpp_add<0xB1>(x);    // quad_perm [1,0,3,2]: lane ^ 1
  x = dpp_add<0x4E>(x);    // quad_perm [2,3,0,1]: lane ^ 2
  x = dpp_add<0x141>(x);   // row_half_mirror: i <-> 7 - i inside 8 lanes (every quad already holds its sum)
  x = dpp_add<0x140>(x);   // row_mirror: i <-> 15 - i inside the 16-lane row
  return x + __shfl_xor(x, 16);
}

__global__ __launch_bounds__(kMW * 64) void mlp_chain_mfma_kernel(simpb_mlp_args args) {
  constexpr int kThreads = kMW * 64;
  __shared__ float act[2][kMR][kMaxDim + 4];   // +4: rows 16 B apart in bank space for the b128 reads
  const simpb_mlp_chain& ch = args.chain[blockIdx.y];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * kMR;
  if (skip_dead_rows(args, args.chain[blockIdx.y], row0, kMR, threadIdx.x, kMW * 64)) return;
  const int N = args.num_rows;
  SIMPB_STAMP(0);

  if (ch.in_mode == SIMPB_MLP_IN_SINE2D) {
    for (int idx = tid; idx < kMR * 256; idx += kThreads) {
      const int r = idx >> 8, j = idx & 255;
      const int row = row0 + r;
      float v = 0.f;
      if (row < N) {
        const int axis = j < 128 ? 1 : 0, i = j & 127;
        const float coord = ch.x[(size_t)row * ch.ldx + axis] * 6.283185307179586f;
        const float dim_t = powf(10000.f, (float)(2 * (i >> 1)) / 128.f);
        const float p = coord / dim_t;
        v = (i & 1) ? cosf(p) : sinf(p);
      }
      act[0][r][j] = v;
    }
  } else if ((ch.in_dim & (ch.in_dim - 1)) == 0 && ch.in_dim >= 4 && (ch.ldx & 3) == 0 && (ch.ldx2 & 3) == 0 &&
             ((reinterpret_cast<size_t>(ch.x) | reinterpret_cast<size_t>(ch.x2)) & 15) == 0) {
    // power-of-two width: 16-byte loads, shifts instead of divisions, every load of the tile in flight at once
    const int q = ch.in_dim >> 2, sh = __ffs(q) - 1;
#pragma unroll 2
    for (int idx = tid; idx < kMR * q; idx += kThreads) {
      const int r = idx >> sh, c4 = idx & (q - 1);
      const int row = row0 + r;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < N) {
        v = *reinterpret_cast<const float4*>(ch.x + (size_t)row * ch.ldx + 4 * c4);
        if (ch.x2) {
          const float4 u = *reinterpret_cast<const float4*>(ch.x2 + (size_t)row * ch.ldx2 + 4 * c4);
          v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
      }
      *reinterpret_cast<float4*>(&act[0][r][4 * c4]) = v;
    }
  } else {
    for (int idx = tid; idx < kMR * ch.in_dim; idx += kThreads) {
      const int r = idx / ch.in_dim, k = idx - r * ch.in_dim;
      const int row = row0 + r;
      float v = 0.f;
      if (row < N) {
        v = ch.x[(size_t)row * ch.ldx + k];
        if (ch.x2) v += ch.x2[(size_t)row * ch.ldx2 + k];
      }
      act[0][r][k] = v;
    }
  }
  __syncthreads();

  SIMPB_STAMP(1);
  int cur = 0;
  int width = ch.in_mode == SIMPB_MLP_IN_SINE2D ? 256 : ch.in_dim;
  for (int o = 0; o < ch.n_ops; ++o) {
    const simpb_mlp_op& op = ch.ops[o];
    SIMPB_STAMP(2 + 4 * o);
    if (op.type == SIMPB_MLP_LINEAR) {
      const int K = op.in_dim, D = op.out_dim;
      if ((K & 63) == 0) {  // any D: columns past D are fed zeros and not stored (D = 2..11 heads use wave 0 only)
        const int col0 = wave * 32;
        if (col0 < D) {
          const int r16 = lane & 15, kq = lane >> 4;
          const bool cv0 = col0 + r16 < D, cv1 = col0 + 16 + r16 < D;
          f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
          const float* w0 = op.w + (size_t)(cv0 ? col0 + r16 : 0) * K + 4 * kq;         // original layout [D][K]
          const float* w1 = op.w + (size_t)(cv1 ? col0 + 16 + r16 : 0) * K + 4 * kq;
          const float m0 = cv0 ? 1.f : 0.f, m1 = cv1 ? 1.f : 0.f;
          const float* ar = &act[cur][r16][4 * kq];
          const int c0 = col0 + r16, c1 = col0 + 16 + r16;
          const float bias0 = (op.b && cv0) ? op.b[c0] : 0.f, bias1 = (op.b && cv1) ? op.b[c1] : 0.f;  // in flight with the weights
          // Two register sets of weights (one 64-wide k-chunk each) ahead of the matrix work. Every
          // request is UNCONDITIONAL (chunk index clamped; a layer narrower than 256 re-requests its last
          // chunk) and the four chunk steps are unrolled at compile time, so the compiler can count the
          // loads in flight and wait for the older set only. (Requesting the whole layer in one burst of
          // four sets was measured too: no faster -- the CU's 64 B/clk load path serialises the eight
          // waves' bursts, 256 KB = 4k cycles per layer, and the last wave starts late.) Measured before (s_memtime stamps,
          // tools/chain_stamps.py): with the next chunk requested behind `if (more)` it waited for
          // everything in front of the matrix instructions, each chunk paid a full round trip (~2.7k
          // cycles) on top of its ~2k cycles of matrix work, and a 256x256 layer took ~23k cycles.
          const int nchunk = K >> 6;
          f32x4 wa[2][4], wb[2][4];
          auto fetch = [&](auto set_c, int chunk) __attribute__((always_inline)) {
            constexpr int set = decltype(set_c)::value;
            const int kc = (chunk < nchunk ? chunk : nchunk - 1) * 64;
            static_for<0, 4>([&](auto j) __attribute__((always_inline)) {
              constexpr int jj = decltype(j)::value;
              wa[set][jj] = *reinterpret_cast<const f32x4*>(w0 + kc + 16 * jj);
              wb[set][jj] = *reinterpret_cast<const f32x4*>(w1 + kc + 16 * jj);
            });
          };
          fetch(std::integral_constant<int, 0>{}, 0);
          fetch(std::integral_constant<int, 1>{}, 1);
          SIMPB_STAMP(3 + 4 * o);
          static_for<0, 4>([&](auto c_c) __attribute__((always_inline)) {
            constexpr int c = decltype(c_c)::value;
            constexpr int set = c & 1;
            if (c < nchunk) {
              static_for<0, 4>([&](auto j) __attribute__((always_inline)) {
                constexpr int jj = decltype(j)::value;
                const float4 a = *reinterpret_cast<const float4*>(ar + c * 64 + 16 * jj);
                const f32x4 b0 = wa[set][jj], b1 = wb[set][jj];
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0[0] * m0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b1[0] * m1, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b0[1] * m0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1[1] * m1, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b0[2] * m0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b1[2] * m1, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b0[3] * m0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b1[3] * m1, acc1, 0, 0, 0);
              });
            }
            if constexpr (c + 2 < 4) fetch(std::integral_constant<int, set>{}, c + 2);
          });
          // C/D of the 16x16 tile: column = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v0 = acc0[r] + bias0, v1 = acc1[r] + bias1;
            if (cv0) act[cur ^ 1][4 * kq + r][c0] = op.relu ? fmaxf(v0, 0.f) : v0;
            if (cv1) act[cur ^ 1][4 * kq + r][c1] = op.relu ? fmaxf(v1, 0.f) : v1;
          }
        }
      } else if (tid < D) {  // narrow / odd layers (K = 2, 3, 12, 32; D = 2..11): one thread per column
        float acc[kMR];
        const float b = op.b ? op.b[tid] : 0.f;
#pragma unroll
        for (int r = 0; r < kMR; ++r) acc[r] = b;
        const float* wr = op.w + (size_t)tid * K;
        for (int k = 0; k < K; ++k) {
          const float wv = wr[k];
#pragma unroll
          for (int r = 0; r < kMR; ++r) acc[r] = fmaf(act[cur][r][k], wv, acc[r]);
        }
#pragma unroll
        for (int r = 0; r < kMR; ++r) act[cur ^ 1][r][tid] = op.relu ? fmaxf(acc[r], 0.f) : acc[r];
      }
      SIMPB_STAMP(4 + 4 * o);
      __syncthreads();
      SIMPB_STAMP(5 + 4 * o);
      cur ^= 1;
      width = D;
    } else {
      // LayerNorm of all 16 rows at once: 32 lanes per row, 8 elements per lane; the 32-lane sums
      // are four DPP steps (quad xor 1, xor 2, half-row mirror, row mirror: register moves, no LDS
      // crossbar) plus one ds_bpermute for the two 16-lane rows of the half wave. (One wave per row
      // with six __shfl_xor steps per sum, two rows in sequence, was ~1.5 us of dependent
      // ds_bpermute latency per LayerNorm, as much as the matrix work of a 128-wide layer.)
      const int D = op.in_dim;
      const int r = tid >> 5, l32 = tid & 31;
      float g[kMaxDim / 32], be[kMaxDim / 32], v[kMaxDim / 32];
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < kMaxDim / 32; ++j) {
        const int e = l32 + 32 * j;
        const bool in = e < D;
        g[j] = in ? op.w[e] : 0.f;   // requested before the reductions need them
        be[j] = in ? op.b[e] : 0.f;
        v[j] = in ? act[cur][r][e] : 0.f;
        sum += v[j];
      }
      sum = half_wave_sum(sum);
      const float mean = sum / (float)D;
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < kMaxDim / 32; ++j) {
        const float d = (l32 + 32 * j) < D ? v[j] - mean : 0.f;
        q += d * d;
      }
      q = half_wave_sum(q);
      const float inv = 1.f / sqrtf(q / (float)D + 1e-5f);
#pragma unroll
      for (int j = 0; j < kMaxDim / 32; ++j) {
        const int e = l32 + 32 * j;
        if (e < D) act[cur][r][e] = (v[j] - mean) * inv * g[j] + be[j];
      }
      SIMPB_STAMP(4 + 4 * o);
      __syncthreads();
      SIMPB_STAMP(5 + 4 * o);
    }
  }
  SIMPB_STAMP(126);
  for (int idx = tid; idx < kMR * width; idx += kThreads) {
    const int r = idx / width, t = idx - r * width;
    const int row = row0 + r;
    if (row < N) {
      float v = act[cur][r][t];
      if (ch.out_scale) v *= ch.out_scale[t];
      if (ch.post) v = post_stage(ch, v, row, t);
      ch.out[(size_t)row * ch.ldo + t] = v;
    }
  }
  SIMPB_STAMP(127);
}


// ---- 4-row variant on the 4x4 matrix blocks (weights_transposed == 2).
// Why: a chain over M = 900 rows in 16-row workgroups is 57 workgroups on 256 CUs (round-1 profile: MFMA busy 0.066, 64 %
// of wave cycles waiting), and a 16x16 tile cannot be made shorter without wasting it. v_mfma_f32_4x4x1f32 multiplies
// SIXTEEN independent 4x4 blocks per instruction at the same flop rate: block b = output columns 4b..4b+3 of a wave's 64
// columns, the four rows of the block = the four rows of the workgroup, k advances by one per instruction. So 4 rows per
// workgroup lose nothing: 225 workgroups for 900 rows, each with a quarter of the matrix work per layer.
//   lane l = 4*b + j: A operand = x[row j][k] (LDS), B operand = W[column 64*wave + l][k], D[i] = out[row i][column].
// Weights come k4-packed, Wp[k / 4][column][k % 4] (host repack, plugin/fused.py): the B operands of four k for a
// lane are one 16-byte load, and a wave-load is 1 KiB contiguous. No weight staging through LDS, no barrier inside a layer.
// Measured (tools/chain_stamps.py, tools/bench_chain.py): ~10.4k cycles per 256x256 layer against ~19k for the 16-row kernel
// (refine3d chain 29.6 us vs 42.3 us, anchor encoder 14.1 vs 18.6, 2D encoder 20.1 vs 28.6). The layer time is the same for
// ONE busy wave (a 256 -> 11 head) as for four and does not change with eight accumulators, a branch-free step, or a
// rotated k order across workgroups: v_mfma_f32_4x4x1f32 issues once per ~38 cycles on gfx950, i.e. at a quarter of the
// 16x16x4 flop rate, so 4 rows per CU cost as many matrix cycles as 16 do. What is gained is the 4x CU count, not
// matrix efficiency; the floor left is the 256 KB of weights per layer through one CU's 64 B/clk load path (4k cycles).
// Also measured and dropped: 16 waves per workgroup with K split four ways and every weight of a layer requested up front
// (partials meeting in LDS): the single wide chain gains (refine3d 29.6 -> 24.2 us, ~7k cycles per layer) but launches of
// several chains lose more (anchor encoder 14.1 -> 60.6 us, refine2d 50.5 -> 78 us: 1024-thread workgroups, two per CU).
constexpr int kR4 = 4;

__device__ __forceinline__ float wave_sum(float x) { x = half_wave_sum(x); return x + __shfl_xor(x, 32); }

__global__ __launch_bounds__(256) void mlp_chain_r4_kernel(simpb_mlp_args args) {
  constexpr int kThreads = 256;
  __shared__ float act[2][kR4][kMaxDim + 4];
  const simpb_mlp_chain& ch = args.chain[blockIdx.y];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * kR4;
  if (skip_dead_rows(args, args.chain[blockIdx.y], row0, kR4, threadIdx.x, 256)) return;
  const int N = args.num_rows;

  if (ch.in_mode == SIMPB_MLP_IN_SINE2D) {
    for (int idx = tid; idx < kR4 * 256; idx += kThreads) {
      const int r = idx >> 8, j = idx & 255;
      const int row = row0 + r;
      float v = 0.f;
      if (row < N) {
        const int axis = j < 128 ? 1 : 0, i = j & 127;
        const float coord = ch.x[(size_t)row * ch.ldx + axis] * 6.283185307179586f;
        const float dim_t = powf(10000.f, (float)(2 * (i >> 1)) / 128.f);
        const float p = coord / dim_t;
        v = (i & 1) ? cosf(p) : sinf(p);
      }
      act[0][r][j] = v;
    }
  } else if (ch.in_mode == SIMPB_MLP_IN_ROWS_LN) {
    // the decoder's `norm` operator in front of this head, inside the launch: wave r = row r, 64 lanes x 4 elements (the
    // LayerNorm stage's arithmetic below), + x2 afterwards; the chain that carries ln_out writes the operator's output
    const int D = ch.in_dim, r = wave, row = row0 + r;
    const int live = args.m_live ? min(N, *args.m_live) : N;
    const bool in_rows = row < N, alive = row < live;
    float g[kMaxDim / 64], be[kMaxDim / 64], v[kMaxDim / 64], x2v[kMaxDim / 64];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < kMaxDim / 64; ++j) {
      const int e = lane + 64 * j;
      const bool in = e < D;
      g[j] = in ? ch.ln_w[e] : 0.f;
      be[j] = in ? ch.ln_b[e] : 0.f;
      v[j] = (in && in_rows) ? ch.x[(size_t)row * ch.ldx + e] : 0.f;
      x2v[j] = (in && in_rows && ch.x2) ? ch.x2[(size_t)row * ch.ldx2 + e] : 0.f;
      sum += v[j];
    }
    sum = wave_sum(sum);
    const float mean = sum / (float)D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < kMaxDim / 64; ++j) {
      const float d = (lane + 64 * j) < D ? v[j] - mean : 0.f;
      q += d * d;
    }
    q = wave_sum(q);
    const float inv = 1.f / sqrtf(q / (float)D + 1e-5f);
#pragma unroll
    for (int j = 0; j < kMaxDim / 64; ++j) {
      const int e = lane + 64 * j;
      if (e < D) {
        const float y = alive ? (v[j] - mean) * inv * g[j] + be[j] : 0.f;   // (capacity rows: zeros, as the LayerNorm launch wrote)
        if (ch.ln_out && in_rows) ch.ln_out[(size_t)row * ch.ld_ln_out + e] = y;
        act[0][r][e] = y + x2v[j];
      }
    }
  } else {
    for (int idx = tid; idx < kR4 * ch.in_dim; idx += kThreads) {
      const int r = idx / ch.in_dim, k = idx - r * ch.in_dim;
      const int row = row0 + r;
      float v = 0.f;
      if (row < N) {
        v = ch.x[(size_t)row * ch.ldx + k];
        if (ch.x2) v += ch.x2[(size_t)row * ch.ldx2 + k];
      }
      act[0][r][k] = v;
    }
  }
  __syncthreads();

  int cur = 0;
  int width = ch.in_mode == SIMPB_MLP_IN_SINE2D ? 256 : ch.in_dim;
  for (int o = 0; o < ch.n_ops; ++o) {
    const simpb_mlp_op& op = ch.ops[o];
    if (op.type == SIMPB_MLP_LINEAR) {
      const int K = op.in_dim, D = op.out_dim;
      if ((K & 3) == 0) {  // k4-packed weights
        if (wave * 64 < D) {
          const int n = tid;
          const bool cv = n < D;
          const float4* wp = reinterpret_cast<const float4*>(op.w) + (cv ? n : 0);
          const float* ar = &act[cur][lane & 3][0];
          const float bias = (op.b && cv) ? op.b[n] : 0.f;
          const int steps = K >> 2;                 // one step = 4 k = one 16-byte B load + one 16-byte A read
          f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
          constexpr int G = 8;                      // steps per register set; two sets in flight
          float4 wa[G], wb[G];
          auto fetch = [&](float4* dst, int g) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < G; ++j) dst[j] = wp[(size_t)min(g * G + j, steps - 1) * D];  // unconditional, clamped
          };
          auto work = [&](const float4* w, int g) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < G; ++j) {
              const int st = g * G + j;
              if (st < steps) {  // wave-uniform
                const float4 a = *reinterpret_cast<const float4*>(ar + 4 * st);
                acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a.x, w[j].x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a.y, w[j].y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a.z, w[j].z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a.w, w[j].w, acc1, 0, 0, 0);
              }
            }
          };
          const int groups = (steps + G - 1) / G;
          fetch(wa, 0);
          for (int g = 0; g < groups; g += 2) {
            fetch(wb, g + 1);
            work(wa, g);
            fetch(wa, g + 2);
            if (g + 1 < groups) work(wb, g + 1);
          }
          if (cv) {
#pragma unroll
            for (int i = 0; i < kR4; ++i) {
              const float v = acc0[i] + acc1[i] + bias;
              act[cur ^ 1][i][n] = op.relu ? fmaxf(v, 0.f) : v;
            }
          }
        }
      } else if (tid < D) {  // K = 2, 3, ...: weights as stored, [D][K]
        float acc[kR4];
        const float b = op.b ? op.b[tid] : 0.f;
#pragma unroll
        for (int r = 0; r < kR4; ++r) acc[r] = b;
        const float* wr = op.w + (size_t)tid * K;
        for (int k = 0; k < K; ++k) {
          const float wv = wr[k];
#pragma unroll
          for (int r = 0; r < kR4; ++r) acc[r] = fmaf(act[cur][r][k], wv, acc[r]);
        }
#pragma unroll
        for (int r = 0; r < kR4; ++r) act[cur ^ 1][r][tid] = op.relu ? fmaxf(acc[r], 0.f) : acc[r];
      }
      __syncthreads();
      cur ^= 1;
      width = D;
    } else {
      // LayerNorm: wave r = row r, 64 lanes x 4 elements
      const int D = op.in_dim;
      const int r = wave;
      float g[kMaxDim / 64], be[kMaxDim / 64], v[kMaxDim / 64];
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < kMaxDim / 64; ++j) {
        const int e = lane + 64 * j;
        const bool in = e < D;
        g[j] = in ? op.w[e] : 0.f;
        be[j] = in ? op.b[e] : 0.f;
        v[j] = in ? act[cur][r][e] : 0.f;
        sum += v[j];
      }
      sum = wave_sum(sum);
      const float mean = sum / (float)D;
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < kMaxDim / 64; ++j) {
        const float d = (lane + 64 * j) < D ? v[j] - mean : 0.f;
        q += d * d;
      }
      q = wave_sum(q);
      const float inv = 1.f / sqrtf(q / (float)D + 1e-5f);
#pragma unroll
      for (int j = 0; j < kMaxDim / 64; ++j) {
        const int e = lane + 64 * j;
        if (e < D) act[cur][r][e] = (v[j] - mean) * inv * g[j] + be[j];
      }
      __syncthreads();
    }
  }
  for (int idx = tid; idx < kR4 * width; idx += kThreads) {
    const int r = idx / width, t = idx - r * width;
    const int row = row0 + r;
    if (row < N) {
      float v = act[cur][r][t];
      if (ch.out_scale) v *= ch.out_scale[t];
      if (ch.post) v = post_stage(ch, v, row, t);
      ch.out[(size_t)row * ch.ldo + t] = v;
    }
  }
}

// ---- 32-row variant on the 32x32 matrix tiles (weights_transposed == 3), for launches with thousands of rows (a batch of
// camera streams per launch, runner independent_streams). The 4-row kernel above exists to spread ~1 k rows over 256 CUs; its
// v_mfma_f32_4x4x1f32 issues at a quarter of the fp32 matrix rate, and at ~9 k rows (bs = 8) that IS the bound: measured
// 33-36 TFLOP/s on every chain (refine2d 213 us, 2D encoder 64 us, per step of 8 frames), the 4x4x1 issue peak. Here:
//   one workgroup = 32 rows x 4 waves; the row tile lives in LDS between layers (two buffers, 66 KB: two workgroups per CU);
//   wave w owns output columns 64w..64w+63 = two 32x32 tiles of v_mfma_f32_32x32x2f32 (full fp32 matrix rate);
//   lane (r32, half) holds k = 16 * half + 0..15 of a 32-deep chunk of its A row (LDS, four ds_read_b128) and of its B
//   column (the same k permutation on both operands, as in csrc/gemm.hip), sixteen matrix steps per tile and chunk;
//   weights arrive FRAGMENT-PACKED by the host (plugin/fused.py): Wq[tile][chunk][q][lane][4] = W[32 tile + r32][32 chunk +
//   16 half + 4 q + 0..3], columns past out_dim as zeros -- a wave's B operand of a chunk is four fully coalesced 1-KiB
//   loads straight into registers (no staging, no barrier inside a layer), two chunks in flight, every request
//   unconditional with a clamped index so that the waits are counted.
// Layers with in_dim % 32 != 0 (the 2-, 3- and 12-wide first layers) take nn.Linear's layout on the vector units.
constexpr int kR32 = 32;
constexpr int kLdAct = kMaxDim + 4;   // row stride = 4 mod 64 floats: the 16 lanes of a ds_read_b128 phase hit 16 distinct 4-bank slots

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 2) void mlp_chain_r32_kernel(simpb_mlp_args args) {
  constexpr int kThreads = WAVES * 64;
  constexpr bool kTwoTiles = WAVES == 4;   // 4 waves: two column tiles per wave; 8 waves: one
  __shared__ float act[2][kR32][kLdAct];
  const simpb_mlp_chain& ch = args.chain[blockIdx.y];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * kR32;
  if (skip_dead_rows(args, args.chain[blockIdx.y], row0, kR32, threadIdx.x, kThreads)) return;
  const int N = args.num_rows;

  if (ch.in_mode == SIMPB_MLP_IN_SINE2D) {
    for (int idx = tid; idx < kR32 * 256; idx += kThreads) {
      const int r = idx >> 8, j = idx & 255;
      const int row = row0 + r;
      float v = 0.f;
      if (row < N) {
        const int axis = j < 128 ? 1 : 0, i = j & 127;
        const float coord = ch.x[(size_t)row * ch.ldx + axis] * 6.283185307179586f;
        const float dim_t = powf(10000.f, (float)(2 * (i >> 1)) / 128.f);
        const float p = coord / dim_t;
        v = (i & 1) ? cosf(p) : sinf(p);
      }
      act[0][r][j] = v;
    }
  } else if (ch.in_mode == SIMPB_MLP_IN_ROWS_LN) {
    // the decoder's `norm` operator in front of this head (the 4-row kernel's arithmetic): one wave per row, eight rows each
    const int D = ch.in_dim;
    const int live = args.m_live ? min(N, *args.m_live) : N;
    float g[kMaxDim / 64], be[kMaxDim / 64];
#pragma unroll
    for (int j = 0; j < kMaxDim / 64; ++j) {
      const int e = lane + 64 * j;
      g[j] = e < D ? ch.ln_w[e] : 0.f;
      be[j] = e < D ? ch.ln_b[e] : 0.f;
    }
    for (int r = wave; r < kR32; r += WAVES) {
      const int row = row0 + r;
      const bool in_rows = row < N, alive = row < live;
      float v[kMaxDim / 64], x2v[kMaxDim / 64];
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < kMaxDim / 64; ++j) {
        const int e = lane + 64 * j;
        const bool in = e < D;
        v[j] = (in && in_rows) ? ch.x[(size_t)row * ch.ldx + e] : 0.f;
        x2v[j] = (in && in_rows && ch.x2) ? ch.x2[(size_t)row * ch.ldx2 + e] : 0.f;
        sum += v[j];
      }
      sum = wave_sum(sum);
      const float mean = sum / (float)D;
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < kMaxDim / 64; ++j) {
        const float d = (lane + 64 * j) < D ? v[j] - mean : 0.f;
        q += d * d;
      }
      q = wave_sum(q);
      const float inv = 1.f / sqrtf(q / (float)D + 1e-5f);
#pragma unroll
      for (int j = 0; j < kMaxDim / 64; ++j) {
        const int e = lane + 64 * j;
        if (e < D) {
          const float y = alive ? (v[j] - mean) * inv * g[j] + be[j] : 0.f;
          if (ch.ln_out && in_rows) ch.ln_out[(size_t)row * ch.ld_ln_out + e] = y;
          act[0][r][e] = y + x2v[j];
        }
      }
    }
  } else {
    for (int idx = tid; idx < kR32 * ch.in_dim; idx += kThreads) {
      const int r = idx / ch.in_dim, k = idx - r * ch.in_dim;
      const int row = row0 + r;
      float v = 0.f;
      if (row < N) {
        v = ch.x[(size_t)row * ch.ldx + k];
        if (ch.x2) v += ch.x2[(size_t)row * ch.ldx2 + k];
      }
      act[0][r][k] = v;
    }
  }
  __syncthreads();

  const int r32 = lane & 31, half = lane >> 5;
  int cur = 0;
  int width = ch.in_mode == SIMPB_MLP_IN_SINE2D ? 256 : ch.in_dim;
  for (int o = 0; o < ch.n_ops; ++o) {
    const simpb_mlp_op& op = ch.ops[o];
    if (op.type == SIMPB_MLP_LINEAR) {
      const int K = op.in_dim, D = op.out_dim;
      if ((K & 31) == 0) {  // fragment-packed weights
        const int ntile = (D + 31) >> 5, nchunk = K >> 5;
        const int t0 = kTwoTiles ? 2 * wave : wave;
        if (t0 < ntile) {
          const bool two = kTwoTiles && t0 + 1 < ntile;     // wave-uniform
          const int t1 = two ? t0 + 1 : t0;                 // (a lone last tile is loaded twice and multiplied once)
          const f32x4* w0 = reinterpret_cast<const f32x4*>(op.w) + (size_t)t0 * nchunk * 256 + lane;
          const f32x4* w1 = reinterpret_cast<const f32x4*>(op.w) + (size_t)t1 * nchunk * 256 + lane;
          const float* ar = &act[cur][r32][16 * half];
          f32x16 acc0, acc1;
#pragma unroll
          for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
          f32x4 wa[2][4], wb[2][4];
          auto fetch = [&](auto set_c, int chunk) __attribute__((always_inline)) {
            constexpr int set = decltype(set_c)::value;
            const int c = min(chunk, nchunk - 1);           // unconditional, clamped: past the end the last chunk again
            static_for<0, 4>([&](auto q) __attribute__((always_inline)) {
              constexpr int qq = decltype(q)::value;
              wa[set][qq] = w0[(size_t)c * 256 + 64 * qq];
              if constexpr (kTwoTiles) wb[set][qq] = w1[(size_t)c * 256 + 64 * qq];
            });
          };
          auto work = [&](auto set_c, int chunk) __attribute__((always_inline)) {
            constexpr int set = decltype(set_c)::value;
            f32x4 a[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = *reinterpret_cast<const f32x4*>(ar + 32 * chunk + 4 * q);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int e = 0; e < 4; ++e) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][e], wa[set][q][e], acc0, 0, 0, 0);
            if (kTwoTiles && two) {
#pragma unroll
              for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][e], wb[set][q][e], acc1, 0, 0, 0);
            }
          };
          fetch(std::integral_constant<int, 0>{}, 0);
          for (int c = 0; c < nchunk; c += 2) {
            fetch(std::integral_constant<int, 1>{}, c + 1);
            work(std::integral_constant<int, 0>{}, c);
            fetch(std::integral_constant<int, 0>{}, c + 2);
            if (c + 1 < nchunk) work(std::integral_constant<int, 1>{}, c + 1);
          }
          // C/D of a 32x32 tile: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * half
          const int c0 = 32 * t0 + r32, c1 = 32 * t0 + 32 + r32;
          const bool cv0 = c0 < D, cv1 = two && c1 < D;
          const float bias0 = (op.b && cv0) ? op.b[c0] : 0.f, bias1 = (op.b && cv1) ? op.b[c1] : 0.f;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int rr = (i & 3) + 8 * (i >> 2) + 4 * half;
            const float v0 = acc0[i] + bias0, v1 = acc1[i] + bias1;
            if (cv0) act[cur ^ 1][rr][c0] = op.relu ? fmaxf(v0, 0.f) : v0;
            if (cv1) act[cur ^ 1][rr][c1] = op.relu ? fmaxf(v1, 0.f) : v1;
          }
        }
      } else if (tid < D) {  // K = 2, 3, 12: weights as stored, [D][K]; one thread per column
        float acc[kR32];
        const float b = op.b ? op.b[tid] : 0.f;
#pragma unroll
        for (int r = 0; r < kR32; ++r) acc[r] = b;
        const float* wr = op.w + (size_t)tid * K;
        for (int k = 0; k < K; ++k) {
          const float wv = wr[k];
#pragma unroll
          for (int r = 0; r < kR32; ++r) acc[r] = fmaf(act[cur][r][k], wv, acc[r]);
        }
#pragma unroll
        for (int r = 0; r < kR32; ++r) act[cur ^ 1][r][tid] = op.relu ? fmaxf(acc[r], 0.f) : acc[r];
      }
      __syncthreads();
      cur ^= 1;
      width = D;
    } else {
      // LayerNorm: one wave per row, 64 lanes x 4 elements, eight rows per wave
      const int D = op.in_dim;
      float g[kMaxDim / 64], be[kMaxDim / 64];
#pragma unroll
      for (int j = 0; j < kMaxDim / 64; ++j) {
        const int e = lane + 64 * j;
        g[j] = e < D ? op.w[e] : 0.f;
        be[j] = e < D ? op.b[e] : 0.f;
      }
#pragma unroll 2
      for (int r = wave; r < kR32; r += WAVES) {
        float v[kMaxDim / 64];
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < kMaxDim / 64; ++j) {
          const int e = lane + 64 * j;
          v[j] = e < D ? act[cur][r][e] : 0.f;
          sum += v[j];
        }
        sum = wave_sum(sum);
        const float mean = sum / (float)D;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < kMaxDim / 64; ++j) {
          const float d = (lane + 64 * j) < D ? v[j] - mean : 0.f;
          q += d * d;
        }
        q = wave_sum(q);
        const float inv = 1.f / sqrtf(q / (float)D + 1e-5f);
#pragma unroll
        for (int j = 0; j < kMaxDim / 64; ++j) {
          const int e = lane + 64 * j;
          if (e < D) act[cur][r][e] = (v[j] - mean) * inv * g[j] + be[j];
        }
      }
      __syncthreads();
    }
  }
  for (int idx = tid; idx < kR32 * width; idx += kThreads) {
    const int r = idx / width, t = idx - r * width;
    const int row = row0 + r;
    if (row < N) {
      float v = act[cur][r][t];
      if (ch.out_scale) v *= ch.out_scale[t];
      if (ch.post) v = post_stage(ch, v, row, t);
      ch.out[(size_t)row * ch.ldo + t] = v;
    }
  }
}

}  // namespace

extern "C" int simpb_mlp_chain_forward(const simpb_mlp_args* args, void* stream) {
  if (!args || args->num_rows <= 0 || args->num_chains <= 0 || args->num_chains > SIMPB_MLP_MAX_CHAINS)
    return SIMPB_EINVAL;
  for (int c = 0; c < args->num_chains; ++c) {
    const simpb_mlp_chain& ch = args->chain[c];
    if (!ch.x || !ch.out || ch.n_ops < 0 || ch.n_ops > SIMPB_MLP_MAX_OPS) return SIMPB_EINVAL;
    int width = ch.in_mode == SIMPB_MLP_IN_SINE2D ? 256 : ch.in_dim;
    if (width <= 0 || width > kMaxDim) return SIMPB_EINVAL;
    if (ch.post < 0 || ch.post > SIMPB_MLP_POST_SIGMOID) return SIMPB_EINVAL;
    if ((ch.post == SIMPB_MLP_POST_REFINE3D || ch.post == SIMPB_MLP_POST_REFINE2D) && (!ch.res || ch.ldres < ch.res_cols))
      return SIMPB_EINVAL;
    if (ch.post == SIMPB_MLP_POST_REFINE3D && ch.div && ch.div_rows <= 0) return SIMPB_EINVAL;
    if (ch.in_mode == SIMPB_MLP_IN_SINE2D && ch.ldx < 2) return SIMPB_EINVAL;
    if (ch.in_mode == SIMPB_MLP_IN_ROWS_LN && ((args->weights_transposed != 2 && args->weights_transposed != 3) || !ch.ln_w || !ch.ln_b ||
                                               (ch.ln_out && ch.ld_ln_out < ch.in_dim)))
      return SIMPB_EINVAL;
    if (ch.in_mode < 0 || ch.in_mode > SIMPB_MLP_IN_ROWS_LN) return SIMPB_EINVAL;
    for (int o = 0; o < ch.n_ops; ++o) {
      const simpb_mlp_op& op = ch.ops[o];
      if (op.type == SIMPB_MLP_LINEAR) {
        if (!op.w || op.in_dim != width || op.out_dim <= 0 || op.out_dim > kMaxDim) return SIMPB_EINVAL;
        width = op.out_dim;
      } else if (op.type == SIMPB_MLP_LAYERNORM) {
        if (!op.w || !op.b || op.in_dim != width) return SIMPB_EINVAL;
      } else {
        return SIMPB_EINVAL;
      }
    }
  }
  (void)hipGetLastError();
  if (args->weights_transposed == 3) {
    // 32-row variant on the 32x32 matrix tiles (fragment-packed weights): launches with thousands of rows
    dim3 grid((args->num_rows + kR32 - 1) / kR32, args->num_chains);
    // (four register sets = three chunks in flight ahead of the one multiplied: 152 vs 147 us for refine2d, no better;
    // 8 waves with one column tile each were measured too: refine2d 204 us against 147 us at 8.9 k rows -- every wave
    // re-reads the A tile, and half the B registers in flight per wave)
    hipLaunchKernelGGL(mlp_chain_r32_kernel<4>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), *args);
  } else if (args->weights_transposed == 2) {
    // 4-row matrix-core variant (k4-packed weights): 4 rows x 4 waves
    dim3 grid((args->num_rows + kR4 - 1) / kR4, args->num_chains);
    hipLaunchKernelGGL(mlp_chain_r4_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), *args);
  } else if (args->weights_transposed) {
    // VALU variant (weights [in][out]): 4 rows x 8 waves, split-K with 16-byte weight loads
    constexpr int R = 4, WAVES = 8;
    dim3 grid((args->num_rows + R - 1) / R, args->num_chains);
    hipLaunchKernelGGL((mlp_chain_kernel<R, WAVES>), grid, dim3(WAVES * 64), 0, static_cast<hipStream_t>(stream), *args);
  } else {
    // matrix-core variant (weights in nn.Linear's own [out][in] layout): 16 rows x 8 waves
    dim3 grid((args->num_rows + kMR - 1) / kMR, args->num_chains);
    hipLaunchKernelGGL(mlp_chain_mfma_kernel, grid, dim3(kMW * 64), 0, static_cast<hipStream_t>(stream), *args);
  }
  return simpb_check_launch();
}
