// Shared device helpers + the auxiliary work blocks that ride on a GEMM launch.
//
// A kernel boundary costs ~4.7 us of timeline on this stack even for a trivial kernel
// (profiles/round1_bench_kernel_stats.csv), so small independent jobs are appended to the first
// GEMM launch of the step as extra workgroups instead of being launched on their own:
//   * the Philox noise fill (eps ~ N(0,1), u ~ U[tiny,1)),
//   * the per-step preparation of the chain kernels' LDS weight images (padding / transposition
//     done ONCE per step in global memory, so each of the 64 chain workgroups only issues a linear
//     asynchronous LDS-DMA copy instead of ~10 serialized L2 round trips).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gmvae {

constexpr int kThreads = 256;   // 4 wavefronts of 64
constexpr float kTiny = 1.17549435e-38f;

// ---------------------------------------------------------------- Philox
// Philox4x32-10 (Salmon et al. 2011), counter = (index, stream, step), key = seed.
// Stands in for tf.random_normal / tf.random_uniform inside the TFP samplers
// (scripts/gmvae.py:240,248; scripts/vae.py:171) -- statistically, not bitwise.
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}
__device__ __forceinline__ float u01(uint32_t b) { return (float)(b >> 8) * 5.9604644775390625e-8f; }  // [0,1)

// 4 consecutive values of ONE ROW of the eps stream (Box-Muller normals) or of the u stream (uniforms in [kTiny, 1)):
// quad `quad` of GLOBAL row `row`, keyed by (seed, step) -- every consumer of the noise calls this one function.
// The counter holds the global row index (GmvaeDims::row0 + local row), never a position in a device's buffer, so a
// batch sharded over G devices draws exactly the rows the single-device step on the whole batch would draw.
__device__ __forceinline__ void noise_vals(const uint64_t row, const uint32_t quad, const bool is_u, const uint64_t seed,
                                           const uint64_t step, float (&o)[4]) {
  uint32_t c[4] = {(uint32_t)row, (quad & 0x00ffffffu) | (((uint32_t)(row >> 32) & 0x3fu) << 24) | (is_u ? 0x80000000u : 0u),
                   (uint32_t)step, (uint32_t)(step >> 32)};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  if (is_u) {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = fmaxf(u01(c[j]), kTiny);
  } else {
#pragma unroll
    // Box-Muller on the hardware transcendentals: v_log_f32 (log2), v_sqrt_f32, and v_sin/v_cos_f32, whose
    // argument is in TURNS -- the uniform goes in as it is.  (The libm forms cost ~10x the instructions, which
    // matters where mega_fwd_bwd draws its panel's noise itself.)
    for (int j = 0; j < 4; j += 2) {
      const float r = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(1.f - u01(c[j])));   // 1-u in (0,1]
      const float t = u01(c[j + 1]);
      o[j] = r * __builtin_amdgcn_cosf(t);
      o[j + 1] = r * __builtin_amdgcn_sinf(t);
    }
  }
}

// thread `i` of the fill of eps [rows][L] and u [rows][K] (either pointer may be null): one quad of one row
__device__ __forceinline__ void noise_item(uint64_t i, float* eps, float* u, uint64_t rows, int L, int K, uint64_t row_base,
                                           uint64_t seed, uint64_t step) {
  const uint32_t qe = eps ? (uint32_t)(L + 3) / 4 : 0u, qu = u ? (uint32_t)(K + 3) / 4 : 0u;
  const uint64_t n_e = rows * qe;
  if (i >= n_e + rows * qu) return;
  const bool is_u = i >= n_e;
  if (is_u) i -= n_e;
  const uint32_t qpr = is_u ? qu : qe;
  const uint64_t row = i / qpr;
  const uint32_t quad = (uint32_t)(i - row * qpr);
  float o[4];
  noise_vals(row_base + row, quad, is_u, seed, step, o);
  const int n = is_u ? K : L;
  float* dst = (is_u ? u : eps) + row * (uint64_t)n + quad * 4;
  if ((n & 3) == 0) {
    *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if ((int)quad * 4 + j < n) dst[j] = o[j];
  }
}
__host__ __device__ inline uint64_t noise_items(bool has_eps, bool has_u, uint64_t rows, int L, int K) {
  return rows * ((has_eps ? (uint64_t)(L + 3) / 4 : 0) + (has_u ? (uint64_t)(K + 3) / 4 : 0));
}

// ------------------------------------------------------- matrix copies
// Cooperative copies of a row-major [rows][cols] matrix by one workgroup; dst is LDS or global.
// Loads are issued in independent batches (4 x 16 B or 8 x 4 B per thread in flight) -- a naive
// one-element-per-iteration loop serialises ~60 L2 round trips per thread.
//   TRANS = false: dst[r*ld + c] = src[r][c]        TRANS = true: dst[c*ld + r] = src[r][c]
template <bool TRANS>
__device__ __forceinline__ void mat_fill(float* __restrict__ dst, const int ld, const float* __restrict__ src,
                                         const int rows, const int cols, const int src_ld, const int tid) {
  const bool vec = ((cols | src_ld) & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
  if (vec) {
    const int q = cols >> 2, n4 = rows * q;
    for (int base = tid; base < n4; base += kThreads * 4) {
      float4 v[4];
      int rr[4], cc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ic = min(base + j * kThreads, n4 - 1);
        rr[j] = ic / q;
        cc[j] = (ic - rr[j] * q) << 2;
        v[j] = *reinterpret_cast<const float4*>(src + (long long)rr[j] * src_ld + cc[j]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (base + j * kThreads < n4) {
          if (!TRANS) {
            float* p = dst + rr[j] * ld + cc[j];
            p[0] = v[j].x; p[1] = v[j].y; p[2] = v[j].z; p[3] = v[j].w;
          } else {
            float* p = dst + cc[j] * ld + rr[j];
            p[0] = v[j].x; p[ld] = v[j].y; p[2 * ld] = v[j].z; p[3 * ld] = v[j].w;
          }
        }
      }
    }
  } else {
    const int n = rows * cols;
    for (int base = tid; base < n; base += kThreads * 8) {
      float v[8];
      int rr[8], cc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ic = min(base + j * kThreads, n - 1);
        rr[j] = ic / cols;
        cc[j] = ic - rr[j] * cols;
        v[j] = src[(long long)rr[j] * src_ld + cc[j]];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (base + j * kThreads < n) dst[TRANS ? cc[j] * ld + rr[j] : rr[j] * ld + cc[j]] = v[j];
    }
  }
}

// Dynamic binarisation of the input pipeline (scripts/runners.py:44-47 `_preprocess`), one quad (4 consecutive pixels
// of one output row): image = cast(pixel, float32) / 255.; x = image < uniform(shape)  (so P[x = 1] = 1 - pixel/255).
// One 4-byte load, one Philox4x32-10 call (counter = output quad index, stream tag 0x40000000, key = seed, step), one
// 4-byte store.
__device__ __forceinline__ void binarize_quad(const uint64_t q, const unsigned char* __restrict__ pixels,
                                              const int32_t* __restrict__ idx, const uint64_t row0, const uint64_t n_rows_src,
                                              const int B, const int D, const uint64_t seed, const uint64_t step,
                                              unsigned char* __restrict__ x, const uint64_t out_row0 = 0) {
  const int qpr = D >> 2;
  if (q >= (uint64_t)B * qpr) return;
  const int b = (int)(q / qpr), d4 = (int)(q - (uint64_t)b * qpr) << 2;
  uint64_t r = idx ? (uint64_t)idx[b] : row0 + b;
  r = r < n_rows_src ? r : n_rows_src - 1;
  const uint32_t w = *reinterpret_cast<const uint32_t*>(pixels + r * D + d4);
  const uint64_t qg = q + out_row0 * (uint64_t)qpr;          // position in the GLOBAL batch (data parallel: out_row0 = rank * B)
  uint32_t c[4] = {(uint32_t)qg, (uint32_t)(qg >> 32) | 0x40000000u, (uint32_t)step, (uint32_t)(step >> 32)};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  uint32_t o = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float img = (float)((w >> (8 * j)) & 0xff) / 255.f;               // IEEE division, as TF's RealDiv
    o |= (img < u01(c[j]) ? 1u : 0u) << (8 * j);
  }
  *reinterpret_cast<uint32_t*>(x + (uint64_t)b * D + d4) = o;
}

// ------------------------------------------------------------ aux blocks
constexpr int kMaxImgTasks = 32;
struct ImgTask {
  float* dst;
  const float* src;
  int ld, rows, cols, src_ld, trans;
};
struct Aux {
  int nblocks;          // extra workgroups after the GEMM tiles (0: none)
  int noise_blocks;     // the first noise_blocks of them run the Philox fill
  int ntasks;           // then one workgroup per image task
  float* eps;
  float* u;
  unsigned long long n_rows, row_base, seed, step;   // eps [n_rows][nL], u [n_rows][nK]; row_base: global index of row 0
  int nL, nK;
  unsigned long long* step_dev;   // [2]: [0] = completed steps, [1] = copy that the last kernel of the step reads
  unsigned* epoch_word;           // bumped once per step: tag of the in-launch hand-offs of mega_fwd_bwd
  ImgTask task[kMaxImgTasks];
};

__device__ __forceinline__ void aux_block(const Aux& ax, const int b) {
  if (b == 0 && threadIdx.x == 0 && ax.epoch_word) *ax.epoch_word += 1u;
  if (b < ax.noise_blocks) {
    const unsigned long long step = ax.step_dev ? ax.step_dev[0] : ax.step;
    noise_item((uint64_t)b * kThreads + threadIdx.x, ax.eps, ax.u, ax.n_rows, ax.nL, ax.nK, ax.row_base, ax.seed, step);
    if (b == 0 && threadIdx.x == 0 && ax.step_dev) ax.step_dev[1] = ax.step_dev[0];
  } else {
    const int t = b - ax.noise_blocks;
    if (t < ax.ntasks) {
      if (ax.task[t].trans) mat_fill<true>(ax.task[t].dst, ax.task[t].ld, ax.task[t].src, ax.task[t].rows, ax.task[t].cols, ax.task[t].src_ld, threadIdx.x);
      else mat_fill<false>(ax.task[t].dst, ax.task[t].ld, ax.task[t].src, ax.task[t].rows, ax.task[t].cols, ax.task[t].src_ld, threadIdx.x);
    }
  }
}

}  // namespace gmvae
