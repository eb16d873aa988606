// Row-local kernels of the ELBO step (everything that is not a GEMM).
// Each kernel cites the reference arithmetic it replaces.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm.hpp"

namespace gmvae {

constexpr float kLog2Pi = 1.8378770664093453f;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// eps ~ N(0,1) (Box-Muller), u ~ U[tiny,1).  One thread -> 4 values (aux.hpp: noise_item).
// Diagnostic (GMVAE_ICACHE_FLUSH=1 in gmvae_train_profile, tools/icache_cold.py): ~100 KB of straight-line code -- 12,288
// fused multiply-adds with distinct 32-bit literals -- run by one wave on every CU, so that the launch that follows starts with
// COLD instruction caches (64 KB shared by two CUs).  The difference of the following launch's in-kernel span with and without it
// is what a complete refetch of its code costs: the yardstick for the handful of misses a warm launch takes.
__global__ __launch_bounds__(64) void icache_flush(float* sink) {
  float x = (float)threadIdx.x * 1e-3f;
#pragma unroll
  for (int i = 0; i < 12288; ++i) x = fmaf(x, 1.0f + (float)(i + 1) * 1.1920929e-7f, 0.25f);
  if (x == 12345.678f) sink[blockIdx.x] = x;       // (never true: keeps the chain alive)
}
__global__ void noise_fill(float* eps, float* u, uint64_t rows, int L, int K, uint64_t row_base, uint64_t seed, uint64_t step,
                           const uint64_t* step_dev) {
  if (step_dev) step = *step_dev;
  noise_item((uint64_t)blockIdx.x * blockDim.x + threadIdx.x, eps, u, rows, L, K, row_base, seed, step);
}

// Dynamic binarisation of the input pipeline as its own launch (aux.hpp: binarize_quad): HBM-bound byte work,
// D + D bytes per row, one thread per 4 pixels.
__global__ void binarize_rows(const unsigned char* __restrict__ pixels, const int32_t* __restrict__ idx, uint64_t row0,
                              uint64_t n_rows_src, int B, int D, uint64_t seed, uint64_t step,
                              const uint64_t* step_dev, unsigned char* __restrict__ x, uint64_t out_row0) {
  if (step_dev) step = *step_dev;
  binarize_quad((uint64_t)blockIdx.x * blockDim.x + threadIdx.x, pixels, idx, row0, n_rows_src, B, D, seed, step, x, out_row0);
}

// ------------------------------------------------ row-panel layers with a TINY contraction (K <= 16)
// out[r][n] = act(bias[n] + addsrc[r / add_div][n] + sum_k A[r][k] W[k][n]) for up to two weight matrices sharing A -- the y rows
// of enc_gmm's first layer and the conditional prior's layer at the reference's K = 10 mixture components (gmvae.py:246-252):
// 10 multiply-adds per output.  As a grouped GEMM (K padded to a 32-deep round, a tile's load -> LDS -> MFMA -> epilogue trip
// for 0.1 GFLOP) the launch took 50 us at 51,200 rows; here a thread owns 4 columns of a row, the weights sit in LDS and the
// launch moves its 39 MB of outputs.
struct SmallKProb {
  const float *W, *bias, *addsrc;   // [K][N]; [N] or null; [R / add_div][ld_add] or null
  float* out;                       // [R][N]
  int N, relu, ld_add, add_div;
};
struct SmallKArgs {
  const float* A;                   // [R][K]
  int R, K, np;
  SmallKProb p[2];
};
constexpr int kSmallKMax = 16, kSmallKCols = 512;
// KC = the contraction length as a compile-time constant (the reference's 10: operands and weights of a trip in registers, no
// branch in the multiply-adds) or 0: any K <= kSmallKMax, weights read from LDS inside the loop.  (A first form with K a run-time
// bound on 16-fold unrolled register arrays took 224 registers + scratch and ran 29 us for 39 MB of outputs.)
template <int KC>
__global__ __launch_bounds__(256) void rows_small_k(const SmallKArgs a) {
  __shared__ __attribute__((aligned(16))) float Ws[kSmallKMax * kSmallKCols];
  const int K = KC ? KC : a.K, N0 = a.p[0].N, NT = N0 + (a.np > 1 ? a.p[1].N : 0), q = NT >> 2;
  {
    // (16-byte pieces, every thread's loads in flight together: element by element a block spent 7 us on dependent loads
    //  before its first row)
    constexpr int NJ = kSmallKMax * kSmallKCols / 4 / 256;
    float4 st[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int i = min((int)threadIdx.x + 256 * j, K * q - 1);      // (clamped: the store below is predicated)
      const int k = i / q, n = (i - k * q) * 4;
      const float* const src = n < N0 ? a.p[0].W + (long long)k * N0 + n : a.p[1].W + (long long)k * a.p[1].N + (n - N0);
      st[j] = *reinterpret_cast<const float4*>(src);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int i = (int)threadIdx.x + 256 * j;
      if (i < K * q) *reinterpret_cast<float4*>(Ws + 4 * i) = st[j];
    }
  }
  __syncthreads();
  // (32-bit index arithmetic: R * q < 2^31 is checked by the host; a thread walks rows with its column quad FIXED -- the grid's
  //  stride is a multiple of q: no division inside the loop -- and every per-problem field is selected once, by value: a
  //  per-lane choice between the two argument records would be served from a scratch copy)
  const int items = a.R * q, stride = (int)gridDim.x * 256;
  const int it = (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (it >= items) return;
  int r = it / q;
  const int n = (it - r * q) * 4, dr = stride / q;
  const bool first = n < N0;
  const int nn = first ? n : n - N0, N = first ? N0 : a.p[1].N;
  const float* const bias = first ? a.p[0].bias : a.p[1].bias;
  const float* const addsrc = first ? a.p[0].addsrc : a.p[1].addsrc;
  float* const out = first ? a.p[0].out : a.p[1].out;
  const int relu = first ? a.p[0].relu : a.p[1].relu, ld_add = first ? a.p[0].ld_add : a.p[1].ld_add;
  const int add_div = first ? a.p[0].add_div : a.p[1].add_div;
  const float4 b4 = bias ? *reinterpret_cast<const float4*>(bias + nn) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float* const wcol = Ws + n;
  if constexpr (KC > 0) {
    float4 w4[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) w4[k] = *reinterpret_cast<const float4*>(wcol + k * NT);
    // four rows per trip: their 4 x K operand loads (and the addend's) are in flight together
    for (; r < a.R; r += 4 * dr) {
      float y[4][KC];
      float4 s4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ru = min(r + u * dr, a.R - 1);
        const float* const ar = a.A + (long long)ru * KC;
#pragma unroll
        for (int k = 0; k < KC; ++k) y[u][k] = ar[k];
        s4[u] = addsrc ? *reinterpret_cast<const float4*>(addsrc + (long long)(ru / add_div) * ld_add + nn) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (r + u * dr >= a.R) break;
        float4 acc = make_float4(b4.x + s4[u].x, b4.y + s4[u].y, b4.z + s4[u].z, b4.w + s4[u].w);
#pragma unroll
        for (int k = 0; k < KC; ++k) {
          acc.x = fmaf(y[u][k], w4[k].x, acc.x); acc.y = fmaf(y[u][k], w4[k].y, acc.y);
          acc.z = fmaf(y[u][k], w4[k].z, acc.z); acc.w = fmaf(y[u][k], w4[k].w, acc.w);
        }
        if (relu) { acc.x = relu_nan(acc.x); acc.y = relu_nan(acc.y); acc.z = relu_nan(acc.z); acc.w = relu_nan(acc.w); }
        *reinterpret_cast<float4*>(out + (long long)(r + u * dr) * N + nn) = acc;
      }
    }
  } else {
    for (; r < a.R; r += dr) {
      float4 acc = b4;
      if (addsrc) {
        const float4 s4 = *reinterpret_cast<const float4*>(addsrc + (long long)(r / add_div) * ld_add + nn);
        acc.x += s4.x; acc.y += s4.y; acc.z += s4.z; acc.w += s4.w;
      }
      const float* const ar = a.A + (long long)r * K;
      for (int k = 0; k < K; ++k) {
        const float y = ar[k];
        const float4 w4 = *reinterpret_cast<const float4*>(wcol + k * NT);
        acc.x = fmaf(y, w4.x, acc.x); acc.y = fmaf(y, w4.y, acc.y); acc.z = fmaf(y, w4.z, acc.z); acc.w = fmaf(y, w4.w, acc.w);
      }
      if (relu) { acc.x = relu_nan(acc.x); acc.y = relu_nan(acc.y); acc.z = relu_nan(acc.z); acc.w = relu_nan(acc.w); }
      *reinterpret_cast<float4*>(out + (long long)r * N + nn) = acc;
    }
  }
}

// log-softmax normaliser of one row of K logits walked by a wave (any K): m = the maximum, l = log(sum_j exp(lg_j - m)) with the
// log1p form when one class holds the maximum (gemm.hpp cat_log_softmax's statement (1)); log pi_k = (lg_k - m) - l.
__device__ __forceinline__ void row_lse_parts(const float* __restrict__ lg, const int K, const int lane, float& m, float& l) {
  m = -INFINITY;
  for (int k = lane; k < K; k += 64) m = fmaxf(m, lg[k]);
  m = wave_max(m);
  float sa = 0.f, sb = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float e = expf(lg[k] - m);
    sa += e;
    sb += lg[k] == m ? 0.f : e;
  }
  sa = wave_sum(sa); sb = wave_sum(sb);
  l = (sa - sb < 1.5f) ? log1pf(sb) : logf(sa);
}
// ------------------------------------------------ q(y|x): Gumbel-softmax head
// RelaxedOneHotCategorical.sample (scripts/base.py:206-209, gmvae.py:240):
//   g = -log(-log u); y = softmax((logits + g)/T)
// and utils.entropy(logits, softmax(logits)) (scripts/utils.py:165-170,
// gmvae.py:262-263): nent_b = sum_k pi_k log pi_k.
// One wave per row r; lanes stride over K (K <= 64 is a single pass).
__global__ void y_head_fwd(const float* __restrict__ logits, const float* __restrict__ u, float* __restrict__ y,
                           float* __restrict__ nent, int R, int S, int K, float invT) {
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  if (K <= 16) {
    // the reference's K = 10: FOUR rows per wave, 16 lanes each (a wave per row kept 54 of 64 lanes idle and paid a row's
    // dependent load -> log -> max -> exp -> sum -> exp chain per 10 outputs: 17.5 us at 51,200 rows)
    const int sub = lane >> 4, k = lane & 15;
    for (long long r4 = ((long long)blockIdx.x * wpb + (threadIdx.x >> 6)) * 4; r4 < R; r4 += (long long)gridDim.x * wpb * 4) {
      const long long r = r4 + sub;
      const bool rv = r < R, kv = rv && k < K;
      const long long rc = rv ? r : R - 1;
      const int b = (int)(rc / S);
      const float lgk = k < K ? logits[(long long)b * K + k] : 0.f;
      const float a = kv ? (lgk + -logf(-logf(u[rc * K + k]))) * invT : -INFINITY;
      float mx = a;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 16));
      float se = kv ? expf(a - mx) : 0.f;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) se += __shfl_xor(se, o, 16);
      const float lse = mx + logf(se);
      if (kv) y[r * K + k] = expf(a - lse);
      // once per x (the row of its first sample): entropy of q(y|x)
      const float lga[1] = {k < K ? lgk : -INFINITY};
      float lpa[1];
      cat_log_softmax<Sub16, 1, true>(lga, lpa);   // log pi (gemm.hpp: accurate for a saturated q(y|x))
      const float lp = lpa[0];
      float ne = k < K ? expf(lp) * lp : 0.f;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) ne += __shfl_xor(ne, o, 16);
      if (rv && k == 0 && r == (long long)b * S) nent[b] = ne;
    }
    return;
  }
  for (int r = blockIdx.x * wpb + (threadIdx.x >> 6); r < R; r += gridDim.x * wpb) {
    const int b = r / S;
    const float* lg = logits + (long long)b * K;
    if (K <= 64) {
      // one pass: the perturbed logit is computed ONCE and kept (the three passes below re-evaluate -log(-log u) -- two
      // precise logarithms -- per pass: at the config-5 sizes, 1.6 M elements, that arithmetic was the kernel's 17 us)
      const bool kv = lane < K;
      const float a = kv ? (lg[lane] + -logf(-logf(u[(long long)r * K + lane]))) * invT : -INFINITY;
      const float mx = wave_max(a);
      const float se = wave_sum(kv ? expf(a - mx) : 0.f);
      const float lse = mx + logf(se);
      if (kv) y[(long long)r * K + lane] = expf(a - lse);
    } else {
    float mx = -INFINITY;
    for (int k = lane; k < K; k += 64) {
      const float g = -logf(-logf(u[(long long)r * K + k]));
      mx = fmaxf(mx, (lg[k] + g) * invT);
    }
    mx = wave_max(mx);
    float se = 0.f;
    for (int k = lane; k < K; k += 64) {
      const float g = -logf(-logf(u[(long long)r * K + k]));
      se += expf((lg[k] + g) * invT - mx);
    }
    se = wave_sum(se);
    const float lse = mx + logf(se);
    for (int k = lane; k < K; k += 64) {
      const float g = -logf(-logf(u[(long long)r * K + k]));
      y[(long long)r * K + k] = expf((lg[k] + g) * invT - lse);
    }
    }
    if (r == b * S) {       // once per x: entropy of q(y|x)
      float m2, l2;
      row_lse_parts(lg, K, lane, m2, l2);
      float ne = 0.f;
      for (int k = lane; k < K; k += 64) {
        const float lp = (lg[k] - m2) - l2;
        ne += expf(lp) * lp;
      }
      ne = wave_sum(ne);
      if (lane == 0) nent[b] = ne;
    }
  }
}

// Backward of the head above:
//   da = y*(dy - sum_k y dy);  dlogits_b = sum_s da/T + pi*(log pi - sum pi log pi)
// dy already carries the IWAE row weight; the entropy term has weight sum_s rw = 1.
// One workgroup of 8 waves per batch row: the waves share the S samples (at S = 50 a single wave per row walked 50
// dependent load + reduction rounds: 40 us at the config-5 sizes), partial sums meet in LDS.
__global__ __launch_bounds__(512) void y_head_bwd(const float* __restrict__ logits, const float* __restrict__ y,
                                                  const float* __restrict__ dy, const float* __restrict__ nent,
                                                  float* __restrict__ dlogits, int B, int S, int K, float invT) {
  __shared__ float red[8][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const float* lg = logits + (long long)b * K;
    float m2, l2;
    row_lse_parts(lg, K, lane, m2, l2);
    const float ne = nent[b];
    if (K <= 64 && S <= 64) {
      // One pass with every load in flight first: a wave's (up to) 8 samples were 8 dependent rounds of load -> three wave
      // reductions (at the config-5 sizes, 7 samples per wave: 18 us for 13 MB).  Same operations in the same order as the loop below.
      const bool kv = lane < K;
      float ys[8], ds[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int sidx = wave + 8 * j;
        const long long r = (long long)b * S + (sidx < S ? sidx : 0);
        ys[j] = (kv && sidx < S) ? y[r * K + lane] : 0.f;
        ds[j] = (kv && sidx < S) ? dy[r * K + lane] : 0.f;
      }
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (wave + 8 * j < S) {                    // (uniform per wave)
          const float ym = wave_max(kv ? fmaxf(0.f, ys[j]) : 0.f);
          const float c = wave_max((kv && ys[j] == ym) ? ds[j] : -INFINITY);
          const float dot = wave_sum(kv ? ys[j] * (ds[j] - c) : 0.f);
          if (kv) acc += ys[j] * ((ds[j] - c) - dot);
        }
      }
      red[wave][lane] = acc;
      __syncthreads();
      if (wave == 0 && kv) {
        float t = red[0][lane];
#pragma unroll
        for (int w = 1; w < 8; ++w) t += red[w][lane];
        const float lp = (lg[lane] - m2) - l2;
        dlogits[(long long)b * K + lane] = t * invT + expf(lp) * (lp - ne);
      }
      __syncthreads();
      continue;
    }
    for (int k0 = 0; k0 < K; k0 += 64) {
      const int k = k0 + lane;
      float acc = 0.f;
      for (int s = wave; s < S; s += 8) {
        const long long r = (long long)b * S + s;
        // y (dy - y . dy) shifted by dy at the largest y (gemm.hpp cat_softmax_bwd's statement (2))
        float ym = 0.f;
        for (int kk = lane; kk < K; kk += 64) ym = fmaxf(ym, y[r * K + kk]);
        ym = wave_max(ym);
        float c = -INFINITY;
        for (int kk = lane; kk < K; kk += 64) c = fmaxf(c, y[r * K + kk] == ym ? dy[r * K + kk] : -INFINITY);
        c = wave_max(c);
        float dot = 0.f;
        for (int kk = lane; kk < K; kk += 64) dot += y[r * K + kk] * (dy[r * K + kk] - c);
        dot = wave_sum(dot);
        if (k < K) acc += y[r * K + k] * ((dy[r * K + k] - c) - dot);
      }
      red[wave][lane] = acc;
      __syncthreads();
      if (wave == 0 && k < K) {
        float t = red[0][lane];                    // fixed order: the same bits whatever the timing
#pragma unroll
        for (int w = 1; w < 8; ++w) t += red[w][lane];
        const float lp = (lg[k] - m2) - l2;
        dlogits[(long long)b * K + k] = t * invT + expf(lp) * (lp - ne);
      }
      __syncthreads();
    }
  }
}

// ------------------------------------------------ q(z|.) head, prior terms
enum { PRIOR_STD = 0, PRIOR_GMP = 1, PRIOR_COND = 2 };

// ConditionalNormal.condition (scripts/base.py:66-72): mu | raw split,
// sigma = max(softplus(raw + c), sigma_min); MultivariateNormalDiag.sample
// z = mu + sigma*eps (gmvae.py:248, vae.py:171) and .log_prob for q(z|.) and,
// for PRIOR_COND, p(z|y) (gmvae.py:258); PRIOR_STD: N(0,I) (vae.py:247-250).
// One wave per row; lanes stride over L.
__global__ void z_head_fwd(const float* __restrict__ qp, int qp_div, const float* __restrict__ pp,
                           const float* __restrict__ eps, float* __restrict__ z, float* __restrict__ logq,
                           float* __restrict__ logp, int R, int L, int prior, float c, float smin) {
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  for (int r = blockIdx.x * wpb + (threadIdx.x >> 6); r < R; r += gridDim.x * wpb) {
    const float* q = qp + (long long)(r / qp_div) * 2 * L;
    float aq = 0.f, ap = 0.f;
    for (int l = lane; l < L; l += 64) {
      const float mu = q[l];
      const float sg = fmaxf(softplus_r(q[L + l] + c), smin);
      const float zz = mu + sg * eps[(long long)r * L + l];
      z[(long long)r * L + l] = zz;
      const float e = (zz - mu) * __builtin_amdgcn_rcpf(sg);        // from z, not eps (A7)
      aq += -0.5f * e * e - 0.5f * kLog2Pi - log_r(sg);
      if (prior == PRIOR_COND) {
        const float* p = pp + (long long)r * 2 * L;
        const float sp = fmaxf(softplus_r(p[L + l] + c), smin);
        const float t = (zz - p[l]) * __builtin_amdgcn_rcpf(sp);
        ap += -0.5f * t * t - 0.5f * kLog2Pi - log_r(sp);
      } else if (prior == PRIOR_STD) {
        ap += -0.5f * zz * zz - 0.5f * kLog2Pi;
      }
    }
    aq = wave_sum(aq);
    ap = wave_sum(ap);
    if (lane == 0) {
      logq[r] = aq;
      if (prior != PRIOR_GMP) logp[r] = ap;
    }
  }
}

// MixtureSameFamily(Categorical(mixture_logits), MVNDiag(loc, softplus(raw_scale_diag))).log_prob
// (scripts/vae.py:231-244, consumed at vae.py:181):
//   log p(z) = logsumexp_k [ log_softmax(mixture_logits)_k + sum_l logN(z_l; loc_kl, s_kl) ]
// The K x L component parameters are staged ONCE per workgroup in LDS as
// (loc, 1/s) with an odd row stride; lanes are (row-in-wave, component k), each
// lane walks l serially, and the K-way logsumexp is a wavefront-shuffle reduction.
// Also writes the responsibilities r_k used by the backward pass.
__global__ void mixture_logprob_lse(const float* __restrict__ z, const float* __restrict__ loc,
                                    const float* __restrict__ raw_scale, const float* __restrict__ mixlog,
                                    float* __restrict__ logp, float* __restrict__ resp, int R, int L, int K,
                                    int Kp /* pow2 >= K, <= 64 */) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int LD = L | 1;
  float* s_loc = sm;                    // [K][LD]
  float* s_inv = s_loc + K * LD;        // [K][LD]  1/s
  float* s_c = s_inv + K * LD;          // [Kp]     log w_k - sum_l log s_kl - L/2 log 2pi
  float* s_z = s_c + 64;                // [waves][rows_per_wave][L]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  for (int i = tid; i < K * L; i += blockDim.x) {
    const int k = i / L, l = i % L;
    const float s = softplusf_(raw_scale[i]);
    s_loc[k * LD + l] = loc[i];
    s_inv[k * LD + l] = 1.f / s;
  }
  __syncthreads();
  // per-component constants: sum_l -log s_kl, a wave per component with the lanes over l (one lane per component walking the L
  // logarithms serially was 1 us of this launch at K = 10, L = 64) ...
  for (int k = wave; k < K; k += nw) {
    float ls = 0.f;
    for (int l = lane; l < L; l += 64) ls += logf(s_inv[k * LD + l]);   // = -log s
    ls = wave_sum(ls);
    if (lane == 0) s_c[k] = ls;
  }
  __syncthreads();
  if (wave == 0) {                      // ... + log_softmax of the mixture logits
    float mx = -INFINITY;
    for (int k = lane; k < K; k += 64) mx = fmaxf(mx, mixlog[k]);
    mx = wave_max(mx);
    float se = 0.f;
    for (int k = lane; k < K; k += 64) se += expf(mixlog[k] - mx);
    se = wave_sum(se);
    const float lse = mx + logf(se);
    for (int k = lane; k < K; k += 64) s_c[k] = mixlog[k] - lse + s_c[k] - 0.5f * kLog2Pi * (float)L;
  }
  __syncthreads();
  const int rpw = 64 / Kp;              // rows handled by one wave at a time
  const int sub = lane / Kp, k = lane % Kp;
  float* zw = s_z + wave * rpw * L;
  for (int r0 = (blockIdx.x * nw + wave) * rpw; r0 < R; r0 += gridDim.x * nw * rpw) {
    for (int i = lane; i < rpw * L; i += 64) {
      const int rr = r0 + i / L;
      zw[i] = rr < R ? z[(long long)rr * L + i % L] : 0.f;
    }
    __builtin_amdgcn_wave_barrier();
    const int r = r0 + sub;
    float comp = -INFINITY;
    if (k < K) {
      float a = 0.f;
      const float* zl = zw + sub * L;
      for (int l = 0; l < L; ++l) {
        const float t = (zl[l] - s_loc[k * LD + l]) * s_inv[k * LD + l];
        a += t * t;
      }
      comp = s_c[k] - 0.5f * a;
    }
    float mx = comp;
    for (int o = Kp >> 1; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float se = (k < K) ? expf(comp - mx) : 0.f;
    for (int o = Kp >> 1; o > 0; o >>= 1) se += __shfl_xor(se, o, 64);
    const float lse = mx + logf(se);
    if (r < R) {
      if (k < K) resp[(long long)r * K + k] = expf(comp - lse);
      if (k == 0) logp[r] = lse;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- the same log-prob for ANY K and L (scripts/vae.py:231-244 puts no bound on mixture_components / latent_size): the
// component parameters do not have to fit LDS.  gmp_consts prepares inv[k][l] = 1 / softplus(raw_scale_diag) and
// cst[k] = log_softmax(mixture_logits)_k - sum_l log s_kl - L/2 log 2 pi (one workgroup per component);
// mixture_logprob_tiled gives a workgroup 16 rows, walks the components 16 at a time and the latent dimension 64 at a
// time through LDS tiles (thread = (row, component of the tile)), keeps a running (max, sum) pair per row for the K-way
// logsumexp (16-lane shuffles), parks the component terms in `resp` and turns them into responsibilities in a second
// sweep by the same threads.  Bound: HBM/L2 (every row tile re-reads loc and inv: 8 K L bytes per 16 rows).
__global__ __launch_bounds__(256) void gmp_consts(const float* __restrict__ raw_scale, const float* __restrict__ mixlog,
                                                  float* __restrict__ inv, float* __restrict__ cst, int L, int K) {
  __shared__ float red[3][4];
  const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float ls = 0.f, mx = -INFINITY;
  for (int l = tid; l < L; l += 256) {
    const float s = softplusf_(raw_scale[(long long)k * L + l]);
    inv[(long long)k * L + l] = 1.f / s;
    ls += logf(s);
  }
  for (int j = tid; j < K; j += 256) mx = fmaxf(mx, mixlog[j]);
  ls = wave_sum(ls);
  mx = wave_max(mx);
  if (lane == 0) { red[0][wave] = ls; red[1][wave] = mx; }
  __syncthreads();
  mx = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
  float se = 0.f;
  for (int j = tid; j < K; j += 256) se += expf(mixlog[j] - mx);
  se = wave_sum(se);
  if (lane == 0) red[2][wave] = se;
  __syncthreads();
  if (tid == 0) {
    const float lse = mx + logf(red[2][0] + red[2][1] + red[2][2] + red[2][3]);
    cst[k] = mixlog[k] - lse - (red[0][0] + red[0][1] + red[0][2] + red[0][3]) - 0.5f * kLog2Pi * (float)L;
  }
}

__global__ __launch_bounds__(256) void mixture_logprob_tiled(const float* __restrict__ z, const float* __restrict__ loc,
                                                             const float* __restrict__ inv, const float* __restrict__ cst,
                                                             float* __restrict__ logp, float* __restrict__ resp, int R, int L,
                                                             int K) {
  constexpr int RT = 16, KT = 16, LC = 64, LD = LC + 1;
  __shared__ float zt[RT * LD], lt[KT * LD], it[KT * LD];
  const int tid = threadIdx.x, rr = tid >> 4, kk = tid & 15;
  for (int r0 = blockIdx.x * RT; r0 < R; r0 += gridDim.x * RT) {
    const int r = r0 + rr;
    float M = -INFINITY, Ssum = 0.f;
    for (int k0 = 0; k0 < K; k0 += KT) {
      float a = 0.f;
      for (int l0 = 0; l0 < L; l0 += LC) {
        __syncthreads();
        for (int i = tid; i < RT * LC; i += 256) {
          const int row = i / LC, l = i % LC;
          const bool okl = l0 + l < L;
          zt[row * LD + l] = (okl && r0 + row < R) ? z[(long long)(r0 + row) * L + l0 + l] : 0.f;
          const bool okk = okl && k0 + row < K;
          lt[row * LD + l] = okk ? loc[(long long)(k0 + row) * L + l0 + l] : 0.f;
          it[row * LD + l] = okk ? inv[(long long)(k0 + row) * L + l0 + l] : 0.f;
        }
        __syncthreads();
        const int nl = min(LC, L - l0);
        for (int l = 0; l < nl; ++l) {
          const float t = (zt[rr * LD + l] - lt[kk * LD + l]) * it[kk * LD + l];
          a += t * t;
        }
      }
      const bool kv = k0 + kk < K;
      const float comp = kv ? cst[k0 + kk] - 0.5f * a : -INFINITY;
      if (kv && r < R) resp[(long long)r * K + k0 + kk] = comp;
      float mt = comp;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) mt = fmaxf(mt, __shfl_xor(mt, o, 64));
      const float Mn = fmaxf(M, mt);                 // (finite from the first tile on: component k0 is always valid)
      float e = kv ? expf(comp - Mn) : 0.f;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) e += __shfl_xor(e, o, 64);
      Ssum = Ssum * expf(M - Mn) + e;
      M = Mn;
    }
    const float lse = M + logf(Ssum);
    if (r < R) {
      if (kk == 0) logp[r] = lse;
      for (int k = kk; k < K; k += KT) resp[(long long)r * K + k] = expf(resp[(long long)r * K + k] - lse);
    }
  }
}

// Backward seeds at z, w = IWAE row weight (1 at S=1):
//   dmu_q = dz_dec + w*prior_term ; dsig_q = dmu_q*eps - w/sig_q ;
//   draw_q = dsig_q*sigmoid(raw_q + c)*[softplus > sigma_min]
//   PRIOR_COND: dmu_p = -w*t/sig_p ; dsig_p = w*(1-t^2)/sig_p ; draw_p likewise
//   prior_term: COND t/sig_p | STD z | GMP sum_k r_k (z-loc_k)/s_k^2
__global__ void z_head_bwd(const float* __restrict__ dz, const float* __restrict__ qp, int qp_div,
                           const float* __restrict__ pp, const float* __restrict__ eps,
                           const float* __restrict__ z, const float* __restrict__ rw,
                           const float* __restrict__ resp, const float* __restrict__ loc,
                           const float* __restrict__ raw_scale, float* __restrict__ dqp,
                           float* __restrict__ dpp, int R, int L, int K, int prior, float c, float smin) {
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  for (int r = blockIdx.x * wpb + (threadIdx.x >> 6); r < R; r += gridDim.x * wpb) {
    const float w = rw ? rw[r] : 1.f;
    const float* q = qp + (long long)(r / qp_div) * 2 * L;
    for (int l = lane; l < L; l += 64) {
      const float rawq = q[L + l] + c;
      const float spq = softplus_r(rawq);
      const float sg = fmaxf(spq, smin);
      const float zz = z[(long long)r * L + l];
      float pterm;
      if (prior == PRIOR_COND) {
        const float* p = pp + (long long)r * 2 * L;
        const float rawp = p[L + l] + c;
        const float spp = softplus_r(rawp);
        const float sp = fmaxf(spp, smin), isp = __builtin_amdgcn_rcpf(sp);
        const float t = (zz - p[l]) * isp;
        pterm = t * isp;
        dpp[(long long)r * 2 * L + l] = -w * pterm;
        dpp[(long long)r * 2 * L + L + l] = (spp > smin) ? w * (1.f - t * t) * isp * sigmoid_r(rawp) : 0.f;
      } else if (prior == PRIOR_STD) {
        pterm = zz;
      } else {
        pterm = 0.f;
        for (int k = 0; k < K; ++k) {
          const float s = softplusf_(raw_scale[k * L + l]);
          pterm += resp[(long long)r * K + k] * (zz - loc[k * L + l]) / (s * s);
        }
      }
      const float dmu = dz[(long long)r * L + l] + w * pterm;
      const float dsg = dmu * eps[(long long)r * L + l] - w * __builtin_amdgcn_rcpf(sg);
      dqp[(long long)r * 2 * L + l] = dmu;
      dqp[(long long)r * 2 * L + L + l] = (spq > smin) ? dsg * sigmoid_r(rawq) : 0.f;
    }
  }
}

// Gradients of the learned mixture prior (vae.py:233-244):
//   dloc_kl = -sum_r w r_k (z-loc)/s^2 ; ds_kl = sum_r w r_k (1-t^2)/s ;
//   dmix_k = -sum_r w (r_k - softmax(mixture_logits)_k)
// Each workgroup reduces a strip of rows into one partial [2*K*L + K].
__global__ void gmp_param_bwd(const float* __restrict__ z, const float* __restrict__ resp,
                              const float* __restrict__ rw, const float* __restrict__ loc,
                              const float* __restrict__ raw_scale, const float* __restrict__ mixlog,
                              float* __restrict__ partial, int R, int L, int K, int KLp) {
  // partial layout mirrors the padded flat layout: [loc KLp | raw_scale KLp | mixture_logits pad4(K)]
  const int KL = K * L;
  const int rows_per = (R + gridDim.x - 1) / gridDim.x;
  const int rb = blockIdx.x * rows_per, re = min(R, rb + rows_per);
  float* out = partial + (long long)blockIdx.x * (2 * KLp + (K + 3) / 4 * 4);
  for (int i = threadIdx.x; i < KL; i += blockDim.x) {
    const int k = i / L, l = i % L;
    const float s = softplusf_(raw_scale[i]), lc = loc[i];
    float a = 0.f, b = 0.f;
    for (int r = rb; r < re; ++r) {
      const float wr = (rw ? rw[r] : 1.f) * resp[(long long)r * K + k];
      const float t = (z[(long long)r * L + l] - lc) / s;
      a -= wr * t / s;
      b += wr * (1.f - t * t) / s;
    }
    out[i] = a;
    out[KLp + i] = b * sigmoidf_(raw_scale[i]);
  }
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    float mx = -INFINITY;
    for (int j = 0; j < K; ++j) mx = fmaxf(mx, mixlog[j]);
    float se = 0.f;
    for (int j = 0; j < K; ++j) se += expf(mixlog[j] - mx);
    const float wk = expf(mixlog[k] - mx) / se;
    float a = 0.f;
    for (int r = rb; r < re; ++r) a -= (rw ? rw[r] : 1.f) * (resp[(long long)r * K + k] - wk);
    out[2 * KLp + k] = a;
  }
}

// sum over the S samples of a row group: out[b,:] = sum_s in[b*S+s,:]
// (N % 4 = 0: a thread sums 4 columns with 16-byte loads, 10 samples' loads in flight -- 52 MB at the config-5 sizes moved at
//  2.6 TB/s with one 4-byte load per thread and sample in flight; same order of additions per column)
__global__ void sum_over_s(const float* __restrict__ in, float* __restrict__ out, int B, int S, int N) {
  if ((N & 3) == 0) {
    const int N4 = N >> 2;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * N4) return;
    const int b = (int)(i / N4), n = 4 * (int)(i - (long long)b * N4);
    const float* p = in + (long long)b * S * N + n;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    int s = 0;
    for (; s + 10 <= S; s += 10) {
      float4 v[10];
#pragma unroll
      for (int j = 0; j < 10; ++j) v[j] = *reinterpret_cast<const float4*>(p + (long long)(s + j) * N);
#pragma unroll
      for (int j = 0; j < 10; ++j) { a.x += v[j].x; a.y += v[j].y; a.z += v[j].z; a.w += v[j].w; }
    }
    for (; s < S; ++s) {
      const float4 v = *reinterpret_cast<const float4*>(p + (long long)s * N);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    *reinterpret_cast<float4*>(out + (long long)b * N + n) = a;
    return;
  }
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * N) return;
  const int b = i / N, n = i % N;
  float a = 0.f;
  for (int s = 0; s < S; ++s) a += in[((long long)b * S + s) * N + n];
  out[i] = a;
}

// ---------------------------------------------------------------- loss
// log w_r = log p(x|z) + log p(z|.) - log q(z|.) - nent_b  (gmvae.py:254-267,
// vae.py:177-185; A15 for S>1).  logpx = sum of the decoder epilogue partials.
// logw64 (S > 1): the same sum kept in fp64.  |log w| ~ D ln 2 (2100 at D = 3072) has an fp32 ulp of 2.4e-4, and the IWAE
// weights softmax_s(log w) -- hence every gradient -- inherit the ABSOLUTE error of log w; iwae_rows forms the weights from
// fp64 differences to the row group's maximum, so only the fp32 error of the per-tile partial sums (|part| ~ 90) is left.
__global__ void row_terms(const float* __restrict__ part, int nparts, const float* __restrict__ logq,
                          const float* __restrict__ logp, const float* __restrict__ nent, int S,
                          float* __restrict__ logpx, float* __restrict__ logw, float* __restrict__ terms4, int R,
                          double* __restrict__ logw64 = nullptr) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  double a64 = 0.0;
  for (int i = 0; i < nparts; ++i) a64 += (double)part[(long long)r * nparts + i];
  const float a = (float)a64;
  const float ne = nent ? nent[r / S] : 0.f;
  const double lw64 = a64 + (double)logp[r] - (double)logq[r] - (double)ne;
  const float lw = (float)lw64;
  if (logw64) logw64[r] = lw64;
  logpx[r] = a;
  logw[r] = lw;
  if (terms4) {
    terms4[4 * r + 0] = a;
    terms4[4 * r + 1] = logq[r];
    terms4[4 * r + 2] = logp[r];
    terms4[4 * r + 3] = lw;
  }
}

// IWAE groups (S > 1), one wavefront per batch row b: bound_b = logsumexp_s(log w) - log S, the normalised row weights
// rw = softmax_s(log w) and the group's sums of the nll / kl terms -> pb[b] = (bound, sum_s -log p(x|z), sum_s (log q - log p)).
// Fixed-order lane reductions: deterministic.  (loss_tail alone walked the S samples of a row in ONE thread: 111 us of
// its single workgroup at B = 512, S = 50.)
__global__ __launch_bounds__(256) void iwae_rows(const double* __restrict__ logw64, const float* __restrict__ logpx,
                                                 const float* __restrict__ logq, const float* __restrict__ logp,
                                                 float* __restrict__ rw, float* __restrict__ pb, int B, int S) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const long long r0 = (long long)b * S;
  double mx = -INFINITY;
  for (int s = lane; s < S; s += 64) mx = fmax(mx, logw64[r0 + s]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
  float se = 0.f, nl = 0.f, kl = 0.f;
  for (int s = lane; s < S; s += 64) {
    se += expf((float)(logw64[r0 + s] - mx));     // differences formed in fp64: no ulp(|log w|) left in the weights
    nl -= logpx[r0 + s];
    kl += logq[r0 + s] - logp[r0 + s];
  }
  se = wave_sum(se); nl = wave_sum(nl); kl = wave_sum(kl);
  const float lrel = logf(se);
  if (rw)
    for (int s = lane; s < S; s += 64) rw[r0 + s] = expf((float)(logw64[r0 + s] - mx) - lrel);
  if (lane == 0) { pb[4 * b] = (float)(mx + (double)lrel - (double)logf((float)S)); pb[4 * b + 1] = nl; pb[4 * b + 2] = kl; pb[4 * b + 3] = 0.f; }
}

// row_terms + iwae_rows in one launch for S <= 64 (a wave per batch row, lane s owns sample row b S + s: its Bernoulli partials,
// log w in fp64 kept in a register instead of a round trip through memory) -- the same operations in the same order as the two
// kernels above, so the same bits; one launch less in every IWAE step and evaluation pass.
__global__ __launch_bounds__(256) void iwae_rows_terms(const float* __restrict__ part, int nparts, const float* __restrict__ logq,
                                                       const float* __restrict__ logp, const float* __restrict__ nent,
                                                       float* __restrict__ logpx, float* __restrict__ logw, float* __restrict__ terms4,
                                                       float* __restrict__ rw, float* __restrict__ pb, int B, int S) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const long long r = (long long)b * S + lane;
  const bool sv = lane < S;
  double lw64 = -INFINITY;
  float a = 0.f, lq = 0.f, lp = 0.f;
  if (sv) {
    double a64 = 0.0;
    for (int i = 0; i < nparts; ++i) a64 += (double)part[r * nparts + i];
    a = (float)a64;
    lq = logq[r]; lp = logp[r];
    const float ne = nent ? nent[b] : 0.f;
    lw64 = a64 + (double)lp - (double)lq - (double)ne;
    const float lw = (float)lw64;
    logpx[r] = a;
    logw[r] = lw;
    if (terms4) {
      terms4[4 * r + 0] = a;
      terms4[4 * r + 1] = lq;
      terms4[4 * r + 2] = lp;
      terms4[4 * r + 3] = lw;
    }
  }
  double mx = lw64;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
  float se = 0.f, nl = 0.f, kl = 0.f;
  if (sv) {
    se += expf((float)(lw64 - mx));
    nl -= a;
    kl += lq - lp;
  }
  se = wave_sum(se); nl = wave_sum(nl); kl = wave_sum(kl);
  const float lrel = logf(se);
  if (rw && sv) rw[r] = expf((float)(lw64 - mx) - lrel);
  if (lane == 0) { pb[4 * b] = (float)(mx + (double)lrel - (double)logf((float)S)); pb[4 * b + 1] = nl; pb[4 * b + 2] = kl; pb[4 * b + 3] = 0.f; }
}

// One workgroup: per-x IWAE bound logsumexp_s(log w) - log S, normalised row
// weights rw = softmax_s(log w), and the tail sums
// [sum_b loss_b, sum nll, sum kl, sum nent, B] (means over s inside a group).
// Deterministic (fixed-order tree).  Also advances the device step counter.
__global__ __launch_bounds__(1024) void loss_tail(const float* __restrict__ logw, const float* __restrict__ logpx,
                                                  const float* __restrict__ logq, const float* __restrict__ logp,
                                                  const float* __restrict__ nent, float* __restrict__ rw,
                                                  float* __restrict__ tail, int B, int S, uint64_t* step_dev,
                                                  const float* __restrict__ pb = nullptr) {
  __shared__ float red[4][1024];
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const float invS = 1.f / (float)S;
  if (pb) {                                      // S > 1: the groups were reduced by iwae_rows
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
      a0 -= pb[4 * b];
      a1 += pb[4 * b + 1] * invS;
      a2 += pb[4 * b + 2] * invS;
      a3 += nent ? nent[b] : 0.f;
    }
  } else
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const long long r0 = (long long)b * S;
    float mx = -INFINITY;
    for (int s = 0; s < S; ++s) mx = fmaxf(mx, logw[r0 + s]);
    float se = 0.f;
    for (int s = 0; s < S; ++s) se += expf(logw[r0 + s] - mx);
    const float lse = mx + logf(se);
    float nl = 0.f, kl = 0.f;
    for (int s = 0; s < S; ++s) {
      if (rw) rw[r0 + s] = expf(logw[r0 + s] - lse);
      nl -= logpx[r0 + s];
      kl += logq[r0 + s] - logp[r0 + s];
    }
    a0 -= (S == 1) ? logw[r0] : lse - logf((float)S);
    a1 += nl * invS;
    a2 += kl * invS;
    a3 += nent ? nent[b] : 0.f;
  }
  red[0][threadIdx.x] = a0; red[1][threadIdx.x] = a1; red[2][threadIdx.x] = a2; red[3][threadIdx.x] = a3;
  __syncthreads();
  for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
      for (int j = 0; j < 4; ++j) red[j][threadIdx.x] += red[j][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    tail[0] = red[0][0]; tail[1] = red[1][0]; tail[2] = red[2][0]; tail[3] = red[3][0];
    tail[4] = (float)B; tail[5] = 0.f; tail[6] = 0.f; tail[7] = 0.f;
    if (step_dev) *step_dev += 1;
  }
}

// ------------------------------------------------------- gradient assembly
// sum of NS split-K slabs at one float4 location
// (loads go out four at a time, clamped to the last slab and predicated at the add: a runtime-count loop of
// load-then-add serialises one cold memory round trip per slab)
__device__ __forceinline__ float4 slab_sum4(const float* __restrict__ p, const long long sstride, const int NS) {
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s0 = 0; s0 < NS; s0 += 4) {
    float4 o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = *reinterpret_cast<const float4*>(p + (long long)min(s0 + j, NS - 1) * sstride);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float w = s0 + j < NS ? 1.f : 0.f;
      r.x += w * o[j].x; r.y += w * o[j].y; r.z += w * o[j].z; r.w += w * o[j].w;
    }
  }
  return r;
}

// ---------------------------------------------------------------- Adam
// TF1 AdamOptimizer / ApplyAdam (scripts/runners.py:181-183):
//   lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; theta -= lr_t*m/(sqrt(v)+eps)
// One launch over the whole flat buffer (multi-tensor by construction).
// ONE statement of the update for every kernel that applies it (adam_tf, finalize_adam's two thread maps, adam_tf_img):
// the roundings are pinned with explicit intrinsics, so the same inputs give the same bits whichever kernel ran
// (the compiler contracts a*b+c differently from one kernel to the next otherwise).
__device__ __forceinline__ void adam_update(float& p, float& m, float& v, const float g, const float gscale, const float lr_t,
                                            const float omb1, const float omb2, const float eps) {
  const float gj = __fmul_rn(g, gscale);
  m = __fmaf_rn(__fsub_rn(gj, m), omb1, m);
  v = __fmaf_rn(__fmaf_rn(gj, gj, -v), omb2, v);
  p = __fsub_rn(p, __fdiv_rn(__fmul_rn(m, lr_t), __fadd_rn(__fsqrt_rn(v), eps)));
}

// Parameter ranges whose weight gradients were split over fewer slabs than the rest (the uint8-activation
// problems of the fused dW launch run on the bf16 matrix cores and take half the splits of the fp32 ones).
constexpr int kSlabRanges = 16;
struct SlabX {
  int n;                      // ranges in use
  int b[kSlabRanges], e[kSlabRanges];   // [begin, end) flat parameter indices, multiples of 4
  int ns[kSlabRanges];        // slabs written for that range (fewer OR more than the launch's default)
};
__device__ __forceinline__ int slab_count(const SlabX& sx, const long long i4, const int dflt) {
  int ns = dflt;
#pragma unroll
  for (int k = 0; k < kSlabRanges; ++k)
    if (k < sx.n && i4 >= sx.b[k] && i4 < sx.e[k]) ns = sx.ns[k];
  return ns;
}

// grads[i] = sum over the split-K slabs (fixed order => bit-reproducible),
// and the mixture-prior partials where present.
// With `ad.p` set the TF-Adam update follows in the same launch (general schedule inside a train graph: the step's
// loss tail and counter are already final -- loss_tail ran before the backward launches -- so t = *t_dev, the scale is
// 1 / tail[4] and a non-finite loss sum skips the update; block 0 also copies the tail to this step's log slot).
struct AdamTail {
  float *p, *m, *v;
  float lr, b1, b2, eps;
  const uint64_t* t_dev;
  float* tail_log;
};
__global__ void finalize_grads(const float* __restrict__ slabs, int nslab, long long P, float* __restrict__ grads,
                               const float* __restrict__ gmp_part, int gmp_n, int gmp_len, long long gmp_off,
                               const SlabX sx, const AdamTail ad) {
  const long long i4 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (ad.p && ad.tail_log && blockIdx.x == 0 && threadIdx.x < 8) ad.tail_log[threadIdx.x] = grads[P + threadIdx.x];
  if (i4 >= P) return;
  float4 a = slab_sum4(slabs + i4, P, slab_count(sx, i4, nslab));
  if (gmp_part && i4 >= gmp_off && i4 < gmp_off + gmp_len) {
    // mixture-prior gradients: sum the per-workgroup partials (rows of gmp_len floats, 16-byte aligned);
    // independent 16-byte loads, 8 in flight
    const float* p0 = gmp_part + (i4 - gmp_off);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int g = 0; g < gmp_n; ++g) {
      const float4 o = *reinterpret_cast<const float4*>(p0 + (long long)g * gmp_len);
      acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
    }
    a = acc;
  }
  *reinterpret_cast<float4*>(grads + i4) = a;
  if (ad.p) {
    if (!__builtin_isfinite(grads[P])) return;       // poisoned step: keep params, m, v
    const unsigned long long t = *ad.t_dev;
    const float gscale = 1.f / grads[P + 4];
    const float lr_t = (float)((double)ad.lr * sqrt(1.0 - pow((double)ad.b2, (double)t)) / (1.0 - pow((double)ad.b1, (double)t)));
    const float omb1 = 1.f - ad.b1, omb2 = 1.f - ad.b2;
    float4 pp = *reinterpret_cast<float4*>(ad.p + i4), mm = *reinterpret_cast<float4*>(ad.m + i4), vv = *reinterpret_cast<float4*>(ad.v + i4);
    float pa[4] = {pp.x, pp.y, pp.z, pp.w}, ma[4] = {mm.x, mm.y, mm.z, mm.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
    const float ga[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) adam_update(pa[j], ma[j], va[j], ga[j], gscale, lr_t, omb1, omb2, ad.eps);
    *reinterpret_cast<float4*>(ad.p + i4) = make_float4(pa[0], pa[1], pa[2], pa[3]);
    *reinterpret_cast<float4*>(ad.m + i4) = make_float4(ma[0], ma[1], ma[2], ma[3]);
    *reinterpret_cast<float4*>(ad.v + i4) = make_float4(va[0], va[1], va[2], va[3]);
  }
}

__global__ void adam_tf(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                        const float* __restrict__ g, long long P, float lr, float b1, float b2, float eps,
                        uint64_t t, const uint64_t* t_dev, float gscale, const float* gscale_dev,
                        const float* loss_sum_dev) {
  // a poisoned step (hand-off timeout: NaN loss sum, here or on any rank of the all-reduce) must not reach the state
  if (loss_sum_dev && !__builtin_isfinite(*loss_sum_dev)) return;
  if (t_dev) t = *t_dev;
  if (gscale_dev) gscale = 1.f / *gscale_dev;
  // TF's ApplyAdam functor, in its own fp32 form (tensorflow/core/kernels/training_ops.cc):
  //   alpha = lr*sqrt(1-b2^t)/(1-b1^t); m += (g-m)*(1-b1); v += (g*g-v)*(1-b2); var -= m*alpha/(sqrt(v)+eps)
  const float lr_t = (float)((double)lr * sqrt(1.0 - pow((double)b2, (double)t)) / (1.0 - pow((double)b1, (double)t)));
  const float omb1 = 1.f - b1, omb2 = 1.f - b2;
  const long long i4 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 + 3 < P) {
    float4 pp = *reinterpret_cast<float4*>(p + i4), mm = *reinterpret_cast<float4*>(m + i4),
           vv = *reinterpret_cast<float4*>(v + i4);
    const float4 gg = *reinterpret_cast<const float4*>(g + i4);
    float pa[4] = {pp.x, pp.y, pp.z, pp.w}, ma[4] = {mm.x, mm.y, mm.z, mm.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
    const float ga[4] = {gg.x, gg.y, gg.z, gg.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) adam_update(pa[j], ma[j], va[j], ga[j], gscale, lr_t, omb1, omb2, eps);
    *reinterpret_cast<float4*>(p + i4) = make_float4(pa[0], pa[1], pa[2], pa[3]);
    *reinterpret_cast<float4*>(m + i4) = make_float4(ma[0], ma[1], ma[2], ma[3]);
    *reinterpret_cast<float4*>(v + i4) = make_float4(va[0], va[1], va[2], va[3]);
  } else {
    for (long long i = i4; i < P; ++i) adam_update(p[i], m[i], v[i], g[i], gscale, lr_t, omb1, omb2, eps);
  }
}

// ----------------------------------------------- fused end-of-step kernel
// One launch for the three tail kernels of the step (each costs ~4.7 us of timeline on its own):
//   blocks [0, nb)  : gradient = fixed-order sum of the split-K slabs (-> grads), then TF-Adam on it
//   block  nb       : loss tail sums (loss_tail) and the device step counter.
// Counter protocol (no intra-kernel race): every block reads t-1 from step_dev[1] (copied from
// step_dev[0] by the FIRST launch of the step, aux.hpp); only the tail block writes step_dev[0].
// One tensor (or row range of one) that also lives in an LDS-image of the next step's mega_fwd_bwd: flat parameter
// indices [begin, end), `cols` per source row.  kind 0: img[base + r * ld + c]; kind 1 (decoder output layer, stored
// as column chunks of `cw`): img[base + (c / cw) * chunk + r * ld + c % cw].
// Kinds 2..6 are the operand images of mega2_fwd_bwd (mega2.hpp): a matrix product's weight is the MFMA A operand
// there, stored [contraction / 4][outputs padded to `ld`][contraction % 4], so that a lane's four consecutive
// contraction steps are ONE 16-byte read:
//   kind 2: contraction index = r (forward: out = in * W):     img[base + ((r >> 2) * ld + c) * 4 + (r & 3)]
//   kind 3: contraction index = c (backward: din = dout * W^T): img[base + ((c >> 2) * ld + r) * 4 + (c & 3)]
//   kinds 4 / 5 / 6: the decoder output layer [H][D] (4: forward, 5: data gradient, 6: its bias), 16-column tiles
//   dealt to the panel's 4 workgroups by m2_dec_part: tile t = c >> 4 belongs to workgroup q as its local tile lt;
//   lc = 16 lt + (c & 15) is the column inside that workgroup's part (`chunk` floats per part).
//   kinds 7 / 8 / 9: the same three images for mega2v_fwd_bwd's SEVEN workgroups per panel (mega2v.hpp): tile t belongs to
//   workgroup t % 7 as its local tile t / 7.
// Which workgroup of a panel owns decoder column tile t, and as which local tile (D = 784: 49 tiles).  The three producer
// quarters (1..3) take 14 / 14 / 13 tiles round-robin, the lead quarter (0) the last 8: ONE tile per wave.  The lead is
// the launch's critical path (it also stores the panel's activations and runs the backward chain), and its hand-off poll
// can only succeed ~1 us after the producers publish: with 13 / 12 / 12 / 12 tiles it finished its decoder stage 1.6 us
// AFTER them (tools/handoff_clock.py); two-deep waves cost the producers nothing extra up to 16 tiles.
// (M2_LEAD_TILES = 1, mega3_step: the producers own 16 tiles each -- still two per wave -- and the lead ONE, so that the
//  decoder layer's weight gradient depends on the producers alone and its tiles can start while the leads run the backward
//  chain; mega3.hpp)
#ifndef M2_LEAD_TILES
#define M2_LEAD_TILES 1
#endif
constexpr int kM2Tiles = 49, kM2LeadTiles = M2_LEAD_TILES, kM2ProdTiles = (kM2Tiles - kM2LeadTiles + 2) / 3;
static_assert(kM2LeadTiles >= 1 && kM2LeadTiles <= 8 && kM2ProdTiles <= 16, "one tile per lead wave, at most two per producer wave");
__host__ __device__ inline void m2_dec_part(const int t, int& q, int& lt) {
  if (t < kM2Tiles - kM2LeadTiles) { q = 1 + t % 3; lt = t / 3; }
  else { q = 0; lt = t - (kM2Tiles - kM2LeadTiles); }
}
__host__ __device__ inline int m2_dec_tile(const int q, const int lt) { return q == 0 ? kM2Tiles - kM2LeadTiles + lt : 3 * lt + (q - 1); }
__host__ __device__ inline int m2_dec_ntiles(const int q) { return q == 0 ? kM2LeadTiles : (kM2Tiles - kM2LeadTiles - (q - 1) + 2) / 3; }
__host__ __device__ inline int img_dst(const int kind, const int base, const int ld, const int chunk, const int r, const int c) {
  switch (kind) {
    case 0: return base + r * ld + c;
    case 1: return base + (c >> 7) * chunk + r * ld + (c & 127);                  // kCW = 128
    case 2: return base + (((r >> 2) * ld + c) << 2) + (r & 3);
    case 3: return base + (((c >> 2) * ld + r) << 2) + (c & 3);
    default: {
      int q, lt;
      if (kind >= 7) { q = (c >> 4) % 7; lt = (c >> 4) / 7; }
      else m2_dec_part(c >> 4, q, lt);
      const int lc = (lt << 4) | (c & 15);
      if (kind == 4 || kind == 7) return base + q * chunk + (((r >> 2) * ld + lc) << 2) + (r & 3);    // [H/4][ld = DC][4]
      if (kind == 5 || kind == 8) return base + q * chunk + (((lc >> 2) * ld + r) << 2) + (lc & 3);   // [DC/4][ld = H][4]
      return base + q * chunk + lc;                                                       // bias [DC]
    }
  }
}
struct ImgMap {
  int begin, end, cols, kind;
  int base, ld, cw, chunk;
  unsigned magic;             // floor(2^32 / cols) + 1: row = (off * magic) >> 32 (checked on the host for the range)
  int which;                  // destination buffer: 0 small-weight image, 1 decoder chunk images; mega2: 2 forward
                              // image, 3 backward image, 4 decoder operand images
};
constexpr int kMaxImgMap = 24;
constexpr int kImgBufs = 5;

struct FinalArgs {
  const float* slabs; int nslab; long long P;
  float *grads, *p, *m, *v;
  float lr, b1, b2, eps;
  int do_adam; float count;            // count = number of rows on this device (single-device scale)
  const float *logw, *logpx, *logq, *logp, *nent;
  float* tail; int B;
  float* tail_log;            // (may be null) a second copy of the tail: slot of this step in a train graph's per-step log
  unsigned long long* step_dev;
  // the updated parameters are also scattered into the next step's weight images (then that step needs no
  // image-building launch); nmap = 0: off
  int nmap, map_lo, map_hi;
  float* img[kImgBufs];       // destination buffers (ImgMap::which)
  unsigned* epoch_word;       // bumped for the next step's in-launch hand-offs
  const unsigned* err_word;   // hand-off timeout flag of this workspace: when set the step is poisoned and NOT applied
  unsigned long long* span;   // measurement: [block][2] wall-clock (100 MHz) at a block's first and last instruction
  // VAE_GMP: the prior variables' gradients are per-workgroup partials of mega_fwd_bwd, not split-K slabs
  const float* gmp_part; int gmp_n, gmp_len; long long gmp_off;
  // "quad" blocks (after the tail block): ONE tensor [q_rows][q_cols] whose image stores the 4 values of 4
  // consecutive ROWS together (img_dst kind 4: the decoder output layer's forward operand) is updated by threads that
  // own (4 rows, 1 column) each: their parameter / moment / slab accesses are 4-byte but lane-contiguous, and the
  // image leaves as ONE 16-byte store per thread, consecutive lanes writing consecutive units.  (Scattered from the
  // row-major thread map these were 4-byte stores 16 bytes apart whose units four different waves completed:
  // +2.8 us on this launch, profiles/round2_notes.md.)  The regular blocks skip that tensor's range.
  int quad_blocks, q_begin, q_end, q_rows, q_cols, q_base, q_ld, q_chunk, q_which;
  int q2_kind, q2_base, q2_ld, q2_chunk, q2_which;   // a second image of the same tensor (kind 0 = none): 4-byte stores
  SlabX sx;
  int mbegin[kMaxImgMap], mend[kMaxImgMap];   // the ranges again, adjacent: one round of scalar loads finds the entry
  ImgMap map[kMaxImgMap];
};
// the loss tail (batch sums of the per-row terms), the device step counter and the hand-off tag: one workgroup
// (256 threads of it) of the step's last launch
__device__ __forceinline__ void finalize_tail_block(const FinalArgs& a, float (*red)[256]) {
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (threadIdx.x < 256) {
    for (int b = threadIdx.x; b < a.B; b += 256) {
      a0 -= a.logw[b];
      a1 -= a.logpx[b];
      a2 += a.logq[b] - a.logp[b];
      a3 += a.nent ? a.nent[b] : 0.f;
    }
    red[0][threadIdx.x] = a0; red[1][threadIdx.x] = a1; red[2][threadIdx.x] = a2; red[3][threadIdx.x] = a3;
  }
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
      for (int j = 0; j < 4; ++j) red[j][threadIdx.x] += red[j][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    a.tail[0] = red[0][0]; a.tail[1] = red[1][0]; a.tail[2] = red[2][0]; a.tail[3] = red[3][0];
    a.tail[4] = (float)a.B; a.tail[5] = 0.f; a.tail[6] = 0.f; a.tail[7] = 0.f;
    if (a.tail_log) {
      a.tail_log[0] = red[0][0]; a.tail_log[1] = red[1][0]; a.tail_log[2] = red[2][0]; a.tail_log[3] = red[3][0];
      a.tail_log[4] = (float)a.B; a.tail_log[5] = 0.f; a.tail_log[6] = 0.f; a.tail_log[7] = 0.f;
    }
    if (a.step_dev) a.step_dev[0] = a.step_dev[1] + 1;
    if (a.epoch_word) *a.epoch_word += 1u;
  }
}
__global__ __launch_bounds__(256) void finalize_adam(const FinalArgs a) {
#define GMVAE_FIN_END() if (a.span && threadIdx.x == 0) a.span[2 * blockIdx.x + 1] = wall_clock64()
  if (a.span && threadIdx.x == 0) a.span[2 * blockIdx.x] = wall_clock64();
  const int nb = gridDim.x - 1 - a.quad_blocks;
  if ((int)blockIdx.x > nb) {                    // quad blocks: (4 rows, 1 column) of the q_ tensor per thread
    const int gq = ((int)blockIdx.x - nb - 1) * 256 + (int)threadIdx.x;
    if (gq >= (a.q_rows >> 2) * a.q_cols) { GMVAE_FIN_END(); return; }
    const int rq = gq / a.q_cols, c = gq - rq * a.q_cols;
    const long long i0 = (long long)a.q_begin + (long long)(4 * rq) * a.q_cols + c;
    float pa[4], ma[4], va[4], ga[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) { pa[j] = a.p[i0 + (long long)j * a.q_cols]; ma[j] = a.m[i0 + (long long)j * a.q_cols]; va[j] = a.v[i0 + (long long)j * a.q_cols]; }
    const int nsl = slab_count(a.sx, (long long)a.q_begin, a.nslab);
    for (int s0 = 0; s0 < nsl; s0 += 8) {          // 8 slabs x 4 rows of loads in flight; summed in slab order (the regular blocks' order)
      float o[8][4];
#pragma unroll
      for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) o[k][j] = a.slabs[(long long)min(s0 + k, nsl - 1) * a.P + i0 + (long long)j * a.q_cols];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float w = s0 + k < nsl ? 1.f : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) ga[j] += w * o[k][j];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) a.grads[i0 + (long long)j * a.q_cols] = ga[j];
    if (a.err_word && *a.err_word) { GMVAE_FIN_END(); return; }
    const unsigned long long t = (a.step_dev ? a.step_dev[1] : 0ull) + 1ull;
    const float lr_t = (float)((double)a.lr * sqrt(1.0 - pow((double)a.b2, (double)t)) / (1.0 - pow((double)a.b1, (double)t)));
    const float omb1 = 1.f - a.b1, omb2 = 1.f - a.b2, gs = 1.f / a.count;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      adam_update(pa[j], ma[j], va[j], ga[j], gs, lr_t, omb1, omb2, a.eps);
      a.p[i0 + (long long)j * a.q_cols] = pa[j];
      a.m[i0 + (long long)j * a.q_cols] = ma[j];
      a.v[i0 + (long long)j * a.q_cols] = va[j];
    }
    *reinterpret_cast<float4*>(a.img[a.q_which] + img_dst(4, a.q_base, a.q_ld, a.q_chunk, 4 * rq, c)) = make_float4(pa[0], pa[1], pa[2], pa[3]);
    if (a.q2_kind) {
      float* const im2 = a.img[a.q2_which];
#pragma unroll
      for (int j = 0; j < 4; ++j) im2[img_dst(a.q2_kind, a.q2_base, a.q2_ld, a.q2_chunk, 4 * rq + j, c)] = pa[j];
    }
    GMVAE_FIN_END();
    return;
  }
  if ((int)blockIdx.x == nb) {
    __shared__ float red[4][256];
    finalize_tail_block(a, red);
    GMVAE_FIN_END();
    return;
  }
  const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  // Image scatter, part 1 (before the slab loads, so that its scalar loads ride under them).  A block covers 1024
  // consecutive parameters: nearly always inside ONE map entry, so the entries are found with block-uniform tests
  // and their fields stay scalar (a per-lane entry index would turn every field read into a waterfall).
  const int blo = (int)blockIdx.x * 1024, bhi = blo + 1024;
  unsigned hit = 0;                                           // entries this block overlaps
  if (a.nmap > 0 && bhi > a.map_lo && blo < a.map_hi) {
#pragma unroll
    for (int k = 0; k < kMaxImgMap; ++k)
      if (k < a.nmap && bhi > a.mbegin[k] && blo < a.mend[k]) hit |= 1u << k;
  }
  const int k0 = hit ? __builtin_ctz(hit) : 0;
  int e_begin = a.map[k0].begin, e_end = a.map[k0].end, e_cols = a.map[k0].cols, e_kind = a.map[k0].kind,
      e_base = a.map[k0].base, e_ld = a.map[k0].ld, e_chunk = a.map[k0].chunk, e_which = a.map[k0].which;
  unsigned e_magic = a.map[k0].magic;
  asm volatile("" ::"s"(hit), "s"(e_begin), "s"(e_end), "s"(e_cols), "s"(e_kind), "s"(e_base), "s"(e_ld), "s"(e_chunk),
               "s"(e_which), "s"(e_magic));
  if (i4 >= a.P || (a.quad_blocks && i4 >= a.q_begin && i4 < a.q_end)) { GMVAE_FIN_END(); return; }
  // every load of the block goes out before the first wait: parameters and moments, then up to 8 slabs at once
  // (the launch is one wave of blocks: its length is its chain of dependent memory round trips)
  float4 pp = make_float4(0.f, 0.f, 0.f, 0.f), mm = pp, vv = pp;
  if (a.do_adam) {
    pp = *reinterpret_cast<const float4*>(a.p + i4);
    mm = *reinterpret_cast<const float4*>(a.m + i4);
    vv = *reinterpret_cast<const float4*>(a.v + i4);
  }
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  const int nsl = slab_count(a.sx, i4, a.nslab);
  {
    // up to 16 slabs go out in ONE batch (a second batch is a second memory round trip of the launch's critical path);
    // summed in slab order whatever the batching, so the gradient bits do not depend on it
    float4 o[16];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = *reinterpret_cast<const float4*>(a.slabs + (long long)min(j, nsl - 1) * a.P + i4);
    const bool more = nsl > 8;
    if (more) {
#pragma unroll
      for (int j = 8; j < 16; ++j) o[j] = *reinterpret_cast<const float4*>(a.slabs + (long long)min(j, nsl - 1) * a.P + i4);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float w = j < nsl ? 1.f : 0.f;
      g.x += w * o[j].x; g.y += w * o[j].y; g.z += w * o[j].z; g.w += w * o[j].w;
    }
    if (more) {
#pragma unroll
      for (int j = 8; j < 16; ++j) {
        const float w = j < nsl ? 1.f : 0.f;
        g.x += w * o[j].x; g.y += w * o[j].y; g.z += w * o[j].z; g.w += w * o[j].w;
      }
    }
    for (int s = 16; s < nsl; ++s) {               // (NS_MAX is 16 today)
      const float4 t = *reinterpret_cast<const float4*>(a.slabs + (long long)s * a.P + i4);
      g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w;
    }
  }
  if (a.gmp_part && i4 >= a.gmp_off && i4 < a.gmp_off + a.gmp_len) {
    const float* p0 = a.gmp_part + (i4 - a.gmp_off);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int k = 0; k < a.gmp_n; ++k) {
      const float4 o = *reinterpret_cast<const float4*>(p0 + (long long)k * a.gmp_len);
      acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
    }
    g = acc;
  }
  *reinterpret_cast<float4*>(a.grads + i4) = g;
  if (!a.do_adam || (a.err_word && *a.err_word)) { GMVAE_FIN_END(); return; }
  const unsigned long long t = (a.step_dev ? a.step_dev[1] : 0ull) + 1ull;
  const float lr_t = (float)((double)a.lr * sqrt(1.0 - pow((double)a.b2, (double)t)) / (1.0 - pow((double)a.b1, (double)t)));
  const float omb1 = 1.f - a.b1, omb2 = 1.f - a.b2, gs = 1.f / a.count;
  float pa[4] = {pp.x, pp.y, pp.z, pp.w}, ma[4] = {mm.x, mm.y, mm.z, mm.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
  const float ga[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) adam_update(pa[j], ma[j], va[j], ga[j], gs, lr_t, omb1, omb2, a.eps);
  *reinterpret_cast<float4*>(a.p + i4) = make_float4(pa[0], pa[1], pa[2], pa[3]);
  *reinterpret_cast<float4*>(a.m + i4) = make_float4(ma[0], ma[1], ma[2], ma[3]);
  *reinterpret_cast<float4*>(a.v + i4) = make_float4(va[0], va[1], va[2], va[3]);
  // Image scatter, part 2: the updated values go to their image positions.
  while (hit) {
    hit &= hit - 1;
    if (i4 >= e_begin && i4 < e_end) {
      float* const img = a.img[e_which];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int idx = (int)i4 + j;
        if (idx < e_end) {
          const unsigned off = (unsigned)(idx - e_begin);
          const int r = (int)(((unsigned long long)off * e_magic) >> 32);
          const int c = (int)off - r * e_cols;
          img[img_dst(e_kind, e_base, e_ld, e_chunk, r, c)] = pa[j];
        }
      }
    }
    if (hit) {                                                // a block on a tensor boundary: next entry
      const int k = __builtin_ctz(hit);
      e_begin = a.map[k].begin; e_end = a.map[k].end; e_cols = a.map[k].cols; e_kind = a.map[k].kind;
      e_base = a.map[k].base; e_ld = a.map[k].ld; e_chunk = a.map[k].chunk; e_which = a.map[k].which;
      e_magic = a.map[k].magic;
    }
  }
  GMVAE_FIN_END();
#undef GMVAE_FIN_END
}

// TF-Adam after the data-parallel all-reduce, with finalize_adam's image scatter: the updated parameters also go to
// their positions in the next step's weight images (so that step can run its first layer inside mega_fwd_bwd).
__host__ __device__ inline unsigned alpha_key(const float b1, const float b2) {      // tag of a cached alpha_t: the moments' rates
  return __builtin_bit_cast(unsigned, b1) * 2654435761u ^ __builtin_bit_cast(unsigned, b2);
}
struct ImgScatter {
  int nmap, lo, hi;
  float* img[kImgBufs];
  unsigned* epoch_word;
  const unsigned* lr_dev;     // {alpha_t bits, the Adam step t it is for, lr bits, alpha_key} left by mega3_step's tail slot (mega3.hpp), or null
  unsigned long long* span;   // measurement (gmvae_dp_profile): [0] = 2^62 - the earliest block start, [1] = the latest block end (100 MHz wall clock), or null
  int mbegin[kMaxImgMap], mend[kMaxImgMap];
  ImgMap map[kMaxImgMap];
};
__global__ __launch_bounds__(256) void adam_tf_img(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                   const float* __restrict__ g, long long P, float lr, float b1, float b2,
                                                   float eps, const uint64_t* t_dev, const float* gscale_dev,
                                                   const float* loss_sum_dev, float* tail_log, const ImgScatter sc) {
  if (sc.span && threadIdx.x == 0) atomicMax(sc.span, (1ull << 62) - wall_clock64());
  struct SpanEnd {                               // (every exit path of the block stamps its end)
    unsigned long long* p;
    __device__ ~SpanEnd() { if (p && threadIdx.x == 0) atomicMax(p + 1, wall_clock64()); }
  } span_end{sc.span};
  if (blockIdx.x == 0 && threadIdx.x == 0 && sc.epoch_word) *sc.epoch_word += 1u;
  if (blockIdx.x == 0 && threadIdx.x < 8 && tail_log && loss_sum_dev) tail_log[threadIdx.x] = loss_sum_dev[threadIdx.x];   // the all-reduced tail
  if (loss_sum_dev && !__builtin_isfinite(*loss_sum_dev)) return;   // poisoned step: keep params, m, v and the images
  const unsigned long long t = *t_dev;
  const float gscale = 1.f / *gscale_dev;
  float lr_t;
  if (sc.lr_dev && sc.lr_dev[1] == (unsigned)t && sc.lr_dev[2] == __float_as_uint(lr) && sc.lr_dev[3] == alpha_key(b1, b2))
    lr_t = __uint_as_float(sc.lr_dev[0]);                      // (the same fp64 form, computed once by mega3_step's tail slot)
  else lr_t = (float)((double)lr * sqrt(1.0 - pow((double)b2, (double)t)) / (1.0 - pow((double)b1, (double)t)));
  const float omb1 = 1.f - b1, omb2 = 1.f - b2;
  const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= P) return;                                        // P is the padded parameter count: a multiple of 4
  float4 pp = *reinterpret_cast<float4*>(p + i4), mm = *reinterpret_cast<float4*>(m + i4), vv = *reinterpret_cast<float4*>(v + i4);
  const float4 gg = *reinterpret_cast<const float4*>(g + i4);
  float pa[4] = {pp.x, pp.y, pp.z, pp.w}, ma[4] = {mm.x, mm.y, mm.z, mm.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
  const float ga[4] = {gg.x, gg.y, gg.z, gg.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) adam_update(pa[j], ma[j], va[j], ga[j], gscale, lr_t, omb1, omb2, eps);
  *reinterpret_cast<float4*>(p + i4) = make_float4(pa[0], pa[1], pa[2], pa[3]);
  *reinterpret_cast<float4*>(m + i4) = make_float4(ma[0], ma[1], ma[2], ma[3]);
  *reinterpret_cast<float4*>(v + i4) = make_float4(va[0], va[1], va[2], va[3]);
  const int blo = (int)blockIdx.x * 1024, bhi = blo + 1024;
  if (sc.nmap <= 0 || bhi <= sc.lo || blo >= sc.hi) return;
  unsigned hit = 0;                                           // entries this block overlaps (block-uniform)
#pragma unroll
  for (int k = 0; k < kMaxImgMap; ++k)
    if (k < sc.nmap && bhi > sc.mbegin[k] && blo < sc.mend[k]) hit |= 1u << k;
  while (hit) {
    const int k = __builtin_ctz(hit);
    hit &= hit - 1;
    const int begin = sc.map[k].begin, end = sc.map[k].end, cols = sc.map[k].cols, kind = sc.map[k].kind;
    const int base = sc.map[k].base, ld = sc.map[k].ld, chunk = sc.map[k].chunk;
    const unsigned magic = sc.map[k].magic;
    float* const img = sc.img[sc.map[k].which];
    if (i4 < begin || i4 >= end) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = (int)i4 + j;
      if (idx < end) {
        const unsigned off = (unsigned)(idx - begin);
        const int r = (int)(((unsigned long long)off * magic) >> 32);
        const int c = (int)off - r * cols;
        img[img_dst(kind, base, ld, chunk, r, c)] = pa[j];
      }
    }
  }
}

// The weight images of the in-launch schedule from the parameters as they are (no update): what the FIRST step of a
// train graph -- runs before mega2_fwd_bwd / mega_fwd_bwd<FLT = 1>, so that every step of a graph is
// the steady-state pair of launches (later steps find the images left by the optimizer's scatter).  Also opens the step:
// bumps the hand-off tag like the first launch of the general schedule.
__global__ __launch_bounds__(256) void img_build(const float* __restrict__ p, long long P, const ImgScatter sc) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && sc.epoch_word) *sc.epoch_word += 1u;
  const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  const int blo = (int)blockIdx.x * 1024, bhi = blo + 1024;
  if (i4 >= P || sc.nmap <= 0 || bhi <= sc.lo || blo >= sc.hi) return;
  const float4 pp = *reinterpret_cast<const float4*>(p + i4);
  const float pa[4] = {pp.x, pp.y, pp.z, pp.w};
  unsigned hit = 0;
#pragma unroll
  for (int k = 0; k < kMaxImgMap; ++k)
    if (k < sc.nmap && bhi > sc.mbegin[k] && blo < sc.mend[k]) hit |= 1u << k;
  while (hit) {
    const int k = __builtin_ctz(hit);
    hit &= hit - 1;
    const int begin = sc.map[k].begin, end = sc.map[k].end, cols = sc.map[k].cols, kind = sc.map[k].kind;
    const int base = sc.map[k].base, ld = sc.map[k].ld, chunk = sc.map[k].chunk;
    const unsigned magic = sc.map[k].magic;
    float* const img = sc.img[sc.map[k].which];
    if (i4 + 3 < begin || i4 >= end) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = (int)i4 + j;
      if (idx >= begin && idx < end) {
        const unsigned off = (unsigned)(idx - begin);
        const int r = (int)(((unsigned long long)off * magic) >> 32);
        const int c = (int)off - r * cols;
        img[img_dst(kind, base, ld, chunk, r, c)] = pa[j];
      }
    }
  }
}

// Input pipeline of a train graph: ALL of a launch's batches binarised by its first kernel.  The uniforms are keyed by
// (seed, the CONSUMING step's index, global pixel quad), none of which depends on the training state, so batch j of the
// launch (consumed by step *step_dev + j) needs no ordering with the steps before it.  (As extra workgroups of every
// step's optimizer launch the same work cost 3.5 us per step; up front it is ~0.4.)
__global__ void binarize_batches(const unsigned char* __restrict__ pixels, const int32_t* __restrict__ idx, uint64_t n_rows_src,
                                 int B, int D, int n_batches, uint64_t seed, const uint64_t* step_dev,
                                 unsigned char* __restrict__ x, uint64_t out_row0) {
  const uint64_t qpb = (uint64_t)B * (uint64_t)(D >> 2);
  const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t j = q / qpb;
  if (j >= (uint64_t)n_batches) return;
  binarize_quad(q - j * qpb, pixels, idx + j * (uint64_t)B, 0, n_rows_src, B, D, seed, *step_dev + j, x + j * (uint64_t)B * D, out_row0);
}

// auxiliary work without a GEMM: the image tasks of the first step of a train graph
__global__ __launch_bounds__(kThreads) void aux_only(const Aux ax) { aux_block(ax, (int)blockIdx.x); }

// ------------------------------------------------------------ cluster_acc
// utils.cluster_acc / mode_tensor (scripts/utils.py:156-191): argmax cluster,
// per-cluster label histogram (LDS-free global atomics on a tiny [K,n_labels]
// table), majority label (first maximum in first-occurrence order is not
// recoverable from a histogram: ties resolve to the smallest label), match rate.
__global__ void cluster_hist(const float* __restrict__ logits, const int64_t* __restrict__ labels, int B, int K,
                             int NL, int32_t* __restrict__ hist, int32_t* __restrict__ pred) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int best = 0;
  float bv = logits[(long long)b * K];
  for (int k = 1; k < K; ++k) {
    const float v = logits[(long long)b * K + k];
    if (v > bv) { bv = v; best = k; }
  }
  pred[b] = best;
  const int lab = (int)labels[b];
  if (lab >= 0 && lab < NL) atomicAdd(&hist[best * NL + lab], 1);
}
__global__ void cluster_match(const int32_t* __restrict__ hist, const int32_t* __restrict__ pred,
                              const int64_t* __restrict__ labels, int B, int K, int NL, float* __restrict__ acc) {
  __shared__ int mode[256];
  __shared__ int cnt[1024];
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    int best = 0, bc = 0;                 // empty cluster -> mode 0 (tf.cond branch, utils.py:183-185)
    for (int l = 0; l < NL; ++l)
      if (hist[k * NL + l] > bc) { bc = hist[k * NL + l]; best = l; }
    mode[k] = best;
  }
  __syncthreads();
  int c = 0;
  for (int b = threadIdx.x; b < B; b += blockDim.x) c += (mode[pred[b]] == (int)labels[b]);
  cnt[threadIdx.x] = c;
  __syncthreads();
  for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) cnt[threadIdx.x] += cnt[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) acc[0] = (float)cnt[0] / (float)B;
}

}  // namespace gmvae
