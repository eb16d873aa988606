// Fused row-panel "chain" kernels for the small-layer middle of the GMVAE step.
//
// At the reference's default sizes (hidden 64, latent <= 64, K <= 64) every
// layer between the two 784-wide GEMMs is a few hundred KFLOP per row panel and
// the step is bound by kernel boundaries, not arithmetic.
// Here ONE workgroup owns a panel of 16 batch rows and carries it through the
// whole row-local chain with all weights resident in LDS:
//
//   chain_fwd:  reduce the split-K slabs of X*[Wy0|Wg0x]  -> relu -> logits (encoder_y,
//               gmvae.py:238) -> Gumbel-softmax y + entropy (gmvae.py:240,262) -> prior head
//               p(z|y) (gmvae.py:243) and encoder_gmm hidden (gmvae.py:246) -> q(z|x,y) head
//               -> z = mu + sigma*eps, log q, log p (gmvae.py:248,258) -> decoder hidden (gmvae.py:251)
//   chain_bwd:  the exact reverse-mode chain of the same span, producing every
//               pre-activation gradient the weight-gradient GEMMs need.
//
// Matrix products use v_mfma_f32_16x16x4_f32 (exact fp32); activations live in LDS
// as [k][row] images with leading dimension 17, so a finished 16x16 accumulator tile
// is written straight into the next product's A operand.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hpp"

namespace gmvae {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kPanel = 16;      // batch rows per workgroup
constexpr int kLDA = 17;        // leading dimension of [k][row] activation images

__device__ __forceinline__ int pad4i(int v) { return (v + 3) & ~3; }
__device__ __forceinline__ int pad16i(int v) { return (v + 15) & ~15; }

// out(row, col) for every 16x16 tile t = wave, wave+4, ... of  A[K][17] x B[K][ldb]
template <class Epi>
__device__ __forceinline__ void panel_gemm(const float* __restrict__ A, const float* __restrict__ Bw, const int ldb,
                                           const int K4, const int ntiles, const int wave, const int lane, Epi epi) {
  const int ln = lane & 15, lk = lane >> 4;
  for (int t = wave; t < ntiles; t += 4) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* ap = A + lk * kLDA + ln;
    const float* bp = Bw + lk * ldb + t * 16 + ln;
#pragma unroll 4
    for (int kk = 0; kk < K4; kk += 4) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[kk * kLDA], bp[kk * ldb], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) epi(lk * 4 + r, t * 16 + ln, acc[r]);
  }
}

// Single 16x16 output tile with the K range split over the 4 waves (reduced through `red`, 4*256 floats).
template <class Epi>
__device__ __forceinline__ void panel_gemm_ksplit(const float* __restrict__ A, const float* __restrict__ Bw,
                                                  const int ldb, const int K4, const int tile, float* __restrict__ red,
                                                  const int wave, const int lane, Epi epi) {
  const int ln = lane & 15, lk = lane >> 4;
  const int steps = K4 / 4;
  const int per = (steps + 3) / 4;
  const int s0 = wave * per, s1 = min(steps, s0 + per);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int s = s0; s < s1; ++s) {
    const int kk = s * 4;
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(kk + lk) * kLDA + ln], Bw[(kk + lk) * ldb + tile * 16 + ln], acc, 0,
                                               0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave * 256 + (lk * 4 + r) * 16 + ln] = acc[r];
  __syncthreads();
  const int tid = wave * 64 + lane;
  const float v = red[tid] + red[256 + tid] + red[512 + tid] + red[768 + tid];
  epi(tid >> 4, tile * 16 + (tid & 15), v);
}

// Cooperative global -> LDS copies of a row-major [rows][cols] matrix.  The destination region has
// been zeroed beforehand (padding).  Loads are issued in independent batches (4 x 16 B or 8 x 4 B per
// thread in flight) -- a naive one-element-per-iteration loop serialises ~60 L2 round trips per thread.
//   TRANS = false: dst[r*ld + c] = src[r][c]        TRANS = true: dst[c*ld + r] = src[r][c]
template <bool TRANS>
__device__ __forceinline__ void lds_fill(float* __restrict__ dst, const int ld, const float* __restrict__ src,
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

__device__ __forceinline__ void lds_zero(float* __restrict__ p, const int nfloats, const int tid) {
  for (int i = tid * 4; i < nfloats; i += kThreads * 4) *reinterpret_cast<float4*>(p + i) = make_float4(0.f, 0.f, 0.f, 0.f);
}

// sum of NS split-K slabs at one float4 location, all loads in flight at once (NS <= 16)
__device__ __forceinline__ float4 slab_sum4(const float* __restrict__ p, const long long sstride, const int NS) {
  float4 v[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) v[s] = *reinterpret_cast<const float4*>(p + (long long)min(s, NS - 1) * sstride);
  float4 r = v[0];
#pragma unroll
  for (int s = 1; s < 16; ++s)
    if (s < NS) { r.x += v[s].x; r.y += v[s].y; r.z += v[s].z; r.w += v[s].w; }
  return r;
}

struct ChainFwdArgs {
  int B, H, L, K, NS;
  float c, smin, invT;
  const float* s1;            // [NS][B][2H] split-K partials of X*[Wy0 | Wg0x]
  const float *by0, *Wy1, *by1, *Wg0y, *bg0, *Wp, *bp, *Wg1, *bg1, *Wd0, *bd0;
  const float *eps, *u;       // [B,L], [B,K]
  float *hy1, *logits, *y, *nent, *hg1, *pp, *qp, *z, *logq, *logp, *hd1;
};

#define GMVAE_P4(n) (((n) + 3) & ~3)
__host__ __device__ inline int chain_fwd_lds_floats(int H, int L, int K) {
  const int KP = (K + 15) & ~15, K2 = (K + 3) & ~3, L2 = 2 * L;
  return GMVAE_P4(H * KP) + GMVAE_P4(K2 * H) + GMVAE_P4(K2 * L2) + GMVAE_P4(H * L2) + GMVAE_P4(L * H)   // weights
         + GMVAE_P4(KP) + GMVAE_P4(H) + GMVAE_P4(L2) + GMVAE_P4(L2) + GMVAE_P4(H)                       // biases
         + GMVAE_P4(H * kLDA) + GMVAE_P4(K2 * kLDA) + GMVAE_P4(H * kLDA) + GMVAE_P4(L * kLDA)           // activation images
         + kPanel * H + kPanel * KP + 2 * kPanel * L2                                                    // gx, logits, qp, pp panels
         + 1024;                                                                                         // k-split reduction scratch
}

__global__ __launch_bounds__(kThreads) void chain_fwd(const ChainFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = a.H, L = a.L, K = a.K, L2 = 2 * L, B = a.B;
  const int KP = pad16i(K), K2 = pad4i(K);
  int lo = 0;
  auto take = [&](int n) { float* p = sm + lo; lo += GMVAE_P4(n); return p; };   // every region 16-byte aligned
  float* W_y1 = take(H * KP);           // [H][KP]
  float* W_g0y = take(K2 * H);          // [K2][H]
  float* W_p = take(K2 * L2);           // [K2][L2]
  float* W_g1 = take(H * L2);           // [H][L2]
  float* W_d0 = take(L * H);            // [L][H]
  float* b_y1 = take(KP);
  float* b_g0 = take(H);
  float* b_p = take(L2);
  float* b_g1 = take(L2);
  float* b_d0 = take(H);
  float* A_hy = take(H * kLDA);         // [H][17]
  float* A_y = take(K2 * kLDA);         // [K2][17]
  float* A_hg = take(H * kLDA);         // [H][17]
  float* A_z = take(L * kLDA);          // [L][17]
  float* P_gx = take(kPanel * H);       // [16][H]
  float* P_lg = take(kPanel * KP);      // [16][KP]
  float* P_qp = take(kPanel * L2);      // [16][L2]
  float* P_pp = take(kPanel * L2);      // [16][L2]
  float* red = take(1024);

  const int r0 = blockIdx.x * kPanel;
  // ---- stage 0: weights -> LDS; reduce the first-layer split-K slabs
  lds_zero(sm, (int)(A_hy - sm), tid);          // weight + bias region (zero padding rows / columns)
  __syncthreads();
  lds_fill<false>(W_y1, KP, a.Wy1, H, K, K, tid);
  lds_fill<false>(W_g0y, H, a.Wg0y, K, H, H, tid);
  lds_fill<false>(W_p, L2, a.Wp, K, L2, L2, tid);
  lds_fill<false>(W_g1, L2, a.Wg1, H, L2, L2, tid);
  lds_fill<false>(W_d0, H, a.Wd0, L, H, H, tid);
  for (int i = tid; i < K; i += kThreads) b_y1[i] = a.by1[i];
  for (int i = tid; i < H; i += kThreads) { b_g0[i] = a.bg0[i]; b_d0[i] = a.bd0[i]; }
  for (int i = tid; i < L2; i += kThreads) { b_p[i] = a.bp[i]; b_g1[i] = a.bg1[i]; }
  {
    const int H2 = 2 * H;
    const long long sstride = (long long)B * H2;
    for (int i = tid; i < kPanel * H2 / 4; i += kThreads) {
      const int row = (i * 4) / H2, col = (i * 4) % H2;
      const int gr = min(r0 + row, B - 1);
      const float4 v = slab_sum4(a.s1 + (long long)gr * H2 + col, sstride, a.NS);
      float vv[4] = {v.x, v.y, v.z, v.w};
      if (col < H) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          vv[j] = fmaxf(vv[j] + a.by0[col + j], 0.f);
          A_hy[(col + j) * kLDA + row] = vv[j];
        }
        if (r0 + row < B) *reinterpret_cast<float4*>(a.hy1 + (long long)gr * H + col) = make_float4(vv[0], vv[1], vv[2], vv[3]);
      } else {
        *reinterpret_cast<float4*>(P_gx + row * H + (col - H)) = v;
      }
    }
  }
  __syncthreads();

  // ---- stage 1: logits = hy1 * Wy1 + by1
  for (int t = 0; t < KP / 16; ++t) {
    panel_gemm_ksplit(A_hy, W_y1, KP, H, t, red, wave, lane, [&](int row, int col, float v) {
      const float lg = v + b_y1[col];
      P_lg[row * KP + col] = lg;
      if (col < K && r0 + row < B) a.logits[(long long)(r0 + row) * K + col] = lg;
    });
    __syncthreads();
  }

  // ---- stage 2: y = softmax((logits + gumbel)/T), nent = sum pi log pi   (16 lanes per row)
  {
    const int row = tid >> 4, sub = tid & 15;
    const int gr = min(r0 + row, B - 1);
    float mx = -INFINITY, m2 = -INFINITY;
    for (int k = sub; k < K; k += 16) {
      const float lg = P_lg[row * KP + k];
      const float g = -logf(-logf(a.u[(long long)gr * K + k]));
      mx = fmaxf(mx, (lg + g) * a.invT);
      m2 = fmaxf(m2, lg);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, o, 64)); m2 = fmaxf(m2, __shfl_xor(m2, o, 64)); }
    float se = 0.f, s2 = 0.f;
    for (int k = sub; k < K; k += 16) {
      const float lg = P_lg[row * KP + k];
      const float g = -logf(-logf(a.u[(long long)gr * K + k]));
      se += expf((lg + g) * a.invT - mx);
      s2 += expf(lg - m2);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { se += __shfl_xor(se, o, 64); s2 += __shfl_xor(s2, o, 64); }
    const float lse = mx + logf(se), l2 = m2 + logf(s2);
    float ne = 0.f;
    for (int k = sub; k < K2; k += 16) {
      float yv = 0.f;
      if (k < K) {
        const float lg = P_lg[row * KP + k];
        const float g = -logf(-logf(a.u[(long long)gr * K + k]));
        yv = expf((lg + g) * a.invT - lse);
        const float lp = lg - l2;
        ne += expf(lp) * lp;
        if (r0 + row < B) a.y[(long long)gr * K + k] = yv;
      }
      A_y[k * kLDA + row] = yv;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) ne += __shfl_xor(ne, o, 64);
    if (sub == 0 && r0 + row < B) a.nent[gr] = ne;
  }
  __syncthreads();

  // ---- stage 3: hg1 = relu(gx + y*Wg0y + bg0);  pp = y*Wp + bp
  panel_gemm(A_y, W_g0y, H, K2, H / 16, wave, lane, [&](int row, int col, float v) {
    const float h = fmaxf(v + P_gx[row * H + col] + b_g0[col], 0.f);
    A_hg[col * kLDA + row] = h;
    if (r0 + row < B) a.hg1[(long long)(r0 + row) * H + col] = h;
  });
  panel_gemm(A_y, W_p, L2, K2, L2 / 16, wave, lane, [&](int row, int col, float v) {
    const float p = v + b_p[col];
    P_pp[row * L2 + col] = p;
    if (r0 + row < B) a.pp[(long long)(r0 + row) * L2 + col] = p;
  });
  __syncthreads();

  // ---- stage 4: qp = hg1 * Wg1 + bg1
  panel_gemm(A_hg, W_g1, L2, H, L2 / 16, wave, lane, [&](int row, int col, float v) {
    const float q = v + b_g1[col];
    P_qp[row * L2 + col] = q;
    if (r0 + row < B) a.qp[(long long)(r0 + row) * L2 + col] = q;
  });
  __syncthreads();

  // ---- stage 5: z = mu + sigma*eps, log q(z|x,y), log p(z|y)
  {
    const int row = tid >> 4, sub = tid & 15;
    const int gr = min(r0 + row, B - 1);
    float aq = 0.f, ap = 0.f;
    for (int l = sub; l < L; l += 16) {
      const float mu = P_qp[row * L2 + l];
      const float sg = fmaxf(softplusf_(P_qp[row * L2 + L + l] + a.c), a.smin);
      const float zz = mu + sg * a.eps[(long long)gr * L + l];
      A_z[l * kLDA + row] = zz;
      if (r0 + row < B) a.z[(long long)gr * L + l] = zz;
      const float e = (zz - mu) / sg;
      aq += -0.5f * e * e - 0.5f * kLog2Pi - logf(sg);
      const float sp = fmaxf(softplusf_(P_pp[row * L2 + L + l] + a.c), a.smin);
      const float t = (zz - P_pp[row * L2 + l]) / sp;
      ap += -0.5f * t * t - 0.5f * kLog2Pi - logf(sp);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { aq += __shfl_xor(aq, o, 64); ap += __shfl_xor(ap, o, 64); }
    if (sub == 0 && r0 + row < B) { a.logq[gr] = aq; a.logp[gr] = ap; }
  }
  __syncthreads();

  // ---- stage 6: hd1 = relu(z * Wd0 + bd0)
  panel_gemm(A_z, W_d0, H, L, H / 16, wave, lane, [&](int row, int col, float v) {
    if (r0 + row < B) a.hd1[(long long)(r0 + row) * H + col] = fmaxf(v + b_d0[col], 0.f);
  });
}

// ------------------------------------------------------------------ backward
struct ChainBwdArgs {
  int B, H, L, K, NS, nparts;
  float c, smin, invT;
  const float* s4;            // [NS][B][H] split-K partials of (sigmoid(lambda)-x) * Wd1^T
  const float *Wy1, *Wg0y, *Wp, *Wg1, *Wd0;
  const float *hd1, *hg1, *hy1, *qp, *pp, *z, *eps, *y, *logits, *nent;
  const float *part, *logq, *logp;                   // Bernoulli partials [B][nparts]
  float *dhd1, *dqp, *dpp, *dhg1, *dlogits, *dhy1;   // pre-activation gradients for the dW GEMMs
  float *logpx, *logw;
};

__host__ __device__ inline int chain_bwd_lds_floats(int H, int L, int K) {
  const int KP = (K + 15) & ~15, K2 = (K + 3) & ~3, L2 = 2 * L, LP = (L + 15) & ~15;
  return GMVAE_P4(H * (LP + 1)) + GMVAE_P4(L2 * (H + 1)) + GMVAE_P4((H + L2) * (KP + 1)) + GMVAE_P4(K2 * (H + 1))  // W^T, odd ld
         + GMVAE_P4(H * kLDA) + GMVAE_P4(L2 * kLDA) + GMVAE_P4((H + L2) * kLDA) + GMVAE_P4(K2 * kLDA)              // A_dhd, A_dqp, A_dhg|A_dpp, A_dl
         + kPanel * LP + kPanel * KP                                                                                 // dz, dy panels
         + 1024;
}

__global__ __launch_bounds__(kThreads) void chain_bwd(const ChainBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = a.H, L = a.L, K = a.K, L2 = 2 * L, B = a.B;
  const int KP = pad16i(K), K2 = pad4i(K), LP = pad16i(L);
  // transposed weight images use ODD leading dimensions: the transposing fill then spreads its LDS
  // writes over the banks (B-fragment reads are conflict-free for any leading dimension)
  const int ldD = LP + 1, ldG = H + 1, ldC = KP + 1, ldY = H + 1;
  int lo = 0;
  auto take = [&](int n) { float* p = sm + lo; lo += GMVAE_P4(n); return p; };
  float* W_d0T = take(H * ldD);           // [H][LP+1]    dz   = dhd * Wd0^T
  float* W_g1T = take(L2 * ldG);          // [L2][H+1]    dhg  = dqp * Wg1^T
  float* W_cT = take((H + L2) * ldC);     // [H+L2][KP+1] dy   = [dhg | dpp] * [Wg0y^T ; Wp^T]
  float* W_y1T = take(K2 * ldY);          // [K2][H+1]    dhy  = dlogits * Wy1^T
  float* A_dhd = take(H * kLDA);          // [H][17]
  float* A_dqp = take(L2 * kLDA);         // [L2][17]
  float* A_cat = take((H + L2) * kLDA);   // [H + L2][17]   = A_dhg followed by A_dpp
  float* A_dl = take(K2 * kLDA);          // [K2][17]
  float* P_dz = take(kPanel * LP);        // [16][LP]
  float* P_dy = take(kPanel * KP);        // [16][KP]
  float* red = take(1024);
  float* A_dhg = A_cat;
  float* A_dpp = A_cat + H * kLDA;

  const int r0 = blockIdx.x * kPanel;
  // ---- stage 0: transposed weights -> LDS; reduce the slabs of the top data gradient; row terms
  lds_zero(sm, (int)(A_dhd - sm), tid);
  __syncthreads();
  lds_fill<true>(W_d0T, ldD, a.Wd0, L, H, H, tid);             // W_d0T[h][l] = Wd0[l][h]
  lds_fill<true>(W_g1T, ldG, a.Wg1, H, L2, L2, tid);           // W_g1T[j][h] = Wg1[h][j]
  lds_fill<true>(W_cT, ldC, a.Wg0y, K, H, H, tid);             // W_cT[h][k]  = Wg0y[k][h]
  lds_fill<true>(W_cT + H * ldC, ldC, a.Wp, K, L2, L2, tid);   // W_cT[H+j][k] = Wp[k][j]
  lds_fill<true>(W_y1T, ldY, a.Wy1, H, K, K, tid);             // W_y1T[k][h] = Wy1[h][k]
  {
    const long long sstride = (long long)B * H;
    for (int i = tid; i < kPanel * H / 4; i += kThreads) {
      const int row = (i * 4) / H, col = (i * 4) % H;
      const int gr = min(r0 + row, B - 1);
      const float4 v = slab_sum4(a.s4 + (long long)gr * H + col, sstride, a.NS);
      const float4 hm = *reinterpret_cast<const float4*>(a.hd1 + (long long)gr * H + col);
      float vv[4] = {hm.x > 0.f ? v.x : 0.f, hm.y > 0.f ? v.y : 0.f, hm.z > 0.f ? v.z : 0.f, hm.w > 0.f ? v.w : 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) A_dhd[(col + j) * kLDA + row] = vv[j];
      if (r0 + row < B) *reinterpret_cast<float4*>(a.dhd1 + (long long)gr * H + col) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    }
    if (tid < kPanel && r0 + tid < B) {      // log p(x|z) = sum of the decoder epilogue partials; log w
      const int gr = r0 + tid;
      float s = 0.f;
      for (int i = 0; i < a.nparts; ++i) s += a.part[(long long)gr * a.nparts + i];
      a.logpx[gr] = s;
      a.logw[gr] = s + a.logp[gr] - a.logq[gr] - a.nent[gr];
    }
  }
  __syncthreads();

  // ---- stage 1: dz_dec = dhd1 * Wd0^T
  panel_gemm(A_dhd, W_d0T, ldD, H, LP / 16, wave, lane, [&](int row, int col, float v) { P_dz[row * LP + col] = v; });
  __syncthreads();

  // ---- stage 2: seeds at z
  {
    const int row = tid >> 4, sub = tid & 15;
    const int gr = min(r0 + row, B - 1);
    const bool ok = r0 + row < B;
    for (int l = sub; l < L; l += 16) {
      const float rawq = a.qp[(long long)gr * L2 + L + l] + a.c;
      const float spq = softplusf_(rawq);
      const float sg = fmaxf(spq, a.smin);
      const float zz = a.z[(long long)gr * L + l];
      const float rawp = a.pp[(long long)gr * L2 + L + l] + a.c;
      const float spp = softplusf_(rawp);
      const float sp = fmaxf(spp, a.smin);
      const float t = (zz - a.pp[(long long)gr * L2 + l]) / sp;
      const float pterm = t / sp;
      const float dmu = P_dz[row * LP + l] + pterm;
      const float dsg = dmu * a.eps[(long long)gr * L + l] - 1.f / sg;
      const float draw = (spq > a.smin) ? dsg * sigmoidf_(rawq) : 0.f;
      const float dmup = -pterm;
      const float drawp = (spp > a.smin) ? (1.f - t * t) / sp * sigmoidf_(rawp) : 0.f;
      A_dqp[l * kLDA + row] = dmu;
      A_dqp[(L + l) * kLDA + row] = draw;
      A_dpp[l * kLDA + row] = dmup;
      A_dpp[(L + l) * kLDA + row] = drawp;
      if (ok) {
        a.dqp[(long long)gr * L2 + l] = dmu;
        a.dqp[(long long)gr * L2 + L + l] = draw;
        a.dpp[(long long)gr * L2 + l] = dmup;
        a.dpp[(long long)gr * L2 + L + l] = drawp;
      }
    }
  }
  __syncthreads();

  // ---- stage 3: dhg1 = (dqp * Wg1^T) * [hg1 > 0]
  panel_gemm(A_dqp, W_g1T, ldG, L2, H / 16, wave, lane, [&](int row, int col, float v) {
    const int gr = min(r0 + row, B - 1);
    const float d = a.hg1[(long long)gr * H + col] > 0.f ? v : 0.f;
    A_dhg[col * kLDA + row] = d;
    if (r0 + row < B) a.dhg1[(long long)gr * H + col] = d;
  });
  __syncthreads();

  // ---- stage 4: dy = dhg1 * Wg0[D:,:]^T + dpp * Wp^T     (one K = H + 2L product)
  for (int t = 0; t < KP / 16; ++t) {
    panel_gemm_ksplit(A_cat, W_cT, ldC, H + L2, t, red, wave, lane,
                      [&](int row, int col, float v) { P_dy[row * KP + col] = v; });
    __syncthreads();
  }

  // ---- stage 5: softmax backward + entropy gradient -> dlogits
  {
    const int row = tid >> 4, sub = tid & 15;
    const int gr = min(r0 + row, B - 1);
    float m2 = -INFINITY;
    for (int k = sub; k < K; k += 16) m2 = fmaxf(m2, a.logits[(long long)gr * K + k]);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) m2 = fmaxf(m2, __shfl_xor(m2, o, 64));
    float s2 = 0.f, dot = 0.f;
    for (int k = sub; k < K; k += 16) {
      s2 += expf(a.logits[(long long)gr * K + k] - m2);
      dot += a.y[(long long)gr * K + k] * P_dy[row * KP + k];
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { s2 += __shfl_xor(s2, o, 64); dot += __shfl_xor(dot, o, 64); }
    const float l2 = m2 + logf(s2), ne = a.nent[gr];
    for (int k = sub; k < K2; k += 16) {
      float dl = 0.f;
      if (k < K) {
        const float lp = a.logits[(long long)gr * K + k] - l2;
        dl = a.y[(long long)gr * K + k] * (P_dy[row * KP + k] - dot) * a.invT + expf(lp) * (lp - ne);
        if (r0 + row < B) a.dlogits[(long long)gr * K + k] = dl;
      }
      A_dl[k * kLDA + row] = dl;
    }
  }
  __syncthreads();

  // ---- stage 6: dhy1 = (dlogits * Wy1^T) * [hy1 > 0]
  panel_gemm(A_dl, W_y1T, ldY, K2, H / 16, wave, lane, [&](int row, int col, float v) {
    const int gr = min(r0 + row, B - 1);
    if (r0 + row < B) a.dhy1[(long long)gr * H + col] = a.hy1[(long long)gr * H + col] > 0.f ? v : 0.f;
  });
}

}  // namespace gmvae
