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
//
// Memory behaviour (measured with s_memtime stamps, tools/stamps.py): the first version spent
// 22 of 49 kcycles loading weights with dependent L2 round trips.  Now the padded / transposed
// LDS weight images are prepared once per step in global memory (aux.hpp) and every workgroup
// pulls its image, its noise rows and (backward) its saved activations with asynchronous
// LDS-DMA (global_load_lds_dwordx4) issued back to back and retired by ONE wait.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hpp"

namespace gmvae {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kPanel = 16;      // batch rows per workgroup
constexpr int kLDA = 17;        // leading dimension of [k][row] activation images

#define GMVAE_P4(n) (((n) + 3) & ~3)
#define GMVAE_P256(n) (((n) + 255) & ~255)

// fast transcendental forms (v_exp_f32 / v_log_f32 based, ~1e-6 relative): the ELBO tolerance is 1e-4
// (flog / fexp / fsoftplus: gemm.hpp)

// Diagnostic stamps (cdna_hip_programming.md section 7): values leave the kernel only through `dbg`.
#define GMVAE_STAMP(idx)                                                                   \
  if (a.dbg && threadIdx.x == 0) a.dbg[(size_t)blockIdx.x * 16 + (idx)] = __builtin_amdgcn_s_memtime()

// ---- LDS layouts (shared by the kernels, the image-preparation tasks and the host) ----------------
struct FwdLay {
  int KP, K2, L2;
  int W_y1, W_g0y, W_p, W_g1, W_d0, b_y0, b_y1, b_g0, b_p, b_g1, b_d0, img;   // image = [0, img)
  int A_hy, A_y, A_hg, A_z, P_gx, P_lg, P_qp, P_pp, P_eps, P_u, red, total;
};
__host__ __device__ inline FwdLay fwd_lay(int H, int L, int K) {
  FwdLay f;
  f.KP = (K + 15) & ~15; f.K2 = (K + 3) & ~3; f.L2 = 2 * L;
  int o = 0;
  auto take = [&](int n) { const int r = o; o += GMVAE_P4(n); return r; };
  f.W_y1 = take(H * f.KP);      // [H][KP]
  f.W_g0y = take(f.K2 * H);     // [K2][H]
  f.W_p = take(f.K2 * f.L2);    // [K2][L2]
  f.W_g1 = take(H * f.L2);      // [H][L2]
  f.W_d0 = take(L * H);         // [L][H]
  f.b_y0 = take(H); f.b_y1 = take(f.KP); f.b_g0 = take(H); f.b_p = take(f.L2); f.b_g1 = take(f.L2); f.b_d0 = take(H);
  o = GMVAE_P256(o);
  f.img = o;
  f.A_hy = take(H * kLDA); f.A_y = take(f.K2 * kLDA); f.A_hg = take(H * kLDA); f.A_z = take(L * kLDA);
  f.P_gx = take(kPanel * H); f.P_lg = take(kPanel * f.KP); f.P_qp = take(kPanel * f.L2); f.P_pp = take(kPanel * f.L2);
  f.P_eps = take(kPanel * L); f.P_u = take(kPanel * K);
  f.red = take(1024);
  f.total = o;
  return f;
}

struct BwdLay {
  int KP, K2, L2, LP, ldD, ldG, ldC, ldY;
  int W_d0T, W_g1T, W_cT, W_y1T, img;
  int A_dhd, A_dqp, A_cat, A_dl, P_dz, P_dy, P_qp, P_pp, P_z, P_eps, P_hd, P_hg, P_hy, P_y, P_lg, red, total;
};
__host__ __device__ inline BwdLay bwd_lay(int H, int L, int K) {
  BwdLay b;
  b.KP = (K + 15) & ~15; b.K2 = (K + 3) & ~3; b.L2 = 2 * L; b.LP = (L + 15) & ~15;
  // transposed weight images use ODD leading dimensions: the transposing fill then spreads its
  // writes over the banks (B-fragment reads are conflict-free for any leading dimension)
  b.ldD = b.LP + 1; b.ldG = H + 1; b.ldC = b.KP + 1; b.ldY = H + 1;
  int o = 0;
  auto take = [&](int n) { const int r = o; o += GMVAE_P4(n); return r; };
  b.W_d0T = take(H * b.ldD);            // [H][LP+1]    dz   = dhd * Wd0^T
  b.W_g1T = take(b.L2 * b.ldG);         // [L2][H+1]    dhg  = dqp * Wg1^T
  b.W_cT = take((H + b.L2) * b.ldC);    // [H+L2][KP+1] dy   = [dhg | dpp] * [Wg0y^T ; Wp^T]
  b.W_y1T = take(b.K2 * b.ldY);         // [K2][H+1]    dhy  = dlogits * Wy1^T
  o = GMVAE_P256(o);
  b.img = o;
  b.A_dhd = take(H * kLDA); b.A_dqp = take(b.L2 * kLDA); b.A_cat = take((H + b.L2) * kLDA); b.A_dl = take(b.K2 * kLDA);
  b.P_dz = take(kPanel * b.LP); b.P_dy = take(kPanel * b.KP);
  b.P_qp = take(kPanel * b.L2); b.P_pp = take(kPanel * b.L2); b.P_z = take(kPanel * L); b.P_eps = take(kPanel * L);
  b.P_hd = take(kPanel * H); b.P_hg = take(kPanel * H); b.P_hy = take(kPanel * H);
  b.P_y = take(kPanel * K); b.P_lg = take(kPanel * K);
  b.red = take(1024);
  b.total = o;
  return b;
}

// asynchronous linear copy global -> LDS (nfloats % 4 == 0, both 16-byte aligned); retire with dma_wait()
__device__ __forceinline__ void dma_copy(float* __restrict__ lds_dst, const float* __restrict__ g, const int nfloats,
                                         const int wave, const int lane) {
  for (int c = wave * 256; c < nfloats; c += 1024) {
    const int idx = c + lane * 4;
    if (idx < nfloats) __builtin_amdgcn_global_load_lds(g + idx, lds_dst + c, 16, 0, 0);
  }
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// out(row, col) for every 16x16 tile t = wave, wave+4, ... of  A[K][17] x B[K][ldb]
// (operand reads are issued 8 k-steps at a time so that the MFMA chain does not wait on each LDS read)
template <class Epi>
__device__ __forceinline__ void panel_gemm(const float* __restrict__ A, const float* __restrict__ Bw, const int ldb,
                                           const int K4, const int ntiles, const int wave, const int lane, Epi epi) {
  const int ln = lane & 15, lk = lane >> 4;
  for (int t = wave; t < ntiles; t += 4) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* ap = A + lk * kLDA + ln;
    const float* bp = Bw + lk * ldb + t * 16 + ln;
    for (int k0 = 0; k0 < K4; k0 += 32) {
      float av[8], bv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int kk = min(k0 + 4 * j, K4 - 4);
        av[j] = ap[kk * kLDA];
        bv[j] = bp[kk * ldb];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (k0 + 4 * j < K4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[j], acc, 0, 0, 0);
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
  for (int sb = s0; sb < s1; sb += 8) {
    float av[8], bv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int kk = min(sb + j, steps - 1) * 4;
      av[j] = A[(kk + lk) * kLDA + ln];
      bv[j] = Bw[(kk + lk) * ldb + tile * 16 + ln];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (sb + j < s1) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[j], acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave * 256 + (lk * 4 + r) * 16 + ln] = acc[r];
  __syncthreads();
  const int tid = wave * 64 + lane;
  const float v = red[tid] + red[256 + tid] + red[512 + tid] + red[768 + tid];
  epi(tid >> 4, tile * 16 + (tid & 15), v);
}

struct ChainFwdArgs {
  int B, H, L, K, NS;
  float c, smin, invT;
  const float* s1;            // [NS][B][2H] split-K partials of X*[Wy0 | Wg0x]
  const float* img;           // prepared LDS weight image (fwd_lay: [0, img))
  const float *eps, *u;       // [B,L], [B,K]
  float *hy1, *logits, *y, *nent, *hg1, *pp, *qp, *z, *logq, *logp, *hd1;
  unsigned long long* dbg;    // diagnostic only: per-workgroup s_memtime stamps [grid][16] (NULL in production)
};

__global__ __launch_bounds__(kThreads) void chain_fwd(const ChainFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = a.H, L = a.L, K = a.K, B = a.B;
  const FwdLay f = fwd_lay(H, L, K);
  const int KP = f.KP, K2 = f.K2, L2 = f.L2;
  float *W_y1 = sm + f.W_y1, *W_g0y = sm + f.W_g0y, *W_p = sm + f.W_p, *W_g1 = sm + f.W_g1, *W_d0 = sm + f.W_d0;
  float *b_y0 = sm + f.b_y0, *b_y1 = sm + f.b_y1, *b_g0 = sm + f.b_g0, *b_p = sm + f.b_p, *b_g1 = sm + f.b_g1, *b_d0 = sm + f.b_d0;
  float *A_hy = sm + f.A_hy, *A_y = sm + f.A_y, *A_hg = sm + f.A_hg, *A_z = sm + f.A_z;
  float *P_gx = sm + f.P_gx, *P_lg = sm + f.P_lg, *P_qp = sm + f.P_qp, *P_pp = sm + f.P_pp, *P_eps = sm + f.P_eps, *P_u = sm + f.P_u;
  float* red = sm + f.red;

  const int r0 = blockIdx.x * kPanel;
  const int nrow = min(kPanel, B - r0);
  GMVAE_STAMP(0);
  // ---- stage 0: asynchronous copies (weight image, noise rows) + reduction of the first-layer slabs
  dma_copy(sm, a.img, f.img, wave, lane);
  dma_copy(P_eps, a.eps + (long long)r0 * L, nrow * L, wave, lane);
  dma_copy(P_u, a.u + (long long)r0 * K, (nrow * K) & ~3, wave, lane);
  {
    const int H2 = 2 * H;
    const long long sstride = (long long)B * H2;
    float4 v[2];
    int rowc[2], colc[2];
    const int nitem = kPanel * H2 / 4;
#pragma unroll
    for (int it = 0; it < 2; ++it) {              // kPanel*2H/4 <= 512 items for H <= 64
      const int i = min(tid + it * kThreads, nitem - 1);
      rowc[it] = (i * 4) / H2;
      colc[it] = (i * 4) % H2;
      v[it] = slab_sum4(a.s1 + (long long)min(r0 + rowc[it], B - 1) * H2 + colc[it], sstride, a.NS);
    }
    for (int e = ((nrow * K) & ~3) + tid; e < nrow * K; e += kThreads) P_u[e] = a.u[(long long)r0 * K + e];   // ragged tail
    dma_wait();
    __syncthreads();                               // image (biases) and noise are in LDS
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      if (tid + it * kThreads < nitem) {
        const int row = rowc[it], col = colc[it];
        float vv[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
        if (col < H) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            vv[j] = fmaxf(vv[j] + b_y0[col + j], 0.f);
            A_hy[(col + j) * kLDA + row] = vv[j];
          }
          if (row < nrow) *reinterpret_cast<float4*>(a.hy1 + (long long)(r0 + row) * H + col) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        } else {
          *reinterpret_cast<float4*>(P_gx + row * H + (col - H)) = v[it];
        }
      }
    }
  }
  __syncthreads();

  GMVAE_STAMP(1);
  // ---- stage 1: logits = hy1 * Wy1 + by1
  for (int t = 0; t < KP / 16; ++t) {
    panel_gemm_ksplit(A_hy, W_y1, KP, H, t, red, wave, lane, [&](int row, int col, float v) {
      const float lg = v + b_y1[col];
      P_lg[row * KP + col] = lg;
      if (col < K && row < nrow) a.logits[(long long)(r0 + row) * K + col] = lg;
    });
    __syncthreads();
  }

  GMVAE_STAMP(2);
  // ---- stage 2: y = softmax((logits + gumbel)/T), nent = sum pi log pi   (16 lanes per row, K <= 64)
  {
    const int row = tid >> 4, sub = tid & 15;
    const bool ok = row < nrow;
    float lgv[4], av[4], lpv[4];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = sub + 16 * j;
      lgv[j] = -INFINITY; av[j] = -INFINITY;
      if (k < K) {
        lgv[j] = P_lg[row * KP + k];
        const float uu = ok ? P_u[row * K + k] : 0.5f;
        av[j] = (lgv[j] - flog(-flog(uu))) * a.invT;
        mx = fmaxf(mx, av[j]);
      }
    }
    mx = Sub16::max(mx);
    float se = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (sub + 16 * j < K) se += fexp(av[j] - mx);
    se = Sub16::sum(se);
    const float lse = mx + flog(se);
    cat_log_softmax<Sub16, 4>(lgv, lpv);           // log pi (gemm.hpp: accurate for a saturated q(y|x))
    float ne = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = sub + 16 * j;
      if (k < K2) {
        float yv = 0.f;
        if (k < K) {
          yv = fexp(av[j] - lse);
          const float lp = lpv[j];
          ne += fexp(lp) * lp;
          if (ok) a.y[(long long)(r0 + row) * K + k] = yv;
        }
        A_y[k * kLDA + row] = yv;
      }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) ne += __shfl_xor(ne, o, 64);
    if (sub == 0 && ok) a.nent[r0 + row] = ne;
  }
  __syncthreads();

  GMVAE_STAMP(3);
  // ---- stage 3: hg1 = relu(gx + y*Wg0y + bg0);  pp = y*Wp + bp
  panel_gemm(A_y, W_g0y, H, K2, H / 16, wave, lane, [&](int row, int col, float v) {
    const float h = fmaxf(v + P_gx[row * H + col] + b_g0[col], 0.f);
    A_hg[col * kLDA + row] = h;
    if (row < nrow) a.hg1[(long long)(r0 + row) * H + col] = h;
  });
  panel_gemm(A_y, W_p, L2, K2, L2 / 16, wave, lane, [&](int row, int col, float v) {
    const float p = v + b_p[col];
    P_pp[row * L2 + col] = p;
    if (row < nrow) a.pp[(long long)(r0 + row) * L2 + col] = p;
  });
  __syncthreads();

  GMVAE_STAMP(4);
  // ---- stage 4: qp = hg1 * Wg1 + bg1
  panel_gemm(A_hg, W_g1, L2, H, L2 / 16, wave, lane, [&](int row, int col, float v) {
    const float q = v + b_g1[col];
    P_qp[row * L2 + col] = q;
    if (row < nrow) a.qp[(long long)(r0 + row) * L2 + col] = q;
  });
  __syncthreads();

  GMVAE_STAMP(5);
  // ---- stage 5: z = mu + sigma*eps, log q(z|x,y), log p(z|y)
  {
    const int row = tid >> 4, sub = tid & 15;
    const bool ok = row < nrow;
    float aq = 0.f, ap = 0.f;
    for (int l = sub; l < L; l += 16) {
      const float mu = P_qp[row * L2 + l];
      const float sg = fmaxf(fsoftplus(P_qp[row * L2 + L + l] + a.c), a.smin);
      const float ee = ok ? P_eps[row * L + l] : 0.f;
      const float zz = mu + sg * ee;
      A_z[l * kLDA + row] = zz;
      if (ok) a.z[(long long)(r0 + row) * L + l] = zz;
      const float e = (zz - mu) / sg;
      aq += -0.5f * e * e - 0.5f * kLog2Pi - flog(sg);
      const float sp = fmaxf(fsoftplus(P_pp[row * L2 + L + l] + a.c), a.smin);
      const float t = (zz - P_pp[row * L2 + l]) / sp;
      ap += -0.5f * t * t - 0.5f * kLog2Pi - flog(sp);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { aq += __shfl_xor(aq, o, 64); ap += __shfl_xor(ap, o, 64); }
    if (sub == 0 && ok) { a.logq[r0 + row] = aq; a.logp[r0 + row] = ap; }
  }
  __syncthreads();

  GMVAE_STAMP(6);
  // ---- stage 6: hd1 = relu(z * Wd0 + bd0)
  panel_gemm(A_z, W_d0, H, L, H / 16, wave, lane, [&](int row, int col, float v) {
    if (row < nrow) a.hd1[(long long)(r0 + row) * H + col] = fmaxf(v + b_d0[col], 0.f);
  });
  GMVAE_STAMP(7);
}

// ------------------------------------------------------------------ backward
struct ChainBwdArgs {
  int B, H, L, K, NS, nparts;
  float c, smin, invT;
  const float* s4;            // [NS][B][H] split-K partials of (sigmoid(lambda)-x) * Wd1^T
  const float* img;           // prepared LDS image of the transposed weights (bwd_lay: [0, img))
  const float *hd1, *hg1, *hy1, *qp, *pp, *z, *eps, *y, *logits, *nent;
  const float *part, *logq, *logp;                   // Bernoulli partials [B][nparts]
  float *dhd1, *dqp, *dpp, *dhg1, *dlogits, *dhy1;   // pre-activation gradients for the dW GEMMs
  float *logpx, *logw;
  unsigned long long* dbg;
};

__global__ __launch_bounds__(kThreads) void chain_bwd(const ChainBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = a.H, L = a.L, K = a.K, B = a.B;
  const BwdLay f = bwd_lay(H, L, K);
  const int KP = f.KP, K2 = f.K2, L2 = f.L2, LP = f.LP;
  const int ldD = f.ldD, ldG = f.ldG, ldC = f.ldC, ldY = f.ldY;
  float *W_d0T = sm + f.W_d0T, *W_g1T = sm + f.W_g1T, *W_cT = sm + f.W_cT, *W_y1T = sm + f.W_y1T;
  float *A_dhd = sm + f.A_dhd, *A_dqp = sm + f.A_dqp, *A_cat = sm + f.A_cat, *A_dl = sm + f.A_dl;
  float *P_dz = sm + f.P_dz, *P_dy = sm + f.P_dy, *P_qp = sm + f.P_qp, *P_pp = sm + f.P_pp, *P_z = sm + f.P_z, *P_eps = sm + f.P_eps;
  float *P_hd = sm + f.P_hd, *P_hg = sm + f.P_hg, *P_hy = sm + f.P_hy, *P_y = sm + f.P_y, *P_lg = sm + f.P_lg;
  float* red = sm + f.red;
  float* A_dhg = A_cat;
  float* A_dpp = A_cat + H * kLDA;

  const int r0 = blockIdx.x * kPanel;
  const int nrow = min(kPanel, B - r0);
  GMVAE_STAMP(0);
  // ---- stage 0: asynchronous copies (weight image + this panel's saved activations); slab reduction; row terms
  dma_copy(sm, a.img, f.img, wave, lane);
  dma_copy(P_qp, a.qp + (long long)r0 * L2, nrow * L2, wave, lane);
  dma_copy(P_pp, a.pp + (long long)r0 * L2, nrow * L2, wave, lane);
  dma_copy(P_z, a.z + (long long)r0 * L, nrow * L, wave, lane);
  dma_copy(P_eps, a.eps + (long long)r0 * L, nrow * L, wave, lane);
  dma_copy(P_hd, a.hd1 + (long long)r0 * H, nrow * H, wave, lane);
  dma_copy(P_hg, a.hg1 + (long long)r0 * H, nrow * H, wave, lane);
  dma_copy(P_hy, a.hy1 + (long long)r0 * H, nrow * H, wave, lane);
  dma_copy(P_y, a.y + (long long)r0 * K, (nrow * K) & ~3, wave, lane);
  dma_copy(P_lg, a.logits + (long long)r0 * K, (nrow * K) & ~3, wave, lane);
  {
    const long long sstride = (long long)B * H;
    const int nitem = kPanel * H / 4;            // <= 256 for H <= 64
    const int i = min(tid, nitem - 1);
    const int row = (i * 4) / H, col = (i * 4) % H;
    const float4 v = slab_sum4(a.s4 + (long long)min(r0 + row, B - 1) * H + col, sstride, a.NS);
    if (tid < nrow) {                            // log p(x|z) = sum of the decoder epilogue partials; log w
      const int gr = r0 + tid;
      float s = 0.f;
      for (int p = 0; p < a.nparts; ++p) s += a.part[(long long)gr * a.nparts + p];
      a.logpx[gr] = s;
      a.logw[gr] = s + a.logp[gr] - a.logq[gr] - a.nent[gr];
    }
    for (int e = ((nrow * K) & ~3) + tid; e < nrow * K; e += kThreads) {     // ragged tails of the K-wide panels
      P_y[e] = a.y[(long long)r0 * K + e];
      P_lg[e] = a.logits[(long long)r0 * K + e];
    }
    dma_wait();
    __syncthreads();
    if (tid < nitem) {
      const float4 hm = *reinterpret_cast<const float4*>(P_hd + row * H + col);
      float vv[4] = {hm.x > 0.f ? v.x : 0.f, hm.y > 0.f ? v.y : 0.f, hm.z > 0.f ? v.z : 0.f, hm.w > 0.f ? v.w : 0.f};
      if (row >= nrow) { vv[0] = vv[1] = vv[2] = vv[3] = 0.f; }
#pragma unroll
      for (int j = 0; j < 4; ++j) A_dhd[(col + j) * kLDA + row] = vv[j];
      if (row < nrow) *reinterpret_cast<float4*>(a.dhd1 + (long long)(r0 + row) * H + col) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    }
  }
  __syncthreads();

  GMVAE_STAMP(1);
  // ---- stage 1: dz_dec = dhd1 * Wd0^T
  panel_gemm(A_dhd, W_d0T, ldD, H, LP / 16, wave, lane, [&](int row, int col, float v) { P_dz[row * LP + col] = v; });
  __syncthreads();

  GMVAE_STAMP(2);
  // ---- stage 2: seeds at z
  {
    const int row = tid >> 4, sub = tid & 15;
    const bool ok = row < nrow;
    for (int l = sub; l < L; l += 16) {
      float dmu = 0.f, draw = 0.f, dmup = 0.f, drawp = 0.f;
      if (ok) {
        const float rawq = P_qp[row * L2 + L + l] + a.c;
        const float spq = fsoftplus(rawq);
        const float sg = fmaxf(spq, a.smin);
        const float zz = P_z[row * L + l];
        const float rawp = P_pp[row * L2 + L + l] + a.c;
        const float spp = fsoftplus(rawp);
        const float sp = fmaxf(spp, a.smin);
        const float t = (zz - P_pp[row * L2 + l]) / sp;
        const float pterm = t / sp;
        dmu = P_dz[row * LP + l] + pterm;
        const float dsg = dmu * P_eps[row * L + l] - 1.f / sg;
        draw = (spq > a.smin) ? dsg * sigmoidf_(rawq) : 0.f;
        dmup = -pterm;
        drawp = (spp > a.smin) ? (1.f - t * t) / sp * sigmoidf_(rawp) : 0.f;
        float* dq = a.dqp + (long long)(r0 + row) * L2;
        float* dp = a.dpp + (long long)(r0 + row) * L2;
        dq[l] = dmu; dq[L + l] = draw; dp[l] = dmup; dp[L + l] = drawp;
      }
      A_dqp[l * kLDA + row] = dmu;
      A_dqp[(L + l) * kLDA + row] = draw;
      A_dpp[l * kLDA + row] = dmup;
      A_dpp[(L + l) * kLDA + row] = drawp;
    }
  }
  __syncthreads();

  GMVAE_STAMP(3);
  // ---- stage 3: dhg1 = (dqp * Wg1^T) * [hg1 > 0]
  panel_gemm(A_dqp, W_g1T, ldG, L2, H / 16, wave, lane, [&](int row, int col, float v) {
    const float d = (row < nrow && P_hg[row * H + col] > 0.f) ? v : 0.f;
    A_dhg[col * kLDA + row] = d;
    if (row < nrow) a.dhg1[(long long)(r0 + row) * H + col] = d;
  });
  __syncthreads();

  GMVAE_STAMP(4);
  // ---- stage 4: dy = dhg1 * Wg0[D:,:]^T + dpp * Wp^T     (one K = H + 2L product)
  for (int t = 0; t < KP / 16; ++t) {
    panel_gemm_ksplit(A_cat, W_cT, ldC, H + L2, t, red, wave, lane,
                      [&](int row, int col, float v) { P_dy[row * KP + col] = v; });
    __syncthreads();
  }

  GMVAE_STAMP(5);
  // ---- stage 5: softmax backward + entropy gradient -> dlogits
  {
    const int row = tid >> 4, sub = tid & 15;
    const bool ok = row < nrow;
    float lgv[4], yv[4], dyv[4], lpv[4], dav[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = sub + 16 * j;
      lgv[j] = -INFINITY; yv[j] = 0.f; dyv[j] = 0.f;
      if (k < K && ok) {
        lgv[j] = P_lg[row * K + k];
        yv[j] = P_y[row * K + k];
        dyv[j] = P_dy[row * KP + k];
      }
    }
    cat_log_softmax<Sub16, 4>(lgv, lpv);           // (gemm.hpp: the forms that survive a saturated softmax)
    cat_softmax_bwd<Sub16, 4>(yv, dyv, dav);
    float ne = 0.f;                                 // nent recomputed from the logits (cheaper than a dependent load)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (sub + 16 * j < K && ok) { const float lp = lpv[j]; ne += fexp(lp) * lp; }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) ne += __shfl_xor(ne, o, 64);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = sub + 16 * j;
      if (k < K2) {
        float dl = 0.f;
        if (k < K && ok) {
          const float lp = lpv[j];
          dl = dav[j] * a.invT + fexp(lp) * (lp - ne);
          a.dlogits[(long long)(r0 + row) * K + k] = dl;
        }
        A_dl[k * kLDA + row] = dl;
      }
    }
  }
  __syncthreads();

  GMVAE_STAMP(6);
  // ---- stage 6: dhy1 = (dlogits * Wy1^T) * [hy1 > 0]
  panel_gemm(A_dl, W_y1T, ldY, K2, H / 16, wave, lane, [&](int row, int col, float v) {
    if (row < nrow) a.dhy1[(long long)(r0 + row) * H + col] = P_hy[row * H + col] > 0.f ? v : 0.f;
  });
  GMVAE_STAMP(7);
}

}  // namespace gmvae
