// Grouped fp32-MFMA GEMM with fused epilogues for gfx950 (MI355X).
//
// One launch executes up to MAXP independent GEMM "problems" (different
// shapes, operand layouts and epilogues) so that a whole dependency level of
// the training step (e.g. dW and dX of one layer) is a single kernel boundary.
// Arithmetic is v_mfma_f32_32x32x2_f32: exact fp32 products and fp32
// accumulation (the 1e-4 ELBO tolerance rules out bf16 inputs).
//
// Replaces: snt.nets.MLP MatMul/BiasAdd/Relu nodes (scripts/base.py:47-60,
// 67,135,198), their TF autodiff counterparts (scripts/runners.py:182) and the
// Independent(Bernoulli).log_prob expansion (scripts/base.py:143-146).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gmvae {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;   // 4 wavefronts of 64
constexpr int kBK = 32;         // K-chunk staged through LDS
constexpr int MAXP = 4;         // problems per launch

enum { EPI_STORE = 0, EPI_BERNOULLI = 1 };

// One GEMM operand as seen by the kernel: a logical [mn][k] matrix.
//   k_contig: element(mn,k) = ptr[(mn/row_div)*ld + k]      (row-major [mn][k])
//   else    : element(mn,k) = ptr[(k /row_div)*ld + mn]     (row-major [k][mn])
// ones_row: logical row mn == n_mn reads 1.0 (folds the bias gradient into dW).
struct Operand {
  const void* ptr;
  int ld;
  int n_mn;
  int row_div;
  unsigned char is_u8, k_contig, ones_row, vec_ok;
};

struct Segment {
  Operand a, b;
  int K;
  const float* kscale;   // optional per-k scale applied to b (IWAE row weights)
};

struct Problem {
  int M, N;                // output extents (M counts the ones row)
  int nseg;
  int tiles_m, tiles_n, splits, tile_begin;
  int epi;
  int ldc;
  int relu;
  int ld_add, add_div;
  int ld_mask;
  int ldx, x_div, nparts;
  float addconst;
  long long split_stride;  // floats between split-K slabs
  float* C;
  float* bias_row_out;     // destination of row M-1 when a.ones_row
  const float* bias;
  const float* addsrc;
  const float* mask;       // keep where mask > 0
  const float* rowscale;
  const unsigned char* x;  // Bernoulli targets
  float* part;             // Bernoulli row partial sums [M][nparts]
  Segment seg[2];
};

struct Launch {
  int nprob;
  Problem p[MAXP];
};

template <int BM_, int BN_, int WM_, int WN_, int WK_>
struct Cfg {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, WK = WK_;
  static constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  static constexpr int LDA = BM + 1, LDB = BN + 1, LDC = BN + 4;
  static constexpr int NSA = BM * kBK / 4 / kThreads, NSB = BN * kBK / 4 / kThreads;
  static constexpr int OPS = 2 * (LDA + LDB) * kBK;                 // floats, double buffered
  static constexpr int CST = WK * BM * LDC;                          // floats, C staging
  static constexpr int LDS_FLOATS = OPS > CST ? OPS : CST;
  static_assert(WM * WN * WK == 4, "4 waves per workgroup");
  static_assert(NSA >= 1 && NSB >= 1, "tile too small for 256 threads");
  static_assert((kBK / WK) % 2 == 0, "k slice per wave must be even");
};
typedef Cfg<32, 32, 1, 1, 4> CfgS;      // latency-bound: 4 waves split K inside the tile
typedef Cfg<64, 64, 2, 2, 1> CfgM;
typedef Cfg<128, 128, 2, 2, 1> CfgL;    // MFMA-bound: 64x64 per wave

__device__ __forceinline__ float sigmoidf_(float v) {
  float e = __expf(-fabsf(v));
  float r = 1.0f / (1.0f + e);
  return v >= 0.f ? r : e * r;
}
__device__ __forceinline__ float softplusf_(float v) { return fmaxf(v, 0.f) + log1pf(__expf(-fabsf(v))); }

// ---- global -> register slot loads -------------------------------------
// A slot is 4 consecutive elements along the source's contiguous dimension.
template <int BMN>
__device__ __forceinline__ void slot_coords(const Operand& op, int s, int& mn, int& k) {
  if (op.k_contig) {              // 8 slots per row: 8 full 128-byte lines per wave instruction
    mn = s >> 3;
    k = (s & 7) << 2;
  } else {                        // lanes: 4 k-rows x 8 slots -> conflict-free LDS writes with odd LD
    constexpr int QN = BMN / 4;
    k = (s & 3) + 4 * (s / (4 * QN));
    mn = ((s >> 2) % QN) << 2;
  }
}

__device__ __forceinline__ float4 load_slot(const Operand& op, int mn, int k, int k_end, const float* kscale) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (op.k_contig) {
    const long long base = (long long)(mn / op.row_div) * op.ld + k;
    if (op.vec_ok && mn < op.n_mn && k + 3 < k_end) {
      if (op.is_u8) {
        uint32_t w = *reinterpret_cast<const uint32_t*>(static_cast<const unsigned char*>(op.ptr) + base);
        v = make_float4((float)(w & 0xff), (float)((w >> 8) & 0xff), (float)((w >> 16) & 0xff), (float)(w >> 24));
      } else {
        v = *reinterpret_cast<const float4*>(static_cast<const float*>(op.ptr) + base);
      }
    } else {
      float t[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        t[j] = 0.f;
        if (k + j < k_end) {
          if (mn < op.n_mn)
            t[j] = op.is_u8 ? (float)static_cast<const unsigned char*>(op.ptr)[base + j]
                            : static_cast<const float*>(op.ptr)[base + j];
          else if (op.ones_row && mn == op.n_mn)
            t[j] = 1.f;
        }
      }
      v = make_float4(t[0], t[1], t[2], t[3]);
    }
    if (kscale) {
      if (k + 0 < k_end) v.x *= kscale[k + 0];
      if (k + 1 < k_end) v.y *= kscale[k + 1];
      if (k + 2 < k_end) v.z *= kscale[k + 2];
      if (k + 3 < k_end) v.w *= kscale[k + 3];
    }
  } else {
    if (k < k_end) {
      const long long base = (long long)(k / op.row_div) * op.ld + mn;
      if (op.vec_ok && mn + 3 < op.n_mn) {
        if (op.is_u8) {
          uint32_t w = *reinterpret_cast<const uint32_t*>(static_cast<const unsigned char*>(op.ptr) + base);
          v = make_float4((float)(w & 0xff), (float)((w >> 8) & 0xff), (float)((w >> 16) & 0xff), (float)(w >> 24));
        } else {
          v = *reinterpret_cast<const float4*>(static_cast<const float*>(op.ptr) + base);
        }
      } else {
        float t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          t[j] = 0.f;
          if (mn + j < op.n_mn)
            t[j] = op.is_u8 ? (float)static_cast<const unsigned char*>(op.ptr)[base + j]
                            : static_cast<const float*>(op.ptr)[base + j];
          else if (op.ones_row && mn + j == op.n_mn)
            t[j] = 1.f;
        }
        v = make_float4(t[0], t[1], t[2], t[3]);
      }
      if (kscale) {
        const float sc = kscale[k];
        v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
      }
    }
  }
  return v;
}

// LDS image of an operand tile is always [k][mn] with an odd leading dimension.
template <int LD>
__device__ __forceinline__ void store_slot(float* T, bool k_contig, int mn, int k, float4 v) {
  if (k_contig) {
    T[(k + 0) * LD + mn] = v.x;
    T[(k + 1) * LD + mn] = v.y;
    T[(k + 2) * LD + mn] = v.z;
    T[(k + 3) * LD + mn] = v.w;
  } else {
    float* p = T + k * LD + mn;
    p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
  }
}

template <class C>
__global__ __launch_bounds__(kThreads) void gemm_grouped(const Launch L) {
  __shared__ float lds[C::LDS_FLOATS];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int pi = 0;
  for (int i = 1; i < L.nprob; ++i)
    if ((int)blockIdx.x >= L.p[i].tile_begin) pi = i;
  const Problem& P = L.p[pi];

  int t = blockIdx.x - P.tile_begin;
  const int split = t % P.splits;
  t /= P.splits;
  const int tn = t % P.tiles_n;
  const int tm = t / P.tiles_n;
  const int m0 = tm * C::BM, n0 = tn * C::BN;

  // wave placement inside the tile
  const int wk = wave / (C::WM * C::WN);
  const int wmn = wave % (C::WM * C::WN);
  const int wm0 = (wmn / C::WN) * (C::TM * 32);
  const int wn0 = (wmn % C::WN) * (C::TN * 32);

  f32x16 acc[C::TM][C::TN];
#pragma unroll
  for (int i = 0; i < C::TM; ++i)
#pragma unroll
    for (int j = 0; j < C::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // chunk schedule over (segment, k-range of this split)
  int kb[2], ke[2], nc[2] = {0, 0};
  for (int s = 0; s < P.nseg; ++s) {
    const int K = P.seg[s].K;
    int kper = (K + P.splits - 1) / P.splits;
    kper = (kper + kBK - 1) / kBK * kBK;
    kb[s] = split * kper < K ? split * kper : K;
    ke[s] = kb[s] + kper < K ? kb[s] + kper : K;
    nc[s] = (ke[s] - kb[s] + kBK - 1) / kBK;
  }
  const int NC = nc[0] + (P.nseg > 1 ? nc[1] : 0);

  float4 ra[C::NSA], rb[C::NSB];
  bool a_kc = true, b_kc = false;

  auto gload = [&](int c) {
    const int s = (c < nc[0]) ? 0 : 1;
    const Segment& sg = P.seg[s];
    const int k0 = kb[s] + (s ? c - nc[0] : c) * kBK;
    a_kc = sg.a.k_contig;
    b_kc = sg.b.k_contig;
#pragma unroll
    for (int i = 0; i < C::NSA; ++i) {
      int mn, k;
      slot_coords<C::BM>(sg.a, tid + i * kThreads, mn, k);
      ra[i] = load_slot(sg.a, m0 + mn, k0 + k, ke[s], nullptr);
    }
#pragma unroll
    for (int i = 0; i < C::NSB; ++i) {
      int mn, k;
      slot_coords<C::BN>(sg.b, tid + i * kThreads, mn, k);
      rb[i] = load_slot(sg.b, n0 + mn, k0 + k, ke[s], sg.kscale);
    }
  };
  auto lstore = [&](int buf, int c) {
    const int s = (c < nc[0]) ? 0 : 1;
    const Segment& sg = P.seg[s];
    float* As = lds + buf * (C::LDA + C::LDB) * kBK;
    float* Bs = As + C::LDA * kBK;
#pragma unroll
    for (int i = 0; i < C::NSA; ++i) {
      int mn, k;
      slot_coords<C::BM>(sg.a, tid + i * kThreads, mn, k);
      store_slot<C::LDA>(As, a_kc, mn, k, ra[i]);
    }
#pragma unroll
    for (int i = 0; i < C::NSB; ++i) {
      int mn, k;
      slot_coords<C::BN>(sg.b, tid + i * kThreads, mn, k);
      store_slot<C::LDB>(Bs, b_kc, mn, k, rb[i]);
    }
  };

  if (NC > 0) {
    gload(0);
    lstore(0, 0);
  }
  __syncthreads();
  const int khalf = lane >> 5, l31 = lane & 31;
  for (int c = 0; c < NC; ++c) {
    if (c + 1 < NC) gload(c + 1);
    const float* As = lds + (c & 1) * (C::LDA + C::LDB) * kBK;
    const float* Bs = As + C::LDA * kBK;
    constexpr int KW = kBK / C::WK;
#pragma unroll
    for (int kk = 0; kk < KW; kk += 2) {
      const int krow = wk * KW + kk + khalf;
      float a[C::TM], b[C::TN];
#pragma unroll
      for (int i = 0; i < C::TM; ++i) a[i] = As[krow * C::LDA + wm0 + i * 32 + l31];
#pragma unroll
      for (int j = 0; j < C::TN; ++j) b[j] = Bs[krow * C::LDB + wn0 + j * 32 + l31];
#pragma unroll
      for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int j = 0; j < C::TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (c + 1 < NC) lstore((c + 1) & 1, c + 1);
    __syncthreads();
  }

  // ---- stage the accumulators to LDS in row-major [BM][LDC] (one image per k-wave)
  float* Cs = lds + wk * C::BM * C::LDC;
#pragma unroll
  for (int i = 0; i < C::TM; ++i)
#pragma unroll
    for (int j = 0; j < C::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        Cs[row * C::LDC + wn0 + j * 32 + l31] = acc[i][j][r];
      }
  __syncthreads();

  // ---- epilogue: each thread owns float4 groups of a row
  constexpr int GPR = C::BN / 4;                 // groups per row (8, 16 or 32 consecutive lanes)
  constexpr int PASSES = C::BM * GPR / kThreads;
  const long long soff = (long long)split * P.split_stride;
#pragma unroll 1
  for (int ps = 0; ps < PASSES; ++ps) {
    const int gidx = tid + ps * kThreads;
    const int row = gidx / GPR, c4 = gidx % GPR;
    float4 v4 = *reinterpret_cast<const float4*>(lds + row * C::LDC + 4 * c4);
#pragma unroll
    for (int w = 1; w < C::WK; ++w) {
      const float4 o = *reinterpret_cast<const float4*>(lds + (w * C::BM + row) * C::LDC + 4 * c4);
      v4.x += o.x; v4.y += o.y; v4.z += o.z; v4.w += o.w;
    }
    float v[4] = {v4.x, v4.y, v4.z, v4.w};
    const int m = m0 + row, nb = n0 + 4 * c4;
    const bool mrow = m < P.M;

    if (P.epi == EPI_STORE) {
      if (mrow) {
        float* dst;
        if (P.bias_row_out && m == P.M - 1) dst = P.bias_row_out + soff + nb;
        else dst = P.C + soff + (long long)m * P.ldc + nb;
        const float rs = P.rowscale ? P.rowscale[m] : 1.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = nb + j;
          if (n < P.N) {
            float x = v[j];
            if (P.bias) x += P.bias[n];
            if (P.addsrc) x += P.addsrc[(long long)(m / P.add_div) * P.ld_add + n];
            x += P.addconst;
            if (P.relu) x = fmaxf(x, 0.f);
            if (P.mask) x = P.mask[(long long)m * P.ld_mask + n] > 0.f ? x : 0.f;
            v[j] = x * rs;
          }
        }
        if (nb + 3 < P.N && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
          *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (nb + j < P.N) dst[j] = v[j];
        }
      }
    } else {  // EPI_BERNOULLI: lambda -> (sigmoid(lambda) - x, sum_d x*lambda - softplus(lambda))
      float rsum = 0.f;
      if (mrow) {
        const unsigned char* xr = P.x + (long long)(m / P.x_div) * P.ldx;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = nb + j;
          if (n < P.N) {
            const float lam = v[j] + P.bias[n] + P.addconst;
            const float xv = (float)xr[n];
            rsum += xv * lam - softplusf_(lam);
            v[j] = sigmoidf_(lam) - xv;
          }
        }
        if (P.C) {
          float* dst = P.C + (long long)m * P.ldc + nb;
          if (nb + 3 < P.N && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
            *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (nb + j < P.N) dst[j] = v[j];
          }
        }
      }
#pragma unroll
      for (int o = GPR / 2; o > 0; o >>= 1) rsum += __shfl_xor(rsum, o, 64);
      if (mrow && c4 == 0) P.part[(long long)m * P.nparts + tn] = rsum;
    }
  }
}

}  // namespace gmvae
