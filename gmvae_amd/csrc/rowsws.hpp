// Row-panel layers over tens of thousands of rows with the WEIGHT STATIONARY (the general schedule at R = B S >= 2048 rows; config-5
// shard: R = 25,600, widths 64 / 128 / 512).  out[R][N] = epi(A[R][K] W) (NN, W [K][N]) or epi(A[R][K] W^T) (NT, W [N][K]: a data
// gradient, scripts' reverse of fcnet layers), K and N in the dozens to hundreds -- a few GFLOP moving 60 - 120 MB: HBM-bound.
// rows_nn_bf6 (skinny.hpp) gives a wave RT row tiles x 64 columns and lets it load AND SPLIT its W fragments per unit: at the
// config-5 sizes 3,200 units each re-splitting a [64 x 64] weight block, one unit per wave, no prefetch -- 28 - 35 us for layers
// whose bytes take 13 (fwd_y_layers, fwd_dec); the grouped fp32-MFMA GEMM ran the thin data gradients at 28 % of its peak (dhg =
// dqp Wg1^T, K = 128: four 32-deep rounds per tile, 75 us for 117 MB).
//
// Here a wave keeps ITS slice of the weight -- KST k-steps of 32 x NTL column tiles of 16, KST NTL = 8 -- as bf16 piece fragments in
// 96 registers for the whole launch (split once), and walks row tiles of 16 rows: two 16-byte loads per k-step bring its A rows
// (prefetched one tile ahead, with the epilogue's addend / mask rows), 7 vector instructions per value split them, 6 KST NTL = 48
// v_mfma_f32_16x16x32_bf16 multiply (exact piece products, smallest first, as sk_mma6), and the lane stores NTL consecutive columns
// of 4 rows.  Waves are independent (no LDS, no barrier): global wave g owns column slice g % ns and the row tiles g / ns,
// g / ns + nw / ns, ...; the launch is ONE resident wave of workgroups (a workgroup per CU).
//   rows_ws<2, 4, false>: K = 64, slices of 64 columns (y / z -> 512 hidden units; the prior net beside them: np = 2)
//   rows_ws<4, 2, true>:  K = 128, slices of 32 columns (dhg = dqp Wg1^T under the ReLU mask; dy's prior part dpp Wp^T)
// rows_ws_k8: K = 512 (+ a second segment of 128), N = 64 (dz = dhd Wd0^T; dy = dhg Wg0y^T + dpp Wp^T): a WORKGROUP walks the row tiles, wave w owns contraction slice
// [64 w, 64 w + 64) with its [64 x 64] weight block as fragments; the eight partial tiles meet in LDS in wave order (two buffers:
// one barrier per tile).
#pragma once
#include "skinny.hpp"

namespace gmvae {

constexpr int kRwsMaxWaves = 4096;     // waves of a rows_ws launch (a partial maximum each: the workspace's pscale areas)

struct RwsProb {
  const float* W;                   // NN: [K][ldw]; NT: [N][ldw]
  const float* W2;                  // rows_ws_k8 only: a second NT weight [N][ldw2] over 128 more contraction steps (A2), or null
  int ldw2;
  const float *bias, *addsrc, *mask;   // [N] or null; [R / add_div][ld_add] or null; [R][ld_mask] (keep where > 0) or null
  float* out;                       // [R][N]
  unsigned* amax;                   // or null: word [global wave] receives the bits of the largest |out| that wave wrote (gemm.hpp amax_final)
  int N, ldw, relu, ld_add, add_div, ld_mask;
};
struct RwsArgs {
  const float* A;                   // [R][lda]
  const float* A2;                  // rows_ws_k8 only: [R][lda2], 128 columns (with p[0].W2), or null
  int lda2;
  int lda, R, np;
  int ns0, ns;                      // column slices of p[0]; of both problems
  RwsProb p[2];
};

template <int NTL> struct RwsVec;
template <> struct RwsVec<4> { typedef float4 T; };
template <> struct RwsVec<2> { typedef float2 T; };
template <int NTL> __device__ __forceinline__ void rws_get(const typename RwsVec<NTL>::T& q, float (&v)[NTL]);
template <> __device__ __forceinline__ void rws_get<4>(const float4& q, float (&v)[4]) { v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
template <> __device__ __forceinline__ void rws_get<2>(const float2& q, float (&v)[2]) { v[0] = q.x; v[1] = q.y; }
template <int NTL> __device__ __forceinline__ typename RwsVec<NTL>::T rws_put(const float (&v)[NTL]);
template <> __device__ __forceinline__ float4 rws_put<4>(const float (&v)[4]) { return make_float4(v[0], v[1], v[2], v[3]); }
template <> __device__ __forceinline__ float2 rws_put<2>(const float (&v)[2]) { return make_float2(v[0], v[1]); }

// "this value is needed NOW": the compiler otherwise leaves a prefetched side row in the register it was loaded into, waits for it
// at its first use in the NEXT trip -- where the counts of the two paths into the loop differ -- and so for every older store
__device__ __forceinline__ void rws_pin(float4& q) { asm volatile("" : "+v"(q.x), "+v"(q.y), "+v"(q.z), "+v"(q.w)); }
__device__ __forceinline__ void rws_pin(float2& q) { asm volatile("" : "+v"(q.x), "+v"(q.y)); }

// a wave's weight block as piece fragments: k-step ks, column tile t (strided columns c0 + t of lane group ln), pieces hi / mid / lo
template <int KST, int NTL, bool NT>
__device__ __forceinline__ void rws_weight(const float* __restrict__ W, const int ldw, const int k00, const int c0, const int lk,
                                           sk_bf16x8 (&Wp)[KST][NTL][3]) {
  typedef typename RwsVec<NTL>::T V;
#pragma unroll
  for (int ks = 0; ks < KST; ++ks) {
    const int k0 = k00 + 32 * ks + 8 * lk;
    if constexpr (!NT) {
      float vv[8][NTL];
#pragma unroll
      for (int e = 0; e < 8; ++e) rws_get<NTL>(*reinterpret_cast<const V*>(W + (long long)(k0 + e) * ldw + c0), vv[e]);
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = vv[e][t];
        sk_pieces(v, Wp[ks][t]);
      }
    } else {
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        const float4 q0 = *reinterpret_cast<const float4*>(W + (long long)(c0 + t) * ldw + k0),
                     q1 = *reinterpret_cast<const float4*>(W + (long long)(c0 + t) * ldw + k0 + 4);
        const float v[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        sk_pieces(v, Wp[ks][t]);
      }
    }
  }
}

template <int KST, int NTL, bool NT>
__global__ __launch_bounds__(kSkThreads) void rows_ws(const RwsArgs a) {
  static_assert(KST * NTL == 8, "a wave's weight block is 8 fragments of 3 pieces: 96 registers");
  typedef typename RwsVec<NTL>::T V;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int ln = lane & 15, lk = lane >> 4;
  const int R = a.R, lda = a.lda;
  const int g = (int)blockIdx.x * kSkWaves + wave, nw = (int)gridDim.x * kSkWaves;
  const int ngr = nw / a.ns, grp = g / a.ns, sl = g - grp * a.ns;
  const RwsProb& P = sl < a.ns0 ? a.p[0] : a.p[1];
  if (grp >= ngr) {                               // (nw % ns waves have no slice)
    if (P.amax && lane == 0) P.amax[g] = 0u;
    return;
  }
  const int c0 = (sl < a.ns0 ? sl : sl - a.ns0) * 16 * NTL + NTL * ln;
  const int N = P.N;
  sk_bf16x8 Wp[KST][NTL][3];
  rws_weight<KST, NTL, NT>(P.W, P.ldw, 0, c0, lk, Wp);
  float bias[NTL];
#pragma unroll
  for (int t = 0; t < NTL; ++t) bias[t] = 0.f;
  if (P.bias) rws_get<NTL>(*reinterpret_cast<const V*>(P.bias + c0), bias);
  // the epilogue's side rows (an addend, or the kept activation a ReLU mask is read from; one of them at most).  Without either the
  // loads still go out -- of one resident line of the weight (row stride 0) -- so that every trip of the loop issues the SAME
  // number of memory instructions: only then can the wait for the prefetched rows be a counted one (below)
  const bool is_add = P.addsrc != nullptr, is_mask = !is_add && P.mask != nullptr;
  const float* const side = is_add ? P.addsrc : is_mask ? P.mask : P.W;
  const int ld_side = is_add ? P.ld_add : is_mask ? P.ld_mask : 0, div_side = is_add ? P.add_div : 1;
  const int soff = (is_add || is_mask) ? c0 : 0;
  const int ntile = (R + 15) >> 4, nfull = R >> 4;
  float4 an[KST][2];
  V sn[4], sc[4];
  sk_bf16x8 Ap[KST][3];
  auto fetch = [&](const int rt_) {
    const int rt = min(rt_, ntile - 1);            // (past the end: the last tile again, unused)
    const float* const ar = a.A + (long long)min(16 * rt + ln, R - 1) * lda + 8 * lk;
#pragma unroll
    for (int ks = 0; ks < KST; ++ks) {
      an[ks][0] = *reinterpret_cast<const float4*>(ar + 32 * ks);
      an[ks][1] = *reinterpret_cast<const float4*>(ar + 32 * ks + 4);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
      sn[r] = *reinterpret_cast<const V*>(side + (long long)(min(16 * rt + 4 * lk + r, R - 1) / div_side) * ld_side + soff);
  };
  auto split = [&]() {
#pragma unroll
    for (int ks = 0; ks < KST; ++ks) {
      const float v[8] = {an[ks][0].x, an[ks][0].y, an[ks][0].z, an[ks][0].w, an[ks][1].x, an[ks][1].y, an[ks][1].z, an[ks][1].w};
      sk_pieces(v, Ap[ks]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { sc[r] = sn[r]; rws_pin(sc[r]); }
  };
  unsigned vmax = 0;
  // one row tile: accumulator [t][r] = out[16 rt + 4 lk + r][c0 + t]
  auto tile = [&](const int rt, const bool tail) {
    f32x4 acc[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KST; ++ks)
#pragma unroll
      for (int t = 0; t < NTL; ++t) sk_mma6(Ap[ks], Wp[ks][t], acc[t]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * rt + 4 * lk + r;
      float v[NTL], s[NTL];
      rws_get<NTL>(sc[r], s);
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        v[t] = acc[t][r] + bias[t];
        if (is_add) v[t] += s[t];
        if (P.relu) v[t] = relu_nan(v[t]);
        if (is_mask) v[t] = s[t] > 0.f ? v[t] : 0.f;
        if (!tail || row < R) { const unsigned b = __float_as_uint(v[t]) & 0x7fffffffu; vmax = b > vmax ? b : vmax; }
      }
      if (!tail || row < R) *reinterpret_cast<V*>(P.out + (long long)row * N + c0) = rws_put<NTL>(v);
    }
  };
  // The loop.  A trip: the NEXT tile's rows are requested first (every trip the same twelve / eight loads: past the end the last
  // tile again), then the tile whose pieces are in registers runs (48 MFMAs, 4 unconditional stores: full tiles only -- a ragged
  // last tile runs once, behind the loop), then the requested rows are waited for and split.  The stores are younger than the
  // loads, so that wait is vmcnt(4): the stores stay in flight.  (First form: multiply, store under `row < R`, a conditional
  // request -- the compiler had to wait vmcnt(0) at the loop's end and every trip paid its stores' round trip to memory.)
  int rt = grp;
  fetch(rt);
  split();
  for (; rt < nfull; rt += ngr) {
    fetch(rt + ngr);
    __builtin_amdgcn_sched_barrier(0);
    tile(rt, false);
    __builtin_amdgcn_sched_barrier(0);
    split();
  }
  if (rt == nfull && (R & 15)) tile(rt, true);
  if (P.amax) { vmax = wave_umax(vmax); if (lane == 0) P.amax[g] = vmax; }
}

// K = 512 (NT), N = 64: a workgroup per row tile, wave w the contraction slice [64 w, 64 w + 64); addsrc: out = addsrc + A W^T.
// The eight partial tiles meet in LDS; thread q sums the pair (row q >> 5, columns 2 (q & 31) ..) in wave order and stores it.
__global__ __launch_bounds__(kSkThreads) void rows_ws_k8(const RwsArgs a) {
  __shared__ __attribute__((aligned(16))) float red[2][kSkWaves * 4 * 64 * 4];      // [buffer][wave][r][lane][t]: 2 x 32 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 15, lk = lane >> 4;
  const int R = a.R, lda = a.lda;
  const RwsProb& P = a.p[0];
  sk_bf16x8 Wp[2][4][3];
  rws_weight<2, 4, true>(P.W, P.ldw, 64 * wave, 4 * ln, lk, Wp);
  // a second segment of 128 contraction steps (dy's prior part: dpp Wp^T beside dhg Wg0y^T): k-step wave & 3 of it is a THIRD k-step
  // of waves 0..3; waves 4..7 run the same instructions on a zero block, so that every wave issues the same loads and multiplies
  // (the loop's counted waits and the barrier need that) -- 24 of 72 instructions per tile idle in half the waves
  const bool seg2 = a.A2 != nullptr;
  sk_bf16x8 Wq[1][4][3];
  if (seg2 && wave < 4) {
    rws_weight<1, 4, true>(P.W2, P.ldw2, 32 * wave, 4 * ln, lk, Wq);
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int q = 0; q < 3; ++q) Wq[0][t][q] = __builtin_bit_cast(sk_bf16x8, sk_u32x4{0u, 0u, 0u, 0u});
  }
  const float* const A2b = seg2 ? a.A2 + 32 * (wave & 3) + 8 * lk : a.A + 8 * lk;      // (no second segment: a resident line, multiplied by zeros)
  const long long lda2 = seg2 ? a.lda2 : 0;
  const int ntile = (R + 15) >> 4, nfull = R >> 4, nwg = (int)gridDim.x;
  const int frow = tid >> 5, fcp = tid & 31;       // the pair this thread finishes
  const int foff = ((frow & 3) * 64 + (frow >> 2) * 16 + (fcp >> 1)) * 4 + 2 * (fcp & 1);
  const bool is_add = P.addsrc != nullptr;
  const float* const side = is_add ? P.addsrc : P.W;        // (as rows_ws: the same number of loads with or without an addend)
  const int ld_side = is_add ? P.ld_add : 0, div_side = is_add ? P.add_div : 1;
  float4 an[3][2];
  float2 sn, sc;
  sk_bf16x8 Ap[3][3];
  auto fetch = [&](const int rt_) {
    const int rt = min(rt_, ntile - 1);
    const float* const ar = a.A + (long long)min(16 * rt + ln, R - 1) * lda + 64 * wave + 8 * lk;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      an[ks][0] = *reinterpret_cast<const float4*>(ar + 32 * ks);
      an[ks][1] = *reinterpret_cast<const float4*>(ar + 32 * ks + 4);
    }
    const float* const a2 = A2b + (long long)min(16 * rt + ln, R - 1) * lda2;
    an[2][0] = *reinterpret_cast<const float4*>(a2);
    an[2][1] = *reinterpret_cast<const float4*>(a2 + 4);
    sn = *reinterpret_cast<const float2*>(side + (long long)(min(16 * rt + frow, R - 1) / div_side) * ld_side + 2 * fcp);
  };
  auto split = [&]() {
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      const float v[8] = {an[ks][0].x, an[ks][0].y, an[ks][0].z, an[ks][0].w, an[ks][1].x, an[ks][1].y, an[ks][1].z, an[ks][1].w};
      sk_pieces(v, Ap[ks]);
    }
    sc = sn;
    rws_pin(sc);
  };
  int buf = 0;
  auto tile = [&](const int rt, const bool tail) {
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int t = 0; t < 4; ++t) sk_mma6(Ap[ks], Wp[ks][t], acc[t]);
    if (seg2) {                                    // (uniform over the launch)
#pragma unroll
      for (int t = 0; t < 4; ++t) sk_mma6(Ap[2], Wq[0][t], acc[t]);
    }
    float* const rb = red[buf];
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *reinterpret_cast<float4*>(rb + ((wave * 4 + r) * 64 + lane) * 4) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
    __syncthreads();                               // (the other buffer's readers passed this barrier one tile ago)
    float v0 = 0.f, v1 = 0.f;
#pragma unroll
    for (int w = 0; w < kSkWaves; ++w) {
      const float2 q = *reinterpret_cast<const float2*>(rb + w * 1024 + foff);
      v0 += q.x; v1 += q.y;
    }
    if (is_add) { v0 += sc.x; v1 += sc.y; }
    const int row = 16 * rt + frow;
    if (!tail || row < R) *reinterpret_cast<float2*>(P.out + (long long)row * P.N + 2 * fcp) = make_float2(v0, v1);
    buf ^= 1;
  };
  // (the loop's shape: rows_ws above -- here one store per trip stays in flight)
  int rt = (int)blockIdx.x;
  fetch(rt);
  split();
  for (; rt < nfull; rt += nwg) {
    fetch(rt + nwg);
    __builtin_amdgcn_sched_barrier(0);
    tile(rt, false);
    __builtin_amdgcn_sched_barrier(0);
    split();
  }
  if (rt == nfull && (R & 15)) tile(rt, true);
}

}  // namespace gmvae
