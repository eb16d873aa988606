// dw_adam: every weight (and bias) gradient of the single-hidden-layer step AND the TF-Adam update in ONE launch.
//
// Replaces, on the single-device train-graph path, the grouped split-K weight-gradient launch + finalize_adam
// (scripts/runners.py:181-183: opt.compute_gradients' MatMul-gradient nodes and apply_gradients): a workgroup owns a
// [64 fan-in rows] x [16 fan-out columns] tile of ONE weight tensor for the WHOLE batch contraction, so no partial
// slabs exist: the tile's gradient is complete in the workgroup and the optimizer runs in the epilogue (13 MB of slab
// writes + reads, a kernel boundary and finalize_adam's cold start disappear).
//
//   dW[m][n] = sum_b A[b][m] dY[b][n]       A = the layer's input rows (uint8 batch or fp32 activations), dY = the
//                                           pre-activation gradients mega2_fwd_bwd left; both k-major (row = batch index)
//
// Operands go global -> registers -> matrix core, no LDS staging: with v_mfma_f32_16x16x4_f32's k index = 4 consecutive
// batch rows (lane / 16), a lane's A operand for FOUR m-tiles is one 16-byte load A[b][m0 + 4 (lane % 16) .. +3]
// (tile t takes the rows m = m0 + 4 i + t: a strided tile), its B operand one 4-byte load dY[b][n0 + lane % 16].
// The 8 waves split the batch rows; partial tiles meet in LDS in a fixed order (bit-reproducible).  A lane ends with
// the 4 x 1 block dW[mb .. mb+3][n]: exactly one 16-byte unit of the row-interleaved operand images mega2_fwd_bwd reads
// (kernels.hpp img_dst kinds 2 / 4), so the update leaves through coalesced stores.  The bias gradient (column sums
// of dY) rides on the tiles of the first tile row.
#pragma once
#include <type_traits>

#include "chain.hpp"

namespace gmvae {

constexpr int kDwMaxT = 10;
constexpr int kDwThreads = 512, kDwWaves = 8;

struct DwTensor {
  const void* A;               // [B][lda] uint8 or fp32
  const float* dY;             // [B][ldy]
  int lda, ldy, M, N, a_u8;
  int mu;                      // 16-row strided tiles per workgroup tile (4: 64 rows; 2: 32 rows -- more, lighter workgroups)
  int w_off, b_off;            // flat parameter offsets of W [M][N] and of the bias [N] (b_off < 0: none)
  int tiles_n, tile_begin;
  int k1, base1, ld1, chunk1, which1;    // operand images of W (kind -1 = none): kernels.hpp img_dst
  int k2, base2, ld2, chunk2, which2;
  int bk, bbase, bchunk, bwhich;         // image of the bias (bk < 0: none; kinds 0 / 6)
};

constexpr int kDwMaxTiles = 512;
struct DwArgs {
  int ntens, total_tiles, B;
  int u8x3;                    // the uint8-activation problems on the bf16 matrix cores (exact bf16x3 form; GMVAE_NO_DW_U8X3 = fp32 MFMA)
  // XCD-aware launch order (speed only): workgroups are dealt round-robin over the 8 XCDs, each with an L2 of its own, so
  // slot b (XCD b % 8) runs tile perm[b], chosen on the host so that the tiles of one XCD share operand columns (an
  // x-problem's XCD keeps the x columns of "its" row blocks, the decoder layer's the g columns of "its" column tiles)
  // instead of every XCD fetching every operand once (37.6 MB of fabric reads + writes per launch before, round2 PMC)
  unsigned short perm[kDwMaxTiles];
  unsigned long long* dbg;     // diagnostic: [block][8] wall-clock stamps (tools/dwstamps.py) or null
  const float* lr_t;           // this step's Adam step size alpha_t, left by mega2_fwd_bwd (mega.hpp MegaArgs::lr_t_out), or null:
  float ln_b1, ln_b2;          // then alpha_t = lr sqrt(-expm1(t ln b2)) / (-expm1(t ln b1)) per thread (ln b rounded from double)
  // VAE_GMP: the learned mixture prior's variables (loc, raw scale, mixture logits: one contiguous parameter range) have
  // no matrix-product gradient -- mega_fwd_bwd leaves one partial per panel (fa.gmp_part); gmp_blocks extra workgroups
  // (after the tail block) sum them in panel order, apply the update and write the variables' LDS-image copies
  int gmp_blocks, gmp_nmap;
  ImgMap gmp_map[3];
  DwTensor t[kDwMaxT];
  FinalArgs fa;                // p, m, v, grads, Adam constants, loss-tail inputs, counters, images
};

// One wave's share [b_lo, b_hi) of the batch contraction for a 64 x 16 tile.  ALL of its operand loads (32 k-steps at
// B = 1024) are in flight before the first MFMA, and they are BRANCH-FREE (clamped address, value selected
// afterwards): a load under `if (in range)` makes the compiler wait inside the branch, one memory round trip per load
// (measured: 13 us for this loop).  The uint8 operand stays packed (one register per k-step) until its MFMAs.
// KB: k-steps (of 4 batch rows) requested per batch of loads: 32 (a wave's 128 rows at B = 1024), or 8 for SMALL batches --
// at B <= 256 a wave's share is <= 32 rows, and 32 clamped k-steps per lane were 56 wasted loads of 64 in front of the few
// matrix instructions that count (BASELINE configs[0] / configs[1]: B = 100 / 256).
// SC = true (mega3_step, mega3.hpp): the fp32 operands were written by OTHER workgroups of the SAME launch (write-through, behind
// a flag): they are read with agent-scope (sc1) loads, which another XCD's L2 cannot serve stale.  The uint8 batch is never
// written by a launch that reads it: plain loads.
#ifndef GMVAE_SC_IMPL
#define GMVAE_SC_IMPL 0      // 0: relaxed agent-scope atomic loads; 1 (diagnostic, WRONG results): plain loads; 2: sc1 global loads in asm-free form
#endif
template <bool SC>
__device__ __forceinline__ float ldg_f(const float* p) {
#if GMVAE_SC_IMPL == 1
  return *p;
#elif GMVAE_SC_IMPL == 2
  if constexpr (SC) return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  else return *p;
#else
  if constexpr (SC) return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  else return *p;
#endif
}
template <bool SC, class T>
__device__ __forceinline__ T ldg_a(const void* p) {
#if GMVAE_SC_IMPL == 1
  return *reinterpret_cast<const T*>(p);
#elif GMVAE_SC_IMPL == 2
  if constexpr (!SC) return *reinterpret_cast<const T*>(p);
  else if constexpr (sizeof(T) == 4)
    return __builtin_bit_cast(T, __hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  else {                                           // an 8-byte operand as two 4-byte agent-scope loads
    const unsigned lo = __hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned hi = __hip_atomic_load(reinterpret_cast<const unsigned*>(p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __builtin_bit_cast(T, ((unsigned long long)hi << 32) | lo);
  }
#else
  if constexpr (!SC) return *reinterpret_cast<const T*>(p);
  else if constexpr (sizeof(T) == 4)
    return __builtin_bit_cast(T, __hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  else {
    static_assert(sizeof(T) == 8, "agent-scope operand loads are 4 or 8 bytes");
    return __builtin_bit_cast(T, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  }
#endif
}
template <bool U8, int MU, int KB = 32, bool SC = false, class Mid>
__device__ __forceinline__ void dw_contract(const void* __restrict__ Ap, const float* __restrict__ dY, const int lda, const int ldy,
                                            const int M, const int N, const int m0, const int n0, const int b_lo, const int b_hi,
                                            const int ln, const int lk, f32x4 (&acc)[4], float& cs, Mid mid) {
  static_assert(MU == 4 || MU == 2 || MU == 1, "1, 2 or 4 strided 16-row tiles per workgroup tile");
  static_assert(!SC || U8 || MU <= 2, "agent-scope fp32 operand loads: at most 8 bytes");
  const int ma = m0 + MU * ln;                   // this lane's MU fan-in rows (one per strided tile)
  const bool a_ok = ma < M, n_ok = n0 + ln < N;  // (M is a multiple of 4 or the source rows are padded to one)
  const int mac = min(ma, ((M + 3) & ~3) - MU), nc = min(n0 + ln, N - 1);
  const unsigned char* const A8 = static_cast<const unsigned char*>(Ap);
  const float* const A32 = static_cast<const float*>(Ap);
  typedef typename std::conditional<MU == 4, typename std::conditional<U8, unsigned, float4>::type,
                                    typename std::conditional<MU == 2, typename std::conditional<U8, unsigned short, float2>::type,
                                                              typename std::conditional<U8, unsigned char, float>::type>::type>::type AT;
  auto mfma4 = [&](const AT& avs, const float bq) {
    float aq[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (U8) {
      const unsigned w = avs;
#pragma unroll
      for (int t = 0; t < MU; ++t) aq[t] = (float)((w >> (8 * t)) & 0xff);
    } else if constexpr (MU == 4) {
      aq[0] = avs.x; aq[1] = avs.y; aq[2] = avs.z; aq[3] = avs.w;
    } else if constexpr (MU == 2) {
      aq[0] = avs.x; aq[1] = avs.y;
    } else {
      aq[0] = avs;
    }
#pragma unroll
    for (int t = 0; t < MU; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_ok ? aq[t] : 0.f, bq, acc[t], 0, 0, 0);
    cs += bq;
  };
  if (b_hi - b_lo == 4 * KB) {
    // the usual case (B a multiple of 256): one batch, rows b_lo + 4 s + lk, no clamps -- the addresses are a base plus
    // compile-time multiples of the row stride (the clamped form spent ~6 VALU instructions per load, 64 loads per lane)
    AT av[KB];
    float bvv[KB];
    const unsigned char* const a8 = A8 + (long long)(b_lo + lk) * lda + mac;
    const float* const a32 = A32 + (long long)(b_lo + lk) * lda + mac;
    const float* const dy = dY + (long long)(b_lo + lk) * ldy + nc;
#pragma unroll
    for (int s = 0; s < KB; ++s) {
      if constexpr (U8) av[s] = *reinterpret_cast<const AT*>(a8 + (long long)(4 * s) * lda);
      else av[s] = ldg_a<SC, AT>(a32 + (long long)(4 * s) * lda);
      bvv[s] = ldg_f<SC>(dy + (long long)(4 * s) * ldy);
    }
#pragma unroll
    for (int s = 0; s < KB; ++s) mfma4(av[s], n_ok ? bvv[s] : 0.f);
    return;
  }
  for (int b0 = b_lo; b0 < b_hi; b0 += 4 * KB) {
    AT av[KB];
    float bvv[KB];
#pragma unroll
    for (int s = 0; s < KB; ++s) {
      const int bc = min(b0 + 4 * s + lk, b_hi - 1);
      if constexpr (U8) av[s] = *reinterpret_cast<const AT*>(A8 + (long long)bc * lda + mac);
      else av[s] = ldg_a<SC, AT>(A32 + (long long)bc * lda + mac);
      bvv[s] = ldg_f<SC>(dY + (long long)bc * ldy + nc);
    }
#pragma unroll
    for (int s = 0; s < KB; ++s) mfma4(av[s], (b0 + 4 * s + lk < b_hi && n_ok) ? bvv[s] : 0.f);   // a zero B operand also voids the clamped A values
  }
}

// The uint8-activation problems (dW = x^T dY, x the batch) on the bf16 matrix cores, EXACT: x is exact in bf16 (8
// significant bits) and dY = hi + mid + lo, three bf16 pieces that reproduce its 24-bit significand (truncation splits,
// exact residuals), so every product is exact and the fp32 accumulation is the only rounding -- as on the fp32 MFMA path,
// at 3 v_mfma_f32_16x16x32_bf16 (16 cycles each) per 32 batch rows and tile instead of 8 v_mfma_f32_16x16x4_f32 (32
// cycles each): 2 waves x 128 MFMAs x 32 cycles per SIMD (3.5 us, the launch's critical path) become 2 x 48 x 16.
// No transposition is needed for the 8-elements-per-lane operands: WHICH batch row is contraction index (lane group h,
// element j) of an MFMA is free as long as A and B agree -- row b_lo + 32 blk + 4 j + h, i.e. exactly the rows the
// register-direct loads of the fp32 form already hold (k-step s = 8 blk + j of lane group h).  Per block of 8 k-steps a
// lane repacks: A tile t = bf16(byte t of its 8 packed words) (v_cvt_f32_ubyte + v_perm), B = the three pieces of its 8
// dY values (and / sub / perm): ~100 VALU instructions beside 12 MFMAs.
__device__ __forceinline__ unsigned pack_hi16(const float f1, const float f0) {     // (bf16 bits of f0) | (bf16 bits of f1) << 16, truncating
  return __builtin_amdgcn_perm(__float_as_uint(f1), __float_as_uint(f0), 0x07060302u);
}
__device__ __forceinline__ void dw_contract_u8x3(const unsigned char* __restrict__ A8, const float* __restrict__ dY, const int lda,
                                                 const int ldy, const int M, const int N, const int m0, const int n0, const int b_lo,
                                                 const int ln, const int lk, f32x4 (&acc)[4], float& cs) {
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const int ma = m0 + 4 * ln;
  const bool a_ok = ma < M, n_ok = n0 + ln < N;
  const int mac = min(ma, ((M + 3) & ~3) - 4), nc = min(n0 + ln, N - 1);
  constexpr int KB = 32;
  unsigned av[KB];
  float bvv[KB];
  const unsigned char* const a8 = A8 + (long long)(b_lo + lk) * lda + mac;
  const float* const dy = dY + (long long)(b_lo + lk) * ldy + nc;
#pragma unroll
  for (int s = 0; s < KB; ++s) {
    av[s] = *reinterpret_cast<const unsigned*>(a8 + (long long)(4 * s) * lda);
    bvv[s] = dy[(long long)(4 * s) * ldy];
  }
#pragma unroll
  for (int blk = 0; blk < KB / 8; ++blk) {
    u32x4 bh, bm, bl;
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
      const float v0 = n_ok ? bvv[8 * blk + 2 * jp] : 0.f, v1 = n_ok ? bvv[8 * blk + 2 * jp + 1] : 0.f;
      cs += v0; cs += v1;
      bh[jp] = pack_hi16(v1, v0);
      const float r0 = v0 - __uint_as_float(__float_as_uint(v0) & 0xffff0000u), r1 = v1 - __uint_as_float(__float_as_uint(v1) & 0xffff0000u);
      bm[jp] = pack_hi16(r1, r0);
      const float s0 = r0 - __uint_as_float(__float_as_uint(r0) & 0xffff0000u), s1 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
      bl[jp] = pack_hi16(s1, s0);
    }
    const bf16x8 Bh = __builtin_bit_cast(bf16x8, bh), Bm = __builtin_bit_cast(bf16x8, bm), Bl = __builtin_bit_cast(bf16x8, bl);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      u32x4 aw;
#pragma unroll
      for (int jp = 0; jp < 4; ++jp) {
        const float f0 = (float)((av[8 * blk + 2 * jp] >> (8 * t)) & 0xffu), f1 = (float)((av[8 * blk + 2 * jp + 1] >> (8 * t)) & 0xffu);
        aw[jp] = a_ok ? pack_hi16(f1, f0) : 0u;
      }
      const bf16x8 At = __builtin_bit_cast(bf16x8, aw);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(At, Bl, acc[t], 0, 0, 0);      // smallest pieces first
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(At, Bm, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(At, Bh, acc[t], 0, 0, 0);
    }
  }
}

// The same contraction in two parts, for mega3_step (mega3.hpp): the uint8 operand (the batch: final before the launch) is
// requested while the workgroup still waits for the pre-activation gradients; those follow with agent-scope loads.
__device__ __forceinline__ void dw_u8x3_load_a(const unsigned char* __restrict__ A8, const int lda, const int M, const int m0,
                                               const int b_lo, const int ln, const int lk, unsigned (&av)[32]) {
  const int mac = min(m0 + 4 * ln, ((M + 3) & ~3) - 4);
  const unsigned char* const a8 = A8 + (long long)(b_lo + lk) * lda + mac;
#pragma unroll
  for (int s = 0; s < 32; ++s) av[s] = *reinterpret_cast<const unsigned*>(a8 + (long long)(4 * s) * lda);
}
template <bool SC>
__device__ __forceinline__ void dw_u8x3_rest(const unsigned (&av)[32], const float* __restrict__ dY, const int ldy, const int M, const int N,
                                             const int m0, const int n0, const int b_lo, const int ln, const int lk, f32x4 (&acc)[4],
                                             float& cs) {
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const bool a_ok = m0 + 4 * ln < M, n_ok = n0 + ln < N;
  const int nc = min(n0 + ln, N - 1);
  constexpr int KB = 32;
  float bvv[KB];
  const float* const dy = dY + (long long)(b_lo + lk) * ldy + nc;
#pragma unroll
  for (int s = 0; s < KB; ++s) bvv[s] = ldg_f<SC>(dy + (long long)(4 * s) * ldy);
#pragma unroll
  for (int blk = 0; blk < KB / 8; ++blk) {
    u32x4 bh, bm, bl;
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
      const float v0 = n_ok ? bvv[8 * blk + 2 * jp] : 0.f, v1 = n_ok ? bvv[8 * blk + 2 * jp + 1] : 0.f;
      cs += v0; cs += v1;
      bh[jp] = pack_hi16(v1, v0);
      const float r0 = v0 - __uint_as_float(__float_as_uint(v0) & 0xffff0000u), r1 = v1 - __uint_as_float(__float_as_uint(v1) & 0xffff0000u);
      bm[jp] = pack_hi16(r1, r0);
      const float s0 = r0 - __uint_as_float(__float_as_uint(r0) & 0xffff0000u), s1 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
      bl[jp] = pack_hi16(s1, s0);
    }
    const bf16x8 Bh = __builtin_bit_cast(bf16x8, bh), Bm = __builtin_bit_cast(bf16x8, bm), Bl = __builtin_bit_cast(bf16x8, bl);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      u32x4 aw;
#pragma unroll
      for (int jp = 0; jp < 4; ++jp) {
        const float f0 = (float)((av[8 * blk + 2 * jp] >> (8 * t)) & 0xffu), f1 = (float)((av[8 * blk + 2 * jp + 1] >> (8 * t)) & 0xffu);
        aw[jp] = a_ok ? pack_hi16(f1, f0) : 0u;
      }
      const bf16x8 At = __builtin_bit_cast(bf16x8, aw);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(At, Bl, acc[t], 0, 0, 0);      // smallest pieces first
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(At, Bm, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(At, Bh, acc[t], 0, 0, 0);
    }
  }
}

// The fp32 contraction of a wave's 128 rows in two parts (mega3_step): the A operand (a forward activation, final long before
// the pre-activation gradients) is requested first; the dY values and the matrix instructions follow behind the tile's flag.
template <int MU>
struct DwA32 { typedef typename std::conditional<MU == 2, float2, float>::type T; };
template <int MU>
__device__ __forceinline__ void dw_f32_load_a(const float* __restrict__ A32, const int lda, const int M, const int m0, const int b_lo,
                                              const int ln, const int lk, typename DwA32<MU>::T (&av)[32]) {
  static_assert(MU == 2 || MU == 1, "32- or 16-row tiles");
  const int mac = min(m0 + MU * ln, ((M + 3) & ~3) - MU);
  const float* const a32 = A32 + (long long)(b_lo + lk) * lda + mac;
#pragma unroll
  for (int s = 0; s < 32; ++s) av[s] = *reinterpret_cast<const typename DwA32<MU>::T*>(a32 + (long long)(4 * s) * lda);
}
// (SC: the dY values with agent-scope loads -- for columns whose lines an earlier phase may have pulled into this XCD's L2
//  before their last bytes were written: the g columns of the lead's decoder tile, mega3.hpp)
template <int MU, bool SC = false>
__device__ __forceinline__ void dw_f32_rest(const typename DwA32<MU>::T (&av)[32], const float* __restrict__ dY, const int ldy, const int M,
                                            const int N, const int m0, const int n0, const int b_lo, const int ln, const int lk,
                                            f32x4 (&acc)[4], float& cs) {
  const bool a_ok = m0 + MU * ln < M, n_ok = n0 + ln < N;
  const int nc = min(n0 + ln, N - 1);
  float bvv[32];
  const float* const dy = dY + (long long)(b_lo + lk) * ldy + nc;
#pragma unroll
  for (int s = 0; s < 32; ++s) bvv[s] = SC ? ldg_f<true>(dy + (long long)(4 * s) * ldy) : dy[(long long)(4 * s) * ldy];
  // all 32 loads in flight before the first matrix instruction (with the A operand resident the scheduler otherwise sinks each
  // load in front of its use to save registers: 32 dependent round trips, 8.5 us measured)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < 32; ++s) {
    const float bq = n_ok ? bvv[s] : 0.f;
    if constexpr (MU == 2) {
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_ok ? av[s].x : 0.f, bq, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_ok ? av[s].y : 0.f, bq, acc[1], 0, 0, 0);
    } else {
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_ok ? av[s] : 0.f, bq, acc[0], 0, 0, 0);
    }
    cs += bq;
  }
}

// (Round 4, measured and reverted: [16 x 16] tiles for the decoder output layer -- 196 instead of 98 -- end at 7.4 us instead
//  of 9.6, but 338 workgroups need two per CU: at <= 128 registers the contraction's 64 in-flight operands spill (124 B of
//  scratch per lane) and the launch takes 14.6 us; at one per CU the last 83 tiles start when the first ones end: 15.7 us.)
__global__ __launch_bounds__(kDwThreads) void dw_adam(const DwArgs a) {
  __shared__ __attribute__((aligned(16))) float red[kDwWaves * 64 * 16];     // [wave][mt * 4 + r][lane]
  __shared__ float redcs[kDwWaves * 64];
  const FinalArgs& fa = a.fa;
  if (fa.span && threadIdx.x == 0) fa.span[2 * blockIdx.x] = wall_clock64();
#define DW_END() if (fa.span && threadIdx.x == 0) fa.span[2 * blockIdx.x + 1] = wall_clock64()
  const int bid = blockIdx.x;
#define DW_ST(i) if (a.dbg && threadIdx.x == 0) a.dbg[(size_t)blockIdx.x * 8 + (i)] = wall_clock64()
  DW_ST(0);
  if (bid >= a.total_tiles) {                    // the loss tail + counters, then the mixture prior's workgroups
    if (bid == a.total_tiles) finalize_tail_block(fa, reinterpret_cast<float(*)[256]>(red));
    else {                                         // mixture-prior variables: partials -> gradient -> TF-Adam -> image
      const int e = (bid - a.total_tiles - 1) * kDwThreads + (int)threadIdx.x;
      if (e < fa.gmp_len) {
        const long long i = fa.gmp_off + e;
        float g = 0.f;
        for (int k = 0; k < fa.gmp_n; ++k) g += fa.gmp_part[(long long)k * fa.gmp_len + e];
        fa.grads[i] = g;
        const bool poisoned = fa.err_word && *fa.err_word;
        if (fa.do_adam && !poisoned) {
          float lr_t;
          if (a.lr_t) lr_t = *a.lr_t;
          else {
            const float tf = (float)((fa.step_dev ? fa.step_dev[1] : 0ull) + 1ull);
            lr_t = fa.lr * sqrtf(-expm1f(tf * a.ln_b2)) / (-expm1f(tf * a.ln_b1));
          }
          float pp = fa.p[i], pm = fa.m[i], pv = fa.v[i];
          adam_update(pp, pm, pv, g, 1.f / fa.count, lr_t, 1.f - fa.b1, 1.f - fa.b2, fa.eps);
          fa.p[i] = pp; fa.m[i] = pm; fa.v[i] = pv;
#pragma unroll
          for (int k = 0; k < 3; ++k)
            if (k < a.gmp_nmap && i >= a.gmp_map[k].begin && i < a.gmp_map[k].end) {
              const ImgMap& mp = a.gmp_map[k];
              const unsigned off = (unsigned)(i - mp.begin);
              const int r = (int)(((unsigned long long)off * mp.magic) >> 32);
              const int c = (int)off - r * mp.cols;
              fa.img[mp.which][img_dst(mp.kind, mp.base, mp.ld, mp.chunk, r, c)] = pp;
            }
        }
      }
    }
    DW_ST(4);
    if (a.dbg && threadIdx.x == 0) a.dbg[(size_t)blockIdx.x * 8 + 5] = 99ull;
    DW_END();
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 15, lk = lane >> 4;
  // perm[slot] = (tensor << 10) | tile inside the tensor: ONE kernel-argument load names the tensor (a search through
  // tile_begin[] was a dependent round of scalar loads in front of every workgroup's first operand load)
  const int pv_ = a.perm[bid];
  const int ti = pv_ >> 10, tl = pv_ & 1023;
  const DwTensor& T = a.t[ti];
  const int M = T.M, N = T.N, lda = T.lda, ldy = T.ldy, B = a.B;
  const int tm = tl / T.tiles_n, tn = tl - tm * T.tiles_n;
  const int MUr = T.mu;
  const int m0 = tm * 16 * MUr, n0 = tn * 16;
  // ---- epilogue owners (threads 0..255): unit (lane slot l, r) = dW[mb .. mb+3][n]
  // (a lane group lk holds the 4 MU consecutive rows m0 + 4 MU lk + o, o = MU r + t for accumulator [tile t][r]; thread
  //  (lane slot el, unit eu) owns the rows o = 4 eu .. 4 eu + 3 of that group)
  const int el = tid & 63, eu = (tid >> 6) & 3;
  const int mb = m0 + 4 * MUr * (el >> 4) + 4 * eu, en = n0 + (el & 15);
  const bool eown = tid < 64 * MUr && mb < M && en < N;
  const bool bown = tm == 0 && T.b_off >= 0 && tid >= 256 && tid < 272 && n0 + (tid - 256) < N;
  float pp[4] = {0.f, 0.f, 0.f, 0.f}, pm[4] = {0.f, 0.f, 0.f, 0.f}, pv[4] = {0.f, 0.f, 0.f, 0.f};
  float bp = 0.f, bm = 0.f, bv = 0.f, lr_t = 0.f;
  bool poisoned = false;
  // what the epilogue needs besides the tile: requested / computed AFTER the contraction (vmcnt retires in order: issued
  // before it, these loads and the ~100 instructions of alpha_t would sit in front of the first MFMA: +1.5 us measured)
  const bool upd = fa.do_adam != 0;                // data parallel: gradients only (all-reduce and adam_tf_img follow)
  auto prologue = [&]() {
    if (!upd) return;
    if (eown) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int mm = min(mb + j, M - 1);
        const long long i = (long long)T.w_off + (long long)mm * N + en;
        pp[j] = fa.p[i]; pm[j] = fa.m[i]; pv[j] = fa.v[i];
      }
    }
    if (bown) { const int i = T.b_off + n0 + tid - 256; bp = fa.p[i]; bm = fa.m[i]; bv = fa.v[i]; }
    poisoned = fa.err_word && *fa.err_word;
    if (a.lr_t) {
      lr_t = *a.lr_t;                              // alpha_t = lr sqrt(1 - b2^t) / (1 - b1^t), computed once by mega2_fwd_bwd
    } else {                                       // 1 - b^t = -expm1(t ln b): no cancellation, ~3e-7 relative to the fp64 form
      const float tf = (float)((fa.step_dev ? fa.step_dev[1] : 0ull) + 1ull);
      lr_t = fa.lr * sqrtf(-expm1f(tf * a.ln_b2)) / (-expm1f(tf * a.ln_b1));
    }
  };
  const float omb1 = 1.f - fa.b1, omb2 = 1.f - fa.b2, gs = 1.f / fa.count;
  DW_ST(1);
  // ---- the contraction: this wave's share of the batch rows, 4 rows (k) per MFMA step
  const int rows_w = (((B + kDwWaves - 1) / kDwWaves) + 3) & ~3;
  const int b_lo = wave * rows_w, b_hi = min(B, b_lo + rows_w);
  prologue();                                    // (variant B: requested BEFORE the contraction's operands)
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float cs = 0.f;
  if (T.a_u8 && b_hi - b_lo == 128 && a.u8x3)
    dw_contract_u8x3(static_cast<const unsigned char*>(T.A), T.dY, lda, ldy, M, N, m0, n0, b_lo, ln, lk, acc, cs);
  else if (rows_w <= 32) {                       // small batch: 8 k-steps per batch of loads
    if (T.a_u8) dw_contract<true, 4, 8>(T.A, T.dY, lda, ldy, M, N, m0, n0, b_lo, b_hi, ln, lk, acc, cs, 0);
    else if (MUr == 2) dw_contract<false, 2, 8>(T.A, T.dY, lda, ldy, M, N, m0, n0, b_lo, b_hi, ln, lk, acc, cs, 0);
    else dw_contract<false, 1, 8>(T.A, T.dY, lda, ldy, M, N, m0, n0, b_lo, b_hi, ln, lk, acc, cs, 0);
  }
  else if (T.a_u8) dw_contract<true, 4>(T.A, T.dY, lda, ldy, M, N, m0, n0, b_lo, b_hi, ln, lk, acc, cs, 0);
  else if (MUr == 2) dw_contract<false, 2>(T.A, T.dY, lda, ldy, M, N, m0, n0, b_lo, b_hi, ln, lk, acc, cs, 0);
  else dw_contract<false, 1>(T.A, T.dY, lda, ldy, M, N, m0, n0, b_lo, b_hi, ln, lk, acc, cs, 0);
  DW_ST(2);
  // ---- the waves' partial tiles meet in LDS (fixed order: bit-reproducible)
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(wave * 16 + 4 * t + r) * 64 + lane] = acc[t][r];     // lane-contiguous: conflict-free
  redcs[wave * 64 + lane] = cs;
  __syncthreads();
  DW_ST(3);
  if (eown) {
    float g[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < kDwWaves; ++w)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int o = 4 * eu + j, r = MUr == 4 ? eu : (MUr == 2 ? o >> 1 : o), t = MUr == 4 ? j : (MUr == 2 ? o & 1 : 0);
        g[j] += red[(w * 16 + 4 * t + r) * 64 + el];
      }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (mb + j < M) {
        const long long i = (long long)T.w_off + (long long)(mb + j) * N + en;
        fa.grads[i] = g[j];
        if (upd && !poisoned) {
          adam_update(pp[j], pm[j], pv[j], g[j], gs, lr_t, omb1, omb2, fa.eps);
          fa.p[i] = pp[j]; fa.m[i] = pm[j]; fa.v[i] = pv[j];
        }
      }
    }
    if (upd && !poisoned) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (mb + j >= M) pp[j] = 0.f;              // rows past the tensor: the image's padding stays zero
      if (T.k1 == 2 || T.k1 == 4 || T.k1 == 7)
        *reinterpret_cast<float4*>(fa.img[T.which1] + img_dst(T.k1, T.base1, T.ld1, T.chunk1, mb, en)) = make_float4(pp[0], pp[1], pp[2], pp[3]);
      else if (T.k1 >= 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (mb + j < M) fa.img[T.which1][img_dst(T.k1, T.base1, T.ld1, T.chunk1, mb + j, en)] = pp[j];
      }
      if (T.k2 >= 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (mb + j < M) fa.img[T.which2][img_dst(T.k2, T.base2, T.ld2, T.chunk2, mb + j, en)] = pp[j];
      }
    }
  }
  if (bown) {                                    // bias gradient = column sum of dY over the batch
    const int c = tid - 256;
    float g = 0.f;
#pragma unroll
    for (int w = 0; w < kDwWaves; ++w)
#pragma unroll
      for (int k = 0; k < 4; ++k) g += redcs[w * 64 + k * 16 + c];
    const int i = T.b_off + n0 + c;
    fa.grads[i] = g;
    if (upd && !poisoned) {
      adam_update(bp, bm, bv, g, gs, lr_t, omb1, omb2, fa.eps);
      fa.p[i] = bp; fa.m[i] = bm; fa.v[i] = bv;
      if (T.bk >= 0) fa.img[T.bwhich][img_dst(T.bk, T.bbase, 0, T.bchunk, 0, n0 + c)] = bp;
    }
  }
  DW_ST(4);
  if (a.dbg && threadIdx.x == 0) a.dbg[(size_t)blockIdx.x * 8 + 5] = (unsigned long long)ti;
  DW_END();
#undef DW_END
#undef DW_ST
}

}  // namespace gmvae
