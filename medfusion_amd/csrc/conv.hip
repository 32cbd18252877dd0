// conv.hip -- 2-D convolution for the UNet / VAE of the sampling path, gfx950.  Replaces torch.nn.Conv2d.forward at
// conv_blocks.py:185,238,66,123-125, unet2.py:259,267, latent_embedders.py:768 (include/medfusion_hip.h has the call-site map).
//
//  * conv_igemm_kernel<BM,BN,WM,WN,BK,MODE,FG>: implicit GEMM on the matrix cores.  M = N*Hout*Wout output pixels, N = Cout,
//    K = KH*KW*Cin walked channel-chunk OUTER / filter-tap INNER (L2 locality).  NHWC activations make every A row of a K-chunk
//    (32 channels of one tap) 128 contiguous bytes; weights are pre-packed [Cout][KH][KW][Cin].  The gather fuses zero padding,
//    stride 2 (BasicDown), nearest x2 upsampling (BasicUp: gather form or the 4-phase sub-pixel form) and the skip concat (two
//    source pointers) -- none of those tensors is ever materialised.  Deterministic split-K (slabs + reducer, which also emits the
//    GroupNorm partial statistics); statistics from the epilogue when K is not split.
//      MODE 0  MF_CONV_FP32: v_mfma_f32_32x32x2_f32.  LDS tiles [rows][32+4] floats, conflict-free ds_read_b128 (k permuted
//              identically for A and B), register-prefetched double buffer, one barrier per chunk at the top.
//      MODE 1  MF_CONV_FP32_SPLIT3: every fp32 operand split exactly into 3 bf16 terms, 6 product terms on
//              v_mfma_f32_32x32x16_bf16, fp32 accumulate.  LDS rows [3 pieces][32 bf16] + pad.  The chunk's other work is cut into
//              units pinned between the MFMAs; ONE barrier per chunk in the middle of the MFMA stream, next chunk's first-step
//              fragments prefetched behind it.
//      MODE 2  ..._CHUNKSUM: MODE 1 with per-chunk MFMA accumulators added by the VALU.
//      MODE 3  ..._W3: MODE 1 with the weights already split at load time (the default of the product).
//      MODE 4  MODE 3 with ONE LDS buffer, fragments held in registers, two 4-wave workgroups per CU (narrow VAE levels).
//      MODE 5  MF_CONV_BF16: opt-in reduced precision (one bf16 term).
//    FG: fast gather addressing (no fused nearest-x2 gather, < 2^24 source pixels).
//  * conv_smallcin_kernel / conv_direct_kernel: the edge convolutions (Cin = 8|3, Cout = 8|3|16, NCHW edges): <0.2 % of the FLOPs.
//  * DESIGN.md section 3 has the measurements and what was tried and rejected.
#include "common.h"
#include "gn_partial.h"
#include <cstdlib>

using namespace mf;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Timing experiments only (scripts/ablate_split.sh builds side libraries with -DMF_ABLATE=bits; results are then WRONG by design):
// 1 no split arithmetic, 2 no LDS stores, 4 no global loads, 8 no barrier, 16 no second-step fragment reads, 32 no MFMAs,
// 64 LDS stores of values that do not depend on the global loads (isolates the wait for the loads).
#ifndef MF_ABLATE
#define MF_ABLATE 0
#endif
constexpr int kAblate = MF_ABLATE;

namespace {

struct ConvP {
  const float* x1;
  const float* x2;
  const float* w;
  const float* bias;
  float* y;
  int N, Hin, Win, C1, C2, Cin, Cout;
  int Hout, Wout, Heff, Weff;
  int KH, KW, stride, pad, ups;
  int M, K, HWout;
  int cchunks, nk, nk_per_split, splitk;
  int tiles_m, tiles_n;
  long slab;
  int in_nchw, out_nchw;
  unsigned bytes1, bytes2, bytesw;  // buffer-descriptor extents (igemm path: all < 4 GiB, checked on the host)
  int subpix, hw_src;               // sub-pixel form of nearest-x2 + 3x3: 4 phase-specific 2x2 convs on the low-res source
  double* gn_partial;               // optional fused GroupNorm statistics: [N][gn_parts][G][2] = {sum, sumsq}
  int gn_groups, gn_parts, gn_cpg;
  GnFinal gn_fin;                   // optional last-arriver finalize -> stats[n][g] = {mean, rstd}
  int fastg;                        // fast gather usable: no fused nearest-x2 gather, < 2^24 source pixels, < 2^22 channels per source
};

// bijective XCD-aware remap: block b runs on XCD b%8; give each XCD a contiguous range of logical ids
__device__ __forceinline__ int xcd_remap(int bid, int total) {
  const int q = total >> 3, r = total & 7;
  const int xcd = bid & 7, within = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + within;
}

// MODE 1 ("fp32 through 3 x bf16"): every fp32 operand is split EXACTLY into three bf16 terms x = h + m + l (truncation: h = top
// 8 significant bits, m = the next 8, l = the last 8), and a*b is accumulated in fp32 on the bf16 matrix cores as the six terms
// of order <= 2: ah*bh + ah*bm + am*bh + am*bm + ah*bl + al*bh.  The dropped terms (am*bl, al*bm, al*bl) are < 2^-23 |a*b|, i.e.
// below fp32 rounding of the product; v_mfma_f32_32x32x16_bf16 runs at 16x the rate of the fp32 MFMA, so 6 of them cost 3/8.
__device__ __forceinline__ void split2_bf16x3(const float x0, const float x1, unsigned& h, unsigned& m, unsigned& l) {
  const unsigned b0 = __float_as_uint(x0), b1 = __float_as_uint(x1);
  const float r0 = x0 - __uint_as_float(b0 & 0xffff0000u), r1 = x1 - __uint_as_float(b1 & 0xffff0000u);  // exact
  const unsigned c0 = __float_as_uint(r0), c1 = __float_as_uint(r1);
  const float s0 = r0 - __uint_as_float(c0 & 0xffff0000u), s1 = r1 - __uint_as_float(c1 & 0xffff0000u);  // exact, <= 8 significant bits left
  // pack the HIGH halves of two words into one: v_perm_b32 (no masking/shift needed)
  h = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
  m = __builtin_amdgcn_perm(c1, c0, 0x07060302u);
  l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned round_bf16x2(const float x0, const float x1) {  // v_cvt_pk_bf16_f32 (round to nearest even)
  const bf16x2 v = {(__bf16)x0, (__bf16)x1};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ u32x2 round_bf16x4(const f32x4 v) {
  const float x0 = v[0], x1 = v[1], x2 = v[2], x3 = v[3];
  return u32x2{round_bf16x2(x0, x1), round_bf16x2(x2, x3)};
}
__device__ __forceinline__ void split_bf16x3(const f32x4 v, u32x2& h, u32x2& m, u32x2& l) {
  // (scalar copies first: __builtin_bit_cast applied directly to a vector ELEMENT read element 0 every time)
  const float x0 = v[0], x1 = v[1], x2 = v[2], x3 = v[3];
  unsigned h0, m0, l0, h1, m1, l1;
  split2_bf16x3(x0, x1, h0, m0, l0);
  split2_bf16x3(x2, x3, h1, m1, l1);
  h = u32x2{h0, h1}; m = u32x2{m0, m1}; l = u32x2{l0, l1};
}

template <int BM, int BN, int WM, int WN, int BK, int MODE, bool FG>
__global__ __launch_bounds__(WM* WN * 64, (MODE == 4 ? 2 : 1)) void conv_igemm_kernel(const ConvP p) {
  constexpr int NT = WM * WN * 64;
  // row pitch in 32-bit words.  MODE 0: BK floats + 4 (conflict-free ds_read_b128 for BK = 32 and 64).
  // MODE 1: [3 pieces][32 bf16] = 48 words + 4: pitch/4 = 13 is odd, so the 16 rows of a quarter-wave b128 read hit distinct banks.
  constexpr int LDK = MODE == 0 ? BK + 4 : (MODE == 5 ? 20 : 52);  // MODE 5: 32 bf16 = 16 words + 4 (pitch/4 = 5, odd)
  static_assert(BK == 32 || BK == 64, "BK");
  static_assert(MODE == 0 || BK == 32, "split mode: BK = 32");
  // MODE 4 = MODE 3 with ONE LDS buffer and 4-wave workgroups, two of them per CU: the fragments of a chunk are pulled into
  // registers (barrier | 24 ds_read_b128 | barrier), then the same buffer is refilled with the next chunk while the MFMAs run from
  // registers.  The two waves of a SIMD then belong to DIFFERENT workgroups with their own barrier cadence, so one computes while the
  // other sits in its barrier/fragment-read window (with two waves of ONE workgroup per SIMD both sit there at the same time: the
  // matrix pipe was 59 % busy, profiles/r01_pmc_conv_split.csv).
#ifndef MF_MIDBARRIER
#define MF_MIDBARRIER 1
#endif
  // MB: the double-buffered split modes run with ONE barrier per chunk in the MIDDLE of the MFMA stream: before it a wave stores its
  // share of chunk k+1 and reads its last fragments of chunk k, after it the first-step fragments of chunk k+1 are prefetched into
  // the registers the first half of the MFMAs has finished with -- the next chunk's MFMAs start without a barrier and without an
  // exposed LDS round trip (the barrier window cost ~15 points of matrix-pipe utilisation: profiles/r01_mimic_probe.txt).
  constexpr bool MB = MF_MIDBARRIER && MODE >= 1 && MODE != 4 && (MODE == 5 ? 1 : 6) * (BM / (WM * 32)) * (BN / (WN * 32)) >= 4;  // >= 8 MFMAs per chunk
  constexpr bool SB = MODE == 4;
  constexpr int NBUF = SB ? 1 : 2;
  // MODE 5 = MF_CONV_BF16 (opt-in, REDUCED precision): operands rounded to bf16 (RNE), one MFMA term, fp32 accumulate; weights
  // arrive already converted (mf_convert_conv_weight_bf16).  Same kernel with NP = 1 piece instead of 3.
  constexpr int NP = MODE == 5 ? 1 : 3, NTERM = MODE == 5 ? 1 : 6;
  constexpr bool WS = MODE == 3 || MODE == 4 || MODE == 5;     // MF_CONV_FP32_SPLIT3_W3: the weights arrive as bf16 triplets [row][K/8][3 pieces][8] (no split, no VALU for B)
  constexpr bool FLUSH = MODE == 2;  // MF_CONV_FP32_SPLIT3_CHUNKSUM: per-chunk MFMA accumulators, added into the running fp32 sum by the VALU (RNE)
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int TPR = BK / 4;   // staging: TPR threads (float4 each) cover one BK-float row
  constexpr int RPP = NT / TPR;
  constexpr int PA = BM / RPP, PB = BN / RPP;
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile/threads mismatch");
  constexpr int RPW = NT / 4;                 // pre-split weights: 4 threads (8 k each: 3 x 16 bytes) cover one 32-k row
  constexpr int PW = WS ? BN / RPW : 1;
  static_assert(!WS || BN % RPW == 0, "tile/threads mismatch (pre-split weights)");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                     // [NBUF][BM][LDK]
  float* Bs = smem + NBUF * BM * LDK;   // [NBUF][BN][LDK]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  const int total = p.tiles_m * p.tiles_n * p.splitk;
  const int logical = xcd_remap(blockIdx.x, total);
  const int tile_m = logical % p.tiles_m;
  const int rest = logical / p.tiles_m;
  const int tile_n = rest % p.tiles_n;
  const int kz = rest / p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kc_beg = kz * p.nk_per_split;
  const int kc_end = min(p.nk, kc_beg + p.nk_per_split);

  const int srow = tid / TPR, skoff = (tid % TPR) * 4;

  // FG ("fast gather", host: no fused nearest-x2 gather, < 2^24 source pixels): per row the pixel index of tap (0,0) and a bit mask
  // of the taps that fall outside the image; per chunk the gather address is then add + 24-bit mad + bfe + or instead of the
  // generic coordinate arithmetic (2 compares, 2 full 32-bit multiplies, selects)
  int a_n[PA], a_iy0[PA], a_ix0[PA], a_pix[PA], a_inv[PA];
  const unsigned skoff4 = (unsigned)skoff * 4u;
#pragma unroll
  for (int q = 0; q < PA; ++q) {
    const int m = m0 + q * RPP + srow;
    if (m < p.M) {
      const int n = m / p.HWout;
      const int rem = m - n * p.HWout;
      a_n[q] = n * p.Hin;
      if (p.subpix) {  // m = (n, phase, y, x) over the SOURCE grid; output pixel (2y + a, 2x + b)
        const int ph = rem / p.hw_src, r2 = rem - ph * p.hw_src;
        const int y = r2 / p.Win, x = r2 - y * p.Win;
        a_iy0[q] = y + (ph >> 1) - 1;
        a_ix0[q] = x + (ph & 1) - 1;
      } else {
        const int oy = rem / p.Wout;
        const int ox = rem - oy * p.Wout;
        a_iy0[q] = oy * p.stride - p.pad;
        a_ix0[q] = ox * p.stride - p.pad;
      }
    } else {
      a_n[q] = 0;
      a_iy0[q] = -(1 << 28);  // rows past M: always "out of bounds" -> zeros (address clamps to pixel 0 of image 0)
      a_ix0[q] = 0;
    }
    if constexpr (FG) {
      unsigned valid = 0;
#pragma unroll
      for (int ty = 0; ty < 3; ++ty)
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {  // tap index ty * KW + tx (KH, KW <= 3)
          const bool in = ty < p.KH && tx < p.KW && (unsigned)(a_iy0[q] + ty) < (unsigned)p.Heff && (unsigned)(a_ix0[q] + tx) < (unsigned)p.Weff;
          valid |= (in ? 1u : 0u) << (ty * p.KW + tx);
        }
      a_inv[q] = (int)~valid;  // bits >= KH*KW stay set: bit 31 is the "chunk past the end" tap
      a_pix[q] = (a_n[q] + a_iy0[q]) * p.Win + a_ix0[q];
    } else {
      a_inv[q] = a_pix[q] = 0;
    }
  }
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.bytesw, 0x00020000);
  // sub-pixel form: the tile lies inside one phase (host guarantees hw_src % BM == 0); each phase has its own [Cout][2][2][Cin] weights
  const int phase_t = p.subpix ? ((m0 % p.HWout) / p.hw_src) : 0;
  const unsigned wboff = (unsigned)((phase_t * p.Cout + n0 + srow) * p.K + skoff) * 4u;
  const int wrow = tid >> 2, wo = tid & 3;
  const unsigned wsoff = (unsigned)((phase_t * p.Cout + n0 + wrow) * p.K) * (2u * NP) + (unsigned)wo * (16u * NP);  // 2 NP bytes per weight

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // K-chunk order: channel chunk OUTER, filter tap INNER.  The KH*KW taps of one 32-channel chunk read the same 128-byte pixel
  // segments (shifted by one pixel), so consecutive chunks hit L1/L2; with the taps outside, a whole pass over the channels
  // (the A footprint of the 32 workgroups of an XCD, > 4 MB L2) lay between two uses of a line and every tap pass missed
  // (rocprofv3: 50 % L2 hit rate, 7.7x the compulsory bytes fetched).  The weights stay [Cout][tap][Cin]: a chunk is still 128
  // contiguous bytes per row.
#ifndef MF_KORDER
#define MF_KORDER 1
#endif
  constexpr bool kTapInner = MF_KORDER == 1;
  const int taps_ = p.KH * p.KW;
  int cc = kTapInner ? kc_beg / taps_ : kc_beg % p.cchunks;
  int tap = kTapInner ? kc_beg - cc * taps_ : kc_beg / p.cchunks;
  int ky = tap / p.KW, kx = tap - ky * p.KW;

  // Two register sets: set 0 holds chunk 0 during the cold start only, set 1 is the steady-state prefetch register set
  // (a gather running TWO chunks ahead through both sets was built and measured: -1...+1 %, not kept).
  f32x4 ra0[PA], rb0[PB], ra1[PA], rb1[PB];
  u32x4 rw0[PW][NP], rw1[PW][NP];
  bf16x8 fra[2][TM][NP], frb[2][TN][NP];  // split modes: MFMA operand fragments (persist across iterations with the mid barrier)

// (macros, not lambdas: by-reference captures of the index arrays were demoted to scratch memory)
// Gather through buffer loads: a descriptor per source tensor, 32-bit byte offsets, and the hardware range check
// supplies the zero padding (an out-of-range offset returns 0) -- no branch, no select on the data, one basic block.
// Chunks past the end of this workgroup's K range are "loaded" the same way (all offsets out of range).
#define MF_GLOAD_SETUP(KC)                                                                                   \
    const bool lv_ = (KC) < kc_end;                                                                          \
    const int c0_ = cc * BK;                                                                                 \
    const bool first_ = c0_ < p.C1;                                                                          \
    const int Cs_ = first_ ? p.C1 : p.C2;                                                                    \
    const int coff_ = (first_ ? c0_ : c0_ - p.C1) + skoff;                                                   \
    const int tapoff_ = ky * p.Win + kx, tsel_ = lv_ ? ky * p.KW + kx : 31;                                  \
    const unsigned Cs4_ = (unsigned)Cs_ * 4u, cb4_ = (unsigned)(first_ ? c0_ : c0_ - p.C1) * 4u;             \
    const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(                                    \
        const_cast<float*>(first_ ? p.x1 : p.x2), 0, first_ ? p.bytes1 : p.bytes2, 0x00020000);
#define MF_GLOAD_A(SET, Q)                                                                                   \
  {                                                                                                          \
    if constexpr (FG) {                                                                                      \
      const unsigned off = (__umul24((unsigned)(a_pix[Q] + tapoff_), Cs4_) + (cb4_ + skoff4)) |              \
                           (unsigned)__builtin_amdgcn_sbfe(a_inv[Q], (unsigned)tsel_, 1u);                   \
      ra##SET[Q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_, off, 0, 0));         \
    } else {                                                                                                 \
      const int iy = a_iy0[Q] + ky, ix = a_ix0[Q] + kx;                                                      \
      const bool ok = lv_ && (unsigned)iy < (unsigned)p.Heff && (unsigned)ix < (unsigned)p.Weff;             \
      const int sy = iy >> p.ups, sx = ix >> p.ups;                                                          \
      const unsigned off = (unsigned)(((a_n[Q] + sy) * p.Win + sx) * Cs_ + coff_) * 4u;                      \
      ra##SET[Q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_, ok ? off : 0xFFFFFFF0u, 0, 0)); \
    }                                                                                                        \
  }
#define MF_GLOAD_W(SET, Q)                                                                                   \
  {  /* one address per row (select + add), the 16-byte pieces through the instruction's immediate offset */  \
    const unsigned wv_ = lv_ ? wsoff + (unsigned)((Q) * RPW * p.K + (ky * p.KW + kx) * p.Cin + cc * BK) * (2u * NP) : 0xFFFFFF00u; \
    _Pragma("unroll") for (int c = 0; c < NP; ++c)                                                           \
      rw##SET[Q][c] = __builtin_amdgcn_raw_buffer_load_b128(rsw, wv_ + c * 16u, 0, 0);                       \
  }
#define MF_GLOAD_B1(SET, Q)                                                                                  \
  rb##SET[Q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(                              \
      rsw, lv_ ? wboff + (unsigned)((Q) * RPP * p.K + (ky * p.KW + kx) * p.Cin + cc * BK) * 4u : 0xFFFFFFF0u, 0, 0));
#define MF_GLOAD_B(SET, KC)                                                                                  \
  if constexpr (WS) {                                                                                        \
    _Pragma("unroll") for (int q = 0; q < PW; ++q) MF_GLOAD_W(SET, q)                                        \
  } else {                                                                                                   \
    _Pragma("unroll") for (int q = 0; q < PB; ++q) MF_GLOAD_B1(SET, q)                                       \
  }
#define MF_GLOAD(SET, KC)                                                                                    \
  {                                                                                                          \
    MF_GLOAD_SETUP(KC)                                                                                       \
    _Pragma("unroll") for (int q = 0; q < PA; ++q) MF_GLOAD_A(SET, q)                                        \
    MF_GLOAD_B(SET, KC)                                                                                      \
  }
#define MF_ADVANCE()                            \
  {                                             \
    if (kTapInner) {                            \
      ++kx;                                     \
      const int w1_ = (kx == p.KW) ? 1 : 0;     \
      kx = w1_ ? 0 : kx;                        \
      ky += w1_;                                \
      const int w2_ = (ky == p.KH) ? 1 : 0;     \
      ky = w2_ ? 0 : ky;                        \
      cc += w2_;                                \
    } else {                                    \
      ++cc;                                     \
      const int w1_ = (cc == p.cchunks) ? 1 : 0; \
      cc = w1_ ? 0 : cc;                        \
      kx += w1_;                                \
      const int w2_ = (kx == p.KW) ? 1 : 0;     \
      kx = w2_ ? 0 : kx;                        \
      ky += w2_;                                \
    }                                           \
  }
#define MF_LDS_STORE(BUF, SET)                                                                               \
  {                                                                                                          \
    if constexpr (MODE == 0) {                                                                               \
      float* a_ = As + (BUF) * BM * LDK + srow * LDK + skoff;                                                \
      float* b_ = Bs + (BUF) * BN * LDK + srow * LDK + skoff;                                                \
      _Pragma("unroll") for (int q = 0; q < PA; ++q) *reinterpret_cast<f32x4*>(a_ + q * RPP * LDK) = ra##SET[q]; \
      _Pragma("unroll") for (int q = 0; q < PB; ++q) *reinterpret_cast<f32x4*>(b_ + q * RPP * LDK) = rb##SET[q]; \
    } else { /* 4 consecutive k of one row -> 4 bf16 (8 bytes) in each of the three piece planes of that row */ \
      float* a_ = As + (BUF) * BM * LDK + srow * LDK + (skoff >> 1);                                         \
      float* b_ = Bs + (BUF) * BN * LDK + srow * LDK + (skoff >> 1);                                         \
      _Pragma("unroll") for (int q = 0; q < PA; ++q) {                                                       \
        if constexpr (NP == 1) {                                                                             \
          *reinterpret_cast<u32x2*>(a_ + q * RPP * LDK) = round_bf16x4(ra##SET[q]);                          \
        } else {                                                                                             \
          u32x2 h_, m_, l_;                                                                                  \
          split_bf16x3(ra##SET[q], h_, m_, l_);                                                              \
          *reinterpret_cast<u32x2*>(a_ + q * RPP * LDK) = h_;                                                \
          *reinterpret_cast<u32x2*>(a_ + q * RPP * LDK + 16) = m_;                                           \
          *reinterpret_cast<u32x2*>(a_ + q * RPP * LDK + 32) = l_;                                           \
        }                                                                                                    \
      }                                                                                                      \
      if constexpr (WS) {                                                                                    \
        float* w_ = Bs + (BUF) * BN * LDK + wrow * LDK + wo * 4;                                             \
        _Pragma("unroll") for (int q = 0; q < PW; ++q)                                                       \
          _Pragma("unroll") for (int c = 0; c < NP; ++c)                                                     \
            *reinterpret_cast<u32x4*>(w_ + q * RPW * LDK + c * 16) = rw##SET[q][c];                          \
      } else {                                                                                               \
        _Pragma("unroll") for (int q = 0; q < PB; ++q) {                                                     \
          u32x2 h_, m_, l_;                                                                                  \
          split_bf16x3(rb##SET[q], h_, m_, l_);                                                              \
          *reinterpret_cast<u32x2*>(b_ + q * RPP * LDK) = h_;                                                \
          *reinterpret_cast<u32x2*>(b_ + q * RPP * LDK + 16) = m_;                                           \
          *reinterpret_cast<u32x2*>(b_ + q * RPP * LDK + 32) = l_;                                           \
        }                                                                                                    \
      }                                                                                                      \
    }                                                                                                        \
  }

  const int frag_off = (lane & 31) * LDK + 4 * (lane >> 5);
  const float* Aw = As + (wm * TM * 32) * LDK + frag_off;
  const float* Bw = Bs + (wn * TN * 32) * LDK + frag_off;

  // Pipeline (one barrier per K-chunk, at the TOP of the iteration):
  //   iteration k:  barrier | fragment reads | LDS-store chunk k+1 (register set (k+1)&1, loaded during iteration k-2) -> buf^1 |
  //                 issue the global loads of chunk k+3 into the same set | MFMAs of chunk k from buf.
  // Every iteration stores and loads (chunks past the end are all-out-of-range loads and a store nobody reads): one body, no tail.
  // Variants that were built, verified and measured SLOWER on MI355X (git history): a ping-pong schedule between
  // the two waves of each SIMD (fp32 and split mode), BK = 64 for the 8-wave tile.
  // Hazards: buf^1 was last read in iteration k-1 (all waves are past this iteration's barrier); chunk k in buf was
  // stored in iteration k-1 and is visible after the barrier (each wave drains lgkmcnt before arriving).
#define MF_COMPUTE(SET, KC, DO_STORE, DO_LOAD)                                                                                        \
  {                                                                                                                  \
    if (!MB && !(kAblate & 8)) __syncthreads();                                                                      \
    const float* Ab = Aw + buf * BM * LDK;                                                                           \
    const float* Bb = Bw + buf * BN * LDK;                                                                           \
    if constexpr (MODE == 0) {                                                                                       \
      f32x4 fa[2][TM], fb[2][TN];                                                                                    \
      _Pragma("unroll") for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDK);  \
      _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[0][j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LDK);  \
      if (DO_STORE) MF_LDS_STORE(buf ^ 1, SET);                                                                      \
      if (DO_LOAD) { MF_ADVANCE(); MF_GLOAD(SET, KC); }                                                              \
      _Pragma("unroll") for (int kk = 0; kk < BK / 8; ++kk) {                                                        \
        const int cur = kk & 1, nxt = cur ^ 1;                                                                       \
        if (kk + 1 < BK / 8) {                                                                                       \
          _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                             \
              fa[nxt][i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDK + (kk + 1) * 8);                        \
          _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                             \
              fb[nxt][j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LDK + (kk + 1) * 8);                        \
        }                                                                                                            \
        _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                                \
          _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                             \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                           \
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i][s], fb[cur][j][s], acc[i][j], 0, 0, 0);    \
      }                                                                                                              \
    } else { /* two 16-deep MFMA steps per chunk; lane half hf reads the 8 consecutive k [16 s + 8 hf, +8) of each piece.   \
                The chunk's other work (3-way split + LDS store of chunk k+1, fragment reads of the second step, gather of      \
                chunk k+3) is cut into small units pinned BETWEEN the MFMAs (sched_barrier fences). */                          \
      if constexpr (!MB) {                                                                                           \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                               \
          _Pragma("unroll") for (int c = 0; c < NP; ++c)                                                             \
            fra[0][i][c] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDK + c * 16));  \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                               \
          _Pragma("unroll") for (int c = 0; c < NP; ++c)                                                             \
            frb[0][j][c] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(Bb + j * 32 * LDK + c * 16));  \
      }                                                                                                              \
      const float* An_ = Aw + (buf ^ 1) * BM * LDK;                                                                  \
      const float* Bn_ = Bw + (buf ^ 1) * BN * LDK;                                                                  \
      float* sa_ = As + (SB ? 0 : buf ^ 1) * BM * LDK + srow * LDK + (skoff >> 1);                                   \
      float* sb_ = Bs + (SB ? 0 : buf ^ 1) * BN * LDK + srow * LDK + (skoff >> 1);                                   \
      unsigned h0_ = 0, m0_ = 0, l0_ = 0, h1_ = 0, m1_ = 0, l1_ = 0;                                                 \
      int coff_ = 0, Cs_ = 0, tapoff_ = 0, tsel_ = 31;                                                               \
      unsigned Cs4_ = 0, cb4_ = 0;                                                                                   \
      bool lv_ = false;                                                                                              \
      __amdgpu_buffer_rsrc_t rs_ = rsw;                                                                              \
      float* sw_ = Bs + (SB ? 0 : buf ^ 1) * BN * LDK + wrow * LDK + wo * 4;                                         \
      constexpr int NM = 2 * NTERM * TM * TN, RU = SB ? 0 : TM + TN, UA = (SB || MB) ? 3 * PA : 3 * (PA - 1);        \
      constexpr int UB = RU + UA + (WS ? PW : 3 * PB);  /* units before the mid barrier (all of them without one) */ \
      constexpr int UI = MB ? UB + 1 + RU : UB;                                                                      \
      constexpr int HS = NM / 2;                                                                                     \
      static_assert(UB <= 8 * (MB ? HS : NM), "units per MFMA slot");                                                \
      { /* K-chunk advance + descriptor of the chunk to gather (scalar work) */                                      \
        MF_ADVANCE();                                                                                                \
        const int c0_ = cc * BK;                                                                                     \
        const bool first_ = c0_ < p.C1;                                                                              \
        lv_ = (KC) < kc_end;                                                                                         \
        Cs_ = first_ ? p.C1 : p.C2;                                                                                  \
        coff_ = (first_ ? c0_ : c0_ - p.C1) + skoff;                                                                 \
        tapoff_ = ky * p.Win + kx; tsel_ = lv_ ? ky * p.KW + kx : 31;                                                \
        Cs4_ = (unsigned)Cs_ * 4u; cb4_ = (unsigned)(first_ ? c0_ : c0_ - p.C1) * 4u;                                \
        rs_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(first_ ? p.x1 : p.x2), 0, first_ ? p.bytes1 : p.bytes2, 0x00020000); \
      }                                                                                                              \
      if constexpr (SB) { /* all fragments of the chunk into registers, then the buffer is free for chunk k+1 */      \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                               \
          _Pragma("unroll") for (int c = 0; c < NP; ++c)                                                              \
            fra[1][i][c] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDK + c * 16 + 8)); \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                               \
          _Pragma("unroll") for (int c = 0; c < NP; ++c)                                                              \
            frb[1][j][c] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(Bb + j * 32 * LDK + c * 16 + 8)); \
        __syncthreads();                                                                                             \
      } else if constexpr (!MB) {                                                                                    \
        MF_ITEM_A(0, 0, SET) MF_ITEM_A(0, 1, SET) MF_ITEM_A(0, 2, SET)  /* covers the latency of the fragment reads */ \
      }                                                                                                              \
      __builtin_amdgcn_sched_barrier(0);                                                                             \
      f32x16 accc[TM][TN];                                                                                           \
      _Pragma("unroll") for (int n = 0; n < NM; ++n) {                                                               \
        constexpr int kCA[6] = {2, 0, 1, 1, 0, 0}, kCB[6] = {0, 2, 1, 0, 1, 0};  /* smallest terms first */         \
        const int j_ = n % TN, i_ = (n / TN) % TM, t_ = NTERM == 1 ? 5 : (n / (TN * TM)) % 6, s_ = n / (TN * TM * NTERM);                 \
        if (kAblate & 32) {                                                                                          \
        } else if (FLUSH) {                                                                                          \
          if (s_ == 0 && t_ == 0) {                                                                                  \
            f32x16 z_;                                                                                               \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) z_[r] = 0.f;                                              \
            accc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fra[s_][i_][kCA[t_]], frb[s_][j_][kCB[t_]], z_, 0, 0, 0); \
          } else {                                                                                                   \
            accc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fra[s_][i_][kCA[t_]], frb[s_][j_][kCB[t_]], accc[i_][j_], 0, 0, 0); \
          }                                                                                                          \
        } else {                                                                                                     \
          acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fra[s_][i_][kCA[t_]], frb[s_][j_][kCB[t_]], acc[i_][j_], 0, 0, 0); \
        }                                                                                                            \
        { /* the units of slot n.  Without the mid barrier: UI units spread evenly over the NM slots.  With it: the UB units  \
             before the barrier over the first half, the barrier after MFMA NM/2 - 1 (the last one that reads the first-step   \
             fragments), the RU prefetch units over the second half.  (No inner loop over u: it stayed rolled for the 48-MFMA   \
             tiles and sent the fragment arrays to scratch.) */                                                                \
          int ulo_, uhi_;                                                                                            \
          if constexpr (MB) {                                                                                        \
            if (n < HS) { ulo_ = (n * UB + HS - 1) / HS; uhi_ = ((n + 1) * UB + HS - 1) / HS + (n == HS - 1 ? 1 : 0); } \
            else { ulo_ = UB + 1 + ((n - HS) * RU + HS - 1) / HS; uhi_ = UB + 1 + ((n - HS + 1) * RU + HS - 1) / HS; } \
          } else {                                                                                                   \
            ulo_ = (n * UI + NM - 1) / NM; uhi_ = ((n + 1) * UI + NM - 1) / NM;                                      \
          }                                                                                                          \
          if (ulo_ < uhi_) { MF_UNIT(ulo_, SET, KC) }                                                                \
          if (ulo_ + 1 < uhi_) { MF_UNIT(ulo_ + 1, SET, KC) }                                                        \
          if constexpr (UB > (MB ? HS : NM)) { /* few MFMAs per chunk (4-wave tiles, the one-term bf16 mode): up to 9 units */ \
            if (ulo_ + 2 < uhi_) { MF_UNIT(ulo_ + 2, SET, KC) }                                                      \
            if (ulo_ + 3 < uhi_) { MF_UNIT(ulo_ + 3, SET, KC) }                                                      \
            if (ulo_ + 4 < uhi_) { MF_UNIT(ulo_ + 4, SET, KC) }                                                      \
            if (ulo_ + 5 < uhi_) { MF_UNIT(ulo_ + 5, SET, KC) }                                                      \
            if (ulo_ + 6 < uhi_) { MF_UNIT(ulo_ + 6, SET, KC) }                                                      \
            if (ulo_ + 7 < uhi_) { MF_UNIT(ulo_ + 7, SET, KC) }                                                      \
            if (ulo_ + 8 < uhi_) { MF_UNIT(ulo_ + 8, SET, KC) }                                                      \
          }                                                                                                          \
        }                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
      }                                                                                                              \
      if (FLUSH) {                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                               \
          _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                             \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[i][j][r] += accc[i][j][r];                            \
      }                                                                                                              \
    }                                                                                                                \
    if (!SB) buf ^= 1;                                                                                               \
  }
// work unit U of a split-mode chunk (see MF_COMPUTE): [0, RU) second-step fragment reads of one 32-row sub-tile; then 3 units
// per staging item (items 1..NI-1).
#define MF_UNIT(U, SET, KC)                                                                                          \
  {                                                                                                                  \
    const int u = (U);                                                                                               \
    if (u < RU) {                                                                                                    \
      if (kAblate & 16) {                                                                                            \
      } else if (u < TM) {                                                                                           \
        _Pragma("unroll") for (int c = 0; c < NP; ++c)                                                                \
          fra[1][u < TM ? u : 0][c] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(Ab + u * 32 * LDK + c * 16 + 8)); \
      } else {                                                                                                       \
        _Pragma("unroll") for (int c = 0; c < NP; ++c)                                                                \
          frb[1][u >= TM && u < RU ? u - TM : 0][c] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(Bb + (u - TM) * 32 * LDK + c * 16 + 8)); \
      }                                                                                                              \
    } else if (MB && u == UB) {  /* every wave has stored its share of chunk k+1 and read the last fragments of chunk k */ \
      if (!(kAblate & 8)) __syncthreads();                                                                           \
    } else if (MB && u > UB) {   /* first-step fragments of chunk k+1, into the registers MFMA NM/2 - 1 read last */      \
      const int v_ = u - UB - 1;                                                                                     \
      if (v_ < TM) {                                                                                                 \
        _Pragma("unroll") for (int c = 0; c < NP; ++c)                                                               \
          fra[0][v_ < TM ? v_ : 0][c] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(An_ + v_ * 32 * LDK + c * 16)); \
      } else {                                                                                                       \
        _Pragma("unroll") for (int c = 0; c < NP; ++c)                                                               \
          frb[0][v_ >= TM && v_ < RU ? v_ - TM : 0][c] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(Bn_ + (v_ - TM) * 32 * LDK + c * 16)); \
      }                                                                                                              \
    } else if (u < RU + UA) {                                                                                        \
      MF_ITEM_A(((SB || MB) ? 0 : 1) + (u - RU) / 3, (u - RU) % 3, SET)                                              \
    } else if constexpr (WS) {                                                                                       \
      MF_ITEM_W(u - RU - UA, SET)                                                                                    \
    } else {                                                                                                         \
      MF_ITEM_B((u - RU - UA) / 3, (u - RU - UA) % 3, SET)                                                           \
    }                                                                                                                \
  }
// staging items of the chunk being stored.  PART 0/1: split two floats each into bf16 triplets, PART 2: the three 8-byte LDS
// writes, then the register is free: gather the same row of the chunk this register set holds next.
#define MF_SPLIT_PARTS(V, PART, DST)                                                                                 \
    const f32x4 v_ = (kAblate & 64) ? f32x4{(float)tid, 1.f, 2.f, (float)kc} : (V);                                  \
    const float e0_ = v_[0], e1_ = v_[1], e2_ = v_[2], e3_ = v_[3];                                                  \
    if constexpr (NP == 1) {                                                                                         \
      if ((PART) == 0) h0_ = round_bf16x2(e0_, e1_);                                                                 \
      if ((PART) == 1) h1_ = round_bf16x2(e2_, e3_);                                                                 \
    } else if (kAblate & 1) {                                                                                        \
      if ((PART) == 0) { h0_ = __builtin_amdgcn_perm(__float_as_uint(e1_), __float_as_uint(e0_), 0x07060302u); m0_ = h0_; l0_ = h0_; } \
      if ((PART) == 1) { h1_ = __builtin_amdgcn_perm(__float_as_uint(e3_), __float_as_uint(e2_), 0x07060302u); m1_ = h1_; l1_ = h1_; } \
    } else {                                                                                                         \
      if ((PART) == 0) split2_bf16x3(e0_, e1_, h0_, m0_, l0_);                                                       \
      if ((PART) == 1) split2_bf16x3(e2_, e3_, h1_, m1_, l1_);                                                       \
    }                                                                                                                \
    if ((PART) == 2 && !(kAblate & 2)) {                                                                             \
      float* d_ = (DST);                                                                                             \
      *reinterpret_cast<u32x2*>(d_) = u32x2{h0_, h1_};                                                               \
      if constexpr (NP == 3) {                                                                                       \
        *reinterpret_cast<u32x2*>(d_ + 16) = u32x2{m0_, m1_};                                                        \
        *reinterpret_cast<u32x2*>(d_ + 32) = u32x2{l0_, l1_};                                                        \
      }                                                                                                              \
    }
#define MF_ITEM_A(Q, PART, SET)                                                                                      \
  {                                                                                                                  \
    const int qa_ = (Q) < PA ? (Q) : 0;                                                                              \
    MF_SPLIT_PARTS(ra##SET[qa_], PART, sa_ + qa_ * RPP * LDK)                                                        \
    if ((PART) == 2 && !(kAblate & 4)) MF_GLOAD_A(SET, qa_)                                                          \
  }
#define MF_ITEM_B(Q, PART, SET)                                                                                      \
  {                                                                                                                  \
    const int qb_ = (Q) < PB ? (Q) : 0;                                                                              \
    MF_SPLIT_PARTS(rb##SET[qb_], PART, sb_ + qb_ * RPP * LDK)                                                        \
    if ((PART) == 2 && !(kAblate & 4)) MF_GLOAD_B1(SET, qb_)                                                         \
  }
#define MF_ITEM_W(Q, SET)                                                                                            \
  {                                                                                                                  \
    const int qw_ = (Q) < PW ? (Q) : 0;                                                                              \
    _Pragma("unroll") for (int c = 0; c < NP; ++c) *reinterpret_cast<u32x4*>(sw_ + qw_ * RPW * LDK + c * 16) = rw##SET[qw_][c]; \
    MF_GLOAD_W(SET, qw_)                                                                                             \
  }

  if (kc_beg < kc_end) {
    // cold start: chunks 0 and 1 in flight before waiting for either (one memory latency, not two)
    MF_GLOAD(0, kc_beg);
    MF_ADVANCE();
    MF_GLOAD(1, kc_beg + 1);
    MF_LDS_STORE(0, 0);
    if constexpr (MB) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int c = 0; c < NP; ++c) fra[0][i][c] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(Aw + i * 32 * LDK + c * 16));
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int c = 0; c < NP; ++c) frb[0][j][c] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(Bw + j * 32 * LDK + c * 16));
    }
  }

  int buf = 0;
  int kc = kc_beg;
  if constexpr (MODE == 0) {
    for (; kc + 2 < kc_end; ++kc) {  // steady state: branch-free body
      MF_COMPUTE(1, kc + 2, true, true);
    }
    for (; kc < kc_end; ++kc) {      // last two chunks: nothing left to load, then nothing left to store
      if (kc + 1 < kc_end) {
        MF_COMPUTE(1, kc + 2, true, false);
      } else {
        MF_COMPUTE(1, kc + 2, false, false);
      }
    }
  } else {
    // split modes: ONE body -- every iteration stores and gathers; chunks past the end of this workgroup's K range are
    // all-out-of-range loads (zeros, no traffic) and a store into the buffer nobody reads any more
    for (; kc < kc_end; ++kc) {
      MF_COMPUTE(1, kc + 2, true, true);
    }
  }

  // epilogue: D[i][j], lane holds column j = lane&31 and rows (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* out = p.y + (p.splitk > 1 ? (long)kz * p.slab : 0L);
  const bool add_bias = (p.splitk == 1) && p.bias != nullptr;
  int* orow_tab = reinterpret_cast<int*>(smem) + 2048;  // past the statistics scratch (WM*BN*2 <= 2048 floats)
  if (p.subpix) {  // scatter: row m = (n, phase, y, x) -> output pixel (n, 2y + a, 2x + b)
    __syncthreads();
    if (tid < BM) {
      const int m = m0 + tid;
      const int n = m / p.HWout, rem = m - n * p.HWout;
      const int ph = rem / p.hw_src, r2 = rem - ph * p.hw_src;
      const int y = r2 / p.Win, x = r2 - y * p.Win;
      orow_tab[tid] = (n * p.Hout + 2 * y + (ph >> 1)) * p.Wout + 2 * x + (ph & 1);
    }
    __syncthreads();
  }
  const bool do_stats = p.gn_partial != nullptr;  // host guarantees splitk == 1 and HWout % BM == 0 (tile within one sample)
  float cs[TN], cq[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
    const float bv = add_bias ? p.bias[col] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int rbase = m0 + (wm * TM + i) * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rbase + (r & 3) + 8 * (r >> 2);
        const float v = acc[i][j][r] + bv;
        if (row < p.M) {
          const long orow = p.subpix ? orow_tab[row - m0] : row;
          out[orow * p.Cout + col] = v;
          s1 += v;
          s2 = fmaf(v, v, s2);
        }
      }
    }
    cs[j] = s1;
    cq[j] = s2;
  }
  if (do_stats) {
    // fused GroupNorm statistics (conv_blocks.py:186): per-channel sums of this tile -> LDS -> per-group fp64 partials
    __syncthreads();  // LDS tiles are dead; reuse them: chs[WM][BN][2]
    float* chs = smem;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const float s1 = cs[j] + __shfl_xor(cs[j], 32, 64);
      const float s2 = cq[j] + __shfl_xor(cq[j], 32, 64);
      if (lane < 32) {
        float* d = chs + ((wm * BN) + (wn * TN + j) * 32 + lane) * 2;
        d[0] = s1;
        d[1] = s2;
      }
    }
    __syncthreads();
    const int ngl = BN / p.gn_cpg;  // groups covered by this tile
    if (tid < ngl) {
      double s = 0, q = 0;
      for (int w = 0; w < WM; ++w) {
        const float* d = chs + (w * BN + tid * p.gn_cpg) * 2;
        for (int c = 0; c < p.gn_cpg; ++c) { s += (double)d[2 * c]; q += (double)d[2 * c + 1]; }
      }
      const int n = m0 / p.HWout, part = (m0 - n * p.HWout) / BM;
      double* o = p.gn_partial + (((long)n * p.gn_parts + part) * p.gn_groups + (n0 / p.gn_cpg + tid)) * 2;
      o[0] = s;
      o[1] = q;
    }
    if (p.gn_fin.stats) {
      __syncthreads();  // chs[] consumed
      double* red = reinterpret_cast<double*>(smem);  // 2*NT doubles + flag: far below the tile buffers' size
      gn_arrive_and_finalize(p.gn_fin, p.gn_partial, m0 / p.HWout, p.gn_groups, reinterpret_cast<volatile int*>(red + 2 * NT), red);
    }
  }
}

__global__ void splitk_reduce_kernel(const float* __restrict__ slabs, const float* __restrict__ bias, float* __restrict__ y,
                                     long n4, int Cout, int splitk, long slab) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const long e = i * 4;
    float4 s = *reinterpret_cast<const float4*>(slabs + e);
    for (int z = 1; z < splitk; ++z) {
      const float4 v = *reinterpret_cast<const float4*>(slabs + (long)z * slab + e);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (bias) {
      const float4 b = *reinterpret_cast<const float4*>(bias + (e % Cout));
      s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
    }
    *reinterpret_cast<float4*>(y + e) = s;
  }
}

// one thread per output element; any channel count, either layout at either edge
template <bool PIXEL_FAST>
__global__ void conv_direct_kernel(const ConvP p) {
  const long total = (long)p.M * p.Cout;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int m, co;
  if (PIXEL_FAST) {
    m = (int)(idx % p.M);
    co = (int)(idx / p.M);
  } else {
    co = (int)(idx % p.Cout);
    m = (int)(idx / p.Cout);
  }
  const int n = m / p.HWout;
  const int rem = m - n * p.HWout;
  const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
  float acc = p.bias ? p.bias[co] : 0.f;
  const float* wr = p.w + (long)co * p.K;
  for (int ky = 0; ky < p.KH; ++ky) {
    const int iy = oy * p.stride - p.pad + ky;
    if ((unsigned)iy >= (unsigned)p.Heff) continue;
    const int sy = iy >> p.ups;
    for (int kx = 0; kx < p.KW; ++kx) {
      const int ix = ox * p.stride - p.pad + kx;
      if ((unsigned)ix >= (unsigned)p.Weff) continue;
      const int sx = ix >> p.ups;
      const float* wt = wr + (ky * p.KW + kx) * p.Cin;
      if (p.in_nchw) {
        const float* xs = p.x1 + ((long)n * p.C1 * p.Hin + sy) * p.Win + sx;
        const long cs = (long)p.Hin * p.Win;
        for (int ci = 0; ci < p.C1; ++ci) acc = fmaf(xs[ci * cs], wt[ci], acc);
      } else {
        const float* xa = p.x1 + ((long)(n * p.Hin + sy) * p.Win + sx) * p.C1;
        if ((p.C1 & 3) == 0) {
          for (int ci = 0; ci < p.C1; ci += 4) {
            const float4 xv = *reinterpret_cast<const float4*>(xa + ci);
            const float4 wv = *reinterpret_cast<const float4*>(wt + ci);
            acc = fmaf(xv.x, wv.x, acc); acc = fmaf(xv.y, wv.y, acc);
            acc = fmaf(xv.z, wv.z, acc); acc = fmaf(xv.w, wv.w, acc);
          }
        } else {
          for (int ci = 0; ci < p.C1; ++ci) acc = fmaf(xa[ci], wt[ci], acc);
        }
        if (p.C2 > 0) {
          const float* xb = p.x2 + ((long)(n * p.Hin + sy) * p.Win + sx) * p.C2;
          const float* wb = wt + p.C1;
          for (int ci = 0; ci < p.C2; ++ci) acc = fmaf(xb[ci], wb[ci], acc);
        }
      }
    }
  }
  if (p.out_nchw)
    p.y[((long)(n * p.Cout + co) * p.Hout + oy) * p.Wout + ox] = acc;
  else
    p.y[(long)m * p.Cout + co] = acc;
}


// Small-Cin convolution (UNet in_conv 8->256, VAE inc_dec 8->512, VAE inc 3->64): K = KH*KW*Cin <= 160.
// A persistent block keeps a transposed weight tile W^T[k][co] (co <= 256) in LDS, then walks pixel groups:
// the im2col patch of 16 pixels goes to LDS (broadcast reads), lane = output channel => coalesced NHWC stores.
constexpr int kSmallPix = 16, kSmallCo = 256, kSmallMaxK = 160, kSmallPG = 4;

__global__ __launch_bounds__(256) void conv_smallcin_kernel(const ConvP p, int groups) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* wT = sm;                                   // [K][kSmallCo]
  float* patch = sm + (size_t)p.K * kSmallCo;       // [kSmallPG][kSmallPix][K]
  const int tid = threadIdx.x;
  const int co0 = blockIdx.y * kSmallCo;
  const int nco = min(kSmallCo, p.Cout - co0);
  for (int i = tid; i < nco * p.K; i += 256) {      // coalesced read of [co][k], transposed write
    const int co = i / p.K, k = i - co * p.K;
    wT[k * kSmallCo + co] = p.w[(long)(co0 + co) * p.K + k];
  }
  // thread = (pixel group pg of 4, channel quad cq of 64): 4 output channels x 16 pixels in registers
  const int pg = tid >> 6, cq = tid & 63;
  const bool active = cq * 4 < nco;
  float bias4[4] = {0.f, 0.f, 0.f, 0.f};
  if (active && p.bias) {
#pragma unroll
    for (int j = 0; j < 4; ++j) bias4[j] = p.bias[co0 + cq * 4 + j];
  }
  const int PK = kSmallPix * p.K;
  for (int g0 = blockIdx.x * kSmallPG; g0 < groups; g0 += gridDim.x * kSmallPG) {
    __syncthreads();
    for (int i = tid; i < kSmallPG * PK; i += 256) {
      const int gq = i / PK, r = i - gq * PK;
      const int px = r / p.K, k = r - px * p.K;
      const int tap = k / p.Cin, ci = k - tap * p.Cin;
      const int ky = tap / p.KW, kx = tap - ky * p.KW;
      const int m = (g0 + gq) * kSmallPix + px;
      float v = 0.f;
      if (m < p.M) {
        const int n = m / p.HWout, rem = m - n * p.HWout;
        const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
        const int iy = oy * p.stride - p.pad + ky, ix = ox * p.stride - p.pad + kx;
        if ((unsigned)iy < (unsigned)p.Heff && (unsigned)ix < (unsigned)p.Weff) {
          const int sy = iy >> p.ups, sx = ix >> p.ups;
          v = p.in_nchw ? p.x1[((long)(n * p.C1 + ci) * p.Hin + sy) * p.Win + sx]
                        : (ci < p.C1 ? p.x1[((long)(n * p.Hin + sy) * p.Win + sx) * p.C1 + ci]
                                     : p.x2[((long)(n * p.Hin + sy) * p.Win + sx) * p.C2 + (ci - p.C1)]);
        }
      }
      patch[i] = v;
    }
    __syncthreads();
    const int m0 = (g0 + pg) * kSmallPix;
    if (active && m0 < p.M) {
      float acc[kSmallPix][4];
#pragma unroll
      for (int q = 0; q < kSmallPix; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[q][j] = bias4[j];
      const float* pp = patch + pg * PK;
      for (int k = 0; k < p.K; ++k) {
        const float4 wv = *reinterpret_cast<const float4*>(wT + k * kSmallCo + cq * 4);
#pragma unroll
        for (int q = 0; q < kSmallPix; ++q) {
          const float xv = pp[q * p.K + k];
          acc[q][0] = fmaf(xv, wv.x, acc[q][0]); acc[q][1] = fmaf(xv, wv.y, acc[q][1]);
          acc[q][2] = fmaf(xv, wv.z, acc[q][2]); acc[q][3] = fmaf(xv, wv.w, acc[q][3]);
        }
      }
#pragma unroll
      for (int q = 0; q < kSmallPix; ++q)
        if (m0 + q < p.M) *reinterpret_cast<float4*>(p.y + (long)(m0 + q) * p.Cout + co0 + cq * 4) = make_float4(acc[q][0], acc[q][1], acc[q][2], acc[q][3]);
    }
  }
}

__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int KH, int KW) {
  const long total = (long)Cout * Cin * KH * KW;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
    // o indexes packed [co][ky][kx][ci]
    const int ci = (int)(o % Cin);
    long t = o / Cin;
    const int kx = (int)(t % KW); t /= KW;
    const int ky = (int)(t % KH);
    const int co = (int)(t / KH);
    out[o] = w[(((long)co * Cin + ci) * KH + ky) * KW + kx];
  }
}

// Sub-pixel weights of nearest-x2 + 3x3 (pad 1): phase (a, b) of the output reads source rows {y + a - 1, y + a} and columns
// {x + b - 1, x + b}; its 2x2 kernel is the sum of the 3x3 taps that land on the same source pixel:
//   a = 0: ty = 0 <- ky {0},    ty = 1 <- ky {1, 2};      a = 1: ty = 0 <- ky {0, 1},  ty = 1 <- ky {2}     (same for b / kx)
// out: [4 phases][Cout][2][2][Cin]
__global__ void pack_upconv_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin) {
  const long total = 4L * Cout * 4 * Cin;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
    const int ci = (int)(o % Cin);
    long t = o / Cin;
    const int tx = (int)(t & 1); t >>= 1;
    const int ty = (int)(t & 1); t >>= 1;
    const int co = (int)(t % Cout);
    const int ph = (int)(t / Cout);
    const int a = ph >> 1, b = ph & 1;
    const int ky0 = a == 0 ? (ty == 0 ? 0 : 1) : (ty == 0 ? 0 : 2), ky1 = a == 0 ? (ty == 0 ? 0 : 2) : (ty == 0 ? 1 : 2);
    const int kx0 = b == 0 ? (tx == 0 ? 0 : 1) : (tx == 0 ? 0 : 2), kx1 = b == 0 ? (tx == 0 ? 0 : 2) : (tx == 0 ? 1 : 2);
    float acc = 0.f;
    for (int ky = ky0; ky <= ky1; ++ky)
      for (int kx = kx0; kx <= kx1; ++kx) acc += w[(((long)co * Cin + ci) * 3 + ky) * 3 + kx];
    out[o] = acc;
  }
}

// [rows][K] fp32 (either packing) -> [rows][K/8][3 pieces][8] bf16: the exact 3-way split of MODE 1, done once at load time
__global__ void split_weight_kernel(const float* __restrict__ w, u32x4* __restrict__ out, long octets) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < octets; o += stride) {
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(w + o * 8), v1 = *reinterpret_cast<const f32x4*>(w + o * 8 + 4);
    u32x2 h0, m0, l0, h1, m1, l1;
    split_bf16x3(v0, h0, m0, l0);
    split_bf16x3(v1, h1, m1, l1);
    out[o * 3 + 0] = u32x4{h0.x, h0.y, h1.x, h1.y};
    out[o * 3 + 1] = u32x4{m0.x, m0.y, m1.x, m1.y};
    out[o * 3 + 2] = u32x4{l0.x, l0.y, l1.x, l1.y};
  }
}

// [rows][K] fp32 -> [rows][K] bf16 (round to nearest even): the weights of the opt-in MF_CONV_BF16 mode
__global__ void convert_weight_bf16_kernel(const float* __restrict__ w, u32x2* __restrict__ out, long quads) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < quads; o += stride) out[o] = round_bf16x4(*reinterpret_cast<const f32x4*>(w + o * 4));
}

// ------------------------------------------------------------------ host-side planning
struct TileCfg { int id, BM, BN, WM, WN, BK; };
const TileCfg kCfgs[] = {
    {1, 128, 128, 2, 2, 32}, {2, 128, 64, 2, 2, 32}, {3, 64, 128, 2, 2, 32}, {4, 64, 64, 2, 2, 32}, {5, 128, 32, 4, 1, 32}, {6, 64, 32, 2, 1, 32},
    {7, 128, 128, 4, 2, 32}, {8, 128, 128, 2, 4, 32}, {9, 128, 256, 2, 4, 32}, {10, 256, 128, 4, 2, 32},
    {11, 128, 128, 2, 2, 32}, {12, 64, 128, 2, 2, 32}, {13, 128, 64, 2, 2, 32},  // MF_CONV_FP32_SPLIT3_W3 only: single LDS buffer, two 4-wave workgroups per CU
    {23, 64, 128, 2, 2, 64}, {24, 64, 64, 2, 2, 64}, {27, 128, 128, 4, 2, 64}, {28, 128, 128, 2, 4, 64},  // BK = 64 (needs C1, C2 % 64 == 0)
};

struct Plan {
  bool igemm;
  TileCfg cfg;
  int splitk, nk_per_split;
  int Hout, Wout, Heff, Weff, M, K;
};

int fill_geometry(const MfConvDesc* d, Plan* pl) {
  MF_REQUIRE(d != nullptr, MF_EINVAL, "conv: null desc");
  MF_REQUIRE(d->N > 0 && d->Hin > 0 && d->Win > 0 && d->C1 > 0 && d->C2 >= 0 && d->Cout > 0, MF_EINVAL, "conv: bad dims");
  MF_REQUIRE((d->KH == 1 && d->KW == 1) || (d->KH == 3 && d->KW == 3), MF_EUNSUPPORTED, "conv: kernel %dx%d unsupported", d->KH, d->KW);
  MF_REQUIRE(d->stride == 1 || d->stride == 2, MF_EUNSUPPORTED, "conv: stride %d unsupported", d->stride);
  MF_REQUIRE(d->upsample >= 0 && d->upsample <= 2, MF_EINVAL, "conv: upsample flag");
  MF_REQUIRE(d->precision >= 0 && d->precision <= 4, MF_EINVAL, "conv: precision flag %d", d->precision);
  MF_REQUIRE(d->upsample != 2 || (d->KH == 3 && d->stride == 1 && d->pad == 1), MF_EINVAL, "conv: the sub-pixel form is nearest-x2 + 3x3 stride 1 pad 1");
  MF_REQUIRE(d->pad >= 0 && d->pad <= 1, MF_EUNSUPPORTED, "conv: pad %d unsupported", d->pad);
  MF_REQUIRE(!(d->in_layout == MF_LAYOUT_NCHW && d->C2 != 0), MF_EUNSUPPORTED, "conv: NCHW input with two sources");
  const int up = d->upsample ? 1 : 0;
  pl->Heff = d->Hin << up;
  pl->Weff = d->Win << up;
  pl->Hout = (pl->Heff + 2 * d->pad - d->KH) / d->stride + 1;
  pl->Wout = (pl->Weff + 2 * d->pad - d->KW) / d->stride + 1;
  MF_REQUIRE(pl->Hout > 0 && pl->Wout > 0, MF_EINVAL, "conv: empty output");
  const long M = (long)d->N * pl->Hout * pl->Wout;
  MF_REQUIRE(M < (1L << 31) && M * d->Cout < (1L << 40), MF_EUNSUPPORTED, "conv: problem too large");
  pl->M = (int)M;
  pl->K = (d->upsample == 2 ? 4 : d->KH * d->KW) * (d->C1 + d->C2);  // sub-pixel form: 2x2 taps per phase
  return MF_OK;
}

int make_plan(const MfConvDesc* d, Plan* pl) {
  int rc = fill_geometry(d, pl);
  if (rc) return rc;
  const int Cin = d->C1 + d->C2;
  pl->igemm = d->in_layout == MF_LAYOUT_NHWC && d->out_layout == MF_LAYOUT_NHWC && (d->C1 % 32 == 0) && (d->C2 % 32 == 0) &&
              (d->Cout % 32 == 0) && d->tile_hint >= 0;
  pl->splitk = 1;
  pl->nk_per_split = 0;
  const int taps = d->upsample == 2 ? 4 : d->KH * d->KW;
  const int hw_src = d->Hin * d->Win;
  if (d->upsample == 2) {
    MF_REQUIRE(pl->igemm && hw_src % 64 == 0, MF_EUNSUPPORTED, "conv: sub-pixel form needs the implicit-GEMM path and Hin*Win %% 64 == 0 (use upsample = 1)");
  }
  if (!pl->igemm) return MF_OK;
  int nk = taps * (Cin / 32);
  if (d->tile_hint > 0) {
    const TileCfg* c = nullptr;
    for (const auto& k : kCfgs) if (k.id == d->tile_hint) c = &k;
    MF_REQUIRE(c && d->Cout % c->BN == 0, MF_EINVAL, "conv: bad tile_hint %d for Cout %d", d->tile_hint, d->Cout);
    MF_REQUIRE(d->C1 % c->BK == 0 && d->C2 % c->BK == 0, MF_EINVAL, "conv: tile_hint %d needs channel counts divisible by %d", d->tile_hint, c->BK);
    MF_REQUIRE(d->precision == MF_CONV_FP32 || c->BK == 32, MF_EINVAL, "conv: tile_hint %d is not built for the split-bf16 mode", d->tile_hint);
    MF_REQUIRE(d->precision == MF_CONV_FP32_SPLIT3_W3 || c->id < 11 || c->id > 13, MF_EINVAL, "conv: tile_hint %d needs MF_CONV_FP32_SPLIT3_W3", d->tile_hint);
    pl->cfg = *c;
  } else {
    // From scripts/conv_sweep.py on MI355X (profiles/r01_conv_sweep.txt): the 8-wave 128x128 tile (2 waves per SIMD inside
    // ONE workgroup: half the LDS/L2 traffic of two 64x128 workgroups) is best or within 2 % of best for every shape with
    // Cout % 128 == 0; 64x64 for the 64-channel VAE level.  Split-K (below) tops the grid up to >= 512 workgroups.
    int id = 6;
    const bool c64 = d->C1 % 64 == 0 && d->C2 % 64 == 0;
    const double gflop = 2.0 * pl->M * (double)d->Cout * pl->K * 1e-9;
    if (d->precision != MF_CONV_FP32) {
      // split-bf16 mode (sweep: profiles/r01_conv_sweep_split.txt): the matrix work per chunk is 3/8 of the fp32 kernel's while the
      // staging is not smaller, so the tiles with a 64x64 per-wave footprint (128x256 / 256x128: half the LDS fragment traffic and
      // 3/4 of the staging per MAC) win on every large shape; one workgroup per CU (split-K below tops the grid up to 256).
      if (d->Cout % 256 == 0 && pl->M >= 128 && gflop >= 6.0 && !(d->upsample == 2 && hw_src % 128)) id = 9;
      else if (d->Cout % 128 == 0 && pl->M >= 256 && gflop >= 6.0 && !(d->upsample == 2 && hw_src % 256)) id = 10;
      else if (d->Cout % 128 == 0 && pl->M >= 128 && gflop >= 3.0 && !(d->upsample == 2 && hw_src % 128)) id = 8;
      else if (d->Cout % 64 == 0) id = 4;
      // short K and exactly enough 128x128 tiles to fill the chip once: no split-K, no slabs, statistics in the epilogue (4 % faster
      // than 128x256 + split-K 2 at K = 2304; at K = 4608 the wide tile wins again)
      if ((id == 9 || id == 10) && pl->K <= 2304 && !(d->upsample == 2 && hw_src % 128)) {
        const long t8 = (long)cdiv(pl->M, 128) * (d->Cout / 128);
        if (t8 >= 224 && t8 <= 256) id = 8;
      }
      // pre-split weights: the single-buffer two-workgroups-per-CU forms win where Cout is too narrow for the 256-wide tile
      // (VAE decoder levels: 128 ch 0.204 vs 0.227 ms, 64 ch 0.269 vs 0.280 ms; profiles/r01_conv_sweep_split.txt)
      if (d->precision == MF_CONV_FP32_SPLIT3_W3 && d->tile_hint == 0) {
        if (id == 10 && d->Cout == 128 && !(d->upsample == 2 && hw_src % 128)) id = 11;
        else if (id == 4 && pl->M >= 4096 && gflop >= 6.0 && !(d->upsample == 2 && hw_src % 128)) id = 13;
      }
    } else if (d->Cout % 128 == 0 && pl->M >= 128 && gflop >= 6.0 && !(d->upsample == 2 && hw_src % 128)) {
      id = 8;                              // 8 waves, 128x128: best for every large 3x3 shape
    } else if (d->Cout % 64 == 0) {
      id = c64 ? 24 : 4;                   // 64x64 (BK = 64 when the channels allow): short-K 1x1 residual convs, stride-2 convs and
    }                                      // other < 6 GFLOP problems are launch-cost-bound: more, smaller workgroups and less split-K
    for (const auto& k : kCfgs) if (k.id == id) pl->cfg = k;
  }
  nk = taps * (Cin / pl->cfg.BK);
  MF_REQUIRE(d->upsample != 2 || hw_src % pl->cfg.BM == 0, MF_EINVAL, "conv: sub-pixel form needs Hin*Win %% tile rows == 0");
  const long tiles = (long)cdiv(pl->M, pl->cfg.BM) * (d->Cout / pl->cfg.BN);
  int sk = 1;
  if (d->splitk_hint > 0) {
    sk = d->splitk_hint;
  } else {
    if (d->precision != MF_CONV_FP32 && pl->cfg.BM * pl->cfg.BN >= 128 * 128) {
      // >= 106 KB of LDS per workgroup in the split modes: one workgroup per CU -> at most one wave of workgroups (256)
      while (tiles * sk * 2 <= 256 && nk / (sk * 2) >= 4 && sk < 16) sk *= 2;
    } else {
      // aim for >= 2 workgroups per CU (256 CUs); keep >= 4 chunks per split for the 8-wave tile, >= 8 for the small tiles
      const int min_chunks = pl->cfg.WM * pl->cfg.WN == 8 ? 4 : 8;
      while (tiles * sk < 512 && nk / (sk * 2) >= min_chunks && sk < 16) sk *= 2;
    }
    // split mode: the bf16 MFMA adds its 16 products and the accumulator with truncation; keep one accumulation chain short
    // (<= 96 chunks of 32) so that the error stays at the fp32-MFMA kernel's level (tests/test_kernels_gpu.py, scripts/split_accuracy.py)
    if (d->precision != MF_CONV_FP32 && d->precision != MF_CONV_BF16) while (nk / sk > 96 && sk < 16) sk *= 2;
  }
  if (sk > nk) sk = nk;
  pl->nk_per_split = cdiv(nk, sk);
  pl->splitk = cdiv(nk, pl->nk_per_split);
  return MF_OK;
}

template <int BM, int BN, int WM, int WN, int BK, int MODE, bool FG>
int launch_igemm_fg(const ConvP& p, hipStream_t s) {
  constexpr int LDK = MODE == 0 ? BK + 4 : (MODE == 5 ? 20 : 52);
  const size_t lds = (size_t)(MODE == 4 ? 1 : 2) * (BM + BN) * LDK * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, WM, WN, BK, MODE, FG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  const int grid = p.tiles_m * p.tiles_n * p.splitk;
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, BK, MODE, FG>), dim3(grid), dim3(WM * WN * 64), lds, s, p);
  return check_launch("conv_igemm");
}

template <int BM, int BN, int WM, int WN, int BK = 32, int MODE = 0>
int launch_igemm(const ConvP& p, hipStream_t s) {
  return p.fastg ? launch_igemm_fg<BM, BN, WM, WN, BK, MODE, true>(p, s) : launch_igemm_fg<BM, BN, WM, WN, BK, MODE, false>(p, s);
}

}  // namespace

extern "C" {

int mf_pack_conv_weight_f32(const float* w, float* out, int Cout, int Cin, int KH, int KW, void* stream) {
  MF_REQUIRE(w && out && Cout > 0 && Cin > 0 && KH > 0 && KW > 0, MF_EINVAL, "pack_conv_weight: bad args");
  const long total = (long)Cout * Cin * KH * KW;
  ProfScope ps(MF_FAM_MISC, (hipStream_t)stream, 0, 8.0 * total);
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, out, Cout, Cin, KH, KW);
  return check_launch("pack_conv_weight");
}

// Number of per-sample partial records the convolution itself can emit for a following GroupNorm with G groups
// (0: it cannot -- split-K, direct kernels, a tile straddling two samples; use mf_gn_stats_partial_f32 then).
int mf_conv2d_gn_parts(const MfConvDesc* d, int G) {
  Plan pl;
  if (make_plan(d, &pl) != MF_OK || !pl.igemm || G <= 0 || d->Cout % G) return 0;
  const int cpg = d->Cout / G, HW = pl.Hout * pl.Wout;
  if (pl.splitk > 1) return stats_lds_bytes(d->Cout / stats_slices(d->N, HW, d->Cout, G)) <= 64 * 1024 ? stats_chunks(HW) : 0;  // split-K reducer
  if (HW % pl.cfg.BM || pl.cfg.BN % cpg) return 0;                                          // emitted by the conv epilogue
  return HW / pl.cfg.BM;
}

int mf_pack_upconv_weight_f32(const float* w, float* out, int Cout, int Cin, void* stream) {
  MF_REQUIRE(w && out && Cout > 0 && Cin > 0, MF_EINVAL, "pack_upconv_weight: bad args");
  const long total = 16L * Cout * Cin;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(pack_upconv_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, out, Cout, Cin);
  return check_launch("pack_upconv_weight");
}

int mf_conv2d_is_igemm(const MfConvDesc* d) {
  Plan pl;
  return d && make_plan(d, &pl) == MF_OK && pl.igemm ? 1 : 0;
}

int mf_split_conv_weight_bf16x3(const float* w_packed, void* out, long rows, int K, void* stream) {
  MF_REQUIRE(w_packed && out && rows > 0 && K > 0 && K % 8 == 0, MF_EINVAL, "split_conv_weight: bad args (K %% 8 == 0)");
  const long octets = rows * (K / 8);
  ProfScope ps(MF_FAM_MISC, (hipStream_t)stream, 0, 10.0 * rows * K);
  const int blocks = (int)((octets + 255) / 256 > 4096 ? 4096 : (octets + 255) / 256);
  hipLaunchKernelGGL(split_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_packed, reinterpret_cast<u32x4*>(out), octets);
  return check_launch("split_conv_weight");
}

int mf_convert_conv_weight_bf16(const float* w_packed, void* out, long rows, int K, void* stream) {
  MF_REQUIRE(w_packed && out && rows > 0 && K > 0 && K % 8 == 0, MF_EINVAL, "convert_conv_weight_bf16: bad args (K %% 8 == 0)");
  const long quads = rows * (K / 4);
  ProfScope ps(MF_FAM_MISC, (hipStream_t)stream, 0, 6.0 * rows * K);
  const int blocks = (int)((quads + 255) / 256 > 4096 ? 4096 : (quads + 255) / 256);
  hipLaunchKernelGGL(convert_weight_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_packed, reinterpret_cast<u32x2*>(out), quads);
  return check_launch("convert_conv_weight_bf16");
}

/* 1 if `d` (with upsample = 2) can run in the sub-pixel form, else 0 (then use upsample = 1 with the regular packing) */
int mf_conv2d_subpixel_ok(const MfConvDesc* d) {
  Plan pl;
  return d && d->upsample == 2 && make_plan(d, &pl) == MF_OK && pl.igemm ? 1 : 0;
}

size_t mf_conv2d_workspace_bytes(const MfConvDesc* d) {
  Plan pl;
  if (make_plan(d, &pl) != MF_OK) return 0;
  if (!pl.igemm || pl.splitk <= 1) return 0;
  return (size_t)pl.splitk * pl.M * d->Cout * sizeof(float);
}

struct GnOut { double* partial; float* stats; int32_t* counter; int G; float eps; };
static int conv2d_impl(const float* x1, const float* x2, const float* w, const float* bias, float* y, void* workspace, size_t workspace_bytes,
                       const GnOut& gn, const MfConvDesc* d, void* stream);

int mf_conv2d_f32(const float* x1, const float* x2, const float* w, const float* bias, float* y, void* workspace,
                  size_t workspace_bytes, const MfConvDesc* d, void* stream) {
  return conv2d_impl(x1, x2, w, bias, y, workspace, workspace_bytes, GnOut{nullptr, nullptr, nullptr, 0, 0.f}, d, stream);
}

int mf_conv2d_gn_f32(const float* x1, const float* x2, const float* w, const float* bias, float* y, void* workspace,
                     size_t workspace_bytes, double* gn_partial, float* gn_stats, int32_t* gn_counter, int G, float eps, const MfConvDesc* d,
                     void* stream) {
  MF_REQUIRE(gn_partial && mf_conv2d_gn_parts(d, G) > 0, MF_EUNSUPPORTED, "conv_gn: this convolution cannot emit GroupNorm partials (mf_conv2d_gn_parts == 0)");
  MF_REQUIRE((gn_stats == nullptr) == (gn_counter == nullptr), MF_EINVAL, "conv_gn: gn_stats and gn_counter go together");
  MF_REQUIRE(G <= 256, MF_EUNSUPPORTED, "conv_gn: G > 256");
  return conv2d_impl(x1, x2, w, bias, y, workspace, workspace_bytes, GnOut{gn_partial, gn_stats, gn_counter, G, eps}, d, stream);
}

static int conv2d_impl(const float* x1, const float* x2, const float* w, const float* bias, float* y, void* workspace, size_t workspace_bytes,
                       const GnOut& gn, const MfConvDesc* d, void* stream) {
  double* gn_partial = gn.partial;
  const int G = gn.G;
  Plan pl;
  int rc = make_plan(d, &pl);
  if (rc) return rc;
  MF_REQUIRE(x1 && w && y, MF_EINVAL, "conv: null pointer");
  MF_REQUIRE((d->precision != MF_CONV_FP32_SPLIT3_W3 && d->precision != MF_CONV_BF16) || pl.igemm, MF_EINVAL,
             "conv: MF_CONV_FP32_SPLIT3_W3 / MF_CONV_BF16 (converted weights) exist on the implicit-GEMM path only (ask mf_conv2d_is_igemm)");
  MF_REQUIRE(d->C2 == 0 || x2 != nullptr, MF_EINVAL, "conv: C2 > 0 but x2 is null");
  hipStream_t s = (hipStream_t)stream;
  ConvP p;
  p.x1 = x1; p.x2 = x2; p.w = w; p.bias = bias; p.y = y;
  p.N = d->N; p.Hin = d->Hin; p.Win = d->Win; p.C1 = d->C1; p.C2 = d->C2; p.Cin = d->C1 + d->C2; p.Cout = d->Cout;
  p.Hout = pl.Hout; p.Wout = pl.Wout; p.Heff = pl.Heff; p.Weff = pl.Weff;
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.ups = d->upsample == 1 ? 1 : 0;
  p.subpix = d->upsample == 2 ? 1 : 0; p.hw_src = d->Hin * d->Win;
  if (p.subpix) { p.KH = p.KW = 2; p.Heff = d->Hin; p.Weff = d->Win; }  // 2x2 taps on the SOURCE grid, offsets set per phase
  p.M = pl.M; p.K = pl.K; p.HWout = pl.Hout * pl.Wout;
  p.in_nchw = d->in_layout == MF_LAYOUT_NCHW; p.out_nchw = d->out_layout == MF_LAYOUT_NCHW;
  p.cchunks = pl.igemm ? p.Cin / pl.cfg.BK : 0; p.nk = p.KH * p.KW * p.cchunks; p.nk_per_split = pl.nk_per_split; p.splitk = pl.splitk;
  p.tiles_m = 0; p.tiles_n = 0; p.slab = (long)pl.M * d->Cout;
  p.bytes1 = p.bytes2 = p.bytesw = 0;
  p.gn_partial = gn_partial; p.gn_groups = G; p.gn_cpg = G > 0 ? d->Cout / G : 1;
  p.gn_parts = (gn_partial && pl.igemm && pl.splitk == 1) ? (pl.Hout * pl.Wout) / pl.cfg.BM : 0;
  if (pl.igemm && pl.splitk > 1) p.gn_partial = nullptr;  // the reducer, not the conv kernel, emits them
  p.gn_fin = GnFinal{};
  p.fastg = (p.ups == 0 && (long)d->N * d->Hin * d->Win < (1L << 24) && d->C1 < (1 << 22) && d->C2 < (1 << 22)) ? 1 : 0;
  static const bool generic_gather = getenv("MF_CONV_GENERIC_GATHER") != nullptr;  // A/B switch (scripts), read once
  if (generic_gather) p.fastg = 0;
  const int HWo = pl.Hout * pl.Wout;
  if (gn.stats) {  // last-arriver finalize (gn.counter: zero on entry, zero again on exit)
    if (pl.igemm && pl.splitk == 1)
      p.gn_fin = GnFinal{gn.stats, gn.counter, (HWo / pl.cfg.BM) * (d->Cout / pl.cfg.BN), HWo / pl.cfg.BM, (double)HWo * (d->Cout / G), gn.eps};
  }
  // algorithmic FLOPs of the reference op (the sub-pixel form does 4/9 of the MACs of nearest-x2 + 3x3)
  const double flops = 2.0 * pl.M * (double)d->Cout * (d->upsample == 2 ? 9.0 * (d->C1 + d->C2) : (double)pl.K);
  const double bytes = 4.0 * ((double)d->N * d->Hin * d->Win * p.Cin + (double)d->Cout * pl.K + (double)pl.M * d->Cout);

  if (!pl.igemm && !p.out_nchw && p.Cin <= 16 && pl.K <= kSmallMaxK && d->Cout % 64 == 0 &&
      ((size_t)pl.K * kSmallCo + (size_t)kSmallPG * kSmallPix * pl.K) * sizeof(float) <= 150 * 1024) {
    ProfScope ps(MF_FAM_CONV_DIRECT, s, flops, bytes);
    const int groups = cdiv(pl.M, kSmallPix);
    const int cotiles = cdiv(d->Cout, kSmallCo);
    const size_t lds = ((size_t)pl.K * kSmallCo + (size_t)kSmallPG * kSmallPix * pl.K) * sizeof(float);
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_smallcin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr = true;
    }
    int gx = cdiv(groups, kSmallPG);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(conv_smallcin_kernel, dim3(gx, cotiles), dim3(256), lds, s, p, groups);
    return check_launch("conv_smallcin");
  }
  if (!pl.igemm) {
    ProfScope ps(MF_FAM_CONV_DIRECT, s, flops, bytes);
    const long total = (long)pl.M * d->Cout;
    const int blocks = (int)((total + 255) / 256);
    if (p.out_nchw)
      hipLaunchKernelGGL(conv_direct_kernel<true>, dim3(blocks), dim3(256), 0, s, p);
    else
      hipLaunchKernelGGL(conv_direct_kernel<false>, dim3(blocks), dim3(256), 0, s, p);
    return check_launch("conv_direct");
  }

  p.tiles_m = cdiv(pl.M, pl.cfg.BM);
  p.tiles_n = d->Cout / pl.cfg.BN;
  {
    const double b1 = 4.0 * d->N * d->Hin * d->Win * d->C1, b2 = 4.0 * d->N * d->Hin * d->Win * d->C2, bw = (d->precision == MF_CONV_FP32_SPLIT3_W3 ? 6.0 : d->precision == MF_CONV_BF16 ? 2.0 : 4.0) * d->Cout * pl.K * (p.subpix ? 4 : 1);
    MF_REQUIRE(b1 < 4294967040.0 && b2 < 4294967040.0 && bw < 4294967040.0, MF_EUNSUPPORTED,
               "conv: a source tensor exceeds the 4 GiB buffer-descriptor range (shard the batch)");
    p.bytes1 = (unsigned)b1; p.bytes2 = (unsigned)b2; p.bytesw = (unsigned)bw;
  }
  if (pl.splitk > 1) {
    const size_t need = (size_t)pl.splitk * pl.M * d->Cout * sizeof(float);
    MF_REQUIRE(workspace && workspace_bytes >= need, MF_EWORKSPACE, "conv: workspace %zu < %zu", workspace_bytes, need);
    p.y = reinterpret_cast<float*>(workspace);
  }
  {
    ProfScope ps(MF_FAM_CONV_IGEMM, s, flops, bytes);
    if (d->precision == MF_CONV_FP32_SPLIT3_CHUNKSUM) {
      switch (pl.cfg.id) {
        case 1: rc = launch_igemm<128, 128, 2, 2, 32, 2>(p, s); break;
        case 2: rc = launch_igemm<128, 64, 2, 2, 32, 2>(p, s); break;
        case 3: rc = launch_igemm<64, 128, 2, 2, 32, 2>(p, s); break;
        case 4: rc = launch_igemm<64, 64, 2, 2, 32, 2>(p, s); break;
        case 6: rc = launch_igemm<64, 32, 2, 1, 32, 2>(p, s); break;
        case 7: rc = launch_igemm<128, 128, 4, 2, 32, 2>(p, s); break;
        case 8: rc = launch_igemm<128, 128, 2, 4, 32, 2>(p, s); break;
        case 9: rc = launch_igemm<128, 256, 2, 4, 32, 2>(p, s); break;
        case 10: rc = launch_igemm<256, 128, 4, 2, 32, 2>(p, s); break;
        default: set_error("conv: tile config %d is not built for the split-bf16 chunk-sum mode", pl.cfg.id); rc = MF_EINVAL;
      }
    } else if (d->precision == MF_CONV_BF16) {
      switch (pl.cfg.id) {
        case 1: rc = launch_igemm<128, 128, 2, 2, 32, 5>(p, s); break;
        case 2: rc = launch_igemm<128, 64, 2, 2, 32, 5>(p, s); break;
        case 3: rc = launch_igemm<64, 128, 2, 2, 32, 5>(p, s); break;
        case 4: rc = launch_igemm<64, 64, 2, 2, 32, 5>(p, s); break;
        case 6: rc = launch_igemm<64, 32, 2, 1, 32, 5>(p, s); break;
        case 7: rc = launch_igemm<128, 128, 4, 2, 32, 5>(p, s); break;
        case 8: rc = launch_igemm<128, 128, 2, 4, 32, 5>(p, s); break;
        case 9: rc = launch_igemm<128, 256, 2, 4, 32, 5>(p, s); break;
        case 10: rc = launch_igemm<256, 128, 4, 2, 32, 5>(p, s); break;
        default: set_error("conv: tile config %d is not built for the bf16 mode", pl.cfg.id); rc = MF_EINVAL;
      }
    } else if (d->precision == MF_CONV_FP32_SPLIT3_W3) {
      switch (pl.cfg.id) {
        case 1: rc = launch_igemm<128, 128, 2, 2, 32, 3>(p, s); break;
        case 2: rc = launch_igemm<128, 64, 2, 2, 32, 3>(p, s); break;
        case 3: rc = launch_igemm<64, 128, 2, 2, 32, 3>(p, s); break;
        case 4: rc = launch_igemm<64, 64, 2, 2, 32, 3>(p, s); break;
        case 6: rc = launch_igemm<64, 32, 2, 1, 32, 3>(p, s); break;
        case 7: rc = launch_igemm<128, 128, 4, 2, 32, 3>(p, s); break;
        case 8: rc = launch_igemm<128, 128, 2, 4, 32, 3>(p, s); break;
        case 9: rc = launch_igemm<128, 256, 2, 4, 32, 3>(p, s); break;
        case 10: rc = launch_igemm<256, 128, 4, 2, 32, 3>(p, s); break;
        case 11: rc = launch_igemm<128, 128, 2, 2, 32, 4>(p, s); break;
        case 12: rc = launch_igemm<64, 128, 2, 2, 32, 4>(p, s); break;
        case 13: rc = launch_igemm<128, 64, 2, 2, 32, 4>(p, s); break;
        default: set_error("conv: tile config %d is not built for the split-bf16 mode", pl.cfg.id); rc = MF_EINVAL;
      }
    } else if (d->precision == MF_CONV_FP32_SPLIT3) {
      switch (pl.cfg.id) {
        case 1: rc = launch_igemm<128, 128, 2, 2, 32, 1>(p, s); break;
        case 2: rc = launch_igemm<128, 64, 2, 2, 32, 1>(p, s); break;
        case 3: rc = launch_igemm<64, 128, 2, 2, 32, 1>(p, s); break;
        case 4: rc = launch_igemm<64, 64, 2, 2, 32, 1>(p, s); break;
        case 6: rc = launch_igemm<64, 32, 2, 1, 32, 1>(p, s); break;
        case 7: rc = launch_igemm<128, 128, 4, 2, 32, 1>(p, s); break;
        case 8: rc = launch_igemm<128, 128, 2, 4, 32, 1>(p, s); break;
        case 9: rc = launch_igemm<128, 256, 2, 4, 32, 1>(p, s); break;
        case 10: rc = launch_igemm<256, 128, 4, 2, 32, 1>(p, s); break;
        default: set_error("conv: tile config %d is not built for the split-bf16 mode", pl.cfg.id); rc = MF_EINVAL;
      }
    } else
    switch (pl.cfg.id) {
      case 1: rc = launch_igemm<128, 128, 2, 2>(p, s); break;
      case 2: rc = launch_igemm<128, 64, 2, 2>(p, s); break;
      case 3: rc = launch_igemm<64, 128, 2, 2>(p, s); break;
      case 4: rc = launch_igemm<64, 64, 2, 2>(p, s); break;
      case 5: rc = launch_igemm<128, 32, 4, 1>(p, s); break;
      case 6: rc = launch_igemm<64, 32, 2, 1>(p, s); break;
      case 7: rc = launch_igemm<128, 128, 4, 2>(p, s); break;
      case 8: rc = launch_igemm<128, 128, 2, 4>(p, s); break;
      case 9: rc = launch_igemm<128, 256, 2, 4>(p, s); break;
      case 10: rc = launch_igemm<256, 128, 4, 2>(p, s); break;
      case 23: rc = launch_igemm<64, 128, 2, 2, 64>(p, s); break;
      case 24: rc = launch_igemm<64, 64, 2, 2, 64>(p, s); break;
      case 27: rc = launch_igemm<128, 128, 4, 2, 64>(p, s); break;
      case 28: rc = launch_igemm<128, 128, 2, 4, 64>(p, s); break;
      default: set_error("conv: no tile config"); rc = MF_EINVAL;
    }
  }
  if (rc) return rc;
  if (pl.splitk > 1) {
    if (gn_partial) {  // reduction + bias + GroupNorm partial statistics in one streaming pass
      const int HW = pl.Hout * pl.Wout;
      ProfScope ps(MF_FAM_SPLITK_REDUCE, s, 0, 4.0 * pl.M * d->Cout * (pl.splitk + 1));
      const int slices = stats_slices(d->N, HW, d->Cout, G), chunks = stats_chunks(HW);
      const GnFinal fin = gn.stats ? GnFinal{gn.stats, gn.counter, chunks * slices, chunks, (double)HW * (d->Cout / G), gn.eps} : GnFinal{};
      hipLaunchKernelGGL(gn_partial_kernel<true>, dim3(chunks, d->N, slices), dim3(kStatsThreads), stats_lds_bytes(d->Cout / slices), s,
                         reinterpret_cast<const float*>(workspace), gn_partial, HW, d->Cout, G, pl.splitk, p.slab, bias, y, fin);
      return check_launch("splitk_reduce_stats");
    }
    const long n4 = (long)pl.M * d->Cout / 4;
    ProfScope ps(MF_FAM_SPLITK_REDUCE, s, 0, 4.0 * pl.M * d->Cout * (pl.splitk + 1));
    const int blocks = (int)((n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const float*>(workspace), bias, y, n4,
                       d->Cout, pl.splitk, p.slab);
    return check_launch("splitk_reduce");
  }
  return MF_OK;
}

}  // extern "C"
