// conv_f16x2.h -- implicit-GEMM convolution on operands that are ALREADY split into fp16 pairs in HBM (MF_CONV_FP32_F16X2).
// Included by conv.hip only.  DESIGN.md section 3 has the measurements behind the choices.
//
// Arithmetic.  Every fp32 value x (activation or weight) is stored as two fp16 numbers  x ~ hi + lo' / 2048,
//   hi = RN16(x),  lo' = RN16((x - hi) * 2048)   (the subtraction and the scaling are exact)
// which keeps 23 of the 24 significand bits (|x - hi - lo'/2048| <= 2^-23 |x|, i.e. at most one ulp of the fp32 value, zero for 3 values
// out of 4).  A product is accumulated in fp32 on the fp16 matrix cores (v_mfma_f32_32x32x16_f16, 16x the fp32-MFMA rate) as three terms
//   main  += wh * xh            cross += wh * xl' + wl' * xh            result = main + cross / 2048
// the dropped term wl*xl is < 2^-22 |w x|.  Three matrix instructions per product instead of the six of the exact 3 x bf16 split
// (MF_CONV_FP32_SPLIT3_W3).  Range: every operand tensor carries a per-sample power-of-two scale (split_f16.h) -- the kernel multiplies
// its accumulators by the scales of the sample a pixel belongs to, and re-scales them once when the K loop moves from the first source
// of a fused concat to the second (the two tensors have their own scales).
//
// Data movement.  Both operands go HBM/L2 -> LDS with buffer_load_dwordx4 ... lds (LDS-DMA: no VGPR staging, no ds_write, no VALU on
// the data), three LDS stages, one raw s_barrier per 32-channel chunk in the MIDDLE of the chunk's MFMA stream, counted vmcnt so that
// two chunks stay in flight across the barrier.  A row of a chunk (32 channels of one filter tap) is 128 contiguous bytes in HBM:
// [4 groups of 8 channels][hi | lo'][8 fp16]; the LDS image is lane-linear per DMA instruction (8 rows x 128 B), and bank conflicts of
// the ds_read_b128 fragment reads are avoided by permuting the eight 16-byte slots of a row by (row >> 1) & 7 on the SOURCE address,
// with the same involution on the read (CDNA guide rule 21).  Zero padding, rows past M: out-of-range buffer offsets (the DMA writes 0).
//
// Orientation.  The MFMA A operand is the WEIGHT fragment and B the activation fragment, so a lane ends up with 4 CONSECUTIVE output
// channels of one pixel per accumulator quad: the epilogue goes through LDS with 16-byte writes and leaves as full 128-bit rows
// (fp32 NHWC and, optionally, the fp16-pair form for the next convolution).
#pragma once
#include "common.h"
#include "split_f16.h"

#ifndef MFC2_HZ
#define MFC2_HZ 0   // diagnostic hooks (below): 0 in the product build
#endif
#ifndef MFC2_OUT_STORE
#define MFC2_OUT_STORE 0   // 0: plain output stores; 1: non-temporal (A/B builds: medfusion_amd.build.build_variant; r03: +2 % on the convolution alone, nothing on the step)
#endif

namespace mfc2 {
using namespace mf;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// GroupNorm + Swish + residual + embedding fused INTO the convolution's launch (round 4; conv_blocks.py:185-191,236-240,360-363 in one kernel).
// The workgroups that hold the final values of a tile keep them in registers, publish their partial GroupNorm records, meet the other
// tiles of their SAMPLE at a counter (`rv`: one arrive / depart pair per sample, zero between launches), finalize mean / rstd from the
// records of the whole sample and apply the normalisation to their own tile on the way out -- the un-normalised convolution output never
// goes to memory, and the apply pass (a launch boundary + ~10 us per GroupNorm) disappears.  The host only asks for this when EVERY
// workgroup of the launch is resident at once (grid <= CUs x occupancy: nobody waits for a workgroup that cannot start); a rendezvous that
// does not complete within 50 ms (another process filling the device with ITS waiting workgroups) raises rv_err, every later launch stops
// waiting, and the host re-runs the loop un-fused (pipeline.py).  on == 0: the plain epilogue.
struct FuseP {
  int on, act;
  float eps, bconst;
  double count;                 // elements of a group: HW * (Cout / G)
  const float* gamma;           // [Cout] or null (both)
  const float* beta;
  const float* res_f32;         // residual as fp32 NHWC, or
  const void* res_pairs;        // as fp16 pairs (scaled by res_bound), or neither
  const float* res_bound;       // [N] bound of |residual|, or
  const float* res_slots;       // [N][res_nslots] the slot maxima its convolution left
  int res_nslots;
  int tiles_per_sample;         // arrivals a sample's counter waits for
  const float* emb;             // [N][emb_stride] embedding rows added per (n, c), or null
  long emb_stride;
  const float* emb_bound;       // [N]
  float* out_f32;               // fp32 NHWC result, or null (pairs only)
  void* out_pairs;              // fp16-pair result (always)
  float* out_bound;             // [N] bound the pairs were scaled with
  unsigned* rv;                 // [N][2] arrive / depart
  unsigned* rv_err;             // one word: a rendezvous timed out
};

struct ConvP2 {
  const void* x1;   // [N][Hin][Win][C1/8][2][8] fp16 pairs
  const void* x2;   // second source of the fused channel concat, or null
  const void* w;    // [phases][Cout][taps][Cin/8][2][8] fp16 pairs
  const float* bias;
  float* y;         // fp32 NHWC output, or the split-K slabs
  const float* bound1;   // [N] per-sample bounds the sources were scaled with (split_f16.h), or null = unscaled
  const float* bound2;
  int wexp;              // the weights were split as w * 2^-wexp
  float* out_bound;      // optional [N][slots]: measured max |y| of every (tile, wave) of a sample (splitk == 1, a tile inside one sample)
  int bound_slots;
  int N, Hin, Win, C1, C2, Cin, Cout;
  int Hout, Wout, Heff, Weff, KH, KW, stride, pad;
  int M, K, HWout;
  int cgroups, cg_per_split, splitk;   // 32-channel chunks: total, per split-K slice (a slice holds ALL taps of its chunks)
  int tiles_m, tiles_n;
  long slab;
  unsigned bytes1, bytes2, bytesw;
  int subpix, hw_src;
  double* gn_partial;   // optional fused GroupNorm statistics [N][gn_parts][G][2] (splitk == 1, or tree)
  int gn_groups, gn_parts, gn_cpg;
  // split-K reduced INSIDE the launch (tree != 0, splitk a power of two): the partial tiles meet pairwise, level by level; at every level
  // both partners store their tile (write-through, agent scope), bump the pair's counter, and the one that arrives second adds its
  // partner's tile and goes on -- a + b does not depend on who adds, so the result is deterministic.  The last one runs the epilogue.
  int tree;
  float* handoff;       // [tiles][2 (splitk - 1) slots][BM x BN floats]
  unsigned* sync;       // [tiles][splitk - 1] counters, zero between launches (the second arriver of a pair resets its counter)
  FuseP fz;
  // the output ALSO as fp16 pairs, scaled per sample by a bound DERIVED from the operands (round 4): |y| <= bound(x1) wl1_1 + bound(x2) wl1_2 + bmax,
  // wl1 = the largest L1 norm of a filter over the source's channels -- no measuring pass, no separate split launch behind the convolutions
  // whose output feeds convolutions un-normalised (down- / up-sampling).  Final-value workgroups only (splitk == 1 or tree).
  void* y_pairs;
  float* y_pair_bound;   // [N] written
  float wl1_1, wl1_2, bmax;
#if MFC2_HZ & (256 | 512)
  float* dbg;           // diagnostic builds: [tiles][waves][TM][TN][16][64] the accumulators of the surviving workgroup right behind the tree
#endif
};

__device__ __forceinline__ int xcd_remap2(int bid, int total) {  // bijective; block b runs on XCD b % 8
  const int q = total >> 3, r = total & 7;
  const int xcd = bid & 7, within = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + within;
}

// Diagnostic builds (scripts/pk_hunt.py): -DMFC2_HZ=<bit mask> pads single spots of the epilogue with wait states / full waits so that a
// sporadic wrong result can be attributed to ONE producer -> consumer pair.  0 (the product build): every hook expands to nothing.
#ifndef MFC2_HZ
#define MFC2_HZ 0
#endif
#define MFC2_HZ_ON(BIT) ((MFC2_HZ >> (BIT)) & 1)
#ifndef MFC2_HZ_PAD
#define MFC2_HZ_PAD 7
#endif

// un-packed fp32 arithmetic the compiler cannot pair into v_pk_*_f32 (hooks 5, 6)
__device__ __forceinline__ float hz_add(float a, float b) { float r; asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float hz_mul(float a, float b) { float r; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

#define MFC2_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
// MFC2_HZ bit 9 (diagnostic build, scripts/conv_timeline.py): workgroup `blockIdx.x` leaves the 100 MHz real-time counter at four points of its
// life in p.dbg (as uint64 [grid][8]): 0 entry, 1 first chunk landed (end of the ramp), 2 end of the K loop, 3 end of the epilogue; inside the
// epilogue: 4 every wave has left the loop (barrier), 5 split-K tree done, 6 outputs stored (issued), 7 unused
#if MFC2_HZ & 512
#define MFC2_STAMP(I) { if (p.dbg && threadIdx.x == 0) reinterpret_cast<unsigned long*>(p.dbg)[(long)blockIdx.x * 8 + (I)] = __builtin_amdgcn_s_memrealtime(); }
#else
#define MFC2_STAMP(I)
#endif
#define MFC2_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// TERMS = 3: the fp32-class arithmetic above.  TERMS = 1 (MF_CONV_F16, opt-in REDUCED precision): only main += wh * xh -- the operands
// rounded to fp16 (11 significant bits), one matrix instruction per product, fp32 accumulate; same operand images, same DMA, the lo'
// pieces are simply never read from LDS.
template <int BM, int BN, int WM, int WN, int NST, int TERMS = 3>
__global__ __launch_bounds__(WM * WN * 64, 2) void conv_f16x2_kernel(const ConvP2 p) {
  static_assert(TERMS == 3 || TERMS == 1, "three product terms (fp16 pairs) or one (fp16)");
  constexpr int PC = TERMS == 3 ? 2 : 1;                    // pieces of a fragment that are read: hi and lo', or hi only
  constexpr int NW = WM * WN;   // waves per workgroup: 8 (one workgroup per CU) or 4 (two per CU: independent barrier cadences)
  static_assert(NW == 8 || NW == 4, "4 or 8 waves");
  constexpr int FM = BM / WM, FN = BN / WN, TM = FM / 32, TN = FN / 32;
  static_assert(TM >= 1 && TN >= 1 && FM % 32 == 0 && FN % 32 == 0, "per-wave footprint");
  constexpr int GP = BM / (8 * NW), GQ = BN / (8 * NW), NL = GP + GQ;   // DMA instructions per wave and chunk (8 rows x 128 B each)
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile");
  constexpr int ROWB = 128, STAGE = (BM + BN) * ROWB;
  constexpr int NM = TERMS * TM * TN;                       // MFMAs per 16-deep step
  constexpr int NR = PC * (TM + TN);                        // fragment reads per step
  constexpr int NF = (2 * NM + 2) / 3;                      // the reads of a step are issued behind its first NF MFMAs (the rest cover their latency)
  static_assert(NST >= 2 && NST <= 6 && NST * STAGE <= 160 * 1024 && (NST - 1) * NL <= 63, "LDS stages");
  static_assert((NR + NF - 1) / NF <= 3, "at most three reads per slot");

  extern __shared__ __attribute__((aligned(1024))) char smem[];
  MFC2_STAMP(0)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  const int total = p.tiles_m * p.tiles_n * p.splitk;
  const int logical = xcd_remap2(blockIdx.x, total);
  const int tile_m = logical % p.tiles_m;
  const int rest = logical / p.tiles_m;
  const int tile_n = rest % p.tiles_n;
  const int kz = rest / p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int cg_beg = kz * p.cg_per_split;
  const int cg_end = min(p.cgroups, cg_beg + p.cg_per_split);
  const int taps = p.KH * p.KW;
  const int nit = (cg_end - cg_beg) * taps;

  // ---- DMA addressing.  Instruction i of this wave moves tile rows 8 (wave + NW i) + (lane >> 3); LDS slot (lane & 7) of a row holds
  // source slot (lane & 7) ^ key, key = (row >> 1) & 7 = (4 (wave & 1) + (lane >> 4)) & 7 for every i.
  const int lrow = lane >> 3;
  const unsigned slot16 = (unsigned)(((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7)) * 16);
  int a_pix[GP], a_inv[GP];
  // (this arithmetic runs on one wave per SIMD with nothing to hide behind -- ~1 us of a launch's ramp, scripts/conv_timeline.py -- so: divisions
  // by the two uniform divisors through one float reciprocal each + a +-1 correction, exact below 2^23; the 9-bit tap mask from 3 + 3 row /
  // column tests instead of 9 x 2)
  const bool small_m = p.M < (1 << 23);
  const float r_hw = 1.0f / (float)p.HWout, r_w = 1.0f / (float)(p.subpix ? p.Win : p.Wout), r_src = 1.0f / (float)p.hw_src;
  auto qdiv = [small_m](int a, int d, float rd) {
    if (!small_m) return a / d;
    int q = (int)((float)a * rd);
    const int r = a - q * d;
    q += (r >= d ? 1 : 0) - (r < 0 ? 1 : 0);
    return q;
  };
#pragma unroll
  for (int i = 0; i < GP; ++i) {
    const int m = m0 + 8 * (wave + NW * i) + lrow;
    int n_ = 0, iy0 = -(1 << 28), ix0 = 0;   // rows past M: every tap "outside"
    if (m < p.M) {
      const int n = qdiv(m, p.HWout, r_hw);
      const int rem = m - n * p.HWout;
      n_ = n * p.Hin;
      if (p.subpix) {  // m = (n, phase, y, x) over the SOURCE grid; output pixel (2y + a, 2x + b)
        const int ph = qdiv(rem, p.hw_src, r_src), r2 = rem - ph * p.hw_src;
        const int y = qdiv(r2, p.Win, r_w), x = r2 - y * p.Win;
        iy0 = y + (ph >> 1) - 1;
        ix0 = x + (ph & 1) - 1;
      } else {
        const int oy = qdiv(rem, p.Wout, r_w), ox = rem - oy * p.Wout;
        iy0 = oy * p.stride - p.pad;
        ix0 = ox * p.stride - p.pad;
      }
    }
    unsigned vy = 0, vx = 0;   // bit t: tap row / column t exists and lies inside the image
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      vy |= (t < p.KH && (unsigned)(iy0 + t) < (unsigned)p.Heff ? 1u : 0u) << t;
      vx |= (t < p.KW && (unsigned)(ix0 + t) < (unsigned)p.Weff ? 1u : 0u) << t;
    }
    const unsigned valid = ((vy & 1u) ? vx : 0u) | ((vy & 2u) ? vx << p.KW : 0u) | ((vy & 4u) ? vx << (2 * p.KW) : 0u);   // bit ty * KW + tx
    a_inv[i] = (int)~valid;
    a_pix[i] = (n_ + iy0) * p.Win + ix0;
  }
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.bytesw, 0x00020000);
  const int phase_t = p.subpix ? ((m0 % p.HWout) / p.hw_src) : 0;   // a tile lies inside one sub-pixel phase (host: hw_src % BM == 0)
  const unsigned kbytes = (unsigned)p.K * 4u;                       // one weight row: K elements x 2 pieces x 2 bytes
  const unsigned qv = (unsigned)(phase_t * p.Cout + n0 + 8 * wave + lrow) * kbytes + slot16;

  // load iterator (runs NST chunks ahead of the compute)
  int l_cc = cg_beg, l_ky = 0, l_kx = 0, l_it = 0, l_st = 0;
  unsigned pbase[GP], l_off[GP];
  unsigned l_cs4 = 0, l_so = 0;
  int l_lds = 0;
  __amdgpu_buffer_rsrc_t l_rs = rsw;

#define MFC2_CHUNK_SETUP()                                                                                              \
  {                                                                                                                     \
    const int c0_ = l_cc * 32;                                                                                          \
    const bool first_ = c0_ < p.C1;                                                                                     \
    l_cs4 = (unsigned)(first_ ? p.C1 : p.C2) * 4u;                                                                      \
    const unsigned cb4_ = (unsigned)(first_ ? c0_ : c0_ - p.C1) * 4u + slot16;                                          \
    l_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(first_ ? p.x1 : p.x2), 0, first_ ? p.bytes1 : p.bytes2, 0x00020000); \
    _Pragma("unroll") for (int i = 0; i < GP; ++i) pbase[i] = (unsigned)a_pix[i] * l_cs4 + cb4_;                        \
  }
// The DMA of the chunk the load iterator points at, in two parts so that nothing but the issue itself sits behind the barrier:
// PREP (in the first half of an iteration, under its MFMAs): gather offsets of the activation rows, scalar offset of the weight rows,
// LDS base of the target stage; ISSUE U (0 .. NL-1, second half): one buffer_load ... lds each.
#define MFC2_LOAD_PREP()                                                                                                \
  {                                                                                                                     \
    const unsigned ts_ = (unsigned)(l_ky * p.KW + l_kx);                                                                \
    const unsigned tapb_ = (unsigned)(l_ky * p.Win + l_kx) * l_cs4;                                                     \
    _Pragma("unroll") for (int i = 0; i < GP; ++i) l_off[i] = (pbase[i] + tapb_) | (unsigned)__builtin_amdgcn_sbfe(a_inv[i], ts_, 1u); \
    l_so = (ts_ * (unsigned)p.Cin + (unsigned)l_cc * 32u) * 4u;                                                         \
    l_lds = l_st * STAGE + wave * 1024;                                                                                 \
  }
#define MFC2_LOAD_UNIT(U)                                                                                               \
  {                                                                                                                     \
    constexpr int u_ = (U);                                                                                             \
    if constexpr (u_ < GP) {                                                                                            \
      const unsigned off_ = l_off[u_];   /* (a subscript written straight into the builtin's argument list makes the HOST pass drop the kernel stub without a diagnostic: hipcc 7.2) */ \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(l_rs, (__attribute__((address_space(3))) void*)(smem + l_lds + NW * u_ * 1024), 16, off_, 0, 0, 0); \
    } else {                                                                                                            \
      constexpr int q_ = u_ - GP;                                                                                       \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(smem + l_lds + BM * ROWB + NW * q_ * 1024), 16, qv, \
                                               l_so + (unsigned)q_ * (8u * NW) * kbytes, 0, 0);                         \
    }                                                                                                                   \
  }
#define MFC2_LOAD_ADVANCE()                  \
  {                                          \
    ++l_it;                                  \
    l_st = l_st == NST - 1 ? 0 : l_st + 1;   \
    ++l_kx;                                  \
    if (l_kx == p.KW) { l_kx = 0; ++l_ky; }  \
    if (l_ky == p.KH) {                      \
      l_ky = 0;                              \
      ++l_cc;                                \
      if (l_it < nit) MFC2_CHUNK_SETUP()     \
    }                                        \
  }
#define MFC2_LOAD_ALL()                                            \
  {                                                                \
    MFC2_LOAD_PREP()                                               \
    MFC2_LOAD_UNIT(0) MFC2_LOAD_UNIT(1)                            \
    if constexpr (NL > 2) MFC2_LOAD_UNIT(NL > 2 ? 2 : 0)           \
    if constexpr (NL > 3) MFC2_LOAD_UNIT(NL > 3 ? 3 : 0)           \
    if constexpr (NL > 4) MFC2_LOAD_UNIT(NL > 4 ? 4 : 0)           \
    if constexpr (NL > 5) MFC2_LOAD_UNIT(NL > 5 ? 5 : 0)           \
    if constexpr (NL > 6) MFC2_LOAD_UNIT(NL > 6 ? 6 : 0)           \
    if constexpr (NL > 7) MFC2_LOAD_UNIT(NL > 7 ? 7 : 0)           \
  }

  // ---- fragment addressing: lane reads tile row (lane & 31) (+ 32 per sub-tile), 16-byte slot (4 step + 2 (lane >> 5) + piece) ^ key
  const int fkey = (lane >> 1) & 7, fh = lane >> 5;
  int foff[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int c = 0; c < 2; ++c) foff[s][c] = (lane & 31) * ROWB + (((4 * s + 2 * fh + c) ^ fkey) * 16);
  const int xrow0 = wm * FM * ROWB;               // activation rows of this wave
  const int wrow0 = BM * ROWB + wn * FN * ROWB;   // weight rows of this wave

  f32x16 accm[TM][TN], accx[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { accm[i][j][r] = 0.f; accx[i][j][r] = 0.f; }

  f16x8 fx[2][TM][2], fw[2][TN][2];   // [step][sub-tile][piece]

  // per-pixel (= per-lane) operand scales: the accumulators hold sum(w 2^-wexp * x 2^-e), e = e1 or e2 by source.  The exponents are read
  // from the bound arrays HERE, ahead of the first DMA (their round trip hides behind it; loads return in order, so the counted vmcnt waits
  // of the pipeline are unaffected), and stay in 2 TM registers: fetched at the top of the epilogue they were a dependent ~1.5 us round trip
  // with nothing to hide behind (scripts/conv_timeline.py: drain 4.2 us of a 57 us launch).
  const bool first_src1 = cg_beg * 32 < p.C1, last_src2 = (cg_end - 1) * 32 >= p.C1;
  const int it_sw = __builtin_amdgcn_readfirstlane((first_src1 && last_src2) ? (p.C1 / 32 - cg_beg) * taps : -1);   // first iteration that reads the second source
#ifndef MFC2_EXPS_EARLY
#define MFC2_EXPS_EARLY 1   // 0: the round-2 form (bounds fetched where they are used), for A/B builds
#endif
#if MFC2_EXPS_EARLY
#define MFC2_PIXEL_EXPS_DECL()                                                                                          \
  float pb1[TM] = {}, pb2[TM] = {};   /* the raw bounds: converted where they are used, nothing waits for them up here */ \
  f32x4 eb0 = {0.f, 0.f, 0.f, 0.f}, eb1 = {0.f, 0.f, 0.f, 0.f};   /* the 8 bias values this lane adds in the epilogue */
// issued BEHIND the DMA of the prologue (ahead of it they delayed the first chunk by 0.5 us: loads return in order); the counted vmcnt
// waits then see 2 TM younger loads, i.e. at worst wait for that many loads of the next chunk as well
#define MFC2_PIXEL_EXPS_LOAD()                                                                                          \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                                      \
    const int pm_ = min(m0 + wm * FM + i * 32 + (lane & 31), p.M - 1);                                                  \
    const int pn_ = pm_ / p.HWout;                                                                                      \
    pb1[i] = p.bound1 ? p.bound1[pn_] : 0.f;                                                                            \
    pb2[i] = (p.bound2 && p.C2 > 0) ? p.bound2[pn_] : 0.f;                                                              \
  }                                                                                                                     \
  if ((p.splitk == 1 || p.tree) && p.bias) {                                                                            \
    const int bc0_ = n0 + wn * FN + (lane % (FN / 8)) * 8;                                                              \
    eb0 = *reinterpret_cast<const f32x4*>(p.bias + bc0_);                                                               \
    eb1 = *reinterpret_cast<const f32x4*>(p.bias + bc0_ + 4);                                                           \
  }
#define MFC2_PIXEL_EXPS(I) const int e1_ = p.bound1 ? scale_exp_of(pb1[I]) : 0, e2_ = (p.bound2 && p.C2 > 0) ? scale_exp_of(pb2[I]) : 0;
#else
#define MFC2_PIXEL_EXPS_DECL()
#define MFC2_PIXEL_EXPS_LOAD()
#define MFC2_PIXEL_EXPS(I)                                                                                              \
    const int pm_ = min(m0 + wm * FM + (I) * 32 + (lane & 31), p.M - 1);                                                \
    const int pn_ = pm_ / p.HWout;                                                                                      \
    const int e1_ = p.bound1 ? scale_exp_of(p.bound1[pn_]) : 0;                                                         \
    const int e2_ = (p.bound2 && p.C2 > 0) ? scale_exp_of(p.bound2[pn_]) : 0;
#endif
  MFC2_PIXEL_EXPS_DECL()
#define MFC2_SOURCE_SWITCH()                                                                                            \
  if (__builtin_expect(it == it_sw, 0)) {                                                                               \
    if constexpr (MFC2_HZ_ON(0)) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");                                     \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                                    \
      MFC2_PIXEL_EXPS(i)                                                                                                \
      const float f_ = exp2i(e1_) * exp2i(-e2_);                                                                        \
      _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                                    \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) { accm[i][j][r] *= f_; accx[i][j][r] *= f_; }                    \
    }                                                                                                                   \
  }

// fragment read U (0 .. NR-1) of step S from the stage at byte offset SB
#define MFC2_READ_UNIT(S, SB, U)                                                                                        \
  {                                                                                                                     \
    constexpr int u_ = (U);                                                                                             \
    if constexpr (u_ < PC * TM) {                                                                                       \
      fx[S][u_ / PC][u_ % PC] = *reinterpret_cast<const f16x8*>(smem + (SB) + xrow0 + (u_ / PC) * 32 * ROWB + foff[S][u_ % PC]); \
    } else {                                                                                                            \
      constexpr int v_ = u_ - PC * TM;                                                                                  \
      fw[S][v_ / PC][v_ % PC] = *reinterpret_cast<const f16x8*>(smem + (SB) + wrow0 + (v_ / PC) * 32 * ROWB + foff[S][v_ % PC]); \
    }                                                                                                                   \
  }
// MFMA n (0 .. NM-1) of step S: term t outer so that consecutive instructions hit different accumulators
#define MFC2_MFMA(S, N_)                                                                                                \
  {                                                                                                                     \
    constexpr int n_ = (N_);                                                                                            \
    constexpr int j_ = n_ % TN, i_ = (n_ / TN) % TM, t_ = n_ / (TN * TM);                                               \
    if constexpr (t_ == 0) accm[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[S][j_][0], fx[S][i_][0], accm[i_][j_], 0, 0, 0); \
    if constexpr (t_ == 1) accx[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[S][j_][0], fx[S][i_][PC - 1], accx[i_][j_], 0, 0, 0); \
    if constexpr (t_ == 2) accx[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[S][j_][PC - 1], fx[S][i_][0], accx[i_][j_], 0, 0, 0); \
  }

  if (nit > 0) {
    // ---- prologue: up to NST chunks in flight, wait for the first
    MFC2_CHUNK_SETUP()
    MFC2_LOAD_ALL()
    MFC2_LOAD_ADVANCE()
#pragma unroll
    for (int k = 1; k < NST; ++k)
      if (nit > k) { MFC2_LOAD_ALL() MFC2_LOAD_ADVANCE() }
    MFC2_PIXEL_EXPS_LOAD()
    {
      const int g = min(nit, NST) - 1;   // chunk groups that may stay in flight
      if (g >= 5) { MFC2_WAIT_VM(5 * NL <= 63 ? 5 * NL : 0); } else if (g == 4) { MFC2_WAIT_VM(4 * NL <= 63 ? 4 * NL : 0); }
      else if (g == 3) { MFC2_WAIT_VM(3 * NL); } else if (g == 2) { MFC2_WAIT_VM(2 * NL); } else if (g == 1) { MFC2_WAIT_VM(NL); } else { MFC2_WAIT_VM(0); }
    }
    __builtin_amdgcn_s_barrier();
    MFC2_STAMP(1)
    MFC2_READ_UNIT(0, 0, 0) MFC2_READ_UNIT(0, 0, 1)
    if constexpr (NR > 2) MFC2_READ_UNIT(0, 0, NR > 2 ? 2 : 0)
    if constexpr (NR > 3) MFC2_READ_UNIT(0, 0, NR > 3 ? 3 : 0)
    if constexpr (NR > 4) { MFC2_READ_UNIT(0, 0, NR > 4 ? 4 : 0) MFC2_READ_UNIT(0, 0, NR > 4 ? 5 : 0) }
    if constexpr (NR > 6) { MFC2_READ_UNIT(0, 0, NR > 6 ? 6 : 0) MFC2_READ_UNIT(0, 0, NR > 6 ? 7 : 0) }
  }

// slot N_ of a step: one MFMA, then the units whose turn it is.  READ units: NR of them, spread over the NM slots; LOAD units (second
// half of the chunk only): NL of them, spread over the slots after the reads started.
#define MFC2_SLOT_A(N_, SB)                                                                                             \
  {                                                                                                                     \
    MFC2_MFMA(0, N_)                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);   /* the matrix instruction first: everything else of the slot issues under it */  \
    constexpr int lo_ = (N_) < NF ? ((N_) * NR + NF - 1) / NF : NR, hi_ = (N_) < NF ? (((N_) + 1) * NR + NF - 1) / NF : NR; \
    if constexpr (lo_ < hi_ && lo_ < NR) MFC2_READ_UNIT(1, SB, lo_ < NR ? lo_ : 0)                                      \
    if constexpr (lo_ + 1 < hi_ && lo_ + 1 < NR) MFC2_READ_UNIT(1, SB, lo_ + 1 < NR ? lo_ + 1 : 0)                      \
    if constexpr (lo_ + 2 < hi_ && lo_ + 2 < NR) MFC2_READ_UNIT(1, SB, lo_ + 2 < NR ? lo_ + 2 : 0)                      \
    if constexpr ((N_) == NM - 1) MFC2_LOAD_PREP()   /* (addresses of the DMA the second half issues) */                 \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
  }
#define MFC2_SLOT_B(N_, SB, DO_LOAD)                                                                                    \
  {                                                                                                                     \
    MFC2_MFMA(1, N_)                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    constexpr int lo_ = (N_) < NF ? ((N_) * NR + NF - 1) / NF : NR, hi_ = (N_) < NF ? (((N_) + 1) * NR + NF - 1) / NF : NR; \
    if constexpr (lo_ < hi_ && lo_ < NR) MFC2_READ_UNIT(0, SB, lo_ < NR ? lo_ : 0)                                      \
    if constexpr (lo_ + 1 < hi_ && lo_ + 1 < NR) MFC2_READ_UNIT(0, SB, lo_ + 1 < NR ? lo_ + 1 : 0)                      \
    if constexpr (lo_ + 2 < hi_ && lo_ + 2 < NR) MFC2_READ_UNIT(0, SB, lo_ + 2 < NR ? lo_ + 2 : 0)                      \
    constexpr int ll_ = ((N_) * NL + NM - 1) / NM, lh_ = (((N_) + 1) * NL + NM - 1) / NM;                               \
    if (DO_LOAD) {                                                                                                      \
      if constexpr (ll_ < lh_ && ll_ < NL) MFC2_LOAD_UNIT(ll_ < NL ? ll_ : 0)                                           \
      if constexpr (ll_ + 1 < lh_ && ll_ + 1 < NL) MFC2_LOAD_UNIT(ll_ + 1 < NL ? ll_ + 1 : 0)                           \
      if constexpr (ll_ + 2 < lh_ && ll_ + 2 < NL) MFC2_LOAD_UNIT(ll_ + 2 < NL ? ll_ + 2 : 0)                           \
      if constexpr (ll_ + 3 < lh_ && ll_ + 3 < NL) MFC2_LOAD_UNIT(ll_ + 3 < NL ? ll_ + 3 : 0)                           \
      if constexpr (ll_ + 4 < lh_ && ll_ + 4 < NL) MFC2_LOAD_UNIT(ll_ + 4 < NL ? ll_ + 4 : 0)                           \
      if constexpr (ll_ + 5 < lh_ && ll_ + 5 < NL) MFC2_LOAD_UNIT(ll_ + 5 < NL ? ll_ + 5 : 0)                           \
      if constexpr (ll_ + 6 < lh_ && ll_ + 6 < NL) MFC2_LOAD_UNIT(ll_ + 6 < NL ? ll_ + 6 : 0)                           \
      if constexpr (ll_ + 7 < lh_ && ll_ + 7 < NL) MFC2_LOAD_UNIT(ll_ + 7 < NL ? ll_ + 7 : 0)                           \
    }                                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
  }
#define MFC2_REP12(M_, ...)                                                                                             \
  M_(0, __VA_ARGS__)                                                                                                    \
  if constexpr (NM > 1) { M_(NM > 1 ? 1 : 0, __VA_ARGS__) }                                                             \
  if constexpr (NM > 2) { M_(NM > 2 ? 2 : 0, __VA_ARGS__) }                                                             \
  if constexpr (NM > 3 && NM <= 4) { M_(NM > 3 ? 3 : 0, __VA_ARGS__) }                                                  \
  if constexpr (NM > 4) { M_(NM > 3 ? 3 : 0, __VA_ARGS__) M_(NM > 3 ? 4 : 0, __VA_ARGS__) M_(NM > 3 ? 5 : 0, __VA_ARGS__) } \
  if constexpr (NM > 6) { M_(NM > 6 ? 6 : 0, __VA_ARGS__) M_(NM > 6 ? 7 : 0, __VA_ARGS__) M_(NM > 6 ? 8 : 0, __VA_ARGS__)   \
                          M_(NM > 6 ? 9 : 0, __VA_ARGS__) M_(NM > 6 ? 10 : 0, __VA_ARGS__) M_(NM > 6 ? 11 : 0, __VA_ARGS__) }

  // ---- main loop.  Iteration `it` (LDS stage st = it % NST):
  //   first half : MFMAs of k-step 0 (fragments already in registers) | fragment reads of k-step 1 from stage st
  //   middle     : lgkmcnt(0) (stage st is fully read by this wave), vmcnt (chunk it+1 of this wave has landed), s_barrier
  //   second half: MFMAs of k-step 1 | fragment reads of k-step 0 of chunk it+1 | DMA of chunk it+NST into stage st
  int st = 0, it = 0;
  for (; it + NST < nit; ++it) {   // steady state
    MFC2_SOURCE_SWITCH()
    const int sb = st * STAGE;
    const int sn = (st == NST - 1 ? 0 : st + 1) * STAGE;
    MFC2_REP12(MFC2_SLOT_A, sb)
    MFC2_WAIT_LGKM0();
    MFC2_WAIT_VM((NST - 2) * NL);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    MFC2_REP12(MFC2_SLOT_B, sn, true)
    MFC2_LOAD_ADVANCE()
    st = st == NST - 1 ? 0 : st + 1;
  }
  for (; it < nit; ++it) {         // last NST chunks: nothing left to load; chunks it+1 .. nit-1 are still in flight
    MFC2_SOURCE_SWITCH()
    const int sb = st * STAGE;
    const int sn = (st == NST - 1 ? 0 : st + 1) * STAGE;
    MFC2_REP12(MFC2_SLOT_A, sb)
    MFC2_WAIT_LGKM0();
    {
      const int r = nit - 1 - it;    // groups in flight; chunk it+1 must have landed: r - 1 may stay
      if (r >= 5) { MFC2_WAIT_VM(4 * NL <= 63 ? 4 * NL : 0); } else if (r == 4) { MFC2_WAIT_VM(3 * NL); } else if (r == 3) { MFC2_WAIT_VM(2 * NL); }
      else if (r == 2) { MFC2_WAIT_VM(NL); } else { MFC2_WAIT_VM(0); }
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    MFC2_REP12(MFC2_SLOT_B, sn, false)
    st = st == NST - 1 ? 0 : st + 1;
  }

  MFC2_STAMP(2)
#define MFC2_EPILOGUE_LDS_BYTES (NST * STAGE)
#include "conv_f16x2_epilogue.inc"
#undef MFC2_EPILOGUE_LDS_BYTES
  MFC2_STAMP(3)
}

// fp32 [rows][per_row] -> fp16 pairs, 8 consecutive elements per thread; row r is scaled by 2^-scale_exp_of(bound[r]) (bound null: unscaled)
__global__ __launch_bounds__(256) void split_act_f16x2_kernel(const float* __restrict__ x, u32x4* __restrict__ out, long octets,
                                                              const float* __restrict__ bound, long octets_per_row) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < octets; o += stride) {
    const float sc = bound ? exp2i(-scale_exp_of(bound[o / octets_per_row])) : 1.f;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(x + o * 8) * sc, v1 = *reinterpret_cast<const f32x4*>(x + o * 8 + 4) * sc;
    u32x4 hi, lo;
    split8_f16(v0, v1, hi, lo);
    out[o * 2] = hi;
    out[o * 2 + 1] = lo;
  }
}

// The same split with the row's bound still lying as the [rows][nslots] slot maxima its producer left (the y_bound slots of a convolution,
// the partial array of mf_maxabs_rows_f32): every workgroup of a row reduces the slots itself -- no mf_bound_finalize_f32 launch in front
// of the split -- and workgroup 0 of the row publishes bound_out[row].  grid (blocks per row, rows)
__global__ __launch_bounds__(256) void split_act_slots_kernel(const float* __restrict__ x, u32x4* __restrict__ out, long octets_per_row,
                                                              const float* __restrict__ slots, int nslots, float* __restrict__ bound_out) {
  __shared__ float sm[4];
  const int row = blockIdx.y, tid = threadIdx.x;
  float m = 0.f;
  for (int i = tid; i < nslots; i += 256) m = fmaxf(m, slots[(long)row * nslots + i]);
  m = mf::wave_max(m);
  if ((tid & 63) == 0) sm[tid >> 6] = m;
  __syncthreads();
  const float b = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
  if (blockIdx.x == 0 && tid == 0) bound_out[row] = b;
  const float sc = exp2i(-scale_exp_of(b));
  const long base = (long)row * octets_per_row;
  for (long o = (long)blockIdx.x * 256 + tid; o < octets_per_row; o += (long)gridDim.x * 256) {
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(x + (base + o) * 8) * sc, v1 = *reinterpret_cast<const f32x4*>(x + (base + o) * 8 + 4) * sc;
    u32x4 hi, lo;
    split8_f16(v0, v1, hi, lo);
    out[(base + o) * 2] = hi;
    out[(base + o) * 2 + 1] = lo;
  }
}

}  // namespace mfc2
