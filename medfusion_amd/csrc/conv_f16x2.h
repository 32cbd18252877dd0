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
  int out_nt;            // 1: non-temporal output stores (the component GEMM's output is 4x an activation, read once by the tail)
  int wphase_rows;       // > 0: rows [k wphase_rows, (k + 1) wphase_rows) of the GEMM use weight slab k (the component GEMMs of the Winograd form, winograd.h)
  int walk_n_fast;       // 1: the tile walk runs over the output-channel tiles first (an XCD covers all Cout blocks of a pixel range: activations fetched once)
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
#define MFC2_STAMP(I) { if (p.dbg && threadIdx.x == 0) reinterpret_cast<unsigned long*>(p.dbg)[(long)MFC2_BID * 8 + (I)] = __builtin_amdgcn_s_memrealtime(); }
#else
#define MFC2_STAMP(I)
#endif
#define MFC2_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// TERMS = 3: the fp32-class arithmetic above.  TERMS = 1 (MF_CONV_F16, opt-in REDUCED precision): only main += wh * xh -- the operands
// rounded to fp16 (11 significant bits), one matrix instruction per product, fp32 accumulate; same operand images, same DMA, the lo'
// pieces are simply never read from LDS.
#define MFC2_BODY_AS_KERNEL 1
#define MFC2_BID blockIdx.x
#include "conv_f16x2_body.inc"
#undef MFC2_BODY_AS_KERNEL
#undef MFC2_BID
#define MFC2_BODY_AS_KERNEL 0
#define MFC2_BID bid
#include "conv_f16x2_body.inc"
#undef MFC2_BODY_AS_KERNEL
#undef MFC2_BID

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
