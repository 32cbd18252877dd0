// conv_f16x2_halo.h -- the 3x3 stride-1 pad-1 convolution of the fp16-pair arithmetic (MF_CONV_FP32_F16X2 / MF_CONV_F16) with the ACTIVATIONS
// staged ONCE per 32-channel chunk.  Included by conv_f16x2.hip behind conv_f16x2.h (same namespace, same ConvP2, same epilogue).
//
// Why.  conv_f16x2_kernel moves, per 32-channel chunk of one filter tap, BM activation rows and BN weight rows of 128 bytes from L2 to LDS:
// (BM + BN) x 128 B per BM x BN x 32 MACs.  With the matrix work cut to a third (MF_CONV_F16) the same launches are only 1.4x faster
// (profiles/r03_single_term.txt): the loop is bound by what it moves through the vector-memory path (~10-15 TB/s of L2 -> LDS traffic over
// the chip, one 1-KB LDS-DMA instruction costing the issuing wave 60-180 cycles), not by the matrix pipe.  Nine tenths of the activation
// traffic is redundant: the nine taps of a chunk read the same pixels shifted by one.  Here a workgroup's tile is a block of whole image
// rows (BM = R x W pixels, or BM / (H W) whole small images); per chunk it loads the (R + 2) x (W + 2) pixel HALO of that block once
// (zero padding = out-of-range DMA offsets, as before) and the nine taps read their A fragments from it at nine offsets:
//     bytes per chunk and tap:  BN x 128 (weights)  +  (R + 2)(W + 2) x 128 / 9 (halo)     e.g. 256 x 128 tile at 32 x 32: 20.9 KB
// against 48 KB for the same tile (32 KB for 128 x 128, half the MACs) in conv_f16x2_kernel -- and as many fewer LDS-DMA instructions.
//
// LDS: [halo 0][halo 1][weight stage 0..2].  The halo of chunk c+1 is fetched while chunk c is multiplied (one 1-KB piece per tap per wave,
// HG pieces), the weights run through a 3-stage ring like conv_f16x2_kernel's (the weights of tap t + 3 go into the stage tap t has just read).  The nine taps of a chunk are unrolled, so every vmcnt is a
// compile-time count; the last chunk still issues its (out-of-range, zero-filling) prefetches so that every iteration is the same.
// Fragment reads: lane p of a 32-pixel block reads halo pixel hp = hp0(p) + ky (W + 2) + kx, 16-byte slot (4 step + 2 half + piece) ^
// ((hp >> 1) & 7) -- the slot permutation of conv_f16x2_kernel keyed on the halo pixel instead of the tile row.
// Everything behind the loop -- in-launch split-K tree, scaling, staging, bias, bound slots, GroupNorm records -- is the shared epilogue.
#pragma once
#include "conv_f16x2.h"

namespace mfc2 {

// HG: 1-KB halo pieces per wave and chunk (8 halo pixels each): HG x 8 x NW >= halo pixels of the tile
template <int BM, int BN, int WM, int WN, int HG, int TERMS = 3>
__global__ __launch_bounds__(WM * WN * 64, 2) void conv_halo_kernel(const ConvP2 p) {
  static_assert(TERMS == 3 || TERMS == 1, "three product terms (fp16 pairs) or one (fp16)");
  constexpr int PC = TERMS == 3 ? 2 : 1;
  constexpr int NW = WM * WN;
  static_assert(NW == 8 || NW == 4, "4 or 8 waves");
  constexpr int FM = BM / WM, FN = BN / WN, TM = FM / 32, TN = FN / 32;
  static_assert(TM >= 1 && TN >= 1 && FM % 32 == 0 && FN % 32 == 0, "per-wave footprint");
  constexpr int GQ = BN / (8 * NW);                          // weight pieces per wave and iteration
  static_assert(BN % (8 * NW) == 0 && GQ >= 1 && HG >= 1 && HG <= 7, "pieces");
  constexpr int ROWB = 128;
  constexpr int HBYTES = HG * NW * 1024, WST = BN * ROWB, NST = 3;
  constexpr int WBASE = 2 * HBYTES;                          // first weight stage
  static_assert(WBASE + NST * WST <= 160 * 1024, "LDS");
  constexpr int NM = TERMS * TM * TN;

  extern __shared__ __attribute__((aligned(1024))) char smem[];
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
  const int nchunks = cg_end - cg_beg;

  // ---- geometry of the tile: R whole rows of one image, or BM / (H W) whole small images
  const int W = p.Win, H = p.Hin, W2 = W + 2, HW = H * W;
  const bool multi = HW < BM;
  const int R = multi ? H : BM / W;
  const int HS = (R + 2) * W2;                                // halo pixels per image segment
  const int segs = multi ? BM / HW : 1;
  const int HPX = segs * HS;
  const int hn_first = m0 / HW;
  const int hy_first = multi ? 0 : (m0 - hn_first * HW) / W;

  // ---- halo DMA: piece i of this wave = halo pixels 8 (wave + NW i) + (lane >> 3); LDS slot (lane & 7) holds source slot (lane & 7) ^ key
  const int lrow = lane >> 3;
  const unsigned slot16 = (unsigned)(((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7)) * 16);
  int h_gp[HG];   // global pixel index of the halo pixel, or -1 (outside the image / past the halo)
#pragma unroll
  for (int i = 0; i < HG; ++i) {
    const int hp = 8 * (wave + NW * i) + lrow;
    int gp = -1;
    if (hp < HPX) {
      const int seg = hp / HS, r2 = hp - seg * HS;
      const int ry = r2 / W2, rx = r2 - ry * W2;
      const int y = hy_first + ry - 1, x = rx - 1, n = hn_first + seg;
      if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W && n < p.N) gp = (n * H + y) * W + x;
    }
    h_gp[i] = gp;
  }
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.bytesw, 0x00020000);
  const unsigned kbytes = (unsigned)p.K * 4u;
  const unsigned qv = (unsigned)(n0 + 8 * wave + lrow) * kbytes + slot16;    // weight row of piece 0 (piece q: + q * 8 NW rows)

  // ---- A fragments: lane reads halo pixel hp0 + tap offset of its pixel; B fragments: weight row (lane & 31) of its 32-row blocks
  const int fh = lane >> 5;
  int xhp[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int pt = wm * FM + i * 32 + (lane & 31);
    const int seg = pt / (R * W), rem = pt - seg * (R * W);
    const int py = rem / W, px = rem - py * W;
    xhp[i] = seg * HS + py * W2 + px;
  }
  const int fkey = (lane >> 1) & 7;
  int wfoff[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int c = 0; c < 2; ++c) wfoff[s][c] = (lane & 31) * ROWB + (((4 * s + 2 * fh + c) ^ fkey) * 16);
  const int wrow0 = wn * FN * ROWB;

  f32x16 accm[TM][TN], accx[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { accm[i][j][r] = 0.f; accx[i][j][r] = 0.f; }
  f16x8 fx[2][TM][2], fw[2][TN][2];

  const bool first_src1 = cg_beg * 32 < p.C1, last_src2 = (cg_end - 1) * 32 >= p.C1;
  const int c_sw = __builtin_amdgcn_readfirstlane((first_src1 && last_src2) ? p.C1 / 32 : -1);   // first chunk that reads the second source

  // ---- loaders.  Halo of chunk CC into buffer HB (piece U of this wave); weights of (chunk CC, tap T) into ring stage WS.  A chunk past
  // the end of this K slice loads from an out-of-range offset (the DMA fills zeros): every iteration issues the same instructions.
#define MFH_HALO_UNIT(U, CC, HB)                                                                                          \
  {                                                                                                                       \
    constexpr int u_ = (U);                                                                                               \
    const int c0_ = (CC) * 32;                                                                                            \
    const bool ok_ = (CC) < cg_end;                                                                                       \
    const bool s1_ = c0_ < p.C1;                                                                                          \
    const unsigned cs4_ = (unsigned)(s1_ ? p.C1 : p.C2) * 4u;                                                             \
    const unsigned cb4_ = (unsigned)(s1_ ? c0_ : c0_ - p.C1) * 4u + slot16;                                               \
    const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(s1_ ? p.x1 : p.x2), 0, s1_ ? p.bytes1 : p.bytes2, 0x00020000); \
    const unsigned off_ = (ok_ && h_gp[u_] >= 0) ? (unsigned)h_gp[u_] * cs4_ + cb4_ : 0x80000000u;                         \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (__attribute__((address_space(3))) void*)(smem + (HB) * HBYTES + (wave + NW * u_) * 1024), 16, off_, 0, 0, 0); \
  }
#define MFH_W_UNIT(Q, CC, T, WS)                                                                                          \
  {                                                                                                                       \
    constexpr int q_ = (Q);                                                                                               \
    const bool ok_ = (CC) < cg_end;                                                                                       \
    const unsigned vo_ = ok_ ? qv : 0x80000000u;   /* (past the end of the K slice: out of range through the VECTOR offset) */ \
    const unsigned so_ = ok_ ? ((unsigned)(T) * (unsigned)p.Cin + (unsigned)(CC) * 32u) * 4u + (unsigned)q_ * (8u * NW) * kbytes : 0u; \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(smem + WBASE + (WS) * WST + (wave + NW * q_) * 1024), 16, vo_, so_, 0, 0); \
  }
#define MFH_W_ALL(CC, T, WS)                                        \
  {                                                                 \
    MFH_W_UNIT(0, CC, T, WS)                                        \
    if constexpr (GQ > 1) MFH_W_UNIT(GQ > 1 ? 1 : 0, CC, T, WS)     \
    if constexpr (GQ > 2) MFH_W_UNIT(GQ > 2 ? 2 : 0, CC, T, WS)     \
    if constexpr (GQ > 3) MFH_W_UNIT(GQ > 3 ? 3 : 0, CC, T, WS)     \
  }

  // fragment read U (0 .. NR-1) of k-step S: the X pieces of this wave's TM blocks from halo buffer HB at tap offset TOFF (a halo-pixel
  // count), then the W pieces of its TN blocks from ring stage WS
#define MFH_READ_UNIT(S, U, HB, TOFF, WS)                                                                                 \
  {                                                                                                                       \
    constexpr int u_ = (U);                                                                                               \
    if constexpr (u_ < PC * TM) {                                                                                         \
      const int hp_ = xhp[u_ / PC] + (TOFF);                                                                              \
      fx[S][u_ / PC][u_ % PC] = *reinterpret_cast<const f16x8*>(smem + (HB) * HBYTES + hp_ * ROWB + (((4 * (S) + 2 * fh + (u_ % PC)) ^ ((hp_ >> 1) & 7)) << 4)); \
    } else {                                                                                                              \
      constexpr int v_ = u_ - PC * TM;                                                                                    \
      fw[S][v_ / PC][v_ % PC] = *reinterpret_cast<const f16x8*>(smem + WBASE + (WS) * WST + wrow0 + (v_ / PC) * 32 * ROWB + wfoff[S][v_ % PC]); \
    }                                                                                                                     \
  }
#define MFH_MFMA(S, N_)                                                                                                   \
  {                                                                                                                       \
    constexpr int n_ = (N_);                                                                                              \
    constexpr int j_ = n_ % TN, i_ = (n_ / TN) % TM, t_m = n_ / (TN * TM);                                                \
    if constexpr (t_m == 0) accm[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[S][j_][0], fx[S][i_][0], accm[i_][j_], 0, 0, 0); \
    if constexpr (t_m == 1) accx[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[S][j_][0], fx[S][i_][PC - 1], accx[i_][j_], 0, 0, 0); \
    if constexpr (t_m == 2) accx[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[S][j_][PC - 1], fx[S][i_][0], accx[i_][j_], 0, 0, 0); \
  }
  constexpr int NR = PC * (TM + TN);                         // fragment reads per k-step
  constexpr int NF = NM >= 3 ? (2 * NM + 2) / 3 : NM;        // ... issued behind the first NF matrix instructions of the other k-step
  static_assert((NR + NF - 1) / NF <= 3, "at most three reads per slot");
  // slot N_ of the first half of tap T: MFMA N_ of k-step 0, then its share of the k-step-1 reads
#define MFH_SLOT_A(N_, T)                                                                                                 \
  {                                                                                                                       \
    MFH_MFMA(0, N_)                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    constexpr int lo_ = (N_) < NF ? ((N_) * NR + NF - 1) / NF : NR, hi_ = (N_) < NF ? (((N_) + 1) * NR + NF - 1) / NF : NR; \
    if constexpr (lo_ < hi_ && lo_ < NR) MFH_READ_UNIT(1, lo_ < NR ? lo_ : 0, hb, toff_, ws)                              \
    if constexpr (lo_ + 1 < hi_ && lo_ + 1 < NR) MFH_READ_UNIT(1, lo_ + 1 < NR ? lo_ + 1 : 0, hb, toff_, ws)              \
    if constexpr (lo_ + 2 < hi_ && lo_ + 2 < NR) MFH_READ_UNIT(1, lo_ + 2 < NR ? lo_ + 2 : 0, hb, toff_, ws)              \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
  }
  // slot N_ of the second half: MFMA N_ of k-step 1, its share of the NEXT tap's k-step-0 reads, its share of the DMA issue
  // (load unit 0 = halo piece T of the next chunk while T < HG, then the GQ weight pieces of the tap three ahead)
#define MFH_LOAD_UNIT(U, T)                                                                                               \
  {                                                                                                                       \
    constexpr int lu_ = (U), th_ = (T) < HG ? 1 : 0;                                                                      \
    if constexpr (th_ && lu_ == 0) MFH_HALO_UNIT((T) < HG ? (T) : 0, cc + 1, hb ^ 1)                                      \
    else MFH_W_UNIT((lu_ - th_) >= 0 && (lu_ - th_) < GQ ? (lu_ - th_) : 0, cc + ((T) + 3 >= 9 ? 1 : 0), ((T) + 3) % 9, ws) \
  }
#define MFH_SLOT_B(N_, T)                                                                                                 \
  {                                                                                                                       \
    MFH_MFMA(1, N_)                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    constexpr int lo_ = (N_) < NF ? ((N_) * NR + NF - 1) / NF : NR, hi_ = (N_) < NF ? (((N_) + 1) * NR + NF - 1) / NF : NR; \
    if constexpr (lo_ < hi_ && lo_ < NR) MFH_READ_UNIT(0, lo_ < NR ? lo_ : 0, hbn_, toffn_, wsn_)                         \
    if constexpr (lo_ + 1 < hi_ && lo_ + 1 < NR) MFH_READ_UNIT(0, lo_ + 1 < NR ? lo_ + 1 : 0, hbn_, toffn_, wsn_)         \
    if constexpr (lo_ + 2 < hi_ && lo_ + 2 < NR) MFH_READ_UNIT(0, lo_ + 2 < NR ? lo_ + 2 : 0, hbn_, toffn_, wsn_)         \
    constexpr int nld_ = GQ + ((T) < HG ? 1 : 0);                                                                         \
    constexpr int ll_ = ((N_) * nld_ + NM - 1) / NM, lh_ = (((N_) + 1) * nld_ + NM - 1) / NM;                              \
    if constexpr (ll_ < lh_ && ll_ < nld_) MFH_LOAD_UNIT(ll_ < nld_ ? ll_ : 0, T)                                         \
    if constexpr (ll_ + 1 < lh_ && ll_ + 1 < nld_) MFH_LOAD_UNIT(ll_ + 1 < nld_ ? ll_ + 1 : 0, T)                         \
    if constexpr (ll_ + 2 < lh_ && ll_ + 2 < nld_) MFH_LOAD_UNIT(ll_ + 2 < nld_ ? ll_ + 2 : 0, T)                         \
    if constexpr (ll_ + 3 < lh_ && ll_ + 3 < nld_) MFH_LOAD_UNIT(ll_ + 3 < nld_ ? ll_ + 3 : 0, T)                         \
    if constexpr (ll_ + 4 < lh_ && ll_ + 4 < nld_) MFH_LOAD_UNIT(ll_ + 4 < nld_ ? ll_ + 4 : 0, T)                         \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
  }
#define MFH_REP(M_, T)                                                                                                    \
  M_(0, T)                                                                                                                \
  if constexpr (NM > 1) { M_(NM > 1 ? 1 : 0, T) }                                                                         \
  if constexpr (NM > 2) { M_(NM > 2 ? 2 : 0, T) }                                                                         \
  if constexpr (NM > 3) { M_(NM > 3 ? 3 : 0, T) }                                                                         \
  if constexpr (NM > 4) { M_(NM > 4 ? 4 : 0, T) M_(NM > 4 ? 5 : 0, T) }                                                   \
  if constexpr (NM > 6) { M_(NM > 6 ? 6 : 0, T) M_(NM > 6 ? 7 : 0, T) M_(NM > 6 ? 8 : 0, T) M_(NM > 6 ? 9 : 0, T) M_(NM > 6 ? 10 : 0, T) M_(NM > 6 ? 11 : 0, T) }

  // one tap of a chunk, T compile-time (0 .. 8); cc = chunk, hb = its halo buffer, ws = ring stage of (cc, T)
  //   first half : MFMAs of k-step 0 | fragment reads of k-step 1
  //   middle     : lgkmcnt(0), vmcnt(what the previous tap issued may stay in flight), barrier
  //   second half: MFMAs of k-step 1 | k-step-0 reads of the NEXT tap | DMA: halo piece T of the next chunk, weights three taps ahead (into
  //                the stage this tap has just finished reading)
#define MFH_TAP(T)                                                                                                        \
  {                                                                                                                       \
    constexpr int t_ = (T);                                                                                               \
    constexpr int ky_ = t_ / 3, kx_ = t_ % 3, tn_ = (t_ + 1) % 9, kyn_ = tn_ / 3, kxn_ = tn_ % 3;                          \
    const int toff_ = ky_ * W2 + kx_, toffn_ = kyn_ * W2 + kxn_;                                                          \
    const int wsn_ = ws == 2 ? 0 : ws + 1;                                                                                \
    const int hbn_ = t_ == 8 ? (hb ^ 1) : hb;                                                                             \
    /* the fragment addresses of a tap are formed inside the tap: without this the compiler hoists those of all nine taps out of the */ \
    /* chunk loop (they are loop-invariant) and spills the accumulators to make room */                                    \
    _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) asm volatile("" : "+v"(xhp[i_]));                                    \
    MFH_REP(MFH_SLOT_A, T)                                                                                                \
    MFC2_WAIT_LGKM0();                                                                                                    \
    MFC2_WAIT_VM(GQ + ((t_ >= 1 && t_ - 1 < HG) ? 1 : 0));                                                                \
    __builtin_amdgcn_s_barrier();                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    MFH_REP(MFH_SLOT_B, T)                                                                                                \
    ws = wsn_;                                                                                                            \
  }

  MFC2_PIXEL_EXPS_DECL()   // (operand scale exponents of this lane's pixels: conv_f16x2.h)
  if (nchunks > 0) {
    // ---- prologue: halo of the first chunk, weights of its first two taps; wait, barrier, first fragments
    int cc = cg_beg;
#pragma unroll
    for (int u = 0; u < HG; ++u) {
      const int c0_ = cc * 32;
      const bool s1_ = c0_ < p.C1;
      const unsigned cs4_ = (unsigned)(s1_ ? p.C1 : p.C2) * 4u;
      const unsigned cb4_ = (unsigned)(s1_ ? c0_ : c0_ - p.C1) * 4u + slot16;
      const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(s1_ ? p.x1 : p.x2), 0, s1_ ? p.bytes1 : p.bytes2, 0x00020000);
      const unsigned off_ = h_gp[u] >= 0 ? (unsigned)h_gp[u] * cs4_ + cb4_ : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (__attribute__((address_space(3))) void*)(smem + (wave + NW * u) * 1024), 16, off_, 0, 0, 0);
    }
    MFH_W_ALL(cc, 0, 0)
    MFH_W_ALL(cc, 1, 1)
    MFH_W_ALL(cc, 2, 2)
    MFC2_PIXEL_EXPS_LOAD()
    MFC2_WAIT_VM(2 * GQ);             // the halo and the weights of tap 0 have landed (taps 1 and 2 may still fly)
    __builtin_amdgcn_s_barrier();
    int hb = 0, ws = 0;
    {
      const int toff0_ = 0;
      MFH_READ_UNIT(0, 0, 0, toff0_, 0)
      if constexpr (NR > 1) MFH_READ_UNIT(0, NR > 1 ? 1 : 0, 0, toff0_, 0)
      if constexpr (NR > 2) MFH_READ_UNIT(0, NR > 2 ? 2 : 0, 0, toff0_, 0)
      if constexpr (NR > 3) MFH_READ_UNIT(0, NR > 3 ? 3 : 0, 0, toff0_, 0)
      if constexpr (NR > 4) MFH_READ_UNIT(0, NR > 4 ? 4 : 0, 0, toff0_, 0)
      if constexpr (NR > 5) MFH_READ_UNIT(0, NR > 5 ? 5 : 0, 0, toff0_, 0)
      if constexpr (NR > 6) MFH_READ_UNIT(0, NR > 6 ? 6 : 0, 0, toff0_, 0)
      if constexpr (NR > 7) MFH_READ_UNIT(0, NR > 7 ? 7 : 0, 0, toff0_, 0)
    }

    // ---- main loop: chunks, nine unrolled taps each
    for (; cc < cg_end; ++cc) {
      if (__builtin_expect(cc == c_sw, 0)) {   // the K loop passes from the first source of a fused concat to the second: re-scale
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          MFC2_PIXEL_EXPS(i)
          const float f_ = exp2i(e1_) * exp2i(-e2_);
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { accm[i][j][r] *= f_; accx[i][j][r] *= f_; }
        }
      }
      MFH_TAP(0) MFH_TAP(1) MFH_TAP(2) MFH_TAP(3) MFH_TAP(4) MFH_TAP(5) MFH_TAP(6) MFH_TAP(7) MFH_TAP(8)
      hb ^= 1;
    }
    MFC2_WAIT_VM(0);   // (the zero-filling prefetches of the last taps)
  }

#define MFC2_EPILOGUE_LDS_BYTES (WBASE + NST * WST)
#include "conv_f16x2_epilogue.inc"
#undef MFC2_EPILOGUE_LDS_BYTES
}

}  // namespace mfc2
