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
#define MFC2_BODY_AS_KERNEL 1
#define MFC2_BID blockIdx.x
#include "conv_f16x2_halo_body.inc"
#undef MFC2_BODY_AS_KERNEL
#undef MFC2_BID
#define MFC2_BODY_AS_KERNEL 0
#define MFC2_BID bid
#include "conv_f16x2_halo_body.inc"
#undef MFC2_BODY_AS_KERNEL
#undef MFC2_BID

}  // namespace mfc2
