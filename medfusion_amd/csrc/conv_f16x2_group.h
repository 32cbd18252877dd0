// conv_f16x2_group.h -- TWO independent fp16-pair convolutions in ONE launch (round 4; included by conv_f16x2.hip behind the two kernels).
//
// Why.  BasicResBlock (conv_blocks.py:194-240) runs conv_res (1x1) and the 3x3 convolution of its BasicBlock on the SAME input; neither reads
// the other's output.  As two launches on one stream they are serialised by the queue all the same: the 1x1 -- 8 to 64 chunks of K, 10-24 us
// of which ~10 are boundary, ramp and drain (scripts/conv_timeline.py) -- runs alone on the chip, then a kernel boundary, then the 3x3.  A second
// stream was measured slower (profiles/r04_conv_res_overlap.txt: the fork / join events cost more than they hide, and co-resident workgroups
// slow the 3x3's loop).  Here the grid is the 3x3's workgroups FOLLOWED by the 1x1's: the 3x3 fills the chip exactly as before (its
// workgroups come first in dispatch order), and as its workgroups retire the 1x1's take their CUs -- no boundary between the two, the 1x1's
// ramp under the 3x3's drain, x still in the L2s.  Each workgroup runs the unchanged body of its own convolution (conv_f16x2_body.inc /
// conv_f16x2_halo_body.inc as device functions) with its index inside that convolution, so every result is bit-identical to the two launches.
// Both bodies must use the same workgroup size; the launch takes the larger LDS size and the larger register count of the two.
#pragma once
#include "conv_f16x2_halo.h"

namespace mfc2 {

template <int BM_, int BN_, int WM_, int WN_, int NST_>
struct PlainTile {
  static constexpr int threads = WM_ * WN_ * 64;
  static constexpr size_t lds = (size_t)NST_ * (BM_ + BN_) * 128u;
  template <int TERMS> static __device__ __forceinline__ void run(const ConvP2& p, int bid) { conv_f16x2_body<BM_, BN_, WM_, WN_, NST_, TERMS>(p, bid); }
};
template <int BM_, int BN_, int WM_, int WN_, int HG_>
struct HaloTile {
  static constexpr int threads = WM_ * WN_ * 64;
  static constexpr size_t lds = (size_t)2 * HG_ * (WM_ * WN_) * 1024 + (size_t)3 * BN_ * 128;
  template <int TERMS> static __device__ __forceinline__ void run(const ConvP2& p, int bid) { conv_halo_body<BM_, BN_, WM_, WN_, HG_, TERMS>(p, bid); }
};

// workgroups [0, na): convolution A (pa); [na, na + nb): convolution B (pb)
template <class A, class B, int TERMS>
__global__ __launch_bounds__(A::threads, 2) void conv_group_kernel(const ConvP2 pa, const ConvP2 pb, const int na) {
  static_assert(A::threads == B::threads, "one workgroup size");
  const int bid = (int)blockIdx.x;
  if (bid < na) {
    A::template run<TERMS>(pa, bid);
  } else {
    B::template run<TERMS>(pb, bid - na);
  }
}

}  // namespace mfc2
