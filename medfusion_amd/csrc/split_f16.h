// split_f16.h -- the fp16-pair form of an fp32 tensor (MF_CONV_FP32_F16X2 operands), shared by the kernels that produce it.
//   x ~ hi + lo' / 2048,   hi = RN16(x),   lo' = RN16((x - hi) * 2048)      (the subtraction and the scaling are exact)
// 23 of the 24 significand bits survive: |x - hi - lo'/2048| <= 2^-23 |x| (one ulp of the fp32 value at most; exact for 3 values of 4).
// Storage: groups of 8 consecutive channels as 32 bytes [hi x 8][lo' x 8] -- 4 bytes per element, like fp32.
// Range: fp16 reaches 65504, activations do not stop there (an un-normalised residual stream follows the magnitude of the network input),
// so every split tensor carries a per-sample power-of-two scale: with `bound[n]` an upper bound of |x| over sample n (measured, or
// derived from the producing layer), s = floor(log2(bound)) - 14 and the pair stores x * 2^-s, i.e. |x 2^-s| < 2^15.  Scaling by a
// power of two is exact; values more than 28 binades below the bound keep an absolute accuracy of 2^-50 * bound.  The consuming
// convolution multiplies its fp32 accumulators by 2^s (per output pixel = per sample), again exactly.
#pragma once
#include <hip/hip_runtime.h>

namespace mf {

typedef float sf_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int sf_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int sf_u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 sf_f16x2 __attribute__((ext_vector_type(2)));

constexpr float kLoScale = 2048.f, kLoInv = 1.f / 2048.f, kF16Max = 65504.f;

// exponent s of the per-sample scale 2^-s for an upper bound of |x| (0, denormal, inf and NaN bounds are clamped to +-100)
__device__ __forceinline__ int scale_exp_of(float bound) {
  const int e = (int)((__float_as_uint(bound) >> 23) & 0xffu) - 127;
  const int s = e - 14;
  return s < -100 ? -100 : (s > 100 ? 100 : s);
}
__device__ __forceinline__ float exp2i(int s) { return __uint_as_float((unsigned)(127 + s) << 23); }  // 2^s, -126 <= s <= 127

// A finite value that exceeds its bound (it never does when the bound is one) saturates at the fp16 range; NaN and +-Inf are NOT clamped:
// they go through the conversions as NaN / Inf (hi = NaN or Inf, lo' = NaN), the matrix cores carry them on, and the convolution's output
// is non-finite like the fp32 reference's would be -- a clamp (fminf / fmaxf return the other operand for NaN) would turn them into a
// finite, silently wrong number.
__device__ __forceinline__ float clamp_f16_range(float a) {
  const float c = __builtin_fminf(__builtin_fmaxf(a, -kF16Max), kF16Max);
  return __builtin_fabsf(a) <= 3.4028234663852886e38f ? c : a;   // (false for NaN and Inf)
}
// CLAMP = false: for producers whose bound is DERIVED, not handed in (the GroupNorm-apply pass and the fused tail of the convolution:
// |value 2^-s| < 2^15 by construction, so the clamp never acts -- same bits, four VALU instructions per element fewer in VALU-bound code)
template <bool CLAMP = true>
__device__ __forceinline__ void split2_f16(float a, float b, unsigned& hi, unsigned& lo) {
  if (CLAMP) {
    a = clamp_f16_range(a);
    b = clamp_f16_range(b);
  }
  const _Float16 ha = (_Float16)a, hb = (_Float16)b;                             // round to nearest even
  const float ra = (a - (float)ha) * kLoScale, rb = (b - (float)hb) * kLoScale;  // both operations exact
  const sf_f16x2 h = {ha, hb}, l = {(_Float16)ra, (_Float16)rb};
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}
// 8 consecutive channels -> the 32-byte group [hi x 8][lo' x 8]
template <bool CLAMP = true>
__device__ __forceinline__ void split8_f16(const sf_f32x4 v0, const sf_f32x4 v1, sf_u32x4& hi, sf_u32x4& lo) {
  const float a0 = v0[0], a1 = v0[1], a2 = v0[2], a3 = v0[3], b0 = v1[0], b1 = v1[1], b2 = v1[2], b3 = v1[3];
  unsigned h0, h1, h2, h3, l0, l1, l2, l3;
  split2_f16<CLAMP>(a0, a1, h0, l0);
  split2_f16<CLAMP>(a2, a3, h1, l1);
  split2_f16<CLAMP>(b0, b1, h2, l2);
  split2_f16<CLAMP>(b2, b3, h3, l3);
  hi = sf_u32x4{h0, h1, h2, h3};
  lo = sf_u32x4{l0, l1, l2, l3};
}
// 4 consecutive channels starting at element index e (a multiple of 4) of a tensor whose innermost extent is a multiple of 8;
// sc = 2^-s of the sample the element belongs to.  PRECONDITION: lanes 2k and 2k+1 of the wave hold the two halves of the same 8-channel
// group and both execute the call (thread index == float4 index modulo an even stride, even float4 count): the pair swaps one 8-byte
// piece through DPP so that each lane stores ONE full 16-byte slot ([hi x 8] by the even lane, [lo' x 8] by the odd one) -- a wave
// writes 1 KB contiguously with one instruction instead of two half-filled ones.
// NT: the 16-byte store as a non-temporal one (streamed past the L2's dirty set: for outputs that are 4x an activation and read once, by
// another kernel -- the transform-domain tensors of the Winograd form)
template <bool CLAMP = true, bool NT = false>
__device__ __forceinline__ void store_split4(void* ys, long e, float a, float b, float c, float d, float sc) {
  unsigned h0, h1, l0, l1;
  split2_f16<CLAMP>(a * sc, b * sc, h0, l0);
  split2_f16<CLAMP>(c * sc, d * sc, h1, l1);
  const bool odd = (e >> 2) & 1;
  const unsigned s0 = odd ? h0 : l0, s1 = odd ? h1 : l1;                          // what the partner stores
  const unsigned r0 = (unsigned)__builtin_amdgcn_mov_dpp((int)s0, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]: lane ^ 1
  const unsigned r1 = (unsigned)__builtin_amdgcn_mov_dpp((int)s1, 0xB1, 0xF, 0xF, true);
  const sf_u32x4 v = odd ? sf_u32x4{r0, r1, l0, l1} : sf_u32x4{h0, h1, r0, r1};
#if defined(MF_GN_OUT_SC1) && MF_GN_OUT_SC1   /* experiment (round 4): write-through (sc1) stores, see conv_f16x2_epilogue.inc MFC2_OUT_STORE == 2 */
  sf_u32x4* q_ = reinterpret_cast<sf_u32x4*>(ys) + ((e >> 3) * 2 + (odd ? 1 : 0));
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(q_), "v"(v) : "memory");
#else
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<sf_u32x4*>(ys) + ((e >> 3) * 2 + (odd ? 1 : 0)));
  else reinterpret_cast<sf_u32x4*>(ys)[(e >> 3) * 2 + (odd ? 1 : 0)] = v;
#endif
}

// Measured bounds: thousands of waves updating the same few words with atomics serialise on one L2 channel (measured: 66 us for a 10 us
// pass), so every writer stores the max of its own share into its own slot of partial[n][slots] (plain stores, no zero fill needed:
// every slot of the launch is written) and a tiny pass reduces the slots to bound[n] (mf_bound_finalize_f32).
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

}  // namespace mf
