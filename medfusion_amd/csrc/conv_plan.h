// conv_plan.h -- what the two convolution translation units (conv.hip: fp32 / bf16-triplet kernels; conv_f16x2.hip: fp16-pair kernel) share:
// output geometry, the split-K reducer, and the cross-unit planner hooks.
#pragma once
#include "common.h"
#include "split_f16.h"
#include "gn_partial.h"

namespace mf {

struct TileCfg { int id, BM, BN, WM, WN, BK; };

struct Plan {
  bool igemm;
  TileCfg cfg;
  int splitk, nk_per_split;
  int Hout, Wout, Heff, Weff, M, K;
};


inline int fill_geometry(const MfConvDesc* d, Plan* pl) {
  MF_REQUIRE(d != nullptr, MF_EINVAL, "conv: null desc");
  MF_REQUIRE(d->N > 0 && d->Hin > 0 && d->Win > 0 && d->C1 > 0 && d->C2 >= 0 && d->Cout > 0, MF_EINVAL, "conv: bad dims");
  MF_REQUIRE((d->KH == 1 && d->KW == 1) || (d->KH == 3 && d->KW == 3), MF_EUNSUPPORTED, "conv: kernel %dx%d unsupported", d->KH, d->KW);
  MF_REQUIRE(d->stride == 1 || d->stride == 2, MF_EUNSUPPORTED, "conv: stride %d unsupported", d->stride);
  // upsample == 3: the 16 component GEMMs of a Winograd convolution -- N = 16 n pseudo-samples of Hin x Win = 1 x T tile rows, 1x1, NHWC, weights = 16
  // slabs [16][Cout][Cin], rows [k n T, (k + 1) n T) use slab k.  On the fp16-pair arithmetic it is built by mf_conv2d_wino_f16x2 itself (conv_f16x2.hip);
  // on the exact arithmetics (MF_CONV_FP32, MF_CONV_FP32_SPLIT3_W3) a caller passes it to mf_conv2d_f32 directly (ABI 250: mf_wino_input_f32 / mf_wino_tail_f32)
  MF_REQUIRE(d->upsample >= 0 && (d->upsample <= 2 || (d->upsample == 3 && (d->precision == MF_CONV_FP32_F16X2 || d->precision == MF_CONV_FP32 ||
                                                                              d->precision == MF_CONV_FP32_SPLIT3_W3) &&
                                                           d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && d->N % 16 == 0)), MF_EINVAL,
             "conv: upsample flag");
  MF_REQUIRE(d->precision >= 0 && d->precision <= MF_CONV_F16, MF_EINVAL, "conv: precision flag %d", d->precision);
  MF_REQUIRE(d->upsample != 2 || (d->KH == 3 && d->stride == 1 && d->pad == 1), MF_EINVAL, "conv: the sub-pixel form is nearest-x2 + 3x3 stride 1 pad 1");
  MF_REQUIRE(d->pad >= 0 && d->pad <= 1, MF_EUNSUPPORTED, "conv: pad %d unsupported", d->pad);
  MF_REQUIRE(!(d->in_layout == MF_LAYOUT_NCHW && d->C2 != 0), MF_EUNSUPPORTED, "conv: NCHW input with two sources");
  const int up = (d->upsample == 1 || d->upsample == 2) ? 1 : 0;
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


// split-K reduction: y = sum_z slabs[z] + bias (+ the measured max of |y|: slot (block, wave) of out_bound[n][gridDim.x * 4]).
// grid (blocks, N): a block row works inside one sample.
template <int UNUSED = 0>
__global__ void splitk_reduce_kernel(const float* __restrict__ slabs, const float* __restrict__ bias, float* __restrict__ y,
                                     long per_sample4, int Cout, int splitk, long slab, float* __restrict__ out_bound) {
  const long stride = (long)gridDim.x * blockDim.x;
  const long base = (long)blockIdx.y * per_sample4;
  float vmax = 0.f;
  for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < per_sample4; j += stride) {
    const long e = (base + j) * 4;
    float4 s = sum_slabs(slabs + e, slab, splitk);
    if (bias) {
      const float4 b = *reinterpret_cast<const float4*>(bias + (e % Cout));
      s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
    }
    *reinterpret_cast<float4*>(y + e) = s;
    vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(s.x), fabsf(s.y))), fmaxf(fabsf(s.z), fabsf(s.w)));
  }
  if (out_bound) {
    vmax = wave_max(vmax);
    if ((threadIdx.x & 63) == 0) out_bound[((long)blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)] = vmax;
  }
}

// conv_f16x2.hip
int f16x2_gn_parts(const MfConvDesc* d, int G);
size_t f16x2_workspace_bytes(const MfConvDesc* d);
// conv.hip
int igemm_plan_query(const MfConvDesc* d, int32_t* tile_id, int32_t* splitk);

}  // namespace mf
