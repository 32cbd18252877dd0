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
//      MODE 3  MF_CONV_FP32_SPLIT3_W3: every fp32 operand split exactly into 3 bf16 terms (the weights once at load time, the
//              activations by the staging threads), 6 product terms on v_mfma_f32_32x32x16_bf16, fp32 accumulate.  LDS rows
//              [3 pieces][32 bf16] + pad.  The chunk's other work is cut into units pinned between the MFMAs; ONE barrier per chunk in
//              the middle of the MFMA stream, next chunk's first-step fragments prefetched behind it.
//      MODE 5  MF_CONV_BF16: opt-in reduced precision (one bf16 term).
//    (The default arithmetic of the product, MF_CONV_FP32_F16X2, has its own kernel and translation unit: conv_f16x2.{h,hip}.)
//    FG: fast gather addressing (no fused nearest-x2 gather, < 2^24 source pixels).
//  * conv_smallcin_kernel / conv_direct_kernel: the edge convolutions (Cin = 8|3, Cout = 8|3|16, NCHW edges): <0.2 % of the FLOPs.
//  * DESIGN.md section 3 has the measurements and what was tried and rejected.
#include "common.h"
#include "gn_partial.h"
#include <cstdlib>

using namespace mf;

#include "conv_igemm.h"
#include "conv_plan.h"

namespace {

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


// 1x1 output convolutions with a handful of output channels (UNet outc 256 -> 8, VAE outc 64 -> 3, the deep-supervision heads): NHWC
// input, one source, C % 32 == 0, C / 32 a power of two <= 32, Cout <= 8.  C / 32 neighbouring lanes share a pixel, each reads 32 of its
// channels ONCE (the generic kernel reads a pixel row once per output channel), multiplies them into Cout partial sums and the lanes of a
// pixel add up with xor shuffles; lane q of a pixel stores outputs q, q + C/32, ...  grid: pixels / (256 / (C/32))
template <int CO>
__global__ __launch_bounds__(256) void conv_out1x1_kernel(const ConvP p) {
  __shared__ float4 wsm[CO * 256];                        // the whole weight matrix [Cout][C] (C <= 1024)
  const int lpp = p.C1 >> 5;                              // lanes per pixel
  const int q = threadIdx.x & (lpp - 1);
  const long m = ((long)blockIdx.x * 256 + threadIdx.x) / lpp;
  const bool live = m < p.M;
  float4 xv[8];
  if (live) {
    const float* xa = p.x1 + m * p.C1 + q * 32;           // (the activation loads are in flight while the weights go to LDS)
#pragma unroll
    for (int i = 0; i < 8; ++i) xv[i] = *reinterpret_cast<const float4*>(xa + 4 * i);
  }
  const int c4 = p.C1 >> 2;
  for (int i = threadIdx.x; i < p.Cout * c4; i += 256) wsm[i] = reinterpret_cast<const float4*>(p.w)[i];
  __syncthreads();
  float acc[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) acc[c] = 0.f;
  if (live) {
#pragma unroll
    for (int c = 0; c < CO; ++c) {
      if (c < p.Cout) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float4 wv = wsm[c * c4 + q * 8 + i];
          acc[c] = fmaf(xv[i].x, wv.x, acc[c]); acc[c] = fmaf(xv[i].y, wv.y, acc[c]);
          acc[c] = fmaf(xv[i].z, wv.z, acc[c]); acc[c] = fmaf(xv[i].w, wv.w, acc[c]);
        }
      }
    }
  }
  for (int off = 1; off < lpp; off <<= 1) {
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] += __shfl_xor(acc[c], off, 64);
  }
  if (!live) return;
  const int n = (int)(m / p.HWout);
  const long rem = m - (long)n * p.HWout;
#pragma unroll
  for (int c = 0; c < CO; ++c) {
    if (c < p.Cout && (c & (lpp - 1)) == q) {
      const float v = acc[c] + (p.bias ? p.bias[c] : 0.f);
      if (p.out_nchw) p.y[((long)n * p.Cout + c) * p.HWout + rem] = v;
      else p.y[m * p.Cout + c] = v;
    }
  }
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
  // thread = output channel: reads its weight row (stride K between lanes: the rows of a wave stay in L1 for the whole loop) and writes
  // W^T with consecutive lanes on consecutive banks.  (The coalesced-read form -- lane = k -- stored with a stride of 256 floats: every
  // lane of a wave on the same bank, 72 64-way conflicts per thread, ~8 us of the 35 us launch.)
  if (tid < nco) {
    const float* wr = p.w + (long)(co0 + tid) * p.K;
    for (int k = 0; k < p.K; ++k) wT[k * kSmallCo + tid] = wr[k];
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

// Round 6: the same convolution, the same fma chain per output (bias, then k = 0 .. K-1 in order: BIT-IDENTICAL results), with the two parts of the
// kernel above that were not arithmetic taken out of its critical path (27.7 us per launch for 0.6 GFLOP at the UNet's in_conv, B = 16):
//   * the im2col patch was filled element by element with six integer divisions by run-time values per element (~ 200 instructions x 18 elements
//     per thread).  Here a thread owns ONE pixel of the group (its coordinates: two divisions per group) and walks k in steps of 4 through a
//     table of (ky, kx, ci) built once per workgroup;
//   * the patch is laid out [k][pixel] instead of [pixel][k]: the inner loop reads the 16 pixel values of a k as four 16-byte LDS broadcasts
//     instead of sixteen 4-byte ones (5 LDS instructions per 64 fmas instead of 17).
// MF_SMALLCIN=0 launches the round-5 kernel (A/B and the bit-equality test).
__global__ __launch_bounds__(256) void conv_smallcin2_kernel(const ConvP p, int groups) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* wT = sm;                                                  // [K][kSmallCo]
  float* patch = sm + (size_t)p.K * kSmallCo;                      // [kSmallPG][K][kSmallPix]
  int* ktab = reinterpret_cast<int*>(patch + (size_t)kSmallPG * kSmallPix * p.K);   // [K]: ky | kx << 8 | ci << 16
  const int tid = threadIdx.x;
  const int co0 = blockIdx.y * kSmallCo;
  const int nco = min(kSmallCo, p.Cout - co0);
  if (tid < nco) {
    const float* wr = p.w + (long)(co0 + tid) * p.K;
    for (int k = 0; k < p.K; ++k) wT[k * kSmallCo + tid] = wr[k];
  }
  if (tid < p.K) {
    const int tap = tid / p.Cin, ci = tid - tap * p.Cin;
    const int ky = tap / p.KW, kx = tap - ky * p.KW;
    ktab[tid] = ky | (kx << 8) | (ci << 16);
  }
  const int pg = tid >> 6, cq = tid & 63;
  const bool active = cq * 4 < nco;
  float bias4[4] = {0.f, 0.f, 0.f, 0.f};
  if (active && p.bias) {
#pragma unroll
    for (int j = 0; j < 4; ++j) bias4[j] = p.bias[co0 + cq * 4 + j];
  }
  const int fpx = tid >> 2, fk0 = tid & 3;            // fill: pixel 0 .. 63 of the 4 x 16 pixels of an iteration, first k
  const int fgq = fpx >> 4, fq = fpx & 15;
  for (int g0 = blockIdx.x * kSmallPG; g0 < groups; g0 += gridDim.x * kSmallPG) {
    __syncthreads();
    {
      const int m = (g0 + fgq) * kSmallPix + fq;
      const bool inside = m < p.M;
      int n = 0, iy0 = 0, ix0 = 0;
      if (inside) {
        n = m / p.HWout;
        const int rem = m - n * p.HWout;
        const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
        iy0 = oy * p.stride - p.pad;
        ix0 = ox * p.stride - p.pad;
      }
      float* dst = patch + (size_t)fgq * kSmallPix * p.K + fq;
      for (int k = fk0; k < p.K; k += 4) {
        const int e = ktab[k];
        const int iy = iy0 + (e & 255), ix = ix0 + ((e >> 8) & 255), ci = e >> 16;
        float v = 0.f;
        if (inside && (unsigned)iy < (unsigned)p.Heff && (unsigned)ix < (unsigned)p.Weff) {
          const int sy = iy >> p.ups, sx = ix >> p.ups;
          v = p.in_nchw ? p.x1[((long)(n * p.C1 + ci) * p.Hin + sy) * p.Win + sx]
                        : (ci < p.C1 ? p.x1[((long)(n * p.Hin + sy) * p.Win + sx) * p.C1 + ci]
                                     : p.x2[((long)(n * p.Hin + sy) * p.Win + sx) * p.C2 + (ci - p.C1)]);
        }
        dst[k * kSmallPix] = v;
      }
    }
    __syncthreads();
    const int m0 = (g0 + pg) * kSmallPix;
    if (active && m0 < p.M) {
      float acc[kSmallPix][4];
#pragma unroll
      for (int q = 0; q < kSmallPix; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[q][j] = bias4[j];
      const float* pp = patch + (size_t)pg * kSmallPix * p.K;
      for (int k = 0; k < p.K; ++k) {
        const float4 wv = *reinterpret_cast<const float4*>(wT + k * kSmallCo + cq * 4);
        float xq[kSmallPix];
#pragma unroll
        for (int q4 = 0; q4 < kSmallPix / 4; ++q4) {
          const float4 xv4 = *reinterpret_cast<const float4*>(pp + k * kSmallPix + 4 * q4);
          xq[4 * q4] = xv4.x; xq[4 * q4 + 1] = xv4.y; xq[4 * q4 + 2] = xv4.z; xq[4 * q4 + 3] = xv4.w;
        }
#pragma unroll
        for (int q = 0; q < kSmallPix; ++q) {
          const float xv = xq[q];
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
const TileCfg kCfgs[] = {
    {1, 128, 128, 2, 2, 32}, {2, 128, 64, 2, 2, 32}, {3, 64, 128, 2, 2, 32}, {4, 64, 64, 2, 2, 32}, {5, 128, 32, 4, 1, 32}, {6, 64, 32, 2, 1, 32},
    {7, 128, 128, 4, 2, 32}, {8, 128, 128, 2, 4, 32}, {9, 128, 256, 2, 4, 32}, {10, 256, 128, 4, 2, 32},
    {23, 64, 128, 2, 2, 64}, {24, 64, 64, 2, 2, 64}, {27, 128, 128, 4, 2, 64}, {28, 128, 128, 2, 4, 64},  // BK = 64 (needs C1, C2 % 64 == 0)
};

int make_plan(const MfConvDesc* d, Plan* pl) {
  int rc = fill_geometry(d, pl);
  if (rc) return rc;
  MF_REQUIRE(d->precision != 1 && d->precision != 2, MF_EUNSUPPORTED,
             "conv: precision %d (in-kernel weight split / chunk-sum variant of the bf16-triplet arithmetic) was retired in ABI 200: use MF_CONV_FP32_SPLIT3_W3", d->precision);
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
  if (d->upsample == 3) {   // the component GEMMs of a Winograd convolution (conv_plan.h): implicit-GEMM path only, a tile inside one component
    MF_REQUIRE(pl->igemm && ((long)(d->N / 16) * hw_src) % 64 == 0, MF_EUNSUPPORTED, "conv: the component GEMM needs the implicit-GEMM path and rows per component %% 64 == 0");
  }
  if (!pl->igemm) return MF_OK;
  int nk = taps * (Cin / 32);
  if (d->tile_hint > 0) {
    const TileCfg* c = nullptr;
    for (const auto& k : kCfgs) if (k.id == d->tile_hint) c = &k;
    MF_REQUIRE(c && d->Cout % c->BN == 0, MF_EINVAL, "conv: bad tile_hint %d for Cout %d", d->tile_hint, d->Cout);
    MF_REQUIRE(d->C1 % c->BK == 0 && d->C2 % c->BK == 0, MF_EINVAL, "conv: tile_hint %d needs channel counts divisible by %d", d->tile_hint, c->BK);
    MF_REQUIRE(d->precision == MF_CONV_FP32 || c->BK == 32, MF_EINVAL, "conv: tile_hint %d is not built for the split-bf16 mode", d->tile_hint);
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
      // small M, long K (8x8 level at B = 8): the wide tiles cannot reach 256 workgroups even at split-K 16 -> more, smaller tiles
      if ((id == 9 || id == 10) && (long)cdiv(pl->M, id == 9 ? 128 : 256) * (d->Cout / (id == 9 ? 256 : 128)) * 16 < 224 &&
          !(d->upsample == 2 && hw_src % 128))
        id = 8;
    } else if (d->Cout % 128 == 0 && pl->M >= 128 && gflop >= 6.0 && !(d->upsample == 2 && hw_src % 128)) {
      id = 8;                              // 8 waves, 128x128: best for every large 3x3 shape
    } else if (d->Cout % 64 == 0) {
      id = c64 ? 24 : 4;                   // 64x64 (BK = 64 when the channels allow): short-K 1x1 residual convs, stride-2 convs and
    }                                      // other < 6 GFLOP problems are launch-cost-bound: more, smaller workgroups and less split-K
    for (const auto& k : kCfgs) if (k.id == id) pl->cfg = k;
  }
  static const int gemm_rule = [] { const char* e = getenv("MF_WINO_F32_PLAN"); return e ? atoi(e) : 1; }();   // 0: the generic rule above (A/B)
  if (d->upsample == 3 && d->tile_hint <= 0 && gemm_rule) {
    // the component GEMMs of a Winograd convolution (scripts/wino_f32_sweep.py on MI355X, B = 16: the generic rule above was 10 - 25 % off the best on eight of
    // ten shapes): the LARGEST tile that still gives one workgroup per CU without split-K -- 128 x 256, then 128 x 128 --, else 128 x 128 with the K loop
    // split in two where that loop is long enough (>= 32 chunks), else 64 x 128
    const long rows = (long)(d->N / 16) * hw_src;
    auto tiles_of = [&](int bm, int bn) { return rows % bm == 0 && d->Cout % bn == 0 ? (long)cdiv(pl->M, bm) * (d->Cout / bn) : 0L; };
    int id = 0, sk = 0;
    if (tiles_of(128, 256) >= 256) { id = 9; sk = 1; }
    else if (tiles_of(128, 128) >= 256) { id = 8; sk = 1; }
    else if (tiles_of(128, 128) >= 128 && Cin / 32 >= 32) { id = 8; sk = 2; }
    else if (tiles_of(64, 128) > 0) { id = 3; sk = 1; }
    if (id) {
      for (const auto& k : kCfgs) if (k.id == id) pl->cfg = k;
      if (d->splitk_hint <= 0) {
        nk = Cin / pl->cfg.BK;
        pl->nk_per_split = cdiv(nk, sk);
        pl->splitk = cdiv(nk, pl->nk_per_split);
        return MF_OK;
      }
    }
  }
  if (d->upsample == 3 && ((long)(d->N / 16) * hw_src) % pl->cfg.BM) {
    // the chosen tile straddles two components: the largest built tile of the same arithmetic whose rows divide a component
    const long rows = (long)(d->N / 16) * hw_src;
    const TileCfg* alt = nullptr;
    for (const auto& k : kCfgs)
      if (k.BK == 32 && k.id != 5 && k.id != 6 && rows % k.BM == 0 && d->Cout % k.BN == 0 && (alt == nullptr || k.BM * k.BN > alt->BM * alt->BN)) alt = &k;
    MF_REQUIRE(alt != nullptr && d->tile_hint <= 0, MF_EUNSUPPORTED, "conv: no tile whose rows divide a component of this GEMM");
    pl->cfg = *alt;
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
  static DeviceOnce once;
  if (first_use_on_device(once))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, WM, WN, BK, MODE, FG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int grid = p.tiles_m * p.tiles_n * p.splitk;
  MF_LAUNCH((conv_igemm_kernel<BM, BN, WM, WN, BK, MODE, FG>), dim3(grid), dim3(WM * WN * 64), lds, s, p);
  return check_launch("conv_igemm");
}

template <int BM, int BN, int WM, int WN, int BK = 32, int MODE = 0>
int launch_igemm(const ConvP& p, hipStream_t s) {
  return p.fastg ? launch_igemm_fg<BM, BN, WM, WN, BK, MODE, true>(p, s) : launch_igemm_fg<BM, BN, WM, WN, BK, MODE, false>(p, s);
}

}  // namespace

namespace mf {
int igemm_plan_query(const MfConvDesc* d, int32_t* tile_id, int32_t* splitk) {
  Plan pl;
  int rc = make_plan(d, &pl);
  if (rc) return rc;
  if (tile_id) *tile_id = pl.igemm ? pl.cfg.id : 0;
  if (splitk) *splitk = pl.igemm ? pl.splitk : 0;
  return MF_OK;
}
}  // namespace mf

extern "C" {

int mf_pack_conv_weight_f32(const float* w, float* out, int Cout, int Cin, int KH, int KW, void* stream) {
  MF_REQUIRE(w && out && Cout > 0 && Cin > 0 && KH > 0 && KW > 0, MF_EINVAL, "pack_conv_weight: bad args");
  const long total = (long)Cout * Cin * KH * KW;
  ProfScope ps(MF_FAM_MISC, (hipStream_t)stream, 0, 8.0 * total);
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  MF_LAUNCH(pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, out, Cout, Cin, KH, KW);
  return check_launch("pack_conv_weight");
}

// Number of per-sample partial records the convolution itself can emit for a following GroupNorm with G groups
// (0: it cannot -- split-K, direct kernels, a tile straddling two samples; use mf_gn_stats_partial_f32 then).
int mf_conv2d_gn_parts(const MfConvDesc* d, int G) {
  if (d && (d->precision == MF_CONV_FP32_F16X2 || d->precision == MF_CONV_F16)) return mf::f16x2_gn_parts(d, G);
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
  MF_LAUNCH(pack_upconv_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, out, Cout, Cin);
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
  MF_LAUNCH(split_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_packed, reinterpret_cast<u32x4*>(out), octets);
  return check_launch("split_conv_weight");
}

int mf_convert_conv_weight_bf16(const float* w_packed, void* out, long rows, int K, void* stream) {
  MF_REQUIRE(w_packed && out && rows > 0 && K > 0 && K % 8 == 0, MF_EINVAL, "convert_conv_weight_bf16: bad args (K %% 8 == 0)");
  const long quads = rows * (K / 4);
  ProfScope ps(MF_FAM_MISC, (hipStream_t)stream, 0, 6.0 * rows * K);
  const int blocks = (int)((quads + 255) / 256 > 4096 ? 4096 : (quads + 255) / 256);
  MF_LAUNCH(convert_weight_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_packed, reinterpret_cast<u32x2*>(out), quads);
  return check_launch("convert_conv_weight_bf16");
}

/* 1 if `d` (with upsample = 2) can run in the sub-pixel form, else 0 (then use upsample = 1 with the regular packing) */
int mf_conv2d_subpixel_ok(const MfConvDesc* d) {
  Plan pl;
  return d && d->upsample == 2 && make_plan(d, &pl) == MF_OK && pl.igemm ? 1 : 0;
}

size_t mf_conv2d_workspace_bytes(const MfConvDesc* d) {
  if (d && (d->precision == MF_CONV_FP32_F16X2 || d->precision == MF_CONV_F16)) return mf::f16x2_workspace_bytes(d);
  Plan pl;
  if (make_plan(d, &pl) != MF_OK) return 0;
  if (!pl.igemm || pl.splitk <= 1) return 0;
  return (size_t)pl.splitk * pl.M * d->Cout * sizeof(float);
}

struct GnOut { double* partial; int G; };
static int conv2d_impl(const float* x1, const float* x2, const float* w, const float* bias, float* y, void* workspace, size_t workspace_bytes,
                       const GnOut& gn, const MfConvDesc* d, void* stream);

int mf_conv2d_f32(const float* x1, const float* x2, const float* w, const float* bias, float* y, void* workspace,
                  size_t workspace_bytes, const MfConvDesc* d, void* stream) {
  return conv2d_impl(x1, x2, w, bias, y, workspace, workspace_bytes, GnOut{nullptr, 0}, d, stream);
}

int mf_conv2d_gn_f32(const float* x1, const float* x2, const float* w, const float* bias, float* y, void* workspace,
                     size_t workspace_bytes, double* gn_partial, int G, const MfConvDesc* d, void* stream) {
  MF_REQUIRE(gn_partial && mf_conv2d_gn_parts(d, G) > 0, MF_EUNSUPPORTED, "conv_gn: this convolution cannot emit GroupNorm partials (mf_conv2d_gn_parts == 0)");
  return conv2d_impl(x1, x2, w, bias, y, workspace, workspace_bytes, GnOut{gn_partial, G}, d, stream);
}

static int conv2d_impl(const float* x1, const float* x2, const float* w, const float* bias, float* y, void* workspace, size_t workspace_bytes,
                       const GnOut& gn, const MfConvDesc* d, void* stream) {
  double* gn_partial = gn.partial;
  const int G = gn.G;
  MF_REQUIRE(d && d->precision != MF_CONV_FP32_F16X2 && d->precision != MF_CONV_F16, MF_EINVAL,
             "conv: MF_CONV_FP32_F16X2 / MF_CONV_F16 take fp16-pair operands: call mf_conv2d_f16x2");
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
  p.wphase_rows = d->upsample == 3 ? (d->N / 16) * d->Hin * d->Win : 0;   // component GEMMs of a Winograd convolution: weight slab per row block
  p.gn_partial = gn_partial; p.gn_groups = G; p.gn_cpg = G > 0 ? d->Cout / G : 1;
  p.gn_parts = (gn_partial && pl.igemm && pl.splitk == 1) ? (pl.Hout * pl.Wout) / pl.cfg.BM : 0;
  if (pl.igemm && pl.splitk > 1) p.gn_partial = nullptr;  // the reducer, not the conv kernel, emits them
  p.fastg = (p.ups == 0 && (long)d->N * d->Hin * d->Win < (1L << 24) && d->C1 < (1 << 22) && d->C2 < (1 << 22)) ? 1 : 0;
  static const bool generic_gather = getenv("MF_CONV_GENERIC_GATHER") != nullptr;  // A/B switch (scripts), read once
  if (generic_gather) p.fastg = 0;
  // algorithmic FLOPs of the reference op (the sub-pixel form does 4/9 of the MACs of nearest-x2 + 3x3)
  const double flops = 2.0 * pl.M * (double)d->Cout * (d->upsample == 2 ? 9.0 * (d->C1 + d->C2) : (double)pl.K);
  const double bytes = 4.0 * ((double)d->N * d->Hin * d->Win * p.Cin + (double)d->Cout * pl.K + (double)pl.M * d->Cout);

  if (!pl.igemm && !p.out_nchw && p.Cin <= 16 && pl.K <= kSmallMaxK && d->Cout % 64 == 0 &&
      ((size_t)pl.K * kSmallCo + (size_t)kSmallPG * kSmallPix * pl.K) * sizeof(float) <= 150 * 1024) {
    ProfScope ps(MF_FAM_CONV_DIRECT, s, flops, bytes);
    const int groups = cdiv(pl.M, kSmallPix);
    const int cotiles = cdiv(d->Cout, kSmallCo);
    const size_t lds = ((size_t)pl.K * kSmallCo + (size_t)kSmallPG * kSmallPix * pl.K) * sizeof(float);
    static DeviceOnce once;
    if (first_use_on_device(once)) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_smallcin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_smallcin2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    int gx = cdiv(groups, kSmallPG);
    if (gx > 1024) gx = 1024;
    static const int fast = [] { const char* e = getenv("MF_SMALLCIN"); return e ? atoi(e) : 1; }();   // 0: the round-5 kernel (A/B; bit-identical results)
    if (fast && pl.K <= 255 && d->KH <= 255 && p.Cin <= 255) {
      MF_LAUNCH(conv_smallcin2_kernel, dim3(gx, cotiles), dim3(256), lds + (size_t)pl.K * sizeof(int), s, p, groups);
    } else {
      MF_LAUNCH(conv_smallcin_kernel, dim3(gx, cotiles), dim3(256), lds, s, p, groups);
    }
    return check_launch("conv_smallcin");
  }
  if (!pl.igemm && d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && d->upsample == 0 && !p.in_nchw && d->C2 == 0 && d->Cout <= 8 &&
      d->C1 % 32 == 0 && d->C1 / 32 <= 32 && ((d->C1 / 32) & (d->C1 / 32 - 1)) == 0 &&
      (reinterpret_cast<uintptr_t>(x1) & 15) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0) {   // (float4 reads of x and w: else the generic kernel)
    ProfScope ps(MF_FAM_CONV_DIRECT, s, flops, bytes);
    const int lpp = d->C1 / 32;
    const long blocks = ((long)pl.M * lpp + 255) / 256;
    MF_REQUIRE(blocks < (1L << 31), MF_EUNSUPPORTED, "conv(out 1x1): %ld workgroups", blocks);
    if (d->Cout <= 4)
      MF_LAUNCH(conv_out1x1_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, p);
    else
      MF_LAUNCH(conv_out1x1_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, s, p);
    return check_launch("conv_out1x1");
  }
  if (!pl.igemm) {
    ProfScope ps(MF_FAM_CONV_DIRECT, s, flops, bytes);
    const long total = (long)pl.M * d->Cout;
    const int blocks = (int)((total + 255) / 256);
    if (p.out_nchw)
      MF_LAUNCH(conv_direct_kernel<true>, dim3(blocks), dim3(256), 0, s, p);
    else
      MF_LAUNCH(conv_direct_kernel<false>, dim3(blocks), dim3(256), 0, s, p);
    return check_launch("conv_direct");
  }

  p.tiles_m = cdiv(pl.M, pl.cfg.BM);
  p.tiles_n = d->Cout / pl.cfg.BN;
  {
    const double b1 = 4.0 * d->N * d->Hin * d->Win * d->C1, b2 = 4.0 * d->N * d->Hin * d->Win * d->C2, bw = (d->precision == MF_CONV_FP32_SPLIT3_W3 ? 6.0 : d->precision == MF_CONV_BF16 ? 2.0 : 4.0) * d->Cout * pl.K * (p.subpix ? 4 : d->upsample == 3 ? 16 : 1);
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
    const double terms = d->precision == MF_CONV_FP32_SPLIT3_W3 ? 6.0 : 1.0;
    ProfScope ps(MF_FAM_CONV_IGEMM, s, flops, bytes, 2.0 * pl.M * (double)d->Cout * pl.K * terms);
    if (d->precision == MF_CONV_BF16) {
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
      MF_LAUNCH(gn_partial_kernel<true>, dim3(chunks, d->N, slices), dim3(kStatsThreads), stats_lds_bytes(d->Cout / slices), s,
                         reinterpret_cast<const float*>(workspace), gn_partial, HW, d->Cout, G, pl.splitk, p.slab, bias, y, (float*)nullptr);
      return check_launch("splitk_reduce_stats");
    }
    const long p4 = (long)pl.Hout * pl.Wout * d->Cout / 4;   // float4s per sample
    ProfScope ps(MF_FAM_SPLITK_REDUCE, s, 0, 4.0 * pl.M * d->Cout * (pl.splitk + 1));
    int bx = (int)((p4 + 255) / 256);
    const int cap = cdiv(2048, d->N);
    if (bx > cap) bx = cap;
    MF_LAUNCH(splitk_reduce_kernel<0>, dim3(bx, d->N), dim3(256), 0, s, reinterpret_cast<const float*>(workspace), bias, y, p4, d->Cout,
                       pl.splitk, p.slab, (float*)nullptr);
    return check_launch("splitk_reduce");
  }
  return MF_OK;
}


}  // extern "C"
