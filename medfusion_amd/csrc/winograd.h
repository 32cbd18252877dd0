// winograd.h -- the three small passes of the Winograd F(2x2, 3x3) form of a 3x3 stride-1 pad-1 convolution on the fp16-pair arithmetic
// (round 5; conv_blocks.py:185 via unet2.py:250-264: 94.4 % of the UNet's FLOPs).  Included by conv_f16x2.hip only.
//
//   y = A^T [ sum_c (G g G^T) . (B^T d B) ] A        g: 3x3 filter, d: 4x4 input patch of a 2x2 output tile, "." element-wise
//
// 16 products per (2x2 outputs, channel pair) instead of 36: 2.25x fewer matrix instructions.  The 16 element-wise products, summed over the
// input channels, are 16 INDEPENDENT GEMMs  M_k[tile][cout] = sum_c V_k[tile][c] U_k[cout][c]  -- a 1x1 "convolution" whose weight slab
// depends on the component k.  They run on the UNCHANGED implicit-GEMM kernel (conv_f16x2_body.inc; ConvP2::wphase_rows selects the slab per
// row block), both operands as fp16 pairs with per-(component, sample) power-of-two scales.  This file holds what surrounds the GEMM:
//   wino_pack_weight_kernel   U = G g G^T in fp64, rounded once to fp32 (load time; then split into fp16 pairs like any weight)
//   wino_input_kernel         V = B^T d B from the fp16-pair form of the activation -> fp16 pairs [16][N * tiles][C], bound 4 x bound(x)
//   wino_output_kernel        y = A^T M A + bias, fp32 NHWC, + the partial GroupNorm records of the GroupNorm that follows
// Layouts: tiles of a sample row-major over (H/2, W/2); component k = 4 i + j (i: row of the 4x4 transform domain, j: column).
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]     G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]     A^T = [1 1 1 0; 0 1 -1 -1]
// Accuracy: the transforms add +-1 multiples only; V is formed in fp32 from the exactly reconstructed pair values (sums of 4: <= 2 roundings
// of 2^-24 relative to 4 max|d|), U carries one fp32 rounding, the products run on the same three-term pair arithmetic as the direct form,
// A^T M A adds 9 of the 16 M values per output in fp32.  Measured against fp64: profiles/r05_winograd_ab.txt.
#pragma once
#include "common.h"
#include "split_f16.h"

namespace mfw {
using namespace mf;
typedef float wf32x4 __attribute__((ext_vector_type(4)));

// one thread per (cout, cin): reads the 9 taps of OIHW, writes the 16 components of [16][Cout][Cin]
__global__ __launch_bounds__(256) void wino_pack_weight_kernel(const float* __restrict__ w, float* __restrict__ u, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double g[3][3], t[4][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) g[r][c] = (double)w[i * 9 + r * 3 + c];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    t[0][c] = g[0][c];
    t[1][c] = 0.5 * (g[0][c] + g[1][c] + g[2][c]);
    t[2][c] = 0.5 * (g[0][c] - g[1][c] + g[2][c]);
    t[3][c] = g[2][c];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const double u0 = t[r][0], u1 = 0.5 * (t[r][0] + t[r][1] + t[r][2]), u2 = 0.5 * (t[r][0] - t[r][1] + t[r][2]), u3 = t[r][2];
    u[(long)(4 * r + 0) * n + i] = (float)u0;
    u[(long)(4 * r + 1) * n + i] = (float)u1;
    u[(long)(4 * r + 2) * n + i] = (float)u2;
    u[(long)(4 * r + 3) * n + i] = (float)u3;
  }
}

// 4 consecutive channels of a tensor stored as fp16 pairs, as the SCALED fp32 values x 2^-s (hi + lo' / 2048: exact in fp32)
__device__ __forceinline__ float4 wino_load_pairs4(const void* pairs, long e) {
  const uint2* g = reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(pairs) + (e >> 3) * 32 + ((e >> 2) & 1) * 8);
  const uint2 h = g[0], l = g[2];   // [hi x 8] then [lo' x 8]: 16 bytes apart
  const sf_f16x2 h0 = __builtin_bit_cast(sf_f16x2, h.x), h1 = __builtin_bit_cast(sf_f16x2, h.y), l0 = __builtin_bit_cast(sf_f16x2, l.x),
                 l1 = __builtin_bit_cast(sf_f16x2, l.y);
  return make_float4((float)h0[0] + (float)l0[0] * kLoInv, (float)h0[1] + (float)l0[1] * kLoInv, (float)h1[0] + (float)l1[0] * kLoInv,
                     (float)h1[1] + (float)l1[1] * kLoInv);
}

// B^T d B of one channel, in place: d[4 r + c]
__device__ __forceinline__ void wino_bt_d_b(float* d) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {   // rows: B^T d
    const float d0 = d[c], d1 = d[4 + c], d2 = d[8 + c], d3 = d[12 + c];
    d[c] = d0 - d2; d[4 + c] = d1 + d2; d[8 + c] = d2 - d1; d[12 + c] = d1 - d3;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {   // columns: (B^T d) B
    const float t0 = d[4 * r], t1 = d[4 * r + 1], t2 = d[4 * r + 2], t3 = d[4 * r + 3];
    d[4 * r] = t0 - t2; d[4 * r + 1] = t1 + t2; d[4 * r + 2] = t2 - t1; d[4 * r + 3] = t1 - t3;
  }
}

// Input transform.  One thread per (sample, tile, 4 channels); lanes 2k / 2k+1 hold the two halves of one 8-channel group (store_split4).
// xs: fp16 pairs [N][H][W][C] scaled per sample by xbound[n]; vs: fp16 pairs [16][N][T][C], component k of sample n scaled by
// vbound[k N + n] = 4 xbound[n] (every transform-domain value is a +-1 sum of four inputs).
__global__ __launch_bounds__(256) void wino_input_kernel(const void* __restrict__ xs, const float* __restrict__ xbound, void* __restrict__ vs,
                                                          float* __restrict__ vbound, int N, int H, int W, int C) {
  const int C4 = C >> 2, TW = W >> 1, T = (H >> 1) * TW;
  const long total = (long)N * T * C4, stride = (long)gridDim.x * 256;
  const long plane = (long)N * T * C;   // elements of one component
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int c4 = (int)(i % C4);
    const long r = i / C4;
    const int tile = (int)(r % T), n = (int)(r / T);
    const int ty = tile / TW, tx = tile - ty * TW;
    const float b = xbound ? xbound[n] : 0.f;
    const float vb = 4.f * b;
    // x = x' 2^s, v = transform(x') 2^s, stored as v 2^-sv: factor 2^(s - sv) (= 1/4 away from the clamps of scale_exp_of)
    const float f = xbound ? exp2i(scale_exp_of(b)) * exp2i(-scale_exp_of(vb)) : 1.f;
    float d[4][16];
#pragma unroll
    for (int py = 0; py < 4; ++py)
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        const int y = 2 * ty - 1 + py, x = 2 * tx - 1 + px;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) v = wino_load_pairs4(xs, (((long)n * H + y) * W + x) * C + c4 * 4);
        d[0][4 * py + px] = v.x; d[1][4 * py + px] = v.y; d[2][4 * py + px] = v.z; d[3][4 * py + px] = v.w;
      }
#pragma unroll
    for (int k = 0; k < 4; ++k) wino_bt_d_b(d[k]);
    const long e0 = ((long)n * T + tile) * C + c4 * 4;
#pragma unroll
    for (int k = 0; k < 16; ++k) store_split4<true>(vs, (long)k * plane + e0, d[0][k], d[1][k], d[2][k], d[3][k], f);
    if (vbound && tile == 0 && c4 == 0) {
#pragma unroll
      for (int k = 0; k < 16; ++k) vbound[k * N + n] = vb;
    }
  }
}

// The same transform on fp32 tensors (round 6: the Winograd form on the EXACT arithmetics -- MF_CONV_FP32_SPLIT3_W3 / MF_CONV_FP32 -- whose operands
// are plain fp32): x fp32 NHWC [N][H][W][C] -> V = B^T d B fp32 [16][N][T][C].  One thread per (sample, tile, 4 channels).
__global__ __launch_bounds__(256) void wino_input_f32_kernel(const float* __restrict__ x, float* __restrict__ v, int N, int H, int W, int C) {
  const int C4 = C >> 2, TW = W >> 1, T = (H >> 1) * TW;
  const long total = (long)N * T * C4, stride = (long)gridDim.x * 256;
  const long plane = (long)N * T * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int c4 = (int)(i % C4);
    const long r = i / C4;
    const int tile = (int)(r % T), n = (int)(r / T);
    const int ty = tile / TW, tx = tile - ty * TW;
    float d[4][16];
#pragma unroll
    for (int py = 0; py < 4; ++py)
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        const int y = 2 * ty - 1 + py, xx = 2 * tx - 1 + px;
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W) q = *reinterpret_cast<const float4*>(x + (((long)n * H + y) * W + xx) * C + c4 * 4);
        d[0][4 * py + px] = q.x; d[1][4 * py + px] = q.y; d[2][4 * py + px] = q.z; d[3][4 * py + px] = q.w;
      }
#pragma unroll
    for (int k = 0; k < 4; ++k) wino_bt_d_b(d[k]);
    const long e0 = ((long)n * T + tile) * C + c4 * 4;
#pragma unroll
    for (int k = 0; k < 16; ++k) __builtin_nontemporal_store(wf32x4{d[0][k], d[1][k], d[2][k], d[3][k]}, reinterpret_cast<wf32x4*>(v + (long)k * plane + e0));
  }
}

// Output transform + bias (+ GroupNorm partial records).  grid (parts, N); 256 threads = (256 / C4) tiles side by side x C4 float4 columns.
// m: fp32 [16][N * T][C] (the GEMM's output); y: fp32 NHWC [N][H][W][C]; gn_partial: [N][parts][G][2] doubles {sum, sum of squares} or null.
// Host guarantees: C4 = C / 4 divides 256; with statistics (C / G) % 4 == 0 and G <= 256.
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ m, const float* __restrict__ bias, float* __restrict__ y,
                                                           double* __restrict__ gn_partial, int N, int H, int W, int C, int G, int tiles_per_part) {
  __shared__ double sd[2 * 256];
  const int n = blockIdx.y, part = blockIdx.x, parts = gridDim.x, tid = threadIdx.x;
  const int C4 = C >> 2, R = 256 / C4, c4 = tid % C4, tl0 = tid / C4;
  const int TW = W >> 1, T = (H >> 1) * TW;
  const long plane = (long)N * T * C;
  const int t_beg = part * tiles_per_part, t_end = min(T, t_beg + tiles_per_part);
  const float4 b4 = bias ? *reinterpret_cast<const float4*>(bias + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  double s = 0, q = 0;
  for (int t = t_beg + tl0; t < t_end; t += R) {
    const float* src = m + ((long)n * T + t) * C + c4 * 4;
    float4 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = *reinterpret_cast<const float4*>(src + (long)k * plane);
    float4 o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {   // the 4 channels of this thread
      float mm[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) mm[k] = e == 0 ? v[k].x : e == 1 ? v[k].y : e == 2 ? v[k].z : v[k].w;
      float a0[4], a1[4];   // A^T M: rows 0 and 1 over the 4 columns
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        a0[c] = (mm[c] + mm[4 + c]) + mm[8 + c];
        a1[c] = (mm[4 + c] - mm[8 + c]) - mm[12 + c];
      }
      const float be = e == 0 ? b4.x : e == 1 ? b4.y : e == 2 ? b4.z : b4.w;
      const float y00 = ((a0[0] + a0[1]) + a0[2]) + be, y01 = ((a0[1] - a0[2]) - a0[3]) + be;
      const float y10 = ((a1[0] + a1[1]) + a1[2]) + be, y11 = ((a1[1] - a1[2]) - a1[3]) + be;
      if (e == 0) { o[0].x = y00; o[1].x = y01; o[2].x = y10; o[3].x = y11; }
      if (e == 1) { o[0].y = y00; o[1].y = y01; o[2].y = y10; o[3].y = y11; }
      if (e == 2) { o[0].z = y00; o[1].z = y01; o[2].z = y10; o[3].z = y11; }
      if (e == 3) { o[0].w = y00; o[1].w = y01; o[2].w = y10; o[3].w = y11; }
    }
    const int ty = t / TW, tx = t - ty * TW;
    float* dst = y + (((long)n * H + 2 * ty) * W + 2 * tx) * C + c4 * 4;
    *reinterpret_cast<float4*>(dst) = o[0];
    *reinterpret_cast<float4*>(dst + C) = o[1];
    *reinterpret_cast<float4*>(dst + (long)W * C) = o[2];
    *reinterpret_cast<float4*>(dst + (long)W * C + C) = o[3];
    if (gn_partial) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s1 += (o[j].x + o[j].y) + (o[j].z + o[j].w);
        s2 = fmaf(o[j].x, o[j].x, s2); s2 = fmaf(o[j].y, o[j].y, s2); s2 = fmaf(o[j].z, o[j].z, s2); s2 = fmaf(o[j].w, o[j].w, s2);
      }
      s += (double)s1;
      q += (double)s2;
    }
  }
  if (gn_partial) {   // (uniform branch) group g = columns [g cq, (g + 1) cq) of every tile lane, added in a fixed order
    sd[tid] = s;
    sd[256 + tid] = q;
    __syncthreads();
    if (tid < G) {
      const int cq = (C / G) >> 2;
      double a = 0, b = 0;
      for (int tl = 0; tl < R; ++tl)
        for (int k = 0; k < cq; ++k) {
          a += sd[tl * C4 + tid * cq + k];
          b += sd[256 + tl * C4 + tid * cq + k];
        }
      double* o = gn_partial + (((long)n * parts + part) * G + tid) * 2;
      o[0] = a;
      o[1] = b;
    }
  }
}

// ---- the tail of a Winograd convolution that is followed by a GroupNorm (round 5): output transform + bias, the GroupNorm of the WHOLE group,
// Swish, + residual, + embedding row, the fp32 / fp16-pair outputs AND (optionally) the input transform of the NEXT Winograd convolution, in ONE
// launch -- what wino_output_kernel + gn_apply_part_kernel (groupnorm.hip) + wino_input_kernel do in three, without the round trips of the
// un-normalised y and of the normalised tensor in between (BasicBlock.forward / BasicResBlock.forward, conv_blocks.py:185-191,236-240, with the
// `x += emb` of :360-363).  One workgroup per (sample, group): all pixels of the sample x the group's channels sit in LDS (H W cpg floats), so
// the statistics are complete inside the workgroup -- no records, no second pass.
struct WinoTailP {
  const float* m;               // GEMM output [16][N T][C] fp32
  const float* bias;            // [C] or null
  const float* gamma;           // [C] or null (both)
  const float* beta;
  const float* res_f32;         // residual as fp32 NHWC, or
  const void* res_pairs;        // as fp16 pairs (scaled by res_bound), or neither
  const float* res_bound;       // [N] bound of |residual|, or
  const float* res_slots;       // [N][res_nslots] the slot maxima its convolution left
  int res_nslots;
  const float* emb;             // [N][emb_stride] embedding rows added per (n, c), or null
  long emb_stride;
  const float* emb_bound;       // [N]
  float* out_f32;               // fp32 NHWC result, or null
  void* out_pairs;              // fp16-pair result, or null
  float* out_bound;             // [N] bound the pairs (and V) were scaled with: bconst + bound(residual) + bound(embedding row)
  void* out_wino;               // V = B^T d B of the RESULT (as wino_input_kernel would make it from out_pairs), or null
  float* wino_bound;            // [16 N]
  int N, H, W, C, G, act;
  float eps, bconst;
  int v_nt;                     // 1: the transform-domain output goes out with non-temporal stores (MF_WINO_VSTORE, A/B)
  int v_f32;                    // 1: fp32 mode (the exact arithmetics, round 6): out_wino is fp32 [16][N][T][C], no pair rounding, no scales; out_pairs null
  int xcd_map;                  // 1: consecutive groups of a sample run on ONE XCD (round 6, MF_WINO_TAIL_MAP): at 16 channels per group two groups share
                                // every 128-byte line of M / V / the pair outputs -- with the plain (g, n) grid they sat on different XCDs (different L2s)
};

// the value a pair (hi, lo') stands for, still scaled: RN16(a) + RN16((a - RN16(a)) 2048) / 2048 -- what wino_load_pairs4 reads back
__device__ __forceinline__ float wino_pair_round(float a) {
  const _Float16 h = (_Float16)a;
  const _Float16 l = (_Float16)((a - (float)h) * kLoScale);
  return (float)h + (float)l * kLoInv;
}

__global__ __launch_bounds__(256) void wino_tail_kernel(const WinoTailP p) {
  extern __shared__ __attribute__((aligned(16))) float ys[];   // [H W][cpg]
  __shared__ double sd[8];
  __shared__ float sf[8];
  // grid: G N workgroups, one per (sample, group).  Workgroup b runs on XCD b % 8: with xcd_map the logical index is the bijective re-numbering that
  // gives every XCD a contiguous range of (n, g) pairs -- neighbouring groups of one sample, which share cache lines, share an L2
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int lin = blockIdx.x;
  if (p.xcd_map) {
    const int total = p.G * p.N, q = total >> 3, r = total & 7, xcd = lin & 7, within = lin >> 3;
    lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int g = lin % p.G, n = lin / p.G;
  const int C = p.C, cpg = C / p.G, cq = cpg >> 2, c0 = g * cpg;
  const int H = p.H, W = p.W, HW = H * W, TW = W >> 1, T = (H >> 1) * TW;
  const long plane = (long)p.N * T * C;
  const int c4 = tid % cq;   // (host: 256 % cq == 0 -- a thread keeps its 4 channels through every loop)
  // the residual rows of this thread's first RJ elements of phase 2 are requested NOW: they depend on nothing, and their round trip hides behind
  // phase 1 instead of standing between the statistics and the stores (the tail is a chain of dependent memory round trips: every one removed
  // is ~1.5 us of a 14 us launch at the 8 x 8 level)
  constexpr int RJ = 4;
  uint2 rph[RJ], rpl[RJ];
  float4 rpf[RJ];
#pragma unroll
  for (int j = 0; j < RJ; ++j) {
    rph[j] = make_uint2(0u, 0u); rpl[j] = make_uint2(0u, 0u); rpf[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int it = tid + 256 * j;
    if (it < HW * cq) {
      const long eo = ((long)n * HW + it / cq) * C + c0 + c4 * 4;
      if (p.res_pairs) {
        const uint2* gp = reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(p.res_pairs) + (eo >> 3) * 32 + ((eo >> 2) & 1) * 8);
        rph[j] = gp[0]; rpl[j] = gp[2];
      } else if (p.res_f32) {
        rpf[j] = *reinterpret_cast<const float4*>(p.res_f32 + eo);
      }
    }
  }
  // ---- 1. y = A^T M A + bias into LDS, statistics on the way
  const float4 b4 = p.bias ? *reinterpret_cast<const float4*>(p.bias + c0 + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  double s = 0, q = 0;
  for (int it = tid; it < T * cq; it += 256) {
    const int t = it / cq;
    const float* src = p.m + ((long)n * T + t) * C + c0 + c4 * 4;
    float4 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = *reinterpret_cast<const float4*>(src + (long)k * plane);
    float4 o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float mm[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) mm[k] = e == 0 ? v[k].x : e == 1 ? v[k].y : e == 2 ? v[k].z : v[k].w;
      float a0[4], a1[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        a0[c] = (mm[c] + mm[4 + c]) + mm[8 + c];
        a1[c] = (mm[4 + c] - mm[8 + c]) - mm[12 + c];
      }
      const float be = e == 0 ? b4.x : e == 1 ? b4.y : e == 2 ? b4.z : b4.w;
      const float y00 = ((a0[0] + a0[1]) + a0[2]) + be, y01 = ((a0[1] - a0[2]) - a0[3]) + be;
      const float y10 = ((a1[0] + a1[1]) + a1[2]) + be, y11 = ((a1[1] - a1[2]) - a1[3]) + be;
      if (e == 0) { o[0].x = y00; o[1].x = y01; o[2].x = y10; o[3].x = y11; }
      if (e == 1) { o[0].y = y00; o[1].y = y01; o[2].y = y10; o[3].y = y11; }
      if (e == 2) { o[0].z = y00; o[1].z = y01; o[2].z = y10; o[3].z = y11; }
      if (e == 3) { o[0].w = y00; o[1].w = y01; o[2].w = y10; o[3].w = y11; }
    }
    const int ty = t / TW, tx = t - ty * TW;
    float* dst = ys + ((2 * ty) * W + 2 * tx) * cpg + c4 * 4;
    *reinterpret_cast<float4*>(dst) = o[0];
    *reinterpret_cast<float4*>(dst + cpg) = o[1];
    *reinterpret_cast<float4*>(dst + W * cpg) = o[2];
    *reinterpret_cast<float4*>(dst + W * cpg + cpg) = o[3];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s1 += (o[j].x + o[j].y) + (o[j].z + o[j].w);
      s2 = fmaf(o[j].x, o[j].x, s2); s2 = fmaf(o[j].y, o[j].y, s2); s2 = fmaf(o[j].z, o[j].z, s2); s2 = fmaf(o[j].w, o[j].w, s2);
    }
    s += (double)s1;
    q += (double)s2;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_xor(s, off, 64);
    q += __shfl_xor(q, off, 64);
  }
  if (lane == 0) { sd[wave] = s; sd[4 + wave] = q; }
  // the residual's measured bound straight from the slots its convolution wrote (max: any order gives the same bits)
  if (p.res_slots) {
    float mx = 0.f;
    for (int i = tid; i < p.res_nslots; i += 256) mx = fmaxf(mx, p.res_slots[(long)n * p.res_nslots + i]);
    mx = wave_max(mx);
    if (lane == 0) sf[wave] = mx;
  }
  __syncthreads();
  const double S = ((sd[0] + sd[1]) + sd[2]) + sd[3], Q = ((sd[4] + sd[5]) + sd[6]) + sd[7];
  const double count = (double)HW * cpg;
  const double mean_d = S / count;
  double var = Q / count - mean_d * mean_d;
  if (var < 0) var = 0;
  const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(var + (double)p.eps));
  const float rb = p.res_slots ? fmaxf(fmaxf(sf[0], sf[1]), fmaxf(sf[2], sf[3])) : (p.res_bound ? p.res_bound[n] : 0.f);
  const float ob = p.bconst + rb + (p.emb_bound ? p.emb_bound[n] : 0.f);   // (the order of mf_gn_apply_*: bconst + residual + embedding)
  const float osc = exp2i(-scale_exp_of(ob));
  const float rsc = p.res_pairs ? exp2i(scale_exp_of(p.res_bound[n])) : 1.f;
  const float vb = 4.f * ob;
  if (g == 0) {
    if (tid == 0 && p.out_bound) p.out_bound[n] = ob;
    if (tid < 16 && p.wino_bound) p.wino_bound[tid * p.N + n] = vb;
  }
  // ---- 2. normalise, Swish, + residual, + embedding: the per-element arithmetic of gn_apply_part_kernel, same operations in the same order
  float ga[4] = {1.f, 1.f, 1.f, 1.f}, be[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.gamma) {
    const float4 g4 = *reinterpret_cast<const float4*>(p.gamma + c0 + c4 * 4), e4 = *reinterpret_cast<const float4*>(p.beta + c0 + c4 * 4);
    ga[0] = g4.x; ga[1] = g4.y; ga[2] = g4.z; ga[3] = g4.w;
    be[0] = e4.x; be[1] = e4.y; be[2] = e4.z; be[3] = e4.w;
  }
  const float4 em = p.emb ? *reinterpret_cast<const float4*>(p.emb + (long)n * p.emb_stride + c0 + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  const bool keep = p.out_wino != nullptr;
  auto pairs_to_f4 = [](uint2 h, uint2 l) {
    const sf_f16x2 h0 = __builtin_bit_cast(sf_f16x2, h.x), h1 = __builtin_bit_cast(sf_f16x2, h.y), l0 = __builtin_bit_cast(sf_f16x2, l.x),
                   l1 = __builtin_bit_cast(sf_f16x2, l.y);
    return make_float4((float)h0[0] + (float)l0[0] * kLoInv, (float)h0[1] + (float)l0[1] * kLoInv, (float)h1[0] + (float)l1[0] * kLoInv,
                       (float)h1[1] + (float)l1[1] * kLoInv);
  };
#define WINO_TAIL_ELEMENT(IT, RES_EXPR_PAIRS, RES_EXPR_F32)                                                            \
  {                                                                                                                   \
    const int pix = (IT) / cq;                                                                                        \
    float* yp = ys + pix * cpg + c4 * 4;                                                                              \
    const float4 v = *reinterpret_cast<const float4*>(yp);                                                            \
    float e[4] = {v.x, v.y, v.z, v.w};                                                                                \
    _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                                   \
      float t = (e[k] - mean) * rstd;                                                                                 \
      if (p.gamma) t = t * ga[k] + be[k];                                                                             \
      if (p.act == 1) t = swish_apply(t);                                                                             \
      e[k] = t;                                                                                                       \
    }                                                                                                                 \
    const long eo = ((long)n * HW + pix) * C + c0 + c4 * 4;                                                           \
    if (p.res_pairs) {                                                                                                \
      const float4 r = RES_EXPR_PAIRS;                                                                                \
      e[0] += r.x * rsc; e[1] += r.y * rsc; e[2] += r.z * rsc; e[3] += r.w * rsc;                                     \
    } else if (p.res_f32) {                                                                                           \
      const float4 r = RES_EXPR_F32;                                                                                  \
      e[0] += r.x; e[1] += r.y; e[2] += r.z; e[3] += r.w;                                                             \
    }                                                                                                                 \
    if (p.emb) { e[0] += em.x; e[1] += em.y; e[2] += em.z; e[3] += em.w; }                                            \
    if (p.out_f32) *reinterpret_cast<float4*>(p.out_f32 + eo) = make_float4(e[0], e[1], e[2], e[3]);                  \
    if (p.out_pairs) store_split4<false>(p.out_pairs, eo, e[0], e[1], e[2], e[3], osc);   /* (the bound is derived: no clamp, split_f16.h) */ \
    if (keep) *reinterpret_cast<float4*>(yp) = p.v_f32 ? make_float4(e[0], e[1], e[2], e[3])                                                                     \
                                                       : make_float4(wino_pair_round(e[0] * osc), wino_pair_round(e[1] * osc), wino_pair_round(e[2] * osc), wino_pair_round(e[3] * osc)); \
  }
#pragma unroll
  for (int j = 0; j < RJ; ++j) {
    const int it = tid + 256 * j;
    if (it < HW * cq) WINO_TAIL_ELEMENT(it, pairs_to_f4(rph[j], rpl[j]), rpf[j])
  }
  for (int it = tid + 256 * RJ; it < HW * cq; it += 256) WINO_TAIL_ELEMENT(it, wino_load_pairs4(p.res_pairs, eo), *reinterpret_cast<const float4*>(p.res_f32 + eo))
#undef WINO_TAIL_ELEMENT
  if (!keep) return;
  // ---- 3. the input transform of the next Winograd convolution, from the pair-rounded result in LDS: bit for bit what wino_input_kernel
  // makes of out_pairs
  __syncthreads();
  const float f = exp2i(scale_exp_of(ob)) * exp2i(-scale_exp_of(vb));
  for (int it = tid; it < T * cq; it += 256) {
    const int t = it / cq, ty = t / TW, tx = t - ty * TW;
    float d[4][16];
#pragma unroll
    for (int py = 0; py < 4; ++py)
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        const int y = 2 * ty - 1 + py, x = 2 * tx - 1 + px;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) v = *reinterpret_cast<const float4*>(ys + (y * W + x) * cpg + c4 * 4);
        d[0][4 * py + px] = v.x; d[1][4 * py + px] = v.y; d[2][4 * py + px] = v.z; d[3][4 * py + px] = v.w;
      }
#pragma unroll
    for (int k = 0; k < 4; ++k) wino_bt_d_b(d[k]);
    const long e0 = ((long)n * T + t) * C + c0 + c4 * 4;
    if (p.v_f32) {
#pragma unroll
      for (int k = 0; k < 16; ++k)
        __builtin_nontemporal_store(wf32x4{d[0][k], d[1][k], d[2][k], d[3][k]}, reinterpret_cast<wf32x4*>(reinterpret_cast<float*>(p.out_wino) + (long)k * plane + e0));
    } else if (p.v_nt) {
#pragma unroll
      for (int k = 0; k < 16; ++k) store_split4<true, true>(p.out_wino, (long)k * plane + e0, d[0][k], d[1][k], d[2][k], d[3][k], f);
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) store_split4<true>(p.out_wino, (long)k * plane + e0, d[0][k], d[1][k], d[2][k], d[3][k], f);
    }
  }
}

}  // namespace mfw
