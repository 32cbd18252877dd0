// gn_partial.h -- per-(sample, pixel-chunk, group) GroupNorm partial sums, shared by groupnorm.hip (stand-alone statistics
// pass) and conv.hip (fused into the split-K reduction: the reducer already streams every output element once).
#pragma once
#include "common.h"
#include "split_f16.h"

namespace mf {

constexpr int kStatsThreads = 256;
constexpr int kMaxChunks = 16;   // P: every consumer workgroup re-reads P x G records, keep it small

__host__ __device__ inline int stats_chunks(int HW) {
  // >= 4 pixels per chunk, at most kMaxChunks chunks per sample: the pass is latency-bound, it needs workgroups
  int c = (HW + 3) / 4;
  return c < 1 ? 1 : (c > kMaxChunks ? kMaxChunks : c);
}

// channel slices (grid.z): more workgroups without more partial records; a slice must hold whole groups and float4s
inline int stats_slices(int N, int HW, int C, int G) {
  const int cpg = C / G;
  int sl = 1;
  while (sl < 8 && (long)N * stats_chunks(HW) * sl < 512 && (C / (sl * 2)) % cpg == 0 && (C / (sl * 2)) % 4 == 0 && C / (sl * 2) >= 64) sl *= 2;
  return sl;
}

inline size_t stats_lds_bytes(int Cs) {  // Cs = channels per slice
  const int C4 = Cs / 4;
  const int lanes_per_row = C4 < kStatsThreads ? C4 : kStatsThreads;
  return (size_t)(kStatsThreads / lanes_per_row) * Cs * 2 * sizeof(float);
}

// Sum of the split-K slabs of one float4, in the order the in-launch reduction of the fp16-pair kernel uses (conv_f16x2.h): pairwise,
// level by level, for a power-of-two count -- so a convolution gives the same bits whether its slices met inside the launch or here --
// and first to last otherwise.
__device__ __forceinline__ float4 add4(const float4 a, const float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
template <int N_>
__device__ __forceinline__ float4 sum_slabs_tree(const float* __restrict__ p, long slab) {
  if constexpr (N_ == 1) return *reinterpret_cast<const float4*>(p);
  else return add4(sum_slabs_tree<N_ / 2>(p, slab), sum_slabs_tree<N_ / 2>(p + (long)(N_ / 2) * slab, slab));
}
__device__ __forceinline__ float4 sum_slabs(const float* __restrict__ p, long slab, int n) {
  switch (n) {
    case 1: return sum_slabs_tree<1>(p, slab);
    case 2: return sum_slabs_tree<2>(p, slab);
    case 4: return sum_slabs_tree<4>(p, slab);
    case 8: return sum_slabs_tree<8>(p, slab);
    case 16: return sum_slabs_tree<16>(p, slab);
    case 32: return sum_slabs_tree<32>(p, slab);
    default: break;
  }
  float4 v = *reinterpret_cast<const float4*>(p);
  for (int z = 1; z < n; ++z) v = add4(v, *reinterpret_cast<const float4*>(p + (long)z * slab));
  return v;
}

// grid (chunks, N).  Each block reduces its pixel range of sample n for all channels, then per group.
// partial[((n*chunks + chunk)*G + g)*2 + {0,1}] = {sum, sumsq} (double).
// REDUCE: the value is sum_z slabs[z] + bias (split-K reduction) and is also written to y.
template <bool REDUCE>
__global__ __launch_bounds__(kStatsThreads) void gn_partial_kernel(const float* __restrict__ x, double* __restrict__ partial, int HW, int C, int G,
                                                                   int nslabs, long slab, const float* __restrict__ bias, float* __restrict__ y,
                                                                   float* __restrict__ out_bound) {
  extern __shared__ __attribute__((aligned(16))) float sh[];  // [rowphases][Cs][2]
  const int chunks = gridDim.x, chunk = blockIdx.x, n = blockIdx.y;
  const int Cs = C / gridDim.z, c_off = blockIdx.z * Cs;  // this workgroup's channel slice
  const int C4 = Cs >> 2;
  const int lanes_per_row = C4 < kStatsThreads ? C4 : kStatsThreads;  // threads cooperating on one pixel row
  const int rowphases = kStatsThreads / lanes_per_row;
  const int tid = threadIdx.x;
  const int phase = tid / lanes_per_row, col = tid - phase * lanes_per_row;
  const int p_per = (HW + chunks - 1) / chunks;
  const int p0 = chunk * p_per, p1 = min(HW, p0 + p_per);
  const long nbase = (long)n * HW * C;

  float vmax = 0.f;
  for (int c4 = col; c4 < C4; c4 += lanes_per_row) {  // >1 iteration only when C > 1024
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    if (phase < rowphases) {
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (REDUCE && bias) bv = *reinterpret_cast<const float4*>(bias + c_off + c4 * 4);
      for (int px = p0 + phase; px < p1; px += rowphases) {
        const long e = nbase + (long)px * C + c_off + c4 * 4;
        float4 v = REDUCE ? sum_slabs(x + e, slab, nslabs) : *reinterpret_cast<const float4*>(x + e);
        if (REDUCE) {
          v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
          *reinterpret_cast<float4*>(y + e) = v;
          vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
        s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
        q0 = fmaf(v.x, v.x, q0); q1 = fmaf(v.y, v.y, q1); q2 = fmaf(v.z, v.z, q2); q3 = fmaf(v.w, v.w, q3);
      }
      float* d = sh + ((long)phase * Cs + c4 * 4) * 2;
      d[0] = s0; d[1] = q0; d[2] = s1; d[3] = q1; d[4] = s2; d[5] = q2; d[6] = s3; d[7] = q3;
    }
  }
  if (REDUCE && out_bound) {  // measured max of |y|: slot (chunk, slice, wave) of sample n (operand scale of a following fp16-pair convolution)
    vmax = wave_max(vmax);
    const int slots = gridDim.x * gridDim.z * (kStatsThreads / 64);
    if ((tid & 63) == 0) out_bound[(long)n * slots + (chunk * gridDim.z + blockIdx.z) * (kStatsThreads / 64) + (tid >> 6)] = vmax;
  }
  __syncthreads();
  const int cpg = C / G, gs = Cs / cpg;  // groups in this slice
  for (int g = tid; g < gs; g += kStatsThreads) {
    double s = 0, q = 0;
    for (int ph = 0; ph < rowphases; ++ph) {
      const float* d = sh + ((long)ph * Cs + g * cpg) * 2;
      for (int c = 0; c < cpg; ++c) { s += (double)d[2 * c]; q += (double)d[2 * c + 1]; }
    }
    double* o = partial + (((long)n * chunks + chunk) * G + (c_off / cpg + g)) * 2;
    o[0] = s; o[1] = q;
  }
}

}  // namespace mf
