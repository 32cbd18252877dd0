// groupnorm.hip -- GroupNorm statistics and the fused normalise + Swish + residual + embedding pass (NHWC, fp32).
//
// Both kernels are HBM/Infinity-Cache streaming passes: fully coalesced float4 rows, wave-shuffle +
// LDS reductions.  Statistics are accumulated per thread in fp32 over <= a few hundred elements and
// combined in fp64 (a group of the VAE's 256x256 level holds 524 288 elements: E[x^2]-E[x]^2 in fp32
// would lose the parity margin -- SURVEY §7 hard parts).
#include "common.h"
#include <stdlib.h>
#include "gn_partial.h"

using namespace mf;

namespace {

// 16 lanes per (sample, group): each lane sums every 16th partial record, a 16-lane butterfly combines them (one memory latency
// instead of `chunks` of them: the records were just written by another kernel, so every load is an L2/HBM round trip)
__global__ __launch_bounds__(256) void gn_stats_final_kernel(const double* __restrict__ partial, float* __restrict__ stats, int NG, int G, int chunks,
                                                              double count, float eps) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = t >> 4, l = t & 15;
  double s = 0, q = 0;
  if (i < NG) {
    const int n = i / G, g = i - n * G;
    for (int c = l; c < chunks; c += 16) {
      const double* p = partial + (((long)n * chunks + c) * G + g) * 2;
      s += p[0]; q += p[1];
    }
  }
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) {
    s += __shfl_xor(s, off, 64);
    q += __shfl_xor(q, off, 64);
  }
  if (i >= NG || l != 0) return;
  const double mean = s / count;
  double var = q / count - mean * mean;
  if (var < 0) var = 0;
  stats[2 * i] = (float)mean;
  stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// fp16-pair output of the apply pass (MF_CONV_FP32_F16X2 operand): the per-sample scale comes from an upper bound of |out| that the
// pass derives from its inputs -- |act(gn(x) gamma + beta)| <= bconst = max|gamma| sqrt(group size) + max|beta| (host constant), or
// x_bound[n] when nothing is normalised, plus the bounds of the residual and of the embedding row -- and publishes as out_bound[n].
struct GnSplit {
  void* outs;               // fp16-pair copy of `out`, or null
  const float* x_bound;     // [N] bound of |x| (used when stats == null), or null
  const float* res_bound;   // [N] bound of |residual|, or null
  const float* emb_bound;   // [N] bound of |emb[n][:]|, or null
  float bconst;
  float* out_bound;         // [N] written by the pass
  const float* res_slots;   // (from-partials pass) the residual's bound still as the [N][res_nslots] slot maxima its convolution wrote, or null
  int res_nslots;
  const void* res_pairs;    // (from-partials pass) the residual exists ONLY in its fp16-pair form (scaled by res_bound): read from there
};

// 4 consecutive channels starting at element index e (a multiple of 4) of a tensor stored as fp16 pairs, back as fp32: hi + lo' / 2048 is
// exact in fp32 (23 significant bits), the power of two `sc` (2^s of the sample) too: the value is the original fp32 number with its last
// significand bit cleared at most (split_f16.h)
__device__ __forceinline__ float4 load_pairs4(const void* pairs, long e, float sc) {
  const uint2* g = reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(pairs) + (e >> 3) * 32 + ((e >> 2) & 1) * 8);
  const uint2 h = g[0], l = g[2];   // [hi x 8] then [lo' x 8]: 16 bytes apart
  const sf_f16x2 h0 = __builtin_bit_cast(sf_f16x2, h.x), h1 = __builtin_bit_cast(sf_f16x2, h.y), l0 = __builtin_bit_cast(sf_f16x2, l.x),
                 l1 = __builtin_bit_cast(sf_f16x2, l.y);
  return make_float4(((float)h0[0] + (float)l0[0] * kLoInv) * sc, ((float)h0[1] + (float)l0[1] * kLoInv) * sc,
                     ((float)h1[0] + (float)l1[0] * kLoInv) * sc, ((float)h1[1] + (float)l1[1] * kLoInv) * sc);
}

// out = act(gn(x)*gamma + beta) + residual + emb[n][c]
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ residual,
                                                        const float* __restrict__ emb, long emb_stride, float* __restrict__ out, long total4,
                                                        int HW, int C, int G, int act) {
  const int C4 = C >> 2;
  const int cpg = C / G;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
    const int c4 = (int)(i % C4);
    const long pix = i / C4;
    const int n = (int)(pix / HW);
    const int c = c4 * 4;
    float4 v = *reinterpret_cast<const float4*>(x + i * 4);
    float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float t = e[k];
      if (stats) {
        const int g = (c + k) / cpg;
        const float mean = stats[((long)n * G + g) * 2], rstd = stats[((long)n * G + g) * 2 + 1];
        t = (t - mean) * rstd;
        if (gamma) t = t * gamma[c + k] + beta[c + k];
      }
      if (act == 1) t = swish_acc(t);
      e[k] = t;
    }
    if (residual) {
      const float4 r = *reinterpret_cast<const float4*>(residual + i * 4);
      e[0] += r.x; e[1] += r.y; e[2] += r.z; e[3] += r.w;
    }
    if (emb) {
      const float4 m = *reinterpret_cast<const float4*>(emb + (long)n * emb_stride + c);
      e[0] += m.x; e[1] += m.y; e[2] += m.z; e[3] += m.w;
    }
    *reinterpret_cast<float4*>(out + i * 4) = make_float4(e[0], e[1], e[2], e[3]);
  }
}

// The same pass with the fp16-pair copy (4 channels = half a pair group per thread: two 8-byte stores; an 8-channel form measured 25 %
// slower), the per-sample scale recomputed only when the grid-stride loop crosses into another sample.  The arithmetic per element is
// the one above (same operations, same order).
__global__ __launch_bounds__(256) void gn_apply_split_kernel(const float* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ residual,
                                                              const float* __restrict__ emb, long emb_stride, float* __restrict__ out, long total4,
                                                              int HW, int C, int G, int act, const GnSplit sp) {
  const int C4 = C >> 2;
  const int cpg = C / G;
  const long per4 = (long)HW * C4;
  const long stride = (long)gridDim.x * blockDim.x;
  int cur_n = -1;
  float sc = 1.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
    const int c4 = (int)(i % C4);
    const long pix = i / C4;
    const int n = (int)(pix / HW);
    const int c = c4 * 4;
    if (n != cur_n) {
      const float b = (stats || !sp.x_bound ? sp.bconst : sp.x_bound[n]) + (sp.res_bound ? sp.res_bound[n] : 0.f) + (sp.emb_bound ? sp.emb_bound[n] : 0.f);
      sc = exp2i(-scale_exp_of(b));
      cur_n = n;
      if (i == (long)n * per4) sp.out_bound[n] = b;
    }
    float4 v = *reinterpret_cast<const float4*>(x + i * 4);
    float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float t = e[k];
      if (stats) {
        const int g = (c + k) / cpg;
        const float mean = stats[((long)n * G + g) * 2], rstd = stats[((long)n * G + g) * 2 + 1];
        t = (t - mean) * rstd;
        if (gamma) t = t * gamma[c + k] + beta[c + k];
      }
      if (act == 1) t = swish_acc(t);
      e[k] = t;
    }
    if (residual) {
      const float4 r = *reinterpret_cast<const float4*>(residual + i * 4);
      e[0] += r.x; e[1] += r.y; e[2] += r.z; e[3] += r.w;
    }
    if (emb) {
      const float4 m = *reinterpret_cast<const float4*>(emb + (long)n * emb_stride + c);
      e[0] += m.x; e[1] += m.y; e[2] += m.z; e[3] += m.w;
    }
    *reinterpret_cast<float4*>(out + i * 4) = make_float4(e[0], e[1], e[2], e[3]);
    store_split4(sp.outs, i * 4, e[0], e[1], e[2], e[3], sc);
  }
}

// The pass fed by the PARTIAL records of the producing convolution -- no finalize launch in front of it.  grid (blocks per sample, N): a
// block stays inside one sample and handles U float4 per thread and round (thread t: float4 jw + 256 u + t); it issues the loads of its
// first round, and while they are in flight reduces the sample's records (all 256 threads: 256 / G records per group side by side, fp64,
// the finalize kernel's formula) into LDS.  The pass is VALU-bound, not HBM-bound, unless the per-element work is kept to the arithmetic
// itself (r03: 560 -> ~200 instructions per float4): 32-bit indexing inside the sample, and -- when 256 % (C / 4) == 0, every published
// width -- a thread's 4 channels are the same in every round, so mean / rstd / gamma / beta / embedding sit in registers.
struct GnChan {
  float mean[4], rstd[4], ga[4], be[4];
  float4 em;
};

__device__ __forceinline__ GnChan gn_chan_consts(const float* sm, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                 const float* __restrict__ emb_row, int c, int cpg) {
  GnChan k;
  if (cpg % 4 == 0) {   // (c is a multiple of 4: one group)
    const int g = c / cpg;
#pragma unroll
    for (int i = 0; i < 4; ++i) { k.mean[i] = sm[2 * g]; k.rstd[i] = sm[2 * g + 1]; }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int g = (c + i) / cpg; k.mean[i] = sm[2 * g]; k.rstd[i] = sm[2 * g + 1]; }
  }
  if (gamma) {
    const float4 ga = *reinterpret_cast<const float4*>(gamma + c), be = *reinterpret_cast<const float4*>(beta + c);
    k.ga[0] = ga.x; k.ga[1] = ga.y; k.ga[2] = ga.z; k.ga[3] = ga.w;
    k.be[0] = be.x; k.be[1] = be.y; k.be[2] = be.z; k.be[3] = be.w;
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) { k.ga[i] = 1.f; k.be[i] = 0.f; }
  }
  k.em = emb_row ? *reinterpret_cast<const float4*>(emb_row + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  return k;
}

template <bool SPLIT, int U>
__global__ __launch_bounds__(256) void gn_apply_part_kernel(const float* __restrict__ x, const double* __restrict__ partial, int parts, double count, float eps,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ residual, const float* __restrict__ emb, long emb_stride,
                                                            float* __restrict__ out, int HW, int C, int G, int act, const GnSplit sp) {
  __shared__ float sm[2 * 256 + 4];
  __shared__ double sd[2 * 256];
  const int n = blockIdx.y, tid = threadIdx.x;
  const int C4 = C >> 2, cpg = C / G;
  const unsigned per4 = (unsigned)HW * (unsigned)C4;   // (host: < 2^31)
  const long base = (long)n * per4;
  constexpr unsigned SPAN = 256u * U;
  const unsigned step = gridDim.x * SPAN;
  const bool res_p = SPLIT && sp.res_pairs != nullptr;                                    // the residual lives as fp16 pairs only
  const bool res_f = !res_p && residual != nullptr;
  const bool fixed_c = (256 % C4) == 0;
  const float* emb_row = emb ? emb + (long)n * emb_stride : nullptr;

  float4 v[U], r[U];
  uint2 rh[U], rl[U];
  unsigned jw = blockIdx.x * SPAN;
#define GN_LOAD_ROUND()                                                                                                  \
  _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                                        \
    const unsigned j = jw + 256u * u + tid;                                                                              \
    if (j < per4) {                                                                                                      \
      v[u] = *reinterpret_cast<const float4*>(x + (base + j) * 4);                                                       \
      if (res_p) {                                                                                                       \
        const long e = (base + j) * 4;                                                                                   \
        const uint2* g_ = reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(sp.res_pairs) + (e >> 3) * 32 + ((e >> 2) & 1) * 8); \
        rh[u] = g_[0]; rl[u] = g_[2];   /* [hi x 8] then [lo' x 8]: 16 bytes apart (load_pairs4) */                         \
      } else if (res_f) r[u] = *reinterpret_cast<const float4*>(residual + (base + j) * 4);                              \
    }                                                                                                                    \
  }
  if (jw < per4) GN_LOAD_ROUND()

  {   // the sample's statistics: thread (kk, g) adds records kk, kk + PPT, ...; the first G threads add the PPT sums of their group
    const int PPT = 256 / G, g = tid % G, kk = tid / G;
    double s = 0, q = 0;
    if (kk < PPT)
      for (int k = kk; k < parts; k += PPT) {
        const double* p = partial + (((long)n * parts + k) * G + g) * 2;
        s += p[0]; q += p[1];
      }
    sd[tid] = s; sd[256 + tid] = q;
  }
  if (SPLIT && sp.res_slots) {   // the residual's measured bound, straight from the slots its convolution wrote (no finalize launch)
    float m = 0.f;
    for (int i = tid; i < sp.res_nslots; i += 256) m = fmaxf(m, sp.res_slots[(long)n * sp.res_nslots + i]);
    m = wave_max(m);
    if ((tid & 63) == 0) sm[2 * 256 + (tid >> 6)] = m;
  }
  __syncthreads();
  if (tid < G) {
    const int PPT = 256 / G;
    double s = sd[tid], q = sd[256 + tid];
    for (int kk = 1; kk < PPT; ++kk) { s += sd[kk * G + tid]; q += sd[256 + kk * G + tid]; }
    const double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0) var = 0;
    sm[2 * tid] = (float)mean;
    sm[2 * tid + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  float sc = 1.f, rsc = 1.f;
  if (SPLIT) {
    const float rb = sp.res_slots ? fmaxf(fmaxf(sm[512], sm[513]), fmaxf(sm[514], sm[515])) : (sp.res_bound ? sp.res_bound[n] : 0.f);
    const float b = sp.bconst + rb + (sp.emb_bound ? sp.emb_bound[n] : 0.f);
    sc = exp2i(-scale_exp_of(b));
    if (res_p) rsc = exp2i(scale_exp_of(sp.res_bound[n]));
    if (blockIdx.x == 0 && tid == 0) sp.out_bound[n] = b;
  }
  GnChan kc = gn_chan_consts(sm, gamma, beta, emb_row, fixed_c ? (tid % C4) * 4 : 0, cpg);

  while (jw < per4) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned j = jw + 256u * u + tid;
      if (j < per4) {
        if (!fixed_c) kc = gn_chan_consts(sm, gamma, beta, emb_row, (int)(j % (unsigned)C4) * 4, cpg);
        float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float t = (e[k] - kc.mean[k]) * kc.rstd[k];
          if (gamma) t = t * kc.ga[k] + kc.be[k];
          if (act == 1) t = swish_apply(t);
          e[k] = t;
        }
        if (res_p) {
          const sf_f16x2 h0 = __builtin_bit_cast(sf_f16x2, rh[u].x), h1 = __builtin_bit_cast(sf_f16x2, rh[u].y),
                         l0 = __builtin_bit_cast(sf_f16x2, rl[u].x), l1 = __builtin_bit_cast(sf_f16x2, rl[u].y);
          e[0] += ((float)h0[0] + (float)l0[0] * kLoInv) * rsc; e[1] += ((float)h0[1] + (float)l0[1] * kLoInv) * rsc;
          e[2] += ((float)h1[0] + (float)l1[0] * kLoInv) * rsc; e[3] += ((float)h1[1] + (float)l1[1] * kLoInv) * rsc;
        } else if (res_f) { e[0] += r[u].x; e[1] += r[u].y; e[2] += r[u].z; e[3] += r[u].w; }
        if (emb) { e[0] += kc.em.x; e[1] += kc.em.y; e[2] += kc.em.z; e[3] += kc.em.w; }
        if (!SPLIT || out) *reinterpret_cast<float4*>(out + (base + j) * 4) = make_float4(e[0], e[1], e[2], e[3]);   // (null: pairs only)
        if (SPLIT) store_split4<false>(sp.outs, (base + j) * 4, e[0], e[1], e[2], e[3], sc);   // (the bound is derived: no clamp, split_f16.h)
      }
    }
    jw += step;
    if (jw < per4) GN_LOAD_ROUND()
  }
#undef GN_LOAD_ROUND
}

// max of |x| over this block's share of sample n -> partial[n][blockIdx.x * 4 + wave].  grid (blocks, N)
__global__ __launch_bounds__(256) void maxabs_kernel(const float* __restrict__ x, float* __restrict__ partial, long per_sample4) {
  const int n = blockIdx.y;
  const float4* p = reinterpret_cast<const float4*>(x) + (long)n * per_sample4;
  float m = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per_sample4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = p[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) partial[((long)n * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)] = m;
}

// bound[n] = max over the slots of sample n.  One wave per sample.
__global__ __launch_bounds__(64) void bound_finalize_kernel(const float* __restrict__ partial, float* __restrict__ bound, int slots) {
  const int n = blockIdx.x;
  float m = 0.f;
  for (int i = threadIdx.x; i < slots; i += 64) m = fmaxf(m, partial[(long)n * slots + i]);
  m = wave_max(m);
  if (threadIdx.x == 0) bound[n] = m;
}

}  // namespace

extern "C" {

size_t mf_gn_stats_workspace_bytes(int N, int HW, int C, int G) {
  if (N <= 0 || HW <= 0 || G <= 0) return 0;
  return (size_t)N * stats_chunks(HW) * G * 2 * sizeof(double);
}

int mf_gn_stats_f32(const float* x, float* stats, void* workspace, size_t workspace_bytes, int N, int HW, int C, int G, float eps,
                    void* stream) {
  MF_REQUIRE(x && stats && N > 0 && HW > 0 && C > 0 && G > 0, MF_EINVAL, "gn_stats: bad args");
  MF_REQUIRE(C % 4 == 0 && C % G == 0, MF_EUNSUPPORTED, "gn_stats: C=%d G=%d unsupported (need C%%4==0, C%%G==0)", C, G);
  const size_t need = mf_gn_stats_workspace_bytes(N, HW, C, G);
  MF_REQUIRE(workspace && workspace_bytes >= need, MF_EWORKSPACE, "gn_stats: workspace %zu < %zu", workspace_bytes, need);
  hipStream_t s = (hipStream_t)stream;
  const int chunks = stats_chunks(HW), slices = stats_slices(N, HW, C, G);
  const size_t lds = stats_lds_bytes(C / slices);
  MF_REQUIRE(lds <= 64 * 1024, MF_EUNSUPPORTED, "gn_stats: C=%d too wide", C);
  ProfScope ps(MF_FAM_GN_STATS, s, 3.0 * N * HW * C, 4.0 * N * (double)HW * C);
  MF_LAUNCH(gn_partial_kernel<false>, dim3(chunks, N, slices), dim3(kStatsThreads), lds, s, x, reinterpret_cast<double*>(workspace), HW, C, G, 1, 0L,
                     (const float*)nullptr, (float*)nullptr, (float*)nullptr);
  int rc = check_launch("gn_stats_partial");
  if (rc) return rc;
  const int NG = N * G;
  MF_LAUNCH(gn_stats_final_kernel, dim3((NG * 16 + 255) / 256), dim3(256), 0, s, reinterpret_cast<const double*>(workspace), stats, NG, G,
                     chunks, (double)HW * (C / G), eps);
  return check_launch("gn_stats_final");
}

int mf_gn_apply_f32(const float* x, const float* stats, const float* gamma, const float* beta, const float* residual, const float* emb,
                    int64_t emb_stride, float* out, int N, int HW, int C, int G, int act, void* stream) {
  return mf_gn_apply_split_f32(x, stats, gamma, beta, residual, emb, emb_stride, out, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, N, HW, C, G, act,
                               stream);
}

int mf_gn_apply_split_f32(const float* x, const float* stats, const float* gamma, const float* beta, const float* residual, const float* emb,
                          int64_t emb_stride, float* out, void* out_split, const float* x_bound, const float* res_bound, const float* emb_bound,
                          float bconst, float* out_bound, int N, int HW, int C, int G, int act, void* stream) {
  MF_REQUIRE(x && out && N > 0 && HW > 0 && C > 0, MF_EINVAL, "gn_apply: bad args");
  MF_REQUIRE(!out_split || (C % 8 == 0 && out_bound), MF_EUNSUPPORTED, "gn_apply: the fp16-pair output needs C %% 8 == 0 and out_bound");
  MF_REQUIRE(!out_split || stats || x_bound, MF_EINVAL, "gn_apply: the fp16-pair output of an un-normalised pass needs x_bound");
  MF_REQUIRE(!out_split || !residual || res_bound, MF_EINVAL, "gn_apply: the fp16-pair output needs res_bound with a residual");
  MF_REQUIRE(!out_split || !emb || emb_bound, MF_EINVAL, "gn_apply: the fp16-pair output needs emb_bound with an embedding");
  MF_REQUIRE(C % 4 == 0, MF_EUNSUPPORTED, "gn_apply: C=%d must be a multiple of 4", C);
  MF_REQUIRE(!stats || (G > 0 && C % G == 0), MF_EINVAL, "gn_apply: C=%d G=%d", C, G);
  MF_REQUIRE((gamma == nullptr) == (beta == nullptr), MF_EINVAL, "gn_apply: gamma/beta must both be given or both NULL");
  MF_REQUIRE(!emb || emb_stride % 4 == 0, MF_EUNSUPPORTED, "gn_apply: emb_stride must be a multiple of 4");
  hipStream_t s = (hipStream_t)stream;
  const long total4 = (long)N * HW * (C / 4);
  const double nelem = (double)N * HW * C;
  ProfScope ps(MF_FAM_GN_APPLY, s, 8.0 * nelem, 4.0 * nelem * (2 + (residual ? 1 : 0) + (out_split ? 1 : 0)));
  if (out_split) {
    long blocks = (total4 + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    const GnSplit sp{out_split, x_bound, res_bound, emb_bound, bconst, out_bound, nullptr, 0, nullptr};
    MF_LAUNCH(gn_apply_split_kernel, dim3((int)blocks), dim3(256), 0, s, x, stats, gamma, beta, residual, emb, (long)emb_stride, out, total4, HW,
                       C, G > 0 ? G : 1, act, sp);
    return check_launch("gn_apply_split");
  }
  long blocks = (total4 + 255) / 256;
  if (blocks > 256 * 8) blocks = 256 * 8;
  MF_LAUNCH(gn_apply_kernel, dim3((int)blocks), dim3(256), 0, s, x, stats, gamma, beta, residual, emb, (long)emb_stride, out, total4, HW, C,
                     G > 0 ? G : 1, act);
  return check_launch("gn_apply");
}

int mf_gn_apply_from_partials_f32(const float* x, const double* gn_partial, int parts, float eps, const float* gamma, const float* beta,
                                  const float* residual, const float* emb, int64_t emb_stride, float* out, void* out_split, const float* res_bound,
                                  const float* res_bound_slots, int res_nslots, const float* emb_bound, float bconst, float* out_bound, int N, int HW,
                                  int C, int G, int act, void* stream) {
  return mf_gn_apply_from_partials_pairs_f32(x, gn_partial, parts, eps, gamma, beta, residual, nullptr, emb, emb_stride, out, out_split, res_bound,
                                             res_bound_slots, res_nslots, emb_bound, bconst, out_bound, N, HW, C, G, act, stream);
}

int mf_gn_apply_from_partials_pairs_f32(const float* x, const double* gn_partial, int parts, float eps, const float* gamma, const float* beta,
                                        const float* residual, const void* residual_pairs, const float* emb, int64_t emb_stride, float* out,
                                        void* out_split, const float* res_bound, const float* res_bound_slots, int res_nslots,
                                        const float* emb_bound, float bconst, float* out_bound, int N, int HW, int C, int G, int act, void* stream) {
  MF_REQUIRE(x && (out || out_split) && gn_partial && parts > 0 && N > 0 && N <= 65535 && HW > 0 && C > 0, MF_EINVAL, "gn_apply_from_partials: bad args");
  MF_REQUIRE(!residual_pairs || (!residual && out_split && res_bound), MF_EINVAL,
             "gn_apply_from_partials: a residual given as fp16 pairs needs out_split and the res_bound that scaled it, and excludes the fp32 residual");
  MF_REQUIRE(G > 0 && G <= 256 && C % G == 0 && C % 4 == 0, MF_EUNSUPPORTED, "gn_apply_from_partials: C=%d G=%d (C %% 4 == 0, C %% G == 0, G <= 256)", C, G);
  MF_REQUIRE(!out_split || (C % 8 == 0 && out_bound), MF_EUNSUPPORTED, "gn_apply_from_partials: the fp16-pair output needs C %% 8 == 0 and out_bound");
  MF_REQUIRE(!out_split || !residual || res_bound || (res_bound_slots && res_nslots > 0), MF_EINVAL,
             "gn_apply_from_partials: the fp16-pair output needs res_bound (or res_bound_slots) with a residual");
  MF_REQUIRE(!out_split || !emb || emb_bound, MF_EINVAL, "gn_apply_from_partials: the fp16-pair output needs emb_bound with an embedding");
  MF_REQUIRE((gamma == nullptr) == (beta == nullptr), MF_EINVAL, "gn_apply_from_partials: gamma/beta must both be given or both NULL");
  MF_REQUIRE(!emb || emb_stride % 4 == 0, MF_EUNSUPPORTED, "gn_apply_from_partials: emb_stride must be a multiple of 4");
  hipStream_t s = (hipStream_t)stream;
  const long per4 = (long)HW * (C / 4);
  const double nelem = (double)N * HW * C;
  ProfScope ps(MF_FAM_GN_APPLY, s, 8.0 * nelem, 4.0 * nelem * (1 + (out ? 1 : 0) + (residual || residual_pairs ? 1 : 0) + (out_split ? 1 : 0)));
  MF_REQUIRE(per4 < (1L << 31), MF_EUNSUPPORTED, "gn_apply_from_partials: %ld float4 per sample", per4);
  // float4 per thread and round: as many as leave ~4 workgroups per CU over the whole launch (small tensors: more, shorter workgroups)
  const long total_blocks = ((long)N * per4 + 255) / 256;
  static const int force_u = [] { const char* e = getenv("MF_GN_U"); const int v = e ? atoi(e) : 0; return (v == 1 || v == 2 || v == 4) ? v : 0; }();   // (A/B knob)
  const int U = force_u ? force_u : total_blocks >= 4 * 1024 ? 4 : total_blocks >= 2 * 1024 ? 2 : 1;
  long bps = (per4 + 256 * U - 1) / (256 * U);
  // workgroups per CU over the whole launch: every workgroup reduces the sample's records before it starts (~3 us), so FEWER, fatter ones
  // amortise it -- MF_GN_BLOCKS_PER_CU (read once; A/B knob)
  static const int per_cu = [] { const char* e = getenv("MF_GN_BLOCKS_PER_CU"); const int v = e ? atoi(e) : 8; return v < 1 ? 1 : v > 32 ? 32 : v; }();
  const long cap = (256 * per_cu + N - 1) / N;
  if (bps > cap) bps = cap;
  const GnSplit sp{out_split, nullptr, res_bound, emb_bound, bconst, out_bound, res_bound ? nullptr : res_bound_slots, res_nslots, residual_pairs};
  const double count = (double)HW * (C / G);
#define GN_PART_LAUNCH(SPLIT_, U_)                                                                                                          \
  MF_LAUNCH((gn_apply_part_kernel<SPLIT_, U_>), dim3((int)bps, N), dim3(256), 0, s, x, gn_partial, parts, count, eps, gamma, beta, residual, emb, \
            (long)emb_stride, out, HW, C, G, act, sp)
  if (out_split) {
    if (U == 4) GN_PART_LAUNCH(true, 4); else if (U == 2) GN_PART_LAUNCH(true, 2); else GN_PART_LAUNCH(true, 1);
  } else {
    if (U == 4) GN_PART_LAUNCH(false, 4); else if (U == 2) GN_PART_LAUNCH(false, 2); else GN_PART_LAUNCH(false, 1);
  }
#undef GN_PART_LAUNCH
  return check_launch("gn_apply_from_partials");
}

int mf_maxabs_rows_slots(int64_t per_row) {
  const long p4 = per_row / 4;
  int blocks = (int)((p4 + 1023) / 1024);
  if (blocks > 64) blocks = 64;
  if (blocks < 1) blocks = 1;
  return blocks * 4;
}

int mf_maxabs_rows_f32(const float* x, float* partial, float* bound, int N, int64_t per_row, void* stream) {
  MF_REQUIRE(x && partial && N > 0 && per_row > 0 && per_row % 4 == 0 && N <= 65535, MF_EINVAL, "maxabs_rows: bad args (per_row %% 4 == 0)");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(MF_FAM_MISC, s, 0, 4.0 * N * (double)per_row);
  const int slots = mf_maxabs_rows_slots(per_row);
  MF_LAUNCH(maxabs_kernel, dim3(slots / 4, N), dim3(256), 0, s, x, partial, (long)(per_row / 4));
  int rc = check_launch("maxabs_rows");
  if (rc || !bound) return rc;   // (bound null: the slots only -- their consumer reduces them itself, e.g. mf_split_f16x2_slots)
  MF_LAUNCH(bound_finalize_kernel, dim3(N), dim3(64), 0, s, partial, bound, slots);
  return check_launch("bound_finalize");
}

int mf_bound_finalize_f32(const float* partial, float* bound, int N, int slots, void* stream) {
  MF_REQUIRE(partial && bound && N > 0 && slots > 0, MF_EINVAL, "bound_finalize: bad args");
  ProfScope ps(MF_FAM_MISC, (hipStream_t)stream, 0, 4.0 * N * slots);
  MF_LAUNCH(bound_finalize_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, partial, bound, slots);
  return check_launch("bound_finalize");
}


int mf_gn_partial_parts(int HW) { return HW > 0 ? stats_chunks(HW) : 0; }

int mf_gn_finalize_f32(const double* partial, int parts, float* stats, int N, int HW, int C, int G, float eps, void* stream) {
  MF_REQUIRE(partial && stats && parts > 0 && N > 0 && HW > 0 && G > 0 && C % G == 0, MF_EINVAL, "gn_finalize: bad args");
  hipStream_t s = (hipStream_t)stream;
  const int NG = N * G;
  ProfScope ps(MF_FAM_GN_STATS, s, 0, 16.0 * NG * parts);
  MF_LAUNCH(gn_stats_final_kernel, dim3((NG * 16 + 255) / 256), dim3(256), 0, s, partial, stats, NG, G, parts, (double)HW * (C / G), eps);
  return check_launch("gn_finalize");
}

int mf_gn_stats_partial_f32(const float* x, double* partial, int N, int HW, int C, int G, void* stream) {
  MF_REQUIRE(x && partial && N > 0 && HW > 0 && C > 0 && G > 0, MF_EINVAL, "gn_stats_partial: bad args");
  MF_REQUIRE(C % 4 == 0 && C % G == 0, MF_EUNSUPPORTED, "gn_stats_partial: C=%d G=%d unsupported (need C%%4==0, C%%G==0)", C, G);
  const int slices = stats_slices(N, HW, C, G);
  const size_t lds = stats_lds_bytes(C / slices);
  MF_REQUIRE(lds <= 64 * 1024, MF_EUNSUPPORTED, "gn_stats_partial: C=%d too wide", C);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(MF_FAM_GN_STATS, s, 3.0 * N * HW * C, 4.0 * N * (double)HW * C);
  MF_LAUNCH(gn_partial_kernel<false>, dim3(stats_chunks(HW), N, slices), dim3(kStatsThreads), lds, s, x, partial, HW, C, G, 1, 0L, (const float*)nullptr,
                     (float*)nullptr, (float*)nullptr);
  return check_launch("gn_stats_partial");
}

}  // extern "C"
