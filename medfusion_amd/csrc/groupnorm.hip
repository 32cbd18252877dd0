// groupnorm.hip -- GroupNorm statistics and the fused normalise + Swish + residual + embedding pass (NHWC, fp32).
//
// Both kernels are HBM/Infinity-Cache streaming passes: fully coalesced float4 rows, wave-shuffle +
// LDS reductions.  Statistics are accumulated per thread in fp32 over <= a few hundred elements and
// combined in fp64 (a group of the VAE's 256x256 level holds 524 288 elements: E[x^2]-E[x]^2 in fp32
// would lose the parity margin -- SURVEY §7 hard parts).
#include "common.h"

using namespace mf;

namespace {

constexpr int kStatsThreads = 256;
constexpr int kMaxChunks = 64;

__host__ __device__ inline int stats_chunks(int HW) {
  // ~>= 64 pixels per chunk, at most kMaxChunks chunks per sample
  int c = (HW + 63) / 64;
  return c < 1 ? 1 : (c > kMaxChunks ? kMaxChunks : c);
}

// grid (chunks, N).  Each block reduces its pixel range of sample n for all channels, then per group.
// partial[((n*chunks + chunk)*G + g)*2 + {0,1}] = {sum, sumsq} (double)
__global__ __launch_bounds__(kStatsThreads) void gn_stats_partial_kernel(const float* __restrict__ x, double* __restrict__ partial, int HW,
                                                                          int C, int G) {
  extern __shared__ __attribute__((aligned(16))) float sh[];  // [rowphases][C][2]
  const int chunks = gridDim.x, chunk = blockIdx.x, n = blockIdx.y;
  const int C4 = C >> 2;
  const int lanes_per_row = C4 < kStatsThreads ? C4 : kStatsThreads;  // threads cooperating on one pixel row
  const int rowphases = kStatsThreads / lanes_per_row;
  const int tid = threadIdx.x;
  const int phase = tid / lanes_per_row, col = tid - phase * lanes_per_row;
  const int p_per = (HW + chunks - 1) / chunks;
  const int p0 = chunk * p_per, p1 = min(HW, p0 + p_per);
  const float* base = x + (long)n * HW * C;

  // each thread owns columns col, col+lanes_per_row, ... (only >1 when C4 > 256)
  for (int c4 = col; c4 < C4; c4 += lanes_per_row) {
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    if (phase < rowphases) {
      for (int px = p0 + phase; px < p1; px += rowphases) {
        const float4 v = *reinterpret_cast<const float4*>(base + (long)px * C + c4 * 4);
        s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
        q0 = fmaf(v.x, v.x, q0); q1 = fmaf(v.y, v.y, q1); q2 = fmaf(v.z, v.z, q2); q3 = fmaf(v.w, v.w, q3);
      }
      float* d = sh + ((long)phase * C + c4 * 4) * 2;
      d[0] = s0; d[1] = q0; d[2] = s1; d[3] = q1; d[4] = s2; d[5] = q2; d[6] = s3; d[7] = q3;
    }
  }
  __syncthreads();
  // per group: sum over row phases and the group's channels, in double
  const int cpg = C / G;
  for (int g = tid; g < G; g += kStatsThreads) {
    double s = 0, q = 0;
    for (int ph = 0; ph < rowphases; ++ph) {
      const float* d = sh + ((long)ph * C + g * cpg) * 2;
      for (int c = 0; c < cpg; ++c) { s += (double)d[2 * c]; q += (double)d[2 * c + 1]; }
    }
    double* o = partial + (((long)n * chunks + chunk) * G + g) * 2;
    o[0] = s; o[1] = q;
  }
}

__global__ void gn_stats_final_kernel(const double* __restrict__ partial, float* __restrict__ stats, int NG, int G, int chunks, double count,
                                      float eps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NG) return;
  const int n = i / G, g = i - n * G;
  double s = 0, q = 0;
  for (int c = 0; c < chunks; ++c) {
    const double* p = partial + (((long)n * chunks + c) * G + g) * 2;
    s += p[0]; q += p[1];
  }
  const double mean = s / count;
  double var = q / count - mean * mean;
  if (var < 0) var = 0;
  stats[2 * i] = (float)mean;
  stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
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
  const int chunks = stats_chunks(HW);
  const int C4 = C / 4;
  const int lanes_per_row = C4 < kStatsThreads ? C4 : kStatsThreads;
  const int rowphases = kStatsThreads / lanes_per_row;
  const size_t lds = (size_t)rowphases * C * 2 * sizeof(float);
  MF_REQUIRE(lds <= 64 * 1024, MF_EUNSUPPORTED, "gn_stats: C=%d too wide", C);
  ProfScope ps(MF_FAM_GN_STATS, s, 3.0 * N * HW * C, 4.0 * N * (double)HW * C);
  hipLaunchKernelGGL(gn_stats_partial_kernel, dim3(chunks, N), dim3(kStatsThreads), lds, s, x, reinterpret_cast<double*>(workspace), HW, C, G);
  int rc = check_launch("gn_stats_partial");
  if (rc) return rc;
  const int NG = N * G;
  hipLaunchKernelGGL(gn_stats_final_kernel, dim3((NG + 127) / 128), dim3(128), 0, s, reinterpret_cast<const double*>(workspace), stats, NG, G,
                     chunks, (double)HW * (C / G), eps);
  return check_launch("gn_stats_final");
}

int mf_gn_apply_f32(const float* x, const float* stats, const float* gamma, const float* beta, const float* residual, const float* emb,
                    int64_t emb_stride, float* out, int N, int HW, int C, int G, int act, void* stream) {
  MF_REQUIRE(x && out && N > 0 && HW > 0 && C > 0, MF_EINVAL, "gn_apply: bad args");
  MF_REQUIRE(C % 4 == 0, MF_EUNSUPPORTED, "gn_apply: C=%d must be a multiple of 4", C);
  MF_REQUIRE(!stats || (G > 0 && C % G == 0), MF_EINVAL, "gn_apply: C=%d G=%d", C, G);
  MF_REQUIRE((gamma == nullptr) == (beta == nullptr), MF_EINVAL, "gn_apply: gamma/beta must both be given or both NULL");
  MF_REQUIRE(!emb || emb_stride % 4 == 0, MF_EUNSUPPORTED, "gn_apply: emb_stride must be a multiple of 4");
  hipStream_t s = (hipStream_t)stream;
  const long total4 = (long)N * HW * (C / 4);
  const double nelem = (double)N * HW * C;
  ProfScope ps(MF_FAM_GN_APPLY, s, 8.0 * nelem, 4.0 * nelem * (2 + (residual ? 1 : 0)));
  long blocks = (total4 + 255) / 256;
  if (blocks > 256 * 8) blocks = 256 * 8;
  hipLaunchKernelGGL(gn_apply_kernel, dim3((int)blocks), dim3(256), 0, s, x, stats, gamma, beta, residual, emb, (long)emb_stride, out, total4, HW, C,
                     G > 0 ? G : 1, act);
  return check_launch("gn_apply");
}

}  // extern "C"
