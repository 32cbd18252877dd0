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
  const size_t a = (size_t)(kStatsThreads / lanes_per_row) * Cs * 2 * sizeof(float);
  const size_t b = (size_t)(2 * kStatsThreads + 2) * sizeof(double);  // fused finalize scratch + flag
  return a > b ? a : b;
}

// Last-arriver finalize (no spinning, placement-independent; CDNA guide §6 G16 counter form): every producer workgroup of sample n
// drains its partial-record stores, one lane issues an agent-scope release and bumps counter[n]; the workgroup that draws
// total-1 acquires and reduces the P x G records of the sample to stats[n][g] = {mean, rstd} in fp64.
// `flag` is one int of the caller's dynamic LDS (a second static __shared__ object would perturb the conv pipeline's waits).
struct GnFinal {
  float* stats;        // [N][G][2] or nullptr (no fused finalize)
  int* counter;        // [N]: must be ZERO on entry; the last arriver resets it, so it is zero again on exit (allocate zeroed once)
  int total;           // producer workgroups per sample
  int parts;           // P: partial records per sample and group
  double count;        // elements per group
  float eps;
};

__device__ __forceinline__ void gn_arrive_and_finalize(const GnFinal& f, const double* partial, int n, int G, volatile int* flag, double* red) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its partial-record stores have left
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int prev = __hip_atomic_fetch_add(f.counter + n, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = (prev == f.total - 1) ? 1 : 0;
  }
  __syncthreads();
  if (*flag == 0) return;  // uniform per workgroup
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __hip_atomic_store(f.counter + n, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // self-cleaning: ready for the next launch
  }
  __syncthreads();
  // thread = (phase, g): phases stride over the P records, then a tree over phases in `red` (>= 2*blockDim doubles of LDS)
  const int nt = blockDim.x;
  const int phases = G <= nt ? nt / G : 1;
  const int g = threadIdx.x % G, ph = threadIdx.x / G;
  for (int g0 = 0; g0 < G; g0 += nt) {  // G > blockDim only in theory
    const int gg = g0 + g;
    double s = 0, q = 0;
    if (ph < phases && gg < G)
      for (int c = ph; c < f.parts; c += phases) {
        const double* pp = partial + (((long)n * f.parts + c) * G + gg) * 2;
        s += pp[0]; q += pp[1];
      }
    red[threadIdx.x * 2] = s; red[threadIdx.x * 2 + 1] = q;
    __syncthreads();
    if (ph == 0 && gg < G) {
      for (int k = 1; k < phases; ++k) { s += red[(k * G + g) * 2]; q += red[(k * G + g) * 2 + 1]; }
      const double mean = s / f.count;
      double var = q / f.count - mean * mean;
      if (var < 0) var = 0;
      f.stats[((long)n * G + gg) * 2] = (float)mean;
      f.stats[((long)n * G + gg) * 2 + 1] = (float)(1.0 / sqrt(var + (double)f.eps));
    }
    __syncthreads();
  }
}

// grid (chunks, N).  Each block reduces its pixel range of sample n for all channels, then per group.
// partial[((n*chunks + chunk)*G + g)*2 + {0,1}] = {sum, sumsq} (double).
// REDUCE: the value is sum_z slabs[z] + bias (split-K reduction) and is also written to y.
template <bool REDUCE>
__global__ __launch_bounds__(kStatsThreads) void gn_partial_kernel(const float* __restrict__ x, double* __restrict__ partial, int HW, int C, int G,
                                                                   int nslabs, long slab, const float* __restrict__ bias, float* __restrict__ y,
                                                                   const GnFinal fin, unsigned* __restrict__ out_bound = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float sh[];  // [rowphases][Cs][2], then (fused finalize) 2*256 doubles + flag
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
        float4 v = *reinterpret_cast<const float4*>(x + e);
        if (REDUCE) {
          for (int z = 1; z < nslabs; ++z) {
            const float4 u = *reinterpret_cast<const float4*>(x + (long)z * slab + e);
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
          }
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
  if (REDUCE && out_bound) {  // measured upper bound of |y| over the sample (operand scale of a following fp16-pair convolution)
    vmax = wave_max(vmax);
    if ((tid & 63) == 0) atomicMax(out_bound + n, absbits(vmax));
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
  if (fin.stats) {
    __syncthreads();  // the per-channel scratch in sh[] is dead: reuse it
    double* red = reinterpret_cast<double*>(sh);
    gn_arrive_and_finalize(fin, partial, n, G, reinterpret_cast<volatile int*>(red + 2 * kStatsThreads), red);
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Split-K reduction + GroupNorm statistics + finalize + apply (norm, affine, Swish, residual, embedding) in ONE kernel.
// The workgroups (chunk 0..P-1) of one (sample, channel slice) keep their reduced values IN REGISTERS, publish their per-group
// partial sums, meet at a per-(sample, slice) arrival counter, finalize mean / rstd of their own groups and write the finished
// activation: the tensor is never written and re-read un-normalised, and the finalize + apply launches disappear.
// Waiting workgroups cannot starve the ones they wait for: the P workgroups of a domain are consecutive in dispatch order (chunk is
// blockIdx.x), earlier domains are complete, so only the LAST domain can be partially resident and it needs P <= 16 slots.
// The spin is bounded all the same: on timeout the workgroup writes NaNs (a visible failure, never a hang).
constexpr int kFusedMaxVec = 16;  // float4 values a thread carries across the meeting point

struct GnApplyArgs {
  const float* gamma; const float* beta;   // [C] or both null
  const float* residual;                   // [N][HW][C] or null
  const float* emb; long emb_stride;       // [N][emb_stride] (+ c) or null
  float* out;                              // [N][HW][C]
  int* arrive; int* done;                  // [N * slices] each, zero on entry, zero again on exit
  double count; float eps; int act;
};

inline bool fused_apply_ok(int N, int HW, int C, int G) {
  if (G <= 0 || C % G || C % 4) return false;
  const int sl = stats_slices(N, HW, C, G), Cs = C / sl, C4 = Cs / 4;
  if (Cs > 1024) return false;
  const int lanes = C4 < kStatsThreads ? C4 : kStatsThreads, rowphases = kStatsThreads / lanes, chunks = stats_chunks(HW);
  const int p_per = (HW + chunks - 1) / chunks;
  if ((p_per + rowphases - 1) / rowphases > kFusedMaxVec) return false;
  if ((Cs / (C / G)) * chunks > 4 * kStatsThreads) return false;
  return stats_lds_bytes(Cs) <= 64 * 1024;
}

template <int UNUSED = 0>  // (a template only so that the header can be included from two translation units)
__global__ __launch_bounds__(kStatsThreads) void gn_reduce_apply_kernel(const float* __restrict__ slabs, double* __restrict__ partial, int HW, int C, int G,
                                                                        int nslabs, long slab, const float* __restrict__ bias, const GnApplyArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sh[];  // [rowphases][Cs][2] floats; later [gs][2] mean/rstd
  const int chunks = gridDim.x, chunk = blockIdx.x, n = blockIdx.y;
  const int Cs = C / gridDim.z, c_off = blockIdx.z * Cs;
  const int C4 = Cs >> 2;
  const int lanes_per_row = C4 < kStatsThreads ? C4 : kStatsThreads;
  const int rowphases = kStatsThreads / lanes_per_row;
  const int tid = threadIdx.x;
  const int phase = tid / lanes_per_row, col = tid - phase * lanes_per_row;
  const int p_per = (HW + chunks - 1) / chunks;
  const int p0 = chunk * p_per, p1 = min(HW, p0 + p_per);
  const long nbase = (long)n * HW * C;
  const bool active = phase < rowphases && col < C4;   // (Cs <= 1024: one float4 column per thread)
  const int cbase = c_off + col * 4;

  float4 vals[kFusedMaxVec];
  float s0 = 0, s1 = 0, s2 = 0, s3 = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0;
  if (active) {
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) bv = *reinterpret_cast<const float4*>(bias + cbase);
#pragma unroll
    for (int k = 0; k < kFusedMaxVec; ++k) {
      const int px = p0 + phase + k * rowphases;
      if (px < p1) {
        const long e = nbase + (long)px * C + cbase;
        float4 v = *reinterpret_cast<const float4*>(slabs + e);
        for (int z = 1; z < nslabs; ++z) {
          const float4 u = *reinterpret_cast<const float4*>(slabs + (long)z * slab + e);
          v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        vals[k] = v;
        s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
        q0 = fmaf(v.x, v.x, q0); q1 = fmaf(v.y, v.y, q1); q2 = fmaf(v.z, v.z, q2); q3 = fmaf(v.w, v.w, q3);
      }
    }
    float* d = sh + ((long)phase * Cs + col * 4) * 2;
    d[0] = s0; d[1] = q0; d[2] = s1; d[3] = q1; d[4] = s2; d[5] = q2; d[6] = s3; d[7] = q3;
  }
  __syncthreads();
  const int cpg = C / G, gs = Cs / cpg, g_off = c_off / cpg;  // groups of this slice
  for (int g = tid; g < gs; g += kStatsThreads) {
    double s = 0, q = 0;
    for (int ph = 0; ph < rowphases; ++ph) {
      const float* d = sh + ((long)ph * Cs + g * cpg) * 2;
      for (int c = 0; c < cpg; ++c) { s += (double)d[2 * c]; q += (double)d[2 * c + 1]; }
    }
    double* o = partial + (((long)n * chunks + chunk) * G + (g_off + g)) * 2;
    // agent-scope atomic stores / loads for the few bytes that cross workgroups (XCDs): the records travel through the coherence
    // point themselves.  A release/acquire FENCE pair instead writes back and invalidates the whole L2 of the XCD -- including the
    // split-K slabs the convolution has just left there (measured: 9 % slower end to end).
    __hip_atomic_store(o, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(o + 1, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- meeting point of the P workgroups of (n, slice)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int dom = n * gridDim.z + blockIdx.z;
  int* okflag = reinterpret_cast<int*>(sh);  // the per-channel scratch is dead
  if (tid == 0) {  // (every wave has drained its record stores: s_waitcnt above, then the barrier)
    __hip_atomic_fetch_add(a.arrive + dom, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int it = 0;
    while (__hip_atomic_load(a.arrive + dom, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < chunks && ++it < (1 << 22)) __builtin_amdgcn_s_sleep(4);
    *okflag = it < (1 << 22) ? 1 : 0;
  }
  __syncthreads();
  const bool ok = *okflag != 0;
  __syncthreads();
  // ---- finalize the groups of this slice: thread = (record c, group g), tree over the records in LDS
  double* red = reinterpret_cast<double*>(sh);           // [chunks][gs][2] doubles  (<= 4 * 256 * 2 * 8 bytes... checked on the host)
  for (int i = tid; i < chunks * gs; i += kStatsThreads) {
    const int c = i / gs, g = i - c * gs;
    const double* pp = partial + (((long)n * chunks + c) * G + (g_off + g)) * 2;
    red[2 * i] = __hip_atomic_load(pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    red[2 * i + 1] = __hip_atomic_load(pp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  float* st = reinterpret_cast<float*>(red + 2 * chunks * gs);  // [gs][2]
  for (int g = tid; g < gs; g += kStatsThreads) {
    double s = 0, q = 0;
    for (int c = 0; c < chunks; ++c) { s += red[2 * (c * gs + g)]; q += red[2 * (c * gs + g) + 1]; }
    const double mean = s / a.count;
    double var = q / a.count - mean * mean;
    if (var < 0) var = 0;
    st[2 * g] = ok ? (float)mean : __builtin_nanf("");
    st[2 * g + 1] = (float)(1.0 / sqrt(var + (double)a.eps));
  }
  __syncthreads();
  if (tid == 0) {  // self-cleaning counters: the last workgroup through resets both (everyone has passed the spin by then)
    const int prev = __hip_atomic_fetch_add(a.done + dom, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == chunks - 1) {
      __hip_atomic_store(a.arrive + dom, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.done + dom, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // ---- apply from registers
  if (active) {
    float gm[4] = {1.f, 1.f, 1.f, 1.f}, bt[4] = {0.f, 0.f, 0.f, 0.f}, mean[4], rstd[4], em[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int g = (col * 4 + k) / cpg;
      mean[k] = st[2 * g]; rstd[k] = st[2 * g + 1];
      if (a.gamma) { gm[k] = a.gamma[cbase + k]; bt[k] = a.beta[cbase + k]; }
      if (a.emb) em[k] = a.emb[(long)n * a.emb_stride + cbase + k];
    }
#pragma unroll
    for (int k = 0; k < kFusedMaxVec; ++k) {
      const int px = p0 + phase + k * rowphases;
      if (px < p1) {
        const long e = nbase + (long)px * C + cbase;
        const float4 v = vals[k];
        float t[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float u = (t[j] - mean[j]) * rstd[j];
          if (a.gamma) u = u * gm[j] + bt[j];
          if (a.act == 1) u = swish_acc(u);
          t[j] = u;
        }
        if (a.residual) {
          const float4 r = *reinterpret_cast<const float4*>(a.residual + e);
          t[0] += r.x; t[1] += r.y; t[2] += r.z; t[3] += r.w;
        }
        if (a.emb) { t[0] += em[0]; t[1] += em[1]; t[2] += em[2]; t[3] += em[3]; }
        *reinterpret_cast<float4*>(a.out + e) = make_float4(t[0], t[1], t[2], t[3]);
      }
    }
  }
}

inline size_t fused_apply_lds_bytes(int Cs, int chunks, int gs) {
  const size_t a = stats_lds_bytes(Cs);
  const size_t b = (size_t)chunks * gs * 2 * sizeof(double) + (size_t)gs * 2 * sizeof(float) + 16;
  return a > b ? a : b;
}

}  // namespace mf
