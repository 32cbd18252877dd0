// attention.hip -- multi-head attention of compute_attention (attention_blocks.py:35-43), fp32, token-major.
//
// q [B][Nq][H*d], k/v [B][Nk][H*d] -> out [B][Nq][H*d];  softmax_j((q s).(k s)) v,  s = d^-0.25.
// Flash-style: the [Nq][Nk] score matrix is never materialised (the reference builds [B*8, N, N]).
// One workgroup = 64 queries of one (batch, head); K/V tiles of 64 keys staged in LDS and shared by the
// 4 waves; per query the 64 lanes each score one key, wave-shuffle max/sum for the online softmax, then
// lanes switch to the head-dim axis for P.V.  The published model runs with use_attention='none'
//  so this path carries 0 % of the headline FLOPs; an MFMA version is future work.
#include "common.h"

using namespace mf;

namespace {

constexpr int kQPerWave = 16, kWaves = 4, kKeys = 64, kMaxD = 128;

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                         float* __restrict__ out, int H, int Nq, int Nk, int d, float scale) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  const int ldk = d + 1;
  float* Ks = sh;                          // [kKeys][d+1]
  float* Vs = Ks + kKeys * ldk;            // [kKeys][d+1]
  float* Qs = Vs + kKeys * ldk;            // [kWaves*kQPerWave][d]
  float* Ps = Qs + kWaves * kQPerWave * d; // [kWaves][kKeys]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * (kWaves * kQPerWave);
  const int C = H * d;
  const float* qb = q + (long)b * Nq * C + h * d;
  const float* kb = k + (long)b * Nk * C + h * d;
  const float* vb = v + (long)b * Nk * C + h * d;

  for (int i = tid; i < kWaves * kQPerWave * d; i += 256) {
    const int r = i / d, c = i - r * d;
    const int qi = q0 + r;
    Qs[i] = qi < Nq ? qb[(long)qi * C + c] * scale : 0.f;
  }

  float m[kQPerWave], l[kQPerWave], o0[kQPerWave], o1[kQPerWave];
#pragma unroll
  for (int r = 0; r < kQPerWave; ++r) { m[r] = -INFINITY; l[r] = 0.f; o0[r] = 0.f; o1[r] = 0.f; }

  for (int k0 = 0; k0 < Nk; k0 += kKeys) {
    __syncthreads();
    for (int i = tid; i < kKeys * d; i += 256) {
      const int r = i / d, c = i - r * d;
      const int ki = k0 + r;
      Ks[r * ldk + c] = ki < Nk ? kb[(long)ki * C + c] * scale : 0.f;
      Vs[r * ldk + c] = ki < Nk ? vb[(long)ki * C + c] : 0.f;
    }
    __syncthreads();
    const bool key_ok = (k0 + lane) < Nk;
    const float* kr = Ks + lane * ldk;
#pragma unroll
    for (int r = 0; r < kQPerWave; ++r) {
      const float* qr = Qs + (wave * kQPerWave + r) * d;
      float s = 0.f;
      for (int c = 0; c < d; ++c) s = fmaf(qr[c], kr[c], s);
      s = key_ok ? s : -INFINITY;
      const float mn = fmaxf(m[r], wave_max(s));
      const float alpha = __expf(m[r] - mn);  // m[r] = -inf on the first tile -> 0
      const float p = key_ok ? __expf(s - mn) : 0.f;
      l[r] = l[r] * alpha + wave_sum(p);
      m[r] = mn;
      Ps[wave * kKeys + lane] = p;
      __builtin_amdgcn_wave_barrier();
      float a0 = 0.f, a1 = 0.f;
      const float* pw = Ps + wave * kKeys;
      if (lane < d) {
        for (int j = 0; j < kKeys; ++j) a0 = fmaf(pw[j], Vs[j * ldk + lane], a0);
      }
      if (lane + 64 < d) {
        for (int j = 0; j < kKeys; ++j) a1 = fmaf(pw[j], Vs[j * ldk + lane + 64], a1);
      }
      o0[r] = o0[r] * alpha + a0;
      o1[r] = o1[r] * alpha + a1;
      __builtin_amdgcn_wave_barrier();
    }
  }
#pragma unroll
  for (int r = 0; r < kQPerWave; ++r) {
    const int qi = q0 + wave * kQPerWave + r;
    if (qi >= Nq) continue;
    const float inv = 1.0f / l[r];
    float* ob = out + ((long)b * Nq + qi) * C + h * d;
    if (lane < d) ob[lane] = o0[r] * inv;
    if (lane + 64 < d) ob[lane + 64] = o1[r] * inv;
  }
}

}  // namespace

extern "C" {

int mf_attention_f32(const float* q, const float* k, const float* v, float* out, int B, int H, int Nq, int Nk, int d, float scale,
                     void* stream) {
  MF_REQUIRE(q && k && v && out && B > 0 && H > 0 && Nq > 0 && Nk > 0 && d > 0, MF_EINVAL, "attention: bad args");
  MF_REQUIRE(d <= kMaxD, MF_EUNSUPPORTED, "attention: head dim %d > %d", d, kMaxD);
  MF_REQUIRE(B <= 65535 && H <= 65535, MF_EUNSUPPORTED, "attention: grid too large");
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = ((size_t)2 * kKeys * (d + 1) + (size_t)kWaves * kQPerWave * d + kWaves * kKeys) * sizeof(float);
  const double flops = 4.0 * B * H * (double)Nq * Nk * d;
  ProfScope ps(MF_FAM_ATTENTION, s, flops, 4.0 * B * H * d * (2.0 * Nq + 2.0 * Nk));
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(attention_kernel, dim3((Nq + kWaves * kQPerWave - 1) / (kWaves * kQPerWave), H, B), dim3(256), lds, s, q, k, v, out, H, Nq,
                     Nk, d, scale);
  return check_launch("attention");
}

}  // extern "C"
