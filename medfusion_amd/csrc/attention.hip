// attention.hip -- multi-head attention of compute_attention (attention_blocks.py:35-43), fp32, token-major.
//
// q [B][Nq][H*d], k/v [B][Nk][H*d] -> out [B][Nq][H*d];  softmax_j((q s).(k s)) v,  s = d^-0.25.
// Flash-style: the [Nq][Nk] score matrix is never materialised (the reference builds [B*8, N, N]).
// One workgroup = 64 queries of one (batch, head); K/V tiles of 64 keys staged in LDS and shared by the
// 4 waves; per query the 64 lanes each score one key, wave-shuffle max/sum for the online softmax, then
// lanes switch to the head-dim axis for P.V.  This VALU kernel is the general fallback (any head dim <= 128, 1-key
// cross-attention); head dims 8/16/32/64/128 run on the fp32 matrix cores (attention_mfma_kernel below).  The published
// model runs with use_attention='none' (SURVEY F4), so attention carries 0 % of the headline FLOPs.
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


// ------------------------------------------------------------------ fp32-MFMA flash attention (d % 8 == 0, d <= 128)
// One wave = 32 queries, 4 waves per workgroup share K/V tiles of 32 keys in LDS.
//  * S^T = K Q^T ("swapped"): v_mfma_f32_32x32x2_f32 with A = K (row = key), B = Q^T (col = query).  The accumulator layout
//    then gives lane (query j, half h) the scores of ONE query for 16 keys: the softmax row reductions are 16 in-register
//    ops + one cross-half shuffle, and every rescale is a per-lane scalar.
//  * O^T = V^T P: A = V^T (row = head-dim index), B = P.  The k-step -> key mapping of this contraction is free as long as
//    A and B agree, so it is chosen to BE the accumulator layout of S^T: step s, half h <-> key (s&3) + 8(s>>2) + 4h.
//    P therefore feeds the second MFMA straight from the registers it was computed in -- no LDS round trip, no shuffles.
//  * exact fp32 throughout (scores, exp, accumulation).
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4a __attribute__((ext_vector_type(4)));

template <int D>
__global__ __launch_bounds__(256) void attention_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                              float* __restrict__ out, int H, int Nq, int Nk, float scale) {
  constexpr int LDKV = D + 4;          // +4 floats: conflict-free ds_read_b128 of K rows; V rows are read with b32 (lane = column)
  constexpr int DT = (D + 31) / 32;    // 32-wide head-dim tiles of O^T
  __shared__ __attribute__((aligned(16))) float Ks[32 * LDKV];
  __shared__ __attribute__((aligned(16))) float Vs[32 * LDKV];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int b = blockIdx.z, head = blockIdx.y;
  const int C = H * D;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const float* qb = q + (long)b * Nq * C + head * D;
  const float* kb = k + (long)b * Nk * C + head * D;
  const float* vb = v + (long)b * Nk * C + head * D;

  // Q^T operand: lane (query j, half h) keeps Q[q0+j][8kk + 4h .. +3], pre-scaled like the reference's (q * scale)
  f32x4a qf[D / 8];
  {
    const int qi = min(q0 + j, Nq - 1);
#pragma unroll
    for (int kk = 0; kk < D / 8; ++kk) {
      const f32x4a t = *reinterpret_cast<const f32x4a*>(qb + (long)qi * C + kk * 8 + 4 * h);
      qf[kk] = t * scale;
    }
  }
  f32x16 o[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m = -INFINITY, l = 0.f;

  for (int k0 = 0; k0 < Nk; k0 += 32) {
    __syncthreads();
    for (int i = tid; i < 32 * (D / 4); i += 256) {  // stage 32 keys x D of K (scaled) and V
      const int r = i / (D / 4), c4 = i - r * (D / 4);
      const int ki = min(k0 + r, Nk - 1);
      const f32x4a kv = *reinterpret_cast<const f32x4a*>(kb + (long)ki * C + c4 * 4);
      const f32x4a vv = *reinterpret_cast<const f32x4a*>(vb + (long)ki * C + c4 * 4);
      *reinterpret_cast<f32x4a*>(Ks + r * LDKV + c4 * 4) = kv * scale;
      *reinterpret_cast<f32x4a*>(Vs + r * LDKV + c4 * 4) = vv;
    }
    __syncthreads();
    // S^T[key][query] for this wave's 32 queries
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < D / 8; ++kk) {
      const f32x4a kf = *reinterpret_cast<const f32x4a*>(Ks + j * LDKV + kk * 8 + 4 * h);  // A: row = key j
#pragma unroll
      for (int s = 0; s < 4; ++s) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qf[kk][s], st, 0, 0, 0);
    }
    // lane (query j, half h) holds keys (r&3) + 8(r>>2) + 4h, r = 0..15
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (key >= Nk) st[r] = -INFINITY;
      mx = fmaxf(mx, st[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mn = fmaxf(m, mx);
    const float alpha = __expf(m - mn);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      st[r] = __expf(st[r] - mn);   // P, in place
      ps += st[r];
    }
    ps += __shfl_xor(ps, 32, 64);
    l = l * alpha + ps;
    m = mn;
    // O^T[dcol][query] = alpha * O^T + V^T P
#pragma unroll
    for (int t = 0; t < DT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
      const int dc = t * 32 + j;  // A row = head-dim column dc; A[dc][step s, half h] = V[key (s&3)+8(s>>2)+4h][dc]
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float vf = (dc < D) ? Vs[((s & 3) + 8 * (s >> 2) + 4 * h) * LDKV + dc] : 0.f;
        o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, st[s], o[t], 0, 0, 0);
      }
    }
  }
  // lane (query j, half h) holds O^T rows dcol = t*32 + (r&3) + 8(r>>2) + 4h
  const int qi = q0 + j;
  if (qi < Nq) {
    const float inv = 1.0f / l;
    float* ob = out + ((long)b * Nq + qi) * C + head * D;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int dc = t * 32 + 8 * rq + 4 * h;
        if (dc < D) *reinterpret_cast<f32x4a*>(ob + dc) = f32x4a{o[t][4 * rq] * inv, o[t][4 * rq + 1] * inv, o[t][4 * rq + 2] * inv, o[t][4 * rq + 3] * inv};
      }
  }
}

template <int D>
static int launch_attention_mfma(const float* q, const float* k, const float* v, float* out, int B, int H, int Nq, int Nk, float scale, hipStream_t s) {
  MF_LAUNCH(attention_mfma_kernel<D>, dim3((Nq + 127) / 128, H, B), dim3(256), 0, s, q, k, v, out, H, Nq, Nk, scale);
  return check_launch("attention_mfma");
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
  const bool aligned = (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) == 0;
  if (aligned && Nk >= 8) {  // fp32 matrix cores; the VALU kernel below covers odd head dims and the 1-key cross-attention
    switch (d) {
      case 8: return launch_attention_mfma<8>(q, k, v, out, B, H, Nq, Nk, scale, s);
      case 16: return launch_attention_mfma<16>(q, k, v, out, B, H, Nq, Nk, scale, s);
      case 32: return launch_attention_mfma<32>(q, k, v, out, B, H, Nq, Nk, scale, s);
      case 64: return launch_attention_mfma<64>(q, k, v, out, B, H, Nq, Nk, scale, s);
      case 128: return launch_attention_mfma<128>(q, k, v, out, B, H, Nq, Nk, scale, s);
      default: break;
    }
  }
  static DeviceOnce once;
  if (first_use_on_device(once))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  MF_LAUNCH(attention_kernel, dim3((Nq + kWaves * kQPerWave - 1) / (kWaves * kQPerWave), H, B), dim3(256), lds, s, q, k, v, out, H, Nq,
                     Nk, d, scale);
  return check_launch("attention");
}

}  // extern "C"
