// sched_noise.hip -- fused scheduler step (CFG combine + x0/xT algebra + posterior sample + DDIM update) and
// counter-based Gaussian noise.  Pure streaming kernels over [B, C, h, w] latents (8192 floats per sample).
#include "common.h"

using namespace mf;

namespace {

// Every product/sum is rounded on its own so that, given identical inputs, the result is bit-identical to
// ATen's chain of elementwise ops on CPU.  The library is built with -ffp-contract=off (on AMD the __f*_rn
// intrinsics are plain operators and would otherwise be fused into FMAs); the pragma restates it locally.
struct SchedOut { float xn, x0, xT; };

// one element of the step: every product / sum rounded on its own (see above)
__device__ __forceinline__ SchedOut sched_elem(const MfSchedArgs& a, const MfSchedStep& S, float xt, float pred, float pu, float pv, float npost, float nddim) {
#pragma clang fp contract(off)
  if (a.pred_uncond) {  // diffusion_pipeline.py:244  pred_uncond + g * (pred_cond - pred_uncond)
    const float dlt = pred - pu;
    const float sc = a.guidance_scale * dlt;
    pred = pu + sc;
  }
  float x0, xT;
  if (a.objective == 0) {  // 'x_T': gaussian_scheduler.py:119-124
    const float p1 = S.sqrt_recip_ac * xt;
    const float p2 = S.sqrt_recipm1_ac * pred;
    x0 = p1 - p2;
    if (a.clip_x0) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
    xT = pred;
  } else {  // 'x_0': diffusion_pipeline.py:264-267, gaussian_scheduler.py:127-131
    x0 = a.clip_x0 ? fminf(fmaxf(pred, -1.0f), 1.0f) : pred;
    const float p1 = S.sqrt_recip_ac * xt;
    const float df = p1 - x0;
    xT = df / S.sqrt_recipm1_ac;
  }
  // posterior mean / std: gaussian_scheduler.py:95-100
  const float m1 = S.coef1 * x0;
  const float m2 = S.coef2 * xt;
  const float mean = m1 + m2;
  float sd = S.std_fixed;
  if (a.pred_var) {  // learned variance: var_scale = pred_var/2 + 0.5 (diffusion_pipeline.py:256), :110-116
    const float hv = pv / 2.0f;
    const float vs = hv + 0.5f;
    const float l1 = vs * S.log_var_max;
    const float om = 1.0f - vs;
    const float l2 = om * S.log_var_min;
    const float lv = l1 + l2;
    const float hl = 0.5f * lv;
    sd = S.t == 0 ? 0.0f : expf(hl);
  }
  const float sn = sd * npost;
  const float prior = mean + sn;
  float xn = prior;
  if (S.mode == 1) {  // DDIM: x_0*sqrt(a_next) + c*x_T + sigma*noise  (diffusion_pipeline.py:304)
    const float d1 = x0 * S.ddim_sqrt_an;
    const float d2 = S.ddim_c * xT;
    const float d3 = S.ddim_sigma * nddim;
    const float d12 = d1 + d2;
    xn = d12 + d3;
  }
  return SchedOut{xn, x0, xT};
}

__global__ __launch_bounds__(256) void sched_step_kernel(const MfSchedArgs a) {
#pragma clang fp contract(off)
  const int step = a.step_dev ? *a.step_dev : a.step;
  const MfSchedStep S = a.table[step];
  const float* npost = a.noise_post ? a.noise_post + (long)step * a.noise_step_stride : nullptr;
  const float* nddim = a.noise_ddim ? a.noise_ddim + (long)step * a.noise_step_stride : nullptr;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
    const SchedOut o = sched_elem(a, S, a.x_t[i], a.pred[i], a.pred_uncond ? a.pred_uncond[i] : 0.f, a.pred_var ? a.pred_var[i] : 0.f, npost ? npost[i] : 0.0f,
                                  nddim ? nddim[i] : 0.0f);
    a.x_t_out[i] = o.xn;
    if (a.x0_out) a.x0_out[i] = o.x0;
    if (a.xT_out) a.xT_out[i] = o.xT;
  }
}

// out[b][0..row_len) = table[(step * ncol + cols[b]) * row_len ...): the rows of loop iteration `step` (device counter or host value) of a
// [S][ncol][row_len] table, one table column per batch row.  float4 when row_len % 4 == 0.
__global__ __launch_bounds__(256) void gather_step_rows_kernel(const float* __restrict__ table, const long* __restrict__ cols,
                                                               const int* __restrict__ step_dev, int step, int ncol, long row_len,
                                                               float* __restrict__ out, long total) {
  const long st = step_dev ? (long)*step_dev : (long)step;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / row_len, k = i - b * row_len;
    out[i] = table[((st * ncol) + cols[b]) * row_len + k];
  }
}

__global__ void broadcast_from_table_kernel(const float* __restrict__ table, const int32_t* __restrict__ step_dev, int step, float* __restrict__ out, int n) {
  const int st = step_dev ? *step_dev : step;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = table[st];
}

__global__ void counter_add_kernel(int32_t* c, int32_t inc) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *c += inc;
}

__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
  const uint32_t n3 = (uint32_t)p0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * 5.9604644775390625e-08f; }  // 2^-24

// the 4 standard normals of element quad q of sample `sample` in draw `draw` (oracle/synth.py: philox_normal is the spec)
__device__ __forceinline__ float4 philox_quad(uint32_t q, uint32_t sample, uint32_t draw, uint32_t seed_lo, uint32_t seed_hi) {
  uint32_t c0 = q, c1 = sample, c2 = draw, c3 = 0u;
  uint32_t k0 = seed_lo, k1 = seed_hi;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  const float r0 = sqrtf(-2.0f * logf(u01(c0))), t0 = 6.283185307179586f * u01(c1);
  const float r1 = sqrtf(-2.0f * logf(u01(c2))), t1 = 6.283185307179586f * u01(c3);
  float s0, cs0, s1, cs1;
  sincosf(t0, &s0, &cs0);
  sincosf(t1, &s1, &cs1);
  return make_float4(r0 * cs0, r0 * s0, r1 * cs1, r1 * s1);
}

// one thread per quad of 4 consecutive elements of one sample
__global__ __launch_bounds__(256) void philox_normal_kernel(float* __restrict__ out, uint32_t seed_lo, uint32_t seed_hi, int draw_base, int draw_stride,
                                                             const int32_t* step_dev, int step, long sample_offset, int B, long quads_per_sample) {
  const int st = step_dev ? *step_dev : step;
  const uint32_t draw = (uint32_t)(draw_base + draw_stride * st);
  const long total = (long)B * quads_per_sample;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long b = i / quads_per_sample;
    const long q = i - b * quads_per_sample;
    *reinterpret_cast<float4*>(out + i * 4) = philox_quad((uint32_t)q, (uint32_t)(sample_offset + b), draw, seed_lo, seed_hi);
  }
}

// The tail of a denoise iteration in ONE launch (round 4): both noise draws generated in registers (the Philox quads of
// philox_normal_kernel: draw_base + draw_stride * step is the posterior draw, + 1 the DDIM draw), the scheduler step of sched_step_kernel
// on them (sched_elem: the same arithmetic, bit for bit), and the step counter advanced by whichever workgroup finishes LAST (a ticket:
// every workgroup has read the counter before it takes its ticket) -- four launches of the loop body become one.
struct PhiloxP { uint32_t seed_lo, seed_hi; int draw_base, draw_stride; long sample_offset, quads_per_sample; int32_t* step_rw; uint32_t* ticket; };

__global__ __launch_bounds__(256) void sched_step_philox_kernel(const MfSchedArgs a, const PhiloxP ph) {
#pragma clang fp contract(off)
  const int step = *ph.step_rw;
  const MfSchedStep S = a.table[step];
  const uint32_t draw = (uint32_t)(ph.draw_base + ph.draw_stride * step);
  const bool ddim = S.mode == 1;
  const long total = a.n >> 2;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long b = i / ph.quads_per_sample;
    const long q = i - b * ph.quads_per_sample;
    const float4 np = philox_quad((uint32_t)q, (uint32_t)(ph.sample_offset + b), draw, ph.seed_lo, ph.seed_hi);
    const float4 nd = ddim ? philox_quad((uint32_t)q, (uint32_t)(ph.sample_offset + b), draw + 1u, ph.seed_lo, ph.seed_hi) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 xt = *reinterpret_cast<const float4*>(a.x_t + i * 4), pr = *reinterpret_cast<const float4*>(a.pred + i * 4);
    const float4 pu = a.pred_uncond ? *reinterpret_cast<const float4*>(a.pred_uncond + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 pv = a.pred_var ? *reinterpret_cast<const float4*>(a.pred_var + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const SchedOut o0 = sched_elem(a, S, xt.x, pr.x, pu.x, pv.x, np.x, nd.x), o1 = sched_elem(a, S, xt.y, pr.y, pu.y, pv.y, np.y, nd.y);
    const SchedOut o2 = sched_elem(a, S, xt.z, pr.z, pu.z, pv.z, np.z, nd.z), o3 = sched_elem(a, S, xt.w, pr.w, pu.w, pv.w, np.w, nd.w);
    *reinterpret_cast<float4*>(a.x_t_out + i * 4) = make_float4(o0.xn, o1.xn, o2.xn, o3.xn);
    if (a.x0_out) *reinterpret_cast<float4*>(a.x0_out + i * 4) = make_float4(o0.x0, o1.x0, o2.x0, o3.x0);
    if (a.xT_out) *reinterpret_cast<float4*>(a.xT_out + i * 4) = make_float4(o0.xT, o1.xT, o2.xT, o3.xT);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(ph.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1u == gridDim.x) {   // the last workgroup: every other one has read *step_rw (before its own ticket)
      __hip_atomic_store(ph.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ph.step_rw, step + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// up to three gathers of the rows of loop iteration `step` in one launch (blockIdx.y = which): gather_step_rows_kernel's work for the
// embedding rows, the local-embedder rows and their bounds, which share `cols` and the step
struct GatherSeg { const float* table; float* out; long row_len; };
__global__ __launch_bounds__(256) void gather_step_rows3_kernel(const GatherSeg g0, const GatherSeg g1, const GatherSeg g2, const long* __restrict__ cols,
                                                                const int* __restrict__ step_dev, int step, int ncol, int B) {
  const GatherSeg g = blockIdx.y == 0 ? g0 : blockIdx.y == 1 ? g1 : g2;
  const long st = step_dev ? (long)*step_dev : (long)step;
  const long total = (long)B * g.row_len;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / g.row_len, k = i - b * g.row_len;
    g.out[i] = g.table[((st * ncol) + cols[b]) * g.row_len + k];
  }
}

}  // namespace

extern "C" {

int mf_sched_step_f32(const MfSchedArgs* a, void* stream) {
  MF_REQUIRE(a && a->x_t && a->pred && a->x_t_out && a->table && a->n > 0, MF_EINVAL, "sched_step: bad args");
  MF_REQUIRE(a->objective == 0 || a->objective == 1, MF_EINVAL, "sched_step: objective");
  MF_REQUIRE(!(a->pred_var && a->pred_uncond), MF_EUNSUPPORTED,
             "sched_step: learned variance with classifier-free guidance is unreachable in the reference (it raises)");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(MF_FAM_SCHED, s, 12.0 * a->n, 4.0 * a->n * 6);
  long blocks = (a->n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  MF_LAUNCH(sched_step_kernel, dim3((int)blocks), dim3(256), 0, s, *a);
  return check_launch("sched_step");
}

int mf_sched_step_philox_f32(const MfSchedArgs* a, uint64_t seed, int32_t draw_base, int32_t draw_stride, int64_t sample_offset, int B, int32_t* step_counter,
                             uint32_t* ticket, void* stream) {
  MF_REQUIRE(a && a->x_t && a->pred && a->x_t_out && a->table && a->n > 0 && step_counter && ticket && B > 0, MF_EINVAL, "sched_step_philox: bad args");
  MF_REQUIRE(a->objective == 0 || a->objective == 1, MF_EINVAL, "sched_step_philox: objective");
  MF_REQUIRE(!(a->pred_var && a->pred_uncond), MF_EUNSUPPORTED,
             "sched_step_philox: learned variance with classifier-free guidance is unreachable in the reference (it raises)");
  MF_REQUIRE(!a->noise_post && !a->noise_ddim, MF_EINVAL, "sched_step_philox: the noise is generated inside the launch (noise pointers must be NULL)");
  MF_REQUIRE(a->n % (4L * B) == 0, MF_EUNSUPPORTED, "sched_step_philox: elements per sample must be a multiple of 4");
  const uintptr_t al = (uintptr_t)a->x_t | (uintptr_t)a->pred | (uintptr_t)a->x_t_out | (uintptr_t)a->pred_uncond | (uintptr_t)a->pred_var | (uintptr_t)a->x0_out |
                       (uintptr_t)a->xT_out;
  MF_REQUIRE((al & 15) == 0, MF_EINVAL, "sched_step_philox: tensors must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const long quads = a->n / 4;
  ProfScope ps(MF_FAM_SCHED, s, 12.0 * a->n + 200.0 * quads, 4.0 * a->n * 4);
  long blocks = (quads + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  const PhiloxP ph{(uint32_t)(seed & 0xFFFFFFFFu), (uint32_t)(seed >> 32), draw_base, draw_stride, (long)sample_offset, quads / B, step_counter, ticket};
  MF_LAUNCH(sched_step_philox_kernel, dim3((int)blocks), dim3(256), 0, s, *a, ph);
  return check_launch("sched_step_philox");
}

int mf_gather_step_rows3_f32(const float* const* tables, const int64_t* row_lens, float* const* outs, int n_tables, const int64_t* cols, const int32_t* step_dev,
                             int32_t step, int ncol, int B, void* stream) {
  MF_REQUIRE(tables && row_lens && outs && cols && n_tables >= 1 && n_tables <= 3 && ncol > 0 && B > 0, MF_EINVAL, "gather_step_rows3: bad args");
  GatherSeg g[3] = {{nullptr, nullptr, 0}, {nullptr, nullptr, 0}, {nullptr, nullptr, 0}};
  long mx = 0;
  for (int i = 0; i < n_tables; ++i) {
    MF_REQUIRE(tables[i] && outs[i] && row_lens[i] > 0, MF_EINVAL, "gather_step_rows3: table %d", i);
    g[i] = GatherSeg{tables[i], outs[i], (long)row_lens[i]};
    if ((long)B * row_lens[i] > mx) mx = (long)B * row_lens[i];
  }
  long blocks = (mx + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  MF_LAUNCH(gather_step_rows3_kernel, dim3((int)blocks, n_tables), dim3(256), 0, (hipStream_t)stream, g[0], g[1], g[2], reinterpret_cast<const long*>(cols), step_dev, step,
            ncol, B);
  return check_launch("gather_step_rows3");
}

int mf_broadcast_from_table_f32(const float* table, const int32_t* step_dev, int32_t step, float* out, int n, void* stream) {
  MF_REQUIRE(table && out && n > 0, MF_EINVAL, "broadcast_from_table: bad args");
  MF_LAUNCH(broadcast_from_table_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, table, step_dev, step, out, n);
  return check_launch("broadcast_from_table");
}

int mf_gather_step_rows_f32(const float* table, const int64_t* cols, const int32_t* step_dev, int32_t step, int ncol, int64_t row_len, float* out,
                            int B, void* stream) {
  MF_REQUIRE(table && cols && out && ncol > 0 && row_len > 0 && B > 0, MF_EINVAL, "gather_step_rows: bad args");
  const long total = (long)B * row_len;
  long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  MF_LAUNCH(gather_step_rows_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, table, reinterpret_cast<const long*>(cols), step_dev,
                     step, ncol, (long)row_len, out, total);
  return check_launch("gather_step_rows");
}

int mf_counter_add_i32(int32_t* counter, int32_t inc, void* stream) {
  MF_REQUIRE(counter, MF_EINVAL, "counter_add: null");
  MF_LAUNCH(counter_add_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counter, inc);
  return check_launch("counter_add");
}

int mf_philox_normal_f32(float* out, uint64_t seed, int32_t draw_base, int32_t draw_stride, const int32_t* step_dev, int32_t step,
                         int64_t sample_offset, int B, int64_t per_sample, void* stream) {
  MF_REQUIRE(out && B > 0 && per_sample > 0, MF_EINVAL, "philox_normal: bad args");
  MF_REQUIRE(per_sample % 4 == 0, MF_EUNSUPPORTED, "philox_normal: per_sample must be a multiple of 4");
  MF_REQUIRE(((uintptr_t)out & 15) == 0, MF_EINVAL, "philox_normal: out must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const long quads = per_sample / 4;
  const long total = (long)B * quads;
  ProfScope ps(MF_FAM_NOISE, s, 100.0 * total, 16.0 * total);
  long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  MF_LAUNCH(philox_normal_kernel, dim3((int)blocks), dim3(256), 0, s, out, (uint32_t)(seed & 0xFFFFFFFFu), (uint32_t)(seed >> 32), draw_base,
                     draw_stride, step_dev, step, (long)sample_offset, B, quads);
  return check_launch("philox_normal");
}

}  // extern "C"
