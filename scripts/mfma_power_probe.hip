// mfma_power_probe.hip -- what the fp16 matrix pipe of an MI355X sustains when nothing but v_mfma_f32_32x32x16_f16 runs, as a function of the
// operand DATA (zeros / a constant / random fp16): the denominator the conv kernel's matrix-pipe fraction has to be read against.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_power_probe.hip -o /tmp/mfma_power_probe && /tmp/mfma_power_probe
// Every wave runs NACC independent accumulation chains from registers (no LDS, no memory traffic inside the loop), 2 waves per SIMD,
// 256 workgroups of 512 threads = one per CU.  Peak: 1024 flop / clk / SIMD = 32 cycles per instruction = 2516.8 TF at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(512) void probe(const f16x8* __restrict__ in, float* __restrict__ out, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  f16x8 a[NACC], b[NACC];
  f32x16 acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; ++k) {
    a[k] = in[(size_t)tid * 2 * NACC + 2 * k];
    b[k] = in[(size_t)tid * 2 * NACC + 2 * k + 1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k], b[k], acc[k], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NACC; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[k][r];
  out[tid] = s;
}

static float frand() { return (float)rand() / (float)RAND_MAX; }

int main(int argc, char** argv) {
  constexpr int NACC = 4, THREADS = 512, BLOCKS = 256;
  const size_t n16 = (size_t)BLOCKS * THREADS * 2 * NACC * 8;
  std::vector<_Float16> h(n16);
  f16x8* in; float* out;
  hipMalloc(&in, n16 * 2); hipMalloc(&out, sizeof(float) * BLOCKS * THREADS);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[] = {"zeros", "constant 1.0", "random normal", "random normal, pairs (hi | lo*2048) of the same values"};
  for (int mode = 0; mode < 4; ++mode) {
    srand(1);
    for (size_t i = 0; i < n16; ++i) {
      float v = 0.f;
      if (mode == 1) v = 1.f;
      if (mode >= 2) { const float u1 = frand() * 0.999f + 1e-3f, u2 = frand(); v = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2); }
      if (mode == 3 && (i / 8) % 2) { const _Float16 hi = (_Float16)v; v = (v - (float)hi) * 2048.f; }   // the lo' piece: full-range mantissa noise
      h[i] = (_Float16)v;
    }
    hipMemcpy(in, h.data(), n16 * 2, hipMemcpyHostToDevice);
    for (int iters : {400, 4000, 40000, 400000}) {   // ~25 us ... ~25 ms at full rate: does the clock hold when the burst gets long?
      if (argc > 1 && iters != 4000) continue;
      hipLaunchKernelGGL(probe<NACC>, dim3(BLOCKS), dim3(THREADS), 0, 0, in, out, iters);   // warm-up of the same length
      hipEventRecord(e0);
      hipLaunchKernelGGL(probe<NACC>, dim3(BLOCKS), dim3(THREADS), 0, 0, in, out, iters);
      hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flop = (double)BLOCKS * (THREADS / 64) * NACC * (double)iters * 32768.0;
      const double tf = flop / (ms * 1e-3) / 1e12;
      printf("%-58s iters %6d: %9.3f ms  %7.1f TF  = %4.1f %% of 2516.8\n", names[mode], iters, ms, tf, 100.0 * tf / 2516.8);
    }
  }
  // round 6: `mfma_power_probe part` -- the same bursts (random operands, 4000 iterations) from 32 / 64 / 128 / 256 workgroups: is the ~0.62 of
  // nominal a chip-wide budget (fewer active CUs would each run faster) or a per-CU limit (the per-CU rate stays)?
  if (argc > 1) {
    for (int blocks : {32, 64, 128, 192, 256}) {
      const int iters = 4000;
      hipLaunchKernelGGL(probe<NACC>, dim3(blocks), dim3(THREADS), 0, 0, in, out, iters);
      hipEventRecord(e0);
      hipLaunchKernelGGL(probe<NACC>, dim3(blocks), dim3(THREADS), 0, 0, in, out, iters);
      hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flop = (double)blocks * (THREADS / 64) * NACC * (double)iters * 32768.0;
      const double tf = flop / (ms * 1e-3) / 1e12;
      printf("random pairs, %3d workgroups (one per CU): %9.3f ms  %7.1f TF = %4.1f %% of the nominal rate of %d CUs\n", blocks, ms, tf, 100.0 * tf / (2516.8 * blocks / 256.0), blocks);
    }
  }
  return 0;
}
