// issue_probe.hip -- how many plain VALU instructions hide behind one v_mfma_f32_32x32x16_bf16 on gfx950, with one and with
// two waves per SIMD (run on the GPU box; build: hipcc --offload-arch=gfx950 -O3 scripts/issue_probe.hip -o scripts/issue_probe.bin)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NV, int KIND>
__global__ __launch_bounds__(512) void probe(float* out, long long* cyc, int iters) {
  const int tid = threadIdx.x, lane = tid & 63;
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  u32x4 au = {0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, bu = {0x3c003c00u, 0x3c003c00u + lane, 0x3c003c00u, 0x3c003c00u};
  bf16x8 a = __builtin_bit_cast(bf16x8, au), b = __builtin_bit_cast(bf16x8, bu);
  unsigned x[12];
  for (int i = 0; i < 12; ++i) x[i] = 0x3f800000u + lane * 17 + i;
  float f[12];
  for (int i = 0; i < 12; ++i) f[i] = 1.0f + lane * 0.01f + i;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (u & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
      else       asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if (KIND == 0) asm volatile("v_and_b32 %0, 0xffff0fff, %0" : "+v"(x[v % 12]));
        if (KIND == 1) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[v % 12]) : "v"(f[(v + 5) % 12]));
        if (KIND == 2) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[v % 12]) : "v"(x[(v + 5) % 12]), "s"(0x07060302u));
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  for (int i = 0; i < 12; ++i) s += __uint_as_float(x[i]) + f[i];
  out[blockIdx.x * blockDim.x + tid] = s;
  if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + (tid >> 6)] = t1 - t0;
}

template <int NV, int KIND>
void run(int threads) {
  const int iters = 4000, blocks = 256;
  float* out; long long* cyc;
  (void)hipMalloc(&out, sizeof(float) * blocks * threads);
  (void)hipMalloc(&cyc, sizeof(long long) * blocks * threads / 64);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<NV, KIND>), dim3(blocks), dim3(threads), 0, 0, out, cyc, 10);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((probe<NV, KIND>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long c0; (void)hipMemcpy(&c0, cyc, sizeof(c0), hipMemcpyDeviceToHost);
  const double mfma_per_simd = (double)iters * 8 * (threads / 256);
  printf("kind %d  NV %2d  waves/SIMD %d : %.1f ns per MFMA-slot per SIMD, %.1f memtime ticks per slot  (%.0f TF bf16)\n", KIND, NV, threads / 256,
         ms * 1e6 / mfma_per_simd, (double)c0 / (iters * 8) / (threads / 256), 2.0 * 32 * 32 * 16 * mfma_per_simd * 1024 / (ms * 1e-3) * 1e-12);
  (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
  for (int th : {256, 512}) {
    run<0, 0>(th); run<2, 0>(th); run<4, 0>(th); run<6, 0>(th); run<8, 0>(th); run<12, 0>(th); run<16, 0>(th);
    run<6, 1>(th); run<12, 1>(th); run<6, 2>(th); run<12, 2>(th);
  }
  return 0;
}
