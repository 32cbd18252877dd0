// mfma_probe.hip -- cycle costs of the conv main-loop ingredients on gfx950 (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NACC>
__global__ void probe(float* out, long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 2 * 256 * 36; i += blockDim.x) lds[i] = (float)(i % 7) * 0.01f;
  __syncthreads();
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  const float* base = lds + (lane & 31) * 36 + 4 * (lane >> 5) + (tid >> 6) * 32 * 36 % (128 * 36);
  float a0 = lane * 0.001f, b0 = lane * 0.002f;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 3) __syncthreads();
    if (MODE >= 4) {  // LDS store of 4 x 16 B per thread (the staging write of one chunk)
      float* w = lds + 128 * 36 + (tid % 256) * 36 / 8 * 4;
      for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(w + q * 32 * 36 % (120 * 36)) = f32x4{a0, b0, a0, b0};
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f32x4 fa[2], fb[2];
      if (MODE >= 2) {
        fa[0] = *reinterpret_cast<const f32x4*>(base + kk * 8);
        fa[1] = *reinterpret_cast<const f32x4*>(base + 32 * 36 + kk * 8);
        fb[0] = *reinterpret_cast<const f32x4*>(base + 64 * 36 + kk * 8);
        fb[1] = *reinterpret_cast<const f32x4*>(base + 96 * 36 + kk * 8);
      } else {
        fa[0] = f32x4{a0, a0, a0, a0}; fa[1] = fa[0]; fb[0] = f32x4{b0, b0, b0, b0}; fb[1] = fb[0];
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int m = 0; m < 2; ++m)  // 8 MFMA per kk -> 32 per iteration
          acc[(m) % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[m & 1][s], fb[0][s], acc[(m) % NACC], 0, 0, 0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * blockDim.x + tid] = s;
  if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + (tid >> 6)] = t1 - t0;
}

template <int MODE, int NACC>
void run(const char* name, int threads, int blocks) {
  const int iters = 2000;
  float* out; long long* cyc;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipMalloc(&cyc, sizeof(long long) * blocks * threads / 64);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MODE, NACC>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<MODE, NACC>), dim3(blocks), dim3(threads), 74 * 1024, 0, out, cyc, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<MODE, NACC>), dim3(blocks), dim3(threads), 74 * 1024, 0, out, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks * threads / 64);
  hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += v; avg /= h.size();
  const double nm = 16.0 * 2 * iters;  // MFMAs per wave
  const double tf = (double)blocks * (threads / 64) * nm * 4096.0 / (ms * 1e-3) / 1e12;
  printf("%-46s thr %4d blk %4d: %8.1f memtime-ticks/iter (32 MFMA)  %6.1f TF  %.3f ms\n", name, threads, blocks, avg / iters, tf, ms);
  hipFree(out); hipFree(cyc);
}

int main() {
  run<1, 2>("MFMA only, 2 acc", 256, 256);
  run<1, 2>("MFMA only, 2 acc, 2 waves/SIMD", 512, 256);
  run<1, 1>("MFMA only, 1 acc", 256, 256);
  run<2, 2>("+ ds_read_b128 frags", 256, 256);
  run<2, 2>("+ ds_read_b128 frags, 2 waves/SIMD", 512, 256);
  run<3, 2>("+ barrier per 32 MFMA", 256, 256);
  run<3, 2>("+ barrier per 32 MFMA, 2 waves/SIMD", 512, 256);
  run<4, 2>("+ ds_write 4x16B", 256, 256);
  run<4, 2>("+ ds_write 4x16B, 2 waves/SIMD", 512, 256);
  run<4, 2>("+ ds_write, 2 waves/SIMD, 2 blocks/CU", 512, 512);
  return 0;
}
