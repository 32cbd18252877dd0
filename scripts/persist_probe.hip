// persist_probe.hip -- what a kernel boundary costs against a grid barrier inside one persistent launch (MI355X, gfx950).
// The denoise iteration is 98 DEPENDENT launches; every launch pays the platform's floor (4.6-4.9 us for a one-workgroup kernel in the
// bench trace) plus dispatch ramp and drain.  This probe runs the same chain of P dependent streaming phases two ways:
//   (a) P launches of a one-phase kernel (stream order = the dependency),
//   (b) ONE launch of 256 persistent workgroups (one per CU) with a grid barrier between phases: release fence (dirty L2 lines written
//       back: the eight XCD L2s are not coherent with each other), one agent-scope atomic per workgroup, spin, acquire fence (L2 invalidate),
// on buffers of 4 / 16 / 64 MB.  In phase p workgroup w reads the block that workgroup (w + 37 p) % G wrote in phase p - 1, so every
// phase consumes data produced on OTHER CUs and XCDs; the result is checked (every element == P).
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 -o /tmp/persist_probe scripts/persist_probe.hip && /tmp/persist_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int G = 256, T = 256;

__device__ __forceinline__ void phase_body(const float4* __restrict__ src, float4* __restrict__ dst, long per_wg4, int w, int p) {
  const int rw = (w + 37 * p) % G;   // the block another workgroup wrote in the previous phase
  const float4* s = src + (long)rw * per_wg4;
  float4* d = dst + (long)w * per_wg4;
  for (long i = threadIdx.x; i < per_wg4; i += T) {
    float4 v = s[i];
    v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
    d[i] = v;
  }
}

__global__ __launch_bounds__(T) void one_phase(const float4* src, float4* dst, long per_wg4, int p) {
  extern __shared__ char dyn_lds[];   // (size chosen at launch: 0, or the 96 KB of a convolution workgroup -- does the dispatcher care?)
  if (per_wg4 < 0) dyn_lds[threadIdx.x] = 1;
  phase_body(src, dst, per_wg4, blockIdx.x, p);
}

__global__ __launch_bounds__(T) void persistent(float4* a, float4* b, long per_wg4, int phases, unsigned* counter) {
  const int w = blockIdx.x;
  for (int p = 0; p < phases; ++p) {
    phase_body((p & 1) ? b : a, (p & 1) ? a : b, per_wg4, w, p);
    // grid barrier
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)(p + 1) * G;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
}

int main() {
  hipDeviceProp_t pr;
  CK(hipGetDeviceProperties(&pr, 0));
  printf("%s, %d CUs: chain of dependent streaming phases (read a block another workgroup wrote, +1, write), %d workgroups x %d threads\n", pr.gcnArchName,
         pr.multiProcessorCount, G, T);
  const int P = 200;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  unsigned* counter;
  CK(hipMalloc(&counter, 4));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&one_phase), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  for (long mb : {0L, 4L, 16L, 64L}) {
    const long bytes = mb ? mb << 20 : (long)G * T * 16;   // "0": one float4 per thread (pure overhead)
    const long per_wg4 = bytes / 16 / G;
    float4 *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    float ms[3];
    bool ok[3];
    for (int mode = 0; mode < 3; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {   // rep 0: warm-up
        CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes)); CK(hipMemset(counter, 0, 4));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        if (mode == 0 || mode == 2) {
          const size_t lds = mode == 2 ? 96 * 1024 : 0;
          for (int p = 0; p < P; ++p) hipLaunchKernelGGL(one_phase, dim3(G), dim3(T), lds, 0, (p & 1) ? b : a, (p & 1) ? a : b, per_wg4, p);
        } else {
          hipLaunchKernelGGL(persistent, dim3(G), dim3(T), 0, 0, a, b, per_wg4, P, counter);
        }
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms[mode], e0, e1));
      }
      std::vector<float> h(bytes / 4);
      CK(hipMemcpy(h.data(), (P & 1) ? b : a, bytes, hipMemcpyDeviceToHost));   // the last phase wrote: P even -> a
      long bad = 0;
      for (float v : h) bad += v != (float)P;
      ok[mode] = bad == 0;
    }
    printf("%5.2f MB per phase: %d launches %7.2f us per phase (%s), with 96 KB of LDS per workgroup %7.2f (%s) | one persistent launch + grid barriers %7.2f us per phase (%s)\n",
           bytes / 1048576.0, P, 1e3 * ms[0] / P, ok[0] ? "exact" : "WRONG", 1e3 * ms[2] / P, ok[2] ? "exact" : "WRONG", 1e3 * ms[1] / P, ok[1] ? "exact" : "WRONG");
    CK(hipFree(a)); CK(hipFree(b));
  }
  return 0;
}
