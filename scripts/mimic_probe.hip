// mimic_probe.hip -- the split-mode conv main loop reduced to its ingredients (8 waves, 1 workgroup per CU, 48 bf16 MFMAs per wave
// and chunk on 4 accumulators) to find what keeps the matrix pipe at ~60 %.  Flags: 1 barrier per chunk, 2 the 24 fragment
// ds_read_b128, 4 the LDS stores (6 b128 + 6 b64), 8 eight buffer loads, 16 ~60 VALU, 32 operands change every MFMA (as in the
// kernel) instead of one fixed pair.   build: hipcc --offload-arch=gfx950 -O3 scripts/mimic_probe.hip -o scripts/mimic_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int LDK = 52;

template <int F>
__global__ __launch_bounds__(512) void mimic(const float* __restrict__ g, float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 384 * LDK; i += 512) lds[i] = 0.001f * (i % 97);
  __syncthreads();
  const float* Aw = lds + ((wave >> 2) * 64) * LDK + (lane & 31) * LDK + 4 * (lane >> 5);
  const float* Bw = lds + 128 * LDK + ((wave & 3) * 64) * LDK + (lane & 31) * LDK + 4 * (lane >> 5);
  float* sw = lds + (tid >> 2) * LDK + (tid & 3) * 4;
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  bf16x8 fa[2][2][3], fb[2][2][3];
  for (int s = 0; s < 2; ++s) for (int i = 0; i < 2; ++i) for (int c = 0; c < 3; ++c) {
    u32x4 v = {0x3f803f80u + lane + s, 0x3f803f80u + i, 0x3f803f80u + c, 0x3f803f80u};
    fa[s][i][c] = __builtin_bit_cast(bf16x8, v); fb[s][i][c] = __builtin_bit_cast(bf16x8, v);
  }
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g), 0, 1u << 24, 0x00020000);
  u32x4 rg[8];
  for (int q = 0; q < 8; ++q) rg[q] = u32x4{1u, 2u, 3u, (unsigned)q};
  float vx[8];
  for (int q = 0; q < 8; ++q) vx[q] = 1.0f + lane * 0.01f + q;
  unsigned goff = (unsigned)(blockIdx.x * 512 + tid) * 16u;
  for (int it = 0; it < iters; ++it) {
    if (F & 1) __syncthreads();
    if (F & 2) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          fa[0][i][c] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(Aw + i * 32 * LDK + c * 16));
          fb[0][i][c] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(Bw + i * 32 * LDK + c * 16));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 0; n < 48; ++n) {
      constexpr int kCA[6] = {2, 0, 1, 1, 0, 0}, kCB[6] = {0, 2, 1, 0, 1, 0};
      const int j_ = n % 2, i_ = (n / 2) % 2, t_ = (n / 4) % 6, s_ = n / 24;
      if (F & 32) acc[i_ * 2 + j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s_][i_][kCA[t_]], fb[s_][j_][kCB[t_]], acc[i_ * 2 + j_], 0, 0, 0);
      else        acc[i_ * 2 + j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][0][0], fb[0][0][0], acc[i_ * 2 + j_], 0, 0, 0);
      if ((F & 128) && n < 12) {  // second-step fragments one read per slot
        const int c = n % 3, w = n / 3;
        if (w < 2) fa[1][w & 1][c] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(Aw + (w & 1) * 32 * LDK + c * 16 + 8));
        else       fb[1][w & 1][c] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(Bw + (w & 1) * 32 * LDK + c * 16 + 8));
      } else if ((F & 2) && !(F & 128) && n < 4) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          if (n < 2) fa[1][n & 1][c] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(Aw + (n & 1) * 32 * LDK + c * 16 + 8));
          else       fb[1][n & 1][c] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(Bw + (n & 1) * 32 * LDK + c * 16 + 8));
        }
      }
      if ((F & 64) && n >= 4 && n < 34) {  // the same 60 VALU as 2 per slot x 30 slots
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          const int q = (n + v) % 8;
          const unsigned b = __float_as_uint(vx[q]);
          vx[q] = vx[q] - __uint_as_float(b & 0xffff0000u) + 1.0f;
        }
      }
      if ((F & 16) && n >= 4 && n < 16) {  // ~5 VALU per slot x 12 slots
#pragma unroll
        for (int v = 0; v < 5; ++v) {
          const int q = (n + v) % 8;
          const unsigned b = __float_as_uint(vx[q]);
          vx[q] = vx[q] - __uint_as_float(b & 0xffff0000u) + 1.0f;
        }
      }
      if ((F & 256) && n >= 16 && n < 22) {
        const int q = n - 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (size_t)((blockIdx.x * 512 + tid) * 4 + ((it * 6 + q) & 63) * 8192 * 4 % (1 << 21))),
                                         (__attribute__((address_space(3))) void*)(lds + 128 * LDK + wave * 256 + q * 2048), 16, 0, 0);
        *reinterpret_cast<u32x2*>(sw + 64 * LDK + 128 * LDK * (q & 1) + (q >> 1) * 16) = u32x2{rg[q].x, __float_as_uint(vx[q])};
      } else if ((F & 4) && n >= 16 && n < 22) {
        const int q = n - 16;
        *reinterpret_cast<u32x4*>(sw + 128 * LDK * (q & 1) + (q >> 1) * 16) = rg[q];
        *reinterpret_cast<u32x2*>(sw + 64 * LDK + 128 * LDK * (q & 1) + (q >> 1) * 16) = u32x2{rg[q].x, __float_as_uint(vx[q])};
      }
      if ((F & 8) && n >= 24 && n < ((F & 256) ? 26 : 32)) {
        rg[n - 24] = __builtin_amdgcn_raw_buffer_load_b128(rs, goff + (unsigned)((it * 8 + (n - 24)) & 63) * 8192u * 16u % (1u << 23), 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0;
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  for (int q = 0; q < 8; ++q) s += vx[q] + __uint_as_float(rg[q].x);
  out[blockIdx.x * 512 + tid] = s;
}

template <int F>
void run(const float* g, float* out) {
  const int iters = 2000, blocks = 256;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mimic<F>), hipFuncAttributeMaxDynamicSharedMemorySize, 159744);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((mimic<F>), dim3(blocks), dim3(512), 159744, 0, g, out, 20);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((mimic<F>), dim3(blocks), dim3(512), 159744, 0, g, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double per_chunk_ns = ms * 1e6 / iters, ideal_ns = 96 * 32 / 2.0;  // 2 waves per SIMD x 48 MFMAs x 32 cycles at 2.0 GHz
  printf("flags %2d (%s%s%s%s%s%s): %.0f ns per chunk = %.0f %% matrix-pipe busy at 2.0 GHz (%.0f TF bf16)\n", F, F & 1 ? "barrier " : "", F & 2 ? "fragreads " : "",
         F & 256 ? "W-by-glds " : (F & 4 ? "ldswrites " : ""), F & 8 ? "gloads " : "", F & 16 ? "valu5x12 " : (F & 64 ? "valu2x30 " : ""), F & 128 ? "operands reads1/slot " : (F & 32 ? "operands " : ""), per_chunk_ns, 100.0 * ideal_ns / per_chunk_ns,
         2.0 * 32 * 32 * 16 * 48 * 8 * 256.0 * iters / (ms * 1e-3) * 1e-12);
}

int main() {
  float *g, *out;
  (void)hipMalloc(&g, 1u << 25); (void)hipMemset(g, 0, 1u << 25);
  (void)hipMalloc(&out, 256 * 512 * 4);
  run<32>(g, out); run<111>(g, out); run<367>(g, out); run<110>(g, out); run<366>(g, out);
  return 0;
}
