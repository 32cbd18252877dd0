// pk_probe.hip -- stand-alone hazard probes for the "packed fp32 loses its low half in lanes 48..63" symptom of round 2 (gfx950).
// Each probe is the bare producer -> consumer pair of one spot of the conv epilogue, written in inline asm on fixed registers so that
// hipcc cannot pad or reorder it, run by 8 waves per workgroup on every CU for many iterations, every word checked.
//   A  ds_write_b128 data registers overwritten by the next VALU instruction (v_pk_add_f32 / v_add_f32) -- store-data WAR
//   D  the same while the other four waves of the workgroup keep the LDS busy like a co-resident workgroup in its main loop (ds_read_b128
//      streams and / or LDS-DMA), with 0 .. 8 wait states between the write and the overwrite
// build: hipcc --offload-arch=gfx950 -O2 scripts/pk_probe.hip -o scripts/pk_probe.bin ; run: scripts/pk_probe.bin [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// hist[lane][slot]: wrong words seen at (lane, register slot)
template <int PACKED>
__global__ __launch_bounds__(512) void probe_a(unsigned* hist, int iters) {
  extern __shared__ char smem[];
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned addr = wave * 4608u + (lane & 31) * 144u + (lane >> 5) * 16u;   // the staging geometry of the epilogue (PITCH = 144)
  for (int it = 0; it < iters; ++it) {
    const float p = (float)((it & 1023) * 64 + lane);   // small integers: every add below is exact
    float r[16];
    asm volatile(
        "v_mov_b32 v20, %[p]\n\tv_add_f32 v21, 0.5, %[p]\n\tv_add_f32 v22, 0.25, %[p]\n\tv_add_f32 v23, 0.125, %[p]\n\t"
        "v_mov_b32 v24, 1.0\n\tv_mov_b32 v25, 1.0\n\ts_nop 4\n\t"
        "ds_write_b128 %[a], v[20:23]\n\t"
        ".if %[pk]\n\tv_pk_add_f32 v[20:21], v[20:21], v[24:25]\n\tv_pk_add_f32 v[22:23], v[22:23], v[24:25]\n\t"
        ".else\n\tv_add_f32 v20, v20, v24\n\tv_add_f32 v21, v21, v24\n\tv_add_f32 v22, v22, v24\n\tv_add_f32 v23, v23, v24\n\t.endif\n\t"
        "ds_write_b128 %[a], v[20:23] offset:32\n\t"
        ".if %[pk]\n\tv_pk_add_f32 v[20:21], v[20:21], v[24:25]\n\tv_pk_add_f32 v[22:23], v[22:23], v[24:25]\n\t"
        ".else\n\tv_add_f32 v20, v20, v24\n\tv_add_f32 v21, v21, v24\n\tv_add_f32 v22, v22, v24\n\tv_add_f32 v23, v23, v24\n\t.endif\n\t"
        "ds_write_b128 %[a], v[20:23] offset:64\n\t"
        ".if %[pk]\n\tv_pk_add_f32 v[20:21], v[20:21], v[24:25]\n\tv_pk_add_f32 v[22:23], v[22:23], v[24:25]\n\t"
        ".else\n\tv_add_f32 v20, v20, v24\n\tv_add_f32 v21, v21, v24\n\tv_add_f32 v22, v22, v24\n\tv_add_f32 v23, v23, v24\n\t.endif\n\t"
        "ds_write_b128 %[a], v[20:23] offset:96\n\t"
        ".if %[pk]\n\tv_pk_add_f32 v[20:21], v[20:21], v[24:25]\n\tv_pk_add_f32 v[22:23], v[22:23], v[24:25]\n\t"
        ".else\n\tv_add_f32 v20, v20, v24\n\tv_add_f32 v21, v21, v24\n\tv_add_f32 v22, v22, v24\n\tv_add_f32 v23, v23, v24\n\t.endif\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "ds_read_b128 v[28:31], %[a]\n\tds_read_b128 v[32:35], %[a] offset:32\n\tds_read_b128 v[36:39], %[a] offset:64\n\tds_read_b128 v[40:43], %[a] offset:96\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_mov_b32 %[r0], v28\n\tv_mov_b32 %[r1], v29\n\tv_mov_b32 %[r2], v30\n\tv_mov_b32 %[r3], v31\n\t"
        "v_mov_b32 %[r4], v32\n\tv_mov_b32 %[r5], v33\n\tv_mov_b32 %[r6], v34\n\tv_mov_b32 %[r7], v35\n\t"
        "v_mov_b32 %[r8], v36\n\tv_mov_b32 %[r9], v37\n\tv_mov_b32 %[r10], v38\n\tv_mov_b32 %[r11], v39\n\t"
        "v_mov_b32 %[r12], v40\n\tv_mov_b32 %[r13], v41\n\tv_mov_b32 %[r14], v42\n\tv_mov_b32 %[r15], v43\n\t"
        : [r0] "=v"(r[0]), [r1] "=v"(r[1]), [r2] "=v"(r[2]), [r3] "=v"(r[3]), [r4] "=v"(r[4]), [r5] "=v"(r[5]), [r6] "=v"(r[6]), [r7] "=v"(r[7]),
          [r8] "=v"(r[8]), [r9] "=v"(r[9]), [r10] "=v"(r[10]), [r11] "=v"(r[11]), [r12] "=v"(r[12]), [r13] "=v"(r[13]), [r14] "=v"(r[14]), [r15] "=v"(r[15])
        : [a] "v"(addr), [p] "v"(p), [pk] "n"(PACKED)
        : "v20", "v21", "v22", "v23", "v24", "v25", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41",
          "v42", "v43", "memory");
    const float off[4] = {0.f, 0.5f, 0.25f, 0.125f};
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (r[4 * g + k] != p + off[k] + (float)g) atomicAdd(&hist[lane * 16 + 4 * g + k], 1u);
  }
}

// D: probe A's sequence on waves 0-3 of a workgroup (one per SIMD) while waves 4-7 of the SAME workgroup keep the LDS busy the way the main
// loop of a co-resident conv workgroup does: HAMMER bit 0 = ds_read_b128 streams, bit 1 = LDS-DMA (buffer_load_dwordx4 ... lds, 1 KB per
// instruction).  PAD = wait states between the ds_write_b128 and the instruction that overwrites its data registers.
template <int PACKED, int HAMMER, int PAD>
__global__ __launch_bounds__(512) void probe_d(unsigned* hist, const float* src, unsigned src_bytes, int iters) {
  extern __shared__ char smem[];
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= 4) {
    if (HAMMER == 0) return;
    char* region = smem + 40960 + (wave - 4) * 8192;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, src_bytes, 0x00020000);
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
      const unsigned off = ((unsigned)(blockIdx.x * 4 + wave) * 65536u + (unsigned)it * 8192u) % (src_bytes - 8192u) + lane * 16u;
      if (HAMMER & 2) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const unsigned o = off + u * 1024u;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(region + u * 1024), 16, o, 0, 0, 0);
        }
      }
      if (HAMMER & 1) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          typedef float f4 __attribute__((ext_vector_type(4)));
          const f4 v = *reinterpret_cast<volatile f4*>(region + u * 1024 + lane * 16);
          acc += v[0] + v[3];
        }
      }
      if (HAMMER & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (acc == 123.456f) hist[0] = 1;   // (keeps the reads alive)
    return;
  }
  const unsigned addr = wave * 4608u + (lane & 31) * 144u + (lane >> 5) * 16u;
  for (int it = 0; it < iters; ++it) {
    const float p = (float)((it & 1023) * 64 + lane);
    float r[16];
#define PK_CLOBBER ".if %[pad] > 0\n\ts_nop %[pad] - 1\n\t.endif\n\t" \
        ".if %[pk]\n\tv_pk_add_f32 v[20:21], v[20:21], v[24:25]\n\tv_pk_add_f32 v[22:23], v[22:23], v[24:25]\n\t" \
        ".else\n\tv_add_f32 v20, v20, v24\n\tv_add_f32 v21, v21, v24\n\tv_add_f32 v22, v22, v24\n\tv_add_f32 v23, v23, v24\n\t.endif\n\t"
    asm volatile(
        "v_mov_b32 v20, %[p]\n\tv_add_f32 v21, 0.5, %[p]\n\tv_add_f32 v22, 0.25, %[p]\n\tv_add_f32 v23, 0.125, %[p]\n\t"
        "v_mov_b32 v24, 1.0\n\tv_mov_b32 v25, 1.0\n\ts_nop 4\n\t"
        "ds_write_b128 %[a], v[20:23]\n\t" PK_CLOBBER
        "ds_write_b128 %[a], v[20:23] offset:32\n\t" PK_CLOBBER
        "ds_write_b128 %[a], v[20:23] offset:64\n\t" PK_CLOBBER
        "ds_write_b128 %[a], v[20:23] offset:96\n\t" PK_CLOBBER
        "s_waitcnt lgkmcnt(0)\n\t"
        "ds_read_b128 v[28:31], %[a]\n\tds_read_b128 v[32:35], %[a] offset:32\n\tds_read_b128 v[36:39], %[a] offset:64\n\tds_read_b128 v[40:43], %[a] offset:96\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_mov_b32 %[r0], v28\n\tv_mov_b32 %[r1], v29\n\tv_mov_b32 %[r2], v30\n\tv_mov_b32 %[r3], v31\n\t"
        "v_mov_b32 %[r4], v32\n\tv_mov_b32 %[r5], v33\n\tv_mov_b32 %[r6], v34\n\tv_mov_b32 %[r7], v35\n\t"
        "v_mov_b32 %[r8], v36\n\tv_mov_b32 %[r9], v37\n\tv_mov_b32 %[r10], v38\n\tv_mov_b32 %[r11], v39\n\t"
        "v_mov_b32 %[r12], v40\n\tv_mov_b32 %[r13], v41\n\tv_mov_b32 %[r14], v42\n\tv_mov_b32 %[r15], v43\n\t"
        : [r0] "=v"(r[0]), [r1] "=v"(r[1]), [r2] "=v"(r[2]), [r3] "=v"(r[3]), [r4] "=v"(r[4]), [r5] "=v"(r[5]), [r6] "=v"(r[6]), [r7] "=v"(r[7]),
          [r8] "=v"(r[8]), [r9] "=v"(r[9]), [r10] "=v"(r[10]), [r11] "=v"(r[11]), [r12] "=v"(r[12]), [r13] "=v"(r[13]), [r14] "=v"(r[14]), [r15] "=v"(r[15])
        : [a] "v"(addr), [p] "v"(p), [pk] "n"(PACKED), [pad] "n"(PAD)
        : "v20", "v21", "v22", "v23", "v24", "v25", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41",
          "v42", "v43", "memory");
    const float off[4] = {0.f, 0.5f, 0.25f, 0.125f};
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (r[4 * g + k] != p + off[k] + (float)g) atomicAdd(&hist[lane * 16 + 4 * g + k], 1u);
  }
}

// E: the instruction sequence of the conv epilogue around one staging store, literally: the data registers come out of two packed multiplies,
// the ADDRESS register out of a VALU add issued right in front of the ds_write_b128, and the very next instruction is a packed multiply with an
// SGPR operand that overwrites the first data pair.  VADDR = 0: the address is ready long before (as in A).  PAD: wait states after the store.
template <int VADDR, int PAD, int PACKED>
__global__ __launch_bounds__(512) void probe_e(unsigned* hist, int iters) {
  extern __shared__ char smem[];
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned addr = wave * 4608u + (lane & 31) * 144u + (lane >> 5) * 16u;
  for (int it = 0; it < iters; ++it) {
    const float p = (float)((it & 1023) * 64 + lane);
    float r[16];
#define PK_GEN(OFF)                                                                                                     \
        "v_pk_mul_f32 v[20:21], v[36:37], v[26:27] op_sel_hi:[1,0]\n\tv_pk_mul_f32 v[22:23], v[38:39], v[26:27] op_sel_hi:[1,0]\n\t" \
        ".if %[va]\n\tv_add_u32 v30, %[a], v31\n\tds_write_b128 v30, v[20:23] offset:" #OFF "\n\t"                    \
        ".else\n\tds_write_b128 %[a], v[20:23] offset:" #OFF "\n\t.endif\n\t"                                        \
        ".if %[pad] > 0\n\ts_nop %[pad] - 1\n\t.endif\n\t"                                                           \
        ".if %[pk]\n\tv_pk_mul_f32 v[20:21], v[32:33], s[20:21] op_sel_hi:[1,0]\n\tv_pk_mul_f32 v[22:23], v[34:35], s[20:21] op_sel_hi:[1,0]\n\t" \
        "v_pk_add_f32 v[36:37], v[36:37], v[20:21]\n\tv_pk_add_f32 v[38:39], v[38:39], v[22:23]\n\t"                   \
        ".else\n\tv_mul_f32 v20, s20, v32\n\tv_mul_f32 v21, s20, v33\n\tv_mul_f32 v22, s20, v34\n\tv_mul_f32 v23, s20, v35\n\t" \
        "v_add_f32 v36, v36, v20\n\tv_add_f32 v37, v37, v21\n\tv_add_f32 v38, v38, v22\n\tv_add_f32 v39, v39, v23\n\t.endif\n\t"
    asm volatile(
        "v_mov_b32 v36, %[p]\n\tv_add_f32 v37, 0.5, %[p]\n\tv_add_f32 v38, 0.25, %[p]\n\tv_add_f32 v39, 0.125, %[p]\n\t"
        "v_mov_b32 v26, 1.0\n\tv_mov_b32 v27, 0\n\tv_mov_b32 v31, 0\n\t"
        "v_mov_b32 v32, 0x45000000\n\tv_mov_b32 v33, 0x45000000\n\tv_mov_b32 v34, 0x45000000\n\tv_mov_b32 v35, 0x45000000\n\t"   // 2048.0
        "s_mov_b32 s20, 0x3a000000\n\ts_movk_i32 s21, 0x90\n\ts_nop 4\n\t"                                              // 2^-11
        PK_GEN(0) PK_GEN(32) PK_GEN(64) PK_GEN(96)
        "s_waitcnt lgkmcnt(0)\n\t"
        "ds_read_b128 v[40:43], %[a]\n\tds_read_b128 v[44:47], %[a] offset:32\n\tds_read_b128 v[48:51], %[a] offset:64\n\tds_read_b128 v[52:55], %[a] offset:96\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_mov_b32 %[r0], v40\n\tv_mov_b32 %[r1], v41\n\tv_mov_b32 %[r2], v42\n\tv_mov_b32 %[r3], v43\n\t"
        "v_mov_b32 %[r4], v44\n\tv_mov_b32 %[r5], v45\n\tv_mov_b32 %[r6], v46\n\tv_mov_b32 %[r7], v47\n\t"
        "v_mov_b32 %[r8], v48\n\tv_mov_b32 %[r9], v49\n\tv_mov_b32 %[r10], v50\n\tv_mov_b32 %[r11], v51\n\t"
        "v_mov_b32 %[r12], v52\n\tv_mov_b32 %[r13], v53\n\tv_mov_b32 %[r14], v54\n\tv_mov_b32 %[r15], v55\n\t"
        : [r0] "=v"(r[0]), [r1] "=v"(r[1]), [r2] "=v"(r[2]), [r3] "=v"(r[3]), [r4] "=v"(r[4]), [r5] "=v"(r[5]), [r6] "=v"(r[6]), [r7] "=v"(r[7]),
          [r8] "=v"(r[8]), [r9] "=v"(r[9]), [r10] "=v"(r[10]), [r11] "=v"(r[11]), [r12] "=v"(r[12]), [r13] "=v"(r[13]), [r14] "=v"(r[14]), [r15] "=v"(r[15])
        : [a] "v"(addr), [p] "v"(p), [va] "n"(VADDR), [pad] "n"(PAD), [pk] "n"(PACKED)
        : "v20", "v21", "v22", "v23", "v26", "v27", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43",
          "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "s20", "s21", "memory");
    const float off[4] = {0.f, 0.5f, 0.25f, 0.125f};
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (r[4 * g + k] != p + off[k] + (float)g) atomicAdd(&hist[lane * 16 + 4 * g + k], 1u);
  }
}

// F: the instruction form every failing build of conv_f16x2_kernel has and every clean one lacks (scripts/pk_hunt.py, profiles/r03_pk_repro.txt):
//      v_pk_mul_f32 v[d:d+1], v[d:d+1], v[34:35] op_sel:[0,1]      low result = v[d] * v35 (the HIGH half of src1), high result = v[d+1] * v35
// eight of them back to back on waves 0-3 of a workgroup (one per SIMD), while waves 4-7 (the second wave of each SIMD) are idle (AGG 0), issue
// matrix instructions back to back (1), stream ds_read_b128 (2), or both (3) -- the mix a co-resident conv workgroup runs.  The other members
// of the family map which operand selections share the fault; v[34:35] = (100000.0, 3.0), v[36:37] = (3.0, 100000.0), v[38:39] = (0, 0).
template <int AGG>
__device__ __forceinline__ void aggressor(unsigned* hist, char* smem, unsigned lane, unsigned wave, int iters) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef float f16v __attribute__((ext_vector_type(16)));
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  f16v acc0 = {}, acc1 = {};
  h8 a, b;
  for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(0.001f * (float)(lane + k)); b[k] = (_Float16)(0.002f * (float)(lane ^ k)); }
  float sink = 0.f;
  for (int it = 0; it < iters * 4; ++it) {
    if (AGG & 1) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc1, 0, 0, 0);
    }
    if (AGG & 2) {
      const f4 v = *reinterpret_cast<volatile f4*>(smem + 40960 + (wave - 4) * 8192 + ((it & 7) * 1024) + lane * 16);
      sink += v[0];
    }
  }
  if (acc0[0] + acc1[5] + sink == 123.456f) hist[0] = 1;
}
// INSTR: the instruction text with D = the pair v[d:d+1] it updates; WANT_LO / WANT_HI: the exact results from xl = v[d], xh = v[d+1]
#define PK_FAMILY(NAME, INSTR, WANT_LO, WANT_HI)                                                                        \
  template <int AGG>                                                                                                    \
  __global__ __launch_bounds__(512) void NAME(unsigned* hist, float* sample, int iters) {                               \
    extern __shared__ char smem[];                                                                                      \
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;                                                    \
    if (wave >= 4) { if (AGG) aggressor<AGG>(hist, smem, lane, wave, iters); return; }                                  \
    for (int it = 0; it < iters; ++it) {                                                                                \
      const float x = (float)(((it & 255) + 1) * 64 + lane);                                                            \
      float r[16];                                                                                                      \
      asm volatile(                                                                                                     \
          "v_mov_b32 v2, %[x]\n\tv_add_f32 v3, 1.0, %[x]\n\tv_add_f32 v4, 2.0, %[x]\n\tv_add_f32 v5, 3.0, %[x]\n\t"       \
          "v_add_f32 v6, 4.0, %[x]\n\tv_add_f32 v7, 5.0, %[x]\n\tv_add_f32 v8, 6.0, %[x]\n\tv_add_f32 v9, 7.0, %[x]\n\t"  \
          "v_add_f32 v10, 8.0, %[x]\n\tv_add_f32 v11, 9.0, %[x]\n\tv_add_f32 v12, 10.0, %[x]\n\tv_add_f32 v13, 11.0, %[x]\n\t" \
          "v_add_f32 v14, 12.0, %[x]\n\tv_add_f32 v15, 13.0, %[x]\n\tv_add_f32 v16, 14.0, %[x]\n\tv_add_f32 v17, 15.0, %[x]\n\t" \
          "v_mov_b32 v34, 0x47c35000\n\tv_mov_b32 v35, 0x40400000\n\tv_mov_b32 v36, 0x40400000\n\tv_mov_b32 v37, 0x47c35000\n\t" \
          "v_mov_b32 v38, 0\n\tv_mov_b32 v39, 0\n\ts_nop 4\n\t"                                                        \
          INSTR("v[2:3]") INSTR("v[4:5]") INSTR("v[6:7]") INSTR("v[8:9]") INSTR("v[10:11]") INSTR("v[12:13]") INSTR("v[14:15]") INSTR("v[16:17]") \
          "s_nop 4\n\t"                                                                                                 \
          "v_mov_b32 %[r0], v2\n\tv_mov_b32 %[r1], v3\n\tv_mov_b32 %[r2], v4\n\tv_mov_b32 %[r3], v5\n\t"                 \
          "v_mov_b32 %[r4], v6\n\tv_mov_b32 %[r5], v7\n\tv_mov_b32 %[r6], v8\n\tv_mov_b32 %[r7], v9\n\t"                 \
          "v_mov_b32 %[r8], v10\n\tv_mov_b32 %[r9], v11\n\tv_mov_b32 %[r10], v12\n\tv_mov_b32 %[r11], v13\n\t"           \
          "v_mov_b32 %[r12], v14\n\tv_mov_b32 %[r13], v15\n\tv_mov_b32 %[r14], v16\n\tv_mov_b32 %[r15], v17\n\t"         \
          : [r0] "=v"(r[0]), [r1] "=v"(r[1]), [r2] "=v"(r[2]), [r3] "=v"(r[3]), [r4] "=v"(r[4]), [r5] "=v"(r[5]), [r6] "=v"(r[6]), [r7] "=v"(r[7]), \
            [r8] "=v"(r[8]), [r9] "=v"(r[9]), [r10] "=v"(r[10]), [r11] "=v"(r[11]), [r12] "=v"(r[12]), [r13] "=v"(r[13]), [r14] "=v"(r[14]), [r15] "=v"(r[15]) \
          : [x] "v"(x)                                                                                                  \
          : "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v34", "v35", "v36", "v37", "v38", "v39", "memory"); \
      _Pragma("unroll") for (int k = 0; k < 16; k += 2) {                                                               \
        const float xl = x + (float)k, xh = x + (float)(k + 1);                                                         \
        const float wl = (WANT_LO), wh = (WANT_HI);                                                                     \
        if (r[k] != wl) { atomicAdd(&hist[lane * 16 + k], 1u); sample[0] = r[k]; sample[1] = wl; sample[2] = xl; sample[3] = xh; } \
        if (r[k + 1] != wh) { atomicAdd(&hist[lane * 16 + k + 1], 1u); sample[4] = r[k + 1]; sample[5] = wh; sample[6] = xl; sample[7] = xh; } \
      }                                                                                                                 \
    }                                                                                                                   \
  }
#define I_MUL_01(D) "v_pk_mul_f32 " D ", " D ", v[34:35] op_sel:[0,1]\n\t"
#define I_MUL_HI10(D) "v_pk_mul_f32 " D ", " D ", v[36:37] op_sel_hi:[1,0]\n\t"
#define I_MUL_S0_10(D) "v_pk_mul_f32 " D ", v[34:35], " D " op_sel:[1,0]\n\t"
#define I_ADD_01(D) "v_pk_add_f32 " D ", " D ", v[34:35] op_sel:[0,1]\n\t"
#define I_FMA_010(D) "v_pk_fma_f32 " D ", " D ", v[34:35], v[38:39] op_sel:[0,1,0]\n\t"
#define I_FMA_100(D) "v_pk_fma_f32 " D ", v[34:35], " D ", v[38:39] op_sel:[1,0,0]\n\t"
#define I_MUL_SWAP(D) "v_pk_mul_f32 " D ", " D ", v[34:35] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
#define I_MUL_01_NOP(D) "v_pk_mul_f32 " D ", " D ", v[34:35] op_sel:[0,1]\n\ts_nop 3\n\t"
PK_FAMILY(pf_mul_01, I_MUL_01, xl * 3.0f, xh * 3.0f)
PK_FAMILY(pf_mul_hi10, I_MUL_HI10, xl * 3.0f, xh * 3.0f)
PK_FAMILY(pf_mul_s0_10, I_MUL_S0_10, xl * 3.0f, xh * 3.0f)
PK_FAMILY(pf_add_01, I_ADD_01, xl + 3.0f, xh + 3.0f)
PK_FAMILY(pf_fma_010, I_FMA_010, xl * 3.0f, xh * 3.0f)
PK_FAMILY(pf_fma_100, I_FMA_100, xl * 3.0f, xh * 3.0f)
PK_FAMILY(pf_mul_swap, I_MUL_SWAP, xl * 3.0f, xh * 100000.0f)
PK_FAMILY(pf_mul_01_nop, I_MUL_01_NOP, xl * 3.0f, xh * 3.0f)

static void report(const char* name, const std::vector<unsigned>& h, int rows, int cols, const char* rowname, const char* colname) {
  unsigned long total = 0;
  for (unsigned v : h) total += v;
  printf("%s: %lu wrong words\n", name, total);
  if (!total) return;
  for (int r = 0; r < rows; ++r) {
    unsigned long s = 0;
    for (int c = 0; c < cols; ++c) s += h[(size_t)r * cols + c];
    if (!s) continue;
    printf("  %s %d:", rowname, r);
    for (int c = 0; c < cols; ++c) if (h[(size_t)r * cols + c]) printf(" %s%d=%u", colname, c, h[(size_t)r * cols + c]);
    printf("\n");
  }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  hipDeviceProp_t pr;
  CK(hipGetDeviceProperties(&pr, 0));
  const int cus = pr.multiProcessorCount;
  printf("device %s, %d CUs, %d iterations per wave\n", pr.gcnArchName, cus, iters);
  unsigned* d;
  const size_t words = 64 * 64;
  CK(hipMalloc(&d, words * 4));
  std::vector<unsigned> h(words);
  auto run = [&](const char* name, auto launch, int rows, int cols, const char* rn, const char* cn) {
    CK(hipMemset(d, 0, words * 4));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d, words * 4, hipMemcpyDeviceToHost));
    report(name, h, rows, cols, rn, cn);
  };
  const size_t lds = 8 * 4608;
  run("A  ds_write_b128 data overwritten by the next v_pk_add_f32 (8 waves/WG, 1 WG/CU)", [&] { hipLaunchKernelGGL(probe_a<1>, dim3(cus), dim3(512), 96 * 1024, 0, d, iters); }, 64, 16, "lane", "slot");
  run("A  ... same, 4 WG/CU", [&] { hipLaunchKernelGGL(probe_a<1>, dim3(cus * 4), dim3(512), lds, 0, d, iters); }, 64, 16, "lane", "slot");
  run("A' ds_write_b128 data overwritten by the next v_add_f32 x 4 (1 WG/CU)", [&] { hipLaunchKernelGGL(probe_a<0>, dim3(cus), dim3(512), 96 * 1024, 0, d, iters); }, 64, 16, "lane", "slot");
  run("A' ... same, 4 WG/CU", [&] { hipLaunchKernelGGL(probe_a<0>, dim3(cus * 4), dim3(512), lds, 0, d, iters); }, 64, 16, "lane", "slot");
  float* src;
  const unsigned src_bytes = 8u << 20;
  CK(hipMalloc(&src, src_bytes));
  CK(hipMemset(src, 0, src_bytes));
#define RUN_D(PK, HAM, PAD, TEXT) \
  run(TEXT, [&] { hipLaunchKernelGGL((probe_d<PK, HAM, PAD>), dim3(cus), dim3(512), 96 * 1024, 0, d, src, src_bytes, iters); }, 64, 16, "lane", "slot")
  RUN_D(1, 0, 0, "D  packed clobber, waves 4-7 idle");
  RUN_D(1, 1, 0, "D  packed clobber, waves 4-7 stream ds_read_b128");
  RUN_D(1, 2, 0, "D  packed clobber, waves 4-7 stream LDS-DMA (buffer_load_dwordx4 ... lds)");
  RUN_D(1, 3, 0, "D  packed clobber, waves 4-7 stream LDS-DMA + ds_read_b128");
  RUN_D(0, 3, 0, "D' v_add_f32 x 4 clobber, waves 4-7 stream LDS-DMA + ds_read_b128");
  RUN_D(1, 3, 1, "D  packed clobber after 1 wait state, LDS-DMA + ds_read_b128");
  RUN_D(1, 3, 2, "D  packed clobber after 2 wait states, LDS-DMA + ds_read_b128");
  RUN_D(1, 3, 4, "D  packed clobber after 4 wait states, LDS-DMA + ds_read_b128");
  RUN_D(1, 3, 8, "D  packed clobber after 8 wait states, LDS-DMA + ds_read_b128");
#define RUN_E(VA, PAD, PK, GRID, LDS, TEXT) \
  run(TEXT, [&] { hipLaunchKernelGGL((probe_e<VA, PAD, PK>), dim3(GRID), dim3(512), LDS, 0, d, iters); }, 64, 16, "lane", "slot")
  RUN_E(0, 0, 1, cus, 96 * 1024, "E  epilogue sequence, address ready early, packed overwrite right behind the store");
  RUN_E(1, 0, 1, cus, 96 * 1024, "E  epilogue sequence, address from a VALU add right in front of the store, packed overwrite right behind it");
  RUN_E(1, 0, 1, cus * 4, lds, "E  ... same, 4 WG/CU");
  RUN_E(1, 1, 1, cus, 96 * 1024, "E  ... one wait state (s_nop 0) between the store and the packed overwrite");
  RUN_E(1, 0, 0, cus, 96 * 1024, "E' ... unpacked overwrite (v_mul_f32 x 4) right behind the store");
  RUN_E(1, 0, 0, cus * 4, lds, "E' ... same, 4 WG/CU");
  float* sample;
  CK(hipMalloc(&sample, 64));
#define RUN_F(KERNEL, AGG, GRID, LDS, TEXT)                                                                             \
  {                                                                                                                     \
    CK(hipMemset(sample, 0, 64));                                                                                       \
    run(TEXT, [&] { hipLaunchKernelGGL((KERNEL<AGG>), dim3(GRID), dim3(512), LDS, 0, d, sample, iters); }, 64, 16, "lane", "result register"); \
    float hs[8];                                                                                                        \
    CK(hipMemcpy(hs, sample, 32, hipMemcpyDeviceToHost));                                                               \
    if (hs[1] != 0.f) printf("  a wrong LOW result: got %.9g, want %.9g (low source %.9g, high source %.9g)\n", hs[0], hs[1], hs[2], hs[3]); \
    if (hs[5] != 0.f) printf("  a wrong HIGH result: got %.9g, want %.9g (low source %.9g, high source %.9g)\n", hs[4], hs[5], hs[6], hs[7]); \
  }
  RUN_F(pf_mul_01, 0, cus, 96 * 1024, "F  v_pk_mul_f32 D, D, v[34:35] op_sel:[0,1] x 8; second wave of every SIMD idle");
  RUN_F(pf_mul_01, 1, cus, 96 * 1024, "F  ... second wave issues v_mfma_f32_32x32x16_f16 back to back");
  RUN_F(pf_mul_01, 2, cus, 96 * 1024, "F  ... second wave streams ds_read_b128");
  RUN_F(pf_mul_01, 3, cus, 96 * 1024, "F  ... second wave: MFMA + ds_read_b128");
  RUN_F(pf_mul_01, 3, cus * 2, 72 * 1024, "F  ... MFMA + ds_read_b128, two workgroups per CU");
  RUN_F(pf_mul_01_nop, 3, cus, 96 * 1024, "F  the same with s_nop 3 between the eight instructions");
  RUN_F(pf_mul_hi10, 3, cus * 2, 72 * 1024, "F1 control: v_pk_mul_f32 D, D, v[36:37] op_sel_hi:[1,0] (multiplier in the LOW half, broadcast up)");
  RUN_F(pf_mul_s0_10, 3, cus, 96 * 1024, "F2 v_pk_mul_f32 D, v[34:35], D op_sel:[1,0] (src0's high half feeds the low result)");
  RUN_F(pf_add_01, 3, cus, 96 * 1024, "F3 v_pk_add_f32 D, D, v[34:35] op_sel:[0,1]");
  RUN_F(pf_fma_010, 3, cus, 96 * 1024, "F4 v_pk_fma_f32 D, D, v[34:35], 0 op_sel:[0,1,0]");
  RUN_F(pf_fma_100, 3, cus, 96 * 1024, "F5 v_pk_fma_f32 D, v[34:35], D, 0 op_sel:[1,0,0] (the form hipcc emits in conv.hip)");
  RUN_F(pf_mul_swap, 3, cus, 96 * 1024, "F6 v_pk_mul_f32 D, D, v[34:35] op_sel:[0,1] op_sel_hi:[1,0] (halves swapped)");
  CK(hipFree(sample));
  CK(hipFree(src));
  CK(hipFree(d));
  return 0;
}
