// pk_repro_min.hip -- gfx950 (MI355X, ROCm 7.2): a packed fp32 VALU instruction whose LOW result takes the HIGH half of src1
//     v_pk_mul_f32 v[2:3], v[2:3], v[34:35] op_sel:[0,1]        ; v2 = v2 * v35, v3 = v3 * v35
// reads that operand as 0.0 in lanes 48..63 (only there, only for the low result) now and then -- when the OTHER wave of the same SIMD issues
// matrix instructions while LDS reads return to it.  Nothing else of the conv kernel is needed.  The control (multiplier in the LOW half,
// broadcast up with op_sel_hi:[1,0]) never fails.  hipcc 7.2 forms the failing operand selection by itself when it SLP-packs scalar fp32
// code (conv_f16x2_kernel: "(acc + cross / 2048) * scale" with the scale in the odd register of a pair); its hazard recognizer knows nothing.
// build: hipcc --offload-arch=gfx950 -O2 scripts/pk_repro_min.hip -o pk_repro_min ; run: ./pk_repro_min      (profiles/r03_pk_repro.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int HIGH_HALF>   // 1: multiplier 1.0 in v35, selected with op_sel:[0,1]; 0: multiplier in v36, op_sel_hi:[1,0]
__global__ __launch_bounds__(512) void repro(unsigned* wrong /*[64 lanes][2 halves]*/, int iters) {
  __shared__ f4 lds[4 * 8 * 64];
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= 4) {   // waves 4..7 share the SIMDs of waves 0..3: MFMAs back to back + a stream of ds_read_b128
    f16v acc = {};
    h8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(0.001f * (float)(lane + k)); b[k] = (_Float16)(0.002f * (float)(lane ^ k)); }
    float sink = 0.f;
    for (int it = 0; it < 4 * iters; ++it) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
      sink += (*reinterpret_cast<volatile f4*>(&lds[((wave - 4) * 8 + (it & 7)) * 64 + lane]))[0];
    }
    if (acc[0] + sink == 123.456f) wrong[0] = 1;
    return;
  }
  for (int it = 0; it < iters; ++it) {
    const float x = (float)(((it & 255) + 1) * 64 + lane);
    float lo, hi;
    asm volatile(
        "v_mov_b32 v2, %[x]\n\tv_add_f32 v3, 1.0, %[x]\n\t"
        "v_mov_b32 v34, 0x47c35000\n\tv_mov_b32 v35, 1.0\n\t"            // v[34:35] = (100000.0, 1.0)
        "v_mov_b32 v36, 1.0\n\tv_mov_b32 v37, 0x47c35000\n\t"            // v[36:37] = (1.0, 100000.0)
        "s_nop 4\n\t"
        ".rept 8\n\t"                                                    // eight multiplications by 1.0 in a row: one bad read leaves 0
        ".if %[hh]\n\tv_pk_mul_f32 v[2:3], v[2:3], v[34:35] op_sel:[0,1]\n\t"
        ".else\n\tv_pk_mul_f32 v[2:3], v[2:3], v[36:37] op_sel_hi:[1,0]\n\t.endif\n\t"
        ".endr\n\t"
        "s_nop 4\n\tv_mov_b32 %[lo], v2\n\tv_mov_b32 %[hi], v3\n\t"
        : [lo] "=v"(lo), [hi] "=v"(hi) : [x] "v"(x), [hh] "n"(HIGH_HALF) : "v2", "v3", "v34", "v35", "v36", "v37");
    if (lo != x) atomicAdd(&wrong[lane * 2], 1u);
    if (hi != x + 1.f) atomicAdd(&wrong[lane * 2 + 1], 1u);
  }
}

int main() {
  hipDeviceProp_t pr;
  (void)hipGetDeviceProperties(&pr, 0);
  unsigned *d, h[128];
  (void)hipMalloc(&d, sizeof(h));
  const int iters = 100000;
  printf("%s, %d CUs: %d x 8 v_pk_mul_f32 (by 1.0) per wave, 4 waves per CU checked\n", pr.gcnArchName, pr.multiProcessorCount, iters);
  for (int form = 1; form >= 0; --form) {
    (void)hipMemset(d, 0, sizeof(h));
    if (form) hipLaunchKernelGGL(repro<1>, dim3(pr.multiProcessorCount), dim3(512), 0, 0, d, iters);
    else hipLaunchKernelGGL(repro<0>, dim3(pr.multiProcessorCount), dim3(512), 0, 0, d, iters);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long lo_lanes[4] = {}, hi_all = 0;
    for (int l = 0; l < 64; ++l) { lo_lanes[l / 16] += h[2 * l]; hi_all += h[2 * l + 1]; }
    printf("%-62s wrong LOW results in lanes 0-15 / 16-31 / 32-47 / 48-63: %lu / %lu / %lu / %lu; wrong HIGH results: %lu\n",
           form ? "v_pk_mul_f32 v[2:3], v[2:3], v[34:35] op_sel:[0,1]" : "v_pk_mul_f32 v[2:3], v[2:3], v[36:37] op_sel_hi:[1,0] (control)",
           lo_lanes[0], lo_lanes[1], lo_lanes[2], lo_lanes[3], hi_all);
  }
  (void)hipFree(d);
  return 0;
}
