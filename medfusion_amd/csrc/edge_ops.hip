// edge_ops.hip -- per-row scaled combinations (the scheduler's tensor API with a timestep PER ROW) and image egress.
// Streaming passes over small tensors; bit-exact against ATen / numpy elementwise chains (library built with -ffp-contract=off).
#include "common.h"

using namespace mf;

namespace {

// out[b][i] = clamp?( (a[b]*x[b][i] + c[b]*y[b][i]) / d[b] );  y / c / d optional.  Each product and the sum are rounded
// separately: a*x - b*y of gaussian_scheduler.py:121 is evaluated as a*x + (-b)*y, bit-identical.
__global__ __launch_bounds__(256) void rows_axpby_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ a,
                                                          const float* __restrict__ c, const float* __restrict__ d, float* __restrict__ out,
                                                          long per, long total, int do_clamp, float lo, float hi) {
#pragma clang fp contract(off)
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long b = i / per;
    float v = a ? a[b] * x[i] : x[i];
    if (y) {
      const float w = c ? c[b] * y[i] : y[i];
      v = v + w;
    }
    if (d) v = v / d[b];
    if (do_clamp) v = fminf(fmaxf(v, lo), hi);
    out[i] = v;
  }
}

// scripts/helpers/sample_dataset.py:44-53: clip(-1,1) -> (x+1)/2*255 -> CHW->HWC -> astype(uint8) (truncation)
__global__ void image_u8_dataset_kernel(const float* __restrict__ x, uint8_t* __restrict__ out, int N, int C, int H, int W) {
#pragma clang fp contract(off)
  const long total = (long)N * C * H * W;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // index into NHWC output
  if (i >= total) return;
  const int c = (int)(i % C);
  long t = i / C;
  const int w = (int)(t % W); t /= W;
  const int h = (int)(t % H);
  const int n = (int)(t / H);
  float v = x[(((long)n * C + c) * H + h) * W + w];
  v = fminf(fmaxf(v, -1.0f), 1.0f);
  v = v + 1.0f;
  v = v / 2.0f;
  v = v * 255.0f;
  out[i] = (uint8_t)v;
}

// per-image min / max of clamp((x+1)/2, 0, 1)  (scripts/sample.py:49-51 + torchvision save_image(normalize=True, scale_each=True))
__global__ __launch_bounds__(256) void image_minmax_kernel(const float* __restrict__ x, float* __restrict__ mm, long per) {
#pragma clang fp contract(off)
  __shared__ float smin[256], smax[256];
  const int n = blockIdx.x;
  float lo = INFINITY, hi = -INFINITY;
  for (long i = threadIdx.x; i < per; i += 256) {
    float v = x[(long)n * per + i];
    v = v + 1.0f;
    v = v / 2.0f;
    v = fminf(fmaxf(v, 0.0f), 1.0f);
    lo = fminf(lo, v);
    hi = fmaxf(hi, v);
  }
  smin[threadIdx.x] = lo; smax[threadIdx.x] = hi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + s]);
      smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + s]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { mm[2 * n] = smin[0]; mm[2 * n + 1] = smax[0]; }
}

__global__ void image_u8_normalized_kernel(const float* __restrict__ x, const float* __restrict__ mm, uint8_t* __restrict__ out, int N, int C, int H,
                                           int W) {
#pragma clang fp contract(off)
  const long total = (long)N * C * H * W;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  long t = i / C;
  const int w = (int)(t % W); t /= W;
  const int h = (int)(t % H);
  const int n = (int)(t / H);
  float v = x[(((long)n * C + c) * H + h) * W + w];
  v = v + 1.0f;
  v = v / 2.0f;
  v = fminf(fmaxf(v, 0.0f), 1.0f);
  const float lo = mm[2 * n], hi = mm[2 * n + 1];
  v = fminf(fmaxf(v, lo), hi);                 // torchvision norm_ip: clamp_(min, max)
  v = v - lo;
  v = v / fmaxf(hi - lo, 1e-5f);               // .sub_(low).div_(max(high - low, 1e-5))
  v = v * 255.0f;                              // save_image: mul(255).add_(0.5).clamp_(0, 255).to(uint8)
  v = v + 0.5f;
  v = fminf(fmaxf(v, 0.0f), 255.0f);
  out[i] = (uint8_t)v;
}

}  // namespace

extern "C" {

int mf_rows_axpby_f32(const float* x, const float* y, const float* a, const float* c, const float* d, float* out, int B, int64_t per_row,
                      int do_clamp, float lo, float hi, void* stream) {
  MF_REQUIRE(x && out && B > 0 && per_row > 0, MF_EINVAL, "rows_axpby: bad args");
  MF_REQUIRE(y || !c, MF_EINVAL, "rows_axpby: c without y");
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)B * per_row;
  ProfScope ps(MF_FAM_SCHED, s, 3.0 * total, 12.0 * total);
  long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  MF_LAUNCH(rows_axpby_kernel, dim3((int)blocks), dim3(256), 0, s, x, y, a, c, d, out, (long)per_row, total, do_clamp, lo, hi);
  return check_launch("rows_axpby");
}

int mf_image_egress_u8(const float* x_nchw, uint8_t* out_nhwc, float* minmax_ws, int N, int C, int H, int W, int mode, void* stream) {
  MF_REQUIRE(x_nchw && out_nhwc && N > 0 && C > 0 && H > 0 && W > 0, MF_EINVAL, "image_egress: bad args");
  MF_REQUIRE(mode == 0 || mode == 1, MF_EINVAL, "image_egress: mode");
  MF_REQUIRE(mode == 0 || minmax_ws, MF_EWORKSPACE, "image_egress: mode 1 needs a 2*N float workspace");
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)N * C * H * W;
  ProfScope ps(MF_FAM_MISC, s, 4.0 * total, 5.0 * total);
  const int blocks = (int)((total + 255) / 256);
  if (mode == 0) {
    MF_LAUNCH(image_u8_dataset_kernel, dim3(blocks), dim3(256), 0, s, x_nchw, out_nhwc, N, C, H, W);
    return check_launch("image_egress");
  }
  MF_LAUNCH(image_minmax_kernel, dim3(N), dim3(256), 0, s, x_nchw, minmax_ws, (long)C * H * W);
  int rc = check_launch("image_minmax");
  if (rc) return rc;
  MF_LAUNCH(image_u8_normalized_kernel, dim3(blocks), dim3(256), 0, s, x_nchw, minmax_ws, out_nhwc, N, C, H, W);
  return check_launch("image_egress_normalized");
}

}  // extern "C"
