// edge_ops.hip -- per-row scaled combinations (the scheduler's tensor API with a timestep PER ROW) and image egress.
// Streaming passes over small tensors; bit-exact against ATen / numpy elementwise chains (library built with -ffp-contract=off).
#include "common.h"
#include "split_f16.h"

using namespace mf;

namespace {

// The network input as an operand of the fp16-pair convolution (round 4): x NCHW [N][C][HW] (C <= CP) -> the fp16-pair form of the NHWC tensor
// [N][HW][CP], channels C .. CP-1 zero, scaled per sample by its own max |x| -- measured and applied in ONE launch (one workgroup per sample:
// the latents of the sampling path hold 8 x 32 x 32 .. 8 x 64 x 64 values per sample), which also publishes bound_out[n].  With the weights
// zero-padded to CP input channels the 8 -> 256 input convolution of the UNet runs on the matrix cores like every other one.
__global__ __launch_bounds__(256) void pack_nchw_pairs_kernel(const float* __restrict__ x, sf_u32x4* __restrict__ out, float* __restrict__ bound_out, int C, int HW,
                                                              int CP) {
  __shared__ float sm[4];
  const int n = blockIdx.x, tid = threadIdx.x;
  const float* xn = x + (long)n * C * HW;
  float m = 0.f;
  for (int i = tid; i < C * HW; i += 256) m = fmaxf(m, fabsf(xn[i]));
  m = wave_max(m);
  if ((tid & 63) == 0) sm[tid >> 6] = m;
  __syncthreads();
  const float b = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
  if (tid == 0) bound_out[n] = b;
  const float sc = exp2i(-scale_exp_of(b));
  const int octs = CP >> 3;
  for (int i = tid; i < HW * octs; i += 256) {   // (octet-major: consecutive threads take consecutive pixels of one octet -> coalesced NCHW reads)
    const int o = i / HW, px = i - o * HW;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = o * 8 + k;
      v[k] = c < C ? xn[(long)c * HW + px] * sc : 0.f;
    }
    sf_u32x4 hi, lo;
    split8_f16(sf_f32x4{v[0], v[1], v[2], v[3]}, sf_f32x4{v[4], v[5], v[6], v[7]}, hi, lo);
    sf_u32x4* q = out + (((long)n * HW + px) * octs + o) * 2;
    q[0] = hi;
    q[1] = lo;
  }
}

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

// nn.AvgPool2d(k, stride, pad) with count_include_pad (torch's default), NHWC: the window is clipped to the PADDED extent for the divisor and
// to the image for the sum, summed row by row like ATen's scalar loop, then ONE division (conv_blocks.py:57-63, learnable_interpolation=False)
__global__ __launch_bounds__(256) void avgpool_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C4, int Ho, int Wo,
                                                            int k, int stride, int pad) {
#pragma clang fp contract(off)
  const long total = (long)N * Ho * Wo * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    long r = i / C4;
    const int ox = (int)(r % Wo);
    r /= Wo;
    const int oy = (int)(r % Ho), n = (int)(r / Ho);
    int hs = oy * stride - pad, ws = ox * stride - pad;
    int he = min(hs + k, H + pad), we = min(ws + k, W + pad);
    const float pool = (float)((he - hs) * (we - ws));
    hs = max(hs, 0); ws = max(ws, 0); he = min(he, H); we = min(we, W);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int iy = hs; iy < he; ++iy)
      for (int ix = ws; ix < we; ++ix) {
        const float4 v = *reinterpret_cast<const float4*>(x + (((long)n * H + iy) * W + ix) * (C4 * 4L) + c4 * 4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
    *reinterpret_cast<float4*>(y + i * 4) = make_float4(s.x / pool, s.y / pool, s.z / pool, s.w / pool);
  }
}

// F.interpolate(x, scale 2, mode="nearest-exact") on NHWC: out[oy][ox] = in[oy / 2][ox / 2] (conv_blocks.py:128-130)
__global__ __launch_bounds__(256) void upsample_nearest2_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C4) {
  const long total = (long)N * (2 * H) * (2 * W) * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    long r = i / C4;
    const int ox = (int)(r % (2 * W));
    r /= 2 * W;
    const int oy = (int)(r % (2 * H)), n = (int)(r / (2 * H));
    *reinterpret_cast<float4*>(y + i * 4) = *reinterpret_cast<const float4*>(x + (((long)n * H + (oy >> 1)) * W + (ox >> 1)) * (C4 * 4L) + c4 * 4);
  }
}

// y[n][h][w][c * 4 + dy * 2 + dx] += x[n][2 h + dy][2 w + dx][c]   -- nn.PixelUnshuffle(2) added to the stride-2 convolution's output
// (conv_blocks.py:54-55,68-69: BasicDown(use_res=True), out_channels == 4 in_channels).  One thread per output float4 = one input
// channel's 2 x 2 neighbourhood.
__global__ __launch_bounds__(256) void pixel_unshuffle2_add_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int Ho, int Wo, int C) {
  const long total = (long)N * Ho * Wo * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long r = i / C;
    const int w = (int)(r % Wo);
    r /= Wo;
    const int h = (int)(r % Ho), n = (int)(r / Ho);
    const float* s = x + (((long)n * 2 * Ho + 2 * h) * (2 * Wo) + 2 * w) * C + c;
    float4 v = *reinterpret_cast<float4*>(y + i * 4);
    v.x += s[0]; v.y += s[C]; v.z += s[(long)2 * Wo * C]; v.w += s[(long)2 * Wo * C + C];
    *reinterpret_cast<float4*>(y + i * 4) = v;
  }
}

// y[n][2 h + dy][2 w + dx][c] += x[n][h][w][c * 4 + dy * 2 + dx]   -- nn.PixelShuffle(2) added to the up-convolution's output
// (conv_blocks.py:114-115,125-126: BasicUp(use_res=True), out_channels == in_channels / 4).  One thread per input float4.
__global__ __launch_bounds__(256) void pixel_shuffle2_add_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int Co) {
  const long total = (long)N * H * W * Co;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Co);
    long r = i / Co;
    const int w = (int)(r % W);
    r /= W;
    const int h = (int)(r % H), n = (int)(r / H);
    const float4 v = *reinterpret_cast<const float4*>(x + i * 4);
    float* d = y + (((long)n * 2 * H + 2 * h) * (2 * W) + 2 * w) * Co + c;
    d[0] += v.x; d[Co] += v.y; d[(long)2 * W * Co] += v.z; d[(long)2 * W * Co + Co] += v.w;
  }
}

extern "C" {

int mf_pack_nchw_pairs_f32(const float* x, void* out_pairs, float* bound_out, int N, int C, int HW, int CP, void* stream) {
  MF_REQUIRE(x && out_pairs && bound_out && N > 0 && C > 0 && HW > 0 && CP >= C && CP % 32 == 0, MF_EINVAL, "pack_nchw_pairs: bad args (C <= CP, CP %% 32 == 0)");
  MF_REQUIRE((long)C * HW <= (1L << 18), MF_EUNSUPPORTED, "pack_nchw_pairs: one workgroup per sample -- at most 2^18 values per sample");
  ProfScope ps(MF_FAM_MISC, (hipStream_t)stream, 0, 4.0 * N * (double)HW * (C + CP));
  MF_LAUNCH(pack_nchw_pairs_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, x, reinterpret_cast<sf_u32x4*>(out_pairs), bound_out, C, HW, CP);
  return check_launch("pack_nchw_pairs");
}

int mf_pixel_unshuffle2_add_nhwc_f32(const float* x, float* y, int N, int H, int W, int C, void* stream) {
  MF_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0, MF_EINVAL, "pixel_unshuffle2_add: bad args (H, W even)");
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)N * (H / 2) * (W / 2) * C;
  ProfScope ps(MF_FAM_MISC, s, 4.0 * total, 4.0 * 12.0 * total);
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  MF_LAUNCH(pixel_unshuffle2_add_kernel, dim3((int)blocks), dim3(256), 0, s, x, y, N, H / 2, W / 2, C);
  return check_launch("pixel_unshuffle2_add");
}

int mf_pixel_shuffle2_add_nhwc_f32(const float* x, float* y, int N, int H, int W, int C, void* stream) {
  MF_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, MF_EINVAL, "pixel_shuffle2_add: bad args (C %% 4 == 0)");
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)N * H * W * (C / 4);
  ProfScope ps(MF_FAM_MISC, s, 4.0 * total, 4.0 * 12.0 * total);
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  MF_LAUNCH(pixel_shuffle2_add_kernel, dim3((int)blocks), dim3(256), 0, s, x, y, N, H, W, C / 4);
  return check_launch("pixel_shuffle2_add");
}

int mf_avgpool2d_nhwc_f32(const float* x, float* y, int N, int H, int W, int C, int k, int stride, int pad, void* stream) {
  MF_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, MF_EINVAL, "avgpool2d: bad args (C %% 4 == 0)");
  MF_REQUIRE(k >= 1 && stride >= 1 && pad >= 0 && 2 * pad <= k, MF_EINVAL, "avgpool2d: kernel %d stride %d pad %d", k, stride, pad);
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  MF_REQUIRE(Ho > 0 && Wo > 0, MF_EINVAL, "avgpool2d: empty output");
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)N * Ho * Wo * (C / 4);
  ProfScope ps(MF_FAM_MISC, s, (double)k * k * total * 4, 4.0 * ((double)N * H * W * C + (double)total * 4));
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  MF_LAUNCH(avgpool_nhwc_kernel, dim3((int)blocks), dim3(256), 0, s, x, y, N, H, W, C / 4, Ho, Wo, k, stride, pad);
  return check_launch("avgpool2d");
}

int mf_upsample_nearest2x_nhwc_f32(const float* x, float* y, int N, int H, int W, int C, void* stream) {
  MF_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, MF_EINVAL, "upsample_nearest2x: bad args (C %% 4 == 0)");
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)N * 4 * H * W * (C / 4);
  ProfScope ps(MF_FAM_MISC, s, 0, 4.0 * 5 * (double)N * H * W * C);
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  MF_LAUNCH(upsample_nearest2_nhwc_kernel, dim3((int)blocks), dim3(256), 0, s, x, y, N, H, W, C / 4);
  return check_launch("upsample_nearest2x");
}

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
