// small_ops.hip -- embedding MLPs, sinusoidal features, label lookup, layout edges, elementwise helpers.
// All are launch-latency / HBM-bound; none reaches 0.1 % of the path's FLOPs.
#include "common.h"

using namespace mf;

namespace {

constexpr int kLinRows = 8;   // batch rows handled per pass
constexpr int kLinOutPerWave = 1;   // outputs per wave per staged tile: 1 measured best (the op is parallelism-bound: 16 was 3.6x slower)

// y[b][o] = sum_i f(x[b][i]) w[o][i] + bias[o]; one wave per output feature, f(x) staged in LDS,
// lanes stride the In dimension with float4 loads (coalesced weight rows), wave-shuffle reduction.
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x, long x_stride, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ y, long y_stride, int B, int In, int Out,
                                                      int act_in, int act_out, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [kLinRows][In]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int obase = (blockIdx.x * 4 + wave) * kLinOutPerWave;
  for (int b0 = 0; b0 < B; b0 += kLinRows) {
    const int nb = min(kLinRows, B - b0);
    __syncthreads();
    for (int i = tid; i < nb * In; i += 256) {
      const int b = i / In, k = i - b * In;
      float v = x[(long)(b0 + b) * x_stride + k];
      xs[b * In + k] = act_in ? swish_acc(v) : v;
    }
    __syncthreads();
    for (int oo = 0; oo < kLinOutPerWave; ++oo) {
      const int o = obase + oo;
      if (o >= Out) break;
      float acc[kLinRows];
#pragma unroll
      for (int b = 0; b < kLinRows; ++b) acc[b] = 0.f;
      const float* wr = w + (long)o * In;
      if ((In & 3) == 0) {
        for (int k = lane * 4; k < In; k += 256) {
          const float4 wv = *reinterpret_cast<const float4*>(wr + k);
#pragma unroll
          for (int b = 0; b < kLinRows; ++b) {
            if (b < nb) {
              const float4 xv = *reinterpret_cast<const float4*>(xs + b * In + k);
              acc[b] = fmaf(xv.x, wv.x, acc[b]); acc[b] = fmaf(xv.y, wv.y, acc[b]);
              acc[b] = fmaf(xv.z, wv.z, acc[b]); acc[b] = fmaf(xv.w, wv.w, acc[b]);
            }
          }
        }
      } else {
        for (int k = lane; k < In; k += 64) {
          const float wv = wr[k];
#pragma unroll
          for (int b = 0; b < kLinRows; ++b)
            if (b < nb) acc[b] = fmaf(xs[b * In + k], wv, acc[b]);
        }
      }
#pragma unroll
      for (int b = 0; b < kLinRows; ++b) {
        float v = acc[b];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0 && b < nb) {
          v += bias ? bias[o] : 0.f;
          if (act_out) v = swish_acc(v);
          float* dst = y + (long)(b0 + b) * y_stride + o;
          *dst = accumulate ? *dst + v : v;
        }
      }
    }
  }
}

__global__ void sinusoidal_kernel(const float* __restrict__ t, const float* __restrict__ freqs, float* __restrict__ out, int B, int dim, float coef,
                                  int flip) {
#pragma clang fp contract(off)
  const int half = dim >> 1;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * dim) return;
  const int b = i / dim, j = i - b * dim;
  float v = 0.f;
  if (j < 2 * half) {
    int jj = j;
    if (flip) jj = j < half ? j + half : j - half;
    const int k = jj < half ? jj : jj - half;
    // reference: exp(-emb * arange(half)) in fp32, then t * freq, then sin | cos (time_embedder.py:18-21)
    // a 1-ulp difference in freq is amplified by t (up to 999): the host passes the table it computed exactly
    // like the reference (torch.exp on CPU); device expf is the fallback
    const float freq = freqs ? freqs[k] : expf(-coef * (float)k);
    const float a = t[b] * freq;
    v = jj < half ? sinf(a) : cosf(a);
  }
  out[i] = v;
}

// LearnedSinusoidalPosEmb (time_embedder.py:31-49): out[b] = [t_b | sin(a_bk) | cos(a_bk) | 0 if emb_dim is odd], a_bk = ((t_b w_k) 2) pi in the
// reference's left-to-right fp32 order (`x * w * 2 * math.pi`), k < half = emb_dim / 2; row length 1 + 2 half + (emb_dim & 1)
__global__ void learned_sinusoidal_kernel(const float* __restrict__ t, const float* __restrict__ w, float* __restrict__ out, int B, int half, int row) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * row) return;
  const int b = i / row, j = i - b * row;
  float v = 0.f;
  if (j == 0) v = t[b];
  else if (j <= 2 * half) {
    const int k = (j - 1) < half ? j - 1 : j - 1 - half;
    const float a = ((t[b] * w[k]) * 2.0f) * 3.14159265358979323846f;
    v = (j - 1) < half ? sinf(a) : cosf(a);
  }
  out[i] = v;
}

__global__ void embedding_add_kernel(const float* __restrict__ table, const int64_t* __restrict__ idx, float* __restrict__ io, int B, int D,
                                     int rows) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D) return;
  const int b = i / D, d = i - b * D;
  long r = idx[b];
  if (r < 0 || r >= rows) return;  // torch would raise; leave the row untouched (host validates)
  io[i] += table[r * D + d];
}

// tiled transpose of the [C][HW] <-> [HW][C] planes of each sample
__global__ void transpose_planes_kernel(const float* __restrict__ x, float* __restrict__ y, int R, int Cc) {
  // x: [N][R][Cc] -> y: [N][Cc][R]
  __shared__ float tile[32][33];
  const long nbase = (long)blockIdx.z * R * Cc;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int r = r0 + i, c = c0 + threadIdx.x;
    if (r < R && c < Cc) tile[i][threadIdx.x] = x[nbase + (long)r * Cc + c];
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int c = c0 + i, r = r0 + threadIdx.x;
    if (r < R && c < Cc) y[nbase + (long)c * R + r] = tile[threadIdx.x][i];
  }
}

__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, long n) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) o[i] = a[i] + b[i];
}

__global__ void diag_gaussian_kernel(const float* __restrict__ mom, const float* __restrict__ noise, float* __restrict__ z, int N, int C, int HW) {
#pragma clang fp contract(off)
  const long total = (long)N * C * HW;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long per = (long)C * HW;
  const long n = i / per, r = i - n * per;
  const float mean = mom[n * 2 * per + r];
  float logvar = mom[n * 2 * per + per + r];
  logvar = fminf(fmaxf(logvar, -30.0f), 20.0f);
  const float hl = 0.5f * logvar;
  const float sd = expf(hl);
  const float sn = sd * noise[i];
  z[i] = mean + sn;
}

// KL term of DiagonalGaussianDistribution (latent_embedders.py:29-31): 0.5 * sum(mean^2 + var - 1 - logvar) / N over the whole tensor, one
// workgroup, fp64 accumulation (the evaluation-time VAE.forward: off the sampling path)
__global__ __launch_bounds__(1024) void diag_gaussian_kl_kernel(const float* __restrict__ mom, float* __restrict__ out, int N, int C, int HW) {
  __shared__ double red[16];
  const long per = (long)C * HW, total = (long)N * per;
  double acc = 0.0;
  for (long i = threadIdx.x; i < total; i += 1024) {
    const long n = i / per, r = i - n * per;
    const float mean = mom[n * 2 * per + r];
    const float logvar = fminf(fmaxf(mom[n * 2 * per + per + r], -30.0f), 20.0f);
    acc += (double)(mean * mean) + (double)expf(logvar) - 1.0 - (double)logvar;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < 16; ++w) s += red[w];
    out[0] = (float)(0.5 * s / (double)N);
  }
}

__global__ void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ out,
                                 long rows, int C, float eps) {
  // one wave per row
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += xr[c];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  const float mean = s / (float)C;
  float q = 0.f;
  for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; q = fmaf(d, d, q); }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
  const float rstd = 1.0f / sqrtf(q / (float)C + eps);
  for (int c = lane; c < C; c += 64) out[row * C + c] = (xr[c] - mean) * rstd * gamma[c] + beta[c];
}

__global__ void geglu_kernel(const float* __restrict__ h, float* __restrict__ out, long rows, int C) {
  const long total = rows * C;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long r = i / C;
    const int c = (int)(i - r * C);
    const float a = h[r * 2 * C + c], g = h[r * 2 * C + C + c];
    const float gelu = 0.5f * g * (1.0f + erff(g * 0.70710678118654752440f));  // F.gelu default (erf form)
    out[i] = a * gelu;
  }
}

}  // namespace

extern "C" {

int mf_linear_f32(const float* x, int64_t x_stride, const float* w, const float* bias, float* y, int64_t y_stride, int B, int In, int Out,
                  int act_in, int act_out, int accumulate, void* stream) {
  MF_REQUIRE(x && w && y && B > 0 && In > 0 && Out > 0, MF_EINVAL, "linear: bad args");
  const size_t lds = (size_t)kLinRows * In * sizeof(float);
  MF_REQUIRE(lds <= 64 * 1024, MF_EUNSUPPORTED, "linear: In=%d too large", In);
  MF_REQUIRE((In & 3) != 0 || (((uintptr_t)w & 15) == 0), MF_EINVAL, "linear: weight must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(MF_FAM_LINEAR, s, 2.0 * B * In * (double)Out, 4.0 * ((double)In * Out + (double)B * (In + Out)));
  MF_LAUNCH(linear_kernel, dim3((Out + 4 * kLinOutPerWave - 1) / (4 * kLinOutPerWave)), dim3(256), lds, s, x, (long)x_stride, w, bias, y, (long)y_stride, B, In, Out, act_in,
                     act_out, accumulate);
  return check_launch("linear");
}

int mf_sinusoidal_f32(const float* t, const float* freqs, float* out, int B, int dim, float max_period, float shift, int flip, void* stream) {
  MF_REQUIRE(t && out && B > 0 && dim > 1, MF_EINVAL, "sinusoidal: bad args");
  const int half = dim / 2;
  MF_REQUIRE((float)half - shift > 0.f, MF_EINVAL, "sinusoidal: half_dim - shift must be > 0");
  // math.log(max_period) / (half_dim - shift) evaluated in double like the Python reference, then cast
  const float coef = (float)(log((double)max_period) / ((double)half - (double)shift));
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(MF_FAM_MISC, s, 0, 4.0 * B * dim);
  MF_LAUNCH(sinusoidal_kernel, dim3((B * dim + 255) / 256), dim3(256), 0, s, t, freqs, out, B, dim, coef, flip);
  return check_launch("sinusoidal");
}

int mf_learned_sinusoidal_f32(const float* t, const float* weights, float* out, int B, int emb_dim, void* stream) {
  MF_REQUIRE(t && weights && out && B > 0 && emb_dim > 1, MF_EINVAL, "learned_sinusoidal: bad args");
  const int half = emb_dim / 2, row = 1 + 2 * half + (emb_dim & 1);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(MF_FAM_MISC, s, 0, 4.0 * B * row);
  MF_LAUNCH(learned_sinusoidal_kernel, dim3((B * row + 255) / 256), dim3(256), 0, s, t, weights, out, B, half, row);
  return check_launch("learned_sinusoidal");
}

int mf_embedding_add_f32(const float* table, const int64_t* idx, float* io, int B, int D, int num_rows, void* stream) {
  MF_REQUIRE(table && idx && io && B > 0 && D > 0 && num_rows > 0, MF_EINVAL, "embedding_add: bad args");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(MF_FAM_MISC, s, 0, 12.0 * B * D);
  MF_LAUNCH(embedding_add_kernel, dim3((B * D + 255) / 256), dim3(256), 0, s, table, idx, io, B, D, num_rows);
  return check_launch("embedding_add");
}

static int transpose_planes(const float* x, float* y, int N, int R, int Cc, hipStream_t s, const char* what) {
  MF_REQUIRE(x && y && N > 0 && R > 0 && Cc > 0, MF_EINVAL, "%s: bad args", what);
  MF_REQUIRE(N <= 65535, MF_EUNSUPPORTED, "%s: N too large", what);
  ProfScope ps(MF_FAM_MISC, s, 0, 8.0 * N * (double)R * Cc);
  MF_LAUNCH(transpose_planes_kernel, dim3((Cc + 31) / 32, (R + 31) / 32, N), dim3(32, 8), 0, s, x, y, R, Cc);
  return check_launch(what);
}

int mf_nchw_to_nhwc_f32(const float* x, float* y, int N, int C, int H, int W, void* stream) {
  return transpose_planes(x, y, N, C, H * W, (hipStream_t)stream, "nchw_to_nhwc");
}
int mf_nhwc_to_nchw_f32(const float* x, float* y, int N, int C, int H, int W, void* stream) {
  return transpose_planes(x, y, N, H * W, C, (hipStream_t)stream, "nhwc_to_nchw");
}

int mf_add_f32(const float* a, const float* b, float* out, int64_t n, void* stream) {
  MF_REQUIRE(a && b && out && n > 0, MF_EINVAL, "add: bad args");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(MF_FAM_MISC, s, (double)n, 12.0 * n);
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  MF_LAUNCH(add_kernel, dim3((int)blocks), dim3(256), 0, s, a, b, out, (long)n);
  return check_launch("add");
}

int mf_diag_gaussian_sample_f32(const float* moments, const float* noise, float* z, int N, int C, int HW, void* stream) {
  MF_REQUIRE(moments && noise && z && N > 0 && C > 0 && HW > 0, MF_EINVAL, "diag_gaussian: bad args");
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)N * C * HW;
  ProfScope ps(MF_FAM_MISC, s, 4.0 * total, 16.0 * total);
  MF_LAUNCH(diag_gaussian_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, s, moments, noise, z, N, C, HW);
  return check_launch("diag_gaussian");
}

int mf_diag_gaussian_kl_f32(const float* moments, float* kl, int N, int C, int HW, void* stream) {
  MF_REQUIRE(moments && kl && N > 0 && C > 0 && HW > 0, MF_EINVAL, "diag_gaussian_kl: bad args");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(MF_FAM_MISC, s, 6.0 * N * C * HW, 8.0 * N * C * HW);
  MF_LAUNCH(diag_gaussian_kl_kernel, dim3(1), dim3(1024), 0, s, moments, kl, N, C, HW);
  return check_launch("diag_gaussian_kl");
}

int mf_layernorm_f32(const float* x, const float* gamma, const float* beta, float* out, int64_t rows, int C, float eps, void* stream) {
  MF_REQUIRE(x && gamma && beta && out && rows > 0 && C > 0, MF_EINVAL, "layernorm: bad args");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(MF_FAM_MISC, s, 8.0 * rows * C, 8.0 * rows * C);
  MF_LAUNCH(layernorm_kernel, dim3((int)((rows + 3) / 4)), dim3(256), 0, s, x, gamma, beta, out, (long)rows, C, eps);
  return check_launch("layernorm");
}

int mf_geglu_f32(const float* h, float* out, int64_t rows, int C, void* stream) {
  MF_REQUIRE(h && out && rows > 0 && C > 0, MF_EINVAL, "geglu: bad args");
  hipStream_t s = (hipStream_t)stream;
  const long total = rows * C;
  ProfScope ps(MF_FAM_MISC, s, 10.0 * total, 12.0 * total);
  long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  MF_LAUNCH(geglu_kernel, dim3((int)blocks), dim3(256), 0, s, h, out, (long)rows, C);
  return check_launch("geglu");
}

}  // extern "C"
