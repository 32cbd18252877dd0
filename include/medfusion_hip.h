/*
 * medfusion_hip.h -- C-ABI of libmedfusion_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the latent-diffusion SAMPLING path of mueller-franzes/medfusion
 * (DiffusionPipeline.sample -> UNet denoise loop -> VAE.decode; VAE.encode as API surface).
 * The reference is pure PyTorch: its "FFI" for this path is the set of ATen ops each Python
 * method dispatches.  Every entry point below names the reference call site(s) it replaces
 * (paths relative to the reference repo root).  The Python host (medfusion_amd/) binds these
 * with ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *  - extern "C", plain pointers + sizes, no torch types.  All pointers are DEVICE pointers
 *    owned by the caller (torch allocations); the library never allocates, frees or retains
 *    device memory.  Scratch is caller-provided (`*_workspace_bytes`).
 *  - fp32 everywhere.  Activations are NHWC ("channels-last") inside the path; the API edge
 *    tensors of the reference (latents, images) are NCHW and are read/written directly by the
 *    edge convolutions (MF_LAYOUT_NCHW) -- no separate transpose pass.
 *  - Every call is an asynchronous launch on `stream` (a hipStream_t passed as void*);
 *    no host sync, no allocation, no host reads: hipGraph-capture-safe.
 *  - Return 0 on success, negative MF_E* on failure; message via mf_last_error() (thread-local).
 */
#ifndef MEDFUSION_HIP_H
#define MEDFUSION_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MF_VERSION 250 /* 0.2.5 */

enum { MF_OK = 0, MF_EINVAL = -1, MF_EUNSUPPORTED = -2, MF_ELAUNCH = -3, MF_EWORKSPACE = -4 };
enum { MF_LAYOUT_NHWC = 0, MF_LAYOUT_NCHW = 1 };

int mf_version(void);
const char* mf_last_error(void);

/* ------------------------------------------------------------------ convolution
 * Replaces torch.nn.Conv2d.forward at: conv_blocks.py:185 (BasicBlock.conv), :238 (conv_res 1x1),
 * :66 (BasicDown 3x3 stride 2), :123-125 (BasicUp: F.interpolate nearest-exact x2 + 3x3 conv, fused:
 * upsample=1 gathers from the low-res tensor), unet2.py:259 (torch.cat([h, skip]) fused: x2/C2),
 * unet2.py:267 / latent_embedders.py:768 (1x1 out convs).
 * Weights are pre-packed once by mf_pack_conv_weight_f32 (OIHW -> [Cout][KH][KW][Cin]).
 */
typedef struct MfConvDesc {
  int32_t N, Hin, Win;     /* batch, input spatial size (BEFORE the fused x2 upsample) */
  int32_t C1, C2;          /* channels read from x1 and x2 (C2 = 0: single source); Cin = C1 + C2 */
  int32_t Cout;
  int32_t KH, KW;          /* 1x1 or 3x3 */
  int32_t stride;          /* 1 or 2 */
  int32_t pad;             /* MONAI get_padding(k, s) = int((k - s + 1) / 2) */
  int32_t upsample;        /* 1: nearest x2 of the input fused into the gather; 2: the same op in its SUB-PIXEL form (4 phase-specific
                              2x2 convs on the low-res tensor, 4/9 of the MACs; weights from mf_pack_upconv_weight_f32) */
  int32_t in_layout;       /* layout of x1 (NCHW only with C2 == 0) */
  int32_t out_layout;      /* layout of y */
  int32_t tile_hint;       /* 0 = auto; else forces an implicit-GEMM tile config (tuning/tests) */
  int32_t splitk_hint;     /* 0 = auto; else forces split-K factor */
  int32_t precision;       /* arithmetic of the implicit-GEMM path (MF_CONV_*, below); the small/direct kernels are always plain fp32 */
} MfConvDesc;
/* MF_CONV_FP32: v_mfma_f32_32x32x2_f32, products and sums exactly those of an fp32 fma chain.
 * MF_CONV_FP32_SPLIT3_W3: every fp32 operand is split EXACTLY into three bf16 terms (x = h + m + l, 8 significant bits each) and
 *   a*b is accumulated in fp32 on the bf16 matrix cores as the six terms of order <= 2 (hh, hm, mh, mm, hl, lh); the dropped terms
 *   are < 2^-23 |a*b|.  16x the MFMA rate / 6 terms = 3/8 of the matrix time.  `w_packed` points to the weights ALREADY split:
 *   mf_split_conv_weight_bf16x3 turns either fp32 packing ([rows][K]) into [rows][K/8][3 pieces][8] bf16 (6 bytes per weight) once
 *   at load time; the activations are split in the kernel.  Measured against fp64 its error is at or below the fp32-MFMA kernel's on
 *   every shape of the path (the planner keeps one accumulation chain <= 96 chunks via split-K).  Implicit-GEMM path only.
 *   (Values 1 and 2 -- the same arithmetic with the weights split in the kernel, and its chunk-sum variant -- were retired in ABI 200.)
 * MF_CONV_BF16 (opt-in, REDUCED precision): operands rounded to bf16 (round to nearest even), one MFMA term, fp32
 *   accumulate; `w_packed` points to weights converted by mf_convert_conv_weight_bf16 ([rows][K] bf16).  Error vs fp64 ~3e-3 per
 *   convolution (2^-9 per operand); never selected by default, has its own tolerance in the tests.  Implicit-GEMM path only.
 * MF_CONV_FP32_F16X2: fp32 through PAIRS of fp16 on the fp16 matrix cores.  Every operand -- activations and weights -- lives in HBM
 *   as x ~ hi + lo/2048 with hi = RN16(x), lo = RN16((x - hi) * 2048): 23 of the 24 significand bits (error <= 2^-23 |x|, one fp32
 *   ulp at most, zero for 3 values of 4; the format is 4 bytes per element like fp32, groups of 8 channels as [hi x 8][lo x 8]).  a*b is accumulated in
 *   fp32 as wh*xh + (wh*xl + wl*xh)/2048 -- 3 matrix instructions per product instead of 6, dropped term < 2^-22 |a*b|.  Because the
 *   operands need no arithmetic in the kernel, both go HBM -> LDS by LDS-DMA.  Entry point mf_conv2d_f16x2 (operands produced by
 *   mf_split_f16x2 or by the split output of mf_gn_apply_split_f32); every operand tensor carries a per-sample power-of-two scale, so
 *   there is no range limit (mf_gn_apply_split_f32 below has the details).
 * MF_CONV_F16 (opt-in, REDUCED precision; SURVEY 8f row 4, ABI 210): the fp16-pair operands and the LDS-DMA kernel of MF_CONV_FP32_F16X2 with
 *   ONE product term -- main += wh * xh only, i.e. both operands rounded to fp16 (11 significant bits, the per-sample power-of-two
 *   scales keep the range), fp32 accumulate.  Same entry point (mf_conv2d_f16x2 with desc.precision = MF_CONV_F16), same operand
 *   images; a third of the matrix instructions, half of the LDS fragment reads.  Error vs fp64 ~3e-4 per convolution (2^-12 per
 *   operand); never selected by default, own tolerance in the tests, never the headline. */
enum { MF_CONV_FP32 = 0, MF_CONV_FP32_SPLIT3_W3 = 3, MF_CONV_BF16 = 4, MF_CONV_FP32_F16X2 = 5, MF_CONV_F16 = 6 };

int mf_pack_conv_weight_f32(const float* w_oihw, float* w_packed, int Cout, int Cin, int KH, int KW, void* stream);
/* nearest-x2 + 3x3 (conv_blocks.py:123-125) as the transposed-conv-equivalent sub-pixel form: OIHW 3x3 -> [4][Cout][2][2][Cin],
 * each phase kernel = sum of the 3x3 taps that land on the same source pixel.  mf_conv2d_subpixel_ok: can `d` (upsample = 2) run so? */
int mf_pack_upconv_weight_f32(const float* w_oihw, float* w_packed, int Cout, int Cin, void* stream);
int mf_conv2d_subpixel_ok(const MfConvDesc* d);
/* 1 if `d` runs on the implicit-GEMM kernel (the only one that looks at `precision`), 0 for the small-Cin / direct kernels */
int mf_conv2d_is_igemm(const MfConvDesc* d);
/* rows = Cout (x4 for the sub-pixel packing); out: rows * K * 6 bytes */
int mf_split_conv_weight_bf16x3(const float* w_packed, void* w_split, long rows, int K, void* stream);
/* out: rows * K * 2 bytes (MF_CONV_BF16) */
int mf_convert_conv_weight_bf16(const float* w_packed, void* w_bf16, long rows, int K, void* stream);
size_t mf_conv2d_workspace_bytes(const MfConvDesc* d);
/* y = conv(x1 (++ x2 on channels), w) + bias.  bias may be NULL.  workspace >= mf_conv2d_workspace_bytes. */
int mf_conv2d_f32(const float* x1, const float* x2, const float* w_packed, const float* bias, float* y,
                  void* workspace, size_t workspace_bytes, const MfConvDesc* d, void* stream);

/* --- MF_CONV_FP32_F16X2 (same reference call sites as mf_conv2d_f32: conv_blocks.py:185,238,66,123-125, unet2.py:259)
 * mf_split_f16x2: fp32 [rows][per_row] (per_row % 8 == 0) -> the fp16-pair form (4 bytes per element), row r scaled by
 *   2^-(floor(log2 bound[r]) - 14) (bound NULL: no scaling).  Used for packed weights (rows = 1, bound = max|w|) once at load time and for
 *   activations no producer kernel has split (rows = N samples).
 * mf_conv2d_f16x2: x1s / x2s / ws in fp16-pair form with their bounds (x*_bound: [N] floats or NULL = unscaled; w_bound: the scalar the
 *   weights were split with, 0 = unscaled); y fp32 NHWC.  gn_partial optional: statistics of the following GroupNorm(G), [N][parts][G][2]
 *   doubles, parts = mf_conv2d_gn_parts(d, G) > 0.  y_bound optional ([N][slots] floats, slots = mf_conv2d_f16x2_bound_slots(d) > 0,
 *   not together with gn_partial): every (tile, wave) -- or reducer wave -- stores the max |y| of its share of a sample; reduce with
 *   mf_bound_finalize_f32 to the operand bound of y for consumers that take it un-normalised.  Split-K plans reduce through `workspace`
 *   (mf_conv2d_workspace_bytes).  A power-of-two split meets INSIDE the launch: the partial tiles of a pair are handed over through
 *   `workspace` with write-through stores, a counter per pair in `sync` tells the workgroup that arrives second to add its partner's tile
 *   (a + b does not depend on who adds: deterministic), the last one runs the ordinary epilogue -- no reducer launch.  `sync`:
 *   mf_conv2d_f16x2_sync_words(d) 32-bit words, zero before the FIRST launch and left zero by every launch (caller keeps them; one array
 *   per stream is enough).  Other splits (or GroupNorm statistics the epilogue cannot emit; or MF_CONV_TREE=0 in the environment, read once)
 *   take the slab + reducer pass, the reducer emitting y, the statistics and y_bound; it sums the slabs in the same pairwise order, so the
 *   two paths give the same bits.
 * mf_conv2d_plan_query: the tile id and split-K factor the planner picks for `d` (any precision; 0, 0 = not on the implicit-GEMM path). */
int mf_conv2d_f16x2_ok(const MfConvDesc* d);
int mf_conv2d_f16x2_bound_slots(const MfConvDesc* d);
int mf_split_f16x2(const float* x, void* xs, const float* bound, int rows, int64_t per_row, void* stream);
/* The same split when the bound of every row still lies as slot maxima -- slots [rows][nslots]: the y_bound array of mf_conv2d_f16x2 or
 * the `partial` array of mf_maxabs_rows_f32 (bound = NULL there) -- reduced inside the pass, which also publishes bound_out[rows]: one
 * launch instead of mf_bound_finalize_f32 + mf_split_f16x2 (7 launches fewer per denoise iteration of the published UNet). */
int mf_split_f16x2_slots(const float* x, void* xs, const float* slots, int nslots, float* bound_out, int rows, int64_t per_row, void* stream);
int mf_conv2d_f16x2_sync_words(const MfConvDesc* d);
int mf_conv2d_f16x2(const void* x1s, const void* x2s, const void* ws, const float* bias, float* y, const float* x1_bound,
                    const float* x2_bound, float w_bound, float* y_bound, void* workspace, size_t workspace_bytes, uint32_t* sync,
                    double* gn_partial, int G, const MfConvDesc* d, void* stream);
int mf_conv2d_plan_query(const MfConvDesc* d, int32_t* tile_id, int32_t* splitk);
/* A run-time override of the planner's choice for the shape of `d` (ABI 220; tile = 0 removes it): consulted before the built-in table when the
 * descriptor carries no hints.  For tuning inside the real pipeline (scripts/plan_tune.py) -- the table is what ships. */
/* (a caller that caches plan-derived sizes -- workspace bytes, sync words, bound slots, which pairs share a launch -- drops them when it installs an
 * override: the Python host's per-module caches are cleared by the tuning scripts that use this entry, scripts/plan_tune.py:clear_caches) */
int mf_conv2d_plan_override(const MfConvDesc* d, int tile, int splitk);

/* TWO independent fp16-pair convolutions in ONE launch (ABI 220; BasicResBlock.forward, conv_blocks.py:194-240: `conv_res(x)` (:238) and the 3x3
 * convolution of the block (:185) read the same x and neither reads the other's result).  The grid is a's workgroups followed by b's: a -- the
 * large one -- fills the device as it does alone, b's workgroups take the CUs a's workgroups leave; no kernel boundary between the two, b's ramp
 * under a's drain.  Every workgroup runs the unchanged code of its own convolution, so both results are bit-identical to two mf_conv2d_f16x2
 * calls with the same arguments (each call struct holds exactly those arguments).  mf_conv2d_f16x2_group_ok(a, Ga, b, Gb): can these two plans
 * share a launch (Ga / Gb: groups of the GroupNorm statistics asked of a / b, 0 = none) -- both MF_CONV_FP32_F16X2, both finishing inside their
 * launch (no slab + reducer pass), equal workgroup sizes, a tile pair that is instantiated (the pairs cfg2's channel-changing ResBlocks use;
 * descriptor hints choose b's tile).  When both meet their split-K slices inside the launch, their workspaces and sync arrays must not overlap. */
typedef struct MfConvF16x2Call {
  const void* x1s; const void* x2s; const void* ws; const float* bias; float* y;
  const float* x1_bound; const float* x2_bound; float w_bound; float* y_bound;
  void* workspace; size_t workspace_bytes; uint32_t* sync;
  double* gn_partial; int G;
  const MfConvDesc* d;
} MfConvF16x2Call;
int mf_conv2d_f16x2_group_ok(const MfConvDesc* a, int Ga, const MfConvDesc* b, int Gb);
int mf_conv2d_f16x2_group(const MfConvF16x2Call* a, const MfConvF16x2Call* b, void* stream);

/* Convolution + GroupNorm + Swish + residual + embedding in ONE launch (ABI 220; BasicBlock.forward / BasicResBlock.forward,
 * conv_blocks.py:185-191,236-240, with the `x += emb` of :360-363): the fp16-pair convolution whose workgroups keep their final tile in
 * registers, publish the tile's partial GroupNorm sums (gn_partial, as in mf_conv2d_f16x2), meet the other tiles of their SAMPLE at a
 * counter, finalize mean / rstd from the sample's records and apply act(gn(y) gamma + beta) + residual + emb[n][c] on the way out -- what
 * mf_conv2d_f16x2 + mf_gn_apply_from_partials_pairs_f32 compute in two launches, bit for bit (same records, same summation order, same
 * per-element operations); the un-normalised y is never written.  Workgroups WAIT for each other, so the form exists only for plans whose
 * whole grid is resident on the device at once: mf_conv2d_f16x2_fuse_words(d, G) = the zero-initialised 32-bit words `rendezvous` needs
 * (2 per sample; every launch leaves them zero; one array per stream), 0 = this convolution cannot (use the two-launch form).
 * `error_flag`: one zero-initialised word; a wait that does not complete within 50 ms (another process's waiting workgroups filling the
 * device) sets it, later launches stop waiting, and the results since are INVALID: the caller checks it at the end of its loop, zeroes
 * flag and rendezvous words, and re-runs on the two-launch form.  Arguments as in the two calls it replaces; out may be NULL (fp16 pairs
 * only), out_split / out_bound are required.  MEASURED (round 4, profiles/r04_fused_gn_apply_ab.txt, r04_conv_timeline_fused.txt): not
 * faster than the two launches on MI355X -- the tail (records, rendezvous, finalize, apply at one wave per SIMD) costs what the kernel
 * boundary plus the stand-alone pass cost -- so the Python host keeps it opt-in (MEDFUSION_FUSED_APPLY=1). */
typedef struct MfGnFuse {
  const float* gamma;            /* [Cout] or NULL (both) */
  const float* beta;
  const float* residual;         /* fp32 NHWC [N][Ho][Wo][Cout], or NULL */
  const void* residual_pairs;    /* the residual as fp16 pairs scaled by res_bound, or NULL */
  const float* res_bound;        /* [N], or NULL with res_bound_slots */
  const float* res_bound_slots;  /* [N][res_nslots] slot maxima (y_bound of the convolution that made the residual) */
  const float* emb;              /* emb[n * emb_stride + c] or NULL */
  const float* emb_bound;        /* [N] */
  float* out;                    /* fp32 result or NULL */
  void* out_split;               /* fp16-pair result */
  float* out_bound;              /* [N] written */
  uint32_t* rendezvous;
  uint32_t* error_flag;
  int64_t emb_stride;
  int32_t res_nslots;
  int32_t act;                   /* 1: Swish */
  float bconst;                  /* >= max |act(gn(y) gamma + beta)| (mf_gn_apply_from_partials_f32) */
  float eps;
} MfGnFuse;
int mf_conv2d_f16x2_fuse_words(const MfConvDesc* d, int G);
int mf_conv2d_f16x2_gn_apply(const void* x1s, const void* x2s, const void* ws, const float* bias, const float* x1_bound, const float* x2_bound,
                             float w_bound, void* workspace, size_t workspace_bytes, uint32_t* sync, double* gn_partial, int G,
                             const MfGnFuse* f, const MfConvDesc* d, void* stream);

/* The network input as an fp16-pair operand (ABI 220; the NCHW latent of unet2.py:246 / the image of latent_embedders.py:757): x NCHW
 * [N][C][HW], C <= CP -> the pair form of the NHWC tensor [N][HW][CP] (CP % 32 == 0, channels C .. CP-1 zero) scaled per sample by its own
 * max |x|, which the launch measures and publishes (bound_out[N]); one workgroup per sample (C * HW <= 2^18).  With the convolution's
 * weights zero-padded to CP input channels (host, once) the input convolution runs on mf_conv2d_f16x2 like the rest of the network. */
int mf_pack_nchw_pairs_f32(const float* x, void* out_pairs, float* bound_out, int N, int C, int HW, int CP, void* stream);
/* The same convolution writing its output ALSO as fp16 pairs (ABI 220), for outputs that feed convolutions un-normalised (BasicDown /
 * BasicUp, conv_blocks.py:66,123-125): the per-sample scale comes from a bound DERIVED from the operands, |y[n]| <= x1_bound[n] w_l1_1 +
 * x2_bound[n] w_l1_2 + bias_max with w_l1 = max over the output channels of the L1 norm of the filter over that source's channels (host,
 * once per weight tensor) -- rigorous, a few binades above the true maximum (the pair format keeps its 23 bits down to 2^-28 of the bound),
 * so no measuring epilogue, no mf_split_f16x2_slots launch behind the convolution.  y_bound_out[N] receives the bound.  Only for plans
 * whose final values exist inside the launch: mf_conv2d_f16x2_pairs_out_ok(d). */
int mf_conv2d_f16x2_pairs_out_ok(const MfConvDesc* d);
int mf_conv2d_f16x2_pairs_out(const void* x1s, const void* x2s, const void* ws, const float* bias, float* y, void* y_split, float* y_bound_out,
                              const float* x1_bound, const float* x2_bound, float w_bound, float w_l1_1, float w_l1_2, float bias_max,
                              void* workspace, size_t workspace_bytes, uint32_t* sync, const MfConvDesc* d, void* stream);

/* --- Winograd F(2x2, 3x3) form of the 3x3 stride-1 pad-1 convolutions on the fp16-pair arithmetic (ABI 230; conv_blocks.py:185 as reached
 * from unet2.py:250-264 -- the 3x3 convolutions of the UnetResBlocks, 94.4 % of the UNet's FLOPs).
 *   y = A^T [ sum_c (G g G^T) . (B^T d B) ] A   per 2x2 output tile: 16 products per (tile, channel pair) instead of 36 (2.25x fewer matrix
 * instructions).  The 16 element-wise products summed over the input channels are 16 independent GEMMs; they run on the kernel of
 * mf_conv2d_f16x2 (fp16-pair operands, three product terms, fp32 accumulate, in-launch split-K tree), each row block with the weight slab of
 * its component.  Same values as the direct form to fp32 rounding -- a DIFFERENT summation, not the same bits (error vs fp64 in
 * profiles/r05_winograd_ab.txt); MF_CONV_FP32 / MF_CONV_FP32_SPLIT3_W3 (the exact arithmetics) never take this form.
 *  mf_wino_ok(d): can `d` (3x3, stride 1, pad 1, upsample 0, NHWC, precision MF_CONV_FP32_F16X2, H and W even, C1 % 32 == C2 % 32 == 0,
 *    Cout in {64, 128, 256, 512, 1024}) run so?  mf_wino_preferred(d): ... AND is it the faster form on MI355X?  ABI 240: answered for ANY batch by
 *    a rule in (Cin, Cout, H W, N) fitted to the sweeps at B = 4 ... 69 (Cin Cout >= 190 (Cin + Cout), csrc/conv_f16x2_wino.inc) instead of the
 *    exact-shape table of ABI 230 (csrc/wino_plan_table.inc, kept as the record: mf_wino_in_table(d) says whether `d` is one of its rows).
 *  mf_wino_pack_weight_f32: OIHW 3x3 -> U = G g G^T, [16 components][Cout][Cin] fp32 (fp64 arithmetic, one rounding); split it with
 *    mf_split_f16x2(rows = 1, bound = max |U|) like any packed weight.
 *  mf_wino_input_f16x2: the fp16-pair form xs of an NHWC activation (scaled per sample by x_bound[N]) -> V = B^T d B as fp16 pairs
 *    [16][N][(H/2)(W/2)][C], component k of sample n scaled by v_bound[k N + n] = 4 x_bound[n] (v_bound: 16 N floats, written).
 *  mf_conv2d_wino_f16x2: v1s / v2s (the second source of a fused channel concat, or NULL) with their v*_bound arrays, us = the split U with the
 *    scalar u_bound it was split with; y fp32 NHWC [N][H][W][Cout] = conv3x3(x1 ++ x2) + bias.  gn_partial optional: [N][parts][G][2] doubles,
 *    parts = mf_wino_gn_parts(d, G) > 0 -- the records mf_gn_apply_from_partials_* reads.  workspace >= mf_wino_workspace_bytes(d) (the GEMM's
 *    output in the transform domain + its split-K hand-off region), sync = mf_wino_sync_words(d) zero-initialised words as for mf_conv2d_f16x2.
 *    d->tile_hint / splitk_hint address the component GEMM.  Two launches: the GEMM, then A^T M A + bias + statistics. */
int mf_wino_ok(const MfConvDesc* d);
int mf_wino_preferred(const MfConvDesc* d);
int mf_wino_in_table(const MfConvDesc* d);
int mf_wino_pack_weight_f32(const float* w_oihw, float* u, int Cout, int Cin, void* stream);
int mf_wino_input_f16x2(const void* xs, const float* x_bound, void* vs, float* v_bound, int N, int H, int W, int C, void* stream);
size_t mf_wino_workspace_bytes(const MfConvDesc* d);
int mf_wino_sync_words(const MfConvDesc* d);
int mf_wino_gn_parts(const MfConvDesc* d, int G);
int mf_wino_plan_query(const MfConvDesc* d, int32_t* tile_id, int32_t* splitk);
/* --- the Winograd form on the EXACT arithmetics (ABI 250): MF_CONV_FP32_SPLIT3_W3 and MF_CONV_FP32 take plain fp32 operands, so the three pieces are
 * separate calls on fp32 tensors (same reference call sites as above: conv_blocks.py:185-191,236-240,360-363 via unet2.py:250-264):
 *   mf_wino_f32_ok(d, G): can the 3x3 `d` (precision 0 or 3, stride 1, pad 1, NHWC, H and W even, n T % 64 == 0) with the G-group GroupNorm behind it run so?
 *   mf_wino_input_f32: x fp32 NHWC -> V = B^T d B fp32 [16][N][(H/2)(W/2)][C]
 *   the 16 component GEMMs: mf_conv2d_f32 with the descriptor {N = 16 n, Hin = 1, Win = T, C1, C2, Cout, KH = KW = 1, stride 1, pad 0, upsample = 3, NHWC,
 *     precision 0 | 3}: x1 / x2 = V of the two sources, w = U = G g G^T [16][Cout][Cin] from mf_wino_pack_weight_f32 (precision 3: split by
 *     mf_split_conv_weight_bf16x3 over rows = 16 Cout), bias NULL, y = M fp32 [16][n T][Cout]; rows [k n T, (k + 1) n T) use weight slab k
 *   mf_wino_tail_f32: M -> y = A^T M A + bias -> GroupNorm over the whole (sample, group) -> Swish (act = 1) -> + residual (fp32 NHWC or NULL) -> + emb row;
 *     out fp32 NHWC and, if out_wino != NULL, V of the result for the next Winograd convolution.  The kernel is the tail of the fp16-pair form. */
int mf_wino_f32_ok(const MfConvDesc* d, int G);
int mf_wino_input_f32(const float* x, float* v, int N, int H, int W, int C, void* stream);
int mf_wino_tail_f32(const float* m, const float* bias, const float* gamma, const float* beta, const float* residual, const float* emb, int64_t emb_stride,
                     float* out, float* out_wino, int N, int H, int W, int C, int G, int act, float eps, void* stream);
int mf_conv2d_wino_f16x2(const void* v1s, const void* v2s, const void* us, const float* bias, float* y, const float* v1_bound, const float* v2_bound,
                         float u_bound, void* workspace, size_t workspace_bytes, uint32_t* sync, double* gn_partial, int G, const MfConvDesc* d,
                         void* stream);
/* The Winograd convolution WITH what follows it in a ResBlock (BasicBlock.forward / BasicResBlock.forward, conv_blocks.py:185-191,236-240, and
 * the `x += emb` of :360-363): GroupNorm(G) + Swish + residual + embedding row -- and, optionally, the INPUT TRANSFORM of the next Winograd
 * convolution -- in the launch behind the GEMM: one workgroup per (sample, group) holds all pixels of the sample x the group's channels in LDS,
 * so the statistics are complete inside the workgroup (no records, no apply pass, no un-normalised y in memory).  Outputs, each optional: `out`
 * fp32 NHWC, `out_split` its fp16-pair form scaled per sample by out_bound[n] = bconst + bound(residual) + bound(embedding row) (as
 * mf_gn_apply_from_partials_pairs_f32 derives it), `out_wino` = B^T d B of the result as mf_wino_input_f16x2 would make it from out_split (bit
 * for bit; [16][N][(H/2)(W/2)][Cout], wino_bound: 16 N floats).  mf_wino_tail_ok(d, G): mf_wino_ok(d), whole 8-channel groups per GroupNorm
 * group and H W (Cout / G) floats <= 64 KB.  The per-element arithmetic is that of the apply pass; the statistics are summed in another order
 * than the records of mf_conv2d_wino_f16x2 (same values to fp64 rounding). */
typedef struct MfWinoTail {
  const float* gamma; const float* beta;                    /* [Cout], both or neither */
  const float* residual; const void* residual_pairs;         /* fp32 NHWC, or fp16 pairs scaled by res_bound, or neither */
  const float* res_bound; const float* res_bound_slots;      /* [N], or [N][res_nslots] slot maxima (mf_conv2d_f16x2 y_bound) */
  const float* emb; const float* emb_bound;                  /* [N][emb_stride] rows added per (n, c) + their bounds [N], or NULL */
  float* out; void* out_split; float* out_bound;
  void* out_wino; float* wino_bound;
  int64_t emb_stride;
  int32_t res_nslots, act;                                   /* act 1: Swish */
  float bconst, eps;                                         /* bconst >= max |act(gn(y) gamma + beta)| (host constant), GroupNorm eps */
} MfWinoTail;
int mf_wino_tail_ok(const MfConvDesc* d, int G);
/* `guest` (optional): a second, independent fp16-pair convolution -- conv_res of the same ResBlock (conv_blocks.py:238), an ordinary
 * MfConvF16x2Call as for mf_conv2d_f16x2_group -- whose workgroups share the component GEMM's launch (the GEMM's first, the guest's on the CUs
 * they leave); bit-identical to its own mf_conv2d_f16x2 call; its output may be the tail's residual (the tail launches behind both).  Ask
 * mf_wino_group_ok(d, guest->d) first; workspaces and sync arrays of the two must not overlap. */
int mf_wino_group_ok(const MfConvDesc* d, const MfConvDesc* guest);
int mf_conv2d_wino_gn_apply_f16x2(const void* v1s, const void* v2s, const void* us, const float* bias, const float* v1_bound, const float* v2_bound,
                                  float u_bound, void* workspace, size_t workspace_bytes, uint32_t* sync, int G, const MfWinoTail* t,
                                  const MfConvF16x2Call* guest, const MfConvDesc* d, void* stream);

/* Convolution with the statistics of the FOLLOWING GroupNorm (G groups over Cout) fused in: per-tile sums from the
 * epilogue, or from the split-K reducer when the plan splits K.  gn_partial: [N][parts][G][2] doubles {sum, sumsq},
 * parts = mf_conv2d_gn_parts(d, G); 0 means this convolution cannot emit them (use mf_gn_stats_partial_f32). */
int mf_conv2d_gn_parts(const MfConvDesc* d, int G);
int mf_conv2d_gn_f32(const float* x1, const float* x2, const float* w_packed, const float* bias, float* y, void* workspace,
                     size_t workspace_bytes, double* gn_partial, int G, const MfConvDesc* d, void* stream);

/* ------------------------------------------------------------------ GroupNorm + Swish + residual + embedding
 * Replaces nn.GroupNorm + MONAI Swish + `out + residual` + `x += emb` at conv_blocks.py:186-191,
 * :236-240, :360-363 (UnetResBlock) / :298-301 (UnetBasicBlock).  NHWC.
 * stats[n][g] = {mean, rstd} (biased variance, eps inside the sqrt).
 */
size_t mf_gn_stats_workspace_bytes(int N, int HW, int C, int G);
int mf_gn_stats_f32(const float* x, float* stats, void* workspace, size_t workspace_bytes, int N, int HW, int C, int G,
                    float eps, void* stream);
/* Form used on the hot path: partial sums [N][parts][G][2] (doubles; parts = mf_gn_partial_parts(HW), or emitted by
 * mf_conv2d_gn_f32 / mf_conv2d_f16x2), a tiny finalize kernel, then the apply pass. */
int mf_gn_partial_parts(int HW);
int mf_gn_stats_partial_f32(const float* x, double* partial, int N, int HW, int C, int G, void* stream);
/* partial sums -> stats[n][g] = {mean, rstd} (tiny kernel; mf_gn_apply_from_partials_f32 below folds it into the apply pass) */
int mf_gn_finalize_f32(const double* partial, int parts, float* stats, int N, int HW, int C, int G, float eps, void* stream);
/* out = act(gn(x) * gamma + beta) + residual + emb[n*emb_stride + c]; act: 0 none, 1 Swish x*sigmoid(x).
 * gamma/beta NULL => no affine; stats NULL => no normalisation; residual / emb NULL => skipped.
 * out may alias x. */
int mf_gn_apply_f32(const float* x, const float* stats, const float* gamma, const float* beta, const float* residual,
                    const float* emb, int64_t emb_stride, float* out, int N, int HW, int C, int G, int act, void* stream);
/* The same pass, also writing the fp16-pair form of `out` (operand of a following MF_CONV_FP32_F16X2 convolution; C % 8 == 0).
 * A fp16-pair tensor carries a per-sample power-of-two scale derived from an UPPER BOUND of |value| over the sample (fp16 stops at
 * 65504, an un-normalised residual stream does not): the pass derives the bound of its output from its inputs --
 *   bound[n] = (stats ? bconst : x_bound[n]) + res_bound[n] + emb_bound[n],  bconst >= max|act(gn(x) gamma + beta)| =
 *   max|gamma| sqrt(group size) + max|beta| -- scales sample n by 2^-(floor(log2 bound[n]) - 14) and publishes bound[n] in out_bound.
 *   The bounds handed in must BE bounds: the from-partials passes no longer clamp to the fp16 range (round 4: 4 VALU instructions per element in
 *   a VALU-bound pass) -- a value above its sample's bound by more than 2x overflows its pair to Inf instead of saturating at 65504 2^s. */
int mf_gn_apply_split_f32(const float* x, const float* stats, const float* gamma, const float* beta, const float* residual,
                          const float* emb, int64_t emb_stride, float* out, void* out_split, const float* x_bound, const float* res_bound,
                          const float* emb_bound, float bconst, float* out_bound, int N, int HW, int C, int G, int act, void* stream);
/* mf_gn_finalize_f32 + mf_gn_apply_split_f32 in ONE launch: mean / rstd are reduced from the partial records of the producing convolution
 * ([N][parts][G][2], mf_conv2d_gn_f32 / mf_conv2d_f16x2) by the first G threads of every workgroup while its first loads are in flight.
 * out_split NULL => fp32 output only (then the bound arguments are ignored).  The residual's bound is res_bound[N], or -- res_bound NULL --
 * the [N][res_nslots] slot maxima a convolution wrote (y_bound of mf_conv2d_f16x2), reduced inside the pass: no mf_bound_finalize_f32 launch. */
int mf_gn_apply_from_partials_f32(const float* x, const double* gn_partial, int parts, float eps, const float* gamma, const float* beta,
                                  const float* residual, const float* emb, int64_t emb_stride, float* out, void* out_split,
                                  const float* res_bound, const float* res_bound_slots, int res_nslots, const float* emb_bound, float bconst,
                                  float* out_bound, int N, int HW, int C, int G, int act, void* stream);
/* The same pass for tensors that exist ONLY as fp16 pairs between two convolutions (ABI 210; conv_blocks.py:236-240 where the block output
 * feeds convolutions and residual adds alone): `out` may be NULL -- only the pair form out_split is written, 12 instead of 16 bytes per
 * element -- and the residual may be given as pairs (`residual_pairs`, scaled by `res_bound`; `residual` must then be NULL): it is read back
 * as hi + lo/2048 times its scale, i.e. the fp32 value with its last significand bit cleared at most (<= 2^-23 relative, the order of the
 * rounding of the add itself). */
int mf_gn_apply_from_partials_pairs_f32(const float* x, const double* gn_partial, int parts, float eps, const float* gamma, const float* beta,
                                        const float* residual, const void* residual_pairs, const float* emb, int64_t emb_stride, float* out,
                                        void* out_split, const float* res_bound, const float* res_bound_slots, int res_nslots,
                                        const float* emb_bound, float bconst, float* out_bound, int N, int HW, int C, int G, int act, void* stream);
/* bound[n] = max |x[n][:]| over per_row elements: the measured operand bound of tensors no producer bounded analytically (network
 * input convolutions, embedding rows).  Two launches, no atomics: every wave stores the max of its share into its own slot of
 * `partial` (N * mf_maxabs_rows_slots(per_row) floats of caller scratch), then one wave per row reduces the slots.
 * mf_bound_finalize_f32 is that second pass on its own: it turns the slot array a convolution filled (y_bound of mf_conv2d_f16x2)
 * into bound[n]. */
int mf_maxabs_rows_slots(int64_t per_row);
int mf_maxabs_rows_f32(const float* x, float* partial, float* bound, int N, int64_t per_row, void* stream);
int mf_bound_finalize_f32(const float* partial, float* bound, int N, int slots, void* stream);

/* ------------------------------------------------------------------ small dense ops
 * mf_linear_f32: y[b*y_stride + o] = sum_i f(x[b*x_stride + i]) * w[o*In + i] + bias[o] (+ y if accumulate);
 * f = Swish if act_in else identity.  Replaces nn.Linear at time_embedder.py:66-71 and the
 * local_embedder Swish->Linear at conv_blocks.py:340-344,349-353 (all 17 batched into one call by the host).
 */
int mf_linear_f32(const float* x, int64_t x_stride, const float* w, const float* bias, float* y, int64_t y_stride,
                  int B, int In, int Out, int act_in, int act_out, int accumulate, void* stream);
/* SinusoidalPosEmb (time_embedder.py:15-28): out[b][0:half] = sin(t f_k), [half:2half] = cos(t f_k),
 * f_k = exp(-(ln(max_period)/(half - shift)) k); flip swaps halves; odd dim zero-padded.
 * freqs: optional device table f_k[half] precomputed by the host exactly like the reference (a 1-ulp difference
 * in f_k is amplified by t ~ 1000); NULL = compute with device expf. */
int mf_sinusoidal_f32(const float* t, const float* freqs, float* out, int B, int dim, float max_period, float shift, int flip, void* stream);
/* LearnedSinusoidalPosEmb (ABI 220; time_embedder.py:31-49): out[b] = [t_b | sin(2 pi t_b w_k) | cos(2 pi t_b w_k) | 0 if emb_dim is odd],
 * k < emb_dim / 2, row length 1 + 2 (emb_dim / 2) + (emb_dim & 1); the angle is formed in the reference's order ((t w) 2) pi in fp32.
 * (The reference's TimeEmbbeding cannot hold this embedder -- its first Linear takes emb_dim features, this returns emb_dim + 1 -- so it
 * is a stand-alone module there and here.) */
int mf_learned_sinusoidal_f32(const float* t, const float* weights, float* out, int B, int emb_dim, void* stream);
/* LabelEmbedder lookup + save_add (cond_embedders.py:19-24, conv_blocks.py:16-18): io[b][:] += table[idx[b]][:] */
int mf_embedding_add_f32(const float* table, const int64_t* idx, float* io, int B, int D, int num_rows, void* stream);

/* ------------------------------------------------------------------ scheduler step (one fused elementwise pass)
 * Replaces the ~49 ATen ops/step of DiffusionPipeline.forward :240-273 (CFG combine, objective switch),
 * GaussianNoiseScheduler.estimate_x_0 :119-124, estimate_x_T :127-131, estimate_mean_t :104-107,
 * estimate_variance_t :110-116, estimate_x_t_prior_from_x_0 :85-101 and the DDIM update
 * diffusion_pipeline.py:297-304.  Per-step scalars are computed on the host in fp32 exactly as the
 * reference computes them and live in a device table; the step index may come from device memory so a
 * captured hipGraph replays unchanged (no host sync, unlike the reference's `alphas_cumprod[t]`).
 * Products are rounded separately (no FMA contraction) to mirror ATen's elementwise chain.
 */
typedef struct MfSchedStep {
  float sqrt_recip_ac;    /* sqrt_recip_alphas_cumprod[t] */
  float sqrt_recipm1_ac;  /* sqrt_recipm1_alphas_cumprod[t] */
  float coef1, coef2;     /* posterior_mean_coef1/2[t] */
  float std_fixed;        /* exp(0.5*log(clamp(posterior_variance[t],1e-20))), 0 when t == 0 (var_scale == 0 path) */
  float log_var_min;      /* log(clamp(posterior_variance[t])) -- learned-variance path */
  float log_var_max;      /* log(clamp(betas[t])) */
  float ddim_sqrt_an;     /* sqrt(alphas_cumprod[t_next]) */
  float ddim_c;           /* sqrt(1 - alpha_next - sigma^2) */
  float ddim_sigma;       /* eta * sqrt((1 - a/a_next)(1 - a_next)/(1 - a)), eta == 1 */
  int32_t t;              /* timestep value (for t == 0 test) */
  int32_t mode;           /* 0: x_t <- x_t_prior (DDPM / last DDIM iteration); 1: DDIM update */
} MfSchedStep;

typedef struct MfSchedArgs {
  const float* x_t;          /* [n] current latent */
  const float* pred;         /* [n] estimator output (conditional pass when CFG) */
  const float* pred_uncond;  /* [n] or NULL: CFG pred = pu + g*(pred - pu) */
  const float* pred_var;     /* [n] or NULL: learned variance head (estimate_variance=True, g == 1 only) */
  const float* noise_post;   /* base of posterior-noise bank */
  const float* noise_ddim;   /* base of DDIM-noise bank (unused when mode == 0) */
  int64_t noise_step_stride; /* elements between consecutive steps in the banks (0: same buffer each step) */
  float* x_t_out;            /* [n] next latent (may alias x_t) */
  float* x0_out;             /* [n] or NULL: x_0 estimate of this step */
  float* xT_out;             /* [n] or NULL: x_T estimate of this step */
  const MfSchedStep* table;  /* device table, one entry per loop iteration */
  const int32_t* step_dev;   /* device step counter or NULL (then `step` is used) */
  int32_t step;
  int32_t objective;         /* 0: 'x_T', 1: 'x_0' */
  int32_t clip_x0;           /* clamp x_0 to [-1, 1] */
  float guidance_scale;
  int64_t n;                 /* elements */
} MfSchedArgs;
int mf_sched_step_f32(const MfSchedArgs* a, void* stream);
/* The tail of a denoise iteration in ONE launch (ABI 220): the posterior draw and the DDIM draw of mf_philox_normal_f32 (draw indices
 * draw_base + draw_stride * step and + 1, rows sample_offset .. + B) are generated in registers, mf_sched_step_f32's arithmetic runs on
 * them -- bit for bit what the three launches produce -- and *step_counter (read as the step; a->step_dev / a->step are ignored) is
 * incremented by the workgroup that finishes last (`ticket`: one zero-initialised word the launch leaves zero).  a->noise_post / noise_ddim
 * must be NULL.  Replaces torch.randn_like x 2 + the ~49 elementwise ATen ops + the host loop counter of diffusion_pipeline.py:294-304. */
int mf_sched_step_philox_f32(const MfSchedArgs* a, uint64_t seed, int32_t draw_base, int32_t draw_stride, int64_t sample_offset, int B,
                             int32_t* step_counter, uint32_t* ticket, void* stream);
/* out[0..n) = table[step] with step = *step_dev (or `step`): `t.expand(B)` of diffusion_pipeline.py:294 inside a captured graph */
int mf_broadcast_from_table_f32(const float* table, const int32_t* step_dev, int32_t step, float* out, int n, void* stream);
/* out[b][:] = table[step][cols[b]][:] for a [S][ncol][row_len] table, step = *step_dev (or `step`): the per-iteration gather of the
 * embedding rows UNet.precompute_embeddings hoisted out of the loop (unet2.py:229-241, conv_blocks.py:340-353), usable inside a captured graph. */
int mf_gather_step_rows_f32(const float* table, const int64_t* cols, const int32_t* step_dev, int32_t step, int ncol, int64_t row_len,
                            float* out, int B, void* stream);
/* Up to three such gathers in ONE launch (ABI 220): the embedding rows, the local-embedder rows and their bounds share `cols` and the step. */
int mf_gather_step_rows3_f32(const float* const* tables, const int64_t* row_lens, float* const* outs, int n_tables, const int64_t* cols,
                             const int32_t* step_dev, int32_t step, int ncol, int B, void* stream);
/* *counter += inc (one thread); lets a captured graph advance its own step index. */
int mf_counter_add_i32(int32_t* counter, int32_t inc, void* stream);

/* Per-row scaled combination: out[b][:] = clamp?((a[b]*x[b][:] + c[b]*y[b][:]) / d[b]); y, a, c, d optional (NULL).
 * The scheduler's tensor-level API with a timestep PER ROW (gaussian_scheduler.py:61-77 estimate_x_t, :104-107, :119-131)
 * and the lerp of DiffusionPipeline.interpolate (diffusion_pipeline.py:329); coefficient rows are gathered on the host like
 * `extract()` (scheduler_base.py:43-46).  Bit-exact vs ATen's elementwise chain. */
int mf_rows_axpby_f32(const float* x, const float* y, const float* a, const float* c, const float* d, float* out, int B,
                      int64_t per_row, int do_clamp, float lo, float hi, void* stream);

/* ------------------------------------------------------------------ noise
 * Counter-based standard normals (Philox-4x32-10 + Box-Muller), shard-invariant: element quad q of sample
 * (sample_offset + b) in draw `draw` depends only on (seed, draw, sample index, q).  Stands in for
 * torch.randn_like at diffusion_pipeline.py:315(x_final), gaussian_scheduler.py:99, diffusion_pipeline.py:303.
 * draw index = draw_base + draw_stride * step where step = *step_dev (or `step`).
 */
int mf_philox_normal_f32(float* out, uint64_t seed, int32_t draw_base, int32_t draw_stride, const int32_t* step_dev,
                         int32_t step, int64_t sample_offset, int B, int64_t per_sample, void* stream);

/* ------------------------------------------------------------------ attention family (use_attention = 'linear'|'spatial')
 * mf_attention_f32: compute_attention (attention_blocks.py:35-43) on NHWC-token tensors:
 * q [B][Nq][H*d], k,v [B][Nk][H*d] -> out [B][Nq][H*d]; softmax((q s)(k s)^T) v per head, s = d^-0.25.
 */
int mf_attention_f32(const float* q, const float* k, const float* v, float* out, int B, int H, int Nq, int Nk, int d,
                     float scale, void* stream);
/* LayerNorm over the last dim (attention_blocks.py:22), rows x C */
int mf_layernorm_f32(const float* x, const float* gamma, const float* beta, float* out, int64_t rows, int C, float eps,
                     void* stream);
/* GEGLU gate (attention_blocks.py:23-24): out[r][c] = h[r][c] * gelu(h[r][C + c]), h is rows x 2C */
int mf_geglu_f32(const float* h, float* out, int64_t rows, int C, void* stream);
/* out = a + b (elementwise, n floats) -- the `x + out` skips of attention_blocks.py:124,193,230,287 */
int mf_add_f32(const float* a, const float* b, float* out, int64_t n, void* stream);

/* ------------------------------------------------------------------ layout + egress
 * NCHW <-> NHWC for API-edge tensors and tests. */
int mf_nchw_to_nhwc_f32(const float* x, float* y, int N, int C, int H, int W, void* stream);
int mf_nhwc_to_nchw_f32(const float* x, float* y, int N, int C, int H, int W, void* stream);
/* DiagonalGaussianDistribution (latent_embedders.py:22-27): z = mean + exp(0.5*clamp(logvar,-30,20)) * noise,
 * moments NCHW [N][2C][HW] -> z [N][C][HW] */
int mf_diag_gaussian_sample_f32(const float* moments, const float* noise, float* z, int N, int C, int HW, void* stream);
/* its KL term (latent_embedders.py:29-31; ABI 220): kl[0] = 0.5 * sum(mean^2 + exp(logvar) - 1 - logvar) / N, logvar clamped like above, fp64
 * accumulation -- the `emb_loss` VAE.forward returns (:778), evaluation-time only */
int mf_diag_gaussian_kl_f32(const float* moments, float* kl, int N, int C, int HW, void* stream);

/* learnable_interpolation=False (ABI 210): BasicDown = nn.AvgPool2d(k, stride, get_padding(k, stride)) (conv_blocks.py:57-63; count_include_pad
 * like torch's default: the divisor counts the padding), BasicUp = F.interpolate(nearest-exact) to twice the size (conv_blocks.py:128-130).
 * NHWC fp32, C % 4 == 0. */
int mf_avgpool2d_nhwc_f32(const float* x, float* y, int N, int H, int W, int C, int k, int stride, int pad, void* stream);
int mf_upsample_nearest2x_nhwc_f32(const float* x, float* y, int N, int H, int W, int C, void* stream);
/* The `use_res` skips of BasicDown / BasicUp (conv_blocks.py:54-55,68-69 and :114-115,125-126), added in place to the convolution's output:
 * y [N][H/2][W/2][4C] += PixelUnshuffle(2)(x [N][H][W][C])   (channel c * 4 + dy * 2 + dx <- pixel (2h + dy, 2w + dx), torch's order), H, W even;
 * y [N][2H][2W][C/4]  += PixelShuffle(2)(x [N][H][W][C]),    C % 4 == 0. */
int mf_pixel_unshuffle2_add_nhwc_f32(const float* x, float* y, int N, int H, int W, int C, void* stream);
int mf_pixel_shuffle2_add_nhwc_f32(const float* x, float* y, int N, int H, int W, int C, void* stream);

/* Image egress on the device: NCHW float -> NHWC uint8.
 * mode 0: scripts/helpers/sample_dataset.py:44-53  clip(-1,1) -> (x+1)/2*255 -> astype(uint8)   (bit-exact vs numpy)
 * mode 1: scripts/sample.py:49-51 + torchvision save_image(normalize=True, scale_each=True): (x+1)/2, clamp(0,1), per-image
 *         min-max, mul(255).add(0.5).clamp(0,255).to(uint8); minmax_ws = 2*N floats of caller scratch. */
int mf_image_egress_u8(const float* x_nchw, uint8_t* out_nhwc, float* minmax_ws, int N, int C, int H, int W, int mode, void* stream);

/* ------------------------------------------------------------------ built-in launch timing (bench / roofline)
 * When enabled every launch is bracketed by hipEvents on its own stream and attributed to a kernel family.
 * NOT capture-safe; off by default. */
enum { MF_FAM_CONV_IGEMM = 0, MF_FAM_CONV_DIRECT, MF_FAM_SPLITK_REDUCE, MF_FAM_GN_STATS, MF_FAM_GN_APPLY, MF_FAM_LINEAR,
       MF_FAM_SCHED, MF_FAM_NOISE, MF_FAM_ATTENTION, MF_FAM_MISC, MF_FAM_CONV_GN_FUSED, MF_FAM_WINO_XFORM, MF_FAM_COUNT };
int mf_prof_enable(int on);
int mf_prof_reset(void);
/* synchronises outstanding events; returns summed ms, launch count, algorithmic flops and bytes of a family */
int mf_prof_query(int family, double* ms, int64_t* launches, double* flops, double* bytes);
/* the same plus the flops the hardware EXECUTES for those launches (matrix terms per product x the MACs actually done) */
int mf_prof_query2(int family, double* ms, int64_t* launches, double* flops, double* bytes, double* exec_flops);
const char* mf_prof_family_name(int family);
/* The same records grouped by KERNEL INSTANTIATION (ABI 230): `tag` identifies the instantiation -- for the fp16-pair convolution family its tile id
 * (a grouped launch: 1000 + 100 host tile + guest tile; the single-term mode: + 100000) --, `variant` 1 marks the component GEMMs of the Winograd
 * form (same kernel, same name in a profiler).  mf_prof_rows fills up to max_rows rows and returns how many; mf_prof_tag_name writes the kernel's
 * name as rocprofv3 --kernel-trace prints it, so that a row can be checked against the committed kernel_stats.csv. */
typedef struct MfProfRow { int32_t tag, variant; int64_t launches; double ms, flops, bytes, exec_flops; } MfProfRow;
int mf_prof_rows(int family, MfProfRow* rows, int max_rows);
int mf_prof_tag_name(int family, int tag, char* buf, int n);
/* Measurement aid: what the fp16 matrix pipe SUSTAINS on this device for given operand data.  Launches `workgroups` x 512 threads running
 * nothing but v_mfma_f32_32x32x16_f16 from registers (4 independent chains per wave, `iters` instructions per chain; operands = the first
 * workgroups * 512 * 8 sixteen-byte groups of fp16 at `operands`, loaded once; `out`: workgroups * 512 floats) and reports the flops of the
 * launch; the caller times it.  On MI355X random operands sustain ~60-65 % of the nominal 2516.8 TF, zeros ~97 % (profiles/): the matrix
 * pipe is power-limited by the data it multiplies, and bench.py reports the conv kernel against both numbers. */
int mf_mfma_rate_probe_f16(const void* operands, float* out, int workgroups, int iters, double* flops, void* stream);

/* ------------------------------------------------------------------ command lists (ABI 210): the loop body of denoise() replayed from C
 * Replaces the Python loop of diffusion_pipeline.py:290-305 for every iteration after the second: the host side of one iteration is ~160
 * launches through this ABI (2.0 ms of Python -> ctypes work, profiles/r02_host_enqueue_time.txt), all of whose per-iteration values -- t,
 * the scheduler record, the Philox draw indices, the rows of the hoisted embedding table -- come from a DEVICE step counter that the
 * iteration advances itself.  mf_cmdlist_begin() makes the calling thread record every launch it issues through this library (the
 * launches still execute): kernel, grid, block, LDS size and a copy of the kernarg bytes, in issue order.  mf_cmdlist_end() returns the
 * list; mf_cmdlist_replay(list, times, stream) re-issues it `times` times on `stream` -- plain stream-ordered launches (no graph node
 * latency), ~1 us of host time each.  The caller guarantees what a graph capture would: every pointer recorded stays valid and means the
 * same buffer during the replays (pipeline.py records inside a private torch memory pool), and nothing but launches of this library makes
 * up the iteration.  Not recorded: hipFuncSetAttribute (done by the eager warm-up iteration), launch timing (begin refuses while
 * mf_prof_enable is on).  Lists are immutable after end(), replay is thread-safe, free() releases. */
int mf_cmdlist_begin(void);
int mf_cmdlist_end(void** list);
int mf_cmdlist_count(const void* list);
int mf_cmdlist_replay(const void* list, int times, void* stream);
int mf_cmdlist_free(void* list);

#ifdef __cplusplus
}
#endif
#endif /* MEDFUSION_HIP_H */
