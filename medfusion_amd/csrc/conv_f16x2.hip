// conv_f16x2.hip -- MF_CONV_FP32_F16X2: planner, launch and C-ABI of the fp16-pair implicit-GEMM convolution (kernel: conv_f16x2.h).
// Replaces torch.nn.Conv2d.forward at conv_blocks.py:185,238,66,123-125 and unet2.py:259 (include/medfusion_hip.h has the call-site map).
#include "common.h"
#include <string.h>
#include <mutex>
#include <vector>
#include "gn_partial.h"
#include "conv_plan.h"
#include "conv_f16x2.h"
#include "conv_f16x2_halo.h"
#include "conv_f16x2_group.h"
#include "winograd.h"

using namespace mf;

namespace {

// ------------------------------------------------------------------ MF_CONV_FP32_F16X2: planning and launch (kernel: conv_f16x2.h)
// HG = 0: conv_f16x2_kernel, LDS per workgroup = NST * (BM + BN) * 128 bytes (<= 80 KB: two workgroups per CU).
// HG > 0: conv_halo_kernel (conv_f16x2_halo.h; 3x3 stride 1 only): the tile is a block of whole image rows, HG 1-KB halo pieces per wave
//         and chunk; LDS = 2 HG NW KB (two halo buffers) + 3 BN * 128 (weight ring).
struct Tile2 { int id, BM, BN, WM, WN, NST, HG; };
const Tile2 kTiles2[] = {
    {31, 128, 256, 2, 4, 3, 0}, {32, 256, 128, 4, 2, 3, 0}, {33, 128, 128, 2, 4, 3, 0}, {34, 128, 128, 4, 2, 3, 0}, {35, 256, 64, 4, 2, 3, 0},
    {36, 128, 64, 4, 2, 3, 0}, {37, 64, 256, 1, 8, 3, 0},
    // (a four-stage 128 x 128 tile -- one more chunk of look-ahead for the cold weight streams of the Winograd component GEMMs -- measured
    // 46.2 vs 45.6 us on the 8 x 8 GEMM + tail: not kept; profiles/r05_small_ab.txt)
    // 4-wave workgroups, two per CU (independent barrier cadences on the two waves of a SIMD)
    {51, 128, 128, 2, 2, 2, 0}, {52, 128, 128, 2, 2, 3, 0}, {53, 64, 128, 2, 2, 3, 0}, {54, 128, 64, 2, 2, 3, 0},
    // halo tiles: activations staged once per chunk
    {61, 256, 128, 4, 2, 3, 6}, {62, 256, 128, 4, 2, 3, 7}, {63, 128, 128, 2, 4, 3, 4}, {64, 128, 128, 2, 4, 3, 5},
};
inline size_t tile_lds(const Tile2& t) {
  return t.HG ? (size_t)2 * t.HG * (t.WM * t.WN) * 1024 + (size_t)3 * t.BN * 128 : (size_t)t.NST * (t.BM + t.BN) * 128;
}
inline int wgs_per_cu(const Tile2& t) { return tile_lds(t) <= 80 * 1024 ? 2 : 1; }
// can the halo kernel take this convolution on tile t: 3x3 stride 1 pad 1, the tile a block of whole rows of one image or of whole small
// images, its halo inside the HG pieces
inline bool halo_fits(const MfConvDesc* d, const Tile2& t) {
  if (!t.HG) return true;
  if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || d->upsample != 0) return false;
  const int W = d->Win, HW = d->Hin * d->Win;
  int R, segs;
  if (HW >= t.BM) {
    if (HW % t.BM || t.BM % W) return false;
    R = t.BM / W; segs = 1;
  } else {
    if (t.BM % HW) return false;
    R = d->Hin; segs = t.BM / HW;
  }
  return (long)segs * (R + 2) * (W + 2) <= 8L * t.WM * t.WN * t.HG;
}

// MF_CONV_HALO: 1 lets the planner's cost model consider the halo tiles; 0 (default: measured not faster than the 9-copy kernel on any
// cfg2 shape but one, profiles/r03_conv_halo_sweep.txt) keeps them for explicit tile hints
inline bool halo_auto() {
  static const int env = [] { const char* e = getenv("MF_CONV_HALO"); return e ? atoi(e) : 0; }();
  return env != 0;
}
// MF_PLAN_MODEL: 1 (default) the cost model fitted in round 6 to the sweeps at B = 8 ... 200; 0 the round-5 model (A/B, scripts/plan_model_fit.py --model old)
inline int plan_model() {
  static const int env = [] { const char* e = getenv("MF_PLAN_MODEL"); return e ? atoi(e) : 1; }();
  return env;
}
struct PlanEntry { int N, H, W, Cin, Cout, k, stride, ups, tile, sk; };
const PlanEntry kPlanTable[] = {
#include "conv_plan_table.inc"
    {0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
};

// plan overrides set at run time (mf_conv2d_plan_override: the in-pipeline tuner scripts/plan_tune.py tries candidate plans per shape without a
// rebuild); consulted before the static table
std::vector<PlanEntry> g_plan_overrides;
std::mutex g_plan_mu;

struct Plan2 {
  bool ok;
  Tile2 t;
  int splitk, cg_per_split, cgroups, taps;
  int Hout, Wout, Heff, Weff, M, K;
};

int make_plan2(const MfConvDesc* d, Plan2* pl) {
  Plan g;
  int rc = fill_geometry(d, &g);
  if (rc) return rc;
  pl->Hout = g.Hout; pl->Wout = g.Wout; pl->Heff = g.Heff; pl->Weff = g.Weff; pl->M = g.M; pl->K = g.K;
  const int Cin = d->C1 + d->C2;
  const int hw_src = d->Hin * d->Win;
  pl->taps = d->upsample == 2 ? 4 : d->KH * d->KW;
  pl->cgroups = Cin / 32;
  pl->splitk = 1;
  pl->cg_per_split = pl->cgroups;
  // fp16-pair operands exist for NHWC tensors with 32-channel chunks; the gather is the fast form (tap validity mask + 32-bit byte
  // offsets); nearest-x2 only in its sub-pixel form
  pl->ok = d->in_layout == MF_LAYOUT_NHWC && d->out_layout == MF_LAYOUT_NHWC && d->C1 % 32 == 0 && d->C2 % 32 == 0 && d->Cout % 64 == 0 &&
           d->upsample != 1 && d->tile_hint >= 0 && (double)d->N * d->Hin * d->Win * (d->C1 > d->C2 ? d->C1 : d->C2) * 4.0 < 4294967040.0 &&
           (double)d->Cout * pl->K * 4.0 * (d->upsample == 2 ? 4 : d->upsample == 3 ? 16 : 1) < 4294967040.0;
  if (!pl->ok) return MF_OK;
  // Choice of (tile, split-K): the exact entry of the sweep table for the shapes of the published models (conv_plan_table.inc, generated
  // by scripts/conv_sweep.py --emit-table on MI355X), else a cost model fitted to the same sweeps (microseconds):
  //   launch + prologue + epilogue  ~7
  //   main loop                     ceil(workgroups / resident slots) * iterations * t_it(tile)
  //   split-K                       3 per level of the in-launch tree (power-of-two splits), else reducer 4 + (sk + 1) * output bytes / 3.5 TB/s
  const long out_bytes = (long)pl->M * d->Cout * 4;
  // (upsample == 3, the component GEMMs of the Winograd form: a tile lies inside one component = N / 16 pseudo-samples of hw_src rows)
  auto valid = [&](const Tile2& k) {
    return d->Cout % k.BN == 0 && (d->upsample != 2 || hw_src % k.BM == 0) && (d->upsample != 3 || ((long)(d->N / 16) * hw_src) % k.BM == 0) && halo_fits(d, k);
  };
  auto sk_ok = [&](int sk) { return sk >= 1 && sk <= pl->cgroups; };
  auto chain_ok = [&](int sk) { return (long)cdiv(pl->cgroups, sk) * pl->taps <= 96; };  // the matrix core adds with truncation: one chain <= 96 chunks
  const Tile2* c = nullptr;   // fixed by the hint or the table, else chosen by the model
  int sk = 0;                 // 0: not decided yet
  if (d->tile_hint > 0) {
    for (const auto& k : kTiles2) if (k.id == d->tile_hint) c = &k;
    MF_REQUIRE(c && d->Cout % c->BN == 0, MF_EINVAL, "conv(f16x2): bad tile_hint %d for Cout %d", d->tile_hint, d->Cout);
    if (d->upsample == 2 && hw_src % c->BM) { pl->ok = false; return MF_OK; }
    if (d->upsample == 3 && ((long)(d->N / 16) * hw_src) % c->BM) { pl->ok = false; return MF_OK; }
    if (!halo_fits(d, *c)) { pl->ok = false; return MF_OK; }
  } else {
    {
      std::lock_guard<std::mutex> lk(g_plan_mu);
      for (const auto& e : g_plan_overrides) {
        if (e.N == d->N && e.H == d->Hin && e.W == d->Win && e.Cin == Cin && e.Cout == d->Cout && e.k == d->KH && e.stride == d->stride && e.ups == d->upsample) {
          for (const auto& k : kTiles2) if (k.id == e.tile && valid(k)) c = &k;
          if (c && sk_ok(e.sk)) sk = e.sk;
          break;
        }
      }
    }
    if (c == nullptr) for (const auto& e : kPlanTable) {
      if (e.N == d->N && e.H == d->Hin && e.W == d->Win && e.Cin == Cin && e.Cout == d->Cout && e.k == d->KH && e.stride == d->stride &&
          e.ups == d->upsample) {
        for (const auto& k : kTiles2) if (k.id == e.tile && valid(k)) c = &k;
        if (c && sk_ok(e.sk)) sk = e.sk;
        break;
      }
    }
  }
  if (d->splitk_hint > 0) sk = d->splitk_hint;
  if (c == nullptr || sk == 0) {
    double best = 1e30;
    const Tile2* bc = nullptr;
    int bsk = 1;
    // Cost model (microseconds).  Round 6: re-fitted to EVERY sweep on file -- B = 8 / 12 / 16 / 24 / 32 / 69 / 200 at latent 32, B = 8 at latent 64,
    // the VAE decoder at 8 / 16 / 69 / 200 (scripts/plan_model_fit.py replays it on the CPU; profiles/r06_plan_model.txt: its pick is within
    // 0 - 2.8 % of the best of the sweep at every batch, the round-5 model was 5 - 21 % off at the batches its table does not hold):
    //   cost = 4.4 + W (iterations + 8) t_it(tile, dense?) + 2.2 per level of the in-launch split-K tree
    //   W    = ceil(workgroups / resident slots) for grids up to two rounds, else workgroups / slots + 0.24: a grid much larger than the chip does
    //          not run in lock-step rounds (a CU starts its next workgroup when one retires), the last partial round costs its fraction
    //   t_it = per-chunk time of the tile with ONE workgroup per CU (grid <= 256) / with the chip full (kTileCost: the 4-wave tiles run two per CU)
    // The component GEMMs of the Winograd form (upsample == 3: one tap, K loops of 8 - 64 chunks) take the same form with their own constants
    // (kTileCostGemm, 9 iterations of overhead, 3.2 per tree level): a regret-minimising search over the GEMM + tail sweeps at B = 8 ... 200
    // (profiles/r06_wino_tiles_b*.txt) -- within 0.1 - 2.5 % of the best of the search at every batch (the round-5 model: 0 - 6.7 %).
    struct TileCost { int id; float t_sparse, t_dense; };
    static const TileCost kTileCost[] = {{31, 1.248f, 1.332f}, {32, 1.306f, 1.341f}, {33, 0.642f, 0.677f}, {34, 0.621f, 0.667f}, {35, 0.647f, 0.762f},
                                         {36, 0.404f, 0.709f}, {37, 0.661f, 0.835f}, {51, 0.705f, 1.225f}, {52, 0.656f, 0.673f}, {53, 0.360f, 0.675f},
                                         {54, 0.377f, 0.686f}, {61, 1.189f, 1.215f}, {62, 1.192f, 1.234f}, {63, 0.680f, 0.685f}, {64, 0.660f, 0.691f}};
    static const TileCost kTileCostGemm[] = {{31, 1.160f, 1.052f}, {32, 1.306f, 1.341f}, {33, 0.596f, 0.677f}, {34, 0.621f, 0.667f}, {35, 0.647f, 0.762f},
                                             {36, 0.404f, 0.709f}, {37, 0.630f, 0.835f}, {51, 0.705f, 1.225f}, {52, 0.656f, 0.673f}, {53, 0.381f, 0.660f},
                                             {54, 0.377f, 0.686f}};
    const bool fitted = plan_model() == 1, gemm = d->upsample == 3;
    const double k_ovh = gemm ? 9.05 : 8.0, k_tree = gemm ? 3.2 : 2.2;
    for (const auto& k : kTiles2) {
      if (c != nullptr && k.id != c->id) continue;   // tile already fixed
      if (c == nullptr && k.id == 52) continue;       // (A/B form of 51, never chosen automatically)
      // halo tiles: the 256-row ones are part of the fitted model's choice (they win 1 - 8 % on the large 3x3 stride-1 shapes from B = 32 up);
      // the 128-row ones never won a sweep and stay for explicit hints; MF_CONV_HALO=1 lets the round-5 model consider all of them
      if (c == nullptr && k.HG && !(fitted ? (k.id == 61 || k.id == 62) : halo_auto())) continue;
      if (!valid(k)) continue;
      const long tiles = (long)cdiv(pl->M, k.BM) * (d->Cout / k.BN);
      const int percu = wgs_per_cu(k);
      for (int s = 1; s <= 32; s *= 2) {
        if (sk != 0 && s != sk) continue;               // split-K already fixed
        if (!sk_ok(s)) break;
        if (sk == 0 && !chain_ok(s) && sk_ok(s * 2) && s < 32) continue;
        const long wgs = tiles * s;
        const double its = (double)cdiv(pl->cgroups, s) * pl->taps;
        double cost;
        if (fitted) {
          const double slots = 256.0 * percu;
          double t_it = 0.68;
          if (gemm) { for (const auto& tc : kTileCostGemm) if (tc.id == k.id) t_it = wgs > 256 ? tc.t_dense : tc.t_sparse; }
          else { for (const auto& tc : kTileCost) if (tc.id == k.id) t_it = wgs > 256 ? tc.t_dense : tc.t_sparse; }
          const double W = wgs > 2 * slots ? wgs / slots + 0.24 : (double)((wgs + (long)slots - 1) / (long)slots);
          cost = 4.4 + W * (its + k_ovh) * t_it;
          if (s > 1 && !(s & (s - 1))) { int lv = 0; while ((1 << lv) < s) ++lv; cost += k_tree * lv; }
          else if (s > 1) cost += 4.0 + (s + 1) * (double)out_bytes / 3.5e6;
        } else {
          const long waves = (wgs + 256L * percu - 1) / (256L * percu);
          double t_it = 0.68 * k.BM * k.BN / (128.0 * 128.0);                 // 8-wave tiles, one workgroup per CU
          if (k.BM * k.BN <= 128 * 64) t_it = (wgs > 256 ? 0.80 : 0.43);      // half-size tiles: two per CU when the grid is that large
          else if (percu == 2) t_it = (wgs > 256 ? 1.40 : 0.75);              // 4-wave 128 x 128
          cost = 7.0 + waves * (its + 3.0) * t_it;
          if (s > 1 && !(s & (s - 1))) { int lv = 0; while ((1 << lv) < s) ++lv; cost += 3.0 * lv; }   // the slices meet inside the launch
          else if (s > 1) cost += 4.0 + (s + 1) * (double)out_bytes / 3.5e6;                          // slabs + reducer pass
        }
        if (cost < best) { best = cost; bc = &k; bsk = s; }
      }
    }
    if (bc == nullptr) {
      if (c == nullptr) { pl->ok = false; return MF_OK; }
      bc = c;
      bsk = sk ? sk : 1;
    }
    c = bc;
    sk = bsk;
  }
  pl->t = *c;
  if (sk > pl->cgroups) sk = pl->cgroups;
  pl->cg_per_split = cdiv(pl->cgroups, sk);
  pl->splitk = cdiv(pl->cgroups, pl->cg_per_split);
  return MF_OK;
}

inline bool pair_precision(int prec) { return prec == MF_CONV_FP32_F16X2 || prec == MF_CONV_F16; }   // both run on the fp16-pair operands

// How many workgroups of one kernel instantiation are resident on the current device at once (CUs x occupancy; 0: unknown -- no device).
// The fused GroupNorm tail (conv_f16x2.h: FuseP) makes workgroups WAIT for the other tiles of their sample, so the host only asks for it
// when the whole grid is resident from the start.
int resident_workgroups(const void* fn, int threads, size_t lds) {
  int dev = 0, cus = 0, per = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, fn, threads, lds) != hipSuccess || per <= 0) { (void)hipGetLastError(); return 0; }
  return cus * per;
}

// mode 0: launch; mode 1: return resident_workgroups() of the instantiation (no launch)
template <int BM, int BN, int WM, int WN, int NST, int TERMS>
int launch_f16x2_t(const mfc2::ConvP2& p, hipStream_t s, int mode) {
  constexpr size_t lds = (size_t)NST * (BM + BN) * 128u;
  static DeviceOnce once;   // per device: the attribute belongs to the (function, device) pair
  const void* fn = reinterpret_cast<const void*>(&mfc2::conv_f16x2_kernel<BM, BN, WM, WN, NST, TERMS>);
  if (first_use_on_device(once)) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (mode == 1) return resident_workgroups(fn, WM * WN * 64, lds);
  const int grid = p.tiles_m * p.tiles_n * p.splitk;
  MF_LAUNCH((mfc2::conv_f16x2_kernel<BM, BN, WM, WN, NST, TERMS>), dim3(grid), dim3(WM * WN * 64), lds, s, p);
  return check_launch("conv_f16x2");
}
template <int BM, int BN, int WM, int WN, int NST>
int launch_f16x2(const mfc2::ConvP2& p, hipStream_t s, int terms, int mode = 0) {
  return terms == 1 ? launch_f16x2_t<BM, BN, WM, WN, NST, 1>(p, s, mode) : launch_f16x2_t<BM, BN, WM, WN, NST, 3>(p, s, mode);
}
template <int BM, int BN, int WM, int WN, int HG, int TERMS>
int launch_halo_t(const mfc2::ConvP2& p, hipStream_t s, int mode) {
  constexpr size_t lds = (size_t)2 * HG * (WM * WN) * 1024 + (size_t)3 * BN * 128;
  static DeviceOnce once;
  const void* fn = reinterpret_cast<const void*>(&mfc2::conv_halo_kernel<BM, BN, WM, WN, HG, TERMS>);
  if (first_use_on_device(once)) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (mode == 1) return resident_workgroups(fn, WM * WN * 64, lds);
  const int grid = p.tiles_m * p.tiles_n * p.splitk;
  MF_LAUNCH((mfc2::conv_halo_kernel<BM, BN, WM, WN, HG, TERMS>), dim3(grid), dim3(WM * WN * 64), lds, s, p);
  return check_launch("conv_halo");
}
template <int BM, int BN, int WM, int WN, int HG>
int launch_halo(const mfc2::ConvP2& p, hipStream_t s, int terms, int mode = 0) {
  return terms == 1 ? launch_halo_t<BM, BN, WM, WN, HG, 1>(p, s, mode) : launch_halo_t<BM, BN, WM, WN, HG, 3>(p, s, mode);
}
// launch (mode 0) or residency query (mode 1) of tile `id`; -1: no such tile
int dispatch_tile(int id, const mfc2::ConvP2& p, hipStream_t s, int terms, int mode) {
  switch (id) {
    case 31: return launch_f16x2<128, 256, 2, 4, 3>(p, s, terms, mode);
    case 32: return launch_f16x2<256, 128, 4, 2, 3>(p, s, terms, mode);
    case 33: return launch_f16x2<128, 128, 2, 4, 3>(p, s, terms, mode);
    case 34: return launch_f16x2<128, 128, 4, 2, 3>(p, s, terms, mode);
    case 35: return launch_f16x2<256, 64, 4, 2, 3>(p, s, terms, mode);
    case 36: return launch_f16x2<128, 64, 4, 2, 3>(p, s, terms, mode);
    case 37: return launch_f16x2<64, 256, 1, 8, 3>(p, s, terms, mode);
    case 51: return launch_f16x2<128, 128, 2, 2, 2>(p, s, terms, mode);
    case 52: return launch_f16x2<128, 128, 2, 2, 3>(p, s, terms, mode);
    case 53: return launch_f16x2<64, 128, 2, 2, 3>(p, s, terms, mode);
    case 54: return launch_f16x2<128, 64, 2, 2, 3>(p, s, terms, mode);
    case 61: return launch_halo<256, 128, 4, 2, 6>(p, s, terms, mode);
    case 62: return launch_halo<256, 128, 4, 2, 7>(p, s, terms, mode);
    case 63: return launch_halo<128, 128, 2, 4, 4>(p, s, terms, mode);
    case 64: return launch_halo<128, 128, 2, 4, 5>(p, s, terms, mode);
    default: return -1;
  }
}

// ---- two convolutions in one launch (conv_f16x2_group.h): the tile pairs the plans of cfg2's channel-changing ResBlocks need (3x3 host
// tile, 1x1 guest tile), three-term arithmetic only.  mode 0: launch; mode 2: is the pair instantiated (1 / -1)
template <class A, class B>
int launch_group(const mfc2::ConvP2& pa, const mfc2::ConvP2& pb, hipStream_t s, int mode) {
  if (mode == 2) return 1;
  constexpr size_t lds = A::lds > B::lds ? A::lds : B::lds;
  static DeviceOnce once;
  const void* fn = reinterpret_cast<const void*>(&mfc2::conv_group_kernel<A, B, 3>);
  if (first_use_on_device(once)) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int na = pa.tiles_m * pa.tiles_n * pa.splitk, nb = pb.tiles_m * pb.tiles_n * pb.splitk;
  MF_LAUNCH((mfc2::conv_group_kernel<A, B, 3>), dim3(na + nb), dim3(A::threads), lds, s, pa, pb, na);
  return check_launch("conv_f16x2_group");
}
int dispatch_group(int ida, int idb, const mfc2::ConvP2& pa, const mfc2::ConvP2& pb, hipStream_t s, int mode) {
  using T31 = mfc2::PlainTile<128, 256, 2, 4, 3>;
  using T33 = mfc2::PlainTile<128, 128, 2, 4, 3>;
  using T34 = mfc2::PlainTile<128, 128, 4, 2, 3>;
  using T36 = mfc2::PlainTile<128, 64, 4, 2, 3>;
  using T37 = mfc2::PlainTile<64, 256, 1, 8, 3>;
  using T53 = mfc2::PlainTile<64, 128, 2, 2, 3>;
  using T54 = mfc2::PlainTile<128, 64, 2, 2, 3>;
  using T62 = mfc2::HaloTile<256, 128, 4, 2, 7>;
  switch (ida * 100 + idb) {
    case 5353: return launch_group<T53, T53>(pa, pb, s, mode);
    case 5453: return launch_group<T54, T53>(pa, pb, s, mode);
    case 3436: return launch_group<T34, T36>(pa, pb, s, mode);
    case 3437: return launch_group<T34, T37>(pa, pb, s, mode);
    case 6236: return launch_group<T62, T36>(pa, pb, s, mode);
    case 6237: return launch_group<T62, T37>(pa, pb, s, mode);
    // the component GEMM of a Winograd convolution as the host (round 5: conv_res in ITS grid -- mf_conv2d_wino_gn_apply_f16x2(..., guest))
    case 3336: return launch_group<T33, T36>(pa, pb, s, mode);
    case 3337: return launch_group<T33, T37>(pa, pb, s, mode);
    case 3136: return launch_group<T31, T36>(pa, pb, s, mode);
    case 3137: return launch_group<T31, T37>(pa, pb, s, mode);
    default: return -1;
  }
}

// split-K met inside the launch (conv_f16x2.h: ConvP2::tree) instead of slabs + reducer pass: a power-of-two split whose hand-off region
// stays addressable with 32-bit offsets.  MF_CONV_TREE=0 keeps the reducer (A/B).
// MF_CONV_TREE: 0 = slabs + reducer pass, 1 = in-launch hand-off through sc1 stores / sc1 loads, 2 = the same with release / acquire
// fences around the pair counter
int tree_mode() {
  static const int env = [] { const char* e = getenv("MF_CONV_TREE"); const int v = e ? atoi(e) : 1; return v < 0 || v > 2 ? 1 : v; }();
  return env;
}
bool tree_possible(const MfConvDesc* d, const Plan2& pl) {
  if (!tree_mode() || !pl.ok || pl.splitk < 2 || (pl.splitk & (pl.splitk - 1))) return false;
  return 2.0 * (pl.splitk - 1) * pl.t.BM * pl.t.BN * 4.0 < 4294967040.0;
}
size_t tree_handoff_bytes(const MfConvDesc* d, const Plan2& pl) {
  return (size_t)cdiv(pl.M, pl.t.BM) * (d->Cout / pl.t.BN) * 2u * (pl.splitk - 1) * pl.t.BM * pl.t.BN * sizeof(float);
}
// can the epilogue of the (last) workgroup of a tile emit the GroupNorm records: the tile lies inside one sample and holds whole groups
// a tile lies inside one sample, or holds whole samples each made of whole wave rows (small images): the epilogue can tell samples apart
bool tile_sample_aligned(const Plan2& pl) {
  const int HW = pl.Hout * pl.Wout, FM = pl.t.BM / pl.t.WM;
  return HW % pl.t.BM == 0 || (pl.t.BM % HW == 0 && HW % FM == 0);
}
bool epilogue_stats_ok(const MfConvDesc* d, const Plan2& pl, int G) {
  const int cpg = d->Cout / G;
  return tile_sample_aligned(pl) && pl.t.BN % cpg == 0 && cpg % 8 == 0;
}

// partial GroupNorm records the f16x2 convolution (or its split-K reducer) emits; 0: it cannot
int gn_parts2(const MfConvDesc* d, const Plan2& pl, int G) {
  if (!pl.ok || G <= 0 || d->Cout % G) return 0;
  const int HW = pl.Hout * pl.Wout;
  if (pl.splitk > 1 && !(tree_possible(d, pl) && epilogue_stats_ok(d, pl, G)))
    return stats_lds_bytes(d->Cout / stats_slices(d->N, HW, d->Cout, G)) <= 64 * 1024 ? stats_chunks(HW) : 0;
  if (!epilogue_stats_ok(d, pl, G)) return 0;
  return HW >= pl.t.BM ? HW / pl.t.BM : 1;
}


}  // namespace

namespace mf {
int f16x2_gn_parts(const MfConvDesc* d, int G) {
  Plan2 p2;
  return make_plan2(d, &p2) == MF_OK ? gn_parts2(d, p2, G) : 0;
}
size_t f16x2_workspace_bytes(const MfConvDesc* d) {
  Plan2 p2;
  if (make_plan2(d, &p2) != MF_OK || !p2.ok || p2.splitk <= 1) return 0;
  const size_t slabs = (size_t)p2.splitk * p2.M * d->Cout * sizeof(float);   // (a launch that needs reducer-side statistics still takes this path)
  const size_t tree = tree_possible(d, p2) ? tree_handoff_bytes(d, p2) : 0;
  return slabs > tree ? slabs : tree;
}
}  // namespace mf

#if MFC2_HZ & (256 | 512)
static float* g_conv_dbg = nullptr;
extern "C" void mf_debug_set_conv_dump(float* p) { g_conv_dbg = p; }   // diagnostic builds only (scripts/pk_dump.py)
#endif

extern "C" {

/* ------------------------------------------------------------------ MF_CONV_FP32_F16X2 */
int mf_conv2d_f16x2_ok(const MfConvDesc* d) {
  Plan2 pl;
  return d && pair_precision(d->precision) && make_plan2(d, &pl) == MF_OK && pl.ok ? 1 : 0;
}

// slots of the measured-bound array a conv writes per sample (0: it cannot measure -- a tile straddles two samples)
static int bound_slots2(const MfConvDesc* d, const Plan2& pl, bool with_stats) {
  const int HW = pl.Hout * pl.Wout;
  if (pl.splitk == 1 || tree_possible(d, pl)) {   // per (tile inside the sample, n-tile, wave inside the sample)
    if (!tile_sample_aligned(pl)) return 0;
    const int nw = pl.t.WM * pl.t.WN;
    return HW >= pl.t.BM ? (HW / pl.t.BM) * (d->Cout / pl.t.BN) * nw : (d->Cout / pl.t.BN) * (nw * HW / pl.t.BM);
  }
  if (with_stats) return 0;   // (every convolution followed by a GroupNorm is bounded by the normalisation, not by measurement)
  const long p4 = (long)HW * d->Cout / 4;
  int bx = (int)((p4 + 255) / 256);
  const int cap = cdiv(2048, d->N);
  if (bx > cap) bx = cap;
  return bx * 4;
}

int mf_conv2d_f16x2_bound_slots(const MfConvDesc* d) {
  Plan2 pl;
  if (!d || !pair_precision(d->precision) || make_plan2(d, &pl) != MF_OK || !pl.ok) return 0;
  return bound_slots2(d, pl, false);
}

int mf_conv2d_plan_query(const MfConvDesc* d, int32_t* tile_id, int32_t* splitk) {
  MF_REQUIRE(d, MF_EINVAL, "plan_query: null desc");
  if (pair_precision(d->precision)) {
    Plan2 pl;
    int rc = make_plan2(d, &pl);
    if (rc) return rc;
    if (tile_id) *tile_id = pl.ok ? pl.t.id : 0;
    if (splitk) *splitk = pl.ok ? pl.splitk : 0;
    return MF_OK;
  }
  return igemm_plan_query(d, tile_id, splitk);
}

int mf_split_f16x2(const float* x, void* xs, const float* bound, int rows, int64_t per_row, void* stream) {
  MF_REQUIRE(x && xs && rows > 0 && per_row > 0 && per_row % 8 == 0, MF_EINVAL, "split_f16x2: bad args (per_row %% 8 == 0)");
  const long octets = (long)rows * (per_row / 8);
  ProfScope ps(MF_FAM_MISC, (hipStream_t)stream, 0, 8.0 * rows * (double)per_row);
  const int blocks = (int)((octets + 255) / 256 > 8192 ? 8192 : (octets + 255) / 256);
  MF_LAUNCH(mfc2::split_act_f16x2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, reinterpret_cast<mfc2::u32x4*>(xs), octets, bound,
                     (long)(per_row / 8));
  return check_launch("split_f16x2");
}

int mf_split_f16x2_slots(const float* x, void* xs, const float* slots, int nslots, float* bound_out, int rows, int64_t per_row, void* stream) {
  MF_REQUIRE(x && xs && slots && bound_out && nslots > 0 && rows > 0 && rows <= 65535 && per_row > 0 && per_row % 8 == 0, MF_EINVAL,
             "split_f16x2_slots: bad args (per_row %% 8 == 0, rows <= 65535)");
  const long opr = per_row / 8;
  ProfScope ps(MF_FAM_MISC, (hipStream_t)stream, 0, 8.0 * rows * (double)per_row);
  long bpr = (opr + 255) / 256;
  const long cap = (8192 + rows - 1) / rows;
  if (bpr > cap) bpr = cap;
  MF_LAUNCH(mfc2::split_act_slots_kernel, dim3((int)bpr, rows), dim3(256), 0, (hipStream_t)stream, x, reinterpret_cast<mfc2::u32x4*>(xs), opr, slots, nslots,
            bound_out);
  return check_launch("split_f16x2_slots");
}

static int host_scale_exp(float bound) {  // the host-side twin of scale_exp_of (split_f16.h)
  if (!(bound > 0.f)) return 0;
  uint32_t u;
  memcpy(&u, &bound, 4);
  const int s = (int)((u >> 23) & 0xffu) - 127 - 14;
  return s < -100 ? -100 : (s > 100 ? 100 : s);
}

int mf_conv2d_plan_override(const MfConvDesc* d, int tile, int splitk) {
  MF_REQUIRE(d && pair_precision(d->precision), MF_EINVAL, "plan_override: a MF_CONV_FP32_F16X2 / MF_CONV_F16 descriptor");
  std::lock_guard<std::mutex> lk(g_plan_mu);
  const int Cin = d->C1 + d->C2;
  for (size_t i = 0; i < g_plan_overrides.size(); ++i) {
    const PlanEntry& e = g_plan_overrides[i];
    if (e.N == d->N && e.H == d->Hin && e.W == d->Win && e.Cin == Cin && e.Cout == d->Cout && e.k == d->KH && e.stride == d->stride && e.ups == d->upsample) {
      g_plan_overrides.erase(g_plan_overrides.begin() + i);
      break;
    }
  }
  if (tile > 0 && splitk > 0) g_plan_overrides.push_back(PlanEntry{d->N, d->Hin, d->Win, Cin, d->Cout, d->KH, d->stride, d->upsample, tile, splitk});
  return MF_OK;
}

int mf_conv2d_f16x2_sync_words(const MfConvDesc* d) {
  Plan2 pl;
  if (!d || !pair_precision(d->precision) || make_plan2(d, &pl) != MF_OK || !pl.ok || !tree_possible(d, pl)) return 0;
  return cdiv(pl.M, pl.t.BM) * (d->Cout / pl.t.BN) * (pl.splitk - 1);
}

// can this convolution apply the GroupNorm that follows it inside its own launch (conv_f16x2.h: FuseP): statistics from the epilogue of
// the workgroup that holds a tile's final values (no reducer pass), a tile that tells samples apart, and EVERY workgroup of the launch
// resident at once.  -> tiles a sample's counter waits for, 0: no.  (Capability only: whether a caller USES the form is its policy -- the
// Python host keeps it opt-in, MEDFUSION_FUSED_APPLY=1: measured not faster than the two launches, profiles/r04_fused_gn_apply_ab.txt.)
static int fuse_tiles_per_sample(const MfConvDesc* d, const Plan2& pl, int G) {
  if (!pl.ok || G <= 0 || G > 256 || d->Cout % G || d->upsample != 0) return 0;
  if (pl.splitk > 1 && !tree_possible(d, pl)) return 0;
  if (!epilogue_stats_ok(d, pl, G)) return 0;
  const int HW = pl.Hout * pl.Wout;
  if (HW < pl.t.BM && pl.t.BM / HW > 8) return 0;                          // (at most 8 whole samples per tile: the tail's LDS scratch)
  const long grid = (long)cdiv(pl.M, pl.t.BM) * (d->Cout / pl.t.BN) * pl.splitk;
  mfc2::ConvP2 dummy{};
  const int resident = dispatch_tile(pl.t.id, dummy, nullptr, d->precision == MF_CONV_F16 ? 1 : 3, 1);
  if (resident <= 0 || grid > resident) return 0;
  // MEDFUSION_FUSE_FAULT=1 (tests only): one arrival too many is awaited, so every rendezvous times out -- exercises the error flag and the
  // host's re-run on the two-launch form
  static const int fault = [] { const char* e = getenv("MEDFUSION_FUSE_FAULT"); return e ? atoi(e) : 0; }();
  return (HW >= pl.t.BM ? HW / pl.t.BM : 1) * (d->Cout / pl.t.BN) + (fault ? 1 : 0);
}

int mf_conv2d_f16x2_fuse_words(const MfConvDesc* d, int G) {
  Plan2 pl;
  if (!d || !pair_precision(d->precision) || make_plan2(d, &pl) != MF_OK || !pl.ok) return 0;
  return fuse_tiles_per_sample(d, pl, G) > 0 ? 2 * d->N : 0;
}

struct PairsOut { void* y_split; float* y_bound; float wl1_1, wl1_2, bmax; };
static int conv_f16x2_impl(const void* x1s, const void* x2s, const void* ws, const float* bias, float* y, const float* x1_bound, const float* x2_bound,
                           float w_bound, float* y_bound, void* workspace, size_t workspace_bytes, uint32_t* sync, double* gn_partial, int G,
                           const MfGnFuse* fz, const MfConvDesc* d, void* stream, const PairsOut* po = nullptr);

// can the epilogue of this plan write the output's fp16-pair form itself (final values in the launch: no reducer pass behind it)?
int mf_conv2d_f16x2_pairs_out_ok(const MfConvDesc* d) {
  Plan2 pl;
  if (!d || !pair_precision(d->precision) || make_plan2(d, &pl) != MF_OK || !pl.ok || d->Cout % 8) return 0;
  return (pl.splitk == 1 || tree_possible(d, pl)) ? 1 : 0;
}

int mf_conv2d_f16x2_pairs_out(const void* x1s, const void* x2s, const void* ws, const float* bias, float* y, void* y_split, float* y_bound_out,
                              const float* x1_bound, const float* x2_bound, float w_bound, float w_l1_1, float w_l1_2, float bias_max, void* workspace,
                              size_t workspace_bytes, uint32_t* sync, const MfConvDesc* d, void* stream) {
  MF_REQUIRE(y && y_split && y_bound_out && x1_bound, MF_EINVAL, "conv(f16x2, pairs out): y, y_split, y_bound_out and x1_bound are required");
  MF_REQUIRE(d && (d->C2 == 0 || x2_bound), MF_EINVAL, "conv(f16x2, pairs out): x2_bound with a second source");
  MF_REQUIRE(mf_conv2d_f16x2_pairs_out_ok(d), MF_EUNSUPPORTED, "conv(f16x2, pairs out): this plan reduces split-K behind the launch (mf_conv2d_f16x2_pairs_out_ok == 0)");
  MF_REQUIRE(w_l1_1 >= 0.f && w_l1_2 >= 0.f && bias_max >= 0.f, MF_EINVAL, "conv(f16x2, pairs out): norms must be >= 0");
  const PairsOut po{y_split, y_bound_out, w_l1_1, w_l1_2, bias_max};
  return conv_f16x2_impl(x1s, x2s, ws, bias, y, x1_bound, x2_bound, w_bound, nullptr, workspace, workspace_bytes, sync, nullptr, 0, nullptr, d, stream, &po);
}

int mf_conv2d_f16x2(const void* x1s, const void* x2s, const void* ws, const float* bias, float* y, const float* x1_bound, const float* x2_bound,
                    float w_bound, float* y_bound, void* workspace, size_t workspace_bytes, uint32_t* sync, double* gn_partial, int G,
                    const MfConvDesc* d, void* stream) {
  MF_REQUIRE(y, MF_EINVAL, "conv(f16x2): null pointer");
  return conv_f16x2_impl(x1s, x2s, ws, bias, y, x1_bound, x2_bound, w_bound, y_bound, workspace, workspace_bytes, sync, gn_partial, G, nullptr, d, stream);
}

int mf_conv2d_f16x2_gn_apply(const void* x1s, const void* x2s, const void* ws, const float* bias, const float* x1_bound, const float* x2_bound,
                             float w_bound, void* workspace, size_t workspace_bytes, uint32_t* sync, double* gn_partial, int G, const MfGnFuse* f,
                             const MfConvDesc* d, void* stream) {
  MF_REQUIRE(f && gn_partial && G > 0, MF_EINVAL, "conv_gn_apply: needs the fuse arguments, the record array and G");
  MF_REQUIRE(f->out_split && f->out_bound && f->rendezvous && f->error_flag, MF_EINVAL, "conv_gn_apply: out_split, out_bound, rendezvous and error_flag are required");
  MF_REQUIRE((f->gamma == nullptr) == (f->beta == nullptr), MF_EINVAL, "conv_gn_apply: gamma/beta must both be given or both NULL");
  MF_REQUIRE(!(f->residual && f->residual_pairs), MF_EINVAL, "conv_gn_apply: the residual is fp32 OR fp16 pairs");
  MF_REQUIRE(!f->residual_pairs || f->res_bound, MF_EINVAL, "conv_gn_apply: a residual given as fp16 pairs needs the res_bound that scaled it");
  MF_REQUIRE(!f->residual || f->res_bound || (f->res_bound_slots && f->res_nslots > 0), MF_EINVAL, "conv_gn_apply: a residual needs res_bound or res_bound_slots");
  MF_REQUIRE(!f->emb || (f->emb_bound && f->emb_stride % 4 == 0), MF_EINVAL, "conv_gn_apply: an embedding needs emb_bound and emb_stride %% 4 == 0");
  return conv_f16x2_impl(x1s, x2s, ws, bias, nullptr, x1_bound, x2_bound, w_bound, nullptr, workspace, workspace_bytes, sync, gn_partial, G, f, d, stream);
}

// everything of a launch but the launch: checks, plan, kernel parameters.  (mf_conv2d_f16x2_group prepares two and launches once.)
struct Prep { mfc2::ConvP2 p; Plan2 pl; bool tree; double flops, bytes; int terms; };
static int conv_f16x2_prepare(const void* x1s, const void* x2s, const void* ws, const float* bias, float* y, const float* x1_bound, const float* x2_bound,
                              float w_bound, float* y_bound, void* workspace, size_t workspace_bytes, uint32_t* sync, double* gn_partial, int G,
                              const MfGnFuse* fz, const MfConvDesc* d, const PairsOut* po, Prep* out, bool wino_gemm = false) {
  MF_REQUIRE(d && pair_precision(d->precision), MF_EINVAL, "conv(f16x2): desc.precision must be MF_CONV_FP32_F16X2 (or the opt-in MF_CONV_F16)");
  MF_REQUIRE((d->upsample == 3) == wino_gemm, MF_EINVAL, "conv(f16x2): upsample = 3 is internal to mf_conv2d_wino_f16x2");
  Plan2& pl = out->pl;
  mfc2::ConvP2& p = out->p;
  int rc = make_plan2(d, &pl);
  if (rc) return rc;
  MF_REQUIRE(pl.ok, MF_EUNSUPPORTED, "conv(f16x2): this shape/layout is not on the fp16-pair path (ask mf_conv2d_f16x2_ok)");
  MF_REQUIRE(x1s && ws && (y || fz), MF_EINVAL, "conv(f16x2): null pointer");
  MF_REQUIRE(d->C2 == 0 || x2s != nullptr, MF_EINVAL, "conv(f16x2): C2 > 0 but x2 is null");
  MF_REQUIRE(!gn_partial || gn_parts2(d, pl, G) > 0, MF_EUNSUPPORTED, "conv(f16x2): cannot emit GroupNorm partials (mf_conv2d_gn_parts == 0)");
  MF_REQUIRE(!y_bound || (!gn_partial && bound_slots2(d, pl, false) > 0), MF_EUNSUPPORTED,
             "conv(f16x2): cannot measure the output bound of this plan (mf_conv2d_f16x2_bound_slots == 0, or GroupNorm statistics requested)");
  p.x1 = x1s; p.x2 = x2s; p.w = ws; p.bias = bias; p.y = y;
  p.bound1 = x1_bound; p.bound2 = x2_bound; p.wexp = host_scale_exp(w_bound); p.out_bound = y_bound; p.bound_slots = y_bound ? bound_slots2(d, pl, false) : 0;
  p.N = d->N; p.Hin = d->Hin; p.Win = d->Win; p.C1 = d->C1; p.C2 = d->C2; p.Cin = d->C1 + d->C2; p.Cout = d->Cout;
  p.Hout = pl.Hout; p.Wout = pl.Wout; p.Heff = pl.Heff; p.Weff = pl.Weff;
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad;
  p.subpix = d->upsample == 2 ? 1 : 0; p.hw_src = d->Hin * d->Win;
  if (p.subpix) { p.KH = p.KW = 2; p.Heff = d->Hin; p.Weff = d->Win; }
  p.M = pl.M; p.K = pl.K; p.HWout = pl.Hout * pl.Wout;
  p.cgroups = pl.cgroups; p.cg_per_split = pl.cg_per_split; p.splitk = pl.splitk;
  p.tiles_m = cdiv(pl.M, pl.t.BM); p.tiles_n = d->Cout / pl.t.BN;
  p.slab = (long)pl.M * d->Cout;
  p.bytes1 = (unsigned)(4.0 * d->N * d->Hin * d->Win * d->C1);
  p.bytes2 = (unsigned)(4.0 * d->N * d->Hin * d->Win * d->C2);
  p.bytesw = (unsigned)(4.0 * d->Cout * pl.K * (p.subpix ? 4 : wino_gemm ? 16 : 1));
  p.wphase_rows = wino_gemm ? (d->N / 16) * d->Hin * d->Win : 0;
  {
    // tile walk (round 6): output-channel tiles fastest where the activations outweigh the weights -- every XCD then covers all Cout blocks of ITS
    // pixel range and the activation tensor is fetched once instead of tiles_n times (the weights are then fetched by every XCD: the smaller operand).
    // MF_CONV_WALK: 0 the round-5 pixel-fastest walk everywhere (A/B), 1 (default) by operand size, 2 output-channel-fastest everywhere.
    static const int walk = [] { const char* e = getenv("MF_CONV_WALK"); return e ? atoi(e) : 1; }();
    const double act_bytes = 4.0 * d->N * d->Hin * d->Win * (d->C1 + d->C2), w_bytes = 4.0 * d->Cout * pl.K * (p.subpix ? 4 : 1);
    p.walk_n_fast = (!wino_gemm && p.tiles_n > 1 && (walk == 2 || (walk == 1 && act_bytes > w_bytes))) ? 1 : 0;
  }
  p.out_nt = 0;   // (non-temporal stores for the component GEMM's output: 397.54 vs 397.53 ms on the cfg2 step -- nothing; the hook stays for A/B builds)
  p.gn_partial = nullptr; p.gn_groups = G; p.gn_cpg = G > 0 ? d->Cout / G : 8; p.gn_parts = 0;
  p.tree = 0; p.handoff = nullptr; p.sync = nullptr;
#if MFC2_HZ & (256 | 512)
  p.dbg = g_conv_dbg;
#endif
  p.y_pairs = po ? po->y_split : nullptr; p.y_pair_bound = po ? po->y_bound : nullptr;
  p.wl1_1 = po ? po->wl1_1 : 0.f; p.wl1_2 = po ? po->wl1_2 : 0.f; p.bmax = po ? po->bmax : 0.f;
  memset(&p.fz, 0, sizeof(p.fz));
  if (fz) {
    const int tps = fuse_tiles_per_sample(d, pl, G);
    MF_REQUIRE(tps > 0, MF_EUNSUPPORTED, "conv_gn_apply: this plan cannot apply its GroupNorm inside the launch (mf_conv2d_f16x2_fuse_words == 0)");
    p.fz.on = 1; p.fz.act = fz->act; p.fz.eps = fz->eps; p.fz.bconst = fz->bconst; p.fz.count = (double)(pl.Hout * pl.Wout) * (d->Cout / G);
    p.fz.gamma = fz->gamma; p.fz.beta = fz->beta; p.fz.res_f32 = fz->residual; p.fz.res_pairs = fz->residual_pairs;
    p.fz.res_bound = fz->res_bound; p.fz.res_slots = fz->res_bound ? nullptr : fz->res_bound_slots; p.fz.res_nslots = fz->res_nslots;
    p.fz.tiles_per_sample = tps; p.fz.emb = fz->emb; p.fz.emb_stride = (long)fz->emb_stride; p.fz.emb_bound = fz->emb_bound;
    p.fz.out_f32 = fz->out; p.fz.out_pairs = fz->out_split; p.fz.out_bound = fz->out_bound; p.fz.rv = fz->rendezvous; p.fz.rv_err = fz->error_flag;
  }
  const bool tree = pl.splitk > 1 && tree_possible(d, pl) && (!gn_partial || epilogue_stats_ok(d, pl, G));
  if ((pl.splitk == 1 || tree) && gn_partial) { p.gn_partial = gn_partial; p.gn_parts = p.HWout >= pl.t.BM ? p.HWout / pl.t.BM : 1; }
  if (tree) {
    const size_t need = tree_handoff_bytes(d, pl);
    MF_REQUIRE(workspace && workspace_bytes >= need, MF_EWORKSPACE, "conv(f16x2): workspace %zu < %zu", workspace_bytes, need);
    MF_REQUIRE(sync, MF_EINVAL, "conv(f16x2): this plan meets its split-K slices inside the launch and needs `sync` "
                                "(mf_conv2d_f16x2_sync_words(d) zero-initialised words the caller keeps between launches)");
    p.tree = tree_mode(); p.handoff = reinterpret_cast<float*>(workspace); p.sync = sync;
    if (y_bound && p.bound_slots == 0) p.out_bound = nullptr;
  } else if (pl.splitk > 1) {
    const size_t need = (size_t)pl.splitk * pl.M * d->Cout * sizeof(float);
    MF_REQUIRE(workspace && workspace_bytes >= need, MF_EWORKSPACE, "conv(f16x2): workspace %zu < %zu", workspace_bytes, need);
    p.y = reinterpret_cast<float*>(workspace);
    p.out_bound = nullptr;   // the reducer measures
  }
  out->tree = tree;
  out->flops = 2.0 * pl.M * (double)d->Cout * (d->upsample == 2 ? 9.0 * (d->C1 + d->C2) : (double)pl.K);
  out->bytes = 4.0 * ((double)d->N * d->Hin * d->Win * p.Cin + (double)d->Cout * pl.K + (double)pl.M * d->Cout);
  out->terms = d->precision == MF_CONV_F16 ? 1 : 3;
  return MF_OK;
}

static int conv_f16x2_impl(const void* x1s, const void* x2s, const void* ws, const float* bias, float* y, const float* x1_bound, const float* x2_bound,
                           float w_bound, float* y_bound, void* workspace, size_t workspace_bytes, uint32_t* sync, double* gn_partial, int G,
                           const MfGnFuse* fz, const MfConvDesc* d, void* stream, const PairsOut* po) {
  Prep q;
  int rc = conv_f16x2_prepare(x1s, x2s, ws, bias, y, x1_bound, x2_bound, w_bound, y_bound, workspace, workspace_bytes, sync, gn_partial, G, fz, d, po, &q);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  const Plan2& pl = q.pl;
  const mfc2::ConvP2& p = q.p;
  const bool tree = q.tree;
  {
    const int terms = q.terms;
    ProfScope ps(fz ? MF_FAM_CONV_GN_FUSED : MF_FAM_CONV_IGEMM, s, q.flops, q.bytes, 2.0 * pl.M * (double)d->Cout * pl.K * terms);
    ps.set_tag(pl.t.id + (terms == 1 ? 100000 : 0));
    rc = dispatch_tile(pl.t.id, p, s, terms, 0);
    if (rc == -1) { set_error("conv(f16x2): no tile config %d", pl.t.id); rc = MF_EINVAL; }
  }
  if (rc) return rc;
  if (pl.splitk > 1 && !tree) {
    const int HW = p.HWout;
    ProfScope ps(MF_FAM_SPLITK_REDUCE, s, 0, 4.0 * pl.M * d->Cout * (pl.splitk + 1));
    if (gn_partial) {  // reduction + bias + GroupNorm partial statistics (+ measured bound) in one streaming pass
      const int slices = stats_slices(d->N, HW, d->Cout, G), chunks = stats_chunks(HW);
      MF_LAUNCH(gn_partial_kernel<true>, dim3(chunks, d->N, slices), dim3(kStatsThreads), stats_lds_bytes(d->Cout / slices), s,
                         reinterpret_cast<const float*>(workspace), gn_partial, HW, d->Cout, G, pl.splitk, p.slab, bias, y, (float*)nullptr);
      return check_launch("splitk_reduce_stats");
    }
    const long p4 = (long)HW * d->Cout / 4;   // float4s per sample
    int bx = (int)((p4 + 255) / 256);
    const int cap = cdiv(2048, d->N);
    if (bx > cap) bx = cap;
    MF_LAUNCH(splitk_reduce_kernel<0>, dim3(bx, d->N), dim3(256), 0, s, reinterpret_cast<const float*>(workspace), bias, y, p4, d->Cout,
                       pl.splitk, p.slab, y_bound);
    return check_launch("splitk_reduce");
  }
  return MF_OK;
}

/* ------------------------------------------------------------------ two independent convolutions in one launch (conv_f16x2_group.h) */
// does the plan finish inside its launch (no slab + reducer pass behind it), given the GroupNorm statistics asked of it (G = 0: none)?
static bool single_launch(const MfConvDesc* d, const Plan2& pl, int G) {
  if (pl.splitk == 1) return G == 0 || gn_parts2(d, pl, G) > 0;
  return tree_possible(d, pl) && (G == 0 || epilogue_stats_ok(d, pl, G));
}

int mf_conv2d_f16x2_group_ok(const MfConvDesc* a, int Ga, const MfConvDesc* b, int Gb) {
  static const int off = [] { const char* e = getenv("MF_CONV_GROUP"); return e && atoi(e) == 0 ? 1 : 0; }();   // MF_CONV_GROUP=0: never (A/B)
  if (off || !a || !b || a->precision != MF_CONV_FP32_F16X2 || b->precision != MF_CONV_FP32_F16X2) return 0;
  Plan2 pa, pb;
  if (make_plan2(a, &pa) != MF_OK || make_plan2(b, &pb) != MF_OK || !pa.ok || !pb.ok) return 0;
  if (!single_launch(a, pa, Ga) || !single_launch(b, pb, Gb)) return 0;
  mfc2::ConvP2 dummy{};
  return dispatch_group(pa.t.id, pb.t.id, dummy, dummy, nullptr, 2) == 1 ? 1 : 0;
}

int mf_conv2d_f16x2_group(const MfConvF16x2Call* a, const MfConvF16x2Call* b, void* stream) {
  MF_REQUIRE(a && b && a->d && b->d && a->y && b->y, MF_EINVAL, "conv_group: two calls with descriptors and outputs");
  const int Ga = a->gn_partial ? a->G : 0, Gb = b->gn_partial ? b->G : 0;
  MF_REQUIRE(mf_conv2d_f16x2_group_ok(a->d, Ga, b->d, Gb), MF_EUNSUPPORTED,
             "conv_group: these two plans cannot share a launch (mf_conv2d_f16x2_group_ok == 0: a reducer pass behind one of them, workgroup sizes "
             "that differ, or a tile pair that is not instantiated)");
  Prep qa, qb;
  int rc = conv_f16x2_prepare(a->x1s, a->x2s, a->ws, a->bias, a->y, a->x1_bound, a->x2_bound, a->w_bound, a->y_bound, a->workspace, a->workspace_bytes,
                              a->sync, a->gn_partial, a->G, nullptr, a->d, nullptr, &qa);
  if (rc) return rc;
  rc = conv_f16x2_prepare(b->x1s, b->x2s, b->ws, b->bias, b->y, b->x1_bound, b->x2_bound, b->w_bound, b->y_bound, b->workspace, b->workspace_bytes,
                          b->sync, b->gn_partial, b->G, nullptr, b->d, nullptr, &qb);
  if (rc) return rc;
  MF_REQUIRE((qa.pl.splitk == 1 || qa.tree) && (qb.pl.splitk == 1 || qb.tree), MF_EUNSUPPORTED, "conv_group: a plan needs its reducer pass");
  if (qa.tree && qb.tree) {   // both meet their split-K slices inside the launch, at the same time: hand-off regions and counters must be their own
    const char *wa = (const char*)a->workspace, *wb = (const char*)b->workspace;
    const size_t na = tree_handoff_bytes(a->d, qa.pl), nb = tree_handoff_bytes(b->d, qb.pl);
    MF_REQUIRE(wa + na <= wb || wb + nb <= wa, MF_EINVAL, "conv_group: the two workspaces overlap");
    const uint32_t *sa = a->sync, *sb = b->sync;
    const long ca = mf_conv2d_f16x2_sync_words(a->d), cb = mf_conv2d_f16x2_sync_words(b->d);
    MF_REQUIRE(sa + ca <= sb || sb + cb <= sa, MF_EINVAL, "conv_group: the two sync arrays overlap");
  }
  MF_REQUIRE(a->y != b->y, MF_EINVAL, "conv_group: one output for two convolutions");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(MF_FAM_CONV_IGEMM, s, qa.flops + qb.flops, qa.bytes + qb.bytes,
               2.0 * 3.0 * (qa.pl.M * (double)a->d->Cout * qa.pl.K + qb.pl.M * (double)b->d->Cout * qb.pl.K));
  ps.set_tag(1000 + 100 * qa.pl.t.id + qb.pl.t.id);
  rc = dispatch_group(qa.pl.t.id, qb.pl.t.id, qa.p, qb.p, s, 0);
  if (rc == -1) { set_error("conv_group: no tile pair (%d, %d)", qa.pl.t.id, qb.pl.t.id); rc = MF_EINVAL; }
  return rc;
}

int mf_prof_tag_name(int family, int tag, char* buf, int n) {
  MF_REQUIRE(buf && n > 0, MF_EINVAL, "mf_prof_tag_name: bad args");
  auto tile_of = [](int id) -> const Tile2* { for (const auto& k : kTiles2) if (k.id == id) return &k; return nullptr; };
  auto tile_type = [&](int id, char* o, int m) {
    const Tile2* t = tile_of(id);
    if (!t) { snprintf(o, m, "?"); return; }
    if (t->HG) snprintf(o, m, "mfc2::HaloTile<%d, %d, %d, %d, %d>", t->BM, t->BN, t->WM, t->WN, t->HG);
    else snprintf(o, m, "mfc2::PlainTile<%d, %d, %d, %d, %d>", t->BM, t->BN, t->WM, t->WN, t->NST);
  };
  if ((family == MF_FAM_CONV_IGEMM || family == MF_FAM_CONV_GN_FUSED) && tag > 0) {
    const int terms = tag >= 100000 ? 1 : 3, id = tag % 100000;
    if (id >= 1000) {
      char a[96], b[96];
      tile_type((id - 1000) / 100, a, sizeof a);
      tile_type((id - 1000) % 100, b, sizeof b);
      snprintf(buf, n, "void mfc2::conv_group_kernel<%s, %s, 3>(mfc2::ConvP2, mfc2::ConvP2, int)", a, b);
      return MF_OK;
    }
    if (const Tile2* t = tile_of(id)) {
      if (t->HG) snprintf(buf, n, "void mfc2::conv_halo_kernel<%d, %d, %d, %d, %d, %d>(mfc2::ConvP2)", t->BM, t->BN, t->WM, t->WN, t->HG, terms);
      else snprintf(buf, n, "void mfc2::conv_f16x2_kernel<%d, %d, %d, %d, %d, %d>(mfc2::ConvP2)", t->BM, t->BN, t->WM, t->WN, t->NST, terms);
      return MF_OK;
    }
  }
  snprintf(buf, n, "%s (tag %d)", mf_prof_family_name(family), tag);
  return MF_OK;
}

#include "conv_f16x2_wino.inc"

}  // extern "C"
