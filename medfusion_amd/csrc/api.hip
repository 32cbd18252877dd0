// api.hip -- version, thread-local error string, launch-timing registry (mf_prof_*).
#include "common.h"

#include <mutex>
#include <string>
#include <string.h>
#include <vector>

namespace mf {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- command lists (common.h: MF_LAUNCH)
struct CmdEntry {
  const void* func;
  dim3 grid, block;
  size_t lds;
  size_t blob_off;            // first kernarg byte of this launch inside CmdList::blob
  int nargs;
  size_t arg_off_first;       // index of this launch's first entry in CmdList::arg_off
};
struct CmdList {
  std::vector<CmdEntry> entries;
  std::vector<unsigned char> blob;     // kernarg copies, each at its own alignment
  std::vector<size_t> arg_off;         // byte offset of every argument inside blob
};
static thread_local CmdList* g_rec = nullptr;
CmdList* recording_list() { return g_rec; }

void cmdlist_add(CmdList* cl, const void* func, dim3 grid, dim3 block, size_t lds, void* const* argv, const size_t* sizes, const size_t* aligns, int n) {
  CmdEntry e{func, grid, block, lds, cl->blob.size(), n, cl->arg_off.size()};
  for (int i = 0; i < n; ++i) {
    size_t off = (cl->blob.size() + aligns[i] - 1) / aligns[i] * aligns[i];
    cl->blob.resize(off + sizes[i]);
    memcpy(cl->blob.data() + off, argv[i], sizes[i]);
    cl->arg_off.push_back(off);
  }
  cl->entries.push_back(e);
}

struct ProfRec {
  int family;
  hipEvent_t a, b;
  double flops, bytes, exec_flops;
  int tag, variant;
};
static bool g_prof = false;
static std::mutex g_mu;
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;

bool prof_on() { return g_prof; }

static hipEvent_t get_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

static thread_local ProfScope* g_scope = nullptr;   // the innermost timing scope of the calling thread

ProfScope::ProfScope(int family, hipStream_t s, double flops, double bytes, double exec_flops) : idx(-1), stream(s), launches(0), prev(nullptr) {
  if (!g_prof) return;
  std::lock_guard<std::mutex> lk(g_mu);
  ProfRec r{family, get_event(), get_event(), flops, bytes, exec_flops < 0 ? flops : exec_flops, 0, 0};
  g_recs.push_back(r);
  idx = (int)g_recs.size() - 1;
  prev = g_scope;   // (scopes may nest: the enclosing one takes over again when this one ends -- ADVICE r04)
  g_scope = this;
}
ProfScope::~ProfScope() {
  if (idx < 0) return;
  if (g_scope == this) g_scope = prev;
  if (launches == 0) {   // a scope without a launch (an entry point that returned early): both events recorded here so that the query finds them
    std::lock_guard<std::mutex> lk(g_mu);
    (void)hipEventRecord(g_recs[idx].a, stream);
    (void)hipEventRecord(g_recs[idx].b, stream);
  }
}

void ProfScope::set_tag(int tag, int variant) {
  if (idx < 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  g_recs[idx].tag = tag;
  g_recs[idx].variant = variant;
}

bool prof_launch(const void* func, dim3 grid, dim3 block, void** argv, size_t lds, hipStream_t s) {
  ProfScope* sc = g_scope;
  if (sc == nullptr || sc->idx < 0) return false;
  hipEvent_t a, b;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    a = g_recs[sc->idx].a;
    b = g_recs[sc->idx].b;
  }
  if (sc->launches++ == 0) {
    (void)hipExtLaunchKernel(func, grid, block, argv, lds, s, a, b, 0);
    return true;
  }
  (void)hipLaunchKernel(func, grid, block, argv, lds, s);   // second, third ... launch of the scope: the stop event moves behind it
  (void)hipEventRecord(b, s);
  return true;
}

}  // namespace mf

using namespace mf;

namespace {
typedef float pr_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 pr_f16x8 __attribute__((ext_vector_type(8)));
// nothing but v_mfma_f32_32x32x16_f16 from registers: 4 independent chains per wave, operands loaded once (mf_mfma_rate_probe_f16)
__global__ __launch_bounds__(512) void mfma_rate_probe_kernel(const pr_f16x8* __restrict__ in, float* __restrict__ out, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  pr_f16x8 a[4], b[4];
  pr_f32x16 acc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    a[k] = in[(size_t)tid * 8 + 2 * k];
    b[k] = in[(size_t)tid * 8 + 2 * k + 1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k], b[k], acc[k], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[k][r];
  out[tid] = s;
}
}  // namespace

extern "C" {

int mf_version(void) { return MF_VERSION; }
const char* mf_last_error(void) { return g_err; }

int mf_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_prof = on != 0;
  return MF_OK;
}

int mf_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& r : g_recs) {
    (void)hipEventSynchronize(r.b);
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  g_recs.clear();
  return MF_OK;
}

int mf_prof_query(int family, double* ms, int64_t* launches, double* flops, double* bytes) {
  return mf_prof_query2(family, ms, launches, flops, bytes, nullptr);
}

int mf_prof_query2(int family, double* ms, int64_t* launches, double* flops, double* bytes, double* exec_flops) {
  std::lock_guard<std::mutex> lk(g_mu);
  double t = 0, fl = 0, by = 0, ex = 0;
  int64_t n = 0;
  for (auto& r : g_recs) {
    if (r.family != family) continue;
    if (hipEventSynchronize(r.b) != hipSuccess) {
      set_error("mf_prof_query: event sync failed");
      return MF_ELAUNCH;
    }
    float dt = 0;
    (void)hipEventElapsedTime(&dt, r.a, r.b);
    t += dt;
    fl += r.flops;
    by += r.bytes;
    ex += r.exec_flops;
    ++n;
  }
  if (ms) *ms = t;
  if (launches) *launches = n;
  if (flops) *flops = fl;
  if (bytes) *bytes = by;
  if (exec_flops) *exec_flops = ex;
  return MF_OK;
}

int mf_prof_rows(int family, MfProfRow* rows, int max_rows) {
  MF_REQUIRE(rows && max_rows > 0, MF_EINVAL, "mf_prof_rows: bad args");
  std::lock_guard<std::mutex> lk(g_mu);
  int n = 0;
  for (auto& r : g_recs) {
    if (r.family != family) continue;
    if (hipEventSynchronize(r.b) != hipSuccess) {
      set_error("mf_prof_rows: event sync failed");
      return MF_ELAUNCH;
    }
    float dt = 0;
    (void)hipEventElapsedTime(&dt, r.a, r.b);
    int k = 0;
    while (k < n && !(rows[k].tag == r.tag && rows[k].variant == r.variant)) ++k;
    if (k == n) {
      if (n == max_rows) continue;   // (more instantiations than the caller has room for: the rest is dropped, the family totals still hold)
      rows[n++] = MfProfRow{r.tag, r.variant, 0, 0.0, 0.0, 0.0, 0.0};
    }
    rows[k].launches += 1;
    rows[k].ms += dt;
    rows[k].flops += r.flops;
    rows[k].bytes += r.bytes;
    rows[k].exec_flops += r.exec_flops;
  }
  return n;
}

/* ---- command lists: record the launches of one loop iteration, replay them from C (include/medfusion_hip.h) */
int mf_cmdlist_begin(void) {
  MF_REQUIRE(g_rec == nullptr, MF_EINVAL, "cmdlist_begin: this thread is already recording");
  MF_REQUIRE(!g_prof, MF_EINVAL, "cmdlist_begin: launch timing (mf_prof_enable) is on -- replayed launches would not be timed");
  g_rec = new CmdList();
  return MF_OK;
}

int mf_cmdlist_end(void** list) {
  MF_REQUIRE(g_rec != nullptr && list, MF_EINVAL, "cmdlist_end: not recording");
  *list = g_rec;
  g_rec = nullptr;
  return MF_OK;
}

int mf_cmdlist_count(const void* list) { return list ? (int)static_cast<const CmdList*>(list)->entries.size() : 0; }

int mf_cmdlist_replay(const void* list, int times, void* stream) {
  MF_REQUIRE(list && times >= 0, MF_EINVAL, "cmdlist_replay: bad args");
  MF_REQUIRE(g_rec == nullptr, MF_EINVAL, "cmdlist_replay: this thread is recording");
  const CmdList* cl = static_cast<const CmdList*>(list);
  std::vector<void*> argv;
  // kernarg pointers point into the list's own blob (hipLaunchKernel copies the bytes at launch time)
  unsigned char* base = const_cast<unsigned char*>(cl->blob.data());
  size_t most = 0;
  for (const CmdEntry& e : cl->entries) most = e.nargs > (int)most ? (size_t)e.nargs : most;
  argv.resize(most + 1);
  hipStream_t s = (hipStream_t)stream;
  for (int t = 0; t < times; ++t)
    for (const CmdEntry& e : cl->entries) {
      for (int i = 0; i < e.nargs; ++i) argv[i] = base + cl->arg_off[e.arg_off_first + i];
      const hipError_t err = hipLaunchKernel(e.func, e.grid, e.block, argv.data(), e.lds, s);
      if (err != hipSuccess) {
        set_error("cmdlist_replay: launch failed: %s", hipGetErrorString(err));
        return MF_ELAUNCH;
      }
    }
  return MF_OK;
}

int mf_cmdlist_free(void* list) {
  delete static_cast<CmdList*>(list);
  return MF_OK;
}

const char* mf_prof_family_name(int f) {
  static const char* names[MF_FAM_COUNT] = {"conv_igemm", "conv_direct", "splitk_reduce", "gn_stats", "gn_apply",
                                            "linear", "sched", "noise", "attention", "misc", "conv_gn_fused", "wino_xform"};
  return (f >= 0 && f < MF_FAM_COUNT) ? names[f] : "?";
}

int mf_mfma_rate_probe_f16(const void* operands, float* out, int workgroups, int iters, double* flops, void* stream) {
  MF_REQUIRE(operands && out && workgroups > 0 && iters > 0, MF_EINVAL, "mfma_rate_probe: bad args");
  MF_LAUNCH(mfma_rate_probe_kernel, dim3(workgroups), dim3(512), 0, (hipStream_t)stream, reinterpret_cast<const pr_f16x8*>(operands), out, iters);
  if (flops) *flops = (double)workgroups * 8.0 * 4.0 * (double)iters * 32768.0;
  return check_launch("mfma_rate_probe");
}

}  // extern "C"
