// common.h -- shared host-side plumbing for libmedfusion_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <tuple>
#include <type_traits>
#include <utility>

#include "../../include/medfusion_hip.h"

namespace mf {

// thread-local error string (mf_last_error)
void set_error(const char* fmt, ...);

// launch timing (mf_prof_*): when enabled, brackets a launch with two events on its stream.
struct ProfScope {
  // flops: ALGORITHMIC flops of the reference op; bytes: its compulsory HBM bytes; exec_flops: flops the hardware executes for it
  // (e.g. 6 or 3 matrix terms per product, 4/9 of the MACs in the sub-pixel form of nearest-x2 + 3x3); < 0: same as flops
  ProfScope(int family, hipStream_t s, double flops, double bytes, double exec_flops = -1.0);
  ~ProfScope();
  int idx;
  hipStream_t stream;
  int launches;   // launches issued inside this scope so far
  ProfScope* prev;   // the enclosing scope of the calling thread (restored by the destructor)
  void set_tag(int tag, int variant = 0);   // which kernel instantiation the scope's launch is (mf_prof_rows groups by it); no-op when timing is off
};
bool prof_on();
// Timing of a scope (round 4): its FIRST launch goes out through hipExtLaunchKernel with the scope's two events as the dispatch's own start / stop
// events -- the kernel's execution interval as the profiler sees it, no event packets in the queue between dependent kernels (two hipEventRecord
// per launch inflated the conv family by ~4 % against rocprofv3).  A further launch in the same scope re-records the stop event behind itself.
// -> true: `launch_impl` issued the kernel itself.
bool prof_launch(const void* func, dim3 grid, dim3 block, void** argv, size_t lds, hipStream_t s);

// ---- every kernel of the library is launched through MF_LAUNCH: an ordinary hipLaunchKernel -- and, while a command list records on the
// calling thread (mf_cmdlist_begin), the same launch plus a copy of (kernel, geometry, kernarg bytes).  mf_cmdlist_replay re-issues the
// recorded launches from C in their recorded order: the denoise loop's iteration (~160 launches, all of whose per-iteration values come
// from a device step counter) costs ~1 us of host time per launch instead of a Python -> ctypes round trip each.
struct CmdList;
CmdList* recording_list();   // api.hip; null when this thread is not recording
void cmdlist_add(CmdList* cl, const void* func, dim3 grid, dim3 block, size_t lds, void* const* argv, const size_t* sizes, const size_t* aligns, int n);

template <typename... KArgs, typename... Args, size_t... I>
inline void launch_impl(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t s, std::index_sequence<I...>, Args&&... args) {
  std::tuple<std::decay_t<KArgs>...> v{static_cast<std::decay_t<KArgs>>(std::forward<Args>(args))...};   // the kernel's own parameter types
  void* argv[sizeof...(KArgs) + 1] = {static_cast<void*>(&std::get<I>(v))..., nullptr};
  if (CmdList* cl = recording_list()) {
    const size_t sizes[sizeof...(KArgs) + 1] = {sizeof(std::decay_t<KArgs>)..., 0};
    const size_t aligns[sizeof...(KArgs) + 1] = {alignof(std::decay_t<KArgs>)..., 0};
    cmdlist_add(cl, reinterpret_cast<const void*>(kernel), grid, block, lds, argv, sizes, aligns, (int)sizeof...(KArgs));
  }
  if (prof_on() && prof_launch(reinterpret_cast<const void*>(kernel), grid, block, argv, lds, s)) return;
  (void)hipLaunchKernel(reinterpret_cast<const void*>(kernel), grid, block, argv, lds, s);
}
template <typename... KArgs, typename... Args>
inline void launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t s, Args&&... args) {
  static_assert(sizeof...(KArgs) == sizeof...(Args), "kernel argument count");
  launch_impl(kernel, grid, block, lds, s, std::index_sequence_for<KArgs...>{}, std::forward<Args>(args)...);
}
#define MF_LAUNCH(kernel, grid, block, lds, stream, ...) ::mf::launch(kernel, grid, block, lds, stream, __VA_ARGS__)

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return MF_ELAUNCH;
  }
  return MF_OK;
}

#define MF_REQUIRE(cond, code, ...)   \
  do {                                \
    if (!(cond)) {                    \
      ::mf::set_error(__VA_ARGS__);   \
      return (code);                  \
    }                                 \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// hipFuncSetAttribute belongs to the (function, device) pair: one flag per device (a process may drive several -- tests, tools), `seen` is a
// function-local static array of the caller.  True the first time the current device comes by.
struct DeviceOnce { bool seen[64] = {}; };
static inline bool first_use_on_device(DeviceOnce& o) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;   // unknown device: set the attribute every time
  if (o.seen[dev]) return false;
  o.seen[dev] = true;
  return true;
}

// MONAI Swish: x * sigmoid(1.0 * x)  (conv_blocks.py act)
__device__ __forceinline__ float swish(float x) { return x / (1.0f + __expf(-x)); }
// accurate variant used where parity margins are tight (embedding MLPs): expf, IEEE divide
__device__ __forceinline__ float swish_acc(float x) { return x * (1.0f / (1.0f + expf(-x))); }
#ifndef MF_SWISH_APPLY
#define MF_SWISH_APPLY 3
#endif
// the GroupNorm apply pass's form (VALU-bound pass: 28 of its ~55 instructions per element are this function).  0: swish_acc.
// 1: the IEEE division (12 instructions) replaced by v_rcp_f32 + one Newton step (4; error < 1 ulp).  2: additionally expf without its
// overflow / underflow selects (the sigmoid saturates by itself: exp -> inf gives x * 0, exp -> 0 gives x).
// 3 (default since round 4): x * v_rcp(1 + v_exp(x * -log2 e)) -- five instructions.  Both hardware functions are good to 1 ulp; the rounding of
// the exponent's argument adds |x| 6e-8 to the relative error of e^-x, which only shows where the sigmoid is small (x << 0, where the
// result itself is ~0): |error| < 3 ulp of the result, or 1e-8 absolute, whichever is larger -- the fp32 class of the path's 1e-4 tolerance
// (measured at block and trajectory level: profiles/r04_*).  The apply pass and the fused tail of the convolution are VALU-bound
// (scripts/conv_timeline.py --fused: 70 instructions per element, 28 of them the IEEE form of this function), hence the short form.
// x -> -inf gives -inf * 0 = NaN like x * sigmoid(x) of the reference; x -> +inf gives x.
__device__ __forceinline__ float swish_apply(float x) {
#if MF_SWISH_APPLY == 0
  return swish_acc(x);
#elif MF_SWISH_APPLY == 3
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
#else
#if MF_SWISH_APPLY == 1
  const float d = 1.0f + expf(-x);
#else
  const float a = -1.44269504088896341f * x, n = rintf(a);
  const float f = (a - n) + (__builtin_fmaf(x, -1.44269504088896341f, -a) + x * -1.92596299112661746e-8f);   // a's rounding error, log2(e)'s low part
  const float d = 1.0f + ldexpf(__builtin_amdgcn_exp2f(f), (int)n);
#endif
  float r = __builtin_amdgcn_rcpf(d);
  r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
  return x * r;
#endif
}

}  // namespace mf
