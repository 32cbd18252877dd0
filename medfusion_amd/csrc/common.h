// common.h -- shared host-side plumbing for libmedfusion_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

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
};
bool prof_on();

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

}  // namespace mf
