// libnpp_hip: error reporting, version, and the per-family launch profiler used by bench.py.
#include <stdarg.h>
#include <mutex>
#include <vector>
#include "common.h"
#include <stdio.h>
#include <stdlib.h>

static thread_local char g_err[512] = "";

void npp_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int npp_check_launch(const char* what) {
  // debugging aids: NPP_TRACE_LAUNCH=1 names every launch on stderr, NPP_SYNC_LAUNCH=1 waits for it (a fault then points at
  // its kernel).  Never set inside a hipGraph capture.
  static const bool trace = getenv("NPP_TRACE_LAUNCH") != nullptr, sync = getenv("NPP_SYNC_LAUNCH") != nullptr;
  if (trace) { fprintf(stderr, "[npp] %s\n", what); fflush(stderr); }
  if (sync) {
    hipError_t se = hipDeviceSynchronize();
    if (se != hipSuccess) {
      npp_set_error("%s: failed at synchronisation: %s", what, hipGetErrorString(se));
      return NPP_E_HIP;
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    npp_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return NPP_E_HIP;
  }
  return NPP_OK;
}

// HIP keeps the last error until somebody reads it: after a failed hipGraph capture the next npp_check_launch would report
// "operation failed due to a previous error during capture" for a perfectly good launch.  The eager fallback calls this.
extern "C" int npp_clear_hip_error(void) { return (int)hipGetLastError(); }

#ifndef NPP_SRC_HASH
#define NPP_SRC_HASH "unknown"
#endif
// "... src <hash>": the hash build.sh computed over the sources (the value npp_amd._lib.kernel_source_hash() gives for the same files)
extern "C" const char* npp_version(void) { return "npp_hip 0.1 (gfx950) src " NPP_SRC_HASH; }
extern "C" const char* npp_last_error(void) { return g_err; }

// ---- profiler ------------------------------------------------------------------------------------
namespace {
struct ProfState {
  std::mutex mu;
  int family = NPP_FAM_NONE;
  int dtype = -1;
  std::vector<hipEvent_t> ev;  // pairs
  int used = 0;                // events used
  double flops = 0, bytes = 0;
} g_prof;
}  // namespace

ProfScope::ProfScope(int family, int dtype, hipStream_t s, double flops, double bytes)
    : slot(-1), stream(s), flops_(flops), bytes_(bytes) {
  if (g_prof.family != family) return;
  if (g_prof.dtype >= 0 && g_prof.dtype != dtype) return;
  std::lock_guard<std::mutex> lk(g_prof.mu);
  if (g_prof.used + 2 > (int)g_prof.ev.size()) {
    if (g_prof.ev.size() >= 400000) return;
    for (int i = 0; i < 2; ++i) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return;
      g_prof.ev.push_back(e);
    }
  }
  slot = g_prof.used;
  g_prof.used += 2;
  g_prof.flops += flops;
  g_prof.bytes += bytes;
  hipEventRecord(g_prof.ev[slot], s);
}
ProfScope::~ProfScope() {
  if (slot >= 0) (void)hipEventRecord(g_prof.ev[slot + 1], stream);
}

void ProfScope::cancel() {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lk(g_prof.mu);
  if (g_prof.used == slot + 2) {
    g_prof.used -= 2;
    g_prof.flops -= flops_;
    g_prof.bytes -= bytes_;
  }
  slot = -1;
}

extern "C" int npp_prof_begin(int family, int dtype_filter) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  g_prof.family = family;
  g_prof.dtype = dtype_filter;
  g_prof.used = 0;
  g_prof.flops = g_prof.bytes = 0;
  return NPP_OK;
}

extern "C" int npp_prof_end(double* ms_total, double* flops_total, double* bytes_total, int64_t* launches) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  double ms = 0;
  for (int i = 0; i + 1 < g_prof.used; i += 2) {
    float t = 0;
    if (hipEventSynchronize(g_prof.ev[i + 1]) != hipSuccess) continue;
    if (hipEventElapsedTime(&t, g_prof.ev[i], g_prof.ev[i + 1]) == hipSuccess) ms += t;
  }
  if (ms_total) *ms_total = ms;
  if (flops_total) *flops_total = g_prof.flops;
  if (bytes_total) *bytes_total = g_prof.bytes;
  if (launches) *launches = g_prof.used / 2;
  g_prof.family = NPP_FAM_NONE;
  g_prof.used = 0;
  return NPP_OK;
}
