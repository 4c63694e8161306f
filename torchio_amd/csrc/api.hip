// api.hip — error reporting and introspection for libtio_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "common.hpp"

namespace tio {

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}

int fail(int status, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
  return status;
}

// Launch errors only (hipGetLastError); never synchronises.
int check_launch(const char* what) {
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail(TIO_ERR_LAUNCH, "%s: launch failed: %s", what, hipGetErrorString(err));
  return TIO_OK;
}

// ---- environment switches: parsed once, re-parsed by tio_reload_env() --------------------------------------------
static std::atomic<const EnvSwitches*> g_env{nullptr};
static std::mutex g_env_mu;

static const EnvSwitches* parse_env() {
  EnvSwitches* e = new EnvSwitches();  // (snapshots are never freed: a launch may still be reading the previous one; tests only)
  auto num = [](const char* name, int unset) { const char* v = getenv(name); return v != nullptr ? atoi(v) : unset; };
  e->nearest_kernel = num("TIO_NEAREST_KERNEL", 1) != 0;
  if (const char* v = getenv("TIO_NEAREST_EPS")) { e->has_nearest_eps = 1; e->nearest_eps = static_cast<float>(atof(v)); }
  if (const char* v = getenv("TIO_RESAMPLE_PATH")) e->resample_path = strcmp(v, "gather") == 0 ? 1 : (strcmp(v, "tile") == 0 ? 2 : 0);
  e->tile_variant = num("TIO_TILE_VARIANT", 0);
  e->tile_lds_floats = num("TIO_TILE_LDS_FLOATS", 0);
  e->tile_ablate = num("TIO_TILE_ABLATE", 0);
  e->resample_exact = getenv("TIO_RESAMPLE_EXACT") != nullptr;
  if (const char* v = getenv("TIO_FAST_KERNEL")) e->fast_kernel = strcmp(v, "brick") == 0 ? 1 : (strcmp(v, "planned") == 0 ? 2 : 0);
  e->planned_lean = num("TIO_PLANNED_LEAN", 1) != 0;
  e->dma_packed = num("TIO_DMA_PACKED", 1) != 0;
  e->exact_plan = num("TIO_EXACT_PLAN", -1);
  e->exact_lean = num("TIO_EXACT_LEAN", -1);
  e->lean_interleave = num("TIO_LEAN_INTERLEAVE", 1);
  e->lean_multi = num("TIO_LEAN_MULTI", 1);
  e->nearest_exact = num("TIO_NEAREST_EXACT", 1);
  e->lean_pair = num("TIO_LEAN_PAIR", 1);
  e->lean_label = num("TIO_LEAN_LABEL", 1);
  e->nearest_lds = num("TIO_NEAREST_LDS", -1);
  e->fast_fill_recheck = num("TIO_FAST_FILL_RECHECK", 1) != 0;
  e->conv_no_fuse = getenv("TIO_CONV_NO_FUSE") != nullptr;
  e->conv_ring = getenv("TIO_CONV_RING") != nullptr;
  e->march_segs = num("TIO_MARCH_SEGS", 0);
  e->march_order = num("TIO_MARCH_ORDER", -1);
  e->min_blocks = num("TIO_MIN_BLOCKS", 0);
  return e;
}

const EnvSwitches& env_switches() {
  const EnvSwitches* e = g_env.load(std::memory_order_acquire);
  if (e != nullptr) return *e;
  std::lock_guard<std::mutex> lock(g_env_mu);
  e = g_env.load(std::memory_order_acquire);
  if (e == nullptr) { e = parse_env(); g_env.store(e, std::memory_order_release); }
  return *e;
}

}  // namespace tio

extern "C" void tio_reload_env(void) {
  std::lock_guard<std::mutex> lock(tio::g_env_mu);
  tio::g_env.store(tio::parse_env(), std::memory_order_release);
}

extern "C" int tio_abi_version(void) { return TIO_ABI_VERSION; }

extern "C" const char* tio_last_error(void) { return tio::g_error; }

extern "C" int tio_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}
