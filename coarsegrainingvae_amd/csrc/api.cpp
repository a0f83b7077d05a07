// Version / error plumbing of the C ABI (include/cgvae_hip.h).
#include <stdarg.h>
#include <string.h>
#include <atomic>
#include "cgv_common.h"

namespace cgv {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Option table (cgv_set_option): process-wide A/B switches with fixed defaults.  Relaxed atomics: a launcher reads
// its options once, at the top of the call.
static const int g_opt_default[CGV_OPT_COUNT] = {
    /* CGV_OPT_MSG_FWD_SPLIT */ -1, /* CGV_OPT_MSG_BWD_SPLIT */ -1, /* CGV_OPT_MSG_FWD_KERNEL */ 0,
    /* CGV_OPT_GRP_WAVES */ 4,      /* CGV_OPT_GRP_RECORDS */ 0,    /* CGV_OPT_CSR_BUILD */ 0,
    /* CGV_OPT_PSEUDO_CHUNKS */ 0,  /* CGV_OPT_WGRAD_TILING */ 0,   /* CGV_OPT_TILE_FWD_LDS_MIN */ 448,
    /* CGV_OPT_BWD_INPUT_WAVES */ 0, /* CGV_OPT_PSEUDO_FWD */ 0, /* CGV_OPT_DECODER_FAT */ 1, /* CGV_OPT_DECODER_WLDS */ 1,
    /* CGV_OPT_SKINNY_ROWS */ 0, /* CGV_OPT_TILE_FWD_BAL */ 1, /* CGV_OPT_OPTIM_ONE_LAUNCH */ 0, /* CGV_OPT_DECODER_COLSPLIT */ 2, /* CGV_OPT_DECODER_NODESPLIT */ 1,
    /* CGV_OPT_MSG_FWD_BALANCED */ 3, /* CGV_OPT_BWD_INPUT_SPLIT */ -1,
    /* CGV_OPT_MSG_BWD_MFMA */ -1, /* CGV_OPT_STREAMK */ 0};
static std::atomic<int> g_opt[CGV_OPT_COUNT] = {{-1}, {-1}, {0}, {4}, {0}, {0}, {0}, {0}, {448}, {0}, {0}, {1}, {1}, {0}, {1}, {0}, {2}, {1}, {3}, {-1}, {-1}, {0}};
int option(int id) { return g_opt[id].load(std::memory_order_relaxed); }
}  // namespace cgv

extern "C" {
int cgv_version(void) { return CGV_VERSION; }
int cgv_set_option(int option, int value) {
  CGV_REQUIRE(option >= 0 && option < CGV_OPT_COUNT, "unknown option");
  cgv::g_opt[option].store(value, std::memory_order_relaxed);
  return 0;
}
int cgv_get_option(int option) {
  if (option < 0 || option >= CGV_OPT_COUNT) return INT32_MIN;
  return cgv::g_opt[option].load(std::memory_order_relaxed);
}
int cgv_reset_options(void) {
  for (int i = 0; i < CGV_OPT_COUNT; ++i) cgv::g_opt[i].store(cgv::g_opt_default[i], std::memory_order_relaxed);
  return 0;
}
const char* cgv_last_error_string(void) { return cgv::g_err; }
int cgv_rbf_supported(int R) {
  switch (R) {
#define X(n) case n:
    CGV_RBF_LIST(X)
#undef X
    return 1;
    default:
      return 0;
  }
}
int cgv_geom_stride(int R) { return cgv::geom_stride(R); }
int cgv_geom_unit_offset(int R) { return cgv::geom_unit_offset(R); }
}
