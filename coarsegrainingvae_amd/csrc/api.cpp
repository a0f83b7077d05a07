// Version / error plumbing of the C ABI (include/cgvae_hip.h).
#include <stdarg.h>
#include <string.h>
#include "cgv_common.h"

namespace cgv {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace cgv

extern "C" {
int cgv_version(void) { return CGV_VERSION; }
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
