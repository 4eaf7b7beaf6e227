// api.hip — error reporting and version string of libseal3d_hip.
#include "s3d_common.hpp"
#include <stdarg.h>

namespace s3d {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace s3d

S3D_EXPORT const char* s3d_last_error(void) { return s3d::g_err; }
S3D_EXPORT const char* s3d_version(void) { return "seal3d-hip 0.1 gfx950"; }
