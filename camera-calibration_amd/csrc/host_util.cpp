// host_util.cpp -- thread-local error message behind ccalib_last_error().
#include "host_util.hpp"

#include <cstdio>

#include "ccalib.h"

namespace ccalib {
static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace ccalib

extern "C" const char* ccalib_last_error(void) { return ccalib::g_err; }
extern "C" const char* ccalib_version(void) { return "ccalib 0.1 (gfx950)"; }

extern "C" int ccalib_dist_coeffs_out(int flags, int stereo) {
    if (!(flags & (CCALIB_RATIONAL_MODEL | CCALIB_THIN_PRISM_MODEL | CCALIB_TILTED_MODEL))) return 5;
    if (!stereo && (flags & CCALIB_THIN_PRISM_MODEL) && !(flags & CCALIB_TILTED_MODEL)) return 12;
    return 14;
}
