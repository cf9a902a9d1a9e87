// host_util.cpp -- thread-local error message behind ccalib_last_error().
#include "host_util.hpp"

#include <cstdio>
#include <cstring>
#include <string>

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

// Distinct files named libamdhip64.so* that are mapped into this process.  More than one means two HIP runtimes live
// side by side (e.g. a framework's bundled copy and the system's): each has its own devices, streams and memory, and
// a stream or device pointer created by one is "invalid argument" to the other.
int hip_runtime_copies(char* first_two, size_t cap) {
    std::FILE* f = std::fopen("/proc/self/maps", "r");
    if (!f) return 0;
    std::string seen[8];
    int n = 0;
    char line[1024];
    while (std::fgets(line, sizeof(line), f)) {
        const char* path = std::strchr(line, '/');
        if (!path || !std::strstr(path, "libamdhip64.so")) continue;
        std::string p(path);
        while (!p.empty() && (p.back() == '\n' || p.back() == ' ')) p.pop_back();
        bool known = false;
        for (int i = 0; i < n; i++) known = known || seen[i] == p;
        if (!known && n < 8) seen[n++] = p;
    }
    std::fclose(f);
    if (first_two && cap) std::snprintf(first_two, cap, "%s%s%s", n > 0 ? seen[0].c_str() : "", n > 1 ? " and " : "", n > 1 ? seen[1].c_str() : "");
    return n;
}

const char* hip_runtime_note() {
    static thread_local char note[700];
    char two[600];
    if (hip_runtime_copies(two, sizeof(two)) < 2) return "";
    std::snprintf(note, sizeof(note),
                  " [two HIP runtimes are mapped in this process (%s): streams and device pointers of one are invalid in the "
                  "other; load the framework that bundles its own runtime BEFORE libccalib.so so that both use one copy]",
                  two);
    return note;
}
}  // namespace ccalib

extern "C" const char* ccalib_last_error(void) { return ccalib::g_err; }
extern "C" const char* ccalib_version(void) { return "ccalib 0.1 (gfx950)"; }

extern "C" int ccalib_dist_coeffs_out(int flags, int stereo) {
    if (!(flags & (CCALIB_RATIONAL_MODEL | CCALIB_THIN_PRISM_MODEL | CCALIB_TILTED_MODEL))) return 5;
    if (!stereo && (flags & CCALIB_THIN_PRISM_MODEL) && !(flags & CCALIB_TILTED_MODEL)) return 12;
    return 14;
}
