// yaml_out.cpp -- the result files the reference writes with cv::FileStorage, in the same YAML dialect
// ("%YAML:1.0", "!!opencv-matrix" nodes, dt: d), so the reference's own readers (FileStorage::READ at
// binocular_chessboard.cpp:285-295) load them unchanged:
//   monocular:  imageSize, M, D, R (sequence of 3x1), t (sequence of 3x1)      monocular_chessboard.cpp:321-328
//   stereo:     M1, D1, M2, D2                                                 binocular_chessboard.cpp:203-210
//               imageSize, R, T (+ E, F)                                       binocular_chessboard.cpp:234-241
// The rectification entries of the reference's extrinsics file (R1, R2, P1, P2, Q) come from
// cv::stereoRectify, which is outside this library; a caller that has them appends them itself.
#include <cmath>
#include <cstdio>
#include <string>

#include "ccalib.h"
#include "host_util.hpp"

namespace {

// cv::FileStorage prints integral doubles as "1." and everything else with 17 significant digits.
std::string number(double v) {
    char buf[64];
    if (std::isnan(v)) return ".Nan";
    if (std::isinf(v)) return v < 0 ? "-.Inf" : ".Inf";
    if (v == std::floor(v) && std::fabs(v) < 1e9)
        std::snprintf(buf, sizeof buf, "%d.", (int)v);
    else
        std::snprintf(buf, sizeof buf, "%.16e", v);
    return buf;
}

void matrix(std::FILE* f, const char* indent, int rows, int cols, const double* d) {
    std::fprintf(f, "!!opencv-matrix\n%srows: %d\n%scols: %d\n%sdt: d\n%sdata: [ ", indent, rows, indent, cols, indent, indent);
    int col = (int)std::string(indent).size() + 8;
    for (int i = 0; i < rows * cols; i++) {
        std::string s = number(d[i]);
        if (i + 1 < rows * cols) s += ",";
        if (col + (int)s.size() + 1 > 80 && i > 0) {  // FileStorage wraps long lines; readers do not care
            std::fprintf(f, "\n%s    ", indent);
            col = (int)std::string(indent).size() + 4;
        }
        std::fprintf(f, "%s ", s.c_str());
        col += (int)s.size() + 1;
    }
    std::fprintf(f, "]\n");
}

void named_matrix(std::FILE* f, const char* name, int rows, int cols, const double* d) {
    std::fprintf(f, "%s: ", name);
    matrix(f, "   ", rows, cols, d);
}

void matrix_sequence(std::FILE* f, const char* name, int n, int rows, const double* d) {
    std::fprintf(f, "%s:\n", name);
    for (int i = 0; i < n; i++) {
        std::fprintf(f, "   - ");
        matrix(f, "      ", rows, 1, d + (size_t)rows * i);
    }
}

}  // namespace

extern "C" int ccalib_write_mono_yaml(const char* path, int width, int height, const double* K, const double* dist,
                                      int ndist, int nviews, const double* rvecs, const double* tvecs) {
    if (!path || !K || (ndist > 0 && !dist) || ndist < 0 || nviews < 0 || (nviews > 0 && (!rvecs || !tvecs)))
        return ccalib::set_error(CCALIB_ERR_ARG, "ccalib_write_mono_yaml: bad arguments");
    std::FILE* f = std::fopen(path, "w");
    if (!f) return ccalib::set_error(CCALIB_ERR_ARG, "cannot open %s for writing", path);
    std::fprintf(f, "%%YAML:1.0\n---\nimageSize: [ %d, %d ]\n", width, height);
    named_matrix(f, "M", 3, 3, K);
    named_matrix(f, "D", ndist, 1, dist);
    matrix_sequence(f, "R", nviews, 3, rvecs);
    matrix_sequence(f, "t", nviews, 3, tvecs);
    std::fclose(f);
    return CCALIB_OK;
}

extern "C" int ccalib_write_stereo_yaml(const char* intrinsics_path, const char* extrinsics_path, int width, int height,
                                        const double* K1, const double* dist1, int ndist1, const double* K2,
                                        const double* dist2, int ndist2, const double* R, const double* T,
                                        const double* E, const double* F) {
    if (!intrinsics_path || !extrinsics_path || !K1 || !K2 || !dist1 || !dist2 || !R || !T || ndist1 < 1 || ndist2 < 1)
        return ccalib::set_error(CCALIB_ERR_ARG, "ccalib_write_stereo_yaml: bad arguments");
    std::FILE* f = std::fopen(intrinsics_path, "w");
    if (!f) return ccalib::set_error(CCALIB_ERR_ARG, "cannot open %s for writing", intrinsics_path);
    std::fprintf(f, "%%YAML:1.0\n---\n");
    named_matrix(f, "M1", 3, 3, K1);
    named_matrix(f, "D1", 1, ndist1, dist1);
    named_matrix(f, "M2", 3, 3, K2);
    named_matrix(f, "D2", 1, ndist2, dist2);
    std::fclose(f);
    f = std::fopen(extrinsics_path, "w");
    if (!f) return ccalib::set_error(CCALIB_ERR_ARG, "cannot open %s for writing", extrinsics_path);
    std::fprintf(f, "%%YAML:1.0\n---\nimageSize: [ %d, %d ]\n", width, height);
    named_matrix(f, "R", 3, 3, R);
    named_matrix(f, "T", 3, 1, T);
    if (E) named_matrix(f, "E", 3, 3, E);
    if (F) named_matrix(f, "F", 3, 3, F);
    std::fclose(f);
    return CCALIB_OK;
}
