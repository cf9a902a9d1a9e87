// host_util.hpp -- error reporting shared by the host-side sources.
#pragma once
#include <cstdarg>
#include <cstddef>
#include <vector>

namespace ccalib {
// Records a thread-local message and returns `code` (so call sites can `return set_error(...)`).
int set_error(int code, const char* fmt, ...);
// Number of distinct libamdhip64.so files mapped into the process (two = two HIP runtimes side by side), and a note
// for error messages that says so ("" when there is one copy).
int hip_runtime_copies(char* first_two, size_t cap);
const char* hip_runtime_note();

bool solve_lu(std::vector<double> A, std::vector<double>& b, int n);
void solve_sym_svd(const std::vector<double>& A, std::vector<double>& b, int n);
void eig_sym(std::vector<double> A, int n, std::vector<double>& w, std::vector<double>& V);
void lm_trial_step(int ns, int nf, const int* fidx, const double* U, const double* gs, const double* Ssum,
                   const double* bsum, double lambda, int solve_method, const double* prev, double* shared, double* dsh);
}  // namespace ccalib
