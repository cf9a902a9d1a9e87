// host_rodrigues.hpp -- cvRodrigues2 on the host (both directions).
#pragma once

namespace ccalib {

// vector -> matrix; dRdr27 (optional) receives dR/dr as three row-major 3x3 blocks
void host_rodrigues_v2m(const double r[3], double R[9], double* dRdr27);
// matrix -> vector (the matrix is first replaced by the nearest rotation, as OpenCV's SVD step does)
void host_rodrigues_m2v(const double R[9], double r[3]);

}  // namespace ccalib
