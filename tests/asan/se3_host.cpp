// Host-only build of the library's SE(3) entry points (include/tcsfm.h: tcsfm_pose_to_matrix ... tcsfm_se3_inv) for the CPU
// sanitizer leg (make test-asan): the same header the HIP library compiles (csrc/se3_math.h, host path), g++ instead of hipcc,
// so that AddressSanitizer / UBSan can watch it.  Test infrastructure; the product builds these from csrc/tcsfm_api.hip.
#include "../../include/tcsfm.h"
#include "../../tightly_coupled_sfm_amd/csrc/se3_math.h"

extern "C" {
void tcsfm_pose_to_matrix(const double pose[6], double T[12]) { tc::pose_to_T(pose, T); }
void tcsfm_matrix_to_pose(const double T[12], double pose[6]) { tc::T_to_pose(T, pose); }
void tcsfm_se3_exp(const double xi[6], double T[12]) { tc::se3_exp(xi, T); }
void tcsfm_se3_log(const double T[12], double xi[6]) { tc::se3_log(T, xi); }
void tcsfm_se3_mul(const double A[12], const double B[12], double C[12]) { tc::se3_mul(A, B, C); }
void tcsfm_se3_inv(const double A[12], double B[12]) { tc::se3_inv(A, B); }
}
