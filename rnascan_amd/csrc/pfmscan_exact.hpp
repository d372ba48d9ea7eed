// pfmscan_exact.hpp -- the decision `structure score > threshold` (rnascan.py:310) taken the way the reference's own
// arithmetic takes it, for every thresholded structure path (k_profile / k_profile_fixed hits, k_struct_at, k_wide,
// k_library phase B, k_profile_lib).  Not installed.
//
// The reference scores a window as  score += nan_to_num(np.dot(profile[i + j, :], pssm[j, :]))  (rnascan.py:302-307): each
// row-dot is ROUNDED, then added.  np.dot's own order is BLAS-defined; the restatement the parity tests pin (goldens from
// the reference run here) takes it k-ascending with every product and every addition rounded separately.  The kernels do
// not: a row is one multiply and six FMAs, and with an all-finite PSSM the seven FMAs go straight into the window sum --
// same terms, same order, results within ~1e-14 (the contract for scores is 1e-6), but a different LAST bit.  For the
// scores that is all there is to say; for HIT POSITIONS it would mean that a window whose score sits within a rounding
// error of the threshold can fall on either side.  So every hits path decides in two steps:
//   1. fast score F as before;
//   2. only if |F - thr| <= band: the window is scored again in the rounded, k-ascending order (struct_window_rounded)
//      and THAT value is compared and reported.
// band (struct_band, host) bounds |F - rounded| rigorously: both are sums of the same <= 7 m products in which every
// term passes through at most 7 + m roundings, so each differs from the exact sum by at most gamma(8 m) A with
// A = sum |r_jk P_jk| <= max|r| sum |P_jk| over the finite cells; band = 24 m 2^-53 STRUCT_ROW_MAX sum |P_jk| (> 2 gamma(8 m) A).
// Profile entries are probabilities; the bound holds for any |entry| <= STRUCT_ROW_MAX = 1024.  Non-finite rows and cells
// take the same nan_to_num values in both orders (0, +-DBL_MAX) and drop out of the difference.  The band is ~1e-10
// score units at w = 12: the second step runs for one window in ~10^11.
#pragma once
#include <float.h>
#include <math.h>
#include <hip/hip_runtime.h>

namespace pfmscan {

constexpr double STRUCT_ROW_MAX = 1024.0;

// host: half-width of the re-score band of one structure PSSM [m][7] (row-major, any cells)
inline double struct_band(const double *pssm, int m)
{
    double s = 0.0;
    for (int i = 0; i < m * 7; ++i)
        if (std::isfinite(pssm[i])) s += std::fabs(pssm[i]);
    return 24.0 * (double)m * 0x1p-53 * STRUCT_ROW_MAX * s;
}

__device__ __forceinline__ double exact_nan_to_num(double d)       // numpy.nan_to_num defaults (rnascan.py:306)
{
    const double c = fmin(fmax(d, -DBL_MAX), DBL_MAX);
    return (d != d) ? 0.0 : c;
}

// The window's score in the rounded order: rows[j * 7 + k] = profile row j of the window (any address space, float or
// double storage), cell(j, k) = PSSM cell.  Rolled on purpose: this is the one-in-10^11 path.
template <typename ROW_T, typename CellF>
__device__ __forceinline__ double struct_window_rounded(const ROW_T *rows, int m, CellF cell)
{
#pragma clang fp contract(off)
    double score = 0.0;
#pragma unroll 1
    for (int j = 0; j < m; ++j) {
        double d = 0.0;
#pragma unroll 1
        for (int k = 0; k < 7; ++k) {
            const double prod = (double)rows[j * 7 + k] * cell(j, k);       // rounded product ...
            d = d + prod;                                                    // ... rounded sum: no FMA (contract off)
        }
        score = score + exact_nan_to_num(d);
    }
    return score;
}

// true when the fast score cannot decide by itself
__device__ __forceinline__ bool struct_near(double fast, double thr, double band) { return fabs(fast - thr) <= band; }

}  // namespace pfmscan
