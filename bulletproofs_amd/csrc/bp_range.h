// Launchers of the range-proof verifier kernels (bp_vec.hip), used by bp_api.hip.
#pragma once
#include "bp_internal.h"

namespace bp {

// Scalars of gs_i and hs_i in the verifier's deferred MSM of an aggregated range proof, i = j*n_bits + bit:
//   g[i] = a*ss[i] + z,   h[i] = yinv[i] * (b*ssi[i] - zpow[j]*pow2[bit]) - z
// with ss / ssi the s-vector and its inverse (src/ipa/mod.rs:134-166), yinv[i] = y^-i
// (src/range/aggregate.rs:231-237,272), zpow[j] = z^(2+j), pow2[bit] = 2^bit (:239-261).
hipError_t launch_range_verify_scalars(int curve, const u32* ss, const u32* ssi, const u32* yinv, const u32* zpow,
                                       const u32* pow2, unsigned n_bits, size_t n, FeArg a, FeArg b, FeArg z, u32* g,
                                       u32* h, hipStream_t st);

}  // namespace bp
