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

// ---- batches of aggregated range proofs (bp_vec.hip): one grid per phase serves `batch` proofs ----
// Per-proof rows of the table array (element offsets inside a row of `row` elements):
//   zpow: z^(2+j), j < m | ylo: y^i, i < lo | yhi: y^(lo i), i < hi | yilo, yihi: the same for y^-1
struct RangeBatchTabs {
  u32 row, zpow, ylo, yhi, yilo, yihi, lo, hi;
};
// out[b][i] = draw first_draw + i of the SplitMix64 stream seeded seeds[b] as a scalar (see k_rangeb_blinders);
// *rejected (cleared by the caller) is set when a draw was not below the modulus
hipError_t launch_ipab_gather_cols(const u32* cg, const u32* ch, u32 n0, u32 cur, int batch, u32* out, hipStream_t st);
hipError_t launch_ipab_rebase_pack(int curve, const u32* a, const u32* b, u32 n0, u32 cur, int batch, const u32* tabs,
                                   RangeBatchTabs tl, u32* a2, u32* b2, u32* cg, u32* ch, const u32* ux, const u32* uy, u32* px,
                                   u32* py, uint8_t* pinf, hipStream_t st);
hipError_t launch_rangeb_blinders(int curve, const uint64_t* seeds, u32 first_draw, u32 count, int batch, u32* out, u32* rejected,
                                  hipStream_t st);
// tabs[b] = the table row of proof b from yz[b] = (y, y^-1, z)
hipError_t launch_rangeb_tables(int curve, const u32* yz, RangeBatchTabs tl, u32 m, int batch, u32* tabs, hipStream_t st);
// sa[b] = a_L || a_R of proof b (2n elements per proof) from its m values
hipError_t launch_rangeb_commit_scalars(int curve, const uint64_t* values, u32 m, u32 n_bits, u32 n, int batch, u32* sa,
                                        hipStream_t st);
// launch_range_poly per proof: s_lr = s_L || s_R (2n per proof), zs: z per proof, l0/r0/r1: n per proof,
// partials: 3 * ceil(n/256) per proof, out: (t1, t2) per proof (may be host memory the device can write)
hipError_t launch_rangeb_poly(int curve, const uint64_t* values, u32 m, u32 n_bits, u32 n, int batch, const u32* tabs,
                              RangeBatchTabs tl, const u32* pow2, const u32* s_lr, const u32* zs, u32* l0, u32* r0, u32* r1,
                              u32* partials, u32* out, hipStream_t st);
// launch_range_eval per proof: xs: x per proof; l, r: n per proof; out: t_hat per proof (may be host memory)
hipError_t launch_rangeb_eval(int curve, const u32* l0, const u32* s_lr, const u32* r0, const u32* r1, u32 n, int batch,
                              const u32* xs, u32* l, u32* r, u32* partials, u32* out, hipStream_t st);
// generator coefficients of `batch` inner-product arguments over gs, y^-i hs: cg = 1, ch_i = y^-i
hipError_t launch_ipab_init(int curve, const u32* tabs, RangeBatchTabs tl, u32 n, int batch, u32* cg, u32* ch, hipStream_t st);
// launch_ipa_round_prep / launch_ipa_fold_prep for `batch` sessions (rows of n0 elements; dots: 4 elements per
// session: c_l, c_r, factor c_l, factor c_r; factors, xs, xis: one element per session)
hipError_t launch_ipab_round_prep(int curve, const u32* a, const u32* b, const u32* cg, const u32* ch, u32 n0, u32 half,
                                  int batch, const u32* factors, u32* eg, u32* eh, u32* dots, hipStream_t st);
hipError_t launch_ipab_fold_prep(int curve, const u32* a, const u32* b, u32* a2, u32* b2, u32* cg, u32* ch, u32 n0, u32 half,
                                 int batch, const u32* xs, const u32* xis, const u32* factors, u32* eg, u32* eh, u32* dots,
                                 hipStream_t st);
// the fold of two elements to one: out[b] = (a', b') of session b (may be host memory the device can write)
hipError_t launch_ipab_last_fold(int curve, const u32* a, const u32* b, u32 n0, int batch, const u32* xs, const u32* xis,
                                 u32* out, hipStream_t st);

}  // namespace bp
