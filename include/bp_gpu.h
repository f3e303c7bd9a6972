/* bp_gpu.h -- C ABI of libbp_gpu.so: the MI355X (gfx950) implementation of the
 * data-parallel hot path of the l-adic/bulletproofs crate (multi-scalar
 * multiplication and the inner-product-argument vector kernels).
 *
 * The reference has no FFI of its own: the path is reached by static generic
 * dispatch into arkworks (G::msm_unchecked, G::normalize_batch, Field ops).
 * Each entry point below names the reference call site(s) (file:line under the
 * reference root) it stands in for.
 *
 * Conventions
 *  - Field elements: 4 little-endian u64 limbs holding a*2^256 mod p, i.e. the
 *    in-memory value of ark_ff::Fp<MontBackend<_,4>> (`Fp.0.0`).  "Fr" is the
 *    curve's scalar field, "Fq" its base field.
 *  - Affine points: structure of arrays -- x plane (n*4 u64), y plane (n*4 u64),
 *    infinity plane (n bytes, non-zero = identity; x,y then ignored on input and
 *    written as 0 on output).  A NULL infinity plane on input means "none".
 *  - Group results ("xyz"): 12 u64 = Jacobian X,Y,Z (x = X/Z^2, y = Y/Z^3), the
 *    layout of ark_ec::short_weierstrass::Projective.  Results are normalised:
 *    Z = 1 (Montgomery one), or (1,1,0) for the identity.
 *  - Every function returns BP_OK (0) or a negative BP_E_* code; no exception
 *    crosses the boundary.  bp_last_error() gives the text of the last failure
 *    on a context.
 *  - Threading: a bp_ctx serialises its calls with an internal mutex, so any
 *    host thread may call into any context; use one context per thread (or per
 *    concurrent rayon task) for overlap.  Host buffers are only read/written
 *    during the call.  "_dev" variants take device pointers on the context's
 *    GPU and enqueue on the context's stream; they return after the result
 *    needed on the host (if any) has arrived, otherwise asynchronously --
 *    call bp_sync() before reading device outputs from another stream.
 *  - The library never falls back to a CPU implementation: if no gfx950 device
 *    is usable bp_init fails with BP_E_HIP.
 */
#ifndef BP_GPU_H
#define BP_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bp_ctx bp_ctx;
typedef struct bp_bases bp_bases;
typedef struct bp_ipa bp_ipa;
typedef struct bp_batch bp_batch;

enum { BP_SECP256K1 = 0, BP_VESTA = 1 };

enum {
  BP_OK = 0,
  BP_E_ARG = -1,    /* null pointer, bad size, bad enum */
  BP_E_CURVE = -2,  /* unsupported curve id */
  BP_E_HIP = -3,    /* HIP runtime error / no usable device */
  BP_E_OOM = -4,    /* device or host allocation failed */
  BP_E_STATE = -5   /* call out of order on a session object */
};

/* vector op selectors for bp_vec_op: src/vector_ops.rs:5-125 */
enum { BP_VEC_ADD = 0, BP_VEC_SUB = 1, BP_VEC_HADAMARD = 2, BP_VEC_SCALE = 3, BP_VEC_AXPY = 4 };

/* IPA variants: which half pairs with which challenge differs between
 * src/ipa/mod.rs:33-95 and src/poly_commit/mod.rs:69-105 */
enum { BP_IPA_BULLETPROOF = 0, BP_IPA_POLYCOMMIT = 1 };

/* ---- context ------------------------------------------------------------ */

/* One context = one GPU (HIP device ordinal) + one stream + cached workspace. */
int bp_init(int device, bp_ctx** out);
void bp_shutdown(bp_ctx* ctx);
const char* bp_last_error(bp_ctx* ctx);
int bp_sync(bp_ctx* ctx);
/* Window width override for MSM (0 = choose from n). */
int bp_set_msm_window(bp_ctx* ctx, int c);

/* ---- multi-scalar multiplication ----------------------------------------- */

/* sum_i scalars[i] * bases[i].  Stands in for G::msm_unchecked(&bases,&scalars):
 * src/msm.rs:84-87 (Msm::execute), src/ipa/mod.rs:54,70,206,
 * src/ipa/types.rs:81,82,105, src/range/mod.rs:77,78,228,
 * src/range/aggregate.rs:81,82,226,259, src/circuit/mod.rs:60,65,70,
 * src/poly_commit/mod.rs:78,83,111,159, src/poly_commit/types.rs:141.
 * Identity bases, zero scalars and repeated bases are allowed; n may be 0. */
int bp_msm(bp_ctx* ctx, int curve, const uint64_t* bases_x, const uint64_t* bases_y,
           const uint8_t* bases_inf, const uint64_t* scalars, size_t n, uint64_t out_xyz[12]);

/* As bp_msm with the bases as an array of structures: n records of BP_AOS_STRIDE bytes
 * {x: 4 u64, y: 4 u64, infinity: 1 byte (non-zero = identity), 7 bytes of padding}, the in-memory
 * value of ark_ec::short_weierstrass::Affine<P> over a 4-limb base field as rustc lays it out
 * today (a shim asserts size_of::<Affine<P>>() == 72 and the three offset_of! values once).  The
 * slice `&[G]` that Msm::execute hands to msm_unchecked (src/msm.rs:79-87) is passed as it is; the
 * planes are separated on the device after one contiguous upload. */
#define BP_AOS_STRIDE 72
int bp_msm_aos(bp_ctx* ctx, int curve, const void* bases, const uint64_t* scalars, size_t n,
               uint64_t out_xyz[12]);

/* Same with every input already in device memory (scalars: n*4 u64). */
int bp_msm_dev(bp_ctx* ctx, int curve, const uint64_t* d_x, const uint64_t* d_y,
               const uint8_t* d_inf, const uint64_t* d_scalars, size_t n, uint64_t out_xyz[12]);

/* Two-part MSM without concatenation: sum over part 0 then part 1.  The shape of
 * the L/R cross terms (src/ipa/mod.rs:44-54: bases g_hi||h_lo, scalars a_lo||b_hi). */
int bp_msm2_dev(bp_ctx* ctx, int curve, const uint64_t* d_x0, const uint64_t* d_y0,
                const uint8_t* d_inf0, const uint64_t* d_s0, size_t n0, const uint64_t* d_x1,
                const uint64_t* d_y1, const uint8_t* d_inf1, const uint64_t* d_s1, size_t n1,
                uint64_t out_xyz[12]);

/* Resident bases: the CRS (ipa::types::CRS src/ipa/types.rs:10-14,
 * poly_commit CRS) is static, so it is uploaded once. */
int bp_bases_register(bp_ctx* ctx, int curve, const uint64_t* x, const uint64_t* y,
                      const uint8_t* inf, size_t n, bp_bases** out);
/* bp_bases_register from an array of BP_AOS_STRIDE-byte records (see bp_msm_aos): `crs.gs`,
 * `crs.hs` (src/ipa/types.rs:10-14) handed over as the Vec<G> they are. */
int bp_bases_register_aos(bp_ctx* ctx, int curve, const void* bases, size_t n, bp_bases** out);
void bp_bases_free(bp_bases* b);
size_t bp_bases_len(const bp_bases* b);
/* Optional: builds window tables for a registered base set (2^(8w) * P for w < 32: 2 KiB per
 * base, one lane per base, a few ms).  MSMs and IPA sessions over these bases with at most
 * 2^17 points then read the multiples from the tables, and the host has no doublings left to
 * do when it combines the window sums.  Results are unchanged.  Up to 17408 points per pass the
 * launches are those of the path without tables (every digit window keeps its own bucket set;
 * only the points that are read differ) and the window sums are simply added: measured on
 * MI355X an IPA round over 2 * 4096 generators takes 0.220 ms against 0.243 ms, and the range
 * proof drivers in bulletproofs_amd/host build the tables for their CRS.  Beyond that size all
 * digit windows share one bucket set per sum (128 buckets with 32 times the entries each), a
 * longer dependent chain on the GPU that only pays where the host core is slow.  No counterpart
 * in the reference (arkworks' msm_unchecked takes the bases fresh on every call). */
int bp_bases_precompute(bp_bases* b);
/* MSM over bases[offset .. offset+n) with host scalars (crs.gs[0..n] slices:
 * src/ipa/types.rs:81-82). */
int bp_msm_resident(bp_ctx* ctx, const bp_bases* bases, size_t offset, size_t n,
                    const uint64_t* scalars, uint64_t out_xyz[12]);
/* sum_{i<n} scalars[i] * bases[offset + i] + sum_{j<n_extra} extra_scalars[j] * extra_j: the
 * deferred MSM of a verifier (Msm::execute, src/msm.rs:84-87) whose bases are mostly CRS points:
 * those stay on the device, only the scalars and the few other points (proof elements,
 * commitments) are uploaded.  extra_inf may be NULL. */
int bp_msm_resident_plus(bp_ctx* ctx, const bp_bases* bases, size_t offset, size_t n, const uint64_t* scalars,
                         const uint64_t* extra_x, const uint64_t* extra_y, const uint8_t* extra_inf,
                         const uint64_t* extra_scalars, size_t n_extra, uint64_t out_xyz[12]);
/* Two sums over the same n registered points in one pass: out_a = sum scalars_a[i] * P_i,
 * out_b = sum scalars_b[i] * P_i.  The A and S commitments of the range proofs, which the
 * reference computes side by side under rayon::join (src/range/aggregate.rs:80-83,
 * src/range/mod.rs:76-79), are one launch sequence instead of two. */
int bp_msm_resident_pair(bp_ctx* ctx, const bp_bases* bases, size_t offset, size_t n,
                         const uint64_t* scalars_a, const uint64_t* scalars_b, uint64_t out_a_xyz[12],
                         uint64_t out_b_xyz[12]);
/* MSM over bases[offset .. offset+2^k) with the s-vector of k challenges as scalars,
 * built on the device and never copied: `G::msm_unchecked(&crs.gs, &h_poly.coeffs())`
 * (src/poly_commit/mod.rs:109-111,152-159).  ch_inv[j] = ch[j]^-1; k <= 28. */
int bp_msm_resident_s_vector(bp_ctx* ctx, const bp_bases* bases, size_t offset, const uint64_t* ch,
                             const uint64_t* ch_inv, size_t k, uint64_t out_xyz[12]);

/* Per-window partial sums of an MSM, left on the device for a multi-GPU combine:
 * writes *n_windows and *window_bits, and n_windows XYZZ points (4*4 u64 each)
 * to d_windows (device, capacity >= 64 points).  bp_msm_combine_windows_dev sums
 * `n_parts` such arrays (laid out part-major, e.g. an all-gather result) window
 * by window on the device and finishes sum_w 2^(c*w) W_w. */
int bp_msm_windows_dev(bp_ctx* ctx, int curve, const uint64_t* d_x, const uint64_t* d_y,
                       const uint8_t* d_inf, const uint64_t* d_scalars, size_t n,
                       uint64_t* d_windows, int* n_windows, int* window_bits);
int bp_msm_combine_windows_dev(bp_ctx* ctx, int curve, const uint64_t* d_windows, int n_parts,
                               int n_windows, int window_bits, uint64_t out_xyz[12]);
/* One MSM split over several GPUs (one context per GPU, one process or thread per context): the
 * points are cut into n_shards contiguous ranges whose sizes differ by at most one, every shard
 * computes its window sums with the SAME window width, the caller exchanges the n_windows * 128
 * bytes per shard (an all-gather over RCCL, or peer copies) and any shard finishes with
 * bp_msm_combine_windows_dev.  The sum over ranges of Msm::execute, src/msm.rs:84-87; the
 * reference itself runs one sum on one host.
 *  - bp_msm_shard_window_bits: the common width, a function of the total and the shard count only
 *    (the width the library would pick for the largest shard), so every rank computes it alone.
 *    Needs no context and no GPU.  Returns the width (>= 2) or BP_E_ARG.
 *  - bp_msm_shard_windows_dev: window sums of one shard whose inputs are in device memory, with
 *    that width; n may be 0 (all sums are the identity).
 *  - bp_msm_shard_windows_resident: the same over a range of a base set registered on this
 *    context's GPU (a CRS is registered shard by shard, each GPU holding its range), host scalars. */
int bp_msm_shard_window_bits(size_t n_total, int n_shards);
int bp_msm_shard_windows_dev(bp_ctx* ctx, int curve, const uint64_t* d_x, const uint64_t* d_y,
                             const uint8_t* d_inf, const uint64_t* d_scalars, size_t n, int window_bits,
                             uint64_t* d_windows, int* n_windows);
int bp_msm_shard_windows_resident(bp_ctx* ctx, const bp_bases* bases, size_t offset, size_t n,
                                  const uint64_t* scalars, int window_bits, uint64_t* d_windows,
                                  int* n_windows);

/* ---- batch verification over a registered base set ---------------------------- */

/* verify_batch_aux (src/msm.rs:91-115) for deferred MSMs whose bases are mostly points of one
 * registered base set (a CRS): the accumulator holds one scalar per registered point on the
 * device, every proof's scalars are added with the weight alpha^k there (Msm::scale src/msm.rs:50-54
 * and Msm::batch :73-77 as one multiply-add per element), and the combined sum is one MSM whose
 * bases never leave the GPU.  The few bases outside the set (proof elements, commitments) are merged
 * by the caller and passed to bp_batch_msm with their scalars.
 *  - bp_batch_begin: accumulator over bases[0 .. n), all zero.
 *  - bp_batch_axpy / _dev: acc[offset + i] += weight * scalars[i], i < count (host or device scalars).
 *  - bp_batch_msm: sum_i acc[i] * bases[i] + sum_j extra_scalars[j] * extra_j; the accumulator is
 *    left as it is (more proofs may be added and the sum taken again). */
int bp_batch_begin(bp_ctx* ctx, const bp_bases* bases, size_t n, bp_batch** out);
int bp_batch_axpy(bp_batch* b, size_t offset, const uint64_t* scalars, size_t count, const uint64_t weight[4]);
int bp_batch_axpy_dev(bp_batch* b, size_t offset, const uint64_t* d_scalars, size_t count,
                      const uint64_t weight[4]);
/* One aggregated range proof's share of a batch, built where it is added: acc[0, 2n + 1) +=
 * weight * (scalars of gs | hs | u as bp_range_agg_verify_msm builds them), n = m * n_bits = 2^k.
 * The accumulator must cover the CRS layout gs | hs | u (bp_batch_begin with n >= 2 * m * n_bits + 1).
 * The proof's other terms (A, S, T1, T2, V_j, L_j, R_j, g, h) go to bp_batch_msm as extras, scaled by
 * the caller (verify_batch_aux, src/msm.rs:91-115). */
int bp_batch_axpy_range_agg(bp_batch* b, const uint64_t weight[4], const uint64_t* challenges,
                            const uint64_t* challenge_invs, size_t k, const uint64_t a[4],
                            const uint64_t bb[4], const uint64_t y_inv[4], const uint64_t z[4], size_t m,
                            unsigned n_bits, const uint64_t u_scalar[4]);
/* One inner-product argument's share of a batch, built where it is added: acc[g_offset, +2^k) +=
 * weight * a * s, acc[h_offset, +2^k) += weight * b * s^-1 (the s-vectors of the challenges and of
 * their inverses), acc[u_index] += weight * u_scalar: the CRS part of ipa::verify_aux's accumulator
 * (src/ipa/mod.rs:125-166) under verify_batch_aux's scaling (src/msm.rs:91-115).  Enqueue only. */
int bp_batch_axpy_ipa(bp_batch* b, const uint64_t weight[4], size_t g_offset, size_t h_offset,
                      size_t u_index, const uint64_t* challenges, const uint64_t* challenge_invs,
                      size_t k, const uint64_t a[4], const uint64_t bb[4], const uint64_t u_scalar[4]);
/* One polynomial-commitment opening's share of a batch: acc[offset, +2^k) += weight * s, s the
 * coefficients of h_poly (HPoly::coeffs, src/poly_commit/mod.rs:194-219) for the opening's k
 * challenges -- the scalars of the CRS in poly_commit::verify's <h_poly.coeffs(), gs> (:152-159),
 * with weight = alpha^i * a of the opening, under verify_batch_aux's scaling (src/msm.rs:91-115).
 * Built where it is added; enqueue only. */
int bp_batch_axpy_s_vector(bp_batch* b, const uint64_t weight[4], size_t offset, const uint64_t* challenges,
                           const uint64_t* challenge_invs, size_t k);
int bp_batch_msm(bp_batch* b, const uint64_t* extra_x, const uint64_t* extra_y, const uint8_t* extra_inf,
                 const uint64_t* extra_scalars, size_t n_extra, uint64_t out_xyz[12]);
void bp_batch_free(bp_batch* b);

/* ---- field-vector kernels (host buffers) --------------------------------- */

/* sum_i a[i]*b[i] in Fr: inner_product, src/vector_ops.rs:159-171. */
int bp_inner_product(bp_ctx* ctx, int curve, const uint64_t* a, const uint64_t* b, size_t n,
                     uint64_t out[4]);
/* sum_i a[i]: sum, src/vector_ops.rs:173-179. */
int bp_sum(bp_ctx* ctx, int curve, const uint64_t* a, size_t n, uint64_t out[4]);
/* out[i] = l[i]*x + r[i]*y: fold_scalars, src/ipa/mod.rs:229-234 and
 * src/poly_commit/mod.rs:24-29. */
int bp_fold_scalars(bp_ctx* ctx, int curve, const uint64_t* l, const uint64_t* r, size_t h,
                    const uint64_t x[4], const uint64_t y[4], uint64_t* out);
/* lane-wise add/sub/hadamard (a op b), scale (s*a), axpy (a + s*b):
 * VectorOps adaptors src/vector_ops.rs:5-125; Msm::scale src/msm.rs:50-54;
 * the alpha^i-weighted sum of fold_todos_witness src/poly_commit/mod.rs:228-244. */
int bp_vec_op(bp_ctx* ctx, int curve, int op, const uint64_t* a, const uint64_t* b,
              const uint64_t s[4], size_t n, uint64_t* out);
/* out[i] = prod_j (bit j of i ? ch[k-1-j] : ch_inv[k-1-j]), i < 2^k:
 * src/ipa/mod.rs:134-148, HPoly::coeffs src/poly_commit/mod.rs:194-219.
 * Swapping ch and ch_inv gives the inverse vector of src/ipa/mod.rs:150-154. */
int bp_s_vector(bp_ctx* ctx, int curve, const uint64_t* ch, const uint64_t* ch_inv, size_t k,
                uint64_t* out);
/* out = sum_{i<m} weights[i] * s_vector(ch + i*k, ch_inv + i*k), 2^k elements: the witness of
 * fold_todos_witness src/poly_commit/mod.rs:228-244 (weights = alpha^i, one h_poly per deferred
 * check) accumulated on the device; only the sum crosses to the host.  ch, ch_inv: m x k
 * scalars, weights: m scalars, m >= 1. */
int bp_fold_s_vectors(bp_ctx* ctx, int curve, const uint64_t* ch, const uint64_t* ch_inv,
                      const uint64_t* weights, size_t k, size_t m, uint64_t* out);
/* Vector polynomials of the aggregated range proof, src/range/aggregate.rs:50-64 and :88-127, built
 * on the device from the m values (bits taken low to high, n_bits <= 64 each; n = m * n_bits), the
 * blinders s_l, s_r (n scalars each) and the challenges y, z:
 *   l0 = a_L - z,  l1 = s_L,  r0 = y^k o (a_R + z) + z^(2+j) 2^i,  r1 = y^k o s_R
 * with a_L the bits and a_R = a_L - 1.  Returns t1 = <l0, r1> + <l1, r0> and t2 = <l1, r1>
 * (VectorPolynomial::inner_product, src/range/types.rs:98-117); the four vectors stay in the
 * context for bp_range_agg_eval. */
int bp_range_agg_poly(bp_ctx* ctx, int curve, const uint64_t* values, size_t m, unsigned n_bits,
                      const uint64_t* s_l, const uint64_t* s_r, const uint64_t y[4], const uint64_t z[4],
                      uint64_t t1[4], uint64_t t2[4]);
/* l = l0 + x l1, r = r0 + x r1 (VectorPolynomial::evaluate, src/range/types.rs:119-129, as called
 * at src/range/aggregate.rs:143-144) and t_hat = <l, r> (:146), from the vectors the last
 * bp_range_agg_poly of this context left; n must be that call's m * n_bits (BP_E_STATE otherwise). */
int bp_range_agg_eval(bp_ctx* ctx, int curve, const uint64_t x[4], size_t n, uint64_t* l, uint64_t* r,
                      uint64_t t_hat[4]);
/* ---- batches of aggregated range proofs -----------------------------------
 * B proofs of one shape (m values of n_bits bits, N = m * n_bits a power of two, 2 <= N <= 4096,
 * B <= 128) proved together: each phase of range::aggregate::prove (src/range/aggregate.rs:23-170) and
 * each round of the inner-product argument (src/ipa/mod.rs:38-97) is ONE chain of launches whose
 * grids serve all B proofs, so a host thread pays the launch and synchronisation latency of a
 * proof once per batch.  Proof b's messages are those of the single-proof calls on its inputs.
 * The base set holds gs[0..N) | hs[0..N) | u | g | h (2N + 3 points) with window tables
 * (bp_bases_precompute).  Every array argument is B rows, row b for proof b; points come back as
 * normalised Jacobian triples (12 u64, Z = 1, the identity as (1, 1, 0)).  The calls follow the
 * order of the protocol: commit, poly, tcommit, eval, ipa_begin, then round_lr / round_fold until
 * bp_range_batch_ipa_len is 1, then finish; the object can then start again at commit. */
typedef struct bp_range_batch bp_range_batch;
int bp_range_batch_begin(bp_ctx* ctx, const bp_bases* bases, size_t batch, size_t m, unsigned n_bits,
                         bp_range_batch** out);
void bp_range_batch_free(bp_range_batch* rb);
size_t bp_range_batch_size(const bp_range_batch* rb);
/* A_b = <a_L, gs> + <a_R, hs> + alpha_b h,  S_b = <s_L, gs> + <s_R, hs> + rho_b h
 * (src/range/aggregate.rs:50-83).  values: B x m; alpha, rho: B scalars; out_as_xyz: B x (A, S).
 * Blinders: s_lr = B x (s_L | s_R) (2N scalars per proof) from the caller's generator, or NULL to
 * draw them on the device from B SplitMix64 streams -- element i of proof b is draw number
 * first_draw + i of the stream with state seeds[b], a draw being four consecutive outputs as
 * little-endian limbs (the generator of this library's seeded drivers; not for production
 * secrets).  BP_E_STATE when a device draw was not below the group order (the rejection
 * sampling of a host generator would have shifted the later draws): pass s_lr then. */
int bp_range_batch_commit(bp_range_batch* rb, const uint64_t* values, const uint64_t* s_lr,
                          const uint64_t* seeds, unsigned first_draw, const uint64_t* alpha,
                          const uint64_t* rho, uint64_t* out_as_xyz);
/* bp_range_agg_poly for every proof of the batch (src/range/aggregate.rs:88-127): y, y_inv, z:
 * B scalars each; out_t1t2: B x (t1, t2). */
int bp_range_batch_poly(bp_range_batch* rb, const uint64_t* y, const uint64_t* y_inv, const uint64_t* z,
                        uint64_t* out_t1t2);
/* T1_b = t1 g + tao1 h, T2_b = t2 g + tao2 h (src/range/aggregate.rs:126-127).  t_scalars: B x
 * (t1, tao1, t2, tao2); out_tt_xyz: B x (T1, T2). */
int bp_range_batch_tcommit(bp_range_batch* rb, const uint64_t* t_scalars, uint64_t* out_tt_xyz);
/* bp_range_agg_eval for every proof (src/range/aggregate.rs:143-146): x: B scalars; out_t_hat: B
 * scalars.  l and r stay on the device as the witnesses of the B inner-product arguments. */
int bp_range_batch_eval(bp_range_batch* rb, const uint64_t* x, uint64_t* out_t_hat);
/* Starts the B arguments over gs, hs' = y^-i hs, u' = u_factor_b u (src/range/aggregate.rs:148-166,
 * src/ipa/extended.rs:15-35); u_factor: B scalars. */
int bp_range_batch_ipa_begin(bp_range_batch* rb, const uint64_t* u_factor);
size_t bp_range_batch_ipa_len(const bp_range_batch* rb);
/* bp_ipa_round_lr / bp_ipa_round_fold for every proof: out_lr_xyz: B x (L, R) with the u' terms
 * included (src/ipa/mod.rs:43-71); x, x_inv: B challenges and their inverses (:77-92). */
int bp_range_batch_round_lr(bp_range_batch* rb, uint64_t* out_lr_xyz);
int bp_range_batch_round_fold(bp_range_batch* rb, const uint64_t* x, const uint64_t* x_inv);
/* the final a, b of every proof (src/ipa/mod.rs:94-97): out_ab: B x (a, b) */
int bp_range_batch_finish(bp_range_batch* rb, uint64_t* out_ab);

/* Scalars of gs_i and hs_i (i = j*n_bits + bit, n = m*n_bits = 2^k) in the single deferred MSM that
 * the checks of range::aggregate::verify_aux add up to by linearity:
 *   out_g[i] = a*s_i + z                                    (src/ipa/mod.rs:134-160; src/range/aggregate.rs:239-245)
 *   out_h[i] = y^-i * (b*s_i^-1 - z^(2+j) 2^bit) - z        (src/ipa/mod.rs:150-166; src/range/aggregate.rs:246-261,272)
 * with s the s-vector of the k challenges, a, b the proof's final scalars.  Built on the device. */
int bp_range_agg_verify_scalars(bp_ctx* ctx, int curve, const uint64_t* challenges, const uint64_t* challenge_invs,
                                size_t k, const uint64_t a[4], const uint64_t b[4], const uint64_t y_inv[4],
                                const uint64_t z[4], size_t m, unsigned n_bits, uint64_t* out_g, uint64_t* out_h);
/* The whole deferred check of the aggregated range verifier in one call: the scalars of
 * bp_range_agg_verify_scalars are built on the device and stay there as the scalars of
 * bases[0, n) (gs) and bases[n, 2n) (hs), u = bases[2n] takes u_scalar, the n_extra other terms
 * (A, S, T1, T2, V_j, L_j, R_j, g, h: src/range/aggregate.rs:217-227,259-264,278-279) come from the
 * host, and out_xyz is the MSM over all of them: the proof is valid iff it is the identity.  Saves
 * the round trip of 2n scalars to the host and back that the two-step form pays. */
int bp_range_agg_verify_msm(bp_ctx* ctx, const bp_bases* bases, const uint64_t* challenges,
                            const uint64_t* challenge_invs, size_t k, const uint64_t a[4],
                            const uint64_t b[4], const uint64_t y_inv[4], const uint64_t z[4], size_t m,
                            unsigned n_bits, const uint64_t u_scalar[4], const uint64_t* extra_x,
                            const uint64_t* extra_y, const uint8_t* extra_inf,
                            const uint64_t* extra_scalars, size_t n_extra, uint64_t out_xyz[12]);
/* The deferred check of the inner-product verifier (ipa::verify_aux, src/ipa/mod.rs:102-193) in one
 * call over a registered CRS: gs = bases[g_offset, +2^k) take a * s_i, hs = bases[h_offset, +2^k) take
 * b * s_i^-1 (the s-vector of the challenges and of their inverses, :134-166, built and scaled on the
 * device), u = bases[u_index] takes u_scalar (a * b, :125), the n_extra other terms (P, L_j, R_j with
 * -1, -x_j^2, -x_j^-2, :170-190) come from the host; out_xyz is the MSM over all of them, the
 * identity iff the proof is valid.  The three index ranges must not overlap. */
int bp_ipa_verify_msm(bp_ctx* ctx, const bp_bases* bases, size_t g_offset, size_t h_offset,
                      size_t u_index, const uint64_t* challenges, const uint64_t* challenge_invs,
                      size_t k, const uint64_t a[4], const uint64_t b[4], const uint64_t u_scalar[4],
                      const uint64_t* extra_x, const uint64_t* extra_y, const uint8_t* extra_inf,
                      const uint64_t* extra_scalars, size_t n_extra, uint64_t out_xyz[12]);
/* out[i] = x^i, i < n: powers_of_x src/poly_commit/mod.rs:16-22, y_vec
 * src/range/aggregate.rs:88-91, hs' scalars src/range/utils.rs:22-29. */
int bp_powers(bp_ctx* ctx, int curve, const uint64_t x[4], size_t n, uint64_t* out);
/* In-place batch inverse, zeros stay zero: ark_ff::batch_inversion at
 * src/ipa/mod.rs:130,152, src/poly_commit/mod.rs:197.  field: 0 = Fr, 1 = Fq. */
int bp_batch_inverse(bp_ctx* ctx, int curve, int field, uint64_t* v, size_t n);

/* ---- curve-vector kernels (host buffers) --------------------------------- */

/* out[i] = x*L[i] + y*R[i], affine: fold_generators src/ipa/mod.rs:215-227,
 * src/poly_commit/mod.rs:31-43.  x,y in Fr. */
int bp_fold_generators(bp_ctx* ctx, int curve, const uint64_t* lx, const uint64_t* ly,
                       const uint8_t* linf, const uint64_t* rx, const uint64_t* ry,
                       const uint8_t* rinf, size_t h, const uint64_t x[4], const uint64_t y[4],
                       uint64_t* out_x, uint64_t* out_y, uint8_t* out_inf);
/* out[i] = s[i]*P[i], affine: the hs' construction h_i * y^-i at
 * src/range/mod.rs:136-141,208-214, src/range/aggregate.rs:148-153,231-237,
 * src/circuit/mod.rs:178-185,298-304. */
int bp_scale_points(bp_ctx* ctx, int curve, const uint64_t* px, const uint64_t* py,
                    const uint8_t* pinf, const uint64_t* s, size_t n, uint64_t* out_x,
                    uint64_t* out_y, uint8_t* out_inf);
/* Wire format of points: the compressed encoding the reference's proof and CRS types get from
 * ark-serialize 0.5 (CanonicalSerialize, Compress::Yes, behind the Encoding / NargDeserialize
 * derives: src/ipa/types.rs:48,140; prover_messages at src/ipa/mod.rs:75,97).  33 bytes per point on
 * both curves: the canonical x in 32 little-endian bytes, then a flag byte (bit 7: y > p - y, bit 6:
 * the identity, written with x = 0).  bp_points_compress writes n * 33 bytes.
 * bp_points_decompress recomputes y = sqrt(x^3 + b) on the device (the smaller root unless bit 7 is
 * set) and sets out_ok[i] = 0 where the bytes are no point (x >= p, both flags, no square root; x, y
 * are then 0).  The reference holds no byte vector of this encoding: it restates the published format. */
#define BP_POINT_WIRE_BYTES 33
int bp_points_compress(bp_ctx* ctx, int curve, const uint64_t* px, const uint64_t* py,
                       const uint8_t* pinf, size_t n, uint8_t* out);
int bp_points_decompress(bp_ctx* ctx, int curve, const uint8_t* in, size_t n, uint64_t* out_x,
                         uint64_t* out_y, uint8_t* out_inf, uint8_t* out_ok);
/* bp_bases_register for a CRS that arrives as n compressed points (n * 33 bytes): decoded straight
 * into the base set on the device.  BP_E_ARG (and no base set) if any element is no point; the
 * message of bp_last_error names the first one. */
int bp_bases_register_compressed(bp_ctx* ctx, int curve, const uint8_t* bytes, size_t n,
                                 bp_bases** out);
/* Scalars on the wire: the canonical integer in 32 little-endian bytes (CanonicalSerialize of a
 * scalar-field element, e.g. the a, b, tao_x, mu, t_hat of a proof or a coefficient vector read from
 * bytes).  bp_scalars_to_bytes writes n * 32 bytes; bp_scalars_from_bytes returns Montgomery limbs
 * and out_ok[i] = 0 (value 0) where the integer is not below the group order. */
int bp_scalars_to_bytes(bp_ctx* ctx, int curve, const uint64_t* v, size_t n, uint8_t* out);
int bp_scalars_from_bytes(bp_ctx* ctx, int curve, const uint8_t* in, size_t n, uint64_t* out,
                          uint8_t* out_ok);
/* Jacobian (n*12 u64) -> affine: G::normalize_batch at src/ipa/mod.rs:187,226,
 * src/range/aggregate.rs:223. */
int bp_normalize_batch(bp_ctx* ctx, int curve, const uint64_t* xyz, size_t n, uint64_t* out_x,
                       uint64_t* out_y, uint8_t* out_inf);

/* ---- device-pointer forms of the streaming kernels ----------------------- */
int bp_inner_product_dev(bp_ctx* ctx, int curve, const uint64_t* d_a, const uint64_t* d_b,
                         size_t n, uint64_t out[4]);
/* bp_inner_product_dev for callers that cannot allocate (code under HIP graph capture, serving
 * loops and harnesses that forbid hipMalloc): no counterpart in the reference.  Same kernels and
 * same value as bp_inner_product_dev, but the block partial sums live in a static device array
 * instead of the context's workspace, the result (4 u64) is written to the device pointer d_out,
 * and the call only enqueues on the context's stream: it calls no hipMalloc, hipFree,
 * pinned-memory or other allocating routine and does not synchronise (bp_sync, or stream order,
 * before reading d_out).  n <= BP_NOALLOC_MAX_N, larger sizes are refused with BP_E_ARG.  Each GPU
 * holds one copy of the array (1 MiB: 32 slices of 32 KiB) for the life of the process; a context
 * owns one slice from bp_init to bp_shutdown, so calls on different contexts of a GPU do not
 * interfere, and calls on one context are ordered by its lock and stream.  The 33rd context alive
 * on one GPU gets no slice: this entry point then returns BP_E_STATE on it (every other entry point
 * works).  The slice keeps the last call's partial sums; every call overwrites the ones it reads. */
#define BP_NOALLOC_MAX_N ((size_t)1 << 28)
int bp_inner_product_dev_noalloc(bp_ctx* ctx, int curve, const uint64_t* d_a, const uint64_t* d_b,
                                 size_t n, uint64_t* d_out);
int bp_fold_scalars_dev(bp_ctx* ctx, int curve, const uint64_t* d_l, const uint64_t* d_r,
                        size_t h, const uint64_t x[4], const uint64_t y[4], uint64_t* d_out);
int bp_fold_generators_dev(bp_ctx* ctx, int curve, const uint64_t* d_lx, const uint64_t* d_ly,
                           const uint8_t* d_linf, const uint64_t* d_rx, const uint64_t* d_ry,
                           const uint8_t* d_rinf, size_t h, const uint64_t x[4],
                           const uint64_t y[4], uint64_t* d_out_x, uint64_t* d_out_y,
                           uint8_t* d_out_inf);
int bp_s_vector_dev(bp_ctx* ctx, int curve, const uint64_t* ch, const uint64_t* ch_inv, size_t k,
                    uint64_t* d_out);

/* ---- IPA round session ---------------------------------------------------- */

/* Keeps G, H, a, b on the device across the `while n != 1` loop of ipa::prove
 * (src/ipa/mod.rs:33-95) or poly_commit::prove (src/poly_commit/mod.rs:69-105;
 * pass hs = NULL).  n must be a power of two >= 1.  The transcript stays with
 * the caller: round_lr returns the MSM parts of L and R and the two cross
 * inner products; the caller adds u*c (unless bp_ipa_set_u was called) and
 * h*l_j, derives the challenge and passes it (and its inverse) to round_fold.
 * The generator vectors are not folded round by round (src/ipa/mod.rs:80-83 is
 * carried as one coefficient per original generator); bp_ipa_read materialises
 * the folded generators on request. */
int bp_ipa_begin(bp_ctx* ctx, int curve, int variant, const uint64_t* gx, const uint64_t* gy,
                 const uint8_t* ginf, const uint64_t* hx, const uint64_t* hy, const uint8_t* hinf,
                 const uint64_t* a, const uint64_t* b, size_t n, bp_ipa** out);
/* As bp_ipa_begin with the generators taken from a registered base set, not copied:
 * G = bases[g_offset, g_offset+n), H = bases[h_offset, h_offset+n) (h_offset ignored
 * for BP_IPA_POLYCOMMIT).  Replaces the per-proof `crs.gs[0..n].to_vec()` /
 * `crs.hs[0..n].to_vec()` of src/ipa/mod.rs:29-30.  The base set must outlive the
 * session. */
int bp_ipa_begin_resident(bp_ctx* ctx, const bp_bases* bases, int variant, size_t g_offset,
                          size_t h_offset, const uint64_t* a, const uint64_t* b, size_t n,
                          bp_ipa** out);
/* As bp_ipa_begin_resident with b_i = x^i (i < n) built on the device: the
 * `powers_of_x(statement.x).take(n)` that src/poly_commit/mod.rs:65 collects before the round
 * loop, which the caller would otherwise compute, download and upload again (32 bytes per
 * coefficient each way). */
int bp_ipa_begin_resident_powers(bp_ctx* ctx, const bp_bases* bases, int variant, size_t g_offset,
                                 size_t h_offset, const uint64_t* a, const uint64_t x[4], size_t n,
                                 bp_ipa** out);
/* From now on the session behaves as if H_i had been scale[i] * H_i (n host scalars):
 * the hs' = y^-i * h_i that src/range/aggregate.rs:148-153 materialises with n scalar
 * multiplications (create_hs_prime, src/range/utils.rs:22-29) before ipa::extended::prove.
 * BP_IPA_BULLETPROOF sessions, before the first fold only (BP_E_STATE otherwise). */
int bp_ipa_scale_h(bp_ipa* s, const uint64_t* scale);
/* Sets the point u (affine x|y, 8 u64, not the identity; NULL clears it).  With u set,
 * bp_ipa_round_lr returns L = c_l*u + <..>, R = c_r*u + <..> complete
 * (src/ipa/mod.rs:43,59; src/poly_commit/mod.rs:80,85) in the same device pass. */
int bp_ipa_set_u(bp_ipa* s, const uint64_t u_xy[8]);
/* As bp_ipa_set_u with u = factor * bases[u_index], for sessions begun with
 * bp_ipa_begin_resident: the u' = x * crs.u of ipa::extended::prove
 * (src/ipa/extended.rs:29-33) without a scalar multiplication; u is then reached through
 * the base set's window tables when it has them. */
int bp_ipa_set_u_resident(bp_ipa* s, size_t u_index, const uint64_t factor[4]);
int bp_ipa_round_lr(bp_ipa* s, uint64_t l_xyz[12], uint64_t r_xyz[12], uint64_t c_l[4],
                    uint64_t c_r[4]);
int bp_ipa_round_fold(bp_ipa* s, const uint64_t x[4], const uint64_t x_inv[4]);
size_t bp_ipa_len(const bp_ipa* s);
/* Current vectors (length bp_ipa_len) copied to the host; any pointer may be NULL. */
int bp_ipa_read(bp_ipa* s, uint64_t* gx, uint64_t* gy, uint8_t* ginf, uint64_t* hx, uint64_t* hy,
                uint8_t* hinf, uint64_t* a, uint64_t* b);
void bp_ipa_free(bp_ipa* s);

/* ---- measurement helpers --------------------------------------------------- */

/* Issues `iters` rounds of 256 independent 32x32+64 multiply-accumulates per lane
 * on every CU and returns the measured rate in MAC/s: the denominator of the
 * integer roofline. */
int bp_bench_int_mad_peak(bp_ctx* ctx, int iters, double* mac_per_s);
/* Times one launch (after a warm-up launch) of an arithmetic micro-benchmark:
 * id 0 = 2*iters dependent Fq multiplies per lane (secp256k1), 1 = iters add+sub
 * pairs, 2 = iters XYZZ mixed additions, 3 = iters Jacobian doublings, 4 = as 0 on
 * Vesta's Fq.  256 (ids 0,1,4) or 128 lanes per block.
 * id 5 = scattered read of every 64-byte record of a 2^iters-record table exactly once
 * (the MSM point-gather pattern with a known byte count, 64 * 2^iters, for calibrating
 * memory counters); `blocks` is ignored, 10 <= iters <= 26.
 * id 6 = as 0 and id 7 = as 2 on the secp256k1 base field in its plain (non-Montgomery) form,
 * the representation the MSM pipeline uses internally; id 8 = self-check of that multiply against
 * its portable form over chains of `iters` products: *ms receives the number of lanes that
 * disagreed (0 = pass); id 9 = as 0 on the scalar field of secp256k1 (the multiply of the
 * streaming kernels), id 10 = 2*iters dual products a*b + c*d with one reduction on that field;
 * id 11 = `iters` independent affine additions per lane with one field inversion for all of them
 * (batch-affine accumulation: 6 products per addition, the running products parked in memory;
 * 128 lanes per block, operands made in registers), id 12 = self-check of that arithmetic on
 * multiples of the secp256k1 generator against the XYZZ formulas (*ms = lanes that disagreed,
 * 64 lanes per block, `iters` ignored). */
int bp_microbench(bp_ctx* ctx, int id, int blocks, int iters, float* ms);
/* The event markers behind bp_last_kernel_ms for MSMs (four per MSM, a few microseconds each on the
 * stream) can be turned off for latency-bound callers (the IPA rounds of a proof); on by default.
 * No counterpart in the reference. */
int bp_set_timing(bp_ctx* ctx, int on);
/* Duration in ms of the most recent launch of the named kernel group on this
 * context, measured with HIP events on the context's stream.
 * groups: 0 = MSM bucket accumulation, 1 = whole MSM device phase, 2 = last
 * streaming kernel (fold/dot/s-vector). */
int bp_last_kernel_ms(bp_ctx* ctx, int group, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* BP_GPU_H */
