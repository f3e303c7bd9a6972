/* bp_oracle.c -- CPU oracle (TEST INFRASTRUCTURE, never shipped or linked into
 * the product) for the Bulletproofs hot path: a plain-C restatement of the
 * arithmetic the reference reaches through arkworks.
 *
 * PARITY UNPINNED by reference byte vectors: the reference (l-adic/bulletproofs)
 * holds no golden vectors for this path and its arithmetic lives in the
 * un-vendored registry crates ark-ff / ark-ec 0.5.0 (Cargo.toml:7-8).  This file
 * restates the published algorithms (Montgomery multiplication; Jacobian
 * short-Weierstrass formulas; Pippenger's bucket method as VariableBaseMSM
 * uses it: windows in parallel, per-window buckets + running sum, windows
 * combined by repeated doubling) and is pinned by tests/test_oracle.py against
 * the big-integer Python oracle, the SEC2 generator multiples and the literal
 * expectations of src/vector_ops.rs:238-314.  All outputs are canonical, so
 * equality with the mathematics is equality with arkworks.
 *
 * Data format = the arkworks memory form: field element = 4 little-endian u64
 * limbs of a*2^256 mod p; affine point vectors as x plane, y plane, infinity
 * bytes; group results as 12 u64 (X, Y, Z Jacobian, normalised to Z = 1).
 *
 * Build: see oracle/Makefile (gcc -O3 -fopenmp -shared).
 */
#include "bp_oracle_arith.h"

/* ---- exported: vector kernels (src/vector_ops.rs, src/ipa/mod.rs:229-234) ------- */
int bpo_inner_product(int curve, const uint64_t* a, const uint64_t* b, size_t n, uint64_t out[4]) {
  CURVE_OR_FAIL(curve); const field_t* f = &CURVES[curve].fr;
  fe acc = {{0, 0, 0, 0}};                       /* src/vector_ops.rs:159-171: left fold */
  for (size_t i = 0; i < n; i++) { fe x, y; memcpy(&x, a + 4 * i, 32); memcpy(&y, b + 4 * i, 32); acc = fe_add(f, acc, fe_mul(f, x, y)); }
  memcpy(out, acc.l, 32); return 0;
}
int bpo_sum(int curve, const uint64_t* a, size_t n, uint64_t out[4]) {
  CURVE_OR_FAIL(curve); const field_t* f = &CURVES[curve].fr;
  fe acc = {{0, 0, 0, 0}};                       /* src/vector_ops.rs:173-179 */
  for (size_t i = 0; i < n; i++) { fe x; memcpy(&x, a + 4 * i, 32); acc = fe_add(f, acc, x); }
  memcpy(out, acc.l, 32); return 0;
}
int bpo_fold_scalars(int curve, const uint64_t* l, const uint64_t* r, size_t h, const uint64_t x[4], const uint64_t y[4], uint64_t* out, int threads) {
  CURVE_OR_FAIL(curve); const field_t* f = &CURVES[curve].fr;
  fe fx, fy; memcpy(&fx, x, 32); memcpy(&fy, y, 32);
  (void)threads;
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
  for (long i = 0; i < (long)h; i++) {            /* src/ipa/mod.rs:229-234 */
    fe a, b; memcpy(&a, l + 4 * i, 32); memcpy(&b, r + 4 * i, 32);
    fe o = fe_add(f, fe_mul(f, a, fx), fe_mul(f, b, fy));
    memcpy(out + 4 * i, o.l, 32);
  }
  return 0;
}
int bpo_s_vector(int curve, const uint64_t* ch, const uint64_t* chi, size_t k, uint64_t* out, int threads) {
  CURVE_OR_FAIL(curve); const field_t* f = &CURVES[curve].fr;
  (void)threads;
  size_t n = (size_t)1 << k;
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
  for (long i = 0; i < (long)n; i++) {            /* src/ipa/mod.rs:134-148 */
    fe acc = f->one;
    for (size_t j = 0; j < k; j++) {
      size_t idx = k - j - 1; fe v;
      memcpy(&v, (((size_t)i >> j) & 1) ? ch + 4 * idx : chi + 4 * idx, 32);
      acc = fe_mul(f, acc, v);
    }
    memcpy(out + 4 * i, acc.l, 32);
  }
  return 0;
}

/* ---- exported: curve kernels ------------------------------------------------------- */
/* out_i = x*L_i + y*R_i -> affine (src/ipa/mod.rs:215-227) */
int bpo_fold_generators(int curve, const uint64_t* lx, const uint64_t* ly, const uint8_t* linf, const uint64_t* rx,
                        const uint64_t* ry, const uint8_t* rinf, size_t h, const uint64_t x[4], const uint64_t y[4],
                        uint64_t* ox, uint64_t* oy, uint8_t* oinf, int threads) {
  CURVE_OR_FAIL(curve); const field_t* fq = &CURVES[curve].fq; const field_t* fr = &CURVES[curve].fr;
  fe fx, fy; memcpy(&fx, x, 32); memcpy(&fy, y, 32);
  fe kx = fe_from_mont(fr, fx), ky = fe_from_mont(fr, fy);
  (void)threads;
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 16)
  for (long i = 0; i < (long)h; i++) {
    fe Lx, Ly, Rx, Ry; memcpy(&Lx, lx + 4 * i, 32); memcpy(&Ly, ly + 4 * i, 32); memcpy(&Rx, rx + 4 * i, 32); memcpy(&Ry, ry + 4 * i, 32);
    jac a = jac_scalar_mul(fq, kx.l, &Lx, &Ly, linf && linf[i]);
    jac b = jac_scalar_mul(fq, ky.l, &Rx, &Ry, rinf && rinf[i]);
    jac s = jac_add(fq, &a, &b);
    fe X, Y; int inf = jac_to_affine(fq, &s, &X, &Y);
    memcpy(ox + 4 * i, X.l, 32); memcpy(oy + 4 * i, Y.l, 32); if (oinf) oinf[i] = (uint8_t)inf;
  }
  return 0;
}
/* out_i = s_i * P_i -> affine (src/range/aggregate.rs:148-153) */
int bpo_scale_points(int curve, const uint64_t* px, const uint64_t* py, const uint8_t* pinf, const uint64_t* s, size_t n,
                     uint64_t* ox, uint64_t* oy, uint8_t* oinf, int threads) {
  CURVE_OR_FAIL(curve); const field_t* fq = &CURVES[curve].fq; const field_t* fr = &CURVES[curve].fr;
  (void)threads;
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 16)
  for (long i = 0; i < (long)n; i++) {
    fe Px, Py, k; memcpy(&Px, px + 4 * i, 32); memcpy(&Py, py + 4 * i, 32); memcpy(&k, s + 4 * i, 32);
    k = fe_from_mont(fr, k);
    jac a = jac_scalar_mul(fq, k.l, &Px, &Py, pinf && pinf[i]);
    fe X, Y; int inf = jac_to_affine(fq, &a, &X, &Y);
    memcpy(ox + 4 * i, X.l, 32); memcpy(oy + 4 * i, Y.l, 32); if (oinf) oinf[i] = (uint8_t)inf;
  }
  return 0;
}

/* G::msm_unchecked (src/msm.rs:84-87): see msm_planes in bp_oracle_arith.h */
int bpo_msm(int curve, const uint64_t* bx, const uint64_t* by, const uint8_t* binf, const uint64_t* s, size_t n,
            int threads, uint64_t out_xyz[12]) {
  CURVE_OR_FAIL(curve);
  jac total;
  int rc = msm_planes(&CURVES[curve], bx, by, binf, s, n, threads, &total);
  if (rc) return rc;
  jac_out(&CURVES[curve].fq, &total, out_xyz);
  return 0;
}

/* naive sum of scalar multiples: an independent cross-check of bpo_msm */
int bpo_msm_naive(int curve, const uint64_t* bx, const uint64_t* by, const uint8_t* binf, const uint64_t* s, size_t n,
                  uint64_t out_xyz[12]) {
  CURVE_OR_FAIL(curve); const field_t* fq = &CURVES[curve].fq; const field_t* fr = &CURVES[curve].fr;
  jac total = jac_identity(fq);
  for (size_t i = 0; i < n; i++) {
    fe px, py, k; memcpy(&px, bx + 4 * i, 32); memcpy(&py, by + 4 * i, 32); memcpy(&k, s + 4 * i, 32);
    k = fe_from_mont(fr, k);
    jac t = jac_scalar_mul(fq, k.l, &px, &py, binf && binf[i]);
    total = jac_add(fq, &total, &t);
  }
  jac_out(fq, &total, out_xyz);
  return 0;
}

/* P_i = (r0 + i*q)*G for the benchmark inputs: start = r0*G and step = q*G are
 * given (affine, Montgomery); fills count points by repeated addition and
 * per-block normalisation. */
int bpo_gen_points(int curve, const uint64_t start_xy[8], const uint64_t step_xy[8], size_t count, uint64_t* ox, uint64_t* oy) {
  CURVE_OR_FAIL(curve); const field_t* fq = &CURVES[curve].fq;
  fe sx, sy, qx, qy; memcpy(&sx, start_xy, 32); memcpy(&sy, start_xy + 4, 32); memcpy(&qx, step_xy, 32); memcpy(&qy, step_xy + 4, 32);
  jac cur; cur.X = sx; cur.Y = sy; cur.Z = fq->one;
  enum { BLK = 256 };
  jac tmp[BLK]; fe pre[BLK];
  for (size_t base = 0; base < count; base += BLK) {
    size_t m = count - base < BLK ? count - base : BLK;
    for (size_t i = 0; i < m; i++) { tmp[i] = cur; cur = jac_add_affine(fq, &cur, &qx, &qy); }
    fe acc = fq->one;                                 /* simultaneous inversion of the Z's */
    for (size_t i = 0; i < m; i++) { pre[i] = acc; acc = fe_mul(fq, acc, tmp[i].Z); }
    acc = fe_inv(fq, acc);
    for (size_t i = m; i-- > 0;) {
      fe zi = fe_mul(fq, acc, pre[i]); acc = fe_mul(fq, acc, tmp[i].Z);
      fe zi2 = fe_sqr(fq, zi);
      fe X = fe_mul(fq, tmp[i].X, zi2), Y = fe_mul(fq, fe_mul(fq, tmp[i].Y, zi2), zi);
      memcpy(ox + 4 * (base + i), X.l, 32); memcpy(oy + 4 * (base + i), Y.l, 32);
    }
  }
  return 0;
}

int bpo_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
