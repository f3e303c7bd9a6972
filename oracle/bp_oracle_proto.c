/* bp_oracle_proto.c -- CPU oracle (TEST INFRASTRUCTURE, never shipped or linked
 * into the product) for the callers of the Bulletproofs hot path: plain-C
 * restatements of the reference's provers and verifiers, message by message, so
 * that the GPU drivers can be checked at the full sizes of the benchmark
 * configurations.
 *
 *   ipa::prove / verify_aux / verify          src/ipa/mod.rs:19-212
 *   ipa::extended::prove / verify_aux         src/ipa/extended.rs:15-70
 *   range::prove / verify_aux (one value)     src/range/mod.rs:26-268
 *   range::aggregate::prove / verify_aux      src/range/aggregate.rs:23-300
 *   poly_commit::{prove, verify, lazy_verify, fold_todos_*}
 *                                             src/poly_commit/mod.rs:45-315
 *   Witness::statement                        src/ipa/types.rs:80-88,
 *                                             src/poly_commit/types.rs:137-151
 *
 * PARITY UNPINNED by reference byte vectors (see bp_oracle.c): the reference's
 * tests for these protocols are prove -> verify round trips on OsRng inputs.
 * This file is pinned by tests/test_oracle.py against the big-integer Python
 * oracle on small sizes and by the round-trip property itself (the verifier
 * restated here accepts what the prover restated here emits, and rejects a
 * corrupted proof).
 *
 * Stand-ins for what is outside the hot path: the Fiat-Shamir transcript
 * (spongefish) is replaced by seeded challenges -- challenge number j of a run is
 * the first field element of SplitMix64(seed + j) -- and OsRng by one SplitMix64
 * stream drawn in the reference's order.  A field element is four consecutive
 * outputs as little-endian limbs, rejected while >= the modulus.
 *
 * Data format = the arkworks memory form (see bp_oracle.c).  Group elements are
 * returned as 12 u64 (X, Y, Z normalised to Z = 1; identity = (1, 1, 0)). */
#include "bp_oracle_arith.h"

/* ---- seeded randomness and challenges ------------------------------------------ */
typedef struct { uint64_t s; } sm64;
static uint64_t sm64_next(sm64* g) {
  g->s += 0x9E3779B97F4A7C15ull;
  uint64_t z = g->s;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static fe sm64_field(sm64* g, const field_t* f) {      /* Montgomery form */
  for (;;) {
    fe v;
    for (int k = 0; k < 4; k++) v.l[k] = sm64_next(g);
    if (!geq(v.l, f->p)) return fe_mul(f, v, f->r2);
  }
}
typedef struct { uint64_t seed, counter; } chal_t;
static fe chal_next(chal_t* t, const field_t* fr) { sm64 g = {t->seed + t->counter++}; return sm64_field(&g, fr); }

/* ---- small helpers ---------------------------------------------------------------- */
typedef struct { fe x, y; int inf; } aff;
static aff aff_from_xy8(const uint64_t xy[8]) { aff a; memcpy(&a.x, xy, 32); memcpy(&a.y, xy + 4, 32); a.inf = 0; return a; }
static jac jac_from_aff(const field_t* fq, const aff* a) {
  if (a->inf) return jac_identity(fq);
  jac r; r.X = a->x; r.Y = a->y; r.Z = fq->one; return r;
}
static aff aff_from_jac(const field_t* fq, const jac* p) { aff a; a.inf = jac_to_affine(fq, p, &a.x, &a.y); return a; }
static jac jac_from_xyz(const field_t* fq, const uint64_t xyz[12]) {   /* Z is 0 or 1 by contract */
  jac r; memcpy(&r.X, xyz, 32); memcpy(&r.Y, xyz + 4, 32); memcpy(&r.Z, xyz + 8, 32);
  if (fe_is_zero(&r.Z)) return jac_identity(fq);
  return r;
}
static jac jac_neg(const field_t* fq, const jac* p) { jac r = *p; r.Y = fe_neg(fq, p->Y); return r; }
/* k * P, k in Montgomery form */
static jac smul(const curve_t* cv, fe k_mont, const aff* p) {
  fe k = fe_from_mont(&cv->fr, k_mont);
  return jac_scalar_mul(&cv->fq, k.l, &p->x, &p->y, p->inf);
}
static jac smul_jac(const curve_t* cv, fe k_mont, const jac* p) { aff a = aff_from_jac(&cv->fq, p); return smul(cv, k_mont, &a); }
static fe fe_from_u64(const field_t* f, uint64_t v) { fe t = {{v, 0, 0, 0}}; return fe_mul(f, t, f->r2); }
static fe fe_pow_u64(const field_t* f, fe b, uint64_t e) {
  fe r = f->one;
  for (; e; e >>= 1) { if (e & 1) r = fe_mul(f, r, b); b = fe_sqr(f, b); }
  return r;
}
/* src/vector_ops.rs:159-171 */
static fe dot(const field_t* f, const fe* a, const fe* b, size_t n) {
  fe acc = {{0, 0, 0, 0}};
  for (size_t i = 0; i < n; i++) acc = fe_add(f, acc, fe_mul(f, a[i], b[i]));
  return acc;
}
/* ark_ff::batch_inversion: one inversion and three products per element; zeros stay zero */
static int batch_inverse(const field_t* f, fe* v, size_t n) {
  fe* pre = (fe*)malloc((n ? n : 1) * sizeof(fe));
  if (!pre) return -4;
  fe run = f->one;
  for (size_t i = 0; i < n; i++) { pre[i] = run; if (!fe_is_zero(&v[i])) run = fe_mul(f, run, v[i]); }
  fe acc = fe_inv(f, run);
  for (size_t i = n; i-- > 0;) {
    if (fe_is_zero(&v[i])) continue;
    fe t = fe_mul(f, acc, pre[i]); acc = fe_mul(f, acc, v[i]); v[i] = t;
  }
  free(pre);
  return 0;
}
/* s_i = prod_j (bit j of i ? ch[k-1-j] : ch_inv[k-1-j]): src/ipa/mod.rs:134-148, HPoly::coeffs
 * src/poly_commit/mod.rs:194-219 */
static void s_vector(const field_t* f, const fe* ch, const fe* chi, size_t k, fe* out, int threads) {
  size_t n = (size_t)1 << k;
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
  for (long i = 0; i < (long)n; i++) {
    fe acc = f->one;
    for (size_t j = 0; j < k; j++) { size_t idx = k - j - 1; acc = fe_mul(f, acc, (((size_t)i >> j) & 1) ? ch[idx] : chi[idx]); }
    out[i] = acc;
  }
}
/* out_i = l_i*x + r_i*y: fold_scalars src/ipa/mod.rs:229-234 (in place when out == l) */
static void fold_scalars(const field_t* f, const fe* l, const fe* r, size_t h, fe x, fe y, fe* out, int threads) {
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
  for (long i = 0; i < (long)h; i++) out[i] = fe_add(f, fe_mul(f, l[i], x), fe_mul(f, r[i], y));
}
/* out_i = x*L_i + y*R_i normalised to affine: fold_generators src/ipa/mod.rs:215-227,
 * src/poly_commit/mod.rs:31-43.  The two scalar multiplications share their doublings
 * (the group element is the same); blocks of outputs share one inversion, which is what
 * G::normalize_batch does.  out may alias L. */
static void fold_generators(const curve_t* cv, const fe* lx, const fe* ly, const uint8_t* linf, const fe* rx, const fe* ry,
                            const uint8_t* rinf, size_t h, fe x_mont, fe y_mont, fe* ox, fe* oy, uint8_t* oinf, int threads) {
  const field_t* fq = &cv->fq;
  fe kx = fe_from_mont(&cv->fr, x_mont), ky = fe_from_mont(&cv->fr, y_mont);
  enum { BLK = 128 };
  long n_blk = (long)((h + BLK - 1) / BLK);
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 1)
  for (long blk = 0; blk < n_blk; blk++) {
    jac res[BLK]; fe pre[BLK];
    size_t base = (size_t)blk * BLK, m = h - base < BLK ? h - base : BLK;
    for (size_t t = 0; t < m; t++) {
      size_t i = base + t;
      int li = linf && linf[i], ri = rinf && rinf[i];
      jac lj, rj, both, acc = jac_identity(fq);
      aff la = {lx[i], ly[i], li}, ra = {rx[i], ry[i], ri};
      lj = jac_from_aff(fq, &la); rj = jac_from_aff(fq, &ra);
      both = jac_add(fq, &lj, &rj);
      for (int bit = 255; bit >= 0; bit--) {
        acc = jac_double(fq, &acc);
        int bx = (int)((kx.l[bit >> 6] >> (bit & 63)) & 1) && !li, by = (int)((ky.l[bit >> 6] >> (bit & 63)) & 1) && !ri;
        if (bx && by) acc = jac_add(fq, &acc, &both);
        else if (bx) acc = jac_add_affine(fq, &acc, &la.x, &la.y);
        else if (by) acc = jac_add_affine(fq, &acc, &ra.x, &ra.y);
      }
      res[t] = acc;
    }
    fe run = fq->one;
    for (size_t t = 0; t < m; t++) { pre[t] = run; if (!fe_is_zero(&res[t].Z)) run = fe_mul(fq, run, res[t].Z); }
    fe inv = fe_inv(fq, run);
    for (size_t t = m; t-- > 0;) {
      size_t i = base + t;
      if (fe_is_zero(&res[t].Z)) { memset(&ox[i], 0, 32); memset(&oy[i], 0, 32); if (oinf) oinf[i] = 1; continue; }
      fe zi = fe_mul(fq, inv, pre[t]); inv = fe_mul(fq, inv, res[t].Z);
      fe zi2 = fe_sqr(fq, zi);
      ox[i] = fe_mul(fq, res[t].X, zi2); oy[i] = fe_mul(fq, fe_mul(fq, res[t].Y, zi2), zi);
      if (oinf) oinf[i] = 0;
    }
  }
}
static int msm_fe(const curve_t* cv, const fe* bx, const fe* by, const uint8_t* binf, const fe* s, size_t n, int threads, jac* out) {
  return msm_planes(cv, (const uint64_t*)bx, (const uint64_t*)by, binf, (const uint64_t*)s, n, threads, out);
}

/* A growable list of (base, scalar) pairs: the content of the reference's Msm accumulator
 * (src/msm.rs:8-88).  Merging duplicate bases does not change the group element of
 * execute(), so the pairs are kept as they come. */
typedef struct { fe *x, *y, *s; uint8_t* inf; size_t n, cap; } acc_t;
static int acc_init(acc_t* a, size_t cap) {
  a->n = 0; a->cap = cap ? cap : 1;
  a->x = (fe*)malloc(a->cap * sizeof(fe)); a->y = (fe*)malloc(a->cap * sizeof(fe)); a->s = (fe*)malloc(a->cap * sizeof(fe));
  a->inf = (uint8_t*)malloc(a->cap);
  return (a->x && a->y && a->s && a->inf) ? 0 : -4;
}
static void acc_free(acc_t* a) { free(a->x); free(a->y); free(a->s); free(a->inf); }
static void acc_push(acc_t* a, const aff* p, fe s) {   /* capacity is fixed by the caller */
  if (a->n >= a->cap) abort();
  a->x[a->n] = p->x; a->y[a->n] = p->y; a->inf[a->n] = (uint8_t)p->inf; a->s[a->n] = s; a->n++;
}
static int acc_execute(const curve_t* cv, const acc_t* a, int threads, jac* out) {   /* src/msm.rs:84-87 */
  return msm_fe(cv, a->x, a->y, a->inf, a->s, a->n, threads, out);
}

/* ================================================================================== */
/* inner-product argument: src/ipa/mod.rs                                              */
/* ================================================================================== */

/* The `while n != 1` loop of ipa::prove (src/ipa/mod.rs:33-95).  g, h, a, b (length n) are
 * consumed (folded in place).  out_l / out_r: log2(n) normalised points, `stride` u64 apart. */
static int ipa_prove_core(const curve_t* cv, fe* gx, fe* gy, uint8_t* gi, fe* hx, fe* hy, uint8_t* hi, fe* a, fe* b,
                          size_t n, const aff* u, chal_t* t, int threads, uint64_t* out_l, uint64_t* out_r, size_t stride,
                          fe* a0, fe* b0) {
  const field_t* fq = &cv->fq; const field_t* fr = &cv->fr;
  fe* bx = (fe*)malloc((n ? n : 1) * sizeof(fe)); fe* by = (fe*)malloc((n ? n : 1) * sizeof(fe));
  fe* sc = (fe*)malloc((n ? n : 1) * sizeof(fe)); uint8_t* bi = (uint8_t*)malloc(n ? n : 1);
  if (!bx || !by || !sc || !bi) { free(bx); free(by); free(sc); free(bi); return -4; }
  size_t round = 0;
  int rc = 0;
  while (n != 1 && !rc) {
    n /= 2;                                                             /* :34-38 */
    fe c_left = dot(fr, a, b + n, n);                                   /* :42 */
    memcpy(bx, gx + n, n * 32); memcpy(by, gy + n, n * 32); memcpy(bi, gi + n, n);        /* g_right || h_left :44-48 */
    memcpy(bx + n, hx, n * 32); memcpy(by + n, hy, n * 32); memcpy(bi + n, hi, n);
    memcpy(sc, a, n * 32); memcpy(sc + n, b + n, n * 32);                                 /* a_left || b_right :49-53 */
    jac m, left, right;
    if ((rc = msm_fe(cv, bx, by, bi, sc, 2 * n, threads, &m))) break;                     /* :54 */
    left = smul(cv, c_left, u); left = jac_add(fq, &left, &m);                            /* :43 */
    fe c_right = dot(fr, a + n, b, n);                                  /* :58 */
    memcpy(bx, gx, n * 32); memcpy(by, gy, n * 32); memcpy(bi, gi, n);                    /* g_left || h_right :60-64 */
    memcpy(bx + n, hx + n, n * 32); memcpy(by + n, hy + n, n * 32); memcpy(bi + n, hi + n, n);
    memcpy(sc, a + n, n * 32); memcpy(sc + n, b, n * 32);                                 /* a_right || b_left :65-69 */
    if ((rc = msm_fe(cv, bx, by, bi, sc, 2 * n, threads, &m))) break;                     /* :70 */
    right = smul(cv, c_right, u); right = jac_add(fq, &right, &m);                        /* :59 */
    jac_out(fq, &left, out_l + stride * round); jac_out(fq, &right, out_r + stride * round);  /* :75 */
    fe alpha = chal_next(t, fr), alpha_inv = fe_inv(fr, alpha);         /* :77-78 */
    fold_generators(cv, gx, gy, gi, gx + n, gy + n, gi + n, n, alpha_inv, alpha, gx, gy, gi, threads);  /* :81 */
    fold_generators(cv, hx, hy, hi, hx + n, hy + n, hi + n, n, alpha, alpha_inv, hx, hy, hi, threads);  /* :82 */
    fold_scalars(fr, a, a + n, n, alpha, alpha_inv, a, threads);        /* :88 */
    fold_scalars(fr, b, b + n, n, alpha_inv, alpha, b, threads);        /* :89 */
    round++;
  }
  *a0 = a[0]; *b0 = b[0];                                               /* :97 */
  free(bx); free(by); free(sc); free(bi);
  return rc;
}

/* ipa::verify_aux (src/ipa/mod.rs:102-193) appended to `acc`: u <-> a*b*u_scale, gs_i <-> a*s_i,
 * hs_i <-> b*s_i^-1*h_scale_i, and -(P + sum x_j^2 L_j + x_j^-2 R_j).  u_scale is the factor
 * of Msm::scale_elem(crs.u, x) (src/ipa/extended.rs:68; one for the plain argument) and h_scale
 * the factors of Msm::scale_elems(scaled_hs) (src/range/aggregate.rs:272; NULL for none). */
static int ipa_verify_core(const curve_t* cv, const fe* gx, const fe* gy, const fe* hx, const fe* hy, const aff* u,
                           const jac* p, size_t n, const uint64_t* L, const uint64_t* R, size_t stride, fe a, fe b,
                           fe u_scale, const fe* h_scale, chal_t* t, int threads, acc_t* acc) {
  const field_t* fq = &cv->fq; const field_t* fr = &cv->fr;
  size_t k = 0;
  while (((size_t)1 << k) < n) k++;                                     /* :110-111 */
  fe* xs = (fe*)malloc((k ? k : 1) * sizeof(fe)); fe* xinv = (fe*)malloc((k ? k : 1) * sizeof(fe));
  fe* ss = (fe*)malloc(n * sizeof(fe)); fe* ssi = (fe*)malloc(n * sizeof(fe));
  if (!xs || !xinv || !ss || !ssi) { free(xs); free(xinv); free(ss); free(ssi); return -4; }
  for (size_t j = 0; j < k; j++) { xs[j] = chal_next(t, fr); xinv[j] = xs[j]; }   /* :113-119 */
  int rc = batch_inverse(fr, xinv, k);                                  /* :128-132 */
  acc_push(acc, u, fe_mul(fr, fe_mul(fr, a, b), u_scale));              /* :125 */
  s_vector(fr, xs, xinv, k, ss, threads);                               /* :134-148 */
  memcpy(ssi, ss, n * sizeof(fe));
  if (!rc) rc = batch_inverse(fr, ssi, n);                              /* :150-154 */
  for (size_t i = 0; i < n && !rc; i++) { aff g = {gx[i], gy[i], 0}; acc_push(acc, &g, fe_mul(fr, ss[i], a)); }   /* :156-161 */
  for (size_t i = 0; i < n && !rc; i++) {                               /* :162-167 */
    aff h = {hx[i], hy[i], 0};
    fe s = fe_mul(fr, ssi[i], b);
    if (h_scale) s = fe_mul(fr, s, h_scale[i]);
    acc_push(acc, &h, s);
  }
  fe minus_one = fe_neg(fr, fr->one);                                   /* :170-190 */
  aff pa = aff_from_jac(fq, p);
  acc_push(acc, &pa, minus_one);
  for (size_t j = 0; j < k && !rc; j++) {
    jac l = jac_from_xyz(fq, L + stride * j), r = jac_from_xyz(fq, R + stride * j);
    aff la = aff_from_jac(fq, &l), ra = aff_from_jac(fq, &r);
    acc_push(acc, &la, fe_mul(fr, minus_one, fe_sqr(fr, xs[j])));
    acc_push(acc, &ra, fe_mul(fr, minus_one, fe_sqr(fr, xinv[j])));
  }
  free(xs); free(xinv); free(ss); free(ssi);
  return rc;
}

static size_t log2_exact(size_t n) { size_t k = 0; while (((size_t)1 << k) < n) k++; return k; }

/* Witness::statement, src/ipa/types.rs:80-88: P = <a, G> + <b, H> + <a, b> u */
int bpo_ipa_statement(int curve, const uint64_t* gx, const uint64_t* gy, const uint64_t* hx, const uint64_t* hy,
                      const uint64_t u_xy[8], const uint64_t* a, const uint64_t* b, size_t n, int threads, uint64_t out_xyz[12]) {
  CURVE_OR_FAIL(curve); const curve_t* cv = &CURVES[curve];
  jac g, h; int rc;
  if ((rc = msm_planes(cv, gx, gy, NULL, a, n, threads, &g))) return rc;
  if ((rc = msm_planes(cv, hx, hy, NULL, b, n, threads, &h))) return rc;
  aff u = aff_from_xy8(u_xy);
  jac c = smul(cv, dot(&cv->fr, (const fe*)a, (const fe*)b, n), &u);
  jac p = jac_add(&cv->fq, &g, &h); p = jac_add(&cv->fq, &p, &c);
  jac_out(&cv->fq, &p, out_xyz);
  return 0;
}

/* ipa::prove, src/ipa/mod.rs:19-99.  out_l, out_r: log2(n) x 12 u64; out_ab: a[0], b[0]. */
int bpo_ipa_prove(int curve, const uint64_t* gx, const uint64_t* gy, const uint64_t* hx, const uint64_t* hy,
                  const uint64_t u_xy[8], const uint64_t* a, const uint64_t* b, size_t n, uint64_t chal_seed, int threads,
                  uint64_t* out_l, uint64_t* out_r, uint64_t out_ab[8]) {
  CURVE_OR_FAIL(curve); const curve_t* cv = &CURVES[curve];
  if (n == 0 || (n & (n - 1))) return -1;
  fe *g_x = (fe*)malloc(n * 32), *g_y = (fe*)malloc(n * 32), *h_x = (fe*)malloc(n * 32), *h_y = (fe*)malloc(n * 32);
  fe *wa = (fe*)malloc(n * 32), *wb = (fe*)malloc(n * 32);
  uint8_t *gi = (uint8_t*)calloc(n, 1), *hi = (uint8_t*)calloc(n, 1);
  int rc = -4;
  if (g_x && g_y && h_x && h_y && wa && wb && gi && hi) {
    memcpy(g_x, gx, n * 32); memcpy(g_y, gy, n * 32); memcpy(h_x, hx, n * 32); memcpy(h_y, hy, n * 32);   /* :29-31 */
    memcpy(wa, a, n * 32); memcpy(wb, b, n * 32);
    aff u = aff_from_xy8(u_xy);
    chal_t t = {chal_seed, 0};
    fe a0, b0;
    rc = ipa_prove_core(cv, g_x, g_y, gi, h_x, h_y, hi, wa, wb, n, &u, &t, threads, out_l, out_r, 12, &a0, &b0);
    memcpy(out_ab, a0.l, 32); memcpy(out_ab + 4, b0.l, 32);
  }
  free(g_x); free(g_y); free(h_x); free(h_y); free(wa); free(wb); free(gi); free(hi);
  return rc;
}

/* ipa::verify, src/ipa/mod.rs:196-212: the value of the deferred MSM (the identity iff the proof
 * is accepted). */
int bpo_ipa_verify(int curve, const uint64_t* gx, const uint64_t* gy, const uint64_t* hx, const uint64_t* hy,
                   const uint64_t u_xy[8], const uint64_t p_xyz[12], size_t n, const uint64_t* L, const uint64_t* R,
                   const uint64_t ab[8], uint64_t chal_seed, int threads, uint64_t out_xyz[12]) {
  CURVE_OR_FAIL(curve); const curve_t* cv = &CURVES[curve];
  if (n == 0 || (n & (n - 1))) return -1;
  size_t k = log2_exact(n);
  acc_t acc;
  if (acc_init(&acc, 2 * n + 2 * k + 2)) { acc_free(&acc); return -4; }
  aff u = aff_from_xy8(u_xy);
  jac p = jac_from_xyz(&cv->fq, p_xyz);
  fe a, b; memcpy(&a, ab, 32); memcpy(&b, ab + 4, 32);
  chal_t t = {chal_seed, 0};
  int rc = ipa_verify_core(cv, (const fe*)gx, (const fe*)gy, (const fe*)hx, (const fe*)hy, &u, &p, n, L, R, 12, a, b,
                           cv->fr.one, NULL, &t, threads, &acc);
  jac v;
  if (!rc) rc = acc_execute(cv, &acc, threads, &v);                     /* :205-206 */
  if (!rc) jac_out(&cv->fq, &v, out_xyz);
  acc_free(&acc);
  return rc;
}

/* ================================================================================== */
/* range proofs: src/range/mod.rs (one value), src/range/aggregate.rs (m values)       */
/* ================================================================================== */

/* V_j = v_j g + gamma_j h: Statement::new, src/range/types.rs:62-68,167-176 */
static jac value_commitment(const curve_t* cv, const aff* g, const aff* h, uint64_t v, fe gamma) {
  jac a = smul(cv, fe_from_u64(&cv->fr, v), g), b = smul(cv, gamma, h);
  return jac_add(&cv->fq, &a, &b);
}

/* range::aggregate::prove, src/range/aggregate.rs:23-170 (with m = 1 the expressions are those of
 * range::prove, src/range/mod.rs:26-159, term by term: z^(2+j) is z^2, sigma_summand is
 * z^2 gamma).  ugh_xy = affine u, g, h.  out_points = A, S, T1, T2; out_scalars = tao_x, mu,
 * t_hat, a, b; out_lr = log2(N) x (L | R), 24 u64 per round. */
int bpo_range_agg_prove(int curve, const uint64_t* gx, const uint64_t* gy, const uint64_t* hx, const uint64_t* hy,
                        const uint64_t* ugh_xy, const uint64_t* values, const uint64_t* gamma, size_t m, unsigned n_bits,
                        uint64_t seed_rand, uint64_t seed_chal, int threads, uint64_t* out_points, uint64_t* out_scalars,
                        uint64_t* out_lr) {
  CURVE_OR_FAIL(curve); const curve_t* cv = &CURVES[curve];
  const field_t* fq = &cv->fq; const field_t* fr = &cv->fr;
  const size_t n = (size_t)n_bits * m;
  if (n == 0 || (n & (n - 1)) || n_bits > 64) return -1;
  const aff u = aff_from_xy8(ugh_xy), g = aff_from_xy8(ugh_xy + 8), h = aff_from_xy8(ugh_xy + 16);
  sm64 rng = {seed_rand};
  chal_t t = {seed_chal, 0};
  fe* buf = (fe*)malloc(14 * n * sizeof(fe));
  uint8_t* flags = (uint8_t*)calloc(2 * n, 1);
  if (!buf || !flags) { free(buf); free(flags); return -4; }
  fe *a_l = buf, *a_r = buf + n /* contiguous: a_l || a_r */, *s_l = buf + 2 * n, *s_r = buf + 3 * n /* s_l || s_r */;
  fe *bx = buf + 4 * n, *by = buf + 6 * n /* gs || hs */, *l0 = buf + 8 * n, *r0 = buf + 9 * n, *r1 = buf + 10 * n;
  fe *yv = buf + 11 * n, *hpx = buf + 12 * n, *hpy = buf + 13 * n;
  const fe one = fr->one, zero = {{0, 0, 0, 0}};
  for (size_t j = 0; j < m; j++)                                        /* :50-64 */
    for (unsigned i = 0; i < n_bits; i++) {
      int bit = (int)((values[j] >> i) & 1);
      a_l[j * n_bits + i] = bit ? one : zero;
      a_r[j * n_bits + i] = fe_sub(fr, a_l[j * n_bits + i], one);
    }
  fe alpha = sm64_field(&rng, fr), rho = sm64_field(&rng, fr);          /* :66-67 */
  for (size_t i = 0; i < n; i++) s_l[i] = sm64_field(&rng, fr);         /* :68-70 */
  for (size_t i = 0; i < n; i++) s_r[i] = sm64_field(&rng, fr);         /* :71-73 */
  memcpy(bx, gx, n * 32); memcpy(bx + n, hx, n * 32); memcpy(by, gy, n * 32); memcpy(by + n, hy, n * 32);   /* :76 */
  jac A, S, tmp; int rc;
  if ((rc = msm_fe(cv, bx, by, NULL, a_l, 2 * n, threads, &A))) goto done;      /* :81 */
  tmp = smul(cv, alpha, &h); A = jac_add(fq, &tmp, &A);
  if ((rc = msm_fe(cv, bx, by, NULL, s_l, 2 * n, threads, &S))) goto done;      /* :82 */
  tmp = smul(cv, rho, &h); S = jac_add(fq, &tmp, &S);
  jac_out(fq, &A, out_points); jac_out(fq, &S, out_points + 12);        /* :85 */
  fe y = chal_next(&t, fr), z = chal_next(&t, fr);                      /* :86 */
  yv[0] = one;
  for (size_t i = 1; i < n; i++) yv[i] = fe_mul(fr, yv[i - 1], y);      /* :88-91 */
  fe zz = fe_sqr(fr, z), zp = zz, two = fe_from_u64(fr, 2);
  for (size_t j = 0; j < m; j++) {                                      /* :93-119 */
    fe p2 = one;
    for (unsigned i = 0; i < n_bits; i++) {
      size_t k = j * n_bits + i;
      l0[k] = fe_sub(fr, a_l[k], z);
      r0[k] = fe_add(fr, fe_mul(fr, yv[k], fe_add(fr, a_r[k], z)), fe_mul(fr, zp, p2));
      r1[k] = fe_mul(fr, yv[k], s_r[k]);
      p2 = fe_mul(fr, p2, two);
    }
    zp = fe_mul(fr, zp, z);
  }
  fe tao1 = sm64_field(&rng, fr), tao2 = sm64_field(&rng, fr);          /* :121-122 */
  /* VectorPolynomial::inner_product, src/range/types.rs:98-117: t1 = <l0,r1> + <l1,r0>, t2 = <l1,r1> */
  fe t1 = fe_add(fr, dot(fr, l0, r1, n), dot(fr, s_l, r0, n)), t2 = dot(fr, s_l, r1, n);
  jac T1 = smul(cv, t1, &g), T2 = smul(cv, t2, &g);
  tmp = smul(cv, tao1, &h); T1 = jac_add(fq, &T1, &tmp);                /* :126 */
  tmp = smul(cv, tao2, &h); T2 = jac_add(fq, &T2, &tmp);                /* :127 */
  jac_out(fq, &T1, out_points + 24); jac_out(fq, &T2, out_points + 36); /* :129 */
  fe x = chal_next(&t, fr);                                             /* :133 */
  fe sigma = zero; zp = zz;
  for (size_t j = 0; j < m; j++) { fe gj; memcpy(&gj, gamma + 4 * j, 32); sigma = fe_add(fr, sigma, fe_mul(fr, zp, gj)); zp = fe_mul(fr, zp, z); }   /* :136-139 */
  fe tao_x = fe_add(fr, fe_add(fr, fe_mul(fr, tao1, x), fe_mul(fr, tao2, fe_sqr(fr, x))), sigma);   /* :140 */
  fe mu = fe_add(fr, alpha, fe_mul(fr, rho, x));                        /* :142 */
  /* VectorPolynomial::evaluate, src/range/types.rs:119-129; l and r overwrite l0 and r0 */
  for (size_t i = 0; i < n; i++) { l0[i] = fe_add(fr, l0[i], fe_mul(fr, s_l[i], x)); r0[i] = fe_add(fr, r0[i], fe_mul(fr, r1[i], x)); }   /* :143-144 */
  fe t_hat = dot(fr, l0, r0, n);                                        /* Witness::new, src/ipa/types.rs:56-62 */
  /* hs'_i = y^-i h_i: create_hs_prime src/range/utils.rs:22-29 + normalize_batch, :148-153 */
  {
    fe y_inv = fe_inv(fr, y), cur = one;
    for (size_t i = 0; i < n; i++) { yv[i] = cur; cur = fe_mul(fr, cur, y_inv); }
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 16)
    for (long i = 0; i < (long)n; i++) {
      aff hi = {((const fe*)hx)[i], ((const fe*)hy)[i], 0};
      jac p = smul(cv, yv[i], &hi);
      aff o = aff_from_jac(fq, &p);
      hpx[i] = o.x; hpy[i] = o.y; flags[n + i] = (uint8_t)o.inf;
    }
  }
  memcpy(out_scalars, tao_x.l, 32); memcpy(out_scalars + 4, mu.l, 32); memcpy(out_scalars + 8, t_hat.l, 32);   /* :159 */
  /* ipa::extended::prove, src/ipa/extended.rs:15-35: u' = x' u; the statement's p is never sent */
  {
    fe xe = chal_next(&t, fr);                                          /* extended.rs:24 */
    jac uj = smul(cv, xe, &u);
    aff u2 = aff_from_jac(fq, &uj);                                     /* extended.rs:32 */
    memcpy(bx, gx, n * 32); memcpy(by, gy, n * 32);                     /* gs.to_vec() :161 */
    fe a0, b0;
    rc = ipa_prove_core(cv, bx, by, flags, hpx, hpy, flags + n, l0, r0, n, &u2, &t, threads, out_lr, out_lr + 12, 24, &a0, &b0);   /* :166 */
    memcpy(out_scalars + 12, a0.l, 32); memcpy(out_scalars + 16, b0.l, 32);
  }
done:
  free(buf); free(flags);
  return rc;
}

/* range::aggregate::verify_aux + verify, src/range/aggregate.rs:173-300, in the reference's order of
 * operations (with m = 1: range::verify_aux, src/range/mod.rs:162-268, where statement.v.mul(z^2)
 * is the one-term case of the MSM at aggregate.rs:226).  The value commitments are built from
 * values and gamma (Statement::new).  `points`, `scalars`, `lr` are a proof in the layout
 * bpo_range_agg_prove writes; seed_rand seeds the verifier's rng (one draw, :278).  Returns the
 * value of the deferred MSM: the identity iff the proof is accepted. */
int bpo_range_agg_verify(int curve, const uint64_t* gx, const uint64_t* gy, const uint64_t* hx, const uint64_t* hy,
                         const uint64_t* ugh_xy, const uint64_t* values, const uint64_t* gamma, size_t m, unsigned n_bits,
                         const uint64_t* points, const uint64_t* scalars, const uint64_t* lr, uint64_t seed_rand,
                         uint64_t seed_chal, int threads, uint64_t out_xyz[12]) {
  CURVE_OR_FAIL(curve); const curve_t* cv = &CURVES[curve];
  const field_t* fq = &cv->fq; const field_t* fr = &cv->fr;
  const size_t n = (size_t)n_bits * m;
  if (n == 0 || (n & (n - 1)) || n_bits > 64) return -1;
  const size_t k = log2_exact(n);
  const aff u = aff_from_xy8(ugh_xy), g = aff_from_xy8(ugh_xy + 8), h = aff_from_xy8(ugh_xy + 16);
  sm64 rng = {seed_rand};
  chal_t t = {seed_chal, 0};
  const fe one = fr->one, zero = {{0, 0, 0, 0}};
  jac A = jac_from_xyz(fq, points), S = jac_from_xyz(fq, points + 12), T1 = jac_from_xyz(fq, points + 24), T2 = jac_from_xyz(fq, points + 36);
  fe tao_x, mu, t_hat, pa, pb;
  memcpy(&tao_x, scalars, 32); memcpy(&mu, scalars + 4, 32); memcpy(&t_hat, scalars + 8, 32);
  memcpy(&pa, scalars + 12, 32); memcpy(&pb, scalars + 16, 32);
  fe y = chal_next(&t, fr), z = chal_next(&t, fr);                      /* :186-189 */
  fe x = chal_next(&t, fr);                                             /* :191 */
  fe* buf = (fe*)malloc((6 * n + 2 * m + 64) * sizeof(fe));
  acc_t acc;
  int rc = acc_init(&acc, 2 * n + 2 * k + 3);
  if (!buf || rc) { free(buf); acc_free(&acc); return -4; }
  fe *yv = buf, *yiv = buf + n, *sc = buf + 2 * n /* 2n */, *hpx = buf + 4 * n, *hpy = buf + 5 * n, *zv = buf + 6 * n, *two_vec = buf + 6 * n + m;
  fe two = fe_from_u64(fr, 2), zz = fe_sqr(fr, z);
  two_vec[0] = one;
  for (unsigned i = 1; i < n_bits; i++) two_vec[i] = fe_mul(fr, two_vec[i - 1], two);   /* :194-198 */
  yv[0] = one;
  for (size_t i = 1; i < n; i++) yv[i] = fe_mul(fr, yv[i - 1], y);      /* :199-202 */
  /* first closure, :205-227 */
  jac gchk;
  {
    fe d = zero, zsum = zero, zp = fe_mul(fr, zz, z), sum_y = zero;
    for (unsigned i = 0; i < n_bits; i++) d = fe_add(fr, d, two_vec[i]);
    for (size_t j = 0; j < m; j++) { zsum = fe_add(fr, zsum, zp); zp = fe_mul(fr, zp, z); }
    fe sigma = fe_mul(fr, zsum, d);                                     /* :209-213 */
    for (size_t i = 0; i < n; i++) sum_y = fe_add(fr, sum_y, yv[i]);
    fe delta = fe_sub(fr, fe_mul(fr, fe_sub(fr, z, zz), sum_y), sigma); /* :215 */
    zp = zz;
    fe *vx = (fe*)malloc(m * 32), *vy = (fe*)malloc(m * 32); uint8_t* vi = (uint8_t*)malloc(m);
    if (!vx || !vy || !vi) { free(vx); free(vy); free(vi); rc = -4; goto done; }
    for (size_t j = 0; j < m; j++) {                                    /* :217-223 */
      zv[j] = zp; zp = fe_mul(fr, zp, z);
      fe gj; memcpy(&gj, gamma + 4 * j, 32);
      jac V = value_commitment(cv, &g, &h, values[j], gj);
      aff va = aff_from_jac(fq, &V);
      vx[j] = va.x; vy[j] = va.y; vi[j] = (uint8_t)va.inf;
    }
    jac vz;
    rc = msm_fe(cv, vx, vy, vi, zv, m, threads, &vz);                   /* :226 */
    free(vx); free(vy); free(vi);
    if (rc) goto done;
    jac lhs = smul(cv, fe_sub(fr, t_hat, delta), &g), th = smul(cv, tao_x, &h);
    lhs = jac_add(fq, &lhs, &th);                                       /* :225 */
    jac r1 = smul_jac(cv, x, &T1), r2 = smul_jac(cv, fe_sqr(fr, x), &T2);
    jac rhs = jac_add(fq, &vz, &r1); rhs = jac_add(fq, &rhs, &r2);
    rhs = jac_neg(fq, &rhs);
    gchk = jac_add(fq, &lhs, &rhs);                                     /* :225-226 */
  }
  /* second closure, :228-274 */
  {
    fe y_inv = fe_inv(fr, y), cur = one;
    for (size_t i = 0; i < n; i++) { yiv[i] = cur; cur = fe_mul(fr, cur, y_inv); }     /* create_hs_prime :231 */
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 16)
    for (long i = 0; i < (long)n; i++) {                                /* :232-237 */
      aff hi = {((const fe*)hx)[i], ((const fe*)hy)[i], 0};
      jac p = smul(cv, yiv[i], &hi);
      aff o = aff_from_jac(fq, &p);
      hpx[i] = o.x; hpy[i] = o.y;
    }
    fe neg_z = fe_neg(fr, z), zp = zz;
    for (size_t j = 0; j < m; j++) {                                    /* :240-252 */
      for (unsigned i = 0; i < n_bits; i++) {
        size_t idx = j * n_bits + i;
        sc[idx] = neg_z;
        sc[n + idx] = fe_add(fr, fe_mul(fr, yv[idx], z), fe_mul(fr, two_vec[i], zp));
      }
      zp = fe_mul(fr, zp, z);
    }
    fe *bx = (fe*)malloc(2 * n * 32), *by = (fe*)malloc(2 * n * 32);
    if (!bx || !by) { free(bx); free(by); rc = -4; goto done; }
    memcpy(bx, gx, n * 32); memcpy(by, gy, n * 32); memcpy(bx + n, hpx, n * 32); memcpy(by + n, hpy, n * 32);   /* :255-256 */
    jac base;
    rc = msm_fe(cv, bx, by, NULL, sc, 2 * n, threads, &base);           /* :259 */
    free(bx); free(by);
    if (rc) goto done;
    jac sx = smul_jac(cv, x, &S);
    jac p = jac_add(fq, &A, &sx); p = jac_add(fq, &p, &base);
    jac mh = smul(cv, fe_neg(fr, mu), &h);
    p = jac_add(fq, &p, &mh);                                           /* :264 */
    /* ipa::extended::verify_aux, src/ipa/extended.rs:54-70, over the unscaled crs.ipa_crs */
    fe xe = chal_next(&t, fr);                                          /* extended.rs:62 */
    jac uc = smul(cv, fe_mul(fr, xe, t_hat), &u);
    p = jac_add(fq, &p, &uc);                                           /* extended.rs:64 */
    rc = ipa_verify_core(cv, (const fe*)gx, (const fe*)gy, (const fe*)hx, (const fe*)hy, &u, &p, n, lr, lr + 12, 24, pa, pb,
                         xe /* scale_elem(crs.u, x) extended.rs:68 */, yiv /* scale_elems(scaled_hs) :272 */, &t, threads, &acc);
    if (rc) goto done;
  }
  {
    fe alpha = sm64_field(&rng, fr);                                    /* :278 */
    aff ga = aff_from_jac(fq, &gchk);
    acc_push(&acc, &ga, alpha);                                         /* :279 */
    jac v;
    rc = acc_execute(cv, &acc, threads, &v);                            /* :295 */
    if (!rc) jac_out(fq, &v, out_xyz);
  }
done:
  free(buf); acc_free(&acc);
  return rc;
}

/* The m value commitments V_j (Statement::new, src/range/types.rs:167-176) as m x 12 u64 */
int bpo_range_value_commitments(int curve, const uint64_t* ugh_xy, const uint64_t* values, const uint64_t* gamma, size_t m,
                                uint64_t* out_xyz) {
  CURVE_OR_FAIL(curve); const curve_t* cv = &CURVES[curve];
  const aff g = aff_from_xy8(ugh_xy + 8), h = aff_from_xy8(ugh_xy + 16);
  for (size_t j = 0; j < m; j++) {
    fe gj; memcpy(&gj, gamma + 4 * j, 32);
    jac V = value_commitment(cv, &g, &h, values[j], gj);
    jac_out(&cv->fq, &V, out_xyz + 12 * j);
  }
  return 0;
}

/* ================================================================================== */
/* polynomial commitment: src/poly_commit                                              */
/* ================================================================================== */

/* HPoly::evaluate, src/poly_commit/mod.rs:181-192 */
static fe hpoly_evaluate(const field_t* fr, const fe* ui, size_t k, fe x) {
  fe acc = fr->one, xp = x;
  for (size_t i = 0; i < k; i++) {
    fe u = ui[k - 1 - i];
    acc = fe_mul(fr, acc, fe_add(fr, fe_inv(fr, u), fe_mul(fr, u, xp)));
    xp = fe_sqr(fr, xp);
  }
  return acc;
}

/* Witness::statement, src/poly_commit/types.rs:137-151: coefficients zero-padded to n; h_gen_xy =
 * affine h, group generator.  rx = r, x. */
int bpo_poly_statement(int curve, const uint64_t* gx, const uint64_t* gy, const uint64_t* h_gen_xy, const uint64_t* coeffs,
                       size_t n_coeffs, size_t n, const uint64_t* rx, int threads, uint64_t out_commit[12], uint64_t out_eval[4]) {
  CURVE_OR_FAIL(curve); const curve_t* cv = &CURVES[curve];
  const field_t* fr = &cv->fr;
  if (n_coeffs > n) return -1;
  fe* c = (fe*)calloc(n ? n : 1, sizeof(fe));
  if (!c) return -4;
  memcpy(c, coeffs, n_coeffs * 32);                                     /* :139-140 */
  aff h = aff_from_xy8(h_gen_xy);
  fe r, x; memcpy(&r, rx, 32); memcpy(&x, rx + 4, 32);
  jac g; int rc = msm_fe(cv, (const fe*)gx, (const fe*)gy, NULL, c, n, threads, &g);   /* :141 */
  if (!rc) {
    jac bl = smul(cv, r, &h); g = jac_add(&cv->fq, &g, &bl);            /* :142-143 */
    jac_out(&cv->fq, &g, out_commit);
    fe ev = {{0, 0, 0, 0}}, xp = fr->one;                               /* p.evaluate(&x) :145 */
    for (size_t i = 0; i < n_coeffs; i++) { ev = fe_add(fr, ev, fe_mul(fr, c[i], xp)); xp = fe_mul(fr, xp, x); }
    memcpy(out_eval, ev.l, 32);
  }
  free(c);
  return rc;
}

/* poly_commit::prove, src/poly_commit/mod.rs:45-119.  Challenges: u_coeff first, then one u_j per
 * round; randomness: (l_j, r_j) per round from one stream.  out_l, out_r: k x 12 u64; out_ar = a[0], r;
 * out_g = folded generator x | y | infinity (9 u64); out_ui: the k challenges.  Returns -7 if the
 * reference's assertion at :108-113 (g[0] == <h_poly.coeffs(), gs>) would fail. */
int bpo_poly_prove(int curve, const uint64_t* gx, const uint64_t* gy, const uint64_t* h_gen_xy, const uint64_t* coeffs,
                   size_t n_coeffs, size_t n, const uint64_t* rx, uint64_t seed_rand, uint64_t seed_chal, int threads,
                   uint64_t* out_l, uint64_t* out_r, uint64_t out_ar[8], uint64_t out_g[9], uint64_t* out_ui) {
  CURVE_OR_FAIL(curve); const curve_t* cv = &CURVES[curve];
  const field_t* fq = &cv->fq; const field_t* fr = &cv->fr;
  if (n == 0 || (n & (n - 1)) || n_coeffs > n) return -1;
  const size_t k = log2_exact(n);
  sm64 rng = {seed_rand};
  chal_t t = {seed_chal, 0};
  aff h = aff_from_xy8(h_gen_xy), gen = aff_from_xy8(h_gen_xy + 8);
  fe r, x; memcpy(&r, rx, 32); memcpy(&x, rx + 4, 32);
  jac uj = smul(cv, chal_next(&t, fr), &gen);                           /* :55-58 */
  aff u = aff_from_jac(fq, &uj);
  fe *g_x = (fe*)malloc(n * 32), *g_y = (fe*)malloc(n * 32), *a = (fe*)calloc(n, 32), *b = (fe*)malloc(n * 32);
  fe *ui = (fe*)malloc((k ? k : 1) * 32), *uinv = (fe*)malloc((k ? k : 1) * 32);
  uint8_t* gi = (uint8_t*)calloc(n, 1);
  int rc = 0;
  if (!g_x || !g_y || !a || !b || !ui || !uinv || !gi) { rc = -4; goto done; }
  memcpy(g_x, gx, n * 32); memcpy(g_y, gy, n * 32);                     /* :63 */
  memcpy(a, coeffs, n_coeffs * 32);                                     /* :64 (zero-padded to the CRS size) */
  b[0] = fr->one;
  for (size_t i = 1; i < n; i++) b[i] = fe_mul(fr, b[i - 1], x);        /* :65 */
  size_t len = n, round = 0;
  while (len != 1) {                                                    /* :69-105 */
    len /= 2;
    fe l_j = sm64_field(&rng, fr), r_j = sm64_field(&rng, fr);          /* :75 */
    jac m, left, right, tmp;
    if ((rc = msm_fe(cv, g_x + len, g_y + len, gi + len, a, len, threads, &m))) goto done;   /* :78 */
    tmp = smul(cv, l_j, &h); left = jac_add(fq, &m, &tmp);              /* :79 */
    tmp = smul(cv, dot(fr, a, b + len, len), &u); left = jac_add(fq, &left, &tmp);           /* :80 */
    if ((rc = msm_fe(cv, g_x, g_y, gi, a + len, len, threads, &m))) goto done;               /* :83 */
    tmp = smul(cv, r_j, &h); right = jac_add(fq, &m, &tmp);             /* :84 */
    tmp = smul(cv, dot(fr, a + len, b, len), &u); right = jac_add(fq, &right, &tmp);         /* :85 */
    jac_out(fq, &left, out_l + 12 * round); jac_out(fq, &right, out_r + 12 * round);         /* :89 */
    fe u_j = chal_next(&t, fr), u_inv = fe_inv(fr, u_j);                /* :91-93 */
    ui[round] = u_j; uinv[round] = u_inv;
    fold_scalars(fr, a + len, a, len, u_inv, u_j, a, threads);          /* :96 */
    fold_scalars(fr, b, b + len, len, u_inv, u_j, b, threads);          /* :97 */
    fold_generators(cv, g_x, g_y, gi, g_x + len, g_y + len, gi + len, len, u_inv, u_j, g_x, g_y, gi, threads);   /* :102 */
    r = fe_add(fr, r, fe_add(fr, fe_mul(fr, l_j, fe_sqr(fr, u_j)), fe_mul(fr, r_j, fe_sqr(fr, u_inv))));      /* :104 */
    round++;
  }
  memcpy(out_ar, a[0].l, 32); memcpy(out_ar + 4, r.l, 32);              /* :107 */
  memcpy(out_g, g_x[0].l, 32); memcpy(out_g + 4, g_y[0].l, 32); out_g[8] = gi[0];
  memcpy(out_ui, ui, k * 32);
  {                                                                     /* :108-113 */
    fe* ss = (fe*)malloc(n * 32);
    if (!ss) { rc = -4; goto done; }
    s_vector(fr, ui, uinv, k, ss, threads);
    jac gc;
    rc = msm_fe(cv, (const fe*)gx, (const fe*)gy, NULL, ss, n, threads, &gc);
    free(ss);
    if (rc) goto done;
    aff ga = aff_from_jac(fq, &gc);
    if (ga.inf != gi[0] || (!ga.inf && (!fe_equal(&ga.x, &g_x[0]) || !fe_equal(&ga.y, &g_y[0])))) rc = -7;
  }
done:
  free(g_x); free(g_y); free(a); free(b); free(ui); free(uinv); free(gi);
  return rc;
}

/* poly_commit::verify (src/poly_commit/mod.rs:121-173) when g_assumed == NULL, lazy_verify
 * (:264-315) otherwise (g_assumed = x | y | infinity, 9 u64).  Returns `res` of :166 / :304 (the
 * identity iff accepted) and the challenges u_j (the deferred check's h_poly). */
int bpo_poly_verify(int curve, const uint64_t* gx, const uint64_t* gy, const uint64_t* h_gen_xy, size_t n,
                    const uint64_t commit_xyz[12], const uint64_t* x_eval /* x, evaluation */, const uint64_t* L,
                    const uint64_t* R, const uint64_t ar[8], const uint64_t* g_assumed, uint64_t seed_chal, int threads,
                    uint64_t out_res[12], uint64_t* out_ui) {
  CURVE_OR_FAIL(curve); const curve_t* cv = &CURVES[curve];
  const field_t* fq = &cv->fq; const field_t* fr = &cv->fr;
  if (n == 0 || (n & (n - 1))) return -1;
  const size_t k = log2_exact(n);                                       /* :129-130 */
  chal_t t = {seed_chal, 0};
  aff h = aff_from_xy8(h_gen_xy), gen = aff_from_xy8(h_gen_xy + 8);
  fe x, ev, a, r; memcpy(&x, x_eval, 32); memcpy(&ev, x_eval + 4, 32); memcpy(&a, ar, 32); memcpy(&r, ar + 4, 32);
  jac uj = smul(cv, chal_next(&t, fr), &gen);                           /* :132-135 */
  aff u = aff_from_jac(fq, &uj);
  jac q = jac_from_xyz(fq, commit_xyz), tmp = smul(cv, ev, &u);
  q = jac_add(fq, &q, &tmp);                                            /* :137 */
  fe *ui = (fe*)malloc((k ? k : 1) * 32), *uinv = (fe*)malloc((k ? k : 1) * 32);
  if (!ui || !uinv) { free(ui); free(uinv); return -4; }
  for (size_t j = 0; j < k; j++) {                                      /* :139-150 */
    ui[j] = chal_next(&t, fr); uinv[j] = fe_inv(fr, ui[j]);
    jac l = jac_from_xyz(fq, L + 12 * j), rr = jac_from_xyz(fq, R + 12 * j);
    jac a1 = smul_jac(cv, fe_sqr(fr, ui[j]), &l), a2 = smul_jac(cv, fe_sqr(fr, uinv[j]), &rr);
    q = jac_add(fq, &q, &a1); q = jac_add(fq, &q, &a2);
  }
  int rc = 0;
  jac g; fe b;
  if (!g_assumed) {
    fe* ss = (fe*)malloc(2 * n * 32);
    if (!ss) { free(ui); free(uinv); return -4; }
    s_vector(fr, ui, uinv, k, ss, threads);                             /* :152-157 */
    rc = msm_fe(cv, (const fe*)gx, (const fe*)gy, NULL, ss, n, threads, &g);   /* :159 */
    fe* pw = ss + n; pw[0] = fr->one;
    for (size_t i = 1; i < n; i++) pw[i] = fe_mul(fr, pw[i - 1], x);
    b = dot(fr, ss, pw, n);                                             /* :160-163 */
    free(ss);
  } else {
    aff ga; memcpy(&ga.x, g_assumed, 32); memcpy(&ga.y, g_assumed + 4, 32); ga.inf = g_assumed[8] != 0;
    g = jac_from_aff(fq, &ga);
    b = hpoly_evaluate(fr, ui, k, x);                                   /* :301 */
  }
  if (!rc) {
    jac res = smul_jac(cv, a, &g);                                      /* :166 / :304 */
    tmp = smul(cv, r, &h); res = jac_add(fq, &res, &tmp);
    tmp = smul(cv, fe_mul(fr, a, b), &u); res = jac_add(fq, &res, &tmp);
    tmp = jac_neg(fq, &q); res = jac_add(fq, &res, &tmp);
    jac_out(fq, &res, out_res);
    if (out_ui) memcpy(out_ui, ui, k * 32);
  }
  free(ui); free(uinv);
  return rc;
}

/* fold_todos_witness (src/poly_commit/mod.rs:228-244) and fold_todos_statement (:246-262): `count`
 * deferred checks of k challenges each (uis: count x k scalars, gs_xyi: count x 9 u64 x|y|inf);
 * out_coeffs: 2^k scalars = sum_i alpha^i coeffs(h_poly_i); out_commit = sum_i alpha^i g_i;
 * out_eval = sum_i alpha^i h_poly_i(x).  alpha_x = alpha, x. */
int bpo_poly_fold_todos(int curve, const uint64_t* uis, const uint64_t* gs_xyi, size_t count, size_t k,
                        const uint64_t* alpha_x, int threads, uint64_t* out_coeffs, uint64_t out_commit[12], uint64_t out_eval[4]) {
  CURVE_OR_FAIL(curve); const curve_t* cv = &CURVES[curve];
  const field_t* fq = &cv->fq; const field_t* fr = &cv->fr;
  if (count == 0) return -1;
  const size_t n = (size_t)1 << k;
  fe alpha, x; memcpy(&alpha, alpha_x, 32); memcpy(&x, alpha_x + 4, 32);
  fe *ss = (fe*)malloc(n * 32), *uinv = (fe*)malloc((k ? k : 1) * 32), *acc = (fe*)calloc(n, 32);
  if (!ss || !uinv || !acc) { free(ss); free(uinv); free(acc); return -4; }
  fe ai = fr->one, ev = {{0, 0, 0, 0}};
  jac com = jac_identity(fq);
  for (size_t i = 0; i < count; i++) {
    const fe* ui = (const fe*)uis + i * k;
    for (size_t j = 0; j < k; j++) uinv[j] = ui[j];
    batch_inverse(fr, uinv, k);                                         /* :195-199 */
    s_vector(fr, ui, uinv, k, ss, threads);                             /* :237 */
    for (size_t e = 0; e < n; e++) acc[e] = fe_add(fr, acc[e], fe_mul(fr, ss[e], ai));   /* :240-242 */
    aff ga; memcpy(&ga.x, gs_xyi + 9 * i, 32); memcpy(&ga.y, gs_xyi + 9 * i + 4, 32); ga.inf = gs_xyi[9 * i + 8] != 0;
    jac gj = smul(cv, ai, &ga);
    com = jac_add(fq, &com, &gj);                                       /* :253-260 */
    ev = fe_add(fr, ev, fe_mul(fr, ai, hpoly_evaluate(fr, ui, k, x)));
    ai = fe_mul(fr, ai, alpha);
  }
  memcpy(out_coeffs, acc, n * 32);
  jac_out(fq, &com, out_commit);
  memcpy(out_eval, ev.l, 32);
  free(ss); free(uinv); free(acc);
  return 0;
}
