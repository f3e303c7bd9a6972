/* bp_oracle_arith.h -- field and curve arithmetic shared by the CPU oracle's
 * translation units (bp_oracle.c: kernels; bp_oracle_proto.c: protocol drivers).
 * TEST INFRASTRUCTURE, never shipped or linked into the product; see bp_oracle.c
 * for what pins it.  Everything here is file-local (static). */
#ifndef BP_ORACLE_ARITH_H
#define BP_ORACLE_ARITH_H
#if defined(__GNUC__)
#pragma GCC diagnostic ignored "-Wunused-function"
#endif
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <x86intrin.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe;
typedef struct { uint64_t p[4]; uint64_t ninv; fe one; fe r2; } field_t;
typedef struct { fe X, Y, Z; } jac;   /* identity: Z = 0 */
typedef struct { field_t fq, fr; } curve_t;

static curve_t CURVES[2];
static int INITED = 0;

/* ---- field setup ---------------------------------------------------------- */
static int geq(const uint64_t a[4], const uint64_t b[4]) {
  for (int i = 3; i >= 0; i--) { if (a[i] > b[i]) return 1; if (a[i] < b[i]) return 0; }
  return 1;
}
static void sub_n(uint64_t a[4], const uint64_t b[4]) {
  u128 br = 0;
  for (int i = 0; i < 4; i++) { u128 d = (u128)a[i] - b[i] - (uint64_t)br; a[i] = (uint64_t)d; br = (d >> 64) & 1; }
}
/* a = 2a mod p (used only to derive R and R^2 at start-up) */
static void dbl_mod(uint64_t a[4], const uint64_t p[4]) {
  uint64_t c = a[3] >> 63;
  for (int i = 3; i > 0; i--) a[i] = (a[i] << 1) | (a[i - 1] >> 63);
  a[0] <<= 1;
  if (c || geq(a, p)) sub_n(a, p);
}
static void field_init(field_t* f, const uint64_t p[4]) {
  memcpy(f->p, p, 32);
  uint64_t inv = 1;                       /* Newton iteration for p^-1 mod 2^64 */
  for (int i = 0; i < 6; i++) inv *= 2 - p[0] * inv;
  f->ninv = (uint64_t)0 - inv;
  uint64_t t[4] = {1, 0, 0, 0};
  for (int i = 0; i < 256; i++) dbl_mod(t, p);   /* 2^256 mod p */
  memcpy(f->one.l, t, 32);
  for (int i = 0; i < 256; i++) dbl_mod(t, p);   /* 2^512 mod p */
  memcpy(f->r2.l, t, 32);
}
static void init_all(void) {
  if (INITED) return;
  static const uint64_t SECP_P[4] = {0xFFFFFFFEFFFFFC2Full, 0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFFFFFFFFFFull};
  static const uint64_t SECP_N[4] = {0xBFD25E8CD0364141ull, 0xBAAEDCE6AF48A03Bull, 0xFFFFFFFFFFFFFFFEull, 0xFFFFFFFFFFFFFFFFull};
  static const uint64_t VESTA_P[4] = {0x8C46EB2100000001ull, 0x224698FC0994A8DDull, 0x0ull, 0x4000000000000000ull};
  static const uint64_t VESTA_N[4] = {0x992D30ED00000001ull, 0x224698FC094CF91Bull, 0x0ull, 0x4000000000000000ull};
  field_init(&CURVES[0].fq, SECP_P); field_init(&CURVES[0].fr, SECP_N);
  field_init(&CURVES[1].fq, VESTA_P); field_init(&CURVES[1].fr, VESTA_N);
  INITED = 1;
}

/* ---- field arithmetic ------------------------------------------------------- */
static inline int fe_is_zero(const fe* a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fe_equal(const fe* a, const fe* b) { return memcmp(a, b, 32) == 0; }
/* add and subtract without data-dependent branches (carry chains + masked correction) */
static inline fe fe_add(const field_t* f, fe a, fe b) {
  unsigned long long r0, r1, r2, r3, s0, s1, s2, s3;
  unsigned char c = 0, br = 0;
  c = _addcarry_u64(c, a.l[0], b.l[0], &r0); c = _addcarry_u64(c, a.l[1], b.l[1], &r1);
  c = _addcarry_u64(c, a.l[2], b.l[2], &r2); c = _addcarry_u64(c, a.l[3], b.l[3], &r3);
  br = _subborrow_u64(br, r0, f->p[0], &s0); br = _subborrow_u64(br, r1, f->p[1], &s1);
  br = _subborrow_u64(br, r2, f->p[2], &s2); br = _subborrow_u64(br, r3, f->p[3], &s3);
  uint64_t keep = (uint64_t)0 - (uint64_t)((c == 0) & (br != 0));   /* all ones: the sum was already below p */
  fe o = {{(r0 & keep) | (s0 & ~keep), (r1 & keep) | (s1 & ~keep), (r2 & keep) | (s2 & ~keep), (r3 & keep) | (s3 & ~keep)}};
  return o;
}
static inline fe fe_sub(const field_t* f, fe a, fe b) {
  unsigned long long r0, r1, r2, r3;
  unsigned char br = 0, c = 0;
  br = _subborrow_u64(br, a.l[0], b.l[0], &r0); br = _subborrow_u64(br, a.l[1], b.l[1], &r1);
  br = _subborrow_u64(br, a.l[2], b.l[2], &r2); br = _subborrow_u64(br, a.l[3], b.l[3], &r3);
  uint64_t m = (uint64_t)0 - (uint64_t)br;             /* add p back after a borrow */
  c = _addcarry_u64(c, r0, f->p[0] & m, &r0); c = _addcarry_u64(c, r1, f->p[1] & m, &r1);
  c = _addcarry_u64(c, r2, f->p[2] & m, &r2); c = _addcarry_u64(c, r3, f->p[3] & m, &r3);
  fe o = {{r0, r1, r2, r3}};
  return o;
}
static inline fe fe_neg(const field_t* f, fe a) { fe z = {{0, 0, 0, 0}}; return fe_sub(f, z, a); }
/* Montgomery product, operand-scanning form */
static inline fe fe_mul(const field_t* f, fe a, fe b) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) { c += (u128)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * f->ninv;
    c = ((u128)m * f->p[0] + t[0]) >> 64;
    for (int j = 1; j < 4; j++) { c += (u128)m * f->p[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
  }
  fe r; memcpy(r.l, t, 32);
  if (t[4] || geq(r.l, f->p)) sub_n(r.l, f->p);
  return r;
}
static inline fe fe_sqr(const field_t* f, fe a) { return fe_mul(f, a, a); }
static fe fe_from_mont(const field_t* f, fe a) { fe one = {{1, 0, 0, 0}}; return fe_mul(f, a, one); }
static fe fe_inv(const field_t* f, fe a) {          /* a^(p-2); 0 -> 0 */
  uint64_t e[4]; memcpy(e, f->p, 32);
  uint64_t two[4] = {2, 0, 0, 0}; sub_n(e, two);
  fe r = f->one;
  for (int i = 255; i >= 0; i--) {
    r = fe_sqr(f, r);
    if ((e[i >> 6] >> (i & 63)) & 1) r = fe_mul(f, r, a);
  }
  return r;
}

/* ---- curve arithmetic (a = 0), Jacobian ---------------------------------------- */
static jac jac_identity(const field_t* f) { jac r; r.X = f->one; r.Y = f->one; memset(&r.Z, 0, 32); return r; }
static jac jac_double(const field_t* f, const jac* p) {
  if (fe_is_zero(&p->Z)) return *p;
  fe A = fe_sqr(f, p->X), B = fe_sqr(f, p->Y), C = fe_sqr(f, B);
  fe t = fe_add(f, p->X, B);
  fe D = fe_sub(f, fe_sub(f, fe_sqr(f, t), A), C); D = fe_add(f, D, D);
  fe E = fe_add(f, fe_add(f, A, A), A), F = fe_sqr(f, E);
  jac r;
  r.X = fe_sub(f, F, fe_add(f, D, D));
  fe C8 = fe_add(f, C, C); C8 = fe_add(f, C8, C8); C8 = fe_add(f, C8, C8);
  r.Y = fe_sub(f, fe_mul(f, E, fe_sub(f, D, r.X)), C8);
  fe yz = fe_mul(f, p->Y, p->Z); r.Z = fe_add(f, yz, yz);
  return r;
}
static jac jac_add_affine(const field_t* f, const jac* p, const fe* qx, const fe* qy) {
  if (fe_is_zero(&p->Z)) { jac r; r.X = *qx; r.Y = *qy; r.Z = f->one; return r; }
  fe Z1Z1 = fe_sqr(f, p->Z), U2 = fe_mul(f, *qx, Z1Z1), S2 = fe_mul(f, fe_mul(f, *qy, p->Z), Z1Z1);
  if (fe_equal(&U2, &p->X)) { if (fe_equal(&S2, &p->Y)) return jac_double(f, p); return jac_identity(f); }
  fe H = fe_sub(f, U2, p->X), HH = fe_sqr(f, H), I = fe_add(f, HH, HH); I = fe_add(f, I, I);
  fe J = fe_mul(f, H, I), r = fe_sub(f, S2, p->Y); r = fe_add(f, r, r);
  fe V = fe_mul(f, p->X, I);
  jac o;
  o.X = fe_sub(f, fe_sub(f, fe_sqr(f, r), J), fe_add(f, V, V));
  fe YJ = fe_mul(f, p->Y, J);
  o.Y = fe_sub(f, fe_mul(f, r, fe_sub(f, V, o.X)), fe_add(f, YJ, YJ));
  fe zh = fe_add(f, p->Z, H);
  o.Z = fe_sub(f, fe_sub(f, fe_sqr(f, zh), Z1Z1), HH);
  return o;
}
static jac jac_add(const field_t* f, const jac* p, const jac* q) {
  if (fe_is_zero(&q->Z)) return *p;
  if (fe_is_zero(&p->Z)) return *q;
  fe Z1Z1 = fe_sqr(f, p->Z), Z2Z2 = fe_sqr(f, q->Z);
  fe U1 = fe_mul(f, p->X, Z2Z2), U2 = fe_mul(f, q->X, Z1Z1);
  fe S1 = fe_mul(f, fe_mul(f, p->Y, q->Z), Z2Z2), S2 = fe_mul(f, fe_mul(f, q->Y, p->Z), Z1Z1);
  if (fe_equal(&U1, &U2)) { if (fe_equal(&S1, &S2)) return jac_double(f, p); return jac_identity(f); }
  fe H = fe_sub(f, U2, U1), I = fe_add(f, H, H); I = fe_sqr(f, I);
  fe J = fe_mul(f, H, I), r = fe_sub(f, S2, S1); r = fe_add(f, r, r);
  fe V = fe_mul(f, U1, I);
  jac o;
  o.X = fe_sub(f, fe_sub(f, fe_sqr(f, r), J), fe_add(f, V, V));
  fe SJ = fe_mul(f, S1, J);
  o.Y = fe_sub(f, fe_mul(f, r, fe_sub(f, V, o.X)), fe_add(f, SJ, SJ));
  fe zs = fe_add(f, p->Z, q->Z);
  o.Z = fe_mul(f, fe_sub(f, fe_sub(f, fe_sqr(f, zs), Z1Z1), Z2Z2), H);
  return o;
}
/* returns 1 for the identity; x,y canonical affine (Montgomery form) */
static int jac_to_affine(const field_t* f, const jac* p, fe* x, fe* y) {
  if (fe_is_zero(&p->Z)) { memset(x, 0, 32); memset(y, 0, 32); return 1; }
  fe zi = fe_inv(f, p->Z), zi2 = fe_sqr(f, zi);
  *x = fe_mul(f, p->X, zi2); *y = fe_mul(f, fe_mul(f, p->Y, zi2), zi);
  return 0;
}
static void jac_out(const field_t* f, const jac* p, uint64_t out[12]) {
  fe x, y;
  if (jac_to_affine(f, p, &x, &y)) { memcpy(out, f->one.l, 32); memcpy(out + 4, f->one.l, 32); memset(out + 8, 0, 32); return; }
  memcpy(out, x.l, 32); memcpy(out + 4, y.l, 32); memcpy(out + 8, f->one.l, 32);
}
/* k canonical (non-Montgomery) 256-bit integer */
static jac jac_scalar_mul(const field_t* f, const uint64_t k[4], const fe* px, const fe* py, int p_inf) {
  jac acc = jac_identity(f);
  if (p_inf) return acc;
  for (int i = 255; i >= 0; i--) {
    acc = jac_double(f, &acc);
    if ((k[i >> 6] >> (i & 63)) & 1) acc = jac_add_affine(f, &acc, px, py);
  }
  return acc;
}

#define CURVE_OR_FAIL(cv) do { init_all(); if ((cv) < 0 || (cv) > 1) return -2; } while (0)
/* sum_i s_i * P_i by the bucket method (the algorithm behind G::msm_unchecked,
 * src/msm.rs:84-87): window width c = ln(n)+2 (3 below 32 points), windows
 * processed in parallel, each window: 2^c-1 buckets filled by mixed additions,
 * folded by a running sum; windows combined high to low with c doublings.
 * Points as x/y planes (+ optional infinity bytes), scalars in Montgomery form. */
static unsigned ln_without_floats(size_t a) { unsigned l = 0; while (a >>= 1) l++; return l * 69 / 100; }
static int msm_planes(const curve_t* cv, const uint64_t* bx, const uint64_t* by, const uint8_t* binf, const uint64_t* s,
                      size_t n, int threads, jac* out) {
  const field_t* fq = &cv->fq; const field_t* fr = &cv->fr;
  if (threads < 1) threads = 1;
  unsigned c = n < 32 ? 3 : ln_without_floats(n) + 2;
  unsigned n_win = (256 + c - 1) / c;
  uint64_t* ks = (uint64_t*)malloc((n ? n : 1) * 32);
  jac* wins = (jac*)malloc(sizeof(jac) * n_win);
  if (!ks || !wins) { free(ks); free(wins); return -4; }
#pragma omp parallel for num_threads(threads) schedule(static)
  for (long i = 0; i < (long)n; i++) { fe k; memcpy(&k, s + 4 * i, 32); k = fe_from_mont(fr, k); memcpy(ks + 4 * i, k.l, 32); }
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
  for (long w = 0; w < (long)n_win; w++) {
    size_t nb = ((size_t)1 << c) - 1;
    jac* buckets = (jac*)malloc(sizeof(jac) * nb);
    for (size_t b = 0; b < nb; b++) buckets[b] = jac_identity(fq);
    unsigned bit = (unsigned)w * c;
    for (size_t i = 0; i < n; i++) {
      if (binf && binf[i]) continue;
      const uint64_t* k = ks + 4 * i;
      unsigned limb = bit >> 6, sh = bit & 63;
      uint64_t d = k[limb] >> sh;
      if (sh + c > 64 && limb < 3) d |= k[limb + 1] << (64 - sh);
      d &= ((uint64_t)1 << c) - 1;
      if (!d) continue;
      fe px, py; memcpy(&px, bx + 4 * i, 32); memcpy(&py, by + 4 * i, 32);
      buckets[d - 1] = jac_add_affine(fq, &buckets[d - 1], &px, &py);
    }
    jac run = jac_identity(fq), acc = jac_identity(fq);
    for (size_t b = nb; b-- > 0;) { run = jac_add(fq, &run, &buckets[b]); acc = jac_add(fq, &acc, &run); }
    wins[w] = acc;
    free(buckets);
  }
  jac total = jac_identity(fq);
  for (long w = (long)n_win - 1; w >= 0; w--) {
    for (unsigned k = 0; k < c; k++) total = jac_double(fq, &total);
    total = jac_add(fq, &total, &wins[w]);
  }
  *out = total;
  free(ks); free(wins);
  return 0;
}

#endif /* BP_ORACLE_ARITH_H */
