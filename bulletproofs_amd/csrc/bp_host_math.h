// Host-side 256-bit field and curve arithmetic (4 x u64 Montgomery limbs,
// unsigned __int128 products).  Used by the library only for strictly sequential
// epilogues: the ~256-doubling window combine of an MSM and single-point fix-ups.
// It is not a CPU implementation of the data-parallel path.
#pragma once
#include <stdint.h>
#include <string.h>

#include "bp_field_consts.h"

namespace bp {
namespace host {

typedef unsigned __int128 u128;

struct HFe {
  uint64_t v[4];
};

inline HFe h_zero() { return HFe{{0, 0, 0, 0}}; }
template <class F>
inline HFe h_one() {
  return HFe{{F::one64(0), F::one64(1), F::one64(2), F::one64(3)}};
}
inline bool h_is_zero(const HFe& a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; }
inline bool h_eq(const HFe& a, const HFe& b) { return memcmp(a.v, b.v, 32) == 0; }

template <class F>
inline bool h_geq_mod(const HFe& a) {
  for (int i = 3; i >= 0; i--) {
    if (a.v[i] > F::mod64(i)) return true;
    if (a.v[i] < F::mod64(i)) return false;
  }
  return true;
}
template <class F>
inline void h_sub_mod(HFe& a) {
  u128 br = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)a.v[i] - F::mod64(i) - (uint64_t)br;
    a.v[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
}
// a + b + carry-in -> low 64 bits, carry-out in *c (0 or 1); likewise the borrow chain
inline uint64_t h_adc(uint64_t a, uint64_t b, uint64_t* c) {
  uint64_t s, t;
  uint64_t c1 = __builtin_add_overflow(a, b, &s);
  uint64_t c2 = __builtin_add_overflow(s, *c, &t);
  *c = c1 | c2;
  return t;
}
inline uint64_t h_sbb(uint64_t a, uint64_t b, uint64_t* br) {
  uint64_t d, t;
  uint64_t b1 = __builtin_sub_overflow(a, b, &d);
  uint64_t b2 = __builtin_sub_overflow(d, *br, &t);
  *br = b1 | b2;
  return t;
}
// (carry * 2^256 + s) mod p for a value below 2p: one subtraction, chosen without a branch
template <class F>
inline HFe h_reduce_once(const HFe& s, uint64_t carry) {
  HFe d;
  uint64_t br = 0;
  for (int i = 0; i < 4; i++) d.v[i] = h_sbb(s.v[i], F::mod64(i), &br);
  const uint64_t keep = (uint64_t)0 - (uint64_t)((carry == 0) & (br != 0));  // all ones: s < p
  HFe r;
  for (int i = 0; i < 4; i++) r.v[i] = (s.v[i] & keep) | (d.v[i] & ~keep);
  return r;
}
template <class F>
inline HFe h_add(const HFe& a, const HFe& b) {
  HFe s;
  uint64_t c = 0;
  for (int i = 0; i < 4; i++) s.v[i] = h_adc(a.v[i], b.v[i], &c);
  return h_reduce_once<F>(s, c);
}
template <class F>
inline HFe h_sub(const HFe& a, const HFe& b) {
  HFe d, r;
  uint64_t br = 0, c = 0;
  for (int i = 0; i < 4; i++) d.v[i] = h_sbb(a.v[i], b.v[i], &br);
  const uint64_t mask = (uint64_t)0 - br;  // all ones when a < b: add p back
  for (int i = 0; i < 4; i++) r.v[i] = h_adc(d.v[i], F::mod64(i) & mask, &c);
  return r;
}
template <class F>
inline HFe h_dbl(const HFe& a) {
  HFe s;
  s.v[0] = a.v[0] << 1;
  for (int i = 1; i < 4; i++) s.v[i] = (a.v[i] << 1) | (a.v[i - 1] >> 63);
  return h_reduce_once<F>(s, a.v[3] >> 63);
}
template <class F>
inline HFe h_mul(const HFe& a, const HFe& b) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) {
      c += (u128)a.v[j] * b.v[i] + t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[4] = (uint64_t)c;
    t[5] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * F::kInv64;
    c = ((u128)m * F::mod64(0) + t[0]) >> 64;
    for (int j = 1; j < 4; j++) {
      c += (u128)m * F::mod64(j) + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[3] = (uint64_t)c;
    t[4] = t[5] + (uint64_t)(c >> 64);
  }
  HFe r{{t[0], t[1], t[2], t[3]}};
  if (t[4] || h_geq_mod<F>(r)) h_sub_mod<F>(r);
  return r;
}
template <class F>
inline HFe h_sqr(const HFe& a) {
  return h_mul<F>(a, a);
}
template <class F>
inline HFe h_inv(const HFe& a) {  // a^(p-2); 0 -> 0
  HFe r = h_one<F>();
  for (int i = 3; i >= 0; i--) {
    uint64_t w = F::pm264(i);
    for (int b = 63; b >= 0; b--) {
      r = h_sqr<F>(r);
      if ((w >> b) & 1) r = h_mul<F>(r, a);
    }
  }
  return r;
}
// a^-1 by Kaliski's almost-inverse: phase 1 finds a^-1 2^k (256 <= k <= 512) with plain shifts,
// additions and subtractions (nothing is reduced on the way: r and s stay below 2p), phase 2
// multiplies the power of two away.  In and out in Montgomery form; 0 -> 0.  Variable time: for
// public values (challenges, normalisation of results).
template <class F>
inline HFe h_inv_kaliski(const HFe& a) {
  if (h_is_zero(a)) return a;
  typedef unsigned long long ull;
  ull u[4] = {F::mod64(0), F::mod64(1), F::mod64(2), F::mod64(3)}, v[4] = {a.v[0], a.v[1], a.v[2], a.v[3]};
  ull r[5] = {0, 0, 0, 0, 0}, s[5] = {1, 0, 0, 0, 0};
  auto shr1 = [](ull* x) {
    x[0] = (x[0] >> 1) | (x[1] << 63);
    x[1] = (x[1] >> 1) | (x[2] << 63);
    x[2] = (x[2] >> 1) | (x[3] << 63);
    x[3] >>= 1;
  };
  auto shl1 = [](ull* x) {
    x[4] = (x[4] << 1) | (x[3] >> 63);
    x[3] = (x[3] << 1) | (x[2] >> 63);
    x[2] = (x[2] << 1) | (x[1] >> 63);
    x[1] = (x[1] << 1) | (x[0] >> 63);
    x[0] <<= 1;
  };
  auto gt = [](const ull* x, const ull* y) {
    for (int i = 3; i >= 0; i--)
      if (x[i] != y[i]) return x[i] > y[i];
    return false;
  };
  auto subn = [](ull* x, const ull* y, int n) {  // x -= y over n limbs
    ull b = 0;
    for (int i = 0; i < n; i++) {
      u128 d = (u128)x[i] - y[i] - b;
      x[i] = (ull)d;
      b = (ull)(d >> 64) & 1;
    }
  };
  auto sub4 = [&subn](ull* x, const ull* y) { subn(x, y, 4); };
  auto add5 = [](ull* x, const ull* y) {
    u128 c = 0;
    for (int i = 0; i < 5; i++) {
      c += (u128)x[i] + y[i];
      x[i] = (ull)c;
      c >>= 64;
    }
  };
  int k = 0;
  while (v[0] | v[1] | v[2] | v[3]) {
    if (!(u[0] & 1)) {
      shr1(u);
      shl1(s);
    } else if (!(v[0] & 1)) {
      shr1(v);
      shl1(r);
    } else if (gt(u, v)) {
      sub4(u, v);
      shr1(u);
      add5(r, s);
      shl1(s);
    } else {
      sub4(v, u);
      shr1(v);
      add5(s, r);
      shl1(r);
    }
    k++;
  }
  // r < 2p: a^-1 2^k = p - (r mod p)
  const ull p[5] = {F::mod64(0), F::mod64(1), F::mod64(2), F::mod64(3), 0};
  auto geq5 = [](const ull* x, const ull* y) {
    for (int i = 4; i >= 0; i--)
      if (x[i] != y[i]) return x[i] > y[i];
    return true;
  };
  if (geq5(r, p)) subn(r, p, 5);
  ull pr[4] = {p[0], p[1], p[2], p[3]};
  sub4(pr, r);
  HFe x{{pr[0], pr[1], pr[2], pr[3]}};
  // x = (aR)^-1 2^k; wanted a^-1 R = x 2^(512 - k): Montgomery-multiply by 2^m R with m = 512 - k
  // (h_mul(x, y) = x y / R), in steps of at most 250 bits so that 2^step is below every modulus
  const HFe r2{{F::r264(0), F::r264(1), F::r264(2), F::r264(3)}};
  int m = 512 - k;
  while (true) {
    int step = m > 250 ? 250 : m;
    HFe pw{{0, 0, 0, 0}};
    pw.v[step >> 6] = 1ull << (step & 63);
    x = h_mul<F>(x, h_mul<F>(pw, r2));
    m -= step;
    if (m == 0) break;
  }
  return x;
}

template <class F>
inline HFe h_from_mont(const HFe& a) {
  HFe one{{1, 0, 0, 0}};
  return h_mul<F>(a, one);
}
template <class F>
inline HFe h_neg(const HFe& a) {
  if (h_is_zero(a)) return a;
  HFe m{{F::mod64(0), F::mod64(1), F::mod64(2), F::mod64(3)}};
  u128 br = 0;
  HFe r;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)m.v[i] - a.v[i] - (uint64_t)br;
    r.v[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
  return r;
}

struct HXyzz {
  HFe x, y, zz, zzz;
};

template <class F>
inline HXyzz hx_identity() {
  return HXyzz{h_one<F>(), h_one<F>(), h_zero(), h_zero()};
}
inline bool hx_is_identity(const HXyzz& p) { return h_is_zero(p.zz); }

template <class F>
inline HXyzz hx_dbl(const HXyzz& p) {
  if (hx_is_identity(p) || h_is_zero(p.y)) return hx_identity<F>();
  HFe u = h_dbl<F>(p.y);
  HFe v = h_sqr<F>(u);
  HFe w = h_mul<F>(u, v);
  HFe s = h_mul<F>(p.x, v);
  HFe xx = h_sqr<F>(p.x);
  HFe m = h_add<F>(h_dbl<F>(xx), xx);
  HXyzz r;
  r.x = h_sub<F>(h_sqr<F>(m), h_dbl<F>(s));
  r.y = h_sub<F>(h_mul<F>(m, h_sub<F>(s, r.x)), h_mul<F>(w, p.y));
  r.zz = h_mul<F>(v, p.zz);
  r.zzz = h_mul<F>(w, p.zzz);
  return r;
}

template <class F>
inline HXyzz hx_add(const HXyzz& p, const HXyzz& q) {
  if (hx_is_identity(q)) return p;
  if (hx_is_identity(p)) return q;
  HFe u1 = h_mul<F>(p.x, q.zz), u2 = h_mul<F>(q.x, p.zz);
  HFe s1 = h_mul<F>(p.y, q.zzz), s2 = h_mul<F>(q.y, p.zzz);
  HFe pd = h_sub<F>(u2, u1), rd = h_sub<F>(s2, s1);
  if (h_is_zero(pd)) return h_is_zero(rd) ? hx_dbl<F>(p) : hx_identity<F>();
  HFe pp = h_sqr<F>(pd), ppp = h_mul<F>(pd, pp), qq = h_mul<F>(u1, pp);
  HXyzz r;
  r.x = h_sub<F>(h_sub<F>(h_sqr<F>(rd), ppp), h_dbl<F>(qq));
  r.y = h_sub<F>(h_mul<F>(rd, h_sub<F>(qq, r.x)), h_mul<F>(s1, ppp));
  r.zz = h_mul<F>(h_mul<F>(p.zz, q.zz), pp);
  r.zzz = h_mul<F>(h_mul<F>(p.zzz, q.zzz), ppp);
  return r;
}

// -> normalised Jacobian (x, y, 1) or (1, 1, 0); 12 u64
template <class F>
inline void hx_to_xyz(const HXyzz& p, uint64_t out[12]) {
  HFe one = h_one<F>();
  if (hx_is_identity(p)) {
    memcpy(out, one.v, 32);
    memcpy(out + 4, one.v, 32);
    memset(out + 8, 0, 32);
    return;
  }
  HFe i3 = h_inv_kaliski<F>(p.zzz);
  HFe t = h_mul<F>(p.zz, i3);
  HFe i2 = h_sqr<F>(t);
  HFe x = h_mul<F>(p.x, i2), y = h_mul<F>(p.y, i3);
  memcpy(out, x.v, 32);
  memcpy(out + 4, y.v, 32);
  memcpy(out + 8, one.v, 32);
}

// sum_w 2^(c*w) * win[w], win = n_windows XYZZ points of 16 u64 each
template <class F>
inline HXyzz hx_combine_windows(const uint64_t* win, int n_windows, int c) {
  HXyzz acc = hx_identity<F>();
  for (int w = n_windows - 1; w >= 0; w--) {
    if (!hx_is_identity(acc))
      for (int k = 0; k < c; k++) acc = hx_dbl<F>(acc);
    HXyzz p;
    memcpy(&p, win + (size_t)w * 16, 128);
    acc = hx_add<F>(acc, p);
  }
  return acc;
}

// Two independent combinations in lockstep: the two dependency chains interleave in the
// core's out-of-order window, which nearly doubles the throughput of this latency-bound loop.
template <class F>
inline void hx_combine_windows2(const uint64_t* win_a, const uint64_t* win_b, int n_windows, int c, HXyzz* out_a,
                                HXyzz* out_b) {
  HXyzz a = hx_identity<F>(), b = hx_identity<F>();
  for (int w = n_windows - 1; w >= 0; w--) {
    for (int k = 0; k < c; k++) {
      a = hx_dbl<F>(a);
      b = hx_dbl<F>(b);
    }
    HXyzz pa, pb;
    memcpy(&pa, win_a + (size_t)w * 16, 128);
    memcpy(&pb, win_b + (size_t)w * 16, 128);
    a = hx_add<F>(a, pa);
    b = hx_add<F>(b, pb);
  }
  *out_a = a;
  *out_b = b;
}

// ---- window combination on Jacobian coordinates -------------------------------------------
// The combine is a chain of (256 - c) doublings: its time is the host's share of every MSM.  A
// Jacobian doubling with a = 0 is 2M + 5S against the 6M + 3S of an XYZZ doubling, and for
// secp256k1 the chain runs on plain residues, where p = 2^256 - 0x1000003D1 turns the reduction of
// a 512-bit product into two multiplications by a 33-bit constant and a squaring needs 10 limb
// products instead of 16.  The window sums arrive as XYZZ points in Montgomery form; the result
// leaves as a normalised Montgomery-form point, as before.

// field operations of the chain: generic Montgomery form
template <class F>
struct MontOps {
  typedef F Field;
  static HFe mul(const HFe& a, const HFe& b) { return h_mul<F>(a, b); }
  static HFe sqr(const HFe& a) { return h_mul<F>(a, a); }
  static HFe import(const HFe& mont) { return mont; }
  static HFe to_mont(const HFe& v) { return v; }
};

// secp256k1 base field on plain residues
struct SecpPlainOps {
  typedef Secp256k1Fq Field;
  static HFe reduce512(const uint64_t t[8]) {
    const uint64_t C = 0x1000003D1ull;  // 2^256 mod p
    uint64_t r[4];
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
      c += (u128)t[4 + i] * C + t[i];
      r[i] = (uint64_t)c;
      c >>= 64;
    }
    // what is left above 2^256 (below 2^34) folds once more, and a carry out of that a last time
    c = (u128)(uint64_t)c * C + r[0];
    r[0] = (uint64_t)c;
    c >>= 64;
    for (int i = 1; i < 4; i++) {
      c += r[i];
      r[i] = (uint64_t)c;
      c >>= 64;
    }
    if (c) {
      c = (u128)r[0] + C;
      r[0] = (uint64_t)c;
      c >>= 64;
      for (int i = 1; i < 4; i++) {
        c += r[i];
        r[i] = (uint64_t)c;
        c >>= 64;
      }
    }
    HFe out{{r[0], r[1], r[2], r[3]}};
    if (h_geq_mod<Secp256k1Fq>(out)) h_sub_mod<Secp256k1Fq>(out);
    return out;
  }
  static HFe mul(const HFe& a, const HFe& b) {
    uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
      u128 c = 0;
      for (int j = 0; j < 4; j++) {
        c += (u128)a.v[j] * b.v[i] + t[i + j];
        t[i + j] = (uint64_t)c;
        c >>= 64;
      }
      t[i + 4] = (uint64_t)c;
    }
    return reduce512(t);
  }
  static HFe sqr(const HFe& a) {
    uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    u128 c;
    // the six cross products once ...
    c = (u128)a.v[0] * a.v[1];
    t[1] = (uint64_t)c;
    c >>= 64;
    c += (u128)a.v[0] * a.v[2];
    t[2] = (uint64_t)c;
    c >>= 64;
    c += (u128)a.v[0] * a.v[3];
    t[3] = (uint64_t)c;
    t[4] = (uint64_t)(c >> 64);
    c = (u128)a.v[1] * a.v[2] + t[3];
    t[3] = (uint64_t)c;
    c >>= 64;
    c += (u128)a.v[1] * a.v[3] + t[4];
    t[4] = (uint64_t)c;
    t[5] = (uint64_t)(c >> 64);
    c = (u128)a.v[2] * a.v[3] + t[5];
    t[5] = (uint64_t)c;
    t[6] = (uint64_t)(c >> 64);
    // ... doubled ...
    t[7] = t[6] >> 63;
    for (int i = 6; i > 1; i--) t[i] = (t[i] << 1) | (t[i - 1] >> 63);
    t[1] <<= 1;
    // ... plus the four squares
    c = 0;
    for (int i = 0; i < 4; i++) {
      const u128 sq = (u128)a.v[i] * a.v[i];
      c += (u128)t[2 * i] + (uint64_t)sq;
      t[2 * i] = (uint64_t)c;
      c >>= 64;
      c += (u128)t[2 * i + 1] + (uint64_t)(sq >> 64);
      t[2 * i + 1] = (uint64_t)c;
      c >>= 64;
    }
    return reduce512(t);
  }
  static HFe import(const HFe& mont) { return h_from_mont<Secp256k1Fq>(mont); }
  static HFe to_mont(const HFe& v) {
    const HFe r2{{Secp256k1Fq::r264(0), Secp256k1Fq::r264(1), Secp256k1Fq::r264(2), Secp256k1Fq::r264(3)}};
    return h_mul<Secp256k1Fq>(v, r2);
  }
};

struct HJac {
  HFe x, y, z;  // z = 0: identity
};

// dbl-2009-l (a = 0): 2M + 5S
template <class O>
inline HJac hj_dbl(const HJac& p) {
  typedef typename O::Field F;
  HFe a = O::sqr(p.x), b = O::sqr(p.y), c = O::sqr(b);
  HFe t = h_sub<F>(h_sub<F>(O::sqr(h_add<F>(p.x, b)), a), c);
  HFe d = h_dbl<F>(t);
  HFe e = h_add<F>(h_dbl<F>(a), a);
  HFe f = O::sqr(e);
  HJac r;
  r.x = h_sub<F>(f, h_dbl<F>(d));
  HFe c8 = h_dbl<F>(h_dbl<F>(h_dbl<F>(c)));
  r.y = h_sub<F>(O::mul(e, h_sub<F>(d, r.x)), c8);
  r.z = h_dbl<F>(O::mul(p.y, p.z));  // 0 stays 0: the identity doubles to itself
  return r;
}

template <class O>
inline HJac hj_add(const HJac& p, const HJac& q) {
  typedef typename O::Field F;
  if (h_is_zero(q.z)) return p;
  if (h_is_zero(p.z)) return q;
  HFe z1z1 = O::sqr(p.z), z2z2 = O::sqr(q.z);
  HFe u1 = O::mul(p.x, z2z2), u2 = O::mul(q.x, z1z1);
  HFe s1 = O::mul(O::mul(p.y, q.z), z2z2), s2 = O::mul(O::mul(q.y, p.z), z1z1);
  HFe h = h_sub<F>(u2, u1), r = h_sub<F>(s2, s1);
  if (h_is_zero(h)) {
    if (h_is_zero(r)) return hj_dbl<O>(p);
    return HJac{p.x, p.y, h_zero()};
  }
  HFe hh = O::sqr(h), hhh = O::mul(h, hh), v = O::mul(u1, hh);
  HJac o;
  o.x = h_sub<F>(h_sub<F>(O::sqr(r), hhh), h_dbl<F>(v));
  o.y = h_sub<F>(O::mul(r, h_sub<F>(v, o.x)), O::mul(s1, hhh));
  o.z = O::mul(O::mul(p.z, q.z), h);
  return o;
}

// XYZZ (x, y, zz = z^2, zzz = z^3, Montgomery form) -> the Jacobian representative with Z = z^3:
// scaled by lambda = z^2 the point reads (x zz^2, y zz^3, zzz), all from known values
template <class O>
inline HJac hj_from_xyzz(const uint64_t* w) {
  HFe x, y, zz, zzz;
  memcpy(x.v, w, 32);
  memcpy(y.v, w + 4, 32);
  memcpy(zz.v, w + 8, 32);
  memcpy(zzz.v, w + 12, 32);
  if (h_is_zero(zz)) return HJac{h_zero(), h_zero(), h_zero()};
  x = O::import(x);
  y = O::import(y);
  zz = O::import(zz);
  zzz = O::import(zzz);
  HFe t = O::sqr(zz);
  return HJac{O::mul(x, t), O::mul(O::mul(y, t), zz), zzz};
}

// sum_w 2^(c*w) * win[w] -> normalised Jacobian (x, y, 1) or (1, 1, 0) in Montgomery form
template <class O>
inline void hj_combine_windows(const uint64_t* win, int n_windows, int c, uint64_t out[12]) {
  typedef typename O::Field F;
  HJac acc{h_zero(), h_zero(), h_zero()};
  for (int w = n_windows - 1; w >= 0; w--) {
    if (!h_is_zero(acc.z))
      for (int k = 0; k < c; k++) acc = hj_dbl<O>(acc);
    acc = hj_add<O>(acc, hj_from_xyzz<O>(win + (size_t)w * 16));
  }
  const HFe one = h_one<F>();
  if (h_is_zero(acc.z)) {
    memcpy(out, one.v, 32);
    memcpy(out + 4, one.v, 32);
    memset(out + 8, 0, 32);
    return;
  }
  const HFe X = O::to_mont(acc.x), Y = O::to_mont(acc.y), Z = O::to_mont(acc.z);
  const HFe zi = h_inv_kaliski<F>(Z);
  const HFe zi2 = h_sqr<F>(zi);
  const HFe x = h_mul<F>(X, zi2), y = h_mul<F>(Y, h_mul<F>(zi2, zi));
  memcpy(out, x.v, 32);
  memcpy(out + 4, y.v, 32);
  memcpy(out + 8, one.v, 32);
}

}  // namespace host
}  // namespace bp
