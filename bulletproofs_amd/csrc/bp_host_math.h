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
template <class F>
inline HFe h_add(const HFe& a, const HFe& b) {
  HFe r;
  u128 c = 0;
  for (int i = 0; i < 4; i++) {
    c += (u128)a.v[i] + b.v[i];
    r.v[i] = (uint64_t)c;
    c >>= 64;
  }
  if (c || h_geq_mod<F>(r)) h_sub_mod<F>(r);
  return r;
}
template <class F>
inline HFe h_sub(const HFe& a, const HFe& b) {
  HFe r;
  u128 br = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)a.v[i] - b.v[i] - (uint64_t)br;
    r.v[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
  if (br) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
      c += (u128)r.v[i] + F::mod64(i);
      r.v[i] = (uint64_t)c;
      c >>= 64;
    }
  }
  return r;
}
template <class F>
inline HFe h_dbl(const HFe& a) {
  return h_add<F>(a, a);
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

}  // namespace host
}  // namespace bp
