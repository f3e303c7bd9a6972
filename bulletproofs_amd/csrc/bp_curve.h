// Short-Weierstrass (a = 0) group arithmetic for gfx950 on top of bp_field.h.
// Coordinates in Montgomery form.  Two projective systems are used:
//   XYZZ   (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; identity ZZ = 0) for bucket
//          accumulation: mixed add 8M+2S, no doubling on the common path;
//   Jacobian (x = X/Z^2, y = Y/Z^3; identity Z = 0) for double-and-add
//          (doubling 2M+5S), and as the C ABI's result form, which is the
//          layout of ark-ec's short_weierstrass::Projective.
// Replaces the ark-ec 0.5.0 group operations behind G::msm_unchecked,
// Affine::mul and G::normalize_batch (reference call sites:
// src/ipa/mod.rs:54,70,206,221-226; src/msm.rs:84-87).
#pragma once
#include "bp_field.h"

namespace bp {

// Fq is the field the kernels compute in, FqIn the form coordinates have at the library's
// boundary (arkworks' Montgomery form).  They differ for Secp256k1P, the curve as the MSM
// pipeline sees it: coordinates are converted to plain residues once, where the points are
// packed, because p = 2^256 - 2^32 - 977 reduces a plain product by folding (72 multiply-
// accumulates per field product instead of 88; see fe_mul_dev_Secp256k1FqP), and the window
// sums are converted back where they leave the pipeline.
template <class Fq_, class Fr_, class FqIn_ = Fq_>
struct CurveT {
  using Fq = Fq_;
  using Fr = Fr_;
  using FqIn = FqIn_;
  static constexpr bool kConverts = !__is_same(Fq_, FqIn_);
};
using Secp256k1P = CurveT<Secp256k1FqP, Secp256k1Fr, Secp256k1Fq>;
// the curves at the boundary; Pipeline = the type the MSM pipeline computes on
struct Secp256k1 : CurveT<Secp256k1Fq, Secp256k1Fr> {
  using Pipeline = Secp256k1P;
};
struct Vesta : CurveT<VestaFq, VestaFr> {
  using Pipeline = Vesta;
};

// boundary form -> the form C's kernels compute in, and back
template <class C>
BP_DEV Fe fe_import(const Fe& a) {
  if constexpr (C::kConverts)
    return fe_from_mont<typename C::FqIn>(a);
  else
    return a;
}
template <class C>
BP_DEV Fe fe_export(const Fe& a) {
  if constexpr (C::kConverts)
    return fe_to_mont<typename C::FqIn>(a);
  else
    return a;
}

struct Xyzz {
  Fe x, y, zz, zzz;
};
struct Jac {
  Fe x, y, z;
};

template <class F>
BP_DEV Xyzz xyzz_identity() {
  Xyzz r;
  r.x = fe_one<F>();
  r.y = fe_one<F>();
  r.zz = fe_zero();
  r.zzz = fe_zero();
  return r;
}

BP_DEV bool xyzz_is_identity(const Xyzz& p) { return fe_is_zero(p.zz); }

template <class F>
BP_DEV Xyzz xyzz_from_affine(const Fe& x, const Fe& y) {
  Xyzz r;
  r.x = x;
  r.y = y;
  r.zz = fe_one<F>();
  r.zzz = fe_one<F>();
  return r;
}

// 2*(x, y) for an affine input (mdbl-2008-s-1, a = 0)
template <class F>
BP_DEV Xyzz xyzz_dbl_affine(const Fe& x, const Fe& y) {
  Xyzz r;
  if (fe_is_zero(y)) return xyzz_identity<F>();
  Fe u = fe_dbl<F>(y);
  Fe v = fe_sqr<F>(u);
  Fe w = fe_mul<F>(u, v);
  Fe s = fe_mul<F>(x, v);
  Fe xx = fe_sqr<F>(x);
  Fe m = fe_add<F>(fe_dbl<F>(xx), xx);
  r.x = fe_sub<F>(fe_sqr<F>(m), fe_dbl<F>(s));
  r.y = fe_sub<F>(fe_mul<F>(m, fe_sub<F>(s, r.x)), fe_mul<F>(w, y));
  r.zz = v;
  r.zzz = w;
  return r;
}

// dbl-2008-s-1, a = 0
template <class F>
BP_DEV Xyzz xyzz_dbl(const Xyzz& p) {
  if (xyzz_is_identity(p) || fe_is_zero(p.y)) return xyzz_identity<F>();
  Xyzz r;
  Fe u = fe_dbl<F>(p.y);
  Fe v = fe_sqr<F>(u);
  Fe w = fe_mul<F>(u, v);
  Fe s = fe_mul<F>(p.x, v);
  Fe xx = fe_sqr<F>(p.x);
  Fe m = fe_add<F>(fe_dbl<F>(xx), xx);
  r.x = fe_sub<F>(fe_sqr<F>(m), fe_dbl<F>(s));
  r.y = fe_sub<F>(fe_mul<F>(m, fe_sub<F>(s, r.x)), fe_mul<F>(w, p.y));
  r.zz = fe_mul<F>(v, p.zz);
  r.zzz = fe_mul<F>(w, p.zzz);
  return r;
}

// p += (x2, y2) affine, not the identity (madd-2008-s)
template <class F>
BP_DEV void xyzz_madd(Xyzz& p, const Fe& x2, const Fe& y2) {
  if (xyzz_is_identity(p)) {
    p = xyzz_from_affine<F>(x2, y2);
    return;
  }
  Fe u2 = fe_mul<F>(x2, p.zz);
  Fe s2 = fe_mul<F>(y2, p.zzz);
  Fe pd = fe_sub<F>(u2, p.x);
  Fe rd = fe_sub<F>(s2, p.y);
  if (fe_is_zero(pd)) {
    if (fe_is_zero(rd))
      p = xyzz_dbl_affine<F>(x2, y2);
    else
      p = xyzz_identity<F>();
    return;
  }
  Fe pp = fe_sqr<F>(pd);
  Fe ppp = fe_mul<F>(pd, pp);
  Fe q = fe_mul<F>(p.x, pp);
  Fe x3 = fe_sub<F>(fe_sub<F>(fe_sqr<F>(rd), ppp), fe_dbl<F>(q));
  Fe y3 = fe_sub<F>(fe_mul<F>(rd, fe_sub<F>(q, x3)), fe_mul<F>(p.y, ppp));
  p.zz = fe_mul<F>(p.zz, pp);
  p.zzz = fe_mul<F>(p.zzz, ppp);
  p.x = x3;
  p.y = y3;
}

// p += q (add-2008-s)
template <class F>
BP_DEV void xyzz_add(Xyzz& p, const Xyzz& q) {
  if (xyzz_is_identity(q)) return;
  if (xyzz_is_identity(p)) {
    p = q;
    return;
  }
  Fe u1 = fe_mul<F>(p.x, q.zz);
  Fe u2 = fe_mul<F>(q.x, p.zz);
  Fe s1 = fe_mul<F>(p.y, q.zzz);
  Fe s2 = fe_mul<F>(q.y, p.zzz);
  Fe pd = fe_sub<F>(u2, u1);
  Fe rd = fe_sub<F>(s2, s1);
  if (fe_is_zero(pd)) {
    if (fe_is_zero(rd))
      p = xyzz_dbl<F>(p);
    else
      p = xyzz_identity<F>();
    return;
  }
  Fe pp = fe_sqr<F>(pd);
  Fe ppp = fe_mul<F>(pd, pp);
  Fe qq = fe_mul<F>(u1, pp);
  Fe x3 = fe_sub<F>(fe_sub<F>(fe_sqr<F>(rd), ppp), fe_dbl<F>(qq));
  Fe y3 = fe_sub<F>(fe_mul<F>(rd, fe_sub<F>(qq, x3)), fe_mul<F>(s1, ppp));
  p.zz = fe_mul<F>(fe_mul<F>(p.zz, q.zz), pp);
  p.zzz = fe_mul<F>(fe_mul<F>(p.zzz, q.zzz), ppp);
  p.x = x3;
  p.y = y3;
}

// XYZZ -> affine.  1/zz = (zz/zzz)^2 because zz^3 = zzz^2.
template <class F>
BP_DEV bool xyzz_to_affine(const Xyzz& p, Fe& x, Fe& y) {
  if (xyzz_is_identity(p)) {
    x = fe_zero();
    y = fe_zero();
    return true;
  }
  Fe i3 = fe_inv<F>(p.zzz);
  Fe t = fe_mul<F>(p.zz, i3);
  Fe i2 = fe_sqr<F>(t);
  x = fe_mul<F>(p.x, i2);
  y = fe_mul<F>(p.y, i3);
  return false;
}

// ---- Jacobian -------------------------------------------------------------

template <class F>
BP_DEV Jac jac_identity() {
  Jac r;
  r.x = fe_one<F>();
  r.y = fe_one<F>();
  r.z = fe_zero();
  return r;
}

BP_DEV bool jac_is_identity(const Jac& p) { return fe_is_zero(p.z); }

// dbl-2009-l (a = 0): 2M + 5S
template <class F>
BP_DEV Jac jac_dbl(const Jac& p) {
  Jac r;
  Fe a = fe_sqr<F>(p.x);
  Fe b = fe_sqr<F>(p.y);
  Fe c = fe_sqr<F>(b);
  Fe xb = fe_add<F>(p.x, b);
  Fe d = fe_dbl<F>(fe_sub<F>(fe_sub<F>(fe_sqr<F>(xb), a), c));
  Fe e = fe_add<F>(fe_dbl<F>(a), a);
  Fe f = fe_sqr<F>(e);
  r.x = fe_sub<F>(f, fe_dbl<F>(d));
  Fe c8 = fe_dbl<F>(fe_dbl<F>(fe_dbl<F>(c)));
  r.z = fe_dbl<F>(fe_mul<F>(p.y, p.z));  // before y is overwritten; z = 0 stays 0
  r.y = fe_sub<F>(fe_mul<F>(e, fe_sub<F>(d, r.x)), c8);
  return r;
}

// p += (x2, y2) affine, not the identity (madd-2007-bl): 7M + 4S
template <class F>
BP_DEV void jac_madd(Jac& p, const Fe& x2, const Fe& y2) {
  if (jac_is_identity(p)) {
    p.x = x2;
    p.y = y2;
    p.z = fe_one<F>();
    return;
  }
  Fe z1z1 = fe_sqr<F>(p.z);
  Fe u2 = fe_mul<F>(x2, z1z1);
  Fe s2 = fe_mul<F>(fe_mul<F>(y2, p.z), z1z1);
  Fe h = fe_sub<F>(u2, p.x);
  Fe rr = fe_sub<F>(s2, p.y);
  if (fe_is_zero(h)) {
    if (fe_is_zero(rr))
      p = jac_dbl<F>(p);
    else
      p = jac_identity<F>();
    return;
  }
  Fe hh = fe_sqr<F>(h);
  Fe i = fe_dbl<F>(fe_dbl<F>(hh));
  Fe j = fe_mul<F>(h, i);
  Fe r = fe_dbl<F>(rr);
  Fe v = fe_mul<F>(p.x, i);
  Fe x3 = fe_sub<F>(fe_sub<F>(fe_sqr<F>(r), j), fe_dbl<F>(v));
  Fe y3 = fe_sub<F>(fe_mul<F>(r, fe_sub<F>(v, x3)), fe_dbl<F>(fe_mul<F>(p.y, j)));
  Fe zh = fe_add<F>(p.z, h);
  p.z = fe_sub<F>(fe_sub<F>(fe_sqr<F>(zh), z1z1), hh);
  p.x = x3;
  p.y = y3;
}

// p += q (add-2007-bl): 11M + 5S
template <class F>
BP_DEV void jac_add(Jac& p, const Jac& q) {
  if (jac_is_identity(q)) return;
  if (jac_is_identity(p)) {
    p = q;
    return;
  }
  Fe z1z1 = fe_sqr<F>(p.z);
  Fe z2z2 = fe_sqr<F>(q.z);
  Fe u1 = fe_mul<F>(p.x, z2z2);
  Fe u2 = fe_mul<F>(q.x, z1z1);
  Fe s1 = fe_mul<F>(fe_mul<F>(p.y, q.z), z2z2);
  Fe s2 = fe_mul<F>(fe_mul<F>(q.y, p.z), z1z1);
  Fe h = fe_sub<F>(u2, u1);
  Fe rr = fe_sub<F>(s2, s1);
  if (fe_is_zero(h)) {
    if (fe_is_zero(rr))
      p = jac_dbl<F>(p);
    else
      p = jac_identity<F>();
    return;
  }
  Fe i = fe_sqr<F>(fe_dbl<F>(h));
  Fe j = fe_mul<F>(h, i);
  Fe r = fe_dbl<F>(rr);
  Fe v = fe_mul<F>(u1, i);
  Fe x3 = fe_sub<F>(fe_sub<F>(fe_sqr<F>(r), j), fe_dbl<F>(v));
  Fe y3 = fe_sub<F>(fe_mul<F>(r, fe_sub<F>(v, x3)), fe_dbl<F>(fe_mul<F>(s1, j)));
  Fe zs = fe_add<F>(p.z, q.z);
  p.z = fe_mul<F>(fe_sub<F>(fe_sub<F>(fe_sqr<F>(zs), z1z1), z2z2), h);
  p.x = x3;
  p.y = y3;
}

template <class F>
BP_DEV bool jac_to_affine(const Jac& p, Fe& x, Fe& y) {
  if (jac_is_identity(p)) {
    x = fe_zero();
    y = fe_zero();
    return true;
  }
  Fe zi = fe_inv<F>(p.z);
  Fe zi2 = fe_sqr<F>(zi);
  x = fe_mul<F>(p.x, zi2);
  y = fe_mul<F>(fe_mul<F>(p.y, zi2), zi);
  return false;
}

}  // namespace bp
