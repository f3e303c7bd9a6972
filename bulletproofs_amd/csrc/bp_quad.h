// Point addition and doubling spread over four adjacent lanes (a quad).
//
// A chain of dependent point additions on a few lanes is bound by the time of ONE addition.  An
// XYZZ addition is 14 field multiplications, but only four of them are sequential: {U1, U2, S1,
// S2}, then {P^2, R^2, ZZ1*ZZ2, ZZZ1*ZZZ2}, then {P^3, U1*P^2, ZZ1*ZZ2*P^2}, then {R*(Q - X3),
// S1*P^3, ZZZ1*ZZZ2*P^3}.  The four lanes of a quad take one multiplication of each stage and
// exchange the products through a private piece of LDS (all four sit in one wave, so a wait on the
// LDS counter orders the exchange; no block barrier).  Which operand a lane reads is a matter of
// its address, not of a branch, so the four lanes run one instruction stream.  Doubling is three
// stages the same way.
#pragma once
#include "bp_curve.h"

namespace bp {

constexpr int kQuadScratch = 8;  // field elements of exchange space per quad

// orders a quad's LDS writes before the reads that follow (same wave: LDS operations
// complete in order) and keeps the compiler from moving either across
BP_DEV void quad_sync() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}

// coordinate (0 x, 1 y, 2 zz, 3 zzz) that the lane with this role holds when an operation ends
BP_DEV int quad_coord(int role) { return role == 0 ? 1 : role == 1 ? 0 : role == 2 ? 3 : 2; }

BP_DEV Fe quad_identity_coord(int coord, const Fe& one) { return coord < 2 ? one : fe_zero(); }

// pts[dst] + pts[src] (XYZZ, 32 words each), by the four lanes of a quad.  Nothing is written
// to pts: on return the lane holds coordinate quad_coord(role) of the sum in `out`, and the
// result says whether dst changes at all (false: src is the identity).  scr: the quad's
// exchange space; zero: eight zero words in LDS.
template <class F>
BP_DEV bool quad_add(const u32* pts, u32* scr, const u32* zero, u32 dst, u32 src, int role, Fe& out) {
  const u32* p = pts + dst * 32;
  const u32* q = pts + src * 32;
  const int coord = quad_coord(role);
  const Fe one = fe_one<F>();
  if (fe_is_zero(fe_load(q + 16))) return false;
  if (fe_is_zero(fe_load(p + 16))) {
    out = fe_load(q + coord * 8);
    return true;
  }
  const int hi = role >> 1, odd = role & 1;
  // stage 1: U1 = X1*ZZ2, U2 = X2*ZZ1, S1 = Y1*ZZZ2, S2 = Y2*ZZZ1
  {
    Fe a = fe_load((odd ? q : p) + hi * 8);
    Fe b = fe_load((odd ? p : q) + 16 + hi * 8);
    fe_store(scr + role * 8, fe_mul<F>(a, b));
  }
  quad_sync();
  // stage 2: P^2 with P = U2 - U1, R^2 with R = S2 - S1, ZZ1*ZZ2, ZZZ1*ZZZ2
  Fe pr;  // P on role 0, R on role 1
  {
    const u32* a1 = role < 2 ? scr + (2 * role + 1) * 8 : p + 16 + odd * 8;
    const u32* a2 = role < 2 ? scr + (2 * role) * 8 : zero;
    const u32* b1 = role < 2 ? a1 : q + 16 + odd * 8;
    pr = fe_sub<F>(fe_load(a1), fe_load(a2));
    Fe b = fe_select(role < 2, pr, fe_load(b1));
    const int flag = fe_is_zero(pr) ? 1 : 0;
    const int lane0 = (int)(threadIdx.x & 63u) & ~3;
    const int p_zero = __shfl(flag, lane0, 64), r_zero = __shfl(flag, lane0 + 1, 64);
    if (p_zero) {
      // same x: the same point twice (doubled by every lane on its own) or a point and its negative
      if (r_zero) {
        Xyzz v;
        v.x = fe_load(p);
        v.y = fe_load(p + 8);
        v.zz = fe_load(p + 16);
        v.zzz = fe_load(p + 24);
        v = xyzz_dbl<F>(v);
        out = coord == 0 ? v.x : coord == 1 ? v.y : coord == 2 ? v.zz : v.zzz;
      } else {
        out = quad_identity_coord(coord, one);
      }
      return true;
    }
    fe_store(scr + (4 + role) * 8, fe_mul<F>(pr, b));
  }
  quad_sync();
  // stage 3: P^3 = P*P^2, Q = U1*P^2, ZZ3 = ZZ1*ZZ2*P^2 (role 3 has nothing to do and repeats Q).
  // The products replace values that are dead once this stage has read its operands (the quad
  // runs one instruction stream, so every lane's loads come before any lane's store):
  // P^3 -> U2's place, Q -> U1's, ZZ3 -> ZZ1*ZZ2's, role 3's -> P^2's.
  {
    const u32* a1 = role == 2 ? scr + 6 * 8 : scr;
    Fe a = fe_select(role == 0, pr, fe_load(a1));
    Fe pp = fe_load(scr + 4 * 8);
    u32* d = role == 0 ? scr + 1 * 8 : role == 1 ? scr : role == 2 ? scr + 6 * 8 : scr + 4 * 8;
    fe_store(d, fe_mul<F>(a, pp));
  }
  quad_sync();
  // stage 4: X3 = R^2 - P^3 - 2Q on every lane; R*(Q - X3), S1*P^3, ZZZ3 = ZZZ1*ZZZ2*P^3
  Fe x3, m;
  {
    Fe ppp = fe_load(scr + 1 * 8), qq = fe_load(scr), rr = fe_load(scr + 5 * 8);
    x3 = fe_sub<F>(fe_sub<F>(rr, ppp), fe_dbl<F>(qq));
    const u32* a1 = role == 0 ? scr + 3 * 8 : role == 2 ? scr + 7 * 8 : scr + 2 * 8;
    const u32* a2 = role == 0 ? scr + 2 * 8 : zero;
    Fe a = fe_sub<F>(fe_load(a1), fe_load(a2));
    Fe b = fe_select(role == 0, fe_sub<F>(qq, x3), ppp);
    m = fe_mul<F>(a, b);
    if (role == 1) fe_store(scr + 4 * 8, m);
  }
  quad_sync();
  {
    Fe t = fe_load(role == 3 ? scr + 6 * 8 : scr + 4 * 8);  // ZZ3 for role 3, S1*P^3 otherwise
    Fe y3 = fe_sub<F>(m, t);
    out = role == 0 ? y3 : role == 1 ? x3 : role == 2 ? m : t;
  }
  return true;
}

// pt (32 words in LDS, owned by this quad) doubled in place (dbl-2008-s-1, a = 0)
template <class F>
BP_DEV void quad_dbl(u32* pt, u32* scr, const u32* zero, int role) {
  const int coord = quad_coord(role);
  const Fe one = fe_one<F>();
  if (fe_is_zero(fe_load(pt + 16)) || fe_is_zero(fe_load(pt + 8))) {
    quad_sync();
    fe_store(pt + coord * 8, quad_identity_coord(coord, one));
    quad_sync();
    return;
  }
  // stage 1: V = U^2 with U = 2Y (role 0), XX = X^2 (role 1); roles 2 and 3 repeat them
  Fe u = fe_dbl<F>(fe_load(pt + 8));
  {
    Fe a = fe_select((role & 1) == 0, u, fe_load(pt));
    fe_store(scr + role * 8, fe_mul<F>(a, a));
  }
  quad_sync();
  // stage 2: W = U*V, S = X*V, M^2 with M = 3 XX, ZZ3 = V*ZZ
  Fe mm;  // M
  {
    Fe v = fe_load(scr), xx = fe_load(scr + 8);
    mm = fe_add<F>(fe_dbl<F>(xx), xx);
    Fe a = fe_load(role == 1 ? pt : role == 3 ? pt + 16 : zero);
    a = fe_select(role == 0, u, fe_select(role == 2, mm, a));
    Fe b = fe_select(role == 2, mm, v);
    fe_store(scr + (4 + role) * 8, fe_mul<F>(a, b));
  }
  quad_sync();
  // stage 3: X3 = M^2 - 2S on every lane; M*(S - X3), W*Y, ZZZ3 = W*ZZZ
  Fe x3, m;
  {
    Fe w = fe_load(scr + 4 * 8), s = fe_load(scr + 5 * 8), m2 = fe_load(scr + 6 * 8);
    x3 = fe_sub<F>(m2, fe_dbl<F>(s));
    Fe a = fe_select(role == 0, mm, w);
    Fe b = fe_load(role == 2 ? pt + 24 : pt + 8);
    b = fe_select(role == 0, fe_sub<F>(s, x3), b);
    m = fe_mul<F>(a, b);
    if (role == 1) fe_store(scr, m);
  }
  quad_sync();
  {
    Fe t = fe_load(role == 3 ? scr + 7 * 8 : scr);  // ZZ3 for role 3, W*Y otherwise
    Fe y3 = fe_sub<F>(m, t);
    Fe out = role == 0 ? y3 : role == 1 ? x3 : role == 2 ? m : t;
    fe_store(pt + coord * 8, out);
  }
  quad_sync();
}

}  // namespace bp
