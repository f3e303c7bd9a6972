// Host-side C++ mirror of the reference's types around the hot path, on top of the
// C ABI (include/bp_gpu.h): the deferred-MSM accumulator `Msm` (src/msm.rs:8-115) and
// the inner-product-argument prover / verifier loops (src/ipa/mod.rs:19-212) with the
// Fiat-Shamir transcript supplied by the caller (spongefish is out of scope).
// Heavy work -- every MSM, fold, dot product, s-vector -- goes through libbp_gpu; the
// host keeps the bookkeeping the reference also keeps on the host (hash map merges,
// a handful of single scalar multiplications per round).
#pragma once
#include <stdint.h>
#include <string.h>

#include <stdio.h>
#include <stdlib.h>

#include <chrono>
#include <stdexcept>
#include <unordered_map>
#include <vector>

#include "../../include/bp_gpu.h"
#include "../csrc/bp_host_math.h"

namespace bp {
namespace host {

struct Scalar {
  uint64_t l[4];
  bool operator==(const Scalar& o) const { return memcmp(l, o.l, 32) == 0; }
};
struct Affine {
  uint64_t x[4], y[4];
  uint8_t inf;
  bool operator==(const Affine& o) const {
    if (inf || o.inf) return inf && o.inf;
    return memcmp(x, o.x, 32) == 0 && memcmp(y, o.y, 32) == 0;
  }
};
struct AffineHash {
  size_t operator()(const Affine& p) const {
    if (p.inf) return 0x9E3779B97F4A7C15ull;
    uint64_t h = 0xcbf29ce484222325ull;
    for (int i = 0; i < 4; i++) h = (h ^ p.x[i]) * 0x100000001b3ull, h = (h ^ p.y[i]) * 0x100000001b3ull;
    return (size_t)h;
  }
};
struct Projective {  // normalised Jacobian as returned by the ABI
  uint64_t xyz[12];
};

// BP_HOST_TRACE=1 prints the wall time between marks of a driver to stderr
struct PhaseTrace {
  bool on = getenv("BP_HOST_TRACE") != nullptr;
  std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
  void mark(const char* what) {
    if (!on) return;
    auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[bp trace] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count());
    last = now;
  }
};

// What the reference's verifiers return as Err(VerificationError) (src/lib.rs:9-10): a proof whose
// shape does not fit the statement.  Drivers treat it as a rejection.
struct VerificationError : std::runtime_error {
  explicit VerificationError(const char* what) : std::runtime_error(what) {}
};

// log2 of the statement size must equal the number of (L, R) pairs the proof carries: the reference
// reads exactly log2(n) pairs from the transcript (src/ipa/mod.rs:110-119), so a proof with more or
// fewer rounds fails to deserialise there.  Checked before any allocation or device call.
inline size_t checked_rounds(size_t n_l, size_t n_r, size_t n) {
  if (n_l != n_r) throw VerificationError("proof with unequal numbers of L and R");
  if (n_l >= 64 || (size_t(1) << n_l) != n) throw VerificationError("proof of another size than the statement");
  return n_l;
}

inline void check(bp_ctx* ctx, int rc, const char* what) {
  if (rc != BP_OK) throw std::runtime_error(std::string(what) + ": " + bp_last_error(ctx));
}

// ---- scalar-field helpers (Montgomery form in and out) --------------------------
template <class Fr>
struct ScalarOps {
  static HFe to(const Scalar& s) { HFe r; memcpy(r.v, s.l, 32); return r; }
  static Scalar from(const HFe& f) { Scalar s; memcpy(s.l, f.v, 32); return s; }
  static Scalar add(const Scalar& a, const Scalar& b) { return from(h_add<Fr>(to(a), to(b))); }
  static Scalar mul(const Scalar& a, const Scalar& b) { return from(h_mul<Fr>(to(a), to(b))); }
  static Scalar neg(const Scalar& a) { return from(h_neg<Fr>(to(a))); }
  // Fermat inverse with Montgomery products: Montgomery form in, Montgomery form out
  static Scalar inv(const Scalar& a) { return from(h_inv_kaliski<Fr>(to(a))); }  // public values only: variable time
  static bool is_zero(const Scalar& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
  // ark_ff::batch_inversion (src/ipa/mod.rs:128-132): one inversion and 3 multiplications per
  // element; zeros stay zero
  static void batch_inv(const std::vector<Scalar>& in, std::vector<Scalar>& out) {
    const size_t n = in.size();
    out.assign(n, zero());
    std::vector<Scalar> pre(n);
    Scalar run = one();
    for (size_t i = 0; i < n; i++) {
      pre[i] = run;
      if (!is_zero(in[i])) run = mul(run, in[i]);
    }
    Scalar acc = inv(run);
    for (size_t i = n; i-- > 0;) {
      if (is_zero(in[i])) continue;
      out[i] = mul(acc, pre[i]);
      acc = mul(acc, in[i]);
    }
  }
  static Scalar to_mont(const Scalar& canonical) {
    HFe r2{{Fr::r264(0), Fr::r264(1), Fr::r264(2), Fr::r264(3)}};
    return from(h_mul<Fr>(to(canonical), r2));
  }
  static Scalar one() { return from(h_one<Fr>()); }
  static Scalar zero() { return Scalar{{0, 0, 0, 0}}; }
};

// ---- single-point helpers on the host (a few per IPA round) ----------------------
template <class Fq, class Fr>
struct PointOps {
  static HXyzz lift(const Affine& p) {
    if (p.inf) return hx_identity<Fq>();
    HXyzz r;
    memcpy(r.x.v, p.x, 32); memcpy(r.y.v, p.y, 32);
    r.zz = h_one<Fq>(); r.zzz = h_one<Fq>();
    return r;
  }
  static HXyzz lift(const Projective& p) {
    HFe z; memcpy(z.v, p.xyz + 8, 32);
    if (h_is_zero(z)) return hx_identity<Fq>();
    Affine a; memcpy(a.x, p.xyz, 32); memcpy(a.y, p.xyz + 4, 32); a.inf = 0;  // Z == 1 by contract
    return lift(a);
  }
  static HXyzz mul(const HXyzz& p, const Scalar& k_mont) {
    HFe km; memcpy(km.v, k_mont.l, 32);
    HFe k = h_from_mont<Fr>(km);
    HXyzz acc = hx_identity<Fq>();
    for (int i = 255; i >= 0; i--) {
      acc = hx_dbl<Fq>(acc);
      if ((k.v[i >> 6] >> (i & 63)) & 1) acc = hx_add<Fq>(acc, p);
    }
    return acc;
  }
  static Projective lower(const HXyzz& p) { Projective r; hx_to_xyz<Fq>(p, r.xyz); return r; }
  static Affine to_affine(const Projective& p) {
    Affine a; HFe z; memcpy(z.v, p.xyz + 8, 32);
    a.inf = h_is_zero(z) ? 1 : 0;
    if (a.inf) { memset(a.x, 0, 32); memset(a.y, 0, 32); } else { memcpy(a.x, p.xyz, 32); memcpy(a.y, p.xyz + 4, 32); }
    return a;
  }
};

// ---- fixed-base multiplication for the CRS points used one at a time (g, h, u) ------------
// table[w][d-1] = d * 2^(8w) * P in affine coordinates (w < 32, d = 1..255, 522 KB): a
// scalar multiplication is then at most 32 mixed additions instead of 256 doublings and
// ~128 additions.  Built once per CRS.
template <class Fq, class Fr>
struct FixedBase {
  std::vector<HFe> tx, ty;
  bool ready = false;

  static void madd(HXyzz& p, const HFe& x2, const HFe& y2) {  // madd-2008-s, a = 0
    if (hx_is_identity(p)) { p.x = x2; p.y = y2; p.zz = h_one<Fq>(); p.zzz = h_one<Fq>(); return; }
    HFe u2 = h_mul<Fq>(x2, p.zz), s2 = h_mul<Fq>(y2, p.zzz);
    HFe pd = h_sub<Fq>(u2, p.x), rd = h_sub<Fq>(s2, p.y);
    if (h_is_zero(pd)) {
      HXyzz q; q.x = x2; q.y = y2; q.zz = h_one<Fq>(); q.zzz = h_one<Fq>();
      p = h_is_zero(rd) ? hx_dbl<Fq>(q) : hx_identity<Fq>();
      return;
    }
    HFe pp = h_sqr<Fq>(pd), ppp = h_mul<Fq>(pd, pp), q = h_mul<Fq>(p.x, pp);
    HFe x3 = h_sub<Fq>(h_sub<Fq>(h_sqr<Fq>(rd), ppp), h_dbl<Fq>(q));
    HFe y3 = h_sub<Fq>(h_mul<Fq>(rd, h_sub<Fq>(q, x3)), h_mul<Fq>(p.y, ppp));
    p.zz = h_mul<Fq>(p.zz, pp); p.zzz = h_mul<Fq>(p.zzz, ppp); p.x = x3; p.y = y3;
  }

  void build(const Affine& pt) {
    ready = false;
    if (pt.inf) return;
    const size_t total = 32 * 255;
    std::vector<HXyzz> pts(total);
    HXyzz base = PointOps<Fq, Fr>::lift(pt);
    for (int w = 0; w < 32; w++) {
      HXyzz cur = base;
      for (int d = 1; d <= 255; d++) {
        pts[(size_t)w * 255 + d - 1] = cur;
        cur = hx_add<Fq>(cur, base);
      }
      base = cur;  // 256 * base
    }
    // batch inversion of the zzz coordinates (none is zero: the order is prime and > 2^254)
    std::vector<HFe> pre(total);
    HFe run = h_one<Fq>();
    for (size_t i = 0; i < total; i++) { pre[i] = run; run = h_mul<Fq>(run, pts[i].zzz); }
    HFe inv = h_inv<Fq>(run);
    tx.resize(total); ty.resize(total);
    for (size_t i = total; i-- > 0;) {
      HFe i3 = h_mul<Fq>(inv, pre[i]);
      inv = h_mul<Fq>(inv, pts[i].zzz);
      HFe t = h_mul<Fq>(pts[i].zz, i3);  // 1/zz = (zz/zzz)^2
      tx[i] = h_mul<Fq>(pts[i].x, h_sqr<Fq>(t));
      ty[i] = h_mul<Fq>(pts[i].y, i3);
    }
    ready = true;
  }

  // k * P (k in Montgomery form); the identity when the base is the identity
  HXyzz mul(const Scalar& k_mont) const {
    HXyzz acc = hx_identity<Fq>();
    if (!ready) return acc;
    HFe km; memcpy(km.v, k_mont.l, 32);
    HFe k = h_from_mont<Fr>(km);
    for (int w = 0; w < 32; w++) {
      unsigned d = (unsigned)((k.v[w >> 3] >> ((w & 7) * 8)) & 0xff);
      if (d) madd(acc, tx[(size_t)w * 255 + d - 1], ty[(size_t)w * 255 + d - 1]);
    }
    return acc;
  }
};

// ---- Msm: src/msm.rs:8-88 ----------------------------------------------------------
template <class Fq, class Fr, int CURVE>
class Msm {
 public:
  using S = ScalarOps<Fr>;
  explicit Msm(bp_ctx* ctx) : ctx_(ctx) {}
  // An accumulator whose bases are mostly points of a registered base set (a CRS): those are
  // addressed by their index in the set and keep their scalars in a dense vector, so that
  // neither a hash lookup per generator nor an upload of the generators is needed; any other
  // point goes through the map as before.  The group element of execute() is that of the
  // plain accumulator holding the same (base, scalar) pairs.
  Msm(bp_ctx* ctx, const bp_bases* bases, size_t n_bases) : ctx_(ctx), bases_(bases), dense_(n_bases, S::zero()) {}
  void upsert_resident(size_t index, const Scalar& scalar) { dense_.at(index) = S::add(dense_.at(index), scalar); }
  void upsert_resident_range(size_t offset, const std::vector<Scalar>& scalars) {
    if (offset + scalars.size() > dense_.size()) throw std::out_of_range("resident range");
    for (size_t i = 0; i < scalars.size(); i++) dense_[offset + i] = S::add(dense_[offset + i], scalars[i]);
  }
  void scale_resident(size_t index, const Scalar& factor) { dense_.at(index) = S::mul(dense_.at(index), factor); }
  // src/msm.rs:33-38
  void upsert(const Affine& base, const Scalar& scalar) {
    auto it = map_.find(base);
    if (it == map_.end()) map_.emplace(base, scalar); else it->second = S::add(it->second, scalar);
  }
  // src/msm.rs:41-48
  void upsert_batch(const std::vector<Affine>& bases, const std::vector<Scalar>& scalars) {
    size_t n = bases.size() < scalars.size() ? bases.size() : scalars.size();
    for (size_t i = 0; i < n; i++) upsert(bases[i], scalars[i]);
  }
  // src/msm.rs:50-54
  void scale(const Scalar& s) {
    for (auto& kv : map_) kv.second = S::mul(kv.second, s);
    for (auto& v : dense_) v = S::mul(v, s);
  }
  // src/msm.rs:62-71
  void scale_elems(const std::vector<Affine>& bases, const std::vector<Scalar>& factors) {
    for (size_t i = 0; i < bases.size() && i < factors.size(); i++) {
      auto it = map_.find(bases[i]);
      if (it != map_.end()) it->second = S::mul(it->second, factors[i]);
    }
  }
  // src/msm.rs:73-77
  void batch(const Msm& rhs) {
    for (auto& kv : rhs.map_) upsert(kv.first, kv.second);
    if (!rhs.dense_.empty()) {
      if (bases_ != rhs.bases_ || dense_.size() != rhs.dense_.size()) throw std::invalid_argument("accumulators over different base sets");
      for (size_t i = 0; i < dense_.size(); i++) dense_[i] = S::add(dense_[i], rhs.dense_[i]);
    }
  }
  size_t size() const { return map_.size(); }
  // src/msm.rs:79-87: one MSM over the de-duplicated map (iteration order is unspecified
  // there too; the group element does not depend on it)
  Projective execute() const {
    size_t n = map_.size();
    std::vector<uint64_t> x(n * 4), y(n * 4), s(n * 4);
    std::vector<uint8_t> inf(n);
    size_t i = 0;
    for (auto& kv : map_) {
      memcpy(&x[4 * i], kv.first.x, 32); memcpy(&y[4 * i], kv.first.y, 32); inf[i] = kv.first.inf;
      memcpy(&s[4 * i], kv.second.l, 32);
      i++;
    }
    Projective out;
    if (bases_)
      check(ctx_, bp_msm_resident_plus(ctx_, bases_, 0, dense_.size(), dense_.empty() ? nullptr : dense_[0].l, x.data(),
                                       y.data(), inf.data(), s.data(), n, out.xyz), "bp_msm_resident_plus");
    else
      check(ctx_, bp_msm(ctx_, CURVE, x.data(), y.data(), inf.data(), s.data(), n, out.xyz), "bp_msm");
    return out;
  }
  bp_ctx* ctx() const { return ctx_; }
  const bp_bases* bases() const { return bases_; }
  const std::vector<Scalar>& dense() const { return dense_; }
  const std::unordered_map<Affine, Scalar, AffineHash>& map() const { return map_; }
  Msm empty_like() const { return bases_ ? Msm(ctx_, bases_, dense_.size()) : Msm(ctx_); }

 private:
  bp_ctx* ctx_;
  const bp_bases* bases_ = nullptr;
  std::vector<Scalar> dense_;  // scalar of bases_[i]
  std::unordered_map<Affine, Scalar, AffineHash> map_;
};

inline bool is_zero(const Projective& p) { return (p.xyz[8] | p.xyz[9] | p.xyz[10] | p.xyz[11]) == 0; }

// src/msm.rs:91-115 with the random alpha supplied by the caller
template <class M>
bool verify_batch_aux(std::vector<M>& proofs, const Scalar& alpha) {
  using S = typename M::S;
  if (proofs.empty()) throw std::invalid_argument("non-empty vec");
  Scalar pw = S::one();
  M acc = proofs[0].empty_like();
  for (size_t k = 0; k < proofs.size(); k++) {
    proofs[k].scale(pw);
    acc.batch(proofs[k]);
    pw = S::mul(pw, alpha);
  }
  return is_zero(acc.execute());
}

// The same check with the combination kept on the device (bp_batch_*): the accumulators must share
// one registered base set; their dense scalars are added there with the weights alpha^k, the few
// bases outside the set are merged in a plain accumulator on the host, and one MSM finishes.  The
// accumulators are left unscaled.
template <class M>
bool verify_batch_aux_resident(const std::vector<M>& proofs, const Scalar& alpha) {
  using S = typename M::S;
  if (proofs.empty()) throw std::invalid_argument("non-empty vec");
  bp_ctx* ctx = proofs[0].ctx();
  const bp_bases* bases = proofs[0].bases();
  const size_t n = proofs[0].dense().size();
  if (!bases) throw std::invalid_argument("accumulators are not over a registered base set");
  bp_batch* batch = nullptr;
  check(ctx, bp_batch_begin(ctx, bases, n, &batch), "bp_batch_begin");
  struct Release { bp_batch* b; ~Release() { bp_batch_free(b); } } release{batch};
  M rest(ctx);
  Scalar pw = S::one();
  for (const M& p : proofs) {
    if (p.bases() != bases || p.dense().size() != n) throw std::invalid_argument("accumulators over different base sets");
    if (n) check(ctx, bp_batch_axpy(batch, 0, p.dense()[0].l, n, pw.l), "bp_batch_axpy");
    for (auto& kv : p.map()) rest.upsert(kv.first, S::mul(kv.second, pw));
    pw = S::mul(pw, alpha);
  }
  const size_t ne = rest.map().size();
  std::vector<uint64_t> x(ne * 4 + 4), y(ne * 4 + 4), sc(ne * 4 + 4);
  std::vector<uint8_t> inf(ne + 1);
  size_t i = 0;
  for (auto& kv : rest.map()) {
    memcpy(&x[4 * i], kv.first.x, 32); memcpy(&y[4 * i], kv.first.y, 32); inf[i] = kv.first.inf;
    memcpy(&sc[4 * i], kv.second.l, 32);
    i++;
  }
  Projective out;
  check(ctx, bp_batch_msm(batch, x.data(), y.data(), inf.data(), sc.data(), ne, out.xyz), "bp_batch_msm");
  return is_zero(out);
}

// ---- transcript stand-in -------------------------------------------------------------
struct Transcript {
  virtual void prover_points(const Projective* pts, size_t n) = 0;  // prover_message(&[L, R])
  virtual void prover_scalars(const Scalar* s, size_t n) = 0;
  virtual Scalar verifier_scalar() = 0;                              // verifier_message()
  virtual ~Transcript() {}
};

struct IpaProof {
  std::vector<Projective> L, R;
  Scalar a, b;
};

template <class Fq, class Fr, int CURVE>
struct Ipa {
  using S = ScalarOps<Fr>;
  using P = PointOps<Fq, Fr>;
  struct Crs { std::vector<Affine> gs, hs; Affine u; };

  static void planes(const std::vector<Affine>& v, size_t n, std::vector<uint64_t>& x, std::vector<uint64_t>& y,
                     std::vector<uint8_t>& inf) {
    x.resize(n * 4); y.resize(n * 4); inf.resize(n);
    for (size_t i = 0; i < n; i++) { memcpy(&x[4 * i], v[i].x, 32); memcpy(&y[4 * i], v[i].y, 32); inf[i] = v[i].inf; }
  }

  // Witness::statement, src/ipa/types.rs:80-88: P = <a,G> + <b,H> + <a,b> u
  static Projective statement(bp_ctx* ctx, const Crs& crs, const std::vector<Scalar>& a, const std::vector<Scalar>& b) {
    size_t n = a.size();
    std::vector<uint64_t> gx, gy, hx, hy; std::vector<uint8_t> gi, hi;
    planes(crs.gs, n, gx, gy, gi); planes(crs.hs, n, hx, hy, hi);
    Projective g, h; Scalar c;
    check(ctx, bp_msm(ctx, CURVE, gx.data(), gy.data(), gi.data(), a[0].l, n, g.xyz), "bp_msm");
    check(ctx, bp_msm(ctx, CURVE, hx.data(), hy.data(), hi.data(), b[0].l, n, h.xyz), "bp_msm");
    check(ctx, bp_inner_product(ctx, CURVE, a[0].l, b[0].l, n, c.l), "bp_inner_product");
    HXyzz p = hx_add<Fq>(hx_add<Fq>(P::lift(g), P::lift(h)), P::mul(P::lift(crs.u), c));
    return P::lower(p);
  }

  // ipa::prove, src/ipa/mod.rs:19-99
  static IpaProof prove(bp_ctx* ctx, const Crs& crs, const std::vector<Scalar>& a, const std::vector<Scalar>& b,
                        Transcript& t) {
    size_t n = a.size();
    std::vector<uint64_t> gx, gy, hx, hy; std::vector<uint8_t> gi, hi;
    planes(crs.gs, n, gx, gy, gi); planes(crs.hs, n, hx, hy, hi);
    bp_ipa* s = nullptr;
    check(ctx, bp_ipa_begin(ctx, CURVE, BP_IPA_BULLETPROOF, gx.data(), gy.data(), gi.data(), hx.data(), hy.data(),
                            hi.data(), a[0].l, b[0].l, n, &s), "bp_ipa_begin");
    return prove_session(ctx, s, crs.u, t);
  }

  // The same loop over generators that are already resident on the GPU (bases[g_off..),
  // bases[h_off..)), optionally with H_i scaled by h_scale[i] (the hs' of
  // src/range/aggregate.rs:148-153 without materialising them).
  static IpaProof prove_resident(bp_ctx* ctx, const bp_bases* bases, size_t g_off, size_t h_off, const Scalar* h_scale,
                                 size_t u_index, const Scalar& u_factor, const std::vector<Scalar>& a,
                                 const std::vector<Scalar>& b, Transcript& t) {
    // u = u_factor * bases[u_index]
    bp_ipa* s = nullptr;
    check(ctx, bp_ipa_begin_resident(ctx, bases, BP_IPA_BULLETPROOF, g_off, h_off, a[0].l, b[0].l, a.size(), &s),
          "bp_ipa_begin_resident");
    int rc = h_scale ? bp_ipa_scale_h(s, h_scale[0].l) : BP_OK;
    if (rc == BP_OK) rc = bp_ipa_set_u_resident(s, u_index, u_factor.l);
    if (rc != BP_OK) { bp_ipa_free(s); check(ctx, rc, "bp_ipa_scale_h / bp_ipa_set_u_resident"); }
    Affine none{};
    none.inf = 1;  // u is already with the session
    return prove_session(ctx, s, none, t, true);
  }

  // takes ownership of the session
  static IpaProof prove_session(bp_ctx* ctx, bp_ipa* s, const Affine& u_aff, Transcript& t, bool u_is_set = false) {
    IpaProof proof;
    HXyzz u = P::lift(u_aff);
    try {
      // u*c_l and u*c_r ride in the device pass (two 256-bit scalar multiplications per
      // round would otherwise be done here on one host core)
      const bool u_on_device = u_is_set || !u_aff.inf;
      if (u_on_device && !u_is_set) {
        uint64_t uxy[8];
        memcpy(uxy, u_aff.x, 32); memcpy(uxy + 4, u_aff.y, 32);
        check(ctx, bp_ipa_set_u(s, uxy), "bp_ipa_set_u");
      }
      while (bp_ipa_len(s) != 1) {
        Projective lm, rm; Scalar cl, cr;
        check(ctx, bp_ipa_round_lr(s, lm.xyz, rm.xyz, cl.l, cr.l), "bp_ipa_round_lr");
        Projective lr[2] = {lm, rm};                                            // :43-55, :59-71
        if (!u_on_device) {
          lr[0] = P::lower(hx_add<Fq>(P::mul(u, cl), P::lift(lm)));
          lr[1] = P::lower(hx_add<Fq>(P::mul(u, cr), P::lift(rm)));
        }
        t.prover_points(lr, 2);                                                 // :75
        Scalar alpha = t.verifier_scalar();                                     // :77
        Scalar alpha_inv = S::inv(alpha);                                       // :78
        check(ctx, bp_ipa_round_fold(s, alpha.l, alpha_inv.l), "bp_ipa_round_fold");  // :80-92
        proof.L.push_back(lr[0]); proof.R.push_back(lr[1]);
      }
      check(ctx, bp_ipa_read(s, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, proof.a.l, proof.b.l), "bp_ipa_read");
    } catch (...) { bp_ipa_free(s); throw; }
    bp_ipa_free(s);
    Scalar ab[2] = {proof.a, proof.b};
    t.prover_scalars(ab, 2);                                                    // :97
    return proof;
  }

  // ipa::verify_aux over a CRS that is registered on the GPU as gs = bases[g_off..g_off+n),
  // hs = bases[h_off..h_off+n), u = bases[u_index] (n_bases points in all): the same accumulator
  // content as verify_aux below, with the CRS scalars addressed by index, so the verifier neither
  // hashes nor uploads 2n generators.
  static Msm<Fq, Fr, CURVE> verify_aux_resident(bp_ctx* ctx, const bp_bases* bases, size_t n_bases, size_t g_off,
                                                size_t h_off, size_t u_index, const Projective& p, size_t n,
                                                const IpaProof& proof, Transcript& t) {
    const size_t k = checked_rounds(proof.L.size(), proof.R.size(), n);
    std::vector<Scalar> xs(k), xinv(k);
    for (size_t j = 0; j < k; j++) {
      Projective lr[2] = {proof.L[j], proof.R[j]};
      t.prover_points(lr, 2);
      xs[j] = t.verifier_scalar();
    }
    S::batch_inv(xs, xinv);                                                     // :128-132
    Msm<Fq, Fr, CURVE> msm(ctx, bases, n_bases);
    msm.upsert_resident(u_index, S::mul(proof.a, proof.b));                     // :125
    std::vector<Scalar> ss(n), ssi(n);
    if (k) {
      check(ctx, bp_s_vector(ctx, CURVE, xs[0].l, xinv[0].l, k, ss[0].l), "bp_s_vector");   // :134-148
      check(ctx, bp_s_vector(ctx, CURVE, xinv[0].l, xs[0].l, k, ssi[0].l), "bp_s_vector");  // :150-154
    } else {
      ss[0] = S::one(); ssi[0] = S::one();
    }
    std::vector<Scalar> as(n), bs(n);
    check(ctx, bp_vec_op(ctx, CURVE, BP_VEC_SCALE, ss[0].l, nullptr, proof.a.l, n, as[0].l), "bp_vec_op");   // :160
    check(ctx, bp_vec_op(ctx, CURVE, BP_VEC_SCALE, ssi[0].l, nullptr, proof.b.l, n, bs[0].l), "bp_vec_op");  // :166
    msm.upsert_resident_range(g_off, as);
    msm.upsert_resident_range(h_off, bs);
    Scalar minus_one = S::neg(S::one());                                        // :170-190
    msm.upsert(P::to_affine(p), minus_one);
    for (size_t j = 0; j < k; j++) {
      msm.upsert(P::to_affine(proof.L[j]), S::mul(minus_one, S::mul(xs[j], xs[j])));
      msm.upsert(P::to_affine(proof.R[j]), S::mul(minus_one, S::mul(xinv[j], xinv[j])));
    }
    return msm;
  }

  // The same check without the accumulator (bp_ipa_verify_msm): both s-vectors are built, scaled by a and
  // b and used as the scalars of gs and hs on the device; returns the group element the deferred MSM
  // evaluates to (the value of verify_aux_resident(...).execute()), the identity iff the proof is valid.
  static Projective verify_value_resident(bp_ctx* ctx, const bp_bases* bases, size_t g_off, size_t h_off, size_t u_index,
                                          const Projective& p, size_t n, const IpaProof& proof, Transcript& t) {
    const size_t k = checked_rounds(proof.L.size(), proof.R.size(), n);
    std::vector<Scalar> xs(k), xinv(k);
    for (size_t j = 0; j < k; j++) {
      Projective lr[2] = {proof.L[j], proof.R[j]};
      t.prover_points(lr, 2);
      xs[j] = t.verifier_scalar();
    }
    S::batch_inv(xs, xinv);                                                     // :128-132
    const size_t ne = 2 * k + 1;
    std::vector<uint64_t> x(4 * ne), y(4 * ne), sc(4 * ne);
    std::vector<uint8_t> inf(ne);
    Scalar minus_one = S::neg(S::one());                                        // :170-190
    auto put = [&](size_t i, const Affine& q, const Scalar& s) {
      memcpy(&x[4 * i], q.x, 32); memcpy(&y[4 * i], q.y, 32); inf[i] = q.inf; memcpy(&sc[4 * i], s.l, 32);
    };
    put(0, P::to_affine(p), minus_one);
    for (size_t j = 0; j < k; j++) {
      put(1 + 2 * j, P::to_affine(proof.L[j]), S::mul(minus_one, S::mul(xs[j], xs[j])));
      put(2 + 2 * j, P::to_affine(proof.R[j]), S::mul(minus_one, S::mul(xinv[j], xinv[j])));
    }
    Scalar ab = S::mul(proof.a, proof.b), dummy = S::one();                     // :125
    Projective out;
    check(ctx, bp_ipa_verify_msm(ctx, bases, g_off, h_off, u_index, k ? xs[0].l : dummy.l, k ? xinv[0].l : dummy.l, k,
                                 proof.a.l, proof.b.l, ab.l, x.data(), y.data(), inf.data(), sc.data(), ne, out.xyz),
          "bp_ipa_verify_msm");
    return out;
  }
  // ipa::verify (src/ipa/mod.rs:196-212) over the registered CRS; a proof of the wrong shape is rejected
  static bool verify_resident(bp_ctx* ctx, const bp_bases* bases, size_t g_off, size_t h_off, size_t u_index,
                              const Projective& p, size_t n, const IpaProof& proof, Transcript& t) {
    try {
      return is_zero(verify_value_resident(ctx, bases, g_off, h_off, u_index, p, n, proof, t));
    } catch (const VerificationError&) {
      return false;
    }
  }

  // verify_batch_aux (src/msm.rs:91-115) over inner-product arguments of one registered CRS with the CRS
  // part of every accumulator built and added on the device (bp_batch_axpy_ipa); P, L_j, R_j are scaled
  // and merged in a host accumulator, one MSM finishes.
  struct BatchItem {
    const Projective* p;
    const IpaProof* proof;
    Transcript* transcript;
  };
  static bool verify_batch_resident(bp_ctx* ctx, const bp_bases* bases, size_t n_bases, size_t g_off, size_t h_off,
                                    size_t u_index, size_t n, const std::vector<BatchItem>& items, const Scalar& alpha) {
    if (items.empty()) throw std::invalid_argument("non-empty vec");
    bp_batch* batch = nullptr;
    check(ctx, bp_batch_begin(ctx, bases, n_bases, &batch), "bp_batch_begin");
    struct Release { bp_batch* b; ~Release() { bp_batch_free(b); } } release{batch};
    Msm<Fq, Fr, CURVE> rest(ctx);
    Scalar pw = S::one(), dummy = S::one(), minus_one = S::neg(S::one());
    for (const BatchItem& it : items) {
      const IpaProof& proof = *it.proof;
      // a proof of the wrong shape rejects the batch (it has no accumulator to contribute)
      size_t k;
      try {
        k = checked_rounds(proof.L.size(), proof.R.size(), n);
      } catch (const VerificationError&) {
        return false;
      }
      std::vector<Scalar> xs(k), xinv(k);
      for (size_t j = 0; j < k; j++) {
        Projective lr[2] = {proof.L[j], proof.R[j]};
        it.transcript->prover_points(lr, 2);
        xs[j] = it.transcript->verifier_scalar();
      }
      S::batch_inv(xs, xinv);
      Scalar ab = S::mul(proof.a, proof.b);
      check(ctx, bp_batch_axpy_ipa(batch, pw.l, g_off, h_off, u_index, k ? xs[0].l : dummy.l, k ? xinv[0].l : dummy.l, k,
                                   proof.a.l, proof.b.l, ab.l),
            "bp_batch_axpy_ipa");
      Scalar mw = S::mul(minus_one, pw);
      rest.upsert(P::to_affine(*it.p), mw);
      for (size_t j = 0; j < k; j++) {
        rest.upsert(P::to_affine(proof.L[j]), S::mul(mw, S::mul(xs[j], xs[j])));
        rest.upsert(P::to_affine(proof.R[j]), S::mul(mw, S::mul(xinv[j], xinv[j])));
      }
      pw = S::mul(pw, alpha);
    }
    const size_t ne = rest.map().size();
    std::vector<uint64_t> x(ne * 4 + 4), y(ne * 4 + 4), sc(ne * 4 + 4);
    std::vector<uint8_t> inf(ne + 1);
    size_t i = 0;
    for (auto& kv : rest.map()) {
      memcpy(&x[4 * i], kv.first.x, 32); memcpy(&y[4 * i], kv.first.y, 32); inf[i] = kv.first.inf;
      memcpy(&sc[4 * i], kv.second.l, 32);
      i++;
    }
    Projective out;
    check(ctx, bp_batch_msm(batch, x.data(), y.data(), inf.data(), sc.data(), ne, out.xyz), "bp_batch_msm");
    return is_zero(out);
  }

  // ipa::verify_aux, src/ipa/mod.rs:102-193; challenges are re-derived from the transcript
  static Msm<Fq, Fr, CURVE> verify_aux(bp_ctx* ctx, const Crs& crs, const Projective& p, size_t n, const IpaProof& proof,
                                       Transcript& t) {
    const size_t k = checked_rounds(proof.L.size(), proof.R.size(), n);
    if (crs.gs.size() < n || crs.hs.size() < n) throw VerificationError("CRS smaller than the statement");
    std::vector<Scalar> xs(k), xinv(k);
    for (size_t j = 0; j < k; j++) {
      Projective lr[2] = {proof.L[j], proof.R[j]};
      t.prover_points(lr, 2);
      xs[j] = t.verifier_scalar();
    }
    S::batch_inv(xs, xinv);                                                     // batch_inversion :128-132
    Msm<Fq, Fr, CURVE> msm(ctx);
    msm.upsert(crs.u, S::mul(proof.a, proof.b));                                // :125
    std::vector<Scalar> ss(n), ssi(n);
    if (k) {
      check(ctx, bp_s_vector(ctx, CURVE, xs[0].l, xinv[0].l, k, ss[0].l), "bp_s_vector");   // :134-148
      check(ctx, bp_s_vector(ctx, CURVE, xinv[0].l, xs[0].l, k, ssi[0].l), "bp_s_vector");  // :150-154
    } else {
      ss[0] = S::one(); ssi[0] = S::one();
    }
    std::vector<Scalar> as(n), bs(n);
    check(ctx, bp_vec_op(ctx, CURVE, BP_VEC_SCALE, ss[0].l, nullptr, proof.a.l, n, as[0].l), "bp_vec_op");   // :160
    check(ctx, bp_vec_op(ctx, CURVE, BP_VEC_SCALE, ssi[0].l, nullptr, proof.b.l, n, bs[0].l), "bp_vec_op");  // :166
    std::vector<Affine> gs(crs.gs.begin(), crs.gs.begin() + n), hs(crs.hs.begin(), crs.hs.begin() + n);
    msm.upsert_batch(gs, as);
    msm.upsert_batch(hs, bs);
    // :170-190: -(P + sum x_j^2 L_j + x_j^-2 R_j)
    Scalar minus_one = S::neg(S::one());
    msm.upsert(P::to_affine(p), minus_one);
    for (size_t j = 0; j < k; j++) {
      msm.upsert(P::to_affine(proof.L[j]), S::mul(minus_one, S::mul(xs[j], xs[j])));
      msm.upsert(P::to_affine(proof.R[j]), S::mul(minus_one, S::mul(xinv[j], xinv[j])));
    }
    return msm;
  }
};

}  // namespace host
}  // namespace bp
