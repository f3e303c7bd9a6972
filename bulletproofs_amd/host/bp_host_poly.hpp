// Polynomial commitment opening proofs (the halo-style IPA of src/poly_commit) on top of
// the GPU hot path: prove, verify, lazy_verify and the fold of deferred checks
// (src/poly_commit/mod.rs:45-315, src/poly_commit/types.rs:137-151) restated in C++.
// Every MSM, the prover's round loop, the s-vector and the lane-wise vector work run
// through libbp_gpu; the transcript stand-in, the blinders and a few single scalar
// multiplications per round stay on the host, as in the reference.
#pragma once
#include "bp_host_range.hpp"  // RandomSource

namespace bp {
namespace host {

struct PolyProof {
  std::vector<Projective> L, R;
  Scalar a, r;
};

struct Todo {  // src/poly_commit/mod.rs:222-226
  Affine g;
  std::vector<Scalar> ui;
};

template <class Fq, class Fr, int CURVE>
struct PolyCommit {
  using S = ScalarOps<Fr>;
  using P = PointOps<Fq, Fr>;

  // poly_commit::types::CRS (src/poly_commit/types.rs:11-15) plus the group generator
  // (G::generator(), used for u) and the gs resident on the GPU.
  struct Crs {
    std::vector<Affine> gs;
    Affine h, gen;
    bp_bases* bases = nullptr;
    FixedBase<Fq, Fr> h_tab, gen_tab;  // multiples of h and of the generator (built by register_bases)
  };
  struct Statement {  // src/poly_commit/types.rs:104-108
    Projective commitment;
    Scalar x, evaluation;
  };

  static void register_bases(bp_ctx* ctx, Crs& crs) {
    size_t n = crs.gs.size();
    crs.h_tab.build(crs.h); crs.gen_tab.build(crs.gen);
    std::vector<uint64_t> x(4 * n), y(4 * n);
    std::vector<uint8_t> inf(n);
    for (size_t i = 0; i < n; i++) { memcpy(&x[4 * i], crs.gs[i].x, 32); memcpy(&y[4 * i], crs.gs[i].y, 32); inf[i] = crs.gs[i].inf; }
    check(ctx, bp_bases_register(ctx, CURVE, x.data(), y.data(), inf.data(), n, &crs.bases), "bp_bases_register");
    // window tables (2 KiB per generator, built once per CRS): a long opening replaces its generators by the
    // folded ones after four rounds, and with tables that is a sum of small multiples of table records
    // (2^20 generators: 3.6 ms instead of 5.3)
    if (n >= kTableFrom && !getenv("BP_POLY_NO_TABLES")) check(ctx, bp_bases_precompute(crs.bases), "bp_bases_precompute");
  }
  static constexpr size_t kTableFrom = (size_t)1 << 15;

  // HPoly::evaluate, src/poly_commit/mod.rs:181-192
  static Scalar hpoly_evaluate(const std::vector<Scalar>& ui, const Scalar& x) {
    Scalar acc = S::one(), xp = x;  // xp = x^(2^i)
    for (size_t i = 0; i < ui.size(); i++) {
      const Scalar& u = ui[ui.size() - 1 - i];
      acc = S::mul(acc, S::add(S::inv(u), S::mul(u, xp)));
      xp = S::mul(xp, xp);
    }
    return acc;
  }

  // Witness::statement, src/poly_commit/types.rs:137-151; coeffs are zero-padded to the CRS size
  static Statement statement(bp_ctx* ctx, const Crs& crs, const std::vector<Scalar>& coeffs_in, const Scalar& r,
                             const Scalar& x) {
    const size_t n = crs.gs.size();
    std::vector<Scalar> padded;
    if (coeffs_in.size() != n) { padded = coeffs_in; padded.resize(n, S::zero()); }
    const std::vector<Scalar>& coeffs = padded.empty() ? coeffs_in : padded;
    Statement st;
    st.x = x;
    Projective g;
    check(ctx, bp_msm_resident(ctx, crs.bases, 0, n, coeffs[0].l, g.xyz), "bp_msm_resident");        // :141
    st.commitment = P::lower(hx_add<Fq>(P::lift(g), crs.h_tab.mul(r)));                               // :142-143
    std::vector<Scalar> pw(n);
    check(ctx, bp_powers(ctx, CURVE, x.l, n, pw[0].l), "bp_powers");
    check(ctx, bp_inner_product(ctx, CURVE, coeffs[0].l, pw[0].l, n, st.evaluation.l), "bp_inner_product");  // :145
    return st;
  }

  // poly_commit::prove, src/poly_commit/mod.rs:45-119
  static Todo prove(bp_ctx* ctx, const Crs& crs, const Statement& st, const std::vector<Scalar>& coeffs_in, Scalar r,
                    Transcript& t, RandomSource& rng, PolyProof& proof) {
    PhaseTrace trace;
    const size_t n = crs.gs.size();
    // zero-padded to the CRS size (:61-64); a polynomial that already has that many coefficients is
    // read where it lies (a copy of 2^20 coefficients costs 7 ms of page faults and memcpy)
    std::vector<Scalar> padded;
    if (coeffs_in.size() != n) { padded = coeffs_in; padded.resize(n, S::zero()); }
    const std::vector<Scalar>& coeffs = padded.empty() ? coeffs_in : padded;
    Affine u = P::to_affine(P::lower(crs.gen_tab.mul(t.verifier_scalar())));             // :55-58
    trace.mark("padding, u");
    // b = powers_of_x(x) (:65) is built where the session keeps it: no 32 bytes per coefficient
    // down to the host and up again
    bp_ipa* s = nullptr;
    check(ctx, bp_ipa_begin_resident_powers(ctx, crs.bases, BP_IPA_POLYCOMMIT, 0, 0, coeffs[0].l, st.x.l, n, &s),
          "bp_ipa_begin_resident_powers");
    trace.mark("session begin");
    Todo todo;
    HXyzz ux = P::lift(u);
    try {
      const bool u_on_device = !u.inf;
      if (u_on_device) {
        uint64_t uxy[8];
        memcpy(uxy, u.x, 32); memcpy(uxy + 4, u.y, 32);
        check(ctx, bp_ipa_set_u(s, uxy), "bp_ipa_set_u");
      }
      while (bp_ipa_len(s) != 1) {
        Scalar l_j = rng.scalar(), r_j = rng.scalar();                                  // :75
        Projective lm, rm; Scalar cl, cr;
        check(ctx, bp_ipa_round_lr(s, lm.xyz, rm.xyz, cl.l, cr.l), "bp_ipa_round_lr");  // :78,80 / :83,85
        HXyzz left = hx_add<Fq>(P::lift(lm), crs.h_tab.mul(l_j));                       // :79
        HXyzz right = hx_add<Fq>(P::lift(rm), crs.h_tab.mul(r_j));                      // :84
        if (!u_on_device) { left = hx_add<Fq>(left, P::mul(ux, cl)); right = hx_add<Fq>(right, P::mul(ux, cr)); }
        Projective lr[2] = {P::lower(left), P::lower(right)};
        t.prover_points(lr, 2);                                                         // :89
        Scalar u_j = t.verifier_scalar();                                               // :91
        Scalar u_inv = S::inv(u_j);
        todo.ui.push_back(u_j);
        check(ctx, bp_ipa_round_fold(s, u_j.l, u_inv.l), "bp_ipa_round_fold");          // :95-102
        r = S::add(r, S::add(S::mul(l_j, S::mul(u_j, u_j)), S::mul(r_j, S::mul(u_inv, u_inv))));  // :104
        proof.L.push_back(lr[0]); proof.R.push_back(lr[1]);
      }
      trace.mark("rounds");
      // a[0] and the folded generator g[0]; the session computes g[0] as <s-vector, gs>,
      // which is the value the reference asserts it to equal (:108-113)
      uint8_t ginf = 0;
      check(ctx, bp_ipa_read(s, todo.g.x, todo.g.y, &ginf, nullptr, nullptr, nullptr, proof.a.l, nullptr), "bp_ipa_read");
      todo.g.inf = ginf;
      trace.mark("final a, g");
    } catch (...) { bp_ipa_free(s); throw; }
    bp_ipa_free(s);
    trace.mark("session free");
    proof.r = r;
    Scalar msg[2] = {proof.a, proof.r};
    t.prover_scalars(msg, 2);                                                           // :107
    return todo;
  }

  // The common tail of verify (:147-172) and lazy_verify (:292-308):
  // a*g + r*h + (a*b)*u - commitment - evaluation*u - sum_j (u_j^2 L_j + u_j^-2 R_j) == 0, as one MSM.
  static bool final_check(bp_ctx* ctx, const Crs& crs, const Statement& st, const PolyProof& proof, const Affine& u,
                          const std::vector<Scalar>& ui, const Affine& g, const Scalar& b) {
    const size_t k = ui.size();
    std::vector<Affine> pts;
    std::vector<Scalar> sc;
    const Scalar minus_one = S::neg(S::one());
    pts.push_back(g); sc.push_back(proof.a);
    pts.push_back(crs.h); sc.push_back(proof.r);
    pts.push_back(u); sc.push_back(S::add(S::mul(proof.a, b), S::neg(st.evaluation)));
    pts.push_back(P::to_affine(st.commitment)); sc.push_back(minus_one);
    for (size_t j = 0; j < k; j++) {
      Scalar u_inv = S::inv(ui[j]);
      pts.push_back(P::to_affine(proof.L[j])); sc.push_back(S::neg(S::mul(ui[j], ui[j])));
      pts.push_back(P::to_affine(proof.R[j])); sc.push_back(S::neg(S::mul(u_inv, u_inv)));
    }
    const size_t m = pts.size();
    std::vector<uint64_t> x(4 * m), y(4 * m), s(4 * m);
    std::vector<uint8_t> inf(m);
    for (size_t i = 0; i < m; i++) { memcpy(&x[4 * i], pts[i].x, 32); memcpy(&y[4 * i], pts[i].y, 32); inf[i] = pts[i].inf; memcpy(&s[4 * i], sc[i].l, 32); }
    Projective res;
    check(ctx, bp_msm(ctx, CURVE, x.data(), y.data(), inf.data(), s.data(), m, res.xyz), "bp_msm");
    return is_zero(res);
  }

  static void replay(const Crs& crs, const PolyProof& proof, Transcript& t, Affine& u, std::vector<Scalar>& ui) {
    // the reference reads exactly log2(crs.size()) pairs (:129-145, :274-290)
    checked_rounds(proof.L.size(), proof.R.size(), crs.gs.size());
    u = P::to_affine(P::lower(crs.gen_tab.mul(t.verifier_scalar())));                   // :132-135
    for (size_t j = 0; j < proof.L.size(); j++) {
      Projective lr[2] = {proof.L[j], proof.R[j]};
      t.prover_points(lr, 2);
      ui.push_back(t.verifier_scalar());                                                // :139-145
    }
    Scalar msg[2] = {proof.a, proof.r};
    t.prover_scalars(msg, 2);                                                           // :165
  }

  // poly_commit::verify, src/poly_commit/mod.rs:121-173
  static bool verify(bp_ctx* ctx, const Crs& crs, const Statement& st, const PolyProof& proof, Transcript& t) {
    Affine u; std::vector<Scalar> ui;
    try { replay(crs, proof, t, u, ui); } catch (const VerificationError&) { return false; }
    const size_t k = ui.size();
    std::vector<Scalar> uinv(k);
    for (size_t j = 0; j < k; j++) uinv[j] = S::inv(ui[j]);
    Projective g;  // <h_poly.coeffs(), gs> with the coefficients built and consumed on the device (:152-159)
    Scalar dummy = S::one();
    check(ctx, bp_msm_resident_s_vector(ctx, crs.bases, 0, k ? ui[0].l : dummy.l, k ? uinv[0].l : dummy.l, k, g.xyz),
          "bp_msm_resident_s_vector");
    // b = <ss, powers(x)> (:160-163) equals HPoly::evaluate(x) (src/poly_commit/mod.rs:324-346)
    Scalar b = hpoly_evaluate(ui, st.x);
    return final_check(ctx, crs, st, proof, u, ui, P::to_affine(g), b);
  }

  // Several openings over one CRS checked together, the way verify_batch_aux (src/msm.rs:91-115) combines
  // deferred MSMs: sum_i alpha^i * (opening i's final check) == 0.  What makes one verification expensive, the
  // MSM <h_poly_i.coeffs(), gs> over the whole CRS (src/poly_commit/mod.rs:152-159), is paid once for the
  // batch: every opening's coefficients are built on the device and added into one accumulator over the
  // resident CRS with the weight alpha^i a_i (bp_batch_axpy_s_vector); the other terms (h, the generator of
  // the u_i, commitments, L_j, R_j) join that one MSM as extra points.  A malformed proof rejects the batch.
  struct BatchItem {
    const Statement* st;
    const PolyProof* proof;
    Transcript* t;
  };
  static bool verify_batch(bp_ctx* ctx, const Crs& crs, const std::vector<BatchItem>& items, const Scalar& alpha) {
    if (items.empty()) throw std::invalid_argument("non-empty vec");
    const size_t n = crs.gs.size();
    bp_batch* batch = nullptr;
    check(ctx, bp_batch_begin(ctx, crs.bases, n, &batch), "bp_batch_begin");
    struct Release { bp_batch* b; ~Release() { bp_batch_free(b); } } release{batch};
    std::vector<Affine> pts;
    std::vector<Scalar> sc;
    Scalar w = S::one(), h_scalar = S::zero(), gen_scalar = S::zero();
    for (const BatchItem& it : items) {
      const PolyProof& proof = *it.proof;
      // replay with the scalar of u kept: u_i = c_i * gen (:132-135), so all u_i share the point gen
      try { checked_rounds(proof.L.size(), proof.R.size(), n); } catch (const VerificationError&) { return false; }
      const Scalar c = it.t->verifier_scalar();
      std::vector<Scalar> ui, uinv;
      for (size_t j = 0; j < proof.L.size(); j++) {
        Projective lr[2] = {proof.L[j], proof.R[j]};
        it.t->prover_points(lr, 2);
        ui.push_back(it.t->verifier_scalar());                                            // :139-145
      }
      Scalar msg[2] = {proof.a, proof.r};
      it.t->prover_scalars(msg, 2);                                                       // :165
      S::batch_inv(ui, uinv);
      const size_t k = ui.size();
      const Scalar b = hpoly_evaluate(ui, it.st->x);                                      // :160-163
      const Scalar wa = S::mul(w, proof.a);
      Scalar dummy = S::one();
      check(ctx, bp_batch_axpy_s_vector(batch, wa.l, 0, k ? ui[0].l : dummy.l, k ? uinv[0].l : dummy.l, k),
            "bp_batch_axpy_s_vector");
      h_scalar = S::add(h_scalar, S::mul(w, proof.r));
      gen_scalar = S::add(gen_scalar, S::mul(S::mul(w, c), S::add(S::mul(proof.a, b), S::neg(it.st->evaluation))));
      const Scalar mw = S::neg(w);
      pts.push_back(P::to_affine(it.st->commitment)); sc.push_back(mw);
      for (size_t j = 0; j < k; j++) {
        pts.push_back(P::to_affine(proof.L[j])); sc.push_back(S::mul(mw, S::mul(ui[j], ui[j])));
        pts.push_back(P::to_affine(proof.R[j])); sc.push_back(S::mul(mw, S::mul(uinv[j], uinv[j])));
      }
      w = S::mul(w, alpha);
    }
    pts.push_back(crs.h); sc.push_back(h_scalar);
    pts.push_back(crs.gen); sc.push_back(gen_scalar);
    const size_t m = pts.size();
    std::vector<uint64_t> x(4 * m), y(4 * m), s(4 * m);
    std::vector<uint8_t> inf(m);
    for (size_t i = 0; i < m; i++) { memcpy(&x[4 * i], pts[i].x, 32); memcpy(&y[4 * i], pts[i].y, 32); inf[i] = pts[i].inf; memcpy(&s[4 * i], sc[i].l, 32); }
    Projective res;
    check(ctx, bp_batch_msm(batch, x.data(), y.data(), inf.data(), s.data(), m, res.xyz), "bp_batch_msm");
    return is_zero(res);
  }

  // poly_commit::lazy_verify, src/poly_commit/mod.rs:264-315: the MSM over the CRS is deferred
  static bool lazy_verify(bp_ctx* ctx, const Crs& crs, const Statement& st, const Affine& assumption, const PolyProof& proof,
                          Transcript& t, std::vector<Todo>& todos) {
    Affine u; std::vector<Scalar> ui;
    try { replay(crs, proof, t, u, ui); } catch (const VerificationError&) { return false; }
    Scalar b = hpoly_evaluate(ui, st.x);                                                // :301
    if (!final_check(ctx, crs, st, proof, u, ui, assumption, b)) return false;
    todos.push_back(Todo{assumption, ui});                                              // :309-313
    return true;
  }

  // fold_todos_witness, src/poly_commit/mod.rs:228-244: sum_i alpha^i * coeffs(h_poly_i), r = 0
  static std::vector<Scalar> fold_todos_witness(bp_ctx* ctx, const std::vector<Todo>& todos, const Scalar& alpha) {
    if (todos.empty()) throw std::invalid_argument("Non empty statement list");
    const size_t k = todos[0].ui.size(), n = (size_t)1 << k, m = todos.size();
    // all h_poly coefficient vectors are built and summed on the device (bp_fold_s_vectors)
    std::vector<Scalar> acc(n), ch(m * k), chi(m * k), w(m);
    Scalar ai = S::one();
    for (size_t i = 0; i < m; i++) {
      if (todos[i].ui.size() != k) throw std::invalid_argument("deferred checks of different sizes");
      for (size_t j = 0; j < k; j++) {
        ch[i * k + j] = todos[i].ui[j];
        chi[i * k + j] = S::inv(todos[i].ui[j]);
      }
      w[i] = ai;
      ai = S::mul(ai, alpha);
    }
    Scalar dummy = S::one();
    check(ctx, bp_fold_s_vectors(ctx, CURVE, k ? ch[0].l : dummy.l, k ? chi[0].l : dummy.l, w[0].l, k, m, acc[0].l),
          "bp_fold_s_vectors");
    return acc;
  }

  // fold_todos_statement, src/poly_commit/mod.rs:246-262
  static Statement fold_todos_statement(bp_ctx* ctx, const std::vector<Todo>& todos, const Scalar& alpha, const Scalar& x) {
    if (todos.empty()) throw std::invalid_argument("Non empty statement list");
    const size_t m = todos.size();
    std::vector<uint64_t> gx(4 * m), gy(4 * m), s(4 * m);
    std::vector<uint8_t> inf(m);
    Statement st;
    st.x = x;
    st.evaluation = S::zero();
    Scalar ai = S::one();
    for (size_t i = 0; i < m; i++) {
      memcpy(&gx[4 * i], todos[i].g.x, 32); memcpy(&gy[4 * i], todos[i].g.y, 32); inf[i] = todos[i].g.inf;
      memcpy(&s[4 * i], ai.l, 32);
      st.evaluation = S::add(st.evaluation, S::mul(ai, hpoly_evaluate(todos[i].ui, x)));
      ai = S::mul(ai, alpha);
    }
    check(ctx, bp_msm(ctx, CURVE, gx.data(), gy.data(), inf.data(), s.data(), m, st.commitment.xyz), "bp_msm");
    return st;
  }
};

}  // namespace host
}  // namespace bp
