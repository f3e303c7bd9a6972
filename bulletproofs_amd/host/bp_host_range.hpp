// Aggregated range proof (Bulletproofs section 4.3) on top of the GPU hot path: the
// caller-side protocol logic of src/range/aggregate.rs:23-300 restated in C++ so that
// the BASELINE configuration "m x n_bits-bit aggregated range proof" can be driven and
// timed end to end.  Every MSM, the hs' scaling, the IPA rounds and the verifier's
// final MSM run through libbp_gpu; the O(N) scalar-vector construction, the transcript
// stand-in and a handful of single scalar multiplications stay on the host, as they do
// in the reference.  Randomness and challenges are supplied by the caller (seeded).
#pragma once
#include "bp_host.hpp"

namespace bp {
namespace host {

struct RandomSource {
  virtual Scalar scalar() = 0;  // uniform element of Fr, Montgomery form
  // A source whose scalars are consecutive draws of a SplitMix64 stream (four outputs per draw, little-endian
  // limbs, a draw at or above the group order discarded) reports the stream's state, and skips `draws` draws
  // that were all accepted: the batched prover then has the device produce a proof's 2N blinders from that
  // state (bp_range_batch_commit) instead of uploading them.  Any other source returns false.
  virtual bool splitmix_state(uint64_t* state) const { (void)state; return false; }
  virtual void splitmix_skip(size_t draws) { (void)draws; }
  virtual ~RandomSource() {}
};

struct RangeProof {
  Projective a, s, tt1, tt2;
  Scalar tao_x, mu, t_hat;
  IpaProof ipa;
};

template <class Fq, class Fr, int CURVE>
struct RangeAggregate {
  using S = ScalarOps<Fr>;
  using P = PointOps<Fq, Fr>;
  using I = Ipa<Fq, Fr, CURVE>;
  using M = Msm<Fq, Fr, CURVE>;

  // range::types::CRS (src/range/types.rs:13-17); the first N gs then the first N hs are
  // also kept resident on the GPU as one base set (bases = gs || hs).
  struct Crs {
    typename I::Crs ipa;
    Affine g, h;
    size_t n = 0;            // N = n_bits * m actually used
    bp_bases* gh = nullptr;  // gs[0..N) || hs[0..N) || ipa.u
    bp_bases* ghb = nullptr; // gs[0..N) || hs[0..N) || ipa.u || g || h with window tables (register_batch_bases)
    FixedBase<Fq, Fr> g_tab, h_tab, u_tab;  // multiples of g, h, ipa.u (built by register_bases)
  };

  static void register_bases(bp_ctx* ctx, Crs& crs, size_t n) {
    std::vector<uint64_t> x(8 * n + 4), y(8 * n + 4);
    std::vector<uint8_t> inf(2 * n + 1, 0);
    for (size_t i = 0; i < n; i++) {
      memcpy(&x[4 * i], crs.ipa.gs[i].x, 32); memcpy(&y[4 * i], crs.ipa.gs[i].y, 32);
      memcpy(&x[4 * (n + i)], crs.ipa.hs[i].x, 32); memcpy(&y[4 * (n + i)], crs.ipa.hs[i].y, 32);
    }
    memcpy(&x[8 * n], crs.ipa.u.x, 32); memcpy(&y[8 * n], crs.ipa.u.y, 32); inf[2 * n] = crs.ipa.u.inf;
    crs.n = n;
    crs.g_tab.build(crs.g); crs.h_tab.build(crs.h); crs.u_tab.build(crs.ipa.u);
    check(ctx, bp_bases_register(ctx, CURVE, x.data(), y.data(), inf.data(), 2 * n + 1, &crs.gh), "bp_bases_register");
    // window tables of the CRS (2 KiB per point, built once): the sums of the argument's rounds then
    // come back with nothing left to double on the host, which is a fifth of a round of 2 * 4096 points
    if (2 * n + 1 <= kTablePoints) check(ctx, bp_bases_precompute(crs.gh), "bp_bases_precompute");
  }
  static constexpr size_t kTablePoints = (size_t)1 << 15;

  static Scalar from_u64(uint64_t v) { Scalar c{{v, 0, 0, 0}}; return S::to_mont(c); }
  static Scalar sub(const Scalar& a, const Scalar& b) { return S::add(a, S::neg(b)); }

  static Projective msm_gh(bp_ctx* ctx, const Crs& crs, const std::vector<Scalar>& lo, const std::vector<Scalar>& hi) {
    std::vector<Scalar> s(lo);
    s.insert(s.end(), hi.begin(), hi.end());
    Projective out;
    check(ctx, bp_msm_resident(ctx, crs.gh, 0, s.size(), s[0].l, out.xyz), "bp_msm_resident");
    return out;
  }

  // hs'_i = y^-i * h_i (create_hs_prime, src/range/utils.rs:22-29, + normalize_batch)
  static void hs_prime(bp_ctx* ctx, const Crs& crs, const Scalar& y, std::vector<Scalar>& y_inv_pows,
                       std::vector<Affine>& out) {
    size_t n = crs.n;
    Scalar y_inv = S::inv(y);
    y_inv_pows.resize(n);
    check(ctx, bp_powers(ctx, CURVE, y_inv.l, n, y_inv_pows[0].l), "bp_powers");
    std::vector<uint64_t> hx, hy, ox(4 * n), oy(4 * n); std::vector<uint8_t> hi, oi(n);
    I::planes(crs.ipa.hs, n, hx, hy, hi);
    check(ctx, bp_scale_points(ctx, CURVE, hx.data(), hy.data(), hi.data(), y_inv_pows[0].l, n, ox.data(), oy.data(),
                               oi.data()), "bp_scale_points");
    out.resize(n);
    for (size_t i = 0; i < n; i++) { memcpy(out[i].x, &ox[4 * i], 32); memcpy(out[i].y, &oy[4 * i], 32); out[i].inf = oi[i]; }
  }

  static Scalar dot(bp_ctx* ctx, const std::vector<Scalar>& a, const std::vector<Scalar>& b) {
    Scalar r;
    check(ctx, bp_inner_product(ctx, CURVE, a[0].l, b[0].l, a.size(), r.l), "bp_inner_product");
    return r;
  }

  // src/range/aggregate.rs:23-170.  values: m integers below 2^n_bits; gamma: m blinders.
  static RangeProof prove(bp_ctx* ctx, const Crs& crs, const std::vector<uint64_t>& values,
                          const std::vector<Scalar>& gamma, unsigned n_bits, Transcript& t, RandomSource& rng) {
    const size_t m = values.size(), n = (size_t)n_bits * m;
    if (n != crs.n) throw std::invalid_argument("CRS registered for another size");
    const Scalar one = S::one(), zero = S::zero();
    PhaseTrace tr;
    std::vector<Scalar> a_l(n), a_r(n);                      // :50-64
    for (size_t j = 0; j < m; j++)
      for (unsigned i = 0; i < n_bits; i++) {
        bool bit = (values[j] >> i) & 1;
        a_l[j * n_bits + i] = bit ? one : zero;
        a_r[j * n_bits + i] = bit ? zero : S::neg(one);
      }
    Scalar alpha = rng.scalar(), rho = rng.scalar();         // :66-67
    std::vector<Scalar> s_l(n), s_r(n);
    for (auto& v : s_l) v = rng.scalar();                    // :68-73
    for (auto& v : s_r) v = rng.scalar();
    RangeProof pr;
    tr.mark("bits, blinders");
    // :80-83 -- the two MSMs of rayon::join as one pass with two sums over gs || hs
    Projective a_msm, s_msm;
    {
      std::vector<Scalar> sa(a_l), ss(s_l);
      sa.insert(sa.end(), a_r.begin(), a_r.end());
      ss.insert(ss.end(), s_r.begin(), s_r.end());
      check(ctx, bp_msm_resident_pair(ctx, crs.gh, 0, 2 * n, sa[0].l, ss[0].l, a_msm.xyz, s_msm.xyz), "bp_msm_resident_pair");
    }
    pr.a = P::lower(hx_add<Fq>(crs.h_tab.mul(alpha), P::lift(a_msm)));   // :81
    pr.s = P::lower(hx_add<Fq>(crs.h_tab.mul(rho), P::lift(s_msm)));     // :82
    tr.mark("A, S commitments");
    Projective as[2] = {pr.a, pr.s};
    t.prover_points(as, 2);                                  // :85
    Scalar y = t.verifier_scalar(), z = t.verifier_scalar(); // :86
    // :88-127 -- y^n, l(X) = (a_L - z) + s_L X, r(X) = y^n o (a_R + z) + z^(2+j) 2^i + (y^n o s_R) X and
    // t1 = <l0,r1> + <l1,r0>, t2 = <l1,r1> (src/range/types.rs:98-117) are built on the device from
    // the values and the blinders; the vectors stay there for the evaluation at x
    Scalar zz = S::mul(z, z), zpow = zz;
    Scalar t1, t2;
    check(ctx, bp_range_agg_poly(ctx, CURVE, values.data(), m, n_bits, s_l[0].l, s_r[0].l, y.l, z.l, t1.l, t2.l),
          "bp_range_agg_poly");
    tr.mark("powers, l/r polynomials");
    Scalar tao1 = rng.scalar(), tao2 = rng.scalar();         // :121-122
    pr.tt1 = P::lower(hx_add<Fq>(crs.g_tab.mul(t1), crs.h_tab.mul(tao1)));  // :126
    pr.tt2 = P::lower(hx_add<Fq>(crs.g_tab.mul(t2), crs.h_tab.mul(tao2)));  // :127
    tr.mark("t1, t2, T1, T2");
    Projective tt[2] = {pr.tt1, pr.tt2};
    t.prover_points(tt, 2);                                  // :129
    Scalar x = t.verifier_scalar();                          // :133
    Scalar sigma = zero;                                     // :136-139
    zpow = zz;
    for (size_t j = 0; j < m; j++) { sigma = S::add(sigma, S::mul(zpow, gamma[j])); zpow = S::mul(zpow, z); }
    pr.tao_x = S::add(S::add(S::mul(tao1, x), S::mul(tao2, S::mul(x, x))), sigma);  // :140
    pr.mu = S::add(alpha, S::mul(rho, x));                   // :142
    std::vector<Scalar> l(n), r(n);                          // :143-144, Witness::new :146
    check(ctx, bp_range_agg_eval(ctx, CURVE, x.l, n, l[0].l, r[0].l, pr.t_hat.l), "bp_range_agg_eval");
    // :148-153 -- hs'_i = y^-i h_i is not materialised (N scalar multiplications in the
    // reference): the IPA session over the resident gs || hs carries y^-i as the initial
    // coefficient of h_i, which yields the same L, R, a, b.
    std::vector<Scalar> y_inv_pows(n);
    Scalar y_inv = S::inv(y);
    check(ctx, bp_powers(ctx, CURVE, y_inv.l, n, y_inv_pows[0].l), "bp_powers");
    // :155-157 -- extended_statement.p is updated by the prover but never sent
    // (src/ipa/mod.rs:94 is its only use), so it is not computed; its c is t_hat.
    Scalar msg[3] = {pr.tao_x, pr.mu, pr.t_hat};
    t.prover_scalars(msg, 3);                                // :159
    // ipa::extended::prove (src/ipa/extended.rs:15-35) over gs, hs', u' = x'*u
    Scalar xe = t.verifier_scalar();
    tr.mark("l, r, t_hat, y^-i, u'");
    // u' = x' * u is bases[2N] with the factor x' (no scalar multiplication on the host)
    pr.ipa = I::prove_resident(ctx, crs.gh, 0, n, y_inv_pows.data(), 2 * n, xe, l, r, t);  // :166
    tr.mark("ipa rounds");
    return pr;
  }

  // The base set of prove_batch: every point a proof's commitments use, so that A, S, T1, T2 and the
  // rounds' L, R all come from passes over window tables.  Freed by the caller (bp_bases_free(crs.ghb)).
  static void register_batch_bases(bp_ctx* ctx, Crs& crs) {
    const size_t n = crs.n, total = 2 * n + 3;
    std::vector<uint64_t> x(4 * total), y(4 * total);
    std::vector<uint8_t> inf(total, 0);
    auto put = [&](size_t i, const Affine& p) { memcpy(&x[4 * i], p.x, 32); memcpy(&y[4 * i], p.y, 32); inf[i] = p.inf; };
    for (size_t i = 0; i < n; i++) { put(i, crs.ipa.gs[i]); put(n + i, crs.ipa.hs[i]); }
    put(2 * n, crs.ipa.u); put(2 * n + 1, crs.g); put(2 * n + 2, crs.h);
    check(ctx, bp_bases_register(ctx, CURVE, x.data(), y.data(), inf.data(), total, &crs.ghb), "bp_bases_register");
    check(ctx, bp_bases_precompute(crs.ghb), "bp_bases_precompute");
  }

  // B = bp_range_batch_size(rb) proofs of src/range/aggregate.rs:23-170 proved together: the phases of
  // prove() above with every device step shared by the batch (bp_range_batch_*), the transcripts and the
  // random sources kept apart.  values: B x m, gamma: B x m; proof b talks to ts[b] and draws from rngs[b] in
  // prove()'s order, so it is the proof prove() returns for the same inputs.
  static std::vector<RangeProof> prove_batch(bp_ctx* ctx, const Crs& crs, bp_range_batch* rb, const uint64_t* values,
                                             const Scalar* gamma, size_t m, unsigned n_bits, Transcript* const* ts,
                                             RandomSource* const* rngs) {
    const size_t B = bp_range_batch_size(rb), n = (size_t)n_bits * m;
    if (n != crs.n || B == 0) throw std::invalid_argument("CRS registered for another size");
    std::vector<RangeProof> prs(B);
    PhaseTrace tr;
    std::vector<Scalar> alpha(B), rho(B);
    for (size_t b = 0; b < B; b++) { alpha[b] = rngs[b]->scalar(); rho[b] = rngs[b]->scalar(); }   // :66-67
    std::vector<Projective> pts(2 * B);
    {
      // :68-73 -- blinders from the device when every source is a stream it can reproduce
      std::vector<uint64_t> seeds(B);
      bool on_device = CURVE == BP_SECP256K1;  // a 256-bit draw is below Vesta's order one time in four
      for (size_t b = 0; b < B && on_device; b++) on_device = rngs[b]->splitmix_state(&seeds[b]);
      int rc = BP_E_STATE;
      if (on_device) rc = bp_range_batch_commit(rb, values, nullptr, seeds.data(), 0, alpha[0].l, rho[0].l, pts[0].xyz);
      if (on_device && rc == BP_OK) {
        for (size_t b = 0; b < B; b++) rngs[b]->splitmix_skip(2 * n);
      } else if (rc == BP_E_STATE) {
        std::vector<Scalar> s_lr(B * 2 * n);
        for (size_t b = 0; b < B; b++)
          for (size_t i = 0; i < 2 * n; i++) s_lr[b * 2 * n + i] = rngs[b]->scalar();
        rc = bp_range_batch_commit(rb, values, s_lr[0].l, nullptr, 0, alpha[0].l, rho[0].l, pts[0].xyz);
      }
      check(ctx, rc, "bp_range_batch_commit");                                                     // :80-83
    }
    tr.mark("batch: A, S");
    std::vector<Scalar> y(B), z(B), y_inv;
    for (size_t b = 0; b < B; b++) {
      prs[b].a = pts[2 * b]; prs[b].s = pts[2 * b + 1];
      ts[b]->prover_points(&pts[2 * b], 2);                                                        // :85
      y[b] = ts[b]->verifier_scalar(); z[b] = ts[b]->verifier_scalar();                            // :86
    }
    S::batch_inv(y, y_inv);
    std::vector<Scalar> t12(2 * B), tsc(4 * B);
    check(ctx, bp_range_batch_poly(rb, y[0].l, y_inv[0].l, z[0].l, t12[0].l), "bp_range_batch_poly");  // :88-120
    tr.mark("batch: transcript, polynomials");
    for (size_t b = 0; b < B; b++) {
      tsc[4 * b] = t12[2 * b]; tsc[4 * b + 1] = rngs[b]->scalar();                                 // :121-122
      tsc[4 * b + 2] = t12[2 * b + 1]; tsc[4 * b + 3] = rngs[b]->scalar();
    }
    check(ctx, bp_range_batch_tcommit(rb, tsc[0].l, pts[0].xyz), "bp_range_batch_tcommit");        // :126-127
    tr.mark("batch: T1, T2");
    std::vector<Scalar> x(B), xi;
    for (size_t b = 0; b < B; b++) {
      prs[b].tt1 = pts[2 * b]; prs[b].tt2 = pts[2 * b + 1];
      ts[b]->prover_points(&pts[2 * b], 2);                                                        // :129
      x[b] = ts[b]->verifier_scalar();                                                             // :133
      Scalar sigma = S::zero(), zpow = S::mul(z[b], z[b]);                                         // :136-139
      for (size_t j = 0; j < m; j++) { sigma = S::add(sigma, S::mul(zpow, gamma[b * m + j])); zpow = S::mul(zpow, z[b]); }
      prs[b].tao_x = S::add(S::add(S::mul(tsc[4 * b + 1], x[b]), S::mul(tsc[4 * b + 3], S::mul(x[b], x[b]))), sigma);  // :140
      prs[b].mu = S::add(alpha[b], S::mul(rho[b], x[b]));                                          // :142
    }
    std::vector<Scalar> t_hat(B), xe(B);
    check(ctx, bp_range_batch_eval(rb, x[0].l, t_hat[0].l), "bp_range_batch_eval");                // :143-146
    for (size_t b = 0; b < B; b++) {
      prs[b].t_hat = t_hat[b];
      Scalar msg[3] = {prs[b].tao_x, prs[b].mu, prs[b].t_hat};
      ts[b]->prover_scalars(msg, 3);                                                               // :159
      xe[b] = ts[b]->verifier_scalar();                                                            // src/ipa/extended.rs:22
    }
    check(ctx, bp_range_batch_ipa_begin(rb, xe[0].l), "bp_range_batch_ipa_begin");                 // :148-166
    tr.mark("batch: transcript, eval");
    while (bp_range_batch_ipa_len(rb) != 1) {
      check(ctx, bp_range_batch_round_lr(rb, pts[0].xyz), "bp_range_batch_round_lr");              // src/ipa/mod.rs:43-71
      tr.mark("batch: round L, R");
      for (size_t b = 0; b < B; b++) {
        ts[b]->prover_points(&pts[2 * b], 2);                                                      // :75
        x[b] = ts[b]->verifier_scalar();                                                           // :77
        prs[b].ipa.L.push_back(pts[2 * b]); prs[b].ipa.R.push_back(pts[2 * b + 1]);
      }
      S::batch_inv(x, xi);                                                                         // :78
      check(ctx, bp_range_batch_round_fold(rb, x[0].l, xi[0].l), "bp_range_batch_round_fold");     // :80-92
      tr.mark("batch: transcript, fold");
    }
    check(ctx, bp_range_batch_finish(rb, t12[0].l), "bp_range_batch_finish");
    for (size_t b = 0; b < B; b++) {
      prs[b].ipa.a = t12[2 * b]; prs[b].ipa.b = t12[2 * b + 1];
      ts[b]->prover_scalars(&t12[2 * b], 2);                                                       // :97
    }
    return prs;
  }

  // src/range/aggregate.rs:173-282 as ONE deferred MSM.  The reference evaluates three group
  // expressions on the way (the V_j combination :226, P :259-264, the T1/T2 terms :222-224) and
  // hands the inner-product verifier a point; every one of them enters the returned accumulator
  // linearly, so the same group element is reached by adding their terms to the accumulator
  // instead: gs_i and hs_i collect a*s_i + z and y^-i (b*s_i^-1 - z^(2+j) 2^i) - z (built on the
  // device, bp_range_agg_verify_scalars), and A, S, T1, T2, V_j, L_j, R_j, g, h, u carry the rest.
  // execute() returns the point verify_aux_stepwise's accumulator evaluates to.
  // The host's part of verify_aux: the challenges, and every term of the deferred MSM that is not a
  // scalar of gs or hs.  No device call.
  struct VerifyPlan {
    size_t k = 0, m = 0, n = 0;
    std::vector<Scalar> xs, xinv;
    Scalar y_inv, z, u_scalar;
    std::vector<std::pair<Affine, Scalar>> terms;  // in the order the reference meets them
  };
  static VerifyPlan verify_plan(const Crs& crs, const std::vector<Projective>& v_commit, unsigned n_bits,
                                const RangeProof& pr, Transcript& t, RandomSource& rng) {
    VerifyPlan vp;
    const size_t m = v_commit.size(), n = (size_t)n_bits * m;
    if (n != crs.n) throw std::invalid_argument("CRS registered for another size");
    vp.m = m; vp.n = n;
    const Scalar one = S::one();
    Projective as[2] = {pr.a, pr.s};
    t.prover_points(as, 2);
    Scalar y = t.verifier_scalar(), z = t.verifier_scalar();
    Projective tt[2] = {pr.tt1, pr.tt2};
    t.prover_points(tt, 2);
    Scalar x = t.verifier_scalar();
    Scalar msg[3] = {pr.tao_x, pr.mu, pr.t_hat};
    t.prover_scalars(msg, 3);
    Scalar xe = t.verifier_scalar();
    // the inner-product argument's challenges (src/ipa/mod.rs:116-132)
    const size_t k = checked_rounds(pr.ipa.L.size(), pr.ipa.R.size(), n);
    vp.k = k;
    std::vector<Scalar>& xs = vp.xs;
    std::vector<Scalar>& xinv = vp.xinv;
    xs.resize(k); xinv.resize(k);
    for (size_t j = 0; j < k; j++) {
      Projective lr[2] = {pr.ipa.L[j], pr.ipa.R[j]};
      t.prover_points(lr, 2);
      xs[j] = t.verifier_scalar();
    }
    // one inversion for the k challenges, y and y - 1 (ark_ff::batch_inversion, src/ipa/mod.rs:128-132)
    Scalar y_inv, ym1_inv;
    {
      std::vector<Scalar> in(xs), out;
      in.push_back(y);
      in.push_back(sub(y, one));
      S::batch_inv(in, out);
      for (size_t j = 0; j < k; j++) xinv[j] = out[j];
      y_inv = out[k];
      ym1_inv = out[k + 1];
    }
    vp.y_inv = y_inv; vp.z = z;
    Scalar zz = S::mul(z, z);
    // delta(y,z) = (z - z^2) <1, y^N> - sum_j z^(3+j) <1, 2^n>   (:209-215); <1, y^N> = (y^N - 1)/(y - 1)
    Scalar sum_y;
    {
      Scalar yn = one, base = y;
      for (size_t e = n; e; e >>= 1) { if (e & 1) yn = S::mul(yn, base); base = S::mul(base, base); }
      sum_y = S::is_zero(ym1_inv) ? from_u64(n) : S::mul(sub(yn, one), ym1_inv);
    }
    Scalar two = from_u64(2), d = S::zero(), p2 = one;
    for (unsigned i = 0; i < n_bits; i++) { d = S::add(d, p2); p2 = S::mul(p2, two); }
    Scalar zsum = S::zero(), zp = S::mul(zz, z);
    for (size_t j = 0; j < m; j++) { zsum = S::add(zsum, zp); zp = S::mul(zp, z); }
    Scalar delta = sub(S::mul(sub(z, zz), sum_y), S::mul(zsum, d));
    // u: a*b*x' from the inner-product verifier (src/ipa/mod.rs:125, scale_elem :270) minus x'*t_hat from P
    vp.u_scalar = S::mul(xe, sub(S::mul(pr.ipa.a, pr.ipa.b), pr.t_hat));
    auto term = [&](const Affine& p, const Scalar& s) { vp.terms.emplace_back(p, s); };
    Scalar minus_one = S::neg(one);
    // -P = -A - x S + mu h (+ the gs, hs and u parts)   (:259-264)
    term(P::to_affine(pr.a), minus_one);
    term(P::to_affine(pr.s), S::neg(x));
    term(crs.h, pr.mu);
    for (size_t j = 0; j < k; j++) {                                             // src/ipa/mod.rs:170-190
      term(P::to_affine(pr.ipa.L[j]), S::neg(S::mul(xs[j], xs[j])));
      term(P::to_affine(pr.ipa.R[j]), S::neg(S::mul(xinv[j], xinv[j])));
    }
    // rho * ((t_hat - delta) g + tao_x h - sum_j z^(2+j) V_j - x T1 - x^2 T2)   (:217-227, :278-279)
    Scalar rho = rng.scalar();
    term(crs.g, S::mul(rho, sub(pr.t_hat, delta)));
    term(crs.h, S::mul(rho, pr.tao_x));
    zp = zz;
    for (size_t j = 0; j < m; j++) {
      term(P::to_affine(v_commit[j]), S::neg(S::mul(rho, zp)));
      zp = S::mul(zp, z);
    }
    term(P::to_affine(pr.tt1), S::neg(S::mul(rho, x)));
    term(P::to_affine(pr.tt2), S::neg(S::mul(rho, S::mul(x, x))));
    return vp;
  }

  static M verify_aux(bp_ctx* ctx, const Crs& crs, const std::vector<Projective>& v_commit, unsigned n_bits,
                      const RangeProof& pr, Transcript& t, RandomSource& rng) {
    const VerifyPlan vp = verify_plan(crs, v_commit, n_bits, pr, t, rng);
    const size_t n = vp.n, k = vp.k;
    // scalars of gs and hs
    std::vector<Scalar> gsc(n), hsc(n);
    Scalar dummy = S::one();
    check(ctx, bp_range_agg_verify_scalars(ctx, CURVE, k ? vp.xs[0].l : dummy.l, k ? vp.xinv[0].l : dummy.l, k, pr.ipa.a.l,
                                           pr.ipa.b.l, vp.y_inv.l, vp.z.l, vp.m, n_bits, gsc[0].l, hsc[0].l),
          "bp_range_agg_verify_scalars");
    M msm(ctx, crs.gh, 2 * n + 1);
    msm.upsert_resident_range(0, gsc);
    msm.upsert_resident_range(n, hsc);
    msm.upsert_resident(2 * n, vp.u_scalar);
    for (const auto& pt : vp.terms) msm.upsert(pt.first, pt.second);
    return msm;
  }

  // The same check without the accumulator: the scalars of gs and hs never leave the device
  // (bp_range_agg_verify_msm).  Returns the group element the deferred MSM evaluates to (the value of
  // verify_aux(...).execute()): the identity iff the proof is valid.
  static Projective verify_value(bp_ctx* ctx, const Crs& crs, const std::vector<Projective>& v_commit, unsigned n_bits,
                                 const RangeProof& pr, Transcript& t, RandomSource& rng) {
    const VerifyPlan vp = verify_plan(crs, v_commit, n_bits, pr, t, rng);
    const size_t ne = vp.terms.size(), k = vp.k;
    std::vector<uint64_t> x(4 * ne + 4), y(4 * ne + 4), sc(4 * ne + 4);
    std::vector<uint8_t> inf(ne + 1);
    for (size_t i = 0; i < ne; i++) {
      memcpy(&x[4 * i], vp.terms[i].first.x, 32); memcpy(&y[4 * i], vp.terms[i].first.y, 32);
      inf[i] = vp.terms[i].first.inf;
      memcpy(&sc[4 * i], vp.terms[i].second.l, 32);
    }
    Scalar dummy = S::one();
    Projective out;
    check(ctx, bp_range_agg_verify_msm(ctx, crs.gh, k ? vp.xs[0].l : dummy.l, k ? vp.xinv[0].l : dummy.l, k, pr.ipa.a.l,
                                       pr.ipa.b.l, vp.y_inv.l, vp.z.l, vp.m, n_bits, vp.u_scalar.l, x.data(), y.data(),
                                       inf.data(), sc.data(), ne, out.xyz),
          "bp_range_agg_verify_msm");
    return out;
  }
  // A proof of the wrong shape is a rejection, as the reference's Err(VerificationError) is to its callers
  // (src/range/aggregate.rs:284-299), not a failure of the call.
  static bool verify(bp_ctx* ctx, const Crs& crs, const std::vector<Projective>& v_commit, unsigned n_bits,
                     const RangeProof& pr, Transcript& t, RandomSource& rng) {
    try {
      return is_zero(verify_value(ctx, crs, v_commit, n_bits, pr, t, rng));
    } catch (const VerificationError&) {
      return false;
    }
  }

  // verify_batch_aux (src/msm.rs:91-115) over aggregated range proofs of one CRS with nothing but the
  // handful of non-CRS terms on the host: every proof's gs | hs | u scalars are built on the device and
  // added to the batch accumulator there with the weight alpha^i (bp_batch_axpy_range_agg), the other
  // terms are scaled and merged in a host accumulator, one MSM finishes.
  struct BatchItem {
    const std::vector<Projective>* v_commit;
    const RangeProof* proof;
    Transcript* transcript;
    RandomSource* rng;
  };
  static bool verify_batch(bp_ctx* ctx, const Crs& crs, unsigned n_bits, const std::vector<BatchItem>& items,
                           const Scalar& alpha) {
    if (items.empty()) throw std::invalid_argument("non-empty vec");
    const size_t n = crs.n;
    bp_batch* batch = nullptr;
    check(ctx, bp_batch_begin(ctx, crs.gh, 2 * n + 1, &batch), "bp_batch_begin");
    struct Release { bp_batch* b; ~Release() { bp_batch_free(b); } } release{batch};
    M rest(ctx);
    Scalar pw = S::one(), dummy = S::one();
    for (const BatchItem& it : items) {
      // one malformed proof rejects the batch (verify_batch_aux takes the proofs' accumulators, and
      // verify_aux has none to give for it: src/msm.rs:91-115)
      VerifyPlan vp;
      try {
        vp = verify_plan(crs, *it.v_commit, n_bits, *it.proof, *it.transcript, *it.rng);
      } catch (const VerificationError&) {
        return false;
      }
      const size_t k = vp.k;
      check(ctx, bp_batch_axpy_range_agg(batch, pw.l, k ? vp.xs[0].l : dummy.l, k ? vp.xinv[0].l : dummy.l, k,
                                         it.proof->ipa.a.l, it.proof->ipa.b.l, vp.y_inv.l, vp.z.l, vp.m, n_bits,
                                         vp.u_scalar.l),
            "bp_batch_axpy_range_agg");
      for (const auto& pt : vp.terms) rest.upsert(pt.first, S::mul(pt.second, pw));
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

  // The same verifier in the reference's order of operations (every intermediate group element
  // computed as there); kept as the cross-check of verify_aux.
  // src/range/aggregate.rs:173-282: returns the deferred MSM that must evaluate to zero.
  // v_commit: the m value commitments V_j = v_j g + gamma_j h.
  static M verify_aux_stepwise(bp_ctx* ctx, const Crs& crs, const std::vector<Projective>& v_commit, unsigned n_bits,
                      const RangeProof& pr, Transcript& t, RandomSource& rng) {
    const size_t m = v_commit.size(), n = (size_t)n_bits * m;
    if (n != crs.n) throw std::invalid_argument("CRS registered for another size");
    checked_rounds(pr.ipa.L.size(), pr.ipa.R.size(), n);
    const Scalar one = S::one();
    Projective as[2] = {pr.a, pr.s};
    t.prover_points(as, 2);
    Scalar y = t.verifier_scalar(), z = t.verifier_scalar();
    Projective tt[2] = {pr.tt1, pr.tt2};
    t.prover_points(tt, 2);
    Scalar x = t.verifier_scalar();
    Scalar msg[3] = {pr.tao_x, pr.mu, pr.t_hat};
    t.prover_scalars(msg, 3);
    std::vector<Scalar> y_vec(n);
    check(ctx, bp_powers(ctx, CURVE, y.l, n, y_vec[0].l), "bp_powers");   // :199-202
    Scalar two = from_u64(2), zz = S::mul(z, z);
    // :209-215  delta(y,z) = (z - z^2) <1, y^N> - sum_j z^(3+j) <1, 2^n>
    Scalar sum_y, d = S::zero(), p2 = one;
    check(ctx, bp_sum(ctx, CURVE, y_vec[0].l, n, sum_y.l), "bp_sum");
    for (unsigned i = 0; i < n_bits; i++) { d = S::add(d, p2); p2 = S::mul(p2, two); }
    Scalar zsum = S::zero(), zp = S::mul(zz, z);
    for (size_t j = 0; j < m; j++) { zsum = S::add(zsum, zp); zp = S::mul(zp, z); }
    Scalar delta = sub(S::mul(sub(z, zz), sum_y), S::mul(zsum, d));
    // :217-227  g-side check value
    std::vector<uint64_t> vx(4 * m), vy(4 * m), zs(4 * m); std::vector<uint8_t> vi(m);
    zp = zz;
    for (size_t j = 0; j < m; j++) {
      Affine a = P::to_affine(v_commit[j]);
      memcpy(&vx[4 * j], a.x, 32); memcpy(&vy[4 * j], a.y, 32); vi[j] = a.inf;
      memcpy(&zs[4 * j], zp.l, 32);
      zp = S::mul(zp, z);
    }
    Projective vz;
    check(ctx, bp_msm(ctx, CURVE, vx.data(), vy.data(), vi.data(), zs.data(), m, vz.xyz), "bp_msm");  // :226
    HXyzz lhs = hx_add<Fq>(crs.g_tab.mul(sub(pr.t_hat, delta)), crs.h_tab.mul(pr.tao_x));
    HXyzz rhs = hx_add<Fq>(hx_add<Fq>(P::lift(vz), P::mul(P::lift(pr.tt1), x)), P::mul(P::lift(pr.tt2), S::mul(x, x)));
    HXyzz neg_rhs = rhs;
    neg_rhs.y = h_neg<Fq>(rhs.y);
    Projective g_check = P::lower(hx_add<Fq>(lhs, neg_rhs));
    // :229-261  P = A + x S + <-z, gs> + <z y^i + z^(2+j) 2^i, hs'> - mu h
    // :231-237 -- the reference materialises hs'_i = y^-i h_i (N scalar multiplications);
    // the same group element is reached by moving y^-i onto the scalars, so only the
    // powers are needed here
    std::vector<Scalar> y_inv_pows(n);
    Scalar y_inv = S::inv(y);
    check(ctx, bp_powers(ctx, CURVE, y_inv.l, n, y_inv_pows[0].l), "bp_powers");
    std::vector<Scalar> gsc(n, S::neg(z)), hsc(n);
    zp = zz;
    for (size_t j = 0; j < m; j++) {
      p2 = one;
      for (unsigned i = 0; i < n_bits; i++) {
        size_t k = j * n_bits + i;
        hsc[k] = S::add(S::mul(z, y_vec[k]), S::mul(zp, p2));
        p2 = S::mul(p2, two);
      }
      zp = S::mul(zp, z);
    }
    // MSM over gs || hs' : scale the hs scalars by y^-i and use the resident gs || hs
    std::vector<Scalar> hsc_scaled(n);
    check(ctx, bp_vec_op(ctx, CURVE, BP_VEC_HADAMARD, hsc[0].l, y_inv_pows[0].l, nullptr, n, hsc_scaled[0].l), "hadamard");
    Projective base = msm_gh(ctx, crs, gsc, hsc_scaled);                       // :259 (same group element)
    HXyzz p = hx_add<Fq>(hx_add<Fq>(P::lift(pr.a), P::mul(P::lift(pr.s), x)), P::lift(base));
    p = hx_add<Fq>(p, crs.h_tab.mul(S::neg(pr.mu)));                           // :264
    // ipa::extended::verify_aux over the unscaled CRS (src/ipa/extended.rs:54-70) ...
    Scalar xe = t.verifier_scalar();
    Projective p_stmt = P::lower(hx_add<Fq>(p, crs.u_tab.mul(S::mul(xe, pr.t_hat))));
    M msm = I::verify_aux(ctx, crs.ipa, p_stmt, n, pr.ipa, t);
    std::vector<Affine> uu{crs.ipa.u};
    std::vector<Scalar> xx{xe};
    msm.scale_elems(uu, xx);                                                   // scale_elem(crs.u, x)
    // ... then the h_i scalars are multiplied by y^-i (:272)
    std::vector<Affine> hs(crs.ipa.hs.begin(), crs.ipa.hs.begin() + n);
    msm.scale_elems(hs, y_inv_pows);
    msm.upsert(P::to_affine(g_check), rng.scalar());                           // :278-279
    return msm;
  }
};

// Single-value range proof (Bulletproofs section 4.1-4.2; src/range/mod.rs:26-268; the shape of
// benches/range_bench.rs, n_bits = 8..64).  Every expression of that file is the one-value case of
// the aggregated protocol, term by term: the r-polynomial's two_vec * z^2 (:108) is z^(2+j) 2^i at
// j = 0 (src/range/aggregate.rs:109-114), tao_x's z^2 * gamma (:129) is the one-term sigma_summand
// (aggregate.rs:136-139), delta's z^3 * sum(two_vec) (:198-200) is aggregate.rs:209-215 with one
// power of z, and statement.v.mul(z^2) (:203) is the one-point MSM of aggregate.rs:226.  The
// same device work therefore serves both; this type fixes m = 1 and keeps the single-value
// signatures (one value, one blinder, one commitment).
template <class Fq, class Fr, int CURVE>
struct Range {
  using A = RangeAggregate<Fq, Fr, CURVE>;
  using Crs = typename A::Crs;
  using M = typename A::M;
  // Statement::new, src/range/types.rs:62-68: V = v g + gamma h
  static Projective statement(const Crs& crs, uint64_t v, const Scalar& gamma) {
    using P = PointOps<Fq, Fr>;
    return P::lower(hx_add<Fq>(crs.g_tab.mul(A::from_u64(v)), crs.h_tab.mul(gamma)));
  }
  // range::prove, src/range/mod.rs:26-159
  static RangeProof prove(bp_ctx* ctx, const Crs& crs, uint64_t v, const Scalar& gamma, unsigned n_bits, Transcript& t,
                          RandomSource& rng) {
    return A::prove(ctx, crs, std::vector<uint64_t>{v}, std::vector<Scalar>{gamma}, n_bits, t, rng);
  }
  // range::verify_aux, src/range/mod.rs:162-249
  static M verify_aux(bp_ctx* ctx, const Crs& crs, const Projective& v_commit, unsigned n_bits, const RangeProof& pr,
                      Transcript& t, RandomSource& rng) {
    return A::verify_aux(ctx, crs, std::vector<Projective>{v_commit}, n_bits, pr, t, rng);
  }
  // range::verify, src/range/mod.rs:252-268
  static Projective verify_value(bp_ctx* ctx, const Crs& crs, const Projective& v_commit, unsigned n_bits,
                                 const RangeProof& pr, Transcript& t, RandomSource& rng) {
    return A::verify_value(ctx, crs, std::vector<Projective>{v_commit}, n_bits, pr, t, rng);
  }
  static bool verify(bp_ctx* ctx, const Crs& crs, const Projective& v_commit, unsigned n_bits, const RangeProof& pr,
                     Transcript& t, RandomSource& rng) {
    try {
      return is_zero(verify_value(ctx, crs, v_commit, n_bits, pr, t, rng));
    } catch (const VerificationError&) {
      return false;
    }
  }
};

}  // namespace host
}  // namespace bp
