// The polynomial commitment of src/poly_commit (commit, open, verify, fold of deferred checks) with
// the vectors spread over several GPUs, one rank (process or thread) and one bp_ctx per GPU.
//
// Partition: rank r of `world` (a power of two) keeps the elements i = r (mod world) of the generator
// vector gs and of every coefficient vector.  A fold pairs i with i + n/2 (src/poly_commit/mod.rs:
// 69-105), which share a residue class while the half length is a multiple of world, so every rank
// runs an ordinary round session (bp_ipa_*) over its strided slice and only a round's two points
// cross between ranks: L = sum_r L_r, R = sum_r R_r.  When the local vectors reach length one, the
// world remaining (generator, a, b) triples are gathered and every rank finishes the last
// log2(world) rounds by itself.  The commitment MSM (src/poly_commit/types.rs:141) is the sum of the
// ranks' MSMs over their slices; the verifier's <h_poly.coeffs(), gs> (src/poly_commit/mod.rs:152-159)
// is too, because on a residue class the coefficient vector is the s-vector of the first
// k - log2(world) challenges times one constant; and the fold of deferred checks
// (fold_todos_witness, :228-244) is built slice by slice where the next opening uses it.  Every
// exchange is an all-gather of at most 24 u64 per rank between HOST buffers (the results of a round
// arrive on the host anyway), through the Exchange interface: threads of one process
// (LocalExchange), or a caller's collective (CallbackExchange: e.g. a gloo or MPI all-gather).
// All ranks derive the same challenges and blinders from identically seeded transcripts.
#pragma once
#include <condition_variable>
#include <mutex>

#include "bp_host_poly.hpp"

namespace bp {
namespace host {

struct Exchange {
  size_t rank = 0, world = 1;
  // all[r * len .. (r + 1) * len) = rank r's `mine`, on every rank
  virtual void all_gather(const uint64_t* mine, size_t len, uint64_t* all) = 0;
  virtual ~Exchange() {}
};

// the ranks are threads of one process
struct LocalExchangeHub {
  explicit LocalExchangeHub(size_t world_) : world(world_) {}
  size_t world;
  std::mutex m;
  std::condition_variable cv;
  std::vector<uint64_t> buf;
  size_t arrived = 0, departed = 0;
  bool draining = false, failed = false;
  // a rank that fails wakes the others, which then fail too instead of waiting for it
  void abort() {
    std::lock_guard<std::mutex> lk(m);
    failed = true;
    cv.notify_all();
  }
};

struct LocalExchange : Exchange {
  LocalExchangeHub* hub;
  LocalExchange(LocalExchangeHub* h, size_t r) : hub(h) { rank = r; world = h->world; }
  void all_gather(const uint64_t* mine, size_t len, uint64_t* all) override {
    std::unique_lock<std::mutex> lk(hub->m);
    hub->cv.wait(lk, [&] { return !hub->draining || hub->failed; });  // the previous exchange has been read by all
    if (hub->failed) throw std::runtime_error("another rank failed");
    if (hub->buf.size() < world * len) hub->buf.resize(world * len);
    memcpy(&hub->buf[rank * len], mine, len * 8);
    if (++hub->arrived == world) {
      hub->draining = true;
      hub->cv.notify_all();
    } else {
      hub->cv.wait(lk, [&] { return hub->draining || hub->failed; });
      if (hub->failed) throw std::runtime_error("another rank failed");
    }
    memcpy(all, hub->buf.data(), world * len * 8);
    if (++hub->departed == world) {
      hub->arrived = hub->departed = 0;
      hub->draining = false;
      hub->cv.notify_all();
    }
  }
};

// the caller's collective: fn(user, mine, len, all) returns 0 on success
struct CallbackExchange : Exchange {
  int (*fn)(void*, const uint64_t*, size_t, uint64_t*) = nullptr;
  void* user = nullptr;
  void all_gather(const uint64_t* mine, size_t len, uint64_t* all) override {
    if (world == 1) { memcpy(all, mine, len * 8); return; }
    if (!fn || fn(user, mine, len, all) != 0) throw std::runtime_error("exchange callback failed");
  }
};

template <class Fq, class Fr, int CURVE>
struct ShardedPolyCommit {
  using PC = PolyCommit<Fq, Fr, CURVE>;
  using S = ScalarOps<Fr>;
  using P = PointOps<Fq, Fr>;
  using Statement = typename PC::Statement;

  struct Crs {
    size_t n = 0;               // generators over all ranks (a power of two, >= world)
    size_t rank = 0, world = 1;
    std::vector<Affine> gs;     // this rank's: gs[rank], gs[rank + world], ...
    Affine h, gen;
    bp_bases* bases = nullptr;  // gs, registered on this rank's GPU
    FixedBase<Fq, Fr> h_tab, gen_tab;
    size_t local() const { return n / world; }
    size_t log_world() const { size_t t = 0; while (((size_t)1 << t) < world) t++; return t; }
  };

  static void register_bases(bp_ctx* ctx, Crs& crs) {
    if (crs.world == 0 || (crs.world & (crs.world - 1)) || crs.n == 0 || (crs.n & (crs.n - 1)) || crs.n < crs.world ||
        crs.rank >= crs.world || crs.gs.size() != crs.local())
      throw std::invalid_argument("sharded CRS: sizes must be powers of two, world <= n, one slice of n / world generators");
    const size_t n = crs.gs.size();
    crs.h_tab.build(crs.h); crs.gen_tab.build(crs.gen);
    std::vector<uint64_t> x(4 * n), y(4 * n);
    std::vector<uint8_t> inf(n);
    for (size_t i = 0; i < n; i++) { memcpy(&x[4 * i], crs.gs[i].x, 32); memcpy(&y[4 * i], crs.gs[i].y, 32); inf[i] = crs.gs[i].inf; }
    check(ctx, bp_bases_register(ctx, CURVE, x.data(), y.data(), inf.data(), n, &crs.bases), "bp_bases_register");
  }

  static Scalar pow(const Scalar& x, size_t e) {
    Scalar acc = S::one(), base = x;
    for (; e; e >>= 1) { if (e & 1) acc = S::mul(acc, base); base = S::mul(base, base); }
    return acc;
  }

  // sum of one point per rank (normalised Jacobian, 12 u64 each)
  static HXyzz sum_points(const uint64_t* all, size_t world, size_t stride, size_t offset) {
    HXyzz acc = hx_identity<Fq>();
    for (size_t r = 0; r < world; r++) {
      Projective p;
      memcpy(p.xyz, all + r * stride + offset, 96);
      acc = hx_add<Fq>(acc, P::lift(p));
    }
    return acc;
  }

  // the constant by which rank `rank`'s residue class of coeffs(h_poly) exceeds the s-vector of the first
  // k - log2(world) challenges: prod_{j < log2 world} (bit j of rank ? u[k-1-j] : u[k-1-j]^-1)
  // (src/poly_commit/mod.rs:194-219: bit j of the index selects challenge k-1-j)
  static Scalar class_factor(const std::vector<Scalar>& ui, const std::vector<Scalar>& uinv, size_t rank, size_t log_world) {
    const size_t k = ui.size();
    Scalar f = S::one();
    for (size_t j = 0; j < log_world; j++) f = S::mul(f, (rank >> j) & 1 ? ui[k - 1 - j] : uinv[k - 1 - j]);
    return f;
  }

  // Witness::statement, src/poly_commit/types.rs:137-151; coeffs: this rank's slice, n / world scalars
  static Statement statement(bp_ctx* ctx, const Crs& crs, Exchange& ex, const std::vector<Scalar>& coeffs, const Scalar& r,
                             const Scalar& x) {
    const size_t nl = crs.local(), world = crs.world;
    if (coeffs.size() != nl) throw std::invalid_argument("coefficient slice of another length than the generator slice");
    Projective part;
    check(ctx, bp_msm_resident(ctx, crs.bases, 0, nl, coeffs[0].l, part.xyz), "bp_msm_resident");   // :141, this rank's terms
    // <coeffs, powers(x)> (:145) over the class: x^rank * <slice, powers(x^world)>
    std::vector<Scalar> pw(nl);
    const Scalar xw = pow(x, world);
    check(ctx, bp_powers(ctx, CURVE, xw.l, nl, pw[0].l), "bp_powers");
    Scalar dot;
    check(ctx, bp_inner_product(ctx, CURVE, coeffs[0].l, pw[0].l, nl, dot.l), "bp_inner_product");
    dot = S::mul(dot, pow(x, crs.rank));
    uint64_t mine[16];
    memcpy(mine, part.xyz, 96); memcpy(mine + 12, dot.l, 32);
    std::vector<uint64_t> all(16 * world);
    ex.all_gather(mine, 16, all.data());
    Statement st;
    st.x = x;
    st.commitment = P::lower(hx_add<Fq>(sum_points(all.data(), world, 16, 0), crs.h_tab.mul(r)));       // :142-143
    st.evaluation = S::zero();
    for (size_t q = 0; q < world; q++) { Scalar e; memcpy(e.l, &all[16 * q + 12], 32); st.evaluation = S::add(st.evaluation, e); }
    return st;
  }

  // one round of the loop at src/poly_commit/mod.rs:69-105 on session s; `shared`: the session holds a slice
  // and the round's two points are summed over the ranks
  static void round(bp_ctx* ctx, const Crs& crs, Exchange& ex, bp_ipa* s, bool shared, Scalar& r, Transcript& t, RandomSource& rng,
                    PolyProof& proof, Todo& todo) {
    Scalar l_j = rng.scalar(), r_j = rng.scalar();                                     // :75
    Projective lm, rm; Scalar cl, cr;
    check(ctx, bp_ipa_round_lr(s, lm.xyz, rm.xyz, cl.l, cr.l), "bp_ipa_round_lr");      // :78,80 / :83,85
    HXyzz left, right;
    if (shared && crs.world > 1) {
      uint64_t mine[24];
      memcpy(mine, lm.xyz, 96); memcpy(mine + 12, rm.xyz, 96);
      std::vector<uint64_t> all(24 * crs.world);
      ex.all_gather(mine, 24, all.data());
      left = sum_points(all.data(), crs.world, 24, 0);
      right = sum_points(all.data(), crs.world, 24, 12);
    } else {
      left = P::lift(lm); right = P::lift(rm);
    }
    left = hx_add<Fq>(left, crs.h_tab.mul(l_j));                                        // :79
    right = hx_add<Fq>(right, crs.h_tab.mul(r_j));                                      // :84
    Projective lr[2] = {P::lower(left), P::lower(right)};
    t.prover_points(lr, 2);                                                             // :89
    Scalar u_j = t.verifier_scalar();                                                   // :91
    Scalar u_inv = S::inv(u_j);
    todo.ui.push_back(u_j);
    check(ctx, bp_ipa_round_fold(s, u_j.l, u_inv.l), "bp_ipa_round_fold");              // :95-102
    r = S::add(r, S::add(S::mul(l_j, S::mul(u_j, u_j)), S::mul(r_j, S::mul(u_inv, u_inv))));  // :104
    proof.L.push_back(lr[0]); proof.R.push_back(lr[1]);
  }

  // poly_commit::prove, src/poly_commit/mod.rs:45-119; every rank returns the same proof and deferred check
  static Todo prove(bp_ctx* ctx, const Crs& crs, Exchange& ex, const Statement& st, const std::vector<Scalar>& coeffs, Scalar r,
                    Transcript& t, RandomSource& rng, PolyProof& proof) {
    const size_t nl = crs.local(), world = crs.world;
    if (coeffs.size() != nl) throw std::invalid_argument("coefficient slice of another length than the generator slice");
    const Scalar uch = t.verifier_scalar();
    const Affine u = P::to_affine(P::lower(crs.gen_tab.mul(uch)));                      // :55-58
    // b = powers_of_x(x) (:65) on the class is x^rank * powers(x^world): the session builds the powers
    // of x^world and the constant rides on u (every use of b in a round is the inner product that
    // multiplies u, :80,85), u_r = x^rank * u
    const Scalar xr = pow(st.x, crs.rank), xw = pow(st.x, world);
    const Affine u_r = P::to_affine(P::lower(crs.gen_tab.mul(S::mul(uch, xr))));
    bp_ipa* s = nullptr;
    check(ctx, bp_ipa_begin_resident_powers(ctx, crs.bases, BP_IPA_POLYCOMMIT, 0, 0, coeffs[0].l, xw.l, nl, &s),
          "bp_ipa_begin_resident_powers");
    Todo todo;
    auto set_u = [&](bp_ipa* session, const Affine& pt) {
      if (pt.inf) return;  // a zero challenge: the u terms vanish
      uint64_t uxy[8];
      memcpy(uxy, pt.x, 32); memcpy(uxy + 4, pt.y, 32);
      check(ctx, bp_ipa_set_u(session, uxy), "bp_ipa_set_u");
    };
    try {
      set_u(s, u_r);
      while (bp_ipa_len(s) != 1) round(ctx, crs, ex, s, true, r, t, rng, proof, todo);
      if (world > 1) {
        // the world elements that are left, in index order = rank order
        uint64_t mine[17];
        Affine g; Scalar a, b; uint8_t ginf = 0;
        check(ctx, bp_ipa_read(s, g.x, g.y, &ginf, nullptr, nullptr, nullptr, a.l, b.l), "bp_ipa_read");
        b = S::mul(b, xr);
        memcpy(mine, g.x, 32); memcpy(mine + 4, g.y, 32); mine[8] = ginf; memcpy(mine + 9, a.l, 32); memcpy(mine + 13, b.l, 32);
        std::vector<uint64_t> all(17 * world);
        ex.all_gather(mine, 17, all.data());
        std::vector<uint64_t> gx(4 * world), gy(4 * world), ta(4 * world), tb(4 * world);
        std::vector<uint8_t> gi(world);
        for (size_t q = 0; q < world; q++) {
          const uint64_t* p = &all[17 * q];
          memcpy(&gx[4 * q], p, 32); memcpy(&gy[4 * q], p + 4, 32); gi[q] = (uint8_t)p[8];
          memcpy(&ta[4 * q], p + 9, 32); memcpy(&tb[4 * q], p + 13, 32);
        }
        bp_ipa* tail = nullptr;
        check(ctx, bp_ipa_begin(ctx, CURVE, BP_IPA_POLYCOMMIT, gx.data(), gy.data(), gi.data(), nullptr, nullptr, nullptr, ta.data(),
                                tb.data(), world, &tail), "bp_ipa_begin");
        bp_ipa_free(s);
        s = tail;
        set_u(s, u);
        while (bp_ipa_len(s) != 1) round(ctx, crs, ex, s, false, r, t, rng, proof, todo);
      }
      // a[0] and the folded generator, which is <h_poly.coeffs(), gs> (the value :108-113 asserts)
      uint8_t ginf = 0;
      check(ctx, bp_ipa_read(s, todo.g.x, todo.g.y, &ginf, nullptr, nullptr, nullptr, proof.a.l, nullptr), "bp_ipa_read");
      todo.g.inf = ginf;
    } catch (...) { bp_ipa_free(s); throw; }
    bp_ipa_free(s);
    proof.r = r;
    Scalar msg[2] = {proof.a, proof.r};
    t.prover_scalars(msg, 2);                                                           // :107
    return todo;
  }

  static void replay(const Crs& crs, const PolyProof& proof, Transcript& t, Affine& u, std::vector<Scalar>& ui) {
    checked_rounds(proof.L.size(), proof.R.size(), crs.n);                              // :129-145, :274-290
    u = P::to_affine(P::lower(crs.gen_tab.mul(t.verifier_scalar())));                   // :132-135
    for (size_t j = 0; j < proof.L.size(); j++) {
      Projective lr[2] = {proof.L[j], proof.R[j]};
      t.prover_points(lr, 2);
      ui.push_back(t.verifier_scalar());                                                // :139-145
    }
    Scalar msg[2] = {proof.a, proof.r};
    t.prover_scalars(msg, 2);                                                           // :165
  }

  // <h_poly.coeffs(), gs> (:152-159) as the sum of the ranks' shares
  static Affine folded_generator(bp_ctx* ctx, const Crs& crs, Exchange& ex, const std::vector<Scalar>& ui) {
    const size_t k = ui.size(), tw = crs.log_world(), kl = k - tw;
    std::vector<Scalar> uinv;
    S::batch_inv(ui, uinv);
    Projective part;
    if (kl == 0) {
      part = P::lower(P::lift(crs.gs[0]));
    } else {
      check(ctx, bp_msm_resident_s_vector(ctx, crs.bases, 0, ui[0].l, uinv[0].l, kl, part.xyz), "bp_msm_resident_s_vector");
    }
    Projective scaled = P::lower(P::mul(P::lift(part), class_factor(ui, uinv, crs.rank, tw)));
    std::vector<uint64_t> all(12 * crs.world);
    ex.all_gather(scaled.xyz, 12, all.data());
    return P::to_affine(P::lower(sum_points(all.data(), crs.world, 12, 0)));
  }

  static typename PC::Crs plain_crs(const Crs& crs) {
    typename PC::Crs c;
    c.h = crs.h; c.gen = crs.gen;
    return c;
  }

  // poly_commit::verify, src/poly_commit/mod.rs:121-173
  static bool verify(bp_ctx* ctx, const Crs& crs, Exchange& ex, const Statement& st, const PolyProof& proof, Transcript& t) {
    Affine u; std::vector<Scalar> ui;
    // every rank reaches the same verdict on the proof's shape, so all or none enter the exchange
    try { replay(crs, proof, t, u, ui); } catch (const VerificationError&) { return false; }
    const Affine g = folded_generator(ctx, crs, ex, ui);
    const Scalar b = PC::hpoly_evaluate(ui, st.x);                                      // :160-163
    return PC::final_check(ctx, plain_crs(crs), st, proof, u, ui, g, b);
  }

  // poly_commit::lazy_verify, :264-315: O(log n) besides one small MSM, the same on every rank
  static bool lazy_verify(bp_ctx* ctx, const Crs& crs, const Statement& st, const Affine& assumption, const PolyProof& proof,
                          Transcript& t, std::vector<Todo>& todos) {
    Affine u; std::vector<Scalar> ui;
    try { replay(crs, proof, t, u, ui); } catch (const VerificationError&) { return false; }
    const Scalar b = PC::hpoly_evaluate(ui, st.x);
    if (!PC::final_check(ctx, plain_crs(crs), st, proof, u, ui, assumption, b)) return false;
    todos.push_back(Todo{assumption, ui});
    return true;
  }

  // this rank's slice of fold_todos_witness (:228-244): sum_i alpha^i * coeffs(h_poly_i) on the class, built
  // where the opening of the folded statement reads it; nothing crosses between ranks
  static std::vector<Scalar> fold_todos_witness(bp_ctx* ctx, const Crs& crs, const std::vector<Todo>& todos, const Scalar& alpha) {
    if (todos.empty()) throw std::invalid_argument("Non empty statement list");
    const size_t k = todos[0].ui.size(), tw = crs.log_world(), kl = k - tw, m = todos.size();
    if (((size_t)1 << k) != crs.n) throw std::invalid_argument("deferred checks of another size than the CRS");
    std::vector<Scalar> ch(m * (kl ? kl : 1)), chi(m * (kl ? kl : 1)), w(m);
    Scalar ai = S::one(), total = S::zero();
    for (size_t i = 0; i < m; i++) {
      if (todos[i].ui.size() != k) throw std::invalid_argument("deferred checks of different sizes");
      std::vector<Scalar> uinv;
      S::batch_inv(todos[i].ui, uinv);
      for (size_t j = 0; j < kl; j++) { ch[i * kl + j] = todos[i].ui[j]; chi[i * kl + j] = uinv[j]; }
      w[i] = S::mul(ai, class_factor(todos[i].ui, uinv, crs.rank, tw));
      total = S::add(total, w[i]);
      ai = S::mul(ai, alpha);
    }
    if (kl == 0) return std::vector<Scalar>{total};  // one element per rank: the s-vector of no challenges is (1)
    std::vector<Scalar> acc((size_t)1 << kl);
    check(ctx, bp_fold_s_vectors(ctx, CURVE, ch[0].l, chi[0].l, w[0].l, kl, m, acc[0].l), "bp_fold_s_vectors");
    return acc;
  }

  // fold_todos_statement (:246-262): a sum over the deferred checks, the same on every rank
  static Statement fold_todos_statement(bp_ctx* ctx, const std::vector<Todo>& todos, const Scalar& alpha, const Scalar& x) {
    return PC::fold_todos_statement(ctx, todos, alpha, x);
  }
};

}  // namespace host
}  // namespace bp
