// C entry points over the C++ host mirror (bp_host.hpp) so that the Python parity
// tests can drive the Msm accumulator and the IPA prover/verifier loops.
#include "bp_host.hpp"

using namespace bp;
using namespace bp::host;

// A proof of the wrong shape that reached a driver which returns group elements rather than a verdict
// (the verifiers that answer yes or no report it as a rejection, ok = 0 with BP_OK): distinct from the
// device and argument errors of include/bp_gpu.h.
#define BPH_E_VERIFY (-16)

namespace {

// Transcript stand-in: challenge number j of a run is SplitMix64(seed + j), four
// outputs as little-endian limbs, rejected while >= r.  Prover messages are counted,
// not hashed (Fiat-Shamir itself is outside this library's scope).
template <class Fr>
struct SeededTranscript : Transcript {
  uint64_t seed, counter = 0, absorbed = 0;
  explicit SeededTranscript(uint64_t s) : seed(s) {}
  void prover_points(const Projective*, size_t n) override { absorbed += n; }
  void prover_scalars(const Scalar*, size_t n) override { absorbed += n; }
  Scalar verifier_scalar() override {
    uint64_t st = seed + counter++;
    auto next = [&]() {
      st += 0x9E3779B97F4A7C15ull;
      uint64_t z = st;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      return z ^ (z >> 31);
    };
    for (;;) {
      HFe v;
      for (int k = 0; k < 4; k++) v.v[k] = next();
      if (!h_geq_mod<Fr>(v)) {
        Scalar c;
        memcpy(c.l, v.v, 32);
        return ScalarOps<Fr>::to_mont(c);
      }
    }
  }
};

std::vector<Affine> unpack(const uint64_t* x, const uint64_t* y, const uint8_t* inf, size_t n) {
  std::vector<Affine> v(n);
  for (size_t i = 0; i < n; i++) {
    memcpy(v[i].x, x + 4 * i, 32);
    memcpy(v[i].y, y + 4 * i, 32);
    v[i].inf = inf ? inf[i] : 0;
  }
  return v;
}
std::vector<Scalar> unpack_s(const uint64_t* s, size_t n) {
  std::vector<Scalar> v(n);
  for (size_t i = 0; i < n; i++) memcpy(v[i].l, s + 4 * i, 32);
  return v;
}

template <class Fq, class Fr, int CURVE>
int msm_accumulate(bp_ctx* ctx, const uint64_t* x, const uint64_t* y, const uint8_t* inf, const uint64_t* s, size_t n,
                   uint64_t out_xyz[12], size_t* distinct) {
  Msm<Fq, Fr, CURVE> m(ctx);
  m.upsert_batch(unpack(x, y, inf, n), unpack_s(s, n));
  if (distinct) *distinct = m.size();
  Projective r = m.execute();
  memcpy(out_xyz, r.xyz, 96);
  return 0;
}

template <class Fq, class Fr, int CURVE>
int ipa_roundtrip(bp_ctx* ctx, const uint64_t* gx, const uint64_t* gy, const uint64_t* hx, const uint64_t* hy,
                  const uint64_t u_xy[8], const uint64_t* a, const uint64_t* b, size_t n, size_t n_proofs, uint64_t seed,
                  int tamper, uint64_t* out_l, uint64_t* out_r, uint64_t* out_ab, uint64_t* out_p, int* single_ok,
                  int* batch_ok) {
  using I = Ipa<Fq, Fr, CURVE>;
  typename I::Crs crs;
  crs.gs = unpack(gx, gy, nullptr, n);
  crs.hs = unpack(hx, hy, nullptr, n);
  memcpy(crs.u.x, u_xy, 32);
  memcpy(crs.u.y, u_xy + 4, 32);
  crs.u.inf = 0;
  size_t k = 0;
  while (((size_t)1 << k) < n) k++;
  // the CRS once on the GPU as gs || hs || u: proofs after the first and every verifier use it
  bp_bases* gh = nullptr;
  {
    std::vector<uint64_t> x(8 * n + 4), y(8 * n + 4);
    std::vector<uint8_t> inf(2 * n + 1, 0);
    memcpy(&x[0], gx, 32 * n); memcpy(&y[0], gy, 32 * n);
    memcpy(&x[4 * n], hx, 32 * n); memcpy(&y[4 * n], hy, 32 * n);
    memcpy(&x[8 * n], u_xy, 32); memcpy(&y[8 * n], u_xy + 4, 32);
    check(ctx, bp_bases_register(ctx, CURVE, x.data(), y.data(), inf.data(), 2 * n + 1, &gh), "bp_bases_register");
  }
  struct Release { bp_bases* b; ~Release() { bp_bases_free(b); } } release{gh};
  std::vector<Msm<Fq, Fr, CURVE>> msms;
  std::vector<Projective> batch_p;
  std::vector<IpaProof> batch_proofs;
  *single_ok = 1;
  for (size_t pi = 0; pi < n_proofs; pi++) {
    auto wa = unpack_s(a + 4 * n * pi, n), wb = unpack_s(b + 4 * n * pi, n);
    Projective p = I::statement(ctx, crs, wa, wb);
    SeededTranscript<Fr> tp(seed + 1000 * pi);
    // first proof: generators handed over with the call, as ipa::prove does; the others: resident CRS
    IpaProof proof = pi == 0 ? I::prove(ctx, crs, wa, wb, tp)
                             : I::prove_resident(ctx, gh, 0, n, nullptr, 2 * n, ScalarOps<Fr>::one(), wa, wb, tp);
    if (pi == 0) {
      for (size_t j = 0; j < k; j++) {
        memcpy(out_l + 12 * j, proof.L[j].xyz, 96);
        memcpy(out_r + 12 * j, proof.R[j].xyz, 96);
      }
      memcpy(out_ab, proof.a.l, 32);
      memcpy(out_ab + 4, proof.b.l, 32);
      memcpy(out_p, p.xyz, 96);
    }
    if (tamper == 1 && pi == n_proofs - 1) proof.a = ScalarOps<Fr>::add(proof.a, ScalarOps<Fr>::one());
    if (tamper == 2 && pi == n_proofs - 1 && !proof.L.empty()) { proof.L.pop_back(); proof.R.pop_back(); }   // a round dropped
    if (tamper == 3 && pi == n_proofs - 1) { proof.L.push_back(p); proof.R.push_back(p); }                   // a round appended
    if (tamper == 4 && pi == n_proofs - 1) proof.R.push_back(p);                                             // ragged
    SeededTranscript<Fr> tv(seed + 1000 * pi);
    if (tamper >= 2 && pi == n_proofs - 1) {
      // a proof of the wrong shape is refused before any device work, by both accumulators
      int refused = 0;
      try { I::verify_aux_resident(ctx, gh, 2 * n + 1, 0, n, 2 * n, p, n, proof, tv); } catch (const VerificationError&) { refused++; }
      SeededTranscript<Fr> tw(seed + 1000 * pi);
      try { I::verify_aux(ctx, crs, p, n, proof, tw); } catch (const VerificationError&) { refused++; }
      if (refused != 2) throw std::runtime_error("malformed proof not refused");
      // ... and is a plain rejection for the verifiers that answer yes or no, alone or in a batch
      SeededTranscript<Fr> ty(seed + 1000 * pi);
      if (I::verify_resident(ctx, gh, 0, n, 2 * n, p, n, proof, ty)) throw std::runtime_error("malformed proof accepted");
      batch_p.push_back(p);
      batch_proofs.push_back(proof);
      std::vector<SeededTranscript<Fr>> ts;
      ts.reserve(batch_p.size());
      std::vector<typename I::BatchItem> items(batch_p.size());
      for (size_t q = 0; q < batch_p.size(); q++) ts.emplace_back(seed + 1000 * q);
      for (size_t q = 0; q < batch_p.size(); q++) items[q] = typename I::BatchItem{&batch_p[q], &batch_proofs[q], &ts[q]};
      SeededTranscript<Fr> tb(seed ^ 0xA1FAull);
      if (I::verify_batch_resident(ctx, gh, 2 * n + 1, 0, n, 2 * n, n, items, tb.verifier_scalar()))
        throw std::runtime_error("batch with a malformed proof accepted");
      *single_ok = 0;
      *batch_ok = 0;
      return 0;
    }
    auto m = I::verify_aux_resident(ctx, gh, 2 * n + 1, 0, n, 2 * n, p, n, proof, tv);
    // ipa::verify, src/ipa/mod.rs:196-212
    Projective value = m.execute();
    if (!is_zero(value)) *single_ok = 0;
    if (pi == 0 || pi == n_proofs - 1) {
      // the accumulator keyed by points (verify_aux) evaluates to the same group element
      SeededTranscript<Fr> tw(seed + 1000 * pi);
      if (memcmp(I::verify_aux(ctx, crs, p, n, proof, tw).execute().xyz, value.xyz, 96) != 0)
        throw std::runtime_error("indexed and keyed accumulators disagree");
      // and so does the verifier whose scalars never leave the device (bp_ipa_verify_msm)
      SeededTranscript<Fr> tx(seed + 1000 * pi);
      if (memcmp(I::verify_value_resident(ctx, gh, 0, n, 2 * n, p, n, proof, tx).xyz, value.xyz, 96) != 0)
        throw std::runtime_error("verifier with device-resident scalars and accumulator verifier disagree");
    }
    msms.push_back(std::move(m));
    batch_p.push_back(p);
    batch_proofs.push_back(proof);
  }
  SeededTranscript<Fr> ta(seed ^ 0xA1FAull);
  const Scalar alpha = ta.verifier_scalar();
  // the batch with the CRS part of every accumulator built and added on the device reaches the same verdict
  {
    std::vector<SeededTranscript<Fr>> ts;
    ts.reserve(n_proofs);
    std::vector<typename I::BatchItem> items(n_proofs);
    for (size_t pi = 0; pi < n_proofs; pi++) ts.emplace_back(seed + 1000 * pi);
    for (size_t pi = 0; pi < n_proofs; pi++) items[pi] = typename I::BatchItem{&batch_p[pi], &batch_proofs[pi], &ts[pi]};
    const bool direct = I::verify_batch_resident(ctx, gh, 2 * n + 1, 0, n, 2 * n, n, items, alpha);
    *batch_ok = verify_batch_aux(msms, alpha) ? 1 : 0;
    if (direct != (*batch_ok != 0)) throw std::runtime_error("device-side batch and accumulator batch disagree");
  }
  return 0;
}

}  // namespace

extern "C" {

int bph_msm_accumulate(bp_ctx* ctx, int curve, const uint64_t* x, const uint64_t* y, const uint8_t* inf,
                       const uint64_t* s, size_t n, uint64_t out_xyz[12], size_t* distinct) {
  try {
    if (curve == BP_SECP256K1) return msm_accumulate<Secp256k1Fq, Secp256k1Fr, BP_SECP256K1>(ctx, x, y, inf, s, n, out_xyz, distinct);
    if (curve == BP_VESTA) return msm_accumulate<VestaFq, VestaFr, BP_VESTA>(ctx, x, y, inf, s, n, out_xyz, distinct);
    return BP_E_CURVE;
  } catch (const VerificationError&) {
    return BPH_E_VERIFY;
  } catch (const std::invalid_argument&) {
    return BP_E_ARG;
  } catch (const std::exception&) {
    return BP_E_HIP;
  }
}

// Proves `n_proofs` IPA statements over one CRS (witness p at a + 4*n*p), verifies each
// (single_ok) and all together with verify_batch_aux (batch_ok).  tamper != 0 corrupts
// the last proof's `a`.  Outputs of proof 0: L/R (k*12 u64 each), a,b (8 u64), P (12).
int bph_ipa_roundtrip(bp_ctx* ctx, int curve, const uint64_t* gx, const uint64_t* gy, const uint64_t* hx,
                      const uint64_t* hy, const uint64_t u_xy[8], const uint64_t* a, const uint64_t* b, size_t n,
                      size_t n_proofs, uint64_t seed, int tamper, uint64_t* out_l, uint64_t* out_r, uint64_t* out_ab,
                      uint64_t* out_p, int* single_ok, int* batch_ok) {
  try {
    if (curve == BP_SECP256K1)
      return ipa_roundtrip<Secp256k1Fq, Secp256k1Fr, BP_SECP256K1>(ctx, gx, gy, hx, hy, u_xy, a, b, n, n_proofs, seed, tamper,
                                                                  out_l, out_r, out_ab, out_p, single_ok, batch_ok);
    if (curve == BP_VESTA)
      return ipa_roundtrip<VestaFq, VestaFr, BP_VESTA>(ctx, gx, gy, hx, hy, u_xy, a, b, n, n_proofs, seed, tamper, out_l,
                                                      out_r, out_ab, out_p, single_ok, batch_ok);
    return BP_E_CURVE;
  } catch (const VerificationError&) {
    return BPH_E_VERIFY;
  } catch (const std::invalid_argument&) {
    return BP_E_ARG;
  } catch (const std::exception&) {
    return BP_E_HIP;
  }
}

}  // extern "C"

// ---- aggregated range proof round trip ------------------------------------------------
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "bp_host_range.hpp"

namespace {

template <class Fr>
struct SeededRandom : RandomSource {
  uint64_t st;
  explicit SeededRandom(uint64_t seed) : st(seed) {}
  uint64_t next() {
    st += 0x9E3779B97F4A7C15ull;
    uint64_t z = st;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  Scalar scalar() override {
    for (;;) {
      HFe v;
      for (int k = 0; k < 4; k++) v.v[k] = next();
      if (!h_geq_mod<Fr>(v)) {
        Scalar c;
        memcpy(c.l, v.v, 32);
        return ScalarOps<Fr>::to_mont(c);
      }
    }
  }
  bool splitmix_state(uint64_t* state) const override { *state = st; return true; }
  void splitmix_skip(size_t draws) override { st += 4 * (uint64_t)draws * 0x9E3779B97F4A7C15ull; }
};

// The same stream without the device hand-over: the batched prover then uploads the blinders
template <class Fr>
struct SeededRandomHostOnly : SeededRandom<Fr> {
  using SeededRandom<Fr>::SeededRandom;
  bool splitmix_state(uint64_t*) const override { return false; }
};

// `batch` aggregated range proofs through RangeAggregate::prove_batch, `reps` times from one host thread;
// proof b proves values[b], gamma[b] with the random stream seed_rand + b and the challenges of seed_chal + b.
template <class Fq, class Fr, int CURVE>
int range_prove_batch(bp_ctx* ctx, const uint64_t* gx, const uint64_t* gy, const uint64_t* hx, const uint64_t* hy,
                      const uint64_t* ugh_xy, const uint64_t* values, const uint64_t* gamma, size_t batch, size_t m,
                      unsigned n_bits, uint64_t seed_rand, uint64_t seed_chal, int reps, int host_blinders, int verify,
                      uint64_t* out_points, uint64_t* out_scalars, uint64_t* out_lr, double* batch_ms, int* ok) {
  using R = RangeAggregate<Fq, Fr, CURVE>;
  using PO = PointOps<Fq, Fr>;
  const size_t n = (size_t)n_bits * m;
  typename R::Crs crs;
  crs.ipa.gs = unpack(gx, gy, nullptr, n);
  crs.ipa.hs = unpack(hx, hy, nullptr, n);
  auto pt = [&](int k) { Affine a; memcpy(a.x, ugh_xy + 8 * k, 32); memcpy(a.y, ugh_xy + 8 * k + 4, 32); a.inf = 0; return a; };
  crs.ipa.u = pt(0); crs.g = pt(1); crs.h = pt(2);
  crs.n = n;
  struct Release {
    typename R::Crs& c; bp_range_batch* rb = nullptr;
    ~Release() { bp_range_batch_free(rb); bp_bases_free(c.ghb); bp_bases_free(c.gh); }
  } release{crs};
  R::register_batch_bases(ctx, crs);
  check(ctx, bp_range_batch_begin(ctx, crs.ghb, batch, m, n_bits, &release.rb), "bp_range_batch_begin");
  struct Timing { bp_ctx* c; Timing(bp_ctx* c_) : c(c_) { bp_set_timing(c, 0); } ~Timing() { bp_set_timing(c, 1); } } timing(ctx);
  const std::vector<Scalar> gam = unpack_s(gamma, batch * m);
  std::vector<RangeProof> proofs;
  double total = 0;
  for (int r = 0; r < reps + 1; r++) {  // the first pass grows the context's workspace and is not timed
    std::vector<SeededTranscript<Fr>> tr;
    std::vector<SeededRandom<Fr>> rd;
    std::vector<SeededRandomHostOnly<Fr>> rh;
    tr.reserve(batch); rd.reserve(batch); rh.reserve(batch);
    std::vector<Transcript*> tp(batch);
    std::vector<RandomSource*> rp(batch);
    for (size_t b = 0; b < batch; b++) {
      tr.emplace_back(seed_chal + b);
      rd.emplace_back(seed_rand + b);
      rh.emplace_back(seed_rand + b);
      tp[b] = &tr[b];
      rp[b] = host_blinders ? (RandomSource*)&rh[b] : (RandomSource*)&rd[b];
    }
    auto t0 = std::chrono::steady_clock::now();
    proofs = R::prove_batch(ctx, crs, release.rb, values, gam.data(), m, n_bits, tp.data(), rp.data());
    auto t1 = std::chrono::steady_clock::now();
    if (r > 0) total += std::chrono::duration<double, std::milli>(t1 - t0).count();
  }
  *batch_ms = total / reps;
  size_t k = 0;
  while (((size_t)1 << k) < n) k++;
  for (size_t b = 0; b < batch; b++) {
    const RangeProof& p = proofs[b];
    uint64_t* op = out_points + 48 * b;
    memcpy(op, p.a.xyz, 96); memcpy(op + 12, p.s.xyz, 96); memcpy(op + 24, p.tt1.xyz, 96); memcpy(op + 36, p.tt2.xyz, 96);
    uint64_t* os = out_scalars + 20 * b;
    memcpy(os, p.tao_x.l, 32); memcpy(os + 4, p.mu.l, 32); memcpy(os + 8, p.t_hat.l, 32);
    memcpy(os + 12, p.ipa.a.l, 32); memcpy(os + 16, p.ipa.b.l, 32);
    for (size_t j = 0; j < p.ipa.L.size(); j++) {
      memcpy(out_lr + (b * k + j) * 24, p.ipa.L[j].xyz, 96);
      memcpy(out_lr + (b * k + j) * 24 + 12, p.ipa.R[j].xyz, 96);
    }
  }
  *ok = 1;
  if (verify) {
    // every proof through the single-proof verifier over the usual base set
    R::register_bases(ctx, crs, n);
    for (size_t b = 0; b < batch; b++) {
      std::vector<Projective> vcom(m);
      for (size_t j = 0; j < m; j++)
        vcom[j] = PO::lower(hx_add<Fq>(crs.g_tab.mul(R::from_u64(values[b * m + j])), crs.h_tab.mul(gam[b * m + j])));
      SeededTranscript<Fr> t2(seed_chal + b);
      SeededRandom<Fr> vrng((seed_rand + b) ^ 0x5EEDull);
      if (!R::verify(ctx, crs, vcom, n_bits, proofs[b], t2, vrng)) *ok = 0;
    }
  }
  return 0;
}

template <class Fq, class Fr, int CURVE>
int range_roundtrip(bp_ctx* ctx, const uint64_t* gx, const uint64_t* gy, const uint64_t* hx, const uint64_t* hy,
                    const uint64_t* ugh_xy, const uint64_t* values, const uint64_t* gamma, size_t m, unsigned n_bits,
                    uint64_t seed_rand, uint64_t seed_chal, int tamper, int reps, int single, uint64_t* out_points,
                    uint64_t* out_scalars, uint64_t* out_lr, uint64_t* out_value, double* prove_ms, double* verify_ms,
                    int* ok) {
  using R = RangeAggregate<Fq, Fr, CURVE>;
  using R1 = Range<Fq, Fr, CURVE>;
  if (single && m != 1) return BP_E_ARG;
  using PO = PointOps<Fq, Fr>;
  size_t n = (size_t)n_bits * m;
  typename R::Crs crs;
  crs.ipa.gs = unpack(gx, gy, nullptr, n);
  crs.ipa.hs = unpack(hx, hy, nullptr, n);
  auto pt = [&](int k) { Affine a; memcpy(a.x, ugh_xy + 8 * k, 32); memcpy(a.y, ugh_xy + 8 * k + 4, 32); a.inf = 0; return a; };
  crs.ipa.u = pt(0); crs.g = pt(1); crs.h = pt(2);
  R::register_bases(ctx, crs, n);
  std::vector<uint64_t> vals(values, values + m);
  std::vector<Scalar> gam = unpack_s(gamma, m);
  // statement: V_j = v_j g + gamma_j h  (src/range/types.rs:167-176)
  std::vector<Projective> vcom(m);
  for (size_t j = 0; j < m; j++)
    vcom[j] = PO::lower(hx_add<Fq>(PO::mul(PO::lift(crs.g), R::from_u64(vals[j])), PO::mul(PO::lift(crs.h), gam[j])));
  if (single && memcmp(R1::statement(crs, vals[0], gam[0]).xyz, vcom[0].xyz, 96) != 0) { bp_bases_free(crs.gh); return BP_E_STATE; }
  RangeProof proof;
  Projective value{};
  double tp = 0, tv = 0;
  *ok = 1;
  // a proof is a chain of small launches: no event markers around its MSMs
  struct Timing { bp_ctx* c; Timing(bp_ctx* c_) : c(c_) { bp_set_timing(c, 0); } ~Timing() { bp_set_timing(c, 1); } } timing(ctx);
  try {
    for (int r = 0; r < reps; r++) {
      SeededTranscript<Fr> t1(seed_chal);
      SeededRandom<Fr> rng(seed_rand);
      auto a = std::chrono::steady_clock::now();
      proof = single ? R1::prove(ctx, crs, vals[0], gam[0], n_bits, t1, rng) : R::prove(ctx, crs, vals, gam, n_bits, t1, rng);
      auto b = std::chrono::steady_clock::now();
      RangeProof check_proof = proof;
      if (tamper == 1) check_proof.t_hat = ScalarOps<Fr>::add(check_proof.t_hat, ScalarOps<Fr>::one());
      if (tamper == 2 && !check_proof.ipa.L.empty()) { check_proof.ipa.L.pop_back(); check_proof.ipa.R.pop_back(); }
      if (tamper == 3) { check_proof.ipa.L.push_back(proof.a); check_proof.ipa.R.push_back(proof.s); }
      SeededTranscript<Fr> t2(seed_chal);
      SeededRandom<Fr> vrng(seed_rand ^ 0x5EEDull);
      if (tamper >= 2) {
        // a proof of the wrong shape is refused before any device work, by both verifiers
        int refused = 0;
        try { R::verify_aux(ctx, crs, vcom, n_bits, check_proof, t2, vrng); } catch (const VerificationError&) { refused++; }
        SeededTranscript<Fr> t3(seed_chal);
        try { R::verify_aux_stepwise(ctx, crs, vcom, n_bits, check_proof, t3, vrng); } catch (const VerificationError&) { refused++; }
        if (refused != 2) throw std::runtime_error("malformed proof not refused");
        // ... and is a plain rejection for the verifiers that answer yes or no, alone or in a batch
        SeededTranscript<Fr> t5(seed_chal);
        SeededRandom<Fr> vrng5(seed_rand ^ 0x5EEDull);
        const bool accepted = single ? R1::verify(ctx, crs, vcom[0], n_bits, check_proof, t5, vrng5)
                                     : R::verify(ctx, crs, vcom, n_bits, check_proof, t5, vrng5);
        if (accepted) throw std::runtime_error("malformed proof accepted");
        {
          SeededTranscript<Fr> tg(seed_chal), tbad(seed_chal), ta(seed_chal ^ 0xA1FAull);
          SeededRandom<Fr> rg(seed_rand ^ 0x5EEDull), rbad(seed_rand ^ 0x5EEDull);
          std::vector<typename R::BatchItem> items{{&vcom, &proof, &tg, &rg}, {&vcom, &check_proof, &tbad, &rbad}};
          if (R::verify_batch(ctx, crs, n_bits, items, ta.verifier_scalar()))
            throw std::runtime_error("batch with a malformed proof accepted");
        }
        *ok = 0;
        continue;
      }
      auto c = std::chrono::steady_clock::now();
      // the verifier with the scalars of gs and hs kept on the device (bp_range_agg_verify_msm)
      value = single ? R1::verify_value(ctx, crs, vcom[0], n_bits, check_proof, t2, vrng)
                     : R::verify_value(ctx, crs, vcom, n_bits, check_proof, t2, vrng);
      bool good = is_zero(value);  // src/range/aggregate.rs:294-299, src/range/mod.rs:261-267
      auto d = std::chrono::steady_clock::now();
      if (!good) *ok = 0;
      if (r == 0) {
        // the accumulator form (scalars through the host, as batch verification uses it) and the verifier
        // in the reference's order of operations reach the same group element (the identity for a
        // valid proof, some other point for a corrupted one)
        SeededTranscript<Fr> t4(seed_chal);
        SeededRandom<Fr> vrng4(seed_rand ^ 0x5EEDull);
        auto msm = single ? R1::verify_aux(ctx, crs, vcom[0], n_bits, check_proof, t4, vrng4)
                          : R::verify_aux(ctx, crs, vcom, n_bits, check_proof, t4, vrng4);
        if (memcmp(msm.execute().xyz, value.xyz, 96) != 0)
          throw std::runtime_error("verifier with device-resident scalars and accumulator verifier disagree");
        SeededTranscript<Fr> t3(seed_chal);
        SeededRandom<Fr> vrng3(seed_rand ^ 0x5EEDull);
        Projective stepwise = R::verify_aux_stepwise(ctx, crs, vcom, n_bits, check_proof, t3, vrng3).execute();
        if (memcmp(stepwise.xyz, value.xyz, 96) != 0)
          throw std::runtime_error("single-MSM verifier and stepwise verifier disagree");
      }
      tp += std::chrono::duration<double, std::milli>(b - a).count();
      tv += std::chrono::duration<double, std::milli>(d - c).count();
    }
  } catch (...) {
    bp_bases_free(crs.gh);
    throw;
  }
  bp_bases_free(crs.gh);
  *prove_ms = tp / reps;
  *verify_ms = tv / reps;
  if (out_value) memcpy(out_value, value.xyz, 96);
  memcpy(out_points, proof.a.xyz, 96); memcpy(out_points + 12, proof.s.xyz, 96);
  memcpy(out_points + 24, proof.tt1.xyz, 96); memcpy(out_points + 36, proof.tt2.xyz, 96);
  memcpy(out_scalars, proof.tao_x.l, 32); memcpy(out_scalars + 4, proof.mu.l, 32); memcpy(out_scalars + 8, proof.t_hat.l, 32);
  memcpy(out_scalars + 12, proof.ipa.a.l, 32); memcpy(out_scalars + 16, proof.ipa.b.l, 32);
  for (size_t j = 0; j < proof.ipa.L.size(); j++) {
    memcpy(out_lr + 24 * j, proof.ipa.L[j].xyz, 96);
    memcpy(out_lr + 24 * j + 12, proof.ipa.R[j].xyz, 96);
  }
  return 0;
}

// `workers` host threads, each with its own context (stream + workspace) on one GPU, prove
// and then verify `reps` proofs each at the same time: a single proof is a chain of small
// dependent launches that leaves most of the GPU idle, so independent proofs overlap.
template <class Fq, class Fr, int CURVE>
int range_throughput(int device, const uint64_t* gx, const uint64_t* gy, const uint64_t* hx, const uint64_t* hy,
                     const uint64_t* ugh_xy, const uint64_t* values, const uint64_t* gamma, size_t m, unsigned n_bits,
                     uint64_t seed_rand, uint64_t seed_chal, int workers, int reps, double* prove_wall_ms,
                     double* verify_wall_ms, int* ok) {
  using R = RangeAggregate<Fq, Fr, CURVE>;
  using PO = PointOps<Fq, Fr>;
  const size_t n = (size_t)n_bits * m;
  std::vector<uint64_t> vals(values, values + m);
  std::vector<Scalar> gam = unpack_s(gamma, m);
  auto pt = [&](int k) { Affine a; memcpy(a.x, ugh_xy + 8 * k, 32); memcpy(a.y, ugh_xy + 8 * k + 4, 32); a.inf = 0; return a; };
  const Affine u = pt(0), g = pt(1), h = pt(2);
  std::vector<Projective> vcom(m);
  for (size_t j = 0; j < m; j++)
    vcom[j] = PO::lower(hx_add<Fq>(PO::mul(PO::lift(g), R::from_u64(vals[j])), PO::mul(PO::lift(h), gam[j])));
  std::mutex mu;
  std::condition_variable cv;
  int phase_ready[3] = {0, 0, 0};  // workers that reached: set-up done, proving done, verifying done
  std::atomic<int> failures{0};
  using clock = std::chrono::steady_clock;
  std::vector<clock::time_point> p0(workers), p1(workers), v0(workers), v1(workers);
  auto rendezvous = [&](int phase) {
    std::unique_lock<std::mutex> lk(mu);
    if (++phase_ready[phase] == workers) cv.notify_all();
    cv.wait(lk, [&] { return phase_ready[phase] >= workers; });
  };
  auto body = [&](int w) {
    bp_ctx* ctx = nullptr;
    bool reached[3] = {false, false, false};
    try {
      if (bp_init(device, &ctx) != BP_OK) throw std::runtime_error("bp_init");
      bp_set_timing(ctx, 0);  // latency-bound chains of small launches: no event markers
      typename R::Crs crs;
      crs.ipa.gs = unpack(gx, gy, nullptr, n);
      crs.ipa.hs = unpack(hx, hy, nullptr, n);
      crs.ipa.u = u; crs.g = g; crs.h = h;
      R::register_bases(ctx, crs, n);
      std::vector<RangeProof> proofs(reps);
      {  // warm-up: grows the context's workspace
        SeededTranscript<Fr> t(seed_chal + w);
        SeededRandom<Fr> rng(seed_rand + w);
        proofs[0] = R::prove(ctx, crs, vals, gam, n_bits, t, rng);
        SeededTranscript<Fr> t2(seed_chal + w);
        SeededRandom<Fr> vrng(seed_rand ^ 0x5EEDull);
        if (!R::verify(ctx, crs, vcom, n_bits, proofs[0], t2, vrng)) failures++;
      }
      reached[0] = true; rendezvous(0);
      p0[w] = clock::now();
      for (int r = 0; r < reps; r++) {
        SeededTranscript<Fr> t(seed_chal + w + 1000 * r);
        SeededRandom<Fr> rng(seed_rand + w + 1000 * r);
        proofs[r] = R::prove(ctx, crs, vals, gam, n_bits, t, rng);
      }
      p1[w] = clock::now();
      reached[1] = true; rendezvous(1);
      v0[w] = clock::now();
      for (int r = 0; r < reps; r++) {
        SeededTranscript<Fr> t(seed_chal + w + 1000 * r);
        SeededRandom<Fr> vrng(seed_rand ^ 0x5EEDull);
        if (!R::verify(ctx, crs, vcom, n_bits, proofs[r], t, vrng)) failures++;
      }
      v1[w] = clock::now();
      reached[2] = true; rendezvous(2);
      bp_bases_free(crs.gh);
    } catch (...) {
      failures++;
      for (int ph = 0; ph < 3; ph++)
        if (!reached[ph]) rendezvous(ph);  // never leave the other workers waiting
    }
    if (ctx) bp_shutdown(ctx);
  };
  std::vector<std::thread> th;
  for (int w = 0; w < workers; w++) th.emplace_back(body, w);
  for (auto& t : th) t.join();
  auto span = [&](std::vector<clock::time_point>& a, std::vector<clock::time_point>& b) {
    clock::time_point lo = a[0], hi = b[0];
    for (int w = 1; w < workers; w++) { if (a[w] < lo) lo = a[w]; if (b[w] > hi) hi = b[w]; }
    return std::chrono::duration<double, std::milli>(hi - lo).count();
  };
  *ok = failures.load() == 0;
  *prove_wall_ms = span(p0, p1);
  *verify_wall_ms = span(v0, v1);
  return 0;
}

// verify_batch of the reference's range benches (benches/aggregate_range_bench.rs:65-84, BATCH_SIZE 100):
// `distinct` proofs over different values are proved once (untimed); a batch of `batch` of them (cycling)
// is then verified: every proof's verify_aux (collect_ms), then the random linear combination and its
// one MSM, with the combination kept on the device (device_merge_ms) and on the host as the reference
// does it (host_merge_ms).  tamper != 0: the last proof of the batch carries t_hat + 1 and both
// combinations must reject.  values: distinct x m, gamma: distinct x m x 4.
template <class Fq, class Fr, int CURVE>
int range_verify_batch(bp_ctx* ctx, const uint64_t* gx, const uint64_t* gy, const uint64_t* hx, const uint64_t* hy,
                       const uint64_t* ugh_xy, const uint64_t* values, const uint64_t* gamma, size_t m, unsigned n_bits,
                       uint64_t seed_rand, uint64_t seed_chal, int distinct, int batch, int tamper, double* collect_ms,
                       double* device_merge_ms, double* host_merge_ms, int* ok_device, int* ok_host, double* direct_ms,
                       int* ok_direct) {
  using R = RangeAggregate<Fq, Fr, CURVE>;
  using PO = PointOps<Fq, Fr>;
  using M = Msm<Fq, Fr, CURVE>;
  const size_t n = (size_t)n_bits * m;
  typename R::Crs crs;
  crs.ipa.gs = unpack(gx, gy, nullptr, n);
  crs.ipa.hs = unpack(hx, hy, nullptr, n);
  auto pt = [&](int k) { Affine a; memcpy(a.x, ugh_xy + 8 * k, 32); memcpy(a.y, ugh_xy + 8 * k + 4, 32); a.inf = 0; return a; };
  crs.ipa.u = pt(0); crs.g = pt(1); crs.h = pt(2);
  R::register_bases(ctx, crs, n);
  struct Release { bp_bases* b; ~Release() { bp_bases_free(b); } } release{crs.gh};
  struct Timing { bp_ctx* c; Timing(bp_ctx* c_) : c(c_) { bp_set_timing(c, 0); } ~Timing() { bp_set_timing(c, 1); } } timing(ctx);
  std::vector<RangeProof> proofs(distinct);
  std::vector<std::vector<Projective>> vcoms(distinct);
  for (int p = 0; p < distinct; p++) {
    std::vector<uint64_t> vals(values + (size_t)p * m, values + (size_t)(p + 1) * m);
    std::vector<Scalar> gam = unpack_s(gamma + (size_t)p * m * 4, m);
    vcoms[p].resize(m);
    for (size_t j = 0; j < m; j++)
      vcoms[p][j] = PO::lower(hx_add<Fq>(PO::mul(PO::lift(crs.g), R::from_u64(vals[j])), PO::mul(PO::lift(crs.h), gam[j])));
    SeededTranscript<Fr> t1(seed_chal + p);
    SeededRandom<Fr> rng(seed_rand + p);
    proofs[p] = R::prove(ctx, crs, vals, gam, n_bits, t1, rng);
  }
  using clock = std::chrono::steady_clock;
  auto t0 = clock::now();
  std::vector<M> msms;
  msms.reserve(batch);
  for (int k = 0; k < batch; k++) {
    const int p = k % distinct;
    RangeProof pr = proofs[p];
    if (tamper && k == batch - 1) pr.t_hat = ScalarOps<Fr>::add(pr.t_hat, ScalarOps<Fr>::one());
    SeededTranscript<Fr> t2(seed_chal + p);
    SeededRandom<Fr> vrng((seed_rand + p) ^ 0x5EEDull);
    msms.push_back(R::verify_aux(ctx, crs, vcoms[p], n_bits, pr, t2, vrng));
  }
  auto t1 = clock::now();
  SeededTranscript<Fr> ta(seed_chal ^ 0xA1FAull);
  const Scalar alpha = ta.verifier_scalar();
  *ok_device = verify_batch_aux_resident(msms, alpha) ? 1 : 0;
  auto t2 = clock::now();
  *ok_host = verify_batch_aux(msms, alpha) ? 1 : 0;  // scales the accumulators in place: last
  auto t3 = clock::now();
  *collect_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
  *device_merge_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
  *host_merge_ms = std::chrono::duration<double, std::milli>(t3 - t2).count();
  // the whole batch with every proof's CRS scalars built and added on the device (R::verify_batch): same
  // proofs, same transcripts, same alpha, hence the same verdict
  {
    std::vector<RangeProof> prs(batch);
    std::vector<SeededTranscript<Fr>> ts;
    std::vector<SeededRandom<Fr>> rs;
    ts.reserve(batch); rs.reserve(batch);
    std::vector<typename R::BatchItem> items(batch);
    for (int k = 0; k < batch; k++) {
      const int p = k % distinct;
      prs[k] = proofs[p];
      if (tamper && k == batch - 1) prs[k].t_hat = ScalarOps<Fr>::add(prs[k].t_hat, ScalarOps<Fr>::one());
      ts.emplace_back(seed_chal + p);
      rs.emplace_back((seed_rand + p) ^ 0x5EEDull);
    }
    for (int k = 0; k < batch; k++) items[k] = typename R::BatchItem{&vcoms[k % distinct], &prs[k], &ts[k], &rs[k]};
    auto t4 = clock::now();
    *ok_direct = R::verify_batch(ctx, crs, n_bits, items, alpha) ? 1 : 0;
    auto t5 = clock::now();
    *direct_ms = std::chrono::duration<double, std::milli>(t5 - t4).count();
  }
  return 0;
}

}  // namespace

extern "C" {

// workers x reps aggregated range proofs proved, then verified, concurrently on one GPU
// (one context per worker thread).  Wall times cover all workers' proving / verifying.
int bph_range_agg_throughput(int device, int curve, const uint64_t* gx, const uint64_t* gy, const uint64_t* hx,
                             const uint64_t* hy, const uint64_t* ugh_xy, const uint64_t* values, const uint64_t* gamma,
                             size_t m, unsigned n_bits, uint64_t seed_rand, uint64_t seed_chal, int workers, int reps,
                             double* prove_wall_ms, double* verify_wall_ms, int* ok) {
  if (workers < 1 || workers > 16 || reps < 1) return BP_E_ARG;
  try {
    if (curve == BP_SECP256K1)
      return range_throughput<Secp256k1Fq, Secp256k1Fr, BP_SECP256K1>(device, gx, gy, hx, hy, ugh_xy, values, gamma, m, n_bits,
                                                                     seed_rand, seed_chal, workers, reps, prove_wall_ms,
                                                                     verify_wall_ms, ok);
    if (curve == BP_VESTA)
      return range_throughput<VestaFq, VestaFr, BP_VESTA>(device, gx, gy, hx, hy, ugh_xy, values, gamma, m, n_bits, seed_rand,
                                                         seed_chal, workers, reps, prove_wall_ms, verify_wall_ms, ok);
    return BP_E_CURVE;
  } catch (const VerificationError&) {
    return BPH_E_VERIFY;
  } catch (const std::invalid_argument&) {
    return BP_E_ARG;
  } catch (const std::exception&) {
    return BP_E_HIP;
  }
}

int bph_range_agg_verify_batch(bp_ctx* ctx, int curve, const uint64_t* gx, const uint64_t* gy, const uint64_t* hx,
                               const uint64_t* hy, const uint64_t* ugh_xy, const uint64_t* values, const uint64_t* gamma,
                               size_t m, unsigned n_bits, uint64_t seed_rand, uint64_t seed_chal, int distinct, int batch,
                               int tamper, double* collect_ms, double* device_merge_ms, double* host_merge_ms, int* ok_device,
                               int* ok_host, double* direct_ms, int* ok_direct) {
  if (distinct < 1 || batch < 1 || !direct_ms || !ok_direct) return BP_E_ARG;
  try {
    if (curve == BP_SECP256K1)
      return range_verify_batch<Secp256k1Fq, Secp256k1Fr, BP_SECP256K1>(ctx, gx, gy, hx, hy, ugh_xy, values, gamma, m, n_bits,
                                                                       seed_rand, seed_chal, distinct, batch, tamper, collect_ms,
                                                                       device_merge_ms, host_merge_ms, ok_device, ok_host, direct_ms,
                                                                       ok_direct);
    if (curve == BP_VESTA)
      return range_verify_batch<VestaFq, VestaFr, BP_VESTA>(ctx, gx, gy, hx, hy, ugh_xy, values, gamma, m, n_bits, seed_rand,
                                                           seed_chal, distinct, batch, tamper, collect_ms, device_merge_ms,
                                                           host_merge_ms, ok_device, ok_host, direct_ms, ok_direct);
    return BP_E_CURVE;
  } catch (const VerificationError&) {
    return BPH_E_VERIFY;
  } catch (const std::invalid_argument&) {
    return BP_E_ARG;
  } catch (const std::exception&) {
    return BP_E_HIP;
  }
}

// batch x (m values of n_bits bits) aggregated range proofs proved together (RangeAggregate::prove_batch),
// reps times; batch_ms: wall time of one batch on the calling thread.  values: batch x m; gamma: batch x m
// scalars; proof b uses the seeds seed_rand + b, seed_chal + b.  host_blinders != 0: the blinders are drawn on
// the host and uploaded.  verify != 0: every proof is then checked by RangeAggregate::verify (ok).  Outputs
// per proof as bph_range_agg_roundtrip's: out_points batch x 48, out_scalars batch x 20, out_lr batch x
// log2(N) x 24 u64.
int bph_range_agg_prove_batch(bp_ctx* ctx, int curve, const uint64_t* gx, const uint64_t* gy, const uint64_t* hx,
                              const uint64_t* hy, const uint64_t* ugh_xy, const uint64_t* values, const uint64_t* gamma,
                              size_t batch, size_t m, unsigned n_bits, uint64_t seed_rand, uint64_t seed_chal, int reps,
                              int host_blinders, int verify, uint64_t* out_points, uint64_t* out_scalars, uint64_t* out_lr,
                              double* batch_ms, int* ok) {
  if (reps < 1 || batch < 1 || !out_points || !out_scalars || !out_lr || !batch_ms || !ok) return BP_E_ARG;
  try {
    if (curve == BP_SECP256K1)
      return range_prove_batch<Secp256k1Fq, Secp256k1Fr, BP_SECP256K1>(ctx, gx, gy, hx, hy, ugh_xy, values, gamma, batch, m,
                                                                      n_bits, seed_rand, seed_chal, reps, host_blinders, verify,
                                                                      out_points, out_scalars, out_lr, batch_ms, ok);
    if (curve == BP_VESTA)
      return range_prove_batch<VestaFq, VestaFr, BP_VESTA>(ctx, gx, gy, hx, hy, ugh_xy, values, gamma, batch, m, n_bits, seed_rand,
                                                          seed_chal, reps, host_blinders, verify, out_points, out_scalars,
                                                          out_lr, batch_ms, ok);
    return BP_E_CURVE;
  } catch (const VerificationError&) {
    return BPH_E_VERIFY;
  } catch (const std::invalid_argument&) {
    return BP_E_ARG;
  } catch (const std::exception&) {
    return BP_E_HIP;
  }
}

// Aggregated range proof over m values of n_bits bits (N = m*n_bits a power of two):
// prove, then verify, `reps` times with identical seeds; averages the wall time of
// each phase.  ugh_xy = affine u, g, h (3 x 8 u64).  Outputs of the last proof:
// out_points = A, S, T1, T2 (4 x 12 u64); out_scalars = tao_x, mu, t_hat, a, b
// (5 x 4 u64); out_lr = log2(N) x (L, R) (24 u64 each); out_value (may be NULL) = the value of the
// verifier's deferred MSM on the checked proof.  tamper: 0 none, 1 t_hat + 1, 2 a round dropped,
// 3 a round appended (2 and 3 must be refused for their shape).  single != 0 (m = 1) drives the
// single-value types of src/range/mod.rs.
int bph_range_agg_roundtrip(bp_ctx* ctx, int curve, const uint64_t* gx, const uint64_t* gy, const uint64_t* hx,
                            const uint64_t* hy, const uint64_t* ugh_xy, const uint64_t* values, const uint64_t* gamma,
                            size_t m, unsigned n_bits, uint64_t seed_rand, uint64_t seed_chal, int tamper, int reps,
                            int single, uint64_t* out_points, uint64_t* out_scalars, uint64_t* out_lr, uint64_t* out_value,
                            double* prove_ms, double* verify_ms, int* ok) {
  try {
    if (curve == BP_SECP256K1)
      return range_roundtrip<Secp256k1Fq, Secp256k1Fr, BP_SECP256K1>(ctx, gx, gy, hx, hy, ugh_xy, values, gamma, m, n_bits,
                                                                    seed_rand, seed_chal, tamper, reps, single, out_points,
                                                                    out_scalars, out_lr, out_value, prove_ms, verify_ms, ok);
    if (curve == BP_VESTA)
      return range_roundtrip<VestaFq, VestaFr, BP_VESTA>(ctx, gx, gy, hx, hy, ugh_xy, values, gamma, m, n_bits, seed_rand,
                                                        seed_chal, tamper, reps, single, out_points, out_scalars, out_lr,
                                                        out_value, prove_ms, verify_ms, ok);
    return BP_E_CURVE;
  } catch (const VerificationError&) {
    return BPH_E_VERIFY;
  } catch (const std::invalid_argument&) {
    return BP_E_ARG;
  } catch (const std::exception&) {
    return BP_E_HIP;
  }
}

}  // extern "C"

// ---- polynomial commitment round trip -----------------------------------------------------
#include "bp_host_poly.hpp"

namespace {

template <class Fq, class Fr, int CURVE>
int poly_roundtrip(bp_ctx* ctx, const uint64_t* gx, const uint64_t* gy, const uint64_t* h_gen_xy, const uint64_t* coeffs,
                   size_t n_coeffs, size_t n, const uint64_t* r_x, uint64_t seed_rand, uint64_t seed_chal, int tamper,
                   int halo, uint64_t* out_l, uint64_t* out_r, uint64_t* out_ar, uint64_t* out_g, uint64_t* out_commit,
                   uint64_t* out_eval, double* prove_ms, double* verify_ms, int* ok_verify, int* ok_lazy, int* ok_halo,
                   uint64_t* halo_out, uint64_t* halo_w) {
  using PC = PolyCommit<Fq, Fr, CURVE>;
  using S = ScalarOps<Fr>;
  typename PC::Crs crs;
  crs.gs = unpack(gx, gy, nullptr, n);
  auto pt = [&](int k) { Affine a; memcpy(a.x, h_gen_xy + 8 * k, 32); memcpy(a.y, h_gen_xy + 8 * k + 4, 32); a.inf = 0; return a; };
  crs.h = pt(0); crs.gen = pt(1);
  PC::register_bases(ctx, crs);
  *ok_verify = *ok_lazy = *ok_halo = 0;
  try {
    std::vector<Scalar> p = unpack_s(coeffs, n_coeffs);
    Scalar r, x;
    memcpy(r.l, r_x, 32); memcpy(x.l, r_x + 4, 32);
    auto st = PC::statement(ctx, crs, p, r, x);
    PolyProof proof;
    SeededTranscript<Fr> t1(seed_chal);
    SeededRandom<Fr> rng(seed_rand);
    auto a0 = std::chrono::steady_clock::now();
    Todo todo = PC::prove(ctx, crs, st, p, r, t1, rng, proof);
    auto a1 = std::chrono::steady_clock::now();
    auto st_check = st;
    if (tamper == 1) st_check.evaluation = S::add(st_check.evaluation, S::one());
    PolyProof check_proof = proof;
    if (tamper == 2 && !check_proof.L.empty()) { check_proof.L.pop_back(); check_proof.R.pop_back(); }   // a round dropped
    if (tamper == 3) { check_proof.L.push_back(st.commitment); check_proof.R.push_back(st.commitment); }  // a round appended
    if (tamper == 4) check_proof.R.push_back(st.commitment);                                              // ragged
    SeededTranscript<Fr> t2(seed_chal);
    auto b0 = std::chrono::steady_clock::now();
    *ok_verify = PC::verify(ctx, crs, st_check, check_proof, t2) ? 1 : 0;
    auto b1 = std::chrono::steady_clock::now();
    *prove_ms = std::chrono::duration<double, std::milli>(a1 - a0).count();
    *verify_ms = std::chrono::duration<double, std::milli>(b1 - b0).count();
    std::vector<Todo> todos;
    SeededTranscript<Fr> t3(seed_chal);
    *ok_lazy = PC::lazy_verify(ctx, crs, st_check, todo.g, check_proof, t3, todos) ? 1 : 0;
    size_t k = proof.L.size();
    for (size_t j = 0; j < k; j++) { memcpy(out_l + 12 * j, proof.L[j].xyz, 96); memcpy(out_r + 12 * j, proof.R[j].xyz, 96); }
    memcpy(out_ar, proof.a.l, 32); memcpy(out_ar + 4, proof.r.l, 32);
    memcpy(out_g, todo.g.x, 32); memcpy(out_g + 4, todo.g.y, 32); out_g[8] = todo.g.inf;
    memcpy(out_commit, st.commitment.xyz, 96);
    memcpy(out_eval, st.evaluation.l, 32);
    if (halo && !tamper) {
      // a second opening (reversed coefficients), both checked lazily; then the two deferred
      // MSMs are folded into one statement that is proved and fully verified (the
      // "halo trick": src/poly_commit/mod.rs:228-262, benches/poly_commit_bench.rs)
      std::vector<Scalar> p2(p.rbegin(), p.rend());
      auto st2 = PC::statement(ctx, crs, p2, r, x);
      PolyProof proof2;
      SeededTranscript<Fr> t4(seed_chal + 7777);
      SeededRandom<Fr> rng2(seed_rand + 7777);
      Todo todo2 = PC::prove(ctx, crs, st2, p2, r, t4, rng2, proof2);
      SeededTranscript<Fr> t5(seed_chal + 7777);
      bool ok = PC::lazy_verify(ctx, crs, st2, todo2.g, proof2, t5, todos) && todos.size() == 2;
      SeededTranscript<Fr> tc(seed_chal ^ 0xA1FAull);
      Scalar alpha = tc.verifier_scalar(), x2 = tc.verifier_scalar();
      std::vector<Scalar> w = PC::fold_todos_witness(ctx, todos, alpha);
      auto fst = PC::fold_todos_statement(ctx, todos, alpha, x2);
      auto wst = PC::statement(ctx, crs, w, S::zero(), x2);
      ok = ok && memcmp(wst.commitment.xyz, fst.commitment.xyz, 96) == 0 && wst.evaluation == fst.evaluation;
      PolyProof proof3;
      SeededTranscript<Fr> t6(seed_chal + 9999);
      SeededRandom<Fr> rng3(seed_rand + 9999);
      PC::prove(ctx, crs, fst, w, S::zero(), t6, rng3, proof3);
      SeededTranscript<Fr> t7(seed_chal + 9999);
      ok = ok && PC::verify(ctx, crs, fst, proof3, t7);
      *ok_halo = ok ? 1 : 0;
      if (halo_out) {
        uint64_t* o = halo_out;
        auto put_proof = [&](const PolyProof& pf) {
          for (size_t j = 0; j < k; j++) { memcpy(o, pf.L[j].xyz, 96); o += 12; }
          for (size_t j = 0; j < k; j++) { memcpy(o, pf.R[j].xyz, 96); o += 12; }
          memcpy(o, pf.a.l, 32); memcpy(o + 4, pf.r.l, 32); o += 8;
        };
        put_proof(proof2);
        memcpy(o, todo2.g.x, 32); memcpy(o + 4, todo2.g.y, 32); o[8] = todo2.g.inf; o += 9;
        memcpy(o, st2.commitment.xyz, 96); o += 12;
        memcpy(o, st2.evaluation.l, 32); o += 4;
        memcpy(o, alpha.l, 32); memcpy(o + 4, x2.l, 32); o += 8;
        memcpy(o, fst.commitment.xyz, 96); o += 12;
        memcpy(o, fst.evaluation.l, 32); o += 4;
        put_proof(proof3);
      }
      if (halo_w) memcpy(halo_w, w[0].l, 32 * w.size());
    }
  } catch (...) {
    bp_bases_free(crs.bases);
    throw;
  }
  bp_bases_free(crs.bases);
  return 0;
}

}  // namespace

extern "C" {

// Opening proof of a polynomial of n_coeffs coefficients under a CRS of n = 2^k generators:
// statement, prove, verify, lazy_verify, and (halo != 0) the fold of two deferred checks.
// h_gen_xy = affine h, group generator (2 x 8 u64); r_x = blinder r, evaluation point x.
// Outputs: L, R (k x 12 u64 each), (a, r), folded generator g (x|y|inf: 9 u64), the
// commitment (12) and the evaluation (4) of the first statement.  tamper: 0 none, 1 evaluation + 1,
// 2 a round dropped, 3 a round appended, 4 one more R than L.  With halo != 0, halo_out (may be NULL;
// 48k + 65 u64) receives, in this order: the second opening (reversed coefficients; transcript seed
// seed_chal + 7777) as L (12k), R (12k), a, r (8), its folded generator (9), its commitment (12) and
// evaluation (4); the fold challenges alpha, x (8); the folded statement's commitment (12) and
// evaluation (4); the opening of the folded statement (seed_chal + 9999) as L, R, a, r; and halo_w (may
// be NULL; n x 4 u64) the folded witness's coefficients.
int bph_poly_roundtrip(bp_ctx* ctx, int curve, const uint64_t* gx, const uint64_t* gy, const uint64_t* h_gen_xy,
                       const uint64_t* coeffs, size_t n_coeffs, size_t n, const uint64_t* r_x, uint64_t seed_rand,
                       uint64_t seed_chal, int tamper, int halo, uint64_t* out_l, uint64_t* out_r, uint64_t* out_ar,
                       uint64_t* out_g, uint64_t* out_commit, uint64_t* out_eval, double* prove_ms, double* verify_ms,
                       int* ok_verify, int* ok_lazy, int* ok_halo, uint64_t* halo_out, uint64_t* halo_w) {
  try {
    if (curve == BP_SECP256K1)
      return poly_roundtrip<Secp256k1Fq, Secp256k1Fr, BP_SECP256K1>(ctx, gx, gy, h_gen_xy, coeffs, n_coeffs, n, r_x, seed_rand,
                                                                   seed_chal, tamper, halo, out_l, out_r, out_ar, out_g,
                                                                   out_commit, out_eval, prove_ms, verify_ms, ok_verify,
                                                                   ok_lazy, ok_halo, halo_out, halo_w);
    if (curve == BP_VESTA)
      return poly_roundtrip<VestaFq, VestaFr, BP_VESTA>(ctx, gx, gy, h_gen_xy, coeffs, n_coeffs, n, r_x, seed_rand, seed_chal,
                                                       tamper, halo, out_l, out_r, out_ar, out_g, out_commit, out_eval,
                                                       prove_ms, verify_ms, ok_verify, ok_lazy, ok_halo, halo_out, halo_w);
    return BP_E_CURVE;
  } catch (const VerificationError&) {
    return BPH_E_VERIFY;
  } catch (const std::invalid_argument&) {
    return BP_E_ARG;
  } catch (const std::exception&) {
    return BP_E_HIP;
  }
}

}  // extern "C"

// ---- several openings verified together -----------------------------------------------------------
namespace {

// `count` openings over one CRS (opening i: coefficients coeffs[i], blinder and point r_x[i], seeds + i) proved
// one after the other, then checked together by PolyCommit::verify_batch (`reps` times, averaged) and, for
// comparison, one by one by PolyCommit::verify.  tamper: 0 none; 1 the evaluation of opening tamper_at + 1;
// 2 a round dropped from its proof.  batch > count cycles through the openings (the shape of the reference's
// verify_batch benches).
template <class Fq, class Fr, int CURVE>
int poly_verify_batch(bp_ctx* ctx, const uint64_t* gx, const uint64_t* gy, const uint64_t* h_gen_xy, const uint64_t* coeffs,
                      size_t n_coeffs, size_t n, const uint64_t* r_x, size_t count, size_t batch, uint64_t seed_rand,
                      uint64_t seed_chal, int tamper, size_t tamper_at, int reps, uint64_t* out_l, uint64_t* out_r, uint64_t* out_ar,
                      uint64_t* out_commit, uint64_t* out_eval, double* batch_ms, double* one_by_one_ms, int* ok_batch,
                      int* ok_each) {
  using PC = PolyCommit<Fq, Fr, CURVE>;
  using S = ScalarOps<Fr>;
  typename PC::Crs crs;
  crs.gs = unpack(gx, gy, nullptr, n);
  auto pt = [&](int k) { Affine a; memcpy(a.x, h_gen_xy + 8 * k, 32); memcpy(a.y, h_gen_xy + 8 * k + 4, 32); a.inf = 0; return a; };
  crs.h = pt(0); crs.gen = pt(1);
  PC::register_bases(ctx, crs);
  struct Release { bp_bases* b; ~Release() { bp_bases_free(b); } } release{crs.bases};
  size_t k = 0;
  while (((size_t)1 << k) < n) k++;
  std::vector<typename PC::Statement> sts(count);
  std::vector<PolyProof> proofs(count);
  for (size_t i = 0; i < count; i++) {
    std::vector<Scalar> p = unpack_s(coeffs + 4 * n_coeffs * i, n_coeffs);
    Scalar r, x;
    memcpy(r.l, r_x + 8 * i, 32); memcpy(x.l, r_x + 8 * i + 4, 32);
    sts[i] = PC::statement(ctx, crs, p, r, x);
    SeededTranscript<Fr> t(seed_chal + i);
    SeededRandom<Fr> rng(seed_rand + i);
    PC::prove(ctx, crs, sts[i], p, r, t, rng, proofs[i]);
    for (size_t j = 0; j < k; j++) {
      memcpy(out_l + (i * k + j) * 12, proofs[i].L[j].xyz, 96);
      memcpy(out_r + (i * k + j) * 12, proofs[i].R[j].xyz, 96);
    }
    memcpy(out_ar + 8 * i, proofs[i].a.l, 32); memcpy(out_ar + 8 * i + 4, proofs[i].r.l, 32);
    memcpy(out_commit + 12 * i, sts[i].commitment.xyz, 96);
    memcpy(out_eval + 4 * i, sts[i].evaluation.l, 32);
  }
  std::vector<typename PC::Statement> bst(batch);
  std::vector<PolyProof> bpr(batch);
  for (size_t q = 0; q < batch; q++) { bst[q] = sts[q % count]; bpr[q] = proofs[q % count]; }
  if (tamper == 1) bst[tamper_at % batch].evaluation = S::add(bst[tamper_at % batch].evaluation, S::one());
  if (tamper == 2 && !bpr[tamper_at % batch].L.empty()) { bpr[tamper_at % batch].L.pop_back(); bpr[tamper_at % batch].R.pop_back(); }
  SeededTranscript<Fr> ta(seed_chal ^ 0xA1FAull);
  const Scalar alpha = ta.verifier_scalar();
  using clock = std::chrono::steady_clock;
  double tb = 0, te = 0;
  *ok_batch = 1; *ok_each = 1;
  for (int rep = 0; rep < reps + 1; rep++) {  // the first pass grows the workspace and is not timed
    std::vector<SeededTranscript<Fr>> ts;
    ts.reserve(batch);
    std::vector<typename PC::BatchItem> items(batch);
    for (size_t q = 0; q < batch; q++) { ts.emplace_back(seed_chal + q % count); items[q] = typename PC::BatchItem{&bst[q], &bpr[q], &ts[q]}; }
    auto t0 = clock::now();
    const bool ok = PC::verify_batch(ctx, crs, items, alpha);
    auto t1 = clock::now();
    if (rep) tb += std::chrono::duration<double, std::milli>(t1 - t0).count();
    *ok_batch = ok ? 1 : 0;
  }
  {
    auto t0 = clock::now();
    for (size_t q = 0; q < batch; q++) {
      SeededTranscript<Fr> t(seed_chal + q % count);
      if (!PC::verify(ctx, crs, bst[q], bpr[q], t)) *ok_each = 0;
    }
    te = std::chrono::duration<double, std::milli>(clock::now() - t0).count();
  }
  *batch_ms = tb / reps;
  *one_by_one_ms = te;
  return 0;
}

}  // namespace

extern "C" int bph_poly_verify_batch(bp_ctx* ctx, int curve, const uint64_t* gx, const uint64_t* gy, const uint64_t* h_gen_xy,
                                     const uint64_t* coeffs, size_t n_coeffs, size_t n, const uint64_t* r_x, size_t count,
                                     size_t batch, uint64_t seed_rand, uint64_t seed_chal, int tamper, size_t tamper_at, int reps,
                                     uint64_t* out_l, uint64_t* out_r, uint64_t* out_ar, uint64_t* out_commit, uint64_t* out_eval,
                                     double* batch_ms, double* one_by_one_ms, int* ok_batch, int* ok_each) {
  if (!ctx || count < 1 || batch < 1 || reps < 1 || !batch_ms || !one_by_one_ms || !ok_batch || !ok_each) return BP_E_ARG;
  try {
    if (curve == BP_SECP256K1)
      return poly_verify_batch<Secp256k1Fq, Secp256k1Fr, BP_SECP256K1>(ctx, gx, gy, h_gen_xy, coeffs, n_coeffs, n, r_x, count, batch,
                                                                      seed_rand, seed_chal, tamper, tamper_at, reps, out_l, out_r,
                                                                      out_ar, out_commit, out_eval, batch_ms, one_by_one_ms, ok_batch,
                                                                      ok_each);
    if (curve == BP_VESTA)
      return poly_verify_batch<VestaFq, VestaFr, BP_VESTA>(ctx, gx, gy, h_gen_xy, coeffs, n_coeffs, n, r_x, count, batch, seed_rand,
                                                          seed_chal, tamper, tamper_at, reps, out_l, out_r, out_ar, out_commit,
                                                          out_eval, batch_ms, one_by_one_ms, ok_batch, ok_each);
    return BP_E_CURVE;
  } catch (const VerificationError&) {
    return BPH_E_VERIFY;
  } catch (const std::invalid_argument&) {
    return BP_E_ARG;
  } catch (const std::exception&) {
    return BP_E_HIP;
  }
}

// ---- polynomial commitment over several ranks ------------------------------------------------
#include "bp_host_sharded.hpp"

namespace {

struct SpolyOut {
  uint64_t *l, *r, *ar, *g, *commit, *eval, *halo_out, *halo_w;
  double *prove_ms, *verify_ms;
  int *ok_verify, *ok_lazy, *ok_halo;
};

// One rank's run of the flow of poly_roundtrip (statement, prove, verify, lazy_verify and, with a second
// coefficient slice, the fold of the two deferred checks and its opening) over slices: gx, gy and the
// coefficient vectors hold the elements rank, rank + world, ... of the whole vectors.  Outputs as
// bph_poly_roundtrip; halo_w receives this rank's slice of the folded witness (n / world scalars).
template <class Fq, class Fr, int CURVE>
int spoly_roundtrip(bp_ctx* ctx, Exchange& ex, const uint64_t* gx, const uint64_t* gy, const uint64_t* h_gen_xy,
                    const uint64_t* coeffs, const uint64_t* coeffs2, size_t n_total, const uint64_t* r_x, uint64_t seed_rand,
                    uint64_t seed_chal, int tamper, const SpolyOut& o) {
  using SP = ShardedPolyCommit<Fq, Fr, CURVE>;
  using S = ScalarOps<Fr>;
  typename SP::Crs crs;
  crs.n = n_total; crs.rank = ex.rank; crs.world = ex.world;
  const size_t nl = n_total / ex.world;
  crs.gs = unpack(gx, gy, nullptr, nl);
  auto pt = [&](int k) { Affine a; memcpy(a.x, h_gen_xy + 8 * k, 32); memcpy(a.y, h_gen_xy + 8 * k + 4, 32); a.inf = 0; return a; };
  crs.h = pt(0); crs.gen = pt(1);
  SP::register_bases(ctx, crs);
  *o.ok_verify = *o.ok_lazy = *o.ok_halo = 0;
  try {
    std::vector<Scalar> p = unpack_s(coeffs, nl);
    Scalar r, x;
    memcpy(r.l, r_x, 32); memcpy(x.l, r_x + 4, 32);
    auto st = SP::statement(ctx, crs, ex, p, r, x);
    PolyProof proof;
    SeededTranscript<Fr> t1(seed_chal);
    SeededRandom<Fr> rng(seed_rand);
    auto a0 = std::chrono::steady_clock::now();
    Todo todo = SP::prove(ctx, crs, ex, st, p, r, t1, rng, proof);
    auto a1 = std::chrono::steady_clock::now();
    auto st_check = st;
    if (tamper == 1) st_check.evaluation = S::add(st_check.evaluation, S::one());
    PolyProof check_proof = proof;
    if (tamper == 2 && !check_proof.L.empty()) { check_proof.L.pop_back(); check_proof.R.pop_back(); }
    if (tamper == 3) { check_proof.L.push_back(st.commitment); check_proof.R.push_back(st.commitment); }
    SeededTranscript<Fr> t2(seed_chal);
    auto b0 = std::chrono::steady_clock::now();
    *o.ok_verify = SP::verify(ctx, crs, ex, st_check, check_proof, t2) ? 1 : 0;
    auto b1 = std::chrono::steady_clock::now();
    *o.prove_ms = std::chrono::duration<double, std::milli>(a1 - a0).count();
    *o.verify_ms = std::chrono::duration<double, std::milli>(b1 - b0).count();
    std::vector<Todo> todos;
    SeededTranscript<Fr> t3(seed_chal);
    *o.ok_lazy = SP::lazy_verify(ctx, crs, st_check, todo.g, check_proof, t3, todos) ? 1 : 0;
    const size_t k = proof.L.size();
    for (size_t j = 0; j < k; j++) { memcpy(o.l + 12 * j, proof.L[j].xyz, 96); memcpy(o.r + 12 * j, proof.R[j].xyz, 96); }
    memcpy(o.ar, proof.a.l, 32); memcpy(o.ar + 4, proof.r.l, 32);
    memcpy(o.g, todo.g.x, 32); memcpy(o.g + 4, todo.g.y, 32); o.g[8] = todo.g.inf;
    memcpy(o.commit, st.commitment.xyz, 96);
    memcpy(o.eval, st.evaluation.l, 32);
    if (coeffs2 && !tamper) {
      std::vector<Scalar> p2 = unpack_s(coeffs2, nl);
      auto st2 = SP::statement(ctx, crs, ex, p2, r, x);
      PolyProof proof2;
      SeededTranscript<Fr> t4(seed_chal + 7777);
      SeededRandom<Fr> rng2(seed_rand + 7777);
      Todo todo2 = SP::prove(ctx, crs, ex, st2, p2, r, t4, rng2, proof2);
      SeededTranscript<Fr> t5(seed_chal + 7777);
      bool ok = SP::lazy_verify(ctx, crs, st2, todo2.g, proof2, t5, todos) && todos.size() == 2;
      SeededTranscript<Fr> tc(seed_chal ^ 0xA1FAull);
      Scalar alpha = tc.verifier_scalar(), x2 = tc.verifier_scalar();
      std::vector<Scalar> w = SP::fold_todos_witness(ctx, crs, todos, alpha);
      auto fst = SP::fold_todos_statement(ctx, todos, alpha, x2);
      auto wst = SP::statement(ctx, crs, ex, w, S::zero(), x2);
      ok = ok && memcmp(wst.commitment.xyz, fst.commitment.xyz, 96) == 0 && wst.evaluation == fst.evaluation;
      PolyProof proof3;
      SeededTranscript<Fr> t6(seed_chal + 9999);
      SeededRandom<Fr> rng3(seed_rand + 9999);
      SP::prove(ctx, crs, ex, fst, w, S::zero(), t6, rng3, proof3);
      SeededTranscript<Fr> t7(seed_chal + 9999);
      ok = SP::verify(ctx, crs, ex, fst, proof3, t7) && ok;
      *o.ok_halo = ok ? 1 : 0;
      if (o.halo_out) {
        uint64_t* q = o.halo_out;
        auto put_proof = [&](const PolyProof& pf) {
          for (size_t j = 0; j < k; j++) { memcpy(q, pf.L[j].xyz, 96); q += 12; }
          for (size_t j = 0; j < k; j++) { memcpy(q, pf.R[j].xyz, 96); q += 12; }
          memcpy(q, pf.a.l, 32); memcpy(q + 4, pf.r.l, 32); q += 8;
        };
        put_proof(proof2);
        memcpy(q, todo2.g.x, 32); memcpy(q + 4, todo2.g.y, 32); q[8] = todo2.g.inf; q += 9;
        memcpy(q, st2.commitment.xyz, 96); q += 12;
        memcpy(q, st2.evaluation.l, 32); q += 4;
        memcpy(q, alpha.l, 32); memcpy(q + 4, x2.l, 32); q += 8;
        memcpy(q, fst.commitment.xyz, 96); q += 12;
        memcpy(q, fst.evaluation.l, 32); q += 4;
        put_proof(proof3);
      }
      if (o.halo_w) memcpy(o.halo_w, w[0].l, 32 * w.size());
    }
  } catch (...) {
    bp_bases_free(crs.bases);
    throw;
  }
  bp_bases_free(crs.bases);
  return 0;
}

template <class F>
int spoly_dispatch(int curve, F&& f) {
  try {
    return f(curve);
  } catch (const VerificationError&) {
    return BPH_E_VERIFY;
  } catch (const std::invalid_argument&) {
    return BP_E_ARG;
  } catch (const std::exception&) {
    return BP_E_HIP;
  }
}

}  // namespace

extern "C" {

// One rank of the polynomial-commitment flow over `world` ranks (bp_host_sharded.hpp); `gather` is the
// caller's all-gather of small host packets: gather(user, mine, len_u64, all) fills all[r * len ..] with rank
// r's packet on every rank and returns 0.  gx, gy: this rank's n_total / world generators (rank, rank +
// world, ...); coeffs, coeffs2 (may be NULL: no second opening and no fold): the same slice of each
// coefficient vector, zero-padded to the slice length.  Outputs as bph_poly_roundtrip, identical on every
// rank, except halo_w: this rank's slice of the folded witness.
int bph_spoly_roundtrip(bp_ctx* ctx, int curve, size_t rank, size_t world, int (*gather)(void*, const uint64_t*, size_t, uint64_t*),
                        void* user, const uint64_t* gx, const uint64_t* gy, const uint64_t* h_gen_xy, const uint64_t* coeffs,
                        const uint64_t* coeffs2, size_t n_total, const uint64_t* r_x, uint64_t seed_rand, uint64_t seed_chal,
                        int tamper, uint64_t* out_l, uint64_t* out_r, uint64_t* out_ar, uint64_t* out_g, uint64_t* out_commit,
                        uint64_t* out_eval, double* prove_ms, double* verify_ms, int* ok_verify, int* ok_lazy, int* ok_halo,
                        uint64_t* halo_out, uint64_t* halo_w) {
  if (!ctx || world == 0 || rank >= world || (world > 1 && !gather)) return BP_E_ARG;
  CallbackExchange ex;
  ex.rank = rank; ex.world = world; ex.fn = gather; ex.user = user;
  SpolyOut o{out_l, out_r, out_ar, out_g, out_commit, out_eval, halo_out, halo_w, prove_ms, verify_ms, ok_verify, ok_lazy, ok_halo};
  return spoly_dispatch(curve, [&](int c) {
    if (c == BP_SECP256K1)
      return spoly_roundtrip<Secp256k1Fq, Secp256k1Fr, BP_SECP256K1>(ctx, ex, gx, gy, h_gen_xy, coeffs, coeffs2, n_total, r_x, seed_rand,
                                                                    seed_chal, tamper, o);
    if (c == BP_VESTA)
      return spoly_roundtrip<VestaFq, VestaFr, BP_VESTA>(ctx, ex, gx, gy, h_gen_xy, coeffs, coeffs2, n_total, r_x, seed_rand, seed_chal,
                                                        tamper, o);
    return (int)BP_E_CURVE;
  });
}

// The same flow with the ranks as threads of this process: rank r runs on HIP device first_device +
// r mod n_devices with a context of its own, the exchange is a barrier over a shared buffer.  Takes the whole
// vectors (gx, gy: n generators; coeffs: n_coeffs <= n scalars, zero-padded; halo != 0: the second opening is
// of the reversed coefficient vector, as in bph_poly_roundtrip) and cuts the slices itself.  Returns rank 0's
// outputs after checking that every rank produced the same; halo_w: the whole folded witness (n scalars).
// prove_ms / verify_ms: the slowest rank's.
int bph_spoly_roundtrip_threads(int first_device, int n_devices, size_t world, int curve, const uint64_t* gx, const uint64_t* gy,
                                const uint64_t* h_gen_xy, const uint64_t* coeffs, size_t n_coeffs, size_t n, const uint64_t* r_x,
                                uint64_t seed_rand, uint64_t seed_chal, int tamper, int halo, uint64_t* out_l, uint64_t* out_r,
                                uint64_t* out_ar, uint64_t* out_g, uint64_t* out_commit, uint64_t* out_eval, double* prove_ms,
                                double* verify_ms, int* ok_verify, int* ok_lazy, int* ok_halo, uint64_t* halo_out, uint64_t* halo_w) {
  if (world == 0 || (world & (world - 1)) || world > 64 || n == 0 || (n & (n - 1)) || n < world || n_coeffs > n || n_devices < 1)
    return BP_E_ARG;
  size_t k = 0;
  while (((size_t)1 << k) < n) k++;
  const size_t nl = n / world, halo_len = 48 * k + 65, kk = k ? k : 1;
  struct RankOut {
    std::vector<uint64_t> l, r, ar, g, commit, eval, halo_out, halo_w;
    double prove_ms = 0, verify_ms = 0;
    int ok_verify = 0, ok_lazy = 0, ok_halo = 0, rc = BP_E_HIP;
  };
  std::vector<RankOut> outs(world);
  LocalExchangeHub hub(world);
  std::vector<std::thread> threads;
  for (size_t rank = 0; rank < world; rank++) {
    threads.emplace_back([&, rank] {
      RankOut& ro = outs[rank];
      ro.l.assign(12 * kk, 0); ro.r.assign(12 * kk, 0); ro.ar.assign(8, 0); ro.g.assign(9, 0); ro.commit.assign(12, 0);
      ro.eval.assign(4, 0); ro.halo_out.assign(halo_len, 0); ro.halo_w.assign(4 * nl, 0);
      // this rank's slices: elements rank, rank + world, ...
      std::vector<uint64_t> sgx(4 * nl), sgy(4 * nl), sc(4 * nl, 0), sc2(4 * nl, 0);
      for (size_t j = 0; j < nl; j++) {
        const size_t i = rank + world * j;
        memcpy(&sgx[4 * j], gx + 4 * i, 32); memcpy(&sgy[4 * j], gy + 4 * i, 32);
        if (i < n_coeffs) memcpy(&sc[4 * j], coeffs + 4 * i, 32);
        // the reversed vector (n_coeffs elements) zero-padded: element i is coeffs[n_coeffs - 1 - i]
        if (i < n_coeffs) memcpy(&sc2[4 * j], coeffs + 4 * (n_coeffs - 1 - i), 32);
      }
      bp_ctx* ctx = nullptr;
      if (bp_init(first_device + (int)(rank % (size_t)n_devices), &ctx) != BP_OK) { ro.rc = BP_E_HIP; hub.abort(); return; }
      LocalExchange ex(&hub, rank);
      SpolyOut o{ro.l.data(), ro.r.data(), ro.ar.data(), ro.g.data(), ro.commit.data(), ro.eval.data(), ro.halo_out.data(),
                 ro.halo_w.data(), &ro.prove_ms, &ro.verify_ms, &ro.ok_verify, &ro.ok_lazy, &ro.ok_halo};
      ro.rc = spoly_dispatch(curve, [&](int c) {
        if (c == BP_SECP256K1)
          return spoly_roundtrip<Secp256k1Fq, Secp256k1Fr, BP_SECP256K1>(ctx, ex, sgx.data(), sgy.data(), h_gen_xy, sc.data(),
                                                                        halo ? sc2.data() : nullptr, n, r_x, seed_rand, seed_chal,
                                                                        tamper, o);
        if (c == BP_VESTA)
          return spoly_roundtrip<VestaFq, VestaFr, BP_VESTA>(ctx, ex, sgx.data(), sgy.data(), h_gen_xy, sc.data(),
                                                            halo ? sc2.data() : nullptr, n, r_x, seed_rand, seed_chal, tamper, o);
        return (int)BP_E_CURVE;
      });
      if (ro.rc != BP_OK) hub.abort();
      bp_shutdown(ctx);
    });
  }
  for (auto& t : threads) t.join();
  for (size_t rank = 0; rank < world; rank++)
    if (outs[rank].rc != BP_OK) return outs[rank].rc;
  const RankOut& z = outs[0];
  *prove_ms = *verify_ms = 0;
  for (size_t rank = 0; rank < world; rank++) {
    const RankOut& ro = outs[rank];
    if (ro.l != z.l || ro.r != z.r || ro.ar != z.ar || ro.g != z.g || ro.commit != z.commit || ro.eval != z.eval ||
        ro.halo_out != z.halo_out || ro.ok_verify != z.ok_verify || ro.ok_lazy != z.ok_lazy || ro.ok_halo != z.ok_halo)
      return BP_E_STATE;  // the ranks disagree
    if (ro.prove_ms > *prove_ms) *prove_ms = ro.prove_ms;
    if (ro.verify_ms > *verify_ms) *verify_ms = ro.verify_ms;
    if (halo_w)
      for (size_t j = 0; j < nl; j++) memcpy(halo_w + 4 * (rank + world * j), &ro.halo_w[4 * j], 32);
  }
  memcpy(out_l, z.l.data(), 96 * k); memcpy(out_r, z.r.data(), 96 * k);
  memcpy(out_ar, z.ar.data(), 64); memcpy(out_g, z.g.data(), 72); memcpy(out_commit, z.commit.data(), 96);
  memcpy(out_eval, z.eval.data(), 32);
  if (halo_out) memcpy(halo_out, z.halo_out.data(), 8 * halo_len);
  *ok_verify = z.ok_verify; *ok_lazy = z.ok_lazy; *ok_halo = z.ok_halo;
  return BP_OK;
}

}  // extern "C"

// ---- sums of a handful of results on the host ----------------------------------------------------
extern "C" {

// out = sum of `count` group results (normalised Jacobian, 12 u64 each, as every entry point of the
// library returns them), normalised: the per-rank shares of a sharded round's L and R are added here
// rather than by a launch chain on the device.  No device call.
int bph_points_sum(int curve, const uint64_t* xyz, size_t count, uint64_t out[12]) {
  if (!out || (count && !xyz)) return BP_E_ARG;
  auto run = [&](auto fq, auto fr) {
    using Fq = decltype(fq);
    using Fr = decltype(fr);
    HXyzz acc = hx_identity<Fq>();
    for (size_t i = 0; i < count; i++) {
      Projective p;
      memcpy(p.xyz, xyz + 12 * i, 96);
      acc = hx_add<Fq>(acc, PointOps<Fq, Fr>::lift(p));
    }
    hx_to_xyz<Fq>(acc, out);
  };
  if (curve == BP_SECP256K1) run(Secp256k1Fq{}, Secp256k1Fr{});
  else if (curve == BP_VESTA) run(VestaFq{}, VestaFr{});
  else return BP_E_CURVE;
  return BP_OK;
}

// out = sum of `count` scalars (Montgomery limbs)
int bph_scalars_sum(int curve, const uint64_t* s, size_t count, uint64_t out[4]) {
  if (!out || (count && !s)) return BP_E_ARG;
  if (curve != BP_SECP256K1 && curve != BP_VESTA) return BP_E_CURVE;
  Scalar acc{{0, 0, 0, 0}};
  for (size_t i = 0; i < count; i++) {
    Scalar v;
    memcpy(v.l, s + 4 * i, 32);
    acc = curve == BP_SECP256K1 ? ScalarOps<Secp256k1Fr>::add(acc, v) : ScalarOps<VestaFr>::add(acc, v);
  }
  memcpy(out, acc.l, 32);
  return BP_OK;
}

}  // extern "C"
