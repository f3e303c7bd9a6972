// C entry points over the C++ host mirror (bp_host.hpp) so that the Python parity
// tests can drive the Msm accumulator and the IPA prover/verifier loops.
#include "bp_host.hpp"

using namespace bp;
using namespace bp::host;

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
  std::vector<Msm<Fq, Fr, CURVE>> msms;
  *single_ok = 1;
  for (size_t pi = 0; pi < n_proofs; pi++) {
    auto wa = unpack_s(a + 4 * n * pi, n), wb = unpack_s(b + 4 * n * pi, n);
    Projective p = I::statement(ctx, crs, wa, wb);
    SeededTranscript<Fr> tp(seed + 1000 * pi);
    IpaProof proof = I::prove(ctx, crs, wa, wb, tp);
    if (pi == 0) {
      for (size_t j = 0; j < k; j++) {
        memcpy(out_l + 12 * j, proof.L[j].xyz, 96);
        memcpy(out_r + 12 * j, proof.R[j].xyz, 96);
      }
      memcpy(out_ab, proof.a.l, 32);
      memcpy(out_ab + 4, proof.b.l, 32);
      memcpy(out_p, p.xyz, 96);
    }
    if (tamper && pi == n_proofs - 1) proof.a = ScalarOps<Fr>::add(proof.a, ScalarOps<Fr>::one());
    SeededTranscript<Fr> tv(seed + 1000 * pi);
    auto m = I::verify_aux(ctx, crs, p, n, proof, tv);
    // ipa::verify, src/ipa/mod.rs:196-212
    if (!is_zero(m.execute())) *single_ok = 0;
    msms.push_back(std::move(m));
  }
  SeededTranscript<Fr> ta(seed ^ 0xA1FAull);
  *batch_ok = verify_batch_aux(msms, ta.verifier_scalar()) ? 1 : 0;
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
  } catch (const std::exception&) {
    return BP_E_HIP;
  }
}

}  // extern "C"
