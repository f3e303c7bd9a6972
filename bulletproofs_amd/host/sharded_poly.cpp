// sharded_poly: the polynomial commitment of src/poly_commit (commit, open, verify, lazy_verify, fold of
// two deferred checks and the opening of the folded statement) with its vectors spread over several
// GPU contexts, driven through the C entry points of libbp_host.so / libbp_gpu.so alone (no Python, no
// torch): the shape a native host uses for BASELINE's "poly commitment ... on 8 GPUs".
//
//   sharded_poly [log2 n = 16] [ranks = 2]
//
// Rank r keeps the generators and coefficients i = r (mod ranks), lives on HIP device r mod (number of
// devices) with a context of its own -- on a one-GPU machine the ranks share the device, on a node every
// rank has its own -- and runs as a thread; a round's two points and, once, the last `ranks` elements
// cross between the ranks through host memory (bp_host_sharded.hpp).  The result is compared, message by
// message, with the same flow on one context.  Exit status 0 = equal and every proof verified.
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/bp_gpu.h"

extern "C" {
int bph_poly_roundtrip(bp_ctx* ctx, int curve, const uint64_t* gx, const uint64_t* gy, const uint64_t* h_gen_xy,
                       const uint64_t* coeffs, size_t n_coeffs, size_t n, const uint64_t* r_x, uint64_t seed_rand,
                       uint64_t seed_chal, int tamper, int halo, uint64_t* out_l, uint64_t* out_r, uint64_t* out_ar,
                       uint64_t* out_g, uint64_t* out_commit, uint64_t* out_eval, double* prove_ms, double* verify_ms,
                       int* ok_verify, int* ok_lazy, int* ok_halo, uint64_t* halo_out, uint64_t* halo_w);
int bph_spoly_roundtrip_threads(int first_device, int n_devices, size_t world, int curve, const uint64_t* gx, const uint64_t* gy,
                                const uint64_t* h_gen_xy, const uint64_t* coeffs, size_t n_coeffs, size_t n, const uint64_t* r_x,
                                uint64_t seed_rand, uint64_t seed_chal, int tamper, int halo, uint64_t* out_l, uint64_t* out_r,
                                uint64_t* out_ar, uint64_t* out_g, uint64_t* out_commit, uint64_t* out_eval, double* prove_ms,
                                double* verify_ms, int* ok_verify, int* ok_lazy, int* ok_halo, uint64_t* halo_out, uint64_t* halo_w);
}

namespace {

// secp256k1 generator, Montgomery form (x * 2^256 mod p), little-endian u64 limbs
const uint64_t kGx[4] = {0xd7362e5a487e2097ull, 0x231e295329bc66dbull, 0x979f48c033fd129cull, 0x9981e643e9089f48ull};
const uint64_t kGy[4] = {0xb15ea6d2d3dbabe2ull, 0x8dfc5d5d1f1dc64dull, 0x70b6b59aac19c136ull, 0xcf3f851fd4a582d6ull};

uint64_t splitmix(uint64_t& s) {
  uint64_t z = (s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

// any value below 2^255 is below the group order: a valid scalar in memory form
void random_scalars(std::vector<uint64_t>& v, size_t n, uint64_t seed) {
  v.resize(n * 4);
  for (size_t i = 0; i < n; i++) {
    for (int k = 0; k < 4; k++) v[i * 4 + k] = splitmix(seed);
    v[i * 4 + 3] &= (1ull << 63) - 1;
  }
}

struct Flow {
  std::vector<uint64_t> l, r, ar, g, commit, eval, halo, w;
  double prove_ms = 0, verify_ms = 0;
  int ok_verify = 0, ok_lazy = 0, ok_halo = 0;
  explicit Flow(size_t k, size_t n) : l(12 * k), r(12 * k), ar(8), g(9), commit(12), eval(4), halo(48 * k + 65), w(4 * n) {}
  bool same(const Flow& o) const {
    return l == o.l && r == o.r && ar == o.ar && g == o.g && commit == o.commit && eval == o.eval && halo == o.halo && w == o.w;
  }
  bool all_ok() const { return ok_verify && ok_lazy && ok_halo; }
};

}  // namespace

int main(int argc, char** argv) {
  const size_t k = argc > 1 ? strtoull(argv[1], nullptr, 0) : 16;
  const size_t world = argc > 2 ? strtoull(argv[2], nullptr, 0) : 2;
  if (k < 1 || k > 22 || world < 1 || world > 64 || (world & (world - 1)) || ((size_t)1 << k) < world) {
    fprintf(stderr, "usage: sharded_poly [log2 n: 1..22] [ranks: a power of two <= min(64, n)]\n");
    return 2;
  }
  const size_t n = (size_t)1 << k;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) {
    fprintf(stderr, "no HIP device\n");
    return 2;
  }
  bp_ctx* ctx = nullptr;
  if (bp_init(0, &ctx) != BP_OK) {
    fprintf(stderr, "bp_init failed\n");
    return 2;
  }
  // CRS: n + 1 distinct multiples of the generator made by the library itself (gs, then h); the group generator
  std::vector<uint64_t> gx((n + 1) * 4), gy((n + 1) * 4), px((n + 1) * 4), py((n + 1) * 4), mult, coeffs, rx;
  std::vector<uint8_t> pinf(n + 1);
  for (size_t i = 0; i <= n; i++) {
    memcpy(&gx[i * 4], kGx, 32);
    memcpy(&gy[i * 4], kGy, 32);
  }
  random_scalars(mult, n + 1, 1);
  random_scalars(coeffs, n, 2);
  random_scalars(rx, 2, 3);
  if (bp_scale_points(ctx, BP_SECP256K1, gx.data(), gy.data(), nullptr, mult.data(), n + 1, px.data(), py.data(), pinf.data()) != BP_OK) {
    fprintf(stderr, "bp_scale_points failed: %s\n", bp_last_error(ctx));
    return 2;
  }
  uint64_t hg[16];
  memcpy(hg, &px[4 * n], 32); memcpy(hg + 4, &py[4 * n], 32);
  memcpy(hg + 8, kGx, 32); memcpy(hg + 12, kGy, 32);

  Flow one(k, n), many(k, n);
  int rc = bph_poly_roundtrip(ctx, BP_SECP256K1, px.data(), py.data(), hg, coeffs.data(), n, n, rx.data(), 11, 12, 0, 1, one.l.data(),
                              one.r.data(), one.ar.data(), one.g.data(), one.commit.data(), one.eval.data(), &one.prove_ms,
                              &one.verify_ms, &one.ok_verify, &one.ok_lazy, &one.ok_halo, one.halo.data(), one.w.data());
  if (rc != BP_OK) {
    fprintf(stderr, "bph_poly_roundtrip failed: %d (%s)\n", rc, bp_last_error(ctx));
    return 2;
  }
  rc = bph_spoly_roundtrip_threads(0, n_dev, world, BP_SECP256K1, px.data(), py.data(), hg, coeffs.data(), n, n, rx.data(), 11, 12, 0, 1,
                                   many.l.data(), many.r.data(), many.ar.data(), many.g.data(), many.commit.data(), many.eval.data(),
                                   &many.prove_ms, &many.verify_ms, &many.ok_verify, &many.ok_lazy, &many.ok_halo, many.halo.data(),
                                   many.w.data());
  if (rc != BP_OK) {
    fprintf(stderr, "bph_spoly_roundtrip_threads failed: %d\n", rc);
    return 2;
  }
  const bool same = one.same(many), ok = one.all_ok() && many.all_ok();
  printf("{\"log2n\": %zu, \"ranks\": %zu, \"devices\": %d, \"one_context\": {\"prove_ms\": %.3f, \"verify_ms\": %.3f}, "
         "\"sharded\": {\"prove_ms\": %.3f, \"verify_ms\": %.3f}, \"equal\": %s, \"verified\": %s}\n",
         k, world, (size_t)n_dev < world ? n_dev : (int)world, one.prove_ms, one.verify_ms, many.prove_ms, many.verify_ms,
         same ? "true" : "false", ok ? "true" : "false");
  bp_shutdown(ctx);
  return same && ok ? 0 : 1;
}
