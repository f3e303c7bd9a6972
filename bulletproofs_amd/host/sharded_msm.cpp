// sharded_msm: one multi-scalar multiplication split over several GPU contexts through the C ABI
// of libbp_gpu.so alone (no Python, no torch): the shape a native host uses for the multi-GPU
// form of Msm::execute (src/msm.rs:84-87).
//
//   sharded_msm [n_total] [n_shards]
//
// Shard r owns a contiguous range of the points (sizes differ by at most one) and lives on HIP
// device r mod (number of devices), so on a one-GPU machine the shards share the device and on a
// node every shard has its own.  Each shard registers ITS range of the bases on its device once
// (the CRS is static), computes its window sums with the common width
// bp_msm_shard_window_bits(n_total, n_shards), the window sums (n_windows * 128 bytes per shard)
// are copied into one array on shard 0's device -- the place of the all-gather: hipMemcpy between
// devices here, ncclAllGather between processes -- and shard 0 combines them.  The result is
// checked against bp_msm of the whole input on one context.  Exit status 0 = equal.
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/bp_gpu.h"

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

#define BPCHK(call)                                                                      \
  do {                                                                                   \
    int rc_ = (call);                                                                    \
    if (rc_ != BP_OK) {                                                                  \
      fprintf(stderr, "%s failed: %d (%s)\n", #call, rc_, ctx_err ? bp_last_error(ctx_err) : ""); \
      return 2;                                                                          \
    }                                                                                    \
  } while (0)
#define HIPCHK(call)                                                      \
  do {                                                                    \
    hipError_t e_ = (call);                                               \
    if (e_ != hipSuccess) {                                               \
      fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_));   \
      return 2;                                                           \
    }                                                                     \
  } while (0)

struct Shard {
  int device = 0;
  size_t begin = 0, end = 0;
  bp_ctx* ctx = nullptr;
  bp_bases* bases = nullptr;
  uint64_t* d_windows = nullptr;  // on this shard's device, 64 XYZZ points
  int n_windows = 0;
  int rc = BP_OK;
};

}  // namespace

int main(int argc, char** argv) {
  const size_t n = argc > 1 ? strtoull(argv[1], nullptr, 0) : ((size_t)1 << 19) - 1;
  const int n_shards = argc > 2 ? atoi(argv[2]) : 2;
  if (n_shards < 1 || n_shards > 64) {
    fprintf(stderr, "n_shards must be 1..64\n");
    return 2;
  }
  int n_dev = 0;
  HIPCHK(hipGetDeviceCount(&n_dev));
  if (n_dev < 1) {
    fprintf(stderr, "no HIP device\n");
    return 2;
  }
  bp_ctx* ctx_err = nullptr;

  // distinct valid points: s_i * G with arbitrary scalars, made by the library itself
  bp_ctx* ctx0 = nullptr;
  BPCHK(bp_init(0, &ctx0));
  ctx_err = ctx0;
  std::vector<uint64_t> gx(n * 4), gy(n * 4), px(n * 4), py(n * 4), mult, scalars;
  std::vector<uint8_t> pinf(n);
  for (size_t i = 0; i < n; i++) {
    memcpy(&gx[i * 4], kGx, 32);
    memcpy(&gy[i * 4], kGy, 32);
  }
  random_scalars(mult, n, 1);
  random_scalars(scalars, n, 2);
  BPCHK(bp_scale_points(ctx0, BP_SECP256K1, gx.data(), gy.data(), nullptr, mult.data(), n, px.data(), py.data(), pinf.data()));

  uint64_t expect[12], got[12];
  BPCHK(bp_msm(ctx0, BP_SECP256K1, px.data(), py.data(), pinf.data(), scalars.data(), n, expect));

  const int window_bits = bp_msm_shard_window_bits(n, n_shards);
  if (window_bits < 2) {
    fprintf(stderr, "bp_msm_shard_window_bits failed: %d\n", window_bits);
    return 2;
  }
  // set-up, untimed: one context per shard, its range of the bases registered on its device
  std::vector<Shard> shards(n_shards);
  for (int r = 0; r < n_shards; r++) {
    Shard& s = shards[r];
    s.device = r % n_dev;
    const size_t base = n / n_shards, rem = n % n_shards;
    s.begin = r * base + ((size_t)r < rem ? r : rem);
    s.end = s.begin + base + ((size_t)r < rem ? 1 : 0);
    BPCHK(bp_init(s.device, &s.ctx));
    ctx_err = s.ctx;
    BPCHK(bp_bases_register(s.ctx, BP_SECP256K1, px.data() + s.begin * 4, py.data() + s.begin * 4, pinf.data() + s.begin,
                            s.end - s.begin, &s.bases));
    HIPCHK(hipSetDevice(s.device));
    HIPCHK(hipMalloc((void**)&s.d_windows, 64 * 128));
  }
  uint64_t* d_gathered = nullptr;
  HIPCHK(hipSetDevice(shards[0].device));
  HIPCHK(hipMalloc((void**)&d_gathered, (size_t)n_shards * 64 * 128));

  double best_ms = 1e30;
  for (int rep = 0; rep < 3; rep++) {
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> threads;
    for (Shard& s : shards)
      threads.emplace_back([&s, &scalars, window_bits] {
        s.rc = bp_msm_shard_windows_resident(s.ctx, s.bases, 0, s.end - s.begin, scalars.data() + s.begin * 4, window_bits,
                                             s.d_windows, &s.n_windows);
      });
    for (auto& t : threads) t.join();
    for (Shard& s : shards) {
      ctx_err = s.ctx;
      BPCHK(s.rc);
      if (s.n_windows != shards[0].n_windows) {
        fprintf(stderr, "shards disagree on the number of windows\n");
        return 2;
      }
    }
    const int nw = shards[0].n_windows;
    // the exchange: shard r's window sums to slot r of the part-major array on shard 0's device
    for (int r = 0; r < n_shards; r++)
      HIPCHK(hipMemcpy(d_gathered + (size_t)r * nw * 16, shards[r].d_windows, (size_t)nw * 128, hipMemcpyDefault));
    ctx_err = shards[0].ctx;
    BPCHK(bp_msm_combine_windows_dev(shards[0].ctx, BP_SECP256K1, d_gathered, n_shards, nw, window_bits, got));
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (ms < best_ms) best_ms = ms;
  }
  const bool same = memcmp(expect, got, sizeof(got)) == 0;
  printf("{\"n_total\": %zu, \"shards\": %d, \"devices\": %d, \"window_bits\": %d, \"windows\": %d, \"ms\": %.3f, \"equal\": %s}\n", n,
         n_shards, n_dev < n_shards ? n_dev : n_shards, window_bits, shards[0].n_windows, best_ms, same ? "true" : "false");
  for (Shard& s : shards) {
    (void)hipSetDevice(s.device);
    (void)hipFree(s.d_windows);
    bp_bases_free(s.bases);
    bp_shutdown(s.ctx);
  }
  (void)hipSetDevice(shards[0].device);
  (void)hipFree(d_gathered);
  bp_shutdown(ctx0);
  return same ? 0 : 1;
}
