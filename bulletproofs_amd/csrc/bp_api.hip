// C ABI of libbp_gpu (see include/bp_gpu.h): contexts, workspace, host<->device
// staging, and the IPA round session.  No CPU fallback exists: without a usable
// HIP device bp_init fails.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <utility>
#include <vector>

#include "bp_host_math.h"
#include "bp_internal.h"
#include "bp_range.h"

using namespace bp;

namespace {

enum Slot {
  S_IN0, S_IN1, S_IN2, S_IN3, S_IN4, S_IN5, S_IN6, S_AOS, S_REBASE,
  S_OUT0, S_OUT1, S_OUT2,
  S_KEYS_A, S_KEYS_B, S_VALS_A, S_VALS_B, S_SORT_TMP, S_BSTART, S_BEND, S_BUCKETS, S_SEGS,
  S_WINDOWS, S_TABLE, S_SMALL, S_NCH, S_CHOFF, S_SCAN_TMP, S_CHUNK_SUMS, S_HEAVY, S_CKEY_A, S_CKEY_B, S_CVAL_A, S_CVAL_B, S_DESC_B, S_DESC_S, S_CSORT_TMP, S_SEG_GROUPS, S_PACKED, S_RANGE, S_COUNT
};

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

}  // namespace

// One helper thread per context, started on first use: the host ends of the two sums of an IPA
// round (some 250 sequential doublings and a field inversion each) run side by side.
struct HostWorker {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::function<void()> job;
  bool pending = false, quit = false;

  void run() {
    std::unique_lock<std::mutex> lk(m);
    for (;;) {
      cv.wait(lk, [this] { return pending || quit; });
      if (quit) return;
      lk.unlock();
      job();
      lk.lock();
      pending = false;
      cv.notify_all();
    }
  }
  // false: no thread could be started (the caller then runs the job itself)
  bool submit(std::function<void()> f) {
    std::unique_lock<std::mutex> lk(m);
    if (!th.joinable()) {
      try {
        th = std::thread([this] { run(); });
      } catch (...) {
        return false;
      }
    }
    job = std::move(f);
    pending = true;
    cv.notify_all();
    return true;
  }
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [this] { return !pending; });
  }
  ~HostWorker() {
    {
      std::unique_lock<std::mutex> lk(m);
      quit = true;
      cv.notify_all();
    }
    if (th.joinable()) th.join();
  }
};

struct bp_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::mutex mu;
  std::string err;
  DevBuf buf[S_COUNT];
  void* pinned = nullptr;
  size_t pinned_cap = 0;
  // 64 slots of 64 bytes of pinned host memory, one per IPA session while it lives: the launch that
  // takes a round's two inner products writes them there as well, so no copy command follows the MSM
  void* host_slots = nullptr;
  uint64_t host_slots_free = ~0ull;
  // pinned ring for the small host arguments of calls that return before their launches have run
  // (stage_h2d): the caller's buffers are free on return whatever the runtime does with pageable memory
  char* stage = nullptr;
  size_t stage_off = 0;
  hipEvent_t ev[6] = {};
  int c_override = 0;
  float last_ms[3] = {0, 0, 0};
  void* ipa_cache = nullptr;  // device block of the last freed IPA session, reused by the next
  size_t ipa_cache_bytes = 0;
  MsmPipe pipe;         // second and third stream of large MSMs
  int last_stages = 1;  // batches of the last MSM (its accumulation time is the sum over pipe.cs0/cs1)
  HostWorker worker;
  bool timing = true;  // bp_set_timing
  // vector polynomials left on the device by bp_range_agg_poly for bp_range_agg_eval (S_RANGE)
  size_t range_n = 0;
  int range_curve = -1;
  // this context's slice of its device's copy of the file-scope array of bp_inner_product_dev_noalloc
  // (bp_init; null when the address could not be taken or every slice of the device is in use)
  u32* static_scratch = nullptr;
  int noalloc_slot = -1;
};

// Scratch of bp_inner_product_dev_noalloc: the per-block partial sums (kDotBlocks field elements,
// 32 KiB) of up to kNoallocSlots contexts per GPU, one slice each (1 MiB in all).  Every GPU the
// library runs on holds its own copy at its own address until the process ends; host code reaches
// it only through the pointer bp_init obtains with hipGetSymbolAddress while that GPU is current
// (the name itself is a host-side stand-in, not a device address).  A context owns its slice from
// bp_init to bp_shutdown, so calls on different contexts of one GPU never share partial sums;
// calls on one context are ordered by its lock and its stream.
constexpr int kNoallocSlots = 32;
constexpr int kNoallocMaxDevices = 64;
static __device__ __align__(256) u32 g_noalloc_dot_partials[(size_t)kNoallocSlots * kDotBlocks * 8];
static std::mutex g_noalloc_mu;
static uint32_t g_noalloc_used[kNoallocMaxDevices];  // per device: bit s set = slice s is owned by a context

struct bp_bases {
  bp_ctx* ctx;
  int curve;
  size_t n;
  u32* x;
  u32* y;
  uint8_t* inf;
  u32* tab = nullptr;  // window tables (bp_bases_precompute): kTableWindows * n records of 64 bytes
};

// Accumulator of bp_batch_*: one scalar per point of a registered base set, on the device
struct bp_batch {
  bp_ctx* ctx;
  const bp_bases* bases;
  size_t n;
  u32* acc;
};

// size of a context's pinned staging ring (stage_h2d)
constexpr size_t kStageBytes = (size_t)1 << 20;

// sessions over at least this many generators replace them by the folded ones every few rounds
constexpr size_t kIpaRebaseMinPoints = (size_t)1 << 15;

// MSMs over at most this many points use the window tables of a base set that has them;
// beyond, the 16-bit-window pipeline does half the additions and wins
constexpr size_t kTableMaxPoints = (size_t)1 << 17;

// The prover never folds the generator vectors.  It keeps the vectors it was given and, per
// generator, the product of the challenges that folding would have multiplied it by (cg, ch);
// the cross terms L, R of a round are then MSMs over the original generators with scalars
// cg[k]*a[..] / ch[k]*b[..].  Folded generators exist only when bp_ipa_read asks for them.
struct bp_ipa {
  bp_ctx* ctx = nullptr;
  int curve = 0;
  int variant = 0;
  size_t n = 0;   // current length of a, b
  size_t n0 = 0;  // length at bp_ipa_begin
  // n0 points each, as given: inside `block`, or inside a registered base set
  const u32 *gx = nullptr, *gy = nullptr, *hx = nullptr, *hy = nullptr;
  const uint8_t *ginf = nullptr, *hinf = nullptr;
  void* block = nullptr;             // one allocation behind every pointer below up to `dots`
  size_t block_bytes = 0;
  u32 *a = nullptr, *b = nullptr;    // current vectors (folded in place, or out of place into a2, b2 and swapped)
  u32 *a2 = nullptr, *b2 = nullptr;  // n0/2 elements each: the other side of the out-of-place fold
  bool prepped = false;              // the last fold already wrote this round's dots and MSM scalars
  bool h_scaled = false;             // bp_ipa_scale_h was applied: the H coefficients differ from column to column
  void* rebase_block = nullptr;      // folded generators that replaced the original ones (owned)
  u32 *cg = nullptr, *ch = nullptr;  // n0 coefficients each
  u32 *eg = nullptr, *eh = nullptr;  // n0 cross-term scalars each
  u32* u_xy = nullptr;               // the point u (x|y) when has_u
  bool has_u = false;
  // u given as factor * bases[u_index] (bp_ipa_set_u_resident): its MSM scalars are
  // factor * c_l, factor * c_r and it can be addressed through the base set's window tables
  bool u_scaled = false;
  size_t u_index = 0;
  FeArg u_factor{};
  const bp_bases* res = nullptr;     // generators live in this base set (not owned)
  size_t g_off = 0, h_off = 0;
  u32* dots = nullptr;               // partial sums and results of the two inner products
  int host_slot = -1;                // slot of ctx->host_slots that receives c_l, c_r (-1: none free, copy instead)
  bool dots_on_host = false;         // the launch that prepared this round wrote c_l, c_r to the slot
  // folded generators, materialised on demand (n0/2 points each)
  u32 *fgx = nullptr, *fgy = nullptr, *fhx = nullptr, *fhy = nullptr;
  uint8_t *fginf = nullptr, *fhinf = nullptr;
  std::vector<FeArg> x_canon, xinv_canon;  // challenges so far, canonical integers
};

namespace {

int fail(bp_ctx* ctx, int code, const char* what, hipError_t e = hipSuccess) {
  if (ctx) {
    char tmp[512];
    if (e != hipSuccess)
      snprintf(tmp, sizeof tmp, "%s: %s", what, hipGetErrorString(e));
    else
      snprintf(tmp, sizeof tmp, "%s", what);
    ctx->err = tmp;
  }
  return code;
}

#define HIPCHK(ctx, call)                                     \
  do {                                                        \
    hipError_t _e = (call);                                   \
    if (_e != hipSuccess) {                                   \
      int _c = (_e == hipErrorOutOfMemory) ? BP_E_OOM : BP_E_HIP; \
      return fail(ctx, _c, #call, _e);                        \
    }                                                         \
  } while (0)

#define CHK(call)              \
  do {                         \
    int _r = (call);           \
    if (_r != BP_OK) return _r; \
  } while (0)

struct Guard {
  std::unique_lock<std::mutex> lk;
  explicit Guard(bp_ctx* c) : lk(c->mu) { (void)hipSetDevice(c->device); }
};

int reserve(bp_ctx* ctx, int slot, size_t bytes, void** out) {
  DevBuf& b = ctx->buf[slot];
  if (bytes < 256) bytes = 256;
  if (b.cap < bytes) {
    if (b.p) {
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      HIPCHK(ctx, hipFree(b.p));
      b.p = nullptr;
      b.cap = 0;
    }
    size_t want = bytes + bytes / 8;
    want = (want + 4095) & ~(size_t)4095;
    HIPCHK(ctx, hipMalloc(&b.p, want));
    b.cap = want;
  }
  *out = b.p;
  return BP_OK;
}

int reserve_pinned(bp_ctx* ctx, size_t bytes) {
  if (ctx->pinned_cap >= bytes) return BP_OK;
  if (ctx->pinned) {
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipHostFree(ctx->pinned));
    ctx->pinned = nullptr;
    ctx->pinned_cap = 0;
  }
  size_t want = (bytes + 65535) & ~(size_t)65535;
  HIPCHK(ctx, hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault));
  ctx->pinned_cap = want;
  return BP_OK;
}

// Host bytes to the device through the context's pinned ring: copied into the ring now, from the ring
// to `dst` in stream order.  The ring is reused only after the stream has drained, so a call may
// return before its launches have run and its host arguments (locals included) may go out of scope.
int stage_h2d(bp_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return BP_OK;
  const size_t al = (bytes + 63) & ~(size_t)63;
  if (al > kStageBytes) {  // too large for the ring: copy and wait
    HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return BP_OK;
  }
  if (ctx->stage_off + al > kStageBytes) {
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stage_off = 0;
  }
  memcpy(ctx->stage + ctx->stage_off, src, bytes);
  HIPCHK(ctx, hipMemcpyAsync(dst, ctx->stage + ctx->stage_off, bytes, hipMemcpyHostToDevice, ctx->stream));
  ctx->stage_off += al;
  return BP_OK;
}

bool curve_ok(int curve) { return curve == BP_SECP256K1 || curve == BP_VESTA; }

int upload(bp_ctx* ctx, int slot, const void* host, size_t bytes, void** dev) {
  CHK(reserve(ctx, slot, bytes, dev));
  if (bytes) HIPCHK(ctx, hipMemcpyAsync(*dev, host, bytes, hipMemcpyHostToDevice, ctx->stream));
  return BP_OK;
}

FeArg to_arg(const uint64_t v[4]) {
  FeArg a;
  memcpy(a.v, v, 32);
  return a;
}

FeArg to_canonical_fr(int curve, const uint64_t v[4]) {
  host::HFe m;
  memcpy(m.v, v, 32);
  host::HFe c = curve == BP_SECP256K1 ? host::h_from_mont<Secp256k1Fr>(m) : host::h_from_mont<VestaFr>(m);
  FeArg a;
  memcpy(a.v, c.v, 32);
  return a;
}

// zpow[j] = z^(2+j), j < m; pow2[i] = 2^i, i < n_bits (Montgomery form, 4 u64 each)
template <class Fr>
void range_tables(const uint64_t z[4], size_t m, unsigned n_bits, uint64_t* zpow, uint64_t* pow2) {
  host::HFe zz, zp, p2 = host::h_one<Fr>();
  memcpy(zz.v, z, 32);
  zp = host::h_mul<Fr>(zz, zz);
  for (size_t j = 0; j < m; j++) {
    memcpy(zpow + 4 * j, zp.v, 32);
    zp = host::h_mul<Fr>(zp, zz);
  }
  for (unsigned i = 0; i < n_bits; i++) {
    memcpy(pow2 + 4 * i, p2.v, 32);
    p2 = host::h_dbl<Fr>(p2);
  }
}

// sum_w 2^(c*w) win[w] on the host (c = 0: the plain sum), normalised: the sequential tail of every MSM
void finish_windows(int curve, const uint64_t* win, int n_windows, int c, uint64_t out_xyz[12]) {
  if (curve == BP_SECP256K1)
    host::hj_combine_windows<host::SecpPlainOps>(win, n_windows, c, out_xyz);
  else
    host::hj_combine_windows<host::MontOps<VestaFq>>(win, n_windows, c, out_xyz);
}

// device phase of an MSM: leaves plan->n_windows XYZZ window sums in S_WINDOWS or, with win_host, in
// the context's pinned host buffer (the last launch writes them there itself: they are only ever
// written, and the caller that wants them on the host needs no copy command after the launches;
// room for `host_extra` more bytes is kept behind them)
int msm_device(bp_ctx* ctx, int curve, const MsmPart& p0, const MsmPart& p1, MsmPlan* plan,
               u32** d_windows, int groups = 1, const MsmExtra* extra = nullptr, const MsmTable* table = nullptr,
               bool win_host = false, size_t host_extra = 0, int batch = 1) {
  size_t n = p0.n + p1.n + (extra ? 2 : 0);
  hipError_t e = msm_make_plan(curve, n, ctx->c_override, groups, plan, table ? table->stride : 0, true, batch, table ? table->bits : 8);
  if (e != hipSuccess) return fail(ctx, BP_E_ARG, "msm plan (n too large?)", e);
  MsmWorkspace ws{};
  void* p;
  CHK(reserve(ctx, S_PACKED, plan->packed_bytes, &p)); ws.packed = (u32*)p;
  CHK(reserve(ctx, S_KEYS_A, plan->keys_bytes, &p)); ws.keys_a = (u32*)p;
  CHK(reserve(ctx, S_KEYS_B, plan->keys_bytes, &p)); ws.keys_b = (u32*)p;
  CHK(reserve(ctx, S_VALS_A, plan->keys_bytes, &p)); ws.vals_a = (u32*)p;
  CHK(reserve(ctx, S_VALS_B, plan->keys_bytes, &p)); ws.vals_b = (u32*)p;
  CHK(reserve(ctx, S_SORT_TMP, plan->sort_tmp_bytes, &p)); ws.sort_tmp = p;
  CHK(reserve(ctx, S_BSTART, plan->bounds_bytes, &p)); ws.bucket_start = (u32*)p;
  CHK(reserve(ctx, S_BEND, plan->bounds_bytes, &p)); ws.bucket_end = (u32*)p;
  CHK(reserve(ctx, S_BUCKETS, plan->bucket_bytes, &p)); ws.buckets = (u32*)p;
  CHK(reserve(ctx, S_NCH, plan->nch_bytes, &p)); ws.nch = (u32*)p;
  CHK(reserve(ctx, S_CHOFF, plan->nch_bytes, &p)); ws.choff = (u32*)p;
  CHK(reserve(ctx, S_SCAN_TMP, plan->scan_tmp_bytes, &p)); ws.scan_tmp = p;
  CHK(reserve(ctx, S_CHUNK_SUMS, plan->chunk_sum_bytes, &p)); ws.chunk_sums = (u32*)p;
  CHK(reserve(ctx, S_CKEY_A, plan->cdesc_bytes, &p)); ws.ckey_a = (u32*)p;
  CHK(reserve(ctx, S_CKEY_B, plan->cdesc_bytes, &p)); ws.ckey_b = (u32*)p;
  CHK(reserve(ctx, S_CVAL_A, plan->cdesc_bytes, &p)); ws.cval_a = (u32*)p;
  CHK(reserve(ctx, S_CVAL_B, plan->cdesc_bytes, &p)); ws.cval_b = (u32*)p;
  CHK(reserve(ctx, S_DESC_B, plan->cdesc_bytes, &p)); ws.desc_b = (u32*)p;
  CHK(reserve(ctx, S_DESC_S, plan->cdesc_bytes, &p)); ws.desc_s = (u32*)p;
  CHK(reserve(ctx, S_CSORT_TMP, plan->csort_tmp_bytes, &p)); ws.csort_tmp = p;
  CHK(reserve(ctx, S_HEAVY, plan->nch_bytes + 256, &p)); ws.heavy = (u32*)p + 64; ws.heavy_count = (u32*)p;
  CHK(reserve(ctx, S_SEGS, plan->seg_bytes, &p)); ws.segs = (u32*)p;
  CHK(reserve(ctx, S_SEG_GROUPS, plan->seg_group_bytes, &p)); ws.seg_groups = (u32*)p;
  if (win_host) {
    CHK(reserve_pinned(ctx, plan->win_bytes + host_extra));
    ws.windows = (u32*)ctx->pinned;
  } else {
    CHK(reserve(ctx, S_WINDOWS, plan->win_bytes, &p)); ws.windows = (u32*)p;
  }
  // four event markers per MSM cost a few microseconds each on the stream: only with timing on
  const bool timed = ctx->timing || plan->stages > 1;
  if (timed) HIPCHK(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
  ctx->last_stages = plan->stages;
  e = msm_run(curve, p0, p1, extra, *plan, ws, ctx->stream, timed ? ctx->ev[2] : nullptr, timed ? ctx->ev[3] : nullptr,
              table, &ctx->pipe);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "msm_run", e);
  if (timed) HIPCHK(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
  *d_windows = ws.windows;
  return BP_OK;
}

void msm_collect_times(bp_ctx* ctx, bool had_points) {
  if (!ctx->timing && ctx->last_stages <= 1) return;
  float ms = 0;
  if (hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]) == hipSuccess) ctx->last_ms[1] = ms;
  if (!had_points) return;
  if (ctx->last_stages <= 1) {
    if (hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]) == hipSuccess) ctx->last_ms[0] = ms;
    return;
  }
  float total = 0;
  for (int g = 0; g < ctx->last_stages; g++)
    if (hipEventElapsedTime(&ms, ctx->pipe.cs0[g], ctx->pipe.cs1[g]) == hipSuccess) total += ms;
  ctx->last_ms[0] = total;
}

int msm_full(bp_ctx* ctx, int curve, const MsmPart& p0, const MsmPart& p1, uint64_t out_xyz[12],
             const MsmTable* table = nullptr) {
  MsmPlan plan;
  u32* d_win = nullptr;
  CHK(msm_device(ctx, curve, p0, p1, &plan, &d_win, 1, nullptr, table, true));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  msm_collect_times(ctx, plan.n_total != 0);
  finish_windows(curve, (const uint64_t*)ctx->pinned, plan.n_windows, plan.finish_c, out_xyz);
  return BP_OK;
}

int dot_dev(bp_ctx* ctx, int curve, const u32* a, const u32* b, size_t n, uint64_t out[4]) {
  void* part;
  CHK(reserve(ctx, S_SMALL, (size_t)(kDotBlocks + 1) * 32, &part));
  u32* partials = (u32*)part;
  u32* d_out = partials + (size_t)kDotBlocks * 8;
  HIPCHK(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
  hipError_t e = launch_inner_product(curve, a, b, n, partials, d_out, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "inner_product", e);
  HIPCHK(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
  CHK(reserve_pinned(ctx, 32));
  HIPCHK(ctx, hipMemcpyAsync(ctx->pinned, d_out, 32, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(out, ctx->pinned, 32);
  float ms;
  if (hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5]) == hipSuccess) ctx->last_ms[2] = ms;
  return BP_OK;
}

}  // namespace

extern "C" {

int bp_init(int device, bp_ctx** out) {
  if (!out) return BP_E_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return BP_E_HIP;
  if (device < 0 || device >= count) return BP_E_ARG;
  if (hipSetDevice(device) != hipSuccess) return BP_E_HIP;
  bp_ctx* c = new (std::nothrow) bp_ctx();
  if (!c) return BP_E_OOM;
  c->device = device;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return BP_E_HIP;
  }
  for (auto& e : c->ev) {
    if (hipEventCreate(&e) != hipSuccess) {
      bp_shutdown(c);
      return BP_E_HIP;
    }
  }
  {
    MsmPipe& p = c->pipe;
    bool ok = hipStreamCreateWithFlags(&p.front, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&p.reduce, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&p.start, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&p.done, hipEventDisableTiming) == hipSuccess;
    for (int g = 0; ok && g < kMsmMaxStages; g++)
      ok = hipEventCreateWithFlags(&p.sorted[g], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&p.summed[g], hipEventDisableTiming) == hipSuccess &&
           hipEventCreate(&p.cs0[g]) == hipSuccess && hipEventCreate(&p.cs1[g]) == hipSuccess;
    if (!ok) {
      bp_shutdown(c);
      return BP_E_HIP;
    }
  }
  if (hipHostMalloc(&c->host_slots, 64 * 64, hipHostMallocDefault) != hipSuccess) {
    c->host_slots = nullptr;
    bp_shutdown(c);
    return BP_E_HIP;
  }
  if (hipHostMalloc((void**)&c->stage, kStageBytes, hipHostMallocDefault) != hipSuccess) {
    c->stage = nullptr;
    bp_shutdown(c);
    return BP_E_HIP;
  }
  // address of this device's copy of the static array (the device is current: hipSetDevice above) and
  // a free slice of it.  Neither failing makes the library unusable: only the no-allocation entry
  // point then reports BP_E_STATE on this context.
  if (device < kNoallocMaxDevices) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_noalloc_dot_partials)) == hipSuccess && p) {
      std::lock_guard<std::mutex> lk(g_noalloc_mu);
      for (int s = 0; s < kNoallocSlots; s++)
        if (!(g_noalloc_used[device] >> s & 1u)) {
          g_noalloc_used[device] |= 1u << s;
          c->noalloc_slot = s;
          c->static_scratch = (u32*)p + (size_t)s * kDotBlocks * 8;
          break;
        }
    } else {
      (void)hipGetLastError();  // the failure is this entry point's alone
    }
  }
  *out = c;
  return BP_OK;
}

void bp_shutdown(bp_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  for (auto& b : ctx->buf)
    if (b.p) (void)hipFree(b.p);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  if (ctx->host_slots) (void)hipHostFree(ctx->host_slots);
  if (ctx->stage) (void)hipHostFree(ctx->stage);
  if (ctx->ipa_cache) (void)hipFree(ctx->ipa_cache);
  for (auto& e : ctx->ev)
    if (e) (void)hipEventDestroy(e);
  {
    MsmPipe& p = ctx->pipe;
    hipEvent_t singles[2] = {p.start, p.done};
    for (hipEvent_t e : singles)
      if (e) (void)hipEventDestroy(e);
    for (int g = 0; g < kMsmMaxStages; g++) {
      hipEvent_t evs[4] = {p.sorted[g], p.summed[g], p.cs0[g], p.cs1[g]};
      for (hipEvent_t e : evs)
        if (e) (void)hipEventDestroy(e);
    }
    if (p.front) (void)hipStreamDestroy(p.front);
    if (p.reduce) (void)hipStreamDestroy(p.reduce);
  }
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  if (ctx->noalloc_slot >= 0) {
    std::lock_guard<std::mutex> lk(g_noalloc_mu);
    g_noalloc_used[ctx->device] &= ~(1u << ctx->noalloc_slot);
  }
  delete ctx;
}

const char* bp_last_error(bp_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int bp_sync(bp_ctx* ctx) {
  if (!ctx) return BP_E_ARG;
  Guard g(ctx);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return BP_OK;
}

int bp_set_msm_window(bp_ctx* ctx, int c) {
  if (!ctx || c < 0 || c == 1 || c > 20) return BP_E_ARG;
  Guard g(ctx);
  ctx->c_override = c;
  return BP_OK;
}

// ---- MSM ---------------------------------------------------------------------

int bp_msm_dev(bp_ctx* ctx, int curve, const uint64_t* d_x, const uint64_t* d_y, const uint8_t* d_inf,
               const uint64_t* d_scalars, size_t n, uint64_t out_xyz[12]) {
  if (!ctx || !out_xyz || (n && (!d_x || !d_y || !d_scalars))) return fail(ctx, BP_E_ARG, "bp_msm_dev: null argument");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  Guard g(ctx);
  MsmPart p0{(const u32*)d_x, (const u32*)d_y, d_inf, (const u32*)d_scalars, n};
  MsmPart p1{nullptr, nullptr, nullptr, nullptr, 0};
  return msm_full(ctx, curve, p0, p1, out_xyz);
}

int bp_msm2_dev(bp_ctx* ctx, int curve, const uint64_t* d_x0, const uint64_t* d_y0, const uint8_t* d_inf0,
                const uint64_t* d_s0, size_t n0, const uint64_t* d_x1, const uint64_t* d_y1,
                const uint8_t* d_inf1, const uint64_t* d_s1, size_t n1, uint64_t out_xyz[12]) {
  if (!ctx || !out_xyz || (n0 && (!d_x0 || !d_y0 || !d_s0)) || (n1 && (!d_x1 || !d_y1 || !d_s1)))
    return fail(ctx, BP_E_ARG, "bp_msm2_dev: null argument");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  Guard g(ctx);
  MsmPart p0{(const u32*)d_x0, (const u32*)d_y0, d_inf0, (const u32*)d_s0, n0};
  MsmPart p1{(const u32*)d_x1, (const u32*)d_y1, d_inf1, (const u32*)d_s1, n1};
  return msm_full(ctx, curve, p0, p1, out_xyz);
}

int bp_msm(bp_ctx* ctx, int curve, const uint64_t* bases_x, const uint64_t* bases_y, const uint8_t* bases_inf,
           const uint64_t* scalars, size_t n, uint64_t out_xyz[12]) {
  if (!ctx || !out_xyz || (n && (!bases_x || !bases_y || !scalars))) return fail(ctx, BP_E_ARG, "bp_msm: null argument");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  Guard g(ctx);
  void *dx, *dy, *di = nullptr, *ds;
  CHK(upload(ctx, S_IN0, bases_x, n * 32, &dx));
  CHK(upload(ctx, S_IN1, bases_y, n * 32, &dy));
  if (bases_inf) CHK(upload(ctx, S_IN2, bases_inf, n, &di));
  CHK(upload(ctx, S_IN3, scalars, n * 32, &ds));
  MsmPart p0{(const u32*)dx, (const u32*)dy, (const uint8_t*)di, (const u32*)ds, n};
  MsmPart p1{nullptr, nullptr, nullptr, nullptr, 0};
  return msm_full(ctx, curve, p0, p1, out_xyz);
}

int bp_msm_aos(bp_ctx* ctx, int curve, const void* bases, const uint64_t* scalars, size_t n, uint64_t out_xyz[12]) {
  if (!ctx || !out_xyz || (n && (!bases || !scalars))) return fail(ctx, BP_E_ARG, "bp_msm_aos: null argument");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  Guard g(ctx);
  void *raw, *dx, *dy, *di, *ds;
  CHK(upload(ctx, S_AOS, bases, n * BP_AOS_STRIDE, &raw));
  CHK(reserve(ctx, S_IN0, n * 32, &dx));
  CHK(reserve(ctx, S_IN1, n * 32, &dy));
  CHK(reserve(ctx, S_IN2, n, &di));
  hipError_t e = launch_aos_split((const u32*)raw, n, (u32*)dx, (u32*)dy, (uint8_t*)di, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "aos_split", e);
  CHK(upload(ctx, S_IN3, scalars, n * 32, &ds));
  MsmPart p0{(const u32*)dx, (const u32*)dy, (const uint8_t*)di, (const u32*)ds, n};
  MsmPart p1{nullptr, nullptr, nullptr, nullptr, 0};
  return msm_full(ctx, curve, p0, p1, out_xyz);
}

namespace {
int bases_register_impl(bp_ctx* ctx, int curve, const uint64_t* x, const uint64_t* y, const uint8_t* inf, const void* aos,
                        size_t n, bp_bases** out, const uint8_t* wire = nullptr);
}

int bp_bases_register(bp_ctx* ctx, int curve, const uint64_t* x, const uint64_t* y, const uint8_t* inf, size_t n,
                      bp_bases** out) {
  if (!ctx || !out || (n && (!x || !y))) return fail(ctx, BP_E_ARG, "bp_bases_register: null argument");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  return bases_register_impl(ctx, curve, x, y, inf, nullptr, n, out);
}

int bp_bases_register_aos(bp_ctx* ctx, int curve, const void* bases, size_t n, bp_bases** out) {
  if (!ctx || !out || (n && !bases)) return fail(ctx, BP_E_ARG, "bp_bases_register_aos: null argument");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  return bases_register_impl(ctx, curve, nullptr, nullptr, nullptr, bases, n, out);
}

namespace {
// planes (x, y, inf) or, when aos is given, n records of BP_AOS_STRIDE bytes split on the device
int bases_register_impl(bp_ctx* ctx, int curve, const uint64_t* x, const uint64_t* y, const uint8_t* inf, const void* aos,
                        size_t n, bp_bases** out, const uint8_t* wire) {
  Guard g(ctx);
  bp_bases* b = new (std::nothrow) bp_bases();
  if (!b) return fail(ctx, BP_E_OOM, "host allocation");
  b->ctx = ctx;
  b->curve = curve;
  b->n = n;
  b->x = b->y = nullptr;
  b->inf = nullptr;
  size_t bytes = n ? n * 32 : 32;
  hipError_t e;
  // every exit but the last releases what was allocated so far
  struct Undo {
    bp_bases* b;
    ~Undo() {
      if (!b) return;
      if (b->x) (void)hipFree(b->x);
      if (b->y) (void)hipFree(b->y);
      if (b->inf) (void)hipFree(b->inf);
      delete b;
    }
  } undo{b};
  if ((e = hipMalloc((void**)&b->x, bytes)) != hipSuccess || (e = hipMalloc((void**)&b->y, bytes)) != hipSuccess ||
      (e = hipMalloc((void**)&b->inf, n ? n : 1)) != hipSuccess)
    return fail(ctx, e == hipErrorOutOfMemory ? BP_E_OOM : BP_E_HIP, "bp_bases_register: hipMalloc", e);
  if (n && wire) {
    // compressed bytes: decoded straight into the base set; one flag per point comes back
    void *raw, *dok;
    CHK(upload(ctx, S_AOS, wire, n * BP_POINT_WIRE_BYTES, &raw));
    CHK(reserve(ctx, S_IN1, n, &dok));
    e = launch_points_decompress(curve, (const uint8_t*)raw, n, b->x, b->y, b->inf, (uint8_t*)dok, ctx->stream);
    if (e != hipSuccess) return fail(ctx, BP_E_HIP, "points_decompress", e);
    std::unique_ptr<uint8_t[]> ok(new (std::nothrow) uint8_t[n]);
    if (!ok) return fail(ctx, BP_E_OOM, "host allocation");
    HIPCHK(ctx, hipMemcpyAsync(ok.get(), dok, n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < n; i++)
      if (!ok[i]) {
        char msg[96];
        snprintf(msg, sizeof msg, "bp_bases_register_compressed: bytes of point %zu are no point", i);
        return fail(ctx, BP_E_ARG, msg);
      }
  } else if (n && aos) {
    void* raw;
    CHK(upload(ctx, S_AOS, aos, n * BP_AOS_STRIDE, &raw));
    e = launch_aos_split((const u32*)raw, n, b->x, b->y, b->inf, ctx->stream);
    if (e != hipSuccess) return fail(ctx, BP_E_HIP, "aos_split", e);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  } else if (n) {
    HIPCHK(ctx, hipMemcpyAsync(b->x, x, n * 32, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(b->y, y, n * 32, hipMemcpyHostToDevice, ctx->stream));
    if (inf)
      HIPCHK(ctx, hipMemcpyAsync(b->inf, inf, n, hipMemcpyHostToDevice, ctx->stream));
    else
      HIPCHK(ctx, hipMemsetAsync(b->inf, 0, n, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  }
  undo.b = nullptr;
  *out = b;
  return BP_OK;
}
}  // namespace

int bp_bases_register_compressed(bp_ctx* ctx, int curve, const uint8_t* bytes, size_t n, bp_bases** out) {
  if (!ctx || !out || (n && !bytes)) return fail(ctx, BP_E_ARG, "bp_bases_register_compressed: null argument");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  return bases_register_impl(ctx, curve, nullptr, nullptr, nullptr, nullptr, n, out, bytes);
}

void bp_bases_free(bp_bases* b) {
  if (!b) return;
  Guard g(b->ctx);
  (void)hipStreamSynchronize(b->ctx->stream);
  (void)hipFree(b->x);
  (void)hipFree(b->y);
  (void)hipFree(b->inf);
  if (b->tab) (void)hipFree(b->tab);
  delete b;
}

int bp_bases_precompute(bp_bases* b) {
  if (!b) return BP_E_ARG;
  bp_ctx* ctx = b->ctx;
  Guard g(ctx);
  if (b->tab || b->n == 0) return BP_OK;
  if ((size_t)kTableWindows * b->n >= ((size_t)1 << 30)) return fail(ctx, BP_E_ARG, "bp_bases_precompute: base set too large");
  u32* tab = nullptr;
  HIPCHK(ctx, hipMalloc((void**)&tab, (size_t)kTableWindows * b->n * 64));
  // one inversion per point where the Jacobian chain fits a temporary of at most 1 GiB (128 bytes per record:
  // up to 2^18 points; 8 195 points: 8.4 -> 1.4 ms), one per record beyond
  hipError_t e;
  void* scratch = nullptr;
  const size_t scratch_bytes = (size_t)kTableWindows * b->n * 128;
  if (scratch_bytes <= ((size_t)1 << 30) && hipMalloc(&scratch, scratch_bytes) == hipSuccess) {
    e = launch_window_tables_inv1(b->curve, b->x, b->y, b->inf, b->n, kTableWindowBits, kTableWindows, (u32*)scratch, tab, ctx->stream);
  } else {
    (void)hipGetLastError();
    e = launch_window_tables(b->curve, b->x, b->y, b->inf, b->n, kTableWindowBits, kTableWindows, tab, ctx->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (scratch) (void)hipFree(scratch);
  if (e != hipSuccess) {
    (void)hipFree(tab);
    return fail(ctx, BP_E_HIP, "bp_bases_precompute", e);
  }
  b->tab = tab;
  return BP_OK;
}

size_t bp_bases_len(const bp_bases* b) { return b ? b->n : 0; }

int bp_msm_resident(bp_ctx* ctx, const bp_bases* bases, size_t offset, size_t n, const uint64_t* scalars,
                    uint64_t out_xyz[12]) {
  if (!ctx || !bases || !out_xyz || (n && !scalars)) return fail(ctx, BP_E_ARG, "bp_msm_resident: null argument");
  if (bases->ctx != ctx) return fail(ctx, BP_E_ARG, "bp_msm_resident: bases belong to another context");
  if (offset > bases->n || n > bases->n - offset) return fail(ctx, BP_E_ARG, "bp_msm_resident: range outside bases");
  Guard g(ctx);
  void* ds;
  CHK(upload(ctx, S_IN3, scalars, n * 32, &ds));
  MsmPart p0{bases->x + offset * 8, bases->y + offset * 8, bases->inf + offset, (const u32*)ds, n};
  MsmPart p1{nullptr, nullptr, nullptr, nullptr, 0};
  p0.tab_offset = offset;
  MsmTable table{bases->tab, bases->n};
  return msm_full(ctx, bases->curve, p0, p1, out_xyz, bases->tab && n <= kTableMaxPoints ? &table : nullptr);
}

int bp_msm_resident_pair(bp_ctx* ctx, const bp_bases* bases, size_t offset, size_t n, const uint64_t* scalars_a,
                         const uint64_t* scalars_b, uint64_t out_a_xyz[12], uint64_t out_b_xyz[12]) {
  if (!ctx || !bases || !out_a_xyz || !out_b_xyz || (n && (!scalars_a || !scalars_b)))
    return fail(ctx, BP_E_ARG, "bp_msm_resident_pair: null argument");
  if (bases->ctx != ctx) return fail(ctx, BP_E_ARG, "bp_msm_resident_pair: bases belong to another context");
  if (offset > bases->n || n > bases->n - offset) return fail(ctx, BP_E_ARG, "bp_msm_resident_pair: range outside bases");
  Guard g(ctx);
  void *da, *db;
  CHK(upload(ctx, S_IN3, scalars_a, n * 32, &da));
  CHK(upload(ctx, S_IN4, scalars_b, n * 32, &db));
  // one pass, two sums: the points once with each scalar vector, as groups 0 and 1
  MsmPart p0{bases->x + offset * 8, bases->y + offset * 8, bases->inf + offset, (const u32*)da, n};
  MsmPart p1{bases->x + offset * 8, bases->y + offset * 8, bases->inf + offset, (const u32*)db, n};
  p1.group_lo = p1.group_hi = 1;
  MsmPlan plan;
  u32* d_win = nullptr;
  CHK(msm_device(ctx, bases->curve, p0, p1, &plan, &d_win, 2, nullptr, nullptr, true));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  msm_collect_times(ctx, plan.n_total != 0);
  const int curve = bases->curve, gw = plan.group_windows, c = plan.finish_c;
  const uint64_t* win = (const uint64_t*)ctx->pinned;
  const uint64_t* win_b = win + (size_t)gw * 16;
  const bool helped = ctx->worker.submit([=] { finish_windows(curve, win_b, gw, c, out_b_xyz); });
  finish_windows(curve, win, gw, c, out_a_xyz);
  if (helped)
    ctx->worker.wait();
  else
    finish_windows(curve, win_b, gw, c, out_b_xyz);
  return BP_OK;
}

int bp_msm_resident_plus(bp_ctx* ctx, const bp_bases* bases, size_t offset, size_t n, const uint64_t* scalars,
                         const uint64_t* extra_x, const uint64_t* extra_y, const uint8_t* extra_inf,
                         const uint64_t* extra_scalars, size_t n_extra, uint64_t out_xyz[12]) {
  if (!ctx || !bases || !out_xyz || (n && !scalars) || (n_extra && (!extra_x || !extra_y || !extra_scalars)))
    return fail(ctx, BP_E_ARG, "bp_msm_resident_plus: null argument");
  if (bases->ctx != ctx) return fail(ctx, BP_E_ARG, "bp_msm_resident_plus: bases belong to another context");
  if (offset > bases->n || n > bases->n - offset) return fail(ctx, BP_E_ARG, "bp_msm_resident_plus: range outside bases");
  Guard g(ctx);
  void *ds, *dx = nullptr, *dy = nullptr, *di = nullptr, *des = nullptr;
  CHK(upload(ctx, S_IN3, scalars, n * 32, &ds));
  if (n_extra) {
    CHK(upload(ctx, S_IN0, extra_x, n_extra * 32, &dx));
    CHK(upload(ctx, S_IN1, extra_y, n_extra * 32, &dy));
    if (extra_inf) CHK(upload(ctx, S_IN2, extra_inf, n_extra, &di));
    CHK(upload(ctx, S_IN4, extra_scalars, n_extra * 32, &des));
  }
  MsmPart p0{bases->x + offset * 8, bases->y + offset * 8, bases->inf + offset, (const u32*)ds, n};
  MsmPart p1{(const u32*)dx, (const u32*)dy, (const uint8_t*)di, (const u32*)des, n_extra};
  return msm_full(ctx, bases->curve, p0, p1, out_xyz);
}

namespace {
// window sums of one part copied to d_windows; window_bits 0 = the width the library picks for n
int windows_to(bp_ctx* ctx, int curve, const MsmPart& p0, int window_bits, uint64_t* d_windows, int* n_windows,
               int* window_bits_out) {
  MsmPart p1{nullptr, nullptr, nullptr, nullptr, 0};
  MsmPlan plan;
  u32* d_win = nullptr;
  const int saved = ctx->c_override;
  if (window_bits) ctx->c_override = window_bits;
  int rc = msm_device(ctx, curve, p0, p1, &plan, &d_win);
  ctx->c_override = saved;
  if (rc != BP_OK) return rc;
  if (plan.n_windows > 64) return fail(ctx, BP_E_ARG, "msm windows: more than 64 windows; raise the window width");
  HIPCHK(ctx, hipMemcpyAsync(d_windows, d_win, plan.win_bytes, hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  msm_collect_times(ctx, p0.n != 0);
  *n_windows = plan.n_windows;
  if (window_bits_out) *window_bits_out = plan.c;
  return BP_OK;
}
}  // namespace

int bp_msm_windows_dev(bp_ctx* ctx, int curve, const uint64_t* d_x, const uint64_t* d_y, const uint8_t* d_inf,
                       const uint64_t* d_scalars, size_t n, uint64_t* d_windows, int* n_windows,
                       int* window_bits) {
  if (!ctx || !d_windows || !n_windows || !window_bits || (n && (!d_x || !d_y || !d_scalars)))
    return fail(ctx, BP_E_ARG, "bp_msm_windows_dev: null argument");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  Guard g(ctx);
  MsmPart p0{(const u32*)d_x, (const u32*)d_y, d_inf, (const u32*)d_scalars, n};
  return windows_to(ctx, curve, p0, 0, d_windows, n_windows, window_bits);
}

int bp_msm_shard_window_bits(size_t n_total, int n_shards) {
  if (n_shards < 1) return BP_E_ARG;
  // the largest shard of a split into ranges whose sizes differ by at most one
  const size_t largest = n_total / (size_t)n_shards + (n_total % (size_t)n_shards ? 1 : 0);
  return msm_default_window(largest);
}

int bp_msm_shard_windows_dev(bp_ctx* ctx, int curve, const uint64_t* d_x, const uint64_t* d_y, const uint8_t* d_inf,
                             const uint64_t* d_scalars, size_t n, int window_bits, uint64_t* d_windows, int* n_windows) {
  if (!ctx || !d_windows || !n_windows || (n && (!d_x || !d_y || !d_scalars)))
    return fail(ctx, BP_E_ARG, "bp_msm_shard_windows_dev: null argument");
  if (window_bits < 2 || window_bits > 20) return fail(ctx, BP_E_ARG, "bp_msm_shard_windows_dev: window_bits outside 2..20");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  Guard g(ctx);
  MsmPart p0{(const u32*)d_x, (const u32*)d_y, d_inf, (const u32*)d_scalars, n};
  return windows_to(ctx, curve, p0, window_bits, d_windows, n_windows, nullptr);
}

int bp_msm_shard_windows_resident(bp_ctx* ctx, const bp_bases* bases, size_t offset, size_t n, const uint64_t* scalars,
                                  int window_bits, uint64_t* d_windows, int* n_windows) {
  if (!ctx || !bases || !d_windows || !n_windows || (n && !scalars))
    return fail(ctx, BP_E_ARG, "bp_msm_shard_windows_resident: null argument");
  if (bases->ctx != ctx) return fail(ctx, BP_E_ARG, "bp_msm_shard_windows_resident: bases belong to another context");
  if (offset > bases->n || n > bases->n - offset) return fail(ctx, BP_E_ARG, "bp_msm_shard_windows_resident: range outside bases");
  if (window_bits < 2 || window_bits > 20) return fail(ctx, BP_E_ARG, "bp_msm_shard_windows_resident: window_bits outside 2..20");
  Guard g(ctx);
  void* ds;
  CHK(upload(ctx, S_IN3, scalars, n * 32, &ds));
  MsmPart p0{bases->x + offset * 8, bases->y + offset * 8, bases->inf + offset, (const u32*)ds, n};
  return windows_to(ctx, bases->curve, p0, window_bits, d_windows, n_windows, nullptr);
}

int bp_msm_combine_windows_dev(bp_ctx* ctx, int curve, const uint64_t* d_windows, int n_parts, int n_windows,
                               int window_bits, uint64_t out_xyz[12]) {
  if (!ctx || !d_windows || !out_xyz || n_parts < 1 || n_windows < 1 || n_windows > 4096 || window_bits < 1)
    return fail(ctx, BP_E_ARG, "bp_msm_combine_windows_dev: bad argument");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  Guard g(ctx);
  void* d_out;
  size_t bytes = (size_t)n_windows * 128;
  CHK(reserve(ctx, S_WINDOWS, bytes, &d_out));
  hipError_t e = launch_window_combine(curve, (const u32*)d_windows, n_parts, n_windows, (u32*)d_out, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "window_combine", e);
  CHK(reserve_pinned(ctx, bytes));
  HIPCHK(ctx, hipMemcpyAsync(ctx->pinned, d_out, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  finish_windows(curve, (const uint64_t*)ctx->pinned, n_windows, window_bits, out_xyz);
  return BP_OK;
}

// ---- batched deferred MSMs over a registered base set ---------------------------------------

int bp_batch_begin(bp_ctx* ctx, const bp_bases* bases, size_t n, bp_batch** out) {
  if (!ctx || !bases || !out) return fail(ctx, BP_E_ARG, "bp_batch_begin: null argument");
  if (bases->ctx != ctx) return fail(ctx, BP_E_ARG, "bp_batch_begin: bases belong to another context");
  if (n > bases->n) return fail(ctx, BP_E_ARG, "bp_batch_begin: more scalars than registered points");
  Guard g(ctx);
  bp_batch* b = new (std::nothrow) bp_batch{ctx, bases, n, nullptr};
  if (!b) return fail(ctx, BP_E_OOM, "host allocation");
  hipError_t e = hipMalloc((void**)&b->acc, n ? n * 32 : 32);
  if (e == hipSuccess) e = hipMemsetAsync(b->acc, 0, n ? n * 32 : 32, ctx->stream);
  if (e != hipSuccess) {
    if (b->acc) (void)hipFree(b->acc);
    delete b;
    return fail(ctx, e == hipErrorOutOfMemory ? BP_E_OOM : BP_E_HIP, "bp_batch_begin", e);
  }
  *out = b;
  return BP_OK;
}

void bp_batch_free(bp_batch* b) {
  if (!b) return;
  Guard g(b->ctx);
  (void)hipStreamSynchronize(b->ctx->stream);
  (void)hipFree(b->acc);
  delete b;
}

namespace {
int batch_axpy(bp_batch* b, size_t offset, const u32* d_scalars, size_t count, const uint64_t weight[4]) {
  bp_ctx* ctx = b->ctx;
  hipError_t e = launch_vec_op(b->bases->curve, BP_VEC_AXPY, b->acc + offset * 8, d_scalars, to_arg(weight), count,
                               b->acc + offset * 8, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "bp_batch_axpy", e);
  return BP_OK;
}
}  // namespace

int bp_batch_axpy(bp_batch* b, size_t offset, const uint64_t* scalars, size_t count, const uint64_t weight[4]) {
  if (!b || !weight || (count && !scalars)) return BP_E_ARG;
  bp_ctx* ctx = b->ctx;
  if (offset > b->n || count > b->n - offset) return fail(ctx, BP_E_ARG, "bp_batch_axpy: range outside the accumulator");
  Guard g(ctx);
  if (count == 0) return BP_OK;
  void* ds;
  CHK(upload(ctx, S_IN3, scalars, count * 32, &ds));
  CHK(batch_axpy(b, offset, (const u32*)ds, count, weight));
  // the staging buffer is reused by the next call: the host may not run ahead of this launch
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return BP_OK;
}

int bp_batch_axpy_dev(bp_batch* b, size_t offset, const uint64_t* d_scalars, size_t count, const uint64_t weight[4]) {
  if (!b || !weight || (count && !d_scalars)) return BP_E_ARG;
  bp_ctx* ctx = b->ctx;
  if (offset > b->n || count > b->n - offset) return fail(ctx, BP_E_ARG, "bp_batch_axpy_dev: range outside the accumulator");
  Guard g(ctx);
  if (count == 0) return BP_OK;
  return batch_axpy(b, offset, (const u32*)d_scalars, count, weight);
}

int bp_batch_msm(bp_batch* b, const uint64_t* extra_x, const uint64_t* extra_y, const uint8_t* extra_inf,
                 const uint64_t* extra_scalars, size_t n_extra, uint64_t out_xyz[12]) {
  if (!b || !out_xyz || (n_extra && (!extra_x || !extra_y || !extra_scalars))) return BP_E_ARG;
  bp_ctx* ctx = b->ctx;
  Guard g(ctx);
  void *dx = nullptr, *dy = nullptr, *di = nullptr, *des = nullptr;
  if (n_extra) {
    CHK(upload(ctx, S_IN0, extra_x, n_extra * 32, &dx));
    CHK(upload(ctx, S_IN1, extra_y, n_extra * 32, &dy));
    if (extra_inf) CHK(upload(ctx, S_IN2, extra_inf, n_extra, &di));
    CHK(upload(ctx, S_IN4, extra_scalars, n_extra * 32, &des));
  }
  MsmPart p0{b->bases->x, b->bases->y, b->bases->inf, b->acc, b->n};
  MsmPart p1{(const u32*)dx, (const u32*)dy, (const uint8_t*)di, (const u32*)des, n_extra};
  return msm_full(ctx, b->bases->curve, p0, p1, out_xyz);
}

// ---- field-vector kernels ------------------------------------------------------

int bp_inner_product_dev(bp_ctx* ctx, int curve, const uint64_t* d_a, const uint64_t* d_b, size_t n,
                         uint64_t out[4]) {
  if (!ctx || !out || (n && (!d_a || !d_b))) return fail(ctx, BP_E_ARG, "bp_inner_product_dev: null argument");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  Guard g(ctx);
  if (n == 0) {
    memset(out, 0, 32);
    return BP_OK;
  }
  return dot_dev(ctx, curve, (const u32*)d_a, (const u32*)d_b, n, out);
}

int bp_inner_product_dev_noalloc(bp_ctx* ctx, int curve, const uint64_t* d_a, const uint64_t* d_b, size_t n,
                                 uint64_t* d_out) {
  if (!ctx || !d_out || (n && (!d_a || !d_b))) return fail(ctx, BP_E_ARG, "bp_inner_product_dev_noalloc: null argument");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  if (n > BP_NOALLOC_MAX_N) return fail(ctx, BP_E_ARG, "bp_inner_product_dev_noalloc: n exceeds BP_NOALLOC_MAX_N (2^28)");
  Guard g(ctx);
  if (!ctx->static_scratch)
    return fail(ctx, BP_E_STATE, "bp_inner_product_dev_noalloc: this context has no slice of the static array (more than 32 contexts on the GPU?)");
  if (n == 0) {
    HIPCHK(ctx, hipMemsetAsync(d_out, 0, 32, ctx->stream));
    return BP_OK;
  }
  // the partial sums of blocks [0, blocks) are all written before they are read, so the array
  // needs no clearing between calls; the same two kernels as bp_inner_product_dev
  hipError_t e = launch_inner_product(curve, (const u32*)d_a, (const u32*)d_b, n, ctx->static_scratch, (u32*)d_out, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "bp_inner_product_dev_noalloc", e);
  return BP_OK;
}

int bp_inner_product(bp_ctx* ctx, int curve, const uint64_t* a, const uint64_t* b, size_t n, uint64_t out[4]) {
  if (!ctx || !out || (n && (!a || !b))) return fail(ctx, BP_E_ARG, "bp_inner_product: null argument");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  Guard g(ctx);
  if (n == 0) {
    memset(out, 0, 32);
    return BP_OK;
  }
  void *da, *db;
  CHK(upload(ctx, S_IN0, a, n * 32, &da));
  CHK(upload(ctx, S_IN1, b, n * 32, &db));
  return dot_dev(ctx, curve, (const u32*)da, (const u32*)db, n, out);
}

int bp_sum(bp_ctx* ctx, int curve, const uint64_t* a, size_t n, uint64_t out[4]) {
  if (!ctx || !out || (n && !a)) return fail(ctx, BP_E_ARG, "bp_sum: null argument");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  Guard g(ctx);
  if (n == 0) {
    memset(out, 0, 32);
    return BP_OK;
  }
  void* da;
  CHK(upload(ctx, S_IN0, a, n * 32, &da));
  return dot_dev(ctx, curve, (const u32*)da, nullptr, n, out);
}

int bp_fold_scalars_dev(bp_ctx* ctx, int curve, const uint64_t* d_l, const uint64_t* d_r, size_t h,
                        const uint64_t x[4], const uint64_t y[4], uint64_t* d_out) {
  if (!ctx || !x || !y || (h && (!d_l || !d_r || !d_out))) return fail(ctx, BP_E_ARG, "bp_fold_scalars_dev: null argument");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  Guard g(ctx);
  HIPCHK(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
  hipError_t e = launch_fold_scalars(curve, (const u32*)d_l, (const u32*)d_r, h, to_arg(x), to_arg(y), (u32*)d_out,
                                     ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "fold_scalars", e);
  HIPCHK(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  float ms;
  if (hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5]) == hipSuccess) ctx->last_ms[2] = ms;
  return BP_OK;
}

int bp_fold_scalars(bp_ctx* ctx, int curve, const uint64_t* l, const uint64_t* r, size_t h, const uint64_t x[4],
                    const uint64_t y[4], uint64_t* out) {
  if (!ctx || !x || !y || (h && (!l || !r || !out))) return fail(ctx, BP_E_ARG, "bp_fold_scalars: null argument");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  Guard g(ctx);
  if (h == 0) return BP_OK;
  void *dl, *dr, *dout;
  CHK(upload(ctx, S_IN0, l, h * 32, &dl));
  CHK(upload(ctx, S_IN1, r, h * 32, &dr));
  CHK(reserve(ctx, S_OUT0, h * 32, &dout));
  hipError_t e = launch_fold_scalars(curve, (const u32*)dl, (const u32*)dr, h, to_arg(x), to_arg(y), (u32*)dout,
                                     ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "fold_scalars", e);
  HIPCHK(ctx, hipMemcpyAsync(out, dout, h * 32, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return BP_OK;
}

int bp_vec_op(bp_ctx* ctx, int curve, int op, const uint64_t* a, const uint64_t* b, const uint64_t s[4], size_t n,
              uint64_t* out) {
  if (!ctx || op < BP_VEC_ADD || op > BP_VEC_AXPY) return fail(ctx, BP_E_ARG, "bp_vec_op: bad op");
  bool need_b = op != BP_VEC_SCALE, need_s = (op == BP_VEC_SCALE || op == BP_VEC_AXPY);
  if ((n && (!a || !out || (need_b && !b))) || (need_s && !s)) return fail(ctx, BP_E_ARG, "bp_vec_op: null argument");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  Guard g(ctx);
  if (n == 0) return BP_OK;
  void *da, *db = nullptr, *dout;
  CHK(upload(ctx, S_IN0, a, n * 32, &da));
  if (need_b) CHK(upload(ctx, S_IN1, b, n * 32, &db));
  CHK(reserve(ctx, S_OUT0, n * 32, &dout));
  FeArg sa{};
  if (need_s) sa = to_arg(s);
  hipError_t e = launch_vec_op(curve, op, (const u32*)da, (const u32*)db, sa, n, (u32*)dout, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "vec_op", e);
  HIPCHK(ctx, hipMemcpyAsync(out, dout, n * 32, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return BP_OK;
}

static int s_vector_common(bp_ctx* ctx, int curve, const uint64_t* ch, const uint64_t* ch_inv, size_t k, u32* d_out) {
  void *dch, *dchi, *dtab;
  CHK(upload(ctx, S_IN0, ch, k * 32, &dch));
  CHK(upload(ctx, S_IN1, ch_inv, k * 32, &dchi));
  CHK(reserve(ctx, S_TABLE, s_vector_table_elems((int)k) * 32, &dtab));
  HIPCHK(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
  hipError_t e = launch_s_vector(curve, (const u32*)dch, (const u32*)dchi, (int)k, (u32*)dtab, d_out, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "s_vector", e);
  HIPCHK(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
  return BP_OK;
}

int bp_fold_s_vectors(bp_ctx* ctx, int curve, const uint64_t* ch, const uint64_t* ch_inv, const uint64_t* weights,
                      size_t k, size_t m, uint64_t* out) {
  if (!ctx || !out || !m || !weights || (k && (!ch || !ch_inv))) return fail(ctx, BP_E_ARG, "bp_fold_s_vectors: null argument");
  if (k > 28) return fail(ctx, BP_E_ARG, "bp_fold_s_vectors: k > 28");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  Guard g(ctx);
  const size_t n = (size_t)1 << k;
  void *dch = nullptr, *dchi = nullptr, *dtab, *dacc;
  if (k) {
    CHK(upload(ctx, S_IN0, ch, m * k * 32, &dch));
    CHK(upload(ctx, S_IN1, ch_inv, m * k * 32, &dchi));
  }
  CHK(reserve(ctx, S_TABLE, s_vector_table_elems((int)k) * 32, &dtab));
  CHK(reserve(ctx, S_OUT0, n * 32, &dacc));
  for (size_t i = 0; i < m; i++) {
    hipError_t e = launch_s_vector_axpy(curve, (const u32*)dch + i * k * 8, (const u32*)dchi + i * k * 8, (int)k,
                                        to_arg(weights + 4 * i), i == 0, (u32*)dtab, (u32*)dacc, ctx->stream);
    if (e != hipSuccess) return fail(ctx, BP_E_HIP, "s_vector_axpy", e);
  }
  HIPCHK(ctx, hipMemcpyAsync(out, dacc, n * 32, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return BP_OK;
}

int bp_s_vector_dev(bp_ctx* ctx, int curve, const uint64_t* ch, const uint64_t* ch_inv, size_t k, uint64_t* d_out) {
  if (!ctx || !d_out || (k && (!ch || !ch_inv))) return fail(ctx, BP_E_ARG, "bp_s_vector_dev: null argument");
  if (k > 28) return fail(ctx, BP_E_ARG, "bp_s_vector_dev: k > 28");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  Guard g(ctx);
  CHK(s_vector_common(ctx, curve, ch, ch_inv, k, (u32*)d_out));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  float ms;
  if (hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5]) == hipSuccess) ctx->last_ms[2] = ms;
  return BP_OK;
}

int bp_s_vector(bp_ctx* ctx, int curve, const uint64_t* ch, const uint64_t* ch_inv, size_t k, uint64_t* out) {
  if (!ctx || !out || (k && (!ch || !ch_inv))) return fail(ctx, BP_E_ARG, "bp_s_vector: null argument");
  if (k > 28) return fail(ctx, BP_E_ARG, "bp_s_vector: k > 28");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  Guard g(ctx);
  size_t n = (size_t)1 << k;
  void* dout;
  CHK(reserve(ctx, S_OUT0, n * 32, &dout));
  CHK(s_vector_common(ctx, curve, ch, ch_inv, k, (u32*)dout));
  HIPCHK(ctx, hipMemcpyAsync(out, dout, n * 32, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return BP_OK;
}

int bp_msm_resident_s_vector(bp_ctx* ctx, const bp_bases* bases, size_t offset, const uint64_t* ch,
                             const uint64_t* ch_inv, size_t k, uint64_t out_xyz[12]) {
  if (!ctx || !bases || !out_xyz || (k && (!ch || !ch_inv))) return fail(ctx, BP_E_ARG, "bp_msm_resident_s_vector: null argument");
  if (bases->ctx != ctx) return fail(ctx, BP_E_ARG, "bp_msm_resident_s_vector: bases belong to another context");
  if (k > 28) return fail(ctx, BP_E_ARG, "bp_msm_resident_s_vector: k > 28");
  const size_t n = (size_t)1 << k;
  if (offset > bases->n || n > bases->n - offset) return fail(ctx, BP_E_ARG, "bp_msm_resident_s_vector: range outside bases");
  Guard g(ctx);
  // the 2^k scalars never leave the device
  void* ds;
  CHK(reserve(ctx, S_IN3, n * 32, &ds));
  CHK(s_vector_common(ctx, bases->curve, ch, ch_inv, k, (u32*)ds));
  MsmPart p0{bases->x + offset * 8, bases->y + offset * 8, bases->inf + offset, (const u32*)ds, n};
  MsmPart p1{nullptr, nullptr, nullptr, nullptr, 0};
  return msm_full(ctx, bases->curve, p0, p1, out_xyz);
}

int bp_powers(bp_ctx* ctx, int curve, const uint64_t x[4], size_t n, uint64_t* out) {
  if (!ctx || !x || (n && !out)) return fail(ctx, BP_E_ARG, "bp_powers: null argument");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  Guard g(ctx);
  if (n == 0) return BP_OK;
  void *dtab, *dout;
  CHK(reserve(ctx, S_TABLE, powers_table_elems(n) * 32, &dtab));
  CHK(reserve(ctx, S_OUT0, n * 32, &dout));
  hipError_t e = launch_powers(curve, to_arg(x), n, (u32*)dtab, (u32*)dout, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "powers", e);
  HIPCHK(ctx, hipMemcpyAsync(out, dout, n * 32, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return BP_OK;
}

int bp_range_agg_poly(bp_ctx* ctx, int curve, const uint64_t* values, size_t m, unsigned n_bits, const uint64_t* s_l,
                      const uint64_t* s_r, const uint64_t y[4], const uint64_t z[4], uint64_t t1[4], uint64_t t2[4]) {
  if (!ctx || !values || !s_l || !s_r || !y || !z || !t1 || !t2) return fail(ctx, BP_E_ARG, "bp_range_agg_poly: null argument");
  if (m == 0 || n_bits == 0 || n_bits > 64 || m > ((size_t)1 << 24)) return fail(ctx, BP_E_ARG, "bp_range_agg_poly: bad size");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  Guard g(ctx);
  const size_t n = m * n_bits, blocks = (n + 255) / 256;
  ctx->range_n = 0;
  std::vector<uint64_t> tab(4 * (m + n_bits));
  if (curve == BP_SECP256K1)
    range_tables<Secp256k1Fr>(z, m, n_bits, tab.data(), tab.data() + 4 * m);
  else
    range_tables<VestaFr>(z, m, n_bits, tab.data(), tab.data() + 4 * m);
  void *dval, *dtab, *dsr, *dyv, *dpt, *dpart, *drange;
  CHK(upload(ctx, S_IN0, values, m * 8, &dval));
  CHK(upload(ctx, S_IN1, tab.data(), tab.size() * 8, &dtab));
  CHK(upload(ctx, S_IN2, s_r, n * 32, &dsr));
  CHK(reserve(ctx, S_RANGE, 6 * n * 32, &drange));
  u32 *l0 = (u32*)drange, *r0 = l0 + n * 8, *r1 = r0 + n * 8, *l1 = r1 + n * 8;
  HIPCHK(ctx, hipMemcpyAsync(l1, s_l, n * 32, hipMemcpyHostToDevice, ctx->stream));
  CHK(reserve(ctx, S_TABLE, powers_table_elems(n) * 32, &dpt));
  CHK(reserve(ctx, S_OUT0, n * 32, &dyv));
  CHK(reserve(ctx, S_SMALL, (3 * blocks + 2) * 32, &dpart));
  hipError_t e = launch_powers(curve, to_arg(y), n, (u32*)dpt, (u32*)dyv, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "powers", e);
  u32* dout = (u32*)dpart + 3 * blocks * 8;
  e = launch_range_poly(curve, (const uint64_t*)dval, n_bits, n, (const u32*)dtab, (const u32*)dtab + 8 * m,
                        (const u32*)dyv, l1, (const u32*)dsr, to_arg(z), l0, r0, r1, (u32*)dpart, dout, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "range_poly", e);
  CHK(reserve_pinned(ctx, 64));
  HIPCHK(ctx, hipMemcpyAsync(ctx->pinned, dout, 64, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(t1, ctx->pinned, 32);
  memcpy(t2, (char*)ctx->pinned + 32, 32);
  ctx->range_n = n;
  ctx->range_curve = curve;
  return BP_OK;
}

int bp_range_agg_eval(bp_ctx* ctx, int curve, const uint64_t x[4], size_t n, uint64_t* l, uint64_t* r,
                      uint64_t t_hat[4]) {
  if (!ctx || !x || !l || !r || !t_hat) return fail(ctx, BP_E_ARG, "bp_range_agg_eval: null argument");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  Guard g(ctx);
  if (n == 0 || ctx->range_n != n || ctx->range_curve != curve)
    return fail(ctx, BP_E_STATE, "bp_range_agg_eval: no vector polynomials of this size (call bp_range_agg_poly first)");
  const size_t blocks = (n + 255) / 256;
  void *drange, *dpart;
  CHK(reserve(ctx, S_RANGE, 6 * n * 32, &drange));
  CHK(reserve(ctx, S_SMALL, (blocks + 1) * 32, &dpart));
  u32 *l0 = (u32*)drange, *r0 = l0 + n * 8, *r1 = r0 + n * 8, *l1 = r1 + n * 8, *dl = l1 + n * 8, *dr = dl + n * 8;
  u32* dout = (u32*)dpart + blocks * 8;
  hipError_t e = launch_range_eval(curve, l0, l1, r0, r1, n, to_arg(x), dl, dr, (u32*)dpart, dout, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "range_eval", e);
  CHK(reserve_pinned(ctx, 32));
  HIPCHK(ctx, hipMemcpyAsync(l, dl, n * 32, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(r, dr, n * 32, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ctx->pinned, dout, 32, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(t_hat, ctx->pinned, 32);
  return BP_OK;
}

int bp_range_agg_verify_scalars(bp_ctx* ctx, int curve, const uint64_t* challenges, const uint64_t* challenge_invs,
                                size_t k, const uint64_t a[4], const uint64_t b[4], const uint64_t y_inv[4],
                                const uint64_t z[4], size_t m, unsigned n_bits, uint64_t* out_g, uint64_t* out_h) {
  if (!ctx || !a || !b || !y_inv || !z || !out_g || !out_h || (k && (!challenges || !challenge_invs)))
    return fail(ctx, BP_E_ARG, "bp_range_agg_verify_scalars: null argument");
  if (k > 28 || m == 0 || n_bits == 0 || n_bits > 64 || m * (size_t)n_bits != ((size_t)1 << k))
    return fail(ctx, BP_E_ARG, "bp_range_agg_verify_scalars: m * n_bits must be 2^k");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  Guard g(ctx);
  const size_t n = (size_t)1 << k;
  std::vector<uint64_t> tab(4 * (m + n_bits));
  if (curve == BP_SECP256K1)
    range_tables<Secp256k1Fr>(z, m, n_bits, tab.data(), tab.data() + 4 * m);
  else
    range_tables<VestaFr>(z, m, n_bits, tab.data(), tab.data() + 4 * m);
  void *dch = nullptr, *dchi = nullptr, *dtab, *dt, *dss, *dssi, *dyi, *dg, *dh;
  if (k) {
    CHK(upload(ctx, S_IN0, challenges, k * 32, &dch));
    CHK(upload(ctx, S_IN1, challenge_invs, k * 32, &dchi));
  }
  CHK(upload(ctx, S_IN2, tab.data(), tab.size() * 8, &dtab));
  size_t te = s_vector_table_elems((int)k), pe = powers_table_elems(n);
  CHK(reserve(ctx, S_TABLE, (te > pe ? te : pe) * 32, &dt));
  CHK(reserve(ctx, S_OUT0, n * 32, &dss));
  CHK(reserve(ctx, S_OUT1, n * 32, &dssi));
  CHK(reserve(ctx, S_OUT2, n * 32, &dyi));
  CHK(reserve(ctx, S_IN5, n * 32, &dg));
  CHK(reserve(ctx, S_IN6, n * 32, &dh));
  // the launches share the table buffer; they run one after the other on the context's stream
  hipError_t e = launch_s_vector(curve, (const u32*)dch, (const u32*)dchi, (int)k, (u32*)dt, (u32*)dss, ctx->stream);
  if (e == hipSuccess) e = launch_s_vector(curve, (const u32*)dchi, (const u32*)dch, (int)k, (u32*)dt, (u32*)dssi, ctx->stream);
  if (e == hipSuccess) e = launch_powers(curve, to_arg(y_inv), n, (u32*)dt, (u32*)dyi, ctx->stream);
  if (e == hipSuccess)
    e = launch_range_verify_scalars(curve, (const u32*)dss, (const u32*)dssi, (const u32*)dyi, (const u32*)dtab,
                                    (const u32*)dtab + 8 * m, n_bits, n, to_arg(a), to_arg(b), to_arg(z), (u32*)dg,
                                    (u32*)dh, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "range_verify_scalars", e);
  HIPCHK(ctx, hipMemcpyAsync(out_g, dg, n * 32, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(out_h, dh, n * 32, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return BP_OK;
}

namespace {
// The scalars of gs | hs | u of the aggregated range verifier in one device vector of 2n + 1 elements
// (S_IN5), n = m * n_bits = 2^k: bp_range_agg_verify_scalars' values and u_scalar behind them.
int range_verify_dense(bp_ctx* ctx, int curve, const uint64_t* challenges, const uint64_t* challenge_invs, size_t k,
                       const uint64_t a[4], const uint64_t b[4], const uint64_t y_inv[4], const uint64_t z[4], size_t m,
                       unsigned n_bits, const uint64_t u_scalar[4], u32** out) {
  const size_t n = (size_t)1 << k;
  std::vector<uint64_t> tab(4 * (m + n_bits));
  if (curve == BP_SECP256K1)
    range_tables<Secp256k1Fr>(z, m, n_bits, tab.data(), tab.data() + 4 * m);
  else
    range_tables<VestaFr>(z, m, n_bits, tab.data(), tab.data() + 4 * m);
  // host arguments go through the staging ring: bp_batch_axpy_range_agg returns before the launches run
  void *dch = nullptr, *dchi = nullptr, *dtab, *dt, *dss, *dssi, *dyi, *dsc;
  if (k) {
    CHK(reserve(ctx, S_IN0, k * 32, &dch));
    CHK(reserve(ctx, S_IN1, k * 32, &dchi));
    CHK(stage_h2d(ctx, dch, challenges, k * 32));
    CHK(stage_h2d(ctx, dchi, challenge_invs, k * 32));
  }
  CHK(reserve(ctx, S_IN2, tab.size() * 8, &dtab));
  CHK(stage_h2d(ctx, dtab, tab.data(), tab.size() * 8));
  size_t te = s_vector_table_elems((int)k), pe = powers_table_elems(n);
  CHK(reserve(ctx, S_TABLE, (te > pe ? te : pe) * 32, &dt));
  CHK(reserve(ctx, S_OUT0, n * 32, &dss));
  CHK(reserve(ctx, S_OUT1, n * 32, &dssi));
  CHK(reserve(ctx, S_OUT2, n * 32, &dyi));
  CHK(reserve(ctx, S_IN5, (2 * n + 1) * 32, &dsc));
  u32* dg = (u32*)dsc;
  u32* dh = dg + n * 8;
  hipError_t e = launch_s_vector(curve, (const u32*)dch, (const u32*)dchi, (int)k, (u32*)dt, (u32*)dss, ctx->stream);
  if (e == hipSuccess) e = launch_s_vector(curve, (const u32*)dchi, (const u32*)dch, (int)k, (u32*)dt, (u32*)dssi, ctx->stream);
  if (e == hipSuccess) e = launch_powers(curve, to_arg(y_inv), n, (u32*)dt, (u32*)dyi, ctx->stream);
  if (e == hipSuccess)
    e = launch_range_verify_scalars(curve, (const u32*)dss, (const u32*)dssi, (const u32*)dyi, (const u32*)dtab,
                                    (const u32*)dtab + 8 * m, n_bits, n, to_arg(a), to_arg(b), to_arg(z), dg, dh, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "range_verify_scalars", e);
  CHK(stage_h2d(ctx, dg + 2 * n * 8, u_scalar, 32));
  *out = dg;
  return BP_OK;
}
}  // namespace

int bp_range_agg_verify_msm(bp_ctx* ctx, const bp_bases* bases, const uint64_t* challenges, const uint64_t* challenge_invs,
                            size_t k, const uint64_t a[4], const uint64_t b[4], const uint64_t y_inv[4], const uint64_t z[4],
                            size_t m, unsigned n_bits, const uint64_t u_scalar[4], const uint64_t* extra_x,
                            const uint64_t* extra_y, const uint8_t* extra_inf, const uint64_t* extra_scalars, size_t n_extra,
                            uint64_t out_xyz[12]) {
  if (!ctx || !bases || !a || !b || !y_inv || !z || !u_scalar || !out_xyz || (k && (!challenges || !challenge_invs)) ||
      (n_extra && (!extra_x || !extra_y || !extra_scalars)))
    return fail(ctx, BP_E_ARG, "bp_range_agg_verify_msm: null argument");
  if (bases->ctx != ctx) return fail(ctx, BP_E_ARG, "bp_range_agg_verify_msm: bases belong to another context");
  if (k > 28 || m == 0 || n_bits == 0 || n_bits > 64 || m * (size_t)n_bits != ((size_t)1 << k))
    return fail(ctx, BP_E_ARG, "bp_range_agg_verify_msm: m * n_bits must be 2^k");
  const size_t n = (size_t)1 << k;
  if (bases->n < 2 * n + 1) return fail(ctx, BP_E_ARG, "bp_range_agg_verify_msm: base set shorter than 2 * m * n_bits + 1");
  const int curve = bases->curve;
  Guard g(ctx);
  u32* dsc = nullptr;
  CHK(range_verify_dense(ctx, curve, challenges, challenge_invs, k, a, b, y_inv, z, m, n_bits, u_scalar, &dsc));
  void *dx = nullptr, *dy = nullptr, *di = nullptr, *des = nullptr;
  if (n_extra) {
    // the slots of the challenges and tables are free again in stream order
    CHK(upload(ctx, S_IN3, extra_x, n_extra * 32, &dx));
    CHK(upload(ctx, S_IN4, extra_y, n_extra * 32, &dy));
    if (extra_inf) CHK(upload(ctx, S_IN6, extra_inf, n_extra, &di));
    CHK(upload(ctx, S_AOS, extra_scalars, n_extra * 32, &des));
  }
  MsmPart p0{bases->x, bases->y, bases->inf, (const u32*)dsc, 2 * n + 1};
  MsmPart p1{(const u32*)dx, (const u32*)dy, (const uint8_t*)di, (const u32*)des, n_extra};
  return msm_full(ctx, curve, p0, p1, out_xyz);
}

int bp_batch_axpy_range_agg(bp_batch* b, const uint64_t weight[4], const uint64_t* challenges,
                            const uint64_t* challenge_invs, size_t k, const uint64_t a[4], const uint64_t bb[4],
                            const uint64_t y_inv[4], const uint64_t z[4], size_t m, unsigned n_bits,
                            const uint64_t u_scalar[4]) {
  if (!b || !weight || !a || !bb || !y_inv || !z || !u_scalar || (k && (!challenges || !challenge_invs))) return BP_E_ARG;
  bp_ctx* ctx = b->ctx;
  if (k > 28 || m == 0 || n_bits == 0 || n_bits > 64 || m * (size_t)n_bits != ((size_t)1 << k))
    return fail(ctx, BP_E_ARG, "bp_batch_axpy_range_agg: m * n_bits must be 2^k");
  const size_t n = (size_t)1 << k;
  if (b->n < 2 * n + 1) return fail(ctx, BP_E_ARG, "bp_batch_axpy_range_agg: accumulator shorter than 2 * m * n_bits + 1");
  Guard g(ctx);
  u32* dsc = nullptr;
  CHK(range_verify_dense(ctx, b->bases->curve, challenges, challenge_invs, k, a, bb, y_inv, z, m, n_bits, u_scalar, &dsc));
  return batch_axpy(b, 0, dsc, 2 * n + 1, weight);
}

int bp_batch_axpy_s_vector(bp_batch* b, const uint64_t weight[4], size_t offset, const uint64_t* challenges,
                           const uint64_t* challenge_invs, size_t k) {
  if (!b || !weight || (k && (!challenges || !challenge_invs))) return BP_E_ARG;
  bp_ctx* ctx = b->ctx;
  if (k > 28) return fail(ctx, BP_E_ARG, "bp_batch_axpy_s_vector: too many rounds");
  const size_t n = (size_t)1 << k;
  if (offset > b->n || n > b->n - offset) return fail(ctx, BP_E_ARG, "bp_batch_axpy_s_vector: range outside the accumulator");
  Guard g(ctx);
  void *dch = nullptr, *dchi = nullptr, *dt;
  if (k) {
    CHK(reserve(ctx, S_IN0, k * 32, &dch));
    CHK(reserve(ctx, S_IN1, k * 32, &dchi));
    CHK(stage_h2d(ctx, dch, challenges, k * 32));
    CHK(stage_h2d(ctx, dchi, challenge_invs, k * 32));
  }
  CHK(reserve(ctx, S_TABLE, s_vector_table_elems((int)k) * 32, &dt));
  hipError_t e = launch_s_vector_axpy(b->bases->curve, (const u32*)dch, (const u32*)dchi, (int)k, to_arg(weight), 0, (u32*)dt,
                                      b->acc + offset * 8, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "s_vector_axpy", e);
  return BP_OK;
}

int bp_batch_axpy_ipa(bp_batch* b, const uint64_t weight[4], size_t g_offset, size_t h_offset, size_t u_index,
                      const uint64_t* challenges, const uint64_t* challenge_invs, size_t k, const uint64_t a[4],
                      const uint64_t bb[4], const uint64_t u_scalar[4]) {
  if (!b || !weight || !a || !bb || !u_scalar || (k && (!challenges || !challenge_invs))) return BP_E_ARG;
  bp_ctx* ctx = b->ctx;
  if (k > 28) return fail(ctx, BP_E_ARG, "bp_batch_axpy_ipa: too many rounds");
  const size_t n = (size_t)1 << k, nb = b->n;
  if (g_offset > nb || n > nb - g_offset || h_offset > nb || n > nb - h_offset || u_index >= nb)
    return fail(ctx, BP_E_ARG, "bp_batch_axpy_ipa: range outside the accumulator");
  const int curve = b->bases->curve;
  // weight * a, weight * b, weight * u_scalar (Montgomery products on the host)
  uint64_t wa[4], wb[4], wu[4];
  {
    host::HFe hw, ha, hb, hu;
    memcpy(hw.v, weight, 32); memcpy(ha.v, a, 32); memcpy(hb.v, bb, 32); memcpy(hu.v, u_scalar, 32);
    host::HFe ra, rb, ru;
    if (curve == BP_SECP256K1) {
      ra = host::h_mul<Secp256k1Fr>(hw, ha); rb = host::h_mul<Secp256k1Fr>(hw, hb); ru = host::h_mul<Secp256k1Fr>(hw, hu);
    } else {
      ra = host::h_mul<VestaFr>(hw, ha); rb = host::h_mul<VestaFr>(hw, hb); ru = host::h_mul<VestaFr>(hw, hu);
    }
    memcpy(wa, ra.v, 32); memcpy(wb, rb.v, 32); memcpy(wu, ru.v, 32);
  }
  Guard g(ctx);
  void *dch = nullptr, *dchi = nullptr, *dt, *dtmp;
  if (k) {
    CHK(reserve(ctx, S_IN0, k * 32, &dch));
    CHK(reserve(ctx, S_IN1, k * 32, &dchi));
    CHK(stage_h2d(ctx, dch, challenges, k * 32));
    CHK(stage_h2d(ctx, dchi, challenge_invs, k * 32));
  }
  CHK(reserve(ctx, S_TABLE, s_vector_table_elems((int)k) * 32, &dt));
  // acc[gs] += (weight a) s, acc[hs] += (weight b) s^-1   (src/ipa/mod.rs:134-166 under src/msm.rs:91-115)
  hipError_t e = launch_s_vector_axpy(curve, (const u32*)dch, (const u32*)dchi, (int)k, to_arg(wa), 0, (u32*)dt,
                                      b->acc + g_offset * 8, ctx->stream);
  if (e == hipSuccess)
    e = launch_s_vector_axpy(curve, (const u32*)dchi, (const u32*)dch, (int)k, to_arg(wb), 0, (u32*)dt, b->acc + h_offset * 8,
                             ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "s_vector_axpy", e);
  // acc[u] += weight * u_scalar
  CHK(reserve(ctx, S_IN2, 32, &dtmp));
  CHK(stage_h2d(ctx, dtmp, wu, 32));
  e = launch_vec_op(curve, BP_VEC_ADD, b->acc + u_index * 8, (const u32*)dtmp, FeArg{}, 1, b->acc + u_index * 8, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "bp_batch_axpy_ipa", e);
  return BP_OK;
}

int bp_ipa_verify_msm(bp_ctx* ctx, const bp_bases* bases, size_t g_offset, size_t h_offset, size_t u_index,
                      const uint64_t* challenges, const uint64_t* challenge_invs, size_t k, const uint64_t a[4],
                      const uint64_t b[4], const uint64_t u_scalar[4], const uint64_t* extra_x, const uint64_t* extra_y,
                      const uint8_t* extra_inf, const uint64_t* extra_scalars, size_t n_extra, uint64_t out_xyz[12]) {
  if (!ctx || !bases || !a || !b || !u_scalar || !out_xyz || (k && (!challenges || !challenge_invs)) ||
      (n_extra && (!extra_x || !extra_y || !extra_scalars)))
    return fail(ctx, BP_E_ARG, "bp_ipa_verify_msm: null argument");
  if (bases->ctx != ctx) return fail(ctx, BP_E_ARG, "bp_ipa_verify_msm: bases belong to another context");
  if (k > 28) return fail(ctx, BP_E_ARG, "bp_ipa_verify_msm: too many rounds");
  const size_t n = (size_t)1 << k, nb = bases->n;
  if (g_offset > nb || n > nb - g_offset || h_offset > nb || n > nb - h_offset || u_index >= nb)
    return fail(ctx, BP_E_ARG, "bp_ipa_verify_msm: range outside bases");
  const bool overlap = (g_offset < h_offset + n && h_offset < g_offset + n) || (u_index >= g_offset && u_index < g_offset + n) ||
                       (u_index >= h_offset && u_index < h_offset + n);
  if (overlap) return fail(ctx, BP_E_ARG, "bp_ipa_verify_msm: gs, hs and u must not overlap");
  const int curve = bases->curve;
  Guard g(ctx);
  // one scalar per point of bases[lo, hi): a s_i on gs, b s_i^-1 on hs, u_scalar on u, zero elsewhere
  const size_t lo = std::min(std::min(g_offset, h_offset), u_index);
  const size_t hi = std::max(std::max(g_offset + n, h_offset + n), u_index + 1);
  void *dch = nullptr, *dchi = nullptr, *dt, *dss, *dsc;
  if (k) {
    CHK(upload(ctx, S_IN0, challenges, k * 32, &dch));
    CHK(upload(ctx, S_IN1, challenge_invs, k * 32, &dchi));
  }
  CHK(reserve(ctx, S_TABLE, s_vector_table_elems((int)k) * 32, &dt));
  CHK(reserve(ctx, S_OUT0, n * 32, &dss));
  CHK(reserve(ctx, S_IN5, (hi - lo) * 32, &dsc));
  u32* dense = (u32*)dsc;
  if (hi - lo != 2 * n + 1) HIPCHK(ctx, hipMemsetAsync(dense, 0, (hi - lo) * 32, ctx->stream));
  // no rounds: the s-vector of no challenges is (1)
  hipError_t e = k ? launch_s_vector(curve, (const u32*)dch, (const u32*)dchi, (int)k, (u32*)dt, (u32*)dss, ctx->stream)
                   : launch_fill_one(curve, (u32*)dss, 1, ctx->stream);
  if (e == hipSuccess)  // src/ipa/mod.rs:134-148,160
    e = launch_vec_op(curve, BP_VEC_SCALE, (const u32*)dss, nullptr, to_arg(a), n, dense + (g_offset - lo) * 8, ctx->stream);
  if (e == hipSuccess && k)
    e = launch_s_vector(curve, (const u32*)dchi, (const u32*)dch, (int)k, (u32*)dt, (u32*)dss, ctx->stream);
  if (e == hipSuccess)  // :150-154,166
    e = launch_vec_op(curve, BP_VEC_SCALE, (const u32*)dss, nullptr, to_arg(b), n, dense + (h_offset - lo) * 8, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "ipa_verify scalars", e);
  HIPCHK(ctx, hipMemcpyAsync(dense + (u_index - lo) * 8, u_scalar, 32, hipMemcpyHostToDevice, ctx->stream));
  void *dx = nullptr, *dy = nullptr, *di = nullptr, *des = nullptr;
  if (n_extra) {
    CHK(upload(ctx, S_IN3, extra_x, n_extra * 32, &dx));
    CHK(upload(ctx, S_IN4, extra_y, n_extra * 32, &dy));
    if (extra_inf) CHK(upload(ctx, S_IN6, extra_inf, n_extra, &di));
    CHK(upload(ctx, S_AOS, extra_scalars, n_extra * 32, &des));
  }
  MsmPart p0{bases->x + lo * 8, bases->y + lo * 8, bases->inf + lo, (const u32*)dsc, hi - lo};
  MsmPart p1{(const u32*)dx, (const u32*)dy, (const uint8_t*)di, (const u32*)des, n_extra};
  return msm_full(ctx, curve, p0, p1, out_xyz);
}

int bp_batch_inverse(bp_ctx* ctx, int curve, int field, uint64_t* v, size_t n) {
  if (!ctx || (n && !v) || (field != 0 && field != 1)) return fail(ctx, BP_E_ARG, "bp_batch_inverse: bad argument");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  Guard g(ctx);
  if (n == 0) return BP_OK;
  void* dv;
  CHK(upload(ctx, S_IN0, v, n * 32, &dv));
  hipError_t e = launch_batch_inverse(curve, field, (u32*)dv, n, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "batch_inverse", e);
  HIPCHK(ctx, hipMemcpyAsync(v, dv, n * 32, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return BP_OK;
}

// ---- curve-vector kernels -------------------------------------------------------

int bp_fold_generators_dev(bp_ctx* ctx, int curve, const uint64_t* d_lx, const uint64_t* d_ly, const uint8_t* d_linf,
                           const uint64_t* d_rx, const uint64_t* d_ry, const uint8_t* d_rinf, size_t h,
                           const uint64_t x[4], const uint64_t y[4], uint64_t* d_out_x, uint64_t* d_out_y,
                           uint8_t* d_out_inf) {
  if (!ctx || !x || !y || (h && (!d_lx || !d_ly || !d_rx || !d_ry || !d_out_x || !d_out_y)))
    return fail(ctx, BP_E_ARG, "bp_fold_generators_dev: null argument");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  Guard g(ctx);
  HIPCHK(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
  hipError_t e = launch_fold_generators(curve, (const u32*)d_lx, (const u32*)d_ly, d_linf, (const u32*)d_rx,
                                        (const u32*)d_ry, d_rinf, h, to_canonical_fr(curve, x),
                                        to_canonical_fr(curve, y), (u32*)d_out_x, (u32*)d_out_y, d_out_inf,
                                        ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "fold_generators", e);
  HIPCHK(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  float ms;
  if (hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5]) == hipSuccess) ctx->last_ms[2] = ms;
  return BP_OK;
}

int bp_fold_generators(bp_ctx* ctx, int curve, const uint64_t* lx, const uint64_t* ly, const uint8_t* linf,
                       const uint64_t* rx, const uint64_t* ry, const uint8_t* rinf, size_t h, const uint64_t x[4],
                       const uint64_t y[4], uint64_t* out_x, uint64_t* out_y, uint8_t* out_inf) {
  if (!ctx || !x || !y || (h && (!lx || !ly || !rx || !ry || !out_x || !out_y)))
    return fail(ctx, BP_E_ARG, "bp_fold_generators: null argument");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  Guard g(ctx);
  if (h == 0) return BP_OK;
  void *dlx, *dly, *dli = nullptr, *drx, *dry, *dri = nullptr, *dox, *doy, *doi;
  CHK(upload(ctx, S_IN0, lx, h * 32, &dlx));
  CHK(upload(ctx, S_IN1, ly, h * 32, &dly));
  if (linf) CHK(upload(ctx, S_IN2, linf, h, &dli));
  CHK(upload(ctx, S_IN3, rx, h * 32, &drx));
  CHK(upload(ctx, S_IN4, ry, h * 32, &dry));
  if (rinf) CHK(upload(ctx, S_IN5, rinf, h, &dri));
  CHK(reserve(ctx, S_OUT0, h * 32, &dox));
  CHK(reserve(ctx, S_OUT1, h * 32, &doy));
  CHK(reserve(ctx, S_OUT2, h, &doi));
  hipError_t e = launch_fold_generators(curve, (const u32*)dlx, (const u32*)dly, (const uint8_t*)dli, (const u32*)drx,
                                        (const u32*)dry, (const uint8_t*)dri, h, to_canonical_fr(curve, x),
                                        to_canonical_fr(curve, y), (u32*)dox, (u32*)doy, (uint8_t*)doi, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "fold_generators", e);
  HIPCHK(ctx, hipMemcpyAsync(out_x, dox, h * 32, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(out_y, doy, h * 32, hipMemcpyDeviceToHost, ctx->stream));
  if (out_inf) HIPCHK(ctx, hipMemcpyAsync(out_inf, doi, h, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return BP_OK;
}

int bp_scale_points(bp_ctx* ctx, int curve, const uint64_t* px, const uint64_t* py, const uint8_t* pinf,
                    const uint64_t* s, size_t n, uint64_t* out_x, uint64_t* out_y, uint8_t* out_inf) {
  if (!ctx || (n && (!px || !py || !s || !out_x || !out_y))) return fail(ctx, BP_E_ARG, "bp_scale_points: null argument");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  Guard g(ctx);
  if (n == 0) return BP_OK;
  void *dx, *dy, *di = nullptr, *ds, *dox, *doy, *doi;
  CHK(upload(ctx, S_IN0, px, n * 32, &dx));
  CHK(upload(ctx, S_IN1, py, n * 32, &dy));
  if (pinf) CHK(upload(ctx, S_IN2, pinf, n, &di));
  CHK(upload(ctx, S_IN3, s, n * 32, &ds));
  CHK(reserve(ctx, S_OUT0, n * 32, &dox));
  CHK(reserve(ctx, S_OUT1, n * 32, &doy));
  CHK(reserve(ctx, S_OUT2, n, &doi));
  hipError_t e = launch_scale_points(curve, (const u32*)dx, (const u32*)dy, (const uint8_t*)di, (const u32*)ds, n,
                                     (u32*)dox, (u32*)doy, (uint8_t*)doi, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "scale_points", e);
  HIPCHK(ctx, hipMemcpyAsync(out_x, dox, n * 32, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(out_y, doy, n * 32, hipMemcpyDeviceToHost, ctx->stream));
  if (out_inf) HIPCHK(ctx, hipMemcpyAsync(out_inf, doi, n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return BP_OK;
}

int bp_scalars_to_bytes(bp_ctx* ctx, int curve, const uint64_t* v, size_t n, uint8_t* out) {
  if (!ctx || (n && (!v || !out))) return fail(ctx, BP_E_ARG, "bp_scalars_to_bytes: null argument");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  Guard g(ctx);
  if (n == 0) return BP_OK;
  void *dv, *dout;
  CHK(upload(ctx, S_IN0, v, n * 32, &dv));
  CHK(reserve(ctx, S_OUT0, n * 32, &dout));
  hipError_t e = launch_scalars_to_bytes(curve, (const u32*)dv, n, (u32*)dout, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "scalars_to_bytes", e);
  HIPCHK(ctx, hipMemcpyAsync(out, dout, n * 32, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return BP_OK;
}

int bp_scalars_from_bytes(bp_ctx* ctx, int curve, const uint8_t* in, size_t n, uint64_t* out, uint8_t* out_ok) {
  if (!ctx || (n && (!in || !out || !out_ok))) return fail(ctx, BP_E_ARG, "bp_scalars_from_bytes: null argument");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  Guard g(ctx);
  if (n == 0) return BP_OK;
  void *din, *dv, *dok;
  CHK(upload(ctx, S_IN0, in, n * 32, &din));
  CHK(reserve(ctx, S_OUT0, n * 32, &dv));
  CHK(reserve(ctx, S_OUT2, n, &dok));
  hipError_t e = launch_scalars_from_bytes(curve, (const u32*)din, n, (u32*)dv, (uint8_t*)dok, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "scalars_from_bytes", e);
  HIPCHK(ctx, hipMemcpyAsync(out, dv, n * 32, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(out_ok, dok, n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return BP_OK;
}

int bp_points_compress(bp_ctx* ctx, int curve, const uint64_t* px, const uint64_t* py, const uint8_t* pinf, size_t n,
                       uint8_t* out) {
  if (!ctx || (n && (!px || !py || !out))) return fail(ctx, BP_E_ARG, "bp_points_compress: null argument");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  Guard g(ctx);
  if (n == 0) return BP_OK;
  void *dx, *dy, *di = nullptr, *dout;
  CHK(upload(ctx, S_IN0, px, n * 32, &dx));
  CHK(upload(ctx, S_IN1, py, n * 32, &dy));
  if (pinf) CHK(upload(ctx, S_IN2, pinf, n, &di));
  CHK(reserve(ctx, S_OUT0, n * BP_POINT_WIRE_BYTES, &dout));
  hipError_t e = launch_points_compress(curve, (const u32*)dx, (const u32*)dy, (const uint8_t*)di, n, (uint8_t*)dout,
                                        ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "points_compress", e);
  HIPCHK(ctx, hipMemcpyAsync(out, dout, n * BP_POINT_WIRE_BYTES, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return BP_OK;
}

int bp_points_decompress(bp_ctx* ctx, int curve, const uint8_t* in, size_t n, uint64_t* out_x, uint64_t* out_y,
                         uint8_t* out_inf, uint8_t* out_ok) {
  if (!ctx || (n && (!in || !out_x || !out_y || !out_inf || !out_ok)))
    return fail(ctx, BP_E_ARG, "bp_points_decompress: null argument");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  Guard g(ctx);
  if (n == 0) return BP_OK;
  void *din, *dox, *doy, *doi, *dok;
  CHK(upload(ctx, S_IN0, in, n * BP_POINT_WIRE_BYTES, &din));
  CHK(reserve(ctx, S_OUT0, n * 32, &dox));
  CHK(reserve(ctx, S_OUT1, n * 32, &doy));
  CHK(reserve(ctx, S_OUT2, n, &doi));
  CHK(reserve(ctx, S_IN1, n, &dok));
  hipError_t e = launch_points_decompress(curve, (const uint8_t*)din, n, (u32*)dox, (u32*)doy, (uint8_t*)doi,
                                          (uint8_t*)dok, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "points_decompress", e);
  HIPCHK(ctx, hipMemcpyAsync(out_x, dox, n * 32, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(out_y, doy, n * 32, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(out_inf, doi, n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(out_ok, dok, n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return BP_OK;
}

int bp_normalize_batch(bp_ctx* ctx, int curve, const uint64_t* xyz, size_t n, uint64_t* out_x, uint64_t* out_y,
                       uint8_t* out_inf) {
  if (!ctx || (n && (!xyz || !out_x || !out_y))) return fail(ctx, BP_E_ARG, "bp_normalize_batch: null argument");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  Guard g(ctx);
  if (n == 0) return BP_OK;
  void *din, *dox, *doy, *doi;
  CHK(upload(ctx, S_IN0, xyz, n * 96, &din));
  CHK(reserve(ctx, S_OUT0, n * 32, &dox));
  CHK(reserve(ctx, S_OUT1, n * 32, &doy));
  CHK(reserve(ctx, S_OUT2, n, &doi));
  hipError_t e = launch_normalize_jac(curve, (const u32*)din, n, (u32*)dox, (u32*)doy, (uint8_t*)doi, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "normalize_batch", e);
  HIPCHK(ctx, hipMemcpyAsync(out_x, dox, n * 32, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(out_y, doy, n * 32, hipMemcpyDeviceToHost, ctx->stream));
  if (out_inf) HIPCHK(ctx, hipMemcpyAsync(out_inf, doi, n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return BP_OK;
}

// ---- IPA session -----------------------------------------------------------------

namespace {

// Everything a session owns on the device lives in one block, so a prove costs one
// allocation; a freed block is parked on the context for the next session.
struct IpaLayout {
  size_t gx, gy, ginf, hx, hy, hinf, a, b, a2, b2, cg, ch, eg, eh, u, dots, total;
};

IpaLayout ipa_layout(size_t n, bool has_h, bool owns_gens) {
  IpaLayout l{};
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t at = off;
    off += (bytes + 255) & ~(size_t)255;
    return at;
  };
  if (owns_gens) {
    l.gx = take(n * 32); l.gy = take(n * 32); l.ginf = take(n);
    if (has_h) { l.hx = take(n * 32); l.hy = take(n * 32); l.hinf = take(n); }
  }
  l.a = take(n * 32); l.b = take(n * 32);
  l.a2 = take(n * 16 + 32); l.b2 = take(n * 16 + 32);
  l.cg = take(n * 32); l.eg = take(n * 32);
  if (has_h) { l.ch = take(n * 32); l.eh = take(n * 32); }
  l.u = take(64);
  l.dots = take(((size_t)2 * kDotBlocks + 4) * 32);
  l.total = off;
  return l;
}

int ipa_take_block(bp_ctx* ctx, size_t bytes, void** out, size_t* cap) {
  if (ctx->ipa_cache && ctx->ipa_cache_bytes >= bytes) {
    *out = ctx->ipa_cache;
    *cap = ctx->ipa_cache_bytes;
    ctx->ipa_cache = nullptr;
    ctx->ipa_cache_bytes = 0;
    return BP_OK;
  }
  HIPCHK(ctx, hipMalloc(out, bytes));
  *cap = bytes;
  return BP_OK;
}

void ipa_release_block(bp_ctx* ctx, void* block, size_t cap) {
  if (!block) return;
  if (!ctx->ipa_cache) {
    ctx->ipa_cache = block;
    ctx->ipa_cache_bytes = cap;
    return;
  }
  (void)hipStreamSynchronize(ctx->stream);
  if (cap > ctx->ipa_cache_bytes) {
    (void)hipFree(ctx->ipa_cache);
    ctx->ipa_cache = block;
    ctx->ipa_cache_bytes = cap;
  } else {
    (void)hipFree(block);
  }
}

int ipa_begin_common(bp_ctx* ctx, int curve, int variant, const bp_bases* res, size_t g_off, size_t h_off,
                     const uint64_t* gx, const uint64_t* gy, const uint8_t* ginf, const uint64_t* hx,
                     const uint64_t* hy, const uint8_t* hinf, const uint64_t* a, const uint64_t* b, size_t n,
                     bp_ipa** out, const uint64_t* b_power = nullptr) {
  // b == nullptr: b_i = b_power^i, built on the device
  void* pow_tab = nullptr;
  if (!b) CHK(reserve(ctx, S_TABLE, powers_table_elems(n) * 32, &pow_tab));
  bp_ipa* s = new (std::nothrow) bp_ipa();
  if (!s) return fail(ctx, BP_E_OOM, "host allocation");
  s->ctx = ctx;
  s->curve = curve;
  s->variant = variant;
  s->n = s->n0 = n;
  const bool has_h = variant == BP_IPA_BULLETPROOF;
  const IpaLayout l = ipa_layout(n, has_h, res == nullptr);
  void* blk = nullptr;
  int rc = ipa_take_block(ctx, l.total, &blk, &s->block_bytes);
  if (rc != BP_OK) {
    delete s;
    return rc;
  }
  s->block = blk;
  char* base = (char*)blk;
  auto at = [&](size_t off) { return (u32*)(base + off); };
  s->a = at(l.a); s->b = at(l.b); s->a2 = at(l.a2); s->b2 = at(l.b2); s->cg = at(l.cg); s->eg = at(l.eg);
  if (has_h) { s->ch = at(l.ch); s->eh = at(l.eh); }
  s->u_xy = at(l.u);
  s->dots = at(l.dots);
  if (ctx->host_slots && ctx->host_slots_free) {
    s->host_slot = __builtin_ctzll(ctx->host_slots_free);
    ctx->host_slots_free &= ~(1ull << s->host_slot);
  }
  hipError_t e = hipSuccess;
  auto put = [&](void* d, const void* h, size_t bytes) {
    if (e == hipSuccess) e = hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, ctx->stream);
  };
  if (res) {
    s->res = res; s->g_off = g_off; s->h_off = h_off;
    s->gx = res->x + g_off * 8; s->gy = res->y + g_off * 8; s->ginf = res->inf + g_off;
    if (has_h) { s->hx = res->x + h_off * 8; s->hy = res->y + h_off * 8; s->hinf = res->inf + h_off; }
  } else {
    u32 *dgx = at(l.gx), *dgy = at(l.gy);
    uint8_t* dgi = (uint8_t*)(base + l.ginf);
    s->gx = dgx; s->gy = dgy; s->ginf = dgi;
    put(dgx, gx, n * 32); put(dgy, gy, n * 32);
    if (e == hipSuccess) e = ginf ? hipMemcpyAsync(dgi, ginf, n, hipMemcpyHostToDevice, ctx->stream) : hipMemsetAsync(dgi, 0, n, ctx->stream);
    if (has_h) {
      u32 *dhx = at(l.hx), *dhy = at(l.hy);
      uint8_t* dhi = (uint8_t*)(base + l.hinf);
      s->hx = dhx; s->hy = dhy; s->hinf = dhi;
      put(dhx, hx, n * 32); put(dhy, hy, n * 32);
      if (e == hipSuccess) e = hinf ? hipMemcpyAsync(dhi, hinf, n, hipMemcpyHostToDevice, ctx->stream) : hipMemsetAsync(dhi, 0, n, ctx->stream);
    }
  }
  put(s->a, a, n * 32);
  if (b)
    put(s->b, b, n * 32);
  else if (e == hipSuccess)
    e = launch_powers(curve, to_arg(b_power), n, (u32*)pow_tab, s->b, ctx->stream);
  if (e == hipSuccess) e = launch_fill_one(curve, s->cg, n, ctx->stream);
  if (e == hipSuccess && has_h) e = launch_fill_one(curve, s->ch, n, ctx->stream);
  // the host arrays may be reused by the caller as soon as this returns
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    int code = fail(ctx, e == hipErrorOutOfMemory ? BP_E_OOM : BP_E_HIP, "bp_ipa_begin", e);
    ipa_release_block(ctx, s->block, s->block_bytes);
    if (s->host_slot >= 0) ctx->host_slots_free |= 1ull << s->host_slot;
    delete s;
    return code;
  }
  *out = s;
  return BP_OK;
}

// The step lists of launch_ipa_rebase_tab for one generator set: `coef` = the m shared coefficients
// (Montgomery form); lists for 2^log_s lanes, each `*len` steps long (padded), appended to `out`.
// The signed 8-bit digits of the coefficients are sorted by size, largest first, and dealt out in equal
// runs.  A lane walks its run as a bucket reduction walks buckets: run += record; total += run once per
// digit value passed; what is left below its smallest digit c, (c - 1) * run, is made by double-and-add
// in tmp and added to total.
// Length every list gets (padded with no-operations): a lane's share of the m * 32 digits, a step per
// digit size it can pass, and the double-and-add of what is left below its smallest digit.
size_t rebase_tab_list_len(size_t m, int log_s) {
  const size_t lanes = (size_t)1 << log_s;
  return (m * kTableWindows + lanes - 1) / lanes + 128 + 16;
}

// writes 2^log_s lists of rebase_tab_list_len(m, log_s) words to out; no allocation (called once per proof
// and generator set of a batch)
bool rebase_tab_lists(int curve, const uint64_t* coef, size_t m, int log_s, u32* out) {
  constexpr size_t kMaxEntries = 8192;
  if (m * kTableWindows > kMaxEntries) return false;
  u32 unsorted[kMaxEntries], sorted[kMaxEntries];  // mag << 24 | neg << 23 | w << 16 | t
  u32 count[130] = {0};
  size_t n_entries = 0;
  for (size_t t = 0; t < m; t++) {
    host::HFe c;
    memcpy(c.v, coef + 4 * t, 32);
    c = curve == BP_SECP256K1 ? host::h_from_mont<Secp256k1Fr>(c) : host::h_from_mont<VestaFr>(c);
    const host::HFe nc = curve == BP_SECP256K1 ? host::h_neg<Secp256k1Fr>(c) : host::h_neg<VestaFr>(c);  // r - c on the canonical value
    u32 flip = 0;
    // the smaller of c and r - c
    for (int q = 3; q >= 0; q--) {
      if (c.v[q] != nc.v[q]) {
        if (c.v[q] > nc.v[q]) { c = nc; flip = 1; }
        break;
      }
    }
    u32 carry = 0;
    for (int w = 0; w < kTableWindows; w++) {
      const u32 v = (u32)((c.v[w >> 3] >> (8 * (w & 7))) & 255u) + carry;
      carry = v > 128u ? 1u : 0u;
      const u32 mag = carry ? 256u - v : v;
      if (mag) {
        unsorted[n_entries++] = mag << 24 | (carry ^ flip) << 23 | (u32)w << 16 | (u32)t;
        count[mag]++;
      }
    }
    if (carry) return false;  // cannot happen: the value is below 2^255
  }
  // counting sort, largest digit size first, stable
  u32 start[130];
  {
    u32 at = 0;
    for (int mag = 128; mag >= 1; mag--) { start[mag] = at; at += count[mag]; }
  }
  for (size_t k = 0; k < n_entries; k++) sorted[start[unsorted[k] >> 24]++] = unsorted[k];
  const size_t lanes = (size_t)1 << log_s, per = (n_entries + lanes - 1) / lanes, len = rebase_tab_list_len(m, log_s);
  enum : u32 { RUN = 0, TOTAL = 1, TMP = 2, TABLE = 3 };
  auto step = [](u32 dst, u32 src) { return dst << 30 | src << 28; };
  for (size_t s = 0; s < lanes; s++) {
    u32* l = out + s * len;
    size_t at = 0;
    const size_t b = std::min(n_entries, s * per), e = std::min(n_entries, b + per);
    if (b < e) {
      u32 cur = sorted[b] >> 24;
      for (size_t k = b; k < e; k++) {
        const u32 en = sorted[k], mag = en >> 24;
        for (; cur > mag; cur--) l[at++] = step(TOTAL, RUN);
        l[at++] = step(RUN, TABLE) | ((en >> 23) & 1u) << 27 | ((en >> 16) & 31u) << 22 | (en & 0xFFFFu);
      }
      l[at++] = step(TOTAL, RUN);
      const u32 rest = cur - 1;  // the digit values below the run's smallest: total += rest * run
      if (rest) {
        l[at++] = step(TMP, RUN);
        int top = 31;
        while (!((rest >> top) & 1u)) top--;
        for (int bit = top - 1; bit >= 0; bit--) {
          l[at++] = step(TMP, TMP);
          if ((rest >> bit) & 1u) l[at++] = step(TMP, RUN);
        }
        l[at++] = step(TOTAL, TMP);
      }
    }
    for (; at < len; at++) l[at] = kRebaseTabNop;
  }
  return true;
}

bool rebase_tab_program(int curve, const uint64_t* coef, size_t m, int log_s, std::vector<u32>& out, int* len) {
  const size_t l = rebase_tab_list_len(m, log_s), lanes = (size_t)1 << log_s;
  if (l * lanes > kRebaseTabProgMax) return false;
  const size_t at = out.size();
  out.resize(at + l * lanes);
  *len = (int)l;
  return rebase_tab_lists(curve, coef, m, log_s, out.data() + at);
}

// Replaces the session's generators by the folded ones of the current length n (see
// launch_ipa_rebase): from here on the rounds run over n (2n with H) points instead of n0 each, the
// coefficients start again at one and the fold history is empty.  Only for sessions whose H
// coefficients are the same in every column (no bp_ipa_scale_h).
int ipa_rebase(bp_ipa* s) {
  bp_ctx* ctx = s->ctx;
  const size_t n = s->n, n0 = s->n0, m = n0 / n;
  const bool has_h = s->variant == BP_IPA_BULLETPROOF;
  const int sets = has_h ? 2 : 1;
  // the m coefficients of column 0 (the same in every column): cg[n t], ch[n t]
  std::vector<uint64_t> coef((size_t)sets * m * 4);
  {
    void* d_col;
    CHK(reserve(ctx, S_IN6, (size_t)sets * m * 32, &d_col));
    hipError_t ge = launch_gather_column(s->cg, n, m, (u32*)d_col, ctx->stream);
    if (ge == hipSuccess && has_h) ge = launch_gather_column(s->ch, n, m, (u32*)d_col + m * 8, ctx->stream);
    if (ge != hipSuccess) return fail(ctx, BP_E_HIP, "gather_column", ge);
    CHK(reserve_pinned(ctx, (size_t)sets * m * 32));
    HIPCHK(ctx, hipMemcpyAsync(ctx->pinned, d_col, (size_t)sets * m * 32, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(coef.data(), ctx->pinned, (size_t)sets * m * 32);
  }
  // new planes: x | y per set, then the infinity flags
  const size_t plane = (n * 32 + 255) & ~(size_t)255;
  void* blk = nullptr;
  HIPCHK(ctx, hipMalloc(&blk, (size_t)sets * (2 * plane + ((n + 255) & ~(size_t)255))));
  char* base = (char*)blk;
  u32* ngx = (u32*)base;
  u32* ngy = (u32*)(base + plane);
  u32* nhx = (u32*)(base + 2 * plane);
  u32* nhy = (u32*)(base + 3 * plane);
  uint8_t* ngi = (uint8_t*)(base + (size_t)sets * 2 * plane);
  uint8_t* nhi = ngi + ((n + 255) & ~(size_t)255);
  hipError_t e = hipErrorUnknown;
  bool launched = false;
  static const bool no_tab_rebase = getenv("BP_IPA_REBASE_NO_TAB") != nullptr;
  if (s->res && s->res->tab && !no_tab_rebase) {
    // the generators are those of a base set with window tables: the folded ones are sums of small multiples
    // of table records.  Lanes per output: as many as fill the machine about once.
    int log_s = 0;
    while (log_s < kRebaseTabMaxLogS && ((n * (size_t)sets) << (log_s + 1)) <= 65536) log_s++;
    if (const char* ev = getenv("BP_IPA_REBASE_LOG_S")) log_s = std::min(kRebaseTabMaxLogS, std::max(0, atoi(ev)));
    std::vector<u32> prog;
    int len[2] = {0, 0};
    bool ok = true;
    for (int k = 0; k < sets && ok; k++)
      ok = rebase_tab_program(s->curve, &coef[(size_t)k * m * 4], m, log_s, prog, &len[k]);
    if (ok && sets == 2 && len[0] != len[1]) {
      // one launch serves both sets: the shorter lists are padded to the longer
      const int longest = std::max(len[0], len[1]);
      std::vector<u32> padded;
      size_t at = 0;
      for (int k = 0; k < 2; k++)
        for (size_t lane = 0; lane < ((size_t)1 << log_s); lane++) {
          padded.insert(padded.end(), prog.begin() + at, prog.begin() + at + len[k]);
          padded.resize(padded.size() + (size_t)(longest - len[k]), kRebaseTabNop);
          at += (size_t)len[k];
        }
      prog.swap(padded);
      len[0] = len[1] = longest;
      ok = ((size_t)longest << log_s) <= kRebaseTabProgMax;
    }
    if (ok) {
      void* d_prog;
      int rc = upload(ctx, S_IN5, prog.data(), prog.size() * sizeof(u32), &d_prog);
      if (rc != BP_OK) { (void)hipFree(blk); return rc; }
      const size_t per_set = (size_t)len[0] << log_s;
      RebaseTabSet ts[2] = {{(u32)s->g_off, s->res->inf, (const u32*)d_prog, ngx, ngy, ngi},
                            {(u32)s->h_off, s->res->inf, (const u32*)d_prog + per_set, nhx, nhy, nhi}};
      e = launch_ipa_rebase_tab(s->curve, s->res->tab, s->res->n, ts, sets, n, log_s, len[0], ctx->stream);
      launched = true;
    }
  }
  if (!launched) {
  // signed 3-bit digits of the canonical values: v = sum_w d_w 8^w, d_w in [-4, 4]
  std::vector<int8_t> digits((size_t)sets * m * kRebaseWindows);
  for (size_t r = 0; r < (size_t)sets * m; r++) {
    host::HFe c;
    memcpy(c.v, &coef[r * 4], 32);
    c = s->curve == BP_SECP256K1 ? host::h_from_mont<Secp256k1Fr>(c) : host::h_from_mont<VestaFr>(c);
    unsigned carry = 0;
    for (int w = 0; w < kRebaseWindows; w++) {
      const int bit = 3 * w;
      unsigned v = carry;
      if (bit < 256) {
        unsigned raw = (unsigned)(c.v[bit >> 6] >> (bit & 63));
        if ((bit & 63) > 61 && (bit >> 6) < 3) raw |= (unsigned)(c.v[(bit >> 6) + 1] << (64 - (bit & 63)));
        v += raw & 7u;
      }
      if (v > 4) {
        digits[r * kRebaseWindows + w] = (int8_t)((int)v - 8);
        carry = 1;
      } else {
        digits[r * kRebaseWindows + w] = (int8_t)v;
        carry = 0;
      }
    }
    if (carry) { (void)hipFree(blk); return fail(ctx, BP_E_STATE, "ipa_rebase: digit overflow"); }  // cannot happen below 2^256
  }
  void *d_digits, *tmp;
  int rc = upload(ctx, S_IN5, digits.data(), digits.size(), &d_digits);
  if (rc == BP_OK) rc = reserve(ctx, S_REBASE, (size_t)sets * n * kRebaseWindows * 128, &tmp);
  if (rc != BP_OK) { (void)hipFree(blk); return rc; }
  RebaseSet rs[2] = {
      {s->gx, s->gy, s->ginf, (const int8_t*)d_digits, (u32*)tmp, ngx, ngy, ngi},
      {s->hx, s->hy, s->hinf, (const int8_t*)d_digits + m * kRebaseWindows, (u32*)tmp + n * kRebaseWindows * 32, nhx, nhy, nhi}};
  e = launch_ipa_rebase(s->curve, rs, sets, n, m, ctx->stream);
  }
  if (e == hipSuccess) e = launch_fill_one(s->curve, s->cg, n, ctx->stream);
  if (e == hipSuccess && has_h) e = launch_fill_one(s->curve, s->ch, n, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    (void)hipFree(blk);
    return fail(ctx, BP_E_HIP, "ipa_rebase", e);
  }
  if (s->rebase_block) (void)hipFree(s->rebase_block);
  s->rebase_block = blk;
  s->gx = ngx; s->gy = ngy; s->ginf = ngi;
  if (has_h) { s->hx = nhx; s->hy = nhy; s->hinf = nhi; }
  s->res = nullptr;  // the session no longer reads the registered base set
  s->g_off = s->h_off = 0;
  s->n0 = n;
  s->x_canon.clear();
  s->xinv_canon.clear();
  s->prepped = false;
  return BP_OK;
}

}  // namespace

void bp_ipa_free(bp_ipa* s) {
  if (!s) return;
  Guard g(s->ctx);
  void* ptrs[] = {s->fgx, s->fgy, s->fginf, s->fhx, s->fhy, s->fhinf};
  bool any = false;
  for (void* p : ptrs) any = any || p;
  if (any) (void)hipStreamSynchronize(s->ctx->stream);
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (s->rebase_block) (void)hipFree(s->rebase_block);
  ipa_release_block(s->ctx, s->block, s->block_bytes);
  if (s->host_slot >= 0) s->ctx->host_slots_free |= 1ull << s->host_slot;
  delete s;
}

int bp_ipa_begin(bp_ctx* ctx, int curve, int variant, const uint64_t* gx, const uint64_t* gy, const uint8_t* ginf,
                 const uint64_t* hx, const uint64_t* hy, const uint8_t* hinf, const uint64_t* a, const uint64_t* b,
                 size_t n, bp_ipa** out) {
  if (!ctx || !out || !gx || !gy || !a || !b) return fail(ctx, BP_E_ARG, "bp_ipa_begin: null argument");
  if (variant != BP_IPA_BULLETPROOF && variant != BP_IPA_POLYCOMMIT) return fail(ctx, BP_E_ARG, "bp_ipa_begin: bad variant");
  if (variant == BP_IPA_BULLETPROOF && (!hx || !hy)) return fail(ctx, BP_E_ARG, "bp_ipa_begin: hs required");
  if (n == 0 || (n & (n - 1))) return fail(ctx, BP_E_ARG, "bp_ipa_begin: n must be a power of two");
  if (!curve_ok(curve)) return fail(ctx, BP_E_CURVE, "unsupported curve");
  Guard g(ctx);
  return ipa_begin_common(ctx, curve, variant, nullptr, 0, 0, gx, gy, ginf, hx, hy, hinf, a, b, n, out);
}

int bp_ipa_begin_resident(bp_ctx* ctx, const bp_bases* bases, int variant, size_t g_offset, size_t h_offset,
                          const uint64_t* a, const uint64_t* b, size_t n, bp_ipa** out) {
  if (!ctx || !out || !bases || !a || !b) return fail(ctx, BP_E_ARG, "bp_ipa_begin_resident: null argument");
  if (bases->ctx != ctx) return fail(ctx, BP_E_ARG, "bp_ipa_begin_resident: bases belong to another context");
  if (variant != BP_IPA_BULLETPROOF && variant != BP_IPA_POLYCOMMIT) return fail(ctx, BP_E_ARG, "bp_ipa_begin_resident: bad variant");
  if (n == 0 || (n & (n - 1))) return fail(ctx, BP_E_ARG, "bp_ipa_begin_resident: n must be a power of two");
  if (g_offset > bases->n || n > bases->n - g_offset) return fail(ctx, BP_E_ARG, "bp_ipa_begin_resident: g range out of bounds");
  if (variant == BP_IPA_BULLETPROOF && (h_offset > bases->n || n > bases->n - h_offset))
    return fail(ctx, BP_E_ARG, "bp_ipa_begin_resident: h range out of bounds");
  Guard g(ctx);
  return ipa_begin_common(ctx, bases->curve, variant, bases, g_offset, h_offset, nullptr, nullptr, nullptr, nullptr,
                          nullptr, nullptr, a, b, n, out);
}

int bp_ipa_begin_resident_powers(bp_ctx* ctx, const bp_bases* bases, int variant, size_t g_offset, size_t h_offset,
                                 const uint64_t* a, const uint64_t x[4], size_t n, bp_ipa** out) {
  if (!ctx || !out || !bases || !a || !x) return fail(ctx, BP_E_ARG, "bp_ipa_begin_resident_powers: null argument");
  if (bases->ctx != ctx) return fail(ctx, BP_E_ARG, "bp_ipa_begin_resident_powers: bases belong to another context");
  if (variant != BP_IPA_BULLETPROOF && variant != BP_IPA_POLYCOMMIT)
    return fail(ctx, BP_E_ARG, "bp_ipa_begin_resident_powers: bad variant");
  if (n == 0 || (n & (n - 1))) return fail(ctx, BP_E_ARG, "bp_ipa_begin_resident_powers: n must be a power of two");
  if (g_offset > bases->n || n > bases->n - g_offset)
    return fail(ctx, BP_E_ARG, "bp_ipa_begin_resident_powers: g range out of bounds");
  if (variant == BP_IPA_BULLETPROOF && (h_offset > bases->n || n > bases->n - h_offset))
    return fail(ctx, BP_E_ARG, "bp_ipa_begin_resident_powers: h range out of bounds");
  Guard g(ctx);
  return ipa_begin_common(ctx, bases->curve, variant, bases, g_offset, h_offset, nullptr, nullptr, nullptr, nullptr,
                          nullptr, nullptr, a, nullptr, n, out, x);
}

int bp_ipa_scale_h(bp_ipa* s, const uint64_t* scale) {
  if (!s || !scale) return BP_E_ARG;
  bp_ctx* ctx = s->ctx;
  Guard g(ctx);  // session state is read under the context's lock
  if (s->variant != BP_IPA_BULLETPROOF) return fail(ctx, BP_E_ARG, "bp_ipa_scale_h: session has no H vector");
  if (s->n != s->n0) return fail(ctx, BP_E_STATE, "bp_ipa_scale_h: only before the first fold");
  // ch *= scale: from here on the session behaves as if H_i had been scale_i * H_i
  s->h_scaled = true;
  HIPCHK(ctx, hipMemcpyAsync(s->eh, scale, s->n0 * 32, hipMemcpyHostToDevice, ctx->stream));
  hipError_t e = launch_vec_op(s->curve, BP_VEC_HADAMARD, s->ch, s->eh, FeArg{}, s->n0, s->ch, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "bp_ipa_scale_h", e);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return BP_OK;
}

int bp_ipa_set_u(bp_ipa* s, const uint64_t u_xy[8]) {
  if (!s) return BP_E_ARG;
  bp_ctx* ctx = s->ctx;
  Guard g(ctx);
  s->has_u = false;
  s->u_scaled = false;
  s->prepped = false;  // the scalars of u are part of what a fold prepares
  if (!u_xy) return BP_OK;
  HIPCHK(ctx, hipMemcpyAsync(s->u_xy, u_xy, 64, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  s->has_u = true;
  return BP_OK;
}

int bp_ipa_set_u_resident(bp_ipa* s, size_t u_index, const uint64_t factor[4]) {
  if (!s || !factor) return BP_E_ARG;
  bp_ctx* ctx = s->ctx;
  Guard g(ctx);
  if (!s->res) return fail(ctx, BP_E_STATE, "bp_ipa_set_u_resident: session was not begun over a registered base set");
  if (u_index >= s->res->n) return fail(ctx, BP_E_ARG, "bp_ipa_set_u_resident: index outside the base set");
  // x|y side by side for plans that read the point itself
  HIPCHK(ctx, hipMemcpyAsync(s->u_xy, s->res->x + u_index * 8, 32, hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(s->u_xy + 8, s->res->y + u_index * 8, 32, hipMemcpyDeviceToDevice, ctx->stream));
  uint8_t inf = 0;
  HIPCHK(ctx, hipMemcpyAsync(&inf, s->res->inf + u_index, 1, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  s->has_u = s->u_scaled = false;
  s->prepped = false;
  if (inf) return BP_OK;  // a multiple of the identity adds nothing
  s->u_index = u_index;
  s->u_factor = to_arg(factor);
  s->has_u = s->u_scaled = true;
  return BP_OK;
}

size_t bp_ipa_len(const bp_ipa* s) { return s ? s->n : 0; }

int bp_ipa_round_lr(bp_ipa* s, uint64_t l_xyz[12], uint64_t r_xyz[12], uint64_t c_l[4], uint64_t c_r[4]) {
  if (!s || !l_xyz || !r_xyz || !c_l || !c_r) return BP_E_ARG;
  bp_ctx* ctx = s->ctx;
  Guard g(ctx);
  if (s->n < 2) return fail(ctx, BP_E_STATE, "bp_ipa_round_lr: vectors already folded to length 1");
  const size_t h = s->n / 2, n0 = s->n0;
  const bool has_h = s->variant == BP_IPA_BULLETPROOF;
  const u32 *a_lo = s->a, *a_hi = s->a + h * 8, *b_lo = s->b, *b_hi = s->b + h * 8;
  // c_l = <a_lo, b_hi>, c_r = <a_hi, b_lo> (src/ipa/mod.rs:42,58); they stay on the device as
  // the scalars of u when one is set
  // dots: partial sums of the two products, then c_l, c_r, factor*c_l, factor*c_r
  u32* part_l = s->dots;
  u32* part_r = part_l + (size_t)kDotBlocks * 8;
  u32 *d_cl = part_r + (size_t)kDotBlocks * 8, *d_cr = d_cl + 8, *d_scaled = d_cl + 16;
  hipError_t e;
  u32* host_dots = s->host_slot >= 0 ? (u32*)((char*)ctx->host_slots + (size_t)s->host_slot * 64) : nullptr;
  if (s->prepped) {
    // the fold that led here left c_l, c_r and the MSM scalars of this round (k_ipa_fold_prep)
    s->prepped = false;
  } else if (h <= kIpaPrepMaxHalf) {
    // short vectors: a round is a chain of launches, so dots, scaling and MSM scalars share one
    const bool scaled = s->has_u && s->u_scaled;
    e = launch_ipa_round_prep(s->curve, s->a, s->b, s->cg, has_h ? s->ch : nullptr, n0, h,
                              scaled ? s->u_factor : FeArg{}, scaled ? 1 : 0, s->eg, s->eh, d_cl, host_dots, ctx->stream);
    if (e != hipSuccess) return fail(ctx, BP_E_HIP, "ipa_round_prep", e);
    s->dots_on_host = host_dots != nullptr;
  } else {
    s->dots_on_host = false;
    e = launch_inner_product(s->curve, a_lo, b_hi, h, part_l, d_cl, ctx->stream);
    if (e == hipSuccess) e = launch_inner_product(s->curve, a_hi, b_lo, h, part_r, d_cr, ctx->stream);
    if (e == hipSuccess && s->has_u && s->u_scaled)
      e = launch_vec_op(s->curve, BP_VEC_SCALE, d_cl, nullptr, s->u_factor, 2, d_scaled, ctx->stream);
    if (e != hipSuccess) return fail(ctx, BP_E_HIP, "inner_product", e);
    e = launch_ipa_scalars(s->curve, s->a, s->b, s->cg, has_h ? s->ch : nullptr, n0, h, s->eg, s->eh, ctx->stream);
    if (e != hipSuccess) return fail(ctx, BP_E_HIP, "ipa_scalars", e);
  }
  // One pass computes both cross terms: group 0 = L, group 1 = R.
  //   L: G_hi with a_lo, H_lo with b_hi (src/ipa/mod.rs:44-54); R: G_lo with a_hi, H_hi with b_lo (:60-70)
  //   poly-commit: L = <g_hi, a_lo>, R = <g_lo, a_hi> (src/poly_commit/mod.rs:78,83)
  //   with u set: L += c_l*u, R += c_r*u (src/ipa/mod.rs:43,59; src/poly_commit/mod.rs:80,85)
  MsmPart pg{s->gx, s->gy, s->ginf, s->eg, n0};
  pg.sel_half = h; pg.group_hi = 0; pg.group_lo = 1;
  MsmPart ph{nullptr, nullptr, nullptr, nullptr, 0};
  if (has_h) {
    ph = MsmPart{s->hx, s->hy, s->hinf, s->eh, n0};
    ph.sel_half = h; ph.group_lo = 0; ph.group_hi = 1;
  }
  MsmExtra extra{{s->u_xy, s->u_xy}, {d_cl, d_cr}};
  if (s->u_scaled) {
    extra.s[0] = d_scaled;
    extra.s[1] = d_scaled + 8;
    extra.tab_index[0] = extra.tab_index[1] = s->u_index;
  }
  // window tables of the base set, when it has them and u (if any) is one of its points
  MsmTable table;
  const MsmTable* use_table = nullptr;
  if (s->res && s->res->tab && n0 * (has_h ? 2 : 1) + 2 <= kTableMaxPoints && (!s->has_u || s->u_scaled)) {
    table = MsmTable{s->res->tab, s->res->n};
    pg.tab_offset = s->g_off;
    ph.tab_offset = s->h_off;
    use_table = &table;
  }
  MsmPlan plan;
  u32* d_win = nullptr;
  // the window sums arrive in pinned host memory straight from the last launch, c_l and c_r from
  // the launch that took them (or, for long vectors and when no slot was free, by one copy)
  CHK(msm_device(ctx, s->curve, pg, ph, &plan, &d_win, 2, s->has_u ? &extra : nullptr, use_table, true, 64));
  char* host = (char*)ctx->pinned;
  const char* host_c = host + plan.win_bytes;
  if (s->dots_on_host)
    host_c = (const char*)host_dots;
  else
    HIPCHK(ctx, hipMemcpyAsync(host + plan.win_bytes, d_cl, 64, hipMemcpyDeviceToHost, ctx->stream));  // c_l, c_r
  s->dots_on_host = false;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  msm_collect_times(ctx, true);
  const uint64_t* win = (const uint64_t*)host;
  {
    const int curve = s->curve, gw = plan.group_windows, c = plan.finish_c;
    const uint64_t* win_r = win + (size_t)gw * 16;
    const bool helped = ctx->worker.submit([=] { finish_windows(curve, win_r, gw, c, r_xyz); });
    finish_windows(curve, win, gw, c, l_xyz);
    if (helped)
      ctx->worker.wait();
    else
      finish_windows(curve, win_r, gw, c, r_xyz);
  }
  memcpy(c_l, host_c, 32);
  memcpy(c_r, host_c + 32, 32);
  return BP_OK;
}

int bp_ipa_round_fold(bp_ipa* s, const uint64_t x[4], const uint64_t x_inv[4]) {
  if (!s || !x || !x_inv) return BP_E_ARG;
  bp_ctx* ctx = s->ctx;
  if (s->n < 2) return fail(ctx, BP_E_STATE, "bp_ipa_round_fold: vectors already folded to length 1");
  Guard g(ctx);
  size_t h = s->n / 2;
  FeArg xm = to_arg(x), xim = to_arg(x_inv);
  FeArg xc = to_canonical_fr(s->curve, x), xic = to_canonical_fr(s->curve, x_inv);
  hipError_t e;
  // G' = x^-1*G_lo + x*G_hi (both variants: src/ipa/mod.rs:81, src/poly_commit/mod.rs:102) and
  // H' = x*H_lo + x^-1*H_hi (src/ipa/mod.rs:82) become coefficient updates of the original
  // generators: upper-half G entries and lower-half H entries gain x, the others x^-1.
  const bool has_h = s->variant == BP_IPA_BULLETPROOF;
  // The same launch folds the scalars; outputs overwrite the low halves in place (output i reads
  // only inputs i and i+h): a' = a_lo*x + a_hi*x^-1, b' = b_lo*x^-1 + b_hi*x (src/ipa/mod.rs:88-89;
  // src/poly_commit/mod.rs:96-97 writes the same pair as a_hi*u^-1 + a_lo*u, b_lo*u^-1 + b_hi*u)
  // After a few rounds over many generators the folded generators are computed once and replace
  // the original ones: every later round then costs an MSM over the current length instead of the
  // first one's.  m = n0 / h original generators per folded one; BP_IPA_REBASE_M sets it (0: never).
  static const size_t rebase_m = getenv("BP_IPA_REBASE_M") ? (size_t)atol(getenv("BP_IPA_REBASE_M")) : 16;
  static const size_t rebase_min = getenv("BP_IPA_REBASE_MIN") ? (size_t)atol(getenv("BP_IPA_REBASE_MIN")) : kIpaRebaseMinPoints;
  const bool rebase_due = rebase_m >= 2 && !s->h_scaled && s->n0 >= rebase_min && s->n0 / h >= rebase_m && h >= 2;
  if (rebase_due) {
    e = launch_ipa_round_fold(s->curve, s->a, s->b, s->cg, has_h ? s->ch : nullptr, s->n0, h, xm, xim, ctx->stream);
    if (e != hipSuccess) return fail(ctx, BP_E_HIP, "ipa_round_fold", e);
    s->x_canon.push_back(xc);
    s->xinv_canon.push_back(xic);
    s->n = h;
    return ipa_rebase(s);
  }
  if (h >= 2 && h / 2 <= kIpaPrepMaxHalf && h * 2 <= s->n0) {
    // short vectors, and not the first fold (a2, b2 hold n0/2 elements): fold out of place and
    // prepare the next round in the same launch; bp_ipa_round_lr then goes straight to its MSM
    const bool scaled = s->has_u && s->u_scaled;
    u32* d_cl = s->dots + (size_t)2 * kDotBlocks * 8;
    u32* host_dots = s->host_slot >= 0 ? (u32*)((char*)ctx->host_slots + (size_t)s->host_slot * 64) : nullptr;
    e = launch_ipa_fold_prep(s->curve, s->a, s->b, s->a2, s->b2, s->cg, has_h ? s->ch : nullptr, s->n0, h, xm, xim,
                             scaled ? s->u_factor : FeArg{}, scaled ? 1 : 0, s->eg, s->eh, d_cl, host_dots, ctx->stream);
    if (e != hipSuccess) return fail(ctx, BP_E_HIP, "ipa_fold_prep", e);
    s->dots_on_host = host_dots != nullptr;
    std::swap(s->a, s->a2);
    std::swap(s->b, s->b2);
    s->prepped = true;
  } else {
    e = launch_ipa_round_fold(s->curve, s->a, s->b, s->cg, has_h ? s->ch : nullptr, s->n0, h, xm, xim, ctx->stream);
    if (e != hipSuccess) return fail(ctx, BP_E_HIP, "ipa_round_fold", e);
  }
  s->x_canon.push_back(xc);
  s->xinv_canon.push_back(xic);
  s->n = h;
  return BP_OK;
}

int bp_ipa_read(bp_ipa* s, uint64_t* gx, uint64_t* gy, uint8_t* ginf, uint64_t* hx, uint64_t* hy, uint8_t* hinf,
                uint64_t* a, uint64_t* b) {
  if (!s) return BP_E_ARG;
  bp_ctx* ctx = s->ctx;
  Guard g(ctx);
  const size_t n = s->n, n0 = s->n0;
  const bool has_h = s->variant == BP_IPA_BULLETPROOF;
  const bool want_g = gx || gy || ginf, want_h = hx || hy || hinf;
  if (want_h && !has_h) return fail(ctx, BP_E_ARG, "bp_ipa_read: session has no H vector");
  const u32 *sgx = s->gx, *sgy = s->gy, *shx = s->hx, *shy = s->hy;
  const uint8_t *sgi = s->ginf, *shi = s->hinf;
  if ((want_g || want_h) && n == 1 && n0 > 1) {
    // fully folded: G = <cg, G0>, H = <ch, H0>, two sums in one MSM pass
    MsmPart pg{s->gx, s->gy, s->ginf, s->cg, n0};
    MsmPart ph{nullptr, nullptr, nullptr, nullptr, 0};
    if (has_h) {
      ph = MsmPart{s->hx, s->hy, s->hinf, s->ch, n0};
      ph.group_lo = ph.group_hi = 1;
    }
    MsmPlan plan;
    u32* d_win = nullptr;
    MsmTable table;
    const MsmTable* use_table = nullptr;
    if (s->res && s->res->tab && n0 * (has_h ? 2 : 1) <= kTableMaxPoints) {
      table = MsmTable{s->res->tab, s->res->n};
      pg.tab_offset = s->g_off;
      ph.tab_offset = s->h_off;
      use_table = &table;
    }
    CHK(msm_device(ctx, s->curve, pg, ph, &plan, &d_win, 2, nullptr, use_table, true));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const uint64_t* win = (const uint64_t*)ctx->pinned;
    uint64_t g_xyz[12], h_xyz[12];
    finish_windows(s->curve, win, plan.group_windows, plan.finish_c, g_xyz);
    finish_windows(s->curve, win + (size_t)plan.group_windows * 16, plan.group_windows, plan.finish_c, h_xyz);
    auto put = [](const uint64_t xyz[12], uint64_t* ox, uint64_t* oy, uint8_t* oinf) {
      bool inf = !(xyz[8] | xyz[9] | xyz[10] | xyz[11]);
      // affine outputs encode the identity as x = y = 0 with the flag set
      if (ox) { if (inf) memset(ox, 0, 32); else memcpy(ox, xyz, 32); }
      if (oy) { if (inf) memset(oy, 0, 32); else memcpy(oy, xyz + 4, 32); }
      if (oinf) *oinf = inf ? 1 : 0;
    };
    put(g_xyz, gx, gy, ginf);
    if (has_h) put(h_xyz, hx, hy, hinf);
    gx = gy = hx = hy = nullptr;
    ginf = hinf = nullptr;
  } else if ((want_g || want_h) && n < n0) {
    // partially folded: replay the folds over the stored challenges
    hipError_t e = hipSuccess;
    auto alloc = [&](void** p, size_t bytes) {
      if (!*p && e == hipSuccess) e = hipMalloc(p, bytes);
    };
    const size_t cap = n0 / 2;
    alloc((void**)&s->fgx, cap * 32); alloc((void**)&s->fgy, cap * 32); alloc((void**)&s->fginf, cap);
    if (has_h) { alloc((void**)&s->fhx, cap * 32); alloc((void**)&s->fhy, cap * 32); alloc((void**)&s->fhinf, cap); }
    if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? BP_E_OOM : BP_E_HIP, "bp_ipa_read: hipMalloc", e);
    size_t len = n0;
    for (size_t j = 0; j < s->x_canon.size(); j++) {
      const size_t h = len / 2;
      const FeArg xc = s->x_canon[j], xic = s->xinv_canon[j];
      // G' = x^-1*G_lo + x*G_hi ; H' = x*H_lo + x^-1*H_hi (src/ipa/mod.rs:81-82)
      FoldJob jg{sgx, sgy, sgi, sgx + h * 8, sgy + h * 8, sgi + h, s->fgx, s->fgy, s->fginf, xic, xc};
      FoldJob jh{};
      if (has_h) jh = FoldJob{shx, shy, shi, shx + h * 8, shy + h * 8, shi + h, s->fhx, s->fhy, s->fhinf, xc, xic};
      e = launch_fold_generators2(s->curve, jg, jh, h, ctx->stream);
      if (e != hipSuccess) return fail(ctx, BP_E_HIP, "fold_generators", e);
      sgx = s->fgx; sgy = s->fgy; sgi = s->fginf;
      shx = s->fhx; shy = s->fhy; shi = s->fhinf;
      len = h;
    }
  }
  struct { void* h; const void* d; size_t bytes; } cp[] = {
      {gx, sgx, n * 32}, {gy, sgy, n * 32}, {ginf, sgi, n}, {hx, shx, n * 32},
      {hy, shy, n * 32}, {hinf, shi, n}, {a, s->a, n * 32}, {b, s->b, n * 32}};
  for (auto& c : cp) {
    if (!c.h) continue;
    HIPCHK(ctx, hipMemcpyAsync(c.h, c.d, c.bytes, hipMemcpyDeviceToHost, ctx->stream));
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return BP_OK;
}

}  // extern "C"

// ---- batches of aggregated range proofs ---------------------------------------------------------

struct bp_range_batch {
  bp_ctx* ctx = nullptr;
  const bp_bases* bases = nullptr;
  int curve = 0, batch = 0;
  size_t m = 0, n = 0;
  unsigned n_bits = 0;
  RangeBatchTabs tl{};
  void* block = nullptr;   // one device allocation behind every pointer below
  char* host = nullptr;    // pinned: results the last launch of a phase writes itself
  uint64_t* values = nullptr;                   // [B][m]
  uint64_t* seeds = nullptr;                    // [B]
  u32 *s_lr = nullptr, *sa = nullptr;           // [B][2n]: s_L || s_R, a_L || a_R
  u32 *l0 = nullptr, *r0 = nullptr, *r1 = nullptr;                   // [B][n]
  u32 *a = nullptr, *b = nullptr, *a2 = nullptr, *b2 = nullptr;      // [B][n]: the arguments' vectors and the other side of the fold
  u32 *cg = nullptr, *ch = nullptr, *eg = nullptr, *eh = nullptr;    // [B][n]
  u32 *tabs = nullptr, *pow2 = nullptr;         // [B][tl.row], [n_bits]
  u32 *yz = nullptr;                            // [3][B]: y, y^-1, z
  u32 *sc0 = nullptr, *sc1 = nullptr;           // [B]: one scalar per proof (alpha / x / challenge, rho / factor / inverse)
  u32 *factors = nullptr;                       // [B]: the factor of u
  u32 *tsc = nullptr;                           // [B][4]: t1, tao1, t2, tao2
  u32 *dots = nullptr;                          // [B][4]
  u32 *partials = nullptr;                      // [B][3 * ceil(n / 256)]
  u32* flag = nullptr;                          // one word
  size_t cur_n = 0;
  bool prepped = false, has_poly = false, in_ipa = false;
  // After reb_rounds rounds the sessions' generators are replaced by the folded ones (rangeb_rebase): `row` is
  // then the length of a row of the per-proof arrays (n before, n >> reb_rounds after), and the rounds read
  // the batch's own points (B rows of 2 row + 1: G' | H' | u) and their window tables.
  int reb_rounds = 0;
  int reb_bits = 4;  // digit width of the tables over the batch's own points
  bool rebased = false;
  size_t row = 0;
  void* reb_block = nullptr;
  u32 *reb_x = nullptr, *reb_y = nullptr, *reb_tab = nullptr, *reb_scratch = nullptr, *reb_cols = nullptr, *reb_prog = nullptr;
  uint8_t* reb_inf = nullptr;
  size_t reb_prog_words = 0;
};

namespace {

// affine (Z = 1) forms of `count` XYZZ sums (boundary form, 16 u64 each) with one shared inversion
template <class Fq>
void normalize_xyzz_batch(const uint64_t* xyzz, size_t count, uint64_t* out_xyz) {
  using namespace host;
  std::vector<HFe> pre(count);
  HFe run = h_one<Fq>();
  const HFe one = h_one<Fq>();
  for (size_t i = 0; i < count; i++) {
    HFe zzz;
    memcpy(zzz.v, xyzz + 16 * i + 12, 32);
    pre[i] = run;
    if (!h_is_zero(zzz)) run = h_mul<Fq>(run, zzz);
  }
  HFe inv = h_inv_kaliski<Fq>(run);
  for (size_t i = count; i-- > 0;) {
    HFe x, y, zz, zzz;
    memcpy(x.v, xyzz + 16 * i, 32); memcpy(y.v, xyzz + 16 * i + 4, 32);
    memcpy(zz.v, xyzz + 16 * i + 8, 32); memcpy(zzz.v, xyzz + 16 * i + 12, 32);
    uint64_t* o = out_xyz + 12 * i;
    if (h_is_zero(zzz)) {  // the identity: (1, 1, 0)
      memcpy(o, one.v, 32); memcpy(o + 4, one.v, 32); memset(o + 8, 0, 32);
      continue;
    }
    const HFe i3 = h_mul<Fq>(inv, pre[i]);  // 1 / zzz
    inv = h_mul<Fq>(inv, zzz);
    const HFe t = h_mul<Fq>(zz, i3);        // 1 / zz = (zz / zzz)^2
    const HFe ax = h_mul<Fq>(x, h_sqr<Fq>(t)), ay = h_mul<Fq>(y, i3);
    memcpy(o, ax.v, 32); memcpy(o + 4, ay.v, 32); memcpy(o + 8, one.v, 32);
  }
}

// the window sums of a batched two-group table plan added per (proof, group) on the device into the batch's
// pinned buffer, then normalised on the host: out[b][g] (12 u64 each)
int rangeb_collect(bp_range_batch* rb, const MsmPlan& plan, const u32* d_win, uint64_t* out) {
  bp_ctx* ctx = rb->ctx;
  const int sums = rb->batch * 2;
  hipError_t e = launch_window_group_sums(rb->curve, d_win, plan.group_windows, sums, (u32*)rb->host, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "window_group_sums", e);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (rb->curve == BP_SECP256K1)
    normalize_xyzz_batch<Secp256k1Fq>((const uint64_t*)rb->host, (size_t)sums, out);
  else
    normalize_xyzz_batch<VestaFq>((const uint64_t*)rb->host, (size_t)sums, out);
  return BP_OK;
}

// Replaces the generators of every session of the batch by the folded ones of the current length (what
// ipa_rebase does for one session over window tables).  The H coefficients of a range proof carry y^-k,
// which differs from column to column: ch[i + cur t] = y^-i * ch[cur t], so the folded H are built from
// column 0's coefficients and y^-i stays behind as the new coefficient of column i.
int rangeb_rebase(bp_range_batch* rb) {
  bp_ctx* ctx = rb->ctx;
  const size_t B = (size_t)rb->batch, n0 = rb->n, cur = rb->cur_n, m = n0 / cur, pts = 2 * cur + 1;
  const bp_bases* bs = rb->bases;
  if (!rb->reb_block) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
    const size_t o_x = take(B * pts * 32), o_y = take(B * pts * 32), o_inf = take(B * pts);
    const size_t tab_windows = 256 / (size_t)rb->reb_bits;
    const size_t o_tab = take(tab_windows * B * pts * 64), o_scr = take(tab_windows * B * pts * 128);
    const size_t o_cols = take(B * 2 * m * 32), o_prog = take(B * 2 * kRebaseTabProgMax * sizeof(u32));
    hipError_t e = hipMalloc(&rb->reb_block, off);
    if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? BP_E_OOM : BP_E_HIP, "rangeb_rebase: allocation", e);
    char* base = (char*)rb->reb_block;
    rb->reb_x = (u32*)(base + o_x); rb->reb_y = (u32*)(base + o_y); rb->reb_inf = (uint8_t*)(base + o_inf);
    rb->reb_tab = (u32*)(base + o_tab); rb->reb_scratch = (u32*)(base + o_scr);
    rb->reb_cols = (u32*)(base + o_cols); rb->reb_prog = (u32*)(base + o_prog);
    rb->reb_prog_words = B * 2 * kRebaseTabProgMax;
  }
  // column 0 of every session's coefficients
  hipError_t e = launch_ipab_gather_cols(rb->cg, rb->ch, (u32)n0, (u32)cur, rb->batch, rb->reb_cols, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "ipab_gather_cols", e);
  const size_t col_bytes = B * 2 * m * 32;
  CHK(reserve_pinned(ctx, col_bytes));
  HIPCHK(ctx, hipMemcpyAsync(ctx->pinned, rb->reb_cols, col_bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  const uint64_t* coef = (const uint64_t*)ctx->pinned;
  // lanes per output: as many as fill the machine about once
  int log_s = 0;
  while (log_s < kRebaseTabMaxLogS && ((B * 2 * cur) << (log_s + 1)) <= 65536) log_s++;
  if (const char* ev = getenv("BP_IPA_REBASE_LOG_S")) log_s = std::min(kRebaseTabMaxLogS, std::max(0, atoi(ev)));
  const size_t lanes = (size_t)1 << log_s;
  const int longest = (int)rebase_tab_list_len(m, log_s);
  if ((size_t)longest * lanes > kRebaseTabProgMax) return fail(ctx, BP_E_STATE, "rangeb_rebase: step lists too long");
  const size_t per_set = (size_t)longest * lanes;
  std::vector<u32> prog(B * 2 * per_set);
  for (size_t q = 0; q < B * 2; q++)
    if (!rebase_tab_lists(rb->curve, coef + q * m * 4, m, log_s, &prog[q * per_set]))
      return fail(ctx, BP_E_STATE, "rangeb_rebase: step lists");
  HIPCHK(ctx, hipMemcpyAsync(rb->reb_prog, prog.data(), prog.size() * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
  RebaseTabSet ts[2] = {{0u, bs->inf, rb->reb_prog, rb->reb_x, rb->reb_y, rb->reb_inf},
                        {(u32)n0, bs->inf, rb->reb_prog + per_set, rb->reb_x + cur * 8, rb->reb_y + cur * 8, rb->reb_inf + cur}};
  for (RebaseTabSet& t : ts) { t.prog_z_stride = 2 * per_set; t.out_z_stride = pts; }
  e = launch_ipa_rebase_tab(rb->curve, bs->tab, bs->n, ts, 2, cur, log_s, longest, ctx->stream, rb->batch);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "ipa_rebase_tab", e);
  // witnesses to rows of cur elements, coefficients 1 and y^-i, u as every session's last point
  e = launch_ipab_rebase_pack(rb->curve, rb->a, rb->b, (u32)n0, (u32)cur, rb->batch, rb->tabs, rb->tl, rb->a2, rb->b2, rb->cg, rb->ch,
                              bs->x + 2 * n0 * 8, bs->y + 2 * n0 * 8, rb->reb_x, rb->reb_y, rb->reb_inf, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "ipab_rebase_pack", e);
  std::swap(rb->a, rb->a2);
  std::swap(rb->b, rb->b2);
  e = launch_window_tables_inv1(rb->curve, rb->reb_x, rb->reb_y, rb->reb_inf, B * pts, rb->reb_bits, 256 / rb->reb_bits, rb->reb_scratch,
                                rb->reb_tab, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "window_tables_inv1", e);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // `prog` goes out of scope
  rb->row = cur;
  rb->rebased = true;
  rb->prepped = false;
  return BP_OK;
}

}  // namespace

extern "C" {

void bp_range_batch_free(bp_range_batch* rb) {
  if (!rb) return;
  Guard g(rb->ctx);
  (void)hipStreamSynchronize(rb->ctx->stream);
  if (rb->block) (void)hipFree(rb->block);
  if (rb->reb_block) (void)hipFree(rb->reb_block);
  if (rb->host) (void)hipHostFree(rb->host);
  delete rb;
}

int bp_range_batch_begin(bp_ctx* ctx, const bp_bases* bases, size_t batch, size_t m, unsigned n_bits, bp_range_batch** out) {
  if (!ctx || !bases || !out) return fail(ctx, BP_E_ARG, "bp_range_batch_begin: null argument");
  if (bases->ctx != ctx) return fail(ctx, BP_E_ARG, "bp_range_batch_begin: bases belong to another context");
  const size_t n = m * (size_t)n_bits;
  if (batch < 1 || batch > (size_t)kMsmMaxBatch || m == 0 || n_bits == 0 || n_bits > 64 || n < 2 || (n & (n - 1)) || n > 4096)
    return fail(ctx, BP_E_ARG, "bp_range_batch_begin: 1 <= batch <= 128, m * n_bits a power of two in 2..4096");
  if (bases->n < 2 * n + 3) return fail(ctx, BP_E_ARG, "bp_range_batch_begin: base set shorter than gs | hs | u | g | h (2 m n_bits + 3)");
  if (!bases->tab) return fail(ctx, BP_E_STATE, "bp_range_batch_begin: the base set has no window tables (bp_bases_precompute)");
  Guard g(ctx);
  bp_range_batch* rb = new (std::nothrow) bp_range_batch();
  if (!rb) return fail(ctx, BP_E_OOM, "host allocation");
  rb->ctx = ctx; rb->bases = bases; rb->curve = bases->curve; rb->batch = (int)batch; rb->m = m; rb->n = n; rb->n_bits = n_bits;
  rb->row = n;
  {
    // rounds before the generators are replaced: 4 (a sixteenth of the points remains) for arguments that have
    // at least five rounds left after them and a batch whose rounds are bound by their volume; BP_RANGE_BATCH_REBASE=0 turns it off
    int rounds = 4;
    if (const char* ev = getenv("BP_RANGE_BATCH_REBASE")) rounds = atoi(ev);
    if (rounds < 1 || rounds > 8 || (n >> rounds) < 32 || batch < 4) rounds = 0;
    rb->reb_rounds = rounds;
    // BP_RANGE_BATCH_TAB_BITS=8 keeps byte digits over the replaced generators (comparison runs)
    if (const char* ev = getenv("BP_RANGE_BATCH_TAB_BITS")) rb->reb_bits = atoi(ev) == 8 ? 8 : 4;
  }
  // powers of y by two tables of about sqrt(n) entries
  u32 lo = 1;
  while ((size_t)lo * lo < n) lo <<= 1;
  const u32 hi = (u32)((n + lo - 1) / lo);
  RangeBatchTabs& tl = rb->tl;
  tl.lo = lo; tl.hi = hi;
  tl.zpow = 0; tl.ylo = (u32)m; tl.yhi = tl.ylo + lo; tl.yilo = tl.yhi + hi; tl.yihi = tl.yilo + lo; tl.row = tl.yihi + hi;
  const size_t B = batch, blocks = (n + 255) / 256;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
  const size_t o_values = take(B * m * 8), o_seeds = take(B * 8), o_slr = take(B * 2 * n * 32), o_sa = take(B * 2 * n * 32);
  const size_t o_l0 = take(B * n * 32), o_r0 = take(B * n * 32), o_r1 = take(B * n * 32);
  const size_t o_a = take(B * n * 32), o_b = take(B * n * 32), o_a2 = take(B * n * 32), o_b2 = take(B * n * 32);
  const size_t o_cg = take(B * n * 32), o_ch = take(B * n * 32), o_eg = take(B * n * 32), o_eh = take(B * n * 32);
  const size_t o_tabs = take(B * tl.row * 32), o_pow2 = take((size_t)n_bits * 32), o_yz = take(B * 96);
  const size_t o_sc0 = take(B * 32), o_sc1 = take(B * 32), o_fac = take(B * 32), o_tsc = take(B * 128), o_dots = take(B * 128);
  const size_t o_part = take(B * 3 * blocks * 32), o_flag = take(256);
  hipError_t e = hipMalloc(&rb->block, off);
  if (e == hipSuccess) e = hipHostMalloc((void**)&rb->host, B * 2 * 128 + B * 64, hipHostMallocDefault);
  if (e != hipSuccess) {
    if (rb->block) (void)hipFree(rb->block);
    delete rb;
    return fail(ctx, e == hipErrorOutOfMemory ? BP_E_OOM : BP_E_HIP, "bp_range_batch_begin", e);
  }
  char* base = (char*)rb->block;
  auto at = [&](size_t o) { return (u32*)(base + o); };
  rb->values = (uint64_t*)(base + o_values); rb->seeds = (uint64_t*)(base + o_seeds);
  rb->s_lr = at(o_slr); rb->sa = at(o_sa); rb->l0 = at(o_l0); rb->r0 = at(o_r0); rb->r1 = at(o_r1);
  rb->a = at(o_a); rb->b = at(o_b); rb->a2 = at(o_a2); rb->b2 = at(o_b2);
  rb->cg = at(o_cg); rb->ch = at(o_ch); rb->eg = at(o_eg); rb->eh = at(o_eh);
  rb->tabs = at(o_tabs); rb->pow2 = at(o_pow2); rb->yz = at(o_yz); rb->sc0 = at(o_sc0); rb->sc1 = at(o_sc1);
  rb->factors = at(o_fac); rb->tsc = at(o_tsc); rb->dots = at(o_dots); rb->partials = at(o_part); rb->flag = at(o_flag);
  // 2^i, i < n_bits
  std::vector<uint64_t> p2(4 * (size_t)n_bits);
  {
    host::HFe v = rb->curve == BP_SECP256K1 ? host::h_one<Secp256k1Fr>() : host::h_one<VestaFr>();
    for (unsigned i = 0; i < n_bits; i++) {
      memcpy(&p2[4 * i], v.v, 32);
      v = rb->curve == BP_SECP256K1 ? host::h_dbl<Secp256k1Fr>(v) : host::h_dbl<VestaFr>(v);
    }
  }
  int rc = stage_h2d(ctx, rb->pow2, p2.data(), p2.size() * 8);
  if (rc == BP_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(ctx, BP_E_HIP, "bp_range_batch_begin: sync");
  if (rc != BP_OK) {
    (void)hipFree(rb->block);
    (void)hipHostFree(rb->host);
    delete rb;
    return rc;
  }
  *out = rb;
  return BP_OK;
}

size_t bp_range_batch_size(const bp_range_batch* rb) { return rb ? (size_t)rb->batch : 0; }

int bp_range_batch_commit(bp_range_batch* rb, const uint64_t* values, const uint64_t* s_lr, const uint64_t* seeds,
                          unsigned first_draw, const uint64_t* alpha, const uint64_t* rho, uint64_t* out_as_xyz) {
  if (!rb || !values || (!s_lr && !seeds) || !alpha || !rho || !out_as_xyz) return BP_E_ARG;
  bp_ctx* ctx = rb->ctx;
  Guard g(ctx);
  const size_t B = (size_t)rb->batch, n = rb->n;
  rb->has_poly = rb->in_ipa = false;
  rb->rebased = false;
  rb->row = n;
  CHK(stage_h2d(ctx, rb->values, values, B * rb->m * 8));
  CHK(stage_h2d(ctx, rb->sc0, alpha, B * 32));
  CHK(stage_h2d(ctx, rb->sc1, rho, B * 32));
  hipError_t e;
  if (s_lr) {
    HIPCHK(ctx, hipMemcpyAsync(rb->s_lr, s_lr, B * 2 * n * 32, hipMemcpyHostToDevice, ctx->stream));
  } else {
    CHK(stage_h2d(ctx, rb->seeds, seeds, B * 8));
    HIPCHK(ctx, hipMemsetAsync(rb->flag, 0, 4, ctx->stream));
    e = launch_rangeb_blinders(rb->curve, rb->seeds, first_draw, (u32)(2 * n), rb->batch, rb->s_lr, rb->flag, ctx->stream);
    if (e != hipSuccess) return fail(ctx, BP_E_HIP, "rangeb_blinders", e);
  }
  e = launch_rangeb_commit_scalars(rb->curve, rb->values, (u32)rb->m, rb->n_bits, (u32)n, rb->batch, rb->sa, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "rangeb_commit_scalars", e);
  // A_b = <a_L || a_R, gs || hs> + alpha_b h, S_b = <s_L || s_R, gs || hs> + rho_b h (src/range/aggregate.rs:80-83) as
  // groups 0 and 1 of every proof's pass over the same 2n points; h rides along as each group's extra point
  const bp_bases* bs = rb->bases;
  MsmPart p0{bs->x, bs->y, bs->inf, rb->sa, 2 * n};
  MsmPart p1{bs->x, bs->y, bs->inf, rb->s_lr, 2 * n};
  p1.group_lo = p1.group_hi = 1;
  p0.batch_stride = p1.batch_stride = 2 * n * 8;
  MsmExtra extra{{bs->x, bs->x}, {rb->sc0, rb->sc1}};
  extra.tab_index[0] = extra.tab_index[1] = 2 * n + 2;
  extra.batch_stride[0] = extra.batch_stride[1] = 8;
  MsmTable table{bs->tab, bs->n};
  MsmPlan plan;
  u32* d_win = nullptr;
  CHK(msm_device(ctx, rb->curve, p0, p1, &plan, &d_win, 2, &extra, &table, false, 0, rb->batch));
  CHK(rangeb_collect(rb, plan, d_win, out_as_xyz));
  if (!s_lr) {
    uint32_t rej = 0;
    HIPCHK(ctx, hipMemcpyAsync(&rej, rb->flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (rej) return fail(ctx, BP_E_STATE, "bp_range_batch_commit: a generated blinder was not below the modulus; pass the blinders instead");
  }
  return BP_OK;
}

int bp_range_batch_poly(bp_range_batch* rb, const uint64_t* y, const uint64_t* y_inv, const uint64_t* z, uint64_t* out_t1t2) {
  if (!rb || !y || !y_inv || !z || !out_t1t2) return BP_E_ARG;
  bp_ctx* ctx = rb->ctx;
  Guard g(ctx);
  const size_t B = (size_t)rb->batch;
  CHK(stage_h2d(ctx, rb->yz, y, B * 32));
  CHK(stage_h2d(ctx, rb->yz + B * 8, y_inv, B * 32));
  CHK(stage_h2d(ctx, rb->yz + 2 * B * 8, z, B * 32));
  hipError_t e = launch_rangeb_tables(rb->curve, rb->yz, rb->tl, (u32)rb->m, rb->batch, rb->tabs, ctx->stream);
  if (e == hipSuccess)
    e = launch_rangeb_poly(rb->curve, rb->values, (u32)rb->m, rb->n_bits, (u32)rb->n, rb->batch, rb->tabs, rb->tl, rb->pow2, rb->s_lr,
                           rb->yz + 2 * B * 8, rb->l0, rb->r0, rb->r1, rb->partials, (u32*)rb->host, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "rangeb_poly", e);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(out_t1t2, rb->host, B * 64);
  rb->has_poly = true;
  return BP_OK;
}

int bp_range_batch_tcommit(bp_range_batch* rb, const uint64_t* t_scalars, uint64_t* out_tt_xyz) {
  if (!rb || !t_scalars || !out_tt_xyz) return BP_E_ARG;
  bp_ctx* ctx = rb->ctx;
  Guard g(ctx);
  const size_t B = (size_t)rb->batch, n = rb->n;
  CHK(stage_h2d(ctx, rb->tsc, t_scalars, B * 128));
  // T1_b = t1 g + tao1 h, T2_b = t2 g + tao2 h (src/range/aggregate.rs:126-127): two points per group
  const bp_bases* bs = rb->bases;
  const size_t gi = 2 * n + 1;
  MsmPart p0{bs->x + gi * 8, bs->y + gi * 8, bs->inf + gi, rb->tsc, 2};
  MsmPart p1{bs->x + gi * 8, bs->y + gi * 8, bs->inf + gi, rb->tsc + 16, 2};
  p1.group_lo = p1.group_hi = 1;
  p0.tab_offset = p1.tab_offset = gi;
  p0.batch_stride = p1.batch_stride = 32;
  MsmTable table{bs->tab, bs->n};
  MsmPlan plan;
  u32* d_win = nullptr;
  CHK(msm_device(ctx, rb->curve, p0, p1, &plan, &d_win, 2, nullptr, &table, false, 0, rb->batch));
  return rangeb_collect(rb, plan, d_win, out_tt_xyz);
}

int bp_range_batch_eval(bp_range_batch* rb, const uint64_t* x, uint64_t* out_t_hat) {
  if (!rb || !x || !out_t_hat) return BP_E_ARG;
  bp_ctx* ctx = rb->ctx;
  Guard g(ctx);
  if (!rb->has_poly) return fail(ctx, BP_E_STATE, "bp_range_batch_eval: no vector polynomials (call bp_range_batch_poly first)");
  const size_t B = (size_t)rb->batch;
  CHK(stage_h2d(ctx, rb->sc0, x, B * 32));
  hipError_t e = launch_rangeb_eval(rb->curve, rb->l0, rb->s_lr, rb->r0, rb->r1, (u32)rb->n, rb->batch, rb->sc0, rb->a, rb->b,
                                    rb->partials, (u32*)rb->host, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "rangeb_eval", e);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(out_t_hat, rb->host, B * 32);
  return BP_OK;
}

int bp_range_batch_ipa_begin(bp_range_batch* rb, const uint64_t* u_factor) {
  if (!rb || !u_factor) return BP_E_ARG;
  bp_ctx* ctx = rb->ctx;
  Guard g(ctx);
  if (!rb->has_poly) return fail(ctx, BP_E_STATE, "bp_range_batch_ipa_begin: no evaluated vectors (bp_range_batch_poly, _eval first)");
  CHK(stage_h2d(ctx, rb->factors, u_factor, (size_t)rb->batch * 32));
  hipError_t e = launch_ipab_init(rb->curve, rb->tabs, rb->tl, (u32)rb->n, rb->batch, rb->cg, rb->ch, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "ipab_init", e);
  rb->cur_n = rb->n;
  rb->row = rb->n;
  rb->rebased = false;
  rb->prepped = false;
  rb->in_ipa = true;
  return BP_OK;
}

size_t bp_range_batch_ipa_len(const bp_range_batch* rb) { return rb && rb->in_ipa ? rb->cur_n : 0; }

int bp_range_batch_round_lr(bp_range_batch* rb, uint64_t* out_lr_xyz) {
  if (!rb || !out_lr_xyz) return BP_E_ARG;
  bp_ctx* ctx = rb->ctx;
  Guard g(ctx);
  if (!rb->in_ipa || rb->cur_n < 2) return fail(ctx, BP_E_STATE, "bp_range_batch_round_lr: no round left");
  if (rb->reb_rounds && !rb->rebased && rb->row == rb->n && (rb->cur_n << rb->reb_rounds) == rb->n) CHK(rangeb_rebase(rb));
  const size_t n = rb->row, h = rb->cur_n / 2;
  hipError_t e;
  if (!rb->prepped) {
    e = launch_ipab_round_prep(rb->curve, rb->a, rb->b, rb->cg, rb->ch, (u32)n, (u32)h, rb->batch, rb->factors, rb->eg, rb->eh,
                               rb->dots, ctx->stream);
    if (e != hipSuccess) return fail(ctx, BP_E_HIP, "ipab_round_prep", e);
  }
  rb->prepped = false;
  // L: G_hi with a_lo, H_lo with b_hi; R: G_lo with a_hi, H_hi with b_lo; u' = factor * u with factor c_l, factor c_r
  // (src/ipa/mod.rs:43-71) -- bp_ipa_round_lr's pass for every proof of the batch at once
  const bp_bases* bs = rb->bases;
  const u32 *gx = bs->x, *gy = bs->y, *hx = bs->x + n * 8, *hy = bs->y + n * 8;
  const uint8_t *ginf = bs->inf, *hinf = bs->inf + n;
  MsmTable table{bs->tab, bs->n};
  if (rb->rebased) {
    // the batch's own points: row b = G'_b (n) | H'_b (n) | u
    gx = rb->reb_x; gy = rb->reb_y; hx = rb->reb_x + n * 8; hy = rb->reb_y + n * 8;
    ginf = rb->reb_inf; hinf = rb->reb_inf + n;
    table = MsmTable{rb->reb_tab, (size_t)rb->batch * (2 * n + 1), 2 * n + 1, rb->reb_bits};
  }
  MsmPart pg{gx, gy, ginf, rb->eg, n};
  pg.sel_half = h; pg.group_hi = 0; pg.group_lo = 1;
  MsmPart ph{hx, hy, hinf, rb->eh, n};
  ph.sel_half = h; ph.group_lo = 0; ph.group_hi = 1;
  ph.tab_offset = n;
  pg.batch_stride = ph.batch_stride = n * 8;
  MsmExtra extra{{bs->x, bs->x}, {rb->dots + 16, rb->dots + 24}};
  extra.tab_index[0] = extra.tab_index[1] = 2 * n;
  extra.batch_stride[0] = extra.batch_stride[1] = 32;
  MsmPlan plan;
  u32* d_win = nullptr;
  CHK(msm_device(ctx, rb->curve, pg, ph, &plan, &d_win, 2, &extra, &table, false, 0, rb->batch));
  return rangeb_collect(rb, plan, d_win, out_lr_xyz);
}

int bp_range_batch_round_fold(bp_range_batch* rb, const uint64_t* x, const uint64_t* x_inv) {
  if (!rb || !x || !x_inv) return BP_E_ARG;
  bp_ctx* ctx = rb->ctx;
  Guard g(ctx);
  if (!rb->in_ipa || rb->cur_n < 2) return fail(ctx, BP_E_STATE, "bp_range_batch_round_fold: no round left");
  const size_t B = (size_t)rb->batch, h = rb->cur_n / 2;
  CHK(stage_h2d(ctx, rb->sc0, x, B * 32));
  CHK(stage_h2d(ctx, rb->sc1, x_inv, B * 32));
  hipError_t e;
  if (h >= 2) {
    e = launch_ipab_fold_prep(rb->curve, rb->a, rb->b, rb->a2, rb->b2, rb->cg, rb->ch, (u32)rb->row, (u32)h, rb->batch, rb->sc0, rb->sc1,
                              rb->factors, rb->eg, rb->eh, rb->dots, ctx->stream);
    if (e != hipSuccess) return fail(ctx, BP_E_HIP, "ipab_fold_prep", e);
    std::swap(rb->a, rb->a2);
    std::swap(rb->b, rb->b2);
    rb->prepped = true;
  } else {
    // two elements to one: the proofs' final a, b go to the pinned buffer
    e = launch_ipab_last_fold(rb->curve, rb->a, rb->b, (u32)rb->row, rb->batch, rb->sc0, rb->sc1, (u32*)rb->host, ctx->stream);
    if (e != hipSuccess) return fail(ctx, BP_E_HIP, "ipab_last_fold", e);
  }
  rb->cur_n = h;
  return BP_OK;
}

int bp_range_batch_finish(bp_range_batch* rb, uint64_t* out_ab) {
  if (!rb || !out_ab) return BP_E_ARG;
  bp_ctx* ctx = rb->ctx;
  Guard g(ctx);
  if (!rb->in_ipa || rb->cur_n != 1) return fail(ctx, BP_E_STATE, "bp_range_batch_finish: rounds left");
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(out_ab, rb->host, (size_t)rb->batch * 64);
  rb->in_ipa = false;
  return BP_OK;
}

}  // extern "C"

extern "C" {

// ---- measurement helpers -----------------------------------------------------------

int bp_bench_int_mad_peak(bp_ctx* ctx, int iters, double* mac_per_s) {
  if (!ctx || !mac_per_s || iters < 1) return fail(ctx, BP_E_ARG, "bp_bench_int_mad_peak: bad argument");
  Guard g(ctx);
  void* sink;
  CHK(reserve(ctx, S_SMALL, 256, &sink));
  hipDeviceProp_t prop;
  HIPCHK(ctx, hipGetDeviceProperties(&prop, ctx->device));
  int blocks = prop.multiProcessorCount * 8;
  hipError_t e = launch_int_mad_peak(iters, (u32*)sink, blocks, ctx->stream);  // warm-up
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "int_mad_peak", e);
  HIPCHK(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
  e = launch_int_mad_peak(iters, (u32*)sink, blocks, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "int_mad_peak", e);
  HIPCHK(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  float ms = 0;
  HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5]));
  double macs = (double)blocks * 256.0 * (double)iters * 256.0;
  *mac_per_s = macs / ((double)ms * 1e-3);
  return BP_OK;
}

int bp_microbench(bp_ctx* ctx, int id, int blocks, int iters, float* ms) {
  if (!ctx || !ms || blocks < 1 || iters < 1) return fail(ctx, BP_E_ARG, "bp_microbench: bad argument");
  Guard g(ctx);
  void* io;
  if (id == 5) {
    if (iters < 10 || iters > 26) return fail(ctx, BP_E_ARG, "bp_microbench: id 5 takes 10 <= iters <= 26");
    const size_t records = (size_t)1 << iters;
    void* table;
    CHK(reserve(ctx, S_PACKED, records * 64, &table));
    CHK(reserve(ctx, S_SMALL, 256, &io));
    HIPCHK(ctx, hipMemsetAsync(table, 0x5a, records * 64, ctx->stream));
    hipError_t ge = launch_gather_bench((const u32*)table, records, (u32*)io, ctx->stream);
    if (ge != hipSuccess) return fail(ctx, BP_E_HIP, "bp_microbench: gather", ge);
    HIPCHK(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
    ge = launch_gather_bench((const u32*)table, records, (u32*)io, ctx->stream);
    if (ge != hipSuccess) return fail(ctx, BP_E_HIP, "bp_microbench: gather", ge);
    HIPCHK(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipEventElapsedTime(ms, ctx->ev[4], ctx->ev[5]));
    return BP_OK;
  }
  CHK(reserve(ctx, S_TABLE, 1024 * 32, &io));
  std::vector<uint32_t> seed(1024 * 8);
  uint64_t s = 0x9E3779B97F4A7C15ull;
  for (auto& v : seed) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    v = (uint32_t)(s >> 33);
  }
  for (size_t i = 7; i < seed.size(); i += 8) seed[i] &= 0x3fffffffu;  // below every modulus
  if (id == 8 || id == 12) seed[0] = 0;  // the self-checks count disagreeing lanes there
  HIPCHK(ctx, hipMemcpyAsync(io, seed.data(), seed.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  if (id == 12) {
    hipError_t ce = launch_batch_affine_check(blocks, (u32*)io, ctx->stream);
    if (ce != hipSuccess) return fail(ctx, BP_E_HIP, "bp_microbench", ce);
    uint32_t bad = 0;
    HIPCHK(ctx, hipMemcpyAsync(&bad, io, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *ms = (float)bad;
    return BP_OK;
  }
  if (id == 11) {
    const size_t park_bytes = (size_t)blocks * 128 * (size_t)iters * 32;
    if (park_bytes > ((size_t)8 << 30)) return fail(ctx, BP_E_ARG, "bp_microbench: id 11 takes blocks * 128 * iters * 32 <= 8 GiB");
    void *park, *sink;
    CHK(reserve(ctx, S_PACKED, park_bytes, &park));
    CHK(reserve(ctx, S_SMALL, 256, &sink));
    hipError_t be = launch_batch_affine_bench(blocks, iters, (const u32*)io, (u32*)park, (u32*)sink, ctx->stream);
    if (be != hipSuccess) return fail(ctx, BP_E_HIP, "bp_microbench", be);
    HIPCHK(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
    be = launch_batch_affine_bench(blocks, iters, (const u32*)io, (u32*)park, (u32*)sink, ctx->stream);
    if (be != hipSuccess) return fail(ctx, BP_E_HIP, "bp_microbench", be);
    HIPCHK(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipEventElapsedTime(ms, ctx->ev[4], ctx->ev[5]));
    return BP_OK;
  }
  if (id == 8) {
    hipError_t ce = launch_microbench(id, blocks, iters, (u32*)io, ctx->stream);
    if (ce != hipSuccess) return fail(ctx, BP_E_HIP, "bp_microbench", ce);
    uint32_t bad = 0;
    HIPCHK(ctx, hipMemcpyAsync(&bad, io, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *ms = (float)bad;
    return BP_OK;
  }
  hipError_t e = launch_microbench(id, blocks, iters, (u32*)io, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_ARG, "bp_microbench: bad id", e);
  HIPCHK(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
  e = launch_microbench(id, blocks, iters, (u32*)io, ctx->stream);
  if (e != hipSuccess) return fail(ctx, BP_E_HIP, "bp_microbench", e);
  HIPCHK(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  HIPCHK(ctx, hipEventElapsedTime(ms, ctx->ev[4], ctx->ev[5]));
  return BP_OK;
}

int bp_set_timing(bp_ctx* ctx, int on) {
  if (!ctx) return BP_E_ARG;
  Guard g(ctx);
  ctx->timing = on != 0;
  return BP_OK;
}

int bp_last_kernel_ms(bp_ctx* ctx, int group, float* ms) {
  if (!ctx || !ms || group < 0 || group > 2) return fail(ctx, BP_E_ARG, "bp_last_kernel_ms: bad argument");
  Guard g(ctx);
  *ms = ctx->last_ms[group];
  return BP_OK;
}

}  // extern "C"
