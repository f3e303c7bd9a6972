// Multi-scalar multiplication for gfx950: Pippenger's bucket method with signed
// c-bit digits.
//   1. k_digits      scalars: Montgomery -> integer (folded to |k| <= r/2) -> W signed
//                    digits; emits one (bucket, sign | window | point index) pair per point
//                    and window, and packs x|y of every point into one 64-byte record
//   2. radix sort    pairs by bucket (rocPRIM; stable, input window-major)
//   3. k_bounds      first/last sorted position of every (window, bucket)
//   4. k_chunk_*     buckets cut into chunks of bounded length, chunk ids sorted by length,
//                    one lane per chunk sums it in XYZZ coordinates with mixed additions;
//                    k_bucket_finish / k_bucket_heavy join the chunks of a bucket
//   5. k_reduce_segments, k_group_sum   sum_b b*B_b per window by segmented running sums
// Up to kSmallMaxPoints points the launch count dominates, and steps 1-3 are one kernel
// (k_small_front), step 4 a lane-group tree per bucket (k_bucket_tree) and step 5 one block
// per window (k_window_reduce_small).  A plan can carry two output groups (two independent
// sums from one pass) and one extra device-side (point, scalar) per group.
// The W window sums go back to the host, which finishes sum_w 2^(c*w) W_w (a
// strictly sequential chain of ~256 doublings, slower on a GPU lane than on a CPU
// core).  Stands in for ark-ec 0.5.0 VariableBaseMSM::msm_unchecked as called at
// src/msm.rs:84-87 and the direct call sites listed in include/bp_gpu.h.
#include <cstdlib>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "bp_curve.h"
#include "bp_internal.h"
#include "bp_quad.h"

namespace bp {

// C = the curve as the pipeline computes on it (secp256k1: plain residues, see CurveT)
#define DISPATCH_CURVE(curve, STMT)              \
  switch (curve) {                               \
    case BP_SECP256K1: {                         \
      using C = Secp256k1P;                      \
      constexpr int kBits = 256;                 \
      (void)kBits;                               \
      STMT;                                      \
    } break;                                     \
    case BP_VESTA: {                             \
      using C = Vesta;                           \
      constexpr int kBits = 255;                 \
      (void)kBits;                               \
      STMT;                                      \
    } break;                                     \
    default:                                     \
      return hipErrorInvalidValue;               \
  }

constexpr u32 kChunkKeyInvalid = 63;  // sort key of the unused tail of the chunk table
// limits of the one-launch front end for small problems (k_small_front)
constexpr int kSmallThreads = 1024;
// chunk table (k_ct_*): buckets per lane and per tile of one 1024-lane block, number of sort keys
constexpr u32 kCtPer = 2;
constexpr u32 kCtTile = kSmallThreads * kCtPer;
constexpr u32 kCtKeys = kChunkKeyInvalid + 1;
constexpr int kSmallIters = 17;
constexpr size_t kSmallMaxPoints = (size_t)kSmallThreads * kSmallIters;
constexpr u32 kSmallMaxBuckets = 4096;
// the sliced counting sort for mid-sized problems (k_mid_*)
constexpr int kMidIters = 8;
constexpr u32 kMidSlice = kSmallThreads * kMidIters;
constexpr int kMidWindowsPerBlock = 4;  // k_mid_count_multi
constexpr u32 kMidMaxBuckets = 256;
constexpr int kTwoLevelLoBits = 8;       // see the two-level counting sort
constexpr u32 kTwoLevelMaxBins = 4096;
constexpr int kTwoLevelIdxBits = 22;
constexpr u32 kRowcolCols = 256;         // columns of the bucket matrix of the row/column reduction
constexpr int kRowcolStrip = 8;          // buckets a lane adds before the lanes of a row or column join

static inline int ilog2_floor(size_t n) {
  int l = 0;
  while (n >>= 1) l++;
  return l;
}

// Window width by problem size, from measurements with uniform scalars on MI355X.  Widths
// that divide 256 (4, 8, 16) leave no short top window: a top window of t bits puts all
// points into 2^(t-1) buckets, which then need a deep tree of partial sums.
int msm_choose_window(size_t n) {
  int l = ilog2_floor(n ? n : 1);
  // 16-bit windows from 2^17 points: measured 0.41 ms against 0.49 with 8-bit windows at 2^17,
  // 0.38 against 0.31 at 2^16 (device phase, secp256k1)
  if (l >= 17) return 16;
  if (l >= 9) return 8;
  return 4;
}

// width msm_make_plan picks for one sum over n points with no override and no window tables
int msm_default_window(size_t n) {
  // below kSmallMaxPoints the one-launch front end and the tree kernels apply, and 8-bit
  // windows (128 buckets, no short top window) are fastest from 2^9 points up
  if (n > 0 && n <= kSmallMaxPoints && !getenv("BP_MSM_NO_SMALL")) return n >= 512 ? 8 : 4;
  return msm_choose_window(n);
}

static int key_bits_for(u32 max_key) {
  int b = 1;
  while ((1ull << b) <= max_key) b++;
  return b;
}

static hipError_t make_plan_impl(int curve, size_t n_total, int c_override, int groups, MsmPlan* plan, size_t table_stride,
                                 bool allow_stages, int batch, int table_bits, bool allow_spread) {
  MsmPlan p{};
  if (groups < 1 || groups > 2 || batch < 1 || batch > kMsmMaxBatch) return hipErrorInvalidValue;
  p.groups = groups;
  p.batch = batch;
  p.tab = table_stride != 0 && n_total > 0;
  p.c = (c_override >= 2 && c_override <= 20) ? c_override : msm_choose_window(n_total / groups);
  const bool small_ok = n_total > 0 && n_total <= kSmallMaxPoints && !getenv("BP_MSM_NO_SMALL");
  if (small_ok && !(c_override >= 2 && c_override <= 20)) p.c = n_total / groups >= 512 ? 8 : 4;
  if (p.tab && table_bits != kTableWindowBits && !(table_bits == 4 && batch > 1)) return hipErrorInvalidValue;
  if (p.tab) p.c = table_bits;
  // k_digits maps a scalar k > r/2 to r - k (and negates the point), so digits
  // come from a value below 2^255 (secp256k1) or 2^254 (Vesta); one more bit
  // absorbs the carry of the signed recoding.
  const int value_bits = curve == BP_VESTA ? 254 : 255;
  p.group_windows = (value_bits + 1 + p.c - 1) / p.c;
  // A width that does not divide 256 leaves a short top window: all points in a few buckets, which then need a
  // deep tree of partial sums.  Where the front end can (the two-level counting sort, checked below) the scalars
  // are spread instead: k + t r with t < 2^s fills W c - 1 bits evenly (the modulus is below 2^mod_bits), W the
  // fewest windows that leave s >= 1.
  const int mod_bits = curve == BP_VESTA ? 255 : 256;
  int spread_windows = 0;
  if (!p.tab && batch == 1 && !small_ok && 256 % p.c != 0 && p.c >= kTwoLevelLoBits + 2 && p.c <= 16 &&
      allow_spread) {
    spread_windows = (mod_bits + 2 + p.c - 1) / p.c;
    int s_bits = spread_windows * p.c - 1 - mod_bits;
    if (s_bits > 16) s_bits = 16;
    p.spread_t = 1u << s_bits;
    p.group_windows = spread_windows;
  }
  p.digit_windows = p.group_windows;
  p.tab_virtual = p.tab && small_ok && (p.digit_windows * groups <= 64 || batch > 1) && !getenv("BP_MSM_TAB_MERGED");
  p.finish_c = p.tab_virtual ? 0 : p.c;
  if (p.tab) {
    if (p.digit_windows > 256 / table_bits) return hipErrorInvalidValue;
    // every digit window lands in the one bucket set of its group: one output "window" per group
    // (unless the plan keeps them apart, see tab_virtual)
    if (!p.tab_virtual) p.group_windows = 1;
  }
  p.n_windows = p.group_windows * groups * batch;
  p.n_buckets = 1u << (p.c - 1);
  p.n_total = n_total;
  p.entries = (size_t)p.digit_windows * n_total;
  if (p.entries >= (1ull << 31)) return hipErrorInvalidValue;
  // Buckets per reduction lane: the reduction is a dependent chain of 2*seg_len additions
  // plus a multiplication by the segment offset, so short segments win as long as the
  // lanes still fit the machine about once (256 CUs * 4 SIMDs * 64 lanes).
  // Batches of windows for the three-stream pipeline (single sums on the chunked path): off
  // unless BP_MSM_STAGES asks for it.  Measured on MI355X at 2^20 points it loses (2.38 ms in
  // one batch, 2.90 in two, 3.40 in four): the streams do overlap, but every batch pays the
  // launch-bound part of the front end again (the chunk-length sort of a quarter of the chunks
  // is a merge sort of some forty launches) and its reduction is the same dependent chain
  // whatever the number of windows, slower still while the accumulation shares the SIMDs.
  p.stages = 1;
  {
    const bool chunked = !(small_ok && p.n_buckets <= kSmallMaxBuckets) && !p.tab &&
                         !(p.n_buckets >= 64 && p.n_buckets <= kMidMaxBuckets && n_total <= ((size_t)1 << 20));
    int want = 1;
    if (const char* ev = getenv("BP_MSM_STAGES")) want = atoi(ev);
    if (allow_stages && chunked && groups == 1 && want > 1) {
      if (want > kMsmMaxStages) want = kMsmMaxStages;
      if (want > p.n_windows) want = p.n_windows;
      p.stages = want;
    }
  }
  const int stage_windows = (p.n_windows + p.stages - 1) / p.stages;  // windows reduced by one launch
  p.seg_len = 16;
  while (p.seg_len > 4 && (size_t)stage_windows * p.n_buckets / (p.seg_len / 2) <= 65536) p.seg_len /= 2;
  if (const char* ev = getenv("BP_MSM_SEG_LEN")) p.seg_len = atoi(ev);
  if ((u32)p.seg_len > p.n_buckets) p.seg_len = (int)p.n_buckets;
  p.n_segs = p.n_buckets / p.seg_len;
  size_t nbt = (size_t)p.n_windows * p.n_buckets;
  p.keys_bytes = ((p.entries * sizeof(u32)) + 255) & ~(size_t)255;
  p.bounds_bytes = ((nbt + 2 + kMsmMaxStages) * sizeof(u32) + 255) & ~(size_t)255;
  p.bucket_bytes = nbt * 4 * 32;
  p.seg_bytes = (size_t)p.n_windows * p.n_segs * 128;
  // 512..65536 buckets per window: row and column sums instead of segments (see k_rowcol_sums);
  // the same buffer then holds 256 column sums, n_buckets/256 row sums and two partial results
  // per window
  p.rowcol = p.stages == 1 && p.n_buckets >= 2 * kRowcolCols && p.n_buckets <= kRowcolCols * kRowcolCols &&
             !getenv("BP_MSM_SEGMENTS");
  if (p.rowcol) {
    size_t need = (size_t)p.n_windows * (kRowcolCols + p.n_buckets / kRowcolCols + 2) * 128;
    if (need > p.seg_bytes) p.seg_bytes = need;
  }
  p.seg_group_bytes = (size_t)p.n_windows * (p.n_segs / 1024 + 1) * 128;
  p.win_bytes = (size_t)p.n_windows * 128;
  p.packed_bytes = (n_total ? n_total : 1) * 64;
  size_t tmp = 0;
  // value word: sign (bit 31) | window | point index
  int wbits = 0;
  while ((1 << wbits) < p.n_windows) wbits++;
  p.wshift = 31 - wbits;
  if (batch > 1) p.wshift = 31;  // batched plans keep no window number in the value word (nothing reads it on their path)
  if (n_total >= ((size_t)1 << p.wshift)) return hipErrorInvalidValue;
  if (p.tab && (size_t)(256 / table_bits) * table_stride >= ((size_t)1 << p.wshift)) return hipErrorInvalidValue;
  p.sort_tmp_bytes = 256;
  // chunk length: up to 24 entries, shorter when the problem is small so that there are still
  // tens of thousands of lanes (2^20 points, 16-bit windows, per step: 1.485 ms at 16, 1.450 at 24,
  // 1.477 at 32, 1.466 at 64: shorter chunks balance the accumulation, longer ones leave fewer
  // chunk sums to join)
  size_t t = p.entries / 65536;
  p.chunk = (u32)(t < 4 ? 4 : (t > 24 ? 24 : t));
  // buckets of about 16 entries (2^19 points at 16-bit windows): one chunk per bucket and nothing left to join
  // (0.775 -> 0.738 ms device phase; at 32 entries per bucket the longer chunks gain nothing)
  if (p.chunk == 24 && p.entries / (nbt ? nbt : 1) >= 12 && p.entries / (nbt ? nbt : 1) <= 20) p.chunk = 48;
  // small problems are a dependent chain of `chunk` mixed additions (10 M each) and then one
  // full addition (14 M) per chunk of the bucket: for E entries per bucket the chain is
  // shortest near chunk = sqrt(1.4 E)
  {
    size_t per_bucket = n_total / (size_t)groups / p.n_buckets + 1;
    u32 bal = 1;
    while ((size_t)bal * bal * 10 < per_bucket * 14) bal++;
    if (bal > p.chunk && bal <= 32) p.chunk = bal;
    if (getenv("BP_MSM_CHUNK")) p.chunk = (u32)atoi(getenv("BP_MSM_CHUNK"));  // tuning runs
    if (p.chunk < 1) p.chunk = 1;
  }
  // very large inputs: about 2^21 chunks are plenty of lanes, and longer chunks keep the
  // number of partial sums per bucket small
  if ((p.entries >> 21) > p.chunk) p.chunk = (u32)((p.entries >> 21) > 4096 ? 4096 : (p.entries >> 21));
  p.nc_cap = p.entries / p.chunk + nbt + 1 + kMsmMaxStages;  // room for one rounding slot per batch
  p.small = small_ok && p.n_buckets <= kSmallMaxBuckets && (!p.tab || p.tab_virtual);
  if (batch > 1 && !(p.small && p.tab_virtual)) return hipErrorInvalidValue;
  // narrow windows on more points than one block sorts: the sliced counting sort
  p.mid = (p.tab && !p.tab_virtual) || (!p.small && n_total > 0 && p.n_buckets >= 64 && p.n_buckets <= kMidMaxBuckets &&
                    n_total <= ((size_t)1 << 20) && !getenv("BP_MSM_NO_MID"));
  if (p.small || p.mid) {
    // every output window owns a fixed range of slots of the sorted-entry array: n_total, or
    // all of a group's digit windows together for table plans
    size_t need = (size_t)p.n_windows * (p.tab && !p.tab_virtual ? p.entries : n_total) * sizeof(u32);
    if (need > p.keys_bytes) p.keys_bytes = (need + 255) & ~(size_t)255;
  }
  // Short slices where the windows of a slice share a block (k_mid_count_multi): the blocks of a launch stay
  // as many as the CUs while a scalar is converted once per four windows
  p.mid_iters = (p.mid && !p.tab && n_total <= ((size_t)1 << 18) && p.digit_windows % kMidWindowsPerBlock == 0 &&
                 !getenv("BP_MSM_MID_SINGLE")) ? 2 : kMidIters;
  if (p.mid) {
    // the sliced counting sort keeps one histogram per (group, digit window, slice) in the second
    // key buffer: a size that does not shrink with n (16 KiB per group at one slice)
    const size_t n_slices = (n_total + (size_t)kSmallThreads * p.mid_iters - 1) / ((size_t)kSmallThreads * p.mid_iters);
    size_t need = (size_t)groups * p.digit_windows * n_slices * p.n_buckets * sizeof(u32);
    if (need > p.keys_bytes) p.keys_bytes = (need + 255) & ~(size_t)255;
  }
  // the chunked path sorts with the two-level counting sort when the bucket number has a high
  // part and the counters of all (window, partition) pairs fit one block
  p.two_level = !p.small && !p.mid && !p.tab && p.stages == 1 && n_total > 0 &&
                n_total <= ((size_t)1 << kTwoLevelIdxBits) && p.n_buckets >= (2u << kTwoLevelLoBits) &&
                (size_t)groups * (p.n_buckets >> kTwoLevelLoBits) <= (1u << kTwoLevelLoBits) &&
                (size_t)p.n_windows * (p.n_buckets >> kTwoLevelLoBits) <= kTwoLevelMaxBins &&
                nbt + 2 >= 2 * (size_t)kTwoLevelMaxBins && !getenv("BP_MSM_RADIX");
  // no spreading on the other front ends: the caller plans again with folded scalars
  if (p.spread_t && !p.two_level) return hipErrorNotSupported;
  if (!p.small && !p.mid && !p.two_level) {
    // only the plans that still sort with the library's radix sort (more than 2^22 points, window
    // widths without a high bucket part, BP_MSM_RADIX) ask it for its temporary storage
    hipError_t e = rocprim::radix_sort_pairs<rocprim::default_config, const u32*, u32*, const u32*, u32*>(
        nullptr, tmp, nullptr, nullptr, nullptr, nullptr, p.entries ? p.entries : 1, 0, key_bits_for(p.n_buckets), 0,
        false);
    if (e != hipSuccess) return e;
    p.sort_tmp_bytes = (tmp + 255) & ~(size_t)255;
  }
  p.nch_bytes = ((nbt + 2 + kMsmMaxStages) * sizeof(u32) + 255) & ~(size_t)255;
  p.chunk_sum_bytes = p.nc_cap * 4 * 32;
  {
    // chunk table scratch: tile sums, tile offsets, per-tile key histograms, key cursors
    const size_t n_tiles = (nbt + 1 + kCtTile - 1) / kCtTile;
    p.scan_tmp_bytes = (((2 + kCtKeys) * n_tiles + kCtKeys) * sizeof(u32) + 255) & ~(size_t)255;
  }
  p.cdesc_bytes = ((p.nc_cap * sizeof(u32)) + 255) & ~(size_t)255;
  p.csort_tmp_bytes = 256;
  *plan = p;
  return hipSuccess;
}

hipError_t msm_make_plan(int curve, size_t n_total, int c_override, int groups, MsmPlan* plan, size_t table_stride,
                         bool allow_stages, int batch, int table_bits) {
  static const bool no_spread = getenv("BP_MSM_NO_SPREAD") != nullptr;  // comparison runs
  hipError_t e = make_plan_impl(curve, n_total, c_override, groups, plan, table_stride, allow_stages, batch, table_bits, !no_spread);
  if (e == hipErrorNotSupported)
    e = make_plan_impl(curve, n_total, c_override, groups, plan, table_stride, allow_stages, batch, table_bits, false);
  return e;
}

// ---- 1. digits --------------------------------------------------------------
// Group selection of the two parts (see MsmPart): half == 0 puts the whole part in group lo.
struct DigitGroups {
  u32 half0, half1;
  u32 lo0, hi0, lo1, hi1;
  // n_extra = 2: one more point per group after the two parts, extra e in group e; its
  // coordinates (x|y, 16 u32) and scalar are read from device memory
  u32 n_extra;
  const u32* extra_xy[2];
  const u32* extra_s[2];
  u32 extra_tab[2];  // table plans: index of the extra point in the base set
  // batched plans: problem b = blockIdx.y reads its scalars b * stride u32 further on
  u32 s_stride0 = 0, s_stride1 = 0, extra_stride[2] = {0, 0};
  u32 tab_sess_stride = 0;  // table plans: problem b's base indices start b * tab_sess_stride further on
};

template <class C>
__global__ __launch_bounds__(256) void k_digits(const u32* __restrict__ s0, const uint8_t* __restrict__ inf0,
                                                size_t n0, const u32* __restrict__ s1,
                                                const uint8_t* __restrict__ inf1, size_t n1, int c,
                                                int n_windows, u32 n_buckets, int wshift,
                                                u32* __restrict__ keys, u32* __restrict__ vals,
                                                const u32* __restrict__ x0,
                                                const u32* __restrict__ y0, const u32* __restrict__ x1,
                                                const u32* __restrict__ y1, u32* __restrict__ packed,
                                                DigitGroups sel) {
  using Fr = typename C::Fr;
  const size_t n01 = n0 + n1;
  const size_t n = n01 + sel.n_extra;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool is_extra = i >= n01;
  // Output group of this point and its position inside a window's slice of the entry
  // table.  Positions put group 0 before group 1, so that after the stable sort on the
  // bucket number the entries of one (bucket, window, group) are still contiguous.
  u32 grp;
  size_t pos;
  {
    const bool second = i >= n0;
    const u32 half_sel = second ? sel.half1 : sel.half0;
    const size_t gsize = n01 / 2 + (sel.n_extra ? 1 : 0);  // positions per group in split mode
    if (is_extra) {
      grp = (u32)(i - n01);
      pos = (size_t)grp * gsize + n01 / 2;
    } else if (half_sel == 0) {
      grp = second ? sel.lo1 : sel.lo0;
      pos = i;
    } else {
      size_t li = second ? i - n0 : i;
      size_t q = li / (2 * (size_t)half_sel), r = li % (2 * (size_t)half_sel);
      bool up = r >= half_sel;
      grp = up ? (second ? sel.hi1 : sel.hi0) : (second ? sel.lo1 : sel.lo0);
      pos = (size_t)grp * gsize + (second ? n0 / 2 : 0) + q * half_sel + (up ? r - half_sel : r);
    }
  }
  const u32 wbase = grp * (u32)n_windows;
  // x and y of a point side by side (64 contiguous bytes): the accumulation kernel
  // gathers points in bucket order, and one 64-byte piece costs half the memory
  // transactions of two 32-byte pieces from separate planes
  {
    const u32* px = i < n0 ? x0 + i * 8 : x1 + (i - n0) * 8;
    const u32* py = i < n0 ? y0 + i * 8 : y1 + (i - n0) * 8;
    if (is_extra) {
      px = sel.extra_xy[i - n01];
      py = px + 8;
    }
    fe_store(packed + i * 16, fe_import<C>(fe_load(px)));
    fe_store(packed + i * 16 + 8, fe_import<C>(fe_load(py)));
  }
  const u32* sp = i < n0 ? s0 + i * 8 : s1 + (i - n0) * 8;
  if (is_extra) sp = sel.extra_s[i - n01];
  bool is_inf = is_extra ? false : (i < n0 ? (inf0 && inf0[i]) : (inf1 && inf1[i - n0]));
  Fe k = fe_from_mont<Fr>(fe_load(sp));
  // |k| <= r/2: use r - k with the point negated when that is the smaller value
  u32 flip = 0;
  {
    Fe nk = fe_neg<Fr>(k);
    bool gt = false, decided = false;
#pragma unroll
    for (int t = 7; t >= 0; t--) {
      if (!decided && k.v[t] != nk.v[t]) {
        gt = k.v[t] > nk.v[t];
        decided = true;
      }
    }
    if (gt) {
      k = nk;
      flip = 0x80000000u;
    }
  }
  const u32 invalid = n_buckets;  // sorts behind every bucket
  const u32 half = 1u << (c - 1);
  const u32 mask = (1u << c) - 1;
  u32 carry = 0;
  for (int w = 0; w < n_windows; w++) {
    int bit = w * c;
    int limb = bit >> 5, sh = bit & 31;
    u32 raw = 0;
    if (limb < 8) {
      // dynamic limb index: select without scratch
      u32 lo = 0, hi = 0;
#pragma unroll
      for (int t = 0; t < 8; t++) {
        lo = (t == limb) ? k.v[t] : lo;
        hi = (t == limb + 1) ? k.v[t] : hi;
      }
      unsigned long long two = ((unsigned long long)hi << 32) | lo;
      raw = (u32)(two >> sh) & mask;
    }
    u32 v = raw + carry;
    u32 key, val;
    if (v > half) {
      carry = 1;
      u32 mag = (1u << c) - v;  // |v - 2^c|
      key = mag ? mag - 1 : invalid;  // v == 2^c: digit 0, carry 1
      val = ((u32)i | ((wbase + (u32)w) << wshift) | 0x80000000u) ^ flip;
    } else {
      carry = 0;
      key = v ? v - 1 : invalid;
      val = ((u32)i | ((wbase + (u32)w) << wshift)) ^ flip;
    }
    if (is_inf) key = invalid;
    keys[(size_t)w * n + pos] = key;
    vals[(size_t)w * n + pos] = val;
  }
}

// ---- 1-3 for small problems ---------------------------------------------------------
// Up to kSmallMaxPoints points: one 1024-lane block per (group, window) does what k_digits,
// the radix sort and k_bounds do for large inputs -- digits of its window, a counting sort
// in LDS (bucket histogram with LDS atomics whose return value is the entry's rank inside
// its bucket, block scan, scatter) and the bucket bounds -- in one launch instead of about
// thirty; the accumulation (k_bucket_tree) needs no chunk table.  A small MSM
// is a chain of dependent launches, so the launch count is what it costs.  The order of
// entries inside a bucket is arbitrary here; the bucket's sum does not depend on it.

// exclusive prefix sum of one value per lane over a 1024-lane block; *total = sum of all
__device__ __forceinline__ u32 block_exclusive_scan(u32 v, u32* wave_sums, u32* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32 inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    u32 o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  if (lane == 63) wave_sums[wave] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 run = 0;
    for (int w = 0; w < kSmallThreads / 64; w++) {
      u32 t = wave_sums[w];
      wave_sums[w] = run;
      run += t;
    }
    wave_sums[kSmallThreads / 64] = run;
  }
  __syncthreads();
  u32 res = wave_sums[wave] + inc - v;
  *total = wave_sums[kSmallThreads / 64];
  __syncthreads();
  return res;
}

template <class C>
__global__ __launch_bounds__(1024) void k_small_front(
    const u32* __restrict__ s0, const uint8_t* __restrict__ inf0, size_t n0, const u32* __restrict__ s1,
    const uint8_t* __restrict__ inf1, size_t n1, const u32* __restrict__ x0, const u32* __restrict__ y0,
    const u32* __restrict__ x1, const u32* __restrict__ y1, u32* __restrict__ packed, DigitGroups sel, int c,
    int group_windows, u32 n_buckets, int wshift, u32* __restrict__ sorted_vals, u32* __restrict__ bucket_start,
    u32* __restrict__ bucket_end, u32* __restrict__ heavy_count, u32 tab_stride, u32 tab_off0, u32 tab_off1) {
  // tab_stride != 0: the points are records of a base set's window tables (record of window w and
  // base j at w * tab_stride + j); nothing is packed and an entry names its table record
  using Fr = typename C::Fr;
  __shared__ u32 hist[kSmallMaxBuckets];
  __shared__ u32 off[kSmallMaxBuckets];
  __shared__ u32 wave_sums[kSmallThreads / 64 + 1];
  __shared__ u32 thr[8], msk[8];  // window w: carry-in threshold T_w and the mask of the low c*w bits
  // problem blockIdx.y of a batch: its own scalars, its own windows (gwid), the same points
  const u32 sess = blockIdx.y;
  s0 += (size_t)sess * sel.s_stride0;
  s1 += (size_t)sess * sel.s_stride1;
  sel.extra_s[0] += (size_t)sess * sel.extra_stride[0];
  sel.extra_s[1] += (size_t)sess * sel.extra_stride[1];
  if (inf0) inf0 += (size_t)sess * sel.tab_sess_stride;
  if (inf1) inf1 += (size_t)sess * sel.tab_sess_stride;
  const u32 wid = blockIdx.x, gwid = sess * gridDim.x + wid;
  const u32 g = wid / (u32)group_windows, w = wid % (u32)group_windows;
  const size_t n01 = n0 + n1, n = n01 + sel.n_extra;
  const u32 half = 1u << (c - 1), mask = (1u << c) - 1;
  for (u32 b = threadIdx.x; b < n_buckets; b += kSmallThreads) hist[b] = 0;
  if (threadIdx.x == 0) {
    // A value's signed digits d_j in (-half, half] below window w represent its low c*w bits
    // without a carry exactly when those bits are <= T_w = half * sum_{j<w} 2^(c*j).
    for (int t = 0; t < 8; t++) thr[t] = msk[t] = 0;
    for (u32 j = 0; j < w; j++) {
      u32 bit = j * (u32)c + (u32)c - 1;  // position of `half` in digit j
      if (bit < 256) thr[bit >> 5] |= 1u << (bit & 31);
    }
    u32 low = w * (u32)c;
    for (u32 bit = 0; bit < low && bit < 256; bit++) msk[bit >> 5] |= 1u << (bit & 31);
    if (gwid == 0) *heavy_count = 0;
  }
  __syncthreads();
  u32 my_key[kSmallIters], my_val[kSmallIters], my_rank[kSmallIters];
  const u32 invalid = n_buckets;
  const int bit0 = (int)(w * (u32)c), limb = bit0 >> 5, sh = bit0 & 31;
  // Packing: every point's x|y record is written (in the pipeline's form) by exactly one lane of
  // the grid, ahead of the digit loop so that the conversion does not share registers with it.
#pragma unroll 1
  for (size_t i = (size_t)threadIdx.x * gridDim.x + wid; i < n && tab_stride == 0; i += (size_t)kSmallThreads * gridDim.x) {
    const bool second = i >= n0;
    const u32* px = second ? x1 + (i - n0) * 8 : x0 + i * 8;
    const u32* py = second ? y1 + (i - n0) * 8 : y0 + i * 8;
    if (i >= n01) {
      px = sel.extra_xy[i - n01];
      py = px + 8;
    }
    fe_store(packed + i * 16, fe_import<C>(fe_load(px)));
    fe_store(packed + i * 16 + 8, fe_import<C>(fe_load(py)));
  }
  // The block walks over the entries of its own group only (the lower or upper halves of a part
  // whose halves go to different groups, see DigitGroups), so that every wave it runs is a full
  // one whatever the length of the halves: entry t of the group is point `i` of the input.
  const bool split0 = sel.half0 != 0 && sel.lo0 != sel.hi0, split1 = sel.half1 != 0 && sel.lo1 != sel.hi1;
  size_t cnt0, cnt1;
  {
    const size_t q0 = split0 ? n0 / (2 * (size_t)sel.half0) : 0, r0 = split0 ? n0 % (2 * (size_t)sel.half0) : 0;
    const size_t q1 = split1 ? n1 / (2 * (size_t)sel.half1) : 0, r1 = split1 ? n1 % (2 * (size_t)sel.half1) : 0;
    cnt0 = !split0 ? (sel.lo0 == g ? n0 : 0)
           : sel.hi0 == g ? q0 * sel.half0 + (r0 > sel.half0 ? r0 - sel.half0 : 0)
           : sel.lo0 == g ? q0 * sel.half0 + (r0 < sel.half0 ? r0 : sel.half0) : 0;
    cnt1 = !split1 ? (sel.lo1 == g ? n1 : 0)
           : sel.hi1 == g ? q1 * sel.half1 + (r1 > sel.half1 ? r1 - sel.half1 : 0)
           : sel.lo1 == g ? q1 * sel.half1 + (r1 < sel.half1 ? r1 : sel.half1) : 0;
  }
  const size_t cnt = cnt0 + cnt1 + (g < (u32)sel.n_extra ? 1 : 0);
#pragma unroll
  for (int it = 0; it < kSmallIters; it++) {
    my_key[it] = invalid;
    my_val[it] = 0;
    my_rank[it] = 0;
    const size_t t = (size_t)it * kSmallThreads + threadIdx.x;
    if (t >= cnt) continue;
    size_t i;
    if (t < cnt0) {
      i = split0 ? (t / sel.half0) * 2 * (size_t)sel.half0 + t % sel.half0 + (sel.hi0 == g ? sel.half0 : 0) : t;
    } else if (t < cnt0 + cnt1) {
      const size_t u = t - cnt0;
      i = n0 + (split1 ? (u / sel.half1) * 2 * (size_t)sel.half1 + u % sel.half1 + (sel.hi1 == g ? sel.half1 : 0) : u);
    } else {
      i = n01 + g;
    }
    const bool is_extra = i >= n01, second = i >= n0;
    const u32* sp = second ? s1 + (i - n0) * 8 : s0 + i * 8;
    if (is_extra) sp = sel.extra_s[i - n01];
    const bool is_inf = is_extra ? false : (second ? (inf1 && inf1[i - n0]) : (inf0 && inf0[i]));
    Fe k = fe_from_mont<Fr>(fe_load(sp));
    u32 flip = 0;
    {
      Fe nk = fe_neg<Fr>(k);
      bool gt = false, decided = false;
#pragma unroll
      for (int t = 7; t >= 0; t--) {
        if (!decided && k.v[t] != nk.v[t]) {
          gt = k.v[t] > nk.v[t];
          decided = true;
        }
      }
      if (gt) {
        k = nk;
        flip = 0x80000000u;
      }
    }
    // carry into window w: (k mod 2^(c*w)) > T_w
    u32 carry = 0;
    {
      bool gt = false, decided = false;
#pragma unroll
      for (int t = 7; t >= 0; t--) {
        u32 kv = k.v[t] & msk[t], tv = thr[t];
        if (!decided && kv != tv) {
          gt = kv > tv;
          decided = true;
        }
      }
      carry = gt ? 1u : 0u;
    }
    u32 raw = 0;
    if (limb < 8) {
      u32 lo = 0, hi = 0;
#pragma unroll
      for (int t = 0; t < 8; t++) {
        lo = (t == limb) ? k.v[t] : lo;
        hi = (t == limb + 1) ? k.v[t] : hi;
      }
      unsigned long long two = ((unsigned long long)hi << 32) | lo;
      raw = (u32)(two >> sh) & mask;
    }
    const u32 wfield = wshift < 31 ? wid << wshift : 0u;  // batched plans keep no window number (msm_make_plan)
    u32 v = raw + carry, key, val = (u32)i | wfield;
    if (tab_stride) {
      const u32 j = sess * sel.tab_sess_stride +
                    (is_extra ? sel.extra_tab[i - n01] : second ? tab_off1 + (u32)(i - n0) : tab_off0 + (u32)i);
      val = (w * tab_stride + j) | wfield;
    }
    if (v > half) {
      u32 mag = (1u << c) - v;
      key = mag ? mag - 1 : invalid;
      val = (val | 0x80000000u) ^ flip;
    } else {
      key = v ? v - 1 : invalid;
      val ^= flip;
    }
    if (is_inf) key = invalid;
    my_key[it] = key;
    my_val[it] = val;
    if (key != invalid) my_rank[it] = atomicAdd(&hist[key], 1u);
  }
  __syncthreads();
  // bucket offsets inside this window's slice: lane t owns buckets [t*per, (t+1)*per)
  const u32 per = (n_buckets + kSmallThreads - 1) / kSmallThreads;
  u32 cnt_sum = 0;
  for (u32 q = 0; q < per; q++) {
    u32 b = threadIdx.x * per + q;
    if (b < n_buckets) cnt_sum += hist[b];
  }
  u32 total_cnt;
  u32 cnt_base = block_exclusive_scan(cnt_sum, wave_sums, &total_cnt);
  for (u32 q = 0; q < per; q++) {
    u32 b = threadIdx.x * per + q;
    if (b < n_buckets) {
      off[b] = cnt_base;
      cnt_base += hist[b];
    }
  }
  __syncthreads();
  const u32 slice = gwid * (u32)n;  // this window's part of sorted_vals
#pragma unroll
  for (int it = 0; it < kSmallIters; it++)
    if (my_key[it] != invalid) sorted_vals[slice + off[my_key[it]] + my_rank[it]] = my_val[it];
  for (u32 b = threadIdx.x; b < n_buckets; b += kSmallThreads) {
    const u32 gb = gwid * n_buckets + b, st = slice + off[b];
    bucket_start[gb] = st;
    bucket_end[gb] = st + hist[b];
  }
}

// ---- 1-3 for batched table plans ------------------------------------------------------------
// k_small_front gives every digit window a block of its own, and each of them converts every scalar of
// its group again to take one digit of it.  That buys the shortest chain for ONE problem; a batch has
// blocks enough, so here a block takes kTabFrontWindows digit windows of one group of one problem and
// converts a scalar once for all of them: a counting pass (one histogram per window in LDS), the bucket
// offsets, then the same walk again to place the entries.  Table plans only (8-bit digits, entries name
// table records); the outputs are k_small_front's.
// CB: digit width of the tables, 8 (digits are the bytes of the scalar; 8 windows of 128 buckets per block) or
// 4 (its nibbles; all 64 windows of 8 buckets in one block).
template <int CB>
struct TabFront {
  static_assert(CB == 8 || CB == 4, "byte or nibble digits");
  static constexpr int kAllWindows = 256 / CB;
  static constexpr u32 kBuckets = 1u << (CB - 1);
  static constexpr int kWindows = CB == 8 ? 8 : 64;  // per block
  static constexpr u32 kSlots = (u32)kWindows * kBuckets;
  static_assert(kSlots <= (u32)kSmallThreads && kAllWindows % kWindows == 0, "a bucket per lane at most");
};

template <class C, int CB>
__global__ __launch_bounds__(1024) void k_tab_front(const u32* __restrict__ s0, const uint8_t* __restrict__ inf0, size_t n0,
                                                    const u32* __restrict__ s1, const uint8_t* __restrict__ inf1, size_t n1,
                                                    DigitGroups sel, int group_windows, int wshift,
                                                    u32* __restrict__ sorted_vals, u32* __restrict__ bucket_start,
                                                    u32* __restrict__ bucket_end, u32* __restrict__ heavy_count, u32 tab_stride,
                                                    u32 tab_off0, u32 tab_off1) {
  using Fr = typename C::Fr;
  using TF = TabFront<CB>;
  constexpr int kTabFrontWindows = TF::kWindows;
  constexpr u32 kTabFrontBuckets = TF::kBuckets;
  __shared__ u32 hist[TF::kSlots];
  __shared__ u32 cursor[TF::kSlots];
  __shared__ u32 wave_sums[kSmallThreads / 64 + 1];
  const u32 sess = blockIdx.y;
  s0 += (size_t)sess * sel.s_stride0;
  s1 += (size_t)sess * sel.s_stride1;
  sel.extra_s[0] += (size_t)sess * sel.extra_stride[0];
  sel.extra_s[1] += (size_t)sess * sel.extra_stride[1];
  // problems with points of their own (tab_sess_stride != 0) have identity flags of their own, laid out as the points
  if (inf0) inf0 += (size_t)sess * sel.tab_sess_stride;
  if (inf1) inf1 += (size_t)sess * sel.tab_sess_stride;
  const u32 parts = (u32)group_windows / kTabFrontWindows;  // blocks per group
  const u32 g = blockIdx.x / parts, w_first = (blockIdx.x % parts) * kTabFrontWindows;
  const u32 n_groups = gridDim.x / parts;
  const u32 gw_first = (sess * n_groups + g) * (u32)group_windows + w_first;  // first of the block's windows among all
  const size_t n01 = n0 + n1, n = n01 + sel.n_extra;
  for (u32 b = threadIdx.x; b < kTabFrontWindows * kTabFrontBuckets; b += blockDim.x) hist[b] = 0;
  if (blockIdx.x == 0 && sess == 0 && threadIdx.x == 0) *heavy_count = 0;
  // entry t of the group is point i of the input (see k_small_front)
  const bool split0 = sel.half0 != 0 && sel.lo0 != sel.hi0, split1 = sel.half1 != 0 && sel.lo1 != sel.hi1;
  size_t cnt0, cnt1;
  {
    const size_t q0 = split0 ? n0 / (2 * (size_t)sel.half0) : 0, r0 = split0 ? n0 % (2 * (size_t)sel.half0) : 0;
    const size_t q1 = split1 ? n1 / (2 * (size_t)sel.half1) : 0, r1 = split1 ? n1 % (2 * (size_t)sel.half1) : 0;
    cnt0 = !split0 ? (sel.lo0 == g ? n0 : 0)
           : sel.hi0 == g ? q0 * sel.half0 + (r0 > sel.half0 ? r0 - sel.half0 : 0)
           : sel.lo0 == g ? q0 * sel.half0 + (r0 < sel.half0 ? r0 : sel.half0) : 0;
    cnt1 = !split1 ? (sel.lo1 == g ? n1 : 0)
           : sel.hi1 == g ? q1 * sel.half1 + (r1 > sel.half1 ? r1 - sel.half1 : 0)
           : sel.lo1 == g ? q1 * sel.half1 + (r1 < sel.half1 ? r1 : sel.half1) : 0;
  }
  const size_t cnt = cnt0 + cnt1 + (g < (u32)sel.n_extra ? 1 : 0);
  __syncthreads();
  // PLACE false: count the digits; true: write the entries at the buckets' cursors
  auto walk = [&](auto place_tag) {
    constexpr bool PLACE = decltype(place_tag)::value;
#pragma unroll 1
    for (size_t t = threadIdx.x; t < cnt; t += blockDim.x) {
      size_t i;
      if (t < cnt0) {
        i = split0 ? (t / sel.half0) * 2 * (size_t)sel.half0 + t % sel.half0 + (sel.hi0 == g ? sel.half0 : 0) : t;
      } else if (t < cnt0 + cnt1) {
        const size_t u = t - cnt0;
        i = n0 + (split1 ? (u / sel.half1) * 2 * (size_t)sel.half1 + u % sel.half1 + (sel.hi1 == g ? sel.half1 : 0) : u);
      } else {
        i = n01 + g;
      }
      const bool is_extra = i >= n01, second = i >= n0;
      const u32* sp = second ? s1 + (i - n0) * 8 : s0 + i * 8;
      if (is_extra) sp = sel.extra_s[i - n01];
      const bool is_inf = is_extra ? false : (second ? (inf1 && inf1[i - n0]) : (inf0 && inf0[i]));
      if (is_inf) continue;
      Fe k = fe_from_mont<Fr>(fe_load(sp));
      u32 flip = 0;
      {
        Fe nk = fe_neg<Fr>(k);
        bool gt = false, decided = false;
#pragma unroll
        for (int q = 7; q >= 0; q--) {
          if (!decided && k.v[q] != nk.v[q]) {
            gt = k.v[q] > nk.v[q];
            decided = true;
          }
        }
        if (gt) {
          k = nk;
          flip = 0x80000000u;
        }
      }
      const u32 j = sess * sel.tab_sess_stride +
                    (is_extra ? sel.extra_tab[i - n01] : second ? tab_off1 + (u32)(i - n0) : tab_off0 + (u32)i);
      // signed digits in (-2^(CB-1), 2^(CB-1)] from the bytes or nibbles, low to high, with the carry of the windows below
      u32 carry = 0;
#pragma unroll
      for (int w = 0; w < TF::kAllWindows; w++) {
        const u32 v = CB == 8 ? ((k.v[w >> 2] >> (8 * (w & 3))) & 255u) + carry : ((k.v[w >> 3] >> (4 * (w & 7))) & 15u) + carry;
        carry = v > kTabFrontBuckets ? 1u : 0u;
        if ((u32)w < w_first || (u32)w >= w_first + kTabFrontWindows || (int)w >= group_windows) continue;
        const u32 mag = carry ? (1u << CB) - v : v;
        if (mag == 0) continue;
        const u32 slot = ((u32)w - w_first) * kTabFrontBuckets + mag - 1;
        if constexpr (!PLACE) {
          atomicAdd(&hist[slot], 1u);
        } else {
          u32 val = (u32)w * tab_stride + j;  // no window number: see msm_make_plan
          val = (carry ? val | 0x80000000u : val) ^ flip;
          sorted_vals[atomicAdd(&cursor[slot], 1u)] = val;
        }
      }
    }
  };
  walk(std::false_type{});
  __syncthreads();
  // bucket offsets: lane t owns bucket t of the block (windows x buckets per window, at most one per lane)
  {
    const bool owner = threadIdx.x < TF::kSlots;  // lanes beyond the block's buckets only take part in the scan
    const u32 b = owner ? threadIdx.x : 0u, wl = b / kTabFrontBuckets;
    const u32 mine = owner ? hist[b] : 0u;
    u32 total;
    const u32 before = block_exclusive_scan(mine, wave_sums, &total);  // over all of the block's windows
    // the entries of window wl start at its slice; those of the windows before it in this block are not in it
    __shared__ u32 win_base[kTabFrontWindows];
    if (owner && (b % kTabFrontBuckets) == 0) win_base[wl] = before;
    __syncthreads();
    const u32 gwid = gw_first + wl;
    const u32 st = gwid * (u32)n + (before - win_base[wl]);
    if (owner) cursor[b] = st;
    if (owner && (int)(w_first + wl) < group_windows) {
      bucket_start[(size_t)gwid * kTabFrontBuckets + (b % kTabFrontBuckets)] = st;
      bucket_end[(size_t)gwid * kTabFrontBuckets + (b % kTabFrontBuckets)] = st + mine;
    }
  }
  __syncthreads();
  walk(std::true_type{});
}

// ---- 1-3 for mid-sized problems with narrow windows ---------------------------------------
// The same counting sort over more points than one block can hold: the points are cut into
// slices of kMidSlice; block (slice, window) takes the digits of its slice for window w of
// both groups, ranks them inside the slice with LDS atomics and leaves bucket | rank | sign
// per (window, point) plus its histogram (k_mid_count); one block per (group, window) turns
// the histograms into bucket bounds and per-slice offsets (k_mid_offsets); the first grid
// again scatters the entries (k_mid_scatter).  Three launches instead of about thirty.

__device__ __forceinline__ void window_carry_consts(u32 w, int c, u32* thr, u32* msk) {
  // A value's signed digits d_j in (-half, half] below window w represent its low c*w bits
  // without a carry exactly when those bits are <= T_w = half * sum_{j<w} 2^(c*j).
  for (int t = 0; t < 8; t++) thr[t] = msk[t] = 0;
  for (u32 j = 0; j < w; j++) {
    u32 bit = j * (u32)c + (u32)c - 1;
    if (bit < 256) thr[bit >> 5] |= 1u << (bit & 31);
  }
  u32 low = w * (u32)c;
  for (u32 bit = 0; bit < low && bit < 256; bit++) msk[bit >> 5] |= 1u << (bit & 31);
}

__device__ __forceinline__ u32 digit_group_of(const DigitGroups& sel, size_t i, size_t n0, size_t n01) {
  const bool second = i >= n0;
  const u32 half_sel = second ? sel.half1 : sel.half0;
  if (i >= n01) return (u32)(i - n01);
  if (half_sel == 0) return second ? sel.lo1 : sel.lo0;
  size_t li = second ? i - n0 : i;
  bool up = (li % (2 * (size_t)half_sel)) >= half_sel;
  return up ? (second ? sel.hi1 : sel.hi0) : (second ? sel.lo1 : sel.lo0);
}

// tmp word: bucket (12 bits; n_buckets = no entry) | rank inside the slice << 12 | sign << 31
template <class C>
__global__ __launch_bounds__(1024) void k_mid_count(
    const u32* __restrict__ s0, const uint8_t* __restrict__ inf0, size_t n0, const u32* __restrict__ s1,
    const uint8_t* __restrict__ inf1, size_t n1, const u32* __restrict__ x0, const u32* __restrict__ y0,
    const u32* __restrict__ x1, const u32* __restrict__ y1, u32* __restrict__ packed, DigitGroups sel, int c,
    int group_windows, int groups, u32 n_buckets, u32 n_slices, int tab, u32* __restrict__ tmp,
    u32* __restrict__ ghist, u32* __restrict__ heavy_count) {
  // tab != 0: the points come from window tables, every digit window feeds the one bucket
  // set of its group, and (window, slice) pairs take the place of slices
  using Fr = typename C::Fr;
  __shared__ u32 hist[2 * kMidMaxBuckets];
  __shared__ u32 thr[8], msk[8];
  const u32 slice = blockIdx.x, w = blockIdx.y;
  const size_t n01 = n0 + n1, n = n01 + sel.n_extra;
  const u32 half = 1u << (c - 1), mask = (1u << c) - 1;
  for (u32 b = threadIdx.x; b < 2 * n_buckets; b += kSmallThreads) hist[b] = 0;
  if (threadIdx.x == 0) {
    window_carry_consts(w, c, thr, msk);
    if (slice == 0 && w == 0) *heavy_count = 0;
  }
  __syncthreads();
  const u32 invalid = n_buckets;
  const int bit0 = (int)(w * (u32)c), limb = bit0 >> 5, sh = bit0 & 31;
  for (int it = 0; it < kMidIters; it++) {
    size_t i = (size_t)slice * kMidSlice + (size_t)it * kSmallThreads + threadIdx.x;
    if (i >= n) continue;
    const bool is_extra = i >= n01, second = i >= n0;
    const u32 grp = digit_group_of(sel, i, n0, n01);
    if ((u32)(i % (size_t)gridDim.y) == w && !tab) {  // packed once, spread over the window blocks
      const u32* px = second ? x1 + (i - n0) * 8 : x0 + i * 8;
      const u32* py = second ? y1 + (i - n0) * 8 : y0 + i * 8;
      if (is_extra) {
        px = sel.extra_xy[i - n01];
        py = px + 8;
      }
      fe_store(packed + i * 16, fe_import<C>(fe_load(px)));
      fe_store(packed + i * 16 + 8, fe_import<C>(fe_load(py)));
    }
    const u32* sp = second ? s1 + (i - n0) * 8 : s0 + i * 8;
    if (is_extra) sp = sel.extra_s[i - n01];
    const bool is_inf = is_extra ? false : (second ? (inf1 && inf1[i - n0]) : (inf0 && inf0[i]));
    Fe k = fe_from_mont<Fr>(fe_load(sp));
    u32 flip = 0;
    {
      Fe nk = fe_neg<Fr>(k);
      bool gt = false, decided = false;
#pragma unroll
      for (int t = 7; t >= 0; t--) {
        if (!decided && k.v[t] != nk.v[t]) {
          gt = k.v[t] > nk.v[t];
          decided = true;
        }
      }
      if (gt) {
        k = nk;
        flip = 0x80000000u;
      }
    }
    u32 carry = 0;
    {
      bool gt = false, decided = false;
#pragma unroll
      for (int t = 7; t >= 0; t--) {
        u32 kv = k.v[t] & msk[t], tv = thr[t];
        if (!decided && kv != tv) {
          gt = kv > tv;
          decided = true;
        }
      }
      carry = gt ? 1u : 0u;
    }
    u32 raw = 0;
    if (limb < 8) {
      u32 lo = 0, hi = 0;
#pragma unroll
      for (int t = 0; t < 8; t++) {
        lo = (t == limb) ? k.v[t] : lo;
        hi = (t == limb + 1) ? k.v[t] : hi;
      }
      unsigned long long two = ((unsigned long long)hi << 32) | lo;
      raw = (u32)(two >> sh) & mask;
    }
    u32 v = raw + carry, key, sign = flip;
    if (v > half) {
      u32 mag = (1u << c) - v;
      key = mag ? mag - 1 : invalid;
      sign ^= 0x80000000u;
    } else {
      key = v ? v - 1 : invalid;
    }
    if (is_inf) key = invalid;
    u32 rank = 0;
    if (key != invalid) rank = atomicAdd(&hist[grp * n_buckets + key], 1u);
    tmp[(size_t)w * n + i] = key | (rank << 12) | sign;
  }
  __syncthreads();
  for (u32 q = threadIdx.x; q < (u32)groups * n_buckets; q += kSmallThreads) {
    const u32 g = q / n_buckets, b = q % n_buckets;
    const u32 wid = tab ? g : g * (u32)group_windows + w;
    const u32 sl = tab ? w * n_slices + slice : slice;
    const u32 ns = tab ? (u32)gridDim.y * n_slices : n_slices;
    ghist[((size_t)wid * ns + sl) * n_buckets + b] = hist[q];
  }
}

// k_mid_count for plans without tables with kMidWindowsPerBlock windows of a slice per block: a scalar is
// converted to an integer (a Montgomery product) and folded to |k| <= r/2 once for them, where the
// one-window blocks repeat that for every window.  Slices are `iters` points per lane.
template <class C>
__global__ __launch_bounds__(1024) void k_mid_count_multi(
    const u32* __restrict__ s0, const uint8_t* __restrict__ inf0, size_t n0, const u32* __restrict__ s1,
    const uint8_t* __restrict__ inf1, size_t n1, const u32* __restrict__ x0, const u32* __restrict__ y0,
    const u32* __restrict__ x1, const u32* __restrict__ y1, u32* __restrict__ packed, DigitGroups sel, int c,
    int group_windows, int groups, u32 n_buckets, u32 n_slices, int iters, u32* __restrict__ tmp,
    u32* __restrict__ ghist, u32* __restrict__ heavy_count) {
  using Fr = typename C::Fr;
  __shared__ u32 hist[kMidWindowsPerBlock * 2 * kMidMaxBuckets];
  __shared__ u32 thr[kMidWindowsPerBlock][8], msk[kMidWindowsPerBlock][8];
  const u32 slice = blockIdx.x, w_first = blockIdx.y * kMidWindowsPerBlock;
  const size_t n01 = n0 + n1, n = n01 + sel.n_extra;
  const u32 half = 1u << (c - 1), mask = (1u << c) - 1;
  for (u32 b = threadIdx.x; b < kMidWindowsPerBlock * 2 * n_buckets; b += kSmallThreads) hist[b] = 0;
  if (threadIdx.x < kMidWindowsPerBlock) window_carry_consts(w_first + threadIdx.x, c, thr[threadIdx.x], msk[threadIdx.x]);
  if (threadIdx.x == 0 && slice == 0 && blockIdx.y == 0) *heavy_count = 0;
  __syncthreads();
  const u32 invalid = n_buckets;
  const size_t slice_points = (size_t)kSmallThreads * (size_t)iters;
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
    const size_t i = (size_t)slice * slice_points + (size_t)it * kSmallThreads + threadIdx.x;
    if (i >= n) continue;
    const bool is_extra = i >= n01, second = i >= n0;
    const u32 grp = digit_group_of(sel, i, n0, n01);
    if ((u32)(i % (size_t)gridDim.y) == blockIdx.y) {  // packed once, spread over the window blocks
      const u32* px = second ? x1 + (i - n0) * 8 : x0 + i * 8;
      const u32* py = second ? y1 + (i - n0) * 8 : y0 + i * 8;
      if (is_extra) {
        px = sel.extra_xy[i - n01];
        py = px + 8;
      }
      fe_store(packed + i * 16, fe_import<C>(fe_load(px)));
      fe_store(packed + i * 16 + 8, fe_import<C>(fe_load(py)));
    }
    const u32* sp = second ? s1 + (i - n0) * 8 : s0 + i * 8;
    if (is_extra) sp = sel.extra_s[i - n01];
    const bool is_inf = is_extra ? false : (second ? (inf1 && inf1[i - n0]) : (inf0 && inf0[i]));
    Fe k = fe_from_mont<Fr>(fe_load(sp));
    u32 flip = 0;
    {
      Fe nk = fe_neg<Fr>(k);
      bool gt = false, decided = false;
#pragma unroll
      for (int t = 7; t >= 0; t--) {
        if (!decided && k.v[t] != nk.v[t]) {
          gt = k.v[t] > nk.v[t];
          decided = true;
        }
      }
      if (gt) {
        k = nk;
        flip = 0x80000000u;
      }
    }
#pragma unroll 1
    for (u32 wl = 0; wl < (u32)kMidWindowsPerBlock; wl++) {
      const u32 w = w_first + wl;
      u32 carry = 0;
      {
        bool gt = false, decided = false;
#pragma unroll
        for (int t = 7; t >= 0; t--) {
          u32 kv = k.v[t] & msk[wl][t], tv = thr[wl][t];
          if (!decided && kv != tv) {
            gt = kv > tv;
            decided = true;
          }
        }
        carry = gt ? 1u : 0u;
      }
      const int bit0 = (int)(w * (u32)c), limb = bit0 >> 5, sh = bit0 & 31;
      u32 raw = 0;
      if (limb < 8) {
        u32 lo = 0, hi = 0;
#pragma unroll
        for (int t = 0; t < 8; t++) {
          lo = (t == limb) ? k.v[t] : lo;
          hi = (t == limb + 1) ? k.v[t] : hi;
        }
        unsigned long long two = ((unsigned long long)hi << 32) | lo;
        raw = (u32)(two >> sh) & mask;
      }
      u32 v = raw + carry, key, sign = flip;
      if (v > half) {
        u32 mag = (1u << c) - v;
        key = mag ? mag - 1 : invalid;
        sign ^= 0x80000000u;
      } else {
        key = v ? v - 1 : invalid;
      }
      if (is_inf) key = invalid;
      u32 rank = 0;
      if (key != invalid) rank = atomicAdd(&hist[(wl * 2 + grp) * n_buckets + key], 1u);
      tmp[(size_t)w * n + i] = key | (rank << 12) | sign;
    }
  }
  __syncthreads();
  for (u32 q = threadIdx.x; q < (u32)kMidWindowsPerBlock * (u32)groups * n_buckets; q += kSmallThreads) {
    const u32 wl = q / ((u32)groups * n_buckets), r = q % ((u32)groups * n_buckets);
    const u32 g = r / n_buckets, b = r % n_buckets;
    const u32 wid = g * (u32)group_windows + w_first + wl;
    ghist[((size_t)wid * n_slices + slice) * n_buckets + b] = hist[(wl * 2 + g) * n_buckets + b];
  }
}

// one block of n_buckets lanes per (group, window): ghist[wid][s][b] becomes the number of
// entries of bucket b in the slices before s; the bucket bounds follow from the totals
__global__ __launch_bounds__(256) void k_mid_offsets(u32* __restrict__ ghist, u32 n_buckets, u32 n_slices, size_t stride,
                                                     u32* __restrict__ bucket_start, u32* __restrict__ bucket_end) {
  __shared__ u32 tot[kMidMaxBuckets];
  const u32 wid = blockIdx.x, b = threadIdx.x;
  u32 run = 0;
  for (u32 s = 0; s < n_slices; s++) {
    size_t at = ((size_t)wid * n_slices + s) * n_buckets + b;
    u32 v = ghist[at];
    ghist[at] = run;
    run += v;
  }
  tot[b] = run;
  __syncthreads();
  u32 before = 0;
  for (u32 q = 0; q < b; q++) before += tot[q];
  const u32 st = wid * (u32)stride + before;  // output window wid owns [wid*stride, (wid+1)*stride)
  bucket_start[(size_t)wid * n_buckets + b] = st;
  bucket_end[(size_t)wid * n_buckets + b] = st + run;
}

__global__ __launch_bounds__(1024) void k_mid_scatter(const u32* __restrict__ tmp, const u32* __restrict__ ghist,
                                                      const u32* __restrict__ bucket_start, DigitGroups sel,
                                                      size_t n0, size_t n1, int group_windows, u32 n_buckets,
                                                      u32 n_slices, int wshift, int tab, size_t tab_stride,
                                                      size_t tab_off0, size_t tab_off1,
                                                      u32* __restrict__ sorted_vals, int iters) {
  const u32 slice = blockIdx.x, w = blockIdx.y;
  const u32 sl = tab ? w * n_slices + slice : slice;
  const u32 ns = tab ? (u32)gridDim.y * n_slices : n_slices;
  const size_t n01 = n0 + n1, n = n01 + sel.n_extra;
  for (int it = 0; it < iters; it++) {
    size_t i = (size_t)slice * ((size_t)kSmallThreads * (size_t)iters) + (size_t)it * kSmallThreads + threadIdx.x;
    if (i >= n) continue;
    const u32 t = tmp[(size_t)w * n + i];
    const u32 key = t & 0xfffu;
    if (key >= n_buckets) continue;
    const u32 rank = (t >> 12) & 0x7ffffu, sign = t & 0x80000000u;
    const u32 grp = digit_group_of(sel, i, n0, n01);
    const u32 wid = tab ? grp : grp * (u32)group_windows + w;
    const u32 pos = bucket_start[(size_t)wid * n_buckets + key] + ghist[((size_t)wid * ns + sl) * n_buckets + key] + rank;
    // table plans address record (w, base index); the others the packed copy of point i
    size_t idx = i;
    if (tab) {
      size_t base_index = i >= n01 ? sel.extra_tab[i - n01] : (i >= n0 ? tab_off1 + (i - n0) : tab_off0 + i);
      idx = (size_t)w * tab_stride + base_index;
    }
    sorted_vals[pos] = (u32)idx | (wid << wshift) | sign;
  }
}

// ---- 1-3 for large single sums: two-level counting sort -----------------------------------
// The bucket number splits into a high part (the partition) and kTwoLoBits low bits.
// k_two_digits writes every point's digits once (16 bits each) and counts the entries of each
// (window, partition); k_two_scatter moves a word (point index | low bits | sign) per entry
// into its partition, reserving room for a block's share of a partition with one global
// atomic; k_two_sort_lo finishes each partition in LDS (one block each) and writes the sorted
// entries and the bucket bounds.  Three passes over 2-4 bytes per entry instead of the
// key-value radix sort's passes over 8 and a bounds pass.  The order of entries inside a bucket is arbitrary, the bucket sums are not.
constexpr int kTwoLoBits = kTwoLevelLoBits;
constexpr u32 kTwoLo = 1u << kTwoLoBits;
constexpr u32 kTwoMaxBins = kTwoLevelMaxBins;  // windows * partitions per window, counters of one block
constexpr int kTwoIdxBits = kTwoLevelIdxBits;  // point index inside a word; low bucket bits sit above it
constexpr u32 kTwoDigitPoints = 2048;   // points per block of k_two_digits
constexpr u32 kTwoScatterPoints = 4096; // points per block of k_two_scatter
constexpr int kTwoIters = 15;        // entries per lane that k_two_sort_lo keeps (registers, then 60 KB of LDS)
constexpr uint16_t kTwoNoDigit = 0xffffu;  // (bucket 2^15 - 1, negative digit) cannot occur

// Rank of this lane's entry in the LDS counter hist[key]; every lane of the wave must call.
// When all valid lanes hold the same key (crowded buckets, degenerate scalar sets) the wave
// adds once instead of serialising 64 atomics on one address.
__device__ __forceinline__ u32 lds_rank(u32* hist, u32 key, bool valid) {
  const unsigned long long m = __ballot(valid);
  if (m == 0) return 0;
  const int leader = __ffsll((long long)m) - 1;
  const u32 k0 = (u32)__shfl((int)key, leader, 64);
  if (__all(!valid || key == k0)) {
    u32 base = 0;
    if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(&hist[k0], (u32)__popcll(m));
    base = (u32)__shfl((int)base, leader, 64);
    return base + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
  }
  return valid ? atomicAdd(&hist[key], 1u) : 0u;
}

template <class C>
__global__ __launch_bounds__(1024) void k_two_digits(const u32* __restrict__ s0, const uint8_t* __restrict__ inf0,
                                                     size_t n0, const u32* __restrict__ s1,
                                                     const uint8_t* __restrict__ inf1, size_t n1,
                                                     const u32* __restrict__ x0, const u32* __restrict__ y0,
                                                     const u32* __restrict__ x1, const u32* __restrict__ y1,
                                                     DigitGroups sel, int c, int group_windows, int groups,
                                                     u32 n_hi, u32* __restrict__ packed,
                                                     uint16_t* __restrict__ digits, uint8_t* __restrict__ flips,
                                                     u32* __restrict__ part_count, u32 spread_t) {
  using Fr = typename C::Fr;
  __shared__ u32 hist[kTwoMaxBins];
  const size_t n01 = n0 + n1, n = n01 + sel.n_extra;
  const int n_windows = group_windows;  // digit windows; output window = group * group_windows + w
  const u32 bins = (u32)(groups * group_windows) * n_hi;
  for (u32 q = threadIdx.x; q < bins; q += blockDim.x) hist[q] = 0;
  __syncthreads();
  const u32 half = 1u << (c - 1), mask = (1u << c) - 1;
  for (u32 it = 0; it < kTwoDigitPoints / 1024; it++) {
    const size_t i = (size_t)blockIdx.x * kTwoDigitPoints + (size_t)it * 1024 + threadIdx.x;
    const bool live = i < n;
    const size_t ii = live ? i : 0;
    const bool second = ii >= n0, is_extra = ii >= n01;
    const u32 wbase = (groups > 1 ? digit_group_of(sel, ii, n0, n01) : 0u) * (u32)group_windows;
    Fe k = fe_zero();
    u32 k8 = 0;  // ninth limb of a spread scalar
    bool is_inf = true;
    u32 flip = 0;
    if (live) {
      const u32* px = second ? x1 + (ii - n0) * 8 : x0 + ii * 8;
      const u32* py = second ? y1 + (ii - n0) * 8 : y0 + ii * 8;
      const u32* sp = second ? s1 + (ii - n0) * 8 : s0 + ii * 8;
      if (is_extra) {
        px = sel.extra_xy[ii - n01];
        py = px + 8;
        sp = sel.extra_s[ii - n01];
      }
      fe_store(packed + ii * 16, fe_import<C>(fe_load(px)));
      fe_store(packed + ii * 16 + 8, fe_import<C>(fe_load(py)));
      is_inf = is_extra ? false : (second ? (inf1 && inf1[ii - n0]) : (inf0 && inf0[ii]));
      k = fe_from_mont<Fr>(fe_load(sp));
      if (spread_t) {
        // k + t r, t < spread_t by the point's number (see MsmPlan::spread_t): nine limbs, no folding
        const u32 t = (u32)(((u32)ii * 2654435761u) >> 12) & (spread_t - 1u);
        unsigned long long acc = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) {
          acc += (unsigned long long)t * Fr::mod(q) + k.v[q];
          k.v[q] = (u32)acc;
          acc >>= 32;
        }
        k8 = (u32)acc;
      } else {
        // |k| <= r/2: use r - k with the point negated when that is the smaller value
        Fe nk = fe_neg<Fr>(k);
        bool gt = false, decided = false;
#pragma unroll
        for (int t = 7; t >= 0; t--) {
          if (!decided && k.v[t] != nk.v[t]) {
            gt = k.v[t] > nk.v[t];
            decided = true;
          }
        }
        if (gt) {
          k = nk;
          flip = 1;
        }
      }
      flips[ii] = (uint8_t)flip;
    }
    u32 carry = 0;
    for (int w = 0; w < n_windows; w++) {
      const int bit = w * c, limb = bit >> 5, sh = bit & 31;
      u32 raw = 0;
      if (limb < 9) {
        u32 lo = 0, hi = 0;
#pragma unroll
        for (int t = 0; t < 8; t++) {
          lo = (t == limb) ? k.v[t] : lo;
          hi = (t == limb + 1) ? k.v[t] : hi;
        }
        lo = limb == 8 ? k8 : lo;
        hi = limb == 7 ? k8 : hi;
        unsigned long long two = ((unsigned long long)hi << 32) | lo;
        raw = (u32)(two >> sh) & mask;
      }
      const u32 v = raw + carry;
      u32 mag, neg;
      if (v > half) {
        carry = 1;
        mag = (1u << c) - v;
        neg = 1;
      } else {
        carry = 0;
        mag = v;
        neg = 0;
      }
      const bool valid = live && !is_inf && mag != 0;
      const u32 key = mag - 1;
      if (live) digits[(size_t)w * n + ii] = valid ? (uint16_t)(key | (neg << 15)) : kTwoNoDigit;
      lds_rank(hist, (wbase + (u32)w) * n_hi + (key >> kTwoLoBits), valid);
    }
  }
  __syncthreads();
  for (u32 q = threadIdx.x; q < bins; q += blockDim.x)
    if (hist[q]) atomicAdd(&part_count[q], hist[q]);
}

// Digit window w owns entries [w*n, (w+1)*n) of the sorted array, group 0 first (size0 = points of
// group 0), so output window wid = g*group_windows + w starts at w*n + g*size0.  Returns the start
// of partition `threadIdx.x` of window wid (lanes < n_hi); every lane of the block calls.
__device__ __forceinline__ u32 two_partition_scan(const u32* __restrict__ part_count, u32 wid, u32 group_windows,
                                                  u32 n_hi, size_t n, size_t size0, u32* wave_sums) {
  u32 total;
  const u32 mine = threadIdx.x < n_hi ? part_count[wid * n_hi + threadIdx.x] : 0u;
  const size_t base = (size_t)(wid % group_windows) * n + (size_t)(wid / group_windows) * size0;
  return (u32)base + block_exclusive_scan(mine, wave_sums, &total);
}

// word: point index | low bucket bits << kTwoIdxBits | sign << 31.  The block's entries are
// ordered by partition in LDS first, so that what goes to one partition leaves as one run.
__global__ __launch_bounds__(1024) void k_two_scatter(const uint16_t* __restrict__ digits,
                                                      const uint8_t* __restrict__ flips,
                                                      const u32* __restrict__ part_count, DigitGroups sel,
                                                      size_t n0, size_t n1, size_t size0, u32 group_windows,
                                                      u32 groups, u32 n_hi, u32* __restrict__ cursor,
                                                      u32* __restrict__ words) {
  __shared__ u32 hist[kTwoLo], gbase[kTwoLo], loff[kTwoLo];  // bins of a block: groups * n_hi <= kTwoLo
  const size_t n01 = n0 + n1, n = n01 + sel.n_extra;
  const u32 bins = groups * n_hi;
  __shared__ u32 st_word[kTwoScatterPoints], st_dst[kTwoScatterPoints];
  __shared__ u32 wave_sums[kSmallThreads / 64 + 1];
  const u32 w = blockIdx.y;
  if (threadIdx.x < kTwoLo) hist[threadIdx.x] = 0;
  __syncthreads();
  constexpr int kIters = kTwoScatterPoints / 1024;
  u32 my_word[kIters], my_hi[kIters], my_rank[kIters];
#pragma unroll
  for (int it = 0; it < kIters; it++) {
    const size_t i = (size_t)blockIdx.x * kTwoScatterPoints + (size_t)it * 1024 + threadIdx.x;
    u32 d = kTwoNoDigit;
    u32 flip = 0;
    if (i < n) {
      d = digits[(size_t)w * n + i];
      flip = flips[i];
    }
    const bool valid = d != kTwoNoDigit;
    const u32 key = d & 0x7fffu, sign = (d >> 15) ^ flip;
    const u32 grp = groups > 1 && i < n ? digit_group_of(sel, i, n0, n01) : 0u;
    const u32 bin = grp * n_hi + (key >> kTwoLoBits);
    my_hi[it] = valid ? bin : bins;
    my_word[it] = (u32)i | ((key & (kTwoLo - 1)) << kTwoIdxBits) | (sign << 31);
    my_rank[it] = lds_rank(hist, bin, valid);
  }
  __syncthreads();
  // partition starts of this digit window in group 0, then group 1
  u32 part = two_partition_scan(part_count, w, group_windows, n_hi, n, size0, wave_sums);
  if (groups > 1) {
    __shared__ u32 part1[kTwoLo];
    const u32 p1 = two_partition_scan(part_count, group_windows + w, group_windows, n_hi, n, size0, wave_sums);
    if (threadIdx.x < n_hi) part1[threadIdx.x] = p1;
    __syncthreads();
    if (threadIdx.x >= n_hi && threadIdx.x < bins) part = part1[threadIdx.x - n_hi];
  }
  u32 total;
  const u32 mine = threadIdx.x < bins ? hist[threadIdx.x] : 0u;
  const u32 before = block_exclusive_scan(mine, wave_sums, &total);
  if (threadIdx.x < bins) {
    const u32 wid = (threadIdx.x / n_hi) * group_windows + w;
    loff[threadIdx.x] = before;
    gbase[threadIdx.x] = mine ? part + atomicAdd(&cursor[wid * n_hi + threadIdx.x % n_hi], mine) : 0u;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < kIters; it++)
    if (my_hi[it] < bins) {
      const u32 j = loff[my_hi[it]] + my_rank[it];
      st_word[j] = my_word[it];
      st_dst[j] = gbase[my_hi[it]] + my_rank[it];
    }
  __syncthreads();
  for (u32 j = threadIdx.x; j < total; j += blockDim.x) words[st_dst[j]] = st_word[j];
}

// one block per (partition, window): counting sort by the low bucket bits, bucket bounds.
// Partitions of up to kTwoIters * 1024 entries are read once and leave through LDS in order.
__global__ __launch_bounds__(1024) void k_two_sort_lo(const u32* __restrict__ words,
                                                      const u32* __restrict__ part_count, size_t n, size_t size0,
                                                      u32 group_windows, u32 n_hi, u32 n_buckets, int wshift,
                                                      u32* __restrict__ sorted_vals,
                                                      u32* __restrict__ bucket_start, u32* __restrict__ bucket_end) {
  __shared__ u32 hist[kTwoLo], off[kTwoLo];
  __shared__ u32 wave_sums[kSmallThreads / 64 + 1];
  __shared__ u32 staged[kTwoIters * 1024];
  __shared__ u32 s_start;
  const u32 h = blockIdx.x, w = blockIdx.y;
  const u32 m = part_count[w * n_hi + h];
  if (threadIdx.x < kTwoLo) hist[threadIdx.x] = 0;
  {
    const u32 part = two_partition_scan(part_count, w, group_windows, n_hi, n, size0, wave_sums);  // syncs inside
    if (threadIdx.x == h) s_start = part;
  }
  __syncthreads();
  const u32 s = s_start;
  const u32 idx_mask = (1u << kTwoIdxBits) - 1u, wbits = w << wshift;
  const bool cached = m <= (u32)kTwoIters * 1024u;
  u32 my_word[kTwoIters], my_rank[kTwoIters];
  if (cached) {
#pragma unroll
    for (int it = 0; it < kTwoIters; it++) {
      const u32 p = (u32)it * 1024u + threadIdx.x;
      const bool valid = p < m;
      const u32 word = valid ? words[s + p] : 0u;
      my_word[it] = word;
      my_rank[it] = lds_rank(hist, (word >> kTwoIdxBits) & (kTwoLo - 1), valid);
    }
  } else {
    for (u32 p0 = 0; p0 < m; p0 += 1024u) {
      const u32 p = p0 + threadIdx.x;
      const bool valid = p < m;
      const u32 word = valid ? words[s + p] : 0u;
      lds_rank(hist, (word >> kTwoIdxBits) & (kTwoLo - 1), valid);
    }
  }
  __syncthreads();
  u32 total;
  const u32 mine = threadIdx.x < kTwoLo ? hist[threadIdx.x] : 0u;
  const u32 before = block_exclusive_scan(mine, wave_sums, &total);
  if (threadIdx.x < kTwoLo) {
    off[threadIdx.x] = before;
    const size_t gb = (size_t)w * n_buckets + (size_t)h * kTwoLo + threadIdx.x;
    bucket_start[gb] = s + before;
    bucket_end[gb] = s + before + mine;
  }
  __syncthreads();
  if (cached) {
#pragma unroll
    for (int it = 0; it < kTwoIters; it++) {
      const u32 p = (u32)it * 1024u + threadIdx.x;
      if (p < m) {
        const u32 word = my_word[it];
        staged[off[(word >> kTwoIdxBits) & (kTwoLo - 1)] + my_rank[it]] = (word & idx_mask) | wbits | (word & 0x80000000u);
      }
    }
    __syncthreads();
    for (u32 p = threadIdx.x; p < m; p += blockDim.x) sorted_vals[s + p] = staged[p];
  } else {
    // too many entries to keep: read them again, off[] now counts upwards as the cursor
    for (u32 p0 = 0; p0 < m; p0 += 1024u) {
      const u32 p = p0 + threadIdx.x;
      const bool valid = p < m;
      const u32 word = valid ? words[s + p] : 0u;
      const u32 pos = lds_rank(off, (word >> kTwoIdxBits) & (kTwoLo - 1), valid);
      if (valid) sorted_vals[s + pos] = (word & idx_mask) | wbits | (word & 0x80000000u);
    }
  }
}

// ---- 3. bucket bounds ---------------------------------------------------------
__global__ __launch_bounds__(256) void k_bounds(const u32* __restrict__ keys, const u32* __restrict__ vals,
                                                size_t entries, u32 n_buckets, int wshift, u32 window_base,
                                                u32* __restrict__ start, u32* __restrict__ end) {
  // The sort key is the bucket number alone; the input was window-major and the sort
  // is stable, so equal keys stay in window order: runs of equal (bucket, window) are
  // contiguous.  start/end are indexed window-major (w * n_buckets + bucket).
  size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= entries) return;
  u32 k = keys[p];
  if (k >= n_buckets) return;  // zero digit / identity base
  const u32 wmask = 0x7fffffffu >> wshift << wshift;
  u32 w = vals[p] & wmask;
  u32 b = ((w >> wshift) - window_base) * n_buckets + k;
  if (p == 0 || keys[p - 1] != k || (vals[p - 1] & wmask) != w) start[b] = (u32)p;
  if (p + 1 == entries || keys[p + 1] != k || (vals[p + 1] & wmask) != w) end[b] = (u32)(p + 1);
}

// ---- 4. bucket accumulation -----------------------------------------------------
BP_DEV void xyzz_store_soa(u32* base, size_t nbt, size_t b, const Xyzz& p) {
  fe_store(base + (0 * nbt + b) * 8, p.x);
  fe_store(base + (1 * nbt + b) * 8, p.y);
  fe_store(base + (2 * nbt + b) * 8, p.zz);
  fe_store(base + (3 * nbt + b) * 8, p.zzz);
}
BP_DEV Xyzz xyzz_load_soa(const u32* base, size_t nbt, size_t b) {
  Xyzz p;
  p.x = fe_load(base + (0 * nbt + b) * 8);
  p.y = fe_load(base + (1 * nbt + b) * 8);
  p.zz = fe_load(base + (2 * nbt + b) * 8);
  p.zzz = fe_load(base + (3 * nbt + b) * 8);
  return p;
}
BP_DEV void xyzz_store_aos(u32* p32, const Xyzz& p) {
  fe_store(p32, p.x);
  fe_store(p32 + 8, p.y);
  fe_store(p32 + 16, p.zz);
  fe_store(p32 + 24, p.zzz);
}
BP_DEV Xyzz xyzz_load_aos(const u32* p32) {
  Xyzz p;
  p.x = fe_load(p32);
  p.y = fe_load(p32 + 8);
  p.zz = fe_load(p32 + 16);
  p.zzz = fe_load(p32 + 24);
  return p;
}

// a point that leaves the pipeline: coordinates back to the boundary form (see CurveT)
template <class C>
BP_DEV Xyzz xyzz_export(const Xyzz& p) {
  Xyzz r;
  r.x = fe_export<C>(p.x);
  r.y = fe_export<C>(p.y);
  r.zz = fe_export<C>(p.zz);
  r.zzz = fe_export<C>(p.zzz);
  return r;
}

// Work is cut into chunks of at most `chunk` sorted entries that never cross a
// bucket boundary, so a bucket of any size (the top window's buckets are denser;
// degenerate scalar sets put every point in one bucket) costs every lane at most
// `chunk` additions.  nch[b] = chunks of bucket b; choff = its exclusive scan.
// Chunk descriptors: for chunk t its bucket, first sorted position and a sort key
// (chunk - length, so that longer chunks come first; kChunkKeyInvalid past the
// last chunk).  Sorting the chunk ids by that key gives every wave chunks of equal
// length, which removes the SIMD idle time of uneven chunks.

// ---- chunk table without library calls -------------------------------------------------------
// The same table as k_chunk_counts + scan + k_chunk_desc + a sort of the chunk ids by length, in
// four short launches written for this data: the sort key has at most 64 values, so the "sort" is
// a counting sort whose histogram falls out of the pass that counts the chunks.
//   k_ct_count    per tile of kCtTile buckets: nch[b], the tile's chunk total and its histogram of keys
//   k_ct_scan     one block: tile offsets, first slot of every key, total (choff[nbt]), queue reset
//   k_ct_emit     per bucket: choff[b] = tile offset + scan inside the tile; its chunks' first
//                 entries and keys; every chunk id goes to the next free slot of its key (rank in
//                 LDS, one global reservation per key and block)
// tmp layout (u32): tile_sum[n_tiles] | tile_off[n_tiles] | tile_hist[n_tiles][64] | cursor[64]

__device__ __forceinline__ u32 chunk_key(u32 chunk, u32 len) {  // 0 = full chunk ... kChunkKeyInvalid - 1 = shortest
  return chunk < kChunkKeyInvalid ? chunk - len : (u32)((size_t)(chunk - len) * (kChunkKeyInvalid - 1) / chunk);
}

__global__ __launch_bounds__(kSmallThreads) void k_ct_count(const u32* __restrict__ start, const u32* __restrict__ end,
                                                            size_t nbt, u32 chunk, u32 n_tiles, u32* __restrict__ nch,
                                                            u32* __restrict__ tmp) {
  __shared__ u32 hist[kCtKeys];
  __shared__ u32 wave_sums[kSmallThreads / 64 + 1];
  if (threadIdx.x < kCtKeys) hist[threadIdx.x] = 0;
  __syncthreads();
  u32 mine = 0, full_sum = 0;
#pragma unroll
  for (u32 i = 0; i < kCtPer; i++) {
    const size_t b = (size_t)blockIdx.x * kCtTile + (size_t)i * kSmallThreads + threadIdx.x;
    if (b > nbt) continue;
    u32 c = 0;
    if (b < nbt) {
      const u32 len = end[b] - start[b];
      const u32 full = len / chunk, rest = len - full * chunk;
      c = full + (rest ? 1u : 0u);
      full_sum += full;
      if (rest) atomicAdd(&hist[chunk_key(chunk, rest)], 1u);
    }
    nch[b] = c;
    mine += c;
  }
  // full chunks all share key 0: one LDS addition per wave
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) full_sum += __shfl_down(full_sum, d, 64);
  if ((threadIdx.x & 63) == 0 && full_sum) atomicAdd(&hist[0], full_sum);
  u32 total;
  (void)block_exclusive_scan(mine, wave_sums, &total);
  if (threadIdx.x == 0) tmp[blockIdx.x] = total;
  if (threadIdx.x < kCtKeys) tmp[2 * (size_t)n_tiles + (size_t)blockIdx.x * kCtKeys + threadIdx.x] = hist[threadIdx.x];
}

__global__ __launch_bounds__(kSmallThreads) void k_ct_scan(u32 n_tiles, size_t nbt, u32* __restrict__ tmp,
                                                           u32* __restrict__ choff, u32* __restrict__ heavy_count) {
  __shared__ u32 wave_sums[kSmallThreads / 64 + 1];
  __shared__ u32 keys[kCtKeys];
  u32 run = 0;
  for (u32 base = 0; base < n_tiles; base += kSmallThreads) {
    const u32 t = base + threadIdx.x;
    const u32 v = t < n_tiles ? tmp[t] : 0;
    u32 total;
    const u32 ex = block_exclusive_scan(v, wave_sums, &total);
    if (t < n_tiles) tmp[n_tiles + t] = run + ex;
    run += total;
  }
  if (threadIdx.x == 0) choff[nbt] = run;  // number of chunks
  // the queue counter and, behind it, the per-window arrival counters of the row/column reduction
  if (threadIdx.x < 64) heavy_count[threadIdx.x] = 0;
  // first slot of every key: chunks of key k start after all chunks of smaller keys.  Lane
  // (key, part) sums every 16th tile's count of its key; the 16 parts are joined in LDS.
  {
    __shared__ u32 part_sum[kSmallThreads];
    const u32 k = threadIdx.x % kCtKeys, part = threadIdx.x / kCtKeys;
    u32 sum = 0;
    for (u32 t = part; t < n_tiles; t += kSmallThreads / kCtKeys) sum += tmp[2 * (size_t)n_tiles + (size_t)t * kCtKeys + k];
    part_sum[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x < kCtKeys) {
      u32 all = 0;
      for (u32 q = 0; q < kSmallThreads / kCtKeys; q++) all += part_sum[q * kCtKeys + threadIdx.x];
      keys[threadIdx.x] = all;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 at = 0;
    u32* cursor = tmp + 2 * (size_t)n_tiles + (size_t)n_tiles * kCtKeys;
    for (u32 k = 0; k < kCtKeys; k++) {
      cursor[k] = at;
      at += keys[k];
    }
  }
}

// One lane per bucket (kCtPer consecutive buckets): choff[b] from the tile offset and a scan
// inside the tile, then the bucket's chunk descriptors and their slots in the order sorted by
// key.  A bucket's full chunks (key 0) take consecutive slots; a bucket with more than
// kCtSerial chunks is emitted by its whole wave.
constexpr u32 kCtSerial = 8;
__global__ __launch_bounds__(kSmallThreads) void k_ct_emit(const u32* __restrict__ start, const u32* __restrict__ end,
                                                           const u32* __restrict__ nch, size_t nbt, u32 chunk,
                                                           u32 n_tiles, u32* __restrict__ tmp, u32* __restrict__ choff,
                                                           u32* __restrict__ ckey, u32* __restrict__ cval,
                                                           u32* __restrict__ desc_b, u32* __restrict__ desc_s) {
  __shared__ u32 wave_sums[kSmallThreads / 64 + 1];
  __shared__ u32 hist[kCtKeys], base[kCtKeys];
  if (threadIdx.x < kCtKeys) hist[threadIdx.x] = 0;
  const size_t b0 = (size_t)blockIdx.x * kCtTile + (size_t)threadIdx.x * kCtPer;
  u32 c[kCtPer], mine = 0;
#pragma unroll
  for (u32 i = 0; i < kCtPer; i++) {
    c[i] = b0 + i < nbt ? nch[b0 + i] : 0;
    mine += c[i];
  }
  u32 total;
  u32 at = tmp[n_tiles + blockIdx.x] + block_exclusive_scan(mine, wave_sums, &total);  // syncs: hist is clear
  u32 first[kCtPer], s0[kCtPer], full[kCtPer], rest[kCtPer], rank0[kCtPer], rank1[kCtPer];
  u32 full_mine = 0;
#pragma unroll
  for (u32 i = 0; i < kCtPer; i++) {
    first[i] = at;
    at += c[i];
    s0[i] = full[i] = rest[i] = rank1[i] = 0;
    if (b0 + i < nbt) {
      choff[b0 + i] = first[i];
      s0[i] = start[b0 + i];
      const u32 len = end[b0 + i] - s0[i];
      full[i] = len / chunk;
      rest[i] = len - full[i] * chunk;
      if (rest[i]) rank1[i] = atomicAdd(&hist[chunk_key(chunk, rest[i])], 1u);
    }
    rank0[i] = full_mine;
    full_mine += full[i];
  }
  // slots of the full chunks (key 0): a prefix sum over the wave, one LDS addition per wave
  {
    const int lane = threadIdx.x & 63;
    u32 inc = full_mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const u32 o = __shfl_up(inc, d, 64);
      if (lane >= d) inc += o;
    }
    u32 wave_first = 0;
    if (lane == 63 && inc) wave_first = atomicAdd(&hist[0], inc);
    wave_first = __shfl(wave_first, 63, 64);
#pragma unroll
    for (u32 i = 0; i < kCtPer; i++) rank0[i] += wave_first + inc - full_mine;
  }
  __syncthreads();
  if (threadIdx.x < kCtKeys && hist[threadIdx.x]) {
    u32* cursor = tmp + 2 * (size_t)n_tiles + (size_t)n_tiles * kCtKeys;
    base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], hist[threadIdx.x]);
  }
  __syncthreads();
#pragma unroll
  for (u32 i = 0; i < kCtPer; i++) {
    const u32 bkt = (u32)(b0 + i);
    if (rest[i]) {
      const u32 key = chunk_key(chunk, rest[i]), t = first[i] + full[i], pos = base[key] + rank1[i];
      desc_b[t] = bkt;
      desc_s[t] = s0[i] + full[i] * chunk;
      ckey[pos] = key;
      cval[pos] = t;
    }
    const u32 slot0 = full[i] ? base[0] + rank0[i] : 0;
    if (full[i] <= kCtSerial) {
      for (u32 j = 0; j < full[i]; j++) {
        desc_b[first[i] + j] = bkt;
        desc_s[first[i] + j] = s0[i] + j * chunk;
        ckey[slot0 + j] = 0;
        cval[slot0 + j] = first[i] + j;
      }
    }
    // crowded buckets: the wave writes them together, one after the other
    unsigned long long big = __ballot(full[i] > kCtSerial);
    while (big) {
      const int src = __ffsll((long long)big) - 1;
      big &= big - 1;
      const u32 n_full = __shfl(full[i], src, 64), f0 = __shfl(first[i], src, 64), st0 = __shfl(s0[i], src, 64);
      const u32 sl0 = __shfl(slot0, src, 64), bk = __shfl(bkt, src, 64);
      for (u32 j = threadIdx.x & 63; j < n_full; j += 64) {
        desc_b[f0 + j] = bk;
        desc_s[f0 + j] = st0 + j * chunk;
        ckey[sl0 + j] = 0;
        cval[sl0 + j] = f0 + j;
      }
    }
  }
}

template <class C>
__global__ __launch_bounds__(128) void k_chunk_sums(const u32* __restrict__ packed,
                                                    const u32* __restrict__ vals,
                                                    const u32* __restrict__ choff_total,
                                                    const u32* __restrict__ cval,
                                                    const u32* __restrict__ desc_b,
                                                    const u32* __restrict__ desc_s,
                                                    const u32* __restrict__ bucket_end,
                                                    const u32* __restrict__ nch, size_t nbt, u32 chunk,
                                                    size_t nc_cap, u32 idx_mask, size_t n_entries,
                                                    u32* __restrict__ buckets, u32* __restrict__ chunk_sums) {
  using F = typename C::Fq;
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nc_cap || j >= choff_total[0]) return;  // slots past the last chunk are unused
  u32 t = cval[j];
  u32 b = desc_b[t];
  u32 s = desc_s[t], e = s + chunk;
  {
    u32 be = bucket_end[b];
    e = e < be ? e : be;
  }
  Xyzz acc = xyzz_identity<F>();
  // A lane takes its entries eight at a time (two 16-byte loads at a 4-byte-aligned address) and
  // hands them out of registers: read one by one, every step of a wave touches 64 different lines of
  // `vals` again, which by then have left the caches (38 bytes per entry fetched instead of 4).
  u32 w0 = 0, w1 = 0, w2 = 0, w3 = 0, w4 = 0, w5 = 0, w6 = 0, w7 = 0;
  for (u32 p = s; p < e; p++) {
    if (((p - s) & 7u) == 0) {
      if ((size_t)p + 8 <= n_entries) {
        uint4 lo, hi;
        __builtin_memcpy(&lo, vals + p, 16);
        __builtin_memcpy(&hi, vals + p + 4, 16);
        w0 = lo.x; w1 = lo.y; w2 = lo.z; w3 = lo.w;
        w4 = hi.x; w5 = hi.y; w6 = hi.z; w7 = hi.w;
      } else {  // the last entries of the array
        w0 = vals[p];
        w1 = (size_t)p + 1 < n_entries ? vals[p + 1] : 0;
        w2 = (size_t)p + 2 < n_entries ? vals[p + 2] : 0;
        w3 = (size_t)p + 3 < n_entries ? vals[p + 3] : 0;
        w4 = (size_t)p + 4 < n_entries ? vals[p + 4] : 0;
        w5 = (size_t)p + 5 < n_entries ? vals[p + 5] : 0;
        w6 = (size_t)p + 6 < n_entries ? vals[p + 6] : 0;
        w7 = 0;
      }
    }
    const u32 v = w0;
    w0 = w1; w1 = w2; w2 = w3; w3 = w4; w4 = w5; w5 = w6; w6 = w7;
    size_t idx = v & idx_mask;
    Fe qx = fe_load(packed + idx * 16);
    Fe qy = fe_load(packed + idx * 16 + 8);
    if (v >> 31) qy = fe_neg<F>(qy);
    xyzz_madd<F>(acc, qx, qy);
  }
  if (nch[b] == 1)
    xyzz_store_soa(buckets, nbt, b, acc);
  else
    xyzz_store_soa(chunk_sums, nc_cap, t, acc);
}

// Buckets with 0 chunks become the identity, with 2..kFinishSerial chunks are
// summed by one lane, with more are queued for k_bucket_heavy.
constexpr u32 kFinishSerial = 8;
// queued buckets with up to this many chunks get one wave each, larger ones a block each
constexpr u32 kHeavyWaveMax = 1024;

template <class C>
__global__ __launch_bounds__(128) void k_bucket_finish(const u32* __restrict__ nch,
                                                       const u32* __restrict__ choff, size_t nbt,
                                                       size_t nc_cap, const u32* __restrict__ chunk_sums,
                                                       u32* __restrict__ buckets, u32* __restrict__ heavy,
                                                       u32* __restrict__ heavy_count) {
  using F = typename C::Fq;
  size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nbt) return;
  u32 n = nch[b];
  if (n == 1) return;
  if (n > kFinishSerial) {
    u32 slot = atomicAdd(heavy_count, 1u);
    heavy[slot] = (u32)b;
    return;
  }
  Xyzz acc = xyzz_identity<F>();
  u32 off = choff[b];
  for (u32 k = 0; k < n; k++) {
    Xyzz v = xyzz_load_soa(chunk_sums, nc_cap, off + k);
    xyzz_add<F>(acc, v);
  }
  xyzz_store_soa(buckets, nbt, b, acc);
}

// one block per queued bucket (grid-stride over the queue)
template <class C>
__global__ __launch_bounds__(256) void k_bucket_heavy(const u32* __restrict__ nch,
                                                      const u32* __restrict__ choff, size_t nbt,
                                                      size_t nc_cap, const u32* __restrict__ chunk_sums,
                                                      u32* __restrict__ buckets,
                                                      const u32* __restrict__ heavy,
                                                      const u32* __restrict__ heavy_count) {
  using F = typename C::Fq;
  __shared__ u32 lds[256 * 32];
  u32 count = *heavy_count;
  for (u32 q = blockIdx.x; q < count; q += gridDim.x) {
    u32 b = heavy[q];
    u32 n = nch[b], off = choff[b];
    if (n <= kHeavyWaveMax) continue;  // k_bucket_heavy_wave's share (uniform over the block)
    Xyzz acc = xyzz_identity<F>();
    for (u32 k = threadIdx.x; k < n; k += blockDim.x) {
      Xyzz v = xyzz_load_soa(chunk_sums, nc_cap, off + k);
      xyzz_add<F>(acc, v);
    }
    xyzz_store_aos(lds + threadIdx.x * 32, acc);
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) {
        Xyzz v = xyzz_load_aos(lds + (threadIdx.x + o) * 32);
        xyzz_add<F>(acc, v);
        xyzz_store_aos(lds + threadIdx.x * 32, acc);
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) xyzz_store_soa(buckets, nbt, b, acc);
    __syncthreads();
  }
}

// ---- 4'. bucket accumulation for small problems ---------------------------------------
// With few buckets there are lanes to spare, and what counts is the length of the
// dependent chain: `lanes` (a power of two <= 64) lanes share a bucket, each adds every
// lanes-th entry, and a shuffle tree joins them.  Buckets with more than `heavy_from`
// entries (16 per lane) are queued for a whole block.

BP_DEV Xyzz xyzz_shfl_down(const Xyzz& p, int delta, int width) {
  Xyzz r;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    r.x.v[i] = __shfl_down(p.x.v[i], delta, width);
    r.y.v[i] = __shfl_down(p.y.v[i], delta, width);
    r.zz.v[i] = __shfl_down(p.zz.v[i], delta, width);
    r.zzz.v[i] = __shfl_down(p.zzz.v[i], delta, width);
  }
  return r;
}

// one wave per queued bucket of moderate size (grid-stride over the queue): when a whole
// window is crowded (a top window that the scalar field only partly fills) thousands of
// buckets are queued, far too many for a block each
template <class C>
__global__ __launch_bounds__(256) void k_bucket_heavy_wave(const u32* __restrict__ nch,
                                                           const u32* __restrict__ choff, size_t nbt,
                                                           size_t nc_cap, const u32* __restrict__ chunk_sums,
                                                           u32* __restrict__ buckets,
                                                           const u32* __restrict__ heavy,
                                                           const u32* __restrict__ heavy_count) {
  using F = typename C::Fq;
  const u32 count = *heavy_count;
  const u32 lane = threadIdx.x & 63;
  const u32 waves = gridDim.x * (blockDim.x >> 6);
  for (u32 q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); q < count; q += waves) {
    const u32 b = heavy[q];
    const u32 n = nch[b], off = choff[b];
    if (n > kHeavyWaveMax) continue;  // uniform over the wave
    Xyzz acc = xyzz_identity<F>();
    for (u32 k = lane; k < n; k += 64) {
      Xyzz v = xyzz_load_soa(chunk_sums, nc_cap, off + k);
      xyzz_add<F>(acc, v);
    }
    for (int o = 32; o > 0; o >>= 1) {
      Xyzz other = xyzz_shfl_down(acc, o, 64);
      if ((int)lane < o) xyzz_add<F>(acc, other);
    }
    if (lane == 0) xyzz_store_soa(buckets, nbt, b, acc);
  }
}

// The first kTreeHeavyBlocks blocks of the launch take the crowded buckets (more than
// `heavy_from` entries, up to kTreeHeavyMax) a whole block at a time, while the other blocks
// run the lane groups: block h looks at buckets h, h + kTreeHeavyBlocks, ... so a crowded
// window is spread over all of them.  Buckets beyond kTreeHeavyMax are queued for
// k_bucket_heavy_entries.
constexpr u32 kTreeHeavyBlocks = 128;
constexpr u32 kTreeHeavyMax = 8192;
constexpr u32 kTreeHeavyList = 128;  // buckets a block looks at with the default windows: <= 58 * 256 / 128

// A crowded bucket (more than heavy_from entries, up to heavy_max) taken by a whole 256-lane block:
// block h of the first kTreeHeavyBlocks looks at buckets h, h + kTreeHeavyBlocks, ...
template <class C>
BP_DEV void tree_heavy_block(const u32* __restrict__ packed, const u32* __restrict__ vals,
                             const u32* __restrict__ start, const u32* __restrict__ end, size_t nbt, u32 heavy_from,
                             u32 heavy_max, u32 idx_mask, u32* __restrict__ buckets, u32* __restrict__ heavy,
                             u32* __restrict__ heavy_count) {
  using F = typename C::Fq;
  __shared__ u32 list[kTreeHeavyList];
  __shared__ u32 n_list;
  __shared__ u32 wave_sum[4 * 32];
  if (threadIdx.x == 0) n_list = 0;
  __syncthreads();
  // block h looks at one bucket of every run of kTreeHeavyBlocks: in run q the one at offset
  // h + q + q / kTreeHeavyBlocks (mod kTreeHeavyBlocks), so that crowded buckets which sit a multiple of
  // kTreeHeavyBlocks apart (bucket 0 of the same window of every problem of a batch) go to different blocks
  for (size_t q = threadIdx.x; q * kTreeHeavyBlocks < nbt; q += blockDim.x) {
    const size_t b = q * kTreeHeavyBlocks + ((blockIdx.x + q + q / kTreeHeavyBlocks) & (kTreeHeavyBlocks - 1));
    if (b >= nbt) continue;
    const u32 cnt = end[b] - start[b];
    if (cnt > heavy_from && cnt <= heavy_max) {
      const u32 slot = atomicAdd(&n_list, 1u);
      if (slot < kTreeHeavyList)
        list[slot] = (u32)b;
      else  // more than the list holds (very wide windows): leave it to the queue
        heavy[atomicAdd(heavy_count, 1u)] = (u32)b;
    }
  }
  __syncthreads();
  const u32 count = n_list < kTreeHeavyList ? n_list : kTreeHeavyList;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (u32 q = 0; q < count; q++) {
    const u32 b = list[q];
    const u32 s = start[b], e = end[b];
    Xyzz acc = xyzz_identity<F>();
    for (u32 p = s + threadIdx.x; p < e; p += blockDim.x) {
      u32 v = vals[p];
      size_t idx = v & idx_mask;
      Fe qx = fe_load(packed + idx * 16);
      Fe qy = fe_load(packed + idx * 16 + 8);
      if (v >> 31) qy = fe_neg<F>(qy);
      xyzz_madd<F>(acc, qx, qy);
    }
    for (int off = 32; off > 0; off >>= 1) {
      Xyzz o = xyzz_shfl_down(acc, off, 64);
      if (lane < off) xyzz_add<F>(acc, o);
    }
    if (lane == 0) xyzz_store_aos(wave_sum + wave * 32, acc);
    __syncthreads();
    if (wave == 0) {
      acc = lane < 4 ? xyzz_load_aos(wave_sum + lane * 32) : xyzz_identity<F>();
      for (int off = 2; off > 0; off >>= 1) {
        Xyzz o = xyzz_shfl_down(acc, off, 64);
        if (lane < off) xyzz_add<F>(acc, o);
      }
      if (lane == 0) xyzz_store_soa(buckets, nbt, b, acc);
    }
    __syncthreads();
  }
}

// QT: the lanes of a bucket are joined by additions that four lanes share (bp_quad.h) instead of a
// shuffle tree of one-lane additions.  The tree is the dependent chain that sets the time of a small
// launch (one wave or less per SIMD): a quad addition is 4 multiplications deep, a one-lane
// addition 14.  Every lane leaves its partial sum in LDS; addition `task` of a level goes to quad
// `task mod 64` of the block.
template <class C, bool QT>
__global__ __launch_bounds__(256) void k_bucket_tree(const u32* __restrict__ packed, const u32* __restrict__ vals,
                                                     const u32* __restrict__ start, const u32* __restrict__ end,
                                                     size_t nbt, int lanes, u32 heavy_from, u32 heavy_max,
                                                     u32 idx_mask, u32* __restrict__ buckets,
                                                     u32* __restrict__ heavy, u32* __restrict__ heavy_count) {
  using F = typename C::Fq;
  if (blockIdx.x < kTreeHeavyBlocks) {
    tree_heavy_block<C>(packed, vals, start, end, nbt, heavy_from, heavy_max, idx_mask, buckets, heavy, heavy_count);
    return;
  }
  size_t t = (size_t)(blockIdx.x - kTreeHeavyBlocks) * blockDim.x + threadIdx.x;
  size_t b = t / (size_t)lanes;
  const u32 sub = (u32)(t % (size_t)lanes);
  bool live = b < nbt;  // whole lane groups are live or not; all lanes take part in the shuffles
  u32 s = 0, e = 0;
  if (live) {
    s = start[b];
    e = end[b];
  }
  if constexpr (!QT) {
    // A wave runs as long as its fullest bucket.  The block's buckets are dealt out again by size (a counting sort
    // on the entries per lane in LDS, fullest first), so that the lanes of a wave hold about the same number of
    // entries (batch of 64 range proofs: 17.2 -> 16.4 ms with a lane per bucket; blocks of 512 or 1024 lanes,
    // whose waves would come closer still to the mean, lose: 17.1, 17.5 ms).
    constexpr u32 kBins = 64;
    __shared__ u32 bin_count[kBins], bin_start[kBins];
    __shared__ u32 order[256];
    const u32 tid = threadIdx.x, grp = tid / (u32)lanes;
    if (tid < kBins) bin_count[tid] = 0;
    __syncthreads();
    u32 bin = 0, rank = 0;
    if (sub == 0) {
      const u32 share = live ? (e - s + (u32)lanes - 1) / (u32)lanes : 0u;
      bin = kBins - 1 - (share < kBins ? share : kBins - 1);  // fullest first
      rank = atomicAdd(&bin_count[bin], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      u32 run = 0;
      for (u32 q = 0; q < kBins; q++) {
        bin_start[q] = run;
        run += bin_count[q];
      }
    }
    __syncthreads();
    if (sub == 0) order[bin_start[bin] + rank] = grp;
    __syncthreads();
    b = (size_t)(blockIdx.x - kTreeHeavyBlocks) * (blockDim.x / (u32)lanes) + order[grp];
    live = b < nbt;
    s = e = 0;
    if (live) {
      s = start[b];
      e = end[b];
    }
  }
  const bool is_heavy = e - s > heavy_from;
  if (is_heavy) {
    if (sub == 0 && e - s > heavy_max) heavy[atomicAdd(heavy_count, 1u)] = (u32)b;
    e = s;
  }
  Xyzz acc = xyzz_identity<F>();
  // the entry word of the next step is fetched while this step's point is added: one memory latency per step
  // (the record's) instead of two in a row (2^16 points: 0.287 -> 0.270 ms device phase)
  u32 v_next = s + sub < e ? vals[s + sub] : 0u;
  for (u32 p = s + sub; p < e; p += (u32)lanes) {
    const u32 v = v_next;
    if (p + (u32)lanes < e) v_next = vals[p + (u32)lanes];
    size_t idx = v & idx_mask;
    Fe qx = fe_load(packed + idx * 16);
    Fe qy = fe_load(packed + idx * 16 + 8);
    if (v >> 31) qy = fe_neg<F>(qy);
    xyzz_madd<F>(acc, qx, qy);
  }
  if constexpr (!QT) {
    for (int off = lanes >> 1; off > 0; off >>= 1) {
      Xyzz o = xyzz_shfl_down(acc, off, lanes);
      if ((int)sub < off) xyzz_add<F>(acc, o);
    }
    if (live && !is_heavy && sub == 0) xyzz_store_soa(buckets, nbt, b, acc);
  } else {
    __shared__ u32 pts[256 * 32];
    __shared__ u32 scr_all[64 * kQuadScratch * 8];
    __shared__ u32 zero[8];
    __shared__ u32 keep[256];  // per lane group: its bucket is written by this block
    const u32 tid = threadIdx.x, quad = tid >> 2;
    const int role = (int)(tid & 3u), coord = quad_coord(role);
    u32* scr = scr_all + quad * (kQuadScratch * 8);
    if (tid < 8) zero[tid] = 0;
    xyzz_store_aos(pts + tid * 32, acc);
    if (sub == 0) keep[tid / (u32)lanes] = live && !is_heavy ? 1u : 0u;
    __syncthreads();
    const u32 groups = 256u / (u32)lanes;
    for (u32 half = (u32)lanes >> 1; half > 0; half >>= 1) {
      const u32 adds = groups * half;
      for (u32 task = quad; task < adds; task += 64) {
        const u32 dst = (task / half) * (u32)lanes + task % half;
        Fe out;
        const bool changed = quad_add<F>(pts, scr, zero, dst, dst + half, role, out);
        quad_sync();
        if (changed) fe_store(pts + dst * 32 + coord * 8, out);
        quad_sync();
      }
      __syncthreads();
    }
    const size_t b0 = (size_t)(blockIdx.x - kTreeHeavyBlocks) * blockDim.x / (size_t)lanes;
    for (u32 i = tid; i < groups * 4; i += blockDim.x) {
      const u32 grp = i >> 2, c = i & 3;
      if (keep[grp]) fe_store(buckets + ((size_t)c * nbt + b0 + grp) * 8, fe_load(pts + grp * (u32)lanes * 32 + c * 8));
    }
  }
}

// k_bucket_tree with every addition spread over four lanes (bp_quad.h): 2^log_q quads per bucket,
// each adds its share of the bucket's entries (strided) into a point it keeps in LDS, then a tree
// over the quads of the bucket.  What counts in the launches this serves (a few thousand entries
// per window: the rounds of an inner-product argument) is the length of the dependent chain, and
// a quad addition takes about half the time of a one-lane addition.  64 quads per block.
constexpr int kBucketQuadThreads = 256;

template <class C>
__global__ __launch_bounds__(kBucketQuadThreads) void k_bucket_quad(
    const u32* __restrict__ packed, const u32* __restrict__ vals, const u32* __restrict__ start,
    const u32* __restrict__ end, size_t nbt, int log_q, u32 heavy_from, u32 heavy_max, u32 idx_mask,
    u32* __restrict__ buckets, u32* __restrict__ heavy, u32* __restrict__ heavy_count) {
  using F = typename C::Fq;
  if (blockIdx.x < kTreeHeavyBlocks) {
    tree_heavy_block<C>(packed, vals, start, end, nbt, heavy_from, heavy_max, idx_mask, buckets, heavy, heavy_count);
    return;
  }
  constexpr int kQuads = kBucketQuadThreads / 4;
  __shared__ u32 pts[kQuads * 2 * 32];  // per quad: its running sum, the entry being added
  __shared__ u32 scr_all[kQuads * kQuadScratch * 8];
  __shared__ u32 zero[8];
  const u32 tid = threadIdx.x, quad = tid >> 2;
  const int role = (int)(tid & 3u), coord = quad_coord(role);
  u32* scr = scr_all + quad * (kQuadScratch * 8);
  u32* acc = pts + quad * 64;
  if (tid < 8) zero[tid] = 0;
  const Fe one = fe_one<F>();
  const u32 per_block = (u32)kQuads >> log_q;
  const size_t b = (size_t)(blockIdx.x - kTreeHeavyBlocks) * per_block + (quad >> log_q);
  const u32 sub = quad & ((1u << log_q) - 1u), n_sub = 1u << log_q;
  const bool live = b < nbt;
  u32 s = 0, e = 0;
  if (live) {
    s = start[b];
    e = end[b];
  }
  const bool is_heavy = e - s > heavy_from;
  if (is_heavy) {
    if (sub == 0 && role == 0 && e - s > heavy_max) heavy[atomicAdd(heavy_count, 1u)] = (u32)b;
    e = s;
  }
  fe_store(acc + role * 8, quad_identity_coord(role, one));
  __syncthreads();  // zero[] and the identities
  for (u32 p = s + sub; p < e; p += n_sub) {
    const u32 v = vals[p];
    const size_t idx = v & idx_mask;
    // the entry as an XYZZ point behind the sum: x, y (negated for a negative digit), 1, 1
    Fe val = one;
    if (role < 2) val = fe_load(packed + idx * 16 + role * 8);
    if (role == 1 && (v >> 31)) val = fe_neg<F>(val);
    fe_store(acc + 32 + role * 8, val);
    quad_sync();
    Fe out;
    const bool changed = quad_add<F>(pts, scr, zero, quad * 2, quad * 2 + 1, role, out);
    quad_sync();
    if (changed) fe_store(acc + coord * 8, out);
    quad_sync();
  }
  for (int l = 0; l < log_q; l++) {
    __syncthreads();
    if ((sub & ((2u << l) - 1u)) == 0) {
      Fe out;
      const bool changed = quad_add<F>(pts, scr, zero, quad * 2, (quad + (1u << l)) * 2, role, out);
      quad_sync();
      if (changed) fe_store(acc + coord * 8, out);
      quad_sync();
    }
  }
  if (live && !is_heavy && sub == 0)
    fe_store(buckets + ((size_t)role * nbt + b) * 8, fe_load(acc + role * 8));
}

// one 1024-lane block per queued bucket: strided mixed additions, shuffle tree per wave,
// then the 16 wave sums by the first wave
template <class C>
__global__ __launch_bounds__(1024) void k_bucket_heavy_entries(const u32* __restrict__ packed,
                                                               const u32* __restrict__ vals,
                                                               const u32* __restrict__ start,
                                                               const u32* __restrict__ end, size_t nbt,
                                                               u32 idx_mask, u32* __restrict__ buckets,
                                                               const u32* __restrict__ heavy,
                                                               const u32* __restrict__ heavy_count) {
  using F = typename C::Fq;
  __shared__ u32 lds[16 * 32];
  // heavy == null: every bucket, in order
  const u32 count = heavy ? *heavy_count : (u32)nbt;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (u32 q = blockIdx.x; q < count; q += gridDim.x) {
    const u32 b = heavy ? heavy[q] : q;
    const u32 s = start[b], e = end[b];
    Xyzz acc = xyzz_identity<F>();
    for (u32 p = s + threadIdx.x; p < e; p += blockDim.x) {
      u32 v = vals[p];
      size_t idx = v & idx_mask;
      Fe qx = fe_load(packed + idx * 16);
      Fe qy = fe_load(packed + idx * 16 + 8);
      if (v >> 31) qy = fe_neg<F>(qy);
      xyzz_madd<F>(acc, qx, qy);
    }
    for (int off = 32; off > 0; off >>= 1) {
      Xyzz o = xyzz_shfl_down(acc, off, 64);
      if (lane < off) xyzz_add<F>(acc, o);
    }
    if (lane == 0) xyzz_store_aos(lds + wave * 32, acc);
    __syncthreads();
    if (wave == 0) {
      acc = lane < 16 ? xyzz_load_aos(lds + lane * 32) : xyzz_identity<F>();
      for (int off = 8; off > 0; off >>= 1) {
        Xyzz o = xyzz_shfl_down(acc, off, 64);
        if (lane < off) xyzz_add<F>(acc, o);
      }
      if (lane == 0) xyzz_store_soa(buckets, nbt, b, acc);
    }
    __syncthreads();
  }
}

// Joins the chunk sums of every bucket with `lanes` lanes per bucket (strided serial
// additions, then a shuffle tree): for plans whose buckets all hold many chunks (narrow
// windows on mid-sized inputs), where one lane per bucket would be a long chain and one
// block per bucket a waste.
template <class C>
__global__ __launch_bounds__(256) void k_bucket_finish_tree(const u32* __restrict__ nch,
                                                            const u32* __restrict__ choff, size_t nbt,
                                                            size_t nc_cap, const u32* __restrict__ chunk_sums,
                                                            int lanes, u32* __restrict__ buckets) {
  using F = typename C::Fq;
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t b = t / (size_t)lanes;
  const u32 sub = (u32)(t % (size_t)lanes);
  const bool live = b < nbt;
  u32 n = 0, off = 0;
  if (live) {
    n = nch[b];
    off = choff[b];
  }
  const bool direct = n == 1;  // k_chunk_sums wrote the bucket itself
  if (direct) n = 0;
  Xyzz acc = xyzz_identity<F>();
  for (u32 k = sub; k < n; k += (u32)lanes) {
    Xyzz v = xyzz_load_soa(chunk_sums, nc_cap, off + k);
    xyzz_add<F>(acc, v);
  }
  for (int o = lanes >> 1; o > 0; o >>= 1) {
    Xyzz other = xyzz_shfl_down(acc, o, lanes);
    if ((int)sub < o) xyzz_add<F>(acc, other);
  }
  if (live && !direct && sub == 0) xyzz_store_soa(buckets, nbt, b, acc);
}

// ---- 5'. window value for 64..256 buckets in one block -------------------------------------
// V = sum_{m=1}^{NB} m * B_{m-1}.  Slot m of an LDS array holds B_{m-1} (slot 0 the
// identity; m = NB is kept aside).  A binary sum tree is built in place: after step t the
// slots whose index is a multiple of 2^t hold the sums of 2^t consecutive slots.  The slots
// with exactly b trailing zero bits are the odd blocks of level b, and their sum S_b is
// the sum of all B whose weight m has bit b set; each S_b is reduced in place by a tree of
// its own that starts as soon as level b exists, so all of them are complete after
// log2(NB) - 1 steps of one addition each.  Then V = sum_b 2^b S_b + NB * B_{NB-1}.
// One block; buckets base .. base + nb - 1 of an SoA array with plane stride nbt, nb a power
// of two in 2..256; the value, doubled post_dbl times, goes to out (XYZZ, 32 words).
template <class C, bool EXPORT>
BP_DEV void window_reduce_block(const u32* __restrict__ buckets, size_t nbt, size_t base, u32 nb, int log_nb,
                                int post_dbl, u32* __restrict__ out) {
  using F = typename C::Fq;
  __shared__ u32 slot[256 * 32];
  __shared__ u32 term[16 * 32];
  const u32 tid = threadIdx.x;
  if (tid < nb) {
    Xyzz v = tid == 0 ? xyzz_identity<F>() : xyzz_load_soa(buckets, nbt, base + tid - 1);
    xyzz_store_aos(slot + tid * 32, v);
  }
  __syncthreads();
  for (int t = 1; t <= log_nb - 1; t++) {
    // tasks of this step: the sum-tree level (nb >> t additions), then for every level
    // b < t one step of the S_b tree (nb >> (t+1) additions each)
    const u32 chain = nb >> t, per = nb >> (t + 1);
    u32 dst = 0, src = 0;
    bool work = false;
    if (tid < chain) {
      dst = tid << t;
      src = dst + (1u << (t - 1));
      work = true;
    } else {
      u32 q = tid - chain;
      u32 b = q / per, j = q % per;
      if (b < (u32)t) {
        const int r = t - (int)b;          // step number inside the S_b tree
        u32 k = j << r;                    // k = 0 mod 2^r
        dst = (2 * k + 1) << b;
        src = (2 * (k + (1u << (r - 1))) + 1) << b;
        work = true;
      }
    }
    Xyzz a, o;
    if (work) {
      a = xyzz_load_aos(slot + dst * 32);
      o = xyzz_load_aos(slot + src * 32);
    }
    __syncthreads();
    if (work) {
      xyzz_add<F>(a, o);
      xyzz_store_aos(slot + dst * 32, a);
    }
    __syncthreads();
  }
  // terms: T_b = S_b = slot[2^b] for b < log_nb, T_{log_nb} = B_{NB-1}
  const int n_terms = log_nb + 1;
  if ((int)tid < n_terms) {
    Xyzz v = (int)tid < log_nb ? xyzz_load_aos(slot + ((size_t)1 << tid) * 32)
                               : xyzz_load_soa(buckets, nbt, base + nb - 1);
    xyzz_store_aos(term + tid * 32, v);
  }
  __syncthreads();
  // term b carries the weight 2^(b + post_dbl): every term is doubled by a lane of its own, all
  // at once (log_nb + post_dbl doublings on the longest path instead of 1 + 2 + 4 + ... one
  // level after the other), and a tree of plain additions joins them
  if ((int)tid < n_terms) {
    Xyzz v = xyzz_load_aos(term + tid * 32);
    for (int d = 0; d < (int)tid + post_dbl; d++) v = xyzz_dbl<F>(v);
    xyzz_store_aos(term + tid * 32, v);
  }
  __syncthreads();
  for (int sft = 1; sft < n_terms; sft <<= 1) {
    const bool work = (int)tid < n_terms && (tid % (2 * sft)) == 0 && (int)tid + sft < n_terms;
    Xyzz a, o;
    if (work) {
      a = xyzz_load_aos(term + tid * 32);
      o = xyzz_load_aos(term + (tid + sft) * 32);
      xyzz_add<F>(a, o);
    }
    __syncthreads();
    if (work) xyzz_store_aos(term + tid * 32, a);
    __syncthreads();
  }
  if (tid == 0) {
    Xyzz v = xyzz_load_aos(term);
    if constexpr (EXPORT) v = xyzz_export<C>(v);
    xyzz_store_aos(out, v);
  }
}

template <class C>
__global__ __launch_bounds__(256) void k_window_reduce_small(const u32* __restrict__ buckets, size_t nbt,
                                                             u32 n_buckets, int log_nb,
                                                             u32* __restrict__ windows) {
  window_reduce_block<C, true>(buckets, nbt, (size_t)blockIdx.x * n_buckets, n_buckets, log_nb, 0,
                               windows + (size_t)blockIdx.x * 32);
}

// ---- 5''. bucket reduction by row and column sums -----------------------------------------
// Window value = sum_j (j+1)*B_j over NB = R*256 buckets.  Written as a matrix B[r][q] with
// j = 256 r + q the weight is 256 r + (q+1), so the value is
//   256 * sum_r r*Rsum_r + sum_q (q+1)*Csum_q,   Rsum_r = sum_q B[r][q],  Csum_q = sum_r B[r][q].
// Every bucket is added twice, as in a running sum, but no partial sum has to be multiplied by
// the offset of its segment (that multiplication was half of the lane-work of
// k_reduce_segments), and the two weighted sums that remain have 256 and R terms per window,
// which is the size window_reduce_block handles.  Lanes of the first row_blocks blocks each add
// a strip of the buckets of a row (row_lanes lanes per row, lane i takes q = i mod row_lanes:
// coalesced) and join by a shuffle tree; lanes of the other blocks do the same down a column (col_lanes lanes
// per column).  Row sum r goes to slot r-1 of the window's row array (row 0 has weight 0; the
// last slot gets the identity), column sum q to slot q of its column array.
// Joins the partial sums that groups of `lanes` adjacent lanes of a 256-lane block hold (lanes: a
// power of two): on return lane 0 of every group has the group's sum.  A shuffle tree would keep
// every wave busy for every level while half, a quarter, ... of its lanes have an addition to do;
// here the partial sums go through LDS and the additions of a level are dealt to the first lanes
// of the block, so a level costs as many waves as it has additions.  A level's additions read
// slots that no addition of the same level writes, so one barrier per level is enough.  Slots are
// 36 words apart: 16-byte accesses of neighbouring lanes then fall into different banks.
constexpr u32 kTreeSlot = 36;
template <class F>
BP_DEV void group_tree_lds(Xyzz& acc, int lanes, u32* lds) {
  if (lanes <= 1) return;
  const int tid = (int)threadIdx.x;
  xyzz_store_aos(lds + tid * kTreeSlot, acc);
  __syncthreads();
  const int groups = 256 / lanes;
  for (int half = lanes >> 1; half > 0; half >>= 1) {
    if (tid < groups * half) {
      const int dst = (tid / half) * lanes + tid % half;
      Xyzz a = xyzz_load_aos(lds + dst * kTreeSlot);
      Xyzz o = xyzz_load_aos(lds + (dst + half) * kTreeSlot);
      xyzz_add<F>(a, o);
      xyzz_store_aos(lds + dst * kTreeSlot, a);
    }
    __syncthreads();
  }
  if (tid % lanes == 0) acc = xyzz_load_aos(lds + tid * kTreeSlot);
}

template <class C>
__global__ __launch_bounds__(256) void k_rowcol_sums(const u32* __restrict__ buckets, size_t nbt, u32 n_buckets,
                                                     int n_windows, u32 row_blocks, int row_lanes, int col_lanes,
                                                     u32* __restrict__ cols, u32* __restrict__ rows,
                                                     u32* __restrict__ done) {
  using F = typename C::Fq;
  __shared__ __attribute__((aligned(16))) u32 lds[256 * kTreeSlot];
  const u32 n_rows = n_buckets / kRowcolCols;
  // arrival counters of the launch that follows (k_rowcol_final_quad joins a window's two parts)
  if (done != nullptr && blockIdx.x == 0 && (int)threadIdx.x < n_windows) done[threadIdx.x] = 0;
  Xyzz acc = xyzz_identity<F>();
  if (blockIdx.x < row_blocks) {
    const u32 kLanes = (u32)row_lanes;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t row_id = t / kLanes;
    const u32 i = (u32)(t % kLanes);
    const bool live = row_id < (size_t)n_windows * n_rows;
    const u32 w = (u32)(row_id / n_rows), r = (u32)(row_id % n_rows);
    if (live) {
      const size_t base = (size_t)w * n_buckets + (size_t)r * kRowcolCols + i;
      for (u32 q = 0; q < kRowcolCols; q += kLanes) {
        Xyzz bk = xyzz_load_soa(buckets, nbt, base + q);
        xyzz_add<F>(acc, bk);
      }
    }
    group_tree_lds<F>(acc, row_lanes, lds);
    if (live && i == 0) {
      if (r == 0) acc = xyzz_identity<F>();
      xyzz_store_soa(rows, (size_t)n_windows * n_rows, (size_t)w * n_rows + (r ? r - 1 : n_rows - 1), acc);
    }
  } else {
    const size_t t = (size_t)(blockIdx.x - row_blocks) * blockDim.x + threadIdx.x;
    const size_t col_id = t / (u32)col_lanes;
    const u32 i = (u32)(t % (u32)col_lanes);
    const bool live = col_id < (size_t)n_windows * kRowcolCols;
    const u32 w = (u32)(col_id / kRowcolCols), q = (u32)(col_id % kRowcolCols);
    if (live) {
      const size_t base = (size_t)w * n_buckets + q;
      for (u32 r = i; r < n_rows; r += (u32)col_lanes) {
        Xyzz bk = xyzz_load_soa(buckets, nbt, base + (size_t)r * kRowcolCols);
        xyzz_add<F>(acc, bk);
      }
    }
    group_tree_lds<F>(acc, col_lanes, lds);
    if (live && i == 0) xyzz_store_soa(cols, (size_t)n_windows * kRowcolCols, (size_t)w * kRowcolCols + q, acc);
  }
}

// Blocks 0 .. n_windows-1: parts[w] = sum_q (q+1)*Csum_q; blocks n_windows .. 2 n_windows-1:
// parts[n_windows + w] = 256 * sum_r r*Rsum_r (the row tree is one level shorter, so its eight
// doublings finish while the column blocks are still at work).  Window value = sum of the two.
template <class C>
__global__ __launch_bounds__(256) void k_rowcol_final(const u32* __restrict__ cols, const u32* __restrict__ rows,
                                                      int n_windows, u32 n_rows, int log_rows,
                                                      u32* __restrict__ parts) {
  const bool is_rows = blockIdx.x >= (u32)n_windows;
  const u32 w = is_rows ? blockIdx.x - (u32)n_windows : blockIdx.x;
  const u32 nb = is_rows ? n_rows : kRowcolCols;
  window_reduce_block<C, false>(is_rows ? rows : cols, (size_t)n_windows * nb, (size_t)w * nb, nb, is_rows ? log_rows : 8,
                         is_rows ? 8 : 0, parts + (size_t)blockIdx.x * 32);
}

// ---- 5. bucket reduction ----------------------------------------------------------
// Window value = sum_{j=0}^{NB-1} (j+1)*B_j.  A thread takes seg_len consecutive
// buckets: T = sum (j-j0+1)*B_j by a running sum, S = sum B_j, and emits
// T + j0*S (j0*S by double-and-add on a < 2^19 multiplier).
template <class C>
__global__ __launch_bounds__(256) void k_reduce_segments(const u32* __restrict__ buckets, size_t nbt,
                                                         u32 n_buckets, int seg_len, u32 n_segs,
                                                         int n_windows, u32* __restrict__ segs) {
  using F = typename C::Fq;
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)n_windows * n_segs) return;
  u32 w = (u32)(t / n_segs), sg = (u32)(t % n_segs);
  u32 j0 = sg * (u32)seg_len;
  size_t base = (size_t)w * n_buckets + j0;
  Xyzz run = xyzz_identity<F>(), acc = xyzz_identity<F>();
  for (int j = seg_len - 1; j >= 0; j--) {
    Xyzz bk = xyzz_load_soa(buckets, nbt, base + j);
    xyzz_add<F>(run, bk);
    xyzz_add<F>(acc, run);
  }
  if (j0 != 0 && !xyzz_is_identity(run)) {
    // j0 * run, right to left over the non-adjacent form of j0 (digits 0, +1, -1; at most
    // every other digit is non-zero, which matters because a wave pays for an addition
    // whenever any of its lanes has one)
    Xyzz m = xyzz_identity<F>(), pw = run;
    u32 k = j0;
    while (k) {
      if (k & 1) {
        Xyzz term = pw;
        if (k & 2) {
          term.y = fe_neg<F>(term.y);
          k += 1;
        } else {
          k -= 1;
        }
        xyzz_add<F>(m, term);
      }
      k >>= 1;
      if (k) pw = xyzz_dbl<F>(pw);
    }
    xyzz_add<F>(acc, m);
  }
  xyzz_store_aos(segs + t * 32, acc);
}

// Block b sums `count` consecutive XYZZ points (AoS) starting at in + b*count:
// every lane first adds its strided share serially, then a tree over the lanes that
// hold data (depth = ceil(count/256) - 1 + log2(min(count, 256)) additions).
template <class C, bool EXPORT>
__global__ __launch_bounds__(256) void k_group_sum(const u32* __restrict__ in, u32 count,
                                                   u32* __restrict__ out) {
  using F = typename C::Fq;
  __shared__ u32 lds[256 * 32];
  const u32* base = in + (size_t)blockIdx.x * count * 32;
  Xyzz acc = xyzz_identity<F>();
  for (u32 s = threadIdx.x; s < count; s += blockDim.x) {
    Xyzz v = xyzz_load_aos(base + (size_t)s * 32);
    xyzz_add<F>(acc, v);
  }
  xyzz_store_aos(lds + threadIdx.x * 32, acc);
  __syncthreads();
  int top = 128;
  while (top >= 2 && (u32)top >= count) top >>= 1;  // lanes >= count hold the identity
  for (int off = top; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      Xyzz o = xyzz_load_aos(lds + (threadIdx.x + off) * 32);
      xyzz_add<F>(acc, o);
      xyzz_store_aos(lds + threadIdx.x * 32, acc);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if constexpr (EXPORT) acc = xyzz_export<C>(acc);
    xyzz_store_aos(out + (size_t)blockIdx.x * 32, acc);
  }
}

// window-wise sum over parts (multi-GPU combine): out[w] = sum_p in[p][w]
template <class C, bool EXPORT>
__global__ __launch_bounds__(64) void k_window_combine(const u32* __restrict__ in, int n_parts,
                                                       int n_windows, u32* __restrict__ out) {
  using F = typename C::Fq;
  int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_windows) return;
  Xyzz acc = xyzz_identity<F>();
  for (int p = 0; p < n_parts; p++) {
    Xyzz v = xyzz_load_aos(in + ((size_t)p * n_windows + w) * 32);
    xyzz_add<F>(acc, v);
  }
  if constexpr (EXPORT) acc = xyzz_export<C>(acc);
  xyzz_store_aos(out + (size_t)w * 32, acc);
}

// Window sums that have left the pipeline (several GPUs' partial results) are in the boundary
// form, so this sum runs on the curve types of the boundary.
hipError_t launch_window_combine(int curve, const u32* windows, int n_parts, int n_windows, u32* out,
                                 hipStream_t st) {
  const dim3 grid((n_windows + 63) / 64), block(64);
  if (curve == BP_SECP256K1)
    hipLaunchKernelGGL((k_window_combine<Secp256k1, false>), grid, block, 0, st, windows, n_parts, n_windows, out);
  else if (curve == BP_VESTA)
    hipLaunchKernelGGL((k_window_combine<Vesta, false>), grid, block, 0, st, windows, n_parts, n_windows, out);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

// Window values of plans with many windows (batched plans): `lpw` lanes per window, where the kernel of
// bp_reduce.hip puts a block of quads on every window to make ONE window's chain short.  Thousands of
// windows fill the machine with lanes, so what counts is the additions a lane runs in a row:
//   lane j takes buckets [j cnt, (j + 1) cnt) by running sums: run_j = their sum, acc_j = sum_k (k + 1) B_{j cnt + k};
//   the window value is sum_j acc_j + cnt * sum_{t >= 1} Suffix_t with Suffix_t = sum_{j >= t} run_j:
//   a scan over the lanes makes the suffixes (log2 lpw additions), log2 cnt doublings and one addition
//   fold them into acc, and a tree adds the acc (log2 lpw additions).
// Every lane keeps its two points in LDS and ONE inlined addition serves every step (operands are chosen by
// address; a doubling is the addition of a point to itself): with an addition per step the body was 100 KB
// of code against a 64 KiB instruction cache, and a lane waited for instructions more than for results.

// point `which` (0 run, 1 acc) of lane `lane` of a block of 256: word k at (which * 32 + k) * 256 + lane, so
// the lanes of a wave touch different banks
BP_DEV Xyzz reduce_lds_load(const u32* pts, u32 which, u32 lane) {
  Xyzz p;
  const u32* b = pts + which * 32 * 256 + lane;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    p.x.v[k] = b[k * 256];
    p.y.v[k] = b[(8 + k) * 256];
    p.zz.v[k] = b[(16 + k) * 256];
    p.zzz.v[k] = b[(24 + k) * 256];
  }
  return p;
}
BP_DEV void reduce_lds_store(u32* pts, u32 which, u32 lane, const Xyzz& p) {
  u32* b = pts + which * 32 * 256 + lane;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    b[k * 256] = p.x.v[k];
    b[(8 + k) * 256] = p.y.v[k];
    b[(16 + k) * 256] = p.zz.v[k];
    b[(24 + k) * 256] = p.zzz.v[k];
  }
}

template <class C>
__global__ __launch_bounds__(256) void k_window_reduce_lanes(const u32* __restrict__ buckets, size_t nbt, u32 n_buckets,
                                                             int log_lpw, int log_cnt, u32 n_windows,
                                                             u32* __restrict__ windows) {
  using F = typename C::Fq;
  __shared__ u32 pts[2 * 32 * 256];  // per lane: run, acc
  const u32 tid = threadIdx.x, lpw = 1u << log_lpw, cnt = 1u << log_cnt;
  const u32 t = blockIdx.x * blockDim.x + tid;
  const u32 w = t >> log_lpw, j = t & (lpw - 1);
  const bool live = w < n_windows;  // whole lane groups (parts of one wave) are live or not
  constexpr u32 RUN = 0, ACC = 1, NONE = 2;
  {
    const Xyzz id = xyzz_identity<F>();
    reduce_lds_store(pts, RUN, tid, id);
    reduce_lds_store(pts, ACC, tid, id);
  }
  quad_sync();
  const size_t base = (size_t)w * n_buckets + (size_t)j * cnt;
  const int local_steps = 2 * (int)cnt, scan_end = local_steps + log_lpw, dbl_end = scan_end + log_cnt;
  const int total = dbl_end + 1 + log_lpw;
#pragma unroll 1
  for (int step = 0; step < total; step++) {
    // dst += src: points named by (which, lane); src NONE with from_bucket: a bucket sum from memory
    u32 dst = NONE, src = NONE, src_lane = tid;
    bool from_bucket = false;
    size_t bucket = 0;
    if (step < local_steps) {
      if ((step & 1) == 0) {
        dst = RUN;
        from_bucket = true;
        bucket = base + (size_t)(cnt - 1 - (u32)(step >> 1));
      } else {
        dst = ACC;
        src = RUN;
      }
    } else if (step < scan_end) {
      const u32 d = 1u << (step - local_steps);
      if (j + d < lpw) {
        dst = RUN;
        src = RUN;
        src_lane = tid + d;
      }
    } else if (step < dbl_end) {
      if (j >= 1) dst = src = RUN;
    } else if (step == dbl_end) {
      if (j >= 1) {
        dst = ACC;
        src = RUN;
      }
    } else {
      const u32 d = lpw >> (step - dbl_end);
      if (j < d) {
        dst = src = ACC;
        src_lane = tid + d;
      }
    }
    if (live && dst != NONE) {
      Xyzz p = reduce_lds_load(pts, dst, tid);
      const Xyzz q = from_bucket ? xyzz_load_soa(buckets, nbt, bucket) : reduce_lds_load(pts, src, src_lane);
      xyzz_add<F>(p, q);
      quad_sync();  // every lane of the wave has read its operands
      reduce_lds_store(pts, dst, tid, p);
    }
    quad_sync();
  }
  if (live && j == 0) xyzz_store_aos(windows + (size_t)w * 32, xyzz_export<C>(reduce_lds_load(pts, ACC, tid)));
}

// One block of 32 quads per sum: the count <= 64 points sit in LDS and a tree joins them, every addition spread
// over four lanes (bp_quad.h).  The tree is a chain of six dependent additions on lanes that have the CU to
// themselves, so what counts is the time of one addition: a wave with a lane per point and a shuffle tree
// took 60 us where this takes about 20.
constexpr int kGroupSumThreads = 128;

template <class C>
__global__ __launch_bounds__(kGroupSumThreads) void k_window_group_sums(const u32* __restrict__ in, int count,
                                                                        u32* __restrict__ out) {
  using F = typename C::Fq;
  __shared__ u32 pts[64 * 32];
  __shared__ u32 scr_all[(kGroupSumThreads / 4) * kQuadScratch * 8];
  __shared__ u32 zero[8];
  const u32 tid = threadIdx.x, quad = tid >> 2;
  const int role = (int)(tid & 3u), coord = quad_coord(role);
  u32* scr = scr_all + quad * (kQuadScratch * 8);
  if (tid < 8) zero[tid] = 0;
  const Fe one = fe_one<F>();
  for (u32 i = tid; i < 64 * 4; i += kGroupSumThreads) {
    const u32 slot = i >> 2, c = i & 3;
    const Fe v = (int)slot < count ? fe_load(in + ((size_t)blockIdx.x * count + slot) * 32 + c * 8) : quad_identity_coord((int)c, one);
    fe_store(pts + slot * 32 + c * 8, v);
  }
  __syncthreads();
  for (u32 half = 32; half > 0; half >>= 1) {
    if (quad < half) {
      Fe v;
      const bool changed = quad_add<F>(pts, scr, zero, quad, quad + half, role, v);
      quad_sync();
      if (changed) fe_store(pts + quad * 32 + coord * 8, v);
    }
    __syncthreads();
  }
  if (quad == 0) fe_store(out + (size_t)blockIdx.x * 32 + coord * 8, fe_load(pts + coord * 8));
}

hipError_t launch_window_group_sums(int curve, const u32* in, int count, int n_sums, u32* out, hipStream_t st) {
  if (count < 1 || count > 64 || n_sums < 1) return hipErrorInvalidValue;
  if (curve == BP_SECP256K1)
    hipLaunchKernelGGL(k_window_group_sums<Secp256k1>, dim3((unsigned)n_sums), dim3(kGroupSumThreads), 0, st, in, count, out);
  else if (curve == BP_VESTA)
    hipLaunchKernelGGL(k_window_group_sums<Vesta>, dim3((unsigned)n_sums), dim3(kGroupSumThreads), 0, st, in, count, out);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

static hipError_t msm_run_tail(int curve, const MsmPlan& plan, const MsmWorkspace& ws, hipStream_t st,
                               hipEvent_t ev_acc0, hipEvent_t ev_acc1, const u32* table_rec = nullptr);
static hipError_t msm_sort_and_chunk(const MsmPlan& plan, const MsmWorkspace& ws, hipStream_t st, u32 window_base);
static hipError_t msm_chunk_table(const MsmPlan& plan, const MsmWorkspace& ws, hipStream_t st);
static hipError_t msm_accumulate(int curve, const MsmPlan& plan, const MsmWorkspace& ws, hipStream_t st,
                                 hipEvent_t ev_acc0, hipEvent_t ev_acc1, const u32* table_rec);
static hipError_t msm_reduce(int curve, const MsmPlan& plan, const MsmWorkspace& ws, hipStream_t st);

hipError_t msm_run(int curve, const MsmPart& p0, const MsmPart& p1, const MsmExtra* extra, const MsmPlan& plan,
                   const MsmWorkspace& ws, hipStream_t st, hipEvent_t ev_acc0, hipEvent_t ev_acc1,
                   const MsmTable* table, const MsmPipe* pipe) {
  const size_t n = plan.n_total;
  hipError_t e;
  if (n == 0) {
    // all windows = identity (zz = 0)
    return hipMemsetAsync(ws.windows, 0, plan.win_bytes, st);
  }
  DigitGroups sel{(u32)p0.sel_half, (u32)p1.sel_half, p0.group_lo, p0.group_hi, p1.group_lo, p1.group_hi, 0,
                  {nullptr, nullptr}, {nullptr, nullptr}, {0, 0}};
  if (plan.groups == 1) sel = DigitGroups{0, 0, 0, 0, 0, 0, 0, {nullptr, nullptr}, {nullptr, nullptr}, {0, 0}};
  if (extra) {
    // one extra point per group; parts that go whole to one group only with the one-launch front end
    // (the other front ends place an extra point behind the halves of split parts)
    if (plan.groups != 2 || n != p0.n + p1.n + 2) return hipErrorInvalidValue;
    if (!plan.small && (!p0.sel_half || (p1.n && !p1.sel_half))) return hipErrorInvalidValue;
    sel.n_extra = 2;
    for (int e = 0; e < 2; e++) {
      sel.extra_xy[e] = extra->xy[e];
      sel.extra_s[e] = extra->s[e];
      sel.extra_tab[e] = (u32)extra->tab_index[e];
      sel.extra_stride[e] = (u32)extra->batch_stride[e];
    }
  } else if (n != p0.n + p1.n) {
    return hipErrorInvalidValue;
  }
  // split parts must tile evenly, and split and whole parts cannot be mixed
  if (plan.groups == 2) {
    if ((p0.sel_half != 0) != (p1.sel_half != 0) && p0.n && p1.n) return hipErrorInvalidValue;
    if (p0.sel_half && p0.n % (2 * p0.sel_half)) return hipErrorInvalidValue;
    if (p1.sel_half && p1.n % (2 * p1.sel_half)) return hipErrorInvalidValue;
    if (!p0.sel_half && !p1.sel_half && p0.n && p1.n && p0.group_lo > p1.group_lo) return hipErrorInvalidValue;
  }
  if (plan.tab && (!table || !table->rec)) return hipErrorInvalidValue;
  if (plan.batch > 1 && !plan.small) return hipErrorInvalidValue;
  if (plan.small) {
    sel.s_stride0 = (u32)p0.batch_stride;
    sel.s_stride1 = (u32)p1.batch_stride;
    static const bool no_tab_front = getenv("BP_MSM_NO_TAB_FRONT") != nullptr;
    const int front_windows = plan.c == 8 ? TabFront<8>::kWindows : TabFront<4>::kWindows;
    if (plan.batch > 1 && plan.tab_virtual && (plan.c == 8 || plan.c == 4) && plan.group_windows % front_windows == 0 && !no_tab_front) {
      // a batch has blocks enough for each to take several digit windows and convert a scalar once for them
      const unsigned parts = (unsigned)(plan.group_windows / front_windows);
      sel.tab_sess_stride = (u32)table->sess_stride;
      DISPATCH_CURVE(curve, {
        if (plan.c == 8)
          hipLaunchKernelGGL((k_tab_front<C, 8>), dim3(parts * (unsigned)plan.groups, (unsigned)plan.batch), dim3(kSmallThreads), 0,
                             st, p0.s, p0.inf, p0.n, p1.s, p1.inf, p1.n, sel, plan.group_windows, plan.wshift, ws.vals_b,
                             ws.bucket_start, ws.bucket_end, ws.heavy_count, (u32)table->stride, (u32)p0.tab_offset,
                             (u32)p1.tab_offset);
        else
          hipLaunchKernelGGL((k_tab_front<C, 4>), dim3(parts * (unsigned)plan.groups, (unsigned)plan.batch), dim3(kSmallThreads), 0,
                             st, p0.s, p0.inf, p0.n, p1.s, p1.inf, p1.n, sel, plan.group_windows, plan.wshift, ws.vals_b,
                             ws.bucket_start, ws.bucket_end, ws.heavy_count, (u32)table->stride, (u32)p0.tab_offset,
                             (u32)p1.tab_offset);
      });
      if ((e = hipGetLastError()) != hipSuccess) return e;
      return msm_run_tail(curve, plan, ws, st, ev_acc0, ev_acc1, table->rec);
    }
    // sorted entries and bucket bounds straight into the buffers the accumulation reads
    if (plan.tab) sel.tab_sess_stride = (u32)table->sess_stride;
    DISPATCH_CURVE(curve, hipLaunchKernelGGL(k_small_front<C>, dim3((unsigned)(plan.n_windows / plan.batch), (unsigned)plan.batch),
                                             dim3(kSmallThreads), 0, st,
                                             p0.s, p0.inf, p0.n, p1.s, p1.inf, p1.n, p0.x, p0.y, p1.x, p1.y, ws.packed,
                                             sel, plan.c, plan.group_windows, plan.n_buckets, plan.wshift, ws.vals_b,
                                             ws.bucket_start, ws.bucket_end, ws.heavy_count,
                                             plan.tab ? (u32)table->stride : 0u, (u32)p0.tab_offset, (u32)p1.tab_offset));
    if ((e = hipGetLastError()) != hipSuccess) return e;
    return msm_run_tail(curve, plan, ws, st, ev_acc0, ev_acc1, plan.tab ? table->rec : nullptr);
  }
  if (plan.mid) {
    const size_t slice_points = (size_t)kSmallThreads * (size_t)plan.mid_iters;
    const u32 n_slices = (u32)((n + slice_points - 1) / slice_points);
    const int tab = plan.tab ? 1 : 0;
    const u32 n_slices_eff = tab ? n_slices * (u32)plan.digit_windows : n_slices;
    const size_t stride = tab ? plan.entries : n;  // sorted entries per output window
    u32* tmp_words = ws.keys_a;  // digit_windows * n words
    u32* ghist = ws.keys_b;      // n_windows * n_slices_eff * n_buckets words
    if (plan.mid_iters != kMidIters) {
      DISPATCH_CURVE(curve, hipLaunchKernelGGL(k_mid_count_multi<C>, dim3(n_slices, (unsigned)(plan.digit_windows / kMidWindowsPerBlock)),
                                               dim3(kSmallThreads), 0, st, p0.s, p0.inf, p0.n, p1.s, p1.inf, p1.n, p0.x, p0.y,
                                               p1.x, p1.y, ws.packed, sel, plan.c, plan.group_windows, plan.groups, plan.n_buckets,
                                               n_slices, plan.mid_iters, tmp_words, ghist, ws.heavy_count));
    } else {
      DISPATCH_CURVE(curve, hipLaunchKernelGGL(k_mid_count<C>, dim3(n_slices, (unsigned)plan.digit_windows),
                                               dim3(kSmallThreads), 0, st, p0.s, p0.inf, p0.n, p1.s, p1.inf, p1.n, p0.x,
                                               p0.y, p1.x, p1.y, ws.packed, sel, plan.c, plan.group_windows, plan.groups,
                                               plan.n_buckets, n_slices, tab, tmp_words, ghist, ws.heavy_count));
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(k_mid_offsets, dim3((unsigned)plan.n_windows), dim3(plan.n_buckets), 0, st, ghist, plan.n_buckets,
                       n_slices_eff, stride, ws.bucket_start, ws.bucket_end);
    hipLaunchKernelGGL(k_mid_scatter, dim3(n_slices, (unsigned)plan.digit_windows), dim3(kSmallThreads), 0, st, tmp_words,
                       ghist, ws.bucket_start, sel, p0.n, p1.n, plan.group_windows, plan.n_buckets, n_slices, plan.wshift,
                       tab, tab ? table->stride : 0, p0.tab_offset, p1.tab_offset, ws.vals_b, plan.mid_iters);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    return msm_run_tail(curve, plan, ws, st, ev_acc0, ev_acc1, tab ? table->rec : nullptr);
  }
  if (plan.two_level) {
    const u32 n_hi = plan.n_buckets >> kTwoLoBits;
    // counters live in the chunk-offset array, which is not written before the chunk table
    u32* part_count = ws.choff;
    u32* cursor = ws.choff + kTwoMaxBins;
    uint16_t* digits = (uint16_t*)ws.keys_b;
    uint8_t* flips = (uint8_t*)ws.vals_a;
    if ((e = hipMemsetAsync(part_count, 0, 2 * kTwoMaxBins * sizeof(u32), st)) != hipSuccess) return e;
    // points of group 0: its share of every digit window's slice of the sorted entries
    size_t size0 = 0;
    if (plan.groups > 1) {
      const MsmPart* parts[2] = {&p0, &p1};
      for (const MsmPart* pt : parts) {
        if (pt->sel_half == 0) {
          if (pt->group_lo == 0) size0 += pt->n;
        } else {
          size0 += (pt->group_lo == 0 ? pt->n / 2 : 0) + (pt->group_hi == 0 ? pt->n / 2 : 0);
        }
      }
      if (extra) size0 += 1;
    }
    const unsigned gw = (unsigned)plan.group_windows;
    DISPATCH_CURVE(curve, hipLaunchKernelGGL(k_two_digits<C>, dim3((unsigned)((n + kTwoDigitPoints - 1) / kTwoDigitPoints)),
                                             dim3(1024), 0, st, p0.s, p0.inf, p0.n, p1.s, p1.inf, p1.n, p0.x, p0.y, p1.x,
                                             p1.y, sel, plan.c, plan.group_windows, plan.groups, n_hi, ws.packed, digits,
                                             flips, part_count, plan.spread_t));
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(k_two_scatter, dim3((unsigned)((n + kTwoScatterPoints - 1) / kTwoScatterPoints), gw), dim3(1024), 0,
                       st, digits, flips, part_count, sel, p0.n, p1.n, size0, gw, (u32)plan.groups, n_hi, cursor,
                       ws.keys_a);
    hipLaunchKernelGGL(k_two_sort_lo, dim3(n_hi, (unsigned)plan.n_windows), dim3(1024), 0, st, ws.keys_a, part_count, n,
                       size0, gw, n_hi, plan.n_buckets, plan.wshift, ws.vals_b, ws.bucket_start, ws.bucket_end);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if ((e = msm_chunk_table(plan, ws, st)) != hipSuccess) return e;
    return msm_run_tail(curve, plan, ws, st, ev_acc0, ev_acc1);
  }
  DISPATCH_CURVE(curve, hipLaunchKernelGGL(k_digits<C>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                                           st, p0.s, p0.inf, p0.n, p1.s, p1.inf, p1.n, plan.c,
                                           plan.group_windows, plan.n_buckets, plan.wshift, ws.keys_a, ws.vals_a, p0.x, p0.y, p1.x, p1.y,
                                           ws.packed, sel));
  if ((e = hipGetLastError()) != hipSuccess) return e;
  if (plan.stages <= 1) {
    if ((e = msm_sort_and_chunk(plan, ws, st, 0)) != hipSuccess) return e;
    return msm_run_tail(curve, plan, ws, st, ev_acc0, ev_acc1);
  }
  // batches of windows on three streams (see MsmPipe)
  if (!pipe || !pipe->front || !pipe->reduce) return hipErrorInvalidValue;
  if ((e = hipEventRecord(pipe->start, st)) != hipSuccess) return e;
  if ((e = hipStreamWaitEvent(pipe->front, pipe->start, 0)) != hipSuccess) return e;
  if ((e = hipStreamWaitEvent(pipe->reduce, pipe->start, 0)) != hipSuccess) return e;
  const int per = (plan.n_windows + plan.stages - 1) / plan.stages;
  size_t chunk_base = 0;
  for (int g = 0, w0 = 0; w0 < plan.n_windows; g++, w0 += per) {
    const int wn = plan.n_windows - w0 < per ? plan.n_windows - w0 : per;
    // the batch as a plan and a workspace of its own: windows [w0, w0 + wn)
    MsmPlan sub = plan;
    sub.stages = 1;
    sub.n_windows = sub.group_windows = sub.digit_windows = wn;
    sub.entries = (size_t)wn * n;
    const size_t nb = (size_t)wn * plan.n_buckets, b0 = (size_t)w0 * plan.n_buckets;
    sub.nc_cap = sub.entries / plan.chunk + nb + 1;
    sub.bounds_bytes = (nb + 1) * sizeof(u32);
    MsmWorkspace wg = ws;
    wg.keys_a += (size_t)w0 * n;
    wg.keys_b += (size_t)w0 * n;
    wg.vals_a += (size_t)w0 * n;
    wg.vals_b += (size_t)w0 * n;
    wg.bucket_start += b0 + g;
    wg.bucket_end += b0 + g;
    wg.nch += b0 + g;
    wg.choff += b0 + g;
    wg.heavy += b0;
    wg.heavy_count += g;
    wg.buckets += b0 * 32;
    wg.chunk_sums += chunk_base * 32;
    wg.ckey_a += chunk_base;
    wg.ckey_b += chunk_base;
    wg.cval_a += chunk_base;
    wg.cval_b += chunk_base;
    wg.desc_b += chunk_base;
    wg.desc_s += chunk_base;
    wg.segs += (size_t)w0 * plan.n_segs * 32;
    wg.seg_groups += (size_t)w0 * (plan.n_segs / 1024 + 1) * 32;
    wg.windows += (size_t)w0 * 32;
    chunk_base += sub.nc_cap;
    if (chunk_base > plan.nc_cap) return hipErrorInvalidValue;
    if ((e = msm_sort_and_chunk(sub, wg, pipe->front, (u32)w0)) != hipSuccess) return e;
    if ((e = hipEventRecord(pipe->sorted[g], pipe->front)) != hipSuccess) return e;
    if ((e = hipStreamWaitEvent(st, pipe->sorted[g], 0)) != hipSuccess) return e;
    if ((e = msm_accumulate(curve, sub, wg, st, pipe->cs0[g], pipe->cs1[g], nullptr)) != hipSuccess) return e;
    if ((e = hipEventRecord(pipe->summed[g], st)) != hipSuccess) return e;
    if ((e = hipStreamWaitEvent(pipe->reduce, pipe->summed[g], 0)) != hipSuccess) return e;
    if ((e = msm_reduce(curve, sub, wg, pipe->reduce)) != hipSuccess) return e;
  }
  if ((e = hipEventRecord(pipe->done, pipe->reduce)) != hipSuccess) return e;
  return hipStreamWaitEvent(st, pipe->done, 0);
}

// chunked path, after the digits: radix sort of (bucket, entry) pairs, bucket bounds, chunk table
// sorted by chunk length.  window_base = number of the first window of the entries given.
static hipError_t msm_sort_and_chunk(const MsmPlan& plan, const MsmWorkspace& ws, hipStream_t st, u32 window_base) {
  hipError_t e;
  size_t tmp = plan.sort_tmp_bytes;
  e = rocprim::radix_sort_pairs<rocprim::default_config, const u32*, u32*, const u32*, u32*>(
      ws.sort_tmp, tmp, ws.keys_a, ws.keys_b, ws.vals_a, ws.vals_b, plan.entries, 0,
      key_bits_for(plan.n_buckets), st, false);
  if (e != hipSuccess) return e;
  if ((e = hipMemsetAsync(ws.bucket_start, 0, plan.bounds_bytes, st)) != hipSuccess) return e;
  if ((e = hipMemsetAsync(ws.bucket_end, 0, plan.bounds_bytes, st)) != hipSuccess) return e;
  hipLaunchKernelGGL(k_bounds, dim3((unsigned)((plan.entries + 255) / 256)), dim3(256), 0, st, ws.keys_b,
                     ws.vals_b, plan.entries, plan.n_buckets, plan.wshift, window_base, ws.bucket_start, ws.bucket_end);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  return msm_chunk_table(plan, ws, st);
}

// chunk table of sorted entries with known bucket bounds, sorted by chunk length
static hipError_t msm_chunk_table(const MsmPlan& plan, const MsmWorkspace& ws, hipStream_t st) {
  const size_t nbt = (size_t)plan.n_windows * plan.n_buckets;
  const u32 n_tiles = (u32)((nbt + 1 + kCtTile - 1) / kCtTile);
  u32* tmp = (u32*)ws.scan_tmp;
  hipLaunchKernelGGL(k_ct_count, dim3(n_tiles), dim3(kSmallThreads), 0, st, ws.bucket_start, ws.bucket_end, nbt,
                     plan.chunk, n_tiles, ws.nch, tmp);
  hipLaunchKernelGGL(k_ct_scan, dim3(1), dim3(kSmallThreads), 0, st, n_tiles, nbt, tmp, ws.choff, ws.heavy_count);
  hipLaunchKernelGGL(k_ct_emit, dim3(n_tiles), dim3(kSmallThreads), 0, st, ws.bucket_start, ws.bucket_end, ws.nch, nbt,
                     plan.chunk, n_tiles, tmp, ws.choff, ws.ckey_b, ws.cval_b, ws.desc_b, ws.desc_s);
  return hipGetLastError();
}

// accumulation and reduction over a finished entry order and chunk table
static hipError_t msm_run_tail(int curve, const MsmPlan& plan, const MsmWorkspace& ws, hipStream_t st,
                               hipEvent_t ev_acc0, hipEvent_t ev_acc1, const u32* table_rec) {
  hipError_t e = msm_accumulate(curve, plan, ws, st, ev_acc0, ev_acc1, table_rec);
  return e != hipSuccess ? e : msm_reduce(curve, plan, ws, st);
}

// bucket sums from the sorted entries
static hipError_t msm_accumulate(int curve, const MsmPlan& plan, const MsmWorkspace& ws, hipStream_t st,
                                 hipEvent_t ev_acc0, hipEvent_t ev_acc1, const u32* table_rec) {
  const size_t nbt = (size_t)plan.n_windows * plan.n_buckets;
  const u32* points = table_rec ? table_rec : ws.packed;  // 64-byte x|y records the entries index
  hipError_t e;
  const u32 idx_mask = (1u << plan.wshift) - 1u;
  if (ev_acc0 && (e = hipEventRecord(ev_acc0, st)) != hipSuccess) return e;
  if (plan.small || plan.mid) {
    // lanes per bucket: as many as keep the launch at about one wave per SIMD (two on the
    // sliced-sort path, whose launches are bound by their work rather than their chain: 2^16 points
    // 0.209 -> 0.170 ms, no change at 2^17 where the next rule asks for as many anyway; four waves per
    // SIMD lose again, 0.25 ms), and at least one per 32 expected entries
    int lanes = 64;
    // batched plans are sized for throughput: up to four waves per SIMD
    size_t lane_cap = plan.mid ? 131072 : plan.batch > 1 ? 262144 : 65536;
    if (const char* ev = getenv("BP_MSM_TREE_LANES")) lane_cap = (size_t)atol(ev);  // tuning runs
    while (lanes > 1 && nbt * (size_t)lanes > lane_cap) lanes >>= 1;
    const size_t per_bucket = plan.entries * (size_t)plan.batch / (nbt ? nbt : 1);
    while (lanes < 64 && (size_t)lanes * 32 < per_bucket) lanes <<= 1;
    // no more lanes than expected entries: every halving saves a level of the shuffle tree
    // (IPA round at 64 points: 0.21 -> 0.17 ms)
    while (lanes > 1 && (size_t)lanes > (per_bucket ? per_bucket : 1)) lanes >>= 1;
    // a bucket with more than twice the expected entries per lane (and more than 16 per lane)
    // would set the length of the launch's chain: it goes to a block of its own.  A scalar
    // field well below 2^256 fills its top window this unevenly on every input.
    size_t per_lane = 2 * ((per_bucket + (size_t)lanes - 1) / (size_t)lanes);
    static const char* heavy_env = getenv("BP_MSM_TREE_HEAVY_MIN");  // tuning runs
    const size_t heavy_min = heavy_env ? (size_t)atol(heavy_env) : 16;
    if (per_lane < heavy_min) per_lane = heavy_min;
    const u32 heavy_from = (u32)(per_lane * (size_t)lanes);
    const size_t tree_threads = nbt * (size_t)lanes;
    static const bool plan_trace = getenv("BP_MSM_PLAN_TRACE") != nullptr;  // tuning runs: one line per launch
    if (plan_trace)
      fprintf(stderr, "msm_accumulate: n_total %zu batch %d c %d windows %d buckets %u entries %zu per_bucket %zu lanes %d heavy_from %u\n",
              plan.n_total, plan.batch, plan.c, plan.n_windows, plan.n_buckets, plan.entries, per_bucket, lanes, heavy_from);
    // A bucket holds at most the entries of its window, n_total.  When that is within an eighth
    // of kTreeHeavyMax (the IPA rounds of a 4096-bit range proof: 8194) the blocks inside the
    // launch take every crowded bucket, nothing can be queued, and the launch for the queue is
    // left out (about 10 us of a round's chain of launches).
    const bool no_queue = (!plan.tab || plan.tab_virtual) && plan.n_total <= kTreeHeavyMax + kTreeHeavyMax / 8 &&
                          nbt <= (size_t)kTreeHeavyBlocks * kTreeHeavyList;
    const u32 heavy_max = no_queue && plan.n_total > kTreeHeavyMax ? (u32)plan.n_total : kTreeHeavyMax;
    // the lanes of a bucket joined by quad additions (k_bucket_tree<C, true>) while the launch is
    // about a wave per SIMD or less, where the chain of additions sets its time; BP_MSM_TREE_JOIN=0/1
    // forces a choice
    static const char* join_env = getenv("BP_MSM_TREE_JOIN");
    const bool quad_join = join_env ? atoi(join_env) != 0 : (lanes >= 2 && tree_threads <= 2 * 65536);
    // Four lanes per addition (k_bucket_quad) where few buckets hold many entries each (table plans
    // whose digit windows share a bucket set): there the chain of dependent additions sets the time.
    // With a few dozen entries per bucket the launch is bound by its total work and the one-lane tree
    // wins (IPA round of 2 * 4096 points: 60 us against 78 us).  BP_MSM_TREE_QUAD=1/0 forces a choice.
    static const char* quad_env = getenv("BP_MSM_TREE_QUAD");
    const bool quad_tree = quad_env ? atoi(quad_env) != 0 : (nbt <= 512 && per_bucket >= 512);
    if (quad_tree) {
      // quads per bucket: about four entries each, at most the 64 of a block
      int log_q = 0;
      while (log_q < 6 && ((size_t)4 << log_q) < per_bucket) log_q++;
      const size_t n_quads = (size_t)1 << log_q;
      size_t per_quad = 2 * ((per_bucket + n_quads - 1) / n_quads);
      if (per_quad < 16) per_quad = 16;
      const u32 q_heavy_from = (u32)(per_quad * n_quads);
      const size_t per_block = (size_t)(kBucketQuadThreads / 4) >> log_q;
      DISPATCH_CURVE(curve, {
        hipLaunchKernelGGL(k_bucket_quad<C>, dim3((unsigned)((nbt + per_block - 1) / per_block) + kTreeHeavyBlocks),
                           dim3(kBucketQuadThreads), 0, st, points, ws.vals_b, ws.bucket_start, ws.bucket_end, nbt, log_q,
                           q_heavy_from, heavy_max, idx_mask, ws.buckets, ws.heavy, ws.heavy_count);
        if (!no_queue)
          hipLaunchKernelGGL(k_bucket_heavy_entries<C>, dim3(256), dim3(1024), 0, st, points, ws.vals_b,
                             ws.bucket_start, ws.bucket_end, nbt, idx_mask, ws.buckets, ws.heavy, ws.heavy_count);
      });
    } else if (nbt <= 512 && per_bucket >= 512) {
      // few buckets with many entries each (table plans on small inputs): a block per bucket
      DISPATCH_CURVE(curve, hipLaunchKernelGGL(k_bucket_heavy_entries<C>, dim3((unsigned)nbt), dim3(1024), 0, st, points,
                                               ws.vals_b, ws.bucket_start, ws.bucket_end, nbt, idx_mask, ws.buckets,
                                               (const u32*)nullptr, (const u32*)nullptr));
    } else
    DISPATCH_CURVE(curve, {
      if (quad_join)
        hipLaunchKernelGGL((k_bucket_tree<C, true>), dim3((unsigned)((tree_threads + 255) / 256) + kTreeHeavyBlocks),
                           dim3(256), 0, st, points, ws.vals_b, ws.bucket_start, ws.bucket_end, nbt, lanes, heavy_from,
                           heavy_max, idx_mask, ws.buckets, ws.heavy, ws.heavy_count);
      else
        hipLaunchKernelGGL((k_bucket_tree<C, false>), dim3((unsigned)((tree_threads + 255) / 256) + kTreeHeavyBlocks),
                           dim3(256), 0, st, points, ws.vals_b, ws.bucket_start, ws.bucket_end, nbt, lanes, heavy_from,
                           heavy_max, idx_mask, ws.buckets, ws.heavy, ws.heavy_count);
      if (!no_queue)
        hipLaunchKernelGGL(k_bucket_heavy_entries<C>, dim3(256), dim3(1024), 0, st, points, ws.vals_b,
                           ws.bucket_start, ws.bucket_end, nbt, idx_mask, ws.buckets, ws.heavy, ws.heavy_count);
    });
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (ev_acc1 && (e = hipEventRecord(ev_acc1, st)) != hipSuccess) return e;
  } else {
    DISPATCH_CURVE(curve, hipLaunchKernelGGL(k_chunk_sums<C>, dim3((unsigned)((plan.nc_cap + 127) / 128)), dim3(128),
                                             0, st, ws.packed, ws.vals_b, ws.choff + nbt, ws.cval_b, ws.desc_b, ws.desc_s,
                                             ws.bucket_end, ws.nch, nbt, plan.chunk, plan.nc_cap, idx_mask,
                                             (size_t)plan.entries, ws.buckets, ws.chunk_sums));
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (ev_acc1 && (e = hipEventRecord(ev_acc1, st)) != hipSuccess) return e;
    // expected chunks per bucket; beyond a handful every bucket gets a lane group
    const size_t per_bucket_chunks = plan.entries / (nbt ? nbt : 1) / plan.chunk;
    if (per_bucket_chunks > kFinishSerial / 2 && nbt * 16 <= ((size_t)1 << 22)) {
      const int lanes = 16;
      DISPATCH_CURVE(curve, hipLaunchKernelGGL(k_bucket_finish_tree<C>, dim3((unsigned)((nbt * lanes + 255) / 256)),
                                               dim3(256), 0, st, ws.nch, ws.choff, nbt, plan.nc_cap, ws.chunk_sums,
                                               lanes, ws.buckets));
    } else {
      DISPATCH_CURVE(curve, {
        hipLaunchKernelGGL(k_bucket_finish<C>, dim3((unsigned)((nbt + 127) / 128)), dim3(128), 0, st, ws.nch, ws.choff,
                           nbt, plan.nc_cap, ws.chunk_sums, ws.buckets, ws.heavy, ws.heavy_count);
        hipLaunchKernelGGL(k_bucket_heavy_wave<C>, dim3(1024), dim3(256), 0, st, ws.nch, ws.choff, nbt, plan.nc_cap,
                           ws.chunk_sums, ws.buckets, ws.heavy, ws.heavy_count);
        hipLaunchKernelGGL(k_bucket_heavy<C>, dim3(256), dim3(256), 0, st, ws.nch, ws.choff, nbt, plan.nc_cap,
                           ws.chunk_sums, ws.buckets, ws.heavy, ws.heavy_count);
      });
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  return hipSuccess;
}

// window sums from the bucket sums
static hipError_t msm_reduce(int curve, const MsmPlan& plan, const MsmWorkspace& ws, hipStream_t st) {
  const size_t nbt = (size_t)plan.n_windows * plan.n_buckets;
  // four lanes per point addition (bp_reduce.hip) unless BP_MSM_REDUCE_SERIAL asks for one
  static const bool serial_reduce = getenv("BP_MSM_REDUCE_SERIAL") != nullptr;
  // many windows (batched plans): lanes that each run a short chain fill the machine; a block of quads per
  // window would leave most of it waiting for 4096 chains made as short as one can be
  static const int lanes_from = getenv("BP_MSM_REDUCE_LANES_FROM") ? atoi(getenv("BP_MSM_REDUCE_LANES_FROM")) : 512;
  static const int lanes_log = getenv("BP_MSM_REDUCE_LPW_LOG") ? atoi(getenv("BP_MSM_REDUCE_LPW_LOG")) : 4;
  if (plan.n_windows >= lanes_from && plan.n_buckets >= 8 && plan.n_buckets <= 256) {
    int log_nb = 0;
    while ((1u << log_nb) < plan.n_buckets) log_nb++;
    int log_lpw = lanes_log >= 1 && lanes_log <= 6 ? lanes_log : 4;
    if (log_lpw > log_nb) log_lpw = log_nb;  // a lane per bucket at most (8 buckets per window: 8 lanes)
    const size_t threads = (size_t)plan.n_windows << log_lpw;
    DISPATCH_CURVE(curve, hipLaunchKernelGGL(k_window_reduce_lanes<C>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st,
                                             ws.buckets, nbt, plan.n_buckets, log_lpw, log_nb - log_lpw, (u32)plan.n_windows,
                                             ws.windows));
    return hipGetLastError();
  }
  if (plan.n_buckets >= 64 && plan.n_buckets <= 256) {
    if (!serial_reduce)
      return launch_window_reduce_quad(curve, ws.buckets, nbt, plan.n_buckets, plan.n_windows, ws.windows, st);
    int log_nb = 0;
    while ((1u << log_nb) < plan.n_buckets) log_nb++;
    DISPATCH_CURVE(curve, hipLaunchKernelGGL(k_window_reduce_small<C>, dim3((unsigned)plan.n_windows),
                                             dim3(plan.n_buckets), 0, st, ws.buckets, nbt, plan.n_buckets, log_nb,
                                             ws.windows));
    return hipGetLastError();
  }
  if (plan.rowcol) {
    const u32 n_rows = plan.n_buckets / kRowcolCols;
    int log_rows = 0;
    while ((1u << log_rows) < n_rows) log_rows++;
    int strip = kRowcolStrip;
    if (const char* ev = getenv("BP_MSM_ROWCOL_STRIP")) strip = atoi(ev);  // tuning runs
    if (strip != 4 && strip != 8 && strip != 16 && strip != 32) strip = kRowcolStrip;
    const int row_lanes = (int)kRowcolCols / strip;
    int col_lanes = (int)n_rows / strip;
    if (col_lanes < 1) col_lanes = 1;
    const size_t nw = (size_t)plan.n_windows;
    const u32 row_blocks = (u32)((nw * n_rows * (size_t)row_lanes + 255) / 256);
    const u32 col_blocks = (u32)((nw * kRowcolCols * (size_t)col_lanes + 255) / 256);
    u32* cols = ws.segs;
    u32* rows = cols + nw * kRowcolCols * 32;
    u32* parts = rows + nw * n_rows * 32;
    DISPATCH_CURVE(curve, {
      hipLaunchKernelGGL(k_rowcol_sums<C>, dim3(row_blocks + col_blocks), dim3(256), 0, st, ws.buckets, nbt,
                         plan.n_buckets, plan.n_windows, row_blocks, row_lanes, col_lanes, cols, rows,
                         serial_reduce ? nullptr : ws.heavy_count + 1);
      if (serial_reduce) {
        hipLaunchKernelGGL(k_rowcol_final<C>, dim3(2 * (unsigned)plan.n_windows), dim3(256), 0, st, cols, rows,
                           plan.n_windows, n_rows, log_rows, parts);
        hipLaunchKernelGGL((k_window_combine<C, true>), dim3((plan.n_windows + 63) / 64), dim3(64), 0, st, parts, 2,
                           plan.n_windows, ws.windows);
      } else {
        // the second of a window's two blocks to finish adds the parts (counters behind heavy_count,
        // cleared by k_rowcol_sums)
        if (plan.n_windows > 63) return hipErrorInvalidValue;
        hipError_t e = launch_rowcol_final_quad(curve, cols, rows, plan.n_windows, n_rows, parts, ws.heavy_count + 1,
                                                ws.windows, st);
        if (e != hipSuccess) return e;
      }
    });
    return hipGetLastError();
  }
  size_t nseg_threads = (size_t)plan.n_windows * plan.n_segs;
  DISPATCH_CURVE(curve, {
    // 256-lane blocks: one wave on each SIMD of a CU.  The kernel is a long dependent chain
    // per lane, and with smaller blocks the dispatcher stacks waves of different blocks on
    // one SIMD while others idle.
    hipLaunchKernelGGL(k_reduce_segments<C>, dim3((unsigned)((nseg_threads + 255) / 256)), dim3(256), 0, st,
                       ws.buckets, nbt, plan.n_buckets, plan.seg_len, plan.n_segs, plan.n_windows,
                       ws.segs);
    // window value = sum of its segment values, in one or two tree stages
    if (plan.n_segs > 1024) {
      u32 groups = plan.n_segs / 1024;  // n_segs is a power of two
      hipLaunchKernelGGL((k_group_sum<C, false>), dim3(plan.n_windows * groups), dim3(256), 0, st, ws.segs, 1024u, ws.seg_groups);
      hipLaunchKernelGGL((k_group_sum<C, true>), dim3(plan.n_windows), dim3(256), 0, st, ws.seg_groups, groups, ws.windows);
    } else {
      hipLaunchKernelGGL((k_group_sum<C, true>), dim3(plan.n_windows), dim3(256), 0, st, ws.segs, plan.n_segs, ws.windows);
    }
  });
  return hipGetLastError();
}

}  // namespace bp
