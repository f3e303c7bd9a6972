// Internal host-side declarations shared by the translation units of libbp_gpu.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <mutex>
#include <string>

#include "../../include/bp_gpu.h"

namespace bp {

typedef uint32_t u32;

struct FeArg {  // field element passed by value as a kernel argument
  u32 v[8];
};

// ---- streaming field-vector kernels (bp_vec.hip); all pointers are device ----
// field: 0 = scalar field Fr, 1 = base field Fq
hipError_t launch_fold_scalars(int curve, const u32* l, const u32* r, size_t h, FeArg x, FeArg y,
                               u32* out, hipStream_t st);
// b may be null (plain sum).  partials: >= kDotBlocks*8 u32.  out: 8 u32.
constexpr int kDotBlocks = 1024;
hipError_t launch_inner_product(int curve, const u32* a, const u32* b, size_t n, u32* partials,
                                u32* out, hipStream_t st);
hipError_t launch_vec_op(int curve, int op, const u32* a, const u32* b, FeArg s, size_t n, u32* out,
                         hipStream_t st);
// ch, ch_inv: k elements each; table: >= (2^lo + 2^(k-lo)) * 8 u32 with lo = min(k, 8)
hipError_t launch_s_vector(int curve, const u32* ch, const u32* ch_inv, int k, u32* table, u32* out,
                           hipStream_t st);
// acc (2^k elements) += w * s_vector(ch, ch_inv); first != 0 overwrites acc.  w in Montgomery form.
hipError_t launch_s_vector_axpy(int curve, const u32* ch, const u32* ch_inv, int k, FeArg w, int first, u32* table,
                                u32* acc, hipStream_t st);
size_t s_vector_table_elems(int k);
// table: >= (256 + ceil(n/256)) * 8 u32
// compressed point encoding of ark-serialize (33 bytes per point), bp_wire.hip
hipError_t launch_points_compress(int curve, const u32* x, const u32* y, const uint8_t* inf, size_t n, uint8_t* out,
                                  hipStream_t st);
hipError_t launch_scalars_to_bytes(int curve, const u32* v, size_t n, u32* out, hipStream_t st);
hipError_t launch_scalars_from_bytes(int curve, const u32* in, size_t n, u32* v, uint8_t* ok, hipStream_t st);
hipError_t launch_points_decompress(int curve, const uint8_t* in, size_t n, u32* x, u32* y, uint8_t* inf, uint8_t* ok,
                                    hipStream_t st);
hipError_t launch_powers(int curve, FeArg x, size_t n, u32* table, u32* out, hipStream_t st);
size_t powers_table_elems(size_t n);
hipError_t launch_batch_inverse(int curve, int field, u32* v, size_t n, hipStream_t st);
// IPA rounds over unfolded generators: per-generator coefficients cg, ch (n0 each; ch null for
// the one-generator-vector variant) and the cross-term scalars eg, eh (n0 each) of the round
// whose current length is 2*half.
hipError_t launch_ipa_coef_update(int curve, u32* cg, u32* ch, size_t n0, size_t half, FeArg x, FeArg x_inv,
                                  hipStream_t st);
// vector polynomials of the aggregated range proof (src/range/aggregate.rs:93-127): l0, r0, r1 from
// the bits of `values` (m = n / n_bits of them), zpow[j] = z^(2+j), pow2[i] = 2^i, yv[k] = y^k and
// the blinders; out[0] = t1, out[1] = t2.  partials: >= 3 * ceil(n/256) elements.
hipError_t launch_range_poly(int curve, const uint64_t* values, unsigned n_bits, size_t n, const u32* zpow,
                             const u32* pow2, const u32* yv, const u32* s_l, const u32* s_r, FeArg z, u32* l0, u32* r0,
                             u32* r1, u32* partials, u32* out, hipStream_t st);
// l = l0 + x*l1, r = r0 + x*r1, out = <l, r> (:143-146).  partials: >= ceil(n/256) elements.
hipError_t launch_range_eval(int curve, const u32* l0, const u32* l1, const u32* r0, const u32* r1, size_t n, FeArg x,
                             u32* l, u32* r, u32* partials, u32* out, hipStream_t st);
// coefficient update and both scalar folds of a round in one launch (a, b folded in place)
hipError_t launch_ipa_round_fold(int curve, u32* a, u32* b, u32* cg, u32* ch, size_t n0, size_t half, FeArg x,
                                 FeArg x_inv, hipStream_t st);
// both inner products of a round (one block), their multiples by `factor` when scale != 0, and the
// MSM scalars in one launch; half <= kIpaPrepMaxHalf.  dots: c_l, c_r, factor*c_l, factor*c_r;
// host_dots (may be null): 64 bytes of host memory the device can write, which also receive c_l, c_r
constexpr size_t kIpaPrepMaxHalf = 16384;
hipError_t launch_ipa_round_prep(int curve, const u32* a, const u32* b, const u32* cg, const u32* ch, size_t n0,
                                 size_t half, FeArg factor, int scale, u32* eg, u32* eh, u32* dots, u32* host_dots,
                                 hipStream_t st);
// bp_ipa_round_fold and the launch_ipa_round_prep of the next round in one launch; (a, b) are folded
// out of place into (a2, b2), half = length of the folded vectors >= 2, half / 2 <= kIpaPrepMaxHalf
hipError_t launch_ipa_fold_prep(int curve, const u32* a, const u32* b, u32* a2, u32* b2, u32* cg, u32* ch, size_t n0,
                                size_t half, FeArg x, FeArg x_inv, FeArg factor, int scale, u32* eg, u32* eh, u32* dots,
                                u32* host_dots, hipStream_t st);
hipError_t launch_ipa_scalars(int curve, const u32* a, const u32* b, const u32* cg, const u32* ch, size_t n0,
                              size_t half, u32* eg, u32* eh, hipStream_t st);
hipError_t launch_fill_one(int curve, u32* v, size_t n, hipStream_t st);
hipError_t launch_int_mad_peak(int iters, u32* sink, int blocks, hipStream_t st);

// io: 1024*8 u32 of seed data; ids: 0 fe_mul pair, 1 add+sub pair, 2 xyzz_madd, 3 jac_dbl, 4 fe_mul (Vesta Fq)
hipError_t launch_microbench(int id, int blocks, int iters, u32* io, hipStream_t st);
// records: power of two; table: records * 16 u32
hipError_t launch_gather_bench(const u32* table, size_t records, u32* sink, hipStream_t st);
// batch-affine additions (bp_microbench ids 11, 12): park = blocks * 128 * iters field elements
hipError_t launch_batch_affine_bench(int blocks, int iters, const u32* io, u32* park, u32* sink, hipStream_t st);
hipError_t launch_batch_affine_check(int blocks, u32* io, hipStream_t st);

// ---- curve-vector kernels (bp_curvevec.hip) ----
// scalars are canonical (non-Montgomery) little-endian 256-bit integers here
hipError_t launch_fold_generators(int curve, const u32* lx, const u32* ly, const uint8_t* linf,
                                  const u32* rx, const u32* ry, const uint8_t* rinf, size_t h,
                                  FeArg x_canon, FeArg y_canon, u32* ox, u32* oy, uint8_t* oinf,
                                  hipStream_t st);
struct FoldJob {
  const u32 *lx = nullptr, *ly = nullptr;
  const uint8_t* linf = nullptr;
  const u32 *rx = nullptr, *ry = nullptr;
  const uint8_t* rinf = nullptr;
  u32 *ox = nullptr, *oy = nullptr;
  uint8_t* oinf = nullptr;
  FeArg x{}, y{};  // canonical integers
};
// two independent folds of the same length in one launch (j1.lx == nullptr: one fold)
hipError_t launch_fold_generators2(int curve, const FoldJob& j0, const FoldJob& j1, size_t h, hipStream_t st);
// s: Montgomery-form scalars, one per point
hipError_t launch_scale_points(int curve, const u32* px, const u32* py, const uint8_t* pinf,
                               const u32* s, size_t n, u32* ox, u32* oy, uint8_t* oinf,
                               hipStream_t st);
// table: n_windows * n records of 16 u32
hipError_t launch_window_tables(int curve, const u32* px, const u32* py, const uint8_t* pinf, size_t n, int c,
                                int n_windows, u32* table, hipStream_t st);
hipError_t launch_normalize_jac(int curve, const u32* xyz, size_t n, u32* ox, u32* oy, uint8_t* oinf,
                                hipStream_t st);
// Folded generators G'_i = sum_{t<m} s_t * G0[i + n t], i < n (affine planes out), from the signed
// 3-bit digits of the m shared coefficients (m x kRebaseWindows int8 in [-4, 4], device memory);
// tmp_windows: n * kRebaseWindows XYZZ points (128 bytes each).  One or two sets (the G and H of a
// session) per call.
constexpr int kRebaseWindows = 86;
struct RebaseSet {
  const u32* gx;
  const u32* gy;
  const uint8_t* ginf;
  const int8_t* d_digits;
  u32* tmp_windows;
  u32* ox;
  u32* oy;
  uint8_t* oinf;
};
hipError_t launch_ipa_rebase(int curve, const RebaseSet* sets, int n_sets, size_t n, size_t m, hipStream_t st);
hipError_t launch_gather_column(const u32* src, size_t stride_elems, size_t count, u32* dst, hipStream_t st);
// The same folded generators from the window tables of a registered base set (launch_window_tables:
// record (w, j) = 2^(8 w) * P_j): G'_i = sum over the non-zero signed 8-bit digits d of the m shared
// coefficients s_t of d * T[w][off + i + n t], a sum of small multiples of known points.  Output i is
// made by 2^log_s adjacent lanes, each running its own list of `len` steps "point += point" over three
// points it keeps (run, total, tmp) and the table records (rebase_tab_program builds the lists: entries
// sorted by digit size, run/total as in a bucket reduction, the rest of a lane's weight by double-and-add);
// the lanes' totals are added and normalised.  Steps are u32 words:
//   dst << 30 | src << 28 | neg << 27 | w << 22 | t     dst, src: 0 run, 1 total, 2 tmp; src 3: table record
//   (w, t), negated if neg; dst 3: no operation
constexpr u32 kRebaseTabNop = 0xFFFFFFFFu;
constexpr size_t kRebaseTabProgMax = 4096;  // words of all lists of one set together (2^log_s * len)
constexpr int kRebaseTabMaxLogS = 4;
struct RebaseTabSet {
  u32 off;             // index of the set's first generator in the base set
  const uint8_t* inf;  // identity flags of the base set's points, may be null
  const u32* prog;     // 2^log_s lists of len steps, device memory
  u32* ox;
  u32* oy;
  uint8_t* oinf;
  // several problems over the same table in one launch (n_problems > 1): problem z reads its lists
  // prog_z_stride words and writes its outputs out_z_stride points further on
  size_t prog_z_stride = 0, out_z_stride = 0;
};
hipError_t launch_ipa_rebase_tab(int curve, const u32* table, size_t stride, const RebaseTabSet* sets, int n_sets, size_t n,
                                 int log_s, int len, hipStream_t st, int n_problems = 1);
// window tables (launch_window_tables) with one inversion per base; scratch: n_windows * n * 128 bytes
hipError_t launch_window_tables_inv1(int curve, const u32* px, const u32* py, const uint8_t* pinf, size_t n, int c,
                                     int n_windows, u32* scratch, u32* table, hipStream_t st);
// n records of 72 bytes {x, y, infinity byte, padding} -> planes
hipError_t launch_aos_split(const u32* aos, size_t n, u32* x, u32* y, uint8_t* inf, hipStream_t st);

// ---- MSM (bp_msm.hip) ----
struct MsmPart {
  const u32* x;
  const u32* y;
  const uint8_t* inf;  // may be null
  const u32* s;        // Montgomery-form scalars
  size_t n;
  // Two-group plans only: which of the two sums a point contributes to.  sel_half == 0: the
  // whole part goes to group_lo.  Otherwise index i goes to group_hi when
  // (i mod 2*sel_half) >= sel_half and to group_lo when not (n must be a multiple of 2*sel_half).
  size_t sel_half = 0;
  u32 group_lo = 0, group_hi = 0;
  size_t tab_offset = 0;  // table plans: index of this part's first point in the base set
  // batched plans (MsmPlan::batch > 1): problem b reads its scalars at s + b * batch_stride (u32 units);
  // the points are the same for every problem of the batch
  size_t batch_stride = 0;
};

// Window tables of a registered base set (launch_window_tables): record (w, j) at w*stride + j
struct MsmTable {
  const u32* rec = nullptr;
  size_t stride = 0;
  // batched plans: problem b's points sit sess_stride base indices after problem 0's (0: one base set for all)
  size_t sess_stride = 0;
  // digit width of the tables: record (w, j) = 2^(bits w) * P_j, 256 / bits windows.  8 for registered base sets;
  // 4 for the folded generators of a batch (few points per window: eight buckets instead of 128 to reduce)
  int bits = 8;
};
constexpr int kTableWindowBits = 8;
constexpr int kTableWindows = 32;

struct MsmPlan {
  int c;          // window bits
  int groups;     // independent sums computed in one pass (1 or 2)
  int group_windows;  // windows per sum
  int n_windows;  // over all groups; group g owns windows [g*group_windows, (g+1)*group_windows)
  int wshift;     // bit position of the window number inside a sorted value word
  u32 n_buckets;  // per window = 2^(c-1)
  size_t n_total; // points over both parts
  size_t entries; // W * n_total
  // workspace sizes in bytes
  size_t keys_bytes, sort_tmp_bytes, bounds_bytes, bucket_bytes, seg_bytes, seg_group_bytes, win_bytes, packed_bytes;
  size_t scan_tmp_bytes, nch_bytes, chunk_sum_bytes, cdesc_bytes, csort_tmp_bytes;
  u32 chunk;      // max sorted entries per accumulation chunk
  size_t nc_cap;  // upper bound on the number of chunks
  int seg_len;    // buckets per reduction segment
  u32 n_segs;     // per window
  bool small;     // one-launch front end (digits, counting sort, bounds) per window, tree accumulation
  bool mid;       // sliced counting sort (three launches), tree accumulation
  int mid_iters;  // sliced counting sort: points per lane and slice (a slice is 1024 * mid_iters points)
  bool tab;       // points come from window tables: all digit windows share one bucket set per group
  // table plan small enough for the one-launch front end: the digit windows keep their own bucket
  // sets (the launches are those of the plan without tables, only the points read differ) and the
  // window sums, which the tables have already multiplied by their powers of two, are added up
  // without doublings
  bool tab_virtual;
  int finish_c;   // doublings between consecutive window sums when they are combined: c, or 0 (tab_virtual)
  int digit_windows;  // windows of the digit decomposition (= group_windows unless tab)
  int stages;     // > 1: the windows are processed in this many consecutive batches on three streams
  bool two_level; // chunked path: two-level counting sort in place of the key-value radix sort
  bool rowcol;    // bucket reduction by row and column sums of the (n_buckets/256) x 256 bucket matrix
  // > 1: `batch` independent problems of the same shape over the same points (table plans of the
  // one-launch front end only): one grid per phase serves them all; n_windows counts the windows of all
  // problems (problem b owns windows [b * n_windows / batch, (b + 1) * n_windows / batch)), n_total and
  // entries those of one problem
  int batch;
  // > 0 (two-level plans whose width does not divide 256): every scalar k is taken as k + t r with a per-point
  // t < spread_t (a power of two; r P = 0, the sum is the same), so that the values fill group_windows * c - 1
  // bits evenly and the top window is as full as the others; no folding to |k| <= r/2 then
  u32 spread_t;
};

// Large single-sum plans run their windows in batches: sorting and the chunk table of batch
// g+1 (memory-bound) and the bucket reduction of batch g-1 (a long dependent chain on few
// lanes) overlap the accumulation of batch g (multiplier-bound).  Streams and events are owned
// by the caller; cs0/cs1 bracket each batch's accumulation kernel.
constexpr int kMsmMaxStages = 8;
struct MsmPipe {
  hipStream_t front = nullptr, reduce = nullptr;
  hipEvent_t start = nullptr, done = nullptr;
  hipEvent_t sorted[kMsmMaxStages] = {}, summed[kMsmMaxStages] = {};
  hipEvent_t cs0[kMsmMaxStages] = {}, cs1[kMsmMaxStages] = {};
};

int msm_choose_window(size_t n);
int msm_default_window(size_t n);
hipError_t msm_make_plan(int curve, size_t n_total, int c_override, int groups, MsmPlan* plan, size_t table_stride = 0,
                         bool allow_stages = false, int batch = 1, int table_bits = 8);
constexpr int kMsmMaxBatch = 128;
// out[q] = sum of the `count` XYZZ points (AoS, 32 words, boundary form) in + (q * count) * 32, q < n_sums: the
// window sums of a table plan whose windows carry their powers of two already (finish_c == 0), added on the
// device; out may be host memory the device can write
hipError_t launch_window_group_sums(int curve, const u32* in, int count, int n_sums, u32* out, hipStream_t st);

struct MsmWorkspace {
  u32* packed;        // points as x|y pairs, 16 u32 each
  u32* keys_a;
  u32* keys_b;
  u32* vals_a;
  u32* vals_b;
  void* sort_tmp;
  u32* bucket_start;  // (W*NB + 2)
  u32* bucket_end;
  u32* nch;           // chunks per bucket (W*NB + 1)
  u32* choff;         // exclusive scan of nch
  void* scan_tmp;
  u32* chunk_sums;    // XYZZ SoA: 4 planes * nc_cap * 8 u32
  u32 *ckey_a, *ckey_b, *cval_a, *cval_b;  // chunk-length sort (nc_cap each)
  u32 *desc_b, *desc_s;                    // chunk -> bucket, first sorted position
  void* csort_tmp;
  u32* heavy;         // queue of buckets with many chunks (W*NB)
  u32* heavy_count;
  u32* buckets;       // XYZZ SoA: 4 planes * (W*NB) * 8 u32
  u32* segs;          // XYZZ AoS: W * n_segs * 32 u32
  u32* seg_groups;    // XYZZ AoS: partial sums of 1024 segments
  u32* windows;       // XYZZ AoS: W * 32 u32
};

// Two-group split plans only: one more (point, scalar) per group, both read from device memory
// when the digits kernel runs (so a scalar may be produced by a kernel queued just before).
// The plan's n_total counts them.
struct MsmExtra {
  const u32* xy[2];  // affine x|y, 16 u32, not the identity; extra e goes to group e
  const u32* s[2];   // Montgomery-form scalar, 8 u32
  size_t tab_index[2] = {0, 0};  // table plans: index of the point in the base set
  size_t batch_stride[2] = {0, 0};  // batched plans: problem b reads extra e's scalar at s[e] + b * batch_stride[e] (u32 units)
};

// Runs digits -> sort -> bucket accumulation -> bucket reduction; leaves W window
// sums (XYZZ, 32 u32 each) in ws.windows.  ev_acc0/ev_acc1 (optional) bracket the
// accumulation kernel.
hipError_t msm_run(int curve, const MsmPart& p0, const MsmPart& p1, const MsmExtra* extra, const MsmPlan& plan,
                   const MsmWorkspace& ws, hipStream_t st, hipEvent_t ev_acc0, hipEvent_t ev_acc1,
                   const MsmTable* table = nullptr, const MsmPipe* pipe = nullptr);
// sums n_parts arrays of n_windows XYZZ points window by window into out (n_windows XYZZ)
// window values of 2..256 buckets per window, four lanes per point addition (bp_reduce.hip)
hipError_t launch_window_reduce_quad(int curve, const u32* buckets, size_t nbt, u32 n_buckets, int n_windows,
                                     u32* windows, hipStream_t st);
hipError_t launch_rowcol_final_quad(int curve, const u32* cols, const u32* rows, int n_windows, u32 n_rows, u32* parts,
                                    u32* done, u32* windows, hipStream_t st);
hipError_t launch_window_combine(int curve, const u32* windows, int n_parts, int n_windows, u32* out,
                                 hipStream_t st);

}  // namespace bp
