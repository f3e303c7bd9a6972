// 256-bit prime-field arithmetic for gfx950, Montgomery form (a*2^256 mod p),
// eight 32-bit limbs per element held in VGPRs.  The multiply is a CIOS
// Montgomery product written so that every partial product is one
// v_mad_u64_u32 (32x32+64 -> 64); there is no floating point and no MFMA use:
// the work is not a dense contraction.
//
// Replaces, on the device, the ark-ff 0.5.0 `Fp<MontBackend<_,4>>` operations
// the reference reaches through `Field` (src/vector_ops.rs:159-179,
// src/ipa/mod.rs:229-234).  Memory format is exactly arkworks': four
// little-endian u64 limbs = eight little-endian u32 limbs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bp_field_consts.h"

namespace bp {

struct Fe {
  uint32_t v[8];
};

#define BP_DEV __host__ __device__ __forceinline__

BP_DEV Fe fe_load(const uint32_t* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 lo = q[0], hi = q[1];
  Fe r;
  r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
  r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
  return r;
}

BP_DEV void fe_store(uint32_t* p, const Fe& a) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
  q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}

// streaming forms: data that is read or written once should not displace what the caches hold
BP_DEV Fe fe_load_nt(const uint32_t* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef uint32_t v4u __attribute__((ext_vector_type(4)));
  const v4u* q = reinterpret_cast<const v4u*>(p);
  v4u lo = __builtin_nontemporal_load(q), hi = __builtin_nontemporal_load(q + 1);
  Fe r;
  r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
  r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
  return r;
#else
  return fe_load(p);
#endif
}

BP_DEV void fe_store_nt(uint32_t* p, const Fe& a) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef uint32_t v4u __attribute__((ext_vector_type(4)));
  v4u* q = reinterpret_cast<v4u*>(p);
  v4u lo = {a.v[0], a.v[1], a.v[2], a.v[3]}, hi = {a.v[4], a.v[5], a.v[6], a.v[7]};
  __builtin_nontemporal_store(lo, q);
  __builtin_nontemporal_store(hi, q + 1);
#else
  fe_store(p, a);
#endif
}

BP_DEV Fe fe_zero() {
  Fe r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = 0;
  return r;
}

template <class F>
BP_DEV Fe fe_one() {
  Fe r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = F::one(i);
  return r;
}

BP_DEV bool fe_is_zero(const Fe& a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) o |= a.v[i];
  return o == 0;
}

BP_DEV bool fe_eq(const Fe& a, const Fe& b) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) o |= a.v[i] ^ b.v[i];
  return o == 0;
}

BP_DEV Fe fe_select(bool c, const Fe& a, const Fe& b) {  // c ? a : b
  Fe r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = c ? a.v[i] : b.v[i];
  return r;
}

// Carry chains are written with __builtin_addc/__builtin_subc, which hipcc lowers
// to v_add_co_u32 / v_addc_co_u32 (v_sub_co / v_subb_co) runs with the modulus
// limbs as literals.

// reduce a value known to be < 2p (carry = bit 256 of the value)
template <class F>
BP_DEV Fe fe_reduce_once(const Fe& a, uint32_t carry) {
  Fe d;
  uint32_t bw = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) d.v[i] = __builtin_subc(a.v[i], F::mod(i), bw, &bw);
  bool keep = F::kFullWidth ? (carry == 0 && bw != 0) : (bw != 0);
  return fe_select(keep, a, d);
}

// carry * 2^256 + a, below 3p, brought below p (the dual product's tail)
template <class F>
BP_DEV Fe fe_reduce_twice(Fe a, uint32_t carry) {
#pragma unroll
  for (int round = 0; round < 2; round++) {
    Fe d;
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d.v[i] = __builtin_subc(a.v[i], F::mod(i), bw, &bw);
    const bool ge = carry != 0 || bw == 0;
    a = fe_select(ge, d, a);
    carry = ge ? carry - bw : carry;
  }
  return a;
}

template <class F>
BP_DEV Fe fe_add(const Fe& a, const Fe& b) {
  Fe s;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s.v[i] = __builtin_addc(a.v[i], b.v[i], c, &c);
  return fe_reduce_once<F>(s, c);
}

template <class F>
BP_DEV Fe fe_sub(const Fe& a, const Fe& b) {
  Fe d, r;
  uint32_t bw = 0, c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) d.v[i] = __builtin_subc(a.v[i], b.v[i], bw, &bw);
  uint32_t mask = 0u - bw;  // all ones when a < b
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = __builtin_addc(d.v[i], F::mod(i) & mask, c, &c);
  return r;
}

template <class F>
BP_DEV Fe fe_neg(const Fe& a) {
  Fe r;
  uint32_t bw = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = __builtin_subc(F::mod(i), a.v[i], bw, &bw);
  return fe_select(fe_is_zero(a), a, r);
}

template <class F>
BP_DEV Fe fe_dbl(const Fe& a) {
  Fe s;
  uint32_t carry = a.v[7] >> 31;
#pragma unroll
  for (int i = 7; i > 0; i--) s.v[i] = (a.v[i] << 1) | (a.v[i - 1] >> 31);
  s.v[0] = a.v[0] << 1;
  return fe_reduce_once<F>(s, carry);
}

// Montgomery product a*b*2^-256 mod p, portable form (host builds of this header
// and the reference for the device form below).  CIOS, 8 outer steps.
template <class F>
BP_DEV Fe fe_mul_portable(const Fe& a, const Fe& b) {
  uint32_t t[9];
#pragma unroll
  for (int i = 0; i < 9; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t c = 0;
    const uint32_t bi = b.v[i];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      uint64_t x = (uint64_t)a.v[j] * bi + t[j] + c;
      t[j] = (uint32_t)x;
      c = x >> 32;
    }
    uint64_t top = (uint64_t)t[8] + c;
    const uint32_t m = t[0] * F::kInv32;
    c = ((uint64_t)m * F::mod(0) + t[0]) >> 32;
#pragma unroll
    for (int j = 1; j < 8; j++) {
      uint64_t x = (uint64_t)m * F::mod(j) + t[j] + c;
      t[j - 1] = (uint32_t)x;
      c = x >> 32;
    }
    top += c;
    t[7] = (uint32_t)top;
    t[8] = (uint32_t)(top >> 32);
  }
  Fe r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = t[i];
  return fe_reduce_once<F>(r, t[8]);
}

// ---- secp256k1 base field on plain residues (Secp256k1FqP) ---------------------------------
// p = 2^256 - c, c = 2^32 + 977: a value r + 2^256 t (r eight limbs, t < 2^40) is congruent to
// r + c t.  Adds c t, folds the carry that may leave limb 7 the same way, and subtracts p once
// if needed: the result is the canonical residue.
BP_DEV Fe fe_plain_secp_finish(const Fe& r, uint64_t t) {
  // c t = 977 t + 2^32 t with t < 2^38: 977 t = w_hi 2^32 + w_lo (w_hi < 2^16)
  const uint32_t t_lo = (uint32_t)t, t_hi = (uint32_t)(t >> 32);
  const uint64_t w = (uint64_t)t_lo * 977u;
  const uint32_t w_lo = (uint32_t)w, w_hi = (uint32_t)(w >> 32) + t_hi * 977u;
  Fe u = r;
  uint32_t c = 0, c2 = 0;
  u.v[0] = __builtin_addc(r.v[0], w_lo, 0u, &c);
  u.v[1] = __builtin_addc(r.v[1], w_hi, c, &c);
  u.v[1] = __builtin_addc(u.v[1], t_lo, 0u, &c2);
  u.v[2] = __builtin_addc(r.v[2], t_hi + c2, c, &c);
  // c t has about 70 bits, so a carry out of limb 2 has probability about 2^-27 and a value in
  // [p, 2^256) needs limbs 2..7 all ones (2^-192): both go to a branch that is practically never
  // taken instead of a carry chain and a selection on every product
  const uint32_t top = u.v[2] & u.v[3] & u.v[4] & u.v[5] & u.v[6] & u.v[7];
  if (__builtin_expect((c | (top == 0xffffffffu)) != 0, 0)) {
#pragma unroll
    for (int i = 3; i < 8; i++) u.v[i] = __builtin_addc(u.v[i], 0u, c, &c);
    // value = u + 2^256 c (c is 0 or 1) = u + c (2^32 + 977) mod p; s2 = u + (2^32 + 977)
    Fe s2;
    uint32_t q = 0;
    s2.v[0] = __builtin_addc(u.v[0], 977u, q, &q);
    s2.v[1] = __builtin_addc(u.v[1], 1u, q, &q);
#pragma unroll
    for (int i = 2; i < 8; i++) s2.v[i] = __builtin_addc(u.v[i], 0u, q, &q);
    u = fe_select((c | q) != 0, s2, u);
  }
  return u;
}

// portable plain multiply: schoolbook product, then the same fold (host builds; reference for
// the device form)
BP_DEV Fe fe_mul_plain_secp_portable(const Fe& a, const Fe& b) {
  uint32_t t[16];
#pragma unroll
  for (int i = 0; i < 16; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      uint64_t x = (uint64_t)a.v[j] * b.v[i] + t[i + j] + c;
      t[i + j] = (uint32_t)x;
      c = x >> 32;
    }
    t[i + 8] = (uint32_t)c;
  }
  Fe r;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    acc += (uint64_t)t[k] + (uint64_t)t[8 + k] * 977u + (k ? t[8 + k - 1] : 0u);
    r.v[k] = (uint32_t)acc;
    acc >>= 32;
  }
  return fe_plain_secp_finish(r, acc + t[15]);
}

#if defined(__HIP_DEVICE_COMPILE__)
// Device form: product scanning (column by column) in straight-line inline asm,
// generated per field by tools/gen_fe_mul.py.  A column's partial products are
// summed into a 96-bit accumulator {acc (64 bits), ovf (32 bits)}: each product is
// one v_mad_u64_u32 whose 64-bit addend is the accumulator itself, followed by one
// v_addc_co_u32 that counts the carry-out: two instructions per 32x32 product.
#include "bp_field_mul_gen.h"

template <class F>
BP_DEV Fe fe_mul(const Fe& a, const Fe& b) {
  return fe_mul_dev<F>(a, b);
}
#else
template <class F>
BP_DEV Fe fe_mul(const Fe& a, const Fe& b) {
  return fe_mul_portable<F>(a, b);
}
template <>
BP_DEV Fe fe_mul<Secp256k1FqP>(const Fe& a, const Fe& b) {
  return fe_mul_plain_secp_portable(a, b);
}
#endif

// a*b + c*d (Montgomery form in and out): one reduction for both products on the device
template <class F>
BP_DEV Fe fe_mul2(const Fe& a, const Fe& b, const Fe& c, const Fe& d) {
#if defined(__HIP_DEVICE_COMPILE__)
  return fe_mul2_dev<F>(a, b, c, d);
#else
  return fe_add<F>(fe_mul<F>(a, b), fe_mul<F>(c, d));
#endif
}

template <class F>
BP_DEV Fe fe_sqr(const Fe& a) {
  return fe_mul<F>(a, a);
}

// Montgomery form -> canonical integer: the reduction alone on the device (a multiplication by 1
// whose partial products are never issued), a multiplication by 1 elsewhere
template <class F>
BP_DEV Fe fe_from_mont(const Fe& a) {
#if defined(__HIP_DEVICE_COMPILE__)
  return fe_redc_dev<F>(a);
#else
  Fe one = fe_zero();
  one.v[0] = 1;
  return fe_mul<F>(a, one);
#endif
}

template <class F>
BP_DEV Fe fe_to_mont(const Fe& a) {
  Fe r2;
#pragma unroll
  for (int i = 0; i < 8; i++) r2.v[i] = F::r2(i);
  return fe_mul<F>(a, r2);
}

// a^(p-2): Fermat inverse; 0 maps to 0 (the convention of ark-ff's
// batch_inversion, which leaves zeros untouched).  Left-to-right sliding window of
// width 3 over the constant exponent: 255 squarings + ~64 multiplies.
template <class F>
__host__ __device__ __noinline__ Fe fe_inv(const Fe& a) {
  Fe a2 = fe_sqr<F>(a);
  Fe t1 = a, t3 = fe_mul<F>(a, a2), t5 = fe_mul<F>(t3, a2), t7 = fe_mul<F>(t5, a2);
  Fe r = fe_one<F>();
  int i = 255;
  while (i >= 0) {
    if (!((F::pm2(i >> 5) >> (i & 31)) & 1)) {
      r = fe_sqr<F>(r);
      i--;
      continue;
    }
    int j = i - 2 < 0 ? 0 : i - 2;
    while (!((F::pm2(j >> 5) >> (j & 31)) & 1)) j++;  // window [i..j] ends in a set bit
    uint32_t v = 0;
    for (int k = i; k >= j; k--) {
      r = fe_sqr<F>(r);
      v = (v << 1) | ((F::pm2(k >> 5) >> (k & 31)) & 1);
    }
    Fe m = fe_select(v == 1, t1, fe_select(v == 3, t3, fe_select(v == 5, t5, t7)));
    r = fe_mul<F>(r, m);
    i = j - 1;
  }
  return r;
}

}  // namespace bp
