#!/usr/bin/env python3
"""Generates bulletproofs_amd/csrc/bp_field_mul_gen.h: the device Montgomery multiply
of each supported prime field as straight-line gfx950 inline asm.

Product scanning, one column at a time.  A column's partial products are summed into
a 96-bit accumulator {acc (64-bit VGPR pair), ovf (32-bit)}: every 32x32 product is
one v_mad_u64_u32 whose addend is acc, followed by one v_addc_co_u32 that adds the
carry-out into ovf.  All products of a column that share an operand kind sit in ONE
asm statement (the compiler pads every asm statement with an s_nop, so few large
statements beat many small ones).  Modulus limbs equal to 0 are skipped, limbs equal
to 0xffffffff / 1 use inline constants, the rest are SGPR operands.

A run of top limbs p_s..p_7 that are all 0xffffffff (secp256k1: s = 2 for the base
field, 5 for the scalar field) is not multiplied out: m * sum_{j=s..7} p_j 2^(32j) =
m * (2^256 - 2^(32s)), so m_i is subtracted from column i+s and added to column i+8.
The accumulator is then a signed (two's complement) 96-bit value and the shift to the
next column sign-extends it.
"""
import os

FIELDS = {
    "Secp256k1Fq": (1 << 256) - (1 << 32) - 977,
    "Secp256k1Fr": 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141,
    "VestaFq": 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001,
    "VestaFr": 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001,
}


def limbs(v):
    return [(v >> (32 * i)) & 0xFFFFFFFF for i in range(8)]


class Asm:
    """one asm statement: in/out operands acc, ovf plus read-only inputs"""

    def __init__(self, fresh_ovf=False):
        self.lines = []
        self.inputs = []  # (constraint, expr)
        self.index = {}
        # fresh_ovf: the first carry count of this statement sets ovf (0 + 0 + carry) instead of
        # adding to it, so the caller need not clear it
        self.fresh_ovf = fresh_ovf

    def operand(self, constraint, expr):
        key = (constraint, expr)
        if key not in self.index:
            self.index[key] = 2 + len(self.inputs)
            self.inputs.append(key)
        return "%%%d" % self.index[key]

    def mac(self, a_ref, b_ref, carry=True):
        """carry=False: the caller knows that the 64-bit accumulator cannot overflow here"""
        self.lines.append("v_mad_u64_u32 %%0, vcc, %s, %s, %%0" % (a_ref, b_ref))
        if carry:
            if self.fresh_ovf:
                self.lines.append("v_addc_co_u32 %1, vcc, 0, 0, vcc")
                self.fresh_ovf = False
            else:
                self.lines.append("v_addc_co_u32 %1, vcc, 0, %1, vcc")

    def emit(self, out, indent="  "):
        if not self.lines:
            return
        assert len(self.inputs) + 2 <= 30, "asm operand limit"
        body = "\\n\\t".join(self.lines)
        ins = ", ".join('"%s"(%s)' % (c, e) for c, e in self.inputs)
        out.append('%sasm("%s"\n%s    : "+&v"(acc), "+&v"(ovf)\n%s    : %s\n%s    : "vcc");' % (indent, body, indent, indent, ins, indent))


class ColumnAsm:
    """one asm statement for a whole column whose accumulator and carry count are OUTPUTS: the
    first multiply-accumulate reads the incoming 64-bit value (or the constant 0) as its addend
    and writes a new register pair, the first carry count sets its register.  With in/out operands
    the compiler has to hand the statement a copy of the old carry count, which lives on as the
    upper half of the incoming accumulator: one v_mov per column that this form does not need."""

    def __init__(self, acc_in):
        self.lines = []
        self.inputs = []
        self.index = {}
        self.acc_in = acc_in  # C expression of the incoming value, None for zero
        self.first = True
        self.counted = False

    def operand(self, constraint, expr):
        key = (constraint, expr)
        if key not in self.index:
            self.index[key] = 2 + len(self.inputs)
            self.inputs.append(key)
        return "%%%d" % self.index[key]

    def mac(self, a_ref, b_ref, carry=True):
        if self.first:
            addend = self.operand("v", self.acc_in) if self.acc_in else "0"
            self.first = False
        else:
            addend = "%0"
        self.lines.append("v_mad_u64_u32 %%0, vcc, %s, %s, %s" % (a_ref, b_ref, addend))
        if carry:
            self.lines.append("v_addc_co_u32 %%1, vcc, 0, %s, vcc" % ("%1" if self.counted else "0"))
            self.counted = True

    def emit(self, out, acc_out, ovf_out, indent="  "):
        assert self.lines and self.counted, "a column needs a carry count to define its output"
        assert len(self.inputs) + 2 <= 30, "asm operand limit"
        body = "\\n\\t".join(self.lines)
        ins = ", ".join('"%s"(%s)' % (c, e) for c, e in self.inputs)
        out.append('%sasm("%s"\n%s    : "=&v"(%s), "=&v"(%s)\n%s    : %s\n%s    : "vcc");'
                   % (indent, body, indent, acc_out, ovf_out, indent, ins, indent))


def const_ref(asm, value):
    if value == 0xFFFFFFFF:
        return "-1"
    if value <= 64:
        return str(value)
    return asm.operand("s", "0x%08xu" % value)


def ones_run_start(pl):
    """first limb of the run of 0xffffffff limbs that ends at limb 7 (8 if the run is
    shorter than 3 limbs: then multiplying out is no more expensive)"""
    s = 8
    while s > 0 and pl[s - 1] == 0xFFFFFFFF:
        s -= 1
    return s if 8 - s >= 3 and s >= 1 else 8


def gen_field(name, p, redc=False, dual=False):
    """redc=True: the Montgomery reduction alone, a * 2^-256 mod p (the multiply with b = 1: a's
    limbs enter their columns directly, the 64 partial products are not issued).
    dual=True: (a*b + c*d) * 2^-256 mod p with ONE reduction: both products are summed column by
    column before the multiples of p are added (the shape of fold_scalars and of the terms of an
    inner product); the value before the final subtractions is below 3p"""
    pl = limbs(p)
    inv = (-pow(p, -1, 1 << 32)) % (1 << 32)
    run = ones_run_start(pl)
    signed = run < 8
    out = []
    out.append("// %s: p = 0x%064x" % (name, p))
    if redc:
        out.append("__device__ __forceinline__ Fe fe_redc_dev_%s(const Fe& a) {" % name)
    elif dual:
        out.append("__device__ __forceinline__ Fe fe_mul2_dev_%s(const Fe& a, const Fe& b, const Fe& c, const Fe& d) {" % name)
    else:
        out.append("__device__ __forceinline__ Fe fe_mul_dev_%s(const Fe& a, const Fe& b) {" % name)
    out.append("  uint64_t acc = 0;")
    out.append("  uint32_t ovf = 0;")
    out.append("  uint32_t m0, m1, m2, m3, m4, m5, m6, m7;")
    out.append("  Fe r;")
    for k in range(15):
        out.append("  // column %d" % k)
        ab = Asm()
        if redc:
            if k < 8:
                # unsigned accumulator: carry of the last column (a few bits above 2^32) + a_k cannot
                # overflow; a signed one (telescoped modulus) may be negative, and its carry counts
                ab.mac(ab.operand("v", "a.v[%d]" % k), "1", carry=signed)
        else:
            for i in range(8):
                j = k - i
                if 0 <= j <= 7:
                    ab.mac(ab.operand("v", "a.v[%d]" % i), ab.operand("v", "b.v[%d]" % j))
        ab.emit(out)
        if dual:
            cd = Asm()
            for i in range(8):
                j = k - i
                if 0 <= j <= 7:
                    cd.mac(cd.operand("v", "c.v[%d]" % i), cd.operand("v", "d.v[%d]" % j))
            cd.emit(out)
        mp = Asm()
        for i in range(8):
            j = k - i
            if 0 <= j <= 7 and not (k < 8 and i == k) and pl[j] != 0 and j < run:
                mp.mac(mp.operand("v", "m%d" % i), const_ref(mp, pl[j]))
        if signed and 0 <= k - 8 <= 7:
            mp.mac(mp.operand("v", "m%d" % (k - 8)), "1")
        mp.emit(out)
        if signed and 0 <= k - run <= 7:
            out.append("  {  // acc:ovf -= m%d" % (k - run))
            out.append("    uint32_t lo = (uint32_t)acc, hi = (uint32_t)(acc >> 32);")
            out.append('    asm("v_sub_co_u32 %0, vcc, %0, %3\\n\\tv_subbrev_co_u32 %1, vcc, 0, %1, vcc\\n\\tv_subbrev_co_u32 %2, vcc, 0, %2, vcc"')
            out.append('        : "+v"(lo), "+v"(hi), "+v"(ovf) : "v"(m%d) : "vcc");' % (k - run))
            out.append("    acc = lo | ((uint64_t)hi << 32);")
            out.append("  }")
        if k < 8:
            out.append("  m%d = (uint32_t)acc * 0x%08xu;" % (k, inv))
            last = Asm()
            last.mac(last.operand("v", "m%d" % k), const_ref(last, pl[0]))
            last.emit(out)
        else:
            out.append("  r.v[%d] = (uint32_t)acc;" % (k - 8))
        out.append("  acc = (acc >> 32) | ((uint64_t)ovf << 32);")
        out.append("  ovf = (uint32_t)((int32_t)ovf >> 31);" if signed and k >= run else "  ovf = 0;")
    if signed:
        out.append("  acc += m7;  // column 15")
    out.append("  r.v[7] = (uint32_t)acc;")
    out.append("  return fe_reduce_%s<%s>(r, (uint32_t)(acc >> 32));" % ("twice" if dual else "once", name))
    out.append("}")
    return out


def gen_secp_plain():
    """secp256k1 base field on plain residues (no Montgomery factor): p = 2^256 - c with
    c = 2^32 + 977, so the upper half of a product folds into the lower half by a multiplication
    with c.  The high columns (8..15) are summed first, without the carry of column 7; their
    limbs h_k then enter the low columns as 977*h_k + h_(k-1) next to the partial products, and what
    is left above column 7 (its carry plus h_7) is folded once more.  64 + 8 multiplications and
    8 additions by a multiply-accumulate with 1, against 88 of the Montgomery form."""
    name = "Secp256k1FqP"
    out = []
    out.append("// %s: residues mod p = 2^256 - 2^32 - 977 as plain integers" % name)
    out.append("__device__ __forceinline__ Fe fe_mul_dev_%s(const Fe& a, const Fe& b) {" % name)
    out.append("  uint64_t acc;")
    out.append("  uint32_t h0, h1, h2, h3, h4, h5, h6, h7;")
    out.append("  Fe r;")
    for k in range(8, 15):
        out.append("  // column %d" % k)
        ab = ColumnAsm(None if k == 8 else "acc")
        first = k == 8  # the accumulator starts at zero: one product cannot overflow it
        for i in range(8):
            j = k - i
            if 0 <= j <= 7:
                ab.mac(ab.operand("v", "a.v[%d]" % i), ab.operand("v", "b.v[%d]" % j), carry=not first)
                first = False
        out.append("  uint64_t acc%d;" % k)
        out.append("  uint32_t ovf%d;" % k)
        ab.emit(out, "acc%d" % k, "ovf%d" % k)
        out.append("  h%d = (uint32_t)acc%d;" % (k - 8, k))
        out.append("  acc = (acc%d >> 32) | ((uint64_t)ovf%d << 32);" % (k, k))
    out.append("  h7 = (uint32_t)acc;  // column 15: the upper half of a 512-bit product has 256 bits")
    for k in range(8):
        out.append("  // column %d, with 977*h%d%s" % (k, k, " + h%d" % (k - 1) if k else ""))
        ab = ColumnAsm("acc" if k else None)
        # the fold terms first: carry of the last column (< 2^37) + 977 h_k (< 2^42) + h_(k-1)
        # stays far below 2^64, so these two need no carry count
        ab.mac(ab.operand("v", "h%d" % k), ab.operand("s", "977u"), carry=False)
        if k:
            ab.mac(ab.operand("v", "h%d" % (k - 1)), "1", carry=False)
        for i in range(8):
            j = k - i
            if 0 <= j <= 7:
                ab.mac(ab.operand("v", "a.v[%d]" % i), ab.operand("v", "b.v[%d]" % j))
        out.append("  uint64_t acc%d;" % k)
        out.append("  uint32_t ovf%d;" % k)
        ab.emit(out, "acc%d" % k, "ovf%d" % k)
        out.append("  r.v[%d] = (uint32_t)acc%d;" % (k, k))
        out.append("  acc = (acc%d >> 32) | ((uint64_t)ovf%d << 32);" % (k, k))
    out.append("  return fe_plain_secp_finish(r, acc + h7);")
    out.append("}")
    return out


def main():
    out = ["// Generated by tools/gen_fe_mul.py -- do not edit.", "#pragma once",
           "// included by bp_field.h inside namespace bp, device compilation only", ""]
    for name, p in FIELDS.items():
        out += gen_field(name, p)
        out.append("")
        out += gen_field(name, p, redc=True)
        out.append("")
        if name.endswith("Fr"):
            out += gen_field(name, p, dual=True)
            out.append("")
    out += gen_secp_plain()
    out.append("")
    out.append("template <class F>\n__device__ __forceinline__ Fe fe_mul_dev(const Fe& a, const Fe& b);")
    for name in list(FIELDS) + ["Secp256k1FqP"]:
        out.append("template <>\n__device__ __forceinline__ Fe fe_mul_dev<%s>(const Fe& a, const Fe& b) {\n  return fe_mul_dev_%s(a, b);\n}" % (name, name))
    out.append("template <class F>\n__device__ __forceinline__ Fe fe_redc_dev(const Fe& a);")
    for name in FIELDS:
        out.append("template <>\n__device__ __forceinline__ Fe fe_redc_dev<%s>(const Fe& a) {\n  return fe_redc_dev_%s(a);\n}" % (name, name))
    out.append("template <class F>\n__device__ __forceinline__ Fe fe_mul2_dev(const Fe& a, const Fe& b, const Fe& c, const Fe& d);")
    for name in FIELDS:
        if name.endswith("Fr"):
            out.append("template <>\n__device__ __forceinline__ Fe fe_mul2_dev<%s>(const Fe& a, const Fe& b, const Fe& c, const Fe& d) {\n  return fe_mul2_dev_%s(a, b, c, d);\n}" % (name, name))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bulletproofs_amd", "csrc", "bp_field_mul_gen.h")
    open(path, "w").write("\n".join(out) + "\n")
    print("wrote", os.path.normpath(path))


if __name__ == "__main__":
    main()
