#!/usr/bin/env python3
"""Generate accumulation_amd/csrc/fp_mul_gfx950.h: Montgomery multiplication for each field as a
column-wise (Comba) schedule of v_mad_u64_u32 + v_addc_co_u32, one inline-asm statement per column.

Why generated asm: hipcc turns the portable CIOS loop (fp.h:fe_mul_ref) into 88 MADs + 144 64-bit adds +
316 v_mov per Pallas multiplication; the hand schedule below needs 2 VALU instructions per limb product
(the MAD and one carry add).  Hazard honoured: on gfx950 a VALU write of an SGPR pair (the MAD's carry-out)
needs 2 wait states before a VALU reads it as carry-in, so each v_addc trails its v_mad by >= 2 slots
(three rotating SGPR pairs), with s_nop padding only in the 1- and 2-product columns.

Column k accumulates  sum_{i+j=k} a_i*b_j + sum_{i<min(k,L)} m_i*p_{k-i}  into (acc = 64-bit pair, c2 = carry
count).  Modulus limbs equal to 0 are skipped at generation time (Pallas: 4 of 8); p_0 is handled by the
column epilogue.  For k < L the epilogue derives m_k = -acc.lo * INV... (INV = -p^-1 mod 2^32), adds m_k*p_0 so
the low word becomes 0 and shifts the accumulator down one word; for k >= L it emits result word k-L.
"""
import sys

FIELDS = {
    # name: (L, modulus as int, INV)
}

def field(name, L, mod):
    inv = (-pow(mod, -1, 1 << 32)) % (1 << 32)
    FIELDS[name] = (L, mod, inv)

field("PallasFq", 8, 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001)
field("PallasFr", 8, 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001)
field("Bls12381Fq", 12, 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB)
field("Bls12381Fr", 8, 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001)

MAX_TERMS = 11  # terms per asm statement (operand limit 30: 2*terms + acc + c2 + 3 sgpr temps)


def emit_column_stmt(terms, first):
    """terms: list of (x_expr, y_expr) C expressions (u32 VGPR or SGPR-constant operands).
    Returns C source of one asm volatile statement doing acc += sum x*y with carries into c2."""
    n = len(terms)
    ops_in = []
    names = {}
    def operand(expr, cons):
        key = (expr, cons)
        if key not in names:
            names[key] = f"i{len(names)}"
            ops_in.append(f'[{names[key]}] "{cons}"({expr})')
        return f"%[{names[key]}]"
    lines = []
    sg = ["%[s0]", "%[s1]", "%[s2]"]
    mads = []
    for t, (x, xc, y, yc) in enumerate(terms):
        mads.append(f"v_mad_u64_u32 %[acc], {sg[t % 3]}, {operand(x, xc)}, {operand(y, yc)}, %[acc]")
    addcs = [f"v_addc_co_u32 %[c2], vcc, 0, %[c2], {sg[t % 3]}" for t in range(n)]
    if first:  # c2 is write-only here: the first carry add initialises it (saves a v_mov per column)
        addcs[0] = f"v_addc_co_u32 %[c2], vcc, 0, 0, {sg[0]}"
    # schedule: M1 M2 M3 A1 M4 A2 ... ; pad for short columns
    seq = []
    if n == 1:
        seq = [mads[0], "s_nop 1", addcs[0]]
    elif n == 2:
        seq = [mads[0], mads[1], "s_nop 0", addcs[0], addcs[1]]
    else:
        tagged = [("m", 0), ("m", 1), ("m", 2)]
        nxt = 3
        for t in range(n):
            tagged.append(("a", t))
            if nxt < n:
                tagged.append(("m", nxt))
                nxt += 1
        # ordering check: addc t comes >= 3 slots after mad t (2 wait states) and before mad t+3 reuses the pair
        pos = {x: i for i, x in enumerate(tagged)}
        for t in range(n):
            assert pos[("a", t)] - pos[("m", t)] >= 3, (n, t)
            if t + 3 < n:
                assert pos[("m", t + 3)] > pos[("a", t)]
        seq = [mads[t] if kind == "m" else addcs[t] for kind, t in tagged]
    body = "\\n\\t".join(seq)
    c2c = '"=&v"' if first else '"+v"'
    outs = f'[acc] "+v"(acc), [c2] {c2c}(c2), [s0] "=&s"(s0), [s1] "=&s"(s1), [s2] "=&s"(s2)'
    return f'  asm volatile("{body}"\n               : {outs}\n               : {", ".join(ops_in)}\n               : "vcc");\n'


def gen_field(name, T=1):
    """T = 1: fe_mul(a, b).  T = 2, 3: fe_dot2 / fe_dot3 -- sum_t a_t b_t with ONE Montgomery reduction: the T products of a
    column are accumulated before the column's m_k is derived; the value before the final subtractions is below
    (T p^2 + 2^(32 L) p) / 2^(32 L), i.e. needs ceil(T p / 2^(32 L)) conditional subtractions."""
    L, mod, inv = FIELDS[name]
    p = [(mod >> (32 * i)) & 0xFFFFFFFF for i in range(L)]
    out = []
    if T == 1:
        out.append(f"// ---- {name}: L = {L}, INV = 0x{inv:08x} ----")
        out.append("template <>")
        out.append(f"AMSM_DEV Fe<{name}> fe_mul<{name}>(const Fe<{name}>& a, const Fe<{name}>& b) {{")
        pairs = [("a", "b")]
    else:
        pairs = [(f"a{t}", f"b{t}") for t in range(T)]
        args = ", ".join(f"const Fe<{name}>& {x}, const Fe<{name}>& {y}" for x, y in pairs)
        out.append(f"// ---- {name}: sum of {T} products, one reduction ----")
        out.append("template <>")
        out.append(f"AMSM_DEV Fe<{name}> fe_dot{T}<{name}>({args}) {{")
    out.append("  u64 acc = 0, s0, s1, s2;")
    out.append("  u32 c2;")
    out.append(f"  u32 m[{L}];")
    out.append(f"  Fe<{name}> r;")
    # modulus constants as SGPR operands (wave-uniform), skip 0 and handle p0 in epilogue
    for k in range(2 * L):
        terms = []
        for x, y in pairs:
            for i in range(L):
                j = k - i
                if 0 <= j < L:
                    terms.append((f"{x}.v[{i}]", "v", f"{y}.v[{j}]", "v"))
        for i in range(min(k, L)):
            j = k - i
            if 1 <= j < L and p[j] != 0:
                terms.append((f"m[{i}]", "v", f"0x{p[j]:08x}u", "s"))
        if k == 2 * L - 1:
            assert not terms or all(t[0].startswith("m[") for t in terms) or True
        # split into statements
        for s in range(0, len(terms), MAX_TERMS):
            chunk = terms[s:s + MAX_TERMS]
            out.append(emit_column_stmt(chunk, s == 0).rstrip("\n"))
        if k < L:
            # m_k = lo * INV; acc += m_k * p0 (p0 is odd; lo + m_k*p0 == 0 mod 2^32); shift down
            out.append("  {")
            out.append("    u32 lo = (u32)acc, hi = (u32)(acc >> 32);")
            if inv == 0xFFFFFFFF:
                out.append(f"    m[{k}] = 0u - lo;")
            else:
                out.append(f"    m[{k}] = lo * 0x{inv:08x}u;")
            if p[0] == 1:
                # lo + m = 0 or 2^32: carry = (lo != 0)
                out.append("    u32 cy = lo != 0 ? 1u : 0u;")
                out.append("    u32 nlo = hi + cy;")
                out.append("    u32 nhi = c2 + (nlo < cy ? 1u : 0u);")
            else:
                out.append(f"    u64 t = (u64)m[{k}] * 0x{p[0]:08x}u + lo;  // low word is 0 by construction")
                out.append("    u32 cy = (u32)(t >> 32);")
                out.append("    u32 nlo = hi + cy;")
                out.append("    u32 nhi = c2 + (nlo < cy ? 1u : 0u);")
            out.append("    acc = ((u64)nhi << 32) | nlo;")
            out.append("  }")
        else:
            out.append(f"  r.v[{k - L}] = (u32)acc;")
            if k < 2 * L - 1:
                out.append("  acc = ((u64)c2 << 32) | (u32)(acc >> 32);")
    n_sub = -(-(T * mod) // (1 << (32 * L)))  # ceil(T p / R)
    assert (T * mod * mod + (mod << (32 * L))) >> (32 * L) < (1 << (32 * L + 32))
    out.append(f"  fe_cond_sub<{name}>(r, (u32)(acc >> 32));")
    for _ in range(n_sub - 1):
        out.append(f"  fe_cond_sub<{name}>(r, 0);")
    out.append("  return r;")
    out.append("}")
    return "\n".join(out)


def main():
    hdr = []
    hdr.append("// GENERATED by tools/gen_fp_asm.py -- do not edit.  Comba-Montgomery multiplication for gfx950:")
    hdr.append("// v_mad_u64_u32 + v_addc_co_u32 per limb product, one asm statement per column (see the generator).")
    hdr.append("#pragma once")
    hdr.append("namespace amsm {")
    for name in FIELDS:
        hdr.append(gen_field(name))
        hdr.append("")
    for name in ("PallasFr", "Bls12381Fr"):  # the scalar-field vector kernels (vec_kernels.h)
        for T in (2, 3):
            hdr.append(gen_field(name, T))
            hdr.append("")
    hdr.append("}  // namespace amsm")
    sys.stdout.write("\n".join(hdr) + "\n")


if __name__ == "__main__":
    main()
