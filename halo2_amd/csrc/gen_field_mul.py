#!/usr/bin/env python3
"""Generates field_mul.inc: the fully unrolled product-scanning Montgomery multiply for the Pasta
fields as 16 per-column inline-asm statements (gfx950).

Why generated: hipcc schedules every `asm` statement as one opaque instruction and pads one wait
state after each; one statement per *column* (instead of per partial product) keeps the carry chain
  v_mad_u64_u32 acc, vcc, a_i, b_j, acc ; v_addc_co_u32 hi, vcc, 0, hi, vcc
back to back (16 pads instead of 96).

Column k accumulates  sum_{i+j=k} a_i*b_j + sum_i m_i*p_{k-i}  into {lo:mid (aligned pair), hi};
p = [1, P1, P2, P3, 0, 0, 0, 2^30];  m_k = -lo  (because -p^-1 = -1 mod 2^32).

Run:  python halo2_amd/csrc/gen_field_mul.py   (rewrites field_mul.inc next to it)
"""
import os

MAC = "v_mad_u64_u32 %[acc], vcc, {x}, {y}, %[acc]\\n\\tv_addc_co_u32 %[hi], vcc, 0, %[hi], vcc\\n\\t"


def column(k, square=False):
    ins = []          # (name, constraint, c-expr)
    body = ""
    names = {}

    def op(expr, cons="v"):
        if expr not in names:
            nm = f"i{len(names)}"
            names[expr] = nm
            ins.append((nm, cons, expr))
        return f"%[{names[expr]}]"

    for i in range(8):
        j = k - i
        if 0 <= j < 8:
            body += MAC.format(x=op(f"a.v[{i}]"), y=op(f"b.v[{j}]"))
    for (d, const) in ((1, "K1"), (2, "K2"), (3, "K3"), (7, "K7")):
        i = k - d
        if 0 <= i < 8:
            body += MAC.format(x=op(f"m{i}"), y=op(const, "s"))
    in_s = ", ".join(f'[{n}] "{c}"({e})' for n, c, e in ins)
    body = body[:-4]  # drop trailing \n\t
    code = f'    asm("{body}"\n        : [acc] "+v"(acc), [hi] "+v"(hi)\n        : {in_s}\n        : "vcc");\n'
    if k < 8:
        # m_k = -lo ; {lo,mid,hi} = {mid + (lo != 0), hi + carry, 0}   (glue left to the compiler:
        # AMDGPU inline asm cannot name the halves of an aligned 64-bit VGPR pair operand)
        code += (f"    {{ u32 lo = (u32)acc, mid = (u32)(acc >> 32), c1; m{k} = 0u - lo;\n"
                 f"      u32 nl = __builtin_addc(mid, 0u, (u32)(lo != 0), &c1);\n"
                 f"      acc = ((u64)(hi + c1) << 32) | nl; hi = 0; }}\n")
    else:
        code += (f"    r.v[{k - 8}] = (u32)acc; acc = (acc >> 32) | ((u64)hi << 32); hi = 0;\n")
    return code


def block_multiplier():
    """One asm statement for the whole multiplication.  The 96-bit column accumulator lives in pinned
    VGPRs v8:v9 (lo:mid, an aligned pair as v_mad_u64_u32 needs) and v10 (hi), declared as clobbers, so the
    column shift can address the halves directly -- AMDGPU inline asm cannot name the halves of a 64-bit
    operand.  Quotient digits m_k live in the output registers r_k until r_k itself is produced (column k+8;
    m_k is last read in column k+7)."""
    L = []
    first_mac_of_mul = True
    for k in range(16):
        macs = []
        for i in range(8):
            j = k - i
            if 0 <= j < 8:
                macs.append((f"%[a{i}]", f"%[b{j}]"))
        for (d, const) in ((1, "%[k1]"), (2, "%[k2]"), (3, "%[k3]"), (7, "%[k7]")):
            i = k - d
            if 0 <= i < 8:
                macs.append((f"%[r{i}]", const))
        for n_, (x, y) in enumerate(macs):
            if first_mac_of_mul:
                L.append(f"v_mad_u64_u32 v[8:9], vcc, {x}, {y}, 0")
                L.append("v_mov_b32 v10, 0")
                first_mac_of_mul = False
            else:
                L.append(f"v_mad_u64_u32 v[8:9], vcc, {x}, {y}, v[8:9]")
                # the first product of a column (re)creates hi from the carry alone
                L.append("v_addc_co_u32 v10, vcc, 0, 0, vcc" if n_ == 0 else "v_addc_co_u32 v10, vcc, 0, v10, vcc")
        if k < 8:
            L += [f"v_sub_u32 %[r{k}], 0, v8",          # m_k = -lo
                  "v_cmp_ne_u32 vcc, 0, v8",             # adding m_k * p0 = m_k clears lo, carries iff lo != 0
                  "v_addc_co_u32 v8, vcc, 0, v9, vcc",
                  "v_addc_co_u32 v9, vcc, 0, v10, vcc"]
        else:
            L.append(f"v_mov_b32 %[r{k - 8}], v8")
            if k < 15:
                L += ["v_mov_b32 v8, v9", "v_mov_b32 v9, v10"]
    body = "\\n\\t".join(L)
    outs = ", ".join(f'[r{i}] "=&v"(r.v[{i}])' for i in range(8))
    ins = ", ".join([f'[a{i}] "v"(a.v[{i}])' for i in range(8)] + [f'[b{i}] "v"(b.v[{i}])' for i in range(8)] +
                    ['[k1] "s"(K1)', '[k2] "s"(K2)', '[k3] "s"(K3)', '[k7] "s"(K7)'])
    return (f'    fe r;\n    const u32 K1 = Mod<F>::P1, K2 = Mod<F>::P2, K3 = Mod<F>::P3, K7 = P7;\n'
            f'    asm("{body}"\n        : {outs}\n        : {ins}\n        : "vcc", "v8", "v9", "v10");\n')


def scheduled_multiplier():
    """Whole multiplication as ONE asm statement, list-scheduled by this script so that the gfx940/gfx950
    hazard "VALU writes an SGPR (a carry) -> VALU reads it: 2 wait states" is honoured with real work
    instead of s_nop wherever possible (hipcc pads this hazard in its own code but does not look inside
    asm strings).  Carries rotate through three SGPR pairs so that a product's carry add can be issued two
    instructions after its multiply.  Accumulator {lo:mid, hi} lives in pinned v8:v9, v10."""
    CAR = ["s[40:41]", "s[42:43]", "s[44:45]"]
    CS, CG = "s[46:47]", "vcc"
    ops = []          # dict(text, deps=[(op index, min distance)], prio)

    def add(text, deps, prio=0):
        ops.append(dict(text=text, deps=list(deps), prio=prio))
        return len(ops) - 1

    last_acc = None    # op that last wrote v[8:9]
    last_hi = None     # op that last wrote v10
    ncar = 0
    car_reader = [None, None, None]   # last op that READ each rotating carry pair (a multiply may not redefine it earlier)
    for k in range(16):
        macs = []
        for i in range(8):
            j = k - i
            if 0 <= j < 8:
                macs.append((f"%[a{i}]", f"%[b{j}]", None))
        for (d, const) in ((1, "%[k1]"), (2, "%[k2]"), (3, "%[k3]"), (7, "%[k7]")):
            i = k - d
            if 0 <= i < 8:
                macs.append((f"%[r{i}]", const, ("m", i)))
        a_ops = []
        for n_, (x, y, _) in enumerate(macs):
            ci = ncar % 3
            car = CAR[ci]
            ncar += 1
            deps = [(last_acc, 1)] if last_acc is not None else []
            if car_reader[ci] is not None:
                deps.append((car_reader[ci], 1))
            if k == 0 and n_ == 0:
                m = add(f"v_mad_u64_u32 v[8:9], {car}, {x}, {y}, 0", deps, prio=2)
                last_acc = m
                last_hi = add("v_mov_b32 v10, 0", [], prio=0)   # no carry possible into an empty accumulator
                continue
            m = add(f"v_mad_u64_u32 v[8:9], {car}, {x}, {y}, v[8:9]", deps, prio=2)
            last_acc = m
            hdeps = [(m, 3)] + ([(last_hi, 1)] if last_hi is not None else [])
            fresh = (n_ == 0 and k > 0)   # first product of a column recreates hi from its carry alone
            a = add(f"v_addc_co_u32 v10, {car}, 0, {'0' if fresh else 'v10'}, {car}", hdeps, prio=1)
            last_hi = a
            car_reader[ci] = a
            a_ops.append(a)
        if k < 8:
            sdep = [(last_acc, 1)]
            s_ = add(f"v_sub_co_u32 %[r{k}], {CS}, 0, v8", sdep, prio=2)              # m_k = -lo, borrow = (lo != 0)
            g1 = add(f"v_addc_co_u32 v8, {CG}, 0, v9, {CS}", [(s_, 3)], prio=2)       # lo' = mid + borrow
            g2 = add(f"v_addc_co_u32 v9, {CG}, 0, v10, {CG}", [(g1, 3), (last_hi, 1)], prio=2)  # mid' = hi + carry
            last_acc = g2
            last_hi = g2      # v10 is consumed; the next column's first carry add rewrites it
        else:
            r_ = add(f"v_mov_b32 %[r{k - 8}], v8", [(last_acc, 1)], prio=2)
            if k < 15:
                m1 = add("v_mov_b32 v8, v9", [(r_, 1)], prio=2)
                m2 = add("v_mov_b32 v9, v10", [(m1, 1), (last_hi, 1)], prio=2)
                last_acc = m2
                last_hi = m2
    # list scheduling
    emitted_at = {}
    order = []
    slot = 0
    remaining = set(range(len(ops)))
    while remaining:
        ready = []
        for i in sorted(remaining):
            ok = True
            for (d, dist) in ops[i]["deps"]:
                if d not in emitted_at or slot - emitted_at[d] < dist:
                    ok = False
                    break
            if ok:
                ready.append(i)
        # keep program order among equal priority (dependencies are chains), favour the multiply chain
        if ready:
            # do not let an op overtake an earlier op it shares a destination with: ops only ever depend forward,
            # and v10 / v8 / v9 writers are chained through deps, so any ready op is safe
            best = max(ready, key=lambda i: (ops[i]["prio"], -i))
            order.append(ops[best]["text"])
            emitted_at[best] = slot
            remaining.remove(best)
        else:
            order.append("s_nop 0")
        slot += 1
    body = "\\n\\t".join(order)
    outs = ", ".join(f'[r{i}] "=&v"(r.v[{i}])' for i in range(8))
    ins = ", ".join([f'[a{i}] "v"(a.v[{i}])' for i in range(8)] + [f'[b{i}] "v"(b.v[{i}])' for i in range(8)] +
                    ['[k1] "s"(K1)', '[k2] "s"(K2)', '[k3] "s"(K3)', '[k7] "s"(K7)'])
    nops = sum(1 for t in order if t.startswith("s_nop"))
    code = (f'    // {len(order)} instructions, {nops} of them s_nop\n'
            f'    fe r;\n    const u32 K1 = Mod<F>::P1, K2 = Mod<F>::P2, K3 = Mod<F>::P3, K7 = P7;\n'
            f'    asm("{body}"\n        : {outs}\n        : {ins}\n'
            f'        : "vcc", "v8", "v9", "v10", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");\n')
    return code, len(order), nops


def main():
    out = ["// GENERATED by gen_field_mul.py -- do not edit.\n",
           "// Included inside  template <int F> fe fe_mul_asm(const fe &a, const fe &b).\n",
           "    u32 m0, m1, m2, m3, m4, m5, m6, m7;\n",
           "    fe r;\n",
           "    u64 acc = 0;\n",
           "    u32 hi = 0;\n",
           "    const u32 K1 = Mod<F>::P1, K2 = Mod<F>::P2, K3 = Mod<F>::P3, K7 = P7;\n"]
    for k in range(16):
        out.append(column(k))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "field_mul.inc")
    open(path, "w").write("".join(out))
    print("wrote", path)
    path2 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "field_mul_blk.inc")
    open(path2, "w").write("// GENERATED by gen_field_mul.py -- do not edit.\n"
                           "// Included inside  template <int F> fe fe_mul_blk(const fe &a, const fe &b).\n" + block_multiplier())
    print("wrote", path2)
    path4 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "field_mul_sched.inc")
    code4, n_ins, n_nop = scheduled_multiplier()
    open(path4, "w").write("// GENERATED by gen_field_mul.py -- do not edit.\n"
                           "// Included inside  template <int F> fe fe_mul_sched(const fe &a, const fe &b).\n" + code4)
    print("wrote", path4, n_ins, "instructions,", n_nop, "s_nop")


if __name__ == "__main__":
    main()
