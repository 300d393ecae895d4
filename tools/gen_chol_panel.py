#!/usr/bin/env python3
"""Writes dvm_slam_amd/csrc/chol_panel.inc: the 16-column panel of k_chol_diag as ONE straight-line, hand-ordered instruction
stream.

Why a generator.  The panel runs on a single wave that owns its SIMD.  Such a wave issues an FP64 instruction that depends on
the previous one every ~8.3 cycles and an independent one every ~5.3 (profiles/r02_valu_issue2.jsonl, one wave per SIMD); the
compiler's scheduler does not know that and emits a column's pivot chain (rsq, two Newton steps, scale, broadcast: ten
dependent steps) as one run and the updates of the next pivot column as a second dependent run.  Here every statement is
pinned where this file puts it -- chain step, one or two fillers, chain step, ... -- so that the fillers (the deferred rank-1
updates with the previous column, the LDS traffic) issue in the chain's shadow, and the stream carries nothing the
factorisation does not need:
  * no sqrt for the diagonal entry: L_jj itself is read by nobody (the inverse uses 1 / L_jj, the trailing updates and the
    L^-1 assembly the entries below the diagonal); the diagonal slot keeps d * rsqrt(d), L_jj to an ulp or two;
  * no per-column positivity test: a pivot <= 0 turns 1 / L_jj into inf / NaN, which every later pivot inherits (each one
    subtracts the square of an entry scaled by it) -- the LAST column's 1 / L_jj is tested once;
  * 1 / sqrt(d) by v_rsq_f64 and one cubic correction instead of two Newton steps (see column_block).

Pinning is done with empty `asm volatile` statements: the asm behind statement i names what statement i produced and an input
of statement i + 2 as "+v" / "+s" operands (the value passes "through" the asm); volatile asms keep their order and the data
dependences do the rest.  Not an input of statement i + 1: the hazard recogniser treats every register an inline asm defines as
written with dst_sel and puts an s_nop in front of a VALU instruction that reads it right behind the asm.  Each statement is
held in a window of +-1 position, which is enough (check the ISA: chain step and fillers alternate as written).  No
instruction is emitted for the asms.

Arithmetic: every a[c] receives the same fma's in the same order as the rolled loop this replaces (same bits off the diagonal).

Usage: python tools/gen_chol_panel.py [updates per slot, default 2]   (rewrites the .inc, which is committed)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "dvm_slam_amd", "csrc", "chol_panel.inc")
N = 16


class Stmt:
    def __init__(self, code, outs=(), ins=()):
        self.code = code        # C++ text
        self.outs = list(outs)  # (name, 'v' | 's'): produced here, may be named by the asm behind it
        self.ins = list(ins)    # (name, 'v' | 's'): register inputs, in the order of preference for pinning


def V(v):
    return (v, "v")


def S(v):
    return (v, "s")


def column_block(j, spread):
    """Statements of column j in issue order: the chain steps, fillers in the slots behind them."""
    d, y, t, z, q, l, n, x = f"d{j}", f"y{j}", f"t{j}", f"z{j}", f"q{j}", f"a[{j}]", f"n{j}", f"x{j}"   # l: the finished column lives on in a[j]
    # 1 / sqrt(d): v_rsq_f64 (2^-24) and ONE cubic step, y += y r (1/2 + 3/8 r), r = 1 - d y^2: 1.24 ulp at most, exactly what two
    # Newton steps reach (tools/rsq_acc.hip, 4 M arguments over 26 decades), in five instructions four deep instead of
    # seven six deep
    ch = [
        ("rsq", Stmt(f"double {y} = __builtin_amdgcn_rsq({d});", [V(y)], [V(d)])),
        ("t", Stmt(f"double {t} = (-{d}) * {y};", [V(t)], [V(y), V(d)])),
        ("r", Stmt(f"{t} = __builtin_fma({t}, {y}, 1.0);", [V(t)], [V(t), V(y)])),
        ("z", Stmt(f"double {z} = {y} * {t};", [V(z)], [V(t), V(y)])),
        ("y", Stmt(f"{y} = __builtin_fma({z}, {q}, {y});", [V(y)], [V(z), V(q), V(y)])),
        ("l", Stmt(f"{l} = {l} * {y};", [V(l)], [V(y), V(l)])),
    ]
    # (the next pivot from its PRE-update value, broadcast off the chain, and the square of l_{j+1,j}: broadcasting the updated
    #  a[j+1] instead saves two instructions per column but puts a second cross-lane hop on the chain -- 100-200 cycles per
    #  panel slower)
    if j + 1 < N:
        ch.append(("n", Stmt(f"double {n} = bcast_lane({l}, {j + 1});", [S(n)], [V(l)])))
        ch.append(("d", Stmt(f"double d{j + 1} = __builtin_fma(-{n}, {n}, {x});", [V(f"d{j + 1}")], [S(n), S(x)])))
    slots = {name: [] for name, _ in ch}
    last = ch[-1][0]
    slots["z"].append(Stmt(f"double {q} = __builtin_fma(0.375, {t}, 0.5);", [V(q)], [V(t)]))

    # Deferred updates, a[c] -= L_.p * L_cp.  Column p's broadcasts are read from LDS behind its store (slot "n" of block p), so
    # its updates start behind step "r" of block p + 1 -- the next pivot column c = p + 2 first: its broadcast for the pivot is
    # taken behind step "y" -- and run on through the first slots of block p + 2 (a[c], c >= p + 3, is not needed earlier, and
    # the updates of one a[c] stay in column order: block p + 2 applies column p + 1 from slot "r" on).  Where exactly a filler
    # sits matters little: this wave issues in order and nothing overlaps, a filler between two dependent chain steps saves the
    # ~3 cycles of the dependent issue (8.3 instead of 5.3), that is all.
    def upd(p, c):
        pv = f"p{p}_{c}" if (c == p + 2 and c & 1) else f"p{p}_{c & ~1}.{'xy'[c & 1]}"   # pairs from p + 2 on; an odd p + 2 came alone
        return Stmt(f"a[{c}] = __builtin_fma(-a[{p}], {pv}, a[{c}]);", [V(f"a[{c}]")], [V(f"a[{c}]"), V(f"a[{p}]")])

    def placement(p):
        one = [(p + 1, "r"), (p + 1, "y"), (p + 1, "l"), (p + 1, "n"), (p + 2, "rsq"), (p + 2, "rsq"), (p + 2, "t")]
        pos = one + one[1:] + one[1:]
        out = []
        for i, c in enumerate(range(p + 2, N)):
            b_, s_ = pos[i]
            if b_ >= N:                       # no block p + 2 for the last columns: everything in block p + 1
                b_, s_ = p + 1, ["r", "y", "l", "n"][i % 4]
            out.append((b_, s_, c))
        return out
    for p in (j - 1, j - 2):
        if p < 0:
            continue
        for b_, s_, c in placement(p):
            if b_ == j:
                slots[s_ if s_ in slots else last].append(upd(p, c))
    # the next pivot's pre-update value, broadcast off the chain
    if j + 1 < N:
        slots["y"].append(Stmt(f"double {x} = bcast_lane(a[{j + 1}], {j + 1});", [S(x)], [V(f"a[{j + 1}]")]))
    # behind the broadcast of l_{j+1,j}: publish the column, fetch its broadcasts for the next blocks' updates, then the
    # immediate update of the next pivot column.  (1 / L_jj is not published: the one consumer left, the inversion of sub-block 0
    # beside panel 1, divides by the L_jj it finds in the tile -- sixteen divisions on an idle wave against a store per column
    # here.)
    pub = "n" if j + 1 < N else last
    slots[pub].append(Stmt(f"Pcol[{j}][lane] = {l};", [], [V(l)]))
    # (the first broadcast alone when it would be the upper half of a pair: a pair whose lower half is dead invites the register
    #  allocator to reuse that half while the read is in flight, and the chain then waits for the LDS)
    if (j + 2) & 1 and j + 2 < N:
        slots[pub].append(Stmt(f"const double p{j}_{j + 2} = Pcol[{j}][{j + 2}];"))
    for c in range((j + 3) & ~1, N, 2):
        slots[pub].append(Stmt(f"const v2f64 p{j}_{c} = *(const lds_v2f64*)&Pcol[{j}][{c}];"))
    if j + 1 < N:
        slots["n"].append(Stmt(f"a[{j + 1}] = __builtin_fma(-{l}, {n}, a[{j + 1}]);", [V(f"a[{j + 1}]")], [V(f"a[{j + 1}]"), V(l), S(n)]))
    seq = []
    for name, c in ch:
        seq.append(c)
        seq.extend(slots[name])
    return seq


def emit(seq, lag=3):
    """The asm behind statement i names (a) what statement i produced, unless one of the next lag - 1 statements reads it (the
    data dependence then orders them), and (b) an input of statement i + lag that none of the statements in between touches."""
    declared = {"a[%d]" % c for c in range(N)} | {"d0"}
    lines = []
    for i, st in enumerate(seq):
        lines.append("  " + st.code)
        for name, _ in st.outs:
            declared.add(name)
        between = seq[i + 1:i + lag]
        touched = {v for q in between for v, _ in q.ins} | {v for q in between for v, _ in q.outs}
        ops = [(v, k) for v, k in st.outs if v not in touched]
        if i + lag < len(seq):
            for v, k in seq[i + lag].ins:
                if v in declared and v not in touched and (v, k) not in ops:
                    ops.append((v, k))
                    break
        if ops:
            lines.append('  asm volatile("" : ' + ", ".join(f'"+{k}"({v})' for v, k in ops) + ");")
    return lines


def main():
    out_path = OUT
    argv = sys.argv[1:]
    if "--out" in argv:                       # tests/test_host_logic.py regenerates into a temporary file and compares
        out_path = argv[argv.index("--out") + 1]
        del argv[argv.index("--out"):argv.index("--out") + 2]
    spread = int(argv[0]) if argv else 2
    seq = []
    for j in range(N):
        seq.extend(column_block(j, spread))
    out = [
        "// GENERATED by tools/gen_chol_panel.py -- do not edit; the generator's header explains the schedule.",
        "// In: double a[16] (the panel row of this lane), d0 (pivot a_00, wave-uniform), lane, Pcol.",
        "// Out: a[] = the row of L (diagonal slot: d * rsqrt(d)), Pcol published column by column, bad.",
        "{",
    ]
    out += emit(seq, int(os.environ.get('PANEL_LAG', '4')))
    out.append(f"  bad = !(y{N - 1} > 0.0 && y{N - 1} < __builtin_inf());   // a pivot <= 0 anywhere in the panel ends here as inf / NaN")
    out.append("}")
    with open(out_path, "w") as f:
        f.write("\n".join(out) + "\n")
    print(f"{out_path}: {len(seq)} statements")


if __name__ == "__main__":
    main()
