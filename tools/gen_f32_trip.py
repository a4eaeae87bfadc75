#!/usr/bin/env python3
"""
Generates csrc/rsf_f32_trip.inc: the float32 sampler's incremental TRIP — eight RK4 steps of rsf::f32::rk4_incr
(rsf_device_f32.h) for the lane's two chains, with the emission of their samples and the trip's guard sums — as
hand-scheduled gfx950 assembly: one asm statement per variant (with / without radiation damping).

Why.  Config 5's shape is one wave per SIMD: the wave is bound by its own instruction issue, 4 cycles per vector
instruction (tools/microbench_issue.hip, profiles/r04/microbench_issue.log: 4.06 independent, 5.05 when the instruction reads
the result of the one before it, and 8.05 per pair with the `s_nop 0` that hipcc puts between such a pair — its hazard
recogniser takes op_sel_hi of source 0, set on every packed-f32 instruction, for a destination op_sel; the hardware needs no
wait state there: tools/microbench_pk_f32.hip computes bit-identical results without).  An RK4 step is one long dependency
chain of ~90 operations.  Left to hipcc it became 79 packed operations + 16 emission and guard operations + ~16 wait states
per step, ordered for register pressure (534 cycles per step measured; -misched=gcn-max-ilp does worse, gcn-iterative-ilp
crashes the compiler); statement order is not honoured, a scheduler barrier does not stop IR-level sinking, and an empty-asm
pin on a result draws a wait state itself.  So the trip is written here: the dataflow, an order of issue in which an
instruction rarely follows its own producer (list scheduling of the whole trip, one step's tail under the next step's head),
a register allocation, and the instructions — 88 per step, no wait states.

Registers.  The sampler kernel is compiled with amdgpu_num_vgpr(kCompilerVgprs): the compiler allocates below that and never
touches what lies above — the solve's PRIVATE FILE, v[PRIV0 .. 255].  It holds the per-chain constants (parked once per solve
by the setup statement this script also prints), the trip's table values and its temporaries.  The chain state (w, Rh, ms),
the sums of squares and the guard sums are ordinary in/out operands, so the C++ around the trip keeps the trip's start
values for the rare replay.

The dataflow is rk4_incr's, operation for operation and operand for operand (the generic C++ function remains the
definition: step_any runs it wherever a trip is replayed step by step and in the one-chain form, and oracle/rsf_oracle.c
restates it), so a chain's values do not depend on which of the two ran — tested bit for bit (GPU trip against the
restatement; the two-chain sampler against the one-chain forward kernel).

  python tools/gen_f32_trip.py            writes the file, prints the model's cycle counts and the register budget
  python tools/gen_f32_trip.py --check    verifies the committed file is what this script produces
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "bayesian-markov-chain-monte-carlo_amd", "csrc", "rsf_f32_trip.inc")

NU = 8
PREFETCH = os.environ.get("RSF_TRIP_PREFETCH", "1") != "0"   # (0: an experiment — every trip reads its own tables first and waits)
ISSUE = 4                              # cycles between two issues of one wave
# Spacing the list scheduler aims for between an instruction and the first reader of its result (every operation of the
# trip is a packed-float one: a float64 or scalar instruction among them costs the wave ~8 cycles, tools/microbench_issue.hip).  Not measured latencies: a reader directly behind its producer costs one cycle (header), and with
# these targets the scheduler keeps them apart wherever the dataflow offers something else to issue.
LAT = {"pk": 10}
COMPILER_VGPRS = 154                   # the sampler kernel is limited to v0..v153 (amdgpu_num_vgpr = half of this); the private file is v[154:255]
PRIV0 = COMPILER_VGPRS
CONSTS = ["hhd", "hd", "khh", "kh", "kh6", "boa", "nhboa", "kvk", "bh", "cv", "vref", "h6", "c16", "c13"]
# in/out operands of the trip statement, in this order; then the two LDS byte addresses
OPERANDS = ["w", "Rh", "ms", "s32", "g2r", "g2d"]
ADDR_VV, ADDR_OB = len(OPERANDS), len(OPERANDS) + 1      # the NEXT trip's tables (prefetched while this trip computes)
# the loop form (several trips in one statement): state and total in/out, the next trip's table addresses (advanced by the statement),
# the count of trips left, two scalar temporaries; then the squared guard bounds
LOOP_OPERANDS = ["w", "Rh", "ms", "hi", "lo", "vv_addr", "ob_addr", "n", "m0", "m1", "t2r", "t2d"]


class Src:
    def __init__(self, kind, val, neg=False):
        self.kind, self.val, self.neg = kind, val, neg   # kind: t (value by name) | c (constant) | vl (table index) | imm


def T(name, neg=False): return Src("t", name, neg)
def C(name, neg=False): return Src("c", name, neg)
def VL(i): return Src("vl", i)
def IMM(text): return Src("imm", text)


class Op:
    def __init__(self, name, kind, opcode, srcs, step, dest_fixed=None):
        self.name, self.kind, self.opcode, self.srcs, self.step, self.dest_fixed = name, kind, opcode, srcs, step, dest_fixed
        self.users, self.prio, self.t_issue = [], 0, None

    def deps(self):
        return [s.val for s in self.srcs if s.kind == "t"]


def build(damp):
    ops, by = [], {}

    def op(name, kind, opcode, srcs, step, dest_fixed=None):
        o = Op(name, kind, opcode, srcs, step, dest_fixed)
        ops.append(o)
        by[name] = o
        return name

    w, Rh, ms = "w", "Rh", "ms"            # operand names: the state at the trip's start
    g2r = g2d = s32 = None
    for j in range(NU):
        s = f"_{j}"

        def fma(n, a, b, c): return op(n + s, "pk", "fma", [a, b, c], j)
        def mul(n, a, b): return op(n + s, "pk", "mul", [a, b], j)
        def add(n, a, b): return op(n + s, "pk", "add", [a, b], j)

        def rhs(tag, wv, xr, brx, vl):
            d1 = fma(tag + "1", T(wv, True), xr, T(Rh))                  # d1' = Rh - w xr
            d0r = fma(tag + "0r", C("vref", True), T(wv), VL(vl))         # V_l - V_ref w
            gr = fma("g" + tag + "r", Src(brx.kind, brx.val, True), T(d1), T(d0r))
            if not damp:
                return d0r, d1, gr
            kw = mul("kw" + tag, C("kvk"), T(wv))
            d0 = fma(tag + "0", T(kw, True), T(gr), T(d0r))
            g = fma("g" + tag, T(kw, True), T(gr), T(gr))
            return d0, d1, g

        def incr(tag, rho, kd, d0):
            P = fma("P" + tag, T(rho), C("nhboa"), C("boa"))
            rP = mul("rP" + tag, T(rho), T(P))
            dl = fma("dl" + tag, C(kd), T(d0), T(rP, True))
            d2 = mul("d2" + tag, T(dl), T(dl))
            A = fma("A" + tag, T(dl), T(w), T(w))
            B = fma("B" + tag, T(dl), T("w06" + s), T("w02" + s))
            wn = fma("w" + tag, T(d2), T(B), T(A))
            qn = fma("q" + tag, T(rho), T(rho), T(rho, True))
            return wn, qn, dl

        mul("w02", T(w), IMM("0.5"))
        mul("w06", T(w), C("c16"))
        a0, a1, ga = rhs("a", w, C("hhd"), C("bh"), 2 * j)
        sv = mul("sv1", T(w), T(ga))
        wb, qa, _ = incr("a", a1, "khh", a0)
        xrb = fma("xrb", C("hhd"), T(a1), C("hhd"))
        brb = fma("brb", C("bh"), T(qa), C("bh"))
        b0, b1, gb = rhs("b", wb, T(xrb), T(brb), 2 * j + 1)
        sm = mul("sm1", T(wb), T(gb))
        wc, qb, _ = incr("b", b1, "khh", b0)
        xrc = fma("xrc", C("hhd"), T(b1), C("hhd"))
        brc = fma("brc", C("bh"), T(qb), C("bh"))
        c0, c1, gc = rhs("c", wc, T(xrc), T(brc), 2 * j + 1)
        sm = fma("sm2", T(wc), T(gc), T(sm))
        bc0 = add("bc0", T(b0), T(c0))
        T0 = fma("T0", IMM("2.0"), T(bc0), T(a0))
        bc1 = add("bc1", T(b1), T(c1))
        T1 = fma("T1", IMM("2.0"), T(bc1), T(a1))
        T13 = mul("T13", T(T1), C("c13"))
        rc = add("rc", T(c1), T(c1))
        we, qc, _ = incr("c", rc, "kh", c0)
        xre = fma("xre", C("hd"), T(c1), C("hhd"))
        bre = fma("bre", C("bh"), T(qc), C("bh"))
        e0, e1, ge = rhs("e", we, T(xre), T(bre), 2 * j + 2)
        sv = fma("sv2", T(we), T(ge), T(sv))
        t0 = add("t0", T(T0), T(e0))
        rhoE = fma("rhoE", T(e1), C("c13"), T(T13))
        wn, qn, dlE = incr("n", rhoE, "kh6", t0)
        msn = fma("msn", C("h6"), T(t0), T(ms))
        Rhn = fma("Rhn", T(Rh), T(qn), T(Rh))
        dv = fma("dv", IMM("2.0"), T(sm), T(sv))
        # the trip's guard sums, accumulated in place in their operands
        if g2r is None:
            g2r = op("g2r" + s, "pk", "mul", [T(rhoE), T(rhoE)], j, dest_fixed="g2r")
            g2d = op("g2d" + s, "pk", "mul", [T(dlE), T(dlE)], j, dest_fixed="g2d")
        else:
            g2r = op("g2r" + s, "pk", "fma", [T(rhoE), T(rhoE), T(g2r)], j, dest_fixed="g2r")
            g2d = op("g2d" + s, "pk", "fma", [T(dlE), T(dlE), T(g2d)], j, dest_fixed="g2d")
        ak = mul("ak", T(dv), C("cv"))        # RateStateModel.py:388, from the interval's velocity increment
        # residuals of both chains and the group's float32 sum of their squares (rsf_device_f32.h, Out32): packed, like all else
        r = op("r" + s, "pk", "add", [T(ak), Src("ob", j, True)], j)
        if s32 is None:
            s32 = op("s32" + s, "pk", "mul", [T(r), T(r)], j, dest_fixed="s32")
        else:
            s32 = op("s32" + s, "pk", "fma", [T(r), T(r), T(s32)], j, dest_fixed="s32")
        w, Rh, ms = wn, Rhn, msn
    final = {"w": w, "Rh": Rh, "ms": ms}
    return ops, by, final


def op_deps(o):
    d = o.deps()
    for s in o.srcs:
        if s.kind == "half":
            d.append(s.val[0])
    return d


def schedule(ops, by):
    for o in ops:
        for s in op_deps(o):
            if s in by:
                by[s].users.append(o)
    for o in reversed(ops):      # priority: the longest latency-weighted path to the end of the trip
        o.prio = LAT[o.kind] + max((u.prio for u in o.users), default=0)
    ready_at, order, t, done, left = {}, [], 0, set(), list(ops)
    while left:
        cur = min(o.step for o in left)
        cands = [o for o in left if o.step <= cur + 1 and all((s not in by) or (s in done) for s in op_deps(o))]
        avail = [(max([ready_at.get(s, 0) for s in op_deps(o)], default=0), o) for o in cands]
        now = [(r, o) for r, o in avail if r <= t]
        if now:
            pick = max(now, key=lambda ro: (ro[1].prio, -ro[1].step))[1]
        else:                    # nothing is ready: the wave waits for the earliest result
            pick = min(avail, key=lambda ro: (ro[0], -ro[1].prio))[1]
            t = max(ready_at.get(s, 0) for s in op_deps(pick))
        pick.t_issue = t
        ready_at[pick.name] = t + LAT[pick.kind]
        done.add(pick.name)
        left.remove(pick)
        order.append(pick)
        t += ISSUE
    return order, t


class Regs:
    """Private file layout: constants, tables, then a pool of pairs for temporaries (linear scan over the schedule)."""

    def __init__(self):
        r = PRIV0
        self.const = {}
        for c in CONSTS:
            self.const[c] = r
            r += 2
        self.acc = {"s32": r, "g2r": r + 2, "g2d": r + 4}   # loop form: the group's sum of squares and the guard sums (halves compared in the statement)
        r += 6
        self.vv = r          # 2 NU + 1 floats, base aligned to 4 registers for ds_read_b128
        assert self.vv % 4 == 0, "table base must be aligned for ds_read_b128"
        r += 2 * NU + 2
        self.ob = r          # NU floats
        r += NU
        self.pool = list(range(r, 256, 2))
        self.max_used = 0
        self.n_pool = len(self.pool)

    def alloc(self):
        if not self.pool:
            sys.exit("private register file exhausted: lower the scheduling window or raise the file")
        p = self.pool.pop(0)
        self.max_used = max(self.max_used, self.n_pool - len(self.pool))
        return p

    def free(self, p):
        self.pool.insert(0, p)   # reuse the most recently freed pair first keeps the footprint small


def pair(r): return f"v[{r}:{r + 1}]"


def emit_asm(order, by, final, regs, loop=False):
    last_use = {}
    for i, o in enumerate(order):
        for s in op_deps(o):
            last_use[s] = i
    for v in final.values():
        last_use[v] = len(order)
    names = LOOP_OPERANDS if loop else OPERANDS
    where = {name: f"%{i}" for i, name in enumerate(names)}        # operands print as %N (register pairs)
    if loop:                                                       # the accumulators of the loop form live in the private file
        for k, r in regs.acc.items():
            where[k] = pair(r)
    a_vv, a_ob = (names.index("vv_addr"), names.index("ob_addr")) if loop else (ADDR_VV, ADDR_OB)
    phys = {}
    lines = []
    reads = []   # per line: the table registers it reads
    for i, o in enumerate(order):
        # destination
        if o.dest_fixed:
            dst = where[o.dest_fixed]
        else:
            phys[o.name] = regs.alloc()
            dst = pair(phys[o.name]) if o.kind != "f" else f"v{phys[o.name]}"

        def loc(name):
            if name in phys:
                return pair(phys[name])
            if name in where:
                return where[name]
            fixed = by[name].dest_fixed
            return where[fixed]

        tab = set()
        if o.kind == "pk":
            n = len(o.srcs)
            txt, sel, selhi, neg = [], [0] * n, [1] * n, [0] * n
            for k, s in enumerate(o.srcs):
                if s.kind == "t":
                    txt.append(loc(s.val))
                elif s.kind == "c":
                    txt.append(pair(regs.const[s.val]))
                elif s.kind == "vl":          # one float of the table, for both chains: the register pair that holds it, one half
                    r = regs.vv + s.val
                    tab.add(r)
                    txt.append(pair(r & ~1))
                    sel[k], selhi[k] = r & 1, r & 1
                elif s.kind == "ob":          # the observation of step s.val, for both chains
                    r = regs.ob + s.val
                    tab.add(r)
                    txt.append(pair(r & ~1))
                    sel[k], selhi[k] = r & 1, r & 1
                elif s.kind == "imm":
                    txt.append(s.val)
                    selhi[k] = 0
                neg[k] = 1 if s.neg else 0
            mods = ""
            if any(sel):
                mods += " op_sel:[" + ",".join(map(str, sel)) + "]"
            if not all(selhi):
                mods += " op_sel_hi:[" + ",".join(map(str, selhi)) + "]"
            if any(neg):
                mods += " neg_lo:[" + ",".join(map(str, neg)) + "] neg_hi:[" + ",".join(map(str, neg)) + "]"
            ins = f"v_pk_{o.opcode}_f32 {dst}, " + ", ".join(txt) + mods
        lines.append(ins)
        reads.append(tab)
        # registers whose value is dead after this instruction
        for s in dict.fromkeys(op_deps(o)):   # (in order: the register assignment must not depend on hash seeds)
            if last_use.get(s) == i and s in phys:
                regs.free(phys.pop(s))
        if o.name not in last_use and o.name in phys:      # a result nobody reads (none expected)
            regs.free(phys.pop(o.name))
    tail = []
    if loop:
        # The guard BEFORE anything is written back: on an alarm the statement is left with the state operands still holding the
        # trip's start values (they are only written below) and the count operand telling which trip it was.
        m0, m1, n = where["m0"], where["m1"], where["n"]
        gr, gd = regs.acc["g2r"], regs.acc["g2d"]
        tail += [f"v_cmp_nlt_f32_e64 {m0}, v{gr}, {where['t2r']}", f"v_cmp_nlt_f32_e64 {m1}, v{gr + 1}, {where['t2r']}", f"s_or_b64 {m0}, {m0}, {m1}",
                 f"v_cmp_nlt_f32_e64 {m1}, v{gd}, {where['t2d']}", f"s_or_b64 {m0}, {m0}, {m1}",
                 f"v_cmp_nlt_f32_e64 {m1}, v{gd + 1}, {where['t2d']}", f"s_or_b64 {m0}, {m0}, {m1}",
                 f"s_and_b64 {m0}, {m0}, exec", "s_cbranch_scc1 .Lrsf_trips_exit_%=", ".p2align 3"]
    for k, v in final.items():
        tail.append(f"v_pk_mov_b32 {where[k]}, {pair(phys[v])}, {pair(phys[v])} op_sel:[0,1]")
    if loop:
        # the group's sum into the total (hi, lo) by the exact two-sum (rsf_device_f32.h, Out32::flush)
        x, hi, lo = where["s32"], where["hi"], where["lo"]
        t = [pair(regs.alloc()) for _ in range(4)]
        tail += [f"v_pk_add_f32 {t[0]}, {hi}, {x}",                                          # s = hi + x
                 f"v_pk_add_f32 {t[1]}, {t[0]}, {hi} neg_lo:[0,1] neg_hi:[0,1]",               # bb = s - hi
                 f"v_pk_add_f32 {t[2]}, {t[0]}, {t[1]} neg_lo:[0,1] neg_hi:[0,1]",             # t = s - bb
                 f"v_pk_add_f32 {t[3]}, {x}, {t[1]} neg_lo:[0,1] neg_hi:[0,1]",                # e2 = x - bb
                 f"v_pk_add_f32 {t[2]}, {hi}, {t[2]} neg_lo:[0,1] neg_hi:[0,1]",               # e1 = hi - t
                 f"v_pk_mov_b32 {hi}, {t[0]}, {t[0]} op_sel:[0,1]",                            # hi = s
                 f"v_pk_add_f32 {t[2]}, {t[2]}, {t[3]}",                                       # e1 + e2
                 f"v_pk_add_f32 {lo}, {lo}, {t[2]}",                                           # lo += e
                 f"v_add_u32_e64 {where['vv_addr']}, {where['vv_addr']}, {8 * NU}",            # the tables of the trip after the next
                 f"v_add_u32_e64 {where['ob_addr']}, {where['ob_addr']}, {4 * NU}",
                 f"s_sub_u32 {n}, {n}, 1", f"s_cmp_lg_u32 {n}, 0", "s_cbranch_scc1 .Lrsf_trips_loop_%="]
    lines += tail
    reads += [set()] * len(tail)
    if not PREFETCH:
        return [l for l, _ in loads(regs, a_vv, a_ob)] + ["s_waitcnt lgkmcnt(0)", ".p2align 3"] + lines
    # the NEXT trip's table values: each load as soon as the registers it overwrites have been read for the last time
    out = list(zip(lines, reads))
    for ins, dest in reversed(loads(regs, a_vv, a_ob)):
        last = max((i for i, (_, rd) in enumerate(out) if rd & dest), default=-1)
        out.insert(last + 1, (ins, set()))
    return [l for l, _ in out]


def loads(regs, op_vv, op_ob):
    """ds_read instructions of one trip's tables into the private file: (instruction, set of destination registers)."""
    out = []
    for i in range(0, 2 * NU, 4):
        out.append((f"ds_read_b128 v[{regs.vv + i}:{regs.vv + i + 3}], %{op_vv}" + (f" offset:{4 * i}" if i else ""), set(range(regs.vv + i, regs.vv + i + 4))))
    out.append((f"ds_read_b32 v{regs.vv + 2 * NU}, %{op_vv} offset:{8 * NU}", {regs.vv + 2 * NU}))
    for j in range(0, NU, 2):
        out.append((f"ds_read2_b32 v[{regs.ob + j}:{regs.ob + j + 1}], %{op_ob} offset0:{j} offset1:{j + 1}", {regs.ob + j, regs.ob + j + 1}))
    return out


def c_string(lines, indent="      "):
    return "\n".join(f'{indent}"{l}\\n\\t"' for l in lines)


def generate():
    out = []
    out.append("// GENERATED by tools/gen_f32_trip.py — do not edit; `python tools/gen_f32_trip.py --check` verifies it.")
    out.append("// The float32 sampler's incremental trip (rsf::f32::trip32, two chains per lane) as hand-scheduled gfx950 assembly, and")
    out.append("// the statement that parks the solve's constants in the private register file.  Why and how: that script's header.")
    regs0 = Regs()
    out.append(f"#define RSF_F32_TRIP_COMPILER_VGPRS {COMPILER_VGPRS}")
    out.append(f"#define RSF_F32_TRIP_STEPS {NU}")
    out.append("")
    out.append("// per-chain constants of the solve → the private file (once per solve); the literals 1/6 and 1/3 are written there too")
    setup = []
    names = [c for c in CONSTS if c not in ("c16", "c13")]
    for i, c in enumerate(names):
        r = regs0.const[c]
        setup.append(f"v_pk_mov_b32 {pair(r)}, %{i}, %{i} op_sel:[0,1]")
    for c, bits in (("c16", "0x3e2aaaab"), ("c13", "0x3eaaaaab")):
        r = regs0.const[c]
        setup.append(f"v_mov_b32_e32 v{r}, {bits}")
        setup.append(f"v_mov_b32_e32 v{r + 1}, {bits}")
    out.append("#define RSF_F32_TRIP_SETUP(L) \\")
    out.append("  asm volatile( \\")
    out.append("\n".join(f'      "{l}\\n\\t" \\' for l in setup))
    out.append("      : : " + ", ".join(f'"v"((L).{c})' for c in names) + ' : "v255")')
    out.append("")
    out.append("// the tables of a chunk's FIRST trip → the private file (every later trip's are prefetched by the trip before it)")
    out.append("#define RSF_F32_TRIP_PRELOAD(vv_addr, ob_addr) \\")
    out.append("  asm volatile( \\")
    out.append("\n".join(f'      "{l}\\n\\t" \\' for l, _ in loads(regs0, 0, 1)))
    out.append('      : : "v"(vv_addr), "v"(ob_addr) : "memory")')
    out.append("")
    report = []
    for damp in (True, False):
        ops, by, final = build(damp)
        order, cycles = schedule(ops, by)
        regs = Regs()
        # Every instruction of the trip is 8 bytes long (VOP3P).  Begun on a 4-byte boundary — as the code before the statement
        # leaves it half the time — the same stream takes 5.07 cycles per instruction instead of 4.06 (an instruction that
        # straddles a fetch boundary costs the wave extra; profiles/r04/ab_f32_alignment.log: the kernel 21.7 ms against 19.2 ms
        # with one, three, seven or fifteen s_nop in front of the body).  The assembler pads code with s_nop.
        # (the s_waitcnt is a 4-byte instruction: the alignment comes behind it)
        body = (["s_waitcnt lgkmcnt(0)"] if PREFETCH else []) + [".p2align 3"] + emit_asm(order, by, final, regs)
        npk = sum(o.kind == "pk" for o in ops)
        tag = "DAMPED" if damp else "UNDAMPED"
        report.append(f"{tag.lower()}: {len(ops)} operations ({npk} packed), model {cycles} cycles = {cycles / NU:.0f} per step; private file: "
                      f"{2 * len(CONSTS)} constants + {2 * NU + 2 + NU} table + {2 * regs.max_used} of {2 * regs.n_pool} temporaries")
        out.append(f"// {report[-1]}")
        out.append(f"#define RSF_F32_TRIP_{tag}(w, Rh, ms, s32, g2r, g2d, next_vv_addr, next_ob_addr) \\")
        out.append("  asm volatile( \\")
        out.append("\n".join(f'      "{l}\\n\\t" \\' for l in body))
        out.append('      : "+v"(w), "+v"(Rh), "+v"(ms), "=&v"(s32), "=&v"(g2r), "=&v"(g2d) : "v"(next_vv_addr), "v"(next_ob_addr) : "memory")')
        out.append("")
        # The loop form: `n` trips in ONE statement — what the compiled code does between two trips (the guard test, the group's sum
        # into the total, addresses, loop control: ~60 instructions) shrinks to ~25 of its own.  It is left early, with `n` > 0 and
        # the state of the failing trip's START, when a guard sum is not below its bound (the caller replays that trip step by
        # step; the read-ahead has overwritten its tables: RSF_F32_TRIP_PRELOAD first).  Only for waves without a full-evaluation chain.
        regs = Regs()
        body = emit_asm(order, by, final, regs, loop=True)
        out.append(f"#define RSF_F32_TRIPS_LOOP_{tag}(w, Rh, ms, hi, lo, next_vv_addr, next_ob_addr, n, m0, m1, t2r, t2d) \\")
        out.append("  asm volatile( \\")
        out.append("\n".join(f'      "{l}\\n\\t" \\' for l in [".p2align 3", ".Lrsf_trips_loop_%=:", "s_waitcnt lgkmcnt(0)", ".p2align 3"] + body + [".Lrsf_trips_exit_%=:"]))
        out.append('      : "+v"(w), "+v"(Rh), "+v"(ms), "+v"(hi), "+v"(lo), "+v"(next_vv_addr), "+v"(next_ob_addr), "+s"(n), "=&s"(m0), "=&s"(m1) : "s"(t2r), "s"(t2d) : "memory", "scc")')
        out.append("")
    return "\n".join(out), report


ALIGN = os.environ.get("RSF_TRIP_ALIGN", ".p2align 3")   # (experiments: ".p2align 3\ns_nop 0" starts the body on a 4-byte boundary)


def bench_variants(path):
    """Mutated copies of the damped trip for tools/microbench_trip.hip (results are garbage: only the time is looked at):
    A: every source a fixed register (no dependencies, no variety of sources);  B: destinations redirected to a rotating set of
    eight pairs (no dependencies, the sources' variety kept);  C: the original instructions in a shuffled order."""
    import random
    import re
    ops, by, final = build(True)
    order, _ = schedule(ops, by)
    regs = Regs()
    body = emit_asm(order, by, final, regs)
    valu = [l for l in body if l.startswith("v_pk_")]
    rnd = random.Random(1)
    out = ["// GENERATED by tools/gen_f32_trip.py --bench: timing variants of the damped trip (not part of the product)"]

    def macro(name, lines):
        out.append(f"#define RSF_F32_TRIP_BENCH_{name}() \\")
        out.append("  asm volatile( \\")
        out.append("\n".join(f'      "{l}\\n\\t" \\' for l in [ALIGN] + lines))
        out.append('      : : : "memory")')
        out.append("")

    def fix_operands(l):   # operands of the statement → private registers
        return re.sub(r"%(\d+)", lambda m: pair(PRIV0 + 2 * int(m.group(1))), l)

    base = [fix_operands(l) for l in valu]
    macro("ORIGINAL", base)
    a = []
    for l in base:
        head, rest = l.split(" ", 1)
        parts = rest.split(", ")
        dst, srcs = parts[0], parts[1:]
        tail = ""
        if " " in srcs[-1]:
            srcs[-1], tail = srcs[-1].split(" ", 1)
            tail = " " + tail
        fixed = [pair(PRIV0 + 2 * k) for k in range(len(srcs))]
        a.append(f"{head} {dst}, " + ", ".join(fixed) + tail)
    macro("FIXED_SOURCES", a)
    b = []
    for i, l in enumerate(base):
        head, rest = l.split(" ", 1)
        dst, others = rest.split(", ", 1)
        b.append(f"{head} {pair(240 + 2 * (i % 8))}, {others}")
    macro("ROTATING_DESTINATIONS", b)
    c = list(base)
    rnd.shuffle(c)
    macro("SHUFFLED", c)
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    print(f"{len(base)} instructions per variant -> {path}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--bench", metavar="FILE", help="write timing variants of the damped trip (tools/microbench_trip.hip)")
    args = ap.parse_args()
    if args.bench:
        bench_variants(args.bench)
        return
    text, report = generate()
    if args.check:
        if open(OUT).read() != text:
            sys.exit(f"{OUT} is not what tools/gen_f32_trip.py generates: regenerate it")
        print("ok")
        return
    with open(OUT, "w") as f:
        f.write(text)
    print("\n".join(report))


if __name__ == "__main__":
    main()
