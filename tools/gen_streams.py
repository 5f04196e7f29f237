#!/usr/bin/env python3
"""Generate instruction streams for the issue-rate microbenchmark (mini_nbody_amd/csrc/microbench_streams.inc).

Each stream is one inline-asm block with explicit physical VGPRs (so that register banks and instruction order are
exactly what is written here) and a count of the instructions in it.  The streams answer:
  dep<d>     how far apart must dependent VALU instructions be to issue at 2 cycles?
  bank_*     which operand/bank combinations cost extra cycles?
  pair_c<C>  the 12-instruction pair interaction for C bodies per lane, interleaved op by op (dependency distance
             C), registers chosen so that no instruction reads three VGPRs of one bank
Run: python tools/gen_streams.py   (the output file is committed)
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "mini_nbody_amd", "csrc", "microbench_streams.inc")

streams = []   # (name, description, [instr], n_instr, clobbers)


def add(name, desc, instrs, regs):
    n = len([i for i in instrs if not i.startswith('.') and not i.startswith('s_nop')])
    streams.append((name, desc, instrs, n, sorted(set(regs), key=lambda r: int(r[1:]))))


def vregs(instrs):
    import re
    out = set()
    for i in instrs:
        out.update(re.findall(r"\bv(\d+)\b", i))
    return ["v%s" % r for r in out]


# ---- dependency distance: 'd' independent accumulators round-robin, banks all different from the two fixed sources
for d in (1, 2, 3, 4, 6):
    accs = [40 + 4 * k + 1 for k in range(d)]          # bank 1
    ins = []
    for rep in range(48 // d * d // d):
        for a in accs:
            ins.append("v_fma_f32 v%d, v%d, v34, v35" % (a, a))   # v34 bank 2, v35 bank 3, acc bank 1
    add("dep%d" % d, "v_fma_f32 chain, dependent instruction %d slots later" % d, ins, vregs(ins))

# the same for the transcendental's consumer: rsq then a dependent mul 'd' slots later, fillers are independent fmas
for d in (1, 2, 4, 8):
    ins = []
    for rep in range(4):
        ins.append("v_rsq_f32 v41, v45")
        for k in range(d - 1):
            ins.append("v_fma_f32 v%d, v%d, v34, v35" % (49 + 4 * k, 49 + 4 * k))
        ins.append("v_mul_f32 v45, v41, v41")
        for k in range(11 - d):
            ins.append("v_fma_f32 v%d, v%d, v34, v35" % (49 + 4 * (k % 6), 49 + 4 * (k % 6)))
    add("rsqdep%d" % d, "v_rsq_f32 -> dependent v_mul_f32 %d slots later (+ independent fma fillers, 12 per group)" % d, ins, vregs(ins))

# ---- bank rules (independent instructions: 8 accumulators)
def bank_stream(name, desc, fmt):
    ins = []
    for rep in range(8):
        for k in range(8):
            ins.append(fmt(k))
    add(name, desc, ins, vregs(ins))

# vop3 fma d, s0, s1, s2 with d == s2
bank_stream("bank_s0s1", "v_fma_f32: src0/src1 same bank, src2 other", lambda k: "v_fma_f32 v%d, v36, v40, v%d" % (65 + 4 * k, 65 + 4 * k))
bank_stream("bank_s1s2", "v_fma_f32: src1/src2 same bank, src0 other", lambda k: "v_fma_f32 v%d, v37, v40, v%d" % (64 + 4 * k, 64 + 4 * k))
bank_stream("bank_s0s2", "v_fma_f32: src0/src2 same bank, src1 other", lambda k: "v_fma_f32 v%d, v36, v41, v%d" % (64 + 4 * k, 64 + 4 * k))
bank_stream("bank_all3", "v_fma_f32: all three same bank", lambda k: "v_fma_f32 v%d, v36, v40, v%d" % (64 + 4 * k, 64 + 4 * k))
bank_stream("bank_none", "v_fma_f32: three different banks", lambda k: "v_fma_f32 v%d, v37, v42, v%d" % (64 + 4 * k, 64 + 4 * k))
bank_stream("bank_mod2", "v_fma_f32: regs equal mod 2 but not mod 4 (36, 38, acc mod4=0)", lambda k: "v_fma_f32 v%d, v38, v42, v%d" % (64 + 4 * k, 64 + 4 * k))
bank_stream("bank_2src_same", "v_mul_f32: two sources same bank", lambda k: "v_mul_f32 v%d, v36, v40" % (65 + 4 * k))
bank_stream("bank_samereg", "v_fma_f32 d, a, a, c: a and c same bank (2 distinct regs)", lambda k: "v_fma_f32 v%d, v36, v36, v%d" % (64 + 4 * k, 64 + 4 * k))
bank_stream("bank_dst", "v_mul_f32: destination same bank as both sources (3 different regs, sources differ in bank)", lambda k: "v_mul_f32 v%d, v36, v41" % (64 + 4 * k))


# ---- the pair interaction, C chains interleaved op by op.
# per chain registers (bank in brackets): xi[1] yi[2] zi[3] | dx[1] dy[2] dz[3] | t[0] u[1] | ax[2] ay[3] az[1]
# 3-VGPR-source instructions are only the final fmacs: (ax,dx,t) = banks (2,1,0), (ay,dy,t) = (3,2,0), (az,dz,t) = (1,3,0)
def chain_regs(c, base=32):
    b = base + 12 * c          # 12 registers per chain, base multiple of 4
    return dict(t=b + 0, xi=b + 1, yi=b + 2, zi=b + 3, u=b + 5 - 0, dx=b + 5 + 4 - 0, dy=b + 6, dz=b + 7,
                ax=b + 10, ay=b + 11, az=b + 9 - 0)


def chain_regs2(c, base=32):
    # explicit, checked below: 11 registers out of a 12-register window
    b = base + 12 * c
    r = dict(t=b + 0, xi=b + 1, yi=b + 2, zi=b + 3, u=b + 4 + 1, dx=b + 8 + 1, dy=b + 4 + 2, dz=b + 4 + 3,
             ax=b + 8 + 2, ay=b + 8 + 3, az=b + 4 + 0)
    # banks: t0 xi1 yi2 zi3 u1 dx1 dy2 dz3 ax2 ay3 az0 -> (az,dz,t) = (0,3,0): two equal, fine; keep all-different where possible
    return r


def pair_ops(r, sx, sy, sz):
    return [
        "v_sub_f32 v%d, %s, v%d" % (r["dx"], sx, r["xi"]),
        "v_sub_f32 v%d, %s, v%d" % (r["dy"], sy, r["yi"]),
        "v_sub_f32 v%d, %s, v%d" % (r["dz"], sz, r["zi"]),
        "v_fmaak_f32 v%d, v%d, v%d, 0x3089705f" % (r["t"], r["dz"], r["dz"]),
        "v_fmac_f32 v%d, v%d, v%d" % (r["t"], r["dy"], r["dy"]),
        "v_fmac_f32 v%d, v%d, v%d" % (r["t"], r["dx"], r["dx"]),
        "v_rsq_f32 v%d, v%d" % (r["t"], r["t"]),
        "v_mul_f32 v%d, v%d, v%d" % (r["u"], r["t"], r["t"]),
        "v_mul_f32 v%d, v%d, v%d" % (r["t"], r["t"], r["u"]),
        "v_fmac_f32 v%d, v%d, v%d" % (r["ax"], r["dx"], r["t"]),
        "v_fmac_f32 v%d, v%d, v%d" % (r["ay"], r["dy"], r["t"]),
        "v_fmac_f32 v%d, v%d, v%d" % (r["az"], r["dz"], r["t"]),
    ]


def check_banks(ins):
    import re
    for i in ins:
        regs = [int(x) for x in re.findall(r"\bv(\d+)\b", i)]
        srcs = regs[1:] if not i.startswith("v_fmac") else regs   # fmac reads its destination
        distinct = set(srcs)
        if len(srcs) >= 3:
            assert len({x % 2 for x in srcs}) >= 2, i


for C in (1, 2, 3, 4):
    ins = []
    regs = [chain_regs2(c) for c in range(C)]
    for j in range(4):               # 4 sources per block; the SGPR triple alternates so no two adjacent j share it
        ops = [pair_ops(regs[c], "s20", "s21", "s22") for c in range(C)]
        for k in range(12):
            for c in range(C):
                ins.append(ops[c][k])
    add("pair_c%d" % C, "pair interaction, %d bodies per lane interleaved op by op (SGPR sources)" % C, ins, vregs(ins))

# software-skewed variant for C=2: chain B runs 6 ops behind chain A, so every dependent pair is >= 2 slots apart and the
# rsq's consumer 2 slots after it is an op of the OTHER chain's non-transcendental half
ins = []
regs = [chain_regs2(0), chain_regs2(1)]
seqA = []
seqB = []
for j in range(5):
    seqA += pair_ops(regs[0], "s20", "s21", "s22")
    seqB += pair_ops(regs[1], "s20", "s21", "s22")
seqB = seqB[6:]                      # skew by half a pair
n = min(len(seqA), len(seqB))
n = 48 // 2
for k in range(n):
    ins.append(seqA[k])
    ins.append(seqB[k])
add("pair_c2_skew", "pair interaction, 2 bodies per lane, second chain skewed by 6 ops", ins, vregs(ins))



# ---- VALU result -> v_rsq_f32 source, d slots later (groups of 12: 1 producer fma, 1 rsq, 10 independent fillers)
for d in (1, 2, 3, 4, 6, 8, 10):
    ins = []
    for rep in range(4):
        ins.append("v_fma_f32 v45, v37, v42, v45")
        fill = ["v_fma_f32 v%d, v%d, v34, v35" % (49 + 4 * (k % 7), 49 + 4 * (k % 7)) for k in range(10)]
        ins += fill[:d - 1]
        ins.append("v_rsq_f32 v40, v45")
        ins += fill[d - 1:]
    add("valu2rsq%d" % d, "VALU result -> v_rsq_f32 source %d slots later (1 rsq per 12 instructions)" % d, ins, vregs(ins))


# ---- 3-stage software pipeline of the pair interaction (what the kernel's hand-scheduled loop will issue):
# body(j) = [mul, mul, fmac x3 of pair j] [rsq of pair j+1] [sub x3, fma x3 of pair j+2], C chains interleaved op by op.
# parity-safe registers: t (d2 / inv / inv3) even, dx dy dz odd  =>  no instruction reads three same-parity VGPRs.
def p3_regs(c, base):
    # per chain: t0 t1 t2 (even), d[3][3] (odd), xi yi zi u ax ay az (free): 3 + 9 + 7 = 19 -> window of 20
    b = base + 20 * c
    ev = [b + 2 * k for k in range(10)]
    od = [b + 2 * k + 1 for k in range(10)]
    return dict(t=[ev[0], ev[1], ev[2]], d=[[od[0], od[1], od[2]], [od[3], od[4], od[5]], [od[6], od[7], od[8]]],
                xi=ev[3], yi=ev[4], zi=ev[5], u=ev[6], ax=ev[7], ay=ev[8], az=ev[9])


def p3_body(regs, phase, sj):
    """One body for a list of chains; phase selects which of the 3 rotating register sets plays which role."""
    A, B, Cc = phase % 3, (phase + 1) % 3, (phase + 2) % 3   # A: pair j (finish), B: pair j+1 (rsq), Cc: pair j+2 (start)
    per_chain = []
    for r in regs:
        tA, tB, tC = r["t"][A], r["t"][B], r["t"][Cc]
        dA, dC = r["d"][A], r["d"][Cc]
        per_chain.append([
            "v_mul_f32 v%d, v%d, v%d" % (r["u"], tA, tA),
            "v_mul_f32 v%d, v%d, v%d" % (tA, tA, r["u"]),
            "v_fmac_f32 v%d, v%d, v%d" % (r["ax"], dA[0], tA),
            "v_fmac_f32 v%d, v%d, v%d" % (r["ay"], dA[1], tA),
            "v_fmac_f32 v%d, v%d, v%d" % (r["az"], dA[2], tA),
            "v_rsq_f32 v%d, v%d" % (tB, tB),
            "v_sub_f32 v%d, %s, v%d" % (dC[0], sj[0], r["xi"]),
            "v_sub_f32 v%d, %s, v%d" % (dC[1], sj[1], r["yi"]),
            "v_sub_f32 v%d, %s, v%d" % (dC[2], sj[2], r["zi"]),
            "v_fmaak_f32 v%d, v%d, v%d, 0x3089705f" % (tC, dC[2], dC[2]),
            "v_fmac_f32 v%d, v%d, v%d" % (tC, dC[1], dC[1]),
            "v_fmac_f32 v%d, v%d, v%d" % (tC, dC[0], dC[0]),
        ])
    out = []
    for k in range(12):
        for pc in per_chain:
            out.append(pc[k])
    return out


for C in (1, 2, 3):
    regs = [p3_regs(c, 32) for c in range(C)]
    ins = []
    for phase in range(3):
        ins += p3_body(regs, phase, ("s20", "s21", "s22"))
    check_banks(ins)
    add("pipe3_c%d" % C, "3-stage software-pipelined pair interaction, %d bodies per lane, parity-safe registers" % C, ins, vregs(ins))

# the plain op-by-op interleave again, now parity-safe (t even, d odd), to separate the bank effect from the hazard
for C in (2, 4):
    regs = [p3_regs(c, 32) for c in range(C)]
    ins = []
    for j in range(2):
        per = []
        for r in regs:
            t, d = r["t"][0], r["d"][0]
            per.append([
                "v_sub_f32 v%d, s20, v%d" % (d[0], r["xi"]), "v_sub_f32 v%d, s21, v%d" % (d[1], r["yi"]),
                "v_sub_f32 v%d, s22, v%d" % (d[2], r["zi"]), "v_fmaak_f32 v%d, v%d, v%d, 0x3089705f" % (t, d[2], d[2]),
                "v_fmac_f32 v%d, v%d, v%d" % (t, d[1], d[1]), "v_fmac_f32 v%d, v%d, v%d" % (t, d[0], d[0]),
                "v_rsq_f32 v%d, v%d" % (t, t), "v_mul_f32 v%d, v%d, v%d" % (r["u"], t, t), "v_mul_f32 v%d, v%d, v%d" % (t, t, r["u"]),
                "v_fmac_f32 v%d, v%d, v%d" % (r["ax"], d[0], t), "v_fmac_f32 v%d, v%d, v%d" % (r["ay"], d[1], t),
                "v_fmac_f32 v%d, v%d, v%d" % (r["az"], d[2], t)])
        for k in range(12):
            for pc in per:
                ins.append(pc[k])
    check_banks(ins)
    add("pairsafe_c%d" % C, "pair interaction, %d bodies per lane op by op, parity-safe registers" % C, ins, vregs(ins))



# ---- bisection of the pair stream: substitute one instruction kind at a time (4 chains, parity-safe registers)
def pairsafe(C, sub=None, nj=2):
    regs = [p3_regs(c, 32) for c in range(C)]
    ins = []
    for j in range(nj):
        per = []
        for r in regs:
            t, d = r["t"][0], r["d"][0]
            ops = [
                "v_sub_f32 v%d, s20, v%d" % (d[0], r["xi"]), "v_sub_f32 v%d, s21, v%d" % (d[1], r["yi"]),
                "v_sub_f32 v%d, s22, v%d" % (d[2], r["zi"]), "v_fmaak_f32 v%d, v%d, v%d, 0x3089705f" % (t, d[2], d[2]),
                "v_fmac_f32 v%d, v%d, v%d" % (t, d[1], d[1]), "v_fmac_f32 v%d, v%d, v%d" % (t, d[0], d[0]),
                "v_rsq_f32 v%d, v%d" % (t, t), "v_mul_f32 v%d, v%d, v%d" % (r["u"], t, t), "v_mul_f32 v%d, v%d, v%d" % (t, t, r["u"]),
                "v_fmac_f32 v%d, v%d, v%d" % (r["ax"], d[0], t), "v_fmac_f32 v%d, v%d, v%d" % (r["ay"], d[1], t),
                "v_fmac_f32 v%d, v%d, v%d" % (r["az"], d[2], t)]
            if sub:
                ops = sub(ops, r, t, d)
            per.append(ops)
        for k in range(12):
            for pc in per:
                ins.append(pc[k])
    return ins


def sub_norsq(ops, r, t, d):
    ops[6] = "v_mov_b32 v%d, v%d" % (t, t)
    return ops


def sub_fma_sgpr_eps(ops, r, t, d):
    ops[3] = "v_fma_f32 v%d, v%d, v%d, s23" % (t, d[2], d[2])
    return ops


def sub_vgpr_src(ops, r, t, d):
    ops[0] = "v_sub_f32 v%d, v%d, v%d" % (d[0], r["u"] + 100, r["xi"])
    ops[1] = "v_sub_f32 v%d, v%d, v%d" % (d[1], r["u"] + 100, r["yi"])
    ops[2] = "v_sub_f32 v%d, v%d, v%d" % (d[2], r["u"] + 100, r["zi"])
    return ops


def sub_mul_for_fmac_tail(ops, r, t, d):
    # the three accumulating fmacs (3 VGPR reads each) -> 2-read muls into scratch
    ops[9] = "v_mul_f32 v%d, v%d, v%d" % (r["ax"], d[0], t)
    ops[10] = "v_mul_f32 v%d, v%d, v%d" % (r["ay"], d[1], t)
    ops[11] = "v_mul_f32 v%d, v%d, v%d" % (r["az"], d[2], t)
    return ops


def sub_all_vop3(ops, r, t, d):
    # everything as VOP3 v_fma_f32 with explicit operands (no VOP2 tied destination forms)
    ops[4] = "v_fma_f32 v%d, v%d, v%d, v%d" % (t, d[1], d[1], t)
    ops[5] = "v_fma_f32 v%d, v%d, v%d, v%d" % (t, d[0], d[0], t)
    ops[9] = "v_fma_f32 v%d, v%d, v%d, v%d" % (r["ax"], d[0], t, r["ax"])
    ops[10] = "v_fma_f32 v%d, v%d, v%d, v%d" % (r["ay"], d[1], t, r["ay"])
    ops[11] = "v_fma_f32 v%d, v%d, v%d, v%d" % (r["az"], d[2], t, r["az"])
    return ops


for nm, fn, desc in (("bis_norsq", sub_norsq, "rsq -> v_mov_b32"), ("bis_sgpreps", sub_fma_sgpr_eps, "v_fmaak literal -> v_fma with SGPR eps"),
                     ("bis_vgprsrc", sub_vgpr_src, "SGPR sources -> VGPR sources"), ("bis_multail", sub_mul_for_fmac_tail, "3 accumulate fmacs -> v_mul"),
                     ("bis_vop3", sub_all_vop3, "v_fmac (VOP2) -> v_fma (VOP3)")):
    ins = pairsafe(4, fn)
    add(nm, "pairsafe_c4 with " + desc, ins, vregs(ins))

# the synthetic mix with explicit registers in this harness: 11 independent VOP3 fma + 1 rsq (should be 2.5)
ins = []
for rep in range(4):
    for k in range(6):
        ins.append("v_fma_f32 v%d, v%d, v34, v35" % (49 + 4 * k, 49 + 4 * k))
    ins.append("v_rsq_f32 v81, v81")
    for k in range(5):
        ins.append("v_fma_f32 v%d, v%d, v34, v35" % (49 + 4 * k, 49 + 4 * k))
add("mix_explicit", "11 independent v_fma_f32 + 1 v_rsq_f32 (self-dependent only), explicit registers", ins, vregs(ins))
ins = []
for rep in range(4):
    for k in range(6):
        ins.append("v_fmac_f32 v%d, v34, v35" % (49 + 4 * k))
    ins.append("v_rsq_f32 v81, v81")
    for k in range(5):
        ins.append("v_fmac_f32 v%d, v34, v35" % (49 + 4 * k))
add("mix_fmac", "11 independent v_fmac_f32 (VOP2) + 1 v_rsq_f32", ins, vregs(ins))
ins = []
for rep in range(4):
    for k in range(6):
        ins.append("v_fmac_f32 v%d, v34, v35" % (49 + 4 * k))
    ins.append("v_rsq_f32 v82, v81")
    for k in range(5):
        ins.append("v_fmac_f32 v%d, v34, v35" % (49 + 4 * k))
add("mix_fmac_rsqdst", "same, rsq writes a different register than it reads", ins, vregs(ins))



# ---- does the transcendental's result write collide with VALU writes of the same bank?
for rd, tag in ((80, "0"), (82, "2"), (83, "3"), (81, "1")):
    ins = []
    for rep in range(4):
        for k in range(6):
            ins.append("v_fma_f32 v%d, v%d, v34, v35" % (49 + 4 * k, 49 + 4 * k))
        ins.append("v_rsq_f32 v%d, v%d" % (rd, rd))
        for k in range(5):
            ins.append("v_fma_f32 v%d, v%d, v34, v35" % (49 + 4 * k, 49 + 4 * k))
    add("mixw_rsq%s" % tag, "11 v_fma_f32 writing regs = 1 mod 4 + 1 v_rsq_f32 writing a reg = %s mod 4" % tag, ins, vregs(ins))
# fma destinations spread over all four residues (like the compiler-allocated original), rsq in v81
ins = []
for rep in range(4):
    dsts = [48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58]
    for k in range(6):
        ins.append("v_fma_f32 v%d, v%d, v34, v35" % (dsts[k], dsts[k]))
    ins.append("v_rsq_f32 v81, v81")
    for k in range(5):
        ins.append("v_fma_f32 v%d, v%d, v34, v35" % (dsts[k], dsts[k]))
add("mixw_spread", "11 v_fma_f32 writing consecutive registers + 1 v_rsq_f32 (v81)", ins, vregs(ins))
# exactly the register numbers the compiler picked in microbench.hip's mix (2.52 cycles there)
ins = []
for rep in range(4):
    for d in (3, 10, 11, 12, 13, 14):
        ins.append("v_fma_f32 v%d, v%d, v17, v18" % (d + 40, d + 40))
    ins.append("v_rsq_f32 v55, v55")
    for d in (16, 3, 10, 11, 12):
        ins.append("v_fma_f32 v%d, v%d, v17, v18" % (d + 40, d + 40))
add("mixw_orig", "the compiler's register pattern of microbench.hip's mix, shifted by 40", [i.replace("v17", "v57").replace("v18", "v58") for i in ins],
    vregs([i.replace("v17", "v57").replace("v18", "v58") for i in ins]))



# ---- code placement: the same mix with the block start forced to 0 / 4 mod 8 bytes (and 0 mod 64)
def mix_vop3():
    ins = []
    for rep in range(4):
        for k in range(6):
            ins.append("v_fma_f32 v%d, v%d, v34, v35" % (49 + 4 * k, 49 + 4 * k))
        ins.append("v_rsq_f32 v81, v81")
        for k in range(5):
            ins.append("v_fma_f32 v%d, v%d, v34, v35" % (49 + 4 * k, 49 + 4 * k))
    return ins


base = mix_vop3()
add("align_8", "mix (VOP3 fma + rsq), block start aligned to 8 bytes", [".p2align 3"] + base, vregs(base))
add("align_8p4", "mix, block start = 4 mod 8 bytes", [".p2align 3", "s_nop 0"] + base, vregs(base))
add("align_64", "mix, block start aligned to 64 bytes", [".p2align 6"] + base, vregs(base))
add("align_64p4", "mix, block start = 4 mod 64", [".p2align 6", "s_nop 0"] + base, vregs(base))
# rsq as a 64-bit encoding (_e64) so that every instruction of the block is 8 bytes
base2 = [i.replace("v_rsq_f32 v81, v81", "v_rsq_f32_e64 v81, v81") for i in base]
add("align_8_rsq64", "mix with v_rsq_f32_e64 (all instructions 8 bytes), 8-byte aligned", [".p2align 3"] + base2, vregs(base2))
add("align_8p4_rsq64", "mix with v_rsq_f32_e64, start = 4 mod 8", [".p2align 3", "s_nop 0"] + base2, vregs(base2))



# ---- the pair stream with every instruction in a 64-bit encoding, at both placements
def sub_e64(ops, r, t, d):
    return [
        "v_sub_f32_e64 v%d, s20, v%d" % (d[0], r["xi"]), "v_sub_f32_e64 v%d, s21, v%d" % (d[1], r["yi"]),
        "v_sub_f32_e64 v%d, s22, v%d" % (d[2], r["zi"]), "v_fma_f32 v%d, v%d, v%d, s23" % (t, d[2], d[2]),
        "v_fma_f32 v%d, v%d, v%d, v%d" % (t, d[1], d[1], t), "v_fma_f32 v%d, v%d, v%d, v%d" % (t, d[0], d[0], t),
        "v_rsq_f32_e64 v%d, v%d" % (t, t), "v_mul_f32_e64 v%d, v%d, v%d" % (r["u"], t, t), "v_mul_f32_e64 v%d, v%d, v%d" % (t, t, r["u"]),
        "v_fma_f32 v%d, v%d, v%d, v%d" % (r["ax"], d[0], t, r["ax"]), "v_fma_f32 v%d, v%d, v%d, v%d" % (r["ay"], d[1], t, r["ay"]),
        "v_fma_f32 v%d, v%d, v%d, v%d" % (r["az"], d[2], t, r["az"])]


for C in (1, 2, 4):
    ins = pairsafe(C, sub_e64, nj=4 if C == 1 else 2)
    add("e64_c%d_a0" % C, "pair stream, %d chains, all 64-bit encodings, start 0 mod 8" % C, [".p2align 3"] + ins, vregs(ins))
    add("e64_c%d_a4" % C, "pair stream, %d chains, all 64-bit encodings, start 4 mod 8" % C, [".p2align 3", "s_nop 0"] + ins, vregs(ins))
    ins32 = pairsafe(C, sub_fma_sgpr_eps, nj=4 if C == 1 else 2)   # VOP2 everywhere except the eps fma (8 bytes)
    add("mixenc_c%d_a0" % C, "pair stream, %d chains, compiler-like encodings (VOP2 + one VOP3), start 0 mod 8" % C, [".p2align 3"] + ins32, vregs(ins32))
    add("mixenc_c%d_a4" % C, "same, start 4 mod 8" % (), [".p2align 3", "s_nop 0"] + ins32, vregs(ins32))



# ---- placement sweep: the loop body starts (4k + loop header) bytes into a 64-byte line, k = 0..15.
# PRE:<text> instructions go into an asm statement executed once BEFORE the loop.
def placed(name, desc, ins):
    for k in range(16):
        pre = [".p2align 6"] + ["s_nop 0"] * k
        streams.append((name + "_k%02d" % k, desc + ", %d bytes past a 64-byte line + header" % (4 * k), ins,
                        len(ins), sorted(set(vregs(ins)), key=lambda r: int(r[1:])), pre))


mix8 = [i.replace("v_rsq_f32 v81, v81", "v_rsq_f32_e64 v81, v81") for i in mix_vop3()]
placed("pl_mix8", "mix, all 8-byte encodings", mix8)
placed("pl_mix", "mix, 8-byte fma + 4-byte rsq", mix_vop3())
placed("pl_e64c1", "e64 pair stream, 1 chain", pairsafe(1, sub_e64, nj=4))
placed("pl_e64c2", "e64 pair stream, 2 chains", pairsafe(2, sub_e64, nj=2))



# ---- the kernel's real loop body (tools/gen_force_loop.py) as a stream: distinct SGPRs per body, software-pipelined order
import importlib.util as _ilu
_spec = _ilu.spec_from_file_location("gen_force_loop", os.path.join(ROOT, "tools", "gen_force_loop.py"))
_gfl = _ilu.module_from_spec(_spec)
_spec.loader.exec_module(_gfl)


body = []
for _k in range(8):
    body += _gfl.body(_k, 36 if _k < 4 else 52, _k & 3)
for k in (0, 1):
    pre = [".p2align 6"] + ["s_nop 0"] * k
    streams.append(("real_loop_k%d" % k, "the ISA kernel's loop body (96 VALU, s36..s67 sources, pinned registers), phase %d" % k, body,
                    len([i for i in body if i.startswith("v_")]), sorted(set(vregs(body)), key=lambda r: int(r[1:])), pre))


# the same loop WITH its scalar loads and waits (sources read from the kernarg segment, s[0:1]: garbage values, timing only)
body_ld = (["s_waitcnt lgkmcnt(0)", "s_nop 0", "s_load_dwordx16 s[52:67], s[0:1], 0x0"] + body[:48] + ["s_nop 0", "s_nop 0"] +
           ["s_waitcnt lgkmcnt(0)", "s_nop 0", "s_load_dwordx16 s[36:51], s[0:1], 0x0"] + body[48:])
streams.append(("real_loop_loads_k0", "the ISA kernel's loop body + its two s_load_dwordx16 and waits per 8 bodies", body_ld,
                len([i for i in body_ld if i.startswith("v_")]), sorted(set(vregs(body_ld)), key=lambda r: int(r[1:])), [".p2align 6"]))
# loads only, no waits (the SGPRs are overwritten while being read: timing only)
body_ld2 = (["s_nop 0", "s_nop 0", "s_load_dwordx16 s[52:67], s[0:1], 0x0"] + body[:48] + ["s_nop 0", "s_nop 0"] +
            ["s_nop 0", "s_nop 0", "s_load_dwordx16 s[36:51], s[0:1], 0x0"] + body[48:])
streams.append(("real_loop_loadsnowait_k0", "same without the waits", body_ld2,
                len([i for i in body_ld2 if i.startswith("v_")]), sorted(set(vregs(body_ld2)), key=lambda r: int(r[1:])), [".p2align 6"]))
# loads into SGPRs nothing reads
body_ld3 = (["s_waitcnt lgkmcnt(0)", "s_nop 0", "s_load_dwordx16 s[72:87], s[0:1], 0x0"] + body[:48] + ["s_nop 0", "s_nop 0"] +
            ["s_waitcnt lgkmcnt(0)", "s_nop 0", "s_load_dwordx16 s[72:87], s[0:1], 0x0"] + body[48:])
streams.append(("real_loop_loadsscratch_k0", "same, loads land in SGPRs nothing reads", body_ld3,
                len([i for i in body_ld3 if i.startswith("v_")]), sorted(set(vregs(body_ld3)), key=lambda r: int(r[1:])), [".p2align 6"]))


# ---- legal orderings of the single-chain pair stream (>= 1 instruction between v_rsq_f32 and its consumer)
TT = [20, 24]; UU = 22; DD = [(21, 23, 25), (27, 29, 31)]


def ops_for(k):
    t, (dx, dy, dz) = TT[k & 1], DD[k & 1]
    sx = 36 + 4 * (k % 8)
    return dict(
        Sx="v_sub_f32_e64 v%d, s%d, v0" % (dx, sx), Sy="v_sub_f32_e64 v%d, s%d, v1" % (dy, sx + 1), Sz="v_sub_f32_e64 v%d, s%d, v2" % (dz, sx + 2),
        F1="v_fma_f32 v%d, v%d, v%d, s23" % (t, dz, dz), F2="v_fma_f32 v%d, v%d, v%d, v%d" % (t, dy, dy, t), F3="v_fma_f32 v%d, v%d, v%d, v%d" % (t, dx, dx, t),
        R="v_rsq_f32_e64 v%d, v%d" % (t, t), M1="v_mul_f32_e64 v%d, v%d, v%d" % (UU, t, t), M2="v_mul_f32_e64 v%d, v%d, v%d" % (t, t, UU),
        Ax="v_fma_f32 v4, v%d, v%d, v4" % (dx, t), Ay="v_fma_f32 v5, v%d, v%d, v5" % (dy, t), Az="v_fma_f32 v6, v%d, v%d, v6" % (dz, t))


def order_stream(pattern, nbodies=8):
    """pattern: list of (op, body offset) executed for k = 0..nbodies-1 in a steady-state loop (offsets wrap around)"""
    ins = []
    for k in range(nbodies):
        for op, off in pattern:
            ins.append(ops_for((k + off) % nbodies)[op])
    return ins


ORDERS = {
    # name: steady-state pattern per body k
    "ord_serial_viol": [("Sx", 0), ("Sy", 0), ("Sz", 0), ("F1", 0), ("F2", 0), ("F3", 0), ("R", 0), ("M1", 0), ("M2", 0), ("Ax", 0), ("Ay", 0), ("Az", 0)],
    "ord_sep1": [("R", 0), ("Sx", 1), ("M1", 0), ("M2", 0), ("Ax", 0), ("Ay", 0), ("Az", 0), ("Sy", 1), ("Sz", 1), ("F1", 1), ("F2", 1), ("F3", 1)],
    "ord_sep2": [("R", 0), ("Sx", 1), ("Sy", 1), ("M1", 0), ("M2", 0), ("Ax", 0), ("Ay", 0), ("Az", 0), ("Sz", 1), ("F1", 1), ("F2", 1), ("F3", 1)],
    "ord_sep3": [("R", 0), ("Sx", 1), ("Sy", 1), ("Sz", 1), ("M1", 0), ("M2", 0), ("Ax", 0), ("Ay", 0), ("Az", 0), ("F1", 1), ("F2", 1), ("F3", 1)],
    "ord_sep6": [("R", 0), ("Sx", 1), ("Sy", 1), ("Sz", 1), ("F1", 1), ("F2", 1), ("F3", 1), ("M1", 0), ("M2", 0), ("Ax", 0), ("Ay", 0), ("Az", 0)],
    # accumulate fmas of the previous body sit between the rsq and its consumer
    "ord_defA": [("Sx", 0), ("Sy", 0), ("Sz", 0), ("F1", 0), ("F2", 0), ("F3", 0), ("R", 0), ("Ax", -1), ("Ay", -1), ("Az", -1), ("M1", 0), ("M2", 0)],
    # rsq early in the body, everything else of the neighbours around it
    "ord_rsq_first": [("R", 0), ("Ax", -1), ("Ay", -1), ("Az", -1), ("Sx", 1), ("Sy", 1), ("Sz", 1), ("M1", 0), ("M2", 0), ("F1", 1), ("F2", 1), ("F3", 1)],
    "ord_rsq_mid": [("Sx", 1), ("Sy", 1), ("Sz", 1), ("R", 0), ("F1", 1), ("F2", 1), ("Ax", -1), ("Ay", -1), ("Az", -1), ("M1", 0), ("M2", 0), ("F3", 1)],
}
for nm, pat in ORDERS.items():
    body = order_stream(pat)
    for k in (0, 1):
        pre = [".p2align 6"] + ["s_nop 0"] * k
        streams.append((nm + "_k%d" % k, "single chain, order " + " ".join("%s%+d" % (o, d) if d else o for o, d in pat), body, len(body),
                        sorted(set(vregs(body)), key=lambda r: int(r[1:])), pre))


# ---- 3-stage software pipeline, single chain, good phase: slot k = S,F of body k | rsq of body k-1 | M,A of body k-2
T3 = [20, 24, 26]
D3 = [(21, 23, 25), (27, 29, 31), (33, 35, 37)]


def ops3(k):
    t, (dx, dy, dz) = T3[k % 3], D3[k % 3]
    sx = 36 + 4 * (k % 8)
    return dict(
        S=["v_sub_f32_e64 v%d, s%d, v8" % (dx, sx), "v_sub_f32_e64 v%d, s%d, v9" % (dy, sx + 1), "v_sub_f32_e64 v%d, s%d, v10" % (dz, sx + 2)],
        F=["v_fma_f32 v%d, v%d, v%d, s34" % (t, dz, dz), "v_fma_f32 v%d, v%d, v%d, v%d" % (t, dy, dy, t), "v_fma_f32 v%d, v%d, v%d, v%d" % (t, dx, dx, t)],
        R=["v_rsq_f32_e64 v%d, v%d" % (t, t)],
        M=["v_mul_f32_e64 v22, v%d, v%d" % (t, t), "v_mul_f32_e64 v%d, v%d, v22" % (t, t)],
        A=["v_fma_f32 v12, v%d, v%d, v12" % (dx, t), "v_fma_f32 v13, v%d, v%d, v13" % (dy, t), "v_fma_f32 v14, v%d, v%d, v14" % (dz, t)])


for nm, order in (("p3_SFRMA", "SFRMA"), ("p3_RSFMA", "RSFMA"), ("p3_SFMAR", "SFMAR"), ("p3_MASFR", "MASFR")):
    body3 = []
    for k in range(24):            # 24 bodies = lcm(3 register sets, 8 sources)
        part = {"S": ops3(k)["S"], "F": ops3(k)["F"], "R": ops3(k - 1)["R"], "M": ops3(k - 2)["M"], "A": ops3(k - 2)["A"]}
        for ch in order:
            body3 += part[ch]
    check_banks(body3)
    for kk in (0, 1):
        pre = [".p2align 6"] + ["s_nop 0"] * kk
        streams.append(("%s_k%d" % (nm, kk), "3-stage pipeline, single chain, slot order %s" % order, body3, len(body3),
                        sorted(set(vregs(body3)), key=lambda r: int(r[1:])), pre))


def main():
    with open(OUT, "w") as f:
        f.write("// GENERATED by tools/gen_streams.py — do not edit.\n")
        f.write("// name, description, instruction count, asm text, clobber list\n")
        for entry in streams:
            name, desc, ins, n, regs = entry[:5]
            pre = entry[5] if len(entry) > 5 else []
            text = "\\n\\t".join(ins)
            clob = ", ".join('"%s"' % r for r in regs)
            # well-conditioned operands: every VGPR 0.5 (+ a little per register), the SGPR sources 0.25 / -0.125 / 0.75
            init = ["v_mov_b32 %s, 0x%08x" % (r, 0x3F000000 + 4096 * int(r[1:])) for r in regs]
            init += ["s_mov_b32 s20, 0.25", "s_mov_b32 s21, 0xbe000000", "s_mov_b32 s22, 0x3f400000", "s_mov_b32 s23, 0x3089705f"]
            init += ["s_mov_b32 s%d, 0x%08x" % (r, 0x3E000000 + 65536 * r) for r in range(36, 68)] + ["s_mov_b32 s34, 0x3089705f"]
            itext = "\\n\\t".join(init)
            ptext = "\\n\\t".join(pre)
            f.write("STREAM(%s, \"%s\", %d, \"%s\", \"%s\", \"%s\", %s)\n" % (name, desc, n, text, itext, ptext, clob))
    print("wrote %s: %d streams" % (OUT, len(streams)))


if __name__ == "__main__":
    main()
