#!/usr/bin/env python3
"""Generate mini-nbody_amd/csrc/force_loop_gfx950.inc — the hand-scheduled inner loop of the fp32 force kernel.

Why hand-written ISA (measurements: profiles/r01_microbench_streams.txt, tools/gen_streams.py):
  * every instruction of the pair interaction issues at 2 cycles per wave64 EXCEPT that the cost of the
    quarter-rate v_rsq_f32 depends on code placement: with uniform 64-bit encodings the stream costs 2.6
    cycles/instruction (31.5 per pair) when instructions start at one 4-byte phase of an 8-byte window and 3.9
    (47 per pair) at the other; hipcc's mix of 32- and 64-bit encodings drifts between the two (34-36 per pair);
  * a VOP3 instruction whose three VGPR sources have the same parity (register number mod 2) costs 4 cycles instead
    of 2: temporaries are pinned to physical registers, t/u even, dx/dy/dz odd, so no instruction does that;
  * gfx940+ needs one wait state between a transcendental and a VALU instruction that reads its result
    (hipcc does not pad inline asm): the loop is software-pipelined by one stage so that three independent
    subtractions (or two scalar instructions) sit between every v_rsq_f32 and its first consumer.

Loop shape (one wave, one body i per lane, sources delivered as wave-uniform scalar loads):
    A = s[36:51], B = s[52:67]: two buffers of 4 bodies {x,y,z,w}; pointer s[68:69], group counter s70, stride s71
    prologue: load A
    loop:  wait A | load B | 4 bodies from A (48 VALU) | advance pointer | wait B | load A (next group) | 4 bodies from B
The arithmetic and its order per body are exactly pair_f32<0> of nbody_kernels.hpp (sources ascending), so the
result is bit-identical to the C++ kernels.
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "mini-nbody_amd", "csrc", "force_loop_gfx950.inc")

T, U = 20, 22                      # even
DSETS = [(21, 23, 25), (27, 29, 31)]   # odd
A_BASE, B_BASE = 36, 52
PTR, CNT, STRIDE = 68, 70, 71


def S(b, sbase, dset):
    dx, dy, dz = DSETS[dset]
    s = sbase + 4 * b
    return ["v_sub_f32_e64 v%d, s%d, %%[xi]" % (dx, s), "v_sub_f32_e64 v%d, s%d, %%[yi]" % (dy, s + 1),
            "v_sub_f32_e64 v%d, s%d, %%[zi]" % (dz, s + 2)]


def F(dset):
    dx, dy, dz = DSETS[dset]
    return ["v_fma_f32 v%d, v%d, v%d, %%[eps]" % (T, dz, dz), "v_fma_f32 v%d, v%d, v%d, v%d" % (T, dy, dy, T),
            "v_fma_f32 v%d, v%d, v%d, v%d" % (T, dx, dx, T)]


def R():
    return ["v_rsq_f32_e64 v%d, v%d" % (T, T)]


def M(dset):
    dx, dy, dz = DSETS[dset]
    return ["v_mul_f32_e64 v%d, v%d, v%d" % (U, T, T), "v_mul_f32_e64 v%d, v%d, v%d" % (T, T, U),
            "v_fma_f32 %%[ax], v%d, v%d, %%[ax]" % (dx, T), "v_fma_f32 %%[ay], v%d, v%d, %%[ay]" % (dy, T),
            "v_fma_f32 %%[az], v%d, v%d, %%[az]" % (dz, T)]


def half(sbase, filler):
    """4 bodies from one SGPR buffer; `filler` = the two 4-byte scalar instructions that separate the last
    v_rsq_f32 from its consumer (wait state) and keep the 8-byte phase."""
    out = []
    out += S(0, sbase, 0) + F(0) + R()
    for b in (1, 2, 3):
        d, pd = b & 1, (b - 1) & 1
        out += S(b, sbase, d) + M(pd) + F(d) + R()
    out += filler + M(1)
    return out


def half_serial(sbase):
    """no software pipelining: S F R nop nop M per body (the two s_nop are the trans->VALU wait state, 8 bytes)"""
    out = []
    for b in range(4):
        out += S(b, sbase, 0) + F(0) + R() + ["s_nop 0", "s_nop 0"] + M(0)
    return out


def build_debug(kind):
    """timing-only / alternative loops selected by NBODY_OPT_ISA_PHASE >= 2 (bring-up, not used by default)"""
    ins = []
    ins.append("s_mov_b64 s[%d:%d], %%[p]" % (PTR, PTR + 1))
    ins.append("s_mov_b32 s%d, %%[groups]" % CNT)
    ins.append("s_movk_i32 s%d, 0x80" % STRIDE)
    ins.append("s_load_dwordx16 s[%d:%d], s[%d:%d], 0x0" % (A_BASE, A_BASE + 15, PTR, PTR + 1))
    if kind == "noreload":
        ins.append("s_load_dwordx16 s[%d:%d], s[%d:%d], 0x40" % (B_BASE, B_BASE + 15, PTR, PTR + 1))
        ins.append("s_waitcnt lgkmcnt(0)")
    ins += [".p2align 3", "s_nop 0", "1:"]
    if kind == "noreload":       # WRONG RESULTS (same 8 sources every time): prices the scalar loads
        ins += ["s_nop 0", "s_nop 0"]
        ins += half(A_BASE, ["s_add_u32 s%d, s%d, s%d" % (PTR, PTR, STRIDE), "s_addc_u32 s%d, s%d, 0" % (PTR + 1, PTR + 1)])
        ins += ["s_nop 0", "s_sub_u32 s%d, s%d, 1" % (CNT, CNT)]
        ins += half(B_BASE, ["s_nop 0", "s_nop 0"])
    elif kind == "serial":
        ins += ["s_waitcnt lgkmcnt(0)", "s_nop 0"]
        ins.append("s_load_dwordx16 s[%d:%d], s[%d:%d], 0x40" % (B_BASE, B_BASE + 15, PTR, PTR + 1))
        ins += half_serial(A_BASE)
        ins += ["s_add_u32 s%d, s%d, s%d" % (PTR, PTR, STRIDE), "s_addc_u32 s%d, s%d, 0" % (PTR + 1, PTR + 1)]
        ins += ["s_waitcnt lgkmcnt(0)", "s_sub_u32 s%d, s%d, 1" % (CNT, CNT)]
        ins.append("s_load_dwordx16 s[%d:%d], s[%d:%d], 0x0" % (A_BASE, A_BASE + 15, PTR, PTR + 1))
        ins += half_serial(B_BASE)
    ins += ["s_cmp_lg_u32 s%d, 0" % CNT, "s_cbranch_scc1 1b", "s_waitcnt lgkmcnt(0)"]
    return ins


T2 = [20, 24]                       # two d2/inv/inv3 registers (even), alternating with the body index
# live-in / live-out values are copied to fixed registers too, so that the loop's text (and with it register
# parity and code bytes) does not depend on hipcc's allocation around the asm statement
PIN = {"%[xi]": "v8", "%[yi]": "v9", "%[zi]": "v10", "%[ax]": "v12", "%[ay]": "v13", "%[az]": "v14", "%[eps]": "s34"}
PIN_VGPRS = [8, 9, 10, 12, 13, 14]


def pinned(ins):
    """wrap a loop: copy operands into the pinned registers, rewrite the body, copy the sums back"""
    head = ["v_mov_b32 v8, %[xi]", "v_mov_b32 v9, %[yi]", "v_mov_b32 v10, %[zi]", "v_mov_b32 v12, %[ax]", "v_mov_b32 v13, %[ay]",
            "v_mov_b32 v14, %[az]", "s_mov_b32 s34, %[eps]"]
    body = []
    for i in ins:
        for k, v in PIN.items():
            i = i.replace(k, v)
        body.append(i)
    tail = ["v_mov_b32 %[ax], v12", "v_mov_b32 %[ay], v13", "v_mov_b32 %[az], v14"]
    return head + body + tail



def body_defA(k, sbase, b):
    """ord_defA of tools/gen_streams.py (2.63 cycles/instruction in the stream microbenchmark):
         S S S F F F R | A A A of the previous body | M M
    The previous body's three accumulating fmas are the wait state between v_rsq_f32 and its consumer."""
    t, tp = T2[k & 1], T2[(k - 1) & 1]
    dx, dy, dz = DSETS[k & 1]
    px, py, pz = DSETS[(k - 1) & 1]
    s0 = sbase + 4 * b
    return [
        "v_sub_f32_e64 v%d, s%d, %%[xi]" % (dx, s0), "v_sub_f32_e64 v%d, s%d, %%[yi]" % (dy, s0 + 1), "v_sub_f32_e64 v%d, s%d, %%[zi]" % (dz, s0 + 2),
        "v_fma_f32 v%d, v%d, v%d, %%[eps]" % (t, dz, dz), "v_fma_f32 v%d, v%d, v%d, v%d" % (t, dy, dy, t), "v_fma_f32 v%d, v%d, v%d, v%d" % (t, dx, dx, t),
        "v_rsq_f32_e64 v%d, v%d" % (t, t),
        "v_fma_f32 %%[ax], v%d, v%d, %%[ax]" % (px, tp), "v_fma_f32 %%[ay], v%d, v%d, %%[ay]" % (py, tp), "v_fma_f32 %%[az], v%d, v%d, %%[az]" % (pz, tp),
        "v_mul_f32_e64 v%d, v%d, v%d" % (U, t, t), "v_mul_f32_e64 v%d, v%d, v%d" % (t, t, U)]


def build_defA(phase_nop):
    ins = []
    ins.append("s_mov_b64 s[%d:%d], %%[p]" % (PTR, PTR + 1))
    ins.append("s_mov_b32 s%d, %%[groups]" % CNT)
    ins.append("s_movk_i32 s%d, 0x80" % STRIDE)
    ins.append("s_load_dwordx16 s[%d:%d], s[%d:%d], 0x0" % (A_BASE, A_BASE + 15, PTR, PTR + 1))
    # "previous body" of the very first body: d = 0, inv3 = 0 -> fma(0, 0, acc) leaves acc as it is
    px, py, pz = DSETS[1]
    ins += ["v_mov_b32 v%d, 0" % r for r in (px, py, pz, T2[1])]
    if isinstance(phase_nop, bool):
        ins.append(".p2align 3")
        if phase_nop:
            ins.append("s_nop 0")
    else:                       # deterministic placement: loop head = 4 * phase_nop bytes past a 64-byte line
        ins.append(".p2align 6")
        ins += ["s_nop 0"] * int(phase_nop)
    ins.append("1:")
    ins += ["s_waitcnt lgkmcnt(0)", "s_sub_u32 s%d, s%d, 1" % (CNT, CNT)]
    ins.append("s_load_dwordx16 s[%d:%d], s[%d:%d], 0x40" % (B_BASE, B_BASE + 15, PTR, PTR + 1))
    for b in range(4):
        ins += body_defA(b, A_BASE, b)
    ins += ["s_add_u32 s%d, s%d, s%d" % (PTR, PTR, STRIDE), "s_addc_u32 s%d, s%d, 0" % (PTR + 1, PTR + 1)]
    ins += ["s_waitcnt lgkmcnt(0)", "s_nop 0"]
    ins.append("s_load_dwordx16 s[%d:%d], s[%d:%d], 0x0" % (A_BASE, A_BASE + 15, PTR, PTR + 1))
    for b in range(4):
        ins += body_defA(4 + b, B_BASE, b)
    ins += ["s_cmp_lg_u32 s%d, 0" % CNT, "s_cbranch_scc1 1b"]
    # drain: the accumulate of the last body (body 7 -> d set 1, t register 1), then retire the unused prefetch
    ins += ["v_fma_f32 %%[ax], v%d, v%d, %%[ax]" % (px, T2[1]), "v_fma_f32 %%[ay], v%d, v%d, %%[ay]" % (py, T2[1]),
            "v_fma_f32 %%[az], v%d, v%d, %%[az]" % (pz, T2[1])]
    ins += ["s_waitcnt lgkmcnt(0)"]
    return ins


def build(phase_nop):
    ins = []
    ins.append("s_mov_b64 s[%d:%d], %%[p]" % (PTR, PTR + 1))
    ins.append("s_mov_b32 s%d, %%[groups]" % CNT)
    ins.append("s_movk_i32 s%d, 0x80" % STRIDE)
    ins.append("s_load_dwordx16 s[%d:%d], s[%d:%d], 0x0" % (A_BASE, A_BASE + 15, PTR, PTR + 1))
    ins.append(".p2align 3")
    if phase_nop:
        ins.append("s_nop 0")
    ins.append("1:")
    ins += ["s_waitcnt lgkmcnt(0)", "s_nop 0"]
    ins.append("s_load_dwordx16 s[%d:%d], s[%d:%d], 0x40" % (B_BASE, B_BASE + 15, PTR, PTR + 1))
    ins += half(A_BASE, ["s_add_u32 s%d, s%d, s%d" % (PTR, PTR, STRIDE), "s_addc_u32 s%d, s%d, 0" % (PTR + 1, PTR + 1)])
    ins += ["s_waitcnt lgkmcnt(0)", "s_sub_u32 s%d, s%d, 1" % (CNT, CNT)]
    ins.append("s_load_dwordx16 s[%d:%d], s[%d:%d], 0x0" % (A_BASE, A_BASE + 15, PTR, PTR + 1))
    ins += half(B_BASE, ["s_nop 0", "s_nop 0"])
    ins += ["s_cmp_lg_u32 s%d, 0" % CNT, "s_cbranch_scc1 1b"]
    ins += ["s_waitcnt lgkmcnt(0)"]      # the prefetched (unused) A of the group after the last
    return ins


def build_variant(v):
    """Numbered loops for NBODY_OPT_ISA_PHASE (0/1 are the product phases; >= 2 are bring-up experiments).
    All use the deferred-accumulate order; what differs is how the scalar loads are handled."""
    if v == 0:                  # the slow code-placement phase (evidence for DESIGN.md; never the default)
        return pinned(build_defA(14))
    if v == 1:                  # THE PRODUCT LOOP: loop head 60 bytes past a 64-byte line
        return pinned(build_defA(15))
    if 6 <= v < 14:             # placement sweep: loop head at 64-byte line + 4 * (2m + 1) bytes, m = v - 6
        return build_defA(2 * (v - 6) + 1)
    ins = []
    ins.append("s_mov_b64 s[%d:%d], %%[p]" % (PTR, PTR + 1))
    ins.append("s_mov_b32 s%d, %%[groups]" % CNT)
    ins.append("s_movk_i32 s%d, 0x80" % STRIDE)
    ins.append("s_load_dwordx16 s[%d:%d], s[%d:%d], 0x0" % (A_BASE, A_BASE + 15, PTR, PTR + 1))
    px, py, pz = DSETS[1]
    ins += ["v_mov_b32 v%d, 0" % r for r in (px, py, pz, T2[1])]
    if v in (2, 3):
        ins.append("s_load_dwordx16 s[%d:%d], s[%d:%d], 0x40" % (B_BASE, B_BASE + 15, PTR, PTR + 1))
        ins.append("s_waitcnt lgkmcnt(0)")
    ins += [".p2align 3", "s_nop 0", "1:"]
    if v == 2:      # TIMING ONLY: no loads, no waits in the loop
        ins += ["s_nop 0", "s_sub_u32 s%d, s%d, 1" % (CNT, CNT)]
        for b in range(4):
            ins += body_defA(b, A_BASE, b)
        ins += ["s_add_u32 s%d, s%d, s%d" % (PTR, PTR, STRIDE), "s_addc_u32 s%d, s%d, 0" % (PTR + 1, PTR + 1)]
        for b in range(4):
            ins += body_defA(4 + b, B_BASE, b)
    elif v == 3:    # TIMING ONLY: loads into SCRATCH SGPRs (s72..s87) that nothing reads, with the waits
        ins += ["s_waitcnt lgkmcnt(0)", "s_sub_u32 s%d, s%d, 1" % (CNT, CNT)]
        ins.append("s_load_dwordx16 s[72:87], s[%d:%d], 0x40" % (PTR, PTR + 1))
        for b in range(4):
            ins += body_defA(b, A_BASE, b)
        ins += ["s_add_u32 s%d, s%d, s%d" % (PTR, PTR, STRIDE), "s_addc_u32 s%d, s%d, 0" % (PTR + 1, PTR + 1)]
        ins += ["s_waitcnt lgkmcnt(0)", "s_nop 0"]
        ins.append("s_load_dwordx16 s[72:87], s[%d:%d], 0x0" % (PTR, PTR + 1))
        for b in range(4):
            ins += body_defA(4 + b, B_BASE, b)
    elif v == 4:    # loads issued in the MIDDLE of the other half (two bodies in), waits at the half boundaries
        ins += ["s_waitcnt lgkmcnt(0)", "s_sub_u32 s%d, s%d, 1" % (CNT, CNT)]
        for b in range(4):
            if b == 2:
                ins.append("s_load_dwordx16 s[%d:%d], s[%d:%d], 0x40" % (B_BASE, B_BASE + 15, PTR, PTR + 1))
            ins += body_defA(b, A_BASE, b)
        ins += ["s_add_u32 s%d, s%d, s%d" % (PTR, PTR, STRIDE), "s_addc_u32 s%d, s%d, 0" % (PTR + 1, PTR + 1)]
        ins += ["s_waitcnt lgkmcnt(0)", "s_nop 0"]
        for b in range(4):
            if b == 2:
                ins.append("s_load_dwordx16 s[%d:%d], s[%d:%d], 0x0" % (A_BASE, A_BASE + 15, PTR, PTR + 1))
            ins += body_defA(4 + b, B_BASE, b)
    elif v == 5:    # eight s_load_dwordx4 (one body each) instead of two x16, issued one body ahead of use
        # buffers: body b of the group lives in s[36+4b : 39+4b]; body b+8's load is issued right after body b's subs
        ins += ["s_waitcnt lgkmcnt(0)", "s_sub_u32 s%d, s%d, 1" % (CNT, CNT)]
        ins.append("s_load_dwordx16 s[%d:%d], s[%d:%d], 0x40" % (B_BASE, B_BASE + 15, PTR, PTR + 1))
        for b in range(4):
            ins += body_defA(b, A_BASE, b)
        ins += ["s_add_u32 s%d, s%d, s%d" % (PTR, PTR, STRIDE), "s_addc_u32 s%d, s%d, 0" % (PTR + 1, PTR + 1)]
        ins += ["s_waitcnt lgkmcnt(0)", "s_nop 0"]
        for q in range(4):
            ins.append("s_load_dwordx4 s[%d:%d], s[%d:%d], 0x%x" % (A_BASE + 4 * q, A_BASE + 4 * q + 3, PTR, PTR + 1, 16 * q))
        for b in range(4):
            ins += body_defA(4 + b, B_BASE, b)
    ins += ["s_cmp_lg_u32 s%d, 0" % CNT, "s_cbranch_scc1 1b"]
    ins += ["v_fma_f32 %%[ax], v%d, v%d, %%[ax]" % (px, T2[1]), "v_fma_f32 %%[ay], v%d, v%d, %%[ay]" % (py, T2[1]),
            "v_fma_f32 %%[az], v%d, v%d, %%[az]" % (pz, T2[1])]
    ins += ["s_waitcnt lgkmcnt(0)"]
    return ins


def build_g16(pad, group=16):
    """16 (or 8) bodies per iteration, sources addressed as base s[68:69] + byte offset s70 (+ immediate), one 32-bit
    add per iteration, count-down fused with the branch (s_add_u32 cnt, cnt, -1 sets SCC while cnt was >= 1).
    Buffers: A = s[36:67] (8 bodies, two x16 loads), B = s[72:103]?  SGPRs end at s101, so B = s[72:99] + ... does not
    fit: use A = s[36:67], B = s[4..]?  -> keep to what fits: group 16 = buffers of 8 bodies at s[36:67] and s[68:99],
    pointer s[100:101], offset and counter in the two operands the compiler supplies (%[off], %[cnt])."""
    a_base, b_base = 36, 68
    half_bodies = group // 2
    ins = []
    for q in range(half_bodies // 4):
        ins.append("s_load_dwordx16 s[%d:%d], %%[p], %%[off] offset:0x%x" % (a_base + 16 * q, a_base + 16 * q + 15, 64 * q))
    px, py, pz = DSETS[1]
    ins += ["v_mov_b32 v%d, 0" % r for r in (px, py, pz, T2[1])]
    ins.append(".p2align 6")
    ins += ["s_nop 0"] * pad
    ins.append("1:")
    stride = 16 * group
    # half 1: wait for A, fetch B, compute A
    ins += ["s_waitcnt lgkmcnt(0)", "s_nop 0"]
    for q in range(half_bodies // 4):
        ins.append("s_load_dwordx16 s[%d:%d], %%[p], %%[off] offset:0x%x" % (b_base + 16 * q, b_base + 16 * q + 15, 16 * half_bodies + 64 * q))
    for b in range(half_bodies):
        ins += body_defA(b, a_base, b)
    # half 2: advance, wait for B, fetch next A, compute B
    ins += ["s_add_u32 %%[off], %%[off], 0x%x" % stride]          # 8 bytes (32-bit literal)
    ins += ["s_waitcnt lgkmcnt(0)", "s_nop 0"]
    for q in range(half_bodies // 4):
        ins.append("s_load_dwordx16 s[%d:%d], %%[p], %%[off] offset:0x%x" % (a_base + 16 * q, a_base + 16 * q + 15, 64 * q))
    for b in range(half_bodies):
        ins += body_defA(half_bodies + b, b_base, b)
    ins += ["s_add_u32 %[cnt], %[cnt], -1", "s_cbranch_scc1 1b"]
    last = (group - 1) & 1
    lx, ly, lz = DSETS[last]
    ins += ["v_fma_f32 %%[ax], v%d, v%d, %%[ax]" % (lx, T2[last]), "v_fma_f32 %%[ay], v%d, v%d, %%[ay]" % (ly, T2[last]),
            "v_fma_f32 %%[az], v%d, v%d, %%[az]" % (lz, T2[last])]
    ins += ["s_waitcnt lgkmcnt(0)"]
    return ins


N_VARIANTS = 14


def clobbers():
    regs = ["v%d" % r for r in [T, U] + T2 + PIN_VGPRS + [x for d in DSETS for x in d]]
    regs = sorted(set(regs), key=lambda r: int(r[1:]))
    regs += ["s34"] + ["s%d" % r for r in range(A_BASE, 88)]
    return regs + ["scc", "memory"]


def main():
    with open(OUT, "w") as f:
        f.write("// GENERATED by tools/gen_force_loop.py — do not edit.  See that file for the why.\n")
        for v in range(N_VARIANTS):
            f.write("#define NB_FORCE_LOOP_V%d \"%s\"\n" % (v, "\\n\\t".join(build_variant(v))))
        f.write("#define NB_FORCE_LOOP_NVARIANTS %d\n" % N_VARIANTS)
        for k, (pad, group) in enumerate(((1, 16), (3, 16), (9, 16), (1, 8))):
            f.write("#define NB_FORCE_LOOP_G%d \"%s\"\n" % (k, "\\n\\t".join(build_g16(pad, group))))
            f.write("#define NB_FORCE_LOOP_G%d_GROUP %d\n" % (k, group))
        g_regs = ["v%d" % r for r in sorted(set([U] + T2 + [x for d in DSETS for x in d]))] + ["s%d" % r for r in range(36, 100)] + ["scc", "memory"]
        f.write("#define NB_FORCE_LOOP_G_CLOBBERS %s\n" % ", ".join('"%s"' % c for c in g_regs))
        f.write("#define NB_FORCE_LOOP_CLOBBERS %s\n" % ", ".join('"%s"' % c for c in clobbers()))
        f.write("#define NB_FORCE_LOOP_GROUP 8\n")
    n_valu = len([i for i in build_defA(15) if i.startswith("v_")])
    print("wrote %s (%d VALU instructions per group of 8 bodies)" % (OUT, n_valu))


if __name__ == "__main__":
    main()
