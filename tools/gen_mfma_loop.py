#!/usr/bin/env python3
"""Generate mini_nbody_amd/csrc/force_loop_mfma_gfx950.inc — the fp32 force loop whose three coordinate differences per
pair are produced by the matrix pipe.

Why.  The force loop is VALU-issue-bound: 11 full-rate instructions + v_rsq_f32 per pair (tools/gen_force_loop.py,
DESIGN.md §3.1) while the matrix cores idle.  v_mfma_f32_32x32x2_f32 with A[m] = {r_j[m], 1} and B[.][n] = {1, -r_i[n]}
returns D[m][n] = r_j[m]*1 + 1*(-r_i[n]): two exact products and ONE rounding of their sum — the IEEE subtraction
S/dxy.vhd:94-98 and S/dzsoft.vhd:186-187 ask for, bit for bit (build/microbench_mfma: 4 M operand pairs over the whole
exponent range, subnormals, infinities; the one deviation is (-0) - (+0) = +0 instead of -0, the accumulator's +0 being
part of the sum) — 1024 differences per instruction and none of the VALU's issue slots.  What is left on the VALU per
pair: 3 fma (d2), v_rsq_f32, 2 mul, 3 fma = 8 full-rate + 1 quarter-rate instruction, all operands in VGPRs.

Shape.  A wave owns 32 rows (lane n and lane n + 32 hold row n).  D's row m = 8q + 4h + c lands in lane half h, register
4q + c, so the sources are dealt sigma(m) = piece h, tile offset 4q + c: lane half h walks piece h of the wave's sources in
ascending order, 16 sources per tile, and the two half sums are joined afterwards — exactly two pieces of the engine's wave
split (nbody_kernels.hpp, ForceArgs::wsplit), so the bits are those of the scalar-delivery kernels.
Per tile (1024 pairs): one half-wave global_load_dwordx4 (lanes 0-31: the 32 sources; lanes 32-63 keep 1.0), 3 MFMA into
the OTHER register set, 16 x 9 VALU on this one.  Two sets of 48 difference registers: the MFMAs of tile k+1 run beside
the VALU work on tile k (an MFMA's result may be read 19 issue slots after it at the earliest and nothing interlocks that:
here the distance is a tile).  128 VGPRs, 4 waves per SIMD.

Hardware rules kept (tools/gen_force_loop.py measured them): every 8-byte instruction of the loop starts at 4 mod 8 bytes
(4-byte scalar instructions come in pairs); no VALU instruction reads three VGPRs of the same parity (difference registers
of pair r have r's parity — the three bases are even — so d2/inv3 of pair r live in a register of the other parity); one
independent instruction between v_rsq_f32 and its consumer (the previous pair's accumulates; for a tile's first pair the
first fma of the second).
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "mini_nbody_amd", "csrc", "force_loop_mfma_gfx950.inc")

SOFT_BITS = 0x3089705F          # S/dzsoft.vhd:177
T = {0: 21, 1: 20}              # d2 / inv / inv3 of pair r: parity opposite to r's
U = 22                          # inv^2
A = [(24, 25, 26), (28, 29, 30)]            # A operands x, y, z of the two tiles in flight (dwordx4 loads: +27, +31 take w)
D = [(32, 48, 64), (80, 96, 112)]           # difference registers x, y, z (16 each) of the two sets
PTR, CNT, LO, HI = 36, 38, 40, 42           # s[36:37] tile pointer, s38 tile pairs left, s[40:41] lanes 0-31, s[42:43] lanes 32-63
TILE_BYTES = 256                            # 16 sources x 16 B per lane half


def pair_head(r, ds):
    """d2 chain and 1/sqrt of pair r"""
    dx, dy, dz = (b + r for b in ds)
    t = T[r & 1]
    return ["v_fmaak_f32 v%d, v%d, v%d, 0x%08x" % (t, dz, dz, SOFT_BITS),      # S/dzsoft.vhd:201-202
            "v_fma_f32 v%d, v%d, v%d, v%d" % (t, dy, dy, t),                   # S/dxy.vhd:113-122, S/dxyz_soft.vhd:149-150 (fma-contracted)
            "v_fma_f32 v%d, v%d, v%d, v%d" % (t, dx, dx, t),
            "v_rsq_f32_e64 v%d, v%d" % (t, t)]                                 # S/fxyz.vhd:101-102


def pair_cube(r):
    t = T[r & 1]
    return ["v_mul_f32_e64 v%d, v%d, v%d" % (U, t, t), "v_mul_f32_e64 v%d, v%d, v%d" % (t, t, U)]   # S/cube.vhd:66-70


def pair_acc(r, ds):
    dx, dy, dz = (b + r for b in ds)
    t = T[r & 1]
    return ["v_fma_f32 %%[ax], v%d, v%d, %%[ax]" % (dx, t), "v_fma_f32 %%[ay], v%d, v%d, %%[ay]" % (dy, t),   # S/fxyz.vhd:120-127
            "v_fma_f32 %%[az], v%d, v%d, %%[az]" % (dz, t)]


TIMING_BF16 = False     # TIMING-ONLY form (wrong results): the three MFMAs as v_mfma_f32_32x32x16_bf16 on whatever the registers hold —
                        # what would the loop cost if the differences came from the bf16 pipe, which does run beside the VALU?


def mfma(ds, a, k):
    if TIMING_BF16:
        return "v_mfma_f32_32x32x16_bf16 v[%d:%d], v[%d:%d], v[%d:%d], 0" % (ds[k], ds[k] + 15, a[0], a[0] + 3, a[0], a[0] + 3)
    return "v_mfma_f32_32x32x2_f32 v[%d:%d], v%d, %%[b%s], 0" % (ds[k], ds[k] + 15, a[k], "xyz"[k])


def load(a, off):
    return ["s_mov_b64 exec, s[%d:%d]" % (LO, LO + 1),
            "global_load_dwordx4 v[%d:%d], %%[voff], s[%d:%d] offset:%d" % (a[0], a[0] + 3, PTR, PTR + 1, off),
            "s_mov_b64 exec, -1"]


def phase(vs, ms, a, load_off, grouped=False):
    """VALU work on difference set vs (one tile); the MFMAs of the next tile go into set ms from operand set a, which is then
    refilled (two tiles ahead)."""
    ins = []
    if grouped:
        ins += [mfma(ms, a, 0), mfma(ms, a, 1), mfma(ms, a, 2)]
    # pair 0 has no predecessor in this tile: the wait state after its v_rsq_f32 is the first instruction of pair 1
    h0, h1 = pair_head(0, vs), pair_head(1, vs)
    ins += h0 + [h1[0]] + pair_cube(0)
    if not grouped:
        ins.append(mfma(ms, a, 0))
    ins += h1[1:] + pair_acc(0, vs) + pair_cube(1)
    for r in range(2, 16):
        ins += pair_head(r, vs) + pair_acc(r - 1, vs) + pair_cube(r)
        if not grouped and r == 3:
            ins.append(mfma(ms, a, 1))
        if not grouped and r == 6:
            ins.append(mfma(ms, a, 2))
        if r == 7:
            ins += load(a, load_off)
    ins += pair_acc(15, vs)
    return ins


def build(pad, grouped=False):
    ins = [
        "s_mov_b64 s[%d:%d], %%[p]" % (PTR, PTR + 1),
        "s_mov_b32 s%d, %%[iters]" % CNT,
        "s_mov_b64 s[%d:%d], 0xffffffff" % (LO, LO + 1),
        "s_not_b64 s[%d:%d], s[%d:%d]" % (HI, HI + 1, LO, LO + 1),
        "s_mov_b64 exec, s[%d:%d]" % (HI, HI + 1),
    ]
    ins += ["v_mov_b32 v%d, 1.0" % r for a in A for r in a]           # A[m][1] = 1: never overwritten (the loads run on lanes 0-31)
    ins += ["s_mov_b64 exec, s[%d:%d]" % (LO, LO + 1),
            "global_load_dwordx4 v[%d:%d], %%[voff], s[%d:%d]" % (A[0][0], A[0][0] + 3, PTR, PTR + 1),
            "global_load_dwordx4 v[%d:%d], %%[voff], s[%d:%d] offset:%d" % (A[1][0], A[1][0] + 3, PTR, PTR + 1, TILE_BYTES),
            "s_mov_b64 exec, -1",
            "s_waitcnt vmcnt(1)",
            mfma(D[0], A[0], 0), mfma(D[0], A[0], 1), mfma(D[0], A[0], 2)]
    ins += load(A[0], 2 * TILE_BYTES)
    ins += ["s_add_u32 s%d, s%d, 0x%x" % (PTR, PTR, 3 * TILE_BYTES), "s_addc_u32 s%d, s%d, 0" % (PTR + 1, PTR + 1),
            "s_nop 15", "s_nop 15"]                                   # >= 19 issue slots between an MFMA and the first read of its result
    ins.append(".p2align 6")
    ins += ["s_nop 0"] * pad
    ins.append("1:")
    ins += ["s_waitcnt vmcnt(1)", "s_sub_u32 s%d, s%d, 1" % (CNT, CNT)]
    ins += phase(D[0], D[1], A[1], 0, grouped)
    ins += ["s_waitcnt vmcnt(1)", "s_nop 0"]
    ins += phase(D[1], D[0], A[0], TILE_BYTES, grouped)
    ins += ["s_add_u32 s%d, s%d, 0x%x" % (PTR, PTR, 2 * TILE_BYTES), "s_addc_u32 s%d, s%d, 0" % (PTR + 1, PTR + 1),
            "s_cmp_lg_u32 s%d, 0" % CNT, "s_cbranch_scc1 1b"]
    ins += ["s_waitcnt vmcnt(0)", "s_nop 15", "s_nop 3"]              # retire the prefetches and the last (unused) MFMAs
    return ins


def check(ins):
    pre = ins[ins.index(".p2align 6") + 1:ins.index("1:")]
    assert all(i.startswith("s_nop") for i in pre)
    head = 4 * len(pre)
    loop = ins[ins.index("1:") + 1:]
    loop = loop[:loop.index("s_cbranch_scc1 1b") + 1]
    nbytes, phase_of_valu = 0, set()
    for i in loop:
        op = i.split()[0]
        size = 4 if (op.startswith("s_") and not op.startswith("s_load")) else 8
        if op.startswith("v_") and not op.startswith("v_mfma"):
            regs = [int(x) for x in re.findall(r"\bv(\d+)\b", i)]
            if "%[a" in i:      # the accumulates: difference register and inv3 register differ in parity whatever the accumulator's is
                assert len(regs) == 2 and len({r & 1 for r in regs}) == 2, i
            elif len(regs[1:]) == 3:
                assert len({r & 1 for r in regs[1:]}) == 2, i
        if op.startswith("v_"):
            phase_of_valu.add((head + nbytes) % 8)
        nbytes += size
    assert len(phase_of_valu) == 1, phase_of_valu
    return head % 64, phase_of_valu.pop()


def main():
    clob = ["v%d" % r for r in range(20, 32)] + ["v%d" % r for r in range(32, 128)] + \
           ["s%d" % r for r in (PTR, PTR + 1, CNT, LO, LO + 1, HI, HI + 1)] + ["scc", "memory"]
    with open(OUT, "w") as f:
        f.write("// GENERATED by tools/gen_mfma_loop.py — do not edit.  See that file for the why.\n")
        global TIMING_BF16
        for v, pad, grouped in ((0, 14, False), (1, 15, False), (2, 15, True), (3, 15, False), (4, 15, True)):
            TIMING_BF16 = v >= 3
            ins = build(pad, grouped)
            head, ph = check(ins)
            f.write("// V%d: loop head %d bytes past a 64-byte line, vector instructions at %d mod 8%s%s\n" % (v, head, ph, ", MFMAs grouped at the head of a tile" if grouped else "",
                    "; TIMING ONLY, WRONG RESULTS: bf16 MFMAs on arbitrary operands" if TIMING_BF16 else ""))
            f.write("#define NB_MFMA_LOOP_V%d \"%s\"\n" % (v, "\\n\\t".join(ins)))
        f.write("#define NB_MFMA_LOOP_CLOBBERS %s\n" % ", ".join('"%s"' % c for c in clob))
        f.write("#define NB_MFMA_LOOP_TILE %d\n" % 16)
    print("wrote %s" % OUT)


if __name__ == "__main__":
    main()
