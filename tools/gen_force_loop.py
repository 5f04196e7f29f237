#!/usr/bin/env python3
"""Generate mini_nbody_amd/csrc/force_loop_gfx950.inc — the hand-scheduled inner loop of the fp32 force kernel.

Why the hot loop is written in ISA (measurements: profiles/r01_microbench_streams.txt via tools/gen_streams.py,
profiles/r01_sweep_isa.txt):
  * every instruction of the pair interaction issues at 2 cycles per wave64 EXCEPT that the cost of the
    quarter-rate v_rsq_f32 depends on code placement: with uniform 64-bit encodings the stream costs 2.6
    cycles/instruction (31.5 per pair) when instructions start at 4 mod 8 bytes and 3.9 (47 per pair) at 0 mod 8;
    hipcc's mix of 32- and 64-bit encodings drifts between the two (35 per pair in the real kernel).  Here every
    instruction of the loop is 8 bytes and 4-byte scalar instructions come in pairs, so the phase never flips.
  * an instruction whose three VGPR source reads have the same parity (register number mod 2) costs 4 cycles instead
    of 2: temporaries are pinned to physical registers, t/u even, dx/dy/dz odd, so no instruction does that.
  * gfx940+ needs one wait state between a transcendental and a VALU instruction that reads its result, and hipcc
    does not pad inline asm: per body the order is
        sub sub sub | fma fma fma | rsq | 3 accumulating fma of the PREVIOUS body | mul mul
    — the previous body's accumulates are the wait state, and this was the fastest legal single-chain order in the
    stream harness (ord_defA, 2.63 cycles/instruction).
  * live-in / live-out values are copied into fixed registers (v8-v18, s33) so that the loop's bytes, banks and
    placement do not depend on hipcc's register allocation around the asm statement; the loop head sits 60 bytes
    past a 64-byte line.
  * a single fp32 accumulator per axis is off by 1e-4 of the force after 2^20 terms; the reference keeps 16 partial
    sums and an adder tree for the same reason (S/fxyz.vhd:129-145, S/final_adder.vhd:88-104).  Here the sum has two
    levels: every BLK groups (sum_block / 8, default 128) the three accumulators are added to second-level ones and
    restart from zero — 3 v_add + 7 v_mov + 4 scalar instructions per 12288 VALU instructions.

Loop shape (one wave, one body i per lane, sources delivered as wave-uniform scalar loads):
    A = s[36:51], B = s[52:67]: two buffers of 4 bodies {x,y,z,w}; pointer s[34:35], group counter s68, stride s69,
    groups left s70, groups per block s71, full-block flag s33 in the product loop (eps is a literal there; s72 in the forms
    that keep eps in s33): 78-79 SGPRs with VCC etc., 8 waves per SIMD fit
    prologue: load A
    block: s68 = min(s70, s71) groups
      loop:  wait A | load B | 4 bodies from A (48 VALU) | advance pointer | wait B | load A (next group) | 4 bodies from B
      accumulate the block's last body; a full block is folded into level 2, a short one (the segment's last) is left
      in level 1 for the caller
    (the load of the group after the last one reads <= 64 bytes past the segment: inside the 1 KiB pad of the arrays)
The arithmetic and its order per body are exactly pair_f32<0> of nbody_kernels.hpp (sources ascending), so the result
is bit-identical to the C++ kernels (tests/test_gpu_parity.py).

Forms emitted (NBODY_OPT_ISA_PHASE selects one; include/nbody.h lists them): NB_FORCE_LOOP_V1 = the product loop (head 60
bytes past a 64-byte line, eps as the literal of v_fmaak_f32); V0 = the same instructions one 4-byte phase off (3.4k vs 4.6k G/s); V2 staggered
s_load_dwordx8; V9-V13, V16, V17 other encodings of the SGPR-reading instructions (all bit-identical, all slower); V3-V8, V14, V15
TIMING-ONLY diagnostic forms with wrong results that price one part of the loop inside the real kernel
(profiles/r02_loop_diagnostics.md); NB_FORCE_LOOP_LONG = 8-body buffers for launches with few waves per SIMD.
Bring-up experiments that did not help (16-source groups, base+offset addressing, fused count-down, loads issued
mid-buffer, other in-body orders) are recorded in DESIGN.md §3.1 and in the git history of this file.
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "mini_nbody_amd", "csrc", "force_loop_gfx950.inc")

U = 22                                  # inv^2 (even)
T2 = [20, 24]                           # d2 / inv / inv3 (even), alternating with the body index
DSETS = [(21, 23, 25), (27, 29, 31)]    # dx dy dz (odd), alternating with the body index
XI, YI, ZI, AX, AY, AZ, EPS = "v8", "v9", "v10", "v12", "v13", "v14", "s33"
A_BASE, B_BASE = 36, 52            # s_load_dwordx16 destinations: multiples of 4
PTR, CNT, STRIDE = 34, 68, 69      # highest SGPR of the product loop: s71 (s72 in the diagnostic forms) -> 8 waves per SIMD fit
                                   # (81-96 SGPRs: 7, MI355X_MICROARCH.md "Occupancy API" row)
GROUP = 8


SOFT_BITS = 0x3089705F   # S/dzsoft.vhd:177 (the engine's only softening: nbody_kernels.hpp kSoftBits)
EPSV = "v18"    # eps in a VGPR (even: dz, dz, eps are then not three same-parity reads)


def body(k, sbase, b, style="fmaak"):
    """body k of the group reads source b of the SGPR buffer at sbase; accumulates body k-1.
    style: how the instructions that read an SGPR are encoded.  Measured inside the real kernel
    (profiles/r02_loop_diagnostics.md): a VOP3 (64-bit) instruction with an SGPR source costs ~0.9 cycles more than the
    same instruction reading VGPRs; the 32-bit VOP2 encoding of v_sub_f32 with the SGPR in src0 costs ~0.2 more.
      fmaak   source coordinates from SGPRs (VOP3), eps as the 32-bit literal of v_fmaak_f32 (VOP2 + literal = 8 bytes,
              the placement holds): THE PRODUCT LOOP — no SGPR and no third VGPR read for eps (+0.3 % over vgpreps)
      vgpreps every instruction VOP3, source coordinates from SGPRs, eps from a VGPR (+0.4..1.0 % over e64; NBODY_OPT_ISA_PHASE 18)
      e64     every instruction VOP3, source coordinates and eps from SGPRs (round 1's loop; NBODY_OPT_ISA_PHASE 12)
      e32sub  the three subtractions VOP2 + one s_nop (keeps every 8-byte instruction at 4 mod 8), eps from a VGPR
      e32sub_nofill  the same without the s_nop (the phase alternates from body to body)
      e32sub_seps    e32sub with eps still in an SGPR
      e32all  every instruction that has a 32-bit encoding uses it (phase not controlled)"""
    t, tp = T2[k & 1], T2[(k - 1) & 1]
    dx, dy, dz = DSETS[k & 1]
    px, py, pz = DSETS[(k - 1) & 1]
    s0 = sbase + 4 * b
    if style == "pksub":
        # experiment: dx and dy in ONE packed subtraction (v_pk_add_f32 with src1 negated: the same IEEE result), so the
        # SGPR-operand surcharge is paid twice per pair instead of three times; 11 instructions per pair instead of 12.
        # Needs (dx, dy) in an aligned register pair, so the parities differ from the product loop's: see build_pk()
        assert dy == dx + 1 and dx % 2 == 0
        return ["v_pk_add_f32 v[%d:%d], s[%d:%d], v[8:9] neg_lo:[0,1] neg_hi:[0,1]" % (dx, dy, s0, s0 + 1),
                "v_sub_f32_e64 v%d, s%d, %s" % (dz, s0 + 2, ZI),
                "v_fma_f32 v%d, v%d, v%d, %s" % (t, dz, dz, EPSV),
                "v_fma_f32 v%d, v%d, v%d, v%d" % (PK_T, dy, dy, t),
                "v_fma_f32 v%d, v%d, v%d, v%d" % (t, dx, dx, PK_T),
                "v_rsq_f32_e64 v%d, v%d" % (t, t),
                "v_fma_f32 %s, v%d, v%d, %s" % (AX, px, tp, AX), "v_fma_f32 %s, v%d, v%d, %s" % (AY, py, tp, AY),
                "v_fma_f32 %s, v%d, v%d, %s" % (AZ, pz, tp, AZ),
                "v_mul_f32_e64 v%d, v%d, v%d" % (U, t, t), "v_mul_f32_e64 v%d, v%d, v%d" % (t, t, U)]
    if style in ("sdwasub", "e32sub2"):
        # round 4 experiments on the SGPR-operand surcharge (a VOP3 instruction reading an SGPR costs ~0.9 cycles more on a shared SIMD,
        # the VOP2 encoding ~0.2, profiles/r02_loop_diagnostics.md), both with the product loop's placement (8-byte units) and eps literal:
        #   sdwasub   the three subtractions in the SDWA encoding (VOP2 + a selector dword = 8 bytes; gfx9 SDWA takes an SGPR in src0)
        #   e32sub2   dx and dy as two 4-byte VOP2 (one 8-byte unit), dz as VOP3
        if style == "sdwasub":
            sel = "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD"
            out = ["v_sub_f32_sdwa v%d, s%d, %s %s" % (dx, s0, XI, sel), "v_sub_f32_sdwa v%d, s%d, %s %s" % (dy, s0 + 1, YI, sel),
                   "v_sub_f32_sdwa v%d, s%d, %s %s" % (dz, s0 + 2, ZI, sel)]
        else:
            out = ["v_sub_f32_e32 v%d, s%d, %s" % (dx, s0, XI), "v_sub_f32_e32 v%d, s%d, %s" % (dy, s0 + 1, YI), "v_sub_f32_e64 v%d, s%d, %s" % (dz, s0 + 2, ZI)]
        out.append("v_fmaak_f32 v%d, v%d, v%d, 0x%08x" % (t, dz, dz, SOFT_BITS))
        out += ["v_fma_f32 v%d, v%d, v%d, v%d" % (t, dy, dy, t), "v_fma_f32 v%d, v%d, v%d, v%d" % (t, dx, dx, t),
                "v_rsq_f32_e64 v%d, v%d" % (t, t),
                "v_fma_f32 %s, v%d, v%d, %s" % (AX, px, tp, AX), "v_fma_f32 %s, v%d, v%d, %s" % (AY, py, tp, AY),
                "v_fma_f32 %s, v%d, v%d, %s" % (AZ, pz, tp, AZ),
                "v_mul_f32_e64 v%d, v%d, v%d" % (U, t, t), "v_mul_f32_e64 v%d, v%d, v%d" % (t, t, U)]
        return out
    if style != "e64":
        sub = "v_sub_f32_e64" if style in ("vgpreps", "subrev", "fmaak") else "v_sub_f32_e32"
        eps = EPS if style == "e32sub_seps" else EPSV
        out = ["%s v%d, s%d, %s" % (sub, dx, s0, XI), "%s v%d, s%d, %s" % (sub, dy, s0 + 1, YI), "%s v%d, s%d, %s" % (sub, dz, s0 + 2, ZI)]
        if style == "subrev":   # the SGPR in src1 instead of src0: D = S1 - S0 = target - this, the same value
            out = ["v_subrev_f32_e64 v%d, %s, s%d" % (dx, XI, s0), "v_subrev_f32_e64 v%d, %s, s%d" % (dy, YI, s0 + 1), "v_subrev_f32_e64 v%d, %s, s%d" % (dz, ZI, s0 + 2)]
        if style in ("e32sub", "e32sub_seps"):
            out.append("s_nop 0")
        if style == "fmaak":    # eps as the 32-bit literal of a VOP2 v_fmaak_f32 (8 bytes too): one VGPR read fewer per pair (S/dzsoft.vhd:201-202)
            out.append("v_fmaak_f32 v%d, v%d, v%d, 0x%08x" % (t, dz, dz, SOFT_BITS))
        else:
            out.append("v_fma_f32 v%d, v%d, v%d, %s" % (t, dz, dz, eps))
        if style == "e32all":
            out += ["v_fmac_f32_e32 v%d, v%d, v%d" % (t, dy, dy), "v_fmac_f32_e32 v%d, v%d, v%d" % (t, dx, dx),
                    "v_rsq_f32_e32 v%d, v%d" % (t, t),
                    "v_fmac_f32_e32 %s, v%d, v%d" % (AX, px, tp), "v_fmac_f32_e32 %s, v%d, v%d" % (AY, py, tp),
                    "v_fmac_f32_e32 %s, v%d, v%d" % (AZ, pz, tp),
                    "v_mul_f32_e32 v%d, v%d, v%d" % (U, t, t), "v_mul_f32_e32 v%d, v%d, v%d" % (t, t, U)]
        else:
            out += ["v_fma_f32 v%d, v%d, v%d, v%d" % (t, dy, dy, t), "v_fma_f32 v%d, v%d, v%d, v%d" % (t, dx, dx, t),
                    "v_rsq_f32_e64 v%d, v%d" % (t, t),
                    "v_fma_f32 %s, v%d, v%d, %s" % (AX, px, tp, AX), "v_fma_f32 %s, v%d, v%d, %s" % (AY, py, tp, AY),
                    "v_fma_f32 %s, v%d, v%d, %s" % (AZ, pz, tp, AZ),
                    "v_mul_f32_e64 v%d, v%d, v%d" % (U, t, t), "v_mul_f32_e64 v%d, v%d, v%d" % (t, t, U)]
        return out
    return [
        "v_sub_f32_e64 v%d, s%d, %s" % (dx, s0, XI),               # S/dxy.vhd:94-95   target - this
        "v_sub_f32_e64 v%d, s%d, %s" % (dy, s0 + 1, YI),           # S/dxy.vhd:97-98
        "v_sub_f32_e64 v%d, s%d, %s" % (dz, s0 + 2, ZI),           # S/dzsoft.vhd:186-187
        "v_fma_f32 v%d, v%d, v%d, %s" % (t, dz, dz, EPS),          # S/dzsoft.vhd:201-202
        "v_fma_f32 v%d, v%d, v%d, v%d" % (t, dy, dy, t),           # d2 (fma-contracted)
        "v_fma_f32 v%d, v%d, v%d, v%d" % (t, dx, dx, t),
        "v_rsq_f32_e64 v%d, v%d" % (t, t),                         # S/fxyz.vhd:101-102
        "v_fma_f32 %s, v%d, v%d, %s" % (AX, px, tp, AX),           # S/fxyz.vhd:120-127 (previous body)
        "v_fma_f32 %s, v%d, v%d, %s" % (AY, py, tp, AY),
        "v_fma_f32 %s, v%d, v%d, %s" % (AZ, pz, tp, AZ),
        "v_mul_f32_e64 v%d, v%d, v%d" % (U, t, t),                 # S/cube.vhd:66-67
        "v_mul_f32_e64 v%d, v%d, v%d" % (t, t, U),                 # S/cube.vhd:69-70
    ]


BX, BY, BZ = "v15", "v16", "v17"        # level-2 accumulators (finished blocks)
TOT, BLK, FULL = 70, 71, 72             # groups still to do after this block, groups per block, "this block is a full one"
HEAD_BYTES = 16                         # the four 4-byte scalar instructions between the label `2:` and the loop head

# register maps: the product loop (two 4-body buffers) and the long-buffer loop (two 8-body buffers)
SHORT = dict(bodies=4, a=A_BASE, b=B_BASE, ptr=PTR, cnt=CNT, stride=STRIDE, tot=TOT, blk=BLK, full=FULL)
LONG = dict(bodies=8, a=36, b=68, ptr=28, cnt=35, stride=34, tot=31, blk=30, full=27)   # s32, s100, s101 are reserved by hipcc
GROUP_LONG = 16


def loads(m, base, first_byte):
    """s_load_dwordx16 instructions filling the buffer at SGPR `base` from PTR + first_byte"""
    return ["s_load_dwordx16 s[%d:%d], s[%d:%d], 0x%x" % (base + 16 * k, base + 16 * k + 15, m["ptr"], m["ptr"] + 1, first_byte + 64 * k)
            for k in range(m["bodies"] // 4)]


def half_loads(m, base, first_byte):
    """the same 64 bytes as one s_load_dwordx16, as two s_load_dwordx8 (experiment: staggered delivery)"""
    return ["s_load_dwordx8 s[%d:%d], s[%d:%d], 0x%x" % (base + 8 * k, base + 8 * k + 7, m["ptr"], m["ptr"] + 1, first_byte + 32 * k) for k in range(2)]


def diagnostic(ins, no_rsq, no_loads):
    """TIMING-ONLY forms of a loop (wrong results): v_rsq_f32 -> v_mov_b32 of the same registers (same encoding size,
    same dependency chain, no transcendental), and/or the loop's s_load_dwordx16 -> two s_nop (same bytes, no scalar
    memory traffic; the prologue's load stays so the SGPRs hold numbers).  They price the transcendental and the
    scalar side inside the real kernel (profiles/r02_cycles_per_wave_pair.md)."""
    out = []
    in_loop = False
    for i in ins:
        if i == "2:":
            in_loop = True
        if no_rsq and i.startswith("v_rsq_f32_e64"):
            i = i.replace("v_rsq_f32_e64", "v_mov_b32_e64")
        if no_loads and in_loop and i.startswith("s_load_dwordx16"):
            out += ["s_nop 0", "s_nop 0"]
            continue
        out.append(i)
    return out


def diag_body(kind, k, sbase, b):
    """TIMING-ONLY stand-ins for body() (12 full-rate instructions, no transcendental): what limits full-rate issue?
       indep    12 v_fma_f32 with no dependency between them (distinct destinations)
       e32      the pair interaction in 32-bit encodings wherever one exists (v_mov instead of v_rsq)
       vgprsrc  the pair interaction with the source coordinates read from VGPRs instead of SGPRs"""
    t, tp = T2[k & 1], T2[(k - 1) & 1]
    dx, dy, dz = DSETS[k & 1]
    px, py, pz = DSETS[(k - 1) & 1]
    s0 = sbase + 4 * b
    if kind == "indep":
        dst = [20, 21, 22, 23, 24, 25, 27, 29, 31, 12, 13, 14]
        return ["v_fma_f32 v%d, v8, v9, v%d" % (d, d) for d in dst]
    if kind == "e32":
        return [
            "v_sub_f32_e32 v%d, s%d, %s" % (dx, s0, XI), "v_sub_f32_e32 v%d, s%d, %s" % (dy, s0 + 1, YI),
            "v_sub_f32_e32 v%d, s%d, %s" % (dz, s0 + 2, ZI),
            "v_fma_f32 v%d, v%d, v%d, %s" % (t, dz, dz, EPS),
            "v_fmac_f32_e32 v%d, v%d, v%d" % (t, dy, dy), "v_fmac_f32_e32 v%d, v%d, v%d" % (t, dx, dx),
            "v_mov_b32_e32 v%d, v%d" % (t, t),
            "v_fmac_f32_e32 %s, v%d, v%d" % (AX, px, tp), "v_fmac_f32_e32 %s, v%d, v%d" % (AY, py, tp), "v_fmac_f32_e32 %s, v%d, v%d" % (AZ, pz, tp),
            "v_mul_f32_e32 v%d, v%d, v%d" % (U, t, t), "v_mul_f32_e32 v%d, v%d, v%d" % (t, t, U),
        ]
    if kind in ("vgprsrc", "vgprsrc_rsq", "vgprsrc_rsq_lds"):
        out = body(k, sbase, b, "vgpreps")
        out[0] = "v_sub_f32_e64 v%d, v16, %s" % (dx, XI)     # v15..v17 hold the level-2 sums: any numbers will do here
        out[1] = "v_sub_f32_e64 v%d, v15, %s" % (dy, YI)
        out[2] = "v_sub_f32_e64 v%d, v16, %s" % (dz, ZI)
        if kind == "vgprsrc":
            out[6] = out[6].replace("v_rsq_f32_e64", "v_mov_b32_e64")
        if kind == "vgprsrc_rsq_lds":      # + one broadcast LDS read per source into registers nothing reads
            out.insert(3, "ds_read_b128 v[40:43], v44")
        return out
    raise ValueError(kind)


def build(pad, m=SHORT, stagger=False, diag=None, style="fmaak"):
    """pad: s_nop count after .p2align 6, so that the inner loop's head `1:` sits 4*pad + HEAD_BYTES bytes past a
    64-byte line (60 for the product loop: every VALU instruction of the loop then starts at 4 mod 8 bytes).
    m: register map.  SHORT = the product loop: buffers of 4 bodies, a buffer's load is in flight for the 48 VALU
    instructions of the other one — hidden when >= 4 waves share the SIMD.  LONG = buffers of 8 bodies (two loads each):
    scalar loads return out of order, so a wave can only wait for ALL of them (lgkmcnt(0)) and the prefetch distance is one
    buffer; with 1-2 waves per SIMD (small N) a 4-body buffer is computed in ~100 ns against ~280 ns of load latency
    (measured: 71 ns per source), an 8-body buffer in ~200 ns per wave."""
    px, py, pz = DSETS[1]
    nb = m["bodies"]
    buf_bytes = 16 * nb
    if style == "fmaak" and not diag and m is SHORT:
        # the product loop reads eps as a literal, so EPS's SGPR (s33) is free inside the loop: the "full block" flag lives there
        # and the loop's highest scalar is s71 (with a workgroup of 16 waves hipcc is told to fit 8 waves per SIMD and then
        # treats s72 as reserved)
        m = dict(m, full=int(EPS[1:]))
    ins = [
        "v_mov_b32 %s, %%[xi]" % XI, "v_mov_b32 %s, %%[yi]" % YI, "v_mov_b32 %s, %%[zi]" % ZI,
        "v_mov_b32 %s, %%[ax]" % AX, "v_mov_b32 %s, %%[ay]" % AY, "v_mov_b32 %s, %%[az]" % AZ,
        "v_mov_b32 %s, %%[bx]" % BX, "v_mov_b32 %s, %%[by]" % BY, "v_mov_b32 %s, %%[bz]" % BZ,
        "s_mov_b32 %s, %%[eps]" % EPS,
        "v_mov_b32 %s, %%[eps]" % EPSV,
        "v_mov_b32 v44, 0",
        "s_mov_b64 s[%d:%d], %%[p]" % (m["ptr"], m["ptr"] + 1),
        "s_mov_b32 s%d, %%[groups]" % m["tot"],
        "s_mov_b32 s%d, %%[blk]" % m["blk"],
        "s_movk_i32 s%d, 0x%x" % (m["stride"], 2 * buf_bytes),
    ] + loads(m, m["a"], 0)
    # "previous body" of the very first body: d = 0 and inv3 = 0, so fma(0, 0, acc) leaves acc as it is
    ins += ["v_mov_b32 v%d, 0" % r for r in (px, py, pz, T2[1])]
    ins.append(".p2align 6")
    ins += ["s_nop 0"] * pad
    # ---- one block of the two-level sum: CNT = min(groups left, groups per block) iterations of the inner loop
    ins.append("2:")
    ins += ["s_min_u32 s%d, s%d, s%d" % (m["cnt"], m["tot"], m["blk"]),
            "s_cmp_eq_u32 s%d, s%d" % (m["cnt"], m["blk"]),
            "s_cselect_b32 s%d, 1, 0" % m["full"],
            "s_sub_u32 s%d, s%d, s%d" % (m["tot"], m["tot"], m["cnt"])]
    ins.append("1:")
    ins += ["s_waitcnt lgkmcnt(0)", "s_sub_u32 s%d, s%d, 1" % (m["cnt"], m["cnt"])]
    if stagger:
        # experiment: the other buffer arrives as two s_load_dwordx8, the second one issued two
        # bodies later, so that the returning data is written to the SGPR file in two bursts instead of one
        lb = half_loads(m, m["b"], buf_bytes)
        ins.append(lb[0])
        for b in range(nb):
            ins += body(b, m["a"], b)
            if b == 1:
                ins.append(lb[1])
    elif diag:
        ins += ["s_nop 0", "s_nop 0"] * (nb // 4)
        for b in range(nb):
            ins += diag_body(diag, b, m["a"], b)
    else:
        ins += loads(m, m["b"], buf_bytes)
        for b in range(nb):
            ins += body(b, m["a"], b, style)
    ins += ["s_add_u32 s%d, s%d, s%d" % (m["ptr"], m["ptr"], m["stride"]), "s_addc_u32 s%d, s%d, 0" % (m["ptr"] + 1, m["ptr"] + 1)]
    ins += ["s_waitcnt lgkmcnt(0)", "s_nop 0"]
    if stagger:
        la = half_loads(m, m["a"], 0)
        ins.append(la[0])
        for b in range(nb):
            ins += body(nb + b, m["b"], b)
            if b == 1:
                ins.append(la[1])
    elif diag:
        ins += ["s_nop 0", "s_nop 0"] * (nb // 4)
        for b in range(nb):
            ins += diag_body(diag, nb + b, m["b"], b)
    else:
        ins += loads(m, m["a"], 0)
        for b in range(nb):
            ins += body(nb + b, m["b"], b, style)
    ins += ["s_cmp_lg_u32 s%d, 0" % m["cnt"], "s_cbranch_scc1 1b"]
    # the block's last body has not been accumulated yet (an odd body index: d set 1, t register 1)
    ins += ["v_fma_f32 %s, v%d, v%d, %s" % (AX, px, T2[1], AX), "v_fma_f32 %s, v%d, v%d, %s" % (AY, py, T2[1], AY),
            "v_fma_f32 %s, v%d, v%d, %s" % (AZ, pz, T2[1], AZ)]
    # a block that stopped short of BLK groups is the segment's last one: its sum stays in level 1 (the caller adds
    # the leftover sources to it and folds it)
    ins += ["s_cmp_eq_u32 s%d, 0" % m["full"], "s_cbranch_scc1 3f"]
    # fold: level 2 += level 1 (ascending block order), level 1 = 0, and again no "previous body"
    ins += ["v_add_f32_e64 %s, %s, %s" % (BX, BX, AX), "v_add_f32_e64 %s, %s, %s" % (BY, BY, AY),
            "v_add_f32_e64 %s, %s, %s" % (BZ, BZ, AZ)]
    ins += ["v_mov_b32 %s, 0" % r for r in (AX, AY, AZ)]
    ins += ["v_mov_b32 v%d, 0" % r for r in (px, py, pz, T2[1])]
    ins += ["s_cmp_lg_u32 s%d, 0" % m["tot"], "s_cbranch_scc1 2b"]
    ins.append("3:")
    # retire the unused prefetch; sums out
    ins += ["s_waitcnt lgkmcnt(0)", "v_mov_b32 %%[ax], %s" % AX, "v_mov_b32 %%[ay], %s" % AY, "v_mov_b32 %%[az], %s" % AZ,
            "v_mov_b32 %%[bx], %s" % BX, "v_mov_b32 %%[by], %s" % BY, "v_mov_b32 %%[bz], %s" % BZ]
    return ins


PK_T = 25                                     # odd: d2 after the dy term
PK_DSETS = [(26, 27, 21), (28, 29, 23)]       # (dx, dy) an aligned pair, dz odd
PK_AX = "v11"                                 # odd: fma(AX, dx (even), inv3 (even), AX) must not read three even registers


def build_pk(pad):
    """the product loop with style "pksub" and the register parities that form needs"""
    global DSETS, AX
    keep = DSETS, AX
    DSETS, AX = PK_DSETS, PK_AX
    try:
        return build(pad, style="pksub")
    finally:
        DSETS, AX = keep


# ---- fp64 (BASELINE config 5's arithmetic): same structure, 16 instructions per pair (round 1: 19, rounds 2-3: 17), all VOP3 (8 bytes)
# once v_rsq_f64 is written in its 64-bit encoding; 4 sources per iteration (two buffers of 2 bodies x 32 bytes).
# the inverse cube = v_rsq_f64 seed + one third-order step on the cube, the 6-operation form of inv3_f64() in nbody_kernels.hpp (body_f64 below)
D_T, D_HX, D_R, D_E, D_U = 32, 38, 40, 42, 44
D_Y = [34, 36]
D_DSETS = [(20, 22, 24), (26, 28, 30)]
D_XI, D_YI, D_ZI, D_AX, D_AY, D_AZ, D_EPS = 8, 10, 12, 14, 16, 18, 34   # eps in s[34:35]
GROUP_F64 = 4


def vp(r):
    return "v[%d:%d]" % (r, r + 1)


def sp(r):
    return "s[%d:%d]" % (r, r + 1)


D_EPSV, D_KV, D_KV2 = 46, 48, 50      # v[48:49] = 3/2 always (an instruction reads at most ONE SGPR constant on gfx9: 15/8 takes that slot);
                                      # experiment (F64_V2): eps and 15/8 in VGPR pairs too instead of SGPR pairs


def body_f64(k, sbase, b, vconst=False):
    y, yp = D_Y[k & 1], D_Y[(k - 1) & 1]
    dx, dy, dz = D_DSETS[k & 1]
    px, py, pz = D_DSETS[(k - 1) & 1]
    s0 = sbase + 8 * b
    out = [
        "v_add_f64 %s, %s, -%s" % (vp(dx), sp(s0), vp(D_XI)),
        "v_add_f64 %s, %s, -%s" % (vp(dy), sp(s0 + 2), vp(D_YI)),
        "v_add_f64 %s, %s, -%s" % (vp(dz), sp(s0 + 4), vp(D_ZI)),
        "v_fma_f64 %s, %s, %s, %s" % (vp(D_T), vp(dz), vp(dz), vp(D_EPSV) if vconst else sp(D_EPS)),
        "v_fma_f64 %s, %s, %s, %s" % (vp(D_T), vp(dy), vp(dy), vp(D_T)),
        "v_fma_f64 %s, %s, %s, %s" % (vp(D_T), vp(dx), vp(dx), vp(D_T)),
        "v_rsq_f64_e64 %s, %s" % (vp(y), vp(D_T)),
        "v_fma_f64 %s, %s, %s, %s" % (vp(D_AX), vp(px), vp(yp), vp(D_AX)),
        "v_fma_f64 %s, %s, %s, %s" % (vp(D_AY), vp(py), vp(yp), vp(D_AY)),
        "v_fma_f64 %s, %s, %s, %s" % (vp(D_AZ), vp(pz), vp(yp), vp(D_AZ)),
    ]
    # inv3 = x^(-3/2) straight from the v_rsq_f64 seed y (about 2^-24 relative), one third-order step on the CUBE (inv3_f64 in
    # nbody_kernels.hpp, round 4): with e = 1 - x*y^2,  x^(-3/2) = y^3 (1 - e)^(-3/2) = y^3 (1 + e (3/2 + 15/8 e) + (35/16) e^3 ...),
    # the e^3 term is < 2^-70: full binary64 in SIX operations — y2 = y*y, e = fma(-x, y2, 1), y3 = y2*y, p = fma(e, 15/8, 3/2), q = e*p,
    # inv3 = fma(y3, q, y3) — where refining y first and cubing it afterwards took seven (round 2/3: 17 instructions per pair, now 16)
    out += ["v_mul_f64 %s, %s, %s" % (vp(D_U), vp(y), vp(y)),
            "v_fma_f64 %s, -%s, %s, 1.0" % (vp(D_E), vp(D_T), vp(D_U)),
            "v_mul_f64 %s, %s, %s" % (vp(y), vp(D_U), vp(y)),
            "v_fma_f64 %s, %s, %s, %s" % (vp(D_HX), vp(D_E), vp(D_KV2) if vconst else sp(D_K1875), vp(D_KV)),
            "v_mul_f64 %s, %s, %s" % (vp(D_E), vp(D_E), vp(D_HX)),
            "v_fma_f64 %s, %s, %s, %s" % (vp(y), vp(y), vp(D_E), vp(y))]
    return out


F64_PTR, F64_CNT, F64_STRIDE = 68, 70, 71    # the fp64 loop keeps its own scalars (it is VGPR-limited to 5 waves anyway)
D_K1875 = 72                                 # s[72:73] = 15/8 (VOP3 takes no literal on gfx9: the constant lives in an SGPR pair)


def build_f64(pad, vconst=False):
    px, py, pz = D_DSETS[1]
    ins = [
        "v_mov_b64 %s, %%[xi]" % vp(D_XI), "v_mov_b64 %s, %%[yi]" % vp(D_YI), "v_mov_b64 %s, %%[zi]" % vp(D_ZI),
        "v_mov_b64 %s, %%[ax]" % vp(D_AX), "v_mov_b64 %s, %%[ay]" % vp(D_AY), "v_mov_b64 %s, %%[az]" % vp(D_AZ),
        "s_mov_b64 %s, %%[eps]" % sp(D_EPS),
        "s_mov_b64 s[%d:%d], %%[p]" % (F64_PTR, F64_PTR + 1),
        "s_mov_b32 s%d, %%[groups]" % F64_CNT,
        "s_movk_i32 s%d, 0x80" % F64_STRIDE,
        "s_mov_b32 s%d, 0" % D_K1875, "s_mov_b32 s%d, 0x3ffe0000" % (D_K1875 + 1),      # 1.875
        "v_mov_b32 v%d, 0" % D_KV, "v_mov_b32 v%d, 0x3ff80000" % (D_KV + 1),           # 1.5
        "s_load_dwordx16 s[%d:%d], s[%d:%d], 0x0" % (A_BASE, A_BASE + 15, F64_PTR, F64_PTR + 1),
    ]
    if vconst:
        ins += ["v_mov_b64 %s, %s" % (vp(D_EPSV), sp(D_EPS)), "v_mov_b64 %s, %s" % (vp(D_KV2), sp(D_K1875))]
    ins += ["v_mov_b64 %s, 0" % vp(r) for r in (px, py, pz, D_Y[1])]
    ins.append(".p2align 6")
    ins += ["s_nop 0"] * pad
    ins.append("1:")
    ins += ["s_waitcnt lgkmcnt(0)", "s_sub_u32 s%d, s%d, 1" % (F64_CNT, F64_CNT)]
    ins.append("s_load_dwordx16 s[%d:%d], s[%d:%d], 0x40" % (B_BASE, B_BASE + 15, F64_PTR, F64_PTR + 1))
    for b in range(2):
        ins += body_f64(b, A_BASE, b, vconst)
    ins += ["s_add_u32 s%d, s%d, s%d" % (F64_PTR, F64_PTR, F64_STRIDE), "s_addc_u32 s%d, s%d, 0" % (F64_PTR + 1, F64_PTR + 1)]
    ins += ["s_waitcnt lgkmcnt(0)", "s_nop 0"]
    ins.append("s_load_dwordx16 s[%d:%d], s[%d:%d], 0x0" % (A_BASE, A_BASE + 15, F64_PTR, F64_PTR + 1))
    for b in range(2):
        ins += body_f64(2 + b, B_BASE, b, vconst)
    ins += ["s_cmp_lg_u32 s%d, 0" % F64_CNT, "s_cbranch_scc1 1b"]
    ins += ["v_fma_f64 %s, %s, %s, %s" % (vp(D_AX), vp(px), vp(D_Y[1]), vp(D_AX)),
            "v_fma_f64 %s, %s, %s, %s" % (vp(D_AY), vp(py), vp(D_Y[1]), vp(D_AY)),
            "v_fma_f64 %s, %s, %s, %s" % (vp(D_AZ), vp(pz), vp(D_Y[1]), vp(D_AZ))]
    ins += ["s_waitcnt lgkmcnt(0)", "v_mov_b64 %%[ax], %s" % vp(D_AX), "v_mov_b64 %%[ay], %s" % vp(D_AY), "v_mov_b64 %%[az], %s" % vp(D_AZ)]
    return ins


def check(ins, strict=True):
    """the hardware rules the loop is built on (strict: every 8-byte instruction of the loop starts at 4 mod 8 bytes)"""
    # position of the loop head relative to the 64-byte line set by .p2align 6
    pre = ins[ins.index(".p2align 6") + 1:ins.index("1:")]
    head = sum(0 if i.endswith(":") else 4 for i in pre)      # only 4-byte scalar instructions there
    assert all(i.endswith(":") or i.startswith("s_") for i in pre)
    loop = ins[ins.index("1:"):]
    loop = loop[:loop.index("s_cbranch_scc1 1b") + 1]
    nbytes = 0
    first_valu = None
    for i in loop:
        if i.endswith(":"):
            continue
        op = i.split()[0]
        size = 4 if (op.startswith("s_") and not op.startswith("s_load")) or op.endswith("_e32") else 8
        if op.startswith("v_"):
            assert op.endswith("_e64") or op.endswith("_e32") or op in ("v_fma_f32", "v_pk_add_f32", "v_fmaak_f32", "v_sub_f32_sdwa"), i
            regs = [int(x) for x in re.findall(r"\bv(\d+)\b", i)]
            regs = regs if op.startswith("v_fmac") else regs[1:]                     # fmac reads its destination
            if len(regs) == 3:
                assert len({r & 1 for r in regs}) == 2, i                            # never three same-parity VGPR reads
        if size == 8:
            if strict:
                assert nbytes % 8 == 0, i                                            # 4-byte instructions only in pairs
            if op.startswith("v_") and first_valu is None:
                first_valu = (head + nbytes) % 8
        nbytes += size
    assert not strict or nbytes % 8 == 0
    return head % 64, first_valu


def main():
    regs = sorted(set([U] + T2 + [x for d in DSETS for x in d] + [int(r[1:]) for r in (XI, YI, ZI, AX, AY, AZ, BX, BY, BZ, EPSV)] + [40, 41, 42, 43, 44] + [PK_T, int(PK_AX[1:])] + [x for d in PK_DSETS for x in d]))
    clob = ["v%d" % r for r in regs] + [EPS] + ["s%d" % r for r in range(PTR, FULL)] + ["scc", "memory"]      # product loop: up to s71
    clob_diag = ["v%d" % r for r in regs] + [EPS] + ["s%d" % r for r in range(PTR, FULL + 1)] + ["scc", "memory"]
    with open(OUT, "w") as f:
        f.write("// GENERATED by tools/gen_force_loop.py — do not edit.  See that file for the why.\n")
        for v, pad in ((0, 14 - HEAD_BYTES // 4), (1, 15 - HEAD_BYTES // 4)):
            ins = build(pad)
            head, phase = check(ins)
            assert (head, phase) == ((56, 0) if v == 0 else (60, 4)), (head, phase)   # the placement round 1 measured
            f.write("#define NB_FORCE_LOOP_V%d \"%s\"\n" % (v, "\\n\\t".join(ins)))
        f.write("#define NB_FORCE_LOOP_CLOBBERS %s\n" % ", ".join('"%s"' % c for c in clob))
        f.write("#define NB_FORCE_LOOP_DIAG_CLOBBERS %s\n" % ", ".join('"%s"' % c for c in clob_diag))
        f.write("#define NB_FORCE_LOOP_GROUP %d\n" % GROUP)
        # Everything below up to NB_FORCE_LOOP_LONG exists only in the diagnostic build (`make diag`, -DNBODY_DIAG_LOOPS ->
        # libnbody_hip_diag.so): experiment encodings of the same operations (bit-identical, slower) and TIMING-ONLY forms
        # with wrong results.  The product library holds V0, V1 and LONG only and refuses the other NBODY_OPT_ISA_PHASE values.
        f.write("#ifdef NBODY_DIAG_LOOPS\n")
        ins = build(15 - HEAD_BYTES // 4, SHORT, stagger=True)
        assert check(ins) == (60, 4)
        f.write("#define NB_FORCE_LOOP_V2 \"%s\"\n" % "\\n\\t".join(ins))
        for v, (nr, nl) in ((3, (True, False)), (4, (False, True)), (5, (True, True))):
            ins = diagnostic(build(15 - HEAD_BYTES // 4), nr, nl)
            assert check(ins) == (60, 4)
            f.write("#define NB_FORCE_LOOP_V%d \"%s\"\n" % (v, "\\n\\t".join(ins)))
        for v, style, strict in ((9, "e32sub", True), (10, "e32sub_nofill", False), (11, "e32all", False), (12, "e64", True), (13, "e32sub_seps", True), (16, "subrev", True), (18, "vgpreps", True),
                                 (19, "sdwasub", True), (20, "e32sub2", True)):
            ins = build(15 - HEAD_BYTES // 4, style=style)
            assert check(ins, strict)[0] == 60 and (not strict or check(ins, strict)[1] == 4), (style, check(ins, strict))
            f.write("#define NB_FORCE_LOOP_V%d \"%s\"\n" % (v, "\\n\\t".join(ins)))
        ins = build_pk(15 - HEAD_BYTES // 4)
        assert check(ins) == (60, 4)
        f.write("#define NB_FORCE_LOOP_V17 \"%s\"\n" % "\\n\\t".join(ins))
        for v, kind in ((6, "indep"), (7, "e32"), (8, "vgprsrc"), (14, "vgprsrc_rsq"), (15, "vgprsrc_rsq_lds")):
            ins = build(15 - HEAD_BYTES // 4, diag=kind)
            f.write("#define NB_FORCE_LOOP_V%d \"%s\"\n" % (v, "\\n\\t".join(ins)))
        f.write("#endif  // NBODY_DIAG_LOOPS\n")
        ins = build(15 - HEAD_BYTES // 4, LONG)
        assert check(ins) == (60, 4)
        f.write("#define NB_FORCE_LOOP_LONG \"%s\"\n" % "\\n\\t".join(ins))
        sregs = sorted(set(list(range(LONG["a"], LONG["a"] + 64)) + [LONG["ptr"], LONG["ptr"] + 1] + [LONG[k] for k in ("cnt", "stride", "tot", "blk", "full")]))
        clob_long = ["v%d" % r for r in regs] + [EPS] + ["s%d" % r for r in sregs] + ["scc", "memory"]
        f.write("#define NB_FORCE_LOOP_LONG_CLOBBERS %s\n" % ", ".join('"%s"' % c for c in clob_long))
        f.write("#define NB_FORCE_LOOP_LONG_GROUP %d\n" % GROUP_LONG)
        # fp64 placement (measured with the 17-instruction body, profiles/r02_sweep_fp64_placement.txt): every pad that puts
        # the VALU instructions at 0 mod 8 bytes gives 1855-1863 G pairs/s at N = 262144, the two tried at 4 mod 8 1832-1840
        # — the opposite phase to the fp32 loop's.  V1 = the product loop, V0 = one 4-byte phase off.
        for v, pad in ((0, 15), (1, 14)):
            f.write("#define NB_FORCE_LOOP_F64_V%d \"%s\"\n" % (v, "\\n\\t".join(build_f64(pad))))
        f.write("#define NB_FORCE_LOOP_F64_V2 \"%s\"\n" % "\\n\\t".join(build_f64(14, vconst=True)))
        clob64 = ["v%d" % r for r in range(8, 52)] + ["s34", "s35"] + ["s%d" % r for r in range(A_BASE, D_K1875 + 2)] + ["scc", "memory"]
        f.write("#define NB_FORCE_LOOP_F64_CLOBBERS %s\n" % ", ".join('"%s"' % c for c in clob64))
        f.write("#define NB_FORCE_LOOP_F64_GROUP %d\n" % GROUP_F64)
    n_valu = len([i for i in build(11) if i.startswith("v_")])
    print("wrote %s (%d VALU instructions per iteration of %d bodies + prologue/drain)" % (OUT, n_valu, GROUP))


if __name__ == "__main__":
    main()
