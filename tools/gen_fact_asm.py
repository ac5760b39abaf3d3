#!/usr/bin/env python3
"""Generates csrc/rpgp_fact_asm_loop.inc: the hand-scheduled gfx950 inner loop of the factorised fused MVM
(`mvm_fact_asm_kernel` in csrc/rpgp_fact_asm.hip; JT = 20 projections, T = 1, two rows per lane).

One execution = one 64-column subtile = 64 steps; at step s lane l meets column (l + s) of the subtile (the LDS image holds
the 64 column records followed by a copy of the first 63, so the rotating index is a RUNNING POINTER + immediate offsets:
no per-step address arithmetic).  Per step and lane: 2 rows x 20 projections = 40 pair-terms, each
    t = a * 2b - b^2      (v_pk_fma_f32 over the lane's two rows, the column operands broadcast by op_sel)
    e = exp2(t)           (2 x v_exp_f32)
    K += e * Ea           (v_pk_fma_f32)
i.e. 20 + 40 + 20 vector instructions that are the floor of the exact-fp32 formulation, plus 4 per step for the two
products (accR += K v_col: one packed FMA; accT += K . v_row: two FMAs) and the DPP rotation of the transposed
accumulator, plus one pointer add per two steps.  Software pipeline (what the compiler's schedule lacked):
  * every `ds_read_b128` of step s+1 is issued right after the two FMAs that consumed the same quad of step s — a full
    step ahead of its use, one constant `s_waitcnt lgkmcnt(10)` per quad;
  * the K-FMAs of a quad are issued one quad late, interleaved with the next quad's exponentials: no v_exp_f32 result is
    read by the next instruction (no s_nop, no dependency bubble);
  * the finish of step s (last two K-FMAs, the three product instructions, the rotation) rides inside step s+1.

Register map (physical; the C++ side pins its operands to the same numbers):
  v[10:49]   A[j]  = {a_row0, a_row1}  of projection j          (input)
  v[50:89]   E[j]  = {Ea_row0, Ea_row1}                         (input)
  v[90:91]   accR  = {row0, row1} row products                  (in/out)
  v[92:93]   vrow  = {v_row0, v_row1}                           (input)
  v94        ptr   LDS byte address of this lane's record       (in/out: advanced by 64 records)
  v95, v96   accT  transposed accumulator, alternating per step (v95 = output)
  v[100:139] R[q]  column record quads {2b_e, -b_e^2, 2b_o, -b_o^2}, refilled in place
  v[140:141] vcol  the column's v, alternating per step
  v[142:149] TA0 TA1 TB0 TB1  t / e temporaries of even / odd quads
  v[150:153] K0 K1 kernel-value pairs of even / odd steps
"""
import os
import sys

REC_BYTES = 176            # 44 floats: 10 quads + v_col + 3 pad; (176 / 16) odd -> conflict-free per-lane ds_read_b128
VCOL_OFF = 160
NQ = 10


def A(j):
    return "v[%d:%d]" % (10 + 2 * j, 11 + 2 * j)


def E(j):
    return "v[%d:%d]" % (50 + 2 * j, 51 + 2 * j)


def Rq(q):
    return "v[%d:%d]" % (100 + 4 * q, 103 + 4 * q)


def Rpair(q, odd):
    b = 100 + 4 * q + (2 if odd else 0)
    return "v[%d:%d]" % (b, b + 1)


ACCR, VROW, PTR = "v[90:91]", "v[92:93]", "v94"
ACCT = ["v95", "v96"]
VCOLPAIR = "v[140:141]"
VCOL = ["v140", "v141"]
T = {("A", 0): 142, ("A", 1): 144, ("B", 0): 146, ("B", 1): 148}
K = [150, 152]


def pair(b):
    return "v[%d:%d]" % (b, b + 1)


def tfma(dst, j, q, odd):
    # {t_row0, t_row1} = A[j] * splat(2b) + splat(-b^2): src1 = lo half, src2 = hi half of the same 64-bit pair
    return "v_pk_fma_f32 %s, %s, %s, %s op_sel:[0,0,1] op_sel_hi:[1,0,1]" % (pair(dst), A(j), Rpair(q, odd), Rpair(q, odd))


def step(lines, p, off_next):
    """One step of parity p (K[p], accT[p], vcol[p] are this step's; the other parity's are the previous step's, finished
    here).  off_next: immediate offset of the NEXT step's record relative to the pointer register."""
    kc, ko = K[p], K[1 - p]
    for q in range(NQ):
        X = "A" if q % 2 == 0 else "B"
        Y = "B" if q % 2 == 0 else "A"
        x0, x1, y0, y1 = T[(X, 0)], T[(X, 1)], T[(Y, 0)], T[(Y, 1)]
        lines.append("s_waitcnt lgkmcnt(10)")
        lines.append(tfma(x0, 2 * q, q, False))
        lines.append(tfma(x1, 2 * q + 1, q, True))
        lines.append("ds_read_b128 %s, %s offset:%d" % (Rq(q), PTR, off_next + 16 * q))
        # K-FMAs of the previous quad (q = 0: the last quad of the previous step, into the previous step's K)
        if q == 0:
            k1 = "v_pk_fma_f32 %s, %s, %s, %s" % (pair(ko), pair(y0), E(18), pair(ko))
            k2 = "v_pk_fma_f32 %s, %s, %s, %s" % (pair(ko), pair(y1), E(19), pair(ko))
        elif q == 1:
            k1 = "v_pk_mul_f32 %s, %s, %s" % (pair(kc), pair(y0), E(0))
            k2 = "v_pk_fma_f32 %s, %s, %s, %s" % (pair(kc), pair(y1), E(1), pair(kc))
        else:
            k1 = "v_pk_fma_f32 %s, %s, %s, %s" % (pair(kc), pair(y0), E(2 * q - 2), pair(kc))
            k2 = "v_pk_fma_f32 %s, %s, %s, %s" % (pair(kc), pair(y1), E(2 * q - 1), pair(kc))
        extra = []
        if q == 1:
            # finish of the previous step: rotate the transposed accumulator, then the row product
            extra = ["v_mov_b32_dpp %s, %s wave_rol:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" % (ACCT[p], ACCT[1 - p]),
                     "v_pk_fma_f32 %s, %s, %s, %s op_sel:[0,%d,0] op_sel_hi:[1,%d,1]" % (ACCR, pair(ko), VCOLPAIR, ACCR, 1 - p, 1 - p)]
        elif q == 2:
            extra = ["v_fmac_f32_e32 %s, v%d, v92" % (ACCT[p], ko),
                     "v_fmac_f32_e32 %s, v%d, v93" % (ACCT[p], ko + 1)]
        lines.append(k1)
        lines.append("v_exp_f32_e32 v%d, v%d" % (x0, x0))
        lines.append(k2)
        lines.append("v_exp_f32_e32 v%d, v%d" % (x0 + 1, x0 + 1))
        if extra:
            lines.append(extra[0])
        lines.append("v_exp_f32_e32 v%d, v%d" % (x1, x1))
        if extra:
            lines.append(extra[1])
        lines.append("v_exp_f32_e32 v%d, v%d" % (x1 + 1, x1 + 1))
    lines.append("ds_read_b32 %s, %s offset:%d" % (VCOL[1 - p], PTR, off_next + VCOL_OFF))


def generate():
    L = []
    # ---- prologue: "step -1" state is all zeros, so its finish inside step 0 adds nothing
    for r in (ACCT[1], VCOL[1], "v%d" % K[1], "v%d" % (K[1] + 1), "v%d" % T[("B", 0)], "v%d" % (T[("B", 0)] + 1),
              "v%d" % T[("B", 1)], "v%d" % (T[("B", 1)] + 1)):
        L.append("v_mov_b32_e32 %s, 0" % r)
    for q in range(NQ):
        L.append("ds_read_b128 %s, %s offset:%d" % (Rq(q), PTR, 16 * q))
    L.append("ds_read_b32 %s, %s offset:%d" % (VCOL[0], PTR, VCOL_OFF))
    L.append("s_mov_b32 %[cnt], 32")
    L.append("1:")
    step(L, 0, REC_BYTES)
    step(L, 1, 2 * REC_BYTES)
    L.append("v_add_u32_e32 %s, %d, %s" % (PTR, 2 * REC_BYTES, PTR))
    L.append("s_sub_u32 %[cnt], %[cnt], 1")
    L.append("s_cmp_lg_u32 %[cnt], 0")
    L.append("s_cbranch_scc1 1b")
    # ---- finish of step 63 (parity 1): last two K-FMAs, products, and the final rotation into v95
    ko = K[1]
    L.append("v_pk_fma_f32 %s, %s, %s, %s" % (pair(ko), pair(T[("B", 0)]), E(18), pair(ko)))
    L.append("v_mov_b32_dpp %s, %s wave_rol:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" % ("v97", ACCT[1]))
    L.append("v_pk_fma_f32 %s, %s, %s, %s" % (pair(ko), pair(T[("B", 1)]), E(19), pair(ko)))
    L.append("s_waitcnt lgkmcnt(0)")            # the look-ahead reads of the step after the last (discarded)
    L.append("s_nop 1")
    L.append("v_pk_fma_f32 %s, %s, %s, %s op_sel:[0,1,0] op_sel_hi:[1,1,1]" % (ACCR, pair(ko), VCOLPAIR, ACCR))
    L.append("v_fmac_f32_e32 v97, v%d, v92" % ko)
    L.append("v_fmac_f32_e32 v97, v%d, v93" % (ko + 1))
    L.append("s_nop 1")
    L.append("v_mov_b32_dpp %s, v97 wave_rol:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" % ACCT[0])
    return L


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(
        os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "randomly-projected-additive-gps_amd", "csrc",
        "rpgp_fact_asm_loop.inc")
    L = generate()
    clob = [97] + list(range(100, 154))
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_fact_asm.py — do not edit; the schedule and the register map are described there.\n")
        f.write("#define RPGP_FACT_ASM_LOOP \\\n")
        for ln in L:
            f.write('  "%s\\n" \\\n' % ln)
        f.write('  ""\n')
        f.write("#define RPGP_FACT_ASM_CLOBBERS " + ", ".join('"v%d"' % c for c in clob) + ', "scc"\n')
        n_valu = sum(1 for ln in L if ln.startswith("v_"))
        f.write("// %d lines, %d vector instructions (prologue + 2 unrolled steps + finish)\n" % (len(L), n_valu))
    print("wrote %s: %d asm lines" % (out, len(L)))


if __name__ == "__main__":
    main()
