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

NQ_MAIN = 10               # the J = 20 kernel: 10 quads {2b_e, -b_e^2, 2b_o, -b_o^2} per column record
THIN_JTS = (2, 3, 4, 5, 8, 10)   # round 6: the J-slice loops (generate_thin) for the piece sizes of csrc/rpgp_kernels.hip (kJPieces)


def rec_floats(nq):
    """Floats per column record: nq quads + v_col, padded to whole 16-byte granules with an ODD granule count (conflict-free
    per-lane ds_read_b128): 10 -> 44 (176 B), 5 -> 28, 4 -> 20, 3 -> 20, 2 -> 12, 1 -> 12."""
    f = 4 * nq + 1
    f = (f + 3) // 4 * 4
    if (f // 4) % 2 == 0:
        f += 4
    return f


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


def tset(nq, p, q):
    """Temporary set of quad q of a step of parity p: the sets alternate along the GLOBAL quad sequence (for an even quad
    count that is q % 2; for an odd one it also depends on the step's parity).  Global index -1 (the "previous quad" of the very
    first slot) is odd: set B, which the prologue zeroes."""
    return "A" if (p * nq + q) % 2 == 0 else "B"


def step(lines, p, off_next, nq=NQ_MAIN):
    """One step of parity p (K[p], accT[p], vcol[p] are this step's; the other parity's are the previous step's, finished
    here).  off_next: immediate offset of the NEXT step's record relative to the pointer register."""
    kc, ko = K[p], K[1 - p]
    vcol_off = 16 * nq
    # the finish of the previous step — rotate the transposed accumulator, the row product, the two transposed FMAs — rides
    # in this step's exp gaps: quads 1 and 2 of the J = 20 loop; a thin loop has fewer quads: quads 0 and 1, or (one quad per
    # step) all four in quad 0, the last two behind its exponentials
    fin = ["v_mov_b32_dpp %s, %s wave_rol:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" % (ACCT[p], ACCT[1 - p]),
           "v_pk_fma_f32 %s, %s, %s, %s op_sel:[0,%d,0] op_sel_hi:[1,%d,1]" % (ACCR, pair(ko), VCOLPAIR, ACCR, 1 - p, 1 - p),
           "v_fmac_f32_e32 %s, v%d, v92" % (ACCT[p], ko),
           "v_fmac_f32_e32 %s, v%d, v93" % (ACCT[p], ko + 1)]
    first = 1 if nq >= 3 else 0
    for q in range(nq):
        X, Y = tset(nq, p, q), tset(nq, p, q - 1)
        x0, x1, y0, y1 = T[(X, 0)], T[(X, 1)], T[(Y, 0)], T[(Y, 1)]
        lines.append("s_waitcnt lgkmcnt(%d)" % nq)
        lines.append(tfma(x0, 2 * q, q, False))
        lines.append(tfma(x1, 2 * q + 1, q, True))
        lines.append("ds_read_b128 %s, %s offset:%d" % (Rq(q), PTR, off_next + 16 * q))
        # K-FMAs of the previous quad (q = 0: the last quad of the previous step, into the previous step's K)
        if q == 0:
            k1 = "v_pk_fma_f32 %s, %s, %s, %s" % (pair(ko), pair(y0), E(2 * nq - 2), pair(ko))
            k2 = "v_pk_fma_f32 %s, %s, %s, %s" % (pair(ko), pair(y1), E(2 * nq - 1), pair(ko))
            if nq == 1:       # the only quad is first and last: its products START the previous step's K
                k1 = "v_pk_mul_f32 %s, %s, %s" % (pair(ko), pair(y0), E(0))
        elif q == 1:
            k1 = "v_pk_mul_f32 %s, %s, %s" % (pair(kc), pair(y0), E(0))
            k2 = "v_pk_fma_f32 %s, %s, %s, %s" % (pair(kc), pair(y1), E(1), pair(kc))
        else:
            k1 = "v_pk_fma_f32 %s, %s, %s, %s" % (pair(kc), pair(y0), E(2 * q - 2), pair(kc))
            k2 = "v_pk_fma_f32 %s, %s, %s, %s" % (pair(kc), pair(y1), E(2 * q - 1), pair(kc))
        extra, tail = [], []
        if q == first:
            extra = fin[0:2]
            if nq == 1:
                tail = fin[2:4]
        elif q == first + 1:
            extra = fin[2:4]
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
        lines.extend(tail)
    lines.append("ds_read_b32 %s, %s offset:%d" % (VCOL[1 - p], PTR, off_next + vcol_off))


def generate(nq=NQ_MAIN):
    rec_bytes = 4 * rec_floats(nq)
    L = []
    # ---- prologue: "step -1" state is all zeros, so its finish inside step 0 adds nothing
    for r in (ACCT[1], VCOL[1], "v%d" % K[1], "v%d" % (K[1] + 1), "v%d" % T[("B", 0)], "v%d" % (T[("B", 0)] + 1),
              "v%d" % T[("B", 1)], "v%d" % (T[("B", 1)] + 1)):
        L.append("v_mov_b32_e32 %s, 0" % r)
    for q in range(nq):
        L.append("ds_read_b128 %s, %s offset:%d" % (Rq(q), PTR, 16 * q))
    L.append("ds_read_b32 %s, %s offset:%d" % (VCOL[0], PTR, 16 * nq))
    L.append("s_mov_b32 %[cnt], 32")
    L.append("1:")
    step(L, 0, rec_bytes, nq)
    step(L, 1, 2 * rec_bytes, nq)
    L.append("v_add_u32_e32 %s, %d, %s" % (PTR, 2 * rec_bytes, PTR))
    L.append("s_sub_u32 %[cnt], %[cnt], 1")
    L.append("s_cmp_lg_u32 %[cnt], 0")
    L.append("s_cbranch_scc1 1b")
    # ---- finish of step 63 (parity 1): last two K-FMAs, products, and the final rotation into v95
    ko = K[1]
    last = tset(nq, 1, nq - 1)                 # (2 nq - 1 is odd: always set B)
    if nq == 1:
        L.append("v_pk_mul_f32 %s, %s, %s" % (pair(ko), pair(T[(last, 0)]), E(0)))
    else:
        L.append("v_pk_fma_f32 %s, %s, %s, %s" % (pair(ko), pair(T[(last, 0)]), E(2 * nq - 2), pair(ko)))
    L.append("v_mov_b32_dpp %s, %s wave_rol:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" % ("v97", ACCT[1]))
    L.append("v_pk_fma_f32 %s, %s, %s, %s" % (pair(ko), pair(T[(last, 1)]), E(2 * nq - 1), pair(ko)))
    L.append("s_waitcnt lgkmcnt(0)")            # the look-ahead reads of the step after the last (discarded)
    L.append("s_nop 1")
    L.append("v_pk_fma_f32 %s, %s, %s, %s op_sel:[0,1,0] op_sel_hi:[1,1,1]" % (ACCR, pair(ko), VCOLPAIR, ACCR))
    L.append("v_fmac_f32_e32 v97, v%d, v92" % ko)
    L.append("v_fmac_f32_e32 v97, v%d, v93" % (ko + 1))
    L.append("s_nop 1")
    L.append("v_mov_b32_dpp %s, v97 wave_rol:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" % ACCT[0])
    return L


# ---- thin loops: J-slices of 2 .. 8 projections (round 6; 9 - 10 projections: generate(5) above) --------------------------------
# A rank of north_star's J-split (J = 20 over 8 GPUs: 3 or 2 projections) runs steps of 60 - 100 issue cycles instead of 517: a
# look-ahead of ONE step no longer covers the LDS latency, and a branch every two steps is a tenth of the loop.  Same schedule
# idea, other proportions:
#   * the column record is [jt pairs {2b, -b^2}][v_col][pad]: full quads by ds_read_b128, an odd last projection by ds_read_b64
#     (no padded projection: its two exponentials would be a third of a 3-projection step);
#   * records are requested D steps ahead into D register sets (D = 8 / 4 / 4 / 2 / 2 for jt = 2 / 3 / 4 / 5 / 8), the column's v
#     D - 1 steps ahead into a ring of D registers (it is consumed one step late, in the finish of its step);
#   * D steps per loop trip (64 / D trips).
# Register map: A, E, accR, vrow, ptr, accT, K, temporaries as above; R sets at v100 + k * RS (k = step mod D), the v ring at the
# next even register behind them.
THIN_DEPTH = {2: 8, 3: 4, 4: 4, 5: 2, 6: 2, 7: 2, 8: 2}


def thin_params(jt):
    """Depth, record size and a COMPACT register map (the J = 20 map pins v10 - v153: 154 registers, three waves per SIMD; a
    3-projection loop needs ~70, and with six waves per SIMD the other waves cover what the look-ahead does not):
      A pairs | E pairs | accR | vrow | ptr | accT0 accT1 | tmp | R sets (D x RS) | v ring (D) | temporaries (8) | K (4)"""
    D = THIN_DEPTH[jt]
    nfull, half = jt // 2, jt % 2
    RS = 4 * nfull + 2 * half
    f = (2 * jt + 1 + 3) // 4 * 4
    if (f // 4) % 2 == 0:
        f += 4
    A0 = 10
    E0 = A0 + 2 * jt
    nxt = E0 + 2 * jt
    R = dict(jt=jt, D=D, nfull=nfull, half=half, RS=RS, rec_floats=f, A0=A0, E0=E0, ACCR=nxt, VROW=nxt + 2, PTR=nxt + 4,
             ACCT=(nxt + 5, nxt + 6), TMP=nxt + 7, RB=nxt + 8)
    R["VCB"] = R["RB"] + D * RS
    R["TB"] = (R["VCB"] + D + 1) // 2 * 2
    R["KB"] = R["TB"] + 8
    R["TOP"] = R["KB"] + 4
    assert D >= 2 and R["RB"] % 2 == 0 and R["VCB"] % 2 == 0 and (D * (nfull + half)) % 2 == 0, (jt, R)
    return R


def generate_thin(jt):
    P = thin_params(jt)
    D, nfull, half, RS, VCB, RB = P["D"], P["nfull"], P["half"], P["RS"], P["VCB"], P["RB"]
    rec_bytes = 4 * P["rec_floats"]
    nslots = nfull + half
    ACCRp, PTRr, TMPr = pair(P["ACCR"]), "v%d" % P["PTR"], "v%d" % P["TMP"]
    ACCTr = ["v%d" % P["ACCT"][0], "v%d" % P["ACCT"][1]]
    VR0, VR1 = "v%d" % P["VROW"], "v%d" % (P["VROW"] + 1)
    Kr = [P["KB"], P["KB"] + 2]
    Tr = {("A", 0): P["TB"], ("A", 1): P["TB"] + 2, ("B", 0): P["TB"] + 4, ("B", 1): P["TB"] + 6}
    L = []

    def Aj(j):
        return pair(P["A0"] + 2 * j)

    def Ej(j):
        return pair(P["E0"] + 2 * j)

    def rpair(k, i, odd=False):                # 64-bit pair of projection 2i (+1) in register set k
        return pair(RB + k * RS + 4 * i + (2 if odd else 0))

    def vc(step):                              # (pair, half) of the ring register holding the v of `step`'s column
        r = VCB + step % D
        return pair(r & ~1), r & 1

    def tf(dst, j, src):
        return "v_pk_fma_f32 %s, %s, %s, %s op_sel:[0,0,1] op_sel_hi:[1,0,1]" % (pair(dst), Aj(j), src, src)

    def read_rec(k, i, off):
        b = RB + k * RS + 4 * i
        if i < nfull:
            return "ds_read_b128 v[%d:%d], %s offset:%d" % (b, b + 3, PTRr, off + 16 * i)
        return "ds_read_b64 v[%d:%d], %s offset:%d" % (b, b + 1, PTRr, off + 16 * i)

    def kops(pend):
        out = []
        for (treg, j, kreg, first) in pend:
            if first:
                out.append("v_pk_mul_f32 %s, %s, %s" % (pair(kreg), pair(treg), Ej(j)))
            else:
                out.append("v_pk_fma_f32 %s, %s, %s, %s" % (pair(kreg), pair(treg), Ej(j), pair(kreg)))
        return out

    def slot_temps(g):                         # temporaries alternate along the GLOBAL slot sequence (D * nslots is even)
        X = "A" if g % 2 == 0 else "B"
        return Tr[(X, 0)], Tr[(X, 1)]

    def pend_of(g, i, kreg):                   # the K operations slot g (projection pair i) leaves for the next slot
        x0, x1 = slot_temps(g)
        return [(x0, 2 * i, kreg, i == 0)] + ([(x1, 2 * i + 1, kreg, False)] if i < nfull else [])

    # ---- prologue: "step -1" is all zeros (its finish inside step 0 adds nothing); records of steps 0 .. D - 1 and the v of
    # steps 0 .. D - 2 are requested
    for r in [ACCTr[1], "v%d" % Kr[1], "v%d" % (Kr[1] + 1), "v%d" % (VCB + (-1) % D)] + ["v%d" % r for r in range(P["TB"], P["TB"] + 8)]:
        L.append("v_mov_b32_e32 %s, 0" % r)
    # (request order = the steady state's: the records of a step, then the v of the column D - 1 steps behind them — a dummy
    #  for "step -1", into the scratch register — so that the constant wait counts of the loop hold from the first slot on)
    for s0 in range(D):
        for i in range(nslots):
            L.append(read_rec(s0, i, s0 * rec_bytes))
        if s0 == 0:
            L.append("ds_read_b32 %s, %s offset:%d" % (TMPr, PTRr, 8 * jt))
        else:
            L.append("ds_read_b32 v%d, %s offset:%d" % (VCB + s0 - 1, PTRr, (s0 - 1) * rec_bytes + 8 * jt))
    L.append("s_mov_b32 %%[cnt], %d" % (64 // D))
    L.append("1:")
    for s in range(D):                         # one trip = D steps; step s uses register set s and refills it for step s + D
        p = s % 2
        kc, ko = Kr[p], Kr[1 - p]
        vpair, vhalf = vc(s - 1)
        fin = ["v_mov_b32_dpp %s, %s wave_rol:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" % (ACCTr[p], ACCTr[1 - p]),
               "v_pk_fma_f32 %s, %s, %s, %s op_sel:[0,%d,0] op_sel_hi:[1,%d,1]" % (ACCRp, pair(ko), vpair, ACCRp, vhalf, vhalf),
               "v_fmac_f32_e32 %s, v%d, %s" % (ACCTr[p], ko, VR0),
               "v_fmac_f32_e32 %s, v%d, %s" % (ACCTr[p], ko + 1, VR1)]
        for i in range(nslots):
            g = s * nslots + i
            x0, x1 = slot_temps(g)
            full = i < nfull
            # outstanding LDS requests younger than the one this slot needs; slot 0 also needs the v of the PREVIOUS step's
            # column (requested D - 1 steps ago, at the end of a step): (D - 1)(nslots + 1) requests were issued since
            need = (D - 1) * (nslots + 1) if i == 0 else D * (nslots + 1) - 1
            L.append("s_waitcnt lgkmcnt(%d)" % min(15, need))
            L.append(tf(x0, 2 * i, rpair(s, i, False)))
            if full:
                L.append(tf(x1, 2 * i + 1, rpair(s, i, True)))
            L.append(read_rec(s, i, (s + D) * rec_bytes))
            prev_i = (i - 1) % nslots
            prev_k = kc if i > 0 else ko
            ne = kops(pend_of(g - 1, prev_i, prev_k))
            if i == 0:
                ne += fin if nslots == 1 else fin[0:2]
            elif i == 1:
                ne += fin[2:4]
            exps = ["v_exp_f32_e32 v%d, v%d" % (x0, x0), "v_exp_f32_e32 v%d, v%d" % (x0 + 1, x0 + 1)]
            if full:
                exps += ["v_exp_f32_e32 v%d, v%d" % (x1, x1), "v_exp_f32_e32 v%d, v%d" % (x1 + 1, x1 + 1)]
            while ne or exps:
                if ne:
                    L.append(ne.pop(0))
                if exps:
                    L.append(exps.pop(0))
        L.append("ds_read_b32 v%d, %s offset:%d" % (VCB + (s + D - 1) % D, PTRr, (s + D - 1) * rec_bytes + 8 * jt))
    L.append("v_add_u32_e32 %s, %d, %s" % (PTRr, D * rec_bytes, PTRr))
    L.append("s_sub_u32 %[cnt], %[cnt], 1")
    L.append("s_cmp_lg_u32 %[cnt], 0")
    L.append("s_cbranch_scc1 1b")
    # ---- finish of step 63 (parity 1): its last slot's K operations, the products, the final rotation into accT0
    ko = Kr[1]
    gl = D * nslots - 1                        # the last slot of a trip
    pk = kops(pend_of(gl, nslots - 1, ko))
    vpair, vhalf = vc(63)
    L.append(pk[0])
    L.append("v_mov_b32_dpp %s, %s wave_rol:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" % (TMPr, ACCTr[1]))
    L.extend(pk[1:])
    L.append("s_waitcnt lgkmcnt(0)")            # the look-ahead requests of the steps after the last (discarded)
    L.append("s_nop 1")
    L.append("v_pk_fma_f32 %s, %s, %s, %s op_sel:[0,%d,0] op_sel_hi:[1,%d,1]" % (ACCRp, pair(ko), vpair, ACCRp, vhalf, vhalf))
    L.append("v_fmac_f32_e32 %s, v%d, %s" % (TMPr, ko, VR0))
    L.append("v_fmac_f32_e32 %s, v%d, %s" % (TMPr, ko + 1, VR1))
    L.append("s_nop 1")
    L.append("v_mov_b32_dpp %s, %s wave_rol:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" % (ACCTr[0], TMPr))
    return L, P


def selftest():
    """Executes the generated text of every loop (10 quads and the thin ones) on the 64-lane numpy interpreter of
    tools/gen_bil_asm.py (extended by the four instructions only this loop uses) against the direct formula: register map,
    pipeline and the finish of the last step are checked on the CPU before the GPU sees them."""
    import re
    import numpy as np
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from gen_bil_asm import Machine

    class M(Machine):
        def run(self, lines):
            f32 = np.float32
            rest_lines = []
            for ln in lines:
                ln = ln.strip()
                op = ln.split(None, 1)[0] if ln else ""
                if op == "v_exp_f32_e32":
                    d, s_ = [t.strip() for t in ln.split(None, 1)[1].split(",")]
                    self.v[int(d[1:])] = np.exp2(self.v[int(s_[1:])].astype(np.float64)).astype(f32)
                elif op == "v_fmac_f32_e32":
                    d, a, b = [t.strip() for t in ln.split(None, 1)[1].split(",")]
                    self.v[int(d[1:])] = (self.v[int(a[1:])] * self.v[int(b[1:])] + self.v[int(d[1:])]).astype(f32)
                elif op == "ds_read_b32":
                    m_ = re.fullmatch(r"v(\d+),\s*v(\d+)\s+offset:(\d+)", ln.split(None, 1)[1].strip())
                    addr = (self.v[int(m_.group(2))].view(np.uint32).astype(np.int64) + int(m_.group(3))) // 4
                    self.v[int(m_.group(1))] = self.lds[addr]
                elif op.endswith(":"):
                    continue
                else:
                    Machine.run(self, [ln])

    rng = np.random.default_rng(1)
    worst = 0.0
    for nq in (NQ_MAIN, 5):
        L = generate(nq)
        i0, i1 = L.index("1:"), L.index("s_cbranch_scc1 1b")
        pro, body, epi = L[:i0], L[i0 + 1:i1 + 1], L[i1 + 1:]
        rf = rec_floats(nq)
        jp = 2 * nq                                           # projections incl. the pad of an odd count
        b2 = rng.standard_normal((64, jp)).astype(np.float32)          # 2b of column c, projection j
        nb2 = -(0.25 * b2 * b2).astype(np.float32)                      # -b^2
        vcol = rng.standard_normal(64).astype(np.float32)
        lds = np.zeros(128 * rf + 64, dtype=np.float32)
        for rec in range(128):
            c = rec % 64
            for q in range(nq):
                lds[rec * rf + 4 * q: rec * rf + 4 * q + 4] = [b2[c, 2 * q], nb2[c, 2 * q], b2[c, 2 * q + 1], nb2[c, 2 * q + 1]]
            lds[rec * rf + 4 * nq] = vcol[c]
        a = (rng.standard_normal((2, 64, jp)) * 0.7).astype(np.float32)
        ea = np.exp2(-(a.astype(np.float64) ** 2)).astype(np.float32)
        vrow = rng.standard_normal((2, 64)).astype(np.float32)
        acc0 = rng.standard_normal((2, 64)).astype(np.float32)
        m = M(lds)
        for j in range(jp):
            m.v[10 + 2 * j], m.v[11 + 2 * j] = a[0, :, j], a[1, :, j]
            m.v[50 + 2 * j], m.v[51 + 2 * j] = ea[0, :, j], ea[1, :, j]
        m.v[90], m.v[91] = acc0[0], acc0[1]
        m.v[92], m.v[93] = vrow[0], vrow[1]
        m.v[94] = (np.arange(64, dtype=np.uint32) * (4 * rf)).view(np.float32)
        m.run(pro)
        for _ in range(32):
            m.run(body)
        m.run(epi)
        # direct formula (float64): K[r][l][c] = sum_j exp2(a 2b - b^2) Ea
        t = a.astype(np.float64)[:, :, None, :] * b2.astype(np.float64)[None, None, :, :] + nb2.astype(np.float64)[None, None, :, :]
        Kd = (np.exp2(t) * ea.astype(np.float64)[:, :, None, :]).sum(-1)               # (2, 64 lanes, 64 columns)
        accR = acc0.astype(np.float64) + (Kd * vcol.astype(np.float64)[None, None, :]).sum(-1)
        accT = (Kd * vrow.astype(np.float64)[:, :, None]).sum((0, 1))                 # per column; lane l ends with column l
        eR = np.abs(np.stack([m.v[90], m.v[91]]) - accR).max() / np.abs(accR).max()
        eT = np.abs(m.v[95] - accT).max() / np.abs(accT).max()
        ptr_ok = np.array_equal(m.v[94].view(np.uint32), (np.arange(64, dtype=np.uint32) + 64) * np.uint32(4 * rf))
        n_valu = sum(1 for ln in body if ln.startswith("v_") and not ln.startswith("v_add_u32"))
        print("nq %2d: record %2d floats, %3d vector instructions per two steps, rel err row products %.2e transposed %.2e, "
              "pointer advanced by 64 records: %s" % (nq, rf, n_valu, eR, eT, ptr_ok))
        assert eR < 2e-5 and eT < 2e-5 and ptr_ok, (nq, eR, eT, ptr_ok)
        worst = max(worst, eR, eT)
    # ---- the J-slice loops (exact projection counts, deep look-ahead)
    class M2(M):
        def run(self, lines):
            rest = []
            for ln in lines:
                ln = ln.strip()
                if ln.startswith("ds_read_b64"):
                    m_ = re.fullmatch(r"v\[(\d+):(\d+)\],\s*v(\d+)\s+offset:(\d+)", ln.split(None, 1)[1].strip())
                    addr = (self.v[int(m_.group(3))].view(np.uint32).astype(np.int64) + int(m_.group(4))) // 4
                    self.v[int(m_.group(1))] = self.lds[addr]
                    self.v[int(m_.group(1)) + 1] = self.lds[addr + 1]
                else:
                    M.run(self, [ln])

    for jt in THIN_JTS:
        if jt == 10:
            continue                          # served by the 5-quad loop checked above
        L, P = generate_thin(jt)
        i0, i1 = L.index("1:"), L.index("s_cbranch_scc1 1b")
        pro, body, epi = L[:i0], L[i0 + 1:i1 + 1], L[i1 + 1:]
        rf, D = P["rec_floats"], P["D"]
        b2 = rng.standard_normal((64, jt)).astype(np.float32)
        nb2 = -(0.25 * b2 * b2).astype(np.float32)
        vcol = rng.standard_normal(64).astype(np.float32)
        lds = np.full((136 + 2) * rf + 64, np.float32(np.nan), dtype=np.float32)      # (what is never consumed may be anything)
        for rec in range(136):
            c = rec % 64
            for j in range(jt):
                lds[rec * rf + 2 * j], lds[rec * rf + 2 * j + 1] = b2[c, j], nb2[c, j]
            lds[rec * rf + 2 * jt] = vcol[c]
        a = (rng.standard_normal((2, 64, jt)) * 0.7).astype(np.float32)
        ea = np.exp2(-(a.astype(np.float64) ** 2)).astype(np.float32)
        vrow = rng.standard_normal((2, 64)).astype(np.float32)
        acc0 = rng.standard_normal((2, 64)).astype(np.float32)
        m = M2(lds)
        m.v[:] = np.float32(np.nan)                           # every register the loop reads must have been written by it or by us
        for j in range(jt):
            m.v[P["A0"] + 2 * j], m.v[P["A0"] + 2 * j + 1] = a[0, :, j], a[1, :, j]
            m.v[P["E0"] + 2 * j], m.v[P["E0"] + 2 * j + 1] = ea[0, :, j], ea[1, :, j]
        m.v[P["ACCR"]], m.v[P["ACCR"] + 1] = acc0[0], acc0[1]
        m.v[P["VROW"]], m.v[P["VROW"] + 1] = vrow[0], vrow[1]
        m.v[P["PTR"]] = (np.arange(64, dtype=np.uint32) * (4 * rf)).view(np.float32)
        m.run(pro)
        for _ in range(64 // D):
            m.run(body)
        m.run(epi)
        t = a.astype(np.float64)[:, :, None, :] * b2.astype(np.float64)[None, None, :, :] + nb2.astype(np.float64)[None, None, :, :]
        Kd = (np.exp2(t) * ea.astype(np.float64)[:, :, None, :]).sum(-1)
        accR = acc0.astype(np.float64) + (Kd * vcol.astype(np.float64)[None, None, :]).sum(-1)
        accT = (Kd * vrow.astype(np.float64)[:, :, None]).sum((0, 1))
        eR = np.abs(np.stack([m.v[P["ACCR"]], m.v[P["ACCR"] + 1]]) - accR).max() / np.abs(accR).max()
        eT = np.abs(m.v[P["ACCT"][0]] - accT).max() / np.abs(accT).max()
        ptr_ok = np.array_equal(m.v[P["PTR"]].view(np.uint32), (np.arange(64, dtype=np.uint32) + 64) * np.uint32(4 * rf))
        used = set(int(x) for ln in L for x in re.findall(r"v(\d+)", re.sub(r"v\[(\d+):(\d+)\]", lambda mm: " ".join("v%d" % q for q in range(int(mm.group(1)), int(mm.group(2)) + 1)), ln)))
        assert max(used) < P["TOP"], (jt, max(used), P["TOP"])
        n_valu = sum(1 for ln in body if ln.startswith("v_") and not ln.startswith("v_add_u32"))
        n_exp = sum(1 for ln in body if ln.startswith("v_exp"))
        cyc = (n_exp * 8.2 + (n_valu - n_exp) * 4.4) / (D * 2 * jt)
        print("jt %2d: depth %d, record %2d floats, registers v10 - v%d, %3d vector instructions per %d steps (~%.1f issue cycles per "
              "64 pair-terms), rel err row products %.2e transposed %.2e, pointer ok: %s" % (jt, D, rf, P["TOP"] - 1, n_valu, D, cyc, eR, eT, ptr_ok))
        assert eR < 2e-5 and eT < 2e-5 and ptr_ok, (jt, eR, eT, ptr_ok)
        worst = max(worst, eR, eT)
    print("selftest ok (worst %.2e)" % worst)


def main():
    if "--selftest" in sys.argv:
        selftest()
        return
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
        # the J-slice loops (the ranks of north_star's J-split, J = d models)
        for jt in THIN_JTS:
            if jt == 10:
                Lt, rf, depth = generate(5), rec_floats(5), 1
            else:
                Lt, P = generate_thin(jt)
                rf, depth = P["rec_floats"], P["D"]
            f.write("#define RPGP_FACT_ASM_LOOP_JT%d \\\n" % jt)
            for ln in Lt:
                f.write('  "%s\\n" \\\n' % ln)
            f.write('  ""\n')
            f.write("#define RPGP_FACT_ASM_REC_FLOATS_JT%d %d\n" % (jt, rf))
            f.write("#define RPGP_FACT_ASM_DEPTH_JT%d %d\n" % (jt, depth))
            if jt != 10:
                # the compact register map of this loop, as inline-asm constraint strings and a clobber list
                rr = lambda lo, n: '"{v[%d:%d]}"' % (lo, lo + n - 1) if n > 1 else '"{v%d}"' % lo
                f.write("#define RPGP_FACT_ASM_CA_JT%d %s\n" % (jt, rr(P["A0"], 2 * jt)))
                f.write("#define RPGP_FACT_ASM_CE_JT%d %s\n" % (jt, rr(P["E0"], 2 * jt)))
                f.write("#define RPGP_FACT_ASM_CACCR_JT%d %s\n" % (jt, '"+{v[%d:%d]}"' % (P["ACCR"], P["ACCR"] + 1)))
                f.write("#define RPGP_FACT_ASM_CVROW_JT%d %s\n" % (jt, rr(P["VROW"], 2)))
                f.write("#define RPGP_FACT_ASM_CPTR_JT%d %s\n" % (jt, '"+{v%d}"' % P["PTR"]))
                f.write("#define RPGP_FACT_ASM_CACCT_JT%d %s\n" % (jt, '"={v%d}"' % P["ACCT"][0]))
                cl = [P["ACCT"][1], P["TMP"]] + list(range(P["RB"], P["TOP"]))
                f.write("#define RPGP_FACT_ASM_CLOB_JT%d %s, \"scc\"\n" % (jt, ", ".join('"v%d"' % c for c in cl)))
    print("wrote %s: %d asm lines (+ J-slice loops for %s projections)" % (out, len(L), ", ".join(str(n) for n in THIN_JTS)))


if __name__ == "__main__":
    main()
