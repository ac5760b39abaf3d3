#!/usr/bin/env python3
"""Generates csrc/rpgp_bil_asm_loop.inc: the hand-scheduled gfx950 inner loop of the symmetric bilinear-derivative sweep
(`bilinear_sym_asm_kernel`, csrc/rpgp_bil_asm.hip; JT = 20 projections, TT = 12 right-hand-side slots, two rows per lane) —
the backward pass of one optimiser step in the CG regime (GAMFunction.backward,
gp_models/kernels/memory_efficient_gam_kernel.py:33-59; replaces GPyTorch's `_quad_form_derivative`).

One execution = one 64-column subtile = 64 steps; at step s lane l meets column (l + s) of the subtile (the LDS image holds
the 64 column records {z[20], L[12], R[12]} followed by a copy of the first 66: running pointer + immediate offsets).
Per step and lane, for the lane's two rows r (packed lanes of the v_pk_* instructions):
    S_r   = sum_t  li[r][t] pr[t] + ri[r][t] pl[t]                           24 packed FMAs (computed ONE STEP AHEAD, two
                                                                             partial sums: no dependent chain)
    per projection j:   d = a[r][j] - z[j];  e = exp2(-d^2);  ks += e;  se = S e
                        accG[r][j] += se d;   tg[j] = rot(tg[j]) - se_0 d_0 - se_1 d_1   (tg: the travelling accumulators of
                                                                             the transposed half, DPP-rotated one lane per step)
    accS[r] += S ks;  tg[20] = rot(tg[20]) + S_0 ks_0 + S_1 ks_1
Schedule: a three-deep software pipeline over the projections (subtract / square for j+2, the two exponentials for j+1, the
five consumers for j in one slot) that runs across the step boundary, every ds_read issued a full step before its use, the
S FMAs of the next step and the closing operations of the previous step slotted between dependent instructions.

`python3 tools/gen_bil_asm.py --selftest` executes the generated text on a small interpreter of the instruction subset
(numpy, 64 lanes) against the direct formula — register map and pipeline are checked on the CPU before the GPU sees them.
"""
import os
import re
import sys

import numpy as np

JT, TT = 20, 12
REC_FLOATS = JT + 2 * TT          # 44
REC_BYTES = 4 * REC_FLOATS        # 176: (176 / 16) odd -> conflict-free per-lane ds_read_b128
NQUAD = REC_FLOATS // 4           # 11: quads 0-4 z, 5-7 L (pl), 8-10 R (pr)
NREC = 130                        # 64 records + a copy of the first 66 (look-ahead: z one record, L / R two records)

# ---- register map ----------------------------------------------------------------------------------------------------
A0, LR0, RR0, G0, ACCS, TG0, PTR, REC0 = 10, 50, 74, 98, 138, 140, 161, 164
Z0, PL0, PR0 = REC0, REC0 + 20, REC0 + 32
S = [208, 210]                    # S of even / odd steps (ping-pong: the next step's S is formed during this one)
P1 = 212                          # second partial sum of the next step's S
KS = [214, 246]                   # sum_j e of even / odd steps
SETS = [(216 + 6 * k, 218 + 6 * k, 220 + 6 * k) for k in range(4)]    # (DD, ME, SE) of projection j: set j % 4
TMP = [240, 241]
SK = 242
TGOUT0 = 140                      # outputs = the TG registers themselves (final rotation through TMP)
NVGPR_TOP = 248


def pr2(b):
    return "v[%d:%d]" % (b, b + 1)


def A(j):
    return pr2(A0 + 2 * j)


def Lr(t):
    return pr2(LR0 + 2 * t)


def Rr(t):
    return pr2(RR0 + 2 * t)


def G(j):
    return pr2(G0 + 2 * j)


def splat(reg):
    """(pair containing `reg`, half index) for an op_sel broadcast of one 32-bit register."""
    base = reg & ~1
    return pr2(base), reg & 1


class Step:
    """Emits one step; `reads` collects (instruction index, tag) so that the s_waitcnt counts can be derived afterwards."""

    def __init__(self, parity, u):
        self.p, self.u = parity, u
        self.ins = []              # dicts: text, issues (tag or None), needs (list of tags)

    def emit(self, text, issues=None, needs=()):
        self.ins.append({"text": text, "issues": issues, "needs": list(needs)})


def zquad_of(j):
    return j // 4


def emit_A1(st, j, setk):
    dd = SETS[setk][0]
    zp, sel = splat(Z0 + j)
    st.emit("v_pk_add_f32 %s, %s, %s op_sel:[0,%d] op_sel_hi:[1,%d] neg_lo:[0,1] neg_hi:[0,1]" % (pr2(dd), A(j), zp, sel, sel),
            needs=[("z", zquad_of(j))])


def emit_A2(st, setk):
    dd, me = SETS[setk][0], SETS[setk][1]
    st.emit("v_pk_mul_f32 %s, %s, %s" % (pr2(me), pr2(dd), pr2(dd)))


def emit_B(st, setk, half):
    me = SETS[setk][1] + half
    st.emit("v_exp_f32_e64 v%d, -v%d" % (me, me))


def gen_step(p, u):
    """Step of parity p (u = position inside the two-step loop body: immediate offsets are relative to the pointer of u = 0)."""
    st = Step(p, u)
    s_cur, s_next = S[p], S[1 - p]
    ks_cur, ks_prev = KS[p], KS[1 - p]
    s_prev = S[1 - p]
    off_z = (u + 1) * REC_BYTES       # next record's z quads
    off_lr = (u + 2) * REC_BYTES      # the record after next: L / R quads
    for j in range(JT):
        dd, me, se = SETS[j % 4]
        ja, jb = j + 2, j + 1         # projections whose front stages run in this slot (>= JT: the NEXT step's)
        extra = []
        if j == 0:
            # closing operations of the PREVIOUS step (its S and ks are still intact: S_next's first write is in slot 2)
            extra = ["v_pk_mul_f32 %s, %s, %s" % (pr2(SK), pr2(s_prev), pr2(ks_prev)),
                     "v_mov_b32_dpp v%d, v%d wave_rol:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" % (TMP[1], TG0 + JT)]
        elif j == 1:
            extra = ["v_pk_add_f32 %s, %s, %s" % (pr2(ACCS), pr2(ACCS), pr2(SK)),
                     "v_add_f32_e32 v%d, v%d, v%d" % (TG0 + JT, TMP[1], SK)]
        elif 2 <= j <= 13:
            t = j - 2
            prp, prs = splat(PR0 + t)
            plp, pls = splat(PL0 + t)
            if t == 0:
                e1 = "v_pk_mul_f32 %s, %s, %s op_sel:[0,%d] op_sel_hi:[1,%d]" % (pr2(s_next), Lr(t), prp, prs, prs)
                e2 = "v_pk_mul_f32 %s, %s, %s op_sel:[0,%d] op_sel_hi:[1,%d]" % (pr2(P1), Rr(t), plp, pls, pls)
            else:
                e1 = "v_pk_fma_f32 %s, %s, %s, %s op_sel:[0,%d,0] op_sel_hi:[1,%d,1]" % (pr2(s_next), Lr(t), prp, pr2(s_next), prs, prs)
                e2 = "v_pk_fma_f32 %s, %s, %s, %s op_sel:[0,%d,0] op_sel_hi:[1,%d,1]" % (pr2(P1), Rr(t), plp, pr2(P1), pls, pls)
            extra = [(e1, [("lr", 8 - 5 + t // 4)]), (e2, [("lr", t // 4)])]     # lr tags: 0-2 = L quads (5-7), 3-5 = R quads (8-10)
        elif j == 14:
            extra = ["v_pk_add_f32 %s, %s, %s" % (pr2(s_next), pr2(s_next), pr2(P1)),
                     "v_add_f32_e32 v%d, v%d, v%d" % (TG0 + JT, TG0 + JT, SK + 1)]
        ex = []
        for e in extra:
            ex.append(e if isinstance(e, tuple) else (e, []))
        # ---- the slot
        emit_A1(st, ja % JT, ja % 4)
        emit_B(st, jb % 4, 0)
        if j == 0:
            st.emit("v_pk_mul_f32 %s, %s, 1.0 op_sel_hi:[1,0]" % (pr2(ks_cur), pr2(me)))
        else:
            st.emit("v_pk_add_f32 %s, %s, %s" % (pr2(ks_cur), pr2(ks_cur), pr2(me)))
        emit_A2(st, ja % 4)
        emit_B(st, jb % 4, 1)
        st.emit("v_pk_mul_f32 %s, %s, %s" % (pr2(se), pr2(s_cur), pr2(me)))
        st.emit("v_mov_b32_dpp v%d, v%d wave_rol:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" % (TMP[0], TG0 + j))
        if len(ex) > 0:
            st.emit(ex[0][0], needs=ex[0][1])
        st.emit("v_pk_fma_f32 %s, %s, %s, %s" % (G(j), pr2(se), pr2(dd), G(j)))
        st.emit("v_fma_f32 v%d, -v%d, v%d, v%d" % (TG0 + j, se, dd, TMP[0]))
        if len(ex) > 1:
            st.emit(ex[1][0], needs=ex[1][1])
        st.emit("v_fma_f32 v%d, -v%d, v%d, v%d" % (TG0 + j, se + 1, dd + 1, TG0 + j))
        # ---- refills: a z quad right after its last reader (A1 of projection 4q + 3 runs in slot 4q + 1), the L / R quads
        # after the S FMAs of their four columns (slots 5, 9, 13)
        if j % 4 == 1:
            q = (j - 1) // 4
            st.emit("ds_read_b128 v[%d:%d], v%d offset:%d" % (REC0 + 4 * q, REC0 + 4 * q + 3, PTR, off_z + 16 * q), issues=("z", q))
        if j in (5, 9, 13):
            i = (j - 5) // 4
            st.emit("ds_read_b128 v[%d:%d], v%d offset:%d" % (REC0 + 4 * (5 + i), REC0 + 4 * (5 + i) + 3, PTR, off_lr + 16 * (5 + i)),
                    issues=("lr", i))
            st.emit("ds_read_b128 v[%d:%d], v%d offset:%d" % (REC0 + 4 * (8 + i), REC0 + 4 * (8 + i) + 3, PTR, off_lr + 16 * (8 + i)),
                    issues=("lr", 3 + i))
    return st


def add_waits(steps):
    """Steady state: every step issues the same eleven reads in the same order.  Before the first consumer of a tag insert
    s_waitcnt lgkmcnt(n) with n = the number of reads issued after the MOST RECENT read of that tag (in the previous step, or
    earlier in this one for the look-ahead stages of the next step's projections), capped at 15 (the counter's range: a
    tighter wait than needed is still correct); skipped when an earlier wait already covers that read."""
    out = []
    order = [i["issues"] for i in steps[0].ins if i["issues"] is not None]      # same order in every step
    nreads = len(order)
    for st in steps:
        seq_of = {tag: k for k, tag in enumerate(order)}      # sequence numbers: previous step 0 .. 10, this step 11 ..
        issued = nreads
        covered = -1                                          # every read with sequence number <= covered is complete
        lines = []
        for ins in st.ins:
            for tag in ins["needs"]:
                k = seq_of[tag]
                if k <= covered:
                    continue
                younger = issued - 1 - k
                lines.append("s_waitcnt lgkmcnt(%d)" % min(younger, 15))
                covered = k
            lines.append(ins["text"])
            if ins["issues"] is not None:
                seq_of[ins["issues"]] = issued
                issued += 1
        out.append(lines)
    return out


def generate():
    """Returns (prologue, body, epilogue) as lists of instruction lines (body = two steps; executed 32 times)."""
    pro = []
    zero = [TG0 + q for q in range(JT + 1)] + [S[1], S[1] + 1, KS[1], KS[1] + 1, SK, SK + 1, TMP[1]]
    for r in zero:
        pro.append("v_mov_b32_e32 v%d, 0" % r)
    for q in range(NQUAD):
        pro.append("ds_read_b128 v[%d:%d], v%d offset:%d" % (REC0 + 4 * q, REC0 + 4 * q + 3, PTR, 16 * q))
    pro.append("s_waitcnt lgkmcnt(0)")
    # S of step 0 from record 0 (two partial sums), then the L / R quads of record 1
    for t in range(TT):
        prp, prs = splat(PR0 + t)
        plp, pls = splat(PL0 + t)
        if t == 0:
            pro.append("v_pk_mul_f32 %s, %s, %s op_sel:[0,%d] op_sel_hi:[1,%d]" % (pr2(S[0]), Lr(t), prp, prs, prs))
            pro.append("v_pk_mul_f32 %s, %s, %s op_sel:[0,%d] op_sel_hi:[1,%d]" % (pr2(P1), Rr(t), plp, pls, pls))
        else:
            pro.append("v_pk_fma_f32 %s, %s, %s, %s op_sel:[0,%d,0] op_sel_hi:[1,%d,1]" % (pr2(S[0]), Lr(t), prp, pr2(S[0]), prs, prs))
            pro.append("v_pk_fma_f32 %s, %s, %s, %s op_sel:[0,%d,0] op_sel_hi:[1,%d,1]" % (pr2(P1), Rr(t), plp, pr2(P1), pls, pls))
    pro.append("v_pk_add_f32 %s, %s, %s" % (pr2(S[0]), pr2(S[0]), pr2(P1)))
    # front stages of projections 0 and 1 of step 0 (in the loop they run in slots 18 / 19 of the step before)
    st = Step(0, 0)
    emit_A1(st, 0, 0)
    emit_A2(st, 0)
    emit_A1(st, 1, 1)
    emit_A2(st, 1)
    emit_B(st, 0, 0)
    emit_B(st, 0, 1)
    pro += [i["text"] for i in st.ins]
    # reads in the order (and number) a loop step leaves behind: z quads 0-4 of the CURRENT record are already in place, so
    # the dummy re-reads keep the counter arithmetic of the steady state exact: [z0, z? ...] — simplest: re-issue the step's
    # eleven reads for record "0 + look-ahead" in the loop's own order
    order = [i for i in gen_step(0, -1).ins if i["issues"] is not None]
    for i in order:
        pro.append(i["text"])
    steps = [gen_step(0, 0), gen_step(1, 1)]
    body = []
    for lines in add_waits(steps):
        body += lines
    body.append("v_add_u32_e32 v%d, %d, v%d" % (PTR, 2 * REC_BYTES, PTR))
    epi = []
    # closing operations of step 63 (parity 1) and the final rotation of the 21 travelling sums
    epi.append("s_waitcnt lgkmcnt(0)")
    epi.append("v_pk_mul_f32 %s, %s, %s" % (pr2(SK), pr2(S[1]), pr2(KS[1])))
    epi.append("v_mov_b32_dpp v%d, v%d wave_rol:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" % (TMP[1], TG0 + JT))
    epi.append("v_pk_add_f32 %s, %s, %s" % (pr2(ACCS), pr2(ACCS), pr2(SK)))
    epi.append("v_add_f32_e32 v%d, v%d, v%d" % (TG0 + JT, TMP[1], SK))
    epi.append("v_add_f32_e32 v%d, v%d, v%d" % (TG0 + JT, TG0 + JT, SK + 1))
    epi.append("s_nop 1")
    for q in range(JT + 1):
        epi.append("v_mov_b32_dpp v%d, v%d wave_rol:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" % (TMP[q % 2], TG0 + q))
        epi.append("v_mov_b32_e32 v%d, v%d" % (TG0 + q, TMP[q % 2]))
    return pro, body, epi


# ---- interpreter of the instruction subset (self-test) -----------------------------------------------------------------
class Machine:
    def __init__(self, lds):
        self.v = np.zeros((256, 64), dtype=np.float32)
        self.lds = lds                      # float32 array (LDS image), byte addresses / 4

    def _src(self, tok, half):
        """value array (64,) of half `half` (0 / 1) of a source token: v[a:b], vN, or a constant."""
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        if m:
            return self.v[int(m.group(1)) + half]
        m = re.fullmatch(r"v(\d+)", tok)
        if m:
            return self.v[int(m.group(1))]
        return np.full(64, np.float32(float(tok)), dtype=np.float32)

    def run(self, lines):
        f32 = np.float32
        for ln in lines:
            ln = ln.strip()
            if not ln or ln.startswith("s_"):
                continue
            op, rest = ln.split(None, 1)
            mods = {}
            for name in ("op_sel_hi", "op_sel", "neg_lo", "neg_hi"):
                m = re.search(name + r":\[([0-9,]+)\]", rest)
                if m:
                    mods[name] = [int(x) for x in m.group(1).split(",")]
                    rest = rest.replace(m.group(0), "")
            if op.startswith("v_pk_"):
                toks = [t.strip() for t in rest.split(",") if t.strip()]
                dst, srcs = toks[0], toks[1:]
                n = len(srcs)
                osl = mods.get("op_sel", [0] * n) + [0] * n
                osh = mods.get("op_sel_hi", [1] * n) + [1] * n
                nlo = mods.get("neg_lo", [0] * n) + [0] * n
                nhi = mods.get("neg_hi", [0] * n) + [0] * n
                lo = [self._src(srcs[i], osl[i]) * (f32(-1) if nlo[i] else f32(1)) for i in range(n)]
                hi = [self._src(srcs[i], osh[i]) * (f32(-1) if nhi[i] else f32(1)) for i in range(n)]
                if op == "v_pk_add_f32":
                    rl, rh = lo[0] + lo[1], hi[0] + hi[1]
                elif op == "v_pk_mul_f32":
                    rl, rh = lo[0] * lo[1], hi[0] * hi[1]
                elif op == "v_pk_fma_f32":
                    rl, rh = lo[0] * lo[1] + lo[2], hi[0] * hi[1] + hi[2]
                else:
                    raise ValueError(ln)
                d = int(re.fullmatch(r"v\[(\d+):(\d+)\]", dst).group(1))
                self.v[d], self.v[d + 1] = rl.astype(f32), rh.astype(f32)
            elif op == "v_exp_f32_e64":
                d, s = [t.strip() for t in rest.split(",")]
                neg = s.startswith("-")
                x = self.v[int(s.lstrip("-")[1:])]
                self.v[int(d[1:])] = np.exp2((-x if neg else x).astype(np.float64)).astype(f32)
            elif op == "v_fma_f32":
                d, a, b, c = [t.strip() for t in rest.split(",")]
                av = self.v[int(a.lstrip("-")[1:])] * (f32(-1) if a.startswith("-") else f32(1))
                self.v[int(d[1:])] = (av * self.v[int(b[1:])] + self.v[int(c[1:])]).astype(f32)
            elif op == "v_add_f32_e32":
                d, a, b = [t.strip() for t in rest.split(",")]
                self.v[int(d[1:])] = self.v[int(a[1:])] + self.v[int(b[1:])]
            elif op == "v_mov_b32_dpp":
                d, s = [t.strip() for t in rest.split()[0:2]]
                self.v[int(d.strip(",")[1:])] = np.roll(self.v[int(s.strip(",")[1:])], -1)       # lane l <- lane l + 1
            elif op == "v_mov_b32_e32":
                d, s = [t.strip() for t in rest.split(",")]
                self.v[int(d[1:])] = self.v[int(s[1:])] if s.startswith("v") else f32(float(s))
            elif op == "v_add_u32_e32":
                d, imm, s = [t.strip() for t in rest.split(",")]
                self.v[int(d[1:])] = (self.v[int(s[1:])].view(np.uint32) + np.uint32(int(imm))).view(f32)
            elif op == "ds_read_b128":
                m = re.fullmatch(r"v\[(\d+):(\d+)\],\s*v(\d+)\s+offset:(\d+)", rest.strip())
                d0, ptr, off = int(m.group(1)), int(m.group(3)), int(m.group(4))
                addr = (self.v[ptr].view(np.uint32).astype(np.int64) + off) // 4
                for k in range(4):
                    self.v[d0 + k] = self.lds[addr + k]
            else:
                raise ValueError("unknown instruction: " + ln)


def selftest():
    rng = np.random.default_rng(0)
    pro, body, epi = generate()
    z = rng.standard_normal((64, JT)).astype(np.float32) * 0.8
    Lc = rng.standard_normal((64, TT)).astype(np.float32)
    Rc = rng.standard_normal((64, TT)).astype(np.float32)
    lds = np.zeros(NREC * REC_FLOATS + 64, dtype=np.float32)
    for rec in range(NREC):
        c = rec % 64
        lds[rec * REC_FLOATS: rec * REC_FLOATS + JT] = z[c]
        lds[rec * REC_FLOATS + JT: rec * REC_FLOATS + JT + TT] = Lc[c]
        lds[rec * REC_FLOATS + JT + TT: rec * REC_FLOATS + JT + 2 * TT] = Rc[c]
    a = rng.standard_normal((2, 64, JT)).astype(np.float32) * 0.8
    li = rng.standard_normal((2, 64, TT)).astype(np.float32)
    ri = rng.standard_normal((2, 64, TT)).astype(np.float32)
    g0 = rng.standard_normal((2, 64, JT)).astype(np.float32)
    s0 = rng.standard_normal((2, 64)).astype(np.float32)
    m = Machine(lds)
    for j in range(JT):
        m.v[A0 + 2 * j], m.v[A0 + 2 * j + 1] = a[0, :, j], a[1, :, j]
        m.v[G0 + 2 * j], m.v[G0 + 2 * j + 1] = g0[0, :, j], g0[1, :, j]
    for t in range(TT):
        m.v[LR0 + 2 * t], m.v[LR0 + 2 * t + 1] = li[0, :, t], li[1, :, t]
        m.v[RR0 + 2 * t], m.v[RR0 + 2 * t + 1] = ri[0, :, t], ri[1, :, t]
    m.v[ACCS], m.v[ACCS + 1] = s0[0], s0[1]
    m.v[PTR] = (np.arange(64, dtype=np.uint32) * REC_BYTES).view(np.float32)
    m.run(pro)
    for _ in range(32):
        m.run(body)
    m.run(epi)
    # ---- direct formula (float64)
    G = g0.astype(np.float64).copy()
    Sacc = s0.astype(np.float64).copy()
    TG = np.zeros((64, JT + 1))
    for r in range(2):
        for l in range(64):
            Srow = li[r, l].astype(np.float64) @ Rc.astype(np.float64).T + ri[r, l].astype(np.float64) @ Lc.astype(np.float64).T   # (64,)
            d = a[r, l].astype(np.float64)[None, :] - z.astype(np.float64)                                                        # (64, JT)
            e = np.exp2(-d * d)
            G[r, l] += (Srow[:, None] * e * d).sum(0)
            Sacc[r, l] += (Srow * e.sum(1)).sum()
            TG[:, :JT] -= Srow[:, None] * e * d
            TG[:, JT] += Srow * e.sum(1)
    got_G = np.stack([[m.v[G0 + 2 * j + r] for j in range(JT)] for r in range(2)]).transpose(0, 2, 1)
    got_S = np.stack([m.v[ACCS], m.v[ACCS + 1]])
    got_T = np.stack([m.v[TG0 + q] for q in range(JT + 1)]).T
    errs = {"accG": np.abs(got_G - G).max() / np.abs(G).max(), "accS": np.abs(got_S - Sacc).max() / np.abs(Sacc).max(),
            "tg": np.abs(got_T - TG).max() / np.abs(TG).max()}
    print("selftest relative errors:", errs)
    assert max(errs.values()) < 5e-5, errs
    n_valu = sum(1 for ln in body if ln.startswith("v_") and not ln.startswith("v_add_u32"))
    print("loop body: %d lines, %d vector instructions per two steps" % (len(body), n_valu))


def main():
    if "--selftest" in sys.argv:
        selftest()
        return
    out = [a for a in sys.argv[1:] if not a.startswith("--")]
    out = out[0] if out else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                          "randomly-projected-additive-gps_amd", "csrc", "rpgp_bil_asm_loop.inc")
    pro, body, epi = generate()
    lines = ["s_waitcnt lgkmcnt(0)"] + pro + ["s_mov_b32 %[cnt], 32", "1:"] + body + \
            ["s_sub_u32 %[cnt], %[cnt], 1", "s_cmp_lg_u32 %[cnt], 0", "s_cbranch_scc1 1b"] + epi
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_bil_asm.py — do not edit; schedule and register map are described there.\n")
        f.write("#define RPGP_BIL_ASM_LOOP \\\n")
        for ln in lines:
            f.write('  "%s\\n" \\\n' % ln)
        f.write('  ""\n')
        clob = [PTR + 1, PTR + 2] + list(range(REC0, NVGPR_TOP))
        f.write("#define RPGP_BIL_ASM_CLOBBERS " + ", ".join('"v%d"' % c for c in clob) + ', "scc"\n')
    print("wrote %s: %d asm lines" % (out, len(lines)))


if __name__ == "__main__":
    main()
