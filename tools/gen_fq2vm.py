"""Generator of the Fq2 "virtual machine" programs behind the Groth16 verifier (libzkp_amd/csrc/fq2vm_programs.h).

The pairing check of one envelope is a fixed straight-line computation over Fq2 (the loop bits are constants of BN254), so it is written
here ONCE as formulas over symbolic Fq2 values (the same tower as bn254_pairing.h: Fq12 = Fq6[w]/(w^2 - v), Fq6 = Fq2[v]/(v^3 - xi)),
traced into operation lists, list-scheduled onto K cooperating wavefronts (rounds separated by workgroup barriers; a round gives every
wave operations of about one Fq2 product's cost), register-allocated onto an LDS-resident file of Fq2 registers, and emitted as tables
of 32-bit micro-operations.  The device side (fq2vm.h) is a small interpreter whose whole code fits the instruction cache; the old
lane-per-chain kernels streamed ~500 KB of straight-line code per Miller iteration.

  python tools/gen_fq2vm.py --check     numeric self-test of the formulas, the schedules and the register allocation (pure Python)
  python tools/gen_fq2vm.py             rewrite libzkp_amd/csrc/fq2vm_programs.h

Reference for what is computed: ark-groth16's verifier equation under /root/reference/src/backend/snark.rs:377-401,455-495 (pairing
product == 1); the pairing here is the optimal ate pairing with loop count 6x + 2 followed by the x-chain final exponentiation of
bn254_pairing.h (a fixed power of ark's pairing: the product check holds under one iff under the other)."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))

P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
X = 4965661367192848881
ATE = 6 * X + 2

# ---------------------------------------------------------------- opcodes (fq2vm.h holds the same list)
NOP, MUL, SQ, ADD, SUB, MULXI, CONJ, MUL0, MUL1, INV, LDG, STG, LDC, MOV, NEG, LDK, STC, T3M, T3P, END = range(20)
OPNAME = ["NOP", "MUL", "SQ", "ADD", "SUB", "MULXI", "CONJ", "MUL0", "MUL1", "INV", "LDG", "STG", "LDC", "MOV", "NEG", "LDK", "STC", "T3M", "T3P", "END"]
COST = {NOP: 0, MUL: 20, SQ: 15, ADD: 4, SUB: 4, MULXI: 7, CONJ: 3, MUL0: 13, MUL1: 13, INV: 700, LDG: 3, STG: 3, LDC: 3, MOV: 2, NEG: 3, LDK: 3, STC: 3, T3M: 5, T3P: 5}      # ~50 instructions each
BAR = 0x80


# ---------------------------------------------------------------- plain Fq2 arithmetic (the evaluator of traced programs)
def f2mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
def f2add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def f2sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
def f2xi(a): return ((9 * a[0] - a[1]) % P, (a[0] + 9 * a[1]) % P)
def f2inv(a):
    d = pow(a[0] * a[0] + a[1] * a[1], P - 2, P)
    return (a[0] * d % P, -a[1] * d % P)
def f2pow(a, e):
    r = (1, 0)
    while e:
        if e & 1: r = f2mul(r, a)
        a = f2mul(a, a); e >>= 1
    return r


def apply_op(op, a, b):
    if op == MUL: return f2mul(a, b)
    if op == SQ: return f2mul(a, a)
    if op == ADD: return f2add(a, b)
    if op == SUB: return f2sub(a, b)
    if op == MULXI: return f2xi(a)
    if op == CONJ: return (a[0], -a[1] % P)
    if op == MUL0: return (a[0] * b[0] % P, a[1] * b[0] % P)
    if op == MUL1: return (a[0] * b[1] % P, a[1] * b[1] % P)
    if op == INV: return f2inv(a)
    if op == MOV: return a
    if op == NEG: return (-a[0] % P, -a[1] % P)
    if op == T3M: return ((3 * a[0] - 2 * b[0]) % P, (3 * a[1] - 2 * b[1]) % P)
    if op == T3P: return ((3 * a[0] + 2 * b[0]) % P, (3 * a[1] + 2 * b[1]) % P)
    raise ValueError(op)


# ---------------------------------------------------------------- constants of the tower
XI = (9, 1)
def gamma(j, k): return f2pow(XI, k * (P**j - 1) // 6)
CONSTS = []            # list of fq2 values; LDC's operand is the index
CONST_IX = {}
def const_ix(v):
    v = (v[0] % P, v[1] % P)
    if v not in CONST_IX:
        CONST_IX[v] = len(CONSTS); CONSTS.append(v)
    return CONST_IX[v]


# ---------------------------------------------------------------- tracing
class Prog:
    """One program: operations in SSA form over virtual registers.  Pinned virtual registers are the program's interface: `ins`
    (name -> vreg, live on entry) and `outs` (name -> vreg, must be in that pinned register on exit)."""
    def __init__(self, name):
        self.name, self.ops, self.nv, self.ins, self.outs, self.dead = name, [], 0, {}, {}, set()      # dead: pinned names whose content nobody needs after this program
    def new(self):
        self.nv += 1; return self.nv - 1
    def emit(self, op, a=None, b=None, imm=None):
        d = self.new() if op not in (STG, STC) else None
        self.ops.append((op, d, a, b, imm)); return d


class V:
    """symbolic Fq2 value"""
    __slots__ = ("p", "r")
    def __init__(self, p, r): self.p, self.r = p, r
    def _bin(self, op, o): return V(self.p, self.p.emit(op, self.r, o.r))
    def _un(self, op): return V(self.p, self.p.emit(op, self.r))
    def __mul__(self, o): return self._bin(MUL, o) if o is not self else self._un(SQ)
    def __add__(self, o): return self._bin(ADD, o)
    def __sub__(self, o): return self._bin(SUB, o)
    def __neg__(self): return self._un(NEG)
    def sq(self): return self._un(SQ)
    def xi(self): return self._un(MULXI)
    def conj(self): return self._un(CONJ)
    def dbl(self): return self._bin(ADD, self)
    def mul0(self, o): return self._bin(MUL0, o)      # self * (c0 of o), an Fq scalar
    def mul1(self, o): return self._bin(MUL1, o)      # self * (c1 of o)
    def inv(self): return self._un(INV)
    def t3m(self, o): return self._bin(T3M, o)        # 3 self - 2 o (the cyclotomic squaring's output step as one operation)
    def t3p(self, o): return self._bin(T3P, o)        # 3 self + 2 o


def inp(p, name):
    v = p.new(); p.ins[name] = v; return V(p, v)
def const(p, val): return V(p, p.emit(LDC, imm=const_ix(val)))
def ldg(p, slot): return V(p, p.emit(LDG, imm=slot))
def stg(p, v, slot): p.emit(STG, v.r, imm=slot)
def out(p, name, v): p.outs[name] = v.r


# ---------------------------------------------------------------- tower formulas over V (mirrors bn254_pairing.h)
def f6_add(a, b): return tuple(x + y for x, y in zip(a, b))
def f6_sub(a, b): return tuple(x - y for x, y in zip(a, b))
def f6_neg(a): return tuple(-x for x in a)
def f6_mulv(a): return (a[2].xi(), a[0], a[1])
def f6_mul(a, b):
    v0, v1, v2 = a[0] * b[0], a[1] * b[1], a[2] * b[2]
    t0 = (a[1] + a[2]) * (b[1] + b[2]) - v1 - v2
    t1 = (a[0] + a[1]) * (b[0] + b[1]) - v0 - v1
    t2 = (a[0] + a[2]) * (b[0] + b[2]) - v0 - v2
    return (v0 + t0.xi(), t1 + v2.xi(), t2 + v1)
def f6_sq(a):        # CH-SQR2: five products instead of six
    s0 = a[0].sq(); ab = a[0] * a[1]; s1 = ab.dbl(); s2 = (a[0] - a[1] + a[2]).sq(); bc = a[1] * a[2]; s3 = bc.dbl(); s4 = a[2].sq()
    return (s0 + s3.xi(), s1 + s4.xi(), s1 + s2 + s3 - s0 - s4)
def f12_mul(a, b):
    t0, t1 = f6_mul(a[0], b[0]), f6_mul(a[1], b[1])
    m = f6_mul(f6_add(a[0], a[1]), f6_add(b[0], b[1]))
    return (f6_add(t0, f6_mulv(t1)), f6_sub(f6_sub(m, t0), t1))
def f12_sq(a):
    t = f6_mul(a[0], a[1])
    s = f6_mul(f6_add(a[0], a[1]), f6_add(a[0], f6_mulv(a[1])))
    return (f6_sub(f6_sub(s, t), f6_mulv(t)), f6_add(t, t))
def f12_conj(a): return (a[0], f6_neg(a[1]))
def f6_mul_f2(a, k): return tuple(x * k for x in a)
def f6_sparse2(a, b0, b1):
    v0, v1 = a[0] * b0, a[1] * b1
    mid = (a[0] + a[1]) * (b0 + b1) - v0 - v1
    return (v0 + (a[2] * b1).xi(), mid, v1 + a[2] * b0)
def f12_mul_line(f, l):
    A, B, C = l
    t0, t1 = f6_mul_f2(f[0], A), f6_sparse2(f[1], B, C)
    m = f6_sparse2(f6_add(f[0], f[1]), A + B, C)
    return (f6_add(t0, f6_mulv(t1)), f6_sub(f6_sub(m, t0), t1))
def line_mul(l1, l2):        # (A1 + (B1 + C1 v) w)(A2 + (B2 + C2 v) w): six products
    (A1, B1, C1), (A2, B2, C2) = l1, l2
    aa, bb, cc = A1 * A2, B1 * B2, C1 * C2
    bc = (B1 + C1) * (B2 + C2) - bb - cc
    ab = (A1 + B1) * (A2 + B2) - aa - bb
    ac = (A1 + C1) * (A2 + C2) - aa - cc
    return ((aa + cc.xi(), bb, bc), (ab, ac, None))
def f12_mul_ll(f, ll):       # f * (product of two lines): c1.a2 of the second factor is zero
    (c0, c1), (d0, (e0, e1, _)) = f, ll
    t0 = f6_mul(c0, d0)
    t1 = f6_sparse2(c1, e0, e1)
    m = f6_mul(f6_add(c0, c1), (d0[0] + e0, d0[1] + e1, d0[2]))
    return (f6_add(t0, f6_mulv(t1)), f6_sub(f6_sub(m, t0), t1))
def f6_inv(a):
    A = a[0].sq() - (a[1] * a[2]).xi()
    B = a[2].sq().xi() - a[0] * a[1]
    C = a[1].sq() - a[0] * a[2]
    F = a[0] * A + (a[2] * B + a[1] * C).xi()
    return f6_mul_f2((A, B, C), F.inv())
def f12_inv(a):
    t = f6_inv(f6_sub(f6_mul(a[0], a[0]), f6_mulv(f6_mul(a[1], a[1]))))
    return (f6_mul(a[0], t), f6_neg(f6_mul(a[1], t)))
def coeffs(a): return [a[0][0], a[1][0], a[0][1], a[1][1], a[0][2], a[1][2]]          # a_k of w^k, k = 0..5
def from_coeffs(c): return ((c[0], c[2], c[4]), (c[1], c[3], c[5]))
def f12_frob(p, a, j):
    c = coeffs(a)
    o = []
    for k in range(6):
        x = c[k].conj() if j & 1 else c[k]
        o.append(x if k == 0 else x * const(p, gamma(j, k)))
    return from_coeffs(o)
def f4_sq(x, y):
    xx, yy = x.sq(), y.sq()
    return xx + yy.xi(), (x + y).sq() - xx - yy
def f12_cyclo_sq(f):
    (a0, a1, a2), (b0, b1, b2) = f
    t0, t1 = f4_sq(a0, b1); t2, t3 = f4_sq(b0, a2); t4, t5 = f4_sq(a1, b2)
    m = lambda t, z: t.t3m(z)
    q = lambda t, z: t.t3p(z)
    return ((m(t0, a0), m(t2, a1), m(t4, a2)), (q(t5.xi(), b0), q(t1, b1), q(t3, b2)))


# G2 steps (Jacobian over Fq2, a = 0).  Pp packs the G1 point: c0 = xp, c1 = yp.
def step_double(T, Pp):
    Xj, Yj, Zj = T
    XX, YY, ZZ, YZ = Xj.sq(), Yj.sq(), Zj.sq(), Yj * Zj
    E = XX.dbl() + XX
    raw = ((YZ * ZZ).dbl(), -(E * ZZ), E * Xj - YY.dbl())          # the line is (raw0 yp, raw1 xp, raw2)
    C4 = YY.sq()
    D = ((Xj + YY).sq() - XX - C4).dbl()
    X3 = E.sq() - D.dbl()
    C8 = C4.dbl().dbl().dbl()
    Y3 = E * (D - X3) - C8
    return (X3, Y3, YZ.dbl()), (at_point(raw, Pp) if Pp is not None else raw)
def at_point(raw, Pp): return (raw[0].mul1(Pp), raw[1].mul0(Pp), raw[2])
def step_add(T, Q, Pp, want_line=True):
    Xj, Yj, Zj = T
    ZZ = Zj.sq()
    U2, S2 = Q[0] * ZZ, Q[1] * (Zj * ZZ)
    H, rr = U2 - Xj, S2 - Yj
    Z3 = Zj * H
    HH = H.sq(); HHH = H * HH; Vv = Xj * HH
    X3 = rr.sq() - HHH - Vv.dbl()
    Y3 = rr * (Vv - X3) - Yj * HHH
    line = None
    if want_line:
        raw = (Z3, -rr, rr * Q[0] - Z3 * Q[1])
        line = at_point(raw, Pp) if Pp is not None else raw
    return (X3, Y3, Z3), line, (H, rr)
def g2_double(T):
    Xj, Yj, Zj = T
    XX, YY = Xj.sq(), Yj.sq()
    E = XX.dbl() + XX
    C4 = YY.sq()
    D = ((Xj + YY).sq() - XX - C4).dbl()
    X3 = E.sq() - D.dbl()
    return (X3, E * (D - X3) - C4.dbl().dbl().dbl(), (Yj * Zj).dbl())


# ---------------------------------------------------------------- the programs
# global-memory slots of LDG / STG (fq2vm.h: the launcher maps slot -> address)
# One buffer per batch, [slot][20 words][n envelopes]: pair j of an envelope owns slots 12 j .. 12 j + 11 (a Miller chain is launched on its
# pair's slice and numbers slots from 0); the subgroup chain works on pair 0's slice; the finish chain sees the whole buffer.
SLOT_QX, SLOT_QY, SLOT_P, SLOT_F0 = 0, 1, 2, 3            # miller: inputs Q.x, Q.y, (xp, yp); output f = slots 3..8
SLOT_SZ, SLOT_SH, SLOT_SR = 9, 10, 11                     # subgroup: outputs Z of [6x^2] B and the two differences to psi(B)
PAIR_SLOTS, SLOT_RES = 12, 36                             # finish: input coefficient k of pair j = slot 12 j + 3 + k; output 36..41
SLOT_SAVE_R, SLOT_SAVE_Y1, SLOT_SAVE_Y3, SLOT_SAVE_Y4 = 42, 48, 54, 60      # finish: values parked between the three power loops
N_SLOTS = 66
LINE_SLOT0 = 2            # chain L (one point of the key, n = 1): inputs Q.x, Q.y in slots 0, 1; the line table follows
def SLOT_FIN(j, k): return PAIR_SLOTS * j + SLOT_F0 + k
F_NAMES = ["f%d" % i for i in range(6)]
def in12(p, names): return from_coeffs([inp(p, n) for n in names])
def out12(p, names, a):
    for n, v in zip(names, coeffs(a)): out(p, n, v)
def inT(p): return (inp(p, "TX"), inp(p, "TY"), inp(p, "TZ"))
def outT(p, T):
    out(p, "TX", T[0]); out(p, "TY", T[1]); out(p, "TZ", T[2])


def build_programs():
    progs = {}
    def prog(name):
        p = Prog(name); progs[name] = p; return p
    # ---- Miller chain of one (Q, P) pair
    p = prog("M_INIT")
    qx, qy, pp = ldg(p, SLOT_QX), ldg(p, SLOT_QY), ldg(p, SLOT_P)
    out(p, "QX", qx); out(p, "QY", qy); out(p, "PP", pp)
    one, zero = const(p, (1, 0)), const(p, (0, 0))
    out(p, "TX", V(p, p.emit(MOV, qx.r))); out(p, "TY", V(p, p.emit(MOV, qy.r))); out(p, "TZ", one)
    out12(p, F_NAMES, from_coeffs([V(p, p.emit(MOV, one.r))] + [V(p, p.emit(MOV, zero.r)) for _ in range(5)]))
    p = prog("M_DBL")
    f, T, pp = in12(p, F_NAMES), inT(p), inp(p, "PP")
    T2, line = step_double(T, pp)
    out12(p, F_NAMES, f12_mul_line(f12_sq(f), line)); outT(p, T2)
    for sign in (+1, -1):
        p = prog("M_ADD" if sign > 0 else "M_SUB")
        f, T, pp, qx, qy = in12(p, F_NAMES), inT(p), inp(p, "PP"), inp(p, "QX"), inp(p, "QY")
        T2, line, _ = step_add(T, (qx, qy if sign > 0 else -qy), pp)
        out12(p, F_NAMES, f12_mul_line(f, line)); outT(p, T2)
    p = prog("M_FROB")
    f, T, pp, qx, qy = in12(p, F_NAMES), inT(p), inp(p, "PP"), inp(p, "QX"), inp(p, "QY")
    q1 = (qx.conj() * const(p, gamma(1, 2)), qy.conj() * const(p, gamma(1, 3)))
    T2, line, _ = step_add(T, q1, pp)
    f = f12_mul_line(f, line)
    g23 = gamma(2, 3); assert g23 == (P - 1, 0)
    nq2 = (qx * const(p, gamma(2, 2)), V(p, p.emit(MOV, qy.r)))          # -pi^2(Q) = (x gamma_22, -y gamma_23) = (x gamma_22, y)
    _, line, _ = step_add(T2, nq2, pp)
    f = f12_mul_line(f, line)
    for k, v in enumerate(coeffs(f)): stg(p, v, SLOT_F0 + k)
    # ---- chain B: the two pairs whose G2 point belongs to the key (gamma, delta) in one chain, on line coefficients computed when the
    # key is loaded (chain L below): no point arithmetic, one squaring of f for both.  The table is read through the cursor (LDK).
    BP = F_NAMES + ["P1", "P2"]
    # the public-input point of pair 1 comes in Jacobian form (no inversion in the parsing kernel): P1 = (X Z, -Y) and P1Z = (Z^3, .) evaluate its
    # line scaled by Z^3, a factor in Fq that the final exponentiation removes (g16_verify.h g16_vm_pair1)
    def table_lines(p, p1, p1z, p2, base):
        r1 = tuple(V(p, p.emit(LDK, imm=base + k)) for k in range(3))
        return (r1[0].mul1(p1), r1[1].mul0(p1), r1[2].mul0(p1z)), at_point(tuple(V(p, p.emit(LDK, imm=base + 3 + k)) for k in range(3)), p2)
    p = prog("B_INIT")
    out(p, "P1", ldg(p, SLOT_P)); out(p, "P1Z", ldg(p, SLOT_QX)); out(p, "P2", ldg(p, PAIR_SLOTS + SLOT_P))
    one, zero = const(p, (1, 0)), const(p, (0, 0))
    out12(p, F_NAMES, from_coeffs([one] + [V(p, p.emit(MOV, zero.r)) for _ in range(5)]))
    p = prog("B_DBL")
    f, p1, p1z, p2 = in12(p, F_NAMES), inp(p, "P1"), inp(p, "P1Z"), inp(p, "P2")
    l1, l2 = table_lines(p, p1, p1z, p2, 0)
    out12(p, F_NAMES, f12_mul_ll(f12_sq(f), line_mul(l1, l2)))
    p = prog("B_ADD")
    f, p1, p1z, p2 = in12(p, F_NAMES), inp(p, "P1"), inp(p, "P1Z"), inp(p, "P2")
    l1, l2 = table_lines(p, p1, p1z, p2, 0)
    out12(p, F_NAMES, f12_mul_ll(f, line_mul(l1, l2)))
    p = prog("B_FROB")
    f, p1, p1z, p2 = in12(p, F_NAMES), inp(p, "P1"), inp(p, "P1Z"), inp(p, "P2")
    l1, l2 = table_lines(p, p1, p1z, p2, 0); l3, l4 = table_lines(p, p1, p1z, p2, 6)
    f = f12_mul_ll(f12_mul_ll(f, line_mul(l1, l2)), line_mul(l3, l4))
    for k, v in enumerate(coeffs(f)): stg(p, v, SLOT_F0 + k)
    # ---- chain L (once per key and G2 point, on the host): the same steps as chain A without the G1 point, storing the raw line
    # coefficients at the cursor (STC)
    def store_line(p, raw, base):
        for k in range(3): p.emit(STC, raw[k].r, imm=LINE_SLOT0 + base + k)
    p = prog("L_INIT")
    qx, qy = ldg(p, SLOT_QX), ldg(p, SLOT_QY)
    out(p, "QX", qx); out(p, "QY", qy)
    out(p, "TX", V(p, p.emit(MOV, qx.r))); out(p, "TY", V(p, p.emit(MOV, qy.r))); out(p, "TZ", const(p, (1, 0)))
    p = prog("L_DBL")
    T2, raw = step_double(inT(p), None); store_line(p, raw, 0); outT(p, T2)
    for sign in (+1, -1):
        p = prog("L_ADD" if sign > 0 else "L_SUB")
        T, qx, qy = inT(p), inp(p, "QX"), inp(p, "QY")
        T2, raw, _ = step_add(T, (qx, qy if sign > 0 else -qy), None); store_line(p, raw, 0); outT(p, T2)
    p = prog("L_FROB")
    T, qx, qy = inT(p), inp(p, "QX"), inp(p, "QY")
    q1 = (qx.conj() * const(p, gamma(1, 2)), qy.conj() * const(p, gamma(1, 3)))
    T2, raw, _ = step_add(T, q1, None); store_line(p, raw, 0)
    nq2 = (qx * const(p, gamma(2, 2)), V(p, p.emit(MOV, qy.r)))
    _, raw, _ = step_add(T2, nq2, None); store_line(p, raw, 6)
    # ---- subgroup chain: r * Q
    p = prog("S_INIT")
    qx, qy = ldg(p, SLOT_QX), ldg(p, SLOT_QY)
    out(p, "QX", qx); out(p, "QY", qy)
    out(p, "TX", V(p, p.emit(MOV, qx.r))); out(p, "TY", V(p, p.emit(MOV, qy.r))); out(p, "TZ", const(p, (1, 0)))
    p = prog("S_DBL"); outT(p, g2_double(inT(p)))
    for sign in (+1, -1):
        p = prog("S_ADD" if sign > 0 else "S_SUB")
        T, qx, qy = inT(p), inp(p, "QX"), inp(p, "QY")
        T2, _, _ = step_add(T, (qx, qy if sign > 0 else -qy), None, want_line=False); outT(p, T2)
    # B is in G2 iff psi(B) = [6x^2] B (psi = untwist-Frobenius-twist; the check ark-bn254 makes, eprint 2022/352 section 4.3).  Exact for
    # every point of the twist: psi^2 - t psi + p = 0 there, so psi(B) = [t - 1] B gives [(t-1)^2 - t(t-1) + p] B = [p + 1 - t] B = [r] B = 0,
    # and the r-torsion of E'(Fp2) is G2 (the cofactor 2p - r is prime to r).  The incomplete formulas meet a degenerate case only on a
    # point of order < r, where Z = 0 sticks and the verdict is a (correct) rejection.  T = [6x^2] B Jacobian: Z != 0, X = psi_x Z^2, Y = psi_y Z^3.
    p = prog("S_LAST")
    T, qx, qy = inT(p), inp(p, "QX"), inp(p, "QY")
    ZZ = T[2].sq()
    px, py = qx.conj() * const(p, gamma(1, 2)), qy.conj() * const(p, gamma(1, 3))
    stg(p, T[2], SLOT_SZ); stg(p, px * ZZ - T[0], SLOT_SH); stg(p, py * (T[2] * ZZ) - T[1], SLOT_SR)
    # ---- final exponentiation chain of one envelope
    A, Bn = ["a%d" % i for i in range(6)], ["b%d" % i for i in range(6)]
    def save12(p, slot0, a):
        for k, v in enumerate(coeffs(a)): stg(p, v, slot0 + k)
    def load12(p, slot0): return from_coeffs([ldg(p, slot0 + k) for k in range(6)])
    def dup(p, a): return from_coeffs([V(p, p.emit(MOV, v.r)) for v in coeffs(a)])
    p = prog("F_INIT")          # product of the three Miller values and the key's constant, then the easy part; r -> acc, base, memory
    ld12 = lambda j: from_coeffs([ldg(p, SLOT_FIN(j, k)) for k in range(6)])
    ldk12 = from_coeffs([V(p, p.emit(LDK, imm=k)) for k in range(6)])          # the key's constant factor: Miller value of (beta, -alpha)
    f = f12_mul(f12_mul(ld12(0), ld12(1)), ldk12)          # chain A's value (pair 0), chain B's (pairs 1 and 2), the key's
    e1 = f12_mul(f12_conj(f), f12_inv(f))
    r = f12_mul(f12_frob(p, e1, 2), e1)
    out12(p, A, r); out12(p, Bn, dup(p, r)); save12(p, SLOT_SAVE_R, r)
    p = prog("F_CSQ"); out12(p, A, f12_cyclo_sq(in12(p, A)))
    p = prog("F_MULB")
    a, b = in12(p, A), in12(p, Bn); out12(p, A, f12_mul(a, b))
    p = prog("F_MULC")          # acc * base^-1: in the cyclotomic subgroup the inverse is the conjugate (the powers walk x in signed digits)
    a, b = in12(p, A), in12(p, Bn); out12(p, A, f12_mul(a, f12_conj(b)))
    # the values the chain comes back to (r, y1, y3, y4) wait in the slot buffer, not in registers: the power loops keep the whole
    # register file for the products of one Fq12 multiplication
    p = prog("F_G1")            # acc = r^x: y0 = conj, y1 = y0^2, y3 = y1^2 y1; next power on y3
    y0 = f12_conj(in12(p, A)); y1 = f12_cyclo_sq(y0); y3 = f12_mul(f12_cyclo_sq(y1), y1)
    save12(p, SLOT_SAVE_Y1, y1); save12(p, SLOT_SAVE_Y3, y3); out12(p, A, y3); out12(p, Bn, dup(p, y3))
    p.dead = set(Bn)
    p = prog("F_G2")            # acc = y3^x: y4 = conj; next power on y4^2
    y4 = f12_conj(in12(p, A)); y5 = f12_cyclo_sq(y4)
    save12(p, SLOT_SAVE_Y4, y4); out12(p, A, y5); out12(p, Bn, dup(p, y5))
    p.dead = set(Bn)
    p = prog("F_G3")            # acc = y6
    y6 = in12(p, A)
    y4 = load12(p, SLOT_SAVE_Y4)
    y64 = f12_mul(y6, y4)
    y8 = f12_mul(y64, f12_conj(load12(p, SLOT_SAVE_Y3)))
    y9 = f12_mul(y8, load12(p, SLOT_SAVE_Y1))
    r = load12(p, SLOT_SAVE_R)
    y11 = f12_mul(f12_mul(y8, y4), r)
    y13 = f12_mul(f12_frob(p, y9, 1), y11)
    y14 = f12_mul(f12_frob(p, y8, 2), y13)
    res = f12_mul(f12_frob(p, f12_mul(f12_conj(r), y9), 3), y14)
    for k, v in enumerate(coeffs(res)): stg(p, v, SLOT_RES + k)
    p.dead = set(A + Bn)
    return progs


def naf(n):
    o = []
    while n:
        if n & 1:
            d = 2 - (n & 3); n -= d
        else: d = 0
        o.append(d); n >>= 1
    return o
def scripts():
    d = naf(ATE)
    assert d[-1] == 1
    m = ["M_INIT"]
    for i in range(len(d) - 2, -1, -1):
        m.append("M_DBL")
        if d[i]: m.append("M_ADD" if d[i] > 0 else "M_SUB")
    m.append("M_FROB")
    bb, ll = ["B_INIT"], ["L_INIT"]          # (program, cursor advance after it): six table slots per step
    for i in range(len(d) - 2, -1, -1):
        bb.append(("B_DBL", 6)); ll.append(("L_DBL", 6))
        if d[i]: bb.append(("B_ADD", 6)); ll.append(("L_ADD" if d[i] > 0 else "L_SUB", 6))
    bb.append("B_FROB"); ll.append("L_FROB")
    s = ["S_INIT"]
    ds = naf(6 * X * X)
    assert ds[-1] == 1
    for i in range(len(ds) - 2, -1, -1):
        s.append("S_DBL")
        if ds[i]: s.append("S_ADD" if ds[i] > 0 else "S_SUB")
    s.append("S_LAST")
    dx = naf(X)
    assert dx[-1] == 1
    def powx():
        o = []
        for i in range(len(dx) - 2, -1, -1):
            o.append("F_CSQ")
            if dx[i]: o.append("F_MULB" if dx[i] > 0 else "F_MULC")
        return o
    f = ["F_INIT"] + powx() + ["F_G1"] + powx() + ["F_G2"] + powx() + ["F_G3"]
    return {"miller": m, "subgroup": s, "finish": f, "miller_b": bb, "lines": ll}
N_LINE_SLOTS = 6 * (sum(1 for e in scripts()["lines"] if isinstance(e, tuple)) + 2)


# ---------------------------------------------------------------- reference evaluation of the traced (unscheduled) programs
def run_script(progs, script, gmem, state=None):
    """gmem: slot -> fq2 (inputs read by LDG, outputs written by STG / STC; ("k", i) = the launch's constant table read by LDK).  Pinned names
    carry state from program to program; script entries are names or (name, cursor advance)."""
    state = dict(state or {})
    cur = 0
    for ent in script:
        name, adv = ent if isinstance(ent, tuple) else (ent, 0)
        p = progs[name]
        val = {}
        for n, v in p.ins.items(): val[v] = state[n]
        for op, d, a, b, imm in p.ops:
            if op == LDG: val[d] = gmem[imm]
            elif op == LDC: val[d] = CONSTS[imm]
            elif op == LDK: val[d] = gmem[("k", cur + imm)]
            elif op == STG: gmem[imm] = val[a]
            elif op == STC: gmem[cur + imm] = val[a]
            else: val[d] = apply_op(op, val[a], val[b] if b is not None else None)
        for n, v in p.outs.items(): state[n] = val[v]
        cur += adv
    return state


# ---------------------------------------------------------------- scheduling onto K waves and register allocation
class Sched:
    """rounds[r][w] = list of op indices that wave w executes in round r (in order); a barrier follows every round."""
    pass


def schedule(p, K, window, noise=None, budget0=None, slots=2):
    """rounds[r][w] = list of STEPS of wave w in round r; a step is up to `slots` operations of the same opcode (the parts of the wave:
    with two slots lanes 0-31 and 32-63 work for the same 32 envelopes on different operands; with four, 16 lanes each for 16 envelopes)."""
    n = len(p.ops)
    defs = {}
    for i, (op, d, a, b, imm) in enumerate(p.ops):
        if d is not None: defs[d] = i
    preds = [[defs[x] for x in (a, b) if x is not None and x in defs] for (op, d, a, b, imm) in p.ops]
    succs = [[] for _ in range(n)]
    for i, ps in enumerate(preds):
        for j in ps: succs[j].append(i)
    cost = [COST[o[0]] for o in p.ops]
    prio = [0] * n
    for i in range(n - 1, -1, -1): prio[i] = cost[i] + max([prio[j] for j in succs[i]], default=0)
    if noise is not None: prio = [x * (1 + noise[1] * noise[0].random()) for x in prio]          # (rng, scale): the search of compile_all
    done_round = [None] * n
    unsched = list(range(n))
    rounds = []
    while unsched:
        r = len(rounds)
        budget = budget0 or COST[MUL]
        loads = [0] * K
        lanes = [[] for _ in range(K)]
        placed = {}          # op -> wave, this round
        cand = unsched[:window]
        progress = True
        while progress:
            progress = False
            ready = []
            for i in cand:
                if i in placed: continue
                ok, same = True, None
                for j in preds[i]:
                    if done_round[j] is not None and done_round[j] < r: continue
                    if j in placed:
                        if same is None or same == placed[j]: same = placed[j]
                        else: ok = False
                    else: ok = False
                if ok: ready.append((-prio[i], i, same))
            ready.sort()
            for _, i, same in ready:
                fits = lambda q: loads[q] == 0 or loads[q] + cost[i] <= budget
                if same is not None:
                    if not fits(same): continue
                    w = same
                else:
                    ws = [q for q in range(K) if fits(q)]
                    if not ws: continue
                    w = min(ws, key=lambda q: loads[q])
                step = [i]
                for _, j, same2 in ready:
                    if len(step) == slots: break
                    if j != i and p.ops[j][0] == p.ops[i][0] and (same2 is None or same2 == w): step.append(j)
                lanes[w].append(step); loads[w] += cost[i]; progress = True
                for q in step: placed[q] = w
                budget = max(budget, loads[w])
                break
        assert placed, "scheduler stuck in " + p.name
        for i in placed: done_round[i] = r
        unsched = [i for i in unsched if i not in placed]
        rounds.append(lanes)
    return rounds, done_round




def allocate(p, rounds, done_round, pins, nreg, slots=2):
    """pins: name -> physical register.  Returns (streams, nused): streams[w] = list of [op, [(d, a, b), ...], bar] with physical registers
    (one triple per occupied slot of the step; b = the immediate of LDG / STG / LDC / LDK)."""
    K = len(rounds[0])
    last_use = {}
    for i, (op, d, a, b, imm) in enumerate(p.ops):
        for x in (a, b):
            if x is not None: last_use[x] = max(last_use.get(x, -1), done_round[i])
    for i, (op, d, a, b, imm) in enumerate(p.ops):
        if d is not None and d not in last_use: last_use[d] = done_round[i]
    nrounds = len(rounds)
    out_of = {}                                          # vreg -> [pin names] it must end in
    for nme, v in p.outs.items(): out_of.setdefault(v, []).append(nme)
    for v in out_of: last_use[v] = nrounds               # live to the end (copied in the fix-up rounds if not already in its pin)
    phys = {}
    reusable = {pins[nme] for nme in p.dead if nme not in p.outs}          # pins this program may use as temporaries
    live_in_pins = {pins[nme] for nme in p.ins}
    free = [r for r in range(nreg - 1, -1, -1) if r not in pins.values() or (r in reusable and r not in live_in_pins)]
    nfree0 = len(free)
    pin_busy = {}                                        # physical pin -> vreg currently holding it (live-in values)
    for nme, v in p.ins.items():
        phys[v] = pins[nme]; pin_busy[pins[nme]] = v
    peak = 0
    for r, lanes in enumerate(rounds):
        for w in range(K):
            for step in lanes[w]:
                for i in step:
                    op, d, a, b, imm = p.ops[i]
                    if d is None: continue
                    got = None
                    for nme in out_of.get(d, []):
                        pr = pins[nme]
                        holder = pin_busy.get(pr)
                        if holder is None or (holder != d and last_use.get(holder, -1) < r and holder not in out_of):
                            got = pr; pin_busy[pr] = d; break
                    if got is None:
                        if not free: return None, None
                        got = free.pop()
                    phys[d] = got
        peak = max(peak, nfree0 - len(free))
        # free what died in this round: temporaries, and live-in values sitting in pins that are dead on exit
        for v, lu in list(last_use.items()):
            if lu == r and v in phys and v not in out_of and (phys[v] not in pins.values() or phys[v] in reusable):
                free.append(phys[v]); del last_use[v]
    def triple(i):
        op, d, a, b, imm = p.ops[i]
        return (phys[d] if d is not None else 0, phys[a] if a is not None else 0, phys[b] if b is not None else (imm if imm is not None else 0))
    streams = [[] for _ in range(K)]
    for r, lanes in enumerate(rounds):
        for w in range(K):
            for step in lanes[w]:
                streams[w].append([p.ops[step[0]][0], [triple(i) for i in step], 0])
            if not lanes[w]: streams[w].append([NOP, [(0, 0, 0)], 0])
            streams[w][-1][2] = 1
    # fix-up: outputs that are not yet in their pins
    moves = []
    for v, names in out_of.items():
        for nme in names:
            if phys[v] != pins[nme]: moves.append((pins[nme], phys[v]))
    per = slots
    while moves:                                         # a copy may only overwrite a register that no remaining copy still reads
        srcs = {s for d, s in moves}
        ready = [m for m in moves if m[0] not in srcs]
        assert ready, "copy cycle in " + p.name
        ready = ready[:K * per]
        for w in range(K):
            mine = ready[w * per:(w + 1) * per]
            if mine: streams[w].append([MOV, [(m[0], m[1], 0) for m in mine], 1])
            else: streams[w].append([NOP, [(0, 0, 0)], 1])
        moves = [m for m in moves if m not in ready]
    return streams, (len(pins) - len([1 for q in reusable if q not in live_in_pins])) + peak


def simulate(streams, regs, gmem, cur=0):
    """Executes scheduled streams round by round on a physical register file; checks that no wave reads or overwrites, inside a round,
    a register that another wave writes in the same round, and that the two halves of a step do not touch each other's result."""
    K = len(streams)
    pc = [0] * K
    two = (MUL, ADD, SUB, MUL0, MUL1, T3M, T3P)
    while pc[0] < len(streams[0]):
        writes, reads = [set() for _ in range(K)], [set() for _ in range(K)]
        new = [dict() for _ in range(K)]
        for w in range(K):
            local = {}
            while True:
                op, hs, bar = streams[w][pc[w]]; pc[w] += 1
                get = lambda x: local[x] if x in local else regs[x]
                results = []
                for h in hs:
                    if op == NOP: continue
                    d, a, b = h
                    if op == LDG: results.append((d, gmem[b]))
                    elif op == LDC: results.append((d, CONSTS[b]))
                    elif op == LDK: results.append((d, gmem[("k", cur + b)]))
                    elif op in (STG, STC):
                        if a not in local: reads[w].add(a)
                        gmem[b + (cur if op == STC else 0)] = get(a)
                    else:
                        for x in ((a, b) if op in two else (a,)):
                            if x not in local: reads[w].add(x)
                        results.append((d, apply_op(op, get(a), get(b) if op in two else None)))
                assert len({d for d, v in results}) == len(results), "two slots of a step write one register"
                for d, v in results: local[d] = v; writes[w].add(d)
                if bar: break
            new[w] = local
        for w in range(K):
            for v in range(K):
                if v != w:
                    assert not (writes[w] & writes[v]), "two waves write one register in a round"
                    assert not (writes[w] & reads[v]), "a wave reads a register another wave writes in the same round"
        for w in range(K): regs.update(new[w])
    assert all(pc[w] == len(streams[w]) for w in range(K)), "streams out of step"


# pinned register maps of the three chains
def pin_maps():
    miller = {n: i for i, n in enumerate(F_NAMES + ["TX", "TY", "TZ", "QX", "QY", "PP"])}
    sub = {n: i for i, n in enumerate(["TX", "TY", "TZ", "QX", "QY"])}
    names = ["a%d" % i for i in range(6)] + ["b%d" % i for i in range(6)]
    fin = {n: i for i, n in enumerate(names)}
    cb = {n: i for i, n in enumerate(F_NAMES + ["P1", "P1Z", "P2"])}
    return {"M": miller, "S": sub, "F": fin, "B": cb, "L": sub}


def estimate(streams):
    """cost units of a scheduled program: the heaviest wave of every round plus a barrier"""
    K = len(streams); pc = [0] * K; tot = 0
    while pc[0] < len(streams[0]):
        mx = 0
        for w in range(K):
            l = 0
            while True:
                op, hs, bar = streams[w][pc[w]]; pc[w] += 1; l += COST[op] + 1
                if bar: break
            mx = max(mx, l)
        tot += mx + 2
    return tot


def compile_all(K, nreg, verbose=False, search=True):
    """Schedules and allocates every program.  The list scheduler has three knobs -- the look-ahead window (register pressure), noise on
    the critical-path priorities, the size of a round -- and the allocation either fits the chain's register budget or not: a seeded
    search over the knobs keeps the fitting schedule with the smallest estimated time (deterministic: the emitted header is reproducible)."""
    progs = build_programs()
    pins = pin_maps()
    outp = {}
    for name, p in progs.items():
        pm = pins[name[0]]
        trials = [(w, None, None) for w in (100000, 96, 64, 48, 32, 24, 16, 12, 8)]
        if search:
            rnd = random.Random(sum(ord(c) for c in name) * 1000 + K)
            nt = 240 if len(p.ops) < 300 else 24
            for _ in range(nt):
                trials.append((rnd.choice([8, 12, 16, 20, 24, 28, 32, 40, 48, 64, 96, 100000]), (random.Random(rnd.randrange(1 << 30)), rnd.choice([0.0, 0.1, 0.3, 0.6, 1.0])),
                               rnd.choice([COST[MUL], COST[MUL], COST[MUL] * 3 // 2, COST[MUL] * 2])))
        best = None
        for window, noise, budget0 in trials:
            rounds, done = schedule(p, K, window, noise, budget0, SLOTS[name[0]])
            streams, used = allocate(p, rounds, done, pm, nreg[name[0]], SLOTS[name[0]])
            if streams is None: continue
            e = estimate(streams)
            if best is None or e < best[4]: best = (streams, used, window, len(rounds), e)
            if not search: break
        assert best, "no schedule of %s fits %d registers" % (name, nreg[name[0]])
        outp[name] = best
        if verbose:
            tot = sum(COST[o[0]] for o in p.ops)
            print("%-7s K=%d ops %4d cost %5d rounds %3d regs %2d window %6d estimate %5d" % (name, K, len(p.ops), tot, best[3], best[1], best[2], best[4]))
    return progs, outp


# ---------------------------------------------------------------- checks (pure Python)
def check():
    import bn254 as O
    rnd = random.Random(5)
    progs = build_programs()
    sc = scripts()
    def miller(q, pt):
        g = {SLOT_QX: q[0], SLOT_QY: q[1], SLOT_P: (pt[0], pt[1])}
        run_script(progs, sc["miller"], g)
        return [g[SLOT_F0 + k] for k in range(6)]
    one12 = [(1, 0)] + [(0, 0)] * 5
    def finish(fs):          # fs = [chain A's value, chain B's value, the key's constant]
        g = {}
        for j, f in enumerate(fs):
            for k in range(6): g[SLOT_FIN(j, k) if j < 2 else ("k", k)] = f[k]
        run_script(progs, sc["finish"], g)
        return [g[SLOT_RES + k] for k in range(6)]
    def subgroup(q):
        g = {SLOT_QX: q[0], SLOT_QY: q[1]}
        run_script(progs, sc["subgroup"], g)
        return g[SLOT_SZ] != (0, 0) and g[SLOT_SH] == (0, 0) and g[SLOT_SR] == (0, 0)
    def lines(q1, q2):       # chain L twice: the table chain B reads, [step][point][3]
        tab = {}
        for j, q in enumerate((q1, q2)):
            g = {SLOT_QX: q[0], SLOT_QY: q[1]}
            run_script(progs, sc["lines"], g)
            for k, v in g.items():
                if k >= 6 * 0 and k not in (SLOT_QX, SLOT_QY) or k >= 2: pass
            for k in range(N_LINE_SLOTS // 6):
                for c in range(3): tab[("k", 6 * k + 3 * j + c)] = g[LINE_SLOT0 + 6 * k + c]
        return tab
    def proj(p1):           # an affine G1 point as the parsing kernel hands it over: a random Jacobian form (X, Y, Z), packed (X Z, Y), (Z^3, 0)
        zz = rnd.randrange(1, P)
        Xj, Yj = p1[0] * zz * zz % P, p1[1] * zz * zz * zz % P
        return (Xj * zz % P, Yj), (zz * zz * zz % P, 0)
    def miller_b(tab, p1, p2):
        g = dict(tab); g[SLOT_P], g[SLOT_QX] = proj(p1); g[PAIR_SLOTS + SLOT_P] = p2
        run_script(progs, sc["miller_b"], g)
        return [g[SLOT_F0 + k] for k in range(6)]
    a, b = rnd.randrange(1, R), rnd.randrange(1, R)
    Pt, Qt = O.G1C.mul_pt(O.G1, rnd.randrange(1, R)), O.G2C.mul_pt(O.G2, rnd.randrange(1, R))
    aP, bQ, abP = O.G1C.mul_pt(Pt, a), O.G2C.mul_pt(Qt, b), O.G1C.mul_pt(Pt, a * b % R)
    # e(aP, bQ) e(-abP, Q) == 1;  e(P, Q) != 1
    f1, f2 = miller(bQ, aP), miller(Qt, O.G1C.neg_pt(abP))
    assert finish([f1, f2, one12]) == one12, "bilinearity"
    assert finish([miller(Qt, Pt), one12, one12]) != one12, "non-degeneracy"
    assert finish([f1, miller(Qt, O.G1C.neg_pt(aP)), one12]) != one12
    assert O.pairing_product_is_one([(aP, bQ), (O.G1C.neg_pt(abP), Qt)])
    # chain B on the line table of (bQ, Q) = the product of the two separate Miller values
    tab = lines(bQ, Qt)
    fb = miller_b(tab, aP, O.G1C.neg_pt(abP))
    assert finish([fb, one12, one12]) == one12, "chain B"
    assert finish([miller_b(tab, aP, O.G1C.neg_pt(aP)), one12, one12]) != one12
    c = rnd.randrange(1, R)
    assert finish([miller(bQ, O.G1C.mul_pt(Pt, c)), miller_b(tab, O.G1C.neg_pt(O.G1C.mul_pt(Pt, c)), Pt), miller(Qt, O.G1C.neg_pt(Pt))]) == one12
    assert subgroup(Qt) and subgroup(bQ)
    # a point of the twist outside the subgroup: random x until the curve equation has a root
    while True:
        x = (rnd.randrange(P), rnd.randrange(P))
        rhs = O.f2_add(O.f2_mul(O.f2_sq(x), x), O.B2)
        y = f2sqrt(rhs)
        if y is not None: break
    assert O.G2C.is_on_curve((x, y)) and O.G2C.mul_pt((x, y), R, reduce=False) is not None
    assert not subgroup((x, y))
    cof = O.G2C.mul_pt((x, y), R, reduce=False)                       # a point of the cofactor part (order divides 2p - r)
    assert O.G2C.is_on_curve(cof) and not subgroup(cof) and not subgroup(O.G2C.add_pts(cof, Qt))
    print("formulas ok (bilinearity, non-degeneracy, chain B against chain A, subgroup check)")
    # scheduled + allocated programs against the traced ones, on the same inputs
    for K, nreg in ((1, {"M": 64, "S": 32, "F": 96, "B": 64, "L": 32}), (4, NREG)):
        progs2, comp = compile_all(K, nreg, verbose=True, search=K > 1)
        gb = dict(tab); gb[SLOT_P], gb[SLOT_QX] = proj(aP); gb[PAIR_SLOTS + SLOT_P] = O.G1C.neg_pt(abP)
        for chain, gm in (("miller", {SLOT_QX: bQ[0], SLOT_QY: bQ[1], SLOT_P: aP}), ("subgroup", {SLOT_QX: bQ[0], SLOT_QY: bQ[1]}),
                          ("lines", {SLOT_QX: bQ[0], SLOT_QY: bQ[1]}), ("miller_b", gb),
                          ("finish", {(SLOT_FIN(j, k) if j < 2 else ("k", k)): [f1, f2, one12][j][k] for j in range(3) for k in range(6)})):
            g_ref = dict(gm); run_script(progs, sc[chain], g_ref)
            g = dict(gm); regs = {r: (rnd.randrange(P), rnd.randrange(P)) for r in range(256)}
            cur = 0
            for ent in sc[chain]:
                name, adv = ent if isinstance(ent, tuple) else (ent, 0)
                simulate(comp[name][0], regs, g, cur); cur += adv
            assert g == g_ref, "scheduled %s differs at K = %d" % (chain, K)
        tot = {c: sum(comp[e[0] if isinstance(e, tuple) else e][3] for e in sc[c]) for c in sc}
        print("K = %d: rounds per chain %s" % (K, tot))
        print("K = %d: estimated cost units per chain %s" % (K, {c: sum(comp[e[0] if isinstance(e, tuple) else e][4] for e in sc[c]) for c in sc}))
    print("schedules ok")


def f2sqrt(a):
    # Fq2 square root (p = 3 mod 4): standard algorithm 9 of Adj & Rodriguez-Henriquez
    if a == (0, 0): return (0, 0)
    a1 = f2pow(a, (P - 3) // 4)
    alpha = f2mul(f2mul(a1, a1), a)
    a0 = f2mul(f2pow(alpha, P), alpha)
    if a0 == (P - 1, 0): return None
    x0 = f2mul(a1, a)
    if alpha == (P - 1, 0): return f2mul((0, 1), x0)
    b = f2pow(f2add((1, 0), alpha), (P - 1) // 2)
    return f2mul(b, x0)


NREG = {"M": 46, "S": 16, "F": 62, "B": 46, "L": 24}
# slots of a wave (= operations of one opcode it runs at a time): two = 32 envelopes per workgroup.  (Four slots = 16 envelopes per workgroup were
# built for the final exponentiation -- 256 workgroups instead of 128 per 4096 envelopes, 80 registers: estimate -20 %, measured 3.57 against
# 3.39 ms and less throughput on large batches; the interpreter keeps the two-slot form only.)
SLOTS = {"M": 2, "S": 2, "F": 2, "B": 2, "L": 2}       # what the LDS holds with 32 envelopes per workgroup: 46 x 80 bytes x 32 (Miller) + 16 x 80 x 32 (subgroup) on one CU; 62 x 80 x 32 (finish)


# ---------------------------------------------------------------- emission
def mont_limbs(x):
    v = x * (1 << 260) % P
    return [(v >> (26 * i)) & 0x3ffffff if i < 9 else v >> 234 for i in range(10)]


def emit(path, Ks=(4,)):
    sc = scripts()
    L = []
    L.append("// GENERATED by tools/gen_fq2vm.py -- do not edit.  Micro-operation tables of the Fq2 virtual machine (fq2vm.h): the Groth16")
    L.append("// verifier's Miller loops (optimal ate, loop count 6x + 2 in NAF: chain A = a pair with a proof's G2 point, chain B = the two pairs on the")
    L.append("// key's gamma and delta over line tables that chain L computes per key), G2 subgroup check (psi(Q) = [6x^2] Q) and final exponentiation (x-chain, signed digits),")
    L.append("// scheduled for K cooperating wavefronts.  A micro-operation is one word per slot of the wave: op | barrier << 7 | dst << 8 | a << 16 | b << 24 for")
    L.append("// the first slot, then present | dst << 8 | a << 16 | b << 24 for the others (the same opcode on other registers of the same envelopes).")
    L.append("#pragma once")
    L.append("#include <cstdint>")
    L.append("namespace zkp { namespace fq2vm {")
    allp = None
    names = None
    for K in Ks:
        progs, comp = compile_all(K, NREG)
        names = list(progs.keys())
        code, off = [], []
        for name in names:
            streams = comp[name][0]
            for w in range(K):
                off.append(len(code))
                ns = SLOTS[name[0]]
                for op, hs, bar in streams[w]:
                    assert all(x < 256 for h in hs for x in h) and len(hs) <= ns
                    code.append(op | (BAR if bar else 0) | hs[0][0] << 8 | hs[0][1] << 16 | hs[0][2] << 24)
                    for q in range(1, ns): code.append((1 | hs[q][0] << 8 | hs[q][1] << 16 | hs[q][2] << 24) if q < len(hs) else 0)
                code += [END] + [0] * (ns - 1)
        L.append("static const uint32_t CODE_K%d[%d] = {" % (K, len(code)))
        for i in range(0, len(code), 12): L.append("    " + ", ".join("0x%08xu" % c for c in code[i:i + 12]) + ",")
        L.append("};")
        L.append("static const uint32_t OFF_K%d[%d] = {%s};      // [program][wave]" % (K, len(off), ", ".join(str(o) for o in off)))
        L.append("static const uint32_t REGS_K%d[5] = {%s};      // registers used by the chains: miller A, subgroup, finish, miller B, lines" % (
            K, ", ".join(str(max(comp[n][1] for n in names if n[0] == c)) for c in "MSFBL")))
    L.append("enum { %s, N_PROGRAMS };" % ", ".join("P_" + n for n in names))
    L.append("// script entry = program | (cursor advance after it) << 8; the cursor offsets LDK (and STC) operands")
    for chain in ("miller", "subgroup", "finish", "miller_b", "lines"):
        s = [(e if isinstance(e, tuple) else (e, 0)) for e in sc[chain]]
        L.append("static const uint16_t SCRIPT_%s[%d] = {%s};" % (chain.upper(), len(s), ", ".join("P_%s | %d << 8" % e if e[1] else "P_" + e[0] for e in s)))
    L.append("static const uint32_t N_CONSTS = %d;" % len(CONSTS))
    L.append("static const uint32_t CONSTS[%d][20] = {      // Montgomery limbs (bn254_fq.h), c0 then c1" % len(CONSTS))
    for c in CONSTS: L.append("    {" + ", ".join("0x%xu" % x for x in mont_limbs(c[0]) + mont_limbs(c[1])) + "},")
    L.append("};")
    L.append("enum { SLOT_QX = %d, SLOT_QY = %d, SLOT_P = %d, SLOT_F0 = %d, SLOT_SZ = %d, SLOT_SH = %d, SLOT_SR = %d, PAIR_SLOTS = %d, SLOT_RES = %d, N_SLOTS = %d, LINE_SLOT0 = %d, N_LINE_SLOTS = %d };" % (
        SLOT_QX, SLOT_QY, SLOT_P, SLOT_F0, SLOT_SZ, SLOT_SH, SLOT_SR, PAIR_SLOTS, SLOT_RES, N_SLOTS, LINE_SLOT0, N_LINE_SLOTS))
    L.append("} }  // namespace zkp::fq2vm")
    open(path, "w").write("\n".join(L) + "\n")
    print("wrote", path)


if __name__ == "__main__":
    if "--check" in sys.argv: check()
    else: emit(os.path.join(ROOT, "libzkp_amd", "csrc", "fq2vm_programs.h"))
