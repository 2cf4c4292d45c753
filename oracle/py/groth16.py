"""ORACLE (test infrastructure only -- never imported by the product path).

MiMC-5 commitments, the equality / membership R1CS circuits, Groth16 setup, prover and verifier over BN254,
restating what the reference does through ark-groth16 / ark-r1cs-std / ark-relations ^0.5 (Cargo.toml:16-27; crates
not vendored, unpinned, unbuildable here).  PARITY UNPINNED at the proof-byte level: the reference's proving key comes
from an OsRng trusted setup and its prover draws r, s from OsRng (snark.rs:310,331,363,441); nothing in its tests pins
bytes.  Pinned here: MiMC (pure bigint from the reference's own constants recipe, snark.rs:186-211), and soundness --
every proof must pass the pairing check e(A,B) = e(alpha,beta) e(sum x_i IC_i, gamma) e(C, delta) and fail for a wrong
public input (snark.rs:639-640).

Reference lines followed:
  snark.rs:186-199   MiMC round constants  c_i = SHA-256("libzkp_mimc_v1:" || u64le(i)) as LE integer mod r
  snark.rs:201-221   mimc_hash_native, fr_to_commitment (32-byte LE canonical)
  snark.rs:232-247   mimc_hash_circuit (t*t, t2*t2, t4*t per round)
  snark.rs:262-291   EqualityCircuit.generate_constraints (allocation order a, b | MiMC | commitment)
  snark.rs:514-585   MembershipCircuit.generate_constraints
  snark.rs:343-374   prove_equality_zk / serialization; 405-452 prove_membership_zk
  snark.rs:377-401, 455-495 verify_* and the public-input order
The Groth16 algebra follows the published protocol with ark-groth16's LibsnarkReduction conventions:
domain = next power of two >= constraints + instance variables, extra rows a[n_constraints + i] = z_i, coset offset =
Fr multiplicative generator 5, h = (a*b - c)/Z on the coset.

Randomness (project tape): r, s = from_wide(draw64(seed, 0x47313600, 0/1)) mod r_BN254; the setup's toxic waste is
derived from a 32-byte setup seed the same way (index 0x47313601), generators are the standard (1,2) / EIP-197 G2.
"""
import hashlib

from . import bn254 as bn
from .merlin import shake256

R = bn.R
MIMC_ROUNDS = 110
FR_GENERATOR = 5
MAX_SET_SIZE = 64
TAPE_DOMAIN = b"libzkp-amd/tape/v1"


def mimc_constants():
    return [int.from_bytes(hashlib.sha256(b"libzkp_mimc_v1:" + i.to_bytes(8, "little")).digest(), "little") % R
            for i in range(MIMC_ROUNDS)]


_C = mimc_constants()


def mimc_hash_native(value):
    x = value % R
    for c in _C:
        t = (x + c) % R
        x = pow(t, 5, R)
    return x


def commit_value_snark(value):
    """utils/commitment.rs:14-16."""
    return mimc_hash_native(value).to_bytes(32, "little")


def draw_fr(seed, idx, slot):
    return int.from_bytes(shake256(TAPE_DOMAIN + seed + idx.to_bytes(4, "little") + slot.to_bytes(4, "little"), 64), "little") % R


# ---------------------------------------------------------------- R1CS
class R1CS:
    def __init__(self):
        self.n_inst = 1            # variable 0 of the instance block is the constant ONE
        self.n_wit = 0
        self.rows = []             # (A, B, C) dicts keyed by ('i', k) / ('w', k)
        self.inst_vals = [1]
        self.wit_vals = []

    def new_input(self, val):
        self.inst_vals.append(val % R)
        self.n_inst += 1
        return {("i", self.n_inst - 1): 1}

    def new_witness(self, val):
        self.wit_vals.append(val % R)
        self.n_wit += 1
        return {("w", self.n_wit - 1): 1}

    def value(self, lc):
        return sum(c * (self.inst_vals[k] if t == "i" else self.wit_vals[k]) for (t, k), c in lc.items()) % R

    def enforce(self, a, b, c):
        self.rows.append((dict(a), dict(b), dict(c)))

    def mul(self, a, b):
        """AllocatedFp::mul: allocate the product as a witness and enforce a * b = product."""
        p = self.new_witness(self.value(a) * self.value(b))
        self.enforce(a, b, p)
        return p

    def enforce_equal(self, a, b):
        self.enforce(lc_sub(a, b), ONE, {})

    def column(self, key):
        t, k = key
        return k if t == "i" else self.n_inst + k

    def assignment(self):
        return self.inst_vals + self.wit_vals


ONE = {("i", 0): 1}


def lc_add(a, b):
    out = dict(a)
    for k, c in b.items():
        out[k] = (out.get(k, 0) + c) % R
    return {k: c for k, c in out.items() if c}


def lc_scale(a, s):
    return {k: c * s % R for k, c in a.items() if c * s % R}


def lc_sub(a, b):
    return lc_add(a, lc_scale(b, R - 1))


def mimc_circuit(cs, x):
    for c in _C:
        t = lc_add(x, lc_scale(ONE, c))
        t2 = cs.mul(t, t)
        t4 = cs.mul(t2, t2)
        x = cs.mul(t4, t)
    return x


def equality_circuit(a, b, commitment_fr):
    """snark.rs:262-291."""
    cs = R1CS()
    a_var = cs.new_witness(a)
    b_var = cs.new_witness(b)
    cs.enforce_equal(a_var, b_var)
    h = mimc_circuit(cs, a_var)
    c_var = cs.new_input(commitment_fr)
    cs.enforce_equal(h, c_var)
    return cs


def membership_circuit(value, sel, set_values, is_real, commitment_fr):
    """snark.rs:514-585."""
    assert len(sel) == len(set_values) == len(is_real) == MAX_SET_SIZE
    cs = R1CS()
    v = cs.new_witness(value)
    h = mimc_circuit(cs, v)
    c_var = cs.new_input(commitment_fr)
    cs.enforce_equal(h, c_var)
    set_vars = [cs.new_input(x) for x in set_values]
    real = []
    for b in is_real:                     # Boolean::new_input enforces (1 - b) * b = 0
        bv = cs.new_input(int(b))
        cs.enforce(lc_sub(ONE, bv), bv, {})
        real.append(bv)
    sels = []
    for b in sel:                         # Boolean::new_witness likewise
        bv = cs.new_witness(int(b))
        cs.enforce(lc_sub(ONE, bv), bv, {})
        sels.append(bv)
    total = {}
    for i in range(MAX_SET_SIZE):
        total = lc_add(total, sels[i])
        prod = cs.mul(sels[i], lc_sub(ONE, real[i]))
        cs.enforce_equal(prod, {})
    cs.enforce_equal(total, ONE)
    acc = {}
    for i in range(MAX_SET_SIZE):
        acc = lc_add(acc, cs.mul(sels[i], lc_sub(v, set_vars[i])))
    cs.enforce_equal(acc, {})
    return cs


def is_satisfied(cs):
    return all(cs.value(a) * cs.value(b) % R == cs.value(c) for a, b, c in cs.rows)


# ---------------------------------------------------------------- radix-2 domain
def domain_size(cs):
    n = len(cs.rows) + cs.n_inst
    m = 1
    while m < n:
        m *= 2
    return m


def root_of_unity(m):
    return pow(FR_GENERATOR, (R - 1) // m, R)


def ntt(vals, w):
    n = len(vals)
    a = list(vals)
    j = 0
    for i in range(1, n):                 # bit reversal
        bit = n >> 1
        while j & bit:
            j ^= bit
            bit >>= 1
        j ^= bit
        if i < j:
            a[i], a[j] = a[j], a[i]
    length = 2
    while length <= n:
        wl = pow(w, n // length, R)
        for s in range(0, n, length):
            x = 1
            for k in range(length // 2):
                u, v = a[s + k], a[s + k + length // 2] * x % R
                a[s + k], a[s + k + length // 2] = (u + v) % R, (u - v) % R
                x = x * wl % R
        length *= 2
    return a


def intt(vals, w):
    n = len(vals)
    ninv = pow(n, R - 2, R)
    return [x * ninv % R for x in ntt(vals, pow(w, R - 2, R))]


def coset_ntt(coeffs, w, g):
    return ntt([c * pow(g, i, R) % R for i, c in enumerate(coeffs)], w)


def coset_intt(evals, w, g):
    gi = pow(g, R - 2, R)
    return [c * pow(gi, i, R) % R for i, c in enumerate(intt(evals, w))]


def witness_map(cs):
    """LibsnarkReduction::witness_map_from_matrices -> h coefficients (length m)."""
    m = domain_size(cs)
    w = root_of_unity(m)
    nc = len(cs.rows)
    a, b, c = [0] * m, [0] * m, [0] * m
    for i, (A, B, C) in enumerate(cs.rows):
        a[i], b[i], c[i] = cs.value(A), cs.value(B), cs.value(C)
    for i in range(cs.n_inst):
        a[nc + i] = cs.inst_vals[i]
    g = FR_GENERATOR
    a = coset_ntt(intt(a, w), w, g)
    b = coset_ntt(intt(b, w), w, g)
    c = coset_ntt(intt(c, w), w, g)
    zinv = pow((pow(g, m, R) - 1) % R, R - 2, R)
    ab = [((x * y - z) % R) * zinv % R for x, y, z in zip(a, b, c)]
    return coset_intt(ab, w, g)


# ---------------------------------------------------------------- setup (generate_random_parameters, fixed generators)
class ProvingKey:
    pass


def setup(cs, setup_seed, circuit_tag):
    """Circuit-specific Groth16 setup with toxic waste derived from (setup_seed, circuit_tag)."""
    tag = int.from_bytes(hashlib.sha256(circuit_tag).digest()[:3], "little")
    alpha, beta, gamma, delta, tau = (draw_fr(setup_seed, 0x47313601, 8 * tag + k) or 1 for k in range(5))
    m = domain_size(cs)
    w = root_of_unity(m)
    nc, nv = len(cs.rows), cs.n_inst + cs.n_wit
    zt = (pow(tau, m, R) - 1) % R
    # Lagrange basis at tau: L_j = Z(tau)/m * w^j / (tau - w^j)
    minv = pow(m, R - 2, R)
    lag = []
    wj = 1
    for j in range(m):
        lag.append(zt * minv % R * wj % R * pow((tau - wj) % R, -1, R) % R)
        wj = wj * w % R
    At, Bt, Ct = [0] * nv, [0] * nv, [0] * nv
    for j, (A, B, C) in enumerate(cs.rows):
        for key, coef in A.items():
            At[cs.column(key)] = (At[cs.column(key)] + coef * lag[j]) % R
        for key, coef in B.items():
            Bt[cs.column(key)] = (Bt[cs.column(key)] + coef * lag[j]) % R
        for key, coef in C.items():
            Ct[cs.column(key)] = (Ct[cs.column(key)] + coef * lag[j]) % R
    for i in range(cs.n_inst):
        At[i] = (At[i] + lag[nc + i]) % R
    pk = ProvingKey()
    pk.trapdoor = dict(alpha=alpha, beta=beta, gamma=gamma, delta=delta, tau=tau, At=At, Bt=Bt, Ct=Ct, zt=zt)
    pk.m, pk.n_inst, pk.n_wit = m, cs.n_inst, cs.n_wit
    g1, g2 = bn.G1C, bn.G2C
    ginv, dinv = pow(gamma, R - 2, R), pow(delta, R - 2, R)
    pk.alpha_g1 = g1.mul_pt(bn.G1, alpha)
    pk.beta_g1 = g1.mul_pt(bn.G1, beta)
    pk.delta_g1 = g1.mul_pt(bn.G1, delta)
    pk.beta_g2 = g2.mul_pt(bn.G2, beta)
    pk.gamma_g2 = g2.mul_pt(bn.G2, gamma)
    pk.delta_g2 = g2.mul_pt(bn.G2, delta)
    pk.a_query = [g1.mul_pt(bn.G1, x) for x in At]
    pk.b_g1_query = [g1.mul_pt(bn.G1, x) for x in Bt]
    pk.b_g2_query = [g2.mul_pt(bn.G2, x) for x in Bt]
    pk.gamma_abc_g1 = [g1.mul_pt(bn.G1, (beta * At[i] + alpha * Bt[i] + Ct[i]) % R * ginv % R) for i in range(cs.n_inst)]
    pk.l_query = [g1.mul_pt(bn.G1, (beta * At[k] + alpha * Bt[k] + Ct[k]) % R * dinv % R) for k in range(cs.n_inst, nv)]
    pk.h_query = [g1.mul_pt(bn.G1, pow(tau, i, R) * zt % R * dinv % R) for i in range(m - 1)]
    return pk


def serialize_pk(pk):
    """ark-serialize uncompressed ProvingKey<Bn254>: vk{alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1}, beta_g1,
    delta_g1, a_query, b_g1_query, b_g2_query, h_query, l_query; Vec = u64 LE length + items."""
    def vec(items, ser):
        return len(items).to_bytes(8, "little") + b"".join(ser(x) for x in items)
    return (bn.ser_g1(pk.alpha_g1) + bn.ser_g2(pk.beta_g2) + bn.ser_g2(pk.gamma_g2) + bn.ser_g2(pk.delta_g2) + vec(pk.gamma_abc_g1, bn.ser_g1)
            + bn.ser_g1(pk.beta_g1) + bn.ser_g1(pk.delta_g1) + vec(pk.a_query, bn.ser_g1) + vec(pk.b_g1_query, bn.ser_g1)
            + vec(pk.b_g2_query, bn.ser_g2) + vec(pk.h_query, bn.ser_g1) + vec(pk.l_query, bn.ser_g1))


# ---------------------------------------------------------------- prover (create_proof_with_reduction) and verifier
def prove(pk, cs, r, s):
    assert is_satisfied(cs)
    z = cs.assignment()
    h = witness_map(cs)
    g1, g2 = bn.G1C, bn.G2C

    def coeff(initial, query, vk_param, curve):
        acc = curve.add_pts(initial, query[0])
        acc = curve.add_pts(acc, curve.msm(z[1:], query[1:]))
        return curve.add_pts(acc, vk_param)

    g_a = coeff(g1.mul_pt(pk.delta_g1, r), pk.a_query, pk.alpha_g1, g1)
    g1_b = coeff(g1.mul_pt(pk.delta_g1, s), pk.b_g1_query, pk.beta_g1, g1)
    g2_b = coeff(g2.mul_pt(pk.delta_g2, s), pk.b_g2_query, pk.beta_g2, g2)
    h_acc = g1.msm(h[: pk.m - 1], pk.h_query)
    l_acc = g1.msm(cs.wit_vals, pk.l_query)
    g_c = g1.add_pts(g1.mul_pt(g_a, s), g1.mul_pt(g1_b, r))
    g_c = g1.add_pts(g_c, g1.neg_pt(g1.mul_pt(pk.delta_g1, r * s % R)))
    g_c = g1.add_pts(g1.add_pts(g_c, l_acc), h_acc)
    return bn.ser_g1(g_a) + bn.ser_g2(g2_b) + bn.ser_g1(g_c)


def prove_with_trapdoor(pk, cs, r, s):
    """Independent route to the same proof: evaluate everything in the exponent with the toxic waste (no MSM, no FFT)."""
    t = pk.trapdoor
    z = cs.assignment()
    At = sum(x * y for x, y in zip(z, t["At"])) % R
    Bt = sum(x * y for x, y in zip(z, t["Bt"])) % R
    Ct = sum(x * y for x, y in zip(z, t["Ct"])) % R
    dinv = pow(t["delta"], R - 2, R)
    a = (t["alpha"] + At + r * t["delta"]) % R
    b = (t["beta"] + Bt + s * t["delta"]) % R
    h_tau = (At * Bt - Ct) % R * pow(t["zt"], R - 2, R) % R          # h(tau) = (A(tau) B(tau) - C(tau)) / Z(tau)
    wit = sum(z[k] * ((t["beta"] * t["At"][k] + t["alpha"] * t["Bt"][k] + t["Ct"][k]) % R) for k in range(pk.n_inst, len(z))) % R
    c = (wit * dinv + h_tau * t["zt"] % R * dinv + s * a + r * b - r * s % R * t["delta"]) % R
    return bn.ser_g1(bn.G1C.mul_pt(bn.G1, a)) + bn.ser_g2(bn.G2C.mul_pt(bn.G2, b)) + bn.ser_g1(bn.G1C.mul_pt(bn.G1, c))


def verify(pk, public_inputs, proof):
    """Groth16 verification equation (ark-groth16 verify_with_processed_vk), public_inputs excludes the constant 1."""
    if len(proof) != 256 or len(public_inputs) + 1 != len(pk.gamma_abc_g1):
        return False
    ok_a, A = bn.de_g1(proof[:64])
    ok_b, B = bn.de_g2(proof[64:192])
    ok_c, C = bn.de_g1(proof[192:])
    if not (ok_a and ok_b and ok_c):
        return False
    acc = pk.gamma_abc_g1[0]
    for x, pt in zip(public_inputs, pk.gamma_abc_g1[1:]):
        acc = bn.G1C.add_pts(acc, bn.G1C.mul_pt(pt, x))
    neg = bn.G1C.neg_pt
    return bn.pairing_product_is_one([(A, B), (neg(pk.alpha_g1), pk.beta_g2), (neg(acc), pk.gamma_g2), (neg(C), pk.delta_g2)])


class VerifyingKey:
    """The verifying key that leads an ark-serialize uncompressed ProvingKey<Bn254> file (snark.rs:31-38: `{prefix}_pk.bin`):
    alpha_g1 | beta_g2 | gamma_g2 | delta_g2 | u64 n | gamma_abc_g1[n].  Enough for verify() above (it reads exactly these fields)."""


def vk_from_pk_bytes(blob):
    vk = VerifyingKey()
    ok, vk.alpha_g1 = bn.de_g1(blob[0:64])
    ok2, vk.beta_g2 = bn.de_g2(blob[64:192])
    ok3, vk.gamma_g2 = bn.de_g2(blob[192:320])
    ok4, vk.delta_g2 = bn.de_g2(blob[320:448])
    n = int.from_bytes(blob[448:456], "little")
    if not (ok and ok2 and ok3 and ok4) or n > 1 << 16 or len(blob) < 456 + 64 * n:
        raise ValueError("malformed key file")
    vk.gamma_abc_g1 = []
    for i in range(n):
        ok, pt = bn.de_g1(blob[456 + 64 * i:520 + 64 * i])
        if not ok:
            raise ValueError("malformed key file")
        vk.gamma_abc_g1.append(pt)
    return vk


def verify_equality_envelope_under(vk, proof_env):
    """verify_equality_with_commitment's framing checks with the envelope's own commitment, under an explicit key."""
    if len(proof_env) != 298 or proof_env[:2] != bytes([2, 2]):
        return False
    if int.from_bytes(proof_env[2:6], "little") != 256 or int.from_bytes(proof_env[6:10], "little") != 32:
        return False
    c = int.from_bytes(proof_env[266:298], "little")
    if c >= R:
        return False
    return verify(vk, [c], proof_env[10:266])


# ---------------------------------------------------------------- libzkp framing
_KEYS = {}


def equality_key(setup_seed):
    if ("eq", setup_seed) not in _KEYS:
        _KEYS[("eq", setup_seed)] = setup(equality_circuit(0, 0, 0), setup_seed, b"equality_mimc")     # snark.rs:329-339 dummy circuit
    return _KEYS[("eq", setup_seed)]


def membership_key(setup_seed):
    if ("mem", setup_seed) not in _KEYS:
        dummy = membership_circuit(0, [False] * MAX_SET_SIZE, [0] * MAX_SET_SIZE, [False] * MAX_SET_SIZE, 0)   # snark.rs:309-320
        _KEYS[("mem", setup_seed)] = setup(dummy, setup_seed, b"membership_mimc")
    return _KEYS[("mem", setup_seed)]


def envelope(scheme, proof, commitment):
    return bytes([2, scheme]) + len(proof).to_bytes(4, "little") + len(commitment).to_bytes(4, "little") + proof + commitment


def prove_equality(val1, val2, setup_seed, seed):
    """proof::equality_proof::prove_equality (equality_proof.rs:10-32) -> 298-byte envelope."""
    if val1 != val2:
        raise ValueError("values are not equal")
    commitment = commit_value_snark(val1)
    cs = equality_circuit(val1, val2, int.from_bytes(commitment, "little"))
    pr = prove(equality_key(setup_seed), cs, draw_fr(seed, 0x47313600, 0), draw_fr(seed, 0x47313600, 1))
    return envelope(2, pr, commitment)


def verify_equality_with_commitment(proof_env, commitment, setup_seed):
    if len(proof_env) != 298 or proof_env[:2] != bytes([2, 2]) or proof_env[10 + 256:] != commitment:
        return False
    if int.from_bytes(proof_env[2:6], "little") != 256 or int.from_bytes(proof_env[6:10], "little") != 32:     # Proof::from_bytes (proof/mod.rs:58-76)
        return False
    c = int.from_bytes(commitment, "little")
    if c >= R:
        return False
    return verify(equality_key(setup_seed), [c], proof_env[10:266])


def membership_inputs(value, the_set):
    pos = the_set.index(value)
    set_values = list(the_set) + [0] * (MAX_SET_SIZE - len(the_set))
    is_real = [True] * len(the_set) + [False] * (MAX_SET_SIZE - len(the_set))
    sel = [i == pos for i in range(MAX_SET_SIZE)]
    return sel, set_values, is_real


def prove_membership(value, the_set, setup_seed, seed):
    """proof::set_membership::prove_membership (set_membership.rs:12-38)."""
    if not the_set:
        raise ValueError("set cannot be empty")
    if value not in the_set:
        raise ValueError("value %d is not in the provided set" % value)
    if len(the_set) > MAX_SET_SIZE:
        raise ValueError("set size %d exceeds maximum allowed size %d" % (len(the_set), MAX_SET_SIZE))
    commitment = commit_value_snark(value)
    sel, set_values, is_real = membership_inputs(value, list(the_set))
    cs = membership_circuit(value, sel, set_values, is_real, int.from_bytes(commitment, "little"))
    pr = prove(membership_key(setup_seed), cs, draw_fr(seed, 0x47313600, 0), draw_fr(seed, 0x47313600, 1))
    payload = len(the_set).to_bytes(4, "little") + b"".join(x.to_bytes(8, "little") for x in the_set) + pr
    return envelope(4, payload, commitment)


def verify_membership(proof_env, the_set, setup_seed):
    n = len(the_set)
    if proof_env[:2] != bytes([2, 4]) or len(proof_env) != 10 + 4 + 8 * n + 256 + 32:
        return False
    if int.from_bytes(proof_env[2:6], "little") != 4 + 8 * n + 256 or int.from_bytes(proof_env[6:10], "little") != 32:
        return False
    payload, commitment = proof_env[10:-32], proof_env[-32:]
    if int.from_bytes(payload[:4], "little") != n or payload[4: 4 + 8 * n] != b"".join(x.to_bytes(8, "little") for x in the_set):
        return False
    c = int.from_bytes(commitment, "little")
    pub = [c] + list(the_set) + [0] * (MAX_SET_SIZE - n) + [1] * n + [0] * (MAX_SET_SIZE - n)     # snark.rs:482-492
    return verify(membership_key(setup_seed), pub, payload[4 + 8 * n:])
