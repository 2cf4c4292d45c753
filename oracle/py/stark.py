"""ORACLE (test infrastructure only -- never imported by the product path).

libzkp's improvement proof: a STARK over the 128-bit field p = 2^128 - 45*2^40 + 1 for the 8-row, 1-column trace
old, old+s, ..., old+7s = new with s = (new - old)/7 IN THE FIELD, restating what the reference obtains from
winterfell ^0.10 (Cargo.toml:22-23; crate not vendored, unpinned, unbuildable here) through
/root/reference/src/backend/stark.rs.  PARITY UNPINNED: the reference's tests only round-trip prove/verify
(stark.rs:258-266), no byte string, hash or field element of a proof is pinned anywhere, and Winterfell's exact
`Proof::write_into` layout is restated from its published design, not from source.  What IS pinned:
BLAKE3 by the specification's known answers, the field's 2^40-th root of unity by g^((p-1)/2^40) with g = 3, and
soundness by the verifier below (every proof accepted; tampered bytes, a wrong old/new and a non-linear trace rejected).
The whole proof is deterministic (no prover randomness: grinding 0, no zero-knowledge blinding).

Reference lines followed:
  stark.rs:15-84     ImprovementAir: one transition constraint of degree 1  next - cur - step = 0, two assertions
                     (column 0, step 0) = old and (column 0, step 7) = new, public inputs [old, new]
  stark.rs:94-101    ProofOptions(32 queries, blowup 8, grinding 0, FieldExtension::None, FRI folding 8, remainder <= 31)
  stark.rs:120-128   Blake3_256 hasher, MerkleTree vector commitment, DefaultRandomCoin
  stark.rs:151-186   trace construction and `proof.write_into`
  improvement_proof.rs:10-35, utils/commitment.rs:38-50   payload old || new || stark, SHA-256 binding commitment

Details restated from Winterfell's published design WITHOUT a source or vector to check them against (any of them may
differ from the crate; none affects soundness, all affect bytes):
  * coin seed elements: TraceInfo as [(width << 8) | aux segments, trace length], the modulus as two 8-byte halves,
    options as [(extension << 16) | (folding << 8) | remainder degree, grinding, blowup, queries], then the public inputs;
  * coin: seed' = H(seed || digest) on reseed, draw = first 16 bytes of H(seed || u64le(++counter)) with rejection of
    non-canonical values, integers = low bits of the first 8 bytes after seed' = H(seed || u64le(nonce));
  * coefficient draw order: transition, boundary (step 0, step 7); then z; then one DEEP coefficient per trace column
    and per composition column; T(z) and T(z g) share one coefficient;
  * Merkle tree node numbering and the batch-opening node order (one list per opened sibling pair, siblings appended
    to the list at the same position in each level's index list);
  * serialisation: context (trace info, modulus, options) | unique query count | commitments (u16 length) | trace
    queries | constraint queries (each: vint-prefixed values, vint-prefixed opening) | OOD frame (u16-prefixed trace
    states starting with the frame size 2, u16-prefixed evaluations) | FRI (layer count 0, u16-prefixed remainder,
    partition count 1) | u64 proof-of-work nonce | one zero byte for the absent GKR proof.

Pipeline (Winterfell's prover, restated): coin seed = BLAKE3(context elements || public inputs) -> interpolate the
trace column (domain 8) -> evaluate on the LDE coset 3*<w_64> -> hash rows, Merkle root, reseed -> draw 1 transition +
2 boundary coefficients -> constraint evaluations on the 16-point coset divided by their divisors -> interpolate, keep
the 8 coefficients of the single composition column -> evaluate on the LDE coset, hash rows, Merkle root, reseed ->
draw z -> send T(z), T(z g), reseed; send H(z), reseed -> draw 2 DEEP coefficients -> DEEP polynomial (degree 6) ->
FRI with zero folding layers (64 <= 32*8): the remainder is the 8 coefficients, committed by their hash, reseed ->
nonce 0 -> 32 query positions, sorted and deduplicated -> batch Merkle openings -> serialization.
"""
import hashlib

from .blake3 import blake3

P = 2**128 - 45 * 2**40 + 1
GENERATOR = 3                                   # multiplicative generator = LDE domain offset
TWO_ADICITY = 40
TWO_ADIC_ROOT = pow(GENERATOR, (P - 1) >> TWO_ADICITY, P)
TRACE_LEN, BLOWUP, CE_BLOWUP = 8, 8, 2
LDE_SIZE, CE_SIZE = TRACE_LEN * BLOWUP, TRACE_LEN * CE_BLOWUP
NUM_QUERIES, GRINDING, FOLDING, REMAINDER_MAX_DEGREE = 32, 0, 8, 31
FIELD_EXTENSION_NONE = 1


def inv(a):
    return pow(a % P, -1, P)


def root_of_unity(n):
    assert n & (n - 1) == 0
    return pow(TWO_ADIC_ROOT, (1 << TWO_ADICITY) // n, P)


def el_bytes(x):
    return (x % P).to_bytes(16, "little")


def hash_elements(elems):
    return blake3(b"".join(el_bytes(e) for e in elems))


def merge(a, b):
    return blake3(a + b)


def merge_with_int(seed, value):
    return blake3(seed + value.to_bytes(8, "little"))


# ---- polynomials over the field (coefficient lists, low degree first)
def interpolate(evals, offset=1):
    """values on offset*<w_n> (natural order) -> n coefficients"""
    n = len(evals)
    w_inv, n_inv, o_inv = inv(root_of_unity(n)), inv(n), inv(offset)
    out = []
    for k in range(n):
        acc = 0
        for i, e in enumerate(evals):
            acc += e * pow(w_inv, i * k, P)
        out.append(acc % P * n_inv % P * pow(o_inv, k, P) % P)
    return out


def horner(coefs, x):
    acc = 0
    for c in reversed(coefs):
        acc = (acc * x + c) % P
    return acc


def evaluate_on_coset(coefs, n, offset):
    w = root_of_unity(n)
    return [horner(coefs, offset * pow(w, i, P) % P) for i in range(n)]


def syn_div(coefs, a):
    """(poly(x)) / (x - a), exact division assumed (the remainder is dropped)"""
    out = [0] * (len(coefs) - 1)
    carry = 0
    for i in range(len(coefs) - 1, 0, -1):
        carry = (coefs[i] + carry * a) % P
        out[i - 1] = carry
    return out


# ---- Merkle tree (node i has children 2i, 2i+1; leaves hang below nodes n/2 .. n-1)
class MerkleTree:
    def __init__(self, leaves):
        n = len(leaves)
        assert n >= 2 and n & (n - 1) == 0
        self.leaves = list(leaves)
        self.nodes = [bytes(32)] * n
        for i in range(n - 1, 0, -1):
            if i >= n // 2:
                l, r = leaves[2 * (i - n // 2)], leaves[2 * (i - n // 2) + 1]
            else:
                l, r = self.nodes[2 * i], self.nodes[2 * i + 1]
            self.nodes[i] = merge(l, r)
        self.depth = n.bit_length() - 1

    @property
    def root(self):
        return self.nodes[1]

    def prove_batch(self, indexes):
        """indexes sorted and unique.  Returns the node lists of the batch opening (leaf values are sent separately)."""
        n = len(self.leaves)
        index_set = set(indexes)
        pairs = sorted({i & ~1 for i in indexes})
        nodes, nxt = [], []
        for p in pairs:
            nodes.append([self.leaves[i] for i in (p, p + 1) if i not in index_set])
            nxt.append((p + n) >> 1)
        for _ in range(1, self.depth):
            cur, nxt = nxt, []
            i = 0
            while i < len(cur):
                sib = cur[i] ^ 1
                if i + 1 < len(cur) and cur[i + 1] == sib:
                    i += 1
                else:
                    nodes[i].append(self.nodes[sib])
                nxt.append(sib >> 1)
                i += 1
        return nodes


def batch_root(leaf_values, indexes, nodes, depth):
    """recompute the root from opened leaves (by sorted unique index) and the batch proof's node lists; None if malformed"""
    n = 1 << depth
    index_map = {idx: k for k, idx in enumerate(indexes)}
    pairs = sorted({i & ~1 for i in indexes})
    if len(nodes) != len(pairs):
        return None
    ptr = [0] * len(nodes)
    cur_idx, cur_val = [], []
    for k, p in enumerate(pairs):
        vals = []
        for i in (p, p + 1):
            if i in index_map:
                vals.append(leaf_values[index_map[i]])
            else:
                if ptr[k] >= len(nodes[k]):
                    return None
                vals.append(nodes[k][ptr[k]]); ptr[k] += 1
        cur_idx.append((p + n) >> 1); cur_val.append(merge(vals[0], vals[1]))
    for _ in range(1, depth):
        nxt_idx, nxt_val = [], []
        i = 0
        while i < len(cur_idx):
            sib = cur_idx[i] ^ 1
            if i + 1 < len(cur_idx) and cur_idx[i + 1] == sib:
                l, r = cur_val[i], cur_val[i + 1]
                i += 1
            else:
                if i >= len(nodes) or ptr[i] >= len(nodes[i]):
                    return None
                s = nodes[i][ptr[i]]; ptr[i] += 1
                l, r = (cur_val[i], s) if cur_idx[i] & 1 == 0 else (s, cur_val[i])
            nxt_idx.append(sib >> 1); nxt_val.append(merge(l, r))
            i += 1
        cur_idx, cur_val = nxt_idx, nxt_val
    if len(cur_val) != 1 or any(ptr[k] != len(nodes[k]) for k in range(len(nodes))):
        return None
    return cur_val[0]


# ---- random coin
class Coin:
    def __init__(self, seed_elements):
        self.seed, self.counter = hash_elements(seed_elements), 0

    def reseed(self, digest):
        self.seed, self.counter = merge(self.seed, digest), 0

    def next(self):
        self.counter += 1
        return merge_with_int(self.seed, self.counter)

    def draw(self):
        for _ in range(1000):
            v = int.from_bytes(self.next()[:16], "little")
            if v < P:
                return v
        raise RuntimeError("coin exhausted")

    def draw_integers(self, count, domain_size, nonce):
        self.seed, self.counter = merge_with_int(self.seed, nonce), 0
        mask = domain_size - 1
        return [int.from_bytes(self.next()[:8], "little") & mask for _ in range(count)]


def context_elements():
    """TraceInfo (width 1, no auxiliary segment, length 8), the field modulus as two 8-byte halves, the proof options"""
    modulus = P.to_bytes(16, "little")
    return [(1 << 8) | 0, TRACE_LEN,
            int.from_bytes(modulus[:8], "little"), int.from_bytes(modulus[8:], "little"),
            (FIELD_EXTENSION_NONE << 16) | (FOLDING << 8) | REMAINDER_MAX_DEGREE, GRINDING, BLOWUP, NUM_QUERIES]


def context_bytes():
    """Context::write_into: trace info (width, aux segments, log2 length, metadata length), modulus, options"""
    return (bytes([1, 0, TRACE_LEN.bit_length() - 1]) + (0).to_bytes(2, "little") + bytes([16]) + P.to_bytes(16, "little") +
            bytes([NUM_QUERIES, BLOWUP, GRINDING, FIELD_EXTENSION_NONE, FOLDING, REMAINDER_MAX_DEGREE]))


def vint(n):
    """winter-utils variable-length usize: the number of trailing zero bits of the first byte gives the extra byte count"""
    bits = max(n.bit_length(), 1)
    nbytes = (bits + 6) // 7
    if nbytes > 8:
        return b"\0" + n.to_bytes(8, "little")
    return (((n << 1) | 1) << (nbytes - 1)).to_bytes(nbytes, "little")


def read_vint(buf, pos):
    first = buf[pos]
    if first == 0:
        return int.from_bytes(buf[pos + 1:pos + 9], "little"), pos + 9
    nbytes = (first & -first).bit_length()
    v = int.from_bytes(buf[pos:pos + nbytes], "little") >> nbytes
    return v, pos + nbytes


def trace_column(old, new):
    step = (new - old) * inv(TRACE_LEN - 1) % P
    return [(old + i * step) % P for i in range(TRACE_LEN)], step


def constraint_evaluation(x, cur, nxt, old, new, step, coef):
    """combined constraint value at x from a trace frame: transition / Z_t + boundary terms / their divisors"""
    g = root_of_unity(TRACE_LEN)
    last = pow(g, TRACE_LEN - 1, P)
    z_t = (pow(x, TRACE_LEN, P) - 1) * inv(x - last) % P
    t = coef[0] * (nxt - cur - step) % P * inv(z_t)
    b0 = coef[1] * (cur - old) % P * inv(x - 1)
    b1 = coef[2] * (cur - new) % P * inv(x - last)
    return (t + b0 + b1) % P


def _queries_bytes(values, batch_nodes, depth):
    vb = b"".join(values)
    ob = bytes([depth]) + vint(len(batch_nodes))
    for lst in batch_nodes:
        ob += bytes([len(lst)]) + b"".join(lst)
    return vint(len(vb)) + vb + vint(len(ob)) + ob


def prove(old, new, detail=None):
    """the bare STARK proof bytes (StarkBackend::prove_improvement, stark.rs:151-186)"""
    assert 0 <= old < new < 2**64
    col, step = trace_column(old, new)
    coin = Coin(context_elements() + [old, new])
    g = root_of_unity(TRACE_LEN)
    # 1. trace commitment
    t_poly = interpolate(col)
    t_lde = evaluate_on_coset(t_poly, LDE_SIZE, GENERATOR)
    t_tree = MerkleTree([hash_elements([v]) for v in t_lde])
    coin.reseed(t_tree.root)
    # 2. constraint evaluations on the 16-point coset
    coef = [coin.draw() for _ in range(3)]
    w_ce = root_of_unity(CE_SIZE)
    stride = LDE_SIZE // CE_SIZE
    ce = []
    for i in range(CE_SIZE):
        x = GENERATOR * pow(w_ce, i, P) % P
        ce.append(constraint_evaluation(x, t_lde[i * stride], t_lde[(i * stride + BLOWUP) % LDE_SIZE], old, new, step, coef))
    h_full = interpolate(ce, GENERATOR)
    assert all(c == 0 for c in h_full[TRACE_LEN:]), "composition degree exceeds one column"
    h_poly = h_full[:TRACE_LEN]
    h_lde = evaluate_on_coset(h_poly, LDE_SIZE, GENERATOR)
    h_tree = MerkleTree([hash_elements([v]) for v in h_lde])
    coin.reseed(h_tree.root)
    # 3. out-of-domain frame
    z = coin.draw()
    tz, tzg = horner(t_poly, z), horner(t_poly, z * g % P)
    coin.reseed(hash_elements([tz, tzg]))
    hz = horner(h_poly, z)
    coin.reseed(hash_elements([hz]))
    # 4. DEEP composition polynomial
    deep = [coin.draw() for _ in range(2)]
    t1 = syn_div([(t_poly[0] - tz) % P] + t_poly[1:], z)
    t2 = syn_div([(t_poly[0] - tzg) % P] + t_poly[1:], z * g % P)
    c1 = syn_div([(h_poly[0] - hz) % P] + h_poly[1:], z)
    d_poly = [(deep[0] * (a + b) + deep[1] * c) % P for a, b, c in zip(t1, t2, c1)] + [0]
    # 5. FRI: zero folding layers, the remainder is the polynomial itself
    remainder = d_poly[:LDE_SIZE // BLOWUP]
    rem_commit = hash_elements(remainder)
    coin.reseed(rem_commit)
    # 6. queries
    positions = sorted(set(coin.draw_integers(NUM_QUERIES, LDE_SIZE, 0)))
    depth = LDE_SIZE.bit_length() - 1
    tq = _queries_bytes([el_bytes(t_lde[p]) for p in positions], t_tree.prove_batch(positions), depth)
    cq = _queries_bytes([el_bytes(h_lde[p]) for p in positions], h_tree.prove_batch(positions), depth)
    # 7. serialization
    commitments = t_tree.root + h_tree.root + rem_commit
    ood_trace = bytes([2]) + el_bytes(tz) + el_bytes(tzg)
    ood_eval = el_bytes(hz)
    rem_bytes = b"".join(el_bytes(c) for c in remainder)
    out = (context_bytes() + bytes([len(positions)]) + len(commitments).to_bytes(2, "little") + commitments + tq + cq +
           len(ood_trace).to_bytes(2, "little") + ood_trace + len(ood_eval).to_bytes(2, "little") + ood_eval +
           bytes([0]) + len(rem_bytes).to_bytes(2, "little") + rem_bytes + bytes([1]) +
           (0).to_bytes(8, "little") + bytes([0]))
    if detail is not None:
        detail.update(coef=coef, z=z, tz=tz, tzg=tzg, hz=hz, deep=deep, positions=positions, t_poly=t_poly, h_poly=h_poly,
                      remainder=remainder, t_root=t_tree.root, h_root=h_tree.root, t_lde=t_lde, h_lde=h_lde)
    return out


class _Reader:
    def __init__(self, b):
        self.b, self.p = bytes(b), 0

    def take(self, n):
        if self.p + n > len(self.b):
            raise ValueError("truncated")
        v = self.b[self.p:self.p + n]
        self.p += n
        return v

    def u8(self):
        return self.take(1)[0]

    def u16(self):
        return int.from_bytes(self.take(2), "little")

    def vint(self):
        if self.p >= len(self.b):
            raise ValueError("truncated")
        first = self.b[self.p]
        need = 9 if first == 0 else (first & -first).bit_length()
        if self.p + need > len(self.b):
            raise ValueError("truncated")
        v, self.p = read_vint(self.b, self.p)
        return v

    def element(self):
        v = int.from_bytes(self.take(16), "little")
        if v >= P:
            raise ValueError("non-canonical element")
        return v


def _read_queries(r, count):
    nvals = r.vint()
    if nvals != 16 * count:
        raise ValueError("query value count")
    values = [r.element() for _ in range(count)]
    olen = r.vint()
    end = r.p + olen
    depth = r.u8()
    nlists = r.vint()
    nodes = []
    for _ in range(nlists):
        k = r.u8()
        nodes.append([r.take(32) for _ in range(k)])
    if r.p != end:
        raise ValueError("opening proof length")
    return values, nodes, depth


def verify(proof, old, new):
    """winterfell::verify restated for this AIR (stark.rs:190-211): True iff the proof is accepted"""
    try:
        if not (0 <= old < 2**64 and 0 <= new < 2**64):
            return False
        r = _Reader(proof)
        if r.take(len(context_bytes())) != context_bytes():
            return False
        nq = r.u8()
        if r.u16() != 96:
            return False
        t_root, h_root, rem_commit = r.take(32), r.take(32), r.take(32)
        if not 1 <= nq <= NUM_QUERIES:
            return False
        t_vals, t_nodes, t_depth = _read_queries(r, nq)
        h_vals, h_nodes, h_depth = _read_queries(r, nq)
        if r.u16() != 33 or r.u8() != 2:
            return False
        tz, tzg = r.element(), r.element()
        if r.u16() != 16:
            return False
        hz = r.element()
        if r.u8() != 0 or r.u16() != 128:
            return False
        remainder = [r.element() for _ in range(8)]
        if r.u8() != 1 or r.take(8) != bytes(8) or r.u8() != 0 or r.p != len(r.b):
            return False
    except ValueError:
        return False
    step = (new - old) * inv(TRACE_LEN - 1) % P
    g = root_of_unity(TRACE_LEN)
    coin = Coin(context_elements() + [old % P, new % P])
    coin.reseed(t_root)
    coef = [coin.draw() for _ in range(3)]
    coin.reseed(h_root)
    z = coin.draw()
    coin.reseed(hash_elements([tz, tzg]))
    # out-of-domain consistency: the composition value implied by the trace frame must equal the one sent
    if constraint_evaluation(z, tz, tzg, old, new, step, coef) != hz:
        return False
    coin.reseed(hash_elements([hz]))
    deep = [coin.draw() for _ in range(2)]
    if hash_elements(remainder) != rem_commit:
        return False
    coin.reseed(rem_commit)
    positions = sorted(set(coin.draw_integers(NUM_QUERIES, LDE_SIZE, 0)))
    if len(positions) != nq:
        return False
    depth = LDE_SIZE.bit_length() - 1
    if t_depth != depth or h_depth != depth:
        return False
    if batch_root([hash_elements([v]) for v in t_vals], positions, t_nodes, depth) != t_root:
        return False
    if batch_root([hash_elements([v]) for v in h_vals], positions, h_nodes, depth) != h_root:
        return False
    # DEEP composition at every query must lie on the committed remainder polynomial (degree < 8; FRI has no layers)
    w = root_of_unity(LDE_SIZE)
    zg = z * g % P
    for pos, tv, hv in zip(positions, t_vals, h_vals):
        x = GENERATOR * pow(w, pos, P) % P
        d = (deep[0] * ((tv - tz) * inv(x - z) + (tv - tzg) * inv(x - zg)) + deep[1] * (hv - hz) * inv(x - z)) % P
        if d != horner(remainder, x):
            return False
    return True


# ---- libzkp framing
def commit_improvement(old, new):
    """utils/commitment.rs:38-50"""
    return hashlib.sha256(b"libzkp_improvement_v1" + old.to_bytes(8, "little") + new.to_bytes(8, "little")).digest()


def prove_improvement(old, new):
    """proof/improvement_proof.rs:10-35: envelope [2][5][u32 len][u32 32][old || new || stark][commitment]"""
    if new <= old:
        raise ValueError("new value must be greater than old value")
    payload = old.to_bytes(8, "little") + new.to_bytes(8, "little") + prove(old, new)
    return bytes([2, 5]) + len(payload).to_bytes(4, "little") + (32).to_bytes(4, "little") + payload + commit_improvement(old, new)


def verify_improvement(envelope, old):
    """proof/improvement_proof.rs:37-68"""
    b = bytes(envelope)
    if len(b) < 10 or b[0] != 2 or b[1] != 5:
        return False
    plen, clen = int.from_bytes(b[2:6], "little"), int.from_bytes(b[6:10], "little")
    if len(b) != 10 + plen + clen or plen < 16 or clen != 32:
        return False
    payload, commitment = b[10:10 + plen], b[10 + plen:]
    stored_old, new = int.from_bytes(payload[:8], "little"), int.from_bytes(payload[8:16], "little")
    if stored_old != old or new <= old or commitment != commit_improvement(old, new):
        return False
    return verify(payload[16:], old, new)
