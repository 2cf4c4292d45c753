/* ORACLE (test infrastructure only; never linked into the product library).
 *
 * CPU restatement, in plain C, of the arithmetic the reference reaches through the third-party crates
 * curve25519-dalek ^4.1, merlin ^3.0 and bulletproofs ^5.0 (Cargo.toml:12-14; not vendored under
 * /root/reference, not version-pinned, not buildable here: no cargo/rustc).  PARITY UNPINNED at the
 * proof-byte level (see oracle/py/bulletproofs.py header); this C code is pinned to the Python model
 * (tests/test_oracle_c.py), which is itself pinned to hashlib / libsodium / the Merlin KAT.
 *
 * Field GF(2^255-19) in five 51-bit limbs, scalars mod l in four 64-bit limbs (Montgomery),
 * edwards25519 extended coordinates, ristretto255 encode/decode/one-way map (RFC 9496).
 */
#ifndef ZKP_ORACLE_CURVE_H
#define ZKP_ORACLE_CURVE_H
#include <stdint.h>
#include <stddef.h>

typedef uint64_t fe[5];
typedef struct { fe X, Y, Z, T; } ge;
typedef struct { uint64_t v[4]; } sc;

void oracle_curve_init(void);

/* field */
void fe_frombytes(fe h, const uint8_t s[32]);
void fe_tobytes(uint8_t s[32], const fe h);
void fe_mul(fe h, const fe f, const fe g);
void fe_sq(fe h, const fe f);
void fe_add(fe h, const fe f, const fe g);
void fe_sub(fe h, const fe f, const fe g);
void fe_neg(fe h, const fe f);
void fe_copy(fe h, const fe f);
int fe_isneg(const fe f);
int fe_iszero(const fe f);
int fe_eq(const fe f, const fe g);
int fe_sqrt_ratio_m1(fe r, const fe u, const fe v);

/* group */
extern ge GE_IDENTITY, GE_BASEPOINT;
void ge_add(ge* r, const ge* p, const ge* q);
void ge_sub(ge* r, const ge* p, const ge* q);
void ge_dbl(ge* r, const ge* p);
void ge_neg(ge* r, const ge* p);
int ge_eq(const ge* p, const ge* q); /* ristretto equality */
int ge_is_identity(const ge* p);
void ge_encode(uint8_t s[32], const ge* p);
int ge_decode(ge* p, const uint8_t s[32]);
void ge_from_uniform(ge* p, const uint8_t b[64]);
/* vartime Straus, width-5 NAF (what dalek's vartime_multiscalar_mul does below ~190 points) */
void ge_msm_vartime(ge* r, size_t n, const sc* scalars, const ge* points);

/* scalars mod l */
extern const sc SC_ZERO, SC_ONE;
void sc_from_u64(sc* r, uint64_t x);
void sc_from_bytes_mod_order(sc* r, const uint8_t b[32]);
void sc_from_bytes_mod_order_wide(sc* r, const uint8_t b[64]);
int sc_from_canonical_bytes(sc* r, const uint8_t b[32]);
void sc_tobytes(uint8_t b[32], const sc* a);
void sc_add(sc* r, const sc* a, const sc* b);
void sc_sub(sc* r, const sc* a, const sc* b);
void sc_neg(sc* r, const sc* a);
void sc_mul(sc* r, const sc* a, const sc* b);
void sc_muladd(sc* r, const sc* a, const sc* b, const sc* c); /* a*b + c */
void sc_invert(sc* r, const sc* a);
int sc_iszero(const sc* a);

/* cumulative operation counters (instrumented counts for DESIGN.md / the VALU roofline) */
extern uint64_t oracle_fe_mul_count, oracle_sc_mul_count;
#endif
