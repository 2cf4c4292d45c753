/* ORACLE (test infrastructure only; never linked into or called by the product path).
 *
 * BN254 field and group arithmetic in plain C (4 x 64-bit Montgomery limbs), restating what the reference reaches
 * through the third-party crates ark-bn254 / ark-ff / ark-ec ^0.5 (Cargo.toml:16-21; not vendored, not pinned, not
 * buildable here).  PARITY UNPINNED at the proof-byte level (oracle/py/groth16.py header); this C code is pinned to
 * the Python bigint model (tests/test_oracle_c_snark.py), which is pinned by bilinearity / group-order checks and the
 * reference's own MiMC constant recipe.
 */
#ifndef ZKP_ORACLE_BN254_H
#define ZKP_ORACLE_BN254_H
#include <stdint.h>
#include <stddef.h>

typedef struct { uint64_t v[4]; } fp;           /* Montgomery form; which modulus is a matter of the function family */
typedef struct { fp c0, c1; } fq2;
typedef struct { fp x, y; int inf; } g1a;      /* affine */
typedef struct { fp X, Y, Z; } g1j;            /* Jacobian, Z = 0: infinity */
typedef struct { fq2 x, y; int inf; } g2a;
typedef struct { fq2 X, Y, Z; } g2j;

void bn254_init(void);

/* base field Fq */
void fq_add(fp* r, const fp* a, const fp* b);
void fq_sub(fp* r, const fp* a, const fp* b);
void fq_neg(fp* r, const fp* a);
void fq_mul(fp* r, const fp* a, const fp* b);
void fq_inv(fp* r, const fp* a);
int fq_is_zero(const fp* a);
int fq_from_bytes(fp* r, const uint8_t b[32]);       /* canonical LE -> Montgomery; 0 if >= p */
void fq_to_bytes(uint8_t b[32], const fp* a);
int fq_lex_larger(const fp* a);                       /* a > -a as canonical integers (ark's sign flag) */

/* scalar field Fr */
extern fp FR_ONE, FR_ZERO;
void fr_add(fp* r, const fp* a, const fp* b);
void fr_sub(fp* r, const fp* a, const fp* b);
void fr_neg(fp* r, const fp* a);
void fr_mul(fp* r, const fp* a, const fp* b);
void fr_inv(fp* r, const fp* a);
void fr_pow_u64(fp* r, const fp* a, uint64_t e);
void fr_from_u64(fp* r, uint64_t x);
void fr_from_bytes_mod_order(fp* r, const uint8_t b[32]);        /* from_le_bytes_mod_order */
void fr_from_bytes_wide(fp* r, const uint8_t b[64]);
void fr_to_bytes(uint8_t b[32], const fp* a);
void fr_to_raw(uint64_t w[4], const fp* a);                       /* canonical integer limbs */
void fr_root_of_unity(fp* r, uint32_t m);                         /* 5^((r-1)/m) */

/* groups */
void g1j_set_inf(g1j* r);
void g1j_from_affine(g1j* r, const g1a* p);
void g1j_add(g1j* r, const g1j* p, const g1j* q);
void g1j_madd(g1j* r, const g1j* p, const g1a* q);
void g1j_dbl(g1j* r, const g1j* p);
void g1j_neg(g1j* r, const g1j* p);
void g1j_mul(g1j* r, const g1j* p, const fp* scalar_fr);
void g1j_to_affine(g1a* r, const g1j* p);
int g1_parse(g1a* r, const uint8_t b[64]);                        /* ark uncompressed; 0 if malformed */
void g1_serialize(uint8_t b[64], const g1a* p);
void g1_msm(g1j* r, size_t n, const fp* scalars_fr, const g1a* bases);      /* bucket method (ark's VariableBaseMSM shape) */

void g2j_set_inf(g2j* r);
void g2j_from_affine(g2j* r, const g2a* p);
void g2j_add(g2j* r, const g2j* p, const g2j* q);
void g2j_madd(g2j* r, const g2j* p, const g2a* q);
void g2j_dbl(g2j* r, const g2j* p);
void g2j_mul(g2j* r, const g2j* p, const fp* scalar_fr);
void g2j_to_affine(g2a* r, const g2j* p);
int g2_parse(g2a* r, const uint8_t b[128]);
void g2_serialize(uint8_t b[128], const g2a* p);
void g2_msm(g2j* r, size_t n, const fp* scalars_fr, const g2a* bases);
#endif
