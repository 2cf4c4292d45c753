/* ORACLE (test infrastructure only). Keccak-f[1600], SHAKE256, SHA3-512, STROBE-128/Merlin, SHA-256.
 * Restates merlin ^3.0 (Cargo.toml:14, not vendored); call sites /root/reference/src/backend/bulletproofs.rs:137,149,343,395,642. */
#ifndef ZKP_ORACLE_TRANSCRIPT_H
#define ZKP_ORACLE_TRANSCRIPT_H
#include <stdint.h>
#include <stddef.h>

void keccak_f1600(uint64_t st[25]);
void shake256(uint8_t* out, size_t outlen, const uint8_t* in, size_t inlen);
void sha3_512(uint8_t out[64], const uint8_t* in, size_t inlen);
void sha256(uint8_t out[32], const uint8_t* in, size_t inlen);

typedef struct {
    union { uint64_t w[25]; uint8_t b[200]; } st;
    uint8_t pos, pos_begin, cur_flags;
} merlin_t;

void merlin_init(merlin_t* t, const char* label);
void merlin_append(merlin_t* t, const char* label, const uint8_t* msg, uint32_t len);
void merlin_append_u64(merlin_t* t, const char* label, uint64_t x);
void merlin_challenge(merlin_t* t, const char* label, uint8_t* out, uint32_t len);
#endif
