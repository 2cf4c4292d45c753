// SHA-256 (FIPS 180-4) usable on the device and on the host: libzkp's binding commitments
// (/root/reference/src/utils/commitment.rs:38-50, /root/reference/src/backend/bulletproofs.rs:430-434).
#pragma once
#include "zkp_common.h"

namespace zkp {

ZKP_HD constexpr uint32_t sha256_k(int i) {
    constexpr uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
        0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
        0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
        0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
        0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
        0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    return K[i];
}
ZKP_HD inline uint32_t sha_rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
ZKP_HD_NOINLINE inline void sha256_block(uint32_t h[8], const uint8_t blk[64]) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++) w[i] = ((uint32_t)blk[4 * i] << 24) | ((uint32_t)blk[4 * i + 1] << 16) | ((uint32_t)blk[4 * i + 2] << 8) | blk[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        const uint32_t s0 = sha_rotr(w[i - 15], 7) ^ sha_rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = sha_rotr(w[i - 2], 17) ^ sha_rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
        const uint32_t t1 = hh + (sha_rotr(e, 6) ^ sha_rotr(e, 11) ^ sha_rotr(e, 25)) + ((e & f) ^ (~e & g)) + sha256_k(i) + w[i];
        const uint32_t t2 = (sha_rotr(a, 2) ^ sha_rotr(a, 13) ^ sha_rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
// One padded block given as sixteen big-endian words, compressed from the initial state: fully unrolled with a rolling 16-word
// schedule, so everything stays in registers (the STARK prover's binding commitment: a fixed 37-byte message, no scratch memory).
ZKP_HD inline void sha256_one_block_words(uint32_t h[8], const uint32_t m[16]) {
    uint32_t w[16];
    ZKP_UNROLL for (int i = 0; i < 16; i++) w[i] = m[i];
    uint32_t a = 0x6a09e667u, b = 0xbb67ae85u, c = 0x3c6ef372u, d = 0xa54ff53au, e = 0x510e527fu, f = 0x9b05688cu, g = 0x1f83d9abu, hh = 0x5be0cd19u;
    ZKP_UNROLL for (int i = 0; i < 64; i++) {
        if (i >= 16) {
            const uint32_t w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
            const uint32_t s0 = sha_rotr(w15, 7) ^ sha_rotr(w15, 18) ^ (w15 >> 3), s1 = sha_rotr(w2, 17) ^ sha_rotr(w2, 19) ^ (w2 >> 10);
            w[i & 15] = w[i & 15] + s0 + w[(i + 9) & 15] + s1;
        }
        const uint32_t t1 = hh + (sha_rotr(e, 6) ^ sha_rotr(e, 11) ^ sha_rotr(e, 25)) + ((e & f) ^ (~e & g)) + sha256_k(i) + w[i & 15];
        const uint32_t t2 = (sha_rotr(a, 2) ^ sha_rotr(a, 13) ^ sha_rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] = 0x6a09e667u + a; h[1] = 0xbb67ae85u + b; h[2] = 0x3c6ef372u + c; h[3] = 0xa54ff53au + d;
    h[4] = 0x510e527fu + e; h[5] = 0x9b05688cu + f; h[6] = 0x1f83d9abu + g; h[7] = 0x5be0cd19u + hh;
}
// digest of `len` bytes at `in`
ZKP_HD_NOINLINE inline void sha256_bytes(uint8_t out[32], const uint8_t* in, uint64_t len) {
    uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    uint64_t off = 0;
    for (; off + 64 <= len; off += 64) sha256_block(h, in + off);
    uint8_t blk[64];
    const uint32_t rem = (uint32_t)(len - off);
    for (uint32_t i = 0; i < 64; i++) blk[i] = i < rem ? in[off + i] : 0;
    blk[rem] = 0x80;
    if (rem >= 56) { sha256_block(h, blk); for (int i = 0; i < 64; i++) blk[i] = 0; }
    const uint64_t bits = len * 8;
    for (int i = 0; i < 8; i++) blk[56 + i] = (uint8_t)(bits >> (8 * (7 - i)));
    sha256_block(h, blk);
    for (int k = 0; k < 8; k++) { out[4 * k] = (uint8_t)(h[k] >> 24); out[4 * k + 1] = (uint8_t)(h[k] >> 16); out[4 * k + 2] = (uint8_t)(h[k] >> 8); out[4 * k + 3] = (uint8_t)h[k]; }
}

}  // namespace zkp
