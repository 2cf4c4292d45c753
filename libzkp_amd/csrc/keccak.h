// Keccak-f[1600] in registers (25 x u64 per lane), the project's SHAKE256 randomness tape, and the
// STROBE-128 / Merlin transcript operations bulletproofs uses for Fiat-Shamir.
// Replaces merlin::Transcript as used at /root/reference/src/backend/bulletproofs.rs:137,149,343,395,642.
//
// The STROBE state of one proof lives in a word-strided memory image (LDS on the GPU: word i of lane t at
// base[i*stride + t], conflict-free; a plain array with stride 1 in the host emulation) because absorb /
// squeeze positions are data-dependent byte offsets; keccak-f itself runs on registers.
#pragma once
#include "zkp_common.h"

namespace zkp {

ZKP_HD inline uint64_t rol64(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }

ZKP_HD inline void keccak_f1600(uint64_t a[25]) {
    const uint64_t RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL,
        0x000000000000808BULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
        0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
        0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    for (int r = 0; r < 24; r++) {
        uint64_t c0 = a[0] ^ a[5] ^ a[10] ^ a[15] ^ a[20];
        uint64_t c1 = a[1] ^ a[6] ^ a[11] ^ a[16] ^ a[21];
        uint64_t c2 = a[2] ^ a[7] ^ a[12] ^ a[17] ^ a[22];
        uint64_t c3 = a[3] ^ a[8] ^ a[13] ^ a[18] ^ a[23];
        uint64_t c4 = a[4] ^ a[9] ^ a[14] ^ a[19] ^ a[24];
        const uint64_t d0 = c4 ^ rol64(c1, 1), d1 = c0 ^ rol64(c2, 1), d2 = c1 ^ rol64(c3, 1), d3 = c2 ^ rol64(c4, 1), d4 = c3 ^ rol64(c0, 1);
        ZKP_UNROLL for (int y = 0; y < 25; y += 5) { a[y] ^= d0; a[y + 1] ^= d1; a[y + 2] ^= d2; a[y + 3] ^= d3; a[y + 4] ^= d4; }
        // rho + pi
        uint64_t b[25];
        b[0] = a[0];
        b[10] = rol64(a[1], 1);   b[20] = rol64(a[2], 62);  b[5] = rol64(a[3], 28);   b[15] = rol64(a[4], 27);
        b[16] = rol64(a[5], 36);  b[1] = rol64(a[6], 44);   b[11] = rol64(a[7], 6);   b[21] = rol64(a[8], 55);  b[6] = rol64(a[9], 20);
        b[7] = rol64(a[10], 3);   b[17] = rol64(a[11], 10); b[2] = rol64(a[12], 43);  b[12] = rol64(a[13], 25); b[22] = rol64(a[14], 39);
        b[23] = rol64(a[15], 41); b[8] = rol64(a[16], 45);  b[18] = rol64(a[17], 15); b[3] = rol64(a[18], 21);  b[13] = rol64(a[19], 8);
        b[14] = rol64(a[20], 18); b[24] = rol64(a[21], 2);  b[9] = rol64(a[22], 61);  b[19] = rol64(a[23], 56); b[4] = rol64(a[24], 14);
        // chi
        ZKP_UNROLL for (int y = 0; y < 25; y += 5) {
            a[y] = b[y] ^ (~b[y + 1] & b[y + 2]);
            a[y + 1] = b[y + 1] ^ (~b[y + 2] & b[y + 3]);
            a[y + 2] = b[y + 2] ^ (~b[y + 3] & b[y + 4]);
            a[y + 3] = b[y + 3] ^ (~b[y + 4] & b[y]);
            a[y + 4] = b[y + 4] ^ (~b[y] & b[y + 1]);
        }
        a[0] ^= RC[r];
    }
}

// Project-defined randomness tape (oracle/py/bulletproofs.py header):
//   draw64(seed, proof_idx, slot) = SHAKE256("libzkp-amd/tape/v1" || seed[32] || u32le(proof_idx) || u32le(slot))[0:64]
// 58 input bytes -> a single permutation; output = first 16 words of the state.
ZKP_HD inline void tape_draw64(uint32_t out[16], const uint32_t seed[8], uint32_t proof_idx, uint32_t slot) {
    // message bytes: 18-byte domain, 32-byte seed, two u32 -> packed little-endian into 64-bit lanes
    uint8_t m[64];
    const char dom[19] = "libzkp-amd/tape/v1";
    ZKP_UNROLL for (int i = 0; i < 18; i++) m[i] = (uint8_t)dom[i];
    ZKP_UNROLL for (int i = 0; i < 32; i++) m[18 + i] = (uint8_t)(seed[i >> 2] >> (8 * (i & 3)));
    ZKP_UNROLL for (int i = 0; i < 4; i++) { m[50 + i] = (uint8_t)(proof_idx >> (8 * i)); m[54 + i] = (uint8_t)(slot >> (8 * i)); }
    m[58] = 0x1F;  // SHAKE domain separation + first pad bit
    ZKP_UNROLL for (int i = 59; i < 64; i++) m[i] = 0;
    uint64_t a[25];
    ZKP_UNROLL for (int i = 0; i < 8; i++) {
        uint64_t w = 0;
        ZKP_UNROLL for (int k = 0; k < 8; k++) w |= (uint64_t)m[8 * i + k] << (8 * k);
        a[i] = w;
    }
    ZKP_UNROLL for (int i = 8; i < 25; i++) a[i] = 0;
    a[16] ^= 0x8000000000000000ULL;  // last byte of the 136-byte rate
    keccak_f1600(a);
    ZKP_UNROLL for (int i = 0; i < 8; i++) { out[2 * i] = (uint32_t)a[i]; out[2 * i + 1] = (uint32_t)(a[i] >> 32); }
}

// ---------------------------------------------------------------- STROBE-128 / Merlin
struct Strobe {
    uint32_t* base;     // 50 state words, word i at base[i * stride]
    uint32_t stride;
    uint32_t pos, pos_begin;
};
enum { STROBE_R = 166, STROBE_FLAG_I = 1, STROBE_FLAG_A = 2, STROBE_FLAG_C = 4, STROBE_FLAG_M = 16, STROBE_FLAG_K = 32 };

ZKP_HD inline void strobe_xor_byte(Strobe& s, uint32_t pos, uint32_t byte) { s.base[(pos >> 2) * s.stride] ^= byte << (8 * (pos & 3)); }

ZKP_HD inline void strobe_permute(Strobe& s) {
    uint64_t a[25];
    ZKP_UNROLL for (int i = 0; i < 25; i++) a[i] = (uint64_t)s.base[(2 * i) * s.stride] | ((uint64_t)s.base[(2 * i + 1) * s.stride] << 32);
    keccak_f1600(a);
    ZKP_UNROLL for (int i = 0; i < 25; i++) { s.base[(2 * i) * s.stride] = (uint32_t)a[i]; s.base[(2 * i + 1) * s.stride] = (uint32_t)(a[i] >> 32); }
}

ZKP_HD inline void strobe_run_f(Strobe& s) {
    strobe_xor_byte(s, s.pos, s.pos_begin);
    strobe_xor_byte(s, s.pos + 1, 0x04);
    strobe_xor_byte(s, STROBE_R + 1, 0x80);
    strobe_permute(s);
    s.pos = 0; s.pos_begin = 0;
}
ZKP_HD inline void strobe_absorb_byte(Strobe& s, uint32_t byte) {
    strobe_xor_byte(s, s.pos, byte);
    if (++s.pos == STROBE_R) strobe_run_f(s);
}
ZKP_HD inline uint32_t strobe_squeeze_byte(Strobe& s) {
    uint32_t& w = s.base[(s.pos >> 2) * s.stride];
    const uint32_t sh = 8 * (s.pos & 3);
    const uint32_t byte = (w >> sh) & 0xffu;
    w &= ~(0xffu << sh);
    if (++s.pos == STROBE_R) strobe_run_f(s);
    return byte;
}
ZKP_HD inline void strobe_begin_op(Strobe& s, uint32_t flags) {
    const uint32_t old_begin = s.pos_begin;
    s.pos_begin = s.pos + 1;
    strobe_absorb_byte(s, old_begin);
    strobe_absorb_byte(s, flags);
    if ((flags & (STROBE_FLAG_C | STROBE_FLAG_K)) && s.pos != 0) strobe_run_f(s);
}

// Strobe128::new(b"Merlin v1.0") then Transcript::new(label): the caller zeroes nothing -- this sets all 50 words.
ZKP_HD inline void merlin_absorb_cstr(Strobe& s, const char* str, uint32_t len) { for (uint32_t i = 0; i < len; i++) strobe_absorb_byte(s, (uint8_t)str[i]); }
ZKP_HD inline void merlin_absorb_u32le(Strobe& s, uint32_t x) { for (int i = 0; i < 4; i++) strobe_absorb_byte(s, (x >> (8 * i)) & 0xffu); }

ZKP_HD inline void merlin_begin_append(Strobe& s, const char* label, uint32_t label_len, uint32_t msg_len) {
    strobe_begin_op(s, STROBE_FLAG_M | STROBE_FLAG_A); merlin_absorb_cstr(s, label, label_len);
    merlin_absorb_u32le(s, msg_len);                                   // meta_ad(len, more = true)
    strobe_begin_op(s, STROBE_FLAG_A);
}
ZKP_HD inline void merlin_append_bytes(Strobe& s, const char* label, uint32_t label_len, const char* msg, uint32_t msg_len) {
    merlin_begin_append(s, label, label_len, msg_len); merlin_absorb_cstr(s, msg, msg_len);
}
ZKP_HD inline void merlin_append_words(Strobe& s, const char* label, uint32_t label_len, const uint32_t* w, uint32_t nwords) {
    merlin_begin_append(s, label, label_len, 4 * nwords);
    for (uint32_t i = 0; i < nwords; i++) merlin_absorb_u32le(s, w[i]);
}
ZKP_HD inline void merlin_append_u64(Strobe& s, const char* label, uint32_t label_len, uint64_t x) {
    const uint32_t w[2] = {(uint32_t)x, (uint32_t)(x >> 32)};
    merlin_append_words(s, label, label_len, w, 2);
}
ZKP_HD inline void merlin_challenge_words(Strobe& s, const char* label, uint32_t label_len, uint32_t* out, uint32_t nwords) {
    strobe_begin_op(s, STROBE_FLAG_M | STROBE_FLAG_A); merlin_absorb_cstr(s, label, label_len);
    merlin_absorb_u32le(s, 4 * nwords);
    strobe_begin_op(s, STROBE_FLAG_I | STROBE_FLAG_A | STROBE_FLAG_C);
    for (uint32_t i = 0; i < nwords; i++) {
        uint32_t w = 0;
        for (int k = 0; k < 4; k++) w |= strobe_squeeze_byte(s) << (8 * k);
        out[i] = w;
    }
}
ZKP_HD inline void merlin_init(Strobe& s, const char* label, uint32_t label_len) {
    for (int i = 0; i < 50; i++) s.base[i * s.stride] = 0;
    // bytes 1, R+2, 1, 0, 1, 96, "STROBEv1.0.2"
    const uint8_t hdr[18] = {1, STROBE_R + 2, 1, 0, 1, 96, 'S', 'T', 'R', 'O', 'B', 'E', 'v', '1', '.', '0', '.', '2'};
    for (uint32_t i = 0; i < 18; i++) strobe_xor_byte(s, i, hdr[i]);
    strobe_permute(s);
    s.pos = 0; s.pos_begin = 0;
    strobe_begin_op(s, STROBE_FLAG_M | STROBE_FLAG_A); merlin_absorb_cstr(s, "Merlin v1.0", 11);
    merlin_append_bytes(s, "dom-sep", 7, label, label_len);
}

}  // namespace zkp
