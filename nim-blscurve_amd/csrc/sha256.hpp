// SHA-256 (FIPS 180-4) for device lanes: blinding-scalar hash chain
// (blst_min_pubkey_sig_core.nim:497-507,551-554; sha256_abi.nim:52-74) and expand_message_xmd.
#pragma once
#include "fp.hpp"

namespace bls {

BLS_CONST uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

BLS_HD uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

// one compression; w[16] = big-endian words of the block (clobbered)
BLS_HD void sha256_compress_core(uint32_t (&h)[8], uint32_t (&w)[16]) {
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        if (i >= 16) {
            uint32_t w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
            uint32_t s0 = rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3);
            uint32_t s1 = rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10);
            w[i & 15] = w[i & 15] + s0 + w[(i + 9) & 15] + s1;
        }
        uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25);
        uint32_t ch = (e & f) ^ (~e & g);
        uint32_t t1 = hh + S1 + ch + SHA_K[i] + w[i & 15];
        uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22);
        uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        uint32_t t2 = S0 + mj;
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
#if defined(__HIP_DEVICE_COMPILE__)
// ONE out-of-line copy of the 64 rounds, state and block in registers (24 scalar arguments, result in v0..v7):
// inlined at every block of expand_message_xmd the rounds made hash_to_field 280 KB of straight-line code, which
// k_hash_map spent a fifth of its cycles fetching.
typedef uint32_t bls_u32x8 __attribute__((ext_vector_type(8)));
__device__ __noinline__ bls_u32x8 sha256_compress_regs(uint32_t h0, uint32_t h1, uint32_t h2, uint32_t h3, uint32_t h4, uint32_t h5, uint32_t h6,
                                                       uint32_t h7, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t w4, uint32_t w5,
                                                       uint32_t w6, uint32_t w7, uint32_t w8, uint32_t w9, uint32_t w10, uint32_t w11, uint32_t w12,
                                                       uint32_t w13, uint32_t w14, uint32_t w15) {
    uint32_t h[8] = {h0, h1, h2, h3, h4, h5, h6, h7};
    uint32_t w[16] = {w0, w1, w2, w3, w4, w5, w6, w7, w8, w9, w10, w11, w12, w13, w14, w15};
    sha256_compress_core(h, w);
    bls_u32x8 r = {h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]};
    return r;
}
__device__ __forceinline__ void sha256_compress(uint32_t (&h)[8], uint32_t (&w)[16]) {
    bls_u32x8 r = sha256_compress_regs(h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8], w[9],
                                       w[10], w[11], w[12], w[13], w[14], w[15]);
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] = r[i];
}
#else
BLS_HD void sha256_compress(uint32_t (&h)[8], uint32_t (&w)[16]) { sha256_compress_core(h, w); }
#endif

BLS_HD void sha256_init(uint32_t (&h)[8]) {
    h[0] = 0x6a09e667; h[1] = 0xbb67ae85; h[2] = 0x3c6ef372; h[3] = 0xa54ff53a;
    h[4] = 0x510e527f; h[5] = 0x9b05688c; h[6] = 0x1f83d9ab; h[7] = 0x5be0cd19;
}

// Streaming context over a word buffer; bytes are appended big-endian into w[].
struct sha256_ctx {
    uint32_t h[8];
    uint32_t w[16];
    uint32_t fill;      // bytes currently in w
    uint32_t total;     // total bytes absorbed
};

BLS_HD void sha256_begin(sha256_ctx& c) {
    sha256_init(c.h);
#pragma unroll
    for (int i = 0; i < 16; i++) c.w[i] = 0;
    c.fill = 0;
    c.total = 0;
}

BLS_MID void sha256_put(sha256_ctx& c, uint8_t byte) {
    uint32_t idx = c.fill >> 2, sh = 24 - 8 * (c.fill & 3);
    // static-index update keeps w[] in registers
#pragma unroll
    for (int i = 0; i < 16; i++)
        if ((uint32_t)i == idx) c.w[i] |= (uint32_t)byte << sh;
    c.fill++;
    c.total++;
    if (c.fill == 64) {
        sha256_compress(c.h, c.w);
#pragma unroll
        for (int i = 0; i < 16; i++) c.w[i] = 0;
        c.fill = 0;
    }
}

BLS_HD void sha256_update(sha256_ctx& c, const uint8_t* p, uint32_t n) {
    for (uint32_t i = 0; i < n; i++) sha256_put(c, p[i]);
}

// absorb one block of 64 zero bytes into an EMPTY buffer (Z_pad of expand_message_xmd): one compression, no byte-wise buffering
BLS_HD void sha256_zero_block(sha256_ctx& c) {
#pragma unroll
    for (int i = 0; i < 16; i++) c.w[i] = 0;
    sha256_compress(c.h, c.w);
    c.total += 64;
}

// absorb a digest given as 8 big-endian words
BLS_HD void sha256_update_words(sha256_ctx& c, const uint32_t (&d)[8]) {
    if (c.fill == 0) {                                  // word-aligned (every use in expand_message_xmd): the words go in whole
#pragma unroll
        for (int i = 0; i < 8; i++) c.w[i] = d[i];
        c.fill = 32;
        c.total += 32;
        return;
    }
    for (int i = 0; i < 8; i++) {
        sha256_put(c, (uint8_t)(d[i] >> 24));
        sha256_put(c, (uint8_t)(d[i] >> 16));
        sha256_put(c, (uint8_t)(d[i] >> 8));
        sha256_put(c, (uint8_t)d[i]);
    }
}

BLS_HD void sha256_end(sha256_ctx& c, uint32_t (&out)[8]) {
    uint32_t bits = c.total * 8;
    sha256_put(c, 0x80);
    while (c.fill != 56) sha256_put(c, 0);
    c.w[14] = 0;
    c.w[15] = bits;
    sha256_compress(c.h, c.w);
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = c.h[i];
}

// digest of exactly 32 bytes given as 8 BE words (the blinding chain's seed <- SHA256(seed))
BLS_HD void sha256_of_digest(const uint32_t (&in)[8], uint32_t (&out)[8]) {
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = in[i];
    w[8] = 0x80000000u;
#pragma unroll
    for (int i = 9; i < 15; i++) w[i] = 0;
    w[15] = 256;
    uint32_t h[8];
    sha256_init(h);
    sha256_compress(h, w);
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = h[i];
}

BLS_HD uint32_t bswap32(uint32_t x) { return (x >> 24) | ((x >> 8) & 0xff00) | ((x << 8) & 0xff0000) | (x << 24); }

// little-endian u64 made of the first 8 digest bytes (blst_min_pubkey_sig_core.nim:545-556)
BLS_HD uint64_t digest_low_u64_le(const uint32_t (&d)[8]) { return (uint64_t)bswap32(d[0]) | ((uint64_t)bswap32(d[1]) << 32); }

}  // namespace bls
