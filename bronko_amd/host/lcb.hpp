// lcb.hpp -- k-mer primitives of the LCB scheme, host side (product code; the oracle has its own copy).
//
// Behaviour follows /root/reference/src/lcb.rs: nt_to_bits :47-55 (non-ACGT encodes as A), kmer_to_u64
// :67-74 (first base in the highest 2 bits), reverse_complement_u64 :76-85, canonical_kmer :87-95
// ((fwd,false) iff fwd < rc), assign_buckets :1-45 (wrapping u64 rank of (wildcard position, other bases)).
#pragma once
#include <cstdint>

// (the same primitives run in the device index build, bronko_amd/csrc/bk_build.hip)
#if defined(__HIPCC__)
#define BRONKO_HD __host__ __device__
#else
#define BRONKO_HD
#endif

namespace bronko {

BRONKO_HD inline uint8_t nt_to_bits(uint8_t c) {
    // one 256-entry table would do; a switch keeps the mapping readable (lcb.rs:47-55)
    switch (c | 0x20) {
        case 'c': return 1;
        case 'g': return 2;
        case 't': return 3;
        default: return 0;  // 'a' and everything else
    }
}

// -1 for symbols that break a k-mer run in the read path (KMC semantics), else 0..3
inline int acgt_code(uint8_t c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return -1;
    }
}

BRONKO_HD inline uint64_t kmer_mask(int k) { return k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1); }

BRONKO_HD inline uint64_t kmer_to_u64(const uint8_t* s, int k) {
    uint64_t v = 0;
    for (int i = 0; i < k; i++) v = (v << 2) | nt_to_bits(s[i]);
    return v;
}

// reverse complement of a 2k-bit value by swapping 2-bit groups, then complementing
BRONKO_HD inline uint64_t reverse_complement_u64(uint64_t v, int k) {
    v = ((v >> 2) & 0x3333333333333333ull) | ((v & 0x3333333333333333ull) << 2);
    v = ((v >> 4) & 0x0f0f0f0f0f0f0f0full) | ((v & 0x0f0f0f0f0f0f0f0full) << 4);
    v = ((v >> 8) & 0x00ff00ff00ff00ffull) | ((v & 0x00ff00ff00ff00ffull) << 8);
    v = ((v >> 16) & 0x0000ffff0000ffffull) | ((v & 0x0000ffff0000ffffull) << 16);
    v = (v >> 32) | (v << 32);   // (byte swap spelled out: the same code on host and device)
    return (~v) >> (64 - 2 * k);
}

struct Canon { uint64_t kmer; bool rc; };
BRONKO_HD inline Canon canonical_u64(uint64_t fwd, int k) {
    const uint64_t rev = reverse_complement_u64(fwd, k);
    return fwd < rev ? Canon{fwd, false} : Canon{rev, true};
}
BRONKO_HD inline Canon canonical_kmer(const uint8_t* s, int k) { return canonical_u64(kmer_to_u64(s, k), k); }

// ids[j] for j in [0,k): rank of (wildcard position j counted from the left, the k-1 other bases).
// Written as prefix/suffix sums of the per-digit weights of lcb.rs:12-40; u64 arithmetic wraps for k = 31.
BRONKO_HD inline void assign_buckets(uint64_t kmer, int k, uint64_t* ids) {
    uint64_t mu[32], suffix[32], zeros_before[32];
    uint64_t total = 0, rest = kmer, nz = 0;
    for (int i = 0; i < k; i++) {
        const int sh = 2 * (k - 1 - i);
        const uint64_t digit = (kmer >> sh) & 3;
        rest -= digit << sh;  // value of the digits to the right of position i
        zeros_before[i] = nz;
        suffix[i] = rest;
        mu[i] = digit ? (1ull << sh) + (((digit << sh) >> 2) * (uint64_t)(k - 1 - i)) : rest;
        total += mu[i];
        nz += (digit == 0);
    }
    for (int i = 0; i < k; i++) {
        const uint64_t cur = kmer & (3ull << (2 * (k - 1 - i)));
        ids[i] = total - mu[i] + suffix[i] - zeros_before[i] * cur + 1 + zeros_before[i];
    }
}

}  // namespace bronko
