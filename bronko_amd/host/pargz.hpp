// pargz.hpp -- one gzip file inflated on several host threads (host side of `bronko call`: the reads' way in).
//
// KMC, which the reference hands its FASTQ files to, reads them with its -t threads (/root/reference/src/call.rs:1166-1181); zlib's
// gzread inflates one stream on one core (~1 M 150-bp reads a second), three orders of magnitude below what the device path takes.
// A deflate stream has no index, but it can be entered at any block boundary if the 32 KB of text before it are treated as
// unknowns (the approach of Kerbiriou & Chikhi, "Parallel decompression of gzip-compressed files and random access to DNA
// sequences", 2019 -- restated here from the format, RFC 1951 / 1952; no code of theirs):
//
//   1. the compressed bytes are cut into chunks; a thread looks for the first block boundary in its chunk -- a bit position at
//      which a non-final dynamic-Huffman block header parses (complete code-length, literal/length and distance codes), the block
//      decodes to text bytes only, and another well-formed block header follows;
//   2. from there it inflates into 16-bit symbols: a byte, or "whatever stood at position p of the 32 KB window before this
//      chunk" -- copies carry such symbols along like bytes -- and stops at the boundary the next chunk's thread started from (a
//      start that turns out not to be a boundary of the real stream is run over, and that chunk's work dropped);
//      a wave's first chunk starts where the text is known and is inflated straight into bytes (ByteBuf);
//      (amplicon reads repeat what was read a few KB before: most of a chunk's text is copies of copies that lead back into the
//      unknown window -- a list of the unknowns' places instead of a symbol per byte was tried and is 25x slower on such data);
//   3. the chunks' last 32 KB are resolved one after the other (each needs the one before), then every chunk is turned into bytes
//      by its own thread and its CRC-32 taken; the member's CRC and length are checked when its trailer comes by
//      (crc32_combine).
//
// Members written by bgzip (BGZF: an extra field that names the member's size) are independent blocks of <= 64 KB and take the
// short way: zlib, one member per task.  Concatenated members, stored and fixed-code blocks are handled; a stream without
// dynamic blocks to enter at simply decodes on one thread.  Output and error behaviour are gzread's: the same bytes, an error
// for a damaged stream or a CRC / length mismatch, bytes behind the last member ignored.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <exception>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace bronko {
namespace pargz {

constexpr uint32_t kWin = 32768;            // deflate's window
constexpr uint16_t kUnknown = 0x8000;       // symbol: kUnknown | position in the window before the chunk (0 = 32 KB back)
constexpr uint64_t kNone = ~0ull;
constexpr size_t kMaxChunkText = 192u << 20;   // symbols of one chunk before it ends its wave (a 2 MB chunk of FASTQ is 8 MB of text)

struct Corrupt : std::runtime_error { using std::runtime_error::runtime_error; };

// ---- bits, least significant first ----------------------------------------------------------------------------------------
struct Bits {
    const uint8_t* base = nullptr;   // the file
    const uint8_t* p = nullptr;      // next byte to load
    const uint8_t* end = nullptr;    // one past the file's last byte
    uint8_t tail[32];                // the file's last bytes, zero padded: loads of 8 bytes never leave the mapping
    const uint8_t* tail_at = nullptr;
    bool in_tail = false;
    uint64_t buf = 0;
    unsigned cnt = 0;

    void open(const uint8_t* b, size_t n) {
        base = b; end = b + n;
        const size_t t = std::min<size_t>(n, 16);
        memset(tail, 0, sizeof tail);
        memcpy(tail, end - t, t);
        tail_at = end - t;
    }
    void seek(uint64_t bit) {
        p = base + (bit >> 3); in_tail = false; buf = 0; cnt = 0;
        refill();
        take((unsigned)(bit & 7));
    }
    inline void refill() {
        if (__builtin_expect(!in_tail && p >= tail_at, 0)) { p = tail + std::min<size_t>((size_t)(p - tail_at), 24); in_tail = true; }
        else if (__builtin_expect(in_tail && p > tail + 24, 0)) p = tail + 24;   // (far behind the end: zeros, and past_end() says so)
        uint64_t w;
        memcpy(&w, p, 8);
        buf |= w << cnt;
        const unsigned adv = (63u - cnt) >> 3;
        p += adv; cnt += adv * 8u;
    }
    inline uint32_t peek(unsigned n) const { return (uint32_t)(buf & ((1ull << n) - 1ull)); }
    inline void take(unsigned n) { buf >>= n; cnt -= n; }
    inline uint32_t get(unsigned n) { const uint32_t v = peek(n); take(n); return v; }
    uint64_t pos() const {   // bit position in the file of the next unread bit
        const uint64_t byte = in_tail ? (uint64_t)(tail_at - base) + (uint64_t)(p - tail) : (uint64_t)(p - base);
        return byte * 8u - cnt;
    }
    bool past_end() const { return pos() > (uint64_t)(end - base) * 8u; }
};

// ---- Huffman decoding tables: a first level of kPrim bits, longer codes through second-level tables ------------------------
// entry: bits 0-7 code length to take (second level: the part behind the first kPrim bits), bits 8-11 kind, bits 12-15 extra bits,
// bits 16-31 value (byte / base length / base distance / offset of the second-level table)
enum : uint32_t { kLit = 1u << 8, kEob = 2u << 8, kSub = 4u << 8, kBad = 8u << 8, kKindMask = 0xfu << 8 };
constexpr unsigned kPrimLit = 10, kPrimDist = 8;

struct Table {
    std::vector<uint32_t> e;
    unsigned prim = 0;
};

inline uint32_t rev_bits(uint32_t v, unsigned n) {
    uint32_t r = 0;
    for (unsigned i = 0; i < n; i++) { r = (r << 1) | (v & 1u); v >>= 1; }
    return r;
}

// 0 complete, 1 incomplete, -1 over-subscribed.  `ent(sym)` makes the entry of a symbol but for its length field.
template <class MakeEntry>
int build_table(const uint8_t* lens, unsigned n, unsigned prim, Table& t, MakeEntry ent) {
    unsigned count[16] = {0};
    for (unsigned i = 0; i < n; i++) count[lens[i]]++;
    count[0] = 0;
    int64_t left = 1;
    unsigned maxlen = 0;
    for (unsigned l = 1; l <= 15; l++) {
        left = left * 2 - (int64_t)count[l];
        if (left < 0) return -1;
        if (count[l]) maxlen = l;
    }
    uint32_t next[16];
    {
        uint32_t code = 0;
        for (unsigned l = 1; l <= 15; l++) { code = (code + count[l - 1]) << 1; next[l] = code; }
    }
    t.prim = prim;
    const uint32_t psize = 1u << prim;
    t.e.assign(psize, kBad | 1u);
    // second-level tables: the longest code behind each first-level prefix decides their size
    std::vector<uint8_t> sub_bits;
    std::vector<uint32_t> codes(n);
    if (maxlen > prim) sub_bits.assign(psize, 0);
    for (unsigned i = 0; i < n; i++) {
        const unsigned l = lens[i];
        if (!l) continue;
        const uint32_t r = rev_bits(next[l]++, l);
        codes[i] = r;
        if (l > prim) { uint8_t& sb = sub_bits[r & (psize - 1)]; sb = std::max<uint8_t>(sb, (uint8_t)(l - prim)); }
    }
    if (maxlen > prim)
        for (uint32_t pfx = 0; pfx < psize; pfx++)
            if (sub_bits[pfx]) {
                const uint32_t off = (uint32_t)t.e.size();
                t.e.resize(t.e.size() + (1u << sub_bits[pfx]), kBad | 1u);
                t.e[pfx] = kSub | ((uint32_t)sub_bits[pfx] << 12) | (off << 16) | prim;
            }
    for (unsigned i = 0; i < n; i++) {
        const unsigned l = lens[i];
        if (!l) continue;
        const uint32_t r = codes[i];
        if (l <= prim) {
            const uint32_t v = ent(i) | l;
            for (uint32_t x = r; x < psize; x += 1u << l) t.e[x] = v;
        } else {
            const uint32_t head = t.e[r & (psize - 1)];
            const uint32_t off = head >> 16, sb = (head >> 12) & 15u;
            const uint32_t v = ent(i) | (l - prim);
            for (uint32_t x = r >> prim; x < (1u << sb); x += 1u << (l - prim)) t.e[off + x] = v;
        }
    }
    return left > 0 ? 1 : 0;
}

inline const uint16_t* len_base() { static const uint16_t b[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258}; return b; }
inline const uint8_t* len_extra() { static const uint8_t b[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0}; return b; }
inline const uint16_t* dist_base() { static const uint16_t b[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577}; return b; }
inline const uint8_t* dist_extra() { static const uint8_t b[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13}; return b; }

inline uint32_t lit_entry(unsigned s) {
    if (s < 256) return kLit | ((uint32_t)s << 16);
    if (s == 256) return kEob;
    if (s > 285) return kBad;                                        // (286, 287: in the fixed code, never in data)
    return ((uint32_t)len_extra()[s - 257] << 12) | ((uint32_t)len_base()[s - 257] << 16);
}
inline uint32_t dist_entry(unsigned s) {
    if (s > 29) return kBad;
    return ((uint32_t)dist_extra()[s] << 12) | ((uint32_t)dist_base()[s] << 16);
}

struct Codes {
    Table lit, dist;
};

inline const Codes& fixed_codes() {
    static const Codes c = [] {
        Codes f;
        uint8_t l[288];
        for (int i = 0; i < 144; i++) l[i] = 8;
        for (int i = 144; i < 256; i++) l[i] = 9;
        for (int i = 256; i < 280; i++) l[i] = 7;
        for (int i = 280; i < 288; i++) l[i] = 8;
        build_table(l, 288, kPrimLit, f.lit, lit_entry);
        uint8_t d[32];
        for (int i = 0; i < 32; i++) d[i] = 5;
        build_table(d, 32, kPrimDist, f.dist, dist_entry);
        return f;
    }();
    return c;
}

// the header of a dynamic block (RFC 1951 3.2.7), as strict as zlib: false where zlib says "invalid"
inline bool read_dynamic_header(Bits& in, Codes& c) {
    in.refill();
    const unsigned hlit = in.get(5) + 257, hdist = in.get(5) + 1, hclen = in.get(4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t cl[19] = {0};
    for (unsigned i = 0; i < hclen; i++) {
        if ((i & 7) == 0) in.refill();
        cl[order[i]] = (uint8_t)in.get(3);
    }
    {
        unsigned kraft = 0;                                          // complete code or nothing (zlib: "invalid code lengths set")
        for (unsigned i = 0; i < 19; i++) if (cl[i]) kraft += 128u >> cl[i];
        if (kraft != 128u) return false;
    }
    Table ct;
    if (build_table(cl, 19, 7, ct, [](unsigned s) { return (uint32_t)s << 16; }) != 0) return false;
    uint8_t lens[286 + 30 + 138];
    unsigned n = 0;
    const unsigned total = hlit + hdist;
    while (n < total) {
        in.refill();
        const uint32_t e = ct.e[in.peek(7)];
        if (e & kBad) return false;
        in.take(e & 0xffu);
        const unsigned s = e >> 16;
        if (s < 16) { lens[n++] = (uint8_t)s; continue; }
        unsigned rep;
        uint8_t v = 0;
        if (s == 16) { if (!n) return false; v = lens[n - 1]; rep = 3 + in.get(2); }
        else if (s == 17) rep = 3 + in.get(3);
        else rep = 11 + in.get(7);
        if (n + rep > total) return false;
        while (rep--) lens[n++] = v;
    }
    if (in.past_end()) return false;
    if (!lens[256]) return false;                                    // no end-of-block code
    // an incomplete code only where it is a single code of one bit (zlib's inflate_table), or no distance code at all
    auto incomplete_ok = [](const uint8_t* l, unsigned n) {
        unsigned used = 0, mx = 0;
        for (unsigned i = 0; i < n; i++) if (l[i]) { used++; mx = std::max<unsigned>(mx, l[i]); }
        return used == 0 || (used == 1 && mx == 1);
    };
    const int lr = build_table(lens, hlit, kPrimLit, c.lit, lit_entry);
    if (lr < 0 || (lr > 0 && !incomplete_ok(lens, hlit))) return false;
    const int dr = build_table(lens + hlit, hdist, kPrimDist, c.dist, dist_entry);
    if (dr < 0 || (dr > 0 && !incomplete_ok(lens + hlit, hdist))) return false;
    return true;
}

// ---- a chunk's text as symbols ----------------------------------------------------------------------------------------------
struct MemberEnd { uint64_t at; uint32_t crc, isize; };   // a member ended after `at` symbols of this chunk

// Large buffers (a chunk's symbols, a chunk's text): anonymous mappings kept for the whole process and handed round -- several
// lanes' files are read at once in one address space, and fresh pages by the hundred thousand per file are faults under that
// address space's one lock.
struct Block {
    char* p = nullptr;
    size_t cap = 0;
    void* base = nullptr;
    size_t len = 0;
    Block() = default;
    Block(const Block&) = delete;
    Block& operator=(const Block&) = delete;
    Block(Block&& o) noexcept : p(o.p), cap(o.cap), base(o.base), len(o.len) { o.p = nullptr; o.cap = 0; o.base = nullptr; o.len = 0; }
    Block& operator=(Block&& o) noexcept;
    ~Block();
};
class BlockPool {
public:
    static BlockPool& get() { static BlockPool* pool = new BlockPool; return *pool; }   // (never destroyed: blocks may outlive main)
    Block take(size_t want) {
        {
            std::unique_lock<std::mutex> lk(m_);
            size_t best = free_.size();
            for (size_t i = 0; i < free_.size(); i++)
                if (free_[i].cap >= want && (best == free_.size() || free_[i].cap < free_[best].cap)) best = i;
            if (best < free_.size() && free_[best].cap <= 4 * want + (8u << 20)) {
                Block b = std::move(free_[best]);
                free_.erase(free_.begin() + (long)best);
                held_ -= b.cap;
                return b;
            }
        }
        constexpr size_t kHuge = 2u << 20;
        Block b;
        b.cap = (want + want / 4 + kHuge) & ~(kHuge - 1);
        b.len = b.cap + kHuge;
        b.base = mmap(nullptr, b.len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (b.base == MAP_FAILED) { b.base = nullptr; b.cap = 0; b.len = 0; throw std::bad_alloc(); }
        b.p = reinterpret_cast<char*>(((uintptr_t)b.base + kHuge - 1) & ~(uintptr_t)(kHuge - 1));
        if (getenv("BRONKO_PARGZ_THP")) madvise(b.p, b.cap, MADV_HUGEPAGE);   // (2 MB pages: measured a loss where the kernel compacts memory to make them)
        return b;
    }
    void give(Block&& b) {
        if (!b.base) return;
        std::unique_lock<std::mutex> lk(m_);
        if (held_ + b.cap > kKeep) { lk.unlock(); munmap(b.base, b.len); b.p = nullptr; b.cap = 0; b.base = nullptr; b.len = 0; return; }
        held_ += b.cap;
        free_.push_back(std::move(b));
    }
private:
    static constexpr size_t kKeep = 6ull << 30;
    std::mutex m_;
    std::vector<Block> free_;
    size_t held_ = 0;
};
inline Block& Block::operator=(Block&& o) noexcept {
    if (this != &o) {
        BlockPool::get().give(std::move(*this));
        p = o.p; cap = o.cap; base = o.base; len = o.len;
        o.p = nullptr; o.cap = 0; o.base = nullptr; o.len = 0;
    }
    return *this;
}
inline Block::~Block() { if (base) BlockPool::get().give(std::move(*this)); }

struct SymBuf {
    Block b;
    uint16_t* d = nullptr;
    size_t n = 0, cap = 0;
    size_t member_from = 0;      // where the gzip member being inflated began, if it began inside this buffer (member_inside):
    bool member_inside = false;  // a distance may not reach past it (zlib: "invalid distance too far back")
    SymBuf() = default;
    SymBuf(SymBuf&& o) noexcept : b(std::move(o.b)), d(o.d), n(o.n), cap(o.cap), member_from(o.member_from), member_inside(o.member_inside) { o.d = nullptr; o.n = 0; o.cap = 0; }
    SymBuf& operator=(SymBuf&& o) noexcept {
        b = std::move(o.b); d = o.d; n = o.n; cap = o.cap; member_from = o.member_from; member_inside = o.member_inside;
        o.d = nullptr; o.n = 0; o.cap = 0; return *this;
    }
    void room(size_t extra) {
        if (n + extra <= cap) return;
        Block nb = BlockPool::get().take(std::max<size_t>(cap * 2, n + extra + (1u << 16)) * sizeof(uint16_t));
        if (n) memcpy(nb.p, d, n * sizeof(uint16_t));
        b = std::move(nb);
        d = reinterpret_cast<uint16_t*>(b.p); cap = b.cap / sizeof(uint16_t);
    }
};

// A chunk that starts where the text before it is known -- a wave's first, and every chunk when one thread does the work -- needs no
// symbols: it is inflated straight into bytes, the 32 KB before it copied in front of the buffer so that early copies find them.
struct ByteBuf {
    Block b;
    uint8_t* d = nullptr;        // the chunk's text; d[-before .. 0) is the text before it
    size_t n = 0, cap = 0;
    uint32_t before = 0;
    size_t member_from = 0;      // as SymBuf's
    bool member_inside = false;
    void start(const uint8_t* window, size_t n_window, size_t want) {
        b = BlockPool::get().take(kWin + want);
        d = reinterpret_cast<uint8_t*>(b.p) + kWin; cap = b.cap - kWin; n = 0;
        member_from = 0; member_inside = false;
        before = (uint32_t)std::min<size_t>(n_window, kWin);
        if (before) memcpy(d - before, window + (n_window - before), before);
    }
    void room(size_t extra) {
        if (n + extra <= cap) return;
        Block nb = BlockPool::get().take(kWin + std::max<size_t>(cap * 2, n + extra + (1u << 16)));
        memcpy(nb.p + kWin - before, d - before, before + n);
        b = std::move(nb);
        d = reinterpret_cast<uint8_t*>(b.p) + kWin; cap = b.cap - kWin;
    }
};
template <class B> struct BufTraits;
template <> struct BufTraits<SymBuf> { typedef uint16_t sym; static constexpr bool symbolic = true; static uint32_t before(const SymBuf&) { return 0u; } };
template <> struct BufTraits<ByteBuf> { typedef uint8_t sym; static constexpr bool symbolic = false; static uint32_t before(const ByteBuf& b) { return b.before; } };

inline bool text_byte(uint32_t b) { return (b >= 32 && b < 127) || b == '\n' || b == '\r' || b == '\t'; }

// One block's data behind its header.  TEXT: stop with false at a byte that is no text (the test of a candidate boundary).
template <bool TEXT, class B>
inline bool inflate_block(Bits& in, const Codes& c, B& out, size_t max_out) {
    typedef typename BufTraits<B>::sym sym_t;
    const uint32_t* lt = c.lit.e.data();
    const uint32_t* dt = c.dist.e.data();
    for (;;) {
        out.room(260);
        if (out.n > max_out) return false;
        if (__builtin_expect(in.in_tail, 0) && in.past_end()) return false;
        in.refill();
        uint32_t e = lt[in.peek(kPrimLit)];
        if (e & kSub) { in.take(kPrimLit); e = lt[(e >> 16) + in.peek((e >> 12) & 15u)]; }
        in.take(e & 0xffu);
        if (e & kLit) {
            if (TEXT && !text_byte(e >> 16)) return false;
            out.d[out.n++] = (sym_t)(e >> 16);
            // a second literal from the same refill (codes are at most 15 bits: 56 - 15 - 15 > 15)
            e = lt[in.peek(kPrimLit)];
            if ((e & (kKindMask)) == kLit) {
                if (TEXT && !text_byte(e >> 16)) return false;
                in.take(e & 0xffu);
                out.d[out.n++] = (sym_t)(e >> 16);
            }
            continue;
        }
        if (e & (kEob | kBad)) {
            if (e & kBad) return false;
            return !in.past_end();
        }
        const uint32_t len = (e >> 16) + in.get((e >> 12) & 15u);
        uint32_t d = dt[in.peek(kPrimDist)];
        if (d & kSub) { in.take(kPrimDist); d = dt[(d >> 16) + in.peek((d >> 12) & 15u)]; }
        if (d & kBad) return false;
        in.take(d & 0xffu);
        const uint32_t dist = (d >> 16) + in.get((d >> 12) & 15u);
        sym_t* o = out.d + out.n;
        // how far back this member's text goes in the buffer (bytes: the text before the chunk lies in front of the buffer)
        const size_t reach = out.member_inside ? out.n - out.member_from : out.n + BufTraits<B>::before(out);
        if (dist <= reach) {
            const sym_t* s = o - dist;
            if (dist >= len) memcpy(o, s, len * sizeof(sym_t));
            else for (uint32_t i = 0; i < len; i++) o[i] = s[i];
        } else if constexpr (BufTraits<B>::symbolic) {
            // reaches into the window before the chunk
            if (out.member_inside) return false;                     // ... which is another member's text
            const uint32_t before = dist - (uint32_t)out.n;          // symbols back from the chunk's start, first one copied
            if (before > kWin) return false;
            for (uint32_t i = 0; i < len; i++) {
                const int64_t src = (int64_t)out.n + i - dist;
                o[i] = src >= 0 ? out.d[(size_t)src] : (sym_t)(kUnknown | (uint32_t)(kWin + src));
            }
        } else return false;                                         // a distance before the start of the text
        out.n += len;
    }
}

// gzip member header at a byte position (RFC 1952); returns the position of the deflate data, kNone if there is no member here.
// bgzf_size: BSIZE + 1 of a BGZF member, else 0
inline uint64_t member_header(const uint8_t* b, uint64_t at, uint64_t n, uint32_t* bgzf_size = nullptr) {
    if (bgzf_size) *bgzf_size = 0;
    if (at + 18 > n || b[at] != 0x1f || b[at + 1] != 0x8b || b[at + 2] != 8) return kNone;
    const unsigned flg = b[at + 3];
    if (flg & 0xe0) return kNone;
    uint64_t p = at + 10;
    if (flg & 4) {
        if (p + 2 > n) return kNone;
        const uint64_t xlen = b[p] | ((uint64_t)b[p + 1] << 8);
        p += 2;
        if (p + xlen > n) return kNone;
        for (uint64_t q = p; q + 4 <= p + xlen;) {
            const uint64_t sl = b[q + 2] | ((uint64_t)b[q + 3] << 8);
            if (b[q] == 'B' && b[q + 1] == 'C' && sl == 2 && q + 6 <= p + xlen && bgzf_size) *bgzf_size = (b[q + 4] | ((uint32_t)b[q + 5] << 8)) + 1u;
            q += 4 + sl;
        }
        p += xlen;
    }
    if (flg & 8) { while (p < n && b[p]) p++; p++; }
    if (flg & 16) { while (p < n && b[p]) p++; p++; }
    if (flg & 2) p += 2;
    return p + 8 <= n ? p : kNone;
}

// a stretch of the text on its way to the reader
struct Piece {
    Block b;
    char* d = nullptr;
    size_t n = 0, cap = 0;
    Piece() = default;
    Piece(Piece&& o) noexcept : b(std::move(o.b)), d(o.d), n(o.n), cap(o.cap) { o.d = nullptr; o.n = 0; o.cap = 0; }
    Piece& operator=(Piece&& o) noexcept { b = std::move(o.b); d = o.d; n = o.n; cap = o.cap; o.d = nullptr; o.n = 0; o.cap = 0; return *this; }
    void room(size_t want) { if (want > cap) { b = BlockPool::get().take(want); d = b.p; cap = b.cap; } }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
};

struct Chunk {
    uint64_t nominal = 0;     // bit position the search for a boundary starts from
    uint64_t start = kNone;   // the boundary decoding starts from (kNone: none found)
    uint64_t stop = 0;        // where decoding ended (a block boundary)
    int next = -1;            // the chunk that starts at `stop` (-1: end of the wave's data)
    bool eof = false;         // ended behind the last member
    bool failed = false;      // ran into something that is no deflate data (the stream, or a start that was no boundary)
    std::string what;
    SymBuf sym;
    ByteBuf raw;              // ... or its text itself, if it was inflated from a known window (as_bytes)
    bool as_bytes = false;
    size_t text_len() const { return as_bytes ? raw.n : sym.n; }
    std::vector<MemberEnd> ends;
    Piece bytes;
    std::vector<uint32_t> seg_crc;   // CRC-32 of the stretches between member ends (ends.size() + 1 of them)
};

// Blocks from `c.start` on, up to the boundary another chunk of the wave starts from, or the first one at or behind `wave_end`.
template <class B>
inline void inflate_chunk_into(const uint8_t* file, uint64_t n_bytes, std::vector<Chunk>& cs, size_t me, uint64_t wave_end, B& out) {
    Chunk& c = cs[me];
    Bits in;
    in.open(file, n_bytes);
    in.seek(c.start);
    out.room((size_t)(((me + 1 < cs.size() ? cs[me + 1].nominal : wave_end) - std::min(c.start, wave_end)) / 8u) * 5u);   // (text is ~4x its gzip)
    size_t m = me + 1;
    Codes dyn;
    try {
        for (;;) {
            in.refill();
            const uint32_t final = in.get(1), type = in.get(2);
            if (type == 0) {
                in.take(in.cnt & 7u);
                in.refill();
                const uint32_t len = in.get(16), nlen = in.get(16);
                if ((len ^ 0xffffu) != nlen) throw Corrupt("stored block length");
                const uint64_t at = in.pos() >> 3;
                if (at + len > n_bytes) throw Corrupt("stored block runs past the end");
                out.room(len);
                for (uint32_t i = 0; i < len; i++) out.d[out.n + i] = file[at + i];
                out.n += len;
                in.seek((at + len) * 8u);
            } else if (type == 1) {
                if (!inflate_block<false>(in, fixed_codes(), out, ~(size_t)0)) throw Corrupt("bad data");
            } else if (type == 2) {
                if (!read_dynamic_header(in, dyn)) throw Corrupt("bad code lengths");
                if (!inflate_block<false>(in, dyn, out, ~(size_t)0)) throw Corrupt("bad data");
            } else throw Corrupt("bad block type");
            if (final) {
                in.take(in.cnt & 7u);
                uint64_t at = in.pos() >> 3;
                if (at + 8 > n_bytes) throw Corrupt("no trailer");
                MemberEnd me_;
                me_.at = out.n;
                memcpy(&me_.crc, file + at, 4); memcpy(&me_.isize, file + at + 4, 4);
                c.ends.push_back(me_);
                out.member_from = out.n; out.member_inside = true;   // what follows is a new member: its distances stop here
                const uint64_t nx = member_header(file, at + 8, n_bytes);
                if (nx == kNone) { c.eof = true; c.stop = (at + 8) * 8u; return; }   // (gzread: what follows the last member is ignored)
                in.seek(nx * 8u);
            }
            const uint64_t b = in.pos();
            // later chunks whose start was run over (no boundary of this stream) or that found none in their stretch
            while (m < cs.size()) {
                const uint64_t s = cs[m].start;
                if (s != kNone ? s < b : (m + 1 < cs.size() ? cs[m + 1].nominal : wave_end) <= b) m++;
                else break;
            }
            if (m < cs.size() && cs[m].start == b) { c.stop = b; c.next = (int)m; return; }
            // text far out of proportion to its gzip (long runs of one symbol): the wave ends here, what later chunks made of it is
            // dropped and the next wave starts from this boundary -- memory stays bounded, such a file is read chunk by chunk
            if (out.n > kMaxChunkText) { c.stop = b; c.next = -1; return; }
            if (m == cs.size() && b >= wave_end) { c.stop = b; c.next = -1; return; }
        }
    } catch (const Corrupt& e) {
        c.failed = true; c.what = e.what();
    }
}

inline void inflate_chunk(const uint8_t* file, uint64_t n_bytes, std::vector<Chunk>& cs, size_t me, uint64_t wave_end) {
    if (cs[me].as_bytes) inflate_chunk_into(file, n_bytes, cs, me, wave_end, cs[me].raw);
    else inflate_chunk_into(file, n_bytes, cs, me, wave_end, cs[me].sym);
}

// The first boundary at or behind bit `from`, before `to` (the test of the header comment); kNone if there is none.
inline uint64_t find_boundary(const uint8_t* file, uint64_t n_bytes, uint64_t from, uint64_t to) {
    Bits in;
    in.open(file, n_bytes);
    Codes dyn, dyn2;
    SymBuf scratch;
    const uint64_t total_bits = n_bytes * 8u;
    to = std::min(to, total_bits > 64 ? total_bits - 64 : 0);
    for (uint64_t p = from; p < to; p++) {
        // BFINAL = 0, BTYPE = 2, HLIT <= 29, HDIST <= 29: looked at in the bytes before anything is set up
        uint64_t w = 0;
        memcpy(&w, file + (p >> 3), std::min<uint64_t>(8, n_bytes - (p >> 3)));
        w >>= (p & 7);
        if ((w & 7u) != 4u) continue;
        if (((w >> 3) & 31u) > 29u || ((w >> 8) & 31u) > 29u) continue;
        in.seek(p + 3);
        if (!read_dynamic_header(in, dyn)) continue;
        scratch.n = 0;
        if (!inflate_block<true>(in, dyn, scratch, 4u << 20)) continue;
        if (scratch.n < 1024) continue;                              // (too little to tell text from chance)
        // what follows: a block header again
        in.refill();
        const uint32_t f2 = in.get(1), t2 = in.get(2);
        (void)f2;
        if (t2 == 3) continue;
        if (t2 == 2 && !read_dynamic_header(in, dyn2)) continue;
        if (t2 == 0) {
            in.take(in.cnt & 7u);
            in.refill();
            const uint32_t len = in.get(16), nlen = in.get(16);
            if ((len ^ 0xffffu) != nlen) continue;
        }
        return p;
    }
    return kNone;
}

template <class F>
inline void parallel_for(size_t n, unsigned threads, F f) {
    if (n == 0) return;
    const unsigned nt = (unsigned)std::min<size_t>(std::max(1u, threads), n);
    if (nt == 1) { for (size_t i = 0; i < n; i++) f(i); return; }
    std::atomic<size_t> next{0};
    std::vector<std::thread> ts;
    std::exception_ptr first;          // (an exception that left a thread would end the process: the first one is thrown again behind the join)
    std::mutex em;
    auto work = [&] {
        try { for (size_t i; (i = next.fetch_add(1)) < n;) f(i); }
        catch (...) { std::unique_lock<std::mutex> lk(em); if (!first) first = std::current_exception(); next = n; }
    };
    for (unsigned t = 1; t < nt; t++) ts.emplace_back(work);
    work();
    for (auto& t : ts) t.join();
    if (first) std::rethrow_exception(first);
}

}  // namespace pargz

// The file's text, in order, from read(): a producer thread runs the waves and keeps a bounded queue of finished chunks ahead.
class ParallelGunzip {
public:
    // A gzip FILE that can be mapped: regular and not empty.  A FIFO, /dev/fd/N (process substitution), a terminal are never opened
    // here -- a probe would take bytes off the stream, and there is nothing to map -- and stay with zlib's gzread, which reads them
    // in one pass (`bronko call -r <(zcat x.gz)` gave 0 reads with rc 0, a plain-text pipe lost its first three bytes).
    static bool is_gzip(const std::string& path) {
        struct stat st;
        if (::stat(path.c_str(), &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 3) return false;
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        unsigned char h[3] = {0, 0, 0};
        const ssize_t r = ::pread(fd, h, 3, 0);
        ::close(fd);
        return r == 3 && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8;
    }
    ParallelGunzip(const std::string& path, unsigned threads, size_t chunk_bytes = 0) : path_(path), threads_(std::max(1u, threads)), chunk_(chunk_bytes) {
        fd_ = ::open(path.c_str(), O_RDONLY);
        if (fd_ < 0) throw std::runtime_error("cannot open " + path);
        struct stat st;
        if (fstat(fd_, &st) != 0) { ::close(fd_); throw std::runtime_error("cannot stat " + path); }
        if (!S_ISREG(st.st_mode)) { ::close(fd_); throw std::runtime_error(path + " is not a regular file (pargz maps its input; read streams with gzread)"); }
        n_ = (uint64_t)st.st_size;
        if (n_) {
            void* m = mmap(nullptr, n_, PROT_READ, MAP_PRIVATE, fd_, 0);
            if (m == MAP_FAILED) { ::close(fd_); throw std::runtime_error("cannot map " + path); }
            file_ = static_cast<const uint8_t*>(m);
            madvise(m, n_, MADV_SEQUENTIAL);
        }
        producer_ = std::thread([this] { produce(); });
    }
    ~ParallelGunzip() {
        { std::unique_lock<std::mutex> lk(m_); quit_ = true; }
        cv_.notify_all();
        if (producer_.joinable()) producer_.join();
        if (file_) munmap(const_cast<uint8_t*>(file_), n_);
        if (fd_ >= 0) ::close(fd_);
    }
    ParallelGunzip(const ParallelGunzip&) = delete;
    ParallelGunzip& operator=(const ParallelGunzip&) = delete;

    // The next piece of the text as it stands (no copy): false at the end.  Throws like read().  (Not to be mixed with read().)
    bool take(pargz::Piece& out) {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return !q_.empty() || done_; });
        if (q_.empty()) {
            if (!error_.empty()) throw std::runtime_error(error_ + " in " + path_);
            return false;
        }
        queued_ -= q_.front().size();
        out = std::move(q_.front()); q_.pop_front();
        lk.unlock();
        cv_.notify_all();
        return true;
    }

    // up to n bytes; 0 at the end of the text.  Throws std::runtime_error (damaged file) like a failed gzread.
    size_t read(char* dst, size_t n) {
        size_t got = 0;
        while (got < n) {
            if (cur_at_ == cur_.size()) {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return !q_.empty() || done_; });
                if (q_.empty()) {
                    if (!error_.empty()) throw std::runtime_error(error_ + " in " + path_);
                    break;
                }
                queued_ -= q_.front().size();
                cur_ = std::move(q_.front()); q_.pop_front(); cur_at_ = 0;
                lk.unlock();
                cv_.notify_all();
                continue;
            }
            const size_t k = std::min(n - got, cur_.size() - cur_at_);
            memcpy(dst + got, cur_.d + cur_at_, k);
            got += k; cur_at_ += k;
        }
        return got;
    }

private:
    static pargz::Piece fresh(size_t want) {
        pargz::Piece p;
        p.room(want);
        return p;
    }
    bool put(pargz::Piece&& v) {   // false: the reader went away
        if (v.empty()) return true;
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return queued_ < kAhead || quit_; });
        if (quit_) return false;
        queued_ += v.size();
        q_.push_back(std::move(v));
        lk.unlock();
        cv_.notify_all();
        return true;
    }
    void finish(const std::string& err) {
        { std::unique_lock<std::mutex> lk(m_); error_ = err; done_ = true; }
        cv_.notify_all();
    }
    // CRC and length of the member that is being read, over the pieces as they come
    struct Check {
        uint32_t crc = 0;
        uint64_t len = 0;
        void add(uint32_t c, uint64_t n) { crc = (uint32_t)crc32_combine(crc, c, (z_off_t)n); len += n; }
        bool ends(uint32_t c, uint32_t isize) { const bool ok = crc == c && (uint32_t)len == isize; crc = 0; len = 0; return ok; }
    };

    void produce() {
        try {
            if (n_ == 0) { finish(""); return; }   // (gzread: an empty file is an empty text)
            uint32_t bsize = 0;
            const uint64_t d0 = pargz::member_header(file_, 0, n_, &bsize);
            if (d0 == pargz::kNone) { finish("not a gzip file, or damaged"); return; }
            if (bsize) { if (!bgzf(0)) return; finish(""); return; }
            deflate_waves(d0 * 8u);
        } catch (const std::exception& e) {
            finish(e.what());
        }
    }

    // BGZF: members of known size, each inflated by zlib on its own.  Falls back to the general way at a member that is none.
    bool bgzf(uint64_t at) {
        struct Job { uint64_t data, end; std::vector<char> out; bool ok = true; };
        for (;;) {
            std::vector<Job> jobs;
            uint64_t bytes = 0;
            while (at < n_ && jobs.size() < 4096 && bytes < (64u << 20)) {
                uint32_t bs = 0;
                const uint64_t d = pargz::member_header(file_, at, n_, &bs);
                if (d == pargz::kNone) { if (at == 0) { finish("not a gzip file, or damaged"); return false; } at = n_; break; }   // (trailing bytes: ignored)
                uint32_t isz = 0;
                if (bs && at + bs <= n_ && at + bs >= d + 8) memcpy(&isz, file_ + at + bs - 4, 4);
                if (!bs || at + bs > n_ || at + bs < d + 8 || isz > 65536u) {   // a member that is not BGZF (whose text is at most 64 KB): the rest goes the general way
                    if (!flush_bgzf(jobs)) return false;
                    deflate_waves(d * 8u);
                    return false;
                }
                Job j; j.data = d; j.end = at + bs;
                jobs.push_back(std::move(j));
                bytes += bs; at += bs;
            }
            if (jobs.empty()) return true;
            if (!flush_bgzf(jobs)) return false;
        }
    }
    template <class Jobs>
    bool flush_bgzf(Jobs& jobs) {
        pargz::parallel_for(jobs.size(), threads_, [&](size_t i) {
            auto& j = jobs[i];
            uint32_t crc, isize;
            memcpy(&crc, file_ + j.end - 8, 4); memcpy(&isize, file_ + j.end - 4, 4);
            j.out.resize(isize);
            z_stream z;
            memset(&z, 0, sizeof z);
            if (inflateInit2(&z, -15) != Z_OK) { j.ok = false; return; }
            z.next_in = const_cast<Bytef*>(file_ + j.data); z.avail_in = (uInt)(j.end - 8 - j.data);
            Bytef none[1];                                           // (bgzip's empty last member: zlib wants a place to write to all the same)
            z.next_out = j.out.empty() ? none : reinterpret_cast<Bytef*>(j.out.data()); z.avail_out = (uInt)j.out.size();
            const int r = inflate(&z, Z_FINISH);
            j.ok = r == Z_STREAM_END && z.avail_out == 0 && (uint32_t)crc32(0, reinterpret_cast<const Bytef*>(j.out.data()), (uInt)j.out.size()) == crc;
            inflateEnd(&z);
        });
        // (members of 64 KB are gathered into pieces of a few MB for the queue)
        constexpr size_t kPiece = 4u << 20;
        pargz::Piece piece = fresh(kPiece + 65536);
        for (auto& j : jobs) {
            if (!j.ok) { if (!put(std::move(piece))) return false; finish("damaged BGZF block"); return false; }
            if (piece.n + j.out.size() > piece.cap) { if (!put(std::move(piece))) return false; piece = fresh(std::max(kPiece + 65536, j.out.size())); }
            if (!j.out.empty()) memcpy(piece.d + piece.n, j.out.data(), j.out.size());
            piece.n += j.out.size();
            if (piece.n >= kPiece) { if (!put(std::move(piece))) return false; piece = fresh(kPiece + 65536); }
        }
        jobs.clear();
        return put(std::move(piece));
    }

    void deflate_waves(uint64_t bit) {
        using namespace pargz;
        std::vector<uint8_t> window;            // the text's last 32 KB so far
        Check check;
        const uint64_t total_bits = n_ * 8u;
        for (;;) {
            // a wave: one chunk per thread from the known boundary `bit`
            const uint64_t left = n_ - (bit >> 3);
            uint64_t cb = chunk_ ? chunk_ : std::min<uint64_t>(std::max<uint64_t>(left / threads_ + 1, 256u << 10), 2u << 20);
            const size_t nc = (size_t)std::min<uint64_t>(threads_, (left + cb - 1) / cb);
            std::vector<Chunk> cs(std::max<size_t>(nc, 1));
            for (size_t i = 0; i < cs.size(); i++) cs[i].nominal = bit + (uint64_t)i * cb * 8u;
            const uint64_t wave_end = std::min(total_bits, bit + (uint64_t)cs.size() * cb * 8u);
            cs[0].start = bit;
            cs[0].as_bytes = true;       // the text before it is known: inflated straight into bytes
            cs[0].raw.start(window.data(), window.size(), (size_t)cb * 5u);
            const auto t0 = std::chrono::steady_clock::now();
            parallel_for(cs.size() - 1, threads_, [&](size_t i) {
                Chunk& c = cs[i + 1];
                c.start = find_boundary(file_, n_, c.nominal, std::min(wave_end, c.nominal + cb * 8u));
            });
            const auto t1 = std::chrono::steady_clock::now();
            parallel_for(cs.size(), threads_, [&](size_t i) {
                if (cs[i].start != kNone) inflate_chunk(file_, n_, cs, i, wave_end);
            });
            const auto t2 = std::chrono::steady_clock::now();
            // the chunks that make up the stream, in order
            std::vector<size_t> chain;
            for (int i = 0; i >= 0; i = cs[(size_t)i].next) {
                chain.push_back((size_t)i);
                if (cs[(size_t)i].failed) break;
            }
            // their last 32 KB, one after the other: the window each of them starts from
            std::vector<std::vector<uint8_t>> wins(chain.size());
            for (size_t x = 0; x < chain.size(); x++) {
                wins[x] = window;
                const Chunk& c = cs[chain[x]];
                if (c.failed) break;
                // (a member that ended inside the chunk: the window holds the text since the last such end only, so that the next
                // chunk's distances cannot reach into another member's text -- gzread's "invalid distance too far back")
                const size_t n = c.text_len(), since = c.ends.empty() ? 0 : (size_t)c.ends.back().at;
                const size_t keep = std::min<size_t>(n - since, kWin);
                std::vector<uint8_t> nw;
                nw.reserve(kWin);
                if (keep < kWin && !window.empty() && c.ends.empty()) {
                    const size_t from_old = std::min<size_t>(kWin - keep, window.size());
                    nw.insert(nw.end(), window.end() - from_old, window.end());
                }
                if (c.as_bytes) nw.insert(nw.end(), c.raw.d + (n - keep), c.raw.d + n);
                else for (size_t i = n - keep; i < n; i++) {
                    const uint16_t s = c.sym.d[i];
                    if (s < 256) nw.push_back((uint8_t)s);
                    else {
                        const uint32_t pos = s & (kWin - 1);           // position in a full 32 KB window; `window` may be shorter (start of the text)
                        const size_t missing = kWin - window.size();
                        if (pos < missing) throw std::runtime_error("damaged data (distance before the start)");
                        nw.push_back(window[pos - missing]);
                    }
                }
                window.swap(nw);
            }
            // symbols -> bytes and CRCs, every chunk on its own
            for (size_t x = 0; x < chain.size(); x++) {
                Chunk& c = cs[chain[x]];
                if (c.failed) continue;
                if (c.as_bytes) {                                    // its buffer is the piece
                    c.bytes.b = std::move(c.raw.b);
                    c.bytes.d = reinterpret_cast<char*>(c.raw.d); c.bytes.n = c.raw.n; c.bytes.cap = c.raw.cap;
                    c.raw.d = nullptr;
                } else c.bytes = fresh(c.sym.n + 1);
            }
            std::atomic<bool> bad{false};
            parallel_for(chain.size(), threads_, [&](size_t x) {
                Chunk& c = cs[chain[x]];
                if (c.failed) return;
                const std::vector<uint8_t>& w = wins[x];
                const size_t missing = kWin - w.size();
                if (!c.as_bytes) c.bytes.n = c.sym.n;
                const uint16_t* s = c.sym.d;
                char* o = c.bytes.d;
                // symbol -> byte through a table (64 KB: the bytes themselves and the window); with amplicon reads most of a chunk is unknowns
                std::unique_ptr<uint8_t[]> lut;
                if (missing == 0 && !c.as_bytes) {
                    lut.reset(new uint8_t[65536]);
                    for (unsigned v = 0; v < 256; v++) lut[v] = (uint8_t)v;
                    memcpy(lut.get() + kUnknown, w.data(), kWin);
                }
                auto resolve = [&](size_t i, const size_t end) {
                    if (c.as_bytes) return;
                    if (lut) { const uint8_t* t = lut.get(); for (; i < end; i++) o[i] = (char)t[s[i]]; return; }
                    for (; i + 16 <= end; i += 16) {                 // sixteen plain bytes at a time where there are
                        uint16_t any = 0;
                        for (int j = 0; j < 16; j++) any |= s[i + j];
                        if (!(any & 0xff00u)) { for (int j = 0; j < 16; j++) o[i + j] = (char)s[i + j]; continue; }
                        for (int j = 0; j < 16; j++) {
                            const uint16_t v = s[i + j];
                            if (v < 256) o[i + j] = (char)v;
                            else { const uint32_t pos = v & (kWin - 1); if (pos < missing) { bad = true; o[i + j] = 0; } else o[i + j] = (char)w[pos - missing]; }
                        }
                    }
                    for (; i < end; i++) {
                        const uint16_t v = s[i];
                        if (v < 256) o[i] = (char)v;
                        else { const uint32_t pos = v & (kWin - 1); if (pos < missing) { bad = true; o[i] = 0; } else o[i] = (char)w[pos - missing]; }
                    }
                };
                // (the CRC of a stretch right behind its bytes, while they are in the cache)
                size_t from = 0;
                for (size_t k = 0; k <= c.ends.size(); k++) {
                    const size_t to = k < c.ends.size() ? (size_t)c.ends[k].at : c.bytes.n;
                    uLong crc = 0;
                    for (size_t at = from; at < to; at += 32768) {
                        const size_t e = std::min(to, at + 32768);
                        resolve(at, e);
                        crc = crc32_z(crc, reinterpret_cast<const Bytef*>(o + at), e - at);
                    }
                    c.seg_crc.push_back((uint32_t)crc);
                    from = to;
                }
            });
            if (bad) throw std::runtime_error("damaged data (distance before the start)");
            if (stats_) {
                const auto t3 = std::chrono::steady_clock::now();
                auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
                size_t n_text = 0;
                for (size_t x : chain) n_text += cs[x].bytes.n;
                fprintf(stderr, "[pargz] wave of %zu chunks (%zu in the stream), %.1f MB of text: boundaries %.1f ms, inflate %.1f ms, windows + bytes + CRC %.1f ms\n", cs.size(),
                        chain.size(), n_text / 1e6, ms(t0, t1), ms(t1, t2), ms(t2, t3));
            }
            for (size_t x = 0; x < chain.size(); x++) {
                Chunk& c = cs[chain[x]];
                if (c.failed) throw std::runtime_error("damaged data (" + c.what + ")");
                size_t from = 0;
                for (size_t k = 0; k <= c.ends.size(); k++) {
                    const size_t to = k < c.ends.size() ? (size_t)c.ends[k].at : c.bytes.n;
                    check.add(c.seg_crc[k], to - from);
                    if (k < c.ends.size() && !check.ends(c.ends[k].crc, c.ends[k].isize)) {
                        // what was read before the damaged member is still delivered, as gzread does
                        c.bytes.n = to;
                        put(std::move(c.bytes));
                        throw std::runtime_error("damaged data (CRC or length mismatch)");
                    }
                    from = to;
                }
                if (!put(std::move(c.bytes))) return;
            }
            const Chunk& last = cs[chain.back()];
            if (last.eof) { finish(""); return; }
            bit = last.stop;
            if (bit >= total_bits) throw std::runtime_error("damaged data (unexpected end)");
        }
    }

    static constexpr size_t kAhead = 512u << 20;   // text kept ready ahead of the reader
    std::string path_;
    unsigned threads_;
    size_t chunk_;
    bool stats_ = getenv("BRONKO_PARGZ_STATS") != nullptr;
    int fd_ = -1;
    const uint8_t* file_ = nullptr;
    uint64_t n_ = 0;
    std::thread producer_;
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<pargz::Piece> q_;
    size_t queued_ = 0;
    bool done_ = false, quit_ = false;
    std::string error_;
    pargz::Piece cur_;
    size_t cur_at_ = 0;
};

}  // namespace bronko
