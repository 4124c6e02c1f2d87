// index.cpp -- see index.hpp.
#include "index.hpp"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <thread>

#include "fastx.hpp"
#include "lcb.hpp"

namespace bronko {

uint64_t Index::total_cells() const {
    uint64_t n = 0;
    for (const auto& f : files) for (const auto& s : f.sequences) n += s.len;
    return n;
}
uint64_t Index::genome_len(size_t f) const {
    uint64_t n = 0;
    for (const auto& s : files[f].sequences) n += s.len;
    return n;
}

std::vector<FastaRecord> read_fasta(const std::string& path) {
    GzLineReader in(path);
    std::vector<FastaRecord> recs;
    std::string line;
    while (in.next(line)) {
        if (!line.empty() && line[0] == '>') {
            recs.emplace_back();
            recs.back().header = line.substr(1);
        } else if (!recs.empty() && !line.empty()) {
            auto& s = recs.back().seq;
            s.insert(s.end(), line.begin(), line.end());
        }
    }
    return recs;
}

namespace {

std::string path_file_stem(const std::string& path) {   // Path::file_stem (build.rs:161-165)
    const size_t slash = path.find_last_of('/');
    std::string base = slash == std::string::npos ? path : path.substr(slash + 1);
    const size_t dot = base.find_last_of('.');
    if (dot != std::string::npos && dot != 0) base.resize(dot);
    return base;
}

std::string first_token(const std::string& h) {          // split_whitespace().next() (build.rs:178-182)
    size_t a = 0;
    while (a < h.size() && isspace((unsigned char)h[a])) a++;
    size_t b = a;
    while (b < h.size() && !isspace((unsigned char)h[b])) b++;
    return h.substr(a, b - a);
}

struct Pair { uint64_t id; BucketInfo e; };

// All (bucket id, BucketInfo) pairs of one genome file, in generation order (sequence, location, j).
void index_file(int k, uint16_t file_id, const FileMeta& fm, std::vector<Pair>& out) {
    uint64_t ids[32];
    size_t total = 0;
    for (const auto& s : fm.sequences) if (s.len >= (uint64_t)k) total += (s.len - k + 1) * (size_t)k;
    out.reserve(total);
    for (size_t sid = 0; sid < fm.sequences.size(); sid++) {
        const SeqMeta& s = fm.sequences[sid];
        if (s.len < (uint64_t)k) continue;   // upstream slices seq[i..i+k] and would panic; nothing to index
        const uint64_t mask = kmer_mask(k);
        uint64_t fwd = 0;
        for (int i = 0; i < k - 1; i++) fwd = (fwd << 2) | nt_to_bits(s.seq[i]);
        for (uint64_t i = 0; i + k <= s.len; i++) {
            fwd = ((fwd << 2) | nt_to_bits(s.seq[i + k - 1])) & mask;     // kmer_to_u64 of seq[i..i+k]
            const Canon c = canonical_u64(fwd, k);                         // build.rs:193
            assign_buckets(c.kmer, k, ids);                                // build.rs:194
            for (int j = 0; j < k; j++) {                                  // build.rs:196-204
                Pair p;
                p.id = ids[j];
                std::memset(&p.e, 0, sizeof p.e);
                p.e.file_id = file_id; p.e.seq_id = (uint8_t)sid; p.e.location = (uint32_t)i;
                p.e.idx = (uint8_t)j; p.e.canonical = c.rc ? 1 : 0;
                out.push_back(p);
            }
        }
    }
}

void finish(Index& ix, std::vector<Pair>& pairs) {
    // stable: keeps file order, then generation order inside a bucket (build.rs:223-228 appends per file)
    std::stable_sort(pairs.begin(), pairs.end(), [](const Pair& a, const Pair& b) { return a.id < b.id; });
    ix.entries.resize(pairs.size());
    for (size_t i = 0; i < pairs.size(); i++) {
        if (i == 0 || pairs[i].id != pairs[i - 1].id) { ix.ids.push_back(pairs[i].id); ix.off.push_back(i); }
        ix.entries[i] = pairs[i].e;
    }
    ix.off.push_back(pairs.size());
}

}  // namespace

Index build_indexes_mem(int k, std::vector<FileMeta> files, int threads) {
    if (files.size() > 65536) throw std::runtime_error("more than 65536 genome files (file_id is u16)");
    for (const auto& f : files)
        if (f.sequences.size() > 256) throw std::runtime_error(f.name + ": more than 256 sequences (seq_id is u8)");
    Index ix;
    ix.k = k; ix.meta_k = k;
    std::vector<std::vector<Pair>> per_file(files.size());
    const int nt = std::max(1, std::min<int>(threads, (int)files.size()));
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; t++)
        pool.emplace_back([&, t] {
            for (size_t f = t; f < files.size(); f += nt) index_file(k, (uint16_t)f, files[f], per_file[f]);
        });
    for (auto& th : pool) th.join();
    size_t total = 0;
    for (auto& v : per_file) total += v.size();
    std::vector<Pair> pairs;
    pairs.reserve(total);
    for (auto& v : per_file) { pairs.insert(pairs.end(), v.begin(), v.end()); std::vector<Pair>().swap(v); }
    ix.files = std::move(files);
    finish(ix, pairs);
    return ix;
}

std::vector<FileMeta> read_genomes(const std::vector<std::string>& genomes) {
    std::vector<FileMeta> files(genomes.size());
    for (size_t f = 0; f < genomes.size(); f++) {
        std::vector<FastaRecord> recs;
        try { recs = read_fasta(genomes[f]); }
        catch (const std::exception& e) { throw std::runtime_error(std::string(e.what()) + " | Failed to parse fasta file: " + genomes[f]); }
        files[f].name = path_file_stem(genomes[f]);
        for (auto& r : recs) {
            SeqMeta sm;
            sm.name = first_token(r.header);
            sm.len = r.seq.size();
            sm.seq = std::move(r.seq);
            files[f].sequences.push_back(std::move(sm));
        }
    }
    return files;
}

Index build_indexes(int k, const std::vector<std::string>& genomes, int threads) {
    return build_indexes_mem(k, read_genomes(genomes), threads);
}

// ---- .bkdb: bincode 2 standard config = little endian, varint integers ----------------------
namespace {

struct Writer {
    FILE* fp;
    void raw(const void* p, size_t n) { if (n && fwrite(p, 1, n, fp) != n) throw std::runtime_error("write failed"); }
    void u8(uint8_t v) { raw(&v, 1); }
    void varint(uint64_t v) {
        uint8_t b[9];
        if (v < 251) { b[0] = (uint8_t)v; raw(b, 1); }
        else if (v <= 0xffffu) { b[0] = 251; for (int i = 0; i < 2; i++) b[1 + i] = (uint8_t)(v >> (8 * i)); raw(b, 3); }
        else if (v <= 0xffffffffull) { b[0] = 252; for (int i = 0; i < 4; i++) b[1 + i] = (uint8_t)(v >> (8 * i)); raw(b, 5); }
        else { b[0] = 253; for (int i = 0; i < 8; i++) b[1 + i] = (uint8_t)(v >> (8 * i)); raw(b, 9); }
    }
    void str(const std::string& s) { varint(s.size()); raw(s.data(), s.size()); }
};

struct Reader {
    const uint8_t* p; const uint8_t* end;
    [[noreturn]] void bad() const { throw std::runtime_error("unexpected end of data"); }
    uint8_t u8() { if (p >= end) bad(); return *p++; }
    uint64_t varint() {
        const uint8_t b = u8();
        if (b < 251) return b;
        int nb;
        switch (b) { case 251: nb = 2; break; case 252: nb = 4; break; case 253: nb = 8; break; default: throw std::runtime_error("unsupported varint marker"); }
        if (end - p < nb) bad();
        uint64_t v = 0;
        for (int i = 0; i < nb; i++) v |= (uint64_t)p[i] << (8 * i);
        p += nb;
        return v;
    }
    std::string str() {
        const uint64_t n = varint();
        if ((uint64_t)(end - p) < n) bad();
        std::string s((const char*)p, (size_t)n);
        p += n;
        return s;
    }
};

}  // namespace

void save_index(const Index& ix, const std::string& path) {
    FILE* fp = fopen(path.c_str(), "wb");
    if (!fp) throw std::runtime_error("File path " + path + " not valid");
    try {
        Writer w{fp};
        w.varint((uint64_t)ix.k);                                   // BronkoIndex.k
        w.varint(ix.ids.size());                                    // global_index: map length
        for (size_t b = 0; b < ix.ids.size(); b++) {                // ascending id (upstream: hash order)
            w.varint(ix.ids[b]);
            w.varint(ix.off[b + 1] - ix.off[b]);
            for (uint64_t i = ix.off[b]; i < ix.off[b + 1]; i++) {
                const BucketInfo& e = ix.entries[i];
                w.varint(e.file_id); w.u8(e.seq_id); w.varint(e.location); w.u8(e.idx); w.u8(e.canonical ? 1 : 0);
            }
        }
        w.varint(ix.files.size());                                  // ViralMetadata
        for (const auto& f : ix.files) {
            w.str(f.name);
            w.varint(f.sequences.size());
            for (const auto& s : f.sequences) {
                w.str(s.name);
                w.varint(s.len);
                w.varint(s.seq.size());
                w.raw(s.seq.data(), s.seq.size());
            }
        }
        w.varint((uint64_t)ix.meta_k);
    } catch (...) { fclose(fp); throw; }
    if (fclose(fp) != 0) throw std::runtime_error("write failed: " + path);
}

Index load_index(const std::string& path) {
    FILE* fp = fopen(path.c_str(), "rb");
    if (!fp) throw std::runtime_error("Failed to open file '" + path + "'");
    std::vector<uint8_t> buf;
    {
        fseek(fp, 0, SEEK_END);
        const long sz = ftell(fp);
        fseek(fp, 0, SEEK_SET);
        buf.resize(sz > 0 ? (size_t)sz : 0);
        const size_t got = buf.empty() ? 0 : fread(buf.data(), 1, buf.size(), fp);
        fclose(fp);
        if (got != buf.size()) throw std::runtime_error("Failed to read Bronko Index from '" + path + "'");
    }
    try {
        Reader r{buf.data(), buf.data() + buf.size()};
        Index ix;
        ix.k = (int)r.varint();
        const uint64_t map_len = r.varint();
        std::vector<Pair> pairs;
        for (uint64_t m = 0; m < map_len; m++) {
            const uint64_t key = r.varint();
            const uint64_t cnt = r.varint();
            for (uint64_t i = 0; i < cnt; i++) {
                Pair p;
                p.id = key;
                std::memset(&p.e, 0, sizeof p.e);
                p.e.file_id = (uint16_t)r.varint();
                p.e.seq_id = r.u8();
                p.e.location = (uint32_t)r.varint();
                p.e.idx = r.u8();
                p.e.canonical = r.u8();
                pairs.push_back(p);
            }
        }
        const uint64_t n_files = r.varint();
        ix.files.resize(n_files);
        for (auto& f : ix.files) {
            f.name = r.str();
            f.sequences.resize(r.varint());
            for (auto& s : f.sequences) {
                s.name = r.str();
                s.len = r.varint();
                const uint64_t n = r.varint();
                if ((uint64_t)(r.end - r.p) < n) r.bad();
                s.seq.assign(r.p, r.p + n);
                r.p += n;
                if (s.len != n) throw std::runtime_error("sequence length field disagrees with its data");
            }
        }
        ix.meta_k = (int)r.varint();
        if (r.p != r.end) throw std::runtime_error("trailing bytes");
        finish(ix, pairs);
        return ix;
    } catch (const std::exception& e) {
        throw std::runtime_error("Failed to read Bronko Index from '" + path + "': " + e.what());
    }
}

}  // namespace bronko
