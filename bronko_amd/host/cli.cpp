// cli.cpp -- the `bronko` command line: `bronko build` and `bronko call` with the reference's flag names,
// defaults, validations, exit codes and output files (drop-in surface).
//
// Reference: /root/reference/src/main.rs:14-29 (banner, dispatch, elapsed), src/cli.rs:29-166 (flags),
// src/consts.rs (defaults), src/build.rs:62-120 (build + its checks), src/call.rs:30-136 (call checks),
// src/call.rs:151-402 (per-sample orchestration).  The k-mer counting + map_kmers stages of the per-sample loop
// run on the GPU through the C ABI of include/bronko_hip.h; everything else here is host code.
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include <sys/stat.h>
#include <unistd.h>

#include <hip/hip_runtime_api.h>   // streams and buffers of the one-sample-over-several-GPUs path (RCCL collectives on the engines' streams)
#include <rccl/rccl.h>

#include "../../include/bronko_hip.h"
#include "caller.hpp"
#include "fastq_pack.hpp"
#include "fastx.hpp"
#include "index.hpp"

namespace {

using namespace bronko;

const char* kVersion = "0.1.0";   // Cargo.toml:3 / consts.rs:1

int g_level = 2;   // 0 error, 1 warn, 2 info, 3 debug, 4 trace (simple_logger levels, call.rs:31-43)
void logf(int lvl, const char* tag, const char* target, const std::string& msg) {
    if (lvl > g_level) return;
    printf("%-5s [%s] %s\n", tag, target, msg.c_str());
    fflush(stdout);
}
#define LOG_ERROR(t, m) logf(0, "ERROR", t, m)
#define LOG_WARN(t, m) logf(1, "WARN", t, m)
#define LOG_INFO(t, m) logf(2, "INFO", t, m)
#define LOG_TRACE(t, m) logf(4, "TRACE", t, m)

// `error!(..); std::process::exit(1)`.  Every thread leaves through _exit() (streams flushed first; the first error wins):
// exit() would run the static destructors -- the HIP runtime's among them -- under the lane and completion threads' HIP calls,
// from the main thread as much as from one of them, and end in a crash or a hang instead of exit code 1.
std::mutex g_die_mu;
[[noreturn]] void die(const char* target, const std::string& msg) {
    g_die_mu.lock();   // (never released: a second failing thread waits here while the first one ends the process)
    LOG_ERROR(target, msg);
    fflush(stdout);
    fflush(stderr);
    _exit(1);
}

bool ends_with(const std::string& s, const char* suf) {
    const size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}
bool check_fastq(const std::string& f) {   // util.rs:4-15
    return ends_with(f, ".fq") || ends_with(f, ".fastq") || ends_with(f, ".fq.gz") || ends_with(f, "fastq.gz") ||
           ends_with(f, "fnq") || ends_with(f, "fnq.gz");
}
bool check_fasta(const std::string& f) {   // util.rs:17-28
    return ends_with(f, ".fa") || ends_with(f, ".fasta") || ends_with(f, ".fa.gz") || ends_with(f, "fasta.gz") ||
           ends_with(f, "fna") || ends_with(f, "fna.gz");
}

// ---- argument parsing (clap derive surface of cli.rs) -------------------------------------------------------
struct Args {
    std::string mode;
    std::vector<std::string> genomes, reads, first_pairs, second_pairs;
    bool has_genomes = false;
    std::string db;
    bool has_db = false;
    long kmer = 21;                 // consts.rs:3
    long min_kmers = 3;             // consts.rs:5
    bool use_full_kmer = false;
    long n_fixed = 2;               // consts.rs:17
    double min_af = 0.03;
    bool no_end_filter = false, no_strand_filter = false, no_strand_balance_filter = false;
    double balance_ratio = 0.1;
    long n_per_strand = 2;
    double strand_odds = 6.0;
    long min_depth = 300;
    long min_variant_depth = 3;
    double noise_multiplier = 1.5;
    std::string output;             // default depends on the mode
    bool pileup = false, alignment = false, keep_kmer_info = false;
    long threads = 4;
    bool debug = false, verbose = false;
};

[[noreturn]] void usage(int code) {
    fputs("Usage: bronko <COMMAND>\n\nCommands:\n"
          "  build  Create an bronko index of existing viral references for a given species\n"
          "  call   Perform rapid viral variant calling of viral sequencing data\n\n"
          "bronko build -g <GENOMES>... [-k <KMER>] [-o <OUTPUT>] [-t <THREADS>] [--debug] [--verbose]\n"
          "bronko call (-g <GENOMES>... | -d <DB>) [-r <READS>...] [-1 <R1>... -2 <R2>...] [-k <KMER>] [--min-kmers N]\n"
          "            [--use-full-kmer] [--n-fixed N] [--min-af F] [--no-end-filter] [--no-strand-filter]\n"
          "            [--no-strand-balance-filter] [--balance-ratio F] [--n-per-strand N] [--strand_odds F]\n"
          "            [--min-depth N] [--min-variant-depth N] [--noise-multiplier F] [-o <DIR>] [--pileup]\n"
          "            [--alignment] [--keep-kmer-info] [-t <THREADS>] [--debug] [--verbose]\n", stderr);
    exit(code);
}

long to_long(const std::string& opt, const std::string& v) {
    char* end = nullptr;
    const long x = strtol(v.c_str(), &end, 10);
    if (v.empty() || *end || x < 0) { fprintf(stderr, "error: invalid value '%s' for '%s'\n", v.c_str(), opt.c_str()); exit(2); }
    return x;
}
double to_double(const std::string& opt, const std::string& v) {
    char* end = nullptr;
    const double x = strtod(v.c_str(), &end);
    if (v.empty() || *end) { fprintf(stderr, "error: invalid value '%s' for '%s'\n", v.c_str(), opt.c_str()); exit(2); }
    return x;
}

Args parse_args(int argc, char** argv) {
    if (argc < 2) usage(2);
    Args a;
    a.mode = argv[1];
    if (a.mode == "--help" || a.mode == "-h" || a.mode == "help") usage(0);
    if (a.mode == "--version" || a.mode == "-V") { printf("bronko %s\n", kVersion); exit(0); }
    if (a.mode != "build" && a.mode != "call") { fprintf(stderr, "error: unrecognized subcommand '%s'\n", a.mode.c_str()); usage(2); }
    if (argc < 3) usage(2);   // arg_required_else_help
    const bool call = a.mode == "call";
    a.output = call ? "bronko_output" : "bronko";   // consts.rs:20-21

    std::vector<std::string> tok;
    for (int i = 2; i < argc; i++) tok.push_back(argv[i]);
    size_t i = 0;
    auto is_flag = [](const std::string& s) { return s.size() >= 2 && s[0] == '-' && !(isdigit((unsigned char)s[1]) && s.size() > 2 && s[1] != '1' && s[1] != '2'); };
    while (i < tok.size()) {
        std::string opt = tok[i++], inline_val;
        bool has_inline = false;
        if (opt.rfind("--", 0) == 0) {
            const size_t eq = opt.find('=');
            if (eq != std::string::npos) { inline_val = opt.substr(eq + 1); opt.resize(eq); has_inline = true; }
        } else if (opt.size() > 2 && opt[0] == '-') {   // -k21 / -k=21
            inline_val = opt.substr(opt[2] == '=' ? 3 : 2);
            opt.resize(2);
            has_inline = true;
        }
        if (opt == "-h" || opt == "--help") usage(0);
        if (opt == "-V" || opt == "--version") { printf("bronko %s\n", kVersion); exit(0); }
        auto one = [&]() -> std::string {
            if (has_inline) return inline_val;
            if (i >= tok.size()) { fprintf(stderr, "error: a value is required for '%s'\n", opt.c_str()); exit(2); }
            return tok[i++];
        };
        auto many = [&](std::vector<std::string>& dst) {   // num_args = 1..
            if (has_inline) dst.push_back(inline_val);
            const size_t before = dst.size();
            while (i < tok.size() && !(tok[i].size() >= 2 && tok[i][0] == '-')) dst.push_back(tok[i++]);
            if (!has_inline && dst.size() == before) { fprintf(stderr, "error: a value is required for '%s'\n", opt.c_str()); exit(2); }
        };
        (void)is_flag;
        if (opt == "-g" || opt == "--genomes") { many(a.genomes); a.has_genomes = true; }
        else if (opt == "-k" || opt == "--kmer-size") a.kmer = to_long(opt, one());
        else if (opt == "-o" || opt == "--output") a.output = one();
        else if (opt == "-t" || opt == "--threads") a.threads = to_long(opt, one());
        else if (opt == "--debug") a.debug = true;
        else if (opt == "--verbose") a.verbose = true;
        else if (!call) { fprintf(stderr, "error: unexpected argument '%s' found\n", opt.c_str()); exit(2); }
        else if (opt == "-d" || opt == "--db") { a.db = one(); a.has_db = true; }
        else if (opt == "-r" || opt == "--reads") many(a.reads);
        else if (opt == "-1" || opt == "--first-pairs") many(a.first_pairs);
        else if (opt == "-2" || opt == "--second-pairs") many(a.second_pairs);
        else if (opt == "--min-kmers") a.min_kmers = to_long(opt, one());
        else if (opt == "--use-full-kmer") a.use_full_kmer = true;
        else if (opt == "--n-fixed") a.n_fixed = to_long(opt, one());
        else if (opt == "--min-af") a.min_af = to_double(opt, one());
        else if (opt == "--no-end-filter") a.no_end_filter = true;
        else if (opt == "--no-strand-filter") a.no_strand_filter = true;
        else if (opt == "--no-strand-balance-filter") a.no_strand_balance_filter = true;
        else if (opt == "--balance-ratio") a.balance_ratio = to_double(opt, one());
        else if (opt == "--n-per-strand") a.n_per_strand = to_long(opt, one());
        else if (opt == "--strand_odds") a.strand_odds = to_double(opt, one());
        else if (opt == "--min-depth") a.min_depth = to_long(opt, one());
        else if (opt == "--min-variant-depth") a.min_variant_depth = to_long(opt, one());
        else if (opt == "--noise-multiplier") a.noise_multiplier = to_double(opt, one());
        else if (opt == "--pileup") a.pileup = true;
        else if (opt == "--alignment") a.alignment = true;
        else if (opt == "--keep-kmer-info") a.keep_kmer_info = true;
        else { fprintf(stderr, "error: unexpected argument '%s' found\n", opt.c_str()); exit(2); }
    }
    return a;
}

void init_logging(const Args& a) { g_level = a.verbose ? 4 : a.debug ? 3 : 2; }

void check_common(const Args& a, const char* target) {
    if (a.kmer % 2 != 1 || a.kmer > 31 || a.kmer < 15)   // build.rs:79, call.rs:46
        die(target, "Invalid kmer size, must be odd and between [15-31]");
    const long avail = (long)std::thread::hardware_concurrency();
    if (a.threads <= 0) die(target, "Number of threads must be greater than 0");
    if (avail > 0 && a.threads > avail)
        die(target, "You requested " + std::to_string(a.threads) + " threads but only have " + std::to_string(avail) + " available on your system");
}

// ---- bronko build (build.rs:102-120) ------------------------------------------------------------------------
// build_indexes (build.rs:145-231): on the GPU when one is visible (bk_build_index: one thread per k-mer, stable device sort), on
// host threads otherwise -- `bronko build` also runs on a machine without a GPU; the result is the same index either way
// (BRONKO_BUILD_ON_HOST=1 forces the host path).
Index build_index_any(const char* T, int k, const std::vector<std::string>& genomes, int threads) {
    std::vector<FileMeta> files = read_genomes(genomes);
    if (!getenv("BRONKO_BUILD_ON_HOST") && bk_device_count() > 0) {
        if (files.size() > 65536) throw std::runtime_error("more than 65536 genome files (file_id is u16)");
        std::vector<int32_t> n_seqs;
        std::vector<uint64_t> seq_lens;
        std::vector<const uint8_t*> seqs;
        for (const auto& f : files) {
            if (f.sequences.size() > 256) throw std::runtime_error(f.name + ": more than 256 sequences (seq_id is u8)");
            n_seqs.push_back((int32_t)f.sequences.size());
            for (const auto& s : f.sequences) { seq_lens.push_back(s.len); seqs.push_back(s.seq.data()); }
        }
        if (seqs.empty()) { seq_lens.push_back(0); seqs.push_back(nullptr); }
        int device = 0;
        if (const char* dv = getenv("BRONKO_DEVICE")) device = atoi(dv);
        bk_built_index b{};
        if (bk_build_index(k, (int32_t)files.size(), n_seqs.data(), seq_lens.data(), seqs.data(), device, &b) == BK_OK) {
            LOG_TRACE(T, "index built on the GPU: " + std::to_string(b.n_buckets) + " buckets, " + std::to_string(b.n_entries) + " entries");
            Index ix;
            ix.k = k; ix.meta_k = k;
            ix.ids.assign(b.bucket_ids, b.bucket_ids + b.n_buckets);
            ix.off.assign(b.bucket_off, b.bucket_off + b.n_buckets + 1);
            ix.entries.resize(b.n_entries);
            static_assert(sizeof(BucketInfo) == sizeof(bk_bucket_info), "BucketInfo layouts must agree");
            if (b.n_entries) std::memcpy(ix.entries.data(), b.entries, b.n_entries * sizeof(BucketInfo));
            bk_built_index_free(&b);
            ix.files = std::move(files);
            return ix;
        }
        LOG_WARN(T, std::string("GPU index build unavailable (") + bk_build_last_error() + "), building on the host");
    }
    return build_indexes_mem(k, std::move(files), threads);
}

int run_build(const Args& a) {
    const char* T = "bronko::build";
    init_logging(a);
    if (a.kmer % 2 != 1 || a.kmer > 31 || a.kmer < 15) die(T, "Invalid kmer size, must be odd and between [15-31]");
    for (const auto& g : a.genomes)
        if (!check_fasta(g)) die(T, g + " does not appear to be a fasta file (must be .fa(.gz)/.fasta(.gz)/.fna(.gz))");
    check_common(a, T);
    LOG_INFO(T, "Building indexes from fasta files");
    Index ix;
    try { ix = build_index_any(T, (int)a.kmer, a.genomes, (int)a.threads); }
    catch (const std::exception& e) { die(T, std::string(e.what()) + " | Reference failed to build"); }
    const std::string out = a.output + ".bkdb";
    LOG_INFO(T, "Saving index to " + out);
    try { save_index(ix, out); }
    catch (const std::exception& e) { die(T, std::string(e.what()) + " | Unable to save index"); }
    return 0;
}

// ---- bronko call ----------------------------------------------------------------------------------------------
void check_call_args(const Args& a) {   // call.rs:30-136
    const char* T = "bronko::call";
    if (a.kmer % 2 != 1 || a.kmer > 31 || a.kmer < 15) die(T, "Invalid kmer size, must be odd and between [15-31]");
    for (const auto& f : a.reads)
        if (!check_fastq(f)) die(T, f + " does not appear to be a fastq file (must be .fq(.gz)/.fastq(.gz)/.fnq(.gz))");
    if (a.has_genomes && a.has_db) die(T, "Please provide either a db or the genomes you would like to index, not both.");
    if (!a.has_genomes && !a.has_db) die(T, "Please provide either a db or the genomes you would like to index.");
    for (const auto& g : a.genomes)
        if (!check_fasta(g)) die(T, g + " does not appear to be a fasta file (must be .fa(.gz)/.fasta(.gz)/.fna(.gz))");
    check_common(a, T);
    if (a.min_af < 0.01) LOG_WARN(T, "Minimum allele frequency set below 0.01, more false positive variants will be returned. We suggest setting this to a more realistic threshold (0.01-0.05)");
    else if (a.min_af > 1.0) die(T, "Minimum allele frequency set above 1, please set between 0-1 (recommended between 0.01-0.05)");
    else if (a.min_af >= 0.5) LOG_WARN(T, "Minimum allele frequency set equal to or greater than 0.5, no minor variants will be returned");
    if (a.n_per_strand <= 0) LOG_WARN(T, "Number of kmers per strand set to 0, this is equivalent to no strand filtering");
    else if (a.n_per_strand >= a.kmer) die(T, "Number of kmers per strand set >= k, please set lower value (recommended 2-4, default 2)");
    else if (a.n_per_strand >= 5) LOG_WARN(T, "Number of kmers per strand set very high, only strongly supported variants will be returned");
    if (a.balance_ratio < 0.0) die(T, "Strand balance ratio is set to below 0, must be between 0.0 and 1.0");
    else if (a.balance_ratio > 1.0) die(T, "Strand balance ratio is set above 1, must be between 0.0 and 1.0");
    else if (a.balance_ratio == 1.0) LOG_WARN(T, "Strand balance ratio is set to 1, all variants will pass this filter");
    if (a.noise_multiplier < 1.0) die(T, "Noise multiplier for variant detection is set to below 1.0, must be greater than 1.0 (recommended between 1.3-2.0)");
    else if (a.noise_multiplier > 2.0) LOG_WARN(T, "Strand balance ratio is set above 2, may experience a drop in recall (we recommend ~1.5)");
    else if (a.noise_multiplier == 1.0) LOG_WARN(T, "Noise multiplier for variant detection set to 1.0, all variants will pass this filter");
    if (a.first_pairs.size() != a.second_pairs.size()) die(T, "Number of paired end sequences do not match, exiting.");
}

struct Engine {
    bk_engine* e = nullptr;
    ~Engine() { if (e) bk_engine_destroy(e); }
};

void hip_check(int rc, const char* what) {
    if (rc != 0) die("bronko::call", std::string(what) + ": " + bk_last_error());
}

// The mate files of one sample: FASTQ(.gz) -> batches of sequence lines -> bk_push_reads_ascii (packed on the GPU,
// asynchronous: the next batch is parsed while the previous ones are copied, packed and scanned).  Every mate file is
// inflated and parsed by its own host thread (upstream runs the two KMC processes of a pair concurrently too,
// call.rs:301-307); the engine is only ever called from this thread.  Returns reads seen.
struct FastqBatch {
    std::string buf; std::vector<uint64_t> off{0};   // sequence lines back to back (the line loop: streams, one thread) ...
    PackedBatch packed; bool is_packed = false;      // ... or 2-bit records, parsed and packed on several threads (fastq_pack.hpp)
    bool last = false; std::string error;
    size_t bytes() const { return is_packed ? packed.bytes() : buf.size(); }
};
// The sequence text that samples read ahead of their turn hold in their queues, all of them together: counted as it is queued
// (a batch's real bytes, not an estimate from the compressed size: amplicon FASTQ inflates 8-10x), released as lanes consume.
struct AheadGate {
    std::mutex m;
    std::condition_variable cv;
    uint64_t held = 0, budget = 0;
};
struct BatchQueue {
    std::mutex m;
    std::condition_variable cv;
    std::deque<FastqBatch> q;
    std::vector<FastqBatch> spare;   // consumed batches, handed back: their 40 MB buffers are reused instead of being unmapped and
                                     // mapped again (with dozens of lanes the page faults of fresh buffers cost more than the parsing)
    static constexpr size_t kDepth = 3;
    // A sample read ahead of its turn (ReadAhead below): the whole file may wait here as long as the gate has room.  Once a lane has
    // claimed the sample the queue is an ordinary one again (kDepth batches ahead of the lane) and no longer waits for the gate: the
    // lane must never wait for text that later samples' queues hold.
    AheadGate* gate = nullptr;
    std::atomic<bool> claimed{false};
    std::atomic<bool> abandoned{false};   // nobody will take from this queue any more (a run that ends early): the reader stops
    void put(FastqBatch&& b) {
        const uint64_t sz = b.bytes();
        if (gate) {
            std::unique_lock<std::mutex> gl(gate->m);
            gate->cv.wait(gl, [&] { return abandoned.load() || claimed.load() || gate->held == 0 || gate->held + sz <= gate->budget; });
            if (abandoned.load()) return;
            gate->held += sz;
        }
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return abandoned.load() || (gate && !claimed.load()) || q.size() < kDepth; });
        if (abandoned.load()) return;
        q.push_back(std::move(b));
        cv.notify_all();
    }
    void abandon() {
        abandoned.store(true);
        if (gate) { std::unique_lock<std::mutex> gl(gate->m); gate->cv.notify_all(); }
        { std::unique_lock<std::mutex> lk(m); cv.notify_all(); }
    }
    FastqBatch take() {
        FastqBatch b;
        {
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return !q.empty(); });
            b = std::move(q.front());
            q.pop_front();
            cv.notify_all();
        }
        if (gate) {
            { std::unique_lock<std::mutex> gl(gate->m); gate->held -= std::min<uint64_t>(gate->held, b.bytes()); }
            gate->cv.notify_all();
        }
        return b;
    }
    void claim() {   // a lane takes the sample over
        claimed.store(true);
        if (gate) { std::unique_lock<std::mutex> gl(gate->m); gate->cv.notify_all(); }
        { std::unique_lock<std::mutex> lk(m); cv.notify_all(); }
    }
    void recycle(FastqBatch&& b) {
        std::unique_lock<std::mutex> lk(m);
        if (spare.size() < kDepth + 2) spare.push_back(std::move(b));
    }
    FastqBatch fresh() {
        FastqBatch b;
        {
            std::unique_lock<std::mutex> lk(m);
            if (!spare.empty()) { b = std::move(spare.back()); spare.pop_back(); }
        }
        b.buf.clear(); b.off.clear(); b.off.push_back(0); b.packed.clear(); b.is_packed = false; b.last = false; b.error.clear();
        return b;
    }
};
// threads a FASTQ file's inflate may take (pargz.hpp): -t over the files that are read at the same time (set by call)
unsigned g_inflate_threads = 1;
unsigned g_ahead_inflate_threads = 1;   // ... for the files that are read ahead of their turn: -t over the files ReadAhead has open at once
int g_kmer = 21;   // (set by call: the records a reader thread packs drop runs shorter than k)
void parse_fastq(const std::string& path, BatchQueue& out, unsigned inflate_threads) {
    constexpr uint64_t kBatchReads = 1u << 16;   // (10 MB of bases: the engine pins three staging slots of that size per lane)
    FastqBatch cur;
    try {
        if (inflate_threads > 1) {
            // threads to spare: the file's text is taken apart and 2-bit packed piece by piece on as many threads (fastq_pack.hpp);
            // a few MB of text make a piece, pieces are gathered into batches of a quarter of a million records (a scan launch has
            // a fixed cost: small pushes are slow pushes)
            constexpr uint64_t kBatchRecords = 1u << 18;
            FastqPacker in(path, g_kmer, inflate_threads);
            PackedBatch b;
            cur.is_packed = true;
            while (in.next(b)) {
                if (cur.packed.n_records && (cur.packed.stride != b.stride || cur.packed.n_records + b.n_records > 2 * kBatchRecords)) {
                    out.put(std::move(cur)); cur = out.fresh(); cur.is_packed = true;
                    if (out.abandoned.load()) break;
                }
                if (!cur.packed.n_records) { const uint64_t r = cur.packed.n_reads; cur.packed = std::move(b); cur.packed.n_reads += r; b = PackedBatch(); }
                else {
                    cur.packed.words.insert(cur.packed.words.end(), b.words.begin(), b.words.end());
                    cur.packed.lens.insert(cur.packed.lens.end(), b.lens.begin(), b.lens.end());
                    cur.packed.n_records += b.n_records; cur.packed.n_reads += b.n_reads;
                }
                if (cur.packed.n_records >= kBatchRecords) { out.put(std::move(cur)); cur = out.fresh(); cur.is_packed = true; if (out.abandoned.load()) break; }
            }
        } else {
            GzLineReader in(path, inflate_threads);
            uint64_t n = 0;
            for (uint64_t ln = 0;; ln++) {               // 4-line FASTQ records: @id / sequence / + / quality
                if ((ln & 3) != 1) { if (!in.skip_next()) break; continue; }
                if (!in.append_next(cur.buf)) break;     // (the sequence line goes straight into the batch)
                cur.off.push_back(cur.buf.size());
                if (++n % kBatchReads == 0) { out.put(std::move(cur)); cur = out.fresh(); if (out.abandoned.load()) break; }
            }
        }
    } catch (const std::exception& e) {
        cur = FastqBatch();
        cur.error = e.what();
    }
    cur.last = true;
    out.put(std::move(cur));
}
// The files of the samples to come are read while the index and the engine's tables are being made (seconds with a hundred
// genomes: host work that leaves most cores idle) and while earlier samples are on their way: a manager thread starts the
// readers of sample after sample, `concurrency` files at a time, as long as the text the started samples hold in their queues
// stays within `budget` bytes (AheadGate: real bytes; readers wait when it is full and go on as lanes consume); a lane that
// reaches a sample takes its readers over (claim) or, if they were not started, reads it itself as before.  Inputs that are not
// regular files (a FIFO, /dev/fd/N) are never read ahead: their size is unknown and they can be read once.
struct SampleReaders {
    std::deque<BatchQueue> queues;   // (a deque: BatchQueue holds a mutex and does not move)
    std::vector<std::thread> readers;
};
class ReadAhead {
public:
    ReadAhead(const std::vector<std::vector<std::string>>& samples, unsigned concurrency, uint64_t budget)
        : samples_(samples), state_(samples.size(), 0), held_(samples.size()), concurrency_(std::max(1u, concurrency)) {
        gate_.budget = budget;
        uint64_t all = 0;
        bool regular = true;
        for (const auto& m : samples) { uint64_t n = 0; regular = text_estimate(m, &n) && regular; all += n; }
        covers_all_ = regular && all <= budget;
        manager_ = std::thread([this] { run(); });
    }
    bool covers_all() const { return covers_all_; }   // every sample's text fits the budget (by the estimate): the lanes only push
    void set_concurrency(unsigned n) {                // (few files at a time while the engine's tables are made on the same cores, more behind that)
        { std::unique_lock<std::mutex> lk(m_); concurrency_ = std::max(1u, n); }
        cv_.notify_all();
    }
    ~ReadAhead() {
        { std::unique_lock<std::mutex> lk(m_); stop_ = true; }
        cv_.notify_all();
        if (manager_.joinable()) manager_.join();
        // (readers of samples no lane came for -- a run that ended early -- must not wait for room that nobody will make)
        for (auto& h : held_) if (h) for (auto& q : h->queues) q.abandon();
        for (auto& h : held_) if (h) for (auto& t : h->readers) if (t.joinable()) t.join();
    }
    // the readers of sample i if it is being read ahead; otherwise nullptr, and it will not be
    std::unique_ptr<SampleReaders> claim(size_t i) {
        std::unique_lock<std::mutex> lk(m_);
        if (state_[i] == 1) {
            state_[i] = 2;
            for (auto& q : held_[i]->queues) q.claim();
            return std::move(held_[i]);
        }
        state_[i] = 2;
        return nullptr;
    }
private:
    // bytes of sequence lines a sample's batches will hold, estimated from the files' sizes; false: a mate is not a regular file
    static bool text_estimate(const std::vector<std::string>& mates, uint64_t* out) {
        uint64_t n = 0;
        bool regular = true;
        for (const auto& p : mates) {
            struct stat st;
            if (stat(p.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) { regular = false; continue; }
            n += (uint64_t)st.st_size * 2;   // (FASTQ text is ~3.5x its gzip, the sequence lines half of it)
        }
        *out = n;
        return regular;
    }
    void run() {
        for (size_t i = 0; i < samples_.size(); i++) {
            // room for another sample?  (real bytes: three quarters of the budget held means the readers already started fill the rest)
            for (;;) {
                { std::unique_lock<std::mutex> lk(m_); if (stop_) return; }
                std::unique_lock<std::mutex> gl(gate_.m);
                if (gate_.held <= gate_.budget / 4 * 3) break;
                gate_.cv.wait_for(gl, std::chrono::milliseconds(20));
            }
            std::unique_lock<std::mutex> lk(m_);
            cv_.wait(lk, [&] { return stop_ || active_ + samples_[i].size() <= concurrency_; });
            if (stop_) return;
            if (state_[i] != 0) continue;                 // a lane got there first
            uint64_t need = 0;
            if (!text_estimate(samples_[i], &need)) continue;   // (a stream: its lane reads it)
            auto sr = std::unique_ptr<SampleReaders>(new SampleReaders);
            for (size_t m = 0; m < samples_[i].size(); m++) { sr->queues.emplace_back(); sr->queues.back().gate = &gate_; }
            for (size_t m = 0; m < samples_[i].size(); m++) {
                active_++;
                sr->readers.emplace_back([this, i, m, q = &sr->queues[m]] {
                    parse_fastq(samples_[i][m], *q, g_ahead_inflate_threads);
                    { std::unique_lock<std::mutex> lk2(m_); active_--; }
                    cv_.notify_all();
                });
            }
            state_[i] = 1;
            held_[i] = std::move(sr);
        }
    }
    const std::vector<std::vector<std::string>>& samples_;
    std::vector<int> state_;                              // 0 not started, 1 being read ahead, 2 taken by its lane
    std::vector<std::unique_ptr<SampleReaders>> held_;
    unsigned concurrency_, active_ = 0;
    AheadGate gate_;
    bool stop_ = false, covers_all_ = false;
    std::mutex m_;
    std::condition_variable cv_;
    std::thread manager_;
};

// engs: one engine (the sample's reads all go there) or one per GPU of a sharded sample -- batches are dealt to them in turn
uint64_t push_fastqs(const std::vector<bk_engine*>& engs, const std::vector<std::string>& mates, std::unique_ptr<SampleReaders> ahead = nullptr) {
    const size_t nm = mates.size();
    size_t n_batches = 0;
    std::unique_ptr<SampleReaders> own;
    if (!ahead) {
        own.reset(new SampleReaders);
        for (size_t m = 0; m < nm; m++) own->queues.emplace_back();
        for (size_t m = 0; m < nm; m++) own->readers.emplace_back(parse_fastq, std::cref(mates[m]), std::ref(own->queues[m]), g_inflate_threads);
    }
    SampleReaders& sr = ahead ? *ahead : *own;
    std::deque<BatchQueue>& queues = sr.queues;
    std::vector<std::thread>& readers = sr.readers;
    uint64_t n_reads = 0;
    std::string error;
    std::vector<bool> done(nm, false);
    for (size_t left = nm; left;) {
        for (size_t m = 0; m < nm; m++) {
            if (done[m]) continue;
            FastqBatch b = queues[m].take();
            if (!b.error.empty() && error.empty()) error = b.error;
            if (error.empty() && b.is_packed) {
                if (b.packed.n_records)
                    hip_check(bk_push_reads_packed(engs[n_batches++ % engs.size()], (int)m, b.packed.words.data(), b.packed.stride, b.packed.lens.data(), b.packed.n_records), "bk_push_reads_packed");
                n_reads += b.packed.n_reads;
            } else if (error.empty() && b.off.size() > 1) {
                hip_check(bk_push_reads_ascii(engs[n_batches++ % engs.size()], (int)m, reinterpret_cast<const uint8_t*>(b.buf.data()), b.off.data(), b.off.size() - 1), "bk_push_reads_ascii");
                n_reads += b.off.size() - 1;
            }
            if (b.last) { done[m] = true; left--; }
            else queues[m].recycle(std::move(b));   // (bk_push_reads_ascii has copied it to its pinned ring)
        }
    }
    for (auto& t : readers) t.join();
    if (!error.empty()) throw std::runtime_error(error);
    return n_reads;
}

// ---- one sample over several GPUs (BASELINE config 4: one 200 M-read sample, eight GPUs) ---------------------
// The reads' batches are dealt to one engine per GPU; what is additive -- the k-mer occurrence counter planes -- is
// reduce-scattered by RCCL over xGMI on the engines' own streams (the engine packs a plane to 16- or 32-bit elements first,
// bk_shard_transport), every GPU maps its part (bk_sample_finalize_shard), and the small results are combined: max of the depth
// planes, sums of the #k-mer planes and of the statistics.  Pileups of read shards are never summed (thresholds and max are not
// linear).  One process, one communicator per device (ncclCommInitAll), collectives grouped over the devices.
void nccl_check(ncclResult_t r, const char* what) {
    if (r != ncclSuccess) die("bronko::call", std::string(what) + ": " + ncclGetErrorString(r));
}
void hipx(hipError_t r, const char* what) {
    if (r != hipSuccess) die("bronko::call", std::string(what) + ": " + hipGetErrorString(r));
}
struct ShardGroup {
    std::vector<int> devices;
    std::vector<bk_engine*> engs;
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> streams;
    int n() const { return (int)engs.size(); }
};
// narrowest width at which the reduce-scatter over n ranks is exact (bronko_amd/dist.py::pick_width; include/bronko_hip.h)
int pick_width(uint64_t max_e, uint64_t max_v, int n) {
    if (max_v * (uint64_t)n <= 32767 && max_e < (1ull << 32)) return 16;
    if (std::max(max_e, max_v) * (uint64_t)n <= 2147483647ull) return 32;
    return 64;
}
// between the last push and the finalize of a sample whose batches went to g.engs in turn
void sharded_finalize(ShardGroup& g, int n_mates, uint64_t cells4) {
    const int S = g.n();
    // KMC's distinct / counted k-mer totals (full_kmer_stats): every k-mer that touches no bucket moves to its owner GPU
    {
        std::vector<void*> keys((size_t)S), cnts((size_t)S), rkeys((size_t)S, nullptr), rcnts((size_t)S, nullptr);
        std::vector<std::vector<uint64_t>> off((size_t)S, std::vector<uint64_t>((size_t)S + 1));
        for (int s = 0; s < S; s++) hip_check(bk_kmer_table_partition(g.engs[(size_t)s], S, &keys[(size_t)s], &cnts[(size_t)s], off[(size_t)s].data()), "bk_kmer_table_partition");
        std::vector<uint64_t> n_in((size_t)S, 0);
        for (int r = 0; r < S; r++) for (int s = 0; s < S; s++) n_in[(size_t)r] += off[(size_t)s][(size_t)r + 1] - off[(size_t)s][(size_t)r];
        for (int r = 0; r < S; r++) {
            hipx(hipSetDevice(g.devices[(size_t)r]), "hipSetDevice");
            hipx(hipMalloc(&rkeys[(size_t)r], std::max<uint64_t>(n_in[(size_t)r], 1) * 8), "hipMalloc");
            hipx(hipMalloc(&rcnts[(size_t)r], std::max<uint64_t>(n_in[(size_t)r], 1) * 4), "hipMalloc");
        }
        std::vector<uint64_t> at((size_t)S, 0);   // fill of each receiver
        nccl_check(ncclGroupStart(), "ncclGroupStart");
        for (int s = 0; s < S; s++)
            for (int r = 0; r < S; r++) {
                const uint64_t n = off[(size_t)s][(size_t)r + 1] - off[(size_t)s][(size_t)r], o = off[(size_t)s][(size_t)r];
                if (!n) continue;
                uint64_t* dk = static_cast<uint64_t*>(rkeys[(size_t)r]) + at[(size_t)r];
                uint32_t* dc = static_cast<uint32_t*>(rcnts[(size_t)r]) + at[(size_t)r];
                at[(size_t)r] += n;
                if (s == r) {   // (its own group: a copy on its stream)
                    hipx(hipSetDevice(g.devices[(size_t)s]), "hipSetDevice");
                    hipx(hipMemcpyAsync(dk, static_cast<uint64_t*>(keys[(size_t)s]) + o, n * 8, hipMemcpyDeviceToDevice, g.streams[(size_t)s]), "hipMemcpyAsync");
                    hipx(hipMemcpyAsync(dc, static_cast<uint32_t*>(cnts[(size_t)s]) + o, n * 4, hipMemcpyDeviceToDevice, g.streams[(size_t)s]), "hipMemcpyAsync");
                    continue;
                }
                nccl_check(ncclSend(static_cast<uint64_t*>(keys[(size_t)s]) + o, n, ncclUint64, r, g.comms[(size_t)s], g.streams[(size_t)s]), "ncclSend");
                nccl_check(ncclSend(static_cast<uint32_t*>(cnts[(size_t)s]) + o, n, ncclUint32, r, g.comms[(size_t)s], g.streams[(size_t)s]), "ncclSend");
                nccl_check(ncclRecv(dk, n, ncclUint64, s, g.comms[(size_t)r], g.streams[(size_t)r]), "ncclRecv");
                nccl_check(ncclRecv(dc, n, ncclUint32, s, g.comms[(size_t)r], g.streams[(size_t)r]), "ncclRecv");
            }
        nccl_check(ncclGroupEnd(), "ncclGroupEnd");
        for (int r = 0; r < S; r++) hip_check(bk_kmer_table_replace(g.engs[(size_t)r], rkeys[(size_t)r], rcnts[(size_t)r], n_in[(size_t)r]), "bk_kmer_table_replace");
        for (int r = 0; r < S; r++) {   // (the table was rebuilt from them on the engine's stream)
            hipx(hipSetDevice(g.devices[(size_t)r]), "hipSetDevice");
            hipx(hipStreamSynchronize(g.streams[(size_t)r]), "hipStreamSynchronize");
            hipx(hipFree(rkeys[(size_t)r]), "hipFree"); hipx(hipFree(rcnts[(size_t)r]), "hipFree");
        }
    }
    for (int m = 0; m < n_mates; m++) {
        // the narrowest exact width: the largest E count and |V element| over all GPUs' planes
        uint64_t max_e = 0, max_v = 0;
        std::vector<void*> dmax((size_t)S);
        for (int s = 0; s < S; s++) hip_check(bk_shard_measure(g.engs[(size_t)s], m, &dmax[(size_t)s]), "bk_shard_measure");
        for (int s = 0; s < S; s++) {
            uint64_t mx[2] = {0, 0};
            hipx(hipSetDevice(g.devices[(size_t)s]), "hipSetDevice");
            hipx(hipMemcpyAsync(mx, dmax[(size_t)s], sizeof mx, hipMemcpyDeviceToHost, g.streams[(size_t)s]), "hipMemcpyAsync");
            hipx(hipStreamSynchronize(g.streams[(size_t)s]), "hipStreamSynchronize");
            max_e = std::max(max_e, mx[0]); max_v = std::max(max_v, mx[1]);
        }
        int width = pick_width(max_e, max_v, S);
        std::vector<void*> send((size_t)S), recv((size_t)S);
        uint64_t part_bytes = 0;
        for (int s = 0; s < S; s++) {
            int rc = bk_shard_transport(g.engs[(size_t)s], m, S, width, &send[(size_t)s], &part_bytes, &recv[(size_t)s]);
            if (rc != 0 && width == 16 && s == 0) { width = 32; rc = bk_shard_transport(g.engs[0], m, S, width, &send[0], &part_bytes, &recv[0]); }   // (16 does not shrink this plane at S shards)
            hip_check(rc, "bk_shard_transport");
        }
        const ncclDataType_t dt = width == 64 ? ncclInt64 : ncclInt32;
        const size_t count = (size_t)(part_bytes / (width == 64 ? 8 : 4));
        nccl_check(ncclGroupStart(), "ncclGroupStart");
        for (int s = 0; s < S; s++) nccl_check(ncclReduceScatter(send[(size_t)s], recv[(size_t)s], count, dt, ncclSum, g.comms[(size_t)s], g.streams[(size_t)s]), "ncclReduceScatter");
        nccl_check(ncclGroupEnd(), "ncclGroupEnd");
        for (int s = 0; s < S; s++) hip_check(bk_shard_received(g.engs[(size_t)s], m, s, S, width), "bk_shard_received");
    }
    for (int s = 0; s < S; s++) hip_check(bk_sample_finalize_shard(g.engs[(size_t)s], n_mates, s, S), "bk_sample_finalize_shard");
    // the small results: depth = max, #k-mers and statistics add up
    std::vector<void*> pile((size_t)S), sums((size_t)S);
    uint64_t n_sums = 0;
    for (int s = 0; s < S; s++) {
        hip_check(bk_pileup_device_ptr(g.engs[(size_t)s], &pile[(size_t)s]), "bk_pileup_device_ptr");
        hip_check(bk_shard_sums_device_ptr(g.engs[(size_t)s], &sums[(size_t)s], &n_sums), "bk_shard_sums_device_ptr");
    }
    nccl_check(ncclGroupStart(), "ncclGroupStart");
    for (int s = 0; s < S; s++) {
        uint64_t* p = static_cast<uint64_t*>(pile[(size_t)s]);
        nccl_check(ncclAllReduce(p, p, (size_t)(2 * cells4), ncclUint64, ncclMax, g.comms[(size_t)s], g.streams[(size_t)s]), "ncclAllReduce");
        nccl_check(ncclAllReduce(p + 2 * cells4, p + 2 * cells4, (size_t)(2 * cells4), ncclUint64, ncclSum, g.comms[(size_t)s], g.streams[(size_t)s]), "ncclAllReduce");
        nccl_check(ncclAllReduce(sums[(size_t)s], sums[(size_t)s], (size_t)n_sums, ncclUint64, ncclSum, g.comms[(size_t)s], g.streams[(size_t)s]), "ncclAllReduce");
    }
    nccl_check(ncclGroupEnd(), "ncclGroupEnd");
    for (int s = 0; s < S; s++) hip_check(bk_sample_merge_shards(g.engs[(size_t)s]), "bk_sample_merge_shards");
}

int run_call(const Args& a) {
    const char* T = "bronko::call";
    init_logging(a);
    check_call_args(a);
    LOG_TRACE(T, "k=" + std::to_string(a.kmer) + ", threads=" + std::to_string(a.threads));
    if (mkdir(a.output.c_str(), 0777) != 0 && errno != EEXIST) {
        // create_dir_all: create missing parents too
        std::string partial;
        for (size_t i = 0; i <= a.output.size(); i++) {
            if (i == a.output.size() || a.output[i] == '/') { if (!partial.empty()) mkdir(partial.c_str(), 0777); }
            if (i < a.output.size()) partial += a.output[i];
        }
        struct stat st;
        if (stat(a.output.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) die(T, "Unable to create outputs in output directory 2");
    }

    // the samples, in input order; their files are read ahead from here on (ReadAhead): the index and the engine's tables take
    // seconds to make with many genomes, and a lane's next sample need not wait for its previous one's reads
    std::vector<std::vector<std::string>> samples;
    for (const auto& r : a.reads) samples.push_back({r});
    for (size_t i = 0; i < a.first_pairs.size(); i++) samples.push_back({a.first_pairs[i], a.second_pairs[i]});
    {   // inflate threads per file (pargz.hpp) for what is read ahead: -t over the files that will be open at once
        const size_t files = samples.size() * (a.first_pairs.empty() ? 1 : 2), open_files = std::max<size_t>(1, std::min<size_t>(files, (size_t)a.threads / 2));
        g_ahead_inflate_threads = open_files > 8 ? 1u : (unsigned)std::max<size_t>(1, std::min<size_t>(64, (size_t)a.threads / open_files));
        if (const char* it = getenv("BRONKO_INFLATE_THREADS")) g_ahead_inflate_threads = (unsigned)std::max(1, atoi(it));
        g_inflate_threads = g_ahead_inflate_threads;
    }
    g_kmer = (int)a.kmer;   // (a database of another k is refused below: the readers may pack before it is read)
    std::unique_ptr<ReadAhead> ahead;
    if (!getenv("BRONKO_NO_READ_AHEAD")) {
        const uint64_t ram = (uint64_t)sysconf(_SC_PHYS_PAGES) * (uint64_t)sysconf(_SC_PAGE_SIZE);
        ahead.reset(new ReadAhead(samples, (unsigned)std::max<long>(2, a.threads / 8), std::min<uint64_t>(ram / 4, 32ull << 30)));
    }

    Index ix;
    if (a.has_genomes) {                                            // call.rs:170-178
        LOG_INFO(T, "Creating bronko index from provided reference genomes");
        try { ix = build_index_any(T, (int)a.kmer, a.genomes, (int)a.threads); }
        catch (const std::exception& e) { die(T, std::string(e.what()) + " | Reference failed to build"); }
    } else {                                                        // call.rs:179-200
        LOG_INFO(T, "Reading in provided bronko index");
        try { ix = load_index(a.db); }
        catch (const std::exception& e) { die(T, e.what()); }
        if (ix.k != a.kmer)
            die(T, "Database k is not the same as provided, please set -k to " + std::to_string(ix.k) + " or build a new index");
    }
    // (a few genomes: the engine's tables are made in a fraction of a second, the cores are the readers' from here on; with many
    // the readers stay few until the engines stand -- bk_engine_create runs on all cores for seconds)
    if (ahead && ix.files.size() <= 8) ahead->set_concurrency((unsigned)std::max<long>(2, a.threads / 2));

    // decoded index -> GPU engine(s) (include/bronko_hip.h).  Samples are independent (call.rs:212 / :297 handle them one after
    // the other), so whole samples are dealt to *lanes* in turn -- no collective.  A lane is a host thread that ingests its
    // samples (gunzip + parse are host work: ~1 M reads/s per FASTQ file, a thousand times slower than the scan behind them)
    // into its own pair of engines; the lanes of one device share that device's tables (bk_engine_fork), every device holds
    // its own copy.  BRONKO_DEVICES=0,1,.. names the devices (default: all visible ones; naming a device twice doubles its
    // lanes), BRONKO_DEVICE=d the single device of earlier versions, BRONKO_LANES=n the lanes per device (default: -t / 2
    // over the devices, at most 16 -- a 256-thread host inflates 8 gzip streams side by side at full speed and 32 at half --
    // and no more than fit six tenths of the device's free memory: a lane keeps two samples'
    // counter planes there -- 0.2 GB for one SARS-CoV-2 genome, 9 GB for a hundred at k = 31).
    std::vector<int> devices;
    if (const char* dl = getenv("BRONKO_DEVICES")) {
        for (const char* q = dl; *q;) {
            char* end = nullptr;
            const long d = strtol(q, &end, 10);
            if (end == q) break;
            devices.push_back((int)d);
            q = *end == ',' ? end + 1 : end;
        }
    } else if (const char* dv = getenv("BRONKO_DEVICE")) {
        devices.push_back(atoi(dv));
    } else {
        const int nd = bk_device_count();
        for (int d = 0; d < std::max(nd, 1); d++) devices.push_back(d);
    }
    if (devices.empty()) devices.push_back(0);
    const size_t n_samples_total = a.reads.size() + a.first_pairs.size();
    // Fewer samples than GPUs (BASELINE config 4: ONE 200 M-read sample, eight GPUs): a sample's batches are dealt to all of them and
    // the counter planes are reduce-scattered by RCCL (sharded_finalize above).  BRONKO_SHARD=1 / 0 forces / forbids it (1 with a single
    // GPU runs every collective on a communicator of one rank).  The shard count is a power of two (it divides 64).
    std::vector<int> shard_devices;
    for (int d : devices) if (std::find(shard_devices.begin(), shard_devices.end(), d) == shard_devices.end()) shard_devices.push_back(d);
    bool shard_mode = shard_devices.size() >= 2 && n_samples_total < shard_devices.size();
    if (const char* sh = getenv("BRONKO_SHARD")) shard_mode = atoi(sh) != 0;
    { size_t S = 1; while (S * 2 <= std::min<size_t>(shard_devices.size(), 64)) S *= 2; shard_devices.resize(S); }
    if (devices.size() > std::max<size_t>(n_samples_total, 1)) devices.resize(std::max<size_t>(n_samples_total, 1));   // no more lanes than samples
    auto make_engine = [&](int device, Engine& out, bool selected_only = true) {
        std::vector<int32_t> n_seqs;
        std::vector<uint64_t> seq_lens;
        std::vector<const uint8_t*> seqs;
        for (const auto& f : ix.files) {
            n_seqs.push_back((int32_t)f.sequences.size());
            for (const auto& s : f.sequences) { seq_lens.push_back(s.len); seqs.push_back(s.seq.data()); }
        }
        bk_index_desc d{};
        d.k = ix.k; d.n_buckets = ix.ids.size(); d.bucket_ids = ix.ids.data(); d.bucket_off = ix.off.data();
        d.entries = reinterpret_cast<const bk_bucket_info*>(ix.entries.data()); d.n_entries = ix.entries.size();
        d.n_files = (int32_t)ix.files.size(); d.n_seqs = n_seqs.data(); d.seq_lens = seq_lens.data(); d.seqs = seqs.data();
        bk_params p;
        bk_params_default(&p);
        p.n_fixed = (int32_t)a.n_fixed; p.use_full_kmer = a.use_full_kmer ? 1 : 0; p.ci = (uint64_t)a.min_kmers;
        p.pileup_selected_only = selected_only ? 1 : 0;   // calls, pileup TSV and overview read the selected genome's rows only (call.rs:229-293)
        p.full_kmer_stats = 1;   // KMC's "unique counted k-mers" feeds num_unmapped_kmers and the <0.2 warning (call.rs:242-248)
        if (const char* tl = getenv("BRONKO_KMER_TABLE_LOG2")) p.kmer_table_log2 = (uint32_t)atoi(tl);
        p.device = device;
        hip_check(bk_engine_create(&d, &p, &out.e), "bk_engine_create");
    };
    ShardGroup shards;
    std::vector<Engine> shard_engines;
    if (shard_mode) {
        // one engine per GPU, every genome's rows (the two-pass selected-only finalize cannot be sharded: the selection needs the
        // statistics of all parts first); an index so large that its planes are kept sparse cannot be sharded at all
        shard_engines.resize(shard_devices.size());
        std::vector<std::thread> th;
        for (size_t q = 0; q < shard_devices.size(); q++) th.emplace_back([&, q] { make_engine(shard_devices[q], shard_engines[q], ix.files.size() <= 1); });
        for (auto& t : th) t.join();
        if (!bk_can_shard(shard_engines[0].e)) {
            LOG_WARN(T, "The index keeps its counter planes sparse: a sample cannot be sharded over GPUs, whole samples go to the GPUs in turn");
            shard_mode = false;
            shard_engines.clear();
        }
    }
    if (shard_mode) {
        shards.devices = shard_devices;
        shards.comms.resize(shard_devices.size());
        nccl_check(ncclCommInitAll(shards.comms.data(), (int)shard_devices.size(), shard_devices.data()), "ncclCommInitAll");
        for (auto& en : shard_engines) { shards.engs.push_back(en.e); shards.streams.push_back(static_cast<hipStream_t>(bk_engine_get_stream(en.e))); }
        LOG_INFO(T, "Every sample's reads go to " + std::to_string(shard_devices.size()) + " GPU(s); RCCL reduce-scatter of the k-mer counter planes");
        devices.clear();   // (no whole-sample lanes)
    }
    struct Lane { int device = 0; int parent = -1; Engine eng, fork; std::vector<size_t> mine; };   // parent: the lane whose engine built the device's tables
    std::vector<Engine> first(devices.size());   // the first engine of every device named: its tables, and what a sample's state weighs
    std::vector<int> first_dev(devices.size(), -1);
    {
        std::vector<std::thread> th;   // table construction is host work: the devices' engines are created side by side
        for (size_t l = 0; l < devices.size(); l++) {
            bool seen = false;
            for (size_t q = 0; q < l; q++) seen = seen || devices[q] == devices[l];
            if (!seen) { first_dev[l] = devices[l]; th.emplace_back([&make_engine, &devices, &first, l] { make_engine(devices[l], first[l]); }); }
        }
        for (auto& t : th) t.join();
    }
    if (!devices.empty()) {
        size_t per_device = std::min<size_t>(16, std::max<size_t>(1, (size_t)a.threads / 2 / devices.size()));
        for (size_t q = 0; q < first.size(); q++) {
            if (!first[q].e) continue;
            // what a lane's two engines keep on the device: two counter planes, the deferred lists and touch lists (as much
            // again), pileups, the k-mer statistics table (it starts at 0.8 GB) -- against six tenths of what the device has free
            const double per_engine = 4.0 * 8.0 * (double)bk_counter_len(first[q].e) + 64.0 * (double)bk_total_cells(first[q].e) + 1.0e9;
            uint64_t free_b = 0, total_b = 0;
            if (bk_device_memory(first_dev[q], &free_b, &total_b) != 0) free_b = 64ull << 30;
            per_device = std::min<size_t>(per_device, std::max<size_t>(1, (size_t)(0.6 * (double)free_b / (2.0 * per_engine))));
        }
        // with every file read ahead of its turn a lane only pushes, finalizes and writes: two per device are what pays (32 x 1 M reads
        // 2.9 -> 2.4 s, 64 samples against 100 strains 13.9 -> 11.6 s; a lane's engines and their forks are not free)
        if (ahead && ahead->covers_all()) per_device = std::min<size_t>(per_device, 2);
        if (const char* nl = getenv("BRONKO_LANES")) per_device = std::max<size_t>(1, (size_t)atoi(nl));
        std::vector<int> lanes_on;
        for (size_t r = 0; r < per_device; r++)                       // device-major rounds: every device gets a lane before any gets two
            for (int d : devices) lanes_on.push_back(d);
        if (lanes_on.size() > std::max<size_t>(n_samples_total, 1)) lanes_on.resize(std::max<size_t>(n_samples_total, 1));
        devices.swap(lanes_on);
    }
    std::vector<Lane> lanes(devices.size());
    for (size_t l = 0; l < lanes.size(); l++) {
        lanes[l].device = devices[l];
        for (size_t q = 0; q < l && lanes[l].parent < 0; q++)
            if (lanes[q].device == devices[l]) lanes[l].parent = (int)(lanes[q].parent < 0 ? q : (size_t)lanes[q].parent);
    }
    if (lanes.size() > 1) {
        std::string names;
        for (int d : devices) names += (names.empty() ? "" : ",") + std::to_string(d);
        LOG_INFO(T, "Samples go to " + std::to_string(lanes.size()) + " GPU lanes in turn (devices " + names + ")");
    }
    for (auto& ln : lanes)
        for (size_t q = 0; q < first.size() && ln.parent < 0; q++)
            if (first[q].e && !ln.eng.e && first_dev[q] == ln.device) std::swap(ln.eng.e, first[q].e);
    {
        std::vector<std::thread> th;   // (a fork allocates and zeroes a sample's planes: gigabytes with a large index)
        for (auto& ln : lanes)
            if (ln.parent >= 0) th.emplace_back([&lanes, &ln] { hip_check(bk_engine_fork(lanes[(size_t)ln.parent].eng.e, &ln.eng.e), "bk_engine_fork"); });
        for (auto& t : th) t.join();
    }

    {
        // KMC reads a sample's files with all of -t (call.rs:1166-1181); here -t is shared by the files that are open at once: the
        // lanes' samples (one being read per lane) times their mate files
        // (with many files open at once the files are the parallelism: 16 lanes x 4 inflate threads measured slower than 16 x 1 --
        // 3.5 s against 2.8 s for 32 x 1 M reads -- while 7 lanes x 9 threads, a hundred-genome index, gain 19.2 -> 13.8 s)
        const size_t open_files = std::max<size_t>(1, lanes.size()) * (a.first_pairs.empty() ? 1 : 2);
        g_inflate_threads = open_files > 8 ? 1u : (unsigned)std::max<size_t>(1, std::min<size_t>(64, (size_t)a.threads / open_files));
        if (const char* it = getenv("BRONKO_INFLATE_THREADS")) g_inflate_threads = (unsigned)std::max(1, atoi(it));
        if (g_inflate_threads > 1) LOG_INFO(T, "gzip input is inflated on " + std::to_string(g_inflate_threads) + " threads per file");
        if (ahead) ahead->set_concurrency((unsigned)std::max<long>(2, a.threads / 2));   // (the engines are made: the cores are the readers')
    }
    CallParams cp;
    cp.k = (int)a.kmer; cp.min_af = a.min_af; cp.no_end_filter = a.no_end_filter; cp.no_strand_filter = a.no_strand_filter;
    cp.no_strand_balance_filter = a.no_strand_balance_filter; cp.strand_balance_ratio = a.balance_ratio;
    cp.n_per_strand = (uint64_t)a.n_per_strand; cp.strand_odds_max = a.strand_odds; cp.min_depth = (uint64_t)a.min_depth;
    cp.min_variant_depth = (uint64_t)a.min_variant_depth; cp.variant_multiplier = a.noise_multiplier;

    const size_t n_files = ix.files.size();
    const uint64_t cells4 = ix.total_cells() * 4;
    std::vector<OverviewRow> overview(n_samples_total);     // by sample, in input order
    std::vector<SampleCalls> all_calls(a.alignment ? n_samples_total : 0);   // --alignment

    // one sample = one -r file (call.rs:213-293) or one R1/R2 pair (call.rs:298-386); outputs are named after R1.
    // A sample has two halves: ingest (parse the FASTQ files, push the reads: host-bound, the scan runs behind it) and
    // complete (finalize on the GPU, download, pick the genome, call variants, write the files).  With several samples the two
    // halves of consecutive samples overlap: sample i+1 is ingested into a second engine on the same device tables
    // (bk_engine_fork) while a worker thread completes sample i.  Results are reported in input order.
    auto ingest = [&](const std::vector<bk_engine*>& engs, const std::vector<std::string>& mates, size_t sample_id) -> uint64_t {
        for (bk_engine* e : engs) hip_check(bk_sample_begin(e), "bk_sample_begin");
        uint64_t total_reads = 0;
        try { total_reads = push_fastqs(engs, mates, ahead ? ahead->claim(sample_id) : nullptr); }
        catch (const std::exception& ex) { die(T, ex.what()); }
        LOG_INFO(T, std::to_string(total_reads) + " reads counted from " + mates[0]);
        return total_reads;
    };
    auto complete = [&](bk_engine* e, const std::vector<std::string>& mates, size_t sample_id, bool finalized = false) {
        const int n_mates = (int)mates.size();
        Pileup p;
        std::vector<uint64_t> stats((size_t)n_mates * n_files * 3), kstats((size_t)n_mates * 4);
        std::vector<uint8_t> present((size_t)n_mates * n_files);
        LOG_INFO(T, "Mapping kmers to all genomes (" + mates[0] + ")");
        // finalize, then reference selection + baseline noise + variant calls, all on the device and asynchronous
        // (bk_sample_call); the pileup arrays only travel when --pileup wants them written
        if (!finalized) hip_check(bk_sample_finalize(e, n_mates), "bk_sample_finalize");   // (a sharded sample: sharded_finalize has done it)
        bk_call_params dcp;
        bk_call_params_default(&dcp);
        dcp.k = cp.k; dcp.no_end_filter = cp.no_end_filter; dcp.no_strand_filter = cp.no_strand_filter;
        dcp.no_strand_balance_filter = cp.no_strand_balance_filter; dcp.min_af = cp.min_af; dcp.strand_balance_ratio = cp.strand_balance_ratio;
        dcp.strand_odds_max = cp.strand_odds_max; dcp.variant_multiplier = cp.variant_multiplier; dcp.n_per_strand = cp.n_per_strand;
        dcp.min_depth = cp.min_depth; dcp.min_variant_depth = cp.min_variant_depth;
        hip_check(bk_sample_call(e, n_mates, &dcp), "bk_sample_call");
        if (a.pileup) { p.fwd_depth.resize(cells4); p.rev_depth.resize(cells4); }
        hip_check(bk_sample_download(e, n_mates, a.pileup ? p.fwd_depth.data() : nullptr, a.pileup ? p.rev_depth.data() : nullptr, nullptr, nullptr,
                                     stats.data(), present.data(), kstats.data()), "bk_sample_download");
        uint64_t longest = 1;   // at most three alternative bases per position of the selected genome
        for (size_t f = 0; f < n_files; f++) longest = std::max<uint64_t>(longest, ix.genome_len(f));
        std::vector<bk_call_record> drecs((size_t)(3 * longest));
        bk_call_summary summ{};
        hip_check(bk_sample_download_calls(e, &summ, drecs.data(), drecs.size()), "bk_sample_download_calls");
        p.stats.assign(n_files * 3, 0);
        p.present.assign(n_files, 0);
        uint64_t kept = 0;   // KMC "No. of unique counted k-mers", summed over mate files (call.rs:336)
        bool kept_exact = true;
        for (int m = 0; m < n_mates; m++) {                          // pick_best_genome_paired sums R1 + R2 (call.rs:457-474)
            for (size_t i = 0; i < n_files * 3; i++) p.stats[i] += stats[(size_t)m * n_files * 3 + i];
            for (size_t f = 0; f < n_files; f++) p.present[f] |= present[(size_t)m * n_files + f];
            if (kstats[(size_t)m * 4 + 3] == ~0ull) kept_exact = false; else kept += kstats[(size_t)m * 4 + 3];
        }
        // (the engine grows the table with the sample; only a sample with more than 2^30 distinct erroneous k-mers gets here)
        if (!kept_exact) die(T, "k-mer statistics table overflowed: num_unmapped_kmers cannot be reported for " + mates[0]);
        LOG_INFO(T, "Selecting the most representative genome");
        const int best = summ.file_id;
        if (best < 0) die(T, "Unable to pick a best genome");
        const std::string& gname = ix.files[best].name;
        LOG_INFO(T, "Selected a representative genome: " + gname);
        const uint64_t n_perfect = p.stats[(size_t)best * 3], n_variant = p.stats[(size_t)best * 3 + 1];
        const uint64_t n_unmapped = (kept_exact && kept >= n_perfect + n_variant) ? kept - n_perfect - n_variant : 0;   // call.rs:242,336
        if (kept_exact && kept > 0 && (double)(n_variant + n_perfect) / (double)kept < 0.2)                              // call.rs:246-248
            LOG_WARN(T, "Percent of kmers found is very low for this reference, suggesting lack of a representative reference, a bad sequencing run, contamination in sample, or some other issue");
        LOG_INFO(T, "Mapped " + std::to_string(n_perfect) + "/" + std::to_string(kept) + " kmers perfectly (" +
                        std::to_string(p.stats[(size_t)best * 3 + 2]) + " unique among refs), " + std::to_string(n_variant) + "/" +
                        std::to_string(kept) + " had a variant");
        LOG_INFO(T, "Calling variants for " + gname);
        CallSummary cs;
        cs.n_major = summ.n_major; cs.n_minor = summ.n_minor;
        cs.breadth = (double)summ.covered / (double)summ.positions;               // call.rs:1144
        cs.depth = (double)summ.coverage / (double)summ.covered;                  // call.rs:1145 (NaN when nothing is covered)
        for (uint64_t i = 0; i < std::min<uint64_t>(summ.n_records, drecs.size()); i++) {
            const bk_call_record& r = drecs[i];
            // SOR as printed: the reference's expression on the host's libm (the device's ln made the decision; the two agree
            // to the last ulps, the printed three decimals are the host's)
            double sor = cp.strand_odds_max + 1.0;
            if (!cp.no_strand_filter) {                                            // call.rs:1059-1096
                const double fa = (double)r.fwd_ref + 1.0, fb = (double)r.rev_ref + 1.0, fc = (double)r.fwd_alt + 1.0, fd = (double)r.rev_alt + 1.0;
                const double min_strand = std::fmin(fa + fc, fb + fd) / (fa + fb + fc + fd);
                if (!cp.no_strand_balance_filter || min_strand >= cp.strand_balance_ratio) {
                    const double q = (fa * fd) / (fb * fc);
                    sor = std::log(q + 1.0 / q) + std::log(std::fmin(fa, fb) / std::fmax(fa, fb)) - std::log(std::fmin(fc, fd) / std::fmax(fc, fd));
                } else {
                    sor = -1.0;
                }
            }
            cs.records.push_back(VcfRecord{r.seq_id, r.pos, r.ref_base, r.alt_base, r.fwd_ref, r.rev_ref, r.fwd_alt, r.rev_alt, r.depth, r.af, sor});
        }
        LOG_INFO(T, "Called " + std::to_string(cs.n_major) + " major variants, " + std::to_string(cs.n_minor) + " minor above maf = " + std::to_string(a.min_af));
        const std::string stem = clean_sample_id(mates[0]);
        try {
            if (a.pileup) { LOG_INFO(T, "Writing output to pileup"); write_pileup_tsv(a.output + "/" + stem + ".tsv", ix, best, p); }
            LOG_INFO(T, "Writing output to VCF");
            write_vcf(a.output + "/" + stem + ".vcf", mates[0], ix, best, cs.records);
        } catch (const std::exception& ex) { die(T, ex.what()); }
        overview[sample_id] = OverviewRow{mates[0], gname, cs.n_major, cs.n_minor, cs.breadth, cs.depth, n_perfect, n_variant, n_unmapped};
        if (a.alignment) all_calls[sample_id] = SampleCalls{mates[0], gname, cs.breadth, cs.records};
    };

    if (shard_mode) {
        for (size_t i = 0; i < samples.size(); i++) {
            const auto& mates = samples[i];
            LOG_INFO(T, mates.size() == 1 ? "Processing " + mates[0] : "Processing paired reads " + mates[0] + ", " + mates[1]);
            ingest(shards.engs, mates, i);
            sharded_finalize(shards, (int)mates.size(), cells4);
            complete(shards.engs[0], mates, i, true);
        }
        for (auto c : shards.comms) nccl_check(ncclCommDestroy(c), "ncclCommDestroy");
        shard_engines.clear();
    }
    for (size_t i = 0; i < samples.size() && !lanes.empty(); i++) lanes[i % lanes.size()].mine.push_back(i);
    auto run_lane = [&](Lane& ln) {
        if (ln.mine.size() > 1) hip_check(bk_engine_fork(ln.eng.e, &ln.fork.e), "bk_engine_fork");
        std::thread worker;     // completes the lane's previous sample
        for (size_t n = 0; n < ln.mine.size(); n++) {
            const size_t i = ln.mine[n];
            const auto& mates = samples[i];
            LOG_INFO(T, mates.size() == 1 ? "Processing " + mates[0] : "Processing paired reads " + mates[0] + ", " + mates[1]);
            bk_engine* e = (n & 1) ? ln.fork.e : ln.eng.e;   // (its previous sample, n - 2, was completed before sample n - 1's worker started)
            ingest(std::vector<bk_engine*>{e}, mates, i);
            if (worker.joinable()) worker.join();
            worker = std::thread([&complete, e, &mates, i] { complete(e, mates, i); });
        }
        if (worker.joinable()) worker.join();
        if (ln.fork.e) { bk_engine_destroy(ln.fork.e); ln.fork.e = nullptr; }   // (the fork goes before its parent)
    };
    if (lanes.size() == 1) run_lane(lanes[0]);
    else if (!lanes.empty()) {
        std::vector<std::thread> th;
        for (auto& ln : lanes) th.emplace_back([&run_lane, &ln] { run_lane(ln); });
        for (auto& t : th) t.join();
    }
    {   // (forks go before the engine they were forked from; side by side: releasing dozens of engines one after the other takes a second)
        std::vector<std::thread> th;
        for (auto& ln : lanes)
            if (ln.parent >= 0 && ln.eng.e) th.emplace_back([&ln] { bk_engine_destroy(ln.eng.e); ln.eng.e = nullptr; });
        for (auto& t : th) t.join();
    }
    LOG_INFO(T, "Printing overview");
    try { write_overview_tsv(a.output + "/bronko_overview.tsv", overview); }
    catch (const std::exception& e) { die(T, e.what()); }
    LOG_INFO(T, "All samples processed successfully");
    if (a.alignment) {                                                                  // call.rs:394-397
        LOG_INFO(T, "Building alignment(s)");
        try { write_alignments(a.output, ix, all_calls, [](const std::string& m) { LOG_INFO("bronko::call", m); }); }
        catch (const std::exception& e) { die(T, e.what()); }
    }
    LOG_INFO(T, "");
    LOG_INFO(T, "bronko complete!");
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    printf("bronko v%s\nMI355X (gfx950) k-mer -> pileup engine; drop-in for treangenlab/bronko's build / call\n\n", kVersion);
    fflush(stdout);
    const auto t0 = std::chrono::steady_clock::now();
    // (HIP maps a process's streams onto four hardware queues by default: with more engines in flight than that -- lanes and their
    // forks -- two streams share a queue and their kernels wait for each other; set before the runtime initialises, an explicit
    // setting in the environment wins)
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    const Args a = parse_args(argc, argv);
    const int rc = a.mode == "build" ? run_build(a) : run_call(a);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    fprintf(stderr, "\nbronko v%s finished in %gs\n", kVersion, dt);   // main.rs:28
    return rc;
}
