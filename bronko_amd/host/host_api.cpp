// host_api.cpp -- flat C entry points over the C++ host code (index build / .bkdb codec) so that the Python
// test + bench harness can drive the same code the `bronko` binary runs.  Not the drop-in boundary (that is
// include/bronko_hip.h); errors are returned as NULL / non-zero with bh_last_error().
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "caller.hpp"
#include "index.hpp"

namespace {
thread_local std::string g_err;
}

extern "C" {

const char* bh_last_error(void) { return g_err.c_str(); }

void* bh_index_build(int k, const char* const* paths, int n, int threads) {
    try {
        std::vector<std::string> g(paths, paths + n);
        return new bronko::Index(bronko::build_indexes(k, g, threads));
    } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}

// files given in memory: one sequence list per file (names/seqs flattened in (file, seq) order)
void* bh_index_build_mem(int k, int n_files, const char* const* file_names, const int* n_seqs, const char* const* seq_names,
                         const uint8_t* const* seqs, const uint64_t* seq_lens, int threads) {
    try {
        std::vector<bronko::FileMeta> files(n_files);
        size_t q = 0;
        for (int f = 0; f < n_files; f++) {
            files[f].name = file_names[f];
            for (int s = 0; s < n_seqs[f]; s++, q++) {
                bronko::SeqMeta sm;
                sm.name = seq_names[q];
                sm.len = seq_lens[q];
                sm.seq.assign(seqs[q], seqs[q] + seq_lens[q]);
                files[f].sequences.push_back(std::move(sm));
            }
        }
        return new bronko::Index(bronko::build_indexes_mem(k, std::move(files), threads));
    } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}

void* bh_index_load(const char* path) {
    try { return new bronko::Index(bronko::load_index(path)); }
    catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}

int bh_index_save(const void* h, const char* path) {
    try { bronko::save_index(*static_cast<const bronko::Index*>(h), path); return 0; }
    catch (const std::exception& e) { g_err = e.what(); return -1; }
}

void bh_index_free(void* h) { delete static_cast<bronko::Index*>(h); }

#define IX static_cast<const bronko::Index*>(h)
int bh_index_k(const void* h) { return IX->k; }
int bh_index_meta_k(const void* h) { return IX->meta_k; }
uint64_t bh_index_n_buckets(const void* h) { return IX->ids.size(); }
uint64_t bh_index_n_entries(const void* h) { return IX->entries.size(); }
const uint64_t* bh_index_bucket_ids(const void* h) { return IX->ids.data(); }
const uint64_t* bh_index_bucket_off(const void* h) { return IX->off.data(); }
const void* bh_index_entries(const void* h) { return IX->entries.data(); }
int bh_index_n_files(const void* h) { return (int)IX->files.size(); }
const char* bh_index_file_name(const void* h, int f) { return IX->files[f].name.c_str(); }
int bh_index_n_seqs(const void* h, int f) { return (int)IX->files[f].sequences.size(); }
const char* bh_index_seq_name(const void* h, int f, int s) { return IX->files[f].sequences[s].name.c_str(); }
uint64_t bh_index_seq_len(const void* h, int f, int s) { return IX->files[f].sequences[s].len; }
const uint8_t* bh_index_seq(const void* h, int f, int s) { return IX->files[f].sequences[s].seq.data(); }
uint64_t bh_index_total_cells(const void* h) { return IX->total_cells(); }
#undef IX

// ---- caller stages (so that tests can drive the same code the `bronko` binary runs) ---------------------------
struct bh_call_params {   // mirrors bronko::CallParams
    int32_t k; double min_af; int32_t no_end_filter, no_strand_filter, no_strand_balance_filter; double strand_balance_ratio;
    uint64_t n_per_strand; double strand_odds_max; uint64_t min_depth, min_variant_depth; double variant_multiplier;
};

int bh_pick_best_genome(const void* h, const uint64_t* stats, const uint8_t* present) {
    const auto* ix = static_cast<const bronko::Index*>(h);
    const size_t nf = ix->files.size();
    return bronko::pick_best_genome(*ix, std::vector<uint64_t>(stats, stats + nf * 3), std::vector<uint8_t>(present, present + nf));
}

void bh_baseline_noise_max(const uint64_t* fwd4, const uint64_t* rev4, uint64_t len, double* out) {
    const std::vector<double> v = bronko::baseline_noise_max(fwd4, rev4, len);
    std::memcpy(out, v.data(), v.size() * sizeof(double));
}

// call_variants on the given arrays + writers.  summary = {n_records, n_major, n_minor}; cov = {breadth, depth}.
int bh_call_and_write(const void* h, int file_id, const uint64_t* fwd_depth, const uint64_t* rev_depth, const uint64_t* fwd_nk,
                      const uint64_t* rev_nk, const bh_call_params* cp, const char* vcf_path, const char* reads_path,
                      const char* pileup_path, uint64_t* summary, double* cov) {
    try {
        const auto* ix = static_cast<const bronko::Index*>(h);
        const size_t n = ix->total_cells() * 4;
        bronko::Pileup p;
        p.fwd_depth.assign(fwd_depth, fwd_depth + n); p.rev_depth.assign(rev_depth, rev_depth + n);
        p.fwd_nk.assign(fwd_nk, fwd_nk + n); p.rev_nk.assign(rev_nk, rev_nk + n);
        bronko::CallParams c;
        c.k = cp->k; c.min_af = cp->min_af; c.no_end_filter = cp->no_end_filter; c.no_strand_filter = cp->no_strand_filter;
        c.no_strand_balance_filter = cp->no_strand_balance_filter; c.strand_balance_ratio = cp->strand_balance_ratio;
        c.n_per_strand = cp->n_per_strand; c.strand_odds_max = cp->strand_odds_max; c.min_depth = cp->min_depth;
        c.min_variant_depth = cp->min_variant_depth; c.variant_multiplier = cp->variant_multiplier;
        const bronko::CallSummary cs = bronko::call_variants(*ix, file_id, p, c);
        if (vcf_path) bronko::write_vcf(vcf_path, reads_path ? reads_path : "", *ix, file_id, cs.records);
        if (pileup_path) bronko::write_pileup_tsv(pileup_path, *ix, file_id, p);
        if (summary) { summary[0] = cs.records.size(); summary[1] = cs.n_major; summary[2] = cs.n_minor; }
        if (cov) { cov[0] = cs.breadth; cov[1] = cs.depth; }
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

void bh_clean_sample_id(const char* path, char* buf, size_t n) {
    const std::string s = bronko::clean_sample_id(path);
    snprintf(buf, n, "%s", s.c_str());
}

}  // extern "C"
