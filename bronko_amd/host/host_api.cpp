// host_api.cpp -- flat C entry points over the C++ host code (index build / .bkdb codec) so that the Python
// test + bench harness can drive the same code the `bronko` binary runs.  Not the drop-in boundary (that is
// include/bronko_hip.h); errors are returned as NULL / non-zero with bh_last_error().
#include <cstring>
#include <string>
#include <vector>

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

}  // extern "C"
