// caller.hpp -- host stages after the pileup: reference selection, baseline-noise filter, variant calling and
// the output writers (product code).  Reference behaviour: /root/reference/src/call.rs:422-502 (selection),
// :792-967 (noise), :969-1150 (calls), :648-695 (pileup TSV), :698-732 (overview TSV), :735-774 (VCF);
// file naming /root/reference/src/util.rs:30-50.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "index.hpp"

namespace bronko {

// The four OutputData arrays of one sample (call.rs:1235-1239,1451-1454), flat in (file, seq, pos, base) order,
// plus map_kmers' per-genome statistics (call.rs:1272), already summed over mate files.
struct Pileup {
    std::vector<uint64_t> fwd_depth, rev_depth, fwd_nk, rev_nk;   // total_cells * 4 each
    std::vector<uint64_t> stats;                                   // n_files * 3: perfect, variant, unique
    std::vector<uint8_t>  present;                                 // n_files
};

struct CallParams {                  // CallArgs fields that reach calling (cli.rs:92-135; defaults consts.rs)
    int      k = 21;
    double   min_af = 0.03;
    bool     no_end_filter = false;
    bool     no_strand_filter = false;
    bool     no_strand_balance_filter = false;
    double   strand_balance_ratio = 0.1;
    uint64_t n_per_strand = 2;
    double   strand_odds_max = 6.0;
    uint64_t min_depth = 300;
    uint64_t min_variant_depth = 3;
    double   variant_multiplier = 1.5;
};

struct VcfRecord {                   // call.rs:776-789
    int      seq_id;
    uint64_t pos;                    // 1-based
    uint8_t  ref_base, alt_base;     // 2-bit codes
    uint64_t fwd_ref, rev_ref, fwd_alt, rev_alt, depth;
    double   af, sor;
};

struct CallSummary {
    std::vector<VcfRecord> records;
    uint64_t n_major = 0, n_minor = 0;
    double breadth = 0.0, depth = 0.0;
};

// call.rs:422-450 / :452-502.  Ties are broken towards the lower file id (upstream: hash-map order).  -1 = None.
int pick_best_genome(const Index& ix, const std::vector<uint64_t>& stats, const std::vector<uint8_t>& present);

// call.rs:799-967 -- only Noise.max is consumed downstream (call.rs:1107); returned per position.
std::vector<double> baseline_noise_max(const uint64_t* fwd4, const uint64_t* rev4, uint64_t len);

// call.rs:969-1150 over the sequences of `file_id` in metadata order.
CallSummary call_variants(const Index& ix, int file_id, const Pileup& p, const CallParams& prm);

std::string clean_sample_id(const std::string& path);                               // util.rs:30-50
void write_vcf(const std::string& out_path, const std::string& reads_path_as_given, const Index& ix, int file_id,
               const std::vector<VcfRecord>& recs);                                 // call.rs:735-774
void write_pileup_tsv(const std::string& out_path, const Index& ix, int file_id, const Pileup& p);  // call.rs:648-695

struct OverviewRow {                 // call.rs:138-149
    std::string filename, selected_genome;
    uint64_t n_major, n_minor;
    double breadth, depth;
    uint64_t n_perfect, n_variant, n_unmapped;
};
void write_overview_tsv(const std::string& out_path, const std::vector<OverviewRow>& rows);  // call.rs:698-732

// --alignment (call.rs:504-628): per selected genome with at least three samples of breadth >= 0.90, OUT/<genome>.mfa =
// the columns of all positions at which some sample has a major variant (AF >= 0.5), reference row first, then one row per
// sample (its alternative base where it has a major variant, the reference base elsewhere).  Columns are ordered by
// (sequence name, position).  Upstream emits genomes and samples in hash-map order; here: index order, input order.
// `note` receives the "Skipping ..." / "Building ..." log lines of call.rs:521,541,551.
struct SampleCalls {
    std::string filename, selected_genome;
    double breadth;
    std::vector<VcfRecord> records;
};
void write_alignments(const std::string& out_dir, const Index& ix, const std::vector<SampleCalls>& samples,
                      void (*note)(const std::string&));

}  // namespace bronko
