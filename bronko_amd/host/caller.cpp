// caller.cpp -- see caller.hpp.
#include "caller.hpp"

#include <algorithm>
#include <array>
#include <cmath>
#include <map>
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <stdexcept>

#include "lcb.hpp"

namespace bronko {

namespace {

// Student-t quantile StudentsT(0,1,n-2).inverse_cdf(1 - 0.001/n) for n = 3..300 (call.rs:922-925).  statrs
// is not vendored with the reference; the window never holds more than 300 values (call.rs:802,813), so the
// quantile is tabulated (oracle/gen_tcrit.py, exact to the last bit of the mathematical value).
const double kTCrit[298] = {
#include "tcrit_table.inc"
};

double thompson_tau(uint64_t n) {
    if (n <= 2) return INFINITY;                                   // call.rs:927-929
    const double t = (n <= 300) ? kTCrit[n - 3] : NAN;
    const double dn = (double)n;
    return (t * (dn - 1.0)) / (std::sqrt(dn) * std::sqrt(dn - 2.0 + t * t));   // call.rs:926
}

std::string first_token(const std::string& s) {
    size_t a = 0;
    while (a < s.size() && isspace((unsigned char)s[a])) a++;
    size_t b = a;
    while (b < s.size() && !isspace((unsigned char)s[b])) b++;
    return s.substr(a, b - a);
}

char base_char(unsigned b) { return b < 4 ? "ACGT"[b] : 'N'; }   // lcb.rs:57-65

// Rust `{:.N}` of an f64: correctly rounded decimal; "NaN" / "inf" / "-inf" for the specials.
std::string fixed(double v, int prec) {
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v > 0 ? "inf" : "-inf";
    char buf[512];
    snprintf(buf, sizeof buf, "%.*f", prec, v);
    return buf;
}

struct File {
    FILE* fp;
    explicit File(const std::string& path) : fp(fopen(path.c_str(), "w")) {}
    ~File() { if (fp) fclose(fp); }
};

}  // namespace

int pick_best_genome(const Index& ix, const std::vector<uint64_t>& stats, const std::vector<uint8_t>& present) {
    int best = -1;
    double best_score = 0.0;
    for (size_t f = 0; f < ix.files.size(); f++) {
        if (!present[f]) continue;                                  // no key in the mapping data
        const double score = (double)stats[f * 3] / (double)ix.genome_len(f) / 2.0;   // call.rs:435
        if (score > best_score) { best_score = score; best = (int)f; }                 // strict >, call.rs:443
    }
    return best;
}

std::vector<double> baseline_noise_max(const uint64_t* fwd4, const uint64_t* rev4, uint64_t len) {
    constexpr int kWindow = 100, kTop = kWindow / 10, kHalf = kWindow / 2;   // call.rs:802-804,824
    std::vector<double> out(len, 0.0);
    std::array<double, kWindow * 3> ring{};      // minor-allele frequencies currently in the window
    std::array<uint8_t, kWindow * 3> flagged{};  // "in_max" flag of each ring slot
    std::array<double, kTop> top{};              // the largest values, descending
    uint64_t n = 0;
    double s = 0.0, s2 = 0.0;

    for (uint64_t i = 0; i < len + kHalf; i++) {
        std::array<double, 4> freq{0.0, 0.0, 0.0, 0.0};
        if (i < len) {                                              // call.rs:831-845
            std::array<uint64_t, 4> cnt;
            for (int b = 0; b < 4; b++) cnt[b] = fwd4[i * 4 + b] + rev4[i * 4 + b];
            std::sort(cnt.begin(), cnt.end(), [](uint64_t a, uint64_t b) { return a > b; });
            const uint64_t depth = cnt[0] + cnt[1] + cnt[2] + cnt[3];
            if (depth != 0) for (int b = 0; b < 4; b++) freq[b] = (double)cnt[b] / (double)depth;
        }
        const size_t slot0 = (size_t)(i % kWindow) * 3;
        for (int r = 1; r < 4; r++) {                               // minor ranks 1..3, call.rs:848
            const size_t slot = slot0 + (size_t)(r - 1);
            const double old = ring[slot];
            if (old > 0.0) {                                        // evict, call.rs:853-870
                n -= 1; s -= old; s2 -= old * old;
                if (flagged[slot]) {
                    for (int q = 0; q < kTop; q++) {
                        if (std::fabs(top[q] - old) < 1e-12) {
                            for (int z = q; z + 1 < kTop; z++) top[z] = top[z + 1];
                            top[kTop - 1] = 0.0;
                            break;
                        }
                    }
                    flagged[slot] = 0;
                }
            }
            const double maf = freq[r];
            if (maf > 0.0) {                                        // insert, call.rs:873-890
                n += 1; s += maf; s2 += maf * maf;
                for (int q = kTop - 1; q >= 0; q--) {
                    if (!(maf > top[q])) break;
                    if (q + 1 < kTop) top[q + 1] = top[q];
                    top[q] = maf;
                }
                flagged[slot] = 1;                                  // set whether or not it entered the table
            } else {
                flagged[slot] = 0;
            }
            ring[slot] = maf;
        }

        double mu = 0.0, var = 0.0;
        if (n != 0) { mu = s / (double)n; var = s2 / (double)n - mu * mu; }   // population variance, call.rs:901-907
        int idx = 0;
        uint64_t cn = n;
        double cs = s, cs2 = s2;
        while (idx < kTop && top[idx] != 0.0) {                     // strip outliers, call.rs:917-950
            const double cand = top[idx];
            if (!(std::fabs(cand - mu) > thompson_tau(cn) * std::sqrt(var))) break;
            cs -= cand;
            cs2 -= cand;                                            // sic (call.rs:936): the value, not its square
            cn -= 1;
            if (cn > 0) { mu = cs / (double)cn; var = cs2 / (double)cn - mu * mu; }
            else { mu = 0.0; var = 0.0; }
            idx++;
        }
        if (i >= (uint64_t)kHalf && i - kHalf < len)                // call.rs:953-962
            out[i - kHalf] = idx < kTop ? top[idx] : 0.0;           // idx == kTop would index out of bounds upstream
    }
    return out;
}

CallSummary call_variants(const Index& ix, int file_id, const Pileup& p, const CallParams& prm) {
    CallSummary out;
    uint64_t covered = 0, positions = 0, coverage = 0;
    uint64_t cell = 0;
    for (int f = 0; f < file_id; f++) cell += ix.genome_len(f);
    const FileMeta& fm = ix.files[file_id];
    for (size_t sid = 0; sid < fm.sequences.size(); sid++) {        // metadata order (upstream: DashMap order)
        const SeqMeta& sm = fm.sequences[sid];
        const uint64_t len = sm.len;
        const uint64_t* fd = p.fwd_depth.data() + cell * 4;
        const uint64_t* rd = p.rev_depth.data() + cell * 4;
        const uint64_t* fk = p.fwd_nk.data() + cell * 4;
        const uint64_t* rk = p.rev_nk.data() + cell * 4;
        const std::vector<double> noise = baseline_noise_max(fd, rd, len);          // call.rs:1002
        int64_t start = 0, end = (int64_t)len;
        if (!prm.no_end_filter) { start = prm.k; end = (int64_t)len - prm.k; }      // call.rs:1013-1016
        positions += len;
        for (int64_t i = start; i < end; i++) {
            const uint64_t* row = fd + i * 4;
            const uint64_t* rrow = rd + i * 4;
            const unsigned ref = nt_to_bits(sm.seq[i]);                             // non-ACGT counts as A
            uint64_t tot[4], depth = 0;
            for (int b = 0; b < 4; b++) { tot[b] = row[b] + rrow[b]; depth += tot[b]; }
            if (depth == 0) continue;
            covered += 1;
            coverage += depth;
            for (unsigned alt = 0; alt < 4; alt++) {
                if (alt == ref || tot[alt] == 0) continue;
                double sor = prm.strand_odds_max + 1.0;
                if (!prm.no_strand_filter) {                                        // call.rs:1059-1096
                    const double a = (double)row[ref] + 1.0, b = (double)rrow[ref] + 1.0;
                    const double c = (double)row[alt] + 1.0, d = (double)rrow[alt] + 1.0;
                    const double min_strand = std::fmin(a + c, b + d) / (a + b + c + d);
                    if (!prm.no_strand_balance_filter || min_strand >= prm.strand_balance_ratio) {
                        const double r = (a * d) / (b * c);
                        sor = std::log(r + 1.0 / r) + std::log(std::fmin(a, b) / std::fmax(a, b))
                              - std::log(std::fmin(c, d) / std::fmax(c, d));
                        if (sor > prm.strand_odds_max) continue;
                        if (fk[i * 4 + alt] < prm.n_per_strand && rk[i * 4 + alt] < prm.n_per_strand) continue;
                    } else {
                        sor = -1.0;
                    }
                }
                const double af = (double)tot[alt] / (double)depth;
                const double y0 = prm.variant_multiplier;
                const double factor = y0 + 0.5 * std::pow(0.03, 100.0 * af);        // call.rs:1102-1105
                if (af < prm.min_af || af < std::fmax(factor, y0) * noise[i]) continue;
                if (af >= 0.5) {
                    out.n_major += 1;
                } else {
                    if (depth < prm.min_depth) continue;
                    if (tot[alt] < prm.min_variant_depth) continue;
                    out.n_minor += 1;
                }
                out.records.push_back(VcfRecord{(int)sid, (uint64_t)i + 1, (uint8_t)ref, (uint8_t)alt, row[ref], rrow[ref],
                                                row[alt], rrow[alt], depth, af, sor});
            }
        }
        cell += len;
    }
    out.breadth = (double)covered / (double)positions;              // call.rs:1144
    out.depth = (double)coverage / (double)covered;                 // call.rs:1145 (NaN when nothing is covered)
    return out;
}

std::string clean_sample_id(const std::string& path) {
    static const char* const kSuffixes[] = {".fastq.gz", ".fasta.gz", "fna.gz", "fnq.gz", ".fq.gz", ".fastq",
                                            ".fasta", ".fnq", ".fna", ".fa", ".fq"};
    const size_t slash = path.find_last_of('/');
    std::string name = slash == std::string::npos ? path : path.substr(slash + 1);
    auto ends_with = [](const std::string& s, const std::string& suf) {
        return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
    };
    for (const char* suf : kSuffixes) {
        if (ends_with(name, suf)) {
            const std::string sfx(suf);
            while (ends_with(name, sfx)) name.resize(name.size() - sfx.size());    // trim_end_matches
            return name;
        }
    }
    const size_t dot = name.find_last_of('.');
    if (dot != std::string::npos && dot != 0) name.resize(dot);
    return name;
}

void write_vcf(const std::string& out_path, const std::string& reads_path, const Index& ix, int file_id, const std::vector<VcfRecord>& recs) {
    File f(out_path);
    if (!f.fp) throw std::runtime_error("Failed to create vcf output file");
    const FileMeta& fm = ix.files[file_id];
    fprintf(f.fp, "##fileformat=VCFv4.5\n##source=bronko-v0.1.0\n##reference=file://%s\n", reads_path.c_str());
    for (const auto& s : fm.sequences)
        fprintf(f.fp, "##contig=<ID=%s,length=%llu>\n", first_token(s.name).c_str(), (unsigned long long)s.len);
    fputs("##INFO=<ID=DP,Number=1,Type=Integer,Description=\"Total Depth\">\n"
          "##INFO=<ID=AF,Number=1,Type=Float,Description=\"Allele Frequency\">\n"
          "##INFO=<ID=DP4,Number=4,Type=Integer,Description=\"Fwd_ref,Rev_ref,Fwd_alt,Rev_alt\">\n"
          "##INFO=<ID=SOR,Number=4,Type=Float,Description=\"SOR\">\n"
          "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n", f.fp);
    for (const VcfRecord& v : recs)
        fprintf(f.fp, "%s\t%llu\t.\t%c\t%c\t.\tPASS\tDP=%llu;AF=%s;DP4=%llu,%llu,%llu,%llu;SOR=%s\n",
                first_token(fm.sequences[v.seq_id].name).c_str(), (unsigned long long)v.pos, base_char(v.ref_base),
                base_char(v.alt_base), (unsigned long long)v.depth, fixed(v.af, 3).c_str(), (unsigned long long)v.fwd_ref,
                (unsigned long long)v.rev_ref, (unsigned long long)v.fwd_alt, (unsigned long long)v.rev_alt, fixed(v.sor, 3).c_str());
}

void write_pileup_tsv(const std::string& out_path, const Index& ix, int file_id, const Pileup& p) {
    File f(out_path);
    if (!f.fp) throw std::runtime_error("Failed to create tsv pileup file");
    fputs("reference\tindex\tref\tA\tC\tG\tT\ta\tc\tg\tt\n", f.fp);
    uint64_t cell = 0;
    for (int g = 0; g < file_id; g++) cell += ix.genome_len(g);
    for (const auto& s : ix.files[file_id].sequences) {
        for (uint64_t i = 0; i < s.len; i++, cell++) {
            const uint64_t* a = p.fwd_depth.data() + cell * 4;
            const uint64_t* b = p.rev_depth.data() + cell * 4;
            fprintf(f.fp, "%s\t%llu\t%c\t%llu\t%llu\t%llu\t%llu\t%llu\t%llu\t%llu\t%llu\n", s.name.c_str(),
                    (unsigned long long)(i + 1), (char)s.seq[i], (unsigned long long)a[0], (unsigned long long)a[1],
                    (unsigned long long)a[2], (unsigned long long)a[3], (unsigned long long)b[0], (unsigned long long)b[1],
                    (unsigned long long)b[2], (unsigned long long)b[3]);
        }
    }
}

void write_overview_tsv(const std::string& out_path, const std::vector<OverviewRow>& rows) {
    File f(out_path);
    if (!f.fp) throw std::runtime_error("Failed to create tsv file");
    fputs("filename\tselected_genome\tnum_major_variants\tnum_minor_variants\tbreadth_coverage\tdepth_coverage\t"
          "num_perfect_kmers\tnum_variant_kmers\tnum_unmapped_kmers\n", f.fp);
    for (const auto& r : rows)
        fprintf(f.fp, "%s\t%s\t%llu\t%llu\t%s\t%s\t%llu\t%llu\t%llu\n", r.filename.c_str(), r.selected_genome.c_str(),
                (unsigned long long)r.n_major, (unsigned long long)r.n_minor, fixed(r.breadth, 4).c_str(), fixed(r.depth, 4).c_str(),
                (unsigned long long)r.n_perfect, (unsigned long long)r.n_variant, (unsigned long long)r.n_unmapped);
}

void write_alignments(const std::string& out_dir, const Index& ix, const std::vector<SampleCalls>& samples,
                      void (*note)(const std::string&)) {
    for (size_t g = 0; g < ix.files.size(); g++) {
        const FileMeta& fm = ix.files[g];
        std::vector<const SampleCalls*> group;
        for (const SampleCalls& sc : samples) {
            if (sc.selected_genome != fm.name) continue;
            if (sc.breadth < 0.90) {                                                    // call.rs:520-523
                if (note) note("Skipping " + sc.filename + " (breadth of coverage = " + std::to_string(sc.breadth) + ")");
                continue;
            }
            group.push_back(&sc);
        }
        if (group.empty()) continue;
        if (group.size() < 3) {                                                         // call.rs:540-543
            if (note) note("Skipping " + fm.name + " (only " + std::to_string(group.size()) + " samples)");
            continue;
        }
        if (note) note("Building alignment for genome " + fm.name + " with " + std::to_string(group.size()) + " samples");
        // every (sequence name, position) with a major variant in some sample -> its reference base (call.rs:570-582)
        std::map<std::pair<std::string, uint64_t>, uint8_t> columns;
        std::vector<std::map<std::pair<std::string, uint64_t>, uint8_t>> own(group.size());
        for (size_t i = 0; i < group.size(); i++)
            for (const VcfRecord& r : group[i]->records) {
                if (!(r.af >= 0.5)) continue;
                const std::pair<std::string, uint64_t> key(fm.sequences[(size_t)r.seq_id].name, r.pos);
                columns[key] = r.ref_base;
                own[i][key] = r.alt_base;
            }
        const std::string path = out_dir + "/" + fm.name + ".mfa";
        FILE* f = fopen(path.c_str(), "wb");
        if (!f) throw std::runtime_error(std::string(strerror(errno)) + " | Failed to create mfa alignment file");
        std::string row;
        for (const auto& kv : columns) row.push_back(base_char(kv.second));            // std::map iterates in (name, position) order
        fprintf(f, ">%s\n%s\n", fm.name.c_str(), row.c_str());
        for (size_t i = 0; i < group.size(); i++) {
            row.clear();
            for (const auto& kv : columns) {
                const auto it = own[i].find(kv.first);
                row.push_back(base_char(it != own[i].end() ? it->second : kv.second));
            }
            fprintf(f, ">%s\n%s\n", clean_sample_id(group[i]->filename).c_str(), row.c_str());
        }
        fclose(f);
    }
}

}  // namespace bronko
