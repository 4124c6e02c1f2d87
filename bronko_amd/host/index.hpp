// index.hpp -- BronkoIndex: build from FASTA files and (de)serialise as .bkdb (host side, product code).
//
// Reference behaviour: types /root/reference/src/build.rs:23-60; build_indexes build.rs:145-231;
// save_index build.rs:122-143; decode call.rs:179-200 (bincode 2.0.1 `config::standard()`).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace bronko {

// build.rs:52-60, #[repr(C)]: u16 @0, u8 @2, u32 @4, u8 @8, bool @9 -> 12 bytes (== bk_bucket_info)
struct BucketInfo {
    uint16_t file_id;
    uint8_t  seq_id;
    uint32_t location;
    uint8_t  idx;
    uint8_t  canonical;
};
static_assert(sizeof(BucketInfo) == 12, "BucketInfo must match the reference's #[repr(C)] layout");

struct SeqMeta {              // build.rs:31-36
    std::string name;         // first whitespace-delimited token of the FASTA header
    uint64_t len = 0;
    std::vector<uint8_t> seq; // bytes as in the FASTA, line terminators removed, case kept
};
struct FileMeta {             // build.rs:39-43
    std::string name;         // file stem of the FASTA path
    std::vector<SeqMeta> sequences;
};

// BronkoIndex (build.rs:23-28) with the hash map flattened to CSR: buckets sorted by id, entries of one
// bucket in insertion order (file order, then sequence, then location) -- the order hpv.bkdb decodes to.
struct Index {
    int k = 0;
    std::vector<uint64_t> ids;          // distinct bucket ids, ascending
    std::vector<uint64_t> off;          // ids.size() + 1
    std::vector<BucketInfo> entries;
    std::vector<FileMeta> files;        // ViralMetadata.files
    int meta_k = 0;                     // ViralMetadata.k

    uint64_t total_cells() const;
    uint64_t genome_len(size_t f) const;
};

// One FASTA(.gz) record as needletail yields it to build.rs:171-189
struct FastaRecord { std::string header; std::vector<uint8_t> seq; };
std::vector<FastaRecord> read_fasta(const std::string& path);      // throws std::runtime_error

std::vector<FileMeta> read_genomes(const std::vector<std::string>& genomes);       // the FASTA side of build.rs:155-189 (throws)
Index build_indexes(int k, const std::vector<std::string>& genomes, int threads);  // build.rs:145-231
Index build_indexes_mem(int k, std::vector<FileMeta> files, int threads);
void  save_index(const Index& ix, const std::string& path);                          // build.rs:122-143
Index load_index(const std::string& path);                                            // call.rs:179-200

}  // namespace bronko
