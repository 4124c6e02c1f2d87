// fastx.hpp -- line reader over plain or gzip files for FASTA / FASTQ input (host side): zlib's gzread, or with more than one
// thread to spend on the file pargz.hpp's inflate on several threads (the same bytes and errors).
#pragma once
#include <zlib.h>

#include <cstring>

#include <memory>
#include <stdexcept>
#include <string>

#include "pargz.hpp"

namespace bronko {

class GzLineReader {
public:
    explicit GzLineReader(const std::string& path, unsigned inflate_threads = 1) : path_(path) {
        if (inflate_threads > 1 && ParallelGunzip::is_gzip(path)) { par_.reset(new ParallelGunzip(path, inflate_threads)); return; }
        g_ = gzopen(path.c_str(), "rb");
        if (!g_) throw std::runtime_error("cannot open " + path);
        gzbuffer(g_, 1 << 20);
    }
    ~GzLineReader() { if (g_) gzclose(g_); }
    GzLineReader(const GzLineReader&) = delete;
    GzLineReader& operator=(const GzLineReader&) = delete;

    // Next line without its terminator ("\n" or "\r\n"); false at end of file.
    bool next(std::string& line) {
        line.clear();
        bool any = false;
        for (;;) {
            if (pos_ == len_) {
                const size_t n = fill();
                if (n == 0) break;
                pos_ = 0; len_ = n;
            }
            any = true;
            const char* p = buf_ + pos_;
            const char* nl = (const char*)memchr(p, '\n', len_ - pos_);
            if (nl) {
                line.append(p, nl - p);
                pos_ += (size_t)(nl - p) + 1;
                break;
            }
            line.append(p, len_ - pos_);
            pos_ = len_;
        }
        if (!any) return false;
        while (!line.empty() && line.back() == '\r') line.pop_back();
        return true;
    }

    // The next line appended to `dst` (no terminator) / passed over without a copy; false at end of file.  FASTQ ingest copies one
    // line in four, straight into the batch it is pushed from.
    bool append_next(std::string& dst) { return scan(&dst); }
    bool skip_next() { return scan(nullptr); }

private:
    size_t fill() {
        if (par_) return par_->read(buf_, sizeof buf_);
        const int n = gzread(g_, buf_, sizeof buf_);
        if (n < 0) throw std::runtime_error("read error in " + path_);
        return (size_t)n;
    }
    bool scan(std::string* dst) {
        bool any = false;
        const size_t start = dst ? dst->size() : 0;
        for (;;) {
            if (pos_ == len_) {
                const size_t n = fill();
                if (n == 0) break;
                pos_ = 0; len_ = n;
            }
            any = true;
            const char* p = buf_ + pos_;
            const char* nl = (const char*)memchr(p, '\n', len_ - pos_);
            if (nl) {
                if (dst) dst->append(p, nl - p);
                pos_ += (size_t)(nl - p) + 1;
                break;
            }
            if (dst) dst->append(p, len_ - pos_);
            pos_ = len_;
        }
        if (!any) return false;
        if (dst) while (dst->size() > start && dst->back() == '\r') dst->pop_back();
        return true;
    }

    std::string path_;
    gzFile g_ = nullptr;
    std::unique_ptr<ParallelGunzip> par_;
    char buf_[1 << 16];
    size_t pos_ = 0, len_ = 0;
};

}  // namespace bronko
