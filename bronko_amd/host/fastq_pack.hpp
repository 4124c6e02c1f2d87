// fastq_pack.hpp -- one FASTQ(.gz) file -> batches of 2-bit packed records (include/bronko_hip.h: the layout bk_push_reads_packed
// takes), parsed and packed on several threads: what `bronko call` feeds an engine from (round 6).
//
// The reference hands its FASTQ files to KMC, which reads them with `-t` threads (call.rs:1166-1181).  Until round 6 a file's text --
// inflated on several threads by pargz.hpp -- went through ONE line loop per file (14 M reads/s) into batches of sequence lines that
// the engine packed on the device (bk_push_reads_ascii: 110 M reads/s over PCIe).  Here the text's pieces (a few MB each, as
// pargz.hpp's waves deliver them, or slices of a mapped plain file) are taken apart where they are:
//   stage A (a thread per piece)   the positions of the piece's line ends
//   in order (one thread)          the number of the piece's first line -- a FASTQ record is four lines, the sequence is line 1
//                                  mod 4 (the rule of the line loop it replaces, fastx.hpp / needletail's FASTQ reader) -- and the
//                                  piece of a line that the pieces before it left over
//   stage B (a thread per piece)   its sequence lines 2-bit packed (bk_pack_reads: KMC's splitting at non-ACGT symbols, runs
//                                  shorter than k dropped) into one PackedBatch
// and delivered in file order.  Inputs that cannot be mapped (a FIFO, /dev/fd/N) or a single thread keep the line loop; the
// batches are the same records either way (tests/test_fastq_pack.py).
#pragma once
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bronko_hip.h"
#include "fastx.hpp"
#include "pargz.hpp"

namespace bronko {

struct PackedBatch {
    std::vector<uint32_t> words;   // [n_records][stride]
    std::vector<uint16_t> lens;    // [n_records]
    uint32_t stride = 0;
    uint64_t n_records = 0, n_reads = 0;   // records packed; sequence lines seen (a read with an N is several records, a short one none)
    size_t bytes() const { return words.size() * 4 + lens.size() * 2; }
    void clear() { words.clear(); lens.clear(); stride = 0; n_records = 0; n_reads = 0; }
};

// 2-bit code of a sequence symbol (A C G T, either case), 4 = anything else (KMC splits a read there)
struct AcgtLut {
    uint8_t c[256];
    AcgtLut() { memset(c, 4, sizeof c); c['A'] = c['a'] = 0; c['C'] = c['c'] = 1; c['G'] = c['g'] = 2; c['T'] = c['t'] = 3; }
};
inline const AcgtLut& acgt_lut() { static const AcgtLut l; return l; }

// The sequence lines `ptr[i]` (length `len[i]`) packed into `out` (overwritten): the records bk_pack_reads makes of them, in the same
// order (include/bronko_hip.h: one record per maximal ACGT run of at least k symbols, runs longer than a record cut into chunks
// that overlap by k - 1).  A line that is one run and fits a record -- nearly every line -- takes the short way: sixteen table
// look-ups a word, no branch per symbol (the byte-at-a-time packer of the C ABI made 1 us a read of it: with the text inflated
// on 32 threads the packing was the slower half).
inline void pack_lines(const std::vector<const uint8_t*>& ptr, const std::vector<uint64_t>& len, int k, PackedBatch& out) {
    out.clear();
    out.n_reads = ptr.size();
    uint64_t longest = (uint64_t)k;
    for (uint64_t l : len) longest = std::max(longest, l);
    const uint32_t stride = (uint32_t)std::min<uint64_t>((longest + 15) / 16, 4095);   // (the engine's own rule for a batch of sequence lines: bk_push_reads_ascii)
    out.stride = stride;
    const uint64_t maxb = std::min<uint64_t>((uint64_t)stride * 16, 65535);
    const uint8_t* const lut = acgt_lut().c;
    out.words.resize((ptr.size() + 16) * (size_t)stride);
    out.lens.resize(ptr.size() + 16);
    uint64_t n = 0;
    auto room = [&]() {
        if (n == out.lens.size()) { out.lens.resize(n + n / 2 + 64); out.words.resize(out.lens.size() * (size_t)stride); }
    };
    auto emit = [&](const uint8_t* s, uint64_t l) {   // one record of l <= maxb ACGT symbols
        room();
        uint32_t* w = out.words.data() + n * stride;
        uint64_t i = 0;
        for (uint32_t wi = 0; wi < stride; wi++) {
            uint32_t x = 0;
            const uint64_t e = std::min<uint64_t>(l, i + 16);
            for (uint32_t sh = 0; i < e; i++, sh += 2) x |= (uint32_t)lut[s[i]] << sh;
            w[wi] = x;
        }
        out.lens[n++] = (uint16_t)l;
    };
    for (size_t r = 0; r < ptr.size(); r++) {
        const uint8_t* s = ptr[r];
        const uint64_t l = len[r];
        if (l >= (uint64_t)k && l <= maxb) {
            // the short way: packed as if it were one run, sixteen symbols a word; "anything else" shows in bit 2 of a code
            room();
            uint32_t* w = out.words.data() + n * stride;
            uint32_t bad = 0;
            uint64_t i = 0;
            uint32_t wi = 0;
            for (; i + 16 <= l; i += 16, wi++) {
                uint32_t x = 0, y = 0;
                for (uint32_t j = 0; j < 16; j++) { const uint32_t c = lut[s[i + j]]; y |= c; x |= (c & 3u) << (2 * j); }
                w[wi] = x; bad |= y;
            }
            if (i < l) {
                uint32_t x = 0, y = 0;
                for (uint32_t j = 0; i + j < l; j++) { const uint32_t c = lut[s[i + j]]; y |= c; x |= (c & 3u) << (2 * j); }
                w[wi++] = x; bad |= y;
            }
            if (!(bad & 4u)) {
                for (; wi < stride; wi++) w[wi] = 0u;
                out.lens[n++] = (uint16_t)l;
                continue;
            }
        }
        // the long way: run by run (what the C ABI's packer does)
        uint64_t start = 0;
        for (uint64_t i = 0; i <= l; i++) {
            if (i == l || lut[s[i]] > 3) {
                const uint64_t rl = i - start;
                if (rl >= (uint64_t)k) {
                    uint64_t pos = 0;
                    for (;;) {
                        const uint64_t take = std::min(maxb, rl - pos);
                        emit(s + start + pos, take);
                        if (pos + take >= rl) break;
                        pos += take - (uint64_t)(k - 1);      // next chunk re-reads k-1 bases: no k-mer lost or doubled
                    }
                }
                start = i + 1;
            }
        }
    }
    out.n_records = n;
    out.words.resize((size_t)n * stride);
    out.lens.resize((size_t)n);
}

class FastqPacker {
public:
    // threads: parse / pack threads (and, for gzip input, as many inflate threads again: they take turns)
    FastqPacker(const std::string& path, int k, unsigned threads) : path_(path), k_(k), threads_(std::max(1u, threads)) {
        struct stat st;
        const bool regular = ::stat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0;
        if (threads_ > 1 && regular) {
            if (ParallelGunzip::is_gzip(path)) gz_.reset(new ParallelGunzip(path, std::min(threads_, 32u)));   // (a file's inflate gains nothing past 32 threads)
            else if (!looks_gzip(path)) map_plain(st.st_size);
        }
        if (gz_ || plain_) {
            // (taking the text apart and packing it is a tenth of inflating it -- 0.1 us against 1 us a read --: a few workers keep
            // up with any number of inflate threads, and sixty-four of them only queue up at the one lock)
            const unsigned n_workers = gz_ ? std::min(8u, std::max(2u, threads_ / 4)) : std::min(16u, threads_);
            for (unsigned t = 0; t < n_workers; t++) workers_.emplace_back([this] { work(); });
            feeder_ = std::thread([this] { feed(); });
            sequencer_ = std::thread([this] { sequence(); });
        } else {
            lines_.reset(new GzLineReader(path, 1));
        }
    }
    ~FastqPacker() {
        { std::unique_lock<std::mutex> lk(m_); quit_ = true; }
        cv_.notify_all();
        if (feeder_.joinable()) feeder_.join();
        if (sequencer_.joinable()) sequencer_.join();
        for (auto& w : workers_) if (w.joinable()) w.join();
        gz_.reset();
        if (plain_) munmap(const_cast<uint8_t*>(plain_), plain_n_);
        if (fd_ >= 0) ::close(fd_);
    }
    FastqPacker(const FastqPacker&) = delete;
    FastqPacker& operator=(const FastqPacker&) = delete;

    // The file's next batch; false at its end.  Throws std::runtime_error for a damaged or unreadable file.
    bool next(PackedBatch& out) {
        if (lines_) return next_serial(out);
        std::shared_ptr<Job> j;
        {
            std::unique_lock<std::mutex> lk(m_);
            cv_.wait(lk, [&] { return !q2_.empty() || seq_done_; });
            if (q2_.empty()) {
                if (!error_.empty()) throw std::runtime_error(error_);
                return false;
            }
            j = q2_.front(); q2_.pop_front();
        }
        cv_.notify_all();
        j->done_b.get_future().wait();
        if (!j->error.empty()) throw std::runtime_error(j->error);
        std::swap(out, j->out);   // (the caller's old buffers stay with the reader)
        { std::unique_lock<std::mutex> lk(m_); spare_.push_back(std::move(j->out)); if (spare_.size() > 64) spare_.pop_front(); }
        return true;
    }

private:
    struct Job {
        pargz::Piece piece;                 // gzip input: the piece's own buffer
        const uint8_t* d = nullptr;         // the piece's text (the buffer above, or a slice of the mapped file)
        size_t n = 0;
        std::vector<uint32_t> nl;           // positions of its '\n'
        uint64_t line_base = 0;             // number of the line that its first '\n' ends
        std::string head;                   // what the pieces before it hold of that line
        bool last = false;                  // the text's end: what is left behind the last '\n' is a line too
        PackedBatch out;
        std::string error;
        std::promise<void> done_a, done_b;
    };
    static constexpr size_t kSlice = 4u << 20;   // a mapped plain file is taken in slices of this size
    static constexpr size_t kAheadJobs = 2;      // jobs in flight per thread (memory: a piece of a few MB each)

    static bool looks_gzip(const std::string& path) {   // (a file of fewer than three bytes, or unreadable: the line loop says what is wrong)
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return true;
        unsigned char h[2] = {0, 0};
        const ssize_t r = ::pread(fd, h, 2, 0);
        ::close(fd);
        return r < 2 || (h[0] == 0x1f && h[1] == 0x8b);
    }
    void map_plain(off_t size) {
        fd_ = ::open(path_.c_str(), O_RDONLY);
        if (fd_ < 0) return;
        void* m = mmap(nullptr, (size_t)size, PROT_READ, MAP_PRIVATE, fd_, 0);
        if (m == MAP_FAILED) { ::close(fd_); fd_ = -1; return; }
        plain_ = static_cast<const uint8_t*>(m);
        plain_n_ = (size_t)size;
        madvise(m, plain_n_, MADV_SEQUENTIAL);
    }

    // ---- the worker pool -------------------------------------------------------------------------------------------------
    void submit(std::function<void()> f) {
        { std::unique_lock<std::mutex> lk(m_); tasks_.push_back(std::move(f)); }
        cv_.notify_all();
    }
    void work() {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return quit_ || !tasks_.empty(); });
                if (tasks_.empty()) return;   // (quit, nothing left)
                f = std::move(tasks_.front()); tasks_.pop_front();
            }
            f();
        }
    }

    // ---- stage A: the line ends of a piece ----------------------------------------------------------------------------------
    void stage_a(Job& j) {
        try {
            { std::unique_lock<std::mutex> lk(m_); if (!spare_nl_.empty()) { j.nl = std::move(spare_nl_.back()); spare_nl_.pop_back(); } }
            j.nl.clear();
            j.nl.reserve(j.n / 64 + 16);
            const uint8_t* p = j.d;
            const uint8_t* const end = j.d + j.n;
            while (p < end) {
                const uint8_t* q = static_cast<const uint8_t*>(memchr(p, '\n', (size_t)(end - p)));
                if (!q) break;
                j.nl.push_back((uint32_t)(q - j.d));
                p = q + 1;
            }
        } catch (const std::exception& e) { j.error = e.what(); }
        j.done_a.set_value();
    }
    // ---- stage B: the piece's sequence lines, packed ----------------------------------------------------------------------
    void stage_b(Job& j) {
        try {
            if (j.error.empty()) {
                static thread_local std::vector<const uint8_t*> ptr;   // (a worker's own: their memory is used again, piece after piece)
                static thread_local std::vector<uint64_t> len;
                ptr.clear(); len.clear();
                ptr.reserve(j.nl.size() / 4 + 2); len.reserve(j.nl.size() / 4 + 2);
                auto add = [&](const uint8_t* s, size_t n) {
                    while (n && s[n - 1] == '\r') n--;   // ("\r\n": the line loop strips it)
                    ptr.push_back(s); len.push_back(n);
                };
                std::string first;   // the line the pieces before this one began, whole
                size_t from = 0;
                for (size_t i = 0; i < j.nl.size(); i++) {
                    const size_t to = j.nl[i];
                    if (((j.line_base + i) & 3u) == 1u) {
                        if (i == 0 && !j.head.empty()) { first = j.head; first.append(reinterpret_cast<const char*>(j.d), to); add(reinterpret_cast<const uint8_t*>(first.data()), first.size()); }
                        else add(j.d + from, to - from);
                    }
                    from = to + 1;
                }
                std::string tail;   // the text's last line has no '\n'
                if (j.last && ((j.line_base + j.nl.size()) & 3u) == 1u) {
                    tail = j.nl.empty() ? j.head : std::string();
                    tail.append(reinterpret_cast<const char*>(j.d + from), j.n - from);
                    if (!tail.empty()) add(reinterpret_cast<const uint8_t*>(tail.data()), tail.size());
                }
                { std::unique_lock<std::mutex> lk(m_); if (!spare_.empty()) { j.out = std::move(spare_.back()); spare_.pop_back(); } }
                pack_lines(ptr, len, k_, j.out);
            }
        } catch (const std::exception& e) { j.error = e.what(); }
        j.piece = pargz::Piece();   // (its buffer goes back to the pool)
        { std::unique_lock<std::mutex> lk(m_); if (spare_nl_.size() < 256) spare_nl_.push_back(std::move(j.nl)); }
        j.done_b.set_value();
    }

    // ---- pieces in, jobs out ------------------------------------------------------------------------------------------------
    void push_job(std::shared_ptr<Job> j) {
        Job* jp = j.get();
        {
            std::unique_lock<std::mutex> lk(m_);
            cv_.wait(lk, [&] { return quit_ || q1_.size() + q2_.size() < (size_t)threads_ * kAheadJobs; });
            if (quit_) return;
            q1_.push_back(std::move(j));
            tasks_.push_back([this, jp] { stage_a(*jp); });
        }
        cv_.notify_all();
    }
    void feed() {
        try {
            if (gz_) {
                pargz::Piece p;
                while (gz_->take(p)) {
                    auto j = std::make_shared<Job>();
                    j->piece = std::move(p); j->d = reinterpret_cast<const uint8_t*>(j->piece.d); j->n = j->piece.n;
                    push_job(std::move(j));
                    { std::unique_lock<std::mutex> lk(m_); if (quit_) return; }
                }
            } else {
                for (size_t at = 0; at < plain_n_; at += kSlice) {
                    auto j = std::make_shared<Job>();
                    j->d = plain_ + at; j->n = std::min(kSlice, plain_n_ - at);
                    push_job(std::move(j));
                    { std::unique_lock<std::mutex> lk(m_); if (quit_) return; }
                }
            }
        } catch (const std::exception& e) {
            std::unique_lock<std::mutex> lk(m_);
            if (error_.empty()) error_ = e.what();
        }
        { std::unique_lock<std::mutex> lk(m_); feed_done_ = true; }
        cv_.notify_all();
    }
    // in order: every piece learns its first line's number and the beginning of that line; then it is packed
    void sequence() {
        uint64_t lines = 0;
        std::string head;
        std::shared_ptr<Job> held;   // the newest piece: it is sent on once it is known whether it is the text's last
        auto send = [&](std::shared_ptr<Job> j, bool last) {
            j->line_base = lines; j->head = head; j->last = last;
            lines += j->nl.size();
            if (j->nl.empty()) head.append(reinterpret_cast<const char*>(j->d), j->n);
            else head.assign(reinterpret_cast<const char*>(j->d + j->nl.back() + 1), j->n - j->nl.back() - 1);
            Job* jp = j.get();
            {
                std::unique_lock<std::mutex> lk(m_);
                q2_.push_back(std::move(j));
                tasks_.push_back([this, jp] { stage_b(*jp); });
            }
            cv_.notify_all();
        };
        for (;;) {
            std::shared_ptr<Job> j;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return quit_ || !q1_.empty() || feed_done_; });
                if (quit_) return;
                if (q1_.empty()) break;
                j = q1_.front(); q1_.pop_front();
            }
            cv_.notify_all();
            j->done_a.get_future().wait();
            if (held) send(std::move(held), false);
            held = std::move(j);
        }
        if (held) send(std::move(held), true);
        { std::unique_lock<std::mutex> lk(m_); seq_done_ = true; }
        cv_.notify_all();
    }

    // ---- the line loop (streams, one thread): the same batches, 64 Ki reads at a time ------------------------------------------
    bool next_serial(PackedBatch& out) {
        if (serial_end_) return false;
        constexpr uint64_t kBatchReads = 1u << 16;
        buf_.clear(); off_.assign(1, 0);
        for (; off_.size() <= kBatchReads;) {
            if ((serial_line_ & 3u) != 1u) { if (!lines_->skip_next()) { serial_end_ = true; break; } serial_line_++; continue; }
            if (!lines_->append_next(buf_)) { serial_end_ = true; break; }
            serial_line_++;
            off_.push_back(buf_.size());
        }
        if (off_.size() == 1 && serial_end_) return false;
        std::vector<const uint8_t*> ptr(off_.size() - 1);
        std::vector<uint64_t> len(off_.size() - 1);
        for (size_t i = 0; i + 1 < off_.size(); i++) { ptr[i] = reinterpret_cast<const uint8_t*>(buf_.data()) + off_[i]; len[i] = off_[i + 1] - off_[i]; }
        pack_lines(ptr, len, k_, out);
        return true;
    }

    std::string path_;
    int k_;
    unsigned threads_;
    std::unique_ptr<ParallelGunzip> gz_;
    std::unique_ptr<GzLineReader> lines_;
    int fd_ = -1;
    const uint8_t* plain_ = nullptr;
    size_t plain_n_ = 0;
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<std::function<void()>> tasks_;
    std::deque<std::shared_ptr<Job>> q1_, q2_;   // waiting for their line ends / for their batch, in file order
    std::deque<PackedBatch> spare_;              // batches the caller handed back: their buffers are used again
    std::vector<std::vector<uint32_t>> spare_nl_;   // ... and the jobs' lists of line ends
    std::vector<std::thread> workers_;
    std::thread feeder_, sequencer_;
    bool quit_ = false, feed_done_ = false, seq_done_ = false;
    std::string error_;
    std::string buf_;
    std::vector<uint64_t> off_;
    uint64_t serial_line_ = 0;
    bool serial_end_ = false;
};

}  // namespace bronko
