#include "bgzf.h"

#include <atomic>

#include "inflate.h"

#include <zlib.h>

#include <cstdlib>
#include <cstring>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace bvchost {

// blocks the fast decoder declined and zlib inflated (expected: none; tests/test_host.py reads it)
std::atomic<long> g_zlib_fallbacks(0);
long bgzf_zlib_fallbacks() { return g_zlib_fallbacks.load(); }
// blocks whose inflated bytes did not hash to the CRC32 of their trailer (expected: none)
std::atomic<long> g_crc_errors(0);
long bgzf_crc_errors() { return g_crc_errors.load(); }

// ---- CRC32 of a block (gzip trailer, RFC 1952) -----------------------------------------------------------------------
// htslib verifies the CRC32 of every BGZF block it inflates, so the reference's bt_s does (bgzf_getline, src/BaseVarC.cpp:406);
// so does this reader.  zlib's table-driven crc32 costs about as much as the block inflate of inflate.cpp itself, so on x86-64 with
// PCLMULQDQ the bulk of a block goes through carry-less multiplication instead: four 128-bit lanes folded by x^512 mod P per 64
// bytes, then by x^128, then a Barrett reduction (Gopal, Ozturk, Guilford, Wolrich, Feghali, Dixon, Karakoyunlu: "Fast CRC
// computation for generic polynomials using PCLMULQDQ instruction", Intel 2009; constants for the reflected polynomial 0xEDB88320).
// The bytes in front of and behind the multiple-of-16 middle go through zlib.  Checked against zlib in tests/test_host.py.
#if defined(__x86_64__)
__attribute__((target("pclmul,sse4.1")))
static inline __m128i crc32_fold128(__m128i x, __m128i k3k4, __m128i next)      // x * x^128 mod P, plus the next 16 bytes
{
    const __m128i a = _mm_clmulepi64_si128(x, k3k4, 0x00);
    return _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x, k3k4, 0x11), a), next);
}

__attribute__((target("pclmul,sse4.1")))
static uint32_t crc32_fold(const unsigned char *buf, size_t len, uint32_t state)      // len >= 64 and a multiple of 16; raw state
{
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596, 0x0154442bd4);
    const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009e, 0x01751997d0);
    const __m128i k5 = _mm_set_epi64x(0, 0x0163cd6124);
    const __m128i poly = _mm_set_epi64x(0x01f7011641, 0x01db710641);
    const __m128i *p = reinterpret_cast<const __m128i *>(buf);
    __m128i x1 = _mm_loadu_si128(p), x2 = _mm_loadu_si128(p + 1), x3 = _mm_loadu_si128(p + 2), x4 = _mm_loadu_si128(p + 3);
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)state));
    p += 4; len -= 64;
    while (len >= 64) {
        const __m128i a1 = _mm_clmulepi64_si128(x1, k1k2, 0x00), a2 = _mm_clmulepi64_si128(x2, k1k2, 0x00);
        const __m128i a3 = _mm_clmulepi64_si128(x3, k1k2, 0x00), a4 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
        x1 = _mm_clmulepi64_si128(x1, k1k2, 0x11); x2 = _mm_clmulepi64_si128(x2, k1k2, 0x11);
        x3 = _mm_clmulepi64_si128(x3, k1k2, 0x11); x4 = _mm_clmulepi64_si128(x4, k1k2, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, a1), _mm_loadu_si128(p));
        x2 = _mm_xor_si128(_mm_xor_si128(x2, a2), _mm_loadu_si128(p + 1));
        x3 = _mm_xor_si128(_mm_xor_si128(x3, a3), _mm_loadu_si128(p + 2));
        x4 = _mm_xor_si128(_mm_xor_si128(x4, a4), _mm_loadu_si128(p + 3));
        p += 4; len -= 64;
    }
    x1 = crc32_fold128(x1, k3k4, x2); x1 = crc32_fold128(x1, k3k4, x3); x1 = crc32_fold128(x1, k3k4, x4);
    while (len >= 16) { x1 = crc32_fold128(x1, k3k4, _mm_loadu_si128(p)); ++p; len -= 16; }
    // 128 -> 64 bits
    const __m128i mask32 = _mm_setr_epi32(~0, 0, ~0, 0);
    __m128i t = _mm_clmulepi64_si128(x1, k3k4, 0x10);
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), t);
    t = _mm_srli_si128(x1, 4);
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, mask32), k5, 0x00), t);
    // Barrett reduction to 32 bits
    t = _mm_clmulepi64_si128(_mm_and_si128(x1, mask32), poly, 0x10);
    t = _mm_clmulepi64_si128(_mm_and_si128(t, mask32), poly, 0x00);
    x1 = _mm_xor_si128(x1, t);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
#endif

uint32_t bgzf_crc32(const unsigned char *buf, size_t len)
{
    uint32_t crc = (uint32_t)crc32(0L, Z_NULL, 0);
#if defined(__x86_64__)
    static const bool have_clmul = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    if (have_clmul && len >= 64) {
        const size_t mid = len & ~(size_t)15;
        crc = ~crc32_fold(buf, mid, ~crc);
        buf += mid; len -= mid;
    }
#endif
    return len ? (uint32_t)crc32(crc, buf, (uInt)len) : crc;
}


static const size_t kBlockIn = 0xff00;          // uncompressed bytes per block
static const unsigned char kEofMarker[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0,
                                             0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};

BgzfWriter::BgzfWriter(const std::string &path, int level, bool background, int workers)
    : fp_(std::fopen(path.c_str(), "wb")), level_(level), failed_(false), background_(background && fp_ != nullptr)
{
    buf_.reserve(kBlockIn);
    if (background_)
        for (int w = 0; w < (workers > 1 ? workers : 1); ++w)
            workers_.emplace_back([this] {
                std::vector<unsigned char> blk, out;
                for (;;) {
                    uint64_t seq;
                    {
                        std::unique_lock<std::mutex> g(mu_);
                        cv_.wait(g, [&] { return !pending_.empty() || closing_; });
                        if (pending_.empty()) return;
                        blk.swap(pending_.front());
                        pending_.pop_front();
                        seq = taken_++;
                    }
                    cv_.notify_all();               // room for the producer
                    out.clear();
                    deflate_block(blk, out);
                    {
                        std::unique_lock<std::mutex> g(mu_);
                        cv_.wait(g, [&] { return written_ == seq; });     // in the order the blocks were handed over
                        if (!out.empty() && std::fwrite(out.data(), 1, out.size(), fp_) != out.size()) failed_ = true;
                        ++written_;
                    }
                    cv_.notify_all();
                }
            });
}

BgzfWriter::~BgzfWriter() { if (fp_) close(); }

// One or more BGZF blocks from `in` (all of it), appended to `out`.
void BgzfWriter::deflate_block(const std::vector<unsigned char> &in, std::vector<unsigned char> &outv)
{
    size_t n = in.size();
    if (n == 0) return;
    unsigned char out[0x10000];
    size_t at = 0;
    for (size_t take = n;;) {                   // shrink the input if the deflated block would not fit 64 KiB
        z_stream zs;
        std::memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, level_, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { failed_ = true; return; }
        zs.next_in = const_cast<unsigned char *>(in.data()) + at;
        zs.avail_in = (uInt)take;
        zs.next_out = out + 18;
        zs.avail_out = (uInt)(sizeof out - 18 - 8);
        const int rc = deflate(&zs, Z_FINISH);
        const size_t clen = zs.total_out;
        deflateEnd(&zs);
        if (rc != Z_STREAM_END) { take = take / 2; if (take == 0) { failed_ = true; return; } continue; }
        const size_t bsize = clen + 18 + 8;
        static const unsigned char hdr[16] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0};
        std::memcpy(out, hdr, 16);
        out[16] = (unsigned char)((bsize - 1) & 0xff);
        out[17] = (unsigned char)((bsize - 1) >> 8);
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), in.data() + at, (uInt)take);
        const uint32_t isize = (uint32_t)take;
        for (int i = 0; i < 4; ++i) { out[18 + clen + i] = (crc >> (8 * i)) & 0xff; out[22 + clen + i] = (isize >> (8 * i)) & 0xff; }
        outv.insert(outv.end(), out, out + bsize);
        at += take;
        if (at == n) return;
        take = n - at;
    }
}

void BgzfWriter::deflate_and_write(std::vector<unsigned char> &in)
{
    if (!fp_ || in.empty()) return;
    std::vector<unsigned char> out;
    deflate_block(in, out);
    if (!out.empty() && std::fwrite(out.data(), 1, out.size(), fp_) != out.size()) failed_ = true;
}

void BgzfWriter::flush_block(size_t n)
{
    if (!fp_ || n == 0) return;
    std::vector<unsigned char> blk(buf_.begin(), buf_.begin() + n);
    buf_.erase(buf_.begin(), buf_.begin() + n);
    if (!background_) { deflate_and_write(blk); return; }
    std::unique_lock<std::mutex> g(mu_);
    cv_.wait(g, [&] { return pending_.size() < 256; });          // at most 16 MB waiting for the deflater
    pending_.push_back(std::move(blk));
    g.unlock();
    cv_.notify_all();
}

// Background mode: every block handed over so far has reached the file.
void BgzfWriter::drain()
{
    if (!background_ || workers_.empty()) return;
    { std::lock_guard<std::mutex> g(mu_); closing_ = true; }
    cv_.notify_all();
    for (auto &w : workers_) w.join();
    workers_.clear();
    background_ = false;
}

void BgzfWriter::write(const char *data, size_t n)
{
    while (n > 0) {
        const size_t room = kBlockIn - buf_.size();
        const size_t take = n < room ? n : room;
        buf_.insert(buf_.end(), data, data + take);
        data += take; n -= take;
        if (buf_.size() == kBlockIn) flush_block(buf_.size());
    }
}

bool BgzfWriter::append_file(const std::string &path)
{
    if (!fp_) return false;
    flush_block(buf_.size());
    drain();
    FILE *in = std::fopen(path.c_str(), "rb");
    if (!in) return false;
    std::fseek(in, 0, SEEK_END);
    long size = std::ftell(in);
    std::fseek(in, 0, SEEK_SET);
    if (size >= 28) {                           // drop the trailing EOF marker when present
        unsigned char tail[28];
        std::fseek(in, size - 28, SEEK_SET);
        if (std::fread(tail, 1, 28, in) == 28 && std::memcmp(tail, kEofMarker, 28) == 0) size -= 28;
        std::fseek(in, 0, SEEK_SET);
    }
    std::vector<unsigned char> chunk(1 << 20);
    bool good = true;
    while (size > 0 && good) {
        const size_t want = size < (long)chunk.size() ? (size_t)size : chunk.size();
        const size_t got = std::fread(chunk.data(), 1, want, in);
        if (got == 0) { good = false; break; }
        if (std::fwrite(chunk.data(), 1, got, fp_) != got) { failed_ = true; good = false; }
        size -= (long)got;
    }
    std::fclose(in);
    return good;
}

bool BgzfWriter::close()
{
    if (!fp_) return false;
    flush_block(buf_.size());
    drain();
    if (std::fwrite(kEofMarker, 1, sizeof kEofMarker, fp_) != sizeof kEofMarker) failed_ = true;
    const bool good = std::fclose(fp_) == 0 && !failed_;
    fp_ = nullptr;
    return good;
}

BgzfReader::BgzfReader(const std::string &path)
    : fp_(std::fopen(path.c_str(), "rb")), bgzf_(false), block_addr_(0), next_addr_(0), pos_(0), eof_(false)
{
    file_at_ = UINT64_MAX; zs_ready_ = false;
    if (fp_) {
        unsigned char h[16];
        bgzf_ = std::fread(h, 1, 16, fp_) == 16 && h[0] == 0x1f && h[1] == 0x8b && (h[3] & 4) && h[12] == 'B' && h[13] == 'C';
        std::fseek(fp_, 0, SEEK_SET);
    }
}

BgzfReader::~BgzfReader()
{
    settle();
    if (zs_ready_) inflateEnd(&zs_);
    if (fp_) std::fclose(fp_);
}

bool BgzfReader::fetch(uint64_t from, std::vector<unsigned char> &out, uint64_t &at, uint64_t &next)
{
    for (;;) {
        at = from;
        unsigned char h[18];
        // sequential reading leaves the file where the next block starts: seek only after BgzfReader::seek or at the start
        if (file_at_ != at && std::fseek(fp_, (long)at, SEEK_SET) != 0) return false;
        file_at_ = UINT64_MAX;
        if (std::fread(h, 1, 18, fp_) != 18) return false;
        if (h[0] != 0x1f || h[1] != 0x8b || h[12] != 'B' || h[13] != 'C') return false;
        const size_t bsize = ((size_t)h[16] | ((size_t)h[17] << 8)) + 1;
        if (bsize < 18 + 8) return false;
        comp_.resize(bsize - 18);
        if (std::fread(comp_.data(), 1, comp_.size(), fp_) != comp_.size()) return false;
        next = at + bsize;
        file_at_ = next;
        const size_t clen = comp_.size() - 8;
        uint32_t isize = 0;
        for (int i = 0; i < 4; ++i) isize |= (uint32_t)comp_[clen + 4 + i] << (8 * i);
        if (isize == 0) { from = next; continue; }      // empty block (the EOF marker): try the next one
        out.resize(isize);
        uint32_t want_crc = 0;
        for (int i = 0; i < 4; ++i) want_crc |= (uint32_t)comp_[clen + i] << (8 * i);
        // BVC_HOST_NO_CRC=1: the trailer's CRC32 is not compared (measurement aid; htslib always compares)
        static const bool check_crc = getenv("BVC_HOST_NO_CRC") == nullptr;
        auto crc_ok = [&]() {
            if (!check_crc || bgzf_crc32(out.data(), isize) == want_crc) return true;
            g_crc_errors.fetch_add(1, std::memory_order_relaxed);
            std::fprintf(stderr, "ERROR: BGZF block at file offset %llu fails its CRC32\n", (unsigned long long)at);
            return false;
        };
        // the block decoder of inflate.cpp (both buffers whole in memory, size known); zlib only if it declines the stream
        if (fast_inflate(comp_.data(), clen, out.data(), isize) == (long)isize) return crc_ok();
        g_zlib_fallbacks.fetch_add(1, std::memory_order_relaxed);
        // one inflate state for the reader's life (inflateInit2 allocates and clears about 40 KB)
        if (!zs_ready_) {
            std::memset(&zs_, 0, sizeof zs_);
            if (inflateInit2(&zs_, -15) != Z_OK) return false;
            zs_ready_ = true;
        } else if (inflateReset(&zs_) != Z_OK) {
            return false;
        }
        zs_.next_in = comp_.data(); zs_.avail_in = (uInt)clen;
        zs_.next_out = out.data(); zs_.avail_out = isize;
        return inflate(&zs_, Z_FINISH) == Z_STREAM_END && crc_ok();
    }
}

bool BgzfReader::load_block()
{
    bool got = false;
    if (!eof_) {
        if (pool_) {
            {
                std::unique_lock<std::mutex> lk(pool_->mu_);
                if (ahead_state_ == kNone) { ahead_from_ = next_addr_; ahead_state_ = kQueued; pool_->q_.push_back(this); pool_->work_cv_.notify_one(); }
                pool_->done_cv_.wait(lk, [this] { return ahead_state_ == kReady; });
                ahead_state_ = kNone;
            }
            got = ahead_ok_;
            if (got) {
                block_.swap(ahead_block_);
                block_addr_ = ahead_at_; next_addr_ = ahead_next_;
                pool_->submit(this);                    // the block after this one, while this one is consumed
            }
        } else {
            got = fetch(next_addr_, block_, block_addr_, next_addr_);
        }
    }
    pos_ = 0;
    if (!got) { eof_ = true; block_.clear(); }
    return got;
}

void BgzfReader::settle()
{
    if (!pool_) return;
    std::unique_lock<std::mutex> lk(pool_->mu_);
    pool_->done_cv_.wait(lk, [this] { return ahead_state_ != kQueued; });
    ahead_state_ = kNone;
}

void BgzfReader::attach(InflatePool *pool)
{
    settle();
    pool_ = pool;
    if (pool_ && fp_ && !eof_) pool_->submit(this);
}

InflatePool::InflatePool(int n_threads)
{
    for (int i = 0; i < (n_threads > 0 ? n_threads : 1); ++i) th_.emplace_back([this] { run(); });
}

InflatePool::~InflatePool()
{
    { std::lock_guard<std::mutex> g(mu_); stop_ = true; }
    work_cv_.notify_all();
    for (auto &t : th_) t.join();
}

void InflatePool::submit(BgzfReader *r)
{
    std::lock_guard<std::mutex> g(mu_);
    if (r->ahead_state_ != BgzfReader::kNone) return;
    r->ahead_from_ = r->next_addr_;
    r->ahead_state_ = BgzfReader::kQueued;
    q_.push_back(r);
    work_cv_.notify_one();
}

void InflatePool::run()
{
    for (;;) {
        BgzfReader *r;
        {
            std::unique_lock<std::mutex> lk(mu_);
            work_cv_.wait(lk, [this] { return stop_ || !q_.empty(); });
            if (q_.empty()) return;                     // stop_, and nothing left to do
            r = q_.front(); q_.pop_front();
        }
        r->ahead_ok_ = r->fetch(r->ahead_from_, r->ahead_block_, r->ahead_at_, r->ahead_next_);
        { std::lock_guard<std::mutex> g(mu_); r->ahead_state_ = BgzfReader::kReady; }
        done_cv_.notify_all();
    }
}

size_t BgzfReader::read(void *dst, size_t n)
{
    size_t got = 0;
    unsigned char *d = static_cast<unsigned char *>(dst);
    while (got < n) {
        if (pos_ >= block_.size() && !load_block()) break;
        const size_t take = std::min(n - got, block_.size() - pos_);
        std::memcpy(d + got, block_.data() + pos_, take);
        pos_ += take; got += take;
    }
    return got;
}

bool BgzfReader::getline(std::string &line)
{
    line.clear();
    bool any = false;
    for (;;) {
        if (pos_ >= block_.size() && !load_block()) return any;
        any = true;
        const unsigned char *b = block_.data() + pos_;
        const unsigned char *nl = static_cast<const unsigned char *>(std::memchr(b, '\n', block_.size() - pos_));
        if (nl) { line.append(reinterpret_cast<const char *>(b), nl - b); pos_ += (nl - b) + 1; return true; }
        line.append(reinterpret_cast<const char *>(b), block_.size() - pos_);
        pos_ = block_.size();
    }
}

size_t BgzfReader::read_lines(size_t n, std::vector<char> &out, uint32_t *starts)
{
    size_t got = 0;
    bool at_line_start = true;                  // the next byte copied begins line `got`
    const size_t out0 = out.size();
    size_t pending_from = out0;                 // start (in out) of the line being assembled
    while (got < n) {
        if (pos_ >= block_.size() && !load_block()) break;
        const unsigned char *b = block_.data() + pos_, *end = block_.data() + block_.size();
        const unsigned char *p = b;
        const size_t base = out.size();         // where b[0] will land in out
        while (got < n && p < end) {
            if (at_line_start) { starts[got] = (uint32_t)(base + (size_t)(p - b)); pending_from = base + (size_t)(p - b); at_line_start = false; }
            const unsigned char *nl = static_cast<const unsigned char *>(std::memchr(p, '\n', (size_t)(end - p)));
            if (!nl) { p = end; break; }
            p = nl + 1;
            ++got;
            at_line_start = true;
        }
        out.insert(out.end(), reinterpret_cast<const char *>(b), reinterpret_cast<const char *>(p));
        pos_ += (size_t)(p - b);
    }
    if (!at_line_start) out.resize(pending_from);      // the file ended inside a line: drop the fragment
    starts[got] = (uint32_t)out.size();
    return got;
}

bool BgzfReader::seek(uint64_t voffset)
{
    settle();
    eof_ = false;
    next_addr_ = voffset >> 16;
    block_.clear(); pos_ = 0;
    if (!load_block()) return (voffset & 0xffff) == 0;
    pos_ = (size_t)(voffset & 0xffff);
    return pos_ <= block_.size();
}

bool BgzfReader::has_eof_marker(const std::string &path)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    unsigned char tail[28];
    bool good = std::fseek(f, -28, SEEK_END) == 0 && std::fread(tail, 1, 28, f) == 28 && std::memcmp(tail, kEofMarker, 28) == 0;
    std::fclose(f);
    return good;
}

}  // namespace bvchost
