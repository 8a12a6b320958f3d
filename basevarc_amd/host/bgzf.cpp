#include "bgzf.h"

#include <zlib.h>

#include <cstring>

namespace bvchost {

static const size_t kBlockIn = 0xff00;          // uncompressed bytes per block
static const unsigned char kEofMarker[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0,
                                             0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};

BgzfWriter::BgzfWriter(const std::string &path, int level, bool background)
    : fp_(std::fopen(path.c_str(), "wb")), level_(level), failed_(false), background_(background && fp_ != nullptr)
{
    buf_.reserve(kBlockIn);
    if (background_)
        worker_ = std::thread([this] {
            for (;;) {
                std::vector<unsigned char> blk;
                {
                    std::unique_lock<std::mutex> g(mu_);
                    cv_.wait(g, [&] { return !pending_.empty() || closing_; });
                    if (pending_.empty()) return;
                    blk.swap(pending_.front());
                    pending_.pop_front();
                }
                cv_.notify_all();                   // room for the producer
                deflate_and_write(blk);
            }
        });
}

BgzfWriter::~BgzfWriter() { if (fp_) close(); }

// One or more BGZF blocks from `in` (all of it), written to the file.
void BgzfWriter::deflate_and_write(std::vector<unsigned char> &in)
{
    size_t n = in.size();
    if (!fp_ || n == 0) return;
    unsigned char out[0x10000];
    size_t at = 0;
    for (size_t take = n;;) {                   // shrink the input if the deflated block would not fit 64 KiB
        z_stream zs;
        std::memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, level_, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { failed_ = true; return; }
        zs.next_in = in.data() + at;
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
        if (std::fwrite(out, 1, bsize, fp_) != bsize) failed_ = true;
        at += take;
        if (at == n) return;
        take = n - at;
    }
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
    if (!background_ || !worker_.joinable()) return;
    { std::lock_guard<std::mutex> g(mu_); closing_ = true; }
    cv_.notify_all();
    worker_.join();
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
    if (zs_ready_) inflateEnd(&zs_);
    if (fp_) std::fclose(fp_);
}

bool BgzfReader::load_block()
{
    while (!eof_) {
        block_addr_ = next_addr_;
        unsigned char h[18];
        // sequential reading leaves the file where the next block starts: seek only after BgzfReader::seek or at the start
        if (file_at_ != block_addr_ && std::fseek(fp_, (long)block_addr_, SEEK_SET) != 0) { eof_ = true; break; }
        file_at_ = UINT64_MAX;
        if (std::fread(h, 1, 18, fp_) != 18) { eof_ = true; break; }
        if (h[0] != 0x1f || h[1] != 0x8b || h[12] != 'B' || h[13] != 'C') { eof_ = true; break; }
        const size_t bsize = ((size_t)h[16] | ((size_t)h[17] << 8)) + 1;
        if (bsize < 18 + 8) { eof_ = true; break; }
        comp_.resize(bsize - 18);
        if (std::fread(comp_.data(), 1, comp_.size(), fp_) != comp_.size()) { eof_ = true; break; }
        next_addr_ = block_addr_ + bsize;
        file_at_ = next_addr_;
        const size_t clen = comp_.size() - 8;
        uint32_t isize = 0;
        for (int i = 0; i < 4; ++i) isize |= (uint32_t)comp_[clen + 4 + i] << (8 * i);
        block_.resize(isize);
        pos_ = 0;
        if (isize == 0) continue;               // empty block (the EOF marker): try the next one
        // one inflate state for the reader's life (inflateInit2 allocates and clears about 40 KB: per block it cost a
        // tenth of the inflate itself)
        if (!zs_ready_) {
            std::memset(&zs_, 0, sizeof zs_);
            if (inflateInit2(&zs_, -15) != Z_OK) { eof_ = true; break; }
            zs_ready_ = true;
        } else if (inflateReset(&zs_) != Z_OK) {
            eof_ = true; break;
        }
        zs_.next_in = comp_.data(); zs_.avail_in = (uInt)clen;
        zs_.next_out = block_.data(); zs_.avail_out = isize;
        const int rc = inflate(&zs_, Z_FINISH);
        if (rc != Z_STREAM_END) { eof_ = true; block_.clear(); break; }
        return true;
    }
    block_.clear(); pos_ = 0;
    return false;
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

bool BgzfReader::seek(uint64_t voffset)
{
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
