// bgzf.h -- minimal BGZF (blocked gzip) reader/writer on zlib.  The reference uses htslib's bgzf_* for its
// temp-batch files and its .vcf.gz/.cvg.gz outputs (src/BaseVarC.cpp:10, 218-233, 320-330, 498-527);
// htslib is absent from the reference tree, so the format is implemented from the SAM specification (4.1).
#ifndef BVC_HOST_BGZF_H
#define BVC_HOST_BGZF_H

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <zlib.h>

namespace bvchost {

class BgzfWriter {
 public:
    // background = true: full blocks are deflated and written by a thread of the writer's own, in order; write() only
    // copies bytes.  (The compute phase's outputs: deflating the CVG text was the largest single item of its last stage.)
    explicit BgzfWriter(const std::string &path, int level = 6, bool background = false);
    ~BgzfWriter();
    bool ok() const { return fp_ != nullptr && !failed_; }
    void write(const char *data, size_t n);
    void write(const std::string &s) { write(s.data(), s.size()); }
    bool close();                               // flushes and appends the EOF marker block
    // Appends a finished BGZF file block for block (its EOF marker dropped): BGZF blocks are independent gzip
    // members, so concatenating files is concatenating their blocks -- no inflate/deflate round trip.
    bool append_file(const std::string &path);
 private:
    void flush_block(size_t n);
    void deflate_and_write(std::vector<unsigned char> &in);
    void drain();
    FILE *fp_;
    int level_;
    std::atomic<bool> failed_;      // set by whichever thread deflates
    std::vector<unsigned char> buf_;
    // background mode
    bool background_;
    std::thread worker_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::vector<unsigned char>> pending_;
    bool closing_ = false;
};

class BgzfReader {
 public:
    explicit BgzfReader(const std::string &path);
    ~BgzfReader();
    BgzfReader(const BgzfReader &) = delete;            // owns a FILE and an inflate state
    BgzfReader &operator=(const BgzfReader &) = delete;
    bool ok() const { return fp_ != nullptr; }
    bool is_bgzf() const { return bgzf_; }
    // one line without its terminator; false at end of file
    bool getline(std::string &line);
    size_t read(void *dst, size_t n);           // up to n bytes of the uncompressed stream
    bool seek(uint64_t voffset);                // BGZF virtual offset: compressed block start << 16 | within-block
    uint64_t tell() const { return (block_addr_ << 16) | (uint64_t)pos_; }
    static bool has_eof_marker(const std::string &path);   // bgzf_check_EOF
 private:
    bool load_block();
    FILE *fp_;
    bool bgzf_;
    uint64_t block_addr_;
    uint64_t next_addr_;
    std::vector<unsigned char> block_;
    size_t pos_;
    bool eof_;
    uint64_t file_at_;                          // file offset the FILE is known to stand at (UINT64_MAX: unknown)
    std::vector<unsigned char> comp_;           // the compressed block, reused
    z_stream zs_;                               // one inflate state for the reader's life
    bool zs_ready_;
};

}  // namespace bvchost
#endif
