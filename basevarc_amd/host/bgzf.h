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
    // workers > 1 (background mode): that many threads deflate, each block is written when the ones before it have been (round 5:
    // one thread deflating the VCF text -- 400 KB a called position at 1e5 samples -- was what a position loop ended up waiting for).
    explicit BgzfWriter(const std::string &path, int level = 6, bool background = false, int workers = 1);
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
    void deflate_block(const std::vector<unsigned char> &in, std::vector<unsigned char> &out);
    void deflate_and_write(std::vector<unsigned char> &in);
    void drain();
    FILE *fp_;
    int level_;
    std::atomic<bool> failed_;      // set by whichever thread deflates
    std::vector<unsigned char> buf_;
    // background mode
    bool background_;
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::vector<unsigned char>> pending_;
    uint64_t taken_ = 0, written_ = 0;          // blocks handed to a worker / whose bytes are in the file
    bool closing_ = false;
};

class BgzfReader;

// Read-ahead for BgzfReaders: a few threads that inflate, for every reader attached to the pool, the block after the one
// being consumed.  The position loop of the host program reads one line (or record) of every temp batch per position --
// a hundred readers per thread at 1e5 samples -- and inflating their blocks was 70 % of the loop with the text form.
long bgzf_zlib_fallbacks();                       // blocks the fast decoder (inflate.cpp) declined so far: zlib took them
long bgzf_crc_errors();                           // blocks whose inflated bytes failed the CRC32 of their trailer so far
uint32_t bgzf_crc32(const unsigned char *buf, size_t len);   // CRC32 (RFC 1952) of a buffer: PCLMULQDQ folding where the CPU has it

class InflatePool {
 public:
    explicit InflatePool(int n_threads);
    ~InflatePool();
    InflatePool(const InflatePool &) = delete;
    InflatePool &operator=(const InflatePool &) = delete;
 private:
    friend class BgzfReader;
    void submit(BgzfReader *r);                 // r's read-ahead runs on a worker
    void wait(BgzfReader *r);                   // until it is done
    void run();
    std::mutex mu_;
    std::condition_variable work_cv_, done_cv_;
    std::deque<BgzfReader *> q_;
    bool stop_ = false;
    std::vector<std::thread> th_;
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
    // Appends the next n lines WITH their '\n' to `out` and notes where each starts: starts[i] = offset of line i in `out`,
    // starts[got] = one past the last line's '\n'.  Returns the lines read (fewer than n: the file ended; a last line without a
    // newline is not returned).  The tiles the device parses take a batch's lines this way: no per-line copy, no tokenising.
    size_t read_lines(size_t n, std::vector<char> &out, uint32_t *starts);
    bool seek(uint64_t voffset);                // BGZF virtual offset: compressed block start << 16 | within-block
    uint64_t tell() const { return (block_addr_ << 16) | (uint64_t)pos_; }
    static bool has_eof_marker(const std::string &path);   // bgzf_check_EOF
    // From now on the block after the current one is inflated by `pool` while this one is consumed.  The pool must
    // outlive the reader.
    void attach(InflatePool *pool);
 private:
    friend class InflatePool;
    bool load_block();
    // the first non-empty block at or after file offset `from`, inflated into `out`; false at the end of the file
    bool fetch(uint64_t from, std::vector<unsigned char> &out, uint64_t &at, uint64_t &next);
    void settle();                              // waits for a read-ahead in flight and drops its result
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
    // read-ahead (fp_, comp_, zs_, file_at_ and the ahead_* fields belong to the worker while ahead_state_ is kQueued)
    enum { kNone = 0, kQueued = 1, kReady = 2 };
    InflatePool *pool_ = nullptr;
    int ahead_state_ = kNone;                   // guarded by the pool's mutex
    bool ahead_ok_ = false;
    uint64_t ahead_from_ = 0, ahead_at_ = 0, ahead_next_ = 0;
    std::vector<unsigned char> ahead_block_;
};

}  // namespace bvchost
#endif
