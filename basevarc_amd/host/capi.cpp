// capi.cpp -- C entry points of the host-side pieces, for tests and for callers in other languages.
#include <algorithm>
#include <atomic>
#include <thread>
#include <chrono>
#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include <sstream>

#include "bam.h"
#include "bgzf.h"
#include "inflate.h"
#include "pileup.h"
#include "stats.h"

using namespace bvchost;

extern "C" {

double bvchost_erfc(double x) { return kf_erfc(x); }
double bvchost_normsf(double x) { return normsf(x); }
double bvchost_fisher_phred(int n11, int n12, int n21, int n22) { return bt_fisher_exact(n11, n12, n21, n22); }
double bvchost_fisher_two_sided(int n11, int n12, int n21, int n22)
{
    double l, r, t;
    kt_fisher_exact(n11, n12, n21, n22, &l, &r, &t);
    return t;
}
double bvchost_ranksum(const double *x, int n1, const double *y, int n2)
{
    std::vector<double> a(x, x + n1), b(y, y + n2);
    return RankSumTest(a, b);
}

// Parses the concatenated batch lines of ONE position (lines separated by '\n') and formats the CVG line and, when
// `bt` says called, the VCF line.  Returns the number of bytes needed; output is truncated to cap.
static size_t copy_out(const std::string &s, char *out, size_t cap)
{
    if (out && cap) { const size_t n = s.size() < cap - 1 ? s.size() : cap - 1; std::memcpy(out, s.data(), n); out[n] = 0; }
    return s.size() + 1;
}

void bvchost_reset_parser(void) { reset_parser_carry(); }

// inflate.cpp against zlib (tests): bytes written or -1
long bvchost_fast_inflate(const unsigned char *in, size_t n, unsigned char *out, size_t cap) { return fast_inflate(in, n, out, cap); }
// orchestration hooks (tests/test_host.py): the functions main.cpp itself calls
void bvchost_thread_window(int64_t psize, int32_t thread, int32_t ithread, int64_t *lo, int64_t *hi)
{
    size_t l, h;
    thread_window((size_t)psize, thread, ithread, l, h);
    *lo = (int64_t)l; *hi = (int64_t)h;
}
int32_t bvchost_device_of_thread(int32_t ithread, int32_t devices_present, int32_t gpus_option) { return device_of_thread(ithread, devices_present, gpus_option); }
int32_t bvchost_merge_subfiles(const char *out_prefix, const char *suffix, int32_t thread)
{
    BgzfWriter dst(std::string(out_prefix) + suffix);
    if (!dst.ok() || !merge_subfiles(out_prefix, suffix, thread, dst)) return 0;
    return dst.close() ? 1 : 0;
}
// BgzfReader::read_lines, n lines per call until the file ends: the concatenated lines into out (returns their byte count, or -1 when cap is
// too small / -2 when a `starts` entry does not sit at the beginning of a line), the number of lines in *n_lines.  threads > 0: with read-ahead.
int64_t bvchost_bgzf_read_lines(const char *path, int64_t n_per_call, int32_t threads, char *out, int64_t cap, int64_t *n_lines)
{
    std::unique_ptr<InflatePool> pool(threads > 0 ? new InflatePool(threads) : nullptr);
    BgzfReader rd(path);
    if (pool) rd.attach(pool.get());
    std::vector<char> buf;
    std::vector<uint32_t> starts((size_t)n_per_call + 1);
    int64_t total = 0, lines = 0;
    for (;;) {
        buf.assign(7, '#');                                 // lines are appended behind what the buffer holds
        const size_t got = rd.read_lines((size_t)n_per_call, buf, starts.data());
        for (size_t i = 0; i < got; ++i) {
            if (starts[i] < 7 || starts[i + 1] <= starts[i] || buf[starts[i + 1] - 1] != '\n') return -2;
            if (i > 0 && buf[starts[i] - 1] != '\n') return -2;
        }
        if (starts[got] != buf.size() || (got > 0 && starts[0] != 7)) return -2;
        if (total + (int64_t)buf.size() - 7 > cap) return -1;
        std::memcpy(out + total, buf.data() + 7, buf.size() - 7);
        total += (int64_t)buf.size() - 7;
        lines += (int64_t)got;
        if (got < (size_t)n_per_call) break;
    }
    *n_lines = lines;
    return total;
}
long bvchost_zlib_fallbacks(void) { return bgzf_zlib_fallbacks(); }
long bvchost_crc_errors(void) { return bgzf_crc_errors(); }
uint32_t bvchost_crc32(const unsigned char *buf, size_t len) { return bgzf_crc32(buf, len); }

// Micro-benchmarks of the two CPU costs of a text temp batch (tools/host_micro.py): seconds per pass over the input.
// parse: every line of `text` through parse_pileup_line into one reused SiteColumn; inflate: every BGZF block of a file image
// through fast_inflate (use_zlib = 0) -- *out_bytes gets the inflated size of one pass.
double bvchost_bench_parse(const char *text, size_t len, int reps, int64_t *entries_out)
{
    SiteColumn col;
    reset_parser_carry();
    int64_t entries = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) {
        const char *p = text, *end = text + len;
        while (p < end) {
            const char *nl = static_cast<const char *>(std::memchr(p, '\n', (size_t)(end - p)));
            const size_t n = nl ? (size_t)(nl - p) : (size_t)(end - p);
            col.clear();
            parse_pileup_line(p, n, 0, col);
            entries += (int64_t)col.aiv.size();
            p += n + 1;
        }
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (entries_out) *entries_out = entries / (reps > 0 ? reps : 1);
    return dt / (reps > 0 ? reps : 1);
}
double bvchost_bench_inflate(const unsigned char *file, size_t len, int reps, int64_t *out_bytes)
{
    std::vector<unsigned char> out(1 << 16);
    int64_t total = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) {
        size_t off = 0;
        total = 0;
        while (off + 28 <= len) {
            // the BGZF header fetch() insists on: gzip magic, the 'BC' subfield, a block that holds its 18 + 8 frame bytes
            if (file[off] != 0x1f || file[off + 1] != 0x8b || file[off + 12] != 'B' || file[off + 13] != 'C') return -1.0;
            const size_t bsize = ((size_t)file[off + 16] | ((size_t)file[off + 17] << 8)) + 1;
            if (bsize < 18 + 8 || off + bsize > len) return -1.0;
            const unsigned char *tail = file + off + bsize - 4;
            const size_t isize = (size_t)tail[0] | ((size_t)tail[1] << 8) | ((size_t)tail[2] << 16) | ((size_t)tail[3] << 24);
            if (isize > out.size()) return -1.0;
            if (isize && fast_inflate(file + off + 18, bsize - 18 - 8, out.data(), isize) != (long)isize) return -2.0;
            total += (int64_t)isize;
            off += bsize;
        }
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (out_bytes) *out_bytes = total;
    return dt / (reps > 0 ? reps : 1);
}

// site handle API (tests): build a SiteColumn from batch lines, then ask for its pieces
struct bvchost_site { SiteColumn col; };

bvchost_site *bvchost_site_parse(const char *lines, int32_t pos)
{
    bvchost_site *s = new bvchost_site();
    s->col.pos = pos;
    int32_t j = 0;
    const char *p = lines;
    while (*p) {
        const char *nl = std::strchr(p, '\n');
        const size_t len = nl ? (size_t)(nl - p) : std::strlen(p);
        j += parse_pileup_line(p, len, j, s->col);
        p += len + (nl ? 1 : 0);
    }
    return s;
}
// The same from binary batch records: `blob` = per batch  u32 samples-in-batch | u32 payload bytes | payload.
bvchost_site *bvchost_site_parse_bin(const char *blob, size_t len, int32_t n_batches, int32_t pos)
{
    bvchost_site *s = new bvchost_site();
    s->col.pos = pos;
    int32_t j = 0;
    const unsigned char *p = reinterpret_cast<const unsigned char *>(blob), *end = p + len;
    auto u32 = [](const unsigned char *q) { return (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24); };
    for (int32_t b = 0; b < n_batches && end - p >= 8; ++b) {
        const uint32_t n_in = u32(p), n = u32(p + 4);
        p += 8;
        if ((size_t)(end - p) < n || !parse_pileup_bin(p, n, j, s->col)) { delete s; return nullptr; }
        p += n;
        j += (int32_t)n_in;
    }
    return s;
}
void bvchost_site_free(bvchost_site *s) { delete s; }
int32_t bvchost_site_size(const bvchost_site *s) { return (int32_t)s->col.aiv.size(); }
// fields: 0 base, 1 mapq, 2 qual, 3 rpr, 4 strand, 5 is_indel, 6 sample index
int32_t bvchost_site_field(const bvchost_site *s, int32_t k, int field)
{
    const Entry &a = s->col.aiv[(size_t)k];
    switch (field) {
    case 0: return a.base; case 1: return a.mapq; case 2: return a.qual; case 3: return a.rpr;
    case 4: return a.strand; case 5: return a.is_indel; default: return s->col.sample[(size_t)k];
    }
}
size_t bvchost_cvg_line(const bvchost_site *s, const char *chr, int8_t ref_base, const bvc_group_result *grp, int n_groups,
                        char *out, size_t cap)
{
    return copy_out(cvg_line(chr, s->col.pos, ref_base, s->col, grp, n_groups), out, cap);
}
size_t bvchost_vcf_line(const bvchost_site *s, const bvc_site_result *bt, const char *chr, int8_t ref_base, int32_t n_samples,
                        const char *extra_info_keys, const char *extra_info_vals, char *out, size_t cap)
{
    std::map<std::string, std::string> info;
    if (extra_info_keys && extra_info_vals) {                    // ';'-separated parallel lists
        std::string k(extra_info_keys), v(extra_info_vals);
        size_t a = 0, b = 0;
        while (a < k.size()) {
            const size_t ea = k.find(';', a), eb = v.find(';', b);
            info.insert({k.substr(a, ea - a), v.substr(b, eb - b)});
            if (ea == std::string::npos) break;
            a = ea + 1; b = eb + 1;
        }
    }
    return copy_out(vcf_line(*bt, chr, s->col.pos, ref_base, s->col, info, n_samples), out, cap);
}
size_t bvchost_format_token(int base, int mapq, int qual, int rpr, int strand, const char *indel, char *out, size_t cap)
{
    AlleleInfo a;
    a.base = (uint8_t)base; a.mapq = (uint8_t)mapq; a.qual = (uint8_t)qual; a.rpr = (uint8_t)rpr; a.strand = (uint8_t)strand;
    if (indel) { a.is_indel = 1; a.indel = indel; }
    std::string s;
    format_pileup_token(base < 0 ? nullptr : &a, s);
    return copy_out(s, out, cap);
}

// The per-sample pileup rule on caller-made reads (tests).  `reads`: one read per line,
// "pos0 flag mapq cigar seq q,q,q,..." (pos0 = 0-based leftmost, qualities raw phred); `positions`: ascending, 1-based.
// Output: the temp-batch tokens of those positions for this one sample, in order (". " where the sample has no entry).
size_t bvchost_pileup_tokens(const char *reads, const int32_t *positions, int32_t n_pos, int32_t rg_s, const char *refseq,
                             char *out, size_t cap)
{
    std::vector<BamRecord> rv;
    std::istringstream in(reads);
    std::string line;
    while (std::getline(in, line)) {
        if (line.empty()) continue;
        std::istringstream ls(line);
        BamRecord r;
        int flag = 0, mapq = 0;
        std::string cigar, quals;
        ls >> r.pos >> flag >> mapq >> cigar >> r.seq >> quals;
        r.flag = (uint16_t)flag; r.mapq = (uint8_t)mapq; r.ref_id = 0;
        for (size_t i = 0; i < cigar.size();) {
            int32_t l = 0;
            while (i < cigar.size() && cigar[i] >= '0' && cigar[i] <= '9') l = l * 10 + (cigar[i++] - '0');
            r.cigar.emplace_back(cigar[i++], l);
        }
        std::istringstream qs(quals);
        std::string tok;
        while (std::getline(qs, tok, ',')) r.qual.push_back((char)std::atoi(tok.c_str()));
        rv.push_back(r);
    }
    std::vector<int32_t> pv(positions, positions + n_pos);
    PosAlleleMap m;
    std::string res;
    try {
        find_snp_at_pos(rv, rg_s, refseq, pv, m);
    } catch (const std::exception &e) {
        return copy_out(std::string("!") + e.what(), out, cap);
    }
    for (int32_t p : pv) {
        auto it = m.find(p);
        format_pileup_token(it == m.end() ? nullptr : &it->second, res);
    }
    return copy_out(res, out, cap);
}

// Synthetic temp-batch files for the host benchmark (tools/host_bench.py): <out>.tmp.thread.<t>/batch.<b> for
// n_samples samples in batches of `batch`, n_pos positions split over `thread` windows as the load phase splits them,
// each sample covered with probability cov_permille / 1000 (reference base A, errors at the rate the quality states).
// Counter-based (splitmix64 of seed, position, sample): the text and the binary form hold the same entries.
// The directories must exist.  Returns the number of entries written, -1 on an I/O error.
// Test hook: writes `n` bytes in pieces of `piece` bytes through a BgzfWriter (background = its own deflate thread) and
// closes it.  Returns 1 when the writer reports success.
int bvchost_bgzf_write(const char *path, const char *data, int64_t n, int64_t piece, int level, int background)
{
    BgzfWriter w(path, level, background != 0, background);        // background = the number of deflating threads (0: the caller's)
    if (!w.ok()) return 0;
    for (int64_t i = 0; i < n; i += piece) w.write(data + i, (size_t)(n - i < piece ? n - i : piece));
    return w.close() ? 1 : 0;
}

// Test hook: reads `n_files` BGZF files the way the position loop does -- round robin, one line (by_line) or `piece` bytes
// of each per turn -- through readers attached to an InflatePool of `threads` threads (0: no pool), a seek back to the
// start of file 0 after `seek_after` turns, and returns a 64-bit FNV-1a hash over everything read (order included).
uint64_t bvchost_bgzf_read_hash(const char *const *paths, int32_t n_files, int32_t threads, int32_t by_line, int64_t piece,
                                int64_t seek_after, int64_t *bytes_out)
{
    uint64_t h = 1469598103934665603ULL;
    auto mix = [&h](const char *p, size_t n) { for (size_t i = 0; i < n; ++i) { h ^= (unsigned char)p[i]; h *= 1099511628211ULL; } };
    int64_t total = 0;
    {
        std::unique_ptr<InflatePool> pool(threads > 0 ? new InflatePool(threads) : nullptr);
        std::vector<std::unique_ptr<BgzfReader>> rd;
        for (int32_t i = 0; i < n_files; ++i) {
            rd.emplace_back(new BgzfReader(paths[i]));
            if (!rd.back()->ok()) return 0;
            if (pool) rd.back()->attach(pool.get());
        }
        std::vector<char> live((size_t)n_files, 1);
        std::string line;
        std::vector<char> buf((size_t)(piece > 0 ? piece : 1));
        int n_live = n_files;
        for (int64_t turn = 0; n_live > 0; ++turn) {
            if (turn == seek_after && n_files > 0) { rd[0]->seek(0); if (!live[0]) { live[0] = 1; ++n_live; } }
            for (int32_t i = 0; i < n_files; ++i) {
                if (!live[(size_t)i]) continue;
                if (by_line) {
                    if (!rd[(size_t)i]->getline(line)) { live[(size_t)i] = 0; --n_live; continue; }
                    mix(line.data(), line.size()); mix("\n", 1); total += (int64_t)line.size() + 1;
                } else {
                    const size_t got = rd[(size_t)i]->read(buf.data(), buf.size());
                    if (got == 0) { live[(size_t)i] = 0; --n_live; continue; }
                    mix(buf.data(), got); total += (int64_t)got;
                }
            }
        }
    }   // readers go before the pool
    if (bytes_out) *bytes_out = total;
    return h;
}

int64_t bvchost_write_synth_batches(const char *out_prefix, int32_t n_samples, int32_t n_pos, int32_t thread, int32_t batch,
                                    int32_t cov_permille, uint64_t seed, int32_t bin)
{
    auto mix = [](uint64_t z) { z += 0x9E3779B97F4A7C15ULL; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
                                z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); };
    const int32_t nb = 1 + (n_samples - 1) / batch;
    const size_t window = (size_t)n_pos % thread + (size_t)n_pos / thread;
    // text and binary batches at zlib's level 6, what bt_r (host/main.cpp) and the reference's bgzf_write write; "raw": stored.
    // BVC_SYNTH_LEVEL overrides it (round 4's records were taken on level-1 files).
    const int level = bin == 2 ? 0 : (getenv("BVC_SYNTH_LEVEL") ? atoi(getenv("BVC_SYNTH_LEVEL")) : 6);
    std::atomic<int64_t> entries(0);
    std::atomic<int> failed(0), next_t(0);
    auto work = [&]() {
        std::string out, payload;
        for (int32_t t; (t = next_t++) < thread;) {
            const size_t lo = std::min((size_t)n_pos, (size_t)t * window);
            const size_t hi = t == thread - 1 ? (size_t)n_pos : std::min((size_t)n_pos, (size_t)(t + 1) * window);
            int64_t mine = 0;
            for (int32_t ib = 0; ib < nb; ++ib) {
                const int32_t j0 = ib * batch, j1 = std::min(n_samples, (ib + 1) * batch);
                // bin: 0 text, 1 binary deflated, 2 binary stored ("raw")
                BgzfWriter fp(std::string(out_prefix) + ".tmp.thread." + std::to_string(t) + "/batch." + std::to_string(ib), level);
                if (!fp.ok()) { failed = 1; return; }
                std::string names;
                for (int32_t j = j0; j < j1; ++j) names += "S" + std::to_string(j) + "\t";
                names += "\n";
                out.clear();
                if (bin) bin_batch_header((uint32_t)(j1 - j0), names, out); else out = names;
                fp.write(out);
                for (size_t p = lo; p < hi; ++p) {
                    out.clear(); payload.clear();
                    for (int32_t j = j0; j < j1; ++j) {
                        const uint64_t h = mix(seed * 0x100000001B3ULL + p * 0x9E3779B1ULL + (uint64_t)j);
                        const bool covered = (int32_t)(h % 1000) < cov_permille;
                        AlleleInfo a;
                        if (covered) {
                            a.mapq = (uint8_t)(20 + (h >> 24) % 41); a.qual = (uint8_t)(10 + (h >> 32) % 31);
                            // a sequencing error with the probability the quality states, so that sites are monomorphic
                            // the way real ones mostly are (every 64th position carries a real ALT at frequency 2 %)
                            const bool err = (double)((h >> 10) % 1000000) < 1e6 * std::pow(10.0, -0.1 * a.qual);
                            const bool alt = (p % 64) == 7 && ((h >> 44) % 50) == 0;
                            a.base = err ? (uint8_t)(1 + (h >> 20) % 3) : (alt ? 2 : 0);
                            a.rpr = (uint8_t)(1 + (h >> 40) % 150); a.strand = (uint8_t)((h >> 50) & 1);
                            ++mine;
                        }
                        if (bin) { if (covered) bin_batch_entry(a, (uint32_t)(j - j0), payload); }
                        else format_pileup_token(covered ? &a : nullptr, out);
                    }
                    if (bin) {
                        const uint32_t n = (uint32_t)payload.size();
                        for (int k = 0; k < 4; ++k) out.push_back((char)((n >> (8 * k)) & 0xff));
                        out += payload;
                    } else {
                        out += "\n";
                    }
                    fp.write(out);
                }
                if (!fp.close()) { failed = 1; return; }
            }
            entries += mine;
        }
    };
    // the threads' files are independent: written side by side (generating 1e5-sample batches was minutes on one thread)
    std::vector<std::thread> ws;
    for (int i = 0; i < std::max(1, std::min<int>(thread, 16)); ++i) ws.emplace_back(work);
    for (auto &w : ws) w.join();
    return failed ? -1 : entries.load();
}

}  // extern "C"
