// main.cpp -- `BaseVarC basetype` on MI355X: the reference's command line, temp-batch files and outputs, with
// the per-site BaseType work of a whole tile of positions handed to libbvc in one call.
//
// Counterparts in the reference (src/BaseVarC.cpp): main :151-174, runBaseType :176-314, bt_r :467-534,
// bt_s :316-465, bt_f :536-669, parseOptions :797-827.  Same options and defaults, same file names
// (<out>.vcf.gz, <out>.cvg.gz, <out>.tmp.thread.<t>/batch.<b>), same --load/--rerun/--keep_tmp behaviour.
// Additive: --gpus <n> (devices to spread the threads over; default all), --tile <sites per device call>.
#include <getopt.h>
#include <sched.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <iomanip>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <future>
#include <iostream>
#include <chrono>
#include <condition_variable>
#include <sys/resource.h>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <thread>
#include <tuple>
#include <unordered_map>

#include "bam.h"
#include "bgzf.h"
#include "pileup.h"

using namespace bvchost;

static const char *BASEVARC_USAGE_MESSAGE =
    "Contact: Zilong Li [zimusen94@gmail.com]\n"
    "Usage  : BaseVarC <command> [options]\n\n"
    "Commands:\n"
    "         basetype       Variants Caller\n"
    "         popmatrix      Create population matrix\n"
    "         concat         Concat popmatrix\n";

static const char *BASETYPE_MESSAGE =
    "Commands: BaseVarC basetype\n"
    "Usage   : BaseVarC basetype [options]\n\n"
    "Options :\n"
    "  --input,      -i         BAM/CRAM file list, one file per row\n"
    "  --output,     -o         Output file prefix\n"
    "  --reference,  -r         Reference file\n"
    "  --region,     -s         Samtools-like region <chr:start-end>\n"
    "  --group,      -g         Population group information <SampleID Group>\n"
    "  --mapq,       -q <INT>   Mapping quality >= INT [10]\n"
    "  --thread,     -t <INT>   Number of threads\n"
    "  --batch,      -b <INT>   Number of samples each batch\n"
    "  --maf,        -a <FLOAT> Minimum allele count frequency [min(0.001, 100/N, maf)]\n"
    "  --load,                  Load data only\n"
    "  --rerun,                 Read previous loaded data and rerun\n"
    "  --keep_tmp,              Don't remove tmp files when basetype finished\n"
    "  --verbose,    -v         Set verbose output\n"
    "  --gpus           <INT>   MI355X devices to use [all]\n"
    "  --tile           <INT>   Positions per device call [4096]\n"
    "  --tmp-format     <STR>   Temp-batch files written by the load phase: text (as BaseVarC), bin (binary\n"
    "                           records, deflated) or raw (binary records, stored: fastest to read back) [text]\n";

namespace opt {
static bool verbose = false, rerun = false, load = false, keep_tmp = false;
static int mapq = 10, thread = 1, batch = 10, gpus = 0, tile = 4096;
static double maf = 0.001;
static std::string input, reference, posfile, group, region, output, tmp_format = "text";
}  // namespace opt

static const char *shortopts = "hva:i:r:p:s:o:q:t:b:g:";
static const struct option longopts[] = {
    {"help", no_argument, NULL, 'h'},        {"verbose", no_argument, NULL, 'v'},
    {"keep_tmp", no_argument, NULL, 6},      {"load", no_argument, NULL, 7},
    {"rerun", no_argument, NULL, 8},         {"maf", required_argument, NULL, 'a'},
    {"input", required_argument, NULL, 'i'}, {"reference", required_argument, NULL, 'r'},
    {"posfile", required_argument, NULL, 'p'}, {"group", required_argument, NULL, 'g'},
    {"region", required_argument, NULL, 's'}, {"output", required_argument, NULL, 'o'},
    {"batch", required_argument, NULL, 'b'}, {"thread", required_argument, NULL, 't'},
    {"mapq", required_argument, NULL, 'q'},  {"gpus", required_argument, NULL, 9},
    {"tile", required_argument, NULL, 10},   {"tmp-format", required_argument, NULL, 11},
    {NULL, 0, NULL, 0}};

static void parse_options(int argc, char **argv, const char *msg)          // src/BaseVarC.cpp:797-827
{
    bool die = false;
    for (int c; (c = getopt_long(argc, argv, shortopts, longopts, NULL)) != -1;) {
        std::istringstream arg(optarg != NULL ? optarg : "");
        switch (c) {
        case 'q': arg >> opt::mapq; break;
        case 'b': arg >> opt::batch; break;
        case 't': arg >> opt::thread; break;
        case 'i': arg >> opt::input; break;
        case 'r': arg >> opt::reference; break;
        case 'p': arg >> opt::posfile; break;
        case 's': arg >> opt::region; break;
        case 'g': arg >> opt::group; break;
        case 'o': arg >> opt::output; break;
        case 'a': arg >> opt::maf; break;
        case 8: opt::rerun = true; break;
        case 7: opt::load = true; break;
        case 6: opt::keep_tmp = true; break;
        case 9: arg >> opt::gpus; break;
        case 10: arg >> opt::tile; break;
        case 11: arg >> opt::tmp_format; break;
        case 'v': opt::verbose = true; break;
        default: die = true;
        }
    }
    if (opt::tmp_format != "text" && opt::tmp_format != "bin" && opt::tmp_format != "raw") die = true;
    if (die || opt::input.empty() || opt::output.empty()) {
        std::cerr << msg;
        std::exit(die ? EXIT_FAILURE : EXIT_SUCCESS);
    }
}

static std::tuple<std::string, int32_t, int32_t> splitrg(std::string rg)   // src/BaseVarUtils.h:49-75
{
    if (rg.find(":") == std::string::npos || rg.find("-") == std::string::npos)
        throw std::invalid_argument("region must be feed with samtools-like format, i.e. chr:start-end");
    size_t p = rg.find(":");
    std::string chr = rg.substr(0, p);
    rg.erase(0, p + 1);
    p = rg.find("-");
    const int32_t s = std::stoi(rg.substr(0, p));
    rg.erase(0, p + 1);
    const int32_t e = std::stoi(rg) - 1;                                 // the reference's end - 1
    return std::make_tuple(chr, s, e);
}

static bool file_exists(const std::string &f) { struct stat st; return ::stat(f.c_str(), &st) == 0; }

static std::string tmp_name(int ithread, int ibatch)
{
    return opt::output + ".tmp.thread." + std::to_string(ithread) + "/batch." + std::to_string(ibatch);
}

// ---- phase 1: BAM -> temp-batch pileup text (bt_r, src/BaseVarC.cpp:467-534) ------------------------------------
static void bt_r(const std::vector<std::string> &bams, const std::vector<int32_t> &pv, const std::string &refseq,
                 const std::string &chr, int32_t rg_s, int32_t rg_e, int nb, int bc, int ib, int thread)
{
    const size_t b0 = (size_t)ib * bc, b1 = (ib == nb - 1) ? bams.size() : (size_t)(ib + 1) * bc;
    std::vector<PosAlleleMap> allele_mv;
    std::string names;
    for (size_t b = b0; b < b1; ++b) {
        BamFile reader;
        if (!reader.open(bams[b])) throw std::runtime_error("ERROR: can not open file " + bams[b]);
        if (!reader.sorted())
            throw std::runtime_error("ERROR: BAM file does not appear to be sorted (no SO:coordinate) found in header.\n       Sorted BAMs are required.");
        const std::string sm = reader.sample_name();
        PosAlleleMap m;
        std::vector<BamRecord> rv;
        const int rid = reader.ref_index(chr);
        // region padded by 1000 on both sides (gr.Pad(1000), src/BamProcess.cpp:290); [beg, end) 0-based
        if (rid < 0 || !reader.fetch(rid, std::max(0, rg_s - 1 - 1000), rg_e + 1000, opt::mapq, rv) || rv.empty())
            std::cerr << "warning: " << sm << " region " << opt::region << " is empty." << std::endl;
        else
            find_snp_at_pos(rv, rg_s, refseq, pv, m);
        allele_mv.push_back(std::move(m));
        names += sm + '\t';
    }
    names += "\n";
    const bool bin = opt::tmp_format != "text";
    // "raw": the same binary records in stored (deflate level 0) BGZF blocks -- still valid BGZF with the EOF block the
    // --rerun check looks for, but reading them back is a copy instead of an inflate (half of the compute phase's
    // host time once the tokenising is gone)
    const int level = opt::tmp_format == "raw" ? 0 : 6;
    std::vector<BgzfWriter *> fpv;
    for (int i = 0; i < thread; ++i) {
        BgzfWriter *fp = new BgzfWriter(tmp_name(i, ib), level);
        if (!fp->ok()) throw std::runtime_error("ERROR: fail to write " + tmp_name(i, ib));
        if (bin) { std::string h; bin_batch_header((uint32_t)(b1 - b0), names, h); fp->write(h); }
        else fp->write(names);
        fpv.push_back(fp);
    }
    const size_t psize = pv.size();
    const size_t window = psize % thread + psize / thread;
    std::string out, payload;
    for (size_t i = 0, j = 0; i < psize; ++i) {
        const int32_t p = pv[i];
        out.clear();
        if (bin) {                              // one record per position: u32 payload bytes + the entries present
            payload.clear();
            for (size_t k = 0; k < allele_mv.size(); ++k) {
                auto it = allele_mv[k].find(p);
                if (it != allele_mv[k].end()) bin_batch_entry(it->second, (uint32_t)k, payload);
            }
            const uint32_t n = (uint32_t)payload.size();
            for (int t = 0; t < 4; ++t) out.push_back((char)((n >> (8 * t)) & 0xff));
            out += payload;
        } else {
            for (auto const &m : allele_mv) {
                auto it = m.find(p);
                format_pileup_token(it == m.end() ? nullptr : &it->second, out);
            }
            out += "\n";
        }
        if (i == (j + 1) * window && j + 1 < (size_t)thread) ++j;
        fpv[j]->write(out);
    }
    for (auto fp : fpv) {
        if (!fp->close()) std::cerr << "warning: file cannot be closed" << std::endl;
        delete fp;
    }
}

// ---- phase 2: temp-batch text -> tiles -> libbvc -> CVG/VCF (successor of bt_s + bt_f) ------------------------
// BVC_HOST_PROFILE=1: where a phase-2 thread spends its time (seconds), printed when the thread ends
struct StageClock {
    double read = 0, parse = 0, pack = 0, gpu = 0, cvg = 0, vcf = 0, write = 0;
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
};

// One tile of positions on its way through a runner: parsed (stage 1), handed to libbvc (stage 2), written out (stage 3).
struct Tile {
    // the tile's positions: slots [0, n_used) of `sites` (the slots and their vectors are reused from tile to tile)
    std::vector<SiteColumn> sites;
    size_t n_used = 0;
    size_t entries = 0;                  // observations held by the slots in use
    std::vector<int8_t> refs;
    SiteColumn &slot()
    {
        if (n_used == sites.size()) sites.emplace_back();
        return sites[n_used];
    }
    // buffers of the library call and its records: live as long as the tile, refilled, not reallocated
    std::vector<bvc_site_result> res;
    std::vector<bvc_group_result> gres;
    std::vector<int64_t> offsets;
    std::vector<int8_t> bases, quals;
    std::vector<uint8_t> packed;
    // -- a tile whose text the DEVICE parses (bvc_pileup_begin / bvc_pileup_finish): the inflated lines of its positions, batch
    //    after batch, as they came out of the temp files; what comes back are the columns and tallies the CPU parser would have built
    bool dev = false;
    size_t n_pos = 0;                    // positions of the tile (with or without entries)
    std::vector<int32_t> pos;            // their coordinates
    std::vector<char> text;
    std::vector<uint32_t> line_start;    // [n_batches][n_pos + 1]
    std::vector<int64_t> entry_off;
    std::vector<int64_t> called_off;     // where the entries of a CALLED position are in ent / samples (empty: entry_off says, every position's are there)
    std::vector<int32_t> tally, samples;
    std::vector<bvc_pileup_entry> ent;
    std::vector<bvc_pileup_indel> indels;
    bool dev_parsed = false;             // false after the call: a line was not regular and the tile went through the CPU parser (sites)
    void reset() { n_used = 0; entries = 0; refs.clear(); dev = false; dev_parsed = false; n_pos = 0; pos.clear(); }
};
static_assert(sizeof(bvc_pileup_entry) == sizeof(Entry), "the device parser's entry is the host's");

// A runner's three stages on three threads with three tiles in flight: while libbvc works on tile i the thread that
// reads the temp batches parses tile i + 1 and the writer formats and compresses the lines of tile i - 1 (the
// reference does all of it position by position on one thread, src/BaseVarC.cpp:403-454, 548-666).  Tiles pass
// through in order, so the outputs are those of the one-thread loop byte for byte.
class TileQueue {
 public:
    void push(Tile *t) { { std::lock_guard<std::mutex> g(mu_); q_.push_back(t); } cv_.notify_one(); }
    // nullptr = the queue was closed and is empty
    Tile *pop()
    {
        std::unique_lock<std::mutex> g(mu_);
        cv_.wait(g, [&] { return !q_.empty() || closed_; });
        if (q_.empty()) return nullptr;
        Tile *t = q_.front();
        q_.pop_front();
        return t;
    }
    void close() { { std::lock_guard<std::mutex> g(mu_); closed_ = true; } cv_.notify_all(); }
 private:
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Tile *> q_;
    bool closed_ = false;
};

static double cpus_per_loop();

struct TileRunner {
    StageClock clk;                      // stage 1 (the caller's thread): read, parse
    StageClock clk_dev, clk_out;         // stage 2: pack, gpu; stage 3: cvg, vcf, write
    bvc_ctx *ctx = nullptr;
    const Groups *groups = nullptr;
    std::string chr;
    int32_t n_samples = 0, ithread = 0;
    double min_af = 0;
    BgzfWriter *fvcf = nullptr, *fcvg = nullptr;

    static const int kTiles = 3;
    Tile tiles[kTiles];
    Tile *cur = nullptr;                 // the tile stage 1 is filling
    TileQueue free_q, dev_q, out_q;
    std::thread dev_thread, out_thread;
    std::mutex err_mu;
    std::string err;                     // first failure of stage 2 or 3
    bool started = false;
    int64_t tiles_one_byte = 0, tiles_two_byte = 0;   // library calls by tile form (stage 2's thread; read after finish())
    int64_t tiles_dev_parsed = 0, tiles_cpu_parsed = 0;   // tiles of text: parsed on the device / handed back (a line was not regular)
    std::vector<int32_t> sample0, n_in_batch;         // per temp batch: its first sample and its samples (device-parsed tiles)
    uint8_t carry[5] = {0, 0, 0, 0, 0};               // the parser's long-lived AlleleInfo between tiles (stage 2's thread)
    std::atomic<int64_t> sites_done{0};

    void fail(const std::string &what)
    {
        { std::lock_guard<std::mutex> g(err_mu); if (err.empty()) err = what; }
        // let every stage run dry: tiles still flow back to the parser, which sees the error at its next flush
    }
    bool failed() { std::lock_guard<std::mutex> g(err_mu); return !err.empty(); }

    void start()
    {
        for (int i = 0; i < kTiles; ++i) free_q.push(&tiles[i]);
        cur = free_q.pop();
        dev_thread = std::thread([this] {
            for (Tile *t; (t = dev_q.pop()) != nullptr;) {
                if (!failed()) { try { if (t->dev) run_device_text(*t); else run_device(*t); } catch (const std::exception &e) { fail(e.what()); } }
                out_q.push(t);
            }
            out_q.close();
        });
        out_thread = std::thread([this] {
            for (Tile *t; (t = out_q.pop()) != nullptr;) {
                if (!failed()) { try { write_out(*t); } catch (const std::exception &e) { fail(e.what()); } }
                t->reset();
                free_q.push(t);
            }
        });
        started = true;
    }

    SiteColumn &slot() { return cur->slot(); }

    // stage 1 hands its tile on and takes a free one (waits when both later stages are still busy)
    void flush()
    {
        if (cur->n_used == 0 && !(cur->dev && cur->n_pos)) return;
        dev_q.push(cur);
        cur = free_q.pop();
        if (failed()) { std::lock_guard<std::mutex> g(err_mu); throw std::runtime_error(err); }
    }

    // end of the window: drain the pipeline; rethrows the first failure of a later stage
    void finish()
    {
        if (!started) return;
        if (cur && (cur->n_used || (cur->dev && cur->n_pos))) dev_q.push(cur);
        dev_q.close();
        dev_thread.join();
        out_thread.join();
        started = false;
        std::lock_guard<std::mutex> g(err_mu);
        if (!err.empty()) throw std::runtime_error(err);
    }
    ~TileRunner()
    {
        if (started) { dev_q.close(); dev_thread.join(); out_thread.join(); }
    }

    // A tile whose lines are in T.text / T.line_start through the reference's own rules (strtok_r, atoi) on the CPU, position by position,
    // batch by batch; then the ragged call.  (A line of the tile is not what the reference's writer produces.)
    void cpu_parse_tile(Tile &T)
    {
        const double t0 = StageClock::now();
        tiles_cpu_parsed += 1;
        set_parser_carry(carry);
        const size_t nb = sample0.size();
        for (size_t t = 0; t < T.n_pos; ++t) {
            SiteColumn &site = T.slot();
            site.clear();
            site.pos = T.pos[t];
            int32_t j = 0;
            for (size_t b = 0; b < nb; ++b) {
                const uint32_t s0 = T.line_start[b * (T.n_pos + 1) + t], s1 = T.line_start[b * (T.n_pos + 1) + t + 1];
                j += parse_pileup_line(T.text.data() + s0, (size_t)(s1 - s0 - 1), j, site);
            }
            if (!site.aiv.empty()) { T.refs[T.n_used] = T.refs[t]; ++T.n_used; }
        }
        T.refs.resize(T.n_used);
        get_parser_carry(carry);
        clk_dev.pack += StageClock::now() - t0;
        if (T.n_used) run_device(T);
    }

    // bvc_pileup_finish into the tile's arrays (after a begin that returned BVC_OK).  text_on_device: the indel tokens' text comes back
    // in T.text (the tile's own text never was on the host).
    void finish_tile(Tile &T, int64_t n_ent, int64_t n_ind, int64_t ind_bytes, bool text_on_device)
    {
        const int ng = groups ? (int)groups->names.size() : 0;
        T.entry_off.resize(T.n_pos + 1);
        T.tally.resize(T.n_pos * 32);
        T.ent.resize((size_t)n_ent + 1);
        T.samples.resize((size_t)n_ent + 1);
        T.indels.resize((size_t)n_ind + 1);
        T.res.resize(T.n_pos);
        T.gres.resize(T.n_pos * (size_t)ng);
        if (text_on_device) T.text.resize((size_t)ind_bytes + 1);
        uint8_t carry_out[5];
        // the entries come back for the called positions only (WriteVcf is their one reader, src/BaseVarC.cpp:664; BVC_HOST_CALLED_ONLY=0:
        // for every position, as up to round 5's first form of this feed)
        static const bool called_only = !(getenv("BVC_HOST_CALLED_ONLY") && atoi(getenv("BVC_HOST_CALLED_ONLY")) == 0);
        int rc;
        if (called_only) {
            T.called_off.resize(T.n_pos + 1);
            rc = bvc_pileup_finish_called(ctx, T.refs.data(), min_af, carry, carry_out, ng ? groups->of_sample.data() : nullptr,
                                          ng ? (int64_t)groups->of_sample.size() : 0, ng, T.entry_off.data(), T.tally.data(), T.called_off.data(),
                                          n_ent, T.ent.data(), T.samples.data(), T.indels.data(), text_on_device ? T.text.data() : nullptr,
                                          T.res.data(), ng ? T.gres.data() : nullptr);
        } else {
            T.called_off.clear();
            rc = bvc_pileup_finish(ctx, T.refs.data(), min_af, carry, carry_out, ng ? groups->of_sample.data() : nullptr,
                                   ng ? (int64_t)groups->of_sample.size() : 0, ng, T.entry_off.data(), T.tally.data(), T.ent.data(),
                                   T.samples.data(), T.indels.data(), text_on_device ? T.text.data() : nullptr, T.res.data(),
                                   ng ? T.gres.data() : nullptr);
        }
        if (rc != BVC_OK) throw std::runtime_error(std::string("libbvc: ") + bvc_last_error(ctx));
        std::memcpy(carry, carry_out, 5);
        T.indels.resize((size_t)n_ind);
        std::sort(T.indels.begin(), T.indels.end(), [](const bvc_pileup_indel &a, const bvc_pileup_indel &b) { return a.entry < b.entry; });
        T.dev_parsed = true;
        tiles_dev_parsed += 1;
    }

    // stage 2 of a tile of TEXT: parse and LRT on the device; the CPU parser only when a line is not what the writer produces
    void run_device_text(Tile &T)
    {
        const double t0 = StageClock::now();
        int64_t n_ent = 0, n_ind = 0;
        const int rc = bvc_pileup_begin(ctx, T.text.data(), (int64_t)T.text.size(), T.line_start.data(), sample0.data(), n_in_batch.data(),
                                        (int32_t)sample0.size(), (int32_t)T.n_pos, &n_ent, &n_ind);
        if (rc == BVC_PILEUP_IRREGULAR) { cpu_parse_tile(T); return; }
        if (rc != BVC_OK) throw std::runtime_error(std::string("libbvc: ") + bvc_last_error(ctx));
        finish_tile(T, n_ent, n_ind, 0, false);
        clk_dev.gpu += StageClock::now() - t0;
    }

    // stage 2: the tile in the form libbvc takes, and the call
    void run_device(Tile &T)
    {
        const int64_t ns = (int64_t)T.n_used;
        std::vector<SiteColumn> &sites = T.sites;
        T.res.resize((size_t)ns);
        const int ng = groups ? (int)groups->names.size() : 0;
        int rc;
        double t0 = StageClock::now(), t1;
        static const bool two_byte_only = getenv("BVC_HOST_TWO_BYTE_TILES") != nullptr;
        if (ng == 0) {
            // ragged form: exactly the vectors bt_f builds (src/BaseVarC.cpp:550-559) -- as ONE byte per observation
            // (base << 6 | qual, bvc_lrt_csr_packed: half the bytes over the host link) while every base quality of the
            // tile is below 63, as the two vectors otherwise
            T.offsets.assign(1, 0);
            T.packed.clear();
            bool fits = !two_byte_only;
            for (int64_t i = 0; i < ns && fits; ++i) {
                for (auto const &a : sites[(size_t)i].aiv)
                    if (a.is_indel == 0) {
                        if (a.qual > 62u) { fits = false; break; }
                        T.packed.push_back(a.base > 3u ? (uint8_t)0xFF : (uint8_t)((a.base << 6) | a.qual));
                    }
                T.offsets.push_back((int64_t)T.packed.size());
            }
            static const int8_t none = 0;
            (fits ? tiles_one_byte : tiles_two_byte) += 1;
            if (fits) {
                t1 = StageClock::now(); clk_dev.pack += t1 - t0; t0 = t1;
                rc = bvc_lrt_csr_packed(ctx, ns, T.offsets.data(), T.packed.empty() ? reinterpret_cast<const uint8_t *>(&none) : T.packed.data(),
                                        T.refs.data(), min_af, T.res.data(), BVC_PTR_HOST);
            } else {
                T.offsets.assign(1, 0);
                T.bases.clear(); T.quals.clear();
                for (int64_t i = 0; i < ns; ++i) {
                    for (auto const &a : sites[(size_t)i].aiv)
                        if (a.is_indel == 0) { T.bases.push_back((int8_t)a.base); T.quals.push_back((int8_t)a.qual); }
                    T.offsets.push_back((int64_t)T.bases.size());
                }
                t1 = StageClock::now(); clk_dev.pack += t1 - t0; t0 = t1;
                rc = bvc_lrt_csr(ctx, ns, T.offsets.data(), T.bases.empty() ? &none : T.bases.data(),
                                 T.quals.empty() ? &none : T.quals.data(), T.refs.data(), min_af, T.res.data(), BVC_PTR_HOST);
            }
        } else {
            // dense [site][column] tile; columns are the samples ordered by group (Groups::order_columns), the group of
            // each column is shared by all sites.  One byte per sample (base << 6 | qual, 0xFF = no observation:
            // bvc_lrt_dense_groups_packed) as long as every base quality of the tile is below 63; otherwise the
            // two-byte tile (-1 / 0 for "no observation").  BVC_HOST_TWO_BYTE_TILES=1 forces the latter.
            const int64_t stride = ((int64_t)n_samples + 127) / 128 * 128;
            bool fits = !two_byte_only;
            if (fits) {
                T.packed.assign((size_t)(ns * stride), (uint8_t)0xFF);
                for (int64_t s = 0; s < ns && fits; ++s)
                    for (size_t k = 0; k < sites[s].aiv.size(); ++k) {
                        const Entry &a = sites[s].aiv[k];
                        if (a.is_indel == 0) {
                            if (a.base > 3u) continue;             // not A/C/G/T: no observation, as in the two-byte tile
                            if (a.qual > 62u) { fits = false; break; }
                            T.packed[(size_t)(s * stride + groups->column_of[(size_t)sites[s].sample[k]])] = (uint8_t)((a.base << 6) | a.qual);
                        }
                    }
            }
            T.gres.resize((size_t)(ns * ng));
            (fits ? tiles_one_byte : tiles_two_byte) += 1;
            if (fits) {
                t1 = StageClock::now(); clk_dev.pack += t1 - t0; t0 = t1;
                rc = bvc_lrt_dense_groups_packed(ctx, ns, n_samples, stride, T.packed.data(), T.refs.data(), min_af,
                                                 groups->of_column.data(), ng, T.res.data(), T.gres.data(), BVC_PTR_HOST);
            } else {
                T.bases.assign((size_t)(ns * stride), (int8_t)-1);
                T.quals.assign((size_t)(ns * stride), (int8_t)0);
                for (int64_t s = 0; s < ns; ++s)
                    for (size_t k = 0; k < sites[s].aiv.size(); ++k) {
                        const Entry &a = sites[s].aiv[k];
                        if (a.is_indel == 0) {
                            const int64_t col = groups->column_of[(size_t)sites[s].sample[k]];
                            T.bases[(size_t)(s * stride + col)] = (int8_t)a.base;
                            T.quals[(size_t)(s * stride + col)] = (int8_t)a.qual;
                        }
                    }
                t1 = StageClock::now(); clk_dev.pack += t1 - t0; t0 = t1;
                rc = bvc_lrt_dense_groups(ctx, ns, n_samples, stride, T.bases.data(), T.quals.data(), T.refs.data(), min_af,
                                          groups->of_column.data(), ng, T.res.data(), T.gres.data(), BVC_PTR_HOST);
            }
        }
        if (rc != BVC_OK) throw std::runtime_error(std::string("libbvc: ") + bvc_last_error(ctx));
        clk_dev.gpu += StageClock::now() - t0;
    }

    // stage 3: the CVG line of every position, the VCF line of every called one (bt_f, src/BaseVarC.cpp:548-666)
    void write_out(Tile &T)
    {
        const int ng = groups ? (int)groups->names.size() : 0;
        StageClock &c = clk_out;
        double t0 = StageClock::now(), t1;
        if (T.dev && T.dev_parsed) {
            // the position's slice of the arrays the device returned; its tallies (src/BaseVarC.cpp:560-590) from the 32 counters
            size_t ii = 0;                                   // next indel record (sorted by entry)
            std::vector<std::string> ind_text;
            // the VCF lines of the tile's called positions -- one field per SAMPLE each, ~1 ms to format at 1e5 samples -- ahead of the
            // loop, on this thread and two more where the CPUs allow (vcf_line reads the position's entries and its record only)
            std::vector<size_t> called_t;
            for (size_t t = 0; t < T.n_pos; ++t) if (T.res[t].called && T.entry_off[t + 1] > T.entry_off[t]) called_t.push_back(t);
            std::vector<std::string> vcf_pre(called_t.size());
            auto format_share = [&](size_t k0, size_t k1) {
                for (size_t k = k0; k < k1; ++k) {
                    const size_t t = called_t[k];
                    const int64_t e0 = T.entry_off[t], e1 = T.entry_off[t + 1];
                    const int64_t a0 = T.called_off.empty() ? e0 : T.called_off[t];
                    SiteView v;
                    v.pos = T.pos[t];
                    v.aiv = reinterpret_cast<const Entry *>(&T.ent[(size_t)a0]);
                    v.sample = &T.samples[(size_t)a0];
                    v.n = (size_t)(e1 - e0);
                    std::map<std::string, std::string> info;
                    if (ng) group_af_info(T.res[t], &T.gres[t * (size_t)ng], *groups, info);
                    vcf_pre[k] = vcf_line(T.res[t], chr, T.refs[t], v, info, n_samples);
                }
            };
            {
                const size_t nc = called_t.size();
                const size_t shares = (nc >= 3 && cpus_per_loop() >= 4) ? 3 : 1;
                std::vector<std::future<void>> helpers;
                for (size_t h = 1; h < shares; ++h)
                    helpers.push_back(std::async(std::launch::async, format_share, nc * h / shares, nc * (h + 1) / shares));
                format_share(0, nc / shares);
                for (auto &f : helpers) f.get();
                t1 = StageClock::now(); c.vcf += t1 - t0; t0 = t1;
            }
            size_t next_called = 0;
            for (size_t t = 0; t < T.n_pos; ++t) {
                const int64_t e0 = T.entry_off[t], e1 = T.entry_off[t + 1];
                if (e1 == e0) continue;                      // no entry: the reference skips the position (:443)
                const int32_t *ta = &T.tally[t * 32];
                int32_t cnt[4], fwd[8], rev[8];
                for (int b = 0; b < 8; ++b) { rev[b] = ta[b] + ta[16 + b]; fwd[b] = ta[8 + b] + ta[24 + b]; }
                for (int b = 0; b < 4; ++b) cnt[b] = ta[b] + ta[8 + b];
                ind_text.clear();
                for (; ii < T.indels.size() && T.indels[ii].entry < e1; ++ii)
                    ind_text.emplace_back(T.text.data() + T.indels[ii].text_off, (size_t)T.indels[ii].len);
                SiteView v;
                v.pos = T.pos[t];
                // (a position that is not called: the CVG line reads the tallies and the indel strings only)
                const int64_t a0 = T.called_off.empty() ? e0 : T.called_off[t];
                v.aiv = reinterpret_cast<const Entry *>(&T.ent[(size_t)a0]);
                v.sample = &T.samples[(size_t)a0];
                v.n = (size_t)(e1 - e0);
                v.cnt = cnt; v.fwd = fwd; v.rev = rev;
                v.indels = ind_text.data(); v.n_indels = ind_text.size();
                const bvc_group_result *g = ng ? &T.gres[t * (size_t)ng] : nullptr;
                const std::string cl = cvg_line(chr, T.refs[t], v, g, ng);
                t1 = StageClock::now(); c.cvg += t1 - t0; t0 = t1;
                fcvg->write(cl);
                t1 = StageClock::now(); c.write += t1 - t0; t0 = t1;
                if (T.res[t].called) {
                    fvcf->write(vcf_pre[next_called++]);
                    t1 = StageClock::now(); c.write += t1 - t0; t0 = t1;
                }
                const int64_t done = ++sites_done;
                if (!(done % 1000)) std::cerr << "basetype completed " << done << " sites -- thread" << ithread << std::endl;
            }
            return;
        }
        const int64_t ns = (int64_t)T.n_used;
        for (int64_t s = 0; s < ns; ++s) {
            const bvc_group_result *g = ng ? &T.gres[(size_t)(s * ng)] : nullptr;
            const std::string cl = cvg_line(chr, T.sites[s].pos, T.refs[s], T.sites[s], g, ng);
            t1 = StageClock::now(); c.cvg += t1 - t0; t0 = t1;
            fcvg->write(cl);
            t1 = StageClock::now(); c.write += t1 - t0; t0 = t1;
            if (T.res[s].called) {
                std::map<std::string, std::string> info;
                if (ng) group_af_info(T.res[s], g, *groups, info);
                const std::string vl = vcf_line(T.res[s], chr, T.sites[s].pos, T.refs[s], T.sites[s], info, n_samples);
                t1 = StageClock::now(); c.vcf += t1 - t0; t0 = t1;
                fvcf->write(vl);
                t1 = StageClock::now(); c.write += t1 - t0; t0 = t1;
            }
        }
    }
};

// At most BVC_HOST_DEVICE_SLOTS (default 8) threads per device are inside the device's part of a tile (bvc_pileup_begin_bgzf ..
// bvc_pileup_finish) at a time: past eight, the calls of more threads only get in each other's way (profiles/r05_host/README.txt: 4.9e4
// positions/s with 8 threads, 3.7e4 with 16, 2.4e4 with 32); the threads beyond them gather their next blocks and format their lines.
class DeviceSlots {
 public:
    void acquire(int device)
    {
        std::unique_lock<std::mutex> g(mu_);
        if (limit_ < 0) limit_ = getenv("BVC_HOST_DEVICE_SLOTS") ? std::max(1, atoi(getenv("BVC_HOST_DEVICE_SLOTS"))) : 8;
        if ((size_t)device >= used_.size()) used_.resize((size_t)device + 1, 0);
        cv_.wait(g, [&] { return used_[(size_t)device] < limit_; });
        used_[(size_t)device] += 1;
    }
    void release(int device)
    {
        { std::lock_guard<std::mutex> g(mu_); used_[(size_t)device] -= 1; }
        cv_.notify_all();
    }
 private:
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<int> used_;
    int limit_ = -1;
};
static DeviceSlots g_device_slots;

// ---- temp batches as RAW BGZF blocks (the device inflates them: bvc_pileup_begin_bgzf) ------------------------------------------
// One block of a temp-batch file as it is on disk: its deflate payload and the size it inflates to.
// A BGZF block of a temp-batch file as it lies in the file's pages (the file is mapped: nothing is copied before the one copy into the
// tile's page-locked buffer, and no thread reads ahead -- round 5: a read-ahead thread that fread one block at a time into a vector of
// its own delivered 1.2 GB/s and was what a position loop waited for).
struct RawBlock { const unsigned char *payload = nullptr; size_t len = 0; uint32_t isize = 0, crc32 = 0; };

class MappedBlocks {
 public:
    explicit MappedBlocks(const std::vector<std::string> &paths)
    {
        for (auto const &f : paths) {
            File m;
            m.fd = ::open(f.c_str(), O_RDONLY);
            if (m.fd < 0) { release(); throw std::runtime_error("ERROR: can not open " + f); }
            struct stat st;
            if (::fstat(m.fd, &st) != 0) { ::close(m.fd); release(); throw std::runtime_error("ERROR: can not open " + f); }
            m.n = (size_t)st.st_size;
            if (m.n) {
                void *p = ::mmap(nullptr, m.n, PROT_READ, MAP_PRIVATE, m.fd, 0);
                if (p == MAP_FAILED) { ::close(m.fd); release(); throw std::runtime_error("ERROR: can not map " + f); }
                (void)::madvise(p, m.n, MADV_SEQUENTIAL);
                m.p = static_cast<const unsigned char *>(p);
            }
            f_.push_back(m);
        }
    }
    ~MappedBlocks() { release(); }
    MappedBlocks(const MappedBlocks &) = delete;
    MappedBlocks &operator=(const MappedBlocks &) = delete;
    // the next non-empty block of batch b (SAM specification 4.1: 18-byte header with the 'BC' subfield, payload, CRC32, ISIZE);
    // false when the file has ended (throws on a damaged file)
    bool pop(size_t b, RawBlock &out)
    {
        File &m = f_[b];
        for (;;) {
            if (m.at == m.n) return false;
            const unsigned char *h = m.p + m.at;
            if (m.n - m.at < 18 || h[0] != 0x1f || h[1] != 0x8b || h[12] != 'B' || h[13] != 'C') throw std::runtime_error("ERROR: a temp batch is not BGZF");
            const size_t bsize = ((size_t)h[16] | ((size_t)h[17] << 8)) + 1;
            if (bsize < 18 + 8) throw std::runtime_error("ERROR: a temp batch is not BGZF");
            if (m.n - m.at < bsize) throw std::runtime_error("ERROR: truncated temp batch (it ends inside a BGZF block)");
            const unsigned char *t = h + bsize - 8;
            out.payload = h + 18; out.len = bsize - 18 - 8;
            out.crc32 = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
            out.isize = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
            m.at += bsize;
            if (out.isize > 65536) throw std::runtime_error("ERROR: a temp batch is not BGZF (a block of more than 64 KiB)");
            if (out.isize != 0) return true;                      // (empty blocks: the EOF marker)
        }
    }
 private:
    struct File { const unsigned char *p = nullptr; size_t n = 0, at = 0; int fd = -1; };
    void release()
    {
        for (auto &m : f_) {
            if (m.p) ::munmap(const_cast<unsigned char *>(m.p), m.n);
            if (m.fd >= 0) ::close(m.fd);
        }
        f_.clear();
    }
    std::vector<File> f_;
};

// One temp-batch file of a thread, in either form (detected from its first bytes).
struct BatchInput {
    BgzfReader rd;
    bool bin = false;
    uint32_t n_in_batch = 0;                   // binary form: samples of this batch (the text form counts tokens)
    std::string names;                         // first line without its newline: tab-terminated sample names
    std::vector<unsigned char> rec;

    explicit BatchInput(const std::string &path) : rd(path)
    {
        unsigned char head[16];
        const size_t got = rd.read(head, 8);
        if (got == 8 && std::memcmp(head, kBinBatchMagic, 8) == 0) {
            bin = true;
            if (rd.read(head, 8) != 8) throw std::runtime_error("ERROR: truncated temp batch " + path);
            n_in_batch = (uint32_t)head[0] | ((uint32_t)head[1] << 8) | ((uint32_t)head[2] << 16) | ((uint32_t)head[3] << 24);
            const uint32_t l = (uint32_t)head[4] | ((uint32_t)head[5] << 8) | ((uint32_t)head[6] << 16) | ((uint32_t)head[7] << 24);
            names.resize(l);
            if (l && rd.read(&names[0], l) != l) throw std::runtime_error("ERROR: truncated temp batch " + path);
            if (!names.empty() && names.back() == '\n') names.pop_back();
        } else {                                // text form: start over and take the names line
            rd.seek(0);
            rd.getline(names);
        }
    }

    // The next position's entries of this batch; returns the number of samples the batch spans (the `j` advance).
    int32_t next(int32_t j0, SiteColumn &site, std::string &line, StageClock &clk)
    {
        double t0 = StageClock::now();
        if (bin) {
            unsigned char b4[4];
            // a temp batch holds one record per position of the thread's window: running out of records before the window
            // is exhausted means the file was cut short (both forms: the text form below)
            if (rd.read(b4, 4) != 4) throw std::runtime_error("ERROR: truncated temp batch (it ends before the thread's window does)");
            const uint32_t n = (uint32_t)b4[0] | ((uint32_t)b4[1] << 8) | ((uint32_t)b4[2] << 16) | ((uint32_t)b4[3] << 24);
            rec.resize(n);
            if (n && rd.read(rec.data(), n) != n) throw std::runtime_error("ERROR: truncated temp batch record");
            double t1 = StageClock::now();
            clk.read += t1 - t0;
            if (!parse_pileup_bin(rec.data(), n, j0, site)) throw std::runtime_error("ERROR: malformed temp batch record");
            clk.parse += StageClock::now() - t1;
            return (int32_t)n_in_batch;
        }
        const bool got = rd.getline(line);
        if (!got) throw std::runtime_error("ERROR: truncated temp batch (it ends before the thread's window does)");
        double t1 = StageClock::now();
        clk.read += t1 - t0;
        const int32_t adv = parse_pileup_line(line.data(), line.size(), j0, site);
        clk.parse += StageClock::now() - t1;
        return adv;
    }
};

// CPUs this process may use: the cgroup's quota when it has one (cpu.max: "<quota> <period>"), else the CPUs of its affinity mask.
// A position loop runs helper threads beside its own (the tile formatter, the deflaters of its two outputs, the call that finishes a
// tile while the next one's blocks are gathered); how many it starts goes by the CPUs per loop -- past the quota they only get the
// whole process throttled (round 5, 16 CPUs: three VCF deflaters per loop +50 % with one loop, +18 % with four, -15 % with eight).
static double cpus_per_loop()
{
    static const double cpus = [] {
        double c = (double)std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) c = (double)CPU_COUNT(&set);
        std::ifstream f("/sys/fs/cgroup/cpu.max");
        std::string q; double period = 0;
        if (f >> q >> period && q != "max" && period > 0) c = std::min(c, std::max(1.0, atof(q.c_str()) / period));
        return c;
    }();
    return cpus / (double)std::max(1, opt::thread);
}

static void bt_s(const std::vector<std::string> &ftmp_v, const std::vector<int32_t> &pv, const std::string &refseq,
                 const std::string &chr, int32_t rg_s, int32_t N, int thread, int ithread, int device)
{
    const double t_start = StageClock::now();
    // the outputs (a VCF line carries a field per SAMPLE: a megabyte at 1e5 samples) are deflated by threads of their
    // writers' own
    // (the VCF text is one field per SAMPLE and called position: up to three threads deflate it, BVC_HOST_VCF_DEFLATERS)
    static const int vcf_deflaters = getenv("BVC_HOST_VCF_DEFLATERS") ? std::max(1, atoi(getenv("BVC_HOST_VCF_DEFLATERS")))
                                                                          : (cpus_per_loop() >= 4 ? 3 : (cpus_per_loop() >= 3 ? 2 : 1));
    BgzfWriter fpv(opt::output + "." + std::to_string(ithread) + ".vcf.gz", 6, true, vcf_deflaters);
    BgzfWriter fpc(opt::output + "." + std::to_string(ithread) + ".cvg.gz", 6, true);
    // the blocks of the temp batches are inflated ahead of the position loop by threads of their own (the loop takes a
    // line of every batch per position; inflating was 70 % of it with the text form): BVC_HOST_INFLATE_THREADS, 0 = none
    const int n_inflate = getenv("BVC_HOST_INFLATE_THREADS") ? atoi(getenv("BVC_HOST_INFLATE_THREADS")) : 2;
    std::unique_ptr<InflatePool> inflate_pool(n_inflate > 0 ? new InflatePool(n_inflate) : nullptr);
    std::vector<BatchInput *> fpiv;
    for (auto const &f : ftmp_v) fpiv.push_back(new BatchInput(f));
    if (inflate_pool) for (auto fp : fpiv) fp->rd.attach(inflate_pool.get());
    // (BVC_HOST_PROFILE: on a GPU box the first ~0.5 s of every worker thread go to process-wide stalls while the HIP
    // runtime, started by bvc_device_count in main, finishes coming up -- whatever the thread does first pays them.)
    const double t_opened = StageClock::now();
    std::string sams, line;
    for (auto fp : fpiv) sams += fp->names;
    if (!sams.empty()) sams.pop_back();                                 // names are tab-terminated
    std::vector<std::string> names;
    { std::istringstream iss(sams); std::string id; while (std::getline(iss, id, '\t')) names.push_back(id); }
    // population groups (src/BaseVarC.cpp:335-369): name-sorted, a sample not listed belongs to no group
    Groups groups;
    if (!opt::group.empty()) {
        std::ifstream ifg(opt::group);
        std::unordered_map<std::string, std::string> popg_m;
        std::string id, grp;
        while (ifg >> id >> grp) popg_m.insert({id, grp});
        std::map<std::string, std::vector<int>> popg_idx;
        for (size_t i = 0; i < names.size(); ++i) {
            auto it = popg_m.find(names[i]);
            if (it != popg_m.end()) popg_idx[it->second].push_back((int)i);
        }
        if (popg_idx.size() > BVC_MAX_GROUPS) throw std::runtime_error("ERROR: more than 32 population groups");
        groups.of_sample.assign((size_t)N, 255);
        for (auto const &kv : popg_idx) {
            for (int i : kv.second) groups.of_sample[(size_t)i] = (uint8_t)groups.names.size();
            groups.names.push_back(kv.first);
        }
        groups.order_columns();
    }
    if (ithread == 0) {
        fpc.write(cvg_header(groups));
        fpv.write(vcf_header(groups, opt::reference, names));
    }
    std::cerr << "begin to load data and run basetype" << std::endl;
    const double t_header = StageClock::now();
    TileRunner tr;
    const double t_create = StageClock::now();
    if (bvc_create(&tr.ctx, device) != BVC_OK) throw std::runtime_error("ERROR: no usable gfx950 device for libbvc");
    const double create_s = StageClock::now() - t_create;
    tr.groups = groups.empty() ? nullptr : &groups;
    tr.chr = chr;
    tr.n_samples = N;
    double min_af = 100.0 / N;                                          // src/BaseVarC.cpp:541-543
    if (min_af > 0.001) min_af = 0.001;
    if (opt::maf < min_af) min_af = opt::maf;
    tr.min_af = min_af;
    tr.fvcf = &fpv;
    tr.fcvg = &fpc;
    tr.start();
    int64_t tile = opt::tile > 0 ? opt::tile : 4096;
    if (!groups.empty()) tile = std::max<int64_t>(1, std::min<int64_t>(tile, ((int64_t)256 << 20) / std::max(1, N)));
    reset_parser_carry();
    // BVC_HOST_QUAL_SHIFT=k (tests only): k is added to every base quality as it is read, so that data whose qualities
    // stop at 41 can exercise the tiles that do not fit one byte per observation (quality >= 63)
    const int qual_shift = getenv("BVC_HOST_QUAL_SHIFT") ? atoi(getenv("BVC_HOST_QUAL_SHIFT")) : 0;
    size_t lo, hi;
    thread_window(pv.size(), thread, ithread, lo, hi);
    int32_t count = 0;
    tr.ithread = ithread;
    // Tiles of TEXT for the device parser (the default with the reference's text batches): BVC_HOST_DEVICE_PARSE=0 keeps the CPU
    // parser (A/B runs, and what the quality-shift test hook and the binary batch forms use)
    bool dev_parse = !qual_shift && !(getenv("BVC_HOST_DEVICE_PARSE") && atoi(getenv("BVC_HOST_DEVICE_PARSE")) == 0);
    for (auto fp : fpiv) if (fp->bin) dev_parse = false;
    const double t_loop = StageClock::now();
    // ... and with the blocks of the temp batches inflated on the device too (the default): BVC_HOST_DEVICE_INFLATE=0 inflates on the CPU
    const bool dev_inflate = dev_parse && !(getenv("BVC_HOST_DEVICE_INFLATE") && atoi(getenv("BVC_HOST_DEVICE_INFLATE")) == 0);
    if (dev_parse) {
        // per batch: its first sample and its samples, from the names line (tab-terminated names, src/BaseVarC.cpp:495, 503)
        int32_t j0 = 0;
        for (auto fp : fpiv) {
            const int32_t n_in = (int32_t)std::count(fp->names.begin(), fp->names.end(), '\t');
            tr.sample0.push_back(j0); tr.n_in_batch.push_back(n_in);
            j0 += n_in;
        }
        get_parser_carry(tr.carry);                                     // (zeros: reset_parser_carry above)
    }
    if (dev_inflate) {
        // The files go to the device as they are: this thread takes the raw blocks out of the mapped files,
        // hands every batch's next blocks to bvc_pileup_begin_bgzf -- as many as its lines are short of the tile's target, counted from what
        // the calls report back -- and the tile is the positions every batch has whole.  Stage 3 (CVG / VCF lines) runs beside it.
        const size_t nb = fpiv.size();
        std::vector<int32_t> skip(nb);
        for (size_t b = 0; b < nb; ++b) skip[b] = (int32_t)fpiv[b]->names.size() + 1;      // the names line and its newline
        for (auto fp : fpiv) delete fp;
        fpiv.clear();
        const double tile_mb = getenv("BVC_HOST_TILE_MB") ? std::max(1, atoi(getenv("BVC_HOST_TILE_MB"))) : 128;
        const double blocks_per_batch = std::max(1.0, tile_mb * 1048576.0 / (65280.0 * (double)std::max<size_t>(1, nb)));
        MappedBlocks feed(ftmp_v);
        static const bool check_crc = getenv("BVC_HOST_NO_CRC") == nullptr;     // (the CRC32 of every block is compared on the device, as htslib does)
        std::vector<double> lines_per_block(nb, 0.0);                   // running estimate per batch
        std::vector<int64_t> blocks_sent(nb, 0), lines_seen(nb, 0);
        std::vector<int32_t> left_lines(nb, 0), lines(nb, 0), send(nb, 0);
        std::vector<char> ended(nb, 0);
        // the tile's compressed blocks, one after the other, in page-locked memory of the library's: they go over the link from here
        struct PinnedBytes {
            unsigned char *p = nullptr;
            size_t n = 0, cap = 0;
            ~PinnedBytes() { bvc_host_free(p); }
            size_t size() const { return n; }
            bool empty() const { return n == 0; }
            unsigned char *data() { return p; }
            void clear() { n = 0; }
            void append(const unsigned char *src, size_t len)
            {
                const size_t need = ((n + len + 3) & ~(size_t)3) + 16;
                if (need > cap) {
                    const size_t want = std::max(need + need / 2, (size_t)1 << 20);
                    unsigned char *q = static_cast<unsigned char *>(bvc_host_alloc(want));
                    if (!q) throw std::runtime_error("ERROR: page-locked host memory is not to be had (bvc_host_alloc)");
                    if (n) std::memcpy(q, p, n);
                    bvc_host_free(p);
                    p = q; cap = want;
                }
                std::memcpy(p + n, src, len);
                n += len;
                while (n & 3) p[n++] = 0;                                // every payload from a 4-byte boundary
            }
        };
        // two sets of them: the blocks of the tile after this one are gathered while bvc_pileup_finish works on this one
        struct Staged {
            PinnedBytes comp;
            std::vector<bvc_bgzf_block> blocks;
            std::vector<int32_t> send;
            bool any_new = false, ready = false;
            double seconds = 0;
        } stg[2];
        stg[0].send.assign(nb, 0); stg[1].send.assign(nb, 0);
        int cur = 0;
        bool first = true;
        int64_t target = 1;                                             // positions the next tile should hold
        // BVC_HOST_PROFILE=2: what every tile cost this thread (positions, compressed bytes, gathering its blocks, waiting for a device
        // slot, bvc_pileup_begin_bgzf, bvc_pileup_finish)
        const bool tile_log = getenv("BVC_HOST_PROFILE") && atoi(getenv("BVC_HOST_PROFILE")) >= 2;
        static const bool overlap_gather = getenv("BVC_HOST_GATHER_AHEAD") ? atoi(getenv("BVC_HOST_GATHER_AHEAD")) != 0 : cpus_per_loop() >= 3;
        std::vector<std::array<double, 6>> tile_times;
        // every batch's new blocks: enough for `target` lines going by its lines per block so far (the first call: the names line and
        // one block of positions); a batch found without a whole line gets one block more than that
        auto gather = [&](Staged &S) {
            const double t0 = StageClock::now();
            S.comp.clear(); S.blocks.clear();
            S.any_new = false;
            for (size_t b = 0; b < nb; ++b) {
                int64_t want = 0;
                if (first) { want = 1 + skip[b] / 60000; }
                else if (left_lines[b] < target) {
                    const double lpb = lines_per_block[b] > 0 ? lines_per_block[b] : 1.0;
                    want = (int64_t)std::ceil((double)(target - left_lines[b]) / lpb);
                    if (want < 1) want = 1;
                }
                int32_t took = 0;
                RawBlock rb;
                while (took < want && !ended[b]) {
                    if (!feed.pop(b, rb)) { ended[b] = 1; break; }
                    bvc_bgzf_block blk;
                    blk.comp_off = (int64_t)S.comp.size(); blk.out_off = 0; blk.comp_len = (int32_t)rb.len; blk.isize = (int32_t)rb.isize;
                    blk.crc32 = rb.crc32; blk.check_crc = check_crc ? 1u : 0u;
                    S.comp.append(rb.payload, rb.len);
                    S.blocks.push_back(blk);
                    ++took;
                }
                S.send[b] = took;
                blocks_sent[b] += took;
                S.any_new = S.any_new || took > 0;
            }
            S.ready = true;
            S.seconds = StageClock::now() - t0;
            tr.clk.read += S.seconds;
        };
        for (size_t ip = lo; ip < hi;) {
            Staged &S = stg[cur];
            if (!S.ready) gather(S);
            const double t_wait = StageClock::now();
            g_device_slots.acquire(device);
            struct SlotGuard { int d; ~SlotGuard() { g_device_slots.release(d); } } slot_guard{device};
            const double t1 = StageClock::now();
            Tile &tl = *tr.cur;
            const int32_t max_pos = (int32_t)std::min<int64_t>((int64_t)(hi - ip), std::max<int64_t>(2 * target, 64));
            int32_t T = 0;
            int64_t n_ent = 0, n_ind = 0, ind_bytes = 0;
            const int rc = bvc_pileup_begin_bgzf(tr.ctx, S.comp.empty() ? nullptr : S.comp.data(), (int64_t)S.comp.size(), S.blocks.data(), S.send.data(),
                                                 first ? skip.data() : nullptr, tr.sample0.data(), tr.n_in_batch.data(), (int32_t)nb, max_pos,
                                                 first ? 1 : 0, &T, lines.data(), &n_ent, &n_ind, &ind_bytes);
            first = false;
            S.ready = false;
            const double t_begun = StageClock::now();
            if (rc < 0) throw std::runtime_error(std::string("libbvc: ") + bvc_last_error(tr.ctx) + " (BVC_HOST_DEVICE_INFLATE=0 inflates on the CPU)");
            for (size_t b = 0; b < nb; ++b) {
                lines_seen[b] += lines[b] - left_lines[b];
                if (blocks_sent[b] > 0) lines_per_block[b] = (double)lines_seen[b] / (double)blocks_sent[b];
                left_lines[b] = lines[b] - T;
            }
            if (T == 0) {
                // some batch has no whole line yet: it gets more blocks next time round (left_lines < target); nothing new and nothing
                // whole means its file has ended before the window has
                if (!S.any_new) throw std::runtime_error("ERROR: truncated temp batch (it ends before the thread's window does)");
                continue;
            }
            // the tile after this one: the positions that ~tile_mb of text hold, going by the batch with the fewest lines per block
            {
                double lpb_min = 1e30;
                for (size_t b = 0; b < nb; ++b) if (lines_per_block[b] > 0) lpb_min = std::min(lpb_min, lines_per_block[b]);
                if (lpb_min < 1e30) target = std::max<int64_t>(1, std::min<int64_t>(tile, (int64_t)(blocks_per_batch * lpb_min)));
            }
            tl.dev = true; tl.n_pos = (size_t)T;
            tl.refs.resize((size_t)T);
            tl.pos.resize((size_t)T);
            for (int32_t k = 0; k < T; ++k) {
                const int32_t p = pv[ip + (size_t)k];
                const char rcc = refseq[(size_t)(p - rg_s)];
                tl.pos[(size_t)k] = p;
                tl.refs[(size_t)k] = rcc == 'A' ? 0 : rcc == 'C' ? 1 : rcc == 'G' ? 2 : rcc == 'T' ? 3 : -1;
            }
            if (rc == BVC_PILEUP_IRREGULAR) {
                int64_t need = 0;
                if (bvc_pileup_text(tr.ctx, nullptr, 0, &need, nullptr) != BVC_OK) throw std::runtime_error(std::string("libbvc: ") + bvc_last_error(tr.ctx));
                tl.text.resize((size_t)need + 1);
                tl.line_start.resize(nb * ((size_t)T + 1));
                if (bvc_pileup_text(tr.ctx, tl.text.data(), need, &need, tl.line_start.data()) != BVC_OK)
                    throw std::runtime_error(std::string("libbvc: ") + bvc_last_error(tr.ctx));
                tr.cpu_parse_tile(tl);
            } else if (overlap_gather && ip + (size_t)T < hi) {
                // the blocks of the next tile are gathered (out of the mapped files into the other page-locked buffer) beside the call
                // that parses this one, runs its LRT and brings its records back: what the next call needs is known since the begin
                std::future<void> fin = std::async(std::launch::async, [&] { tr.finish_tile(tl, n_ent, n_ind, ind_bytes, true); });
                gather(stg[1 - cur]);                                   // (should it throw, fin's destructor waits for the call)
                fin.get();
                cur = 1 - cur;
            } else {
                tr.finish_tile(tl, n_ent, n_ind, ind_bytes, true);
            }
            tr.clk_dev.gpu += StageClock::now() - t1;
            if (tile_log) tile_times.push_back({(double)T, (double)S.comp.size(), S.seconds, t1 - t_wait, t_begun - t1, StageClock::now() - t_begun});
            ip += (size_t)T;
            // straight to stage 3 (this thread did stage 2's work itself)
            if (tr.failed()) { std::lock_guard<std::mutex> g(tr.err_mu); throw std::runtime_error(tr.err); }
            tr.out_q.push(tr.cur);
            tr.cur = tr.free_q.pop();
        }
        if (tile_log) {
            std::ostringstream os;
            os << "[profile] thread " << ithread << " tiles (positions, comp MB, gather ms, slot wait ms, begin ms, finish ms):";
            for (auto const &t : tile_times)
                os << " (" << t[0] << ", " << std::fixed << std::setprecision(1) << t[1] / 1048576.0 << ", " << t[2] * 1e3 << ", " << t[3] * 1e3 << ", "
                   << t[4] * 1e3 << ", " << t[5] * 1e3 << ")";
            std::cerr << os.str() << std::endl;
        }
    } else if (dev_parse) {
        // a tile: --tile positions at most, and about BVC_HOST_TILE_MB of text (default 32) going by the tile before it
        const double target = 1048576.0 * (getenv("BVC_HOST_TILE_MB") ? std::max(1, atoi(getenv("BVC_HOST_TILE_MB"))) : 32);
        double bytes_per_pos = 0;
        const size_t nb = fpiv.size();
        for (size_t ip = lo; ip < hi;) {
            size_t T = (size_t)std::min<int64_t>(tile, (int64_t)(hi - ip));
            if (bytes_per_pos > 0) T = std::min(T, (size_t)std::max(1.0, target / bytes_per_pos));
            else T = std::min<size_t>(T, 64);
            Tile &tl = *tr.cur;
            tl.dev = true; tl.n_pos = T;
            tl.text.clear();
            tl.line_start.resize(nb * (T + 1));
            double t0 = StageClock::now();
            for (size_t b = 0; b < nb; ++b) {
                tl.text.resize((tl.text.size() + 15) & ~(size_t)15, '\n');    // every batch's lines from a 16-byte boundary
                if (fpiv[b]->rd.read_lines(T, tl.text, &tl.line_start[b * (T + 1)]) != T)
                    throw std::runtime_error("ERROR: truncated temp batch (it ends before the thread's window does)");
                if (tl.text.size() > (size_t)0xF0000000u) throw std::runtime_error("ERROR: more than 3.75 GiB of text in one tile: lower --tile");
            }
            tr.clk.read += StageClock::now() - t0;
            tl.refs.resize(T);
            tl.pos.resize(T);
            for (size_t k = 0; k < T; ++k) {
                const int32_t p = pv[ip + k];
                const char rc = refseq[(size_t)(p - rg_s)];
                tl.pos[k] = p;
                tl.refs[k] = rc == 'A' ? 0 : rc == 'C' ? 1 : rc == 'G' ? 2 : rc == 'T' ? 3 : -1;
            }
            bytes_per_pos = (double)tl.text.size() / (double)T;
            tr.flush();
            ip += T;
        }
    } else
    for (size_t ip = lo; ip < hi; ++ip) {
        const int32_t p = pv[ip];
        SiteColumn &site = tr.slot();                                   // parsed in place: no copy into the tile
        site.clear();
        site.pos = p;
        int32_t j = 0;
        for (auto fp : fpiv) j += fp->next(j, site, line, tr.clk);
        if (qual_shift)                                                 // test hook, see above
            for (auto &a : site.aiv)
                if (a.is_indel == 0) a.qual = (uint32_t)std::min(127, (int)a.qual + qual_shift);
        if (!site.aiv.empty()) {
            const char rc = refseq[(size_t)(p - rg_s)];
            const int8_t ref_base = rc == 'A' ? 0 : rc == 'C' ? 1 : rc == 'G' ? 2 : rc == 'T' ? 3 : -1;
            ++tr.cur->n_used;
            tr.cur->entries += site.aiv.size();
            tr.cur->refs.push_back(ref_base);
            // a tile is full at --tile positions or at 1M observations (16 MB of per-sample records held for the
            // CVG/VCF lines; three tiles are in flight per thread): at 1e5 samples that is a hundred positions, still far
            // more than the device needs, and short enough for the three stages to overlap within a thread's window
            if ((int64_t)tr.cur->n_used >= tile || tr.cur->entries >= ((size_t)1 << 20)) tr.flush();
            if (!(++count % 1000)) std::cerr << "basetype completed " << count << " sites -- thread" << ithread << std::endl;
        }
    }
    tr.finish();
    const double loop_s = StageClock::now() - t_loop, setup_s = t_loop - t_start;
    if (getenv("BVC_HOST_PROFILE")) {
        std::cerr << "[profile] thread " << ithread << ": library calls on one-byte tiles " << tr.tiles_one_byte << ", on two-byte tiles "
                  << tr.tiles_two_byte << "; tiles of text parsed on the device " << tr.tiles_dev_parsed << ", handed back to the CPU parser "
                  << tr.tiles_cpu_parsed << std::endl;
        const StageClock &c = tr.clk, &d = tr.clk_dev, &o = tr.clk_out;
        std::cerr << "[profile] thread " << ithread << ": stage 1 read+inflate " << c.read << " s, parse " << c.parse << " s | stage 2 pack "
                  << d.pack << " s, libbvc " << d.gpu << " s | stage 3 cvg lines " << o.cvg << " s, vcf lines " << o.vcf
                  << " s, compress+write " << o.write << " s; setup (open batches " << t_opened - t_start << " s, names + header "
                  << t_header - t_opened << " s, bvc_create " << create_s << " s) " << setup_s << " s, position loop " << loop_s << " s, thread total "
                  << StageClock::now() - t_start << " s" << std::endl;
    }
    bvc_destroy(tr.ctx);
    if (!fpv.close()) std::cerr << "warning: file cannot be closed" << std::endl;
    if (!fpc.close()) std::cerr << "warning: file cannot be closed" << std::endl;
    for (auto fp : fpiv) delete fp;
    if (!opt::keep_tmp)
        for (auto const &f : ftmp_v) std::remove(f.c_str());
}

template <class F>
static void run_pool(int n_tasks, int n_threads, F fn)
{
    std::atomic<int> next(0);
    std::mutex mu;
    std::string err;
    std::vector<std::thread> ws;
    for (int t = 0; t < std::max(1, std::min(n_threads, n_tasks)); ++t)
        ws.emplace_back([&]() {
            for (int i; (i = next++) < n_tasks;) {
                try { fn(i); } catch (const std::exception &e) { std::lock_guard<std::mutex> g(mu); if (err.empty()) err = e.what(); }
            }
        });
    for (auto &w : ws) w.join();
    if (!err.empty()) throw std::runtime_error(err);
}

static void run_basetype(int argc, char **argv)                          // src/BaseVarC.cpp:176-314
{
    parse_options(argc, argv, BASETYPE_MESSAGE);
    if (opt::reference.empty()) throw std::invalid_argument("reference must be feed");
    time_t tim = time(0);
    const clock_t ctb = clock();
    std::cout << "basetype start -- " << ctime(&tim);
    std::vector<std::string> bams;
    { std::ifstream ibam(opt::input); std::string l; while (std::getline(ibam, l)) bams.push_back(l); }
    const int32_t N = (int32_t)bams.size();
    if (N == 0) throw std::invalid_argument("empty input list");
    std::string chr;
    int32_t rg_s, rg_e;
    const int32_t buf = 1000;
    std::tie(chr, rg_s, rg_e) = splitrg(opt::region);
    std::string refseq, err;
    if (!fetch_reference(opt::reference, chr, rg_s, rg_e + buf, refseq, err)) throw std::runtime_error(err);
    std::vector<int32_t> pv;
    for (size_t i = 0; i + buf < refseq.length(); ++i) {
        const char c = refseq[i];
        if (c == 'A' || c == 'C' || c == 'G' || c == 'T') pv.push_back((int32_t)i + rg_s);
    }
    const int thread = std::max(1, opt::thread);
    for (int i = 0; i < thread; ++i) {
        const std::string d = opt::output + ".tmp.thread." + std::to_string(i);
        if (::mkdir(d.c_str(), 0777) != 0 && !file_exists(d)) throw std::runtime_error("ERROR: fail to run mkdir");
    }
    const int bc = std::max(1, opt::batch);
    const int nb = 1 + (N - 1) / bc;
    std::vector<std::vector<std::string>> ftmp_vv((size_t)thread);
    int ngz = 0, bk = nb - 1;
    for (int j = 0; j < nb; ++j) {
        int k = 0;
        for (int i = 0; i < thread; ++i) {
            const std::string f = tmp_name(i, j);
            ftmp_vv[(size_t)i].push_back(f);
            if (file_exists(f)) {
                if (BgzfReader(f).is_bgzf()) { if (BgzfReader::has_eof_marker(f)) { k += 1; ngz += 1; } }
                else std::cerr << "warning: " << f << " is not bgziped" << std::endl;
            }
        }
        if (k != thread) bk = bk > j ? j : bk;
    }
    int first_batch = 0;
    bool extract = true;
    if (opt::rerun && ngz > 0) { extract = ngz != thread * nb; first_batch = bk; }
    if (extract) {
        std::cerr << "begin to extract reads from bam" << std::endl;
        run_pool(nb - first_batch, thread, [&](int t) { bt_r(bams, pv, refseq, chr, rg_s, rg_e, nb, bc, first_batch + t, thread); });
    }
    time_t tim1 = time(0);
    const double t_loaded = StageClock::now();
    std::cout << "basetype loading done -- " << ctime(&tim1);
    if (opt::load) std::exit(EXIT_SUCCESS);
    const int gpus = bvc_device_count();
    if (gpus <= 0) throw std::runtime_error("ERROR: no gfx950 device (libbvc has no CPU fallback)");
    {
        std::vector<std::thread> workers;
        std::mutex mu;
        std::string werr;
        for (int i = 0; i < thread; ++i)
            workers.emplace_back([&, i]() {
                try { bt_s(ftmp_vv[(size_t)i], pv, refseq, chr, rg_s, N, thread, i, device_of_thread(i, gpus, opt::gpus)); }
                catch (const std::exception &e) { std::lock_guard<std::mutex> g(mu); if (werr.empty()) werr = e.what(); }
            });
        // Every worker is joined BEFORE anything can throw: unwinding past a joinable std::thread is std::terminate,
        // which would lose the error text and leave the other threads in the middle of their device calls.
        for (auto &w : workers) w.join();
        const double t_joined = StageClock::now();
        {
            std::lock_guard<std::mutex> g(mu);
            if (!werr.empty()) {
                for (int i = 0; i < thread; ++i) {                       // partial sub-files are of no use; the temp
                    std::remove((opt::output + "." + std::to_string(i) + ".vcf.gz").c_str());   // batches stay for --rerun
                    std::remove((opt::output + "." + std::to_string(i) + ".cvg.gz").c_str());
                }
                throw std::runtime_error(werr);
            }
        }
        // merge the per-thread sub-files in thread (= position) order (src/BaseVarC.cpp:268-296)
        BgzfWriter fov(opt::output + ".vcf.gz"), foc(opt::output + ".cvg.gz");
        if (!merge_subfiles(opt::output, ".vcf.gz", thread, fov) || !merge_subfiles(opt::output, ".cvg.gz", thread, foc))
            throw std::runtime_error("ERROR: fail to write");
        std::cout << "merge subfiles done" << std::endl;
        if (!fov.close()) std::cerr << "warning: file cannot be closed" << std::endl;
        if (!foc.close()) std::cerr << "warning: file cannot be closed" << std::endl;
        if (getenv("BVC_HOST_PROFILE")) {
            // CPU time the whole process has consumed so far (all threads, the HIP runtime's included): on a box that grants a quota of
            // CPUs this, not the thread count, is what the feed is bounded by
            struct rusage ru;
            getrusage(RUSAGE_SELF, &ru);
            const double cpu = (double)ru.ru_utime.tv_sec + 1e-6 * (double)ru.ru_utime.tv_usec + (double)ru.ru_stime.tv_sec + 1e-6 * (double)ru.ru_stime.tv_usec;
            std::cerr << "[profile] main: compute phase (threads) " << t_joined - t_loaded << " s, merge of sub-files "
                      << StageClock::now() - t_joined << " s; process CPU time " << cpu << " s (user "
                      << (double)ru.ru_utime.tv_sec + 1e-6 * (double)ru.ru_utime.tv_usec << ")" << std::endl;
        }
    }
    for (int i = 0; i < thread; ++i) ::rmdir((opt::output + ".tmp.thread." + std::to_string(i)).c_str());
    time_t tim2 = time(0);
    std::cout << "basetype computing done -- " << ctime(&tim2);
    std::cout << "basetype elapsed cpu secs : " << double(clock() - ctb) / CLOCKS_PER_SEC << std::endl;
    std::cout << "basetype done" << std::endl;
}

int main(int argc, char **argv)
{
    if (argc <= 1) { std::cerr << BASEVARC_USAGE_MESSAGE; return 0; }
    // every thread drives the device through a stream of its own: as many hardware queues as threads that may be inside the device's
    // part of a tile at a time (the runtime's default is four, two streams then share a queue; profiles/r05_host/README.txt)
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    const std::string command(argv[1]);
    try {
        if (command == "basetype") run_basetype(argc - 1, argv + 1);
        else if (command == "popmatrix" || command == "concat") {
            std::cerr << "BaseVarC " << command << " is not part of the MI355X build (only the basetype path is)" << std::endl;
            return 1;
        } else { std::cerr << BASEVARC_USAGE_MESSAGE; return 0; }
    } catch (const std::exception &e) {
        std::cerr << e.what() << std::endl;
        return 1;
    }
    return 0;
}
