// pileup.h -- host-side data structures and text formats either side of the basetype hot path.
//
// Counterparts in the reference (paths under /root/reference):
//   AlleleInfo                      src/BamProcess.h:30-39
//   temp-batch pileup text writer   bt_r, src/BaseVarC.cpp:509-527
//   temp-batch pileup text parser   bt_s, src/BaseVarC.cpp:403-441
//   CVG line                        bt_f, src/BaseVarC.cpp:548-610, 617-663
//   VCF line                        WriteVcf, src/BaseType.cpp:141-234
//   headers                         src/BaseVarC.cpp:65-88, 364-382
#ifndef BVC_HOST_PILEUP_H
#define BVC_HOST_PILEUP_H

#include <cstdint>
#include <map>
#include <algorithm>
#include <string>
#include <memory>
#include <new>
#include <utility>
#include <vector>

#include "../../include/bvc.h"

namespace bvchost {

struct AlleleInfo {                 // src/BamProcess.h:30-39
    uint8_t base = 4;               // 0 A, 1 C, 2 G, 3 T, 4 N, 5 indel marker
    uint8_t mapq = 0;
    uint8_t qual = 0;
    uint8_t rpr = 0;
    uint8_t strand = 0;             // 0 '-', 1 '+'
    uint8_t is_indel = 0;
    std::string indel;
};

// What the position loop keeps of one entry: eight bytes.  (The indel text of an entry, which only the CVG line's
// "Indels" column reads, is kept once per indel entry in SiteColumn::indels -- an AlleleInfo per entry would be 40 bytes
// of which 32 are an empty string for all but a few entries per thousand, and the loops over a position's entries
// are memory-bound at 1e5 samples.)
struct Entry {
    uint8_t base, mapq, qual, rpr, strand, is_indel;
    uint16_t pad;
};

// One position: the entries of the samples that have data, in sample order (aiv), and which sample each
// entry belongs to (the inverse of the reference's `idx` map, src/BaseVarC.cpp:428-429).  The tallies of the CVG line
// (src/BaseVarC.cpp:560-590: depth per base over the non-indel entries, forward / reverse counts per base value over
// ALL entries) are taken as the entries are appended, while each is in registers: cvg_line does not walk them again.
// std::allocator whose value-less construct() default-initialises: resize() of a vector of trivial elements then leaves the new
// elements unwritten instead of zero-filling them.  The parsers size a position's arrays for the most entries its line could hold
// and write the real ones in place; at N = 1e5 samples the zero fill alone was 1.2 MB of stores per position.
template <class T>
struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { typedef NoInitAlloc<U> other; };
    NoInitAlloc() = default;
    template <class U> NoInitAlloc(const NoInitAlloc<U> &) {}
    template <class U> void construct(U *p) { ::new (static_cast<void *>(p)) U; }
    template <class U, class... A> void construct(U *p, A &&... a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
};

struct SiteColumn {
    int32_t pos = 0;
    std::vector<Entry, NoInitAlloc<Entry>> aiv;
    std::vector<int32_t, NoInitAlloc<int32_t>> sample;
    std::vector<std::string> indels;   // indel text of the entries with is_indel = 1, in entry order
    int32_t cnt[4] = {0, 0, 0, 0};
    int32_t fwd[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rev[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    void clear()
    {
        aiv.clear(); sample.clear(); indels.clear();
        for (int i = 0; i < 4; ++i) cnt[i] = 0;
        for (int i = 0; i < 8; ++i) { fwd[i] = 0; rev[i] = 0; }
    }
    void add(const AlleleInfo &a, int32_t j)
    {
        aiv.push_back(Entry{a.base, a.mapq, a.qual, a.rpr, a.strand, a.is_indel, 0});
        sample.push_back(j);
        if (a.is_indel) indels.push_back(a.indel);
        else if (a.base < 4) cnt[a.base] += 1;
        if (a.base < 8u) (a.strand == 1 ? fwd : rev)[a.base] += 1;
    }
};

// What the emitters read of a position, wherever it lives: a SiteColumn the CPU parser filled, or a position's slice of the arrays
// the device parser returned (bvc_pileup_finish).
struct SiteView {
    int32_t pos = 0;
    const Entry *aiv = nullptr;
    const int32_t *sample = nullptr;
    size_t n = 0;                       // entries
    const int32_t *cnt = nullptr, *fwd = nullptr, *rev = nullptr;   // [4], [8], [8]
    const std::string *indels = nullptr;    // indel text of the entries with is_indel = 1, in entry order
    size_t n_indels = 0;
};
inline SiteView view_of(const SiteColumn &c)
{
    SiteView v;
    v.pos = c.pos; v.aiv = c.aiv.data(); v.sample = c.sample.data(); v.n = c.aiv.size();
    v.cnt = c.cnt; v.fwd = c.fwd; v.rev = c.rev; v.indels = c.indels.data(); v.n_indels = c.indels.size();
    return v;
}

// Population groups: name-sorted (std::map order in the reference, src/BaseVarC.cpp:98, 358-362).
struct Groups {
    std::vector<std::string> names;
    std::vector<uint8_t> of_sample;     // group index per sample, 255 = in no group
    // Column order of the dense tiles handed to libbvc: samples ordered by group (ungrouped last), so that every
    // group is a contiguous run of columns and the library's column-range histogram kernel takes the call.
    std::vector<int32_t> column_of;     // sample -> column
    std::vector<uint8_t> of_column;     // group index per column (non-decreasing, 255 last)
    void order_columns()
    {
        const size_t n = of_sample.size();
        std::vector<int32_t> order(n);
        for (size_t i = 0; i < n; ++i) order[i] = (int32_t)i;
        std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return of_sample[(size_t)x] < of_sample[(size_t)y]; });
        column_of.assign(n, 0);
        of_column.assign(n, 255);
        for (size_t c = 0; c < n; ++c) { column_of[(size_t)order[c]] = (int32_t)c; of_column[c] = of_sample[(size_t)order[c]]; }
    }
    bool empty() const { return names.empty(); }
};

// ---- temp-batch pileup text (f1) -----------------------------------------------------------------------
// One line per position; per sample one space-terminated token: "." | "base,mapq,qual,rpr,strand" | indel.
void format_pileup_token(const AlleleInfo *a, std::string &out);           // a == nullptr -> ". "
// Parses the tokens of one batch's line; sample indices start at j0.  Returns the number of tokens seen
// (= samples in that batch).  N bases are dropped, indel tokens kept (src/BaseVarC.cpp:427-436).
int parse_pileup_line(const char *line, size_t len, int32_t j0, SiteColumn &site);
// The parser carries one AlleleInfo across calls within a thread, as bt_s does; call this at thread start.
void reset_parser_carry();
// base, mapq, qual, rpr, strand of that AlleleInfo: tiles parsed on the device and tiles parsed here hand it back and forth
void get_parser_carry(uint8_t out[5]);
void set_parser_carry(const uint8_t in[5]);

// ---- temp-batch binary form (additive: `--tmp-format bin`; the text form above stays the default) ----------------
// Same content as the text form, without the tokenising: a BGZF stream of
//   magic "BVCBAT1\n" | u32 samples in this batch | u32 length + the text form's first line (tab-terminated names)
// and then, per position of the thread's window, one record
//   u32 payload bytes | entries...      entry = u32 sample-in-batch | u8 base mapq qual rpr flags | [u16 n + indel text]
// (flags: bit 0 strand, bit 1 indel).  Only samples WITH data take space: at low coverage the text form spends two
// bytes (". ") on every absent sample and the parser a branch, this form nothing.  Little-endian.
extern const char kBinBatchMagic[8];
void bin_batch_header(uint32_t n_samples_in_batch, const std::string &names_line, std::string &out);
// Appends sample `j`'s entry to a position record under construction (payload only).
void bin_batch_entry(const AlleleInfo &a, uint32_t j, std::string &payload);
// Decodes one position record's payload; sample indices are offset by j0.  Reproduces the text parser's observable
// quirks: N bases dropped, and the one long-lived AlleleInfo whose fields an indel entry inherits
// (src/BaseVarC.cpp:392, 407-440) -- so both forms give identical SiteColumns.  Returns false on a malformed record.
bool parse_pileup_bin(const unsigned char *payload, size_t len, int32_t j0, SiteColumn &site);

// ---- emission (f2) -------------------------------------------------------------------------------------
extern const char *const kCvgHeader;
extern const char *const kVcfHeader;
std::string cvg_header(const Groups &g);
std::string vcf_header(const Groups &g, const std::string &reference, const std::vector<std::string> &sample_names);

// CVG line without the trailing newline handling of groups: pass grp (n_groups records) or nullptr.
std::string cvg_line(const std::string &chr, int8_t ref_base, const SiteView &site, const bvc_group_result *grp, int n_groups);
inline std::string cvg_line(const std::string &chr, int32_t pos, int8_t ref_base, const SiteColumn &site,
                            const bvc_group_result *grp, int n_groups)
{
    SiteView v = view_of(site);
    v.pos = pos;
    return cvg_line(chr, ref_base, v, grp, n_groups);
}
// VCF line for a called site.  info carries the "<group>_AF" entries; the rest is filled here.
std::string vcf_line(const bvc_site_result &bt, const std::string &chr, int8_t ref_base, const SiteView &site,
                     std::map<std::string, std::string> &info, int32_t n_samples);
inline std::string vcf_line(const bvc_site_result &bt, const std::string &chr, int32_t pos, int8_t ref_base,
                            const SiteColumn &site, std::map<std::string, std::string> &info, int32_t n_samples)
{
    SiteView v = view_of(site);
    v.pos = pos;
    return vcf_line(bt, chr, ref_base, v, info, n_samples);
}
// "<group>_AF" values from the group records (src/BaseVarC.cpp:646-658).
void group_af_info(const bvc_site_result &bt, const bvc_group_result *grp, const Groups &g,
                   std::map<std::string, std::string> &info);

std::string fmt_fixed(double v, int prec);          // fmt's {:.Nf}

// ---- orchestration (f3): who works on which positions, on which device, and how the pieces come together again ----------
// Per-thread window of positions (src/BaseVarC.cpp:399-403, 497, 523): thread i takes [lo, hi) of the region's position list.
// The reference's arithmetic (window = psize % T + psize / T) can run a non-last thread past the end when psize % T is
// large; the range is clamped here.  The windows of threads 0..T-1 are disjoint, ascending and cover [0, psize).
void thread_window(size_t psize, int thread, int ithread, size_t &lo, size_t &hi);
// Device of a phase-2 thread (additive: the reference has one CPU path): thread i works on device i mod G, G = the devices present,
// or --gpus when that is given and smaller.  Sites shard by position window, no device ever needs another's data.
int device_of_thread(int ithread, int devices_present, int gpus_option);
// Merge of the per-thread sub-files <out>.<i><suffix>, i = 0..T-1, into `dst` in thread order -- which is position order, since
// the windows ascend with the thread index (src/BaseVarC.cpp:268-296) -- removing each after it is appended.
class BgzfWriter;
bool merge_subfiles(const std::string &out_prefix, const std::string &suffix, int thread, BgzfWriter &dst);

}  // namespace bvchost
#endif
