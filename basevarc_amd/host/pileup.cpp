#include "pileup.h"

#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "bgzf.h"
#include "stats.h"

namespace bvchost {

static const char kBase2Char[6] = {'A', 'C', 'G', 'T', 'N', 'N'};   // src/BaseType.h:24 has four entries; see vcf_line
static const char kStrand[2] = {'-', '+'};                          // src/BaseType.h:23
static const double kMln10To10 = -0.23025850929940458;              // src/BaseType.h:10
static const double kQualThreshold = 60;                            // src/BaseType.h:12

std::string fmt_fixed(double v, int prec)
{
    // fmt prints a NaN as "nan" or "-nan" by its sign bit, which for 0/0 is an accident of code generation;
    // a NaN is printed as "nan" here.
    if (std::isnan(v)) return "nan";
    char buf[64];
    std::snprintf(buf, sizeof buf, "%.*f", prec, v);
    return buf;
}

// ---- temp-batch pileup text ------------------------------------------------------------------------
void format_pileup_token(const AlleleInfo *a, std::string &out)   // src/BaseVarC.cpp:513-520
{
    if (!a) { out += ". "; return; }
    if (a->is_indel == 1) { out += a->indel; out += ' '; return; }
    char buf[48];
    std::snprintf(buf, sizeof buf, "%u,%u,%u,%u,%u ", (unsigned)a->base, (unsigned)a->mapq, (unsigned)a->qual,
                  (unsigned)a->rpr, (unsigned)a->strand);
    out += buf;
}

static inline int atoi_span(const char *&p, const char *end)   // atoi on a field: optional sign, digits
{
    int sign = 1, v = 0;
    if (p < end && (*p == '-' || *p == '+')) { if (*p == '-') sign = -1; ++p; }
    while (p < end && *p >= '0' && *p <= '9') { v = v * 10 + (*p - '0'); ++p; }
    return sign * v;
}

// The reference keeps ONE AlleleInfo alive across tokens, lines and positions (src/BaseVarC.cpp:392, 407-440)
// and an indel token only sets is_indel/indel, so an indel entry carries the base/mapq/qual/rpr/strand of
// the last base token the thread parsed.  Those stale fields reach the strand counts of the CVG line and
// the per-sample column of the VCF line, so the carry is part of the observable behaviour and is kept
// (starting from zeros, where the reference starts from an uninitialised object).
static thread_local AlleleInfo g_carry;

void reset_parser_carry() { g_carry = AlleleInfo(); g_carry.base = 0; }
void get_parser_carry(uint8_t out[5])
{
    out[0] = g_carry.base; out[1] = g_carry.mapq; out[2] = g_carry.qual; out[3] = g_carry.rpr; out[4] = g_carry.strand;
}
void set_parser_carry(const uint8_t in[5])
{
    g_carry.base = in[0]; g_carry.mapq = in[1]; g_carry.qual = in[2]; g_carry.rpr = in[3]; g_carry.strand = in[4];
}

int parse_pileup_line(const char *line, size_t len, int32_t j0, SiteColumn &site)
{
    const char *p = line, *end = line + len;
    int32_t j = j0;
    AlleleInfo &ai = g_carry;
    // As in parse_pileup_bin: room for every entry the line can hold is made once (a token takes at least two bytes), the
    // entries are written in place and the tallies of the base tokens are kept in locals.
    const size_t n0 = site.aiv.size();
    site.aiv.resize(n0 + len / 2 + 1);
    site.sample.resize(n0 + len / 2 + 1);
    Entry *e = site.aiv.data() + n0;
    int32_t *sj = site.sample.data() + n0;
    int32_t tally[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // [strand][base] over the base tokens
    // up to three decimal digits (what the writer prints for an 8-bit field); false = not a digit at *q
    auto num3 = [](const char *&q, unsigned &v) -> bool {
        unsigned d = (unsigned)(unsigned char)*q - '0';
        if (d > 9u) return false;
        v = d; ++q;
        d = (unsigned)(unsigned char)*q - '0';
        if (d <= 9u) {
            v = v * 10u + d; ++q;
            d = (unsigned)(unsigned char)*q - '0';
            if (d <= 9u) { v = v * 10u + d; ++q; }
        }
        return true;
    };
    while (p < end) {
#if defined(__SSE2__)
        if (*p == '.' && end - p >= 32) {
            // "no data" tokens (". ": nine in ten at 10 % coverage) come in runs: 32 bytes -- sixteen samples -- per look, and the
            // length of the run in front of the next real token from the compare mask instead of from a loop over tokens
            const __m128i pat = _mm_set1_epi16(0x202E);                            // '.' (0x2E) then ' ' (0x20), little endian
            const unsigned same = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i *>(p)), pat)) |
                                  ((unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i *>(p + 16)), pat)) << 16);
            if (same == 0xFFFFFFFFu) { p += 32; j += 16; continue; }
            const unsigned skip = (unsigned)__builtin_ctz(~same) & ~1u;            // bytes before the first difference, in whole tokens
            p += skip; j += (int32_t)(skip >> 1);
        }
#endif
        if (*p == '.' && end - p >= 2 && p[1] == ' ') { p += 2; ++j; continue; }   // "no data"
        if (end - p >= 20) {
            // the common data token, "base,mapq,qual,rpr,strand " exactly as format_pileup_token writes it; anything else
            // (a sign, a fourth digit, a missing field, a token at the very end of the line) goes the general way below
            const char *q = p;
            unsigned v0, v1, v2, v3, v4;
            if (num3(q, v0) && *q == ',' && num3(++q, v1) && *q == ',' && num3(++q, v2) && *q == ',' && num3(++q, v3) &&
                *q == ',' && num3(++q, v4) && *q == ' ') {
                ai.is_indel = 0;
                ai.base = (uint8_t)(v0 & 7u);                   // bit-field widths, src/BamProcess.h:32-37
                ai.mapq = (uint8_t)v1; ai.qual = (uint8_t)v2; ai.rpr = (uint8_t)v3;
                ai.strand = (uint8_t)(v4 & 1u);
                if (ai.base != 4) {                             // skip N base, :427
                    *e++ = Entry{ai.base, ai.mapq, ai.qual, ai.rpr, ai.strand, 0, 0};
                    *sj++ = j;
                    tally[ai.strand << 3 | ai.base] += 1;
                }
                p = q + 1;
                ++j;
                continue;
            }
        }
        while (p < end && *p == ' ') ++p;                       // strtok_r skips runs of delimiters
        if (p >= end || *p == '\n') break;
        const char *tok = p;
        while (p < end && *p != ' ' && *p != '\n') ++p;
        const char c = tok[0];
        if (c != '+' && c != '-' && c != 'N' && c != '.') {
            const char *q = tok;
            ai.is_indel = 0;
            int field[5] = {ai.base, ai.mapq, ai.qual, ai.rpr, ai.strand};
            for (int i = 0; i < 5 && q < p; ++i) {              // missing fields keep the previous value
                field[i] = atoi_span(q, p);
                while (q < p && *q != ',') ++q;                 // atoi stops at the first non-digit
                if (q < p) ++q;
            }
            ai.base = (uint8_t)(field[0] & 7);                  // bit-field widths, src/BamProcess.h:32-37
            ai.mapq = (uint8_t)field[1];
            ai.qual = (uint8_t)field[2];
            ai.rpr = (uint8_t)field[3];
            ai.strand = (uint8_t)(field[4] & 1);
            if (ai.base != 4) {                                 // skip N base, :427
                *e++ = Entry{ai.base, ai.mapq, ai.qual, ai.rpr, ai.strand, 0, 0};
                *sj++ = j;
                tally[ai.strand << 3 | ai.base] += 1;
            }
        } else if (c != '.') {
            ai.is_indel = 1;
            ai.indel.assign(tok, p - tok);
            site.indels.push_back(ai.indel);
            *e++ = Entry{ai.base, ai.mapq, ai.qual, ai.rpr, ai.strand, 1, 0};
            *sj++ = j;
            if (ai.base < 8u) (ai.strand == 1 ? site.fwd : site.rev)[ai.base] += 1;   // an indel entry counts with the fields it inherits
        }
        ++j;
    }
    const size_t n_used = (size_t)(e - site.aiv.data());
    site.aiv.resize(n_used);
    site.sample.resize(n_used);
    for (int b = 0; b < 8; ++b) { site.rev[b] += tally[b]; site.fwd[b] += tally[8 + b]; }
    for (int b = 0; b < 4; ++b) site.cnt[b] += tally[b] + tally[8 + b];
    return j - j0;
}

// ---- temp-batch binary form ---------------------------------------------------------------------------
const char kBinBatchMagic[8] = {'B', 'V', 'C', 'B', 'A', 'T', '1', '\n'};

static inline void put_u32(std::string &o, uint32_t v) { for (int i = 0; i < 4; ++i) o.push_back((char)((v >> (8 * i)) & 0xff)); }
static inline uint32_t get_u32(const unsigned char *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

void bin_batch_header(uint32_t n_samples_in_batch, const std::string &names_line, std::string &out)
{
    out.append(kBinBatchMagic, 8);
    put_u32(out, n_samples_in_batch);
    put_u32(out, (uint32_t)names_line.size());
    out += names_line;
}

void bin_batch_entry(const AlleleInfo &a, uint32_t j, std::string &payload)
{
    put_u32(payload, j);
    payload.push_back((char)a.base); payload.push_back((char)a.mapq); payload.push_back((char)a.qual);
    payload.push_back((char)a.rpr);
    payload.push_back((char)((a.strand & 1) | (a.is_indel ? 2 : 0)));
    if (a.is_indel) {
        const size_t n = a.indel.size() < 0xffff ? a.indel.size() : 0xffff;
        payload.push_back((char)(n & 0xff)); payload.push_back((char)(n >> 8));
        payload.append(a.indel.data(), n);
    }
}

bool parse_pileup_bin(const unsigned char *p, size_t len, int32_t j0, SiteColumn &site)
{
    const unsigned char *end = p + len;
    AlleleInfo &ai = g_carry;
    // An entry takes at least nine bytes of payload: room for all of them is made once, the entries are written in place
    // and the tallies are kept in locals (SiteColumn::add does the same one entry at a time: a capacity test and two
    // read-modify-writes through the object per entry, which is most of this loop's time at 1e5 samples).
    const size_t n0 = site.aiv.size();
    site.aiv.resize(n0 + len / 9);
    site.sample.resize(n0 + len / 9);
    Entry *e = site.aiv.data() + n0;
    int32_t *sj = site.sample.data() + n0;
    int32_t tally[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // [strand][base] over the base tokens
    uint8_t base = ai.base, mapq = ai.mapq, qual = ai.qual, rpr = ai.rpr, strand = ai.strand, is_indel = ai.is_indel;
    bool ok = true;
    while (p < end) {
        if (end - p < 9) { ok = false; break; }
        const int32_t j = j0 + (int32_t)get_u32(p);
        const unsigned flags = p[8];
        if (flags & 2) {                                         // indel token: only is_indel / indel change (:431-436)
            if (end - p < 11) { ok = false; break; }
            const size_t n = (size_t)p[9] | ((size_t)p[10] << 8);
            if ((size_t)(end - p) < 11 + n) { ok = false; break; }
            is_indel = 1;
            site.indels.emplace_back(reinterpret_cast<const char *>(p + 11), n);
            *e++ = Entry{base, mapq, qual, rpr, strand, 1, 0};
            *sj++ = j;
            (strand == 1 ? site.fwd : site.rev)[base] += 1;      // an indel entry counts with the fields it inherits
            p += 11 + n;
        } else {                                                 // base token, through the same bit-field widths
            is_indel = 0;
            base = (uint8_t)(p[4] & 7);
            mapq = p[5]; qual = p[6]; rpr = p[7];
            strand = (uint8_t)(flags & 1);
            if (base != 4) {                                     // skip N base, :427
                *e++ = Entry{base, mapq, qual, rpr, strand, 0, 0};
                *sj++ = j;
                tally[strand << 3 | base] += 1;
            }
            p += 9;
        }
    }
    ai.base = base; ai.mapq = mapq; ai.qual = qual; ai.rpr = rpr; ai.strand = strand; ai.is_indel = is_indel;
    const size_t n_used = (size_t)(e - site.aiv.data());
    site.aiv.resize(n_used);
    site.sample.resize(n_used);
    for (int b = 0; b < 8; ++b) { site.rev[b] += tally[b]; site.fwd[b] += tally[8 + b]; }
    for (int b = 0; b < 4; ++b) site.cnt[b] += tally[b] + tally[8 + b];
    return ok;
}

// ---- headers ---------------------------------------------------------------------------------------
const char *const kCvgHeader =
    "##fileformat=CVGv1.0\n"
    "##Group information is the depth of A:C:G:T:Indel\n"
    "#CHROM\tPOS\tREF\tDepth\tA\tC\tG\tT\tIndels\tFS\tSOR\tStrand_Coverage(REF_FWD,REF_REV,ALT_FWD,ALT_REV)";

const char *const kVcfHeader =
    "##fileformat=VCFv4.2\n"
    "##FILTER=<ID=LowQual,Description=\"Low quality (QUAL < 60)\">\n"
    "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n"
    "##FORMAT=<ID=AB,Number=1,Type=String,Description=\"Allele Base\">\n"
    "##FORMAT=<ID=SO,Number=1,Type=String,Description=\"Strand orientation of the mapping base. Marked as + or -\">\n"
    "##FORMAT=<ID=BP,Number=1,Type=String,Description=\"Base Probability which calculate by base quality\">\n"
    "##INFO=<ID=CM_AF,Number=A,Type=Float,Description=\"An ordered, comma delimited list of allele frequencies base on LRT algorithm\">\n"
    "##INFO=<ID=CM_CAF,Number=A,Type=Float,Description=\"An ordered, comma delimited list of allele frequencies just base on read count\">\n"
    "##INFO=<ID=CM_AC,Number=A,Type=Integer,Description=\"An ordered, comma delimited allele depth in CMDB\">\n"
    "##INFO=<ID=CM_DP,Number=A,Type=Integer,Description=\"Total Depth\">\n"
    "##INFO=<ID=SB_REF,Number=A,Type=Integer,Description=\"Read number support REF: Forward,Reverse\">\n"
    "##INFO=<ID=SB_ALT,Number=A,Type=Integer,Description=\"Read number support ALT: Forward,Reverse\">\n"
    "##INFO=<ID=FS,Number=1,Type=Float,Description=\"Phred-scaled p-value using Fisher's exact test to detect strand bias\">\n"
    "##INFO=<ID=BaseQRankSum,Number=1,Type=Float,Description=\"Phred-score from Wilcoxon rank sum test of Alt Vs. Ref base qualities\">\n"
    "##INFO=<ID=SOR,Number=1,Type=Float,Description=\"Symmetric Odds Ratio of 2x2 contingency table to detect strand bias\">\n"
    "##INFO=<ID=MQRankSum,Number=1,Type=Float,Description=\"Phred-score From Wilcoxon rank sum test of Alt vs. Ref read mapping qualities\">\n"
    "##INFO=<ID=ReadPosRankSum,Number=1,Type=Float,Description=\"Phred-score from Wilcoxon rank sum test of Alt vs. Ref read position bias\">\n"
    "##INFO=<ID=QD,Number=1,Type=Float,Description=\"Variant Confidence Quality by Depth\">\n";

std::string cvg_header(const Groups &g)                                   // src/BaseVarC.cpp:318, 366, 381
{
    std::string h = kCvgHeader;
    for (auto const &n : g.names) h += "\t" + n;
    h += "\n";
    return h;
}

std::string vcf_header(const Groups &g, const std::string &reference, const std::vector<std::string> &sample_names)
{
    std::string h = kVcfHeader;                                            // src/BaseVarC.cpp:319, 364-380
    for (auto const &n : g.names)
        h += "##INFO=<ID=" + n + "_AF,Number=A,Type=Float,Description=\"Allele frequency in the " + n +
             " populations calculated based on LRT.[0,1]\">\n";
    if (FILE *f = std::fopen((reference + ".fai").c_str(), "r")) {
        char contig[512], len[64], t1[64], t2[64], t3[64];
        while (std::fscanf(f, "%511s %63s %63s %63s %63s", contig, len, t1, t2, t3) == 5)
            h += std::string("##contig=<ID=") + contig + ",length=" + len + ">\n";
        std::fclose(f);
    }
    h += "##reference=file://" + reference + "\n";
    h += "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t";
    for (size_t i = 0; i < sample_names.size(); ++i) { if (i) h += "\t"; h += sample_names[i]; }
    h += "\n";
    return h;
}

// ---- CVG line: bt_f, src/BaseVarC.cpp:548-610 + group columns :617-663 ------------------------------------
std::string cvg_line(const std::string &chr, int8_t ref_base, const SiteView &site, const bvc_group_result *grp, int n_groups)
{
    const int32_t pos = site.pos;
    // depth per base (non-indel entries) and forward / reverse counts per base value of EVERY entry -- indel entries
    // take part in the strand tallies with the fields they carry (src/BaseVarC.cpp:575-590 reads a.base and a.strand
    // of all of aiv) -- were taken as the entries were appended (SiteColumn::add); the ref and alt columns are picked
    // from them here.
    const int32_t *cnt = site.cnt, *fwd = site.fwd, *rev = site.rev;
    std::map<std::string, int> indel_m;    // the reference iterates a hash map here (:570-573): order by key instead
    for (size_t i = 0; i < site.n_indels; ++i) indel_m[site.indels[i]] += 1;
    std::string indels = ".";
    if (!indel_m.empty()) {
        indels.clear();
        for (auto const &kv : indel_m) indels += kv.first + "|" + std::to_string(kv.second) + ",";
        indels.pop_back();
    }
    // sortidx: indices by descending count; std::sort on four elements is an insertion sort, ties keep order
    int didx[4] = {0, 1, 2, 3};
    std::stable_sort(didx, didx + 4, [cnt](int a, int b) { return cnt[a] > cnt[b]; });
    int alt_base = (didx[0] != ref_base) ? didx[0] : didx[1];
    // an entry counts as ref when its base equals ref_base, else as alt when it equals alt_base (alt_base != ref_base
    // unless ref_base is outside 0..3, where no entry matches it)
    const bool ref_ok = ref_base >= 0 && ref_base < 8;
    const int ref_fwd = ref_ok ? fwd[ref_base] : 0, ref_rev = ref_ok ? rev[ref_base] : 0;
    const int alt_fwd = (ref_ok && alt_base == ref_base) ? 0 : fwd[alt_base], alt_rev = (ref_ok && alt_base == ref_base) ? 0 : rev[alt_base];
    const double fs = bt_fisher_exact(ref_fwd, ref_rev, alt_fwd, alt_rev);
    const double sor = (alt_fwd * ref_rev > 0) ? (double)(ref_fwd * alt_rev) / (ref_rev * alt_fwd) : 10000.0;
    const int dep = cnt[0] + cnt[1] + cnt[2] + cnt[3];
    char buf[256];
    std::string out = chr + "\t" + std::to_string(pos) + "\t" + kBase2Char[ref_base & 3] + "\t";
    std::snprintf(buf, sizeof buf, "%d\t%d\t%d\t%d\t%d\t", dep, cnt[0], cnt[1], cnt[2], cnt[3]);
    out += buf;
    out += indels + "\t" + fmt_fixed(fs, 3) + "\t" + fmt_fixed(sor, 3) + "\t";
    std::snprintf(buf, sizeof buf, "%d,%d,%d,%d\t", ref_fwd, ref_rev, alt_fwd, alt_rev);
    out += buf;
    for (int g = 0; g < n_groups; ++g) {
        std::snprintf(buf, sizeof buf, "%d:%d:%d:%d\t", grp[g].depth[0], grp[g].depth[1], grp[g].depth[2], grp[g].depth[3]);
        out += buf;
    }
    out.pop_back();
    out += "\n";
    return out;
}

void group_af_info(const bvc_site_result &bt, const bvc_group_result *grp, const Groups &g,
                   std::map<std::string, std::string> &info)          // src/BaseVarC.cpp:641-659
{
    for (size_t k = 0; k < g.names.size(); ++k) {
        std::string af;
        if (grp[k].ran) {
            for (int i = 0; i < bt.n_alt; ++i) {
                if (grp[k].present & (1u << i)) af += fmt_fixed(grp[k].af[i], 6) + ",";
                else af += "0,";
            }
            if (!af.empty()) af.pop_back();
        } else {
            af = "0";
        }
        info.insert({g.names[k] + "_AF", af});
    }
}

// The BP sub-field depends only on the 8-bit quality: 256 strings, formatted once.
static const std::string &bp_field(uint8_t qual)
{
    static const std::vector<std::string> table = [] {
        std::vector<std::string> t(256);
        char b[32];
        for (int q = 0; q < 256; ++q) {
            std::snprintf(b, sizeof b, ":%.6f\t", 1 - std::exp(kMln10To10 * q));
            t[(size_t)q] = b;
        }
        return t;
    }();
    return table[qual];
}

// ---- VCF line: WriteVcf, src/BaseType.cpp:141-234 ---------------------------------------------------------
std::string vcf_line(const bvc_site_result &bt, const std::string &chr, int8_t ref_base, const SiteView &site,
                     std::map<std::string, std::string> &info, int32_t n_samples)
{
    const int32_t pos = site.pos;
    std::string alt_gt[8];                                        // genotype string per base code 0..7
    bool has_gt[8] = {false, false, false, false, false, false, false, false};
    auto is_alt = [&bt](int b) { for (int i = 0; i < bt.n_alt; ++i) if (bt.alt_base[i] == b) return true; return false; };
    for (int i = 0; i < bt.n_alt; ++i) { alt_gt[bt.alt_base[i] & 7] = "./" + std::to_string(i + 1); has_gt[bt.alt_base[i] & 7] = true; }
    int ref_fwd = 0, ref_rev = 0, alt_fwd = 0, alt_rev = 0;
    std::vector<double> ref_quals, ref_mapqs, ref_rprs, alt_quals, alt_mapqs, alt_rprs;
    // one field per SAMPLE: "./.\t" for a sample without an entry -- nine in ten at the coverage this tool is for, so the runs between
    // the covered samples are appended whole from a block of the pattern (the sample loop below visits the covered ones only)
    static const std::string kNoCall = [] { std::string p; for (int i = 0; i < 4096; ++i) p += "./.\t"; return p; }();
    auto no_calls = [](std::string &s, int64_t n) {
        for (; n > 0; n -= 4096) s.append(kNoCall, 0, (size_t)(n < 4096 ? n : 4096) * 4);
    };
    std::string samgt;
    samgt.reserve((size_t)n_samples * 4 + site.n * 16 + 16);
    int32_t next = 0;                                              // first sample not written yet
    for (size_t k = 0; k < site.n; ++k) {
        const int32_t i = site.sample[k];
        if (i < next || i >= n_samples) break;                    // (entries are in sample order, one per sample: what bt_s builds;
                                                                   //  an entry out of order would hold every later one back)
        no_calls(samgt, i - next);
        next = i + 1;
        const Entry &a = site.aiv[k];
        if (!has_gt[a.base]) { alt_gt[a.base] = "./."; has_gt[a.base] = true; }
        const std::string &gt = (a.base == ref_base) ? std::string("0/.") : alt_gt[a.base];
        // BASE2CHAR has four entries in the reference (src/BaseType.h:24); an indel entry carrying an N base
        // would index past it there -- 'N' is printed here.
        samgt += gt;
        samgt += ':';
        samgt += kBase2Char[a.base < 6 ? a.base : 5];
        samgt += ':';
        samgt += kStrand[a.strand & 1];
        samgt += bp_field(a.qual);                          // ":<1 - 10^(-qual/10) as {:.6f}>\t"
        if (a.is_indel == 1 || a.base == 4) continue;
        const bool alt = is_alt(a.base);
        if (a.base == ref_base) { ref_quals.push_back(a.qual); ref_mapqs.push_back(a.mapq); ref_rprs.push_back(a.rpr); }
        else if (alt) { alt_quals.push_back(a.qual); alt_mapqs.push_back(a.mapq); alt_rprs.push_back(a.rpr); }
        if (a.strand == 1) { if (a.base == ref_base) ref_fwd += 1; else if (alt) alt_fwd += 1; }
        else { if (a.base == ref_base) ref_rev += 1; else if (alt) alt_rev += 1; }
    }
    no_calls(samgt, n_samples - next);
    const double phred_mapq = RankSumTest(ref_mapqs, alt_mapqs);
    const double phred_qual = RankSumTest(ref_quals, alt_quals);
    const double phred_rpr = RankSumTest(ref_rprs, alt_rprs);
    const double fs = bt_fisher_exact(ref_fwd, ref_rev, alt_fwd, alt_rev);
    const double sor = (alt_fwd * ref_rev > 0) ? (double)(ref_fwd * alt_rev) / (ref_rev * alt_fwd) : 10000.0;
    double ad_sum = 0;
    std::string ac, af, caf, alt;
    for (int i = 0; i < bt.n_alt; ++i) {
        const int b = bt.alt_base[i];
        ad_sum += bt.depth[b];
        alt += kBase2Char[b & 3]; alt += ",";
        ac += std::to_string(bt.depth[b]) + ",";
        af += fmt_fixed(bt.af[i], 6) + ",";
        caf += fmt_fixed(bt.depth[b] / bt.depth_total, 6) + ",";
    }
    alt.pop_back();
    if (!samgt.empty()) samgt.pop_back();
    ac.pop_back(); info.insert({"CM_AC", ac});
    af.pop_back(); info.insert({"CM_AF", af});
    caf.pop_back(); info.insert({"CM_CAF", caf});
    info.insert({"QD", fmt_fixed(bt.var_qual / ad_sum, 3)});
    info.insert({"CM_DP", fmt_fixed(bt.depth_total, 0)});
    info.insert({"MQRankSum", fmt_fixed(phred_mapq, 3)});
    info.insert({"ReadPosRankSum", fmt_fixed(phred_rpr, 3)});
    info.insert({"BaseQRankSum", fmt_fixed(phred_qual, 3)});
    info.insert({"FS", fmt_fixed(fs, 3)});
    info.insert({"SOR", fmt_fixed(sor, 3)});
    info.insert({"SB_REF", std::to_string(ref_fwd) + "," + std::to_string(ref_rev)});
    info.insert({"SB_ALT", std::to_string(alt_fwd) + "," + std::to_string(alt_rev)});
    const char *qt = (bt.var_qual > kQualThreshold) ? "." : "LowQual";
    std::string out = chr + "\t" + std::to_string(pos) + "\t.\t" + kBase2Char[ref_base & 3] + "\t" + alt + "\t" +
                      fmt_fixed(bt.var_qual, 2) + "\t" + qt + "\t";
    for (auto const &kv : info) out += kv.first + "=" + kv.second + ";";
    out.pop_back();
    out += "\tGT:AB:SO:BP\t" + samgt + "\n";
    return out;
}

// ---- orchestration ---------------------------------------------------------------------------------------
void thread_window(size_t psize, int thread, int ithread, size_t &lo, size_t &hi)
{
    const size_t window = psize % (size_t)thread + psize / (size_t)thread;
    lo = std::min(psize, (size_t)ithread * window);
    hi = (ithread == thread - 1) ? psize : std::min(psize, (size_t)(ithread + 1) * window);
}

int device_of_thread(int ithread, int devices_present, int gpus_option)
{
    int g = devices_present;
    if (gpus_option > 0 && gpus_option < g) g = gpus_option;
    return g > 0 ? ithread % g : 0;
}

bool merge_subfiles(const std::string &out_prefix, const std::string &suffix, int thread, BgzfWriter &dst)
{
    for (int i = 0; i < thread; ++i) {
        const std::string sub = out_prefix + "." + std::to_string(i) + suffix;
        // the reference re-reads and re-compresses the sub-files line by line (src/BaseVarC.cpp:279-290); BGZF files
        // concatenate block for block, which yields the same uncompressed stream
        if (!dst.append_file(sub)) return false;
        std::remove(sub.c_str());
    }
    return true;
}

}  // namespace bvchost
