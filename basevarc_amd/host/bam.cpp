#include "bam.h"

#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

#include "bgzf.h"

namespace bvchost {

static const char kSeqNt16[] = "=ACMGRSVTWYHKDBN";
static const char kCigarOps[] = "MIDNSHP=X";

int32_t BamRecord::end_pos() const
{
    int32_t e = pos;
    for (auto const &c : cigar)
        if (c.first == 'M' || c.first == 'D' || c.first == 'N' || c.first == '=' || c.first == 'X') e += c.second;
    return e;
}

static bool read_exact(BgzfReader &r, void *dst, size_t n) { return r.read(dst, n) == n; }

static int32_t le32(const unsigned char *p) { return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24)); }

// Every length field of the file is checked against what was actually read before it is used (the reference gets
// this from htslib): a truncated or corrupt file ends the parse with `false`, never with a read past a buffer.
static const int32_t kMaxHeaderText = 1 << 28, kMaxRefs = 1 << 24, kMaxRefName = 1 << 16, kMaxRecord = 1 << 28;

bool BamFile::open(const std::string &path)
{
    path_ = path;
    BgzfReader r(path);
    if (!r.ok() || !r.is_bgzf()) return false;
    unsigned char b[8];
    if (!read_exact(r, b, 8) || std::memcmp(b, "BAM\1", 4) != 0) return false;
    const int32_t l_text = le32(b + 4);
    if (l_text < 0 || l_text > kMaxHeaderText) return false;
    header_.resize((size_t)l_text);
    if (l_text && !read_exact(r, &header_[0], (size_t)l_text)) return false;
    if (!read_exact(r, b, 4)) return false;
    const int32_t n_ref = le32(b);
    if (n_ref < 0 || n_ref > kMaxRefs) return false;
    ref_names_.clear();
    for (int32_t i = 0; i < n_ref; ++i) {
        if (!read_exact(r, b, 4)) return false;
        const int32_t l = le32(b);
        if (l < 1 || l > kMaxRefName) return false;
        std::string name((size_t)l, '\0');
        if (!read_exact(r, &name[0], (size_t)l) || !read_exact(r, b, 4)) return false;
        name.resize(std::strlen(name.c_str()));
        ref_names_.push_back(name);
    }
    first_record_ = r.tell();
    return true;
}

std::string BamFile::sample_name() const
{
    size_t p = header_.find("SM:");
    if (p == std::string::npos) throw std::runtime_error("ERROR: No SM tag can be found. Please make sure there is SM tag in the bam header");
    std::string hh = header_.substr(p + 3);
    if ((p = hh.find("\n")) != std::string::npos) hh.erase(p);
    if ((p = hh.find("\t")) != std::string::npos) hh.erase(p);
    return hh;
}

int BamFile::ref_index(const std::string &name) const
{
    for (size_t i = 0; i < ref_names_.size(); ++i) if (ref_names_[i] == name) return (int)i;
    return -1;
}

// One alignment record (SAM specification 4.2).  Returns false at end of file AND on a malformed record; `bad` tells
// the two apart.
static bool read_record(BgzfReader &r, BamRecord &rec, std::vector<unsigned char> &buf, bool &bad)
{
    unsigned char b4[4];
    bad = false;
    const size_t got = r.read(b4, 4);
    if (got == 0) return false;                                   // clean end of file
    bad = true;
    if (got != 4) return false;
    const int32_t bs = le32(b4);
    if (bs < 32 || bs > kMaxRecord) return false;
    buf.resize((size_t)bs);
    if (!read_exact(r, buf.data(), buf.size())) return false;
    const unsigned char *p = buf.data();
    rec.ref_id = le32(p);
    rec.pos = le32(p + 4);
    const int64_t l_rn = p[8];
    rec.mapq = p[9];
    const int64_t n_cig = p[12] | (p[13] << 8);
    rec.flag = (uint16_t)(p[14] | (p[15] << 8));
    const int64_t l_seq = le32(p + 16);
    // fixed part + read name + cigar + packed bases + qualities must lie inside the block
    if (l_seq < 0 || 32 + l_rn + 4 * n_cig + (l_seq + 1) / 2 + l_seq > (int64_t)bs) return false;
    const unsigned char *q = p + 32 + l_rn;
    rec.cigar.clear();
    for (int64_t i = 0; i < n_cig; ++i, q += 4) {
        const uint32_t c = (uint32_t)le32(q);
        rec.cigar.emplace_back(kCigarOps[(c & 15) < 9 ? (c & 15) : 0], (int32_t)(c >> 4));
    }
    rec.seq.resize((size_t)l_seq);
    for (int64_t i = 0; i < l_seq; ++i) rec.seq[(size_t)i] = kSeqNt16[(q[i >> 1] >> ((i & 1) ? 0 : 4)) & 15];
    q += (l_seq + 1) / 2;
    rec.qual.assign(reinterpret_cast<const char *>(q), (size_t)l_seq);
    bad = false;
    return true;
}

// Smallest virtual offset of an alignment overlapping the 16 kbp window of `beg`, from the .bai linear index.
static bool bai_linear_offset(const std::string &bam_path, int rid, int32_t beg, uint64_t &voff)
{
    FILE *f = std::fopen((bam_path + ".bai").c_str(), "rb");
    if (!f) {
        std::string alt = bam_path;
        if (alt.size() > 4 && alt.substr(alt.size() - 4) == ".bam") alt = alt.substr(0, alt.size() - 4) + ".bai";
        f = std::fopen(alt.c_str(), "rb");
        if (!f) return false;
    }
    bool found = false;
    unsigned char b[8];
    auto rd32 = [&](int32_t &v) { if (std::fread(b, 1, 4, f) != 4) return false; v = le32(b); return true; };
    int32_t n_ref = 0;
    if (std::fread(b, 1, 4, f) == 4 && std::memcmp(b, "BAI\1", 4) == 0 && rd32(n_ref)) {
        for (int32_t r = 0; r < n_ref; ++r) {
            int32_t n_bin = 0;
            if (!rd32(n_bin)) break;
            bool bad = false;
            for (int32_t i = 0; i < n_bin && !bad; ++i) {
                int32_t bin, n_chunk;
                if (!rd32(bin) || !rd32(n_chunk) || std::fseek(f, (long)n_chunk * 16, SEEK_CUR) != 0) bad = true;
            }
            int32_t n_intv = 0;
            if (bad || !rd32(n_intv)) break;
            if (r == rid) {
                int32_t w = beg >> 14;
                if (w < 0) w = 0;
                if (n_intv > 0) {
                    if (w >= n_intv) w = n_intv - 1;
                    if (std::fseek(f, (long)w * 8, SEEK_CUR) == 0 && std::fread(b, 1, 8, f) == 8) {
                        voff = 0;
                        for (int i = 0; i < 8; ++i) voff |= (uint64_t)b[i] << (8 * i);
                        found = voff != 0;
                    }
                }
                break;
            }
            if (std::fseek(f, (long)n_intv * 8, SEEK_CUR) != 0) break;
        }
    }
    std::fclose(f);
    return found;
}

bool BamFile::fetch(int rid, int32_t beg, int32_t end, int min_mapq, std::vector<BamRecord> &out)
{
    BgzfReader r(path_);
    if (!r.ok()) return false;
    uint64_t voff = 0;
    if (!(bai_linear_offset(path_, rid, beg, voff) && r.seek(voff))) r.seek(first_record_);
    BamRecord rec;
    std::vector<unsigned char> buf;
    bool bad = false;
    while (read_record(r, rec, buf, bad)) {
        if (rec.ref_id < 0 || rec.ref_id < rid) continue;
        if (rec.ref_id > rid || rec.pos >= end) break;                  // coordinate-sorted
        if ((rec.flag & 0x4) || rec.cigar.empty()) continue;
        if (rec.end_pos() <= beg) continue;
        if (rec.duplicate()) continue;                                  // src/BamProcess.cpp:297
        if (rec.mapq < min_mapq) continue;                              // :298
        out.push_back(rec);
    }
    // not "this sample has no reads here": the file is damaged, and a pileup made from the readable part would be
    // silently incomplete
    if (bad) throw std::runtime_error("ERROR: truncated or corrupt BAM record in " + path_);
    return true;
}

// ---- pileup rule ---------------------------------------------------------------------------------------
// What has to come out is fixed by the reference (BamProcess::FindSnpAtPos / GetAllele / GetOffset,
// src/BamProcess.cpp:4-94, 214-261): per position, the entry of the FIRST read that covers it; an indel token when
// the position is the last reference base in front of an insertion or deletion; the next read instead when the first
// one has a deletion or reference skip there.  How it is computed here is different: the reference re-scans a read's
// CIGAR from its first operation for every position (twice: once to find the operation, once for the query offset).
// Here each read's CIGAR is walked ONCE into prefix tables in the coordinates the rule uses, and because positions
// only ascend, a cursor per table only ever moves forward: the work is O(operations + positions), not their product.
//
// The rule's coordinates, kept exactly (they are what the outputs depend on):
//   ref_after[k]   "sx": read start + lengths of every operation except H and I.  Soft clips and pads advance it too.
//   qry_after[k]   "sy": lengths of M, I, S and X ('=' does not count): where an insertion's bases are cut from.
//   aln_after[k]   GetOffset's "track": read start + lengths of every operation except I, S and H.
//   shift_after[k] GetOffset's running correction of the query offset: + I, + S, - D, - P, - N.
namespace {

struct ReadWalk {
    std::vector<int32_t> ref_after, qry_after, aln_after, shift_after;
    size_t k_ref = 0, k_aln = 0;               // cursors (positions are queried in ascending order)
    bool built = false;

    void build(const BamRecord &r)
    {
        const size_t nc = r.cigar.size();
        ref_after.resize(nc); qry_after.resize(nc); aln_after.resize(nc); shift_after.resize(nc);
        int32_t sx = r.pos, sy = 0, track = r.pos, shift = 0;
        for (size_t k = 0; k < nc; ++k) {
            const char op = r.cigar[k].first;
            const int32_t l = r.cigar[k].second;
            if (op == 'M' || op == 'I' || op == 'S' || op == 'X') sy += l;
            if (op != 'H' && op != 'I') sx += l;
            if (op != 'I' && op != 'S' && op != 'H') track += l;
            if (op == 'I' || op == 'S') shift += l;
            else if (op == 'D' || op == 'P' || op == 'N') shift -= l;
            ref_after[k] = sx; qry_after[k] = sy; aln_after[k] = track; shift_after[k] = shift;
        }
        built = true;
    }

    // Operation that holds `pos`: the first one whose ref_after reaches it (never an H or I: those leave ref_after
    // where the operation before them put it, which did not reach pos).  nc when the CIGAR ends short of pos.
    size_t op_at(int32_t pos)
    {
        while (k_ref < ref_after.size() && ref_after[k_ref] < pos) ++k_ref;
        return k_ref;
    }

    // Query offset of `pos` as GetOffset computes it: the corrections of all operations in front of the first one
    // whose aln_after reaches pos.
    int64_t query_offset(const BamRecord &r, int32_t pos)
    {
        while (k_aln < aln_after.size() && aln_after[k_aln] < pos) ++k_aln;
        const int32_t shift = k_aln == 0 ? 0 : shift_after[k_aln - 1];
        return (int64_t)(uint32_t)((uint32_t)(pos - (r.pos + 1)) + (uint32_t)shift);   // unsigned 32-bit, as in the reference
    }
};

inline uint8_t strand_of(const BamRecord &r) { return (r.reverse() || r.mate_reverse()) ? 0 : 1; }

}  // namespace

void find_snp_at_pos(const std::vector<BamRecord> &rv, int32_t rg_s, const std::string &refseq,
                     const std::vector<int32_t> &pv, PosAlleleMap &allele_m)
{
    if (rv.empty()) return;
    std::vector<ReadWalk> walks(rv.size());
    auto walk_of = [&](size_t j) -> ReadWalk & {
        if (!walks[j].built) walks[j].build(rv[j]);
        return walks[j];
    };
    auto covers = [&](size_t j, int32_t pos) { return pos >= rv[j].pos + 1 && pos <= rv[j].end_pos(); };

    size_t first = 0;                          // first read that has not ended in front of the current position
    AlleleInfo ale;                            // one object for the whole call, as in the reference: an indel entry
                                               // keeps the mapq of the base entry made before it
    for (int32_t pos : pv) {
        while (first + 1 < rv.size() && rv[first].end_pos() < pos) ++first;
        if (!covers(first, pos)) continue;     // nobody covers it (later reads start even further right)
        for (size_t j = first;; ++j) {
            const BamRecord &r = rv[j];
            ReadWalk &w = walk_of(j);
            const size_t k = w.op_at(pos);
            if (k >= r.cigar.size()) break;    // CIGAR shorter than the read's span claims (the reference asserts)
            const char op = r.cigar[k].first;
            // last reference base of its operation, with an insertion or a deletion next: the indel token
            if (w.ref_after[k] == pos && k + 1 < r.cigar.size()) {
                const char next_op = r.cigar[k + 1].first;
                const int32_t next_len = r.cigar[k + 1].second;
                // (a pad next never yields a token: the reference's scan for the insertion behind it re-reads the
                // current operation, which is never an insertion, src/BamProcess.cpp:60-66)
                if ((next_op == 'D' || next_op == 'I') && next_len != 0) {
                    ale.strand = strand_of(r);
                    ale.base = 5; ale.qual = r.mapq; ale.rpr = 0; ale.is_indel = 1;
                    ale.indel = next_op == 'D' ? "-" + refseq.substr((size_t)(pos - rg_s + 1), (size_t)next_len)
                                               : "+" + r.seq.substr((size_t)w.qry_after[k], (size_t)next_len);
                    allele_m.insert({pos, ale});
                    break;
                }
            }
            if (op != 'D' && op != 'N') {      // a base of this read
                const int64_t off = w.query_offset(r, pos);
                if (r.seq.empty() || off > (int64_t)r.seq.length() - 1)
                    throw std::out_of_range("index offset is out of range of the sequence");
                const char c = r.seq[(size_t)off];
                ale.base = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4;
                ale.qual = (uint8_t)r.qual[(size_t)off];
                ale.mapq = r.mapq;
                ale.rpr = (uint8_t)(off + 1);
                ale.is_indel = 0;
                ale.strand = strand_of(r);
                ale.indel.clear();
                allele_m.insert({pos, ale});
                break;
            }
            // deleted or skipped in this read: the NEXT read gets the position, if it covers it
            if (j + 1 >= rv.size() || !covers(j + 1, pos)) break;
        }
    }
}

// ---- faidx -------------------------------------------------------------------------------------------
bool fetch_reference(const std::string &fasta, const std::string &chr, int32_t start1, int32_t end1, std::string &seq,
                     std::string &err)
{
    std::ifstream fai(fasta + ".fai");
    if (!fai.is_open()) { err = "ERROR: reference must be index with samtools faidx"; return false; }
    std::string name;
    long long len = 0, off = 0, lb = 0, lw = 0;
    bool found = false;
    while (fai >> name >> len >> off >> lb >> lw) if (name == chr) { found = true; break; }
    if (!found || lb <= 0) { err = "ERROR: contig " + chr + " not in " + fasta + ".fai"; return false; }
    if (start1 < 1) start1 = 1;
    if (end1 > len) end1 = (int32_t)len;
    seq.clear();
    if (end1 < start1) return true;
    const long long b0 = off + (long long)(start1 - 1) / lb * lw + (start1 - 1) % lb;
    const long long b1 = off + (long long)(end1 - 1) / lb * lw + (end1 - 1) % lb + 1;
    std::string raw((size_t)(b1 - b0), '\0');
    BgzfReader probe(fasta);
    if (!probe.ok()) { err = "ERROR: can not open " + fasta; return false; }
    if (probe.is_bgzf()) {
        // BGZF FASTA: the .gzi index lists (compressed offset, uncompressed offset) of every block but the first
        std::vector<std::pair<uint64_t, uint64_t>> idx(1, {0, 0});
        if (FILE *g = std::fopen((fasta + ".gzi").c_str(), "rb")) {
            uint64_t n = 0;
            if (std::fread(&n, 8, 1, g) == 1)
                for (uint64_t i = 0; i < n; ++i) { uint64_t c, u; if (std::fread(&c, 8, 1, g) != 1 || std::fread(&u, 8, 1, g) != 1) break; idx.push_back({c, u}); }
            std::fclose(g);
        }
        size_t k = 0;
        while (k + 1 < idx.size() && idx[k + 1].second <= (uint64_t)b0) ++k;
        if (!probe.seek(idx[k].first << 16)) { err = "ERROR: seek in " + fasta; return false; }
        uint64_t skip = (uint64_t)b0 - idx[k].second;
        char junk[4096];
        while (skip > 0) { const size_t t = probe.read(junk, skip < sizeof junk ? (size_t)skip : sizeof junk); if (!t) break; skip -= t; }
        if (probe.read(&raw[0], raw.size()) != raw.size()) { err = "ERROR: short read from " + fasta; return false; }
    } else {
        std::ifstream fa(fasta, std::ios::binary);
        fa.seekg(b0);
        fa.read(&raw[0], (std::streamsize)raw.size());
        if ((size_t)fa.gcount() != raw.size()) { err = "ERROR: short read from " + fasta; return false; }
    }
    seq.reserve((size_t)(end1 - start1 + 1));
    for (char c : raw) {
        if (c == '\n' || c == '\r') continue;
        if (!((c >= 65 && c <= 90) || c == '-')) c = (char)(c ^ 0x20);   // src/RefReader.h:28-31
        seq.push_back(c);
    }
    return true;
}

}  // namespace bvchost
