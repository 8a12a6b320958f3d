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

bool BamFile::open(const std::string &path)
{
    path_ = path;
    BgzfReader r(path);
    if (!r.ok() || !r.is_bgzf()) return false;
    unsigned char b[8];
    if (!read_exact(r, b, 8) || std::memcmp(b, "BAM\1", 4) != 0) return false;
    const int32_t l_text = le32(b + 4);
    header_.resize((size_t)l_text);
    if (l_text && !read_exact(r, &header_[0], (size_t)l_text)) return false;
    if (!read_exact(r, b, 4)) return false;
    const int32_t n_ref = le32(b);
    ref_names_.clear();
    for (int32_t i = 0; i < n_ref; ++i) {
        if (!read_exact(r, b, 4)) return false;
        const int32_t l = le32(b);
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

static bool read_record(BgzfReader &r, BamRecord &rec, std::vector<unsigned char> &buf)
{
    unsigned char b4[4];
    if (!read_exact(r, b4, 4)) return false;
    const int32_t bs = le32(b4);
    if (bs < 32) return false;
    buf.resize((size_t)bs);
    if (!read_exact(r, buf.data(), buf.size())) return false;
    const unsigned char *p = buf.data();
    rec.ref_id = le32(p);
    rec.pos = le32(p + 4);
    const int l_rn = p[8];
    rec.mapq = p[9];
    const int n_cig = p[12] | (p[13] << 8);
    rec.flag = (uint16_t)(p[14] | (p[15] << 8));
    const int32_t l_seq = le32(p + 16);
    const unsigned char *q = p + 32 + l_rn;
    rec.cigar.clear();
    for (int i = 0; i < n_cig; ++i, q += 4) {
        const uint32_t c = (uint32_t)le32(q);
        rec.cigar.emplace_back(kCigarOps[(c & 15) < 9 ? (c & 15) : 0], (int32_t)(c >> 4));
    }
    rec.seq.resize((size_t)l_seq);
    for (int32_t i = 0; i < l_seq; ++i) rec.seq[i] = kSeqNt16[(q[i >> 1] >> ((i & 1) ? 0 : 4)) & 15];
    q += (l_seq + 1) / 2;
    rec.qual.assign(reinterpret_cast<const char *>(q), (size_t)l_seq);
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
    while (read_record(r, rec, buf)) {
        if (rec.ref_id < 0 || rec.ref_id < rid) continue;
        if (rec.ref_id > rid || rec.pos >= end) break;                  // coordinate-sorted
        if ((rec.flag & 0x4) || rec.cigar.empty()) continue;
        if (rec.end_pos() <= beg) continue;
        if (rec.duplicate()) continue;                                  // src/BamProcess.cpp:297
        if (rec.mapq < min_mapq) continue;                              // :298
        out.push_back(rec);
    }
    return true;
}

// ---- pileup rule ---------------------------------------------------------------------------------------
static int get_offset(const BamRecord &r, int32_t pos)                   // src/BamProcess.cpp:232-261
{
    uint32_t offset = (uint32_t)(pos - (r.pos + 1));
    uint32_t track = (uint32_t)r.pos;
    for (auto const &cf : r.cigar) {
        const char t = cf.first;
        if (t != 'I' && t != 'S' && t != 'H') track += (uint32_t)cf.second;
        if (track < (uint32_t)pos) {
            switch (t) {
            case 'I': case 'S': offset += (uint32_t)cf.second; break;
            case 'D': case 'P': case 'N': offset -= (uint32_t)cf.second; break;
            default: break;
            }
        } else {
            break;
        }
    }
    if (r.seq.empty() || offset > r.seq.length() - 1) throw std::out_of_range("index offset is out of range of the sequence");
    return (int)offset;
}

static void get_allele(const BamRecord &r, int32_t pos, AlleleInfo &ale)  // src/BamProcess.cpp:214-230
{
    const int offset = get_offset(r, pos);
    switch (r.seq[offset]) {
    case 'A': ale.base = 0; break;
    case 'C': ale.base = 1; break;
    case 'G': ale.base = 2; break;
    case 'T': ale.base = 3; break;
    default: ale.base = 4;
    }
    ale.qual = (uint8_t)r.qual[offset];          // qualities[offset] - 33 on the ASCII form
    ale.mapq = r.mapq;
    ale.rpr = (uint8_t)(offset + 1);
    ale.is_indel = 0;
    ale.strand = (r.reverse() || r.mate_reverse()) ? 0 : 1;
}

void find_snp_at_pos(const std::vector<BamRecord> &rv, int32_t rg_s, const std::string &refseq,
                     const std::vector<int32_t> &pv, PosAlleleMap &allele_m)
{
    if (rv.empty()) return;
    size_t i = 0, j = 0;
    const size_t last = rv.size() - 1;
    const BamRecord *r = &rv[0];
    AlleleInfo ale;
    for (int32_t pos : pv) {
        if (pos < r->pos + 1) continue;
        bool eof = false, next = false;
        while (pos > r->end_pos()) {
            if (i == last) { eof = true; break; }
            r = &rv[++i]; j = i;
            if (pos < r->pos + 1) { next = true; break; }
        }
        if (next || eof) continue;
        for (;;) {
            const auto &c = r->cigar;
            const size_t nc = c.size();
            size_t k;
            int sx = r->pos, sy = 0;
            for (k = 0; k < nc; ++k) {
                const char op = c[k].first;
                const int l = c[k].second;
                if (op == 'M' || op == 'I' || op == 'S' || op == 'X') sy += l;
                if (op == 'H' || op == 'I') continue;
                sx += l;                             // note: soft clips and pads advance sx too, as in the reference
                if (pos <= sx) break;
            }
            if (k >= nc) break;                      // the reference asserts here
            const size_t sk = k;
            const char op = c[sk].first;
            int indel = 0;
            std::string indel_str;
            if (sx == pos && sk + 1 < nc) {
                const char op2 = c[sk + 1].first;
                const int l2 = c[sk + 1].second;
                if (op2 == 'D') { indel = -l2; indel_str = "-" + refseq.substr((size_t)(pos - rg_s + 1), (size_t)l2); }
                else if (op2 == 'I') { indel = l2; indel_str = "+" + r->seq.substr((size_t)sy, (size_t)l2); }
                else if (op2 == 'P' && sk + 2 < nc) {
                    indel_str = "N";
                    int l3 = 0;                      // the reference's loop re-reads c[sk] (src/BamProcess.cpp:60-62)
                    for (size_t kk = sk + 2; kk < nc; ++kk) {
                        const char o = c[sk].first;
                        if (o == 'I') l3 += c[sk].second;
                        else if (o == 'D' || o == 'M' || o == 'N' || o == 'X') break;
                    }
                    if (l3 > 0) indel = l3;
                }
            }
            if (indel != 0) {
                ale.strand = (r->reverse() || r->mate_reverse()) ? 0 : 1;
                ale.base = 5; ale.qual = r->mapq; ale.rpr = 0; ale.is_indel = 1; ale.indel = indel_str;
                allele_m.insert({pos, ale});
                break;
            }
            if (op != 'D' && op != 'N') {
                get_allele(*r, pos, ale);
                ale.indel.clear();
                allele_m.insert({pos, ale});
                break;
            } else if (j < last) {
                r = &rv[++j];
                if (pos < r->pos + 1 || pos > r->end_pos()) break;
            } else {
                break;
            }
        }
        r = &rv[i]; j = i;
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
