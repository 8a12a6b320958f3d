// bam.h -- BAM records, region fetch and the per-sample pileup rule (host side, f4).
// Counterparts in the reference: BamProcess::GetBRV / FindSnpAtPos / GetAllele / GetOffset
// (src/BamProcess.cpp:4-94, 214-304) and RefReader::GetTargetBase (src/RefReader.h:17-33), which sit on
// SeqLib + htslib there (both absent from the reference tree).  BAM/BAI/faidx are implemented from the
// SAM specification; the pileup rule follows the reference's source text.
#ifndef BVC_HOST_BAM_H
#define BVC_HOST_BAM_H

#include <cstdint>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "pileup.h"

namespace bvchost {

struct BamRecord {
    int32_t ref_id = -1;
    int32_t pos = 0;                    // 0-based leftmost
    uint8_t mapq = 0;
    uint16_t flag = 0;
    std::vector<std::pair<char, int32_t>> cigar;
    std::string seq;
    std::string qual;                   // raw phred values (not +33)
    int32_t end_pos() const;            // 0-based exclusive = 1-based inclusive (bam_endpos)
    bool reverse() const { return flag & 0x10; }
    bool mate_reverse() const { return flag & 0x20; }
    bool duplicate() const { return flag & 0x400; }
};

class BamFile {
 public:
    bool open(const std::string &path);
    const std::string &header_text() const { return header_; }
    std::string sample_name() const;                         // first "SM:" of the header, src/BamProcess.cpp:273-287
    bool sorted() const { return header_.find("SO:coord") != std::string::npos; }
    int ref_index(const std::string &name) const;
    // Records overlapping [beg, end) of reference `rid` (0-based), duplicates and MAPQ < mapq dropped
    // (src/BamProcess.cpp:295-300).  Uses the .bai linear index when present, else scans the file.
    bool fetch(int rid, int32_t beg, int32_t end, int min_mapq, std::vector<BamRecord> &out);
 private:
    std::string path_, header_;
    std::vector<std::string> ref_names_;
    uint64_t first_record_ = 0;
};

typedef std::unordered_map<int32_t, AlleleInfo> PosAlleleMap;   // src/BamProcess.h:40
// BamProcess::FindSnpAtPos (src/BamProcess.cpp:4-94): first usable read per position.
void find_snp_at_pos(const std::vector<BamRecord> &rv, int32_t rg_s, const std::string &refseq,
                     const std::vector<int32_t> &pv, PosAlleleMap &allele_m);

// RefReader::GetTargetBase: 1-based inclusive region of a faidx-indexed FASTA (plain or BGZF + .gzi),
// upper-cased.
bool fetch_reference(const std::string &fasta, const std::string &chr, int32_t start1, int32_t end1, std::string &seq,
                     std::string &err);

}  // namespace bvchost
#endif
