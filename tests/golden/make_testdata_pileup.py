#!/usr/bin/env python3
"""Builds tests/golden/testdata_pileup.npz: the per-site (base, qual) vectors that `BaseVarC basetype`
would hand to BaseType on the reference's own test data (test/test.sh:3: -q 20, region
chr17:41197700-41276155, the 100 BAMs of test/bam.list), plus the oracle's expected records.

Run in the build container only (it reads /root/reference/test/data, which does not travel):
    python -m tests.golden.make_testdata_pileup

What is restated here (test infrastructure, NOT product code; parity of this stage is unpinned by
anything runnable -- the reference binary cannot be built and test/data/chr17.fa.gz is missing):
  * BAM decoding with gzip + struct (BGZF is multi-member gzip).
  * read selection: BamProcess::GetBRV, src/BamProcess.cpp:264-304 (region padded by 1000, duplicates and
    MAPQ < --mapq dropped).
  * per-sample pileup: BamProcess::FindSnpAtPos, src/BamProcess.cpp:4-94 (first usable read covering the
    position; a deletion / ref-skip falls through to the next read; an indel starting right after the
    position becomes an indel token), GetAllele :214-230, GetOffset :232-261.
  * the caller's filters: N bases dropped (src/BaseVarC.cpp:427), indel tokens kept out of the
    base/qual vectors (:551-559), positions without any sample skipped (:442).
  * reference bases: rebuilt from the reads' MD:Z tags because the FASTA blob is absent
    (.MISSING_LARGE_BLOBS:2); positions whose reference base no read reveals are skipped.
"""
import gzip
import os
import struct

import numpy as np

REF_ROOT = "/root/reference/test"
REGION = ("chr17", 41197700, 41276155)
MAPQ = 20
BASE_CODE = {"A": 0, "C": 1, "G": 2, "T": 3}
SEQ_NT16 = "=ACMGRSVTWYHKDBN"
CIGAR_OPS = "MIDNSHP=X"


def read_bam(path):
    d = gzip.open(path).read()
    assert d[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", d, 4)[0]
    header = d[8:8 + l_text].decode(errors="replace")
    off = 8 + l_text
    n_ref = struct.unpack_from("<i", d, off)[0]
    off += 4
    refs = []
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", d, off)[0]
        off += 4
        refs.append(d[off:off + ln - 1].decode())
        off += ln + 4
    recs = []
    while off < len(d):
        bs = struct.unpack_from("<i", d, off)[0]
        p = off + 4
        ref_id, pos, l_rn, mapq, _bin, n_cig, flag, l_seq = struct.unpack_from("<iiBBHHHi", d, p)
        q = p + 32 + l_rn
        cigar = [(CIGAR_OPS[c & 15], c >> 4) for c in struct.unpack_from(f"<{n_cig}I", d, q)]
        q += 4 * n_cig
        sb = d[q:q + (l_seq + 1) // 2]
        seq = "".join(SEQ_NT16[(sb[i >> 1] >> (4 if i % 2 == 0 else 0)) & 15] for i in range(l_seq))
        q += (l_seq + 1) // 2
        qual = d[q:q + l_seq]
        q += l_seq
        md = None
        end = off + 4 + bs
        while q < end:                     # aux fields: find MD:Z
            tag, typ = d[q:q + 2], chr(d[q + 2])
            q += 3
            if typ == "Z":
                e = d.index(b"\0", q)
                if tag == b"MD":
                    md = d[q:e].decode()
                q = e + 1
            elif typ == "H":
                q = d.index(b"\0", q) + 1
            elif typ in "AcC":
                q += 1
            elif typ in "sS":
                q += 2
            elif typ in "iIf":
                q += 4
            elif typ == "B":
                sub, cnt = chr(d[q]), struct.unpack_from("<i", d, q + 1)[0]
                q += 5 + cnt * {"c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}[sub]
            else:
                raise ValueError(typ)
        recs.append(dict(ref=refs[ref_id] if ref_id >= 0 else "*", pos=pos, mapq=mapq, flag=flag, cigar=cigar,
                         seq=seq, qual=qual, md=md))
        off = end
    return header, recs


def sample_name(header):                   # src/BamProcess.cpp:273-287
    p = header.find("SM:")
    h = header[p + 3:]
    h = h[:h.find("\n")] if "\n" in h else h
    return h[:h.find("\t")] if "\t" in h else h


def end_pos(r):                            # bam_endpos: 0-based exclusive = 1-based inclusive
    return r["pos"] + sum(l for op, l in r["cigar"] if op in "MDN=X")


def get_offset(r, pos):                    # src/BamProcess.cpp:232-261
    offset = pos - (r["pos"] + 1)
    track = r["pos"]
    for op, l in r["cigar"]:
        if op not in "ISH":
            track += l
        if track < pos:
            if op in "IS":
                offset += l
            elif op in "DPN":
                offset -= l
        else:
            break
    if offset < 0 or offset > len(r["seq"]) - 1:
        raise IndexError("offset out of range")
    return offset


def find_snp_at_pos(rv, pv, refseq=None, rg_s=0):
    """src/BamProcess.cpp:4-94.  Returns {pos: AlleleInfo dict}; indel entries have is_indel = 1 and, when the
    region's reference string is given, the indel token the reference would write."""
    out = {}
    if not rv:
        return out
    i = j = 0
    r = rv[0]
    last = len(rv) - 1
    for pos in pv:
        if pos < r["pos"] + 1:
            continue
        eof = nxt = False
        while pos > end_pos(r):
            if i == last:
                eof = True
                break
            i += 1
            r = rv[i]
            j = i
            if pos < r["pos"] + 1:
                nxt = True
                break
        if nxt or eof:
            continue
        while True:
            c = r["cigar"]
            sx, sy, sk = r["pos"], 0, None
            for k, (op, l) in enumerate(c):
                if op in "MISX":
                    sy += l
                if op in "HI":
                    continue
                sx += l
                if pos <= sx:
                    sk = k
                    break
            assert sk is not None
            op = c[sk][0]
            indel, indel_str = 0, ""
            strand = 0 if (r["flag"] & 0x10 or r["flag"] & 0x20) else 1
            if sx == pos and sk + 1 < len(c):
                op2, l2 = c[sk + 1]
                if op2 == "D":
                    indel = -l2
                    indel_str = "-" + (refseq[pos - rg_s + 1:pos - rg_s + 1 + l2] if refseq is not None else "")
                elif op2 == "I":
                    indel = l2
                    indel_str = "+" + r["seq"][sy:sy + l2]
                elif op2 == "P" and sk + 2 < len(c):
                    # the reference re-reads c[sk] in this loop (src/BamProcess.cpp:60-62), so l3 stays 0
                    # unless c[sk] itself is an insertion, which cannot be the operation containing pos
                    indel = 0
            if indel != 0:
                out.setdefault(pos, dict(base=5, mapq=0, qual=r["mapq"], rpr=0, strand=strand, is_indel=1, indel=indel_str))
                break
            if op not in "DN":
                o = get_offset(r, pos)
                out.setdefault(pos, dict(base=BASE_CODE.get(r["seq"][o], 4), mapq=r["mapq"], qual=r["qual"][o], rpr=(o + 1) & 255,
                                         strand=strand, is_indel=0, indel=""))
                break
            elif j < last:
                j += 1
                r = rv[j]
                if pos < r["pos"] + 1 or pos > end_pos(r):
                    break
            else:
                break
        r = rv[i]
        j = i
    return out


def md_ref_bases(r, into):
    """Reference base at every reference position a read aligns to (M/=/X) or deletes, from MD:Z."""
    if r["md"] is None:
        return
    # expand MD into a per-reference-position list over the aligned+deleted span
    md, k, refspan = r["md"], 0, []
    while k < len(md):
        if md[k].isdigit():
            e = k
            while e < len(md) and md[e].isdigit():
                e += 1
            refspan += [None] * int(md[k:e])          # match: reference = read base
            k = e
        elif md[k] == "^":
            e = k + 1
            while e < len(md) and md[e].isalpha():
                e += 1
            refspan += list(md[k + 1:e])              # deleted reference bases
            k = e
        else:
            refspan.append(md[k])                     # mismatch: the reference base
            k += 1
    rpos, qpos, m = r["pos"] + 1, 0, 0               # 1-based reference position
    for op, l in r["cigar"]:
        if op in "M=X":
            for t in range(l):
                if m < len(refspan):
                    b = refspan[m] if refspan[m] is not None else r["seq"][qpos + t]
                    into.setdefault(rpos + t, b.upper())
                m += 1
            rpos += l
            qpos += l
        elif op == "D":
            for t in range(l):
                if m < len(refspan) and refspan[m] is not None:
                    into.setdefault(rpos + t, refspan[m].upper())
                m += 1
            rpos += l
        elif op == "N":
            rpos += l
        elif op in "IS":
            qpos += l


def load_samples(root, bam_list, region=REGION, mapq=MAPQ):
    """Reads every BAM of the list: (names, per-sample filtered record lists, reference bases known from MD tags)."""
    chrom, start, end = region
    rg_s, rg_e = start, end - 1                               # splitrg quirk, src/BaseVarUtils.h:71
    bams = [l.strip() for l in open(bam_list) if l.strip()]
    rvs, ref_at, names = [], {}, []
    for b in bams:
        header, recs = read_bam(os.path.join(root, b))
        assert "SO:coord" in header
        names.append(sample_name(header))
        rv = [r for r in recs if r["ref"] == chrom and not (r["flag"] & 0x4) and r["cigar"]
              and end_pos(r) > rg_s - 1000 - 1 and r["pos"] < rg_e + 1000
              and not (r["flag"] & 0x400) and r["mapq"] >= mapq]
        for r in rv:
            md_ref_bases(r, ref_at)
        rvs.append(rv)
    return names, rvs, ref_at


def region_reference(ref_at, rg_s, rg_e, buf=1000):
    """The string RefReader::GetTargetBase would return for [rg_s, rg_e + buf], with N where no read tells."""
    return "".join(ref_at.get(p, "N") if ref_at.get(p, "N") in BASE_CODE else "N" for p in range(rg_s, rg_e + buf + 1))


def main():
    from oracle import orc
    from tests.golden.golden_io import save_golden
    chrom, start, end = REGION
    rg_s, rg_e = start, end - 1
    pv = list(range(rg_s, rg_e + 1))
    names, rvs, ref_at = load_samples(REF_ROOT, os.path.join(REF_ROOT, "bam.list"))
    per_sample = [find_snp_at_pos(rv, pv) for rv in rvs]
    refseq = region_reference(ref_at, rg_s, rg_e)
    with open(os.path.join(os.path.dirname(__file__), "testdata", "chr17_41197700_region.txt"), "w") as f:
        f.write(f"chr17 {rg_s} 81195210\n{refseq}\n")       # contig, 1-based start, contig length / bases
    n_samples = len(names)
    min_af = min(100.0 / n_samples, 0.001)                   # src/BaseVarC.cpp:541-543 with default --maf
    sites, mafs, positions = [], [], []
    skipped_ref = 0
    for pos in pv:
        entries = [m[pos] for m in per_sample if pos in m]
        entries = [e for e in entries if e["is_indel"] or e["base"] != 4]   # N bases dropped (:427)
        if not entries:
            continue                                                     # `if (!aiv.empty())`, :442
        rb = ref_at.get(pos)
        if rb not in BASE_CODE:
            skipped_ref += 1
            continue
        obs = [e for e in entries if not e["is_indel"]]
        sites.append((np.array([e["base"] for e in obs], dtype=np.int8), np.array([e["qual"] for e in obs], dtype=np.int8),
                      BASE_CODE[rb]))
        mafs.append(min_af)
        positions.append(pos)
    exp = [orc.basetype_lrt(b, q, r, m) for (b, q, r), m in zip(sites, mafs)]
    save_golden("testdata_pileup.npz", sites, mafs, exp)
    # positions kept alongside (small)
    z = dict(np.load(os.path.join(os.path.dirname(__file__), "testdata_pileup.npz")))
    z["positions"] = np.array(positions, dtype=np.int32)
    np.savez_compressed(os.path.join(os.path.dirname(__file__), "testdata_pileup.npz"), **z)
    depth = np.array([len(b) for b, _, _ in sites])
    print(f"{len(sites)} sites from {n_samples} samples ({names[0]} ...), depth {depth.min()}..{depth.max()} "
          f"(mean {depth.mean():.1f}), {sum(e['called'] for e in exp)} called, "
          f"{skipped_ref} positions skipped for an unknown reference base")


if __name__ == "__main__":
    main()
