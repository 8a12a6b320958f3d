"""Python pipeline for the reference's `BaseVarC basetype` run on its own test data (test infrastructure):
BAM -> per-sample pileup -> temp-batch text -> parse -> BaseType (oracle) -> CVG/VCF text.  Used to check the
C++ host program (basevarc_amd/host) file by file.  Everything here restates reference source text
(see oracle/emit_oracle.py and tests/golden/make_testdata_pileup.py); parity is unpinned."""
import os

from oracle import emit_oracle as eo
from tests.golden import make_testdata_pileup as tp

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "golden", "testdata")
REGION = "chr17:41197700-41276155"


def region_fixture():
    head, seq = open(os.path.join(DATA, "chr17_41197700_region.txt")).read().split("\n")[:2]
    contig, start, length = head.split()
    return contig, int(start), int(length), seq


def write_fasta(path):
    """A full-length single-contig FASTA (N outside the fixture region) + .fai, so that chr17 coordinates work."""
    contig, start, length, seq = region_fixture()
    with open(path, "w") as f:
        f.write(f">{contig}\n")
        f.write("N" * (start - 1))
        f.write(seq)
        f.write("N" * (length - (start - 1) - len(seq)))
        f.write("\n")
    with open(path + ".fai", "w") as f:
        f.write(f"{contig}\t{length}\t{len(contig) + 2}\t{length}\t{length + 1}\n")
    return path


def write_bam_list(path):
    with open(path, "w") as f:
        for l in open(os.path.join(DATA, "bam.list")):
            if l.strip():
                f.write(os.path.join(DATA, l.strip().replace("data/", "", 1)) + "\n")
    return path


def thread_window(psize, thread, ithread):
    window = psize % thread + psize // thread
    lo = min(psize, ithread * window)
    hi = psize if ithread == thread - 1 else min(psize, (ithread + 1) * window)
    return lo, hi


class Pipeline:
    def __init__(self, mapq=20, batch=10, thread=1, maf=0.001):
        self.batch, self.thread = batch, thread
        contig, start, _, refseq = region_fixture()
        self.chr, self.refseq = contig, refseq
        chrom, s, e = tp.REGION
        self.rg_s, self.rg_e = s, e - 1
        self.pv = [self.rg_s + i for i in range(len(refseq) - 1000) if refseq[i] in "ACGT"]
        lst = os.path.join(DATA, "bam.list")
        names, rvs, _ = tp.load_samples(DATA, self._list_for_loader(lst), mapq=mapq)
        self.names = names
        self.maps = [tp.find_snp_at_pos(rv, self.pv, refseq, self.rg_s) for rv in rvs]
        self.n = len(names)
        self.min_af = min(100.0 / self.n, 0.001, maf)

    @staticmethod
    def _list_for_loader(lst):
        import tempfile
        t = tempfile.NamedTemporaryFile("w", suffix=".list", delete=False)
        for l in open(lst):
            if l.strip():
                t.write(l.strip().replace("data/", "", 1) + "\n")
        t.close()
        return t.name

    def batch_files(self):
        """{(ithread, ibatch): text} as bt_r writes them."""
        nb = 1 + (self.n - 1) // self.batch
        out = {}
        psize = len(self.pv)
        for ib in range(nb):
            ms = self.maps[ib * self.batch:(ib + 1) * self.batch]
            names = "".join(n + "\t" for n in self.names[ib * self.batch:(ib + 1) * self.batch]) + "\n"
            for t in range(self.thread):
                lo, hi = thread_window(psize, self.thread, t)
                lines = [names]
                for p in self.pv[lo:hi]:
                    lines.append("".join(eo.format_token(m.get(p)) for m in ms) + "\n")
                out[(t, ib)] = "".join(lines)
        return out

    def outputs(self, group_of=None):
        """(vcf_text, cvg_text) bodies of the merged outputs.  group_of: {sample name: group name} or None."""
        import numpy as np
        from oracle import orc
        gnames, gidx = [], None
        if group_of:
            gnames = sorted(set(group_of[n] for n in self.names if n in group_of))
            gidx = np.array([gnames.index(group_of[n]) if n in group_of else 255 for n in self.names], dtype=np.uint8)
        files = self.batch_files()
        nb = 1 + (self.n - 1) // self.batch
        vcf, cvg = [], []
        contig, _, length, _ = region_fixture()
        for t in range(self.thread):
            P = eo.Parser()
            per_batch = [files[(t, ib)].split("\n")[1:] for ib in range(nb)]
            lo, hi = thread_window(len(self.pv), self.thread, t)
            for k, p in enumerate(self.pv[lo:hi]):
                aiv, sample = P.parse([pb[k] for pb in per_batch])
                if not aiv:
                    continue
                ref = "ACGT".index(self.refseq[p - self.rg_s])
                b = [a["base"] for a in aiv if not a["is_indel"]]
                q = [a["qual"] for a in aiv if not a["is_indel"]]
                e = orc.basetype_lrt(b, q, ref, self.min_af)
                gd, info = None, {}
                if gnames:
                    db = np.full(self.n, -1, dtype=np.int8)
                    dq = np.zeros(self.n, dtype=np.int8)
                    for a, j in zip(aiv, sample):
                        if not a["is_indel"]:
                            db[j], dq[j] = a["base"], a["qual"]
                    _, gdep, gaf, ran, pres = orc.dense_site_groups(db, dq, ref, self.min_af, gidx, len(gnames))
                    gd = gdep.tolist()
                    for g, name in enumerate(gnames):            # src/BaseVarC.cpp:646-659
                        if ran[g]:
                            info[name + "_AF"] = ",".join(eo.fx(gaf[g][i], 6) if pres[g] >> i & 1 else "0"
                                                          for i in range(e["n_alt"]))
                        else:
                            info[name + "_AF"] = "0"
                cvg.append(eo.cvg_line(self.chr, p, ref, aiv, gd))
                if e["called"]:
                    vcf.append(eo.vcf_line(e, self.chr, p, ref, aiv, sample, self.n, info))
        return "".join(vcf), "".join(cvg)


def headers(reference_path, names, group_names=()):
    contig, _, length, _ = region_fixture()
    # CVG_HEADER / VCF_HEADER, src/BaseVarC.cpp:65-88; assembled as bt_s does (:364-382)
    from tests.hostref_headers import CVG_HEADER, VCF_HEADER
    vh = VCF_HEADER
    for g in group_names:
        vh += (f"##INFO=<ID={g}_AF,Number=A,Type=Float,Description=\"Allele frequency in the {g} populations "
               "calculated based on LRT.[0,1]\">\n")
    vh += f"##contig=<ID={contig},length={length}>\n" + f"##reference=file://{reference_path}\n"
    vh += "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names) + "\n"
    return vh, CVG_HEADER + "".join("\t" + g for g in group_names) + "\n"
