"""Python restatement of the text formats and annotation statistics either side of the basetype path.

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (the reference binary cannot be built, no expected outputs exist).
Follows /root/reference: bt_r writer src/BaseVarC.cpp:509-527, bt_s parser :403-441, bt_f CVG line :548-610 and
group columns :617-663, WriteVcf src/BaseType.cpp:141-234, RankSumTest / bt_fisher_exact / normsf
src/Algorithm.cpp:9-67; kt_fisher_exact and kf_erfc are htslib kfunc.c algorithms (absent from the tree).
Written independently of basevarc_amd/host/*.cpp so that the two check each other.
"""
import math

BASE2CHAR = "ACGTNN"
MLN10TO10 = -0.23025850929940458


def fx(v, p):
    if math.isnan(v):
        return "nan"
    if math.isinf(v):
        return "-inf" if v < 0 else "inf"
    return f"{v:.{p}f}"


# ------------------------------------------------------------------------------------------- statistics
def _lbinom(n, k):
    if k == 0 or n == k:
        return 0.0
    return math.lgamma(n + 1) - math.lgamma(k + 1) - math.lgamma(n - k + 1)


def _hypergeo(n11, n1_, n_1, n):
    return math.exp(_lbinom(n1_, n11) + _lbinom(n - n1_, n_1 - n11) - _lbinom(n, n_1))


class _Acc:
    def __init__(self):
        self.n11 = self.n1_ = self.n_1 = self.n = 0
        self.p = 0.0

    def step(self, n11, n1_=0, n_1=0, n=0):
        if n1_ or n_1 or n:
            self.n11, self.n1_, self.n_1, self.n = n11, n1_, n_1, n
        else:
            if n11 % 11 and n11 + self.n - self.n1_ - self.n_1:
                if n11 == self.n11 + 1:
                    self.p *= (self.n1_ - self.n11) / n11 * (self.n_1 - self.n11) / (n11 + self.n - self.n1_ - self.n_1)
                    self.n11 = n11
                    return self.p
                if n11 == self.n11 - 1:
                    self.p *= self.n11 / (self.n1_ - n11) * (self.n11 + self.n - self.n1_ - self.n_1) / (self.n_1 - n11)
                    self.n11 = n11
                    return self.p
            self.n11 = n11
        self.p = _hypergeo(self.n11, self.n1_, self.n_1, self.n)
        return self.p


def fisher_two_sided(n11, n12, n21, n22):
    n1_, n_1, n = n11 + n12, n11 + n21, n11 + n12 + n21 + n22
    mx = min(n_1, n1_)
    mn = max(0, n1_ + n_1 - n)
    if mn == mx:
        return 1.0
    a = _Acc()
    q = a.step(n11, n1_, n_1, n)
    p = a.step(mn)
    left, i = 0.0, mn + 1
    while p < 0.99999999 * q and i <= mx:
        left += p
        p = a.step(i)
        i += 1
    i -= 1
    if p < 1.00000001 * q:
        left += p
    p = a.step(mx)
    right, j = 0.0, mx - 1
    while p < 0.99999999 * q and j >= 0:
        right += p
        p = a.step(j)
        j -= 1
    if p < 1.00000001 * q:
        right += p
    return min(1.0, left + right)


def bt_fisher_exact(n11, n12, n21, n22):
    two = fisher_two_sided(n11, n12, n21, n22)
    if two == 0:
        return 10000.0
    p = -10 * math.log10(two)
    return 0.0 if p == 0 else p


def normsf(x):
    return math.erfc(x / math.sqrt(2.0)) / 2.0          # kf_erfc is erfc to ~1e-15


def rank_sum_test(x, y):
    n1, n2 = len(x), len(y)
    v = list(x) + list(y)
    idx = sorted(range(len(v)), key=lambda i: -v[i])     # stable, descending (std::sort's order among ties does not
    s = len(idx)                                         # matter: tied elements share the average rank)
    r1, k, n = 0.0, 0, 0
    for i in range(s):
        if i + 1 < s and v[idx[i]] == v[idx[i + 1]]:
            k += i + 1
            n += 1
        elif k > 0:
            k += i + 1
            avg = k / (n + 1)
            for j in range(i - n, i + 1):
                if idx[j] < n1:
                    r1 += avg
            k = n = 0
        elif idx[i] < n1:
            r1 += i + 1
    expected = (n1 * (n1 + n2 + 1)) / 2.0
    den = math.sqrt((n1 * n2 * (n1 + n2 + 1)) / 12.0)
    if den == 0:
        z = float("nan") if r1 - expected == 0 else math.copysign(float("inf"), r1 - expected)
    else:
        z = (r1 - expected) / den
    if math.isnan(z):
        return float("nan")
    sf = 2 * normsf(abs(z))
    if sf == 0:
        return 10000.0
    p = -10 * math.log10(sf)
    return 0.0 if p == 0 else p


# ------------------------------------------------------------------------------------------- pileup text
def format_token(a):
    if a is None:
        return ". "
    if a["is_indel"]:
        return a["indel"] + " "
    return f"{a['base']},{a['mapq']},{a['qual']},{a['rpr']},{a['strand']} "


class Parser:
    """bt_s's token parser with its one long-lived AlleleInfo (indel tokens inherit the last base token's fields)."""

    def __init__(self):
        self.ai = dict(base=0, mapq=0, qual=0, rpr=0, strand=0, is_indel=0, indel="")

    def parse(self, lines):
        aiv, sample, j = [], [], 0
        for line in lines:
            for tok in line.split(" "):
                if tok == "" or tok == "\n":
                    continue
                c = tok[0]
                if c not in "+-N.":
                    f = [int(x) for x in tok.split(",")[:5]]
                    keys = ["base", "mapq", "qual", "rpr", "strand"]
                    for kk, val in zip(keys, f):
                        self.ai[kk] = val
                    self.ai["base"] &= 7
                    self.ai["strand"] &= 1
                    for kk in ("mapq", "qual", "rpr"):
                        self.ai[kk] &= 255
                    self.ai["is_indel"] = 0
                    if self.ai["base"] != 4:
                        aiv.append(dict(self.ai, indel=""))
                        sample.append(j)
                elif c != ".":
                    self.ai["is_indel"] = 1
                    self.ai["indel"] = tok
                    aiv.append(dict(self.ai))
                    sample.append(j)
                j += 1
        return aiv, sample


# ------------------------------------------------------------------------------------------- CVG / VCF lines
def cvg_line(chr_, pos, ref_base, aiv, grp_depths=None):
    cnt = [0, 0, 0, 0]
    indel_m = {}
    for a in aiv:
        if a["is_indel"] == 0:
            if a["base"] < 4:
                cnt[a["base"]] += 1
        else:
            indel_m[a["indel"]] = indel_m.get(a["indel"], 0) + 1
    indels = ",".join(f"{k}|{v}" for k, v in sorted(indel_m.items())) if indel_m else "."
    didx = sorted(range(4), key=lambda i: -cnt[i])
    alt_base = didx[0] if didx[0] != ref_base else didx[1]
    rf = rr = af = ar = 0
    for a in aiv:
        if a["strand"] == 1:
            if a["base"] == ref_base:
                rf += 1
            elif a["base"] == alt_base:
                af += 1
        else:
            if a["base"] == ref_base:
                rr += 1
            elif a["base"] == alt_base:
                ar += 1
    fs = bt_fisher_exact(rf, rr, af, ar)
    sor = (rf * ar) / (rr * af) if af * rr > 0 else 10000.0
    out = f"{chr_}\t{pos}\t{BASE2CHAR[ref_base]}\t{sum(cnt)}\t{cnt[0]}\t{cnt[1]}\t{cnt[2]}\t{cnt[3]}\t{indels}\t{fx(fs, 3)}\t{fx(sor, 3)}\t{rf},{rr},{af},{ar}\t"
    for d in grp_depths or []:
        out += f"{d[0]}:{d[1]}:{d[2]}:{d[3]}\t"
    return out[:-1] + "\n"


def vcf_line(bt, chr_, pos, ref_base, aiv, sample, n_samples, info=None):
    """bt: oracle dict (alt_base, af, depth, depth_total, var_qual)."""
    info = dict(info or {})
    alt_gt = {b: f"./{i + 1}" for i, b in enumerate(bt["alt_base"])}
    ent = dict(zip(sample, aiv))
    rq, rm, rp, aq, am, ap = [], [], [], [], [], []
    rf = rr = af = ar = 0
    cols = []
    for i in range(n_samples):
        a = ent.get(i)
        if a is None:
            cols.append("./.")
            continue
        alt_gt.setdefault(a["base"], "./.")
        gt = "0/." if a["base"] == ref_base else alt_gt[a["base"]]
        cols.append(f"{gt}:{BASE2CHAR[min(a['base'], 5)]}:{'-+'[a['strand']]}:{1 - math.exp(MLN10TO10 * a['qual']):.6f}")
        if a["is_indel"] == 1 or a["base"] == 4:
            continue
        is_alt = a["base"] in bt["alt_base"]
        if a["base"] == ref_base:
            rq.append(a["qual"]); rm.append(a["mapq"]); rp.append(a["rpr"])
        elif is_alt:
            aq.append(a["qual"]); am.append(a["mapq"]); ap.append(a["rpr"])
        if a["strand"] == 1:
            if a["base"] == ref_base:
                rf += 1
            elif is_alt:
                af += 1
        else:
            if a["base"] == ref_base:
                rr += 1
            elif is_alt:
                ar += 1
    fs = bt_fisher_exact(rf, rr, af, ar)
    sor = (rf * ar) / (rr * af) if af * rr > 0 else 10000.0
    ad_sum = sum(bt["depth"][b] for b in bt["alt_base"])
    new = {
        "CM_AC": ",".join(str(bt["depth"][b]) for b in bt["alt_base"]),
        "CM_AF": ",".join(fx(f, 6) for f in bt["af"]),
        "CM_CAF": ",".join(fx(bt["depth"][b] / bt["depth_total"], 6) for b in bt["alt_base"]),
        "QD": fx(bt["var_qual"] / ad_sum if ad_sum else (float("inf") if bt["var_qual"] > 0 else float("nan")), 3),
        "CM_DP": fx(bt["depth_total"], 0),
        "MQRankSum": fx(rank_sum_test(rm, am), 3),
        "ReadPosRankSum": fx(rank_sum_test(rp, ap), 3),
        "BaseQRankSum": fx(rank_sum_test(rq, aq), 3),
        "FS": fx(fs, 3), "SOR": fx(sor, 3), "SB_REF": f"{rf},{rr}", "SB_ALT": f"{af},{ar}",
    }
    for k, v in new.items():
        info.setdefault(k, v)
    qt = "." if bt["var_qual"] > 60 else "LowQual"
    alt = ",".join(BASE2CHAR[b] for b in bt["alt_base"])
    out = f"{chr_}\t{pos}\t.\t{BASE2CHAR[ref_base]}\t{alt}\t{fx(bt['var_qual'], 2)}\t{qt}\t"
    out += ";".join(f"{k}={info[k]}" for k in sorted(info))
    out += "\tGT:AB:SO:BP\t" + "\t".join(cols) + "\n"
    return out
