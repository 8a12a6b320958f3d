/*
 * oracle/basetype_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see basetype_oracle.h).
 * PARITY UNPINNED (no reference golden vectors exist; reference TUs unbuildable here).
 *
 * Plain-C restatement of the reference's algorithm for the per-site basetype path:
 *   BaseType::BaseType      /root/reference/src/BaseType.cpp:5-23
 *   BaseType::SetAlleleFreq /root/reference/src/BaseType.cpp:25-39
 *   BaseType::UpdateF       /root/reference/src/BaseType.cpp:41-71
 *   BaseType::LRT           /root/reference/src/BaseType.cpp:73-139
 *   combs_                  /root/reference/src/BaseType.cpp:237-255
 *   singleEM / EM / delta   /root/reference/src/Algorithm.cpp:69-130
 *   chisf                   /root/reference/src/Algorithm.cpp:3-7  -> htslib kf_gammaq
 *   caller's group loop     /root/reference/src/BaseVarC.cpp:617-661
 * The per-sample path keeps the reference's operation order (sums run j = 0..3 then i = 0..n-1,
 * a fresh zeroed 4*n scratch per E/M pass) so that it is also a fair CPU timing baseline.
 *
 * Attribution: orc_kf_lgamma / orc_kf_gammaq (chisf) below restate, nearly statement for statement,
 * the numerical routines of htslib's kfunc.c (https://github.com/samtools/htslib, MIT/Expat licence, (c) Genome Research Ltd. and
 * Attractive Chaos), which the reference links through SeqLib (src/Algorithm.cpp:3-25; .gitmodules:1-3, submodule absent from the
 * reference tree).  The arithmetic has to be htslib's for the outputs to match the reference's; the constants are htslib's.
 *
 * ORC_MODE_COMPENSATED is NOT the reference's arithmetic: the same per-sample loops with the three
 * sums that run over all samples (M-step sums, delta, log-likelihood) accumulated in long double
 * (64-bit significand on x86-64), i.e. without the drift a double accumulator picks up over 1e6
 * nearly equal addends.  It exists to separate "the reference's own rounding drift" from "a
 * difference in the algorithm" when the GPU path is compared at N = 1e6.
 */
#include "basetype_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * chi-square survival function.  The reference calls kf_gammaq(k/2, x/2) from htslib's kfunc.c
 * (nested submodule of SeqLib, .gitmodules:1-3; directory empty in /root/reference, version
 * unpinned).  Restated here from the published algorithm: Lanczos-type log-gamma, power series
 * for the lower regularised function when z <= 1 or z < s, modified-Lentz continued fraction
 * for the upper one otherwise; both stop at 1e-14 or 100 terms.
 * ---------------------------------------------------------------------------------------- */
#define KF_EPS 1e-14
#define KF_TINY 1e-290

double orc_kf_lgamma(double z)
{
    static const double num[8] = { 676.5203681218835, -1259.139216722289, 771.3234287757674,
                                   -176.6150291498386, 12.50734324009056, -0.1385710331296526,
                                   0.9934937113930748e-05, 0.1659470187408462e-06 };
    double x = 0.0;
    int k;
    /* summed from the smallest term up, as upstream does */
    for (k = 7; k >= 0; --k) x += num[k] / (z + k);
    x += 0.9999999999995183;
    return log(x) - 5.58106146679532777 - z + (z - 0.5) * log(z + 6.5);
}

static double kf_lower_series(double s, double z)
{
    double term = 1.0, sum = 1.0;
    int k;
    for (k = 1; k < 100; ++k) {
        term *= z / (s + k);
        sum += term;
        if (term / sum < KF_EPS) break;
    }
    return exp(s * log(z) - z - orc_kf_lgamma(s + 1.0) + log(sum));
}

static double kf_upper_cf(double s, double z)
{
    double f = 1.0 + z - s, C = f, D = 0.0;
    int j;
    for (j = 1; j < 100; ++j) {
        double a = j * (s - j), b = (j << 1) + 1 + z - s, d;
        D = b + a * D;
        if (D < KF_TINY) D = KF_TINY;
        C = b + a / C;
        if (C < KF_TINY) C = KF_TINY;
        D = 1.0 / D;
        d = C * D;
        f *= d;
        if (fabs(d - 1.0) < KF_EPS) break;
    }
    return exp(s * log(z) - z - orc_kf_lgamma(s) - log(f));
}

double orc_kf_gammaq(double s, double z)
{
    return (z <= 1.0 || z < s) ? 1.0 - kf_lower_series(s, z) : kf_upper_cf(s, z);
}

/* src/Algorithm.cpp:3-7 */
double orc_chisf(double x, double k) { return orc_kf_gammaq(k / 2.0, x / 2.0); }

/* ------------------------------------------------------------------------------------------
 * A "site model": either per-sample likelihood rows (faithful) or (base,qual) classes.
 * ---------------------------------------------------------------------------------------- */
typedef struct site_model {
    int32_t n;            /* rows: samples (faithful) or non-empty classes (hist) */
    const double *L;      /* n*4 likelihoods, row-major i*4+j (src/BaseType.cpp:13,15) */
    const double *w;      /* NULL (faithful: weight 1) or class counts as doubles */
    double nsample;       /* divisor of the M step: nind (Algorithm.cpp:90) */
    int32_t depth[4];
    double depth_total;
    int32_t n_fits, n_passes;
    int compensated;      /* 0: the reference's double accumulators; 1: long double (see header) */
    double lle[4];        /* NOT the reference's: per base b, sum over its observations of log(eps/3) -- what they add to the
                             log-likelihood of a model without b (their marginal is eps/3 exactly); only for the pruned counts */
    uint8_t seen_q[4][128]; /* input facts for orc_result.max_quals / min_qual (diagnostics, not results) */
} site_model;

/* One E+M pass: src/Algorithm.cpp:69-93.  marginal[] and expect[] arrive zeroed. */
static void em_pass(site_model *sm, const double *freq, double *marginal, double *expect)
{
    const int32_t n = sm->n;
    double lik[ORC_NTYPE];
    /* the reference allocates and zeroes this 4*n scratch on every pass (Algorithm.cpp:71) */
    double *post = (double *)calloc((size_t)ORC_NTYPE * (size_t)(n > 0 ? n : 1), sizeof(double));
    int32_t i;
    int j;
    for (i = 0; i < n; ++i) {
        for (j = 0; j < ORC_NTYPE; ++j) {
            lik[j] = freq[j] * sm->L[(size_t)i * ORC_NTYPE + j];
            marginal[i] += lik[j];
        }
        for (j = 0; j < ORC_NTYPE; ++j) post[(size_t)j * n + i] = lik[j] / marginal[i];
    }
    for (j = 0; j < ORC_NTYPE; ++j) {
        if (sm->compensated) {
            long double acc = expect[j];
            if (sm->w) { for (i = 0; i < n; ++i) acc += (long double)sm->w[i] * post[(size_t)j * n + i]; }
            else { for (i = 0; i < n; ++i) acc += post[(size_t)j * n + i]; }
            expect[j] = (double)acc;
        } else if (sm->w) {
            for (i = 0; i < n; ++i) expect[j] += sm->w[i] * post[(size_t)j * n + i];
        } else {
            for (i = 0; i < n; ++i) expect[j] += post[(size_t)j * n + i];
        }
        expect[j] = expect[j] / sm->nsample;
    }
    free(post);
    sm->n_passes++;
}

/* EM driver: src/Algorithm.cpp:115-130 (update_allele_freq :95-101, delta_bylog :103-113). */
static void em_fit(site_model *sm, double *freq, double *marginal, double *expect,
                   int iter_num, double epsilon)
{
    const int32_t n = sm->n;
    double *next = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
    int it, j;
    int32_t i;
    em_pass(sm, freq, marginal, expect);
    for (it = 0; it < iter_num; ++it) {
        double delta = 0.0;
        long double delta_c = 0.0L;
        for (j = 0; j < ORC_NTYPE; ++j) { freq[j] = expect[j]; expect[j] = 0.0; }
        em_pass(sm, freq, next, expect);
        for (i = 0; i < n; ++i) {
            double d = fabs(log(next[i]) - log(marginal[i]));
            if (sm->compensated) delta_c += sm->w ? (long double)sm->w[i] * d : (long double)d;
            else delta += sm->w ? sm->w[i] * d : d;
            marginal[i] = next[i];
            next[i] = 0.0;
        }
        if (sm->compensated) delta = (double)delta_c;
        if (delta < epsilon) break;
    }
    free(next);
    sm->n_fits++;
}

/* k-subsets of positions 0..n-1 in lexicographic order: what combs_ (src/BaseType.cpp:237-255)
 * produces with prev_permutation on a k-ones mask.  Returns the number of subsets (<= 6). */
static int subsets_lex(int n, int k, int8_t out[6][4])
{
    int idx[4], cnt = 0, i;
    if (k > n || k <= 0) return 0;
    for (i = 0; i < k; ++i) idx[i] = i;
    for (;;) {
        for (i = 0; i < k; ++i) out[cnt][i] = (int8_t)idx[i];
        cnt++;
        i = k - 1;
        while (i >= 0 && idx[i] == n - k + i) --i;
        if (i < 0) break;
        idx[i]++;
        for (++i; i < k; ++i) idx[i] = idx[i - 1] + 1;
    }
    return cnt;
}

typedef struct fit_set {
    int n_comb;                 /* bc.size(): every subset, fitted or skipped */
    int8_t comb[6][4];          /* bc[c][0..k-1] as base codes */
    int n_fit;                  /* lr.size() == bp.size(): only the non-skipped ones */
    double lr[6];
    double bp[6][ORC_NTYPE];
    int32_t passes[6];          /* singleEM calls of each fitted subset (diagnostic) */
} fit_set;

/* BaseType::UpdateF, src/BaseType.cpp:41-71 (SetAlleleFreq :25-39 inlined). */
static void update_f(site_model *sm, const int8_t *bases, int n, int k, fit_set *fs,
                     double *marginal)
{
    int8_t pos[6][4];
    int c, t, j;
    int32_t i;
    fs->n_comb = subsets_lex(n, k, pos);
    fs->n_fit = 0;
    for (c = 0; c < fs->n_comb; ++c)
        for (t = 0; t < k; ++t) fs->comb[c][t] = bases[pos[c][t]];
    for (c = 0; c < fs->n_comb; ++c) {
        double freq[ORC_NTYPE] = { 0, 0, 0, 0 }, expect[ORC_NTYPE] = { 0, 0, 0, 0 };
        double freq_sum = 0.0, loglik = 0.0;
        long double loglik_c = 0.0L;
        int32_t depth_sum = 0;
        for (t = 0; t < k; ++t) depth_sum += sm->depth[fs->comb[c][t]];
        if (depth_sum > 0)
            for (t = 0; t < k; ++t)
                freq[fs->comb[c][t]] = (double)sm->depth[fs->comb[c][t]] / depth_sum;
        for (j = 0; j < ORC_NTYPE; ++j) freq_sum += freq[j];
        if (freq_sum == 0) continue;                      /* src/BaseType.cpp:54 */
        fs->passes[fs->n_fit] = -sm->n_passes;
        em_fit(sm, freq, marginal, expect, 100, 0.001);   /* src/BaseType.cpp:45-46,56 */
        fs->passes[fs->n_fit] += sm->n_passes;
        for (i = 0; i < sm->n; ++i) {
            double lm = log(marginal[i]);
            if (sm->compensated) loglik_c += sm->w ? (long double)sm->w[i] * lm : (long double)lm;
            else loglik += sm->w ? sm->w[i] * lm : lm;
            marginal[i] = 0.0;
        }
        if (sm->compensated) loglik = (double)loglik_c;
        fs->lr[fs->n_fit] = loglik;
        for (j = 0; j < ORC_NTYPE; ++j) fs->bp[fs->n_fit][j] = expect[j];
        fs->n_fit++;
    }
}

/* BaseType::LRT, src/BaseType.cpp:73-139, on a prepared site model. */
static int lrt_on_model(site_model *sm, int8_t ref_base, double min_af,
                        const int8_t *base_comb, int32_t n_comb, orc_result *out)
{
    static const int8_t default_comb[4] = { 0, 1, 2, 3 };   /* src/BaseType.h:79 */
    int8_t bases[8];
    int n = 0, k, j, c;
    double base_frq[ORC_NTYPE] = { 0, 0, 0, 0 };
    double lr_alt = 0.0, chi = 0.0;
    double *marginal;
    fit_set fs;
    int32_t skipped_fits = 0, skipped_passes = 0;

    memset(out, 0, sizeof(*out));
    out->tie_gap = HUGE_VAL;
    out->prune_edge = HUGE_VAL;
    for (j = 0; j < 4; ++j) out->depth[j] = sm->depth[j];
    out->depth_total = sm->depth_total;
    out->min_qual = 127;
    for (j = 0; j < 4; ++j) {                              /* input facts (diagnostics) */
        int q, nq = 0;
        for (q = 0; q < 128; ++q)
            if (sm->seen_q[j][q]) { nq++; if (q < out->min_qual) out->min_qual = q; }
        if (nq > out->max_quals) out->max_quals = nq;
    }
    if (sm->depth_total == 0) return 0;                    /* :75 */
    if (!base_comb) { base_comb = default_comb; n_comb = 4; }
    for (c = 0; c < n_comb && n < 8; ++c) {
        int8_t b = base_comb[c];
        if ((sm->depth[b] / sm->depth_total) >= min_af) {  /* :79 */
            for (j = 0; j < n; ++j) if (bases[j] == b) out->dup_candidate = 1;
            bases[n++] = b;
        }
    }
    if (n == 0) return 0;                                  /* :84 */
    if (n > 4) { out->status = 2; return 0; }              /* base_comb longer than 4: not a caller case */

    marginal = (double *)calloc((size_t)(sm->n > 0 ? sm->n : 1), sizeof(double));
    update_f(sm, bases, n, n, &fs, marginal);              /* :88 */
    if (fs.n_fit == 0) {   /* reference reads bp[0] of an empty vector (:89): undefined */
        out->status = 1;
        free(marginal);
        out->n_fits = out->n_fits_pruned = sm->n_fits; out->n_passes = out->n_passes_pruned = sm->n_passes;
        return 0;
    }
    for (j = 0; j < ORC_NTYPE; ++j) base_frq[j] = fs.bp[0][j];
    lr_alt = fs.lr[0];
    for (k = n - 1; k > 0; --k) {                          /* :93-110 */
        double best, chi_c[6];
        int i_min = 0;
        update_f(sm, bases, n, k, &fs, marginal);
        if (fs.n_fit == 0) { out->status = 1; break; }     /* min_element on empty range: undefined */
        for (c = 0; c < fs.n_fit; ++c) chi_c[c] = 2.0 * (lr_alt - fs.lr[c]);
        best = chi_c[0];
        for (c = 1; c < fs.n_fit; ++c)                     /* std::min_element: first minimum, '<' */
            if (chi_c[c] < best) { best = chi_c[c]; i_min = c; }
        for (c = 0; c < fs.n_fit; ++c)                     /* diagnostic: how close the runner-up was */
            if (c != i_min && chi_c[c] - best < out->tie_gap) out->tie_gap = chi_c[c] - best;
        /* Diagnostic, NOT the reference's: what a level need not have run (see orc_result.n_fits_pruned).  The k = n - 1
         * subset without the deepest candidate (first one on ties) is number n - 1 - p in lexicographic order. */
        if (fs.n_fit == fs.n_comb && fs.n_fit >= 2 && k >= 2) {   /* (one-allele levels need no EM at all: nothing to skip) */
            int p_deep = 0, c_last, t, bsel;
            double best_other = 0.0, u_c = 0.0, bound, slack;
            int have = 0;
            for (j = 1; j < n; ++j) if (sm->depth[bases[j]] > sm->depth[bases[p_deep]]) p_deep = j;
            c_last = n - 1 - p_deep;
            for (c = 0; c < fs.n_fit; ++c)
                if (c != c_last && (!have || chi_c[c] < best_other)) { best_other = chi_c[c]; have = 1; }
            for (bsel = 0; bsel < 4; ++bsel) {
                int inside = 0;
                for (t = 0; t < k; ++t) if (fs.comb[c_last][t] == bsel) inside = 1;
                if (!inside) u_c += sm->lle[bsel];
            }
            bound = 2.0 * (lr_alt - u_c);
            slack = 1.0 + 1e-6 * fabs(u_c);
            if (bound > best_other + slack && bound < HUGE_VAL) {
                skipped_fits += 1;
                skipped_passes += fs.passes[c_last];
            }
            if (bound == bound && best_other == best_other) {
                double edge = fabs(bound - (best_other + slack)) / (fabs(bound) > 1.0 ? fabs(bound) : 1.0);
                if (edge < out->prune_edge) out->prune_edge = edge;
            } else {
                out->prune_edge = 0.0;
            }
        }
        lr_alt = fs.lr[i_min];
        chi = chi_c[i_min];
        if (chi < ORC_LRT_THRESHOLD) {
            /* bc is indexed by i_min although lr/bp skip zero-coverage subsets (:54 vs :104) */
            for (j = 0; j < k; ++j) bases[j] = fs.comb[i_min][j];
            n = k;
            for (j = 0; j < ORC_NTYPE; ++j) base_frq[j] = fs.bp[i_min][j];
        } else {
            break;
        }
    }
    free(marginal);

    out->n_kept = n;
    for (j = 0; j < n; ++j) out->kept[j] = bases[j];
    for (j = 0; j < ORC_NTYPE; ++j) out->base_frq[j] = base_frq[j];
    out->lr_alt = lr_alt;
    out->chi = chi;
    out->n_fits = sm->n_fits;
    out->n_passes = sm->n_passes;
    out->n_fits_pruned = sm->n_fits - skipped_fits;
    out->n_passes_pruned = sm->n_passes - skipped_passes;
    for (j = 0; j < n; ++j) {                              /* :111-116 */
        if (bases[j] != ref_base && out->n_alt < 4) {
            out->alt_base[out->n_alt] = bases[j];
            out->af[out->n_alt] = base_frq[bases[j]];
            out->n_alt++;
        }
    }
    if (out->n_alt > 0) {                                  /* :118-135 */
        double r = sm->depth[bases[0]] / sm->depth_total;
        if (n == 1 && sm->depth_total > 10 && r > 0.5) {
            out->var_qual = 5000.0;
        } else {
            if (chi <= 0) {
                out->var_qual = 0.0;
            } else {
                double p = orc_chisf(chi, 1.0);
                if (p) out->var_qual = -10 * log10(p);     /* NaN is truthy, as in the reference */
                else out->var_qual = 10000;
                if (out->var_qual == 0) out->var_qual = 0.0;
            }
        }
        out->called = 1;
        return 1;
    }
    return 0;
}

/* BaseType constructor, src/BaseType.cpp:5-23: likelihood rows + depths. */
int orc_basetype_lrt(int32_t nind, const int8_t *bases, const int8_t *quals,
                     int8_t ref_base, double min_af,
                     const int8_t *base_comb, int32_t n_comb, orc_result *out)
{
    return orc_basetype_lrt_mode(nind, bases, quals, ref_base, min_af, base_comb, n_comb, 0, out);
}

int orc_basetype_lrt_mode(int32_t nind, const int8_t *bases, const int8_t *quals,
                          int8_t ref_base, double min_af,
                          const int8_t *base_comb, int32_t n_comb, int mode, orc_result *out)
{
    site_model sm;
    double *L = (double *)malloc(sizeof(double) * ORC_NTYPE * (size_t)(nind > 0 ? nind : 1));
    int32_t i;
    int j, rc;
    memset(&sm, 0, sizeof(sm));
    for (i = 0; i < nind; ++i) {
        if (bases[i] < 0 || bases[i] > 3) { free(L); memset(out, 0, sizeof(*out)); out->status = 3; return 0; }
        for (j = 0; j < ORC_NTYPE; ++j) {
            /* exp() is evaluated inside the j loop, exactly like :13,15 */
            if (bases[i] == j) L[(size_t)i * 4 + j] = 1.0 - exp(ORC_MLN10TO10 * quals[i]);
            else L[(size_t)i * 4 + j] = exp(ORC_MLN10TO10 * quals[i]) / 3.0;
        }
        sm.depth[bases[i]] += 1;
        sm.lle[bases[i]] += log(exp(ORC_MLN10TO10 * quals[i]) / 3.0);
        if (quals[i] >= 0) sm.seen_q[bases[i]][quals[i]] = 1;
    }
    for (j = 0; j < 4; ++j) sm.depth_total += sm.depth[j];
    sm.n = nind; sm.L = L; sm.w = NULL; sm.nsample = nind;
    sm.compensated = (mode & ORC_MODE_COMPENSATED) != 0;
    rc = lrt_on_model(&sm, ref_base, min_af, base_comb, n_comb, out);
    free(L);
    return rc;
}

/* Histogram form: each non-empty (base, qual) class is one weighted row. */
int orc_hist_lrt(const uint32_t *counts512, int8_t ref_base, double min_af,
                 const int8_t *base_comb, int32_t n_comb, orc_result *out)
{
    return orc_hist_lrt_mode(counts512, ref_base, min_af, base_comb, n_comb, 0, out);
}

int orc_hist_lrt_mode(const uint32_t *counts512, int8_t ref_base, double min_af,
                      const int8_t *base_comb, int32_t n_comb, int mode, orc_result *out)
{
    site_model sm;
    double L[512 * 4], w[512];
    int b, q, j, n = 0, rc;
    int64_t total = 0;
    memset(&sm, 0, sizeof(sm));
    for (b = 0; b < 4; ++b)
        for (q = 0; q < 128; ++q) {
            uint32_t c = counts512[b * 128 + q];
            if (!c) continue;
            for (j = 0; j < 4; ++j) {
                if (b == j) L[n * 4 + j] = 1.0 - exp(ORC_MLN10TO10 * (int8_t)q);
                else L[n * 4 + j] = exp(ORC_MLN10TO10 * (int8_t)q) / 3.0;
            }
            w[n] = (double)c;
            sm.seen_q[b][q] = 1;
            sm.lle[b] += (double)c * log(exp(ORC_MLN10TO10 * (int8_t)q) / 3.0);
            sm.depth[b] += (int32_t)c;
            total += c;
            n++;
        }
    for (j = 0; j < 4; ++j) sm.depth_total += sm.depth[j];
    sm.n = n; sm.L = L; sm.w = w; sm.nsample = (double)(int32_t)total;
    sm.compensated = (mode & ORC_MODE_COMPENSATED) != 0;
    rc = lrt_on_model(&sm, ref_base, min_af, base_comb, n_comb, out);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * Dense [site][sample] rows: covered samples are those with base in 0..3 and qual >= 0
 * (the reference's caller drops N bases and indels before BaseType sees them:
 * src/BaseVarC.cpp:427, 551-559).
 * ---------------------------------------------------------------------------------------- */
static int covered(int8_t b, int8_t q) { return b >= 0 && b <= 3 && q >= 0; }

void orc_dense_hist(int64_t n_samples, const int8_t *bases_row, const int8_t *quals_row,
                    const uint8_t *group_of_sample, int32_t group, uint32_t *counts512)
{
    int64_t i;
    memset(counts512, 0, 512 * sizeof(uint32_t));
    for (i = 0; i < n_samples; ++i) {
        if (!covered(bases_row[i], quals_row[i])) continue;
        if (group_of_sample && group >= 0 && group_of_sample[i] != (uint8_t)group) continue;
        counts512[bases_row[i] * 128 + quals_row[i]]++;
    }
}

static int dense_site_subset(int64_t n_samples, const int8_t *bases_row, const int8_t *quals_row,
                             const uint8_t *group_of_sample, int32_t group,
                             int8_t ref_base, double min_af, const int8_t *base_comb,
                             int32_t n_comb, int mode, orc_result *out)
{
    if (mode & ORC_MODE_HIST) {
        uint32_t counts[512];
        orc_dense_hist(n_samples, bases_row, quals_row, group_of_sample, group, counts);
        return orc_hist_lrt_mode(counts, ref_base, min_af, base_comb, n_comb, mode, out);
    } else {
        int8_t *b = (int8_t *)malloc((size_t)(n_samples > 0 ? n_samples : 1));
        int8_t *q = (int8_t *)malloc((size_t)(n_samples > 0 ? n_samples : 1));
        int64_t i;
        int32_t n = 0;
        int rc;
        for (i = 0; i < n_samples; ++i) {
            if (!covered(bases_row[i], quals_row[i])) continue;
            if (group_of_sample && group >= 0 && group_of_sample[i] != (uint8_t)group) continue;
            b[n] = bases_row[i]; q[n] = quals_row[i]; n++;
        }
        rc = orc_basetype_lrt_mode(n, b, q, ref_base, min_af, base_comb, n_comb, mode, out);
        free(b); free(q);
        return rc;
    }
}

int orc_dense_site(int64_t n_samples, const int8_t *bases_row, const int8_t *quals_row,
                   int8_t ref_base, double min_af, orc_result *out)
{
    return dense_site_subset(n_samples, bases_row, quals_row, NULL, -1, ref_base, min_af,
                             NULL, 0, 0, out);
}

int orc_dense_batch(int64_t n_sites, int64_t n_samples, int64_t row_stride,
                    const int8_t *bases, const int8_t *quals, const int8_t *ref_base,
                    double min_af, int use_hist, int threads, orc_result *out)
{
    int used = 1;
    int64_t s;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
    used = threads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#else
    (void)threads;
#endif
    for (s = 0; s < n_sites; ++s)
        dense_site_subset(n_samples, bases + s * row_stride, quals + s * row_stride, NULL, -1,
                          ref_base[s], min_af, NULL, 0, use_hist, &out[s]);
    return used;
}

/* Caller's per-group loop: src/BaseVarC.cpp:617-661.  Groups are numbered in the caller's
 * std::map (name-sorted) order; samples with group id >= n_groups belong to no group (:352-356). */
int orc_dense_site_groups(int64_t n_samples, const int8_t *bases_row, const int8_t *quals_row,
                          int8_t ref_base, double min_af,
                          const uint8_t *group_of_sample, int32_t n_groups, int use_hist,
                          orc_result *overall, int32_t *grp_depth, double *grp_af,
                          int32_t *grp_ran, int32_t *grp_present)
{
    int8_t comb[4];
    int32_t n_comb, g;
    int i, t;
    int ok = dense_site_subset(n_samples, bases_row, quals_row, NULL, -1, ref_base, min_af,
                               NULL, 0, use_hist, overall);
    comb[0] = ref_base;                                    /* :614-615 */
    n_comb = 1;
    for (i = 0; i < overall->n_alt && n_comb < 4; ++i) comb[n_comb++] = overall->alt_base[i];
    for (g = 0; g < n_groups; ++g) {
        uint32_t counts[512];
        int64_t depth = 0;
        orc_dense_hist(n_samples, bases_row, quals_row, group_of_sample, g, counts);
        for (i = 0; i < 4; ++i) {
            int32_t d = 0;
            for (t = 0; t < 128; ++t) d += (int32_t)counts[i * 128 + t];
            grp_depth[g * 4 + i] = d;                      /* na:nc:ng:nt, :640 */
            depth += d;
        }
        for (i = 0; i < 3; ++i) grp_af[g * 3 + i] = 0.0;
        grp_ran[g] = 0;
        grp_present[g] = 0;
        if (ok && depth > 0) {                             /* :633-636, :641 */
            orc_result gr;
            dense_site_subset(n_samples, bases_row, quals_row, group_of_sample, g, ref_base,
                              min_af, comb, n_comb, use_hist, &gr);   /* :642-644 */
            grp_ran[g] = 1;
            for (i = 0; i < overall->n_alt && i < 3; ++i)  /* :646-652 */
                for (t = 0; t < gr.n_alt; ++t)
                    if (gr.alt_base[t] == overall->alt_base[i]) {     /* gr_bt.af_lrt.count(b), :647 */
                        grp_af[g * 3 + i] = gr.af[t];
                        grp_present[g] |= 1 << i;
                    }
        }
    }
    return ok;
}

/* ------------------------------------------------------------------------------------------
 * Synthetic pileup generator, integer arithmetic only so that host and device
 * agree bit for bit.  Site parameters come from the site hash; sample draws from two more.
 * ---------------------------------------------------------------------------------------- */
static uint64_t mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}

/* round(AF * 2^32) for 64 log-spaced allele frequencies 1e-4 .. 0.5 (AF_k = 1e-4 * 5000^(k/63)). */
#include "synth_tables.inc"

void orc_synth_site(uint64_t seed, int64_t site, int64_t n_samples, uint32_t cov_thr16,
                    int8_t *bases_row, int8_t *quals_row, int8_t *ref_base)
{
    const uint64_t hs = mix64(seed * 0x9E3779B97F4A7C15ULL + (uint64_t)site + 0x632BE59BD9B4E019ULL);
    const uint32_t ref = (uint32_t)(hs & 3);
    const uint32_t alt = (ref + 1 + (uint32_t)((hs >> 2) & 0xFFFF) % 3) & 3;
    const uint32_t alt2 = (alt == ((ref + 1) & 3)) ? ((ref + 2) & 3) : ((ref + 1) & 3);
    const int poly = ((hs >> 20) & 0xFFFF) % 100 < 20;          /* 20 % polymorphic */
    const int second = ((hs >> 36) & 0xFFFF) % 100 < 2;         /* 2 % of those: second ALT */
    const uint32_t thr1 = poly ? SYNTH_AF_THR[(hs >> 52) & 63] : 0;
    const uint32_t thr2 = (poly && second) ? thr1 / 4 : 0;
    const uint64_t hs2 = mix64(hs ^ 0xD1B54A32D192ED03ULL);
    int64_t i;
    *ref_base = (int8_t)ref;
    for (i = 0; i < n_samples; ++i) {
        const uint64_t h1 = mix64(hs2 + (uint64_t)i * 0x9E3779B97F4A7C15ULL);
        const uint64_t h2 = mix64(h1 + 0x9E3779B97F4A7C15ULL);
        const uint32_t r_allele = (uint32_t)h1;
        const uint32_t q = 10 + (uint32_t)((((h1 >> 32) & 0xFFFF) * 31) >> 16);   /* 10..40 */
        const uint32_t r_err = (uint32_t)h2;
        const uint32_t r_sub = (uint32_t)((((h2 >> 32) & 0xFFFF) * 3) >> 16);     /* 0..2 */
        const uint32_t r_cov = (uint32_t)(h2 >> 48);
        uint32_t b = ref;
        if (r_allele < thr1) b = alt;
        else if (r_allele - thr1 < thr2) b = alt2;
        if (r_err < SYNTH_ERR_THR[q]) b = (b + 1 + r_sub) & 3;
        if (r_cov < cov_thr16) { bases_row[i] = (int8_t)b; quals_row[i] = (int8_t)q; }
        else { bases_row[i] = (int8_t)-1; quals_row[i] = 0; }
    }
}
