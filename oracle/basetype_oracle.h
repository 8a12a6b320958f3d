/*
 * oracle/basetype_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of BaseVarC's per-site basetype hot path, used only as the
 * checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * Nothing under basevarc_amd/ may include, link or call this.
 *
 * PARITY UNPINNED: the reference ships no golden vectors or unit tests for this
 * path and its hot-path translation units cannot be built
 * in this image without writing stand-in headers for the absent SeqLib/htslib
 * submodules (src/Algorithm.h:6 includes htslib/kfunc.h, src/BaseType.h:6
 * includes BamProcess.h -> SeqLib/BamReader.h), which the build rules forbid.
 * This oracle is therefore a source-text restatement, cross-checked against an
 * independent numpy restatement, analytic known answers and scipy.
 *
 * Every function cites the reference file:line (under /root/reference) it follows.
 */
#ifndef BASEVARC_ORACLE_H
#define BASEVARC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NTYPE 4                       /* src/BaseType.h:13 */
#define ORC_LRT_THRESHOLD 24.0            /* src/BaseType.h:9  */
#define ORC_MLN10TO10 -0.23025850929940458 /* src/BaseType.h:10 */

/* `mode` / `use_hist` arguments below are a bit set (0 = the faithful per-sample path): */
#define ORC_MODE_HIST        1   /* EM on the (base, qual) count histogram instead of per-sample rows */
#define ORC_MODE_COMPENSATED 2   /* NOT the reference's arithmetic: the sums over samples (M step, delta,
                                    log-likelihood) in long double, to expose the reference's own drift */

/* What BaseType exposes after LRT() (src/BaseType.h:70-74) plus diagnostics. */
typedef struct orc_result {
    int32_t called;        /* return value of BaseType::LRT() */
    int32_t n_alt;         /* alt_bases.size() */
    int8_t  alt_base[4];   /* alt_bases in reference order (<=3 used) */
    double  af[4];         /* af_lrt[alt_base[i]] */
    double  var_qual;
    double  chi;           /* chi_sqrt_t when LRT() left the loop */
    double  depth_total;
    int32_t depth[4];
    int32_t n_kept;        /* final `bases` after model reduction */
    int8_t  kept[4];
    double  base_frq[4];   /* base_frq (indexed by base 0..3) */
    double  lr_alt;        /* lr_alt_t at exit */
    int32_t n_fits;        /* EM() calls */
    int32_t n_passes;      /* singleEM() calls */
    int32_t status;        /* 0 ok; 1 = reference behaviour undefined (bp[0] on empty vector) */
    double  tie_gap;       /* diagnostic: smallest (runner-up chi - best chi) over the nested levels that had >= 2
                              fitted subsets (+inf when none had): which subset std::min_element picks is decided by
                              the last bits of the sums when this is at rounding level */
    /* Diagnostics of what a level NEED NOT RUN (not the reference's behaviour: it runs everything).  A level only goes on
     * with the first minimum of chi over its subsets (src/BaseType.cpp:99-105); the subset without the deepest candidate has
     * loglik <= U = sum, over the alleles outside it, of their observations' log(eps/3) (every marginal <= 1), so when
     * 2 (lr_alt - U) exceeds the minimum chi of the other subsets by more than 1 + 1e-6 |U| it cannot be that minimum (levels
     * of subsets of two alleles and more; the one-allele models of the last level cost nothing and always count).  The
     * library's item engine skips such fits (include/bvc.h "em_prune"); its n_fits / n_passes then equal these: */
    int32_t n_fits_pruned;
    int32_t n_passes_pruned;
    double  prune_edge;    /* smallest relative distance of such a test from its threshold (+inf: no test): a test this close
                              to rounding level may fall either way on another evaluation order */
    /* Facts about the INPUT (not results): which of the library's two stage-2 engines takes the site depends on them
     * (include/bvc.h "em_engine"), and the tests use them to demand the right pair of run counts from each engine. */
    int32_t max_quals;     /* most distinct base qualities among the observations of one base */
    int32_t min_qual;      /* lowest base quality of an observation (127 when there is none) */
    int32_t dup_candidate; /* 1 when a base passes the min_af filter twice (a SetBase list that repeats a base) */
} orc_result;

/* htslib kfunc.c restatement (third-party, absent from /root/reference). */
double orc_kf_lgamma(double z);
double orc_kf_gammaq(double s, double z);
double orc_chisf(double x, double k);               /* src/Algorithm.cpp:3-7 */

/* Faithful per-sample path: BaseType ctor + SetBase + LRT (src/BaseType.cpp:5-139). */
int orc_basetype_lrt(int32_t nind, const int8_t *bases, const int8_t *quals,
                     int8_t ref_base, double min_af,
                     const int8_t *base_comb, int32_t n_comb, orc_result *out);

int orc_basetype_lrt_mode(int32_t nind, const int8_t *bases, const int8_t *quals,
                          int8_t ref_base, double min_af,
                          const int8_t *base_comb, int32_t n_comb, int mode, orc_result *out);

/* Same control flow, EM run on a (base, qual) count histogram: counts[b*128+q].
 * Derived checker for full-size inputs; validated against orc_basetype_lrt in tests. */
int orc_hist_lrt(const uint32_t *counts512, int8_t ref_base, double min_af,
                 const int8_t *base_comb, int32_t n_comb, orc_result *out);

int orc_hist_lrt_mode(const uint32_t *counts512, int8_t ref_base, double min_af,
                      const int8_t *base_comb, int32_t n_comb, int mode, orc_result *out);

/* Dense tile helpers: row = site, uncovered sample = base byte outside 0..3 or qual < 0. */
int orc_dense_site(int64_t n_samples, const int8_t *bases_row, const int8_t *quals_row,
                   int8_t ref_base, double min_af, orc_result *out);
void orc_dense_hist(int64_t n_samples, const int8_t *bases_row, const int8_t *quals_row,
                    const uint8_t *group_of_sample, int32_t group, uint32_t *counts512);
/* Batch over sites with OpenMP (threads <= 0: all cores). Returns threads used. */
int orc_dense_batch(int64_t n_sites, int64_t n_samples, int64_t row_stride,
                    const int8_t *bases, const int8_t *quals, const int8_t *ref_base,
                    double min_af, int use_hist, int threads, orc_result *out);

/* Caller's --group loop (src/BaseVarC.cpp:617-661) for one site.
 * grp_depth[g*4+b], grp_af[g*3+i] (af of overall alt i, 0 when absent), grp_ran[g],
 * grp_present[g] bit i = overall alt i is among the group's alt bases. */
int orc_dense_site_groups(int64_t n_samples, const int8_t *bases_row, const int8_t *quals_row,
                          int8_t ref_base, double min_af,
                          const uint8_t *group_of_sample, int32_t n_groups, int use_hist,
                          orc_result *overall, int32_t *grp_depth, double *grp_af,
                          int32_t *grp_ran, int32_t *grp_present);

/* Counter-based synthetic pileup, integer-only, bit-identical to the device
 * generator in basevarc_amd/csrc. cov_thr16: sample covered iff r16 < cov_thr16 (65536 = dense). */
void orc_synth_site(uint64_t seed, int64_t site, int64_t n_samples, uint32_t cov_thr16,
                    int8_t *bases_row, int8_t *quals_row, int8_t *ref_base);

#ifdef __cplusplus
}
#endif
#endif
