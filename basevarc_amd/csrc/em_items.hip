// em_items.hip -- stage 2 of the basetype path on gfx950, "item engine": the EM fits of a call are the work items,
// eight of them share a wavefront, and a TEAM of two wavefronts takes a REGION of 8 sites through the whole likelihood-ratio
// test: classes -> fits of the first level -> decisions -> fits of the next level -> ... with every intermediate
// (class tables, fit descriptors and results, site state) in the region's own 11 KB of LDS.  The team meets at an arrival
// counter in that LDS between the phases; no workgroup barrier after kernel entry (round 3 gave a region to a workgroup of
// four wavefronts that met at s_barrier after every phase: about half of the resident wavefronts were waiting at any time);
// the teams of a workgroup are independent and share nothing but the launch.
//
// Follows (paths under /root/reference), with the per-sample sums regrouped by class as in em_kernel.hip:
//   BaseType::SetAlleleFreq  src/BaseType.cpp:25-39     -> region_emit
//   BaseType::UpdateF        src/BaseType.cpp:41-71     -> one FitItem per subset: fit_body (EM), site_decide (log-likelihood)
//   BaseType::LRT            src/BaseType.cpp:73-139    -> site_classes (:75-88), site_decide (one level of :93-110 per phase)
//   combs_                   src/BaseType.cpp:237-255   -> subset_masks
//   singleEM / EM / delta    src/Algorithm.cpp:69-130   -> fit_body
//
// Why: with one wavefront per site (em_kernel.hip) 39 of the 63 vector instructions of an EM pass are cross-lane
// reduction and control, paid once per fit-pass.  Here a fit ("item") owns 2 lanes per allele: lane (unit u, half h)
// keeps 16 of the <= 32 quality classes of one allele in registers, so
//   * D_b (the allele's own sum) needs one DPP step, E one swap per level of the 2x2 row tree, shared by the 8 (four
//     alleles per item) or 16 (two alleles per item) items of the wavefront;
//   * alleles outside the fitted subset have f = 0: their classes have marginal e, add exactly their depth to E and
//     nothing to the stop rule, so they take no lanes at all -- the nested levels with two alleles
//     (src/BaseType.cpp:93-110, k = 2 and 1) run 16 items per wavefront;
//   * the stop rule needs no per-class work: m' - m = (f' - f) d for every class of an allele, hence
//        sum_c n_c |m'_c / m_c - 1| = sum_b |f'_b - f_b| * D_b(previous pass)          (d > 0: quality >= 2)
//     and delta = sum_c n_c |log m'_c - log m_c| < 1e-3 is decided from that bracket exactly as in em_kernel.hip
//     (delta itself is evaluated only when the bracket straddles 1e-3);
//   * 16 independent classes per lane hide the FP64 latency that one dependency chain per wavefront exposes;
//   * a = 1 - 3e, so the class marginal f a + (1 - f) e is f + (1 - 4 f) e: one FMA on e alone, and with
//     Y = sum n / m the allele's own sum is D = Y - 4 E_own -- a lane keeps n and e per class, nothing else;
//   * the 16 reciprocals of a lane come from ONE v_rcp_f64 (16 issue cycles, four FMAs' worth): 1 / (m_0 ... m_15)
//     refined once, times the other fifteen marginals by a product tree, 2.8 multiplications per class.
// The items of a wavefront run in lockstep passes (each with its own pass counter), the wavefronts of a region meet
// at a team barrier after every phase, and the hardware dispatches the regions dynamically.  Fits differ 7x in passes at
// N = 1e4, so what shares a wavefront matters: the fits of one site converge alike, which is why a site's items are placed
// together.  The one exception -- the subset that leaves out the site's deepest allele always creeps to the iteration cap --
// is not run at all when a bound shows that it cannot be its level's minimum (site_decide), and one-allele models need no EM.
// Sites the engine does not take (more than 32 quality values on one allele, a class of quality 0 or 1 -- d < 0 --,
// min_af <= 0, duplicate candidates) stay with em_kernel.hip's one-wavefront-per-site kernels, flagged per site.
// FP64 throughout; no MFMA (nothing here is a dense contraction).
#include <algorithm>

#include "bvc_device.h"
#include "bvc_internal.h"

namespace bvc {
namespace {

// -DBVC_CHECK_LDS (bvc_device.h): every LDS index the region kernels derive from LDS contents goes through BVC_LDS_OK, which
// records a violation for the host (bvc_debug_report) and lets the code take a harmless path.  Check ids 11..16.

constexpr double kLrtThreshold = 24.0;    // LRT_THRESHOLD, src/BaseType.h:9
constexpr int kEmIters = 100;             // src/BaseType.cpp:46
constexpr double kEmEpsilon = 0.001;      // src/BaseType.cpp:45
constexpr double kVarQualPending = -1.0;  // as in em_kernel.hip: var_qual_kernel finishes these records

// Quality classes of one allele the engine holds: CPB = 32 (two lanes x 16, three waves per SIMD) for the regions whose
// sites all fit; CPB = 48 (two lanes x 24, two waves per SIMD) for the regions with a wider site -- Illumina's unbinned 41
// values; CPB = 8 (ONE lane per allele: 16 four-allele fits per wavefront) for the regions whose sites all have at most
// 8 -- binned qualities (NovaSeq: 4).  A site's table (LDS): counts uint32 [base][class], then quality indices uint8
// [base][class]: 5 * 4 * CPB bytes.
constexpr int kTiny = 8, kNarrow = 32, kWide = 48;
template <int CPB> constexpr int site_table_bytes() { return 4 * CPB * 5; }

constexpr int kEmptyQ = 128;               // quality index of an empty class place: QualLut::e_empty = 1/4
constexpr uint32_t hi_word(double x) { return (uint32_t)(__builtin_bit_cast(uint64_t, x) >> 32); }
constexpr uint32_t kSureBelowHi = hi_word(kEmEpsilon / (1.0 + 0.00390625));        // hi(A) <  this: converged
constexpr uint32_t kSureAboveHi = hi_word(kEmEpsilon / (1.0 - 0.00390625)) + 1u;   // hi(A) >= this: not converged

// One fit: the alleles of a subset in candidate order ("units"), their starting frequencies, and what the alleles
// outside the subset add to E (a constant of the fit).  64 bytes.
struct FitItem {
    int32_t site;           // site of the region (0 .. kRegionSites - 1)
    uint8_t base[4];        // allele of unit u; 0xFF = no such unit
    double f0[4];           // SetAlleleFreq: depth / depth of the subset
    double e_excl;          // sum of the depths of the alleles outside the subset
    double inv_n;           // 1 / depth_total
    double ll_excl;         // what their classes add to the log-likelihood: sum n log e (their marginal is e)
};
static_assert(sizeof(FitItem) == 64, "FitItem layout");

struct FitOut {
    double ex[4];           // expect_allele_prob of the last pass, per unit
    double fl[4];           // the frequencies that pass ran on, per unit (kept here, not in registers, until the epilogue)
    double ll;              // log-likelihood at those frequencies (UpdateF, src/BaseType.cpp:58-62)
    int32_t passes;
    int32_t pad;
};
static_assert(sizeof(FitOut) == 80, "FitOut layout");

// Per-site state between the phases.
struct ItemSite {
    double lr_alt, chi;
    double base_frq[4];
    double lle[4];          // per allele: sum over its classes of n log e
    double lla[4];          // per allele: sum over its classes of n log a (a = 1 - 3 e: the likelihood of the allele itself)
    double best_chi, best_lr, best_bp[4];   // second round of a level: the first minimum among the subsets of the first round
    int32_t depth[4];
    int32_t passes, fits;
    int32_t item[4];        // pending items: index into the region's item / result arrays
    uint32_t sets;          // fits to emit next: 4-bit position masks over blist, packed
    uint32_t blist;         // candidates in order, 4 bits each
    int8_t n_emit, p_deepest;
    int8_t n;               // candidates
    int8_t k;               // subset size of the pending level
    uint8_t state;          // 0 = left to em_kernel.hip, 1 = running here, 2 = record written
    uint8_t first;          // the pending level is the first one: full model + its (n-1)-subsets
    uint8_t later;          // position mask of the level's LAST-RESORT subset (the one without the deepest candidate), 0 = none
    uint8_t round2;         // the pending fit is that subset: the level's other subsets have been run and decided among
    int8_t best_i;          // lexicographic index of the first minimum so far (round2)
};

// list 0: items of 3-4 units (four rows per item); list 1: items of 1-2 units (two rows per item)
constexpr int kLists = 2;
// Engine rounds per site: the two levels of the reference that need EM (the full model with its (n-1)-subsets, and the
// (n-2)-subsets when n = 4; one-allele models need none), each at most twice (the last-resort subset of a level, when the bound
// cannot rule it out: see site_decide)
constexpr int kRounds = 4;
#ifndef BVC_REGION_SITES
#define BVC_REGION_SITES 8
#endif
// sites of a region.  (Round 3, when every subset was run: 8 x (full model + 4 subsets) = 32 + 8 fits; 6 sites left the slow
// wavefronts a quarter empty, 12 are too few regions per launch: profiles/r03_region_sites.txt.)
constexpr int kRegionSites = BVC_REGION_SITES;
// places of the region's item arrays: a site has at most FOUR fits pending in any round (the full model and the three
// (n-1)-subsets that keep the deepest candidate); the lists of a round lie one behind the other
constexpr int kPlaces = 4 * kRegionSites;

// k-subsets of positions 0..n-1 in lexicographic order (what combs_ yields), as 4-bit position masks packed
// least-significant first; count returned through `cnt`.
__device__ __forceinline__ uint32_t subset_masks(int n, int k, int &cnt)
{
    switch (n * 8 + k) {
    case 1 * 8 + 1: cnt = 1; return 0x1u;
    case 2 * 8 + 2: cnt = 1; return 0x3u;
    case 2 * 8 + 1: cnt = 2; return 0x21u;
    case 3 * 8 + 3: cnt = 1; return 0x7u;
    case 3 * 8 + 2: cnt = 3; return 0x653u;
    case 3 * 8 + 1: cnt = 3; return 0x421u;
    case 4 * 8 + 4: cnt = 1; return 0xFu;
    case 4 * 8 + 3: cnt = 4; return 0xEDB7u;
    case 4 * 8 + 2: cnt = 6; return 0xCA6953u;
    case 4 * 8 + 1: cnt = 4; return 0x8421u;
    default: cnt = 0; return 0u;
    }
}

__device__ __forceinline__ int pick4i(const int32_t (&v)[4], int j)
{
    return j == 0 ? v[0] : (j == 1 ? v[1] : (j == 2 ? v[2] : v[3]));
}

// Log-likelihood of the model "allele b alone" (f_b = 1): every observation's marginal is its own likelihood of b, a for the
// observations of b and e for the others -- no EM needed (site_decide).
__device__ __forceinline__ double single_allele_loglik(const double (&lla)[4], const double (&lle)[4], int b)
{
    double ll = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) ll += j == b ? lla[j] : lle[j];
    return ll;
}

// Histogram of pseudo-site p: plain calls have one histogram per site; group calls run one pseudo-site per
// (site, group) on the per-group histograms [site][n_groups + 1][512].
// (32-bit division: a call has fewer than 2^31 / 64 sites, bvc_api.hip check_common, and at most 32 groups)
__device__ __forceinline__ int64_t hist_index(int64_t p, int n_groups)
{
    if (n_groups <= 0) return p;
    const uint32_t site = (uint32_t)p / (uint32_t)n_groups;
    return (int64_t)site * (n_groups + 1) + ((uint32_t)p - site * (uint32_t)n_groups);
}

// Site whose reference base pseudo-site p compares its alleles with.
__device__ __forceinline__ int64_t ref_index(int64_t p, int n_groups)
{
    return n_groups > 0 ? (int64_t)((uint32_t)p / (uint32_t)n_groups) : p;
}

// List of a fit: four rows per item for 3-4 alleles, two for 1-2.
__device__ __forceinline__ int list_of(uint32_t pm) { return __popc(pm) >= 3 ? 0 : 1; }

// `masks` (4-bit masks packed least-significant first) without its c-th one.
__device__ __forceinline__ uint32_t drop_nibble(uint32_t masks, int c)
{
    const uint32_t low = masks & ((1u << (4 * c)) - 1u);
    return low | ((masks >> (4 * c + 4)) << (4 * c));
}

// Position (in the candidate list) of the deepest candidate, first one on ties.
__device__ __forceinline__ int deepest_position(const int32_t (&depth)[4], uint32_t blist, int n)
{
    int best = 0, best_depth = -1;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int dp = pick4i(depth, (int)((blist >> (4 * p)) & 3u));
        if (p < n && dp > best_depth) { best_depth = dp; best = p; }
    }
    return best;
}

__device__ __forceinline__ void count_wanted(uint32_t sets, int n_emit, int (&want)[kLists])
{
    want[0] = want[1] = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (c < n_emit) {
            const int l = list_of((sets >> (4 * c)) & 0xFu);
            want[0] += l == 0; want[1] += l == 1;
        }
}

// The (n-1)-subsets of n candidates that a level runs first, and the one it keeps for later (`later`): the subset without the
// deepest candidate (position p_deepest) is, of the n subsets in lexicographic order, number n - 1 - p_deepest.
__device__ __forceinline__ uint32_t level_subsets(int n, int p_deepest, int &n_first, uint32_t &later)
{
    int cnt = 0;
    const uint32_t masks = subset_masks(n, n - 1, cnt);
    const int c_last = n - 1 - p_deepest;
    later = (masks >> (4 * c_last)) & 0xFu;
    n_first = cnt - 1;
    return drop_nibble(masks, c_last);
}

// What a region keeps in LDS.
template <int CPB>
struct Region {
    ItemSite site[kRegionSites];
    FitItem items[kPlaces];
    FitOut outs[kPlaces];
    int want[kRegionSites][kLists];
    int need;                                            // most class places a taken site of the region needs on an allele
    int next_slot;                                       // wavefront-slots of the running level handed out so far (teams of 2 or 4)
    uint32_t arrive;                                     // team barrier: arrivals so far (never reset)
    alignas(8) uint8_t tab[kRegionSites][site_table_bytes<CPB>()];
};

// Two narrow regions fit the 32 KiB of LDS that two 64 KiB histogram workgroups leave on a CU (launch_lrt_items), four of any
// kind one workgroup's dynamic LDS.
static_assert(kRegionSites != 8 || (2 * sizeof(Region<kNarrow>) <= 32 * 1024 && 2 * sizeof(Region<kWide>) <= 32 * 1024), "stage 2 beside two histogram workgroups");
static_assert(kRegionSites != 8 || 4 * sizeof(Region<kWide>) <= 64 * 1024, "a workgroup's regions");
static_assert(4 * kRegionSites <= kWave, "region_emit: one lane per fit of the region");

// Emits the fits of the pending round for the whole region at once: lane 4 * ls + c builds fit c of site ls (its subset's alleles,
// SetAlleleFreq's starting frequencies -- four f64 divisions --, what the alleles outside the subset add to E and to the
// log-likelihood) and writes it to its place: list after list, each in site order, a site's fits in the order of its subsets.
// (Round 3 did this site by site with every lane computing the same fit: 20 divisions in a row per site.)
template <class RegionT>
__device__ __forceinline__ void region_emit(RegionT &R, int lane, const int (&first)[kLists])
{
    // (opaque, as in fit_body: what depends on the lane id only is then recomputed at every level instead of being hoisted out
    // of the level loop and kept -- spilled, at the narrow kernel's 168 VGPRs -- across the fits)
    asm volatile("" : "+v"(lane));
    const int ls = lane >> 2, c = lane & 3;
    if (ls >= kRegionSites) return;
    ItemSite &S = R.site[ls];
    const uint32_t sets = S.sets;
    const int n_emit = S.n_emit;
    if (S.state != 1 || c >= n_emit) return;
    const uint32_t pm = (sets >> (4 * c)) & 0xFu;
    const int l = list_of(pm);
    // place: the list's first place + the fits of the sites before this one in that list + this site's earlier fits in it
    int idx = l == 0 ? first[0] : first[1];
    for (int w = 0; w < ls; ++w) idx += R.want[w][l];
#pragma unroll
    for (int c2 = 0; c2 < 3; ++c2)
        if (c2 < c) idx += list_of((sets >> (4 * c2)) & 0xFu) == l;
    const int depth[4] = {S.depth[0], S.depth[1], S.depth[2], S.depth[3]};
    const int total_i = depth[0] + depth[1] + depth[2] + depth[3];
    const uint32_t blist = S.blist;
    if (!BVC_LDS_OK(11, idx, kPlaces)) return;
    FitItem &fi = R.items[idx];                                  // written field by field, straight to LDS
    S.item[c] = idx;
    fi.site = ls;
    int depth_sum = 0, u = 0;
    uint32_t bases = 0xFFFFFFFFu, in_set = 0;
#pragma unroll
    for (int p = 0; p < 4; ++p)
        if ((pm >> p) & 1u) {
            const int b = (blist >> (4 * p)) & 3u;
            bases = (bases & ~(0xFFu << (8 * u))) | ((uint32_t)b << (8 * u));
            in_set |= 1u << b;
            depth_sum += pick4i(depth, b);
            ++u;
        }
    double ll_excl = 0.0;
#pragma unroll
    for (int b = 0; b < 4; ++b) ll_excl += ((in_set >> b) & 1u) ? 0.0 : S.lle[b];
    fi.ll_excl = ll_excl;
    fi.e_excl = (double)(total_i - depth_sum);
    fi.inv_n = 1.0 / (double)total_i;
    *reinterpret_cast<uint32_t *>(fi.base) = bases;
#pragma unroll
    for (int q = 0; q < 4; ++q) {                                // SetAlleleFreq (:25-39)
        const int b = (bases >> (8 * q)) & 0xFFu;
        fi.f0[q] = q < u ? (double)pick4i(depth, b & 3) / (double)depth_sum : 0.0;
    }
}

__device__ __forceinline__ void store_record(bvc_site_result *dst, const ItemSite &S, int ref, int n, uint32_t blist)
{
    bvc_site_result r;
    const int total_i = S.depth[0] + S.depth[1] + S.depth[2] + S.depth[3];
    const double depth_total = (double)total_i;
    r.var_qual = 0.0; r.chi = S.chi; r.depth_total = depth_total; r.lr_alt = S.lr_alt;
    int n_alt = 0;
    int a0 = 0, a1 = 0, a2 = 0;
    double g0 = 0.0, g1 = 0.0, g2 = 0.0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {                                // src/BaseType.cpp:111-116
        const int b = (blist >> (4 * p)) & 3;
        if (p < n && b != ref && n_alt < 3) {
            const double fr = b == 0 ? S.base_frq[0] : (b == 1 ? S.base_frq[1] : (b == 2 ? S.base_frq[2] : S.base_frq[3]));
            if (n_alt == 0) { a0 = b; g0 = fr; } else if (n_alt == 1) { a1 = b; g1 = fr; } else { a2 = b; g2 = fr; }
            ++n_alt;
        }
    }
    r.alt_base[0] = (int8_t)a0; r.alt_base[1] = (int8_t)a1; r.alt_base[2] = (int8_t)a2;
    r.af[0] = g0; r.af[1] = g1; r.af[2] = g2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r.base_frq[j] = S.base_frq[j];
        r.depth[j] = S.depth[j];
        r.kept[j] = (j < n) ? (int8_t)((blist >> (4 * j)) & 3u) : 0;
    }
    r.n_passes = S.passes; r.n_alt = (uint8_t)n_alt; r.called = 0;
    r.n_kept = (uint8_t)n; r.status = 0; r.n_fits = (uint8_t)S.fits;
    if (n_alt > 0) {                                             // src/BaseType.cpp:117-135
        const double rr = (double)pick4i(S.depth, (int)(blist & 3u)) / depth_total;
        if (n == 1 && depth_total > 10 && rr > 0.5) r.var_qual = 5000.0;
        else if (S.chi <= 0) r.var_qual = 0.0;
        else r.var_qual = kVarQualPending;                       // chisf(chi, 1): finished by var_qual_kernel
        r.called = 1;
    }
    *dst = r;
}

// ---- site_classes: the wavefront, one site at a time (its 512 class counts already in registers: cnt8) --------------
// Compacts the non-empty classes of each allele (ascending quality) into the site's table: class c of allele b at
// [b][c], its count and its quality (e = eps / 3 comes from the context's table; empty places: n = 0 and the index of
// e = 1/4, whose marginal f + (1 - 4 f) e is 1/4 whatever f: weight 0 in every sum and harmless in the product of a
// lane's marginals).  Then the head of BaseType::LRT (src/BaseType.cpp:75-88): candidates by min_af, and the first
// level's fits: the full model and, because they depend on nothing but the candidate list, its (n-1)-subsets.
template <int CPB>
__device__ __forceinline__ int site_classes(Region<CPB> &R, int ls, int lane, int64_t site, const uint32_t (&cnt8)[8],
                                            const QualLut *__restrict__ lut, double min_af,
                                            const int8_t *__restrict__ comb, const uint8_t *__restrict__ n_comb,
                                            uint8_t *__restrict__ taken)
{
    const int row = lane >> 4, t = lane & 15;
    ItemSite S{};
    uint32_t sets = 0;
    int n_emit = 0, p_deepest = 0;
    bool finished = false;
    uint32_t *tab_n = reinterpret_cast<uint32_t *>(R.tab[ls]) + row * CPB;
    uint8_t *tab_q = R.tab[ls] + 16 * CPB + row * CPB;
    int cnt_row = 0, depth_lane = 0;
    double lle = 0.0, lla = 0.0;
    bool low_q = false;
#pragma unroll
    for (int lvl = 0; lvl < 8; ++lvl) {
        const int q = t + 16 * lvl;
        const uint32_t c = cnt8[lvl];
        const uint64_t nzmask = __ballot(c != 0);
        const uint32_t rowbits = (uint32_t)(nzmask >> (16 * row)) & 0xFFFFu;
        if (c != 0) {
            const int pos = cnt_row + __popc(rowbits & ((1u << t) - 1u));
            if (pos < CPB) {
                tab_n[pos] = c;
                tab_q[pos] = (uint8_t)q;
            }
            lle = fma((double)c, lut->log_e[q], lle);
            lla = fma((double)c, lut->log_a[q], lla);
            low_q |= q < 2;
        }
        cnt_row += __popc(rowbits);
        depth_lane += (int)c;
    }
    // the places this allele leaves empty
#pragma unroll
    for (int part = 0; part < (CPB + 15) / 16; ++part) {
        const int pos = t + 16 * part;
        if (pos >= cnt_row && pos < CPB) {
            tab_n[pos] = 0u;
            tab_q[pos] = (uint8_t)kEmptyQ;
        }
    }
    const int depth_row = row_sum(depth_lane);
    lle = row_sum(lle);
    lla = row_sum(lla);
    S.lle[0] = lane_value<0>(lle); S.lle[1] = lane_value<16>(lle); S.lle[2] = lane_value<32>(lle); S.lle[3] = lane_value<48>(lle);
    S.lla[0] = lane_value<0>(lla); S.lla[1] = lane_value<16>(lla); S.lla[2] = lane_value<32>(lla); S.lla[3] = lane_value<48>(lla);
    S.depth[0] = __builtin_amdgcn_readlane(depth_row, 0);
    S.depth[1] = __builtin_amdgcn_readlane(depth_row, 16);
    S.depth[2] = __builtin_amdgcn_readlane(depth_row, 32);
    S.depth[3] = __builtin_amdgcn_readlane(depth_row, 48);
    const bool too_wide = __ballot(cnt_row > kWide) != 0;
    int need = cnt_row;                                          // most class places on an allele of this site
    need = max(need, __shfl_xor(need, 16, kWave));
    need = max(need, __shfl_xor(need, 32, kWave));
    const bool any_low = __ballot(low_q) != 0;

    const int total_i = S.depth[0] + S.depth[1] + S.depth[2] + S.depth[3];
    const double depth_total = (double)total_i;
    uint32_t blist = 0;
    int n = 0;
    bool dup = false;
    if (total_i > 0) {                                           // src/BaseType.cpp:75
        uint32_t list = 0x3210u;                                 // default base_comb, src/BaseType.h:79
        int nc = 4;
        if (comb) {
            const int want_c = min((int)n_comb[site], 4);
            list = 0; nc = 0;
            for (int c = 0; c < want_c; ++c) {
                const int b = comb[site * 4 + c];
                if ((unsigned)b < 4u) { list |= (uint32_t)b << (4 * nc); ++nc; }
            }
        }
        uint32_t seen = 0;
        for (int c = 0; c < nc; ++c) {                           // src/BaseType.cpp:77-83
            const int b = (list >> (4 * c)) & 3;
            if ((double)pick4i(S.depth, b) / depth_total >= min_af) {
                dup |= ((seen >> b) & 1u) != 0;
                seen |= 1u << b;
                blist |= (uint32_t)b << (4 * n);
                ++n;
            }
        }
    }
    const bool mine = !(too_wide || any_low || dup);
    S.blist = blist; S.n = (int8_t)n; S.k = (int8_t)n; S.first = 1;
    S.state = mine ? 1 : 0;
    uint32_t later = 0;
    if (mine) {
        if (n == 0) {
            finished = true;                                     // :75, :84
        } else if (n == 1) {
            // one candidate: the full model is "that allele alone" (:88-90) and there is no nested level.  Its EM needs no
            // arithmetic: f = 1, every posterior is 1, expect = nind / nind = 1 exactly, delta = 0 after the second pass.
            const int b0 = (int)(blist & 3u);
            S.lr_alt = single_allele_loglik(S.lla, S.lle, b0);
            S.base_frq[0] = b0 == 0 ? 1.0 : 0.0; S.base_frq[1] = b0 == 1 ? 1.0 : 0.0;
            S.base_frq[2] = b0 == 2 ? 1.0 : 0.0; S.base_frq[3] = b0 == 3 ? 1.0 : 0.0;
            S.passes = 2; S.fits = 1;
            S.k = 0; S.first = 0;
            finished = true;
        } else {
            sets = (1u << n) - 1u;                               // the full model (:88)
            n_emit = 1;
            if (n >= 3) {
                // ... and, with it, the (n-1)-subsets that keep the deepest candidate; the one without it waits (site_decide).
                // (n = 2: the subsets are one-allele models, which site_decide evaluates without EM)
                int n_first = 0;
                p_deepest = deepest_position(S.depth, blist, n);
                sets |= level_subsets(n, p_deepest, n_first, later) << 4;
                n_emit += n_first;
            }
        }
    }
    S.sets = sets; S.n_emit = (int8_t)n_emit; S.p_deepest = (int8_t)p_deepest; S.later = (uint8_t)later;
    int want[kLists];
    count_wanted(sets, n_emit, want);
    if (lane == 0) {
        if (finished) S.state = 3;                               // record pending: written by the kernel that owns the region
        R.site[ls] = S;
        R.want[ls][0] = want[0]; R.want[ls][1] = want[1];
        taken[site] = mine ? 1 : 0;
    }
    return mine ? need : 0;                                      // most class places on an allele of a site the engine takes
}

// ---- the fits: 64 / (ROWS * G) items per wavefront -----------------------------------------------------------------
// An item takes ROWS DPP rows x G lanes: unit (allele) = row (ROWS = 4) or row & 1 (ROWS = 2: the row pairs (0,1) and
// (2,3) hold items of their own); the G = 2^LOG2G lanes of a unit split the allele's 32 class places, 32 / G each.

// 1 / x: v_rcp_f64 (about 2^-26) and one third-order step, y (1 + t + t^2) with t = 1 - x y: residual t^3.
__device__ __forceinline__ double rcp_cubic(double x)
{
    const double y = __builtin_amdgcn_rcp(x);
    const double t = fma(-x, y, 1.0);
    return fma(y, fma(t, t, t), y);
}

// y[i] = 1 / m[i] for N positive values from one reciprocal: products up a binary tree, 1 / (m_0 ... m_{N-1}), and
// down again each node's reciprocal = parent's reciprocal x sibling.  3 (N - 1) multiplications + one rcp_cubic.
template <int N>
__device__ __forceinline__ void rcp_all(const double (&m)[N], double (&y)[N])
{
    if constexpr (N == 1) {
        y[0] = rcp_cubic(m[0]);
    } else if constexpr ((N & (N - 1)) != 0) {
        // not a power of two (24 = 16 + 8, 12 = 8 + 4): a tree and a reciprocal for each part
        constexpr int P = N >= 32 ? 32 : (N >= 16 ? 16 : (N >= 8 ? 8 : (N >= 4 ? 4 : 2)));
        double a[P], ya[P], b[N - P], yb[N - P];
#pragma unroll
        for (int i = 0; i < P; ++i) a[i] = m[i];
#pragma unroll
        for (int i = 0; i < N - P; ++i) b[i] = m[P + i];
        rcp_all<P>(a, ya);
        rcp_all<N - P>(b, yb);
#pragma unroll
        for (int i = 0; i < P; ++i) y[i] = ya[i];
#pragma unroll
        for (int i = 0; i < N - P; ++i) y[P + i] = yb[i];
    } else {
        double p[N / 2], q[N / 2];
#pragma unroll
        for (int i = 0; i < N / 2; ++i) p[i] = m[2 * i] * m[2 * i + 1];
        rcp_all<N / 2>(p, q);
#pragma unroll
        for (int i = 0; i < N / 2; ++i) {
            y[2 * i] = q[i] * m[2 * i + 1];
            y[2 * i + 1] = q[i] * m[2 * i];
        }
    }
}

// Sum over the G lanes of a unit, result in each of them.
template <int LOG2G>
__device__ __forceinline__ double unit_sum(double v)
{
    if (LOG2G >= 1) v += dpp_f64<kDppXor1>(v);
    if (LOG2G >= 2) v += dpp_f64<kDppXor2>(v);
    if (LOG2G >= 3) v += dpp_f64<kDppHalfMirror>(v);
    return v;
}

// Sum over the lanes of an item of x, and of y, both delivered to every lane of the item.
template <int ROWS, int LOG2G>
__device__ __forceinline__ void item_sum2(double &x, double &y)
{
    double v;
    if (ROWS == 4) {
        const DPair h = swap32(x, y);          // lower half: x over rows (r, r + 2); upper half: y
        v = h.a + h.b;
        const DPair g = swap16(v, v);          // + the neighbouring row
        v = g.a + g.b;
    } else {
        const DPair g = swap16(x, y);          // even rows: x over the row pair; odd rows: y
        v = g.a + g.b;
    }
    v = unit_sum<LOG2G>(v);
    const DPair b = ROWS == 4 ? swap32(v, v) : swap16(v, v);
    x = b.a;
    y = b.b;
}

template <int ROWS, int LOG2G>
__device__ __forceinline__ double item_sum(double x)
{
    if (ROWS == 4) {
        const DPair h = swap32(x, x);
        x = h.a + h.b;
    }
    const DPair g = swap16(x, x);
    x = g.a + g.b;
    return unit_sum<LOG2G>(x);
}

template <int ROWS, int LOG2G, int CPB>
__device__ __forceinline__ void fit_body(int item0, int item_end, const FitItem *items, FitOut *outs, const uint8_t *cls,
                                         const double *lut_e)
{
    constexpr int G = 1 << LOG2G, kSlots = CPB / G;
    constexpr int kGroupsPerRow = 16 / G;
    // (opaque to the optimiser: what is derived from the lane id below is then recomputed in every slot -- a handful of
    // integer instructions -- instead of being kept in registers across the levels of the region, which the narrow kernel's
    // 168 VGPRs have no room for)
    int tid = (int)threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & (kWave - 1);
    const int row = lane >> 4, sub = lane & (G - 1), grp = (lane & 15) >> LOG2G;
    const int unit = ROWS == 4 ? row : (row & 1);
    const int item = item0 + (ROWS == 4 ? grp : (row >> 1) * kGroupsPerRow + grp);
    const bool valid = item < item_end && BVC_LDS_OK(12, item, kPlaces);

    double n[kSlots], e[kSlots];
    double fb = 0.0;
    bool active = false;
    // the fit's two constants are read from its descriptor in every pass (two LDS loads that have the whole pass to arrive)
    // instead of living in four registers: the narrow kernel sits at the 168 VGPRs of three wavefronts per SIMD
    const FitItem *fi_c = items + (valid ? item : item0);
    if (valid) {
        const FitItem *fi = items + item;
        const int base = fi->base[unit];
        if (base != 0xFF && BVC_LDS_OK(13, fi->site, kRegionSites) && BVC_LDS_OK(14, base, 4)) {
            active = true;
            fb = fi->f0[unit];
            const uint8_t *tab = cls + fi->site * site_table_bytes<CPB>();
            const uint32_t *tab_n = reinterpret_cast<const uint32_t *>(tab) + base * CPB + sub;
            const uint8_t *tab_q = tab + 16 * CPB + base * CPB + sub;
#pragma unroll
            for (int k = 0; k < kSlots; ++k) {
                n[k] = (double)tab_n[k * G];
                e[k] = lut_e[tab_q[k * G]];
            }
        }
    }
    if (!active) {
#pragma unroll
        for (int k = 0; k < kSlots; ++k) { n[k] = 0.0; e[k] = 0.25; }
    }

    // EM (src/Algorithm.cpp:115-130): pass 0, then at most kEmIters passes each followed by the stop rule.  An item
    // that stops writes its fit at once and runs on (its lanes are not masked: the passes of a converged fit are
    // ordinary arithmetic, and nothing of it is read again).
    double fprev = fb, dprev = 0.0;
    int it = 0;
    bool done = !valid;
    while (true) {
        // stop-rule bracket of THIS pass: A = sum_b |f_b - f_b(previous pass)| D_b(previous pass), lane partial
        double ta = fabs(fb - fprev) * dprev;
        const double e_excl = fi_c->e_excl, fb_scale = fi_c->inv_n;
        const double g = fma(-4.0, fb, 1.0);
        double ysum0 = 0.0, ysum1 = 0.0, acc_e0 = 0.0, acc_e1 = 0.0;
#ifdef BVC_RCP_CHUNK
        constexpr int CH = kSlots > BVC_RCP_CHUNK ? BVC_RCP_CHUNK : kSlots;
#else
        constexpr int CH = kSlots;
#endif
#pragma unroll
        for (int c0 = 0; c0 < kSlots; c0 += CH) {
            constexpr int kLast = kSlots % CH == 0 ? CH : kSlots % CH;
            if (c0 + CH <= kSlots) {
                double m[CH], y[CH];
#pragma unroll
                for (int k = 0; k < CH; ++k) m[k] = fma(g, e[c0 + k], fb);      // class marginal f + (1 - 4 f) e
                rcp_all<CH>(m, y);
#pragma unroll
                for (int k = 0; k < CH; k += 2) {
                    const double r0 = n[c0 + k] * y[k], r1 = n[c0 + k + 1] * y[k + 1];
                    ysum0 += r0; ysum1 += r1;
                    acc_e0 = fma(r0, e[c0 + k], acc_e0); acc_e1 = fma(r1, e[c0 + k + 1], acc_e1);
                }
            } else {
                double m[kLast], y[kLast];
#pragma unroll
                for (int k = 0; k < kLast; ++k) m[k] = fma(g, e[c0 + k], fb);
                rcp_all<kLast>(m, y);
#pragma unroll
                for (int k = 0; k < kLast; k += 2) {
                    const double r0 = n[c0 + k] * y[k], r1 = n[c0 + k + 1] * y[k + 1];
                    ysum0 += r0; ysum1 += r1;
                    acc_e0 = fma(r0, e[c0 + k], acc_e0); acc_e1 = fma(r1, e[c0 + k + 1], acc_e1);
                }
            }
        }
        const double acc_e = acc_e0 + acc_e1;
        const double acc_d = fma(-4.0, acc_e, ysum0 + ysum1);   // sum n d / m with d = 1 - 4 e
        double etot = acc_e;
        item_sum2<ROWS, LOG2G>(etot, ta);
        const double dunit = unit_sum<LOG2G>(acc_d);            // D of the lane's allele
        const double ex = (fb * fb_scale) * (dunit + (etot + e_excl));   // expect_allele_prob (src/Algorithm.cpp:84-91)
        const uint32_t a_hi = (uint32_t)__double2hiint(ta);
        bool conv = it > 0 && a_hi < kSureBelowHi;
        const bool straddle = !done && it > 0 && a_hi >= kSureBelowHi && a_hi < kSureAboveHi;
        if (__ballot(straddle) != 0) {
            // rare: delta itself.  u = m' / m - 1 = (f' - f) d / m per class, log1p as a cubic (|u| < 2^-9 here)
            double dl = 0.0;
            const double df = fb - fprev, gp = fma(-4.0, fprev, 1.0);
#pragma unroll
            for (int k = 0; k < kSlots; ++k) {
                const double mo = fma(gp, e[k], fprev);
                const double u = df * fma(-4.0, e[k], 1.0) * rcp_cubic(mo);
                double p = fma(-0.25, u, 1.0 / 3.0);
                p = fma(p, u, -0.5);
                p = fma(p, u, 1.0);
                dl = fma(n[k], fabs(u * p), dl);
            }
            dl = item_sum<ROWS, LOG2G>(dl);
            conv = conv || (straddle && dl < kEmEpsilon);
        }
        const bool stop = !done && (conv || it == kEmIters);
        if (__ballot(stop) != 0) {
            if (stop && sub == 0) {
                FitOut *o = outs + item;
                o->ex[unit] = active ? ex : 0.0;
                o->fl[unit] = fb;                                // the frequency this last pass ran on
                if (ROWS == 2) o->ex[unit + 2] = 0.0;
                if (unit == 0) { o->passes = it + 1; o->pad = 0; }
            }
            done = done || stop;
            if (__ballot(!done) == 0) break;
        }
        fprev = fb;
        dprev = acc_d;
        fb = ex;
        ++it;
    }
    // UpdateF's log-likelihood (src/BaseType.cpp:58-62) at the frequencies of each item's last pass: the lane's
    // classes here, the alleles outside the subset from the item's constant
    {
        const double fl = outs[valid ? item : item0].fl[unit];   // (a place that holds no item: its sum is never stored)
        const double g = fma(-4.0, fl, 1.0);
        double ll = 0.0;
#pragma unroll
        for (int k = 0; k < kSlots; ++k) {
            ll = fma(n[k], log_pos(fma(g, e[k], fl)), ll);
            // four logarithms side by side are enough to fill the pipeline; all sixteen at once cost registers the kernel
            // does not have at three wavefronts per SIMD
            if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        ll = item_sum<ROWS, LOG2G>(ll);
        if (valid && unit == 0 && sub == 0) outs[item].ll = ll + items[item].ll_excl;
    }
}

// Lanes per allele of the two item shapes: two for four-row items (8 per wavefront), four for two-row items (8 per
// wavefront: the nested levels have few items per region, so they are spread over more wavefronts).
// (narrow and wide regions; the tiny ones take one lane per allele for four-row items and two for two-row items)
constexpr int kLog2G4 = 1, kLog2G2 = 2;
template <int CPB> constexpr int log2g4() { return CPB == kTiny ? 0 : kLog2G4; }
template <int CPB> constexpr int log2g2() { return CPB == kTiny ? 1 : kLog2G2; }
// ---- site_decide: one lane per site, one round of BaseType::LRT ---------------------------------------------------------
// Reads the fits of the pending round, takes the reference's decision (src/BaseType.cpp:93-110) and either sets up the next
// round's fits or writes the record.
//
// The subset a level may not have to run.  A level fits every (n-1)-subset of the current candidates and goes on with the
// FIRST MINIMUM of chi_c = 2 (lr_alt - loglik_c) (std::min_element, :99); nothing else of the other subsets is ever read
// (:100-105).  The subset without the deepest candidate explains that allele's observations as errors, fits worst by far and
// -- its EM runs to the cap of 101 passes whatever the site -- costs a quarter to a half of all the passes of a site.  Every
// class marginal is at most 1 (a mixture of likelihoods with weights that sum to 1), and an observation of an allele outside
// the subset has marginal e exactly (f = 0), so
//        loglik_c  <=  sum over the alleles b outside c of  sum_q n_bq log e_q  =: U_c        (ItemSite::lle, no EM needed)
// and chi_c >= 2 (lr_alt - U_c).  The level first runs the subsets that keep the deepest candidate; when that bound is above
// their minimum chi by more than kPruneSlack (rounding of the sums: far below it), the last subset cannot be the first
// minimum and is NOT RUN -- the level's decision, and with it every field of the record that the reference defines, is what
// it would have been.  Otherwise (sites of a few observations; never on the bench tiles) it is run in a second round of the
// level and joins the minimum exactly as std::min_element would have met it.  n_fits / n_passes of the record count what was
// run.  `prune` = 0 (em_prune, include/bvc.h): the second round always runs.
constexpr double kPruneSlackAbs = 1.0, kPruneSlackRel = 1e-6;

template <class RegionT>
__device__ __forceinline__ void site_decide(RegionT &R, int ls, int lane, int64_t site, int n_groups, int prune,
                                            const int8_t *__restrict__ ref_base, bvc_site_result *__restrict__ results)
{
    ItemSite S = R.site[ls];
    if (S.state != 1) {
        // (an opaque zero: the optimiser otherwise builds the bytes of zeros once per kernel, keeps them in registers
        // across the fits of every level and, at the narrow kernel's 168 VGPRs, spills them to scratch)
        int zero = 0;
        asm volatile("" : "+v"(zero));
        if (lane == 0) { R.want[ls][0] = zero; R.want[ls][1] = zero; R.site[ls].n_emit = 0; }
        return;
    }
    uint32_t sets = 0;
    int n_emit = 0, p_deepest = 0;
    bool finished = false;

    auto fit_loglik = [&](int idx, double (&ex)[4], int &passes) -> double {
        if (!BVC_LDS_OK(15, idx, kPlaces)) idx = 0;
        const FitOut &o = R.outs[idx];
#pragma unroll
        for (int u = 0; u < 4; ++u) ex[u] = o.ex[u];
        passes = o.passes;
        return o.ll;
    };
    // expect_allele_prob by allele from the per-unit values of a fit
    auto by_base = [&](uint32_t pm, const double (&ex)[4], double (&bp)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bp[j] = 0.0;
        int u = 0;
#pragma unroll
        for (int p = 0; p < 4; ++p)
            if ((pm >> p) & 1u) {
                const int b = (S.blist >> (4 * p)) & 3u;
                const double v = u == 0 ? ex[0] : (u == 1 ? ex[1] : (u == 2 ? ex[2] : ex[3]));
                bp[0] = b == 0 ? v : bp[0]; bp[1] = b == 1 ? v : bp[1];
                bp[2] = b == 2 ? v : bp[2]; bp[3] = b == 3 ? v : bp[3];
                ++u;
            }
    };

    int first_sub = 0;
    if (S.first) {                                               // the full model (:88-90)
        double ex[4];
        int passes = 0;
        S.lr_alt = fit_loglik(S.item[0], ex, passes);
        by_base((1u << S.n) - 1u, ex, S.base_frq);
        S.passes += passes; S.fits += 1;
        first_sub = 1;
        S.k = (int8_t)(S.n - 1);
        S.first = 0;
    }
    const int n = S.n, k = S.k;
    bool again = false;                                          // the level needs its second round
    if (k < 1) {
        finished = true;                                         // n == 1: no nested level
    } else if (k >= 2) {
        // lexicographic number of the subset kept for later (level_subsets); the others keep their order around it
        const int c_last = n - 1 - S.p_deepest;
        int i_min = S.best_i;
        double best_chi = S.best_chi, best_lr = S.best_lr;
        double best_bp[4] = {S.best_bp[0], S.best_bp[1], S.best_bp[2], S.best_bp[3]};
        if (!S.round2) {
            const uint32_t masks = S.sets >> (4 * first_sub);
            const int cnt = S.n_emit - first_sub;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (c >= cnt) break;
                const uint32_t pm = (masks >> (4 * c)) & 0xFu;
                double ex[4];
                int passes = 0;
                const double ll = fit_loglik(first_sub ? S.item[c + 1] : S.item[c], ex, passes);
                S.passes += passes; S.fits += 1;
                const double chi_c = 2.0 * (S.lr_alt - ll);
                if (c == 0 || chi_c < best_chi) {                // std::min_element: first minimum, '<'
                    best_chi = chi_c; best_lr = ll; i_min = (S.later && c >= c_last) ? c + 1 : c;
                    by_base(pm, ex, best_bp);
                }
            }
            if (S.later) {
                double u_c = 0.0;                                // U_c: the alleles outside the subset, candidates or not
                uint32_t in_set = 0;
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    if ((S.later >> p) & 1u) in_set |= 1u << ((S.blist >> (4 * p)) & 3u);
#pragma unroll
                for (int b = 0; b < 4; ++b) u_c += ((in_set >> b) & 1u) ? 0.0 : S.lle[b];
                const double bound = 2.0 * (S.lr_alt - u_c);
                // (NaN or infinities on either side: not ruled out)
                const bool ruled_out = prune && bound > best_chi + (kPruneSlackAbs + kPruneSlackRel * fabs(u_c)) && bound < __builtin_huge_val();
                again = !ruled_out;
            }
        } else {
            // second round: the subset kept for later has been run (S.item[0]); it meets the first minimum of the others
            double ex[4];
            int passes = 0;
            const double ll = fit_loglik(S.item[0], ex, passes);
            S.passes += passes; S.fits += 1;
            const double chi_c = 2.0 * (S.lr_alt - ll);
            if (chi_c < best_chi || (chi_c == best_chi && c_last < i_min)) {
                best_chi = chi_c; best_lr = ll; i_min = c_last;
                by_base(S.later, ex, best_bp);
            }
        }
        if (again) {
            S.best_chi = best_chi; S.best_lr = best_lr; S.best_i = (int8_t)i_min;
#pragma unroll
            for (int j = 0; j < 4; ++j) S.best_bp[j] = best_bp[j];
            S.round2 = 1;
            sets = S.later; n_emit = 1; p_deepest = S.p_deepest;
        } else {
            S.round2 = 0;
            S.lr_alt = best_lr;                                  // overwritten before the threshold test (:100-101)
            S.chi = best_chi;
            if (best_chi < kLrtThreshold) {
                // the subset that won, by its lexicographic number
                int cnt_all = 0;
                const uint32_t pm = (subset_masks(n, k, cnt_all) >> (4 * i_min)) & 0xFu;
                uint32_t nl = 0;
                int nn = 0;
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    if ((pm >> p) & 1u) { nl |= ((S.blist >> (4 * p)) & 3u) << (4 * nn); ++nn; }
                S.blist = nl;
                S.n = (int8_t)k;
#pragma unroll
                for (int j = 0; j < 4; ++j) S.base_frq[j] = best_bp[j];
                S.k = (int8_t)(k - 1);
                S.later = 0;
                if (k - 1 >= 2) {
                    uint32_t later = 0;
                    p_deepest = deepest_position(S.depth, nl, k);
                    sets = level_subsets(k, p_deepest, n_emit, later);
                    S.later = (uint8_t)later;
                }                                                // (k - 1 == 1: the one-allele level, below)
            } else {
                finished = true;
            }
        }
    }
    if (!finished && !again && S.k == 1) {
        // The last level (:93-110 with k = 1): the two one-allele models of the two candidates left.  Such a model needs no EM:
        // f = 1, so every observation's marginal is its own likelihood of the allele (a for the allele's observations, e for the
        // others: the sums over classes are the site's lla / lle), every posterior is 1 and expect = nind / nind = 1 exactly; the
        // reference's EM stops after its second pass with delta = 0.  Both are "run" (two fits, four passes).
        const int b0 = (int)(S.blist & 3u), b1 = (int)((S.blist >> 4) & 3u);
        const double ll0 = single_allele_loglik(S.lla, S.lle, b0), ll1 = single_allele_loglik(S.lla, S.lle, b1);
        const double chi0 = 2.0 * (S.lr_alt - ll0), chi1 = 2.0 * (S.lr_alt - ll1);
        const bool second = chi1 < chi0;                         // std::min_element: first minimum, '<'
        const double best_chi = second ? chi1 : chi0;
        const int bb = second ? b1 : b0;
        S.passes += 4; S.fits += 2;
        S.lr_alt = second ? ll1 : ll0;
        S.chi = best_chi;
        if (best_chi < kLrtThreshold) {
            S.blist = (uint32_t)bb;
            S.n = 1;
            S.base_frq[0] = bb == 0 ? 1.0 : 0.0; S.base_frq[1] = bb == 1 ? 1.0 : 0.0;
            S.base_frq[2] = bb == 2 ? 1.0 : 0.0; S.base_frq[3] = bb == 3 ? 1.0 : 0.0;
        }
        S.k = 0;
        finished = true;
    }
    S.sets = sets; S.n_emit = (int8_t)n_emit; S.p_deepest = (int8_t)p_deepest;
    int want[kLists];
    count_wanted(sets, n_emit, want);
    if (lane == 0) {
        if (finished) {
            store_record(results + site, S, (int)ref_base[ref_index(site, n_groups)], S.n, S.blist);
            S.state = 2;
        }
        R.site[ls] = S;
        R.want[ls][0] = want[0]; R.want[ls][1] = want[1];
    }
}

// ---- region kernels: one WAVEFRONT per region of eight sites -------------------------------------------------------------
struct RegionArgs {
    int64_t n_sites;
    int n_groups;
    const uint32_t *counts;
    int64_t hist_stride;
    const QualLut *lut;
    const int8_t *ref_base;
    double min_af;
    const int8_t *comb;
    const uint8_t *n_comb;
    uint8_t *taken;
    bvc_site_result *results;
    uint32_t *kind_epoch;        // [0] / [1]: set to `epoch` by the narrow launch when some region is the tiny / the wide launch's
    uint32_t epoch;
    int dbg_levels;
    int tiny_regions;            // LaunchState::em_tiny_regions
    int prune;                   // LaunchState::em_prune (site_decide)
    int64_t region0;             // first region of this launch (a call may be a sequence of launches)
};

// Orders the wavefront's own LDS traffic around a point: everything a region keeps in LDS is private to its wavefront, the
// other wavefronts of the workgroup work on other regions and are never waited for.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// The 512 class counts of a site as the lanes of site_classes hold them: lane (row, t) has qualities t, t + 16, ... of allele `row`.
__device__ __forceinline__ void load_counts(uint32_t (&c)[8], const uint32_t *__restrict__ counts, int64_t hist_stride, int64_t site,
                                            int n_groups, int lane)
{
    const uint32_t *hist = counts + hist_index(site, n_groups) * hist_stride + (lane >> 4) * 128 + (lane & 15);
#pragma unroll
    for (int lvl = 0; lvl < 8; ++lvl) c[lvl] = hist[16 * lvl];
}

// A region belongs to ONE launch by the most class places any of its sites needs on an allele: <= 32 narrow, <= 48 wide
// (and, with LaunchState::em_tiny_regions, <= 8 tiny).  The narrow and the wide kernel place class c of an allele in the
// same lane and slot and add in the same order (the wide one's extra slots are empty for a narrow site and add exact
// zeros; rcp_all<24> is rcp_all<16> on the first sixteen), so a site's record does not depend on which of the two its
// region went to, i.e. not on its neighbours.  The tiny kernel (one lane per allele) adds in another order: opt-in.  Each
// runs the classes phase of the regions it looks at and goes on only with its own; the narrow launch comes first, and the
// other two return at once when it has met no region of theirs in this call.
//
// The wavefront-slots of a round -- eight fits each -- run one after the other; a slot lasts as long as its slowest fit.
// TEAM wavefronts share a region (1, 2 or 4; a workgroup holds 4 / TEAM regions): they split the site phases by site and the
// fit phases by wavefront-slot (an LDS counter hands the slots out), and meet at a team barrier between
// the phases.  The barrier is an arrival counter in the region's LDS, never reset (phase p is over when it reads p * TEAM):
// only the region's own wavefronts wait for one another, the rest of the workgroup is never involved.
template <int TEAM, class RegionT>
__device__ __forceinline__ void team_sync(RegionT &R, uint32_t &phase, int lane)
{
    if (TEAM == 1) { wave_sync(); return; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    phase += TEAM;
    if (lane == 0) {
        __hip_atomic_fetch_add(&R.arrive, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(&R.arrive, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) - phase > 0x7FFFFFFFu) __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <bool WALK, int CPB, int TEAM>
__device__ __forceinline__ void region_body(Region<CPB> *regions, const RegionArgs &A)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int teams = (int)(blockDim.x >> 6) / TEAM;             // regions a workgroup works on at a time
    const int member = wave % TEAM;
    const int lane = threadIdx.x & (kWave - 1);
    Region<CPB> &R = regions[wave / TEAM];
    const int64_t n_sites = A.n_sites;
    const int64_t n_regions = (n_sites + kRegionSites - 1) / kRegionSites;
    // the narrow launch comes first and stamps these words when it meets a region for one of the others
    if (CPB == kWide && A.kind_epoch[1] != A.epoch) return;
    if (CPB == kTiny && A.kind_epoch[0] != A.epoch) return;
    uint32_t phase = 0;
    if (TEAM > 1) {
        if (member == 0 && lane == 0) R.arrive = 0u;
        __syncthreads();                                         // the only workgroup barrier: once, before any team barrier
    }
    // team t of workgroup b takes region region0 + b * teams + t.  WALK = false: exactly that one -- the launcher sizes the
    // grid, or cuts the call into a sequence of launches when stage 2 may hold only a few wavefronts per CU -- and without a
    // loop nothing is kept live across regions, which is what lets the narrow kernel hold three wavefronts per SIMD (168
    // VGPRs; the walking form needs 257).  WALK = true (experiments only): + gridDim.x * teams, ... to the end.
    int64_t region = A.region0 + (int64_t)blockIdx.x * teams + wave / TEAM;
    for (bool first = true; region < n_regions && (WALK || first); first = false, region += (int64_t)gridDim.x * teams) {
        const int64_t site0 = region * kRegionSites;
        if (member == 0 && lane == 0) { R.need = 0; R.next_slot = 0; }
        if (TEAM > 1) team_sync<TEAM>(R, phase, lane);
        // ---- classes: a member's sites one after the other, the next site's counts in flight while the current one is compacted
        int need = 0;
        uint32_t cnext[8];
        if (site0 + member < n_sites) load_counts(cnext, A.counts, A.hist_stride, site0 + member, A.n_groups, lane);
#pragma unroll 1
        for (int ls = member; ls < kRegionSites; ls += TEAM) {
            uint32_t c8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) c8[k] = cnext[k];
            if (ls + TEAM < kRegionSites && site0 + ls + TEAM < n_sites) load_counts(cnext, A.counts, A.hist_stride, site0 + ls + TEAM, A.n_groups, lane);
            if (site0 + ls < n_sites) {
                need = max(need, site_classes<CPB>(R, ls, lane, site0 + ls, c8, A.lut, A.min_af, A.comb, A.n_comb, A.taken));
            } else if (lane == 0) {
                R.site[ls].state = 2; R.site[ls].n_emit = 0;
                R.want[ls][0] = 0; R.want[ls][1] = 0;
            }
        }
        if (TEAM > 1) {
            if (lane == 0) atomicMax(&R.need, need);
            team_sync<TEAM>(R, phase, lane);
            need = R.need;
        } else {
            wave_sync();
        }
        // whose region this is: tiny when every taken site has at most kTiny class places per allele, wide when some site
        // has more than kNarrow, narrow otherwise
        const int kind = (A.tiny_regions && need <= kTiny) ? kTiny : (need <= kNarrow ? kNarrow : kWide);
        if (CPB == kNarrow && kind != kNarrow && member == 0 && lane == 0) A.kind_epoch[kind == kWide ? 1 : 0] = A.epoch;
        if (kind != CPB) {                                       // (the whole team: `need` is the region's)
            if (TEAM > 1) team_sync<TEAM>(R, phase, lane);       // nobody rewrites R.need before everybody has read it
            continue;
        }
        // records of the sites that ended in the classes phase (no observation, no candidate)
        if (member == 0 && lane < kRegionSites && R.site[lane].state == 3) {
            const int64_t site = site0 + lane;
            store_record(A.results + site, R.site[lane], (int)A.ref_base[ref_index(site, A.n_groups)], R.site[lane].n,
                         R.site[lane].blist);
            R.site[lane].state = 2;
        }
        team_sync<TEAM>(R, phase, lane);
        for (int round = 0; round < kRounds && round < (A.dbg_levels >> 1); ++round) {
            // the fits of this round: list after list, each in site order
            int cnt[kLists] = {0, 0};
#pragma unroll
            for (int w = 0; w < kRegionSites; ++w) {
#pragma unroll
                for (int l = 0; l < kLists; ++l) cnt[l] += R.want[w][l];
            }
            if (cnt[0] + cnt[1] == 0) break;                     // (the same LDS words for the whole team)
            const int first[kLists] = {0, cnt[0]};
            if (!BVC_LDS_OK(16, cnt[0] + cnt[1], kPlaces + 1)) break;
            if (member == 0) region_emit(R, lane, first);
            team_sync<TEAM>(R, phase, lane);
            constexpr int kPerWave4 = 16 >> log2g4<CPB>(), kPerWave2 = 2 * (16 >> log2g2<CPB>());
            const FitItem *items = R.items;
            FitOut *outs = R.outs;
            const uint8_t *tabs = &R.tab[0][0];
            const double *lut_e = A.lut->e;
            // wavefront-slots of the round: w0 of list 0 (eight four-row fits each), then list 1; an LDS counter hands them out
            const int w0 = (cnt[0] + kPerWave4 - 1) / kPerWave4, w1 = w0 + (cnt[1] + kPerWave2 - 1) / kPerWave2;
            int slot = 0;
#pragma unroll 1
            for (;;) {
                if (TEAM > 1) {
                    if (lane == 0) slot = atomicAdd(&R.next_slot, 1);
                    slot = __builtin_amdgcn_readfirstlane(slot);
                }
                if (slot >= w1) break;
                if (slot < w0) fit_body<4, log2g4<CPB>(), CPB>(slot * kPerWave4, cnt[0], items, outs, tabs, lut_e);
                else fit_body<2, log2g2<CPB>(), CPB>(cnt[0] + (slot - w0) * kPerWave2, cnt[0] + cnt[1], items, outs, tabs, lut_e);
                if (TEAM == 1) ++slot;
            }
            team_sync<TEAM>(R, phase, lane);
            if (TEAM > 1 && member == 0 && lane == 0) R.next_slot = 0;      // nobody takes a slot again before the barrier behind the next emit
            if (round + 1 == (A.dbg_levels >> 1) && (A.dbg_levels & 1)) break;
            // the decisions of the round, one LANE per site (the reference's few dozen scalar steps per site -- read the fits, first
            // minimum, threshold, next subsets or the record -- once for the region instead of once per site)
            int my = lane;
            asm volatile("" : "+v"(my));                          // (opaque: nothing of the site's addressing is hoisted out of the round loop)
            if (member == 0 && my < kRegionSites && site0 + my < n_sites)
                site_decide(R, my, 0, site0 + my, A.n_groups, A.prune, A.ref_base, A.results);
            team_sync<TEAM>(R, phase, lane);
        }
        team_sync<TEAM>(R, phase, lane);                         // the region's LDS is reused by the next one
    }
}

// (diagnostic builds carry extra code: no occupancy target there, so that they do not spill where the product does not)
#define BVC_OCCUPANCY(n) BVC_WAVES_PER_EU(n, n)
#ifdef BVC_RCP_CHUNK
#define BVC_REGION_WAVES_PER_EU 4
#else
#define BVC_REGION_WAVES_PER_EU 3
#endif
constexpr int kMaxRegionWaves = 4;               // wavefronts of a workgroup (4 / team regions in dynamic LDS)
#ifndef BVC_REGION_TEAM
#define BVC_REGION_TEAM 2
#endif
constexpr int kTeam = BVC_REGION_TEAM;          // wavefronts that share a region

// One region per team of the launch: three narrow (or tiny) wavefronts per SIMD (168 VGPRs), two wide ones (24 classes per lane).
// CPB = 32 narrow, 48 wide (a site of 33..48 quality values on an allele), 8 tiny (binned qualities: one lane per allele, 16
// four-allele fits per wavefront; opt-in).
template <int CPB>
__global__ __launch_bounds__(64 * kMaxRegionWaves) BVC_OCCUPANCY(CPB == kWide ? 2 : BVC_REGION_WAVES_PER_EU) void region_kernel(RegionArgs A)
{
    BVC_POISON_LDS();
    extern __shared__ __attribute__((aligned(16))) unsigned char region_lds[];
    region_body<false, CPB, kTeam>(reinterpret_cast<Region<CPB> *>(region_lds), A);
}

}  // namespace

#ifdef BVC_CHECK_LDS
BVC_DEFINE_DEBUG_READER(debug_read_items)
#endif

size_t em_items_scratch_bytes(int64_t n_sites)
{
    return (size_t)n_sites + 256 + 64;                            // the `taken` flags and the "wide regions seen" word; everything else lives in LDS
}

// Stage 2 with the item engine.  `scratch` holds em_items_scratch_bytes(n_sites).  Sites it does not take are left
// to the one-wavefront-per-site kernels, which skip the others (`taken`).
hipError_t launch_lrt_items(const LaunchState &st, hipStream_t stream, int64_t n_sites, int n_groups,
                            const uint32_t *counts, int64_t hist_stride, const int8_t *ref_base, double min_af,
                            const QualLut *lut, const int8_t *comb, const uint8_t *n_comb, bvc_site_result *results,
                            void *scratch, const uint8_t **taken_out, bool shared)
{
    uint8_t *taken = static_cast<uint8_t *>(scratch);
    const int64_t regions = (n_sites + kRegionSites - 1) / kRegionSites;
    // A team of wavefronts is a region's whole engine, so the shape of the launch is free:
    //  * the chip to itself: as many teams per workgroup (<= 4 wavefronts: one per SIMD) as leave every CU a workgroup, one
    //    region per team, the dispatcher balances;
    //  * underneath a streaming histogram pass (overlap mode, long rows): at most ONE workgroup (four wavefronts) per CU at a
    //    time -- a call is cut into a sequence of launches of n_cu workgroups (a 4000-site call is one launch of 250).  The
    //    histogram kernels keep two 64 KiB workgroups on a CU, which leaves 32 KiB of its LDS (4 x 5.5 KB here) and, beside four
    //    packed-histogram wavefronts of 64 VGPRs per SIMD, room for one narrow wavefront of 168.
    // em_waves_per_cu (bvc_set_tuning) overrides: stage-2 wavefronts per CU and launch.
    int per_cu = 0;
    if (st.em_waves_per_cu > 0) per_cu = st.em_waves_per_cu;
    else if (shared) per_cu = 4;
    constexpr int kMaxTeams = kMaxRegionWaves / kTeam;           // regions a workgroup can work on at a time
    int teams;
    int64_t wgs, per_launch;
    if (per_cu > 0) {
        const int want_teams = std::max(1, per_cu / kTeam);
        teams = std::min(want_teams, kMaxTeams);
        wgs = (regions + teams - 1) / teams;
        per_launch = (int64_t)st.n_cu * ((want_teams + teams - 1) / teams);
    } else {
        teams = (int)std::min<int64_t>(kMaxTeams, std::max<int64_t>(1, (regions + st.n_cu - 1) / st.n_cu));
        wgs = (regions + teams - 1) / teams;
        per_launch = wgs;
    }
    RegionArgs A;
    A.n_sites = n_sites; A.n_groups = n_groups; A.counts = counts; A.hist_stride = hist_stride; A.lut = lut;
    A.ref_base = ref_base; A.min_af = min_af; A.comb = comb; A.n_comb = n_comb; A.taken = taken; A.results = results;
    A.kind_epoch = reinterpret_cast<uint32_t *>(taken + (((size_t)n_sites + 255) & ~(size_t)255));
    A.epoch = ++st.em_epoch;                                     // never 0; a stale word can only cost the wide launch a scan
    if (A.epoch == 0) A.epoch = ++st.em_epoch;
    A.dbg_levels = st.dbg_levels > 0 ? st.dbg_levels : 2 * kRounds;
    A.tiny_regions = st.em_tiny_regions;
    A.prune = st.em_prune;
    A.region0 = 0;
    // the wide kernels' dynamic LDS (4 x 14 KB) is beyond the 48 KiB a launch may ask for without the attribute
    constexpr uint32_t kSlotRegionWide = 60;
    if (!(st.attr_done & ((uint64_t)1 << kSlotRegionWide))) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(region_kernel<kWide>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)(kMaxTeams * sizeof(Region<kWide>)));
        if (e != hipSuccess) return e;
        st.attr_done |= (uint64_t)1 << kSlotRegionWide;
    }
    const dim3 block(64 * teams * kTeam);
    auto sequence = [&](auto kernel, size_t lds_per_team) {
        for (int64_t w0 = 0; w0 < wgs; w0 += per_launch) {
            A.region0 = w0 * teams;
            hipLaunchKernelGGL(kernel, dim3((unsigned)std::min(per_launch, wgs - w0)), block, teams * lds_per_team, stream, A);
        }
        A.region0 = 0;
    };
    sequence(region_kernel<kNarrow>, sizeof(Region<kNarrow>));
    if (st.em_tiny_regions) sequence(region_kernel<kTiny>, sizeof(Region<kTiny>));
    // (A probe instead of the wide grid -- 32 single-wavefront workgroups that walk the regions, full grids only while a host-visible
    // hint says recent calls had wide regions -- was built and measured in round 4: no gain, its wavefronts need MORE than half a
    // SIMD's registers (268 VGPRs against 254) and wait for room just like the grid it replaces.  profiles/r04_stage2_teams.txt)
    sequence(region_kernel<kWide>, sizeof(Region<kWide>));
    *taken_out = taken;
    return hipGetLastError();
}

}  // namespace bvc
