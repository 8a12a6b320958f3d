// em_kernel.hip -- stage 2 of the basetype path on gfx950: EM over four allele frequencies and the
// nested likelihood-ratio test, run on a site's (base, qual) count histogram by ONE wavefront.
//
// Follows, with the per-sample sums regrouped by class (a class = all samples with equal base and qual,
// which have identical likelihood rows):
//   BaseType::SetAlleleFreq  /root/reference/src/BaseType.cpp:25-39
//   BaseType::UpdateF        /root/reference/src/BaseType.cpp:41-71
//   BaseType::LRT            /root/reference/src/BaseType.cpp:73-139
//   combs_                   /root/reference/src/BaseType.cpp:237-255
//   singleEM / EM / delta    /root/reference/src/Algorithm.cpp:69-130
//
// Lane layout: the wave's four DPP rows of 16 lanes own the four bases; lane (b, t) holds the
// t-th, (t+16)-th ... non-empty quality class of base b in registers.  For a class of base b with
// likelihoods a = 1-eps (match) and e = eps/3 (mismatch), d = a - e:
//   marginal      m   = f_b * d + F * e                       F = f_A + f_C + f_G + f_T
//   M step        f_j' = f_j / N * (D_j + E)                  D_j = sum_{c in j} n_c d_c / m_c
//                                                             E   = sum_c       n_c e_c / m_c
//   stop rule     delta = sum_c n_c |log m_c' - log m_c| < 1e-3
// FP64 throughout; no MFMA (nothing here is a dense contraction).
#include "bvc_device.h"
#include "bvc_internal.h"

namespace bvc {
namespace {

constexpr double kLrtThreshold = 24.0;    // LRT_THRESHOLD, src/BaseType.h:9
constexpr int kEmIters = 100;             // src/BaseType.cpp:46
// var_qual is >= 0 or NaN; this marks records whose chi-square tail is still to be evaluated.  The libm-style
// log/exp/log10 of that step live in their own small kernel so that their constants and registers stay out
// of the EM kernel (hoisted into VGPRs across the site loop they cost it half its occupancy).
constexpr double kVarQualPending = -1.0;
constexpr double kEmEpsilon = 0.001;      // src/BaseType.cpp:45

// Register-resident classes of one lane.  NS (slots per lane) is a template parameter: the kernel is
// instantiated for NS = 2, 4 and 8 and each site is handled by the smallest variant that holds it, so the
// common case (<= 32 quality values per base) runs with a small register footprint and many waves per SIMD.
template <int NS>
struct Slots {
    double n[NS];    // class count (0 for an empty slot)
    double e[NS];    // eps / 3 : likelihood of the observed base given a non-matching allele
    double d[NS];    // (1 - eps) - eps / 3 : matching minus non-matching likelihood
    double yp[NS];   // 1 / (class marginal) from the previous pass
};

// k-subsets of positions 0..n-1 in lexicographic order (what combs_ yields), as 4-bit position masks
// packed least-significant first; count returned through `cnt`.
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

__device__ __forceinline__ double pick4(const double (&v)[4], int j)
{
    return j == 0 ? v[0] : (j == 1 ? v[1] : (j == 2 ? v[2] : v[3]));
}

__device__ __forceinline__ int pick4(const int (&v)[4], int j)
{
    return j == 0 ? v[0] : (j == 1 ? v[1] : (j == 2 ? v[2] : v[3]));
}

// Frequencies live per lane: fb = frequency of the lane's own base.  The class marginal is the reference's
// sum_j f_j * L_ij (src/Algorithm.cpp:74-78) with the three equal terms grouped and the other three
// frequencies taken as 1 - fb (the four sum to 1 after every M step, sum_j expect_j = (1/N) sum_i 1, and at
// the start, up to the few ulp the reference's own f_j carry):
//     m = fb * a + (1 - fb) * e = fb * (a - e) + e = fma(fb, d, e)
// One rounding of the exact value; for a = 0 (Q = 0) d = -e exactly and m = e * (1 - fb) comes out with full
// relative accuracy, including m = 0 when fb = 1 -- the case the reference leaves unguarded (:81-82).
struct Freq {
    double fb;
};

constexpr int kDppRor8 = 0x128;                 // row_ror:8
constexpr int kDppRowBcast15 = 0x142;           // row_bcast:15 (GFX9 family): lane 15 of each row to every lane of the next row
template <int L> constexpr int dpp_newbcast() { return 0x150 + L; }   // row_newbcast:L (lane L of each row)

struct PassOut {
    double ex_own;          // expect_allele_prob of the lane's own base (uniform within the row)
    bool converged;         // delta = sum_c n_c |log m_c' - log m_c| < 1e-3 (wave-uniform)
};

// One E+M pass (singleEM, src/Algorithm.cpp:69-93) plus the convergence test on delta_bylog's delta (:103-113).
//   M step: expect_j = f_j / N * (D_j + E),  D_j = sum_{c in j} n_c (a_c - e_c) / m_c,  E = sum_c n_c e_c / m_c
//   u     : m' / m - 1 = m' * yp - 1  with yp = 1/m kept from the previous pass
//   1/m'  : yp / (1 + u) -> yp * (1 - u) refined by two Newton steps; v_rcp_f64 + two Newton steps only when some
//           |u| > 2^-6 (the first passes of a fit)
//   delta : sum_c n_c |log1p(u_c)| is used ONLY in the test delta < eps = 1e-3, so it is bracketed instead of
//           evaluated: with A = sum_c n_c |u_c| (one FMA per class),
//             A >= eps / (1 - 2^-8): not converged.  Either some |u| >= 2^-9, and that class alone (n_c >= 1) gives
//                  delta > 1.9e-3; or every |u| < 2^-9, where |log1p(u)| >= |u| (1 - 2^-9), so delta >= eps.
//             A <  eps / (1 + 2^-8): every n_c |u_c| < eps, so every |u| < 2^-9, |log1p(u)| <= |u| (1 + 2^-9) and
//                  delta < eps: converged.
//             in between (a few passes per fit at most): delta itself, log1p as a cubic (truncation 3e-12 relative),
//                  with a reduction of its own.
//           A is a sum of non-negative doubles (or NaN), so both comparisons are unsigned compares of its high
//           word, done on the scalar unit; NaN and +inf compare high: "NaN never converges", as in the reference.
// Empty slots have d = 0, e = 1, so m = 1 and u = 0 to an ulp: they add nothing.
// The NS slots are independent dependency chains with no branch between them, so they interleave.
constexpr uint32_t hi_word(double x) { return (uint32_t)(__builtin_bit_cast(uint64_t, x) >> 32); }
constexpr uint32_t kSureBelowHi = hi_word(kEmEpsilon / (1.0 + 0.00390625));        // hi(A) <  this: converged
constexpr uint32_t kSureAboveHi = hi_word(kEmEpsilon / (1.0 - 0.00390625)) + 1u;   // hi(A) >= this: not converged

// NA <= NS: slots that can hold a class at this site.  A site whose bases have at most 16 * NA quality values leaves
// slots NA..NS-1 empty in every lane, and an empty slot adds exact zeros to every sum (n = 0) and 0 to max|u|
// (m = y = 1), so leaving them out changes no bit of the result -- only the instruction count (NS = 2 with one active
// slot: binned qualities, <= 16 values; NS = 4 with three: Illumina's ~40 values; NS = 8 with six: BAM's 0..93).
template <int NS, int NA = NS>
__device__ __forceinline__ PassOut em_pass(Slots<NS> &S, const Freq f, double inv_n, int lane)
{
    static_assert(NA >= 1 && NA <= NS, "active slots");
    double acc_d = 0.0, acc_e = 0.0, acc_a;
    double m[NS], u[NS];
    double umax = 0.0;
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        m[k] = fma(f.fb, S.d[k], S.e[k]);
        u[k] = fma(m[k], S.yp[k], -1.0);
        umax = fmax(umax, fabs(u[k]));
        // acc_a += n |u| in one instruction (the compiler materialises |u| with two extra moves otherwise)
        if (k == 0) asm("v_mul_f64 %0, %1, |%2|" : "=v"(acc_a) : "v"(S.n[k]), "v"(u[k]));
        else asm("v_fma_f64 %0, %1, |%2|, %0" : "+v"(acc_a) : "v"(S.n[k]), "v"(u[k]));
    }
    // The Newton path runs unconditionally and the rare jump passes redo 1/m afterwards: the vote on |u| is then
    // long decided when its branch comes, instead of stalling the wave between u and the Newton steps.
    const bool jump = __ballot(umax > kLog1pMaxU) != 0;
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        double y = fma(-S.yp[k], u[k], S.yp[k]);
        y = fma(y, fma(-m[k], y, 1.0), y);
        S.yp[k] = fma(y, fma(-m[k], y, 1.0), y);
        asm volatile("" : "+v"(S.yp[k]));                       // keeps these steps ahead of the branch below
    }
    if (jump) {
#pragma unroll
        for (int k = 0; k < NA; ++k) S.yp[k] = fast_rcp(m[k]);
    }
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const double r = S.n[k] * S.yp[k];
        acc_d = fma(r, S.d[k], acc_d);
        acc_e = fma(r, S.e[k], acc_e);
    }
    // Three sums in one 16-lane reduction.  swap32 folds the wave's halves: z = E over lanes (l, l+32) in the
    // lower half, A in the upper half.  Within each row, lanes 0-7 then reduce D and lanes 8-15 reduce z
    // (one exchange across the row's halves, three butterfly steps); swap16 adds the row pairs.
    const DPair h = swap32(acc_e, acc_a);
    const double z = h.a + h.b;
    const bool hi = (lane & 8) != 0;
    double v = (hi ? z : acc_d) + dpp_f64<kDppRor8>(hi ? acc_d : z);
    v += dpp_f64<kDppXor1>(v);
    v += dpp_f64<kDppXor2>(v);
    v += dpp_f64<kDppHalfMirror>(v);
    const double drow = dpp_f64<dpp_newbcast<0>()>(v);          // lanes 0-7 of the row: D of the row's base
    // rows 1 and 3 add the z of the row before them (row_bcast:15 hands lane 15 of a row to the next row): lanes 24-31
    // then hold E, lanes 56-63 A.  (Two instructions fewer than copying v and swapping rows with v_permlane16_swap.)
    const double t = v + dpp_f64<kDppRowBcast15>(v);
    const double etot = lane_value<24>(t);
    const uint32_t a_hi = (uint32_t)__builtin_amdgcn_readlane(__double2hiint(t), 56);
    PassOut o;
    o.converged = a_hi < kSureBelowHi;
    if (a_hi >= kSureBelowHi && a_hi < kSureAboveHi) {           // rare: the bracket straddles eps
        double ex = 0.0;
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            double p = fma(-0.25, u[k], 1.0 / 3.0);
            p = fma(p, u[k], -0.5);
            p = fma(p, u[k], 1.0);
            ex = fma(S.n[k], fabs(u[k] * p), ex);
        }
        o.converged = rows_total(row_sum(ex)) < kEmEpsilon;
    }
    o.ex_own = f.fb * inv_n * (drow + etot);
    return o;
}

// EM (src/Algorithm.cpp:115-130) followed by UpdateF's log-likelihood sum (src/BaseType.cpp:58-62).
// f0 = the lane's initial frequency.  Returns the log-likelihood of the last pass; ex = expect_allele_prob
// of that pass (one M step ahead of the frequencies the log-likelihood belongs to, as in the reference).
template <int NS, int NA = NS>
__device__ __forceinline__ double em_fit(Slots<NS> &S, int lane, double f0, double inv_n,
                                         double (&ex)[4], int &passes)
{
    Freq f{f0};
    PassOut o;
    // every fit starts from yp = 1 (not from the previous fit's, which may hold NaNs): its first pass then
    // has |u| = |m - 1| and takes the big tier, whose reciprocal does not depend on yp
#pragma unroll
    for (int k = 0; k < NA; ++k) S.yp[k] = 1.0;
    // pass 0 + at most kEmIters update passes; two passes per trip so that 1/m can alternate between two register
    // sets instead of being copied back every pass
    for (int it = 0;; it += 2) {
        o = em_pass<NS, NA>(S, f, inv_n, lane);
        passes += 1;
        if (it > 0 && o.converged) break;
        if (it == kEmIters) break;
        f.fb = o.ex_own;
        o = em_pass<NS, NA>(S, f, inv_n, lane);
        passes += 1;
        if (o.converged) break;
        if (it + 1 == kEmIters) break;
        f.fb = o.ex_own;
    }
    ex[0] = lane_value<0>(o.ex_own);
    ex[1] = lane_value<16>(o.ex_own);
    ex[2] = lane_value<32>(o.ex_own);
    ex[3] = lane_value<48>(o.ex_own);
    double ll = 0.0;                           // sum_c n_c log m_c = -sum_c n_c log(1/m_c)
#pragma unroll
    for (int k = 0; k < NA; ++k) ll = fma(-S.n[k], log_pos(S.yp[k]), ll);
    return rows_total(row_sum(ll));
}

struct SiteOut {
    double var_qual, chi, depth_total, lr_alt;
    double af[3], base_frq[4];
    int depth[4];
    int n_passes, n_fits;
    int alt_base[3], n_alt, called, n_kept, kept[4], status;
};

// Orders a wavefront's LDS accesses around a point (its scratch is private to the wave; the workgroup's other
// waves work on other sites and are never waited for).
template <int WPB>
__device__ __forceinline__ void wave_lds_sync()
{
    if (WPB == 1) {
        __syncthreads();                       // single-wave workgroup: no s_barrier is emitted, and the compiler
                                               // keeps the site loop's register footprint at 76 VGPRs
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

// The whole per-site computation for one wavefront.  `hist` points at 512 class counts.
// comb_list: candidate bases packed 4 bits each in SetBase order; n_comb entries.
// Returns false (and leaves `out` untouched) when the site needs a different NS variant.
// ANY: the variant takes every site whatever its quality spectrum (the remainder launch behind the item engine).
template <int NS, int WPB, bool ANY = false>
__device__ bool lrt_site(const uint32_t *__restrict__ hist, int ref, double min_af,
                         uint32_t comb_list, int n_comb, const QualLut *__restrict__ lut,
                         uint32_t *s_n, uint8_t *s_q, SiteOut &out)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int row = lane >> 4, t = lane & 15;

    // ---- load the histogram, compact each base's non-empty classes (order: ascending qual) ----------
    int cnt_row = 0, depth_lane = 0;
#pragma unroll
    for (int lvl = 0; lvl < 8; ++lvl) {
        const int q = t + 16 * lvl;
        const uint32_t c = hist[row * 128 + q];
        const uint64_t nzmask = __ballot(c != 0);
        const uint32_t rowbits = (uint32_t)(nzmask >> (16 * row)) & 0xFFFFu;
        if (c != 0) {
            const int pos = cnt_row + __popc(rowbits & ((1u << t) - 1u));
            if (BVC_LDS_OK(21, pos, 128)) {
                s_n[row * 128 + pos] = c;
                s_q[row * 128 + pos] = (uint8_t)q;
            }
        }
        cnt_row += __popc(rowbits);
        depth_lane += (int)c;
    }
    const int depth_row = row_sum(depth_lane);
    int depth[4];
    depth[0] = __builtin_amdgcn_readlane(depth_row, 0);
    depth[1] = __builtin_amdgcn_readlane(depth_row, 16);
    depth[2] = __builtin_amdgcn_readlane(depth_row, 32);
    depth[3] = __builtin_amdgcn_readlane(depth_row, 48);
    int maxcnt = max(max(__builtin_amdgcn_readlane(cnt_row, 0), __builtin_amdgcn_readlane(cnt_row, 16)),
                     max(__builtin_amdgcn_readlane(cnt_row, 32), __builtin_amdgcn_readlane(cnt_row, 48)));
    const int nslots = (maxcnt + 15) >> 4;
    // variant gate (wave-uniform): NS = 2 takes 0..2 slots, NS = 4 takes 3..4, NS = 8 takes 5..8
    if (nslots > NS || (!ANY && NS > 2 && nslots <= NS / 2)) return false;
    wave_lds_sync<WPB>();                      // the wave's own LDS writes above, read back below

    Slots<NS> S;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int idx = t + 16 * k;
        S.n[k] = 0.0; S.d[k] = 0.0; S.e[k] = 1.0; S.yp[k] = 1.0;     // empty slot: m = 1, u = 0, weight 0
        if (idx < cnt_row) {
            const int q = s_q[row * 128 + idx];
            S.n[k] = (double)s_n[row * 128 + idx];
            S.e[k] = lut->e[q];
            S.d[k] = lut->a[q] - S.e[k];
        }
    }

    // ---- BaseType::LRT ---------------------------------------------------------------------------
    const int total_i = depth[0] + depth[1] + depth[2] + depth[3];
    const double depth_total = (double)total_i;
    out = SiteOut{};
#pragma unroll
    for (int j = 0; j < 4; ++j) out.depth[j] = depth[j];
    out.depth_total = depth_total;
    if (total_i == 0) return true;                              // src/BaseType.cpp:75
    const double inv_n = 1.0 / depth_total;

    // candidate list: bases of base_comb whose count frequency >= min_af (:77-83)
    uint32_t blist = 0;
    int n = 0;
    for (int c = 0; c < n_comb; ++c) {
        const int b = (comb_list >> (4 * c)) & 3;
        if ((double)pick4(depth, b) / depth_total >= min_af) { blist |= (uint32_t)b << (4 * n); ++n; }
    }
    if (n == 0) return true;                                     // :84

    int passes = 0, fits = 0;
    double base_frq[4] = {0, 0, 0, 0};
    double lr_alt = 0.0, chi = 0.0;
    int status = 0;

    // fit of one subset given as a mask over base codes; returns false when UpdateF skips it (:54)
    auto fit_set = [&](uint32_t setmask, double &loglik, double (&ex)[4]) __attribute__((always_inline)) -> bool {
        int depth_sum = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) depth_sum += ((setmask >> j) & 1u) ? depth[j] : 0;
        double f[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            f[j] = (depth_sum > 0 && ((setmask >> j) & 1u)) ? (double)depth[j] / (double)depth_sum : 0.0;
        const double freq_sum = ((f[0] + f[1]) + f[2]) + f[3];
        if (freq_sum == 0) return false;
        // (wave-uniform) the variant's narrower form when the site leaves its last slots empty
        constexpr int kNarrow = NS == 2 ? 1 : (NS == 4 ? 3 : 6);   // <= 16 values (binned qualities), <= 48, <= 96
        if (kNarrow < NS && nslots <= kNarrow) loglik = em_fit<NS, kNarrow>(S, lane, pick4(f, row), inv_n, ex, passes);
        else loglik = em_fit<NS>(S, lane, pick4(f, row), inv_n, ex, passes);
        fits += 1;
        return true;
    };

    // Level k = n is the full model (:88-90); levels n-1 .. 1 are the nested reduction (:93-110).
    bool full_level = true;
    for (int k = n; k > 0; --k) {
        int ncomb = 0;
        const uint32_t masks = subset_masks(n, k, ncomb);
        int n_fit = 0, i_min = 0;
        double best_chi = 0.0, best_lr = 0.0, best_bp[4] = {0, 0, 0, 0};
        for (int c = 0; c < ncomb; ++c) {
            const uint32_t pm = (masks >> (4 * c)) & 0xFu;
            uint32_t setmask = 0;
            for (int p = 0; p < 4; ++p)
                if ((pm >> p) & 1u) setmask |= 1u << ((blist >> (4 * p)) & 3u);
            double ex[4], ll;
            if (!fit_set(setmask, ll, ex)) continue;
            const double chi_c = 2.0 * (lr_alt - ll);
            if (n_fit == 0 || chi_c < best_chi) {                // std::min_element: first minimum, '<'
                best_chi = chi_c; best_lr = ll; i_min = n_fit;
#pragma unroll
                for (int j = 0; j < 4; ++j) best_bp[j] = ex[j];
            }
            ++n_fit;
        }
        if (n_fit == 0) { status = 1; break; }                   // reference: bp[0] / min_element on empty
        lr_alt = best_lr;
        if (full_level) {
            full_level = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) base_frq[j] = best_bp[j];
            continue;
        }
        chi = best_chi;
        if (chi < kLrtThreshold) {
            // bases = bc[i_min]: indexed over ALL subsets although lr/bp skip zero-coverage ones
            const uint32_t pm = (masks >> (4 * i_min)) & 0xFu;
            uint32_t nl = 0;
            int nn = 0;
            for (int p = 0; p < 4; ++p)
                if ((pm >> p) & 1u) { nl |= ((blist >> (4 * p)) & 3u) << (4 * nn); ++nn; }
            blist = nl;
            n = k;
#pragma unroll
            for (int j = 0; j < 4; ++j) base_frq[j] = best_bp[j];
        } else {
            break;
        }
    }
    if (status == 1 && full_level) { out.status = 1; out.n_passes = passes; out.n_fits = fits; return true; }

    out.status = status;
    out.n_passes = passes;
    out.n_fits = fits;
    out.lr_alt = lr_alt;
    out.chi = chi;
    out.n_kept = n;
#pragma unroll
    for (int j = 0; j < 4; ++j) { out.base_frq[j] = base_frq[j]; out.kept[j] = (j < n) ? (int)((blist >> (4 * j)) & 3u) : 0; }
    int n_alt = 0;
    int a0 = 0, a1 = 0, a2 = 0;
    double g0 = 0.0, g1 = 0.0, g2 = 0.0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {                                // :111-116
        const int b = (blist >> (4 * p)) & 3;
        if (p < n && b != ref && n_alt < 3) {
            const double fr = pick4(base_frq, b);
            if (n_alt == 0) { a0 = b; g0 = fr; } else if (n_alt == 1) { a1 = b; g1 = fr; } else { a2 = b; g2 = fr; }
            ++n_alt;
        }
    }
    out.alt_base[0] = a0; out.alt_base[1] = a1; out.alt_base[2] = a2;
    out.af[0] = g0; out.af[1] = g1; out.af[2] = g2;
    out.n_alt = n_alt;
    if (n_alt > 0) {                                             // :117-135
        const double r = (double)pick4(depth, (int)(blist & 3u)) / depth_total;
        double vq;
        if (n == 1 && depth_total > 10 && r > 0.5) {
            vq = 5000.0;
        } else if (chi <= 0) {
            vq = 0.0;
        } else {
            vq = kVarQualPending;                                // chisf(chi, 1): finished by var_qual_kernel
        }
        out.var_qual = vq;
        out.called = 1;
    }
    return true;
}

__device__ __forceinline__ void store_result(bvc_site_result *dst, const SiteOut &o)
{
    bvc_site_result r;
    r.var_qual = o.var_qual; r.chi = o.chi; r.depth_total = o.depth_total; r.lr_alt = o.lr_alt;
    for (int j = 0; j < 3; ++j) { r.af[j] = o.af[j]; r.alt_base[j] = (int8_t)o.alt_base[j]; }
    for (int j = 0; j < 4; ++j) { r.base_frq[j] = o.base_frq[j]; r.depth[j] = o.depth[j]; r.kept[j] = (int8_t)o.kept[j]; }
    r.n_passes = o.n_passes; r.n_alt = (uint8_t)o.n_alt; r.called = (uint8_t)o.called;
    r.n_kept = (uint8_t)o.n_kept; r.status = (uint8_t)o.status; r.n_fits = (uint8_t)o.n_fits;
    *dst = r;
}

// Workgroups of WPB independent wavefronts (4 by default): the dispatcher spreads a workgroup's waves over the
// CU's four SIMDs, which keeps the SIMDs evenly loaded.  Single-wave workgroups (WPB = 1, kept for A/B runs:
// BVC_EM_WPB) are placed unevenly, and this latency-bound kernel runs at the pace of its most crowded SIMD:
// 0.57 -> 0.48 ms per 4000 sites alone, 1.25 -> 1.19 ms underneath the histogram kernel at 8 waves per CU.
// The wave index is made scalar (readfirstlane): the site state then lives in SGPRs and branches stay scalar;
// derived from threadIdx directly it is "divergent" to the compiler and the kernel needs 116 instead of 76 VGPRs.
template <int NS, int WPB, bool ANY = false>
__global__ __launch_bounds__(64 * WPB) void lrt_kernel(int64_t n_sites, const uint32_t *__restrict__ counts,
                                                 int64_t hist_stride, const int8_t *__restrict__ ref_base,
                                                 double min_af, const QualLut *__restrict__ lut,
                                                 const int8_t *__restrict__ comb,
                                                 const uint8_t *__restrict__ n_comb,
                                                 const uint8_t *__restrict__ taken,
                                                 bvc_site_result *__restrict__ results)
{
    BVC_POISON_LDS();
    __shared__ uint32_t s_n_all[WPB][512];
    __shared__ uint8_t s_q_all[WPB][512];
    // wave-uniform by construction: tell the compiler, so that the site state stays in scalar registers
    const int wave = WPB == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // A bounded number of waves walks the sites: the launcher sizes the grid so that this FP64-bound kernel
    // holds only a few wave slots per SIMD and the HBM-bound histogram kernel of the next tile, which runs
    // at the same time in overlap mode, keeps its occupancy.
    for (int64_t site = (int64_t)blockIdx.x * WPB + wave; site < n_sites; site += (int64_t)gridDim.x * WPB) {
        if (taken && taken[site]) continue;                      // the item engine (em_items.hip) has this site
        uint32_t *s_n = s_n_all[wave];
        uint8_t *s_q = s_q_all[wave];
        uint32_t list = 0x3210u;                                 // default base_comb, src/BaseType.h:79
        int nc = 4;
        if (comb) {
            // entries outside 0..3 are dropped: in the reference depth[b] of such a key is 0, so the min_af
            // filter (src/BaseType.cpp:79) removes it for every min_af > 0
            const int want = min((int)n_comb[site], 4);
            list = 0; nc = 0;
            for (int c = 0; c < want; ++c) {
                const int b = comb[site * 4 + c];
                if ((unsigned)b < 4u) { list |= (uint32_t)b << (4 * nc); ++nc; }
            }
        }
        SiteOut o;
        const bool mine = lrt_site<NS, WPB, ANY>(counts + site * hist_stride, ref_base[site], min_af, list, nc, lut, s_n, s_q, o);
        if (mine && (threadIdx.x & 63) == 0) store_result(results + site, o);
        wave_lds_sync<WPB>();                                    // s_n / s_q are reused by the next site
    }
}

// Caller's --group loop (src/BaseVarC.cpp:617-661): one wavefront per (site, group).
template <int NS, int WPB, bool ANY = false>
__global__ __launch_bounds__(64 * WPB) void lrt_groups_kernel(int64_t n_sites, int n_groups,
                                                        const uint32_t *__restrict__ grp_counts,
                                                        const int8_t *__restrict__ ref_base, double min_af,
                                                        const QualLut *__restrict__ lut,
                                                        const bvc_site_result *__restrict__ overall,
                                                        const uint8_t *__restrict__ taken,
                                                        bvc_group_result *__restrict__ grp_results)
{
    BVC_POISON_LDS();
    __shared__ uint32_t s_n_all[WPB][512];
    __shared__ uint8_t s_q_all[WPB][512];
    // wave-uniform by construction: tell the compiler, so that the site state stays in scalar registers
    const int wave = WPB == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t *s_n = s_n_all[wave];
    uint8_t *s_q = s_q_all[wave];
    const int64_t n_work = n_sites * n_groups;
    for (int64_t w = (int64_t)blockIdx.x * WPB + wave; w < n_work; w += (int64_t)gridDim.x * WPB) {
        if (taken && taken[w]) continue;                         // the item engine (em_items.hip) has this (site, group)
        const int64_t site = w / n_groups;
        const int g = (int)(w % n_groups);
        const uint32_t *hist = grp_counts + (site * (n_groups + 1) + g) * BVC_NCLASS;
        if (!overall[site].called) {
            // Not called overall: no group run at all (:633-636), only the depth columns na:nc:ng:nt (:640).  One
            // variant writes them from a plain sum of the histogram rows; the others have nothing to do here.  (Four
            // sites in five are like this: no compaction, no ballots, and two of the three variants touch no memory.)
            if (NS == 2 || ANY) {
                const int lane = threadIdx.x & 63, row = lane >> 4, t = lane & 15;
                int d = 0;
#pragma unroll
                for (int lvl = 0; lvl < 8; ++lvl) d += (int)hist[row * 128 + t + 16 * lvl];
                d = row_sum(d);
                const int d0 = __builtin_amdgcn_readlane(d, 0), d1 = __builtin_amdgcn_readlane(d, 16);
                const int d2 = __builtin_amdgcn_readlane(d, 32), d3 = __builtin_amdgcn_readlane(d, 48);
                if (lane == 0) {
                    bvc_group_result r;
                    for (int j = 0; j < 3; ++j) r.af[j] = 0.0;
                    r.depth[0] = d0; r.depth[1] = d1; r.depth[2] = d2; r.depth[3] = d3;
                    r.ran = 0; r.present = 0;
                    for (int j = 0; j < 6; ++j) r.pad[j] = 0;
                    grp_results[site * n_groups + g] = r;
                }
            }
            continue;
        }
        const bvc_site_result ov = overall[site];
        const int ref = ref_base[site];
        // base_comb = {ref} + alt_bases (:614-615); a ref outside 0..3 has depth 0 in the reference and falls to
        // the min_af filter, so it is left out here
        uint32_t list = (unsigned)ref < 4u ? (uint32_t)ref : 0u;
        int nc = (unsigned)ref < 4u ? 1 : 0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < ov.n_alt) { list |= (uint32_t)(ov.alt_base[i] & 3) << (4 * nc); ++nc; }
        SiteOut o;
        // called overall: the group's own LRT, when it has covered samples (:641; lrt_site returns at once otherwise)
        const bool mine = lrt_site<NS, WPB, ANY>(hist, ref, min_af, list, nc, lut, s_n, s_q, o);
        if (mine && (threadIdx.x & 63) == 0) {
            bvc_group_result r;
            for (int j = 0; j < 4; ++j) r.depth[j] = o.depth[j];
            for (int j = 0; j < 6; ++j) r.pad[j] = 0;
            r.ran = o.depth_total > 0 ? 1 : 0;
            r.present = 0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                double af = 0.0;                                 // literal 0 when the group lacks the ALT (:650)
                if (r.ran && i < ov.n_alt) {
#pragma unroll
                    for (int tt = 0; tt < 3; ++tt)
                        if (tt < o.n_alt && o.alt_base[tt] == ov.alt_base[i]) { af = o.af[tt]; r.present |= (uint8_t)(1u << i); }
                }
                r.af[i] = af;
            }
            grp_results[site * n_groups + g] = r;
        }
        wave_lds_sync<WPB>();
    }
}

// var_qual = -10 log10(chisf(chi, 1)) for the records lrt_kernel left pending (src/BaseType.cpp:127-133).
__global__ void var_qual_kernel(int64_t n_sites, bvc_site_result *__restrict__ results)
{
    BVC_POISON_LDS();
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_sites) return;
    if (!results[s].called || results[s].var_qual != kVarQualPending) return;
    const double p = kf_gammaq_dev(0.5, results[s].chi / 2.0);   // chisf(chi, 1), src/Algorithm.cpp:3-7
    double vq = (p != 0.0) ? -10 * log10(p) : 10000.0;           // NaN != 0 -> NaN, like `if (chi_prob)`
    if (vq == 0) vq = 0.0;
    results[s].var_qual = vq;
}

__global__ void sum_groups_kernel(int64_t total, int n_hist, const uint32_t *__restrict__ grp_counts,
                                  uint32_t *__restrict__ counts)
{
    BVC_POISON_LDS();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over n_sites * 512
    if (i >= total) return;
    const int64_t site = i / BVC_NCLASS;
    const int key = (int)(i % BVC_NCLASS);
    uint32_t s = 0;
    for (int h = 0; h < n_hist; ++h) s += grp_counts[(site * n_hist + h) * BVC_NCLASS + key];
    counts[i] = s;
}

// Group mode through the item engine: every (site, group) is a pseudo-site with SetBase({ref} + alt_bases) when the
// site was called overall (src/BaseVarC.cpp:614-615, 642-644) and no candidate at all otherwise (:633-636).
__global__ void group_comb_kernel(int64_t n_pseudo, int n_groups, const int8_t *__restrict__ ref_base,
                                  const bvc_site_result *__restrict__ overall, int8_t *__restrict__ comb,
                                  uint8_t *__restrict__ n_comb)
{
    BVC_POISON_LDS();
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pseudo) return;
    const int64_t site = p / n_groups;
    int nc = 0;
    int8_t list[4] = {0, 0, 0, 0};
    if (overall[site].called) {
        const int ref = ref_base[site];
        if ((unsigned)ref < 4u) list[nc++] = (int8_t)ref;
        const int n_alt = overall[site].n_alt;
        for (int i = 0; i < 3; ++i)
            if (i < n_alt) list[nc++] = (int8_t)(overall[site].alt_base[i] & 3);
    }
    for (int c = 0; c < 4; ++c) comb[p * 4 + c] = list[c];
    n_comb[p] = (uint8_t)nc;
}

// The group record of a pseudo-site the item engine took, from its site-style record (:640-652).
__global__ void group_records_kernel(int64_t n_pseudo, int n_groups, const bvc_site_result *__restrict__ overall,
                                     const bvc_site_result *__restrict__ pseudo, const uint8_t *__restrict__ taken,
                                     bvc_group_result *__restrict__ grp_results)
{
    BVC_POISON_LDS();
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pseudo || !taken[p]) return;
    const int64_t site = p / n_groups;
    const bvc_site_result t = pseudo[p];
    bvc_group_result r;
    for (int j = 0; j < 4; ++j) r.depth[j] = t.depth[j];
    for (int j = 0; j < 6; ++j) r.pad[j] = 0;
    for (int j = 0; j < 3; ++j) r.af[j] = 0.0;
    r.ran = 0; r.present = 0;
    if (overall[site].called) {
        r.ran = t.depth_total > 0 ? 1 : 0;
        const int n_alt = overall[site].n_alt;
        for (int i = 0; i < 3; ++i)
            if (r.ran && i < n_alt)
                for (int tt = 0; tt < 3; ++tt)
                    if (tt < t.n_alt && t.alt_base[tt] == overall[site].alt_base[i]) { r.af[i] = t.af[tt]; r.present |= (uint8_t)(1u << i); }
    }
    grp_results[p] = r;
}

}  // namespace

#ifdef BVC_CHECK_LDS
BVC_DEFINE_DEBUG_READER(debug_read_wave_engine)
#endif

// Waves the EM kernels keep on the chip.  `shared` = the launch runs underneath a streaming histogram kernel
// (overlap mode with long rows): 8 per CU = two 4-wave workgroups (swept 4..24 on MI355X at N = 1e6) leaves that
// kernel its wave slots and registers.  Otherwise the kernel has the chip to itself for most of its life and
// takes 24 per CU.  The context's em_waves_per_cu (bvc_set_tuning / BVC_EM_WAVES_PER_CU) overrides both.
//
// Group mode's stage 2 (sum, overall LRT, per-group LRT) is a third longer and its histogram pass slower than the
// plain call's; swept 8..24 on MI355X (k = 5, N = 1e6, round 2 kernels): 8 waves per CU in flight is as good as any
// (2.68e6 sites/s at 8, 2.58-2.65e6 at 12-24) -- and since group calls alternate between two stage-2 streams
// (bvc_api.hip, em_stream) each launch takes half of that: kGroupSharedWavesPerCu in bvc_internal.h.

static int64_t em_grid_cap(const LaunchState &st, bool shared, int shared_waves_per_cu = 0)
{
    int per_cu = shared ? (shared_waves_per_cu > 0 ? shared_waves_per_cu : 8) : 24;
    if (st.em_waves_per_cu > 0) per_cu = st.em_waves_per_cu;
    return (int64_t)per_cu * st.n_cu;
}

template <int WPB>
static void launch_lrt_variants(hipStream_t stream, int64_t want_waves, int64_t n_sites,
                                const uint32_t *counts, int64_t hist_stride, const int8_t *ref_base, double min_af,
                                const QualLut *lut, const int8_t *comb, const uint8_t *n_comb, const uint8_t *taken,
                                bvc_site_result *results)
{
    const dim3 grid((unsigned)((want_waves + WPB - 1) / WPB)), block(64 * WPB);
    if (taken) {
        // behind the item engine: the few sites it left (wide quality spectra, qualities 0 and 1, ...) in ONE launch of
        // the widest variant, which narrows itself to the slots a site fills
        hipLaunchKernelGGL((lrt_kernel<8, WPB, true>), grid, block, 0, stream, n_sites, counts, hist_stride, ref_base,
                           min_af, lut, comb, n_comb, taken, results);
        return;
    }
    // Every variant visits every site; a wave skips a site at once when it belongs to another variant.
    hipLaunchKernelGGL((lrt_kernel<2, WPB>), grid, block, 0, stream, n_sites, counts, hist_stride, ref_base,
                       min_af, lut, comb, n_comb, taken, results);
    hipLaunchKernelGGL((lrt_kernel<4, WPB>), grid, block, 0, stream, n_sites, counts, hist_stride, ref_base, min_af,
                       lut, comb, n_comb, taken, results);
    hipLaunchKernelGGL((lrt_kernel<8, WPB>), grid, block, 0, stream, n_sites, counts, hist_stride, ref_base, min_af,
                       lut, comb, n_comb, taken, results);
}

hipError_t launch_lrt(const LaunchState &st, hipStream_t stream, int64_t n_sites, const uint32_t *counts,
                      int64_t hist_stride, const int8_t *ref_base, double min_af, const QualLut *lut,
                      const int8_t *comb, const uint8_t *n_comb, bvc_site_result *results, bool shared,
                      int shared_waves_per_cu, void *scratch)
{
    if (n_sites <= 0) return hipSuccess;
    // The item engine takes the sites it can (em_items.hip); the kernels below take the rest.  min_af <= 0 lets
    // zero-depth alleles through the filter and UpdateF skip subsets (src/BaseType.cpp:54): left to lrt_site.
    const uint8_t *taken = nullptr;
    if (scratch && st.em_engine != 1 && min_af > 0.0) {
        const hipError_t e = launch_lrt_items(st, stream, n_sites, 0, counts, hist_stride, ref_base, min_af, lut, comb, n_comb,
                                              results, scratch, &taken, shared);
        if (e != hipSuccess) return e;
    }
    const int64_t cap = em_grid_cap(st, shared, shared_waves_per_cu);
    const int64_t want_waves = n_sites < cap ? n_sites : cap;
    if (st.em_wpb == 1) launch_lrt_variants<1>(stream, want_waves, n_sites, counts, hist_stride, ref_base, min_af, lut, comb, n_comb, taken, results);
    else launch_lrt_variants<4>(stream, want_waves, n_sites, counts, hist_stride, ref_base, min_af, lut, comb, n_comb, taken, results);
    hipLaunchKernelGGL(var_qual_kernel, dim3((unsigned)((n_sites + 255) / 256)), dim3(256), 0, stream, n_sites, results);
    return hipGetLastError();
}

template <int WPB>
static void launch_group_variants(hipStream_t stream, int64_t want_waves, int64_t n_sites, int n_groups,
                                  const uint32_t *grp_counts, const int8_t *ref_base, double min_af,
                                  const QualLut *lut, const bvc_site_result *overall, const uint8_t *taken,
                                  bvc_group_result *grp_results)
{
    const dim3 grid((unsigned)((want_waves + WPB - 1) / WPB)), block(64 * WPB);
    if (taken) {                                                 // behind the item engine: what it left, in one launch
        hipLaunchKernelGGL((lrt_groups_kernel<8, WPB, true>), grid, block, 0, stream, n_sites, n_groups, grp_counts,
                           ref_base, min_af, lut, overall, taken, grp_results);
        return;
    }
    hipLaunchKernelGGL((lrt_groups_kernel<2, WPB>), grid, block, 0, stream, n_sites, n_groups, grp_counts, ref_base,
                       min_af, lut, overall, taken, grp_results);
    hipLaunchKernelGGL((lrt_groups_kernel<4, WPB>), grid, block, 0, stream, n_sites, n_groups, grp_counts, ref_base,
                       min_af, lut, overall, taken, grp_results);
    hipLaunchKernelGGL((lrt_groups_kernel<8, WPB>), grid, block, 0, stream, n_sites, n_groups, grp_counts, ref_base,
                       min_af, lut, overall, taken, grp_results);
}

size_t em_group_scratch_bytes(int64_t n_sites, int n_groups)
{
    const size_t n_pseudo = (size_t)n_sites * (size_t)n_groups;
    return ((n_pseudo * 4 + 255) & ~(size_t)255) + ((n_pseudo + 255) & ~(size_t)255) +
           ((n_pseudo * sizeof(bvc_site_result) + 255) & ~(size_t)255) + em_items_scratch_bytes((int64_t)n_pseudo);
}

hipError_t launch_lrt_groups(const LaunchState &st, hipStream_t stream, int64_t n_sites, int n_groups,
                             const uint32_t *grp_counts, const int8_t *ref_base, double min_af, const QualLut *lut,
                             const bvc_site_result *overall, bvc_group_result *grp_results, bool shared,
                             int shared_waves_per_cu, void *scratch)
{
    if (n_sites <= 0 || n_groups <= 0) return hipSuccess;
    const int64_t cap = em_grid_cap(st, shared, shared_waves_per_cu);
    const int64_t n_work = n_sites * n_groups;
    const int64_t want_waves = n_work < cap ? n_work : cap;
    const uint8_t *taken = nullptr;
    if (scratch && st.em_engine != 1 && min_af > 0.0) {
        // every (site, group) a pseudo-site of the item engine; its records become group records afterwards
        char *p = static_cast<char *>(scratch);
        auto take = [&](size_t bytes) { char *q = p; p += (bytes + 255) & ~(size_t)255; return q; };
        int8_t *comb = reinterpret_cast<int8_t *>(take((size_t)n_work * 4));
        uint8_t *n_comb = reinterpret_cast<uint8_t *>(take((size_t)n_work));
        bvc_site_result *pseudo = reinterpret_cast<bvc_site_result *>(take((size_t)n_work * sizeof(bvc_site_result)));
        const dim3 tgrid((unsigned)((n_work + 255) / 256)), tblock(256);
        hipLaunchKernelGGL(group_comb_kernel, tgrid, tblock, 0, stream, n_work, n_groups, ref_base, overall, comb, n_comb);
        const hipError_t e = launch_lrt_items(st, stream, n_work, n_groups, grp_counts, BVC_NCLASS, ref_base, min_af, lut, comb,
                                              n_comb, pseudo, p, &taken, shared);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(group_records_kernel, tgrid, tblock, 0, stream, n_work, n_groups, overall, pseudo, taken, grp_results);
    }
    if (st.em_wpb == 1) launch_group_variants<1>(stream, want_waves, n_sites, n_groups, grp_counts, ref_base, min_af, lut, overall, taken, grp_results);
    else launch_group_variants<4>(stream, want_waves, n_sites, n_groups, grp_counts, ref_base, min_af, lut, overall, taken, grp_results);
    return hipGetLastError();
}

hipError_t launch_sum_groups(hipStream_t stream, int64_t n_sites, int n_hist, const uint32_t *grp_counts,
                             uint32_t *counts)
{
    const int64_t total = n_sites * BVC_NCLASS;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(sum_groups_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, total,
                       n_hist, grp_counts, counts);
    return hipGetLastError();
}

}  // namespace bvc
