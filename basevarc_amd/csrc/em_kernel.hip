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
#include <atomic>
#include <cstdlib>

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

template <int NS>
__device__ __forceinline__ PassOut em_pass(Slots<NS> &S, const Freq f, double inv_n, int lane)
{
    double acc_d = 0.0, acc_e = 0.0, acc_a;
    double m[NS], u[NS];
    double umax = 0.0;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
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
    for (int k = 0; k < NS; ++k) {
        double y = fma(-S.yp[k], u[k], S.yp[k]);
        y = fma(y, fma(-m[k], y, 1.0), y);
        S.yp[k] = fma(y, fma(-m[k], y, 1.0), y);
        asm volatile("" : "+v"(S.yp[k]));                       // keeps these steps ahead of the branch below
    }
    if (jump) {
#pragma unroll
        for (int k = 0; k < NS; ++k) S.yp[k] = fast_rcp(m[k]);
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) {
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
    const DPair w = swap16(v, v);
    const double t = w.a + w.b;                                  // lanes 8-15: rows 0,1 -> E, rows 2,3 -> A
    const double etot = lane_value<8>(t);
    const uint32_t a_hi = (uint32_t)__builtin_amdgcn_readlane(__double2hiint(t), 40);
    PassOut o;
    o.converged = a_hi < kSureBelowHi;
    if (a_hi >= kSureBelowHi && a_hi < kSureAboveHi) {           // rare: the bracket straddles eps
        double ex = 0.0;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
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
template <int NS>
__device__ __forceinline__ double em_fit(Slots<NS> &S, int lane, double f0, double inv_n,
                                         double (&ex)[4], int &passes)
{
    Freq f{f0};
    PassOut o;
    // every fit starts from yp = 1 (not from the previous fit's, which may hold NaNs): its first pass then
    // has |u| = |m - 1| and takes the big tier, whose reciprocal does not depend on yp
#pragma unroll
    for (int k = 0; k < NS; ++k) S.yp[k] = 1.0;
    // pass 0 + at most kEmIters update passes; two passes per trip so that 1/m can alternate between two register
    // sets instead of being copied back every pass
    for (int it = 0;; it += 2) {
        o = em_pass<NS>(S, f, inv_n, lane);
        passes += 1;
        if (it > 0 && o.converged) break;
        if (it == kEmIters) break;
        f.fb = o.ex_own;
        o = em_pass<NS>(S, f, inv_n, lane);
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
    for (int k = 0; k < NS; ++k) ll = fma(-S.n[k], log_pos(S.yp[k]), ll);
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
template <int NS, int WPB>
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
            s_n[row * 128 + pos] = c;
            s_q[row * 128 + pos] = (uint8_t)q;
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
    if (nslots > NS || (NS > 2 && nslots <= NS / 2)) return false;
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
        loglik = em_fit<NS>(S, lane, pick4(f, row), inv_n, ex, passes);
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
template <int NS, int WPB>
__global__ __launch_bounds__(64 * WPB) void lrt_kernel(int64_t n_sites, const uint32_t *__restrict__ counts,
                                                 int64_t hist_stride, const int8_t *__restrict__ ref_base,
                                                 double min_af, const QualLut *__restrict__ lut,
                                                 const int8_t *__restrict__ comb,
                                                 const uint8_t *__restrict__ n_comb,
                                                 bvc_site_result *__restrict__ results)
{
    __shared__ uint32_t s_n_all[WPB][512];
    __shared__ uint8_t s_q_all[WPB][512];
    // wave-uniform by construction: tell the compiler, so that the site state stays in scalar registers
    const int wave = WPB == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // A bounded number of waves walks the sites: the launcher sizes the grid so that this FP64-bound kernel
    // holds only a few wave slots per SIMD and the HBM-bound histogram kernel of the next tile, which runs
    // at the same time in overlap mode, keeps its occupancy.
    for (int64_t site = (int64_t)blockIdx.x * WPB + wave; site < n_sites; site += (int64_t)gridDim.x * WPB) {
        uint32_t *s_n = s_n_all[wave];
        uint8_t *s_q = s_q_all[wave];
        uint32_t list = 0x3210u;                                 // default base_comb, src/BaseType.h:79
        int nc = 4;
        if (comb) {
            nc = min((int)n_comb[site], 4);
            list = 0;
            for (int c = 0; c < nc; ++c) list |= (uint32_t)(comb[site * 4 + c] & 3) << (4 * c);
        }
        SiteOut o;
        const bool mine = lrt_site<NS, WPB>(counts + site * hist_stride, ref_base[site], min_af, list, nc, lut, s_n, s_q, o);
        if (mine && (threadIdx.x & 63) == 0) store_result(results + site, o);
        wave_lds_sync<WPB>();                                    // s_n / s_q are reused by the next site
    }
}

// =====================================================================================================
// Rows layout: FOUR sites per wavefront.  Each DPP row of 16 lanes runs its own site through its own
// fit / iteration state; the four rows share the instruction stream of the E+M pass, whose cross-lane
// reductions then cost a quarter per site (quad sums for D, one packed row sum for E and delta, no
// cross-row step at all).  Within a row, quad b (4 lanes) owns base b and each lane up to 8 classes, so a row
// holds sites with at most 32 quality values per base -- the sites lrt_kernel<2> would take; sites with more
// are left to lrt_kernel<4>/<8>.  Rows are not in lockstep: a row that finishes a fit moves on at once; only
// the short transition code (log-likelihood, subset bookkeeping, result) runs with the other rows masked.
// =====================================================================================================
constexpr int kRowSlots = 8;
constexpr double kFarU = 0.001953125;      // 2^-9: a row with a class at or above it cannot have converged
constexpr double kNotConverged = 1.0;      // any value >= kEmEpsilon

template <int NSL>
__device__ __forceinline__ void rows_pass(const double (&sn)[kRowSlots], const double (&sd)[kRowSlots],
                                          const double (&se)[kRowSlots], double (&syp)[kRowSlots], double fb,
                                          double inv_n, int lane, double &ex_own, double &delta)
{
    const int row = lane >> 4;
    double m[NSL], u[NSL], y[NSL];
    double umax = 0.0;
#pragma unroll
    for (int k = 0; k < NSL; ++k) {
        m[k] = fma(fb, sd[k], se[k]);
        u[k] = fma(m[k], syp[k], -1.0);
        umax = fmax(umax, fabs(u[k]));
    }
    const uint64_t far_m = __ballot(umax >= kFarU), jump_m = __ballot(umax > kLog1pMaxU);
    const bool far_row = ((far_m >> (16 * row)) & 0xFFFFull) != 0;
    const bool jump_row = ((jump_m >> (16 * row)) & 0xFFFFull) != 0;
    double acc_d = 0.0, acc_e = 0.0, acc_delta = 0.0;
#pragma unroll
    for (int k = 0; k < NSL; ++k) {
        double p = fma(-0.25, u[k], 1.0 / 3.0);
        p = fma(p, u[k], -0.5);
        p = fma(p, u[k], 1.0);
        acc_delta = fma(sn[k], fabs(u[k] * p), acc_delta);
        double t = fma(-syp[k], u[k], syp[k]);
        t = fma(t, fma(-m[k], t, 1.0), t);
        y[k] = fma(t, fma(-m[k], t, 1.0), t);
    }
    if (jump_m != 0) {                                          // some row is in the first passes of a fit
#pragma unroll
        for (int k = 0; k < NSL; ++k) {
            const double yj = fast_rcp(m[k]);
            y[k] = jump_row ? yj : y[k];
        }
    }
#pragma unroll
    for (int k = 0; k < NSL; ++k) {
        syp[k] = y[k];
        const double r = sn[k] * y[k];
        acc_d = fma(r, sd[k], acc_d);
        acc_e = fma(r, se[k], acc_e);
    }
    // D: sum over the quad (the lane's base).  E and delta: lanes 0-7 of the row end with E, lanes 8-15 with delta.
    const bool hi = (lane & 8) != 0;
    const double send = hi ? acc_e : acc_delta, keep = hi ? acc_delta : acc_e;
    double z = keep + dpp_f64<kDppRor8>(send);
    double dq = acc_d;
    double tz = dpp_f64<kDppXor1>(z), td = dpp_f64<kDppXor1>(dq);
    z += tz; dq += td;
    tz = dpp_f64<kDppXor2>(z); td = dpp_f64<kDppXor2>(dq);
    z += tz; dq += td;
    z += dpp_f64<kDppHalfMirror>(z);
    // lanes 0-7 hold sum(keep of lanes 0-7) + sum(send of lanes 8-15) = E of the row; lanes 8-15 hold delta
    const double e_row = dpp_f64<dpp_newbcast<0>()>(z);
    const double d_row = dpp_f64<dpp_newbcast<8>()>(z);
    ex_own = fb * inv_n * (dq + e_row);
    delta = far_row ? kNotConverged : d_row;
}

__global__ __launch_bounds__(64) void lrt_rows_kernel(int64_t n_sites, const uint32_t *__restrict__ counts,
                                                      int64_t hist_stride, const int8_t *__restrict__ ref_base,
                                                      double min_af, const QualLut *__restrict__ lut,
                                                      const int8_t *__restrict__ comb,
                                                      const uint8_t *__restrict__ n_comb,
                                                      bvc_site_result *__restrict__ results)
{
    __shared__ uint32_t s_n[4][4][32];           // [row][base][rank] non-empty classes in ascending quality
    __shared__ uint8_t s_q[4][4][32];
    enum { P_FETCH = 0, P_LEVEL = 1, P_EM = 2, P_DONE = 3 };
    const int lane = threadIdx.x & 63;
    const int row = lane >> 4, base = (lane >> 2) & 3, sub = lane & 3;
    const bool writer = (lane & 15) == 0;
    const int64_t n_rows = (int64_t)gridDim.x * 4;
    int64_t site = (int64_t)blockIdx.x * 4 + row - n_rows;       // advanced before use

    double sn[kRowSlots], sd[kRowSlots], se[kRowSlots], syp[kRowSlots];
    int phase = P_FETCH, rslots = 0;
    int dep0 = 0, dep1 = 0, dep2 = 0, dep3 = 0, ref = 0;
    double inv_n = 0.0, depth_total = 0.0;
    int n = 0, k = 0, c = 0, ncomb = 0, n_fit = 0, i_min = 0, status = 0, passes = 0, fits = 0, it = 0;
    bool full_level = true;
    uint32_t blist = 0, masks = 0;
    double best_chi = 0.0, best_lr = 0.0, bb0 = 0, bb1 = 0, bb2 = 0, bb3 = 0;
    double lr_alt = 0.0, chi = 0.0, bf0 = 0, bf1 = 0, bf2 = 0, bf3 = 0;
    double fb = 0.0, ex_own = 0.0;
#pragma unroll
    for (int j = 0; j < kRowSlots; ++j) { sn[j] = 0.0; sd[j] = 0.0; se[j] = 1.0; syp[j] = 1.0; }

    auto depth_of = [&](int b) { return b == 0 ? dep0 : (b == 1 ? dep1 : (b == 2 ? dep2 : dep3)); };

    for (;;) {
        // ---------------------------------------------------------------- transitions (rows diverge here)
        while (phase == P_FETCH || phase == P_LEVEL) {
            if (phase == P_FETCH) {
                site += n_rows;
                if (site >= n_sites) { phase = P_DONE; break; }
                // this lane's 32 counts: qualities [32 sub, 32 sub + 32) of its base
                const uint32_t *h = counts + site * hist_stride + base * 128 + sub * 32;
                uint32_t cv[32];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(h + 4 * j);
                    cv[4 * j] = v.x; cv[4 * j + 1] = v.y; cv[4 * j + 2] = v.z; cv[4 * j + 3] = v.w;
                }
                int cnt_l = 0, dep_l = 0;
#pragma unroll
                for (int j = 0; j < 32; ++j) { cnt_l += cv[j] != 0; dep_l += (int)cv[j]; }
                // counts of the quad's four lanes, by quad broadcasts
                const int c0 = dpp_i32<0x00>(cnt_l), c1 = dpp_i32<0x55>(cnt_l), c2 = dpp_i32<0xAA>(cnt_l), c3 = dpp_i32<0xFF>(cnt_l);
                const int cnt_b = c0 + c1 + c2 + c3;
                const int prefix = (sub > 0 ? c0 : 0) + (sub > 1 ? c1 : 0) + (sub > 2 ? c2 : 0);
                int dep_b = dep_l + dpp_i32<kDppXor1>(dep_l);
                dep_b += dpp_i32<kDppXor2>(dep_b);
                dep0 = dpp_i32<dpp_newbcast<0>()>(dep_b); dep1 = dpp_i32<dpp_newbcast<4>()>(dep_b);
                dep2 = dpp_i32<dpp_newbcast<8>()>(dep_b); dep3 = dpp_i32<dpp_newbcast<12>()>(dep_b);
                const int m01 = max(dpp_i32<dpp_newbcast<0>()>(cnt_b), dpp_i32<dpp_newbcast<4>()>(cnt_b));
                const int m23 = max(dpp_i32<dpp_newbcast<8>()>(cnt_b), dpp_i32<dpp_newbcast<12>()>(cnt_b));
                const int maxcnt = max(m01, m23);
                if (maxcnt > 32) continue;                           // lrt_kernel<4>/<8> take this site
                int pos = prefix;
#pragma unroll
                for (int j = 0; j < 32; ++j)
                    if (cv[j] != 0) {
                        if (pos < 32) { s_n[row][base][pos] = cv[j]; s_q[row][base][pos] = (uint8_t)(sub * 32 + j); }
                        ++pos;
                    }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                rslots = (maxcnt + 3) >> 2;
#pragma unroll
                for (int j = 0; j < kRowSlots; ++j) {
                    const int idx = sub + 4 * j;
                    sn[j] = 0.0; sd[j] = 0.0; se[j] = 1.0; syp[j] = 1.0;
                    if (idx < cnt_b) {
                        const int q = s_q[row][base][idx];
                        sn[j] = (double)s_n[row][base][idx];
                        se[j] = lut->e[q];
                        sd[j] = lut->a[q] - se[j];
                    }
                }
                __builtin_amdgcn_wave_barrier();
                ref = ref_base[site];
                const int total_i = dep0 + dep1 + dep2 + dep3;
                depth_total = (double)total_i;
                passes = 0; fits = 0; status = 0; lr_alt = 0.0; chi = 0.0;
                bf0 = bf1 = bf2 = bf3 = 0.0;
                blist = 0; n = 0;
                if (total_i > 0) {
                    inv_n = 1.0 / depth_total;
                    uint32_t list = 0x3210u;                         // default base_comb, src/BaseType.h:79
                    int nc = 4;
                    if (comb) {
                        nc = min((int)n_comb[site], 4);
                        list = 0;
                        for (int t = 0; t < nc; ++t) list |= (uint32_t)(comb[site * 4 + t] & 3) << (4 * t);
                    }
                    for (int t = 0; t < nc; ++t) {                   // candidates with count frequency >= min_af
                        const int b = (list >> (4 * t)) & 3;
                        if ((double)depth_of(b) / depth_total >= min_af) { blist |= (uint32_t)b << (4 * n); ++n; }
                    }
                }
                if (n == 0) {                                        // depth_total == 0 or no candidate: no call
                    if (writer) {
                        SiteOut o = SiteOut{};
                        o.depth[0] = dep0; o.depth[1] = dep1; o.depth[2] = dep2; o.depth[3] = dep3;
                        o.depth_total = depth_total;
                        store_result(results + site, o);
                    }
                    continue;                                        // next site
                }
                k = n; full_level = true; c = 0; n_fit = 0; i_min = 0;
                masks = subset_masks(n, k, ncomb);
                phase = P_LEVEL;
            } else {
                if (c < ncomb) {                                     // next subset of this level
                    const uint32_t pm = (masks >> (4 * c)) & 0xFu;
                    ++c;
                    uint32_t setmask = 0;
                    for (int p = 0; p < 4; ++p)
                        if ((pm >> p) & 1u) setmask |= 1u << ((blist >> (4 * p)) & 3u);
                    int depth_sum = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) depth_sum += ((setmask >> j) & 1u) ? depth_of(j) : 0;
                    if (depth_sum <= 0) continue;                    // freq_sum == 0: UpdateF skips it (:54)
                    fb = ((setmask >> base) & 1u) ? (double)depth_of(base) / (double)depth_sum : 0.0;
#pragma unroll
                    for (int j = 0; j < kRowSlots; ++j) syp[j] = 1.0;
                    it = 0;
                    phase = P_EM;
                } else {                                             // level finished (:96-109)
                    bool finish = false;
                    if (n_fit == 0) { status = 1; finish = true; }
                    else {
                        lr_alt = best_lr;
                        if (full_level) { full_level = false; bf0 = bb0; bf1 = bb1; bf2 = bb2; bf3 = bb3; }
                        else {
                            chi = best_chi;
                            if (chi < kLrtThreshold) {
                                const uint32_t pm = (masks >> (4 * i_min)) & 0xFu;
                                uint32_t nl = 0;
                                int nn = 0;
                                for (int p = 0; p < 4; ++p)
                                    if ((pm >> p) & 1u) { nl |= ((blist >> (4 * p)) & 3u) << (4 * nn); ++nn; }
                                blist = nl; n = k;
                                bf0 = bb0; bf1 = bb1; bf2 = bb2; bf3 = bb3;
                            } else {
                                finish = true;
                            }
                        }
                        if (!finish) {
                            --k;
                            if (k == 0) finish = true;
                            else { masks = subset_masks(n, k, ncomb); c = 0; n_fit = 0; i_min = 0; }
                        }
                    }
                    if (finish) {
                        if (writer) {
                            SiteOut o = SiteOut{};
                            o.depth[0] = dep0; o.depth[1] = dep1; o.depth[2] = dep2; o.depth[3] = dep3;
                            o.depth_total = depth_total;
                            o.status = status; o.n_passes = passes; o.n_fits = fits;
                            const bool no_model = status == 1 && full_level;     // bp[0] on an empty vector
                            if (!no_model) {
                                o.lr_alt = lr_alt; o.chi = chi; o.n_kept = n;
                                o.base_frq[0] = bf0; o.base_frq[1] = bf1; o.base_frq[2] = bf2; o.base_frq[3] = bf3;
                                int n_alt = 0, a0 = 0, a1 = 0, a2 = 0;
                                double g0 = 0.0, g1 = 0.0, g2 = 0.0;
#pragma unroll
                                for (int p = 0; p < 4; ++p) {
                                    const int b = (blist >> (4 * p)) & 3;
                                    o.kept[p] = (p < n) ? b : 0;
                                    if (p < n && b != ref && n_alt < 3) {
                                        const double fr = b == 0 ? bf0 : (b == 1 ? bf1 : (b == 2 ? bf2 : bf3));
                                        if (n_alt == 0) { a0 = b; g0 = fr; } else if (n_alt == 1) { a1 = b; g1 = fr; } else { a2 = b; g2 = fr; }
                                        ++n_alt;
                                    }
                                }
                                o.alt_base[0] = a0; o.alt_base[1] = a1; o.alt_base[2] = a2;
                                o.af[0] = g0; o.af[1] = g1; o.af[2] = g2;
                                o.n_alt = n_alt;
                                if (n_alt > 0) {
                                    const double r = (double)depth_of((int)(blist & 3u)) / depth_total;
                                    double vq;
                                    if (n == 1 && depth_total > 10 && r > 0.5) vq = 5000.0;
                                    else if (chi <= 0) vq = 0.0;
                                    else vq = kVarQualPending;
                                    o.var_qual = vq;
                                    o.called = 1;
                                }
                            }
                            store_result(results + site, o);
                        }
                        phase = P_FETCH;
                    }
                }
            }
        }
        if (__ballot(phase == P_EM) == 0) break;                     // every row has run out of sites

        // ---------------------------------------------------------------- one E+M pass for the rows inside a fit
        // slots evaluated: the largest count among the rows currently fitting (wave-uniform)
        const int act = (phase == P_EM) ? rslots : 0;
        const int wslots = max(max(__builtin_amdgcn_readlane(act, 0), __builtin_amdgcn_readlane(act, 16)),
                               max(__builtin_amdgcn_readlane(act, 32), __builtin_amdgcn_readlane(act, 48)));
        if (phase == P_EM) {
            double delta;
            if (wslots <= 2) rows_pass<2>(sn, sd, se, syp, fb, inv_n, lane, ex_own, delta);
            else if (wslots <= 4) rows_pass<4>(sn, sd, se, syp, fb, inv_n, lane, ex_own, delta);
            else rows_pass<kRowSlots>(sn, sd, se, syp, fb, inv_n, lane, ex_own, delta);
            passes += 1;
            const bool done = (it > 0 && delta < kEmEpsilon) || it == kEmIters;   // NaN never converges
            if (!done) { fb = ex_own; ++it; }
            else {
                // fit finished (UpdateF, src/BaseType.cpp:58-68): log-likelihood and the four expected frequencies
                double ll = 0.0;
#pragma unroll
                for (int j = 0; j < kRowSlots; ++j) ll = fma(-sn[j], log_pos(syp[j]), ll);   // empty slots: n = 0, yp = 1
                ll = row_sum(ll);
                const double e0 = dpp_f64<dpp_newbcast<0>()>(ex_own), e1 = dpp_f64<dpp_newbcast<4>()>(ex_own);
                const double e2 = dpp_f64<dpp_newbcast<8>()>(ex_own), e3 = dpp_f64<dpp_newbcast<12>()>(ex_own);
                const double chi_c = 2.0 * (lr_alt - ll);
                if (n_fit == 0 || chi_c < best_chi) {                // std::min_element: first minimum, '<'
                    best_chi = chi_c; best_lr = ll; i_min = n_fit;
                    bb0 = e0; bb1 = e1; bb2 = e2; bb3 = e3;
                }
                ++n_fit; ++fits;
                phase = P_LEVEL;
            }
        }
    }
}

// Caller's --group loop (src/BaseVarC.cpp:617-661): one wavefront per (site, group).
template <int NS, int WPB>
__global__ __launch_bounds__(64 * WPB) void lrt_groups_kernel(int64_t n_sites, int n_groups,
                                                        const uint32_t *__restrict__ grp_counts,
                                                        const int8_t *__restrict__ ref_base, double min_af,
                                                        const QualLut *__restrict__ lut,
                                                        const bvc_site_result *__restrict__ overall,
                                                        bvc_group_result *__restrict__ grp_results)
{
    __shared__ uint32_t s_n_all[WPB][512];
    __shared__ uint8_t s_q_all[WPB][512];
    // wave-uniform by construction: tell the compiler, so that the site state stays in scalar registers
    const int wave = WPB == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t *s_n = s_n_all[wave];
    uint8_t *s_q = s_q_all[wave];
    const int64_t n_work = n_sites * n_groups;
    for (int64_t w = (int64_t)blockIdx.x * WPB + wave; w < n_work; w += (int64_t)gridDim.x * WPB) {
        const int64_t site = w / n_groups;
        const int g = (int)(w % n_groups);
        const uint32_t *hist = grp_counts + (site * (n_groups + 1) + g) * BVC_NCLASS;
        const bvc_site_result ov = overall[site];
        const int ref = ref_base[site];
        uint32_t list = (uint32_t)(ref & 3);                     // base_comb = {ref} + alt_bases (:614-615)
        int nc = 1;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < ov.n_alt) { list |= (uint32_t)(ov.alt_base[i] & 3) << (4 * nc); ++nc; }
        SiteOut o;
        // The histogram is always loaded (depths are reported for every group, :640); the LRT itself runs only
        // when the overall call succeeded and the group has covered samples (:633-636, :641).
        const bool mine = lrt_site<NS, WPB>(hist, ref, min_af, list, ov.called ? nc : 0, lut, s_n, s_q, o);
        if (mine && (threadIdx.x & 63) == 0) {
            bvc_group_result r;
            for (int j = 0; j < 4; ++j) r.depth[j] = o.depth[j];
            for (int j = 0; j < 6; ++j) r.pad[j] = 0;
            r.ran = (ov.called && o.depth_total > 0) ? 1 : 0;
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
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over n_sites * 512
    if (i >= total) return;
    const int64_t site = i / BVC_NCLASS;
    const int key = (int)(i % BVC_NCLASS);
    uint32_t s = 0;
    for (int h = 0; h < n_hist; ++h) s += grp_counts[(site * n_hist + h) * BVC_NCLASS + key];
    counts[i] = s;
}

}  // namespace

// -1 (default) and 0: one site per wave; 1: four sites per wave for the <= 32-class sites (BVC_EM_ROWS or bvc_set_tuning)
static std::atomic<int> g_em_rows_mode{[] { const char *e = getenv("BVC_EM_ROWS"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }()};
static std::atomic<int> g_em_waves_per_cu{[] { const char *e = getenv("BVC_EM_WAVES_PER_CU"); const int v = e ? atoi(e) : 0; return (v > 0 && v <= 32) ? v : 0; }()};

// waves per EM workgroup: 0 = default (4), 1 or 4 forced (BVC_EM_WPB; experiments only)
static std::atomic<int> g_em_wpb{[] { const char *e = getenv("BVC_EM_WPB"); const int v = e ? atoi(e) : 0; return (v == 1 || v == 4) ? v : 0; }()};

void set_em_tuning(int rows_mode, int waves_per_cu)
{
    if (rows_mode >= -1 && rows_mode <= 1) g_em_rows_mode.store(rows_mode);
    if (waves_per_cu >= 0 && waves_per_cu <= 32) g_em_waves_per_cu.store(waves_per_cu);
}

// Waves the EM kernels keep on the chip.  `shared` = the launch runs underneath a streaming histogram kernel
// (overlap mode with long rows): 8 per CU = two 4-wave workgroups (swept 4..24 on MI355X at N = 1e6) leaves that kernel its wave slots and registers.
// Otherwise the kernel has the chip to itself for most of its life and takes 24 per CU.
// BVC_EM_WAVES_PER_CU / bvc_set_tuning override both.

// Group mode's stage 2 (sum, overall LRT, per-group LRT) is a third longer and its histogram pass slower than the
// plain call's: 12 waves per CU balance the two streams there (swept 4..32 on MI355X, k = 5, N = 1e6).
constexpr int kGroupSharedWavesPerCu = 12;

static int64_t em_grid_cap(bool shared, int shared_waves_per_cu = 0)
{
    static std::atomic<int> n_cu_dev[kMaxDevices];
    std::atomic<int> &n_cu_a = n_cu_dev[current_device_slot()];
    int n_cu = n_cu_a.load();
    if (n_cu == 0) {
        n_cu = 256;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t p;
            if (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) n_cu = p.multiProcessorCount;
        }
        n_cu_a.store(n_cu);
    }
    int per_cu = shared ? (shared_waves_per_cu > 0 ? shared_waves_per_cu : 8) : 24;
    if (g_em_waves_per_cu.load() > 0) per_cu = g_em_waves_per_cu.load();
    return (int64_t)per_cu * n_cu;
}

template <int WPB>
static void launch_lrt_variants(hipStream_t stream, int64_t want_waves, bool skip2, int64_t n_sites,
                                const uint32_t *counts, int64_t hist_stride, const int8_t *ref_base, double min_af,
                                const QualLut *lut, const int8_t *comb, const uint8_t *n_comb,
                                bvc_site_result *results)
{
    // Every variant visits every site; a wave skips a site at once when it belongs to another variant.
    const dim3 grid((unsigned)((want_waves + WPB - 1) / WPB)), block(64 * WPB);
    if (!skip2)
        hipLaunchKernelGGL((lrt_kernel<2, WPB>), grid, block, 0, stream, n_sites, counts, hist_stride, ref_base,
                           min_af, lut, comb, n_comb, results);
    hipLaunchKernelGGL((lrt_kernel<4, WPB>), grid, block, 0, stream, n_sites, counts, hist_stride, ref_base, min_af,
                       lut, comb, n_comb, results);
    hipLaunchKernelGGL((lrt_kernel<8, WPB>), grid, block, 0, stream, n_sites, counts, hist_stride, ref_base, min_af,
                       lut, comb, n_comb, results);
}

hipError_t launch_lrt(hipStream_t stream, int64_t n_sites, const uint32_t *counts, int64_t hist_stride,
                      const int8_t *ref_base, double min_af, const QualLut *lut,
                      const int8_t *comb, const uint8_t *n_comb, bvc_site_result *results, bool shared,
                      int64_t depth_hint, int shared_waves_per_cu)
{
    if (n_sites <= 0) return hipSuccess;
    const int64_t cap = em_grid_cap(shared, shared_waves_per_cu);
    const int64_t want_waves = n_sites < cap ? n_sites : cap;
    // Layout of the common (<= 32 classes per base) sites: one site per wave, or four (rows).  Four per wave quarter
    // the reduction work per site but need four times the sites to fill the chip and 248 registers per lane.  The
    // rows layout was the faster one for deep tiles of >= 12,288 sites until the site-per-wave kernel got its even
    // SIMD load and its shorter pass; measured since (MI355X, serial mode, N = 1e6): 16,000 sites 1.66 vs 1.69 ms,
    // 32,000 sites 3.33 vs 3.31 ms, and N = 1e4, 40,000 sites 2.58 vs 2.87 ms -- so it is used only on request
    // (bvc_set_tuning("em_rows", 1) / BVC_EM_ROWS=1) and stays as the A/B alternative.
    (void)depth_hint;
    const bool rows = g_em_rows_mode.load() > 0;
    if (rows) {
        // four sites per wave: a quarter of the waves hold the same number of sites in flight
        const int64_t want = (n_sites + 3) / 4;
        const dim3 rgrid((unsigned)(want < cap ? want : cap));
        hipLaunchKernelGGL(lrt_rows_kernel, rgrid, dim3(64), 0, stream, n_sites, counts, hist_stride,
                           ref_base, min_af, lut, comb, n_comb, results);
    }
    const int wpb = g_em_wpb.load() ? g_em_wpb.load() : 4;
    if (wpb == 1) launch_lrt_variants<1>(stream, want_waves, rows, n_sites, counts, hist_stride, ref_base, min_af, lut, comb, n_comb, results);
    else launch_lrt_variants<4>(stream, want_waves, rows, n_sites, counts, hist_stride, ref_base, min_af, lut, comb, n_comb, results);
    hipLaunchKernelGGL(var_qual_kernel, dim3((unsigned)((n_sites + 255) / 256)), dim3(256), 0, stream, n_sites, results);
    return hipGetLastError();
}

template <int WPB>
static void launch_group_variants(hipStream_t stream, int64_t want_waves, int64_t n_sites, int n_groups,
                                  const uint32_t *grp_counts, const int8_t *ref_base, double min_af,
                                  const QualLut *lut, const bvc_site_result *overall, bvc_group_result *grp_results)
{
    const dim3 grid((unsigned)((want_waves + WPB - 1) / WPB)), block(64 * WPB);
    hipLaunchKernelGGL((lrt_groups_kernel<2, WPB>), grid, block, 0, stream, n_sites, n_groups, grp_counts, ref_base,
                       min_af, lut, overall, grp_results);
    hipLaunchKernelGGL((lrt_groups_kernel<4, WPB>), grid, block, 0, stream, n_sites, n_groups, grp_counts, ref_base,
                       min_af, lut, overall, grp_results);
    hipLaunchKernelGGL((lrt_groups_kernel<8, WPB>), grid, block, 0, stream, n_sites, n_groups, grp_counts, ref_base,
                       min_af, lut, overall, grp_results);
}

hipError_t launch_lrt_groups(hipStream_t stream, int64_t n_sites, int n_groups, const uint32_t *grp_counts,
                             const int8_t *ref_base, double min_af, const QualLut *lut,
                             const bvc_site_result *overall, bvc_group_result *grp_results, bool shared)
{
    if (n_sites <= 0 || n_groups <= 0) return hipSuccess;
    const int64_t cap = em_grid_cap(shared, kGroupSharedWavesPerCu);
    const int64_t n_work = n_sites * n_groups;
    const int64_t want_waves = n_work < cap ? n_work : cap;
    const int wpb = g_em_wpb.load() ? g_em_wpb.load() : 4;
    if (wpb == 1) launch_group_variants<1>(stream, want_waves, n_sites, n_groups, grp_counts, ref_base, min_af, lut, overall, grp_results);
    else launch_group_variants<4>(stream, want_waves, n_sites, n_groups, grp_counts, ref_base, min_af, lut, overall, grp_results);
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
