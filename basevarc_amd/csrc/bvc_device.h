// bvc_device.h -- wave64 device helpers for the basetype kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bvc {

constexpr int kWave = 64;          // CDNA wavefront

// ---- diagnostic builds (never the product) ---------------------------------------------------------------------------
// -DBVC_CHECK_LDS  every LDS address or index that is derived from DATA (input bytes, LDS contents) is checked against the
//                  workgroup's LDS allocation / the array it indexes.  A violation is RECORDED -- (check id, value, limit,
//                  blockIdx.x, threadIdx.x) in a per-translation-unit device buffer the host reads with bvc_debug_report() --
//                  and the access is skipped or made harmless, so the run goes on and leaves evidence instead of a dead queue.
// -DBVC_POISON     every kernel first fills its whole LDS allocation with 0xFF bytes, and every device scratch buffer the
//                  context allocates is filled with 0xFF instead of zeros: a read-before-write of LDS or scratch then reads
//                  the same loud pattern in every run instead of whatever the previous kernel or process left there.
// tools/poison_run.sh builds with both and runs the GPU parity suite and the bench legs once.
#if defined(BVC_CHECK_LDS) || defined(BVC_POISON)
// bytes of LDS of the running workgroup: static part + the dynamic part of the launch (code object v5 hidden argument)
__device__ __forceinline__ uint32_t lds_bytes_of_workgroup()
{
    return __builtin_amdgcn_groupstaticsize() + ((const uint32_t *)__builtin_amdgcn_implicitarg_ptr())[30];
}
#endif

#ifdef BVC_CHECK_LDS
// [0] violations so far, [1..5] the first one: check id, value, limit, blockIdx.x, threadIdx.x
static __device__ uint32_t g_lds_violation[8];
__device__ __forceinline__ bool lds_check_fail(uint32_t id, uint32_t value, uint32_t limit)
{
    if (atomicAdd(&g_lds_violation[0], 1u) == 0u) {
        g_lds_violation[1] = id; g_lds_violation[2] = value; g_lds_violation[3] = limit;
        g_lds_violation[4] = blockIdx.x; g_lds_violation[5] = threadIdx.x;
    }
    return false;
}
// true when value < limit; records the violation otherwise
#define BVC_LDS_OK(id, value, limit) ((uint32_t)(value) < (uint32_t)(limit) ? true : bvc::lds_check_fail((id), (uint32_t)(value), (uint32_t)(limit)))
#define BVC_DEFINE_DEBUG_READER(fn)                                                                              \
    hipError_t fn(uint32_t *out8, bool reset)                                                                     \
    {                                                                                                             \
        hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_lds_violation), 8 * sizeof(uint32_t));             \
        if (e == hipSuccess && reset) {                                                                           \
            const uint32_t zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};                                                    \
            e = hipMemcpyToSymbol(HIP_SYMBOL(g_lds_violation), zero, sizeof zero);                                \
        }                                                                                                         \
        return e;                                                                                                 \
    }
#else
#define BVC_LDS_OK(id, value, limit) (true)
#endif

#ifdef BVC_POISON
// First statement of every kernel: the whole allocation to 0xFF, then a barrier (the kernel's own initialisation follows).
__device__ __forceinline__ void poison_lds()
{
    typedef __attribute__((address_space(3))) uint32_t lds_word;
    const uint32_t words = lds_bytes_of_workgroup() >> 2;
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x)
        __hip_atomic_store((lds_word *)(uintptr_t)(i << 2), 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
}
#define BVC_POISON_LDS() bvc::poison_lds()
#else
#define BVC_POISON_LDS() do { } while (0)
#endif

// Occupancy targets are for the product: the diagnostic builds carry extra code and must not spill where the product does not.
#if defined(BVC_CHECK_LDS) || defined(BVC_POISON)
#define BVC_WAVES_PER_EU(lo, hi)
#else
#define BVC_WAVES_PER_EU(lo, hi) __attribute__((amdgpu_waves_per_eu(lo, hi)))
#endif

// ---- cross-lane movement -------------------------------------------------------------------------
// DPP controls (ISA: quad_perm 0x00-0xFF, row_mirror 0x140, row_half_mirror 0x141)
constexpr int kDppXor1 = 0xB1;     // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;     // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141;
constexpr int kDppMirror = 0x140;

// All controls used here (quad_perm, row mirrors) read a valid lane for every lane, so the "old" operand
// is dead: bound_ctrl = true lets the compiler leave it undefined instead of zero-initialising it.
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v)
{
    return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true);
}

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const int lo = dpp_i32<CTRL>(__double2loint(v));
    const int hi = dpp_i32<CTRL>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// Sum over the 16 lanes of a DPP row, result in every lane of the row.  Each step adds the partner's
// value to the lane's own, and IEEE addition commutes, so all 16 lanes end with the same bits.
__device__ __forceinline__ double row_sum(double v)
{
    v += dpp_f64<kDppXor1>(v);
    v += dpp_f64<kDppXor2>(v);
    v += dpp_f64<kDppHalfMirror>(v);
    v += dpp_f64<kDppMirror>(v);
    return v;
}

// Two independent row sums, step by step side by side: one chain's DPP wait states are filled by the other.
__device__ __forceinline__ void row_sum2(double &x, double &y)
{
    double tx = dpp_f64<kDppXor1>(x), ty = dpp_f64<kDppXor1>(y);
    x += tx; y += ty;
    tx = dpp_f64<kDppXor2>(x); ty = dpp_f64<kDppXor2>(y);
    x += tx; y += ty;
    tx = dpp_f64<kDppHalfMirror>(x); ty = dpp_f64<kDppHalfMirror>(y);
    x += tx; y += ty;
    tx = dpp_f64<kDppMirror>(x); ty = dpp_f64<kDppMirror>(y);
    x += tx; y += ty;
}

__device__ __forceinline__ int row_sum(int v)
{
    v += dpp_i32<kDppXor1>(v);
    v += dpp_i32<kDppXor2>(v);
    v += dpp_i32<kDppHalfMirror>(v);
    v += dpp_i32<kDppMirror>(v);
    return v;
}

// Value held by `lane` (compile-time), broadcast through SGPRs: wave-uniform by construction.
template <int LANE>
__device__ __forceinline__ double lane_value(double v)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), LANE);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), LANE);
    return __hiloint2double(hi, lo);
}

// Sum of the four row values (each already uniform within its row), fixed order -> uniform result.
__device__ __forceinline__ double rows_total(double v)
{
    return ((lane_value<0>(v) + lane_value<16>(v)) + lane_value<32>(v)) + lane_value<48>(v);
}

// ---- gfx950 cross-row exchange ---------------------------------------------------------------------
// v_permlane32_swap vdst, src0: lanes [63:32] of vdst <-> lanes [31:0] of src0.
// v_permlane16_swap vdst, src0: odd rows of vdst <-> even rows of src0.
// Passing (x, y) and ADDING the two results gives, with no select:
//   swap32: lanes 0-31: x[l] + y[l+32]      lanes 32-63: y[l] + x[l-32]
//   swap16: even rows : x[l] + y[l+16]      odd rows   : y[l] + x[l-16]
// so (x, x) is a pair sum and (x, y) sums x over one half/row-pair and y over the other.
struct DPair { double a, b; };

__device__ __forceinline__ DPair swap32(double x, double y)
{
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(y), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(y), false, false);
    return DPair{__hiloint2double((int)hi[0], (int)lo[0]), __hiloint2double((int)hi[1], (int)lo[1])};
}

__device__ __forceinline__ DPair swap16(double x, double y)
{
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(x), __double2loint(y), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(x), __double2hiint(y), false, false);
    return DPair{__hiloint2double((int)hi[0], (int)lo[0]), __hiloint2double((int)hi[1], (int)lo[1])};
}

// ---- FP64 helpers for the EM inner loop ------------------------------------------------------------
// 1/x for x in the normal range: v_rcp_f64 (about single precision) + two Newton steps.  No scaling or
// fix-up: x is a mixture probability in (0, 1].  x = 0 gives NaN (inf * 0 in the first step), which is
// what the reference's 0/0 gives (src/Algorithm.cpp:81-82).
__device__ __forceinline__ double fast_rcp(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    double t = fma(-x, y, 1.0);
    y = fma(y, t, y);
    t = fma(-x, y, 1.0);
    y = fma(y, t, y);
    return y;
}

// Above this relative change of a class marginal the reciprocal is taken from v_rcp_f64 instead of a Newton
// update of the previous one (em_kernel.hip).
constexpr double kLog1pMaxU = 0.015625;   // 2^-6

// Natural log of a positive finite double (about 2 ulp): frexp, then 2*atanh((m-1)/(m+1)) on
// m in [sqrt(1/2), sqrt(2)).  0 -> -inf, NaN -> NaN.
__device__ __forceinline__ double log_pos(double x)
{
    double m = __builtin_amdgcn_frexp_mant(x);
    int ex = __builtin_amdgcn_frexp_exp(x);
    const bool low = m < 0.70710678118654752440;
    m = low ? m + m : m;
    ex = low ? ex - 1 : ex;
    const double s = (m - 1.0) * fast_rcp(m + 1.0);
    const double s2 = s * s;
    double p = 1.0 / 21.0;
    p = fma(p, s2, 1.0 / 19.0);
    p = fma(p, s2, 1.0 / 17.0);
    p = fma(p, s2, 1.0 / 15.0);
    p = fma(p, s2, 1.0 / 13.0);
    p = fma(p, s2, 1.0 / 11.0);
    p = fma(p, s2, 1.0 / 9.0);
    p = fma(p, s2, 1.0 / 7.0);
    p = fma(p, s2, 1.0 / 5.0);
    p = fma(p, s2, 1.0 / 3.0);
    const double lm = 2.0 * fma(s * s2, p, s);
    const double e = (double)ex;
    const double r = fma(e, 6.93147180369123816490e-01, fma(e, 1.90821492927058770002e-10, lm));
    return x == 0.0 ? -__builtin_huge_val() : r;
}

// ---- chi-square survival function, df = 1 --------------------------------------------------------
// Attribution: kf_lgamma_dev / kf_gammaq_dev below restate, nearly statement for statement,
// the numerical routines of htslib's kfunc.c (https://github.com/samtools/htslib, MIT/Expat licence, (c) Genome Research Ltd. and
// Attractive Chaos), which the reference links through SeqLib (src/Algorithm.cpp:3-25; .gitmodules:1-3, submodule absent from the
// reference tree).  The arithmetic has to be htslib's for the outputs to match the reference's; the constants are htslib's.
// The reference's chisf(x, 1) = kf_gammaq(0.5, x/2) (src/Algorithm.cpp:3-7; htslib kfunc.c, absent from
// the reference tree).  Same published algorithm as the reference links against: Lanczos-type
// log-gamma, power series for P when z <= 1 or z < s, modified-Lentz continued fraction for Q otherwise;
// stop at 1e-14 or 100 terms.
__device__ inline double kf_lgamma_dev(double z)
{
    double x = 0.0;
    x += 0.1659470187408462e-06 / (z + 7);
    x += 0.9934937113930748e-05 / (z + 6);
    x -= 0.1385710331296526 / (z + 5);
    x += 12.50734324009056 / (z + 4);
    x -= 176.6150291498386 / (z + 3);
    x += 771.3234287757674 / (z + 2);
    x -= 1259.139216722289 / (z + 1);
    x += 676.5203681218835 / z;
    x += 0.9999999999995183;
    return log(x) - 5.58106146679532777 - z + (z - 0.5) * log(z + 6.5);
}

__device__ inline double kf_gammaq_dev(double s, double z)
{
    constexpr double kEps = 1e-14, kTiny = 1e-290;
    if (z <= 1.0 || z < s) {
        double term = 1.0, sum = 1.0;
        for (int k = 1; k < 100; ++k) {
            term *= z / (s + k);
            sum += term;
            if (term / sum < kEps) break;
        }
        return 1.0 - exp(s * log(z) - z - kf_lgamma_dev(s + 1.0) + log(sum));
    }
    double f = 1.0 + z - s, C = f, D = 0.0;
    for (int j = 1; j < 100; ++j) {
        const double a = j * (s - j), b = (j << 1) + 1 + z - s;
        D = b + a * D;
        if (D < kTiny) D = kTiny;
        C = b + a / C;
        if (C < kTiny) C = kTiny;
        D = 1.0 / D;
        const double d = C * D;
        f *= d;
        if (fabs(d - 1.0) < kEps) break;
    }
    return exp(s * log(z) - z - kf_lgamma_dev(s) - log(f));
}

}  // namespace bvc
