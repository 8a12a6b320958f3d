// hist_kernel.hip -- stage 1 of the basetype path on gfx950: one streaming pass over a site's base and
// base-quality bytes, counted into a (base, qual) histogram in LDS.
//
// This replaces the per-sample likelihood table the reference materialises in the BaseType constructor
// (/root/reference/src/BaseType.cpp:5-23: 4 doubles per sample) and re-reads ~700 times per site in
// singleEM (/root/reference/src/Algorithm.cpp:69-93): a sample's likelihood row depends only on its
// (base, qual) pair, so the 512 class counts carry everything the EM and the LRT need.
//
// HBM-bound by design: 2 bytes per (site, sample), each read exactly once with 16-byte loads per lane.
// The LDS histogram is replicated kCopies times with copy = lane % kCopies, laid out [class][copy], so the
// bank of an update is lane % 32 whatever the data are: the heavily skewed keys of real pileups (>90 %
// reference base, a handful of quality values) cannot cause bank or same-address conflicts.
#include <atomic>

#include "bvc_device.h"
#include "bvc_internal.h"

namespace bvc {
namespace {

#ifndef BVC_HIST_THREADS
#define BVC_HIST_THREADS 512
#endif
#ifndef BVC_HIST_UNROLL
#define BVC_HIST_UNROLL 2
#endif
constexpr int kHistThreads = BVC_HIST_THREADS;
constexpr int kCopies = 32;                         // one copy per LDS bank
constexpr int kLdsWords = BVC_NCLASS * kCopies;     // 16384 words = 64 KiB
constexpr int kUnroll = BVC_HIST_UNROLL;            // 16-byte loads in flight per lane and array

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void count_word(uint32_t *__restrict__ hist, uint32_t bw, uint32_t qw, uint32_t lane_off)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t b = (bw >> (8 * i)) & 0xFFu;
        const uint32_t q = (qw >> (8 * i)) & 0xFFu;
        const uint32_t key = (b << 7) | q;                       // class = base * 128 + qual
        __hip_atomic_fetch_add(&hist[key * kCopies + lane_off], 1u, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

__device__ __forceinline__ void count_word_checked(uint32_t *__restrict__ hist, uint32_t bw, uint32_t qw,
                                                   uint32_t lane_off)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t b = (bw >> (8 * i)) & 0xFFu;
        const uint32_t q = (qw >> (8 * i)) & 0xFFu;
        if (b < 4u && q < 128u)                                   // covered sample
            __hip_atomic_fetch_add(&hist[((b << 7) | q) * kCopies + lane_off], 1u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// 16 samples of one lane.  The common case (every sample of the wave covered) skips the per-sample test.
__device__ __forceinline__ void count_chunk(uint32_t *__restrict__ hist, const u32x4 b, const u32x4 q,
                                            uint32_t lane_off)
{
    const uint32_t bad = ((b.x | b.y | b.z | b.w) & 0xFCFCFCFCu) | ((q.x | q.y | q.z | q.w) & 0x80808080u);
    if (__ballot(bad != 0) == 0) {
        count_word(hist, b.x, q.x, lane_off); count_word(hist, b.y, q.y, lane_off);
        count_word(hist, b.z, q.z, lane_off); count_word(hist, b.w, q.w, lane_off);
    } else {
        count_word_checked(hist, b.x, q.x, lane_off); count_word_checked(hist, b.y, q.y, lane_off);
        count_word_checked(hist, b.z, q.z, lane_off); count_word_checked(hist, b.w, q.w, lane_off);
    }
}

// One workgroup per (site, split).  ALIGNED: row starts and n16 chunks are 16-byte aligned.
template <bool ALIGNED>
__global__ __launch_bounds__(kHistThreads) void hist_dense_kernel(
    int64_t n_sites, int64_t n_samples, int64_t row_stride, const int8_t *__restrict__ bases,
    const int8_t *__restrict__ quals, uint32_t *__restrict__ counts, int split)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];   // [class][copy]
    const int tid = threadIdx.x;
    const uint32_t lane_off = (uint32_t)(tid & (kCopies - 1));
    // This kernel is HBM-bound and may share the chip with the FP64-bound EM kernel of the previous tile
    // (overlap mode): its few instructions go first so the memory pipeline never waits on the VALU.
    __builtin_amdgcn_s_setprio(3);

    for (int i = tid * 4; i < kLdsWords; i += kHistThreads * 4)
        *reinterpret_cast<u32x4 *>(&hist[i]) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();

    const int64_t n_work = n_sites * split;
    for (int64_t w = blockIdx.x; w < n_work; w += gridDim.x) {
        const int64_t site = w / split;
        const int part = (int)(w % split);
        const int8_t *brow = bases + site * row_stride;
        const int8_t *qrow = quals + site * row_stride;
        // The parts of a site take its 16-sample chunks round robin in blocks of kUnroll * kHistThreads chunks
        // (16 KiB per array), so the workgroups that share a row sweep it together: on MI355X a few wide
        // sweeping windows read HBM faster than one narrow stream per workgroup (tools/micro/read_bw.hip).
        // The last part also takes the ragged tail.
        const int64_t n16 = n_samples >> 4;
        constexpr int64_t kBlockChunks = (int64_t)kUnroll * kHistThreads;

        if (ALIGNED) {
            const u32x4 *bv = reinterpret_cast<const u32x4 *>(brow);
            const u32x4 *qv = reinterpret_cast<const u32x4 *>(qrow);
            for (int64_t cb = (int64_t)part * kBlockChunks; cb < n16; cb += (int64_t)split * kBlockChunks) {
                const int64_t c = cb + tid;
                if (cb + kBlockChunks <= n16) {
                    // kUnroll independent 16-byte loads per array in flight, then the LDS updates
                    u32x4 b[kUnroll], q[kUnroll];
#pragma unroll
                    for (int u = 0; u < kUnroll; ++u) {
                        b[u] = __builtin_nontemporal_load(&bv[c + (int64_t)u * kHistThreads]);
                        q[u] = __builtin_nontemporal_load(&qv[c + (int64_t)u * kHistThreads]);
                    }
#pragma unroll
                    for (int u = 0; u < kUnroll; ++u) count_chunk(hist, b[u], q[u], lane_off);
                } else {
                    for (int64_t ct = c; ct < n16; ct += kHistThreads) {
                        const u32x4 b = __builtin_nontemporal_load(&bv[ct]);
                        const u32x4 q = __builtin_nontemporal_load(&qv[ct]);
                        // not every lane of the wave is here: per-sample test, no wave-wide vote
                        count_word_checked(hist, b.x, q.x, lane_off); count_word_checked(hist, b.y, q.y, lane_off);
                        count_word_checked(hist, b.z, q.z, lane_off); count_word_checked(hist, b.w, q.w, lane_off);
                    }
                }
            }
        } else {
            for (int64_t cb = (int64_t)part * kBlockChunks; cb < n16; cb += (int64_t)split * kBlockChunks) {
                const int64_t i1 = (cb + kBlockChunks < n16 ? cb + kBlockChunks : n16) * 16;
                for (int64_t i = cb * 16 + tid; i < i1; i += kHistThreads) {
                    const uint32_t b = (uint8_t)brow[i], q = (uint8_t)qrow[i];
                    if (b < 4u && q < 128u)
                        __hip_atomic_fetch_add(&hist[((b << 7) | q) * kCopies + lane_off], 1u, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        if (part == split - 1) {
            for (int64_t i = (n16 << 4) + tid; i < n_samples; i += kHistThreads) {
                const uint32_t b = (uint8_t)brow[i], q = (uint8_t)qrow[i];
                if (b < 4u && q < 128u)
                    __hip_atomic_fetch_add(&hist[((b << 7) | q) * kCopies + lane_off], 1u, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        __syncthreads();

        // fold the copies of each class, publish, and clear for the next site
        for (int key = tid; key < BVC_NCLASS; key += kHistThreads) {
            uint32_t s = 0;
#pragma unroll
            for (int v = 0; v < kCopies; v += 4) {
                // rotate the starting copy by the key so that the 16 lanes of a ds_read_b128 group spread over banks
                const int cc = (v + 4 * (key & 7)) & (kCopies - 1);
                u32x4 *p = reinterpret_cast<u32x4 *>(&hist[key * kCopies + cc]);
                const u32x4 x = *p;
                s += x.x + x.y + x.z + x.w;
                *p = u32x4{0u, 0u, 0u, 0u};
            }
            if (split == 1) counts[site * BVC_NCLASS + key] = s;
            else if (s) atomicAdd(&counts[site * BVC_NCLASS + key], s);
        }
        __syncthreads();
    }
}

// Group mode (--group, /root/reference/src/BaseVarC.cpp:617-661): one pass builds n_groups + 1 histograms per
// site (the last one collects samples that belong to no group, :352-356).  LDS holds [hist][class][copy]
// with as many copies as fit 64 KiB (copies = 1 << log2c); fewer copies than banks means some conflicts,
// which the spread of keys over groups softens.
// LOG2C >= 0: the number of copies is a compile-time constant (the address is then five VALU instructions per
// sample); LOG2C < 0: taken from the argument.
template <bool ALIGNED, int LOG2C>
__global__ __launch_bounds__(kHistThreads) void hist_dense_groups_kernel(
    int64_t n_sites, int64_t n_samples, int64_t row_stride, const int8_t *__restrict__ bases,
    const int8_t *__restrict__ quals, const uint8_t *__restrict__ group_of_sample, int n_groups, int log2c_arg,
    uint32_t *__restrict__ grp_counts, const int64_t *__restrict__ bounds)
{
    const int log2c = LOG2C >= 0 ? LOG2C : log2c_arg;
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
    if (bounds && bounds[0] == 0) return;        // samples are ordered by group: hist_dense_ranges_kernel has the call
    __builtin_amdgcn_s_setprio(3);               // as in hist_dense_kernel: ahead of the EM kernels it shares the chip with
    const int tid = threadIdx.x;
    const int n_hist = n_groups + 1;
    const int words = (n_hist * BVC_NCLASS) << log2c;
    const uint32_t lane_off = (uint32_t)tid & ((1u << log2c) - 1u);
    for (int i = tid; i < words; i += kHistThreads) hist[i] = 0;
    __syncthreads();

    auto add_valid = [&](uint32_t b, uint32_t q, uint32_t g) {
        const uint32_t h = g < (uint32_t)n_groups ? g : (uint32_t)n_groups;
        __hip_atomic_fetch_add(&hist[((h << (9 + log2c)) | (b << (7 + log2c)) | (q << log2c)) + lane_off], 1u,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto add = [&](uint32_t b, uint32_t q, uint32_t g) {
        if (b < 4u && q < 128u) add_valid(b, q, g);
    };

    for (int64_t site = blockIdx.x; site < n_sites; site += gridDim.x) {
        const int8_t *brow = bases + site * row_stride;
        const int8_t *qrow = quals + site * row_stride;
        const int64_t n16 = ALIGNED ? (n_samples >> 4) : 0;
        if (ALIGNED) {
            const u32x4 *bv = reinterpret_cast<const u32x4 *>(brow);
            const u32x4 *qv = reinterpret_cast<const u32x4 *>(qrow);
            const u32x4 *gv = reinterpret_cast<const u32x4 *>(group_of_sample);
            // the common case (every sample of the wave covered) skips the per-sample test, as in count_chunk
            auto count16 = [&](const u32x4 b, const u32x4 q, const u32x4 g) {
                const uint32_t bw[4] = {b.x, b.y, b.z, b.w}, qw[4] = {q.x, q.y, q.z, q.w}, gw[4] = {g.x, g.y, g.z, g.w};
                const uint32_t bad = ((b.x | b.y | b.z | b.w) & 0xFCFCFCFCu) | ((q.x | q.y | q.z | q.w) & 0x80808080u);
                if (__ballot(bad != 0) == 0) {
#pragma unroll
                    for (int w = 0; w < 4; ++w)
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            add_valid((bw[w] >> (8 * k)) & 0xFFu, (qw[w] >> (8 * k)) & 0xFFu, (gw[w] >> (8 * k)) & 0xFFu);
                } else {
#pragma unroll
                    for (int w = 0; w < 4; ++w)
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            add((bw[w] >> (8 * k)) & 0xFFu, (qw[w] >> (8 * k)) & 0xFFu, (gw[w] >> (8 * k)) & 0xFFu);
                }
            };
            int64_t c = tid;
            for (; c + kHistThreads < n16; c += 2 * kHistThreads) {        // two 16-byte loads per array in flight
                const u32x4 b0 = __builtin_nontemporal_load(&bv[c]), b1 = __builtin_nontemporal_load(&bv[c + kHistThreads]);
                const u32x4 q0 = __builtin_nontemporal_load(&qv[c]), q1 = __builtin_nontemporal_load(&qv[c + kHistThreads]);
                const u32x4 g0 = gv[c], g1 = gv[c + kHistThreads];
                count16(b0, q0, g0);
                count16(b1, q1, g1);
            }
            for (; c < n16; c += kHistThreads)
                count16(__builtin_nontemporal_load(&bv[c]), __builtin_nontemporal_load(&qv[c]), gv[c]);
        }
        for (int64_t i = (n16 << 4) + tid; i < n_samples; i += kHistThreads)
            add((uint8_t)brow[i], (uint8_t)qrow[i], group_of_sample[i]);
        __syncthreads();
        for (int key = tid; key < n_hist * BVC_NCLASS; key += kHistThreads) {
            uint32_t s = 0;
            for (int v = 0; v < (1 << log2c); ++v) {
                s += hist[(key << log2c) + v];
                hist[(key << log2c) + v] = 0;
            }
            grp_counts[site * n_hist * BVC_NCLASS + key] = s;
        }
        __syncthreads();
    }
}

// Group mode, samples ordered by group (every group a contiguous run of columns, ungrouped samples last): the
// histogram of (site, group) is then the plain histogram of a column range, so the pass keeps the dense kernel's
// 32 conflict-free LDS copies and its wave-wide fast path, and needs no per-sample group byte.
// scratch[0] = 0 when group_of_sample is non-decreasing after clamping to n_groups, scratch[1 + h] = first sample of
// histogram h, scratch[1 + n_hist] = n_samples (scratch is zeroed before the launch).
__global__ void group_bounds_kernel(const uint8_t *__restrict__ group_of_sample, int64_t n_samples, int n_groups,
                                    int64_t *__restrict__ scratch)
{
    const int n_hist = n_groups + 1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_samples; i += (int64_t)gridDim.x * blockDim.x) {
        const int h = min((int)group_of_sample[i], n_groups);
        if (i == 0)
            for (int t = 0; t <= h; ++t) scratch[1 + t] = 0;
        if (i == n_samples - 1) {
            for (int t = h + 1; t <= n_hist; ++t) scratch[1 + t] = n_samples;
        } else {
            const int hn = min((int)group_of_sample[i + 1], n_groups);
            if (hn < h) scratch[0] = 1;                          // not ordered: the general kernel takes the call
            for (int t = h + 1; t <= hn; ++t) scratch[1 + t] = i + 1;
        }
    }
}

template <bool ALIGNED>
__global__ __launch_bounds__(kHistThreads) void hist_dense_ranges_kernel(
    int64_t n_sites, int64_t n_samples, int64_t row_stride, const int8_t *__restrict__ bases,
    const int8_t *__restrict__ quals, int n_hist, const int64_t *__restrict__ scratch, uint32_t *__restrict__ grp_counts)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];   // [class][copy]
    (void)n_samples;
    if (scratch[0] != 0) return;
    const int tid = threadIdx.x;
    const uint32_t lane_off = (uint32_t)(tid & (kCopies - 1));
    __builtin_amdgcn_s_setprio(3);
    for (int i = tid * 4; i < kLdsWords; i += kHistThreads * 4)
        *reinterpret_cast<u32x4 *>(&hist[i]) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();

    auto scalar = [&](const int8_t *brow, const int8_t *qrow, int64_t i0, int64_t i1) {
        for (int64_t i = i0 + tid; i < i1; i += kHistThreads) {
            const uint32_t b = (uint8_t)brow[i], q = (uint8_t)qrow[i];
            if (b < 4u && q < 128u)
                __hip_atomic_fetch_add(&hist[((b << 7) | q) * kCopies + lane_off], 1u, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    const int64_t n_work = n_sites * n_hist;
    for (int64_t w = blockIdx.x; w < n_work; w += gridDim.x) {
        const int64_t site = w / n_hist;
        const int h = (int)(w % n_hist);
        const int8_t *brow = bases + site * row_stride;
        const int8_t *qrow = quals + site * row_stride;
        const int64_t s0 = scratch[1 + h], s1 = scratch[2 + h];
        // [s0, s1) = unaligned head, whole 16-sample chunks [c0, c1), unaligned tail
        const int64_t c0 = (s0 + 15) >> 4, c1 = s1 >> 4;
        if (ALIGNED && c0 < c1) {
            scalar(brow, qrow, s0, c0 << 4);
            const u32x4 *bv = reinterpret_cast<const u32x4 *>(brow);
            const u32x4 *qv = reinterpret_cast<const u32x4 *>(qrow);
            constexpr int64_t kBlockChunks = (int64_t)kUnroll * kHistThreads;
            int64_t cb = c0;
            for (; cb + kBlockChunks <= c1; cb += kBlockChunks) {
                u32x4 b[kUnroll], q[kUnroll];
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    b[u] = __builtin_nontemporal_load(&bv[cb + tid + (int64_t)u * kHistThreads]);
                    q[u] = __builtin_nontemporal_load(&qv[cb + tid + (int64_t)u * kHistThreads]);
                }
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) count_chunk(hist, b[u], q[u], lane_off);
            }
            for (int64_t ct = cb + tid; ct < c1; ct += kHistThreads) {
                const u32x4 b = __builtin_nontemporal_load(&bv[ct]);
                const u32x4 q = __builtin_nontemporal_load(&qv[ct]);
                count_word_checked(hist, b.x, q.x, lane_off); count_word_checked(hist, b.y, q.y, lane_off);
                count_word_checked(hist, b.z, q.z, lane_off); count_word_checked(hist, b.w, q.w, lane_off);
            }
            scalar(brow, qrow, c1 << 4, s1);
        } else {
            scalar(brow, qrow, s0, s1);
        }
        __syncthreads();
        for (int key = tid; key < BVC_NCLASS; key += kHistThreads) {
            uint32_t sum = 0;
#pragma unroll
            for (int v = 0; v < kCopies; v += 4) {
                const int cc = (v + 4 * (key & 7)) & (kCopies - 1);
                u32x4 *p = reinterpret_cast<u32x4 *>(&hist[key * kCopies + cc]);
                const u32x4 x = *p;
                sum += x.x + x.y + x.z + x.w;
                *p = u32x4{0u, 0u, 0u, 0u};
            }
            grp_counts[w * BVC_NCLASS + key] = sum;
        }
        __syncthreads();
    }
}

// Ragged pileup: site s owns elements offsets[s] .. offsets[s+1]).  Byte loads (rows start anywhere).

__global__ __launch_bounds__(kHistThreads) void hist_csr_kernel(
    int64_t n_sites, const int64_t *__restrict__ offsets, const int8_t *__restrict__ bases,
    const int8_t *__restrict__ quals, uint32_t *__restrict__ counts)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
    const int tid = threadIdx.x;
    const uint32_t lane_off = (uint32_t)(tid & (kCopies - 1));
    for (int i = tid * 4; i < kLdsWords; i += kHistThreads * 4)
        *reinterpret_cast<u32x4 *>(&hist[i]) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    for (int64_t site = blockIdx.x; site < n_sites; site += gridDim.x) {
        const int64_t o0 = offsets[site], o1 = offsets[site + 1];
        for (int64_t i = o0 + tid; i < o1; i += kHistThreads) {
            const uint32_t b = (uint8_t)bases[i], q = (uint8_t)quals[i];
            if (b < 4u && q < 128u)
                __hip_atomic_fetch_add(&hist[((b << 7) | q) * kCopies + lane_off], 1u, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();
        for (int key = tid; key < BVC_NCLASS; key += kHistThreads) {
            uint32_t s = 0;
            for (int v = 0; v < kCopies; ++v) {
                s += hist[key * kCopies + v];
                hist[key * kCopies + v] = 0;
            }
            counts[site * BVC_NCLASS + key] = s;
        }
        __syncthreads();
    }
}

// Plain streaming read, 16 B per lane, nothing else: the empirical HBM read ceiling the histogram kernel is
// compared with next to the 8 TB/s spec figure.  The XOR keeps the loads alive.
__global__ __launch_bounds__(512) void stream_read_kernel(const u32x4 *__restrict__ src, int64_t n16, uint32_t *__restrict__ sink)
{
    uint32_t acc = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + stride < n16; i += 2 * stride) {
        const u32x4 a = __builtin_nontemporal_load(&src[i]), b = __builtin_nontemporal_load(&src[i + stride]);
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
    }
    for (; i < n16; i += stride) { const u32x4 a = __builtin_nontemporal_load(&src[i]); acc ^= a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x9E3779B9u) sink[0] = acc;       // practically never: the store only makes the result observable
}

}  // namespace

hipError_t launch_stream_read(hipStream_t stream, const void *src, int64_t bytes, uint32_t *sink)
{
    const int64_t n16 = bytes / 16;
    if (n16 <= 0) return hipSuccess;
    hipLaunchKernelGGL(stream_read_kernel, dim3(4096), dim3(512), 0, stream, reinterpret_cast<const u32x4 *>(src), n16, sink);
    return hipGetLastError();
}

int choose_hist_split(int64_t n_sites, int64_t n_samples, int n_cu)
{
    static const int forced = [] { const char *e = getenv("BVC_HIST_SPLIT"); const int v = e ? atoi(e) : 0; return (v >= 1 && v <= 64) ? v : 0; }();
    if (forced) return forced;
    // Two 512-thread workgroups fit a CU (2 x 64 KiB LDS).  With fewer sites than that, cut each site's
    // sample range so the whole chip streams; parts are merged with global atomics on 512 words.
    const int64_t want = (int64_t)n_cu * 4;
    if (n_sites >= want || n_samples < (1 << 16)) return 1;
    int64_t split = (want + n_sites - 1) / n_sites;
    const int64_t max_split = n_samples / (1 << 15);             // keep >= 32k samples per part
    if (split > max_split) split = max_split;
    if (split < 1) split = 1;
    if (split > 64) split = 64;
    return (int)split;
}

hipError_t launch_hist_dense(hipStream_t stream, int64_t n_sites, int64_t n_samples, int64_t row_stride,
                             const int8_t *bases, const int8_t *quals, const uint8_t *group_of_sample,
                             int n_groups, uint32_t *counts, int split, int64_t *group_scratch)
{
    if (n_sites <= 0) return hipSuccess;
    static std::atomic<bool> attr_done_dev[kMaxDevices][2], gattr_done_dev[kMaxDevices][7], rattr_done_dev[kMaxDevices][2];
    std::atomic<bool> *attr_done = attr_done_dev[current_device_slot()];
    std::atomic<bool> *gattr_done = gattr_done_dev[current_device_slot()];
    std::atomic<bool> *rattr_done = rattr_done_dev[current_device_slot()];
    const bool aligned = ((reinterpret_cast<uintptr_t>(bases) | reinterpret_cast<uintptr_t>(quals)) & 15u) == 0 &&
                         (row_stride & 15) == 0;
    const size_t lds = (size_t)kLdsWords * sizeof(uint32_t);
    if (group_of_sample) {                       // counts = [site][n_groups + 1][512]
        const bool galigned = aligned && (reinterpret_cast<uintptr_t>(group_of_sample) & 15u) == 0;
        int log2c = 0;
        while (log2c < 5 && (size_t)(n_groups + 1) * BVC_NCLASS * (2u << log2c) <= (size_t)kLdsWords) ++log2c;
        using GroupKernel = void (*)(int64_t, int64_t, int64_t, const int8_t *, const int8_t *, const uint8_t *, int, int,
                                     uint32_t *, const int64_t *);
        static const GroupKernel aligned_kernels[6] = {
            hist_dense_groups_kernel<true, 0>, hist_dense_groups_kernel<true, 1>, hist_dense_groups_kernel<true, 2>,
            hist_dense_groups_kernel<true, 3>, hist_dense_groups_kernel<true, 4>, hist_dense_groups_kernel<true, 5>};
        GroupKernel gk = galigned ? aligned_kernels[log2c] : hist_dense_groups_kernel<false, -1>;
        const int gslot = galigned ? 1 + log2c : 0;
        if (!gattr_done[gslot]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gk),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (int)lds);
            if (e != hipSuccess) return e;
            gattr_done[gslot] = true;
        }
        const size_t glds = ((size_t)(n_groups + 1) * BVC_NCLASS << log2c) * sizeof(uint32_t);
        const int64_t ggrid = n_sites < 4096 ? n_sites : 4096;
        const bool try_ranges = group_scratch != nullptr && n_samples > 0;
        if (try_ranges) {
            // Decided on the device, without a host round trip: the bounds kernel marks whether the samples are
            // ordered by group; the range kernel and the general kernel are both launched and the one whose turn
            // it is not returns at once.
            const int n_hist = n_groups + 1;
            auto rk = aligned ? hist_dense_ranges_kernel<true> : hist_dense_ranges_kernel<false>;
            if (!rattr_done[aligned]) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(rk),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return e;
                rattr_done[aligned] = true;
            }
            hipError_t e = hipMemsetAsync(group_scratch, 0, (size_t)(BVC_MAX_GROUPS + 4) * sizeof(int64_t), stream);
            if (e != hipSuccess) return e;
            const int64_t bgrid = (n_samples + 255) / 256;
            hipLaunchKernelGGL(group_bounds_kernel, dim3((unsigned)(bgrid < 1024 ? bgrid : 1024)), dim3(256), 0, stream,
                               group_of_sample, n_samples, n_groups, group_scratch);
            const int64_t n_work = n_sites * n_hist;
            hipLaunchKernelGGL(rk, dim3((unsigned)(n_work < 4096 ? n_work : 4096)), dim3(kHistThreads), lds, stream,
                               n_sites, n_samples, row_stride, bases, quals, n_hist, group_scratch, counts);
        }
        hipLaunchKernelGGL(gk, dim3((unsigned)ggrid), dim3(kHistThreads), glds, stream, n_sites, n_samples,
                           row_stride, bases, quals, group_of_sample, n_groups, log2c, counts,
                           try_ranges ? group_scratch : nullptr);
        return hipGetLastError();
    }
    auto kern = aligned ? hist_dense_kernel<true> : hist_dense_kernel<false>;
    if (!attr_done[aligned]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done[aligned] = true;
    }
    const int64_t n_work = n_sites * split;
    const int64_t grid = n_work < 4096 ? n_work : 4096;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kHistThreads), lds, stream, n_sites, n_samples,
                       row_stride, bases, quals, counts, split);
    return hipGetLastError();
}

hipError_t launch_hist_csr(hipStream_t stream, int64_t n_sites, const int64_t *offsets,
                           const int8_t *bases, const int8_t *quals, uint32_t *counts)
{
    if (n_sites <= 0) return hipSuccess;
    static std::atomic<bool> attr_done_dev[kMaxDevices];
    std::atomic<bool> &attr_done = attr_done_dev[current_device_slot()];
    const size_t lds = (size_t)kLdsWords * sizeof(uint32_t);
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(hist_csr_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int64_t grid = n_sites < 4096 ? n_sites : 4096;
    hipLaunchKernelGGL(hist_csr_kernel, dim3((unsigned)grid), dim3(kHistThreads), lds, stream, n_sites, offsets,
                       bases, quals, counts);
    return hipGetLastError();
}

}  // namespace bvc
