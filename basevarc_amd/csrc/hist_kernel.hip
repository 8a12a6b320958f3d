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
constexpr int kGroupDepth = 2;                      // chunks prefetched per trip of the group kernel's sample loop (swept 1..4)
constexpr int64_t kWaveRowMax = 16384;              // dense rows up to this length go one wavefront per site

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// (byte BYTE of `word`) << shift in ONE VALU instruction (SDWA source-byte select); the compiler's own selection for
// the same expression is a shift plus a mask-and-or.  The histogram kernels are HBM-bound on their own but share
// the chip with the FP64-VALU-bound EM kernel in overlap mode, so every VALU instruction they do not issue is the
// EM kernel's.
template <int BYTE>
__device__ __forceinline__ uint32_t shl_byte(uint32_t word, uint32_t shift)
{
    uint32_t r;
    static_assert(BYTE >= 0 && BYTE < 4, "byte index");
    if (BYTE == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(shift), "v"(word));
    if (BYTE == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(shift), "v"(word));
    if (BYTE == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(shift), "v"(word));
    if (BYTE == 3) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(shift), "v"(word));
    return r;
}

// LDS addresses as plain 32-bit integers: the fast paths below assemble the byte address of a counter with bit
// operations and add one to it; going through a generic pointer would cost an extra VALU add of the array's base per
// sample.  lds_address(hist) + offsets are formed once per lane.
typedef __attribute__((address_space(3))) uint32_t lds_u32;

__device__ __forceinline__ uint32_t lds_address(uint32_t *p) { return (uint32_t)(uintptr_t)(lds_u32 *)p; }

__device__ __forceinline__ void lds_add_one(uint32_t lds_byte_address)
{
#ifdef BVC_CHECK_LDS
    // the address was assembled from input bytes: inside this workgroup's allocation?
    if (!BVC_LDS_OK(1, lds_byte_address, lds_bytes_of_workgroup() - 3u)) return;
#endif
    __hip_atomic_fetch_add((lds_u32 *)(uintptr_t)lds_byte_address, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// hist[idx] += 1 where idx was computed from a sample's bytes behind an explicit "covered?" test
__device__ __forceinline__ void hist_add(uint32_t *hist, uint32_t idx)
{
#ifdef BVC_CHECK_LDS
    if (!BVC_LDS_OK(2, lds_address(hist) + (idx << 2), lds_bytes_of_workgroup() - 3u)) return;
#endif
    __hip_atomic_fetch_add(&hist[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ void lds_store(uint32_t lds_byte_address, uint32_t v)
{
    __hip_atomic_store((lds_u32 *)(uintptr_t)lds_byte_address, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ uint32_t lds_load(uint32_t lds_byte_address)
{
    return __hip_atomic_load((lds_u32 *)(uintptr_t)lds_byte_address, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Workgroup barrier that orders LDS traffic only: the wave's LDS operations have completed (lgkmcnt) but its global
// loads may still be in flight, which __syncthreads() -- a fence over all memory -- would wait for.  Used where loads
// for the next work item are issued ahead of the fold of the current one.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// (Tried: a scheduling fence after each batch of independent loads, so that all 2 * kUnroll loads issue back to back --
// the compiler otherwise hoists the first use of the first load above the last two.  The compiler's order measures 1 %
// FASTER on the headline, 0.5 % slower on the column-range kernel: profiles/r02_hist_load_scheduling.txt.)

// The any-order group kernel's sample loop: each trip issues the loads of the next D chunks (of three 16-byte
// streams) before it counts the current D, so that a wave has bytes in flight while it works through its LDS atomics.
// Chunk c belongs to thread c mod kHistThreads; an out-of-range chunk reads nothing and carries base bytes 0xFF, which
// the counting skips.  (Depth 1 / 2 / 3 / 4 measure 1.34 / 1.29 / 1.31 / 1.47 ms alone:
// profiles/r02_slot_kernel_experiment.txt.)
template <int D, class Load, class Count>
__device__ __forceinline__ void stream_chunks(int64_t n16, int tid, Load load, Count count)
{
    u32x4 b[D], q[D], g[D], nb[D], nq[D], ng[D];
    // c0 = chunk of lane 0 (workgroup-uniform).  A trip whose D chunks all lie inside the row -- every trip but the last
    // one or two of a row -- loads without a test and without first filling the registers with the "skip me" pattern
    // (twelve moves per chunk that the common case used to pay for the rare one).
    auto fetch = [&](int64_t c0, u32x4 (&xb)[D], u32x4 (&xq)[D], u32x4 (&xg)[D]) {
        if (c0 + (int64_t)D * kHistThreads <= n16) {
#pragma unroll
            for (int d = 0; d < D; ++d) load(c0 + tid + (int64_t)d * kHistThreads, xb[d], xq[d], xg[d]);
        } else {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int64_t c = c0 + tid + (int64_t)d * kHistThreads;
                xb[d] = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
                xq[d] = u32x4{0u, 0u, 0u, 0u}; xg[d] = xq[d];
                if (c < n16) load(c, xb[d], xq[d], xg[d]);
            }
        }
    };
    int64_t c0 = 0;
    fetch(c0, b, q, g);
    while (c0 < n16) {
        const int64_t cn = c0 + (int64_t)D * kHistThreads;
        fetch(cn, nb, nq, ng);
#pragma unroll
        for (int d = 0; d < D; ++d) count(b[d], q[d], g[d]);
#pragma unroll
        for (int d = 0; d < D; ++d) { b[d] = nb[d]; q[d] = nq[d]; g[d] = ng[d]; }
        c0 = cn;
    }
}

// Four covered samples (every base byte 0..3, every qual byte 0..127): byte address of the counter =
// base << 14 | qual << 7 | copy << 2 (class = base * 128 + qual, 32 copies of 4 bytes), three VALU instructions each.
// lane_base = LDS address of the lane's copy of class 0 (the array is 64 KiB aligned in its address bits below 16).
__device__ __forceinline__ void count_word(uint32_t lane_base, uint32_t bw, uint32_t qw)
{
    lds_add_one(shl_byte<0>(bw, 14u) + shl_byte<0>(qw, 7u) + lane_base);   // disjoint bits: + is |, and one v_add3_u32
    lds_add_one(shl_byte<1>(bw, 14u) + shl_byte<1>(qw, 7u) + lane_base);   // disjoint bits: + is |, and one v_add3_u32
    lds_add_one(shl_byte<2>(bw, 14u) + shl_byte<2>(qw, 7u) + lane_base);   // disjoint bits: + is |, and one v_add3_u32
    lds_add_one(shl_byte<3>(bw, 14u) + shl_byte<3>(qw, 7u) + lane_base);   // disjoint bits: + is |, and one v_add3_u32
}

__device__ __forceinline__ void count_word_checked(uint32_t *__restrict__ hist, uint32_t bw, uint32_t qw,
                                                   uint32_t lane_off)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t b = (bw >> (8 * i)) & 0xFFu;
        const uint32_t q = (qw >> (8 * i)) & 0xFFu;
        if (b < 4u && q < 128u)                                   // covered sample
            hist_add(hist, ((b << 7) | q) * kCopies + lane_off);
    }
}

// 16 samples of one lane.  The common case (every sample of the wave covered) skips the per-sample test.
__device__ __forceinline__ void count_chunk(uint32_t *__restrict__ hist, const u32x4 b, const u32x4 q,
                                            uint32_t lane_off)
{
    const uint32_t bad = ((b.x | b.y | b.z | b.w) & 0xFCFCFCFCu) | ((q.x | q.y | q.z | q.w) & 0x80808080u);
    if (__ballot(bad != 0) == 0) {
        const uint32_t lane_base = lds_address(hist) + (lane_off << 2);
        count_word(lane_base, b.x, q.x); count_word(lane_base, b.y, q.y);
        count_word(lane_base, b.z, q.z); count_word(lane_base, b.w, q.w);
    } else {
        count_word_checked(hist, b.x, q.x, lane_off); count_word_checked(hist, b.y, q.y, lane_off);
        count_word_checked(hist, b.z, q.z, lane_off); count_word_checked(hist, b.w, q.w, lane_off);
    }
}

// The last, partial block of a run of 16-sample chunks [.., c1): lanes past the end load nothing and carry base bytes
// 0xFF, which the counting skips, so every wave that lies wholly inside the run still takes count_chunk's fast path
// (only the one wave that straddles c1 tests its samples one by one, and waves wholly outside do nothing).
__device__ __forceinline__ void count_partial_block(uint32_t *__restrict__ hist, const u32x4 *__restrict__ bv,
                                                    const u32x4 *__restrict__ qv, int64_t cb, int64_t c1, int tid,
                                                    uint32_t lane_off)
{
    u32x4 b[kUnroll], q[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
        const int64_t c = cb + tid + (int64_t)u * kHistThreads;
        b[u] = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        q[u] = u32x4{0u, 0u, 0u, 0u};
        if (c < c1) {
            b[u] = __builtin_nontemporal_load(&bv[c]);
            q[u] = __builtin_nontemporal_load(&qv[c]);
        }
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
        const int64_t wave_first = cb + (tid & ~63) + (int64_t)u * kHistThreads;     // wave-uniform
        if (wave_first < c1) count_chunk(hist, b[u], q[u], lane_off);
    }
}

// One workgroup per (site, split).  ALIGNED: row starts and n16 chunks are 16-byte aligned.
template <bool ALIGNED>
__global__ __launch_bounds__(kHistThreads) void hist_dense_kernel(
    int64_t n_sites, int64_t n_samples, int64_t row_stride, const int8_t *__restrict__ bases,
    const int8_t *__restrict__ quals, uint32_t *__restrict__ counts, int split)
{
    BVC_POISON_LDS();
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
                    count_partial_block(hist, bv, qv, cb, n16, tid, lane_off);
                }
            }
        } else {
            for (int64_t cb = (int64_t)part * kBlockChunks; cb < n16; cb += (int64_t)split * kBlockChunks) {
                const int64_t i1 = (cb + kBlockChunks < n16 ? cb + kBlockChunks : n16) * 16;
                for (int64_t i = cb * 16 + tid; i < i1; i += kHistThreads) {
                    const uint32_t b = (uint8_t)brow[i], q = (uint8_t)qrow[i];
                    if (b < 4u && q < 128u)
                        hist_add(hist, ((b << 7) | q) * kCopies + lane_off);
                }
            }
        }
        if (part == split - 1) {
            for (int64_t i = (n16 << 4) + tid; i < n_samples; i += kHistThreads) {
                const uint32_t b = (uint8_t)brow[i], q = (uint8_t)qrow[i];
                if (b < 4u && q < 128u)
                    hist_add(hist, ((b << 7) | q) * kCopies + lane_off);
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
// with as many copies as fit 64 KiB (copies = 1 << log2c); fewer copies than banks means bank conflicts
// (PMC: 69 % of this kernel's LDS cycles), which the spread of keys over groups softens.
//
// What bounds it (rocprofv3 counters, profiles/r02_pmc_groups.md): NOT a third load stream -- the group bytes are
// served by L2 and HBM traffic is 1.007 x the algorithmic bytes -- but issue: per sample one LDS atomic with ~3-way
// bank conflicts plus the VALU instructions that build its address, and under the EM kernels of the previous call
// the VALU is the shared resource.  Hence the fast kernel below spends 3.25 VALU instructions per sample (the
// compiler's selection for the plain expression needs more than eight).

// Generic form: any alignment, group vector as the caller gave it (labels >= n_groups mean "no group").
__global__ __launch_bounds__(kHistThreads) void hist_dense_groups_bytes_kernel(
    int64_t n_sites, int64_t n_samples, int64_t row_stride, const int8_t *__restrict__ bases,
    const int8_t *__restrict__ quals, const uint8_t *__restrict__ group_of_sample, int n_groups, int log2c,
    uint32_t *__restrict__ grp_counts, const int64_t *__restrict__ bounds)
{
    BVC_POISON_LDS();
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
    if (bounds && bounds[0] == 0) return;        // samples are ordered by group: hist_dense_ranges_kernel has the call
    const int tid = threadIdx.x;
    const int n_hist = n_groups + 1;
    const int words = (n_hist * BVC_NCLASS) << log2c;
    const uint32_t lane_off = (uint32_t)tid & ((1u << log2c) - 1u);
    for (int i = tid; i < words; i += kHistThreads) hist[i] = 0;
    __syncthreads();
    for (int64_t site = blockIdx.x; site < n_sites; site += gridDim.x) {
        const int8_t *brow = bases + site * row_stride;
        const int8_t *qrow = quals + site * row_stride;
        for (int64_t i = tid; i < n_samples; i += kHistThreads) {
            const uint32_t b = (uint8_t)brow[i], q = (uint8_t)qrow[i], g = group_of_sample[i];
            const uint32_t h = g < (uint32_t)n_groups ? g : (uint32_t)n_groups;
            if (b < 4u && q < 128u)
                hist_add(hist, ((h << (9 + log2c)) | (b << (7 + log2c)) | (q << log2c)) + lane_off);
        }
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

// Sum and clear the 1 << LOG2C copies of counter `slot` (layout [slot][copy]).  With four copies or more the lane reads
// and clears 16 bytes at a time, and the order in which it visits its quads is rotated by the slot so that the 16 lanes
// of a ds_read_b128 lane group (consecutive slots) touch 64 distinct banks: quad id = (slot * QUADS + j) mod 16 takes only
// 16 / QUADS values over 16 consecutive slots, and slots that share one differ in slot / (16 / QUADS) mod QUADS.  (The plain
// loop -- one dword at a time at a stride of 1 << LOG2C dwords across lanes -- is a 16-way conflict at 16 copies: a
// quarter of the 1024-thread kernel's time.)
template <int LOG2C>
__device__ __forceinline__ uint32_t fold_copies(uint32_t *hist, int slot)
{
    constexpr int C = 1 << LOG2C;
    uint32_t sum = 0;
    if (C >= 4) {
        constexpr int QUADS = C / 4;
        const int rot = QUADS > 1 ? slot / (16 / QUADS) : 0;
#pragma unroll
        for (int j = 0; j < QUADS; ++j) {
            u32x4 *p = reinterpret_cast<u32x4 *>(&hist[slot * C + 4 * ((j + rot) & (QUADS - 1))]);
            const u32x4 x = *p;
            sum += x.x + x.y + x.z + x.w;
            *p = u32x4{0u, 0u, 0u, 0u};
        }
    } else {
#pragma unroll
        for (int v = 0; v < C; ++v) {
            sum += hist[slot * C + v];
            hist[slot * C + v] = 0;
        }
    }
    return sum;
}

// Four covered samples of one lane in group mode: counter byte address =
//   hist << (11 + L) | base << (9 + L) | qual << (2 + L) | copy << 2      (L = log2 of the copies)
template <int LOG2C>
__device__ __forceinline__ void count_group_word(uint32_t lane_base, uint32_t bw, uint32_t qw, uint32_t gw)
{
    // counter = hist << (11 + LOG2C) | base << (9 + LOG2C) | qual << (2 + LOG2C) | copy << 2.  The histogram and base
    // fields are adjacent, so the four samples' (hist << 2 | base) bytes are formed by ONE instruction per word (bases
    // are 0..3 and labels <= 32 here: no carry between bytes) and each sample then costs two byte shifts and a
    // three-operand add: 3.25 VALU instructions per sample instead of 5.
    constexpr uint32_t SB = 9 + LOG2C, SQ = 2 + LOG2C;
    const uint32_t hb = (gw << 2) | bw;
    lds_add_one(shl_byte<0>(hb, SB) + shl_byte<0>(qw, SQ) + lane_base);
    lds_add_one(shl_byte<1>(hb, SB) + shl_byte<1>(qw, SQ) + lane_base);
    lds_add_one(shl_byte<2>(hb, SB) + shl_byte<2>(qw, SQ) + lane_base);
    lds_add_one(shl_byte<3>(hb, SB) + shl_byte<3>(qw, SQ) + lane_base);
}

// Fast form: 16-byte aligned rows; `hist_of_sample` is the group vector already clamped to 0..n_groups by
// group_bounds_kernel (one pass over 1 byte per sample per CALL, shared by all the sites of the tile).
// PIPE: the loads of the next two 16-sample chunks are issued before the current two are counted (stream_chunks).
template <int LOG2C, bool PIPE>
__global__ __launch_bounds__(kHistThreads) void hist_dense_groups_kernel(
    int64_t n_sites, int64_t n_samples, int64_t row_stride, const int8_t *__restrict__ bases,
    const int8_t *__restrict__ quals, const uint8_t *__restrict__ hist_of_sample, int n_groups,
    uint32_t *__restrict__ grp_counts, const int64_t *__restrict__ bounds, const uint8_t *__restrict__ only)
{
    BVC_POISON_LDS();
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
    if (bounds && bounds[0] == 0) return;        // samples are ordered by group: hist_dense_ranges_kernel has the call
    __builtin_amdgcn_s_setprio(3);               // as in hist_dense_kernel: ahead of the EM kernels it shares the chip with
    const int tid = threadIdx.x;
    const int n_hist = n_groups + 1;
    const int words = (n_hist * BVC_NCLASS) << LOG2C;
    const uint32_t lane_off = (uint32_t)tid & ((1u << LOG2C) - 1u);
    const uint32_t lane_base = lds_address(hist) + (lane_off << 2);
    for (int i = tid * 4; i < words; i += kHistThreads * 4)
        *reinterpret_cast<u32x4 *>(&hist[i]) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();

    auto add_checked = [&](uint32_t b, uint32_t q, uint32_t h) {
        if (b < 4u && q < 128u)
            hist_add(hist, ((h << (9 + LOG2C)) | (b << (7 + LOG2C)) | (q << LOG2C)) + lane_off);
    };
    // the common case (every sample of the wave covered) skips the per-sample test, as in count_chunk
    auto count16 = [&](const u32x4 b, const u32x4 q, const u32x4 g) {
        const uint32_t bad = ((b.x | b.y | b.z | b.w) & 0xFCFCFCFCu) | ((q.x | q.y | q.z | q.w) & 0x80808080u);
        if (__ballot(bad != 0) == 0) {
            count_group_word<LOG2C>(lane_base, b.x, q.x, g.x); count_group_word<LOG2C>(lane_base, b.y, q.y, g.y);
            count_group_word<LOG2C>(lane_base, b.z, q.z, g.z); count_group_word<LOG2C>(lane_base, b.w, q.w, g.w);
        } else {
            const uint32_t bw[4] = {b.x, b.y, b.z, b.w}, qw[4] = {q.x, q.y, q.z, q.w}, gw[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    add_checked((bw[w] >> (8 * k)) & 0xFFu, (qw[w] >> (8 * k)) & 0xFFu, (gw[w] >> (8 * k)) & 0xFFu);
        }
    };

    const int64_t n16 = n_samples >> 4;
    const u32x4 *gv = reinterpret_cast<const u32x4 *>(hist_of_sample);
    for (int64_t site = blockIdx.x; site < n_sites; site += gridDim.x) {
        if (only && !only[site]) continue;       // hist_dense_groups_slots_kernel has done this site
        const int8_t *brow = bases + site * row_stride;
        const int8_t *qrow = quals + site * row_stride;
        const u32x4 *bv = reinterpret_cast<const u32x4 *>(brow);
        const u32x4 *qv = reinterpret_cast<const u32x4 *>(qrow);
        auto load = [&](int64_t c, u32x4 &b, u32x4 &q, u32x4 &g) {
            b = __builtin_nontemporal_load(&bv[c]); q = __builtin_nontemporal_load(&qv[c]); g = gv[c];
        };
        if (PIPE) {
            stream_chunks<kGroupDepth>(n16, tid, load, count16);
        } else {
            int64_t c = tid;
            for (; c + kHistThreads < n16; c += 2 * kHistThreads) {        // two 16-byte loads per array in flight
                u32x4 b0, q0, g0, b1, q1, g1;
                load(c, b0, q0, g0);
                load(c + kHistThreads, b1, q1, g1);
                count16(b0, q0, g0);
                count16(b1, q1, g1);
            }
            for (; c < n16; c += kHistThreads) { u32x4 b0, q0, g0; load(c, b0, q0, g0); count16(b0, q0, g0); }
        }
        for (int64_t i = (n16 << 4) + tid; i < n_samples; i += kHistThreads)
            add_checked((uint8_t)brow[i], (uint8_t)qrow[i], hist_of_sample[i]);
        __syncthreads();
        for (int key = tid; key < n_hist * BVC_NCLASS; key += kHistThreads) {
            grp_counts[site * n_hist * BVC_NCLASS + key] = fold_copies<LOG2C>(hist, key);
        }
        __syncthreads();
    }
}

// Group mode, samples ordered by group (every group a contiguous run of columns, ungrouped samples last): the
// histogram of (site, group) is then the plain histogram of a column range, so the pass keeps the dense kernel's
// 32 conflict-free LDS copies and its wave-wide fast path, and needs no per-sample group byte.
// scratch[0] = 0 when group_of_sample is non-decreasing after clamping to n_groups, scratch[1 + h] = first sample of
// histogram h, scratch[1 + n_hist] = n_samples (scratch is zeroed before the launch).  The same pass writes the
// clamped labels (hist_of_sample) the any-order kernel indexes its histograms with.
__global__ void group_bounds_kernel(const uint8_t *__restrict__ group_of_sample, int64_t n_samples, int n_groups,
                                    int64_t *__restrict__ scratch, uint8_t *__restrict__ hist_of_sample)
{
    BVC_POISON_LDS();
    const int n_hist = n_groups + 1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_samples; i += (int64_t)gridDim.x * blockDim.x) {
        const int h = min((int)group_of_sample[i], n_groups);
        hist_of_sample[i] = (uint8_t)h;                          // label clamped to "no group" = n_groups
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
    BVC_POISON_LDS();
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];   // [class][copy]
    (void)n_samples;
    if (scratch[0] != 0) return;
    const int tid = threadIdx.x;
    const uint32_t lane_off = (uint32_t)(tid & (kCopies - 1));
    __builtin_amdgcn_s_setprio(3);
    // Work items are the NON-EMPTY column ranges only.  An empty one (typically "no group", when every sample has a
    // group) costs nothing but would still take a turn in the round-robin of items over workgroups -- and workgroups go
    // round-robin over the 8 XCDs, so with n_hist even the empty turns all fall on the same XCDs, which then run out of
    // work early (k = 5: four XCDs with 2/3 of the others' bytes, 1.34 ms instead of 1.23; k = 1: four XCDs idle, 1.63 ms).
    __shared__ uint8_t real_h[BVC_MAX_GROUPS + 1];
    __shared__ int n_real_s;
    if (tid == 0) {
        int c = 0;
        for (int h = 0; h < n_hist; ++h)
            if (scratch[1 + h] < scratch[2 + h]) real_h[c++] = (uint8_t)h;
        n_real_s = c;
    }
    for (int i = tid * 4; i < kLdsWords; i += kHistThreads * 4)
        *reinterpret_cast<u32x4 *>(&hist[i]) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    const int n_real = n_real_s;
    if (n_real < n_hist) {                               // the empty ranges' histograms: zeros, no LDS involved
        for (int64_t site = blockIdx.x; site < n_sites; site += gridDim.x)
            for (int h = 0; h < n_hist; ++h)
                if (scratch[1 + h] >= scratch[2 + h])
                    for (int key = tid; key < BVC_NCLASS; key += kHistThreads)
                        grp_counts[(site * n_hist + h) * BVC_NCLASS + key] = 0;
    }
    if (n_real == 0) return;

    auto scalar = [&](const int8_t *brow, const int8_t *qrow, int64_t i0, int64_t i1) {
        for (int64_t i = i0 + tid; i < i1; i += kHistThreads) {
            const uint32_t b = (uint8_t)brow[i], q = (uint8_t)qrow[i];
            if (b < 4u && q < 128u)
                hist_add(hist, ((b << 7) | q) * kCopies + lane_off);
        }
    };
    // A work item is one (site, histogram) = one column range of one row.  The loads of the NEXT item's first block are
    // issued before the barrier that ends the current one, so the workgroup has bytes in flight while it folds its
    // copies: with k = 5 a range is a fifth of a row and the fold comes five times as often as in hist_dense_kernel.
    struct Range { const int8_t *brow, *qrow; int64_t s0, s1, c0, c1, out; };
    auto range_of = [&](int64_t w) {
        const int64_t site = w / n_real;
        const int h = real_h[w % n_real];
        Range r;
        r.out = (site * n_hist + h) * BVC_NCLASS;
        r.brow = bases + site * row_stride;
        r.qrow = quals + site * row_stride;
        r.s0 = scratch[1 + h]; r.s1 = scratch[2 + h];
        r.c0 = (r.s0 + 15) >> 4; r.c1 = r.s1 >> 4;      // [s0, s1) = unaligned head, whole chunks [c0, c1), unaligned tail
        return r;
    };
    constexpr int64_t kBlockChunks = (int64_t)kUnroll * kHistThreads;
    u32x4 pb[kUnroll], pq[kUnroll];
    bool have = false;                                   // pb/pq hold chunks [c0, c0 + kBlockChunks) of the item
    auto prefetch = [&](const Range &r) {
        have = ALIGNED && r.c0 + kBlockChunks <= r.c1;
        if (have) {
            const u32x4 *bv = reinterpret_cast<const u32x4 *>(r.brow);
            const u32x4 *qv = reinterpret_cast<const u32x4 *>(r.qrow);
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                pb[u] = __builtin_nontemporal_load(&bv[r.c0 + tid + (int64_t)u * kHistThreads]);
                pq[u] = __builtin_nontemporal_load(&qv[r.c0 + tid + (int64_t)u * kHistThreads]);
            }
        }
    };
    const int64_t n_work = n_sites * n_real;
    // (Dealing each XCD a contiguous eighth of the item sequence instead of every eighth item measures the same:
    // profiles/r02_ranges_item_mapping.txt.)
    int64_t w = blockIdx.x;
    Range r{nullptr, nullptr, 0, 0, 0, 0, 0};
    if (w < n_work) { r = range_of(w); prefetch(r); }
    while (w < n_work) {
        const int8_t *brow = r.brow, *qrow = r.qrow;
        if (ALIGNED && r.c0 < r.c1) {
            const u32x4 *bv = reinterpret_cast<const u32x4 *>(brow);
            const u32x4 *qv = reinterpret_cast<const u32x4 *>(qrow);
            int64_t cb = r.c0;
            if (have) {
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) count_chunk(hist, pb[u], pq[u], lane_off);
                cb += kBlockChunks;
            }
            for (; cb + kBlockChunks <= r.c1; cb += kBlockChunks) {
                u32x4 b[kUnroll], q[kUnroll];
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    b[u] = __builtin_nontemporal_load(&bv[cb + tid + (int64_t)u * kHistThreads]);
                    q[u] = __builtin_nontemporal_load(&qv[cb + tid + (int64_t)u * kHistThreads]);
                }
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) count_chunk(hist, b[u], q[u], lane_off);
            }
            if (cb < r.c1) count_partial_block(hist, bv, qv, cb, r.c1, tid, lane_off);
            scalar(brow, qrow, r.s0, r.c0 << 4);
            scalar(brow, qrow, r.c1 << 4, r.s1);
        } else {
            scalar(brow, qrow, r.s0, r.s1);
        }
        const int64_t wn = w + gridDim.x;
        const int64_t out = r.out;
        have = false;
        if (wn < n_work) { r = range_of(wn); prefetch(r); }
        lds_barrier();
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
            grp_counts[out + key] = sum;
        }
        lds_barrier();
        w = wn;
    }
}

// Ragged (CSR) pileup: site s owns elements offsets[s] .. offsets[s+1]) of the concatenated arrays -- the vectors
// bt_f builds per position (/root/reference/src/BaseVarC.cpp:550-559).  Real pileups are ragged: at the depths of the
// reference's test data a site has 1-28 observations, at CMDB scale (1e6 samples, low coverage) 1e4-1e5.  Two kernels
// split the sites by length (each walks all sites and takes its own):
//   hist_wave_kernel       len <  kCsrLong: one WAVEFRONT per site, a private two-copy histogram per wave (4 KiB);
//                          short sites would otherwise pay a 64 KiB fold each.  The same kernel takes dense tiles
//                          with short rows (BASELINE configs[1]: 1e4 samples per site)
//   hist_csr_block_kernel  len >= kCsrLong: one workgroup per site with the dense kernel's 32 conflict-free copies,
//                          16-byte loads over the aligned middle of the range, head and tail sample by sample
// Ragged sites in the ONE-byte form (bvc_lrt_csr_packed: base << 6 | qual, qual <= 62, qual bits 63 = no observation):
// four bytes of a word become the base word and the qual word of the two-byte kernels; a "no observation" byte gets
// base byte 0xFF, which their counting skips.
__device__ __forceinline__ void unpack_word(uint32_t w, uint32_t &bw, uint32_t &qw)
{
    qw = w & 0x3F3F3F3Fu;
    const uint32_t none = ((qw + 0x01010101u) >> 6) & 0x01010101u;       // 1 in the bytes whose qual bits are 63
    bw = ((w >> 6) & 0x03030303u) | (none * 0xFFu);
}

__device__ __forceinline__ void unpack_chunk(const u32x4 p, u32x4 &b, u32x4 &q)
{
    uint32_t b0, b1, b2, b3, q0, q1, q2, q3;
    unpack_word(p.x, b0, q0); unpack_word(p.y, b1, q1); unpack_word(p.z, b2, q2); unpack_word(p.w, b3, q3);
    b = u32x4{b0, b1, b2, b3};
    q = u32x4{q0, q1, q2, q3};
}

constexpr int64_t kCsrLong = 4096;
constexpr int kCsrWaves = kHistThreads / 64;
constexpr int kWaveCopies = 2;                     // copies of a wave's private histogram (copy = lane & 1)

// One wavefront counts elements [s0, s1) of the arrays into its private LDS histogram [class][kWaveCopies].
// ALIGNED: both arrays start on a 16-byte boundary, so whole 16-element chunks of the range load as one dwordx4 per
// lane; head and tail element by element.
template <bool ALIGNED, bool PACKED = false>
__device__ __forceinline__ void wave_count_range(uint32_t *__restrict__ hist, const int8_t *__restrict__ bases,
                                                 const int8_t *__restrict__ quals, int64_t s0, int64_t s1, int lane)
{
    const uint32_t copy = (uint32_t)lane & (kWaveCopies - 1);
    auto one = [&](uint32_t b, uint32_t q) {
        if (b < 4u && q < 128u)
            hist_add(hist, ((b << 7) | q) * kWaveCopies + copy);
    };
    auto scalar = [&](int64_t i0, int64_t i1) {
        for (int64_t i = i0 + lane; i < i1; i += 64) {
            if (PACKED) { const uint32_t v = (uint8_t)bases[i]; one((v & 63u) == 63u ? 0xFFu : v >> 6, v & 63u); }
            else one((uint8_t)bases[i], (uint8_t)quals[i]);
        }
    };
    const int64_t c0 = (s0 + 15) >> 4, c1 = s1 >> 4;
    if (ALIGNED && c0 < c1) {
        scalar(s0, c0 << 4);
        const u32x4 *bv = reinterpret_cast<const u32x4 *>(bases);
        const u32x4 *qv = reinterpret_cast<const u32x4 *>(quals);
        for (int64_t c = c0 + lane; c < c1; c += 64) {
            u32x4 b = bv[c], q;
            if (PACKED) unpack_chunk(b, b, q); else q = qv[c];
            const uint32_t bw[4] = {b.x, b.y, b.z, b.w}, qw[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int k = 0; k < 4; ++k) one((bw[w] >> (8 * k)) & 0xFFu, (qw[w] >> (8 * k)) & 0xFFu);
        }
        scalar(c1 << 4, s1);
    } else {
        scalar(s0, s1);
    }
}

// One wavefront per site: short ragged sites (DENSE = false: [offsets[s], offsets[s+1]), the long ones are left to
// hist_csr_block_kernel) or short dense rows (DENSE = true: [s * row_stride, s * row_stride + n_samples)).
template <bool ALIGNED, bool DENSE, bool PACKED = false>
__global__ __launch_bounds__(kHistThreads) void hist_wave_kernel(
    int64_t n_sites, const int64_t *__restrict__ offsets, int64_t n_samples, int64_t row_stride,
    const int8_t *__restrict__ bases, const int8_t *__restrict__ quals, uint32_t *__restrict__ counts)
{
    BVC_POISON_LDS();
    __shared__ __attribute__((aligned(16))) uint32_t hist_all[kCsrWaves][BVC_NCLASS * kWaveCopies];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t *hist = hist_all[wave];
    for (int k = lane; k < BVC_NCLASS * kWaveCopies; k += 64) hist[k] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int64_t site = (int64_t)blockIdx.x * kCsrWaves + wave; site < n_sites; site += (int64_t)gridDim.x * kCsrWaves) {
        const int64_t o0 = DENSE ? site * row_stride : offsets[site];
        const int64_t o1 = DENSE ? o0 + n_samples : offsets[site + 1];
        if (!DENSE && o1 - o0 >= kCsrLong) continue;             // hist_csr_block_kernel's
        wave_count_range<ALIGNED, PACKED>(hist, bases, quals, o0, o1, lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        uint32_t *dst = counts + site * BVC_NCLASS;
#pragma unroll
        for (int k = 0; k < BVC_NCLASS / 64; ++k) {
            const int key = k * 64 + lane;
            uint32_t sum = 0;
#pragma unroll
            for (int v = 0; v < kWaveCopies; ++v) { sum += hist[key * kWaveCopies + v]; hist[key * kWaveCopies + v] = 0; }
            dst[key] = sum;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
}

// A long ragged site is 1e4-1e5 observations: a handful of load blocks, so what a site costs is not its bytes but the
// round trips to memory it waits for one after the other.  Round 3 had five or six of them per site (its two offsets, the
// unaligned head sample by sample, every block's loads before that block's counting, the tail) and read at 0.49 of the HBM
// peak; here the next site's offsets are loaded while the current site is counted, head and tail (at most 15 observations
// each) are loaded with the first block and counted last, and the loads of block k + 1 are issued before block k is counted.
// (At most 80 VGPRs: the kernel shares the chip with two stage-2 launches of the calls before it, whose wavefronts hold 168
// VGPRs each; a 512-thread workgroup needs two wavefronts on every SIMD of a CU at once, and 2 x 80 fit beside two of those
// where 2 x 100 would wait for one of them to retire.)
template <bool ALIGNED, bool PACKED = false>
__global__ __launch_bounds__(kHistThreads) BVC_WAVES_PER_EU(6, 8) void hist_csr_block_kernel(
    int64_t n_sites, const int64_t *__restrict__ offsets, const int8_t *__restrict__ bases,
    const int8_t *__restrict__ quals, uint32_t *__restrict__ counts)
{
    BVC_POISON_LDS();
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];   // [class][copy]
    const int tid = threadIdx.x;
    const uint32_t lane_off = (uint32_t)(tid & (kCopies - 1));
    __builtin_amdgcn_s_setprio(3);
    for (int i = tid * 4; i < kLdsWords; i += kHistThreads * 4)
        *reinterpret_cast<u32x4 *>(&hist[i]) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    // one observation given as the two-byte kernels see it (PACKED: the byte decoded first)
    auto one = [&](uint32_t b, uint32_t q) {
        if (PACKED) { q = b & 63u; b = q == 63u ? 0xFFu : b >> 6; }
        if (b < 4u && q < 128u) hist_add(hist, ((b << 7) | q) * kCopies + lane_off);
    };
    auto scalar = [&](int64_t i0, int64_t i1) {
        for (int64_t i = i0 + tid; i < i1; i += kHistThreads) one((uint8_t)bases[i], PACKED ? 0u : (uint32_t)(uint8_t)quals[i]);
    };
    constexpr int64_t kBlockChunks = (int64_t)kUnroll * kHistThreads;
    const u32x4 *bv = reinterpret_cast<const u32x4 *>(bases);
    const u32x4 *qv = reinterpret_cast<const u32x4 *>(quals);
    const u32x4 kSkip = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};   // "no observation" in either form
    // chunks [cb, cb + kBlockChunks) of a site's whole chunks [0, nc) (32-bit indices relative to the site's first chunk):
    // whole blocks load without a test
    auto fetch = [&](const u32x4 *sb, const u32x4 *sq, uint32_t cb, uint32_t nc, u32x4 (&b)[kUnroll], u32x4 (&q)[kUnroll]) {
        if (cb + (uint32_t)kBlockChunks <= nc) {
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                b[u] = __builtin_nontemporal_load(&sb[cb + (uint32_t)tid + (uint32_t)u * kHistThreads]);
                if (!PACKED) q[u] = __builtin_nontemporal_load(&sq[cb + (uint32_t)tid + (uint32_t)u * kHistThreads]);
            }
        } else {
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const uint32_t c = cb + (uint32_t)tid + (uint32_t)u * kHistThreads;
                b[u] = kSkip; q[u] = u32x4{0u, 0u, 0u, 0u};
                if (c < nc) {
                    b[u] = __builtin_nontemporal_load(&sb[c]);
                    if (!PACKED) q[u] = __builtin_nontemporal_load(&sq[c]);
                }
            }
        }
    };
    auto count = [&](uint32_t cb, uint32_t nc, const u32x4 (&b)[kUnroll], const u32x4 (&q)[kUnroll]) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const uint32_t wave_first = cb + (uint32_t)(tid & ~63) + (uint32_t)u * kHistThreads;  // wave-uniform
            if (wave_first >= nc) continue;                        // lanes past the end carry "no observation" and are skipped one by one
            if (PACKED) {
                u32x4 ub, uq;
                unpack_chunk(b[u], ub, uq);
                count_chunk(hist, ub, uq, lane_off);
            } else {
                count_chunk(hist, b[u], q[u], lane_off);
            }
        }
    };

    int64_t site = blockIdx.x;
    int64_t s0 = 0, s1 = 0;
    if (site < n_sites) { s0 = offsets[site]; s1 = offsets[site + 1]; }
    while (site < n_sites) {
        const int64_t next = site + gridDim.x;
        int64_t n0 = 0, n1 = 0;
        if (next < n_sites) { n0 = offsets[next]; n1 = offsets[next + 1]; }        // arrive while this site is counted
        if (s1 - s0 >= kCsrLong) {                                   // shorter ones: hist_wave_kernel's (workgroup-uniform)
            // [s0, s1) = unaligned head, whole 16-observation chunks [c0, c1) of the concatenated arrays, unaligned tail
            const int64_t c0 = (s0 + 15) >> 4, c1 = s1 >> 4;
            if (ALIGNED && c0 < c1) {
                const u32x4 *sb = bv + c0, *sq = qv + c0;
                const uint32_t nc = (uint32_t)(c1 - c0);
                u32x4 b[kUnroll], q[kUnroll], nb[kUnroll], nq[kUnroll];
                fetch(sb, sq, 0u, nc, b, q);
                // head and tail: at most 15 observations each, one per lane of the first 16; their loads travel with the first block's
                if (tid < 16) {
                    const int64_t hi = s0 + tid, ti = (c1 << 4) + tid;
                    uint32_t hb = 0xFFu, hq = 0u, tb = 0xFFu, tq = 0u;
                    if (hi < (c0 << 4)) { hb = (uint8_t)bases[hi]; if (!PACKED) hq = (uint8_t)quals[hi]; }
                    if (ti < s1) { tb = (uint8_t)bases[ti]; if (!PACKED) tq = (uint8_t)quals[ti]; }
                    one(hb, hq); one(tb, tq);                    // (0xFF = "no observation" in either form)
                }
                for (uint32_t cb = 0; cb < nc; cb += (uint32_t)kBlockChunks) {
                    const bool more = cb + (uint32_t)kBlockChunks < nc;
                    if (more) fetch(sb, sq, cb + (uint32_t)kBlockChunks, nc, nb, nq);
                    count(cb, nc, b, q);
                    if (more) {
#pragma unroll
                        for (int u = 0; u < kUnroll; ++u) { b[u] = nb[u]; q[u] = nq[u]; }
                    }
                }
            } else {
                scalar(s0, s1);
            }
            __syncthreads();
            for (int key = tid; key < BVC_NCLASS; key += kHistThreads) {
                uint32_t sum = 0;
#pragma unroll
                for (int v = 0; v < kCopies; v += 4) {
                    const int cc = (v + 4 * (key & 7)) & (kCopies - 1);      // rotated: the lanes of a ds_read_b128 group spread over banks
                    u32x4 *p = reinterpret_cast<u32x4 *>(&hist[key * kCopies + cc]);
                    const u32x4 x = *p;
                    sum += x.x + x.y + x.z + x.w;
                    *p = u32x4{0u, 0u, 0u, 0u};
                }
                counts[site * BVC_NCLASS + key] = sum;
            }
            __syncthreads();
        }
        site = next; s0 = n0; s1 = n1;
    }
}

// ---- packed rows: ONE byte per sample --------------------------------------------------------------------------
// The path is HBM-bound at 2 bytes per (site, sample), and the pair (base 0..3, qual 0..62) fits one byte:
//     packed = base << 6 | qual          0xFF (any byte with qual bits 63) = no observation
// Half the bytes, half the time of stage 1 -- if the counting keeps up: at one byte per sample the two stages together
// are bound by VALU issue, not by HBM.  So the byte IS the histogram slot and the counter address is built by ONE
// instruction: 256 slots x 64 copies (copy = lane: no two lanes of a wave ever share a counter) = 64 KiB of LDS at
// LDS address 0, counter address = slot << 8 | lane << 2, i.e. the sample's byte dropped into byte 1 of a register
// that already holds lane << 2 (v_mov_b32_sdwa dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE).  No "covered?" test
// either: an uncovered sample counts into slot 255, which the fold throws away.  Base qualities of 63 and more
// (PacBio HiFi reaches 93; Illumina stops at 41) do not fit: such tiles stay on the two-byte entry points
// (bvc_pack_dense counts them).
constexpr int kPackedSlots = 256;
constexpr int kPackedCopies = 64;                                // one per lane
constexpr int kPackedLdsWords = kPackedSlots * kPackedCopies;    // 16384 words = 64 KiB
constexpr int kPackedUnroll = 4;                                 // 16-byte loads in flight per lane (one array)

// addr = (addr & ~0xFF00) | byte BYTE of w << 8
template <int BYTE>
__device__ __forceinline__ void put_slot(uint32_t &addr, uint32_t w)
{
    if (BYTE == 0) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0" : "+v"(addr) : "v"(w));
    if (BYTE == 1) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1" : "+v"(addr) : "v"(w));
    if (BYTE == 2) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2" : "+v"(addr) : "v"(w));
    if (BYTE == 3) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3" : "+v"(addr) : "v"(w));
}

// Four samples; a0..a3 hold the lane's counter address bits outside byte 1 (four registers so that consecutive
// samples do not wait on one another's register).
__device__ __forceinline__ void count_packed_word(uint32_t &a0, uint32_t &a1, uint32_t &a2, uint32_t &a3, uint32_t w)
{
    put_slot<0>(a0, w); lds_add_one(a0);
    put_slot<1>(a1, w); lds_add_one(a1);
    put_slot<2>(a2, w); lds_add_one(a2);
    put_slot<3>(a3, w); lds_add_one(a3);
}

template <bool ALIGNED>
__global__ __launch_bounds__(kHistThreads) void hist_packed_kernel(
    int64_t n_sites, int64_t n_samples, int64_t row_stride, const uint8_t *__restrict__ packed,
    uint32_t *__restrict__ counts, int split)
{
    BVC_POISON_LDS();
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];   // [slot][copy], the kernel's only LDS: address 0
    const int tid = threadIdx.x;
    const uint32_t lane_base = lds_address(hist) + ((uint32_t)(tid & (kPackedCopies - 1)) << 2);
    // the one-instruction address needs bits 8..15 of the array's address to be zero (they are: see above)
    const bool at_zero = (lds_address(hist) & 0xFFFFu) == 0u;
    __builtin_amdgcn_s_setprio(3);                // as in hist_dense_kernel; priorities 0 and 1 measure 12 % and 1 % slower under the EM (profiles/r02_packed_sweep.txt)
    for (int i = tid * 4; i < kPackedLdsWords; i += kHistThreads * 4)
        *reinterpret_cast<u32x4 *>(&hist[i]) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();

    uint32_t a0 = lane_base, a1 = lane_base, a2 = lane_base, a3 = lane_base;
    const int64_t n_work = n_sites * split;
    for (int64_t w = blockIdx.x; w < n_work; w += gridDim.x) {
        const int64_t site = w / split;
        const int part = (int)(w % split);
        const uint8_t *row = packed + site * row_stride;
        const int64_t n16 = (ALIGNED && at_zero) ? n_samples >> 4 : 0;
        constexpr int64_t kBlockChunks = (int64_t)kPackedUnroll * kHistThreads;
        if (ALIGNED && at_zero) {
            const u32x4 *rv = reinterpret_cast<const u32x4 *>(row);
            for (int64_t cb = (int64_t)part * kBlockChunks; cb < n16; cb += (int64_t)split * kBlockChunks) {
                u32x4 v[kPackedUnroll];
                if (cb + kBlockChunks <= n16) {                  // a whole block: plain loads
#pragma unroll
                    for (int u = 0; u < kPackedUnroll; ++u) v[u] = __builtin_nontemporal_load(&rv[cb + tid + (int64_t)u * kHistThreads]);
                } else {                                         // the last, partial block: lanes past the end count "no observation"
#pragma unroll
                    for (int u = 0; u < kPackedUnroll; ++u) {
                        const int64_t c = cb + tid + (int64_t)u * kHistThreads;
                        v[u] = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
                        if (c < n16) v[u] = __builtin_nontemporal_load(&rv[c]);
                    }
                }
#pragma unroll
                for (int u = 0; u < kPackedUnroll; ++u) {
                    count_packed_word(a0, a1, a2, a3, v[u].x); count_packed_word(a0, a1, a2, a3, v[u].y);
                    count_packed_word(a0, a1, a2, a3, v[u].z); count_packed_word(a0, a1, a2, a3, v[u].w);
                }
            }
        }
        if (part == split - 1)                                   // the ragged tail (the whole row when not 16-byte aligned)
            for (int64_t i = (n16 << 4) + tid; i < n_samples; i += kHistThreads)
                lds_add_one(((uint32_t)row[i] << 8) + lane_base);
        __syncthreads();
        // fold the copies, publish in the [base][128 quals] form of the two-byte path, clear for the next site
        for (int key = tid; key < BVC_NCLASS; key += kHistThreads) {
            const int q = key & 127;
            uint32_t sum = 0;
            if (q < 64) {
                const int slot = ((key >> 7) << 6) | q;
#pragma unroll
                for (int c = 0; c < kPackedCopies; c += 4) {
                    const int cc = (c + 4 * (slot & 7)) & (kPackedCopies - 1);
                    u32x4 *p = reinterpret_cast<u32x4 *>(&hist[slot * kPackedCopies + cc]);
                    const u32x4 x = *p;
                    sum += x.x + x.y + x.z + x.w;
                    *p = u32x4{0u, 0u, 0u, 0u};
                }
                if (q == 63) sum = 0;                            // qual bits 63: no observation
            }
            if (split == 1) counts[site * BVC_NCLASS + key] = sum;
            else if (sum) atomicAdd(&counts[site * BVC_NCLASS + key], sum);
        }
        __syncthreads();
    }
}

// ---- packed rows in group mode ---------------------------------------------------------------------------------
// Samples ordered by group: a (site, group) histogram is the packed histogram of a column range -- hist_packed_kernel's
// counting (one VALU instruction per sample, 64 conflict-free copies) over hist_dense_ranges_kernel's work items (the
// non-empty ranges only).  scratch as written by group_bounds_kernel.
template <bool ALIGNED>
__global__ __launch_bounds__(kHistThreads) void hist_packed_ranges_kernel(
    int64_t n_sites, int64_t row_stride, const uint8_t *__restrict__ packed, int n_hist,
    const int64_t *__restrict__ scratch, uint32_t *__restrict__ grp_counts)
{
    BVC_POISON_LDS();
    // [slot][copy] at LDS address 0 (the one-instruction counter address needs that: no static LDS in this kernel),
    // followed by the table of non-empty ranges
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
    if (scratch[0] != 0) return;                                  // labels in any order: hist_packed_groups_kernel has the call
    uint8_t *real_h = reinterpret_cast<uint8_t *>(&hist[kPackedLdsWords]);
    int &n_real_s = *reinterpret_cast<int *>(&hist[kPackedLdsWords + 16]);
    const int tid = threadIdx.x;
    const uint32_t lane_base = lds_address(hist) + ((uint32_t)(tid & (kPackedCopies - 1)) << 2);
    const bool at_zero = (lds_address(hist) & 0xFFFFu) == 0u;     // the one-instruction address (hist_packed_kernel)
    __builtin_amdgcn_s_setprio(3);
    if (tid == 0) {
        int c = 0;
        for (int h = 0; h < n_hist; ++h)
            if (scratch[1 + h] < scratch[2 + h]) real_h[c++] = (uint8_t)h;
        n_real_s = c;
    }
    for (int i = tid * 4; i < kPackedLdsWords; i += kHistThreads * 4)
        *reinterpret_cast<u32x4 *>(&hist[i]) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    const int n_real = n_real_s;
    if (n_real < n_hist)                                          // the empty ranges' histograms: zeros
        for (int64_t site = blockIdx.x; site < n_sites; site += gridDim.x)
            for (int h = 0; h < n_hist; ++h)
                if (scratch[1 + h] >= scratch[2 + h])
                    for (int key = tid; key < BVC_NCLASS; key += kHistThreads)
                        grp_counts[(site * n_hist + h) * BVC_NCLASS + key] = 0;
    if (n_real == 0) return;

    uint32_t a0 = lane_base, a1 = lane_base, a2 = lane_base, a3 = lane_base;
    const int64_t n_work = n_sites * n_real;
    for (int64_t w = blockIdx.x; w < n_work; w += gridDim.x) {
        const int64_t site = w / n_real;
        const int h = real_h[w % n_real];
        const uint8_t *row = packed + site * row_stride;
        const int64_t s0 = scratch[1 + h], s1 = scratch[2 + h];
        auto scalar = [&](int64_t i0, int64_t i1) {
            for (int64_t i = i0 + tid; i < i1; i += kHistThreads) lds_add_one(((uint32_t)row[i] << 8) + lane_base);
        };
        const int64_t c0 = (s0 + 15) >> 4, c1 = s1 >> 4;          // [s0, s1) = head, whole 16-sample chunks [c0, c1), tail
        if (ALIGNED && at_zero && c0 < c1) {
            scalar(s0, c0 << 4);
            const u32x4 *rv = reinterpret_cast<const u32x4 *>(row);
            constexpr int64_t kBlockChunks = (int64_t)kPackedUnroll * kHistThreads;
            for (int64_t cb = c0; cb < c1; cb += kBlockChunks) {
                u32x4 v[kPackedUnroll];
                if (cb + kBlockChunks <= c1) {
#pragma unroll
                    for (int u = 0; u < kPackedUnroll; ++u) v[u] = __builtin_nontemporal_load(&rv[cb + tid + (int64_t)u * kHistThreads]);
                } else {
#pragma unroll
                    for (int u = 0; u < kPackedUnroll; ++u) {
                        const int64_t c = cb + tid + (int64_t)u * kHistThreads;
                        v[u] = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
                        if (c < c1) v[u] = __builtin_nontemporal_load(&rv[c]);
                    }
                }
#pragma unroll
                for (int u = 0; u < kPackedUnroll; ++u) {
                    count_packed_word(a0, a1, a2, a3, v[u].x); count_packed_word(a0, a1, a2, a3, v[u].y);
                    count_packed_word(a0, a1, a2, a3, v[u].z); count_packed_word(a0, a1, a2, a3, v[u].w);
                }
            }
            scalar(c1 << 4, s1);
        } else {
            scalar(s0, s1);
        }
        __syncthreads();
        const int64_t out = (site * n_hist + h) * BVC_NCLASS;
        for (int key = tid; key < BVC_NCLASS; key += kHistThreads) {
            const int q = key & 127;
            uint32_t sum = 0;
            if (q < 64) {
                const int slot = ((key >> 7) << 6) | q;
#pragma unroll
                for (int c = 0; c < kPackedCopies; c += 4) {
                    const int cc = (c + 4 * (slot & 7)) & (kPackedCopies - 1);
                    u32x4 *p = reinterpret_cast<u32x4 *>(&hist[slot * kPackedCopies + cc]);
                    const u32x4 x = *p;
                    sum += x.x + x.y + x.z + x.w;
                    *p = u32x4{0u, 0u, 0u, 0u};
                }
                if (q == 63) sum = 0;
            }
            grp_counts[out + key] = sum;
        }
        __syncthreads();
    }
}

// src1 (16-bit half HALF of w) << shift in one instruction (SDWA word select).
template <int HALF>
__device__ __forceinline__ uint32_t shl_half(uint32_t w, uint32_t shift)
{
    uint32_t r;
    if (HALF == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(shift), "v"(w));
    if (HALF == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(shift), "v"(w));
    return r;
}

// 1 << (byte BYTE of `amounts`) in one instruction (SDWA byte select on the shift amount).
template <int BYTE>
__device__ __forceinline__ uint32_t one_shl_byte(uint32_t amounts, uint32_t one)
{
    uint32_t r;
    if (BYTE == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r) : "v"(amounts), "v"(one));
    if (BYTE == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(amounts), "v"(one));
    if (BYTE == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r) : "v"(amounts), "v"(one));
    if (BYTE == 3) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r) : "v"(amounts), "v"(one));
    return r;
}

__device__ __forceinline__ void lds_add_value(uint32_t lds_byte_address, uint32_t v)
{
#ifdef BVC_CHECK_LDS
    if (!BVC_LDS_OK(3, lds_byte_address, lds_bytes_of_workgroup() - 3u)) return;
#endif
    __hip_atomic_fetch_add((lds_u32 *)(uintptr_t)lds_byte_address, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ---- round 5: 32 conflict-free copies with TWO 16-bit counters per LDS word (H16) --------------------------------------------
// The 16-copy forms above put two lanes of every 32-lane group on one bank whenever their slots have the same parity (bank =
// copy + 16 * (slot & 1)): SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 49 % (profiles/r04_pmc_summary.md).  Here slots 2j and 2j + 1
// share a word -- the add is 1 << 16 * (slot & 1) -- so [histogram][128 slot pairs][32 copies] takes the same 96 KiB at k = 5 and the
// bank of an update is lane mod 32 whatever the data, as in hist_dense_kernel.  A copy's counter sees at most N / 32 samples of a
// site: N <= kH16MaxSamples (the launcher keeps the 16-copy kernels beyond it).  Cost: the value of the add (1.5 VALU instructions
// per sample: one mask-and-shift per four samples, one SDWA shift per sample).
constexpr int64_t kH16MaxSamples = 2000000;
constexpr int kH16Threads = 1024;

// the four samples of one packed word and their four labels
__device__ __forceinline__ void count_word_h16(uint32_t pw, uint32_t gw, uint32_t copy4, uint32_t one)
{
    const uint32_t lo = __builtin_amdgcn_perm(gw, pw, 0x05010400u);   // [g1 p1 g0 p0]
    const uint32_t hi = __builtin_amdgcn_perm(gw, pw, 0x07030602u);   // [g3 p3 g2 p2]
    const uint32_t amounts = (pw & 0x01010101u) << 4;                 // 16 * (slot & 1) per sample
    // word address = (slot >> 1) * 128 + copy * 4 = ((slot << 6) & ~127) | copy << 2   (the histograms start at LDS address 0)
    lds_add_value((shl_half<0>(lo, 6) & 0xFFFFFF80u) | copy4, one_shl_byte<0>(amounts, one));
    lds_add_value((shl_half<1>(lo, 6) & 0xFFFFFF80u) | copy4, one_shl_byte<1>(amounts, one));
    lds_add_value((shl_half<0>(hi, 6) & 0xFFFFFF80u) | copy4, one_shl_byte<2>(amounts, one));
    lds_add_value((shl_half<1>(hi, 6) & 0xFFFFFF80u) | copy4, one_shl_byte<3>(amounts, one));
}

// Fold of one site: thread per slot pair; writes every class of every histogram (qualities 63..127 have no slot: zero).
__device__ __forceinline__ void fold_h16(uint32_t *hist, int n_hist, uint32_t *__restrict__ dst, int tid)
{
    for (int pk = tid; pk < n_hist * 128; pk += kH16Threads) {
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            u32x4 *p = reinterpret_cast<u32x4 *>(&hist[pk * 32 + 4 * ((j + pk) & 7)]);
            const u32x4 x = *p;
            lo += (x.x & 0xFFFFu) + (x.y & 0xFFFFu) + (x.z & 0xFFFFu) + (x.w & 0xFFFFu);
            hi += (x.x >> 16) + (x.y >> 16) + (x.z >> 16) + (x.w >> 16);
            *p = u32x4{0u, 0u, 0u, 0u};
        }
        const int h = pk >> 7, r = pk & 127, b = r >> 5, q = (2 * r) & 63;
        uint32_t *d = dst + h * BVC_NCLASS + b * 128 + q;
        d[0] = lo;
        d[1] = q + 1 == 63 ? 0u : hi;                                // slot 63 of an allele: "no observation"
        d[64] = 0u; d[65] = 0u;
    }
}

// hist_packed_groups_kernel's job (labels in any order, packed rows) with the H16 counters; 16-byte aligned rows only.
__global__ __launch_bounds__(kH16Threads) void hist_packed_groups_h16_kernel(
    int64_t n_sites, int64_t n_samples, int64_t row_stride, const uint8_t *__restrict__ packed,
    const uint8_t *__restrict__ hist_of_sample, int n_groups, uint32_t *__restrict__ grp_counts,
    const int64_t *__restrict__ bounds)
{
    BVC_POISON_LDS();
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];   // [hist][128 slot pairs][32 copies]
    if (bounds[0] == 0) return;                                   // ordered by group: hist_packed_ranges_kernel has the call
    __builtin_amdgcn_s_setprio(3);
    constexpr int THREADS = kH16Threads;
    const int tid = threadIdx.x;
    const int n_hist = n_groups + 1;
    const int words = n_hist * 128 * 32;
    const uint32_t copy4 = lds_address(hist) + (((uint32_t)tid & 31u) << 2);
    uint32_t one = 1u;
    asm volatile("" : "+v"(one));                                 // a VGPR for the SDWA shifts
    for (int i = tid * 4; i < words; i += THREADS * 4)
        *reinterpret_cast<u32x4 *>(&hist[i]) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    const int64_t n16 = n_samples >> 4;
    const u32x4 *gv = reinterpret_cast<const u32x4 *>(hist_of_sample);
    for (int64_t site = blockIdx.x; site < n_sites; site += gridDim.x) {
        const uint8_t *row = packed + site * row_stride;
        const u32x4 *rv = reinterpret_cast<const u32x4 *>(row);
        constexpr int64_t kBlockChunks = 2 * (int64_t)THREADS;
        for (int64_t cb = 0; cb < n16; cb += kBlockChunks) {
            u32x4 p[2], g[2];
            if (cb + kBlockChunks <= n16) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int64_t c = cb + tid + (int64_t)u * THREADS;
                    p[u] = __builtin_nontemporal_load(&rv[c]); g[u] = gv[c];
                }
            } else {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int64_t c = cb + tid + (int64_t)u * THREADS;
                    p[u] = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};   // past the end: slot 255 ...
                    g[u] = u32x4{0u, 0u, 0u, 0u};                                       // ... of histogram 0
                    if (c < n16) { p[u] = __builtin_nontemporal_load(&rv[c]); g[u] = gv[c]; }
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                count_word_h16(p[u].x, g[u].x, copy4, one); count_word_h16(p[u].y, g[u].y, copy4, one);
                count_word_h16(p[u].z, g[u].z, copy4, one); count_word_h16(p[u].w, g[u].w, copy4, one);
            }
        }
        for (int64_t i = (n16 << 4) + tid; i < n_samples; i += THREADS) {
            const uint32_t slot = ((uint32_t)hist_of_sample[i] << 8) | row[i];
            lds_add_value(((slot >> 1) << 7) + copy4, 1u << (16u * (slot & 1u)));
        }
        __syncthreads();
        fold_h16(hist, n_hist, grp_counts + site * n_hist * BVC_NCLASS, tid);
        __syncthreads();
    }
}

// Labels in any order on packed rows: LDS [hist][256 slots][copies], copies = 1 << LOG2C as many as fit 64 KiB (8 at
// k = 5, against 4 of the two-byte kernel: half the bank conflicts), counter byte address =
// (label << 8 | packed byte) << (2 + LOG2C) | copy << 2.  The 16-bit (label, byte) pairs of two samples are put
// together by one v_perm_b32, each sample then takes one SDWA shift and one add: 2.5 VALU instructions per sample, two
// load streams instead of three, no "covered?" test (0xFF counts into slot 255 of its histogram and is dropped).
template <int LOG2C, bool ALIGNED, int THREADS = kHistThreads>
__global__ __launch_bounds__(THREADS) void hist_packed_groups_kernel(
    int64_t n_sites, int64_t n_samples, int64_t row_stride, const uint8_t *__restrict__ packed,
    const uint8_t *__restrict__ hist_of_sample, int n_groups, uint32_t *__restrict__ grp_counts,
    const int64_t *__restrict__ bounds)
{
    BVC_POISON_LDS();
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
    if (bounds[0] == 0) return;                                   // ordered by group: hist_packed_ranges_kernel has the call
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x;
    const int n_hist = n_groups + 1;
    const int words = (n_hist * kPackedSlots) << LOG2C;
    const uint32_t lane_off = (uint32_t)tid & ((1u << LOG2C) - 1u);
    const uint32_t lane_base = lds_address(hist) + (lane_off << 2);
    for (int i = tid * 4; i < words; i += THREADS * 4)
        *reinterpret_cast<u32x4 *>(&hist[i]) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    constexpr uint32_t SH = 2 + LOG2C;
    // round 5: each sample's 16-bit (label, byte) key by ONE v_perm_b32 ([0 0 g p], the zero bytes from selector 0x0C) and its counter
    // address by ONE v_lshl_add_u32: 2 VALU instructions per sample where two perms per word + an SDWA shift + an add were 2.5.
    // (The kernel answers to its VALU count, not to its LDS conflicts: profiles/r05_group_h16.txt.)
    auto count_word = [&](uint32_t pw, uint32_t gw) {
        lds_add_one((__builtin_amdgcn_perm(gw, pw, 0x0C0C0400u) << SH) + lane_base);
        lds_add_one((__builtin_amdgcn_perm(gw, pw, 0x0C0C0501u) << SH) + lane_base);
        lds_add_one((__builtin_amdgcn_perm(gw, pw, 0x0C0C0602u) << SH) + lane_base);
        lds_add_one((__builtin_amdgcn_perm(gw, pw, 0x0C0C0703u) << SH) + lane_base);
    };
    const int64_t n16 = ALIGNED ? n_samples >> 4 : 0;
    const u32x4 *gv = reinterpret_cast<const u32x4 *>(hist_of_sample);
    for (int64_t site = blockIdx.x; site < n_sites; site += gridDim.x) {
        const uint8_t *row = packed + site * row_stride;
        if (ALIGNED) {
            const u32x4 *rv = reinterpret_cast<const u32x4 *>(row);
            constexpr int64_t kBlockChunks = 2 * (int64_t)THREADS;
            for (int64_t cb = 0; cb < n16; cb += kBlockChunks) {
                u32x4 p[2], g[2];
                if (cb + kBlockChunks <= n16) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int64_t c = cb + tid + (int64_t)u * THREADS;
                        p[u] = __builtin_nontemporal_load(&rv[c]); g[u] = gv[c];
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int64_t c = cb + tid + (int64_t)u * THREADS;
                        p[u] = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};   // past the end: slot 255 ...
                        g[u] = u32x4{0u, 0u, 0u, 0u};                                       // ... of histogram 0
                        if (c < n16) { p[u] = __builtin_nontemporal_load(&rv[c]); g[u] = gv[c]; }
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    count_word(p[u].x, g[u].x); count_word(p[u].y, g[u].y);
                    count_word(p[u].z, g[u].z); count_word(p[u].w, g[u].w);
                }
            }
        }
        for (int64_t i = (n16 << 4) + tid; i < n_samples; i += THREADS)
            lds_add_one(((((uint32_t)hist_of_sample[i] << 8) | row[i]) << SH) + lane_base);
        __syncthreads();
        for (int key = tid; key < n_hist * BVC_NCLASS; key += THREADS) {
            const int h = key >> 9, cls = key & 511, q = cls & 127;
            uint32_t sum = 0;
            if (q < 64) {
                const int slot = (h << 8) | ((cls >> 7) << 6) | q;
                sum = fold_copies<LOG2C>(hist, slot);
                if (q == 63) sum = 0;
            }
            grp_counts[site * n_hist * BVC_NCLASS + key] = sum;
        }
        __syncthreads();
    }
}

// Labels in any order on TWO-BYTE rows, 4..8 groups: the rows are packed in registers -- a 16-sample chunk whose bases
// are all 0..3 and whose qualities are all below 63 becomes (b << 6) | q per word, one instruction per four samples -- and
// counted the way hist_packed_groups_kernel<4, true, 1024> counts packed rows: 256 slots x 16 copies per histogram, one
// 1024-thread workgroup per CU, every bank exactly two deep (the general two-byte kernel above: 512 classes x 4 copies,
// three to four deep).  Any other chunk goes sample by sample; a site with a covered sample of quality 63..127 -- no
// place in 256 slots -- is flagged in `redo`, and hist_dense_groups_kernel runs after this kernel on the flagged sites only.
template <int LOG2C>
__global__ __launch_bounds__(1024) void hist_dense_groups_slots_kernel(
    int64_t n_sites, int64_t n_samples, int64_t row_stride, const int8_t *__restrict__ bases,
    const int8_t *__restrict__ quals, const uint8_t *__restrict__ hist_of_sample, int n_groups,
    uint32_t *__restrict__ grp_counts, const int64_t *__restrict__ bounds, uint8_t *__restrict__ redo)
{
    BVC_POISON_LDS();
    constexpr int THREADS = 1024;
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];   // [hist][slot 256][copy 16], then one word: the site's flag
    if (bounds[0] == 0) return;                                   // ordered by group: hist_dense_ranges_kernel has the call
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x;
    const int n_hist = n_groups + 1;
    const int words = (n_hist * kPackedSlots) << LOG2C;
    // the site's flag: one LDS word behind the histograms, reached through an LDS-typed address (ds_write_b32 / ds_read_b32;
    // a volatile generic pointer would make these FLAT accesses through the LDS aperture)
    const uint32_t redo_at = lds_address(hist) + ((uint32_t)words << 2);
    const uint32_t lane_off = (uint32_t)tid & ((1u << LOG2C) - 1u);
    const uint32_t lane_base = lds_address(hist) + (lane_off << 2);
    for (int i = tid * 4; i < words; i += THREADS * 4)
        *reinterpret_cast<u32x4 *>(&hist[i]) = u32x4{0u, 0u, 0u, 0u};
    if (tid == 0) lds_store(redo_at, 0u);
    __syncthreads();
    constexpr uint32_t SH = 2 + LOG2C;
    // round 5: each sample's 16-bit (label, byte) key by ONE v_perm_b32 ([0 0 g p], the zero bytes from selector 0x0C) and its counter
    // address by ONE v_lshl_add_u32: 2 VALU instructions per sample where two perms per word + an SDWA shift + an add were 2.5.
    // (The kernel answers to its VALU count, not to its LDS conflicts: profiles/r05_group_h16.txt.)
    auto count_word = [&](uint32_t pw, uint32_t gw) {
        lds_add_one((__builtin_amdgcn_perm(gw, pw, 0x0C0C0400u) << SH) + lane_base);
        lds_add_one((__builtin_amdgcn_perm(gw, pw, 0x0C0C0501u) << SH) + lane_base);
        lds_add_one((__builtin_amdgcn_perm(gw, pw, 0x0C0C0602u) << SH) + lane_base);
        lds_add_one((__builtin_amdgcn_perm(gw, pw, 0x0C0C0703u) << SH) + lane_base);
    };
    bool flagged = false;                                         // this lane met a covered sample of quality 63..127
    auto count_sample = [&](uint32_t b, uint32_t q, uint32_t h) {  // the two-byte rule, one sample
        if (b < 4u && q < 63u) lds_add_one((((h << 8) | (b << 6) | q) << SH) + lane_base);
        else if (b < 4u && q < 128u) flagged = true;
    };
    const int64_t n16 = n_samples >> 4;
    const u32x4 *gv = reinterpret_cast<const u32x4 *>(hist_of_sample);
    for (int64_t site = blockIdx.x; site < n_sites; site += gridDim.x) {
        const int8_t *brow = bases + site * row_stride;
        const int8_t *qrow = quals + site * row_stride;
        const u32x4 *bv = reinterpret_cast<const u32x4 *>(brow);
        const u32x4 *qv = reinterpret_cast<const u32x4 *>(qrow);
        constexpr int64_t kBlockChunks = 2 * (int64_t)THREADS;
        for (int64_t cb = 0; cb < n16; cb += kBlockChunks) {
            u32x4 b[2], q[2], g[2];
            bool in[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int64_t c = cb + tid + (int64_t)u * THREADS;
                in[u] = cb + kBlockChunks <= n16 || c < n16;
                b[u] = u32x4{0u, 0u, 0u, 0u}; q[u] = b[u]; g[u] = b[u];
                if (in[u]) { b[u] = __builtin_nontemporal_load(&bv[c]); q[u] = __builtin_nontemporal_load(&qv[c]); g[u] = gv[c]; }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                // every base 0..3 and every quality below 63?  (q + 1 reaches bit 6 from 63 on; a carry out of a byte can
                // only send a chunk the slow way needlessly)
                const uint32_t qo = q[u].x | q[u].y | q[u].z | q[u].w;
                const uint32_t q1 = (q[u].x + 0x01010101u) | (q[u].y + 0x01010101u) | (q[u].z + 0x01010101u) | (q[u].w + 0x01010101u);
                const uint32_t bad = ((b[u].x | b[u].y | b[u].z | b[u].w) & 0xFCFCFCFCu) | ((qo | q1) & 0xC0C0C0C0u);
                if (__ballot(in[u] && bad != 0) == 0) {
                    if (in[u]) {
                        count_word((b[u].x << 6) | q[u].x, g[u].x); count_word((b[u].y << 6) | q[u].y, g[u].y);
                        count_word((b[u].z << 6) | q[u].z, g[u].z); count_word((b[u].w << 6) | q[u].w, g[u].w);
                    }
                } else if (in[u]) {
                    // some sample of the wavefront is not a covered one of quality below 63 (tiles with coverage below 1: nearly
                    // every chunk): the bytes are sorted out four at a time -- a sample that is not covered becomes 0xFF
                    // (slot 255, dropped by the fold), a covered one of quality 63 and more flags the site -- and counted
                    // like the others.  (Sample by sample this path was VALU-bound: 2.26 ms per 4000 sites at 70 %
                    // coverage against 1.37 at full coverage.)
                    const uint32_t bw[4] = {b[u].x, b[u].y, b[u].z, b[u].w}, qw[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
                    const uint32_t gw[4] = {g[u].x, g[u].y, g[u].z, g[u].w};
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        // 0x80 in every byte of x that is not zero
                        auto nonzero = [](uint32_t x) { return (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u; };
                        const uint32_t b_out = nonzero(bw[w] & 0xFCFCFCFCu);                       // not a base: no observation
                        // quality 63 and more: q + 1 reaches bit 6 (a carry out of a byte of 0xFF can only flag the next one)
                        const uint32_t q_out = nonzero(((qw[w] + 0x01010101u) | qw[w]) & 0xC0C0C0C0u);
                        flagged |= (q_out & ~b_out) != 0u;
                        const uint32_t drop = ((b_out | q_out) >> 7) * 0xFFu;                      // 0xFF in the bytes to drop
                        count_word((((bw[w] & 0x03030303u) << 6) | qw[w]) | drop, gw[w]);
                    }
                }
            }
        }
        for (int64_t i = (n16 << 4) + tid; i < n_samples; i += THREADS)
            count_sample((uint8_t)brow[i], (uint8_t)qrow[i], hist_of_sample[i]);
        if (flagged) lds_store(redo_at, 1u);
        flagged = false;
        __syncthreads();
        for (int key = tid; key < n_hist * BVC_NCLASS; key += THREADS) {
            const int h = key >> 9, cls = key & 511, qq = cls & 127;
            uint32_t sum = 0;
            if (qq < 64) {
                const int slot = (h << 8) | ((cls >> 7) << 6) | qq;
                sum = fold_copies<LOG2C>(hist, slot);
                if (qq == 63) sum = 0;                            // only chunks past a row's end count there
            }
            grp_counts[site * n_hist * BVC_NCLASS + key] = sum;
        }
        if (tid == 0) { redo[site] = (uint8_t)lds_load(redo_at); lds_store(redo_at, 0u); }
        __syncthreads();
    }
}

// (bases, quals) -> packed bytes.  bad += covered samples whose quality does not fit (63..127): written as "no
// observation", so a caller that finds bad != 0 must not use the packed tile.
__global__ void pack_dense_kernel(int64_t n_sites, int64_t n_samples, int64_t stride_in, const int8_t *__restrict__ bases,
                                  const int8_t *__restrict__ quals, int64_t stride_out, uint8_t *__restrict__ packed,
                                  unsigned long long *__restrict__ bad)
{
    BVC_POISON_LDS();
    const int64_t total = n_sites * n_samples;
    unsigned long long mine = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = i / n_samples, k = i - s * n_samples;
        const uint32_t b = (uint8_t)bases[s * stride_in + k], q = (uint8_t)quals[s * stride_in + k];
        uint8_t v = 0xFF;
        if (b < 4u && q < 63u) v = (uint8_t)(b << 6 | q);
        else if (b < 4u && q < 128u) ++mine;
        packed[s * stride_out + k] = v;
    }
    if (mine) atomicAdd(bad, mine);
}

// Plain streaming read, 16 B per lane, nothing else: the empirical HBM read ceiling the histogram kernel is
// compared with next to the 8 TB/s spec figure.  The XOR keeps the loads alive.  Shape = the best of
// tools/micro/read_bw.hip's grid-stride sweeps over a line-aligned 8 GB tile (profiles/r02_read_bw_aligned.txt): four
// non-temporal loads in flight per lane, 16384 workgroups of 512 (6.80 TB/s; the round-2 shape, 4096 x 512 with two
// loads, reads 6.38 and was below the kernel it is meant to cap).
__global__ __launch_bounds__(512) void stream_read_kernel(const u32x4 *__restrict__ src, int64_t n16, uint32_t *__restrict__ sink)
{
    BVC_POISON_LDS();
    uint32_t acc = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const u32x4 a = __builtin_nontemporal_load(&src[i]), b = __builtin_nontemporal_load(&src[i + stride]);
        const u32x4 c = __builtin_nontemporal_load(&src[i + 2 * stride]), d = __builtin_nontemporal_load(&src[i + 3 * stride]);
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < n16; i += stride) { const u32x4 a = __builtin_nontemporal_load(&src[i]); acc ^= a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x9E3779B9u) sink[0] = acc;       // practically never: the store only makes the result observable
}

}  // namespace

#ifdef BVC_CHECK_LDS
BVC_DEFINE_DEBUG_READER(debug_read_hist)
#endif

hipError_t launch_stream_read(hipStream_t stream, const void *src, int64_t bytes, uint32_t *sink)
{
    const int64_t n16 = bytes / 16;
    if (n16 <= 0) return hipSuccess;
    int64_t blocks = (n16 + 4 * 512 - 1) / (4 * 512);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(stream_read_kernel, dim3((unsigned)blocks), dim3(512), 0, stream, reinterpret_cast<const u32x4 *>(src), n16, sink);
    return hipGetLastError();
}

int choose_hist_split(const LaunchState &st, int64_t n_sites, int64_t n_samples)
{
    if (st.hist_split > 0) return st.hist_split;
    // Two 512-thread workgroups fit a CU (2 x 64 KiB LDS).  With fewer sites than that, cut each site's
    // sample range so the whole chip streams; parts are merged with global atomics on 512 words.
    const int64_t want = (int64_t)st.n_cu * 4;
    if (n_sites >= want || n_samples < (1 << 16)) return 1;
    int64_t split = (want + n_sites - 1) / n_sites;
    const int64_t max_split = n_samples / (1 << 15);             // keep >= 32k samples per part
    if (split > max_split) split = max_split;
    if (split < 1) split = 1;
    if (split > 64) split = 64;
    return (int)split;
}

// Kernels with more than 48 KiB of dynamic LDS need the attribute raised once per device; the context remembers
// which of its kernels have been done (no process-wide state).
enum KernelSlot : uint32_t {
    kSlotDense0 = 0, kSlotDense1, kSlotRanges0, kSlotRanges1, kSlotCsr0, kSlotCsr1, kSlotGroupByte,
    kSlotGroup = 8,            // + log2c (0..5)
    kSlotGroupPipe = 16,       // + log2c (0..5)
    kSlotPacked0 = 24, kSlotPacked1 = 25, kSlotPackedRanges0 = 26, kSlotPackedRanges1 = 27,
    kSlotPackedGroups = 32,    // + 6 * aligned + log2c (0..5)
    kSlotCsrPacked0 = 44, kSlotCsrPacked1 = 45,
    kSlotPackedGroupsBig = 46,  // + (4 - log2 copies): 46..48
    kSlotGroupSlots = 49,       // + (4 - log2 copies): 49..51
    kSlotPackedGroupsH16 = 52, kSlotGroupSlotsH16 = 53,
};
constexpr size_t kBigLdsBytes = 144 * 1024;      // a workgroup may take the CU's whole LDS (160 KiB); stage 2 keeps 16 KiB beside it

static hipError_t raise_lds(LaunchState &st, uint32_t slot, const void *kernel, size_t bytes)
{
    if (st.attr_done & ((uint64_t)1 << slot)) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) st.attr_done |= (uint64_t)1 << slot;
    return e;
}

hipError_t launch_hist_dense(LaunchState &st, hipStream_t stream, int64_t n_sites, int64_t n_samples, int64_t row_stride,
                             const int8_t *bases, const int8_t *quals, const uint8_t *group_of_sample,
                             int n_groups, uint32_t *counts, int split, int64_t *group_scratch, uint8_t *hist_of_sample)
{
    if (n_sites <= 0) return hipSuccess;
    const bool aligned = ((reinterpret_cast<uintptr_t>(bases) | reinterpret_cast<uintptr_t>(quals)) & 15u) == 0 &&
                         (row_stride & 15) == 0;
    const size_t lds = (size_t)kLdsWords * sizeof(uint32_t);
    if (group_of_sample) {                       // counts = [site][n_groups + 1][512]
        const int n_hist = n_groups + 1;
        if (n_samples <= 0)                      // no columns at all: every histogram is empty
            return hipMemsetAsync(counts, 0, (size_t)n_sites * n_hist * BVC_NCLASS * sizeof(uint32_t), stream);
        if (!group_scratch || !hist_of_sample) return hipErrorInvalidValue;
        // Decided on the device, without a host round trip: the bounds kernel marks whether the samples are ordered by
        // group (and writes the clamped labels); the range kernel and the any-order kernel are both launched and the
        // one whose turn it is not returns at once.
        auto rk = aligned ? hist_dense_ranges_kernel<true> : hist_dense_ranges_kernel<false>;
        hipError_t e = raise_lds(st, aligned ? kSlotRanges1 : kSlotRanges0, reinterpret_cast<const void *>(rk), lds);
        if (e != hipSuccess) return e;
        e = hipMemsetAsync(group_scratch, 0, (size_t)kGroupScratchWords * sizeof(int64_t), stream);
        if (e != hipSuccess) return e;
        const int64_t bgrid = (n_samples + 255) / 256;
        hipLaunchKernelGGL(group_bounds_kernel, dim3((unsigned)(bgrid < 1024 ? bgrid : 1024)), dim3(256), 0, stream,
                           group_of_sample, n_samples, n_groups, group_scratch, hist_of_sample);
        const int64_t n_work = n_sites * n_hist;
        hipLaunchKernelGGL(rk, dim3((unsigned)(n_work < 4096 ? n_work : 4096)), dim3(kHistThreads), lds, stream,
                           n_sites, n_samples, row_stride, bases, quals, n_hist, group_scratch, counts);
        int log2c = 0;
        while (log2c < 5 && (size_t)n_hist * BVC_NCLASS * (2u << log2c) <= (size_t)kLdsWords) ++log2c;
        if (st.group_log2c >= 0 && st.group_log2c < log2c) log2c = st.group_log2c;
        const size_t glds = ((size_t)n_hist * BVC_NCLASS << log2c) * sizeof(uint32_t);
        // 33 histograms (32 groups + "no group") of one copy each are 66 KiB: the attribute is raised to that once
        constexpr size_t kGroupLdsMax = (size_t)(BVC_MAX_GROUPS + 1) * BVC_NCLASS * sizeof(uint32_t);
        const int64_t ggrid = n_sites < 4096 ? n_sites : 4096;
        if (aligned) {                           // hist_of_sample is the context's own 256-byte aligned buffer
            using FastKernel = void (*)(int64_t, int64_t, int64_t, const int8_t *, const int8_t *, const uint8_t *, int,
                                        uint32_t *, const int64_t *, const uint8_t *);
            static const FastKernel fast[2][6] = {
                {hist_dense_groups_kernel<0, false>, hist_dense_groups_kernel<1, false>, hist_dense_groups_kernel<2, false>,
                 hist_dense_groups_kernel<3, false>, hist_dense_groups_kernel<4, false>, hist_dense_groups_kernel<5, false>},
                {hist_dense_groups_kernel<0, true>, hist_dense_groups_kernel<1, true>, hist_dense_groups_kernel<2, true>,
                 hist_dense_groups_kernel<3, true>, hist_dense_groups_kernel<4, true>, hist_dense_groups_kernel<5, true>}};
            const int pipe = st.group_pipe ? 1 : 0;
            const FastKernel fk = fast[pipe][log2c];
            e = raise_lds(st, (pipe ? kSlotGroupPipe : kSlotGroup) + log2c, reinterpret_cast<const void *>(fk), kGroupLdsMax);
            if (e != hipSuccess) return e;
            // 4..8 groups: packed in registers and counted in 256 slots x 16 copies first; the general kernel then takes the
            // sites that kernel flags (a covered sample of quality 63 or more), all sites otherwise
            const uint8_t *only = nullptr;
            // 256 slots x 16 / 8 / 4 copies per histogram in one workgroup's LDS: up to 9 / 18 / 36 histograms
            int sl = 4;
            while (sl > 2 && ((size_t)n_hist * kPackedSlots << sl) * sizeof(uint32_t) + 16 > kBigLdsBytes) --sl;
            const size_t slds = ((size_t)n_hist * kPackedSlots << sl) * sizeof(uint32_t) + 16;
            if (st.group_big_lds && log2c < 3 && slds <= kBigLdsBytes) {
                using SK = void (*)(int64_t, int64_t, int64_t, const int8_t *, const int8_t *, const uint8_t *, int, uint32_t *,
                                    const int64_t *, uint8_t *);
                const SK sk = sl == 4 ? hist_dense_groups_slots_kernel<4> : (sl == 3 ? hist_dense_groups_slots_kernel<3> : hist_dense_groups_slots_kernel<2>);
                e = raise_lds(st, kSlotGroupSlots + (4 - sl), reinterpret_cast<const void *>(sk), kBigLdsBytes);
                if (e != hipSuccess) return e;
                uint8_t *redo = hist_of_sample + group_redo_offset(n_samples);
                hipLaunchKernelGGL(sk, dim3((unsigned)ggrid), dim3(1024), slds, stream, n_sites, n_samples,
                                   row_stride, bases, quals, hist_of_sample, n_groups, counts, group_scratch, redo);
                only = redo;
            }
            hipLaunchKernelGGL(fk, dim3((unsigned)ggrid), dim3(kHistThreads), glds, stream, n_sites, n_samples, row_stride,
                               bases, quals, hist_of_sample, n_groups, counts, group_scratch, only);
        } else {
            e = raise_lds(st, kSlotGroupByte, reinterpret_cast<const void *>(hist_dense_groups_bytes_kernel), kGroupLdsMax);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(hist_dense_groups_bytes_kernel, dim3((unsigned)ggrid), dim3(kHistThreads), glds, stream,
                               n_sites, n_samples, row_stride, bases, quals, group_of_sample, n_groups, log2c, counts,
                               group_scratch);
        }
        return hipGetLastError();
    }
    if (split == 1 && n_samples <= kWaveRowMax && n_sites >= (int64_t)st.n_cu * 2 * kCsrWaves) {
        // short rows, many sites: one wavefront per site (a 64 KiB fold per 10 KB row is what the block kernel would pay)
        auto wk = aligned ? hist_wave_kernel<true, true> : hist_wave_kernel<false, true>;
        const int64_t wgrid = (n_sites + kCsrWaves - 1) / kCsrWaves;
        hipLaunchKernelGGL(wk, dim3((unsigned)(wgrid < 8192 ? wgrid : 8192)), dim3(kHistThreads), 0, stream, n_sites,
                           (const int64_t *)nullptr, n_samples, row_stride, bases, quals, counts);
        return hipGetLastError();
    }
    auto kern = aligned ? hist_dense_kernel<true> : hist_dense_kernel<false>;
    hipError_t e = raise_lds(st, aligned ? kSlotDense1 : kSlotDense0, reinterpret_cast<const void *>(kern), lds);
    if (e != hipSuccess) return e;
    const int64_t n_work = n_sites * split;
    const int64_t grid = n_work < 4096 ? n_work : 4096;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kHistThreads), lds, stream, n_sites, n_samples,
                       row_stride, bases, quals, counts, split);
    return hipGetLastError();
}

hipError_t launch_hist_packed(LaunchState &st, hipStream_t stream, int64_t n_sites, int64_t n_samples, int64_t row_stride,
                              const uint8_t *packed, uint32_t *counts, int split)
{
    if (n_sites <= 0) return hipSuccess;
    const bool aligned = (reinterpret_cast<uintptr_t>(packed) & 15u) == 0 && (row_stride & 15) == 0;
    auto kern = aligned ? hist_packed_kernel<true> : hist_packed_kernel<false>;
    const size_t lds = (size_t)kPackedLdsWords * sizeof(uint32_t);
    hipError_t e = raise_lds(st, aligned ? kSlotPacked1 : kSlotPacked0, reinterpret_cast<const void *>(kern), lds);
    if (e != hipSuccess) return e;
    const int64_t n_work = n_sites * split;
    hipLaunchKernelGGL(kern, dim3((unsigned)(n_work < 4096 ? n_work : 4096)), dim3(kHistThreads), lds, stream, n_sites,
                       n_samples, row_stride, packed, counts, split);
    return hipGetLastError();
}

hipError_t launch_hist_packed_groups(LaunchState &st, hipStream_t stream, int64_t n_sites, int64_t n_samples,
                                     int64_t row_stride, const uint8_t *packed, const uint8_t *group_of_sample, int n_groups,
                                     uint32_t *counts, int64_t *group_scratch, uint8_t *hist_of_sample)
{
    if (n_sites <= 0) return hipSuccess;
    const int n_hist = n_groups + 1;
    if (n_samples <= 0)
        return hipMemsetAsync(counts, 0, (size_t)n_sites * n_hist * BVC_NCLASS * sizeof(uint32_t), stream);
    if (!group_scratch || !hist_of_sample) return hipErrorInvalidValue;
    const bool aligned = (reinterpret_cast<uintptr_t>(packed) & 15u) == 0 && (row_stride & 15) == 0;
    // as launch_hist_dense in group mode: the bounds kernel decides on the device which of the two kernels has the call
    hipError_t e = hipMemsetAsync(group_scratch, 0, (size_t)kGroupScratchWords * sizeof(int64_t), stream);
    if (e != hipSuccess) return e;
    const int64_t bgrid = (n_samples + 255) / 256;
    hipLaunchKernelGGL(group_bounds_kernel, dim3((unsigned)(bgrid < 1024 ? bgrid : 1024)), dim3(256), 0, stream,
                       group_of_sample, n_samples, n_groups, group_scratch, hist_of_sample);
    auto rk = aligned ? hist_packed_ranges_kernel<true> : hist_packed_ranges_kernel<false>;
    const size_t lds = (size_t)kPackedLdsWords * sizeof(uint32_t);
    e = raise_lds(st, aligned ? kSlotPackedRanges1 : kSlotPackedRanges0, reinterpret_cast<const void *>(rk), lds + 256);
    if (e != hipSuccess) return e;
    const int64_t n_work = n_sites * n_hist;
    hipLaunchKernelGGL(rk, dim3((unsigned)(n_work < 4096 ? n_work : 4096)), dim3(kHistThreads), lds + 256, stream, n_sites,
                       row_stride, packed, n_hist, group_scratch, counts);
    int log2c = 0;
    while (log2c < 5 && (size_t)n_hist * kPackedSlots * (2u << log2c) <= (size_t)kLdsWords) ++log2c;
    if (st.group_log2c >= 0 && st.group_log2c < log2c) log2c = st.group_log2c;
    const size_t glds = ((size_t)n_hist * kPackedSlots << log2c) * sizeof(uint32_t);
    using GK = void (*)(int64_t, int64_t, int64_t, const uint8_t *, const uint8_t *, int, uint32_t *, const int64_t *);
    static const GK gk[2][6] = {
        {hist_packed_groups_kernel<0, false>, hist_packed_groups_kernel<1, false>, hist_packed_groups_kernel<2, false>,
         hist_packed_groups_kernel<3, false>, hist_packed_groups_kernel<4, false>, hist_packed_groups_kernel<5, false>},
        {hist_packed_groups_kernel<0, true>, hist_packed_groups_kernel<1, true>, hist_packed_groups_kernel<2, true>,
         hist_packed_groups_kernel<3, true>, hist_packed_groups_kernel<4, true>, hist_packed_groups_kernel<5, true>}};
    // One workgroup of 1024 threads with 16 copies per histogram (bank = copy + 16 * (slot & 1): exactly two lanes of a
    // 32-lane group on a bank) when that fits the CU's LDS and the 64 KiB form would have fewer copies: 4 <= k <= 8 groups.
    // Alone it is 3 % slower than three 512-thread workgroups with 8 copies (0.78 against 0.76 ms at k = 5, N = 1e6, 4000
    // sites), underneath stage 2 -- the way the calls run -- 5 % faster (0.90 against 0.95) and stage 2 itself a fifth
    // (profiles/r03_group_anyorder_experiments.txt).  The same form of the two-byte kernel (8 copies, 96 KiB) LOSES 15 %
    // underneath stage 2 and is not kept.
    // round 5: two 16-bit counters per word, 32 conflict-free copies (same LDS as 16 copies of words); st.group_h16 picks it
    const size_t hlds = (size_t)n_hist * 128 * 32 * sizeof(uint32_t);
    if (st.group_h16 && st.group_big_lds && aligned && log2c < 4 && hlds <= kBigLdsBytes && n_samples <= kH16MaxSamples) {
        e = raise_lds(st, kSlotPackedGroupsH16, reinterpret_cast<const void *>(hist_packed_groups_h16_kernel), kBigLdsBytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(hist_packed_groups_h16_kernel, dim3((unsigned)(n_sites < 4096 ? n_sites : 4096)), dim3(kH16Threads), hlds, stream,
                           n_sites, n_samples, row_stride, packed, hist_of_sample, n_groups, counts, group_scratch);
        return hipGetLastError();
    }
    int bl = 4;                                                  // 16 / 8 / 4 copies in one workgroup's LDS: up to 9 / 18 / 36 histograms
    while (bl > 2 && ((size_t)n_hist * kPackedSlots << bl) * sizeof(uint32_t) > kBigLdsBytes) --bl;
    if (st.group_big_lds && aligned && log2c < bl && ((size_t)n_hist * kPackedSlots << bl) * sizeof(uint32_t) <= kBigLdsBytes) {
        const GK bk = bl == 4 ? hist_packed_groups_kernel<4, true, 1024> : (bl == 3 ? hist_packed_groups_kernel<3, true, 1024> : hist_packed_groups_kernel<2, true, 1024>);
        const size_t blds = ((size_t)n_hist * kPackedSlots << bl) * sizeof(uint32_t);
        e = raise_lds(st, kSlotPackedGroupsBig + (4 - bl), reinterpret_cast<const void *>(bk), kBigLdsBytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(bk, dim3((unsigned)(n_sites < 4096 ? n_sites : 4096)), dim3(1024), blds, stream, n_sites,
                           n_samples, row_stride, packed, hist_of_sample, n_groups, counts, group_scratch);
        return hipGetLastError();
    }
    const GK k = gk[aligned ? 1 : 0][log2c];
    e = raise_lds(st, kSlotPackedGroups + (aligned ? 6 : 0) + log2c, reinterpret_cast<const void *>(k), (size_t)kLdsWords * sizeof(uint32_t));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3((unsigned)(n_sites < 4096 ? n_sites : 4096)), dim3(kHistThreads), glds, stream, n_sites,
                       n_samples, row_stride, packed, hist_of_sample, n_groups, counts, group_scratch);
    return hipGetLastError();
}

hipError_t launch_pack_dense(hipStream_t stream, int64_t n_sites, int64_t n_samples, int64_t stride_in, const int8_t *bases,
                             const int8_t *quals, int64_t stride_out, uint8_t *packed, unsigned long long *bad)
{
    if (n_sites <= 0 || n_samples <= 0) return hipSuccess;
    hipLaunchKernelGGL(pack_dense_kernel, dim3(4096), dim3(256), 0, stream, n_sites, n_samples, stride_in, bases, quals,
                       stride_out, packed, bad);
    return hipGetLastError();
}

hipError_t launch_hist_csr(LaunchState &st, hipStream_t stream, int64_t n_sites, const int64_t *offsets,
                           const int8_t *bases, const int8_t *quals, uint32_t *counts)
{
    if (n_sites <= 0) return hipSuccess;
    const bool packed = quals == nullptr;                        // one byte per observation in `bases`
    // both arrays are indexed by the same element offsets, so one alignment test covers every site
    const bool aligned = ((reinterpret_cast<uintptr_t>(bases) | reinterpret_cast<uintptr_t>(quals)) & 15u) == 0;
    const size_t lds = (size_t)kLdsWords * sizeof(uint32_t);
    auto bk = packed ? (aligned ? hist_csr_block_kernel<true, true> : hist_csr_block_kernel<false, true>)
                     : (aligned ? hist_csr_block_kernel<true> : hist_csr_block_kernel<false>);
    hipError_t e = raise_lds(st, (aligned ? kSlotCsr1 : kSlotCsr0) + (packed ? kSlotCsrPacked0 - kSlotCsr0 : 0),
                             reinterpret_cast<const void *>(bk), lds);
    if (e != hipSuccess) return e;
    const int64_t wgrid = (n_sites + kCsrWaves - 1) / kCsrWaves;
    auto wk = packed ? (aligned ? hist_wave_kernel<true, false, true> : hist_wave_kernel<false, false, true>)
                     : (aligned ? hist_wave_kernel<true, false> : hist_wave_kernel<false, false>);
    hipLaunchKernelGGL(wk, dim3((unsigned)(wgrid < 8192 ? wgrid : 8192)), dim3(kHistThreads), 0, stream,
                       n_sites, offsets, (int64_t)0, (int64_t)0, bases, quals, counts);
    hipLaunchKernelGGL(bk, dim3((unsigned)(n_sites < 4096 ? n_sites : 4096)), dim3(kHistThreads), lds, stream, n_sites,
                       offsets, bases, quals, counts);
    return hipGetLastError();
}

}  // namespace bvc
