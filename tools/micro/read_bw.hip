// read_bw.hip -- what read bandwidth does a streaming kernel reach on this part, and with which access shape?
// (gfx950 microbenchmark, not product code.)  Variants: bytes in flight per lane, block size, grid size,
// row-per-block (the histogram kernel's shape: one 1 MB row per workgroup, two streams) vs grid-stride.
// build: hipcc --offload-arch=gfx950 -O3 -w -o tools/micro/read_bw tools/micro/read_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT>
__global__ void gs_kernel(const u32x4 *__restrict__ src, int64_t n16, uint32_t *sink)
{
    uint32_t acc = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(&src[i + u * stride]) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n16; i += stride) { const u32x4 a = src[i]; acc ^= a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x9E3779B9u) sink[0] = acc;
}

// one (or `rows_per_block`) contiguous row(s) per workgroup, two arrays read in lockstep (bases + quals)
template <int UNROLL, bool NT>
__global__ void row_kernel(const u32x4 *__restrict__ a, const u32x4 *__restrict__ b, int64_t row16, int64_t n_rows, uint32_t *sink)
{
    uint32_t acc = 0;
    for (int64_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
        const u32x4 *pa = a + r * row16, *pb = b + r * row16;
        int64_t i = threadIdx.x;
        for (; i + (UNROLL - 1) * blockDim.x < row16; i += UNROLL * blockDim.x) {
            u32x4 va[UNROLL], vb[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                va[u] = NT ? __builtin_nontemporal_load(&pa[i + u * blockDim.x]) : pa[i + u * blockDim.x];
                vb[u] = NT ? __builtin_nontemporal_load(&pb[i + u * blockDim.x]) : pb[i + u * blockDim.x];
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc ^= va[u].x ^ va[u].y ^ va[u].z ^ va[u].w ^ vb[u].x ^ vb[u].y ^ vb[u].z ^ vb[u].w;
        }
        for (; i < row16; i += blockDim.x) { const u32x4 x = pa[i], y = pb[i]; acc ^= x.x ^ y.y ^ x.z ^ y.w; }
    }
    if (acc == 0x9E3779B9u) sink[0] = acc;
}

// S workgroups share a row, taking its 16-byte-per-lane chunks round robin (part p takes chunks p, p+S, ...)
template <int UNROLL>
__global__ void coop_kernel(const u32x4 *__restrict__ a, const u32x4 *__restrict__ b, int64_t row16, int64_t n_rows, int S, uint32_t *sink)
{
    uint32_t acc = 0;
    const int64_t r = blockIdx.x / S;
    const int part = blockIdx.x % S;
    const u32x4 *pa = a + r * row16, *pb = b + r * row16;
    const int64_t chunk = (int64_t)blockDim.x * UNROLL;
    for (int64_t c = (int64_t)part * chunk; c < row16; c += (int64_t)S * chunk) {
        u32x4 va[UNROLL], vb[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t i = c + u * blockDim.x + threadIdx.x;
            va[u] = i < row16 ? __builtin_nontemporal_load(&pa[i]) : u32x4{0, 0, 0, 0};
            vb[u] = i < row16 ? __builtin_nontemporal_load(&pb[i]) : u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= va[u].x ^ va[u].y ^ va[u].z ^ va[u].w ^ vb[u].x ^ vb[u].y ^ vb[u].z ^ vb[u].w;
    }
    if (acc == 0x9E3779B9u) sink[0] = acc;
}

// The histogram kernel's real constraints on the cooperative shape: 64 KiB of LDS per workgroup (two per CU),
// zeroed before and folded after every (row, part) item.  DYNAMIC: persistent workgroups take items in order
// from a global counter and issue the first loads of the next item before folding the current one.
template <bool DYNAMIC>
__global__ __launch_bounds__(512) void coop_lds_kernel(const u32x4 *__restrict__ a, const u32x4 *__restrict__ b, int64_t row16,
                                                       int64_t n_rows, int S, uint32_t *sink, unsigned int *counter)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    __shared__ unsigned int s_item;
    const int tid = threadIdx.x;
    const int64_t n_items = n_rows * S;
    uint32_t acc = 0;
    for (int i = tid * 4; i < 16384; i += 512 * 4) *reinterpret_cast<u32x4 *>(&lds[i]) = u32x4{0, 0, 0, 0};
    int64_t item;
    if (DYNAMIC) {
        if (tid == 0) s_item = atomicAdd(counter, 1u);
        __syncthreads();
        item = s_item;
    } else {
        item = blockIdx.x;
    }
    const int64_t chunk = 512;                                   // 16-byte units per block step (u1)
    // first loads of the first item
    u32x4 va = u32x4{0, 0, 0, 0}, vb = va;
    int64_t r = 0, c = 0; int part = 0;
    if (item < n_items) {
        r = item / S; part = (int)(item % S); c = (int64_t)part * chunk + tid;
        if (c < row16) { va = __builtin_nontemporal_load(&a[r * row16 + c]); vb = __builtin_nontemporal_load(&b[r * row16 + c]); }
    }
    while (item < n_items) {
        const u32x4 *pa = a + r * row16, *pb = b + r * row16;
        // stream the item: software-pipelined, the next block's loads are issued before the current one is used
        for (;;) {
            const int64_t cn = c + (int64_t)S * chunk;
            u32x4 na = u32x4{0, 0, 0, 0}, nb = na;
            const bool more = (cn - tid) < row16;                // block-uniform
            if (more && cn < row16) { na = __builtin_nontemporal_load(&pa[cn]); nb = __builtin_nontemporal_load(&pb[cn]); }
            acc ^= va.x ^ va.y ^ va.z ^ va.w ^ vb.x ^ vb.y ^ vb.z ^ vb.w;
            if (!more) break;
            va = na; vb = nb; c = cn;
        }
        int64_t next = n_items;
        if (DYNAMIC) {
            __syncthreads();
            if (tid == 0) s_item = atomicAdd(counter, 1u);
            __syncthreads();
            next = s_item;
            if (next < n_items) {                                 // prefetch the next item's first block before the fold
                r = next / S; part = (int)(next % S); c = (int64_t)part * chunk + tid;
                va = u32x4{0, 0, 0, 0}; vb = va;
                if (c < row16) { va = __builtin_nontemporal_load(&a[r * row16 + c]); vb = __builtin_nontemporal_load(&b[r * row16 + c]); }
            }
        }
        __syncthreads();
        // fold: every thread sums the 32 copies of one class and clears them
        uint32_t sum = 0;
        for (int v = 0; v < 32; v += 4) {
            const int cc = (v + 4 * (tid & 7)) & 31;
            u32x4 *p = reinterpret_cast<u32x4 *>(&lds[tid * 32 + cc]);
            const u32x4 x = *p; sum += x.x + x.y + x.z + x.w;
            *p = u32x4{0, 0, 0, 0};
        }
        if (sum) atomicAdd(&sink[1 + (tid & 7)], sum);
        __syncthreads();
        item = next;
    }
    if (acc == 0x9E3779B9u) sink[0] = acc;
}

template <typename F>
static double best_ms(F launch, int reps = 21)       // median of `reps` timed launches
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    std::vector<float> t;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main(int argc, char **argv)
{
    // the bench tile: 4000 sites x 1e6 samples, 2 arrays; argv[1] = row stride in bytes (round 1: 1000000, rows start
    // mid-line on odd sites; round 2's tiles: 1000064, every row on a 128-byte line)
    const int64_t n_rows = 4000, row_bytes = argc > 1 ? atoll(argv[1]) : 1000064;
    const int64_t bytes_each = n_rows * row_bytes, total = 2 * bytes_each;
    char *buf; uint32_t *sink;
    hipMalloc(&buf, total); hipMalloc(&sink, 64);
    hipMemset(buf, 1, total);
    const u32x4 *src = reinterpret_cast<const u32x4 *>(buf);
    const u32x4 *pa = src, *pb = reinterpret_cast<const u32x4 *>(buf + bytes_each);
    const int64_t n16 = total / 16, row16 = row_bytes / 16;
    auto report = [&](const char *name, int blocks, int threads, double ms) {
        printf("%-34s grid %6d x %4d  %.3f ms  %.0f GB/s\n", name, blocks, threads, ms, total / (ms * 1e-3) / 1e9);
        fflush(stdout);
    };
    for (int threads : {256, 512, 1024})
        for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
            report("grid-stride u2 nt", blocks, threads, best_ms([&] { hipLaunchKernelGGL((gs_kernel<2, true>), dim3(blocks), dim3(threads), 0, 0, src, n16, sink); }));
            report("grid-stride u4 nt", blocks, threads, best_ms([&] { hipLaunchKernelGGL((gs_kernel<4, true>), dim3(blocks), dim3(threads), 0, 0, src, n16, sink); }));
        }
    report("grid-stride u4 plain", 4096, 512, best_ms([&] { hipLaunchKernelGGL((gs_kernel<4, false>), dim3(4096), dim3(512), 0, 0, src, n16, sink); }));
    report("grid-stride u8 nt", 4096, 512, best_ms([&] { hipLaunchKernelGGL((gs_kernel<8, true>), dim3(4096), dim3(512), 0, 0, src, n16, sink); }));
    report("grid-stride u1 nt", 8192, 512, best_ms([&] { hipLaunchKernelGGL((gs_kernel<1, true>), dim3(8192), dim3(512), 0, 0, src, n16, sink); }));
    for (int threads : {256, 512, 1024})
        for (int blocks : {512, 1024, 2048, 4000}) {
            report("row-per-block u1 nt (2 arrays)", blocks, threads, best_ms([&] { hipLaunchKernelGGL((row_kernel<1, true>), dim3(blocks), dim3(threads), 0, 0, pa, pb, row16, n_rows, sink); }));
            report("row-per-block u2 nt (2 arrays)", blocks, threads, best_ms([&] { hipLaunchKernelGGL((row_kernel<2, true>), dim3(blocks), dim3(threads), 0, 0, pa, pb, row16, n_rows, sink); }));
            report("row-per-block u4 nt (2 arrays)", blocks, threads, best_ms([&] { hipLaunchKernelGGL((row_kernel<4, true>), dim3(blocks), dim3(threads), 0, 0, pa, pb, row16, n_rows, sink); }));
        }
    for (int S : {2, 4, 8, 16, 32, 64}) {
        char name[64];
        snprintf(name, sizeof name, "coop S=%d u2 (2 arrays)", S);
        report(name, (int)n_rows * S, 512, best_ms([&] { hipLaunchKernelGGL((coop_kernel<2>), dim3((unsigned)(n_rows * S)), dim3(512), 0, 0, pa, pb, row16, n_rows, S, sink); }));
        snprintf(name, sizeof name, "coop S=%d u1 (2 arrays)", S);
        report(name, (int)n_rows * S, 512, best_ms([&] { hipLaunchKernelGGL((coop_kernel<1>), dim3((unsigned)(n_rows * S)), dim3(512), 0, 0, pa, pb, row16, n_rows, S, sink); }));
    }
    {
        unsigned int *counter; hipMalloc(&counter, 4);
        hipFuncSetAttribute(reinterpret_cast<const void *>(coop_lds_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipFuncSetAttribute(reinterpret_cast<const void *>(coop_lds_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        for (int S : {1, 2, 4, 8, 16, 32}) {
            char name[64];
            snprintf(name, sizeof name, "coop+LDS fold S=%d, 1 item/WG", S);
            report(name, (int)n_rows * S, 512, best_ms([&] { hipLaunchKernelGGL((coop_lds_kernel<false>), dim3((unsigned)(n_rows * S)), dim3(512), 65536, 0, pa, pb, row16, n_rows, S, sink, counter); }));
            snprintf(name, sizeof name, "coop+LDS fold S=%d, dynamic", S);
            report(name, 512, 512, best_ms([&] { hipMemsetAsync(counter, 0, 4, 0); hipLaunchKernelGGL((coop_lds_kernel<true>), dim3(512), dim3(512), 65536, 0, pa, pb, row16, n_rows, S, sink, counter); }));
        }
    }
    report("row-per-block u2 plain", 4000, 512, best_ms([&] { hipLaunchKernelGGL((row_kernel<2, false>), dim3(4000), dim3(512), 0, 0, pa, pb, row16, n_rows, sink); }));
    return 0;
}
