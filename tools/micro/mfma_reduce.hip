// mfma_reduce.hip -- can the FP64 matrix cores do the EM kernel's cross-lane sums?  (gfx950)
//   v_mfma_f64_4x4x4_4b_f64 : 4 blocks of 16 lanes; within a block lane l holds A[i = l & 3][k = (l >> 2) & 3] and
//                             B[k = (l >> 2) & 3][j = l & 3]; one f64 result per lane.
//   v_mfma_f64_16x16x4_f64  : lane l holds A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15]; 4 results per lane,
//                             col = l & 15, row = (l >> 4) + 4 * reg.
// Test 1: prints what each lane holds after  r1 = mfma4x4x4(v, 1, 0)  and  r2 = mfma4x4x4(r1, 1, 0)  for v = distinct
//         powers of two per lane, so that every sum can be decoded.  Hoped for: r2 = sum over the 16 lanes of the block.
// Test 2: cross-block total with mfma16x16x4(1, v, 0).
// Test 3: throughput of (2 x 4x4x4 + 16x16x4) per iteration against the DPP / permlane reduction it would replace,
//         alone and interleaved with independent FP64 VALU work.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_reduce mfma_reduce.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(double *out)
{
    const int l = threadIdx.x;
    const double v = (double)(1ull << (l & 15)) + (double)(l >> 4) * 65536.0 * 0;   // bit (l & 15): which lanes were summed
    const double r1 = __builtin_amdgcn_mfma_f64_4x4x4f64(v, 1.0, 0.0, 0, 0, 0);
    const double r2 = __builtin_amdgcn_mfma_f64_4x4x4f64(r1, 1.0, 0.0, 0, 0, 0);
    const double r1b = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, v, 0.0, 0, 0, 0);
    const double r2b = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, r1b, 0.0, 0, 0, 0);
    const double r2c = __builtin_amdgcn_mfma_f64_4x4x4f64(r1b, 1.0, 0.0, 0, 0, 0);
    const double r2d = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, r1, 0.0, 0, 0, 0);
    // cross-block: one value per block (block index as a power of 16)
    const double w = (double)(1ull << (4 * (l >> 4)));
    d4 z = {0, 0, 0, 0};
    const d4 t = __builtin_amdgcn_mfma_f64_16x16x4f64(1.0, w, z, 0, 0, 0);
    const d4 t2 = __builtin_amdgcn_mfma_f64_16x16x4f64(w, 1.0, z, 0, 0, 0);
    out[l * 12 + 0] = r1; out[l * 12 + 1] = r2; out[l * 12 + 2] = r1b; out[l * 12 + 3] = r2b;
    out[l * 12 + 4] = r2c; out[l * 12 + 5] = r2d;
    out[l * 12 + 6] = t.x; out[l * 12 + 7] = t.y; out[l * 12 + 8] = t2.x; out[l * 12 + 9] = t2.w;
    out[l * 12 + 10] = v; out[l * 12 + 11] = w;
}

template <int CTRL>
__device__ __forceinline__ double dpp(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// MODE 0: DPP row sum + cross-row by row_bcast; MODE 1: MFMA; WORK: independent FP64 FMAs per iteration
template <int MODE, int WORK>
__global__ void rate_kernel(double *out, int iters)
{
    double v = 1.0 + threadIdx.x * 1e-9, acc = 0.0;
    double f0 = 1.000001, f1 = 0.999999, f2 = 1.0000003, f3 = 0.9999997;
    for (int it = 0; it < iters; ++it) {
        double s;
        if (MODE == 0) {
            s = v + dpp<0xB1>(v);
            s += dpp<0x4E>(s);
            s += dpp<0x141>(s);
            s += dpp<0x140>(s);
            s += dpp<0x142>(s);            // row_bcast:15 (partial cross-row, as in the EM kernel)
        } else {
            const double r1 = __builtin_amdgcn_mfma_f64_4x4x4f64(v, 1.0, 0.0, 0, 0, 0);
            const double r2 = __builtin_amdgcn_mfma_f64_4x4x4f64(r1, 1.0, 0.0, 0, 0, 0);
            d4 z = {0, 0, 0, 0};
            const d4 t = __builtin_amdgcn_mfma_f64_16x16x4f64(1.0, r2, z, 0, 0, 0);
            s = t.x;
        }
#pragma unroll
        for (int w = 0; w < WORK; ++w) { f0 = fma(f0, f1, 1e-9); f1 = fma(f1, f2, 1e-9); f2 = fma(f2, f3, 1e-9); f3 = fma(f3, f0, 1e-9); }
        acc += s;
        v = fma(s, 1e-30, v) + f0 * 1e-30;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + f0 + f1 + f2 + f3;
}

template <int MODE, int WORK>
static float run_rate(double *d, int iters)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((rate_kernel<MODE, WORK>), dim3(256 * 4 * 4), dim3(256), 0, 0, d, 100);
    hipEventRecord(a);
    hipLaunchKernelGGL((rate_kernel<MODE, WORK>), dim3(256 * 4 * 4), dim3(256), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main()
{
    double *d; hipMalloc(&d, sizeof(double) * 256 * 4 * 4 * 256);
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, d);
    std::vector<double> h(64 * 12);
    hipMemcpy(h.data(), d, sizeof(double) * 64 * 12, hipMemcpyDeviceToHost);
    printf("lane  v      r1=mfma(v,1)  r2=mfma(r1,1)  r1b=mfma(1,v)  r2b=mfma(1,r1b)  r2c=mfma(r1b,1)  r2d=mfma(1,r1)   | 16x16x4(1,w).x .y  (w,1).x .w   w\n");
    for (int l = 0; l < 64; ++l) {
        if (l < 20 || l % 16 == 0)
            printf("%2d  %6.0f  %8.0f  %8.0f  %8.0f  %8.0f  %8.0f  %8.0f | %8.0f %8.0f %8.0f %8.0f  %6.0f\n", l, h[l * 12 + 10], h[l * 12 + 0],
                   h[l * 12 + 1], h[l * 12 + 2], h[l * 12 + 3], h[l * 12 + 4], h[l * 12 + 5], h[l * 12 + 6], h[l * 12 + 7], h[l * 12 + 8],
                   h[l * 12 + 9], h[l * 12 + 11]);
    }
    const int iters = 20000;
    printf("\n%d waves/SIMD x %d iterations of one 64-lane reduction (+ WORK x 4 independent FP64 FMAs):\n", 4, iters);
    printf("  DPP  work 0: %.3f ms   MFMA work 0: %.3f ms\n", run_rate<0, 0>(d, iters), run_rate<1, 0>(d, iters));
    printf("  DPP  work 4: %.3f ms   MFMA work 4: %.3f ms\n", run_rate<0, 4>(d, iters), run_rate<1, 4>(d, iters));
    printf("  DPP  work 8: %.3f ms   MFMA work 8: %.3f ms\n", run_rate<0, 8>(d, iters), run_rate<1, 8>(d, iters));
    printf("  FMAs only (work 8, no reduction would be ~ work-8 minus reduction)\n");
    return 0;
}
