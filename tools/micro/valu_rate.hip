// valu_rate.hip -- issue rate of the instructions the EM pass is made of (gfx950 microbenchmark, not product code).
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/micro/valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int kIter = 4096;

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(double *out, double a, double b)
{
    double x0 = a + threadIdx.x, x1 = a * 2 + threadIdx.x, x2 = a * 3, x3 = a * 4, x4 = a * 5, x5 = a * 6, x6 = a * 7, x7 = a * 8;
    for (int i = 0; i < kIter; ++i) {
        if (OP == 0) {          // 8 independent v_fma_f64
            x0 = fma(x0, b, a); x1 = fma(x1, b, a); x2 = fma(x2, b, a); x3 = fma(x3, b, a);
            x4 = fma(x4, b, a); x5 = fma(x5, b, a); x6 = fma(x6, b, a); x7 = fma(x7, b, a);
        } else if (OP == 1) {   // 8 independent v_add_f64
            x0 += b; x1 += b; x2 += b; x3 += b; x4 += b; x5 += b; x6 += b; x7 += b;
        } else if (OP == 2) {   // 8 independent v_mul_f64
            x0 *= b; x1 *= b; x2 *= b; x3 *= b; x4 *= b; x5 *= b; x6 *= b; x7 *= b;
        } else if (OP == 3) {   // one dependent chain of v_fma_f64 (latency at 1 wave, rate at many)
            x0 = fma(x0, b, a); x0 = fma(x0, b, a); x0 = fma(x0, b, a); x0 = fma(x0, b, a);
            x0 = fma(x0, b, a); x0 = fma(x0, b, a); x0 = fma(x0, b, a); x0 = fma(x0, b, a);
        } else if (OP == 4) {   // 8 fp32 fma for comparison
            float f0 = (float)x0, f1 = (float)x1, f2 = (float)x2, f3 = (float)x3;
            (void)f0; (void)f1; (void)f2; (void)f3;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int OP>
__global__ __launch_bounds__(256) void rate32_kernel(float *out, float a, float b)
{
    float x0 = a + threadIdx.x, x1 = a * 2, x2 = a * 3, x3 = a * 4, x4 = a * 5, x5 = a * 6, x6 = a * 7, x7 = a * 8;
    for (int i = 0; i < kIter; ++i) {
        if (OP == 0) {
            x0 = fmaf(x0, b, a); x1 = fmaf(x1, b, a); x2 = fmaf(x2, b, a); x3 = fmaf(x3, b, a);
            x4 = fmaf(x4, b, a); x5 = fmaf(x5, b, a); x6 = fmaf(x6, b, a); x7 = fmaf(x7, b, a);
        } else {                // 8 DPP moves + adds (row_ror:8-like quad_perm)
            x0 += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x0), 0xB1, 0xF, 0xF, true));
            x1 += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x1), 0xB1, 0xF, 0xF, true));
            x2 += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x2), 0xB1, 0xF, 0xF, true));
            x3 += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x3), 0xB1, 0xF, 0xF, true));
            x4 += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x4), 0xB1, 0xF, 0xF, true));
            x5 += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x5), 0xB1, 0xF, 0xF, true));
            x6 += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x6), 0xB1, 0xF, 0xF, true));
            x7 += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x7), 0xB1, 0xF, 0xF, true));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <typename F>
static double time_ms(F launch)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int n_cu = p.multiProcessorCount;
    const double clk = p.clockRate * 1e3;    // Hz
    double *out; hipMalloc(&out, sizeof(double) * 256 * n_cu * 16);
    printf("CUs %d clock %.0f MHz\n", n_cu, clk / 1e6);
    const char *names[] = {"v_fma_f64 x8 indep", "v_add_f64 x8 indep", "v_mul_f64 x8 indep", "v_fma_f64 chain of 8"};
    for (int wpc : {4, 8, 16, 32}) {        // waves per CU = blocks*4 per CU
        const int blocks = n_cu * wpc / 4;
        double ms[4];
        ms[0] = time_ms([&] { hipLaunchKernelGGL(rate_kernel<0>, dim3(blocks), dim3(256), 0, 0, out, 1.0, 0.999); });
        ms[1] = time_ms([&] { hipLaunchKernelGGL(rate_kernel<1>, dim3(blocks), dim3(256), 0, 0, out, 1.0, 0.999); });
        ms[2] = time_ms([&] { hipLaunchKernelGGL(rate_kernel<2>, dim3(blocks), dim3(256), 0, 0, out, 1.0, 0.999); });
        ms[3] = time_ms([&] { hipLaunchKernelGGL(rate_kernel<3>, dim3(blocks), dim3(256), 0, 0, out, 1.0, 0.999); });
        for (int k = 0; k < 4; ++k) {
            const double wave_instr_per_simd = (double)kIter * 8 * wpc / 4;      // waves per SIMD * instrs per wave
            const double cyc = ms[k] * 1e-3 * clk;
            printf("waves/CU %2d  %-22s %.3f ms  %.2f cycles per wave-instruction per SIMD\n", wpc, names[k], ms[k], cyc / wave_instr_per_simd);
        }
        float *o32 = reinterpret_cast<float *>(out);
        const double m0 = time_ms([&] { hipLaunchKernelGGL(rate32_kernel<0>, dim3(blocks), dim3(256), 0, 0, o32, 1.0f, 0.999f); });
        const double m1 = time_ms([&] { hipLaunchKernelGGL(rate32_kernel<1>, dim3(blocks), dim3(256), 0, 0, o32, 1.0f, 0.999f); });
        printf("waves/CU %2d  %-22s %.3f ms  %.2f cycles per wave-instruction per SIMD\n", wpc, "v_fma_f32 x8 indep", m0, m0 * 1e-3 * clk / ((double)kIter * 8 * wpc / 4));
        printf("waves/CU %2d  %-22s %.3f ms  %.2f cycles per (dpp mov + add) pair per SIMD\n", wpc, "dpp+add f32 x8", m1, m1 * 1e-3 * clk / ((double)kIter * 8 * wpc / 4));
    }
    return 0;
}
