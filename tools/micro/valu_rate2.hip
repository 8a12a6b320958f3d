// valu_rate2.hip -- issue cost of the instructions the item-engine EM pass is made of (gfx950 microbenchmark, not
// product code): v_rcp_f64, f64<->f32 conversions, v_rcp_f32, the gfx950 lane swaps, DPP moves; 8 independent streams.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate2 tools/micro/valu_rate2.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int kIter = 2048;

template <int OP>
__global__ __launch_bounds__(256) void k(double *out, double a, double b)
{
    double x[8];
    float f[8];
    int w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = a * (i + 1) + threadIdx.x; f[i] = (float)x[i]; w[i] = (int)threadIdx.x * (i + 3); }
    for (int it = 0; it < kIter; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(b), "v"(a));
            if (OP == 1) asm volatile("v_rcp_f64 %0, %0" : "+v"(x[i]));
            if (OP == 2) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(x[i]));
            if (OP == 3) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(x[i]) : "v"(f[i]));
            if (OP == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
            if (OP == 5) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(w[i]), "+v"(w[(i + 1) & 7]));
            if (OP == 6) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(w[i]), "+v"(w[(i + 1) & 7]));
            if (OP == 7) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(w[i]) : "v"(w[(i + 1) & 7]));
            if (OP == 8) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(x[i]) : "v"(w[i]));
            if (OP == 9) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[i]) : "v"(b));
            if (OP == 10) asm volatile("v_log_f32 %0, %0" : "+v"(f[i]));
            if (OP == 11) asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(x[i]));
            if (OP == 12) asm volatile("v_readlane_b32 s20, %0, 5" : : "v"(w[i]) : "s20");
            if (OP == 13) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(w[i]) : "v"(w[(i + 1) & 7]) : );
            if (OP == 14) asm volatile("v_rsq_f64 %0, %0" : "+v"(x[i]));
            if (OP == 15) asm volatile("v_add_f64 %0, %0, |%1|" : "+v"(x[i]) : "v"(b));
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i] + f[i] + w[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
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

template <int OP>
static void run(const char *name, double *out, int n_cu, double clk)
{
    for (int wpc : {4, 8, 16}) {
        const int blocks = n_cu * wpc / 4;
        const double ms = time_ms([&] { hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0, 0.999); });
        printf("%-22s waves/CU %2d  %.3f ms  %.2f cycles per wave-instruction per SIMD\n", name, wpc, ms,
               ms * 1e-3 * clk / ((double)kIter * 8 * wpc / 4));
    }
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int n_cu = p.multiProcessorCount;
    const double clk = p.clockRate * 1e3;
    double *out; hipMalloc(&out, sizeof(double) * 256 * n_cu * 16);
    printf("CUs %d clock %.0f MHz (cycle counts assume the nominal clock)\n", n_cu, clk / 1e6);
    run<0>("v_fma_f64", out, n_cu, clk);
    run<9>("v_mul_f64", out, n_cu, clk);
    run<15>("v_add_f64 |abs|", out, n_cu, clk);
    run<1>("v_rcp_f64", out, n_cu, clk);
    run<14>("v_rsq_f64", out, n_cu, clk);
    run<11>("v_frexp_mant_f64", out, n_cu, clk);
    run<2>("v_cvt_f32_f64", out, n_cu, clk);
    run<3>("v_cvt_f64_f32", out, n_cu, clk);
    run<8>("v_cvt_f64_u32", out, n_cu, clk);
    run<4>("v_rcp_f32", out, n_cu, clk);
    run<10>("v_log_f32", out, n_cu, clk);
    run<5>("v_permlane32_swap", out, n_cu, clk);
    run<6>("v_permlane16_swap", out, n_cu, clk);
    run<7>("v_mov_b32_dpp", out, n_cu, clk);
    run<12>("v_readlane_b32", out, n_cu, clk);
    run<13>("v_cndmask_b32", out, n_cu, clk);
    return 0;
}
