// synth_kernel.hip -- counter-based synthetic pileup generator on the device.
// Integer arithmetic only, so any site can be regenerated bit for bit on the CPU by the test oracle
// without ever storing or moving the tile.  Not part of the reference: it exists because the
// benchmark configs (1e5 sites x 1e6 samples = 200 GB) can only be produced where they are consumed.
#include "bvc_device.h"
#include "bvc_internal.h"

namespace bvc {
namespace {

// The threshold tables are compiled into the code object as constant-address-space data (no runtime upload, so no
// per-device "done once" state): the include below defines SYNTH_AF_THR / SYNTH_ERR_THR in device constant memory.
#define SYNTH_TABLE __device__ __constant__ const
#include "synth_tables.inc"

__device__ __host__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}

struct SiteParams {
    uint64_t hs2;
    uint32_t ref, alt, alt2, thr1, thr2;
};

__device__ __forceinline__ SiteParams site_params(uint64_t seed, int64_t site)
{
    const uint64_t hs = mix64(seed * 0x9E3779B97F4A7C15ULL + (uint64_t)site + 0x632BE59BD9B4E019ULL);
    SiteParams p;
    p.ref = (uint32_t)(hs & 3);
    p.alt = (p.ref + 1 + (uint32_t)((hs >> 2) & 0xFFFF) % 3) & 3;
    p.alt2 = (p.alt == ((p.ref + 1) & 3)) ? ((p.ref + 2) & 3) : ((p.ref + 1) & 3);
    const bool poly = ((hs >> 20) & 0xFFFF) % 100 < 20;
    const bool second = ((hs >> 36) & 0xFFFF) % 100 < 2;
    p.thr1 = poly ? SYNTH_AF_THR[(hs >> 52) & 63] : 0u;
    p.thr2 = (poly && second) ? p.thr1 / 4 : 0u;
    p.hs2 = mix64(hs ^ 0xD1B54A32D192ED03ULL);
    return p;
}

// returns base | qual << 8 (base = 0xFF, qual = 0 when uncovered)
__device__ __forceinline__ uint32_t draw(const SiteParams &p, int64_t i, uint32_t cov_thr16)
{
    const uint64_t h1 = mix64(p.hs2 + (uint64_t)i * 0x9E3779B97F4A7C15ULL);
    const uint64_t h2 = mix64(h1 + 0x9E3779B97F4A7C15ULL);
    const uint32_t r_allele = (uint32_t)h1;
    const uint32_t q = 10u + (uint32_t)((((h1 >> 32) & 0xFFFF) * 31) >> 16);
    const uint32_t r_err = (uint32_t)h2;
    const uint32_t r_sub = (uint32_t)((((h2 >> 32) & 0xFFFF) * 3) >> 16);
    const uint32_t r_cov = (uint32_t)(h2 >> 48);
    uint32_t b = p.ref;
    if (r_allele < p.thr1) b = p.alt;
    else if (r_allele - p.thr1 < p.thr2) b = p.alt2;
    if (r_err < SYNTH_ERR_THR[q]) b = (b + 1 + r_sub) & 3;
    return (r_cov < cov_thr16) ? (b | (q << 8)) : 0xFFu;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// grid.y = site, grid.x strides over 16-sample chunks; 16-byte stores when the rows allow it.
__global__ __launch_bounds__(256) void synth_dense_kernel(uint64_t seed, int64_t site0, int64_t n_samples,
                                                          int64_t row_stride, uint32_t cov_thr16,
                                                          int8_t *__restrict__ bases, int8_t *__restrict__ quals,
                                                          int8_t *__restrict__ ref_base, int aligned)
{
    BVC_POISON_LDS();
    const int64_t s = blockIdx.y;
    const SiteParams p = site_params(seed, site0 + s);
    if (blockIdx.x == 0 && threadIdx.x == 0) ref_base[s] = (int8_t)p.ref;
    int8_t *brow = bases + s * row_stride;
    int8_t *qrow = quals + s * row_stride;
    const int64_t n16 = aligned ? (n_samples >> 4) : 0;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n16; c += (int64_t)gridDim.x * blockDim.x) {
        uint32_t bw[4], qw[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            bw[w] = 0; qw[w] = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t v = draw(p, c * 16 + w * 4 + k, cov_thr16);
                bw[w] |= (v & 0xFFu) << (8 * k);
                qw[w] |= (v >> 8) << (8 * k);
            }
        }
        reinterpret_cast<u32x4 *>(brow)[c] = u32x4{bw[0], bw[1], bw[2], bw[3]};
        reinterpret_cast<u32x4 *>(qrow)[c] = u32x4{qw[0], qw[1], qw[2], qw[3]};
    }
    for (int64_t i = (n16 << 4) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_samples;
         i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t v = draw(p, i, cov_thr16);
        brow[i] = (int8_t)(v & 0xFFu);
        qrow[i] = (int8_t)(v >> 8);
    }
}

}  // namespace

hipError_t launch_synth_dense(hipStream_t stream, uint64_t seed, int64_t site0, int64_t n_sites,
                              int64_t n_samples, int64_t row_stride, uint32_t cov_thr16,
                              int8_t *bases, int8_t *quals, int8_t *ref_base)
{
    if (n_sites <= 0 || n_samples < 0) return hipSuccess;
    const int aligned = ((reinterpret_cast<uintptr_t>(bases) | reinterpret_cast<uintptr_t>(quals)) & 15u) == 0 &&
                        (row_stride & 15) == 0;
    int64_t gx = ((n_samples >> 4) + 255) / 256;
    if (gx < 1) gx = 1;
    if (gx > 64) gx = 64;
    for (int64_t s0 = 0; s0 < n_sites; s0 += 65535) {            // grid.y limit
        const int64_t ns = (n_sites - s0 < 65535) ? n_sites - s0 : 65535;
        hipLaunchKernelGGL(synth_dense_kernel, dim3((unsigned)gx, (unsigned)ns), dim3(256), 0, stream, seed,
                           site0 + s0, n_samples, row_stride, cov_thr16, bases + s0 * row_stride,
                           quals + s0 * row_stride, ref_base + s0, aligned);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace bvc
