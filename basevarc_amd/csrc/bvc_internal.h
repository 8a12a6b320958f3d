// bvc_internal.h -- launcher prototypes shared by the translation units of libbvc (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bvc.h"

namespace bvc {

// Per-device one-time setup flags (function attributes, constant tables).  A process may drive several
// devices from several threads (the host program does: thread i -> device i mod gpus), so "done once" is
// tracked per device and the setup itself is idempotent.
constexpr int kMaxDevices = 64;
inline int current_device_slot()
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) d = 0;
    return d;
}

// Base-quality -> likelihood table, built on the HOST with the same libm exp() the CPU path uses
// (src/BaseType.cpp:13,15) and uploaded once per context:
//   a[q] = 1 - eps   likelihood of the observed base given the matching allele
//   e[q] = eps / 3   likelihood given any other allele
struct QualLut {
    double a[128];
    double e[128];
};

// Stage 1: dense pileup rows -> per-site class counts.  counts must be zeroed by the caller when split > 1.
hipError_t launch_hist_dense(hipStream_t stream, int64_t n_sites, int64_t n_samples, int64_t row_stride,
                             const int8_t *bases, const int8_t *quals, const uint8_t *group_of_sample,
                             int n_groups, uint32_t *counts, int split, int64_t *group_scratch = nullptr);
// group_scratch: BVC_MAX_GROUPS + 4 int64 of device memory; when given, calls whose samples are ordered by group
// take the column-range kernel (decided on the device).
// Number of sample-range splits per site launch_hist_dense should use for this shape (1 = none).
int choose_hist_split(int64_t n_sites, int64_t n_samples, int n_cu);

hipError_t launch_stream_read(hipStream_t stream, const void *src, int64_t bytes, uint32_t *sink);

hipError_t launch_hist_csr(hipStream_t stream, int64_t n_sites, const int64_t *offsets,
                           const int8_t *bases, const int8_t *quals, uint32_t *counts);

// Stage 2: EM + LRT, one wavefront per (site, histogram).
//   hist_stride: uint32 elements between consecutive sites' histograms; hist_sub: which 512-block inside it.
//   comb/n_comb: optional per-site candidate list (SetBase); comb_from: optional results array whose
//   {ref}+alt_bases define the candidates and whose `called` gates the run (group mode).
hipError_t launch_lrt(hipStream_t stream, int64_t n_sites, const uint32_t *counts, int64_t hist_stride,
                      const int8_t *ref_base, double min_af, const QualLut *lut,
                      const int8_t *comb, const uint8_t *n_comb, bvc_site_result *results, bool shared = false,
                      int64_t depth_hint = 0,    // samples per site when the caller knows it (layout choice only)
                      int shared_waves_per_cu = 0);   // 0 = default cap when shared

// rows_mode: -1 auto, 0 one site per wave, 1 four sites per wave; waves_per_cu: 0 = default policy
void set_em_tuning(int rows_mode, int waves_per_cu);

hipError_t launch_lrt_groups(hipStream_t stream, int64_t n_sites, int n_groups, const uint32_t *grp_counts,
                             const int8_t *ref_base, double min_af, const QualLut *lut,
                             const bvc_site_result *overall, bvc_group_result *grp_results, bool shared = false);

// Sum the per-group histograms (+ the "no group" one) into the overall histogram of each site.
hipError_t launch_sum_groups(hipStream_t stream, int64_t n_sites, int n_hist, const uint32_t *grp_counts,
                             uint32_t *counts);

hipError_t launch_synth_dense(hipStream_t stream, uint64_t seed, int64_t site0, int64_t n_sites,
                              int64_t n_samples, int64_t row_stride, uint32_t cov_thr16,
                              int8_t *bases, int8_t *quals, int8_t *ref_base);

}  // namespace bvc
