// bvc_internal.h -- launcher prototypes shared by the translation units of libbvc (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bvc.h"

namespace bvc {

// Launch policy and one-time kernel setup of ONE context (bvc_ctx owns it; nothing here is process-wide, so
// contexts on the same or different devices are independent, include/bvc.h "Threading").
struct LaunchState {
    int n_cu = 256;
    int em_waves_per_cu = 0;   // 0 = default policy (em_kernel.hip); 1..32 resident EM wavefronts per CU
    int em_wpb = 4;            // waves per EM workgroup: 4, or 1 (A/B runs)
    int hist_split = 0;        // 0 = by tile shape; 1..64 workgroups sharing a site in the dense histogram pass
    int64_t host_chunk_bytes = (int64_t)1 << 29;   // BVC_PTR_HOST calls: bytes per array and staging chunk
    int em_streams = 0;        // overlap mode: side streams stage 2 alternates between: 0 = by call shape, 1..3
    int group_pipe = 1;        // any-order group histogram: issue the next chunk's loads before counting the current one
    int group_log2c = -1;      // any-order group histograms: -1 = as many LDS copies per histogram as fit 64 KiB; 0..5 = at most
                               // 2^n copies (A/B runs: fewer copies = smaller workgroups' LDS = more resident wavefronts)
    int group_big_lds = 1;     // any-order group histograms, 4 groups and more: 1 = one 1024-thread workgroup per CU with 256 slots x
                               // 16 / 8 / 4 copies per histogram (up to 9 / 18 / 36 histograms in 144 KiB of LDS; two-byte rows packed in
                               // registers), 0 = 512-thread workgroups of <= 64 KiB
    int group_h16 = 0;         // any-order group histograms on packed rows / packed-in-registers rows: 1 = 32 conflict-free copies of
                               // 16-bit counter pairs (hist_kernel.hip "H16"), 0 = 16 copies of 32-bit counters
    int em_engine = 0;         // stage 2: 0 = item engine (em_items.hip; em_kernel.hip takes the sites it leaves),
                               // 1 = one wavefront per site for every site (em_kernel.hip): A/B runs.  The two agree to
                               // rounding (1e-15 on AF), not bit for bit: a call's records never depend on the call's
                               // size or neighbours with either, but they depend on this choice
    int em_tiny_regions = 0;       // 1: regions whose sites all have <= 8 quality values per allele take the one-lane-per-allele
                               // kernel (binned qualities: 0.23 -> 0.17 ms per 4000 sites).  Off by default: that kernel adds in a
                               // different order, so a site's last bits would depend on whether its five region neighbours are binned too
    int em_prune = 1;              // item engine: 1 = a level does not run the subset without the deepest candidate when a bound on its
                                   // log-likelihood shows that it cannot be the level's first minimum (em_items.hip, site_decide); 0 = it always runs
    int dbg_levels = 0;            // BVC_DBG_LEVELS (timing only, records wrong): cut region_kernel short after a phase; 0 = run all
    mutable uint32_t em_epoch = 0; // stage-2 launches of this context so far (em_items.hip: the narrow launch tells the wide one)
    mutable uint64_t attr_done = 0;    // kernels whose dynamic-LDS attribute has been raised on this context's device
};

// Base-quality -> likelihood table, built on the HOST with the same libm exp() the CPU path uses
// (src/BaseType.cpp:13,15) and uploaded once per context:
//   a[q] = 1 - eps   likelihood of the observed base given the matching allele
//   e[q] = eps / 3   likelihood given any other allele
struct QualLut {
    double a[128];
    double e[128];
    double e_empty;            // e[128]: the "empty class place" of the item engine (em_items.hip), 1/4
    double log_e[128];         // log(e[q]) (libm, on the host): what a class of an allele outside the fitted subset adds to the
                               // log-likelihood per observation (em_items.hip, site_classes)
    double log_a[128];         // log(a[q]): what a class of THE allele of a one-allele model adds per observation (its marginal is a)
};

// Stage 1: dense pileup rows -> per-site class counts.  counts must be zeroed by the caller when split > 1.
hipError_t launch_hist_dense(LaunchState &st, hipStream_t stream, int64_t n_sites, int64_t n_samples, int64_t row_stride,
                             const int8_t *bases, const int8_t *quals, const uint8_t *group_of_sample,
                             int n_groups, uint32_t *counts, int split, int64_t *group_scratch = nullptr,
                             uint8_t *hist_of_sample = nullptr);
constexpr int kGroupScratchWords = BVC_MAX_GROUPS + 4;
// The label buffer of a group call (hist_of_sample): the labels clamped to 0..n_groups, then one flag per site ("redo with
// the general kernel", hist_dense_groups_slots_kernel).
inline size_t group_redo_offset(int64_t n_samples) { return ((size_t)n_samples + 255) & ~(size_t)255; }
inline size_t group_labels_bytes(int64_t n_samples, int64_t n_sites) { return group_redo_offset(n_samples) + (size_t)n_sites + 256; }
// Group mode needs group_scratch (kGroupScratchWords int64 of device memory) and hist_of_sample (group_labels_bytes(),
// 256-byte aligned): calls whose samples are ordered by group take the column-range kernel (decided on the device),
// the others index their histograms with the clamped labels written to hist_of_sample.
// Number of sample-range splits per site launch_hist_dense should use for this shape (1 = none).
int choose_hist_split(const LaunchState &st, int64_t n_sites, int64_t n_samples);

// Stage 1 on packed rows (one byte per sample: base << 6 | qual, qual <= 62; 0xFF = no observation).
hipError_t launch_hist_packed(LaunchState &st, hipStream_t stream, int64_t n_sites, int64_t n_samples, int64_t row_stride,
                              const uint8_t *packed, uint32_t *counts, int split);
// Group mode on packed rows: counts = [site][n_groups + 1][512]; scratch and labels as for launch_hist_dense.
hipError_t launch_hist_packed_groups(LaunchState &st, hipStream_t stream, int64_t n_sites, int64_t n_samples,
                                     int64_t row_stride, const uint8_t *packed, const uint8_t *group_of_sample, int n_groups,
                                     uint32_t *counts, int64_t *group_scratch, uint8_t *hist_of_sample);
hipError_t launch_pack_dense(hipStream_t stream, int64_t n_sites, int64_t n_samples, int64_t stride_in, const int8_t *bases,
                             const int8_t *quals, int64_t stride_out, uint8_t *packed, unsigned long long *bad);

hipError_t launch_stream_read(hipStream_t stream, const void *src, int64_t bytes, uint32_t *sink);

hipError_t launch_hist_csr(LaunchState &st, hipStream_t stream, int64_t n_sites, const int64_t *offsets,
                           const int8_t *bases, const int8_t *quals, uint32_t *counts);

// Stage 2: EM + LRT, one wavefront per (site, histogram).
//   hist_stride: uint32 elements between consecutive sites' histograms.
//   comb/n_comb: optional per-site candidate list (SetBase).
//   shared: the launch runs underneath a streaming histogram kernel (overlap mode) and keeps to a few wave slots.
//   scratch: em_items_scratch_bytes(n_sites) of device memory for the item engine (em_items.hip), or null: every
//            site then takes the one-wavefront-per-site kernels.
hipError_t launch_lrt(const LaunchState &st, hipStream_t stream, int64_t n_sites, const uint32_t *counts,
                      int64_t hist_stride, const int8_t *ref_base, double min_af, const QualLut *lut,
                      const int8_t *comb, const uint8_t *n_comb, bvc_site_result *results, bool shared = false,
                      int shared_waves_per_cu = 0,   // 0 = default cap when shared
                      void *scratch = nullptr);

// Item engine (em_items.hip): the EM fits of the call as work items, 8 or 16 to a wavefront.  Writes the records of
// the sites it takes and flags them in *taken_out ([n_sites] bytes inside scratch) for the kernels of em_kernel.hip.
// n_groups > 0: pseudo-site p = (site, group) on the per-group histograms [site][n_groups + 1][512].
size_t em_items_scratch_bytes(int64_t n_sites);
hipError_t launch_lrt_items(const LaunchState &st, hipStream_t stream, int64_t n_sites, int n_groups,
                            const uint32_t *counts, int64_t hist_stride, const int8_t *ref_base, double min_af,
                            const QualLut *lut, const int8_t *comb, const uint8_t *n_comb, bvc_site_result *results,
                            void *scratch, const uint8_t **taken_out, bool shared = false);

hipError_t launch_lrt_groups(const LaunchState &st, hipStream_t stream, int64_t n_sites, int n_groups,
                             const uint32_t *grp_counts, const int8_t *ref_base, double min_af, const QualLut *lut,
                             const bvc_site_result *overall, bvc_group_result *grp_results, bool shared = false,
                             int shared_waves_per_cu = 0, void *scratch = nullptr);
// scratch: em_group_scratch_bytes(n_sites, n_groups) for the item engine on the (site, group) pseudo-sites, or null
size_t em_group_scratch_bytes(int64_t n_sites, int n_groups);
// EM wavefronts per CU the stage-2 launches of group calls keep in flight underneath a long histogram pass, summed
// over the stage-2 streams in use (em_kernel.hip, em_grid_cap).
constexpr int kGroupSharedWavesPerCu = 8;

// Sum the per-group histograms (+ the "no group" one) into the overall histogram of each site.
hipError_t launch_sum_groups(hipStream_t stream, int64_t n_sites, int n_hist, const uint32_t *grp_counts,
                             uint32_t *counts);

hipError_t launch_synth_dense(hipStream_t stream, uint64_t seed, int64_t site0, int64_t n_sites,
                              int64_t n_samples, int64_t row_stride, uint32_t cov_thr16,
                              int8_t *bases, int8_t *quals, int8_t *ref_base);

// ---- pileup_kernel.hip: temp-batch pileup text -> ragged columns; per-group histograms of ragged columns -----------------
// Device buffers of one tile between bvc_pileup_begin and bvc_pileup_finish (all owned by the context).
struct PileupTile {
    const uint8_t *text = nullptr;
    const uint32_t *line_start = nullptr;    // [n_batches][line_stride]
    const int32_t *sample0 = nullptr, *n_in_batch = nullptr;
    int32_t n_batches = 0, n_pos = 0;        // n_pos: the tile's positions (an upper bound while n_pos_dev decides)
    int32_t line_stride = 0;                 // elements of line_start per batch
    const int32_t *n_pos_dev = nullptr;      // tiles inflated on the device: the positions, decided there
    int64_t n_lines_cap = 0;                 // lines the per-line arrays are laid out for
    uint32_t *line_words = nullptr;          // 4 arrays of n_lines_cap words: entries, observations, last base token, inherited indels
    uint32_t *status = nullptr;              // [0] irregular lines [1] indel entries [2] indel records written [3] carry out
                                             // [4] bytes of indel tokens [5] bytes of indel text gathered
    int64_t *entry_off = nullptr, *obs_off = nullptr, *totals = nullptr;    // [n_pos + 1], [n_pos + 1], [2]
    bvc_pileup_entry *entries = nullptr;
    int32_t *samples = nullptr, *obs_sample = nullptr;
    int8_t *obs_base = nullptr, *obs_qual = nullptr;
    int32_t *tally = nullptr;                // [n_pos][32]
    bvc_pileup_indel *indels = nullptr;
    uint32_t indel_cap = 0;
};
// A batch's region of the text buffer of a tile inflated on the device: [start, start + len) = left_len bytes the tile before left
// (at left_src of the other text buffer) + the output of the batch's new blocks.
struct bvc_pileup_region { uint32_t start, len, left_src, left_len; };
hipError_t launch_region_carry(hipStream_t stream, const uint8_t *old_text, uint8_t *text, const bvc_pileup_region *regions, int32_t n_batches);
// newlines of every region -> lines[b], the tile's positions (*P.n_pos_dev = min(lines, max_pos)), line_start of those positions
hipError_t launch_region_index(hipStream_t stream, const PileupTile &P, const bvc_pileup_region *regions, const uint32_t *seg_base,
                               int64_t n_segments, uint32_t *seg_nl, int32_t *lines, int32_t max_pos);
hipError_t launch_region_ends(hipStream_t stream, const PileupTile &P, uint32_t *ends);
// the text of the indel tokens gathered into dst (records' text_off become offsets into it); *used = bytes
hipError_t launch_indel_text(hipStream_t stream, const PileupTile &P, uint8_t *dst, uint32_t dst_cap, uint32_t *used);
// the entries of the called positions, compacted: called_off [n_pos + 1] (exclusive prefix; the last word is the total)
hipError_t launch_called_scan(hipStream_t stream, const PileupTile &P, const bvc_site_result *results, int64_t *called_off);
hipError_t launch_called_gather(hipStream_t stream, const PileupTile &P, const int64_t *called_off, bvc_pileup_entry *out_entries,
                                int32_t *out_samples);
// count pass + prefix sums (fills line_words, entry_off, obs_off, totals, status[0..1])
hipError_t launch_pileup_count(hipStream_t stream, const PileupTile &P);
// write pass + the entries that inherit across lines (status[2] and tally must be zero; leaves the carry in status[3])
hipError_t launch_pileup_write(hipStream_t stream, const PileupTile &P, uint32_t carry_in);
// counts = [site][n_groups + 1][512] from ragged observations with their sample indices
hipError_t launch_hist_csr_groups(LaunchState &st, hipStream_t stream, int64_t n_sites, const int64_t *offsets, const int8_t *bases,
                                  const int8_t *quals, const int32_t *sample_of_obs, const uint8_t *group_of_sample, int64_t n_samples,
                                  int n_groups, uint32_t *counts);

// inflate_kernel.hip: raw deflate of whole BGZF blocks, one wavefront per block; status[i] != 0: block i is not valid deflate of isize bytes
hipError_t launch_inflate(hipStream_t stream, const uint8_t *comp, const bvc_bgzf_block *blocks, int64_t n_blocks, uint8_t *out, uint32_t *status);

#ifdef BVC_CHECK_LDS
hipError_t debug_read_inflate(uint32_t *out8, bool reset);
hipError_t debug_read_pileup(uint32_t *out8, bool reset);
// diagnostic builds: each translation unit's violation record (bvc_device.h)
hipError_t debug_read_hist(uint32_t *out8, bool reset);
hipError_t debug_read_wave_engine(uint32_t *out8, bool reset);
hipError_t debug_read_items(uint32_t *out8, bool reset);
#endif

}  // namespace bvc
