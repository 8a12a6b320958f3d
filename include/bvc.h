/*
 * bvc.h -- C ABI of libbvc, the MI355X (gfx950) implementation of BaseVarC's per-site
 * basetype hot path.
 *
 * What it replaces in the reference (paths under /root/reference):
 *   - class BaseType: constructor, SetBase, LRT and the public result fields
 *       src/BaseType.h:59-74, src/BaseType.cpp:5-139, 237-255
 *   - EM / singleEM / delta_bylog and chisf
 *       src/Algorithm.cpp:3-7, 69-130
 *   - the per-site call sites in bt_f
 *       src/BaseVarC.cpp:612-615 (overall call), :617-661 (per-group calls)
 *
 * The reference has no FFI: BaseType is a C++ object built and consumed once per site on the caller's
 * stack.  Launching device work per site is not viable, so the ABI is batched: the caller hands over a
 * TILE of sites in site-major layout and receives one fixed-size record per site holding exactly the
 * fields bt_f and WriteVcf read from a BaseType (var_qual, depth_total, alt_bases, depth, af_lrt;
 * src/BaseType.cpp:145-147, 200-212, 221, 226) plus a few diagnostics used for parity pinning.
 *
 * Conventions
 *   - plain C, no exceptions cross the boundary; every entry point returns BVC_OK (0) or a negative code
 *     and bvc_last_error(ctx) gives the text.  "No call" (LRT() == false) is called = 0, not an error.
 *   - the caller owns all input/output buffers; the library owns only device scratch inside the context.
 *   - one context = one device + one HIP stream; contexts are independent (one per host thread, as
 *     bt_f runs on T std::threads in the reference, src/BaseVarC.cpp:263-266); the library keeps no
 *     process-wide mutable state.
 *   - there is NO CPU fallback: without a usable gfx950 device bvc_create fails.
 */
#ifndef BVC_H
#define BVC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BVC_OK               0
#define BVC_ERR_ARG         -1   /* bad argument (null pointer, negative size, base_comb out of range ...) */
#define BVC_ERR_DEVICE      -2   /* HIP runtime error; text in bvc_last_error */
#define BVC_ERR_NO_DEVICE   -3   /* no gfx950 device / device index out of range */
#define BVC_ERR_ALLOC       -4   /* device or host allocation failed */
#define BVC_ERR_DATA        -5   /* bvc_pileup_begin_bgzf: a BGZF block is not valid deflate of its ISIZE bytes */
#define BVC_PILEUP_IRREGULAR 1   /* bvc_pileup_begin only, not an error: a line of the tile is not of the shape the reference's
                                    writer produces; nothing was computed, parse this tile with the reference's own rules */

/* flags for the compute entry points */
#define BVC_PTR_HOST    0u       /* every data pointer is host memory; the call is synchronous */
#define BVC_PTR_DEVICE  1u       /* every data pointer is device memory; the call is asynchronous on the
                                    context's stream (bvc_synchronize or a stream sync completes it) */

#define BVC_NCLASS 512           /* (base, qual) classes per site: base * 128 + qual */
#define BVC_MAX_GROUPS 32

typedef struct bvc_ctx bvc_ctx;

/* One record per site: what a BaseType holds after LRT() (src/BaseType.h:70-74). 120 bytes. */
typedef struct bvc_site_result {
    double  var_qual;       /* BaseType::var_qual */
    double  chi;            /* chi_sqrt_t when LRT() left its loop (src/BaseType.cpp:101) -- diagnostic */
    double  depth_total;    /* BaseType::depth_total */
    double  af[3];          /* af_lrt[alt_base[i]], i < n_alt */
    double  lr_alt;         /* lr_alt_t at exit -- diagnostic */
    double  base_frq[4];    /* fitted frequencies of the accepted model, indexed by base -- diagnostic */
    int32_t depth[4];       /* BaseType::depth[A,C,G,T] */
    int32_t n_passes;       /* number of E+M passes run (singleEM calls) -- diagnostic; see "em_prune" */
    int8_t  alt_base[3];    /* BaseType::alt_bases, in the reference's order */
    uint8_t n_alt;
    uint8_t called;         /* return value of LRT() */
    uint8_t n_kept;         /* size of the accepted model */
    int8_t  kept[4];        /* bases of the accepted model (`bases` at exit) */
    uint8_t status;         /* 0 ok; 1 = the reference's behaviour is undefined for this input
                               (bp[0] / min_element on an empty vector, only reachable with min_af <= 0) */
    uint8_t n_fits;         /* EM() calls run -- diagnostic; see "em_prune" */
} bvc_site_result;

/* Per (site, group) record for the caller's --group loop (src/BaseVarC.cpp:617-661). 48 bytes. */
typedef struct bvc_group_result {
    double  af[3];          /* <group>_AF for overall alt i: gr_bt.af_lrt[alt] or 0 when absent (:646-652) */
    int32_t depth[4];       /* na:nc:ng:nt of the group's covered samples (:640) */
    uint8_t ran;            /* 1 when the group's BaseType was built and LRT() run (:641-644) */
    uint8_t present;        /* bit i set: overall alt i is among the group's alt_bases (af_lrt.count(b), :647);
                               a clear bit is the literal "0" of :650, not an estimated frequency of zero */
    uint8_t pad[6];
} bvc_group_result;

typedef struct bvc_profile {
    double  hist_ms;        /* summed HIP-event time of the pileup->histogram kernel since the last reset */
    double  em_ms;          /* summed HIP-event time of the EM/LRT kernel */
    int64_t hist_launches;
    int64_t em_launches;
    int64_t sites;
} bvc_profile;

/* ---- context -------------------------------------------------------------------------------------- */
const char *bvc_version(void);
/* Number of usable gfx950 devices (0 when there is none or the HIP runtime cannot start). */
int  bvc_device_count(void);
int  bvc_create(bvc_ctx **out, int device);
void bvc_destroy(bvc_ctx *ctx);

/*
 * Additive: page-locked host memory for the buffers a caller hands to the calls that take HOST pointers.  A transfer from or to a
 * buffer that lies inside such an allocation goes straight over the link; any other host buffer goes through the context's own
 * page-locked bounce buffer (one memcpy of the bytes more).  The host program assembles the compressed blocks of a tile there
 * (bvc_pileup_begin_bgzf: a fifth of a tile's bytes, and the largest transfer of the feed).  Any thread, any device; NULL when the
 * allocation fails.  bvc_host_free(NULL) is a no-op.
 */
void *bvc_host_alloc(size_t bytes);
void bvc_host_free(void *p);
const char *bvc_last_error(const bvc_ctx *ctx);
/* Run on the caller's HIP stream (hipStream_t passed as void*; NULL = the device's default stream).  A new context works on a
 * stream of its own (a blocking stream: ordered against the device's default stream, concurrent with other contexts' streams). */
int  bvc_set_stream(bvc_ctx *ctx, void *hip_stream);
int  bvc_synchronize(bvc_ctx *ctx);
/*
 * Overlap mode (off by default).  With it on, every entry point that takes device pointers (bvc_lrt_dense,
 * bvc_lrt_dense_groups, bvc_lrt_csr[_comb]) runs its second stage (EM/LRT, FP64-bound) on internal side streams,
 * so that it executes underneath the first stage (histogram, HBM-bound) of the NEXT call.  Results of a call are
 * then complete only after bvc_join (makes the context's stream wait for all side work) or bvc_synchronize; the
 * caller must keep the ref_base and results buffers of a call alive and untouched until then, and give calls
 * that may be in flight together DIFFERENT results buffers: the second stages of consecutive calls may run
 * side by side (two side streams), so which of two writers of one buffer comes last is not defined.
 */
int  bvc_set_overlap(bvc_ctx *ctx, int on);
int  bvc_join(bvc_ctx *ctx);
/* HIP-event timing of each kernel (adds two event records per launch). */
int  bvc_set_profiling(bvc_ctx *ctx, int on);
int  bvc_get_profile(bvc_ctx *ctx, bvc_profile *out, int reset);

/* ---- the hot path --------------------------------------------------------------------------------- */
/*
 * Dense tile: bases[s * row_stride + i], quals[...] for site s < n_sites, sample i < n_samples.
 * A sample is covered iff its base byte is 0..3 (A,C,G,T) and its qual byte is 0..127; anything else
 * (the caller's "no read / N / indel" cases, src/BaseVarC.cpp:427, 551-559) is skipped.
 * Equivalent per site to:  BaseType bt(bases, quals, ref_base[s], min_af); bt.LRT();
 * (src/BaseVarC.cpp:612-613).  min_af is the caller's value (src/BaseVarC.cpp:541-543).
 * ref_base[s] is 0..3 (A,C,G,T).  Any other value (the caller's -1 for a non-ACGT reference letter) means "no
 * base equals the reference": every kept base is then an ALT, as `b != ref_base` decides in src/BaseType.cpp:112,
 * and in group mode it contributes no candidate to {ref} + alt_bases (its depth is 0 in the reference, so the
 * min_af filter of src/BaseType.cpp:79 drops it).  The reference's caller never produces such a site
 * (src/BaseVarC.cpp:199-203).
 */
int bvc_lrt_dense(bvc_ctx *ctx, int64_t n_sites, int64_t n_samples, int64_t row_stride,
                  const int8_t *bases, const int8_t *quals, const int8_t *ref_base,
                  double min_af, bvc_site_result *results, uint32_t flags);

/*
 * Dense tile with population groups: additionally, for every group g < n_groups, the depth counts and --
 * when the overall call succeeded -- BaseType gr(bases_g, quals_g, ref, min_af);
 * gr.SetBase({ref} + alt_bases); gr.LRT()  (src/BaseVarC.cpp:617-661).
 * group_of_sample[i] >= n_groups means "in no group" (src/BaseVarC.cpp:352-356).
 * grp_results is [n_sites][n_groups].
 * Any column order gives the same records; when group_of_sample is non-decreasing (each group a contiguous run
 * of columns, ungrouped samples last) the call takes a faster histogram kernel.
 */
int bvc_lrt_dense_groups(bvc_ctx *ctx, int64_t n_sites, int64_t n_samples, int64_t row_stride,
                         const int8_t *bases, const int8_t *quals, const int8_t *ref_base,
                         double min_af, const uint8_t *group_of_sample, int32_t n_groups,
                         bvc_site_result *results, bvc_group_result *grp_results, uint32_t flags);

/*
 * Ragged (CSR) pileup: site s owns bases[offsets[s] .. offsets[s+1]) -- exactly the vectors bt_f builds
 * (src/BaseVarC.cpp:550-559).  Every element must be a covered observation or it is skipped as above.
 */
int bvc_lrt_csr(bvc_ctx *ctx, int64_t n_sites, const int64_t *offsets,
                const int8_t *bases, const int8_t *quals, const int8_t *ref_base,
                double min_af, bvc_site_result *results, uint32_t flags);
/*
 * The same with BaseType::SetBase (src/BaseType.h:67): base_comb[s * 4 + c], c < n_comb[s] <= 4, are the candidate
 * bases of site s in SetBase order; both NULL means the default {A,C,G,T} (src/BaseType.h:79).  This is the whole
 * of  BaseType bt(bases, quals, ref, min_af); bt.SetBase(v); bt.LRT();  in one call (src/BaseVarC.cpp:642-644).
 */
int bvc_lrt_csr_comb(bvc_ctx *ctx, int64_t n_sites, const int64_t *offsets,
                     const int8_t *bases, const int8_t *quals, const int8_t *ref_base,
                     double min_af, const int8_t *base_comb, const uint8_t *n_comb,
                     bvc_site_result *results, uint32_t flags);

/*
 * Additive: the ragged form at ONE byte per observation (the packed byte of bvc_lrt_dense_packed below:
 * base << 6 | qual, qual 0..62; a byte whose qual bits are 63 is skipped).  Site s owns packed[offsets[s] .. offsets[s+1]).
 * Same records as bvc_lrt_csr on the same observations, bit for bit.  For host callers this halves what crosses
 * the host link, which is what bounds them (src/BaseVarC.cpp:550-559 builds the two vectors this replaces; the
 * successor of bt_s writes the byte directly).  Observations with base quality >= 63 cannot be packed: such tiles
 * stay on bvc_lrt_csr.
 */
int bvc_lrt_csr_packed(bvc_ctx *ctx, int64_t n_sites, const int64_t *offsets, const uint8_t *packed,
                       const int8_t *ref_base, double min_af, bvc_site_result *results, uint32_t flags);

/*
 * Additive: packed dense tiles, ONE byte per (site, sample) instead of two.  The path is bound by the bytes it reads,
 * and (base 0..3, qual 0..62) fits a byte:
 *     packed[s * row_stride + i] = base << 6 | qual        0xFF (any byte whose qual bits are 63) = no observation
 * Results are those of bvc_lrt_dense on the same observations, bit for bit (the histogram is the same).  Base
 * qualities of 63 and more cannot be packed: keep such tiles on bvc_lrt_dense.  A producer (the successor of bt_s,
 * src/BaseVarC.cpp:403-441, which builds the vectors of :550-559) writes the bytes itself; bvc_pack_dense converts a
 * two-byte tile on the device and reports in *n_unrepresentable how many covered samples did not fit (they are
 * written as "no observation": a tile with n_unrepresentable != 0 must not be used).
 */
int bvc_lrt_dense_packed(bvc_ctx *ctx, int64_t n_sites, int64_t n_samples, int64_t row_stride,
                         const uint8_t *packed, const int8_t *ref_base, double min_af,
                         bvc_site_result *results, uint32_t flags);
/* Group mode on a packed tile: bvc_lrt_dense_groups with the one-byte layout (same records, bit for bit). */
int bvc_lrt_dense_groups_packed(bvc_ctx *ctx, int64_t n_sites, int64_t n_samples, int64_t row_stride,
                                const uint8_t *packed, const int8_t *ref_base, double min_af,
                                const uint8_t *group_of_sample, int32_t n_groups,
                                bvc_site_result *results, bvc_group_result *grp_results, uint32_t flags);
/* Device pointers only (BVC_PTR_DEVICE); synchronises the context's stream. */
int bvc_pack_dense(bvc_ctx *ctx, int64_t n_sites, int64_t n_samples, int64_t row_stride,
                   const int8_t *bases, const int8_t *quals, int64_t packed_stride, uint8_t *packed,
                   int64_t *n_unrepresentable, uint32_t flags);

/*
 * Additive: the caller's --group loop on RAGGED columns (src/BaseVarC.cpp:617-661 works on each site's COVERED samples: the
 * vectors of :550-559 plus, per entry, which sample it came from, :633-636).  bvc_lrt_csr with, per observation, the index of
 * its sample: sample_of_obs[i] in 0..n_samples-1 (anything else: in no group); group_of_sample[n_samples] as in
 * bvc_lrt_dense_groups.  grp_results is [n_sites][n_groups].  Same records as bvc_lrt_dense_groups on the dense tile that holds
 * the same observations, bit for bit (the per-group histograms are the same counts), without building, uploading or reading
 * n_samples bytes per site: at the coverage of the cohorts this tool is for (6-10 %) that is 10-16 x fewer bytes.
 */
int bvc_lrt_csr_groups(bvc_ctx *ctx, int64_t n_sites, const int64_t *offsets, const int8_t *bases, const int8_t *quals,
                       const int32_t *sample_of_obs, const int8_t *ref_base, double min_af,
                       const uint8_t *group_of_sample, int64_t n_samples, int32_t n_groups,
                       bvc_site_result *results, bvc_group_result *grp_results, uint32_t flags);

/*
 * Additive: BGZF blocks inflated on the device.  The reference reads its temp batches through htslib's bgzf_getline
 * (src/BaseVarC.cpp:406; written with bgzf_write, :509-527): one raw-deflate stream (RFC 1951) of at most 64 KiB of output per
 * block, blocks independent of each other.  blocks[i] names the deflate payload of a block inside `comp` (the bytes between the
 * 18-byte BGZF header and the 8-byte CRC32 / ISIZE trailer), its ISIZE and where its output goes in `out`; status[i] = 0 when the
 * block inflated to exactly ISIZE bytes, else a non-zero code (the block's output is then undefined; zlib refuses the same streams).
 * With check_crc the CRC32 of the output is computed on the device too and compared.  Host or device pointers (flags).
 */
typedef struct bvc_bgzf_block {
    int64_t comp_off;      /* offset of the deflate payload in comp */
    int64_t out_off;       /* offset of the block's output in out */
    int32_t comp_len;      /* bytes of deflate payload */
    int32_t isize;         /* bytes the block inflates to (<= 65536) */
    uint32_t crc32;        /* CRC32 of the inflated bytes (the block's trailer, RFC 1952) */
    uint32_t check_crc;    /* non-zero: compare (htslib does for every block it reads); a mismatch is status 10 */
} bvc_bgzf_block;
int bvc_inflate_blocks(bvc_ctx *ctx, const uint8_t *comp, int64_t comp_bytes, const bvc_bgzf_block *blocks, int64_t n_blocks,
                       uint8_t *out, int64_t out_bytes, uint32_t *status, uint32_t flags);

/* ---- the producer of the hot path's input on the device (additive) ------------------------------------------------------ */
/*
 * The reference's position loop (bt_s, src/BaseVarC.cpp:403-441) reads one line of every temp-batch file per position and
 * tokenises it with strtok_r / atoi into the position's entries; bt_f then tallies depths and strands (:548-590), builds the
 * (base, qual) vectors (:550-559) and calls BaseType on them (:612-613; per group :617-661).  These two calls do all of that for
 * a TILE of positions from the inflated TEXT of the temp batches (format: writer :509-527): the text goes to the device as it is,
 * the parsed columns feed the LRT there, and what bt_f and WriteVcf read afterwards comes back.
 *
 *   text         the tile's text: for every temp batch the lines of the tile's positions ('\n' after each), anywhere in the buffer
 *   line_start   [n_batches][n_positions + 1] offsets into text: line t of batch b starts at line_start[b * (n_positions + 1) + t];
 *                entry n_positions of a batch = one past the '\n' of its last line (lines of a batch are consecutive)
 *   sample0      [n_batches] index of the batch's first sample (the running j of :407); n_in_batch: samples (= tokens per line)
 *
 * bvc_pileup_begin uploads and counts; it returns BVC_OK and the sizes of what bvc_pileup_finish will deliver, or
 * BVC_PILEUP_IRREGULAR when some line is not what the reference's writer produces (tokens ". ", "b,m,q,r,s " with 1-3 digits a
 * field, or an indel token starting '+', '-' or 'N'; ONE space after every token; as many tokens as the batch has samples): the
 * reference's parser gives such lines a meaning too (strtok_r, atoi), which the caller then applies itself (host/pileup.cpp).
 * bvc_pileup_finish (only after a begin that returned BVC_OK, same context) parses, runs the LRT of every position on the device
 * (bvc_lrt_csr on the non-indel entries; bvc_lrt_csr_groups when n_groups > 0) and returns
 *   entry_off  [n_positions + 1]  entries of position t = entries[entry_off[t] .. entry_off[t + 1]), in sample order
 *   tally      [n_positions][32]  [strand << 3 | base] counts of the base entries, + 16 for the indel entries (:560-590)
 *   entries / samples             the entry and the sample it belongs to (aiv and the inverse of idx, :428-429)
 *   indels     [n_indels]         which entries are indel tokens and where their text is (any order)
 *   results / grp_results         as bvc_lrt_csr / bvc_lrt_csr_groups
 * carry: base, mapq, qual, rpr, strand of the last base token parsed before the tile / after it -- the reference keeps ONE
 * AlleleInfo alive across tokens, lines and positions and an indel token only sets is_indel / indel (:392, 407-440), so an indel
 * entry shows the fields of the last base token before it.
 * Host pointers only; both calls synchronise the context's stream.
 */
typedef struct bvc_pileup_entry {       /* AlleleInfo without its string, src/BamProcess.h:30-39 */
    uint8_t  base, mapq, qual, rpr, strand, is_indel;
    uint16_t pad;
} bvc_pileup_entry;
typedef struct bvc_pileup_indel {
    int64_t entry;                      /* index into entries */
    int64_t text_off;                   /* the token's text in the caller's buffer */
    int32_t len;
    int32_t pad;
} bvc_pileup_indel;
int bvc_pileup_begin(bvc_ctx *ctx, const char *text, int64_t text_bytes, const uint32_t *line_start,
                     const int32_t *sample0, const int32_t *n_in_batch, int32_t n_batches, int32_t n_positions,
                     int64_t *n_entries, int64_t *n_indels);
int bvc_pileup_finish(bvc_ctx *ctx, const int8_t *ref_base, double min_af, const uint8_t carry_in[5], uint8_t carry_out[5],
                      const uint8_t *group_of_sample, int64_t n_samples, int32_t n_groups,
                      int64_t *entry_off, int32_t *tally, bvc_pileup_entry *entries, int32_t *samples,
                      bvc_pileup_indel *indels, char *indel_text, bvc_site_result *results, bvc_group_result *grp_results);
/*
 * bvc_pileup_finish with the entries of the CALLED positions only (results[t].called != 0).  Of a position that is not called the
 * reference reads the tallies and the indel strings alone (the CVG line, src/BaseVarC.cpp:560-610); the entries themselves are read
 * by WriteVcf (:664), i.e. for the few per cent of the positions that are called -- at 1e5 samples and 10 % coverage the entries are
 * 120 KB per position that a caller otherwise receives and never looks at.
 *   called_off [n_positions + 1]  entries / samples of position t = [called_off[t] .. called_off[t + 1]) (empty unless called);
 *                                 gathered on the device, in position order
 *   called_cap                    room in entries / samples, in entries (n_entries of the begin call always suffices; BVC_ERR_ARG when
 *                                 the called positions hold more)
 * entry_off, indels (records' entry = index among ALL entries, as before) and everything else as bvc_pileup_finish.
 */
int bvc_pileup_finish_called(bvc_ctx *ctx, const int8_t *ref_base, double min_af, const uint8_t carry_in[5], uint8_t carry_out[5],
                             const uint8_t *group_of_sample, int64_t n_samples, int32_t n_groups,
                             int64_t *entry_off, int32_t *tally, int64_t *called_off, int64_t called_cap, bvc_pileup_entry *entries,
                             int32_t *samples, bvc_pileup_indel *indels, char *indel_text, bvc_site_result *results,
                             bvc_group_result *grp_results);
/*
 * The same from the COMPRESSED temp batches: the BGZF blocks go to the device as they are in the files (a fifth of the bytes of
 * their text), are inflated there (bvc_inflate_blocks) and the text never exists on the host.  The caller no longer knows where the
 * lines are, so the call decides the tile's positions itself: every batch's text so far = what the previous call left of it (kept
 * on the device) + its new blocks; the tile = the lines EVERY batch has whole, at most max_positions.
 *   blocks            the new blocks of all batches, batch after batch (out_off is ignored); blocks_of_batch[b] = how many are b's
 *   skip_bytes        NULL, or per batch the bytes at the start of its FIRST block that are not position lines (the sample-names
 *                     line, src/BaseVarC.cpp:495, 503); only read where the batch has nothing left over
 *   reset             non-zero: forget what earlier calls left (a new window of positions)
 *   n_positions       out: the tile's positions T (0: some batch has no whole line yet -- send more of its blocks; lines_of_batch
 *                     says which); lines_of_batch[b]: whole lines batch b had (before the T of this tile are taken away)
 *   indel_text_bytes  out: size of the buffer bvc_pileup_finish's indel_text needs (the indel tokens' text: records' text_off are
 *                     then offsets into it; with bvc_pileup_begin indel_text may be NULL and text_off stays an offset into text)
 * Returns BVC_OK, BVC_PILEUP_IRREGULAR (the tile's T positions are decided and consumed, but some line is not regular: fetch the text
 * with bvc_pileup_text and parse it with the reference's rules), BVC_ERR_DATA (a block is not valid deflate of its ISIZE bytes, or -- check_crc -- fails its CRC32)
 * or another error.
 */
int bvc_pileup_begin_bgzf(bvc_ctx *ctx, const uint8_t *comp, int64_t comp_bytes, const bvc_bgzf_block *blocks,
                          const int32_t *blocks_of_batch, const int32_t *skip_bytes, const int32_t *sample0, const int32_t *n_in_batch,
                          int32_t n_batches, int32_t max_positions, int32_t reset, int32_t *n_positions, int32_t *lines_of_batch,
                          int64_t *n_entries, int64_t *n_indels, int64_t *indel_text_bytes);
/* After bvc_pileup_begin_bgzf: the tile's text (text_bytes from *text_bytes_needed: call with text = NULL first) and its line table
 * [n_batches][n_positions + 1] as bvc_pileup_begin takes it. */
int bvc_pileup_text(bvc_ctx *ctx, char *text, int64_t text_cap, int64_t *text_bytes_needed, uint32_t *line_start);

/* ---- the two stages on their own (used by the parity tests; also valid entry points) -------------- */
/* Stage 1: counts[s * 512 + base * 128 + qual] = number of covered samples of that class (exact). */
int bvc_hist_dense(bvc_ctx *ctx, int64_t n_sites, int64_t n_samples, int64_t row_stride,
                   const int8_t *bases, const int8_t *quals, uint32_t *counts, uint32_t flags);
/* Stage 1 on a packed tile (device pointers only): the same counts[s * 512 + base * 128 + qual]. */
int bvc_hist_dense_packed(bvc_ctx *ctx, int64_t n_sites, int64_t n_samples, int64_t row_stride,
                          const uint8_t *packed, uint32_t *counts, uint32_t flags);
/* Stage 2: EM + LRT on per-site class counts.  base_comb (optional): [n_sites][4] candidate bases in
 * SetBase order with n_comb[s] entries used; NULL means the default {A,C,G,T} (src/BaseType.h:79).
 * base_comb / n_comb contract (also bvc_lrt_csr_comb): with BVC_PTR_HOST an entry outside 0..3 or n_comb[s] > 4 is
 * rejected with BVC_ERR_ARG before anything runs.  With BVC_PTR_DEVICE the library cannot look at the arrays without
 * a round trip, so the kernels apply the reference's own rule instead: a candidate that is not A, C, G or T has
 * depth 0 in BaseType (src/BaseType.cpp:79 reads depth[b]) and falls to the min_af filter, i.e. the entry is
 * ignored (the record equals the one for the list without it), and n_comb[s] > 4 is read as 4. */
int bvc_lrt_hist(bvc_ctx *ctx, int64_t n_sites, const uint32_t *counts, const int8_t *ref_base,
                 double min_af, const int8_t *base_comb, const uint8_t *n_comb,
                 bvc_site_result *results, uint32_t flags);

/* ---- synthetic pileup generator (benchmark / test input); device pointers only ------ */
/* Fills sites site0 .. site0+n_sites-1.  cov_thr16 = 65536 gives dense coverage; smaller values leave
 * a sample uncovered (base = -1) with probability 1 - cov_thr16/65536.  Integer arithmetic only. */
int bvc_synth_dense(bvc_ctx *ctx, uint64_t seed, int64_t site0, int64_t n_sites, int64_t n_samples,
                    int64_t row_stride, uint32_t cov_thr16, int8_t *bases, int8_t *quals,
                    int8_t *ref_base);

/* ---- tuning (per context; additive, no counterpart in the reference) -------------------------------- */
/* Launch policy of THIS context; results never depend on it.  Keys:
 *   "em_waves_per_cu"  0 = default policy, 1..32 = stage-2 wavefronts per CU and launch: a call's stage 2 then runs as a
 *                      SEQUENCE of launches of at most that many wavefronts per CU (a region of eight sites belongs to a team
 *                      of two wavefronts; a workgroup holds up to two teams).  Default: unbounded when stage 2 has the chip to
 *                      itself, 4 underneath a streaming histogram pass (overlap mode, rows of 200,000 samples and more)
 *   "em_wpb"           waves per EM workgroup: 4 (default) or 1
 *   "hist_split"       0 = by tile shape, 1..64 workgroups sharing a site in the dense histogram pass
 *   "em_streams"       overlap mode: stage 2 of consecutive calls on one side stream (1), on two alternating ones (2: the
 *                      next EM launch fills the previous one's tail), or chosen by row length (0, default)
 *   "group_pipe"       any-order group histogram: 1 (default) issues the next chunk's loads before counting the
 *                      current one, 0 loads two chunks then counts both
 *   "group_copies_log2"  any-order group histograms: -1 (default) = as many LDS copies of each histogram as fit 64 KiB,
 *                      0..5 = at most 2^n copies (smaller workgroups, more of them resident; A/B runs)
 *   "group_big_lds"    any-order group histograms, 4 groups and more: 1 (default) = one 1024-thread workgroup per CU with 256
 *                      slots x 16 / 8 / 4 copies per histogram in up to 144 KiB of LDS (two-byte rows are packed in registers for
 *                      it; sites with a covered quality of 63 or more are redone by the general kernel), 0 = 512-thread
 *                      workgroups of 64 KiB
 *   "host_chunk_kib"   BVC_PTR_HOST calls stage the tile through device memory in chunks of sites of at most this
 *                      many KiB per array (default 524288 = 512 MiB); the upload of chunk i+1 runs under the kernels
 *                      of chunk i
 * A new context starts from the environment variables BVC_EM_WAVES_PER_CU, BVC_EM_WPB, BVC_HIST_SPLIT,
 * BVC_GROUP_PIPE, BVC_GROUP_LOG2C, BVC_GROUP_BIG_LDS, BVC_EM_STREAMS when they are set.
 *
 * One key is NOT a launch policy:
 *   "em_engine"        0 (default) = the item engine of stage 2 (fits as work items, csrc/em_items.hip), 1 = one
 *                      wavefront per site for every site (csrc/em_kernel.hip, the engine of rounds 1-2, which otherwise
 *                      only takes the sites the item engine leaves).  The two evaluate the same sums in different
 *                      orders: calls, depths, pass counts and every integer field agree (except where two subsets tie
 *                      to rounding), AF / chi / var_qual agree to ~1e-15 relative, not bit for
 *                      bit.  With either engine a site's record never depends on the call's size or on its neighbours.
 *                      Environment: BVC_EM_ENGINE.
 *   "em_tiny_regions"  0 (default) / 1: with 1, regions of eight sites that all have at most 8 quality values per allele
 *                      (binned qualities) take a kernel with one lane per allele (stage 2 1.36 x faster on such data).  That
 *                      kernel adds in a different order, so a site's AF / chi could differ in the last bits with the
 *                      binning of its seven region neighbours: off by default.  Environment: BVC_EM_TINY_REGIONS.
 *   "em_prune"         1 (default) / 0.  A level of the likelihood-ratio test goes on with the FIRST MINIMUM of chi over its
 *                      subsets and reads nothing else of the others (src/BaseType.cpp:97-105).  With 1 the item engine runs the
 *                      subsets that keep the deepest allele first and does not run the one without it when an upper bound on
 *                      that subset's log-likelihood (every class marginal <= 1; the left-out allele's observations have
 *                      marginal e exactly) puts its chi above the minimum of the others: the decision, and every field of the
 *                      record the reference defines, is unchanged, the diagnostics n_fits / n_passes count what was run.  0 =
 *                      the subset is always run (n_fits / n_passes then equal the reference's counts).  Environment: BVC_EM_PRUNE. */
int bvc_set_tuning(bvc_ctx *ctx, const char *key, int value);

/* ---- measurement aid ------------------------------------------------------------------------------- */
/* Streams `bytes` of device memory once with 16-byte loads per lane and nothing else; HIP-event time in ms.
 * The empirical HBM read ceiling to hold next to the spec peak when judging the histogram kernel. */
int bvc_stream_read_ms(bvc_ctx *ctx, const void *device_ptr, int64_t bytes, int repeats, double *ms_per_pass);

#ifdef __cplusplus
}
#endif
#endif /* BVC_H */
