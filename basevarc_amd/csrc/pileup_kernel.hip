// pileup_kernel.hip -- the producer side of the hot path on the device: temp-batch pileup TEXT -> ragged pileup columns,
// and the per-group histograms of ragged columns.
//
// Reference (paths under /root/reference): the position loop of bt_s parses one line of every temp batch per position with
// strtok_r / atoi (src/BaseVarC.cpp:403-441; writer :509-527), bt_f then walks the position's entries for the depth and strand
// tallies of the CVG line (:548-590) and builds the (base, qual) vectors BaseType takes (:550-559), for the whole cohort and, with
// --group, per population group (:617-661).  On the host that is ~140 us of CPU per position of 1e5 samples; the device takes the
// inflated text as it is.
//
// Layout: the text of a tile = for every temp batch b the T lines of the tile's positions, one after the other; the caller gives
// the offset of every line (it cut the batch's stream at line ends anyway).  One WAVEFRONT per line, 16 bytes of the line per lane
// and step; token starts from a compare mask, token and entry ranks from wave prefix sums; two passes over the text -- count, then
// write -- with a prefix sum over the lines (position-major: the entries of a position are those of its lines in batch order, as
// the reference appends them) in between.  Exactly the observable behaviour of the reference's parser on the lines its writer
// produces (REGULAR lines: tokens ". ", "b,m,q,r,s " with 1-3 digits per field, or an indel token starting '+', '-' or 'N', each
// followed by ONE space; as many tokens as the batch has samples): N bases (base code 4) dropped (:427), the fields' bit widths
// (src/BamProcess.h:32-37), and the long-lived AlleleInfo whose fields an indel entry inherits from the last base token parsed
// before it -- in the same line, in an earlier line of the tile, or in the tile before (:392, 407-440).  A line that is not regular
// is only counted: the call then reports BVC_PILEUP_IRREGULAR and the caller parses that tile on the CPU.
#include <hip/hip_runtime.h>

#include "bvc_device.h"
#include "bvc_internal.h"

namespace bvc {
namespace {

constexpr int kParseWaves = 4;                  // lines per workgroup (one wavefront each)
constexpr uint32_t kTokValid = 0x80u;           // packed base token: bits 0..2 base, 3 strand, 7 valid, 8..15 mapq, 16..23 qual, 24..31 rpr

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const uint32_t u = (uint32_t)__shfl_up((int)v, d, kWave);
        if (lane >= d) v += u;
    }
    return v;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d, kWave);
    return v;
}

// "b,m,q,r,s " at text[p..] (p < e): five fields of one to three decimal digits, four commas, a space.  Returns the packed token,
// 0 for any other shape.  Three aligned 8-byte loads cover the 20 bytes of the longest such token wherever p falls (the text
// buffer is allocated 32 bytes longer than the text).
__device__ __forceinline__ uint32_t parse_base_token(const uint8_t *__restrict__ text, uint32_t p, uint32_t e)
{
    const uint64_t *w = reinterpret_cast<const uint64_t *>(text + (p & ~7u));
    const uint64_t w0 = w[0], w1 = w[1], w2 = w[2];
    const uint32_t sh = p & 7u;
    uint32_t f = 0, nd = 0, acc = 0, v0 = 0, v1 = 0, v2 = 0, v3 = 0;
#pragma unroll
    for (int i = 0; i < 20; ++i) {
        const uint32_t k = sh + (uint32_t)i;
        uint32_t c = (uint32_t)((k < 8u ? w0 >> (8u * k) : (k < 16u ? w1 >> (8u * (k - 8u)) : w2 >> (8u * (k - 16u)))) & 0xFFu);
        if (p + (uint32_t)i >= e) c = 0x20u;
        if (c == 0x20u) {
            if (nd == 0u || f != 4u) return 0u;
            return (v0 & 7u) | ((acc & 1u) << 3) | kTokValid | ((v1 & 0xFFu) << 8) | ((v2 & 0xFFu) << 16) | ((v3 & 0xFFu) << 24);
        }
        if (c == 0x2Cu) {
            if (nd == 0u || f == 4u) return 0u;
            if (f == 0u) v0 = acc; else if (f == 1u) v1 = acc; else if (f == 2u) v2 = acc; else v3 = acc;
            ++f; nd = 0u; acc = 0u;
        } else {
            const uint32_t d = c - 0x30u;
            if (d > 9u || nd == 3u) return 0u;
            acc = acc * 10u + d; ++nd;
        }
    }
    return 0u;
}

struct ParseArgs {
    const uint8_t *text;
    const uint32_t *line_start;      // [n_batches][n_pos + 1]: offset of line t of batch b; the last = one past the batch's last '\n'
    const int32_t *sample0;          // [n_batches]
    const int32_t *n_in_batch;       // [n_batches] tokens of every line of the batch
    int32_t n_batches, n_pos;
    int32_t line_stride;             // elements of line_start per batch (n_pos + 1 when the caller built the table)
    const int32_t *n_pos_dev;        // not null: the number of positions is decided on the device (tiles inflated there) and read from here
    uint32_t *line_entries;          // [n_pos * n_batches] position-major: count pass = entries of the line; after the scan = first entry
    uint32_t *line_obs;              // the same for observations (entries that are not indels)
    uint32_t *line_last;             // last base token of the line (packed) or 0
    uint32_t *line_need;             // indel entries in front of the line's first base token: they inherit from an earlier line
    uint32_t *status;                // [0] irregular lines, [1] indel entries counted, [2] indel records written, [4] bytes of indel tokens
    bvc_pileup_entry *entries;
    int32_t *samples;
    int8_t *obs_base, *obs_qual;
    int32_t *obs_sample;
    int32_t *tally;                  // [n_pos][32]: [strand << 3 | base] of the base entries, + 16 for the indel entries
    bvc_pileup_indel *indels;
    uint32_t indel_cap;
};

template <bool WRITE>
__global__ __launch_bounds__(kParseWaves * kWave) void pileup_parse_kernel(ParseArgs A)
{
    BVC_POISON_LDS();
    __shared__ uint32_t tal_all[kParseWaves][32];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t *tal = tal_all[wave];
    if (lane < 32) tal[lane] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const uint8_t *__restrict__ text = A.text;
    const int64_t n_lines = (int64_t)(A.n_pos_dev ? *A.n_pos_dev : A.n_pos) * A.n_batches;
    for (int64_t line = (int64_t)blockIdx.x * kParseWaves + wave; line < n_lines; line += (int64_t)gridDim.x * kParseWaves) {
        const int t = (int)(line / A.n_batches), b = (int)(line - (int64_t)t * A.n_batches);
        const uint32_t s = A.line_start[(int64_t)b * A.line_stride + t];
        const uint32_t e1 = A.line_start[(int64_t)b * A.line_stride + t + 1];
        bool bad = e1 <= s;
        const uint32_t e = bad ? s : e1 - 1u;                    // the line is [s, e); text[e] is its '\n'
        if (!bad) bad = text[e] != 0x0Au || (e > s && text[e - 1u] != 0x20u);
        const int32_t smp0 = A.sample0[b];
        uint32_t ent_at = 0, obs_at = 0;
        if (WRITE) { ent_at = A.line_entries[line]; obs_at = A.line_obs[line]; }
        uint32_t tok_run = 0, ent_run = 0, obs_run = 0, need = 0, n_ind_line = 0, ind_bytes = 0;
        uint32_t prev_tok = 0;                                   // wave-uniform: last base token of the line so far
        uint32_t prev_sep = 1u;                                  // wave-uniform: was the byte in front of this step a separator?
        for (uint32_t off = s & ~15u; off < e; off += 16u * kWave) {
            const uint32_t a = off + 16u * (uint32_t)lane;
            u32x4 w = u32x4{0x20202020u, 0x20202020u, 0x20202020u, 0x20202020u};
            uint32_t c16 = 0x20u;
            if (a < e && a + 16u > s) {
                w = *reinterpret_cast<const u32x4 *>(text + a);
                if (a + 16u < e) c16 = text[a + 16u];
            }
            const uint32_t wv[4] = {w.x, w.y, w.z, w.w};
            uint32_t sep = 0, dot = 0, dig = 0, ind = 0, inside = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const uint32_t pos = a + (uint32_t)i;
                uint32_t c = (wv[i >> 2] >> (8 * (i & 3))) & 0xFFu;
                const bool in = pos >= s && pos < e;
                if (!in) c = 0x20u;
                inside |= (uint32_t)in << i;
                sep |= (uint32_t)(c == 0x20u) << i;
                dot |= (uint32_t)(c == 0x2Eu) << i;
                dig |= (uint32_t)(c - 0x30u <= 9u) << i;
                ind |= (uint32_t)(c == 0x2Bu || c == 0x2Du || c == 0x4Eu) << i;
            }
            sep |= (uint32_t)(c16 == 0x20u) << 16;
            const uint32_t left = (uint32_t)__shfl_up((int)sep, 1, kWave);
            const uint32_t pb = lane == 0 ? prev_sep : (left >> 15) & 1u;
            const uint32_t after_sep = ((sep << 1) | pb) & 0xFFFFu;
            const uint32_t start = ~sep & after_sep & 0xFFFFu;
            // not regular: two separators in a row inside the line, a token that starts with none of [0-9.+-N], a '.' token longer than the dot
            if ((sep & after_sep & inside) | (start & ~(dot | dig | ind)) | (start & dot & ~(sep >> 1))) bad = true;
            // ---- first walk over the lane's tokens: what each adds
            const uint32_t cand = start & (dig | ind);
            uint32_t ent = 0, obs = 0, n_ind = 0, ind_front = 0, lane_last = 0;
            for (uint32_t m = cand; m;) {
                const int i = __builtin_ctz(m);
                m &= m - 1u;
                if ((dig >> i) & 1u) {
                    const uint32_t tok = parse_base_token(text, a + (uint32_t)i, e);
                    if (tok == 0u) { bad = true; continue; }
                    lane_last = tok;
                    if ((tok & 7u) != 4u) { ++ent; ++obs; }      // skip N base, src/BaseVarC.cpp:427
                } else {
                    ++ent; ++n_ind;
                    if (lane_last == 0u) ++ind_front;
                    if (!WRITE) {                                // the token's length: the caller sizes the buffer for the Indels column's text
                        uint32_t q = a + (uint32_t)i + 1u;
                        while (q < e && text[q] != 0x20u) ++q;
                        ind_bytes += q - (a + (uint32_t)i);
                    }
                }
            }
            const uint32_t packed = (uint32_t)__builtin_popcount(start) | (ent << 10) | (obs << 20);
            const uint32_t incl = wave_incl_scan(packed, lane);
            const uint32_t excl = incl - packed;
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            const uint64_t have = __ballot(lane_last != 0u);
            const bool any_ind = __ballot(n_ind != 0u) != 0ull;
            uint32_t left_last = 0;                              // last base token of the lanes in front of this one (this step)
            if (any_ind) {
                const uint64_t before = have & ((1ull << lane) - 1ull);
                const int src = before ? 63 - __builtin_clzll(before) : 0;
                const uint32_t got = (uint32_t)__shfl((int)lane_last, src, kWave);
                left_last = before ? got : 0u;
                // indel entries in front of the line's first base token
                if (prev_tok == 0u) {
                    const int first = have ? __builtin_ctzll(have) : kWave;
                    need += wave_sum(lane < first ? n_ind : (lane == first ? ind_front : 0u));
                }
                n_ind_line += wave_sum(n_ind);
            }
            // ---- second walk: write
            if (WRITE && !bad) {
                uint32_t ent_i = ent_at + ent_run + ((excl >> 10) & 0x3FFu);
                uint32_t obs_i = obs_at + obs_run + ((excl >> 20) & 0x3FFu);
                const uint32_t tok_i = tok_run + (excl & 0x3FFu);
                uint32_t cur_last = 0;
                for (uint32_t m = cand; m;) {
                    const int i = __builtin_ctz(m);
                    m &= m - 1u;
                    const uint32_t p = a + (uint32_t)i;
                    const int32_t smp = smp0 + (int32_t)(tok_i + (uint32_t)__builtin_popcount(start & ((1u << i) - 1u)));
                    if ((dig >> i) & 1u) {
                        const uint32_t tok = parse_base_token(text, p, e);
                        cur_last = tok;
                        if ((tok & 7u) == 4u) continue;
                        *reinterpret_cast<u32x2 *>(&A.entries[ent_i]) =
                            u32x2{(tok & 7u) | (tok & 0xFFFFFF00u), (tok >> 3) & 1u};
                        A.samples[ent_i] = smp;
                        A.obs_base[obs_i] = (int8_t)(tok & 7u);
                        A.obs_qual[obs_i] = (int8_t)((tok >> 16) & 0xFFu);
                        A.obs_sample[obs_i] = smp;
                        atomicAdd(&tal[tok & 15u], 1u);
                        ++ent_i; ++obs_i;
                    } else {
                        // the entry carries the fields of the last base token parsed before it (src/BaseVarC.cpp:431-436)
                        const uint32_t src = cur_last ? cur_last : (left_last ? left_last : prev_tok);
                        *reinterpret_cast<u32x2 *>(&A.entries[ent_i]) =
                            u32x2{(src & 7u) | (src & 0xFFFFFF00u), ((src >> 3) & 1u) | 0x100u};
                        A.samples[ent_i] = smp;
                        if (src) atomicAdd(&tal[16u + (src & 15u)], 1u);
                        uint32_t q = p + 1u;                         // the token's text, for the CVG line's Indels column
                        while (q < e && text[q] != 0x20u) ++q;
                        const uint32_t k = atomicAdd(&A.status[2], 1u);
                        if (k < A.indel_cap) A.indels[k] = bvc_pileup_indel{(int64_t)ent_i, (int64_t)p, (int32_t)(q - p), 0};
                        ++ent_i;
                    }
                }
            }
            tok_run += total & 0x3FFu; ent_run += (total >> 10) & 0x3FFu; obs_run += (total >> 20) & 0x3FFu;
            if (have) prev_tok = (uint32_t)__shfl((int)lane_last, 63 - __builtin_clzll(have), kWave);
            prev_sep = (uint32_t)__builtin_amdgcn_readlane((int)sep, 63) >> 15 & 1u;
        }
        if (tok_run != (uint32_t)A.n_in_batch[b]) bad = true;
        const bool any_bad = __ballot(bad) != 0ull;
        if (!WRITE) {
            const uint32_t ind_bytes_line = n_ind_line ? wave_sum(ind_bytes) : 0u;      // (n_ind_line is wave-uniform)
            if (lane == 0) {
                if (ind_bytes_line) atomicAdd(&A.status[4], ind_bytes_line);
                A.line_entries[line] = any_bad ? 0u : ent_run;
                A.line_obs[line] = any_bad ? 0u : obs_run;
                A.line_last[line] = prev_tok;
                A.line_need[line] = need;
                if (any_bad) atomicAdd(&A.status[0], 1u);
                if (n_ind_line) atomicAdd(&A.status[1], n_ind_line);
            }
        } else {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            if (lane < 32) {
                const uint32_t v = tal[lane];
                if (v) atomicAdd(&A.tally[(int64_t)t * 32 + lane], (int32_t)v);
                tal[lane] = 0u;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// Exclusive prefix sums of the lines' entry and observation counts, in place, position-major; the positions' offsets; the totals.
// One workgroup walks the arrays 4096 lines at a time: four consecutive lines per thread (one 16-byte load per array: coalesced), a
// scan of the 1024 thread sums by wave shuffles and one LDS exchange, the running base carried in registers.  (Round 5: a contiguous
// run of n / 1024 lines per thread read the arrays with a stride of 340 bytes between lanes: 0.32 ms for the 87,000 lines of a tile.)
__global__ __launch_bounds__(1024) void pileup_scan_kernel(int64_t n_lines, int32_t n_batches, int32_t n_pos, const int32_t *__restrict__ n_pos_dev,
                                                           uint32_t *__restrict__ line_entries,
                                                           uint32_t *__restrict__ line_obs, int64_t *__restrict__ entry_off,
                                                           int64_t *__restrict__ obs_off, int64_t *__restrict__ totals)
{
    BVC_POISON_LDS();
    if (n_pos_dev) { n_pos = *n_pos_dev; n_lines = (int64_t)n_pos * n_batches; }
    __shared__ uint64_t wave_e[16], wave_o[16];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    uint64_t run_e = 0, run_o = 0;                               // everything before this step's lines (the same in every thread)
    const bool vec = (((uintptr_t)line_entries | (uintptr_t)line_obs) & 15u) == 0u;   // (the second array starts n_lines_cap words in)
    for (int64_t base = 0; base < n_lines; base += 4096) {
        const int64_t i0 = base + 4 * (int64_t)tid;
        uint32_t ce[4] = {0, 0, 0, 0}, co[4] = {0, 0, 0, 0};
        if (vec && i0 + 4 <= n_lines) {
            const u32x4 ve = *reinterpret_cast<const u32x4 *>(line_entries + i0), vo = *reinterpret_cast<const u32x4 *>(line_obs + i0);
            ce[0] = ve.x; ce[1] = ve.y; ce[2] = ve.z; ce[3] = ve.w; co[0] = vo.x; co[1] = vo.y; co[2] = vo.z; co[3] = vo.w;
        } else {
            for (int k = 0; k < 4; ++k) if (i0 + k < n_lines) { ce[k] = line_entries[i0 + k]; co[k] = line_obs[i0 + k]; }
        }
        const uint64_t se = (uint64_t)ce[0] + ce[1] + ce[2] + ce[3], so = (uint64_t)co[0] + co[1] + co[2] + co[3];
        uint64_t ie = se, io = so;                               // inclusive scan over the wavefront's 64 thread sums
        for (int d = 1; d < kWave; d <<= 1) {
            const uint64_t ue = __shfl_up(ie, d), uo = __shfl_up(io, d);
            if (lane >= d) { ie += ue; io += uo; }
        }
        if (lane == kWave - 1) { wave_e[wave] = ie; wave_o[wave] = io; }
        __syncthreads();
        uint64_t be = run_e + ie - se, bo = run_o + io - so, te = 0, to = 0;
        for (int w = 0; w < 16; ++w) {
            const uint64_t we = wave_e[w], wo = wave_o[w];
            if (w < wave) { be += we; bo += wo; }
            te += we; to += wo;
        }
        __syncthreads();                                         // (wave_e / wave_o are written again in the next step)
        uint32_t oe[4], oo[4];
        for (int k = 0; k < 4; ++k) {
            oe[k] = (uint32_t)be; oo[k] = (uint32_t)bo;
            const int64_t i = i0 + k;
            if (i < n_lines && (uint32_t)(i % n_batches) == 0u) { entry_off[i / n_batches] = (int64_t)be; obs_off[i / n_batches] = (int64_t)bo; }
            be += ce[k]; bo += co[k];
        }
        if (vec && i0 + 4 <= n_lines) {
            *reinterpret_cast<u32x4 *>(line_entries + i0) = u32x4{oe[0], oe[1], oe[2], oe[3]};
            *reinterpret_cast<u32x4 *>(line_obs + i0) = u32x4{oo[0], oo[1], oo[2], oo[3]};
        } else {
            for (int k = 0; k < 4; ++k) if (i0 + k < n_lines) { line_entries[i0 + k] = oe[k]; line_obs[i0 + k] = oo[k]; }
        }
        run_e += te; run_o += to;
    }
    if (tid == 0) {
        entry_off[n_pos] = (int64_t)run_e; obs_off[n_pos] = (int64_t)run_o;
        totals[0] = (int64_t)run_e; totals[1] = (int64_t)run_o;
    }
}

// The indel entries in front of a line's first base token take the fields of the last base token of the lines before it
// (position-major order: the order the reference parses in), or of the tile before (carry_in).  One thread per line; thread
// n_lines leaves the tile's own last base token for the next tile.
__global__ void pileup_patch_kernel(int64_t n_lines, int32_t n_batches, const uint32_t *__restrict__ line_entries,
                                    const uint32_t *__restrict__ line_last, const uint32_t *__restrict__ line_need, uint32_t carry_in,
                                    bvc_pileup_entry *__restrict__ entries, int32_t *__restrict__ tally, uint32_t *__restrict__ carry_out)
{
    const int64_t line = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (line > n_lines) return;
    const uint32_t need = line < n_lines ? line_need[line] : 0u;
    if (line < n_lines && need == 0u) return;
    uint32_t src = carry_in;
    for (int64_t j = line - 1; j >= 0; --j) {
        const uint32_t l = line_last[j];
        if (l) { src = l; break; }
    }
    if (line == n_lines) { *carry_out = src; return; }
    const uint32_t at = line_entries[line];
    for (uint32_t k = 0; k < need; ++k)
        *reinterpret_cast<u32x2 *>(&entries[at + k]) = u32x2{(src & 7u) | (src & 0xFFFFFF00u), ((src >> 3) & 1u) | 0x100u};
    atomicAdd(&tally[(line / n_batches) * 32 + 16 + (src & 15u)], (int32_t)need);
}

// ---- per-group histograms of ragged columns ----------------------------------------------------------------------------
// counts[site][h][512], h = the group of the observation's sample (labels >= n_groups, or a sample index outside the label
// vector: the "no group" histogram n_groups), as the dense group kernels lay them out.  The reference's group loop runs on each
// site's COVERED samples (src/BaseVarC.cpp:617-661, vectors of :550-559): this is that loop's input without the dense
// [site][N] tile in between.  One workgroup per site, one LDS counter per (histogram, class), LDS atomics: a ragged site is
// 1e1-1e5 observations, far fewer than a dense row, and what bounds a site here is zeroing and folding (k + 1) x 512 counters.
constexpr int kCsrGroupThreads = 512;

// One workgroup per site; LDS [histogram][class][copy] with as many copies (a power of two, copy = lane mod copies) as fit 64 KiB --
// 4 at k = 5 -- so that the handful of hot classes of a pileup (the reference base at ~30 qualities, per group) are not ONE counter
// each for 512 lanes.  6 algorithmic bytes per observation (base, quality, sample index) + a label byte gathered from the L2-resident
// label vector.
__global__ __launch_bounds__(kCsrGroupThreads) void hist_csr_groups_kernel(
    int64_t n_sites, const int64_t *__restrict__ offsets, const int8_t *__restrict__ bases, const int8_t *__restrict__ quals,
    const int32_t *__restrict__ sample_of_obs, const uint8_t *__restrict__ group_of_sample, int64_t n_samples, int n_groups,
    int log2c, uint32_t *__restrict__ counts)
{
    BVC_POISON_LDS();
    extern __shared__ __attribute__((aligned(16))) uint32_t ghist[];        // [n_groups + 1][512][1 << log2c]
    const int tid = threadIdx.x;
    const int n_hist = n_groups + 1;
    const int classes = n_hist * BVC_NCLASS;
    const int words = classes << log2c;
    const uint32_t copy = (uint32_t)tid & ((1u << log2c) - 1u);
    auto one = [&](uint32_t b, uint32_t q, int64_t smp) {
        if (b < 4u && q < 128u) {
            uint32_t g = (uint32_t)n_groups;
            if (smp >= 0 && smp < n_samples) { g = group_of_sample[smp]; if (g > (uint32_t)n_groups) g = (uint32_t)n_groups; }
            const uint32_t at = ((g * BVC_NCLASS + ((b << 7) | q)) << log2c) | copy;
            if (BVC_LDS_OK(0x501, at, words)) atomicAdd(&ghist[at], 1u);
        }
    };
    for (int i = tid * 4; i < words; i += kCsrGroupThreads * 4) *reinterpret_cast<u32x4 *>(&ghist[i]) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    for (int64_t site = blockIdx.x; site < n_sites; site += gridDim.x) {
        const int64_t o0 = offsets[site], o1 = offsets[site + 1];
        // consecutive lanes take consecutive observations: their sample indices ascend, so a wavefront's 64 label gathers fall into
        // a handful of cache lines (at 10 % coverage ~640 bytes of the label vector) -- with 16 observations per lane they were 64
        // different lines per gather instruction and the kernel ran at the L1 miss rate (round 5: 1.9 ms against 1.5 per 4000 sites of
        // 1e5 observations).  Four observations per lane and trip, their loads issued together.
        constexpr int U = 4;
        int64_t i = o0 + tid;
        for (; i + (int64_t)(U - 1) * kCsrGroupThreads < o1; i += (int64_t)U * kCsrGroupThreads) {
            uint32_t b[U], q[U], lab[U];
            int32_t sm[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t at = i + (int64_t)u * kCsrGroupThreads;
                b[u] = (uint8_t)bases[at]; q[u] = (uint8_t)quals[at]; sm[u] = sample_of_obs[at];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) lab[u] = sm[u] >= 0 && (int64_t)sm[u] < n_samples ? group_of_sample[sm[u]] : (uint32_t)n_groups;
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (b[u] < 4u && q[u] < 128u) {
                    const uint32_t g = lab[u] > (uint32_t)n_groups ? (uint32_t)n_groups : lab[u];
                    const uint32_t at = ((g * BVC_NCLASS + ((b[u] << 7) | q[u])) << log2c) | copy;
                    if (BVC_LDS_OK(0x502, at, words)) atomicAdd(&ghist[at], 1u);
                }
        }
        for (; i < o1; i += kCsrGroupThreads) one((uint8_t)bases[i], (uint8_t)quals[i], sample_of_obs[i]);
        __syncthreads();
        uint32_t *dst = counts + site * (int64_t)classes;
        for (int key = tid; key < classes; key += kCsrGroupThreads) {
            uint32_t sum = 0;
            for (int v = 0; v < (1 << log2c); ++v) { const int at = (key << log2c) + ((v + key) & ((1 << log2c) - 1)); sum += ghist[at]; ghist[at] = 0u; }
            dst[key] = sum;
        }
        __syncthreads();
    }
}

}  // namespace

static ParseArgs parse_args_of(const PileupTile &P)
{
    ParseArgs A{};
    A.text = P.text; A.line_start = P.line_start; A.sample0 = P.sample0; A.n_in_batch = P.n_in_batch;
    A.n_batches = P.n_batches; A.n_pos = P.n_pos; A.line_stride = P.line_stride; A.n_pos_dev = P.n_pos_dev;
    A.line_entries = P.line_words; A.line_obs = P.line_words + P.n_lines_cap; A.line_last = P.line_words + 2 * P.n_lines_cap;
    A.line_need = P.line_words + 3 * P.n_lines_cap; A.status = P.status;
    A.entries = P.entries; A.samples = P.samples; A.obs_base = P.obs_base; A.obs_qual = P.obs_qual; A.obs_sample = P.obs_sample;
    A.tally = P.tally; A.indels = P.indels; A.indel_cap = P.indel_cap;
    return A;
}

hipError_t launch_pileup_count(hipStream_t stream, const PileupTile &P)
{
    const ParseArgs A = parse_args_of(P);
    const int64_t n_lines = (int64_t)P.n_pos * P.n_batches;       // (an upper bound when the device decides the positions)
    if (n_lines <= 0) return hipSuccess;
    const int64_t blocks = (n_lines + kParseWaves - 1) / kParseWaves;
    hipLaunchKernelGGL(pileup_parse_kernel<false>, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(kParseWaves * kWave), 0, stream, A);
    hipLaunchKernelGGL(pileup_scan_kernel, dim3(1), dim3(1024), 0, stream, n_lines, P.n_batches, P.n_pos, P.n_pos_dev, A.line_entries,
                       A.line_obs, P.entry_off, P.obs_off, P.totals);
    return hipGetLastError();
}

hipError_t launch_pileup_write(hipStream_t stream, const PileupTile &P, uint32_t carry_in)
{
    ParseArgs A = parse_args_of(P);
    A.n_pos_dev = nullptr;                                       // the caller knows the positions by now
    const int64_t n_lines = (int64_t)P.n_pos * P.n_batches;
    if (n_lines <= 0) return hipSuccess;
    const int64_t blocks = (n_lines + kParseWaves - 1) / kParseWaves;
    hipLaunchKernelGGL(pileup_parse_kernel<true>, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(kParseWaves * kWave), 0, stream, A);
    const int64_t threads = n_lines + 1;
    hipLaunchKernelGGL(pileup_patch_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, n_lines, P.n_batches,
                       A.line_entries, A.line_last, A.line_need, carry_in, P.entries, P.tally, P.status + 3);
    return hipGetLastError();
}

// ---- tiles whose text is inflated on the device: where the lines are, and how many positions every batch has whole ---------
// A batch's region of the text buffer = what the tile before left of it + the output of its new blocks.  One wavefront per
// 1 KiB segment of a region counts its newlines; one thread per batch adds them up (the batch's whole lines) and the tile's
// positions are the fewest any batch has (at most max_pos); the segments then note where each of those lines starts.
struct RegionArgs {
    const uint8_t *text;
    const bvc_pileup_region *regions;      // [n_batches]
    const uint32_t *seg_base;              // [n_batches + 1] first segment of each region
    int32_t n_batches, max_pos, line_stride;
    uint32_t *seg_nl;                      // [segments] newlines of the segment; after the scan: newlines of the region before it
    int32_t *lines;                        // [n_batches] whole lines of the region
    int32_t *n_pos;                        // the tile's positions
    uint32_t *line_start;                  // [n_batches][line_stride]
};

constexpr int kSegBytes = 1024;

template <bool FILL>
__global__ __launch_bounds__(kParseWaves * kWave) void region_lines_kernel(RegionArgs A)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t n_seg = A.seg_base[A.n_batches];
    if (!FILL && blockIdx.x == 0 && threadIdx.x == 0) *A.n_pos = A.max_pos;      // (region_scan_kernel, the next launch, takes minima into it)
    for (int64_t seg = (int64_t)blockIdx.x * kParseWaves + wave; seg < n_seg; seg += (int64_t)gridDim.x * kParseWaves) {
        int lo = 0, hi = A.n_batches;                            // the region of this segment
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int64_t)A.seg_base[mid] <= seg) lo = mid; else hi = mid; }
        const int b = lo;
        const uint32_t r0 = A.regions[b].start, r1 = r0 + A.regions[b].len;
        const uint32_t s0 = r0 + (uint32_t)(seg - A.seg_base[b]) * kSegBytes;
        const uint32_t at = s0 + 16u * (uint32_t)lane;
        uint32_t nl = 0;                                         // bit i: byte at + i is a newline inside the region
        if (at < r1) {
            // (the region starts where the caller put it: bytes, not 16-byte words)
#pragma unroll
            for (int i = 0; i < 16; ++i) nl |= (uint32_t)(at + (uint32_t)i < r1 && A.text[at + (uint32_t)i] == 0x0Au) << i;
        }
        const uint32_t incl = wave_incl_scan((uint32_t)__builtin_popcount(nl), lane);
        if (!FILL) {
            if (lane == 63) A.seg_nl[seg] = incl;
        } else {
            const int32_t T = *A.n_pos;
            uint32_t rank = A.seg_nl[seg] + incl - (uint32_t)__builtin_popcount(nl);       // newlines of the region before this lane's bytes
            for (uint32_t m = nl; m; m &= m - 1u) {
                ++rank;                                          // the line behind this newline is line `rank`
                if ((int32_t)rank <= T) A.line_start[(int64_t)b * A.line_stride + rank] = at + (uint32_t)__builtin_ctz(m) + 1u;
            }
        }
    }
}

// One wavefront per batch: the exclusive prefix of its segments' line counts (64 segments a step, coalesced), its lines, where its
// region starts; the tile's positions = the fewest lines any batch has (*A.n_pos arrives as max_pos from the counting launch).
// (Round 5: one THREAD per batch walked its ~600 segments one dependent load after the other: 0.24 ms a tile.)
__global__ __launch_bounds__(kWave) void region_scan_kernel(RegionArgs A)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= A.n_batches) return;
    const uint32_t s0 = A.seg_base[b], s1 = A.seg_base[b + 1];
    uint32_t run = 0;
    for (uint32_t s = s0; s < s1; s += kWave) {
        const uint32_t i = s + (uint32_t)lane;
        const uint32_t c = i < s1 ? A.seg_nl[i] : 0u;
        const uint32_t incl = wave_incl_scan(c, lane);
        if (i < s1) A.seg_nl[i] = run + incl - c;
        run += (uint32_t)__shfl((int)incl, kWave - 1);
    }
    if (lane == 0) {
        A.lines[b] = (int32_t)run;
        A.line_start[(int64_t)b * A.line_stride] = A.regions[b].start;   // (also of a batch whose region is empty: it has no segment to do it)
        atomicMin(A.n_pos, (int32_t)run);
    }
}

// what the tile before left of every batch, to the front of the batch's region in the other text buffer
__global__ void region_carry_kernel(const uint8_t *__restrict__ old_text, uint8_t *__restrict__ text, const bvc_pileup_region *__restrict__ regions)
{
    const bvc_pileup_region r = regions[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < r.left_len; i += blockDim.x) text[r.start + i] = old_text[r.left_src + i];
}

// the text of the indel tokens, gathered for the caller (tiles whose text never was on the host): any order; each record's text_off
// becomes its offset in `dst`
__global__ void indel_text_kernel(const uint8_t *__restrict__ text, bvc_pileup_indel *__restrict__ indels, const uint32_t *__restrict__ n_written,
                                  uint32_t cap_records, uint8_t *__restrict__ dst, uint32_t dst_cap, uint32_t *__restrict__ used)
{
    const uint32_t n = *n_written < cap_records ? *n_written : cap_records;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const uint32_t len = (uint32_t)indels[k].len, from = (uint32_t)indels[k].text_off;
        const uint32_t at = atomicAdd(used, len);
        if (at + len <= dst_cap)
            for (uint32_t i = 0; i < len; ++i) dst[at + i] = text[from + i];
        indels[k].text_off = (int64_t)at;
    }
}

// ---- the entries of the CALLED positions only (what WriteVcf reads, src/BaseVarC.cpp:664; the CVG line of every position needs the
// tallies and the indel records alone): called_off[t] = entries of the called positions before t, called_off[n_pos] = their total
__global__ __launch_bounds__(1024) void called_scan_kernel(int64_t n_pos, const bvc_site_result *__restrict__ results,
                                                           const int64_t *__restrict__ entry_off, int64_t *__restrict__ called_off)
{
    __shared__ int64_t wave_sum[16];
    __shared__ int64_t running;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) running = 0;
    __syncthreads();
    for (int64_t t0 = 0; t0 < n_pos; t0 += 1024) {
        const int64_t t = t0 + threadIdx.x;
        const int64_t mine = (t < n_pos && results[t].called) ? entry_off[t + 1] - entry_off[t] : 0;
        int64_t incl = mine;
        for (int d = 1; d < kWave; d <<= 1) {
            const int64_t up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        if (lane == kWave - 1) wave_sum[wave] = incl;
        __syncthreads();
        int64_t before = running;
        for (int w = 0; w < wave; ++w) before += wave_sum[w];
        if (t < n_pos) called_off[t] = before + incl - mine;
        __syncthreads();
        if (threadIdx.x == 1023) running = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) called_off[n_pos] = running;
}

__global__ void called_gather_kernel(int64_t n_pos, const int64_t *__restrict__ entry_off, const int64_t *__restrict__ called_off,
                                     const bvc_pileup_entry *__restrict__ entries, const int32_t *__restrict__ samples,
                                     bvc_pileup_entry *__restrict__ out_entries, int32_t *__restrict__ out_samples)
{
    for (int64_t t = blockIdx.x; t < n_pos; t += gridDim.x) {
        const int64_t at = called_off[t], n = called_off[t + 1] - at, from = entry_off[t];
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
            out_entries[at + i] = entries[from + i];
            out_samples[at + i] = samples[from + i];
        }
    }
}

hipError_t launch_called_scan(hipStream_t stream, const PileupTile &P, const bvc_site_result *results, int64_t *called_off)
{
    hipLaunchKernelGGL(called_scan_kernel, dim3(1), dim3(1024), 0, stream, (int64_t)P.n_pos, results, P.entry_off, called_off);
    return hipGetLastError();
}

hipError_t launch_called_gather(hipStream_t stream, const PileupTile &P, const int64_t *called_off, bvc_pileup_entry *out_entries,
                                int32_t *out_samples)
{
    if (P.n_pos <= 0) return hipSuccess;
    hipLaunchKernelGGL(called_gather_kernel, dim3((unsigned)(P.n_pos < 16384 ? P.n_pos : 16384)), dim3(256), 0, stream, (int64_t)P.n_pos,
                       P.entry_off, called_off, P.entries, P.samples, out_entries, out_samples);
    return hipGetLastError();
}

hipError_t launch_region_carry(hipStream_t stream, const uint8_t *old_text, uint8_t *text, const bvc_pileup_region *regions, int32_t n_batches)
{
    if (n_batches <= 0) return hipSuccess;
    hipLaunchKernelGGL(region_carry_kernel, dim3((unsigned)n_batches), dim3(256), 0, stream, old_text, text, regions);
    return hipGetLastError();
}

hipError_t launch_region_index(hipStream_t stream, const PileupTile &P, const bvc_pileup_region *regions, const uint32_t *seg_base,
                               int64_t n_segments, uint32_t *seg_nl, int32_t *lines, int32_t max_pos)
{
    if (P.n_batches <= 0) return hipSuccess;
    RegionArgs A{};
    A.text = P.text; A.regions = regions; A.seg_base = seg_base; A.n_batches = P.n_batches; A.max_pos = max_pos; A.line_stride = P.line_stride;
    A.seg_nl = seg_nl; A.lines = lines; A.n_pos = const_cast<int32_t *>(P.n_pos_dev); A.line_start = const_cast<uint32_t *>(P.line_start);
    const int64_t blocks = (n_segments + kParseWaves - 1) / kParseWaves;
    const unsigned grid = (unsigned)(blocks < 65536 ? (blocks > 0 ? blocks : 1) : 65536);
    hipLaunchKernelGGL(region_lines_kernel<false>, dim3(grid), dim3(kParseWaves * kWave), 0, stream, A);
    hipLaunchKernelGGL(region_scan_kernel, dim3((unsigned)P.n_batches), dim3(kWave), 0, stream, A);
    hipLaunchKernelGGL(region_lines_kernel<true>, dim3(grid), dim3(kParseWaves * kWave), 0, stream, A);
    return hipGetLastError();
}

// ends[b] = where the tile's last line of batch b ends (the start of what the tile leaves of the batch)
__global__ void region_ends_kernel(const uint32_t *__restrict__ line_start, int32_t line_stride, const int32_t *__restrict__ n_pos, int32_t n_batches,
                                   uint32_t *__restrict__ ends)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n_batches) ends[b] = line_start[(int64_t)b * line_stride + *n_pos];
}

hipError_t launch_region_ends(hipStream_t stream, const PileupTile &P, uint32_t *ends)
{
    if (P.n_batches <= 0) return hipSuccess;
    hipLaunchKernelGGL(region_ends_kernel, dim3((unsigned)((P.n_batches + 255) / 256)), dim3(256), 0, stream, P.line_start, P.line_stride, P.n_pos_dev,
                       P.n_batches, ends);
    return hipGetLastError();
}

hipError_t launch_indel_text(hipStream_t stream, const PileupTile &P, uint8_t *dst, uint32_t dst_cap, uint32_t *used)
{
    if (P.indel_cap == 0) return hipSuccess;
    hipLaunchKernelGGL(indel_text_kernel, dim3((P.indel_cap + 255) / 256), dim3(256), 0, stream, P.text, P.indels, P.status + 2, P.indel_cap, dst,
                       dst_cap, used);
    return hipGetLastError();
}

hipError_t launch_hist_csr_groups(LaunchState &st, hipStream_t stream, int64_t n_sites, const int64_t *offsets, const int8_t *bases,
                                  const int8_t *quals, const int32_t *sample_of_obs, const uint8_t *group_of_sample, int64_t n_samples,
                                  int n_groups, uint32_t *counts)
{
    if (n_sites <= 0) return hipSuccess;
    int log2c = 0;                                               // copies: as many as fit 64 KiB
    while (log2c < 5 && ((size_t)(n_groups + 1) * BVC_NCLASS * sizeof(uint32_t) << (log2c + 1)) <= 64 * 1024) ++log2c;
    const size_t lds = (size_t)(n_groups + 1) * BVC_NCLASS * sizeof(uint32_t) << log2c;
    constexpr uint32_t kSlotCsrGroups = 61;
    if (lds > 48 * 1024 && !(st.attr_done & ((uint64_t)1 << kSlotCsrGroups))) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(hist_csr_groups_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)((size_t)(BVC_MAX_GROUPS + 1) * BVC_NCLASS * sizeof(uint32_t)));
        if (e != hipSuccess) return e;
        st.attr_done |= (uint64_t)1 << kSlotCsrGroups;
    }
    hipLaunchKernelGGL(hist_csr_groups_kernel, dim3((unsigned)(n_sites < 8192 ? n_sites : 8192)), dim3(kCsrGroupThreads), lds, stream, n_sites,
                       offsets, bases, quals, sample_of_obs, group_of_sample, n_samples, n_groups, log2c, counts);
    return hipGetLastError();
}

#ifdef BVC_CHECK_LDS
BVC_DEFINE_DEBUG_READER(debug_read_pileup)
#endif

}  // namespace bvc
