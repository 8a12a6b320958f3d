// inflate_kernel.hip -- raw DEFLATE (RFC 1951) of whole BGZF blocks on the device: one WAVEFRONT per block.
//
// Why: the temp batches of `basetype` are BGZF text (the reference's form: bgzf_write in bt_r, src/BaseVarC.cpp:509-527; read back with
// bgzf_getline in bt_s, :406).  With the token parse on the device (pileup_kernel.hip) the inflate is what the host program's CPUs
// spend their time on -- ~200 us per position of 1e5 samples, 300 us when sixteen threads share the box's quota -- and BGZF blocks are
// independent (<= 64 KiB of output each, window inside the block): thousands of them decode side by side.
//
// A block's symbol stream is serial, so the decode is not data-parallel inside a block: every lane of the wavefront follows the same
// (wave-uniform) control flow -- bit buffer, table look-ups in LDS -- and the lanes share the work that IS parallel: staging the
// compressed bytes (1 KiB per refill, 16 bytes a lane), filling the first-level decoding tables, copying matches (lane i copies byte
// i of the match; a match shorter than its distance, or a run, is the same expression: source byte i mod distance), and writing the
// output (1 KiB at a time, 16 bytes a lane).  The block's last 4 KiB of output live in an LDS ring: nearly every match of pileup text
// reaches back less than that (a line is 1-2 KiB) and is copied LDS to LDS; a match that reaches further back -- deflate allows 32 KiB
// -- reads its source from the block's output in global memory, which left the ring at least 3 KiB ago (loads behind a
// release fence: the bytes were written by this wavefront's own stores).  10 KiB of LDS per wavefront instead of the 36 KiB a whole
// window takes: sixteen blocks in flight per CU instead of four, and the decode is a chain of LDS round trips that only other
// wavefronts can hide (round 5, pileup text at 10 % coverage deflated at zlib's level 6: 3.0 ms per block
// either way; 8192 blocks in 9.7 ms = 55 GB/s of text with the 4 KiB ring, 45 with 8 KiB, 34 with 16, 22 with all 32: profiles/r05_inflate.txt).
//
// Decoding tables: 9-bit (literal/length) and 7-bit (distance, code lengths) first-level tables of 32-bit entries
// [valid | code length | kind | extra bits | value] -- a length's or distance's base and extra-bit count ride in the entry, so a
// match is two look-ups, not four; longer codes -- rare in text -- walk the canonical code one bit at a time (count per length +
// symbols sorted by code, as zlib's puff does).  Same acceptance rules as host/inflate.cpp and zlib: over-subscribed code sets are refused,
// incomplete ones too unless the set has a single one-bit code (or none), distances beyond the block's start, output beyond ISIZE
// and input beyond the block are refused.  A refused block leaves a code in status[block]; its output is undefined.  A second kernel
// (crc32_kernel, below) computes the CRC32 of every block's output where the caller asks for the comparison, as htslib does.
// Own code, written from RFC 1951; checked against zlib in tests/test_gpu_round5.py.
#include <hip/hip_runtime.h>

#ifndef BVC_INFLATE_WINDOW
#define BVC_INFLATE_WINDOW 4096
#endif

#include "bvc_device.h"
#include "bvc_internal.h"

namespace bvc {
namespace {

constexpr uint32_t kWinBytes = BVC_INFLATE_WINDOW, kWinMask = kWinBytes - 1;
static_assert((kWinBytes & kWinMask) == 0 && kWinBytes >= 4096 && kWinBytes <= 32768, "the LDS ring: a power of two, 4..32 KiB");
constexpr uint32_t kStageWords = 256;
constexpr int kLitBits = 9, kDistBits = 7, kPreBits = 7;
constexpr uint32_t kValid = 0x80000000u;
// entry: bit 31 valid, 27..30 code length, 24..25 kind, 16..19 extra bits, 0..15 value (literal / length base / distance base / symbol)
constexpr uint32_t kKindLiteral = 0u << 24, kKindEnd = 1u << 24, kKindLength = 2u << 24, kKindBad = 3u << 24, kKindMask = 3u << 24;

__device__ const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
                                           8193, 12289, 16385, 24577};
__device__ const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ const uint8_t kPreOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct InflateLds {
    uint32_t win[kWinBytes / 4];
    uint32_t stage[kStageWords];
    uint32_t lit_tab[1 << kLitBits];
    uint32_t dist_tab[1 << kDistBits];
    uint32_t pre_tab[1 << kPreBits];
    uint16_t lit_sorted[288], dist_sorted[32], pre_sorted[32];
    uint16_t lit_cnt[16], dist_cnt[16], pre_cnt[16];
    uint32_t work_cnt[16], work_next[16], work_offs[16];
    uint32_t len_code[32], dist_code[32];       // base | extra bits << 16 of the length / distance symbols (a global load per match otherwise)
    uint8_t lens[320 + 8];
};

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// error codes in status[]
enum : uint32_t { kOk = 0, kErrType = 1, kErrStored = 2, kErrHeader = 3, kErrCodes = 4, kErrSymbol = 5, kErrDistance = 6, kErrOutput = 7,
                  kErrInput = 8, kErrSize = 9, kErrCrc = 10 };

// Canonical code of `n` symbols from their code lengths lens[0..n): first-level table of `bits` bits, count per length, symbols sorted
// by code.  Wave-uniform; the lanes share the table fill.  false: a code set zlib refuses.
template <class Entry>
__device__ bool build_code(InflateLds &L, const uint8_t *lens, int n, uint32_t *tab, int bits, uint16_t *cnt, uint16_t *sorted, bool may_be_short,
                           int lane, Entry entry_of)
{
    if (lane < 16) L.work_cnt[lane] = 0u;
    for (int i = lane; i < (1 << bits); i += kWave) tab[i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int s = lane; s < n; s += kWave) {
        const uint32_t l = lens[s];
        if (l) atomicAdd(&L.work_cnt[l & 15u], 1u);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    int left = 1, longest = 0;
    uint32_t code = 0, at = 0;
    bool ok = true;
    for (int l = 1; l <= 15; ++l) {
        const uint32_t c = uni(L.work_cnt[l]);
        left = (left << 1) - (int)c;
        if (left < 0) ok = false;                                // over-subscribed
        if (c) longest = l;
        code = (code + (l > 1 ? uni(L.work_cnt[l - 1]) : 0u)) << 1;
        if (lane == 0) { L.work_next[l] = code; L.work_offs[l] = at; cnt[l] = (uint16_t)c; }
        at += c;
    }
    if (lane == 0) cnt[0] = 0;
    if (left > 0 && !(may_be_short && longest <= 1)) ok = false;  // incomplete (zlib: "incomplete set")
    if (!ok) return false;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int s = 0; s < n; ++s) {                                 // codes are handed out in symbol order
        const uint32_t l = uni(lens[s]);
        if (l == 0u) continue;
        const uint32_t c = uni(L.work_next[l]), k = uni(L.work_offs[l]);
        if (lane == 0) { L.work_next[l] = c + 1u; L.work_offs[l] = k + 1u; sorted[k] = (uint16_t)s; }
        if ((int)l <= bits) {
            const uint32_t r = __builtin_bitreverse32(c) >> (32u - l);
            const uint32_t e = kValid | (l << 27) | entry_of((uint32_t)s);
            for (uint32_t i = r + ((uint32_t)lane << l); i < (1u << bits); i += (uint32_t)kWave << l) tab[i] = e;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    return true;
}

__global__ __launch_bounds__(kWave) void inflate_kernel(const uint8_t *__restrict__ comp, const bvc_bgzf_block *__restrict__ blocks, int64_t n_blocks,
                                                        uint8_t *__restrict__ out, uint32_t *__restrict__ status)
{
    BVC_POISON_LDS();
    __shared__ InflateLds L;
    const int lane = threadIdx.x;
    uint8_t *win8 = reinterpret_cast<uint8_t *>(L.win);
    if (lane < 29) L.len_code[lane] = (uint32_t)kLenBase[lane] | ((uint32_t)kLenExtra[lane] << 16);
    if (lane < 30) L.dist_code[lane] = (uint32_t)kDistBase[lane] | ((uint32_t)kDistExtra[lane] << 16);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const uint64_t c0 = (uint64_t)blocks[blk].comp_off, o0 = (uint64_t)blocks[blk].out_off;
        const uint32_t clen = (uint32_t)blocks[blk].comp_len, isize = (uint32_t)blocks[blk].isize;
        const uint8_t *cbase = comp + (c0 & ~(uint64_t)3);
        const uint32_t lead = (uint32_t)(c0 & 3u);
        const uint32_t n_words = (lead + clen + 3u) >> 2;
        uint32_t next_word = 0, stage_base = 0x80000000u;       // (nothing staged yet: any first index is 'outside')
        uint64_t bb = 0;
        int bc = 0;
        uint32_t o = 0, flushed = 0, err = kOk;

        auto get_word = [&](uint32_t i) -> uint32_t {
            if (i - stage_base >= kStageWords) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
                stage_base = i;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t w = i + (uint32_t)(k * kWave + lane);
                    L.stage[k * kWave + lane] = w < n_words ? reinterpret_cast<const uint32_t *>(cbase)[w] : 0u;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
            }
            return uni(L.stage[i - stage_base]);
        };
        auto refill = [&]() {
            while (bc <= 32) { bb |= (uint64_t)get_word(next_word) << bc; ++next_word; bc += 32; }
        };
        auto take = [&](int n) -> uint32_t {                       // n <= 32, bits present
            const uint32_t v = (uint32_t)(bb & ((1ull << n) - 1ull));
            bb >>= n; bc -= n;
            return v;
        };
        // what an entry says beside the code length, by symbol
        auto lit_entry = [&](uint32_t s) -> uint32_t {
            if (s < 256u) return kKindLiteral | s;
            if (s == 256u) return kKindEnd;
            if (s > 285u) return kKindBad;
            const uint32_t lc = L.len_code[s - 257u];
            return kKindLength | (lc & 0xFFFFu) | ((lc >> 16) << 16);
        };
        auto dist_entry = [&](uint32_t s) -> uint32_t {
            if (s > 29u) return kKindBad;
            const uint32_t dc = L.dist_code[s];
            return (dc & 0xFFFFu) | ((dc >> 16) << 16);
        };
        auto pre_entry = [&](uint32_t s) -> uint32_t { return s; };
        // the next symbol's entry (valid bit set), or 0 when the bits are no code of the set
        auto decode = [&](const uint32_t *tab, int bits, const uint16_t *cnt, const uint16_t *sorted, auto entry_of) -> uint32_t {
            const uint32_t e = uni(tab[(uint32_t)bb & ((1u << bits) - 1u)]);
            if (e & kValid) { const int l = (int)((e >> 27) & 15u); bb >>= l; bc -= l; return e; }
            int code = 0, first = 0, index = 0;
            for (int l = 1; l <= 15; ++l) {
                code |= (int)(bb & 1ull); bb >>= 1; bc -= 1;
                const int count = (int)uni(cnt[l]);
                if (code - count < first) return kValid | uni(entry_of((uint32_t)uni(sorted[index + (code - first)])));
                index += count; first += count; first <<= 1; code <<= 1;
            }
            return 0u;
        };
        auto flush_full = [&]() {
            while (o - flushed >= 1024u) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
                const uint32_t at = flushed + 16u * (uint32_t)lane;
                const uint32_t *src = &L.win[(at & kWinMask) >> 2];
                uint32_t v[4] = {src[0], src[1], src[2], src[3]};
                __builtin_memcpy(out + o0 + at, v, 16);
                flushed += 1024u;
            }
        };

        refill();
        if (lead) take((int)(8u * lead));
        bool last = false;
        while (!last && err == kOk) {
            refill();
            last = take(1) != 0u;
            const uint32_t type = take(2);
            if (type == 0u) {                                    // stored
                take(bc & 7);
                refill();
                const uint32_t len = take(16);
                refill();
                const uint32_t nlen = take(16);
                if ((len ^ 0xFFFFu) != nlen) { err = kErrStored; break; }
                if (o + len > isize) { err = kErrOutput; break; }
                for (uint32_t i = 0; i < len; ++i) {
                    refill();
                    const uint32_t c = take(8);
                    if (lane == 0) win8[o & kWinMask] = (uint8_t)c;
                    ++o;
                    if ((o & 1023u) == 0u) flush_full();
                }
                // (the bytes of a stored block must have come from inside the BGZF block too)
                if ((int64_t)next_word * 32 - bc - 8 * (int64_t)lead > (int64_t)clen * 8) err = kErrInput;
                continue;
            }
            if (type == 3u) { err = kErrType; break; }
            int hlit = 288, hdist = 30;
            if (type == 1u) {                                    // fixed codes
                for (int s = lane; s < 288; s += kWave) L.lens[s] = (uint8_t)(s < 144 ? 8 : (s < 256 ? 9 : (s < 280 ? 7 : 8)));
                for (int s = lane; s < 30; s += kWave) L.lens[288 + s] = 5;
                if (lane < 2) L.lens[288 + 30 + lane] = 5;        // (32 five-bit distance codes: the set is complete)
                hdist = 32;
            } else {                                             // dynamic codes
                hlit = (int)take(5) + 257; hdist = (int)take(5) + 1;
                const int hclen = (int)take(4) + 4;
                if (hlit > 286 || hdist > 30) { err = kErrHeader; break; }
                if (lane < 19) L.lens[lane] = 0;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
                for (int i = 0; i < hclen; ++i) {
                    refill();
                    const uint32_t v = take(3);
                    if (lane == 0) L.lens[kPreOrder[i]] = (uint8_t)v;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
                if (!build_code(L, L.lens, 19, L.pre_tab, kPreBits, L.pre_cnt, L.pre_sorted, false, lane, pre_entry)) { err = kErrCodes; break; }
                int n = 0;
                uint32_t prev = 0;
                while (n < hlit + hdist && err == kOk) {
                    refill();
                    const uint32_t pe = decode(L.pre_tab, kPreBits, L.pre_cnt, L.pre_sorted, pre_entry);
                    const int sym = (int)(pe & 0xFFFFu);
                    if (pe == 0u || sym > 18) { err = kErrHeader; break; }
                    if (sym < 16) { if (lane == 0) L.lens[n] = (uint8_t)sym; prev = (uint32_t)sym; ++n; continue; }
                    int rep;
                    uint32_t v = 0;
                    if (sym == 16) { if (n == 0) { err = kErrHeader; break; } v = prev; rep = 3 + (int)take(2); }
                    else if (sym == 17) rep = 3 + (int)take(3);
                    else rep = 11 + (int)take(7);
                    if (n + rep > hlit + hdist) { err = kErrHeader; break; }
                    for (int i = lane; i < rep; i += kWave) L.lens[n + i] = (uint8_t)v;
                    n += rep;
                    prev = v;
                }
                if (err != kOk) break;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
                if (uni(L.lens[256]) == 0u) { err = kErrCodes; break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            // (the code-length table shares L.lens' front with nothing: the literal/length lengths start at 0 only now)
            if (!build_code(L, L.lens, hlit, L.lit_tab, kLitBits, L.lit_cnt, L.lit_sorted, true, lane, lit_entry) ||
                !build_code(L, L.lens + hlit, hdist, L.dist_tab, kDistBits, L.dist_cnt, L.dist_sorted, true, lane, dist_entry)) { err = kErrCodes; break; }
            // ---- the block's symbols
            for (;;) {
                refill();
                const uint32_t e = decode(L.lit_tab, kLitBits, L.lit_cnt, L.lit_sorted, lit_entry);
                if (e == 0u) { err = kErrSymbol; break; }
                const uint32_t kind = e & kKindMask;
                if (kind == kKindLiteral) {
                    if (o >= isize) { err = kErrOutput; break; }
                    if (lane == 0) win8[o & kWinMask] = (uint8_t)e;
                    ++o;
                    if ((o & 1023u) == 0u) flush_full();
                    continue;
                }
                if (kind == kKindEnd) break;
                if (kind == kKindBad) { err = kErrSymbol; break; }
                const uint32_t len = (e & 0xFFFFu) + take((int)((e >> 16) & 15u));
                refill();
                const uint32_t d = decode(L.dist_tab, kDistBits, L.dist_cnt, L.dist_sorted, dist_entry);
                if (d == 0u || (d & kKindMask)) { err = kErrSymbol; break; }
                const uint32_t dist = (d & 0xFFFFu) + take((int)((d >> 16) & 15u));
                if (dist > o) { err = kErrDistance; break; }
                if (o + len > isize) { err = kErrOutput; break; }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
                if (dist + len + 64u <= kWinBytes) {
                    // source and destination are in the ring (and stay there while the match is written)
                    const bool pow2 = (dist & (dist - 1u)) == 0u;
                    for (uint32_t i = (uint32_t)lane; i < len; i += kWave) {
                        const uint32_t k = dist >= len ? i : (pow2 ? (i & (dist - 1u)) : i % dist);
                        win8[(o + i) & kWinMask] = win8[(o - dist + k) & kWinMask];
                    }
                } else {
                    // the source has left the ring: it is in the block's output in global memory, written by this wavefront's flushes at
                    // least kWinBytes - 2 * 258 - 64 bytes of output ago (dist > len here: no overlap with the destination)
                    // (workgroup scope: the stores and the loads are this wavefront's own and go through the same L1 and L2 -- the fence is
                    // a wait for the stores in flight; at agent scope it is a write-back of the XCD's L2 and the loads bypass it: 5 x slower)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    const uint8_t *src = out + o0 + (o - dist);
                    for (uint32_t i = (uint32_t)lane; i < len; i += kWave)
                        win8[(o + i) & kWinMask] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                o += len;
                if (o - flushed >= 1024u) flush_full();
            }
            // the input of this deflate block must have come from inside the BGZF block
            if (err == kOk && (int64_t)next_word * 32 - bc - 8 * (int64_t)lead > (int64_t)clen * 8) err = kErrInput;
        }
        if (err == kOk && o != isize) err = kErrSize;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (err == kOk)
            for (uint32_t i = flushed + (uint32_t)lane; i < o; i += kWave) out[o0 + i] = win8[i & kWinMask];
        if (lane == 0) status[blk] = err;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
}


// ---- CRC32 of the inflated blocks (RFC 1952; htslib checks it for every BGZF block it reads) ---------------------------------
// One wavefront per block: every lane takes a slice of the block's output (slices cut at absolute 16-byte boundaries: aligned 16-byte
// loads), runs the byte-wise table CRC over it, and the 64 slice CRCs are combined in order -- crc(A || B) = crc(A) * x^(8 |B|) mod P
// xor crc(B) in GF(2)[x] (zlib's crc32_combine; the powers x^(2^k) mod P are compile-time constants).
constexpr uint32_t kCrcPoly = 0xEDB88320u;
constexpr uint32_t crc_multmodp(uint32_t a, uint32_t b)
{
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) { p ^= b; if ((a & (m - 1u)) == 0u) break; }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ kCrcPoly : b >> 1;
    }
    return p;
}
struct CrcPowers { uint32_t v[32]; };
constexpr CrcPowers crc_powers()
{
    CrcPowers t{};
    uint32_t p = 1u << 30;                                       // x^1
    t.v[0] = p;
    for (int i = 1; i < 32; ++i) { p = crc_multmodp(p, p); t.v[i] = p; }
    return t;
}
__device__ const CrcPowers kCrcPowers = crc_powers();

// x^(8 n) mod P
__device__ uint32_t crc_shift_op(uint32_t n)
{
    uint32_t p = 1u << 31;
    int k = 3;
    while (n) { if (n & 1u) p = crc_multmodp(kCrcPowers.v[k & 31], p); n >>= 1; ++k; }
    return p;
}

__global__ __launch_bounds__(kWave) void crc32_kernel(const bvc_bgzf_block *__restrict__ blocks, int64_t n_blocks, const uint8_t *__restrict__ out,
                                                      uint32_t *__restrict__ status)
{
    BVC_POISON_LDS();
    __shared__ uint32_t tab[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += kWave) {
        uint32_t c = (uint32_t)i;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? kCrcPoly ^ (c >> 1) : c >> 1;
        tab[i] = c;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        if (blocks[blk].check_crc == 0u || status[blk] != 0u) continue;
        const uint64_t o0 = (uint64_t)blocks[blk].out_off;
        const uint32_t n = (uint32_t)blocks[blk].isize;
        // slice boundaries: o0 rounded down to 16, then every `per` bytes (a multiple of 16), clamped to the block
        const uint64_t base = o0 & ~(uint64_t)15;
        const uint32_t span = (uint32_t)(o0 - base) + n;
        const uint32_t per = ((span + kWave - 1) / kWave + 15u) & ~15u;
        const uint64_t lo64 = base + (uint64_t)per * (uint32_t)lane, hi64 = lo64 + per;
        const uint64_t lo = lo64 < o0 ? o0 : (lo64 > o0 + n ? o0 + n : lo64), hi = hi64 < o0 ? o0 : (hi64 > o0 + n ? o0 + n : hi64);
        uint32_t c = 0xFFFFFFFFu;
        uint64_t at = lo;
        while (at < hi && (at & 15u)) { c = tab[(c ^ out[at]) & 255u] ^ (c >> 8); ++at; }
        for (; at + 16 <= hi; at += 16) {
            const uint4 v = *reinterpret_cast<const uint4 *>(out + at);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int b = 0; b < 4; ++b) c = tab[(c ^ (w[j] >> (8 * b))) & 255u] ^ (c >> 8);
        }
        for (; at < hi; ++at) c = tab[(c ^ out[at]) & 255u] ^ (c >> 8);
        const uint32_t mine = ~c;                                // the slice's CRC32 (0 for an empty slice)
        const uint32_t len = (uint32_t)(hi - lo);
        // in order: acc = acc * x^(8 len_k) xor crc_k; the slices behind the first are `per` bytes but the last (and the empty ones)
        const uint32_t op_full = crc_shift_op(per);
        uint32_t acc = (uint32_t)__builtin_amdgcn_readlane((int)mine, 0);
        for (int k = 1; k < kWave; ++k) {
            const uint32_t lk = (uint32_t)__shfl((int)len, k, kWave), ck = (uint32_t)__shfl((int)mine, k, kWave);
            if (lk == 0u) continue;
            acc = crc_multmodp(lk == per ? op_full : crc_shift_op(lk), acc) ^ ck;
        }
        if (lane == 0 && acc != blocks[blk].crc32) status[blk] = kErrCrc;
    }
}

}  // namespace

hipError_t launch_inflate(hipStream_t stream, const uint8_t *comp, const bvc_bgzf_block *blocks, int64_t n_blocks, uint8_t *out, uint32_t *status)
{
    if (n_blocks <= 0) return hipSuccess;
    hipLaunchKernelGGL(inflate_kernel, dim3((unsigned)(n_blocks < 65536 ? n_blocks : 65536)), dim3(kWave), 0, stream, comp, blocks, n_blocks, out, status);
    // (blocks with check_crc: the CRC32 of what was just written against the trailer's)
    hipLaunchKernelGGL(crc32_kernel, dim3((unsigned)(n_blocks < 65536 ? n_blocks : 65536)), dim3(kWave), 0, stream, blocks, n_blocks, out, status);
    return hipGetLastError();
}

#ifdef BVC_CHECK_LDS
BVC_DEFINE_DEBUG_READER(debug_read_inflate)
#endif

}  // namespace bvc
