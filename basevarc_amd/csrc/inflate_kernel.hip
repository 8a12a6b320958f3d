// inflate_kernel.hip -- raw DEFLATE (RFC 1951) of whole BGZF blocks on the device: one WAVEFRONT per block.
//
// Why: the temp batches of `basetype` are BGZF text (the reference's form: bgzf_write in bt_r, src/BaseVarC.cpp:509-527; read back with
// bgzf_getline in bt_s, :406).  With the token parse on the device (pileup_kernel.hip) the inflate is what the host program's CPUs
// spend their time on -- ~200 us per position of 1e5 samples, 300 us when sixteen threads share the box's quota -- and BGZF blocks are
// independent (<= 64 KiB of output each, window inside the block): thousands of them decode side by side.
//
// A block's symbol stream is serial, so the decode is not data-parallel inside a block, and a wavefront alone on its SIMD issues one
// instruction in four cycles: a block's time is the instruction count of its symbol loop (profiles/r05_inflate.txt: 3.3 ms a block
// with the compiler's loop at ~130 instructions and ~26 branches a symbol, 1.9 ms with this one).  What the lanes share:
//  * the compressed bytes: the next 64 words of the block live in ONE register (lane j: word j), the bit position is wave-uniform,
//    a word is a v_readlane away; no staging buffer, no round trip to read bits;
//  * the table look-ups: lane j looks up the 32 bits that start j bits ahead in BOTH first-level tables (one LDS round trip for the
//    symbols of the next ~64 bits) and the serial walk from symbol to symbol reads those entries with v_readlane -- hand-written
//    (fast_symbols: ~10 instructions a literal, ~55 a match), since the compiler turns every uniform branch of this kernel into
//    mask arithmetic;
//  * the copies: literals are collected in a lane mask and written together (lane j: its entry's byte at its rank); lane i copies
//    byte i of a match, its LDS read left in flight while the next symbols are decoded; a match shorter than its distance, or a
//    run, is the same expression (source byte i mod distance) in the general path;
//  * the output: 1 KiB at a time, 16 bytes a lane, from an LDS ring of the block's last BVC_INFLATE_WINDOW bytes (4 KiB).  A match
//    that reaches further back -- deflate allows 32 KiB, and pileup text at zlib's level 6 uses all of it: 55 % of its matches reach
//    beyond 4 KiB -- reads its source from the block's output in global memory, which left the ring at least 3 KiB ago (this
//    wavefront's own stores, through the same L1 and L2); that load is left in flight like the LDS read of a near match.  9 KiB of
//    LDS per wavefront instead of the 37 KiB a whole window takes: eighteen blocks in flight per CU instead of four.  Round 5,
//    pileup text at 10 % coverage, zlib level 6: a block alone takes 1.9 ms with any ring; 8192 blocks 6.1 ms = 87 GB/s of text with
//    the 4 KiB ring, 72 / 54 / 38 with 8 / 16 / 32 KiB (profiles/r05_inflate.txt; the round's first kernel: 3.2 ms and 50).
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
constexpr uint32_t kDump = kWinBytes;                          // byte index of the dump area in InflateLds::win
constexpr bool kWholeWindow = kWinBytes == 32768;              // every distance deflate allows is inside the ring
static_assert((kWinBytes & kWinMask) == 0 && kWinBytes >= 4096 && kWinBytes <= 32768, "the LDS ring: a power of two, 4..32 KiB");
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
    uint32_t win[kWinBytes / 4 + kWave / 4];     // the ring, and behind it the bytes a lane with nothing to write writes to (kDump)
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
// a or b without a branch: both are computed (the compiler would otherwise sink `a`'s arithmetic into a lane-dependent branch)
__device__ __forceinline__ uint32_t pick(bool c, uint32_t a, uint32_t b)
{
    asm volatile("" : "+v"(a), "+v"(b));
    return c ? a : b;
}
// lane `k` (wave-uniform) of v
__device__ __forceinline__ uint32_t rl(uint32_t v, uint32_t k) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)k); }

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

// ---- the fast path of the symbol loop, hand-written --------------------------------------------------------------------------
// A wavefront alone on its SIMD issues one instruction in four cycles and pays more for every taken branch; the symbol loop is a
// serial chain of scalar steps, so its time IS its instruction count plus the LDS round trips it waits for (round 5: the compiler's
// loop ran ~130 instructions and ~26 branches per symbol -- uniform branches structurised into mask moves because other parts of the
// kernel branch on the lane -- and a block took 3.3 ms whatever the LDS latency was).  The same walk in ~10 instructions a literal
// and ~55 a match:
//   cw0              lane j: word wbase + j of the block (the bit reader's window, see the kernel); bp: the batch's first bit in it
//   le / de / view   lane j: the literal/length entry, the distance entry and the 32 bits of the stream that start j bits after bp
//   off              bits of the batch consumed (the next symbol starts at lane `off`); the stream is at bp + off
//   o                bytes of output so far
// Literals are collected in a lane mask and written together (EXEC = the mask: lane j writes its entry's byte at its rank).  A match
// with len <= 63, dist >= len, dist <= o, o + len <= isize is copied here, lane i byte i, its read (LDS; the block's output in global
// memory when the source has left a ring smaller than the window) left IN
// FLIGHT while the next symbols are decoded: the write follows when the next match (or the way out) asks for it -- LDS operations of
// a wavefront are performed in order, so a later read sees it.  A batch that is used up is followed by the next one (two cross-lane
// reads of cw0 and the two table look-ups) without leaving.  The way out, `what`: 1 a match decoded and not copied (len, dist), 2 a
// symbol the fast path does not take (a code longer than the first-level table, end of block, a bad symbol: at `off`, not
// consumed), 3 a KiB of output is ready to leave the ring, 4 the reader's window must move on (bp >= 1920, off = 0), 7 literals
// beyond ISIZE.  The ring is kWinBytes at LDS address 0, the tables where InflateLds has them.
// Temporaries with fixed names: s[84:85] (the literal mask: its halves are named); everything else is the compiler's choice.
constexpr uint32_t kLitTabOff = (kWinBytes / 4 + kWave / 4) * 4, kDistTabOff = kLitTabOff + (4u << kLitBits);
__device__ __forceinline__ void fast_symbols(uint32_t cw0, uint32_t lane, uint32_t isize, uint32_t flushed, const uint8_t *block_out,
                                             uint32_t &le, uint32_t &de, uint32_t &view, uint32_t &bp, uint32_t &off, uint32_t &o,
                                             uint32_t &what, uint32_t &len, uint32_t &dist)
{
    uint32_t e, l, t, p, xb, x, msk, d, pl, va, vb, vc, vs, vt;
    uint64_t m64;
    const uint32_t mask = kWinMask;
    // (EXEC is all ones in this kernel: it is set back to -1, not saved)
#define BVC_EMIT_LITERALS(done)                                                             \
    "s_cmp_eq_u64 s[84:85], 0\n"                                                            \
    "s_cbranch_scc1 " done "\n"                                                             \
    "s_bcnt1_i32_b64 %[x], s[84:85]\n"                                                      \
    "s_add_u32 %[t], %[o], %[x]\n"                                                          \
    "s_cmp_gt_u32 %[t], %[isize]\n"                                                         \
    "s_cbranch_scc1 9f\n"                                                                   \
    "v_mbcnt_lo_u32_b32 %[vc], s84, 0\n"                                                    \
    "v_mbcnt_hi_u32_b32 %[vc], s85, %[vc]\n"                                                \
    "v_add_u32 %[vc], %[o], %[vc]\n"                                                        \
    "v_and_b32 %[vc], %[mask], %[vc]\n"                                                     \
    "s_mov_b64 exec, s[84:85]\n"                                                            \
    "ds_write_b8 %[vc], %[le]\n"                                                            \
    "s_mov_b64 exec, -1\n"                                                                  \
    "s_mov_b32 %[o], %[t]\n"                                                                \
    "s_mov_b64 s[84:85], 0\n"
    // the write of the copy whose read is in flight
#define BVC_FINISH_COPY(done)                                                               \
    "s_cmp_eq_u32 %[pl], 0\n"                                                               \
    "s_cbranch_scc1 " done "\n"                                                             \
    "s_bfm_b64 exec, %[pl], 0\n"                                                            \
    "s_waitcnt vmcnt(0) lgkmcnt(0)\n"                                                       \
    "ds_write_b8 %[vb], %[va]\n"                                                            \
    "s_mov_b64 exec, -1\n"                                                                  \
    "s_mov_b32 %[pl], 0\n"
    // a ring smaller than the window: a match whose source has left it reads the block's own output in global memory, written by this
    // wavefront's flushes at least kWinBytes - 1.2 KiB of output ago (the wait: for those stores, should one still be on its way;
    // one wavefront's stores and loads to the same addresses go through the same L1 and L2, in order); the load is left in flight
    // like the LDS read of a near match
#if BVC_INFLATE_WINDOW == 32768
#define BVC_NEAR_CHECK ""
#define BVC_FAR_COPY ""
#else
#define BVC_NEAR_CHECK "s_add_u32 %[x], %[dist], %[len]\n s_cmp_gt_u32 %[x], %[lim]\n s_cbranch_scc1 20f\n"
#define BVC_FAR_COPY                                                                        \
    "20:\n"                                                                                 \
    "s_sub_u32 %[x], %[o], %[dist]\n"                                                       \
    "v_add_u32 %[va], %[x], %[lane]\n"                                                      \
    "v_add_u32 %[vb], %[o], %[lane]\n"                                                      \
    "v_and_b32 %[vb], %[mask], %[vb]\n"                                                     \
    "s_waitcnt vmcnt(0)\n"                                                                  \
    "s_bfm_b64 exec, %[len], 0\n"                                                           \
    "global_load_ubyte %[va], %[va], %[outb] sc0\n"                                           \
    "s_mov_b64 exec, -1\n"                                                                  \
    "s_mov_b32 %[pl], %[len]\n"                                                             \
    "s_mov_b32 %[o], %[t]\n"                                                                \
    "s_sub_u32 %[x], %[o], %[flushed]\n"                                                    \
    "s_cmp_ge_u32 %[x], 0x400\n"                                                            \
    "s_cbranch_scc0 1b\n"                                                                   \
    "s_mov_b32 %[what], 3\n"                                                                \
    "s_branch 6f\n"
#endif
    asm volatile(
        "s_mov_b64 s[84:85], 0\n"
        "s_mov_b32 %[pl], 0\n"
        "1:\n"
        "s_cmp_gt_u32 %[off], 63\n"
        "s_cbranch_scc1 8f\n"
        "v_readlane_b32 %[e], %[le], %[off]\n"
        "s_and_b32 %[t], %[e], 0x83000000\n"
        "s_bfe_u32 %[l], %[e], 0x4001b\n"
        "s_cmp_eq_u32 %[t], 0x80000000\n"
        "s_cbranch_scc0 2f\n"
        "s_lshl_b64 %[m], 1, %[off]\n"
        "s_or_b64 s[84:85], s[84:85], %[m]\n"
        "s_add_u32 %[off], %[off], %[l]\n"
        "s_branch 1b\n"
        "2:\n"                                          // not a literal of the first-level table
        "s_cmp_eq_u32 %[t], 0x82000000\n"
        "s_cbranch_scc0 7f\n"
        // a length: its extra bits, the distance code, its extra bits.  The lanes read are off + ... in ascending order; one past 63
        // reads some other lane's entry, which nobody uses: the LAST position is checked and the symbol left to the next batch.
        "s_add_u32 %[p], %[off], %[l]\n"
        "s_bfe_u32 %[xb], %[e], 0x40010\n"
        "v_readlane_b32 %[x], %[view], %[p]\n"
        "s_bfm_b32 %[msk], %[xb], 0\n"
        "s_and_b32 %[x], %[x], %[msk]\n"
        "s_and_b32 %[len], %[e], 0xffff\n"
        "s_add_u32 %[len], %[len], %[x]\n"
        "s_add_u32 %[p], %[p], %[xb]\n"
        "v_readlane_b32 %[d], %[de], %[p]\n"
        "s_bfe_u32 %[l], %[d], 0x4001b\n"
        "s_add_u32 %[p], %[p], %[l]\n"
        "s_cmp_gt_u32 %[p], 63\n"
        "s_cbranch_scc1 8f\n"
        "s_and_b32 %[t], %[d], 0x83000000\n"
        "s_cmp_eq_u32 %[t], 0x80000000\n"
        "s_cbranch_scc0 7f\n"
        "s_bfe_u32 %[xb], %[d], 0x40010\n"
        "v_readlane_b32 %[x], %[view], %[p]\n"
        "s_bfm_b32 %[msk], %[xb], 0\n"
        "s_and_b32 %[x], %[x], %[msk]\n"
        "s_and_b32 %[dist], %[d], 0xffff\n"
        "s_add_u32 %[dist], %[dist], %[x]\n"
        "s_add_u32 %[off], %[p], %[xb]\n"               // the symbol is consumed
        BVC_EMIT_LITERALS("3f")
        "3:\n"
        BVC_FINISH_COPY("12f")
        "12:\n"
        "s_cmp_gt_u32 %[len], 63\n"                     // (whatever says no: the caller copies the match, with its own checks)
        "s_cbranch_scc1 4f\n"
        "s_cmp_lt_u32 %[dist], %[len]\n"
        "s_cbranch_scc1 4f\n"
        "s_cmp_gt_u32 %[dist], %[o]\n"
        "s_cbranch_scc1 4f\n"
        "s_add_u32 %[t], %[o], %[len]\n"
        "s_cmp_gt_u32 %[t], %[isize]\n"
        "s_cbranch_scc1 4f\n"
        BVC_NEAR_CHECK
        "s_sub_u32 %[x], %[o], %[dist]\n"               // the copy: lane i < len moves byte i; the read is left in flight
        "v_add_u32 %[va], %[x], %[lane]\n"
        "v_and_b32 %[va], %[mask], %[va]\n"
        "v_add_u32 %[vb], %[o], %[lane]\n"
        "v_and_b32 %[vb], %[mask], %[vb]\n"
        "s_bfm_b64 exec, %[len], 0\n"
        "ds_read_u8 %[va], %[va]\n"
        "s_mov_b64 exec, -1\n"
        "s_mov_b32 %[pl], %[len]\n"
        "s_mov_b32 %[o], %[t]\n"
        "s_sub_u32 %[x], %[o], %[flushed]\n"            // a KiB to flush?
        "s_cmp_ge_u32 %[x], 0x400\n"
        "s_cbranch_scc0 1b\n"
        "s_mov_b32 %[what], 3\n"
        "s_branch 6f\n"
        BVC_FAR_COPY
        "4:\n"
        "s_mov_b32 %[what], 1\n"
        "s_branch 6f\n"
        "7:\n"                                          // a symbol for the caller
        BVC_EMIT_LITERALS("5f")
        "5:\n"
        "s_mov_b32 %[what], 2\n"
        "s_branch 6f\n"
        "8:\n"                                          // the batch is used up: its literals, then the next one
        "s_mov_b32 %[what], 3\n"
        BVC_EMIT_LITERALS("13f")
        "s_sub_u32 %[x], %[o], %[flushed]\n"
        "s_cmp_ge_u32 %[x], 0x400\n"
        "s_cbranch_scc1 6f\n"
        "13:\n"
        "s_add_u32 %[bp], %[bp], %[off]\n"
        "s_mov_b32 %[off], 0\n"
        "s_mov_b32 %[what], 4\n"
        "s_cmp_ge_u32 %[bp], 1920\n"
        "s_cbranch_scc1 6f\n"
        "s_lshr_b32 %[x], %[bp], 5\n"                   // lane j: the 32 bits from bit bp + j, out of two words of the window
        "s_and_b32 %[t], %[bp], 31\n"
        "v_add_u32 %[vt], %[t], %[lane]\n"
        "v_lshrrev_b32 %[vs], 5, %[vt]\n"
        "v_add_lshl_u32 %[vs], %[vs], %[x], 2\n"
        "ds_bpermute_b32 %[vc], %[vs], %[cw0]\n"
        "ds_bpermute_b32 %[vs], %[vs], %[cw0] offset:4\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_alignbit_b32 %[view], %[vs], %[vc], %[vt]\n"
        "v_lshlrev_b32 %[vs], 2, %[view]\n"             // both first-level tables
        "v_and_b32 %[vc], 0x7fc, %[vs]\n"
        "v_and_b32 %[vs], 0x1fc, %[vs]\n"
        "ds_read_b32 %[le], %[vc] offset:%[lit_off]\n"
        "ds_read_b32 %[de], %[vs] offset:%[dist_off]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "s_branch 1b\n"
        "9:\n"
        "s_mov_b32 %[what], 7\n"
        "6:\n"
        BVC_FINISH_COPY("14f")
        "14:\n"
        : [le] "+v"(le), [de] "+v"(de), [view] "+v"(view), [bp] "+s"(bp), [off] "+s"(off), [o] "+s"(o), [what] "=&s"(what), [len] "=&s"(len),
          [dist] "=&s"(dist), [e] "=&s"(e), [l] "=&s"(l), [t] "=&s"(t), [p] "=&s"(p), [xb] "=&s"(xb), [x] "=&s"(x), [msk] "=&s"(msk),
          [d] "=&s"(d), [pl] "=&s"(pl), [va] "=&v"(va), [vb] "=&v"(vb), [vc] "=&v"(vc), [vs] "=&v"(vs), [vt] "=&v"(vt), [m] "=&s"(m64)
        : [cw0] "v"(cw0), [lane] "v"(lane), [isize] "s"(isize), [flushed] "s"(flushed), [mask] "s"(mask), [lim] "s"(kWinBytes - 64u),
          [outb] "s"(block_out),
          [lit_off] "i"(kLitTabOff), [dist_off] "i"(kDistTabOff)
        : "memory", "scc", "vcc", "s84", "s85");
#undef BVC_NEAR_CHECK
#undef BVC_FAR_COPY
#undef BVC_FINISH_COPY
#undef BVC_EMIT_LITERALS
}

__global__ __launch_bounds__(kWave) void inflate_kernel(const uint8_t *__restrict__ comp, const bvc_bgzf_block *__restrict__ blocks, int64_t n_blocks,
                                                        uint8_t *__restrict__ out, uint32_t *__restrict__ status)
{
    BVC_POISON_LDS();
    __shared__ InflateLds L;
    const int lane = threadIdx.x;
    uint8_t *win8 = reinterpret_cast<uint8_t *>(L.win);
    // (fast_symbols addresses the ring from LDS address 0: L is the kernel's only LDS object and the ring its first member)
    if (uni((uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) void *)L.win)) != 0u) {
        for (int64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) status[blk] = kErrType;
        return;
    }
    if (lane < 29) L.len_code[lane] = (uint32_t)kLenBase[lane] | ((uint32_t)kLenExtra[lane] << 16);
    if (lane < 30) L.dist_code[lane] = (uint32_t)kDistBase[lane] | ((uint32_t)kDistExtra[lane] << 16);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const uint64_t c0 = (uint64_t)blocks[blk].comp_off, o0 = (uint64_t)blocks[blk].out_off;
        const uint32_t clen = (uint32_t)blocks[blk].comp_len, isize = (uint32_t)blocks[blk].isize;
        const uint32_t *cbase32 = reinterpret_cast<const uint32_t *>(comp + (c0 & ~(uint64_t)3));
        const uint32_t lead = (uint32_t)(c0 & 3u);
        const uint32_t n_words = (lead + clen + 3u) >> 2;
        uint32_t o = 0, flushed = 0, err = kOk;

        // ---- the bit reader: the block's next 64 words live in a VGPR (lane j: word wbase + j; the 64 after them in a second one, loaded
        // ahead), the position is wave-uniform; a word is a v_readlane away, so reading bits costs no memory round trip.  When the
        // position passes word 60 the window moves on by 60 words (two cross-lane shuffles and the next load).
        uint32_t wbase = 0, bp = 8u * lead;                       // bp: bits from word wbase, < 1920 after ensure()
        if (n_words == 0u) { if (lane == 0) status[blk] = kErrInput; continue; }
        auto load_words = [&](uint32_t first) -> uint32_t {        // (no branch: see the symbol loop)
            const uint32_t w = first + (uint32_t)lane;
            const uint32_t v = cbase32[w < n_words ? w : n_words - 1u];
            return w < n_words ? v : 0u;
        };
        uint32_t cw0 = load_words(0u), cw1 = load_words(64u);
        // (the words are WAITED for where they are loaded, not where they are first read)
        auto settle = [&]() { asm volatile("" : "+v"(cw0), "+v"(cw1)); };
        settle();
        auto ensure = [&]() {
            while (bp >= 1920u) {
                const uint32_t from = ((uint32_t)lane + 60u) & 63u;
                const uint32_t a = (uint32_t)__shfl((int)cw0, (int)from), b = (uint32_t)__shfl((int)cw1, (int)from);
                cw0 = lane < 4 ? a : b;
                wbase += 60u;
                cw1 = load_words(wbase + 64u);
                bp -= 1920u;
                settle();
            }
        };
        auto seek = [&](uint64_t abs_bits) {
            wbase = (uint32_t)(abs_bits / 1920u) * 60u;
            bp = (uint32_t)(abs_bits - (uint64_t)wbase * 32u);
            cw0 = load_words(wbase); cw1 = load_words(wbase + 64u);
            settle();
        };
        auto word_at = [&](uint32_t k) -> uint32_t { return rl(cw0, k); }; // k <= 63
        auto peek = [&]() -> uint32_t {                            // the next 32 bits (bp < 2048)
            const uint32_t k = bp >> 5, b = bp & 31u;
            const uint64_t v = ((uint64_t)word_at(k + 1u) << 32) | word_at(k);
            return (uint32_t)(v >> b);
        };
        auto take = [&](uint32_t n) -> uint32_t {                  // n <= 16
            ensure();
            const uint32_t v = peek() & ((1u << n) - 1u);
            bp += n;
            return v;
        };
        auto consumed_bits = [&]() -> int64_t { return (int64_t)wbase * 32 + (int64_t)bp - 8 * (int64_t)lead; };

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
        // a code longer than its first-level table (rare in text): the canonical code one bit at a time from `view` (the bits from the
        // symbol's first); cnt_v: lane l holds the number of codes of length l.  The entry with the code's length, or 0: no code.
        auto slow = [&](uint32_t view, uint32_t cnt_v, const uint16_t *sorted, auto entry_of) -> uint32_t {
            int code = 0, first = 0, index = 0;
#pragma unroll 1
            for (int l = 1; l <= 15; ++l) {
                code |= (int)(view & 1u); view >>= 1;
                const int count = (int)rl(cnt_v, (uint32_t)l);
                if (code - count < first)
                    return kValid | ((uint32_t)l << 27) | uni(entry_of((uint32_t)uni(sorted[index + (code - first)])));
                index += count; first += count; first <<= 1; code <<= 1;
            }
            return 0u;
        };
        // one symbol at the reader's position (the code-length alphabet; everything serial)
        auto decode_serial = [&](const uint32_t *tab, int bits, uint32_t cnt_v, const uint16_t *sorted, auto entry_of) -> uint32_t {
            ensure();
            const uint32_t v = peek();
            uint32_t e = uni(tab[v & ((1u << bits) - 1u)]);
            if (!(e & kValid)) e = slow(v, cnt_v, sorted, entry_of);
            bp += (e >> 27) & 15u;
            return e;
        };

        // a full KiB of output leaves the ring (every caller adds at most 258 bytes between two checks: once is enough)
        auto flush_full = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const uint32_t at = flushed + 16u * (uint32_t)lane;
            const uint32_t *src = &L.win[(at & kWinMask) >> 2];
            uint32_t v[4] = {src[0], src[1], src[2], src[3]};
            __builtin_memcpy(out + o0 + at, v, 16);
            flushed = uni(flushed + 1024u);
        };

        bool last = false;
        while (!last && err == kOk) {
            last = take(1) != 0u;
            const uint32_t type = take(2);
            if (type == 0u) {                                    // stored: the bytes straight from the block, 64 at a time
                ensure();
                bp = (bp + 7u) & ~7u;
                const uint32_t len = take(16);
                const uint32_t nlen = take(16);
                if ((len ^ 0xFFFFu) != nlen) { err = kErrStored; break; }
                if (o + len > isize) { err = kErrOutput; break; }
                const uint64_t byte0 = ((uint64_t)wbase * 32u + bp) >> 3;
                if (byte0 + len > (uint64_t)lead + clen) { err = kErrInput; break; }   // (they must have come from inside the BGZF block too)
                const uint32_t ob = o;
                for (uint32_t i0 = 0; i0 < len; i0 += (uint32_t)kWave) {
                    const uint32_t i = i0 + (uint32_t)lane;
                    if (i < len) win8[(ob + i) & kWinMask] = reinterpret_cast<const uint8_t *>(cbase32)[byte0 + i];
                    o += len - i0 < (uint32_t)kWave ? len - i0 : (uint32_t)kWave;
                    if (o - flushed >= 1024u) flush_full();
                }
                seek((uint64_t)wbase * 32u + bp + 8ull * len);
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
                    const uint32_t v = take(3);
                    if (lane == 0) L.lens[kPreOrder[i]] = (uint8_t)v;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
                if (!build_code(L, L.lens, 19, L.pre_tab, kPreBits, L.pre_cnt, L.pre_sorted, false, lane, pre_entry)) { err = kErrCodes; break; }
                const uint32_t pre_cnt_v = lane < 16 ? (uint32_t)L.pre_cnt[lane] : 0u;
                int n = 0;
                uint32_t prev = 0;
                while (n < hlit + hdist && err == kOk) {
                    const uint32_t pe = decode_serial(L.pre_tab, kPreBits, pre_cnt_v, L.pre_sorted, pre_entry);
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
            const uint32_t lit_cnt_v = lane < 16 ? (uint32_t)L.lit_cnt[lane] : 0u, dist_cnt_v = lane < 16 ? (uint32_t)L.dist_cnt[lane] : 0u;

            // (No lane-dependent BRANCH inside this loop: a lane that has nothing to write writes behind the ring (kDump) instead.  With one in it the
            // compiler structurises the whole loop -- every uniform branch becomes a chain of mask moves and flow blocks -- and a
            // single wavefront pays four cycles an instruction and more for every taken branch.)
            // ---- the block's symbols.  One LDS round trip serves the symbols of the next ~64 bits: lane j looks up the 32 bits that
            // start j bits ahead in BOTH first-level tables, and the walk from symbol to symbol -- the serial part -- reads those
            // entries with v_readlane (a few cycles a hop instead of an LDS latency).
            uint32_t view = 0u, le = 0u, de = 0u;               // of the batch at bp (lane j: j bits further on)
            auto gather = [&]() {
                ensure();
                const uint32_t k = bp >> 5, b0 = bp & 31u;
                const uint32_t w0 = word_at(k), w1 = word_at(k + 1u), w2 = word_at(k + 2u), w3 = word_at(k + 3u);
                const uint32_t t = b0 + (uint32_t)lane, sel = t >> 5;
                const uint32_t lo = sel == 0u ? w0 : (sel == 1u ? w1 : w2), hi = sel == 0u ? w1 : (sel == 1u ? w2 : w3);
                view = __builtin_amdgcn_alignbit(hi, lo, t & 31u);
                le = L.lit_tab[view & ((1u << kLitBits) - 1u)];
                de = L.dist_tab[view & ((1u << kDistBits) - 1u)];
            };
            gather();
            uint32_t off = 0u;
            bool end_block = false;
            // fast_symbols() runs literals, simple matches and the step from batch to batch, and comes back for the rest
            while (!end_block && err == kOk) {
                uint32_t what, len, dist;
                fast_symbols(cw0, (uint32_t)lane, isize, flushed, out + o0, le, de, view, bp, off, o, what, len, dist);
                if (o - flushed >= 1024u) flush_full();
                if (what == 3u) continue;
                if (what == 4u) { gather(); continue; }          // (bp has taken `off` up: the window moves, the batch is looked up here)
                if (what == 7u) { err = kErrOutput; break; }
                if (what == 2u) {
                    // one symbol, every case
                    uint32_t e = rl(le, off);
                    if (!(e & kValid)) {
                        e = slow(rl(view, off), lit_cnt_v, L.lit_sorted, lit_entry);
                        if (e == 0u) { err = kErrSymbol; break; }
                    }
                    const uint32_t l = (e >> 27) & 15u, kind = e & kKindMask;
                    if (kind == kKindLiteral) {
                        if (o >= isize) { err = kErrOutput; break; }
                        win8[pick(lane == 0, o & kWinMask, kDump + (uint32_t)lane)] = (uint8_t)e;
                        ++o;
                        off += l;
                        continue;
                    }
                    if (kind == kKindEnd) { off += l; end_block = true; break; }
                    if (kind == kKindBad) { err = kErrSymbol; break; }
                    // a match: every read below must lie in the batch's 64 offsets, else the batch is looked up again FROM this symbol
                    // (from offset 0 a length / distance pair, at most 15 + 5 + 15 + 13 bits, always fits)
                    const uint32_t xb = (e >> 16) & 15u, p1 = off + l, p2 = p1 + xb;
                    uint32_t d = 0u, p3 = 64u;
                    if (p2 <= 63u) {
                        d = rl(de, p2);
                        if (!(d & kValid)) {
                            d = slow(rl(view, p2), dist_cnt_v, L.dist_sorted, dist_entry);
                            if (d == 0u) { err = kErrSymbol; break; }
                        }
                        p3 = p2 + ((d >> 27) & 15u);
                    }
                    if (p3 > 63u) { bp += off; off = 0u; gather(); continue; }
                    if (d & kKindMask) { err = kErrSymbol; break; }
                    const uint32_t dxb = (d >> 16) & 15u;
                    len = (e & 0xFFFFu) + (rl(view, p1) & ((1u << xb) - 1u));
                    dist = (d & 0xFFFFu) + (rl(view, p3) & ((1u << dxb) - 1u));
                    off = p3 + dxb;
                }
                // a match to copy (what == 1, or the one just decoded)
                if (dist > o) { err = kErrDistance; break; }
                if (o + len > isize) { err = kErrOutput; break; }
                if (kWholeWindow || dist + len + 64u <= kWinBytes) {
                    // source and destination are in the ring; a match shorter than its distance, or a run, is the same expression:
                    // source byte i mod distance.  (With the whole 32 KiB window in the ring byte i + (32768 - dist) of the match
                    // lands where source byte i was: a later byte of an ascending copy, every step reads before it writes.)
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    if (dist >= len) {
                        for (uint32_t i0 = 0; i0 < len; i0 += (uint32_t)kWave) {
                            const uint32_t i = i0 + (uint32_t)lane;
                            const uint8_t b = win8[(o - dist + i) & kWinMask];
                            win8[pick(i < len, (o + i) & kWinMask, kDump + (uint32_t)lane)] = b;
                        }
                    } else {                                   // a run: the last `dist` bytes over and over
                        const bool pow2 = (dist & (dist - 1u)) == 0u;
                        for (uint32_t i0 = 0; i0 < len; i0 += (uint32_t)kWave) {
                            const uint32_t i = i0 + (uint32_t)lane;
                            const uint8_t b = win8[(o - dist + (pow2 ? (i & (dist - 1u)) : i % dist)) & kWinMask];
                            win8[pick(i < len, (o + i) & kWinMask, kDump + (uint32_t)lane)] = b;
                        }
                    }
                } else {
                    // (rings smaller than the window only) the source has left the ring: it is in the block's output in global memory,
                    // written by this wavefront's flushes at least kWinBytes - 2 * 258 - 64 bytes of output ago (dist > len here: no
                    // overlap with the destination).  Workgroup scope: the stores and the loads are this wavefront's own and go
                    // through the same L1 and L2 -- the fence is a wait for the stores in flight; at agent scope it is a write-back of
                    // the XCD's L2 and the loads bypass it: 5 x slower
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    const uint8_t *src = out + o0 + (o - dist);
                    for (uint32_t i0 = 0; i0 < len; i0 += (uint32_t)kWave) {
                        const uint32_t i = i0 + (uint32_t)lane;
                        const uint8_t b = __hip_atomic_load(src + (i < len ? i : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        win8[pick(i < len, (o + i) & kWinMask, kDump + (uint32_t)lane)] = b;
                    }
                }
                o += len;
                if (o - flushed >= 1024u) flush_full();
            }
            bp += off;
            // the input of this deflate block must have come from inside the BGZF block
            if (err == kOk && consumed_bits() > (int64_t)clen * 8) err = kErrInput;
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
