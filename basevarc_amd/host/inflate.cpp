// inflate.cpp -- raw DEFLATE (RFC 1951) decoder for whole BGZF blocks: input and output are both complete in memory, the
// output size is known (ISIZE), so there is no streaming state, no window copy and no per-byte bounds test in the fast loop.
//
// Why: the temp batches of `basetype` are BGZF text (the reference's form, src/BaseVarC.cpp:509-527); at N = 1e5 samples a
// position is 200-300 KB of text in ~60 KB of deflate, and zlib's inflate (440 MB/s of output on this data) was two thirds of
// the CPU time of the compute phase's host side -- on a box whose cgroup grants 16 CPUs that caps the feed whatever the
// thread count.  This decoder keeps a 64-bit bit buffer refilled eight bytes at a time, decodes literal/length and distance
// symbols through 11-bit / 8-bit first-level tables (second level for the longer codes) whose entries carry "code bits + extra
// bits" so that a length or a distance is one look at the bit buffer and one shift, takes up to three literals per refill and
// copies matches 16 bytes at a time (runs of ". " -- distance 2 -- by a repeated 8-byte pattern).  While 32 input bytes and a
// longest match + 34 output bytes are left, the loop tests no bound but the match distance; the last symbols go through a loop
// that tests them all.
// Own code, written from RFC 1951; checked against zlib on random and adversarial streams (tests/test_host.py).
#include "inflate.h"

#include <cstring>

namespace bvchost {
namespace {

constexpr int kLitBits = 11, kDistBits = 8, kPreBits = 7;
constexpr int kMaxLitSyms = 288, kMaxDistSyms = 32;
// table entry: bits 0..7 = bits to consume (code length, or first-level bits for a sub-table pointer); bits 8..15 = extra bits
// (lengths, distances) or sub-table index bits; bits 16..31 = literal / base value / sub-table offset; flags in bits 28..31
constexpr uint32_t kLiteral = 1u << 31, kEndOfBlock = 1u << 30, kSubTable = 1u << 29, kInvalid = 1u << 28;

struct Tables {
    uint32_t lit[(1 << kLitBits) + kMaxLitSyms * 16];
    uint32_t dist[(1 << kDistBits) + kMaxDistSyms * 128];
    uint32_t pre[1 << kPreBits];
};

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
                                8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t reverse_bits(uint32_t code, int len)
{
    uint32_t r = 0;
    for (int i = 0; i < len; ++i) { r = (r << 1) | (code & 1u); code >>= 1; }
    return r;
}

// Builds a two-level decoding table from code lengths (canonical Huffman, codes read LSB first).  `value(sym, bits)` gives a symbol's
// entry for a code of `bits` bits (counted within its table level).  Returns false for an over-subscribed code and -- zlib's rule
// (inftrees.c: "incomplete set") -- for an incomplete one, except that a literal/length or distance code whose longest code has one
// bit (a single distance code, say) or that has no code at all may be incomplete (`may_be_short`; the unused patterns decode to
// kInvalid); the code-length code must be complete.  The decoder therefore declines exactly the code sets zlib rejects.
template <class Value>
bool build_table(const uint8_t *lens, int n_syms, int table_bits, uint32_t *table, int table_cap, bool may_be_short, Value value)
{
    int count[16] = {0};
    for (int s = 0; s < n_syms; ++s) count[lens[s]]++;
    count[0] = 0;
    uint32_t next_code[16];
    uint32_t code = 0;
    long left = 1;
    for (int l = 1; l <= 15; ++l) {
        left <<= 1;
        left -= count[l];
        if (left < 0) return false;                              // over-subscribed
        code = (code + (uint32_t)count[l - 1]) << 1;
        next_code[l] = code;
    }
    if (left > 0) {                                                  // incomplete
        int longest = 0;
        for (int l = 1; l <= 15; ++l) if (count[l]) longest = l;
        if (!may_be_short || longest > 1) return false;
    }
    const int main_size = 1 << table_bits;
    for (int i = 0; i < main_size; ++i) table[i] = kInvalid | 1u;
    // longest code under every first-level prefix that needs a second level
    uint8_t sub_bits[1 << 11];
    std::memset(sub_bits, 0, (size_t)main_size);
    uint32_t codes[kMaxLitSyms];
    for (int s = 0; s < n_syms; ++s) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t r = reverse_bits(next_code[l]++, l);
        codes[s] = r;
        if (l > table_bits) {
            const uint32_t prefix = r & (uint32_t)(main_size - 1);
            if (l - table_bits > sub_bits[prefix]) sub_bits[prefix] = (uint8_t)(l - table_bits);
        }
    }
    int next_free = main_size;
    for (int p = 0; p < main_size; ++p)
        if (sub_bits[p]) {
            const int size = 1 << sub_bits[p];
            if (next_free + size > table_cap) return false;
            table[p] = kSubTable | ((uint32_t)sub_bits[p] << 24) | ((uint32_t)next_free << 8) | (uint32_t)table_bits;
            for (int i = 0; i < size; ++i) table[next_free + i] = kInvalid | 1u;
            next_free += size;
        }
    for (int s = 0; s < n_syms; ++s) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t r = codes[s];
        if (l <= table_bits) {
            const uint32_t e = value(s, l);
            for (uint32_t i = r; i < (uint32_t)main_size; i += 1u << l) table[i] = e;
        } else {
            const uint32_t prefix = r & (uint32_t)(main_size - 1);
            const uint32_t head = table[prefix];
            const int sb = (int)((head >> 24) & 0xFu);
            const uint32_t off = (head >> 8) & 0xFFFFu;
            const uint32_t e = value(s, l - table_bits);
            for (uint32_t i = r >> table_bits; i < (1u << sb); i += 1u << (l - table_bits)) table[off + i] = e;
        }
    }
    return true;
}

// entry layouts (low 8 bits = bits to consume: the code, and for lengths / distances their extra bits with it):
//   literal      kLiteral | byte << 8 | code bits
//   length       base << 12 (bits 12..20) | code bits << 8 (bits 8..11) | code bits + extra bits
//   end of block kEndOfBlock | code bits
//   distance     base << 12 (bits 12..26) | code bits << 8 | code bits + extra bits
//   sub-table    kSubTable | sub_bits << 24 | offset << 8 | first-level bits
// (second-level entries count their bits from behind the first-level bits, which the decoder drops first)
// A length or distance is then  base + ((bits >> code bits) & ((1 << extra) - 1))  from ONE look at the bit buffer, and one shift
// consumes the code and its extra bits together.
inline uint32_t lit_value(int s, int code_bits)
{
    if (s < 256) return kLiteral | ((uint32_t)s << 8) | (uint32_t)code_bits;
    if (s == 256) return kEndOfBlock | (uint32_t)code_bits;
    if (s > 285) return kInvalid | (uint32_t)code_bits;
    return ((uint32_t)kLenBase[s - 257] << 12) | ((uint32_t)code_bits << 8) | (uint32_t)(code_bits + kLenExtra[s - 257]);
}
inline uint32_t dist_value(int s, int code_bits)
{
    if (s > 29) return kInvalid | (uint32_t)code_bits;
    return ((uint32_t)kDistBase[s] << 12) | ((uint32_t)code_bits << 8) | (uint32_t)(code_bits + kDistExtra[s]);
}
inline uint32_t pre_value(int s, int code_bits) { return ((uint32_t)s << 8) | (uint32_t)code_bits; }
inline uint32_t value_of(uint32_t e, uint64_t bits)               // length / distance entry: base + its extra bits
{
    const int code = (int)((e >> 8) & 0xFu), extra = (int)(e & 0xFFu) - code;
    return ((e >> 12) & 0x7FFFu) + (uint32_t)((bits >> code) & ((1ull << extra) - 1ull));
}

struct Bits {
    const uint8_t *in, *in_end;
    uint64_t buf = 0;
    int cnt = 0;                                                 // bits in buf (the last `pad` of them zeros from beyond the input)
    int pad = 0;
    inline void refill()
    {
        if (in_end - in >= 8) {                                  // at least 56 bits afterwards
            uint64_t w;
            std::memcpy(&w, in, 8);
            buf |= w << cnt;
            in += (63 - cnt) >> 3;
            cnt |= 56;
        } else {
            while (cnt <= 56) {
                if (in < in_end) buf |= (uint64_t)*in++ << cnt;
                else pad += 8;                                   // zeros beyond the end; using them is an error (overrun())
                cnt += 8;
            }
        }
    }
    inline bool overrun() const { return pad > cnt; }            // bits from beyond the input have been consumed
    inline uint32_t peek(int n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }
    inline void drop(int n) { buf >>= n; cnt -= n; }
    inline uint32_t take(int n) { const uint32_t v = peek(n); drop(n); return v; }
};

inline void copy_match(uint8_t *dst, uint32_t dist, uint32_t len, uint8_t *out_end)
{
    const uint8_t *src = dst - dist;
    if (out_end - dst >= (ptrdiff_t)len + 16) {                  // room to write whole words (up to 7 bytes past the match)
        uint8_t *const end = dst + len;
        if (dist >= 8) {                                         // source and destination words do not overlap
            do { uint64_t w; std::memcpy(&w, src, 8); std::memcpy(dst, &w, 8); src += 8; dst += 8; } while (dst < end);
            return;
        }
        // a period that divides 8: one 8-byte pattern, stored again and again (runs of ". " are distance 2)
        uint64_t w;
        if (dist == 1) w = 0x0101010101010101ull * src[0];
        else if (dist == 2) { uint16_t v; std::memcpy(&v, src, 2); w = 0x0001000100010001ull * v; }
        else if (dist == 4) { uint32_t v; std::memcpy(&v, src, 4); w = 0x0000000100000001ull * v; }
        else { for (uint32_t i = 0; i < len; ++i) dst[i] = src[i]; return; }
        do { std::memcpy(dst, &w, 8); dst += 8; } while (dst < end);
        return;
    }
    for (uint32_t i = 0; i < len; ++i) dst[i] = src[i];
}

}  // namespace

long fast_inflate(const unsigned char *in, size_t in_len, unsigned char *out, size_t out_len)
{
    static thread_local Tables T;
    static thread_local bool fixed_ready = false;
    static thread_local uint32_t fixed_lit[(1 << kLitBits) + kMaxLitSyms * 16], fixed_dist[(1 << kDistBits) + kMaxDistSyms * 128];
    Bits b;
    b.in = in; b.in_end = in + in_len;
    uint8_t *o = out, *const o_end = out + out_len;
    bool last = false;
    while (!last) {
        b.refill();
        last = b.take(1) != 0;
        const uint32_t type = b.take(2);
        const uint32_t *lit, *dist;
        if (type == 0) {                                         // stored: LEN, NLEN on the next byte boundary, then the bytes
            b.drop(b.cnt & 7);
            const int real = b.cnt - b.pad;                      // whole input bytes the bit buffer still holds: handed back
            if (real < 0) return -1;
            b.in -= real >> 3; b.buf = 0; b.cnt = 0; b.pad = 0;
            if (b.in_end - b.in < 4) return -1;
            const uint32_t len = (uint32_t)b.in[0] | ((uint32_t)b.in[1] << 8), nlen = (uint32_t)b.in[2] | ((uint32_t)b.in[3] << 8);
            if ((len ^ 0xFFFFu) != nlen) return -1;
            b.in += 4;
            if ((size_t)(b.in_end - b.in) < len || (size_t)(o_end - o) < len) return -1;
            std::memcpy(o, b.in, len);
            o += len; b.in += len;
            continue;
        }
        if (type == 1) {                                         // fixed codes
            if (!fixed_ready) {
                uint8_t l[kMaxLitSyms];
                for (int s = 0; s < 144; ++s) l[s] = 8;
                for (int s = 144; s < 256; ++s) l[s] = 9;
                for (int s = 256; s < 280; ++s) l[s] = 7;
                for (int s = 280; s < 288; ++s) l[s] = 8;
                uint8_t d[kMaxDistSyms];
                for (int s = 0; s < 32; ++s) d[s] = 5;
                if (!build_table(l, 288, kLitBits, fixed_lit, (int)(sizeof fixed_lit / 4), true, lit_value) ||
                    !build_table(d, 32, kDistBits, fixed_dist, (int)(sizeof fixed_dist / 4), true, dist_value)) return -1;
                fixed_ready = true;
            }
            lit = fixed_lit; dist = fixed_dist;
        } else if (type == 2) {                                  // dynamic codes
            const int hlit = (int)b.take(5) + 257, hdist = (int)b.take(5) + 1, hclen = (int)b.take(4) + 4;
            if (hlit > 286 || hdist > 30) return -1;
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t pl[19] = {0};
            for (int i = 0; i < hclen; ++i) { if (b.cnt < 3) b.refill(); pl[order[i]] = (uint8_t)b.take(3); }
            if (!build_table(pl, 19, kPreBits, T.pre, 1 << kPreBits, false, pre_value)) return -1;
            uint8_t lens[kMaxLitSyms + kMaxDistSyms];
            int n = 0;
            while (n < hlit + hdist) {
                b.refill();
                const uint32_t e = T.pre[b.peek(kPreBits)];
                if (e & kInvalid) return -1;
                b.drop((int)(e & 0xFFu));
                const int sym = (int)((e >> 8) & 0xFFu);
                if (sym < 16) { lens[n++] = (uint8_t)sym; continue; }
                int rep; uint8_t v = 0;
                if (sym == 16) { if (n == 0) return -1; v = lens[n - 1]; rep = 3 + (int)b.take(2); }
                else if (sym == 17) rep = 3 + (int)b.take(3);
                else rep = 11 + (int)b.take(7);
                if (n + rep > hlit + hdist) return -1;
                while (rep--) lens[n++] = v;
            }
            if (b.overrun() || lens[256] == 0) return -1;
            if (!build_table(lens, hlit, kLitBits, T.lit, (int)(sizeof T.lit / 4), true, lit_value) ||
                !build_table(lens + hlit, hdist, kDistBits, T.dist, (int)(sizeof T.dist / 4), true, dist_value)) return -1;
            lit = T.lit; dist = T.dist;
        } else {
            return -1;
        }
        // ---- symbols of the block
        bool end_of_block = false;
        // Fast loop: while at least 32 input bytes and 2 + 258 + 32 output bytes are left, an iteration -- up to two literals and a
        // match -- cannot run out of either, so nothing inside tests a bound but the match's distance; the bit buffer is refilled
        // without a branch (eight bytes loaded, as many whole bytes consumed as there is room for).
        while (b.in_end - b.in >= 32 && o_end - o >= 2 + 258 + 32) {
            uint64_t w;
            std::memcpy(&w, b.in, 8);
            b.buf |= w << b.cnt;
            b.in += (63 - b.cnt) >> 3;
            b.cnt |= 56;                                         // >= 56 bits: two literals (2 x 15) and a length (15 + 5) fit
            uint32_t e = lit[b.buf & ((1u << kLitBits) - 1u)];
            if (e & kLiteral) {
                b.buf >>= (e & 0xFFu); b.cnt -= (int)(e & 0xFFu);
                *o++ = (uint8_t)(e >> 8);
                e = lit[b.buf & ((1u << kLitBits) - 1u)];
                if (e & kLiteral) {
                    b.buf >>= (e & 0xFFu); b.cnt -= (int)(e & 0xFFu);
                    *o++ = (uint8_t)(e >> 8);
                    e = lit[b.buf & ((1u << kLitBits) - 1u)];
                    if (e & kLiteral) {                          // (a third one: 45 bits at most so far)
                        b.buf >>= (e & 0xFFu); b.cnt -= (int)(e & 0xFFu);
                        *o++ = (uint8_t)(e >> 8);                // (three literals leave room for no length code: next iteration)
                        continue;
                    }
                }
            }
            if (e & (kSubTable | kEndOfBlock | kInvalid)) {
                if (e & kSubTable) {
                    b.buf >>= kLitBits; b.cnt -= kLitBits;
                    e = lit[((e >> 8) & 0xFFFFu) + (uint32_t)(b.buf & ((1u << ((e >> 24) & 0xFu)) - 1u))];
                    if (e & kLiteral) {
                        b.buf >>= (e & 0xFFu); b.cnt -= (int)(e & 0xFFu);
                        *o++ = (uint8_t)(e >> 8);
                        continue;
                    }
                }
                if (e & kInvalid) return -1;
                if (e & kEndOfBlock) { b.buf >>= (e & 0xFFu); b.cnt -= (int)(e & 0xFFu); end_of_block = true; break; }
            }
            const uint32_t len = value_of(e, b.buf);
            b.buf >>= (e & 0xFFu); b.cnt -= (int)(e & 0xFFu);
            std::memcpy(&w, b.in, 8);                            // a distance: 15 + 13 bits
            b.buf |= w << b.cnt;
            b.in += (63 - b.cnt) >> 3;
            b.cnt |= 56;
            uint32_t d = dist[b.buf & ((1u << kDistBits) - 1u)];
            if (d & (kSubTable | kInvalid)) {
                if (d & kSubTable) {
                    b.buf >>= kDistBits; b.cnt -= kDistBits;
                    d = dist[((d >> 8) & 0xFFFFu) + (uint32_t)(b.buf & ((1u << ((d >> 24) & 0xFu)) - 1u))];
                }
                if (d & kInvalid) return -1;
            }
            const uint32_t distance = value_of(d, b.buf);
            b.buf >>= (d & 0xFFu); b.cnt -= (int)(d & 0xFFu);
            if (distance > (uint32_t)(o - out)) return -1;
            // the copy may write up to 31 bytes past the match: there is room (the loop's condition)
            const uint8_t *src = o - distance;
            uint8_t *dst = o;
            o += len;
            if (distance >= 16) {
                do { std::memcpy(dst, src, 16); src += 16; dst += 16; } while (dst < o);
            } else if (distance >= 8) {
                do { std::memcpy(dst, src, 8); src += 8; dst += 8; } while (dst < o);
            } else {
                uint64_t pat;
                if (distance == 2) { uint16_t v; std::memcpy(&v, src, 2); pat = 0x0001000100010001ull * v; }
                else if (distance == 1) pat = 0x0101010101010101ull * src[0];
                else if (distance == 4) { uint32_t v; std::memcpy(&v, src, 4); pat = 0x0000000100000001ull * v; }
                else { for (uint32_t i = 0; i < len; ++i) dst[i] = src[i]; continue; }
                do { std::memcpy(dst, &pat, 8); dst += 8; } while (dst < o);
            }
        }
        // the last symbols of the input / output (and of every block whose end the fast loop did not reach): every bound tested
        auto lit_entry = [&]() -> uint32_t {                     // (a second-level entry: the first-level bits are dropped here)
            uint32_t e = lit[b.peek(kLitBits)];
            if (e & kSubTable) {
                b.drop(kLitBits);
                e = lit[((e >> 8) & 0xFFFFu) + b.peek((int)((e >> 24) & 0xFu))];
            }
            return e;
        };
        while (!end_of_block) {
            b.refill();                                          // >= 56 bits: two literals (2 x 15) and a length (15 + 5) fit
            uint32_t e = lit_entry();
            if (e & kLiteral) {
                if (o >= o_end) return -1;
                b.drop((int)(e & 0xFFu));
                *o++ = (uint8_t)(e >> 8);
                e = lit_entry();
                if (e & kLiteral) {
                    if (o >= o_end) return -1;
                    b.drop((int)(e & 0xFFu));
                    *o++ = (uint8_t)(e >> 8);
                    continue;
                }
            }
            if (e & kInvalid) return -1;
            if (e & kEndOfBlock) { b.drop((int)(e & 0xFFu)); break; }
            const uint32_t len = value_of(e, b.buf);
            b.drop((int)(e & 0xFFu));
            if (b.cnt < 28) b.refill();                          // a distance: 15 + 13 bits
            uint32_t d = dist[b.peek(kDistBits)];
            if (d & kSubTable) {
                b.drop(kDistBits);
                d = dist[((d >> 8) & 0xFFFFu) + b.peek((int)((d >> 24) & 0xFu))];
            }
            if (d & kInvalid) return -1;
            const uint32_t distance = value_of(d, b.buf);
            b.drop((int)(d & 0xFFu));
            if (distance > (uint32_t)(o - out) || (size_t)(o_end - o) < len) return -1;
            copy_match(o, distance, len, o_end);
            o += len;
        }
        if (b.overrun()) return -1;
    }
    return (long)(o - out);
}

}  // namespace bvchost
