// inflate.h -- raw DEFLATE decoder for whole in-memory blocks (inflate.cpp).
#pragma once
#include <cstddef>
#include <cstdint>

namespace bvchost {

// Inflates the raw deflate stream in[0..in_len) into out[0..out_len).  Returns the number of bytes written (the caller
// compares it with the size it expects: BGZF's ISIZE), or -1 for a stream that is not valid deflate, needs more input than
// given or more room than out_len.  Never reads outside `in` or writes outside `out`.  Thread-safe (per-thread tables).
long fast_inflate(const unsigned char *in, size_t in_len, unsigned char *out, size_t out_len);

}  // namespace bvchost
