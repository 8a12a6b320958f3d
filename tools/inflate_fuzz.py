#!/usr/bin/env python3
"""One-off differential run of inflate_kernel.hip against zlib on many more streams than the test suite holds (GPU): texts of
random alphabets, periods, token mixes and run lengths, every zlib level and strategy, sizes up to a block's 64 KiB, several deflate
blocks per stream, every payload alignment; the CRC32 of every block compared on the device.
usage: tools/inflate_fuzz.py [n_streams=20000] [seed=1]"""
import os
import sys
import zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from basevarc_amd import Context

n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = Context(0)
bad = done = 0
kinds = {}
while done < n_streams:
    comp, blocks, want = bytearray(), [], []
    for _ in range(min(2000, n_streams - done)):
        kind = int(rng.integers(6))
        n = int(rng.choice([1, 7, 300, 5000, 20000, 40000, 65280, 65536])) if rng.random() < 0.5 else int(rng.integers(1, 65537))
        if kind == 0:                                            # random bytes over a random alphabet
            a = int(rng.integers(1, 257))
            data = rng.integers(0, a, n, dtype=np.uint8).tobytes()
        elif kind == 1:                                          # periodic with noise
            p = int(rng.integers(1, 400))
            base = rng.integers(0, 256, p, dtype=np.uint8)
            arr = np.tile(base, n // p + 1)[:n].copy()
            k = int(rng.integers(0, max(1, n // 50)))
            if k:
                arr[rng.integers(0, n, k)] = rng.integers(0, 256, k, dtype=np.uint8)
            data = arr.tobytes()
        elif kind == 2:                                          # pileup-like tokens
            cov = rng.random()
            toks = [("%d,%d,%d,%d,%d " % (rng.integers(4), rng.integers(20, 61), rng.integers(10, 41), rng.integers(1, 150), rng.integers(2)))
                    if rng.random() < cov else ". " for _ in range(n // 3 + 1)]
            data = "".join(toks).encode()[:n]
        elif kind == 3:                                          # long runs
            parts = []
            while sum(map(len, parts)) < n:
                parts.append(bytes([int(rng.integers(256))]) * int(rng.integers(1, 3000)))
            data = b"".join(parts)[:n]
        elif kind == 4:                                          # far repeats: a chunk, filler, the chunk again
            c = rng.integers(0, 256, int(rng.integers(3, 300)), dtype=np.uint8).tobytes()
            data = (c + rng.integers(0, 4, int(rng.integers(0, 33000)), dtype=np.uint8).tobytes()) * 8
            data = data[:n] if len(data) >= n else (data * (n // max(1, len(data)) + 1))[:n]
        else:                                                    # words of a small dictionary
            words = [rng.integers(97, 123, int(rng.integers(2, 12)), dtype=np.uint8).tobytes() for _ in range(int(rng.integers(2, 200)))]
            data = b" ".join(words[int(i)] for i in rng.integers(0, len(words), n // 4 + 1))[:n]
        level = int(rng.integers(0, 10))
        strategy = int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED]))
        co = zlib.compressobj(level, zlib.DEFLATED, -15, int(rng.integers(1, 10)), strategy)
        if rng.random() < 0.3 and len(data) > 10:
            step = max(1, len(data) // int(rng.integers(2, 6)))
            c = b"".join(co.compress(data[i:i + step]) + co.flush(zlib.Z_FULL_FLUSH) for i in range(0, len(data), step)) + co.flush()
        else:
            c = co.compress(data) + co.flush()
        comp += b"\xA5" * int(rng.integers(0, 4))
        blocks.append((len(comp), len(c), len(data), zlib.crc32(data) & 0xffffffff))
        want.append((kind, level, strategy, data))
        comp += c
        kinds[kind] = kinds.get(kind, 0) + 1
    got, status = ctx.inflate_blocks(bytes(comp), blocks)
    for i in range(len(blocks)):
        if status[i] != 0 or got[i] != want[i][3]:
            bad += 1
            if bad <= 10:
                print("MISMATCH", want[i][:3], len(want[i][3]), "status", int(status[i]))
    done += len(blocks)
    print(f"{done} streams, {bad} bad", flush=True)
print("streams by kind:", kinds, "bad:", bad)
sys.exit(1 if bad else 0)
