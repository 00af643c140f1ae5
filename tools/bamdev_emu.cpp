// bamdev_emu.cpp - the DEFLATE decoder of bgzf_inflate_kernel (vapor_amd/csrc/vapor_bamdev.h) compiled for the host with one
// "lane" doing the wavefront's loops in order, against zlib: streams made by zlib at every level and strategy (stored, fixed,
// dynamic codes; long matches, overlapping matches, distances of 32 K; many small deflate blocks in one stream), sizes from 0 to
// 65 536 bytes, the CRC-32 by slices against zlib's, and damaged streams (every outcome but a wrong "ok" is fine; the address
// and undefined-behaviour sanitizers watch the buffers).  Built and run by tests/test_bamdev_emu.py:
//   g++ -O1 -g -fsanitize=address,undefined -DVBD_EMU -Ivapor_amd/csrc tools/bamdev_emu.cpp -lz -o /tmp/bamdev_emu
#include "vapor_bamdev.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

using namespace vapor_bamdev;

static uint32_t g_pow[64];

static std::vector<uint8_t> deflate_raw(const std::vector<uint8_t>& data, int level, int strategy, int chunk)
{
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, strategy) != Z_OK) abort();
    std::vector<uint8_t> out(deflateBound(&zs, (uLong)data.size()) + 64 + (chunk ? data.size() / (size_t)chunk * 16 : 0));
    zs.next_out = out.data();
    zs.avail_out = (uInt)out.size();
    size_t p = 0;
    if (chunk) {
        // several deflate blocks in the stream: a full flush every `chunk` bytes (an empty stored block follows each)
        while (p + (size_t)chunk < data.size()) {
            zs.next_in = const_cast<Bytef*>(data.data() + p);
            zs.avail_in = (uInt)chunk;
            if (deflate(&zs, Z_FULL_FLUSH) != Z_OK) abort();
            p += (size_t)chunk;
        }
    }
    zs.next_in = const_cast<Bytef*>(data.data() + p);
    zs.avail_in = (uInt)(data.size() - p);
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) abort();
    out.resize(zs.total_out);
    deflateEnd(&zs);
    return out;
}

static int run(const std::vector<uint8_t>& comp, size_t u_len, uint32_t crc, std::vector<uint8_t>& got, int align)
{
    static InflateLds L;
    static CrcTables T;
    static bool have_t = false;
    if (!have_t) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            T.t[0][i] = c;
        }
        have_t = true;
    }
    static std::vector<uint8_t> image;
    image.assign((size_t)U_MAX + 32, 0xAA);
    uint8_t* out = image.data() + align;
    const int rc = inflate_block_wave<1>(L, T, out, comp.data(), (uint32_t)comp.size(), (uint32_t)u_len, crc, g_pow, 0);
    got.assign(out, out + u_len);
    return rc;
}

int main(int argc, char** argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 40;
    for (int l = 0; l < 64; ++l) {
        // x^(8 * 1024 * (63 - l)) mod P by square and multiply (bit 31 = x^0, bit 30 = x^1)
        uint64_t e = (uint64_t)8 * 1024 * (uint64_t)(63 - l);
        uint32_t r = 0x80000000u, b = 0x40000000u;
        while (e) { if (e & 1) r = crc_mulmod(r, b); b = crc_mulmod(b, b); e >>= 1; }
        g_pow[l] = r;
    }
    std::mt19937_64 rng(12345);
    long n_ok = 0, n_damaged = 0, n_damaged_caught = 0, n_tables = 0;
    const int sizes[] = {0, 1, 2, 3, 4, 5, 63, 64, 65, 1023, 1024, 1025, 4096, 16384, 40000, 65279, 65280, 65535, 65536};
    for (int round = 0; round < rounds; ++round) {
        for (int kind = 0; kind < 7; ++kind) {
            for (size_t si = 0; si < sizeof sizes / sizeof sizes[0]; ++si) {
                const size_t n = (size_t)sizes[si];
                std::vector<uint8_t> data(n);
                switch (kind) {
                case 0: for (auto& c : data) c = (uint8_t)rng(); break;                                  // incompressible
                case 1: for (auto& c : data) c = "ACGT"[rng() & 3]; break;                               // four letters
                case 2: for (size_t i = 0; i < n; ++i) data[i] = (uint8_t)(i < 5 ? rng() : data[i - 1 - (rng() % 5)]); break;   // overlapping copies
                case 3: for (auto& c : data) c = 0xFF; break;                                            // one byte (distance 1, length 258)
                case 4: {                                                                                // BAM-like: packed bases, qualities, copies of earlier reads
                    size_t i = 0;
                    while (i < n) {
                        const size_t len = 200 + rng() % 3000;
                        const bool copy = i > 6000 && (rng() & 1);
                        const size_t from = copy ? i - 1 - rng() % std::min<size_t>(i - 1, 32000) : 0;
                        for (size_t j = 0; j < len && i < n; ++j, ++i)
                            data[i] = copy && (rng() % 10) ? data[from + j % (i - from)] : (uint8_t)(((rng() & 3) << 4 | (rng() & 3)) + 17);
                        for (size_t j = 0; j < len && i < n; ++j, ++i) data[i] = (uint8_t)(20 + rng() % 30);
                    }
                    break;
                }
                case 5: for (size_t i = 0; i < n; ++i) data[i] = (uint8_t)((i / 300) * 7 + (rng() % 3 == 0)); break;   // long runs
                default: for (size_t i = 0; i < n; ++i) data[i] = (uint8_t)(i < 32768 ? rng() : data[i - 32768]); break;  // the longest distance
                }
                const int level = (int)(rng() % 10);
                const int strat = (int[]){Z_DEFAULT_STRATEGY, Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED}[rng() % 5];
                const int chunk = (rng() % 3 == 0 && n > 100) ? (int)(50 + rng() % 5000) : 0;
                const std::vector<uint8_t> comp = deflate_raw(data, level, strat, chunk);
                const uint32_t crc = (uint32_t)crc32(0L, data.data(), (uInt)n);
                std::vector<uint8_t> got;
                const int align = (int)(rng() % 16);
                int rc = run(comp, n, crc, got, align);
                if (rc == BLK_TABLES) { ++n_tables; continue; }
                if (rc != BLK_OK || got != data) {
                    fprintf(stderr, "FAIL round %d kind %d size %zu level %d strategy %d chunk %d: rc %d, bytes %s\n", round, kind, n, level, strat, chunk, rc,
                            got == data ? "equal" : "differ");
                    return 1;
                }
                ++n_ok;
                // the same stream with a wrong CRC, a wrong size, cut short, and with a flipped bit
                if (run(comp, n, crc ^ 1u, got, align) != BLK_CRC) { fprintf(stderr, "FAIL: a wrong CRC was accepted (size %zu)\n", n); return 1; }
                if (n > 0 && run(comp, n - 1, crc, got, align) == BLK_OK) { fprintf(stderr, "FAIL: a short ISIZE was accepted\n"); return 1; }
                if (n < (size_t)U_MAX && run(comp, n + 1, crc, got, align) == BLK_OK) { fprintf(stderr, "FAIL: a long ISIZE was accepted\n"); return 1; }
                if (comp.size() > 2) {
                    std::vector<uint8_t> cut(comp.begin(), comp.begin() + (long)(rng() % comp.size()));
                    if (run(cut, n, crc, got, align) == BLK_OK && n > 0) { fprintf(stderr, "FAIL: a truncated stream was accepted (size %zu of %zu)\n", cut.size(), comp.size()); return 1; }
                    std::vector<uint8_t> bad = comp;
                    bad[rng() % bad.size()] ^= (uint8_t)(1u << (rng() % 8));
                    ++n_damaged;
                    rc = run(bad, n, crc, got, align);
                    if (rc != BLK_OK) ++n_damaged_caught;
                    else if (got != data) { fprintf(stderr, "FAIL: a damaged stream gave other bytes under the same CRC\n"); return 1; }
                }
            }
        }
    }
    printf("bamdev_emu: %ld streams equal zlib's bytes and CRC-32 (%ld refused for table size), %ld of %ld damaged streams refused (the rest decode to the same bytes)\n",
           n_ok, n_tables, n_damaged_caught, n_damaged);
    return 0;
}
