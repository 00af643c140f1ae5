// tools/inflate_check.cpp - vapor_amd/csrc/vapor_inflate.h against zlib, for a sanitizer build and for timing.
//   g++ -O2 -g -fsanitize=address,undefined -Ivapor_amd/csrc -o /tmp/inflate_check tools/inflate_check.cpp -lz && ASAN_OPTIONS=detect_leaks=0 /tmp/inflate_check
//   g++ -O3 -Ivapor_amd/csrc -o /tmp/inflate_check tools/inflate_check.cpp -lz && /tmp/inflate_check some.bam      (every BGZF block, both decoders timed)
// 1 200 streams of six kinds of data x sizes x levels x strategies, each also truncated and with flipped bits: the decoder
// must accept exactly what zlib accepts, with the same bytes (tests/test_bamio.py runs a part of this through the C ABI).
#include "vapor_inflate.h"
#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
#include <random>
static bool zinflate(const uint8_t* in, size_t n, uint8_t* out, size_t on) {
    z_stream zs; memset(&zs, 0, sizeof zs); inflateInit2(&zs, -15);
    zs.next_in = (Bytef*)in; zs.avail_in = n; zs.next_out = out; zs.avail_out = on;
    int rc = inflate(&zs, Z_FINISH); inflateEnd(&zs);
    return rc == Z_STREAM_END && zs.avail_out == 0;
}
static std::vector<uint8_t> zdeflate(const std::vector<uint8_t>& src, int level, int strategy) {
    z_stream zs; memset(&zs, 0, sizeof zs); deflateInit2(&zs, level, Z_DEFLATED, -15, 8, strategy);
    std::vector<uint8_t> out(deflateBound(&zs, src.size()) + 64);
    zs.next_in = (Bytef*)src.data(); zs.avail_in = src.size(); zs.next_out = out.data(); zs.avail_out = out.size();
    deflate(&zs, Z_FINISH); out.resize(zs.total_out); deflateEnd(&zs); return out;
}
int main(int argc, char** argv) {
    vapor_inflate::Decoder* dec = new vapor_inflate::Decoder();
    std::mt19937 rng(7);
    long cases = 0, bad_ok = 0;
    // random data of several kinds x levels x strategies
    for (int kind = 0; kind < 6; ++kind)
        for (int size : {0, 1, 2, 7, 100, 1000, 65280, 200000})
            for (int level : {0, 1, 4, 6, 9})
                for (int strat : {Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE, Z_FILTERED}) {
                    std::vector<uint8_t> src(size);
                    for (int i = 0; i < size; ++i) {
                        switch (kind) {
                        case 0: src[i] = rng(); break;
                        case 1: src[i] = "ACGT"[rng() & 3]; break;
                        case 2: src[i] = (i % 37) < 30 ? 'A' + (i / 1000) % 4 : rng() & 0xFF; break;
                        case 3: src[i] = 0; break;
                        case 4: src[i] = (rng() % 100 < 90) ? 'I' : 33 + rng() % 40; break;
                        case 5: src[i] = i < 4 ? rng() : src[i - 1 - (rng() % 3)] ; break;
                        }
                    }
                    auto comp = zdeflate(src, level, strat);
                    std::vector<uint8_t> out(size + 1, 0xEE);
                    if (!vapor_inflate::inflate_raw(comp.data(), comp.size(), out.data(), size, *dec) || (size && memcmp(out.data(), src.data(), size)) || out[size] != 0xEE) {
                        printf("MISMATCH kind %d size %d level %d strat %d\n", kind, size, level, strat); return 1;
                    }
                    ++cases;
                    // wrong expected size must fail
                    if (size > 0 && vapor_inflate::inflate_raw(comp.data(), comp.size(), out.data(), size - 1, *dec)) { printf("short output accepted\n"); return 1; }
                    // truncated / corrupted input never crashes, and agrees with zlib when zlib accepts it
                    for (int t = 0; t < 6 && comp.size() > 2; ++t) {
                        auto c2 = comp;
                        if (t < 2) c2.resize(rng() % comp.size());
                        else c2[rng() % c2.size()] ^= 1u << (rng() & 7);
                        std::vector<uint8_t> o1(size + 1), o2(size + 1);
                        bool a = vapor_inflate::inflate_raw(c2.data(), c2.size(), o1.data(), size, *dec);
                        bool z = zinflate(c2.data(), c2.size(), o2.data(), size);
                        if (a != z || (a && memcmp(o1.data(), o2.data(), size))) {
                            // zlib stops at the end marker and ignores trailing input; so do we. Anything else is a difference.
                            printf("DIFF on corrupted input: ours %d zlib %d (kind %d size %d level %d strat %d t %d)\n", a, z, kind, size, level, strat, t); return 1;
                        }
                        bad_ok += !a;
                    }
                }
    printf("%ld cases ok, %ld corrupted streams rejected alike\n", cases, bad_ok);
    if (argc > 1) {
        FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
        std::vector<uint8_t> buf(n); if (fread(buf.data(), 1, n, f) != (size_t)n) return 2; fclose(f);
        std::vector<uint8_t> out(1 << 16), ref(1 << 16);
        for (int rep = 0; rep < 3; ++rep) {
            for (int which = 0; which < 2; ++which) {
                auto t0 = std::chrono::steady_clock::now();
                size_t p = 0, tot = 0, nb = 0;
                while (p + 18 < (size_t)n) {
                    int bsize = (buf[p + 16] | (buf[p + 17] << 8)) + 1;
                    int xlen = buf[p + 10] | (buf[p + 11] << 8);
                    uint32_t isize; memcpy(&isize, &buf[p + bsize - 4], 4);
                    bool ok = which ? vapor_inflate::inflate_raw(&buf[p + 12 + xlen], bsize - xlen - 20, out.data(), isize, *dec)
                                    : zinflate(&buf[p + 12 + xlen], bsize - xlen - 20, ref.data(), isize);
                    if (!ok) { printf("block at %zu failed (%d)\n", p, which); return 1; }
                    if (which && rep == 0) { zinflate(&buf[p + 12 + xlen], bsize - xlen - 20, ref.data(), isize); if (memcmp(out.data(), ref.data(), isize)) { printf("block at %zu differs\n", p); return 1; } }
                    tot += isize; p += bsize; ++nb;
                }
                double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (!(which && rep == 0)) printf("%s: %zu blocks -> %zu bytes in %.3f s: %.0f MB/s\n", which ? "ours" : "zlib", nb, tot, dt, tot / dt / 1e6);
            }
        }
    }
    return 0;
}
