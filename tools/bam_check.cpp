// tools/bam_check.cpp - the native BAM reader (vapor_amd/csrc/vapor_bam.cpp, host code) for a sanitizer build: walks every
// record of a file through vapor_bam_chop - one index chunk from the given virtual offset to the end of the file, a window
// per call - and prints the status of each call.  A damaged file must end in a status, never in a report of the sanitizer.
//   g++ -O1 -g -fsanitize=address,undefined -Iinclude -Ivapor_amd/csrc -o /tmp/bam_check tools/bam_check.cpp -lz -lpthread
//   ASAN_OPTIONS=detect_leaks=0 /tmp/bam_check file.bam <first virtual offset> <tid> <start> <end> <flank> [threads]
// tests/test_bamio.py builds it and runs it over the damaged files of its other tests and a few hundred randomly damaged ones.
#include "vapor_bam.cpp"
#include <cstdio>
#include <sys/stat.h>

int main(int argc, char** argv)
{
    if (argc < 7) { fprintf(stderr, "usage: bam_check file.bam first_voffset tid start end flank [threads]\n"); return 2; }
    vapor_bam* b = nullptr;
    if (vapor_bam_open(argv[1], &b) != 0) { printf("open: %s\n", vapor_bam_last_error()); return 0; }
    if (argc > 7) vapor_bam_set_threads(b, atoi(argv[7]));
    struct stat st;
    if (stat(argv[1], &st) != 0) return 2;
    const uint64_t chunk[2] = {strtoull(argv[2], nullptr, 10), (uint64_t)st.st_size << 16};
    const int32_t tid = atoi(argv[3]);
    const int64_t start = atoll(argv[4]), end = atoll(argv[5]), flank = atoll(argv[6]);
    // small buffers first (the overflow answer and its sizes), then as asked for
    std::vector<uint8_t> seq(64);
    std::vector<char> names(8);
    std::vector<int64_t> meta(4);
    int32_t n = 0;
    int64_t need[3] = {0, 0, 0};
    int rc = vapor_bam_chop(b, tid, start, end, flank, 1, chunk, seq.data(), (int64_t)seq.size(), names.data(), (int64_t)names.size(),
                            meta.data(), 1, &n, need);
    printf("small buffers: rc %d reads %d need %lld %lld %lld %s\n", rc, n, (long long)need[0], (long long)need[1], (long long)need[2],
           rc ? vapor_bam_last_error() : "");
    if (rc == VAPOR_E_OVERFLOW) {
        seq.resize((size_t)need[0] + 16); names.resize((size_t)need[1] + 16); meta.resize(4 * ((size_t)need[2] + 1));
        rc = vapor_bam_chop(b, tid, start, end, flank, 1, chunk, seq.data(), (int64_t)seq.size(), names.data(), (int64_t)names.size(),
                            meta.data(), (int32_t)need[2] + 1, &n, need);
        long long bases = 0;
        for (int32_t r = 0; r < n && rc == 0; ++r) bases += meta[4 * r + 1];
        printf("sized buffers: rc %d reads %d bases %lld %s\n", rc, n, bases, rc ? vapor_bam_last_error() : "");
    }
    vapor_bam_close(b);
    return 0;
}
