/* Plain-C caller of the multi-device entry point (include/acgpu.h only; links libacgpu.so and nothing else -- what a
 * Rust `extern "C"` binding would see).  The haystack is generated on device 0, cut into N virtual shards (each with its
 * max_pattern_len-1 halo) that are handed to acgpu_find_overlapping_multi as devices {0,0,...}; the gathered stream must
 * equal acgpu_find_overlapping over the whole haystack, record for record.  The summary line also carries the order-
 * sensitive FNV-1a of the gathered stream (the fold of oracle/ac_oracle.c: orc_hash_matches): tests/test_c_multi.py
 * regenerates patterns and haystack and compares count and hash with the CPU oracle's stream.
 *   multi_test [n_shards] [MiB]        exit 0 = identical; prints one summary line. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "acgpu.h"

#define CHECK(call)                                                                                        \
    do {                                                                                                   \
        acgpu_status st_ = (call);                                                                         \
        if (st_ != ACGPU_OK) {                                                                             \
            fprintf(stderr, "%s -> %s (%s / %s)\n", #call, acgpu_status_str(st_), acgpu_last_error(), acgpu_multi_last_error()); \
            return 2;                                                                                      \
        }                                                                                                  \
    } while (0)

static uint64_t fnv_fold(uint64_t h, uint64_t w) {
    for (int i = 0; i < 8; i++) { h ^= (w >> (8 * i)) & 0xFF; h *= 0x100000001B3ull; }
    return h;
}

static uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main(int argc, char** argv) {
    const size_t n_shards = argc > 1 ? (size_t)atoi(argv[1]) : 3;
    const size_t n = (size_t)(argc > 2 ? atoi(argv[2]) : 64) << 20;
    if (argc > 1 && strcmp(argv[1], "--list") == 0) { printf("multi_test: abi %u\n", acgpu_abi_version()); return 0; }
    int32_t ndev = 0;
    CHECK(acgpu_device_count(&ndev));
    if (ndev < 1 || n_shards < 1 || n_shards > 64) { fprintf(stderr, "no device / bad shard count\n"); return 2; }

    /* 1000 patterns of 4..16 printable bytes (the headline set's shape) */
    enum { NP = 1000 };
    static uint8_t pbytes[NP][16];
    const uint8_t* pats[NP];
    size_t lens[NP];
    uint64_t ctr = 0;
    for (int i = 0; i < NP; i++) {
        lens[i] = 4 + (size_t)(splitmix64(0xAC01 ^ ctr++) % 13);
        for (size_t k = 0; k < lens[i]; k++) pbytes[i][k] = (uint8_t)(0x20 + splitmix64(0xAC01 ^ ctr++) % 95);
        pats[i] = pbytes[i];
    }
    acgpu_config cfg;
    acgpu_config_init(&cfg);
    cfg.kind = ACGPU_KIND_DFA;
    acgpu_automaton* aut = NULL;
    CHECK(acgpu_build(&cfg, pats, lens, NP, &aut));
    const size_t L = acgpu_max_pattern_len(aut), halo = L - 1;

    uint8_t* d_hay = NULL;
    CHECK(acgpu_device_malloc(0, n + 64, (void**)&d_hay));
    CHECK(acgpu_gen_haystack(d_hay, 0, n, 0xAC02, 0x20, 95, NULL));
    /* occurrences straddling every seam and scattered inside the shards */
    for (size_t k = 1; k <= 4 * n_shards; k++) {
        size_t pos = k * (n / (4 * n_shards + 1));
        if (k % 4 == 0) pos = (k / 4) * (n / n_shards) - 1 - (k % 7);   /* across the seam of shard k/4 */
        const int p = (int)((7 * k) % NP);
        if (pos + lens[p] <= n) CHECK(acgpu_device_copy(0, d_hay + pos, pats[p], lens[p], 0));
    }

    const size_t cap = 1 << 20;
    acgpu_match *d_one = NULL, *d_multi = NULL;
    CHECK(acgpu_device_malloc(0, cap * sizeof(acgpu_match), (void**)&d_one));
    CHECK(acgpu_device_malloc(0, cap * sizeof(acgpu_match), (void**)&d_multi));
    acgpu_input in;
    memset(&in, 0, sizeof in);
    in.haystack = d_hay; in.haystack_len = n; in.span_start = 0; in.span_end = n;
    in.haystack_on_device = 1; in.out_on_device = 1;
    size_t n_one = 0, n_multi = 0;
    CHECK(acgpu_find_overlapping(aut, &in, d_one, cap, &n_one));

    acgpu_shard sh[64];
    uint64_t counts[64];
    for (size_t i = 0; i < n_shards; i++) {
        const size_t b = i * (n / n_shards), e = i + 1 == n_shards ? n : (i + 1) * (n / n_shards);
        const size_t left = i ? halo : 0;   /* the shard's buffer starts `left` bytes before its first own byte */
        memset(&sh[i], 0, sizeof sh[i]);
        sh[i].device = 0;
        sh[i].haystack = d_hay + (b - left);
        sh[i].haystack_len = (e - b) + left;
        sh[i].span_start = 0; sh[i].span_end = sh[i].haystack_len;
        sh[i].shard_begin = left; sh[i].shard_end = sh[i].haystack_len;
        sh[i].global_offset = b - left;
    }
    CHECK(acgpu_find_overlapping_multi(aut, sh, n_shards, 0, d_multi, cap, &n_multi, counts));

    int bad = n_one != n_multi;
    acgpu_match* a = (acgpu_match*)malloc((n_one + 1) * sizeof *a);
    acgpu_match* b = (acgpu_match*)malloc((n_multi + 1) * sizeof *b);
    CHECK(acgpu_device_copy(0, a, d_one, n_one * sizeof *a, 1));
    CHECK(acgpu_device_copy(0, b, d_multi, n_multi * sizeof *b, 1));
    for (size_t i = 0; !bad && i < n_one; i++)
        bad = a[i].pattern != b[i].pattern || a[i].start != b[i].start || a[i].end != b[i].end;
    uint64_t sum = 0;
    for (size_t i = 0; i < n_shards; i++) sum += counts[i];
    bad |= sum != n_multi;
    uint64_t h = 0xCBF29CE484222325ull;
    for (size_t i = 0; i < n_multi; i++) h = fnv_fold(fnv_fold(fnv_fold(h, b[i].pattern), b[i].start), b[i].end);
    printf("multi_test: %zu shards on device 0, %zu MiB, %zu records (single call %zu), transport %d, hash %016llx: %s\n", n_shards,
           n >> 20, n_multi, n_one, acgpu_multi_last_transport(), (unsigned long long)h, bad ? "MISMATCH" : "identical");
    free(a); free(b);
    acgpu_device_free(0, d_hay); acgpu_device_free(0, d_one); acgpu_device_free(0, d_multi);
    acgpu_free(aut);
    return bad ? 1 : 0;
}
