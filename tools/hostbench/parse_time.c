/* parse_time.c -- host parser timing: ms per frame of j2k_parse() on one file, with 1, 2, 4 and 8 packet threads
 * (j2k_parser_set_packet_threads: only streams with PLT marker segments take the parallel reader), and the share of the
 * three phases.  The numbers of DESIGN.md section 4 ("PLT / TLM") come from this.
 *   gcc -O2 -std=gnu11 -pthread -Iinclude -Iffmpeg-ht_amd/csrc -o parse_time tools/hostbench/parse_time.c \
 *       ffmpeg-ht_amd/csrc/j2k_syntax.c ffmpeg-ht_amd/csrc/j2k_tier2.c ffmpeg-ht_amd/csrc/j2k_plan.c -lm
 *   ./parse_time frame.j2c          (tools/hostbench/make_frames.py writes the bench's 4K frame with and without PLT) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "j2k_plan.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }

int main(int argc, char **argv)
{
    FILE *f;
    long n;
    uint8_t *b;
    htj2k_opts o;
    int th;
    if (argc < 2 || !(f = fopen(argv[1], "rb"))) { fprintf(stderr, "usage: parse_time FILE\n"); return 2; }
    fseek(f, 0, SEEK_END); n = ftell(f); fseek(f, 0, SEEK_SET);
    b = malloc(n + 64);
    if (fread(b, 1, n, f) != (size_t)n) return 1;
    memset(b + n, 0, 64);
    fclose(f);
    memset(&o, 0, sizeof o);
    o.req_pix_fmt = -1;
    for (th = 1; th <= 8; th *= 2) {
        J2kParser *p = j2k_parser_new();
        const J2kPlan *pl = NULL;
        uint32_t tiles, retries;
        double t0, t;
        int i, r = 0;
        const int N = 40;
        j2k_parser_set_gather(p, 0);                       /* as the device path uses it: no code-block byte is touched */
        j2k_parser_set_packet_threads(p, th);
        for (i = 0; i < 5; i++) r = j2k_parse(p, b, (int)n, &o, 0, &pl);
        t0 = now();
        for (i = 0; i < N; i++) r = j2k_parse(p, b, (int)n, &o, 0, &pl);
        t = (now() - t0) / N;
        j2k_parser_parallel_stats(p, &tiles, &retries);
        printf("%s: %d packet thread%s %.3f ms per frame (returns %d, %d blocks; %u tiles read in parallel, %u frames parsed again)\n",
               argv[1], th, th > 1 ? "s" : " ", t * 1e3, r, pl ? pl->nblocks : 0, tiles, retries);
        j2k_parser_free(p);
    }
    free(b);
    return 0;
}
