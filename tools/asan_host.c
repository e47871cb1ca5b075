/* ASan/UBSan harness for the host-side C code (FASTA reader, weights): feeds files given on the
 * command line plus mutated copies of them to gkm_problem_read and walks the result. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "gkmkern_pylib.h"
const int64_t *gkm_problem_offsets(const gkm_problem *p);
const uint8_t *gkm_problem_all_codes(const gkm_problem *p);
static unsigned long long rs = 88172645463325252ULL;
static unsigned rnd(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (unsigned)(rs >> 11); }
static long walk(const char *p, const char *n)
{
    gkm_problem *pr = gkm_problem_read(p, n);
    if (!pr) return -1;
    long s = 0;
    int N = gkm_problem_size(pr);
    const int64_t *off = gkm_problem_offsets(pr);
    const uint8_t *c = gkm_problem_all_codes(pr);
    for (int i = 0; i < N; i++) {
        if (off[i + 1] - off[i] != gkm_problem_seqlen(pr, i)) abort();
        for (int64_t k = off[i]; k < off[i + 1]; k++) { if (c[k] > 3) abort(); s += c[k]; }
    }
    s += gkm_problem_npos(pr) + gkm_problem_invalid_chars(pr) + gkm_problem_truncated(pr);
    gkm_problem_free(pr);
    return s;
}
int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    printf("plain: %ld\n", walk(argv[1], argv[2]));
    /* mutations of the positive file */
    FILE *f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long len = ftell(f); rewind(f);
    char *buf = malloc((size_t)len + 64); if (fread(buf, 1, (size_t)len, f) != (size_t)len) return 3; fclose(f);
    long ok = 0, rej = 0;
    for (int it = 0; it < 3000; it++) {
        char *m = malloc((size_t)len + 64); memcpy(m, buf, (size_t)len); long ml = len;
        int nm = 1 + (int)(rnd() % 8);
        for (int k = 0; k < nm; k++) {
            int op = (int)(rnd() % 5); long at = ml ? (long)(rnd() % (unsigned long)ml) : 0;
            if (op == 0 && ml) m[at] = (char)(rnd() & 255);
            else if (op == 1 && ml) m[at] = '>';
            else if (op == 2 && ml) m[at] = '\n';
            else if (op == 3 && ml > 4) ml = at;                    /* truncate */
            else if (op == 4 && ml) m[at] = '\r';
        }
        FILE *o = fopen("/tmp/gkm_asan_mut.fa", "wb"); fwrite(m, 1, (size_t)ml, o); fclose(o); free(m);
        long r = walk("/tmp/gkm_asan_mut.fa", argv[2]);
        if (r < 0) rej++; else ok++;
        r = walk(argv[2], "/tmp/gkm_asan_mut.fa");
        if (r < 0) rej++; else ok++;
    }
    /* large files: the positive file repeated to ~700 KB, mutated the same way */
    {
        const long reps = 700 * 1024 / (len > 0 ? len : 1) + 1, bl = reps * len;
        char *big = malloc((size_t)bl + 64);
        for (long r = 0; r < reps; r++) memcpy(big + r * len, buf, (size_t)len);
        long bok = 0, brej = 0;
        for (int it = 0; it < 150; it++) {
            char *m = malloc((size_t)bl + 64); memcpy(m, big, (size_t)bl); long ml = bl;
            const int nm = 1 + (int)(rnd() % 64);
            for (int k = 0; k < nm; k++) {
                const int op = (int)(rnd() % 5); const long at = (long)(rnd() % (unsigned long)ml);
                if (op == 0) m[at] = (char)(rnd() & 255);
                else if (op == 1) m[at] = '>';
                else if (op == 2) m[at] = '\n';
                else if (op == 3 && it % 10 == 0) ml = at > 300 * 1024 ? at : ml;   /* truncate (keep it above the threshold) */
                else if (op == 4) m[at] = '\r';
            }
            FILE *o = fopen("/tmp/gkm_asan_big.fa", "wb"); fwrite(m, 1, (size_t)ml, o); fclose(o); free(m);
            if (walk("/tmp/gkm_asan_big.fa", argv[2]) < 0) brej++; else bok++;
            if (walk(argv[2], "/tmp/gkm_asan_big.fa") < 0) brej++; else bok++;
        }
        printf("large files: %ld accepted, %ld rejected\n", bok, brej);
        free(big);
    }
    /* weights for every admissible parameter set */
    double c[13]; uint8_t wt[4096];
    for (int t = 0; t < 6; t++) for (int L = 2; L <= 12; L++) for (int k = 1; k <= L; k++) gkm_mismatch_weights(t, L, k, c);
    for (int n = 1; n < 2100; n += 37) for (int M = 1; M < 256; M += 50) gkm_position_weights(4, n, (uint8_t)M, 50.0, wt);
    printf("mutations: %ld accepted, %ld rejected\n", ok, rej);
    free(buf);
    return 0;
}
