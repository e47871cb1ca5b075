/* ASan/UBSan harness for the per-lane program of the hot kernel as compiled for the CPU
 * (gkm_bitslice.h, gkm_pack.h through bitslice_cpu_probe.cpp): random length distributions through
 * the row packing, the window counting and the hit resolution; the probe's own invariants must
 * hold (return code 0).   Built and run by tools/asan_host.sh. */
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

extern "C" int bsprobe_profile_packed(int W, int L, int d, const uint8_t *codes, const int64_t *off, const int *rows,
                                      int nrows, int col, const uint8_t *wd, int32_t *P, int *lanes_used);
extern "C" int bsprobe_profile(int W, int L, int d, const uint8_t *A, int lenA, const uint8_t *B, int lenB,
                               const uint8_t *wd, int32_t *P);

static unsigned long long rs = 0x9E3779B97F4A7C15ULL;
static unsigned rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (unsigned)(rs >> 11); }

int main()
{
    const int cases[][3] = {{10, 11, 3}, {10, 12, 4}, {10, 6, 2}, {5, 11, 3}};
    std::vector<uint8_t> wd(2048);
    for (size_t i = 0; i < wd.size(); i++) wd[i] = (uint8_t)(1 + rnd() % 255);
    long runs = 0;
    for (int it = 0; it < 120; it++) {
        const int *c = cases[it % 4];
        const int W = c[0], L = c[1], d = c[2];
        const int n = 2 + (int)(rnd() % 70);
        std::vector<int64_t> off(n + 1, 0);
        std::vector<uint8_t> codes;
        for (int i = 0; i < n; i++) {
            int len;
            switch (rnd() % 4) {
            case 0: len = L + (int)(rnd() % 12); break;
            case 1: len = L + (int)(rnd() % 700); break;
            case 2: len = 2047 - (int)(rnd() % 3); break;
            default: len = 300; break;
            }
            for (int k = 0; k < len; k++) codes.push_back((uint8_t)(rnd() & 3));
            off[i + 1] = off[i] + len;
        }
        std::vector<int> rows(n);
        for (int i = 0; i < n; i++) rows[i] = i;
        std::vector<int32_t> P((size_t)n * (d + 1));
        int lanes = 0;
        const int col = (int)(rnd() % n);
        const int rc = bsprobe_profile_packed(W, L, d, codes.data(), off.data(), rows.data(), n, col,
                                              (it & 1) ? wd.data() : nullptr, P.data(), &lanes);
        if (rc != 0) { fprintf(stderr, "bsprobe_profile_packed: invariant %d violated (case %d)\n", rc, it); return 1; }
        int32_t P1[16];
        const int a = (int)(rnd() % n);
        if (bsprobe_profile(10, L == 6 ? 6 : L, L == 6 ? 3 : d, codes.data() + off[a], (int)(off[a + 1] - off[a]),
                            codes.data() + off[col], (int)(off[col + 1] - off[col]), (it & 1) ? wd.data() : nullptr, P1) > 1)
            return 2;
        runs++;
    }
    printf("bit-sliced core under ASan/UBSan: %ld random problems, invariants hold\n", runs);
    return 0;
}
