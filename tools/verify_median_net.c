/* Exhaustive 0/1-principle check of the 25-input median network in median_net25.h:
 * min/max are monotone, so the network selects the median for every real input iff it
 * does for every 0/1 input.  All 2^25 inputs, 64 at a time (min = AND, max = OR). */
#include <stdint.h>
#include <stdio.h>
#include "median_net25.h"
int main(void)
{
    uint64_t bad = 0; int ncx = 0;
#define CNT(a,b) ++ncx;
    DCMT_MED25_NET(CNT, CNT, CNT)
    for (uint64_t base = 0; base < (1ull << 25); base += 64) {
        uint64_t v[25];
        /* bit j of v[i] = bit i of input (base + j) */
        for (int i = 0; i < 25; ++i) {
            uint64_t w = 0;
            for (int j = 0; j < 64; ++j) w |= (((base + j) >> i) & 1ull) << j;
            v[i] = w;
        }
        uint64_t want = 0;
        for (int j = 0; j < 64; ++j) want |= (uint64_t)(__builtin_popcountll(base + j) >= 13) << j;
#define CX(a,b) { uint64_t lo = v[a] & v[b], hi = v[a] | v[b]; v[a] = lo; v[b] = hi; }
#define CMIN(a,b) { v[a] = v[a] & v[b]; v[b] = 0x5555555555555555ull; /* dead: poison */ }
#define CMAX(a,b) { v[b] = v[a] | v[b]; v[a] = 0x5555555555555555ull; }
        DCMT_MED25_NET(CX, CMIN, CMAX)
        bad += __builtin_popcountll(v[12] ^ want);
    }
    printf("comparators=%d mismatches=%llu\n", ncx, (unsigned long long)bad);
    return bad != 0;
}
