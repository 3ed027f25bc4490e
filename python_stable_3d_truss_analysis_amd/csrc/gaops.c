/*
 * gaops.c - the host side of one GA generation, natively (BASELINE config 4: ga.GA.Evolve with nPop = 1024).
 *
 * Reference: slientruss3d/ga.py:155-190 - Select (stable sort by fitness), UpdatePop (per offspring slot ONE
 * random.random() decides crossover / mutation / keep / fresh gene), Crossover (ga.py:162-165), Mutate
 * (ga.py:167-171), GetRandomGene (ga.py:136-137).  A seeded run of the reference walks a fixed trajectory,
 * and this package promises the same one (tests/golden/ga_trace.json), so the order and the NUMBER of draws from
 * Python's global `random` generator are part of the interface.  This file therefore carries CPython's
 * Mersenne Twister (MT19937, Matsumoto & Nishimura; CPython Modules/_randommodule.c) and the few derived draws the
 * GA makes, bit for bit as CPython 3.2 ... 3.12 make them (Lib/random.py):
 *     random()            a = u32 >> 5, b = u32 >> 6, (a * 2^26 + b) / 2^53
 *     _randbelow(n)       k = bit_length(n); r = u32 >> (32 - k) until r < n          (getrandbits, k <= 32)
 *     sample(pop, 2)      n <= 21: pool selection (swap the last unselected item into the vacancy);
 *                         else: draw until distinct
 *     choice(seq)         seq[_randbelow(len(seq))]
 *     randint(a, b)       a + _randbelow(b - a + 1)
 *     choices(range(n), k=m)   floor(random() * n), m times
 * The Python caller hands over `random.getstate()` (624 words + position) and puts the advanced state back with
 * `random.setstate()`: Python code that draws before or after sees exactly the numbers it would have seen.
 */
#include <stdint.h>
#include <string.h>

#define MT_N 624
#define MT_M 397

typedef struct {
    uint32_t mt[MT_N];
    int pos;
} Mt;

static uint32_t mt_u32(Mt *s) {
    static const uint32_t mag01[2] = {0u, 0x9908b0dfu};
    uint32_t y;
    if (s->pos >= MT_N) {
        int kk;
        uint32_t *mt = s->mt;
        for (kk = 0; kk < MT_N - MT_M; kk++) {
            y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + MT_M] ^ (y >> 1) ^ mag01[y & 1u];
        }
        for (; kk < MT_N - 1; kk++) {
            y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ mag01[y & 1u];
        }
        y = (mt[MT_N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
        mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ mag01[y & 1u];
        s->pos = 0;
    }
    y = s->mt[s->pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

static double mt_random(Mt *s) {
    const uint32_t a = mt_u32(s) >> 5, b = mt_u32(s) >> 6;
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
}

static uint32_t mt_randbelow(Mt *s, uint32_t n) {
    int k = 0;
    uint32_t r;
    if (n == 0) return 0;
    while ((n >> k) != 0) ++k; /* bit_length */
    do {
        r = mt_u32(s) >> (32 - k);
    } while (r >= n);
    return r;
}

/* indices random.sample(range(n), 2) returns (population items are their own indices) */
static void mt_sample2(Mt *s, uint32_t n, uint32_t *i0, uint32_t *i1) {
    if (n <= 21) { /* pool path */
        const uint32_t j0 = mt_randbelow(s, n), j1 = mt_randbelow(s, n - 1);
        *i0 = j0;
        *i1 = (j1 == j0) ? n - 1 : j1; /* pool[j0] was replaced by pool[n - 1] */
    } else {
        uint32_t j0 = mt_randbelow(s, n), j1 = mt_randbelow(s, n);
        while (j1 == j0) j1 = mt_randbelow(s, n);
        *i0 = j0;
        *i1 = j1;
    }
}

/* One UpdatePop (ga.py:173-190) on gene matrices of bytes (type index per member).
 *   state      in/out: random.getstate()[1] = 624 words followed by the position
 *   elite      [nElite][nMember]   the selected genes in rank order
 *   pop        [nPop][nMember]     the current population (rows >= nElite may be kept)
 *   out        [nPop][nMember]     the next population (may not alias pop or elite)
 *   counts     [4] or NULL         out: offspring by crossover / mutation / kept / fresh
 * Returns 0, or -1 for arguments the Python path has to handle (fewer than two elites or members or types, more
 * than 256 types). */
int trs_ga_update_pop(uint32_t *state /* [625] */, int nPop, int nElite, int nMember, int nType, double toCross,
                      double toMutate, double toKeep, const uint8_t *elite, const uint8_t *pop, uint8_t *out,
                      int32_t *counts) {
    Mt s;
    int j, i;
    int32_t c[4] = {0, 0, 0, 0};
    if (nElite < 2 || nMember < 2 || nType < 2 || nType > 256 || nPop < nElite || state[MT_N] > MT_N) return -1;
    memcpy(s.mt, state, sizeof(s.mt));
    s.pos = (int)state[MT_N];
    memcpy(out, elite, (size_t)nElite * nMember);
    for (j = nElite; j < nPop; ++j) {
        uint8_t *child = out + (size_t)j * nMember;
        const double p = mt_random(&s);
        if (p <= toCross) { /* Crossover(*random.sample(elitePop, k=2)) */
            uint32_t a, b, c0, c1;
            mt_sample2(&s, (uint32_t)nElite, &a, &b);
            mt_sample2(&s, (uint32_t)nMember, &c0, &c1);
            if (c0 > c1) {
                const uint32_t t = c0;
                c0 = c1;
                c1 = t;
            }
            memcpy(child, elite + (size_t)a * nMember, (size_t)nMember);
            memcpy(child + c0, elite + (size_t)b * nMember + c0, (size_t)(c1 - c0));
            ++c[0];
        } else if (p <= toMutate) { /* Mutate(random.choice(elitePop)) */
            const uint32_t g = mt_randbelow(&s, (uint32_t)nElite);
            const uint32_t at = mt_randbelow(&s, (uint32_t)nMember); /* randint(0, nMember - 1) */
            uint32_t v;
            memcpy(child, elite + (size_t)g * nMember, (size_t)nMember);
            v = mt_randbelow(&s, (uint32_t)(nType - 1)); /* choice of the nType - 1 other types, in order */
            child[at] = (uint8_t)(v + (v >= child[at] ? 1u : 0u));
            ++c[1];
        } else if (p <= toKeep) {
            memcpy(child, pop + (size_t)j * nMember, (size_t)nMember);
            ++c[2];
        } else { /* GetRandomGene: random.choices(range(nType), k=nMember) */
            const double n = (double)nType;
            for (i = 0; i < nMember; ++i) child[i] = (uint8_t)(int)(mt_random(&s) * n); /* floor: non-negative */
            ++c[3];
        }
    }
    memcpy(state, s.mt, sizeof(s.mt));
    state[MT_N] = (uint32_t)s.pos;
    if (counts) memcpy(counts, c, sizeof(c));
    return 0;
}
