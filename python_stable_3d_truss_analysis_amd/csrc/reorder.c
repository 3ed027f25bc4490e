/*
 * reorder.c - reverse Cuthill-McKee ordering of the joints of every truss of a batch (host side,
 * plain C + OpenMP).  SURVEY.md section 8 f-4: cube trusses are not banded in generator order;
 * renumbering their joints shrinks the row envelope of the reduced stiffness matrix, and with it the
 * 16x16 tiles the factorisation has to touch (csrc/trs_common.h), to ~20-30 % of the dense count.
 *
 * The graph has one node per joint that keeps at least one free DOF and one edge per member between
 * two such joints (a member to a fully pinned joint couples nothing in K_ff).  Per connected
 * component: pseudo-peripheral start node (two BFS sweeps from a minimum-degree node, each restarting from
 * the minimum-(degree, id) joint of the deepest level), Cuthill-McKee
 * breadth-first numbering with neighbours taken by ascending degree, the whole order reversed.
 * Fully constrained joints are numbered last.  perm[b][k] = old id of the joint that becomes joint k.
 */
#include <stdint.h>
#include <stdlib.h>

typedef struct {
    int *start, *adj, *adj2, *deg, *level, *queue, *order, *tmp;
} rcm_scratch_t;

static int scratch_alloc(rcm_scratch_t *sc, int nJ_max, int nM_max) {
    sc->start = (int *)malloc(sizeof(int) * (nJ_max + 2));
    sc->adj = (int *)malloc(sizeof(int) * (2 * (size_t)nM_max + 2));
    sc->adj2 = (int *)malloc(sizeof(int) * (2 * (size_t)nM_max + 2));
    sc->deg = (int *)malloc(sizeof(int) * (nJ_max + 2));
    sc->level = (int *)malloc(sizeof(int) * (nJ_max + 1));
    sc->queue = (int *)malloc(sizeof(int) * (nJ_max + 2));
    sc->order = (int *)malloc(sizeof(int) * (nJ_max + 2));
    sc->tmp = (int *)malloc(sizeof(int) * (nJ_max + 1));
    return sc->start && sc->adj && sc->adj2 && sc->deg && sc->level && sc->queue && sc->order && sc->tmp;
}

static void scratch_free(rcm_scratch_t *sc) {
    free(sc->start); free(sc->adj); free(sc->adj2); free(sc->deg); free(sc->level); free(sc->queue);
    free(sc->order); free(sc->tmp);
}

static int bfs_levels(const rcm_scratch_t *sc, int root, int stamp_base, int *last_level_begin, int *count) {
    /* level[] doubles as the visited stamp: level >= stamp_base means visited in this sweep */
    int head = 0, tail = 0, depth = 0, begin = 0;
    sc->queue[tail++] = root;
    sc->level[root] = stamp_base;
    while (head < tail) {
        const int level_end = tail;
        begin = head;
        for (; head < level_end; ++head) {
            const int v = sc->queue[head];
            for (int e = sc->start[v]; e < sc->start[v + 1]; ++e) {
                const int w = sc->adj[e];
                if (sc->level[w] < stamp_base) {
                    sc->level[w] = stamp_base + depth + 1;
                    sc->queue[tail++] = w;
                }
            }
        }
        ++depth;
    }
    *last_level_begin = begin;
    *count = tail;
    return depth;
}

static void rcm_one(rcm_scratch_t *sc, int nJ, int nM, const int32_t *conn, const uint8_t *cbits, int32_t *perm) {
    /* adjacency of the free joints */
    for (int j = 0; j <= nJ; ++j) sc->start[j] = 0;
    for (int m = 0; m < nM; ++m) {
        const int a = conn[2 * m], b = conn[2 * m + 1];
        if (a == b || (cbits[a] & 7) == 7 || (cbits[b] & 7) == 7) continue;
        ++sc->start[a + 1];
        ++sc->start[b + 1];
    }
    for (int j = 0; j < nJ; ++j) { sc->deg[j] = sc->start[j + 1]; sc->start[j + 1] += sc->start[j]; }
    for (int j = 0; j < nJ; ++j) sc->tmp[j] = sc->start[j];
    for (int m = 0; m < nM; ++m) {
        const int a = conn[2 * m], b = conn[2 * m + 1];
        if (a == b || (cbits[a] & 7) == 7 || (cbits[b] & 7) == 7) continue;
        sc->adj2[sc->tmp[a]++] = b;
        sc->adj2[sc->tmp[b]++] = a;
    }
    /* neighbour lists by ascending (degree, id) without sorting them one by one: the joints in that order
     * (counting sort by degree; degrees are < 2 nM + 1 but lists of trusses are short, so count over the
     * occurring range) append themselves to their neighbours' lists */
    {
        int maxdeg = 0;
        for (int j = 0; j < nJ; ++j) maxdeg = sc->deg[j] > maxdeg ? sc->deg[j] : maxdeg;
        if (maxdeg <= nJ) { /* counters fit the [nJ + 2] scratch arrays */
            int *cnt = sc->queue;
            for (int d = 0; d <= maxdeg + 1; ++d) cnt[d] = 0;
            for (int j = 0; j < nJ; ++j) ++cnt[sc->deg[j] + 1];
            for (int d = 0; d < maxdeg; ++d) cnt[d + 1] += cnt[d];
            for (int j = 0; j < nJ; ++j) sc->order[cnt[sc->deg[j]]++] = j;
        } else { /* many parallel members: plain insertion sort of the joints */
            for (int j = 0; j < nJ; ++j) {
                int q = j - 1;
                while (q >= 0 && sc->deg[sc->order[q]] > sc->deg[j]) { sc->order[q + 1] = sc->order[q]; --q; }
                sc->order[q + 1] = j;
            }
        }
        for (int j = 0; j < nJ; ++j) sc->tmp[j] = sc->start[j];
        for (int i = 0; i < nJ; ++i) {
            const int v = sc->order[i];
            for (int e = sc->start[v]; e < sc->start[v + 1]; ++e) {
                const int w = sc->adj2[e];
                sc->adj[sc->tmp[w]++] = v;
            }
        }
    }

    int n_order = 0, stamp = 1;
    for (int j = 0; j < nJ; ++j) sc->level[j] = 0;
    /* components in order of their minimum-degree unvisited node */
    for (;;) {
        int root = -1;
        for (int j = 0; j < nJ; ++j)
            if ((cbits[j] & 7) != 7 && sc->level[j] == 0 && (root < 0 || sc->deg[j] < sc->deg[root])) root = j;
        if (root < 0) break;
        /* pseudo-peripheral node: restart from a minimum-degree node of the deepest level, twice */
        int begin, count;
        for (int sweep = 0; sweep < 2; ++sweep) {
            stamp += nJ + 2;
            bfs_levels(sc, root, stamp, &begin, &count);
            /* a minimum-degree joint of the deepest level, the smallest id among equals: a rule that does not
             * depend on the order INSIDE the level, so the device version (order.hip) needs no sorted queue for
             * these two sweeps */
            int best = sc->queue[begin];
            for (int i = begin; i < count; ++i) {
                const int q = sc->queue[i];
                if (sc->deg[q] < sc->deg[best] || (sc->deg[q] == sc->deg[best] && q < best)) best = q;
            }
            root = best;
        }
        stamp += nJ + 2;
        bfs_levels(sc, root, stamp, &begin, &count); /* adjacency is degree-sorted: this IS Cuthill-McKee */
        for (int i = 0; i < count; ++i) sc->order[n_order++] = sc->queue[i];
        for (int i = 0; i < count; ++i) sc->level[sc->queue[i]] = 0x40000000; /* permanently visited */
    }
    int k = 0;
    for (int i = n_order - 1; i >= 0; --i) perm[k++] = sc->order[i]; /* reverse */
    for (int j = 0; j < nJ; ++j)
        if ((cbits[j] & 7) == 7) perm[k++] = j;
}

/* perm: [B][nJ_max]; entries k >= nJ[b] are set to k (identity on the padding). */
int trs_rcm_order(int B, int nJ_max, int nM_max, const int32_t *conn, const uint8_t *cbits,
                  const int32_t *nJ, const int32_t *nM, int32_t *perm) {
    int rc = 0;
#pragma omp parallel
    {
        rcm_scratch_t sc;
        const int sc_ok = scratch_alloc(&sc, nJ_max, nM_max);
        const int ok = sc_ok;
        if (!ok) {
#pragma omp critical
            rc = -2;
        }
#pragma omp for schedule(dynamic, 16)
        for (int b = 0; b < B; ++b) {
            if (!ok) continue;
            int32_t *p = perm + (size_t)b * nJ_max;
            rcm_one(&sc, nJ[b], nM[b], conn + (size_t)b * 2 * nM_max, cbits + (size_t)b * nJ_max, p);
            for (int k = nJ[b]; k < nJ_max; ++k) p[k] = k;
        }
        scratch_free(&sc);
    }
    return rc;
}

/* ---- profile order: the cheapest of several candidate orders ---------------------------------------
 *
 * What the factorisation pays for is the row envelope of K_ff at 16-row granularity (trs_common.h): per
 * row chunk q the tiles ft[q] .. q, ft = the first coupled tile made non-decreasing from the bottom
 * (assemble.hip, envelope metadata).  With w_q = q - ft[q] + 1 the matrix-core work of the left-looking
 * factorisation grows like sum w_q^2 and the bytes of every stage like sum w_q; on MI355X one tile of
 * traffic through the four stages costs about as much time as twelve tile updates, hence
 *     cost = sum_q w_q (w_q + 12).
 * Candidates: reverse Cuthill-McKee (above) and its reverse, and coordinate sweeps - the joints sorted
 * lexicographically by their coordinates, binned to a quarter of the mean member length (so that the
 * joints of one lattice plane stay together under small geometric noise), in each of the six axis
 * orders, forwards and backwards.  A frame-like truss swept plane by plane along one axis has the
 * cross-section as its front, which for lattice-like trusses (the reference's cube trusses,
 * generate.py:150-340) is 15-35 % cheaper than the diagonal level sets of Cuthill-McKee.  Every candidate
 * is priced with the same cost and the cheapest wins, so the result is never worse than RCM. */
/* cost of the order `ord` of the nf joints that keep a free DOF (ord[k] = old joint id), over the joint
 * adjacency rcm_one left in sc->start / sc->adj; gives up (returning a value >= bound) as soon as the cost
 * accumulated from the bottom rows reaches `bound`.  Scratch: newidx [nJ], c0 / c1 [nf] first and last row
 * chunk of a joint, cmin [n/16 + 2]. */
static double envelope_cost(const rcm_scratch_t *sc, int nf, const uint8_t *cbits, const int *ord, int *newidx,
                            int *c0, int *c1, int *cmin, double bound) {
    int n = 0;
    for (int k = 0; k < nf; ++k) {
        const int old = ord[k];
        newidx[old] = k;
        c0[k] = n >> 4;
        n += 3 - ((cbits[old] & 1) + ((cbits[old] >> 1) & 1) + ((cbits[old] >> 2) & 1));
        c1[k] = (n - 1) >> 4;
    }
    const int nch = (n + 15) >> 4;
    for (int q = 0; q < nch; ++q) cmin[q] = q;
    double cost = 0.0;
    int run = nch, qdone = nch; /* chunks >= qdone are priced; run = their running minimum (ft of chunk qdone) */
    for (int k = nf - 1; k >= 0; --k) {
        /* every joint with rows in a chunk above c1[k] has been seen: those chunks are final */
        while (qdone > c1[k] + 1) {
            --qdone;
            if (cmin[qdone] < run) run = cmin[qdone];
            const double w = (double)(qdone - run + 1);
            cost += w * (w + 12.0);
        }
        if (cost >= bound) return cost;
        const int old = ord[k];
        int m = k; /* the joint's own rows may straddle two chunks */
        for (int e = sc->start[old]; e < sc->start[old + 1]; ++e) {
            const int w = newidx[sc->adj[e]];
            m = w < m ? w : m;
        }
        const int col = c0[m];
        if (cmin[c0[k]] > col) cmin[c0[k]] = col;
        if (cmin[c1[k]] > col) cmin[c1[k]] = col;
    }
    while (qdone > 0) {
        --qdone;
        if (cmin[qdone] < run) run = cmin[qdone];
        const double w = (double)(qdone - run + 1);
        cost += w * (w + 12.0);
    }
    return cost;
}

/* stable counting sort of ids[0..n) by key[id] in [0, nb): out <- sorted; count: [nb + 1] */
static void counting_pass(const int *ids, int *out, int n, const int *key, int stride, int nb, int *count) {
    for (int i = 0; i <= nb; ++i) count[i] = 0;
    for (int i = 0; i < n; ++i) ++count[key[stride * ids[i]] + 1];
    for (int i = 0; i < nb; ++i) count[i + 1] += count[i];
    for (int i = 0; i < n; ++i) out[count[key[stride * ids[i]]]++] = ids[i];
}

/* perm: [B][nJ_max] as trs_rcm_order; choice (may be NULL): [B] winning candidate - 0 RCM, 1 its reverse,
 * 2 + 2 p + r the sweep with axis order p (0..5: xyz xzy yxz yzx zxy zyx, first axis slowest), r = 1 backwards */
#define TRS_ORDER_RCM_BELOW 128   /* effort 3: free joints below which the Cuthill-McKee candidates are still priced */
int trs_profile_order(int B, int nJ_max, int nM_max, const double *xyz, const int32_t *conn,
                      const uint8_t *cbits, const int32_t *nJ, const int32_t *nM, int32_t *perm,
                      int32_t *choice, int effort) {
    static const int axes[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    const int bin_cap = 4 * nJ_max + 64; /* bins per axis (the bin width grows on an axis that would need more) */
    int rc = 0;
#pragma omp parallel
    {
        rcm_scratch_t sc;
        const int sc_ok = scratch_alloc(&sc, nJ_max, nM_max);
        int *cand = (int *)malloc(sizeof(int) * (nJ_max + 1));
        int *ids = (int *)malloc(sizeof(int) * (nJ_max + 1));
        int *newidx = (int *)malloc(sizeof(int) * (nJ_max + 1));
        int *c0 = (int *)malloc(sizeof(int) * (nJ_max + 1));
        int *c1 = (int *)malloc(sizeof(int) * (nJ_max + 1));
        int *cmin = (int *)malloc(sizeof(int) * (3 * nJ_max / 16 + 4));
        int32_t *rcm = (int32_t *)malloc(sizeof(int32_t) * (nJ_max + 1));
        int *bins = (int *)malloc(sizeof(int) * 3 * (nJ_max + 1));
        int *count = (int *)malloc(sizeof(int) * (bin_cap + 2));
        const int ok = sc_ok && cand && ids &&
                       newidx && c0 && c1 && cmin && rcm && bins && count;
        if (!ok) {
#pragma omp critical
            rc = -2;
        }
#pragma omp for schedule(dynamic, 16)
        for (int b = 0; b < B; ++b) {
            if (!ok) continue;
            const int nj = nJ[b], nm = nM[b];
            const int32_t *cn = conn + (size_t)b * 2 * nM_max;
            const uint8_t *cb = cbits + (size_t)b * nJ_max;
            const double *X = xyz + (size_t)b * 3 * nJ_max;
            int32_t *p = perm + (size_t)b * nJ_max;
            rcm_one(&sc, nj, nm, cn, cb, rcm); /* leaves the adjacency of the free joints in sc.start / sc.adj */
            int nf = 0; /* joints with a free DOF: they come first in every candidate, the others keep RCM's tail */
            for (int j = 0; j < nj; ++j) nf += (cb[j] & 7) != 7;
            for (int k = 0; k < nj; ++k) p[k] = rcm[k];
            /* coordinate bins: a quarter of the mean member length */
            double len2 = 0.0; /* root mean square member length (one square root per truss) */
            int nlen = 0;
            for (int m = 0; m < nm; ++m) {
                const double *pa = X + 3 * cn[2 * m], *pb = X + 3 * cn[2 * m + 1];
                const double dx = pb[0] - pa[0], dy = pb[1] - pa[1], dz = pb[2] - pa[2];
                const double l2 = dx * dx + dy * dy + dz * dz;
                if (l2 > 0.0) { len2 += l2; ++nlen; }
            }
            /* rounded to single precision: the device version (order.hip) sums the squares in another order,
             * and the bins - hence the permutation - must not depend on the last bits of that sum */
            const double h = nlen ? (double)(float)(0.25 * __builtin_sqrt(len2 / nlen)) : 0.0;
            const int sweeps = h > 0.0 && h < 1e300 && nf > 1 && effort >= 1;
            /* effort 3: the Cuthill-McKee candidates are priced for SMALL trusses only (fewer than
             * TRS_ORDER_RCM_BELOW free joints): on larger lattice-like trusses a sweep
             * wins (bar-942; every cube truss from 140 cubes up, 97 % of those from 80) and Cuthill-McKee - three
             * breadth-first sweeps - is 40 % of the device kernel's time.  The same rule in csrc/order.hip. */
            const int use_rcm = effort < 3 || nf < TRS_ORDER_RCM_BELOW;
            double best = 1e300, c;
            int best_id = 0;
            if (use_rcm) {
                for (int k = 0; k < nf; ++k) cand[k] = rcm[k];
                best = envelope_cost(&sc, nf, cb, cand, newidx, c0, c1, cmin, 1e300);
                /* candidate 1: plain Cuthill-McKee (RCM backwards) */
                for (int k = 0; k < nf; ++k) cand[k] = rcm[nf - 1 - k];
                c = envelope_cost(&sc, nf, cb, cand, newidx, c0, c1, cmin, best);
                if (c < best) {
                    best = c; best_id = 1;
                    for (int k = 0; k < nf; ++k) p[k] = cand[k];
                }
            }
            if (!use_rcm && !sweeps) { /* nothing to choose from: the free joints in the given order (choice 14) */
                int nk = 0;
                for (int j = 0; j < nj; ++j)
                    if ((cb[j] & 7) != 7) p[nk++] = j;
                best_id = 14;
            }
            if (sweeps) {
                int nb[3];
                for (int a = 0; a < 3; ++a) {
                    double lo = X[a], hi = X[a];
                    for (int j = 1; j < nj; ++j) {
                        lo = X[3 * j + a] < lo ? X[3 * j + a] : lo;
                        hi = X[3 * j + a] > hi ? X[3 * j + a] : hi;
                    }
                    double ha = h;
                    if (!((hi - lo) / ha < (double)(bin_cap - 2))) ha = (hi - lo) / (double)(bin_cap - 2);
                    nb[a] = 1;
                    for (int j = 0; j < nj; ++j) {
                        double q = ha > 0.0 ? (X[3 * j + a] - lo) / ha + 0.5 : 0.0;
                        if (!(q >= 0.0)) q = 0.0; /* NaN coordinates: any bin */
                        if (q > (double)(bin_cap - 1)) q = (double)(bin_cap - 1);
                        bins[3 * j + a] = (int)q;
                        if ((int)q + 1 > nb[a]) nb[a] = (int)q + 1;
                    }
                }
                int nk = 0;
                for (int j = 0; j < nj; ++j)
                    if ((cb[j] & 7) != 7) ids[nk++] = j;
                /* the sweep along the longest extent first (most bins slowest): it usually wins, and the
                 * others then give up early against its cost */
                int first_ax = 0;
                for (int ax = 1; ax < 6; ++ax) {
                    const int *A = axes[ax], *F = axes[first_ax];
                    if (nb[A[0]] > nb[F[0]] || (nb[A[0]] == nb[F[0]] && nb[A[1]] > nb[F[1]])) first_ax = ax;
                }
                for (int i = 0; i < (effort >= 2 ? 6 : (effort == 1 ? 1 : 0)); ++i) {
                    /* lexicographic by (axis 0, axis 1, axis 2, joint id): three stable passes, last key first */
                    const int ax = i == 0 ? first_ax : (i <= first_ax ? i - 1 : i);
                    const int *A = axes[ax];
                    counting_pass(ids, cand, nk, bins + A[2], 3, nb[A[2]], count);
                    counting_pass(cand, sc.order, nk, bins + A[1], 3, nb[A[1]], count);
                    counting_pass(sc.order, cand, nk, bins + A[0], 3, nb[A[0]], count);
                    for (int rev = 0; rev < 2; ++rev) {
                        if (rev)
                            for (int k = 0; k < nk / 2; ++k) {
                                const int t = cand[k];
                                cand[k] = cand[nk - 1 - k];
                                cand[nk - 1 - k] = t;
                            }
                        c = envelope_cost(&sc, nf, cb, cand, newidx, c0, c1, cmin, best);
                        if (c < best) {
                            best = c; best_id = 2 + 2 * ax + rev;
                            for (int k = 0; k < nf; ++k) p[k] = cand[k];
                        }
                    }
                }
            }
            for (int k = nj; k < nJ_max; ++k) p[k] = k;
            if (choice) choice[b] = best_id;
        }
        scratch_free(&sc);
        free(cand); free(ids); free(newidx); free(c0); free(c1); free(cmin); free(rcm); free(bins); free(count);
    }
    return rc;
}

/* ---- envelope reach ---------------------------------------------------------------------------------
 * How far the row envelope of K_ff reaches below its 64 x 64 diagonal blocks, in 16-row chunks, for the
 * numbering the trusses come in: the same metadata trs_assemble derives on the device (chunkmin from every
 * joint's smallest coupled free DOF, ft = its running minimum from the end, last chunk reaching each panel).
 * A caller whose batch stays below TRS_NARROW_MAX_BELOW (24) everywhere may tell the solver that it holds
 * no wide matrix (TRS_ASM_ALL_NARROW + TRS_HINT_NO_WIDE in trs_solver.h), which saves the launches of the
 * kernels that would find nothing to do.  The hint is safe either way: with TRS_ASM_ALL_NARROW the device
 * routes every matrix to the wave-per-matrix kernels regardless of what this function found. */
int trs_envelope_reach(int B, int nJ_max, int nM_max, const int32_t *conn, const uint8_t *cbits,
                       const int32_t *nJ, const int32_t *nM, const int32_t *perm /* [B][nJ_max] or NULL */,
                       int32_t *reach /* [B] */) {
    int rc = 0;
#pragma omp parallel
    {
        int *first = (int *)malloc(sizeof(int) * (size_t)(nJ_max + 1));   /* first free DOF of a joint, -1: none */
        int *mincol = (int *)malloc(sizeof(int) * (size_t)(nJ_max + 1));
        int *cmin = (int *)malloc(sizeof(int) * (size_t)(3 * nJ_max / 16 + 8));
        int *lastc = (int *)malloc(sizeof(int) * (size_t)(3 * nJ_max / 16 + 8));
        const int ok = first && mincol && cmin && lastc;
        if (!ok) {
#pragma omp critical
            rc = -2;
        }
#pragma omp for schedule(static)
        for (int b = 0; b < B; ++b) {
            if (!ok) continue;
            const int nj = nJ[b], nm = nM[b];
            const int32_t *cn = conn + (size_t)b * 2 * nM_max;
            const uint8_t *cb = cbits + (size_t)b * nJ_max;
            const int32_t *pm = perm ? perm + (size_t)b * nJ_max : NULL;  /* joint k of the solve = joint pm[k] here */
            int n = 0;
            for (int k = 0; k < nj; ++k) {
                const int j = pm ? pm[k] : k;
                const int nf = 3 - ((cb[j] & 1) + ((cb[j] >> 1) & 1) + ((cb[j] >> 2) & 1));
                first[j] = nf ? n : -1;
                mincol[j] = nf ? n : 0x7fffffff;
                n += nf;
            }
            for (int m = 0; m < nm; ++m) { /* a joint's rows reach back to the smallest free DOF of any neighbour */
                const int a = cn[2 * m], c = cn[2 * m + 1];
                if (first[c] >= 0 && first[c] < mincol[a]) mincol[a] = first[c];
                if (first[a] >= 0 && first[a] < mincol[c]) mincol[c] = first[a];
            }
            const int npad = (n + 63) / 64 * 64, nch = npad / 16;
            for (int q = 0; q < nch; ++q) cmin[q] = q;
            for (int k = 0, dof = 0; k < nj; ++k) {
                const int j = pm ? pm[k] : k;
                if (first[j] < 0) continue;
                const int nf = 3 - ((cb[j] & 1) + ((cb[j] >> 1) & 1) + ((cb[j] >> 2) & 1));
                for (int s = 0; s < nf; ++s, ++dof)
                    if (mincol[j] / 16 < cmin[dof / 16]) cmin[dof / 16] = mincol[j] / 16;
            }
            for (int q = nch - 2; q >= 0; --q)
                if (cmin[q + 1] < cmin[q]) cmin[q] = cmin[q + 1]; /* ft */
            for (int q = 0; q < nch; ++q) { /* chunk q owns the tiles ft[q] .. ft[q+1]-1 */
                const int hi = q + 1 < nch ? cmin[q + 1] : nch;
                for (int t = cmin[q]; t < hi; ++t) lastc[t] = q;
            }
            int widest = 0;
            for (int j = 0; j < nch / 4; ++j)
                if (lastc[4 * j + 3] - (4 * j + 3) > widest) widest = lastc[4 * j + 3] - (4 * j + 3);
            reach[b] = widest;
        }
        free(first); free(mincol); free(cmin); free(lastc);
    }
    return rc;
}

/* Apply a joint order: joint k of the output is joint perm[b][k] of the input (members keep their
 * order, their end joints are renumbered; padding members stay (0, 0)).  Out-of-place. */
int trs_apply_joint_order(int B, int nJ_max, int nM_max, const int32_t *perm, const int32_t *nM,
                          const double *xyz, const int32_t *conn, const uint8_t *cbits,
                          const double *loads, double *xyz_out, int32_t *conn_out, uint8_t *cbits_out,
                          double *loads_out) {
    int rc = 0;
#pragma omp parallel
    {
        int *inverse = (int *)malloc(sizeof(int) * (size_t)(nJ_max + 1));
        if (!inverse) {
#pragma omp atomic write
            rc = -1;
        } else {
#pragma omp for schedule(static)
            for (int b = 0; b < B; ++b) {
                const int32_t *p = perm + (size_t)b * nJ_max;
                for (int k = 0; k < nJ_max; ++k) {
                    const int old = p[k];
                    inverse[old] = k;
                    for (int a = 0; a < 3; ++a) {
                        xyz_out[((size_t)b * nJ_max + k) * 3 + a] = xyz[((size_t)b * nJ_max + old) * 3 + a];
                        loads_out[((size_t)b * nJ_max + k) * 3 + a] = loads[((size_t)b * nJ_max + old) * 3 + a];
                    }
                    cbits_out[(size_t)b * nJ_max + k] = cbits[(size_t)b * nJ_max + old];
                }
                for (int m = 0; m < nM_max; ++m) {
                    const size_t e = ((size_t)b * nM_max + m) * 2;
                    const int live = m < nM[b];
                    conn_out[e] = live ? inverse[conn[e]] : 0;
                    conn_out[e + 1] = live ? inverse[conn[e + 1]] : 0;
                }
            }
            free(inverse);
        }
    }
    return rc;
}
