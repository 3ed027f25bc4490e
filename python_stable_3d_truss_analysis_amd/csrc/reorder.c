/*
 * reorder.c - reverse Cuthill-McKee ordering of the joints of every truss of a batch (host side,
 * plain C + OpenMP).  SURVEY.md section 8 f-4: cube trusses are not banded in generator order;
 * renumbering their joints shrinks the row envelope of the reduced stiffness matrix, and with it the
 * 16x16 tiles the factorisation has to touch (csrc/trs_common.h), to ~20-30 % of the dense count.
 *
 * The graph has one node per joint that keeps at least one free DOF and one edge per member between
 * two such joints (a member to a fully pinned joint couples nothing in K_ff).  Per connected
 * component: pseudo-peripheral start node (two BFS sweeps from a minimum-degree node), Cuthill-McKee
 * breadth-first numbering with neighbours taken by ascending degree, the whole order reversed.
 * Fully constrained joints are numbered last.  perm[b][k] = old id of the joint that becomes joint k.
 */
#include <stdint.h>
#include <stdlib.h>

typedef struct {
    int *start, *adj, *deg, *level, *queue, *order, *tmp;
} rcm_scratch_t;

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

static void sort_by_degree(int *a, int n, const int *deg) { /* insertion sort: neighbour lists are short */
    for (int i = 1; i < n; ++i) {
        const int v = a[i];
        int p = i - 1;
        while (p >= 0 && (deg[a[p]] > deg[v] || (deg[a[p]] == deg[v] && a[p] > v))) { a[p + 1] = a[p]; --p; }
        a[p + 1] = v;
    }
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
        sc->adj[sc->tmp[a]++] = b;
        sc->adj[sc->tmp[b]++] = a;
    }
    for (int j = 0; j < nJ; ++j) sort_by_degree(sc->adj + sc->start[j], sc->start[j + 1] - sc->start[j], sc->deg);

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
            int best = sc->queue[begin];
            for (int i = begin; i < count; ++i)
                if (sc->deg[sc->queue[i]] < sc->deg[best]) best = sc->queue[i];
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
        sc.start = (int *)malloc(sizeof(int) * (nJ_max + 2));
        sc.adj = (int *)malloc(sizeof(int) * (2 * nM_max + 2));
        sc.deg = (int *)malloc(sizeof(int) * (nJ_max + 1));
        sc.level = (int *)malloc(sizeof(int) * (nJ_max + 1));
        sc.queue = (int *)malloc(sizeof(int) * (nJ_max + 1));
        sc.order = (int *)malloc(sizeof(int) * (nJ_max + 1));
        sc.tmp = (int *)malloc(sizeof(int) * (nJ_max + 1));
        const int ok = sc.start && sc.adj && sc.deg && sc.level && sc.queue && sc.order && sc.tmp;
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
        free(sc.start); free(sc.adj); free(sc.deg); free(sc.level); free(sc.queue); free(sc.order); free(sc.tmp);
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
