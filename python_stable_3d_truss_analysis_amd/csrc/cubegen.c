/*
 * cubegen.c - native generator of random "cube trusses", written straight into the padded batch
 * arrays of the solver (host side, plain C).
 *
 * Same construction and output schema as the reference's GenerateRandomCubeTrusses
 * (slientruss3d/generate.py:152-376), with its own RNG (the parity contract is on solver outputs
 * given inputs, and on the distribution of sizes, not on the Python `random` stream):
 *   - a polycube is grown on an integer grid from a random free cell; the frontier is popped
 *     DFS / BFS / at random (generate.py:266-286), neighbours pushed in shuffled order;
 *   - joints are the cube vertices, numbered in first-seen order, vertex v of a cube at
 *     (x + (v&1), y + ((v>>1)&1), z + ((v>>2)&1)) (generate.py:168-184);
 *   - each cube links 6 face diagonals (one, the other, or both per face: LinkType) and its 12
 *     edges, skipping ordered joint pairs already linked unless parallel members are allowed
 *     (generate.py:186-231);
 *   - joints on the lowest occupied z layer are PIN supports (generate.py:288-298);
 *   - 1..|free joints| random loads on unsupported joints (generate.py:318-328);
 *   - a member type per member drawn from the given table (generate.py:330-336);
 *   - trusses failing the counting test nM + 3 nPin >= 3 nJ (and nRes >= 6) are regenerated
 *     (generate.py:344-374, truss.py:158-164).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint64_t s; } rng_t;

static uint64_t rng_next(rng_t *r) { /* splitmix64 */
    uint64_t z = (r->s += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
/* splitmix64's output function as a stand-alone hash */
static uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
static double rng_unit(rng_t *r) { return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }
static int rng_below(rng_t *r, int n) { return (int)(rng_unit(r) * n); } /* 0 .. n-1 */
static double rng_uniform(rng_t *r, double lo, double hi) { return lo + (hi - lo) * rng_unit(r); }

enum { METHOD_DFS = 0, METHOD_BFS = 1, METHOD_RANDOM = 2 };
enum { LINK_LBRT = 0, LINK_RBLT = 1, LINK_CROSS = 2, LINK_RANDOM = 3 };

/* the six faces: the two diagonals of each, as local vertex pairs (generate.py:210-215) */
static const int FACE_DIAG[6][2][2] = {
    {{0, 5}, {1, 4}}, {{1, 7}, {3, 5}}, {{3, 6}, {2, 7}}, {{2, 4}, {0, 6}}, {{4, 7}, {5, 6}}, {{0, 3}, {1, 2}}};
/* the twelve edges (generate.py:217-224) */
static const int EDGE[12][2] = {{4, 5}, {5, 7}, {6, 7}, {4, 6}, {0, 1}, {0, 2}, {1, 3}, {2, 3},
                                {0, 4}, {1, 5}, {2, 6}, {3, 7}};

typedef struct {
    int gx, gy, gz;
    int *cell_state;  /* 0 free, 1 pending in the frontier, 2 used */
    int *vertex_id;   /* (gx+1)(gy+1)(gz+1) -> joint id or -1 */
    int *frontier;    /* cell indices, as a deque in a ring-free array (front index moves) */
    uint64_t *linked; /* open-addressing set of ordered joint pairs */
    int *linked_slots; /* slots filled during the running attempt (cleared one by one afterwards) */
    int linked_cap, n_linked;
} scratch_t;

static int set_insert(scratch_t *sc, uint64_t key) { /* 1 if newly inserted */
    uint64_t h = key * 0x9e3779b97f4a7c15ULL;
    int cap = sc->linked_cap, i = (int)(h >> 40) & (cap - 1);
    for (;;) {
        if (sc->linked[i] == 0) { sc->linked[i] = key; sc->linked_slots[sc->n_linked++] = i; return 1; }
        if (sc->linked[i] == key) return 0;
        i = (i + 1) & (cap - 1);
    }
}

/* One attempt.  Returns 0 on success, 1 if the truss fails the counting test, -1 if it does
 * not fit nJ_max / nM_max. */
static int generate_one(scratch_t *sc, rng_t *rng, int num_cube, int method, int link_type,
                        int allow_parallel, const double len[3], const double *force_range,
                        int nforce_lo, int nforce_hi, const double *mtypes, int n_types, int nJ_max,
                        int nM_max, double *xyz, int32_t *conn, double *E, double *A, double *rho,
                        uint8_t *cbits, double *loads, int32_t *nJ_out, int32_t *nM_out) {
    const int gx = sc->gx, gy = sc->gy, gz = sc->gz, ncell = gx * gy * gz;
    const int vx = gx + 1, vy = gy + 1, nvert = vx * vy * (gz + 1);
    memset(sc->cell_state, 0, sizeof(int) * ncell);
    for (int i = 0; i < nvert; ++i) sc->vertex_id[i] = -1;
    for (int i = 0; i < sc->n_linked; ++i) sc->linked[sc->linked_slots[i]] = 0;
    sc->n_linked = 0;

    int front = 0, back = 0, n_joint = 0, n_member = 0, n_cube = 0, min_z = gz + 1;
    int start = rng_below(rng, ncell);
    sc->frontier[back++] = start;
    sc->cell_state[start] = 1;
    /* joint grid coordinates, kept to place the joints once min_z is known */
    int *jx = (int *)xyz; /* reuse the output buffer as scratch: 3 ints per joint fit in 3 doubles */

    while (n_cube < num_cube && front < back) {
        int take_back;
        if (method == METHOD_DFS) take_back = 1;
        else if (method == METHOD_BFS) take_back = 0;
        else take_back = rng_unit(rng) <= 0.5;
        int cell;
        if (take_back) cell = sc->frontier[--back];
        else cell = sc->frontier[front++];
        sc->cell_state[cell] = 2;
        const int cx = cell % gx, cy = (cell / gx) % gy, cz = cell / (gx * gy);
        /* push the free neighbours in random order */
        int nb[6], nnb = 0;
        static const int DIR[6][3] = {{-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};
        for (int d = 0; d < 6; ++d) {
            const int x = cx + DIR[d][0], y = cy + DIR[d][1], z = cz + DIR[d][2];
            if (x < 0 || x >= gx || y < 0 || y >= gy || z < 0 || z >= gz) continue;
            const int c = (z * gy + y) * gx + x;
            if (sc->cell_state[c] == 0) nb[nnb++] = c;
        }
        for (int i = nnb - 1; i > 0; --i) { /* Fisher-Yates */
            const int j = rng_below(rng, i + 1), t = nb[i];
            nb[i] = nb[j]; nb[j] = t;
        }
        for (int i = 0; i < nnb; ++i) { sc->frontier[back++] = nb[i]; sc->cell_state[nb[i]] = 1; }

        /* the cube's vertices -> joint ids */
        int vid[8];
        for (int v = 0; v < 8; ++v) {
            const int x = cx + (v & 1), y = cy + ((v >> 1) & 1), z = cz + ((v >> 2) & 1);
            const int vi = (z * vy + y) * vx + x;
            if (sc->vertex_id[vi] < 0) {
                if (n_joint >= nJ_max) return -1;
                sc->vertex_id[vi] = n_joint;
                jx[3 * n_joint] = x; jx[3 * n_joint + 1] = y; jx[3 * n_joint + 2] = z;
                if (z < min_z) min_z = z;
                ++n_joint;
            }
            vid[v] = sc->vertex_id[vi];
        }
        /* members: face diagonals first, then the edges (generate.py:208-229) */
        int pairs[24][2], npairs = 0;
        for (int f = 0; f < 6; ++f) {
            int choice = link_type == LINK_RANDOM ? rng_below(rng, 3) : link_type;
            if (choice == 0 || choice == 2) { pairs[npairs][0] = FACE_DIAG[f][0][0]; pairs[npairs++][1] = FACE_DIAG[f][0][1]; }
            if (choice == 1 || choice == 2) { pairs[npairs][0] = FACE_DIAG[f][1][0]; pairs[npairs++][1] = FACE_DIAG[f][1][1]; }
        }
        for (int e = 0; e < 12; ++e) { pairs[npairs][0] = EDGE[e][0]; pairs[npairs++][1] = EDGE[e][1]; }
        for (int p = 0; p < npairs; ++p) {
            const int a = vid[pairs[p][0]], b = vid[pairs[p][1]];
            if (!(allow_parallel & 1) && !set_insert(sc, ((uint64_t)(a + 1) << 32) | (uint64_t)(b + 1))) continue;
            if (n_member >= nM_max) return -1;
            conn[2 * n_member] = a; conn[2 * n_member + 1] = b;
            ++n_member;
        }
        ++n_cube;
    }

    /* joints: positions and supports (generate.py:288-298) */
    int n_pin = 0;
    for (int j = n_joint - 1; j >= 0; --j) { /* backwards: ints are read before doubles overwrite */
        const int x = jx[3 * j], y = jx[3 * j + 1], z = jx[3 * j + 2];
        const int pin = z == min_z && !(allow_parallel & 2); /* bit 1: isAddPinSupport=False */
        cbits[j] = pin ? 7 : 0;
        n_pin += pin;
        /* cannot write xyz[3j..] yet for j < n_joint: the int scratch of later joints lives there */
        loads[3 * j] = (double)x; loads[3 * j + 1] = (double)y; loads[3 * j + 2] = (double)(z - min_z);
    }
    for (int j = 0; j < n_joint; ++j) {
        xyz[3 * j] = loads[3 * j] * len[0];
        xyz[3 * j + 1] = loads[3 * j + 1] * len[1];
        xyz[3 * j + 2] = loads[3 * j + 2] * len[2];
        loads[3 * j] = loads[3 * j + 1] = loads[3 * j + 2] = 0.0;
    }
    for (int j = n_joint; j < nJ_max; ++j) {
        xyz[3 * j] = xyz[3 * j + 1] = xyz[3 * j + 2] = 0.0;
        loads[3 * j] = loads[3 * j + 1] = loads[3 * j + 2] = 0.0;
        cbits[j] = 0;
    }
    /* counting test (truss.py:158-164) */
    const int n_res = 3 * n_pin;
    if (!(allow_parallel & 2) && (n_res < 6 || n_member + n_res < 3 * n_joint)) return 1;

    /* loads on unsupported joints (generate.py:318-328) */
    const int n_free_joint = n_joint - n_pin;
    if (n_free_joint > 0) {
        int lo = nforce_lo < 1 ? 1 : nforce_lo, hi = nforce_hi < 0 || nforce_hi > n_free_joint ? n_free_joint : nforce_hi;
        if (lo > hi) lo = hi;
        int n_force = lo + rng_below(rng, hi - lo + 1);
        /* selection sampling of n_force joints out of the n_free_joint unsupported ones */
        int seen = 0, need = n_force;
        for (int j = 0; j < n_joint && need > 0; ++j) {
            if (cbits[j]) continue;
            if (rng_unit(rng) * (n_free_joint - seen) < need) {
                for (int a = 0; a < 3; ++a) loads[3 * j + a] = rng_uniform(rng, force_range[2 * a], force_range[2 * a + 1]);
                --need;
            }
            ++seen;
        }
    }
    /* member types (generate.py:330-336): rows of mtypes are (a, e, density) */
    for (int m = 0; m < n_member; ++m) {
        const double *t = mtypes + 3 * rng_below(rng, n_types);
        A[m] = t[0]; E[m] = t[1]; rho[m] = t[2];
    }
    for (int m = n_member; m < nM_max; ++m) {
        conn[2 * m] = conn[2 * m + 1] = 0;
        A[m] = 1.0; E[m] = 1.0; rho[m] = 0.0;
    }
    *nJ_out = n_joint;
    *nM_out = n_member;
    return 0;
}

/* Upper bounds for a grid: every vertex a joint; 24 member candidates per cube. */
int trs_cubegen_bounds(int gx, int gy, int gz, int max_cubes, int allow_parallel, int *nJ_max, int *nM_max) {
    const int cells = gx * gy * gz, cubes = max_cubes < cells ? max_cubes : cells;
    int nj = 8 * cubes, nv = (gx + 1) * (gy + 1) * (gz + 1);
    *nJ_max = nj < nv ? nj : nv;
    *nM_max = 24 * cubes;
    (void)allow_parallel;
    return 0;
}

/*
 * Generate B trusses into the padded batch arrays.  num_cubes[b] is the polycube size of truss b.
 * With xyz == NULL only the sizes nJ[b], nM[b] are produced (same RNG streams, thread-local
 * scratch): callers size the batch arrays by a first sizes-only pass, then generate for real.
 * Returns 0, or -1 when a truss does not fit nJ_max / nM_max, or -2 on allocation failure.
 * retries_out (optional) receives the number of regenerated (count-unstable) attempts.
 */
int trs_cubegen(int B, uint64_t seed, int gx, int gy, int gz, const int32_t *num_cubes, int method,
                int link_type, int allow_parallel, double len_lo, double len_hi,
                const double *force_range /* [3][2] */, int nforce_lo, int nforce_hi,
                const double *mtypes /* [n_types][3] = (a, e, density) */, int n_types, int nJ_max,
                int nM_max, double *xyz, int32_t *conn, double *E, double *A, double *rho,
                uint8_t *cbits, double *loads, int32_t *nJ, int32_t *nM, int64_t *retries_out,
                int64_t first_index) {
    const int ncell = gx * gy * gz, nvert = (gx + 1) * (gy + 1) * (gz + 1);
    int cap = 64;
    while (cap < 64 * ncell) cap <<= 1; /* >= 2x the 24 pairs per cube */
    int64_t retries = 0;
    int rc = 0;
    /* every truss has its own RNG stream, so the result does not depend on the thread count */
#pragma omp parallel reduction(+ : retries)
    {
        scratch_t sc;
        sc.gx = gx; sc.gy = gy; sc.gz = gz;
        sc.linked_cap = cap;
        sc.n_linked = 0;
        sc.cell_state = (int *)malloc(sizeof(int) * ncell);
        sc.vertex_id = (int *)malloc(sizeof(int) * nvert);
        sc.frontier = (int *)malloc(sizeof(int) * (ncell + 8));
        sc.linked = (uint64_t *)calloc(cap, sizeof(uint64_t));
        sc.linked_slots = (int *)malloc(sizeof(int) * (24 * ncell + 8));
        const int sizes_only = xyz == NULL;
        double *t_xyz = NULL, *t_loads = NULL, *t_E = NULL, *t_A = NULL, *t_rho = NULL;
        int32_t *t_conn = NULL;
        uint8_t *t_cbits = NULL;
        if (sizes_only) { /* one truss worth of scratch per thread */
            t_xyz = (double *)malloc(sizeof(double) * 3 * nJ_max);
            t_loads = (double *)malloc(sizeof(double) * 3 * nJ_max);
            t_E = (double *)malloc(sizeof(double) * 3 * nM_max);
            t_A = t_E + nM_max;
            t_rho = t_E + 2 * nM_max;
            t_conn = (int32_t *)malloc(sizeof(int32_t) * 2 * nM_max);
            t_cbits = (uint8_t *)malloc(nJ_max);
        }
        const int ok = sc.cell_state && sc.vertex_id && sc.frontier && sc.linked && sc.linked_slots &&
                       (!sizes_only || (t_xyz && t_loads && t_E && t_conn && t_cbits));
        if (!ok) {
#pragma omp critical
            rc = -2;
        }
#pragma omp for schedule(dynamic, 16)
        for (int b = 0; b < B; ++b) {
            if (!ok || rc != 0) continue;
            rng_t rng;
            /* Per-truss stream: the initial state is a HASH of (seed, global index of the truss = first_index + b),
             * so a dataset generated in chunks or shards is the same dataset.  (An affine function of b with
             * the generator's own increment as the factor would make truss b+1's stream truss b's
             * shifted by one draw.) */
            rng.s = mix64(mix64(seed + 0x632be59bd9b4e019ULL) ^ mix64((uint64_t)first_index + (uint64_t)b + 1));
            for (;;) {
                double len[3];
                for (int a = 0; a < 3; ++a) len[a] = rng_uniform(&rng, len_lo, len_hi);
                int r = sizes_only
                    ? generate_one(&sc, &rng, num_cubes[b], method, link_type, allow_parallel, len,
                                   force_range, nforce_lo, nforce_hi, mtypes, n_types, nJ_max, nM_max,
                                   t_xyz, t_conn, t_E, t_A, t_rho, t_cbits, t_loads, nJ + b, nM + b)
                    : generate_one(&sc, &rng, num_cubes[b], method, link_type, allow_parallel, len,
                                   force_range, nforce_lo, nforce_hi, mtypes, n_types, nJ_max, nM_max,
                                   xyz + (size_t)b * 3 * nJ_max, conn + (size_t)b * 2 * nM_max,
                                   E + (size_t)b * nM_max, A + (size_t)b * nM_max, rho + (size_t)b * nM_max,
                                   cbits + (size_t)b * nJ_max, loads + (size_t)b * 3 * nJ_max, nJ + b, nM + b);
                if (r == 0) break;
                if (r < 0) {
#pragma omp critical
                    rc = -1;
                    break;
                }
                ++retries;
            }
        }
        free(sc.cell_state); free(sc.vertex_id); free(sc.frontier); free(sc.linked); free(sc.linked_slots);
        free(t_xyz); free(t_loads); free(t_E); free(t_conn); free(t_cbits);
    }
    if (retries_out) *retries_out = retries;
    return rc;
}
