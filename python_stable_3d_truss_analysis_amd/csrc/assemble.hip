// Assembly of the reduced stiffness slab (include/trs_solver.h) for a batch of trusses.
//
// Replaces Member.k / cosines / matK (slientruss3d/truss.py:56-86), Truss.GetKMatrix
// (truss.py:307-316), GetExternalForceVector (truss.py:303-304) and the row/column elimination
// matK[mask,:][:,mask], vecF[mask] (truss.py:343).
//
// One kernel, one work-group per truss, no atomics on floating-point data, bit-reproducible, and no
// intermediate in HBM: the only global traffic is the truss's inputs (read once) and its slab rows
// (written once).
//
//  phase 0 (edge-parallel, all in LDS): every member's k, c = (x1 - x0)/L computed once (one thread
//     per member); a sorted joint adjacency (integer counting sort + per-joint insertion sort by
//     (other joint, member)); one thread per joint sums its diagonal 3x3 block IN THAT FIXED ORDER
//     and finds the joint's smallest coupled column (the envelope, trs_common.h).
//
//  phase 1 (owner-computes, HBM-write bound): the work-group walks the slab TR rows at a time.  A
//     row's threads walk its joint's adjacency list; the head of every run of parallel members forms
//     that neighbour's block row from the cached k, c and drops it into an LDS tile (every
//     (row, column) is written by exactly one thread), then each row goes to HBM exactly once with
//     16-byte coalesced stores, the tile being zeroed behind the reads.  Rows wider than the tile
//     are written in column segments.
//
//  phase 1c (narrow envelopes, the default for them): instead of slab rows the matrix leaves the kernel
//     as compact per-tile entry lists (trs_common.h, TrsCompactLayout) and the load vector goes to `uf`:
//     two threads per joint walk its adjacency list once to count and once to emit the joint's 3x3
//     blocks (the values and their summation order are those of phase 1, bit for bit), entries
//     counting-sorted by tile through LDS counters.  The fused factorisation (trs_potrf_fused_kernel)
//     forms the tiles from these lists where it consumes them: K_ff never exists densely in HBM.
#include "trs_common.h"
#include "../../include/trs_solver.h"

// Diagnostic builds only (-DTRS_ASM_STOP_AFTER=k via tools/build_variants.sh, tools/asm_phases.py): the kernel ends
// behind phase k (wrong results by design), so that timing the truncated kernels gives the CUMULATIVE time of the phases
// without an instrument inside them (wave-cycle stamps were tried first: their registers spill in the 1024-thread
// instance and the row loop then takes 8 x as long).  The product library compiles nothing of this.
#ifdef TRS_ASM_STOP_AFTER
#define TRS_ASM_PHASE_END(k) do { if (TRS_ASM_STOP_AFTER == (k)) return; } while (0)
#else
#define TRS_ASM_PHASE_END(k) do { } while (0)
#endif

namespace {

#ifndef TRS_ASM_THREADS
#define TRS_ASM_THREADS 512
#endif
constexpr int NT_DEFAULT = TRS_ASM_THREADS;  // threads per work-group (two work-groups per CU)
#ifndef TRS_ASM_BIG_THREADS
#define TRS_ASM_BIG_THREADS 1024
#endif
constexpr int NT_BIG = TRS_ASM_BIG_THREADS;  // ... when a truss needs a whole CU's LDS (one per CU)
#ifndef TRS_ASM_TPR
#define TRS_ASM_TPR 16
#endif
constexpr int TPR = TRS_ASM_TPR;     // threads per slab row (64 / TPR rows per wave)
static_assert(64 % TPR == 0 && 16 % (64 / TPR) == 0, "a wave's rows must lie in one 16-row chunk");
__host__ __device__ constexpr int tile_rows(int nt) { return nt / TPR; }  // slab rows per block

// LDS carve-up shared by host and device (bytes, every part 16-byte aligned)
struct AsmLds {
    size_t geom, diag, rhs, tile, ints, total;
};
__host__ __device__ inline AsmLds asm_lds_layout(int nJ_max, int nM_max, int n_pad_max, int WT,
                                                 int geom_in_lds, int TR) {
    AsmLds l;
    l.geom = 0;                                                  // double[nM_max][4]: k | c
    l.diag = l.geom + (geom_in_lds ? (size_t)nM_max * 32 : 0);   // double[nJ_max][6]
    l.rhs = l.diag + (size_t)nJ_max * 48;                        // double[n_pad_max]
    l.tile = l.rhs + (size_t)n_pad_max * 8;                      // double[TR][WT + 16]
    l.ints = l.tile + (size_t)TR * (WT + 16) * 8;
    const size_t nints = (size_t)6 * nJ_max + 1 + 2 * (size_t)nM_max + 3 * (n_pad_max / 16) + n_pad_max + 4;
    l.total = (l.ints + nints * 4 + 15) / 16 * 16;
    return l;
}
// LDS tables of the compact path (ints: tile begin, tile fill cursor, tbase); they alias the row tile,
// which that path does not use
__host__ __device__ inline size_t asm_compact_tab_bytes(int n_pad_max) {
    const int nch = n_pad_max / 16;
    return ((size_t)2 * nch * (TRS_NARROW_MAX_BELOW + 4) + nch + 1) * 4;
}

// MODE fixes, at compile time, the address space of the per-truss tables (LDS loads, not flat ones):
//   0  everything in LDS                              (bar-942: 64 KB + the row tile)
//   1  member geometry in the truss's workspace, the rest in LDS
//   2  everything but the row tile in the workspace   (trusses whose tables exceed a CU's LDS:
//      more than ~7000 members; the tables then live in L2 / HBM and the kernel is slower)
template <int MODE, int NT>
__global__ __launch_bounds__(NT) void trs_assemble_kernel(
    const double* __restrict__ xyz, const TrsMembers mem, const double* __restrict__ loads,
    const int* __restrict__ free_index, const int* __restrict__ n_free, const int* __restrict__ nJ_arr,
    const int* __restrict__ nM_arr, const int nJ_max, const int nM_max, const int n_pad_max,
    const int ld, const size_t slab_stride, double* __restrict__ S_all, const int flags,
    unsigned char* __restrict__ work_all, const size_t work_stride, int* __restrict__ env_all,
    const int WT, double* __restrict__ uf_all, const int ld_uf, const size_t ck_off, const int compact_ok,
    const int WTn) {
    extern __shared__ unsigned char lds_raw[];
    constexpr int TR = tile_rows(NT);
    const int b = blockIdx.x, tid = threadIdx.x;
    const int nJ = nJ_arr[b], nM = nM_arr[b];
    const int n = n_free[b];
    const int npad = trs_round_up(n, TRS_NB);
    if (npad == 0) return;
    const int nch = npad / 16;

    const AsmLds lay = asm_lds_layout(nJ_max, nM_max, n_pad_max, MODE == 2 ? -16 : WT, MODE != 1, TR);
    unsigned char* wbase = work_all + (size_t)b * work_stride;
    unsigned char* tb;  // base of the tables
    if constexpr (MODE == 2)
        tb = wbase;
    else
        tb = lds_raw;
    double* mk;  // [nM_max] E A / L
    if constexpr (MODE == 1)
        mk = reinterpret_cast<double*>(wbase);
    else
        mk = reinterpret_cast<double*>(tb + lay.geom);
    double* mc = mk + nM_max;                                   // [nM_max][3] direction cosines
    double* diag = reinterpret_cast<double*>(tb + lay.diag);    // diagonal 3x3 block per joint
    double* rhs = reinterpret_cast<double*>(tb + lay.rhs);      // reduced load vector (in LDS: a global
                                                                // load in the row loop would have to wait
                                                                // for the stores queued before it)
    double* T = reinterpret_cast<double*>(lds_raw + (MODE == 2 ? 0 : lay.tile));  // row tile, always LDS
    int* fi = reinterpret_cast<int*>(tb + lay.ints);            // [3 nJ_max] free index per DOF
    int* cnt = fi + 3 * nJ_max;                                      // [nJ_max]   joint degree
    int* start = cnt + nJ_max;                                       // [nJ_max+1] exclusive scan
    int* fill = start + nJ_max + 1;                                  // [nJ_max]   fill cursor
    unsigned* adj = reinterpret_cast<unsigned*>(fill + nJ_max);      // [2 nM_max] (other << 16) | member
    int* chunkmin = reinterpret_cast<int*>(adj + 2 * nM_max);                                // [n_pad_max/16] first tile per chunk
    int* cendl = chunkmin + n_pad_max / 16;                          // [n_pad_max/16] envelope: stored extent
    int* rowdof = cendl + n_pad_max / 16;                            // [n_pad_max] DOF of a reduced row
    int* wgflag = rowdof + n_pad_max;                                // [4] [0]: this matrix leaves as entry lists
    unsigned* kmask = reinterpret_cast<unsigned*>(wgflag + 4);       // [n_pad_max/16] tiles of a chunk's rows that hold entries of K

    // ---- phase 0 ---------------------------------------------------------------------------------------
    // The end joints of a thread's first MR members stay in registers: three passes over the members need
    // them (geometry, adjacency fill, rank sort) and each re-read would be another trip to L2 in a phase
    // that is bound by exactly such latencies; the loads are issued here, ahead of the first barrier.
    // (Hoisting the coordinate and E, A loads up here as well was tried and gained nothing: they only queue
    // in front of the loads of the first pass.)
    constexpr int MR = 3;
    int cj0[MR], cj1[MR];
#pragma unroll
    for (int r = 0; r < MR; ++r) {
        const int m = tid + r * NT;
        const size_t mm = (size_t)b * nM_max + (m < nM ? m : 0);
        const int2 c = mem.ends(mm);
        cj0[r] = m < nM ? c.x : 0;
        cj1[r] = m < nM ? c.y : 0;
    }
    auto for_members = [&](auto&& body) {  // body(m, j0, j1) for this thread's members
#pragma unroll
        for (int r = 0; r < MR; ++r) {
            const int m = tid + r * NT;
            if (m < nM) body(m, cj0[r], cj1[r]);
        }
        for (int m = tid + MR * NT; m < nM; m += NT) {
            const size_t mm = (size_t)b * nM_max + m;
            const int2 c = mem.ends(mm);
            body(m, c.x, c.y);
        }
    };
    const double* X = xyz + (size_t)b * 3 * nJ_max;
    const double* F = loads + (size_t)b * 3 * nJ_max;
    const int* fi_global = free_index + (size_t)b * 3 * nJ_max;
    for (int d = tid; d < 3 * nJ; d += NT) {
        const int c = fi_global[d];
        fi[d] = c;
        if (c >= 0) {
            rowdof[c] = d;
            rhs[c] = F[d];  // truss.py:303-304, vecF[mask]
        }
    }
    for (int j = tid; j < nJ; j += NT) cnt[j] = 0;
    for (int q = tid; q < nch; q += NT) {
        chunkmin[q] = q;  // padding rows: diagonal only
        kmask[q] = 1u;    // the diagonal tile always holds entries (a row's own diagonal, the identity padding)
    }
    // The adjacency lists are sorted by RANK when the unsorted lists fit the row tile (still unused here):
    // every member counts the keys of its end joints' lists that are smaller than its own - independent LDS
    // reads instead of the dependent chain of a per-joint insertion sort, which cost 0.046 of the stage's
    // 0.43 ms on bar-942 x 4096.  Same order, so the same bits in K.
    const bool rank_sort = MODE != 2 && (size_t)2 * nM * sizeof(unsigned) <= (size_t)TR * (WT + 16) * sizeof(double);
    unsigned* unsorted = rank_sort ? reinterpret_cast<unsigned*>(T) : adj;
    if (!rank_sort)
        for (int x = tid * 2; x < TR * (WT + 16); x += 2 * NT) *reinterpret_cast<d2*>(T + x) = d2{0.0, 0.0};
    __syncthreads();
    TRS_ASM_PHASE_END(0);   // inputs staged, tables zeroed
    for_members([&](int m, int j0, int j1) {
        const size_t mm = (size_t)b * nM_max + m;
        double d[3], len2 = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            d[a] = X[3 * j1 + a] - X[3 * j0 + a];
            len2 += d[a] * d[a];
        }
        const double len = sqrt(len2);
        mk[m] = mem.EA(mm) / len;                                // truss.py:56-58
#pragma unroll
        for (int a = 0; a < 3; ++a) mc[3 * m + a] = d[a] / len;  // truss.py:60-63
        atomicAdd(&cnt[j0], 1);
        atomicAdd(&cnt[j1], 1);
    });
    __syncthreads();
    TRS_ASM_PHASE_END(1);   // member geometry, degree count
    if (tid < 64) {  // exclusive scan of cnt by one wave, 64 joints per step
        int base = 0;
        for (int j0 = 0; j0 < nJ; j0 += 64) {
            const int j = j0 + tid;
            int v = j < nJ ? cnt[j] : 0;
            int incl = v;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int up = __shfl_up(incl, off);
                if (tid >= off) incl += up;
            }
            if (j < nJ) start[j] = base + incl - v;
            base += __shfl(incl, 63);
        }
        if (tid == 0) start[nJ] = base;
    }
    for (int j = tid; j < nJ; j += NT) fill[j] = 0;
    __syncthreads();
    TRS_ASM_PHASE_END(2);   // scan
    // (the caller may know that skipping does not pay for this batch - TRS_ASM_ALL_TILES -: no mask is formed then)
    const bool want_mask = env_all != nullptr && (flags & (TRS_ASM_ALL_TILES | TRS_ASM_FULL_SYMMETRIC)) == 0;
    // kmask (trs_common.h): which tiles of a chunk's rows hold an entry of K - edge-parallel.  A joint's free rows
    // are consecutive and lie in at most two chunks, its columns in at most two tiles: a member couples the rows of
    // either end with the columns of the other, (2 x 2 tiles) x 2 directions, upper part only; a joint's own block
    // adds the tile right of the diagonal tile when its rows straddle two chunks.
    auto span_of = [&](int j, int& lo, int& hi) {
        const int f0 = fi[3 * j], f1 = fi[3 * j + 1], f2 = fi[3 * j + 2];
        hi = max(f0, max(f1, f2));
        lo = min(f0 >= 0 ? f0 : 0x7fffffff, min(f1 >= 0 ? f1 : 0x7fffffff, f2 >= 0 ? f2 : 0x7fffffff));
    };
    auto couple = [&](int rlo, int rhi, int qlo, int qhi) {  // rows rlo .. rhi x columns qlo .. qhi, upper part
        const int r0c = rlo >> 4, r1c = rhi >> 4, q0t = qlo >> 4, q1t = qhi >> 4;
        unsigned m0 = 0u, m1 = 0u;
        if (q0t >= r0c && q0t - r0c < 32) m0 |= 1u << (q0t - r0c);
        if (q1t >= r0c && q1t - r0c < 32) m0 |= 1u << (q1t - r0c);
        if (q0t >= r1c && q0t - r1c < 32) m1 |= 1u << (q0t - r1c);
        if (q1t >= r1c && q1t - r1c < 32) m1 |= 1u << (q1t - r1c);
        // (most members find their bits set already: a plain read first, the atomic only for new bits)
        if ((kmask[r0c] & m0) != m0) atomicOr(&kmask[r0c], m0);
        if (r1c != r0c && (kmask[r1c] & m1) != m1) atomicOr(&kmask[r1c], m1);
    };
    for_members([&](int m, int j0, int j1) {
        unsorted[start[j0] + atomicAdd(&fill[j0], 1)] = ((unsigned)j1 << 16) | (unsigned)m;
        unsorted[start[j1] + atomicAdd(&fill[j1], 1)] = ((unsigned)j0 << 16) | (unsigned)m;
        if (want_mask) {
            int alo, ahi, blo, bhi;
            span_of(j0, alo, ahi);
            span_of(j1, blo, bhi);
            if (ahi >= 0 && bhi >= 0) {
                couple(alo, ahi, blo, bhi);
                couple(blo, bhi, alo, ahi);
            }
        }
    });
    if (want_mask)
        for (int j = tid; j < nJ; j += NT) {  // a joint's own block
            int lo, hi;
            span_of(j, lo, hi);
            if (hi >= 0 && (hi >> 4) != (lo >> 4)) atomicOr(&kmask[lo >> 4], 2u);
        }
    __syncthreads();
    TRS_ASM_PHASE_END(3);   // adjacency fill, tile mask
    if (rank_sort) {
        for_members([&](int m, int j0, int j1) {
            const unsigned k0 = ((unsigned)j1 << 16) | (unsigned)m, k1 = ((unsigned)j0 << 16) | (unsigned)m;
            const int s0 = start[j0], s1 = start[j1], d0 = cnt[j0], d1 = cnt[j1];
            int r0 = 0, r1 = 0;
            for (int i = 0; i < d0; ++i) r0 += unsorted[s0 + i] < k0 ? 1 : 0;
            adj[s0 + r0] = k0;
            if (j0 == j1) {  // a member from a joint to itself: two equal keys, two slots
                adj[s0 + r0 + 1] = k0;
            } else {
                for (int i = 0; i < d1; ++i) r1 += unsorted[s1 + i] < k1 ? 1 : 0;
                adj[s1 + r1] = k1;
            }
        });
        __syncthreads();
        for (int x = tid * 2; x < TR * (WT + 16); x += 2 * NT) *reinterpret_cast<d2*>(T + x) = d2{0.0, 0.0};
    }
    // one thread per joint: sort its list by (other joint, member); diagonal block; envelope
    for (int a = tid; a < nJ; a += NT) {
        unsigned* list = adj + start[a];
        const int deg = cnt[a];
        for (int i = 1; i < (rank_sort ? 0 : deg); ++i) {  // insertion sort where the rank sort did not apply
            const unsigned key = list[i];
            int p = i - 1;
            while (p >= 0 && list[p] > key) {
                list[p + 1] = list[p];
                --p;
            }
            list[p + 1] = key;
        }
        double dg[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        int mincol = 0x7fffffff;  // smallest reduced column coupled to this joint's rows
#pragma unroll
        for (int s = 0; s < 3; ++s)
            if (fi[3 * a + s] >= 0) mincol = min(mincol, fi[3 * a + s]);
        for (int i = 0; i < deg; ++i) {
            const int other = (int)(list[i] >> 16), m = (int)(list[i] & 0xffffu);
#pragma unroll
            for (int s = 0; s < 3; ++s)
                if (fi[3 * other + s] >= 0) mincol = min(mincol, fi[3 * other + s]);
            const double k = mk[m], cx = mc[3 * m], cy = mc[3 * m + 1], cz = mc[3 * m + 2];
            dg[0] += k * (cx * cx);  // truss.py:65-77, summed over the joint's members in list order
            dg[1] += k * (cx * cy);
            dg[2] += k * (cx * cz);
            dg[3] += k * (cy * cy);
            dg[4] += k * (cy * cz);
            dg[5] += k * (cz * cz);
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) diag[6 * a + q] = dg[q];
#pragma unroll
        for (int s = 0; s < 3; ++s)
            if (fi[3 * a + s] >= 0) atomicMin(&chunkmin[fi[3 * a + s] / 16], mincol / 16);
    }
    __syncthreads();
    TRS_ASM_PHASE_END(4);   // rank sort, per-joint diagonal blocks
    const bool full = (flags & TRS_ASM_FULL_SYMMETRIC) != 0;
    const bool has_env = env_all != nullptr;
    if (has_env) {
        // envelope metadata (trs_common.h), by wave 0 with shuffles (LDS operations of one wave
        // complete in order, so no barrier is needed between the steps)
        if (tid < 64) {
            int* env = env_all + (size_t)b * trs_env_stride(n_pad_max);
            int* ft = env;
            int* last = env + n_pad_max / 16;
            int* cend = env + trs_env_cend_offset(n_pad_max);
            // ft = running minimum of chunkmin from the end, 64 chunks per step
            int carry = nch;
            for (int base = (nch - 1) / 64 * 64; base >= 0; base -= 64) {
                const int q = base + tid;
                int v = q < nch ? chunkmin[q] : nch;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int o = __shfl_down(v, off);
                    if (tid + off < 64) v = min(v, o);
                }
                v = min(v, carry);
                if (q < nch) {
                    chunkmin[q] = v;
                    ft[q] = v;
                }
                carry = __shfl(v, 0);
            }
            __builtin_amdgcn_wave_barrier();
            // lastc[t] = last chunk q with ft[q] <= t: chunk q owns the tiles ft[q] .. ft[q+1]-1
            for (int q = tid; q < nch; q += 64) {
                const int hi = q + 1 < nch ? chunkmin[q + 1] : nch;
                for (int t = chunkmin[q]; t < hi; ++t) cendl[t] = q;
            }
            __builtin_amdgcn_wave_barrier();
            int widest = 0;
            for (int j = tid; j < nch / 4; j += 64) {
                const int l = cendl[4 * j + 3];
                last[j] = l;
                widest = max(widest, l - (4 * j + 3));
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) widest = max(widest, __shfl_xor(widest, off));
            // which factorisation kernel will take this matrix decides the shape of the stored part
            const bool narrow = (widest <= TRS_NARROW_MAX_BELOW || (flags & TRS_ASM_ALL_NARROW) != 0) && (flags & TRS_ASM_ALL_WIDE) == 0;
            // narrow envelopes leave as compact entry lists (phase 1c) when the caller asked for them
            // (TRS_ASM_COMPACT) and the lists can be held: the tables fit next to this batch's other tables,
            // and the envelope is narrow by its own reach - a matrix only FORCED narrow (TRS_ASM_ALL_NARROW)
            // may store more tiles per chunk than the lists are sized for (ntile_cap) and keeps the slab
            const bool compact = narrow && widest <= TRS_NARROW_MAX_BELOW && compact_ok != 0 && !full &&
                                 uf_all != nullptr && (flags & TRS_ASM_COMPACT) != 0;
            if (tid == 0) {
                int* meta = env + n_pad_max / 16 + n_pad_max / 64;
                meta[0] = ((narrow ? TRS_NARROW_ITEM : TRS_WIDE_ITEM) - 1) | (compact ? TRS_ENV_COMPACT : 0) |
                          (narrow && widest > TRS_NARROW_RS4_ABOVE ? TRS_ENV_RS4 : 0);
                if (compact) {  // where the factorisation finds the lists: byte offsets / 16 from `work`
                    const TrsCompactLayout ck = trs_compact_layout(nJ_max, nM_max, n_pad_max);
                    const size_t base = (size_t)b * work_stride + ck_off;
                    meta[1] = (int)((base + ck.tdesc) >> 4);
                    meta[2] = (int)((base + ck.tbase) >> 4);
                    meta[3] = (int)((base + ck.epos) >> 4);
                    meta[4] = (int)((base + ck.eval) >> 4);
                }
                wgflag[0] = compact ? 1 : 0;
                // wave-per-matrix kernels read the load vector from uf: no load column in the slab
                wgflag[1] = (narrow && uf_all != nullptr) ? 1 : 0;
                // ... and skip the tiles of the envelope that hold no entry of K (kmask, trs_common.h); a matrix only
                // FORCED narrow may reach further than the 32 bits of a mask word: it keeps every tile
                wgflag[2] = (narrow && uf_all != nullptr && widest <= TRS_NARROW_MAX_BELOW && !full && (flags & TRS_ASM_ALL_TILES) == 0) ? 1 : 0;
            }
            int empty = 0, stored_tiles = 0;
            for (int t = tid; t < nch; t += 64) {  // t | 3 lies in the same step: reads precede the writes
                const int e = narrow ? max(cendl[t] + 1, (t | 3) + 1) : cendl[t | 3] + 1 + (TRS_WIDE_ITEM - 1);
                __builtin_amdgcn_wave_barrier();
                cendl[t] = min(nch, e);
                cend[t] = min(nch, e);
                const int w = min(nch, e) - t;
                stored_tiles += w;
                empty += w - __popc(kmask[t] & (w >= 32 ? 0xffffffffu : ((1u << w) - 1u)));
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                empty += __shfl_xor(empty, off);
                stored_tiles += __shfl_xor(stored_tiles, off);
            }
            // skipping pays only when enough tiles are skipped (the row loop tests a bit per 16 bytes stored):
            // a tower-like truss (bar-942: 6 of 158 tiles without an entry) keeps every tile
            const bool worth = 8 * empty >= stored_tiles;
            if (tid == 0 && !worth) wgflag[2] = 0;
            {
                int* km = env + trs_env_kmask_offset(n_pad_max);
                __builtin_amdgcn_wave_barrier();
                const bool masked = wgflag[2] != 0;
                for (int t = tid; t < nch; t += 64) km[t] = masked ? (int)kmask[t] : -1;
            }
        }
        __syncthreads();
    }

    TRS_ASM_PHASE_END(5);   // envelope metadata (one wave; the others wait at the barrier)
    // ---- phase 1c: compact per-tile entry lists (narrow envelopes) -----------------------------------------
    if (has_env && wgflag[0] != 0) {
        const TrsCompactLayout ck = trs_compact_layout(nJ_max, nM_max, n_pad_max);
        unsigned char* cbase = work_all + (size_t)b * work_stride + ck_off;
        int* tdesc = reinterpret_cast<int*>(cbase + ck.tdesc);
        int* tbase_g = reinterpret_cast<int*>(cbase + ck.tbase);
        unsigned short* epos = reinterpret_cast<unsigned short*>(cbase + ck.epos);
        double* eval = reinterpret_cast<double*>(cbase + ck.eval);
        int* tbeg = reinterpret_cast<int*>(lds_raw + lay.tile);  // [ntile] first entry of a tile (after the scan)
        int* tfill = tbeg + ck.ntile_cap;                        // [ntile] entries counted / emitted so far
        int* tbase = tfill + ck.ntile_cap;                       // [nch + 1] first tile id of chunk t
        if (tid < 64) {  // tbase = exclusive scan of the stored extent per chunk
            int base = 0;
            for (int t0 = 0; t0 < nch; t0 += 64) {
                const int t = t0 + tid;
                const int v = t < nch ? cendl[t] - t : 0;
                int incl = v;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int up = __shfl_up(incl, off);
                    if (tid >= off) incl += up;
                }
                if (t < nch) {
                    tbase[t] = base + incl - v;
                    tbase_g[t] = base + incl - v;
                }
                base += __shfl(incl, 63);
            }
            if (tid == 0) {
                tbase[nch] = base;
                tbase_g[nch] = base;
            }
        }
        __syncthreads();
        const int ntile = tbase[nch];
        for (int x = tid; x < ntile; x += NT) tfill[x] = 0;
        // load vector: f at the free rows, zero on the padding
        double* ufb = uf_all + (size_t)b * ld_uf;
        for (int c = tid; c < npad; c += NT) ufb[c] = c < n ? rhs[c] : 0.0;
        __syncthreads();
        // Two threads per joint walk its sorted adjacency list (positions h, h + 2, ...); thread h = 0 also
        // takes the joint's own block.  PASS 0 counts the entries per tile, PASS 1 emits them.
        // An entry (row c, column q) is stored when q >= 16 floor(c / 16): the upper part by 16-row tiles,
        // diagonal tiles whole - exactly the slab entries of phase 1.
        auto tile_of = [&](int c, int q) { return tbase[c >> 4] + ((q >> 4) - (c >> 4)); };
        // position inside the chunk's run of tiles: (tile - first tile of the chunk) << 8 | D-form slot
        auto slot_of = [&](int c, int q) {
            return ((((q >> 4) - (c >> 4)) << 8) | (((c & 15) >> 2) * 64 + (c & 3) * 16 + (q & 15)));
        };
        for (int pass = 0; pass < 2; ++pass) {
            for (int w = tid; w < 2 * nJ; w += NT) {
                const int a = w >> 1, h = w & 1;
                const int ca[3] = {fi[3 * a], fi[3 * a + 1], fi[3 * a + 2]};
                if ((ca[0] & ca[1] & ca[2]) < 0) continue;  // all three constrained: no rows
                auto emit = [&](int c, int q, double v) {
                    if (c < 0 || q < 0 || q < (c & ~15)) return;
                    const int tl = tile_of(c, q);
                    if (pass == 0) {
                        atomicAdd(&tfill[tl], 1);
                    } else {
                        const int e = tbeg[tl] + atomicAdd(&tfill[tl], 1);
                        eval[e] = v;
                        epos[e] = (unsigned short)slot_of(c, q);
                    }
                };
                if (h == 0) {  // the joint's own 3x3 block [xx xy xz; xy yy yz; xz yz zz]
                    const double* dg = diag + 6 * a;
                    const double d[3][3] = {{dg[0], dg[1], dg[2]}, {dg[1], dg[3], dg[4]}, {dg[2], dg[4], dg[5]}};
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int sI = 0; sI < 3; ++sI) emit(ca[r], ca[sI], d[r][sI]);
                }
                const unsigned* list = adj + start[a];
                const int deg = cnt[a];
                for (int i = h; i < deg; i += 2) {
                    const int other = (int)(list[i] >> 16);
                    if (i > 0 && (int)(list[i - 1] >> 16) == other) continue;  // not the head of a run
                    const int co[3] = {fi[3 * other], fi[3 * other + 1], fi[3 * other + 2]};
                    double v[3][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
                    if (pass == 1) {
                        int q = i;
                        do {  // parallel members between the same two joints, in member order (as phase 1)
                            const int m = (int)(list[q] & 0xffffu);
                            const double k = mk[m];
                            const double c3[3] = {mc[3 * m], mc[3 * m + 1], mc[3 * m + 2]};
#pragma unroll
                            for (int r = 0; r < 3; ++r)
#pragma unroll
                                for (int sI = 0; sI < 3; ++sI) v[r][sI] -= k * (c3[r] * c3[sI]);
                            ++q;
                        } while (q < deg && (int)(list[q] >> 16) == other);
                    }
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int sI = 0; sI < 3; ++sI) emit(ca[r], co[sI], v[r][sI]);
                }
            }
            for (int c = n + tid; c < npad; c += NT) {  // identity padding
                const int tl = tile_of(c, c);
                if (pass == 0) {
                    atomicAdd(&tfill[tl], 1);
                } else {
                    const int e = tbeg[tl] + atomicAdd(&tfill[tl], 1);
                    eval[e] = 1.0;
                    epos[e] = (unsigned short)slot_of(c, c);
                }
            }
            __syncthreads();
            if (pass == 0) {
                if (tid < 64) {  // exclusive scan of the counts; descriptors to memory
                    int base = 0;
                    for (int x0 = 0; x0 < ntile; x0 += 64) {
                        const int x = x0 + tid;
                        const int v = x < ntile ? tfill[x] : 0;
                        int incl = v;
#pragma unroll
                        for (int off = 1; off < 64; off <<= 1) {
                            const int up = __shfl_up(incl, off);
                            if (tid >= off) incl += up;
                        }
                        if (x < ntile) {
                            tbeg[x] = base + incl - v;
                            tdesc[2 * x] = base + incl - v;
                            tdesc[2 * x + 1] = v;
                            tfill[x] = 0;
                        }
                        base += __shfl(incl, 63);
                    }
                }
                __syncthreads();
            }
        }
        return;
    }

    // ---- phase 1 ---------------------------------------------------------------------------------------
    const bool with_col = !(has_env && wgflag[1] != 0);  // the 16-wide load-column chunk rides in the slab
    const bool masked = has_env && wgflag[2] != 0;       // tiles without an entry of K are not written
    // A matrix whose load vector has gone to uf needs neither the staged vector nor the tile's 16 columns for
    // it any more: the row tile then takes over both (the vector lies right in front of it in LDS) and is wide
    // enough - WTn columns - for a whole row of a narrow envelope, which otherwise goes out in two segments.
    double* Tt = T;
    int WTe = WT, Wstride = WT + 16;
    if (!with_col) {
        double* ufb = uf_all + (size_t)b * ld_uf;
        for (int c = tid; c < npad; c += NT) ufb[c] = c < n ? rhs[c] : 0.0;
        if (MODE != 2 && WTn > WT) {
            __syncthreads();   // every thread has read its part of the vector
            for (int c = tid; c < n_pad_max; c += NT) rhs[c] = 0.0;
            __syncthreads();
            Tt = rhs;
            WTe = WTn;
            Wstride = WTn;
        }
    }
    double* S = S_all + (size_t)b * slab_stride;
    const int rr = tid / TPR, e_first = tid % TPR;
    // Thread TPR-1 of a row carries the joint's own block, threads 0 .. TPR-2 the heads of the runs of
    // its adjacency list (stride TPR-1).  The first piece of every thread is formed one block AHEAD,
    // while the previous block's stores drain, so the scatter itself is three LDS writes.
    int pq0 = -1, pq1 = -1, pq2 = -1, pdeg = 0;
    double pv0 = 0.0, pv1 = 0.0, pv2 = 0.0;
    // block row of the neighbour at list position i (head of a run), row r of the joint
    auto run_values = [&](const unsigned* list, int deg, int i, int r, int& q0, int& q1, int& q2,
                          double& v0, double& v1, double& v2) {
        const int other = (int)(list[i] >> 16);
        q0 = q1 = q2 = -1;
        if (i > 0 && (int)(list[i - 1] >> 16) == other) return;  // not the head of a run
        q0 = fi[3 * other];
        q1 = fi[3 * other + 1];
        q2 = fi[3 * other + 2];
        v0 = v1 = v2 = 0.0;
        int q = i;
        do {  // parallel members between the same two joints, in member order
            const int m = (int)(list[q] & 0xffffu);
            const double k = mk[m], cr = mc[3 * m + r];
            v0 -= k * (cr * mc[3 * m]);
            v1 -= k * (cr * mc[3 * m + 1]);
            v2 -= k * (cr * mc[3 * m + 2]);
            ++q;
        } while (q < deg && (int)(list[q] >> 16) == other);
    };
    auto prepare = [&](int c0) {
        pq0 = pq1 = pq2 = -1;
        pdeg = 0;
        const int c = c0 + rr;
        if (c >= n) return;
        const int dof = rowdof[c];
        const int a = dof / 3, r = dof - 3 * a;
        if (e_first == TPR - 1) {
            const double* dg = diag + 6 * a;  // row r of [xx xy xz; xy yy yz; xz yz zz]
            pv0 = dg[r];
            pv1 = dg[r == 0 ? 1 : (r == 1 ? 3 : 4)];
            pv2 = dg[r == 0 ? 2 : (r == 1 ? 4 : 5)];
            pq0 = fi[3 * a];
            pq1 = fi[3 * a + 1];
            pq2 = fi[3 * a + 2];
        } else {
            pdeg = cnt[a];
            if (e_first < pdeg) run_values(adj + start[a], pdeg, e_first, r, pq0, pq1, pq2, pv0, pv1, pv2);
        }
    };
    prepare(0);
    for (int c0 = 0; c0 < npad; c0 += TR) {
        // stored part of these rows: columns [i_lo, i_hi) (diagonal tile .. end of the envelope of
        // the panel the rows belong to); the 16-wide load-column chunk rides with the last segment
        const int chunk = __builtin_amdgcn_readfirstlane(c0 + rr) >> 4;  // of this wave's rows
        const int i_lo = full ? 0 : 16 * chunk;
        const int i_hi = (has_env && !full) ? 16 * cendl[chunk] : npad;
        const unsigned kmw = masked ? kmask[chunk] : 0xffffffffu;  // (wave-uniform: a wave's rows lie in one chunk)
        for (int seg_lo = i_lo; seg_lo < i_hi; seg_lo += WTe) {
            const int seg_hi = min(i_hi, seg_lo + WTe);
            const int Ws = seg_hi - seg_lo;  // multiple of 16
            const bool is_last = seg_hi == i_hi;
            const int W = Ws + (is_last && with_col ? 16 : 0);
            {   // scatter: every (row, column) of the tile is written by exactly one thread
                double* row = Tt + (size_t)rr * Wstride - seg_lo;
                if (pq0 >= seg_lo && pq0 < seg_hi) row[pq0] = pv0;
                if (pq1 >= seg_lo && pq1 < seg_hi) row[pq1] = pv1;
                if (pq2 >= seg_lo && pq2 < seg_hi) row[pq2] = pv2;
                if (pdeg > e_first + TPR - 1) {  // joints with more than TPR-1 list entries: rare
                    const int dof = rowdof[c0 + rr];
                    const int a = dof / 3, r = dof - 3 * a;
                    for (int i = e_first + TPR - 1; i < pdeg; i += TPR - 1) {
                        int q0, q1, q2;
                        double v0, v1, v2;
                        run_values(adj + start[a], pdeg, i, r, q0, q1, q2, v0, v1, v2);
                        if (q0 >= seg_lo && q0 < seg_hi) row[q0] = v0;
                        if (q1 >= seg_lo && q1 < seg_hi) row[q1] = v1;
                        if (q2 >= seg_lo && q2 < seg_hi) row[q2] = v2;
                    }
                }
                if (e_first == TPR - 1) {
                    const int cc = c0 + rr;
                    if (is_last && with_col) row[seg_lo + Ws] = cc < n ? rhs[cc] : 0.0;  // load column
                    if (cc >= n && cc >= seg_lo && cc < seg_hi) row[cc] = 1.0;   // identity padding
                }
            }
            // A row's tile slice is written and read by the same wave only (TPR divides 64), and a
            // wave's LDS operations complete in order: no work-group barrier in this loop, the waves
            // drift apart and hide each other's latencies.
            __builtin_amdgcn_wave_barrier();
            {   // all TR rows in parallel: TPR threads per row, 16 bytes per thread and pass
                double* dst = S + (size_t)(c0 + rr) * ld;
                double* src = Tt + (size_t)rr * Wstride;
                for (int x = e_first * 2; x < W; x += 2 * TPR) {
                    const int col = x < Ws ? seg_lo + x : npad + (x - Ws);  // envelope part | load column
                    // a tile without an entry of K: nothing was scattered into it, nothing is stored (masked rows
                    // carry no load column, so col is an envelope column here)
                    if (masked && ((kmw >> (((unsigned)col >> 4) - (unsigned)chunk)) & 1u) == 0u) continue;
                    *reinterpret_cast<d2*>(dst + col) = *reinterpret_cast<const d2*>(src + x);
                    *reinterpret_cast<d2*>(src + x) = d2{0.0, 0.0};
                }
            }
            if (is_last && c0 + TR < npad) prepare(c0 + TR);  // overlaps with the stores in flight
            __builtin_amdgcn_wave_barrier();
        }
    }
    TRS_ASM_PHASE_END(6);   // phase 1: the row loop
}

// tile width, table placement (kernel MODE) and work-group size for a batch shape
struct AsmPlan {
    int WT, mode, big;
    int WTn;           // tile width of a matrix whose load vector went to uf (tile + staged vector), 0: as WT
    size_t lds, work;  // LDS per work-group, workspace bytes per truss (tables/geometry + compact lists)
    int compact_ok;    // the tables of the compact path fit (they alias the row tile)
    size_t ck_off;     // offset of the compact-list region inside a truss's workspace
};
inline AsmPlan asm_finish(AsmPlan p, int nJ_max, int nM_max, int n_pad_max, size_t tile_bytes) {
    p.WTn = 0;
    if (p.mode != 2) {
        const int TR = tile_rows(p.big ? NT_BIG : NT_DEFAULT);
        int w = (int)((tile_bytes + (size_t)n_pad_max * 8) / ((size_t)TR * 8)) / 16 * 16;
        if (w > n_pad_max) w = n_pad_max;
        p.WTn = w > p.WT ? w : 0;
    }
    p.ck_off = p.work;
    p.compact_ok = p.mode != 2 && tile_bytes >= asm_compact_tab_bytes(n_pad_max);
    p.work += trs_compact_layout(nJ_max, nM_max < 1 ? 1 : nM_max, n_pad_max).total;
    return p;
}
inline AsmPlan asm_plan(int nJ_max, int nM_max, int n_pad_max) {
    // Tables and member geometry in LDS first - two 512-thread work-groups per CU (80 KiB each), else
    // one 1024-thread work-group with the whole 160 KiB: a global load in the row loop has to wait for
    // every store queued before it (one vmcnt counter), which costs far more than anything else.
    // Geometry in the workspace only when even a whole CU's LDS cannot hold it, and everything but
    // the row tile there when the tables alone do not fit.
    const size_t geom_bytes = ((size_t)(nM_max < 1 ? 1 : nM_max) * 32 + 255) / 256 * 256;
    const int first_g = 1;
    for (int gi = 0; gi < 2; ++gi) {
        const int g = gi == 0 ? first_g : 1 - first_g;
        for (int big = 0; big <= 1; ++big) {
            if (g == 0 && first_g == 0 && gi == 0 && big == 1) continue;   // (the experiment wants the small form only)
            const size_t budget = big ? 160 * 1024 : 80 * 1024;
            const int TR = tile_rows(big ? NT_BIG : NT_DEFAULT);
            const size_t fixed = asm_lds_layout(nJ_max, nM_max, n_pad_max, -16, g, TR).total;
            if (fixed + (size_t)TR * (32 + 16) * 8 > budget) continue;
            int WT = (int)((budget - fixed) / (TR * 8)) - 16;
            WT = WT / 16 * 16;
            if (WT > n_pad_max) WT = n_pad_max;
            return asm_finish(AsmPlan{WT, g ? 0 : 1, big, 0, asm_lds_layout(nJ_max, nM_max, n_pad_max, WT, g, TR).total,
                                      geom_bytes, 0, 0},
                              nJ_max, nM_max, n_pad_max, (size_t)TR * (WT + 16) * 8);
        }
    }
    const int TR = tile_rows(NT_DEFAULT);
    const int WT = n_pad_max < 240 ? n_pad_max : 240;  // 64 KiB of tile: two work-groups per CU
    const size_t tables = asm_lds_layout(nJ_max, nM_max, n_pad_max, -16, 1, TR).total;
    return asm_finish(AsmPlan{WT, 2, 0, 0, (size_t)TR * (WT + 16) * 8, (tables + 255) / 256 * 256, 0, 0}, nJ_max,
                      nM_max, n_pad_max, 0);
}

}  // namespace

extern "C" size_t trs_assemble_work_bytes(int nJ_max, int nM_max, int n_max) {
    return asm_plan(nJ_max, nM_max, trs_round_up(n_max < 1 ? 1 : n_max, TRS_NB)).work;
}

extern "C" int trs_assemble_launch(int B, int nJ_max, int nM_max, const double* xyz, const TrsMembers* members,
                                   const double* loads,
                                   const int* free_index, const int* n_free, const int* nJ,
                                   const int* nM, int ld, size_t slab_stride, int n_pad_max,
                                   double* S, int flags, void* work, int* env, double* uf, int ld_uf,
                                   hipStream_t stream) {
    if (B <= 0 || n_pad_max <= 0) return 0;
    // adjacency keys (unsigned): other joint (16 bits) | member (16 bits)
    if (nJ_max >= 65536 || nM_max >= 65536) return (int)hipErrorInvalidValue;
    const AsmPlan plan = asm_plan(nJ_max, nM_max, n_pad_max);
    // compact-list offsets travel as (byte offset / 16) in 32-bit metadata fields
    const int compact_ok = plan.compact_ok && uf != nullptr && ld_uf >= n_pad_max &&
                           (((size_t)B * plan.work) >> 4) < ((size_t)1 << 31);
    // the dynamic-LDS ceiling of an instantiation is raised once per process, not per launch
#define TRS_LAUNCH_ASSEMBLE(MODE, NTV)                                                                   \
    do {                                                                                                 \
        static const int lds_limit_set = (int)hipFuncSetAttribute(                                       \
            reinterpret_cast<const void*>(trs_assemble_kernel<MODE, NTV>),                               \
            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                     \
        (void)lds_limit_set;                                                                             \
        hipLaunchKernelGGL((trs_assemble_kernel<MODE, NTV>), dim3(B), dim3(NTV), plan.lds, stream, xyz,  \
                           *members, loads, free_index, n_free, nJ, nM, nJ_max, nM_max, n_pad_max, ld,   \
                           slab_stride, S, flags, static_cast<unsigned char*>(work), plan.work, env,     \
                           plan.WT, uf, ld_uf, plan.ck_off, compact_ok, plan.WTn);                                 \
    } while (0)
    if (plan.mode == 2)
        TRS_LAUNCH_ASSEMBLE(2, NT_DEFAULT);
    else if (plan.mode == 0 && !plan.big)
        TRS_LAUNCH_ASSEMBLE(0, NT_DEFAULT);
    else if (plan.mode == 0)
        TRS_LAUNCH_ASSEMBLE(0, NT_BIG);
    else if (!plan.big)
        TRS_LAUNCH_ASSEMBLE(1, NT_DEFAULT);
    else
        TRS_LAUNCH_ASSEMBLE(1, NT_BIG);
#undef TRS_LAUNCH_ASSEMBLE
    return (int)hipGetLastError();
}
