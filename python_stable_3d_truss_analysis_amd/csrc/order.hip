// Joint order on the device: the renumbering of every truss's joints that makes the row envelope of K_ff -
// and with it the 16 x 16 tiles the factorisation touches (trs_common.h) - cheapest, found, applied and priced
// without a host pass.  No reference counterpart (slientruss3d numbers joints in insertion order,
// truss.py:175); the workload is the reference's GenerateRandomCubeTrusses loop (generate.py:342-374), whose
// trusses are far from banded in generator order.
//
// Same candidates, same cost and same tie-breaks as the host version csrc/reorder.c (trs_profile_order), so the
// two return the SAME permutation (asserted in tests/test_gpu_order.py):
//   candidates   reverse Cuthill-McKee (per component: pseudo-peripheral root by two breadth-first sweeps from a
//                minimum-degree joint, neighbours by ascending (degree, id)), its reverse, and the joints sorted
//                lexicographically by coordinates binned to a quarter of the RMS member length, in each of the
//                six axis orders, forwards and backwards;
//   cost         w_q = tiles of row chunk q from its first coupled tile (made non-decreasing from the bottom) to
//                the diagonal, cost = sum_q w_q (w_q + 12).
//
// One 256-thread work-group per truss, everything in LDS:
//   phase A (work-group)  free-joint adjacency (counting sort, lists ordered by (degree, id) with a rank sort),
//                         Cuthill-McKee LEVEL-SYNCHRONOUSLY: a level's joints discover the next level with
//                         integer atomics (max for the visit stamp, min for the parent's position) and the new
//                         level is rank-sorted by (parent position, degree, id) - which IS the queue order of
//                         the serial algorithm; three sweeps per component;
//   phase B (per WAVE)    the four waves take different candidates at the same time, no work-group barriers: a
//                         sweep is ONE rank sort of 64-bit keys (bin, bin, bin, id), pricing is a wave scan of
//                         the free-DOF counts, a neighbour minimum per joint, integer atomic minima per row
//                         chunk and a suffix minimum with shuffles;
//   phase C (work-group)  the cheapest (cost, evaluation order) wins; permutation, envelope reach (the launch hint
//                         of trs_solver.h) and, if asked, the renumbered inputs go to HBM.
#include "trs_common.h"
#include "../../include/trs_solver.h"
#include <type_traits>

namespace {

#ifndef TRS_ORDER_THREADS
#define TRS_ORDER_THREADS 256   // (512 = eight waves, one candidate task per wave instead of two on half of the waves, was
#endif                          // measured: 1.35-2.1 x SLOWER - the kernel's rate follows the work-groups per CU: R5.7)
constexpr int NT = TRS_ORDER_THREADS;
constexpr int NWAVE = NT / 64;
static_assert(NT == 128 || NT == 256 || NT == 512, "two, four or eight waves");
constexpr int ID_BITS = 13;                 // joint ids in the sort keys: nJ_max < 8192
constexpr int PERMANENT = 0x40000000;       // visit stamp of a joint that has its final Cuthill-McKee number
constexpr unsigned long long NO_COST = ~0ull;
constexpr int RCM_BELOW = 128;              // effort 3: free joints below which the Cuthill-McKee candidates are still
                                            // priced (= TRS_ORDER_RCM_BELOW of csrc/reorder.c)

// LDS carve-up shared by host and device (byte offsets, 16-byte aligned parts)
struct OrdLds {
    size_t nfr, deg, start, fill, lvl, ppos, queue, nextq, order, ids, frank, bins, adj, adjU, keys,
        cand, best, newidx, c01, cmin, ctrl, red, total;
    int nch_max, n4;
};
__host__ __device__ inline size_t ord_up(size_t v) { return (v + 15) / 16 * 16; }
__host__ __device__ inline OrdLds ord_layout(int nJ_max, int nM_max) {
    OrdLds l;
    const size_t n = (size_t)nJ_max, e = 2 * (size_t)(nM_max < 1 ? 1 : nM_max);
    l.nch_max = 3 * nJ_max / 16 + 8;
    size_t p = 0;
    auto take = [&](size_t bytes) { const size_t at = p; p += ord_up(bytes); return at; };
    l.nfr = take(n);                        // u8  [n]    free DOFs of a joint (0: fully constrained)
    l.deg = take(4 * n);                    // int [n]    adjacency entries of a joint (parallel members count)
    l.start = take(4 * (n + 1));            // int [n+1]
    l.fill = take(4 * n);                   // int [n]    fill cursor; later: inverse permutation
    l.lvl = take(4 * n);                    // int [n]    visit stamps of the breadth-first sweeps
    l.ppos = take(4 * n);                   // int [n]    queue position of the earliest parent; later: the whole permutation
    l.queue = take(2 * n);                  // u16 [n]    Cuthill-McKee queue of the running sweep
    l.nextq = take(2 * n);                  // u16 [n]    the level being discovered, unsorted
    l.order = take(2 * n);                  // u16 [n]    Cuthill-McKee order of all components so far
    l.ids = take(2 * n);                    // u16 [n]    free joints by ascending id
    l.frank = take(2 * (n + 1));            // u16 [n+1]  number of free joints with a smaller id
    l.bins = take(2 * 3 * n);               // u16 [n][3] coordinate bins
    l.adj = take(2 * (e + 4));              // u16 [2 nM + 4] neighbour lists, by (degree, id)
    // the unsorted lists are dead once `adj` is sorted, before any key or candidate exists: they share their space
    const size_t shared_at = p;
    l.adjU = take(4 * (e + 4));             // u32 [2 nM + 4] ... as filled: (degree of the neighbour << 13) | neighbour
    const size_t lists_end = p;
    p = shared_at;
    // The LDS per work-group decides how many work-groups a CU holds, and the kernel's rate follows that number
    // almost linearly (2 -> 3 work-groups: 1.36 x): 32-bit sort keys (a 64-bit sort borrows the neighbouring wave's
    // slice), first chunk and "straddles two chunks" of a joint in one 16-bit word, 16-bit queues - 39 instead of
    // 50 KB for the largest cube trusses, four work-groups per CU instead of three.
    l.n4 = (nJ_max + 3) / 4 * 4;
    l.keys = take(4 * (size_t)l.n4 * NWAVE); // u32 [4][n4] sort keys (phase A: one array of u64 [2 n4] for the levels)
    l.cand = take(2 * n * NWAVE);           // u16 [4][n] candidate order of a wave
    l.best = take(2 * n * NWAVE);           // u16 [4][n] cheapest order a wave has seen
    l.newidx = take(2 * n * NWAVE);         // u16 [4][n]
    l.c01 = take(2 * n * NWAVE);            // u16 [4][n] (first row chunk of a joint << 1) | its rows straddle two chunks
    p = p > lists_end ? p : lists_end;
    l.cmin = take(4 * (size_t)l.nch_max * NWAVE);  // int [4][nch]
    l.ctrl = take(4 * 32);
    l.red = take(8 * 64);
    l.total = p;
    return l;
}

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(v, off);
        if (lane >= off) v += up;
    }
    return v;
}

// ---- work-group reductions (256 threads; every thread receives the result) ---------------------------------
template <typename T, typename F>
__device__ __forceinline__ T block_reduce(T v, T* red, F op) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = op(v, __shfl_xor(v, off));
    __syncthreads();  // red may still be read from an earlier reduction
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    T r = red[0];
#pragma unroll
    for (int w = 1; w < NWAVE; ++w) r = op(r, red[w]);
    return r;
}

#ifdef TRS_ORDER_STAMPS  // diagnostic build (tools/build_variants.sh): wave-cycles per phase, summed over all waves
__device__ unsigned long long g_ord_stamps[16];
struct OrdStamp {
    unsigned long long t0;
    __device__ OrdStamp() : t0(__builtin_amdgcn_s_memtime()) {}
    __device__ void mark(int slot) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if ((threadIdx.x & 63) == 0) atomicAdd(&g_ord_stamps[slot], t - t0);
        t0 = t;
    }
};
#else
struct OrdStamp {
    __device__ void mark(int) {}
};
#endif


struct Tables {
    unsigned char* nfr;
    int *deg, *start, *fill, *lvl, *ppos;
    unsigned short *queue, *nextq, *order;
    unsigned short *ids, *frank, *bins, *adj;
    unsigned* adjU;
    unsigned long long* keys;   // phase A: u64 [2 n4] level-sort keys; phase B: u32 [4][n4] (see the waves' slices)
    unsigned short *cand, *best, *newidx, *c01;
    int *cmin, *ctrl;
    unsigned long long* red;
};

// One breadth-first sweep from `root` by the whole work-group, level by level; four lanes share a joint's
// neighbour list.  On return queue[0 .. count) holds the joints of root's component level by level and
// *last_begin = index of the first joint of the deepest level.
//   SORTED   every level in the order the serial Cuthill-McKee (reorder.c bfs_levels over (degree, id)-sorted
//            lists) enqueues it: by (queue position of the earliest parent, degree, id) - a rank sort per level;
//   !SORTED  levels as sets only (the two sweeps that look for a pseudo-peripheral root): one barrier per level.
template <bool SORTED>
__device__ int bfs_sweep(const Tables& t, int nj, int root, int stamp, int* last_begin) {
    const int tid = threadIdx.x;
    // Joints discovered per level: THREE counters taken in turn (level d adds to found_cnt[d % 3]).  The unsorted
    // sweeps have one barrier per level, so a thread may read its level's count while faster waves are already
    // adding to the next level's: with a single running counter such a thread saw a tail that included joints of
    // the level after - its loop bounds then differed from its work-group's, it walked queue entries nobody had
    // written yet and the barriers fell out of step (a joint order that is no permutation, stores through garbage
    // indices: what concurrent kernels on other streams, which pull the waves of a work-group apart, brought out -
    // EXPERIMENTS R5.1).  Counter (d + 1) % 3 is cleared during level d: its last readers read it behind the
    // barrier of level d - 2 and have all passed the barrier of level d - 1 since.
    // (The ONE running counter of rounds 3-4 and the probe that counts its hazardous reads are kept as a patch,
    // tools/patches/: the reproducer tools/repro_streams.cpp shows the fault and its cure with them.)
    int* found_cnt = t.ctrl;  // [0 .. 2]
    if constexpr (SORTED)
        for (int j = tid; j < nj; j += NT) t.ppos[j] = 0x7fffffff;
    if (tid == 0) {
        t.queue[0] = (unsigned short)root;
        t.lvl[root] = stamp;
        found_cnt[0] = 0;
    }
    __syncthreads();
    unsigned short* found = SORTED ? t.nextq : t.queue;  // a sorted level is copied into the queue by rank
    int head = 0, tail = 1, depth = 0, begin = 0, turn = 0;
    const int sub = tid & 3;
    while (head < tail) {
        begin = head;
        const int mark = stamp + depth + 1;
        int* my_cnt = found_cnt + turn;
        const int next_turn = turn == 2 ? 0 : turn + 1, slot0 = tail;
        if (tid == 0) found_cnt[next_turn] = 0;
        for (int i = head + (tid >> 2); i < tail; i += NT / 4) {
            const int v = t.queue[i];
            const int e1 = t.start[v + 1];
            for (int e = t.start[v] + sub; e < e1; e += 4) {
                const int w = t.adj[e];
                const int old = atomicMax(&t.lvl[w], mark);
                if (old < stamp) found[slot0 + atomicAdd(my_cnt, 1)] = (unsigned short)w;      // first to reach w
                if (SORTED && (old < stamp || old == mark)) atomicMin(&t.ppos[w], i);          // w's earliest parent
            }
        }
        __syncthreads();
        const int new_tail = slot0 + *my_cnt;
        if constexpr (SORTED) {
            const int m = new_tail - tail;
            if (m > 1) {
                for (int x = tid; x < m; x += NT) {
                    const int w = t.nextq[tail + x];
                    t.keys[x] = ((unsigned long long)t.ppos[w] << 40) | ((unsigned long long)t.deg[w] << 20) |
                                (unsigned long long)w;
                }
                __syncthreads();
                for (int x = tid; x < m; x += NT) {
                    const unsigned long long mine = t.keys[x];
                    int rank = 0;
                    for (int y = 0; y < m; ++y) rank += t.keys[y] < mine ? 1 : 0;
                    t.queue[tail + rank] = (unsigned short)(mine & 0xfffffu);
                }
            } else if (m == 1 && tid == 0) {
                t.queue[tail] = t.nextq[tail];
            }
            __syncthreads();
        }
        head = tail;
        tail = new_tail;
        turn = next_turn;
        ++depth;
    }
    *last_begin = begin;
    return tail;
}

// Cost of an order of the nf free joints (reorder.c envelope_cost), by ONE wave on its own scratch.
// ord(k) = old id of the joint at position k.  Returns sum_q w_q (w_q + 12); *n_out = free DOFs.
template <typename Ord>
__device__ unsigned long long price_order(const Tables& t, int nf, Ord ord, unsigned short* newidx,
                                          unsigned short* c01, int* cmin, int lane, int* n_out) {
    int carry = 0;
    for (int base = 0; base < nf; base += 64) {
        const int k = base + lane;
        const int old = k < nf ? ord(k) : 0;
        const int v = k < nf ? (int)t.nfr[old] : 0;
        const int incl = wave_incl_scan(v, lane);
        const int ds = carry + incl - v;
        if (k < nf) {
            newidx[old] = (unsigned short)k;
            c01[k] = (unsigned short)(((ds >> 4) << 1) | ((((ds + v - 1) >> 4) != (ds >> 4)) ? 1 : 0));
        }
        carry += __shfl(incl, 63);
    }
    const int n = carry, nch = (n + 15) >> 4;
    for (int q = lane; q < nch; q += 64) cmin[q] = q;
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < nf; k += 64) {
        const int old = ord(k);
        int m = k;  // a joint's own rows may straddle two chunks
        const int e1 = t.start[old + 1];
        for (int e = t.start[old]; e < e1; e += 4) {  // four independent look-ups per step (slack behind the lists;
            const int w0 = t.adj[e], w1 = t.adj[e + 1], w2 = t.adj[e + 2], w3 = t.adj[e + 3];   // ids stay < nJ_max)
            const int i0 = newidx[w0], i1 = e + 1 < e1 ? (int)newidx[w1] : k, i2 = e + 2 < e1 ? (int)newidx[w2] : k,
                      i3 = e + 3 < e1 ? (int)newidx[w3] : k;
            m = min(min(m, i0), min(min(i1, i2), i3));
        }
        const int col = c01[m] >> 1, mine = c01[k];
        atomicMin(&cmin[mine >> 1], col);
        if (mine & 1) atomicMin(&cmin[(mine >> 1) + 1], col);
    }
    __builtin_amdgcn_wave_barrier();
    unsigned long long cost = 0;
    int run = nch;
    for (int base = (nch - 1) / 64 * 64; base >= 0; base -= 64) {
        const int q = base + lane;
        int v = q < nch ? cmin[q] : 0x7fffffff;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_down(v, off);
            if (lane + off < 64) v = min(v, o);
        }
        v = min(v, run);
        if (q < nch) {
            const unsigned long long w = (unsigned long long)(q - v + 1);
            cost += w * (w + 12ull);
        }
        run = __shfl(v, 0);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cost += __shfl_xor(cost, off);
    __builtin_amdgcn_wave_barrier();
    *n_out = n;
    return cost;
}

// price_order for an order AND its reverse in one walk over the neighbour lists (the walk - dependent LDS look-ups -
// is most of a pricing): in the reversed order a joint's first coupled position is its neighbours' LAST position of
// the forward order, and its rows start at n - (rows before it) - (its own rows).  fwd(k) = old id of the joint at
// position k; c01r / cminr: a second scratch pair (the wave's slice of the sort keys, dead once the sort is done).
// Same two values as price_order(fwd) and price_order(reversed).
template <typename Fwd>
__device__ void price_order_pair(const Tables& t, int nf, Fwd fwd, unsigned short* newidx, unsigned short* c01,
                                 int* cmin, unsigned short* c01r, int* cminr, int lane, int* n_out,
                                 unsigned long long* cost_fwd, unsigned long long* cost_rev) {
    auto pack = [](int ds, int v) { return (unsigned short)(((ds >> 4) << 1) | ((((ds + v - 1) >> 4) != (ds >> 4)) ? 1 : 0)); };
    int carry = 0;
    for (int base = 0; base < nf; base += 64) {
        const int k = base + lane;
        const int old = k < nf ? fwd(k) : 0;
        const int v = k < nf ? (int)t.nfr[old] : 0;
        const int incl = wave_incl_scan(v, lane);
        const int ds = carry + incl - v;
        if (k < nf) {
            newidx[old] = (unsigned short)k;
            c01[k] = pack(ds, v);
            c01r[k] = (unsigned short)(ds + v);   // rows up to and including this joint; turned into the reversed start below
        }
        carry += __shfl(incl, 63);
    }
    const int n = carry, nch = (n + 15) >> 4;
    for (int q = lane; q < nch; q += 64) {
        cmin[q] = q;
        cminr[q] = q;
    }
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < nf; k += 64) {
        const int end = c01r[k];
        c01r[k] = pack(n - end, (int)t.nfr[fwd(k)]);
    }
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < nf; k += 64) {
        const int old = fwd(k);
        int lo = k, hi = k;  // a joint's own rows may straddle two chunks
        const int e1 = t.start[old + 1];
        for (int e = t.start[old]; e < e1; e += 4) {  // four independent look-ups per step (slack behind the lists;
            const int w0 = t.adj[e], w1 = t.adj[e + 1], w2 = t.adj[e + 2], w3 = t.adj[e + 3];   // ids stay < nJ_max)
            const int i0 = newidx[w0], i1 = e + 1 < e1 ? (int)newidx[w1] : k, i2 = e + 2 < e1 ? (int)newidx[w2] : k,
                      i3 = e + 3 < e1 ? (int)newidx[w3] : k;
            lo = min(min(lo, i0), min(min(i1, i2), i3));
            hi = max(max(hi, i0), max(max(i1, i2), i3));
        }
        const int col = c01[lo] >> 1, mine = c01[k];
        atomicMin(&cmin[mine >> 1], col);
        if (mine & 1) atomicMin(&cmin[(mine >> 1) + 1], col);
        const int colr = c01r[hi] >> 1, miner = c01r[k];
        atomicMin(&cminr[miner >> 1], colr);
        if (miner & 1) atomicMin(&cminr[(miner >> 1) + 1], colr);
    }
    __builtin_amdgcn_wave_barrier();
    unsigned long long cf = 0, cr = 0;
    int runf = nch, runr = nch;
    for (int base = (nch - 1) / 64 * 64; base >= 0; base -= 64) {
        const int q = base + lane;
        int vf = q < nch ? cmin[q] : 0x7fffffff, vr = q < nch ? cminr[q] : 0x7fffffff;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int of = __shfl_down(vf, off), orv = __shfl_down(vr, off);
            if (lane + off < 64) {
                vf = min(vf, of);
                vr = min(vr, orv);
            }
        }
        vf = min(vf, runf);
        vr = min(vr, runr);
        if (q < nch) {
            const unsigned long long wf = (unsigned long long)(q - vf + 1), wr = (unsigned long long)(q - vr + 1);
            cf += wf * (wf + 12ull);
            cr += wr * (wr + 12ull);
        }
        runf = __shfl(vf, 0);
        runr = __shfl(vr, 0);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        cf += __shfl_xor(cf, off);
        cr += __shfl_xor(cr, off);
    }
    __builtin_amdgcn_wave_barrier();
    *n_out = n;
    *cost_fwd = cf;
    *cost_rev = cr;
}

__global__ __launch_bounds__(NT, 4) void trs_joint_order_kernel(
    const double* __restrict__ xyz, const void* __restrict__ conn, const unsigned char* __restrict__ cbits,
    const double* __restrict__ loads, const int* __restrict__ nJ_arr, const int* __restrict__ nM_arr,
    const int nJ_max, const int nM_max, int* __restrict__ perm_out, int* __restrict__ choice_out,
    int* __restrict__ reach_out, double* __restrict__ xyz_out, void* __restrict__ conn_out,
    unsigned char* __restrict__ cbits_out, double* __restrict__ loads_out, const int effort,
    // gather form (trs_joint_order_rows; all null / 0 otherwise): truss b of this launch is row rows[b] of the INPUT
    // arrays, whose rows are nJ_in / nM_in wide; its member sections and counts are copied along
    const long long* __restrict__ rows, const int nJ_in, const int nM_in, const double* __restrict__ E_in,
    const double* __restrict__ A_in, double* __restrict__ E_out, double* __restrict__ A_out,
    int* __restrict__ nJ_out, int* __restrict__ nM_out,
    // table member form (ABI 10; trs_common.h TrsMembers): conn / conn_out are uint16 pairs, and the gather form
    // copies one type index per member (tidx_in -> tidx_out) instead of E and A
    const int conn16, const unsigned char* __restrict__ tidx_in, unsigned char* __restrict__ tidx_out) {
    extern __shared__ unsigned char lds[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t src = rows != nullptr ? (size_t)rows[b] : (size_t)b;   // row of the input arrays
    const int nJ_src = rows != nullptr ? nJ_in : nJ_max, nM_src = rows != nullptr ? nM_in : nM_max;
    const int nj = nJ_arr[src], nm = nM_arr[src];
    const OrdLds lay = ord_layout(nJ_max, nM_max);
    Tables t;
    t.nfr = lds + lay.nfr;
    t.deg = reinterpret_cast<int*>(lds + lay.deg);
    t.start = reinterpret_cast<int*>(lds + lay.start);
    t.fill = reinterpret_cast<int*>(lds + lay.fill);
    t.lvl = reinterpret_cast<int*>(lds + lay.lvl);
    t.ppos = reinterpret_cast<int*>(lds + lay.ppos);
    t.queue = reinterpret_cast<unsigned short*>(lds + lay.queue);
    t.nextq = reinterpret_cast<unsigned short*>(lds + lay.nextq);
    t.order = reinterpret_cast<unsigned short*>(lds + lay.order);
    t.ids = reinterpret_cast<unsigned short*>(lds + lay.ids);
    t.frank = reinterpret_cast<unsigned short*>(lds + lay.frank);
    t.bins = reinterpret_cast<unsigned short*>(lds + lay.bins);
    t.adj = reinterpret_cast<unsigned short*>(lds + lay.adj);
    t.adjU = reinterpret_cast<unsigned*>(lds + lay.adjU);
    t.keys = reinterpret_cast<unsigned long long*>(lds + lay.keys);
    t.cand = reinterpret_cast<unsigned short*>(lds + lay.cand);
    t.best = reinterpret_cast<unsigned short*>(lds + lay.best);
    t.newidx = reinterpret_cast<unsigned short*>(lds + lay.newidx);
    t.c01 = reinterpret_cast<unsigned short*>(lds + lay.c01);
    t.cmin = reinterpret_cast<int*>(lds + lay.cmin);
    t.ctrl = reinterpret_cast<int*>(lds + lay.ctrl);
    t.red = reinterpret_cast<unsigned long long*>(lds + lay.red);

    const unsigned char* CB = cbits + src * nJ_src;
    const double* X = xyz + src * 3 * nJ_src;
    int* P = perm_out + (size_t)b * nJ_max;

    OrdStamp st;
    // ---- phase A.0: free DOFs per joint, degrees, member lengths, coordinate bins ---------------------------
    // Every global READ of the search happens here, coalesced and issued together: the coordinates go to an LDS
    // stage (in the space the adjacency fill will take over later), the members' end joints are kept in
    // registers for the two passes that need them.
    double* Xs = reinterpret_cast<double*>(lds + lay.keys);  // [3 nJ_max] staged coordinates, in the shared region (dead before the fill)
    constexpr int MR = 2048 / NT;   // a thread's first members stay in registers (2048 members without a second read)
    int2 cr[MR];
    const size_t mrow = src * (size_t)nM_src;   // first member of the truss in the input arrays
    auto CNI = [&](int m) {
        if (conn16) {
            const ushort2 c = reinterpret_cast<const ushort2*>(conn)[mrow + m];
            return int2{(int)c.x, (int)c.y};
        }
        return reinterpret_cast<const int2*>(conn)[mrow + m];
    };
#pragma unroll
    for (int r = 0; r < MR; ++r) {
        const int m = tid + r * NT;
        cr[r] = m < nm ? CNI(m) : int2{0, 0};
    }
    auto for_members = [&](auto&& body) {  // body(a, c) for every member of this thread
#pragma unroll
        for (int r = 0; r < MR; ++r)
            if (tid + r * NT < nm) body(cr[r].x, cr[r].y);
        for (int m = tid + MR * NT; m < nm; m += NT) {
            const int2 c = CNI(m);
            body(c.x, c.y);
        }
    };
    double lo[3] = {X[0], X[1], X[2]}, hi[3] = {X[0], X[1], X[2]};
    for (int j = tid; j < nj; j += NT) {
        const int cb = CB[j];
        t.nfr[j] = (unsigned char)(3 - ((cb & 1) + ((cb >> 1) & 1) + ((cb >> 2) & 1)));
        t.deg[j] = 0;
        t.lvl[j] = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double x = X[3 * j + a];
            Xs[3 * j + a] = x;
            lo[a] = x < lo[a] ? x : lo[a];
            hi[a] = x > hi[a] ? x : hi[a];
        }
    }
    __syncthreads();
    double len2 = 0.0;
    int nlen = 0;
    for_members([&](int a, int c) {
        const double dx = Xs[3 * c] - Xs[3 * a], dy = Xs[3 * c + 1] - Xs[3 * a + 1], dz = Xs[3 * c + 2] - Xs[3 * a + 2];
        const double l2 = dx * dx + dy * dy + dz * dz;
        if (l2 > 0.0) {
            len2 += l2;
            ++nlen;
        }
        if (a == c || t.nfr[a] == 0 || t.nfr[c] == 0) return;  // couples nothing in K_ff
        atomicAdd(&t.deg[a], 1);
        atomicAdd(&t.deg[c], 1);
    });
    __syncthreads();
    // scans by wave 0: adjacency offsets, and the free joints by ascending id
    if (tid < 64) {
        int base = 0, fbase = 0;
        for (int j0 = 0; j0 < nj; j0 += 64) {
            const int j = j0 + tid;
            const int v = j < nj ? t.deg[j] : 0;
            const int f = j < nj && t.nfr[j] != 0 ? 1 : 0;
            const int incl = wave_incl_scan(v, tid), fincl = wave_incl_scan(f, tid);
            if (j < nj) {
                t.start[j] = base + incl - v;
                t.frank[j] = (unsigned short)(fbase + fincl - f);
                if (f) t.ids[fbase + fincl - f] = (unsigned short)j;
            }
            base += __shfl(incl, 63);
            fbase += __shfl(fincl, 63);
        }
        if (tid == 0) {
            t.start[nj] = base;
            t.frank[nj] = (unsigned short)fbase;
        }
    }
    for (int j = tid; j < nj; j += NT) t.fill[j] = 0;
    __syncthreads();
    const int nf = t.frank[nj];  // joints that keep a free DOF: they come first in every candidate
    // coordinate bins of the sweeps (while the staged coordinates are still there)
    // member lengths and the bounding box in ONE work-group reduction (eight doubles per wave through LDS)
    {
        double cnt = (double)nlen;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            len2 += __shfl_xor(len2, off);
            cnt += __shfl_xor(cnt, off);
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double l = __shfl_xor(lo[a], off), u = __shfl_xor(hi[a], off);
                lo[a] = l < lo[a] ? l : lo[a];
                hi[a] = u > hi[a] ? u : hi[a];
            }
        }
        double* red = reinterpret_cast<double*>(t.red);  // [NWAVE][8]
        __syncthreads();
        if (lane == 0) {
            red[8 * wave] = len2;
            red[8 * wave + 1] = cnt;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                red[8 * wave + 2 + a] = lo[a];
                red[8 * wave + 5 + a] = hi[a];
            }
        }
        __syncthreads();
        len2 = red[0];
        cnt = red[1];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            lo[a] = red[2 + a];
            hi[a] = red[5 + a];
        }
        for (int w = 1; w < NWAVE; ++w) {  // fixed order: the same sum on every thread
            len2 += red[8 * w];
            cnt += red[8 * w + 1];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                lo[a] = red[8 * w + 2 + a] < lo[a] ? red[8 * w + 2 + a] : lo[a];
                hi[a] = red[8 * w + 5 + a] > hi[a] ? red[8 * w + 5 + a] : hi[a];
            }
        }
        nlen = (int)cnt;
    }
    // a quarter of the RMS member length, rounded to single precision: the sum above runs in another order than
    // the host's, and the bins must not depend on its last bits
    const double h = nlen ? (double)(float)(0.25 * sqrt(len2 / (double)nlen)) : 0.0;
    const bool sweeps = h > 0.0 && h < 1e300 && nf > 1 && effort >= 1;
    // effort 3: on a larger lattice-like truss a coordinate sweep wins (bar-942; every cube truss from 140 cubes up),
    // and Cuthill-McKee - three breadth-first sweeps with a barrier or four per level, plus the (degree, id) sort of
    // the neighbour lists it needs - is 40 % of this kernel's time: its two candidates are priced for small trusses
    // only, or where no sweep is possible.  Same rule as trs_profile_order (csrc/reorder.c).
    // A truss the rule keeps away from Cuthill-McKee that has no usable sweep either (all members of length zero,
    // coordinates that are not numbers) keeps its free joints in the given order.
    const bool use_rcm = effort < 3 || nf < RCM_BELOW;
    const int bin_cap = 4 * nJ_max + 64;
    int nb[3] = {1, 1, 1};
    if (sweeps) {
        int top[3] = {1, 1, 1};
        double ha[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            ha[a] = h;
            if (!((hi[a] - lo[a]) / ha[a] < (double)(bin_cap - 2))) ha[a] = (hi[a] - lo[a]) / (double)(bin_cap - 2);
        }
        for (int j = tid; j < nj; j += NT)
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                double q = ha[a] > 0.0 ? (Xs[3 * j + a] - lo[a]) / ha[a] + 0.5 : 0.0;
                if (!(q >= 0.0)) q = 0.0;  // NaN coordinates: any bin
                if (q > (double)(bin_cap - 1)) q = (double)(bin_cap - 1);
                t.bins[3 * j + a] = (unsigned short)(int)q;
                top[a] = max(top[a], (int)q + 1);
            }
        // nb[a] = bins in use along axis a: the three maxima in one pass
        int t0 = top[0], t1 = top[1], t2 = top[2];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            t0 = max(t0, __shfl_xor(t0, off));
            t1 = max(t1, __shfl_xor(t1, off));
            t2 = max(t2, __shfl_xor(t2, off));
        }
        int* redi = reinterpret_cast<int*>(t.red);
        __syncthreads();
        if (lane == 0) {
            redi[4 * wave] = t0;
            redi[4 * wave + 1] = t1;
            redi[4 * wave + 2] = t2;
        }
        __syncthreads();
        for (int w = 0; w < NWAVE; ++w) {
            nb[0] = max(nb[0], redi[4 * w]);
            nb[1] = max(nb[1], redi[4 * w + 1]);
            nb[2] = max(nb[2], redi[4 * w + 2]);
        }
    }
    __syncthreads();  // the staged coordinates are dead: the adjacency fill takes their space
    for_members([&](int a, int c) {
        if (a == c || t.nfr[a] == 0 || t.nfr[c] == 0) return;
        const int ea = t.start[a] + atomicAdd(&t.fill[a], 1), ec = t.start[c] + atomicAdd(&t.fill[c], 1);
        if (use_rcm) {
            t.adjU[ea] = ((unsigned)t.deg[c] << ID_BITS) | (unsigned)c;
            t.adjU[ec] = ((unsigned)t.deg[a] << ID_BITS) | (unsigned)a;
        } else {  // pricing needs the neighbours of a joint, in any order
            t.adj[ea] = (unsigned short)c;
            t.adj[ec] = (unsigned short)a;
        }
    });
    __syncthreads();
    // neighbour lists by ascending (degree, id): rank of an entry inside its list (equal keys = parallel
    // members: interchangeable, ordered by position)
    // (four lanes share a joint's list: lane `sub` ranks the entries sub, sub + 4, ... - no table of an entry's owner)
    for (int a = tid >> 2; a < (use_rcm ? nj : 0); a += NT / 4) {
        const int s = t.start[a], d = t.deg[a];
        for (int e = s + (tid & 3); e < s + d; e += 4) {
            const unsigned key = t.adjU[e];
            int rank = 0;
            for (int i = 0; i < d; i += 4) {  // four independent reads per step (the list area has four entries of slack)
                const unsigned k0 = t.adjU[s + i], k1 = t.adjU[s + i + 1], k2 = t.adjU[s + i + 2], k3 = t.adjU[s + i + 3];
                rank += (k0 < key || (k0 == key && s + i < e)) ? 1 : 0;
                rank += (i + 1 < d && (k1 < key || (k1 == key && s + i + 1 < e))) ? 1 : 0;
                rank += (i + 2 < d && (k2 < key || (k2 == key && s + i + 2 < e))) ? 1 : 0;
                rank += (i + 3 < d && (k3 < key || (k3 == key && s + i + 3 < e))) ? 1 : 0;
            }
            t.adj[s + rank] = (unsigned short)(key & ((1u << ID_BITS) - 1));
        }
    }
    __syncthreads();

    st.mark(0);
    // ---- phase A.1: Cuthill-McKee, component by component ---------------------------------------------------
    int n_order = 0, stamp = 1;
    while (use_rcm) {
        unsigned best = 0xffffffffu;  // minimum-degree unvisited free joint, smallest id first
        for (int j = tid; j < nj; j += NT)
            if (t.nfr[j] != 0 && t.lvl[j] == 0) best = min(best, ((unsigned)t.deg[j] << ID_BITS) | (unsigned)j);
        best = block_reduce(best, reinterpret_cast<unsigned*>(t.red), [](unsigned x, unsigned y) { return min(x, y); });
        if (best == 0xffffffffu) break;
        int root = (int)(best & ((1u << ID_BITS) - 1)), begin = 0, count = 0;
        for (int sweep = 0; sweep < 2; ++sweep) {  // pseudo-peripheral root: restart from the minimum-(degree, id)
            stamp += nj + 2;                       // joint of the deepest level, twice
            count = bfs_sweep<false>(t, nj, root, stamp, &begin);
            unsigned pick = 0xffffffffu;
            for (int i = begin + tid; i < count; i += NT) {
                const int q = t.queue[i];
                pick = min(pick, ((unsigned)t.deg[q] << ID_BITS) | (unsigned)q);
            }
            pick = block_reduce(pick, reinterpret_cast<unsigned*>(t.red), [](unsigned x, unsigned y) { return min(x, y); });
            root = (int)(pick & ((1u << ID_BITS) - 1));
        }
        stamp += nj + 2;
        count = bfs_sweep<true>(t, nj, root, stamp, &begin);  // neighbour lists are (degree, id)-sorted: Cuthill-McKee
        __syncthreads();
        for (int i = tid; i < count; i += NT) {
            const int v = t.queue[i];
            t.order[n_order + i] = (unsigned short)v;
            t.lvl[v] = PERMANENT;
        }
        n_order += count;
        __syncthreads();
    }
    // (n_order == nf: every free joint belongs to a component)

    st.mark(1);
    __syncthreads();

    st.mark(2);
    // ---- phase B: the candidates, four at a time (one per wave, no work-group barrier) ----------------------
    // axis orders 0..5: xyz xzy yxz yzx zxy zyx (first axis slowest), as bit fields 2 bits per axis
    auto axis_of = [](int ax, int pos) { return (int)((0x192261624ull >> (6 * ax + 2 * pos)) & 3ull); };
    auto nb_of = [&](int a) { return a == 0 ? nb[0] : (a == 1 ? nb[1] : nb[2]); };
    int first_ax = 0;  // the sweep along the longest extent is evaluated first (evaluation order breaks ties)
    for (int ax = 1; ax < 6; ++ax)
        if (nb_of(axis_of(ax, 0)) > nb_of(axis_of(first_ax, 0)) ||
            (nb_of(axis_of(ax, 0)) == nb_of(axis_of(first_ax, 0)) && nb_of(axis_of(ax, 1)) > nb_of(axis_of(first_ax, 1))))
            first_ax = ax;
#ifndef TRS_ORDER_NSWEEP   // (A/B builds: fewer of the six axis orders - the longest-extent sweep, then its partner with
#define TRS_ORDER_NSWEEP 6  //  the same slowest axis; EXPERIMENTS R6.5)
#endif
    const int n_sweep = sweeps ? (effort >= 2 ? TRS_ORDER_NSWEEP : 1) : 0;
    auto bits_for = [](int count) { int b = 1; while ((1 << b) < count) ++b; return b; };  // values 0 .. count-1
    const int bits0 = bits_for(nb[0]), bits1 = bits_for(nb[1]), bits2 = bits_for(nb[2]);
    auto bits_of = [&](int a) { return a == 0 ? bits0 : (a == 1 ? bits1 : bits2); };  // (no indexed private array)
    const int xbits = bits_for(nf > 1 ? nf : 2);
    const int key_bits = bits0 + bits1 + bits2 + xbits;
    unsigned short* cand = t.cand + (size_t)wave * nJ_max;
    unsigned short* wbest = t.best + (size_t)wave * nJ_max;
    unsigned short* newidx = t.newidx + (size_t)wave * nJ_max;
    unsigned short* c01 = t.c01 + (size_t)wave * nJ_max;
    int* cmin = t.cmin + (size_t)wave * lay.nch_max;
    // this wave's slice of the key area: n4 32-bit keys; a 64-bit sort (key fields that do not fit 32 bits: huge
    // coordinate ranges) takes the slices of waves w and w + 1 for an even w, and only the even waves sweep then
    unsigned* k32 = reinterpret_cast<unsigned*>(t.keys) + (size_t)wave * lay.n4;
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(k32);
    unsigned long long my_cost = NO_COST;
    int my_rank = 0x7fffffff, my_choice = 0, ndof = 0;
    auto consider = [&](unsigned long long cost, int eval_rank, int choice, auto ord) {
        if (cost < my_cost || (cost == my_cost && eval_rank < my_rank)) {
            my_cost = cost;
            my_rank = eval_rank;
            my_choice = choice;
            for (int k = lane; k < nf; k += 64) wbest[k] = (unsigned short)ord(k);
            __builtin_amdgcn_wave_barrier();
        }
    };
    if (nf > 0) {
        if (!use_rcm && !sweeps && wave == 0) {  // nothing to choose from: the free joints by ascending id
            auto ord = [&](int k) { return (int)t.ids[k]; };
            consider(price_order(t, nf, ord, newidx, c01, cmin, lane, &ndof), 0, 14, ord);
        }
        if (use_rcm && wave == (n_sweep > 1 ? NWAVE - 2 : 0)) {  // reverse Cuthill-McKee
            auto ord = [&](int k) { return t.order[nf - 1 - k]; };
            consider(price_order(t, nf, ord, newidx, c01, cmin, lane, &ndof), 0, 0, ord);
        }
        if (use_rcm && wave == (n_sweep > 1 ? NWAVE - 1 : 1)) {  // plain Cuthill-McKee
            auto ord = [&](int k) { return t.order[k]; };
            consider(price_order(t, nf, ord, newidx, c01, cmin, lane, &ndof), 1, 1, ord);
        }
        const bool wide_keys = key_bits > 32;
        for (int i = wide_keys ? ((wave & 1) ? n_sweep : wave / 2) : wave; i < n_sweep; i += wide_keys ? NWAVE / 2 : NWAVE) {
#if TRS_ORDER_NSWEEP < 6
            const int ax = i == 0 ? first_ax : (first_ax ^ 1);
#else
            const int ax = i == 0 ? first_ax : (i <= first_ax ? i - 1 : i);
#endif
            const int a0 = axis_of(ax, 0), a1 = axis_of(ax, 1), a2 = axis_of(ax, 2);
            // lexicographic by (bin a0, bin a1, bin a2, id): ONE rank sort.  x = position in the ascending id
            // list, so (bins, x) orders like (bins, id).  32-bit keys when the three bin fields and x fit (any
            // lattice-like truss: a few dozen bins per axis) - four keys per LDS read, one compare each -,
            // 64-bit keys otherwise.
            if (!wide_keys) {
                for (int x = lane; x < nf; x += 64) {
                    const int j = t.ids[x];
                    k32[x] = ((((unsigned)t.bins[3 * j + a0] << bits_of(a1) | (unsigned)t.bins[3 * j + a1]) << bits_of(a2) |
                               (unsigned)t.bins[3 * j + a2]) << xbits) | (unsigned)x;
                }
                for (int x = nf + lane; x < ((nf + 3) & ~3); x += 64) k32[x] = 0xffffffffu;  // pad the last quad
                __builtin_amdgcn_wave_barrier();
                // a lane ranks up to R of its keys in ONE pass over the list: a quad of keys per LDS read is
                // compared with all of them (the reads, not the compares, are what a rank sort waits for)
                auto rank_pass = [&](auto rc, int x0) {
                    constexpr int R = decltype(rc)::value;
                    unsigned mine[R];
                    int rank[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int x = x0 + lane + 64 * r;
                        mine[r] = x < nf ? k32[x] : 0u;
                        rank[r] = 0;
                    }
#pragma unroll 2
                    for (int y = 0; y < nf; y += 4) {  // (broadcast reads: every lane the same four keys)
                        const uint4 kk = *reinterpret_cast<const uint4*>(k32 + y);
#pragma unroll
                        for (int r = 0; r < R; ++r)
                            rank[r] += (kk.x < mine[r] ? 1 : 0) + (kk.y < mine[r] ? 1 : 0) + (kk.z < mine[r] ? 1 : 0) +
                                       (kk.w < mine[r] ? 1 : 0);
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        if (x0 + lane + 64 * r < nf) cand[rank[r]] = t.ids[mine[r] & ((1u << xbits) - 1)];
                };
                int x0 = 0;
                for (; nf - x0 > 128; x0 += 256) rank_pass(std::integral_constant<int, 4>{}, x0);
                if (nf - x0 > 64) rank_pass(std::integral_constant<int, 2>{}, x0);
                else if (nf - x0 > 0) rank_pass(std::integral_constant<int, 1>{}, x0);
            } else {
                for (int x = lane; x < nf; x += 64) {
                    const int j = t.ids[x];
                    keys[x] = ((unsigned long long)t.bins[3 * j + a0] << (32 + ID_BITS)) |
                              ((unsigned long long)t.bins[3 * j + a1] << (16 + ID_BITS)) |
                              ((unsigned long long)t.bins[3 * j + a2] << ID_BITS) | (unsigned long long)j;
                }
                __builtin_amdgcn_wave_barrier();
                for (int x = lane; x < nf; x += 64) {
                    const unsigned long long mine = keys[x];
                    int rank = 0;
                    int y = 0;
                    for (; y + 1 < nf; y += 2) {  // two keys per LDS read
                        const ulonglong2 kk = *reinterpret_cast<const ulonglong2*>(keys + y);
                        rank += (kk.x < mine ? 1 : 0) + (kk.y < mine ? 1 : 0);
                    }
                    if (y < nf) rank += keys[y] < mine ? 1 : 0;
                    cand[rank] = (unsigned short)(mine & ((1ull << ID_BITS) - 1));
                }
            }
            __builtin_amdgcn_wave_barrier();
            st.mark(3);
            auto fwd = [&](int k) { return (int)cand[k]; };
            auto rev = [&](int k) { return (int)cand[nf - 1 - k]; };
            // (second scratch pair: this wave's key slice, dead now - 2 nf + 4 nch <= 4 nf bytes)
            unsigned short* c01r = reinterpret_cast<unsigned short*>(k32);
            int* cminr = reinterpret_cast<int*>(k32 + ((nf + 3) >> 2 << 1));
            unsigned long long cost_fwd, cost_rev;
            price_order_pair(t, nf, fwd, newidx, c01, cmin, c01r, cminr, lane, &ndof, &cost_fwd, &cost_rev);
            consider(cost_fwd, 2 + 2 * i, 2 + 2 * ax, fwd);
            consider(cost_rev, 3 + 2 * i, 3 + 2 * ax, rev);
            st.mark(4);
        }
    }
    st.mark(5);
    // ---- phase C: the winner ---------------------------------------------------------------------------------
    if (lane == 0) {
        t.red[wave] = my_cost;
        t.ctrl[8 + wave] = my_rank;
        t.ctrl[16 + wave] = my_choice;
    }
    __syncthreads();
    st.mark(6);
    int win = 0;
    for (int w = 1; w < NWAVE; ++w)
        if (t.red[w] < t.red[win] || (t.red[w] == t.red[win] && t.ctrl[8 + w] < t.ctrl[8 + win])) win = w;
    const unsigned short* wperm = t.best + (size_t)win * nJ_max;
    int* inverse = t.fill;
    int* fullperm = t.ppos;   // [nJ_max] the whole permutation (the sweeps' parent positions are dead)
    for (int k = tid; k < nJ_max; k += NT) {
        if (k < nf) {
            const int old = wperm[k];
            fullperm[k] = old;
            inverse[old] = k;
        } else if (k >= nj) {
            fullperm[k] = k;  // identity on the padding
        }
    }
    for (int j = tid; j < nj; j += NT)
        if (t.nfr[j] == 0) {  // fully constrained joints follow the free ones in their given order
            const int k = nf + j - (int)t.frank[j];
            fullperm[k] = j;
            inverse[j] = k;
        }
    if (tid == 0 && choice_out != nullptr) choice_out[b] = nf > 0 ? t.ctrl[16 + win] : 0;
    __syncthreads();
    for (int k = tid; k < nJ_max; k += NT) P[k] = fullperm[k];
    // envelope reach of the chosen order below the 64 x 64 diagonal blocks (reorder.c trs_envelope_reach; what
    // trs_assemble will derive): by wave 0 on its scratch
    if (reach_out != nullptr && wave == 0) {
        int widest = 0;
        if (nf > 0) {
            auto ord = [&](int k) { return (int)wperm[k]; };
            int n = 0;
            price_order(t, nf, ord, newidx, c01, cmin, lane, &n);
            const int nch = (n + 15) >> 4, nchp = (n + 63) / 64 * 4;
            int* ft = cmin;  // cmin -> ft: running minimum from the end (padding chunks couple to themselves)
            for (int q = nch + lane; q < nchp; q += 64) ft[q] = q;
            __builtin_amdgcn_wave_barrier();
            int run = nchp;
            for (int base = (nchp - 1) / 64 * 64; base >= 0; base -= 64) {
                const int q = base + lane;
                int v = q < nchp ? ft[q] : 0x7fffffff;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int o = __shfl_down(v, off);
                    if (lane + off < 64) v = min(v, o);
                }
                v = min(v, run);
                if (q < nchp) ft[q] = v;
                run = __shfl(v, 0);
            }
            __builtin_amdgcn_wave_barrier();
            for (int j = lane; j < nchp / 4; j += 64) {  // last chunk q with ft[q] <= 4 j + 3 (ft is non-decreasing)
                const int tcol = 4 * j + 3;
                int q = tcol;
                while (q + 1 < nchp && ft[q + 1] <= tcol) ++q;
                widest = max(widest, q - tcol);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) widest = max(widest, __shfl_xor(widest, off));
        }
        if (lane == 0) reach_out[b] = widest;
    }
    st.mark(7);
    // the renumbered inputs (reorder.c trs_apply_joint_order): joint k := old joint perm[k], members keep their
    // order with renumbered ends, padding members stay (0, 0)
    if (xyz_out != nullptr) {
        const double* F = loads + src * 3 * nJ_src;
        double* XO = xyz_out + (size_t)b * 3 * nJ_max;
        double* FO = loads_out + (size_t)b * 3 * nJ_max;
        unsigned char* CO = cbits_out + (size_t)b * nJ_max;
        for (int j = tid; j < nJ_max; j += NT) {  // a joint per thread, coalesced reads: old joint j -> row inverse[j]
            const int k = j < nj ? inverse[j] : j;
            // (the padding of the output rows is copied from the input's where that exists - the plain form, where
            // input and output rows are equally wide - and zero-filled in the gather form)
            const bool in = rows == nullptr || j < nj;
            const double x0 = in ? X[3 * j] : 0.0, x1 = in ? X[3 * j + 1] : 0.0, x2 = in ? X[3 * j + 2] : 0.0;
            const double f0 = in ? F[3 * j] : 0.0, f1 = in ? F[3 * j + 1] : 0.0, f2 = in ? F[3 * j + 2] : 0.0;
            const unsigned char cb = in ? CB[j] : (unsigned char)0;
            XO[3 * k] = x0; XO[3 * k + 1] = x1; XO[3 * k + 2] = x2;
            FO[3 * k] = f0; FO[3 * k + 1] = f1; FO[3 * k + 2] = f2;
            CO[k] = cb;
        }
        for (int m = tid; m < nM_max; m += NT) {
            int2 c = {0, 0};
            if (m < nm) {
                c = CNI(m);
                c.x = inverse[c.x];
                c.y = inverse[c.y];
            }
            if (conn16) reinterpret_cast<ushort2*>(conn_out)[(size_t)b * nM_max + m] = ushort2{(unsigned short)c.x, (unsigned short)c.y};
            else reinterpret_cast<int2*>(conn_out)[(size_t)b * nM_max + m] = c;
        }
    }
    if (rows != nullptr) {  // gather form: the member sections and the counts travel with the truss
        if (tidx_in != nullptr) {
            const unsigned char* TI = tidx_in + src * nM_src;
            unsigned char* TO = tidx_out + (size_t)b * nM_max;
            for (int m = tid; m < nM_max; m += NT) TO[m] = m < nm ? TI[m] : (unsigned char)0;
        } else {
            const double* EI = E_in + src * nM_src;
            const double* AI = A_in + src * nM_src;
            double* EO = E_out + (size_t)b * nM_max;
            double* AO = A_out + (size_t)b * nM_max;
            for (int m = tid; m < nM_max; m += NT) {
                EO[m] = m < nm ? EI[m] : 0.0;
                AO[m] = m < nm ? AI[m] : 0.0;
            }
        }
        if (tid == 0) {
            nJ_out[b] = nj;
            nM_out[b] = nm;
        }
    }
    st.mark(8);
}

}  // namespace


#ifdef TRS_ORDER_STAMPS
extern "C" int trs_order_debug_stamps(unsigned long long* host_out, int reset) {
    int rc = (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ord_stamps), sizeof(g_ord_stamps));
    if (reset) {
        unsigned long long zero[16] = {0};
        rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ord_stamps), zero, sizeof(zero));
    }
    return rc;
}
#endif

extern "C" int trs_joint_order_fits(int nJ_max, int nM_max) {
    if (nJ_max <= 0 || nM_max < 0 || nJ_max >= (1 << ID_BITS) || nM_max >= 65536) return 0;
    return ord_layout(nJ_max, nM_max).total <= 160 * 1024 ? 1 : 0;
}

extern "C" int trs_joint_order_launch(int B, int nJ_max, int nM_max, const double* xyz, const void* conn,
                                      const unsigned char* cbits, const double* loads, const int* nJ, const int* nM,
                                      int* perm, int* choice, int* reach, double* xyz_out, void* conn_out,
                                      unsigned char* cbits_out, double* loads_out, int effort, hipStream_t stream,
                                      const long long* rows, int nJ_in, int nM_in, const double* E_in,
                                      const double* A_in, double* E_out, double* A_out, int* nJ_out, int* nM_out,
                                      int conn16, const unsigned char* tidx_in, unsigned char* tidx_out) {
    if (B <= 0) return 0;
    if (!trs_joint_order_fits(nJ_max, nM_max)) return (int)hipErrorInvalidValue;
    const size_t lds = ord_layout(nJ_max, nM_max).total
        ;
    // The dynamic-LDS ceiling of the kernel is raised only when a launch needs more than the default 64 KB, once per
    // process (cube trusses: 25-39 KB; only shapes beyond ~500 joints get here).  Raising it for every kernel of the
    // library up front made the host-fed pipeline's CU-masked streams fault on this runtime (EXPERIMENTS R4.7).
    if (lds > 64 * 1024) {
        static const int raised = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(trs_joint_order_kernel),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)raised;
    }
    hipLaunchKernelGGL(trs_joint_order_kernel, dim3(B), dim3(NT), lds, stream, xyz, conn, cbits, loads, nJ, nM, nJ_max,
                       nM_max, perm, choice, reach, xyz_out, conn_out, cbits_out, loads_out, effort, rows, nJ_in, nM_in,
                       E_in, A_in, E_out, A_out, nJ_out, nM_out, conn16, tidx_in, tidx_out);
    return (int)hipGetLastError();
}
