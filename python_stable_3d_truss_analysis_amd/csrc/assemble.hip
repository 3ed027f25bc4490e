// Assembly of the reduced stiffness slab (include/trs_solver.h) for a batch of trusses.
//
// Replaces Member.k / cosines / matK (slientruss3d/truss.py:56-86), Truss.GetKMatrix
// (truss.py:307-316), GetExternalForceVector (truss.py:303-304) and the row/column elimination
// matK[mask,:][:,mask], vecF[mask] (truss.py:343).
//
// Two kernels, no atomics on floating-point data, bit-reproducible:
//
//  1. trs_joint_blocks_kernel - one work-group per truss, edge-parallel.  Every member's
//     k, c = (x1 - x0)/L are computed once (one thread per member); a sorted joint adjacency is
//     built in LDS (integer counting sort + per-joint insertion sort by (other joint, member));
//     one thread per joint then walks its list IN THAT FIXED ORDER and emits the joint's 3x3
//     stiffness blocks - the diagonal block (sum over incident members) and one block per distinct
//     neighbour (parallel members merged) - as "entries" (3 reduced column indices + the 6 unique
//     values of the symmetric block) into a per-truss workspace, plus a per-row directory
//     (rowinfo, rowrhs).  ~0.15 MB per truss, ~1 % of the run time.
//
//  2. trs_expand_kernel - owner-computes, HBM-write bound.  One persistent work-group per truss walks
//     the slab TR rows at a time: scatter the rows' entries into an LDS tile (pure data movement:
//     every (row, column) is written by exactly one thread), add the load column and the identity
//     padding, then write each row to HBM exactly once with 16-byte coalesced stores, zeroing the
//     tile behind the reads.  The next block's entries are prefetched during the stores.
#include "trs_common.h"
#include "../../include/trs_solver.h"

namespace {

// ---- per-truss workspace layout (all offsets in bytes, 16-byte aligned) ---------------------------
struct AsmWork {
    size_t vals, cols, rowinfo, rowrhs, geom, total;
    int nent_max;
};
__host__ __device__ inline AsmWork asm_work_layout(int nJ_max, int nM_max, int n_pad_max) {
    AsmWork w;
    w.nent_max = nJ_max + 2 * nM_max;               // one diagonal entry per joint + one per member end
    w.vals = 0;                                     // double[nent_max][6]
    w.cols = w.vals + (size_t)w.nent_max * 48;      // int[nent_max][4]
    w.rowinfo = w.cols + (size_t)w.nent_max * 16;   // int2[n_pad_max]: (first entry, count | axis << 16)
    w.rowrhs = w.rowinfo + (size_t)n_pad_max * 8;   // double[n_pad_max]
    w.geom = w.rowrhs + (size_t)n_pad_max * 8;      // double[nM_max][4]: k, c (only for trusses whose
                                                    // member geometry does not fit LDS)
    w.total = (w.geom + (size_t)nM_max * 32 + 255) / 256 * 256;
    return w;
}

// ---- kernel 1 ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void trs_joint_blocks_kernel(
    const double* __restrict__ xyz, const int* __restrict__ conn, const double* __restrict__ E,
    const double* __restrict__ A, const double* __restrict__ loads,
    const int* __restrict__ free_index, const int* __restrict__ n_free, const int* __restrict__ nJ_arr,
    const int* __restrict__ nM_arr, const int nJ_max, const int nM_max, const int n_pad_max,
    unsigned char* __restrict__ work_all, int* __restrict__ env_all, const int geom_in_lds) {
    extern __shared__ unsigned char lds_raw[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int nJ = nJ_arr[b], nM = nM_arr[b];
    const AsmWork lay = asm_work_layout(nJ_max, nM_max, n_pad_max);
    unsigned char* work = work_all + (size_t)b * lay.total;
    double* ent_vals = reinterpret_cast<double*>(work + lay.vals);
    int* ent_cols = reinterpret_cast<int*>(work + lay.cols);
    int* rowinfo = reinterpret_cast<int*>(work + lay.rowinfo);
    double* rowrhs = reinterpret_cast<double*>(work + lay.rowrhs);

    // member geometry: in LDS when it fits, else in the truss's workspace (L2); then the integer arrays
    double* mk = geom_in_lds ? reinterpret_cast<double*>(lds_raw)
                             : reinterpret_cast<double*>(work + lay.geom);  // [nM_max]   E A / L
    double* mc = mk + nM_max;                                               // [nM_max][3] direction cosines
    int* cnt = reinterpret_cast<int*>(lds_raw + (geom_in_lds ? (size_t)nM_max * 32 : 0));  // [nJ_max]
    int* start = cnt + nJ_max;                                // [nJ_max+1] exclusive scan of cnt
    int* fill = start + nJ_max + 1;                           // [nJ_max]   fill cursor / entry count
    int* adj = fill + nJ_max;                                 // [2 nM_max] (other joint << 16) | member
    int* chunkmin = adj + 2 * nM_max;                         // [n_pad_max/16] first tile per row chunk
    int* fi = chunkmin + n_pad_max / 16;                      // [3 nJ_max] free index of every DOF

    const double* X = xyz + (size_t)b * 3 * nJ_max;
    const int* fi_global = free_index + (size_t)b * 3 * nJ_max;
    for (int d = tid; d < 3 * nJ; d += 256) fi[d] = fi_global[d];
    for (int j = tid; j < nJ; j += 256) cnt[j] = 0;
    for (int q = tid; q < n_pad_max / 16; q += 256) chunkmin[q] = q;  // padding rows: diagonal only
    __syncthreads();
    for (int m = tid; m < nM; m += 256) {
        const size_t mm = (size_t)b * nM_max + m;
        const int j0 = conn[2 * mm], j1 = conn[2 * mm + 1];
        double d[3], len2 = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            d[a] = X[3 * j1 + a] - X[3 * j0 + a];
            len2 += d[a] * d[a];
        }
        const double len = sqrt(len2);
        mk[m] = E[mm] * A[mm] / len;                    // truss.py:56-58
#pragma unroll
        for (int a = 0; a < 3; ++a) mc[3 * m + a] = d[a] / len;  // truss.py:60-63
        atomicAdd(&cnt[j0], 1);
        atomicAdd(&cnt[j1], 1);
    }
    __syncthreads();
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
    for (int j = tid; j < nJ; j += 256) fill[j] = 0;
    __syncthreads();
    for (int m = tid; m < nM; m += 256) {
        const size_t mm = (size_t)b * nM_max + m;
        const int j0 = conn[2 * mm], j1 = conn[2 * mm + 1];
        adj[start[j0] + atomicAdd(&fill[j0], 1)] = (j1 << 16) | m;
        adj[start[j1] + atomicAdd(&fill[j1], 1)] = (j0 << 16) | m;
    }
    __syncthreads();
    // one thread per joint: sort its list by (other joint, member), then emit its blocks
    for (int a = tid; a < nJ; a += 256) {
        int* list = adj + start[a];
        const int deg = cnt[a];
        for (int i = 1; i < deg; ++i) {  // insertion sort, deg is small (<= ~20 for real trusses)
            const int key = list[i];
            int p = i - 1;
            while (p >= 0 && list[p] > key) {
                list[p + 1] = list[p];
                --p;
            }
            list[p + 1] = key;
        }
        const int e0 = a + start[a];  // first entry of this joint: the diagonal block
        double diag[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        int nent = 1, i = 0;
        int mincol = 0x7fffffff;  // smallest reduced column coupled to this joint's rows (envelope)
#pragma unroll
        for (int s = 0; s < 3; ++s)
            if (fi[3 * a + s] >= 0) mincol = min(mincol, fi[3 * a + s]);
        while (i < deg) {
            const int other = list[i] >> 16;
#pragma unroll
            for (int s = 0; s < 3; ++s)
                if (fi[3 * other + s] >= 0) mincol = min(mincol, fi[3 * other + s]);
            double blk[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            do {  // run of parallel members between the same two joints, in member order
                const int m = list[i] & 0xffff;
                const double k = mk[m], cx = mc[3 * m], cy = mc[3 * m + 1], cz = mc[3 * m + 2];
                const double v[6] = {k * (cx * cx), k * (cx * cy), k * (cx * cz),
                                     k * (cy * cy), k * (cy * cz), k * (cz * cz)};  // truss.py:65-77
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    diag[q] += v[q];
                    blk[q] -= v[q];
                }
                ++i;
            } while (i < deg && (list[i] >> 16) == other);
            const size_t e = (size_t)e0 + nent;
#pragma unroll
            for (int q = 0; q < 6; ++q) ent_vals[6 * e + q] = blk[q];
#pragma unroll
            for (int s = 0; s < 3; ++s) ent_cols[4 * e + s] = fi[3 * other + s];
            ++nent;
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) ent_vals[6 * (size_t)e0 + q] = diag[q];
#pragma unroll
        for (int s = 0; s < 3; ++s) ent_cols[4 * (size_t)e0 + s] = fi[3 * a + s];
        fill[a] = nent;
#pragma unroll
        for (int s = 0; s < 3; ++s)
            if (fi[3 * a + s] >= 0) atomicMin(&chunkmin[fi[3 * a + s] / 16], mincol / 16);
    }
    __syncthreads();
    const double* F = loads + (size_t)b * 3 * nJ_max;
    for (int dof = tid; dof < 3 * nJ; dof += 256) {
        const int c = fi[dof];
        if (c >= 0) {
            const int a = dof / 3, r = dof % 3;
            rowinfo[2 * c] = a + start[a];
            rowinfo[2 * c + 1] = fill[a] | (r << 16);
            rowrhs[c] = F[dof];  // truss.py:303-304, vecF[mask]
        }
    }
    // envelope metadata (trs_common.h): monotone first-tile per chunk, last chunk per panel
    if (env_all != nullptr && tid == 0) {
        int* env = env_all + (size_t)b * trs_env_stride(n_pad_max);
        const int nch = trs_round_up(n_free[b], TRS_NB) / 16;
        int* ft = env;
        int* last = env + n_pad_max / 16;
        int running = nch;
        for (int q = nch - 1; q >= 0; --q) {
            running = min(running, chunkmin[q]);
            ft[q] = running;
        }
        int q = 0, widest = 0;
        for (int j = 0; j < nch / 4; ++j) {
            while (q + 1 < nch && ft[q + 1] <= 4 * j + 3) ++q;
            last[j] = q;
            widest = max(widest, q - (4 * j + 3));
        }
        // which factorisation kernel will take this matrix decides the item size, hence the slack
        env[n_pad_max / 16 + n_pad_max / 64] =
            (widest <= TRS_NARROW_MAX_BELOW ? TRS_NARROW_ITEM : TRS_WIDE_ITEM) - 1;
    }
}

// ---- kernel 2 ------------------------------------------------------------------------------------------
// One persistent work-group per truss walks the slab in blocks of TR rows.  Per block: scatter the
// prefetched entries into the (all-zero) LDS tile, issue the loads of the next block's entries,
// barrier, then every thread reads 16-byte pieces of the tile, streams them to HBM and writes zeros
// back behind itself (the tile is clean again without a separate pass), barrier.  The row directory
// of the whole truss is cached in LDS up front, so a block costs one exposed-latency-free round.
template <int TR>
__global__ __launch_bounds__(256) void trs_expand_kernel(const unsigned char* __restrict__ work_all,
                                                         const int* __restrict__ n_free,
                                                         const int nJ_max, const int nM_max,
                                                         const int n_pad_max, const int ld,
                                                         const size_t slab_stride,
                                                         double* __restrict__ S_all, const int flags,
                                                         const int* __restrict__ env_all) {
    extern __shared__ double lds[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = n_free[b];
    const int npad = trs_round_up(n, TRS_NB);
    if (npad == 0) return;
    const int Wmax = n_pad_max + 16;
    double* T = lds;                                                 // [TR][Wmax] row tile
    int2* dir = reinterpret_cast<int2*>(lds + (size_t)TR * Wmax);    // [npad] row directory

    const AsmWork lay = asm_work_layout(nJ_max, nM_max, n_pad_max);
    const unsigned char* work = work_all + (size_t)b * lay.total;
    const double* ent_vals = reinterpret_cast<const double*>(work + lay.vals);
    const int4* ent_cols = reinterpret_cast<const int4*>(work + lay.cols);
    const int2* rowinfo = reinterpret_cast<const int2*>(work + lay.rowinfo);
    const double* rowrhs = reinterpret_cast<const double*>(work + lay.rowrhs);
    double* S = S_all + (size_t)b * slab_stride;

    for (int c = tid; c < npad; c += 256) dir[c] = c < n ? rowinfo[c] : int2{0, 0};
    for (int x = tid * 2; x < TR * Wmax; x += 512) *reinterpret_cast<d2*>(T + x) = d2{0.0, 0.0};
    __syncthreads();

    constexpr int TPR = 256 / TR;  // threads per row
    const int rr = tid / TPR, e_first = tid % TPR;
    // entry prefetched for the running block: (cols, 3 values of the row's block row)
    int4 pcols = int4{-1, -1, -1, -1};
    double pv0 = 0.0, pv1 = 0.0, pv2 = 0.0, prhs = 0.0;
    auto fetch = [&](int c0) {
        pcols = int4{-1, -1, -1, -1};
        const int c = c0 + rr;
        if (c < npad) {
            const int2 info = dir[c];
            const int count = info.y & 0xffff, r = info.y >> 16;
            if (e_first < count) {
                const size_t ent = (size_t)info.x + e_first;
                pcols = ent_cols[ent];
                const double* v = ent_vals + 6 * ent;  // row r of [xx xy xz; xy yy yz; xz yz zz]
                pv0 = v[r];
                pv1 = v[r == 0 ? 1 : (r == 1 ? 3 : 4)];
                pv2 = v[r == 0 ? 2 : (r == 1 ? 4 : 5)];
            }
        }
        if (tid < TR && c0 + tid < n) prhs = rowrhs[c0 + tid];
    };
    const bool full = (flags & TRS_ASM_FULL_SYMMETRIC) != 0;
    const bool has_env = env_all != nullptr && !full;
    const TrsEnv env = has_env ? trs_env_of(env_all, b, n_pad_max) : TrsEnv{nullptr, nullptr, 0};
    fetch(0);
    for (int c0 = 0; c0 < npad; c0 += TR) {
        // stored part of these rows: columns [i_lo, i_hi) (diagonal tile .. end of the envelope of
        // the panel the rows belong to) followed in the tile by the 16-wide load-column chunk
        const int i_lo = full ? 0 : (c0 & ~15);
        const int i_hi = has_env ? 16 * trs_env_row_end(env, c0 / TRS_NB, npad / 16) : npad;
        const int Wm = i_hi - i_lo;  // multiple of 16
        const int W = Wm + 16;
        {   // scatter: every (row, column) of the tile is written by exactly one thread
            double* row = T + (size_t)rr * W - i_lo;
            if (pcols.x >= i_lo && pcols.x < i_hi) row[pcols.x] = pv0;
            if (pcols.y >= i_lo && pcols.y < i_hi) row[pcols.y] = pv1;
            if (pcols.z >= i_lo && pcols.z < i_hi) row[pcols.z] = pv2;
            const int c = c0 + rr;
            if (c < n) {  // joints with more than TPR - 1 neighbours: rare, not prefetched
                const int2 info = dir[c];
                const int count = info.y & 0xffff, r = info.y >> 16;
                for (int e = e_first + TPR; e < count; e += TPR) {
                    const size_t ent = (size_t)info.x + e;
                    const int4 cols = ent_cols[ent];
                    const double* v = ent_vals + 6 * ent;
                    if (cols.x >= i_lo && cols.x < i_hi) row[cols.x] = v[r];
                    if (cols.y >= i_lo && cols.y < i_hi) row[cols.y] = v[r == 0 ? 1 : (r == 1 ? 3 : 4)];
                    if (cols.z >= i_lo && cols.z < i_hi) row[cols.z] = v[r == 0 ? 2 : (r == 1 ? 4 : 5)];
                }
            }
            if (tid < TR) {
                const int cc = c0 + tid;
                T[(size_t)tid * W + Wm] = cc < n ? prhs : 0.0;            // load column
                if (cc >= n) T[(size_t)tid * W + cc - i_lo] = 1.0;        // identity padding
            }
        }
        if (c0 + TR < npad) fetch(c0 + TR);  // next block's loads fly during the store phase
        __syncthreads();
        {   // all TR rows in parallel: TPR threads per row, 16 bytes per thread and pass
            double* dst = S + (size_t)(c0 + rr) * ld;
            double* src = T + (size_t)rr * W;
            for (int x = e_first * 2; x < W; x += 2 * TPR) {
                const int col = x < Wm ? i_lo + x : npad + (x - Wm);  // envelope part | load column
                *reinterpret_cast<d2*>(dst + col) = *reinterpret_cast<const d2*>(src + x);
                *reinterpret_cast<d2*>(src + x) = d2{0.0, 0.0};
            }
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" size_t trs_assemble_work_bytes(int nJ_max, int nM_max, int n_max) {
    return asm_work_layout(nJ_max, nM_max, trs_round_up(n_max < 1 ? 1 : n_max, TRS_NB)).total;
}

extern "C" int trs_assemble_launch(int B, int nJ_max, int nM_max, const double* xyz, const int* conn,
                                   const double* E, const double* A, const double* loads,
                                   const int* free_index, const int* n_free, const int* nJ,
                                   const int* nM, int ld, size_t slab_stride, int n_pad_max,
                                   double* S, int flags, void* work, int* env, hipStream_t stream) {
    if (B <= 0 || n_pad_max <= 0) return 0;
    if (nJ_max >= 65536 || nM_max >= 65536) return (int)hipErrorInvalidValue;  // packed adjacency keys
    const size_t lds_ints = (size_t)(6 * nJ_max + 1 + 2 * nM_max + n_pad_max / 16) * 4;
    const int geom_in_lds = (size_t)nM_max * 32 + lds_ints <= 64 * 1024;
    const size_t lds1 = lds_ints + (geom_in_lds ? (size_t)nM_max * 32 : 0);
    if (lds1 > 160 * 1024) return (int)hipErrorInvalidValue;
    if (lds1 > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(trs_joint_blocks_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
    hipLaunchKernelGGL(trs_joint_blocks_kernel, dim3(B), dim3(256), lds1, stream, xyz, conn, E, A,
                       loads, free_index, n_free, nJ, nM, nJ_max, nM_max, n_pad_max,
                       static_cast<unsigned char*>(work), env, geom_in_lds);
    int rc = (int)hipGetLastError();
    if (rc) return rc;
    // Rows per block: the largest TR <= TRS_EXPAND_TR_MAX whose LDS (tile + row directory) stays
    // under 64 KiB, so that at least two persistent work-groups share a CU.
#ifndef TRS_EXPAND_TR_MAX
#define TRS_EXPAND_TR_MAX 8
#endif
    const size_t row_bytes = (size_t)(n_pad_max + 16) * sizeof(double);
    const size_t dir_bytes = (size_t)n_pad_max * 8;
    const unsigned char* w = static_cast<const unsigned char*>(work);
#define TRS_LAUNCH_EXPAND(TRV)                                                                          \
    do {                                                                                                \
        const size_t lds2 = TRV * row_bytes + dir_bytes;                                                \
        if (lds2 > 48 * 1024)                                                                           \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(trs_expand_kernel<TRV>),            \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);           \
        hipLaunchKernelGGL(trs_expand_kernel<TRV>, dim3(B), dim3(256), lds2, stream, w, n_free, nJ_max, \
                           nM_max, n_pad_max, ld, slab_stride, S, flags, env);                          \
    } while (0)
    if (TRS_EXPAND_TR_MAX >= 8 && 8 * row_bytes + dir_bytes <= 65536) {
        TRS_LAUNCH_EXPAND(8);
    } else if (TRS_EXPAND_TR_MAX >= 4 && 4 * row_bytes + dir_bytes <= 65536) {
        TRS_LAUNCH_EXPAND(4);
    } else if (TRS_EXPAND_TR_MAX >= 2 && 2 * row_bytes + dir_bytes <= 65536) {
        TRS_LAUNCH_EXPAND(2);
    } else {
        if (row_bytes + dir_bytes > 160 * 1024) return (int)hipErrorInvalidValue;
        TRS_LAUNCH_EXPAND(1);
    }
#undef TRS_LAUNCH_EXPAND
    return (int)hipGetLastError();
}
