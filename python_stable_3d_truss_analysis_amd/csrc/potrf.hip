// Batched FP64 Cholesky (left-looking, 64-wide panels) on v_mfma_f64_16x16x4_f64, with the
// right-hand side carried as one extra row chunk so that the forward substitution is free.
//
// Replaces the factorisation inside np.linalg.solve of the reference
// (slientruss3d/truss.py:343; LAPACK dgesv there, Cholesky here: K_ff is SPD).
//
// One work-group (4 waves) per truss; two work-groups per CU so that one group's serial
// 16x16 diagonal factorisations hide under the other's MFMA stream.
//
// Storage (see include/trs_solver.h): S[c][i] row-major, only i >= tile start of c is used.
// With U = L^T stored in place, "row k of S" holds column k of L, so the MFMA operand
// fragment of 16 rows x 4 columns of L is four 128-byte segments:
//     lane l  <-  S[k0 + (l >> 4)][row0 + (l & 15)]  =  L[row0 + (l & 15)][k0 + (l >> 4)].
//
// Accumulators are kept TRANSPOSED ("D-form"): for a 16-row chunk (rows i) and a 16-column
// tile (columns c) of the panel, lane l component r holds P[i = l & 15][c = (l >> 4) + 4 r].
// In that form a tile is directly the B operand of a following MFMA whose k index is the
// panel column c (component r = k-step r), so the triangular solve against the diagonal
// block runs as MFMAs with no data movement, and loads/stores of a tile are 128-byte
// segments of S rows.
#include "trs_common.h"

// Diagnostic builds only (-DTRS_POTRF_STAMPS via tools/build_variants.sh): per-phase wave-cycle
// sums, read back through trs_debug_stamps().  The product library compiles the empty struct.
#ifdef TRS_POTRF_STAMPS
__device__ unsigned long long g_trs_stamps[8];
struct Stamps {
    unsigned long long acc[8], t;
    __device__ __forceinline__ void start() {
        for (int i = 0; i < 8; ++i) acc[i] = 0;
        t = __builtin_amdgcn_s_memtime();
    }
    __device__ __forceinline__ void mark(int i) {
        const unsigned long long n = __builtin_amdgcn_s_memtime();
        acc[i] += n - t;
        t = n;
    }
    __device__ __forceinline__ void flush() {
        if ((threadIdx.x & 63) == 0)
            for (int i = 0; i < 8; ++i) atomicAdd(&g_trs_stamps[i], acc[i]);
    }
};
#else
struct Stamps {
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void flush() {}
};
#endif

namespace {

constexpr int CT = TRS_NB / TRS_TILE;  // 4 column tiles per panel
constexpr int NW = 4;                  // waves per work-group
#ifndef TRS_POTRF_RS
#define TRS_POTRF_RS 4
#endif
#ifndef TRS_POTRF_WAVES_PER_SIMD
#define TRS_POTRF_WAVES_PER_SIMD 2
#endif
#ifndef TRS_POTRF_DEPTH
#define TRS_POTRF_DEPTH 4
#endif
constexpr int RS = TRS_POTRF_RS;       // row-chunk slots per wave (16 rows each), 1..4
constexpr int DEPTH = TRS_POTRF_DEPTH; // k-steps of operand fragments in flight (divides 16)


__device__ __forceinline__ double lane_bcast(double v, int src) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// Cholesky of one symmetric 16x16 tile by the calling wave, plus the inverse of its factor.
// The tile stays in D-form in four registers per lane (t[r] = T[c = lq + 4 r][i = li]); LDS is
// used only to broadcast the pivot row of each step, so the routine adds almost nothing to the
// register pressure of the accumulators around it.
//   t       : in  the symmetric tile; out U = L^T in D-form with exact zeros below the diagonal
//   sc      : scratch, ChScratch layout below
//   wfrag   : receives inv(L) as A-fragments (layout of PanelLds::W[s])
// Returns the 0-based index of the first non-positive pivot, or -1.
struct ChScratch {
    double U[16][17];   // U[k][i] = L[i][k]
    double Wt[16][17];  // Wt[t][c] = inv(L)[t][c]
    double row[16];     // pivot row of the running step
    double rdiag[16];   // 1 / L[j][j]
};

__device__ __forceinline__ int chol16_invert(d4& t, ChScratch& sc, double* wfrag) {
    const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
    int bad = -1;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        // row j of the running tile sits in comp (j >> 2) of the 16 lanes with lq == (j & 3)
        if (lq == (j & 3)) sc.row[li] = t[j >> 2];
        __builtin_amdgcn_wave_barrier();
        double d = sc.row[j];
        if (!(d > 0.0)) {
            if (bad < 0) bad = j;
            d = 1.0;
        }
        const double sq = sqrt(d);
        const double rinv = 1.0 / sq;
        const double lij = sc.row[li] * rinv;  // L[li][j], meaningful for li > j
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = lq + 4 * r;             // D-form row of this component
            const double lcj = sc.row[c] * rinv;  // L[c][j]
            if (c > j) t[r] -= lcj * lij;
            else if (c == j) t[r] = (li == j) ? sq : lij;
        }
        if (lq == (j & 3)) sc.U[j][li] = (li >= j) ? t[j >> 2] : 0.0;
        if (lane == 0) sc.rdiag[j] = rinv;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (lq + 4 * r > li) t[r] = 0.0;
    // W = inv(L) row by row: W[t][c] = (delta(t,c) - sum_{k<t} L[t][k] W[k][c]) / L[t][t].
    // Lane (lq, li) works on column c = li and sums the terms k = lq (mod 4); the four
    // quarter-waves are then added with two cross-lane exchanges.
#pragma unroll
    for (int tt = 0; tt < 16; ++tt) {
        double part = 0.0;
#pragma unroll
        for (int k4 = 0; k4 < tt; k4 += 4) {
            const int k = k4 + lq;
            if (k < tt) part += sc.U[k][tt] * sc.Wt[k][li];
        }
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        const double wt = ((tt == li ? 1.0 : 0.0) - part) * sc.rdiag[tt];
        if (lq == 0) sc.Wt[tt][li] = wt;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    // A-fragment layout: wfrag[r*64 + lane] = W[t = li][c = 4 r + lq]
#pragma unroll
    for (int r = 0; r < 4; ++r) wfrag[r * 64 + lane] = sc.Wt[li][4 * r + lq];
    return bad;
}

struct PanelLds {
    // inv(L_ss) as MFMA A-fragments: W[s][r*64 + lane] = inv(L_ss)[lane & 15][4 r + (lane >> 4)]
    double W[CT][256];
    // -L_{s2,s} (s2 > s), same fragment layout: rows of tile s2 against columns of tile s
    double Lneg[CT][CT][256];
    ChScratch ch;  // scratch of the scalar 16x16 factorisation
    int info;
};

// Address helper: every global access of the factorisation is "wave-uniform offset + the same
// per-lane offset" (lane (lq, li) -> row lq, column li of a 4 x 16 or 16 x 16 block of S).  The
// slab is addressed through a buffer descriptor: the lane part sits in ONE VGPR (voffset), every
// other part of an address is scalar arithmetic on the soffset operand.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct Slab {
    __amdgpu_buffer_rsrc_t rs;
    int ld;          // leading dimension in doubles
    unsigned loff;   // byte offset of this lane inside a block: ((lane >> 4) * ld + (lane & 15)) * 8
    // byte offset of element (row c, column i)
    __device__ __forceinline__ int at(int c, int i) const { return (c * ld + i) * 8; }
    __device__ __forceinline__ double load(int soff) const {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, loff, soff, 0));
    }
    __device__ __forceinline__ void store(int soff, double v) const {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rs, loff, soff, 0);
    }
};

// D-form tile (rows c0 .. c0+15 of S = panel columns, columns i0 .. i0+15 of S = matrix rows):
// comp r of lane (lq, li) <-> S[c0 + lq + 4 r][i0 + li].
__device__ __forceinline__ void tile_rsub(d4& acc, const Slab& S, int c0, int i0) {
    const int o = S.at(c0, i0);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = S.load(o + r * (S.ld * 32)) - acc[r];
}
__device__ __forceinline__ void tile_store(const d4& acc, const Slab& S, int c0, int i0) {
    const int o = S.at(c0, i0);
#pragma unroll
    for (int r = 0; r < 4; ++r) S.store(o + r * (S.ld * 32), acc[r]);
}

// ---- the panel's 64 x 64 diagonal block -------------------------------------------------------
// Wave w (= NT - 1) owns row chunk w of the block and its tiles s = 0 .. w (lower block
// triangle).  Update with the columns left of the panel, then factor tile by tile:
// chol16 of the diagonal tile by its owner, X = W T for the tiles below it, rank-16 update of the
// tiles to the right.  Leaves inv(L_ss) and -L_{s2,s} in LDS for the rows below the block.
// Returns true when a non-positive pivot was met (uniform over the work-group).
template <int NT>
__device__ __forceinline__ bool diag_group(const Slab& S, const int r0, PanelLds& sm, Stamps& st) {
    constexpr int w = NT - 1;
    const int lane = threadIdx.x & 63;
    d4 acc[NT];
#pragma unroll
    for (int s = 0; s < NT; ++s) acc[s] = d4{0.0, 0.0, 0.0, 0.0};

    // Ring of DEPTH k-steps of fragments in flight; r0 / 4 is a multiple of 16, so of DEPTH.
    // Prefetches past k = r0 stay inside the slab (rows < n_pad) and are never used.
    {
        const int step = S.ld * 32;
        int ok = S.at(0, r0);  // rows k0 .. k0+3 of S, column r0: advanced by 4 rows per k-step
        double fb[DEPTH][NT];
#pragma unroll
        for (int d = 0; d < DEPTH - 1; ++d)
#pragma unroll
            for (int s = 0; s < NT; ++s) fb[d][s] = S.load(ok + d * step + 128 * s);
        for (int k0 = 0; k0 < r0; k0 += 4 * DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int nd = (d + DEPTH - 1) % DEPTH;
#pragma unroll
                for (int s = 0; s < NT; ++s) fb[nd][s] = S.load(ok + (d + DEPTH - 1) * step + 128 * s);
#pragma unroll
                for (int s = 0; s < NT; ++s) acc[s] = mfma_f64(fb[d][s], fb[d][w], acc[s]);
            }
            ok += DEPTH * step;
        }
    }
    st.mark(0);
#pragma unroll
    for (int s = 0; s < NT; ++s) tile_rsub(acc[s], S, r0 + 16 * s, r0 + 16 * w);

#pragma unroll
    for (int s = 0; s < CT; ++s) {
        if (s == w) {
            const int bad = chol16_invert(acc[w], sm.ch, sm.W[s]);
            if (bad >= 0 && lane == 0) sm.info = r0 + 16 * s + bad + 1;
        }
        __syncthreads();
        if (sm.info != 0) return true;
        if (s < w) {  // X_s^T = inv(L_ss) T_s^T ; publish -L_{w,s}
            d4 x = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 4; ++r) x = mfma_f64(sm.W[s][r * 64 + lane], acc[s < NT ? s : 0][r], x);
            acc[s < NT ? s : 0] = x;
#pragma unroll
            for (int r = 0; r < 4; ++r) sm.Lneg[w][s][r * 64 + lane] = -x[r];
        }
        __syncthreads();
        if (s < w) {
#pragma unroll
            for (int s2 = s + 1; s2 < NT; ++s2)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[s2] = mfma_f64(sm.Lneg[s2][s][r * 64 + lane], acc[s < NT ? s : 0][r], acc[s2]);
        }
    }
#pragma unroll
    for (int s = 0; s < NT; ++s) tile_store(acc[s], S, r0 + 16 * s, r0 + 16 * w);
    st.mark(1);
    return false;
}

// ---- rows below the diagonal block -------------------------------------------------------------
// The wave owns NV row chunks (16 rows each, 64 rows apart, first at row `rowbase`) and all four
// column tiles of the panel: update, subtract from K, solve against the factored diagonal block
// (fragments left in LDS by diag_group), store.
template <int NV>
__device__ __forceinline__ void panel_group(const Slab& S, const int r0, const int rowbase,
                                            const PanelLds& sm, Stamps& st) {
    const int lane = threadIdx.x & 63;
    d4 acc[NV][CT];
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int s = 0; s < CT; ++s) acc[v][s] = d4{0.0, 0.0, 0.0, 0.0};

    // acc[v][s](c, i) = sum_{k < r0} L[c][k] L[i][k]
    if (r0 > 0) {
        int ob = S.at(0, r0);       // B side: rows k0 .. k0+3 of S, columns of the panel
        int oa = S.at(0, rowbase);  // A side: same rows of S, columns = the wave's matrix rows
        const int step = S.ld * 32;
        double fb[DEPTH][CT], fa[DEPTH][NV];
#pragma unroll
        for (int d = 0; d < DEPTH - 1; ++d) {
#pragma unroll
            for (int s = 0; s < CT; ++s) fb[d][s] = S.load(ob + d * step + 128 * s);
#pragma unroll
            for (int v = 0; v < NV; ++v) fa[d][v] = S.load(oa + d * step + 512 * v);
        }
        for (int k0 = 0; k0 < r0; k0 += 4 * DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int nd = (d + DEPTH - 1) % DEPTH;
#pragma unroll
                for (int s = 0; s < CT; ++s) fb[nd][s] = S.load(ob + (d + DEPTH - 1) * step + 128 * s);
#pragma unroll
                for (int v = 0; v < NV; ++v) fa[nd][v] = S.load(oa + (d + DEPTH - 1) * step + 512 * v);
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int s = 0; s < CT; ++s) acc[v][s] = mfma_f64(fb[d][s], fa[d][v], acc[v][s]);
            }
            ob += DEPTH * step;
            oa += DEPTH * step;
        }
    }

    st.mark(2);
    // acc = K_panel - acc, one row chunk at a time (bounds the loads in flight)
#pragma unroll
    for (int v = 0; v < NV; ++v) {
#pragma unroll
        for (int s = 0; s < CT; ++s) tile_rsub(acc[v][s], S, r0 + 16 * s, rowbase + 64 * v);
        __builtin_amdgcn_sched_barrier(0);
    }

    st.mark(3);
#pragma unroll
    for (int s = 0; s < CT; ++s) {
        double wf[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) wf[r] = sm.W[s][r * 64 + lane];
#pragma unroll
        for (int v = 0; v < NV; ++v) {  // X_s^T = inv(L_ss) T_s^T
            d4 x = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 4; ++r) x = mfma_f64(wf[r], acc[v][s][r], x);
            acc[v][s] = x;
        }
#pragma unroll
        for (int s2 = s + 1; s2 < CT; ++s2) {  // T_{s2}^T -= L_{s2,s} X_s^T
            double lf[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) lf[r] = sm.Lneg[s2][s][r * 64 + lane];
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[v][s2] = mfma_f64(lf[r], acc[v][s][r], acc[v][s2]);
        }
    }

#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int s = 0; s < CT; ++s) tile_store(acc[v][s], S, r0 + 16 * s, rowbase + 64 * v);
    st.mark(4);
}

__global__ __launch_bounds__(NW * 64, TRS_POTRF_WAVES_PER_SIMD) void trs_potrf_kernel(double* __restrict__ S_all,
                                                               const int* __restrict__ n_free,
                                                               const int ld, const size_t slab_stride,
                                                               int* __restrict__ info) {
    __shared__ PanelLds sm;
    const int b = blockIdx.x;
    const int npad = trs_round_up(n_free[b], TRS_NB);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (threadIdx.x == 0) sm.info = 0;
    __syncthreads();
    if (npad == 0) {
        if (threadIdx.x == 0) info[b] = 0;
        return;
    }
    Slab S;
    S.rs = __builtin_amdgcn_make_buffer_rsrc(S_all + (size_t)b * slab_stride, 0,
                                             (int)(slab_stride * sizeof(double)), 0x00020000);
    S.ld = ld;
    S.loff = ((unsigned)(lane >> 4) * (unsigned)ld + (unsigned)(lane & 15)) * 8u;
    const int nchunks = npad / 16 + 1;  // + the right-hand-side chunk at rows n_pad .. n_pad+15

    Stamps st;
    st.start();
    for (int r0 = 0; r0 < npad; r0 += TRS_NB) {
        bool bad;
        switch (wave) {
            case 0: bad = diag_group<1>(S, r0, sm, st); break;
            case 1: bad = diag_group<2>(S, r0, sm, st); break;
            case 2: bad = diag_group<3>(S, r0, sm, st); break;
            default: bad = diag_group<4>(S, r0, sm, st); break;
        }
        if (bad) {
            if (threadIdx.x == 0) info[b] = sm.info;
            return;
        }
        const int below = nchunks - r0 / 16 - CT;  // row chunks under the diagonal block (>= 1)
        for (int g0 = 0; g0 < below; g0 += NW * RS) {
            const int rem = below - g0 - wave;
            const int nv = rem <= 0 ? 0 : min(RS, (rem + NW - 1) / NW);
            const int rowbase = r0 + (CT + g0 + wave) * 16;
            switch (nv) {
                case 0: break;
                case 1: panel_group<1>(S, r0, rowbase, sm, st); break;
#if TRS_POTRF_RS >= 2
                case 2: panel_group<2>(S, r0, rowbase, sm, st); break;
#endif
#if TRS_POTRF_RS >= 3
                case 3: panel_group<3>(S, r0, rowbase, sm, st); break;
#endif
#if TRS_POTRF_RS >= 4
                case 4: panel_group<4>(S, r0, rowbase, sm, st); break;
#endif
                default: break;
            }
        }
        __syncthreads();  // this panel's stores are visible to the next panel's loads
        st.mark(5);
    }
    st.flush();
    if (threadIdx.x == 0) info[b] = 0;
}

}  // namespace

#ifdef TRS_POTRF_STAMPS
extern "C" int trs_debug_stamps(unsigned long long* host_out, int reset) {
    int rc = (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_trs_stamps), sizeof(g_trs_stamps));
    if (reset) {
        unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_trs_stamps), zero, sizeof(zero));
    }
    return rc;
}
#endif

extern "C" int trs_potrf_launch(int B, const int* n_free, int ld, size_t slab_stride, double* S,
                                int* info, hipStream_t stream) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(trs_potrf_kernel, dim3(B), dim3(NW * 64), 0, stream, S, n_free, ld,
                       slab_stride, info);
    return (int)hipGetLastError();
}
