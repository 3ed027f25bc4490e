// Batched FP64 Cholesky (left-looking, 64-wide panels) on v_mfma_f64_16x16x4_f64, with the
// right-hand side carried as one extra row chunk so that the forward substitution is free.
//
// Replaces the factorisation inside np.linalg.solve of the reference
// (slientruss3d/truss.py:343; LAPACK dgesv there, Cholesky here: K_ff is SPD).
//
// One work-group (4 waves) per truss, two or three work-groups per CU.  Per 64-column panel:
//   D  the four waves update the 64 x 64 diagonal block (5 of its 10 lower tiles x half of the
//      k range each) and hand the partial tiles to wave 0 through LDS;
//   F  wave 0 alone factors the block (four 16 x 16 scalar factorisations + 64 MFMAs) and leaves
//      inv(L_ss) and L_{s2,s} as MFMA operand fragments in LDS;
//   I  meanwhile the other waves (and wave 0 once F is done) pull 64-row work items from an LDS
//      queue: stream the update  K - L[rows, :r0] L[panel, :r0]^T  through the MFMA pipe, wait for
//      F, solve against the diagonal block with MFMAs, store.
// The serial part F is off the other waves' critical path; the queue balances the waves.
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
#include "../../include/trs_solver.h"
#include "trs_common.h"
#include "trs_subst.h"
#include "trs_chol16.h"

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
    // attribute memory waits to the phase that issued the accesses
    __device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
};
#else
struct Stamps {
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void flush() {}
    __device__ __forceinline__ void drain() {}
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
static_assert(RS >= 1 && RS <= TRS_WIDE_ITEM, "the envelope slack covers items of at most TRS_WIDE_ITEM chunks");
constexpr int DEPTH = TRS_POTRF_DEPTH; // k-steps of operand fragments in flight (divides 16)


// pair index of a strictly-lower tile (u, s), s < u < 4, and of a lower tile incl. diagonal
__device__ __forceinline__ constexpr int lf_idx(int u, int s) { return u * (u - 1) / 2 + s; }
__device__ __forceinline__ constexpr int t_idx(int u, int s) { return u * (u + 1) / 2 + s; }

struct PanelLds {
    // diagonal-block tiles (D-form), two k-half partials, handed from the D waves to the factor
    // wave: T[half][t_idx(u,s)].  Once the factor wave holds them in registers the same memory is
    // reused for the operand fragments it produces (the D waves write T again only after the
    // panel's closing barrier):
    //   W(s)   = T[0][s]      inv(L_ss) as MFMA A-fragments, swizzled (wfrag_index / wfrag_lane, trs_chol16.h)
    //   Lf(i)  = T[0][4 + i]  L_{u,s} (u > s), i = lf_idx(u,s): the D-form registers of the tile
    //                          (rows of tile u, columns of tile s) = its A-fragments
    double T[2][10][256];
    __device__ __forceinline__ double* W(int s) { return T[0][s]; }
    __device__ __forceinline__ const double* W(int s) const { return T[0][s]; }
    __device__ __forceinline__ double* Lf(int i) { return T[0][4 + i]; }
    __device__ __forceinline__ const double* Lf(int i) const { return T[0][4 + i]; }
    ChScratch ch;  // scratch of the scalar 16x16 factorisation
    int info;      // 1-based column of the first non-positive pivot, 0 = none
    int d_done;    // diagonal chunks updated so far (4 per panel, monotone over panels)
    int f_done;    // panels whose diagonal block is factored (monotone)
    int queue;     // next work item (monotone over panels)
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
    // Tiles outside the stored envelope are skipped WITHOUT branches: the buffer range check is on
    // the VGPR offset, so an access with lane offset `gone` returns 0 / is dropped and makes no
    // memory traffic (probed with tools/buffer_oob_test.hip).
    static constexpr unsigned gone = 0x80000000u;
    __device__ __forceinline__ unsigned lane_off(bool exists) const { return exists ? loff : gone; }
    __device__ __forceinline__ double load_at(unsigned voff, int soff) const {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0));
    }
    __device__ __forceinline__ void store_at(unsigned voff, int soff, double v) const {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rs, voff, soff, 0);
    }
};

// D-form tile (rows c0 .. c0+15 of S = panel columns, columns i0 .. i0+15 of S = matrix rows):
// comp r of lane (lq, li) <-> S[c0 + lq + 4 r][i0 + li].
__device__ __forceinline__ void tile_load(d4& acc, const Slab& S, int c0, int i0) {
    const int o = S.at(c0, i0);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = S.load(o + r * (S.ld * 32));
}
__device__ __forceinline__ void tile_store(const d4& acc, const Slab& S, int c0, int i0) {
    const int o = S.at(c0, i0);
#pragma unroll
    for (int r = 0; r < 4; ++r) S.store(o + r * (S.ld * 32), acc[r]);
}

// LDS hand-off flags (work-group scope; LDS is coherent inside a work-group).
__device__ __forceinline__ void lds_signal_add(int* flag) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_wait_ge(int* flag, int target) {
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target)
        __builtin_amdgcn_s_sleep(4);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// ---- D: update of the panel's 64 x 64 diagonal block, split 2 x 2 over the four waves ------------
// The ten lower 16 x 16 tiles (u, s), s <= u, are cut into two sets of five and the k range
// [kstart, r0) into two halves (multiples of 16 columns); wave (2*SET + HALF) accumulates its five
// tiles over its half [kbeg, kend).  HALF 0 starts from the K tiles, HALF 1 from zero; the factor
// wave adds the two partials (fixed order, so the result is reproducible).  2.5 tile-streams per
// wave instead of 4 on the busiest one.  kstart > 0 when the envelope says that the block's rows
// have no non-zero left of it.
//   T_{u,s} = K_{u,s} - sum_{k < r0} L[chunk u][k] L[tile s][k]^T
template <int SET, int HALF>
__device__ __forceinline__ void diag_update(const Slab& S, const int r0, const int kbeg, const int kend,
                                            PanelLds& sm, Stamps& st) {
    constexpr int NT = 5;
    constexpr int NFB = SET == 0 ? 3 : 4;  // chunks of the block whose fragments this set needs
    constexpr int TU[2][NT] = {{0, 1, 1, 2, 2}, {2, 3, 3, 3, 3}};
    constexpr int TS[2][NT] = {{0, 0, 1, 0, 1}, {2, 0, 1, 2, 3}};
    const int lane = threadIdx.x & 63;
    d4 acc[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) {
        if (HALF == 0) tile_load(acc[q], S, r0 + 16 * TS[SET][q], r0 + 16 * TU[SET][q]);
        else acc[q] = d4{0.0, 0.0, 0.0, 0.0};
    }
    if (kend > kbeg) {
        // Ring of DEPTH k-steps of fragments in flight; kend - kbeg is a multiple of 16 columns =
        // DEPTH k-steps.  Prefetches past kend stay inside the slab and are never used.
        const int step = S.ld * 32;
        int ok = S.at(kbeg, r0);  // rows k0 .. k0+3 of S, column r0
        double fb[DEPTH][NFB];
#pragma unroll
        for (int d = 0; d < DEPTH - 1; ++d)
#pragma unroll
            for (int c = 0; c < NFB; ++c) fb[d][c] = S.load(ok + d * step + 128 * c);
        for (int k0 = kbeg; k0 < kend; k0 += 4 * DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int nd = (d + DEPTH - 1) % DEPTH;
#pragma unroll
                for (int c = 0; c < NFB; ++c) fb[nd][c] = S.load(ok + (d + DEPTH - 1) * step + 128 * c);
#pragma unroll
                for (int q = 0; q < NT; ++q)
                    acc[q] = mfma_f64_negA(fb[d][TS[SET][q]], fb[d][TU[SET][q]], acc[q]);
            }
            ok += DEPTH * step;
        }
    }
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) sm.T[HALF][t_idx(TU[SET][q], TS[SET][q])][r * 64 + lane] = acc[q][r];
    lds_signal_add(&sm.d_done);
    st.mark(0);
}

// ---- F: factor the 64 x 64 diagonal block (one wave, everything in registers) -------------------
// t[u][s] (s <= u) are the lower tiles in D-form.  For s = 0..3: scalar Cholesky of T_ss, then
// X_{u,s}^T = inv(L_ss) T_{u,s}^T for the tiles below it and the rank-16 update of the tiles to the
// right; a D-form register r of X_{u,s} is at the same time the A-fragment (k-step r) of L_{u,s}.
__device__ __forceinline__ void factor_block(const Slab& S, const int r0, PanelLds& sm, Stamps& st) {
    const int lane = threadIdx.x & 63;
    d4 t[CT][CT];
#pragma unroll
    for (int u = 0; u < CT; ++u)
#pragma unroll
        for (int s = 0; s <= u; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                t[u][s][r] = sm.T[0][t_idx(u, s)][r * 64 + lane] + sm.T[1][t_idx(u, s)][r * 64 + lane];
    bool ok = true;
#pragma unroll
    for (int s = 0; s < CT; ++s) {
        if (ok) {
            st.mark(1);
            const Chol16 f = chol16_invert(t[s][s], sm.ch, sm.W(s));
            t[s][s] = f.u;
            __builtin_amdgcn_wave_barrier();
            const int bad = f.bad;
            st.mark(7);
            if (bad >= 0) {
                if (lane == 0) sm.info = r0 + 16 * s + bad + 1;
                ok = false;
            }
        }
        if (ok) {
            double wf[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) wf[r] = sm.W(s)[wfrag_lane(r, lane)];
#pragma unroll
            for (int u = s + 1; u < CT; ++u) {
                d4 x = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int r = 0; r < 4; ++r) x = mfma_f64(wf[r], t[u][s][r], x);
                t[u][s] = x;
#pragma unroll
                for (int r = 0; r < 4; ++r) sm.Lf(lf_idx(u, s))[r * 64 + lane] = x[r];
            }
#pragma unroll
            for (int u = s + 1; u < CT; ++u)
#pragma unroll
                for (int s2 = s + 1; s2 <= u; ++s2)
#pragma unroll
                    for (int r = 0; r < 4; ++r) t[u][s2] = mfma_f64_negA(t[s2][s][r], t[u][s][r], t[u][s2]);
        }
    }
    if (ok) {
#pragma unroll
        for (int u = 0; u < CT; ++u)
#pragma unroll
            for (int s = 0; s <= u; ++s) tile_store(t[u][s], S, r0 + 16 * s, r0 + 16 * u);
    }
    lds_signal_add(&sm.f_done);
}

// ---- the load vector / forward-substituted load vector as a plain array ---------------------------------
// uf[c] holds f[c] before and y[c] after the factorisation of column c's panel.  As an MFMA operand the
// load vector is a 16-wide chunk of which only column 0 is not zero: lane (lq, li) of a fragment is
// uf[c0 + lq] for li == 0 and zero otherwise - one 8-byte load for four lanes, the rest goes through the
// out-of-range lane offset (no memory traffic).
struct LoadVec {
    __amdgpu_buffer_rsrc_t rs;
    unsigned voff;  // (lane >> 4) * 8 for the lanes li == 0, out of range for the others
    __device__ __forceinline__ double load(int soff) const {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0));
    }
    __device__ __forceinline__ void store(int soff, double v) const {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rs, voff, soff, 0);
    }
};
// D-form tile of the load vector for the 16 columns c0 .. c0+15: comp r of lane (lq, 0) <-> uf[c0 + lq + 4 r]
__device__ __forceinline__ void ytile_load(d4& acc, const LoadVec& Y, int c0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = Y.load((c0 + 4 * r) * 8);
}
__device__ __forceinline__ void ytile_store(const d4& acc, const LoadVec& Y, int c0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) Y.store((c0 + 4 * r) * 8, acc[r]);
}

// ---- I: one work item = NV consecutive 16-row chunks below the diagonal block ---------------------
// Update (streams L from HBM, B-side fragments shared through L1/L2), wait for the factor wave,
// solve against the diagonal block with the fragments it left in LDS, store.
template <int NV>
__device__ __forceinline__ void panel_item(const Slab& S, const int r0, const int rowbase,
                                           const int kstart, PanelLds& sm, const int f_target,
                                           Stamps& st, const LoadVec* Y = nullptr) {
    const int lane = threadIdx.x & 63;
    d4 acc[NV][CT];
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int s = 0; s < CT; ++s) tile_load(acc[v][s], S, r0 + 16 * s, rowbase + 16 * v);

    // acc[v][s](c, i) = K - sum_{kstart <= k < r0} L[c][k] L[i][k]; the item's rows are zero left of
    // kstart (envelope), kstart is a multiple of 16
    if (r0 > kstart) {
        int ob = S.at(kstart, r0);       // B side: rows k0 .. k0+3 of S, columns of the panel
        int oa = S.at(kstart, rowbase);  // A side: same rows of S, columns = the item's matrix rows
        const int step = S.ld * 32;
        double fb[DEPTH][CT], fa[DEPTH][NV];
#pragma unroll
        for (int d = 0; d < DEPTH - 1; ++d) {
#pragma unroll
            for (int s = 0; s < CT; ++s) fb[d][s] = S.load(ob + d * step + 128 * s);
#pragma unroll
            for (int v = 0; v < NV; ++v) fa[d][v] = S.load(oa + d * step + 128 * v);
        }
        for (int k0 = kstart; k0 < r0; k0 += 4 * DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int nd = (d + DEPTH - 1) % DEPTH;
#pragma unroll
                for (int s = 0; s < CT; ++s) fb[nd][s] = S.load(ob + (d + DEPTH - 1) * step + 128 * s);
#pragma unroll
                for (int v = 0; v < NV; ++v) fa[nd][v] = S.load(oa + (d + DEPTH - 1) * step + 128 * v);
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int s = 0; s < CT; ++s) acc[v][s] = mfma_f64_negA(fb[d][s], fa[d][v], acc[v][s]);
            }
            ob += DEPTH * step;
            oa += DEPTH * step;
        }
    }
    st.mark(2);
    lds_wait_ge(&sm.f_done, f_target);
    st.mark(3);
    if (sm.info != 0) return;

#pragma unroll
    for (int s = 0; s < CT; ++s) {
        double wf[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) wf[r] = sm.W(s)[wfrag_lane(r, lane)];
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
            for (int r = 0; r < 4; ++r) lf[r] = sm.Lf(lf_idx(s2, s))[r * 64 + lane];
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[v][s2] = mfma_f64_negA(lf[r], acc[v][s][r], acc[v][s2]);
        }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int s = 0; s < CT; ++s) tile_store(acc[v][s], S, r0 + 16 * s, rowbase + 16 * v);
    if (Y != nullptr) {  // the load column's item: y also goes to uf, where trs_potrs reads it
#pragma unroll
        for (int s = 0; s < CT; ++s) ytile_store(acc[0][s], *Y, r0 + 16 * s);
    }
    st.mark(4);
}

// ---- which kernel factors a matrix --------------------------------------------------------------------
// A matrix whose envelope reaches at most TRS_NARROW_MAX_BELOW row chunks below every diagonal block
// has so little MFMA work per panel that one wave can carry it alone (trs_potrf_narrow_kernel: one
// WAVE per matrix, no barriers, the serial 16x16 factorisations of up to 12 matrices per CU overlap);
// all others go to trs_potrf_kernel (one WORK-GROUP per matrix).  trs_assemble makes the choice
// (it shapes the stored part of the slab for the chosen kernel: exact tile envelope / rectangular per
// panel) and records it in the envelope metadata; both kernels are launched and each skips the
// other's matrices.

__global__ __launch_bounds__(NW * 64, TRS_POTRF_WAVES_PER_SIMD) void trs_potrf_kernel(
    double* __restrict__ S_all, const int* __restrict__ n_free, const int ld, const size_t slab_stride,
    int* __restrict__ info, const int* __restrict__ env_all, const int n_pad_max,
    double* __restrict__ uf_all, const int ld_uf) {
    __shared__ PanelLds sm;
    const int b = blockIdx.x;
    const int npad = trs_round_up(n_free[b], TRS_NB);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (threadIdx.x == 0) {
        sm.info = 0;
        sm.d_done = 0;
        sm.f_done = 0;
        sm.queue = 0;
    }
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
    const int nch = npad / 16;  // row chunks of the matrix; the right-hand side is one more, at row n_pad
    const bool has_env = env_all != nullptr;
    const TrsEnv env = has_env ? trs_env_of(env_all, b, n_pad_max) : TrsEnv{nullptr, nullptr, nullptr, 0};
    if (has_env && trs_env_is_narrow(env)) return;  // trs_potrf_narrow_kernel's matrix
    LoadVec Y;
    Y.rs = __builtin_amdgcn_make_buffer_rsrc(uf_all + (size_t)b * ld_uf, 0, ld_uf * (int)sizeof(double), 0x00020000);
    Y.voff = (lane & 15) == 0 ? (unsigned)(lane >> 4) * 8u : Slab::gone;

    Stamps st;
    st.start();
    int qbase = 0;
    for (int r0 = 0, panel = 0; r0 < npad; r0 += TRS_NB, ++panel) {
        {   // D: k range [kstart, r0) in two halves of whole 16-column tiles
            const int kstart = has_env ? 16 * env.ft[4 * panel] : 0;
            const int kmid = kstart + 16 * (((r0 - kstart) / 16 + 1) / 2);
            switch (wave) {
                case 0: diag_update<0, 0>(S, r0, kstart, kmid, sm, st); break;
                case 1: diag_update<0, 1>(S, r0, kmid, r0, sm, st); break;
                case 2: diag_update<1, 0>(S, r0, kstart, kmid, sm, st); break;
                default: diag_update<1, 1>(S, r0, kmid, r0, sm, st); break;
            }
        }
        if (wave == 0) {
            lds_wait_ge(&sm.d_done, 4 * (panel + 1));
            st.mark(6);
            // the serial factorisation is the panel's critical path: let its VALU stream win
            // issue arbitration against the co-resident work-group's MFMA stream on this SIMD
            __builtin_amdgcn_s_setprio(3);
            factor_block(S, r0, sm, st);
            __builtin_amdgcn_s_setprio(0);
            st.mark(1);
        }
        // Items: the row chunks under the diagonal block that reach into this panel (all of them
        // for a dense matrix, up to last[panel] with an envelope) in groups of RS, then the
        // right-hand-side chunk as an item of its own.
        const int lastq = has_env ? env.last[panel] : nch - 1;
        const int below = lastq - (4 * panel + CT - 1);  // >= 0
        const int nmain = (below + RS - 1) / RS;
        const int nitems = nmain + 1;
        for (;;) {
            int item = 0;
            if (lane == 0) item = __hip_atomic_fetch_add(&sm.queue, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            item = __builtin_amdgcn_readfirstlane(item) - qbase;
            if (item >= nitems) break;
            if (item == nmain) {
                // L y = f rides along as the row n_pad.  Its own envelope is dense, but the panel's
                // rows are zero (and unwritten) left of their envelope: start there.
                panel_item<1>(S, r0, npad, has_env ? 16 * env.ft[4 * panel] : 0, sm, panel + 1, st, &Y);
                continue;
            }
            const int c0 = 4 * panel + CT + item * RS;  // first chunk of the item
            const int nv = min(RS, below - item * RS);
            const int rowbase = c0 * 16;
            const int kstart = has_env ? 16 * env.ft[c0] : 0;
            switch (nv) {
                case 1: panel_item<1>(S, r0, rowbase, kstart, sm, panel + 1, st); break;
#if TRS_POTRF_RS >= 2
                case 2: panel_item<2>(S, r0, rowbase, kstart, sm, panel + 1, st); break;
#endif
#if TRS_POTRF_RS >= 3
                case 3: panel_item<3>(S, r0, rowbase, kstart, sm, panel + 1, st); break;
#endif
#if TRS_POTRF_RS >= 4
                case 4: panel_item<4>(S, r0, rowbase, kstart, sm, panel + 1, st); break;
#endif
                default: break;
            }
        }
        // every wave overshoots the queue by exactly one failed pull per panel
        qbase += nitems + NW;
        __syncthreads();  // this panel's stores are visible to the next panel's loads; LDS reusable
        st.mark(5);
        if (sm.info != 0) break;
    }
    st.flush();
    if (threadIdx.x == 0) info[b] = sm.info;
}

// ---- stiffness tiles formed from the compact per-tile entry lists (trs_common.h, TrsCompactLayout) ------
// The tiles of one slab chunk are numbered consecutively and their entries lie back to back, so the
// tiles a panel needs from a chunk (the diagonal block's column of tiles; an item's one or two tiles)
// are ONE contiguous entry range.  The wave requests a range's entries in rounds of 64 (all rounds in
// flight together), scatters them into a zeroed LDS image of up to four tiles in D-form order
// (ds_write_b64 at the slot the assembly recorded: (tile - first tile) * 256 + r * 64 + lq * 16 + li),
// reads the tiles back as accumulator registers and re-zeroes the image behind the reads.  LDS
// operations of one wave complete in order, so no barrier is needed between the steps.
struct KLists {
    const int* tdesc;              // (first entry, count) per tile
    const int* tbase;              // first tile id per slab chunk
    __amdgpu_buffer_rsrc_t vals;   // double[]
    __amdgpu_buffer_rsrc_t slots;  // unsigned short[]: (tile - chunk) << 8 | D-form slot
    lds_f64* img;                  // the image being formed (all zero before a range is scattered)
};
template <int RPF>
struct RangeFetch {  // the first RPF rounds of a range's entries, in flight
    double v[RPF];
    unsigned p[RPF];
    int beg, cnt;
};
template <int RPF>
__device__ __forceinline__ RangeFetch<RPF> krange_issue(const KLists& K, const int first, const int last,
                                                        const bool exists = true) {
    RangeFetch<RPF> f;
    f.beg = 0;
    f.cnt = 0;
    if (exists) {  // wave-uniform
        f.beg = K.tdesc[2 * first];
        f.cnt = K.tdesc[2 * last] + K.tdesc[2 * last + 1] - f.beg;
    }
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < RPF; ++j) {
        const unsigned e = (unsigned)(f.beg + 64 * j + lane);
        const unsigned gone = 64 * j + lane < f.cnt ? 0u : 0x80000000u;  // past the range: no memory traffic
        f.v[j] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(K.vals, (e * 8u) | gone, 0, 0));
        f.p[j] = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(K.slots, (e * 2u) | gone, 0, 0);
    }
    return f;
}
// `sub` = (first tile - chunk) << 8: the image starts at the range's first tile
template <int RPF>
__device__ __forceinline__ void krange_scatter(const KLists& K, const RangeFetch<RPF>& f, const int sub) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < RPF; ++j)
        if (64 * j + lane < f.cnt) K.img[(int)f.p[j] - sub] = f.v[j];
    for (int e0 = 64 * RPF; e0 < f.cnt; e0 += 64) {  // denser than expected: further rounds, one at a time
        const unsigned e = (unsigned)(f.beg + e0 + lane);
        if (e0 + lane < f.cnt) {
            const double v = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(K.vals, e * 8u, 0, 0));
            const int pp = (int)__builtin_amdgcn_raw_buffer_load_b16(K.slots, e * 2u, 0, 0);
            K.img[pp - sub] = v;
        }
    }
    __builtin_amdgcn_wave_barrier();
}
// tile k of the image -> accumulator registers; REZERO: the tile is zero again afterwards
template <bool REZERO>
__device__ __forceinline__ void ktile_take(d4& acc, const KLists& K, const int k) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = K.img[k * 256 + r * 64 + lane];
    if constexpr (REZERO) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 4; ++r) K.img[k * 256 + r * 64 + lane] = 0.0;
    }
}
// zero what krange_scatter<RPF> wrote (ranges that fit their prefetched rounds; otherwise whole tiles)
template <int RPF>
__device__ __forceinline__ void krange_unscatter(const KLists& K, const RangeFetch<RPF>& f, const int sub,
                                                 const int ntile) {
    const int lane = threadIdx.x & 63;
    __builtin_amdgcn_wave_barrier();
    if (f.cnt <= 64 * RPF) {
#pragma unroll
        for (int j = 0; j < RPF; ++j)
            if (64 * j + lane < f.cnt) K.img[(int)f.p[j] - sub] = 0.0;
    } else {
        for (int x = lane; x < 256 * ntile; x += 64) K.img[x] = 0.0;
    }
}

// ======================================================================================================
// Narrow-envelope variant: one WAVE per matrix, four matrices per work-group, no barriers, no
// hand-offs between waves.  The wave keeps the panel's ten diagonal-block tiles (the six L_{u,s}
// among them are MFMA operand fragments as they stand) in registers; its private LDS slice holds the
// four inv(L_ss) as operand fragments (swizzled, trs_chol16.h) and the scratch of chol16_invert.
// 168 VGPRs: three waves per SIMD.  The load vector (L y = f) rides in the diagonal-block pass as four
// scalars per lane; every access to a tile outside the stored envelope goes through the out-of-range
// lane offset (Slab::gone), and no MFMA is issued for an all-zero operand tile.
// ======================================================================================================
#ifndef TRS_NARROW_DEPTH
#define TRS_NARROW_DEPTH 2
#endif
#ifndef TRS_NARROW_WAVES_PER_SIMD
#define TRS_NARROW_WAVES_PER_SIMD 3   // 168 VGPRs: the slab form fits with two registers spilled
#endif
constexpr int DEPTHN = TRS_NARROW_DEPTH;  // k-steps of fragments in flight (divides 4 k-steps = 16 columns)

// One item of the narrow kernel: NV row chunks c0, c0+1 below the diagonal block of panel r0 / 64.
// Tile (chunk q, column tile tt) is stored iff q < cend[tt] (trs_common.h); what is not stored is an
// exact zero of L: its loads return 0 and its stores are dropped through the lane offset.
//   kstart = 16 ft[c0]: first column of chunk c0's envelope; the chunks after it may start later.
template <int NV, bool FUSED>
__device__ __forceinline__ void narrow_item(const Slab& S, const int r0, const int c0, const int kstart,
                                            const int* __restrict__ ft, const int* __restrict__ cend,
                                            const double* Wl, const d4 (&t)[CT][CT], const KLists& K,
                                            const int (&tb)[CT], const int* __restrict__ kmask) {
    const int rowbase = 16 * c0, lane = threadIdx.x & 63;
    d4 acc[NV][CT];
    // cend is non-decreasing: chunk c0+v is stored in the panel's tiles smin[v] .. 3 (one scalar
    // per chunk instead of a flag and a lane offset per tile: the kernel is short of SGPRs)
    int smin[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) smin[v] = 0;
#pragma unroll
    for (int s = 0; s < CT; ++s) {
        const int ce = cend[r0 / 16 + s];
#pragma unroll
        for (int v = 0; v < NV; ++v) smin[v] += c0 + v >= ce ? 1 : 0;
    }
    if constexpr (FUSED) {
        // The stiffness tiles of the item come from the entry lists: per slab chunk s the item's tiles
        // (chunk c0 and, if stored, c0 + 1: cend is non-decreasing) are one entry range.  All four ranges
        // are requested first, then each is scattered into the image and taken tile by tile.
        static_assert(NV <= 2, "an item's tiles of one slab chunk must fit the image");
        RangeFetch<1> rf[CT];
        int nvs[CT];
#pragma unroll
        for (int s = 0; s < CT; ++s) {
            nvs[s] = 0;
#pragma unroll
            for (int v = 0; v < NV; ++v) nvs[s] += s >= smin[v] ? 1 : 0;
            const int first = tb[s] + c0 - (r0 / 16 + s);
            rf[s] = krange_issue<1>(K, first, first + nvs[s] - 1, nvs[s] > 0);
        }
#pragma unroll
        for (int s = 0; s < CT; ++s) {
            if (nvs[s] > 0) krange_scatter<1>(K, rf[s], (c0 - (r0 / 16 + s)) << 8);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (s >= smin[v]) ktile_take<false>(acc[v][s], K, v);
                else acc[v][s] = d4{0.0, 0.0, 0.0, 0.0};
            }
            if (nvs[s] > 0) krange_unscatter<1>(K, rf[s], (c0 - (r0 / 16 + s)) << 8, NV);
        }
    } else {
#pragma unroll
        for (int s = 0; s < CT; ++s) {
            // stiffness tiles that hold no entry of K_ff were not written by trs_assemble (kmask, trs_common.h):
            // they are zeros, taken through the out-of-range lane offset like the tiles outside the envelope
            // (a mask word covers 32 tiles; only a matrix FORCED narrow reaches further, and its tiles are all written)
            const unsigned km = (unsigned)kmask[r0 / 16 + s], d0 = (unsigned)(c0 - (r0 / 16 + s));
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int o = S.at(r0 + 16 * s, rowbase + 16 * v);
                const unsigned vo = S.lane_off(s >= smin[v] && (d0 + v >= 32u || ((km >> (d0 + v)) & 1u) != 0u));
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[v][s][r] = S.load_at(vo, o + r * (S.ld * 32));
            }
        }
    }
    if (r0 > kstart) {
        int ob = S.at(kstart, r0);
        int oa = S.at(kstart, rowbase);
        const int step = S.ld * 32;
        double fb[DEPTHN][CT], fa[DEPTHN][NV];
        // chunk c0's rows exist from kstart on, and so do the block's (ft is non-decreasing); chunk
        // c0+v only from column 16 ft[c0+v] on
        int kv[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) kv[v] = 16 * ft[c0 + v];
        auto aload = [&](int off, int v, int k) {
            return S.load_at(S.lane_off(v == 0 || k >= kv[v]), off + 128 * v);
        };
        const unsigned fbo = S.loff;
#pragma unroll
        for (int d = 0; d < DEPTHN - 1; ++d) {
#pragma unroll
            for (int s = 0; s < CT; ++s) fb[d][s] = S.load_at(fbo, ob + d * step + 128 * s);
#pragma unroll
            for (int v = 0; v < NV; ++v) fa[d][v] = aload(oa + d * step, v, kstart + 4 * d);
        }
        for (int k0 = kstart; k0 < r0; k0 += 4 * DEPTHN) {
#pragma unroll
            for (int d = 0; d < DEPTHN; ++d) {
                const int nd = (d + DEPTHN - 1) % DEPTHN;
#pragma unroll
                for (int s = 0; s < CT; ++s)
                    fb[nd][s] = S.load_at(fbo, ob + (d + DEPTHN - 1) * step + 128 * s);
#pragma unroll
                for (int v = 0; v < NV; ++v)
                    fa[nd][v] = aload(oa + (d + DEPTHN - 1) * step, v, k0 + 4 * (d + DEPTHN - 1));
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int s = 0; s < CT; ++s)  // (branching on the stored-tile test here costs 60 VGPRs)
                        acc[v][s] = mfma_f64_negA(fb[d][s], fa[d][v], acc[v][s]);
            }
            ob += DEPTHN * step;
            oa += DEPTHN * step;
        }
    }
#pragma unroll
    for (int s = 0; s < CT; ++s) {
#pragma unroll
        for (int v = 0; v < NV; ++v)  // tiles outside the envelope are zero and stay zero: no work
            if (s >= smin[v]) {
                d4 x = d4{0.0, 0.0, 0.0, 0.0};  // X_s^T = inv(L_ss) T_s^T
#pragma unroll
                for (int r = 0; r < 4; ++r) x = mfma_f64(Wl[s * 256 + wfrag_lane(r, lane)], acc[v][s][r], x);
                acc[v][s] = x;
#pragma unroll
                for (int s2 = s + 1; s2 < CT; ++s2)  // T_{s2}^T -= L_{s2,s} X_s^T
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[v][s2] = mfma_f64_negA(t[s2][s][r], acc[v][s][r], acc[v][s2]);
            }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int s = 0; s < CT; ++s) {
            const int o = S.at(r0 + 16 * s, rowbase + 16 * v);
            const unsigned vo = S.lane_off(s >= smin[v]);
#pragma unroll
            for (int r = 0; r < 4; ++r) S.store_at(vo, o + r * (S.ld * 32), acc[v][s][r]);
        }
}

#ifndef TRS_NARROW_MATRICES_PER_WG
#define TRS_NARROW_MATRICES_PER_WG 4
#endif
constexpr int MPW = TRS_NARROW_MATRICES_PER_WG;  // waves (= matrices) per work-group of the narrow kernel
#ifndef TRS_FUSED_WAVES_PER_SIMD
#define TRS_FUSED_WAVES_PER_SIMD 2
#endif
// FUSED = false: the stiffness tiles are read from the slab (trs_assemble wrote them);
// FUSED = true : the stiffness tiles are formed from the compact entry lists - K_ff never existed in HBM in
//                dense form.  Both read the load vector from uf, leave y there and L in the slab.
// RSV = row chunks per item: 2 (three waves per SIMD) or 4 (two waves per SIMD; the matrices whose envelope
// reaches further below the diagonal blocks, routing bit TRS_ENV_RS4; slab form only).
template <bool FUSED, int RSV>
__global__ __launch_bounds__(64 * MPW, FUSED ? TRS_FUSED_WAVES_PER_SIMD : (RSV > 2 ? 2 : TRS_NARROW_WAVES_PER_SIMD)) void trs_potrf_narrow_kernel(
    double* __restrict__ S_all, const int* __restrict__ n_free, const int ld, const size_t slab_stride,
    int* __restrict__ info, const int* __restrict__ env_all, const int n_pad_max, const int B,
    const unsigned char* __restrict__ work, double* __restrict__ uf_all, const int ld_uf, const int substitute) {
    __shared__ ChScratch scratch[MPW];
    __shared__ double wlds[MPW][CT][256];  // per wave: inv(L_ss), s = 0..3, as A-fragments (PanelLds::W layout)
    __shared__ double kimg[FUSED ? MPW : 1][FUSED ? 512 : 1];  // per wave: image of an item's (up to two) stiffness
                                                               // tiles; the diagonal block's images use wlds
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    // Work-groups are dispatched in blockIdx order; the matrices are taken in DESCENDING order: trs_assemble
    // wrote the highest-numbered trusses last, so their stiffness tiles are the ones still in the memory-side
    // cache when this kernel starts, and the lowest-numbered ones are factored last - where
    // trs_potrs_batched, which runs in ascending order, starts.
    const int b = ((int)gridDim.x - 1 - (int)blockIdx.x) * MPW + wave;
    if (b >= B) return;  // no work-group barrier anywhere in this kernel: waves are independent
    const int npad = trs_round_up(n_free[b], TRS_NB);
    if (npad == 0) {
        if (lane == 0) info[b] = 0;
        return;
    }
    const TrsEnv env = trs_env_of(env_all, b, n_pad_max);
    if (!trs_env_is_narrow(env) || trs_env_is_compact(env) != FUSED) return;  // another kernel's matrix
    if (!FUSED && trs_env_is_rs4(env) != (RSV > 2)) return;                   // the other item size's matrix
    Slab S;
    S.rs = __builtin_amdgcn_make_buffer_rsrc(S_all + (size_t)b * slab_stride, 0,
                                             (int)(slab_stride * sizeof(double)), 0x00020000);
    S.ld = ld;
    S.loff = ((unsigned)(lane >> 4) * (unsigned)ld + (unsigned)(lane & 15)) * 8u;
    // the load vector in uf: YR reads 16 consecutive entries (lane li, replicated over the quarter-waves),
    // YRS stores them (quarter-wave 0 only), YK reads 4 consecutive entries (lane lq, replicated over li)
    const int li = lane & 15, lq = lane >> 4;
    LoadVec YR, YRS, YK;
    YR.rs = YRS.rs = YK.rs =
        __builtin_amdgcn_make_buffer_rsrc(uf_all + (size_t)b * ld_uf, 0, ld_uf * (int)sizeof(double), 0x00020000);
    YR.voff = (unsigned)li * 8u;
    YRS.voff = lq == 0 ? (unsigned)li * 8u : Slab::gone;
    YK.voff = (unsigned)lq * 8u;
    KLists K{};
    if constexpr (FUSED) {
        const int* meta = env.last + n_pad_max / 64;  // slack | list offsets / 16 (written by trs_assemble)
        K.tdesc = reinterpret_cast<const int*>(work + ((size_t)(unsigned)meta[1] << 4));
        K.tbase = reinterpret_cast<const int*>(work + ((size_t)(unsigned)meta[2] << 4));
        K.slots = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(work) + ((size_t)(unsigned)meta[3] << 4),
                                                    0, 0x7fffffff, 0x00020000);
        K.vals = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(work) + ((size_t)(unsigned)meta[4] << 4),
                                                   0, 0x7fffffff, 0x00020000);
        K.img = (lds_f64*)&kimg[wave][0];
#pragma unroll
        for (int r = 0; r < 8; ++r) K.img[r * 64 + lane] = 0.0;
    }
    ChScratch& sc = scratch[wave];
    double* Wl = &wlds[wave][0][0];
    int bad_col = 0;
    Stamps st;  // 0 tile loads, 1 block update, 2 factorisation, 3 load column + block stores, 4 items, 5 fence, 6 substitution
    st.start();

    // The diagonal block's stiffness tiles and load-vector entries of a panel are not touched by the panel before
    // it: in the slab form they are requested at the END of that panel, ahead of the fence that waits for its
    // stores, so that their latency and the drain of the stores overlap instead of following each other.
    d4 t[CT][CT];
    double yr[CT];
    auto load_block = [&](int r0) {
#pragma unroll
        for (int s = 0; s < CT; ++s) {
            // (behind the last panel, r0 = npad, any word will do: the tiles requested there lie outside the slab's
            // descriptor and nobody reads them - the index is only kept inside the array)
            const unsigned km = (unsigned)env.kmask[min(r0 / 16 + s, npad / 16 - 1)];
#pragma unroll
            for (int u = s; u < CT; ++u) {
                const int o = S.at(r0 + 16 * s, r0 + 16 * u);
                const unsigned vo = S.lane_off(((km >> (u - s)) & 1u) != 0u);   // no entry of K_ff in the tile: zeros
#pragma unroll
                for (int r = 0; r < 4; ++r) t[u][s][r] = S.load_at(vo, o + r * (S.ld * 32));
            }
        }
#pragma unroll
        for (int s = 0; s < CT; ++s) yr[s] = YR.load((r0 + 16 * s) * 8);
    };
    if constexpr (!FUSED) load_block(0);
    for (int r0 = 0, panel = 0; r0 < npad && bad_col == 0; r0 += TRS_NB, ++panel) {
        const int kd = 16 * env.ft[4 * panel];
        // first written column of each of the block's four row chunks (their envelope)
        const int bks[CT] = {16 * env.ft[4 * panel], 16 * env.ft[4 * panel + 1], 16 * env.ft[4 * panel + 2],
                             16 * env.ft[4 * panel + 3]};
        // D: the ten lower tiles of the diagonal block and the load vector's 64 entries (L y = f rides along:
        // same k range and the same block-side fragments, which are read once).
        // The load vector is ONE column: carried as four scalars per lane (yr[u] = entry of row li of block
        // chunk u, replicated over the quarter-waves) and advanced with plain FMAs and lane reductions.
        // As a 16-wide MFMA operand chunk (round 1) it cost 104 of the ~240 MFMAs of a panel, 15/16 of
        // them on zero columns, on the FP64 datapath that the matrix core and the VALU share.
        int tb[CT] = {0, 0, 0, 0};  // first tile id of the panel's four slab chunks (FUSED)
        if constexpr (FUSED) {
#pragma unroll
            for (int s = 0; s < CT; ++s) tb[s] = K.tbase[4 * panel + s];
#pragma unroll
            for (int s = 0; s < CT; ++s) yr[s] = YR.load((r0 + 16 * s) * 8);
            // Column s of the diagonal block = tiles (slab chunk 4 panel + s, matrix rows chunk 4 panel + u),
            // u = s .. 3: ids tb[s] .. tb[s] + 3 - s, one entry range.  All four ranges are in flight before
            // the first image is formed.  (Requesting them a whole panel ahead was measured: no gain, 38
            // registers.)  The images are formed in the inv(L_ss) buffer, which is dead until the block is
            // factored and is overwritten then: it is zeroed before, not after, its use.
            const RangeFetch<4> f0 = krange_issue<4>(K, tb[0], tb[0] + 3);
            const RangeFetch<3> f1 = krange_issue<3>(K, tb[1], tb[1] + 2);
            const RangeFetch<3> f2 = krange_issue<3>(K, tb[2], tb[2] + 1);
            const RangeFetch<2> f3 = krange_issue<2>(K, tb[3], tb[3]);
            KLists Kd = K;
            Kd.img = (lds_f64*)Wl;
#pragma unroll
            for (int r = 0; r < 16; ++r) Kd.img[r * 64 + lane] = 0.0;
            krange_scatter<4>(Kd, f0, 0);
#pragma unroll
            for (int u = 0; u < CT; ++u) ktile_take<true>(t[u][0], Kd, u);
            krange_scatter<3>(Kd, f1, 0);
#pragma unroll
            for (int u = 1; u < CT; ++u) ktile_take<true>(t[u][1], Kd, u - 1);
            krange_scatter<3>(Kd, f2, 0);
#pragma unroll
            for (int u = 2; u < CT; ++u) ktile_take<true>(t[u][2], Kd, u - 2);
            krange_scatter<2>(Kd, f3, 0);
            ktile_take<false>(t[3][3], Kd, 0);
        }
        st.drain();
        st.mark(0);
        if (r0 > kd) {
            const int step = S.ld * 32;
            int ok = S.at(kd, r0);
            int oy = kd * 8;   // rows kd .. of uf (32 bytes per k-step): y of the earlier panels
            const int ystep = 32;
            auto yload = [&](int off) { return YK.load(off); };   // lane (lq, *) <- y[k + lq]
            double fb[DEPTHN][CT], fy[DEPTHN];
            double ys[CT] = {0.0, 0.0, 0.0, 0.0};  // sum_k L[row][k] y[k], partial over this lane's k = k0 + lq
            auto bload = [&](int off, int c, int k) {  // zero where chunk c is left of its envelope
                return S.load_at(S.lane_off(k >= bks[c]), off + 128 * c);
            };
#pragma unroll
            for (int d = 0; d < DEPTHN - 1; ++d) {
#pragma unroll
                for (int c = 0; c < CT; ++c) fb[d][c] = bload(ok + d * step, c, kd + 4 * d);
                fy[d] = yload(oy + d * ystep);
            }
            for (int k0 = kd; k0 < r0; k0 += 4 * DEPTHN) {
#pragma unroll
                for (int d = 0; d < DEPTHN; ++d) {
                    const int nd = (d + DEPTHN - 1) % DEPTHN;
#pragma unroll
                    for (int c = 0; c < CT; ++c)
                        fb[nd][c] = bload(ok + (d + DEPTHN - 1) * step, c, k0 + 4 * (d + DEPTHN - 1));
                    fy[nd] = yload(oy + (d + DEPTHN - 1) * ystep);
                    // chunk u of the block has non-zeros in these four columns only from bks[u] on
                    // (non-decreasing in u): products with an all-zero operand are not issued
                    const int kk = k0 + 4 * d;
#pragma unroll
                    for (int u = 0; u < CT; ++u)
                        if (kk >= bks[u]) {
#pragma unroll
                            for (int s = 0; s <= u; ++s) t[u][s] = mfma_f64_negA(fb[d][s], fb[d][u], t[u][s]);
                            ys[u] = fma(fb[d][u], fy[d], ys[u]);
                        }
                }
                ok += DEPTHN * step;
                oy += DEPTHN * ystep;
            }
#pragma unroll
            for (int u = 0; u < CT; ++u) {  // sum over the four quarter-waves (k = k0 + lq)
                ys[u] += __shfl_xor(ys[u], 16);
                ys[u] += __shfl_xor(ys[u], 32);
                yr[u] -= ys[u];
            }
        }
        st.mark(1);
        // F: factor the block in registers; inv(L_ss) stays in LDS (operand fragments, read where used)
#pragma unroll
        for (int s = 0; s < CT; ++s) {
            if (bad_col == 0) {
                const Chol16 f = chol16_invert(t[s][s], sc, Wl + s * 256);
                t[s][s] = f.u;
                __builtin_amdgcn_wave_barrier();
                if (f.bad >= 0) bad_col = r0 + 16 * s + f.bad + 1;
            }
            if (bad_col == 0) {
                if (s + 1 < CT) {
                    double wf[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) wf[r] = Wl[s * 256 + wfrag_lane(r, lane)];
#pragma unroll
                    for (int u = s + 1; u < CT; ++u) {
                        d4 x = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int r = 0; r < 4; ++r) x = mfma_f64(wf[r], t[u][s][r], x);
                        t[u][s] = x;
                    }
                }
#pragma unroll
                for (int u = s + 1; u < CT; ++u)
#pragma unroll
                    for (int s2 = s + 1; s2 <= u; ++s2)
#pragma unroll
                        for (int r = 0; r < 4; ++r) t[u][s2] = mfma_f64_negA(t[s2][s][r], t[u][s][r], t[u][s2]);
            }
        }
        st.mark(2);
        if (bad_col != 0) break;
#pragma unroll
        for (int u = 0; u < CT; ++u)
#pragma unroll
            for (int s = 0; s <= u; ++s) tile_store(t[u][s], S, r0 + 16 * s, r0 + 16 * u);
        // the load vector against the factored block: y_s = inv(L_ss) (y_s - sum_{s'<s} L_{s,s'} y_s').
        // Both operands are A-fragments as they stand (lane (lq, li), k-step r <-> entry [row li][col 4 r + lq]):
        // the vector goes into "column order" with four lane permutes, the products are per-lane FMAs and
        // the sum over a row's four quarter-waves two lane reductions.
        const int src0 = (lane & 48) | lq;  // lane holding the entry of row 4 r + lq: src0 + 4 r
#pragma unroll
        for (int s = 0; s < CT; ++s) {
            double yk[4], acc = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) yk[r] = __shfl(yr[s], src0 + 4 * r);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = fma(Wl[s * 256 + wfrag_lane(r, lane)], yk[r], acc);
            acc += __shfl_xor(acc, 16);
            acc += __shfl_xor(acc, 32);
            yr[s] = acc;
            YRS.store((r0 + 16 * s) * 8, acc);  // later panels and trs_potrs read y from uf
            if (s + 1 < CT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) yk[r] = __shfl(acc, src0 + 4 * r);
#pragma unroll
                for (int s2 = s + 1; s2 < CT; ++s2) {
                    double part = 0.0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) part = fma(t[s2][s][r], yk[r], part);
                    part += __shfl_xor(part, 16);
                    part += __shfl_xor(part, 32);
                    yr[s2] -= part;
                }
            }
        }
        st.mark(3);
        // items: the chunks below the block that reach into this panel
        const int lastq = env.last[panel];
        for (int c0 = 4 * panel + CT; c0 <= lastq;) {
            const int ks = 16 * env.ft[c0];
            const int left = lastq - c0 + 1;
            if (RSV >= 4 && left >= 4) {
                narrow_item<(RSV >= 4 ? 4 : 1), FUSED>(S, r0, c0, ks, env.ft, env.cend, Wl, t, K, tb, env.kmask);
                c0 += 4;
            } else if (RSV >= 2 && left >= 2) {
                narrow_item<(RSV >= 2 ? 2 : 1), FUSED>(S, r0, c0, ks, env.ft, env.cend, Wl, t, K, tb, env.kmask);
                c0 += 2;
            } else {
                narrow_item<1, FUSED>(S, r0, c0, ks, env.ft, env.cend, Wl, t, K, tb, env.kmask);
                c0 += 1;
            }
        }
        st.mark(4);
        // (unconditionally: behind the last panel the requests fall outside the descriptors' ranges and return
        // zeros that nobody reads; a condition here would keep the old tiles alive beside the new ones)
        if constexpr (!FUSED) load_block(r0 + TRS_NB);
        // this wave's stores must have landed before its own loads of the next panel's block update
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        st.mark(5);
    }
    if (lane == 0) info[b] = bad_col;
    // The substitution U u = y by the same wave, right behind its factorisation (trs_subst.h): the factor's last
    // panels are still in the caches, the launch of trs_potrs_batched finds nothing left to do for this matrix,
    // and the memory-bound substitution of one wave overlaps the issue-bound factorisation of its neighbours.
    // The solution strip takes the place of the inv(L_ss) fragments (dead now): up to 1024 rows.
    if (substitute != 0 && bad_col == 0 && npad <= CT * 256) {
        trs_subst::narrow_substitute(S.rs, S.ld, npad, env.cend, Wl, uf_all + (size_t)b * ld_uf, ld_uf);
        if (lane == 0) {
            int* meta = const_cast<int*>(env.last) + n_pad_max / 64;
            meta[0] = env.slack | TRS_ENV_SUBSTITUTED;
        }
        st.drain();
        st.mark(6);
    }
    st.flush();
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

extern "C" int trs_potrf_launch(int B, const int* n_free, int ld, size_t slab_stride, int n_pad_max,
                                double* S, int* info, const int* env, const void* work, double* uf,
                                int ld_uf, int compact_possible, int hints, hipStream_t stream) {
    if (B <= 0) return 0;
    // a slab is addressed through one buffer descriptor with 32-bit byte offsets, the upper half of
    // the offset range being the "tile not stored" marker (Slab::gone)
    if (slab_stride * sizeof(double) >= (size_t)1 << 31) return (int)hipErrorInvalidValue;
    if (uf == nullptr || ld_uf < n_pad_max) return (int)hipErrorInvalidValue;
    // the factorising wave of a narrow-envelope matrix substitutes it as well unless the caller keeps the stages apart
    const int fused_substitution = (hints & TRS_HINT_SEPARATE_STAGES) == 0;
    if (env != nullptr) {
        // the wave-per-matrix kernels: each takes the matrices trs_assemble routed to it (csrc/trs_common.h)
        const dim3 grid((B + MPW - 1) / MPW), block(64 * MPW);
        const unsigned char* wk = static_cast<const unsigned char*>(work);
        int rc = 0;
        if (compact_possible) {  // (the compact form is opt-in: no launch while it is switched off)
            hipLaunchKernelGGL((trs_potrf_narrow_kernel<true, 2>), grid, block, 0, stream, S, n_free, ld, slab_stride,
                               info, env, n_pad_max, B, wk, uf, ld_uf, fused_substitution);
            if ((rc = (int)hipGetLastError())) return rc;
        }
        hipLaunchKernelGGL((trs_potrf_narrow_kernel<false, 2>), grid, block, 0, stream, S, n_free, ld, slab_stride,
                           info, env, n_pad_max, B, wk, uf, ld_uf, fused_substitution);
        if ((rc = (int)hipGetLastError())) return rc;
        if (TRS_NARROW_RS4_ABOVE <= TRS_NARROW_MAX_BELOW) {  // (compile-time: see trs_common.h)
            hipLaunchKernelGGL((trs_potrf_narrow_kernel<false, 4>), grid, block, 0, stream, S, n_free, ld, slab_stride,
                               info, env, n_pad_max, B, wk, uf, ld_uf, fused_substitution);
            if ((rc = (int)hipGetLastError())) return rc;
        }
    }
    if (env != nullptr && (hints & TRS_HINT_NO_WIDE) != 0) return 0;  // no matrix for the work-group kernel
    hipLaunchKernelGGL(trs_potrf_kernel, dim3(B), dim3(NW * 64), 0, stream, S, n_free, ld,
                       slab_stride, info, env, n_pad_max, uf, ld_uf);
    return (int)hipGetLastError();
}
