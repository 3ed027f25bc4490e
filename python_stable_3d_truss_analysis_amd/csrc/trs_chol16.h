// 16 x 16 Cholesky + inverse of the factor by one wave on the f64 matrix core - the serial building
// block of every factorisation kernel here (trs_potrf_kernel, trs_potrf_narrow_kernel, trs_solve_small).
#pragma once
#include "trs_common.h"

namespace {

__device__ __forceinline__ double lane_bcast(double v, int src) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// acc -= A * B : the BLGP field of the f64 MFMA is NEG[2:0] (bit 0 negates A; probed on gfx950
// with tools/mfma_neg_test.hip).
__device__ __forceinline__ d4 mfma_f64_negA(double a, double b, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 1);
}

// Cholesky of one symmetric 16x16 tile by the calling wave, plus the inverse of its factor, in
// 4-column blocks so that everything of O(16^2) per step runs on the matrix core:
//   * the 4 columns of block b are gathered row-owner-wise (lane li: c[q] = T[li][4b+q]; one LDS
//     round) and factored with v_readlane broadcasts (compile-time lanes, SGPR results): 4 pivots
//     and 6 multipliers per block instead of 16 and 120 per tile;
//   * the lane-(lq, li) selection p = L[li][4b+lq] of those columns is at once the A and the B
//     fragment of the rank-4 trailing update T -= P P^T (ONE MFMA, NEG-A), and, masked, register b of
//     the result U = L^T in D-form;
//   * inv(L) by block rows: W[b,:] = inv(L_bb) (I - sum_{k<b} L[:,k] W[k,:])[b,:] - one MFMA with the
//     embedded 4x4 inverse (formed from the SGPR multipliers) as A and register b of the running
//     D-form right-hand side as B, and one MFMA (A = p, B = the new W rows) to update the latter.
// VALU work per tile drops from ~1150 to ~450 instructions; 11 dependent MFMAs replace the rest.
//   t       : in  the symmetric tile in D-form (t[r] = T[c = lq + 4 r][i = li]);
//             out U = L^T in D-form on and above the diagonal, the strictly-lower part of inv(L)
//             below it (read by trs_potrs_kernel; no other reader touches that part)
//   sc      : LDS scratch (block gather)
//   wfrag   : receives inv(L) as A-fragments (layout of PanelLds::W[s])
// Returns the tile and the 0-based index of the first non-positive pivot, or -1 (wave-uniform).
// inv(L_ss) as MFMA A-fragments in LDS: element W[t][c] (row t, column c of the 16 x 16 inverse) belongs to
// lane (lq = c & 3, li = t) of k-step r = c >> 2.  Stored SWIZZLED, at r * 64 + lq * 16 + (li ^ c): the writers
// (chol16_invert: 16 lanes of one quarter-wave hold 16 columns c of one row t) and the readers (a wave reads
// k-step r: lanes (lq, li) = 64 consecutive doubles up to the XOR) then both touch every LDS bank once.
// Unswizzled, the 16 writing lanes sit 32 dwords apart: a 16-way bank conflict on every store.
__device__ __forceinline__ int wfrag_index(int t, int c) { return (c >> 2) * 64 + (c & 3) * 16 + (t ^ c); }
// the same for the reading lane: index of this lane's element of k-step r
__device__ __forceinline__ int wfrag_lane(int r, int lane) {
    return r * 64 + (lane & 48) + ((lane & 15) ^ (4 * r + (lane >> 4)));
}

struct ChScratch {
    double G[4][16];  // G[q][row] = T[row][4 b + q] of the running block (conflict-free both ways)
};
struct Chol16 {
    d4 u;
    int bad;
};

// 1/sqrt(d) for a wave-uniform positive d: hardware estimate + one third-order correction
__device__ __forceinline__ double rsqrt_refined(double d) {
    const double y = __builtin_amdgcn_rsq(d);
    const double e = fma(-(d * y), y, 1.0);
    return fma(y * e, fma(0.375, e, 0.5), y);
}

// Not inlined: ONE copy of this long straight-line routine keeps the kernel's code inside the
// instruction cache.  The tile and the pivot status travel by value in registers.
// The scratch and the fragment buffer arrive as LDS-address-space pointers: through generic pointers
// the accesses of this non-inlined routine would be FLAT instructions, which are slower than ds_*
// and also wait on the global-memory counter.
typedef __attribute__((address_space(3))) double lds_f64;

static __device__ __noinline__ Chol16 chol16_invert_lds(d4 t, lds_f64* G, lds_f64* wfrag) {
    const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
    unsigned badmask = 0;
    d4 R, u;  // R: running right-hand side of inv(L) (D-form, starts as the identity); u: result
#pragma unroll
    for (int r = 0; r < 4; ++r) R[r] = (lq + 4 * r == li) ? 1.0 : 0.0;
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    const double dq[4] = {lq == 0 ? 1.0 : 0.0, lq == 1 ? 1.0 : 0.0, lq == 2 ? 1.0 : 0.0, lq == 3 ? 1.0 : 0.0};
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int j0 = 4 * b;
        // row-owner copy of the block's 4 columns (by symmetry row j0+q of the D-form tile)
        G[16 * lq + li] = t[b];
        __builtin_amdgcn_wave_barrier();
        double c[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) c[q] = G[16 * q + li];
        __builtin_amdgcn_wave_barrier();
        double rinv[4], m[4][4];  // wave-uniform: 1 / L[j0+q][j0+q], L[j0+q2][j0+q]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double d = lane_bcast(c[q], j0 + q);
            // a non-positive (or NaN) pivot only sets its bit: what follows it is garbage, the
            // first set bit is the answer
            badmask |= (d > 0.0) ? 0u : 1u << (j0 + q);
            rinv[q] = rsqrt_refined(d);
            c[q] *= rinv[q];  // L[li][j0+q] for li >= j0+q (li == j0+q: d / sqrt(d))
#pragma unroll
            for (int q2 = q + 1; q2 < 4; ++q2) {
                m[q2][q] = lane_bcast(c[q], j0 + q2);
                c[q2] -= c[q] * m[q2][q];
            }
        }
        const double p = lq == 0 ? c[0] : lq == 1 ? c[1] : lq == 2 ? c[2] : c[3];  // L[li][j0+lq]
        if (b < 3) t = mfma_f64_negA(p, p, t);  // rows / columns below j0+4: T -= L[:,blk] L[:,blk]^T
        // column lq of inv(L_bb) by forward substitution on the uniform multipliers: e[k] = M[k][lq]
        // (dq[k] = 1 in quarter-wave k, else 0: the unit right-hand side without selects)
        const double e0 = rinv[0] * dq[0];
        const double e1 = rinv[1] * fma(-m[1][0], e0, dq[1]);
        const double e2 = rinv[2] * fma(-m[2][1], e1, fma(-m[2][0], e0, dq[2]));
        const double e3 = rinv[3] * fma(-m[3][2], e2, fma(-m[3][1], e1, fma(-m[3][0], e0, dq[3])));
        // A operand: M embedded in rows j0 .. j0+3 (row li - j0 = li & 3 there), zero elsewhere
        const double e01 = (li & 1) ? e1 : e0, e23 = (li & 1) ? e3 : e2;
        const double ma = (li >> 2) == b ? ((li & 2) ? e23 : e01) : 0.0;
        const d4 wb = mfma_f64(ma, R[b], zero);  // register b = W[j0+lq][li], the others are zero
        // A-fragment layout of PanelLds::W (swizzled, wfrag_index above): the element
        // (t = j0+lq, c = li) held here lands at 16 li + j0 + lq
        wfrag[wfrag_index(j0 + lq, li)] = wb[b];
        // result tile: U on and above the diagonal; the otherwise unused strictly-lower part carries
        // inv(L) (its diagonal is 1 / diag(U)) for the 16 x 16 steps of the back substitution
        u[b] = (j0 + lq <= li) ? p : wb[b];
        if (b < 3) R = mfma_f64_negA(p, wb[b], R);  // R -= L[:,blk] W[blk,:]
    }
    return Chol16{u, badmask ? __builtin_ctz(badmask) : -1};
}
__device__ __forceinline__ Chol16 chol16_invert(d4 t, ChScratch& sc, double* wfrag) {
    Chol16 out = chol16_invert_lds(t, (lds_f64*)&sc.G[0][0], (lds_f64*)wfrag);
    out.bad = __builtin_amdgcn_readfirstlane(out.bad);
    return out;
}

}  // namespace
