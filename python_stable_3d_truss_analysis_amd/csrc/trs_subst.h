// trs_subst.h - back substitution U u = y of ONE narrow-envelope matrix by ONE wave (shared by
// trs_potrs_narrow_kernel and, fused behind the factorisation, by trs_potrf_narrow_kernel).
//   chunk s (16 rows, from the bottom):
//   t_s = y_s - sum_{q > s} U[s, q] u_q      tiles (s, q) as D-form registers (lane li = column), the
//                                            products accumulated per lane, ONE 16-lane reduction per chunk
//   u_s = inv(L_ss)^T t_s                    with the inverse trs_potrf left below the diagonal of the tile
// The solution lives in the wave's LDS strip `us` (it starts as y, read from `ub`, and is written back there).
#pragma once
#include <hip/hip_runtime.h>

#ifndef TRS_POTRS_PTG
#define TRS_POTRS_PTG 4
#endif

namespace trs_subst {
typedef double sd4 __attribute__((ext_vector_type(4)));
constexpr int PTG = TRS_POTRS_PTG;  // tiles in flight per group of loads

__device__ __forceinline__ void narrow_substitute(const __amdgpu_buffer_rsrc_t rs, const int ld, const int npad,
                                                  const int* __restrict__ cend, double* us, double* ub,
                                                  const int ld_uf) {
    const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
    // y = L^-1 f into the wave's LDS strip, eight requests in flight (one at a time, each waited for, is a
    // serial chain of eleven memory latencies at the start of every wave of the launch)
    for (int c0 = 0; c0 < npad; c0 += 8 * 64) {
        double yv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = c0 + 64 * i + lane;
            yv[i] = c < npad ? ub[c] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = c0 + 64 * i + lane;
            if (c < npad) us[c] = yv[i];
        }
    }
    __builtin_amdgcn_wave_barrier();
    const unsigned loff = ((unsigned)lq * (unsigned)ld + (unsigned)li) * 8u;
    const int rstep = ld * 32;  // four slab rows, bytes
    auto tile = [&](sd4& a, int c0, int i0, bool exists) {  // D-form tile; outside the envelope: zeros, no traffic
        const unsigned vo = exists ? loff : 0x80000000u;
        const int o = (c0 * ld + i0) * 8;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            a[r] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, vo, o + r * rstep, 0));
    };
    // does this lane hold the diagonal entry of its column (row lq + 4 r == li), and in which register
    const bool has_diag = li >= lq && ((li - lq) & 3) == 0;
    const int rdiag = (li - lq) >> 2;
    // The tiles of a chunk do not depend on the solution: the diagonal tile and the first PTG off-diagonal
    // tiles of chunk s - 1 are requested BEFORE chunk s is worked on.  With every matrix of a launch resident
    // (trs_potrs_narrow_kernel) the memory system is the limit either way; behind the factorisation, where the
    // last matrices of a SIMD run alone, it takes the memory latency out of the chain of 44 chunks.
    sd4 dg_n, a_n[PTG];
    auto fetch = [&](int s) {
        const int ce = cend[s];
        tile(dg_n, 16 * s, 16 * s, true);
#pragma unroll
        for (int g = 0; g < PTG; ++g) tile(a_n[g], 16 * s, 16 * (s + 1 + g), s + 1 + g < ce);
    };
    fetch(npad / 16 - 1);
    for (int s = npad / 16 - 1; s >= 0; --s) {
        const int ce = cend[s];
        sd4 dg = dg_n, a[PTG];
#pragma unroll
        for (int g = 0; g < PTG; ++g) a[g] = a_n[g];
        if (s > 0) fetch(s - 1);
        sd4 part = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int g = 0; g < PTG; ++g) {
            const double uq = s + 1 + g < ce ? us[16 * (s + 1 + g) + li] : 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) part[r] += a[g][r] * uq;
        }
        for (int q0 = s + 1 + PTG; q0 < ce; q0 += PTG) {  // envelopes wider than PTG tiles: the rest, not ahead
#pragma unroll
            for (int g = 0; g < PTG; ++g) tile(a[g], 16 * s, 16 * (q0 + g), q0 + g < ce);
#pragma unroll
            for (int g = 0; g < PTG; ++g) {
                const double uq = q0 + g < ce ? us[16 * (q0 + g) + li] : 0.0;
#pragma unroll
                for (int r = 0; r < 4; ++r) part[r] += a[g][r] * uq;
            }
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[r] += __shfl_xor(part[r], off);
        // t[c] for the rows c = lq + 4 r; u_s[li] = t[li] / U[li][li] + sum_{c > li} inv(L)[c][li] t[c]
        double val = 0.0;
        double ddiag = 1.0, tdiag = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double t = us[16 * s + lq + 4 * r] - part[r];
            val += (lq + 4 * r > li ? dg[r] : 0.0) * t;
            if (has_diag && rdiag == r) {
                ddiag = dg[r];
                tdiag = t;
            }
        }
        val += tdiag / ddiag;  // (lanes without the diagonal entry add 0 / 1)
        val += __shfl_xor(val, 16);
        val += __shfl_xor(val, 32);
        __builtin_amdgcn_wave_barrier();
        if (lq == 0) us[16 * s + li] = val;
        __builtin_amdgcn_wave_barrier();
    }
    for (int c = lane; c < npad && c < ld_uf; c += 64) ub[c] = us[c];
}
}  // namespace trs_subst
