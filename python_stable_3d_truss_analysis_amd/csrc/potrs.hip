// Back substitution U u = y for a batch of factored slabs (U = L^T in the upper part of S; y in uf,
// which the solution overwrites; see include/trs_solver.h).  Replaces the solve half of np.linalg.solve
// (slientruss3d/truss.py:343).  HBM-bound: U is streamed once, row by row (contiguous rows).
//
// One work-group per truss.  Blocks of 64 rows from the bottom up: all four waves form
// t = y - U[rows, solved columns] . u for the block (coalesced loads, 16-lane reductions) and stage
// the 64 x 64 diagonal block in LDS; wave 0 then solves the triangle.  The loads of a block do not
// depend on the solution, only the multiplications do: they are issued one block ahead, so the
// serial chain over the blocks sees the triangle solves and not the memory latencies.
#include "../../include/trs_solver.h"
#include "trs_common.h"
#include "trs_subst.h"

namespace {

constexpr int BS = 64;  // rows per block
#ifndef TRS_POTRS_PF
#define TRS_POTRS_PF 1
#endif
#ifndef TRS_POTRS_WAVES_PER_SIMD
#define TRS_POTRS_WAVES_PER_SIMD 4
#endif
constexpr int PF = TRS_POTRS_PF;  // 64-column chunks of the off-diagonal part requested a block ahead

// broadcast of lane `src` (wave-uniform, not necessarily a compile-time constant)
__device__ __forceinline__ double lane_bcast_dyn(double v, int src) {
    const int s = __builtin_amdgcn_readfirstlane(src);
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), s);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), s);
    return __hiloint2double(hi, lo);
}

// TILESTEP selects the form of the 64 x 64 triangle solve at compile time (one form per
// instantiation keeps the kernel inside 128 VGPRs): true for the envelope mode, false for dense.
template <bool TILESTEP>
__global__ __launch_bounds__(256, TRS_POTRS_WAVES_PER_SIMD) void trs_potrs_kernel(const double* __restrict__ S_all,
                                                        const int* __restrict__ n_free, const int ld,
                                                        const size_t slab_stride,
                                                        double* __restrict__ uf, const int ld_uf,
                                                        const int* __restrict__ env_all,
                                                        const int n_pad_max, const int skip_narrow) {
    extern __shared__ double sh[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int npad = trs_round_up(n_free[b], TRS_NB);
    if (npad == 0) return;
    // narrow envelopes are substituted by trs_potrs_narrow_kernel (one wave per matrix)
    if (skip_narrow && env_all != nullptr && trs_env_is_narrow(trs_env_of(env_all, b, n_pad_max))) return;
    double* us = sh;                  // [npad] solution
    double* Ub = sh + npad;           // [64][65] diagonal block
    double* tb = Ub + BS * (BS + 1);  // [64] right-hand side of the block
    const double* S = S_all + (size_t)b * slab_stride;
    const double* yb = uf + (size_t)b * ld_uf;  // y = L^-1 f, left there by trs_potrf_batched
    // lane (g, l): rows 4 g .. 4 g + 3 of the wave's 16 rows, columns l + 16 k of a 64-column chunk.
    // Every load instruction covers four rows x 128 contiguous bytes; 16 independent loads per chunk.
    const int g = lane >> 4, l = lane & 15;
    const int wrow = 16 * wave + 4 * g;  // first of this lane's four rows inside the block

    const int nch = npad / 16;
    const TrsEnv env = env_all != nullptr ? trs_env_of(env_all, b, n_pad_max)
                                           : TrsEnv{nullptr, nullptr, nullptr, 0};
    // columns beyond the stored extent of a wave's 16 rows are exact zeros of U (and were never written)
    auto block_col_end = [&](int cb) { return env_all != nullptr ? 16 * env.cend[cb / 16 + wave] : npad; };

    // Everything a block needs from HBM is requested ONE BLOCK AHEAD, before the previous block's
    // triangle solve: its diagonal 64 x 64 part, its load-column entries and the first PF 64-column
    // chunks of its off-diagonal part (held in registers), so that a block exposes at most one memory
    // latency and narrow envelopes none beyond the first.
    double dv[16], ov[PF][16], yv[4];
    auto request = [&](int cb) {
        const double* rows = S + (size_t)(cb + wrow) * ld;
        const int col_end = block_col_end(cb);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int k = 0; k < 4; ++k)  // tiles left of the wave's diagonal tile: never written, never used
                if (k >= wave) dv[4 * q + k] = rows[(size_t)q * ld + cb + 16 * k + l];
            yv[q] = yb[cb + wrow + q];
        }
#pragma unroll
        for (int p = 0; p < PF; ++p)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c0 = cb + BS * (p + 1) + 16 * k;
                if (c0 < col_end) {  // uniform: the envelope ends on a 16-column boundary
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        ov[p][4 * q + k] = rows[(size_t)q * ld + c0 + l];
                }
            }
    };
    request(npad - BS);
    for (int cb = npad - BS; cb >= 0; cb -= BS) {
        const double* rows = S + (size_t)(cb + wrow) * ld;
        const int col_end = block_col_end(cb);
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int p = 0; p < PF; ++p)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c0 = cb + BS * (p + 1) + 16 * k;
                if (c0 < col_end) {
                    const double ui = us[c0 + l];
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] += ov[p][4 * q + k] * ui;
                }
            }
        for (int i0 = cb + BS * (PF + 1); i0 < col_end; i0 += BS) {  // wide envelopes: streamed
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (i0 + 16 * k < col_end) {
                    const int col = i0 + 16 * k + l;
                    const double ui = us[col];
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] += rows[(size_t)q * ld + col] * ui;
                }
            }
        }
        // the diagonal block rows (columns cb .. cb+63) go to LDS for the triangle solve
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k >= wave) Ub[(wrow + q) * (BS + 1) + 16 * k + l] = dv[4 * q + k];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double v = acc[q];
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 16);
            if (l == 0) tb[wrow + q] = yv[q] - v;  // y - U[c, solved] u
        }
        if (cb >= BS) request(cb - BS);  // in flight during the triangle solve
        __syncthreads();
        if (wave == 0 && TILESTEP) {
            // Back substitution inside the 64 x 64 triangle in four 16 x 16 steps.  trs_potrf left the
            // strictly-lower part of inv(L_ss) below the diagonal of every diagonal tile, so a step is
            // u_s = inv(L_ss)^T t_s (a 16 x 16 product spread over the four quarter-waves) followed by
            // t -= U[:, s] u_s for the rows above (lane = row, u_s broadcast with v_readlane).
            double tr = tb[lane];  // running right-hand side of row `lane`
            const double rdiag = 1.0 / Ub[lane * (BS + 1) + lane];
#pragma unroll
            for (int s = 3; s >= 0; --s) {
                if ((lane >> 4) == s) tb[lane] = tr;  // t_s, final
                __builtin_amdgcn_wave_barrier();
                // lane (g, l): sum over j = 4 g .. 4 g + 3 of inv(L_ss)[j][l] t_s[j]
                double part = 0.0;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int j = 4 * g + jj;
                    const double tj = tb[16 * s + j];
                    const double w = Ub[(16 * s + j) * (BS + 1) + 16 * s + l];
                    part += (j > l ? w : 0.0) * tj;
                }
                part += __shfl_xor(part, 16);
                part += __shfl_xor(part, 32);
                // diagonal term 1 / U[l][l] t_s[l]: the value sits in lane 16 s + l
                const double dterm = __shfl(tr * rdiag, 16 * s + l);
                const double usl = part + dterm;  // u_s[l], the same in all four quarter-waves
                if (g == 0) us[cb + 16 * s + l] = usl;
                if (s > 0) {
#pragma unroll
                    for (int l2 = 0; l2 < 16; ++l2) {
                        const double ul = lane_bcast_dyn(usl, l2);
                        if (lane < 16 * s) tr -= Ub[lane * (BS + 1) + 16 * s + l2] * ul;
                    }
                }
            }
        } else if (wave == 0) {
            // Dense mode: the plain 64-step substitution - lane c owns the running right-hand side of
            // row c, the pivot row's value is broadcast with v_readlane.  Next to three other
            // work-groups streaming U through the same CU its VALU chain hides completely, which the
            // LDS-heavier tile-step form above does not (measured 1.42 vs 1.78 ms per 4096).
            double tc = tb[lane], mine = 0.0;
            const double rinv = 1.0 / Ub[lane * (BS + 1) + lane];
#pragma unroll 8
            for (int cc = BS - 1; cc >= 0; --cc) {
                const double ucc = lane_bcast_dyn(tc, cc) * lane_bcast_dyn(rinv, cc);
                if (lane == cc) mine = ucc;
                if (lane < cc) tc -= Ub[lane * (BS + 1) + cc] * ucc;
            }
            us[cb + lane] = mine;
        }
        __syncthreads();
    }
    double* out = uf + (size_t)b * ld_uf;
    for (int c = tid; c < npad && c < ld_uf; c += 256) out[c] = us[c];
}

// ======================================================================================================
// Narrow envelopes: one WAVE per matrix, four matrices per work-group, no work-group barriers.
// With an envelope of a few tiles per 16-row chunk the work-group kernel above spends its time in
// barriers and in the serial triangle solve of one wave while three wait; here 12 matrices per CU run
// independently, each streaming its own tiles: chunk s (16 rows, from the bottom) is
//   t_s = y_s - sum_{q > s} U[s, q] u_q      tiles (s, q) as D-form registers (lane li = column), the
//                                            products accumulated per lane, ONE 16-lane reduction per chunk
//   u_s = inv(L_ss)^T t_s                    with the inverse trs_potrf left below the diagonal of the tile
// The solution lives in the wave's LDS strip (it starts as y).
// ======================================================================================================
constexpr int PMW = 4;    // matrices (waves) per work-group
__global__ __launch_bounds__(64 * PMW, 3) void trs_potrs_narrow_kernel(
    const double* __restrict__ S_all, const int* __restrict__ n_free, const int ld, const size_t slab_stride,
    double* __restrict__ uf, const int ld_uf, const int* __restrict__ env_all, const int n_pad_max, const int B) {
    extern __shared__ double sh[];  // [PMW][n_pad_max]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x * PMW + wave;
    if (b >= B) return;
    const int npad = trs_round_up(n_free[b], TRS_NB);
    if (npad == 0) return;
    const TrsEnv env = trs_env_of(env_all, b, n_pad_max);
    if (!trs_env_is_narrow(env)) return;       // trs_potrs_kernel's matrix
    if (trs_env_is_substituted(env)) return;   // the factorisation has substituted it already
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double*>(S_all) + (size_t)b * slab_stride, 0, (int)(slab_stride * sizeof(double)), 0x00020000);
    trs_subst::narrow_substitute(rs, ld, npad, env.cend, sh + (size_t)wave * n_pad_max, uf + (size_t)b * ld_uf, ld_uf);
}

}  // namespace

extern "C" int trs_potrs_launch(int B, const int* n_free, int ld, size_t slab_stride, int n_pad_max,
                                const double* S, double* uf, int ld_uf, const int* env, int hints,
                                hipStream_t stream) {
    if (B <= 0 || n_pad_max <= 0) return 0;
    const size_t lds = (size_t)(n_pad_max + BS * (BS + 1) + BS) * sizeof(double);
    if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
    static const int lds_limit_set =   // once per process, not per launch
        (int)hipFuncSetAttribute(reinterpret_cast<const void*>(trs_potrs_kernel<true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) |
        (int)hipFuncSetAttribute(reinterpret_cast<const void*>(trs_potrs_kernel<false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) |
        (int)hipFuncSetAttribute(reinterpret_cast<const void*>(trs_potrs_narrow_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)lds_limit_set;
    // narrow envelopes: one wave per matrix, while the solution strips of four matrices fit 64 KB of LDS
    // (and the slab fits the 31-bit offsets of a buffer descriptor, as in trs_potrf_batched)
    const size_t lds_narrow = (size_t)PMW * n_pad_max * sizeof(double);
    const int narrow = env != nullptr && lds_narrow <= 64 * 1024 && slab_stride * sizeof(double) < ((size_t)1 << 31);
    if (narrow && (hints & TRS_HINT_SUBSTITUTED) != 0 && (hints & TRS_HINT_NO_WIDE) != 0 && n_pad_max <= 1024)
        return 0;  // every matrix was substituted by the wave that factored it
    if (narrow) {
        hipLaunchKernelGGL(trs_potrs_narrow_kernel, dim3((B + PMW - 1) / PMW), dim3(64 * PMW), lds_narrow, stream, S,
                           n_free, ld, slab_stride, uf, ld_uf, env, n_pad_max, B);
        const int rc = (int)hipGetLastError();
        if (rc) return rc;
    }
    if (narrow && (hints & TRS_HINT_NO_WIDE) != 0) return 0;  // no matrix for the work-group kernel
    if (env != nullptr)
        hipLaunchKernelGGL(trs_potrs_kernel<true>, dim3(B), dim3(256), lds, stream, S, n_free, ld,
                           slab_stride, uf, ld_uf, env, n_pad_max, narrow);
    else
        hipLaunchKernelGGL(trs_potrs_kernel<false>, dim3(B), dim3(256), lds, stream, S, n_free, ld,
                           slab_stride, uf, ld_uf, env, n_pad_max, 0);
    return (int)hipGetLastError();
}
