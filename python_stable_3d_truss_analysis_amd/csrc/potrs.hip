// Back substitution U u = y for a batch of factored slabs (U = L^T in the upper part of S, y in
// the column n_pad; see include/trs_solver.h).  Replaces the solve half of np.linalg.solve
// (slientruss3d/truss.py:343).  HBM-bound: U is streamed once, row by row (contiguous rows).
//
// One work-group per truss.  Blocks of 64 rows from the bottom up: all four waves form
// t = y - U[rows, solved columns] . u for the block (coalesced row reads, wave reductions) and
// stage the 64 x 64 diagonal block in LDS; wave 0 then solves the triangle.
#include "trs_common.h"

namespace {

constexpr int BS = 64;  // rows per block

__global__ __launch_bounds__(256) void trs_potrs_kernel(const double* __restrict__ S_all,
                                                        const int* __restrict__ n_free, const int ld,
                                                        const size_t slab_stride,
                                                        double* __restrict__ uf, const int ld_uf) {
    extern __shared__ double sh[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int npad = trs_round_up(n_free[b], TRS_NB);
    if (npad == 0) return;
    double* us = sh;                // [npad] solution
    double* Ub = sh + npad;         // [64][65] diagonal block
    double* tb = Ub + BS * (BS + 1);  // [64] right-hand side of the block
    const double* S = S_all + (size_t)b * slab_stride;

    for (int cb = npad - BS; cb >= 0; cb -= BS) {
        // t[c] = y[c] - sum_{i >= cb+64} U[c][i] u[i]; 16 rows per wave
        for (int rr = 0; rr < 16; ++rr) {
            const int c = cb + 16 * wave + rr;
            const double* row = S + (size_t)c * ld;
            double sum = 0.0;
            for (int i = cb + BS + lane; i < npad; i += 64) sum += row[i] * us[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
            if (lane == 0) tb[16 * wave + rr] = row[npad] - sum;
            // diagonal block row (columns cb .. cb+63)
            Ub[(16 * wave + rr) * (BS + 1) + lane] = row[cb + lane];
        }
        __syncthreads();
        if (wave == 0) {
            double tc = tb[lane], mine = 0.0;
            for (int cc = BS - 1; cc >= 0; --cc) {
                const double ucc = __shfl(tc, cc) / Ub[cc * (BS + 1) + cc];
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

}  // namespace

extern "C" int trs_potrs_launch(int B, const int* n_free, int ld, size_t slab_stride, int n_pad_max,
                                const double* S, double* uf, int ld_uf, hipStream_t stream) {
    if (B <= 0 || n_pad_max <= 0) return 0;
    const size_t lds = (size_t)(n_pad_max + BS * (BS + 1) + BS) * sizeof(double);
    if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(trs_potrs_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(trs_potrs_kernel, dim3(B), dim3(256), lds, stream, S, n_free, ld, slab_stride,
                       uf, ld_uf);
    return (int)hipGetLastError();
}
