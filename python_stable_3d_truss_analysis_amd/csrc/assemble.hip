// Assembly of the reduced stiffness slab (include/trs_solver.h) for a batch of trusses.
//
// Replaces Member.k / cosines / matK (slientruss3d/truss.py:56-86), Truss.GetKMatrix
// (truss.py:307-316), GetExternalForceVector (truss.py:303-304) and the row/column elimination
// matK[mask,:][:,mask], vecF[mask] (truss.py:343).
//
// Owner-computes: a work-group owns TR consecutive rows of one truss's slab, stages them in LDS
// (zero fill), lets one thread per member (edge-parallel) scatter-add the member's 3x3 blocks
// that fall into those rows, adds the load column and the identity padding, and writes each
// row to HBM exactly once with 16-byte coalesced stores.  No global atomics, no memset pass.
#include "trs_common.h"
#include "../../include/trs_solver.h"

namespace {

template <int TR>
__global__ __launch_bounds__(256) void trs_assemble_kernel(
    const double* __restrict__ xyz, const int* __restrict__ conn, const double* __restrict__ E,
    const double* __restrict__ A, const double* __restrict__ loads,
    const int* __restrict__ free_index, const int* __restrict__ n_free, const int* __restrict__ nJ,
    const int* __restrict__ nM, const int nJ_max, const int nM_max, const int ld,
    const size_t slab_stride, double* __restrict__ S_all, const int flags, const int B) {
    extern __shared__ double T[];  // TR rows, row stride W
    const int nblk = gridDim.x / B;  // row blocks per truss; consecutive blocks share a truss
    const int b = blockIdx.x / nblk, c0 = (blockIdx.x % nblk) * TR, tid = threadIdx.x;
    const int n = n_free[b];
    const int npad = trs_round_up(n, TRS_NB);
    if (c0 >= npad) return;
    const int i_lo = (flags & TRS_ASM_FULL_SYMMETRIC) ? 0 : (c0 & ~15);  // first stored column
    const int W = npad + 16 - i_lo;                                     // multiple of 16

    for (int x = tid * 2; x < TR * W; x += 512) *reinterpret_cast<d2*>(T + x) = d2{0.0, 0.0};
    __syncthreads();

    const int* fi = free_index + (size_t)b * 3 * nJ_max;
    const double* X = xyz + (size_t)b * 3 * nJ_max;
    const int members = nM[b];
    for (int m = tid; m < members; m += 256) {
        const size_t mm = (size_t)b * nM_max + m;
        const int j0 = conn[2 * mm], j1 = conn[2 * mm + 1];
        int f[6];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            f[a] = fi[3 * j0 + a];
            f[3 + a] = fi[3 * j1 + a];
        }
        bool hit = false;
#pragma unroll
        for (int p = 0; p < 6; ++p) hit |= (f[p] >= c0) & (f[p] < c0 + TR);
        if (!hit) continue;
        double d[3], len2 = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            d[a] = X[3 * j1 + a] - X[3 * j0 + a];
            len2 += d[a] * d[a];
        }
        const double len = sqrt(len2);
        const double k = E[mm] * A[mm] / len;  // truss.py:56-58
        double c[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) c[a] = d[a] / len;  // truss.py:60-63
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            if (f[p] < c0 || f[p] >= c0 + TR) continue;
            double* row = T + (size_t)(f[p] - c0) * W - i_lo;
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                if (f[q] < i_lo) continue;  // constrained (-1) or left of the stored part
                const double v = k * (c[p % 3] * c[q % 3]);
                atomicAdd(row + f[q], (p / 3 == q / 3) ? v : -v);  // ds_add_f64
            }
        }
    }
    // load column (truss.py:303-304, vecF[mask])
    const double* F = loads + (size_t)b * 3 * nJ_max;
    for (int dof = tid; dof < 3 * nJ[b]; dof += 256) {
        const int r = fi[dof];
        if (r >= c0 && r < c0 + TR) T[(size_t)(r - c0) * W + npad - i_lo] = F[dof];
    }
    // identity padding n <= c < n_pad
    if (tid < TR && c0 + tid >= n) T[(size_t)tid * W + c0 + tid - i_lo] = 1.0;
    __syncthreads();

    double* S = S_all + (size_t)b * slab_stride;
    for (int r = 0; r < TR; ++r) {
        double* dst = S + (size_t)(c0 + r) * ld + i_lo;
        const double* src = T + (size_t)r * W;
        for (int x = tid * 2; x < W; x += 512)
            *reinterpret_cast<d2*>(dst + x) = *reinterpret_cast<const d2*>(src + x);
    }
}

}  // namespace

extern "C" int trs_assemble_launch(int B, int nJ_max, int nM_max, const double* xyz, const int* conn,
                                   const double* E, const double* A, const double* loads,
                                   const int* free_index, const int* n_free, const int* nJ,
                                   const int* nM, int ld, size_t slab_stride, int n_pad_max,
                                   double* S, int flags, hipStream_t stream) {
    if (B <= 0 || n_pad_max <= 0) return 0;
    // rows per work-group: as many as keep the LDS tile under ~48 KiB (3 work-groups per CU)
    const size_t row_bytes = (size_t)(n_pad_max + 16) * sizeof(double);
    if (8 * row_bytes <= 49152) {
        hipLaunchKernelGGL(trs_assemble_kernel<8>, dim3((unsigned)(n_pad_max / 8) * B), dim3(256), 8 * row_bytes,
                           stream, xyz, conn, E, A, loads, free_index, n_free, nJ, nM, nJ_max,
                           nM_max, ld, slab_stride, S, flags, B);
    } else if (4 * row_bytes <= 65536) {
        hipLaunchKernelGGL(trs_assemble_kernel<4>, dim3((unsigned)(n_pad_max / 4) * B), dim3(256), 4 * row_bytes,
                           stream, xyz, conn, E, A, loads, free_index, n_free, nJ, nM, nJ_max,
                           nM_max, ld, slab_stride, S, flags, B);
    } else {
        if (row_bytes > 160 * 1024) return (int)hipErrorInvalidValue;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(trs_assemble_kernel<1>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)row_bytes);
        hipLaunchKernelGGL(trs_assemble_kernel<1>, dim3((unsigned)n_pad_max * B), dim3(256), row_bytes, stream,
                           xyz, conn, E, A, loads, free_index, n_free, nJ, nM, nJ_max, nM_max, ld,
                           slab_stride, S, flags, B);
    }
    return (int)hipGetLastError();
}
