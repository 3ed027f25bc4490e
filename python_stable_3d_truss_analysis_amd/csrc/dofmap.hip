// Free-DOF numbering: the batched form of Truss.GetDisplacementUnknownMask
// (slientruss3d/truss.py:319-326) plus the compaction that the reference gets from boolean-mask
// indexing (truss.py:343).  One wave per truss: ballot + popcount prefix over the DOFs.
#include "trs_common.h"

namespace {

__global__ __launch_bounds__(64) void trs_dofmap_kernel(const uint8_t* __restrict__ cbits,
                                                        const int* __restrict__ nJ, const int nJ_max,
                                                        int* __restrict__ free_index,
                                                        int* __restrict__ n_free) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int ndof = 3 * nJ[b], ndof_max = 3 * nJ_max;
    const uint8_t* cb = cbits + (size_t)b * nJ_max;
    int* out = free_index + (size_t)b * ndof_max;
    int base = 0;
    for (int d0 = 0; d0 < ndof_max; d0 += 64) {
        const int d = d0 + lane;
        bool is_free = false;
        if (d < ndof) is_free = ((cb[d / 3] >> (d % 3)) & 1) == 0;
        const unsigned long long mask = __ballot(is_free);
        const int idx = base + __popcll(mask & ((1ull << lane) - 1ull));
        if (d < ndof_max) out[d] = is_free ? idx : -1;
        base += __popcll(mask);
    }
    if (lane == 0) n_free[b] = base;
}

}  // namespace

extern "C" int trs_dofmap_launch(int B, int nJ_max, const uint8_t* cbits, const int* nJ,
                                 int* free_index, int* n_free, hipStream_t stream) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(trs_dofmap_kernel, dim3(B), dim3(64), 0, stream, cbits, nJ, nJ_max,
                       free_index, n_free);
    return (int)hipGetLastError();
}
