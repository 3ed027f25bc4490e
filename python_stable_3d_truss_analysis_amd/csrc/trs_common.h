// Shared device helpers for the gfx950 truss-solver kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

#define TRS_NB 64           // panel width of the factorisation = padding quantum of n
#define TRS_TILE 16         // MFMA tile edge (v_mfma_f64_16x16x4_f64)
#define TRS_WAVE 64

__host__ __device__ static inline int trs_round_up(int v, int q) { return (v + q - 1) / q * q; }

// One v_mfma_f64_16x16x4_f64: D(16x16) = A(16x4) * B(4x16) + C.
// Lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15] and holds D[(l >> 4) + 4*r][l & 15]
// in component r of the accumulator (cdna_hip_programming.md section 3, f64 map).
__device__ static inline d4 mfma_f64(double a, double b, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// Broadcast of lane `src` (compile-time constant) of a double through v_readlane_b32.
template <int SRC>
__device__ static inline double readlane_f64(double v) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), SRC);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), SRC);
    return __hiloint2double(hi, lo);
}
