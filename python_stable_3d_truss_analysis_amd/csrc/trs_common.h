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

// ---- member description of a batch (ABI 10, include/trs_solver.h "Member forms") ------------------------------------
// General form: end joints as int32 pairs, E and A (and rho where needed) as one double per member.
// Table form  : end joints as uint16 pairs (the library's size limit is 65 535 joints anyway), one uint8 type index per
//               member and a table of (a, e, density) triples - 5 instead of 24 (32 with rho) bytes per member in HBM
//               and over PCIe.  E A is formed as e * a from the table: the same product, the same bits.
struct TrsMembers {
    const void* conn;             // [B][nM_max][2]
    const double* E;              // [B][nM_max]   general form
    const double* A;              // [B][nM_max]   general form
    const double* rho;            // [B][nM_max]   general form, where densities are needed (else null)
    const unsigned char* tidx;    // [B][nM_max]   table form; null = general form
    const double* types;          // [n_types][3] = a, e, density
    __device__ __forceinline__ bool table() const { return tidx != nullptr; }
    __device__ __forceinline__ int2 ends(size_t mm) const {
        if (tidx != nullptr) {
            const ushort2 c = reinterpret_cast<const ushort2*>(conn)[mm];
            return int2{(int)c.x, (int)c.y};
        }
        return reinterpret_cast<const int2*>(conn)[mm];
    }
    __device__ __forceinline__ double EA(size_t mm) const {
        if (tidx != nullptr) {
            const double* t = types + 3 * (int)tidx[mm];
            return t[1] * t[0];
        }
        return E[mm] * A[mm];
    }
    __device__ __forceinline__ double area(size_t mm) const { return tidx != nullptr ? types[3 * (int)tidx[mm]] : A[mm]; }
    __device__ __forceinline__ double density(size_t mm) const { return tidx != nullptr ? types[3 * (int)tidx[mm] + 2] : rho[mm]; }
};
static inline TrsMembers trs_members_general(const int* conn, const double* E, const double* A, const double* rho = nullptr) {
    return TrsMembers{conn, E, A, rho, nullptr, nullptr};
}
static inline TrsMembers trs_members_table(const unsigned short* conn16, const unsigned char* tidx, const double* types) {
    return TrsMembers{conn16, nullptr, nullptr, nullptr, tidx, types};
}

// ---- envelope (profile) metadata of the reduced stiffness matrix, per truss ----------------------
// Written by trs_assemble, read by trs_potrf_batched / trs_potrs_batched (optional: a null pointer
// means "treat the matrix as dense").  All quantities are in units of 16-row chunks / 16-column
// tiles of the padded system (nch = n_pad / 16):
//   ft[q]   q < nch : first tile of row chunk q that can hold a non-zero of L (row envelope of the
//                     lower triangle), made non-decreasing in q (running minimum from the end), so
//                     that the set of chunks reaching into a column tile is a contiguous range;
//   last[j] j < nch/4: last row chunk q with ft[q] <= 4 j + 3, i.e. the last chunk with any non-zero
//                     in the 64 columns of panel j (>= 4 j + 3);
//   slack           : which factorisation kernel takes the matrix (trs_env_is_narrow) and, for the
//                     work-group kernel, how many chunks past last[j] its items may overhang;
//   cend[t] t < nch : the STORED extent of slab rows 16 t .. 16 t + 15 (column tile t of L): tiles
//                     t .. cend[t]-1 plus the load column are written by trs_assemble, factored and
//                     read back; nothing else of those rows is ever touched.
//                       wave-per-matrix kernel: cend[t] = max(lastc[t] + 1, end of t's diagonal block),
//                         lastc[t] = last chunk q with ft[q] <= t - the exact envelope at tile
//                         granularity (tiles outside are skipped with out-of-range buffer offsets);
//                       work-group kernel:      cend[t] = last[t / 4] + 1 + slack (rectangular per panel).
//   kmask[t] t < nch: (wave-per-matrix kernels only) bit d set <=> the STIFFNESS matrix has an entry in tile
//                     (slab rows 16 t .. 16 t + 15, columns of chunk t + d), d < 32.  About 30 % of the tiles inside
//                     the envelope of a cube truss hold no entry of K_ff (they fill in during the factorisation):
//                     trs_assemble does not write them and the factorisation does not read them (it takes zeros
//                     through the out-of-range lane offset); the factor is written to every tile of the envelope
//                     as before.  All ones for a matrix of the work-group kernel (every stored tile is written).
// Cholesky fill stays inside the row envelope, so tiles outside it are exact zeros.
#ifndef TRS_NARROW_MAX_BELOW
#define TRS_NARROW_MAX_BELOW 24  // widest reach below a diagonal block (chunks) for the narrow kernel;
                                 // measured on 65 536 mixed cube trusses (tools/bench_configs.py):
                                 // 12 -> 363 K / 772 K solves/s (generator / RCM order), 16 -> 362 / 827,
                                 // 24 -> 340 / 892, 32 -> 313 / 908, 64 -> 271 / 894
#endif
#define TRS_NARROW_ITEM 2        // chunks per item of the narrow kernel
#define TRS_WIDE_ITEM 4          // chunks per item of the work-group kernel
struct TrsEnv {
    const int* ft;
    const int* last;
    const int* cend;
    int slack;
    const int* kmask;
};
// ints per truss: ft[nch_max] | last[nch_max / 4] | slack + 7 reserved | cend[nch_max] | kmask[nch_max]
__host__ __device__ static inline int trs_env_stride(int n_pad_max) { return 3 * (n_pad_max / 16) + n_pad_max / 64 + 8; }
__host__ __device__ static inline int trs_env_cend_offset(int n_pad_max) { return n_pad_max / 16 + n_pad_max / 64 + 8; }
__host__ __device__ static inline int trs_env_kmask_offset(int n_pad_max) { return 2 * (n_pad_max / 16) + n_pad_max / 64 + 8; }
__host__ __device__ static inline TrsEnv trs_env_of(const int* env, int b, int n_pad_max) {
    const int* base = env + (size_t)b * trs_env_stride(n_pad_max);
    return TrsEnv{base, base + n_pad_max / 16, base + trs_env_cend_offset(n_pad_max),
                  base[n_pad_max / 16 + n_pad_max / 64], base + trs_env_kmask_offset(n_pad_max)};
}
// Routing code in `slack`: low byte = chunks an item may overhang (TRS_NARROW_ITEM - 1: the matrix goes
// to a wave-per-matrix kernel, TRS_WIDE_ITEM - 1: to the work-group kernel); bit 8 set = the stiffness
// matrix was NOT written to the slab but as compact per-tile entry lists (below) for the fused
// wave-per-matrix kernel, which forms the tiles where it consumes them.
#define TRS_ENV_COMPACT 0x100
// bit 9 set = a narrow envelope that still reaches more than TRS_NARROW_RS4_ABOVE chunks below a diagonal
// block: the wave-per-matrix kernel instance with four-chunk items takes it (fewer block-side fragment
// loads per item, two waves per SIMD).  OFF by default (threshold beyond any narrow envelope, the second
// instance is then not launched): with every matrix on four-chunk items 65 536 mixed cube trusses gain 3-4 %
// and bar-942 loses 3 %, but routing PART of a batch to a second kernel costs more than it gains (two
// launches that each fill the chip only partly: threshold 12: -8 %, threshold 5: +1 %).
#define TRS_ENV_RS4 0x200
// bit 10, set by trs_potrf_batched (not by trs_assemble, which resets the word): the wave that factored this
// narrow-envelope matrix went on to substitute it (uf holds u already); trs_potrs_batched skips it.
#define TRS_ENV_SUBSTITUTED 0x400
#ifndef TRS_NARROW_RS4_ABOVE
#define TRS_NARROW_RS4_ABOVE 1000000
#endif
__host__ __device__ static inline bool trs_env_is_narrow(const TrsEnv& e) { return (e.slack & 0xff) == TRS_NARROW_ITEM - 1; }
__host__ __device__ static inline bool trs_env_is_compact(const TrsEnv& e) { return (e.slack & TRS_ENV_COMPACT) != 0; }
__host__ __device__ static inline bool trs_env_is_rs4(const TrsEnv& e) { return (e.slack & TRS_ENV_RS4) != 0; }
__host__ __device__ static inline bool trs_env_is_substituted(const TrsEnv& e) { return (e.slack & TRS_ENV_SUBSTITUTED) != 0; }

// ---- compact stiffness matrix of a narrow-envelope truss (per truss, in the assembly workspace) --------
// K_ff as per-TILE entry lists instead of slab tiles: the factorisation reads ~10 bytes per non-zero
// instead of 2 KB per 16 x 16 tile, and the stiffness matrix never exists in HBM in dense form.
//   tile (t, q), t <= q < cend[t]  = slab rows 16 t .. 16 t + 15 (columns of L), matrix rows 16 q .. 16 q + 15
//   tile id                        = tbase[t] + (q - t)
//   tdesc[id]                      = (first entry, number of entries)
//   entry e                        = value eval[e] at epos[e] = (q - t) << 8 | slot, slot = r * 64 + lq * 16 + li
//                                    the D-form position inside the tile: slab row 16 t + lq + 4 r, column
//                                    16 q + li (the register layout tile_load() produces; potrf.hip).  The
//                                    tiles of a chunk are consecutive, so several of them scatter into one
//                                    LDS image with one subtraction.
// At most TRS_NARROW_MAX_BELOW + 4 tiles per chunk (the narrow condition); entries: every member gives
// one 3 x 3 block in the upper part (9) and possibly its mirror inside a diagonal tile (9), every joint
// one diagonal block (9), plus the identity padding.
struct TrsCompactLayout {
    size_t tdesc, tbase, epos, eval, total;  // byte offsets inside the truss's compact region
    int ntile_cap, ecap;
};
__host__ __device__ static inline TrsCompactLayout trs_compact_layout(int nJ_max, int nM_max, int n_pad_max) {
    TrsCompactLayout l;
    const int nch = n_pad_max / 16;
    l.ntile_cap = nch * (TRS_NARROW_MAX_BELOW + 4);
    l.ecap = 18 * nM_max + 9 * nJ_max + 64;
    l.tdesc = 0;                                                    // int2[ntile_cap]
    l.tbase = l.tdesc + (size_t)l.ntile_cap * 8;                    // int[nch + 1]
    l.epos = (l.tbase + (size_t)(nch + 1) * 4 + 15) / 16 * 16;      // unsigned short[ecap]
    l.eval = (l.epos + (size_t)l.ecap * 2 + 15) / 16 * 16;          // double[ecap]
    l.total = (l.eval + (size_t)l.ecap * 8 + 255) / 256 * 256;
    return l;
}
