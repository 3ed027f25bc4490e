// trs_solve_small: the whole Truss.Solve() path (slientruss3d/truss.py:329-364) of a SMALL truss
// (n_free <= 128: bar-6 ... bar-120, cube-7, every GA individual of ga.py:139-149) in ONE kernel,
// one 256-thread work-group per truss, everything resident in that work-group's LDS:
//
//   dofmap      free-DOF numbering (truss.py:319-326)                         wave 0, ballot prefix
//   assemble    member k, cosines, sorted joint adjacency, reduced K and f    (truss.py:56-86,303-316,343)
//   factor      blocked right-looking Cholesky, 16-column blocks: the 16 x 16 diagonal block and its
//               inverse by wave 0 on the matrix core (chol16_invert), the block column below it as a
//               product with that inverse, the trailing update in 2 x 2 register tiles; the load vector
//               rides along as one more row, so L y = f costs nothing extra      (np.linalg.solve, truss.py:343)
//   substitute  L^T u = y by blocks from the bottom, with the inverses kept in the free upper
//               triangles of the diagonal blocks
//   recover     displacement scatter, member forces, reactions (fixed summation order)  (truss.py:342-359)
//   fitness     optionally the GA reductions weight / stress / displacement excess        (truss.py:166-168,429-462)
//
// HBM traffic is the truss's inputs once and its results once; the stiffness matrix never leaves the
// CU.  No floating-point atomics: results are bit-reproducible.
//
// LDS layout of the matrix: lower block-triangle by 16-row block rows, block row I = 16 rows of
// W_I = 16 (I + 1) + 2 doubles (the 2 pad doubles stagger the rows over the LDS banks); the diagonal
// blocks are kept whole and symmetric until they are factored.  Row 16 nb (one row) is the load vector.
#include "trs_common.h"
#include "trs_chol16.h"
#include "../../include/trs_solver.h"

namespace {

constexpr int SNT = 256;       // threads per work-group
constexpr int SMALL_MAX_N = 128;

__host__ __device__ constexpr int sm_width(int I) { return 16 * (I + 1) + 2; }
__host__ __device__ constexpr int sm_base(int I) { return 128 * I * (I + 1) + 32 * I; }  // doubles before block row I

// LDS carve-up in bytes (every part 16-byte aligned), shared by host and device
struct SmallLds {
    size_t K, invw, scratch, xyz, uvec, mk, mc, diag, fi, cnt, start, fill, adj, flags, total;
};
__host__ __device__ inline SmallLds small_lds_layout(int nJ_max, int nM_max, int nb_max) {
    auto up = [](size_t v) { return (v + 15) / 16 * 16; };
    SmallLds l;
    l.K = 0;                                                                   // matrix + load row
    l.invw = up(l.K + ((size_t)sm_base(nb_max) + sm_width(nb_max)) * 8);       // double[256] inv(L_JJ) fragments
    l.scratch = l.invw + 256 * 8;                                              // double[64] chol16 scratch
    l.xyz = l.scratch + 64 * 8;                                                // double[3 nJ]
    l.uvec = up(l.xyz + (size_t)3 * nJ_max * 8);                               // double[3 nJ] full displacement
    l.mk = up(l.uvec + (size_t)3 * nJ_max * 8);                                // double[nM]   E A / L
    l.mc = up(l.mk + (size_t)nM_max * 8);                                      // double[3 nM] direction cosines
    l.diag = up(l.mc + (size_t)3 * nM_max * 8);                                // double[6 nJ] joint diagonal blocks
    l.fi = up(l.diag + (size_t)6 * nJ_max * 8);                                // int[3 nJ]
    l.cnt = up(l.fi + (size_t)3 * nJ_max * 4);                                 // int[nJ]
    l.start = up(l.cnt + (size_t)nJ_max * 4);                                  // int[nJ + 1]
    l.fill = up(l.start + (size_t)(nJ_max + 1) * 4);                           // int[nJ]
    l.adj = up(l.fill + (size_t)nJ_max * 4);                                   // unsigned[2 nM]
    l.flags = up(l.adj + (size_t)2 * nM_max * 4);                              // int[4]
    l.total = l.flags + 16;
    return l;
}

// deterministic sum of one double per thread over the work-group (same order as trs_fitness_kernel)
__device__ __forceinline__ double wg_sum(double v, double* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const double total = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return total;
}

struct SmallArgs {
    const double* xyz;
    TrsMembers mem;   // end joints and sections, general or table form (trs_common.h)
    const uint8_t* cbits;
    const double* loads;
    const int* nJ;
    const int* nM;
    int nJ_max, nM_max, nb_max;
    double* u;
    double* f_ext;
    double* N;
    int* info;
    int* free_index;  // optional outputs of the dofmap stage
    int* n_free;
    double allow_stress, allow_displace;   // (the densities of the fitness reductions travel in `mem`)
    double* weight;
    double* stress_vio;
    double* disp_vio;
};

__global__ __launch_bounds__(SNT) void trs_solve_small_kernel(const SmallArgs a) {
    extern __shared__ unsigned char lds_raw[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nJ = a.nJ[b], nM = a.nM[b];
    const int nJ_max = a.nJ_max, nM_max = a.nM_max;
    const SmallLds lay = small_lds_layout(nJ_max, nM_max, a.nb_max);
    double* Kb = reinterpret_cast<double*>(lds_raw + lay.K);
    double* invw = reinterpret_cast<double*>(lds_raw + lay.invw);
    double* scratch = reinterpret_cast<double*>(lds_raw + lay.scratch);
    double* X = reinterpret_cast<double*>(lds_raw + lay.xyz);
    double* uvec = reinterpret_cast<double*>(lds_raw + lay.uvec);
    double* mk = reinterpret_cast<double*>(lds_raw + lay.mk);
    double* mc = reinterpret_cast<double*>(lds_raw + lay.mc);
    double* diag = reinterpret_cast<double*>(lds_raw + lay.diag);
    int* fi = reinterpret_cast<int*>(lds_raw + lay.fi);
    int* cnt = reinterpret_cast<int*>(lds_raw + lay.cnt);
    int* start = reinterpret_cast<int*>(lds_raw + lay.start);
    int* fill = reinterpret_cast<int*>(lds_raw + lay.fill);
    unsigned* adj = reinterpret_cast<unsigned*>(lds_raw + lay.adj);
    int* flags = reinterpret_cast<int*>(lds_raw + lay.flags);  // [0] n_free, [1] info

    // ---- dofmap + loads of the joint data ----------------------------------------------------------
    const double* Xg = a.xyz + (size_t)b * 3 * nJ_max;
    const double* Fg = a.loads + (size_t)b * 3 * nJ_max;
    for (int d = tid; d < 3 * nJ; d += SNT) X[d] = Xg[d];
    for (int j = tid; j < nJ; j += SNT) cnt[j] = 0;
    if (wave == 0) {
        const uint8_t* cb = a.cbits + (size_t)b * nJ_max;
        int* fig = a.free_index != nullptr ? a.free_index + (size_t)b * 3 * nJ_max : nullptr;
        int base = 0;
        for (int d0 = 0; d0 < 3 * nJ_max; d0 += 64) {
            const int d = d0 + lane;
            bool is_free = false;
            if (d < 3 * nJ) is_free = ((cb[d / 3] >> (d % 3)) & 1) == 0;
            const unsigned long long mask = __ballot(is_free);
            const int idx = is_free ? base + __popcll(mask & ((1ull << lane) - 1ull)) : -1;
            if (d < 3 * nJ_max) {
                fi[d] = idx;
                if (fig != nullptr) fig[d] = idx;
            }
            base += __popcll(mask);
        }
        if (lane == 0) {
            flags[0] = base;
            flags[1] = 0;
            if (a.n_free != nullptr) a.n_free[b] = base;
        }
    }
    __syncthreads();
    const int n = flags[0];
    const int nb = (n + 15) >> 4;  // 16-row blocks of this truss (<= nb_max by the caller's bound)
    if (nb > a.nb_max) {           // the host-side bound was wrong: refuse loudly instead of overrunning LDS
        if (tid == 0) a.info[b] = -1;
        return;
    }
    const int yrow = sm_base(nb);  // the load vector: one row of width 16 nb (+2)
    {   // zero the matrix and the load row
        const int total = sm_base(nb) + sm_width(nb);  // even
        for (int x = 2 * tid; x < total; x += 2 * SNT) *reinterpret_cast<d2*>(Kb + x) = d2{0.0, 0.0};
    }
    // ---- member geometry, sorted joint adjacency, joint diagonal blocks (as trs_assemble, phase 0) ----
    const size_t mbase = (size_t)b * nM_max;
    auto ends = [&](int m) { return a.mem.ends(mbase + m); };
    for (int m = tid; m < nM; m += SNT) {
        const size_t mm = mbase + m;
        const int2 c01 = ends(m);
        const int j0 = c01.x, j1 = c01.y;
        double d[3], len2 = 0.0;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            d[q] = X[3 * j1 + q] - X[3 * j0 + q];
            len2 += d[q] * d[q];
        }
        const double len = sqrt(len2);
        mk[m] = a.mem.EA(mm) / len;                                // truss.py:56-58
#pragma unroll
        for (int q = 0; q < 3; ++q) mc[3 * m + q] = d[q] / len;    // truss.py:60-63
        atomicAdd(&cnt[j0], 1);
        atomicAdd(&cnt[j1], 1);
    }
    __syncthreads();
    if (wave == 0) {  // exclusive scan of cnt
        int base = 0;
        for (int j0 = 0; j0 < nJ; j0 += 64) {
            const int j = j0 + lane;
            const int v = j < nJ ? cnt[j] : 0;
            int incl = v;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int up = __shfl_up(incl, off);
                if (lane >= off) incl += up;
            }
            if (j < nJ) start[j] = base + incl - v;
            base += __shfl(incl, 63);
        }
        if (lane == 0) start[nJ] = base;
    }
    for (int j = tid; j < nJ; j += SNT) fill[j] = 0;
    __syncthreads();
    for (int m = tid; m < nM; m += SNT) {
        const int2 c01 = ends(m);
        const int j0 = c01.x, j1 = c01.y;
        adj[start[j0] + atomicAdd(&fill[j0], 1)] = ((unsigned)j1 << 16) | (unsigned)m;
        adj[start[j1] + atomicAdd(&fill[j1], 1)] = ((unsigned)j0 << 16) | (unsigned)m;
    }
    __syncthreads();
    for (int j = tid; j < nJ; j += SNT) {  // sort by (other joint, member); diagonal block in that order
        unsigned* list = adj + start[j];
        const int deg = cnt[j];
        for (int i = 1; i < deg; ++i) {
            const unsigned key = list[i];
            int p = i - 1;
            while (p >= 0 && list[p] > key) {
                list[p + 1] = list[p];
                --p;
            }
            list[p + 1] = key;
        }
        double dg[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        for (int i = 0; i < deg; ++i) {
            const int m = (int)(list[i] & 0xffffu);
            const double k = mk[m], cx = mc[3 * m], cy = mc[3 * m + 1], cz = mc[3 * m + 2];
            dg[0] += k * (cx * cx);  // truss.py:65-77
            dg[1] += k * (cx * cy);
            dg[2] += k * (cx * cz);
            dg[3] += k * (cy * cy);
            dg[4] += k * (cy * cz);
            dg[5] += k * (cz * cz);
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) diag[6 * j + q] = dg[q];
    }
    __syncthreads();
    // ---- reduced stiffness matrix: one thread per free DOF row, every entry written by one thread ----
    // stored: columns q of row c with block(q) <= block(c) (lower block-triangle, diagonal blocks whole)
    for (int d = tid; d < 3 * nJ; d += SNT) {
        const int c = fi[d];
        if (c < 0) continue;
        const int jn = d / 3, r = d - 3 * jn;
        double* row = Kb + sm_base(c >> 4) + (c & 15) * sm_width(c >> 4);
        const int qlim = (c | 15);  // last column stored in this row
        {
            const double* dg = diag + 6 * jn;  // row r of [xx xy xz; xy yy yz; xz yz zz]
            const double v0 = dg[r], v1 = dg[r == 0 ? 1 : (r == 1 ? 3 : 4)], v2 = dg[r == 0 ? 2 : (r == 1 ? 4 : 5)];
            const int q0 = fi[3 * jn], q1 = fi[3 * jn + 1], q2 = fi[3 * jn + 2];
            if (q0 >= 0 && q0 <= qlim) row[q0] = v0;
            if (q1 >= 0 && q1 <= qlim) row[q1] = v1;
            if (q2 >= 0 && q2 <= qlim) row[q2] = v2;
        }
        const unsigned* list = adj + start[jn];
        const int deg = cnt[jn];
        for (int i = 0; i < deg;) {
            const int other = (int)(list[i] >> 16);
            double v0 = 0.0, v1 = 0.0, v2 = 0.0;
            do {  // parallel members between the same two joints, in member order
                const int m = (int)(list[i] & 0xffffu);
                const double k = mk[m], cr = mc[3 * m + r];
                v0 -= k * (cr * mc[3 * m]);
                v1 -= k * (cr * mc[3 * m + 1]);
                v2 -= k * (cr * mc[3 * m + 2]);
                ++i;
            } while (i < deg && (int)(list[i] >> 16) == other);
            const int q0 = fi[3 * other], q1 = fi[3 * other + 1], q2 = fi[3 * other + 2];
            if (q0 >= 0 && q0 <= qlim) row[q0] = v0;
            if (q1 >= 0 && q1 <= qlim) row[q1] = v1;
            if (q2 >= 0 && q2 <= qlim) row[q2] = v2;
        }
        Kb[yrow + c] = Fg[d];  // truss.py:303-304, vecF[mask]
    }
    for (int c = n + tid; c < 16 * nb; c += SNT)  // identity padding up to the block boundary
        Kb[sm_base(c >> 4) + (c & 15) * sm_width(c >> 4) + c] = 1.0;
    __syncthreads();

    // ---- blocked Cholesky, the load vector as row 16 nb ------------------------------------------------
    for (int J = 0; J < nb; ++J) {
        const int c0 = 16 * J;
        double* blockJ = Kb + sm_base(J);
        const int WJ = sm_width(J);
        if (wave == 0) {  // A1: factor the diagonal block, keep inv(L_JJ)
            const int li = lane & 15, lq = lane >> 4;
            d4 t;
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = blockJ[li * WJ + c0 + lq + 4 * r];  // symmetric: T[c][i] = K[i][c]
            __builtin_amdgcn_s_setprio(3);
            const Chol16 f = chol16_invert_lds(t, (lds_f64*)scratch, (lds_f64*)invw);
            __builtin_amdgcn_s_setprio(0);
            const int bad = __builtin_amdgcn_readfirstlane(f.bad);
            // L on and below the diagonal (position row i = li, column c = lq + 4 r holds L[i][c]); above it
            // the strictly-lower part of inv(L_JJ), transposed: position (i, c), c > i, holds inv(L)[c][i]
#pragma unroll
            for (int r = 0; r < 4; ++r) blockJ[li * WJ + c0 + lq + 4 * r] = f.u[r];
            if (bad >= 0 && lane == 0) flags[1] = c0 + bad + 1;
        }
        __syncthreads();
        if (flags[1] != 0) break;
        const int rows_below = 16 * (nb - J - 1) + 1;  // rows 16 (J+1) .. 16 nb - 1 and the load row
        {   // A2: X = K_panel inv(L_JJ)^T, in place; a row's 16 threads sit in one wave
            const int k = tid & 15;
            double wk[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) wk[c] = c <= k ? invw[wfrag_index(k, c)] : 0.0;
            for (int rr = tid >> 4; rr < rows_below; rr += SNT / 16) {
                const int i = 16 * (J + 1) + rr;
                double* row = i < 16 * nb ? Kb + sm_base(i >> 4) + (i & 15) * sm_width(i >> 4) : Kb + yrow;
                d2 v[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = *reinterpret_cast<const d2*>(row + c0 + 2 * c);
                double x = 0.0;
#pragma unroll
                for (int c = 0; c < 8; ++c) x += v[c][0] * wk[2 * c] + v[c][1] * wk[2 * c + 1];
                row[c0 + k] = x;
            }
        }
        __syncthreads();
        {   // A3: trailing update C -= X X^T in 16 x 16 tiles, 2 x 2 entries per lane
            const int m = nb - J - 1;
            const int ntile = m * (m + 1) / 2;
            const int la = lane & 7, lb = lane >> 3;
            for (int q = wave; q < ntile + m; q += SNT / 64) {
                if (q < ntile) {
                    int r = 0;
                    while ((r + 1) * (r + 2) / 2 <= q) ++r;
                    const int I = J + 1 + r, Kc = J + 1 + (q - r * (r + 1) / 2);
                    const int WI = sm_width(I), WK = sm_width(Kc);
                    const double* xi = Kb + sm_base(I) + la * WI + c0;
                    const double* xk = Kb + sm_base(Kc) + lb * WK + c0;
                    double s00 = 0.0, s01 = 0.0, s10 = 0.0, s11 = 0.0;
#pragma unroll
                    for (int c = 0; c < 16; c += 2) {
                        const d2 i0 = *reinterpret_cast<const d2*>(xi + c);
                        const d2 i1 = *reinterpret_cast<const d2*>(xi + 8 * WI + c);
                        const d2 k0 = *reinterpret_cast<const d2*>(xk + c);
                        const d2 k1 = *reinterpret_cast<const d2*>(xk + 8 * WK + c);
                        s00 += i0[0] * k0[0] + i0[1] * k0[1];
                        s01 += i0[0] * k1[0] + i0[1] * k1[1];
                        s10 += i1[0] * k0[0] + i1[1] * k0[1];
                        s11 += i1[0] * k1[0] + i1[1] * k1[1];
                    }
                    double* ct = Kb + sm_base(I) + la * WI + 16 * Kc + lb;
                    ct[0] -= s00;
                    ct[8] -= s01;
                    ct[8 * WI] -= s10;
                    ct[8 * WI + 8] -= s11;
                } else if (lane < 16) {  // the load row against block row Kc
                    const int Kc = J + 1 + (q - ntile), WK = sm_width(Kc);
                    const double* xk = Kb + sm_base(Kc) + lane * WK + c0;
                    const double* xy = Kb + yrow + c0;
                    double s = 0.0;
#pragma unroll
                    for (int c = 0; c < 16; ++c) s += xy[c] * xk[c];
                    Kb[yrow + 16 * Kc + lane] -= s;
                }
            }
        }
        __syncthreads();
    }
    const int bad_col = flags[1];

    // ---- back substitution L^T u = y, by blocks from the bottom; u overwrites y ------------------------
    if (bad_col == 0) {
        for (int J = nb - 1; J >= 0; --J) {
            const int c0 = 16 * J, WJ = sm_width(J);
            const double* blockJ = Kb + sm_base(J);
            if (wave == 0) {
                // u_J = inv(L_JJ)^T t_J: u[l] = t[l] / L[l][l] + sum_{j > l} inv(L)[j][l] t[j], the inverse
                // sitting transposed above the diagonal of row l (contiguous)
                const int l = lane & 15, g = lane >> 4;
                double part = 0.0;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int j = 4 * g + jj;
                    const double w = blockJ[l * WJ + c0 + j];
                    const double tj = Kb[yrow + c0 + j];
                    part += (j > l ? w : (j == l ? 1.0 / w : 0.0)) * tj;
                }
                part += __shfl_xor(part, 16);
                part += __shfl_xor(part, 32);
                __builtin_amdgcn_wave_barrier();
                if (g == 0) Kb[yrow + c0 + l] = part;
            }
            __syncthreads();
            if (J > 0) {  // y[c'] -= sum_{i in block J} L[i][c'] u[i] for the columns left of the block
                for (int cc = tid; cc < c0; cc += SNT) {
                    double s = 0.0;
#pragma unroll
                    for (int i = 0; i < 16; ++i) s += blockJ[i * WJ + cc] * Kb[yrow + c0 + i];
                    Kb[yrow + cc] -= s;
                }
                __syncthreads();
            }
        }
    }

    // ---- recover: displacements, member forces, reactions (truss.py:342-359) ------------------------------
    for (int d = tid; d < 3 * nJ_max; d += SNT) {
        const int c = d < 3 * nJ ? fi[d] : -1;
        uvec[d] = c >= 0 ? Kb[yrow + c] : 0.0;
    }
    __syncthreads();
    auto axial_of = [&](int m) {
        const int2 c01 = ends(m);
        const int j0 = c01.x, j1 = c01.y;
        double proj = 0.0;
#pragma unroll
        for (int q = 0; q < 3; ++q) proj += mc[3 * m + q] * (uvec[3 * j1 + q] - uvec[3 * j0 + q]);
        return mk[m] * proj;  // N = (E A / L) c . (u1 - u0), tension positive (truss.py:89-91,354-359)
    };
    double* Ng = a.N + (size_t)b * nM_max;
    for (int m = tid; m < nM_max; m += SNT) Ng[m] = m < nM ? axial_of(m) : 0.0;
    double* ug = a.u + (size_t)b * 3 * nJ_max;
    double* fg = a.f_ext + (size_t)b * 3 * nJ_max;
    for (int j = tid; j < nJ_max; j += SNT) {
        double f[3] = {0.0, 0.0, 0.0};
        if (j < nJ) {
            const bool constrained = (fi[3 * j] < 0) | (fi[3 * j + 1] < 0) | (fi[3 * j + 2] < 0);
            double r[3] = {0.0, 0.0, 0.0};
            if (constrained) {  // reaction = sum of member end forces, in the order of the sorted list
                const unsigned* list = adj + start[j];
                const int deg = cnt[j];
                for (int i = 0; i < deg; ++i) {
                    const int m = (int)(list[i] & 0xffffu);
                    const double ax = axial_of(m);
                    const double sgn = ends(m).y == j ? 1.0 : -1.0;  // N c on joint1, -N c on joint0
#pragma unroll
                    for (int q = 0; q < 3; ++q) r[q] += sgn * ax * mc[3 * m + q];
                }
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) f[q] = fi[3 * j + q] >= 0 ? Fg[3 * j + q] : r[q];
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            ug[3 * j + q] = uvec[3 * j + q];
            fg[3 * j + q] = f[q];
        }
    }
    if (tid == 0) a.info[b] = bad_col;

    // ---- optional: GA fitness reductions (truss.py:166-168,429-462; ga.py:139-149) ----------------------
    if (a.weight != nullptr) {
        double* red = scratch;
        double w = 0.0, sv = 0.0, dv = 0.0;
        for (int m = tid; m < nM; m += SNT) {
            const size_t mm = mbase + m;
            const int2 c01 = ends(m);
            const int j0 = c01.x, j1 = c01.y;
            double len2 = 0.0;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const double dd = X[3 * j1 + q] - X[3 * j0 + q];
                len2 += dd * dd;
            }
            const double area = a.mem.area(mm);
            w += area * sqrt(len2) * a.mem.density(mm);
            const double force = axial_of(m);
            if (fabs(force) >= 1e-10) {
                const double s = fabs(force) / area;
                if (s > a.allow_stress) sv += s - a.allow_stress;
            }
        }
        for (int j = tid; j < nJ; j += SNT) {
            const double ux = uvec[3 * j], uy = uvec[3 * j + 1], uz = uvec[3 * j + 2];
            if (fabs(ux) >= 1e-10 || fabs(uy) >= 1e-10 || fabs(uz) >= 1e-10) {
                const double l = sqrt(ux * ux + uy * uy + uz * uz);
                if (l > a.allow_displace) dv += l - a.allow_displace;
            }
        }
        __syncthreads();  // scratch is free: the factorisation is over
        w = wg_sum(w, red);
        sv = wg_sum(sv, red);
        dv = wg_sum(dv, red);
        if (tid == 0) {
            a.weight[b] = w;
            a.stress_vio[b] = sv;
            a.disp_vio[b] = dv;
        }
    }
}

}  // namespace

extern "C" int trs_solve_small_fits(int nJ_max, int nM_max, int n_max_bound) {
    if (n_max_bound < 0 || n_max_bound > SMALL_MAX_N || nJ_max <= 0 || nM_max < 0) return 0;
    if (nJ_max >= 65536 || nM_max >= 65536) return 0;
    const int nb_max = (n_max_bound + 15) / 16;
    return small_lds_layout(nJ_max, nM_max < 1 ? 1 : nM_max, nb_max).total <= (size_t)160 * 1024;
}

extern "C" int trs_solve_small_launch(int B, int nJ_max, int nM_max, int n_max_bound, const double* xyz,
                                      const TrsMembers* members, const uint8_t* cbits,
                                      const double* loads, const int* nJ, const int* nM, double* u,
                                      double* f_ext, double* N, int* info, int* free_index, int* n_free,
                                      double allow_stress, double allow_displace,
                                      double* weight, double* stress_vio, double* disp_vio,
                                      hipStream_t stream) {
    if (B <= 0) return 0;
    if (!trs_solve_small_fits(nJ_max, nM_max, n_max_bound)) return (int)hipErrorInvalidValue;
    if (weight != nullptr && ((members->rho == nullptr && members->tidx == nullptr) || stress_vio == nullptr || disp_vio == nullptr))
        return (int)hipErrorInvalidValue;
    const int nb_max = (n_max_bound + 15) / 16;
    const size_t lds = small_lds_layout(nJ_max, nM_max < 1 ? 1 : nM_max, nb_max).total;
    static const int lds_limit_set = (int)hipFuncSetAttribute(
        reinterpret_cast<const void*>(trs_solve_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
        160 * 1024);
    (void)lds_limit_set;
    SmallArgs args{xyz,    *members, cbits, loads, nJ, nM, nJ_max, nM_max < 1 ? 1 : nM_max, nb_max, u, f_ext, N, info,
                   free_index, n_free, allow_stress, allow_displace, weight, stress_vio, disp_vio};
    hipLaunchKernelGGL(trs_solve_small_kernel, dim3(B), dim3(SNT), lds, stream, args);
    return (int)hipGetLastError();
}
