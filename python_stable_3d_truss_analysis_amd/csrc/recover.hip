// Result recovery for a batch of solved trusses: displacement scatter (slientruss3d/truss.py:342),
// member axial forces (truss.py:354-359 with Member.IsTension truss.py:89-91: N = (EA/L) c.(u1-u0),
// tension positive) and the external force vector (truss.py:348-349): applied load at free DOFs,
// stiffness reaction K u at constrained DOFs.  The reaction is summed from member end forces
// (N c on joint1, -N c on joint0) instead of a dense K[~mask,:] @ u.
//
// Also the constraint reductions the GA fitness needs (truss.py:166-168,429-462; ga.py:139-149).
#include "trs_common.h"
#include "../../include/trs_solver.h"

namespace {

struct MemberGeom {
    double len, c[3];
};

__device__ __forceinline__ MemberGeom member_geom(const double* X, int j0, int j1) {
    MemberGeom g;
    double d[3], len2 = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        d[a] = X[3 * j1 + a] - X[3 * j0 + a];
        len2 += d[a] * d[a];
    }
    g.len = sqrt(len2);
#pragma unroll
    for (int a = 0; a < 3; ++a) g.c[a] = d[a] / g.len;
    return g;
}

// axial force of a member from the displacements of its end joints: N = (E A / L) c . (u1 - u0)
__device__ __forceinline__ double member_axial(const MemberGeom& g, double EA, const double* u, int j0, int j1) {
    const double k = EA / g.len;
    double proj = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) proj += g.c[a] * (u[3 * j1 + a] - u[3 * j0 + a]);
    return k * proj;
}

// The end force of member (g, axial) on its joint `end` (1: +N c on joint1, 0: -N c on joint0), added to a
// running reaction sum.  ONE function with explicit fused multiply-adds for every path of the kernel: the paths
// then round alike and their reactions are equal bit for bit.
__device__ __forceinline__ void add_end_force(double (&r)[3], const double (&c)[3], double axial, int end) {
    const double s = end ? axial : -axial;
#pragma unroll
    for (int a = 0; a < 3; ++a) r[a] = fma(s, c[a], r[a]);
}

// Reaction at one constrained joint from its list of member ends ((member << 1) | end): the list is sorted by
// member id in place (insertion sort: the lists are short) and summed in that order.
template <class ListPtr>
__device__ __forceinline__ void joint_reaction(ListPtr list, const int deg, const TrsMembers& mem,
                                               const double* __restrict__ X, const size_t mbase, const double* u,
                                               const int* jo, double (&r)[3]) {
    for (int i = 1; i < deg; ++i) {
        const int key = list[i];
        int p = i - 1;
        while (p >= 0 && list[p] > key) {
            list[p + 1] = list[p];
            --p;
        }
        list[p + 1] = key;
    }
    r[0] = r[1] = r[2] = 0.0;
    for (int i = 0; i < deg; ++i) {
        const int m = list[i] >> 1, end = list[i] & 1;
        const int2 c = mem.ends(mbase + m);
        const MemberGeom g = member_geom(X, c.x, c.y);
        const double axial = member_axial(g, mem.EA(mbase + m), u, jo ? jo[c.x] : c.x, jo ? jo[c.y] : c.y);
        add_end_force(r, g.c, axial, end);
    }
}

// entries of the member-end lists the path without LDS staging keeps in LDS (128 KB)
#define TRS_RECOVER_LIST_CAP 32768

// STAGED: u and f_ext of the truss are staged in LDS (gathers and the reaction sums stay on chip), and
// the reactions are summed in a FIXED order: the member ends at constrained joints are counting-sorted
// by joint (integer LDS atomics), every such joint's short list is sorted by member id and one thread
// adds its members' end forces in that order - no floating-point atomics, bit-reproducible.
// !STAGED - trusses with more than ~1500 joints, whose tables exceed a CU's LDS: u and f_ext live directly in
// the output arrays; the reactions are summed in the SAME fixed order (results bit for bit those of the staged
// path).  Only the member ends at CONSTRAINED joints need lists, and those are few (the supports): the lists are
// kept in LDS (TRS_RECOVER_LIST_CAP entries), their per-joint (count, start) pair in the one place of the output
// that is free until the reaction is known - the f_ext entry of the joint's first constrained axis, as two
// 32-bit integers, counted with integer atomics.  A truss with more member ends at its supports than the lists
// hold (or `scan_only`, for tests) takes the slow path without lists: a wave per constrained joint walks ALL
// members in id order.  No floating-point atomics anywhere.
template <bool STAGED>
__global__ __launch_bounds__(256) void trs_recover_kernel(
    const double* __restrict__ xyz, const TrsMembers mem, const double* __restrict__ loads,
    const int* __restrict__ free_index, const int* __restrict__ nJ, const int* __restrict__ nM,
    const int nJ_max, const int nM_max, const double* __restrict__ uf, const int ld_uf,
    double* __restrict__ u_out, double* __restrict__ f_out, double* __restrict__ N_out,
    const int* __restrict__ joint_out,
    // scatter form (trs_recover_rows; null / 0 otherwise): the results of truss b go to row out_rows[b] of result
    // arrays whose rows are nJ_out / nM_out wide (>= this batch's), and its factorisation status to info_out
    const long long* __restrict__ out_rows, const int nJ_out, const int nM_out, const int* __restrict__ info_in,
    int* __restrict__ info_out, const int scan_only) {
    extern __shared__ double sh[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int joints = nJ[b];
    const int ndof = 3 * joints, ndof_max = 3 * nJ_max;
    const size_t orow = out_rows != nullptr ? (size_t)out_rows[b] : (size_t)b;
    const size_t odof = out_rows != nullptr ? (size_t)3 * nJ_out : (size_t)ndof_max;   // doubles per output row of u / f_ext
    const size_t omem = out_rows != nullptr ? (size_t)nM_out : (size_t)nM_max;
    if (out_rows != nullptr && tid == 0 && info_out != nullptr) info_out[orow] = info_in[b];
    double *u, *f;  // [ndof_max] each
    if constexpr (STAGED) {
        u = sh;
        f = sh + ndof_max;
    } else {
        u = u_out + orow * odof;
        f = f_out + orow * odof;
    }
    // STAGED only: integer tables behind u and f
    int* cnt = reinterpret_cast<int*>(sh + 2 * ndof_max);  // [nJ_max]   member ends at a constrained joint
    int* start = cnt + nJ_max;                             // [nJ_max+1] exclusive scan of cnt
    int* ends = start + nJ_max + 1;                        // [2 nM_max] (member << 1) | end, grouped by joint
    unsigned char* held = reinterpret_cast<unsigned char*>(ends + 2 * nM_max);   // [nJ_max] joint has a constrained axis
    const int* fi = free_index + (size_t)b * ndof_max;
    const double* F = loads + (size_t)b * ndof_max;
    const double* ufb = uf + (size_t)b * ld_uf;
    const int members = nM[b];
    // The end joints of a thread's first MR members stay in registers: both passes over the members need them (axial
    // forces; grouping of the member ends at constrained joints), and the second read came from HBM again - with
    // eight trusses per CU the first pass's lines are long gone from L2 (a third of this kernel's excess traffic).
    constexpr int MR = 4;
    int2 cjr[MR];
    const size_t mbase = (size_t)b * nM_max;
    auto CNI = [&](int m) { return mem.ends(mbase + m); };
#pragma unroll
    for (int r = 0; r < MR; ++r) {
        const int m = tid + 256 * r;
        cjr[r] = m < members ? CNI(m) : int2{0, 0};
    }
    // joint_out (optional): results of joint j go to row joint_out[b][j] of u / f_ext - a batch whose joints
    // were renumbered for a narrower envelope delivers its results in the caller's numbering at no cost.
    // STAGED: the map is applied by the final copy out of LDS; otherwise u and f ARE the output arrays and
    // every access goes through it.
    const int* jo = joint_out ? joint_out + (size_t)b * nJ_max : nullptr;
    auto J = [&](int j) { return (!STAGED && jo) ? jo[j] : j; };
    for (int d = tid; d < ndof_max; d += 256) {
        const int r = d < ndof ? fi[d] : -1;
        const int o = 3 * J(d / 3) + d % 3;
        u[o] = r >= 0 ? ufb[r] : 0.0;
        f[o] = r >= 0 ? F[d] : 0.0;  // constrained: reaction accumulated below (load ignored)
    }
    if constexpr (STAGED)
        for (int j = tid; j < nJ_max; j += 256) {
            cnt[j] = 0;
            held[j] = j < joints ? (unsigned char)((fi[3 * j] < 0) | (fi[3 * j + 1] < 0) | (fi[3 * j + 2] < 0)) : (unsigned char)0;
        }
    if constexpr (!STAGED) __threadfence();
    __syncthreads();
    const double* X = xyz + (size_t)b * ndof_max;
    auto constrained = [&](int j) {
        if constexpr (STAGED) return (int)held[j];
        else return (fi[3 * j] < 0) | (fi[3 * j + 1] < 0) | (fi[3 * j + 2] < 0);
    };
    // !STAGED: the (count / fill cursor, list start) pair of constrained joint j - two ints in the f_ext entry of its
    // first constrained axis (zeroed above; the reaction overwrites it at the end)
    auto slot = [&](int j) {
        const int a0 = fi[3 * j] < 0 ? 0 : (fi[3 * j + 1] < 0 ? 1 : 2);
        return reinterpret_cast<int*>(&f[3 * J(j) + a0]);
    };
    auto slot_get = [&](int j, int k) { return __hip_atomic_load(slot(j) + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto slot_set = [&](int j, int k, int v) { __hip_atomic_store(slot(j) + k, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto ends_of = [&](int r, int m) {   // end joints of member m = tid + 256 r of this thread
        int2 c = CNI(m < members ? m : 0);
#pragma unroll
        for (int q = 0; q < MR; ++q)
            if (r == q) c = cjr[q];
        return c;
    };
    for (int m = tid, r = 0; m < nM_max; m += 256, ++r) {
        const size_t mm = (size_t)b * nM_max + m;
        double axial = 0.0;
        if (m < members) {
            const int2 c = r < MR ? ends_of(r, m) : CNI(m);
            const int j0 = c.x, j1 = c.y;
            const MemberGeom g = member_geom(X, j0, j1);
            axial = member_axial(g, mem.EA(mm), u, J(j0), J(j1));
            if constexpr (STAGED) {
                if (constrained(j0)) atomicAdd(&cnt[j0], 1);
                if (constrained(j1)) atomicAdd(&cnt[j1], 1);
            } else {
                if (constrained(j0)) atomicAdd(slot(j0), 1);
                if (constrained(j1)) atomicAdd(slot(j1), 1);
            }
        }
        N_out[orow * omem + m] = axial;
    }
    if constexpr (STAGED) {
        __syncthreads();
        if (tid < 64) {  // exclusive scan of cnt by one wave
            int base = 0;
            for (int j0 = 0; j0 < joints; j0 += 64) {
                const int j = j0 + tid;
                const int v = j < joints ? cnt[j] : 0;
                int incl = v;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int up = __shfl_up(incl, off);
                    if (tid >= off) incl += up;
                }
                if (j < joints) start[j] = base + incl - v;
                base += __shfl(incl, 63);
            }
        }
        __syncthreads();
        for (int j = tid; j < joints; j += 256) cnt[j] = 0;  // reused as the fill cursor
        __syncthreads();
        for (int m = tid, r = 0; m < members; m += 256, ++r) {
            const int2 c = r < MR ? ends_of(r, m) : CNI(m);
            const int j0 = c.x, j1 = c.y;
            if (constrained(j0)) ends[start[j0] + atomicAdd(&cnt[j0], 1)] = m << 1;
            if (constrained(j1)) ends[start[j1] + atomicAdd(&cnt[j1], 1)] = (m << 1) | 1;
        }
        __syncthreads();
        for (int j = tid; j < joints; j += 256) {
            const int deg = cnt[j];
            if (deg == 0) continue;
            double r[3];
            joint_reaction(ends + start[j], deg, mem, X, mbase, u, nullptr, r);
#pragma unroll
            for (int a = 0; a < 3; ++a)
                if (fi[3 * j + a] < 0) f[3 * j + a] = r[a];
        }
        __syncthreads();
        if (jo != nullptr) {
            // results in the caller's numbering: the output rows are written IN ORDER (whole cache lines) and read out
            // of LDS through the inverse of the joint map, built in the dead counter table - not scattered joint by
            // joint (24-byte pieces of 128-byte lines)
            int* inv = cnt;
            for (int j = tid; j < nJ_max; j += 256) inv[jo[j]] = j;
            __syncthreads();
            for (int o = tid; o < ndof_max; o += 256) {
                const int d = 3 * inv[o / 3] + o % 3;
                u_out[orow * odof + o] = u[d];
                f_out[orow * odof + o] = d < ndof ? f[d] : 0.0;
            }
        } else {
            for (int d = tid; d < ndof_max; d += 256) {
                u_out[orow * odof + d] = u[d];
                f_out[orow * odof + d] = d < ndof ? f[d] : 0.0;
            }
        }
    } else {
        int* lists = reinterpret_cast<int*>(sh);        // [TRS_RECOVER_LIST_CAP] member ends, grouped by joint
        int* wsum = lists + TRS_RECOVER_LIST_CAP;        // [4] wave totals of the scan
        // (the phases of this path talk through global memory - the counts, starts and cursors in f_ext: every phase
        // boundary is a device-scope fence and a barrier; the path is rare and the fences are cheap)
        __threadfence();
        __syncthreads();
        // exclusive scan of the counts over the constrained joints, 256 joints a round
        int base = 0;
        for (int j0 = 0; j0 < joints; j0 += 256) {
            const int j = j0 + tid;
            const bool own = j < joints && constrained(j);
            const int v = own ? slot_get(j, 0) : 0;
            int incl = v;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int up = __shfl_up(incl, off);
                if ((tid & 63) >= off) incl += up;
            }
            if ((tid & 63) == 63) wsum[tid >> 6] = incl;
            __syncthreads();
            int before = 0, total = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                before += w < (tid >> 6) ? wsum[w] : 0;
                total += wsum[w];
            }
            if (own) {
                slot_set(j, 1, base + before + incl - v);
                slot_set(j, 0, 0);   // reused as the fill cursor
            }
            base += total;
            __syncthreads();
        }
        __threadfence();
        __syncthreads();
        if (base <= TRS_RECOVER_LIST_CAP && scan_only == 0) {
            for (int m = tid, r = 0; m < members; m += 256, ++r) {
                const int2 c = r < MR ? ends_of(r, m) : CNI(m);
                const int j0 = c.x, j1 = c.y;
                if (constrained(j0)) lists[slot_get(j0, 1) + atomicAdd(slot(j0), 1)] = m << 1;
                if (constrained(j1)) lists[slot_get(j1, 1) + atomicAdd(slot(j1), 1)] = (m << 1) | 1;
            }
            __threadfence();
            __syncthreads();
            for (int j = tid; j < joints; j += 256) {
                if (!constrained(j)) continue;
                double r[3];
                joint_reaction(lists + slot_get(j, 1), slot_get(j, 0), mem, X, mbase, u, jo, r);
#pragma unroll
                for (int a = 0; a < 3; ++a)
                    if (fi[3 * j + a] < 0) f[3 * J(j) + a] = r[a];   // (the joint's own slot among them)
            }
        } else {
            // no lists: a wave per constrained joint walks all members, 64 at a time, and adds the ends it finds in
            // member order (every lane carries the same sum)
            const int wave = tid >> 6, lane = tid & 63;
            for (int j = wave; j < joints; j += 4) {
                if (!constrained(j)) continue;
                double r[3] = {0.0, 0.0, 0.0};
                for (int m0 = 0; m0 < members; m0 += 64) {
                    const int m = m0 + lane;
                    const int2 c = m < members ? CNI(m) : int2{-1, -1};
                    const bool h0 = c.x == j, h1 = c.y == j;
                    const unsigned long long b0 = __ballot(h0), b1 = __ballot(h1);
                    unsigned long long any = b0 | b1;
                    if (any == 0ull) continue;
                    double axial = 0.0, cc[3] = {0.0, 0.0, 0.0};
                    if (h0 || h1) {
                        const MemberGeom g = member_geom(X, c.x, c.y);
                        const size_t mm = (size_t)b * nM_max + m;
                        axial = member_axial(g, mem.EA(mm), u, J(c.x), J(c.y));
#pragma unroll
                        for (int a = 0; a < 3; ++a) cc[a] = g.c[a];
                    }
                    while (any != 0ull) {
                        const int src = __builtin_ctzll(any);
                        any &= any - 1;
                        const double ax = __shfl(axial, src);
                        const double cs[3] = {__shfl(cc[0], src), __shfl(cc[1], src), __shfl(cc[2], src)};
                        if ((b0 >> src) & 1ull) add_end_force(r, cs, ax, 0);
                        if ((b1 >> src) & 1ull) add_end_force(r, cs, ax, 1);
                    }
                }
                if (lane == 0) {
#pragma unroll
                    for (int a = 0; a < 3; ++a)
                        if (fi[3 * j + a] < 0) f[3 * J(j) + a] = r[a];
                }
            }
        }
    }
}

// Deterministic block sum of one double per thread (256 threads).
__device__ __forceinline__ double block_sum(double v, double* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const double total = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return total;
}

__global__ __launch_bounds__(256) void trs_fitness_kernel(
    const double* __restrict__ xyz, const int* __restrict__ conn, const double* __restrict__ A,
    const double* __restrict__ rho, const int* __restrict__ nJ, const int* __restrict__ nM,
    const int nJ_max, const int nM_max, const double* __restrict__ u, const double* __restrict__ N,
    const double allow_stress, const double allow_displace, double* __restrict__ weight,
    double* __restrict__ stress_vio, double* __restrict__ disp_vio) {
    __shared__ double red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const double* X = xyz + (size_t)b * 3 * nJ_max;
    double w = 0.0, sv = 0.0, dv = 0.0;
    for (int m = tid; m < nM[b]; m += 256) {
        const size_t mm = (size_t)b * nM_max + m;
        const MemberGeom g = member_geom(X, conn[2 * mm], conn[2 * mm + 1]);
        w += A[mm] * g.len * rho[mm];  // truss.py:52-54
        const double force = N[mm];
        if (fabs(force) >= 1e-10) {   // members kept in the sparse dict (truss.py:358-359)
            const double s = fabs(force) / A[mm];
            if (s > allow_stress) sv += s - allow_stress;  // truss.py:432
        }
    }
    const double* U = u + (size_t)b * 3 * nJ_max;
    for (int j = tid; j < nJ[b]; j += 256) {
        const double ux = U[3 * j], uy = U[3 * j + 1], uz = U[3 * j + 2];
        if (fabs(ux) >= 1e-10 || fabs(uy) >= 1e-10 || fabs(uz) >= 1e-10) {  // truss.py:344-345
            const double l = sqrt(ux * ux + uy * uy + uz * uz);
            if (l > allow_displace) dv += l - allow_displace;  // truss.py:450
        }
    }
    w = block_sum(w, red);
    sv = block_sum(sv, red);
    dv = block_sum(dv, red);
    if (tid == 0) {
        weight[b] = w;
        stress_vio[b] = sv;
        disp_vio[b] = dv;
    }
}

}  // namespace

extern "C" int trs_recover_launch(int B, int nJ_max, int nM_max, const double* xyz, const TrsMembers* members,
                                  const double* loads,
                                  const int* free_index, const int* nJ, const int* nM,
                                  const double* uf, int ld_uf, double* u, double* f_ext, double* N,
                                  const int* joint_out, int force_unstaged, hipStream_t stream,
                                  const long long* out_rows, int nJ_out, int nM_out, const int* info_in,
                                  int* info_out) {
    if (B <= 0) return 0;
    // u, f_ext (doubles) + member-end tables (ints) + one "has a constrained axis" byte per joint
    const size_t lds = ((size_t)6 * nJ_max * sizeof(double) +
                        ((size_t)2 * nJ_max + 1 + 2 * (size_t)nM_max) * sizeof(int) + (size_t)nJ_max + 15) / 16 * 16;
    if (lds > 160 * 1024 || force_unstaged) {
        const size_t lds_lists = (size_t)(TRS_RECOVER_LIST_CAP + 8) * sizeof(int);
        static const int lists_limit_set = (int)hipFuncSetAttribute(
            reinterpret_cast<const void*>(trs_recover_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
            160 * 1024);
        (void)lists_limit_set;
        hipLaunchKernelGGL(trs_recover_kernel<false>, dim3(B), dim3(256), lds_lists, stream, xyz, *members, loads,
                           free_index, nJ, nM, nJ_max, nM_max, uf, ld_uf, u, f_ext, N, joint_out, out_rows, nJ_out,
                           nM_out, info_in, info_out, (force_unstaged & TRS_HINT_RECOVER_SCAN) != 0);
        return (int)hipGetLastError();
    }
    static const int lds_limit_set = (int)hipFuncSetAttribute(   // once per process, not per launch
        reinterpret_cast<const void*>(trs_recover_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
        160 * 1024);
    (void)lds_limit_set;
    hipLaunchKernelGGL(trs_recover_kernel<true>, dim3(B), dim3(256), lds, stream, xyz, *members, loads,
                       free_index, nJ, nM, nJ_max, nM_max, uf, ld_uf, u, f_ext, N, joint_out, out_rows, nJ_out, nM_out,
                       info_in, info_out, 0);
    return (int)hipGetLastError();
}

// Member sections of a GA population from its gene matrix (ga.py:125-137: locus i of a gene is the type of member i):
// one thread per (individual, member) of the padded arrays; rows past `count` and members past `n_member` get type 0
// (they are solved and ignored), a locus outside the table NaN sections (the solve then reports the individual).
__global__ __launch_bounds__(256) void trs_ga_sections_kernel(
    const unsigned char* __restrict__ genes, const double* __restrict__ table, const int count, const int n_member,
    const int n_type, const int nM_max, const long long total, double* __restrict__ A, double* __restrict__ E,
    double* __restrict__ rho) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int b = (int)(i / nM_max), m = (int)(i - (long long)b * nM_max);
    const int type = (b < count && m < n_member) ? (int)genes[(size_t)b * n_member + m] : 0;
    const bool ok = type < n_type;
    const double nan = __builtin_nan("");
    A[i] = ok ? table[3 * type] : nan;
    E[i] = ok ? table[3 * type + 1] : nan;
    rho[i] = ok ? table[3 * type + 2] : nan;
}

extern "C" int trs_ga_sections_launch(int B, int nM_max, int count, int n_member, int n_type, const unsigned char* genes,
                                      const double* table, double* A, double* E, double* rho, hipStream_t stream) {
    const long long total = (long long)B * nM_max;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(trs_ga_sections_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, genes, table,
                       count, n_member, n_type, nM_max, total, A, E, rho);
    return (int)hipGetLastError();
}

extern "C" int trs_fitness_launch(int B, int nJ_max, int nM_max, const double* xyz, const int* conn,
                                  const double* A, const double* rho, const int* nJ, const int* nM,
                                  const double* u, const double* N, double allow_stress,
                                  double allow_displace, double* weight, double* stress_vio,
                                  double* disp_vio, hipStream_t stream) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(trs_fitness_kernel, dim3(B), dim3(256), 0, stream, xyz, conn, A, rho, nJ, nM,
                       nJ_max, nM_max, u, N, allow_stress, allow_displace, weight, stress_vio,
                       disp_vio);
    return (int)hipGetLastError();
}
