// Random cube trusses generated ON THE DEVICE, straight into the padded batch arrays the solver reads
// ("tensor-native cube-truss generator"; BASELINE configs 3 and 5).
//
// Same construction as the reference's GenerateRandomCubeTrusses (slientruss3d/generate.py:152-376: polycube
// grown on an integer grid DFS / BFS / at random, joints = cube vertices in first-seen order, 6 face diagonals by
// LinkType + 12 edges per cube with ordered joint pairs linked once, pins on the lowest occupied z layer, 1 ..
// |free joints| random loads, a member type per member, count-unstable draws regenerated) and BIT FOR BIT the
// output of the host generator csrc/cubegen.c for the same (seed, global truss index): same per-truss splitmix64
// stream, same order of draws, same floating-point expressions (contraction off: the host file is compiled with
// -ffp-contract=off) - asserted in tests/test_gpu_generate.py.  The distribution is pinned against the reference
// itself through the host generator (tests/test_generate.py, two-sample KS tests).
//
// One WAVE per truss (four per work-group), its grid state in LDS (3.7 KB for a 6 x 6 x 6 grid): the growth of
// the polycube is serial by nature (which cell is popped depends on the draws so far) and runs as wave-uniform
// code; the per-cube work that is not - six neighbour tests, eight vertex look-ups, up to twenty-four candidate
// members with their "already linked" tests (a 27-direction bit mask per vertex instead of the host's hash set
// of ordered pairs) - is spread over the lanes with ballots and prefix counts, which also give the members their
// order.  splitmix64 is counter-based, so draws whose index is known (the six link choices of a cube, the member
// types) are computed in parallel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/trs_solver.h"
#pragma clang fp contract(off)

namespace {

constexpr unsigned long long GAMMA = 0x9e3779b97f4a7c15ULL;
constexpr int WPB = 4;  // waves (= trusses) per work-group

__host__ __device__ inline unsigned long long mix64(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
__device__ inline double unit_of(unsigned long long r) { return (double)(r >> 11) * (1.0 / 9007199254740992.0); }

struct Rng {  // splitmix64 (cubegen.c rng_next): draw k of the stream = mix64(s0 + (k + 1) GAMMA)
    unsigned long long s;
    __device__ unsigned long long next() { s += GAMMA; return mix64(s); }
    __device__ double unit() { return unit_of(next()); }
    __device__ int below(int n) { return (int)(unit() * n); }
    __device__ double uniform(double lo, double hi) { return lo + (hi - lo) * unit(); }
    // the draw `ahead` positions after the current one, without advancing (ahead = 0: the next draw)
    __device__ double unit_at(int ahead) const { return unit_of(mix64(s + (unsigned long long)(ahead + 1) * GAMMA)); }
    __device__ void skip(int n) { s += (unsigned long long)n * GAMMA; }
};

struct GenArgs {
    int B;
    unsigned long long seed;
    long long first_index;
    int gx, gy, gz;
    const int* num_cubes;
    int method, link_type, flags;  // flags: bit 0 keep parallel members, bit 1 no pin supports
    double len_lo, len_hi;
    double frange[6];
    int nforce_lo, nforce_hi;
    const double* mtypes;
    int n_types;
    int nJ_max, nM_max;
    double* xyz;
    int* conn;
    double *E, *A, *rho;
    unsigned char* cbits;
    double* loads;
    int *nJ, *nM, *n_free;
    int* status;  // [0] regenerated attempts (sum), [1] != 0: a truss did not fit nJ_max / nM_max
};

// the 24 candidate members of a cube in the reference's order (generate.py:208-229): slots 2 f, 2 f + 1 = the two
// diagonals of face f, slots 12 .. 23 = the edges; local vertex v sits at (v & 1, (v >> 1) & 1, (v >> 2) & 1)
__device__ const unsigned char SLOT_A[24] = {0, 1, 1, 3, 3, 2, 2, 0, 4, 5, 0, 1, 4, 5, 6, 4, 0, 0, 1, 2, 0, 1, 2, 3};
__device__ const unsigned char SLOT_B[24] = {5, 4, 7, 5, 6, 7, 4, 6, 7, 6, 3, 2, 5, 7, 7, 6, 1, 2, 3, 3, 4, 5, 6, 7};

// FILL = false: sizes only (nJ[b], nM[b], n_free[b]); the same draws, no array writes.
template <bool FILL>
__global__ __launch_bounds__(64 * WPB) void trs_cubegen_kernel(const GenArgs a) {
    extern __shared__ unsigned char lds_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x * WPB + wave;
    if (b >= a.B) return;
    const int gx = a.gx, gy = a.gy, gz = a.gz, ncell = gx * gy * gz;
    const int vx = gx + 1, vy = gy + 1, nvert = vx * vy * (gz + 1);
    // per-wave LDS slice
    const size_t per_wave = ((size_t)ncell + 2 * (size_t)nvert + 2 * (size_t)(ncell + 8) + 4 * (size_t)nvert +
                             3 * (size_t)nvert + 16 + 15) / 16 * 16;
    unsigned char* base = lds_raw + (size_t)wave * per_wave;
    unsigned* linkbits = reinterpret_cast<unsigned*>(base);                  // [nvert] directions already linked from a vertex
    short* vertex_id = reinterpret_cast<short*>(base + 4 * (size_t)nvert);   // [nvert] joint id or -1
    short* frontier = vertex_id + nvert;                                     // [ncell + 8]
    unsigned char* cell_state = reinterpret_cast<unsigned char*>(frontier + ncell + 8);  // [ncell] 0 free, 1 pending, 2 used
    unsigned char* jxyz = cell_state + ncell;                                // [nvert][3] grid coordinates of a joint
    short* nbuf = reinterpret_cast<short*>(base + per_wave - 16);            // [8] the popped cell's free neighbours

    const int num_cube = a.num_cubes[b];
    const bool allow_parallel = (a.flags & 1) != 0, no_pin = (a.flags & 2) != 0;
    const int pa = lane < 24 ? SLOT_A[lane] : 0, pb = lane < 24 ? SLOT_B[lane] : 0;
    // direction bit of candidate (pa -> pb): (dx + 1) + 3 (dy + 1) + 9 (dz + 1)
    const int dirbit = ((pb & 1) - (pa & 1) + 1) + 3 * (((pb >> 1) & 1) - ((pa >> 1) & 1) + 1) +
                       9 * (((pb >> 2) & 1) - ((pa >> 2) & 1) + 1);
    double* XYZ = a.xyz + (size_t)b * 3 * a.nJ_max;
    double* F = a.loads + (size_t)b * 3 * a.nJ_max;
    unsigned char* CB = a.cbits + (size_t)b * a.nJ_max;
    int* CN = a.conn + (size_t)b * 2 * a.nM_max;

    Rng rng;
    rng.s = mix64(mix64(a.seed + 0x632be59bd9b4e019ULL) ^ mix64((unsigned long long)a.first_index + (unsigned long long)b + 1ULL));
    int retries = 0, n_joint = 0, n_member = 0, n_pin = 0;
    bool overflow = false;
    for (;;) {  // attempts
        double len[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) len[k] = rng.uniform(a.len_lo, a.len_hi);
        for (int i = lane; i < ncell; i += 64) cell_state[i] = 0;
        for (int i = lane; i < nvert; i += 64) {
            vertex_id[i] = -1;
            linkbits[i] = 0u;
        }
        if constexpr (FILL) {  // no load anywhere yet (the selected joints are written further down, by lane 0)
            for (int i = lane; i < 3 * a.nJ_max; i += 64) F[i] = 0.0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        }
        __builtin_amdgcn_wave_barrier();
        int front = 0, back = 0, n_cube = 0, min_z = gz + 1;
        n_joint = n_member = 0;
        const int start = rng.below(ncell);
        frontier[0] = (short)start;
        cell_state[start] = 1;
        back = 1;
        __builtin_amdgcn_wave_barrier();
        while (n_cube < num_cube && front < back && !overflow) {
            bool take_back;
            if (a.method == 0) take_back = true;
            else if (a.method == 1) take_back = false;
            else take_back = rng.unit() <= 0.5;
            const int cell = take_back ? frontier[--back] : frontier[front++];
            cell_state[cell] = 2;
            const int cx = cell % gx, cy = (cell / gx) % gy, cz = cell / (gx * gy);
            __builtin_amdgcn_wave_barrier();
            // free neighbours in the order -x +x -y +y -z +z, then a Fisher-Yates shuffle (generate.py:252-262)
            int nc = -1;
            if (lane < 6) {
                const int axis = lane >> 1, sgn = (lane & 1) ? 1 : -1;
                const int x = cx + (axis == 0 ? sgn : 0), y = cy + (axis == 1 ? sgn : 0), z = cz + (axis == 2 ? sgn : 0);
                if (x >= 0 && x < gx && y >= 0 && y < gy && z >= 0 && z < gz) {
                    const int c = (z * gy + y) * gx + x;
                    if (cell_state[c] == 0) nc = c;
                }
            }
            const unsigned long long nmask = __ballot(nc >= 0);
            const int nnb = __popcll(nmask);
            if (nc >= 0) nbuf[__popcll(nmask & ((1ull << lane) - 1))] = (short)nc;
            __builtin_amdgcn_wave_barrier();
            for (int i = nnb - 1; i > 0; --i) {  // (wave-uniform: every lane makes the same swaps)
                const int j = rng.below(i + 1);
                const short ti = nbuf[i], tj = nbuf[j];
                __builtin_amdgcn_wave_barrier();
                nbuf[i] = tj;
                nbuf[j] = ti;
                __builtin_amdgcn_wave_barrier();
            }
            if (lane < nnb) {
                const int c = nbuf[lane];
                frontier[back + lane] = (short)c;
                cell_state[c] = 1;
            }
            back += nnb;
            // the cube's vertices -> joint ids in first-seen order (generate.py:168-184)
            int id = -1, vi = 0, vz = 0;
            bool fresh = false;
            if (lane < 8) {
                const int x = cx + (lane & 1), y = cy + ((lane >> 1) & 1);
                vz = cz + ((lane >> 2) & 1);
                vi = (vz * vy + y) * vx + x;
                id = vertex_id[vi];
                fresh = id < 0;
            }
            const unsigned long long vmask = __ballot(fresh);
            const int n_new = __popcll(vmask);
            if (n_joint + n_new > a.nJ_max) {
                overflow = true;
                break;
            }
            if (fresh) {
                id = n_joint + __popcll(vmask & ((1ull << lane) - 1));
                vertex_id[vi] = (short)id;
                jxyz[3 * id] = (unsigned char)(cx + (lane & 1));
                jxyz[3 * id + 1] = (unsigned char)(cy + ((lane >> 1) & 1));
                jxyz[3 * id + 2] = (unsigned char)vz;
            }
            n_joint += n_new;
            if (vmask & 0x0full) min_z = min(min_z, cz);
            else if (vmask) min_z = min(min_z, cz + 1);
            // candidate members: the six link choices are consecutive draws -> lane f computes choice f
            int choice = a.link_type;
            if (a.link_type == 3) {
                choice = (int)(rng.unit_at(lane < 6 ? lane : 0) * 3);
                rng.skip(6);
            }
            const int face_choice = __shfl(choice, lane < 12 ? (lane >> 1) : 0);
            bool valid = lane < 24;
            if (lane < 12) valid = (lane & 1) == 0 ? (face_choice == 0 || face_choice == 2) : (face_choice == 1 || face_choice == 2);
            const int ja = __shfl(id, pa), jb = __shfl(id, pb);
            const int via = __shfl(vi, pa);
            bool accepted = valid;
            if (valid && !allow_parallel) {  // the ordered pair (ja, jb) = (vertex of ja, direction to jb): linked once
                const unsigned bit = 1u << dirbit;
                accepted = (atomicOr(&linkbits[via], bit) & bit) == 0u;
            }
            const unsigned long long mmask = __ballot(accepted);
            const int n_add = __popcll(mmask);
            if (n_member + n_add > a.nM_max) {
                overflow = true;
                break;
            }
            if constexpr (FILL) {
                if (accepted) {
                    const int m = n_member + __popcll(mmask & ((1ull << lane) - 1));
                    CN[2 * m] = ja;
                    CN[2 * m + 1] = jb;
                }
            }
            n_member += n_add;
            ++n_cube;
            __builtin_amdgcn_wave_barrier();
        }
        if (overflow) break;
        // supports: the joints of the lowest occupied layer (generate.py:288-298)
        n_pin = 0;
        if (!no_pin) {
            for (int j0 = 0; j0 < n_joint; j0 += 64) {
                const int j = j0 + lane;
                n_pin += __popcll(__ballot(j < n_joint && jxyz[3 * j + 2] == min_z));
            }
        }
        // counting test (truss.py:158-164): too few members or supports -> draw again, the stream continues
        if (!no_pin && (3 * n_pin < 6 || n_member + 3 * n_pin < 3 * n_joint)) {
            ++retries;
            continue;
        }
        if constexpr (FILL) {
            for (int j = lane; j < a.nJ_max; j += 64) {
                double x = 0.0, y = 0.0, z = 0.0;
                unsigned char cb = 0;
                if (j < n_joint) {
                    x = (double)jxyz[3 * j] * len[0];
                    y = (double)jxyz[3 * j + 1] * len[1];
                    z = (double)((int)jxyz[3 * j + 2] - min_z) * len[2];
                    cb = (!no_pin && jxyz[3 * j + 2] == min_z) ? 7 : 0;
                }
                XYZ[3 * j] = x;
                XYZ[3 * j + 1] = y;
                XYZ[3 * j + 2] = z;
                CB[j] = cb;
            }
        }
        // loads on unsupported joints: selection sampling in joint order (generate.py:318-328); serial, the
        // number of draws depends on the acceptances
        const int n_free_joint = n_joint - n_pin;
        if (n_free_joint > 0) {
            int lo = a.nforce_lo < 1 ? 1 : a.nforce_lo;
            const int hi = (a.nforce_hi < 0 || a.nforce_hi > n_free_joint) ? n_free_joint : a.nforce_hi;
            if (lo > hi) lo = hi;
            int need = lo + rng.below(hi - lo + 1), seen = 0;
            for (int j = 0; j < n_joint && need > 0; ++j) {
                if (!no_pin && jxyz[3 * j + 2] == min_z) continue;
                if (rng.unit() * (n_free_joint - seen) < need) {
                    double f[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) f[k] = rng.uniform(a.frange[2 * k], a.frange[2 * k + 1]);
                    if constexpr (FILL) {
                        if (lane == 0) {
                            F[3 * j] = f[0];
                            F[3 * j + 1] = f[1];
                            F[3 * j + 2] = f[2];
                        }
                    }
                    --need;
                }
                ++seen;
            }
        }
        // member types: one draw per member, index known (generate.py:330-336)
        if constexpr (FILL) {
            double* Eb = a.E + (size_t)b * a.nM_max;
            double* Ab = a.A + (size_t)b * a.nM_max;
            double* Rb = a.rho + (size_t)b * a.nM_max;
            for (int m = lane; m < a.nM_max; m += 64) {
                if (m < n_member) {
                    const double* t = a.mtypes + 3 * (int)(rng.unit_at(m) * a.n_types);
                    Ab[m] = t[0];
                    Eb[m] = t[1];
                    Rb[m] = t[2];
                } else {
                    CN[2 * m] = 0;
                    CN[2 * m + 1] = 0;
                    Ab[m] = 1.0;
                    Eb[m] = 1.0;
                    Rb[m] = 0.0;
                }
            }
        }
        rng.skip(n_member);
        break;
    }
    if (lane == 0) {
        if (overflow) {
            atomicExch(&a.status[1], 1);
        } else {
            a.nJ[b] = n_joint;
            a.nM[b] = n_member;
            if (a.n_free != nullptr) a.n_free[b] = 3 * (n_joint - n_pin);
        }
        if (retries) atomicAdd(&a.status[0], retries);
    }
}

}  // namespace

extern "C" int trs_cubegen_dev_launch(int B, unsigned long long seed, int gx, int gy, int gz, const int* num_cubes,
                                      int method, int link_type, int flags, double len_lo, double len_hi,
                                      const double* force_range, int nforce_lo, int nforce_hi, const double* mtypes,
                                      int n_types, int nJ_max, int nM_max, double* xyz, int* conn, double* E, double* A,
                                      double* rho, unsigned char* cbits, double* loads, int* nJ, int* nM, int* n_free,
                                      int* status, long long first_index, hipStream_t stream) {
    if (B <= 0) return 0;
    if (gx <= 0 || gy <= 0 || gz <= 0 || gx > 254 || gy > 254 || gz > 254 || n_types <= 0 || !force_range || !mtypes ||
        !num_cubes || !nJ || !nM || !status)
        return (int)hipErrorInvalidValue;
    const long long ncell = (long long)gx * gy * gz, nvert = (long long)(gx + 1) * (gy + 1) * (gz + 1);
    if (nvert >= 32768) return (int)hipErrorInvalidValue;  // joint ids and cells are 16-bit in the LDS tables
    const size_t per_wave = ((size_t)ncell + 2 * (size_t)nvert + 2 * (size_t)(ncell + 8) + 4 * (size_t)nvert +
                             3 * (size_t)nvert + 16 + 15) / 16 * 16;
    const size_t lds = per_wave * WPB;
    if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
    GenArgs a;
    a.B = B; a.seed = seed; a.first_index = first_index; a.gx = gx; a.gy = gy; a.gz = gz; a.num_cubes = num_cubes;
    a.method = method; a.link_type = link_type; a.flags = flags; a.len_lo = len_lo; a.len_hi = len_hi;
    for (int k = 0; k < 6; ++k) a.frange[k] = force_range[k];
    a.nforce_lo = nforce_lo; a.nforce_hi = nforce_hi; a.mtypes = mtypes; a.n_types = n_types;
    a.nJ_max = nJ_max; a.nM_max = nM_max; a.xyz = xyz; a.conn = conn; a.E = E; a.A = A; a.rho = rho; a.cbits = cbits;
    a.loads = loads; a.nJ = nJ; a.nM = nM; a.n_free = n_free; a.status = status;
    static const int lds_limit_set =
        (int)hipFuncSetAttribute(reinterpret_cast<const void*>(trs_cubegen_kernel<true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) |
        (int)hipFuncSetAttribute(reinterpret_cast<const void*>(trs_cubegen_kernel<false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)lds_limit_set;
    const dim3 grid((B + WPB - 1) / WPB), block(64 * WPB);
    if (xyz != nullptr)
        hipLaunchKernelGGL(trs_cubegen_kernel<true>, grid, block, lds, stream, a);
    else
        hipLaunchKernelGGL(trs_cubegen_kernel<false>, grid, block, lds, stream, a);
    return (int)hipGetLastError();
}
