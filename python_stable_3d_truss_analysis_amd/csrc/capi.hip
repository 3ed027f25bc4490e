// extern "C" entry points declared in include/trs_solver.h: argument checks + kernel launches.
#include <hip/hip_runtime.h>
#include <string.h>
#include "../../include/trs_solver.h"
#include "trs_common.h"

extern "C" {
int trs_dofmap_launch(int, int, const uint8_t*, const int*, int*, int*, hipStream_t);
int trs_assemble_launch(int, int, int, const double*, const TrsMembers*, const double*, const int*, const int*,
                        const int*, const int*, int, size_t, int, double*, int, void*, int*, double*, int, hipStream_t);
size_t trs_assemble_work_bytes(int, int, int);
int trs_potrf_launch(int, const int*, int, size_t, int, double*, int*, const int*, const void*, double*, int, int, int,
                     hipStream_t);
int trs_potrs_launch(int, const int*, int, size_t, int, const double*, double*, int, const int*, int, hipStream_t);
int trs_recover_launch(int, int, int, const double*, const TrsMembers*, const double*, const int*, const int*,
                       const int*, const double*, int, double*, double*, double*, const int*, int, hipStream_t,
                       const long long*, int, int, const int*, int*);
int trs_joint_order_launch(int, int, int, const double*, const void*, const unsigned char*, const double*, const int*,
                           const int*, int*, int*, int*, double*, void*, unsigned char*, double*, int, hipStream_t,
                           const long long*, int, int, const double*, const double*, double*, double*, int*, int*, int,
                           const unsigned char*, unsigned char*);
int trs_cubegen_dev_launch(int, unsigned long long, int, int, int, const int*, int, int, int, double, double, const double*,
                           int, int, const double*, int, int, int, double*, int*, double*, double*, double*, unsigned char*,
                           double*, int*, int*, int*, int*, long long, hipStream_t);
int trs_copy_rows_launch(int, const void* const*, const size_t*, void* const*, const size_t*, const size_t*,
                         const size_t*, const int* const*, const size_t*, int* const*, int, const long long*, int, int,
                         hipStream_t);
int trs_solve_small_fits(int, int, int);
int trs_solve_small_launch(int, int, int, int, const double*, const TrsMembers*, const uint8_t*, const double*,
                           const int*, const int*, double*, double*, double*, int*, int*, int*, double, double, double*,
                           double*, double*, hipStream_t);
int trs_ga_sections_launch(int, int, int, int, int, const unsigned char*, const double*, double*, double*, double*,
                           hipStream_t);
int trs_fitness_launch(int, int, int, const double*, const int*, const double*, const double*,
                       const int*, const int*, const double*, const double*, double, double, double*,
                       double*, double*, hipStream_t);
}

namespace {
inline bool bad_slab(int ld, int slab_rows) {
    return ld < 16 + TRS_NB || ld % 16 != 0 || slab_rows < TRS_NB || slab_rows % TRS_NB != 0 ||
           ld < slab_rows + 16;
}
}  // namespace

extern "C" {

int trs_abi_version(void) { return TRS_ABI_VERSION; }

}  // extern "C" (the implementations on a member description follow; the C entry points wrap them)

namespace {

int solve_small_impl(int B, int nJ_max, int nM_max, int n_max_bound, const double* xyz, const TrsMembers& mem,
                     const uint8_t* cbits, const double* loads, const int32_t* nJ, const int32_t* nM, double* u,
                     double* f_ext, double* N, int32_t* info, int32_t* free_index, int32_t* n_free, double allow_stress,
                     double allow_displace, double* weight, double* stress_vio, double* disp_vio, void* stream) {
    if (B < 0) return (int)hipErrorInvalidValue;
    return trs_solve_small_launch(B, nJ_max, nM_max, n_max_bound, xyz, &mem, cbits, loads, nJ, nM, u, f_ext, N, info,
                                  free_index, n_free, allow_stress, allow_displace, weight, stress_vio, disp_vio,
                                  (hipStream_t)stream);
}

int assemble_impl(int B, int nJ_max, int nM_max, const double* xyz, const TrsMembers& mem, const double* loads,
                  const int32_t* free_index, const int32_t* n_free, const int32_t* nJ, const int32_t* nM, int ld,
                  int slab_rows, double* S, int flags, void* work, int32_t* env, double* uf, int ld_uf, void* stream) {
    if (B < 0 || nJ_max <= 0 || nM_max < 0 || bad_slab(ld, slab_rows) || (B > 0 && !work))
        return (int)hipErrorInvalidValue;
    if (uf != nullptr && ld_uf < slab_rows) return (int)hipErrorInvalidValue;
    // with envelope metadata the wave-per-matrix factorisation reads the load vector from uf
    if (env != nullptr && uf == nullptr && B > 0) return (int)hipErrorInvalidValue;
    if ((flags & TRS_ASM_ALL_WIDE) != 0 && (flags & (TRS_ASM_ALL_NARROW | TRS_ASM_COMPACT)) != 0)
        return (int)hipErrorInvalidValue;
    return trs_assemble_launch(B, nJ_max, nM_max, xyz, &mem, loads, free_index, n_free, nJ, nM, ld,
                               (size_t)slab_rows * ld, slab_rows, S, flags, work, env, uf, ld_uf, (hipStream_t)stream);
}

int recover_impl(int B, int nJ_max, int nM_max, const double* xyz, const TrsMembers& mem, const double* loads,
                 const int32_t* free_index, const int32_t* nJ, const int32_t* nM, const double* uf, int ld_uf, double* u,
                 double* f_ext, double* N, const int32_t* joint_out, int hints, void* stream, const int64_t* out_rows,
                 int nJ_out_max, int nM_out_max, const int32_t* info, int32_t* info_out) {
    if (B < 0 || nJ_max <= 0 || nM_max < 0) return (int)hipErrorInvalidValue;
    return trs_recover_launch(B, nJ_max, nM_max, xyz, &mem, loads, free_index, nJ, nM, uf, ld_uf, u, f_ext, N, joint_out,
                              hints & (TRS_HINT_RECOVER_UNSTAGED | TRS_HINT_RECOVER_SCAN), (hipStream_t)stream,
                              reinterpret_cast<const long long*>(out_rows), nJ_out_max, nM_out_max, info, info_out);
}

// dofmap -> assemble -> potrf -> potrs -> recover (out_rows == nullptr: trs_solve, else trs_solve_rows)
int solve_impl(int B, int nJ_max, int nM_max, int n_max_bound, const double* xyz, const TrsMembers& mem,
               const uint8_t* cbits, const double* loads, const int32_t* nJ, const int32_t* nM, int32_t* free_index,
               int32_t* n_free, int ld, int slab_rows, double* S, double* uf, int ld_uf, double* u, double* f_ext,
               double* N, int32_t* info, void* work, int32_t* env, const int32_t* joint_out, const int64_t* out_rows,
               int nJ_out_max, int nM_out_max, int32_t* info_out, int hints, void* stream) {
    if (out_rows == nullptr && (hints & TRS_HINT_NO_SMALL) == 0 && !joint_out &&
        trs_solve_small_fits(nJ_max, nM_max, n_max_bound))  // everything in one kernel
        return solve_small_impl(B, nJ_max, nM_max, n_max_bound, xyz, mem, cbits, loads, nJ, nM, u, f_ext, N, info,
                                free_index, n_free, 0.0, 0.0, nullptr, nullptr, nullptr, stream);
    if (n_max_bound > slab_rows) return (int)hipErrorInvalidValue;
    if (out_rows != nullptr && (nJ_out_max < nJ_max || nM_out_max < nM_max)) return (int)hipErrorInvalidValue;
    int rc = trs_dofmap(B, nJ_max, cbits, nJ, free_index, n_free, stream);
    if (rc) return rc;
    const int all_wide = env != nullptr && (hints & TRS_HINT_ALL_WIDE) != 0;
    const int no_wide = env != nullptr && (hints & TRS_HINT_NO_WIDE) != 0 && !all_wide;
    const int compact = env != nullptr && (hints & TRS_HINT_COMPACT) != 0 && !all_wide;
    const int fused = (hints & TRS_HINT_SEPARATE_STAGES) == 0;
    rc = assemble_impl(B, nJ_max, nM_max, xyz, mem, loads, free_index, n_free, nJ, nM, ld, slab_rows, S,
                       (no_wide ? TRS_ASM_ALL_NARROW : 0) | (compact ? TRS_ASM_COMPACT : 0) |
                           (all_wide ? TRS_ASM_ALL_WIDE : 0) | ((hints & TRS_HINT_ALL_TILES) ? TRS_ASM_ALL_TILES : 0),
                       work, env, uf, ld_uf, stream);
    if (rc) return rc;
    rc = trs_potrf_batched(B, n_free, ld, slab_rows, S, info, env, work, uf, ld_uf,
                           (no_wide ? TRS_HINT_NO_WIDE : 0) | (compact ? TRS_HINT_COMPACT : 0) |
                               (fused ? 0 : TRS_HINT_SEPARATE_STAGES), stream);
    if (rc) return rc;
    rc = trs_potrs_batched(B, n_free, ld, slab_rows, S, uf, ld_uf, env,
                           no_wide ? (TRS_HINT_NO_WIDE | (fused && slab_rows <= 1024 ? TRS_HINT_SUBSTITUTED : 0)) : 0,
                           stream);
    if (rc) return rc;
    if (out_rows != nullptr && info_out != nullptr && info == nullptr) return (int)hipErrorInvalidValue;
    return recover_impl(B, nJ_max, nM_max, xyz, mem, loads, free_index, nJ, nM, uf, ld_uf, u, f_ext, N, joint_out, hints,
                        stream, out_rows, nJ_out_max, nM_out_max, info, info_out);
}

}  // namespace

extern "C" {

int trs_solve_small(int B, int nJ_max, int nM_max, int n_max_bound, const double* xyz, const int32_t* conn,
                    const double* E, const double* A, const uint8_t* cbits, const double* loads,
                    const int32_t* nJ, const int32_t* nM, double* u, double* f_ext, double* N, int32_t* info,
                    int32_t* free_index, int32_t* n_free, const double* rho, double allow_stress,
                    double allow_displace, double* weight, double* stress_vio, double* disp_vio,
                    void* stream) {
    return solve_small_impl(B, nJ_max, nM_max, n_max_bound, xyz, trs_members_general(conn, E, A, rho), cbits, loads, nJ,
                            nM, u, f_ext, N, info, free_index, n_free, allow_stress, allow_displace, weight, stress_vio,
                            disp_vio, stream);
}

int trs_solve_small_tab(int B, int nJ_max, int nM_max, int n_max_bound, const double* xyz, const uint16_t* conn16,
                        const uint8_t* type_idx, const double* types, const uint8_t* cbits, const double* loads,
                        const int32_t* nJ, const int32_t* nM, double* u, double* f_ext, double* N, int32_t* info,
                        int32_t* free_index, int32_t* n_free, double allow_stress, double allow_displace, double* weight,
                        double* stress_vio, double* disp_vio, void* stream) {
    if (B > 0 && (!conn16 || !type_idx || !types)) return (int)hipErrorInvalidValue;
    return solve_small_impl(B, nJ_max, nM_max, n_max_bound, xyz, trs_members_table(conn16, type_idx, types), cbits, loads,
                            nJ, nM, u, f_ext, N, info, free_index, n_free, allow_stress, allow_displace, weight,
                            stress_vio, disp_vio, stream);
}

int trs_slab_rows(int n_max) { return trs_round_up(n_max < 1 ? 1 : n_max, TRS_NB); }

int trs_slab_ld(int n_max) { return trs_slab_rows(n_max) + 16; }

int trs_env_ints(int n_max) { return trs_env_stride(trs_slab_rows(n_max)); }

int trs_dofmap(int B, int nJ_max, const uint8_t* cbits, const int32_t* nJ, int32_t* free_index,
               int32_t* n_free, void* stream) {
    if (B < 0 || nJ_max <= 0) return (int)hipErrorInvalidValue;
    return trs_dofmap_launch(B, nJ_max, cbits, nJ, free_index, n_free, (hipStream_t)stream);
}

int trs_assemble(int B, int nJ_max, int nM_max, const double* xyz, const int32_t* conn,
                 const double* E, const double* A, const double* loads, const int32_t* free_index,
                 const int32_t* n_free, const int32_t* nJ, const int32_t* nM, int ld, int slab_rows,
                 double* S, int flags, void* work, int32_t* env, double* uf, int ld_uf, void* stream) {
    return assemble_impl(B, nJ_max, nM_max, xyz, trs_members_general(conn, E, A), loads, free_index, n_free, nJ, nM, ld,
                         slab_rows, S, flags, work, env, uf, ld_uf, stream);
}

int trs_assemble_tab(int B, int nJ_max, int nM_max, const double* xyz, const uint16_t* conn16, const uint8_t* type_idx,
                     const double* types, const double* loads, const int32_t* free_index, const int32_t* n_free,
                     const int32_t* nJ, const int32_t* nM, int ld, int slab_rows, double* S, int flags, void* work,
                     int32_t* env, double* uf, int ld_uf, void* stream) {
    if (B > 0 && (!conn16 || !type_idx || !types)) return (int)hipErrorInvalidValue;
    return assemble_impl(B, nJ_max, nM_max, xyz, trs_members_table(conn16, type_idx, types), loads, free_index, n_free,
                         nJ, nM, ld, slab_rows, S, flags, work, env, uf, ld_uf, stream);
}

int trs_potrf_batched(int B, const int32_t* n_free, int ld, int slab_rows, double* S, int32_t* info,
                      const int32_t* env, const void* work, double* uf, int ld_uf, int hints, void* stream) {
    if (B < 0 || bad_slab(ld, slab_rows) || (B > 0 && (uf == nullptr || ld_uf < slab_rows)))
        return (int)hipErrorInvalidValue;
    return trs_potrf_launch(B, n_free, ld, (size_t)slab_rows * ld, slab_rows, S, info, env, work, uf, ld_uf,
                            (hints & TRS_HINT_COMPACT) != 0, hints, (hipStream_t)stream);
}

int trs_potrs_batched(int B, const int32_t* n_free, int ld, int slab_rows, const double* S, double* uf,
                      int ld_uf, const int32_t* env, int hints, void* stream) {
    if (B < 0 || bad_slab(ld, slab_rows) || ld_uf < slab_rows) return (int)hipErrorInvalidValue;
    return trs_potrs_launch(B, n_free, ld, (size_t)slab_rows * ld, slab_rows, S, uf, ld_uf, env, hints,
                            (hipStream_t)stream);
}

int trs_recover(int B, int nJ_max, int nM_max, const double* xyz, const int32_t* conn, const double* E,
                const double* A, const double* loads, const int32_t* free_index, const int32_t* nJ,
                const int32_t* nM, const double* uf, int ld_uf, double* u, double* f_ext, double* N,
                const int32_t* joint_out, int hints, void* stream) {
    return recover_impl(B, nJ_max, nM_max, xyz, trs_members_general(conn, E, A), loads, free_index, nJ, nM, uf, ld_uf, u,
                        f_ext, N, joint_out, hints, stream, nullptr, 0, 0, nullptr, nullptr);
}

int trs_recover_tab(int B, int nJ_max, int nM_max, const double* xyz, const uint16_t* conn16, const uint8_t* type_idx,
                    const double* types, const double* loads, const int32_t* free_index, const int32_t* nJ,
                    const int32_t* nM, const double* uf, int ld_uf, double* u, double* f_ext, double* N,
                    const int32_t* joint_out, int hints, void* stream) {
    if (B > 0 && (!conn16 || !type_idx || !types)) return (int)hipErrorInvalidValue;
    return recover_impl(B, nJ_max, nM_max, xyz, trs_members_table(conn16, type_idx, types), loads, free_index, nJ, nM, uf,
                        ld_uf, u, f_ext, N, joint_out, hints, stream, nullptr, 0, 0, nullptr, nullptr);
}

int trs_recover_rows(int B, int nJ_max, int nM_max, const double* xyz, const int32_t* conn, const double* E,
                     const double* A, const double* loads, const int32_t* free_index, const int32_t* nJ,
                     const int32_t* nM, const double* uf, int ld_uf, const int32_t* joint_out, const int32_t* info,
                     const int64_t* out_rows, int nJ_out_max, int nM_out_max, double* u, double* f_ext, double* N,
                     int32_t* info_out, int hints, void* stream) {
    if (out_rows == nullptr || nJ_out_max < nJ_max || nM_out_max < nM_max || (info_out != nullptr && info == nullptr))
        return (int)hipErrorInvalidValue;
    return recover_impl(B, nJ_max, nM_max, xyz, trs_members_general(conn, E, A), loads, free_index, nJ, nM, uf, ld_uf, u,
                        f_ext, N, joint_out, hints, stream, out_rows, nJ_out_max, nM_out_max, info, info_out);
}

int trs_recover_rows_tab(int B, int nJ_max, int nM_max, const double* xyz, const uint16_t* conn16,
                         const uint8_t* type_idx, const double* types, const double* loads, const int32_t* free_index,
                         const int32_t* nJ, const int32_t* nM, const double* uf, int ld_uf, const int32_t* joint_out,
                         const int32_t* info, const int64_t* out_rows, int nJ_out_max, int nM_out_max, double* u,
                         double* f_ext, double* N, int32_t* info_out, int hints, void* stream) {
    if (out_rows == nullptr || nJ_out_max < nJ_max || nM_out_max < nM_max || (info_out != nullptr && info == nullptr) ||
        (B > 0 && (!conn16 || !type_idx || !types)))
        return (int)hipErrorInvalidValue;
    return recover_impl(B, nJ_max, nM_max, xyz, trs_members_table(conn16, type_idx, types), loads, free_index, nJ, nM, uf,
                        ld_uf, u, f_ext, N, joint_out, hints, stream, out_rows, nJ_out_max, nM_out_max, info, info_out);
}

int trs_fitness(int B, int nJ_max, int nM_max, const double* xyz, const int32_t* conn, const double* A,
                const double* rho, const int32_t* nJ, const int32_t* nM, const double* u,
                const double* N, double allow_stress, double allow_displace, double* weight,
                double* stress_vio, double* disp_vio, void* stream) {
    if (B < 0 || nJ_max <= 0 || nM_max < 0) return (int)hipErrorInvalidValue;
    return trs_fitness_launch(B, nJ_max, nM_max, xyz, conn, A, rho, nJ, nM, u, N, allow_stress,
                              allow_displace, weight, stress_vio, disp_vio, (hipStream_t)stream);
}

int trs_ga_sections(int B, int nM_max, int count, int n_member, int n_type, const uint8_t* genes,
                    const double* type_table, double* A, double* E, double* rho, void* stream) {
    if (B < 0 || nM_max < 0 || count < 0 || count > B || n_member < 0 || n_member > nM_max || n_type < 1 || n_type > 256)
        return (int)hipErrorInvalidValue;
    return trs_ga_sections_launch(B, nM_max, count, n_member, n_type, genes, type_table, A, E, rho, (hipStream_t)stream);
}

int trs_cubegen_dev(int B, uint64_t seed, int gx, int gy, int gz, const int32_t* num_cubes, int method, int link_type,
                    int flags, double len_lo, double len_hi, const double* force_range, int nforce_lo, int nforce_hi,
                    const double* mtypes, int n_types, int nJ_max, int nM_max, double* xyz, int32_t* conn, double* E,
                    double* A, double* rho, uint8_t* cbits, double* loads, int32_t* nJ, int32_t* nM, int32_t* n_free,
                    int32_t* status, int64_t first_index, void* stream) {
    if (B < 0 || nJ_max <= 0 || nM_max <= 0 || method < 0 || method > 2 || link_type < 0 || link_type > 3)
        return (int)hipErrorInvalidValue;
    if (xyz != nullptr && (!conn || !E || !A || !rho || !cbits || !loads)) return (int)hipErrorInvalidValue;
    return trs_cubegen_dev_launch(B, (unsigned long long)seed, gx, gy, gz, num_cubes, method, link_type, flags, len_lo,
                                  len_hi, force_range, nforce_lo, nforce_hi, mtypes, n_types, nJ_max, nM_max, xyz, conn,
                                  E, A, rho, cbits, loads, nJ, nM, n_free, status, (long long)first_index,
                                  (hipStream_t)stream);
}

int trs_copy_rows(int nfields, const void* const* src, const size_t* src_pitch, void* const* dst,
                  const size_t* dst_pitch, const size_t* width, const size_t* fill_to, const int32_t* const* counts,
                  const size_t* elem, int32_t* const* live, int count, const int64_t* rows, int scatter,
                  int max_blocks, void* stream) {
    if (nfields < 0 || count < 0 || max_blocks < 0) return (int)hipErrorInvalidValue;
    if (nfields > 0 && count > 0 && (!src || !src_pitch || !dst || !dst_pitch || !width || !rows))
        return (int)hipErrorInvalidValue;
    return trs_copy_rows_launch(nfields, src, src_pitch, dst, dst_pitch, width, fill_to, counts, elem, live, count,
                                reinterpret_cast<const long long*>(rows), scatter, max_blocks, (hipStream_t)stream);
}

int trs_stream_create_masked(const uint32_t* cu_mask, int n_words, void** stream) {
    if (!cu_mask || n_words <= 0 || !stream) return (int)hipErrorInvalidValue;
    bool any = false;
    for (int w = 0; w < n_words; ++w) any = any || cu_mask[w] != 0u;
    if (!any) return (int)hipErrorInvalidValue;   // a stream with no compute unit would never run anything
    hipStream_t s = nullptr;
    hipError_t rc = hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, cu_mask);
    if (rc != hipSuccess) return (int)rc;
    *stream = (void*)s;
    return 0;
}

int trs_stream_destroy(void* stream) {
    if (!stream) return (int)hipErrorInvalidValue;
    return (int)hipStreamDestroy((hipStream_t)stream);
}

}  // extern "C"

namespace {

int joint_order_impl(int B, int nJ_max, int nM_max, const double* xyz, const void* conn, int conn16, const uint8_t* cbits,
                     const double* loads, const int32_t* nJ, const int32_t* nM, int32_t* perm, int32_t* choice,
                     int32_t* reach, double* xyz_out, void* conn_out, uint8_t* cbits_out, double* loads_out, int effort,
                     void* stream) {
    if (B < 0 || nJ_max <= 0 || nM_max < 0 || (B > 0 && perm == nullptr)) return (int)hipErrorInvalidValue;
    const int outs = (xyz_out != nullptr) + (conn_out != nullptr) + (cbits_out != nullptr) + (loads_out != nullptr);
    if (outs != 0 && (outs != 4 || loads == nullptr)) return (int)hipErrorInvalidValue;   // all four or none
    if (outs == 4 && (xyz_out == xyz || conn_out == conn || cbits_out == cbits || loads_out == loads))
        return (int)hipErrorInvalidValue;                                                  // out of place only
    return trs_joint_order_launch(B, nJ_max, nM_max, xyz, conn, cbits, loads, nJ, nM, perm, choice, reach, xyz_out,
                                  conn_out, cbits_out, loads_out, effort, (hipStream_t)stream, nullptr, 0, 0, nullptr,
                                  nullptr, nullptr, nullptr, nullptr, nullptr, conn16, nullptr, nullptr);
}

}  // namespace

extern "C" {

int trs_joint_order(int B, int nJ_max, int nM_max, const double* xyz, const int32_t* conn, const uint8_t* cbits,
                    const double* loads, const int32_t* nJ, const int32_t* nM, int32_t* perm, int32_t* choice,
                    int32_t* reach, double* xyz_out, int32_t* conn_out, uint8_t* cbits_out, double* loads_out,
                    int effort, void* stream) {
    return joint_order_impl(B, nJ_max, nM_max, xyz, conn, 0, cbits, loads, nJ, nM, perm, choice, reach, xyz_out, conn_out,
                            cbits_out, loads_out, effort, stream);
}

int trs_joint_order_tab(int B, int nJ_max, int nM_max, const double* xyz, const uint16_t* conn16, const uint8_t* cbits,
                        const double* loads, const int32_t* nJ, const int32_t* nM, int32_t* perm, int32_t* choice,
                        int32_t* reach, double* xyz_out, uint16_t* conn16_out, uint8_t* cbits_out, double* loads_out,
                        int effort, void* stream) {
    return joint_order_impl(B, nJ_max, nM_max, xyz, conn16, 1, cbits, loads, nJ, nM, perm, choice, reach, xyz_out,
                            conn16_out, cbits_out, loads_out, effort, stream);
}

int trs_joint_order_rows(int B, int nJ_max, int nM_max, const int64_t* rows, int nJ_in_max, int nM_in_max,
                         const double* xyz, const int32_t* conn, const uint8_t* cbits, const double* loads,
                         const double* E, const double* A, const int32_t* nJ, const int32_t* nM, int32_t* perm,
                         int32_t* reach, double* xyz_out, int32_t* conn_out, uint8_t* cbits_out, double* loads_out,
                         double* E_out, double* A_out, int32_t* nJ_out, int32_t* nM_out, int effort, void* stream) {
    if (B < 0 || nJ_max <= 0 || nM_max < 0 || nJ_in_max <= 0 || nM_in_max < 0) return (int)hipErrorInvalidValue;
    if (B > 0 && (!rows || !perm || !xyz_out || !conn_out || !cbits_out || !loads_out || !E_out || !A_out || !nJ_out ||
                  !nM_out || !E || !A || !loads))
        return (int)hipErrorInvalidValue;
    return trs_joint_order_launch(B, nJ_max, nM_max, xyz, conn, cbits, loads, nJ, nM, perm, nullptr, reach, xyz_out,
                                  conn_out, cbits_out, loads_out, effort, (hipStream_t)stream,
                                  reinterpret_cast<const long long*>(rows), nJ_in_max, nM_in_max, E, A, E_out, A_out,
                                  nJ_out, nM_out, 0, nullptr, nullptr);
}

int trs_joint_order_rows_tab(int B, int nJ_max, int nM_max, const int64_t* rows, int nJ_in_max, int nM_in_max,
                             const double* xyz, const uint16_t* conn16, const uint8_t* cbits, const double* loads,
                             const uint8_t* type_idx, const int32_t* nJ, const int32_t* nM, int32_t* perm,
                             int32_t* reach, double* xyz_out, uint16_t* conn16_out, uint8_t* cbits_out,
                             double* loads_out, uint8_t* type_idx_out, int32_t* nJ_out, int32_t* nM_out, int effort,
                             void* stream) {
    if (B < 0 || nJ_max <= 0 || nM_max < 0 || nJ_in_max <= 0 || nM_in_max < 0) return (int)hipErrorInvalidValue;
    if (B > 0 && (!rows || !perm || !xyz_out || !conn16_out || !cbits_out || !loads_out || !type_idx_out || !nJ_out ||
                  !nM_out || !type_idx || !loads))
        return (int)hipErrorInvalidValue;
    return trs_joint_order_launch(B, nJ_max, nM_max, xyz, conn16, cbits, loads, nJ, nM, perm, nullptr, reach, xyz_out,
                                  conn16_out, cbits_out, loads_out, effort, (hipStream_t)stream,
                                  reinterpret_cast<const long long*>(rows), nJ_in_max, nM_in_max, nullptr, nullptr,
                                  nullptr, nullptr, nJ_out, nM_out, 1, type_idx, type_idx_out);
}

int trs_solve(int B, int nJ_max, int nM_max, int n_max_bound, const double* xyz, const int32_t* conn,
              const double* E, const double* A, const uint8_t* cbits, const double* loads,
              const int32_t* nJ, const int32_t* nM, int32_t* free_index, int32_t* n_free, int ld,
              int slab_rows, double* S, double* uf, int ld_uf, double* u, double* f_ext, double* N,
              int32_t* info, void* work, int32_t* env, const int32_t* joint_out, int hints, void* stream) {
    return solve_impl(B, nJ_max, nM_max, n_max_bound, xyz, trs_members_general(conn, E, A), cbits, loads, nJ, nM,
                      free_index, n_free, ld, slab_rows, S, uf, ld_uf, u, f_ext, N, info, work, env, joint_out, nullptr, 0,
                      0, nullptr, hints, stream);
}

int trs_solve_tab(int B, int nJ_max, int nM_max, int n_max_bound, const double* xyz, const uint16_t* conn16,
                  const uint8_t* type_idx, const double* types, const uint8_t* cbits, const double* loads,
                  const int32_t* nJ, const int32_t* nM, int32_t* free_index, int32_t* n_free, int ld, int slab_rows,
                  double* S, double* uf, int ld_uf, double* u, double* f_ext, double* N, int32_t* info, void* work,
                  int32_t* env, const int32_t* joint_out, int hints, void* stream) {
    if (B > 0 && (!conn16 || !type_idx || !types)) return (int)hipErrorInvalidValue;
    return solve_impl(B, nJ_max, nM_max, n_max_bound, xyz, trs_members_table(conn16, type_idx, types), cbits, loads, nJ,
                      nM, free_index, n_free, ld, slab_rows, S, uf, ld_uf, u, f_ext, N, info, work, env, joint_out,
                      nullptr, 0, 0, nullptr, hints, stream);
}

int trs_solve_rows(int B, int nJ_max, int nM_max, int n_max_bound, const double* xyz, const int32_t* conn,
                   const double* E, const double* A, const uint8_t* cbits, const double* loads, const int32_t* nJ,
                   const int32_t* nM, int32_t* free_index, int32_t* n_free, int ld, int slab_rows, double* S,
                   double* uf, int ld_uf, double* u, double* f_ext, double* N, int32_t* info, void* work, int32_t* env,
                   const int32_t* joint_out, const int64_t* out_rows, int nJ_out_max, int nM_out_max,
                   int32_t* info_out, int hints, void* stream) {
    if (out_rows == nullptr) return (int)hipErrorInvalidValue;
    return solve_impl(B, nJ_max, nM_max, n_max_bound, xyz, trs_members_general(conn, E, A), cbits, loads, nJ, nM,
                      free_index, n_free, ld, slab_rows, S, uf, ld_uf, u, f_ext, N, info, work, env, joint_out, out_rows,
                      nJ_out_max, nM_out_max, info_out, hints, stream);
}

int trs_solve_rows_tab(int B, int nJ_max, int nM_max, int n_max_bound, const double* xyz, const uint16_t* conn16,
                       const uint8_t* type_idx, const double* types, const uint8_t* cbits, const double* loads,
                       const int32_t* nJ, const int32_t* nM, int32_t* free_index, int32_t* n_free, int ld,
                       int slab_rows, double* S, double* uf, int ld_uf, double* u, double* f_ext, double* N,
                       int32_t* info, void* work, int32_t* env, const int32_t* joint_out, const int64_t* out_rows,
                       int nJ_out_max, int nM_out_max, int32_t* info_out, int hints, void* stream) {
    if (out_rows == nullptr || (B > 0 && (!conn16 || !type_idx || !types))) return (int)hipErrorInvalidValue;
    return solve_impl(B, nJ_max, nM_max, n_max_bound, xyz, trs_members_table(conn16, type_idx, types), cbits, loads, nJ,
                      nM, free_index, n_free, ld, slab_rows, S, uf, ld_uf, u, f_ext, N, info, work, env, joint_out,
                      out_rows, nJ_out_max, nM_out_max, info_out, hints, stream);
}

}  // extern "C"
