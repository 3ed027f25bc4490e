/*
 * trs_host.h - host-side native helpers around the batched truss solver (library libtrs_host.so;
 * plain C + OpenMP over the batch, no GPU).  They feed / drain the device path of trs_solver.h with
 * the same padded batch arrays (PackedBatch in the Python package):
 *
 *   xyz [B][nJ_max][3] f64 | conn [B][nM_max][2] i32 | E, A, rho [B][nM_max] f64 |
 *   cbits [B][nJ_max] u8 (constrained-axis bits x=1,y=2,z=4) | loads [B][nJ_max][3] f64 | nJ, nM [B] i32
 *
 * Reference counterparts: slientruss3d/generate.py (random cube trusses), slientruss3d/data.py
 * (TrussHeteroDataCreator features); the joint reordering has no counterpart (it only shrinks the
 * envelope the GPU factorisation works on; results are mapped back by the caller).
 */
#ifndef TRS_HOST_H
#define TRS_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Upper bounds of joints / members for polycubes of at most max_cubes cells on a gx*gy*gz grid. */
int trs_cubegen_bounds(int gx, int gy, int gz, int max_cubes, int allow_parallel, int *nJ_max, int *nM_max);

/* Random polycube trusses (generate.py:40-246: grow a polycube, link the cube vertices, supports,
 * loads, member types; count-unstable draws are regenerated).  num_cubes[b] cells for truss b; every
 * truss has its own RNG stream derived from seed.  With xyz == NULL only nJ[b], nM[b] are produced
 * (sizes-only pass).  allow_parallel: bit 0 = keep parallel members (isAllowParallel), bit 1 = no pin
 * supports (isAddPinSupport=False: loads may then sit on any joint and the counting test is left to
 * the caller, who adds supports with an augmenter).  Returns 0, -1 when a truss does not fit nJ_max / nM_max, -2 on allocation failure. */
int trs_cubegen(int B, uint64_t seed, int gx, int gy, int gz, const int32_t *num_cubes, int method,
                int link_type, int allow_parallel, double len_lo, double len_hi,
                const double *force_range /* [3][2] */, int nforce_lo, int nforce_hi,
                const double *mtypes /* [n_types][3] = (a, e, density) */, int n_types, int nJ_max,
                int nM_max, double *xyz, int32_t *conn, double *E, double *A, double *rho,
                uint8_t *cbits, double *loads, int32_t *nJ, int32_t *nM, int64_t *retries_out,
                int64_t first_index /* global index of truss 0: the stream of truss b is keyed by
                                       (seed, first_index + b), so chunks / shards of one dataset agree */);

/* One generation's offspring of the GA (reference ga.py:173-190 UpdatePop with Crossover ga.py:162-165, Mutate
 * ga.py:167-171, GetRandomGene ga.py:136-137) on gene matrices of bytes, drawing from PYTHON's global Mersenne
 * Twister exactly as the reference's calls to random.random / sample / choice / randint / choices would:
 * state = random.getstate()[1] (624 words + position) in, the advanced state out (csrc/gaops.c).  elite
 * [nElite][nMember] in rank order, pop / out [nPop][nMember]; toCross <= toMutate <= toKeep are the cumulative
 * probabilities; counts [4] or NULL receives the offspring by crossover / mutation / kept / fresh.  Returns 0, -1
 * for shapes the caller must handle in Python (fewer than two elites, members or types; more than 256 types). */
int trs_ga_update_pop(uint32_t *state /* [625] */, int nPop, int nElite, int nMember, int nType, double toCross,
                      double toMutate, double toKeep, const uint8_t *elite, const uint8_t *pop, uint8_t *out,
                      int32_t *counts /* [4] or NULL */);

/* Team size of the OpenMP loops of this library: n > 0 sets it, the return value is the size in effect.
 * The Python loader sets it once to the CPUs the process may actually use (affinity mask and cgroup quota;
 * `generate.available_cpus`) unless OMP_NUM_THREADS is set. */
int trs_host_threads(int n);

/* Reverse Cuthill-McKee order of the joints of every truss: perm[b][k] = old id of the joint that
 * becomes joint k (identity on the padding).  Returns 0 or -2 on allocation failure. */
int trs_rcm_order(int B, int nJ_max, int nM_max, const int32_t *conn, const uint8_t *cbits,
                  const int32_t *nJ, const int32_t *nM, int32_t *perm /* [B][nJ_max] */);

/* The cheapest of several candidate joint orders per truss, priced by the 16x16-tile envelope the
 * factorisation works in (csrc/reorder.c): reverse Cuthill-McKee, its reverse, and the twelve binned
 * coordinate sweeps (six axis orders, forwards and backwards).  Never worse than trs_rcm_order; 15-35 %
 * less factorisation work on lattice-like trusses (the reference's cube trusses, generate.py:150-340).
 * perm as trs_rcm_order; choice [B] (may be NULL) receives the winning candidate's id. */
int trs_profile_order(int B, int nJ_max, int nM_max, const double *xyz, const int32_t *conn,
                      const uint8_t *cbits, const int32_t *nJ, const int32_t *nM, int32_t *perm /* [B][nJ_max] */,
                      int32_t *choice /* [B] or NULL */,
                      int effort /* 0: RCM and its reverse, 1: + the sweep along the longest extent, 2: all sweeps,
                                    3: all sweeps, RCM and its reverse only below 128 free joints or without a sweep */);

/* Reach of the row envelope below the 64 x 64 diagonal blocks of K_ff, in 16-row chunks, per truss, for the
 * numbering given or after a renumbering (what trs_assemble derives on the device).  A batch that stays at or below 24 everywhere holds
 * no matrix for the work-group kernels: see TRS_ASM_ALL_NARROW / TRS_HINT_NO_WIDE in trs_solver.h. */
int trs_envelope_reach(int B, int nJ_max, int nM_max, const int32_t *conn, const uint8_t *cbits,
                       const int32_t *nJ, const int32_t *nM,
                       const int32_t *perm /* [B][nJ_max] as trs_profile_order gives it: the reach AFTER that
                                              renumbering; NULL: the numbering as it is */,
                       int32_t *reach /* [B] */);

/* Apply a joint order out of place (members keep their order, their end joints are renumbered). */
int trs_apply_joint_order(int B, int nJ_max, int nM_max, const int32_t *perm, const int32_t *nM,
                          const double *xyz, const int32_t *conn, const uint8_t *cbits,
                          const double *loads, double *xyz_out, int32_t *conn_out, uint8_t *cbits_out,
                          double *loads_out);

/* Graph features of a solved batch of 3D trusses as float32 tensors (data.py:11-282, GetAngles
 * utils.py:105-113): joint_x [B][nJ_max][7 (+3 with a prior)], member_x [B][nM_max][8 (+1 prior)
 * (+1 regression)], joint_y [B][nJ_max][3] and member_y [B][nM_max] (regression only), weight [B].
 * u_* are [B][nJ_max][3], N_* [B][nM_max]; pass NULL for an absent prior / actual result. */
int trs_graph_features(int B, int nJ_max, int nM_max, const double *xyz, const int32_t *conn,
                       const double *A, const double *rho, const uint8_t *cbits, const double *loads,
                       const int32_t *nJ, const int32_t *nM, const double *u_act, const double *N_act,
                       const double *u_pri, const double *N_pri, double fixedArea, double forceScale,
                       double displaceScale, double positionScale, int regression, float *joint_x,
                       float *member_x, float *joint_y, float *member_y, double *weight);

/* Bulk JSON reader (truss.py:401-421 for many files at once; schema detail/combine_with_JSON.md:71-163):
 * B JSON texts -> the padded batch arrays above, plus dim[b] (2 or 3; a 2D truss is embedded
 * with z = 0 and bit 4 set in every cbits entry).  Call once with xyz == NULL for nJ[b], nM[b], dim[b]
 * (sizes-only pass), allocate to the maxima, call again to fill.  Loads below 1e-10 in every component
 * are dropped and unknown keys (the result keys of output files) are skipped, as the reference does.
 * Returns 0 or -(1000 * (index of the first bad input + 1) + code), code: 1 syntax, 2 inconsistent
 * dimension, 3 unknown / invalid support type, 4 joint id out of range, 5 does not fit the padding. */
int trs_json_pack(int B, const char *const *texts, const int64_t *lens, int nJ_max, int nM_max,
                  double *xyz, int32_t *conn, double *E, double *A, double *rho, uint8_t *cbits,
                  double *loads, int32_t *nJ, int32_t *nM, int32_t *dim);
/* Files for trs_json_pack: read natively and in parallel into one buffer per file (bufs[b], lens[b];
 * release with trs_json_free_files).  Returns 0 or -(1000 * (index of the first unreadable file + 1) + 6). */
int trs_json_read_files(int B, const char *const *paths, char **bufs, int64_t *lens);
void trs_json_free_files(int B, char **bufs);

#ifdef __cplusplus
}
#endif
#endif
