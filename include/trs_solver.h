/*
 * trs_solver.h - C ABI of the MI355X (gfx950) batched direct-stiffness truss solver.
 *
 * The reference (leo27945875/Python_Stable_3D_Truss_Analysis, slientruss3d v2.0.3) is pure
 * Python and has no FFI layer: its seam is the method Truss.Solve() (slientruss3d/truss.py:329-364).
 * The entry points below are the stages of that method, batched over B independent trusses,
 * each citing the reference lines it replaces.  INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer owned by the caller (e.g. a PyTorch-ROCm tensor's
 *    data_ptr()); the library allocates nothing.  `stream` is a hipStream_t passed as void*.
 *  - Every call only enqueues work on `stream` and returns; the return value is 0 or a
 *    hipError_t for launch/argument errors.  Numerical status is reported through `info`.
 *  - Batched arrays are padded to the batch maxima: joints to nJ_max, members to nM_max;
 *    per-truss counts are nJ[b], nM[b].  All trusses are 3D in the kernels; a 2D truss is
 *    packed with z = 0 and the z axis constrained at every joint.
 *  - DOF numbering: dof = joint*3 + axis (reference truss.py:312-314,324).
 *  - cbits[b][j]: constrained-axis bit mask of joint j (x=1, y=2, z=4): the encoding of
 *    SupportType.GetResistanceMask (reference type.py:48-74).
 *
 * Stiffness slab layout (written by trs_assemble, factored in place by trs_potrf_batched)
 *    n      = n_free[b]                  number of free DOFs of truss b
 *    n_pad  = round_up(n, 64)
 *    S[c][i], row-major, leading dimension ld (>= n_pad_max + 16, multiple of 16),
 *    one slab of slab_rows*ld doubles per truss (slab_rows >= n_pad_max).
 *      S[c][i] = K_ff[c][i]        for c < n, 16*floor(c/16) <= i < n      (upper part by 16-tiles)
 *      S[c][c] = 1                 for n <= c < n_pad                        (identity padding)
 *      S[c][n_pad] = f_f[c]        right-hand side, stored as one extra column
 *      other entries with i >= 16*floor(c/16), i < n_pad + 16 are 0; entries left of the
 *      diagonal tile are never read or written unless TRS_ASM_FULL_SYMMETRIC is given.
 *    After trs_potrf_batched:  S[c][i] = U[c][i] (i >= c) with K_ff = U^T U (U = L^T), and
 *    uf[c] = y[c], the forward-substituted right-hand side (L y = f_f).  Inside every
 *    16 x 16 diagonal tile the entries below the diagonal (c > i, same tile) hold the strictly
 *    lower part of inv(L_tile) (its diagonal is 1 / U[c][c]); trs_potrs_batched uses it.
 *
 * Compact form (narrow envelopes; opt-in per call, TRS_ASM_COMPACT + TRS_HINT_COMPACT): trusses whose envelope
 *    is narrow enough for the wave-per-matrix factorisation do NOT get their stiffness matrix written to
 *    the slab at all.  trs_assemble leaves it in the truss's share of `work` as per-tile entry lists
 *    (value + position, ~10 bytes per non-zero instead of 2 KB per 16 x 16 tile; layout
 *    TrsCompactLayout in csrc/trs_common.h) and the load vector in uf; trs_potrf_batched forms the
 *    tiles from the lists where it consumes them and writes only the factor to the slab.  K_ff then
 *    never exists in dense form in HBM.  Without the flag every matrix goes through the slab (the default:
 *    measured faster end to end, EXPERIMENTS.md Part II section 3.3 and R4.1).
 *
 * Member forms (ABI 10).  The reference describes a member as [[j0, j1], [a, e, density]] (truss.py:406-413,
 *    MemberType type.py:5-27); most workloads draw the (a, e, density) triple from a short list (the generator's
 *    memberTypes, the GA's type table, generate.py:300-316, ga.py:125-137).
 *      general form   conn int32 [B][nM_max][2], E and A (and rho where densities are needed) double [B][nM_max]:
 *                     24 (32) bytes per member.  The entry points without a suffix.
 *      table form     conn16 uint16 [B][nM_max][2] (the library's size limit is 65 535 joints per truss anyway),
 *                     type_idx uint8 [B][nM_max] and types double [n_types][3] = (a, e, density), n_types <= 256:
 *                     5 bytes per member in HBM and over PCIe.  The entry points with the suffix _tab, which take
 *                     (conn16, type_idx, types) where their twins take (conn, E, A); everything else is the same,
 *                     and so are the results, bit for bit (E A is formed as e * a from the table: the same product).
 *    trs_joint_order_tab / trs_joint_order_rows_tab read and write uint16 end joints (and copy the type indices of a
 *    bucket's rows); trs_copy_rows moves either form (it copies bytes).  The fitness, graph-feature, GA-section and
 *    generator entry points exist in the general form only.
 *
 * Streams and threads: the library keeps NO mutable process-wide state that a result depends on (the only static
 * data are "dynamic-LDS ceiling raised" markers of kernels, set once).  Everything that selects a kernel or a data
 * path is an argument of the call (flags / hints below), every kernel works on the buffers of its call only, and no
 * kernel assumes anything about what else runs on the device: CONCURRENT CALLS ON DISTINCT STREAMS WITH DISTINCT
 * BUFFERS DO NOT INTERACT, and their results are bit for bit those of the same calls issued one after the other.
 * This clause is tested: tools/repro_streams.cpp (a plain HIP program: hipMalloc, hipStream_t and this header, no
 * tensor library) runs the launch sequences of a ragged batch's size buckets on two, three and four streams, compares
 * every step truss by truss with a one-stream step and checks every joint order for being a permutation; and
 * tools/repro_families.cpp runs every OTHER kernel family beside that pipeline - trs_solve_small with the fitness
 * reductions, trs_ga_sections, trs_graph_features_packed, trs_fitness, trs_copy_rows through page-locked host memory
 * and trs_cubegen_dev on a CU-masked stream - each on a stream of its own, every output compared bit for bit with the
 * same calls on one stream (tests/test_gpu_streams.py; 200 steps per test).  Rounds 3-4 of this library violated it: a work-group race in
 * trs_joint_order's kernel (a breadth-first level counter read while faster waves already advanced it) that only
 * concurrent kernels on other streams, which pull the waves of a work-group apart, brought out - wrong joint orders,
 * stores through garbage indices into other calls' buffers, device stalls (EXPERIMENTS.md R4.9, R5.1).
 * Calls that share a buffer (one workspace used by several calls) are ordered by the stream they are issued on, as
 * for any stream-ordered resource; two host threads may call concurrently as long as their buffers are distinct.
 */
#ifndef TRS_SOLVER_H
#define TRS_SOLVER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRS_ABI_VERSION 10

/* trs_assemble flags */
#define TRS_ASM_FULL_SYMMETRIC 1 /* also write the entries left of the diagonal tile (tests); implies the slab form */
#define TRS_ASM_COMPACT 2        /* narrow-envelope matrices leave as compact entry lists in `work` instead of slab
                                    tiles (see "Compact form"); the same batch must then be factored with
                                    TRS_HINT_COMPACT.  Ignored (slab form) where the lists cannot be held: with
                                    TRS_ASM_FULL_SYMMETRIC, without uf, or for an envelope forced narrow by
                                    TRS_ASM_ALL_NARROW that reaches further than the lists are sized for */
#define TRS_ASM_ALL_NARROW 4     /* every matrix is routed to the wave-per-matrix kernels, whatever its envelope */
#define TRS_ASM_ALL_TILES 8      /* write EVERY tile of the envelope, also those that hold no entry of K_ff (by default the
                                    matrices of the wave-per-matrix kernels leave such tiles unwritten and mark them in
                                    the envelope metadata, csrc/trs_common.h `kmask`: the factorisation takes them as zeros
                                    without reading them; about 30 % of the stored tiles of a cube truss) */

#define TRS_ASM_ALL_WIDE 16      /* every matrix is routed to the work-group kernels (four waves per matrix), whatever its
                                    envelope: for batches of FEW, LARGE systems - a batch with fewer matrices than the chip
                                    has SIMDs leaves most of it idle under the wave-per-matrix kernels (bench.py
                                    `large_truss`).  Not together with TRS_ASM_ALL_NARROW / TRS_ASM_COMPACT */

/* hints of trs_potrf_batched / trs_potrs_batched / trs_solve: what the caller knows about the batch, so that
 * kernels that would find no matrix of theirs are not launched at all (each such launch costs 4-9 us).
 * A hint never changes a result when it is true; a FALSE hint leaves matrices unprocessed. */
#define TRS_HINT_NO_WIDE 1       /* trs_assemble ran with TRS_ASM_ALL_NARROW: no matrix for the work-group kernels */
#define TRS_HINT_SUBSTITUTED 2   /* trs_potrs_batched only: trs_potrf_batched ran with the fused substitution (no
                                    TRS_HINT_SEPARATE_STAGES) and no system has more than 1024 rows: nothing is left
                                    for the wave-per-matrix substitution either (a matrix with a failed pivot stays
                                    unsolved: info[b] > 0) */
/* per-call selectors (they replace the process-wide switches of ABI <= 6; none changes a result) */
#define TRS_HINT_COMPACT 4           /* trs_potrf_batched / trs_solve: the batch was (is to be) assembled with
                                        TRS_ASM_COMPACT: also launch the kernel instance that forms tiles from lists */
#define TRS_HINT_SEPARATE_STAGES 8   /* trs_potrf_batched / trs_solve: the factorising wave does NOT go on to the
                                        back substitution; trs_potrs_batched substitutes every matrix */
#define TRS_HINT_NO_SMALL 16         /* trs_solve: never the fused small-system kernel, always the staged pipeline */
#define TRS_HINT_ALL_TILES 64         /* trs_solve: assemble with TRS_ASM_ALL_TILES (the caller has seen - in the envelope
                                        metadata of an earlier solve of the same topology - that no matrix of the
                                        batch skips tiles: the mask need not be formed again) */
#define TRS_HINT_RECOVER_UNSTAGED 32 /* trs_recover / trs_solve: the path for trusses whose tables exceed a CU's LDS
                                        (u, f_ext in the output arrays; the reactions in the same fixed order, bit for
                                        bit those of the staged path), whatever the size */
#define TRS_HINT_ALL_WIDE 256         /* trs_solve / trs_solve_rows: assemble with TRS_ASM_ALL_WIDE (overrides TRS_HINT_NO_WIDE
                                        and TRS_HINT_COMPACT) */
#define TRS_HINT_RECOVER_SCAN 128    /* with TRS_HINT_RECOVER_UNSTAGED (tests): also the fall-back of that path for a truss
                                        with more than 32768 member ends at its supports - no member-end lists, a wave
                                        per constrained joint walks all members; same bits again */

int trs_abi_version(void);

/* Leading dimension / row count of the stiffness slab for a batch whose largest reduced
 * system has n_max free DOFs. */
int trs_slab_ld(int n_max);
int trs_slab_rows(int n_max);

/* Free-DOF numbering.  Replaces Truss.GetDisplacementUnknownMask (truss.py:319-326) and the
 * boolean-mask compaction of truss.py:343: free_index[b][dof] = position of the DOF in the
 * reduced system, or -1 when constrained; n_free[b] = number of free DOFs. */
int trs_dofmap(int B, int nJ_max, const uint8_t *cbits, const int32_t *nJ,
               int32_t *free_index /* [B][nJ_max*3] */, int32_t *n_free /* [B] */, void *stream);

/* Envelope metadata (optional; pass NULL everywhere to treat every matrix as dense).
 * trs_assemble derives, per truss, the row envelope of the reduced stiffness matrix at 16-row
 * granularity (first tile per row chunk, last row chunk per 64-column panel, stored extent per
 * 16-row chunk; layout in csrc/trs_common.h) and writes only the slab tiles inside it; trs_potrf_batched and
 * trs_potrs_batched skip the tiles outside it, which are exact zeros of the Cholesky factor.
 * `env` is an int32 array of B * trs_env_ints(n_max) entries. */
int trs_env_ints(int n_max);

/* Bytes of assembly workspace PER TRUSS for a batch with these maxima: member stiffness and
 * direction cosines when they do not fit the CU's LDS, and the compact entry lists of the stiffness
 * matrix (see "Compact form" above), which trs_potrf_batched reads back.  The caller passes B times
 * this many bytes as `work` to both. */
size_t trs_assemble_work_bytes(int nJ_max, int nM_max, int n_max);

/* Assembly of the reduced stiffness matrix and load vector.  Replaces Member.k/cosines/matK
 * (truss.py:56-86), Truss.GetKMatrix (truss.py:307-316), GetExternalForceVector
 * (truss.py:303-304) and the row/column elimination matK[mask,:][:,mask], vecF[mask]
 * (truss.py:343).  Deterministic: the sums follow a fixed order (sorted joint adjacency), so the
 * slab is bit-identical from run to run and between identical trusses of a batch. */
int trs_assemble(int B, int nJ_max, int nM_max,
                 const double *xyz /* [B][nJ_max][3] */, const int32_t *conn /* [B][nM_max][2] */,
                 const double *E /* [B][nM_max] */, const double *A /* [B][nM_max] */,
                 const double *loads /* [B][nJ_max][3] */, const int32_t *free_index,
                 const int32_t *n_free, const int32_t *nJ, const int32_t *nM,
                 int ld, int slab_rows, double *S /* [B][slab_rows][ld] */, int flags,
                 void *work /* B * trs_assemble_work_bytes(...) */,
                 int32_t *env /* out, B * trs_env_ints(...), or NULL */,
                 double *uf /* out [B][ld_uf]: the load vector of every truss routed to a wave-per-matrix
                               factorisation (narrow envelope); required with env, may be NULL without */,
                 int ld_uf, void *stream);

/* Batched Cholesky factorisation with fused forward substitution of the right-hand-side
 * column.  Replaces the factorisation half of np.linalg.solve (truss.py:343; LAPACK dgesv in
 * the reference, potrf here because K_ff is SPD for a stable truss).
 * info[b] = 0 on success, k > 0 when the pivot of column k (1-based) is not positive
 * (the reference raises numpy.linalg.LinAlgError for an exactly singular matrix).
 * Fused substitution (default; TRS_HINT_SEPARATE_STAGES keeps the stages apart): the wave that
 * factors a narrow-envelope matrix of at most 1024 rows goes straight on to the back substitution, while the
 * factor's last panels are still cached.  uf[b] then holds the reduced DISPLACEMENTS when this call returns,
 * bit 0x400 of the matrix's routing word in env says so, and trs_potrs_batched skips the matrix - callers who
 * run the stages in order notice nothing.  Matrices with a failed pivot, wide envelopes, the dense mode and
 * larger systems are substituted by trs_potrs_batched as before. */
int trs_potrf_batched(int B, const int32_t *n_free, int ld, int slab_rows, double *S,
                      int32_t *info /* [B] */, const int32_t *env /* or NULL */,
                      const void *work /* the buffer trs_assemble filled */,
                      double *uf /* [B][ld_uf]: out y = L^-1 f (in: f, for the trusses in compact form) */,
                      int ld_uf, int hints, void *stream);

/* Back substitution U u_f = y, in place on uf.  Replaces the solve half of np.linalg.solve
 * (truss.py:343).  In: uf[b][c] = y[c] (from trs_potrf_batched); out: uf[b][c] = reduced displacement c
 * (c < n_free[b]); entries up to n_pad are written. */
int trs_potrs_batched(int B, const int32_t *n_free, int ld, int slab_rows, const double *S,
                      double *uf /* inout [B][ld_uf] */, int ld_uf, const int32_t *env /* or NULL */,
                      int hints, void *stream);

/* Result recovery.  Replaces the displacement scatter (truss.py:342), the reactions
 * vecF[~mask] = K[~mask,:] @ u (truss.py:348-349) and the member-force loop
 * (truss.py:354-359, Member.IsTension truss.py:89-91): u and f_ext are dense [nJ_max][3],
 * N[m] is the axial force, tension positive.  f_ext holds the applied load at free DOFs and
 * the stiffness reaction K u at constrained DOFs, as the reference's `external`.
 * joint_out (or NULL): [B][nJ_max], a permutation of 0..nJ_max-1 per truss - the results of joint j are
 * written to row joint_out[b][j] of u / f_ext.  A caller who renumbered the joints for a narrower envelope
 * (trs_profile_order, trs_host.h) passes the order it applied and receives the results in the ORIGINAL
 * numbering with no extra pass (members keep their order, N is unaffected). */
int trs_recover(int B, int nJ_max, int nM_max, const double *xyz, const int32_t *conn,
                const double *E, const double *A, const double *loads,
                const int32_t *free_index, const int32_t *nJ, const int32_t *nM,
                const double *uf, int ld_uf, double *u /* [B][nJ_max][3] */,
                double *f_ext /* [B][nJ_max][3] */, double *N /* [B][nM_max] */,
                const int32_t *joint_out /* [B][nJ_max] or NULL */,
                int hints /* TRS_HINT_RECOVER_UNSTAGED (| TRS_HINT_RECOVER_SCAN) or 0 */, void *stream);

/* trs_recover with the bucket SCATTER of a ragged batch folded in (ABI 9; `batch.RaggedSolver`): the results of truss b
 * go to row out_rows[b] (int64, device) of result arrays whose rows are nJ_out_max joints / nM_out_max members wide
 * (>= nJ_max / nM_max; what lies beyond this batch's width is not touched: the caller keeps it zero), through
 * joint_out as in trs_recover; info[b] (the status trs_potrf_batched left) is copied to info_out[out_rows[b]]
 * (info_out may be NULL).  No separate copy of the bucket's result rows. */
int trs_recover_rows(int B, int nJ_max, int nM_max, const double *xyz, const int32_t *conn, const double *E,
                     const double *A, const double *loads, const int32_t *free_index, const int32_t *nJ,
                     const int32_t *nM, const double *uf, int ld_uf, const int32_t *joint_out /* or NULL */,
                     const int32_t *info, const int64_t *out_rows, int nJ_out_max, int nM_out_max, double *u,
                     double *f_ext, double *N, int32_t *info_out, int hints, void *stream);

/* Member sections of a GA population from its gene matrix (ABI 9; ga.py:125-137 `TranslateGene` /
 * `SetMemberTypesByGene`: locus i of a gene selects the type of member i): for individual b < count and member
 * m < n_member   (A, E, rho)[b][m] = type_table[genes[b][m]]   (type_table [n_type][3] = a, e, density, device;
 * genes uint8 [count][n_member], device); rows b >= count and members m >= n_member of the padded arrays get type 0
 * (solved and ignored); a locus >= n_type gives NaN sections, which the solve reports in info.  One launch in
 * place of the gather + three strided copies of a tensor library. */
int trs_ga_sections(int B, int nM_max, int count, int n_member, int n_type /* 1..256 */, const uint8_t *genes,
                    const double *type_table, double *A /* [B][nM_max] */, double *E, double *rho, void *stream);

/* Constraint reductions of the GA fitness (truss.py:166-168,429-462; ga.py:139-149):
 *   weight[b]   = sum_m A*L*rho
 *   stress_vio[b] = sum_m max(|N|/A - allow_stress, 0) over members with |N| >= 1e-10
 *   disp_vio[b]   = sum_j max(||u_j|| - allow_displace, 0) over joints with a component >= 1e-10 */
int trs_fitness(int B, int nJ_max, int nM_max, const double *xyz, const int32_t *conn,
                const double *A, const double *rho, const int32_t *nJ, const int32_t *nM,
                const double *u, const double *N, double allow_stress, double allow_displace,
                double *weight /* [B] */, double *stress_vio /* [B] */, double *disp_vio /* [B] */,
                void *stream);

/* Fused path for SMALL trusses: the whole of Truss.Solve() (truss.py:329-364: dofmap, assembly,
 * Cholesky with the load vector riding along, back substitution, recovery) in ONE kernel, one
 * work-group per truss, the stiffness matrix resident in that work-group's LDS (it never reaches HBM).
 * For batches whose largest reduced system has n_max_bound <= 128 free DOFs and whose tables fit a CU's
 * LDS: trs_solve_small_fits() says whether a batch shape qualifies (1) or not (0); trs_solve_small
 * returns hipErrorInvalidValue for a shape that does not.  Covers bar-6 ... bar-120, cube-7 and every
 * individual of the reference's GA (ga.py:139-149).
 *   free_index, n_free : optional outputs of the numbering stage (NULL = not wanted)
 *   weight != NULL     : also the GA reductions of trs_fitness (rho, allow_* as there)
 *   info[b]            : 0, k > 0 (pivot k not positive) or -1 (n_free[b] > n_max_bound: nothing solved)
 * Bit-reproducible (fixed summation orders, no floating-point atomics). */
int trs_solve_small_fits(int nJ_max, int nM_max, int n_max_bound);
int trs_solve_small(int B, int nJ_max, int nM_max, int n_max_bound,
                    const double *xyz, const int32_t *conn, const double *E, const double *A,
                    const uint8_t *cbits, const double *loads, const int32_t *nJ, const int32_t *nM,
                    double *u /* [B][nJ_max][3] */, double *f_ext /* [B][nJ_max][3] */,
                    double *N /* [B][nM_max] */, int32_t *info /* [B] */,
                    int32_t *free_index /* [B][nJ_max*3] or NULL */, int32_t *n_free /* [B] or NULL */,
                    const double *rho /* [B][nM_max] or NULL */, double allow_stress, double allow_displace,
                    double *weight /* [B] or NULL */, double *stress_vio /* [B] */, double *disp_vio /* [B] */,
                    void *stream);

/* Graph features of a solved batch of 3D trusses, on the device (the consumer of the path in the
 * dataset workload: TrussHeteroDataCreator, data.py:116-282; GetAngles, utils.py:105-113): float32
 * joint_x [B][nJ_max][7 (+3 with a prior)], member_x [B][nM_max][8 (+1 prior) (+1 regression)],
 * joint_y [B][nJ_max][3] and member_y [B][nM_max] (regression only), weight [B] (double).  u_* are the
 * dense displacements [B][nJ_max][3], N_* the member forces [B][nM_max] of the solve with the real
 * sections (act) and with every member set to one fixed section of area fixedArea (pri); NULL = absent.
 * Bit-identical to the host version trs_graph_features of include/trs_host.h. */
int trs_graph_features_dev(int B, int nJ_max, int nM_max, const double *xyz, const int32_t *conn,
                           const double *A, const double *rho, const uint8_t *cbits, const double *loads,
                           const int32_t *nJ, const int32_t *nM, const double *u_act, const double *N_act,
                           const double *u_pri, const double *N_pri, double fixedArea, double forceScale,
                           double displaceScale, double positionScale, int regression, float *joint_x,
                           float *member_x, float *joint_y, float *member_y, double *weight, void *stream);

/* The same features in PACKED form - the sink of the dataset path (BASELINE config 5: "results streamed to PyG
 * HeteroData"; reference TrussHeteroDataCreator.__CreateGraphData / __CreateEdges, data.py:238-282): the joint rows of
 * all trusses back to back (truss b: rows joint_off[b] .. joint_off[b] + nJ[b] - 1 of joint_x [sum nJ][FJ] and
 * joint_y [sum nJ][3]), the member rows likewise (member_off, member_x [sum nM][FM], member_y [sum nM][1]) - exactly
 * the tensors a HeteroData of every sample slices, with no padding to carry across PCIe - and j2m_joint
 * [sum nM][2] (int32), the members' end joints = row 0 of a sample's `j2m` edge index (row 1 is 0,0,1,1,2,2,...;
 * `m2j` is the two rows swapped; data.py:238-247), or NULL.  joint_off / member_off: exclusive prefix sums of nJ / nM
 * (int64, device).  Values bit-identical to trs_graph_features_dev. */
int trs_graph_features_packed(int B, int nJ_max, int nM_max, const double *xyz, const int32_t *conn,
                              const double *A, const double *rho, const uint8_t *cbits, const double *loads,
                              const int32_t *nJ, const int32_t *nM, const double *u_act, const double *N_act,
                              const double *u_pri, const double *N_pri, double fixedArea, double forceScale,
                              double displaceScale, double positionScale, int regression,
                              const int64_t *joint_off, const int64_t *member_off, float *joint_x, float *member_x,
                              float *joint_y, float *member_y, int32_t *j2m_joint, double *weight, void *stream);

/* Joint order on the device (no reference counterpart: slientruss3d numbers joints in insertion order,
 * truss.py:175; the workload is its GenerateRandomCubeTrusses loop, generate.py:342-374, whose trusses are far
 * from banded in generator order).  The cost of a solve is set by the row envelope of K_ff, i.e. by the joint
 * numbering, so the numbering is the solver's to choose: per truss the cheapest of reverse Cuthill-McKee, its
 * reverse and twelve binned coordinate sweeps, priced by the 16 x 16-tile envelope the factorisation works in
 * (cost = sum over row chunks of w (w + 12), w = tiles from the first coupled tile to the diagonal) - the same
 * candidates, cost and tie-breaks as trs_profile_order of trs_host.h, whose permutation it reproduces, with
 * no host pass.  One work-group per truss, everything in LDS (csrc/order.hip).
 *   perm  [B][nJ_max]   out: old id of the joint that becomes joint k (identity on the padding); this is also the
 *                       joint_out of trs_recover / trs_solve that delivers results in the ORIGINAL numbering
 *   choice [B] or NULL  out: winning candidate (0 RCM, 1 its reverse, 2 + 2 p + r: sweep with axis order p,
 *                       r = 1 backwards)
 *   reach  [B] or NULL  out: how many 16-row chunks the envelope of the CHOSEN order reaches below its 64 x 64
 *                       diagonal blocks - a batch that stays <= 24 everywhere may be solved with
 *                       TRS_ASM_ALL_NARROW / TRS_HINT_NO_WIDE
 *   xyz_out, conn_out, cbits_out, loads_out (all four or all NULL; not the input arrays): the renumbered trusses,
 *                       joint k := old joint perm[k]; members keep their order (N needs no mapping), their end
 *                       joints are renumbered, padding members stay (0, 0)
 *   effort              0: RCM and its reverse, 1: + the sweep along the longest extent, 2: all sweeps, 3: all sweeps,
 *                       RCM and its reverse only for trusses with fewer than 128 free joints or without a usable
 *                       sweep (on larger lattice-like trusses a sweep wins and Cuthill-McKee is 40 % of the kernel's
 *                       time; +0.2 % stored tiles over 2048 mixed cube trusses)
 *                       At effort 3 a truss of 128 or more free joints WITHOUT a usable sweep (all members of length
 *                       zero, coordinates that are not numbers) keeps its free joints in the given order (choice 14).
 * trs_joint_order_fits says whether a batch shape can be ordered on the device (tables within a CU's LDS,
 * nJ_max < 8192); trs_joint_order returns hipErrorInvalidValue for a shape that cannot (callers then use the host
 * version). */
int trs_joint_order_fits(int nJ_max, int nM_max);
/* trs_joint_order with the bucket GATHER of a ragged batch folded in (ABI 9; `batch.RaggedSolver`): truss b of the
 * launch is row rows[b] (int64, device) of the input arrays, whose rows are nJ_in_max joints / nM_in_max members wide;
 * the renumbered truss, its member sections E / A (members keep their order) and its counts nJ / nM are written as
 * row b of the bucket's arrays (nJ_max / nM_max wide, padding zeroed) - no separate copy of the bucket's rows, no
 * intermediate in the caller's numbering.  The truss's own sizes must fit nJ_max / nM_max (the bucket's maxima), for
 * which trs_joint_order_fits must hold.  perm, reach as trs_joint_order. */
int trs_joint_order_rows(int B, int nJ_max, int nM_max, const int64_t *rows, int nJ_in_max, int nM_in_max,
                         const double *xyz, const int32_t *conn, const uint8_t *cbits, const double *loads,
                         const double *E, const double *A, const int32_t *nJ, const int32_t *nM, int32_t *perm,
                         int32_t *reach /* or NULL */, double *xyz_out, int32_t *conn_out, uint8_t *cbits_out,
                         double *loads_out, double *E_out, double *A_out, int32_t *nJ_out, int32_t *nM_out, int effort,
                         void *stream);
int trs_joint_order(int B, int nJ_max, int nM_max, const double *xyz, const int32_t *conn, const uint8_t *cbits,
                    const double *loads /* may be NULL when loads_out is */, const int32_t *nJ, const int32_t *nM,
                    int32_t *perm, int32_t *choice, int32_t *reach, double *xyz_out, int32_t *conn_out,
                    uint8_t *cbits_out, double *loads_out, int effort, void *stream);

/* Random cube trusses generated on the device, straight into the padded batch arrays (the reference's
 * GenerateRandomCubeTrusses, generate.py:152-376: polycube growth DFS / BFS / Random, cube vertices as joints in
 * first-seen order, 6 face diagonals by LinkType + 12 edges per cube, pins on the lowest occupied layer, random
 * loads on unsupported joints, a member type per member, count-unstable draws regenerated).  Bit for bit the output
 * of the host generator trs_cubegen of trs_host.h for the same (seed, first_index + b): same per-truss splitmix64
 * stream, same order of draws, same arithmetic.  One wave per truss, grid state in LDS (csrc/cubegen.hip).
 *   num_cubes [B] (device)  polycube size per truss;  method 0 DFS / 1 BFS / 2 Random;  link_type 0 / 1 / 2 (both
 *   diagonals) / 3 random;  flags bit 0 = keep parallel members, bit 1 = no pin supports;
 *   force_range [3][2] (HOST array, read at call time);  mtypes [n_types][3] = (a, e, density) (device)
 *   xyz == NULL: sizes-only pass - nJ[b], nM[b], n_free[b] (= 3 x unsupported joints; may be NULL) are produced
 *   with nJ_max / nM_max as bounds only; callers size the batch arrays by its maxima and call again to fill.
 *   status [2] (device, zeroed by the caller): [0] += regenerated attempts, [1] = 1 if a truss did not fit.
 * Grids up to 254 cells per axis and 32767 vertices whose tables fit a CU's LDS; hipErrorInvalidValue otherwise. */
int trs_cubegen_dev(int B, uint64_t seed, int gx, int gy, int gz, const int32_t *num_cubes, int method, int link_type,
                    int flags, double len_lo, double len_hi, const double *force_range, int nforce_lo, int nforce_hi,
                    const double *mtypes, int n_types, int nJ_max, int nM_max, double *xyz, int32_t *conn, double *E,
                    double *A, double *rho, uint8_t *cbits, double *loads, int32_t *nJ, int32_t *nM, int32_t *n_free,
                    int32_t *status, int64_t first_index, void *stream);

/* Row gather / scatter between the padded arrays of a ragged batch and those of one of its size buckets
 * (no reference counterpart: the reference solves one truss per call; its ragged workload is the
 * GenerateRandomCubeTrusses loop, generate.py:342-374).  A bucket's arrays are the rows of its trusses trimmed
 * to the bucket's own maxima; every trimmed field is a PREFIX of the full row, so both directions copy, for every
 * field k < nfields (at most 12, one launch) and every i < count, `width[k]` bytes:
 *   scatter == 0:  dst[k] + i * dst_pitch[k]        <-  src[k] + rows[i] * src_pitch[k]     (gather)
 *   scatter != 0:  dst[k] + rows[i] * dst_pitch[k]  <-  src[k] + i * src_pitch[k]           (scatter)
 * fill_to (or NULL): behind the copied prefix the destination row is ZEROED up to fill_to[k] bytes (a scatter into
 * rows wider than the bucket's).
 * counts / elem (or NULL, and counts[k] may be NULL): counts[k] is a DEVICE array of int32, one per bucket row i - only
 * min(width[k], counts[k][i] * elem[k]) bytes of that row are live and copied (the nJ or nM of the truss times the
 * bytes per joint / member); the rest of the row up to fill_to[k] is zeroed, not copied - over PCIe only live
 * bytes cross.
 * live (or NULL; scatter only; live[k] may be NULL): live[k] is a DEVICE array of int32, one per row of the FAR
 * array dst[k], that states an invariant of that array: "row r is zero behind byte live[k][r]".  The scatter
 * then zeroes only [copied bytes, live[k][r]) - what the previous writer of that row left - and records the new
 * extent (fill_to[k] is ignored for such a field): results pushed again and again into the same page-locked
 * arrays pay for their zero padding once, when the array is created (batch.ResultPool).
 * src / dst / pitches / widths / fill_to / counts / elem / live are HOST arrays of device pointers and byte
 * counts (read at call time); rows is a device array of int64 row indices (distinct for a scatter).
 * Either side may be page-locked HOST memory (mapped into the device's address space): the gather then pulls a
 * bucket's rows over PCIe straight out of the caller's host batch, the scatter pushes results into the caller's
 * host arrays; max_blocks > 0 keeps such a launch to that many work-groups (it needs bytes in flight, not CUs);
 * 0 = default. */
int trs_copy_rows(int nfields, const void *const *src, const size_t *src_pitch, void *const *dst,
                  const size_t *dst_pitch, const size_t *width, const size_t *fill_to,
                  const int32_t *const *counts, const size_t *elem, int32_t *const *live, int count,
                  const int64_t *rows, int scatter, int max_blocks, void *stream);

/* A stream whose kernels run only on the compute units set in `cu_mask` (bit i % 32 of word i / 32 = logical CU i
 * of the current device; on MI355X consecutive bits go round the eight XCDs, so bits 0-7 are ONE CU of each XCD).
 * The host-fed pipeline gives its pull and push kernels (trs_copy_rows on page-locked host memory) a few CUs of
 * their own and the solver kernels the rest: a copy wave then never waits for registers the factorisation holds,
 * nor takes them from it.  The stream is created non-blocking; destroy it with trs_stream_destroy after the work
 * queued on it has finished.  No counterpart in the reference (plumbing of ABI 8). */
int trs_stream_create_masked(const uint32_t *cu_mask, int n_words, void **stream);
int trs_stream_destroy(void *stream);

/* The whole Truss.Solve() pipeline (truss.py:329-364) on one stream: trs_solve_small when the batch
 * shape qualifies (the slab, uf, work and env arguments are then not touched), otherwise
 * dofmap -> assemble -> potrf -> potrs -> recover.  Workspace pointers as above; joint_out as in
 * trs_recover (a batch with a joint_out never takes the fused small-system kernel). */
int trs_solve(int B, int nJ_max, int nM_max, int n_max_bound,
              const double *xyz, const int32_t *conn, const double *E, const double *A,
              const uint8_t *cbits, const double *loads, const int32_t *nJ, const int32_t *nM,
              int32_t *free_index, int32_t *n_free, int ld, int slab_rows, double *S,
              double *uf, int ld_uf, double *u, double *f_ext, double *N, int32_t *info,
              void *work, int32_t *env /* workspace for the envelope metadata, or NULL = dense */,
              const int32_t *joint_out /* [B][nJ_max] or NULL */,
              int hints /* TRS_HINT_NO_WIDE: route and treat every matrix as narrow, TRS_HINT_COMPACT,
                           TRS_HINT_SEPARATE_STAGES, TRS_HINT_NO_SMALL, TRS_HINT_RECOVER_UNSTAGED, TRS_HINT_ALL_TILES
                           as above */,
              void *stream);

/* trs_solve of ONE size bucket of a ragged batch with the result scatter folded in (trs_recover_rows as the last
 * stage; never the fused small-system kernel): u, f_ext, N, info_out are the FULL batch's result arrays (rows
 * nJ_out_max / nM_out_max wide), out_rows[b] the row of truss b in them; `info` is the bucket's own status array. */
int trs_solve_rows(int B, int nJ_max, int nM_max, int n_max_bound, const double *xyz, const int32_t *conn,
                   const double *E, const double *A, const uint8_t *cbits, const double *loads, const int32_t *nJ,
                   const int32_t *nM, int32_t *free_index, int32_t *n_free, int ld, int slab_rows, double *S,
                   double *uf, int ld_uf, double *u, double *f_ext, double *N, int32_t *info, void *work, int32_t *env,
                   const int32_t *joint_out, const int64_t *out_rows, int nJ_out_max, int nM_out_max,
                   int32_t *info_out, int hints, void *stream);

/* ---- table member form (ABI 10, "Member forms" above): the twins of the entry points that read members.  Arguments
 * as their twins', with (conn16, type_idx, types) in place of (conn, E, A) [and rho: the densities of the fitness
 * reductions of trs_solve_small_tab come from the table]. ---- */
int trs_assemble_tab(int B, int nJ_max, int nM_max, const double *xyz, const uint16_t *conn16, const uint8_t *type_idx,
                     const double *types, const double *loads, const int32_t *free_index, const int32_t *n_free,
                     const int32_t *nJ, const int32_t *nM, int ld, int slab_rows, double *S, int flags, void *work,
                     int32_t *env, double *uf, int ld_uf, void *stream);
int trs_recover_tab(int B, int nJ_max, int nM_max, const double *xyz, const uint16_t *conn16, const uint8_t *type_idx,
                    const double *types, const double *loads, const int32_t *free_index, const int32_t *nJ,
                    const int32_t *nM, const double *uf, int ld_uf, double *u, double *f_ext, double *N,
                    const int32_t *joint_out, int hints, void *stream);
int trs_recover_rows_tab(int B, int nJ_max, int nM_max, const double *xyz, const uint16_t *conn16,
                         const uint8_t *type_idx, const double *types, const double *loads, const int32_t *free_index,
                         const int32_t *nJ, const int32_t *nM, const double *uf, int ld_uf, const int32_t *joint_out,
                         const int32_t *info, const int64_t *out_rows, int nJ_out_max, int nM_out_max, double *u,
                         double *f_ext, double *N, int32_t *info_out, int hints, void *stream);
int trs_solve_small_tab(int B, int nJ_max, int nM_max, int n_max_bound, const double *xyz, const uint16_t *conn16,
                        const uint8_t *type_idx, const double *types, const uint8_t *cbits, const double *loads,
                        const int32_t *nJ, const int32_t *nM, double *u, double *f_ext, double *N, int32_t *info,
                        int32_t *free_index, int32_t *n_free, double allow_stress, double allow_displace,
                        double *weight /* or NULL */, double *stress_vio, double *disp_vio, void *stream);
int trs_joint_order_tab(int B, int nJ_max, int nM_max, const double *xyz, const uint16_t *conn16, const uint8_t *cbits,
                        const double *loads, const int32_t *nJ, const int32_t *nM, int32_t *perm, int32_t *choice,
                        int32_t *reach, double *xyz_out, uint16_t *conn16_out, uint8_t *cbits_out, double *loads_out,
                        int effort, void *stream);
int trs_joint_order_rows_tab(int B, int nJ_max, int nM_max, const int64_t *rows, int nJ_in_max, int nM_in_max,
                             const double *xyz, const uint16_t *conn16, const uint8_t *cbits, const double *loads,
                             const uint8_t *type_idx, const int32_t *nJ, const int32_t *nM, int32_t *perm,
                             int32_t *reach, double *xyz_out, uint16_t *conn16_out, uint8_t *cbits_out,
                             double *loads_out, uint8_t *type_idx_out, int32_t *nJ_out, int32_t *nM_out, int effort,
                             void *stream);
int trs_solve_tab(int B, int nJ_max, int nM_max, int n_max_bound, const double *xyz, const uint16_t *conn16,
                  const uint8_t *type_idx, const double *types, const uint8_t *cbits, const double *loads,
                  const int32_t *nJ, const int32_t *nM, int32_t *free_index, int32_t *n_free, int ld, int slab_rows,
                  double *S, double *uf, int ld_uf, double *u, double *f_ext, double *N, int32_t *info, void *work,
                  int32_t *env, const int32_t *joint_out, int hints, void *stream);
int trs_solve_rows_tab(int B, int nJ_max, int nM_max, int n_max_bound, const double *xyz, const uint16_t *conn16,
                       const uint8_t *type_idx, const double *types, const uint8_t *cbits, const double *loads,
                       const int32_t *nJ, const int32_t *nM, int32_t *free_index, int32_t *n_free, int ld,
                       int slab_rows, double *S, double *uf, int ld_uf, double *u, double *f_ext, double *N,
                       int32_t *info, void *work, int32_t *env, const int32_t *joint_out, const int64_t *out_rows,
                       int nJ_out_max, int nM_out_max, int32_t *info_out, int hints, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TRS_SOLVER_H */
