/* lorads_hip_dev.h -- measurement and diagnostic entries of liblorads_hip.so (bench.py, profiles/tools/).
 *
 * NOT part of the drop-in surface: nothing here replaces a reference function, and a reference-side shim
 * (integration/lorads_func_hip.c) never includes this header.  The product ABI is lorads_hip.h.
 */
#ifndef LORADS_HIP_DEV_H
#define LORADS_HIP_DEV_H

#include "lorads_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* HIP-event timing of the CG operator application on the library's stream.
 * stats[0..7] = {cg_matvec launches, speculation misses (resumed solves), cg iterations, cg solves,
 *                sampled matvec launches, sampled matvec ms, spmm sampled launches, spmm sampled ms} */
int lorads_hip_profile(lorads_hip_ctx *ctx, int32_t enable, int32_t sample_every);
int lorads_hip_profile_read(lorads_hip_ctx *ctx, double stats[8]);
/* what the next profiling windows time, one launch group in `sample_every`: 0 = CG operator applications (default), 1 = the front of
 * a CG solve (right-hand side + initial residual: k_front_cw / k_spmm2<FRONT>).  The events bracket the launches the iteration makes
 * anyway; nothing is launched differently because it is timed. */
int lorads_hip_profile_target(lorads_hip_ctx *ctx, int32_t target);
/* the individual samples (milliseconds per timed operator application) behind stats[4..5], oldest first: copies
 * min(cap, count) of them to out and returns the count in *n (for a median / spread next to the mean) */
int lorads_hip_profile_samples(lorads_hip_ctx *ctx, double *out, int32_t cap, int32_t *n);
/* `reps` applications of the live CG operator of cone 0 (whatever kernels apply it) to the CG direction buffer, back to back
 * between one event pair, no scalar step riding along -- "the operator alone", boundaries between consecutive launches included;
 * elapsed milliseconds of all of them.  Overwrites the CG scratch vectors; the factors are left alone. */
int lorads_hip_time_operator(lorads_hip_ctx *ctx, int32_t reps, double *ms);
/* DEVELOPMENT build only (liblorads_hip_dev.so; profiles/tools/ubench.py): `reps` back-to-back launches of kernel variant `which`
 * on cone 0, elapsed milliseconds of all of them.  The product library does not export it. */
int lorads_hip_ubench(lorads_hip_ctx *ctx, int32_t which, int32_t reps, double *ms);
/* algorithmic bytes of one CG operator application / one CG iteration of block blk */
int lorads_hip_algorithmic_bytes(lorads_hip_ctx *ctx, int32_t blk, double *bytes_matvec, double *bytes_cg_iter);
/* which kernels apply the CG operator of block blk: 0 = k_pairdots + k_sgram + k_spmm (Gram of the A_i),
 * 1 = k_pairdots + k_cv + k_sval + k_spmm, 2 = k_op_diag (every A_i one diagonal entry), 3 = k_op_entry (every A_i
 * one entry), 4 = k_cw + k_spmm_ell (constraint values straight from the factors, slot coefficients a w_i);
 * + 16: the cone also holds DENSE constraint matrices, whose part of A / A^* runs through the dense GEMM (k_dense_cx_b)
 * + 32: kind 3 in its two-colour form (bipartite entry graph: k_op_entry_bip, one launch per colour) */
int lorads_hip_operator_kind(lorads_hip_ctx *ctx, int32_t blk, int32_t *kind);
/* the device image lorads_hip_create built for block blk (the device-side half of the pre-solve: AConePresolveData,
 * data/lorads_sdp_conic.c:868-1076; sdpDataMatSetData, data/lorads_sdp_data.c:811-828): image[0..15] = {n, rank, constraints held,
 * nnz of the held A_i (sparse ones), nnz of C, unique positions of the A_i, unique positions of C u A_i (A_i alone when C is stored
 * dense), C stored dense (the reference's rule), dense constraint matrices kept full, Max-Cut-type cone, single-entry cone,
 * constraint-wise operator, Gram form available, one-kernel front available, fixed width of the slot list, rows of either colour
 * of a bipartite entry graph (0: not bipartite) packed as colour-0 count} */
int lorads_hip_block_image(lorads_hip_ctx *ctx, int32_t blk, int64_t image[16]);
/* replay of captured launch chains (hipGraph; LORADS_GRAPH=0 switches it off): stats = {chains captured, chains replayed,
 * chains held now, 1 if the replay is enabled for this context} */
int lorads_hip_graph_stats(lorads_hip_ctx *ctx, int64_t stats[4]);
/* pattern work of the pre-solve on the device (lorads_amd/csrc/hip/presolve.inc; what AConePresolveData does with qsort and a hash
 * table, data/lorads_sdp_conic.c:868-1076): stats = {sparsity patterns built by the device sorts, of these compared with the host
 * construction (LORADS_PRESOLVE_CHECK=1; a difference fails lorads_hip_create)} */
int lorads_hip_presolve_stats(lorads_hip_ctx *ctx, int64_t stats[2]);
/* evaluations of separable shards whose four scalars the ranks' hosts have summed (lorads_hip_set_scalar_exchange) instead of a
 * collective on the stream */
int lorads_hip_scalar_exchange_count(lorads_hip_ctx *ctx, int64_t *n);
/* the one-launch ADMM iteration of contexts whose cones are all of Max-Cut type (lorads_amd/csrc/hip/persist.inc; LORADS_PERSIST=0
 * switches it off): stats = {ADMM iterations run as one launch, 1 if the form is available for this context now, workgroups of
 * the launch, rows per 8-lane group, column steps, bytes of LDS per workgroup} */
int lorads_hip_persist_stats(lorads_hip_ctx *ctx, int64_t stats[6]);
/* enable != 0: the leader workgroup of cone 0's team leaves the 100 MHz clock at its phase boundaries in every such launch;
 * ticks[0..15] = those of the latest launch {start, U front done, U solve done, V front done, V solve done, evaluation done,
 * hand-over begins, 0 ...} (reads after synchronising the stream) */
int lorads_hip_persist_stamps(lorads_hip_ctx *ctx, int32_t enable, uint64_t ticks[16]);

/* phase 1, single rank, history length <= 2: setlbfgsHisTwo + LBFGSDirection + LBFGSDirectionUseGrad of an inner iteration
 * (src_semi/lorads_alg/lorads_alm.c:230-391,469-489,540-560) as ONE launch of resident workgroups inside lorads_hip_alm_step
 * (lorads_amd/csrc/hip/lbfgs_team.inc; LORADS_LBFGS_TEAM=0: launch by launch).
 * stats = {launches so far, plan built (0/1), workgroups, pairs of doubles per thread and vector} */
int lorads_hip_lbfgs_team_stats(lorads_hip_ctx *ctx, int64_t stats[4]);

/* kernels this context has enqueued so far (its own launches; the direct hipLaunchKernelGGL sites of the scalar steps -- a handful
 * per solve -- are not counted): bench.py divides the difference over a timed region by its steps */
int lorads_hip_launch_count(lorads_hip_ctx *ctx, int64_t *n);

#ifdef __cplusplus
}
#endif
#endif
