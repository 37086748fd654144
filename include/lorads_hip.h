/* lorads_hip.h -- C ABI of the MI355X (gfx950) backend for the LoRADS per-iteration path.
 *
 * Drop-in boundary: the reference reaches every per-iteration step of both phases
 * through its operator table `lorads_func` (src_semi/data/def_lorads_solver.h:109-127, filled by
 * LORADSInitFuncSet, src_semi/data/lorads_solver.c:717-756) plus three non-table calls.  Each entry
 * point below replaces one of those slots; the citation on each names the reference function whose
 * result it reproduces.  The state the reference keeps in `lorads_solver`/`lorads_variable`
 * (R, U, V, Grad, dualVar, constrVal[], constrValSum, ARDSum, ADDSum, L-BFGS ring, CG workspaces,
 * src_semi/data/def_lorads_solver.h:12-106) lives on the device inside the opaque context; the
 * movers at the end upload/download it in the reference's own layout (column-major n x r, ld = n).
 *
 * Plain C: opaque pointer, int32/double pointers and sizes only; every function returns 0 on
 * success and a non-zero code on failure (lorads_hip_last_error gives the text).  One context per
 * process/GPU; calls on one context must be serialised by the caller (as the reference's single
 * thread does).  INTEGRATION.md shows the shim a reference maintainer would add.
 */
#ifndef LORADS_HIP_H
#define LORADS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lorads_hip_ctx lorads_hip_ctx;

/* One SDP cone after the reference's pre-solve, flat (what AConeProcData leaves in
 * lorads_cone_sdp_dense/sparse, src_semi/data/def_lorads_sdp_conic.h:101-127): lower-triangular
 * triplets, CSR by constraint.  Pointers are HOST pointers and are copied. */
typedef struct {
    int32_t n;              /* cone dimension */
    int32_t rank;           /* current factor rank r (LORADSDetermineRank) */
    int32_t nrow;           /* constraints with a non-zero A_i on this cone */
    const int32_t *row_idx; /* [nrow] global constraint index (sparse cone: rowIdx) */
    const int32_t *a_ptr;   /* [nrow+1] */
    const int32_t *a_row;   /* [a_ptr[nrow]] row >= col */
    const int32_t *a_col;
    const double *a_val;
    int32_t c_nnz; /* objective matrix C (sign as stored by the reference: C = -F0) */
    const int32_t *c_row;
    const int32_t *c_col;
    const double *c_val;
    /* 1: this is the LP block of the file (lorads_lp_cone, data/def_lorads_lp_conic.h; at most one, the last block):
     * n = number of LP columns, rank 1, every entry diagonal ((i,i,a) = coefficient a of column i).  Phase 1 and the
     * evaluations treat it as a diagonal cone; admm_update_var updates it column by column in closed form
     * (LORADSUpdateSDPLPVar, lorads_alg_common.c:225-248). */
    int32_t is_lp;
} lorads_hip_block;

typedef struct {
    int32_t m;          /* number of constraints (ASolver->nRows) */
    const double *b;    /* [m] rowRHS */
    double b_nrm1;      /* ||b||_1 (bRHSNrm1) */
    int32_t nblocks;    /* cones held by this context (multi-GPU: a subset) */
    const lorads_hip_block *blocks;
    int32_t lbfgs_len;  /* params->lbfgsListLength */
    int32_t device;     /* HIP device ordinal, -1 = current */
} lorads_hip_problem;

/* sum `count` doubles in place over all ranks; buf is a DEVICE pointer when on_device != 0.  The
 * library synchronises its stream before the call. */
typedef int (*lorads_hip_allreduce_fn)(void *user, double *buf, int32_t count, int32_t on_device);

enum { LORADS_HIP_PAIR_RR = 0, LORADS_HIP_PAIR_UV = 1 };
enum { LORADS_HIP_MAT_R = 0, LORADS_HIP_MAT_U = 1, LORADS_HIP_MAT_V = 2, LORADS_HIP_MAT_GRAD = 3 };
enum { LORADS_HIP_VEC_LAMBDA = 0, LORADS_HIP_VEC_CONSTR_SUM = 1, LORADS_HIP_VEC_Q1 = 2, LORADS_HIP_VEC_Q2 = 3 };

int lorads_hip_create(const lorads_hip_problem *prob, lorads_hip_ctx **out);
void lorads_hip_destroy(lorads_hip_ctx *ctx);
const char *lorads_hip_last_error(void);

/* lorads_func.InitConstrValAll + InitConstrValSum (lorads_alg/lorads_alg_common.c:78-84,134-142):
 * constrVal[k] = A_k(sym(X Y^T)), constrValSum = sum_k; pair RR uses (R,R), pair UV uses (U,V) */
int lorads_hip_init_constr(lorads_hip_ctx *ctx, int32_t pair);
/* lorads_func.ALMCalGrad (lorads_alg/lorads_alm.c:9-54): Grad_k = 2 (C + sum_i M1_i A_i) R_k with
 * M1 = -lambda - rho b + rho constrValSum; *lag_norm_sq = sum_k ||Grad_k||_F^2 */
int lorads_hip_alm_cal_grad(lorads_hip_ctx *ctx, double rho, double *lag_norm_sq);
/* lorads_func.LBFGSDirection + LBFGSDirUseGrad (lorads_alm.c:230-391,469-489); direction D in U */
int lorads_hip_lbfgs_direction(lorads_hip_ctx *ctx, int32_t inner_iter);
/* lorads_func.ALMCalq12p12 (lorads_alm.c:540-560): q1 = 2A(sym(R D^T)), q2 = A(D D^T) stay on the
 * device; p12 = { 2<C,sym(R D^T)>, <C, D D^T> } */
int lorads_hip_alm_q12p12(lorads_hip_ctx *ctx, double p12[2]);
/* vector half of ALMLineSearch (lorads_alm.c:161-172): coefficients a,b,c,d of the quartic in tau;
 * the host solves the cubic (lorads_alm.c:114-154,173-227) */
int lorads_hip_alm_linesearch_coeffs(lorads_hip_ctx *ctx, double rho, double p1, double p2, double coef[4]);
/* lorads_func.setAsNegGrad (lorads_alm.c:583-598) */
int lorads_hip_set_y_as_neg_grad(lorads_hip_ctx *ctx);
/* lorads_func.ALMupdateVar + constrValSum += tau q1 + tau^2 q2 (lorads_alm.c:619-648,1122-1124) */
int lorads_hip_alm_update_var(lorads_hip_ctx *ctx, double tau);
/* lorads_func.setlbfgsHisTwo (lorads_alm.c:657-678) */
int lorads_hip_set_lbfgs_his_two(lorads_hip_ctx *ctx, double tau);
/* lorads_func.updateDimacsALM / updateDimacsADMM (lorads_alg_common.c:250-290): refreshes
 * constrVal/constrValSum from R R^T (pair UV: after R = (U+V)/2) and returns
 * ||b - constrValSum||_2 / (1 + ||b||_1) */
int lorads_hip_update_dimacs(lorads_hip_ctx *ctx, int32_t pair, double *err1);
/* lorads_func.calObj_alm / calObj_admm (lorads_alm.c:1259-1268, lorads_admm.c:325-337): <C, R R^T>
 * summed over this context's cones (pair UV: after R = (U+V)/2); not divided by scaleObjHis */
int lorads_hip_cal_obj(lorads_hip_ctx *ctx, int32_t pair, double *pobj);
/* Fused phase-1 inner iteration (optional; the slot-by-slot calls above give identical results).  The body of
 * the reference's inner loop (lorads_alm.c:1066-1131) is  [direction, q12p12, line-search sums] -> scalar cubic
 * on the host -> [setAsNegGrad, ALMupdateVar(tau), ALMCalGrad, setlbfgsHisTwo, updateDimacsALM].
 *   alm_front: first bracket for inner-iteration counter `inner`; out = {p1, p2, a, b, c, d}
 *   alm_step : second bracket with the host's tau, then (next_inner >= 0) the first bracket of the NEXT iteration,
 *              enqueued back to back with ONE host synchronisation;
 *              out = {lagNormSq, err1, p1, p2, a, b, c, d} (p, a..d belong to the next iteration)
 * If the host leaves the loop instead, the pre-computed direction is discarded (it lives in D = U, q1, q2). */
int lorads_hip_alm_front(lorads_hip_ctx *ctx, double rho, int32_t inner, double out[6]);
int lorads_hip_alm_step(lorads_hip_ctx *ctx, double rho, double tau, int32_t next_inner, double out[8]);
/* lorads_func.admmUpdateVar = LORADSUpdateSDPVar (lorads_alg_common.c:187-215) with
 * LORADSUpdateSDPVarOne (lorads_admm.c:428-480) and CGSolve (linalg/lorads_cgs.c:81-240);
 * *cg_iters = sum of the CG iteration counts the reference would add to ASolver->cgIter */
int lorads_hip_admm_update_var(lorads_hip_ctx *ctx, double rho, double cg_tol, int32_t cg_max_iter,
                               int32_t *cg_iters);
/* One ADMM iteration up to (not including) the dual update, i.e. admmUpdateVar + calObj_admm +
 * LORADSCalDualObj + updateDimacsADMM in the order of lorads_admm.c:76-81, with ONE host
 * synchronisation: out = { cg iterations, <C,RR^T> (unscaled), b.lambda (unscaled), err1 } */
int lorads_hip_admm_step(lorads_hip_ctx *ctx, double rho, double cg_tol, int32_t cg_max_iter, double out[4]);
/* LORADSUpdateDualVar / LORADSCalDualObj (lorads_alg_common.c:319-340) */
int lorads_hip_update_dual_var(lorads_hip_ctx *ctx, double rho);
int lorads_hip_cal_dual_obj(lorads_hip_ctx *ctx, double *dobj);

/* calculate_dual_infeasibility_solver + dual_infeasible (data/lorads_solver.c:1007-1037,
 * data/lorads_sdp_conic.c:1286-1349): *sum_neg = sum over this context's cones of
 * |min(lambda_min(C_k - sum_i lambda_i A_ik), 0)| -- the caller divides by scaleObjHis (1 + ||C||_1) as
 * :1034-1035 do.  tol, ncv, max_restarts are the ARPACK parameters of the reference (1e-2, 40, 600); the
 * eigenvalue comes from an on-device thick-restart Lanczos with the same subspace size and stopping rule.
 * lam_min ([nblocks], may be NULL) receives the per-cone eigenvalue, *matvecs (may be NULL) the S x count. */
int lorads_hip_dual_infeasibility(lorads_hip_ctx *ctx, double tol, int32_t ncv, int32_t max_restarts, double *sum_neg,
                                  double *lam_min, int32_t *matvecs);

/* state movers ("mutators outside the table") */
int lorads_hip_alm_to_admm(lorads_hip_ctx *ctx);        /* LORADS_ALMtoADMM copies, data/lorads_solver.c:968-983 */
int lorads_hip_average_uv_to_v(lorads_hip_ctx *ctx);    /* averageUV + copyRtoV, main.c:441-448 */
int lorads_hip_scale_obj(lorads_hip_ctx *ctx, double s); /* objScale_dualvar, data/lorads_solver.c:1040-1052 */
int lorads_hip_resize_rank(lorads_hip_ctx *ctx, const int32_t *new_rank); /* AUG_RANK, data/lorads_solver.c:806-906 */
/* column-major n x r host arrays, as lorads_sdp_dense.matElem (data/def_lorads_elements.h:29-33) */
int lorads_hip_set_mat(lorads_hip_ctx *ctx, int32_t which, int32_t blk, const double *colmajor);
int lorads_hip_get_mat(lorads_hip_ctx *ctx, int32_t which, int32_t blk, double *colmajor);
int lorads_hip_set_vec(lorads_hip_ctx *ctx, int32_t which, const double *v);
int lorads_hip_get_vec(lorads_hip_ctx *ctx, int32_t which, double *v);
int lorads_hip_set_allreduce(lorads_hip_ctx *ctx, lorads_hip_allreduce_fn fn, void *user);
/* The library's HIP stream (hipStream_t).  A hook that ENQUEUES its collective on this stream (RCCL
 * ncclAllReduce(..., stream)) can declare itself stream-ordered: the library then does not synchronise
 * the host before calling it, so a multi-GPU ADMM iteration still has a single host sync. */
void *lorads_hip_stream(lorads_hip_ctx *ctx);
int lorads_hip_set_allreduce_stream_ordered(lorads_hip_ctx *ctx, int32_t on);
/* Sharded cones whose constraints are BLOCK-SEPARABLE over the ranks (no constraint touches cones of two ranks): the caller
 * creates each rank's context on the sub-problem over the rank's own constraints (m, b, row indices local; b_nrm1 of the
 * whole problem) and declares it here.  Every m-vector of the method then lives on one rank only, and the library sums
 * SCALARS over the ranks instead of constrValSum, q1, q2: per ADMM iteration one all-reduce of four doubles
 * {||b - A(RR^T)||^2 part, b.lambda part, <C,RR^T> part, "my sweep is unfinished"}, and the rank's iteration is the
 * single-GPU iteration (all its fused paths) around it.  lorads_hip_get_vec / set_vec move the rank's own pieces. */
int lorads_hip_set_separable(lorads_hip_ctx *ctx, int32_t on);
/* Separable shards on the GPUs of ONE node: the four scalars of an ADMM iteration's evaluation are read by nobody but the ranks'
 * hosts, which wait for their GPU's result hand-over at that very point.  With a scalar exchange installed the library takes the
 * collective off the stream: the evaluation leaves the rank's LOCAL sums in the control block, the hand-over kernel delivers them
 * with everything else, and each rank's host calls fn(user, vals, 4) -- which must leave in vals the sums over all ranks, the same
 * bits on every rank -- before it looks at them (lorads_amd/csrc/host/shmx.c: a page of POSIX shared memory, sums in rank order).
 * The sharded iteration's launch chain is then the single-GPU chain + one one-workgroup kernel.  Phase 1's collectives (which feed
 * device-side consumers) and the m-vector form keep the all-reduce hook.  fn = NULL removes it.  (The reference has no counterpart:
 * it sweeps all cones in one process, lorads_alg/lorads_alg_common.c:190-214.) */
typedef int (*lorads_hip_scalar_exchange_fn)(void *user, double *vals, int32_t n);
int lorads_hip_set_scalar_exchange(lorads_hip_ctx *ctx, lorads_hip_scalar_exchange_fn fn, void *user);
/* all-reduces constrValSum through the hook once (lets the caller validate its hook) */
int lorads_hip_selfcheck_allreduce(lorads_hip_ctx *ctx);

int lorads_hip_sync(lorads_hip_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
