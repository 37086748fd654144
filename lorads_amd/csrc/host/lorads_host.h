/* lorads_host.h -- plain-C host side of the MI355X low-rank SDP solver.
 *
 * The host keeps what BASELINE.json's north_star says it keeps: the SDPA reader, the pre-solver,
 * the rank rule, the LoRADS-compatible parameter block and the scalar outer-loop control of both
 * phases.  Every per-iteration numerical step is reached through ONE table, `lrd_backend`, which
 * mirrors the reference's operator table `lorads_func`
 * (reference: src_semi/data/def_lorads_solver.h:109-127, filled by LORADSInitFuncSet,
 * src_semi/data/lorads_solver.c:717-756).  The product wires that table to the HIP C-ABI library
 * (include/lorads_hip.h) and to nothing else; tests wire it to the CPU oracle to check the HIP path.
 */
#ifndef LORADS_HOST_H
#define LORADS_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* return codes of the four *Optimize* loops (reference: src_semi/lorads.h:62-65) */
#define LRD_RET_OK 0
#define LRD_RET_TIME_OUT 1
#define LRD_RET_NUM_ERR 4
#define LRD_RET_BAD_ITER 8

/* solver status (reference: src_semi/lorads.h:45-51) */
enum { LRD_UNKNOWN = 0, LRD_PRIMAL_DUAL_OPTIMAL, LRD_PRIMAL_OPTIMAL, LRD_MAXITER, LRD_TIME_LIMIT };

/* parameter block, field-for-field the reference's lorads_params (src_semi/lorads.h:82-105);
 * defaults in lrd_params_default() are the reference CLI defaults (src_semi/main.c:19-43). */
typedef struct {
    const char *fname;
    double initRho, rhoMax, rhoCellingALM, rhoCellingADMM;
    int maxALMIter, maxADMMIter;
    double timesLogRank;
    int rhoFreq;
    double rhoFactor, ALMRhoFactor, phase1Tol, phase2Tol, timeSecLimit, heuristicFactor;
    int lbfgsListLength;
    double endTauTol, endALMSubTol;
    int l2Rescaling, reoptLevel, dyrankLevel, highAccMode;
    int verbose; /* ours: 0 silences the log lines */
} lrd_params;

/* One SDP cone ("block") as a flat image.  All symmetric matrices are lower-triangular triplets
 * (row >= col) ordered by packed column-major index, which is the order the reference's reader
 * produces (src_semi/io/lorads_file_io.c:273-283, linalg/lorads_sparse_opts.c:37-52). */
typedef struct {
    int n;        /* cone dimension */
    int nrow;     /* constraints whose A_i is non-zero on this cone (reference nRowElem / nnzStat) */
    int *row_idx; /* [nrow] global constraint index, ascending */
    int *a_ptr;   /* [nrow+1] */
    int *a_row, *a_col;
    double *a_val;
    int c_nnz;
    int *c_row, *c_col;
    double *c_val; /* C = -F0 already applied (src_semi/io/lorads_file_io.c:279-281) */
    /* decisions the reference takes in its pre-solver, restated */
    int cone_sparse; /* 1: LORADS_CONETYPE_SPARSE_SDP (io/lorads_user_data.c:58-71, threshold 0.3 m) */
    int dense_mode;  /* 1: union scratch is dense packed (data/lorads_sdp_conic.c:884,970,989,1072) */
    /* union pattern of C and all A_i, unique lower-tri positions sorted by (col,row)
     * (data/lorads_sdp_conic.c:965-1068); the full lower triangle when dense_mode */
    int np;
    int *p_row, *p_col;
    int *a_pidx; /* [a_ptr[nrow]] position of each A entry in the pattern (nnzIdx2ResIdx) */
    int *c_pidx; /* [c_nnz] */
    int rank, rank_max; /* data/lorads_solver.c:290-319 */
    /* 1: this "cone" is the LP block of the file (one diagonal block, last; io/lorads_file_io.c:120-124,260-269):
     * n = number of LP columns, every entry diagonal, rank fixed at 1 -- x_i = r_i^2 (u_i v_i).  In phase 1 and in
     * every evaluation it behaves as a diagonal cone (data/lorads_lp_conic.c:172-217); only the ADMM update differs:
     * column by column in closed form (lorads_alg/lorads_admm.c:595-629, lorads_alg_common.c:225-248). */
    int is_lp;
    int global_id;      /* index of this cone in the file (multi-GPU sharding keeps a subset) */
} lrd_block;

typedef struct {
    int m;     /* number of constraints (global) */
    double *b; /* [m] */
    int nblk;  /* blocks held by THIS process */
    lrd_block *blk;
    int nblk_global;
    int nsdp_global;     /* SDP cones of the file (ASolver->nCones): the LP block is not one of them */
    int sum_dims_global; /* sum of all SDP block dims, for rho0 = 1/sqrt(.) (data/lorads_solver.c:1155-1162) */
    /* norms (data/lorads_solver.c:1054-1073), over ALL blocks of the file */
    double cObjNrm1, cObjNrm2, cObjNrmInf, bNrm1, bNrm2, bNrmInf;
    /* set by lrd_problem_localize: this image is one rank's sub-problem of a block-separable deal -- m, b and the row indices are
     * the rank's own; con_global[i] = index of local constraint i in the file (m_global constraints) */
    int separable, m_global;
    int *con_global;
} lrd_problem;

/* which factor pair an evaluation uses */
enum { LRD_PAIR_RR = 0, LRD_PAIR_UV = 1 };
/* state arrays that can be moved across the boundary (column-major n x r, as the reference) */
enum { LRD_MAT_R = 0, LRD_MAT_U = 1, LRD_MAT_V = 2, LRD_MAT_GRAD = 3 };
enum { LRD_VEC_LAMBDA = 0, LRD_VEC_CONSTR_SUM = 1, LRD_VEC_Q1 = 2, LRD_VEC_Q2 = 3 };

/* Cross-process reduction hook (multi-GPU, one process per GPU): sums `count` doubles in place over
 * all ranks.  `on_device` tells whether buf is a device pointer.  NULL hook = single process. */
typedef int (*lrd_allreduce_fn)(void *user, double *buf, int count, int on_device);

/* The operator table.  One slot per lorads_func slot, plus the non-table calls on the path
 * (LORADSUpdateDualVar / LORADSCalDualObj, lorads_alg/lorads_alg_common.h:22-23; the m-vector part
 * of ALMLineSearch, lorads_alg/lorads_alm.c:161-172) and the state movers listed in SURVEY.md 8(b).
 * All return 0 on success.  Objective values are returned UNSCALED by scaleObjHis. */
typedef struct lrd_backend {
    void *ctx;
    const char *name;
    /* lorads_func.InitConstrValAll + InitConstrValSum  (lorads_alg_common.c:78-84,134-142) */
    int (*init_constr)(void *ctx, int pair);
    /* lorads_func.ALMCalGrad (lorads_alm.c:9-54): Grad_k = 2 (C + sum_i M1_i A_i) R_k, returns sum ||Grad_k||^2 */
    int (*alm_cal_grad)(void *ctx, double rho, double *lag_norm_sq);
    /* lorads_func.LBFGSDirection + LBFGSDirUseGrad (lorads_alm.c:230-391,469-489); D is stored in U */
    int (*lbfgs_direction)(void *ctx, int inner_iter);
    /* lorads_func.ALMCalq12p12 (lorads_alm.c:540-560): q1,q2 stay in the backend; p12 returned */
    int (*alm_q12p12)(void *ctx, double p12[2]);
    /* m-vector part of ALMLineSearch (lorads_alm.c:164-172): quartic coefficients a,b,c,d */
    int (*alm_linesearch_coeffs)(void *ctx, double rho, double p1, double p2, double coef[4]);
    /* lorads_func.setAsNegGrad (lorads_alm.c:583-598) */
    int (*set_y_as_neg_grad)(void *ctx);
    /* lorads_func.ALMupdateVar + the two axpys on constrValSum (lorads_alm.c:619-648,1122-1124) */
    int (*alm_update_var)(void *ctx, double tau);
    /* lorads_func.setlbfgsHisTwo (lorads_alm.c:657-678) */
    int (*set_lbfgs_his_two)(void *ctx, double tau);
    /* lorads_func.updateDimacsALM / updateDimacsADMM (lorads_alg_common.c:250-290): returns
     * ||b - sum_k A_k(.)||_2 / (1 + ||b||_1); ADMM variant first sets R = (U+V)/2 */
    int (*update_dimacs)(void *ctx, int pair, double *err1);
    /* lorads_func.calObj_alm / calObj_admm (lorads_alm.c:1259-1268, lorads_admm.c:325-337) */
    int (*cal_obj)(void *ctx, int pair, double *pobj);
    /* lorads_func.admmUpdateVar (lorads_alg_common.c:187-215): U- and V-solve per cone by CG */
    int (*admm_update_var)(void *ctx, double rho, double cg_tol, int cg_max_iter, int *cg_iters);
    /* LORADSUpdateDualVar / LORADSCalDualObj (lorads_alg_common.c:319-340) */
    int (*update_dual_var)(void *ctx, double rho);
    int (*cal_dual_obj)(void *ctx, double *dobj);
    /* state movers */
    int (*alm_to_admm)(void *ctx);                   /* R -> V -> U   (data/lorads_solver.c:968-983) */
    int (*average_uv_to_v)(void *ctx);               /* R=(U+V)/2; V=R (main.c:441-448) */
    int (*scale_obj)(void *ctx, double s);           /* C *= s, lambda *= s (data/lorads_solver.c:1040-1052) */
    int (*resize_rank)(void *ctx, const int *new_rank); /* AUG_RANK (data/lorads_solver.c:806-906) */
    int (*set_mat)(void *ctx, int which, int blk, const double *colmajor);
    int (*get_mat)(void *ctx, int which, int blk, double *colmajor);
    int (*set_vec)(void *ctx, int which, const double *v);
    int (*get_vec)(void *ctx, int which, double *v);
    int (*set_allreduce)(void *ctx, lrd_allreduce_fn fn, void *user);
    void (*destroy)(void *ctx);
    /* OPTIONAL (may be NULL): admm_update_var + cal_obj(UV) + cal_dual_obj + update_dimacs(UV) in one
     * call, same order and results (lorads_admm.c:76-81); out = {cg iterations, pobj, dobj, err1} */
    int (*admm_step)(void *ctx, double rho, double cg_tol, int cg_max_iter, double out[4]);
    /* OPTIONAL (may be NULL): calculate_dual_infeasibility_solver without its two divisions
     * (data/lorads_solver.c:1007-1033): sum over this table's cones of |min(lambda_min(C_k - A_k^*(lambda)), 0)| */
    int (*dual_infeasibility)(void *ctx, double *sum_neg_eig);
    /* OPTIONAL pair (both or neither): the phase-1 inner iteration in two calls with one host round trip each
     * (lorads_alm.c:1066-1131).  alm_front = lbfgs_direction(inner) + alm_q12p12 + alm_linesearch_coeffs,
     * out = {p1, p2, a, b, c, d};  alm_step = set_y_as_neg_grad + alm_update_var(tau) + alm_cal_grad(rho) +
     * set_lbfgs_his_two(tau) + update_dimacs(RR), then alm_front(next_inner) when next_inner >= 0,
     * out = {lagNormSq, err1, p1, p2, a, b, c, d}.  Same results as the separate slots. */
    int (*alm_front)(void *ctx, double rho, int inner, double out[6]);
    int (*alm_step)(void *ctx, double rho, double tau, int next_inner, double out[8]);
} lrd_backend;

/* iteration states, as the reference's lorads_alm_state / lorads_admm_state
 * (data/def_lorads_solver.h:130-161) */
typedef struct {
    int outerIter, innerIter;
    double rho, l_inf_primal_infeasibility, l_1_primal_infeasibility, l_2_primal_infeasibility;
    double primal_dual_gap, primal_objective_value, dual_objective_value;
    double l_inf_dual_infeasibility, l_1_dual_infeasibility, l_2_dual_infeasibility, tau;
} lrd_alm_state;

typedef struct {
    int iter, nBlks, cg_iter;
    double rho, l_1_dual_infeasibility, l_inf_dual_infeasibility, l_1_primal_infeasibility;
    double l_inf_primal_infeasibility, l_2_primal_infeasibility, l_2_dual_infeasibility;
    double primal_objective_value, dual_objective_value, primal_dual_gap;
} lrd_admm_state;

typedef struct {
    lrd_problem *prob;
    lrd_backend *be;
    lrd_alm_state alm;
    lrd_admm_state admm;
    double pObjVal, dObjVal, err_constr_l1, err_pdgap; /* dimacError[0], [1] */
    double err_dual_l1;   /* dimacError[LORADS_DIMAC_ERROR_DUALFEASIBLE_L1]; -1 = not evaluated */
    double t_dual_infeas; /* seconds spent evaluating it (main.c all_dual_infea) */
    double scaleObjHis;
    int cgIter;     /* cumulative CG iterations of the current ADMM call (ASolver->cgIter) */
    int max_alm_sub_iter; /* the reference's global MAX_ALM_SUB_ITER (lorads_alm.c:7) */
    int status;
    int *rank;      /* [nblk] current ranks */
    lrd_allreduce_fn allreduce;
    void *allreduce_user;
    double t_alm, t_admm;
    int admm_iters_first, cg_iters_first;
    int use_fused_step; /* 1: use lrd_backend.admm_step when the table has it */
    int be_fail;        /* a table slot returned non-zero (device error, refused resize, failed all-reduce): the loops
                         * stop with LRD_RET_NUM_ERR instead of steering on numbers nobody produced */
} lrd_solver;

/* ---- params / problem ---- */
void lrd_params_default(lrd_params *p);
int lrd_params_set(lrd_params *p, const char *key, const char *val); /* reference CLI names, main.c:57-80 */
/* SDPA sparse reader (own implementation; conventions of src_semi/io/lorads_file_io.c:21-293). */
int lrd_read_sdpa(const char *fname, lrd_problem **out);
/* one entry line of an SDPA file, "mat blk i j value": returns the number of fields found (5 = a full entry), exactly as
 * sscanf("%d %d %d %d %lg") would; the value is converted as strtod converts it (problem.c; exported for the tests) */
int lrd_parse_entry_line(const char *line, int ij[4], double *val);
/* digest of the whole image (dimensions, every array, norms): equal digests = the same problem */
uint64_t lrd_problem_digest(const lrd_problem *p);
/* 1 when the start point is drawn by the inline copy of glibc's rand() recurrence (verified against rand() at run time), 0 when by rand() */
int lrd_start_generator_is_inline(void);

/* ---- shmx.c: sums of a few doubles (n <= 16) over the ranks of one node through POSIX shared memory; the scalar exchange of
 * separable shards (include/lorads_hip.h: lorads_hip_set_scalar_exchange).  name: "/..." -- the same on every rank, unique per run;
 * rank 0 creates and removes the segment.  All return 0 on success. */
typedef struct lrd_shmx lrd_shmx;
int lrd_shmx_open(const char *name, int world, int rank, lrd_shmx **out);
int lrd_shmx_allreduce(lrd_shmx *x, double *v, int n);
int lrd_shmx_hook(void *user, double *vals, int32_t n);
void lrd_shmx_close(lrd_shmx *x);
/* Build a problem from arrays (bench / tests; 0-based mat: 0 = F0, blk, row, col); same
 * post-processing as the reader (F0 negated, lower triangle, tiny entries dropped, pre-solve). */
int lrd_problem_from_triplets(int m, const double *b, int nblk, const int *dims, int64_t nent, const int *e_mat,
                              const int *e_blk, const int *e_row, const int *e_col, const double *e_val,
                              lrd_problem **out);
/* multi-GPU sharding: keep only the blocks with keep[k] != 0 (global consts stay global) */
void lrd_problem_select(lrd_problem *p, const int *keep);
int lrd_problem_localize(lrd_problem *p, int world, int rank_id);
void lrd_problem_free(lrd_problem *p);
/* rank rule, data/lorads_solver.c:290-319 */
void lrd_determine_rank(lrd_problem *p, double times_log_rank);
/* start point: srand(925) and the reference's draw order (data/lorads_solver.c:361-371,415,652-653):
 * R for every block, then U,V per block; call BEFORE lrd_problem_select so that every rank draws
 * the whole sequence; returns malloc'd col-major arrays indexed by block. */
int lrd_init_point(const lrd_problem *p, double ***R, double ***U, double ***V);
void lrd_free_point(int nblk, double **R, double **U, double **V);

/* ---- solver ---- */
int lrd_solver_init(lrd_solver *s, lrd_problem *prob, lrd_backend *be, const lrd_params *par);
void lrd_solver_clear(lrd_solver *s);
int lrd_alm_optimize(lrd_params *par, lrd_solver *s, int reopt_variant, int early_stop, double rho_update_factor,
                     double t_start);
int lrd_admm_optimize(lrd_params *par, lrd_solver *s, int reopt_variant, int iter_ceiling, double t_start);
void lrd_alm_to_admm(lrd_params *par, lrd_solver *s);
double lrd_reopt(lrd_params *par, lrd_solver *s, double reopt_param, int reopt_alm_iter, int reopt_admm_iter,
                 double t_start, int *admm_bad_iter_flag, int reopt_level);
/* whole solve = reference main.c:321-398 (dual infeasibility / level-2 reopt need the ARPACK step
 * and are "next") */
int lrd_solve(lrd_params *par, lrd_solver *s);
int lrd_dual_infeasibility(lrd_solver *s); /* data/lorads_solver.c:1007-1037 through the table's optional slot */

/* scalar helpers of the line search (lorads_alm.c:102-228) */
int lrd_cubic_roots(double a, double b, double c, double d, double res[3]);
int lrd_linesearch_tau(const double coef[4], double *tau);
double lrd_time(void);

#ifdef __cplusplus
}
#endif
#endif
