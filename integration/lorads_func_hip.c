/* lorads_func_hip.c -- the reference-side shim: LoRADS' own outer loops drive the MI355X library.
 *
 * What it is.  The reference reaches its per-iteration path through the operator table `lorads_func`
 * (src_semi/data/def_lorads_solver.h:109-127, filled by LORADSInitFuncSet, src_semi/data/lorads_solver.c:717-756) and
 * through a handful of direct calls next to it.  This file provides
 *   (1) a table whose every slot forwards to one entry point of include/lorads_hip.h, and
 *   (2) replacements, under the reference's own symbol names, for LORADSInitFuncSet and for the direct calls that touch
 *       state which now lives on the device (listed below).
 * Built as a shared object that is linked (or LD_PRELOADed) AHEAD of the reference's objects, the dynamic linker binds the
 * reference's calls to these definitions (ordinary ELF symbol interposition: the reference is position-independent code with
 * default visibility), so LORADS_ALMOptimize, LORADSADMMOptimize, their _reopt variants, reopt() and main() run UNCHANGED --
 * not one line of the reference is edited.  (A maintainer who prefers a compile-time switch adds one branch to
 * LORADSInitFuncSet and an `if (use_hip)` at the call sites listed below; the functions to call are the ones in this file.)
 *
 * Built only in the build container, against the reference's headers where they lie (oracle/Makefile, target ref_hip ->
 * oracle/_ref_hip/, git-ignored); tests/test_reference_shim.py runs the reference's own loops through it on the GPU.
 * This file is ours; it includes the reference's headers and copies none of its code.
 *
 * Symbols taken over (each keeps the reference's signature; "orig" = the reference's definition, reached with
 * dlsym(RTLD_NEXT) where its host-side bookkeeping is still wanted):
 *   LORADSInitFuncSet                         -> the table below (one set serves SDP-only and SDP+LP problems)
 *   LORADSInitConstrValAll / ..Sum            called directly by the ADMM prologues (lorads_admm.c:47-48,177-178)
 *   LORADSUpdateDualVar, LORADSCalDualObj     lorads_alg_common.c:319-340 (callers lorads_alm.c:1151,1201,1240; lorads_admm.c:51,80,120)
 *   ALMLineSearch                             lorads_alm.c:161-228: the m-vector sums come from the device, the cubic is the
 *                                             reference's own LORADScubic_equation
 *   LORADS_ALMtoADMM, objScale_dualvar, AUG_RANK   data/lorads_solver.c:968,1040,806: orig (host state, scalars, rankElem) + device
 * The host arrays of lorads_variable (R, U, V, Grad, dualVar, constrValSum, ...) are NOT kept current during a solve; the scalars
 * the loops steer on (pObjVal, dObjVal, dimacError[], cgIter) are.  lorads_func_hip_download() brings factors and multipliers
 * back (end of solve, or before any reference routine that reads them on the host, e.g. the ARPACK dual-infeasibility check).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "def_lorads_solver.h"
#include "lorads_solver.h"
#include "lorads_alm.h"
#include "lorads_admm.h"
#include "lorads_alg_common.h"
#include "lorads_utils.h"

#include "lorads_hip.h"

static lorads_hip_ctx *g_hip;      /* one context per process / GPU */
static lorads_solver *g_solver;    /* the solver it was created for */
static int g_pair = LORADS_HIP_PAIR_RR;
static int g_fail;

/* the reference's slots return void and its loops cannot react to a failure: a failed device call ends the process */
#define HIPCALL(call)                                                                              \
    do {                                                                                           \
        if ((call) != 0) {                                                                         \
            fprintf(stderr, "lorads_func_hip: %s failed: %s\n", #call, lorads_hip_last_error()); \
            g_fail = 1;                                                                            \
            exit(3);                                                                               \
        }                                                                                          \
    } while (0)

int lorads_func_hip_failed(void) { return g_fail; }

/* ------------------------------------------------------------------ flattening */
typedef struct {
    int32_t *ptr, *row, *col, *idx;
    double *val;
    int32_t nrow, nnz, cap;
} flat_csr;

static void push(flat_csr *f, int32_t r, int32_t c, double v) {
    if (v == 0.0) return;
    if (f->nnz == f->cap) {
        f->cap = f->cap ? 2 * f->cap : 1024;
        f->row = (int32_t *)realloc(f->row, sizeof(int32_t) * (size_t)f->cap);
        f->col = (int32_t *)realloc(f->col, sizeof(int32_t) * (size_t)f->cap);
        f->val = (double *)realloc(f->val, sizeof(double) * (size_t)f->cap);
    }
    f->row[f->nnz] = r >= c ? r : c; /* lower triangle: row >= col */
    f->col[f->nnz] = r >= c ? c : r;
    f->val[f->nnz] = v;
    f->nnz++;
}

/* appends the lower-triangular entries of one coefficient matrix (data/def_lorads_sdp_data.h:68-88: zero / sparse triplets /
 * dense packed lower column-major, index n j - j (j + 1) / 2 + i, lorads_utils.h:45-47) */
static void push_coeff(flat_csr *f, const sdp_coeff *a) {
    if (a->dataType == SDP_COEFF_SPARSE) {
        const sdp_coeff_sparse *s = (const sdp_coeff_sparse *)a->dataMat;
        for (lorads_int k = 0; k < s->nTriMatElem; ++k) push(f, (int32_t)s->triMatRow[k], (int32_t)s->triMatCol[k], s->triMatElem[k]);
    } else if (a->dataType == SDP_COEFF_DENSE) {
        const sdp_coeff_dense *d = (const sdp_coeff_dense *)a->dataMat;
        const lorads_int n = d->nSDPCol;
        for (lorads_int j = 0; j < n; ++j)
            for (lorads_int i = j; i < n; ++i) push(f, (int32_t)i, (int32_t)j, d->dsMatElem[n * j - j * (j + 1) / 2 + i]);
    }
}

typedef struct {
    flat_csr a, c;
} flat_cone;

/* what AConeProcData / the pre-solve left in lorads_cone_sdp_dense / _sparse (data/def_lorads_sdp_conic.h:101-127) as CSR by
 * constraint; constraints whose coefficient on this cone is the zero matrix are left out (row_idx names the others) */
static void flatten_cone(const lorads_sdp_cone *cone, int32_t rank, flat_cone *fc, lorads_hip_block *out) {
    memset(fc, 0, sizeof *fc);
    lorads_int nrow_all, ncol;
    sdp_coeff **rows;
    const sdp_coeff *obj;
    const lorads_int *row_idx = NULL;
    if (cone->type == LORADS_CONETYPE_SPARSE_SDP) {
        const lorads_cone_sdp_sparse *s = (const lorads_cone_sdp_sparse *)cone->coneData;
        nrow_all = s->nRowElem; ncol = s->nCol; rows = s->sdpRow; obj = s->sdpObj; row_idx = s->rowIdx;
    } else {
        const lorads_cone_sdp_dense *d = (const lorads_cone_sdp_dense *)cone->coneData;
        nrow_all = d->nRow; ncol = d->nCol; rows = d->sdpRow; obj = d->sdpObj;
    }
    flat_csr *A = &fc->a;
    A->ptr = (int32_t *)calloc((size_t)nrow_all + 1, sizeof(int32_t));
    A->idx = (int32_t *)calloc((size_t)nrow_all + 1, sizeof(int32_t));
    for (lorads_int i = 0; i < nrow_all; ++i) {
        const int32_t before = A->nnz;
        push_coeff(A, rows[i]);
        if (A->nnz == before) continue;
        A->idx[A->nrow] = (int32_t)(row_idx ? row_idx[i] : i);
        A->ptr[++A->nrow] = A->nnz;
    }
    push_coeff(&fc->c, obj);
    memset(out, 0, sizeof *out);
    out->n = (int32_t)ncol; out->rank = rank; out->nrow = A->nrow; out->row_idx = A->idx; out->a_ptr = A->ptr;
    out->a_row = A->row; out->a_col = A->col; out->a_val = A->val;
    out->c_nnz = fc->c.nnz; out->c_row = fc->c.row; out->c_col = fc->c.col; out->c_val = fc->c.val;
}

/* the LP block as one more diagonal block (lorads_hip.h: is_lp): constraint i holds (col, col, a_i,col) for every column with
 * a coefficient in row i (lp_coeff per column, data/def_lorads_lp_data.h), objective (col, col, objMatElem[col]) */
static void flatten_lp(const lorads_lp_cone *lp, lorads_int m, flat_cone *fc, lorads_hip_block *out) {
    memset(fc, 0, sizeof *fc);
    const lorads_lp_cone_data *d = lp->coneData;
    const lorads_int ncol = d->nCol;
    /* gather (row, col, a) from the per-column coefficients, then bucket by constraint row */
    int32_t *cnt = (int32_t *)calloc((size_t)m + 1, sizeof(int32_t));
    for (int pass = 0; pass < 2; ++pass) {
        flat_csr *A = &fc->a;
        if (pass == 1) {
            A->ptr = (int32_t *)calloc((size_t)m + 1, sizeof(int32_t));
            for (lorads_int i = 0; i < m; ++i) A->ptr[i + 1] = A->ptr[i] + cnt[i];
            A->nnz = A->cap = A->ptr[m];
            A->row = (int32_t *)malloc(sizeof(int32_t) * (size_t)(A->nnz + 1));
            A->col = (int32_t *)malloc(sizeof(int32_t) * (size_t)(A->nnz + 1));
            A->val = (double *)malloc(sizeof(double) * (size_t)(A->nnz + 1));
            memset(cnt, 0, sizeof(int32_t) * (size_t)(m + 1));
        }
        for (lorads_int j = 0; j < ncol; ++j) {
            const lp_coeff *cf = d->lpCol[j];
            if (cf->dataType == LP_COEFF_SPARSE) {
                const lp_coeff_sparse *s = (const lp_coeff_sparse *)cf->dataMat;
                for (lorads_int k = 0; k < s->nnz; ++k) {
                    const lorads_int i = s->rowPtr[k];
                    if (s->val[k] == 0.0) continue;
                    if (pass == 1) { const int32_t t = A->ptr[i] + cnt[i]; A->row[t] = A->col[t] = (int32_t)j; A->val[t] = s->val[k]; }
                    cnt[i]++;
                }
            } else if (cf->dataType == LP_COEFF_DENSE) {
                const lp_coeff_dense *s = (const lp_coeff_dense *)cf->dataMat;
                for (lorads_int i = 0; i < s->nRows; ++i) {
                    if (s->val[i] == 0.0) continue;
                    if (pass == 1) { const int32_t t = A->ptr[i] + cnt[i]; A->row[t] = A->col[t] = (int32_t)j; A->val[t] = s->val[i]; }
                    cnt[i]++;
                }
            }
        }
    }
    free(cnt);
    flat_csr *A = &fc->a;
    A->idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(m + 1));
    for (lorads_int i = 0; i < m; ++i) A->idx[i] = (int32_t)i;
    A->nrow = (int32_t)m;
    for (lorads_int j = 0; j < ncol; ++j) push(&fc->c, (int32_t)j, (int32_t)j, d->objMatElem[j]);
    memset(out, 0, sizeof *out);
    out->n = (int32_t)ncol; out->rank = 1; out->nrow = A->nrow; out->row_idx = A->idx; out->a_ptr = A->ptr;
    out->a_row = A->row; out->a_col = A->col; out->a_val = A->val;
    out->c_nnz = fc->c.nnz; out->c_row = fc->c.row; out->c_col = fc->c.col; out->c_val = fc->c.val;
    out->is_lp = 1;
}

static void free_flat(flat_cone *fc) {
    free(fc->a.ptr); free(fc->a.row); free(fc->a.col); free(fc->a.val); free(fc->a.idx);
    free(fc->c.ptr); free(fc->c.row); free(fc->c.col); free(fc->c.val); free(fc->c.idx);
}

static int upload_factors(lorads_solver *S) {
    int rc = 0;
    for (lorads_int k = 0; !rc && k < S->nCones; ++k) {
        rc |= lorads_hip_set_mat(g_hip, LORADS_HIP_MAT_R, (int32_t)k, S->var->R[k]->matElem);
        rc |= lorads_hip_set_mat(g_hip, LORADS_HIP_MAT_U, (int32_t)k, S->var->U[k]->matElem);
        rc |= lorads_hip_set_mat(g_hip, LORADS_HIP_MAT_V, (int32_t)k, S->var->V[k]->matElem);
    }
    if (!rc && S->nLpCols > 0) { /* the LP vectors are the LP block's n x 1 "factors" */
        rc |= lorads_hip_set_mat(g_hip, LORADS_HIP_MAT_R, (int32_t)S->nCones, S->var->rLp->matElem);
        rc |= lorads_hip_set_mat(g_hip, LORADS_HIP_MAT_U, (int32_t)S->nCones, S->var->uLp->matElem);
        rc |= lorads_hip_set_mat(g_hip, LORADS_HIP_MAT_V, (int32_t)S->nCones, S->var->vLp->matElem);
    }
    if (!rc) rc = lorads_hip_set_vec(g_hip, LORADS_HIP_VEC_LAMBDA, S->var->dualVar);
    return rc;
}

/* creates the context for this solver at the first table call that sees it (after LORADSPreprocess, LORADSDetermineRank and
 * the start point, main.c:266-320): flat problem, start point R/U/V, multipliers */
static int attach(lorads_solver *S) {
    if (g_hip && g_solver == S) return 0;
    if (g_hip) { lorads_hip_destroy(g_hip); g_hip = NULL; }
    const int nb = (int)S->nCones + (S->nLpCols > 0 ? 1 : 0);
    lorads_hip_block *blk = (lorads_hip_block *)calloc((size_t)nb + 1, sizeof *blk);
    flat_cone *fc = (flat_cone *)calloc((size_t)nb + 1, sizeof *fc);
    for (lorads_int k = 0; k < S->nCones; ++k) flatten_cone(S->SDPCones[k], (int32_t)S->var->rankElem[k], &fc[k], &blk[k]);
    if (S->nLpCols > 0) flatten_lp(S->lpCone, S->nRows, &fc[S->nCones], &blk[S->nCones]);
    lorads_hip_problem prob;
    memset(&prob, 0, sizeof prob);
    prob.m = (int32_t)S->nRows; prob.b = S->rowRHS; prob.b_nrm1 = S->bRHSNrm1; prob.nblocks = nb; prob.blocks = blk;
    prob.lbfgs_len = (int32_t)S->hisRecT; prob.device = -1;
    int rc = lorads_hip_create(&prob, &g_hip);
    for (int k = 0; k < nb; ++k) free_flat(&fc[k]);
    free(fc); free(blk);
    if (rc) {
        /* no device, no library: there is no CPU path behind this table, and the reference's loops would spin on slots that
         * compute nothing (its kernels return void, lorads_utils.h:11) -- stop here, loudly */
        fprintf(stderr, "lorads_func_hip: lorads_hip_create failed: %s -- the MI355X backend is required, aborting\n", lorads_hip_last_error());
        exit(3);
    }
    g_solver = S;
    rc = upload_factors(S);
    if (rc) { fprintf(stderr, "lorads_func_hip: upload failed: %s\n", lorads_hip_last_error()); g_fail = 1; }
    return rc;
}

/* factors and multipliers back into the reference's host arrays (column-major matElem, as it keeps them) */
int lorads_func_hip_download(lorads_solver *S) {
    if (!g_hip || g_solver != S) return 1;
    int rc = 0;
    for (lorads_int k = 0; !rc && k < S->nCones; ++k) {
        rc |= lorads_hip_get_mat(g_hip, LORADS_HIP_MAT_R, (int32_t)k, S->var->R[k]->matElem);
        rc |= lorads_hip_get_mat(g_hip, LORADS_HIP_MAT_U, (int32_t)k, S->var->U[k]->matElem);
        rc |= lorads_hip_get_mat(g_hip, LORADS_HIP_MAT_V, (int32_t)k, S->var->V[k]->matElem);
    }
    if (!rc && S->nLpCols > 0) {
        rc |= lorads_hip_get_mat(g_hip, LORADS_HIP_MAT_R, (int32_t)S->nCones, S->var->rLp->matElem);
        rc |= lorads_hip_get_mat(g_hip, LORADS_HIP_MAT_U, (int32_t)S->nCones, S->var->uLp->matElem);
        rc |= lorads_hip_get_mat(g_hip, LORADS_HIP_MAT_V, (int32_t)S->nCones, S->var->vLp->matElem);
    }
    if (!rc) rc |= lorads_hip_get_vec(g_hip, LORADS_HIP_VEC_LAMBDA, S->var->dualVar);
    if (!rc) rc |= lorads_hip_get_vec(g_hip, LORADS_HIP_VEC_CONSTR_SUM, S->var->constrValSum);
    return rc;
}

void lorads_func_hip_release(void) {
    if (g_hip) lorads_hip_destroy(g_hip);
    g_hip = NULL; g_solver = NULL;
}

/* ------------------------------------------------------------------ the table */
static void hipInitConstrValAll(lorads_solver *S, lorads_lp_dense *a, lorads_lp_dense *b, lorads_sdp_dense **X, lorads_sdp_dense **Y) {
    (void)a; (void)b; (void)Y;
    if (attach(S)) return;
    g_pair = (X == S->var->R) ? LORADS_HIP_PAIR_RR : LORADS_HIP_PAIR_UV; /* remembered for ..Sum, which always follows */
}
static void hipInitConstrValSum(lorads_solver *S) {
    if (attach(S)) return;
    HIPCALL(lorads_hip_init_constr(g_hip, g_pair));
}
static void hipALMCalGrad(lorads_solver *S, lorads_lp_dense *a, lorads_lp_dense *b, lorads_sdp_dense **R, lorads_sdp_dense **G,
                          double *lagNormSq, double rho) {
    (void)a; (void)b; (void)R; (void)G;
    if (attach(S)) return;
    HIPCALL(lorads_hip_alm_cal_grad(g_hip, rho, lagNormSq));
}
static void hipLBFGSDirection(lorads_params *p, lorads_solver *S, lbfgs_node *h, lorads_lp_dense *a, lorads_lp_dense *b,
                              lorads_sdp_dense **G, lorads_sdp_dense **D, lorads_int innerIter) {
    (void)p; (void)h; (void)a; (void)b; (void)G; (void)D;
    if (attach(S)) return;
    HIPCALL(lorads_hip_lbfgs_direction(g_hip, (int32_t)innerIter)); /* includes LBFGSDirUseGrad */
}
static void hipLBFGSDirUseGrad(lorads_solver *S, lorads_lp_dense *a, lorads_lp_dense *b, lorads_sdp_dense **D, lorads_sdp_dense **G) {
    (void)S; (void)a; (void)b; (void)D; (void)G; /* done inside lorads_hip_lbfgs_direction */
}
static void hipCopyRtoV(lorads_lp_dense *r, lorads_lp_dense *v, lorads_sdp_dense **R, lorads_sdp_dense **V, lorads_int n) {
    (void)r; (void)v; (void)R; (void)V; (void)n;
    /* main.c:441-448 calls averageUV + copyRtoV before the dual-infeasibility round */
    if (g_hip) HIPCALL(lorads_hip_average_uv_to_v(g_hip));
}
static void hipALMCalq12p12(lorads_solver *S, lorads_lp_dense *a, lorads_lp_dense *b, lorads_sdp_dense **R, lorads_sdp_dense **D,
                            double *q1, double *q2, double *p12) {
    (void)a; (void)b; (void)R; (void)D; (void)q1; (void)q2; /* q1, q2 stay on the device */
    if (attach(S)) return;
    HIPCALL(lorads_hip_alm_q12p12(g_hip, p12));
}
static void hipSetAsNegGrad(lorads_solver *S, lorads_lp_dense *a, lorads_sdp_dense **G) {
    (void)a; (void)G;
    if (attach(S)) return;
    HIPCALL(lorads_hip_set_y_as_neg_grad(g_hip));
}
static void hipALMupdateVar(lorads_solver *S, lorads_lp_dense *a, lorads_lp_dense *b, lorads_sdp_dense **R, lorads_sdp_dense **D, double tau) {
    (void)a; (void)b; (void)R; (void)D;
    if (attach(S)) return;
    HIPCALL(lorads_hip_alm_update_var(g_hip, tau)); /* incl. constrValSum += tau q1 + tau^2 q2 (lorads_alm.c:1122-1124) */
}
static void hipSetlbfgsHisTwo(lorads_solver *S, lorads_lp_dense *a, lorads_lp_dense *b, lorads_sdp_dense **G, lorads_sdp_dense **D, double tau) {
    (void)a; (void)b; (void)G; (void)D;
    if (attach(S)) return;
    HIPCALL(lorads_hip_set_lbfgs_his_two(g_hip, tau));
}
static void hipUpdateDimacs(lorads_solver *S, int pair) {
    if (attach(S)) return;
    double e = 0.0;
    HIPCALL(lorads_hip_update_dimacs(g_hip, pair, &e));
    S->dimacError[LORADS_DIMAC_ERROR_CONSTRVIO_L1] = e;
    const double gap = S->pObjVal - S->dObjVal; /* lorads_alg_common.c:273-275 */
    S->dimacError[LORADS_DIMAC_ERROR_PDGAP] = fabs(gap) / (1 + fabs(S->pObjVal) + fabs(S->dObjVal));
}
static void hipUpdateDimacsALM(lorads_solver *S, lorads_sdp_dense **a, lorads_sdp_dense **b, lorads_lp_dense *c, lorads_lp_dense *d) {
    (void)a; (void)b; (void)c; (void)d;
    hipUpdateDimacs(S, LORADS_HIP_PAIR_RR);
}
static void hipUpdateDimacsADMM(lorads_solver *S, lorads_sdp_dense **a, lorads_sdp_dense **b, lorads_lp_dense *c, lorads_lp_dense *d) {
    (void)a; (void)b; (void)c; (void)d;
    hipUpdateDimacs(S, LORADS_HIP_PAIR_UV);
}
static void hipCalObj(lorads_solver *S, int pair) {
    if (attach(S)) return;
    double v = 0.0;
    HIPCALL(lorads_hip_cal_obj(g_hip, pair, &v));
    S->pObjVal = v / S->scaleObjHis;
}
static void hipCalObjALM(lorads_solver *S) { hipCalObj(S, LORADS_HIP_PAIR_RR); }
static void hipCalObjADMM(lorads_solver *S) { hipCalObj(S, LORADS_HIP_PAIR_UV); }
static void hipAdmmUpdateVar(lorads_solver *S, double rho, double tol, lorads_int maxIter) {
    if (attach(S)) return;
    int32_t it = 0;
    HIPCALL(lorads_hip_admm_update_var(g_hip, rho, tol, (int32_t)maxIter, &it));
    S->cgIter += it;
}

/* ------------------------------------------------------------------ symbols taken over from the reference */
void LORADSInitFuncSet(lorads_func **pfunc, lorads_int nLpCols) {
    (void)nLpCols; /* the device handles an LP block internally: ONE set serves both of the reference's */
    lorads_func *f = (lorads_func *)calloc(1, sizeof *f);
    f->InitConstrValAll = hipInitConstrValAll;   f->InitConstrValSum = hipInitConstrValSum;
    f->ALMCalGrad = hipALMCalGrad;               f->LBFGSDirection = hipLBFGSDirection;
    f->LBFGSDirUseGrad = hipLBFGSDirUseGrad;     f->copyRtoV = hipCopyRtoV;
    f->ALMCalq12p12 = hipALMCalq12p12;           f->setAsNegGrad = hipSetAsNegGrad;
    f->ALMupdateVar = hipALMupdateVar;           f->setlbfgsHisTwo = hipSetlbfgsHisTwo;
    f->updateDimacsALM = hipUpdateDimacsALM;     f->updateDimacsADMM = hipUpdateDimacsADMM;
    f->calObj_admm = hipCalObjADMM;              f->calObj_alm = hipCalObjALM;
    f->admmUpdateVar = hipAdmmUpdateVar;
    *pfunc = f;
}

/* the ADMM prologues call these two directly instead of through the table (lorads_admm.c:47-48,177-178) */
void LORADSInitConstrValAll(lorads_solver *S, lorads_lp_dense *u, lorads_lp_dense *v, lorads_sdp_dense **U, lorads_sdp_dense **V) {
    hipInitConstrValAll(S, u, v, U, V);
}
void LORADSInitConstrValSum(lorads_solver *S) { hipInitConstrValSum(S); }

void LORADSUpdateDualVar(lorads_solver *S, double rho) {
    if (attach(S)) return;
    HIPCALL(lorads_hip_update_dual_var(g_hip, rho));
}
void LORADSCalDualObj(lorads_solver *S) {
    if (attach(S)) return;
    double v = 0.0;
    HIPCALL(lorads_hip_cal_dual_obj(g_hip, &v));
    S->dObjVal = v / S->scaleObjHis;
}

static double quartic(double a, double b, double c, double d, double x) { return a * pow(x, 4) + b * pow(x, 3) + c * pow(x, 2) + d * x; }
/* lorads_alm.c:161-228.  The five m-vector sums behind a, b, c, d come from the device (q0 = b - constrValSum, q1, q2 and
 * lambda live there; the host arrays handed in are not read); the cubic is solved by the reference's own LORADScubic_equation
 * and the candidate selection below follows :173-227 (argmin over {0, 1, roots in (1e-20, 1]}, later candidates win ties) */
lorads_int ALMLineSearch(double rho, lorads_int n, double *lambd, double p1, double p2, double *q0, double *q1, double *q2, double *tau) {
    (void)n; (void)lambd; (void)q0; (void)q1; (void)q2;
    double k[4] = {0, 0, 0, 0};
    if (!g_hip) return 0;
    HIPCALL(lorads_hip_alm_linesearch_coeffs(g_hip, rho, p1, p2, k));
    double roots[3] = {0.0, 0.0, 0.0};
    const lorads_int nr = LORADScubic_equation(4 * k[0], 3 * k[1], 2 * k[2], k[3], roots);
    double f[5] = {0.0, quartic(k[0], k[1], k[2], k[3], 1.0), 1e+30, 1e+30, 1e+30};
    const double cand[5] = {0.0, 1.0, roots[0], roots[1], roots[2]};
    if (nr >= 1 && roots[0] > 1e-20 && roots[0] <= 1.0) f[2] = quartic(k[0], k[1], k[2], k[3], roots[0]);
    if (nr >= 2 && roots[1] > 1e-20 && roots[1] <= 1.0) f[3] = quartic(k[0], k[1], k[2], k[3], roots[1]);
    if (nr == 3 && roots[2] > 1e-20 && roots[2] <= 1.0) f[4] = quartic(k[0], k[1], k[2], k[3], roots[2]);
    double fmin = f[0];
    for (int i = 1; i < 5; ++i) fmin = f[i] < fmin ? f[i] : fmin;
    for (int i = 0; i < 5; ++i)
        if (fabs(fmin - f[i]) < 1e-10) tau[0] = cand[i];
    return nr;
}

void LORADS_ALMtoADMM(lorads_solver *S, lorads_params *params, lorads_alm_state *alm, lorads_admm_state *admm) {
    static void (*orig)(lorads_solver *, lorads_params *, lorads_alm_state *, lorads_admm_state *);
    if (!orig) *(void **)(&orig) = dlsym(RTLD_NEXT, "LORADS_ALMtoADMM");
    orig(S, params, alm, admm); /* the state hand-over (rho, errors) and the host copies, data/lorads_solver.c:968-1004 */
    if (attach(S)) return;
    HIPCALL(lorads_hip_alm_to_admm(g_hip));
}

void objScale_dualvar(lorads_solver *S, double *scaleTemp, double *scaleHis) {
    static void (*orig)(lorads_solver *, double *, double *);
    if (!orig) *(void **)(&orig) = dlsym(RTLD_NEXT, "objScale_dualvar");
    orig(S, scaleTemp, scaleHis); /* scaleObjHis and the host copies of C and lambda, data/lorads_solver.c:1040-1052 */
    if (attach(S)) return;
    HIPCALL(lorads_hip_scale_obj(g_hip, scaleTemp[0]));
}

lorads_int AUG_RANK(lorads_solver *S, lorads_int *BlkDims, lorads_int nBlks, double aug_factor) {
    static lorads_int (*orig)(lorads_solver *, lorads_int *, lorads_int, double);
    if (!orig) *(void **)(&orig) = dlsym(RTLD_NEXT, "AUG_RANK");
    if (attach(S)) return orig(S, BlkDims, nBlks, aug_factor);
    const int nb = (int)S->nCones + (S->nLpCols > 0 ? 1 : 0);
    int32_t *nr = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nb + 1));
    for (lorads_int k = 0; k < S->nCones; ++k) nr[k] = (int32_t)S->var->U[k]->rank;
    const lorads_int is_max = orig(S, BlkDims, nBlks, aug_factor); /* new rankElem[], host buffers, CG workspaces (:806-906) */
    int grown = 0;
    for (lorads_int k = 0; k < S->nCones; ++k) {
        grown |= nr[k] != (int32_t)S->var->U[k]->rank;
        nr[k] = (int32_t)S->var->U[k]->rank;
    }
    if (S->nLpCols > 0) nr[S->nCones] = 1;
    /* (every cone already at its maximum: the reference returns before touching anything, :810-814, and so do we) */
    if (grown) HIPCALL(lorads_hip_resize_rank(g_hip, nr)); /* keeps the old columns, new ones as lpRandomDiag draws them (:776-786) */
    free(nr);
    return is_max;
}
