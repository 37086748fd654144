/* ref_driver.c -- drives the UNMODIFIED reference (oracle/_ref/liblorads_ref.so, compiled from
 * /root/reference/src_semi where it lies) to produce golden vectors for this repo's tests.
 *
 * TEST INFRASTRUCTURE.  This file is ours; it only CALLS the reference's public functions through
 * the reference's own headers (include path points into /root/reference at build time).  It is built
 * by oracle/Makefile into oracle/_ref/ref_driver and is never linked into the product.
 *
 * Two modes:
 *   solve  : same call sequence as the reference's main() (src_semi/main.c:266-342, 376-398) minus the
 *            ARPACK dual-infeasibility step (main.c:400; ARPACK is absent from this image, so the
 *            symbols dsaupd_/dseupd_ stay unresolved in the .so and are never called).  Prints the
 *            reference's own log lines and writes final scalars (+ optionally U,V,lambda) to a dump.
 *   trace  : scripted sequence of lorads_func-table calls (data/def_lorads_solver.h:109-127) with the
 *            inputs/outputs of every call dumped -- pins each function on the hot path.
 *
 * Dump container: repeated records  [int32 len][name bytes][int64 n][n doubles]  (little endian).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <math.h>

#include "lorads_file_io.h"
#include "def_lorads_user_data.h"
#include "lorads_user_data.h"
#include "lorads_utils.h"
#include "def_lorads_solver.h"
#include "lorads_solver.h"
#include "lorads_alm.h"
#include "lorads_admm.h"
#include "lorads_alg_common.h"
#include "lorads_vec_opts.h"

static FILE *g_dump = NULL;

static void rec(const char *name, const double *x, int64_t n) {
    if (!g_dump) return;
    int32_t len = (int32_t)strlen(name);
    fwrite(&len, 4, 1, g_dump);
    fwrite(name, 1, (size_t)len, g_dump);
    fwrite(&n, 8, 1, g_dump);
    if (n > 0) fwrite(x, 8, (size_t)n, g_dump);
}
static void rec1(const char *name, double v) { rec(name, &v, 1); }
static void recf(const char *fmt, int a, int b, const double *x, int64_t n) {
    char nm[128];
    snprintf(nm, sizeof nm, fmt, a, b);
    rec(nm, x, n);
}

static void default_params(lorads_params *p) {
    /* defaults of the reference CLI, main.c:19-43 */
    p->fname = "NULL"; p->initRho = 0.0; p->rhoMax = 5000.0; p->rhoCellingALM = 1e8;
    p->rhoCellingADMM = p->rhoMax * 200; p->maxALMIter = 200; p->maxADMMIter = 10000;
    p->timesLogRank = 2.0; p->rhoFreq = 5; p->rhoFactor = 1.2; p->ALMRhoFactor = 2.0;
    p->phase1Tol = 1e-3; p->phase2Tol = 1e-5; p->timeSecLimit = 3600.0; p->heuristicFactor = 1.0;
    p->lbfgsListLength = 2; p->endTauTol = 1e-16; p->endALMSubTol = 1e-10; p->l2Rescaling = false;
    p->reoptLevel = 2; p->dyrankLevel = 2; p->highAccMode = false;
}

static int parse_opts(int argc, char **argv, int first, lorads_params *p, int *n_alm, int *n_admm,
                      int *dump_state, char **uvfile, double *fixrho) {
    for (int i = first; i < argc; ++i) {
        const char *a = argv[i];
        if (strncmp(a, "--", 2) != 0 || i + 1 >= argc) { fprintf(stderr, "bad option %s\n", a); return 1; }
        const char *v = argv[++i];
        a += 2;
        if (!strcmp(a, "initRho")) p->initRho = atof(v);
        else if (!strcmp(a, "rhoMax")) p->rhoMax = atof(v);
        else if (!strcmp(a, "maxALMIter")) p->maxALMIter = atoi(v);
        else if (!strcmp(a, "maxADMMIter")) p->maxADMMIter = atoi(v);
        else if (!strcmp(a, "timesLogRank")) p->timesLogRank = atof(v);
        else if (!strcmp(a, "rhoFreq")) p->rhoFreq = atoi(v);
        else if (!strcmp(a, "rhoFactor")) p->rhoFactor = atof(v);
        else if (!strcmp(a, "ALMRhoFactor")) p->ALMRhoFactor = atof(v);
        else if (!strcmp(a, "phase1Tol")) p->phase1Tol = atof(v);
        else if (!strcmp(a, "phase2Tol")) p->phase2Tol = atof(v);
        else if (!strcmp(a, "timeSecLimit")) p->timeSecLimit = atof(v);
        else if (!strcmp(a, "heuristicFactor")) p->heuristicFactor = atof(v);
        else if (!strcmp(a, "lbfgsListLength")) p->lbfgsListLength = atoi(v);
        else if (!strcmp(a, "endTauTol")) p->endTauTol = atof(v);
        else if (!strcmp(a, "endALMSubTol")) p->endALMSubTol = atof(v);
        else if (!strcmp(a, "reoptLevel")) p->reoptLevel = atoi(v);
        else if (!strcmp(a, "dyrankLevel")) p->dyrankLevel = atoi(v);
        else if (!strcmp(a, "highAccMode")) p->highAccMode = atoi(v);
        else if (!strcmp(a, "nALM")) *n_alm = atoi(v);
        else if (!strcmp(a, "nADMM")) *n_admm = atoi(v);
        else if (!strcmp(a, "dumpState")) *dump_state = atoi(v);
        else if (!strcmp(a, "uv")) *uvfile = (char *)v;
        else if (!strcmp(a, "rho")) *fixrho = atof(v);
        else { fprintf(stderr, "unknown option --%s\n", a); return 1; }
    }
    p->rhoCellingADMM = p->rhoMax * 200; /* main.c:236 */
    return 0;
}

typedef struct {
    lorads_int nConstrs, nBlks, *BlkDims, nLpCols, nCols, nElem;
    double *rowRHS;
    lorads_int **coneMatBeg, **coneMatIdx;
    double **coneMatElem;
    lorads_int *LpMatBeg, *LpMatIdx;
    double *LpMatElem;
    user_data **SDPDatas;
    lorads_solver *S;
    lorads_alm_state alm;
    lorads_admm_state admm;
    SDPConst sdpConst;
} ctx_t;

static lorads_solver *g_S; /* for the LP vectors that travel with the factor arrays in the dumps */

/* setup = main.c:266-304 */
static int setup(ctx_t *c, lorads_params *p) {
    memset(c, 0, sizeof *c);
    if (LReadSDPA(p->fname, &c->nConstrs, &c->nBlks, &c->BlkDims, &c->rowRHS, &c->coneMatBeg, &c->coneMatIdx,
                  &c->coneMatElem, &c->nCols, &c->nLpCols, &c->LpMatBeg, &c->LpMatIdx, &c->LpMatElem,
                  &c->nElem) != LORADS_RETCODE_OK) {
        fprintf(stderr, "read failed\n");
        return 1;
    }
    if (c->nLpCols > 0 && !getenv("LORADS_REF_ALLOW_LP")) { fprintf(stderr, "LP block: out of scope\n"); return 1; }
    LORADS_INIT(c->S, lorads_solver, 1);
    LORADS_INIT(c->S->var, lorads_variable, 1);
    LORADSInitSolver(c->S, c->nConstrs, c->nBlks, c->BlkDims, c->nLpCols);
    LORADS_INIT(c->SDPDatas, user_data *, c->nBlks);
    LORADSSetDualObjective(c->S, c->rowRHS);
    LORADSInitConeData(c->S, c->SDPDatas, c->coneMatElem, c->coneMatBeg, c->coneMatIdx, c->BlkDims, c->nConstrs,
                       c->nBlks, c->nLpCols, c->LpMatBeg, c->LpMatIdx, c->LpMatElem);
    LORADSPreprocess(c->S, c->BlkDims);
    LORADSDetermineRank(c->S, c->BlkDims, p->timesLogRank);
    LORADSInitALMVars(c->S, c->S->var->rankElem, c->BlkDims, c->nBlks, c->nLpCols, p->lbfgsListLength);
    c->S->hisRecT = p->lbfgsListLength;
    LORADSInitADMMVars(c->S, c->S->var->rankElem, c->BlkDims, c->nBlks, c->nLpCols);
    initial_solver_state(p, c->S, &c->alm, &c->admm, &c->sdpConst);
    g_S = c->S;
    return 0;
}

static void dump_problem_consts(ctx_t *c) {
    lorads_solver *S = c->S;
    /* the LP block (if any) is reported as one more "cone": rank 1, dimension nLpCols, branch flags -1 */
    const int nall = (int)c->nBlks + (c->nLpCols > 0 ? 1 : 0);
    double *rk = malloc(sizeof(double) * nall), *nn = malloc(sizeof(double) * nall),
           *ct = malloc(sizeof(double) * nall), *wt = malloc(sizeof(double) * nall);
    if (c->nLpCols > 0) { rk[nall - 1] = 1; nn[nall - 1] = c->nLpCols; ct[nall - 1] = -1; wt[nall - 1] = -1; }
    for (int k = 0; k < c->nBlks; ++k) {
        rk[k] = S->var->rankElem[k];
        nn[k] = c->BlkDims[k];
        ct[k] = (S->SDPCones[k]->type == LORADS_CONETYPE_SPARSE_SDP) ? 1.0 : 0.0;
        wt[k] = (S->SDPCones[k]->sdp_coeff_w_sum->dataType == SDP_COEFF_DENSE) ? 1.0 : 0.0;
    }
    rec("rank", rk, nall);
    rec("blkdims", nn, nall);
    rec("cone_is_sparse", ct, nall);
    rec("wsum_is_dense", wt, nall);
    rec1("m", c->nConstrs);
    rec1("rho0", c->alm.rho);
    rec1("cObjNrm1", S->cObjNrm1); rec1("cObjNrm2", S->cObjNrm2); rec1("cObjNrmInf", S->cObjNrmInf);
    rec1("bRHSNrm1", S->bRHSNrm1); rec1("bRHSNrm2", S->bRHSNrm2); rec1("bRHSNrmInf", S->bRHSNrmInf);
    rec("b", S->rowRHS, S->nRows);
    free(rk); free(nn); free(ct); free(wt);
}

static lorads_lp_dense *g_lp_of(lorads_solver *S, lorads_sdp_dense **M) { /* the LP vector that travels with M */
    if (S->nLpCols <= 0) return NULL;
    if (M == S->var->R) return S->var->rLp;
    if (M == S->var->U) return S->var->uLp;
    if (M == S->var->V) return S->var->vLp;
    if (M == S->var->Grad) return S->var->gradLp;
    return NULL;
}
static void dump_mats(const char *fmt, int it, lorads_sdp_dense **M, int nb) {
    for (int k = 0; k < nb; ++k) recf(fmt, it, k, M[k]->matElem, (int64_t)M[k]->nRows * M[k]->rank);
    lorads_lp_dense *lp = g_S ? g_lp_of(g_S, M) : NULL;
    if (lp) recf(fmt, it, nb, lp->matElem, (int64_t)lp->nCols); /* pseudo-cone index nb */
}

static void dump_final(ctx_t *c, int dump_state) {
    lorads_solver *S = c->S;
    rec1("pObj", S->pObjVal); rec1("dObj", S->dObjVal);
    rec1("err_constr_l1", S->dimacError[LORADS_DIMAC_ERROR_CONSTRVIO_L1]);
    rec1("err_pdgap", S->dimacError[LORADS_DIMAC_ERROR_PDGAP]);
    rec1("alm_outer", c->alm.outerIter); rec1("alm_inner", c->alm.innerIter); rec1("alm_rho", c->alm.rho);
    rec1("admm_iter", c->admm.iter); rec1("admm_cg_iter", c->admm.cg_iter); rec1("admm_rho", c->admm.rho);
    rec1("admm_pinf_l1", c->admm.l_1_primal_infeasibility); rec1("admm_gap", c->admm.primal_dual_gap);
    rec1("alm_pinf_l1", c->alm.l_1_primal_infeasibility); rec1("alm_gap", c->alm.primal_dual_gap);
    double *rk = malloc(sizeof(double) * c->nBlks);
    for (int k = 0; k < c->nBlks; ++k) rk[k] = S->var->U[k]->rank;
    rec("final_rank", rk, c->nBlks);
    free(rk);
    if (dump_state) {
        dump_mats("U_final_%d_%d", 0, S->var->U, c->nBlks);
        dump_mats("V_final_%d_%d", 0, S->var->V, c->nBlks);
        dump_mats("R_final_%d_%d", 0, S->var->R, c->nBlks);
        rec("lambda_final", S->var->dualVar, S->nRows);
    }
}

/* presolve = what AConeProcData / AConePresolveData / LORADSDetermineRank decided for every cone (data/lorads_sdp_conic.c:868-1076,
 * data/lorads_sdp_data.c:811-828, io/lorads_user_data.c:58): one line per cone, read by oracle/make_golden.py into
 * tests/golden/presolve.json.  counts[] = coefficient matrices by type INCLUDING the objective (sdpConeStats). */
static int mode_presolve(ctx_t *c) {
    lorads_solver *S = c->S;
    for (int k = 0; k < c->nBlks; ++k) {
        lorads_sdp_cone *cone = S->SDPCones[k];
        const int sparse_cone = cone->type == LORADS_CONETYPE_SPARSE_SDP;
        long held, nz, nsp, nds;
        int obj_type;
        if (sparse_cone) {
            lorads_cone_sdp_sparse *d = (lorads_cone_sdp_sparse *)cone->coneData;
            held = (long)d->nRowElem; nz = (long)d->sdpConeStats[SDP_COEFF_ZERO]; nsp = (long)d->sdpConeStats[SDP_COEFF_SPARSE];
            nds = (long)d->sdpConeStats[SDP_COEFF_DENSE]; obj_type = (int)d->sdpObj->dataType;
        } else {
            lorads_cone_sdp_dense *d = (lorads_cone_sdp_dense *)cone->coneData;
            held = (long)d->nRow; nz = (long)d->sdpConeStats[SDP_COEFF_ZERO]; nsp = (long)d->sdpConeStats[SDP_COEFF_SPARSE];
            nds = (long)d->sdpConeStats[SDP_COEFF_DENSE]; obj_type = (int)d->sdpObj->dataType;
        }
        const int wdense = cone->sdp_coeff_w_sum->dataType == SDP_COEFF_DENSE;
        const long n = (long)c->BlkDims[k];
        long wnnz = wdense ? n * (n + 1) / 2 : (long)((sdp_coeff_sparse *)cone->sdp_coeff_w_sum->dataMat)->nTriMatElem;
        long onnz = cone->sdp_obj_sum->dataType == SDP_COEFF_DENSE ? n * (n + 1) / 2
                                                                    : (long)((sdp_coeff_sparse *)cone->sdp_obj_sum->dataMat)->nTriMatElem;
        printf("@@REF_PRESOLVE cone=%d n=%ld rank=%ld cone_sparse=%d rows_held=%ld wsum_dense=%d wsum_nnz=%ld objsum_nnz=%ld n_zero=%ld n_sparse=%ld n_dense=%ld obj_type=%d\n",
               k, n, (long)S->var->rankElem[k], sparse_cone, held, wdense, wnnz, onnz, nz, nsp, nds, obj_type);
    }
    printf("@@REF_PRESOLVE_END m=%ld nblk=%ld nlp=%ld\n", (long)c->nConstrs, (long)c->nBlks, (long)c->nLpCols);
    return 0;
}

/* solve = main.c:321-398 without the ARPACK step and without the level-2 reopt loop (which is
 * conditioned on the ARPACK result, main.c:414-476) */
static int mode_solve(ctx_t *c, lorads_params *p, int dump_state) {
    double t0 = LUtilGetTimeStamp();
    dump_problem_consts(c);
    int admm_bad_iter_flag = 0;
    double reopt_param = 5;
    lorads_int alm_reopt_min_iter = 3, admm_reopt_min_iter = p->highAccMode ? 1000 : 50;
    c->S->AStatus = LORADS_UNKNOWN;
    double ta = LUtilGetTimeStamp();
    lorads_int rc1 = LORADS_ALMOptimize(p, c->S, &c->alm, p->maxALMIter, t0);
    double tb = LUtilGetTimeStamp();
    rec1("alm_ret", rc1);
    rec1("alm_seconds", tb - ta);
    rec1("alm_pObj", c->S->pObjVal); rec1("alm_dObj", c->S->dObjVal);
    rec1("alm_end_outer", c->alm.outerIter); rec1("alm_end_inner", c->alm.innerIter);
    rec1("alm_end_rho", c->alm.rho);
    rec1("alm_end_pinf_l1", c->alm.l_1_primal_infeasibility); rec1("alm_end_gap", c->alm.primal_dual_gap);
    if (dump_state) {
        dump_mats("R_alm_%d_%d", 0, c->S->var->R, c->nBlks);
        rec("lambda_alm", c->S->var->dualVar, c->S->nRows);
    }
    LORADS_ALMtoADMM(c->S, p, &c->alm, &c->admm);
    rec1("admm_rho_start", c->admm.rho);
    tb = LUtilGetTimeStamp();
    lorads_int rc2 = LORADSADMMOptimize(p, c->S, &c->admm, p->maxADMMIter, t0);
    double tc = LUtilGetTimeStamp();
    if (rc2 == RET_CODE_BAD_ITER) admm_bad_iter_flag = 1;
    rec1("admm_ret", rc2);
    rec1("admm_seconds", tc - tb);
    rec1("admm_first_iter", c->admm.iter); rec1("admm_first_cg", c->S->cgIter);
    printf("\n@@REF_TIMING alm_s=%.6f admm_s=%.6f admm_iter=%d cg_iter=%d\n", tb - ta, tc - tb, (int)c->admm.iter,
           (int)c->S->cgIter);
    int cnt = 0;
    if (p->reoptLevel >= 1) {
        while ((c->alm.primal_dual_gap > p->phase2Tol || c->alm.l_1_primal_infeasibility > p->phase2Tol) &&
               (c->admm.primal_dual_gap > p->phase2Tol || c->admm.l_1_primal_infeasibility > p->phase2Tol)) {
            if (cnt >= 1) break;
            printf("******  reopt parameter:%.3f\n", reopt_param);
            reopt(p, c->S, &c->alm, &c->admm, &reopt_param, &alm_reopt_min_iter, &admm_reopt_min_iter, t0,
                  &admm_bad_iter_flag, 1);
            cnt += 1;
        }
    }
    rec1("reopt_rounds", cnt);
    dump_final(c, dump_state);
    printf("\n@@REF_FINAL pObj=%.12e dObj=%.12e constrVio=%.6e pdGap=%.6e\n", c->S->pObjVal, c->S->dObjVal,
           c->S->dimacError[LORADS_DIMAC_ERROR_CONSTRVIO_L1], c->S->dimacError[LORADS_DIMAC_ERROR_PDGAP]);
    return 0;
}

/* trace: function-level goldens through the lorads_func table */
static int mode_trace(ctx_t *c, lorads_params *p, int n_alm, int n_admm, double fixrho) {
    lorads_solver *S = c->S;
    lorads_func *f;
    LORADSInitFuncSet(&f, S->nLpCols);
    int nb = (int)c->nBlks;
    lorads_int m = S->nRows, incx = 1;
    double minusOne = -1.0;
    double rho = fixrho > 0 ? fixrho : c->alm.rho;
    dump_problem_consts(c);
    rec1("trace_rho", rho);
    dump_mats("R_%d_%d", 0, S->var->R, nb);
    dump_mats("Uinit_%d_%d", 0, S->var->U, nb);
    dump_mats("Vinit_%d_%d", 0, S->var->V, nb);

    f->InitConstrValAll(S, S->var->rLp, S->var->rLp, S->var->R, S->var->R);
    f->InitConstrValSum(S);
    rec("csum_init", S->var->constrValSum, m);
    double lag = 0.0;
    f->ALMCalGrad(S, S->var->rLp, S->var->gradLp, S->var->R, S->var->Grad, &lag, rho);
    dump_mats("Grad_%d_%d", 0, S->var->Grad, nb);
    recf("lagsq_%d_%d", 0, 0, &lag, 1);
    f->calObj_alm(S);
    rec1("pObj_init", S->pObjVal);

    for (int it = 0; it < n_alm; ++it) {
        /* one inner iteration, same call order as lorads_alm.c:1075-1146 */
        f->LBFGSDirection(p, S, S->lbfgsHis, S->var->gradLp, S->var->uLp, S->var->Grad, S->var->U, it);
        f->LBFGSDirUseGrad(S, S->var->uLp, S->var->gradLp, S->var->U, S->var->Grad);
        dump_mats("D_%d_%d", it, S->var->U, nb);
        double *q0 = S->var->M1temp;
        memcpy(q0, S->rowRHS, sizeof(double) * m);
        axpy(&m, &minusOne, S->var->constrValSum, &incx, q0, &incx);
        double p12[2];
        f->ALMCalq12p12(S, S->var->rLp, S->var->uLp, S->var->R, S->var->U, S->var->ARDSum, S->var->ADDSum, p12);
        recf("q1_%d_%d", it, 0, S->var->ARDSum, m);
        recf("q2_%d_%d", it, 0, S->var->ADDSum, m);
        recf("p12_%d_%d", it, 0, p12, 2);
        double tau = 0.0;
        lorads_int rootNum = ALMLineSearch(rho, m, S->var->dualVar, p12[0], p12[1], q0, S->var->ARDSum,
                                           S->var->ADDSum, &tau);
        double tr[2] = {tau, (double)rootNum};
        recf("tau_%d_%d", it, 0, tr, 2);
        f->setAsNegGrad(S, S->var->gradLp, S->var->Grad);
        f->ALMupdateVar(S, S->var->rLp, S->var->uLp, S->var->R, S->var->U, tau);
        double tau2 = tau * tau;
        axpy(&m, &tau, S->var->ARDSum, &incx, S->var->constrValSum, &incx);
        axpy(&m, &tau2, S->var->ADDSum, &incx, S->var->constrValSum, &incx);
        recf("csum_inc_%d_%d", it, 0, S->var->constrValSum, m);
        f->ALMCalGrad(S, S->var->rLp, S->var->gradLp, S->var->R, S->var->Grad, &lag, rho);
        f->setlbfgsHisTwo(S, S->var->gradLp, S->var->uLp, S->var->Grad, S->var->U, tau);
        f->updateDimacsALM(S, S->var->R, S->var->R, S->var->rLp, S->var->rLp);
        dump_mats("R_%d_%d", it + 1, S->var->R, nb);
        dump_mats("Grad_%d_%d", it + 1, S->var->Grad, nb);
        recf("lagsq_%d_%d", it + 1, 0, &lag, 1);
        recf("csum_%d_%d", it, 0, S->var->constrValSum, m);
        recf("err1_%d_%d", it, 0, &S->dimacError[LORADS_DIMAC_ERROR_CONSTRVIO_L1], 1);
    }
    LORADSUpdateDualVar(S, rho);
    rec("lambda_alm", S->var->dualVar, m);
    f->calObj_alm(S);
    LORADSCalDualObj(S);
    rec1("pObj_alm", S->pObjVal);
    rec1("dObj_alm", S->dObjVal);

    /* warm start for the ADMM part of the trace: the reference's own phase 1 from the current point
     * (a cold ADMM start diverges and pins nothing); its result is dumped as an INPUT of the part */
    {
        double t0 = LUtilGetTimeStamp();
        /* LORADS_REF_NO_WARM=1 (sizes where the reference's phase 1 takes hours): the ADMM part starts from the state
         * the traced ALM iterations left -- still the same functions on the same inputs */
        if (!getenv("LORADS_REF_NO_WARM")) LORADS_ALMOptimize(p, S, &c->alm, p->maxALMIter, t0);
        {
            double rkw[65];
            for (int k = 0; k < nb && k < 64; ++k) rkw[k] = S->var->R[k]->rank;
            if (S->nLpCols > 0) rkw[nb < 64 ? nb : 64] = 1;
            rec("rank_warm", rkw, nb + (S->nLpCols > 0 ? 1 : 0)); /* phase 1 may have grown the rank (AUG_RANK) */
        }
        dump_mats("R_warm_%d_%d", 0, S->var->R, nb);
        rec("lambda_warm", S->var->dualVar, m);
        LORADS_ALMtoADMM(S, p, &c->alm, &c->admm);
        rho = c->admm.rho < p->rhoMax ? c->admm.rho : p->rhoMax;
        rec1("admm_rho", rho);
    }
    /* ADMM prologue, lorads_admm.c:47-52 */
    S->cgIter = 0;
    f->InitConstrValAll(S, S->var->uLp, S->var->vLp, S->var->U, S->var->V);
    f->InitConstrValSum(S);
    f->calObj_admm(S);
    LORADSCalDualObj(S);
    f->updateDimacsADMM(S, S->var->U, S->var->V, S->var->uLp, S->var->vLp);
    rec("admm_csum_init", S->var->constrValSum, m);
    double e0[3] = {S->pObjVal, S->dObjVal, S->dimacError[LORADS_DIMAC_ERROR_CONSTRVIO_L1]};
    rec("admm_init_scalars", e0, 3);
    double l1 = S->dimacError[LORADS_DIMAC_ERROR_CONSTRVIO_L1];
    for (int it = 0; it < n_admm; ++it) {
        /* one ADMM iteration, same call order as lorads_admm.c:76-81,120 */
        double tol = LORADS_MIN(l1 * 1e-2, 1e-8);
        f->admmUpdateVar(S, rho, tol, 800);
        dump_mats("U_%d_%d", it, S->var->U, nb);
        dump_mats("V_%d_%d", it, S->var->V, nb);
        recf("csum_uv_%d_%d", it, 0, S->var->constrValSum, m);
        f->calObj_admm(S);
        LORADSCalDualObj(S);
        f->updateDimacsADMM(S, S->var->U, S->var->V, S->var->uLp, S->var->vLp);
        double sc[6] = {S->pObjVal, S->dObjVal, S->dimacError[LORADS_DIMAC_ERROR_CONSTRVIO_L1],
                        S->dimacError[LORADS_DIMAC_ERROR_PDGAP], (double)S->cgIter, tol};
        recf("admm_scalars_%d_%d", it, 0, sc, 6);
        recf("csum_rr_%d_%d", it, 0, S->var->constrValSum, m);
        LORADSUpdateDualVar(S, rho);
        recf("lambda_%d_%d", it, 0, S->var->dualVar, m);
        l1 = S->dimacError[LORADS_DIMAC_ERROR_CONSTRVIO_L1];
    }
    return 0;
}

/* bench: time K ADMM iterations from given U,V (file of doubles: for each cone U then V, column-major)
 * -- the CPU baseline of bench.py, kind "reference" */
static int mode_admm_bench(ctx_t *c, lorads_params *p, int n_admm, const char *uvfile, double fixrho) {
    lorads_solver *S = c->S;
    lorads_func *f;
    LORADSInitFuncSet(&f, S->nLpCols);
    int nb = (int)c->nBlks;
    /* LORADS_REF_UV_RANKS="r0,r1,...": the factors in the file have these ranks (the run that wrote them grew its ranks in
     * phase 1).  The reference grows ranks by AUG_RANK steps of factor 1.5 (lorads_alm.c:1007,1232): apply that step until the
     * solver's buffers have the file's shape (refused if the file's ranks are not reachable that way). */
    const char *want = getenv("LORADS_REF_UV_RANKS");
    if (uvfile && want && *want) {
        for (int guard = 0; guard < 16; ++guard) {
            int ok = 1, over = 0;
            const char *q = want;
            for (int k = 0; k < nb; ++k) {
                long t = strtol(q, (char **)&q, 10);
                if (*q == ',') ++q;
                if (S->var->U[k]->rank < t) ok = 0;
                if (S->var->U[k]->rank > t) over = 1;
            }
            if (over) { fprintf(stderr, "LORADS_REF_UV_RANKS: not reachable by AUG_RANK steps\n"); return 1; }
            if (ok) break;
            AUG_RANK(S, S->var->rankElem, c->nBlks, 1.5);
        }
    }
    if (uvfile) {
        FILE *fp = fopen(uvfile, "rb");
        if (!fp) { fprintf(stderr, "cannot open %s\n", uvfile); return 1; }
        for (int k = 0; k < nb; ++k) {
            size_t cnt = (size_t)S->var->U[k]->nRows * S->var->U[k]->rank;
            if (fread(S->var->U[k]->matElem, 8, cnt, fp) != cnt) return 1;
            if (fread(S->var->V[k]->matElem, 8, cnt, fp) != cnt) return 1;
        }
        /* optional: dual vector after the factors */
        if (fread(S->var->dualVar, 8, (size_t)S->nRows, fp) != (size_t)S->nRows)
            memset(S->var->dualVar, 0, sizeof(double) * (size_t)S->nRows);
        fclose(fp);
    }
    double rho = fixrho > 0 ? fixrho : c->admm.rho;
    S->cgIter = 0;
    f->InitConstrValAll(S, S->var->uLp, S->var->vLp, S->var->U, S->var->V);
    f->InitConstrValSum(S);
    f->calObj_admm(S);
    LORADSCalDualObj(S);
    f->updateDimacsADMM(S, S->var->U, S->var->V, S->var->uLp, S->var->vLp);
    double l1 = S->dimacError[LORADS_DIMAC_ERROR_CONSTRVIO_L1];
    double t0 = LUtilGetTimeStamp();
    for (int it = 0; it < n_admm; ++it) {
        double tol = LORADS_MIN(l1 * 1e-2, 1e-8);
        f->admmUpdateVar(S, rho, tol, 800);
        f->calObj_admm(S);
        LORADSCalDualObj(S);
        f->updateDimacsADMM(S, S->var->U, S->var->V, S->var->uLp, S->var->vLp);
        LORADSUpdateDualVar(S, rho);
        l1 = S->dimacError[LORADS_DIMAC_ERROR_CONSTRVIO_L1];
    }
    double t1 = LUtilGetTimeStamp();
    printf("@@REF_ADMM_BENCH iters=%d seconds=%.6f cg_iters=%d pObj=%.12e dObj=%.12e err1=%.6e\n", n_admm, t1 - t0,
           (int)S->cgIter, S->pObjVal, S->dObjVal, l1);
    rec1("bench_seconds", t1 - t0);
    rec1("bench_cg_iters", S->cgIter);
    rec1("bench_pObj", S->pObjVal);
    rec1("bench_dObj", S->dObjVal);
    rec1("bench_err1", l1);
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: %s <file.dat-s> <solve|trace|admmbench> <dump.bin|-> [--opt val ...]\n", argv[0]);
        return 2;
    }
    lorads_params p;
    default_params(&p);
    p.fname = argv[1];
    int n_alm = 5, n_admm = 3, dump_state = 0;
    char *uvfile = NULL;
    double fixrho = -1.0;
    if (parse_opts(argc, argv, 4, &p, &n_alm, &n_admm, &dump_state, &uvfile, &fixrho)) return 2;
    if (strcmp(argv[3], "-") != 0) {
        g_dump = fopen(argv[3], "wb");
        if (!g_dump) { fprintf(stderr, "cannot open dump\n"); return 2; }
    }
    ctx_t c;
    if (setup(&c, &p)) return 1;
    int rc = 0;
    if (!strcmp(argv[2], "solve")) rc = mode_solve(&c, &p, dump_state);
    else if (!strcmp(argv[2], "trace")) rc = mode_trace(&c, &p, n_alm, n_admm, fixrho);
    else if (!strcmp(argv[2], "admmbench")) rc = mode_admm_bench(&c, &p, n_admm, uvfile, fixrho);
    else if (!strcmp(argv[2], "presolve")) rc = mode_presolve(&c);
    else { fprintf(stderr, "unknown mode\n"); rc = 2; }
    if (g_dump) fclose(g_dump);
    fflush(stdout);
    return rc;
}
