/* main.c -- command line with the reference's option names (src_semi/main.c:57-80):
 *   lorads file.dat-s [--phase1Tol x] [--timesLogRank x] ... ; solves on the MI355X backend. */
#include <libgen.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "lorads_host.h"

typedef struct lrd_session lrd_session;
lrd_session *lrd_session_open(const char *fname);
int lrd_session_set_param(lrd_session *s, const char *key, const char *val);
int lrd_session_prepare(lrd_session *s, int world, int rank_id);
int lrd_session_attach(lrd_session *s, const lrd_backend *be);
int lrd_session_solve(lrd_session *s);
int lrd_session_results(lrd_session *s, double out[16]);
int lrd_session_results2(lrd_session *s, double out[4]);
lrd_problem *lrd_session_problem(lrd_session *s);
lrd_params *lrd_session_params(lrd_session *s);
void lrd_session_close(lrd_session *s);
int lrd_hip_backend_create(const lrd_problem *p, int lbfgs_len, const char *libpath, lrd_backend *out);

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s file.dat-s [--option value ...]   (options: the reference's long options)\n", argv[0]);
        return 2;
    }
    lrd_session *s = lrd_session_open(argv[1]);
    if (!s) return 1;
    for (int i = 2; i < argc; i += 2) {
        if (i + 1 >= argc) {
            fprintf(stderr, "option %s lacks a value\n", argv[i]);
            return 2;
        }
        if (strncmp(argv[i], "--", 2) || lrd_session_set_param(s, argv[i] + 2, argv[i + 1])) {
            fprintf(stderr, "unknown option %s\n", argv[i]);
            return 2;
        }
    }
    lrd_session_prepare(s, 1, 0);
    char self[4096], lib[4200];
    ssize_t n = readlink("/proc/self/exe", self, sizeof self - 1);
    if (n <= 0) return 1;
    self[n] = 0;
    snprintf(lib, sizeof lib, "%s/liblorads_hip.so", dirname(self));
    lrd_backend be;
    if (lrd_hip_backend_create(lrd_session_problem(s), lrd_session_params(s)->lbfgsListLength, lib, &be)) {
        fprintf(stderr, "lorads: the HIP backend is required (no CPU fallback)\n");
        return 1;
    }
    if (lrd_session_attach(s, &be)) return 1;
    const int rc = lrd_session_solve(s);
    if (rc != 0) {
        fprintf(stderr, "lorads: the solve failed (code %d): a backend call reported an error\n", rc);
        lrd_session_close(s);
        return 3;
    }
    double r[16], r2[4];
    lrd_session_results(s, r);
    lrd_session_results2(s, r2);
    static const char *why[] = {"but the status is unknown", "due to reaching `Official terminate criteria`",
                                "due to reaching `final terminate criteria`", "due to reaching `the maximum number of iterations`",
                                "since time limit"};
    printf("End Program %s:\n", why[(int)r[12] >= 0 && (int)r[12] <= 4 ? (int)r[12] : 0]);
    /* printRes layout (data/lorads_solver.c:908-922) */
    printf("-----------------------------------------------------------------------\n");
    printf("Objective function Value are:\n\t 1.Primal Objective:            : %10.6e\n\t 2.Dual Objective:              : %10.6e\n",
           r[0], r[1]);
    printf("Dimacs Error are:\n\t 1.Constraint Violation(1)      : %10.6e\n\t 2.Dual Infeasibility(1)        : %10.6e\n"
           "\t 3.Primal Dual Gap              : %10.6e\n\t 5.Constraint Violation(Inf)    : %10.6e\n"
           "\t 6.Dual Infeasibility(Inf)      : %10.6e\n", r[2], r2[0], r[3], r[15], r2[1]);
    printf("-----------------------------------------------------------------------\n");
    printf("phase 1: %f s, phase 2: %f s (%d ADMM iterations, %d CG iterations), dual infeasibility: %f s\n", r[10], r[11],
           (int)r[13], (int)r[14], r2[2]);
    lrd_session_close(s);
    return 0;
}
