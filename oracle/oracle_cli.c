/* oracle_cli.c -- TEST INFRASTRUCTURE: runs the host control flow (lorads_amd/csrc/host) over the CPU
 * oracle backend on a .dat-s file; used to pin the restatement against oracle/_ref and as the timed
 * CPU baseline of bench.py (kind "port").  usage: oracle_cli file.dat-s [--key val ...] */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "lorads_host.h"

typedef struct lrd_session lrd_session;
lrd_session *lrd_session_open(const char *fname);
int lrd_session_set_param(lrd_session *s, const char *key, const char *val);
int lrd_session_prepare(lrd_session *s, int world, int rank_id);
int lrd_session_attach(lrd_session *s, const lrd_backend *be);
int lrd_session_solve(lrd_session *s);
int lrd_session_results(lrd_session *s, double out[16]);
lrd_problem *lrd_session_problem(lrd_session *s);
lrd_params *lrd_session_params(lrd_session *s);
void lrd_session_close(lrd_session *s);
int lorads_oracle_backend_create(const lrd_problem *p, int lbfgs_len, lrd_backend *out);

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s file.dat-s [--key val ...]\n", argv[0]); return 2; }
    lrd_session *s = lrd_session_open(argv[1]);
    if (!s) return 1;
    for (int i = 2; i + 1 < argc; i += 2) {
        if (strncmp(argv[i], "--", 2) || lrd_session_set_param(s, argv[i] + 2, argv[i + 1])) {
            fprintf(stderr, "bad option %s\n", argv[i]);
            return 2;
        }
    }
    lrd_session_prepare(s, 1, 0);
    lrd_backend be;
    lorads_oracle_backend_create(lrd_session_problem(s), lrd_session_params(s)->lbfgsListLength, &be);
    lrd_session_attach(s, &be);
    lrd_session_solve(s);
    double r[16];
    lrd_session_results(s, r);
    printf("\n@@ORACLE_TIMING alm_s=%.6f admm_s=%.6f admm_iter=%d cg_iter=%d\n", r[10], r[11], (int)r[13], (int)r[14]);
    printf("\n@@ORACLE_FINAL pObj=%.12e dObj=%.12e constrVio=%.6e pdGap=%.6e\n", r[0], r[1], r[2], r[3]);
    lrd_session_close(s);
    return 0;
}
