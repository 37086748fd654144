// Host cost of enqueuing a chain of N small kernels per "iteration" on MI355X / ROCm 7: (a) launch by launch, (b) one captured graph
// replayed, (c) re-captured every iteration and pushed into the instantiated graph with hipGraphExecUpdate (what a caller whose kernel
// ARGUMENTS change from iteration to iteration would have to do).  Prints microseconds per iteration (host wall clock, stream
// synchronised once per iteration, as the ADMM loop's hand-over does).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
struct Args { double rho, tol; const double *a; double *b; int n, k; };
__global__ void k_small(Args A) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < A.n) A.b[i] = A.a[i] * A.rho + A.tol * A.k;
}
static void enqueue(hipStream_t st, int N, double *x, double *y, int n, double rho) {
    for (int k = 0; k < N; ++k) {
        Args A{rho, 1e-8, (k & 1) ? y : x, (k & 1) ? x : y, n, k};
        hipLaunchKernelGGL(k_small, dim3((n + 255) / 256), dim3(256), 0, st, A);
    }
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 16, iters = argc > 2 ? atoi(argv[2]) : 2000, n = argc > 3 ? atoi(argv[3]) : 100000;
    double *x, *y;
    HC(hipMalloc(&x, sizeof(double) * n)); HC(hipMalloc(&y, sizeof(double) * n));
    HC(hipMemset(x, 0, sizeof(double) * n)); HC(hipMemset(y, 0, sizeof(double) * n));
    hipStream_t st;
    HC(hipStreamCreate(&st));
    for (int w = 0; w < 50; ++w) enqueue(st, N, x, y, n, 1.0);
    HC(hipStreamSynchronize(st));
    double t0 = now();
    for (int it = 0; it < iters; ++it) { enqueue(st, N, x, y, n, 1.0 + it); HC(hipStreamSynchronize(st)); }
    const double ta = (now() - t0) / iters * 1e6;
    // (b) captured once, replayed
    hipGraph_t g; hipGraphExec_t ex;
    HC(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    enqueue(st, N, x, y, n, 1.0);
    HC(hipStreamEndCapture(st, &g));
    double ti0 = now();
    HC(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    const double tinst = (now() - ti0) * 1e6;
    for (int w = 0; w < 50; ++w) HC(hipGraphLaunch(ex, st));
    HC(hipStreamSynchronize(st));
    t0 = now();
    for (int it = 0; it < iters; ++it) { HC(hipGraphLaunch(ex, st)); HC(hipStreamSynchronize(st)); }
    const double tb = (now() - t0) / iters * 1e6;
    // (c) re-captured every iteration (arguments change), pushed into the executable graph
    int upd_fail = 0;
    t0 = now();
    for (int it = 0; it < iters; ++it) {
        hipGraph_t g2;
        HC(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        enqueue(st, N, x, y, n, 2.0 + it);
        HC(hipStreamEndCapture(st, &g2));
        hipGraphExecUpdateResult res;
        hipGraphNode_t errn;
        if (hipGraphExecUpdate(ex, g2, &errn, &res) != hipSuccess) { ++upd_fail; HC(hipGraphExecDestroy(ex)); HC(hipGraphInstantiate(&ex, g2, nullptr, nullptr, 0)); }
        HC(hipGraphDestroy(g2));
        HC(hipGraphLaunch(ex, st));
        HC(hipStreamSynchronize(st));
    }
    const double tc = (now() - t0) / iters * 1e6;
    // GPU-side time of the chain alone (events around a replay)
    hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    HC(hipEventRecord(e0, st)); for (int it = 0; it < 100; ++it) HC(hipGraphLaunch(ex, st)); HC(hipEventRecord(e1, st)); HC(hipEventSynchronize(e1));
    float ms; HC(hipEventElapsedTime(&ms, e0, e1));
    printf("N = %d kernels of %d doubles per iteration: launch by launch %.1f us | graph replay %.1f us (instantiate once: %.0f us) | "
           "re-capture + ExecUpdate + replay %.1f us (update failures %d) | GPU time of the chain, 100 replays back to back: %.1f us\n",
           N, n, ta, tb, tinst, tc, upd_fail, ms * 10.0);
    return 0;
}
