/* lorads_rccl.h -- native all-reduce hook for the sharded ADMM path (liblorads_rccl.so, lorads_amd/csrc/hip/rccl_hook.cpp).
 *
 * The reference is single-process; its multi-GPU form (one cone per GPU, ONE all-reduce of the shared
 * m-vector per ADMM iteration) exists only here.  lorads_hip.h takes that sum through the callback type
 * lorads_hip_allreduce_fn; this library IS such a callback, implemented with one ncclAllReduce enqueued in stream order
 * on the HIP library's own stream (RCCL over xGMI, one process per GPU), so that no interpreter or framework code runs
 * between two kernel launches of an ADMM iteration.  RCCL is bound with dlopen/dlsym from the path the caller names.
 *
 * Usage (every rank): lorads_rccl_open(path); rank 0: lorads_rccl_unique_id(id) and ship the 128 bytes to the others;
 * h = lorads_rccl_comm_create(id, rank, world, lorads_hip_stream(ctx));
 * lorads_hip_set_allreduce(ctx, lorads_rccl_allreduce_hook, h); lorads_hip_set_allreduce_stream_ordered(ctx, 1).
 * All functions return 0 on success (comm_create: non-null); lorads_rccl_last_error() says what went wrong. */
#ifndef LORADS_RCCL_H
#define LORADS_RCCL_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
int lorads_rccl_open(const char *librccl_path);
int lorads_rccl_unique_id(char out[128]);
void *lorads_rccl_comm_create(const char id[128], int rank, int world, void *hip_stream);
int lorads_rccl_allreduce_hook(void *user, double *buf, int32_t count, int32_t on_device);
void lorads_rccl_comm_destroy(void *comm);
const char *lorads_rccl_last_error(void);
#ifdef __cplusplus
}
#endif
#endif
