/* shmx.c -- sums of a few doubles over the ranks of ONE node through a page of POSIX shared memory (host side, plain C).
 *
 * What it is for: separable shards (cones dealt over the GPUs of a node; the reference sweeps them in one process,
 * lorads_alg/lorads_alg_common.c:190-214) share four scalars per ADMM iteration -- ||b - A(X)||^2, b.lambda, <C, X> and "my sweep is
 * unfinished".  Every rank's host already waits for its GPU's result hand-over at that point and is the only reader of the sums:
 * the ranks' hosts exchange the numbers themselves, and no collective kernel sits on the stream between two iterations
 * (include/lorads_hip.h: lorads_hip_set_scalar_exchange).
 *
 * Protocol: one slot per rank, two buffers per slot (call number parity), each its own cache lines.  Call s of rank r: write the
 * values into slot[r].buf[s & 1], store-release its sequence word = s, then for q = 0 .. world-1 load-acquire slot[q].buf[s & 1].seq
 * until it equals s and add that rank's values -- in rank order, so every rank forms the SAME sum bit for bit.  A rank can only be
 * one call ahead of the slowest reader of a buffer: to enter call s + 2 it has left call s + 1, for which every rank had written
 * s + 1, i.e. had left call s.  A rank that never arrives (a dead process) ends the wait after LORADS_HANDOVER_TIMEOUT_S seconds.
 *
 * Opening: rank 0 unlinks whatever carries the name, makes the segment and writes its pid next to the magic word.  The other ranks
 * attach only to a segment that (a) still carries its name after they have read the magic word -- one that rank 0 has replaced in the
 * meantime has no link left -- and (b) whose maker is alive: the leftover of a run that died (a fixed rendezvous port gives every run of
 * a user the same name) is refused and the wait for rank 0's fresh one goes on.  Callers that can should still put a barrier between
 * rank 0's open and the others' (bench.py does): it turns the wait into a formality. */
#include "lorads_host.h"

#include <errno.h>
#include <fcntl.h>
#include <sched.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#define SHMX_MAXN 16
#define SHMX_MAGIC 0x4c52445348584d31ull /* "LRDSHXM1" */

typedef struct {
    volatile uint64_t seq;
    double v[SHMX_MAXN];
    char pad[256 - 8 - 8 * SHMX_MAXN];
} shmx_buf; /* 256 bytes */
typedef struct {
    shmx_buf buf[2];
} shmx_slot;
typedef struct {
    volatile uint64_t magic;
    int32_t world;
    int32_t owner_pid;        /* the rank-0 process that made this segment: a segment whose maker is gone is a leftover, not an invitation */
    char pad[256 - 16];
} shmx_head;

struct lrd_shmx {
    ino_t ino;                /* of the mapped segment */
    char name[128];
    int world, rank, owner;
    uint64_t call;
    size_t bytes;
    shmx_head *head;
    shmx_slot *slot;
    double timeout_s;
};

static double now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

int lrd_shmx_open(const char *name, int world, int rank, lrd_shmx **out) {
    *out = NULL;
    if (!name || name[0] != '/' || strlen(name) >= sizeof(((lrd_shmx *)0)->name) || world < 1 || rank < 0 || rank >= world) return 1;
    lrd_shmx *x = (lrd_shmx *)calloc(1, sizeof *x);
    if (!x) return 1;
    strcpy(x->name, name);
    x->world = world; x->rank = rank; x->owner = rank == 0;
    x->bytes = sizeof(shmx_head) + sizeof(shmx_slot) * (size_t)world;
    x->timeout_s = getenv("LORADS_HANDOVER_TIMEOUT_S") ? atof(getenv("LORADS_HANDOVER_TIMEOUT_S")) : 300.0;
    int fd = -1;
    if (x->owner) { /* rank 0 makes the segment (a leftover of the same name goes first) and opens it for the others by setting the magic word */
        shm_unlink(name);
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)x->bytes) != 0) { if (fd >= 0) close(fd); free(x); return 1; }
    } else {
        const double t0 = now_s();
        for (;;) {
            fd = shm_open(name, O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= x->bytes) break;
            if (fd >= 0) close(fd);
            fd = -1;
            if (now_s() - t0 > 60.0) { free(x); return 1; }
            usleep(1000);
        }
    }
    void *p = mmap(NULL, x->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    { struct stat st0; if (fstat(fd, &st0) == 0) x->ino = st0.st_ino; }
    close(fd);
    if (p == MAP_FAILED) { if (x->owner) shm_unlink(name); free(x); return 1; }
    x->head = (shmx_head *)p;
    x->slot = (shmx_slot *)((char *)p + sizeof(shmx_head));
    if (x->owner) {
        memset(p, 0, x->bytes);
        x->head->world = world;
        x->head->owner_pid = (int32_t)getpid();
        __atomic_store_n(&x->head->magic, SHMX_MAGIC, __ATOMIC_RELEASE);
    } else {
        const double t0 = now_s();
        for (;;) {
            int fresh = 0;
            if (__atomic_load_n(&x->head->magic, __ATOMIC_ACQUIRE) == SHMX_MAGIC) {
                /* is what we have mapped still THE segment of that name, and is its maker alive? */
                struct stat st_name;
                int fd2 = shm_open(name, O_RDWR, 0600);
                const int named = fd2 >= 0 && fstat(fd2, &st_name) == 0 && st_name.st_ino == x->ino;
                if (fd2 >= 0) close(fd2);
                const pid_t op = (pid_t)x->head->owner_pid;
                const int alive = op > 0 && (kill(op, 0) == 0 || errno == EPERM);
                fresh = named && alive;
                if (fresh && x->head->world != world) { munmap(p, x->bytes); free(x); return 1; }
                if (fresh) break;
            }
            if (now_s() - t0 > 60.0) { munmap(p, x->bytes); free(x); return 1; }
            usleep(500);
            /* a leftover (dead maker) or a segment rank 0 has replaced: map whatever carries the name now */
            int fd3 = shm_open(name, O_RDWR, 0600);
            struct stat st3;
            if (fd3 >= 0 && fstat(fd3, &st3) == 0 && (size_t)st3.st_size >= x->bytes && st3.st_ino != x->ino) {
                void *p3 = mmap(NULL, x->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd3, 0);
                if (p3 != MAP_FAILED) {
                    munmap(p, x->bytes);
                    p = p3;
                    x->ino = st3.st_ino;
                    x->head = (shmx_head *)p;
                    x->slot = (shmx_slot *)((char *)p + sizeof(shmx_head));
                }
            }
            if (fd3 >= 0) close(fd3);
        }
    }
    *out = x;
    return 0;
}

/* v[0..n) <- sum over the ranks, in rank order; 0, or 1 after the time limit (some rank never arrived) or on a bad argument */
int lrd_shmx_allreduce(lrd_shmx *x, double *v, int n) {
    if (!x || n < 0 || n > SHMX_MAXN) return 1;
    const uint64_t s = ++x->call;
    shmx_buf *mine = &x->slot[x->rank].buf[s & 1];
    for (int i = 0; i < n; ++i) mine->v[i] = v[i];
    __atomic_store_n(&mine->seq, s, __ATOMIC_RELEASE);
    double acc[SHMX_MAXN];
    for (int i = 0; i < n; ++i) acc[i] = 0.0;
    double t0 = 0.0;
    for (int q = 0; q < x->world; ++q) {
        const shmx_buf *b = &x->slot[q].buf[s & 1];
        for (unsigned long spins = 0; __atomic_load_n(&b->seq, __ATOMIC_ACQUIRE) != s; ++spins) {
            if ((spins & 0xffff) == 0xffff) { /* now and then: give the core away for a moment, and look at the clock */
                sched_yield();
                const double t = now_s();
                if (t0 == 0.0) t0 = t;
                else if (x->timeout_s > 0 && t - t0 > x->timeout_s) return 1;
            }
        }
        for (int i = 0; i < n; ++i) acc[i] += b->v[i];
    }
    for (int i = 0; i < n; ++i) v[i] = acc[i];
    return 0;
}

/* the same as a hook of type lorads_hip_scalar_exchange_fn (user = the lrd_shmx) */
int lrd_shmx_hook(void *user, double *vals, int32_t n) { return lrd_shmx_allreduce((lrd_shmx *)user, vals, (int)n); }

void lrd_shmx_close(lrd_shmx *x) {
    if (!x) return;
    if (x->head) munmap((void *)x->head, x->bytes);
    if (x->owner) shm_unlink(x->name);
    free(x);
}
