/* TEST INFRASTRUCTURE, not product code: a same-host stand-in for the xGMI wire.
 *
 * tests/test_dp_multiprocess.py runs the product's data-parallel step (Trainer + OperandExchange, the pack /
 * rank-segmented Gram / strip kernels, Adam) under N real processes that share ONE MI355X.  RCCL refuses several
 * ranks on one device ("Duplicate GPU detected"), so the collectives of trainer.GradSync are carried here instead:
 * every rank stages its buffer in pinned host memory, and this file moves the bytes between the processes through a
 * POSIX shared-memory segment.  wire_exec has the hipHostFn_t signature: the Python side enqueues it with
 * hipLaunchHostFunc, so a collective is a node in stream order (and a host node of a captured hipGraph) exactly where
 * the RCCL kernel would be.
 *
 * SUM all-reduce adds the ranks' contributions in rank order on every rank: the result is bit-identical everywhere
 * (RCCL guarantees the same).  No HIP, no Python: plain C11 + POSIX, built with gcc by tests/wire/__init__.py.
 */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

typedef struct {
    atomic_int arrived;
    atomic_int sense;
    atomic_int failed;
    int world;
    uint64_t slot_floats;
    atomic_ullong collectives;      /* completed collectives (rank 0 counts) */
} wire_header;

#define HEADER_BYTES 4096

typedef struct {
    int32_t kind;       /* 0: SUM all-reduce in place on `send`; 1: all-gather send[n] -> recv[world * n] (rank major) */
    int32_t reserved;
    float* send;
    float* recv;
    uint64_t n;         /* floats */
} wire_op;

static wire_header* H = NULL;
static float* SLOTS = NULL;
static int RANK = -1, WORLD = 0;
static uint64_t SLOT = 0;
static int LOCAL_SENSE = 0;
static double TIMEOUT_S = 120.0;
static pthread_mutex_t LOCK = PTHREAD_MUTEX_INITIALIZER;
static size_t MAPPED = 0;

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void die(const char* what) {
    fprintf(stderr, "[shm_wire rank %d] %s\n", RANK, what);
    fflush(stderr);
    if (H) atomic_store(&H->failed, 1);
    _exit(86);
}

static void barrier(void) {
    LOCAL_SENSE ^= 1;
    if (atomic_fetch_add(&H->arrived, 1) == WORLD - 1) {
        atomic_store(&H->arrived, 0);
        atomic_store(&H->sense, LOCAL_SENSE);
        return;
    }
    double t0 = now_s();
    unsigned spins = 0;
    while (atomic_load(&H->sense) != LOCAL_SENSE) {
        if (atomic_load(&H->failed)) die("a peer failed");
        if ((++spins & 1023u) == 0) {
            if (now_s() - t0 > TIMEOUT_S) die("barrier timed out (a peer never issued the matching collective)");
            sched_yield();
        }
    }
}

int wire_init(const char* path, int rank, int world, uint64_t slot_floats, double timeout_s) {
    if (H) return -1;
    int fd = open(path, O_RDWR);
    if (fd < 0) return -2;
    size_t bytes = HEADER_BYTES + (size_t)world * slot_floats * sizeof(float);
    struct stat st;
    if (fstat(fd, &st) != 0 || (size_t)st.st_size < bytes) { close(fd); return -3; }
    void* p = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return -4;
    H = (wire_header*)p;
    SLOTS = (float*)((char*)p + HEADER_BYTES);
    RANK = rank; WORLD = world; SLOT = slot_floats; MAPPED = bytes;
    if (timeout_s > 0) TIMEOUT_S = timeout_s;
    if (rank == 0) { H->world = world; H->slot_floats = slot_floats; }
    LOCAL_SENSE = 0;
    return 0;
}

int wire_close(void) {
    if (!H) return 0;
    munmap((void*)H, MAPPED);
    H = NULL; SLOTS = NULL;
    return 0;
}

unsigned long long wire_collectives(void) { return H ? (unsigned long long)atomic_load(&H->collectives) : 0ull; }

/* hipHostFn_t: runs on a HIP runtime thread in stream order */
void wire_exec(void* user) {
    wire_op* op = (wire_op*)user;
    if (!H) die("wire_exec before wire_init");
    pthread_mutex_lock(&LOCK);
    float* mine = SLOTS + (size_t)RANK * SLOT;
    for (uint64_t off = 0; off < op->n; off += SLOT) {
        uint64_t m = op->n - off < SLOT ? op->n - off : SLOT;
        memcpy(mine, op->send + off, m * sizeof(float));
        barrier();
        if (op->kind == 0) {
            float* out = op->send + off;
            const float* s0 = SLOTS;
            for (uint64_t i = 0; i < m; ++i) out[i] = s0[i];
            for (int r = 1; r < WORLD; ++r) {
                const float* s = SLOTS + (size_t)r * SLOT;
                for (uint64_t i = 0; i < m; ++i) out[i] += s[i];
            }
        } else {
            for (int r = 0; r < WORLD; ++r)
                memcpy(op->recv + (size_t)r * op->n + off, SLOTS + (size_t)r * SLOT, m * sizeof(float));
        }
        barrier();                       /* the slots are free again */
    }
    if (RANK == 0) atomic_fetch_add(&H->collectives, 1);
    pthread_mutex_unlock(&LOCK);
}

void* wire_exec_ptr(void) { return (void*)&wire_exec; }
int wire_op_bytes(void) { return (int)sizeof(wire_op); }
