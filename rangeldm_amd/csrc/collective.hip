// RCCL behind the C ABI: the two exchange steps of the path -- the all-gather of the finished range
// images of a sample-sharded sampling run (the reference writes files per rank instead, ldm/inference.py:159-183) and the
// all-reduce of the flat gradient buffer of a data-parallel training step (accelerate's DDP, ldm/train_unconditional.py:
// 402-404,545) -- issued on the caller's HIP stream, so they order behind the sampler's / trainer's kernels without a host
// synchronisation and a non-Python host can run N > 1.
//
// RCCL is bound at RUN time (dlopen / dlsym), not at link time: a PyTorch process already holds its own copy of librccl
// (torch/lib/librccl.so) and a second, different copy in the same process is asking for trouble, so the library first looks
// for an RCCL that is already loaded (RTLD_NOLOAD), then for RLDM_RCCL_LIB, then for the system one (/opt/rocm/lib).
// One process per GPU; xGMI is point-to-point, so both collectives move few, large messages (one per batch / per bucket).
#include "kernels.h"

#include <dlfcn.h>

#include <cstring>
#include <mutex>

namespace rldm {
namespace {

// the slice of rccl.h this file needs (ABI of RCCL 2.x: /opt/rocm/include/rccl/rccl.h:40-43,187,220,260,339,448-466)
constexpr int kUniqueIdBytes = 128;
struct UniqueId { char internal[kUniqueIdBytes]; };
typedef void* Comm;
constexpr int kSum = 0, kAvg = 4, kFloat32 = 7;

struct Rccl {
    void* handle = nullptr;
    std::string origin;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mutex;

int bind_rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return 0;
    void* h = nullptr;
    std::string origin;
    const char* sonames[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : sonames) {                        // 1. whatever RCCL this process already has (PyTorch's)
        h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (h) { origin = std::string(n) + " (already loaded)"; break; }
    }
    if (!h) {
        const char* env = getenv("RLDM_RCCL_LIB");         // 2. an explicit path
        if (env && *env) {
            h = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
            if (h) origin = env;
        }
    }
    const char* paths[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (int i = 0; !h && i < 4; ++i) {                    // 3. the system copy
        h = dlopen(paths[i], RTLD_NOW | RTLD_GLOBAL);
        if (h) origin = paths[i];
    }
    if (!h) {                                              // (dlerror() returns the message ONCE and clears it)
        const char* e = dlerror();
        const std::string why = e ? e : "";
        RLDM_REQUIRE(false, "RCCL not found (librccl.so; set RLDM_RCCL_LIB): " + why);
    }
    Rccl r;
    r.handle = h;
    r.origin = origin;
    *reinterpret_cast<void**>(&r.GetUniqueId) = dlsym(h, "ncclGetUniqueId");
    *reinterpret_cast<void**>(&r.CommInitRank) = dlsym(h, "ncclCommInitRank");
    *reinterpret_cast<void**>(&r.CommDestroy) = dlsym(h, "ncclCommDestroy");
    *reinterpret_cast<void**>(&r.AllGather) = dlsym(h, "ncclAllGather");
    *reinterpret_cast<void**>(&r.AllReduce) = dlsym(h, "ncclAllReduce");
    *reinterpret_cast<void**>(&r.GetErrorString) = dlsym(h, "ncclGetErrorString");
    RLDM_REQUIRE(r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.AllReduce && r.GetErrorString,
                 "RCCL at " + origin + " lacks a required symbol");
    g_rccl = r;
    return 0;
}

#define RLDM_RCCL_CHECK(expr)                                                                       \
    do {                                                                                            \
        const int _r = (expr);                                                                      \
        if (_r != 0) {                                                                              \
            rldm::set_error(std::string(#expr) + ": " + g_rccl.GetErrorString(_r) + " (" + __FILE__ + \
                            ":" + std::to_string(__LINE__) + ")");                                  \
            return 1;                                                                               \
        }                                                                                           \
    } while (0)

}  // namespace
}  // namespace rldm

struct rldm_comm {
    rldm::Comm comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

using namespace rldm;

extern "C" {

int rldm_comm_bind(void) { return bind_rccl(); }

int rldm_comm_unique_id(void* id_out, size_t cap) {
    RLDM_REQUIRE(id_out && cap >= (size_t)kUniqueIdBytes, "unique id buffer must hold RLDM_UNIQUE_ID_BYTES (128) bytes");
    if (bind_rccl()) return 1;
    UniqueId id;
    RLDM_RCCL_CHECK(g_rccl.GetUniqueId(&id));
    memcpy(id_out, id.internal, kUniqueIdBytes);
    return 0;
}

int rldm_comm_create(const void* unique_id, int rank, int world, rldm_comm** out) {
    RLDM_REQUIRE(unique_id && out && world >= 1 && rank >= 0 && rank < world, "bad communicator arguments");
    if (bind_rccl()) return 1;
    auto* c = new rldm_comm();
    c->rank = rank;
    c->world = world;
    if (hipGetDevice(&c->device) != hipSuccess) {
        delete c;
        set_error("rldm_comm_create: no current HIP device");
        return 1;
    }
    UniqueId id;
    memcpy(id.internal, unique_id, kUniqueIdBytes);
    const int r = g_rccl.CommInitRank(&c->comm, world, id, rank);       // collective: every rank calls it with the same id
    if (r != 0) {
        set_error(std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r) + " [" + g_rccl.origin + "]");
        delete c;
        return 1;
    }
    *out = c;
    return 0;
}

void rldm_comm_destroy(rldm_comm* c) {
    if (!c) return;
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
}

int rldm_comm_info(const rldm_comm* c, int* rank, int* world, char* rccl_origin, size_t cap) {
    RLDM_REQUIRE(c != nullptr, "null communicator");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (rccl_origin && cap) {
        strncpy(rccl_origin, g_rccl.origin.c_str(), cap - 1);
        rccl_origin[cap - 1] = 0;
    }
    return 0;
}

// every rank contributes `count` floats (its finished images, (B_local, 2, W, H) contiguous) and receives all of them in
// rank order: all[(r * count) ...] = rank r's buffer.  In place when local == all + rank * count.
int rldm_allgather_images(rldm_comm* c, const float* local, float* all, int64_t count, void* stream) {
    RLDM_REQUIRE(c && local && all && count >= 0, "null argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (count == 0) return 0;
    RLDM_RCCL_CHECK(g_rccl.AllGather(local, all, (size_t)count, kFloat32, c->comm, st));       // (world 1 included: RCCL copies)
    return 0;
}

// in-place sum (average != 0: mean) of `count` floats over the ranks: one bucket of the flat gradient buffer
int rldm_allreduce_grads(rldm_comm* c, float* grads, int64_t count, int average, void* stream) {
    RLDM_REQUIRE(c && grads && count >= 0, "null argument");
    if (count == 0) return 0;
    RLDM_RCCL_CHECK(g_rccl.AllReduce(grads, grads, (size_t)count, kFloat32, average ? kAvg : kSum, c->comm,
                                     reinterpret_cast<hipStream_t>(stream)));
    return 0;
}

}  // extern "C"
