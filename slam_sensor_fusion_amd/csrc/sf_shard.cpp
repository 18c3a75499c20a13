// sf_shard.cpp — multi-GPU plumbing of the sharded registration: the communicator the C side
// all-reduces the normal-equation records on (RCCL over xGMI, resolved at run time), and the routing of scans
// to the map slabs they touch ("all-reduce only when the submap spans tiles", north_star).
//
// No reference counterpart: the reference is one process on one CPU (localization/src/main.cpp:18).
//
// RCCL is bound through dlopen / dlsym rather than at link time: a process must use ONE HIP runtime and ONE RCCL,
// and under Python that is the pair the torch wheel ships (torch is only the launcher / rendezvous here), while a
// C++ host uses /opt/rocm's.  sf_comm_load_rccl names the library; the default is whatever the process has loaded.
#include "sf_common.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <string>

namespace {

// the slice of the RCCL ABI that is used (rccl.h: ncclResult_t = int, ncclUniqueId = 128 bytes, ncclFloat64 = 8, ncclSum = 0)
struct UniqueId { char internal[128]; };
typedef int (*GetUniqueIdFn)(UniqueId *);
typedef int (*CommInitRankFn)(void **, int, UniqueId, int);
typedef int (*CommDestroyFn)(void *);
typedef int (*AllReduceFn)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef const char *(*GetErrorStringFn)(int);
constexpr int NCCL_FLOAT64 = 8, NCCL_SUM = 0;

struct Rccl {
    void *handle = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    AllReduceFn all_reduce = nullptr;
    GetErrorStringFn error_string = nullptr;
    std::string path;
} g_rccl;

int rccl_load(const char *path)
{
    if (g_rccl.handle && (!path || g_rccl.path == path)) return SF_OK;
    const char *candidates[] = {path, "librccl.so.1", "librccl.so"};
    void *h = nullptr;
    std::string used;
    for (const char *c : candidates) {
        if (!c) continue;
        h = dlopen(c, RTLD_NOW | RTLD_NOLOAD); // the copy the process already uses, if any
        if (!h) h = dlopen(c, RTLD_NOW | RTLD_LOCAL);
        if (h) { used = c; break; }
        if (path) break; // an explicit path does not fall back silently
    }
    SF_CHECK(h, SF_ERR_STATE, "cannot load RCCL (%s): %s", path ? path : "librccl.so.1", dlerror());
    Rccl r;
    r.handle = h;
    r.path = used;
    r.get_unique_id = (GetUniqueIdFn)dlsym(h, "ncclGetUniqueId");
    r.comm_init_rank = (CommInitRankFn)dlsym(h, "ncclCommInitRank");
    r.comm_destroy = (CommDestroyFn)dlsym(h, "ncclCommDestroy");
    r.all_reduce = (AllReduceFn)dlsym(h, "ncclAllReduce");
    r.error_string = (GetErrorStringFn)dlsym(h, "ncclGetErrorString");
    SF_CHECK(r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.all_reduce, SF_ERR_STATE, "%s does not export the NCCL API", used.c_str());
    g_rccl = r;
    return SF_OK;
}

const char *rccl_error(int rc) { return g_rccl.error_string ? g_rccl.error_string(rc) : "?"; }

} // namespace

struct sf_comm {
    sf_ctx *ctx = nullptr;
    void *comm = nullptr; // ncclComm_t
    int nranks = 1, rank = 0;
};

extern "C" int sf_comm_load_rccl(const char *library_path) { return rccl_load(library_path); }

extern "C" int sf_comm_unique_id(void *id128)
{
    SF_CHECK(id128, SF_ERR_INVALID, "bad arguments");
    SF_TRY(rccl_load(nullptr));
    UniqueId id;
    const int rc = g_rccl.get_unique_id(&id);
    SF_CHECK(rc == 0, SF_ERR_HIP, "ncclGetUniqueId: %s", rccl_error(rc));
    std::memcpy(id128, &id, sizeof(id));
    return SF_OK;
}

extern "C" int sf_comm_create(sf_ctx *ctx, int nranks, int rank, const void *id128, sf_comm **out)
{
    SF_CHECK(ctx && out && id128 && nranks >= 1 && rank >= 0 && rank < nranks, SF_ERR_INVALID, "bad arguments");
    SF_TRY(rccl_load(nullptr));
    SF_HIP(hipSetDevice(ctx->device));
    UniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    void *comm = nullptr;
    const int rc = g_rccl.comm_init_rank(&comm, nranks, id, rank);
    SF_CHECK(rc == 0 && comm, SF_ERR_HIP, "ncclCommInitRank(%d of %d): %s", rank, nranks, rccl_error(rc));
    sf_comm *c = new (std::nothrow) sf_comm();
    if (!c) {
        g_rccl.comm_destroy(comm);
        sf::set_error("out of host memory");
        return SF_ERR_NOMEM;
    }
    c->ctx = ctx;
    c->comm = comm;
    c->nranks = nranks;
    c->rank = rank;
    sf::ctx_retain(ctx);
    *out = c;
    return SF_OK;
}

extern "C" void sf_comm_destroy(sf_comm *c)
{
    if (!c) return;
    hipError_t e = hipStreamSynchronize(c->ctx->stream);
    (void)e;
    if (c->comm && g_rccl.comm_destroy) g_rccl.comm_destroy(c->comm);
    sf_ctx *ctx = c->ctx;
    delete c;
    sf::ctx_release(ctx);
}

extern "C" int sf_comm_size(const sf_comm *c, int *nranks, int *rank)
{
    SF_CHECK(c, SF_ERR_INVALID, "comm is NULL");
    if (nranks) *nranks = c->nranks;
    if (rank) *rank = c->rank;
    return SF_OK;
}

namespace sf {
sf_ctx *comm_ctx(const sf_comm *c) { return c ? c->ctx : nullptr; }
// in-place sum of `count` float64 on the communicator's context stream (enqueue only)
int comm_allreduce_f64(sf_comm *c, void *d_buf, int64_t count)
{
    SF_CHECK(c && c->comm && d_buf && count >= 0, SF_ERR_INVALID, "bad arguments");
    if (count == 0) return SF_OK;
    const int rc = g_rccl.all_reduce(d_buf, d_buf, (size_t)count, NCCL_FLOAT64, NCCL_SUM, c->comm, c->ctx->stream);
    SF_CHECK(rc == 0, SF_ERR_HIP, "ncclAllReduce: %s", rccl_error(rc));
    return SF_OK;
}
} // namespace sf

extern "C" int sf_comm_allreduce_f64(sf_comm *c, void *d_buf, int64_t count) { return sf::comm_allreduce_f64(c, d_buf, count); }

// ------------------------------------------------------------------ routing
// Slabs are the x-intervals [edges[s], edges[s+1]) (edges[0] = -inf, edges[n_slabs] = +inf).  A scan is routed to
// every slab its points can reach: the x-extent of its axis-aligned bounding box under its initial pose (an affine
// map of a box reaches its extremes at the corners), widened by `margin` on both sides.  margin must cover how far
// the registration may move the scan plus the correspondence distance; a scan whose range is one slab needs no
// collective at all.
extern "C" int sf_shard_route(const float *xyz, int64_t n_per_scan, int batch, const double *inits, const double *edges, int n_slabs, double margin, int32_t *lo,
                              int32_t *hi)
{
    SF_CHECK((xyz || n_per_scan == 0) && n_per_scan >= 0 && batch >= 1 && edges && n_slabs >= 1 && lo && hi && margin >= 0.0, SF_ERR_INVALID, "bad arguments");
    for (int s = 0; s + 1 < n_slabs + 1; ++s) SF_CHECK(edges[s] <= edges[s + 1], SF_ERR_INVALID, "slab edges must ascend");
    for (int b = 0; b < batch; ++b) {
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        const float *p = xyz + 3 * (size_t)b * (size_t)n_per_scan;
        int64_t finite = 0;
        for (int64_t i = 0; i < n_per_scan; ++i, p += 3) {
            if (!(std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]))) continue;
            ++finite;
            for (int d = 0; d < 3; ++d) { mn[d] = std::min(mn[d], p[d]); mx[d] = std::max(mx[d], p[d]); }
        }
        if (finite == 0) { lo[b] = 0; hi[b] = -1; continue; } // nothing to register: no slab
        double ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        const double *T = inits ? inits + 16 * (size_t)b : ident;
        double x0 = INFINITY, x1 = -INFINITY;
        for (int c = 0; c < 8; ++c) {
            const double x = (c & 1) ? mx[0] : mn[0], y = (c & 2) ? mx[1] : mn[1], z = (c & 4) ? mx[2] : mn[2];
            const double q = T[0] * x + T[1] * y + T[2] * z + T[3];
            x0 = std::min(x0, q);
            x1 = std::max(x1, q);
        }
        x0 -= margin;
        x1 += margin;
        int a = 0, e = n_slabs - 1;
        while (a + 1 < n_slabs && edges[a + 1] <= x0) ++a; // first slab whose upper edge lies beyond x0
        while (e > 0 && edges[e] > x1) --e;                // last slab whose lower edge is not beyond x1
        lo[b] = a;
        hi[b] = std::max(a, e);
    }
    return SF_OK;
}
