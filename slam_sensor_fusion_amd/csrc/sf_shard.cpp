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
#include "sf_p2p.hpp"

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>
#include <ctime>

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

// ------------------------------------------------------------------ P2P transport (option ii)
// The record all-reduce without RCCL: every rank owns an exchange region in its device memory (uncached, so that neither
// a per-XCD L2 nor a peer's cache keeps a stale copy), exports it as a hipIpc handle and maps its peers' regions.  Per
// all-reduce ONE kernel of one workgroup runs on the context's stream:
//   publish  every thread stores its elements of the local buffer into slot[parity][me] of EVERY rank's region (its own
//            included) with system-scope stores; every wave drains its stores, workgroup barrier, lane 0 system-scope
//            release fence, then lane r stores the sequence number into flag[me] of rank r's region;
//   wait     lane r polls flag[r] of the OWN region (relaxed system-scope loads, s_sleep) until it reaches the sequence
//            number -- bounded by a wall-clock limit and by the region's abort word; one system-scope acquire fence;
//   sum      every thread adds the nranks slots of its elements in FIXED RANK ORDER and writes the local buffer: the
//            result is bitwise identical on every rank and from run to run (an RCCL ring promises neither).
// Slots are double-buffered by the parity of the sequence number: a rank can pass the wait of collective k+1 only after
// every peer has published k+1, i.e. has finished reading k, so writing k+2 into the slots of k is safe.
// A rank that times out (a peer died or never arrived) or finds the abort word set raises the abort word of EVERY
// region it can reach and its own status word (pinned host memory): all ranks leave their collectives at once and
// sf_comm_status / sf_icp_align_sharded report SF_ERR_COMM instead of waiting in a collective.  sf_comm_abort does the
// same from the host (an error return on one rank mid-loop).  The same code serves 2-4 ranks as separate processes on ONE
// device (how the sharded loop is tested on a one-GPU box) and 8 ranks over xGMI.
namespace {

using namespace sf; // layout constants and device helpers: sf_p2p.hpp
constexpr uint32_t P2P_MAGIC = 0x32504653u; // "SFP2"

struct P2pHandle { // SF_COMM_P2P_HANDLE_BYTES: what the launcher hands round
    uint32_t magic;
    int32_t rank, nranks, device;
    int64_t pid;
    int64_t max_count;
    uint64_t region_bytes;
    uint64_t ptr; // same-process peers (several contexts in one process) use the pointer itself
    uint64_t nonce; // drawn once per process: a pid alone does not tell processes of different pid namespaces (containers: both often pid 1) apart
    hipIpcMemHandle_t ipc;
    char pad[128 - 56 - sizeof(hipIpcMemHandle_t)];
};
static_assert(sizeof(P2pHandle) == SF_COMM_P2P_HANDLE_BYTES, "handle blob size");

// what tells this process from every other on the node, whatever their pid namespaces
static uint64_t process_nonce()
{
    static const uint64_t nonce = [] {
        uint64_t v = 0;
        const int fd = open("/dev/urandom", O_RDONLY);
        if (fd >= 0) {
            if (read(fd, &v, sizeof(v)) != (ssize_t)sizeof(v)) v = 0;
            close(fd);
        }
        if (v == 0) v = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count() * 0x9E3779B97F4A7C15ull ^ (uint64_t)getpid() ^ (uint64_t)(uintptr_t)&v;
        return v | 1ull;
    }();
    return nonce;
}

// status word (pinned host memory): 0 ok, 1 timed out waiting for a peer, 2 aborted by a peer / the host.
// grid.x = chunks of P2P_CHUNK doubles; a workgroup carries its chunk through publish / signal / wait / sum on its own flags
__global__ __launch_bounds__(P2P_BLK) void k_p2p_allreduce(P2pPeers pr, double *__restrict__ buf, int count, int64_t max_count, unsigned long long seq, long long spin_ticks,
                                                          uint32_t *__restrict__ status)
{
    __shared__ int verdict; // 0 ok, 1 timeout, 2 abort
    const int tid = (int)threadIdx.x, R = pr.nranks, me = pr.rank, par = (int)(seq & 1ull), chunk = (int)blockIdx.x;
    const int i0 = chunk * P2P_CHUNK, i1 = min(i0 + P2P_CHUNK, count);
    unsigned char *mine = pr.region[me];
    if (tid == 0) verdict = __hip_atomic_load(p2p_abort(mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) ? 2 : 0;
    __syncthreads();
    if (verdict == 0) {
        // publish
        for (int i = i0 + tid; i < i1; i += P2P_BLK) {
            const double v = buf[i];
            for (int r = 0; r < R; ++r) __hip_atomic_store(p2p_slot(pr.region[r], par, me, R, max_count) + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        // signal, then wait for every rank's signal (lane r: rank r)
        if (tid < R) {
            __hip_atomic_store(p2p_flag(pr.region[tid], chunk, me), seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const long long t0 = wall_clock64();
            int v = 0;
            while (__hip_atomic_load(p2p_flag(mine, chunk, tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
                __builtin_amdgcn_s_sleep(2);
                if (__hip_atomic_load(p2p_abort(mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) { v = 2; break; }
                if (wall_clock64() - t0 > spin_ticks) { v = 1; break; }
            }
            if (v) atomicMax(&verdict, v);
        }
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }
    if (verdict != 0) { // poison every region this rank reaches, report, leave the buffer alone
        if (tid < R) __hip_atomic_store(p2p_abort(pr.region[tid]), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (tid == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u)
            __hip_atomic_store(status, (uint32_t)verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    // sum in rank order
    for (int i = i0 + tid; i < i1; i += P2P_BLK) {
        double s = 0.0;
        for (int r = 0; r < R; ++r) s += __hip_atomic_load(p2p_slot(mine, par, r, R, max_count) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        buf[i] = s;
    }
}

__global__ void k_p2p_abort(P2pPeers pr)
{
    const int tid = (int)threadIdx.x;
    if (tid < pr.nranks && pr.region[tid]) __hip_atomic_store(p2p_abort(pr.region[tid]), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

} // namespace

struct sf_comm {
    sf_ctx *ctx = nullptr;
    int nranks = 1, rank = 0;
    int kind = SF_COMM_RCCL;
    // RCCL
    void *comm = nullptr; // ncclComm_t
    // P2P
    unsigned char *region = nullptr; // this rank's exchange region (device memory, uncached)
    size_t region_bytes = 0;
    int64_t max_count = 0;
    P2pPeers peers{};
    bool opened[P2P_MAX_RANKS] = {};  // peers mapped through hipIpcOpenMemHandle
    bool connected = false;
    unsigned long long seq = 0;       // collectives enqueued so far (every rank counts the same)
    uint32_t *status = nullptr;       // pinned, device-visible
    double timeout_s = 20.0;
    int64_t n_collectives = 0;
};

namespace {

int p2p_status_to_rc(const sf_comm *c)
{
    const uint32_t st = c->status ? *reinterpret_cast<volatile uint32_t *>(c->status) : 0u;
    if (st == 0u) return SF_OK;
    sf::set_error(st == 1u ? "P2P collective: rank %d of %d timed out waiting for a peer (%.1f s); every rank of the communicator has been told to stop"
                           : "P2P collective: rank %d of %d was stopped by a peer's (or the host's) abort", c->rank, c->nranks, c->timeout_s);
    return SF_ERR_COMM;
}

void p2p_release(sf_comm *c)
{
    for (int r = 0; r < c->nranks && r < P2P_MAX_RANKS; ++r)
        if (c->opened[r] && c->peers.region[r]) { hipError_t e = hipIpcCloseMemHandle(c->peers.region[r]); (void)e; c->opened[r] = false; }
    if (c->region) { hipError_t e = hipFree(c->region); (void)e; c->region = nullptr; }
    if (c->status) { hipError_t e = hipHostFree(c->status); (void)e; c->status = nullptr; }
}

} // namespace

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

// P2P, step 1: this rank's region.  max_count = the largest all-reduce (doubles) the communicator will carry.
extern "C" int sf_comm_p2p_create(sf_ctx *ctx, int nranks, int rank, int64_t max_count, sf_comm **out)
{
    SF_CHECK(ctx && out && nranks >= 1 && nranks <= P2P_MAX_RANKS && rank >= 0 && rank < nranks && max_count >= 1 && max_count <= (int64_t)P2P_CHUNK * P2P_MAX_CHUNKS, SF_ERR_INVALID,
             "bad arguments (1..%d ranks, 1..%d doubles)", P2P_MAX_RANKS, P2P_CHUNK * P2P_MAX_CHUNKS);
    SF_HIP(hipSetDevice(ctx->device));
    sf_comm *c = new (std::nothrow) sf_comm();
    SF_CHECK(c, SF_ERR_NOMEM, "out of host memory");
    c->kind = SF_COMM_P2P;
    c->ctx = ctx;
    c->nranks = nranks;
    c->rank = rank;
    c->max_count = max_count;
    c->region_bytes = P2P_SLOTS_OFF + sizeof(double) * 2 * (size_t)nranks * (size_t)max_count;
    void *p = nullptr;
    // uncached: peers' stores must never meet a stale line in an L2 of this device
    hipError_t e = hipExtMallocWithFlags(&p, c->region_bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipExtMallocWithFlags(&p, c->region_bytes, hipDeviceMallocFinegrained); }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        sf::set_error("P2P exchange region (%zu bytes, uncached): %s", c->region_bytes, hipGetErrorString(e));
        delete c;
        return SF_ERR_NOMEM;
    }
    c->region = static_cast<unsigned char *>(p);
    int rc = SF_OK;
    if (hipMemset(c->region, 0, c->region_bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = SF_ERR_HIP;
    if (rc == SF_OK && hipHostMalloc((void **)&c->status, 64, hipHostMallocDefault) != hipSuccess) rc = SF_ERR_NOMEM;
    if (rc != SF_OK) {
        sf::set_error("P2P communicator setup failed: %s", hipGetErrorString(hipGetLastError()));
        p2p_release(c);
        delete c;
        return rc;
    }
    std::memset(c->status, 0, 64);
    c->peers.nranks = nranks;
    c->peers.rank = rank;
    c->peers.region[rank] = c->region;
    c->connected = nranks == 1;
    sf::ctx_retain(ctx);
    *out = c;
    return SF_OK;
}

// step 2: the blob the launcher hands to every other rank (torch.distributed, MPI, a file, sf_comm_p2p_rendezvous)
extern "C" int sf_comm_p2p_handle(sf_comm *c, void *handle)
{
    SF_CHECK(c && handle && c->kind == SF_COMM_P2P, SF_ERR_INVALID, "not a P2P communicator");
    P2pHandle h;
    std::memset(&h, 0, sizeof(h));
    h.magic = P2P_MAGIC;
    h.rank = c->rank;
    h.nranks = c->nranks;
    h.device = c->ctx->device;
    h.pid = (int64_t)getpid();
    h.nonce = process_nonce();
    h.max_count = c->max_count;
    h.region_bytes = c->region_bytes;
    h.ptr = (uint64_t)(uintptr_t)c->region;
    if (c->nranks > 1) SF_HIP(hipIpcGetMemHandle(&h.ipc, c->region));
    std::memcpy(handle, &h, sizeof(h));
    return SF_OK;
}

// step 3: map the peers (handles: nranks blobs in rank order, this rank's own included)
extern "C" int sf_comm_p2p_connect(sf_comm *c, const void *handles)
{
    SF_CHECK(c && handles && c->kind == SF_COMM_P2P, SF_ERR_INVALID, "not a P2P communicator");
    SF_CHECK(!c->connected || c->nranks == 1, SF_ERR_STATE, "already connected");
    SF_HIP(hipSetDevice(c->ctx->device));
    const P2pHandle *hs = static_cast<const P2pHandle *>(handles);
    for (int r = 0; r < c->nranks; ++r) {
        P2pHandle h;
        std::memcpy(&h, hs + r, sizeof(h));
        SF_CHECK(h.magic == P2P_MAGIC && h.rank == r && h.nranks == c->nranks && h.max_count == c->max_count && h.region_bytes == c->region_bytes, SF_ERR_INVALID,
                 "handle %d does not describe rank %d of this communicator (ranks %d, capacity %lld)", r, r, c->nranks, (long long)c->max_count);
        if (r == c->rank) continue;
        if (h.pid == (int64_t)getpid() && h.nonce == process_nonce()) { // several ranks inside one process: the pointer is valid as it is
            if (h.device != c->ctx->device) {
                hipError_t e = hipDeviceEnablePeerAccess(h.device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) SF_HIP(e);
                (void)hipGetLastError();
            }
            c->peers.region[r] = reinterpret_cast<unsigned char *>((uintptr_t)h.ptr);
            continue;
        }
        void *p = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&p, h.ipc, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            sf::set_error("hipIpcOpenMemHandle(rank %d, device %d) failed: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 must be set on this pool)", r, h.device, hipGetErrorString(e));
            (void)hipGetLastError();
            return SF_ERR_HIP;
        }
        c->peers.region[r] = static_cast<unsigned char *>(p);
        c->opened[r] = true;
    }
    c->connected = true;
    return SF_OK;
}

// steps 2 + 3 for ranks of ONE node without a launcher: the blobs meet in a POSIX shared-memory object `name` (the same
// string on every rank, unique per communicator; rank 0 removes it when all ranks have read it)
extern "C" int sf_comm_p2p_rendezvous(sf_comm *c, const char *name, double timeout_s)
{
    SF_CHECK(c && name && name[0] && c->kind == SF_COMM_P2P && timeout_s > 0, SF_ERR_INVALID, "bad arguments");
    if (c->nranks == 1) return SF_OK;
    struct Board { uint32_t posted[P2P_MAX_RANKS]; uint32_t done[P2P_MAX_RANKS]; P2pHandle blob[P2P_MAX_RANKS]; };
    std::string path = name[0] == '/' ? std::string(name) : "/" + std::string(name);
    // An object of this name left behind by a run that died (its rank 0 never unlinked it) would show posted[] = 1 with blobs
    // that pass every check.  The rank that CREATES the object (O_EXCL) is the only one that may find it fresh by
    // construction; a rank that finds it existing accepts it only while it is young -- every rank of a live rendezvous
    // arrives within the time limit of the first -- and otherwise removes it and starts over.
    int fd = -1;
    for (int attempt = 0; attempt < 4 && fd < 0; ++attempt) {
        fd = shm_open(path.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd >= 0) break;
        SF_CHECK(errno == EEXIST, SF_ERR_STATE, "shm_open(%s): %s", path.c_str(), std::strerror(errno));
        fd = shm_open(path.c_str(), O_RDWR, 0600);
        if (fd < 0) continue; // removed in between (a peer's rank 0 has finished, or it cleared a stale one): try to create again
        struct stat sb;
        const double limit = std::max(2.0 * timeout_s, 30.0);
        if (fstat(fd, &sb) == 0 && std::difftime(std::time(nullptr), sb.st_mtime) > limit) {
            close(fd);
            fd = -1;
            shm_unlink(path.c_str());
        }
    }
    SF_CHECK(fd >= 0, SF_ERR_STATE, "shm_open(%s): could not create or join the rendezvous object", path.c_str());
    if (ftruncate(fd, (off_t)sizeof(Board)) != 0) { // a fresh object is zero-filled; growing an existing one to the same size changes nothing
        const int err = errno;
        close(fd);
        sf::set_error("ftruncate(%s): %s", path.c_str(), std::strerror(err));
        return SF_ERR_STATE;
    }
    Board *bd = static_cast<Board *>(mmap(nullptr, sizeof(Board), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
    close(fd);
    SF_CHECK(bd != MAP_FAILED, SF_ERR_STATE, "mmap(%s): %s", path.c_str(), std::strerror(errno));
    int rc = sf_comm_p2p_handle(c, &bd->blob[c->rank]);
    const auto t0 = std::chrono::steady_clock::now();
    auto late = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s; };
    std::vector<P2pHandle> all((size_t)c->nranks);
    if (rc == SF_OK) {
        __atomic_store_n(&bd->posted[c->rank], 1u, __ATOMIC_RELEASE);
        for (int r = 0; r < c->nranks && rc == SF_OK; ++r) {
            while (__atomic_load_n(&bd->posted[r], __ATOMIC_ACQUIRE) != 1u) {
                if (late()) { sf::set_error("P2P rendezvous %s: rank %d did not arrive within %.0f s", path.c_str(), r, timeout_s); rc = SF_ERR_COMM; break; }
                usleep(200);
            }
            if (rc == SF_OK) std::memcpy(&all[(size_t)r], &bd->blob[r], sizeof(P2pHandle));
        }
    }
    if (rc == SF_OK) rc = sf_comm_p2p_connect(c, all.data());
    __atomic_store_n(&bd->done[c->rank], rc == SF_OK ? 1u : 2u, __ATOMIC_RELEASE);
    if (c->rank == 0) { // the object may go once nobody needs it any more (or after the time limit)
        for (int r = 0; r < c->nranks; ++r)
            while (__atomic_load_n(&bd->done[r], __ATOMIC_ACQUIRE) == 0u && !late()) usleep(200);
        shm_unlink(path.c_str());
    }
    if (rc == SF_OK) // a peer that failed to connect will never take part: do not start collectives with it
        for (int r = 0; r < c->nranks && rc == SF_OK; ++r) {
            while (__atomic_load_n(&bd->done[r], __ATOMIC_ACQUIRE) == 0u && !late()) usleep(200);
            if (__atomic_load_n(&bd->done[r], __ATOMIC_ACQUIRE) != 1u) { sf::set_error("P2P rendezvous %s: rank %d could not connect", path.c_str(), r); rc = SF_ERR_COMM; }
        }
    munmap(bd, sizeof(Board));
    return rc;
}

extern "C" int sf_comm_set_timeout(sf_comm *c, double seconds)
{
    SF_CHECK(c && seconds > 0 && seconds <= 600, SF_ERR_INVALID, "timeout must lie in (0, 600] s");
    c->timeout_s = seconds;
    return SF_OK;
}

// poison the communicator from the host: every rank's pending and future collectives return at once (P2P), the RCCL
// communicator is aborted (ncclCommAbort) when the library has it
extern "C" int sf_comm_abort(sf_comm *c)
{
    SF_CHECK(c, SF_ERR_INVALID, "comm is NULL");
    if (c->kind == SF_COMM_P2P) {
        if (c->status && *reinterpret_cast<volatile uint32_t *>(c->status) == 0u) *reinterpret_cast<volatile uint32_t *>(c->status) = 2u;
        hipStream_t side = nullptr; // not the context's stream: a collective may be spinning on it
        if (hipSetDevice(c->ctx->device) == hipSuccess && hipStreamCreateWithFlags(&side, hipStreamNonBlocking) == hipSuccess) {
            hipLaunchKernelGGL(k_p2p_abort, dim3(1), dim3(64), 0, side, c->peers);
            hipError_t e = hipStreamSynchronize(side);
            (void)e;
            e = hipStreamDestroy(side);
            (void)e;
        }
        (void)hipGetLastError();
        return SF_OK;
    }
    typedef int (*CommAbortFn)(void *);
    CommAbortFn ab = g_rccl.handle ? (CommAbortFn)dlsym(g_rccl.handle, "ncclCommAbort") : nullptr;
    if (ab && c->comm) { ab(c->comm); c->comm = nullptr; }
    return SF_OK;
}

// 0 while every collective so far went through; SF_ERR_COMM once one timed out or was aborted (call after synchronising)
extern "C" int sf_comm_status(sf_comm *c)
{
    SF_CHECK(c, SF_ERR_INVALID, "comm is NULL");
    if (c->kind == SF_COMM_P2P) return p2p_status_to_rc(c);
    SF_CHECK(c->comm, SF_ERR_COMM, "the RCCL communicator has been aborted");
    return SF_OK;
}

extern "C" int sf_comm_kind(const sf_comm *c) { return c ? c->kind : SF_ERR_INVALID; }

extern "C" void sf_comm_destroy(sf_comm *c)
{
    if (!c) return;
    hipError_t e = hipStreamSynchronize(c->ctx->stream);
    (void)e;
    if (c->kind == SF_COMM_P2P) p2p_release(c);
    else if (c->comm && g_rccl.comm_destroy) g_rccl.comm_destroy(c->comm);
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
int comm_p2p_begin(sf_comm *c, int64_t count, P2pView *v)
{
    if (!c || c->kind != SF_COMM_P2P) return 0;
    SF_CHECK(c->connected, SF_ERR_STATE, "P2P communicator is not connected (sf_comm_p2p_connect / sf_comm_p2p_rendezvous)");
    SF_CHECK(count <= c->max_count && count / 32 <= P2P_MAX_SCANS, SF_ERR_INVALID, "collective of %lld doubles exceeds the communicator's capacity %lld", (long long)count,
             (long long)c->max_count);
    SF_TRY(p2p_status_to_rc(c)); // poisoned: do not enqueue more
    c->seq += 1;
    c->n_collectives += 1;
    v->peers = c->peers;
    v->max_count = c->max_count;
    v->seq = c->seq;
    v->spin_ticks = (long long)(c->timeout_s * 1e8);
    v->status = c->status;
    return 1;
}

// in-place sum of `count` float64 on the communicator's context stream (enqueue only)
int comm_allreduce_f64(sf_comm *c, void *d_buf, int64_t count)
{
    SF_CHECK(c && d_buf && count >= 0, SF_ERR_INVALID, "bad arguments");
    if (count == 0) return SF_OK;
    if (c->kind == SF_COMM_P2P) {
        SF_CHECK(c->connected, SF_ERR_STATE, "P2P communicator is not connected (sf_comm_p2p_connect / sf_comm_p2p_rendezvous)");
        SF_CHECK(count <= c->max_count, SF_ERR_INVALID, "all-reduce of %lld doubles exceeds the communicator's capacity %lld", (long long)count, (long long)c->max_count);
        SF_TRY(p2p_status_to_rc(c)); // poisoned: do not enqueue more
        c->seq += 1;
        c->n_collectives += 1;
        hipLaunchKernelGGL(k_p2p_allreduce, dim3((unsigned)sf::div_up(count, P2P_CHUNK)), dim3(P2P_BLK), 0, c->ctx->stream, c->peers, static_cast<double *>(d_buf), (int)count, c->max_count, c->seq,
                           (long long)(c->timeout_s * 1e8), c->status); // wall_clock64 ticks at 100 MHz
        SF_HIP(hipGetLastError());
        return SF_OK;
    }
    SF_CHECK(c->comm, SF_ERR_COMM, "the RCCL communicator has been aborted");
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
