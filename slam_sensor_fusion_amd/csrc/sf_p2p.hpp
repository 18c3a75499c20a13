// sf_p2p.hpp — layout of a rank's P2P exchange region and the device-side view of a P2P communicator, shared by
// sf_shard.cpp (the stand-alone all-reduce) and sf_icp.hip (the sharded iteration, which publishes the records from its
// reduce kernel and solves in the kernel that gathers them).  See sf_shard.cpp for the protocol.
#pragma once
#include "sf_common.hpp"

namespace sf {

constexpr int P2P_MAX_RANKS = 16;
constexpr size_t P2P_LINE = 128;
constexpr int P2P_BLK = 1024;
constexpr int P2P_CHUNK = 2048;      // doubles one workgroup of the stand-alone all-reduce carries
constexpr int P2P_MAX_CHUNKS = 64;   // (a 512-scan batch at 8 GPUs is 16 384 doubles = 8 chunks)
constexpr int P2P_MAX_SCANS = P2P_CHUNK * P2P_MAX_CHUNKS / 32; // records of 32 doubles
constexpr size_t P2P_FLAGS_OFF = 0;                                                 // uint64 flag[chunk][r] at (chunk * 16 + r) * 128
constexpr size_t P2P_ABORT_OFF = P2P_LINE * P2P_MAX_RANKS * P2P_MAX_CHUNKS;         // uint32
constexpr size_t P2P_SFLAGS_OFF = P2P_ABORT_OFF + 2 * P2P_LINE;                     // uint64 sflag[scan][r]: per-record flags of the fused sharded step
constexpr size_t P2P_SLOTS_OFF = P2P_SFLAGS_OFF + sizeof(unsigned long long) * P2P_MAX_RANKS * (size_t)P2P_MAX_SCANS; // double slot[2][nranks][max_count]

struct P2pPeers { unsigned char *region[P2P_MAX_RANKS]; int nranks, rank; };

// what a kernel needs of a P2P communicator for ONE collective (sequence number already advanced)
struct P2pView {
    P2pPeers peers;
    int64_t max_count;
    unsigned long long seq;
    long long spin_ticks;
    uint32_t *status; // pinned host memory: 0 ok, 1 timed out waiting for a peer, 2 aborted
};

__device__ __forceinline__ unsigned long long *p2p_flag(unsigned char *region, int chunk, int r)
{
    return reinterpret_cast<unsigned long long *>(region + P2P_FLAGS_OFF + ((size_t)chunk * P2P_MAX_RANKS + (size_t)r) * P2P_LINE);
}
__device__ __forceinline__ unsigned long long *p2p_sflag(unsigned char *region, int scan, int r)
{
    return reinterpret_cast<unsigned long long *>(region + P2P_SFLAGS_OFF) + (size_t)scan * P2P_MAX_RANKS + (size_t)r;
}
__device__ __forceinline__ uint32_t *p2p_abort(unsigned char *region) { return reinterpret_cast<uint32_t *>(region + P2P_ABORT_OFF); }
__device__ __forceinline__ double *p2p_slot(unsigned char *region, int parity, int r, int nranks, int64_t max_count)
{
    return reinterpret_cast<double *>(region + P2P_SLOTS_OFF) + ((size_t)parity * (size_t)nranks + (size_t)r) * (size_t)max_count;
}

// sf_shard.cpp: 1 if `c` is a connected P2P communicator that can carry `count` doubles: *v is then filled for the NEXT
// collective (the sequence number is advanced: the caller must enqueue exactly one publish + gather with it); 0 if the
// communicator is of another kind; < 0 on error (poisoned, too small)
int comm_p2p_begin(sf_comm *c, int64_t count, P2pView *v);

} // namespace sf
