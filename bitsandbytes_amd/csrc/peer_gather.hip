// peer_gather.hip — one-shot all-gather of small per-rank outputs over peer-mapped buffers (xGMI), one kernel per collective.
//
// New functionality on top of the reference, which has no collective code at all. It serves the one exchange step
// of the sharded 4-bit linear layer (bitsandbytes_amd/parallel.py): every rank owns N / G output features and all of
// them need all of y. At decode sizes the message is 2.7 KB per rank (M = 1, Llama FFN): pure latency, for which a ring
// collective is the wrong tool - so every rank WRITES its shard straight into slot `rank` of every peer's gather buffer
// (peer-to-peer stores over xGMI, one workgroup per destination), publishes one flag per destination, and waits for the
// G - 1 flags the peers set in its own buffer, copying every shard that has landed to the caller's output tensor. One launch,
// no host round trip; a plain kernel, so it can be captured in a hipGraph together with the launches around it (the result
// address is the caller's, the same in every replay; the landing area alternates underneath).
//
//  * Buffer of a rank (fine-grained device memory, exported to the peers as a hipIpc handle by the host layer):
//      [0]   u32 counter  - number of collectives completed on this buffer (the epoch lives on the DEVICE: a replayed graph
//                           re-runs the same launch, a host-side epoch argument would be frozen at capture)
//      [4]   u32 status   - sticky: 1 = a wait ran into its bound (a peer never arrived); the host layer raises on it
//      [8]   u32 done     - workgroups of the running launch that have finished
//      [64]  u32 flags[2][kPeerMaxWorld]   flags[e & 1][src] = e once rank src's shard of collective e has landed here
//      [256] data[2][world][slot_stride]   double-buffered by the parity of the collective
//  * Double buffering is enough: a rank can only be ONE collective ahead of a peer (it needs the peer's shard of collective e
//    to finish e, and the peer sends that at the start of its own launch e, i.e. after everything it enqueued before - its
//    reads of the result of e - 2 included).
//  * Order: payload stores (16-byte where alignment allows), system-scope release fence, barrier, then ONE system-scope
//    atomic store of the flag; the reader polls its own flag word with system-scope acquire loads (bounded - tens of seconds by default - then the
//    status word is set and the launch ends instead of hanging the queue) and only then lets the launch end - whatever runs
//    next on the stream sees complete data.
#include "bnb_common.h"

#include "../../include/bnb_mi355x.h"

namespace bnb {
namespace {

constexpr int kPeerMaxWorld = 8;
constexpr size_t kPeerFlagsOffset = 64;
constexpr size_t kPeerDataOffset = 256;
constexpr int kPeerThreads = 256;

struct PeerBufs {
    unsigned char* base[kPeerMaxWorld]; // base[r]: rank r's buffer as mapped into THIS process (base[rank] = the local one)
};

__device__ __forceinline__ uint32_t* peer_flag(unsigned char* base, uint32_t e, int src) {
    return reinterpret_cast<uint32_t*>(base + kPeerFlagsOffset) + (e & 1u) * kPeerMaxWorld + src;
}

// grid = world workgroups: workgroup p delivers this rank's shard to rank p, then waits for rank p's shard
template <typename V> __device__ __forceinline__ void peer_copy(unsigned char* dst, const unsigned char* src, uint32_t bytes, int tid) {
    const V* s = reinterpret_cast<const V*>(src);
    V* d = reinterpret_cast<V*>(dst);
    for (uint32_t i = tid; i < bytes / sizeof(V); i += kPeerThreads)
        d[i] = s[i];
}

__global__ __launch_bounds__(kPeerThreads) void peer_allgather_kernel(PeerBufs bufs, const unsigned char* __restrict__ src,
                                                                      unsigned char* __restrict__ out, uint32_t bytes,
                                                                      uint32_t slot_stride, int rank, int world, uint32_t spin_bound) {
    const int p = blockIdx.x;
    unsigned char* const local = bufs.base[rank];
    uint32_t* const hdr = reinterpret_cast<uint32_t*>(local);
    // (written by the previous launch on this stream, which has ended: plain load)
    const uint32_t e = hdr[0] + 1u;
    unsigned char* const dst = bufs.base[p] + kPeerDataOffset + (static_cast<size_t>(e & 1u) * world + rank) * slot_stride;

    // ---- payload (slots are 256-byte aligned)
    using V16 = __attribute__((ext_vector_type(4))) uint32_t;
    const int tid = threadIdx.x;
    const bool vec = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(out) | bytes) & 15u) == 0;
    if (vec)
        peer_copy<V16>(dst, src, bytes, tid);
    else
        peer_copy<unsigned char>(dst, src, bytes, tid);
    __threadfence_system(); // every lane's stores are visible system-wide before ...
    __syncthreads();
    __shared__ int landed;
    if (tid == 0) {
        landed = 1;
        // ... the flag that announces them
        __hip_atomic_store(peer_flag(bufs.base[p], e, rank), e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        // ---- wait for rank p's shard in the local buffer
        uint32_t* const mine = peer_flag(local, e, p);
        uint32_t spins = 0;
        while (__hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != e) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > spin_bound) {
                __hip_atomic_store(hdr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                landed = 0;
                break;
            }
        }
        // ---- the last workgroup of the launch advances the counter (visible to the next launch: kernel boundary)
        const uint32_t prev = __hip_atomic_fetch_add(hdr + 2, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == static_cast<uint32_t>(world - 1)) {
            hdr[2] = 0u;
            hdr[0] = e;
        }
    }
    __syncthreads();
    // ---- rank p's shard goes to the caller's output (every lane orders its reads behind the flag it did not read itself)
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    unsigned char* const o = out + static_cast<size_t>(p) * bytes;
    if (landed) {
        const unsigned char* const slot = local + kPeerDataOffset + (static_cast<size_t>(e & 1u) * world + p) * slot_stride;
        if (vec)
            peer_copy<V16>(o, slot, bytes, tid);
        else
            peer_copy<unsigned char>(o, slot, bytes, tid);
    } else {
        // the peer never arrived (status word set above): its rows of the result become all-ones bytes - a NaN in fp16, bf16 and
        // fp32 - instead of whatever the caller's buffer held: a timeout cannot pass for data downstream
        for (uint32_t i = tid; i < bytes; i += kPeerThreads)
            o[i] = 0xFFu;
    }
}

} // namespace
} // namespace bnb

using namespace bnb;

extern "C" {
// the library is built with -fvisibility=hidden: the C ABI declared in include/bnb_mi355x.h is ALL it exports
#pragma GCC visibility push(default)

// Bytes a rank's buffer needs for shards of up to `max_bytes` (rounded up to 256) in a group of `world` ranks.
size_t bnb_mi355x_peer_buffer_bytes(int world, size_t max_bytes) {
    const size_t stride = (max_bytes + 255) & ~static_cast<size_t>(255);
    return kPeerDataOffset + 2 * static_cast<size_t>(world) * stride;
}

// Fine-grained device memory on the current device, zeroed (counter, status and flags start at 0). NULL on failure.
void* bnb_mi355x_peer_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(p);
        return nullptr;
    }
    return p;
}
void bnb_mi355x_peer_free(void* p) {
    if (p != nullptr && hipFree(p) != hipSuccess)
        (void)hipGetLastError();
}
// IPC handle of a buffer from bnb_mi355x_peer_alloc (64 bytes, to be sent to the peers). 0 = ok.
int bnb_mi355x_peer_export(void* p, void* handle64) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, p) != hipSuccess) {
        (void)hipGetLastError();
        return 1;
    }
    __builtin_memcpy(handle64, &h, sizeof(h));
    return 0;
}
// Maps a peer's buffer into this process (current device must be able to reach it: same device or a P2P peer). NULL on failure.
void* bnb_mi355x_peer_open(const void* handle64) {
    hipIpcMemHandle_t h;
    __builtin_memcpy(&h, handle64, sizeof(h));
    void* p = nullptr;
    if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}
void bnb_mi355x_peer_close(void* p) {
    if (p != nullptr && hipIpcCloseMemHandle(p) != hipSuccess)
        (void)hipGetLastError();
}
// The collective. bufs[r] (HOST array of `world` device pointers, read during the call): rank r's buffer as mapped into this
// process, bufs[rank] the local one. src: this rank's shard (`bytes` <= the max_bytes the buffers were sized for); out: world x
// bytes, rank-major - what all_gather_into_tensor would produce.
void bnb_mi355x_peer_allgather(void* const* bufs, int world, int rank, const void* src, void* out, size_t bytes, size_t max_bytes,
                               bnb_stream_t stream) {
    if (world < 1 || world > kPeerMaxWorld || rank < 0 || rank >= world || bytes > max_bytes || max_bytes >= (1ull << 31)) {
        fprintf(stderr, "bitsandbytes_amd: peer_allgather: bad arguments (world %d, rank %d, bytes %zu of %zu)\n", world, rank, bytes,
                max_bytes);
        exit(1);
    }
    PeerBufs b;
    for (int r = 0; r < kPeerMaxWorld; ++r)
        b.base[r] = static_cast<unsigned char*>(r < world ? bufs[r] : nullptr);
    const uint32_t stride = static_cast<uint32_t>((max_bytes + 255) & ~static_cast<size_t>(255));
    // A poll is s_sleep 8 (512 cycles) plus a system-scope load round trip: ~1 us. 30 M polls = tens of seconds: long enough for
    // any skew between healthy ranks (a rank that is still compiling, logging, loading), short enough that a dead peer ends as
    // an error (status word, PeerAllGather.check()) instead of a queue that never drains. BNB_MI355X_PEER_WAIT_POLLS overrides.
    static const uint32_t spin_bound = [] {
        const char* e = getenv("BNB_MI355X_PEER_WAIT_POLLS");
        const long long v = e ? atoll(e) : 0;
        return static_cast<uint32_t>(v > 0 && v < 4000000000LL ? v : 30000000LL);
    }();
    hipLaunchKernelGGL(peer_allgather_kernel, dim3(world), dim3(kPeerThreads), 0, static_cast<hipStream_t>(stream), b,
                       static_cast<const unsigned char*>(src), static_cast<unsigned char*>(out), static_cast<uint32_t>(bytes), stride, rank,
                       world, spin_bound);
    BNB_CHECK_LAUNCH();
}
// 0 while every wait of every collective on this buffer found its peer; 1 once one ran into its bound (synchronises the device).
int bnb_mi355x_peer_status(const void* local_buffer) {
    unsigned v[2] = {0, 0};
    if (hipMemcpy(v, local_buffer, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return static_cast<int>(v[1]);
}

#pragma GCC visibility pop
} // extern "C"
