// gemv4_stream.hip — fused 4-bit dequantize + dot product for decode-sized batches (M <= 4 rows of A per pass;
// bf16 / fp16 / fp32 activations) on gfx950:  out[m, n] = sum_k A[m, k] * code[B[n, k]] * scale[n, k / bs]  (+ bias[n])
//
// Replaces, on MI355X, the reference's kgemm_4bit_inference_naive (csrc/kernels.cu:1452-1567) and the small-M range
// of gemm_4bit_simt (csrc/gemm_4bit_simt.cu:109-480) - one logical warp per output column, nibble-at-a-time decode -
// with a structure built from what round 1 measured on the hardware (profiles/r1_*):
//
//  * A 4096 x 4096 launch is 36 KiB of weights per CU: fixed costs decide. The old dot kernel spent ~4700 of its
//    ~9700 cycles before its decode table existed (a dependent global load of the code values, a 64 KiB table build
//    in each of the two workgroups of a CU) and then decoded every row at once with nothing left in flight. Here ONE
//    persistent workgroup per CU builds ONE table, from compile-time literals (the code values of NF4 / FP4 are
//    constants of the format, reference functional.py:788-823): no memory dependency in front of the table, which is
//    ready before the first weight bytes can arrive from HBM.
//  * The decode must overlap the stream. A wavefront owns one 2048-k segment of K for all of its rows, keeps that
//    slice of the activations in REGISTERS (fp32) for the whole launch, and walks its rows with an NS-deep register
//    ring of 1-KiB weight loads: while it decodes row i the loads of rows i+1 .. i+NS-1 are in flight (counted
//    vmcnt, never a drain). Lane l always owns bytes [16 l, 16 l + 16) of the 1-KiB row segment: every load
//    instruction is eight full 128-B lines.
//  * v_dot2c_f32_bf16 measured ~10-12 cycles per wave-instruction; v_fma_f32 issues in 2. The table therefore maps a
//    packed BYTE to the pair (code[hi], code[lo]) in fp32 (ds_read_b64, conflict-free: 32 bank-private copies, the
//    address of a look-up is one v_perm_b32) and the products are plain fp32 FMAs against the register-resident
//    activations: 1 + 2 M VALU issue slots per byte instead of 1 + ~5 M, and the code values are not rounded to
//    bf16 on the way (closer to the fp32 oracle than the old kernel). fp32 activations take the same path.
//  * The per-block scale multiplies the fp32 sum of a lane's 32-nibble run (never straddles a quantization block:
//    blocksize >= 32, K % 32 == 0). Nested (double-quantised) absmax is reconstructed in-kernel:
//    scale = absmax_code[absmax_8bit[b]] * absmax[b >> 8] + offset   (reference autograd/_functions.py:471-485).
//    Two roundings, as the host-side sequence (bnb_common.h nested_scale: written with __fmul_rn / __fadd_rn hipcc emitted ONE
//    v_fma_f32 until round 5 - a last-bit difference in the scale that the un-nested statistics of the sharded layers did not have).
//  * Segment partials of a row are combined in FIXED order (LDS slots, no atomics): results are bit-reproducible and
//    independent of the launch geometry (the reference's test_matmul_4bit_weight_orientation demands exact equality).
//
// Grouped launches (bnb_mi355x_gemm_4bit_grouped): several weight matrices that share the activations (Q/K/V,
// gate/up) are one launch over the concatenated row space - one boundary, one table build, one activation copy.
#include "bnb_common.h"

#include <type_traits>

namespace bnb {

#ifdef BNB_PROFILING
extern unsigned long long* g_dbg_buf; // c_api.hip (profiling builds)
#endif

namespace {

using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

typedef __attribute__((address_space(3))) unsigned char* lds_ptr;

constexpr int kSegK = 2048;       // k covered by one wavefront-wide 16-byte load
constexpr int kLutBytes = 65536;  // 256 entries x 32 copies x 8 B
constexpr int kMaxGroup = 8;      // matrices per grouped launch

enum StreamFlags : int {
    kNested = 1,  // double-quantised absmax reconstructed in-kernel
    kCodePtr = 2, // code values from a caller-supplied device table (legacy gemv_4bit op) instead of literals
    kFp4 = 4,     // literal table = FP4 (else NF4)
    kNT = 8,      // non-temporal weight loads
    kGrouped = 16, // several matrices over one concatenated row space
    kMulti = 32,   // rows longer than the workgroup's segment columns: several phases (a loop around the whole body)
    kPeer = 64     // "peer chain" form (M = 1): x taken from / y delivered to the ranks' exchange buffers - see PeerChain
};

// One weight matrix of a launch.
struct StreamMat {
    const uint8_t* B;
    const float* absmax;
    const uint8_t* absmax8;
    const float* absmax_code;
    const float* absmax_offset;
    void* out;        // [M, N] of this matrix
    const void* bias; // optional [N]
    int N;
    int row_start;    // first row of this matrix in the concatenated row space
};

// Profiling builds (-DBNB_PROFILING, libbitsandbytes_mi355x_prof.so, tools/ and bench.py's span leg only): per-wavefront
// s_memtime stamps. A stamp buffer address with bit 0 set asks for the two s_memrealtime stamps (slots 13 / 14: first
// instruction and end of every wavefront, 100 MHz constant clock) ONLY: the kernel's own span, first wavefront in to last
// wavefront out, without the cost of the sixteen phase stamps.
#ifdef BNB_PROFILING
#define BNB_ST_STAMP(i)                                                                            \
    {                                                                                              \
        if (p.dbg && !(reinterpret_cast<uintptr_t>(p.dbg) & 1) && lane == 0)                       \
            p.dbg[(static_cast<long>(blockIdx.x) * WAVES + wave) * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
    }
#define BNB_ST_DBG_BASE (reinterpret_cast<unsigned long long*>(reinterpret_cast<uintptr_t>(p.dbg) & ~static_cast<uintptr_t>(7)))
#else
#define BNB_ST_STAMP(i) {}
#endif

// Peer chain (kPeer instances; bitsandbytes_amd/peer.py PeerChain): the all-gather that re-assembles y of an
// N-sharded layer is FUSED into the kernels on either side of it. Every rank owns an exchange buffer (ordinary device memory,
// mapped into every process of the node by hipIpc; every store into it is a system-scope write-through store):
//     [4] u32 status  sticky, 1 = a wait ran into its bound          [8] u32 done  workgroups of the running read-out that finished
// and, in ORDINARY (cacheable) device memory that only its own launches touch, one epoch word: exchanges completed on this rank up
// to the last read-out. It lives on the DEVICE because a replayed hipGraph re-runs its launches; the launches of a chain carry
// their distance from it as an argument, and only the read-out that ends a chain - a handful of workgroups - advances it (an
// arrival counter over the 256 workgroups of every gemv launch cost 3 us per layer: 12 ns per atomic on one address; the word
// inside the fine-grained buffer cost ~1 us per launch: an uncached scalar load in front of the granule fetches and another one
// in front of the stores - profiles/r4_peer_chain.txt).
//     [256] granule[64][max_granules]  8 bytes each = {two consecutive T values of y, u32 tag}; exchange e in region e % 64
// Producer (mode bit 1): the thread that holds an even output row packs it with its neighbour's and stores the granule - ONE
// 8-byte system-scope store, data and tag indivisible - into slot (rank * ns + row) / 2 of EVERY rank's buffer, tag = epoch + 1.
// No fence, no flag, no counter: a granule is valid when its tag says so. Consumer (mode bit 0; the NEXT layer's launch): after
// its weight ring is requested, the builder wavefronts fetch the K / 2 granules of the current epoch with system-scope loads,
// re-fetch the ones whose tag is not there yet, and write the values into the activation image in the LDS - the exchange
// latency runs under the ~1.5 us the first weight bytes need anyway. Two parities are enough: a rank can start exchange e + 2
// (same parity as e) only after it consumed ALL of e + 1, which every peer stores at the END of the launch that consumed e.
struct PeerChain {
    unsigned char* base[8]; // the ranks' exchange buffers as mapped into this process; base[rank] = the local one
    unsigned char* local;   // = base[rank]
    uint32_t* epoch_word;   // this rank's epoch word (cacheable memory)
    int world, rank;
    uint32_t max_granules;  // granules per parity region
    uint32_t spin_bound;    // re-fetches of a granule before the launch gives up (status word, NaN)
    uint32_t epoch_offset;  // exchanges produced by earlier launches since the buffer's epoch word was last advanced
    int mode;               // bit 0: x = the current exchange (A is ignored); bit 1: y goes to the exchange (+ to out when non-NULL);
                            // bit 2: rows come in fours (ns and the rows per workgroup are multiples of 4): two granules per store;
                            // bit 3 (with bits 1 and 2): rows are (gate, up) pairs, the exchange gets silu(gate) * up - ns / 2 values
};
constexpr size_t kChainDataOffset = 256;
// The i-th exchange since the last read-out (i = epoch_offset + 1 of the launch that produces it) lives in region i % kChainRegions:
// the region of an exchange is its POSITION in the chain - known on the host, identical in every replay of a captured chain -
// and only its TAG (epoch word + i) tells one chain's use of a region from the next one's. So the consumer's fetch depends on no
// device-side word: it goes out with the first instructions of the launch (waiting for the epoch word first - a cold scalar load
// behind a cold kernarg load - put the x fetch 1300 cycles behind the plain kernel's, the whole +0.75 us of the consumer side;
// profiles/r4_timeline_chain.txt). Safe with chains of AT LEAST TWO exchanges (enforced by the host layer): a rank re-produces
// region i in the next chain only after its read-out of this one, i.e. after every rank produced the LAST exchange - which it does
// behind its consumption of all earlier ones; and the last region is re-produced only behind the consumption of the next chain's
// exchange n - 1, which every rank produces behind its own read-out.
constexpr uint32_t kChainRegions = 64;
constexpr int kChainRounds = 8; // 16-byte fetches per builder lane: K <= 8 * 8 KiB / 4 B = 16384 values
// "Gated" production (mode bit 3; one Llama-style FFN block on the chain, BASELINE.json configs[3]): the launch runs over a matrix
// whose rows INTERLEAVE this rank's gate and up rows - row 2 r = gate row r, row 2 r + 1 = up row r - so the thread pair that
// exchanges its outputs in the epilogue anyway holds g and u of ONE activation, and what goes to the exchange is
//     a[rank * ns / 2 + r] = T(T(silu(g)) * u)          (torch's arithmetic for `F.silu(g) * u` on 16-bit tensors: each op in fp32,
// rounded once to T) - half as many values as rows, two of them per granule, one 8-byte store per four rows. The down projection then
// consumes that exchange in the PLAIN form. (A first build computed the activation in the consumer - 28 exact expf + IEEE divisions
// per builder lane in front of the first barrier: +1.1 us per block, profiles/r5_peer_ffn.txt; here it is ~ R / 2 activations per
// workgroup, in an epilogue that waits for nothing.)
// re-fetches a wait is still worth once a wait on the same buffer has run into its bound (the status word is sticky): the peer
// is gone - every later round and launch would otherwise spin the full bound again, ~30 s each, before a host-side check() runs
constexpr uint32_t kPeerPollsAfterTimeout = 16;

struct StreamArgs {
#ifdef BNB_PROFILING
    unsigned long long* dbg;
#endif
    PeerChain peer{};
    const void* A;
    const float* code16;
    int M, K, bs_shift;
    int rows_total; // sum of N over the group
    int nmat;
    StreamMat mat[kMaxGroup];
};

template <bool FP4> __device__ __forceinline__ float code_literal(int i) {
    // a compare/select tree over literals: no memory access (15 v_cndmask, once per wavefront)
    constexpr float nf4[16] = {BNB_NF4_VALUES};
    constexpr float fp4[16] = {BNB_FP4_VALUES};
    float v = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j)
        v = (i == j) ? (FP4 ? fp4[j] : nf4[j]) : v;
    return v;
}

template <typename T> __device__ __forceinline__ void unpack16(const u32x4& v, float* dst);
template <> __device__ __forceinline__ void unpack16<bf16>(const u32x4& v, float* dst) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        dst[2 * i] = __builtin_bit_cast(float, v[i] << 16);
        dst[2 * i + 1] = __builtin_bit_cast(float, v[i] & 0xFFFF0000u);
    }
}
template <> __device__ __forceinline__ void unpack16<f16>(const u32x4& v, float* dst) {
    using h2 = __attribute__((ext_vector_type(2))) f16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t e = v[i]; // (a copy: __builtin_bit_cast applied to the vector-element lvalue reads element 0)
        const h2 h = __builtin_bit_cast(h2, e);
        dst[2 * i] = static_cast<float>(h[0]);
        dst[2 * i + 1] = static_cast<float>(h[1]);
    }
}
template <> __device__ __forceinline__ void unpack16<float>(const u32x4& v, float* dst) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t e = v[i]; // (a copy: see unpack16<f16>)
        dst[i] = __builtin_bit_cast(float, e);
    }
}

// T in {bf16, f16, float}; MB = activation rows held in registers; WAVES = wavefronts per workgroup (16 at MB = 1,
// 8 above: 32 MB fp32 activation registers per lane); NS = depth of the weight ring.
//
// One CU has ONE scalar unit: with 16 wavefronts resident every scalar instruction of the kernel costs the CU 16
// issue slots (the first version spent 2300 cycles between its first two stamps on integer divisions and descriptor
// arithmetic, profiles/r2_timeline_stream.txt). Everything that depends only on the problem is therefore computed on
// the host and arrives in preloaded SGPRs; what remains per item is a handful of scalar operations.
template <typename T, int MB, int WAVES, int NS, int FLAGS>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(WAVES / 4, WAVES / 4))) void gemv4_stream_kernel(
    // hot arguments as separate scalars: the command processor preloads them into SGPRs (kernarg preload, 14 dwords),
    // so a wavefront does not start with a dependent s_load from a cold kernarg buffer
    const void* hot_A, const uint8_t* hot_B, const float* hot_absmax, const uint8_t* hot_absmax8, int hot_N, int hot_K,
    int hot_packed /* M | bs_shift << 18 | P << 23 */, int hot_geom /* R | SW << 16 | G << 21 */,
    int hot_inv /* ceil(256 / SW) */, const StreamArgs p) {
    constexpr bool NESTED = FLAGS & kNested, CODEPTR = FLAGS & kCodePtr, NT = FLAGS & kNT, GROUPED = FLAGS & kGrouped;
    constexpr bool MULTI = FLAGS & kMulti;
    constexpr bool PEER = (FLAGS & kPeer) != 0;
    static_assert(!PEER || (MB == 1 && !MULTI && !GROUPED && WAVES == 16), "the peer-chain form is the M = 1, single-phase kernel");
    constexpr int THREADS = WAVES * 64;
    constexpr int TB = TypeInfo<T>::bytes;
    constexpr int CH = 2 * TB;  // 16-byte chunks of activations per lane and segment (= 1-KiB DMA pieces per segment)
    constexpr int EPC = 16 / TB; // elements per chunk
    constexpr int LPS = 2; // vector loads per ring stage (weights + scale / 8-bit scale code)
    // The wavefronts of a workgroup start ~90 cycles apart (the last of 16 about 1400 cycles after the first, measured
    // with s_memtime): everything that has to be finished before the first barrier - activation copy, decode table -
    // is given to the first half of them, which have that time to spare; the late ones only issue their loads.
    constexpr int BUILDERS = (WAVES >= 16) ? WAVES / 2 : WAVES; // (an 8-wavefront workgroup starts within ~130 cycles)
    constexpr int IL = (NS % 2 == 0 && WAVES < 16 && MB <= 2) ? 2 : 1; // ring stages decoded together
    static_assert(NS % IL == 0, "the ring is consumed IL stages at a time");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = hot_K;
    // Single-phase instances have no loop around the body: in the looped form LLVM hoists every phase-invariant
    // address computation in front of the loop - i.e. in front of the first loads.
    const int M = hot_packed & 0x3FFFF, bs_shift = (hot_packed >> 18) & 31, P = MULTI ? (hot_packed >> 23) & 511 : 1;
    const int R = hot_geom & 0xFFFF, SW = (hot_geom >> 16) & 31, G = (hot_geom >> 21) & 31;
    const int S = (K + kSegK - 1) >> 11;
    const int rows_total = GROUPED ? p.rows_total : hot_N;
    const int row_begin = blockIdx.x * R;
    if (row_begin >= rows_total)
        return;
    const int nrows = (rows_total - row_begin < R) ? rows_total - row_begin : R;
    const int m0 = blockIdx.y * MB;
    // wave -> (segment column sw, row group g): g = wave / SW by a host-made reciprocal (exact for wave < 16 <= 256 / SW)
    const int g = (wave * hot_inv) >> 8;
    const int sw = wave - g * SW;
#ifdef BNB_PROFILING
    if (p.dbg && lane == 0)
        BNB_ST_DBG_BASE[(static_cast<long>(blockIdx.x) * WAVES + wave) * 16 + 13] = __builtin_amdgcn_s_memrealtime();
#endif
    BNB_ST_STAMP(0)

    // LDS map: table | nested code(s) (1 KiB per matrix) | activation image [MB][SW][2048] T | segment partials [R][S][MB]
    // (the image sits at a compile-time offset: its address is needed before the first loads go out)
    constexpr int kCode2Bytes = (GROUPED ? kMaxGroup : 1) * 1024 + 64; // + the matrices' absmax offsets (grouped launches)
    float* const code2 = reinterpret_cast<float*>(smem + kLutBytes);
    unsigned char* const ximg = smem + kLutBytes + kCode2Bytes;
    float* const part = reinterpret_cast<float*>(ximg + MB * SW * kSegK * TB);

    const T* __restrict__ A = static_cast<const T*>(hot_A);

    // ---- loads that must not sit behind the weight stream: oldest in the queue
    // (nested: entry tid & 255 of the 256-entry absmax code of matrix tid >> 8, + THREADS / 256 matrices per pass)
    constexpr int C2PASS = (kMaxGroup * 256 + THREADS - 1) / THREADS;
    float code2_v[NESTED ? C2PASS : 1];
    float cv = 0.0f;
    const int nmat = GROUPED ? p.nmat : 1;
    float offset = 0.0f;
    if constexpr (NESTED) {
#pragma unroll
        for (int i = 0; i < C2PASS; ++i) {
            const int t = i * THREADS + tid;
            code2_v[i] = (t < nmat * 256) ? p.mat[GROUPED ? (t >> 8) : 0].absmax_code[t & 255] : 0.0f;
        }
        // the absmax offset(s): requested HERE, in front of the weight stream. Fetched after the ring (as this kernel did
        // until round 3) the load is the youngest of the queue, and the wait for it in front of the first decode is a wait
        // for every ring stage: the nested configurations ran 0.5 us behind the fp32-absmax ones
        offset = p.mat[GROUPED ? (tid < nmat ? tid : 0) : 0].absmax_offset[0];
    }
    float* const offs = reinterpret_cast<float*>(smem + kLutBytes + kCode2Bytes - 64);
    if constexpr (CODEPTR)
        cv = p.code16[lane & 15];

    // ---- activation image of phase ph: LDS-DMA, one 1-KiB piece per instruction. The image is lane-linear per
    // piece (hardware), so the bank swizzle is applied on the SOURCE side: slot s = CH l' + j of a segment row
    // holds chunk CH l' + (j ^ f(l')) - a permutation inside 64- / 128-byte groups, the copy stays fully coalesced -
    // and lane l finds its q-th chunk at slot CH l + (q ^ f(l)): conflict-free ds_read_b128 (f(l) = (l >> 2) & 3 for
    // 16-bit activations as in round 1's kernel, (l >> 1) & 7 for fp32). Spelled in asm: with the builtin the
    // compiler sees DMA and ordinary loads on one counter and turns every later register wait into vmcnt(0).
    auto swz = [](int l) -> int { return (CH == 4) ? ((l >> 2) & 3) : ((l >> 1) & 7); };
    auto issue_x = [&](int ph) {
        if (wave >= BUILDERS)
            return;
        const int segs = (S - ph * SW < SW) ? S - ph * SW : SW;
        const int pieces = MB * segs * CH;
        for (int piece = wave; piece < pieces; piece += BUILDERS) {
            const int t = piece / CH, pc = piece - t * CH; // CH is a power of two
            const int m = (MB == 1) ? 0 : t / segs;
            const int sg = t - m * segs;
            const int s = pc * 64 + lane;
            const int lp = s / CH, j = s - lp * CH;
            const int kl = (CH * lp + (j ^ swz(lp))) * EPC;
            const int k = (ph * SW + sg) * kSegK + kl;
            const int mr = (m0 + m < M) ? m0 + m : M - 1;
            const T* src = A + static_cast<long>(mr) * K + ((k < K) ? k : 0);
            const uint32_t dst = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_ptr)ximg)) +
                                 static_cast<uint32_t>(((m * SW + sg) * CH + pc) * 1024);
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(dst) : "memory", "m0");
        }
    };

    // ---- weight ring. Loads go through buffer descriptors so that EVERY issue is unconditional: an item past the
    // end of the wavefront's list (and a lane past the end of the row) is an out-of-range access - the hardware
    // bounds check returns zeros without touching memory - and the number of vector-memory operations in flight is
    // the same on every path. That is what lets the compiler's counted waits (vmcnt((NS-1) * LPS) before stage j is
    // consumed) stay exact from the first item to the last, whatever the item count; with branches around the
    // loads its wait insertion merges the paths and falls back to draining the queue.
    struct Stage {
        u32x4 w;
        float s;      // fp32 absmax of the lane's block (nested: the uint8 code in the low byte)
        float s2a, s2b; // nested: second-level absmax of the item's first / last block (wave-uniform: SCALAR loads)
        uint32_t grp;   // nested: second-level group (block >> 8) of the item's first block
    };
    Stage st[NS];
    constexpr uint32_t kOob = 0xFFFFFFF0u; // beyond num_records
    constexpr int kRsrcFlags = 0x00020000;
    constexpr int kRecords = 0x7FFFFFFF;   // the descriptors only exist for the out-of-range trick: valid offsets are < 2^30
    constexpr int kWeightAux = NT ? 2 : 0; // nt: streamed once, read by one CU

    // matrix of a concatenated row (grouped launches): wave-uniform scan over <= kMaxGroup descriptors
    auto mat_of = [&](int grow) -> int {
        int mi = 0;
        if constexpr (GROUPED) {
#pragma unroll
            for (int i = 1; i < kMaxGroup; ++i)
                mi = (i < p.nmat && grow >= p.mat[i].row_start) ? i : mi;
        }
        return mi;
    };

    int seg = sw;
    int rl_end = 0;       // items of this wavefront: local rows g, g + G, ... < rl_end (0: nothing to do in this phase)
    uint32_t k0 = 0;      // first k of this lane in the current segment
    bool k_ok = false;    // ... and whether it lies inside the row

    // Offsets of an item's loads: a per-phase lane part (k0, or kOob for lanes past the end of the row) OR-ed with a
    // wave-uniform mask that is all-ones-ish for an item past the end of the list - branch-free (written as selects,
    // hipcc turned the predicate into two exec-masked copies of every load).
    uint32_t lane_mask = kOob; // 0 for lanes inside the row, kOob otherwise
    auto issue = [&](Stage& s, int i) {
        const bool valid = g + i * G < rl_end; // wave-uniform
        const int grow = valid ? row_begin + g + i * G : 0;
        const uint32_t inval = (valid ? 0u : kOob) | lane_mask;
        const uint8_t* Bp = hot_B;
        const float* am = hot_absmax;
        const uint8_t* am8 = hot_absmax8;
        int row = grow;
        if constexpr (GROUPED) {
            const int mi = mat_of(grow);
            Bp = p.mat[mi].B;
            am = p.mat[mi].absmax;
            am8 = p.mat[mi].absmax8;
            row = valid ? grow - p.mat[mi].row_start : 0;
        }
        const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(Bp), 0, kRecords, kRsrcFlags);
        s.w = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                            rs_w, (k0 >> 1) | inval,
                                            static_cast<uint32_t>(row) * static_cast<uint32_t>(K >> 1), kWeightAux));
        const uint32_t blk = (static_cast<uint32_t>(row) * static_cast<uint32_t>(K) + k0) >> bs_shift;
        if constexpr (NESTED) {
            const auto rs_q = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(am8), 0, kRecords, kRsrcFlags);
            s.s = __builtin_bit_cast(float, static_cast<uint32_t>(__builtin_amdgcn_raw_buffer_load_b8(rs_q, blk | inval, 0, 0)));
            // second level: the <= 2048 / bs blocks of an item lie in at most TWO groups of 256 blocks (in ONE when K is a
            // multiple of 2048): two scalar loads per item instead of a third vector-memory instruction per ring stage - the
            // nested configurations ran 0.5 us behind the fp32-absmax ones on 4096^2 (profiles/r3_configs_bench_mid_round.txt)
            const uint32_t e0 = static_cast<uint32_t>(row) * static_cast<uint32_t>(K) + static_cast<uint32_t>(seg * kSegK);
            uint32_t k_last = static_cast<uint32_t>(seg * kSegK + kSegK - 1);
            k_last = k_last < static_cast<uint32_t>(K) ? k_last : static_cast<uint32_t>(K - 1);
            const uint32_t e1 = static_cast<uint32_t>(row) * static_cast<uint32_t>(K) + k_last;
            const uint32_t ga = valid && seg < S ? (e0 >> bs_shift) >> 8 : 0u, gb = valid && seg < S ? (e1 >> bs_shift) >> 8 : 0u;
            s.grp = ga;
            // (constant address space: the statistics are read-only for the kernel, a uniform index then compiles to s_load_dword;
            // through the generic pointer hipcc emits a vector load per lane)
            typedef const float __attribute__((address_space(4))) * cfloat_ptr;
            const cfloat_ptr amc = (cfloat_ptr)(reinterpret_cast<uintptr_t>(am));
            s.s2a = amc[__builtin_amdgcn_readfirstlane(ga)];
            s.s2b = amc[__builtin_amdgcn_readfirstlane(gb)];
        } else {
            const auto rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(am), 0, kRecords, kRsrcFlags);
            s.s = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_a, (blk * 4u) | inval, 0, 0));
        }
    };

    uint32_t perm_sel = 0x0C0C0400u; // v_perm_b32 selector {lane offset, weight byte j, 0, 0}
    const uint32_t lane_off = static_cast<uint32_t>(lane & 31) * 8u;
    // the lane's 32 activations of each row as 16 fp32 PAIRS (k, k + 1): a packed byte decodes to the pair (code[hi],
    // code[lo]) and one v_pk_fma_f32 multiplies pair by pair - half the VALU issue slots of two v_fma_f32 (the decode is
    // VALU-bound at streaming rate: ~80 wave-instructions per KiB of weights against ~1 per cycle and CU)
    f32x2 xr[MB][16];

    auto load_slice = [&]() {
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            float xs[32];
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                const int slot = CH * lane + (q ^ swz(lane));
                const u32x4 v = *reinterpret_cast<const u32x4*>(ximg + ((m * SW + sw) * CH * 64 + slot) * 16);
                unpack16<T>(v, &xs[q * EPC]);
            }
            // lanes past the end of the row hold zeros (their weight loads are out of range, their scale is forced to 0);
            // only the last segment of a row whose length is not a multiple of 2048 has such lanes
            if ((seg + 1) * kSegK > K) {
#pragma unroll
                for (int e = 0; e < 32; ++e)
                    xs[e] = k_ok ? xs[e] : 0.0f;
            }
#pragma unroll
            for (int b = 0; b < 16; ++b)
                xr[m][b] = f32x2{xs[2 * b], xs[2 * b + 1]};
        }
    };

    // Decode of IL consecutive ring stages at once (IL = 1 with 16 wavefronts per CU, 2 with 8: half as many
    // wavefronts repeat the per-wavefront work - prologue, table, slice - but each needs its own instruction-level
    // parallelism): all look-ups of the IL items first (one v_perm_b32 + one ds_read_b64 per packed byte), then the
    // FMAs, then IL x MB wave reductions in lock step. The LDS round trip is paid once per group, not once per byte
    // (left alone, the scheduler serialises look-up and use).
    auto compute = [&](int j0, int i0) {
        f32x2 pr[IL][16];
        // nested absmax: the second-level look-up (byte code -> code2 entry) goes out FIRST, in front of the sixteen table
        // look-ups of the item, so that its LDS round trip runs beside theirs instead of behind the FMAs (config 5 was 0.4 us
        // slower than config 2 with fewer bytes: profiles/r2_configs_bench.txt)
        float c2v[IL];
        if constexpr (NESTED) {
#pragma unroll
            for (int u = 0; u < IL; ++u) {
                const uint32_t q8 = __builtin_bit_cast(uint32_t, st[j0 + u].s);
                if constexpr (GROUPED)
                    c2v[u] = code2[mat_of(row_begin + g + (i0 + u) * G) * 256 + q8];
                else
                    c2v[u] = code2[q8];
            }
        }
#pragma unroll
        for (int u = 0; u < IL; ++u)
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t addr = __builtin_amdgcn_perm(st[j0 + u].w[d], lane_off, perm_sel + (j << 8));
                    pr[u][4 * d + j] = *reinterpret_cast<const __attribute__((address_space(3))) f32x2*>(addr);
                }
        __builtin_amdgcn_sched_barrier(0);
        // the block's scale. 16-bit activations: applied once to the lane's fp32 sum of 32 products (exact codes, one extra
        // rounding per 32 terms - well inside the reference's envelope for those types). fp32 activations: applied to the
        // decoded pair BEFORE the FMA - the operand is fl(code * scale), the reference's own arithmetic (dequantize, then a
        // fp32 product: csrc/kernels.cu gemv, default/ops.py) - because its fp32 envelope (tests/test_functional.py:892-895:
        // 1e-8 + 7 * 2e-9 per element and sqrt(dim)) is tighter than the post-scaled sum's rounding (measured 3e-8 on the
        // fc2 shapes)
        constexpr bool PRESCALE = TB == 4;
        float scale_u[IL];
#pragma unroll
        for (int u = 0; u < IL; ++u) {
            const Stage& s = st[j0 + u];
            float scale;
            if constexpr (NESTED) {
                // the lane's second-level group against the item's first: which of the two scalars is this lane's
                int rowl = row_begin + g + (i0 + u) * G;
                int mi = 0;
                if constexpr (GROUPED) {
                    mi = mat_of(rowl);
                    rowl -= p.mat[mi].row_start;
                }
                const uint32_t lane_grp = ((static_cast<uint32_t>(rowl) * static_cast<uint32_t>(K) + k0) >> bs_shift) >> 8;
                const float s2 = lane_grp == s.grp ? s.s2a : s.s2b;
                if constexpr (GROUPED)
                    scale = nested_scale(c2v[u], s2, offs[mi]);
                else
                    scale = nested_scale(c2v[u], s2, offset);
            } else {
                scale = s.s;
            }
            scale_u[u] = k_ok ? scale : 0.0f;
            if constexpr (PRESCALE) {
                const f32x2 sc2 = {scale_u[u], scale_u[u]};
#pragma unroll
                for (int b = 0; b < 16; ++b)
                    pr[u][b] = pr[u][b] * sc2;
            }
        }
        // two independent chains of packed FMAs per (item, row): [chain][even k, odd k] - the same four partial sums, in
        // the same order, as four scalar chains
        f32x2 acc[IL][MB][2];
#pragma unroll
        for (int u = 0; u < IL; ++u)
#pragma unroll
            for (int m = 0; m < MB; ++m)
                acc[u][m][0] = acc[u][m][1] = f32x2{0.0f, 0.0f};
#pragma unroll
        for (int b = 0; b < 16; ++b)
#pragma unroll
            for (int u = 0; u < IL; ++u)
#pragma unroll
                for (int m = 0; m < MB; ++m)
                    acc[u][m][b & 1] = __builtin_elementwise_fma(pr[u][b], xr[m][b], acc[u][m][b & 1]);
        float v[IL * MB];
#pragma unroll
        for (int u = 0; u < IL; ++u) {
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                const float sum = (acc[u][m][0][0] + acc[u][m][0][1]) + (acc[u][m][1][0] + acc[u][m][1][1]);
                v[u * MB + m] = PRESCALE ? sum : sum * scale_u[u];
            }
        }
        wave_sum_n<IL * MB>(v);
#pragma unroll
        for (int u = 0; u < IL; ++u) {
            const int rl = g + (i0 + u) * G;
            if (lane == 0 && rl < rl_end) {
#pragma unroll
                for (int m = 0; m < MB; ++m)
                    part[(rl * S + seg) * MB + m] = v[u * MB + m];
            }
        }
    };

    uint32_t epoch = 0;     // (peer chain) the exchange this launch consumes
    uint32_t epoch_raw = 0; // ... the epoch word as loaded
    for (int ph = 0; ph < P; ++ph) {
        if (ph > 0)
            __syncthreads(); // everyone is done with the previous activation image
        // (1) this phase's activation image (LDS-DMA, oldest in the queue), then the first NS ring stages
        if (ph == 0)
            BNB_ST_STAMP(9)
        // (peer chain) x = the current exchange: 16 bytes = two granules = four values per lane and round, fetched by wavefronts
        // 0 ... 7 (they start first; the decode table is built by 8 ... 15 in this mode), IN FRONT of their weight ring - so that tags can be checked and the image written while the ring is still in flight (behind the
        // ring, the in-order counter made x wait for every weight byte: +1.2 us per layer, profiles/r4_peer_chain.txt). The epoch
        // they depend on is one scalar load; a lane past the end of x is out of range (zeros, no traffic).
        bool x_from_peer = false;
        u32x4 gx[PEER ? kChainRounds : 1];
        if constexpr (PEER) {
            // (the mode travels in the spare bits of a PRELOADED argument and, when x comes from the exchange, the address of the
            // exchange's region in the slot of the unused activation pointer - the region of an exchange is its POSITION in the chain,
            // which the host knows, only its tag carries the device-side epoch: the fetch depends on no load at all)
            const int peer_mode = (hot_packed >> 24) & 15;
            x_from_peer = (peer_mode & 1) != 0;
            if (x_from_peer && wave < WAVES - BUILDERS) {
                const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(hot_A), 0, K * 4, kRsrcFlags);
#pragma unroll
                for (int r = 0; r < kChainRounds; ++r)
                    gx[r] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                          rs_x, static_cast<uint32_t>((r * (WAVES - BUILDERS) + wave) * 64 + lane) * 16u, 0, 17 /* sc0 sc1 */));
            }
        }
        if (!x_from_peer)
            issue_x(ph);
        if (ph == 0)
            BNB_ST_STAMP(10)
        seg = ph * SW + sw;
        rl_end = (g < G && seg < S) ? nrows : 0;
        k0 = static_cast<uint32_t>(seg * kSegK + lane * 32);
        k_ok = (seg < S) && (k0 < static_cast<uint32_t>(K));
        lane_mask = k_ok ? 0u : kOob;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            issue(st[j], j);
            __builtin_amdgcn_sched_barrier(0); // keep the queue in stage order: (weights, scale) of stage 0, of stage 1, ...
        }
        if constexpr (PEER) {
            // the epoch word (the tags of the exchange consumed here are epoch + offset, of the one produced epoch + offset + 1):
            // requested BEHIND every load of the prologue - its address comes from the kernarg segment, whose first access is a cold
            // miss that stalled every wavefront for ~500 cycles in front of its ring when it sat there (profiles/r4_timeline_chain.txt)
            typedef const uint32_t __attribute__((address_space(4))) * cu32_ptr;
            epoch_raw = *(cu32_ptr)(reinterpret_cast<uintptr_t>(p.peer.epoch_word));
            epoch = epoch_raw + p.peer.epoch_offset;
        }
        if (ph == 0)
            BNB_ST_STAMP(1)
        if (ph == 0) {
            // everything below is written AFTER the first loads in program order and fenced there: the scalar unit
            // is shared by the CU's 16 wavefronts, so nothing that is not needed for the loads may run before them
            __builtin_amdgcn_sched_barrier(0);
        }
        // (peer chain, x from the exchange: the wavefronts that start FIRST fetch x - it is the longest pole in front of the barrier -
        // and the table is built by the second half of the workgroup, tid_b = the builder's index among the builders)
        const bool table_builder = (PEER && x_from_peer) ? (wave >= WAVES - BUILDERS) : (wave < BUILDERS);
        const int tid_b = (PEER && x_from_peer) ? tid - (WAVES - BUILDERS) * 64 : tid;
        if (ph == 0 && table_builder) {
            if constexpr (!CODEPTR)
                cv = code_literal<(FLAGS & kFp4) != 0>((lane & 15) + opaque_zero()); // (opaque: not hoisted above the loads)
            // (2) decode table, built while the loads fly: entry e (a packed byte) = 32 copies of
            // (code[e >> 4], code[e & 15]) in fp32, 256 B per entry, copy c at byte 8 c. Chunk c16 of the table
            // (16 B = two copies) is written by thread c16 % THREADS: every ds_write_b128 of a wavefront covers 1 KiB
            // contiguous, conflict-free. The two code values come from lanes (e >> 4) and (e & 15) of `cv`.
            constexpr int BT = BUILDERS * 64;
            constexpr int ITERS = kLutBytes / 16 / BT;
            static_assert((kLutBytes / 16) % BT == 0 && BT % 256 == 0, "whole passes over the table");
            // chunk c16 = it * BT + tid: entry e = c16 >> 4, so code[e & 15] is the same in every pass of a thread and
            // code[e >> 4] = code[c16 >> 8]. All cross-lane fetches are issued before the first store: written as
            // fetch-fetch-store per pass the build is a chain of LDS round trips (measured: 2400 cycles for 16 passes).
            const int cvb = __builtin_bit_cast(int, cv);
            const float lo = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((tid_b >> 4) & 15) * 4, cvb));
            float hi[ITERS];
#pragma unroll
            for (int it = 0; it < ITERS; ++it)
                hi[it] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((it * BT + tid_b) >> 8) * 4, cvb));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int it = 0; it < ITERS; ++it)
                *reinterpret_cast<f32x4*>(smem + (it * BT + tid_b) * 16) = f32x4{hi[it], lo, hi[it], lo};
        }
        if (ph == 0) {
            if constexpr (NESTED) {
#pragma unroll
                for (int i = 0; i < C2PASS; ++i) {
                    const int t = i * THREADS + tid;
                    if (t < nmat * 256)
                        code2[t] = code2_v[i];
                }
                if constexpr (GROUPED)
                    if (tid < nmat)
                        offs[tid] = offset;
            }
        }
        // (3) the activation DMAs are older than the NS ring stages: wait until only those remain in flight
        if (ph == 0)
            BNB_ST_STAMP(2)
        if (!x_from_peer)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS * LPS) : "memory");
        if (ph == 0)
            BNB_ST_STAMP(11) // (this wavefront's x pieces have landed - or it had none; the barrier behind it waits for everyone's)
        if constexpr (PEER) {
            if (x_from_peer && wave < WAVES - BUILDERS) {
                const unsigned char* const src = static_cast<const unsigned char*>(hot_A);
                uint32_t wait_bound = p.peer.spin_bound; // (per lane; drops to a few polls once anything on this buffer timed out)
                // one fetched vector = two granules {pair, tag, pair2, tag}: re-fetched (system scope, bounded) until both tags are there
                auto settle = [&](u32x4 gr, uint32_t off) -> u32x4 {
                    // (the re-fetch loop is spelled in asm: a loop with loads in it makes the compiler's wait insertion
                    // forget what is in flight behind it. Slow path only - a peer that is late.)
                    uint32_t spins = 0;
                    while (gr[1] != epoch || gr[3] != epoch) {
                        if (spins == 0 && wait_bound == p.peer.spin_bound) {
                            // first miss of this lane: has a wait on this buffer ALREADY run into its bound (an earlier round of
                            // this launch, or an earlier launch - the word is sticky)? Then the peer is gone, and paying the full
                            // bound again per round and per layer would block the queue for the length of the chain before any
                            // host-side check() runs: a few polls each from here on.
                            uint32_t st_word;
                            const uint32_t* const st_addr = reinterpret_cast<const uint32_t*>(p.peer.local) + 1;
                            asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(st_word) : "v"(st_addr) : "memory");
                            if (st_word != 0u)
                                wait_bound = kPeerPollsAfterTimeout;
                        }
                        __builtin_amdgcn_s_sleep(4);
                        const unsigned char* const addr = src + off;
                        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(gr) : "v"(addr) : "memory");
                        if (++spins > wait_bound) {
                            __hip_atomic_store(reinterpret_cast<uint32_t*>(p.peer.local) + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            gr = u32x4{0xFFFFFFFFu, epoch, 0xFFFFFFFFu, epoch}; // NaN in fp16 / bf16: a timeout cannot pass for data
                            wait_bound = kPeerPollsAfterTimeout;
                        }
                    }
                    return gr;
                };
                // four values = half of the 16-byte chunk c of x; chunk (l', q) of a segment lives at slot CH l' + (q ^ swz(l'))
                using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;
                auto put4 = [&](uint32_t j4, u32x2 four) {
                    const uint32_t c = j4 >> 1, half = j4 & 1u;
                    const uint32_t sg = c >> 8, cs = c & 255u, lp = cs / CH, q = cs % CH;
                    *reinterpret_cast<u32x2*>(ximg + ((sg * CH * 64 + CH * lp + (q ^ static_cast<uint32_t>(swz(static_cast<int>(lp))))) * 16 + half * 8)) = four;
                };
#pragma unroll
                for (int r = 0; r < kChainRounds; ++r) {
                    const uint32_t off = static_cast<uint32_t>((r * (WAVES - BUILDERS) + wave) * 64 + lane) * 16u;
                    if (off < static_cast<uint32_t>(K) * 4u) {
                        const u32x4 gr = settle(gx[r], off);
                        put4(off >> 4, u32x2{gr[0], gr[2]});
                    }
                }
            }
        }
        __syncthreads();
        if (ph == 0)
            BNB_ST_STAMP(3)
        // an opaque zero ties the decode to program order after the barrier (without it LLVM hoists the first
        // look-up address - and its vmcnt wait - above the table build)
        perm_sel = 0x0C0C0400u + static_cast<uint32_t>(opaque_zero());
        load_slice();
        if (ph == 0)
            BNB_ST_STAMP(4)
#ifdef BNB_PROFILING
        if (ph == 0 && p.dbg && !(reinterpret_cast<uintptr_t>(p.dbg) & 1)) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS * LPS - 1) : "memory");
            BNB_ST_STAMP(15)
        }
#endif
        // (4) rounds of NS items, IL at a time. While another round follows, the refill of a stage is issued right
        // after the stage was consumed - valid or not, see above; the last round (peeled) consumes without refilling.
        int base = 0;
        for (; g + (base + NS) * G < rl_end; base += NS) {
#pragma unroll
            for (int j = 0; j < NS; j += IL) {
                compute(j, base + j);
                if (ph == 0 && base == 0 && j == 0)
                    BNB_ST_STAMP(5)
#pragma unroll
                for (int u = 0; u < IL; ++u)
                    issue(st[j + u], base + j + u + NS);
            }
        }
#pragma unroll
        for (int j = 0; j < NS; j += IL) {
            if (g + (base + j) * G < rl_end)
                compute(j, base + j);
            if (ph == 0 && base == 0 && j == 0)
                BNB_ST_STAMP(5)
        }
    }
    BNB_ST_STAMP(6)
    // The table is addressed with the raw v_perm_b32 result: it must sit at LDS address 0 (this kernel has no static
    // LDS, so the dynamic segment starts there).
    if (reinterpret_cast<uintptr_t>((lds_ptr)smem) != 0)
        __builtin_trap();

    // ---- combine the segment partials of every row in segment order, bias, one rounding
    [[maybe_unused]] uint32_t epoch_out = 0;
    if constexpr (PEER)
        epoch_out = epoch + 1u;
    __syncthreads();
    BNB_ST_STAMP(7)
    for (int idx = tid; idx < nrows * MB; idx += THREADS) {
        const int m = (MB == 1) ? 0 : idx / nrows, rl = idx - m * nrows;
        if (m0 + m >= M)
            continue;
        float v = 0.0f;
        for (int sg = 0; sg < S; ++sg)
            v += part[(rl * S + sg) * MB + m];
        const int grow = row_begin + rl;
        int mi = 0, row = grow;
        if constexpr (GROUPED) {
            mi = mat_of(grow);
            row = grow - p.mat[mi].row_start;
        }
        const T* bias = static_cast<const T*>(p.mat[mi].bias);
        const float b = bias ? static_cast<float>(bias[row]) : 0.0f;
        const T tv = static_cast<T>(v + b);
        if constexpr (PEER) {
            if (p.mat[mi].out != nullptr)
                static_cast<T*>(p.mat[mi].out)[row] = tv;
            if (p.peer.mode & 2) {
                // (rows come in pairs: the host keeps ns and the rows per workgroup even, so a thread and its neighbour are in the
                // loop together and the even one stores both values)
                const uint32_t own = static_cast<uint32_t>(__builtin_bit_cast(unsigned short, tv));
                const uint32_t other = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute((lane ^ 1) << 2, static_cast<int>(own)));
                const uint32_t pair = own | (other << 16);                                                           // (even lanes: rows r, r + 1)
                const uint32_t pair2 = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute((lane ^ 2) << 2, static_cast<int>(pair))); // rows r + 2, r + 3
                const size_t slot = kChainDataOffset + (static_cast<size_t>((p.peer.epoch_offset + 1u) & (kChainRegions - 1u)) * p.peer.max_granules +
                                                        ((static_cast<size_t>(p.peer.rank) * static_cast<size_t>(rows_total) + static_cast<size_t>(row)) >> 1)) * 8u;
                if (p.peer.mode & 8) {
                    // gated production: this thread's row and its neighbour's are (gate, up) of activation row / 2. silu in fp32
                    // ( g / (1 + exp(-g)), exact expf and IEEE division: torch's own kernel ), rounded to T; the product in fp32,
                    // rounded to T - bit for bit what `F.silu(g) * u` gives on the T-valued outputs of the two unsharded layers
                    // (tests/checks/peer_ranks.py sweeps every finite 16-bit pattern of g). Rows come in fours (host-enforced): the
                    // thread of row 4 i stores ONE granule = activations 2 i, 2 i + 1 of this rank.
                    const float gf = static_cast<float>(tv), uf = static_cast<float>(__builtin_bit_cast(T, static_cast<unsigned short>(other)));
                    const T st = static_cast<T>(gf / (1.0f + expf(-gf)));
                    const T at = static_cast<T>(__fmul_rn(static_cast<float>(st), uf));
                    const uint32_t abits = static_cast<uint32_t>(__builtin_bit_cast(unsigned short, at));
                    const uint32_t anext = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute((lane ^ 2) << 2, static_cast<int>(abits)));
                    if (!(row & 3)) {
                        using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;
                        const u32x2 granule = {abits | (anext << 16), epoch_out};
                        const size_t aslot = kChainDataOffset + (static_cast<size_t>((p.peer.epoch_offset + 1u) & (kChainRegions - 1u)) * p.peer.max_granules +
                                                                 ((static_cast<size_t>(p.peer.rank) * static_cast<size_t>(rows_total >> 1) + static_cast<size_t>(row >> 1)) >> 1)) * 8u;
                        for (int pr = 0; pr < p.peer.world; ++pr) {
                            unsigned char* const dst = p.peer.base[pr] + aslot;
                            if (pr == p.peer.rank)
                                asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(dst), "v"(granule) : "memory");
                            else
                                asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(dst), "v"(granule) : "memory");
                        }
                    }
                } else if (p.peer.mode & 4) {
                    // two granules per store: a 16-byte system-scope store costs the fabric what an 8-byte one does (each aligned
                    // 8-byte half carries its own tag, so the two need not land together)
                    if (!(row & 3)) {
                        const u32x4 granules = {pair, epoch_out, pair2, epoch_out};
                        for (int pr = 0; pr < p.peer.world; ++pr) {
                            unsigned char* const dst = p.peer.base[pr] + slot;
                            // (this rank's OWN copy is consumed by a later launch of this device: the kernel boundary makes an
                            // ordinary store visible, and the launch does not end behind a write-through acknowledgement for it)
                            if (pr == p.peer.rank)
                                asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(dst), "v"(granules) : "memory");
                            else
                                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(granules) : "memory");
                        }
                    }
                } else if (!(row & 1)) {
                    using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;
                    const u32x2 granule = {pair, epoch_out};
                    for (int pr = 0; pr < p.peer.world; ++pr) {
                        unsigned char* const dst = p.peer.base[pr] + slot;
                        if (pr == p.peer.rank)
                            asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(dst), "v"(granule) : "memory");
                        else
                            asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(dst), "v"(granule) : "memory");
                    }
                }
            }
        } else {
            static_cast<T*>(p.mat[mi].out)[static_cast<long>(m0 + m) * p.mat[mi].N + row] = tv;
        }
    }
    BNB_ST_STAMP(8)
#ifdef BNB_PROFILING
    if (p.dbg && lane == 0)
        BNB_ST_DBG_BASE[(static_cast<long>(blockIdx.x) * WAVES + wave) * 16 + 14] = __builtin_amdgcn_s_memrealtime();
#endif
}

// ---------------------------------------------------------------------------------------------
// Generic fallback: any K (odd, not a multiple of 32), any pointer alignment. One wavefront per output row, scalar
// byte loads, fp32 math, 16-entry fp32 table in LDS. Correctness path only.
// ---------------------------------------------------------------------------------------------
struct GenericArgs {
    const void* A;
    const uint8_t* B;
    const float* absmax;
    const uint8_t* absmax8;
    const float* absmax_code;
    const float* absmax_offset;
    const float* code16;
    void* out;
    const void* bias;
    int M, N, K, bs_shift, quant_type;
};

template <typename T, bool NESTED> __global__ __launch_bounds__(256) void gemv4_generic_kernel(const GenericArgs p) {
    __shared__ float code[16];
    __shared__ float code2[NESTED ? 256 : 1];
    const int tid = threadIdx.x;
    if (tid < 16)
        code[tid] = p.code16 ? p.code16[tid] : (p.quant_type == kNF4 ? kNF4Code[tid] : kFP4Code[tid]);
    if constexpr (NESTED)
        code2[tid] = p.absmax_code[tid];
    __syncthreads();
    const int lane = tid & 63;
    const int row = blockIdx.x * 4 + (tid >> 6);
    const int m = blockIdx.y;
    if (row >= p.N)
        return;
    const T* __restrict__ A = static_cast<const T*>(p.A) + static_cast<long>(m) * p.K;
    const long base = static_cast<long>(row) * p.K;
    float acc = 0.0f;
    float run = 0.0f; // partial sum inside the current quantization block
    long run_blk = -1;
    auto block_scale = [&](long blk) -> float {
        if constexpr (NESTED)
            return nested_scale(code2[p.absmax8[blk]], p.absmax[blk >> 8], p.absmax_offset[0]);
        else
            return p.absmax[blk];
    };
    for (int k = lane; k < p.K; k += 64) {
        const long e = base + k;
        const uint8_t byte = p.B[e >> 1];
        const int nib = (e & 1) ? (byte & 0xF) : (byte >> 4);
        const long blk = e >> p.bs_shift;
        if (blk != run_blk) {
            if (run_blk >= 0)
                acc = fmaf(block_scale(run_blk), run, acc);
            run = 0.0f;
            run_blk = blk;
        }
        run = fmaf(static_cast<float>(A[k]), code[nib], run);
    }
    if (run_blk >= 0)
        acc = fmaf(block_scale(run_blk), run, acc);
    acc = wave_sum(acc);
    if (lane == 0) {
        const float b = p.bias ? static_cast<float>(static_cast<const T*>(p.bias)[row]) : 0.0f;
        static_cast<T*>(p.out)[static_cast<long>(m) * p.N + row] = static_cast<T>(acc + b);
    }
}

// ---------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------
int device_cu_count() { return device_cu_count_or_default(); }

constexpr size_t kLdsBudget = 156 * 1024; // largest dynamic allocation that launches (157 KiB is refused)

struct Geometry {
    int R, SW, G, P, grid_x;
    size_t lds;
};

// Launch geometry: a pure function of the problem and the template's wavefront count.
//   SW  2048-k segments handled side by side (one wavefront column each), bounded by the wavefronts and by the LDS
//       left for the activation image; P = ceil(S / SW) phases re-use the workgroup for longer rows;
//   G   row groups = WAVES / SW; wavefront (sw, g) walks rows g, g + G, ... of the workgroup;
//   R   rows per workgroup: one workgroup per CU unless the partial-sum slots of that many rows do not fit.
Geometry make_geometry(int rows_total, int K, int mb, int waves, int tbytes, bool grouped, int tune_sw, int tune_rows) {
    Geometry ge;
    const int S = (K + kSegK - 1) / kSegK;
    const int cus = device_cu_count();
    int sw = S < waves ? S : waves;
    if (tune_sw > 0 && tune_sw < sw)
        sw = tune_sw;
    const size_t fixed = kLutBytes + (grouped ? kMaxGroup : 1) * 1024 + 64;
    const size_t ximg_cap = 80 * 1024;
    while (sw > 1 && static_cast<size_t>(mb) * sw * kSegK * tbytes > ximg_cap)
        --sw;
    ge.SW = sw;
    ge.G = waves / sw;
    int R = (rows_total + cus - 1) / cus;
    if (tune_rows > 0)
        R = tune_rows;
    const size_t ximg = static_cast<size_t>(mb) * sw * kSegK * tbytes;
    const size_t part_cap = kLdsBudget - fixed - ximg - 16;
    const int r_cap = static_cast<int>(part_cap / (static_cast<size_t>(S) * mb * 4));
    if (R > r_cap)
        R = r_cap;
    if (R > 0xFFFF)
        R = 0xFFFF;
    if (R < 1)
        R = 1;
    ge.R = R;
    ge.P = (S + sw - 1) / sw;
    ge.grid_x = (rows_total + R - 1) / R;
    ge.lds = fixed + ((static_cast<size_t>(R) * S * mb * 4 + 15) & ~size_t(15)) + ximg;
    return ge;
}

struct StreamTuning {
    TlsKnob ns{0}, sw{0}, rows{0}, nt{-1}, waves{0};
};
thread_local StreamTuning g_tune;

// Production ring depth: 2 stages with 16 wavefronts per CU (2 KiB x 16 in flight already cover bandwidth x latency;
// deeper rings only add refill work at the end of a row list), 4 stages - decoded two at a time - with 8.
constexpr int kRing = 2;
constexpr int ring_depth(int mb, int waves) { return (mb == 1 && waves == 8) ? 4 : kRing; }

template <typename T, int MB, int WAVES, int NS, int FLAGS> void launch_one(const StreamArgs& a, hipStream_t stream) {
    const Geometry ge = make_geometry(a.rows_total, a.K, MB, WAVES, TypeInfo<T>::bytes, (FLAGS & kGrouped) != 0, g_tune.sw.load(std::memory_order_relaxed),
                                      g_tune.rows.load(std::memory_order_relaxed));
    dim3 grid(ge.grid_x, (a.M + MB - 1) / MB);
    const StreamMat& m0 = a.mat[0];
    if constexpr (!(FLAGS & kGrouped) && NS == ring_depth(MB, WAVES) && !(MB == 2 && WAVES == 16)) {
        if (ge.P > 1) {
            auto kern = gemv4_stream_kernel<T, MB, WAVES, NS, FLAGS | kMulti>;
            static LdsLimit lds_limit;
            ensure_dynamic_lds(lds_limit, reinterpret_cast<const void*>(kern), ge.lds);
            hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), ge.lds, stream, a.A, m0.B, m0.absmax, m0.absmax8, m0.N, a.K,
                               (a.M & 0x3FFFF) | (a.bs_shift << 18) | (ge.P << 23),
                               ge.R | (ge.SW << 16) | (ge.G << 21), (256 + ge.SW - 1) / ge.SW, a);
            return;
        }
    }
    if (ge.P > 1) {
        // sweep-only instances exist single-phase only; grouped launches are refused earlier (grouped_fits)
        if constexpr (!(FLAGS & kGrouped) && (NS != ring_depth(MB, WAVES) || (MB == 2 && WAVES == 16)))
            return launch_one<T, MB, (MB == 1 ? 16 : 8), kRing, FLAGS | kNT>(a, stream);
        fprintf(stderr, "bitsandbytes_amd: gemv_4bit: internal error, multi-phase geometry on a single-phase instance\n");
        exit(1);
    }
    auto kern = gemv4_stream_kernel<T, MB, WAVES, NS, FLAGS>;
    static LdsLimit lds_limit;
    ensure_dynamic_lds(lds_limit, reinterpret_cast<const void*>(kern), ge.lds);
    hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), ge.lds, stream, a.A, m0.B, m0.absmax, m0.absmax8, m0.N, a.K,
                       (a.M & 0x3FFFF) | (a.bs_shift << 18) | (ge.P << 23),
                       ge.R | (ge.SW << 16) | (ge.G << 21), (256 + ge.SW - 1) / ge.SW, a);
}

// The sweep-only variants (other ring depths, default cache policy) exist for ONE configuration - bf16, one activation
// row, fp32 absmax, literal NF4 table - so that tools/ can A/B them without multiplying the instance count of the library.
template <typename T, int MB, int WAVES, int FLAGS> void launch_tuned(const StreamArgs& a, hipStream_t stream) {
    if constexpr (std::is_same<T, bf16>::value && MB == 1 && FLAGS == 0) {
        const int ns = g_tune.ns.load(std::memory_order_relaxed);
        const int nt = g_tune.nt.load(std::memory_order_relaxed);
        if constexpr (WAVES == 8) {
            if (ns == 2)
                return launch_one<T, 1, 8, 2, kNT>(a, stream);
            if (nt == 0)
                return launch_one<T, 1, 8, 4, 0>(a, stream);
        } else {
            if (ns == 4)
                return launch_one<T, 1, 16, 4, kNT>(a, stream);
            if (ns == 3)
                return launch_one<T, 1, 16, 3, kNT>(a, stream);
            if (ns == 6)
                return launch_one<T, 1, 16, 6, kNT>(a, stream);
            if (nt == 0)
                return launch_one<T, 1, 16, kRing, 0>(a, stream);
        }
    }
    launch_one<T, MB, WAVES, ring_depth(MB, WAVES), FLAGS | kNT>(a, stream);
}

template <typename T, int MB, int WAVES> void launch_flags(const StreamArgs& a, int quant_type, bool grouped, hipStream_t stream) {
    const bool nested = a.mat[0].absmax8 != nullptr;
    const bool fp4 = quant_type == kFP4;
    if (a.code16 != nullptr) {
        // caller-supplied code table (legacy gemv_4bit op): one activation row, un-nested absmax by construction
        if constexpr (MB == 1) {
            if (!nested && !grouped)
                return launch_one<T, 1, WAVES, ring_depth(1, WAVES), kCodePtr | kNT>(a, stream);
        }
        fprintf(stderr, "bitsandbytes_amd: gemv_4bit: a caller-supplied code table needs fp32 absmax\n");
        exit(1);
    }
    const int sel = (nested ? 1 : 0) | (fp4 ? 2 : 0) | (grouped ? 4 : 0);
    switch (sel) {
    case 0: return launch_tuned<T, MB, WAVES, 0>(a, stream);
    case 1: return launch_tuned<T, MB, WAVES, kNested>(a, stream);
    case 2: return launch_tuned<T, MB, WAVES, kFp4>(a, stream);
    case 3: return launch_tuned<T, MB, WAVES, kFp4 | kNested>(a, stream);
    case 4: return launch_tuned<T, MB, WAVES, kGrouped>(a, stream);
    case 5: return launch_tuned<T, MB, WAVES, kGrouped | kNested>(a, stream);
    case 6: return launch_tuned<T, MB, WAVES, kGrouped | kFp4>(a, stream);
    default: return launch_tuned<T, MB, WAVES, kGrouped | kFp4 | kNested>(a, stream);
    }
}

template <typename T> void launch_mb(const StreamArgs& a, int quant_type, bool grouped, hipStream_t stream) {
    // rows of A held in registers per pass: 1, 2 or 4 (M = 3 runs the 4-row instance, its fourth row a duplicate
    // that is never stored); larger M - only reached for shapes the MFMA kernels do not take - loops passes of 4
    // over grid.y. A caller-supplied code table (legacy op) always runs row by row.
    if (a.M == 1 || a.code16 != nullptr) {
        // 16 wavefronts (ring of 2) for every call. Round 2 routed long row lists (>= 64 items per CU, 8 % S == 0) to the 8-wavefront
        // instance (ring of 4, decoded two stages at a time) on single timings in a fixed order - 8192^2 9.5 vs 10.7 us, 14336 x 4096 8.8
        // vs 9.8. Round 5, round-robin medians on two boxes (profiles/r5_stream_prologue_ab.txt): 16 wavefronts are level or ahead on
        // EVERY shape - 8192^2 8.54 / 8.81 vs 8.69 / 8.79 us, 11008 x 4096 6.42 / 6.56 vs 7.21 / 7.22, 14336 x 4096 7.54 / 7.81 vs 7.93 /
        // 8.11, 28672 x 8192 22.8 / 23.8 vs 23.6 / 24.4, 4096 x 11008 7.44 / 7.41 vs 8.10 / 8.17 - and that harness showed what single
        // timings carry: an 8 % first-measured penalty. The 8-wavefront instance stays for the tuning knob (A/B runs).
        const bool eight = g_tune.waves.load(std::memory_order_relaxed) == 8;
        if (eight)
            return launch_flags<T, 1, 8>(a, quant_type, grouped, stream);
        return launch_flags<T, 1, 16>(a, quant_type, grouped, stream);
    }
    if (a.M == 2) {
        // two rows still fit the 128-register budget of 16 wavefronts when the row is one phase long (measured: equal at
        // 4096^2, 13 % faster at 8192^2 and 11008 x 4096); the multi-phase form spills there and keeps 8 wavefronts
        if (g_tune.waves.load(std::memory_order_relaxed) != 8 &&
            make_geometry(a.rows_total, a.K, 2, 16, TypeInfo<T>::bytes, grouped, 0, 0).P == 1)
            return launch_flags<T, 2, 16>(a, quant_type, grouped, stream);
        return launch_flags<T, 2, 8>(a, quant_type, grouped, stream);
    }
    return launch_flags<T, 4, 8>(a, quant_type, grouped, stream);
}

template <typename T> void launch_generic(const GenericArgs& p, hipStream_t stream) {
    dim3 grid((p.N + 3) / 4, p.M);
    if (p.absmax8)
        hipLaunchKernelGGL((gemv4_generic_kernel<T, true>), grid, dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL((gemv4_generic_kernel<T, false>), grid, dim3(256), 0, stream, p);
}

bool stream_ok(const void* A, int M, int K, int blocksize) {
    // (the phase count travels in 9 bits and grid.y in 16: far beyond any real layer, the generic kernel takes the rest)
    return (K % 32 == 0) && K <= 511 * kSegK && M <= 65535 && blocksize >= 32 && is_pow2(blocksize) && aligned_to(A, 16);
}

void launch_stream_any(int dtype, const StreamArgs& a, int quant_type, bool grouped, hipStream_t stream) {
    if (dtype == 2)
        launch_mb<bf16>(a, quant_type, grouped, stream);
    else if (dtype == 1)
        launch_mb<f16>(a, quant_type, grouped, stream);
    else
        launch_mb<float>(a, quant_type, grouped, stream);
    BNB_CHECK_LAUNCH();
}

// (peer chain) the launch: the M = 1, 16-wavefront, single-phase instance with the kPeer paths compiled in
template <typename T, int FLAGS> void launch_peer(const StreamArgs& a, const Geometry& ge, hipStream_t stream) {
    auto kern = gemv4_stream_kernel<T, 1, 16, kRing, FLAGS | kNT | kPeer>;
    static LdsLimit lds_limit;
    ensure_dynamic_lds(lds_limit, reinterpret_cast<const void*>(kern), ge.lds);
    const StreamMat& m0 = a.mat[0];
    hipLaunchKernelGGL(kern, dim3(ge.grid_x, 1), dim3(16 * 64), ge.lds, stream, a.A, m0.B, m0.absmax, m0.absmax8, m0.N, a.K,
                       (1 & 0x3FFFF) | (a.bs_shift << 18) | (1 << 23) | ((a.peer.mode & 15) << 24),
                       ge.R | (ge.SW << 16) | (ge.G << 21), (256 + ge.SW - 1) / ge.SW, a);
}
template <typename T> void launch_peer_flags(const StreamArgs& a, const Geometry& ge, int quant_type, hipStream_t stream) {
    const int sel = (a.mat[0].absmax8 != nullptr ? 1 : 0) | (quant_type == kFP4 ? 2 : 0);
    switch (sel) {
    case 0: return launch_peer<T, 0>(a, ge, stream);
    case 1: return launch_peer<T, kNested>(a, ge, stream);
    case 2: return launch_peer<T, kFp4>(a, ge, stream);
    default: return launch_peer<T, kFp4 | kNested>(a, ge, stream);
    }
}

// (peer chain) the current exchange as a plain [nvalues] tensor: one granule per lane, re-fetched until its tag is there
template <typename T>
__global__ __launch_bounds__(256) void peer_chain_read_kernel(PeerChain pc, T* __restrict__ out, int nvalues) {
    uint32_t* const hdr = reinterpret_cast<uint32_t*>(pc.local);
    const uint32_t epoch = pc.epoch_word[0] + pc.epoch_offset;
    const unsigned char* const src = pc.local + kChainDataOffset + static_cast<size_t>(pc.epoch_offset & (kChainRegions - 1u)) * pc.max_granules * 8u;
    using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;
    uint32_t wait_bound = pc.spin_bound;
    for (int gi = blockIdx.x * 256 + threadIdx.x; 2 * gi < nvalues; gi += gridDim.x * 256) {
        const unsigned char* const addr = src + static_cast<size_t>(gi) * 8u;
        u32x2 gr;
        uint32_t spins = 0;
        for (;;) {
            asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(gr) : "v"(addr) : "memory");
            if (gr[1] == epoch)
                break;
            if (spins == 0 && wait_bound == pc.spin_bound) {
                // (first miss: a wait on this buffer that already gave up - the sticky status word - caps every later one, see the gemv)
                uint32_t st_word;
                const uint32_t* const st_addr = hdr + 1;
                asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(st_word) : "v"(st_addr) : "memory");
                if (st_word != 0u)
                    wait_bound = kPeerPollsAfterTimeout;
            }
            if (++spins > wait_bound) {
                __hip_atomic_store(hdr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                gr[0] = 0xFFFFFFFFu;
                wait_bound = kPeerPollsAfterTimeout;
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        *reinterpret_cast<uint32_t*>(out + 2 * gi) = gr[0];
    }
    // the last workgroup (of at most 8) advances the buffer's epoch past the chain that ends here (visible to the next launch on
    // this device: kernel boundary). Every wavefront of this workgroup has read the epoch word (first statement of the kernel) and
    // finished its granules before the workgroup counts as done: without the barrier a late-scheduled wavefront of the LAST
    // workgroup could still be in front of its epoch load when thread 0 rewrites the word.
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t prev = __hip_atomic_fetch_add(hdr + 2, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == gridDim.x - 1u) {
            hdr[2] = 0u;
            pc.epoch_word[0] = epoch;
        }
    }
}

// The shape preconditions of the peer-chain form and the geometry it launches with - ONE function, shared by the launcher and by
// the host layer's query (bnb_mi355x_gemv_4bit_peer_serves): what the host decides from shapes is what the launcher accepts.
// Pointer alignment (B, and A when x is a tensor) is the caller's to check on top of this.
static bool peer_geometry(int world, int ns, int K, int blocksize, int mode, long max_values, int wg_limit, Geometry* out_ge, bool* out_quads) {
    if (world < 1 || world > 8 || ns < 2 || (ns & 1) || K < 32 || (K % 32) != 0 || blocksize < 32 || !is_pow2(blocksize) || (mode & 3) == 0 ||
        max_values < 4 || (max_values & 3) || max_values >= (1L << 28))
        return false; // (max_values in fours: max_granules is then even, every region 16-byte aligned - the quad stores and b128 fetches need it)
    if ((mode & 1) && (K > kChainRounds * 2048 || K > max_values))
        return false;
    if ((mode & 8) && (!(mode & 2) || (ns & 3)))
        return false; // gated is a form of production, over (gate, up) row pairs that come in fours
    if ((mode & 2) && static_cast<long>(world) * ((mode & 8) ? ns / 2 : ns) > max_values)
        return false; // (a gated launch puts one activation per row PAIR into the exchange)
    // rows per workgroup: even (granules are row pairs), and no more workgroups than the caller allows (ranks that share one
    // device - a test set-up - must be co-resident: a launch that waits for its peers may not fill the device alone)
    int cus = device_cu_count();
    if (wg_limit > 0 && wg_limit < cus)
        cus = wg_limit;
    int R = (ns + cus - 1) / cus;
    R += R & 1;
    if ((ns & 3) == 0)
        R = (R + 3) & ~3; // rows in fours: two granules per store
    const Geometry ge = make_geometry(ns, K, 1, 16, 2, false, 0, R);
    // (one phase: K within the workgroup's segment columns; an even R survives make_geometry's clamp to the partial-sum slots)
    if (ge.P != 1 || (ge.R & 1))
        return false;
    if (out_ge)
        *out_ge = ge;
    const bool quads = (ns & 3) == 0 && (ge.R & 3) == 0;
    if ((mode & 8) && !quads)
        return false; // (the partial-sum slots clamped the rows per workgroup off a multiple of four)
    if (out_quads)
        *out_quads = quads;
    return true;
}

} // namespace

bool gemv_4bit_peer_serves(int world, int ns, int K, int blocksize, int mode, long max_values, int wg_limit) {
    return peer_geometry(world, ns, K, blocksize, mode, max_values, wg_limit, nullptr, nullptr);
}

// Peer-chain form of the M = 1 gemv (see PeerChain). mode bit 0: x = the current exchange (A ignored, K values), bit 1: y goes
// to every rank's exchange buffer (and to out_local when non-NULL). Returns false - nothing launched - when the problem is
// outside the form's preconditions; the caller then takes the unfused path (kernel + all-gather).
bool gemv_4bit_peer(void* const* bufs, void* epoch_word, int world, int rank, int dtype, const void* A, const uint8_t* B, const float* absmax,
                    const uint8_t* absmax8, const float* absmax_code, const float* absmax_offset, const void* bias, void* out_local,
                    int ns, int K, int blocksize, int quant_type, int mode, long max_values, int wg_limit, uint32_t epoch_offset,
                    uint32_t spin_bound, hipStream_t stream) {
    Geometry ge;
    bool quads = false;
    if (epoch_word == nullptr || (dtype != 1 && dtype != 2) || rank < 0 || rank >= world || !aligned_to(B, 16) ||
        !peer_geometry(world, ns, K, blocksize, mode, max_values, wg_limit, &ge, &quads))
        return false;
    if (!(mode & 1) && !aligned_to(A, 16))
        return false;
    if (!(mode & 2) && out_local == nullptr)
        return false;

    StreamArgs a;
#ifdef BNB_PROFILING
    a.dbg = g_dbg_buf;
#endif
    for (int r = 0; r < 8; ++r)
        a.peer.base[r] = static_cast<unsigned char*>(r < world ? bufs[r] : nullptr);
    a.peer.local = a.peer.base[rank];
    a.peer.epoch_word = static_cast<uint32_t*>(epoch_word);
    a.peer.world = world;
    a.peer.rank = rank;
    a.peer.max_granules = static_cast<uint32_t>(max_values / 2);
    a.peer.spin_bound = spin_bound;
    a.peer.epoch_offset = epoch_offset;
    a.peer.mode = (mode & 3) | (quads ? 4 : 0) | (mode & 8);
    // (x from the exchange: the preloaded pointer slot carries the address of the exchange's region - its position in the chain)
    a.A = (mode & 1) ? static_cast<const void*>(a.peer.local + kChainDataOffset +
                                                 static_cast<size_t>(epoch_offset & (kChainRegions - 1u)) * a.peer.max_granules * 8u)
                     : A;
    a.code16 = nullptr;
    a.M = 1;
    a.K = K;
    a.bs_shift = ilog2(blocksize);
    a.rows_total = ns;
    a.nmat = 1;
    for (int i = 0; i < kMaxGroup; ++i)
        a.mat[i] = StreamMat{B, absmax, absmax8, absmax_code, absmax_offset, out_local, bias, ns, i == 0 ? 0 : 0x7FFFFFFF};
    if (dtype == 2)
        launch_peer_flags<bf16>(a, ge, quant_type, stream);
    else
        launch_peer_flags<f16>(a, ge, quant_type, stream);
    BNB_CHECK_LAUNCH();
    g_last_gemm_kernel = kKernelStream;
    return true;
}

void peer_chain_read(void* const* bufs, void* epoch_word, int world, int rank, int dtype, void* out, int nvalues, long max_values, uint32_t epoch_offset,
                     uint32_t spin_bound, hipStream_t stream) {
    PeerChain pc;
    for (int r = 0; r < 8; ++r)
        pc.base[r] = static_cast<unsigned char*>(r < world ? bufs[r] : nullptr);
    pc.local = pc.base[rank];
    pc.epoch_word = static_cast<uint32_t*>(epoch_word);
    pc.world = world;
    pc.rank = rank;
    pc.max_granules = static_cast<uint32_t>(max_values / 2);
    pc.spin_bound = spin_bound;
    pc.epoch_offset = epoch_offset;
    pc.mode = 0;
    const int granules = (nvalues + 1) / 2;
    const int grid = (granules + 255) / 256 < 8 ? (granules + 255) / 256 : 8;
    if (dtype == 2)
        hipLaunchKernelGGL(peer_chain_read_kernel<bf16>, dim3(grid > 0 ? grid : 1), dim3(256), 0, stream, pc, static_cast<bf16*>(out), nvalues);
    else
        hipLaunchKernelGGL(peer_chain_read_kernel<f16>, dim3(grid > 0 ? grid : 1), dim3(256), 0, stream, pc, static_cast<f16*>(out), nvalues);
    BNB_CHECK_LAUNCH();
}

// Sweep-only overrides (0 / -1 = built-in choice). Atomics: a sweep thread can never corrupt a concurrent launch,
// it can only change which (always correct) geometry that launch uses.
void gemv_4bit_stream_tuning(int ns, int sw, int rows_per_wg, int nt, int waves) {
    g_tune.ns.store(ns, std::memory_order_relaxed);
    g_tune.sw.store(sw, std::memory_order_relaxed);
    g_tune.rows.store(rows_per_wg, std::memory_order_relaxed);
    g_tune.nt.store(nt, std::memory_order_relaxed);
    g_tune.waves.store(waves, std::memory_order_relaxed);
}

// Entry used by c_api.hip. dtype: 0 = f32, 1 = f16, 2 = bf16. The streaming kernel whenever its preconditions hold
// (K % 32 == 0, 16-byte aligned A and B, blocksize a power of two >= 32), else the generic one. Any M >= 1.
void gemv_4bit_stream(int dtype, const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax8,
                   const float* absmax_code, const float* absmax_offset, const float* code16, void* out,
                   const void* bias, int M, int N, int K, int blocksize, int quant_type, hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0)
        return;
    if (stream_ok(A, M, K, blocksize) && aligned_to(B, 16)) {
        StreamArgs a;
#ifdef BNB_PROFILING
        a.dbg = g_dbg_buf;
#endif
        a.A = A;
        a.code16 = code16;
        a.M = M;
        a.K = K;
        a.bs_shift = ilog2(blocksize);
        a.rows_total = N;
        a.nmat = 1;
        for (int i = 0; i < kMaxGroup; ++i)
            a.mat[i] = StreamMat{B, absmax, absmax8, absmax_code, absmax_offset, out, bias, N, i == 0 ? 0 : 0x7FFFFFFF};
        launch_stream_any(dtype, a, quant_type, false, stream);
        g_last_gemm_kernel = kKernelStream;
        return;
    }
    g_last_gemm_kernel = kKernelGeneric;
    GenericArgs p{A, B, absmax, absmax8, absmax_code, absmax_offset, code16, out, bias, M, N, K, ilog2(blocksize), quant_type};
    if (dtype == 0)
        launch_generic<float>(p, stream);
    else if (dtype == 1)
        launch_generic<f16>(p, stream);
    else
        launch_generic<bf16>(p, stream);
    BNB_CHECK_LAUNCH();
}

// Grouped launch: `count` weight matrices (same K, blocksize, quant_type, nested-ness) applied to the same
// activations A[M, K] in one launch. Returns false when the group does not meet the streaming kernel's preconditions
// (the caller then issues the matrices one by one).
bool gemv_4bit_grouped(int dtype, const void* A, int count, const uint8_t* const* B, const float* const* absmax,
                       const uint8_t* const* absmax8, const float* const* absmax_code, const float* const* absmax_offset,
                       void* const* out, const void* const* bias, const int* N, int M, int K, int blocksize, int quant_type,
                       hipStream_t stream) {
    if (count < 1 || count > kMaxGroup || M < 1 || M > 4 || K <= 0 || !stream_ok(A, M, K, blocksize))
        return false;
    StreamArgs a;
#ifdef BNB_PROFILING
    a.dbg = nullptr;
#endif
    a.A = A;
    a.code16 = nullptr;
    a.M = M;
    a.K = K;
    a.bs_shift = ilog2(blocksize);
    a.nmat = count;
    long rows = 0;
    const bool nested = absmax8 != nullptr && absmax8[0] != nullptr;
    for (int i = 0; i < count; ++i) {
        if (N[i] <= 0 || !aligned_to(B[i], 16))
            return false;
        if (((absmax8 != nullptr && absmax8[i] != nullptr)) != nested)
            return false;
        a.mat[i] = StreamMat{B[i], absmax[i], nested ? absmax8[i] : nullptr, nested ? absmax_code[i] : nullptr,
                             nested ? absmax_offset[i] : nullptr, out[i], bias ? bias[i] : nullptr, N[i], static_cast<int>(rows)};
        rows += N[i];
    }
    if (rows > 0x7FFFFFFF)
        return false;
    for (int i = count; i < kMaxGroup; ++i) {
        a.mat[i] = a.mat[0];
        a.mat[i].row_start = 0x7FFFFFFF;
    }
    a.rows_total = static_cast<int>(rows);
    const int mb = M >= 3 ? 4 : M;
    if (make_geometry(a.rows_total, K, mb, mb <= 2 ? 16 : 8, dtype == 0 ? 4 : 2, true, 0, 0).P > 1 ||
        make_geometry(a.rows_total, K, mb, 8, dtype == 0 ? 4 : 2, true, 0, 0).P > 1)
        return false; // rows longer than one workgroup's segment columns: the single-matrix path has the phase loop
    launch_stream_any(dtype, a, quant_type, true, stream);
    g_last_gemm_kernel = kKernelStream;
    return true;
}

} // namespace bnb
