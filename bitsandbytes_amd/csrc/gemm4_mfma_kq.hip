// gemm4_mfma_kq.hip — "K-quarter" MFMA kernel for 17 ... 64-row batches on gfx950 (round 4):
//     out[m, n] = sum_b scale[n, b] * ( sum_{k in block b} A[m, k] * T(code[B[n, k]]) )   (+ bias[n])        T in {bf16, fp16}
//
// The same arithmetic as the producer/consumer kernel of gemm4_mfma.hip (codes through the matrix pipe, the exact fp32 absmax
// applied to the fp32 partial tile of each 64-k block - the idea of the reference's SIMT kernel, csrc/gemm_4bit_simt.cu:436-444,
// on MFMA; the tensor-core capability itself exists in the reference on CUDA only, csrc/gemm_4bit_sm80.cu:127-457,493-689),
// re-cut around what the round-4 timeline of that kernel showed (profiles/r4_timeline_pc_c3.txt: per 256-k chunk 2400 cycles
// of compute, 600 of DMA issue and 600-1600 at the barrier waiting for the activation stage, which can only be requested one
// chunk ahead because every consumer reads ALL of it, fragment by fragment, during the whole chunk):
//
//  * v_mfma_f32_32x32x16: half the matrix-pipe issue slots per flop of the 16x16x32 form, and a decoded weight fragment
//    feeds 64 rows with two instructions.
//  * workgroup = 128 output columns x (32 MT rows, MT = 1 | 2) x one K slice, 8 wavefronts = 2 column groups c (64 columns) x 4
//    K quarters q: wavefront (c, q) multiplies quantization block q (64 k) of every 256-k chunk for its 64 columns and all rows.
//    It therefore needs only 8 (4 MT) activation fragments and 32 weight bytes per lane per chunk - few enough to copy them to
//    REGISTERS at the top of the chunk. After that copy (second barrier of the chunk) the LDS stages are free again: the
//    activation ring is three half-chunk slots, the weight ring two slots, and both are requested up to two chunks ahead.
//    An activation fragment is read from the LDS once per 64 columns (the producer/consumer kernel: once per 16).
//  * the byte -> (code[hi], code[lo]) table holds 64 lane-private copies at a 256-byte stride, so a table address is ONE
//    v_perm_b32 (the 32-copy table of the producer/consumer kernel costs two VALU instructions per byte; the vector-ALU issue
//    port - one non-FMA instruction per ~4.2 cycles and SIMD whatever the number of wavefronts, profiles/r3_issue_rate_ubench.txt
//    - is the first unit that kernel saturates).
//  * every byte enters the LDS by LDS-DMA in full 128-byte lines, XOR swizzles applied on the source side; all eight
//    wavefronts issue the same number of DMA instructions per chunk, interleaved with their MFMAs; waits are exact counts.
//  * the four K quarters of a column group are added once, after the loop, in a fixed order through the (then free) LDS;
//    K slices across workgroups write fp32 slabs that gemm4_finalize adds in slice order: bit-reproducible.
#include "bnb_common.h"

#include <type_traits>

namespace bnb {

#ifdef BNB_PROFILING
extern unsigned long long* g_dbg_buf;
#endif

// gemm4_mfma.hip
void gemm_4bit_finalize(int dtype, const float* ws, const void* bias, void* out, int M, int N, int kslices, hipStream_t stream);
float* gemm_4bit_internal_workspace(size_t bytes, hipStream_t stream);

namespace {

using i32x4 = __attribute__((ext_vector_type(4))) int;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <typename T> struct KqMma;
template <> struct KqMma<bf16> {
    using frag = __attribute__((ext_vector_type(8))) bf16;
    static __device__ __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(frag, a), __builtin_bit_cast(frag, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint32_t pack(float first, float second) {
        using V = __attribute__((ext_vector_type(2))) bf16;
        V v;
        v[0] = static_cast<bf16>(first);
        v[1] = static_cast<bf16>(second);
        return __builtin_bit_cast(uint32_t, v);
    }
};
template <> struct KqMma<f16> {
    using frag = __attribute__((ext_vector_type(8))) f16;
    static __device__ __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(frag, a), __builtin_bit_cast(frag, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint32_t pack(float first, float second) {
        using V = __attribute__((ext_vector_type(2))) f16;
        V v;
        v[0] = static_cast<f16>(first);
        v[1] = static_cast<f16>(second);
        return __builtin_bit_cast(uint32_t, v);
    }
};

constexpr int kKqCols = 128;   // output columns per workgroup: 2 column groups of 64
constexpr int kKqChunk = 256;  // k per chunk: one 64-k quantization block per K quarter
constexpr int kKqWaves = 8;    // 2 column groups x 4 K quarters
constexpr int kKqLut = 65536;  // 256 entries x 64 lane-private copies x 4 B, at LDS address 0

template <int MT, bool NESTED> struct KqLds {
    // All rings are cut in HALF-chunk units ("halves": 128 k = the blocks of one wavefront group, see the kernel): half
    // u = 2 j + g of the slice is chunk j, blocks 2 g and 2 g + 1.
    static constexpr int DA = 3;                       // activation halves in the ring
    static constexpr int DW = 4;                       // weight / scale halves in the ring
    static constexpr int AHB = 32 * MT * 256;          // bytes of an activation half: 32 MT rows x 128 k
    static constexpr int WHB = kKqCols * 64;           // bytes of a weight half: 128 rows x 128 k
    static constexpr int DS = 3;                       // scale CHUNKS in the ring
    static constexpr int SCB = NESTED ? 512 : 2048;    // bytes of a scale chunk: [128 columns][4 blocks] fp32, or (nested) [128] dwords of four 8-bit codes
    static constexpr int ABase = kKqLut;
    static constexpr int WBase = ABase + DA * AHB;
    static constexpr int SBase = WBase + DW * WHB;
    static constexpr int Code2 = SBase + DS * SCB;
    static constexpr int A2T = Code2 + 1024;           // (nested) [2 candidates][128 columns] second-level absmax of the slice
    static constexpr int Bytes = A2T + (NESTED ? 1024 : 0);
    static constexpr int RedBytes = 49152 * MT;        // the epilogue's exchange area (the whole LDS is free by then)
    static constexpr int Alloc = Bytes > RedBytes ? Bytes : RedBytes;
    static_assert(Alloc <= 155 * 1024, "LDS");
};

struct KqArgs {
#ifdef BNB_PROFILING
    unsigned long long* dbg;
#endif
    const float* absmax_code;
    const float* absmax_offset;
    void* out;
    const void* bias;
    float* ws; // fp32 [kslices][M][N] partial slabs when kslices > 1
};

// Time stamps (measurement build): s_memtime values collect in scalar registers and are stored ONCE, at the end of the kernel -
// a store (and the load of its address) at the point of the stamp costs a few hundred cycles and drains the LDS counter.
#ifdef BNB_PROFILING
#define BNB_KQ_STAMP(i) { if constexpr ((ABL & 64) != 0) ts[i] = __builtin_amdgcn_s_memtime(); }
#else
#define BNB_KQ_STAMP(i) {}
#endif

__device__ __forceinline__ float kq_code_literal(int i, bool fp4) {
    // compare/select over literals: no memory access in front of the table
    constexpr float nf4[16] = {BNB_NF4_VALUES};
    constexpr float fp4v[16] = {BNB_FP4_VALUES};
    float v = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j)
        v = (i == j) ? (fp4 ? fp4v[j] : nf4[j]) : v;
    return v;
}

// LDS-DMA, spelled out (the compiler's own tracking of buffer_load ... lds makes every later ds_read wait for ALL outstanding
// DMA): the hand-off is by counted waits + s_barrier (see the chunk loop). LDS base and scalar offset are wavefront-uniform.
__device__ __forceinline__ i32x4 kq_rsrc(const void* base) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    return i32x4{static_cast<int>(a), static_cast<int>((a >> 32) & 0xFFFFu), 0x7FFFFFFF, 0x00020000};
}
__device__ __forceinline__ void kq_dma16(i32x4 rs, uint32_t lds, uint32_t voff, uint32_t soff) {
    lds = __builtin_amdgcn_readfirstlane(lds);
    soff = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
}
// (weights: read once, by one CU - the non-temporal policy shortens issue -> landed by ~18 %, MI355X guide "nt-weights")
__device__ __forceinline__ void kq_dma16_nt(i32x4 rs, uint32_t lds, uint32_t voff, uint32_t soff) {
    lds = __builtin_amdgcn_readfirstlane(lds);
    soff = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds" ::"s"(lds), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
}
__device__ __forceinline__ void kq_dma4(i32x4 rs, uint32_t lds, uint32_t voff) {
    lds = __builtin_amdgcn_readfirstlane(lds);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(lds), "v"(voff), "s"(rs) : "memory", "m0");
}
__device__ __forceinline__ void kq_dma1(i32x4 rs, uint32_t lds, uint32_t voff) {
    // one byte per lane, landing in the lane's dword slot (the other three bytes are not defined: the reader masks)
    lds = __builtin_amdgcn_readfirstlane(lds);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_ubyte %1, %2, 0 offen lds" ::"s"(lds), "v"(voff), "s"(rs) : "memory", "m0");
}
template <int N> __device__ __forceinline__ void kq_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void kq_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
template <int LGKM> __device__ __forceinline__ void kq_barrier_lgkm() {
    // (LDS operations return in order: the LGKM youngest ones - table look-ups whose results nobody else needs - may stay in flight)
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(LGKM) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
template <typename V> __device__ __forceinline__ V kq_lds_read(uint32_t addr) {
    return *reinterpret_cast<const __attribute__((address_space(3))) V*>(addr);
}

// grid = (ceil(N / 128), kslices, ceil(M / (32 MT))); 512 threads.
template <typename T, int MT, bool NESTED, bool SFAST, int ABL = 0>
__global__ __launch_bounds__(kKqWaves * 64) void gemm4_mfma_kq_kernel(
    // hot arguments as separate scalars: preloaded into SGPRs by the command processor (14 dwords)
    const void* hot_A, const uint8_t* hot_B, const float* hot_absmax, const uint8_t* hot_absmax8, int hot_M, int hot_N,
    int hot_K, int hot_flags /* bs_shift | fp4 << 8 */, int hot_cps /* chunks per K slice */, int hot_kslices,
    const KqArgs p) {
    using L = KqLds<MT, NESTED>;
    constexpr int DA = L::DA, DW = L::DW;
    constexpr int AI = 2 * MT;               // activation DMA instructions per wavefront and batch (4 rows x 256 B each)
    constexpr int BATCH = AI + 2;            // activation + weight DMA instructions per wavefront and half
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef BNB_PROFILING
    unsigned long long ts[(ABL & 64) ? 16 : 1];
    if constexpr ((ABL & 64) != 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            ts[i] = 0;
    }
#endif
    BNB_KQ_STAMP(0)
    // Two groups of four wavefronts in opposite phases: while group g computes (matrix pipe, table look-ups), the other one
    // loads (LDS -> registers, DMA requests). Wavefronts w and w + 4 share a SIMD: one of each group.
    const int g = wave >> 2;                 // group = which half of every chunk (blocks 2 g, 2 g + 1)
    const int wi = wave & 3;                 // index in the group
    const int c = wi >> 1;                   // column group (64 columns)
    const int qq = wi & 1;                   // block of the half
    const int q = 2 * g + qq;                // block of the chunk = K quarter
    const int n = lane & 31, h = lane >> 5;  // MFMA roles: column / row n, k half h
    const int M = hot_M, N = hot_N, K = hot_K;
    const int bs_shift = hot_flags & 31;
    const bool fp4 = (hot_flags >> 8) & 1;
    // ablations (template parameter; non-zero instances exist in the measurement build only, selected with bnb_mi355x_set_tuning
    // knob0): 1 no activation DMA after the start-up, 2 no weight / scale DMA after the start-up, 4 no table look-ups, 8 no MFMA,
    // 16 no fragment / weight reads in the load phase, 32 no scale FMAs: results are wrong, the timing tells what each part costs
#define BNB_KQ_ON(bit) (!(ABL & (bit)))
    const int col0 = blockIdx.x * kKqCols;
    const int m_base = blockIdx.z * (32 * MT);
    const int chunks_total = K >> 8;
    const int cb = blockIdx.y * hot_cps;
    int ce = cb + hot_cps;
    ce = ce < chunks_total ? ce : chunks_total;
    const int nc = ce - cb; // chunks of this slice (>= 1: the host makes every slice non-empty)
    const int H = 2 * nc;   // halves of this slice
    if (reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)smem) != 0)
        __builtin_trap(); // the table is addressed with raw v_perm_b32 results: it must sit at LDS address 0

    // ---- sources. Rows past the end (ragged N or M) re-read the last row: MFMA rows / columns are independent and those
    // results are never stored. All byte offsets are < 2^31 (gemm_4bit_kq_supported).
    const i32x4 rs_w = kq_rsrc(hot_B), rs_a = kq_rsrc(hot_A), rs_s = kq_rsrc(hot_absmax), rs_q = kq_rsrc(hot_absmax8);
    uint32_t lo_a[AI], lo_w[2];
#pragma unroll
    for (int t = 0; t < AI; ++t) {
        // activations: instruction i = wi + 4 t of a half covers rows 4 i .. 4 i + 3 (256 B each); lane 16 r + p' writes LDS
        // piece p' of its row and fetches piece p' ^ (row & 15)
        const int row = 4 * (wi + 4 * t) + (lane >> 4);
        int m = m_base + row;
        m = m < M ? m : M - 1;
        lo_a[t] = (static_cast<uint32_t>(m) * static_cast<uint32_t>(K)) * 2u + static_cast<uint32_t>(((lane & 15) ^ (row & 15)) << 4);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        // weights: instruction i = wi + 4 t of a half covers rows 16 i .. 16 i + 15 (64 B each); lane 4 r + p' writes LDS piece
        // p' and fetches piece p' ^ ((row >> 2) & 3)
        const int row = 16 * (wi + 4 * t) + (lane >> 2);
        int col = col0 + row;
        col = col < N ? col : N - 1;
        lo_w[t] = static_cast<uint32_t>(col) * static_cast<uint32_t>(K >> 1) + static_cast<uint32_t>(((lane & 3) ^ ((row >> 2) & 3)) << 4);
    }
    // scales travel per CHUNK, in few, wide requests (one scattered request costs the address unit one look-up per row: the
    // per-half, per-lane form of the first builds cost 0.15 us per chunk and wavefront-instruction, profiles/r4_kq_nested_ablate.txt).
    // Group g requests columns 64 g .. 64 g + 63 of the workgroup:
    //   fast form (blocksize 64, aligned statistics): lane = column; fp32 absmax: ONE 16-byte request per column and chunk (its
    //   four scales), by wavefront 0 of the group; nested: one dword of four 8-bit codes per column and chunk, by wavefront 0
    //   (the second-level absmax of a column changes at most once inside a slice of <= 256 blocks: its two candidates are
    //   fetched once, at start-up, and kept in registers).
    //   generic form (fp32 absmax only): lane 4 r + i = (column 16 wi + r, 64-k block i), one dword each, every wavefront one request.
    uint32_t lo_s;
    const int s_extra = SFAST ? (wi == 0 ? 1 : 0) : 1; // scale requests of this wavefront per compute phase
    {
        int col = col0 + 64 * g + (SFAST ? lane : 16 * wi + (lane >> 2));
        col = col < N ? col : N - 1;
        if constexpr (SFAST)
            lo_s = static_cast<uint32_t>(col) * static_cast<uint32_t>(K >> 6); // index of the column's first block
        else
            lo_s = static_cast<uint32_t>(col) * static_cast<uint32_t>(K) + static_cast<uint32_t>(64 * (lane & 3)); // element index
    }
    // half u of the slice (chunk cb + (u >> 1), blocks 2 (u & 1) ..) into a ring slot - every wavefront of the issuing group its part
    // (the start-up may name halves past the end of a short slice: they become the last half again, into slots nobody reads)
    auto issue_a = [&](int u, int slot) {
        u = u < H ? u : H - 1;
        const uint32_t so = static_cast<uint32_t>(cb + (u >> 1)) * 512u + static_cast<uint32_t>(u & 1) * 256u;
#pragma unroll
        for (int t = 0; t < AI; ++t)
            kq_dma16(rs_a, static_cast<uint32_t>(L::ABase + slot * L::AHB + (wi + 4 * t) * 1024), lo_a[t], so);
    };
    auto issue_w = [&](int u, int slot, int t) {
        u = u < H ? u : H - 1;
        const uint32_t so = static_cast<uint32_t>(cb + (u >> 1)) * 128u + static_cast<uint32_t>(u & 1) * 64u;
        // (the non-temporal policy, which shortens issue -> landed where a CU fetches whole lines once, costs 25 % here: the two
        // 64-byte halves of a line are requested a phase apart, by different wavefronts - profiles/r4_kq_variants.txt)
        if constexpr ((ABL & 128) != 0)
            kq_dma16_nt(rs_w, static_cast<uint32_t>(L::WBase + slot * L::WHB + (wi + 4 * t) * 1024), lo_w[t], so);
        else
            kq_dma16(rs_w, static_cast<uint32_t>(L::WBase + slot * L::WHB + (wi + 4 * t) * 1024), lo_w[t], so);
    };
    // scales of chunk cs (of the slice) into ring slot cs % 3: this group's columns, this wavefront's part
    auto issue_s = [&](int cs, int slot) {
        cs = cs < nc ? cs : nc - 1;
        const uint32_t ca = static_cast<uint32_t>(cb + cs);
        const uint32_t base = static_cast<uint32_t>(L::SBase + slot * L::SCB);
        if constexpr (SFAST) {
            const uint32_t blk = lo_s + ca * 4u;
            if constexpr (NESTED) {
                if (wi == 0)
                    kq_dma4(rs_q, base + static_cast<uint32_t>(g) * 256u, blk);
            } else {
                if (wi == 0)
                    kq_dma16(rs_s, base + static_cast<uint32_t>(g) * 1024u, blk * 4u, 0u);
            }
        } else {
            static_assert(SFAST || !NESTED, "nested statistics take the fast form (the host routes the other cases elsewhere)");
            const uint32_t blk = (lo_s + ca * 256u) >> bs_shift;
            kq_dma4(rs_s, base + static_cast<uint32_t>(g) * 1024u + static_cast<uint32_t>(wi) * 256u, blk * 4u);
        }
    };
    auto issue_w2 = [&](int u, int slot) {
        issue_w(u, slot, 0);
        issue_w(u, slot, 1);
    };

    // ---- start-up. Group 0 requests halves 0 and 2, group 1 halves 1 and 3 (the activation ring holds three): per
    // wavefront [S..][A0 W0][A2 W2] resp. [S..][A1 W1][W3] - half 0 first, the rest once the decode table is built (the first
    // bytes of a launch arrive ~5500 cycles after their request whatever the volume; what is requested with them only delays them).
    // Scales: chunks 0, 1 (group 0's columns) and 0, 1, 2 (group 1's) - the oldest entries of the queues.
    const uint32_t lane4 = static_cast<uint32_t>(lane) * 4u;
    if constexpr (NESTED) {
        // (the second-level code table, 256 floats: one dword DMA by each wavefront of group 1; an ordinary load here would
        // make the compiler drain the queue)
        if (g == 1)
            kq_dma4(kq_rsrc(p.absmax_code), static_cast<uint32_t>(L::Code2 + wi * 256), lane4 + static_cast<uint32_t>(wi) * 256u);
        // second-level absmax: entries idx0 and idx0 + 1 of every column, idx0 = index of the 256-block group of the slice's
        // first block in that column (wavefronts 0 / 1 of group g: candidate 0 / 1 of columns 64 g ..)
        if (wi < 2) {
            int col = col0 + 64 * g + lane;
            col = col < N ? col : N - 1;
            uint32_t idx = ((static_cast<uint32_t>(col) * static_cast<uint32_t>(K >> 6) + static_cast<uint32_t>(cb) * 4u) >> 8) + static_cast<uint32_t>(wi);
            const uint32_t last = (static_cast<uint32_t>(N) * static_cast<uint32_t>(K >> 6) - 1u) >> 8;
            idx = idx < last ? idx : last;
            kq_dma4(rs_s, static_cast<uint32_t>(L::A2T + wi * 512 + g * 256), idx * 4u);
        }
    }
    issue_s(0, 0);
    issue_s(1, 1);
    if (g == 1)
        issue_s(2, 2);
    constexpr bool PACED = (ABL & 2048) == 0; // (variant bit 2048 of the measurement build: everything at once)
    if (g == 0) {
        issue_a(0, 0);
        issue_w2(0, 0);
        if constexpr (!PACED) {
            issue_a(2, 2);
            issue_w2(2, 2);
        }
    } else if constexpr (!PACED) {
        issue_a(1, 1);
        issue_w2(1, 1);
        issue_w2(3, 3);
    }
    BNB_KQ_STAMP(1)
    float offset = 0.0f;
    if constexpr (NESTED) {
        // A SCALAR load, consumed here: as a vector load its first use - the scale arithmetic at the top of every compute phase -
        // got an s_waitcnt vmcnt(0) from the compiler, i.e. every compute phase of a nested call began by draining the
        // wavefront's DMA queue (nested calls were 1.3 us slower than plain ones on config 3; found in the ISA).
        typedef const __attribute__((address_space(4))) float* cfloat_ptr;
        int ob = __builtin_bit_cast(int, *(cfloat_ptr)(p.absmax_offset));
        asm volatile("" : "+s"(ob));
        offset = __builtin_bit_cast(float, ob);
    }
    {
        // decode table: entry e (a packed byte) = 64 copies of T2(code[e >> 4], code[e & 15]), 256 B per entry; the two halves
        // of the workgroup write 8 of its 16 16-byte chunks each, in an order rotated by e (eight lanes -> eight bank quads)
        const float cv = kq_code_literal((lane & 15) + opaque_zero(), fp4);
        const int cvb = __builtin_bit_cast(int, cv);
        const int e = tid & 255;
        const float hi = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((e >> 4) & 15) * 4, cvb));
        const float lov = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((e & 15) * 4, cvb));
        const uint32_t pr = KqMma<T>::pack(hi, lov);
        const u32x4 v = {pr, pr, pr, pr};
        u32x4* const dst = reinterpret_cast<u32x4*>(smem + e * 256);
        const int half = tid >> 8;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            dst[(8 * half + j + e) & 15] = v;
    }
    if constexpr (PACED) {
        if (g == 0) {
            issue_a(2, 2);
            issue_w2(2, 2);
        } else {
            issue_a(1, 1);
            issue_w2(1, 1);
            issue_w2(3, 3);
        }
    }

    const uint32_t perm_sel = 0x0C0C0400u + static_cast<uint32_t>(opaque_zero()); // {lane offset, weight byte, 0, 0}
    // consumer addresses, relative to slot 0
    //   packed weights: row 64 c + lane of the half, 16-byte pieces 2 qq and 2 qq + 1 of its 64 B, stored at piece ^ ((row >> 2) & 3)
    const uint32_t w_rd = static_cast<uint32_t>(L::WBase + (64 * c + lane) * 64 + (((2 * qq) ^ ((lane >> 2) & 3)) << 4));
    //   scale of column 64 c + 32 nt + n, block q of the chunk (nested: the column's dword of four codes; its second-level absmax 512 B behind)
    uint32_t s_rd[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
        s_rd[nt] = static_cast<uint32_t>(L::SBase) + (NESTED ? static_cast<uint32_t>((64 * c + 32 * nt + n) * 4) : static_cast<uint32_t>((64 * c + 32 * nt + n) * 16 + q * 4));
    //   activation fragment of step s, row tile mt: row 32 mt + n of the half, piece 8 qq + 4 h + s, stored at piece ^ (n & 15)
    uint32_t a_rd[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
        a_rd[s] = static_cast<uint32_t>(L::ABase + n * 256 + (((8 * qq + 4 * h + s) ^ (n & 15)) << 4));

    f32x16 acc[2][MT];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                acc[nt][mt][i] = 0.0f;

    // (nested) the two second-level absmax candidates of the lane's columns and the first chunk of the slice that takes the second
    float a2_lo[2] = {0.0f, 0.0f}, a2_hi[2] = {0.0f, 0.0f};
    int a2_jx[2] = {0, 0};

    // the wavefront's share of one half, in registers: loaded in a load phase, multiplied in the following compute phase
    u32x4 wraw0, wraw1, afr[4][MT];
    float sc[2];
    wraw0 = wraw1 = u32x4{0u, 0u, 0u, 0u};
    sc[0] = sc[1] = 0.0f;
    u32x4 bfr[3];   // decoded weight fragments: two look-ups ahead of the MFMAs
    uint32_t wt[2][4];
    // tile 1's partial sums of the half just computed and their scale: accumulated in the wavefront's NEXT load phase, which is
    // the shorter of the two (the compute phase is bound by what its one wavefront can issue)
    f32x16 part1[MT];
    float sc1p = 0.0f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i)
            part1[mt][i] = 0.0f;
    auto lut2 = [&](uint32_t w, int half, u32x4& r) __attribute__((always_inline)) {
#pragma unroll
        for (int b = 2 * half; b < 2 * half + 2; ++b)
            if (BNB_KQ_ON(4))
                r[b] = kq_lds_read<uint32_t>(__builtin_amdgcn_perm(w, lane4, perm_sel + (b << 8)));
    };
    // acc[nt][mt][first .. first + count) += scale * part[...]: pinned where it is written (an empty volatile asm per value: left
    // alone, the compiler sinks every scale FMA of a phase to the end of the phase, out of every MFMA's shadow)
    auto scale_fma = [&](int nt, int mt, int first, int count, float scale, const f32x16& part) __attribute__((always_inline)) {
#pragma unroll
        for (int i = first; i < first + count; i += 2)
            if (BNB_KQ_ON(32)) {
                // (two at a time: v_pk_fma_f32 - this kernel is bound by the number of instructions a wavefront can issue)
                f32x2 v = __builtin_elementwise_fma(f32x2{scale, scale}, f32x2{part[i], part[i + 1]}, f32x2{acc[nt][mt][i], acc[nt][mt][i + 1]});
                asm volatile("" : "+v"(v));
                acc[nt][mt][i] = v[0];
                acc[nt][mt][i + 1] = v[1];
            }
    };

    // ---- one half u: load phase (= phase u - 1: LDS -> registers, the activation requests of the phase, the first table
    // look-ups), barrier, compute phase (= phase u: 16 MT MFMAs with the remaining look-ups, the scale FMAs and the weight / scale
    // requests of the phase between them), barrier.
    //   ISSUE: request activation half u + 2 into the slot of half u - 1 (read one phase ago by the other group).
    //   the compute phase requests weight / scale half u + 4 into the slot this wavefront has just emptied.
    auto one_half = [&](auto issue, auto issue_wc, int u, int sa, int sw, int ss, auto stamp_c) __attribute__((always_inline)) {
        constexpr bool ISSUE = decltype(issue)::value;
        constexpr bool ISSUE_W = decltype(issue_wc)::value;
        constexpr bool stamp = decltype(stamp_c)::value;
        if constexpr (stamp)
            BNB_KQ_STAMP(2)
        if constexpr ((ABL & 1024) != 0)
            __builtin_amdgcn_s_setprio(1);
        if constexpr ((ABL & 512) != 0)
            __builtin_amdgcn_s_setprio(0);
        uint32_t q8[2];
        float a2[2], c2[2];
        if (BNB_KQ_ON(16)) {
            const uint32_t wso = static_cast<uint32_t>(sw * L::WHB), sso = static_cast<uint32_t>(ss * L::SCB), aso = static_cast<uint32_t>(sa * L::AHB);
            wraw0 = kq_lds_read<u32x4>(w_rd + wso);
            wraw1 = kq_lds_read<u32x4>((w_rd ^ 16u) + wso);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                if constexpr (NESTED) {
                    q8[nt] = kq_lds_read<uint32_t>(s_rd[nt] + sso);
                    a2[nt] = (u >> 1) >= a2_jx[nt] ? a2_hi[nt] : a2_lo[nt];
                } else {
                    sc[nt] = kq_lds_read<float>(s_rd[nt] + sso);
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    afr[s][mt] = kq_lds_read<u32x4>(a_rd[s] + aso + mt * 8192);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ISSUE) {
            if (BNB_KQ_ON(1))
                issue_a(u + 2, sa == 0 ? DA - 1 : sa - 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        // (tile 1 of the previous half, while the share reads return)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            scale_fma(1, mt, 0, 16, sc1p, part1[mt]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NESTED) {
            // (the second-level look-up depends on the 8-bit codes read above: issued here, behind the DMA requests and the
            // deferred FMAs, it finds them returned; its result is only combined in the compute phase)
            if (BNB_KQ_ON(16)) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    c2[nt] = kq_lds_read<float>(static_cast<uint32_t>(L::Code2) + __builtin_amdgcn_ubfe(q8[nt], static_cast<uint32_t>(8 * q), 8u) * 4u);
            }
        }
        // deal the packed weights: lane (n, h) gets dword 4 h + s of columns 64 c + n (tile 0) and 64 c + 32 + n (tile 1): the
        // high dwords of the lower half are swapped with the low dwords of the upper half; then the look-ups of the first two
        // fragments (they need nothing but registers and the table)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const auto sw2 = __builtin_amdgcn_permlane32_swap(wraw0[r], wraw1[r], false, false);
            wt[0][r] = sw2[0];
            wt[1][r] = sw2[1];
        }
        lut2(wt[0][0], 0, bfr[0]);
        lut2(wt[0][0], 1, bfr[0]);
        lut2(wt[0][1], 0, bfr[1]);
        lut2(wt[0][1], 1, bfr[1]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (stamp)
            BNB_KQ_STAMP(3)
        // (the share reads are complete - LDS operations return in order - when all but the 8 look-ups have returned; the
        // look-ups may cross the barrier: nobody else needs them)
        kq_barrier_lgkm<8>();
        __builtin_amdgcn_sched_barrier(0); // (an MFMA is not a memory operation: without this the compiler lifts the first one - and the wait for its look-ups - above the barrier)
        if constexpr (stamp) {
            BNB_KQ_STAMP(5)
#ifdef BNB_PROFILING
            ts[10] += ts[5] - ts[3]; // cycles at the barrier that ends the load phases, summed over the slice
            ts[13] += ts[3] - ts[2]; // cycles of load-phase work, summed over the slice
#endif
        }
        // ---- compute phase
        if constexpr (NESTED) {
            if (BNB_KQ_ON(16)) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    sc[nt] = nested_scale(c2[nt], a2[nt], offset);
            }
        }
        if constexpr ((ABL & 1024) != 0)
            __builtin_amdgcn_s_setprio(0);
        if constexpr ((ABL & 512) != 0)
            __builtin_amdgcn_s_setprio(1);
        f32x16 part[2][MT];
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const int nt = f >> 2, s = f & 3;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (s == 0) {
                    f32x16 z;
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        z[i] = 0.0f;
                    part[nt][mt] = z;
                }
                if (BNB_KQ_ON(8))
                    part[nt][mt] = KqMma<T>::run(afr[s][mt], bfr[f % 3], part[nt][mt]);
                // in the shadow of the MFMA: half (all, with one row tile) of the look-ups of fragment f + 2; one of the phase's
                // weight / scale requests; once tile 0 is complete (f >= 4), its scale FMAs
                if (f + 2 < 8) {
                    if (MT == 1) {
                        lut2(wt[(f + 2) >> 2][(f + 2) & 3], 0, bfr[(f + 2) % 3]);
                        lut2(wt[(f + 2) >> 2][(f + 2) & 3], 1, bfr[(f + 2) % 3]);
                    } else {
                        lut2(wt[(f + 2) >> 2][(f + 2) & 3], mt, bfr[(f + 2) % 3]);
                    }
                }
                if (ISSUE_W && mt == MT - 1 && BNB_KQ_ON(2)) {
                    if (f == 0)
                        issue_w(u + 4, sw, 0);
                    if (f == 1)
                        issue_w(u + 4, sw, 1);
                    if (f == 2) // scales of chunk (u >> 1) + 2 + g, into the slot of the chunk whose second half the other group read last
                        issue_s((u >> 1) + 2 + g, ss + (g == 0 ? 2 : 0) >= L::DS ? ss + (g == 0 ? 2 : 0) - L::DS : ss + (g == 0 ? 2 : 0));
                }
                if (nt == 1) {
                    // 16 MT FMAs of tile 0 over the 4 MT MFMAs of tile 1: fewer beside the MFMAs that also carry look-ups
                    constexpr int per_early = 2, per_late = 6; // per MFMA: f = 4, 5 (look-ups beside them) / f = 6, 7
                    const int k0 = (s < 2) ? (s * MT + mt) * per_early : 2 * MT * per_early + ((s - 2) * MT + mt) * per_late;
                    const int cnt = (s < 2) ? per_early : per_late;
#pragma unroll
                    for (int k = k0; k < k0 + cnt; k += 2)
                        scale_fma(0, k / 16, k % 16, 2, sc[0], part[0][k / 16]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // tile 1: in the next load phase
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            part1[mt] = part[1][mt];
        sc1p = sc[1];
        if constexpr (stamp)
            BNB_KQ_STAMP(6)
        // own requests: everything but the weight / scale requests of this phase has landed - among it the activation half
        // that the group reads in its next load phase, and the weight half it reads there (requested two compute phases ago)
        if constexpr (ISSUE_W) {
            if (s_extra)
                kq_wait_vm<3>();
            else
                kq_wait_vm<2>();
        } else {
            kq_wait_vm<0>(); // (the tail of the slice: no weight requests behind the activation ones)
        }
        if constexpr (stamp)
            BNB_KQ_STAMP(4)
        kq_barrier();
        if constexpr (stamp) {
            BNB_KQ_STAMP(8)
#ifdef BNB_PROFILING
            ts[11] += ts[4] - ts[6]; // cycles waiting for own DMA, summed over the slice
            ts[9] += ts[6] - ts[5];  // cycles of compute-phase work, summed over the slice
            ts[12] += ts[8] - ts[4]; // cycles at the barrier that ends the compute phases, summed over the slice
#endif
        }
    };

    auto load_a2 = [&]() __attribute__((always_inline)) {
        if constexpr (NESTED) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int cl = 64 * c + 32 * nt + n;
                int col = col0 + cl;
                col = col < N ? col : N - 1;
                const uint32_t t0 = static_cast<uint32_t>(col) * static_cast<uint32_t>(K >> 6) + static_cast<uint32_t>(cb) * 4u; // first block of the slice in this column
                a2_lo[nt] = kq_lds_read<float>(static_cast<uint32_t>(L::A2T + cl * 4));
                a2_hi[nt] = kq_lds_read<float>(static_cast<uint32_t>(L::A2T + 512 + cl * 4));
                a2_jx[nt] = static_cast<int>((256u * ((t0 >> 8) + 1u) - t0) >> 2); // (K a multiple of 256: both terms are multiples of 4)
            }
        }
    };

    // ---- the phases. Phase t: group t & 1 computes half t and requests weight / scale half t + 4 (slot t % 4, which it has
    // emptied in phase t - 1; needed at the end of phase t + 2), the other group loads half t + 1 and requests activation half
    // t + 3 (slot t % 3, emptied in phase t - 1 by the first group; needed at the end of phase t + 1). Each wavefront waits for
    // its own requests, in order, ONCE per iteration: at the end of its compute phase, everything but that phase's own
    // weight / scale requests. One s_barrier per phase.
    if constexpr ((ABL & 256) != 0)
        if (g == 1)
            __builtin_amdgcn_s_setprio(1); // the second-dispatched half of the workgroup loses every arbitration against its SIMD partner otherwise
    using cT = std::true_type;
    using cF = std::false_type;
    using cSt = std::integral_constant<bool, (ABL & 64) != 0>; // (the stamped instance of the measurement build stamps every iteration: the last one gets stored)
    if (g == 0) {
        // barrier -2: half 0 has landed (own part; everybody's behind the barrier), the table is written
        kq_wait_vm<BATCH>();
        kq_barrier();
        BNB_KQ_STAMP(7)
        load_a2();
        // (the tail is peeled: the last half of a group requests nothing, the last two no weights - u + 2 >= H, u + 4 >= H)
        int sa = 0, sw = 0, ss = 0; // slots of half 2 j, of chunk j's scales
        auto advance = [&]() {
            sa = sa + 2 >= DA ? sa + 2 - DA : sa + 2;
            sw = (sw + 2) & (DW - 1);
            ss = ss + 1 == L::DS ? 0 : ss + 1;
        };
        // half 0: activation half 2 was requested at start-up
        if (nc > 2)
            one_half(cF{}, cT{}, 0, sa, sw, ss, cSt{});
        else
            one_half(cF{}, cF{}, 0, sa, sw, ss, cSt{});
        advance();
        int j = 1;
        for (; j < nc - 2; ++j) {
            one_half(cT{}, cT{}, 2 * j, sa, sw, ss, cSt{});
            advance();
        }
        for (; j < nc - 1; ++j) {
            one_half(cT{}, cF{}, 2 * j, sa, sw, ss, cSt{});
            advance();
        }
        for (; j < nc; ++j) {
            one_half(cF{}, cF{}, 2 * j, sa, sw, ss, cSt{});
            advance();
        }
        kq_barrier(); // phase 2 nc - 1: group 1 computes the last half
    } else {
        // barrier -2: this group's scale (and second-level) requests have landed - the other group's wavefronts read these columns
        // in the phase behind the barrier; [A1 W1][W3] may fly on
        kq_wait_vm<AI + 4>();
        kq_barrier();
        BNB_KQ_STAMP(7)
        load_a2();
        // phase -1 (group 0 loads half 0): nothing to do but to make sure half 1 has landed
        kq_wait_vm<2>();
        kq_barrier();
        int sa = 1, sw = 1, ss = 0; // slots of half 2 j + 1, of chunk j's scales
        auto advance = [&]() {
            sa = sa + 2 >= DA ? sa + 2 - DA : sa + 2;
            sw = (sw + 2) & (DW - 1);
            ss = ss + 1 == L::DS ? 0 : ss + 1;
        };
        int j = 0;
        for (; j < nc - 2; ++j) {
            one_half(cT{}, cT{}, 2 * j + 1, sa, sw, ss, cSt{});
            advance();
        }
        for (; j < nc - 1; ++j) {
            one_half(cT{}, cF{}, 2 * j + 1, sa, sw, ss, cSt{});
            advance();
        }
        for (; j < nc; ++j) {
            one_half(cF{}, cF{}, 2 * j + 1, sa, sw, ss, cSt{});
            advance();
        }
    }
    // (tile 1 of the last half)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        scale_fma(1, mt, 0, 16, sc1p, part1[mt]);
    BNB_KQ_STAMP(14)

    // ---- the four K quarters of a column group, added in a fixed order (q = 0, 1, 2, 3). The wavefront's 32 MT accumulator
    // registers are cut into four sets of 8 MT; wavefront (c, o) owns set o: it receives that set from the other three
    // quarters through the LDS (everything the loop used is dead: outstanding DMA of past-the-end chunks is drained first),
    // adds in the order q = 0..3 and stores. Area: [c][owner][3 sources][2 MT chunks of 16 B][64 lanes].
    kq_wait_vm<0>();
    kq_barrier();
    constexpr int RS = 8 * MT; // registers of a set
    constexpr int CH = RS / 4; // 16-byte chunks of a set per lane
    // Stores go through a buffer descriptor whose range is the M x N matrix: a row past the end of the batch is out of range by
    // construction (rows are the major dimension), a column past N gets an out-of-range lane offset - no branch per element.
    const int ncol0 = col0 + 64 * c + n;
    const uint32_t esz = hot_kslices == 1 ? static_cast<uint32_t>(sizeof(T)) : 4u;
    uint32_t st_lane[2];
    float bv[2];
    {
        const T* const bias = static_cast<const T*>(p.bias);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int ncol = ncol0 + 32 * nt;
            const bool ok = ncol < N;
            bv[nt] = (bias && hot_kslices == 1 && ok) ? static_cast<float>(bias[ncol]) : 0.0f;
            st_lane[nt] = ok ? (static_cast<uint32_t>(m_base + 4 * h) * static_cast<uint32_t>(N) + static_cast<uint32_t>(ncol)) * esz : 0x80000000u;
        }
    }
    const uint32_t row_bytes = static_cast<uint32_t>(N) * esz;
    // K slices across workgroups: the slab is NOT a row-major [M][N] matrix but the accumulator layout itself - per workgroup
    // tile [column group][tile nt * MT + mt][register quad j][lane] 16 bytes (rows 8 j + 4 h + 0..3 of column n): a lane's four
    // consecutive registers leave as ONE 16-byte store, 1 KiB contiguous per wavefront instruction (the row-major form needs four
    // 4-byte stores of two 128-byte row segments each: the store tail of the kernel was bound by the number of store
    // instructions the CU's address unit takes, 128 per workgroup). gemm4_finalize_kq_kernel below reads that layout.
    const long slab_floats = static_cast<long>(gridDim.z) * gridDim.x * (4096 * MT);
    void* const st_base = hot_kslices == 1
                              ? p.out
                              : static_cast<void*>(p.ws + static_cast<long>(blockIdx.y) * slab_floats +
                                                   (static_cast<long>(blockIdx.z) * gridDim.x + blockIdx.x) * (4096 * MT));
    const auto rs_o = __builtin_amdgcn_make_buffer_rsrc(st_base, 0, hot_kslices == 1 ? static_cast<int>(static_cast<uint32_t>(M) * row_bytes) : 16384 * MT, 0x00020000);
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        if (q != o) {
            const int src = q < o ? q : q - 1;
            unsigned char* const dst = smem + ((((c * 4 + o) * 3 + src) * CH) * 64 + lane) * 16;
#pragma unroll
            for (int ch = 0; ch < CH; ++ch) {
                f32x4 v;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int f = o * RS + ch * 4 + k; // flat register: tile (f / 16) = nt * MT + mt, register f % 16
                    v[k] = acc[(f / 16) / MT][(f / 16) % MT][f % 16];
                }
                *reinterpret_cast<f32x4*>(dst + ch * 1024) = v;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        if (q == o) {
#pragma unroll
            for (int ch = 0; ch < CH; ++ch) {
                f32x4 x[3];
#pragma unroll
                for (int src = 0; src < 3; ++src)
                    x[src] = *reinterpret_cast<const f32x4*>(smem + ((((c * 4 + o) * 3 + src) * CH + ch) * 64 + lane) * 16);
                f32x4 v4;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int f = o * RS + ch * 4 + k;
                    const int nt = (f / 16) / MT, mt = (f / 16) % MT, i = f % 16;
                    const float own = acc[nt][mt][i];
                    // canonical order q = 0, 1, 2, 3 with the owner's value at position o
                    float v = o == 0 ? own : x[0][k];
#pragma unroll
                    for (int qq2 = 1; qq2 < 4; ++qq2)
                        v += qq2 == o ? own : x[qq2 < o ? qq2 : qq2 - 1][k];
                    v4[k] = v;
                    if (hot_kslices == 1) {
                        // 32x32 accumulator layout: register i of lane (n, h) = row (i & 3) + 8 (i >> 2) + 4 h, column n
                        const int mrel = 32 * mt + (i & 3) + 8 * (i >> 2);
                        const uint32_t voff = st_lane[nt] + static_cast<uint32_t>(mrel) * row_bytes;
                        const T tv = static_cast<T>(v + bv[nt]);
                        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, tv), rs_o, voff, 0, 0);
                    }
                }
                if (hot_kslices != 1) {
                    const int f0 = o * RS + ch * 4; // tile f0 / 16, register quad (f0 % 16) / 4
                    const uint32_t voff = static_cast<uint32_t>(lane) * 16u + static_cast<uint32_t>(((f0 / 16) * 4 + (f0 % 16) / 4) * 1024);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v4), rs_o, voff, static_cast<uint32_t>(c) * (8192u * MT),
                                                           2 /* nt: read once, by another launch */);
                }
            }
        }
    }
    BNB_KQ_STAMP(15)
#ifdef BNB_PROFILING
    if constexpr ((ABL & 64) != 0)
    if (p.dbg && lane == 0) {
        unsigned long long* const dst = p.dbg + ((static_cast<long>(blockIdx.z) * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * kKqWaves * 16 + wave * 16;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            dst[i] = ts[i];
    }
#endif
}

// out = T(sum over the K slices + bias) from slabs in the accumulator layout the kernel above stores (see its epilogue): one
// thread per 16-byte piece = rows row0 .. row0 + 3 of one column; slices added in slice order (bit-reproducible); the rows and
// columns of a ragged tile are dropped here.
template <typename T, int BATCH>
__global__ __launch_bounds__(256) void gemm4_finalize_kq_kernel(const float* __restrict__ ws, const T* __restrict__ bias, T* __restrict__ out,
                                                                long slab_floats, int M, int N, int gx, int mt, int kslices) {
    const long idx = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
    if (idx * 4 >= slab_floats)
        return;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    // (all loads of a batch in flight before the first add - gemm4_finalize_kernel)
    for (int s0 = 0; s0 < kslices; s0 += BATCH) {
        f32x4 w[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            const int sl = (s0 + j < kslices) ? s0 + j : kslices - 1;
            w[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ws + static_cast<long>(sl) * slab_floats + idx * 4));
        }
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            const bool live = s0 + j < kslices;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                v[k] = live ? v[k] + w[j][k] : v[k];
        }
    }
    const int lane = static_cast<int>(idx & 63);
    long r = idx >> 6;
    const int j = static_cast<int>(r & 3);
    r >>= 2;
    const int t = static_cast<int>(r % (2 * mt));
    r /= (2 * mt);
    const int c = static_cast<int>(r & 1);
    const long wg = r >> 1;
    const int bx = static_cast<int>(wg % gx), bz = static_cast<int>(wg / gx);
    const int col = bx * kKqCols + 64 * c + 32 * (t / mt) + (lane & 31);
    const int row0 = bz * 32 * mt + 32 * (t % mt) + 8 * j + 4 * (lane >> 5);
    if (col >= N)
        return;
    const float b = bias ? static_cast<float>(bias[col]) : 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (row0 + k < M)
            out[static_cast<long>(row0 + k) * N + col] = static_cast<T>(v[k] + b);
}

template <typename T>
void kq_finalize(const float* ws, const void* bias, void* out, long slab_floats, int M, int N, int gx, int mt, int kslices, hipStream_t stream) {
    const dim3 grid(static_cast<unsigned>((slab_floats / 4 + 255) / 256));
    const T* const b = static_cast<const T*>(bias);
    T* const o = static_cast<T*>(out);
    if (kslices <= 2)
        hipLaunchKernelGGL((gemm4_finalize_kq_kernel<T, 2>), grid, dim3(256), 0, stream, ws, b, o, slab_floats, M, N, gx, mt, kslices);
    else if (kslices <= 4)
        hipLaunchKernelGGL((gemm4_finalize_kq_kernel<T, 4>), grid, dim3(256), 0, stream, ws, b, o, slab_floats, M, N, gx, mt, kslices);
    else
        hipLaunchKernelGGL((gemm4_finalize_kq_kernel<T, 8>), grid, dim3(256), 0, stream, ws, b, o, slab_floats, M, N, gx, mt, kslices);
}

struct KqPlan {
    int mt, ks, cps;
};

// floats of one K slice's slab: whole workgroup tiles (128 columns x 32 MT rows), in the accumulator layout
long kq_slab_floats(int M, int N, int mt) {
    const long gx = (N + kKqCols - 1) / kKqCols, gz = (M + 32 * mt - 1) / (32 * mt);
    return gx * gz * 4096L * mt;
}

// Row tiles, K slices and chunks per slice: a pure function of (M, N, K) and the forced slice count, shared by the launch
// and the workspace-size query. One workgroup per CU (~150 KiB of LDS): K slices fill the chip without spilling into a second
// round of workgroups; every slice keeps at least two chunks so the rings have something to overlap.
KqPlan kq_plan(int M, int N, int K, int force_ks) {
    KqPlan pl;
    pl.mt = M > 32 ? 2 : 1;
    const int chunks = K / kKqChunk;
    const int gx = (N + kKqCols - 1) / kKqCols;
    const int gz = (M + 32 * pl.mt - 1) / (32 * pl.mt);
    const int cus = device_cu_count_or_default();
    int ks = force_ks > 0 ? force_ks : cus / (gx * gz);
    const int max_ks = chunks / 2 > 0 ? chunks / 2 : 1;
    ks = ks > max_ks ? max_ks : ks;
    const int min_ks = (chunks + 63) / 64; // a slice is at most 256 blocks of 64 k: a column's second-level absmax (one per 256 blocks) changes at most once inside it
    ks = ks < min_ks ? min_ks : ks;
    ks = ks < 1 ? 1 : ks;
    pl.cps = (chunks + ks - 1) / ks;
    pl.ks = (chunks + pl.cps - 1) / pl.cps; // every slice non-empty
    return pl;
}

template <typename T, int MT, bool NESTED, bool SFAST, int ABL = 0>
void kq_launch_one(const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax8, int M, int N, int K, int flags,
                   const KqPlan& pl, const KqArgs& a, hipStream_t stream) {
    dim3 grid((N + kKqCols - 1) / kKqCols, pl.ks, (M + 32 * MT - 1) / (32 * MT));
    auto kern = gemm4_mfma_kq_kernel<T, MT, NESTED, SFAST, ABL>;
    static LdsLimit lim;
    constexpr int lds = KqLds<MT, NESTED>::Alloc;
    ensure_dynamic_lds(lim, reinterpret_cast<const void*>(kern), lds);
    hipLaunchKernelGGL(kern, grid, dim3(kKqWaves * 64), lds, stream, A, B, absmax, absmax8, M, N, K, flags, pl.cps, pl.ks, a);
}

template <typename T>
void kq_launch(const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax8, int M, int N, int K, int flags,
               bool sfast, const KqPlan& pl, const KqArgs& a, hipStream_t stream) {
#ifdef BNB_PROFILING
    if constexpr (std::is_same<T, bf16>::value) {
        // ablated / variant instances (bf16, two row tiles, fast scale form): see the kernel
        const int abl = flags >> 16;
        flags &= 0xFFFF;
        if (abl != 0 && sfast && pl.mt == 2) {
            switch (abl) {
#define BNB_KQ_ABL(v)                                                                              \
    case v:                                                                                        \
        if (absmax8 != nullptr)                                                                    \
            return kq_launch_one<T, 2, true, true, v>(A, B, absmax, absmax8, M, N, K, flags, pl, a, stream); \
        return kq_launch_one<T, 2, false, true, v>(A, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
                BNB_KQ_ABL(64) BNB_KQ_ABL(2) BNB_KQ_ABL(3) BNB_KQ_ABL(16) BNB_KQ_ABL(19) BNB_KQ_ABL(44) BNB_KQ_ABL(63) BNB_KQ_ABL(2048)
#undef BNB_KQ_ABL
            default:
                break;
            }
        }
    }
#endif
    flags &= 0xFFFF;
    if (absmax8 != nullptr) {
        // (nested statistics: the fast scale form only - gemm_4bit_kq_serves)
        if (pl.mt == 1)
            kq_launch_one<T, 1, true, true>(A, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
        else
            kq_launch_one<T, 2, true, true>(A, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
    } else if (sfast) {
        if (pl.mt == 1)
            kq_launch_one<T, 1, false, true>(A, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
        else
            kq_launch_one<T, 2, false, true>(A, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
    } else {
        if (pl.mt == 1)
            kq_launch_one<T, 1, false, false>(A, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
        else
            kq_launch_one<T, 2, false, false>(A, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
    }
}

// the per-chunk, per-column scale requests (one 16-byte / 4-byte piece per column): blocksize 64 and aligned statistics
bool kq_scales_fast(const float* absmax, const uint8_t* absmax8, int blocksize) {
    if (blocksize != 64)
        return false;
    return absmax8 != nullptr ? (aligned_to(absmax8, 4) && aligned_to(absmax, 4)) : aligned_to(absmax, 16);
}

} // namespace

// Preconditions: 16-bit activations, literal code tables, K a multiple of 256, blocksize >= 64 (a K quarter of 64 k stays
// inside one quantization block), 16-byte aligned A and B; 32-bit element indices / byte offsets in the buffer loads.
bool gemm_4bit_kq_supported(int dtype, const void* A, const uint8_t* B, const float* code16, int M, int N, int K, int blocksize) {
    const long long nk = static_cast<long long>(N) * K, mk = static_cast<long long>(M) * K;
    return (dtype == 1 || dtype == 2) && code16 == nullptr && M >= 1 && N >= 1 && K >= kKqChunk && (K % kKqChunk) == 0 &&
           blocksize >= 64 && is_pow2(blocksize) && aligned_to(A, 16) && aligned_to(B, 16) && nk < (1LL << 31) && mk < (1LL << 30) &&
           static_cast<long long>(M) * N < (1LL << 29);
}

// Nested (double-quantised) statistics are served in the fast scale form only: blocksize 64 (Linear4bit's default), a 4-byte
// aligned code array, and K <= 16384 (a K slice never longer than 256 blocks, whatever the workspace allows - see kq_plan);
// anything else keeps the producer/consumer kernel.
bool gemm_4bit_kq_serves(const float* absmax, const uint8_t* absmax8, int blocksize, int K) {
    return absmax8 == nullptr || (kq_scales_fast(absmax, absmax8, blocksize) && K <= 64 * kKqChunk);
}

size_t gemm_4bit_kq_workspace_bytes(int M, int N, int K, int force_ks) {
    if (M < 1 || N < 1 || K < kKqChunk)
        return 0;
    const KqPlan pl = kq_plan(M, N, K, force_ks);
    return pl.ks > 1 ? static_cast<size_t>(pl.ks) * kq_slab_floats(M, N, pl.mt) * sizeof(float) : 0;
}

// dtype: 1 = f16, 2 = bf16. force_ks (0 = built-in choice): sweeps and tests.
void gemm_4bit_kq(int dtype, const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax8,
                  const float* absmax_code, const float* absmax_offset, void* out, const void* bias, int M, int N, int K,
                  int blocksize, int quant_type, void* workspace, size_t workspace_bytes, int force_ks, int ablate, hipStream_t stream) {
    g_last_gemm_kernel = kKernelKq;
    KqPlan pl = kq_plan(M, N, K, force_ks);
    float* ws = static_cast<float*>(workspace);
    const size_t slab = static_cast<size_t>(kq_slab_floats(M, N, pl.mt)) * sizeof(float);
    if (pl.ks > 1) {
        if (ws == nullptr) {
            ws = gemm_4bit_internal_workspace(slab * pl.ks, stream);
            workspace_bytes = ws ? slab * pl.ks : 0;
        }
        if (workspace_bytes < slab * pl.ks) {
            const int fit = static_cast<int>(workspace_bytes / slab);
            const int chunks = K / kKqChunk;
            const int ks = fit >= 2 ? fit : 1;
            pl.cps = (chunks + ks - 1) / ks;
            pl.ks = (chunks + pl.cps - 1) / pl.cps;
        }
    }
    KqArgs a;
#ifdef BNB_PROFILING
    a.dbg = g_dbg_buf;
#endif
    a.absmax_code = absmax_code;
    a.absmax_offset = absmax_offset;
    a.out = out;
    a.bias = bias;
    a.ws = ws;
    int flags = ilog2(blocksize) | ((quant_type == kFP4) ? 256 : 0);
#ifdef BNB_PROFILING
    flags |= (ablate & 0x7FFF) << 16;
#else
    (void)ablate;
#endif
    const bool sfast = kq_scales_fast(absmax, absmax8, blocksize);
    if (dtype == 2)
        kq_launch<bf16>(A, B, absmax, absmax8, M, N, K, flags, sfast, pl, a, stream);
    else
        kq_launch<f16>(A, B, absmax, absmax8, M, N, K, flags, sfast, pl, a, stream);
    BNB_CHECK_LAUNCH();
    if (pl.ks > 1) {
        const int gx = (N + kKqCols - 1) / kKqCols;
        if (dtype == 2)
            kq_finalize<bf16>(ws, bias, out, kq_slab_floats(M, N, pl.mt), M, N, gx, pl.mt, pl.ks, stream);
        else
            kq_finalize<f16>(ws, bias, out, kq_slab_floats(M, N, pl.mt), M, N, gx, pl.mt, pl.ks, stream);
        BNB_CHECK_LAUNCH();
    }
}

} // namespace bnb
