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
    static constexpr int AHB = 32 * MT * 256;          // bytes of an activation half-slot: 32 MT rows x 128 k
    static constexpr int WSB = kKqCols * 128;          // bytes of a weight slot: 128 rows x 256 k
    static constexpr int SSB = NESTED ? 4096 : 2048;   // bytes of a scale slot: [128 columns][4 blocks] dwords (nested: 8-bit codes in dword slots, then second-level absmax)
    static constexpr int ABase = kKqLut;
    static constexpr int WBase = ABase + 3 * AHB;
    static constexpr int SBase = WBase + 2 * WSB;
    static constexpr int Code2 = SBase + 2 * SSB;
    static constexpr int Bytes = Code2 + 1024;
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

#ifdef BNB_PROFILING
#define BNB_KQ_STAMP(i)                                                                            \
    {                                                                                              \
        if (p.dbg && lane == 0)                                                                    \
            p.dbg[((static_cast<long>(blockIdx.z) * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * kKqWaves * 16 + wave * 16 + (i)] = \
                __builtin_amdgcn_s_memtime();                                                      \
    }
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
template <typename V> __device__ __forceinline__ V kq_lds_read(uint32_t addr) {
    return *reinterpret_cast<const __attribute__((address_space(3))) V*>(addr);
}

// grid = (ceil(N / 128), kslices, ceil(M / (32 MT))); 512 threads.
template <typename T, int MT, bool NESTED>
__global__ __launch_bounds__(kKqWaves * 64) void gemm4_mfma_kq_kernel(
    // hot arguments as separate scalars: preloaded into SGPRs by the command processor (14 dwords)
    const void* hot_A, const uint8_t* hot_B, const float* hot_absmax, const uint8_t* hot_absmax8, int hot_M, int hot_N,
    int hot_K, int hot_flags /* bs_shift | fp4 << 8 */, int hot_cps /* chunks per K slice */, int hot_kslices,
    const KqArgs p) {
    using L = KqLds<MT, NESTED>;
    constexpr int SI = NESTED ? 2 : 1;       // scale-side DMA instructions per wavefront and chunk
    constexpr int TAIL = 2 + SI + MT;        // DMA instructions a wavefront issues BEHIND the urgent activation half of an iteration
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    BNB_KQ_STAMP(0)
    const int c = wave >> 2, q = wave & 3;  // column group, K quarter (= quantization block of the chunk); wavefronts q and 4 + q share a SIMD
    const int n = lane & 31, h = lane >> 5; // MFMA roles: column / row n, k half h
    const int M = hot_M, N = hot_N, K = hot_K;
    const int bs_shift = hot_flags & 31;
    const bool fp4 = (hot_flags >> 8) & 1;
    const int col0 = blockIdx.x * kKqCols;
    const int m_base = blockIdx.z * (32 * MT);
    const int chunks_total = K >> 8;
    const int cb = blockIdx.y * hot_cps;
    int ce = cb + hot_cps;
    ce = ce < chunks_total ? ce : chunks_total;
    const int nc = ce - cb; // chunks of this slice (>= 1: the host makes every slice non-empty)
    if (reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)smem) != 0)
        __builtin_trap(); // the table is addressed with raw v_perm_b32 results: it must sit at LDS address 0

    // ---- sources. Rows past the end (ragged N or M) re-read the last row: MFMA rows / columns are independent and those
    // results are never stored. All byte offsets are < 2^31 (gemm_4bit_kq_supported).
    const i32x4 rs_w = kq_rsrc(hot_B), rs_a = kq_rsrc(hot_A), rs_s = kq_rsrc(hot_absmax), rs_q = kq_rsrc(hot_absmax8);
    uint32_t lo_a[MT], lo_w[2], lo_s;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        // activations: instruction i = wave + 8 t of a half covers rows 4 i .. 4 i + 3 (256 B each); lane 16 r + p' writes LDS
        // piece p' of its row and fetches piece p' ^ (row & 15)
        const int row = 4 * (wave + 8 * t) + (lane >> 4);
        int m = m_base + row;
        m = m < M ? m : M - 1;
        lo_a[t] = (static_cast<uint32_t>(m) * static_cast<uint32_t>(K)) * 2u + static_cast<uint32_t>(((lane & 15) ^ (row & 15)) << 4);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        // weights: instruction i = wave + 8 t covers rows 8 i .. 8 i + 7 (128 B each); lane 8 r + p' writes LDS piece p' and
        // fetches piece p' ^ ((row >> 1) & 7)
        const int row = 8 * (wave + 8 * t) + (lane >> 3);
        int col = col0 + row;
        col = col < N ? col : N - 1;
        lo_w[t] = static_cast<uint32_t>(col) * static_cast<uint32_t>(K >> 1) + static_cast<uint32_t>(((lane & 7) ^ ((row >> 1) & 7)) << 4);
    }
    {
        // scales: the wavefront's instruction covers columns 16 wave .. 16 wave + 15, lane 4 r + i = (column r, block i of the
        // chunk): element index of the block's first weight
        int col = col0 + 16 * wave + (lane >> 2);
        col = col < N ? col : N - 1;
        lo_s = static_cast<uint32_t>(col) * static_cast<uint32_t>(K) + static_cast<uint32_t>(64 * (lane & 3));
    }
    const uint32_t lane4p = static_cast<uint32_t>(lane) * 4u;
    // chunk j of the slice (absolute chunk cb + j; past the end: the last one again, into a slot nobody reads any more)
    auto issue_a = [&](int j, int hh, int slot) {
        const uint32_t ca = static_cast<uint32_t>(cb + (j < nc ? j : nc - 1));
#pragma unroll
        for (int t = 0; t < MT; ++t)
            kq_dma16(rs_a, static_cast<uint32_t>(L::ABase + slot * L::AHB + (wave + 8 * t) * 1024), lo_a[t], ca * 512u + static_cast<uint32_t>(hh) * 256u);
    };
    auto issue_w = [&](int j, int slot) {
        const uint32_t ca = static_cast<uint32_t>(cb + (j < nc ? j : nc - 1));
#pragma unroll
        for (int t = 0; t < 2; ++t)
            kq_dma16(rs_w, static_cast<uint32_t>(L::WBase + slot * L::WSB + (wave + 8 * t) * 1024), lo_w[t], ca * 128u);
    };
    auto issue_s = [&](int j, int slot) {
        const uint32_t ca = static_cast<uint32_t>(cb + (j < nc ? j : nc - 1));
        const uint32_t blk = (lo_s + ca * 256u) >> bs_shift;
        if constexpr (NESTED) {
            kq_dma1(rs_q, static_cast<uint32_t>(L::SBase + slot * L::SSB + wave * 256), blk);
            kq_dma4(rs_s, static_cast<uint32_t>(L::SBase + slot * L::SSB + 2048 + wave * 256), (blk >> 8) * 4u);
        } else {
            kq_dma4(rs_s, static_cast<uint32_t>(L::SBase + slot * L::SSB + wave * 256), blk * 4u);
        }
    };

    // ---- start-up: chunk 0 (both activation halves, weights, scales), then what of chunk 1 has a free slot; the decode
    // table is built while they fly
    if constexpr (NESTED) {
        // (the second-level code table, 256 floats: one dword DMA by each of the first four wavefronts - the oldest entry of
        // their queues, covered by every later counted wait; an ordinary load here would make the compiler drain the queue)
        if (wave < 4)
            kq_dma4(kq_rsrc(p.absmax_code), static_cast<uint32_t>(L::Code2 + wave * 256), lane4p + static_cast<uint32_t>(wave) * 256u);
    }
    issue_a(0, 0, 0);
    issue_a(0, 1, 1);
    issue_w(0, 0);
    issue_s(0, 0);
    issue_w(1, 1);
    issue_s(1, 1);
    issue_a(1, 0, 2);
    BNB_KQ_STAMP(1)
    float offset = 0.0f;
    if constexpr (NESTED)
        offset = p.absmax_offset[0];
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

    const uint32_t perm_sel = 0x0C0C0400u + static_cast<uint32_t>(opaque_zero()); // {lane offset, weight byte, 0, 0}
    const uint32_t lane4 = static_cast<uint32_t>(lane) * 4u;
    // consumer addresses, relative to slot 0
    //   packed weights: row 64 c + lane, 16-byte pieces 2 q and 2 q + 1 of its 128 B, stored at piece ^ ((row >> 1) & 7)
    const uint32_t w_rd = static_cast<uint32_t>(L::WBase + (64 * c + lane) * 128 + (((2 * q) ^ ((lane >> 1) & 7)) << 4));
    //   scale of column 64 c + 32 nt + n, block q
    uint32_t s_rd[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
        s_rd[nt] = static_cast<uint32_t>(L::SBase + ((64 * c + 32 * nt + n) * 4 + q) * 4);
    //   activation fragment of step s, row tile mt: row 32 mt + n of the half q >> 1, piece 8 (q & 1) + 4 h + s, stored at piece ^ (n & 15)
    uint32_t a_rd[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
        a_rd[s] = static_cast<uint32_t>(L::ABase + n * 256 + (((8 * (q & 1) + 4 * h + s) ^ (n & 15)) << 4));

    f32x16 acc[2][MT];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                acc[nt][mt][i] = 0.0f;

    int a0 = 0, a1 = 1; // half-slots of the current chunk's activation halves
    int wsl = 0;        // slot of the current chunk's weights / scales
    for (int j = 0; j < nc; ++j) {
        if (j == 2 || j == 3)
            BNB_KQ_STAMP(2 + 5 * (j - 2))
        // ---- barrier 1: chunk j is in the LDS (own DMA landed up to the urgent half of the previous iteration; everybody's, behind the barrier)
        kq_wait_vm<TAIL>();
        if (j == 2 || j == 3)
            BNB_KQ_STAMP(3 + 5 * (j - 2))
        kq_barrier();
        // ---- the wavefront's share of the chunk moves to registers
        const uint32_t wso = static_cast<uint32_t>(wsl * L::WSB), sso = static_cast<uint32_t>(wsl * L::SSB);
        const uint32_t aso = static_cast<uint32_t>(((q >> 1) ? a1 : a0) * L::AHB);
        const u32x4 wraw0 = kq_lds_read<u32x4>(w_rd + wso);
        const u32x4 wraw1 = kq_lds_read<u32x4>((w_rd ^ 16u) + wso);
        float sc[2];
        uint32_t q8[2];
        float a2[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            if constexpr (NESTED) {
                q8[nt] = kq_lds_read<uint32_t>(s_rd[nt] + sso);
                a2[nt] = kq_lds_read<float>(s_rd[nt] + sso + 2048);
            } else {
                sc[nt] = kq_lds_read<float>(s_rd[nt] + sso);
            }
        }
        u32x4 afr[4][MT];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                afr[s][mt] = kq_lds_read<u32x4>(a_rd[s] + aso + mt * 8192);
        if constexpr (NESTED) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const float c2 = kq_lds_read<float>(static_cast<uint32_t>(L::Code2) + (q8[nt] & 0xFFu) * 4u);
                sc[nt] = __fadd_rn(__fmul_rn(c2, a2[nt]), offset);
            }
        }
        // ---- barrier 2: every wavefront holds its share: the slots of chunk j are free
        kq_barrier();
        if (j == 2 || j == 3)
            BNB_KQ_STAMP(4 + 5 * (j - 2))
        // the urgent request first: the second activation half of chunk j + 1 (it has one chunk of time to land; L2 hits)
        issue_a(j + 1, 1, a0);

        // ---- deal the packed weights: lane (n, h) gets dword 4 h + s of columns 64 c + n (tile 0) and 64 c + 32 + n (tile 1):
        // the high dwords of the lower half are swapped with the low dwords of the upper half
        uint32_t wt[2][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const auto sw = __builtin_amdgcn_permlane32_swap(wraw0[r], wraw1[r], false, false);
            wt[0][r] = sw[0];
            wt[1][r] = sw[1];
        }
        auto lut4 = [&](uint32_t w) {
            u32x4 r;
#pragma unroll
            for (int b = 0; b < 4; ++b)
                r[b] = kq_lds_read<uint32_t>(__builtin_amdgcn_perm(w, lane4, perm_sel + (b << 8)));
            return r;
        };
        // ---- 8 fragments (tile nt, step s), two look-ups ahead of the MFMAs; the rest of the DMA requests ride between them
        f32x16 part[2][MT];
        u32x4 bfr[3];
        bfr[0] = lut4(wt[0][0]);
        bfr[1] = lut4(wt[0][1]);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int nt = g >> 2, s = g & 3;
            if (g + 2 < 8)
                bfr[(g + 2) % 3] = lut4(wt[(g + 2) >> 2][(g + 2) & 3]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (s == 0) {
                    f32x16 z;
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        z[i] = 0.0f;
                    part[nt][mt] = KqMma<T>::run(afr[s][mt], bfr[g % 3], z);
                } else {
                    part[nt][mt] = KqMma<T>::run(afr[s][mt], bfr[g % 3], part[nt][mt]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // DMA requests of the iteration, one group per step
            if (g == 0)
                issue_w(j + 2, wsl);
            if (g == 1)
                issue_s(j + 2, wsl);
            if (g == 2)
                issue_a(j + 2, 0, a1);
            // tile 0 is complete after g = 3: its partial tiles are scaled and accumulated under the MFMAs of tile 1
            if (nt == 1) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int i = 4 * s; i < 4 * s + 4; ++i)
                        acc[0][mt][i] = fmaf(sc[0], part[0][mt][i], acc[0][mt][i]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                acc[1][mt][i] = fmaf(sc[1], part[1][mt][i], acc[1][mt][i]);
        if (j == 2 || j == 3)
            BNB_KQ_STAMP(5 + 5 * (j - 2))
        // next chunk: half-slots advance by two (mod 3), the weight / scale slot alternates
        a0 = a0 + 2 >= 3 ? a0 - 1 : a0 + 2;
        a1 = a1 + 2 >= 3 ? a1 - 1 : a1 + 2;
        wsl ^= 1;
    }
    BNB_KQ_STAMP(14)

    // ---- the four K quarters of a column group, added in a fixed order (q = 0, 1, 2, 3). The wavefront's 32 MT accumulator
    // registers are cut into four sets of 8 MT; wavefront (c, o) owns set o: it receives that set from the other three
    // quarters through the LDS (everything the loop used is dead: outstanding DMA of past-the-end chunks is drained first),
    // adds in the order q = 0..3 and stores. Area: [c][owner][3 sources][2 MT chunks of 16 B][64 lanes].
    kq_wait_vm<0>();
    kq_barrier();
    constexpr int RS = 8 * MT; // registers of a set
    constexpr int CH = RS / 4; // 16-byte chunks of a set per lane
    bool col_ok[2];
    float bv[2];
    {
        const T* const bias = static_cast<const T*>(p.bias);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int ncol = col0 + 64 * c + 32 * nt + n;
            col_ok[nt] = ncol < N;
            bv[nt] = (bias && hot_kslices == 1 && col_ok[nt]) ? static_cast<float>(bias[ncol]) : 0.0f;
        }
    }
    const long out_lane = static_cast<long>(m_base + 4 * h) * N + col0 + 64 * c + n;
    float* const ws_slab = p.ws + static_cast<long>(blockIdx.y) * M * N;
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
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int f = o * RS + ch * 4 + k;
                    const int nt = (f / 16) / MT, mt = (f / 16) % MT, i = f % 16;
                    const float own = acc[nt][mt][i];
                    // canonical order q = 0, 1, 2, 3 with the owner's value at position o
                    float v = o == 0 ? own : x[0][k];
#pragma unroll
                    for (int qq = 1; qq < 4; ++qq)
                        v += qq == o ? own : x[qq < o ? qq : qq - 1][k];
                    // 32x32 accumulator layout: register i of lane (n, h) = row (i & 3) + 8 (i >> 2) + 4 h, column n
                    const int mrel = 32 * mt + (i & 3) + 8 * (i >> 2);
                    if (m_base + 4 * h + mrel < M && col_ok[nt]) {
                        const long o2 = out_lane + static_cast<long>(mrel) * N + 32 * nt;
                        if (hot_kslices == 1)
                            static_cast<T*>(p.out)[o2] = static_cast<T>(v + bv[nt]);
                        else
                            __builtin_nontemporal_store(v, &ws_slab[o2]);
                    }
                }
            }
        }
    }
    BNB_KQ_STAMP(15)
}

struct KqPlan {
    int mt, ks, cps;
};

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
    ks = ks < 1 ? 1 : ks;
    pl.cps = (chunks + ks - 1) / ks;
    pl.ks = (chunks + pl.cps - 1) / pl.cps; // every slice non-empty
    return pl;
}

template <typename T, int MT, bool NESTED>
void kq_launch_one(const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax8, int M, int N, int K, int flags,
                   const KqPlan& pl, const KqArgs& a, hipStream_t stream) {
    dim3 grid((N + kKqCols - 1) / kKqCols, pl.ks, (M + 32 * MT - 1) / (32 * MT));
    auto kern = gemm4_mfma_kq_kernel<T, MT, NESTED>;
    static LdsLimit lim;
    constexpr int lds = KqLds<MT, NESTED>::Alloc;
    ensure_dynamic_lds(lim, reinterpret_cast<const void*>(kern), lds);
    hipLaunchKernelGGL(kern, grid, dim3(kKqWaves * 64), lds, stream, A, B, absmax, absmax8, M, N, K, flags, pl.cps, pl.ks, a);
}

template <typename T>
void kq_launch(const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax8, int M, int N, int K, int flags,
               const KqPlan& pl, const KqArgs& a, hipStream_t stream) {
    if (absmax8 != nullptr) {
        if (pl.mt == 1)
            kq_launch_one<T, 1, true>(A, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
        else
            kq_launch_one<T, 2, true>(A, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
    } else {
        if (pl.mt == 1)
            kq_launch_one<T, 1, false>(A, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
        else
            kq_launch_one<T, 2, false>(A, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
    }
}

} // namespace

// Preconditions: 16-bit activations, literal code tables, K a multiple of 256, blocksize >= 64 (a K quarter of 64 k stays
// inside one quantization block), 16-byte aligned A and B; 32-bit element indices / byte offsets in the buffer loads.
bool gemm_4bit_kq_supported(int dtype, const void* A, const uint8_t* B, const float* code16, int M, int N, int K, int blocksize) {
    const long long nk = static_cast<long long>(N) * K, mk = static_cast<long long>(M) * K;
    return (dtype == 1 || dtype == 2) && code16 == nullptr && M >= 1 && N >= 1 && K >= kKqChunk && (K % kKqChunk) == 0 &&
           blocksize >= 64 && is_pow2(blocksize) && aligned_to(A, 16) && aligned_to(B, 16) && nk < (1LL << 31) && mk < (1LL << 30);
}

size_t gemm_4bit_kq_workspace_bytes(int M, int N, int K, int force_ks) {
    if (M < 1 || N < 1 || K < kKqChunk)
        return 0;
    const KqPlan pl = kq_plan(M, N, K, force_ks);
    return pl.ks > 1 ? static_cast<size_t>(pl.ks) * M * N * sizeof(float) : 0;
}

// dtype: 1 = f16, 2 = bf16. force_ks (0 = built-in choice): sweeps and tests.
void gemm_4bit_kq(int dtype, const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax8,
                  const float* absmax_code, const float* absmax_offset, void* out, const void* bias, int M, int N, int K,
                  int blocksize, int quant_type, void* workspace, size_t workspace_bytes, int force_ks, hipStream_t stream) {
    KqPlan pl = kq_plan(M, N, K, force_ks);
    float* ws = static_cast<float*>(workspace);
    const size_t slab = static_cast<size_t>(M) * N * sizeof(float);
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
    const int flags = ilog2(blocksize) | ((quant_type == kFP4) ? 256 : 0);
    if (dtype == 2)
        kq_launch<bf16>(A, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
    else
        kq_launch<f16>(A, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
    BNB_CHECK_LAUNCH();
    if (pl.ks > 1)
        gemm_4bit_finalize(dtype, ws, bias, out, M, N, pl.ks, stream);
}

} // namespace bnb
