// gemm4_grad_input.hip — fused backward of the 4-bit linear layer on gfx950:
//     grad_A[m, k] = sum_n grad_out[m, n] * T(code[B[n, k]] * scale[n, k / bs])          (bf16 / fp16)
//
// The reference computes this as dequantize_4bit(B) -> [N, K] in T, then a dense matmul (autograd/_functions.py:365-386):
// (0.5 + 2 + 2) bytes per weight through HBM against 0.56 when the dequantized weights never leave the CU. Here they never
// even leave the registers: every weight is decoded with the arithmetic of dequantize_4bit - fp32 product, ONE rounding to
// T - straight into the MFMA operand, so the result equals the unfused path up to the order of the fp32 sums.
//
// Why this is not the forward kernel with the operands swapped: the contraction runs over n, the weight ROW index, while
// the packed format keeps k contiguous and the scale is indexed by (n, k / bs) - it varies along the contraction, so it
// cannot be applied to a partial tile after the matrix instruction and has to be folded into the operand. The MFMA B
// operand wants, per lane, 8 consecutive n at one k: a COLUMN of the row-major weights.
//
// The first version of this kernel dequantized a [64 n][128 k] tile into LDS and read it back transposed
// (ds_read_b64_tr_b16); its timeline (profiles/r2_timeline_bwd.txt) showed it bound by LDS bandwidth: 136 KB through LDS
// per 4 KB of packed weights. This version needs no transposition at all:
//
//  * lane (c, g) of a wavefront (c = lane % 16, g = lane / 16) loads ONE DWORD (8 k: columns 8c .. 8c+7) from each of the 8
//    weight rows n0 + 8g .. n0 + 8g + 7 - a quad of lanes reads 16 contiguous bytes, a wavefront-wide load 4 rows x 64 B.
//    Nibble j of the lane's 8 dwords IS the B operand of an MFMA whose 16 columns are {8c + j}: 8 consecutive n at one k,
//    already in the lane that needs them. The column tiles are strided (j, j + 8, ...), which costs nothing: lane c holds
//    columns 8c .. 8c+7 of an output row in its 8 accumulators and stores them as one contiguous piece.
//  * decode: one v_perm_b32 + one ds_read_b64 per byte through the bank-private byte -> (code[hi], code[lo]) fp32 table,
//    a packed multiply by the row's scale, one convert per (row pair, column): 8 B of LDS traffic per packed byte, nothing else.
//  * a wavefront owns a whole 32-n block x 128 k x 64 batch rows: 32 MFMAs per block with 128 accumulator registers; its
//    grad_out block [64 m][32 n] (coalesced loads) and its 64 scales pass through a PRIVATE LDS patch on their way to the
//    fragment layout - no barrier anywhere in the main loop, the 8 wavefronts of a workgroup run decoupled.
//  * the wavefronts of a workgroup take different n blocks of the workgroup's N slice; their partial tiles are added through
//    LDS in a fixed order at the end. N slices across workgroups fill the chip; fp32 slabs are added in slice order by
//    gemm4_finalize: bit-reproducible.
#include "bnb_common.h"

#include <atomic>

namespace bnb {

#ifdef BNB_PROFILING
extern unsigned long long* g_dbg_buf;
#endif

// gemm4_mfma.hip
void gemm_4bit_finalize(int dtype, const float* ws, const void* bias, void* out, int M, int N, int kslices, hipStream_t stream);

namespace {

using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <typename T> struct GiMma;
template <> struct GiMma<bf16> {
    using frag = __attribute__((ext_vector_type(8))) bf16;
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(frag, a), __builtin_bit_cast(frag, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint32_t pack(float first, float second) {
        using V = __attribute__((ext_vector_type(2))) bf16;
        V v;
        v[0] = static_cast<bf16>(first);
        v[1] = static_cast<bf16>(second);
        return __builtin_bit_cast(uint32_t, v);
    }
};
template <> struct GiMma<f16> {
    using frag = __attribute__((ext_vector_type(8))) f16;
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(frag, a), __builtin_bit_cast(frag, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint32_t pack(float first, float second) {
        using V = __attribute__((ext_vector_type(2))) f16;
        V v;
        v[0] = static_cast<f16>(first);
        v[1] = static_cast<f16>(second);
        return __builtin_bit_cast(uint32_t, v);
    }
};

constexpr int kGiCols = 128;      // k-columns per workgroup = per wavefront (16 lanes x 8)
constexpr int kGiMaxRows = 64;    // batch rows per workgroup: 16 MT, MT = 1, 2 or 4 MFMA row tiles (template parameter)
constexpr int kGiBlockN = 32;     // n per wavefront block (one MFMA reduction: 4 lane groups x 8)
constexpr int kGiWaves = 8;       // wavefronts per workgroup (a 4-wavefront variant with a deeper ring measured the same within noise)
constexpr int kGiLut = 65536;     // 256 entries x 32 copies x 8 B (fp32 pair), at LDS address 0
constexpr int kGiGPatch = kGiMaxRows * kGiBlockN * 2; // one wavefront's grad_out block [<= 64 m][32 n] in T: 4 KiB
constexpr int kGiSPatch = 256;    // ... and its scales [2 halves of the 128 columns][32 rows] fp32
constexpr int gi_acc_bytes(int mt) { return 16 * mt * kGiCols * 4; } // one wavefront's fp32 partial tile: 8 KiB per row tile
// LDS: table | nested code (1 KiB) | per-wavefront patches; the final reduction reuses everything from address 0
constexpr int gi_lds_bytes(int mt) {
    const int loop = kGiLut + 1024 + kGiWaves * (kGiGPatch + kGiSPatch);
    return (4 * gi_acc_bytes(mt) > loop) ? 4 * gi_acc_bytes(mt) : loop;
}

#ifdef BNB_PROFILING
#define BNB_GI_STAMP(i)                                                                            \
    {                                                                                              \
        if (p.dbg && lane == 0)                                                                    \
            p.dbg[((((static_cast<long>(blockIdx.z) * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8) + wave) * 16 + (i)] = \
                __builtin_amdgcn_s_memtime();                                                      \
    }
#else
#define BNB_GI_STAMP(i) {}
#endif

struct GiArgs {
#ifdef BNB_PROFILING
    unsigned long long* dbg;
#endif
    const float* absmax_code;
    const float* absmax_offset;
    void* out;
    float* ws; // fp32 [nslices][M][K] partial slabs when nslices > 1
};

__device__ __forceinline__ float gi_code_literal(int i, bool fp4) {
    constexpr float nf4[16] = {BNB_NF4_VALUES};
    constexpr float fp4v[16] = {BNB_FP4_VALUES};
    float v = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j)
        v = (i == j) ? (fp4 ? fp4v[j] : nf4[j]) : v;
    return v;
}

// grid = (K / 128, nslices, ceil(M / (16 MT))); 512 threads
template <typename T, bool NESTED, int MT>
__global__ __launch_bounds__(kGiWaves * 64) void gemm4_grad_input_kernel(
    const void* hot_G, const uint8_t* hot_B, const float* hot_absmax, const uint8_t* hot_absmax8, int hot_M, int hot_N,
    int hot_K, int hot_flags /* bs_shift | fp4 << 8 */, int hot_bps /* 32-n blocks per N slice */, int hot_nslices, const GiArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = hot_M, N = hot_N, K = hot_K;
    const int bs_shift = hot_flags & 31;
    const bool fp4 = (hot_flags >> 8) & 1;
    BNB_GI_STAMP(0)
    const int k0 = blockIdx.x * kGiCols;
    constexpr int kGiAccBytes = gi_acc_bytes(MT);
    const int m_base = blockIdx.z * (16 * MT);
    const int blocks_total = N / kGiBlockN;
    const int bb = blockIdx.y * hot_bps;
    int be = bb + hot_bps;
    be = be < blocks_total ? be : blocks_total;
    // this wavefront's blocks: bb + wave, bb + wave + 8, ... < be
    static_assert(kGiWaves == 8, "the final reduction is written for 8 partial tiles");
    const int nb = (be - bb - wave + kGiWaves - 1) / kGiWaves > 0 ? (be - bb - wave + kGiWaves - 1) / kGiWaves : 0;
    const int last_blk = be - 1;
    auto block_of = [&](int it) -> int { // clamped: a prefetch past the end re-reads the slice's last block, never used
        const int b = bb + wave + it * kGiWaves;
        return b < last_blk ? b : last_blk;
    };

    float* const code2 = reinterpret_cast<float*>(smem + kGiLut);
    unsigned char* const gpatch = smem + kGiLut + 1024 + wave * (kGiGPatch + kGiSPatch);
    float* const spatch = reinterpret_cast<float*>(gpatch + kGiGPatch);

    const int c = lane & 15, g = lane >> 4;
    // ---- loads of a block. Weights: row n0 + 8 g + i, the dword of columns 8 c .. 8 c + 7. Scale: lane l fetches the scale of
    // (row n0 + l / 2, 64-column half l & 1). grad_out: row 16 t + l / 4 of the batch tile, 16-byte piece l & 3 (8 n).
    const uint8_t* const wsrc = hot_B + static_cast<long>(8 * g) * (K >> 1) + ((k0 + 8 * c) >> 1);
    const long se0 = static_cast<long>(lane >> 1) * K + k0 + 64 * (lane & 1);
    const T* gsrc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        int row = m_base + 16 * t + (lane >> 2);
        row = row < M ? row : M - 1; // rows past the end of the batch re-read the last row: never stored
        gsrc[t] = static_cast<const T*>(hot_G) + static_cast<long>(row) * N + 8 * (lane & 3);
    }
    struct WStage {
        uint32_t w[8];
        uint32_t s, s2;
    };
    auto issue_w = [&](WStage& st, int it) {
        const long n0 = static_cast<long>(block_of(it)) * kGiBlockN;
        const uint8_t* src = wsrc + n0 * (K >> 1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            st.w[i] = *reinterpret_cast<const uint32_t*>(src + static_cast<long>(i) * (K >> 1));
        const long e = (se0 + n0 * K) >> bs_shift;
        if constexpr (NESTED) {
            st.s = hot_absmax8[e];
            st.s2 = __builtin_bit_cast(uint32_t, hot_absmax[e >> 8]);
        } else {
            st.s = __builtin_bit_cast(uint32_t, hot_absmax[e]);
            st.s2 = 0;
        }
    };
    auto issue_g = [&](u32x4 (&gr)[MT], int it) {
        const long n0 = static_cast<long>(block_of(it)) * kGiBlockN;
#pragma unroll
        for (int t = 0; t < MT; ++t)
            gr[t] = *reinterpret_cast<const u32x4*>(gsrc[t] + n0);
    };

    // weights: a D-deep register ring (block it + D is requested when block it has been decoded); grad_out: requested one
    // block ahead, right after the previous block's registers went to LDS. Only the first block's loads go out before the
    // decode table is built (every load instruction costs the CU's address pipeline, and the table build of the other
    // wavefronts waits at the barrier for the slowest issuer); the rest of the ring follows after the barrier.
    constexpr int D = (MT == 4) ? 2 : 3; // (three stages spill at MT = 4: 2 wavefronts per SIMD = 256 registers per lane, 128 of them accumulators)
    WStage st[D];
    u32x4 gr[MT];
    if (be <= bb)
        return; // (never: the host makes every N slice non-empty)
    issue_w(st[0], 0);
    issue_g(gr, 0);
    BNB_GI_STAMP(1)
    __builtin_amdgcn_sched_barrier(0);

    // ---- decode table, built while the first loads fly: entry e (a packed byte) = 32 copies of (code[e >> 4], code[e & 15])
    // in fp32, 256 B per entry; thread (half, e) writes 8 of its 16 chunks in an order rotated by e (eight lanes -> eight
    // bank quads)
    {
        const float cv = gi_code_literal((lane & 15) + opaque_zero(), fp4);
        const int cvb = __builtin_bit_cast(int, cv);
        constexpr int PER = 16 * 256 / (kGiWaves * 64); // 16-byte chunks of an entry per thread
        const int e = tid & 255, part = tid >> 8;
        const float hi = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((e >> 4) & 15) * 4, cvb));
        const float lo = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((e & 15) * 4, cvb));
        const f32x4 v = {hi, lo, hi, lo};
        f32x4* const dst = reinterpret_cast<f32x4*>(smem + e * 256 + part * (PER * 16));
#pragma unroll
        for (int j = 0; j < PER; ++j)
            dst[(j + e) & (PER - 1)] = v;
    }
    float offset = 0.0f;
    if constexpr (NESTED) {
        if (tid < 256)
            code2[tid] = p.absmax_code[tid];
        static_assert(kGiWaves * 64 >= 256, "one thread per entry of the nested code");
        offset = p.absmax_offset[0];
    }
    __syncthreads();
    BNB_GI_STAMP(2)
#pragma unroll
    for (int j = 1; j < D; ++j)
        issue_w(st[j], j);
    if (reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)smem) != 0)
        __builtin_trap(); // the table is addressed with raw v_perm_b32 results: it must sit at LDS address 0

    const uint32_t perm_sel = 0x0C0C0400u + static_cast<uint32_t>(opaque_zero()); // {lane offset, weight byte, 0, 0}
    const uint32_t lane_off = static_cast<uint32_t>(lane & 31) * 8u;

    // private grad_out patch: row r (64 B) holds its 16-byte piece q at q ^ f(r), f = {0, 3, 2, 1}[(r / 4) % 4] - conflict-free
    // for the 8-contiguous-lane passes of ds_write_b128 (two rows x four pieces) and for the 16-lane groups {0-3, 12-15,
    // 20-27}, {4-11, 16-19, 28-31}, + 32 of the ds_read_b128 fragment reads (lane (m, g): row 16 mt + m, piece g)
    auto swz = [](int r) -> int { return (4 - ((r >> 2) & 3)) & 3; };
    const int wrow = lane >> 2, wpiece = lane & 3;
    const uint32_t g_wr = static_cast<uint32_t>(wrow * 64 + ((wpiece ^ swz(wrow)) << 4));  // + t * 1024
    const uint32_t g_rd = static_cast<uint32_t>(c * 64 + ((g ^ swz(c)) << 4));             // + mt * 1024
    const uint32_t s_wr = static_cast<uint32_t>(((lane & 1) * 32 + (lane >> 1)) * 4);
    const uint32_t s_rd = static_cast<uint32_t>(((c >> 3) * 32 + 8 * g) * 4);

    f32x4 acc[MT][8];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            acc[mt][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // one 32-n block: ISSUE = whether later blocks are prefetched (off in the peeled tail)
    auto do_block = [&](WStage& ws, int it) {
        // grad_out block and scales: registers -> private patch (LDS operations of one wavefront execute in order: the
        // fragment reads of the previous block are done before these writes land)
#pragma unroll
        for (int t = 0; t < MT; ++t)
            *reinterpret_cast<u32x4*>(gpatch + t * 1024 + g_wr) = gr[t];
        {
            float scale;
            if constexpr (NESTED) {
                const uint32_t q = ws.s, a2 = ws.s2;
                scale = nested_scale(code2[q & 0xFFu], __builtin_bit_cast(float, a2), offset);
            } else {
                const uint32_t sv = ws.s;
                scale = __builtin_bit_cast(float, sv);
            }
            *reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(spatch) + s_wr) = scale;
        }
        issue_g(gr, it + 1);
        if (it < 3)
            BNB_GI_STAMP(3 + 3 * it)
        const f32x4 s_lo = *reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned char*>(spatch) + s_rd);
        const f32x4 s_hi = *reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned char*>(spatch) + s_rd + 16);
        const float sc[8] = {s_lo[0], s_lo[1], s_lo[2], s_lo[3], s_hi[0], s_hi[1], s_hi[2], s_hi[3]};
        u32x4 af[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            af[mt] = *reinterpret_cast<const u32x4*>(gpatch + mt * 1024 + g_rd);
        if (it < 3)
            BNB_GI_STAMP(4 + 3 * it)
        // byte b of the lane's 8 dwords = columns 8 c + 2 b (high nibbles) and 8 c + 2 b + 1 (low nibbles) of rows 8 g .. 8 g + 7
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            f32x2 pr[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                pr[i] = *reinterpret_cast<const __attribute__((address_space(3))) f32x2*>(
                    __builtin_amdgcn_perm(ws.w[i], lane_off, perm_sel + (b << 8)));
            u32x4 bx, by;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                // the reference's dequantize rounds the fp32 product to T once (csrc/cpu_ops.cpp:419-431). bf16 has no fused
                // multiply-convert on gfx950, so the plain expression already is "product in fp32, then one rounding"; for fp16
                // hipcc would fuse it into v_fma_mix*_f16 (one rounding of the exact product): an opaque (non-volatile: it may
                // be scheduled freely) register copy keeps the two steps apart
                f32x2 p0 = pr[2 * d] * f32x2{sc[2 * d], sc[2 * d]};
                f32x2 p1 = pr[2 * d + 1] * f32x2{sc[2 * d + 1], sc[2 * d + 1]};
                if constexpr (sizeof(T) == 2 && !__is_same(T, bf16)) {
                    float a0 = p0[0], a1 = p0[1], b0 = p1[0], b1 = p1[1];
                    asm("" : "+v"(a0));
                    asm("" : "+v"(a1));
                    asm("" : "+v"(b0));
                    asm("" : "+v"(b1));
                    p0 = f32x2{a0, a1};
                    p1 = f32x2{b0, b1};
                }
                bx[d] = GiMma<T>::pack(p0[0], p1[0]);
                by[d] = GiMma<T>::pack(p0[1], p1[1]);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                acc[mt][2 * b] = GiMma<T>::run(af[mt], bx, acc[mt][2 * b]);
                acc[mt][2 * b + 1] = GiMma<T>::run(af[mt], by, acc[mt][2 * b + 1]);
            }
        }
        if (it < 3)
            BNB_GI_STAMP(5 + 3 * it)
        issue_w(ws, it + D);
    };
    // whole rounds of the ring with nothing conditional around the loads (a branch around a load makes the compiler merge
    // the pending-load state of both paths at the join and wait with vmcnt(0)), then the tail
    int it = 0;
    for (; it + D <= nb; it += D) {
#pragma unroll
        for (int j = 0; j < D; ++j)
            do_block(st[j], it + j);
    }
#pragma unroll
    for (int j = 0; j < D - 1; ++j)
        if (it + j < nb)
            do_block(st[j], it + j);

    BNB_GI_STAMP(12)
    // ---- the partial tiles of the workgroup, added in a fixed order: 8 wavefronts: p_w = tile_w + tile_{w+4} (w = 0..3) first;
    // then (p_0 + p_1) + (p_2 + p_3). A tile in LDS: accumulator register group r = 8 mt + j of lane l at r * 1024 + l * 16.
    __syncthreads(); // every wavefront is done with the table and its patches
    {
        const uint32_t lane16 = static_cast<uint32_t>(lane) * 16u;
        if (wave >= 4) {
            unsigned char* const dst = smem + (wave - 4) * kGiAccBytes + lane16;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    *reinterpret_cast<f32x4*>(dst + (8 * mt + j) * 1024) = acc[mt][j];
        }
        __syncthreads();
        if (wave < 4) {
            unsigned char* const buf = smem + wave * kGiAccBytes + lane16;
            // (eight reads in flight before the first add: written element by element the compiler waits for every read)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x4 o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    o[j] = *reinterpret_cast<const f32x4*>(buf + (8 * mt + j) * 1024);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    *reinterpret_cast<f32x4*>(buf + (8 * mt + j) * 1024) = acc[mt][j] + o[j];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        // the 2 MT units (row tile mt, column half jh: columns 8 c + 4 jh .. + 3 of every lane) are dealt to the wavefronts;
        // lane (c, g) holds rows 16 mt + 4 g + q of those columns
#pragma unroll
        for (int u0 = 0; u0 < 2 * MT; u0 += kGiWaves) {
        const int u = u0 + wave;
        if (u >= 2 * MT)
            break;
        const int mt = u >> 1, jh = u & 1;
        f32x4 v[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const uint32_t o = static_cast<uint32_t>((8 * mt + 4 * jh + jj) * 1024) + lane16;
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(smem + 0 * kGiAccBytes + o);
            const f32x4 t1 = *reinterpret_cast<const f32x4*>(smem + 1 * kGiAccBytes + o);
            const f32x4 t2 = *reinterpret_cast<const f32x4*>(smem + 2 * kGiAccBytes + o);
            const f32x4 t3 = *reinterpret_cast<const f32x4*>(smem + 3 * kGiAccBytes + o);
            v[jj] = (t0 + t1) + (t2 + t3);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = m_base + 16 * mt + 4 * g + q;
            if (m < M) {
                const long idx = static_cast<long>(m) * K + k0 + 8 * c + 4 * jh;
                if (hot_nslices == 1) {
                    using T4 = __attribute__((ext_vector_type(4))) T;
                    T4 o;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
                        o[jj] = static_cast<T>(v[jj][q]);
                    *reinterpret_cast<T4*>(static_cast<T*>(p.out) + idx) = o;
                } else {
                    *reinterpret_cast<f32x4*>(p.ws + static_cast<long>(blockIdx.y) * M * K + idx) = f32x4{v[0][q], v[1][q], v[2][q], v[3][q]};
                }
            }
        }
        }
    }
    BNB_GI_STAMP(13)
}

int gi_cu_count() { return device_cu_count_or_default(); }

struct GiPlan {
    int ns, bps; // N slices, 32-n blocks per slice
    int mt;      // MFMA row tiles per workgroup (16 batch rows each)
};
int gi_row_tiles(int M) { return M <= 16 ? 1 : M <= 32 ? 2 : 4; }
GiPlan gi_slices(int blocks, int ns, int mt) {
    GiPlan pl;
    pl.mt = mt;
    ns = ns < 1 ? 1 : ns;
    pl.bps = (blocks + ns - 1) / ns;
    pl.ns = (blocks + pl.bps - 1) / pl.bps;
    return pl;
}
// N slices to fill the chip (one workgroup per CU), at least two blocks per wavefront each
thread_local TlsKnob g_gi_force_ns{0}; // sweeps (bnb_mi355x_set_tuning reserved1): N slices, 0 = built-in choice
GiPlan gi_plan(int M, int N, int K) {
    const int blocks = N / kGiBlockN;
    const int mt = gi_row_tiles(M);
    const int wgs = (K / kGiCols) * ((M + 16 * mt - 1) / (16 * mt));
    int ns = gi_cu_count() / wgs;
    const int max_ns = blocks / (2 * kGiWaves) > 0 ? blocks / (2 * kGiWaves) : 1;
    ns = ns > max_ns ? max_ns : ns;
    const int force = g_gi_force_ns.load(std::memory_order_relaxed);
    if (force > 0)
        ns = force < blocks ? force : blocks;
    return gi_slices(blocks, ns, mt);
}

template <typename T, bool NESTED, int MT>
void gi_launch_mt(const void* G, const uint8_t* B, const float* absmax, const uint8_t* absmax8, int M, int N, int K, int flags,
                  const GiPlan& pl, const GiArgs& a, hipStream_t stream) {
    dim3 grid(K / kGiCols, pl.ns, (M + 16 * MT - 1) / (16 * MT));
    auto kern = gemm4_grad_input_kernel<T, NESTED, MT>;
    static LdsLimit lim;
    ensure_dynamic_lds(lim, reinterpret_cast<const void*>(kern), gi_lds_bytes(MT));
    hipLaunchKernelGGL(kern, grid, dim3(kGiWaves * 64), gi_lds_bytes(MT), stream, G, B, absmax, absmax8, M, N, K, flags, pl.bps, pl.ns, a);
}
template <typename T, bool NESTED>
void gi_launch_one(const void* G, const uint8_t* B, const float* absmax, const uint8_t* absmax8, int M, int N, int K, int flags,
                   const GiPlan& pl, const GiArgs& a, hipStream_t stream) {
    if (pl.mt == 1)
        gi_launch_mt<T, NESTED, 1>(G, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
    else if (pl.mt == 2)
        gi_launch_mt<T, NESTED, 2>(G, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
    else
        gi_launch_mt<T, NESTED, 4>(G, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
}
template <typename T> void gi_launch(const void* G, const uint8_t* B, const float* absmax, const uint8_t* absmax8, int M, int N, int K,
                                     int flags, const GiPlan& pl, const GiArgs& a, hipStream_t stream) {
    if (absmax8 != nullptr)
        gi_launch_one<T, true>(G, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
    else
        gi_launch_one<T, false>(G, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
}

} // namespace

void gemm_4bit_grad_input_set_slices(int ns) { g_gi_force_ns.store(ns, std::memory_order_relaxed); }

constexpr int kGiNAlign = 64; // N % 64 == 0 (whole 32-n blocks, 16-byte aligned grad_out rows with room to spare)

// Preconditions of the fused backward: 16-bit gradients, whole 128-column tiles and 32-n blocks, blocksize >= 64.
bool gemm_4bit_grad_input_supported(int dtype, const void* G, const uint8_t* B, int M, int N, int K, int blocksize) {
    return (dtype == 1 || dtype == 2) && M >= 1 && N >= kGiNAlign && (N % kGiNAlign) == 0 && K >= kGiCols && (K % kGiCols) == 0 &&
           blocksize >= 64 && is_pow2(blocksize) && aligned_to(G, 16) && aligned_to(B, 16);
}

size_t gemm_4bit_grad_input_workspace_bytes(int M, int N, int K) {
    if (M < 1 || N < kGiNAlign || K < kGiCols)
        return 0;
    const GiPlan pl = gi_plan(M, N, K);
    return pl.ns > 1 ? static_cast<size_t>(pl.ns) * M * K * sizeof(float) : 0;
}

// grad_A[M, K] = grad_out[M, N] * dequant(B)[N, K]. dtype 1 = f16, 2 = bf16. The workspace (fp32 slabs, see
// gemm_4bit_grad_input_workspace_bytes) is required when the plan uses more than one N slice; with a smaller one the launch
// uses as many slices as fit.
void gemm_4bit_grad_input(int dtype, const void* G, const uint8_t* B, const float* absmax, const uint8_t* absmax8,
                          const float* absmax_code, const float* absmax_offset, void* out, int M, int N, int K, int blocksize,
                          int quant_type, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0)
        return;
    if (!gemm_4bit_grad_input_supported(dtype, G, B, M, N, K, blocksize)) {
        fprintf(stderr, "bitsandbytes_amd: gemm_4bit_grad_input: unsupported problem (need fp16/bf16, N %% 64 == 0, K %% 128 == 0, "
                        "blocksize >= 64, 16-byte aligned pointers); M=%d N=%d K=%d blocksize=%d\n", M, N, K, blocksize);
        exit(1);
    }
    GiPlan pl = gi_plan(M, N, K);
    const size_t slab = static_cast<size_t>(M) * K * sizeof(float);
    if (pl.ns > 1 && (workspace == nullptr || workspace_bytes < slab * pl.ns)) {
        const int fit = workspace ? static_cast<int>(workspace_bytes / slab) : 0;
        pl = gi_slices(N / kGiBlockN, fit >= 2 ? fit : 1, pl.mt);
    }
    GiArgs a;
#ifdef BNB_PROFILING
    a.dbg = g_dbg_buf;
#endif
    a.absmax_code = absmax_code;
    a.absmax_offset = absmax_offset;
    a.out = out;
    a.ws = static_cast<float*>(workspace);
    const int flags = ilog2(blocksize) | ((quant_type == kFP4) ? 256 : 0);
    if (dtype == 2)
        gi_launch<bf16>(G, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
    else
        gi_launch<f16>(G, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
    BNB_CHECK_LAUNCH();
    if (pl.ns > 1)
        gemm_4bit_finalize(dtype, a.ws, nullptr, out, M, K, pl.ns, stream);
}

} // namespace bnb
