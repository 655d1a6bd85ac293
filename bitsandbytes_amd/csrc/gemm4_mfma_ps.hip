// gemm4_mfma_ps.hip — "pre-scaled operand" MFMA kernel for batched 4-bit linear layers on gfx950 (round 3):
//     out[m, n] = sum_k A[m, k] * T(code[B[n, k]] * scale[n, k / bs])  (+ bias[n])          T in {bf16, fp16}
//
// Fills, on MI355X, the tensor-core capability the reference has on CUDA only (csrc/gemm_4bit_sm80.cu:127-457,493-689) and
// uses that kernel's arithmetic: every weight is decoded to T(code * scale) - fp32 product, ONE rounding to T, exactly
// csrc/gemm_4bit_sm80.cu:300-307 and exactly what dequantize_4bit produces (csrc/cpu_ops.cpp:419-431) - straight into the MFMA
// B operand, so the result equals dequantize_4bit + a T matmul with fp32 accumulation up to the order of the fp32 sums.
// The round-1/2 kernels (gemm4_mfma.hip, gemm4_mfma_rt.hip) multiplied bf16-rounded codes and applied the fp32 scale to
// the partial tile of every 64-k block with VALU FMAs: that pins the accumulators to architectural VGPRs, costs 4 v_fma
// per (tile, block) and batch tile, and spends one address instruction + one ds_read_b32 per weight byte and batch tile
// pair. Here the accumulators are touched by MFMA instructions only and the decode of a chunk is shared by all batch rows.
//
// Shape of the kernel:
//
//  * v_mfma_f32_32x32x16_{bf16,f16}: A operand = activations (row m = lane % 32), B operand = weights (column n = lane % 32),
//    lane half h = lane / 32 holds k = 8 h + 0..7 of the 16-k step. Half the LDS operand traffic per FLOP of the 16x16x32 form.
//  * one workgroup = 128 output columns x (32 MT batch rows, MT = 1 | 2) x one K slice; 8 wavefronts = 4 column groups of 32
//    x 2 K halves. A "stage" is 256 k: K half q works on its own 128-k chunk of it, so every wavefront decodes ONE chunk
//    (32 columns x 128 k = 2 KiB of packed weights) per stage and multiplies it with all 32 MT rows.
//  * PING-PONG. The first two builds of this kernel interleaved decode and MFMA step by step in every wavefront; their PMC
//    passes (profiles/r3_pmc_ps_v2_c3.txt) showed per stage and SIMD 1024 cycles of matrix pipe, ~1340 of VALU and ~1640 of
//    LDS adding up to the ~3900 cycles a stage took: with all wavefronts in the same phase of the same dependent chain
//    nothing overlaps. Now a wavefront alternates between a DECODE phase (the chunk's 32 table look-ups per lane, scale
//    multiplies, converts: VALU + LDS, no matrix instruction; result = the chunk's 8 B fragments in 32 registers) and an MFMA
//    phase (16 back-to-back MFMAs fed by activation fragments from LDS: ~512 cycles of matrix pipe, almost no VALU), and
//    the two K halves run in OPPOSITE phases - wavefronts g and g + 4 share a SIMD - with one s_barrier per phase: while one
//    wavefront of a SIMD owns the matrix pipe, its partner owns the VALU.
//  * EVERY global load of a wavefront - its two weight loads, its share of the activation chunk (MT loads of 8 rows x
//    256 B... see below) and its scale - is an ordinary coalesced BUFFER load (base in SGPRs, one 32-bit lane offset
//    computed once, the chunk as scalar offset: no address arithmetic per load) into a D-deep REGISTER ring; vmcnt retires in
//    order, so the ring is refilled in consumption order and every wait the compiler emits is a counted one.
//    No LDS-DMA, no producer wavefronts, no inline-asm waits.
//  * weights: lane 4 r + p loads 16 bytes of row r (four neighbouring lanes = 64 contiguous bytes: 16 L1 tag look-ups per
//    instruction); the MFMA wants the row in the low lane bits, so the chunk goes through a 2-KiB tile private to the
//    wavefront (ds_write_b128 / ds_read_b128, same wavefront, in-order LDS, no barrier; XOR swizzle conflict-free under the
//    hardware's lane groups). After it lane (n, h) holds the 64 consecutive k [64 h, 64 h + 64) of column n = ONE
//    quantization block (bs >= 64): one scale per lane and chunk; MFMA step s consumes dword s (k = 64 h + 8 s + 0..7) and
//    the activation fragment of the same k - K order inside an MFMA is free as long as both operands agree.
//  * decode per packed byte: v_perm_b32 (LDS address) + ds_read_b64 (bank-private byte -> (code[hi], code[lo]) fp32 table
//    built from literals) + v_pk_mul_f32 by the lane's scale + one convert-and-pack.
//  * activations: the four wavefronts of a K half write their pieces of the half's chunk into the half's LDS buffer (rows of
//    256 B, 16-byte pieces XOR-swizzled by the row: conflict-free ds_write_b128 and ds_read_b128) in their decode phase and
//    read fragments from it in the MFMA phase that follows: one buffer per K half is enough.
//  * the two K halves of a column group are added through LDS (fixed order), K slices across workgroups write fp32 slabs
//    that gemm4_finalize adds in slice order: bit-reproducible.
#include "bnb_common.h"

namespace bnb {

#ifdef BNB_PROFILING
extern unsigned long long* g_dbg_buf;
#endif

// gemm4_mfma.hip
void gemm_4bit_finalize(int dtype, const float* ws, const void* bias, void* out, int M, int N, int kslices, hipStream_t stream);
float* gemm_4bit_internal_workspace(size_t bytes, hipStream_t stream);

namespace {

using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <typename T> struct PsMma;
template <> struct PsMma<bf16> {
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
template <> struct PsMma<f16> {
    using frag = __attribute__((ext_vector_type(8))) f16;
    static __device__ __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(frag, a), __builtin_bit_cast(frag, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint32_t pack(float first, float second) {
        // (the fp32 products are opaque register values here - see the decode loop - so hipcc cannot fuse the multiply into
        // v_fma_mix*_f16, which would round the exact product once instead of fp32 first, T second)
        using V = __attribute__((ext_vector_type(2))) f16;
        V v;
        v[0] = static_cast<f16>(first);
        v[1] = static_cast<f16>(second);
        return __builtin_bit_cast(uint32_t, v);
    }
};

constexpr int kPsCols = 128;        // output columns per workgroup: 4 column groups of 32
constexpr int kPsStageK = 256;      // k per stage: one 128-k chunk per K half
constexpr int kPsWaves = 8;         // 4 column groups x 2 K halves
constexpr int kPsLut = 65536;       // 256 entries x 32 copies x 8 B (fp32 pair), at LDS address 0
constexpr int kPsABuf = 16384;      // the activation chunk of one K half: up to 64 rows x 256 B
constexpr int kPsABase = kPsLut;    // one buffer per K half
constexpr int kPsTileBase = kPsABase + 2 * kPsABuf; // per-wavefront transposition tiles, 2 KiB each
constexpr int kPsCode2 = kPsTileBase + kPsWaves * 2048;
constexpr int kPsLdsBytes = kPsCode2 + 1024;

struct PsArgs {
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
#define BNB_PS_STAMP(i)                                                                            \
    {                                                                                              \
        if (p.dbg && lane == 0)                                                                    \
            p.dbg[((static_cast<long>(blockIdx.z) * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * kPsWaves * 16 + wave * 16 + (i)] = \
                __builtin_amdgcn_s_memtime();                                                      \
    }
#else
#define BNB_PS_STAMP(i) {}
#endif

__device__ __forceinline__ float ps_code_literal(int i, bool fp4) {
    // compare/select over literals: no memory access in front of the table
    constexpr float nf4[16] = {BNB_NF4_VALUES};
    constexpr float fp4v[16] = {BNB_FP4_VALUES};
    float v = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j)
        v = (i == j) ? (fp4 ? fp4v[j] : nf4[j]) : v;
    return v;
}

// grid = (ceil(N / 128), kslices, ceil(M / (32 MT))); 512 threads. D = depth of the register rings in chunks.
template <typename T, int MT, bool NESTED, int D>
__global__ __launch_bounds__(kPsWaves * 64) void gemm4_mfma_ps_kernel(
    // hot arguments as separate scalars: preloaded into SGPRs by the command processor (14 dwords)
    const void* hot_A, const uint8_t* hot_B, const float* hot_absmax, const uint8_t* hot_absmax8, int hot_M, int hot_N,
    int hot_K, int hot_flags /* bs_shift | fp4 << 8 */, int hot_sps /* stages per K slice */, int hot_kslices,
    const PsArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    BNB_PS_STAMP(0)
    const int g = wave & 3, q = wave >> 2;        // column group, K half (wavefronts g and g + 4 share a SIMD)
    const int r = lane >> 2, pp = lane & 3;       // weight-load roles: row r of a 16-row half tile, 16-byte piece pp of its 64 bytes
    const int n = lane & 31, h = lane >> 5;       // MFMA roles: column / row n, k half h
    const int arow = lane >> 4, apiece = lane & 15; // activation-load roles: row arow of a 4-row group, 16-byte piece of its 256 bytes
    const int M = hot_M, N = hot_N, K = hot_K;
    const int bs_shift = hot_flags & 31;
    const bool fp4 = (hot_flags >> 8) & 1;
    const int col0 = blockIdx.x * kPsCols + 32 * g;
    const int m_base = blockIdx.z * (32 * MT);
    const int stages_total = K >> 8;
    const int sb = blockIdx.y * hot_sps;
    int se = sb + hot_sps;
    se = se < stages_total ? se : stages_total;
    const int ns = se - sb; // stages of this slice = chunks of this wavefront (>= 1: the host makes every slice non-empty)
    // K half q owns the chunks [kq, kq + 128 ns) of the slice [256 sb, 256 se): chunk j of the wavefront = k kq + 128 j
    const uint32_t kq = (static_cast<uint32_t>(sb) << 8) + static_cast<uint32_t>(q) * 128u * static_cast<uint32_t>(ns);

    // ---- sources: buffer loads (base in SGPRs, a 32-bit per-lane byte offset computed ONCE, the chunk as a scalar offset):
    // no per-load address arithmetic on the VALU. All byte offsets are < 2^31 (gemm_4bit_ps_supported). Rows past the end
    // (ragged N or M) re-read the last row: MFMA rows / columns are independent and those results are never stored, so no
    // masking instructions are needed.
    constexpr int kRsrcFlags = 0x00020000;
    constexpr int kRecords = 0x7FFFFFFF;
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(hot_B), 0, kRecords, kRsrcFlags);
    const auto rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(hot_A), 0, kRecords, kRsrcFlags);
    const auto rs_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hot_absmax), 0, kRecords, kRsrcFlags);
    // (nested codes are fetched as ALIGNED dwords - see prep_t: the descriptor starts at the aligned address at or below the
    // array, q_mis = the array's offset in it; a shard's absmax view may start at any byte)
    const uint32_t q_mis = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(hot_absmax8) & 3u);
    const auto rs_q = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(hot_absmax8) - q_mis, 0, kRecords, kRsrcFlags);
    uint32_t wo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int row = col0 + 16 * i + r;
        row = row < N ? row : N - 1;
        wo[i] = static_cast<uint32_t>(row) * static_cast<uint32_t>(K >> 1) + (kq >> 1) + static_cast<uint32_t>(pp * 16);
    }
    constexpr int AI = 2 * MT; // activation loads per wavefront and chunk: 4 rows x 256 B each
    uint32_t ao[AI], a_wr[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int row_local = 8 * MT * g + 4 * i + arow;
        int m = m_base + row_local;
        m = m < M ? m : M - 1;
        ao[i] = (static_cast<uint32_t>(m) * static_cast<uint32_t>(K) + kq + static_cast<uint32_t>(8 * apiece)) * 2u;
        a_wr[i] = static_cast<uint32_t>(kPsABase + q * kPsABuf + row_local * 256 + ((apiece ^ (row_local & 15)) << 4));
    }
    // scale of lane (n, h) for chunk j: block of flat element (row n) * K + kq + 128 j + 64 h
    int srow = col0 + n;
    srow = srow < N ? srow : N - 1;
    const uint32_t se0 = static_cast<uint32_t>(srow) * static_cast<uint32_t>(K) + kq + static_cast<uint32_t>(64 * h);

    struct WSlot {
        u32x4 w[2];  // lane (r, pp) holds bytes [16 pp, 16 pp + 16) of the chunk's 64 bytes of rows r, 16 + r
        uint32_t s;  // fp32 absmax bits of the lane's block (nested: of its second-level block)
        uint32_t s8; // nested: the aligned dword of 8-bit absmax codes that holds the block's (see prep_t)
    };
    struct ASlot {
        u32x4 a[AI]; // this wavefront's share of the K half's activation chunk
    };
#ifdef BNB_PROFILING
    // ablations (profiling build only, bnb_mi355x_set_tuning knob0 bits 0 / 1 / 2): every lane fetches the FIRST lane's scale /
    // activation piece / weight piece - one request per load instruction instead of 64 / 16 / 16; results are wrong, the
    // timing tells what that operand's traffic costs
    const int ablate = hot_flags >> 16;
#define BNB_PS_ABL(bit, v) ((ablate & (bit)) ? static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v))) : (v))
#else
#define BNB_PS_ABL(bit, v) (v)
#endif
    auto issue_w = [&](WSlot& x, int j) {
        j = j < ns ? j : ns - 1; // a prefetch past the end re-reads the last chunk: never used, keeps every wait counted
#pragma unroll
        for (int i = 0; i < 2; ++i)
            x.w[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, BNB_PS_ABL(4, wo[i]), j * 64, 0));
        const uint32_t blk = BNB_PS_ABL(1, (se0 + static_cast<uint32_t>(j) * 128u) >> bs_shift);
        if constexpr (NESTED) {
            x.s8 = __builtin_amdgcn_raw_buffer_load_b32(rs_q, (blk + q_mis) & ~3u, 0, 0);
            x.s = __builtin_amdgcn_raw_buffer_load_b32(rs_s, (blk >> 8) * 4u, 0, 0);
        } else {
            x.s = __builtin_amdgcn_raw_buffer_load_b32(rs_s, blk * 4u, 0, 0);
            x.s8 = 0;
        }
    };
    auto issue_a = [&](ASlot& x, int j) {
        j = j < ns ? j : ns - 1;
#pragma unroll
        for (int i = 0; i < AI; ++i)
            x.a[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, BNB_PS_ABL(2, ao[i]), j * 256, 0));
    };

    WSlot ws[D];
    ASlot as[D];
    // ---- start-up, split by K half. Half 0 (whose first decode phase opens the pipeline) requests its ring before anything
    // else and then only waits; half 1 - which runs one phase behind anyway - builds the decode table first (256 threads, one
    // entry each) and requests its ring afterwards. A CU keeps only a few tens of KiB of loads in flight and every further
    // load instruction BLOCKS its wavefront until an older one returns, so whoever issues loads cannot build tables in
    // time: with all eight wavefronts doing both, the table barrier fell ~8800 cycles into the kernel
    // (profiles/r3_timeline_ps_v3.txt).
    float offset = 0.0f;
    if constexpr (NESTED)
        offset = p.absmax_offset[0];
    float* const code2 = reinterpret_cast<float*>(smem + kPsCode2);
    auto issue_ring = [&]() {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            issue_w(ws[j], j);
            issue_a(as[j], j);
            __builtin_amdgcn_sched_barrier(0); // (program order = queue order: the loop's counted waits assume chunk by chunk)
        }
    };
    if (q == 0) {
        issue_ring();
        BNB_PS_STAMP(1)
    } else {
        // decode table: entry e (a packed byte) = 32 copies of (code[e >> 4], code[e & 15]) in fp32, 256 B per entry, its 16
        // chunks written in an order rotated by e (eight lanes -> eight bank quads). Literals only: no load in front of it.
        float code2_v = 0.0f;
        if constexpr (NESTED)
            code2_v = p.absmax_code[tid & 255];
        const float cv = ps_code_literal((lane & 15) + opaque_zero(), fp4);
        const int cvb = __builtin_bit_cast(int, cv);
        const int e = tid & 255;
        const float hi = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((e >> 4) & 15) * 4, cvb));
        const float lo = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((e & 15) * 4, cvb));
        const f32x4 v = {hi, lo, hi, lo};
        f32x4* const dst = reinterpret_cast<f32x4*>(smem + e * 256);
#pragma unroll
        for (int j = 0; j < 16; ++j)
            dst[(j + e) & 15] = v;
        if constexpr (NESTED)
            code2[e] = code2_v;
        BNB_PS_STAMP(1)
        __builtin_amdgcn_sched_barrier(0);
        issue_ring();
    }
    __syncthreads();
    BNB_PS_STAMP(2)
    if (reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)smem) != 0)
        __builtin_trap(); // the table is addressed with raw v_perm_b32 results: it must sit at LDS address 0

    const uint32_t perm_sel = 0x0C0C0400u + static_cast<uint32_t>(opaque_zero()); // {lane offset, weight byte, 0, 0}
    const uint32_t lane_off = static_cast<uint32_t>(lane & 31) * 8u;

    // transposition tile: row R (64 B) keeps its 16-byte piece P at position P ^ ((R >> 2) & 3): the 8 contiguous lanes one
    // ds_write_b128 pass serves (rows 2 i, 2 i + 1 x pieces 0..3) land in 8 different 16-byte positions mod 128 B, the 16-lane
    // groups of ds_read_b128 ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32: sixteen rows distinct mod 16, one piece) in 16
    // different positions mod 256 B
    unsigned char* const tile = smem + kPsTileBase + wave * 2048;
    uint32_t t_wr[2], t_rd[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 16 * i + r;
        t_wr[i] = static_cast<uint32_t>((row * 4 + (pp ^ ((row >> 2) & 3))) * 16);
        t_rd[i] = static_cast<uint32_t>((n * 4 + ((2 * h + i) ^ ((n >> 2) & 3))) * 16);
    }
    // activation fragment of step s, row tile mt: (a_rd ^ (s << 4)) + mt * 8192 (row n of the tile, piece (8 h + s) ^ (n & 15))
    const uint32_t a_rd = static_cast<uint32_t>(kPsABase + q * kPsABuf + n * 256 + (((8 * h) ^ (n & 15)) << 4));

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i)
            acc[mt][i] = 0.0f;

    // packed weights of a ring slot: coalesced shape -> private tile -> MFMA shape; the scale leaves its ring register
    auto prep_w = [&](WSlot& x) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            *reinterpret_cast<u32x4*>(tile + t_wr[i]) = x.w[i];
    };
    auto prep_t = [&](WSlot& x, int jn, u32x4 (&wt)[2], float& scale) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            wt[i] = *reinterpret_cast<const u32x4*>(tile + t_rd[i]);
        // The scale leaves its ring register through an instruction the compiler cannot move: left to itself hipcc parks the
        // copy in the loop latch, where its wait for this one load becomes s_waitcnt vmcnt(0) - a drain of the whole ring once
        // per round (seen in the ISA of the first build). For the same reason the nested code is fetched as the aligned
        // DWORD that holds it and the byte is cut out here: a byte load's zero extension is an instruction of its own, and
        // hipcc hoisted that one to the loop top behind a vmcnt(5).
        uint32_t sv;
        asm volatile("v_mov_b32 %0, %1" : "=v"(sv) : "v"(x.s));
        if constexpr (NESTED) {
            uint32_t qw;
            asm volatile("v_mov_b32 %0, %1" : "=v"(qw) : "v"(x.s8));
            jn = jn < ns ? jn : ns - 1;
            const uint32_t blk = (se0 + static_cast<uint32_t>(jn) * 128u) >> bs_shift;
            const uint32_t qv = __builtin_amdgcn_ubfe(qw, 8u * ((blk + q_mis) & 3u), 8u);
            scale = __fadd_rn(__fmul_rn(code2[qv], __builtin_bit_cast(float, sv)), offset);
        } else {
            scale = __builtin_bit_cast(float, sv);
        }
    };

    u32x4 wt[2];  // the current chunk's packed weights in MFMA shape: dword s of lane (n, h) = k [64 h + 8 s, + 8) of column n
    float scale;
    u32x4 bfr[8]; // ... decoded: the B fragments of its eight 16-k steps
    prep_w(ws[0]);
    prep_t(ws[0], 0, wt, scale);
    __builtin_amdgcn_sched_barrier(0);
    issue_w(ws[0], D);
    __builtin_amdgcn_sched_barrier(0);
    BNB_PS_STAMP(3)
    // K half 1 runs one phase behind K half 0
    if (q == 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    BNB_PS_STAMP(4)

    // ---- DECODE phase of chunk j (ring slot x): its activation pieces -> the K half's LDS buffer (whose readers, the MFMA
    // phase of chunk j - 1, are behind the last barrier), the slot is re-requested for chunk j + D, then the chunk's 32 table
    // look-ups per lane: ALL of them are issued before the first result is used (64 registers that the MFMA phase reuses for
    // its activation fragments) - in batches of eight with one batch in flight the phase paid an LDS round trip per batch
    // (~1400 cycles per phase for ~130 instructions, profiles/r3_timeline_ps_v3.txt).
    auto decode_phase = [&](ASlot& x, int j) {
#pragma unroll
        for (int i = 0; i < AI; ++i)
            *reinterpret_cast<u32x4*>(smem + a_wr[i]) = x.a[i];
        __builtin_amdgcn_sched_barrier(0);
        issue_a(x, j + D);
        __builtin_amdgcn_sched_barrier(0);
        const f32x2 sc2 = {scale, scale};
        f32x2 pr[8][4];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const uint32_t w = (s < 4) ? wt[0][s & 3] : wt[1][s & 3];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                pr[s][c] = *reinterpret_cast<const __attribute__((address_space(3))) f32x2*>(
                    __builtin_amdgcn_perm(w, lane_off, perm_sel + (c << 8)));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x2 pv = pr[s][c] * sc2;
                float p0 = pv[0], p1 = pv[1];
                if constexpr (!__is_same(T, bf16)) {
                    asm("" : "+v"(p0));
                    asm("" : "+v"(p1));
                }
                bfr[s][c] = PsMma<T>::pack(p0, p1);
            }
        __builtin_amdgcn_sched_barrier(0);
    };
    // ---- MFMA phase of chunk j: eight 16-k steps; ALL activation fragments are requested up front (the registers of the
    // decode phase's look-ups), the wavefront runs at raised priority so that its MFMAs issue the moment their operands are
    // there - its SIMD partner is in its decode phase and competes for the same issue port. Between the steps the NEXT
    // chunk's packed weights (ring slot x) go through the private tile and the slot is re-requested for chunk j + 1 + D.
    auto mfma_phase = [&](WSlot& x, int j) {
        u32x4 af[8][MT];
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                af[s][mt] = *reinterpret_cast<const u32x4*>(smem + (a_rd ^ static_cast<uint32_t>(s << 4)) + mt * 8192);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s == 2)
                prep_w(x);
            else if (s == 5)
                prep_t(x, j + 1, wt, scale);
            else if (s == 6)
                issue_w(x, j + 1 + D);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[mt] = PsMma<T>::run(af[s][mt], bfr[s], acc[mt]);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    auto phase_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto do_chunk = [&](ASlot& xa, WSlot& xw_next, int j) {
        decode_phase(xa, j);
        if (j < 2)
            BNB_PS_STAMP(5 + 4 * j)
        phase_barrier(); // the chunk's activations are complete in LDS; the partner half is done with the matrix pipe
        if (j < 2)
            BNB_PS_STAMP(6 + 4 * j)
        mfma_phase(xw_next, j);
        if (j < 2)
            BNB_PS_STAMP(7 + 4 * j)
        phase_barrier(); // everybody is done with this half's activation buffer
        if (j < 2)
            BNB_PS_STAMP(8 + 4 * j)
    };
    // chunk j lives in ring slot j % D: the loop is unrolled by D so that every slot index is a constant. Whole rounds first,
    // with nothing conditional around the loads (at the join of a branch around a load the compiler merges the pending-load
    // state of both paths and waits conservatively), then the tail.
    {
        int j = 0;
        for (; j + D <= ns; j += D) {
#pragma unroll
            for (int jj = 0; jj < D; ++jj)
                do_chunk(as[jj], ws[(jj + 1) % D], j + jj);
        }
#pragma unroll
        for (int jj = 0; jj < D - 1; ++jj)
            if (j + jj < ns)
                do_chunk(as[jj], ws[(jj + 1) % D], j + jj);
    }
    if (q == 0)
        phase_barrier(); // (K half 1's last MFMA phase)
    BNB_PS_STAMP(13)

    // ---- the two K halves of a column group, added in a fixed order (half 0 + half 1); the parking area reuses the
    // activation buffers (every MFMA phase is behind the last barrier): [g][mt][4 register quads][64 lanes x 16 B]
    unsigned char* const red = smem + kPsABase + (g * MT) * 4096 + lane * 16;
    if (q == 1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                *reinterpret_cast<f32x4*>(red + (mt * 4 + r4) * 1024) =
                    f32x4{acc[mt][4 * r4], acc[mt][4 * r4 + 1], acc[mt][4 * r4 + 2], acc[mt][4 * r4 + 3]};
    }
    __syncthreads();
    if (q == 0) {
        const int ncol = col0 + n;
        const T* const bias = static_cast<const T*>(p.bias);
        const float bv = (bias && hot_kslices == 1 && ncol < N) ? static_cast<float>(bias[ncol]) : 0.0f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x4 o[4];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                o[r4] = *reinterpret_cast<const f32x4*>(red + (mt * 4 + r4) * 1024);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float v = acc[mt][i] + o[i >> 2][i & 3];
                // 32x32 accumulator layout: register i of lane (n, h) = row (i & 3) + 8 (i >> 2) + 4 h, column n
                const int m = m_base + 32 * mt + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (m < M && ncol < N) {
                    const long o2 = static_cast<long>(m) * N + ncol;
                    if (hot_kslices == 1)
                        static_cast<T*>(p.out)[o2] = static_cast<T>(v + bv);
                    else
                        p.ws[static_cast<long>(blockIdx.y) * M * N + o2] = v;
                }
            }
        }
    }
    BNB_PS_STAMP(14)
}

struct PsPlan {
    int mt, ks, sps;
};

// Row tiles, K slices and stages per slice: a pure function of (M, N, K) and the forced slice count, shared by the launch
// and the workspace-size query. One workgroup per CU (145 KiB of LDS): K slices fill the chip without spilling into a second
// round of workgroups; every slice keeps at least two stages so the ring has something to overlap.
PsPlan ps_plan(int M, int N, int K, int force_ks) {
    PsPlan pl;
    pl.mt = M > 32 ? 2 : 1;
    const int stages = K / kPsStageK;
    const int gx = (N + kPsCols - 1) / kPsCols;
    const int gz = (M + 32 * pl.mt - 1) / (32 * pl.mt);
    const int cus = device_cu_count_or_default();
    int ks = force_ks > 0 ? force_ks : cus / (gx * gz);
    const int max_ks = stages / 2 > 0 ? stages / 2 : 1;
    ks = ks > max_ks ? max_ks : ks;
    ks = ks < 1 ? 1 : ks;
    pl.sps = (stages + ks - 1) / ks;
    pl.ks = (stages + pl.sps - 1) / pl.sps; // every slice non-empty
    return pl;
}

template <typename T, int MT, bool NESTED, int D>
void ps_launch_one(const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax8, int M, int N, int K, int flags,
                   const PsPlan& pl, const PsArgs& a, hipStream_t stream) {
    dim3 grid((N + kPsCols - 1) / kPsCols, pl.ks, (M + 32 * MT - 1) / (32 * MT));
    auto kern = gemm4_mfma_ps_kernel<T, MT, NESTED, D>;
    static LdsLimit lim;
    ensure_dynamic_lds(lim, reinterpret_cast<const void*>(kern), kPsLdsBytes);
    hipLaunchKernelGGL(kern, grid, dim3(kPsWaves * 64), kPsLdsBytes, stream, A, B, absmax, absmax8, M, N, K, flags, pl.sps, pl.ks, a);
}

// D = 3 leaves the two-row-tile instances 8-12 registers short (a spill's reload waits with vmcnt(0) and drains the ring):
// they always run with two ring slots.
template <typename T, int D>
void ps_launch(const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax8, int M, int N, int K, int flags,
               const PsPlan& pl, const PsArgs& a, hipStream_t stream) {
    if (absmax8 != nullptr) {
        if (pl.mt == 1)
            ps_launch_one<T, 1, true, D>(A, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
        else
            ps_launch_one<T, 2, true, 2>(A, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
    } else {
        if (pl.mt == 1)
            ps_launch_one<T, 1, false, D>(A, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
        else
            ps_launch_one<T, 2, false, 2>(A, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
    }
}

} // namespace

// Preconditions: 16-bit activations, literal code tables, K a multiple of 256, blocksize >= 64 (a lane's 64 k of a chunk
// stay inside one quantization block), 16-byte aligned A and B.
bool gemm_4bit_ps_supported(int dtype, const void* A, const uint8_t* B, const float* code16, int M, int N, int K, int blocksize) {
    // (byte offsets of the buffer loads are 32-bit and must stay below 2^31)
    const long long nk = static_cast<long long>(N) * K, mk = static_cast<long long>(M) * K;
    return (dtype == 1 || dtype == 2) && code16 == nullptr && M >= 1 && N >= 1 && K >= kPsStageK && (K % kPsStageK) == 0 &&
           blocksize >= 64 && is_pow2(blocksize) && aligned_to(A, 16) && aligned_to(B, 16) && nk < (1LL << 31) && mk < (1LL << 30);
}

size_t gemm_4bit_ps_workspace_bytes(int M, int N, int K, int force_ks) {
    if (M < 1 || N < 1 || K < kPsStageK)
        return 0;
    const PsPlan pl = ps_plan(M, N, K, force_ks);
    return pl.ks > 1 ? static_cast<size_t>(pl.ks) * M * N * sizeof(float) : 0;
}

// dtype: 1 = f16, 2 = bf16. force_ks (0 = built-in choice), variant (0 = two ring slots, 1 = three where they fit): sweeps
// and tests.
void gemm_4bit_ps(int dtype, const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax8,
                  const float* absmax_code, const float* absmax_offset, void* out, const void* bias, int M, int N, int K,
                  int blocksize, int quant_type, void* workspace, size_t workspace_bytes, int force_ks, int variant,
                  int ablate, hipStream_t stream) {
    PsPlan pl = ps_plan(M, N, K, force_ks);
    float* ws = static_cast<float*>(workspace);
    const size_t slab = static_cast<size_t>(M) * N * sizeof(float);
    if (pl.ks > 1) {
        if (ws == nullptr) {
            ws = gemm_4bit_internal_workspace(slab * pl.ks, stream);
            workspace_bytes = ws ? slab * pl.ks : 0;
        }
        if (workspace_bytes < slab * pl.ks) {
            const int fit = static_cast<int>(workspace_bytes / slab);
            const int stages = K / kPsStageK;
            const int ks = fit >= 2 ? fit : 1;
            pl.sps = (stages + ks - 1) / ks;
            pl.ks = (stages + pl.sps - 1) / pl.sps;
        }
    }
    PsArgs a;
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
    flags |= (ablate & 7) << 16;
#else
    (void)ablate;
#endif
    if (dtype == 2) {
        if (variant == 1)
            ps_launch<bf16, 3>(A, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
        else
            ps_launch<bf16, 2>(A, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
    } else {
        if (variant == 1)
            ps_launch<f16, 3>(A, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
        else
            ps_launch<f16, 2>(A, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
    }
    BNB_CHECK_LAUNCH();
    if (pl.ks > 1)
        gemm_4bit_finalize(dtype, ws, bias, out, M, N, pl.ks, stream);
}

} // namespace bnb
