// gemm4_mfma.hip — fused 4-bit dequantize + MFMA GEMM for small batches (3 <= M, typically <= 128) on gfx950.
//   out[M, N] = A[M, K] * dequant(B)[N, K]^T (+ bias)
//
// The reference has tensor-core kernels for this only on NVIDIA (csrc/gemm_4bit_sm80.cu:127-457,
// csrc/gemm_4bit_sm75.cu:97-305: mma.sync + ldmatrix + smem-staged dequantized B tiles); on ROCm it
// writes the whole dequantized weight to HBM and calls hipBLASLt
// (bitsandbytes/backends/cuda/ops.py:904-916). These are the missing MFMA kernels, designed for CDNA4
// rather than translated. Common to all of them:
//
//  * v_mfma_f32_16x16x32_{bf16,f16}: A operand = activations (row m = lane%16), B operand = weights
//    (column n = lane%16), both hold k = 8*(lane/16) + i. Because the weight column of a lane equals
//    the accumulator column of that lane, the per-(column, block) absmax is a per-lane scalar: the fp32
//    partial tile of a 64-k block (two MFMAs) is scaled by the lane's fp32 absmax and added to the running
//    accumulator (4 v_fma per tile), so the scale is applied exactly, in fp32, after the MFMA — the
//    same idea as the reference's CDNA SIMT path (csrc/gemm_4bit_simt.cu:436-444), but on the matrix pipe.
//  * Packed bytes become B fragments through the same bank-private byte -> bf16x2 LDS table as the gemv
//    kernel (conflict-free ds_read_b32).
//  * Cross-workgroup K slices exist only to put ~256 workgroups on the chip. Each slice writes its fp32
//    partial tile to its own slab of a workspace with plain stores; a small finalize kernel adds the slabs
//    in slice order, adds the bias and rounds once. No atomics: results are bit-reproducible run to run
//    (the reference's test_matmul_4bit_weight_orientation demands exact equality between calls).
//
// This file holds the producer/consumer kernel (tall tiles, large matrices), the slab finalize kernel and the dispatcher;
// the register-transposed kernel for small batches on small / medium matrices is gemm4_mfma_rt.hip. Four earlier generations
// (weights straight to registers in fragment-shaped 32-byte pieces; a 16-column LDS-DMA kernel; a 128-column two-stage tile
// with workgroup barriers; the same with a 4-deep ring) were measured and retired - what each taught is in DESIGN.md and their
// sweep records in profiles/r1_sweep_mfma_variants.txt, profiles/r2_mfma_ab.txt.
#include "bnb_common.h"

#include <mutex>
#include <unordered_map>

namespace bnb {

#ifdef BNB_PROFILING
extern unsigned long long* g_dbg_buf; // c_api.hip (profiling builds only)
#endif
// Sweep / test overrides (bnb_mi355x_set_tuning). Atomics, and every call takes ONE snapshot of them: a sweep thread can
// never corrupt a concurrent launch, it can only change which (always correct) geometry that launch uses.
thread_local TlsKnob g_mfma_knob0{0}; // reserved (was: A-image variants of the retired LDS-DMA kernel)
thread_local TlsKnob g_mfma_knob1{0}; // 100 * cfg + K-slice count (0 = heuristic)

namespace {

using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <typename T> struct Mma;
template <> struct Mma<bf16> {
    using frag = __attribute__((ext_vector_type(8))) bf16;
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(frag, a), __builtin_bit_cast(frag, b), c, 0,
                                                       0, 0);
    }
    static __device__ __forceinline__ uint32_t pack(float lo, float hi) {
        using V = __attribute__((ext_vector_type(2))) bf16;
        V v;
        v[0] = static_cast<bf16>(lo);
        v[1] = static_cast<bf16>(hi);
        return __builtin_bit_cast(uint32_t, v);
    }
};
template <> struct Mma<f16> {
    using frag = __attribute__((ext_vector_type(8))) f16;
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(frag, a), __builtin_bit_cast(frag, b), c, 0, 0,
                                                      0);
    }
    static __device__ __forceinline__ uint32_t pack(float lo, float hi) {
        using V = __attribute__((ext_vector_type(2))) f16;
        V v;
        v[0] = static_cast<f16>(lo);
        v[1] = static_cast<f16>(hi);
        return __builtin_bit_cast(uint32_t, v);
    }
};

struct GemmArgs {
    const void* A;
    const uint8_t* B;
    const float* absmax;
    const uint8_t* absmax8;
    const float* absmax_code;
    const float* absmax_offset;
    const float* code16;
    void* out;
    const void* bias;
    float* ws; // fp32 [kslices, M, N] partial-sum slabs when kslices > 1
    int M, N, K;
    int bs_shift;
    int quant_type;
    unsigned long long* dbg; // profiling only: s_memtime stamps, 8 per wavefront (NULL in production)
    int ablate;              // profiling only: 1 = skip decode + MFMA (stream only), 2 = also skip the A loads
    int knob0;               // host only: snapshot of g_mfma_knob0 for this call
    int kslices;     // number of K slices (grid.y)
    int steps_total; // K / 256
};

constexpr int kKC = 256;   // K granularity of the kernels: one pipeline group = 4 quantization blocks of 64

typedef __attribute__((address_space(1))) const void* dma_src_t;
typedef __attribute__((address_space(3))) void* dma_dst_t;

// ---------------------------------------------------------------------------------------------
// gemm4_mfma_pc_kernel ("v5" in profiles/, "producer / consumers"). What the retired two-stage tiled kernel
// ("v4") got wrong (measured ~3 us per 256-k chunk at M = 64 against ~1.3 us of LDS traffic): its double
// buffer kept ONE chunk (16 KiB per CU) of weights in flight, so every chunk eats most of an HBM round trip, and because vmcnt retires in
// order, the short-distance A prefetch of a wavefront forces all of its older weight DMAs to complete.
// Here the two streams are decoupled by giving them to different wavefronts (vmcnt is per wavefront):
//   * CW consumer wavefronts each own NTW*16 weight columns and a PRIVATE D-deep LDS ring of 256-k
//     chunks filled by LDS-DMA in full 128-B lines (+ the chunk's scales, also by DMA, so a consumer's
//     vector-memory queue holds DMAs only and its waits are exact counted vmcnt). A ring slot is refilled
//     right after the wavefront has consumed it: no barrier is involved in the weight stream at all and
//     D-1 chunks per wavefront stay in flight across everything else.
//   * one producer wavefront streams the shared A tile [MT*16][256] into a two-stage buffer (source-side
//     XOR swizzle, conflict-free ds_read_b128 fragment reads) and is the only one that waits
//     for it. ONE s_barrier per chunk: arriving at barrier k the producer guarantees A(k) has landed and
//     the consumers guarantee they are done with A(k-1), whose stage the producer refills right after.
//   * an A fragment is read once per (m-tile, k-step) and used for the wavefront's NTW n-tiles.
// K slices across workgroups write fp32 slabs; gemm4_finalize_kernel adds them.
// ---------------------------------------------------------------------------------------------
template <int kCount> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kCount) : "memory");
}

// kPcProducers wavefronts share the A-tile DMA issue: one LDS-DMA instruction costs its issuing wavefront
// ~100-180 cycles while the CU is busy, so a single producer needs > 3000 cycles for the 32 pieces of a
// 64-row stage - longer than the consumers need for a chunk (measured with the s_memtime stamps below).
// 8 + 4 wavefronts = 3 per SIMD, the same register budget (168) as 8 + 1.
constexpr int kPcProducers = 4;

template <typename T, int MT, bool NESTED, int CW, int NTW, int D>
__global__ __launch_bounds__((CW + kPcProducers) * 64) void gemm4_mfma_pc_kernel(
    // hot arguments as separate scalars: eligible for kernarg preload into SGPRs (see gemv4_stream.hip)
    // (exactly the 14 preloadable dwords: 16 user SGPRs minus the kernarg segment pointer; the output pointer is
    // needed last and stays in the struct)
    const void* hot_A, const uint8_t* hot_B, const float* hot_absmax, const float* hot_code16, int hot_M, int hot_N,
    int hot_K, int hot_bs_shift, int hot_kslices, int hot_quant_type, const GemmArgs p) {
    void* const hot_out = p.out;
    constexpr int kLutBytes = 256 * 32 * 4;
    constexpr int XB = MT * 16 * 512;                          // bytes of one A stage
    constexpr int SB = NTW * 256 * (NESTED ? 2 : 1);           // scale bytes of one slot
    constexpr int WB = NTW * 2048 + SB;                        // bytes of one ring slot of one consumer
    constexpr int XI = MT * 8;                                 // DMA instructions per A stage (2 rows each)
    constexpr int NP = kPcProducers;
    constexpr int XIP = XI / NP;                               // ... per producer wavefront
    static_assert(XI % NP == 0, "A-stage DMAs must divide over the producers");
    constexpr int LPC = NTW * (3 + (NESTED ? 1 : 0));          // consumer vm ops per chunk
    static_assert(XI < 64 && (D - 1) * LPC < 64, "vmcnt immediates");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* lut = reinterpret_cast<uint32_t*>(smem);
    unsigned char* xring = smem + kLutBytes;                   // [2][XB]
    unsigned char* wring = xring + 2 * XB;                     // [CW][D][WB]
    float* code2 = reinterpret_cast<float*>(wring + CW * D * WB);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = hot_N, K = hot_K, M = hot_M;
    const int m_base = blockIdx.z * (MT * 16);

    const int chunks_total = K >> 8;
    const int per_wg = (chunks_total + hot_kslices - 1) / hot_kslices;
    const int c_begin = blockIdx.y * per_wg;
    const int c_end = (c_begin + per_wg < chunks_total) ? c_begin + per_wg : chunks_total;
    const int n = (c_end > c_begin) ? c_end - c_begin : 0;
    // profiling builds only (bnb_mi355x_set_stamp_buffer): 16 s_memtime stamps per wavefront. (Until round 3 the product build
    // carried them as run-time branches on a NULL pointer - three of them inside the chunk loop.)
#ifdef BNB_PROFILING
#define BNB_PC_STAMP(i)                                                                            \
    if (p.dbg && lane == 0)                                                                        \
        p.dbg[((((static_cast<long>(blockIdx.x) * gridDim.y + blockIdx.y) * gridDim.z + blockIdx.z) * (CW + kPcProducers)) + wave) * 16 + (i)] = \
            __builtin_amdgcn_s_memtime();
#else
#define BNB_PC_STAMP(i) {}
#endif
    BNB_PC_STAMP(0)

    if (wave >= CW) {
        // ---------------- producers: the A tile, instruction i of a stage belongs to producer i % NP
        const int pw = wave - CW;
        const T* __restrict__ A = static_cast<const T*>(hot_A);
        const int r2 = lane >> 5, s32 = lane & 31;
        auto issue_a = [&](int c, int stage) {
#pragma unroll
            for (int ii = 0; ii < XIP; ++ii) {
                const int i = ii * NP + pw;
                const int row = 2 * i + r2;
                int m = m_base + row;
                m = (m < M) ? m : M - 1;
                const T* src = A + static_cast<long>(m) * K + (static_cast<long>(c) << 8) + ((s32 ^ (row & 15)) << 3);
                __builtin_amdgcn_global_load_lds((dma_src_t)src, (dma_dst_t)(xring + stage * XB + i * 1024), 16, 0, 0);
            }
        };
        // The 4 producers (256 threads) also build the byte -> pair table, one entry per thread, while the
        // consumers go straight to their weight stream. The two code values are loaded before the DMAs are
        // issued so that their counted wait leaves the DMAs in flight.
        static_assert(NP * 64 == 256, "one table entry per producer thread");
        const int e = tid - CW * 64;
        const gfloat_ptr tbl = (gfloat_ptr)(hot_code16 ? hot_code16 : (hot_quant_type == kNF4 ? kNF4Code : kFP4Code));
        const float code_hi = tbl[e >> 4];
        const float code_lo = tbl[e & 15];
        float code2_v = 0.0f;
        if constexpr (NESTED)
            code2_v = p.absmax_code[e];
        if (n > 0)
            issue_a(c_begin, 0);
        if (n > 1)
            issue_a(c_begin + 1, 1);
        {
            const uint32_t pr = Mma<T>::pack(code_hi, code_lo);
            const u32x4 v = {pr, pr, pr, pr};
            u32x4* dst = reinterpret_cast<u32x4*>(&lut[e * 32]);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                dst[(j + e) & 7] = v; // rotated chunk order: conflict-free ds_write_b128 (see gemv4_stream.hip)
            if constexpr (NESTED)
                code2[e] = code2_v;
        }
        for (int k = 0; k < n; ++k) {
            if (k == 0 && n > 1)
                wait_vmcnt<XIP>(); // this producer's part of A(0) landed, A(1) may still be in flight
            else
                wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // table written (first time)
            __builtin_amdgcn_s_barrier(); // #k
            if (k >= 1 && k + 1 < n)
                issue_a(c_begin + k + 1, (k + 1) & 1); // stage of A(k-1): every consumer is past it
        }
        return;
    }

    // ---------------- consumers
    const int ln = lane & 15, lg = lane >> 4;
    const int colw = blockIdx.x * (CW * NTW * 16) + wave * (NTW * 16); // first column of this wavefront
    const int r8 = lane >> 3, s8 = lane & 7;
    const uint8_t* wsrc[NTW][2];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int row = colw + t * 16 + h * 8 + r8;
            row = (row < N) ? row : N - 1;
            wsrc[t][h] = hot_B + static_cast<long>(row) * (K >> 1) + ((s8 ^ r8) << 4);
        }
    // scale DMA: lane L fetches the scale of (row L >> 2, 64-k block L & 3) -> LDS offset 4 L, i.e. the four
    // scales of a row are 16 contiguous bytes
    long srow[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        int row = colw + t * 16 + (lane >> 2);
        row = (row < N) ? row : N - 1;
        srow[t] = static_cast<long>(row) * K + (lane & 3) * 64;
    }
    unsigned char* wbase = wring + wave * (D * WB);

    auto issue_w = [&](int c, int slot) {
        unsigned char* dst = wbase + slot * WB;
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                __builtin_amdgcn_global_load_lds((dma_src_t)(wsrc[t][h] + static_cast<long>(c) * 128),
                                                 (dma_dst_t)(dst + (t * 2 + h) * 1024), 16, 0, 0);
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const long q = (srow[t] + (static_cast<long>(c) << 8)) >> hot_bs_shift;
            if constexpr (NESTED) {
                __builtin_amdgcn_global_load_lds((dma_src_t)(p.absmax8 + q), (dma_dst_t)(dst + NTW * 2048 + t * 256), 1, 0, 0);
                __builtin_amdgcn_global_load_lds((dma_src_t)(hot_absmax + (q >> 8)),
                                                 (dma_dst_t)(dst + NTW * 2048 + NTW * 256 + t * 256), 4, 0, 0);
            } else {
                __builtin_amdgcn_global_load_lds((dma_src_t)(hot_absmax + q), (dma_dst_t)(dst + NTW * 2048 + t * 256), 4, 0, 0);
            }
        }
    };

    float offset = 0.0f;
    if constexpr (NESTED)
        offset = p.absmax_offset[0]; // scalar load

    // Only chunk 0 goes out now. A wavefront pushes its loads into the memory pipeline back to back, so with the whole ring
    // requested at once the chunk-0 data of the last wavefronts queues behind chunks 1 .. D-1 of the first ones (the first
    // chunk landed ~9100 cycles into the kernel, r1_timeline_mfma_pc_smemtime.txt); the rest of the ring follows after
    // barrier #0, still a chunk of compute ahead of its use.
    constexpr bool STAGGER = (NTW == 1); // (the two-tile instances have no registers to spare for the second issue site)
    issue_w(c_begin, 0);
    if constexpr (!STAGGER) {
#pragma unroll
        for (int j = 1; j < D; ++j)
            if (j < n)
                issue_w(c_begin + j, j);
    }

    f32x4 acc[MT][NTW];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NTW; ++t)
            acc[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    BNB_PC_STAMP(1)

    // LDS byte addresses, split into a per-lane part computed once and compile-time parts that reach the
    // instruction as an XOR constant or an immediate offset (the sums below have disjoint bits, so + is ^):
    //   A fragment (mt, b, jj): row (mt*16 + ln) * 512 + (((b*8 + lg*2 + jj) ^ ln) << 4)
    //   packed weights (t, b) : t*2048 + (ln>>3)*1024 + (ln&7)*128 + (((2b + (lg>>1)) ^ (ln&7)) << 4) + (lg&1)*8
    //   table look-up         : (byte << 7) + (lane & 31) * 4
    const uint32_t a_lane = static_cast<uint32_t>(ln * 512 + (((lg * 2) ^ ln) << 4));
    const uint32_t w_lane = static_cast<uint32_t>((ln >> 3) * 1024 + (ln & 7) * 128 + (((lg >> 1) ^ (ln & 7)) << 4) + (lg & 1) * 8);
    const uint32_t lut_lane = static_cast<uint32_t>((lane & 31) * 4) +
                              static_cast<uint32_t>(reinterpret_cast<uintptr_t>((dma_dst_t)lut));

    for (int k0 = 0; k0 < n; k0 += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const int k = k0 + j;
            if (k >= n)
                break;
            // chunk k of this wavefront has landed; the min(D-1, n-1-k) younger ones stay in flight
            const bool first = STAGGER && (j == 0) && (k0 == 0); // (j is a constant of the unrolled copy: the extra code exists once)
            const int younger = first ? 0 : ((n - 1 - k < D - 1) ? n - 1 - k : D - 1); // (chunk 0: nothing else is in flight yet)
            if (younger <= 0)
                wait_vmcnt<0>();
            else if (younger == 1)
                wait_vmcnt<1 * LPC>();
            else if (younger == 2)
                wait_vmcnt<(D > 2 ? 2 : 1) * LPC>();
            else if (younger == 3)
                wait_vmcnt<(D > 3 ? 3 : 1) * LPC>();
            else
                wait_vmcnt<(D > 4 ? 4 : 1) * LPC>();
            if (k < 4)
                BNB_PC_STAMP(2 + 3 * k)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier(); // #k: A(k) is in LDS (and, the first time, the tables)
#ifdef BNB_PROFILING
            // (experiment queued for the next round, measurement build only, tools/pc_phase_shift_ab.py: the second consumer
            // wavefront of every SIMD - wavefronts 4 .. 7 - runs (knob0 >> 4) x 128 cycles behind the first, so that the two stop
            // doing their table look-ups, their MFMAs and their scale FMAs at the same time: the chunk time of this kernel equals
            // the SUM of LDS, matrix-pipe and VALU time)
            if ((p.knob0 >> 4) != 0 && (wave & 4) != 0)
                for (int i = 0; i < (p.knob0 >> 4); ++i)
                    __builtin_amdgcn_s_sleep(2);
#endif
            if (k < 4)
                BNB_PC_STAMP(3 + 3 * k)
            if (first) {
#pragma unroll
                for (int jj = 1; jj < D; ++jj)
                    if (jj < n)
                        issue_w(c_begin + jj, jj);
            }
            const int zsh = opaque_zero();

            const unsigned char* xb = xring + (k & 1) * XB;
            const unsigned char* wb = wbase + j * WB;
            // chunk-level reads first: the packed weights of all four 64-k blocks (one ds_read_b64 each) and
            // the scales; then a two-stage software pipeline over the blocks: the LDS reads of block b+1
            // (A fragments, table look-ups) are issued before the MFMAs of block b, so the matrix pipe does
            // not sit behind two dependent LDS round trips per block
            u32x2 w2[NTW][4];
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    w2[t][b] = *reinterpret_cast<const u32x2*>(wb + t * 2048 + (w_lane ^ static_cast<uint32_t>(b << 5)));
            f32x4 sc4[NTW];
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                if constexpr (NESTED) {
                    const u32x4 q8 = *reinterpret_cast<const u32x4*>(wb + NTW * 2048 + t * 256 + ln * 16);
                    const f32x4 a2 = *reinterpret_cast<const f32x4*>(wb + NTW * 2048 + NTW * 256 + t * 256 + ln * 16);
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        sc4[t][b] = nested_scale(code2[q8[b] & 0xFFu], a2[b], offset);
                } else {
                    sc4[t] = *reinterpret_cast<const f32x4*>(wb + NTW * 2048 + t * 256 + ln * 16);
                }
            }
            u32x4 af[MT][2];
            u32x4 bf[2][NTW][2];
            auto fetch_a = [&](int b) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
                        af[mt][jj] = *reinterpret_cast<const u32x4*>(xb + mt * 8192 + (a_lane ^ static_cast<uint32_t>((b * 8 + jj) << 4)));
            };
            auto fetch_lut = [&](int b, int st) {
#pragma unroll
                for (int t = 0; t < NTW; ++t)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const uint32_t w = w2[t][b][jj];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            // two VALU ops per byte, spelled out: the compiler's own canonical form is four
                            const uint32_t byte = __builtin_amdgcn_ubfe(w, static_cast<uint32_t>(8 * q + zsh), 8u);
                            const uint32_t addr = (byte << 7) + lut_lane; // lut_lane includes the table's LDS base
                            bf[st][t][jj][q] = *reinterpret_cast<const __attribute__((address_space(3))) uint32_t*>(addr);
                        }
                    }
            };
            fetch_lut(0, 0);
            fetch_a(0);
            // Two-deep software pipeline: block b's MFMAs are issued, and only the partial tiles of block b-1
            // - finished long ago - are scaled and accumulated, so the wavefront never sits waiting for the
            // matrix pipe to drain before it can issue the next block's reads.
            f32x4 part[2][MT][NTW];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int st = b & 1;
                // (1) table look-ups of block b+1 (the second of two dependent LDS round trips) go out first
                if (b + 1 < 4)
                    fetch_lut(b + 1, st ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                // (2) all MFMAs of block b
#pragma unroll
                for (int t = 0; t < NTW; ++t)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        part[st][mt][t] = Mma<T>::run(af[mt][0], bf[st][t][0], f32x4{0.f, 0.f, 0.f, 0.f});
                        part[st][mt][t] = Mma<T>::run(af[mt][1], bf[st][t][1], part[st][mt][t]);
                    }
                __builtin_amdgcn_sched_barrier(0);
                // (3) the A fragments of block b+1 are read while the matrix pipe works on block b
                if (b + 1 < 4)
                    fetch_a(b + 1);
                __builtin_amdgcn_sched_barrier(0);
                // (4) fp32 scale of the 64-k partial tiles of block b-1
                if (b > 0) {
#pragma unroll
                    for (int t = 0; t < NTW; ++t) {
                        const float scale = sc4[t][b - 1];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                acc[mt][t][r] = fmaf(scale, part[st ^ 1][mt][t][r], acc[mt][t][r]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                const float scale = sc4[t][3];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[mt][t][r] = fmaf(scale, part[1][mt][t][r], acc[mt][t][r]);
            }
            if (k < 4)
                BNB_PC_STAMP(4 + 3 * k)
            if (k + D < n) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // this wavefront's reads of the slot are done
                issue_w(c_begin + k + D, j);
            }
        }
    }

    BNB_PC_STAMP(14)
    T* __restrict__ out = static_cast<T*>(hot_out);
    const T* __restrict__ bias = static_cast<const T*>(p.bias);
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int col = colw + t * 16 + ln;
        if (col >= N)
            continue;
        const float bv = (bias && hot_kslices == 1) ? static_cast<float>(bias[col]) : 0.0f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m_base + mt * 16 + lg * 4 + r;
                if (m >= M)
                    continue;
                const long o = static_cast<long>(m) * N + col;
                if (hot_kslices == 1)
                    out[o] = static_cast<T>(acc[mt][t][r] + bv);
                else // (slabs go around the L2: less dirty data to write back at the kernel boundary - with the finalize kernel's
                     // non-temporal loads C3 24.1 -> 23.4 us, 8192^2 M = 32 19.5 -> 18.4, profiles/r3_slab_cache_policy_ab.txt)
                    __builtin_nontemporal_store(acc[mt][t][r], &p.ws[static_cast<long>(blockIdx.y) * M * N + o]);
            }
        }
    }
    BNB_PC_STAMP(15)
#undef BNB_PC_STAMP
}

// out = T(sum_s ws[s] + bias), slabs added in slice order (deterministic)
template <typename T, int BATCH>
__global__ __launch_bounds__(256) void gemm4_finalize_kernel(const float* __restrict__ ws, const T* __restrict__ bias,
                                                             T* __restrict__ out, long total, int N, int kslices) {
    const long i = (static_cast<long>(blockIdx.x) * 256 + threadIdx.x) * 4;
    if (i >= total)
        return;
    if (i + 4 <= total && (total % 4) == 0 && (N % 4) == 0) {
        // All slab loads of a batch are issued before the first add: written as a plain loop the compiler
        // waits for every load before issuing the next one and the launch costs kslices memory round trips
        // (4.6 us measured for 8 slabs). The adds still run in slice order.
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        // (BATCH = 2 / 4 / 8 by slice count: with a fixed batch of 8, four slices meant four wasted re-reads per thread)
        for (int s0 = 0; s0 < kslices; s0 += BATCH) {
            f32x4 w[BATCH];
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                const int sl = (s0 + j < kslices) ? s0 + j : kslices - 1; // clamp: a re-read, never out of range
                // (read once, written by another launch: non-temporal - 28672 x 8192 M = 64 62.3 -> 59.6 us)
                w[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ws + static_cast<long>(sl) * total + i));
            }
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                const bool live = s0 + j < kslices; // wave-uniform
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    v[q] = live ? v[q] + w[j][q] : v[q];
            }
        }
        const int col = static_cast<int>(i % N);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float b = bias ? static_cast<float>(bias[col + j]) : 0.0f;
            out[i + j] = static_cast<T>(v[j] + b);
        }
    } else {
        for (long e = i; e < i + 4 && e < total; ++e) {
            float v = ws[e];
            for (int s = 1; s < kslices; ++s)
                v += ws[static_cast<long>(s) * total + e];
            const float b = bias ? static_cast<float>(bias[e % N]) : 0.0f;
            out[e] = static_cast<T>(v + b);
        }
    }
}

template <typename T>
void launch_finalize(const float* ws, const T* bias, T* out, long total, int N, int kslices, hipStream_t stream) {
    const long threads = (total + 3) / 4;
    const dim3 grid(static_cast<unsigned>((threads + 255) / 256));
    if (kslices <= 2)
        hipLaunchKernelGGL((gemm4_finalize_kernel<T, 2>), grid, dim3(256), 0, stream, ws, bias, out, total, N, kslices);
    else if (kslices <= 4)
        hipLaunchKernelGGL((gemm4_finalize_kernel<T, 4>), grid, dim3(256), 0, stream, ws, bias, out, total, N, kslices);
    else
        hipLaunchKernelGGL((gemm4_finalize_kernel<T, 8>), grid, dim3(256), 0, stream, ws, bias, out, total, N, kslices);
}

// ---------------------------------------------------------------------------------------------
// Library-owned split-K workspace, used only when the caller passes none (the reference ABI has no
// workspace argument): one buffer per (device, stream) so concurrent streams never share slabs;
// created on first use with the stream-ordered allocator (hipMallocAsync: no device-wide synchronisation inside a
// launch call); when a larger one is needed the old buffer is returned with hipFreeAsync ON THE SAME STREAM, i.e. after
// every kernel already enqueued on it - nothing leaks, nothing is freed under a running kernel. Never allocated while
// the stream is being captured (it would invalidate the capture): such a call runs with a single K slice; callers that
// capture graphs pass their own workspace (bnb_mi355x_gemm_4bit, what the Python host does).
// ---------------------------------------------------------------------------------------------
struct WsKey {
    int dev;
    hipStream_t s;
    bool operator==(const WsKey& o) const { return dev == o.dev && s == o.s; }
};
struct WsKeyHash {
    size_t operator()(const WsKey& k) const {
        return std::hash<const void*>()(k.s) ^ (static_cast<size_t>(k.dev) * 0x9E3779B97F4A7C15ull);
    }
};
struct WsBuf {
    float* p = nullptr;
    size_t bytes = 0;
};
std::mutex g_ws_mu;
std::unordered_map<WsKey, WsBuf, WsKeyHash> g_ws;

float* get_internal_workspace(size_t bytes, hipStream_t stream) {
    int dev = 0;
    BNB_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_ws_mu);
    WsBuf& b = g_ws[WsKey{dev, stream}];
    if (b.bytes < bytes) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
            (void)hipGetLastError();
            static std::atomic<bool> warned{false};
            if (!warned.exchange(true))
                fprintf(stderr, "bitsandbytes_amd: cgemm_4bit_* called inside a stream capture without a workspace: the split-K "
                                "scratch cannot be allocated there, this launch runs with a single K slice (slower on small N). "
                                "Warm the stream up once before capturing, or pass a workspace (bnb_mi355x_gemm_4bit).\n");
            return nullptr;
        }
        size_t want = bytes < (size_t(16) << 20) ? (size_t(16) << 20) : bytes;
        float* np = nullptr;
        if (hipMallocAsync(reinterpret_cast<void**>(&np), want, stream) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        if (b.p != nullptr && hipFreeAsync(b.p, stream) != hipSuccess) // stream-ordered: after the kernels that still use it
            (void)hipGetLastError();
        b.p = np;
        b.bytes = want;
    }
    return b.p;
}

struct Plan {
    int mt, ks;
    int cfg; // producer/consumer kernel: 11: 8 consumers x 1 n-tile (128 columns), 12: 4 x 2 (128), 13: 8 x 2 (256),
             // 14: 4 x 1 (64 columns)
             // (0-4 and 7-10 were the register-ring and two-stage tiled generations: retired, their sweep record is
             // profiles/r1_sweep_mfma_variants.txt)
};

constexpr bool cfg_is_pc(int cfg) { return cfg >= 11 && cfg <= 14; }
constexpr int cfg_cols(int cfg) { return cfg == 13 ? 256 : cfg == 14 ? 64 : 128; }

// Kernel, tile shape and K-slice count for a problem: a pure function of (M, N, K) and the tuning knobs,
// shared by the launch and by the workspace-size query.
Plan make_plan(int M, int N, int K, int knob1) {
    Plan pl;
    pl.mt = (M > 48) ? 4 : (M > 32) ? 3 : (M > 16) ? 2 : 1;
    const int groups = K / kKC; // units of 4 blocks
    const int gz = (M + pl.mt * 16 - 1) / (pl.mt * 16);
    int ks = knob1 % 100;
    int cfg = knob1 / 100;
    if (!cfg_is_pc(cfg)) {
        // Calibrated on MI355X (profiles/r1_sweep_mfma_variants.txt): 8 consumer wavefronts x 16 columns share one A tile;
        // on a small matrix with a tall tile 64-column workgroups need half the K slices, i.e. half the slab traffic
        // (4096^2 M = 64: 14.6 vs 15.4 us), on large matrices 128 columns win
        // (64-column workgroups only while they fit one round of workgroups: 4096^2 M = 512 ran 39.6 us in 512 of them against
        // 27.5 in 256 of 128 columns, profiles/r4_route_ab.txt)
        const bool big = static_cast<long>(N) * K >= (32L << 20) && N >= 1024;
        const long wgs14 = static_cast<long>((N + 63) / 64) * gz;
        cfg = (pl.mt >= 3 && !big && wgs14 <= device_cu_count_or_default()) ? 14 : 11;
    }
    if (cfg == 13 && pl.mt > 2)
        cfg = 11; // 8 x 2 consumers with a > 32-row A tile do not fit the 160 KiB of LDS
    pl.cfg = cfg;
    const int gx = (N + cfg_cols(cfg) - 1) / cfg_cols(cfg);
    if (ks == 0)
        ks = 256 / (gx * gz); // one workgroup per CU: fill the 256 CUs but never spill into a second round
                              // (11008 x 4096: 3 slices = 258 workgroups measured 18.8 us, 2 slices 15.2)
    const int max_ks = groups / 2 > 0 ? groups / 2 : 1; // >= 2 chunks per workgroup so the ring has something to overlap
    if (ks > max_ks)
        ks = max_ks;
    if (ks < 1)
        ks = 1;
    const int per = (groups + ks - 1) / ks;
    pl.ks = (groups + per - 1) / per; // every slice non-empty
    return pl;
}

constexpr size_t pc_smem_bytes(int MT, int CW, int NTW, int D, bool nested) {
    return 256 * 32 * 4 + 2 * static_cast<size_t>(MT) * 16 * 512 +
           static_cast<size_t>(CW) * D * (static_cast<size_t>(NTW) * 2048 + static_cast<size_t>(NTW) * 256 * (nested ? 2 : 1)) + 1024;
}
// 151 KiB of dynamic LDS launches, 157 KiB is refused (hipFuncSetAttribute: invalid argument) - stay below
constexpr size_t kPcSmemCap = 155 * 1024;

template <typename T, int MT, bool NESTED, int CW, int NTW, int D> void launch_mfma_pc_one(GemmArgs& p, hipStream_t stream) {
    constexpr size_t smem = pc_smem_bytes(MT, CW, NTW, D, NESTED);
    static_assert(smem <= kPcSmemCap, "ring depth does not fit the LDS");
    constexpr int BN = CW * NTW * 16;
    const int gx = (p.N + BN - 1) / BN;
    const int gz = (p.M + MT * 16 - 1) / (MT * 16);
    dim3 grid(gx, p.kslices, gz);
    auto kern = gemm4_mfma_pc_kernel<T, MT, NESTED, CW, NTW, D>;
    static LdsLimit lds_limit;
    ensure_dynamic_lds(lds_limit, reinterpret_cast<const void*>(kern), smem);
    hipLaunchKernelGGL(kern, grid, dim3((CW + kPcProducers) * 64), smem, stream, p.A, p.B, p.absmax, p.code16, p.M, p.N, p.K, p.bs_shift, p.kslices, p.quant_type, p);
}

// deepest ring (<= 4 chunks) that fits
constexpr int pc_depth(int MT, int CW, int NTW, bool nested) {
    for (int d = 4; d >= 2; --d)
        if (pc_smem_bytes(MT, CW, NTW, d, nested) <= kPcSmemCap)
            return d;
    return 0;
}

template <typename T, int MT, int CW, int NTW> void launch_mfma_pc(GemmArgs& p, hipStream_t stream) {
    if (p.absmax8 != nullptr)
        launch_mfma_pc_one<T, MT, true, CW, NTW, pc_depth(MT, CW, NTW, true)>(p, stream);
    else
        launch_mfma_pc_one<T, MT, false, CW, NTW, pc_depth(MT, CW, NTW, false)>(p, stream);
}

template <typename T, int MT> void launch_mfma(GemmArgs& p, int cfg, hipStream_t stream) {
    // producer/consumer geometries. LDS: 32 KiB table + 2 x MT*8 KiB A stages + CW*D ring slots.
    if (cfg == 12)
        return launch_mfma_pc<T, MT, 4, 2>(p, stream);
    if (cfg == 14)
        return launch_mfma_pc<T, MT, 4, 1>(p, stream);
    if constexpr (pc_depth(MT, 8, 2, true) >= 2) {
        if (cfg == 13)
            return launch_mfma_pc<T, MT, 8, 2>(p, stream);
    }
    return launch_mfma_pc<T, MT, 8, 1>(p, stream); // cfg 11
}

template <typename T> void dispatch_mfma(GemmArgs& p, float* ws, size_t ws_bytes, int knob1, hipStream_t stream) {
    g_last_gemm_kernel = kKernelPc;
    Plan pl = make_plan(p.M, p.N, p.K, knob1);
    const size_t slab = static_cast<size_t>(p.M) * p.N * sizeof(float);
    if (pl.ks > 1) {
        if (ws == nullptr) {
            ws = get_internal_workspace(slab * pl.ks, stream);
            ws_bytes = ws ? slab * pl.ks : 0;
        }
        if (ws_bytes < slab * pl.ks) {
            // not enough room for all slabs: use as many slices as fit (>= 2), else none
            const int fit = static_cast<int>(ws_bytes / slab);
            const int steps = p.steps_total;
            int ks = fit >= 2 ? fit : 1;
            const int per = (steps + ks - 1) / ks;
            pl.ks = (steps + per - 1) / per;
        }
    }
    p.ws = ws;
    p.kslices = pl.ks;
    switch (pl.mt) {
    case 1:
        launch_mfma<T, 1>(p, pl.cfg, stream);
        break;
    case 2:
        launch_mfma<T, 2>(p, pl.cfg, stream);
        break;
    case 3:
        launch_mfma<T, 3>(p, pl.cfg, stream);
        break;
    default:
        launch_mfma<T, 4>(p, pl.cfg, stream);
        break;
    }
    BNB_CHECK_LAUNCH();

    if (pl.ks > 1) {
        launch_finalize<T>(p.ws, static_cast<const T*>(p.bias), static_cast<T*>(p.out), static_cast<long>(p.M) * p.N, p.N, pl.ks, stream);
        BNB_CHECK_LAUNCH();
    }
}

} // namespace

// gemm4_mfma_rt.hip (the register-transposed kernel for small batches on small / medium matrices)
bool gemm_4bit_rt_supported(int dtype, const void* A, const uint8_t* B, const float* code16, int M, int N, int K, int blocksize);
size_t gemm_4bit_rt_workspace_bytes(int M, int N, int K, int force_ks);
bool gemm_4bit_rt_serves(const float* absmax, const uint8_t* absmax8, int blocksize);
void gemm_4bit_rt(int dtype, const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax8,
                  const float* absmax_code, const float* absmax_offset, void* out, const void* bias, int M, int N, int K,
                  int blocksize, int quant_type, void* workspace, size_t workspace_bytes, int force_ks, int force_waves,
                  int variant, hipStream_t stream);

// gemm4_mfma_kq.hip (the K-quarter kernel: 32x32x16 MFMA, shares of a chunk copied to registers, 17 ... 64-row batches)
bool gemm_4bit_kq_supported(int dtype, const void* A, const uint8_t* B, const float* code16, int M, int N, int K, int blocksize);
size_t gemm_4bit_kq_workspace_bytes(int M, int N, int K, int force_ks);
bool gemm_4bit_kq_serves(const float* absmax, const uint8_t* absmax8, int blocksize, int K);
void gemm_4bit_kq(int dtype, const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax8,
                  const float* absmax_code, const float* absmax_offset, void* out, const void* bias, int M, int N, int K,
                  int blocksize, int quant_type, void* workspace, size_t workspace_bytes, int force_ks, int ablate, hipStream_t stream);

// gemv4_stream.hip (what a call outside every MFMA kernel's preconditions runs)
void gemv_4bit_stream(int dtype, const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax8,
                      const float* absmax_code, const float* absmax_offset, const float* code16, void* out,
                      const void* bias, int M, int N, int K, int blocksize, int quant_type, hipStream_t stream);

// gemm4_mfma_sm.hip (the streaming MFMA kernel: one persistent workgroup per CU, activations once per CU; 2 ... 16 rows)
bool gemm_4bit_sm_supported(int dtype, const void* A, const uint8_t* B, const float* code16, int M, int N, int K, int blocksize);
bool gemm_4bit_sm_serves(const float* absmax, const uint8_t* absmax8, int blocksize);
void gemm_4bit_sm(int dtype, const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax8,
                  const float* absmax_code, const float* absmax_offset, void* out, const void* bias, int M, int N, int K,
                  int blocksize, int quant_type, int variant, hipStream_t stream);

// gemm4_mfma_tall.hip (128 x 128 tiles, pre-scaled operand decoded once per 128 rows: tall batches)
bool gemm_4bit_tall_supported(int dtype, const void* A, const uint8_t* B, const float* code16, int M, int N, int K, int blocksize);
void gemm_4bit_tall(int dtype, const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax8, const float* absmax_code,
                    const float* absmax_offset, void* out, const void* bias, int M, int N, int K, int blocksize, int quant_type,
                    int ablate, hipStream_t stream);

// shared with gemm4_mfma_rt.hip: the slab finalize launch and the library-owned workspace
void gemm_4bit_finalize(int dtype, const float* ws, const void* bias, void* out, int M, int N, int kslices, hipStream_t stream) {
    const long total = static_cast<long>(M) * N;
    if (dtype == 2)
        launch_finalize<bf16>(ws, static_cast<const bf16*>(bias), static_cast<bf16*>(out), total, N, kslices, stream);
    else
        launch_finalize<f16>(ws, static_cast<const f16*>(bias), static_cast<f16*>(out), total, N, kslices, stream);
    BNB_CHECK_LAUNCH();
}
float* gemm_4bit_internal_workspace(size_t bytes, hipStream_t stream) { return get_internal_workspace(bytes, stream); }

namespace {
// Which problems go to the register-transposed kernel (tuning knob cfg 20 / 21 / 22 forces it: built-in / 8 / 16 wavefronts).
// Calibrated on MI355X - see DESIGN.md: batches of at most one row tile on matrices small enough that N / 16 workgroups
// need no (or few) K slices.
bool rt_selected(int M, int N, int K, int knob1, int* force_ks, int* force_waves) {
    const int cfg = knob1 / 100;
    *force_ks = 0;
    *force_waves = 0;
    if (cfg >= 20 && cfg <= 22) {
        *force_ks = knob1 % 100;
        *force_waves = cfg == 21 ? 8 : cfg == 22 ? 16 : 0;
        return true;
    }
    if (cfg != 0)
        return false;
    // measured on MI355X (profiles/r2_mfma_ab.txt), us per launch, register-transposed vs producer/consumer kernel:
    // M <= 16: 4096^2 6.5 vs 8.4, 11008 x 4096 11.2 vs 14.7, 8192^2 14.2-15.6 vs 16.3-16.6; M = 32: 4096^2 8.6 vs 10.7 but
    // 11008 x 4096 19.2 vs 17.4; M = 64: 1376 x 4096 7.8 vs 9.5, 4096^2 equal, 8192^2 43 vs 23
    const long weights = static_cast<long>(N) * K;
    if (M <= 16)
        return weights <= (96L << 20);
    if (M <= 32)
        return weights <= (20L << 20);
    // 33 ... 64 rows (two row tiles): round 4's table (profiles/r4_route_ab.txt), register-transposed vs producer/consumer:
    //   M = 40 / 48: 4096^2 8.9 / 9.1 vs 9.4 / 9.5, 3072^2 8.3 vs 8.8, 1376 x 4096 7.5 vs 8.6, 2048 x 4096 6.8 vs 8.6 - but the
    //     elongated 8192 x 2048 10.6 vs 9.7 and 2048 x 8192 11.1 vs 10.1
    //   M = 64: 1376 x 4096 7.8 vs 9.4, 2048 x 4096 8.0 vs 9.5 - but 3072^2 11.8 vs 9.7, 8192 x 2048 12.7 vs 10.9, 2048 x 8192
    //     13.4 vs 11.0, 4096^2 10.8 vs 10.5
    const long longer = N > K ? N : K, shorter = N > K ? K : N;
    if (M <= 48)
        return weights <= (12L << 20) || (weights <= (20L << 20) && longer < 2 * shorter);
    if (M <= 64)
        return weights <= (17L << 19);
    return false;
}
// Which problems go to the streaming MFMA kernel (tuning knob cfg 50 forces it; knob0 bit 1 keeps it out of the built-in route).
// Measured on MI355X (profiles/r6_sm_v3_ab_full.txt), us per launch, streaming MFMA kernel vs what ran before (streaming kernel at
// 2 rows, register-transposed kernel above), M = 2 / 4 / 8 / 16: 4096^2 4.44 / 4.56 / 5.15 / 6.18 vs 4.95 / 5.72 / 6.05 / 6.62; 8192^2
// 10.8 / 10.9 / 12.0 / 14.0 vs 11.2 / 14.1 / 15.0 / 16.0; 11008 x 4096 8.2 / 8.1 / 8.7 / 9.7 vs 8.6 / 11.1 / 11.9 / 13.1; 14336 x 4096
// 9.7 / 10.0 / 10.6 / 11.9 vs 10.3 / 13.6 / 14.6 / 15.6; 5120^2 (blocksize 128) 6.5 / 6.7 / 7.4 / 8.9 vs 6.8 / 8.7 / 9.3 / 10.9; nested
// statistics level with plain ones. Behind only on long rows with more than 8 batch rows (4096 x 11008, M = 9 / 16: 11.9 / 12.2 vs
// 10.7 / 11.9: eight wavefronts there, one chunk switch per item) and at one row (the streaming kernel: 4.20 vs 4.43).
constexpr int kSmMinRows = 128; // (below: a handful of tiles - the kernels of round 5)
bool sm_selected(int M, int N, int K, int knob0, int knob1) {
    const int cfg = knob1 / 100;
    if (cfg == 50)
        return true;
    if (cfg != 0 || (knob0 & 2)) // (knob0 bit 1: the routing as it was before this kernel - A/B runs)
        return false;
    if (M < 2 || N < kSmMinRows)
        return false;
    // rows that are not whole 256-k chunks (K % 64 == 0) are this kernel's alone: its 32-row instances in row passes over grid.y up to
    // 128 rows (profiles/r6_sm_rows32_ab.txt, us, fused vs dequantize + GEMM: 4096 x 2752 M = 64 / 96 / 128 11.5 / 16.7 / 21.2 vs 30.2 /
    // 36.8 / 36.7; 11008 x 1344 13.9 / 20.0 / 25.7 vs 31.6 / 29.9 / 30.2; 14336 x 1088 15.6 / 22.3 / 29.0 vs 30.9 / 32.6 / 28.8; 1376 x 2752
    // 6.6 / 11.1 / 11.5 vs 22.6 / 21.4 / 21.3; the streaming kernel's 4-row passes: 89 us at 64 rows)
    if (K % kKC)
        return M <= 128;
    // matrices that give every CU a 16-row tile (>= ~3/4 of the chip): up to 16 rows, long rows up to 8
    if (N >= 12 * device_cu_count_or_default())
        return M <= 16 && !(M > 8 && K > 2 * N);
    // fewer tiles than CUs (a rank's shard of a projection, small models; profiles/r6_sm_small_n_ab.txt, us against the routing
    // of round 5 = streaming kernel to 4 rows, register-transposed kernel above): 1376 x 4096 M = 2 / 4 / 8 3.81 / 3.83 / 4.11 vs
    // 4.09 / 4.75 / 5.13; 2048 x 4096 3.88 / 3.91 / 4.52 / 5.66 (M = 16) vs 4.32 / 5.12 / 5.40 / 6.01; 2560^2 3.66 / 3.67 / 3.99 /
    // 4.69 vs 4.82 / 5.63 / 4.61 / 5.00; 512 x 4096 3.66 / 3.68 / 4.00 / 5.07 vs 3.90 / 4.18 / 5.79 / 6.01. Behind: two rows on long
    // rows (2048 x 5632 5.12 vs 4.91, 1024 x 8192 5.66 vs 4.64: the streaming kernel splits K over workgroups, this one does not),
    // 5 ... 8 rows on K <= 2048 (level, -3 %) and on 512 x 11008, 9 ... 16 rows on long rows (1280 x 5120 6.35 vs 6.19).
    if (M == 2)
        return K <= 5120;
    if (M <= 4)
        return true;
    if (M <= 8)
        return K >= 2560 && (N >= 1024 || K <= 4096);
    return M <= 16 && (K <= 4096 || K <= 2 * N);
}
// Which problems go to the K-quarter kernel (tuning knob cfg 40 forces it; knob % 100 = K slices).
bool kq_selected(int M, int N, int K, int knob1, int* force_ks) {
    const int cfg = knob1 / 100;
    *force_ks = 0;
    if (cfg == 40) {
        *force_ks = knob1 % 100;
        return true;
    }
    if (cfg != 0)
        return false;
    // measured on MI355X (profiles/r4_kq_ab.txt), us per launch pair, K-quarter vs producer/consumer kernel, M = 17 / 32 / 64:
    // 8192^2 15.3 / 15.9 / 20.0 vs 17.1 / 17.6 / 21.8; 11008 x 4096 13.5 / 14.0 / 17.6 vs 15.1 / 15.5 / 19.3; 4096 x 11008 12.5 /
    // 13.0 / 16.7 vs 14.2 / 14.6 / 17.7; 28672 x 8192 38 / 40 / 47 vs 41 / 42 / 53; 5120^2 and 6144 x 4096 (25 M weights) 0.5 - 0.9
    // ahead; 4096^2 and below: behind the register-transposed kernel (17 ... 32 rows) or level with the producer/consumer one
    // 33 ... 48 rows (a quarter of the second row tile is padding; the producer/consumer kernel has 48-row tiles): 5120^2 13.8 vs
    // 13.2, 8192^2 and 11008 x 4096 level or 0.5 ahead. Taller batches (64-row passes; profiles/r4_route_ab.txt, M = 128 / 256 /
    // 512): 4096^2 13.0 / 17.9 / 24.0 vs 13.5 / 19.7 / 27.5, 8192^2 28.9 / 45.1 / 88.9 vs 32.4 / 52.4 / 101, 11008 x 4096 23.0 /
    // 43.2 / 69.8 vs 26.9 / 50.9 / 80.2, 28672 x 8192 89 / 171 vs 109 / 199 - and level with or ahead of round 3's pre-scaled-operand
    // kernel (same table, column ps), which left the library for it
    const long weights = static_cast<long>(N) * K;
    if (M >= 17 && M <= 32)
        return weights >= (24L << 20);
    if (M >= 33 && M <= 48)
        return weights >= (40L << 20);
    if (M >= 49 && M <= 64)
        return weights >= (24L << 20);
    if (M >= 65)
        return weights >= (16L << 20);
    return false;
}
} // namespace

// Whether the built-in route hands this problem to the streaming MFMA kernel (c_api.hip: such calls take the MFMA route from two
// rows on; pointer alignment is gemm_4bit_mfma_supported's business)
bool gemm_4bit_sm_routes(int dtype, int M, int N, int K, int blocksize) {
    return dtype != 0 && blocksize >= 64 && (K % 64) == 0 &&
           sm_selected(M, N, K, g_mfma_knob0.load(std::memory_order_relaxed), g_mfma_knob1.load(std::memory_order_relaxed));
}

// Preconditions of the MFMA kernels: 16-bit activations, K a multiple of 256, 16-byte aligned A, 8-byte aligned B, and a blocksize
// >= 64 (a 64-k MFMA pair stays inside one quantization block) - or, round 5, blocksize 32 with fp32 absmax (`plain_absmax`: the
// caller knows, the shape-only queries assume it), which the register-transposed kernel's BS32 instances serve at any M.
bool gemm_4bit_mfma_supported(int dtype, const void* A, const uint8_t* B, const float* code16, int M, int N, int K, int blocksize, bool plain_absmax) {
    if (blocksize == 32) // (the BS32 instances' own preconditions: literal code table, 32-bit byte offsets - ADVICE round 5: a call that
                         // passed this test and failed those used to end the process)
        return plain_absmax && gemm_4bit_rt_supported(dtype, A, B, code16, M, N, K, blocksize);
    return dtype != 0 && M >= 1 && N >= 1 && (K % kKC) == 0 && is_pow2(blocksize) && blocksize >= 64 && aligned_to(A, 16) && aligned_to(B, 8);
}

// Bytes of fp32 slab workspace the launch heuristics would like for this problem (0 = none needed).
size_t gemm_4bit_mfma_workspace_bytes(int M, int N, int K, int blocksize) {
    if (M < 1 || N < 1 || K < kKC)
        return 0;
    if (blocksize == 32) // (served by the register-transposed kernel alone, whatever the shape)
        return gemm_4bit_rt_workspace_bytes(M, N, K, 0);
    int fks, fw;
    const int knob1 = g_mfma_knob1.load(std::memory_order_relaxed);
    // (the streaming MFMA kernel needs none; a call it turns down at launch time - a caller-supplied code table, misaligned
    // statistics - runs the kernels below with the library's own buffer or fewer K slices)
    if (blocksize >= 64 && sm_selected(M, N, K, g_mfma_knob0.load(std::memory_order_relaxed), knob1))
        return 0;
    if (K % kKC)
        return 0; // (only the streaming MFMA kernel takes rows that are not whole 256-k chunks)
    int qks;
    size_t kq_bytes = 0;
    if (kq_selected(M, N, K, knob1, &qks)) {
        kq_bytes = gemm_4bit_kq_workspace_bytes(M, N, K, qks);
        // (a call whose statistics or alignment the K-quarter kernel does not serve - gemm_4bit_kq_serves / _supported - runs the
        // kernels below: the query does not know and answers with the larger of the two)
    }
    if (rt_selected(M, N, K, knob1, &fks, &fw)) {
        const size_t b = gemm_4bit_rt_workspace_bytes(M, N, K, fks);
        return b > kq_bytes ? b : kq_bytes;
    }
    const Plan pl = make_plan(M, N, K, knob1);
    const size_t b = pl.ks > 1 ? static_cast<size_t>(pl.ks) * M * N * sizeof(float) : 0;
    return b > kq_bytes ? b : kq_bytes;
}

void gemm_4bit_mfma(int dtype, const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax8,
                    const float* absmax_code, const float* absmax_offset, const float* code16, void* out,
                    const void* bias, int M, int N, int K, int blocksize, int quant_type, void* workspace,
                    size_t workspace_bytes, hipStream_t stream) {
    int fks, fw;
    const int knob0 = g_mfma_knob0.load(std::memory_order_relaxed), knob1 = g_mfma_knob1.load(std::memory_order_relaxed);
    int qks;
    if (blocksize == 32) {
        // blocksize 32: the register-transposed kernel's BS32 instances at any M (row passes over grid.z). The callers
        // route here only what gemm_4bit_mfma_supported(..., plain_absmax) accepted; anything else is a caller's bug.
        if (!gemm_4bit_rt_supported(dtype, A, B, code16, M, N, K, blocksize) || !gemm_4bit_rt_serves(absmax, absmax8, blocksize))
            // (not reachable through route_to_mfma, which asks the same questions; a direct caller gets the streaming kernel, not exit(1))
            return gemv_4bit_stream(dtype, A, B, absmax, absmax8, absmax_code, absmax_offset, code16, out, bias, M, N, K, blocksize, quant_type, stream);
        return gemm_4bit_rt(dtype, A, B, absmax, absmax8, absmax_code, absmax_offset, out, bias, M, N, K, blocksize, quant_type, workspace,
                            workspace_bytes, 0, 0, 0, stream);
    }
    if (knob1 / 100 == 60 && gemm_4bit_tall_supported(dtype, A, B, code16, M, N, K, blocksize)) // (tuning knob cfg 60: the tall-tile kernel)
        return gemm_4bit_tall(dtype, A, B, absmax, absmax8, absmax_code, absmax_offset, out, bias, M, N, K, blocksize, quant_type, knob1 % 100, stream);
    if (sm_selected(M, N, K, knob0, knob1) && gemm_4bit_sm_supported(dtype, A, B, code16, M, N, K, blocksize) && gemm_4bit_sm_serves(absmax, absmax8, blocksize))
        return gemm_4bit_sm(dtype, A, B, absmax, absmax8, absmax_code, absmax_offset, out, bias, M, N, K, blocksize, quant_type, knob0, stream);
    if (K % kKC) // (rows that are not whole 256-k chunks are the streaming MFMA kernel's alone: a call it turned down runs the streaming kernel)
        return gemv_4bit_stream(dtype, A, B, absmax, absmax8, absmax_code, absmax_offset, code16, out, bias, M, N, K, blocksize, quant_type, stream);
    if (kq_selected(M, N, K, knob1, &qks) && gemm_4bit_kq_supported(dtype, A, B, code16, M, N, K, blocksize) &&
        gemm_4bit_kq_serves(absmax, absmax8, blocksize, K))
        return gemm_4bit_kq(dtype, A, B, absmax, absmax8, absmax_code, absmax_offset, out, bias, M, N, K, blocksize, quant_type,
                            workspace, workspace_bytes, qks, knob0, stream);
    if (rt_selected(M, N, K, knob1, &fks, &fw) && gemm_4bit_rt_supported(dtype, A, B, code16, M, N, K, blocksize))
        return gemm_4bit_rt(dtype, A, B, absmax, absmax8, absmax_code, absmax_offset, out, bias, M, N, K, blocksize,
                            quant_type, workspace, workspace_bytes, fks, fw, knob0 & 1, stream);
    GemmArgs p;
    p.A = A;
    p.B = B;
    p.absmax = absmax;
    p.absmax8 = absmax8;
    p.absmax_code = absmax_code;
    p.absmax_offset = absmax_offset;
    p.code16 = code16;
    p.out = out;
    p.bias = bias;
    p.ws = nullptr;
#ifdef BNB_PROFILING
    p.dbg = g_dbg_buf;
#else
    p.dbg = nullptr;
#endif
    p.ablate = 0;
    p.knob0 = knob0;
    p.M = M;
    p.N = N;
    p.K = K;
    p.bs_shift = ilog2(blocksize);
    p.quant_type = quant_type;
    p.kslices = 1;
    p.steps_total = K / kKC;
    if (dtype == 2)
        dispatch_mfma<bf16>(p, static_cast<float*>(workspace), workspace_bytes, knob1, stream);
    else
        dispatch_mfma<f16>(p, static_cast<float*>(workspace), workspace_bytes, knob1, stream);
}

} // namespace bnb
