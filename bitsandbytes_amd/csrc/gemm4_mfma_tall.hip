// gemm4_mfma_tall.hip — tall-batch fused dequantize-GEMM (experiment of round 6; M >= 128, bf16 / fp16, K % 64 == 0):
//   out[m, n] = sum_k A[m, k] * T(code[B[n, k]] * scale[n, k / bs])  (+ bias[n])
//
// The reference keeps its tensor-core kernel to 1536 rows with 128-row tiles and a pre-scaled operand (fp32 product, ONE rounding
// to T: csrc/gemm_4bit_sm80.cu:300-307, 493-689; backends/cuda/ops.py:922-926). Here every batch above 64 rows runs 64-row
// passes of kernels whose bound is the decode's instruction issue - each weight decoded M / 64 times. This kernel decodes a
// weight once per 128 rows:
//
//  * workgroup = 128 rows x 128 columns x all of K; 8 wavefronts: four 64 x 64 output tiles on v_mfma_f32_32x32x16 (64 accumulator
//    registers), each computed by a PAIR of wavefronts that split every 64-k step in two and run it in opposite order (one decodes
//    while the other multiplies: they share a SIMD); K in steps of 64;
//  * A tile (128 x 64, 16 KiB) by LDS-DMA into a 3-stage ring, requested two steps ahead; both tiles live in LDS as
//    [row][64 k] with the 16-byte pieces of a row XOR-swizzled by f(row) (found by exhaustive search over linear maps: fragment
//    reads - ds_read_b128 under gfx950's 16-lane service groups - AND the decoder's ds_write_b128 - 8 contiguous lanes = 4 rows x 2
//    halves - are conflict-free);
//  * the decoded weight tile T(code * scale) (128 x 64, 16 KiB) is written ONCE per step by the workgroup's 512 threads - thread
//    (row n = t / 4, quarter = t % 4) owns 8 packed bytes: one v_perm_b32 + one ds_read_b64 (byte -> two fp32 code values) + one packed
//    multiply + one packed convert per byte - into a 2-stage ring; no scale FMA ever touches an accumulator;
//  * packed weights and scales travel through a 4-deep REGISTER ring (requested ~4.5 steps ahead: HBM latency); every
//    vector-memory operation of the kernel is spelled in asm and every wait is a hand-counted s_waitcnt: the issue pattern per step is
//    identical from the first (virtual) step to the last - items before the start / past the end are out-of-range requests - so the
//    counts are compile-time constants;
//  * one s_barrier per step.
// Arithmetic = dequantize_4bit's (fp32 product rounded once to T) followed by a T x T -> fp32 matrix product: what the unfused
// path (dequantize + library GEMM) computes, up to the order of the fp32 sums.
#include "bnb_common.h"

namespace bnb {

namespace {

using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using i32x4 = __attribute__((ext_vector_type(4))) int;

template <typename T> struct TallMma;
template <> struct TallMma<bf16> {
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
template <> struct TallMma<f16> {
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

constexpr int kTallLut = 65536;   // 256 entries x 32 copies x 8 B (two fp32 code values), at LDS address 0
constexpr int kTallCode2 = 1024;  // nested absmax code
constexpr int kTallStage = 16384; // one [128][64] 16-bit tile
constexpr int kTallAStages = 3, kTallBStages = 2;
constexpr int kTallABase = kTallLut + kTallCode2;
constexpr int kTallBBase = kTallABase + kTallAStages * kTallStage;
constexpr int kTallLds = kTallBBase + kTallBStages * kTallStage;
constexpr int kTallD = 4; // depth of the packed-weight register ring

struct TallArgs {
    const float* absmax_code;
    const float* absmax_offset;
    void* out;
    const void* bias;
};

__device__ __forceinline__ float tall_code_literal(int i, bool fp4) {
    constexpr float nf4[16] = {BNB_NF4_VALUES};
    constexpr float fp4v[16] = {BNB_FP4_VALUES};
    float v = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j)
        v = (i == j) ? (fp4 ? fp4v[j] : nf4[j]) : v;
    return v;
}

__device__ __forceinline__ i32x4 tall_rsrc(const void* base) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    return i32x4{static_cast<int>(a), static_cast<int>((a >> 32) & 0xFFFFu), 0x7FFFFFFF, 0x00020000};
}
// piece swizzle of a tile row: bits (row0 ^ row1) | row2 << 1 | row3 << 2 - found by exhaustive search over the linear maps of the
// row index: the fragment reads (ds_read_b128, 16-lane service groups) AND the decoder's writes (ds_write_b128, 8 contiguous lanes =
// 2 rows x 4 quarters) are bank-conflict-free
__device__ __forceinline__ int tall_swz(int row) { return ((row ^ (row >> 1)) & 1) | (((row >> 2) & 1) << 1) | (((row >> 3) & 1) << 2); }

__device__ __forceinline__ void tall_dma16(i32x4 rs, uint32_t lds, uint32_t voff, uint32_t soff) {
    lds = __builtin_amdgcn_readfirstlane(lds);
    soff = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
}
// register loads spelled in asm: the compiler inserts no wait for them - every wait is written by hand (tall_wait), tied to the
// registers it releases
__device__ __forceinline__ u32x4 tall_load16(i32x4 rs, uint32_t voff, uint32_t soff) {
    u32x4 v;
    soff = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(rs), "s"(soff) : "memory");
    return v;
}
__device__ __forceinline__ u32x2 tall_load8(i32x4 rs, uint32_t voff, uint32_t soff) {
    u32x2 v;
    soff = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(rs), "s"(soff) : "memory");
    return v;
}
__device__ __forceinline__ uint32_t tall_load4(i32x4 rs, uint32_t voff) {
    uint32_t v;
    asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(v) : "v"(voff), "s"(rs) : "memory");
    return v;
}
__device__ __forceinline__ uint32_t tall_load1(i32x4 rs, uint32_t voff) {
    uint32_t v;
    asm volatile("buffer_load_ubyte %0, %1, %2, 0 offen" : "=v"(v) : "v"(voff), "s"(rs) : "memory");
    return v;
}

// grid = (tiles_m * tiles_n); tile id -> (m tile fastest): the workgroups of one XCD (id % 8) share 1 / 8 of A and all of W.
// 8 wavefronts: wavefront w computes output tile (w & 3) = (wm, wn) for K sub-steps {2 kh, 2 kh + 1}, kh = w >> 2 - the two
// wavefronts of a pair (w, w + 4: they share a SIMD) accumulate the two halves of every 64-k step and are added once, at the end.
// The halves run their step in OPPOSITE order (kh = 0: decode, then multiply; kh = 1: multiply, then decode): on every SIMD one
// wavefront feeds the matrix pipe while the other one decodes.
template <typename T, bool NESTED>
__global__ __launch_bounds__(512) void gemm4_mfma_tall_kernel(const void* hot_A, const uint8_t* hot_B, const float* hot_absmax,
                                                             const uint8_t* hot_absmax8, int hot_M, int hot_N, int hot_K,
                                                             int hot_flags /* bs_shift | fp4 << 8 */, int hot_tiles_m, const TallArgs p) {
    constexpr int D = kTallD;
    constexpr int LW = NESTED ? 3 : 2;   // register loads per packed-weight stage
    constexpr int OPS = 2 + LW;          // vector-memory operations a wavefront issues per step
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = hot_M, N = hot_N, K = hot_K;
    const int bs_shift = hot_flags & 31;
    const bool fp4 = (hot_flags >> 8) & 1;
    const int abl = hot_flags >> 16; // (experiment: 1 no decode, 2 no activation DMA, 4 no MFMA, 8 no weight loads - wrong results, timing only)
    const int tile_m = static_cast<int>(blockIdx.x) % hot_tiles_m, tile_n = static_cast<int>(blockIdx.x) / hot_tiles_m;
    const int m0 = tile_m * 128, n0 = tile_n * 128;
    const int steps = K >> 6;

    // ---- per-lane sources
    // activations: instruction i = 2 wave + j of a stage covers rows 8 i ... 8 i + 7; lane = (row 8 i + (lane >> 3), piece' = lane & 7)
    const i32x4 rs_a = tall_rsrc(static_cast<const unsigned char*>(hot_A) + static_cast<size_t>(m0) * K * sizeof(T));
    uint32_t a_voff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = 8 * (2 * wave + j) + (lane >> 3);
        int mr = m0 + row;
        mr = (mr < M ? mr : M - 1) - m0; // rows past the batch: the last row again (never stored)
        a_voff[j] = static_cast<uint32_t>(mr) * static_cast<uint32_t>(K) * 2u + static_cast<uint32_t>(((lane & 7) ^ tall_swz(row)) << 4);
    }
    // packed weights: thread (row n = tid / 4, quarter = tid % 4): 8 bytes = k [16 quarter, + 16) of the step's 64
    const int nl = tid >> 2, quarter = tid & 3;
    int nrow = n0 + nl;
    nrow = nrow < N ? nrow : N - 1;
    // (descriptors are wave-uniform: the row offset goes into the per-lane offset, relative to the tile's first row)
    const int n0c = n0 < N ? n0 : N - 1;
    const i32x4 rs_wt = tall_rsrc(hot_B + static_cast<size_t>(n0c) * (K >> 1));
    const uint32_t w_voff = static_cast<uint32_t>(nrow - n0c) * static_cast<uint32_t>(K >> 1) + static_cast<uint32_t>(quarter * 8);
    const i32x4 rs_s = tall_rsrc(hot_absmax);
    [[maybe_unused]] const i32x4 rs_q = tall_rsrc(hot_absmax8);
    const uint32_t e_row = static_cast<uint32_t>(nrow) * static_cast<uint32_t>(K); // flat element index of the row start (N K < 2^32)
    constexpr uint32_t kOob = 0xFFFFFFF0u;

    struct WStage {
        u32x2 w;
        uint32_t s, s2;
    };
    WStage ws[D];
    // request the packed weights + scale of step j (j < 0 or >= steps: out of range - nothing fetched)
    auto issue_w = [&](WStage& st, int j) {
        const uint32_t inval = (j >= 0 && j < steps) ? 0u : kOob;
        const uint32_t jj = static_cast<uint32_t>(j < 0 ? 0 : j);
        st.w = tall_load8(rs_wt, w_voff | inval, jj * 32u);
        const uint32_t blk = (e_row + (jj << 6)) >> bs_shift;
        if constexpr (NESTED) {
            st.s = tall_load1(rs_q, blk | inval);
            st.s2 = tall_load4(rs_s, ((blk >> 8) << 2) | inval);
        } else {
            st.s = tall_load4(rs_s, (blk << 2) | inval);
            st.s2 = 0;
        }
    };
    auto issue_a = [&](int j, int stage) {
        const uint32_t inval = (j >= 0 && j < steps) ? 0u : kOob;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            tall_dma16(rs_a, static_cast<uint32_t>(kTallABase + stage * kTallStage + (2 * wave + i) * 1024), a_voff[i] | inval,
                       static_cast<uint32_t>(j < 0 ? 0 : j) * 128u);
    };

    // ---- decode table: entry e (a packed byte) = 32 copies of (code[e >> 4], code[e & 15]) in fp32, 256 B per entry
    {
        const float cv = tall_code_literal((lane & 15) + opaque_zero(), fp4);
        const int cvb = __builtin_bit_cast(int, cv);
        using f32x4 = __attribute__((ext_vector_type(4))) float;
        const int e = tid & 255, hf = tid >> 8;
        const float hi = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((e >> 4) & 15) * 4, cvb));
        const float lo = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((e & 15) * 4, cvb));
        const f32x4 v = {hi, lo, hi, lo};
        f32x4* const dst = reinterpret_cast<f32x4*>(smem + e * 256);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            dst[(8 * hf + j + e) & 15] = v;
    }
    float offset = 0.0f;
    if constexpr (NESTED) {
        if (tid < 256)
            reinterpret_cast<float*>(smem + kTallLut)[tid] = p.absmax_code[tid];
        offset = p.absmax_offset[0];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the only compiler-visible loads of the kernel: drained before the counted stream starts)
    }
    if (reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)smem) != 0)
        __builtin_trap(); // the table is addressed with raw v_perm_b32 results

    // ---- consumer addresses
    const int kh = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    const int r32 = lane & 31, kg = lane >> 5;
    uint32_t a_rd[2], b_rd[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ra = 64 * wm + 32 * i + r32, rb = 64 * wn + 32 * i + r32;
        a_rd[i] = static_cast<uint32_t>(ra * 128 + ((kg ^ tall_swz(ra)) << 4));
        b_rd[i] = static_cast<uint32_t>(rb * 128 + ((kg ^ tall_swz(rb)) << 4));
    }
    const uint32_t b_wr = static_cast<uint32_t>(nl * 128);
    const int b_swz = tall_swz(nl);
    const uint32_t lane8 = static_cast<uint32_t>(lane & 31) * 8u;
    const uint32_t perm_sel = 0x0C0C0400u + static_cast<uint32_t>(opaque_zero());

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q)
                acc[i][j][q] = 0.0f;

    // decode a ring slot (the thread's 8 packed bytes of a step) into B stage `stage`
    auto decode = [&](const WStage& st, int stage) {
        float scale;
        if constexpr (NESTED)
            scale = nested_scale(*reinterpret_cast<const __attribute__((address_space(3))) float*>(static_cast<uint32_t>(kTallLut) + (st.s & 0xFFu) * 4u),
                                 __builtin_bit_cast(float, st.s2), offset);
        else
            scale = __builtin_bit_cast(float, st.s);
        f32x2 pr[8];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                pr[4 * j + q] = *reinterpret_cast<const __attribute__((address_space(3))) f32x2*>(__builtin_amdgcn_perm(st.w[j], lane8, perm_sel + (q << 8)));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            u32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                o[q] = TallMma<T>::pack(rounded_f32(pr[4 * j + q][0] * scale), rounded_f32(pr[4 * j + q][1] * scale));
            *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(static_cast<uint32_t>(kTallBBase + stage * kTallStage) + b_wr +
                                                                         static_cast<uint32_t>(((2 * quarter + j) ^ b_swz) << 4)) = o;
        }
    };
    // this wavefront's two K sub-steps of a step
    auto mma_step = [&](int a_stage, int b_stage) {
        const uint32_t ab = static_cast<uint32_t>(kTallABase + a_stage * kTallStage), bb = static_cast<uint32_t>(kTallBBase + b_stage * kTallStage);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const uint32_t ks = static_cast<uint32_t>(2 * kh + kk);
            u32x4 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>((ab + a_rd[i]) ^ (ks << 5));
                bf[i] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>((bb + b_rd[i]) ^ (ks << 5));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = TallMma<T>::run(af[i], bf[j], acc[i][j]);
        }
    };

    // ---- the step loop, from virtual step -D - 1 (requests only) on: every step issues [A stage of step s + 2][weights of step
    // s + 1 + D], so the queue looks the same at every point of the loop and the waits are constants:
    //   weights of step s + 1 (requested in step s - D): (D - 1) whole steps of requests are younger;
    //   A stage of step s + 1 (requested in step s - 1, in front of that step's weight request): LW + 2 + LW requests are younger.
    __syncthreads(); // the decode table (and the nested code) are in place
    int a_req = 1;   // A stage of step s + 2 (rotates mod 3)
    for (int s0 = -D - 1; s0 < steps; s0 += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int s = s0 + u;
            WStage& wnext = ws[u]; // (s + 1) mod D == u: the slot of step s + 1, refilled with step s + 1 + D
            const int a_cur = a_req == 2 ? 0 : a_req + 1; // stage of step s + 3 == stage of step s
            const bool have = s >= 0 && s < steps, next = s + 1 >= 0 && s + 1 < steps;
            if (kh == 1 && have && !(abl & 4))
                mma_step(a_cur, s & 1);
            if (next) {
                asm volatile("s_waitcnt vmcnt(%4)" : "+v"(wnext.w), "+v"(wnext.s), "+v"(wnext.s2), "+v"(offset) : "n"((D - 1) * OPS) : "memory");
                if (!(abl & 1))
                    decode(wnext, (s + 1) & 1);
            }
            issue_a((abl & 2) ? -1 : s + 2, a_req);
            a_req = a_cur;
            issue_w(wnext, (abl & 8) ? -1 : s + 1 + D);
            if (kh == 0 && have && !(abl & 4))
                mma_step(a_cur, s & 1);
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LW + 2 + LW) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- the two K halves of every output tile: wavefronts 4 ... 7 park theirs in LDS (the rings are idle now), 0 ... 3 add them
    {
        float* const park = reinterpret_cast<float*>(smem + kTallABase) + (wave & 3) * 4096; // 64 accumulators x 64 lanes
        if (kh == 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        park[((2 * i + j) * 16 + q) * 64 + lane] = acc[i][j][q];
        }
        __syncthreads();
        if (kh == 1)
            return;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    acc[i][j][q] += park[((2 * i + j) * 16 + q) * 64 + lane];
    }

    // ---- epilogue: C layout of v_mfma_f32_32x32x16: column = lane % 32, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const T* const bias = static_cast<const T*>(p.bias);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + 64 * wn + 32 * j + r32;
        const float bv = (bias != nullptr && n < N) ? static_cast<float>(bias[n]) : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int m = m0 + 64 * wm + 32 * i + (q & 3) + 8 * (q >> 2) + 4 * kg;
                if (m < M && n < N)
                    static_cast<T*>(p.out)[static_cast<long>(m) * N + n] = static_cast<T>(acc[i][j][q] + bv);
            }
    }
}

} // namespace

bool gemm_4bit_tall_supported(int dtype, const void* A, const uint8_t* B, const float* code16, int M, int N, int K, int blocksize) {
    return dtype != 0 && code16 == nullptr && M >= 1 && N >= 1 && K >= 64 && (K % 64) == 0 && blocksize >= 64 && is_pow2(blocksize) &&
           aligned_to(A, 16) && aligned_to(B, 16) && static_cast<long long>(N) * K < (1LL << 32) && K < (1 << 22);
}

// dtype: 1 = f16, 2 = bf16
void gemm_4bit_tall(int dtype, const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax8, const float* absmax_code,
                    const float* absmax_offset, void* out, const void* bias, int M, int N, int K, int blocksize, int quant_type,
                    int ablate, hipStream_t stream) {
    g_last_gemm_kernel = kKernelTall;
    TallArgs a{absmax_code, absmax_offset, out, bias};
    const int tiles_m = (M + 127) / 128, tiles_n = (N + 127) / 128;
    const int flags = ilog2(blocksize) | ((quant_type == kFP4) ? 256 : 0) | (ablate << 16);
    const dim3 grid(static_cast<unsigned>(tiles_m) * static_cast<unsigned>(tiles_n));
#define BNB_TALL_LAUNCH(TT, NE)                                                                                                    \
    {                                                                                                                              \
        auto kern = gemm4_mfma_tall_kernel<TT, NE>;                                                                               \
        static LdsLimit lim;                                                                                                       \
        ensure_dynamic_lds(lim, reinterpret_cast<const void*>(kern), kTallLds);                                                    \
        hipLaunchKernelGGL(kern, grid, dim3(512), kTallLds, stream, A, B, absmax, absmax8, M, N, K, flags, tiles_m, a);            \
    }
    if (dtype == 2) {
        if (absmax8 != nullptr)
            BNB_TALL_LAUNCH(bf16, true)
        else
            BNB_TALL_LAUNCH(bf16, false)
    } else {
        if (absmax8 != nullptr)
            BNB_TALL_LAUNCH(f16, true)
        else
            BNB_TALL_LAUNCH(f16, false)
    }
#undef BNB_TALL_LAUNCH
    BNB_CHECK_LAUNCH();
}

} // namespace bnb
