// gemm4_grad_input.hip — fused backward of the 4-bit linear layer on gfx950:
//     grad_A[m, k] = sum_n grad_out[m, n] * T(code[B[n, k]] * scale[n, k / bs])          (bf16 / fp16)
//
// The reference computes this as dequantize_4bit(B) -> [N, K] in T, then a dense matmul (autograd/_functions.py:365-386):
// (0.5 + 2 + 2) bytes per weight through HBM against 0.56 when the dequantized tile never leaves the CU. Here the weight
// tile is dequantized into LDS - the same arithmetic as dequantize_4bit: fp32 product, ONE rounding to T - and multiplied
// from there, so the result equals the unfused path up to the order of the fp32 sums.
//
// Why this is not the forward kernel with the operands swapped: the contraction now runs over n, the weight ROW index, while
// the packed format keeps k contiguous and the scale is indexed by (n, k / bs) - it varies along the contraction, so it
// cannot be applied to a partial tile after the matrix instruction and has to be folded into the operand. The MFMA B
// operand wants, per lane, 8 consecutive n at one k: a column of the row-major tile. gfx950's ds_read_b64_tr_b16 delivers
// exactly that from a row-major LDS image (semantics probed on the device, tools/ubench/tr_probe.hip: inside a 16-lane
// group lane 4 j + q addresses the 8-byte piece q of row j, and lane i receives column i of the resulting 4 x 16 block).
//
//  * A workgroup (8 wavefronts, two per SIMD: the decode phase of one overlaps the LDS / MFMA latencies of the other) owns
//    128 k-columns x 64 batch rows x a slice of N and walks it in steps of 64 n.
//  * Per step every thread loads 8 packed bytes (16 k of one weight row; eight threads cover a row's 64 bytes) and its
//    scale, decodes through the bank-private byte -> (code[hi], code[lo]) fp32 table (one v_perm_b32 + one ds_read_b64 per
//    byte), multiplies, rounds to T and stores 32 bytes of the [64 n][128 k] tile; the 64 x 64 grad_out tile is staged
//    through LDS as well (coalesced 16-byte pieces in, 16-byte MFMA A fragments out). The loads of steps s + 1 .. s + 3 are
//    in flight while step s is multiplied (register ring); tiles are double-buffered: one barrier per step.
//  * Wavefront w multiplies its 16 columns (one MFMA column tile) with the four 16-row batch tiles: 8 MFMAs per step.
//  * N slices across workgroups fill the chip; fp32 slabs are added in slice order by gemm4_finalize: bit-reproducible.
#include "bnb_common.h"

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

constexpr int kGiCols = 128;      // k-columns per workgroup
constexpr int kGiRows = 64;       // batch rows per workgroup (4 MFMA row tiles)
constexpr int kGiStep = 64;       // n per step (two MFMA k-steps)
constexpr int kGiLut = 65536;     // 256 entries x 32 copies x 8 B (fp32 pair), at LDS address 0
constexpr int kGiWStride = 288;   // bytes per row of the dequantized tile: 256 + 32 (consecutive rows in different banks)
constexpr int kGiGStride = 128;   // bytes per row of the grad_out tile (16-byte chunk c of row m stored at c ^ (m & 7))
constexpr int kGiWTile = kGiStep * kGiWStride;
constexpr int kGiGTile = kGiRows * kGiGStride;
constexpr int kGiLds = kGiLut + 2 * kGiWTile + 2 * kGiGTile + 1024;

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

constexpr int kGiThreads = 512;

// grid = (K / 128, nslices, ceil(M / 64)); 512 threads
template <typename T, bool NESTED>
__global__ __launch_bounds__(kGiThreads) void gemm4_grad_input_kernel(
    const void* hot_G, const uint8_t* hot_B, const float* hot_absmax, const uint8_t* hot_absmax8, int hot_M, int hot_N,
    int hot_K, int hot_flags /* bs_shift | fp4 << 8 */, int hot_sps /* steps per N slice */, int hot_nslices, const GiArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = hot_M, N = hot_N, K = hot_K;
    const int bs_shift = hot_flags & 31;
    const bool fp4 = (hot_flags >> 8) & 1;
    BNB_GI_STAMP(0)
    const int k0 = blockIdx.x * kGiCols;
    const int m_base = blockIdx.z * kGiRows;
    const int steps_total = N / kGiStep;
    const int sb = blockIdx.y * hot_sps;
    int se = sb + hot_sps;
    se = se < steps_total ? se : steps_total;

    unsigned char* const wtiles = smem + kGiLut;
    unsigned char* const gtiles = wtiles + 2 * kGiWTile;
    float* const code2 = reinterpret_cast<float*>(gtiles + 2 * kGiGTile);

    // ---- per-thread roles of the loads: weight row (tid >> 3) of the step, 8-byte piece (tid & 7) = 16 k = one MFMA
    // column tile; grad_out row (tid >> 3) of the batch tile, 16-byte piece (tid & 7) = 8 n
    const int wr = tid >> 3, wp = tid & 7;
    const uint8_t* const wsrc = hot_B + static_cast<long>(wr) * (K >> 1) + ((k0 + 16 * wp) >> 1);
    const long we0 = static_cast<long>(wr) * K + k0 + 16 * wp; // flat element index of the piece at n = 0
    int grow = m_base + wr;
    grow = grow < M ? grow : M - 1; // rows past the end of the batch re-read the last row: never stored
    const T* const gsrc = static_cast<const T*>(hot_G) + static_cast<long>(grow) * N + 8 * wp;

    struct Stage {
        u32x2 w;
        uint32_t s, s2;
        u32x4 g;
    };
    // (the step index is clamped to the slice: a harmless re-read at the tail instead of a branch around the loads, so the
    // compiler's counted waits stay exact)
    auto issue = [&](Stage& st, int step_unclamped) {
        const int step = step_unclamped < se ? step_unclamped : se - 1;
        const long n0 = static_cast<long>(step) * kGiStep;
        st.w = *reinterpret_cast<const u32x2*>(wsrc + n0 * (K >> 1));
        const long e = we0 + n0 * K;
        if constexpr (NESTED) {
            st.s = hot_absmax8[e >> bs_shift];
            st.s2 = __builtin_bit_cast(uint32_t, hot_absmax[(e >> bs_shift) >> 8]);
        } else {
            st.s = __builtin_bit_cast(uint32_t, hot_absmax[e >> bs_shift]);
            st.s2 = 0;
        }
        st.g = *reinterpret_cast<const u32x4*>(gsrc + n0);
    };

    // a three-deep register ring: the loads of step s + 3 are issued while step s is multiplied (one step of compute is far
    // shorter than an HBM round trip)
    constexpr int D = 3;
    if (sb >= se)
        return; // (never: the host makes every N slice non-empty)
    Stage st[D];
#pragma unroll
    for (int j = 0; j < D; ++j)
        issue(st[j], sb + j);
    BNB_GI_STAMP(1)
    __builtin_amdgcn_sched_barrier(0);

    // ---- decode table, built while the first loads fly: entry e (a packed byte) = 32 copies of (code[e >> 4], code[e & 15])
    // in fp32, 256 B per entry; thread (half, e) writes 8 of its 16 chunks in an order rotated by e (eight lanes -> eight
    // bank quads)
    {
        const float cv = gi_code_literal((lane & 15) + opaque_zero(), fp4);
        const int cvb = __builtin_bit_cast(int, cv);
        const int e = tid & 255, half = tid >> 8;
        const float hi = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((e >> 4) & 15) * 4, cvb));
        const float lo = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((e & 15) * 4, cvb));
        const f32x4 v = {hi, lo, hi, lo};
        f32x4* const dst = reinterpret_cast<f32x4*>(smem + e * 256 + half * 128);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            dst[(j + e) & 7] = v;
    }
    float offset = 0.0f;
    if constexpr (NESTED) {
        if (tid < 256)
            code2[tid] = p.absmax_code[tid];
        offset = p.absmax_offset[0];
    }
    __syncthreads();
    BNB_GI_STAMP(2)
    if (reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)smem) != 0)
        __builtin_trap(); // the table is addressed with raw v_perm_b32 results: it must sit at LDS address 0

    const uint32_t perm_sel = 0x0C0C0400u + static_cast<uint32_t>(opaque_zero()); // {lane offset, weight byte, 0, 0}
    const uint32_t lane_off = static_cast<uint32_t>(lane & 31) * 8u;
    const int ln = lane & 15, lg = lane >> 4;

    // dequantize this thread's 32 weights and store them (and its grad_out pieces) into LDS buffer `buf`
    auto stage_to_lds = [&](const Stage& s, int buf) {
        float scale;
        if constexpr (NESTED) {
            const uint32_t q = s.s, a2 = s.s2;
            scale = __fadd_rn(__fmul_rn(code2[q & 0xFFu], __builtin_bit_cast(float, a2)), offset);
        } else {
            const uint32_t sv = s.s;
            scale = __builtin_bit_cast(float, sv);
        }
        // (the 32-byte column blocks of rows 8..15 (mod 16) are stored 4 blocks away: the transpose read serves 32 lanes per
        // pass = rows j and 8 + j of one column block, which a plain row stride puts into the same banks)
        unsigned char* const wrow = wtiles + buf * kGiWTile + wr * kGiWStride + ((wp ^ (4 * ((wr >> 3) & 1))) * 32);
        // all 8 look-ups in flight before the first product
        f32x2 pr[8];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const uint32_t w = s.w[d];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                pr[4 * d + q] = *reinterpret_cast<const __attribute__((address_space(3))) f32x2*>(
                    __builtin_amdgcn_perm(w, lane_off, perm_sel + (q << 8)));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            u32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // the reference's dequantize rounds the fp32 product to T once (csrc/cpu_ops.cpp:419-431). bf16 has no fused
                // multiply-convert on gfx950, so the plain expression already is "product in fp32, then one rounding"; for fp16
                // hipcc would fuse it into v_fma_mix*_f16 (one rounding of the exact product): an opaque (non-volatile: it may
                // be scheduled freely) register copy keeps the two steps apart
                f32x2 pv = pr[4 * d + q] * f32x2{scale, scale};
                if constexpr (sizeof(T) == 2 && !__is_same(T, bf16)) {
                    float p0 = pv[0], p1 = pv[1];
                    asm("" : "+v"(p0));
                    asm("" : "+v"(p1));
                    pv = f32x2{p0, p1};
                }
                o[q] = GiMma<T>::pack(pv[0], pv[1]);
            }
            *reinterpret_cast<u32x4*>(wrow + d * 16) = o;
        }
        // grad_out tile: 16-byte chunk c of row m at position c ^ (m & 7) - conflict-free for the 8-contiguous-lane groups of
        // ds_write_b128 here and for the 16-lane groups {0-3, 12-15, 20-27}, ... of the ds_read_b128 fragment reads below
        *reinterpret_cast<u32x4*>(gtiles + buf * kGiGTile + wr * kGiGStride + ((wp ^ (wr & 7)) << 4)) = s.g;
    };

    f32x4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
        acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane LDS addresses: transpose read - lane (i = ln, group lg) addresses row 8 lg + (i >> 2), 8-byte piece (i & 3) of a
    // 16-column block; A fragment - row ln of a batch tile, 16 bytes at n = 8 lg
    // (column block = wave, stored at block ^ 4 for rows with bit 3 set, i.e. for odd lane groups)
    const uint32_t tr_lane = static_cast<uint32_t>((8 * lg + (ln >> 2)) * kGiWStride + (ln & 3) * 8 +
                                                   ((static_cast<uint32_t>(wave) ^ static_cast<uint32_t>((lg & 1) * 4)) << 5));
    const uint32_t a_row = static_cast<uint32_t>(ln * kGiGStride); // + ((4 ks + lg) ^ (row & 7)) << 4

    auto do_step = [&](Stage& stg, int step) {
        const int buf = (step - sb) & 1;
        stage_to_lds(stg, buf);
        if (step - sb < 3)
            BNB_GI_STAMP(3 + 3 * (step - sb))
        __syncthreads();
        if (step - sb < 3)
            BNB_GI_STAMP(4 + 3 * (step - sb))
        issue(stg, step + D);
        const uint32_t wbase = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)(wtiles + buf * kGiWTile)));
        const unsigned char* const gb = gtiles + buf * kGiGTile;
        // both k-steps' fragments are requested before the first MFMA (one exposed LDS round trip per step instead of two)
        u32x4 bf[2], af[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                u32x2 v;
                const uint32_t addr = wbase + tr_lane + static_cast<uint32_t>((32 * ks + 4 * h) * kGiWStride);
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
                bf[ks][2 * h] = v[0];
                bf[ks][2 * h + 1] = v[1];
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                af[ks][mt] = *reinterpret_cast<const u32x4*>(gb + mt * 16 * kGiGStride + a_row + (((4 * ks + lg) ^ (ln & 7)) << 4));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the transpose reads are invisible to the compiler's counters
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                acc[mt] = GiMma<T>::run(af[ks][mt], bf[ks], acc[mt]);
        if (step - sb < 3)
            BNB_GI_STAMP(5 + 3 * (step - sb))
    };
    // whole rounds of D steps with nothing conditional around the loads (a branch around a load makes the compiler merge
    // the pending-load state of both paths at the join and wait with vmcnt(0): that drained the prefetch ring every step),
    // then the tail
    int base = sb;
    for (; base + D <= se; base += D) {
#pragma unroll
        for (int j = 0; j < D; ++j)
            do_step(st[j], base + j);
    }
#pragma unroll
    for (int j = 0; j < D; ++j)
        if (base + j < se)
            do_step(st[j], base + j);

    BNB_GI_STAMP(12)
    // ---- store. Lane (i = ln, lg) of row tile mt holds rows 16 mt + 4 lg + q of column 16 wave + i: stored as it sits, that
    // is 16 four-byte stores per lane, 64 bytes contiguous each (the first version spent 4100 of its 24 k cycles there). Each
    // PAIR of wavefronts' [64 rows][32 columns] fp32 tile goes through LDS instead (the step tiles are free now) and leaves
    // as 16 bytes per lane, 128 contiguous bytes per row.
    __syncthreads(); // every wavefront is done reading the step tiles
    {
        constexpr int kOutStride = 144; // bytes per row of a staged [64][32] tile: rows 4 apart land 16 banks apart
        unsigned char* const ot = wtiles + (wave >> 1) * (kGiRows * kOutStride);
        static_assert(4 * kGiRows * kOutStride <= 2 * kGiWTile, "the staged output tiles fit over the step tiles");
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float*>(ot + (16 * mt + 4 * lg + q) * kOutStride + (16 * (wave & 1) + ln) * 4) = acc[mt][q];
        __syncthreads(); // (the pair's other wavefront wrote the other 16 columns)
        const int rr = lane >> 3, cq = lane & 7;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int row = 32 * (wave & 1) + 8 * ps + rr; // each wavefront of the pair stores half of the rows
            const f32x4 v = *reinterpret_cast<const f32x4*>(ot + row * kOutStride + cq * 16);
            const int m = m_base + row;
            if (m < M) {
                const long idx = static_cast<long>(m) * K + k0 + 32 * (wave >> 1) + 4 * cq;
                if (hot_nslices == 1) {
                    using T4 = __attribute__((ext_vector_type(4))) T;
                    T4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        o[e] = static_cast<T>(v[e]);
                    *reinterpret_cast<T4*>(static_cast<T*>(p.out) + idx) = o;
                } else {
                    *reinterpret_cast<f32x4*>(p.ws + static_cast<long>(blockIdx.y) * M * K + idx) = v;
                }
            }
        }
    }
    BNB_GI_STAMP(13)
}

int gi_cu_count() { return device_cu_count_or_default(); }

struct GiPlan {
    int ns, sps;
};
// N slices to fill the chip (one workgroup per CU), at least two steps each
GiPlan gi_plan(int M, int N, int K) {
    GiPlan pl;
    const int steps = N / kGiStep;
    const int wgs = (K / kGiCols) * ((M + kGiRows - 1) / kGiRows);
    int ns = gi_cu_count() / wgs;
    const int max_ns = steps / 2 > 0 ? steps / 2 : 1;
    ns = ns > max_ns ? max_ns : ns;
    ns = ns < 1 ? 1 : ns;
    pl.sps = (steps + ns - 1) / ns;
    pl.ns = (steps + pl.sps - 1) / pl.sps;
    return pl;
}

template <typename T> void gi_launch(const void* G, const uint8_t* B, const float* absmax, const uint8_t* absmax8, int M, int N, int K,
                                     int flags, const GiPlan& pl, const GiArgs& a, hipStream_t stream) {
    dim3 grid(K / kGiCols, pl.ns, (M + kGiRows - 1) / kGiRows);
    if (absmax8 != nullptr) {
        auto kern = gemm4_grad_input_kernel<T, true>;
        static LdsLimit lim;
        ensure_dynamic_lds(lim, reinterpret_cast<const void*>(kern), kGiLds);
        hipLaunchKernelGGL(kern, grid, dim3(kGiThreads), kGiLds, stream, G, B, absmax, absmax8, M, N, K, flags, pl.sps, pl.ns, a);
    } else {
        auto kern = gemm4_grad_input_kernel<T, false>;
        static LdsLimit lim;
        ensure_dynamic_lds(lim, reinterpret_cast<const void*>(kern), kGiLds);
        hipLaunchKernelGGL(kern, grid, dim3(kGiThreads), kGiLds, stream, G, B, absmax, absmax8, M, N, K, flags, pl.sps, pl.ns, a);
    }
}

} // namespace

// Preconditions of the fused backward: 16-bit gradients, whole 128-column tiles and 64-row steps, blocksize >= 64.
bool gemm_4bit_grad_input_supported(int dtype, const void* G, const uint8_t* B, int M, int N, int K, int blocksize) {
    return (dtype == 1 || dtype == 2) && M >= 1 && N >= kGiStep && (N % kGiStep) == 0 && K >= kGiCols && (K % kGiCols) == 0 &&
           blocksize >= 64 && is_pow2(blocksize) && aligned_to(G, 16) && aligned_to(B, 16);
}

size_t gemm_4bit_grad_input_workspace_bytes(int M, int N, int K) {
    if (M < 1 || N < kGiStep || K < kGiCols)
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
        const int steps = N / kGiStep;
        const int ns = fit >= 2 ? fit : 1;
        pl.sps = (steps + ns - 1) / ns;
        pl.ns = (steps + pl.sps - 1) / pl.sps;
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
