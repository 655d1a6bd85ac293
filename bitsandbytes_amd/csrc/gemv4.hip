// gemv4.hip — fused 4-bit dequantize + dot-product kernel for decode-sized batches (M <= 4 rows of A
// per pass) on gfx950.   out[m, n] = sum_k A[m, k] * code[B[n, k]] * scale[n, k / bs]  (+ bias[n])
//
// Replaces, on MI355X, the reference's kgemm_4bit_inference_naive (csrc/kernels.cu:1452-1567) and the
// small-M range of gemm_4bit_simt (csrc/gemm_4bit_simt.cu:109-480). It is not a translation of
// either: those map one logical 32-lane warp to an output column and decode nibbles with shifts and
// an LDS/const table per nibble; this kernel is built around three CDNA4 facts:
//
//  * HBM-bound, so the only job is to keep ~8 MB of loads in flight. A wavefront owns RPW whole
//    weight rows; lane l reads bytes [16 l, 16 l + 16) of each 1 KiB row segment with one
//    global_load_dwordx4, i.e. every load instruction covers 1 KiB contiguous (eight full 128-B
//    lines). All loads of an iteration (and of the next one) are issued before any arithmetic.
//  * The VALU budget at 8 TB/s is ~5 lane-ops per nibble, so nibbles are never decoded one by one:
//    a 256-entry table maps a packed BYTE straight to the pair (code[hi], code[lo]) as packed
//    bf16x2 / f16x2, and one v_dot2c_f32_{bf16,f16} consumes the pair against two activations:
//    2 VALU + 1 LDS read per byte.
//  * A random 4-byte LDS gather would be ~4-way bank-conflicted. The table is therefore stored
//    32x replicated, entry e of copy j at dword e*32 + j, and lane l only ever reads copy l%32:
//    each lane owns its bank, so the gather is conflict-free for any data. 32 KiB of the CU's
//    160 KiB LDS buys a 2-cycles-per-64-bytes decode. The table is built from 16 SGPR-resident
//    code values (no vector memory traffic) while the first weight loads are in flight.
//
// The per-block scale multiplies the fp32 partial sum of each 32-nibble run (one run never
// straddles a quantization block because blocksize >= 32 and K % 32 == 0), so absmax is applied in
// full fp32. Nested (double-quantized) absmax is fused:
//   scale = absmax_code[absmax_8bit[b]] * absmax[b >> 8] + offset   (reference autograd/_functions.py:471-485).
#include "bnb_common.h"

namespace bnb {

int g_dot_ablate = 0; // profiling only: see the ablation bits of DotFlags
int g_dot_flags = 0;  // kWaves8 | kNT | kXLds selection (0 = default)
unsigned long long* g_dbg_buf = nullptr; // profiling only: device buffer for s_memtime stamps

// Tuning knobs (overridable for sweeps through bnb_mi355x_set_tuning; see c_api.hip).
int g_dot_rpw = 0;  // rows per wavefront, 0 = heuristic
int g_dot_segs = 0; // 2048-k segments per iteration, 0 = heuristic

namespace {

using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

template <typename T> struct Pair2;
template <> struct Pair2<bf16> {
    static __device__ __forceinline__ uint32_t pack(float lo, float hi) {
        using V = __attribute__((ext_vector_type(2))) bf16;
        V v;
        v[0] = static_cast<bf16>(lo);
        v[1] = static_cast<bf16>(hi);
        return __builtin_bit_cast(uint32_t, v);
    }
    static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
        using V = __attribute__((ext_vector_type(2))) bf16;
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(V, a), __builtin_bit_cast(V, b), c, false);
    }
};
template <> struct Pair2<f16> {
    static __device__ __forceinline__ uint32_t pack(float lo, float hi) {
        using V = __attribute__((ext_vector_type(2))) f16;
        V v;
        v[0] = static_cast<f16>(lo);
        v[1] = static_cast<f16>(hi);
        return __builtin_bit_cast(uint32_t, v);
    }
    static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
        using V = __attribute__((ext_vector_type(2))) f16;
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(V, a), __builtin_bit_cast(V, b), c, false);
    }
};

struct GemvArgs {
    const void* A;              // [M, K] activations, row-major
    const uint8_t* B;           // packed [N, K/2]
    const float* absmax;        // fp32 [N*K/bs]   (nested: fp32 [ceil(N*K/bs/256)])
    const uint8_t* absmax8;     // nested only: uint8 [N*K/bs]
    const float* absmax_code;   // nested only: fp32 [256]
    const float* absmax_offset; // nested only: fp32 scalar
    const float* code16;        // optional caller-supplied 16-entry code (gemv_4bit op), else NULL
    void* out;                  // [M, N]
    const void* bias;           // optional [N]
    int M, N, K;
    int bs_shift;
    int quant_type;
    float code[16];             // the 16 code values by value (kernarg -> SGPRs) when code16 == NULL
    unsigned long long* dbg;    // profiling builds only: per-wavefront s_memtime stamps (8 per wave), else NULL
};

constexpr int kSegK = 2048; // k covered by one wavefront-wide 16-byte load

// Compile-time switches of the dot kernel, packed into one template int.
enum DotFlags : int {
    kSingle = 1,  // the whole K fits one iteration: no prefetch registers are allocated
    kNested = 2,  // double-quantised absmax reconstructed in-kernel
    kWaves8 = 4,  // 512-thread workgroups (8 wavefronts share one table build) instead of 256
    kNT = 8,      // non-temporal weight / absmax loads (streamed once, keep them out of the way of x)
    kXLds = 16,   // activations staged once per workgroup in LDS instead of per-wave global loads (kSingle only)
    kCodePtr = 32, // code table read from a table pointer (device-resident built-in table or the caller's)
    kWaves16 = 64, // 1024-thread workgroups: one table build per CU shared by 16 wavefronts
    // bits 8..: ablation for profiling builds (results are wrong): 1 = stream + reduce raw words, no decode;
    // 2 = no table build; 3 = no weight loads; 4 = weights only (no x / absmax traffic); 5 = empty kernel
};

template <typename T> __device__ __forceinline__ T ld_stream(const T* p, bool nt) {
    return nt ? __builtin_nontemporal_load(p) : *p;
}

// T in {bf16, f16}; MB = activation rows per pass; RPW = weight rows per wavefront;
// SEGS = 2048-k sub-segments per loop iteration.
template <typename T, int MB, int RPW, int SEGS, int FLAGS>
__global__ __launch_bounds__((FLAGS & kWaves16) ? 1024 : (FLAGS & kWaves8) ? 512 : 256) void gemv4_dot_kernel(const GemvArgs p) {
    constexpr bool SINGLE = FLAGS & kSingle, NESTED = FLAGS & kNested, NTL = FLAGS & kNT, CODEPTR = FLAGS & kCodePtr;
    constexpr bool XLDS = FLAGS & kXLds;
    constexpr int WAVES = (FLAGS & kWaves16) ? 16 : (FLAGS & kWaves8) ? 8 : 4;
    constexpr int THREADS = WAVES * 64;
    constexpr int TPE = THREADS / 256; // threads cooperating on one table entry
    constexpr int ABL = FLAGS >> 8;

    __shared__ __attribute__((aligned(16))) uint32_t lut[256 * 32];
    extern __shared__ __attribute__((aligned(16))) unsigned char xs[]; // XLDS: MB * ceil(K/2048) * 4 KiB activation image
    // segments of 2048 k in the activation image: a compile-time constant when the whole K fits one iteration
    const int nseg = SINGLE ? SEGS : (p.K + kSegK - 1) / kSegK;
    __shared__ float code2[NESTED ? 256 : 1];

    const int tid = threadIdx.x;
    // 0) activations: ONE copy per workgroup, by LDS-DMA, issued before anything else. Without it every
    // wavefront pulls the whole activation row through L1 itself (8 KiB per wavefront at K = 4096: twice
    // the bytes of the two weight rows it owns). The image is lane-linear per 1-KiB piece (hardware), so
    // the bank swizzle is applied on the source side: slot s of a row holds the 16-byte chunk
    // s ^ ((s >> 4) & 3) - a permutation inside 64-byte groups, so the copy stays fully coalesced - and
    // lane l finds its q-th chunk (4 l + q) at slot 4 l + (q ^ ((l >> 2) & 3)): conflict-free ds_read_b128.
    // Being the oldest vector-memory ops of the wavefront, the DMAs have landed whenever any later load
    // has (vmcnt retires in order); the explicit counted wait before the barrier below spells that out.
    if constexpr (XLDS) {
        const int pieces = MB * nseg * 4; // 1-KiB pieces
        const int w0 = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int ln0 = tid & 63;
        for (int piece = w0; piece < pieces; piece += WAVES) {
            const int m = (MB == 1) ? 0 : piece / (nseg * 4);
            const int sr = (piece - m * nseg * 4) * 64 + ln0; // slot within the row
            const int k = (sr ^ ((sr >> 4) & 3)) * 8;
            const int mr = (blockIdx.y * MB + m < p.M) ? blockIdx.y * MB + m : p.M - 1;
            const T* src = static_cast<const T*>(p.A) + static_cast<long>(mr) * p.K + ((k < p.K) ? k : 0);
            // Spelled in asm on purpose: with the builtin the compiler sees LDS-DMA and ordinary loads
            // pending on the same counter, treats vmcnt as out-of-order and turns every later wait for a
            // loaded register into vmcnt(0) - draining the weight stream before the table build. Being
            // the oldest vector-memory ops, untracked DMAs leave all of its counted waits valid.
            const uint32_t dst = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(
                                     (__attribute__((address_space(3))) void*)xs)) + piece * 1024;
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                         :
                         : "v"(src), "s"(dst)
                         : "memory", "m0");
        }
    }
    // The two code values this lane needs for its table entry are the next vector loads of the
    // kernel: vmcnt retires in order, so waiting for them later never waits for the weight stream.
    // (caller-supplied table pointer only; the built-in tables travel by value in the kernel arguments)
    float code_hi = 0.f, code_lo = 0.f;
    const int entry = tid / TPE; // table entry this lane (co-)writes
    if constexpr (CODEPTR) {
        const gfloat_ptr tbl = (gfloat_ptr)(p.code16 ? p.code16 : (p.quant_type == kNF4 ? kNF4Code : kFP4Code));
        code_hi = tbl[entry >> 4];
        code_lo = tbl[entry & 15];
    }

    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int N = p.N, K = p.K;
    const int row0 = (blockIdx.x * WAVES + wave) * RPW;
    const int m0 = blockIdx.y * MB;

    const T* __restrict__ A = static_cast<const T*>(p.A);
    const uint8_t* __restrict__ B = p.B;
    const float* __restrict__ absmax = p.absmax;

    if constexpr (ABL == 5) {
        if (lane == 0 && row0 < N)
            static_cast<T*>(p.out)[row0] = static_cast<T>(code_hi);
        return;
    }

    struct Stage {
        u32x4 w[SEGS][RPW];
        float s[SEGS][RPW];
        u32x4 x[XLDS ? 1 : SEGS][XLDS ? 1 : MB][4];
    };

    auto load_stage = [&](Stage& st, int it) {
#pragma unroll
        for (int sg = 0; sg < SEGS; ++sg) {
            const int k0 = (it * SEGS + sg) * kSegK + lane * 32;
            const bool act = k0 < K;
            const int kk = act ? k0 : 0;
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int row = (row0 + r < N) ? row0 + r : N - 1;
                const long e = static_cast<long>(row) * K + kk;
                // lanes past K read the row start instead (valid memory); their scale is forced to 0 below
                if constexpr (ABL == 3)
                    st.w[sg][r] = u32x4{static_cast<uint32_t>(lane), 0x12345678u, static_cast<uint32_t>(row), 0x9abcdef0u};
                else
                    st.w[sg][r] = ld_stream(reinterpret_cast<const u32x4*>(B + (e >> 1)), NTL);
                const long blk = e >> p.bs_shift;
                if constexpr (ABL == 4) {
                    st.s[sg][r] = 1.0f;
                } else if constexpr (NESTED) {
                    // scale reconstructed in compute_stage (needs the LDS code table)
                    st.s[sg][r] = __builtin_bit_cast(float, static_cast<uint32_t>(ld_stream(p.absmax8 + blk, NTL)));
                } else {
                    st.s[sg][r] = ld_stream(absmax + blk, NTL);
                }
            }
            if constexpr (!XLDS && ABL != 4) {
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    const int mr = (m0 + m < p.M) ? m0 + m : p.M - 1;
                    const T* ap = A + static_cast<long>(mr) * K + kk;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        st.x[sg][m][q] = *reinterpret_cast<const u32x4*>(ap + q * 8);
                }
            }
        }
    };

    float acc[MB][RPW];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int r = 0; r < RPW; ++r)
            acc[m][r] = 0.0f;

    const uint32_t lane_slot = static_cast<uint32_t>(lane & 31);
    float offset = 0.0f;
    int zsh = 0; // opaque zero, set after the barrier (see opaque_zero())

    // x fragment (4 x 16 B = this lane's 32 activations of segment sg, row m)
    auto x_frag = [&](const Stage& st, int it, int sg, int m, int q) -> u32x4 {
        if constexpr (ABL == 4) {
            return u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
        } else if constexpr (XLDS) {
            return *reinterpret_cast<const u32x4*>(xs + ((m * nseg + it * SEGS + sg) * 256 + lane * 4 + (q ^ ((lane >> 2) & 3))) * 16);
        } else {
            return st.x[sg][m][q];
        }
    };

    auto compute_stage = [&](const Stage& st, int it) {
#pragma unroll
        for (int sg = 0; sg < SEGS; ++sg) {
            u32x4 xf[MB][4];
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    xf[m][q] = x_frag(st, it, sg, m, q);
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                if constexpr (ABL == 1 || ABL == 4) {
                    const u32x4 w = st.w[sg][r];
                    const uint32_t x0 = xf[0][0][0] ^ xf[0][1][1] ^ xf[0][2][2] ^ xf[0][3][3];
                    acc[0][r] += __builtin_bit_cast(float, ((w[0] ^ w[1] ^ w[2] ^ w[3] ^ x0) & 0x007fffffu) | 0x3f800000u) *
                                 st.s[sg][r];
                    continue;
                }
                // two independent fp32 chains per output so consecutive v_dot2c do not serialise
                float part[MB][2];
#pragma unroll
                for (int m = 0; m < MB; ++m)
                    part[m][0] = part[m][1] = 0.0f;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const uint32_t w = st.w[sg][r][d];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t byte = (w >> (8 * j + zsh)) & 0xFFu;
                        const uint32_t pr = lut[(byte << 5) + lane_slot];
#pragma unroll
                        for (int m = 0; m < MB; ++m)
                            part[m][j & 1] = Pair2<T>::dot2(pr, xf[m][d][j], part[m][j & 1]);
                    }
                }
                const int k0 = (it * SEGS + sg) * kSegK + lane * 32;
                float scale;
                if constexpr (NESTED) {
                    const int kk = (k0 < K) ? k0 : 0;
                    const int row = (row0 + r < N) ? row0 + r : N - 1;
                    const long blk = (static_cast<long>(row) * K + kk) >> p.bs_shift;
                    const uint32_t q8 = __builtin_bit_cast(uint32_t, st.s[sg][r]);
                    scale = __fadd_rn(__fmul_rn(code2[q8], absmax[blk >> 8]), offset);
                } else {
                    scale = st.s[sg][r];
                }
                scale = (k0 < K) ? scale : 0.0f; // lanes past the end of the row contribute nothing
#pragma unroll
                for (int m = 0; m < MB; ++m)
                    acc[m][r] = fmaf(scale, part[m][0] + part[m][1], acc[m][r]);
            }
        }
    };

    const int iters = (K + SEGS * kSegK - 1) / (SEGS * kSegK);

    // 1) put the first stage's loads in flight
    Stage cur;
    load_stage(cur, 0);

    // 2) build the byte -> (code[hi], code[lo]) table while the weights fly.
    if constexpr (ABL == 0 || ABL == 3) {
        if constexpr (!CODEPTR) { // SGPR values picked by a v_cndmask chain, no memory
            const int hi = entry >> 4, lo = entry & 15;
            code_hi = p.code[0];
            code_lo = p.code[0];
#pragma unroll
            for (int j = 1; j < 16; ++j) {
                code_hi = (hi == j) ? p.code[j] : code_hi;
                code_lo = (lo == j) ? p.code[j] : code_lo;
            }
        }
        const uint32_t pr = Pair2<T>::pack(code_hi, code_lo);
        const u32x4 v = {pr, pr, pr, pr};
        // entry `entry` = 8 x 16 B; TPE lanes share it
        u32x4* dst = reinterpret_cast<u32x4*>(&lut[entry * 32]) + (tid % TPE) * (8 / TPE);
#pragma unroll
        for (int j = 0; j < 8 / TPE; ++j)
            dst[j] = v;
        if constexpr (NESTED) {
            if (tid < 256)
                code2[tid] = p.absmax_code[tid];
            offset = p.absmax_offset[0];
        }
    }
    if constexpr (ABL == 1 || ABL == 2 || ABL == 4)
        asm volatile("" ::"v"(code_hi), "v"(code_lo));
    if constexpr (XLDS) // the activation DMAs are older than the stage-0 loads (SEGS*RPW weights + as many scales)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SEGS * RPW * 2) : "memory");
    __syncthreads();
    zsh = opaque_zero();

    // 3) main loop: prefetch iteration it+1, consume iteration it
    if constexpr (SINGLE) {
        compute_stage(cur, 0);
    } else {
        for (int it = 0; it < iters; ++it) {
            Stage nxt;
            if (it + 1 < iters)
                load_stage(nxt, it + 1);
            compute_stage(cur, it);
            if (it + 1 < iters)
                cur = nxt;
        }
    }

    // 4) wavefront reduction + epilogue (bias add in fp32, one rounding to T)
    T* __restrict__ out = static_cast<T*>(p.out);
    const T* __restrict__ bias = static_cast<const T*>(p.bias);
#pragma unroll
    for (int m = 0; m < MB; ++m) {
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const float v = wave_sum(acc[m][r]);
            const int row = row0 + r;
            if (lane == 0 && row < N && m0 + m < p.M) {
                const float b = bias ? static_cast<float>(bias[row]) : 0.0f;
                out[static_cast<long>(m0 + m) * N + row] = static_cast<T>(v + b);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// gemv4_dotx_kernel — the production dot kernel. Same decode scheme as gemv4_dot_kernel above
// (kept as the fallback for activations too large for LDS), restructured after on-device ablation
// (profiles/): at M = 1, N = K = 4096 the weight stream alone costs ~2.2 us on top of a 1.8 us
// launch-to-launch floor, and what was slowing the full kernel down was everything *around* it:
//   * the 16 code values now arrive by value in the kernel arguments (SGPRs) and the table entry of
//     a lane is picked with a v_cndmask chain - the table build needs no memory round trip and
//     starts at cycle 0 (it used to wait ~1 us for a 64-byte gather);
//   * activations are staged ONCE per workgroup into LDS (fully coalesced, issued BEFORE the weight
//     loads so that their vmcnt wait does not drain the weight stream) instead of 8 KiB of
//     per-wavefront global loads - that was 2x the weight bytes in L1/TA traffic; the LDS image is
//     laid out so the per-lane 16-byte fragment reads are conflict-free (chunk lane*4+q at slot
//     q*64+lane of each 2048-k segment);
//   * up to 8 activation rows share one pass over the weights (the decode is shared, only the
//     v_dot2c count grows), which covers M <= 8 without the matrix pipe.
// XCH = 16-byte activation chunks staged per lane (compile time so they can sit in registers while
// the weight loads are issued).
// ---------------------------------------------------------------------------------------------
enum DotxFlags : int { kxNested = 1, kxCodePtr = 2, kxDebug = 4 };

#define BNB_STAMP(i)                                                                               \
    if constexpr (DBG) {                                                                           \
        if (lane == 0)                                                                             \
            p.dbg[(static_cast<long>(blockIdx.x) * WAVES + wave) * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
    }

template <typename T, int MB, int RPW, int XCH, int FLAGS>
__global__ __launch_bounds__(256) void gemv4_dotx_kernel(const GemvArgs p) {
    constexpr bool NESTED = FLAGS & kxNested, CODEPTR = FLAGS & kxCodePtr, DBG = FLAGS & kxDebug;
    constexpr int THREADS = 256, WAVES = 4, SEGS = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* lut = reinterpret_cast<uint32_t*>(smem); // 32 KiB pair table
    unsigned char* xs = smem + 256 * 32 * 4;           // MB * nseg * 4 KiB activation image
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int N = p.N, K = p.K;
    const int nseg = (K + kSegK - 1) / kSegK;
    float* code2 = reinterpret_cast<float*>(xs + MB * nseg * 4096);
    const int row0 = (blockIdx.x * WAVES + wave) * RPW;
    const int m0 = blockIdx.y * MB;

    const T* __restrict__ A = static_cast<const T*>(p.A);
    const uint8_t* __restrict__ B = p.B;
    const float* __restrict__ absmax = p.absmax;

    BNB_STAMP(0)
    // ---- 1) activations -> registers (first vector loads of the kernel)
    float code_hi = 0.f, code_lo = 0.f;
    if constexpr (CODEPTR) {
        const gfloat_ptr tbl = (gfloat_ptr)p.code16;
        code_hi = tbl[tid >> 4];
        code_lo = tbl[tid & 15];
    }
    const int total_chunks = MB * nseg * 256;
    u32x4 xr[XCH];
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
        const int c = tid + i * THREADS;
        const int m = c / (nseg * 256), rem = c - m * (nseg * 256);
        const int k = rem * 8; // chunk rem of the row covers k .. k+8
        const int mr = (m0 + m < p.M) ? m0 + m : p.M - 1;
        xr[i] = u32x4{0, 0, 0, 0};
        if (c < total_chunks && k < K)
            xr[i] = *reinterpret_cast<const u32x4*>(A + static_cast<long>(mr) * K + k);
    }

    // ---- 2) weights + scales of the first iteration
    struct Stage {
        u32x4 w[SEGS][RPW];
        float s[SEGS][RPW];
    };
    auto load_stage = [&](Stage& st, int it) {
#pragma unroll
        for (int sg = 0; sg < SEGS; ++sg) {
            const int k0 = (it * SEGS + sg) * kSegK + lane * 32;
            const int kk = (k0 < K) ? k0 : 0;
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int row = (row0 + r < N) ? row0 + r : N - 1;
                const long e = static_cast<long>(row) * K + kk;
                st.w[sg][r] = *reinterpret_cast<const u32x4*>(B + (e >> 1));
                const long blk = e >> p.bs_shift;
                if constexpr (NESTED)
                    st.s[sg][r] = __builtin_bit_cast(float, static_cast<uint32_t>(p.absmax8[blk]));
                else
                    st.s[sg][r] = absmax[blk];
            }
        }
    };
    Stage cur;
    load_stage(cur, 0);

    // ---- 3) table build (no memory traffic unless the caller handed a code pointer)
    {
        if constexpr (!CODEPTR) {
            const int hi = tid >> 4, lo = tid & 15;
            code_hi = p.code[0];
            code_lo = p.code[0];
#pragma unroll
            for (int j = 1; j < 16; ++j) {
                code_hi = (hi == j) ? p.code[j] : code_hi;
                code_lo = (lo == j) ? p.code[j] : code_lo;
            }
        }
        const uint32_t pr = Pair2<T>::pack(code_hi, code_lo);
        const u32x4 v = {pr, pr, pr, pr};
        u32x4* dst = reinterpret_cast<u32x4*>(&lut[tid * 32]);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            dst[j] = v;
    }
    float offset = 0.0f;
    if constexpr (NESTED) {
        code2[tid] = p.absmax_code[tid];
        offset = p.absmax_offset[0];
    }

    // ---- 4) activation image: chunk g = lane'*4 + q of segment sg -> slot q*64 + lane'
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
        const int c = tid + i * THREADS;
        if (c < total_chunks) {
            const int m = c / (nseg * 256), rem = c - m * (nseg * 256);
            const int sg = rem >> 8, g = rem & 255;
            *reinterpret_cast<u32x4*>(xs + ((m * nseg + sg) * 256 + (g & 3) * 64 + (g >> 2)) * 16) = xr[i];
        }
    }
    BNB_STAMP(1)
    __syncthreads();
    const int zsh = opaque_zero();
    BNB_STAMP(2)
    if constexpr (DBG) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        BNB_STAMP(3)
    }

    float acc[MB][RPW];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int r = 0; r < RPW; ++r)
            acc[m][r] = 0.0f;
    const uint32_t lane_slot = static_cast<uint32_t>(lane & 31);

    auto compute_stage = [&](const Stage& st, int it) {
#pragma unroll
        for (int sg = 0; sg < SEGS; ++sg) {
            const int seg = it * SEGS + sg;
            if (seg >= nseg)
                break;
            float part[RPW][MB][2];
#pragma unroll
            for (int r = 0; r < RPW; ++r)
#pragma unroll
                for (int m = 0; m < MB; ++m)
                    part[r][m][0] = part[r][m][1] = 0.0f;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                u32x4 xf[MB];
#pragma unroll
                for (int m = 0; m < MB; ++m)
                    xf[m] = *reinterpret_cast<const u32x4*>(xs + ((m * nseg + seg) * 256 + d * 64 + lane) * 16);
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    const uint32_t w = st.w[sg][r][d];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t byte = (w >> (8 * j + zsh)) & 0xFFu;
                        const uint32_t pr = lut[(byte << 5) + lane_slot];
#pragma unroll
                        for (int m = 0; m < MB; ++m)
                            part[r][m][j & 1] = Pair2<T>::dot2(pr, xf[m][j], part[r][m][j & 1]);
                    }
                }
            }
            const int k0 = seg * kSegK + lane * 32;
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                float scale;
                if constexpr (NESTED) {
                    const int kk = (k0 < K) ? k0 : 0;
                    const int row = (row0 + r < N) ? row0 + r : N - 1;
                    const long blk = (static_cast<long>(row) * K + kk) >> p.bs_shift;
                    const uint32_t q8 = __builtin_bit_cast(uint32_t, st.s[sg][r]);
                    scale = __fadd_rn(__fmul_rn(code2[q8], absmax[blk >> 8]), offset);
                } else {
                    scale = st.s[sg][r];
                }
                scale = (k0 < K) ? scale : 0.0f; // lanes past the end of the row contribute nothing
#pragma unroll
                for (int m = 0; m < MB; ++m)
                    acc[m][r] = fmaf(scale, part[r][m][0] + part[r][m][1], acc[m][r]);
            }
        }
    };

    const int iters = (nseg + SEGS - 1) / SEGS;
    for (int it = 0; it < iters; ++it) {
        Stage nxt;
        const bool more = it + 1 < iters;
        if (more)
            load_stage(nxt, it + 1);
        compute_stage(cur, it);
        if (more)
            cur = nxt;
    }

    if constexpr (DBG) {
        asm volatile("" ::"v"(acc[0][0]), "v"(acc[MB - 1][RPW - 1]));
        BNB_STAMP(4)
    }
    T* __restrict__ out = static_cast<T*>(p.out);
    const T* __restrict__ bias = static_cast<const T*>(p.bias);
#pragma unroll
    for (int m = 0; m < MB; ++m) {
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const float v = wave_sum(acc[m][r]);
            const int row = row0 + r;
            if (lane == 0 && row < N && m0 + m < p.M) {
                const float b = bias ? static_cast<float>(bias[row]) : 0.0f;
                out[static_cast<long>(m0 + m) * N + row] = static_cast<T>(v + b);
            }
        }
    }
    BNB_STAMP(5)
}
#undef BNB_STAMP

// ---------------------------------------------------------------------------------------------
// Generic fallback: any T (incl. fp32 activations), any K (odd, not a multiple of 32), any pointer
// alignment. One wavefront per output row, scalar byte loads, fp32 math, 16-entry fp32 table in
// LDS. Correctness path only; the dispatcher prefers the kernel above whenever its preconditions
// hold (T in {bf16, f16}, K % 32 == 0, 16-byte aligned A and B).
// ---------------------------------------------------------------------------------------------
template <typename T, bool NESTED> __global__ __launch_bounds__(256) void gemv4_generic_kernel(const GemvArgs p) {
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
    const int bs_mask = (1 << p.bs_shift) - 1;
    float acc = 0.0f;
    float run = 0.0f;       // partial sum inside the current quantization block
    long run_blk = -1;
    auto block_scale = [&](long blk) -> float {
        if constexpr (NESTED)
            return __fadd_rn(__fmul_rn(code2[p.absmax8[blk]], p.absmax[blk >> 8]), p.absmax_offset[0]);
        else
            return p.absmax[blk];
    };
    (void)bs_mask;
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

// runtime knob bits -> the FLAGS template argument (only a curated set of combinations is instantiated)
template <typename T, int MB, int RPW, int SEGS, int EXTRA> void launch_dot(const GemvArgs& p, hipStream_t stream) {
    constexpr int waves = (EXTRA & kWaves16) ? 16 : (EXTRA & kWaves8) ? 8 : 4;
    const int rows_per_block = waves * RPW;
    dim3 grid((p.N + rows_per_block - 1) / rows_per_block, (p.M + MB - 1) / MB);
    dim3 block(waves * 64);
    const bool single = p.K <= SEGS * kSegK;
    // The table always comes through the pointer path (built-in device table unless the caller
    // supplied one): measured faster than passing the 16 values by value and selecting them with a
    // v_cndmask chain (profiles/: 4.9-5.3 us vs 5.4-6.2 us per launch at M = 1, N = K = 4096).
    constexpr int E = (EXTRA & (kWaves8 | kWaves16)) | kCodePtr;
    // activations through LDS (one DMA copy per workgroup) whenever the image fits; debug flag 128 switches
    // it off for A/B measurements
    const int nseg = single ? SEGS : (p.K + kSegK - 1) / kSegK;
    const size_t xbytes = static_cast<size_t>(MB) * nseg * 4096;
    const bool xlds = xbytes <= 96 * 1024 && !(g_dot_flags & 128);
#define BNB_DOT_GO(F)                                                                              \
    do {                                                                                           \
        auto kern = gemv4_dot_kernel<T, MB, RPW, SEGS, (F)>;                                       \
        const size_t dyn = ((F) & kXLds) ? xbytes : 0;                                             \
        static bool attr_done = false;                                                             \
        if (dyn + 34 * 1024 > 64 * 1024 && !attr_done) {                                           \
            BNB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                 \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)); \
            attr_done = true;                                                                      \
        }                                                                                          \
        hipLaunchKernelGGL(kern, grid, block, dyn, stream, p);                                     \
    } while (0)
    const int sel = (single ? 1 : 0) | (p.absmax8 ? 2 : 0) | (xlds ? 4 : 0);
    switch (sel) {
    case 0: BNB_DOT_GO(E); break;
    case 1: BNB_DOT_GO(E | kSingle); break;
    case 2: BNB_DOT_GO(E | kNested); break;
    case 3: BNB_DOT_GO(E | kSingle | kNested); break;
    case 4: BNB_DOT_GO(E | kXLds); break;
    case 5: BNB_DOT_GO(E | kSingle | kXLds); break;
    case 6: BNB_DOT_GO(E | kNested | kXLds); break;
    default: BNB_DOT_GO(E | kSingle | kNested | kXLds); break;
    }
#undef BNB_DOT_GO
}

template <typename T> void launch_generic(const GemvArgs& p, hipStream_t stream) {
    dim3 grid((p.N + 3) / 4, p.M);
    if (p.absmax8)
        hipLaunchKernelGGL((gemv4_generic_kernel<T, true>), grid, dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL((gemv4_generic_kernel<T, false>), grid, dim3(256), 0, stream, p);
}

template <typename T, int MB, int XCH> bool launch_dotx(const GemvArgs& p, hipStream_t stream) {
    constexpr int RPW = 2;
    const int nseg = (p.K + kSegK - 1) / kSegK;
    const size_t smem = 256 * 32 * 4 + static_cast<size_t>(MB) * nseg * 4096 + 1024;
    dim3 grid((p.N + 4 * RPW - 1) / (4 * RPW), (p.M + MB - 1) / MB);
    const int flags = (p.absmax8 ? kxNested : 0) | (p.code16 ? kxCodePtr : 0);
    if constexpr ((MB == 1 && XCH == 2) || (MB == 8 && XCH == 16)) {
        if (g_dbg_buf && flags == 0) {
            GemvArgs q = p;
            q.dbg = g_dbg_buf;
            auto kern = gemv4_dotx_kernel<T, MB, RPW, XCH, kxDebug>;
            static bool attr_dbg = false;
            if (!attr_dbg) {
                BNB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                attr_dbg = true;
            }
            hipLaunchKernelGGL(kern, grid, dim3(256), smem, stream, q);
            return true;
        }
    }
#define BNB_DOTX_LAUNCH(F)                                                                         \
    if (flags == (F)) {                                                                            \
        auto kern = gemv4_dotx_kernel<T, MB, RPW, XCH, (F)>;                                       \
        static bool attr_done = false;                                                             \
        if (!attr_done) {                                                                          \
            BNB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                 \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            attr_done = true;                                                                      \
        }                                                                                          \
        hipLaunchKernelGGL(kern, grid, dim3(256), smem, stream, p);                                \
        return true;                                                                               \
    }
    BNB_DOTX_LAUNCH(0) BNB_DOTX_LAUNCH(kxNested) BNB_DOTX_LAUNCH(kxCodePtr)
#undef BNB_DOTX_LAUNCH
    return false; // nested + caller code pointer never occurs (the gemv op un-nests on the host)
}

// x-in-LDS dot kernel: MB rows per pass (1, 2, 4, 8), XCH = MB * nseg chunks per lane rounded up to a
// power of two <= 16. Returns false when the activations do not fit (caller falls back).
template <typename T> bool dispatch_dotx(const GemvArgs& p, hipStream_t stream) {
    const int nseg = (p.K + kSegK - 1) / kSegK;
    const int mb = p.M >= 5 ? 8 : p.M >= 3 ? 4 : p.M;
    int need = mb * nseg;
    int mbsel = mb;
    while (need > 16 && mbsel > 1) { // too much activation per pass: fewer rows per pass, more passes
        mbsel >>= 1;
        need = mbsel * nseg;
    }
    if (need > 16)
        return false;
    const int xch = need <= 2 ? 2 : need <= 4 ? 4 : need <= 8 ? 8 : 16;
#define BNB_DOTX_CASE(MBV, XV)                                                                     \
    if (mbsel == MBV && xch == XV)                                                                 \
        return launch_dotx<T, MBV, XV>(p, stream);
    BNB_DOTX_CASE(1, 2) BNB_DOTX_CASE(1, 4) BNB_DOTX_CASE(1, 8) BNB_DOTX_CASE(1, 16)
    BNB_DOTX_CASE(2, 2) BNB_DOTX_CASE(2, 4) BNB_DOTX_CASE(2, 8) BNB_DOTX_CASE(2, 16)
    BNB_DOTX_CASE(4, 4) BNB_DOTX_CASE(4, 8) BNB_DOTX_CASE(4, 16)
    BNB_DOTX_CASE(8, 8) BNB_DOTX_CASE(8, 16)
#undef BNB_DOTX_CASE
    return false;
}

template <typename T> void dispatch_dot(const GemvArgs& p, hipStream_t stream) {
    if (g_dot_ablate == 0 && (g_dot_flags & 32) && dispatch_dotx<T>(p, stream))
        return;
    // profiling-only ablations of the M = 1, K <= 4096 configuration
    if (g_dot_ablate != 0 && p.M == 1 && !p.absmax8 && p.K <= 2 * kSegK) {
        dim3 grid((p.N + 7) / 8, 1);
#define BNB_ABL(A)                                                                                 \
    if (g_dot_ablate == A) {                                                                       \
        hipLaunchKernelGGL((gemv4_dot_kernel<T, 1, 2, 2, kSingle | (A << 8)>), grid, dim3(256), 0, stream, p); \
        return;                                                                                    \
    }
        BNB_ABL(1) BNB_ABL(2) BNB_ABL(3) BNB_ABL(4) BNB_ABL(5)
#undef BNB_ABL
    }

    // Calibrated on MI355X (profiles/): 512-thread workgroups (one table build per 8 wavefronts),
    // 2 weight rows per wavefront once that still yields >= 256 workgroups, else 1.
    // Calibrated on MI355X (profiles/r1_dot_ab.txt): with the activations in LDS one row per wavefront wins at
    // M = 1 for every shape tried (4096^2: 4.73 vs 5.35 us; twice the wavefronts to hide latency and nothing
    // left to amortise); two activation rows keep two weight rows per wavefront on the large matrices only
    // (4096^2 M = 2: 5.47 vs 5.93 us; 8192^2: 19.1 vs 17.0; 4096 x 11008: 13.6 vs 11.4).
    int rpw = g_dot_rpw;
    if (rpw == 0)
        rpw = (p.M >= 2 && static_cast<long>(p.N) * p.K > (24L << 20)) ? 2 : 1;
    int segs = g_dot_segs;
    if (segs == 0)
        segs = (p.K > kSegK) ? 2 : 1;
    const int mb = (p.M >= 3) ? 4 : p.M;
    int extra = g_dot_flags & (kWaves8 | kWaves16);
    if (g_dot_rpw == 0 && (g_dot_flags & ~128) == 0)
        extra = (mb <= 2) ? kWaves8 : 0;

#define BNB_DOT_CASE(MBV, RPWV, SEGSV, EX)                                                         \
    if (mb == MBV && rpw == RPWV && segs == SEGSV && extra == (EX)) {                              \
        launch_dot<T, MBV, RPWV, SEGSV, (EX)>(p, stream);                                          \
        return;                                                                                    \
    }
#define BNB_DOT_ALLX(MBV, RPWV, SEGSV)                                                             \
    BNB_DOT_CASE(MBV, RPWV, SEGSV, 0) BNB_DOT_CASE(MBV, RPWV, SEGSV, kWaves8) BNB_DOT_CASE(MBV, RPWV, SEGSV, kWaves16)
    BNB_DOT_ALLX(1, 2, 2) BNB_DOT_ALLX(1, 1, 2) BNB_DOT_ALLX(1, 4, 2) BNB_DOT_ALLX(2, 2, 2) BNB_DOT_ALLX(2, 1, 2)
    BNB_DOT_ALLX(4, 1, 2)
    BNB_DOT_CASE(1, 1, 1, 0) BNB_DOT_CASE(1, 2, 1, 0) BNB_DOT_CASE(1, 4, 1, 0) BNB_DOT_CASE(1, 8, 1, 0)
    BNB_DOT_CASE(2, 1, 1, 0) BNB_DOT_CASE(2, 1, 2, 0) BNB_DOT_CASE(2, 2, 1, 0) BNB_DOT_CASE(2, 4, 1, 0)
    BNB_DOT_CASE(4, 1, 1, 0) BNB_DOT_CASE(4, 2, 1, 0)
#undef BNB_DOT_ALLX
#undef BNB_DOT_CASE
    // unsupported combination requested by a sweep: fall back to a safe one
    launch_dot<T, 1, 1, 1, 0>(p, stream);
}

} // namespace

// Entry used by c_api.hip. dtype: 0 = f32, 1 = f16, 2 = bf16. Chooses the fast kernel when its
// preconditions hold, else the generic one. M may be any value >= 1 (rows handled 4 at a time).
void gemv_4bit_dot(int dtype, const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax8,
                   const float* absmax_code, const float* absmax_offset, const float* code16, void* out,
                   const void* bias, int M, int N, int K, int blocksize, int quant_type, hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0)
        return;
    GemvArgs p;
    p.A = A;
    p.B = B;
    p.absmax = absmax;
    p.absmax8 = absmax8;
    p.absmax_code = absmax_code;
    p.absmax_offset = absmax_offset;
    p.code16 = code16;
    p.out = out;
    p.bias = bias;
    p.M = M;
    p.N = N;
    p.K = K;
    p.bs_shift = ilog2(blocksize);
    p.quant_type = quant_type;
    p.dbg = nullptr;
    {
        static const float nf4[16] = {BNB_NF4_VALUES};
        static const float fp4[16] = {BNB_FP4_VALUES};
        for (int i = 0; i < 16; ++i)
            p.code[i] = (quant_type == kNF4) ? nf4[i] : fp4[i];
    }

    const bool fast_ok = (dtype != 0) && (K % 32 == 0) && (blocksize >= 32) && is_pow2(blocksize) &&
                         aligned_to(A, 16) && aligned_to(B, 16);
    if (fast_ok) {
        if (dtype == 2)
            dispatch_dot<bf16>(p, stream);
        else
            dispatch_dot<f16>(p, stream);
    } else {
        if (dtype == 0)
            launch_generic<float>(p, stream);
        else if (dtype == 1)
            launch_generic<f16>(p, stream);
        else
            launch_generic<bf16>(p, stream);
    }
    BNB_CHECK_LAUNCH();
}

} // namespace bnb
