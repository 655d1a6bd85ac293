// gemv4.hip — fused 4-bit dequantize + dot-product kernel for decode-sized batches (1, 2 or 4 rows of A
// per pass) on gfx950.   out[m, n] = sum_k A[m, k] * code[B[n, k]] * scale[n, k / bs]  (+ bias[n])
//
// Replaces, on MI355X, the reference's kgemm_4bit_inference_naive (csrc/kernels.cu:1452-1567) and the
// small-M range of gemm_4bit_simt (csrc/gemm_4bit_simt.cu:109-480). It is not a translation of
// either: those map one logical 32-lane warp to an output column and decode nibbles with shifts and
// an LDS/const table per nibble; this kernel is built around four CDNA4 facts:
//
//  * HBM-bound, so the first job is to keep the weight bytes in flight. A wavefront owns RPW whole
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
//    160 KiB LDS buys a 2-cycles-per-64-bytes decode. The table is built while the first weight
//    loads are in flight.
//  * The activations are the same for every row: one LDS-DMA copy per workgroup (issued first, source-side
//    bank swizzle) instead of every wavefront pulling them through L1 again - see step 0 of the kernel.
//
// The per-block scale multiplies the fp32 partial sum of each 32-nibble run (one run never
// straddles a quantization block because blocksize >= 32 and K % 32 == 0), so absmax is applied in
// full fp32. Nested (double-quantized) absmax is fused:
//   scale = absmax_code[absmax_8bit[b]] * absmax[b >> 8] + offset   (reference autograd/_functions.py:471-485).
#include "bnb_common.h"

namespace bnb {

int g_dot_ablate = 0; // profiling only: see the ablation bits of DotFlags
int g_dot_flags = 0;  // sweeps (0 = default): 16 = EXPERIMENTAL diagonal-MFMA decode, 32 = force the 32-copy table, 64 = 256-thread workgroups, 128 = activations
                      // per wavefront instead of LDS (M = 1 only), bits 8..15 = KiB of LDS padding (occupancy experiments)
unsigned long long* g_dbg_buf = nullptr; // profiling only: device buffer for s_memtime stamps

// Tuning knobs (overridable for sweeps through bnb_mi355x_set_tuning; see c_api.hip).
int g_dot_rpw = 0;  // rows per wavefront, 0 = heuristic
int g_dot_segs = 0; // 2048-k segments per iteration, 0 = heuristic

namespace {

using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

using f32x4 = __attribute__((ext_vector_type(4))) float;

template <typename T> struct Pair2;
template <> struct Pair2<bf16> {
    // one 16x16x32 MFMA: a, b = this lane's 8 bf16 of the A / B operand (4 packed dwords each)
    static __device__ __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
        using V = __attribute__((ext_vector_type(8))) bf16;
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(V, a), __builtin_bit_cast(V, b), c, 0, 0, 0);
    }
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
    static __device__ __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
        using V = __attribute__((ext_vector_type(8))) f16;
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(V, a), __builtin_bit_cast(V, b), c, 0, 0, 0);
    }
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
    unsigned long long* dbg;    // profiling builds only: per-wavefront s_memtime stamps (8 per wave), else NULL
};

constexpr int kSegK = 2048; // k covered by one wavefront-wide 16-byte load

// Compile-time switches of the dot kernel, packed into one template int.
enum DotFlags : int {
    kSingle = 1,  // the whole K fits one iteration: no prefetch registers are allocated
    kNested = 2,  // double-quantised absmax reconstructed in-kernel
    kWaves8 = 4,  // 512-thread workgroups (8 wavefronts share one table build) instead of 256
    kDiag = 8,    // EXPERIMENTAL (debug flag 16; written after round 1's GPU budget was spent - not yet run on
                  // hardware): products on the matrix pipe instead of v_dot2c, see "diagonal MFMA" in the kernel
    kXLds = 16,   // activations staged once per workgroup in LDS (one LDS-DMA copy) instead of per-wave global loads
    kLut64 = 64,  // 64 table copies, 256 B per entry: the LDS address of a look-up is ONE v_perm_b32
                  // (byte 0 = the lane's offset, byte 1 = the packed weight byte) instead of shift + mask + or
    // bits 8..: ablation for profiling builds (results are wrong): 1 = stream + reduce raw words, no decode;
    // 2 = no table build; 3 = no weight loads; 4 = weights only (no x / absmax traffic); 5 = empty kernel
};

// T in {bf16, f16}; MB = activation rows per pass; RPW = weight rows per wavefront;
// SEGS = 2048-k sub-segments per loop iteration.
template <typename T, int MB, int RPW, int SEGS, int FLAGS>
__global__ __launch_bounds__((FLAGS & kWaves8) ? 512 : 256) void gemv4_dot_kernel(
    // The hot arguments are separate scalars so that the command processor can PRELOAD them into SGPRs
    // (-mllvm -amdgpu-kernarg-preload-count=16, gfx950 kernarg preload): a wavefront otherwise starts with a
    // dependent s_load from a kernarg buffer that is cold in every cache. The rest stays in the struct.
    // (13 of the 14 preloadable dwords - 16 user SGPRs minus the kernarg segment pointer: everything the loads of
    // the first stage and the table build depend on, including the nested uint8 absmax pointer - fetched from the
    // struct it would put a cold s_load in front of the weight stream; the output pointer and the nested
    // offset are needed late and stay in the struct)
    const void* hot_A, const uint8_t* hot_B, const float* hot_absmax, const uint8_t* hot_absmax8, const float* hot_code16,
    int hot_N, int hot_K, int hot_packed /* M | bs_shift << 24 | quant_type << 29 */, const GemvArgs p) {
    const int hot_M = hot_packed & 0xFFFFFF, hot_bs_shift = (hot_packed >> 24) & 31, hot_quant_type = (hot_packed >> 29) & 3;
    void* const hot_out = p.out;
    constexpr bool SINGLE = FLAGS & kSingle, NESTED = FLAGS & kNested, LUT64 = FLAGS & kLut64;
    constexpr int COPIES = LUT64 ? 64 : 32; // table copies = dwords per entry
    constexpr bool XLDS = FLAGS & kXLds;
    constexpr bool DIAG = FLAGS & kDiag;
    static_assert(!DIAG || XLDS, "the diagonal-MFMA decode reads its activation fragments from the LDS image");
    constexpr int WAVES = (FLAGS & kWaves8) ? 8 : 4;
    constexpr int THREADS = WAVES * 64;
    constexpr int TPE = THREADS / 256; // threads cooperating on one table entry
    constexpr int ABL = FLAGS >> 8;
    static_assert(kLut64 < 256, "flag bits below the ablation field");

    __shared__ __attribute__((aligned(16))) uint32_t lut[256 * COPIES];
    extern __shared__ __attribute__((aligned(16))) unsigned char xs[]; // XLDS: MB * ceil(K/2048) * 4 KiB activation image
    // segments of 2048 k in the activation image: a compile-time constant when the whole K fits one iteration
    const int nseg = SINGLE ? SEGS : (hot_K + kSegK - 1) / kSegK;
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
            const int mr = (blockIdx.y * MB + m < hot_M) ? blockIdx.y * MB + m : hot_M - 1;
            const T* src = static_cast<const T*>(hot_A) + static_cast<long>(mr) * hot_K + ((k < hot_K) ? k : 0);
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
    // (the caller's table if one was passed - the legacy gemv op does - else the device-resident built-in one;
    // passing the 16 values by value and selecting with v_cndmask measured 0.6-0.9 us slower)
    const int entry = tid / TPE; // table entry this lane (co-)writes
    const gfloat_ptr tbl = (gfloat_ptr)(hot_code16 ? hot_code16 : (hot_quant_type == kNF4 ? kNF4Code : kFP4Code));
    float code_hi = tbl[entry >> 4];
    float code_lo = tbl[entry & 15];
    float code2_v = 0.0f; // nested: this thread's entry of the 256-entry absmax code, loaded BEFORE the weight stream
    if constexpr (NESTED)
        code2_v = p.absmax_code[tid & 255];

    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int N = hot_N, K = hot_K;
    const int row0 = (blockIdx.x * WAVES + wave) * RPW;
    const int m0 = blockIdx.y * MB;

    const T* __restrict__ A = static_cast<const T*>(hot_A);
    const uint8_t* __restrict__ B = hot_B;
    const float* __restrict__ absmax = hot_absmax;

    if constexpr (ABL == 5) {
        if (lane == 0 && row0 < N)
            static_cast<T*>(hot_out)[row0] = static_cast<T>(code_hi);
        return;
    }

    struct Stage {
        u32x4 w[SEGS][RPW];
        float s[SEGS][RPW];
        float s2[NESTED ? SEGS : 1][NESTED ? RPW : 1]; // nested: second-level absmax of the block's 256-group
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
                    st.w[sg][r] = *reinterpret_cast<const u32x4*>(B + (e >> 1));
                const long blk = e >> hot_bs_shift;
                if constexpr (ABL == 4) {
                    st.s[sg][r] = 1.0f;
                } else if constexpr (NESTED) {
                    // scale reconstructed in compute_stage (needs the LDS code table)
                    st.s[sg][r] = __builtin_bit_cast(float, static_cast<uint32_t>(hot_absmax8[blk]));
                    st.s2[sg][r] = absmax[blk >> 8]; // issued with the stage: a load inside the decode would expose its latency
                } else {
                    st.s[sg][r] = absmax[blk];
                }
            }
            if constexpr (!XLDS && ABL != 4) {
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    const int mr = (m0 + m < hot_M) ? m0 + m : hot_M - 1;
                    const T* ap = A + static_cast<long>(mr) * K + kk;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        st.x[sg][m][q] = *reinterpret_cast<const u32x4*>(ap + q * 8);
                }
            }
        }
    };

    float acc[MB][RPW];
    f32x4 acc4[DIAG ? MB : 1][DIAG ? RPW : 1]; // DIAG: running 16x16 tile per output, only its diagonal is meaningful
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            acc[m][r] = 0.0f;
            if constexpr (DIAG)
                acc4[m][r] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }

    const uint32_t lane_slot = static_cast<uint32_t>(lane & 31);
    float offset = 0.0f;
    int zsh = 0; // opaque zero, set after the barrier (see opaque_zero())
    // kLut64: selector of v_perm_b32(S0 = weight dword, S1 = lane offset): result bytes {S1.b0, S0.b[j], 0, 0}
    uint32_t perm_sel = 0x0C0C0400u;
    const uint32_t lane_off64 = static_cast<uint32_t>(lane) * 4u;
    const auto lut_lds = (const __attribute__((address_space(3))) uint32_t*)lut;

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

    // byte j of weight dword w -> (code[hi nibble], code[lo nibble]) as a packed pair, from this lane's table copy
    auto lut_pair = [&](uint32_t w, int j) -> uint32_t {
        if constexpr (LUT64) {
            // address = byte * 256 + lane * 4, assembled by one byte permute
            const uint32_t addr = __builtin_amdgcn_perm(w, lane_off64, perm_sel + (j << 8));
            return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t*>(
                reinterpret_cast<const __attribute__((address_space(3))) unsigned char*>(lut_lds) + addr);
        } else {
            const uint32_t byte = (w >> (8 * j + zsh)) & 0xFFu;
            return lut[(byte << 5) + lane_slot];
        }
    };
    auto block_scale = [&](const Stage& st, int it, int sg, int r) -> float {
        const int k0 = (it * SEGS + sg) * kSegK + lane * 32;
        float scale;
        if constexpr (NESTED) {
            const uint32_t q8 = __builtin_bit_cast(uint32_t, st.s[sg][r]);
            scale = __fadd_rn(__fmul_rn(code2[q8], st.s2[sg][r]), offset);
        } else {
            scale = st.s[sg][r];
        }
        return (k0 < K) ? scale : 0.0f; // lanes past the end of the row contribute nothing
    };

    // "Diagonal MFMA" decode (kDiag). With this lane's 8 decoded weights as the A operand and the 8 matching
    // activations as the B operand, v_mfma_f32_16x16x32 computes D[i][j] = sum over the four lanes (i, g) x (j, g)
    // of 8-element dots; on the diagonal that is the sum of the dots of lanes {i, i+16, i+32, i+48} - a 64-lane
    // batched dot on the matrix pipe, every lane keeping its coalesced 16-byte ownership of the row. The four
    // lanes of a diagonal element must share one absmax block, so the raw dwords are first transposed 4x4 between
    // "dword d of the lane" and "16-lane row g" (two v_permlane32_swap + two v_permlane16_swap): afterwards
    // lane (i, g) holds, in register d', dword g of original lane L = i + 16 d', the diagonal element i of MFMA
    // d' is the complete 32-nibble run of lane L, and it is scaled by L's fp32 scale (one ds_bpermute per d')
    // before it joins the running tile. Index algebra checked in numpy (tests/checks/emulate_diag_mfma.py).
    auto compute_stage_diag = [&](const Stage& st, int it) {
        const int li = lane & 15, lg = lane >> 4;
#pragma unroll
        for (int sg = 0; sg < SEGS; ++sg) {
            uint32_t wt[RPW][4]; // transposed raw dwords: wt[r][d'] = dword lg of original lane li + 16 d'
            int scale_bits[RPW];
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const auto a = __builtin_amdgcn_permlane32_swap(st.w[sg][r][0], st.w[sg][r][2], false, false);
                const auto b = __builtin_amdgcn_permlane32_swap(st.w[sg][r][1], st.w[sg][r][3], false, false);
                const auto c = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
                const auto d = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
                wt[r][0] = c[0], wt[r][1] = c[1], wt[r][2] = d[0], wt[r][3] = d[1];
                scale_bits[r] = __builtin_bit_cast(int, block_scale(st, it, sg, r));
            }
#pragma unroll
            for (int dp = 0; dp < 4; ++dp) {
                // the 8 activations of x chunk 4 L + lg, L = li + 16 d' (same swizzled slot rule as x_frag), read
                // per d' so that only MB fragments are live at a time
                const int L = li + 16 * dp;
                u32x4 xf[MB];
#pragma unroll
                for (int m = 0; m < MB; ++m)
                    xf[m] = *reinterpret_cast<const u32x4*>(
                        xs + ((m * nseg + it * SEGS + sg) * 256 + L * 4 + (lg ^ ((L >> 2) & 3))) * 16);
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    u32x4 a_frag;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        a_frag[j] = lut_pair(wt[r][dp], j);
                    const float s_dp = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(L * 4, scale_bits[r]));
#pragma unroll
                    for (int m = 0; m < MB; ++m) {
                        const f32x4 t = Pair2<T>::mfma(a_frag, xf[m], f32x4{0.0f, 0.0f, 0.0f, 0.0f});
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc4[m][r][e] = fmaf(s_dp, t[e], acc4[m][r][e]);
                    }
                }
            }
        }
    };

    auto compute_stage = [&](const Stage& st, int it) {
        if constexpr (DIAG) {
            compute_stage_diag(st, it);
            return;
        }
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
                        const uint32_t pr = lut_pair(w, j);
#pragma unroll
                        for (int m = 0; m < MB; ++m)
                            part[m][j & 1] = Pair2<T>::dot2(pr, xf[m][d][j], part[m][j & 1]);
                    }
                }
                const float scale = block_scale(st, it, sg, r);
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
        const uint32_t pr = Pair2<T>::pack(code_hi, code_lo);
        const u32x4 v = {pr, pr, pr, pr};
        // entry `entry` = COPIES dwords; TPE lanes share it
        // All chunks of an entry hold the same value, so the ORDER in which a lane writes its chunks is free:
        // rotating it by the entry's index in the wavefront spreads the 8 lanes that a ds_write_b128 services
        // together over all 32 bank quads (written in natural order they are 4- to 8-way conflicted: their
        // addresses are 64 or 128 bytes apart).
        constexpr int NCH = COPIES / 4 / TPE;
        u32x4* dst = reinterpret_cast<u32x4*>(&lut[entry * COPIES]) + (tid % TPE) * NCH;
        const int rot = (NCH >= 8) ? (tid & 63) : (tid & 63) * NCH / 8; // distinct bank quads within 8 lanes
#pragma unroll
        for (int j = 0; j < NCH; ++j)
            dst[(j + rot) % NCH] = v;
        if constexpr (NESTED) {
            if (tid < 256)
                code2[tid] = code2_v;
            offset = p.absmax_offset[0];
        }
    }
    if constexpr (ABL == 1 || ABL == 2 || ABL == 4)
        asm volatile("" ::"v"(code_hi), "v"(code_lo));
    if constexpr (XLDS) // the activation DMAs are older than the stage-0 loads (SEGS*RPW weights + as many scales, two when nested)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SEGS * RPW * (NESTED ? 3 : 2)) : "memory");
    __syncthreads();
    zsh = opaque_zero();
    perm_sel += static_cast<uint32_t>(zsh); // same fence for the permute form of the decode

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
    T* __restrict__ out = static_cast<T*>(hot_out);
    const T* __restrict__ bias = static_cast<const T*>(p.bias);
#pragma unroll
    for (int m = 0; m < MB; ++m) {
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            if constexpr (DIAG) {
                // the tile's diagonal: element j lives in lane (j, g = j / 4), register j % 4
                const int j = lane & 15;
                const f32x4 t = acc4[m][r];
                const float d = ((j & 3) == 0) ? t[0] : ((j & 3) == 1) ? t[1] : ((j & 3) == 2) ? t[2] : t[3];
                acc[m][r] = ((lane >> 4) == (j >> 2)) ? d : 0.0f;
            }
            const float v = wave_sum(acc[m][r]);
            const int row = row0 + r;
            if (lane == 0 && row < N && m0 + m < hot_M) {
                const float b = bias ? static_cast<float>(bias[row]) : 0.0f;
                out[static_cast<long>(m0 + m) * N + row] = static_cast<T>(v + b);
            }
        }
    }
}


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

constexpr size_t kXLdsMaxBytes = 96 * 1024; // activation image cap (the table takes another 32 KiB)

// bytes of the LDS activation image for MB rows (whole 2048-k segments)
inline size_t x_image_bytes(int mb, int K, int segs) {
    const bool single = K <= segs * kSegK;
    const int nseg = single ? segs : (K + kSegK - 1) / kSegK;
    return static_cast<size_t>(mb) * nseg * 4096;
}

// One geometry (MB activation rows per pass, RPW weight rows per wavefront, SEGS segments per iteration,
// 256- or 512-thread workgroups) -> the right FLAGS instance for this problem.
template <typename T, int MB, int RPW, int SEGS, int EXTRA> void launch_dot(const GemvArgs& p, hipStream_t stream) {
    constexpr int waves = (EXTRA & kWaves8) ? 8 : 4;
    const int rows_per_block = waves * RPW;
    dim3 grid((p.N + rows_per_block - 1) / rows_per_block, (p.M + MB - 1) / MB);
    dim3 block(waves * 64);
    const bool single = p.K <= SEGS * kSegK;
    // The table always comes through the pointer path (built-in device table unless the caller
    // supplied one): measured faster than passing the 16 values by value and selecting them with a
    // v_cndmask chain (profiles/: 4.9-5.3 us vs 5.4-6.2 us per launch at M = 1, N = K = 4096).
    constexpr int E = (EXTRA & kWaves8);
    // activations through LDS (one DMA copy per workgroup) whenever the image fits (the dispatcher picks MB
    // so that it does); debug flag 128 switches it off for A/B measurements. Per-wavefront global loads of
    // the activations are only instantiated for MB = 1: with more rows the prefetch stage spills registers.
    const size_t xbytes = x_image_bytes(MB, p.K, SEGS);
    const size_t lds_pad = static_cast<size_t>((g_dot_flags >> 8) & 0xFF) * 1024; // profiling: occupancy experiments
    const bool xlds = xbytes <= kXLdsMaxBytes && (MB > 1 || !(g_dot_flags & 128));
    // 64-copy table (look-up address = one v_perm_b32) as long as two workgroups still share a CU, i.e. table +
    // nested code table + activation image <= 80 KiB (measured: 84 KiB per workgroup drops to one per CU and
    // costs 1.4 us at 4096^2); debug flag 32 forces the 32-copy table.
    // Every workgroup builds its own table, so the larger one only pays while a CU sees few workgroups
    // (profiles/r1_dot_ab.txt: 4096^2 M = 1 4.48 vs 4.58 us, nested 4.82 vs 5.09, 8192^2 11.6 vs 12.1; but
    // 11008 x 4096 M = 1 = 1376 workgroups 9.3 vs 8.1). At M = 2 it only won in the two-rows-per-wavefront
    // geometry of the large matrices (11008 x 4096: 10.9 vs 11.6; 4096^2: 5.19 vs 5.11).
    const bool few_wgs = static_cast<long>(grid.x) * grid.y <= 1024;
    const bool lut64 = !(g_dot_flags & 32) && few_wgs && MB <= 2 &&
                       64 * 1024 + (p.absmax8 ? 1024 : 0) + (xlds ? xbytes : 0) <= 80 * 1024;
#define BNB_DOT_GO(F)                                                                              \
    do {                                                                                           \
        auto kern = gemv4_dot_kernel<T, MB, RPW, SEGS, (F)>;                                       \
        const size_t dyn = (((F) & kXLds) ? xbytes : 0) + lds_pad;                                 \
        const size_t stat = (((F) & kLut64) ? 64 : 32) * 1024 + 2048;                              \
        static LdsLimit lds_limit;                                                                 \
        ensure_dynamic_lds(lds_limit, reinterpret_cast<const void*>(kern), dyn, stat);             \
        hipLaunchKernelGGL(kern, grid, block, dyn, stream, p.A, p.B, p.absmax, p.absmax8, p.code16, p.N, p.K, \
                           (p.M & 0xFFFFFF) | (p.bs_shift << 24) | (p.quant_type << 29), p);                                    \
    } while (0)
#define BNB_DOT_SEL(X)                                                                             \
    switch ((single ? 1 : 0) | (p.absmax8 ? 2 : 0)) {                                              \
    case 0: BNB_DOT_GO(E | (X)); break;                                                            \
    case 1: BNB_DOT_GO(E | kSingle | (X)); break;                                                  \
    case 2: BNB_DOT_GO(E | kNested | (X)); break;                                                  \
    default: BNB_DOT_GO(E | kSingle | kNested | (X)); break;                                       \
    }
    if constexpr (E != 0) {
        // EXPERIMENTAL diagonal-MFMA decode (debug flag 16), production workgroup size only
        if (xlds && (g_dot_flags & 16)) {
            if constexpr (MB <= 2) {
                if (lut64) {
                    BNB_DOT_SEL(kXLds | kLut64 | kDiag)
                    return;
                }
            }
            BNB_DOT_SEL(kXLds | kDiag)
            return;
        }
    }
    if (xlds) {
        if constexpr (MB <= 2) {
            if (lut64) {
                BNB_DOT_SEL(kXLds | kLut64)
                return;
            }
        }
        BNB_DOT_SEL(kXLds)
        return;
    }
    if constexpr (MB == 1) {
        if (lut64) {
            BNB_DOT_SEL(kLut64)
        } else {
            BNB_DOT_SEL(0)
        }
    } else {
        fprintf(stderr, "bitsandbytes_amd: gemv_4bit: internal error, activation image of %zu bytes does not fit\n", xbytes);
        exit(1);
    }
#undef BNB_DOT_SEL
#undef BNB_DOT_GO
}

template <typename T> void launch_generic(const GemvArgs& p, hipStream_t stream) {
    dim3 grid((p.N + 3) / 4, p.M);
    if (p.absmax8)
        hipLaunchKernelGGL((gemv4_generic_kernel<T, true>), grid, dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL((gemv4_generic_kernel<T, false>), grid, dim3(256), 0, stream, p);
}

template <typename T> void dispatch_dot(const GemvArgs& p, hipStream_t stream) {
    // profiling-only ablations of the M = 1, K <= 4096 configuration, in the production geometry
    // (one row per wavefront, 512 threads, activations in LDS)
    if (g_dot_ablate != 0 && p.M == 1 && !p.absmax8 && p.K <= 2 * kSegK) {
        dim3 grid((p.N + 7) / 8, 1);
#define BNB_ABL(ABLV)                                                                                 \
    if (g_dot_ablate == ABLV) {                                                                       \
        auto kern = gemv4_dot_kernel<T, 1, 1, 2, kSingle | kXLds | kWaves8 | kLut64 | (ABLV << 8)>;        \
        static LdsLimit lds_limit;                                                                 \
        ensure_dynamic_lds(lds_limit, reinterpret_cast<const void*>(kern), 8192, 66 * 1024);       \
        hipLaunchKernelGGL(kern, grid, dim3(512), 2 * 4096, stream, p.A, p.B, p.absmax, p.absmax8, p.code16, p.N, p.K, \
                           (p.M & 0xFFFFFF) | (p.bs_shift << 24) | (p.quant_type << 29), p);                            \
        return;                                                                                    \
    }
        BNB_ABL(1) BNB_ABL(2) BNB_ABL(3) BNB_ABL(4) BNB_ABL(5)
#undef BNB_ABL
    }

    // Geometry, calibrated on MI355X (profiles/r1_dot_ab.txt):
    //  * 512-thread workgroups (one table build and one activation copy per 8 wavefronts);
    //  * with the activations in LDS one weight row per wavefront wins at M = 1 for every shape tried
    //    (4096^2: 4.73 vs 5.35 us - twice the wavefronts to hide latency and nothing left to amortise); two
    //    activation rows keep two weight rows per wavefront on the large matrices only (4096^2 M = 2:
    //    5.47 vs 5.93 us; 8192^2: 19.1 vs 17.0; 4096 x 11008: 13.6 vs 11.4);
    //  * rows of A per pass: 4, 2 or 1 - the largest that M needs and whose LDS image fits.
    int segs = g_dot_segs;
    if (segs != 1 && segs != 2)
        segs = (p.K > kSegK) ? 2 : 1;
    int mb = (p.M >= 3) ? 4 : p.M;
    while (mb > 1 && x_image_bytes(mb, p.K, segs) > kXLdsMaxBytes)
        mb >>= 1;
    int rpw = g_dot_rpw;
    if (rpw != 1 && rpw != 2)
        rpw = (mb == 2 && static_cast<long>(p.N) * p.K > (24L << 20)) ? 2 : 1;
    if (mb == 4)
        rpw = 1;
    const int extra = (g_dot_flags & 64) ? 0 : kWaves8; // debug flag 64: 256-thread workgroups

#define BNB_DOT_CASE(MBV, RPWV, SEGSV)                                                             \
    if (mb == MBV && rpw == RPWV && segs == SEGSV) {                                               \
        if (extra)                                                                                 \
            launch_dot<T, MBV, RPWV, SEGSV, kWaves8>(p, stream);                                   \
        else                                                                                       \
            launch_dot<T, MBV, RPWV, SEGSV, 0>(p, stream);                                         \
        return;                                                                                    \
    }
    BNB_DOT_CASE(1, 1, 2) BNB_DOT_CASE(1, 2, 2) BNB_DOT_CASE(2, 1, 2) BNB_DOT_CASE(2, 2, 2) BNB_DOT_CASE(4, 1, 2)
    BNB_DOT_CASE(1, 1, 1) BNB_DOT_CASE(1, 2, 1) BNB_DOT_CASE(2, 1, 1) BNB_DOT_CASE(2, 2, 1) BNB_DOT_CASE(4, 1, 1)
#undef BNB_DOT_CASE
    launch_dot<T, 1, 1, 2, kWaves8>(p, stream); // not reached
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
