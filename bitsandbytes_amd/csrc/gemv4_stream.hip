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
//  * Segment partials of a row are combined in FIXED order (LDS slots, no atomics): results are bit-reproducible and
//    independent of the launch geometry (the reference's test_matmul_4bit_weight_orientation demands exact equality).
//
// Grouped launches (bnb_mi355x_gemm_4bit_grouped): several weight matrices that share the activations (Q/K/V,
// gate/up) are one launch over the concatenated row space - one boundary, one table build, one activation copy.
#include "bnb_common.h"

#include <type_traits>

namespace bnb {

#ifdef BNB_PROFILING
extern unsigned long long* g_dbg_buf; // gemv4.hip (profiling builds)
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
    kGrouped = 16 // several matrices over one concatenated row space
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

// Profiling builds (-DBNB_PROFILING, libbitsandbytes_mi355x_prof.so, tools/ only): per-wavefront s_memtime stamps.
#ifdef BNB_PROFILING
#define BNB_ST_STAMP(i)                                                                            \
    {                                                                                              \
        if (p.dbg && lane == 0)                                                                    \
            p.dbg[(static_cast<long>(blockIdx.x) * WAVES + wave) * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
    }
#else
#define BNB_ST_STAMP(i) {}
#endif

struct StreamArgs {
#ifdef BNB_PROFILING
    unsigned long long* dbg;
#endif
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
template <typename T, int MB, int WAVES, int NS, int FLAGS>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(WAVES / 4, WAVES / 4))) void gemv4_stream_kernel(
    // hot arguments as separate scalars: the command processor preloads them into SGPRs (kernarg preload), so a
    // wavefront does not start with a dependent s_load from a cold kernarg buffer
    const void* hot_A, const uint8_t* hot_B, const float* hot_absmax, const uint8_t* hot_absmax8, const float* hot_code16,
    int hot_N, int hot_K, int hot_packed /* M | bs_shift << 24 */, int hot_geom /* R | SW << 16 | G << 21 */,
    const StreamArgs p) {
    constexpr bool NESTED = FLAGS & kNested, CODEPTR = FLAGS & kCodePtr, NT = FLAGS & kNT, GROUPED = FLAGS & kGrouped;
    constexpr int THREADS = WAVES * 64;
    constexpr int TB = TypeInfo<T>::bytes;
    constexpr int CH = 2 * TB;  // 16-byte chunks of activations per lane and segment (= 1-KiB DMA pieces per segment)
    constexpr int EPC = 16 / TB; // elements per chunk
    constexpr int LPS = NESTED ? 3 : 2; // vector loads per ring stage

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = hot_K;
    const int M = hot_packed & 0xFFFFFF, bs_shift = (hot_packed >> 24) & 31;
    const int R = hot_geom & 0xFFFF, SW = (hot_geom >> 16) & 31, G = (hot_geom >> 21) & 31;
    const int S = (K + kSegK - 1) / kSegK;
    const int P = (S + SW - 1) / SW;
    const int rows_total = GROUPED ? p.rows_total : hot_N;
    const int row_begin = blockIdx.x * R;
    if (row_begin >= rows_total)
        return;
    const int nrows = (rows_total - row_begin < R) ? rows_total - row_begin : R;
    const int m0 = blockIdx.y * MB;
    const int sw = wave % SW, g = wave / SW;
    BNB_ST_STAMP(0)

    // LDS map: table | nested code(s) (1 KiB per matrix) | segment partials [R][S][MB] | activation image [MB][SW][2048] T
    constexpr int kCode2Bytes = (GROUPED ? kMaxGroup : 1) * 1024;
    float* const code2 = reinterpret_cast<float*>(smem + kLutBytes);
    float* const part = reinterpret_cast<float*>(smem + kLutBytes + kCode2Bytes);
    const int part_bytes = (R * S * MB * 4 + 15) & ~15;
    unsigned char* const ximg = smem + kLutBytes + kCode2Bytes + part_bytes;

    const T* __restrict__ A = static_cast<const T*>(hot_A);

    // ---- loads that must not sit behind the weight stream: oldest in the queue
    // (nested: entry tid & 255 of the 256-entry absmax code of matrix tid >> 8, + THREADS / 256 matrices per pass)
    constexpr int C2PASS = (kMaxGroup * 256 + THREADS - 1) / THREADS;
    float code2_v[NESTED ? C2PASS : 1];
    float cv = 0.0f;
    const int nmat = GROUPED ? p.nmat : 1;
    if constexpr (NESTED) {
#pragma unroll
        for (int i = 0; i < C2PASS; ++i) {
            const int t = i * THREADS + tid;
            code2_v[i] = (t < nmat * 256) ? p.mat[GROUPED ? (t >> 8) : 0].absmax_code[t & 255] : 0.0f;
        }
    }
    if constexpr (CODEPTR)
        cv = hot_code16[lane & 15];
    else
        cv = code_literal<(FLAGS & kFp4) != 0>(lane & 15);

    // ---- activation image of phase ph: LDS-DMA, one 1-KiB piece per instruction. The image is lane-linear per
    // piece (hardware), so the bank swizzle is applied on the SOURCE side: slot s = CH l' + j of a segment row
    // holds chunk CH l' + (j ^ f(l')) - a permutation inside 64- / 128-byte groups, the copy stays fully coalesced -
    // and lane l finds its q-th chunk at slot CH l + (q ^ f(l)): conflict-free ds_read_b128 (f(l) = (l >> 2) & 3 for
    // 16-bit activations as in round 1's kernel, (l >> 1) & 7 for fp32). Spelled in asm: with the builtin the
    // compiler sees DMA and ordinary loads on one counter and turns every later register wait into vmcnt(0).
    auto swz = [](int l) -> int { return (CH == 4) ? ((l >> 2) & 3) : ((l >> 1) & 7); };
    auto issue_x = [&](int ph) {
        const int segs = (S - ph * SW < SW) ? S - ph * SW : SW;
        const int pieces = MB * segs * CH;
        for (int piece = wave; piece < pieces; piece += WAVES) {
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
        float s;  // fp32 absmax of the lane's block (nested: the uint8 code in the low byte)
        float s2; // nested: second-level absmax
    };
    Stage st[NS];
    constexpr uint32_t kOob = 0xFFFFFFF0u; // beyond any num_records (packed weights of < 2^31 elements are < 2^30 bytes)
    constexpr int kRsrcFlags = 0x00020000;
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
    int n_items = 0;
    auto items_of = [&](int sg) -> int { return (g < G && sg < S && g < nrows) ? (nrows - g + G - 1) / G : 0; };

    auto issue = [&](Stage& s, int i) {
        const bool valid = i < n_items; // wave-uniform
        const int grow = row_begin + g + i * G;
        const uint8_t* Bp = hot_B;
        const float* am = hot_absmax;
        const uint8_t* am8 = hot_absmax8;
        int row = grow, nmat_rows = hot_N;
        if constexpr (GROUPED) {
            const int mi = mat_of(valid ? grow : 0);
            Bp = p.mat[mi].B;
            am = p.mat[mi].absmax;
            am8 = p.mat[mi].absmax8;
            row = grow - p.mat[mi].row_start;
            nmat_rows = p.mat[mi].N;
        }
        row = valid ? row : 0;
        const long elems = static_cast<long>(nmat_rows) * K;
        const long blocks = (elems + (1L << bs_shift) - 1) >> bs_shift;
        const int k0 = seg * kSegK + lane * 32;
        const bool lane_ok = valid && (k0 < K);
        const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(Bp), 0, static_cast<int>(elems >> 1), kRsrcFlags);
        s.w = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                            rs_w, lane_ok ? static_cast<uint32_t>(k0 >> 1) : kOob,
                                            static_cast<uint32_t>(row) * static_cast<uint32_t>(K >> 1), kWeightAux));
        const uint32_t blk = static_cast<uint32_t>((static_cast<uint32_t>(row) * static_cast<uint32_t>(K) + static_cast<uint32_t>(k0)) >> bs_shift);
        if constexpr (NESTED) {
            const auto rs_q = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(am8), 0, static_cast<int>(blocks), kRsrcFlags);
            const auto rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(am), 0, static_cast<int>(((blocks + 255) >> 8) * 4), kRsrcFlags);
            s.s = __builtin_bit_cast(float, static_cast<uint32_t>(__builtin_amdgcn_raw_buffer_load_b8(rs_q, lane_ok ? blk : kOob, 0, 0)));
            s.s2 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_a, lane_ok ? (blk >> 8) * 4u : kOob, 0, 0));
        } else {
            const auto rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(am), 0, static_cast<int>(blocks * 4), kRsrcFlags);
            s.s = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_a, lane_ok ? blk * 4u : kOob, 0, 0));
        }
    };

    // The table is addressed with the raw v_perm_b32 result: it must sit at LDS address 0 (this kernel has no static
    // LDS, so the dynamic segment starts there).
    if (reinterpret_cast<uintptr_t>((lds_ptr)smem) != 0)
        __builtin_trap();

    uint32_t perm_sel = 0x0C0C0400u; // v_perm_b32 selector {lane offset, weight byte j, 0, 0}
    const uint32_t lane_off = static_cast<uint32_t>(lane & 31) * 8u;
    float offset = 0.0f;
    float xr[MB][32];

    auto load_slice = [&]() {
        const int k0 = seg * kSegK + lane * 32;
        const bool act = (seg < S) && (k0 < K);
#pragma unroll
        for (int m = 0; m < MB; ++m) {
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                const int slot = CH * lane + (q ^ swz(lane));
                const u32x4 v = *reinterpret_cast<const u32x4*>(ximg + ((m * SW + sw) * CH * 64 + slot) * 16);
                unpack16<T>(v, &xr[m][q * EPC]);
            }
            // lanes past the end of the row hold zeros (their weight loads are out of range, their scale is forced to 0)
#pragma unroll
            for (int e = 0; e < 32; ++e)
                xr[m][e] = act ? xr[m][e] : 0.0f;
        }
    };

    auto compute = [&](const Stage& s, int i) {
        // all 16 look-ups of the item first (one v_perm_b32 + one ds_read_b64 per packed byte), then the FMAs: the
        // LDS round trip is paid once per item, not once per byte (left alone, the scheduler serialises them)
        f32x2 pr[16];
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t addr = __builtin_amdgcn_perm(s.w[d], lane_off, perm_sel + (j << 8));
                pr[4 * d + j] = *reinterpret_cast<const __attribute__((address_space(3))) f32x2*>(addr);
            }
        __builtin_amdgcn_sched_barrier(0);
        float acc[MB][4];
#pragma unroll
        for (int m = 0; m < MB; ++m)
            acc[m][0] = acc[m][1] = acc[m][2] = acc[m][3] = 0.0f;
#pragma unroll
        for (int b = 0; b < 16; ++b)
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                acc[m][(b & 1) * 2] = fmaf(pr[b][0], xr[m][2 * b], acc[m][(b & 1) * 2]);
                acc[m][(b & 1) * 2 + 1] = fmaf(pr[b][1], xr[m][2 * b + 1], acc[m][(b & 1) * 2 + 1]);
            }
        const int grow = row_begin + g + i * G;
        float scale;
        if constexpr (NESTED) {
            const uint32_t q8 = __builtin_bit_cast(uint32_t, s.s);
            if constexpr (GROUPED) {
                const int mi = mat_of(grow);
                scale = __fadd_rn(__fmul_rn(code2[mi * 256 + q8], s.s2), p.mat[mi].absmax_offset[0]);
            } else {
                scale = __fadd_rn(__fmul_rn(code2[q8], s.s2), offset);
            }
        } else {
            scale = s.s;
        }
        const int k0 = seg * kSegK + lane * 32;
        scale = (k0 < K) ? scale : 0.0f;
        const int rl = g + i * G;
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const float v = wave_sum(((acc[m][0] + acc[m][1]) + (acc[m][2] + acc[m][3])) * scale);
            if (lane == 0)
                part[(rl * S + seg) * MB + m] = v;
        }
    };

    for (int ph = 0; ph < P; ++ph) {
        if (ph > 0)
            __syncthreads(); // everyone is done with the previous activation image
        // (1) this phase's activation image (LDS-DMA, oldest in the queue), then the first NS ring stages
        if (ph == 0)
            BNB_ST_STAMP(9)
        issue_x(ph);
        if (ph == 0)
            BNB_ST_STAMP(10)
        seg = ph * SW + sw;
        n_items = items_of(seg);
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            issue(st[j], j);
            __builtin_amdgcn_sched_barrier(0); // keep the queue in stage order: (weights, scale) of stage 0, of stage 1, ...
            if (ph == 0 && j == 0)
                BNB_ST_STAMP(11)
            if (ph == 0 && j == 1)
                BNB_ST_STAMP(12)
        }
        if (ph == 0)
            BNB_ST_STAMP(1)
        if (ph == 0) {
            // (2) decode table, built while the loads fly: entry e (a packed byte) = 32 copies of
            // (code[e >> 4], code[e & 15]) in fp32, 256 B per entry, copy c at byte 8 c. Chunk c16 of the table
            // (16 B = two copies) is written by thread c16 % THREADS: every ds_write_b128 of a wavefront covers 1 KiB
            // contiguous, conflict-free. The two code values come from lanes (e >> 4) and (e & 15) of `cv`.
            static_assert((kLutBytes / 16) % THREADS == 0, "whole passes over the table");
#pragma unroll
            for (int it = 0; it < kLutBytes / 16 / THREADS; ++it) {
                const int c16 = it * THREADS + tid;
                const int e = c16 >> 4;
                const float hi = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((e >> 4) * 4, __builtin_bit_cast(int, cv)));
                const float lo = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((e & 15) * 4, __builtin_bit_cast(int, cv)));
                *reinterpret_cast<f32x4*>(smem + c16 * 16) = f32x4{hi, lo, hi, lo};
            }
            if constexpr (NESTED) {
#pragma unroll
                for (int i = 0; i < C2PASS; ++i) {
                    const int t = i * THREADS + tid;
                    if (t < nmat * 256)
                        code2[t] = code2_v[i];
                }
                if constexpr (!GROUPED)
                    offset = p.mat[0].absmax_offset[0];
            }
        }
        // (3) the activation DMAs are older than the NS ring stages: wait until only those remain in flight
        if (ph == 0)
            BNB_ST_STAMP(2)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS * LPS) : "memory");
        __syncthreads();
        if (ph == 0)
            BNB_ST_STAMP(3)
        // an opaque zero ties the decode to program order after the barrier (without it LLVM hoists the first
        // look-up address - and its vmcnt wait - above the table build)
        perm_sel = 0x0C0C0400u + static_cast<uint32_t>(opaque_zero());
        load_slice();
        if (ph == 0)
            BNB_ST_STAMP(4)
        // (4) rounds of NS items; the refill of a stage is issued right after the stage was consumed, valid or not
        for (int base = 0; base < n_items; base += NS) {
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                if (base + j < n_items)
                    compute(st[j], base + j);
                if (ph == 0 && base == 0 && j == 0)
                    BNB_ST_STAMP(5)
                issue(st[j], base + j + NS);
            }
        }
    }
    BNB_ST_STAMP(6)

    // ---- combine the segment partials of every row in segment order, bias, one rounding
    __syncthreads();
    BNB_ST_STAMP(7)
    for (int idx = tid; idx < nrows * MB; idx += THREADS) {
        const int m = idx / nrows, rl = idx - m * nrows;
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
        static_cast<T*>(p.mat[mi].out)[static_cast<long>(m0 + m) * p.mat[mi].N + row] = static_cast<T>(v + b);
    }
    BNB_ST_STAMP(8)
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
            return __fadd_rn(__fmul_rn(code2[p.absmax8[blk]], p.absmax[blk >> 8]), p.absmax_offset[0]);
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
int device_cu_count() {
    static std::atomic<int> cached[LdsLimit::kMaxDevices] = {};
    int dev = 0;
    BNB_HIP_CHECK(hipGetDevice(&dev));
    const int slot = (dev >= 0 && dev < LdsLimit::kMaxDevices) ? dev : 0;
    int v = cached[slot].load(std::memory_order_relaxed);
    if (v == 0) {
        BNB_HIP_CHECK(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
        v = v > 0 ? v : 256;
        cached[slot].store(v, std::memory_order_relaxed);
    }
    return v;
}

constexpr size_t kLdsBudget = 156 * 1024; // largest dynamic allocation that launches (157 KiB is refused)

struct Geometry {
    int R, SW, G, grid_x;
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
    const size_t fixed = kLutBytes + (grouped ? kMaxGroup : 1) * 1024;
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
    ge.grid_x = (rows_total + R - 1) / R;
    ge.lds = fixed + ((static_cast<size_t>(R) * S * mb * 4 + 15) & ~size_t(15)) + ximg;
    return ge;
}

struct StreamTuning {
    std::atomic<int> ns{0}, sw{0}, rows{0}, nt{-1}, waves{0};
};
StreamTuning g_tune;

template <typename T, int MB, int WAVES, int NS, int FLAGS> void launch_one(const StreamArgs& a, hipStream_t stream) {
    const Geometry ge = make_geometry(a.rows_total, a.K, MB, WAVES, TypeInfo<T>::bytes, (FLAGS & kGrouped) != 0, g_tune.sw.load(std::memory_order_relaxed),
                                      g_tune.rows.load(std::memory_order_relaxed));
    auto kern = gemv4_stream_kernel<T, MB, WAVES, NS, FLAGS>;
    static LdsLimit lds_limit;
    ensure_dynamic_lds(lds_limit, reinterpret_cast<const void*>(kern), ge.lds);
    dim3 grid(ge.grid_x, (a.M + MB - 1) / MB);
    const StreamMat& m0 = a.mat[0];
    hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), ge.lds, stream, a.A, m0.B, m0.absmax, m0.absmax8, a.code16, m0.N, a.K,
                       (a.M & 0xFFFFFF) | (a.bs_shift << 24), ge.R | (ge.SW << 16) | (ge.G << 21), a);
}

// Production instances: ring depth kRing, non-temporal weight loads, 16 wavefronts at MB = 1 and 8 above.
// The sweep-only variants (other ring depths, default cache policy, 8 wavefronts) exist for ONE configuration -
// bf16, one activation row, fp32 absmax, literal NF4 table - so that tools/ can A/B them without multiplying the
// instance count of the library.
constexpr int kRing = 4;

template <typename T, int MB, int WAVES, int FLAGS> void launch_tuned(const StreamArgs& a, hipStream_t stream) {
    if constexpr (std::is_same<T, bf16>::value && MB == 1 && FLAGS == 0) {
        const int ns = g_tune.ns.load(std::memory_order_relaxed);
        const int nt = g_tune.nt.load(std::memory_order_relaxed);
        const int tw = g_tune.waves.load(std::memory_order_relaxed);
        if (tw == 8) {
            if (nt == 0)
                return launch_one<T, 1, 8, kRing, 0>(a, stream);
            return launch_one<T, 1, 8, kRing, kNT>(a, stream);
        }
        if (ns == 2)
            return launch_one<T, 1, 16, 2, kNT>(a, stream);
        if (ns == 3)
            return launch_one<T, 1, 16, 3, kNT>(a, stream);
        if (ns == 6)
            return launch_one<T, 1, 16, 6, kNT>(a, stream);
        if (nt == 0)
            return launch_one<T, 1, 16, kRing, 0>(a, stream);
    }
    launch_one<T, MB, WAVES, kRing, FLAGS | kNT>(a, stream);
}

template <typename T, int MB, int WAVES> void launch_flags(const StreamArgs& a, int quant_type, bool grouped, hipStream_t stream) {
    const bool nested = a.mat[0].absmax8 != nullptr;
    const bool fp4 = quant_type == kFP4;
    if (a.code16 != nullptr) {
        // caller-supplied code table (legacy gemv_4bit op): one activation row, un-nested absmax by construction
        if constexpr (MB == 1) {
            if (!nested && !grouped)
                return launch_one<T, 1, WAVES, kRing, kCodePtr | kNT>(a, stream);
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
    if (a.M == 1 || a.code16 != nullptr)
        return launch_flags<T, 1, 16>(a, quant_type, grouped, stream);
    if (a.M == 2)
        return launch_flags<T, 2, 8>(a, quant_type, grouped, stream);
    return launch_flags<T, 4, 8>(a, quant_type, grouped, stream);
}

template <typename T> void launch_generic(const GenericArgs& p, hipStream_t stream) {
    dim3 grid((p.N + 3) / 4, p.M);
    if (p.absmax8)
        hipLaunchKernelGGL((gemv4_generic_kernel<T, true>), grid, dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL((gemv4_generic_kernel<T, false>), grid, dim3(256), 0, stream, p);
}

bool stream_ok(const void* A, int K, int blocksize) {
    return (K % 32 == 0) && blocksize >= 32 && is_pow2(blocksize) && aligned_to(A, 16);
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

} // namespace

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
    if (stream_ok(A, K, blocksize) && aligned_to(B, 16)) {
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
        return;
    }
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
    if (count < 1 || count > kMaxGroup || M < 1 || M > 4 || K <= 0 || !stream_ok(A, K, blocksize))
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
    launch_stream_any(dtype, a, quant_type, true, stream);
    return true;
}

} // namespace bnb
