// gemm4_mfma_rt.hip — "register-transposed" MFMA kernel ("v6" in profiles/) for small batches on small and medium
// matrices:  out[m, n] = sum_k A[m, k] * code[B[n, k]] * scale[n, k / bs]  (+ bias[n]),  bf16 / fp16, K % 256 == 0.
//
// Fills, on MI355X, the tensor-core capability the reference only has on CUDA (csrc/gemm_4bit_sm80.cu:127-457) for the
// batch sizes between the streaming dot kernel (gemv4_stream.hip, M <= 2) and the producer/consumer MFMA kernel
// (gemm4_mfma.hip, large matrices / tall tiles). What round 1's LDS-DMA kernel measured (profiles/r1_timeline_mfma_v3_*):
// a wavefront spent ~5500 of its ~11000 cycles before its loads were even issued (a dependent global load in front of the
// decode table, integer divisions and per-lane address arithmetic on a scalar unit shared by 16 wavefronts, ~150 cycles per
// LDS-DMA issue) and ~2600 in a serial reduction. This kernel is built the way the streaming kernel was:
//
//  * ONE workgroup of 16 wavefronts per 16 output columns, no split-K between workgroups when N / 16 fills the chip (no
//    slab round trip, no second launch); the wavefronts split K in 256-k chunks and combine once through LDS.
//  * Every global load is a fully coalesced 16-byte-per-lane load whose four neighbouring lanes cover 64 contiguous
//    bytes of ONE row (lane 4r + p: row r, piece p): 16 L1 tag look-ups per instruction. The MFMA operand layout wants the
//    row index in the LOW lane bits (lane r + 16 g) - loading in that shape costs 64 tag look-ups per instruction (four
//    different rows per lane quad), which is what made round 1's fragment-shaped loads crawl. The transposition
//    (4r + p -> r + 16p) is a ds_write_b128 / ds_read_b128 pair through a 1-KiB tile private to the wavefront: same
//    wavefront, in-order LDS, no barrier, bank-conflict-free both ways (slot 16p + (r ^ 2p)).
//  * K order inside an MFMA is free as long as both operands agree. A lane's 16 weight bytes are 32 consecutive k; after
//    the transposition the four lane groups of a column hold 128 consecutive k = two quantization blocks. One
//    v_permlane32_swap per dword pair regroups them so that every MFMA consumes k from ONE block (dwords 0/1 of groups 0,1
//    + dwords 2/3 of groups 0,1 seen from groups 2,3), i.e. the fp32 absmax is applied exactly, per lane, to the partial
//    tile of each 64-k block. The activation loads fetch the matching k (16-byte pieces at a 32-byte stride inside one
//    128-byte line), so no data is ever permuted on the A side.
//  * The decode table (byte -> the pair (code[hi], code[lo]) in T, 32 bank-private copies) is built from compile-time
//    literals while the loads fly: no memory dependency in front of it. Loads are issued before anything else.
//  * No LDS-DMA and no inline-asm waits: every wait is the compiler's counted vmcnt - which it can only compute for
//    straight-line code, so every load is a BRANCH-FREE buffer load (round 3, see BL / BS64 below and DESIGN.md 6b: what
//    branches around loads, run-time format branches and non-uniform scalar offsets did to the waits of round 2's loop);
//    tests/test_cabi.py::test_rt_kernel_isa_keeps_its_loads_in_flight checks the disassembly of the built library.
//
// Results are bit-reproducible: partial tiles are combined in wavefront order, K slices (only when N / 16 workgroups would
// leave most of the chip idle) through fp32 slabs added in slice order by gemm4_finalize (gemm4_mfma.hip).
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
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <typename T> struct RtMma;
template <> struct RtMma<bf16> {
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
template <> struct RtMma<f16> {
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

constexpr int kRtLut = 32768;            // 256 entries x 32 copies x 4 B, at LDS address 0
constexpr int kRtScratch = 2 * 1024 + 256; // per wavefront: two transposition tiles + the scale tile (16 x 16 B)
constexpr int kRtScratch32 = kRtScratch + 256; // blocksize 32: eight scales per row and chunk - two scale tiles
constexpr int kRtChunk = 256;            // k per wavefront step: four 64-k MFMA pairs

struct RtArgs {
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
#define BNB_RT_STAMP(i)                                                                            \
    {                                                                                              \
        if (p.dbg && lane == 0)                                                                    \
            p.dbg[((static_cast<long>(blockIdx.z) * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * WAVES * 16 + wave * 16 + (i)] = \
                __builtin_amdgcn_s_memtime();                                                      \
    }
#else
#define BNB_RT_STAMP(i) {}
#endif

__device__ __forceinline__ float rt_code_literal(int i, bool fp4) {
    // compare/select over literals: no memory access (the inner select is scalar)
    constexpr float nf4[16] = {BNB_NF4_VALUES};
    constexpr float fp4v[16] = {BNB_FP4_VALUES};
    float v = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j)
        v = (i == j) ? (fp4 ? fp4v[j] : nf4[j]) : v;
    return v;
}

// T in {bf16, f16}; MT = 16-row tiles of the batch handled by one workgroup (the weights of a chunk are loaded, transposed
// and decoded ONCE and multiplied with MT activation tiles in turn - while tile t is transposed and multiplied, the loads of
// tile t + 1 refill the same registers); WAVES wavefronts split the workgroup's K range chunk by chunk (chunk c of the slice
// goes to wavefront c % WAVES). grid = (ceil(N / 16), kslices, ceil(M / (16 MT))).
//
// DIRECT (M <= 6, M <= 8 on small matrices - RtPlan::direct_max; one row tile): the activation fragments are loaded straight in
// MFMA shape, no transposition. Only the lanes of rows < M fetch (the others carry an out-of-range offset, see BL), so a load
// instruction touches few lines either way - the fragment-shaped load is only expensive when all 64 lanes take part - and
// 8 ds_write_b128 + 8 ds_read_b128 per chunk and wavefront disappear from the LDS store path (13 cycles per wave-instruction).
//
// BL (every instance of the product library; the measurement build also has BL = false, round 2's form, for A/B runs -
// bnb_mi355x_set_tuning knob0 bit 0): weights and activations are
// fetched through buffer descriptors - a 32-bit per-lane offset computed once plus a scalar offset per chunk / row tile instead
// of 64-bit per-lane address arithmetic in front of every load (16 wavefronts share four VALUs when the kernel starts) - and,
// what matters more, WITHOUT branches: a row past the end of the batch, or a wavefront without a chunk, is an out-of-range
// offset (zeros, nothing fetched) instead of an exec-masked region. With branches around the loads the compiler cannot count
// what is in flight at the top of the chunk loop and drains the queue there (vmcnt(1), vmcnt(0) in the ISA of round 2's loop):
// the weights - requested first - were not transposed and looked up before the activation fragments - requested last, ~4000
// cycles into the kernel (profiles/r3_timeline_rt_direct.txt) - had landed too. Now the top of the loop waits vmcnt(9), (8)
// for the weights and 7 ... 0 fragment by fragment (profiles/r3_rt_branch_free_loads_ab.txt).
//
// BS64 (nested instances only; the others decide at run time): the blocksize is 64 - one dword of 8-bit codes per chunk and row
// instead of two bytes. A compile-time choice: as a run-time branch the two paths loaded into the same registers, and at their
// join the compiler drained the queue (vmcnt(0) between a chunk's weight request and its scale / activation requests: every
// chunk of a nested call paid an extra memory round trip - 4096^2 M = 3 6.7 us nested against 6.25 plain).
//
// BS32 (round 5; fp32 absmax only): blocksize 32 - the reference's fused kernels take any power-of-two blocksize
// (csrc/gemm_4bit_simt.cu:208,225; functional.py:884-969 accepts 32), here such calls ran the streaming kernel in 4-row passes. An
// MFMA contracts over its 32 k at once, so a 32-k block has to BE one MFMA: after the transposition lane group lg holds block lg of
// a 128-k half in its four dwords (dword d = k 32 lg + 8 d ...), the MFMA of block j wants k 32 j + 8 g from group g - dword g of
// group j: a 4 x 4 transposition between lane groups and dwords. Stage one is the regrouping the other instances do anyway
// (v_permlane32_swap on the dword pairs (0, 2) and (1, 3): 2 x 2 blocks change halves), stage two a v_permlane16_swap on the pairs
// (0, 1) and (2, 3) (odd 16-lane rows of the first against even rows of the second; semantics probed on the device,
// tools/ubench/swap16_probe.hip, and the whole algebra replayed lane by lane in tests/checks/emulate_rt_mfma.py). The activation
// fragment of step s is then the plain one: lane group g fetches A[row][32 s + 8 g ...] - four lanes cover 64 contiguous bytes.
// Eight scales per row and chunk (two 16-byte loads, two scale tiles), one scale FMA set per MFMA instead of one per pair.
template <typename T, int MT, int WAVES, bool NESTED, bool DIRECT, bool BL, bool BS64, bool BS32 = false>
__global__ __launch_bounds__(WAVES * 64) void gemm4_mfma_rt_kernel(
    // hot arguments as separate scalars: preloaded into SGPRs by the command processor (14 dwords)
    const void* hot_A, const uint8_t* hot_B, const float* hot_absmax, const uint8_t* hot_absmax8, int hot_M, int hot_N,
    int hot_K, int hot_flags /* bs_shift | fp4 << 8 */, int hot_cps /* chunks per K slice */, int hot_kslices,
    const RtArgs p) {
    constexpr int THREADS = WAVES * 64;
    static_assert(!BS32 || (!NESTED && !BS64 && BL), "blocksize 32: fp32 absmax, branch-free loads");
    constexpr int SCRATCH = BS32 ? kRtScratch32 : kRtScratch;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    BNB_RT_STAMP(0)
    const int r = lane >> 2, pp = lane & 3; // load roles: row r of the 16-row tile, 16-byte piece pp of its 64 bytes
    const int ln = lane & 15, lg = lane >> 4; // MFMA roles: row / column ln, k group lg
    const int M = hot_M, N = hot_N, K = hot_K;
    const int bs_shift = hot_flags & 31;
    const bool fp4 = (hot_flags >> 8) & 1;
    const int col0 = blockIdx.x * 16;
    const int m_base = blockIdx.z * (16 * MT);
    const int cb = blockIdx.y * hot_cps;
    int ce = cb + hot_cps;
    ce = (ce < (K >> 8)) ? ce : (K >> 8);

    // LDS map: table | per-wavefront scratch (tile 0, tile 1, scale tile) | nested absmax code (1 KiB) | parked partial tiles
    unsigned char* const sc = smem + kRtLut + wave * SCRATCH;
    u32x4* const tile0 = reinterpret_cast<u32x4*>(sc);
    u32x4* const tile1 = reinterpret_cast<u32x4*>(sc + 1024);
    u32x4* const stile = reinterpret_cast<u32x4*>(sc + 2048);
    float* const code2 = reinterpret_cast<float*>(smem + kRtLut + WAVES * SCRATCH);
    unsigned char* const red = smem + kRtLut + WAVES * SCRATCH + 1024; // [WAVES][MT][64 lanes][16 B]

    // ---- sources. Rows past the end (ragged N or M) re-read the last row: MFMA rows / columns are independent and those
    // results are never stored, so no masking instructions are needed.
    int wrow = col0 + r;
    wrow = (wrow < N) ? wrow : N - 1;
    const uint8_t* const wsrc = hot_B + static_cast<long>(wrow) * (K >> 1) + pp * 16;
    static_assert(!DIRECT || MT == 1, "direct activation fragments: one row tile");
    // (the lane that fetches row x / k group y of a fragment: the load roles (r, pp), or - DIRECT - the MFMA roles (ln, lg))
    const int arow_l = DIRECT ? ln : r, ag_l = DIRECT ? lg : pp;
    // (k of the lane's 8 elements inside a step: the pair-interleaved order of the 64-k regrouping, or - BS32 - simply 8 ag_l)
    const int a_lane_k = BS32 ? 8 * ag_l : (ag_l & 1) * 32 + (ag_l >> 1) * 16;
    const T* const abase = static_cast<const T*>(hot_A) + a_lane_k;
    const long e0 = static_cast<long>(wrow) * K; // flat element index of the row start
    // (BL: byte offsets are 32-bit and stay below the descriptors' 2^31 records - gemm_4bit_rt_supported)
    constexpr uint32_t kOob = 0xFFFFFFF0u; // beyond num_records
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(hot_B), 0, 0x7FFFFFFF, 0x00020000);
    const auto rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(hot_A), 0, 0x7FFFFFFF, 0x00020000);
    const uint32_t w_off = static_cast<uint32_t>(wrow) * static_cast<uint32_t>(K >> 1) + static_cast<uint32_t>(pp * 16);
    const uint32_t a_off = (static_cast<uint32_t>(m_base + arow_l) * static_cast<uint32_t>(K) + static_cast<uint32_t>(a_lane_k)) * static_cast<uint32_t>(sizeof(T));

    struct Raw {
        u32x4 w[2]; // 128 k each: lane (r, pp) holds k [128 h + 32 pp, + 32) of row r
        u32x4 s;    // the row's scales of the chunk's four 64-k sub-blocks (nested: {4 x uint8, second-level absmax}; BS32: blocks 0 - 3)
        u32x4 s2;   // BS32: the scales of the chunk's 32-k blocks 4 - 7
        u32x4 a[8]; // step (h, j): lane (r, pp) holds A[r][128 h + 64 (j >> 1) + 8 (j & 1) + 32 (pp & 1) + 16 (pp >> 1) + 0..7]
    };
    // Activation rows past the end of the batch are not fetched at all (an out-of-range offset - exec-masked in round 2's form:
    // the loads then cost the L1 M / 16 of a full tile); those lanes hold zeros and the MFMA rows they feed are never stored.
    auto load_a_step = [&](Raw& raw, int c, int mt, int s) {
        const int row = m_base + mt * 16 + arow_l;
        if constexpr (BL) {
            // Branch-free: a row past the end of the batch is an out-of-range offset (zeros, nothing fetched), not an exec-masked
            // region. With branches around the loads the compiler cannot count what is in flight at the top of the chunk loop and
            // waits for EVERYTHING there (vmcnt(0) in the ISA of rounds 2's loop): the weights - requested first - could not be
            // transposed and looked up before the activation fragments - requested last - had landed too.
            const uint32_t inval = (row < M && c < ce) ? 0u : kOob;
            raw.a[s] = __builtin_bit_cast(
                u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                           rs_a, (a_off + static_cast<uint32_t>((BS32 ? 32 * s : 128 * (s >> 2) + 64 * ((s >> 1) & 1) + 8 * (s & 1)) * sizeof(T))) | inval,
                           // (wavefront-uniform by construction; said explicitly, or the scalar offset arrives in a VGPR and every
                           // load is wrapped in a readfirstlane loop - seen in the ISA of the first build)
                           __builtin_amdgcn_readfirstlane((static_cast<uint32_t>(c) * static_cast<uint32_t>(kRtChunk) +
                                                           static_cast<uint32_t>(mt * 16) * static_cast<uint32_t>(K)) *
                                                          static_cast<uint32_t>(sizeof(T))),
                           0));
            return;
        }
        if (row < M)
            raw.a[s] = *reinterpret_cast<const u32x4*>(abase + static_cast<long>(row) * K + static_cast<long>(c) * kRtChunk +
                                                       (BS32 ? 32 * s : 128 * (s >> 2) + 64 * ((s >> 1) & 1) + 8 * (s & 1)));
        else
            raw.a[s] = u32x4{0, 0, 0, 0};
    };
    auto load_a = [&](Raw& raw, int c, int mt) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
            load_a_step(raw, c, mt, s);
    };
    auto issue = [&](Raw& raw, int c) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if constexpr (BL)
                raw.w[h] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                         rs_w, w_off | (c < ce ? 0u : kOob),
                                                         __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(c) * 128u + static_cast<uint32_t>(h * 64)), 0));
            else
                raw.w[h] = *reinterpret_cast<const u32x4*>(wsrc + static_cast<long>(c) * 128 + h * 64);
        }
        // (BL: a wavefront without a chunk issues the same loads - weights and activations out of range, the scales of the last
        // chunk again - so that the start-up has no branch around loads either)
        const long e = e0 + (static_cast<long>(BL && c >= ce ? ce - 1 : c) << 8);
        if constexpr (BS32) {
            raw.s = *reinterpret_cast<const u32x4*>(hot_absmax + (e >> 5));
            raw.s2 = *reinterpret_cast<const u32x4*>(hot_absmax + (e >> 5) + 4);
        } else if (NESTED ? BS64 : bs_shift == 6) {
            if constexpr (NESTED) {
                raw.s[0] = *reinterpret_cast<const uint32_t*>(hot_absmax8 + (e >> 6));
                raw.s[1] = __builtin_bit_cast(uint32_t, hot_absmax[e >> 14]);
                raw.s[2] = raw.s[3] = 0;
            } else {
                raw.s = *reinterpret_cast<const u32x4*>(hot_absmax + (e >> 6));
            }
        } else {
            if constexpr (NESTED) {
                // The bytes stay apart until the chunk is consumed: packed here, the wavefront waited for them here - right
                // behind their own request and in front of the activation loads (nested blocksize-128 calls ran 0.5 us behind
                // blocksize 64, profiles/r3_rt_prefetch_ab.txt). A chunk's four 64-k sub-blocks lie in two blocks at blocksize
                // 128 and in one above: sub-blocks 0 and 2 are fetched, 1 = 0 and 3 = 2.
                raw.s[0] = hot_absmax8[e >> bs_shift];
                raw.s[2] = hot_absmax8[(e + 128) >> bs_shift];
                raw.s[1] = __builtin_bit_cast(uint32_t, hot_absmax[(e >> bs_shift) >> 8]);
                raw.s[3] = 0;
            } else {
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    raw.s[b] = __builtin_bit_cast(uint32_t, hot_absmax[(e + b * 64) >> bs_shift]);
            }
        }
        load_a(raw, c, 0);
    };

    float offset = 0.0f;
    Raw raw;
    int c = cb + wave;
    if (BL || c < ce)
        issue(raw, c);
    BNB_RT_STAMP(1)
    __builtin_amdgcn_sched_barrier(0); // nothing that is not needed for the loads runs before them

    // ---- decode table, built while the loads fly: entry e (a packed byte) = 32 copies of (T(code[e >> 4]), T(code[e & 15])),
    // 128 B per entry. Thread idx writes chunks (2 sub) ^ (e & 1) and (2 sub + 1) ^ (e & 1) of entry e = idx >> 2: the eight
    // lanes one ds_write_b128 services together land in eight different bank quads.
    {
        // (literals only: a caller-supplied code table would put a load - and at the join of the two paths a vmcnt(0)
        // that drains the weight loads - in front of the table; such calls go to the LDS-DMA kernels)
        const float cv = rt_code_literal((lane & 15) + opaque_zero(), fp4);
        const int cvb = __builtin_bit_cast(int, cv);
        u32x4* const lut = reinterpret_cast<u32x4*>(smem);
        // (the wavefronts of a 16-wavefront workgroup start ~90 cycles apart and every one of them first pushes its loads
        // through the CU's address pipeline: the early half builds the table, the late half only issues loads - the barrier
        // then waits for the last issuer, not for the last issuer's share of the table)
        constexpr int BUILD_THREADS = (WAVES >= 16) ? THREADS / 2 : THREADS;
#pragma unroll
        for (int idx = tid; idx < 1024 && tid < BUILD_THREADS; idx += BUILD_THREADS) {
            const int e = idx >> 2, sub = idx & 3;
            const float hi = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((e >> 4) * 4, cvb));
            const float lo = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((e & 15) * 4, cvb));
            const uint32_t pr = RtMma<T>::pack(hi, lo);
            const u32x4 v = {pr, pr, pr, pr};
            lut[e * 8 + ((2 * sub) ^ (e & 1))] = v;
            lut[e * 8 + ((2 * sub + 1) ^ (e & 1))] = v;
        }
    }
    if constexpr (NESTED) {
        // (requested behind the chunk's loads: its wait drains the queue in front of the table barrier. Both alternatives measured
        // slower: in FRONT of the chunk's loads every request of the wavefront waits for a kernarg pointer (+0.4 us), by scalar
        // loads the two dependent scalar-cache misses - pointer, then table - end later than the weights land (+0.2 ... 0.5 us);
        // profiles/r3_rt_branch_free_loads_ab.txt)
        for (int i = tid; i < 256; i += THREADS)
            code2[i] = p.absmax_code[i];
        offset = p.absmax_offset[0];
    }
    BNB_RT_STAMP(2)
    __syncthreads();
    BNB_RT_STAMP(3)

    // Where lane (r, pp) puts its 16 bytes, and where lane (ln, lg) finds those of lane (r = ln, pp = lg). The XOR makes
    // both sides conflict-free under the hardware's lane grouping (guide, LDS table): ds_write_b128 serves 8 CONTIGUOUS lanes
    // per pass over 32 banks - rows 2i, 2i+1 x pieces 0..3 land in 8 different 16-byte positions mod 8 - and ds_read_b128
    // serves the 16-lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... over 64 banks: group 0 = (lg 0: ln in
    // {0-3, 12-15}) + (lg 1: ln in 4..11) reads positions {0-3, 12-15} and {4..11} ^ 2 = {4..11}. (The first version rotated
    // by 2 pp instead - right for contiguous 8-lane read groups, which the hardware does not use: PMC showed 16 % of the LDS
    // cycles as bank conflicts.)
    const int wslot = 16 * pp + (r ^ (2 * pp));
    const int rslot = 16 * lg + (ln ^ (2 * lg));
    const uint32_t lane_off = static_cast<uint32_t>(lane & 31) * 4u + static_cast<uint32_t>(opaque_zero());

    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (; c < ce; c += WAVES) {
        // weights: transpose, then regroup so that dwords 0/1 (2/3) of every lane group belong to block 2h (2h + 1)
        u32x4 wt[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32x4* const tile = h ? tile1 : tile0;
            tile[wslot] = raw.w[h];
            wt[h] = tile[rslot];
        }
        if (pp == 0)
            stile[r] = raw.s;
        if constexpr (BS32) {
            if (pp == 1)
                stile[16 + r] = raw.s2;
        }
        const u32x4 sraw = stile[ln];
        [[maybe_unused]] u32x4 sraw2 = sraw;
        if constexpr (BS32)
            sraw2 = stile[16 + ln];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const auto s02 = __builtin_amdgcn_permlane32_swap(wt[h][0], wt[h][2], false, false);
            const auto s13 = __builtin_amdgcn_permlane32_swap(wt[h][1], wt[h][3], false, false);
            wt[h][0] = s02[0];
            wt[h][2] = s02[1];
            wt[h][1] = s13[0];
            wt[h][3] = s13[1];
            if constexpr (BS32) {
                // stage two of the 4 x 4 transposition: dword j of every lane group <- block j's k 8 g ... of the group
                const auto t01 = __builtin_amdgcn_permlane16_swap(wt[h][0], wt[h][1], false, false);
                const auto t23 = __builtin_amdgcn_permlane16_swap(wt[h][2], wt[h][3], false, false);
                wt[h][0] = t01[0];
                wt[h][1] = t01[1];
                wt[h][2] = t23[0];
                wt[h][3] = t23[1];
            }
        }
        constexpr int NBLK = BS32 ? 8 : 4, PER = BS32 ? 1 : 2; // scaled blocks of a chunk, MFMAs per block
        float scale[NBLK];
        if constexpr (BS32) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t lo = sraw[b], hi = sraw2[b]; // (copies: see below)
                scale[b] = __builtin_bit_cast(float, lo);
                scale[4 + b] = __builtin_bit_cast(float, hi);
            }
        }
#pragma unroll
        for (int b = 0; b < (BS32 ? 0 : 4); ++b) {
            // (copies first: __builtin_bit_cast applied to a vector-element lvalue reads element 0 - hipcc 7.2)
            const uint32_t sb = sraw[b], s1 = sraw[1];
            // (nested, blocksize >= 128: the two bytes fetched for sub-blocks 0 and 2 become the four codes here)
            const uint32_t s0 = BS64 ? sraw[0] : sraw[0] * 0x0101u + sraw[2] * 0x01010000u;
            if constexpr (NESTED)
                scale[b] = nested_scale(code2[(s0 >> (8 * b)) & 0xFFu], __builtin_bit_cast(float, s1), offset);
            else
                scale[b] = __builtin_bit_cast(float, sb);
        }
        if (c == cb + wave)
            BNB_RT_STAMP(4)
        u32x4 bfr[8]; // decoded weight fragments of the chunk: produced with the first activation tile, reused by the others
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int blk = 0; blk < NBLK; ++blk) {
                f32x4 part = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int s = PER * blk + i, h = s >> 2, j = s & 3; // (64-k blocks: steps 2 blk, 2 blk + 1; BS32: step = block)
                    u32x4* const tile = (s & 1) ? tile1 : tile0;
                    u32x4 af;
                    if constexpr (DIRECT) {
                        af = raw.a[s];
                    } else {
                        tile[wslot] = raw.a[s];
                        if (mt + 1 < MT)
                            load_a_step(raw, c, mt + 1, s); // the next tile's piece refills the register just emptied
                        af = tile[rslot];
                    }
                    if (mt == 0) {
                        const uint32_t w = wt[h][j];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const uint32_t byte = __builtin_amdgcn_ubfe(w, 8u * q, 8u);
                            bfr[s][q] = *reinterpret_cast<const __attribute__((address_space(3))) uint32_t*>((byte << 7) + lane_off);
                        }
                    }
                    part = RtMma<T>::run(af, bfr[s], part);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[mt][q] = fmaf(scale[blk], part[q], acc[mt][q]);
            }
        }
        if (c + WAVES < ce)
            issue(raw, c + WAVES);
    }
    BNB_RT_STAMP(5)
    // The table is addressed with raw LDS offsets: it must sit at LDS address 0 (no static LDS in this kernel).
    if (reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)smem) != 0)
        __builtin_trap();

    // ---- combine the wavefronts' partial tiles in wavefront order
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        *reinterpret_cast<f32x4*>(red + ((wave * MT + mt) * 64 + lane) * 16) = acc[mt];
    __syncthreads();
    BNB_RT_STAMP(6)
    for (int o = tid; o < MT * 256; o += THREADS) {
        const int mt = o >> 8, col = o & 15, row = (o >> 4) & 15;
        const int src = (col + 16 * (row >> 2)) * 4 + (row & 3);
        float pv[WAVES];
#pragma unroll
        for (int w = 0; w < WAVES; ++w)
            pv[w] = reinterpret_cast<const float*>(red + (w * MT + mt) * 1024)[src];
        __builtin_amdgcn_sched_barrier(0); // all look-ups in flight before the first add (the adds stay in wavefront order)
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w)
            v += pv[w];
        const int m = m_base + mt * 16 + row, n = col0 + col;
        if (m < M && n < N) {
            const long o2 = static_cast<long>(m) * N + n;
            if (hot_kslices == 1) {
                const T* bias = static_cast<const T*>(p.bias);
                const float b = bias ? static_cast<float>(bias[n]) : 0.0f;
                static_cast<T*>(p.out)[o2] = static_cast<T>(v + b);
            } else {
                p.ws[static_cast<long>(blockIdx.y) * M * N + o2] = v;
            }
        }
    }
    BNB_RT_STAMP(7)
}

int rt_cu_count() { return device_cu_count_or_default(); }

struct RtPlan {
    int ks, cps, waves, mt;
    int bl = 0; // branch-free buffer loads (template parameter BL)
    int direct_max = 4; // batches up to this many rows load their activation fragments directly (see DIRECT)
};

// K slices only when the column tiles alone would leave nearly all of the chip idle (a second launch and a slab round trip
// cost more than a half-empty chip gains: 1376 x 4096 measured 5.3 us un-split against 6.3 us with two slices); every
// wavefront keeps at least one chunk.
RtPlan rt_plan(int M, int N, int K, int force_ks) {
    RtPlan pl;
    const int chunks = K / kRtChunk;
    // Row tiles per workgroup: the whole batch up to 64 rows (a chunk's weights are decoded once for all of them) when the
    // column tiles alone fill the chip; on narrow matrices separate workgroups per row tile fill it better (measured,
    // profiles/r2_mfma_ab.txt: 4096^2 M = 64 11.9 us with 4 tiles per workgroup vs 13.5 with one; 1376 x 4096 10.1 vs 7.8)
    const int cus = rt_cu_count();
    const int col_tiles = (N + 15) / 16;
    pl.mt = M > 48 ? 4 : M > 32 ? 3 : M > 16 ? 2 : 1;
    while (pl.mt > 1 && col_tiles * ((M + 16 * pl.mt - 1) / (16 * pl.mt)) < cus)
        --pl.mt;
    const int wgs = col_tiles * ((M + 16 * pl.mt - 1) / (16 * pl.mt));
    int ks = 1;
    if (force_ks > 0)
        ks = force_ks;
    else if (wgs * 8 <= cus)
        ks = cus / (2 * wgs);
    const int max_ks = chunks / 8 > 0 ? chunks / 8 : 1;
    ks = ks > max_ks ? max_ks : ks;
    ks = ks < 1 ? 1 : ks;
    pl.cps = (chunks + ks - 1) / ks;
    pl.ks = (chunks + pl.cps - 1) / pl.cps;
    // Sixteen wavefronts (one workgroup per CU) only when the launch is a single round of workgroups with long rows;
    // otherwise eight, two workgroups per CU (measured on MI355X, profiles/r2_mfma_ab.txt: 4096^2 6.45 vs 6.7 us,
    // 11008 x 4096 11.2 vs 16.0; 4096 x 11008 11.5 vs 10.8)
    pl.waves = (pl.mt == 1 && wgs * pl.ks <= cus && pl.cps > 16) ? 16 : 8;
    // with directly loaded activation fragments (M <= 4) sixteen wavefronts x one chunk win from 16 chunks on
    // (4096^2 M = 3: 6.07 vs 6.59 us), and - since the early half of the wavefronts builds the table alone - up to M = 8
    // (4096^2 M = 5 / 8: 6.40 / 6.47 vs 6.51 / 6.58 us; M = 16: 6.90 vs 6.78)
    if (M <= 8 && wgs * pl.ks <= cus && pl.cps >= 16)
        pl.waves = 16;
    // direct activation fragments (measured, profiles/r3_rt_direct_fragments_m8_ab.txt, us direct vs transposed): M = 5 / 6 win on
    // every shape (4096^2 5.7 / 5.8 vs 6.4, 11008 x 4096 11.1 / 11.6 vs 12.0, 8192^2 nested 14.1 / 14.8 vs 15.4 / 15.0); M = 8 wins
    // on 4096^2 (6.05 vs 6.46) and loses 0.5 - 0.7 us on the larger matrices; above 8 rows the direct form loses everywhere
    // (4096^2 M = 12 / 16 7.1 / 7.7 vs 6.65 / 6.74, profiles/r3_rt_direct_fragments_m16_ab.txt)
    pl.direct_max = static_cast<long>(N) * K <= (20L << 20) ? 8 : 6;
    return pl;
}

template <typename T, int MT, int WAVES, bool DIRECT, bool BL>
void rt_launch_bl(const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax8, int M, int N, int K, int flags,
                  const RtPlan& pl, const RtArgs& a, hipStream_t stream) {
    const bool bs32 = (flags & 31) == 5;
    const size_t lds = kRtLut + static_cast<size_t>(WAVES) * (bs32 ? kRtScratch32 : kRtScratch) + 1024 + static_cast<size_t>(WAVES) * MT * 1024;
    dim3 grid((N + 15) / 16, pl.ks, (M + 16 * MT - 1) / (16 * MT));
    if constexpr (BL) {
        if (bs32) { // (fp32 absmax only: gemm_4bit_rt_supported; the branch-free form only)
            auto kern = gemm4_mfma_rt_kernel<T, MT, WAVES, false, DIRECT, true, false, true>;
            static LdsLimit lim;
            ensure_dynamic_lds(lim, reinterpret_cast<const void*>(kern), lds);
            hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), lds, stream, A, B, absmax, absmax8, M, N, K, flags, pl.cps, pl.ks, a);
            return;
        }
    }
    if (absmax8 != nullptr && (flags & 31) == 6) {
        auto kern = gemm4_mfma_rt_kernel<T, MT, WAVES, true, DIRECT, BL, true>;
        static LdsLimit lim;
        ensure_dynamic_lds(lim, reinterpret_cast<const void*>(kern), lds);
        hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), lds, stream, A, B, absmax, absmax8, M, N, K, flags, pl.cps, pl.ks, a);
    } else if (absmax8 != nullptr) {
        auto kern = gemm4_mfma_rt_kernel<T, MT, WAVES, true, DIRECT, BL, false>;
        static LdsLimit lim;
        ensure_dynamic_lds(lim, reinterpret_cast<const void*>(kern), lds);
        hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), lds, stream, A, B, absmax, absmax8, M, N, K, flags, pl.cps, pl.ks, a);
    } else {
        auto kern = gemm4_mfma_rt_kernel<T, MT, WAVES, false, DIRECT, BL, false>;
        static LdsLimit lim;
        ensure_dynamic_lds(lim, reinterpret_cast<const void*>(kern), lds);
        hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), lds, stream, A, B, absmax, absmax8, M, N, K, flags, pl.cps, pl.ks, a);
    }
}

template <typename T, int MT, int WAVES, bool DIRECT = false>
void rt_launch(const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax8, int M, int N, int K, int flags,
               const RtPlan& pl, const RtArgs& a, hipStream_t stream) {
#ifdef BNB_PROFILING
    // (round 2's form - pointer loads, exec-masked rows - exists in the measurement build only: the A/B partner of
    // tools/rt_variant_ab.py, run with BNB_MI355X_LIBRARY=.../libbitsandbytes_mi355x_prof.so)
    if (!pl.bl)
        return rt_launch_bl<T, MT, WAVES, DIRECT, false>(A, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
#endif
    return rt_launch_bl<T, MT, WAVES, DIRECT, true>(A, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
}

template <typename T> void rt_launch_mt(const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax8, int M, int N,
                                        int K, int flags, const RtPlan& pl, const RtArgs& a, hipStream_t stream) {
    switch (pl.mt) {
    case 1:
        if (M <= pl.direct_max) { // activation fragments loaded straight in MFMA shape
            if (pl.waves == 16)
                return rt_launch<T, 1, 16, true>(A, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
            return rt_launch<T, 1, 8, true>(A, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
        }
        if (pl.waves == 16)
            return rt_launch<T, 1, 16>(A, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
        return rt_launch<T, 1, 8>(A, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
    case 2: return rt_launch<T, 2, 8>(A, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
    case 3: return rt_launch<T, 3, 8>(A, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
    default: return rt_launch<T, 4, 8>(A, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
    }
}

} // namespace

bool gemm_4bit_rt_supported(int dtype, const void* A, const uint8_t* B, const float* code16, int M, int N, int K, int blocksize) {
    // (blocksize 32 - the BS32 instances - with fp32 absmax only: gemm_4bit_rt_serves)
    return dtype != 0 && code16 == nullptr && M >= 1 && N >= 1 && K >= kRtChunk && (K % kRtChunk) == 0 && blocksize >= 32 && is_pow2(blocksize) &&
           aligned_to(A, 16) && aligned_to(B, 16) &&
           // (byte offsets of the buffer loads are 32-bit and must stay below the descriptors' 2^31 records)
           static_cast<long long>(N) * K < (1LL << 32) && static_cast<long long>(M) * K < (1LL << 30);
}

// statistics the kernel serves at this blocksize: everything from 64 up; at blocksize 32 fp32 absmax on a 16-byte boundary (two
// 16-byte scale loads per row and chunk)
bool gemm_4bit_rt_serves(const float* absmax, const uint8_t* absmax8, int blocksize) {
    return blocksize >= 64 || (absmax8 == nullptr && aligned_to(absmax, 16));
}

size_t gemm_4bit_rt_workspace_bytes(int M, int N, int K, int force_ks) {
    if (M < 1 || N < 1 || K < kRtChunk)
        return 0;
    const RtPlan pl = rt_plan(M, N, K, force_ks);
    return pl.ks > 1 ? static_cast<size_t>(pl.ks) * M * N * sizeof(float) : 0;
}

// dtype: 1 = f16, 2 = bf16. force_ks / force_waves: sweeps and tests (0 = built-in choice).
void gemm_4bit_rt(int dtype, const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax8,
                  const float* absmax_code, const float* absmax_offset, void* out, const void* bias, int M, int N, int K,
                  int blocksize, int quant_type, void* workspace, size_t workspace_bytes, int force_ks, int force_waves,
                  int variant, hipStream_t stream) {
    g_last_gemm_kernel = kKernelRt;
    RtPlan pl = rt_plan(M, N, K, force_ks);
    if (force_waves == 8 || (force_waves == 16 && pl.mt == 1))
        pl.waves = force_waves;
    pl.bl = 1;
#ifdef BNB_PROFILING
    // knob0 bit 0 (measurement build only, tools/rt_variant_ab.py): round 2's form - pointer loads, direct fragments up to 4 rows
    if (variant & 1) {
        pl.bl = 0;
        pl.direct_max = 4;
    }
#else
    (void)variant;
#endif
    float* ws = static_cast<float*>(workspace);
    const size_t slab = static_cast<size_t>(M) * N * sizeof(float);
    if (pl.ks > 1) {
        if (ws == nullptr) {
            ws = gemm_4bit_internal_workspace(slab * pl.ks, stream);
            workspace_bytes = ws ? slab * pl.ks : 0;
        }
        if (workspace_bytes < slab * pl.ks) {
            const int fit = static_cast<int>(workspace_bytes / slab);
            const int chunks = K / kRtChunk;
            const int ks = fit >= 2 ? fit : 1;
            pl.cps = (chunks + ks - 1) / ks;
            pl.ks = (chunks + pl.cps - 1) / pl.cps;
        }
    }
    RtArgs a;
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
        rt_launch_mt<bf16>(A, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
    else
        rt_launch_mt<f16>(A, B, absmax, absmax8, M, N, K, flags, pl, a, stream);
    BNB_CHECK_LAUNCH();
    if (pl.ks > 1)
        gemm_4bit_finalize(dtype, ws, bias, out, M, N, pl.ks, stream);
}

} // namespace bnb
