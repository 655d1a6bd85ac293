// gemm4_mfma_sm.hip — "streaming MFMA" kernel for small decode batches (2 <= M <= 16, bf16 / fp16, K % 256 == 0):
//   out[m, n] = sum_k A[m, k] * code[B[n, k]] * scale[n, k / bs]  (+ bias[n])
//
// Replaces, on MI355X, the small-M window of the reference's fused kernels (csrc/gemm_4bit_simt.cu:109-533, selected by
// bitsandbytes/backends/cuda/ops.py:823-835 for M <= 4, and the first row tile of csrc/gemm_4bit_sm80.cu:127-457 above).
// Round 6: the batched-decode range M = 2 ... 16 cost 21 - 53 % more than M = 1 on the headline shape for < 2 % more bytes
// (4096^2: 4.13 us at M = 1, 5.0 / 5.5 / 5.9 / 6.3 at M = 2 / 4 / 8 / 16; 8192^2 M = 16: 15.4 against 8.6). Two causes,
// both visible in the designs it ran on: the streaming kernel (gemv4_stream.hip) multiplies on the VALU, so every extra
// row is another 16 packed FMAs per weight item of a VALU-bound decode; the register-transposed kernel
// (gemm4_mfma_rt.hip) gives ONE workgroup per 16 output columns, so every workgroup - two per CU on large matrices -
// pulls ALL of A through its L1 (8192^2, M = 16: 2 x 256 KB per CU, 3.4 us at 64 B / clk) behind a prologue of 11 vector
// loads per wavefront. This kernel keeps the streaming kernel's skeleton and lets the matrix pipe do the rows:
//
//  * ONE persistent workgroup per CU owns R = ceil(N / CUs) consecutive output columns (weight rows) in tiles of 16, for the
//    whole of K. Wavefront w owns the 256-k chunks w, w + WAVES, ... of K for ALL of the workgroup's tiles and keeps that
//    chunk's activation fragments in REGISTERS (8 MFMA operands, 32 VGPRs) while it walks the tiles: every byte of A enters
//    the CU exactly once per launch, whatever R is.
//  * A travels by LDS-DMA (buffer_load ... lds) in full lines - instruction i of a chunk covers rows 2 i, 2 i + 1 x 512
//    contiguous bytes - into a staging area private to the wavefront, XOR-swizzled on the SOURCE side so that the
//    fragment reads (ds_read_b128 in MFMA shape: lane (m, kg)) are bank-conflict-free under gfx950's 16-lane read groups:
//    ROWS / 2 vector-memory instructions per chunk (M <= 4: two) instead of eight fragment-shaped loads. The DMAs are
//    spelled in asm and issued FIRST, i.e. they are the oldest entries of the wavefront's in-order queue: the fragments
//    are in registers (and the next chunk's DMA requested) long before the first weight bytes arrive from HBM. The next
//    chunk's fragments wait in the staging area until the wavefront gets there; the hand-off is one counted s_waitcnt.
//  * Weights stream through a two-stage REGISTER ring of branch-free buffer loads exactly as in the register-transposed
//    kernel (lane 4 r + p = 16-byte piece p of row r: a lane quad covers 64 contiguous bytes of one row), are transposed
//    to MFMA shape through a 1-KiB tile private to the wavefront, regrouped by v_permlane32_swap so that every MFMA
//    consumes k from ONE quantization block, decoded by one LDS look-up per packed byte (byte -> two 16-bit code values,
//    table built from literals while the loads fly) and multiplied by v_mfma_f32_16x16x32: one decode serves all rows.
//    An item past the end of a wavefront's list is an out-of-range offset (zeros, nothing fetched): the number of loads in
//    flight is the same on every path, which is what makes the counted waits - the compiler's and the two written here - exact.
//  * The exact fp32 absmax multiplies the fp32 partial tile of each 64-k block (4 v_fma per block and tile), nested
//    statistics are reconstructed in-kernel with the two roundings of the host-side sequence (bnb_common.h nested_scale).
//  * The wavefronts' partial tiles are combined once, through LDS, in FIXED order ((w0 + w1 + w2 + w3) + ... by four
//    threads per output, then two DPP adds): bit-reproducible, no atomics, no second launch, no workspace.
#include "bnb_common.h"

namespace bnb {

#ifdef BNB_PROFILING
extern unsigned long long* g_dbg_buf; // c_api.hip (profiling builds only)
#endif

namespace {

using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using i32x4 = __attribute__((ext_vector_type(4))) int;

template <typename T> struct SmMma;
template <> struct SmMma<bf16> {
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
template <> struct SmMma<f16> {
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

constexpr int kSmLut = 65536;    // 256 entries x 64 lane-private copies x 4 B, at LDS address 0: a table address is ONE v_perm_b32
constexpr int kSmCode2 = 1024;   // nested absmax code (256 floats)
constexpr int kSmScratch = 1536; // per wavefront: one transposition tile (1 KiB) + the scale tile (256 B), padded to 512 B
constexpr int kSmChunk = 256;    // k per item: four 64-k MFMA pairs
constexpr int kSmMaxTiles = 4;   // 16-row tiles of weight rows per workgroup
constexpr int kSmMaxGroup = 8;   // weight matrices per grouped launch

struct SmArgs {
#ifdef BNB_PROFILING
    unsigned long long* dbg;
#endif
    const float* absmax_offset;
    void* out;
    const void* bias;
    // grouped launch (several weight matrices that share the activations, gemm_4bit_sm_grouped): count > 0, and workgroup b works
    // on member i with start[i] <= b < start[i + 1] - its own packed weights, statistics, bias and output, rows of ONE member only
    // (the count travels in hot_geom)
    int start[kSmMaxGroup + 1];
    int gN[kSmMaxGroup];
    const uint8_t* gB[kSmMaxGroup];
    const float* gabsmax[kSmMaxGroup];
    const uint8_t* gabsmax8[kSmMaxGroup];
    const float* gcode2[kSmMaxGroup];
    const float* goffset[kSmMaxGroup];
    void* gout[kSmMaxGroup];
    const void* gbias[kSmMaxGroup];
};

// Time stamps (measurement build): s_memtime values collect in scalar registers and are stored ONCE, at the end of the kernel
// (a store at the point of the stamp sits in the same in-order queue as the counted waits it is meant to observe).
#ifdef BNB_PROFILING
#define BNB_SM_STAMP(i) { ts[i] = __builtin_amdgcn_s_memtime(); }
#else
#define BNB_SM_STAMP(i) {}
#endif

__device__ __forceinline__ float sm_code_literal(int i, bool fp4) {
    // compare/select over literals: no memory access in front of the table
    constexpr float nf4[16] = {BNB_NF4_VALUES};
    constexpr float fp4v[16] = {BNB_FP4_VALUES};
    float v = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j)
        v = (i == j) ? (fp4 ? fp4v[j] : nf4[j]) : v;
    return v;
}

__device__ __forceinline__ i32x4 sm_rsrc(const void* base) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    return i32x4{static_cast<int>(a), static_cast<int>((a >> 32) & 0xFFFFu), 0x7FFFFFFF, 0x00020000};
}
// LDS-DMA, spelled out: the compiler does not know that this is a vector-memory operation, so its own counted waits for the
// weight ring stay counted (with the builtin it treats the counter as out of order: every register wait becomes vmcnt(0)).
// An unseen DMA that is YOUNGER than a load the compiler waits for only makes that wait stricter; the waits for the DMAs
// themselves are written by hand (sm_wait_vm) at the two places that read the staging area.
__device__ __forceinline__ void sm_dma16(i32x4 rs, uint32_t lds, uint32_t voff, uint32_t soff) {
    lds = __builtin_amdgcn_readfirstlane(lds);
    soff = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
}
template <int N> __device__ __forceinline__ void sm_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void sm_wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// swizzle of the activation staging area: piece P (16 B = 8 k) of staged row m lives at piece P ^ swz(m) of the row's 512 B.
// It uses bits 0, 1, 3 of the piece index only: bit 2 is the one that tells lane groups kg and kg ^ 1 - which share a
// ds_read_b128 service group - apart (see the fragment addresses in the kernel).
__device__ __forceinline__ int sm_swz(int m) { return (m & 3) | ((m & 4) << 1); }

// T in {bf16, f16}; ROWS in {4, 8, 16} = activation rows staged per chunk (>= M; rows past M repeat row M - 1 and are never
// stored); WAVES wavefronts per workgroup (16; 8 at ROWS = 16, whose staging area is 8 KiB per wavefront, and at four tiles);
// TT = 16-row tiles of weight rows per workgroup (compile time: the accumulators of a wavefront's TT tiles live in registers and
// the ring stage of an item must be a compile-time index); NESTED: double-quantised statistics; SINGLE: no wavefront has more
// than ONE item (K <= 256 WAVES and one tile - the headline shape): no ring, no refill requests.
// grid = (ceil(N / R), ceil(M / 16)); hot_geom = R | fp4 << 16 | bs_shift << 20 | members of a grouped launch << 25.
template <typename T, int ROWS, int WAVES, int TT, bool NESTED, bool SINGLE, int ORDER = 0, bool GROUPED = false>
__global__ __launch_bounds__(WAVES * 64) void gemm4_mfma_sm_kernel(
    // hot arguments as separate scalars: preloaded into SGPRs by the command processor (14 dwords)
    const void* hot_A, const uint8_t* hot_B, const float* hot_absmax, const uint8_t* hot_absmax8, const float* hot_code2, int hot_M,
    int hot_N, int hot_K, int hot_geom, const SmArgs p) {
    constexpr int THREADS = WAVES * 64;
    // ROWS = 32 (17 ... 32 batch rows): TWO 16-row blocks against every decoded weight fragment, in 128-k chunks - the staging area
    // holds 16 "virtual" rows of 512 bytes = [row m, k 0 .. 127 | row m + 16, k 0 .. 127], so the DMA lane map, the swizzle and the
    // fragment addresses are those of sixteen staged rows of a 256-k chunk: fragment steps 0 - 3 are row block 0, steps 4 - 7 row block 1.
    constexpr int RB = ROWS == 32 ? 2 : 1;        // 16-row blocks of the batch per workgroup pass
    constexpr int SROWS = ROWS == 32 ? 16 : ROWS; // staged (virtual) rows
    constexpr int CKB = ROWS == 32 ? 128 : 256;   // k per chunk
    constexpr int HALVES = CKB / 128;             // 128-k weight loads per item
    constexpr int NBLK = CKB / 64;                // 64-k blocks per item
    constexpr int NA = SROWS / 2;        // DMA instructions per chunk (1 KiB each)
    constexpr int STAGE = SROWS * 512;   // bytes of a wavefront's staging area
    // Summation order = that of SIXTEEN wavefronts, whatever the instance runs on: an 8-wavefront instance of a 4- or 8-row batch
    // (three or four tiles per workgroup) keeps two accumulator sets per tile - chunks c with c % 16 < 8 and >= 8, i.e. the chunk
    // lists of the "virtual" wavefronts w and w + 8 - and the combine step adds sixteen partial tiles in the 16-wavefront order:
    // a row's bits do not depend on how many tiles its workgroup holds (the launch geometry: matrix size, grouped launches, shards).
    // (Sixteen staged rows always run 8 wavefronts: nothing to match.)
    constexpr int V = (ROWS < 16 && WAVES == 8) ? 2 : 1;
    constexpr int REGION = (STAGE + kSmScratch) > V * TT * RB * 1024 ? (STAGE + kSmScratch) : V * TT * RB * 1024;
    // vector-memory loads per ring stage: two weight loads + the lane's scale (fp32 absmax: one dword; nested: its 8-bit code and
    // the second-level absmax)
    constexpr int LPS = HALVES + (NESTED ? 2 : 1);
    constexpr int NS = SINGLE ? 1 : 2;   // ring stages
    static_assert(ROWS == 4 || ROWS == 8 || ROWS == 16 || ROWS == 32, "staged activation rows");
    static_assert(ROWS != 32 || (!SINGLE && WAVES == 8), "32-row instances: ring, 8 wavefronts");
    static_assert(TT >= 1 && TT <= kSmMaxTiles && (!SINGLE || TT == 1), "tiles per workgroup");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef BNB_PROFILING
    unsigned long long ts[12];
#pragma unroll
    for (int i = 0; i < 12; ++i)
        ts[i] = 0;
#endif
    BNB_SM_STAMP(0)
    const int r = lane >> 2, pp = lane & 3;   // weight-load roles: row r of the 16-row tile, 16-byte piece pp of its 64 bytes
    const int ln = lane & 15, lg = lane >> 4; // MFMA roles: row / column ln, k group lg
    const int M = hot_M, K = hot_K;
    int N = hot_N;
    const int R = hot_geom & 0xFFFF;
    const bool fp4 = (hot_geom >> 16) & 1;
    const int bs_shift = (hot_geom >> 20) & 31;
    // the matrix of this workgroup: the preloaded arguments, or - grouped launch - the member whose block range holds blockIdx.x
    // (scalar loads from the kernarg segment: they return while the activation requests below are formed)
    const uint8_t* g_B = hot_B;
    const float* g_absmax = hot_absmax;
    const uint8_t* g_absmax8 = hot_absmax8;
    const float* g_code2 = hot_code2;
    const float* g_offset = p.absmax_offset;
    void* g_out = p.out;
    const void* g_bias = p.bias;
    int block = blockIdx.x;
    // (a compile-time variant: as a run-time branch the member's pointers were loaded from the kernarg segment - speculatively - in
    // front of the first weight request of EVERY launch: +0.2 ... 0.3 us on the single-matrix launch at two and four rows)
    if constexpr (GROUPED) {
        const int gcount = (hot_geom >> 25) & 15;
        int member = 0;
#pragma unroll
        for (int i = 1; i < kSmMaxGroup; ++i)
            member += (i < gcount && static_cast<int>(blockIdx.x) >= p.start[i]) ? 1 : 0;
        block -= p.start[member];
        N = p.gN[member];
        g_B = p.gB[member];
        g_absmax = p.gabsmax[member];
        g_absmax8 = p.gabsmax8[member];
        g_code2 = p.gcode2[member];
        g_offset = p.goffset[member];
        g_out = p.gout[member];
        g_bias = p.gbias[member];
    }
    const int row0 = block * R;
    int row_end = row0 + R;
    row_end = row_end < N ? row_end : N;
    const int m_base = blockIdx.y * (16 * RB);
    const int C = (K + CKB - 1) / CKB; // chunks of a row (K % 64 == 0: the last one may hold fewer 64-k blocks)
    // chunks of this wavefront: wave, wave + WAVES, ... < C
    const int nchunks = (C - wave + WAVES - 1) / WAVES > 0 ? (C - wave + WAVES - 1) / WAVES : 0;
    const int nitems = nchunks * TT;

    // LDS map: table | nested code | per wavefront: staging (512-byte aligned: fragment addresses are formed with XOR) |
    // transposition tile | scale tile. A wavefront that is done parks its partial tiles in its OWN region.
    constexpr int kRegions = kSmLut + kSmCode2;
    static_assert(kRegions % 512 == 0 && REGION % 512 == 0, "staging areas are 512-byte aligned");
    unsigned char* const region = smem + kRegions + wave * REGION;
    u32x4* const tile = reinterpret_cast<u32x4*>(region + STAGE);
    const uint32_t stage_lds = static_cast<uint32_t>(kRegions + wave * REGION);
    const uint32_t stile_lds = stage_lds + static_cast<uint32_t>(STAGE + 1024);

    // ---- activation DMA: instruction i, lane L fills slot 64 i + L of the staging area = piece P' = L & 31 of staged row
    // m = 2 i + (L >> 5), and fetches piece P' ^ swz(m) of batch row m_base + m (rows past the batch: the last row again)
    const i32x4 rs_a = sm_rsrc(hot_A);
    uint32_t a_voff[NA];
    int a_k[NA]; // k of the lane's piece inside the chunk
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = 2 * i + (lane >> 5);
        const int piece = (lane & 31) ^ sm_swz(m);
        const int kp = ROWS == 32 ? (piece & 15) : piece;                 // 16-byte piece of the chunk's k range
        int mr = m_base + m + (ROWS == 32 ? 16 * (piece >> 4) : 0);
        mr = mr < M ? mr : M - 1;
        a_k[i] = 8 * kp;
        a_voff[i] = static_cast<uint32_t>(mr) * static_cast<uint32_t>(K) * 2u + static_cast<uint32_t>(kp << 4);
    }
    constexpr uint32_t kOob = 0xFFFFFFF0u; // beyond num_records
    auto issue_a = [&](int ci) {
        // chunk ci of this wavefront (a wavefront without chunks stages the row's last chunk: nobody reads it). Pieces past the end of
        // a row - the last chunk of a K that is not a multiple of 256 - are out-of-range offsets: nothing is fetched (the MFMA steps
        // that would consume them are skipped, see compute: no value of those slots is ever used).
        int c = wave + ci * WAVES;
        c = c < C ? c : C - 1;
        const int tail = K - c * CKB;
#pragma unroll
        for (int i = 0; i < NA; ++i)
            sm_dma16(rs_a, stage_lds + static_cast<uint32_t>(i * 1024), a_voff[i] | (a_k[i] < tail ? 0u : kOob), static_cast<uint32_t>(c) * static_cast<uint32_t>(CKB * 2));
    };
    if constexpr (NESTED) {
        // the second-level code table (256 floats): ONE 1-KiB DMA by wavefront 0, the oldest entry of its queue - the table's
        // address arrives in a preloaded argument, so nothing waits for the kernarg segment here. Landed behind the wait for the
        // first fragments, published by the barrier behind it.
        if (wave == 0)
            sm_dma16(sm_rsrc(g_code2), static_cast<uint32_t>(kSmLut), static_cast<uint32_t>(lane) * 16u, 0u);
    }
    // ORDER (experiment, single-item instances): 0 = activations first (the DMAs are the oldest entries: fragments in registers
    // before the weights land), 1 = weights first (every wavefront's weight requests go out before anybody's activation requests;
    // the fragments are read behind the weights' arrival)
    static_assert(ORDER == 0 || SINGLE, "weights-first: single-item instances");
    if constexpr (ORDER == 0)
        issue_a(0);

    // ---- weight ring
    struct Stage {
        u32x4 w[2];  // 128 k each: lane (r, pp) holds k [128 h + 32 pp, + 32) of row r of the tile
        uint32_t s;  // lane (r, pp): the fp32 absmax of 64-k sub-block pp of row r's chunk (nested: its 8-bit code)
        uint32_t s2; // nested: the second-level absmax of that block
    };
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(g_B), 0, 0x7FFFFFFF, 0x00020000);
    const auto rs_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g_absmax), 0, 0x7FFFFFFF, 0x00020000);
    [[maybe_unused]] const auto rs_q = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(g_absmax8), 0, 0x7FFFFFFF, 0x00020000);
    // item q of this wavefront = (chunk q / TT of its list, tile q % TT). Every load is branch-free: an item past the end of
    // the list, and a tile row past the end of the workgroup's rows, is an out-of-range offset (zeros, nothing fetched).
    auto issue = [&](Stage& st, int q) {
        const int ci = q / TT, t = q - ci * TT;
        const int c = wave + ci * WAVES;
        const int wrow = row0 + 16 * t + r;
        const uint32_t inval = (q < nitems && wrow < row_end) ? 0u : kOob;
        const uint32_t w_off = static_cast<uint32_t>(wrow) * static_cast<uint32_t>(K >> 1) + static_cast<uint32_t>(pp * 16);
        const uint32_t soff_w = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(c) * static_cast<uint32_t>(CKB / 2));
        const int tail = K - c * CKB; // k left in the row from this chunk on (>= CKB except in the last chunk of a K % CKB != 0 row)
#pragma unroll
        for (int h = 0; h < HALVES; ++h)
            st.w[h] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_off | inval | (128 * h + 32 * pp < tail ? 0u : kOob),
                                                                                       soff_w + static_cast<uint32_t>(h * 64), 0));
        // quantization block of the lane's 64-k sub-block (N K < 2^32: gemm_4bit_sm_supported)
        const int sub = pp % NBLK; // the lane's 64-k block of the item (128-k items: lanes pp = 2, 3 repeat 0, 1)
        const uint32_t blk = (static_cast<uint32_t>(wrow) * static_cast<uint32_t>(K) + static_cast<uint32_t>(c * CKB) + static_cast<uint32_t>(sub * 64)) >> bs_shift;
        const uint32_t inval_s = inval | (64 * sub < tail ? 0u : kOob);
        if constexpr (NESTED) {
            st.s = static_cast<uint32_t>(__builtin_amdgcn_raw_buffer_load_b8(rs_q, blk | inval_s, 0, 0));
            st.s2 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_raw_buffer_load_b32(rs_s, ((blk >> 8) << 2) | inval_s, 0, 0));
        } else {
            st.s = __builtin_bit_cast(uint32_t, __builtin_amdgcn_raw_buffer_load_b32(rs_s, (blk << 2) | inval_s, 0, 0));
            st.s2 = 0;
        }
    };
    Stage st[NS];
    issue(st[0], 0);
    __builtin_amdgcn_sched_barrier(0); // nothing that is not needed for the loads runs before them
    if constexpr (ORDER == 1)
        issue_a(0);
    BNB_SM_STAMP(1)

    // ---- decode table, built while the loads fly: entry e (a packed byte) = 64 copies of (T(code[e >> 4]), T(code[e & 15])),
    // 256 B per entry; the two halves of the builders write 8 of its 16 16-byte chunks each, in an order rotated by e (the eight
    // lanes one ds_write_b128 services together land in eight bank quads)
    {
        // (the wavefronts of a 16-wavefront workgroup start ~90 cycles apart: the early half builds the table)
        constexpr int BUILD_THREADS = 512;
        if (tid < BUILD_THREADS) {
            const float cv = sm_code_literal((lane & 15) + opaque_zero(), fp4);
            const int cvb = __builtin_bit_cast(int, cv);
            const int e = tid & 255;
            const float hi = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((e >> 4) & 15) * 4, cvb));
            const float lov = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((e & 15) * 4, cvb));
            const uint32_t pr = SmMma<T>::pack(hi, lov);
            const u32x4 v = {pr, pr, pr, pr};
            u32x4* const dst = reinterpret_cast<u32x4*>(smem + e * 256);
            const int half = tid >> 8;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                dst[(8 * half + j + e) & 15] = v;
        }
    }
    BNB_SM_STAMP(2)
    float offset = 0.0f;
    if constexpr (NESTED) {
        // a scalar load (constant address space): a vector load here would sit in the counted queue behind the ring
        typedef const __attribute__((address_space(4))) float* cfloat_ptr;
        int ob = __builtin_bit_cast(int, *(cfloat_ptr)(reinterpret_cast<uintptr_t>(g_offset)));
        asm volatile("" : "+s"(ob));
        offset = __builtin_bit_cast(float, ob);
    }

    // ---- the first chunk's fragments: its DMAs are older than the ring stages
    // fragment of step s = 4 h + 2 a + b, lane (m = ln, kg = lg): 8 k from 128 h + 64 a + 8 b + 32 (kg & 1) + 16 (kg >> 1) of row m,
    // i.e. piece 16 h + 8 a + b + 4 (kg & 1) + 2 (kg >> 1) - the k order the weight regrouping below produces. Lanes of rows past
    // ROWS read a staged row again (their MFMA rows are never stored).
    const int mrow = ln & (SROWS - 1);
    const uint32_t frag_base = stage_lds + static_cast<uint32_t>(mrow * 512) +
                               static_cast<uint32_t>(((4 * (lg & 1) + 2 * (lg >> 1)) ^ sm_swz(mrow)) << 4);
    u32x4 af[8];
    auto read_frags = [&]() {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const uint32_t sbits = static_cast<uint32_t>(256 * (s >> 2) + 128 * ((s >> 1) & 1) + 16 * (s & 1));
            af[s] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(frag_base ^ sbits);
        }
    };
    // The table barrier waits for nobody's memory: a wavefront whose activation DMA sits behind other wavefronts' weight requests
    // in the CU's in-order address pipeline (8192^2: the last wavefront's requests went out 7000 cycles into the kernel, and the
    // barrier - then behind the wait for the fragments - held every wavefront until then) delays only itself. The ring's second
    // stage is requested behind the barrier: 32 KiB of weight requests per CU up front fill the memory pipeline, 64 KiB only
    // congest the address pipeline in front of the late wavefronts' first requests.
    __syncthreads();
    BNB_SM_STAMP(5)
    if constexpr (!SINGLE) {
        issue(st[1], 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (ORDER == 1)
        sm_wait_vm<0>();
    else
        sm_wait_vm<NS * LPS>();
    BNB_SM_STAMP(3)
    read_frags();
    if constexpr (!SINGLE) {
        if (nchunks > 1) {
            sm_wait_lgkm0(); // the fragment reads have returned before the staging area is overwritten
            issue_a(1);
        }
    }
    BNB_SM_STAMP(4)
#ifdef BNB_PROFILING
    sm_wait_vm<(NS - 1) * LPS + (SINGLE ? 0 : NA)>(); // (measurement build: when the first ring stage has landed; exact with >= 2 chunks)
    BNB_SM_STAMP(6)
#endif

    // Where lane (r, pp) puts its 16 bytes, and where lane (ln, lg) finds those of lane (r = ln, pp = lg): conflict-free both
    // ways under the hardware's lane grouping (gemm4_mfma_rt.hip).
    const int wslot = 16 * pp + (r ^ (2 * pp));
    const int rslot = 16 * lg + (ln ^ (2 * lg));
    const uint32_t lane4 = static_cast<uint32_t>(lane) * 4u;
    const uint32_t perm_sel = 0x0C0C0400u + static_cast<uint32_t>(opaque_zero()); // {lane offset, weight byte, 0, 0}

    f32x4 acc[V][TT][RB];
#pragma unroll
    for (int z = 0; z < V; ++z)
#pragma unroll
        for (int t = 0; t < TT; ++t)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                acc[z][t][rb] = f32x4{0.f, 0.f, 0.f, 0.f};

    // nb = 64-k blocks of the item's chunk (4; fewer in the last chunk of a K % 256 != 0 row: the MFMA pairs of the missing blocks are
    // skipped - wave-uniform - so neither the staging slots nor the weight registers behind the end of the row are ever multiplied)
    auto compute = [&](const Stage& s, f32x4 (&accv)[RB], int nb) {
        // the lane's scale: lane (r, pp) = 4 r + pp writes dword pp of row r's 16 bytes, lane (ln, lg) reads row ln's four
        float sc;
        if constexpr (NESTED)
            sc = nested_scale(*reinterpret_cast<const __attribute__((address_space(3))) float*>(static_cast<uint32_t>(kSmLut) + s.s * 4u),
                              __builtin_bit_cast(float, s.s2), offset);
        else
            sc = __builtin_bit_cast(float, s.s);
        *reinterpret_cast<__attribute__((address_space(3))) float*>(stile_lds + lane4) = sc;
        // weights: transpose (one tile, the two halves in turn: same wavefront, in-order LDS), then regroup so that dwords 0/1
        // (2/3) of every lane group belong to block 2h (2h + 1)
        u32x4 wt[HALVES];
        tile[wslot] = s.w[0];
        wt[0] = tile[rslot];
        if constexpr (HALVES == 2) {
            tile[wslot] = s.w[1];
            wt[1] = tile[rslot];
        }
        const f32x4 scale = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(stile_lds + static_cast<uint32_t>(ln * 16));
#pragma unroll
        for (int h = 0; h < HALVES; ++h) {
            const auto s02 = __builtin_amdgcn_permlane32_swap(wt[h][0], wt[h][2], false, false);
            const auto s13 = __builtin_amdgcn_permlane32_swap(wt[h][1], wt[h][3], false, false);
            wt[h][0] = s02[0];
            wt[h][2] = s02[1];
            wt[h][1] = s13[0];
            wt[h][3] = s13[1];
        }
        // all look-ups first (one v_perm_b32 + one ds_read_b32 per packed byte), then the MFMAs: written look-up by look-up, every
        // MFMA waited for its own four LDS round trips (2100 cycles per item with four wavefronts per SIMD). With several tiles'
        // accumulators in 128 registers: half a chunk (16 look-ups, four MFMAs) at a time.
        constexpr int STEPS = 4 * HALVES;                                              // MFMA k steps (32 k) of an item's weights
        constexpr int GROUP = (ROWS == 32) ? 4 : (TT == 1 || WAVES == 8) ? 8 : 4; // MFMA steps decoded together
        u32x4 bf[GROUP];
#pragma unroll
        for (int g0 = 0; g0 < STEPS; g0 += GROUP) {
#pragma unroll
            for (int sidx = g0; sidx < g0 + GROUP; ++sidx) {
                const uint32_t w = wt[sidx >> 2][sidx & 3];
#pragma unroll
                for (int qq = 0; qq < 4; ++qq)
                    bf[sidx - g0][qq] = *reinterpret_cast<const __attribute__((address_space(3))) uint32_t*>(__builtin_amdgcn_perm(w, lane4, perm_sel + (qq << 8)));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int blk = g0 / 2; blk < (g0 + GROUP) / 2; ++blk) {
                if (blk >= nb)
                    continue;
                const float sb = scale[blk];
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    // (32 rows: fragment steps 4 rb + 2 blk, + 1 - the row block's half of the virtual rows)
                    f32x4 part = {0.f, 0.f, 0.f, 0.f};
                    part = SmMma<T>::run(af[4 * rb * (RB - 1) + 2 * blk], bf[2 * blk - g0], part);
                    part = SmMma<T>::run(af[4 * rb * (RB - 1) + 2 * blk + 1], bf[2 * blk + 1 - g0], part);
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq)
                        accv[rb][qq] = fmaf(sb, part[qq], accv[rb][qq]);
                }
            }
            if (g0 + GROUP < STEPS)
                __builtin_amdgcn_sched_barrier(0);
        }
    };

    if constexpr (SINGLE) {
        if (nitems > 0) {
            const int left = (K >> 6) - 4 * wave;
            compute(st[0], acc[0][0], left < 4 ? left : 4);
        }
        BNB_SM_STAMP(7)
    } else {
        // ---- items, UNROLL at a time: the ring stage (q & 1) and the tile (q % TT) of an item are compile-time values. Every item
        // is followed by the request of item q + 2 into the stage just emptied - valid or not (see issue): LPS loads are in flight
        // behind the stage about to be consumed at every point of the loop.
        constexpr int UNROLL = (V == 2 || (TT % 2)) ? 2 * TT : TT; // (a multiple of V TT: the accumulator set of an item is a compile-time index)
        for (int q0 = 0; q0 < nitems; q0 += UNROLL) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int q = q0 + u;
                const int t = u % TT;
                if (t == 0 && q > 0 && q < nitems) {
                    // first item of the wavefront's next chunk: its fragments were requested when the previous chunk's were read,
                    // behind at most the ring stage about to be consumed and in front of the other one
                    sm_wait_vm<LPS>();
                    read_frags();
                    if (q / TT + 1 < nchunks) {
                        sm_wait_lgkm0();
                        issue_a(q / TT + 1);
                    }
                }
                if (q < nitems) {
                    const int left = (K >> 6) - NBLK * (wave + (q / TT) * WAVES);
                    compute(st[u & 1], acc[(u / TT) % V][t], left < NBLK ? left : NBLK);
                }
                if (q == 0)
                    BNB_SM_STAMP(7)
                issue(st[u & 1], q + 2);
            }
        }
    }
    // The table is addressed with raw LDS offsets: it must sit at LDS address 0 (no static LDS in this kernel).
    if (reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)smem) != 0)
        __builtin_trap();

    // ---- combine the wavefronts' partial tiles in a fixed order. A wavefront parks its tiles in its own region (its last
    // DMA was waited for at its last chunk switch; the ring's dummy requests do not write LDS).
    BNB_SM_STAMP(8)
#pragma unroll
    for (int z = 0; z < V; ++z)
#pragma unroll
        for (int t = 0; t < TT; ++t)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                *reinterpret_cast<f32x4*>(region + ((z * TT + t) * RB + rb) * 1024 + lane * 16) = acc[z][t][rb];
    __syncthreads();
    BNB_SM_STAMP(9)
    constexpr int PARTS = 4, WPP = WAVES * V / PARTS; // (virtual wavefront v = set v / WAVES of wavefront v % WAVES)
    const int nout = TT * 16 * ROWS;
    for (int idx = tid; idx < nout * PARTS; idx += THREADS) {
        const int o = idx >> 2, part = idx & 3;
        const int col = o & 15, m = (o >> 4) & (ROWS - 1), t = o / (16 * ROWS);
        const int rb = m >> 4, mm16 = m & 15; // (row block of the batch row - 32-row instances -, row inside it)
        const int src = (col + 16 * (mm16 >> 2)) * 4 + (mm16 & 3);
        float pv[WPP];
#pragma unroll
        for (int w = 0; w < WPP; ++w)
            pv[w] = reinterpret_cast<const float*>(smem + kRegions + ((part * WPP + w) % WAVES) * REGION + ((((part * WPP + w) / WAVES) * TT + t) * RB + rb) * 1024)[src];
        float v = pv[0];
#pragma unroll
        for (int w = 1; w < WPP; ++w)
            v += pv[w];
        // (a + b is the same value in both lanes of a pair: the four partial sums combine as (p0 + p1) + (p2 + p3) everywhere)
        v = dpp_add<0xB1>(v);
        v = dpp_add<0x4E>(v);
        const int mm = m_base + m, n = row0 + 16 * t + col;
        if (part == 0 && mm < M && n < row_end) {
            const T* bias = static_cast<const T*>(g_bias);
            const float b = bias ? static_cast<float>(bias[n]) : 0.0f;
            static_cast<T*>(g_out)[static_cast<long>(mm) * N + n] = static_cast<T>(v + b);
        }
    }
#ifdef BNB_PROFILING
    BNB_SM_STAMP(10)
    if (p.dbg && lane == 0) {
        unsigned long long* const d = p.dbg + ((static_cast<long>(blockIdx.y) * gridDim.x + blockIdx.x) * WAVES + wave) * 16;
#pragma unroll
        for (int i = 0; i < 11; ++i)
            d[i] = ts[i];
    }
#endif
}

int sm_cu_count() { return device_cu_count_or_default(); }

struct SmPlan {
    int R, tt, grid_x, rows;
    int variant = 0; // experiment bits (bnb_mi355x_set_tuning knob0)
    bool grouped = false;
};

// rows per workgroup: one workgroup per CU when 64 rows are enough, else whole rounds of workgroups
SmPlan sm_plan(int M, int N) {
    SmPlan pl;
    const int cus = sm_cu_count();
    const long per_round = static_cast<long>(cus) * 16 * kSmMaxTiles;
    const int rounds = static_cast<int>((N + per_round - 1) / per_round);
    int R = (N + cus * rounds - 1) / (cus * rounds);
    R = R < 16 ? 16 : R;
    pl.R = R;
    pl.tt = (R + 15) / 16;
    pl.grid_x = (N + R - 1) / R;
    // Above 16 rows: TWO row blocks per decoded fragment in row passes of 32 (ROWS = 32) once passes of 16 would not all find a free CU;
    // while they do (small matrices: 1376 x 4096 at 32 rows 5.4 us against 7.3), passes of 16 side by side (profiles/r6_sm_rows32_ab.txt)
    pl.rows = M <= 4 ? 4 : M <= 8 ? 8 : (M <= 16 || static_cast<long>(pl.grid_x) * ((M + 15) / 16) <= cus) ? 16 : 32;
    return pl;
}

template <typename T, int ROWS, int WAVES, int TT, bool NESTED, bool SINGLE, int ORDER = 0, bool GROUPED = false>
void sm_launch_one(const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax8, const float* code2, int M, int N, int K, int geom,
                   const SmPlan& pl, const SmArgs& a, hipStream_t stream) {
    constexpr int V = (ROWS < 16 && WAVES == 8) ? 2 : 1; // (accumulator sets per tile: the kernel's V)
    constexpr int RB = ROWS == 32 ? 2 : 1, SROWS = ROWS == 32 ? 16 : ROWS;
    constexpr size_t region = (SROWS * 512 + kSmScratch) > V * TT * RB * 1024 ? (SROWS * 512 + kSmScratch) : V * TT * RB * 1024;
    constexpr size_t lds = kSmLut + kSmCode2 + static_cast<size_t>(WAVES) * region;
    auto kern = gemm4_mfma_sm_kernel<T, ROWS, WAVES, TT, NESTED, SINGLE, ORDER, GROUPED>;
    static LdsLimit lim;
    ensure_dynamic_lds(lim, reinterpret_cast<const void*>(kern), lds);
    hipLaunchKernelGGL(kern, dim3(pl.grid_x, (M + 16 * RB - 1) / (16 * RB)), dim3(WAVES * 64), lds, stream, A, B, absmax, absmax8, code2, M, N, K, geom, a);
}

template <typename T, int ROWS, int WAVES, int TT>
void sm_launch_kind(const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax8, const float* code2, int M, int N, int K, int geom,
                    const SmPlan& pl, const SmArgs& a, hipStream_t stream) {
    const bool nested = absmax8 != nullptr;
    if (pl.grouped) { // (ring instances only)
        if (nested)
            return sm_launch_one<T, ROWS, WAVES, TT, true, false, 0, true>(A, B, absmax, absmax8, code2, M, N, K, geom, pl, a, stream);
        return sm_launch_one<T, ROWS, WAVES, TT, false, false, 0, true>(A, B, absmax, absmax8, code2, M, N, K, geom, pl, a, stream);
    }
    if constexpr (TT == 1 && ROWS != 32) {
        if ((K + kSmChunk - 1) / kSmChunk <= WAVES) { // one item per wavefront at most: no ring
            if (nested)
                return sm_launch_one<T, ROWS, WAVES, 1, true, true>(A, B, absmax, absmax8, code2, M, N, K, geom, pl, a, stream);
            if (pl.variant & 4)
                return sm_launch_one<T, ROWS, WAVES, 1, false, true, 1>(A, B, absmax, absmax8, code2, M, N, K, geom, pl, a, stream);
            return sm_launch_one<T, ROWS, WAVES, 1, false, true>(A, B, absmax, absmax8, code2, M, N, K, geom, pl, a, stream);
        }
    }
    if (nested)
        return sm_launch_one<T, ROWS, WAVES, TT, true, false>(A, B, absmax, absmax8, code2, M, N, K, geom, pl, a, stream);
    return sm_launch_one<T, ROWS, WAVES, TT, false, false>(A, B, absmax, absmax8, code2, M, N, K, geom, pl, a, stream);
}

// 16 wavefronts per workgroup, 8 where 128 registers (three or four tiles' accumulators and item bodies beside the fragments and the ring) or the LDS
// (8-KiB staging areas at ROWS = 16) do not allow them
template <typename T, int ROWS>
void sm_launch_tt(const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax8, const float* code2, int M, int N, int K, int geom,
                  const SmPlan& pl, const SmArgs& a, hipStream_t stream) {
    constexpr int W = ROWS >= 16 ? 8 : 16;
    switch (pl.tt) {
    case 1: return sm_launch_kind<T, ROWS, W, 1>(A, B, absmax, absmax8, code2, M, N, K, geom, pl, a, stream);
    case 2: return sm_launch_kind<T, ROWS, W, 2>(A, B, absmax, absmax8, code2, M, N, K, geom, pl, a, stream);
    case 3: return sm_launch_kind<T, ROWS, 8, 3>(A, B, absmax, absmax8, code2, M, N, K, geom, pl, a, stream);
    default: return sm_launch_kind<T, ROWS, 8, 4>(A, B, absmax, absmax8, code2, M, N, K, geom, pl, a, stream);
    }
}

template <typename T>
void sm_launch_rows(const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax8, const float* code2, int M, int N, int K, int geom,
                    const SmPlan& pl, const SmArgs& a, hipStream_t stream) {
    switch (pl.rows) {
    case 4: return sm_launch_tt<T, 4>(A, B, absmax, absmax8, code2, M, N, K, geom, pl, a, stream);
    case 8: return sm_launch_tt<T, 8>(A, B, absmax, absmax8, code2, M, N, K, geom, pl, a, stream);
    case 16: return sm_launch_tt<T, 16>(A, B, absmax, absmax8, code2, M, N, K, geom, pl, a, stream);
    default: return sm_launch_tt<T, 32>(A, B, absmax, absmax8, code2, M, N, K, geom, pl, a, stream);
    }
}

} // namespace

// Preconditions of the kernel (the statistics' alignment is checked by gemm_4bit_sm_serves): 16-bit activations, literal code
// table, K a multiple of 64 (the last 256-k chunk of a row may be partial; the reference's fused kernels take any K % blocksize
// == 0, csrc/gemm_4bit_simt.cu:208,225), blocksize >= 64, 16-byte aligned A and B, 32-bit byte offsets below the descriptors' records.
bool gemm_4bit_sm_supported(int dtype, const void* A, const uint8_t* B, const float* code16, int M, int N, int K, int blocksize) {
    return dtype != 0 && code16 == nullptr && M >= 1 && N >= 1 && K >= 64 && (K % 64) == 0 && blocksize >= 64 && is_pow2(blocksize) &&
           aligned_to(A, 16) && aligned_to(B, 16) && static_cast<long long>(N) * K < (1LL << 32) && static_cast<long long>(M) * K < (1LL << 30);
}

// (statistics travel as one dword / one byte per lane: no alignment beyond the element's)
bool gemm_4bit_sm_serves(const float* absmax, const uint8_t* absmax8, int blocksize) {
    (void)absmax8;
    (void)blocksize;
    return aligned_to(absmax, 4);
}

// dtype: 1 = f16, 2 = bf16. Any M (row passes of 16 over grid.y); meant for M <= 16.
void gemm_4bit_sm(int dtype, const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax8,
                  const float* absmax_code, const float* absmax_offset, void* out, const void* bias, int M, int N, int K,
                  int blocksize, int quant_type, int variant, hipStream_t stream) {
    g_last_gemm_kernel = kKernelSm;
    SmPlan pl = sm_plan(M, N);
    pl.variant = variant;
    if ((variant & 16) && pl.rows == 32) // (A/B: row passes of 16, as before the 32-row instances)
        pl.rows = 16;
    SmArgs a{};
#ifdef BNB_PROFILING
    a.dbg = g_dbg_buf;
#endif
    a.absmax_offset = absmax_offset;
    a.out = out;
    a.bias = bias;
    const int geom = pl.R | ((quant_type == kFP4) ? (1 << 16) : 0) | (ilog2(blocksize) << 20);
    if (dtype == 2)
        sm_launch_rows<bf16>(A, B, absmax, absmax8, absmax_code, M, N, K, geom, pl, a, stream);
    else
        sm_launch_rows<f16>(A, B, absmax, absmax8, absmax_code, M, N, K, geom, pl, a, stream);
    BNB_CHECK_LAUNCH();
}

// Several weight matrices that share the activations (Q/K/V, gate/up; reference: one gemm_4bit per Linear4bit, nn/modules.py:609-637)
// in ONE launch: the workgroups are dealt to the members in proportion to their rows, a workgroup works on one member only, and
// every output row is computed exactly as the single-matrix launch computes it (the kernel's summation order does not depend on the
// number of tiles per workgroup). false = not a shape this launch takes (the caller issues the members one by one); nothing was
// launched then. All members share K, blocksize, quant_type and the kind of statistics.
bool gemm_4bit_sm_grouped(int dtype, const void* A, int count, const uint8_t* const* B, const float* const* absmax,
                          const uint8_t* const* absmax8, const float* const* absmax_code, const float* const* absmax_offset,
                          void* const* out, const void* const* bias, const int* N, int M, int K, int blocksize, int quant_type,
                          hipStream_t stream) {
    if (count < 1 || count > kSmMaxGroup || M < 1)
        return false;
    const bool nested = absmax8 != nullptr && absmax8[0] != nullptr;
    long total = 0;
    for (int i = 0; i < count; ++i) {
        if ((absmax8 != nullptr && absmax8[i] != nullptr) != nested || !gemm_4bit_sm_supported(dtype, A, B[i], nullptr, M, N[i], K, blocksize) ||
            !gemm_4bit_sm_serves(absmax[i], nested ? absmax8[i] : nullptr, blocksize))
            return false;
        if (nested && (absmax_code == nullptr || absmax_code[i] == nullptr || absmax_offset == nullptr || absmax_offset[i] == nullptr))
            return false;
        total += N[i];
    }
    if (total >= (1L << 30))
        return false;
    SmPlan pl = sm_plan(M, static_cast<int>(total));
    // rows per workgroup: every member rounds its own share up - not more workgroups than the plan's whole rounds of the chip
    const long limit = static_cast<long>(pl.grid_x) > sm_cu_count() ? static_cast<long>(pl.grid_x) : sm_cu_count();
    SmArgs a{};
    for (;; ++pl.R) {
        long blocks = 0;
        for (int i = 0; i < count; ++i)
            blocks += (N[i] + pl.R - 1) / pl.R;
        if (blocks <= limit || pl.R >= 16 * kSmMaxTiles) {
            pl.grid_x = static_cast<int>(blocks);
            break;
        }
    }
    pl.tt = (pl.R + 15) / 16;
    pl.grouped = true;
#ifdef BNB_PROFILING
    a.dbg = g_dbg_buf;
#endif
    a.absmax_offset = nested ? absmax_offset[0] : nullptr;
    a.out = out[0];
    a.bias = bias ? bias[0] : nullptr;
    a.start[0] = 0;
    for (int i = 0; i < kSmMaxGroup; ++i) {
        const int j = i < count ? i : 0;
        a.start[i + 1] = a.start[i] + (i < count ? (N[i] + pl.R - 1) / pl.R : 0);
        a.gN[i] = N[j];
        a.gB[i] = B[j];
        a.gabsmax[i] = absmax[j];
        a.gabsmax8[i] = nested ? absmax8[j] : nullptr;
        a.gcode2[i] = nested ? absmax_code[j] : nullptr;
        a.goffset[i] = nested ? absmax_offset[j] : nullptr;
        a.gout[i] = out[j];
        a.gbias[i] = bias ? bias[j] : nullptr;
    }
    g_last_gemm_kernel = kKernelSm;
    const int geom = pl.R | ((quant_type == kFP4) ? (1 << 16) : 0) | (ilog2(blocksize) << 20) | (count << 25);
    if (dtype == 2)
        sm_launch_rows<bf16>(A, B[0], absmax[0], a.gabsmax8[0], a.gcode2[0], M, N[0], K, geom, pl, a, stream);
    else
        sm_launch_rows<f16>(A, B[0], absmax[0], a.gabsmax8[0], a.gcode2[0], M, N[0], K, geom, pl, a, stream);
    BNB_CHECK_LAUNCH();
    return true;
}

} // namespace bnb
