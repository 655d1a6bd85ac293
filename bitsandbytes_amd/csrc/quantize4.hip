// quantize4.hip — blockwise NF4/FP4 quantization for gfx950 (kernel behind cquantize_blockwise_<T>_{nf4,fp4}).
//
// Semantics are those of the reference CPU backend's quantize_4bit
// (reference bitsandbytes/backends/default/ops.py:233-259), reproduced bit-for-bit:
//   full blocks : absmax = max|x|  (stored as is; 0 for an all-zero block)
//                 s = clamp(x * (1 / max(absmax, 1e-38)), -1, 1)
//   tail block  : am = max(max|x|, 1e-38) (stored clamped);  s = clamp(x / am, -1, 1)
//   code        : position = #{fp32 midpoints of the *sorted* code table  <  s}   (torch.bucketize, right=False)
//                 NF4: nibble = position; FP4: nibble = argsort-order[position]
//   packing     : element 2i -> high nibble, 2i+1 -> low nibble; odd n pads with the code of s = 0
// (The reference's CUDA kernels use hand-typed FP4 thresholds and send 0*inf to code 0,
//  reference csrc/kernels.cu:64-153; the CPU oracle rule above is the parity target.)
//
// Mapping to the machine: the op is a pure HBM stream (reads n*sizeof(T), writes n/2 + 4n/bs), so the
// job of the kernel is to keep the per-element VALU cost far below the stream time (a wave64 VALU op
// costs 4 cycles on a 16-lane SIMD; a compare tree is ~25 ops/element and was the measured bound).
//   * one 256-thread workgroup owns a tile of CH x 2048 contiguous elements; every lane pulls 8
//     consecutive elements per chunk with one (16-bit types) or two (fp32) 16-byte loads, all CH chunks
//     in flight before the first use, so a wavefront reads 1-2 KiB contiguous per instruction;
//   * the block max is taken on the raw |bit patterns| (unsigned max: packed 16-bit for fp16/bf16),
//     which orders like the float max AND carries NaN/inf upwards, so one compare per block tells the
//     common finite case from the special one; lanes of a block combine by xor-shuffles (+ one LDS
//     hop when a block spans wavefronts);
//   * finite case: the code of s = x * (1/absmax) comes from a *cell table* in LDS instead of a
//     compare tree. [-1, 1] is cut into 2*S+1 uniform cells (S = 16 for NF4, 256 for FP4 - the coarsest powers of two with no
//     two decision bounds in one cell, static_assert-ed: FP4's tightest bounds are -1/384, 0 (between its two zeros) and
//     +1/384, which land in cells S - 1, S, S + 1 from S = 256 on; rounds 1 - 4 carried 512, i.e. an 8-KB table per workgroup
//     where 4 KB do); cell(s) = round(s*S + S) falls out of ONE fma against the
//     2^23 magic constant (exact, single rounding), the cell stores (bound inside it or +inf, code
//     below | code above << 16), and the code is `below + (s > bound)`: fma, shift-add, ds_read_b64,
//     compare, add - 5 VALU ops + 1 LDS read per element, bit-identical to counting the 15 bounds
//     (cell() of s and of every bound use the same exactly-rounded expression, see make_cells);
//   * special case (inf/NaN in the block) and the division of the ragged tail block keep the plain
//     compare tree - they are wave-uniform branches that normal data never takes;
//   * each lane emits one packed dword per chunk (a wavefront writes 256 B contiguous).
#include "bnb_common.h"

namespace bnb {

namespace {

// Sorted code tables and their fp32 midpoints, built at compile time in IEEE fp32.
struct Bounds {
    float b[15];
};

constexpr Bounds make_bounds(const float (&sorted)[16]) {
    Bounds r{};
    for (int i = 0; i < 15; ++i)
        r.b[i] = (sorted[i] + sorted[i + 1]) / 2;
    return r;
}

constexpr float kNF4Sorted[16] = {BNB_NF4_VALUES};
constexpr float kFP4Unsorted[16] = {BNB_FP4_VALUES};
// ascending order of the FP4 table as torch.argsort yields it (+0 = nibble 0 before the second zero = nibble 8)
constexpr int kFP4Order[16] = {11, 10, 13, 12, 15, 14, 9, 0, 8, 1, 6, 7, 4, 5, 2, 3};
constexpr float kFP4Sorted[16] = {
    kFP4Unsorted[11], kFP4Unsorted[10], kFP4Unsorted[13], kFP4Unsorted[12], kFP4Unsorted[15], kFP4Unsorted[14],
    kFP4Unsorted[9],  kFP4Unsorted[0],  kFP4Unsorted[8],  kFP4Unsorted[1],  kFP4Unsorted[6],  kFP4Unsorted[7],
    kFP4Unsorted[4],  kFP4Unsorted[5],  kFP4Unsorted[2],  kFP4Unsorted[3]};

constexpr Bounds kNF4Bounds = make_bounds(kNF4Sorted);
constexpr Bounds kFP4Bounds = make_bounds(kFP4Sorted);

constexpr int nibble_of(int qt, int pos) { return qt == kNF4 ? pos : kFP4Order[pos]; }

// ---- cell table -----------------------------------------------------------------------------------
struct QCell {
    float bound;   // the decision bound lying in this cell, +inf if none
    uint32_t code; // nibble for s <= bound | nibble for s > bound << 16
};

template <int QT> struct CellGrid {
    static constexpr int S = (QT == kNF4) ? 16 : 256; // cells per unit; power of two => s*S exact
    static constexpr int N = 2 * S + 1;
};

// round-half-even of b*S + S, evaluated exactly (b is an fp32 value, S a power of two: the sum is exact
// in double). This is what the device computes with fmaf(s, S, S + 2^23): one rounding of the exact sum
// to a float whose ulp is 1.
constexpr int cell_of(float b, int S) {
    const double t = static_cast<double>(b) * S + S;
    const long fl = static_cast<long>(t);
    const double fr = t - static_cast<double>(fl);
    if (fr > 0.5)
        return static_cast<int>(fl + 1);
    if (fr < 0.5)
        return static_cast<int>(fl);
    return static_cast<int>((fl & 1) ? fl + 1 : fl);
}

template <int QT> struct CellTable {
    QCell c[CellGrid<QT>::N];
};

// Cell i holds at most one bound (checked below). For s in cell i every bound of a lower cell is < s and
// every bound of a higher cell is >= s (s == bound would put both in the same cell, since both go
// through the same exactly-rounded expression), so #{bounds < s} = #{bounds in lower cells} + (s > bound_i).
template <int QT> constexpr CellTable<QT> make_cells() {
    constexpr int S = CellGrid<QT>::S;
    constexpr int N = CellGrid<QT>::N;
    const Bounds B = (QT == kNF4) ? kNF4Bounds : kFP4Bounds;
    CellTable<QT> t{};
    int next = 0; // bounds are ascending: index of the first bound not yet below the current cell
    for (int i = 0; i < N; ++i) {
        t.c[i].bound = __builtin_huge_valf();
        t.c[i].code = static_cast<uint32_t>(nibble_of(QT, next)) | (static_cast<uint32_t>(nibble_of(QT, next)) << 16);
        if (next < 15 && cell_of(B.b[next], S) == i) {
            t.c[i].bound = B.b[next];
            t.c[i].code = static_cast<uint32_t>(nibble_of(QT, next)) |
                          (static_cast<uint32_t>(nibble_of(QT, next + 1)) << 16);
            ++next;
        }
    }
    return t;
}

template <int QT> constexpr bool cells_are_valid() {
    constexpr int S = CellGrid<QT>::S;
    const Bounds B = (QT == kNF4) ? kNF4Bounds : kFP4Bounds;
    for (int i = 0; i < 15; ++i) {
        const int c = cell_of(B.b[i], S);
        if (c < 0 || c >= CellGrid<QT>::N)
            return false;
        if (i > 0 && c <= cell_of(B.b[i - 1], S)) // two bounds in one cell, or not ascending
            return false;
    }
    return true;
}
static_assert(cells_are_valid<kNF4>(), "NF4 cell grid too coarse");
static_assert(cells_are_valid<kFP4>(), "FP4 cell grid too coarse");

__device__ static const CellTable<kNF4> kNF4Cells = make_cells<kNF4>();
__device__ static const CellTable<kFP4> kFP4Cells = make_cells<kFP4>();

// finite s in [-1 - ulp, 1 + ulp]
template <int QT> __device__ __forceinline__ uint32_t encode_cell(float s, const QCell* cells) {
    constexpr float S = static_cast<float>(CellGrid<QT>::S);
    constexpr uint32_t kMagicBits = 0x4B000000u; // 2^23
    const float t = __fmaf_rn(s, S, S + 8388608.0f);
    const uint32_t off = (__float_as_uint(t) << 3) - (kMagicBits << 3); // cell index * sizeof(QCell)
    const uint2 e = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(cells) + off);
    const bool above = s > __uint_as_float(e.x);
    if constexpr (QT == kNF4)
        return (e.y & 0xFFFFu) + (above ? 1u : 0u); // sorted table: the code above is the code below + 1
    else
        return above ? (e.y >> 16) : (e.y & 0xFFFFu);
}

// #{bounds < s} by a 4-level compare tree; used for the inf/NaN and tail-division paths only.
template <int QT> __device__ __forceinline__ int encode4(float s) {
    constexpr Bounds B = (QT == kNF4) ? kNF4Bounds : kFP4Bounds;
    int pos = (s > B.b[7]) ? 8 : 0;
    pos += (s > ((pos == 8) ? B.b[11] : B.b[3])) ? 4 : 0;
    {
        const float lo = (pos & 4) ? B.b[5] : B.b[1];
        const float hi = (pos & 4) ? B.b[13] : B.b[9];
        pos += (s > ((pos & 8) ? hi : lo)) ? 2 : 0;
    }
    {
        float t = B.b[0];
#pragma unroll
        for (int j = 1; j < 8; ++j)
            t = (pos == 2 * j) ? B.b[2 * j] : t;
        pos += (s > t) ? 1 : 0;
    }
    pos = (s != s) ? 15 : pos; // bucketize sorts NaN last
    if (QT == kNF4)
        return pos;
    constexpr uint64_t order = [] {
        uint64_t w = 0;
        for (int i = 0; i < 16; ++i)
            w |= static_cast<uint64_t>(kFP4Order[i]) << (4 * i);
        return w;
    }();
    return static_cast<int>((order >> (4 * pos)) & 0xF);
}

__device__ __forceinline__ float clamp_pm1(float v) {
    // NaN passes through (torch.clamp); written with compares so no fmin/fmax NaN-dropping
    v = (v < -1.0f) ? -1.0f : v;
    v = (v > 1.0f) ? 1.0f : v;
    return v;
}

// ---- 8 consecutive elements as raw bits -------------------------------------------------------------
template <typename T> struct Raw8 {
    static constexpr int W = sizeof(T) == 2 ? 4 : 8;
    uint32_t w[W];

    __device__ __forceinline__ float elem(int i) const {
        if constexpr (sizeof(T) == 4) {
            return __uint_as_float(w[i]);
        } else {
            const uint16_t h = static_cast<uint16_t>(w[i >> 1] >> (16 * (i & 1)));
            return static_cast<float>(__builtin_bit_cast(T, h));
        }
    }
    // max over the 8 elements of the |x| bit pattern, widened to the fp32 pattern domain's ordering:
    // returned in the element type's own pattern space (16-bit patterns for fp16/bf16)
    __device__ __forceinline__ uint32_t abs_bits_max() const {
        if constexpr (sizeof(T) == 4) {
            uint32_t m = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                m = max(m, w[i] & 0x7FFFFFFFu);
            return m;
        } else {
            typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
            u16x2 m = {0, 0};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                m = __builtin_elementwise_max(m, __builtin_bit_cast(u16x2, w[i] & 0x7FFF7FFFu));
            return max(static_cast<uint32_t>(m.x), static_cast<uint32_t>(m.y));
        }
    }
};

// |x| pattern in T's space -> fp32 value (NaN stays NaN, inf stays inf)
template <typename T> __device__ __forceinline__ float abs_pattern_to_f32(uint32_t p) {
    if constexpr (sizeof(T) == 4)
        return __uint_as_float(p);
    else
        return static_cast<float>(__builtin_bit_cast(T, static_cast<uint16_t>(p)));
}
template <typename T> __device__ __forceinline__ bool abs_pattern_is_special(uint32_t p) {
    if constexpr (sizeof(T) == 4)
        return p >= 0x7F800000u;
    else if constexpr (__is_same(T, bf16))
        return p >= 0x7F80u;
    else
        return p >= 0x7C00u;
}

template <typename T>
__device__ __forceinline__ void load8(const T* __restrict__ A, long base, long n, bool vec_ok, Raw8<T>& r) {
    if (vec_ok) { // caller guarantees base + 8 <= n
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4* p = reinterpret_cast<const u32x4*>(A + base);
        const u32x4 a = stream_load<sizeof(T) == 2>(p);
        r.w[0] = a.x, r.w[1] = a.y, r.w[2] = a.z, r.w[3] = a.w;
        if constexpr (sizeof(T) == 4) {
            const u32x4 b = stream_load<sizeof(T) == 2>(p + 1);
            r.w[4] = b.x, r.w[5] = b.y, r.w[6] = b.z, r.w[7] = b.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < Raw8<T>::W; ++i)
            r.w[i] = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (base + i < n) {
                if constexpr (sizeof(T) == 4)
                    r.w[i] = __float_as_uint(A[base + i]);
                else
                    r.w[i >> 1] |= static_cast<uint32_t>(__builtin_bit_cast(uint16_t, A[base + i])) << (16 * (i & 1));
            }
        }
    }
}

// max over aligned groups of WIDTH lanes, result in every lane of the group. Steps 1,2,4,8 are DPP
// operand modifiers of the v_max itself (lane^1, lane^2, mirror within 8, mirror within 16 - once the
// smaller group is uniform a mirror is as good as an xor); 16 and 32 cross rows through ds_bpermute.
template <int CTRL> __device__ __forceinline__ uint32_t dpp_max_u32(uint32_t v) {
    const int moved = __builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, 0xF, 0xF, true);
    return max(v, static_cast<uint32_t>(moved));
}
template <int WIDTH> __device__ __forceinline__ uint32_t group_max_u32(uint32_t v) {
    if constexpr (WIDTH >= 2)
        v = dpp_max_u32<0xB1>(v);
    if constexpr (WIDTH >= 4)
        v = dpp_max_u32<0x4E>(v);
    if constexpr (WIDTH >= 8)
        v = dpp_max_u32<0x141>(v);
    if constexpr (WIDTH >= 16)
        v = dpp_max_u32<0x140>(v);
    if constexpr (WIDTH >= 32) {
        const int lane = static_cast<int>(threadIdx.x) & 63;
        v = max(v, static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute((lane ^ 16) << 2, static_cast<int>(v))));
        if constexpr (WIDTH >= 64)
            v = max(v, static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, static_cast<int>(v))));
    }
    return v;
}

// One workgroup = 256 threads = CH chunks of 2048 elements. BS <= 2048: a chunk holds 2048/BS blocks;
// BS = 4096: a block is two chunks (CH even).
// PIPE (whole, aligned tiles only; round 4): a workgroup walks several tiles, grid-strided, with the NEXT tile's loads in
// flight while it encodes the current one. Built on the theory that the one-tile form - exactly one round of workgroups on the
// chip for a 4096^2 weight, 8 per CU, all loading, then all encoding, then all storing - adds its three phases up. Measured
// (profiles/r4_quantize4_pipelined_ab.txt, 4096^2 bf16): NF4 11.26 vs 11.06 us - the theory is wrong there, the NF4 kernel keeps
// the one-tile form; FP4 (1025-cell table, more encode work per element) 12.33 vs 13.20: FP4 takes the pipelined form.
template <typename T, int BS, int QT, int CH, bool PIPE = false>
__global__ __launch_bounds__(256) void quantize4_kernel(const T* __restrict__ A, float* __restrict__ absmax,
                                                        uint8_t* __restrict__ out, long n, int vec_ok) {
    constexpr int TILE = 2048 * CH;
    constexpr int CPB = BS > 2048 ? BS / 2048 : 1;             // chunks per block
    constexpr int NB = CH / CPB;                               // blocks (or block groups) per tile along chunks
    constexpr int GROUP = (BS < 2048 ? BS : 2048) / 8;         // lanes sharing one quant block within a chunk
    static_assert(CH % CPB == 0, "tile must hold whole blocks");
    constexpr int NCELL = CellGrid<QT>::N;
    __shared__ QCell cells[NCELL];
    __shared__ uint32_t wave_max[NB][4];

    const int tid = threadIdx.x;
    long tile_base = static_cast<long>(blockIdx.x) * TILE;
    const long ntiles = n / TILE; // (PIPE: n is a whole number of tiles)
    long tile = blockIdx.x;

    Raw8<T> x[CH];
    [[maybe_unused]] Raw8<T> xn[PIPE ? CH : 1];
    if (PIPE || (vec_ok != 0 && tile_base + TILE <= n)) { // whole tile in range: CH back-to-back vector loads
#pragma unroll
        for (int c = 0; c < CH; ++c)
            load8<T>(A, tile_base + c * 2048 + static_cast<long>(tid) * 8, n, true, x[c]);
    } else {
#pragma unroll
        for (int c = 0; c < CH; ++c)
            load8<T>(A, tile_base + c * 2048 + static_cast<long>(tid) * 8, n, false, x[c]);
    }

    {
        const QCell* src = (QT == kNF4) ? kNF4Cells.c : kFP4Cells.c;
        for (int i = tid; i < NCELL; i += 256)
            cells[i] = src[i];
    }
  bool more = false;
  do { // (one pass unless PIPE)
    if constexpr (PIPE) {
        // the next tile of this workgroup - or, behind the last one, the same tile again (an L2 hit that is never used): the
        // loads are unconditional, so the wait in front of the encode stays a counted one (the CH loads just issued may fly on)
        const long next = tile + gridDim.x < ntiles ? tile + gridDim.x : tile;
#pragma unroll
        for (int c = 0; c < CH; ++c)
            load8<T>(A, next * TILE + c * 2048 + static_cast<long>(tid) * 8, n, true, xn[c]);
    }

    // block-wide max of the |x| patterns
    uint32_t mb[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        uint32_t m = 0;
#pragma unroll
        for (int j = 0; j < CPB; ++j)
            m = max(m, x[b * CPB + j].abs_bits_max());
        mb[b] = group_max_u32<(GROUP < 64 ? GROUP : 64)>(m);
        if constexpr (GROUP > 64) {
            if ((tid & 63) == 0)
                wave_max[b][tid >> 6] = mb[b];
        }
    }
    __syncthreads(); // cell table (and wave_max) visible
    if constexpr (GROUP > 64) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if constexpr (GROUP == 128)
                mb[b] = max(wave_max[b][(tid >> 7) * 2], wave_max[b][(tid >> 7) * 2 + 1]);
            else
                mb[b] = max(max(wave_max[b][0], wave_max[b][1]), max(wave_max[b][2], wave_max[b][3]));
        }
    }

    const long nblocks = (n + BS - 1) / BS;
    const long rem = n % BS;
    const float tiny = 1e-38f;

#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const long base = tile_base + c * 2048 + static_cast<long>(tid) * 8;
        if constexpr (!PIPE) {
            if (base >= n)
                break;
        }
        const long blk = base / BS;
        const bool tail = (rem != 0) && (blk == nblocks - 1);
        const uint32_t pat = mb[c / CPB];
        float am = abs_pattern_to_f32<T>(pat);
        if (tail)
            am = (am != am) ? am : fmaxf(am, tiny); // torch.clamp(min=) keeps NaN
        if ((c % CPB) == 0 && (base % BS) == 0)
            absmax[blk] = am;

        uint32_t wq = 0; // packed byte i = (q[2i] << 4) | q[2i+1]
        uint32_t q[8];
        if (!tail && !abs_pattern_is_special<T>(pat)) {
            const float inv = 1.0f / fmaxf(am, tiny);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                q[i] = encode_cell<QT>(x[c].elem(i) * inv, cells);
        } else if (!tail) {
            const float inv = 1.0f / fmaxf(am, tiny); // fmaxf drops NaN here exactly as the oracle's clamp does not:
            const float inv_nan = (am != am) ? am : inv; // keep NaN so every code of the block becomes 15
#pragma unroll
            for (int i = 0; i < 8; ++i)
                q[i] = encode4<QT>(clamp_pm1(x[c].elem(i) * inv_nan));
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                q[i] = encode4<QT>(clamp_pm1(x[c].elem(i) / am));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            wq |= ((q[2 * i] << 4) | q[2 * i + 1]) << (8 * i);

        if (base + 8 <= n) {
            stream_store<sizeof(T) == 2>(wq, reinterpret_cast<uint32_t*>(out + (base >> 1)));
        } else {
            // ragged end: byte stores; an odd n pads the last low nibble with the code of s = 0
            const uint32_t pad = encode4<QT>(0.0f);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long e = base + 2 * i;
                if (e < n) {
                    const uint32_t lo = (e + 1 < n) ? q[2 * i + 1] : pad;
                    out[e >> 1] = static_cast<uint8_t>((q[2 * i] << 4) | lo);
                }
            }
        }
    }
    if constexpr (PIPE) {
        tile += gridDim.x;
        more = tile < ntiles;
        tile_base = tile * TILE;
#pragma unroll
        for (int c = 0; c < CH; ++c)
            x[c] = xn[c];
        if constexpr (GROUP > 64)
            __syncthreads(); // wave_max is rewritten by the next pass
    }
  } while (PIPE && more);
}

// (sweeps / tests: 1 = the one-tile form everywhere, anything else = built-in choice; thread-local like the other knobs)
thread_local TlsKnob g_q4_variant{0};

template <typename T, int QT> void launch_quantize4(const T* A, float* absmax, uint8_t* out, int blocksize, long n,
                                                    hipStream_t stream) {
    if (n <= 0)
        return;
    const int vec_ok = aligned_to(A, 16) && aligned_to(out, 4);
    // 4 chunks per workgroup once that still leaves >= 4 workgroups per CU (amortises the cell-table fill
    // and puts 4 loads per lane in flight); small inputs keep the smallest tile to spread over the CUs.
    const bool wide = n >= 4L * 256 * 8192;
    // the pipelined form: whole aligned tiles, and enough of them that every workgroup gets >= 4 (4 workgroups per CU)
    const int q4_variant = g_q4_variant.load(std::memory_order_relaxed);
    const long pipe_grid = 4L * device_cu_count_or_default();
#define BNB_Q4_LAUNCH_PIPE(BS, CH)                                                                 \
    {                                                                                              \
        hipLaunchKernelGGL((quantize4_kernel<T, BS, QT, CH, true>), dim3(static_cast<unsigned>(pipe_grid)), dim3(256), 0, stream, A, absmax, \
                           out, n, vec_ok);                                                        \
    }
#define BNB_Q4_LAUNCH(BS, CH)                                                                      \
    {                                                                                              \
        constexpr long TILE = 2048L * CH;                                                          \
        const long grid = (n + TILE - 1) / TILE;                                                   \
        hipLaunchKernelGGL((quantize4_kernel<T, BS, QT, CH>), dim3(static_cast<unsigned>(grid)), dim3(256), 0, \
                           stream, A, absmax, out, n, vec_ok);                                     \
    }
#define BNB_Q4_CASE(BS)                                                                            \
    case BS: {                                                                                     \
        constexpr int MINCH = BS > 2048 ? BS / 2048 : 1;                                           \
        constexpr int PCH = BS > 2048 ? BS / 2048 : 2;                                             \
        if (q4_variant == 3) /* A/B: 8 chunks per workgroup (round 5: 8 - 13 % slower everywhere) */ \
            BNB_Q4_LAUNCH(BS, 8)                                                                   \
        else if (q4_variant == 2 && MINCH <= 2) /* A/B: 2 chunks per workgroup whatever the code */ \
            BNB_Q4_LAUNCH(BS, 2)                                                                   \
        else if (q4_variant == 4 && wide) /* A/B: round 4's 4 chunks per workgroup */             \
            BNB_Q4_LAUNCH(BS, 4)                                                                   \
        else if (QT == kFP4 && q4_variant != 1 && vec_ok && (n % (2048L * PCH)) == 0 && n / (2048L * PCH) >= 4 * pipe_grid) \
            BNB_Q4_LAUNCH_PIPE(BS, PCH)                                                            \
        else if (wide && QT == kNF4 && MINCH <= 2)                                                 \
            BNB_Q4_LAUNCH(BS, 2) /* round 5 (profiles/r5_stream_kernels_ab.txt): 2 chunks per workgroup, 4096^2 bf16 10.64 -> 9.94 us, fp32 16.6 -> 15.7, 8192^2 30.8 -> 30.1 */ \
        else if (wide)                                                                             \
            BNB_Q4_LAUNCH(BS, 4)                                                                   \
        else                                                                                       \
            BNB_Q4_LAUNCH(BS, MINCH)                                                               \
        break;                                                                                     \
    }
    switch (blocksize) {
        BNB_Q4_CASE(32)
        BNB_Q4_CASE(64)
        BNB_Q4_CASE(128)
        BNB_Q4_CASE(256)
        BNB_Q4_CASE(512)
        BNB_Q4_CASE(1024)
        BNB_Q4_CASE(2048)
        BNB_Q4_CASE(4096)
    default:
        fprintf(stderr, "bitsandbytes_amd: quantize_4bit: unsupported blocksize %d\n", blocksize);
        exit(1);
    }
#undef BNB_Q4_CASE
#undef BNB_Q4_LAUNCH
#undef BNB_Q4_LAUNCH_PIPE
    BNB_CHECK_LAUNCH();
}

} // namespace

void quantize_4bit_set_variant(int variant) { g_q4_variant.store(variant, std::memory_order_relaxed); }

void quantize_4bit_f32(const float* A, float* absmax, uint8_t* out, int blocksize, long n, int quant_type,
                       hipStream_t s) {
    if (quant_type == kNF4)
        launch_quantize4<float, kNF4>(A, absmax, out, blocksize, n, s);
    else
        launch_quantize4<float, kFP4>(A, absmax, out, blocksize, n, s);
}
void quantize_4bit_f16(const void* A, float* absmax, uint8_t* out, int blocksize, long n, int quant_type,
                       hipStream_t s) {
    if (quant_type == kNF4)
        launch_quantize4<f16, kNF4>(static_cast<const f16*>(A), absmax, out, blocksize, n, s);
    else
        launch_quantize4<f16, kFP4>(static_cast<const f16*>(A), absmax, out, blocksize, n, s);
}
void quantize_4bit_bf16(const void* A, float* absmax, uint8_t* out, int blocksize, long n, int quant_type,
                        hipStream_t s) {
    if (quant_type == kNF4)
        launch_quantize4<bf16, kNF4>(static_cast<const bf16*>(A), absmax, out, blocksize, n, s);
    else
        launch_quantize4<bf16, kFP4>(static_cast<const bf16*>(A), absmax, out, blocksize, n, s);
}

} // namespace bnb
