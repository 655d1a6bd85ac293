// blockwise8.hip — 8-bit blockwise quantize / dequantize with a 256-entry code (General8bit).
//
// On the 4-bit path these only serve double quantization (`compress_statistics=True`): the fp32
// absmax vector of the 4-bit blocks is itself quantized in blocks of 256 with the dynamic map
// (reference bitsandbytes/functional.py:938-951, :613-769).
//
// quantize: parity target is the reference CPU backend's native kernel, whose rule is NOT
// nearest-code but a 65536-bin discretisation followed by a table lookup
// (reference csrc/cpu_ops.cpp:501-520 build_quantize_lut, :569-572 norm_to_lut_index, :574-665):
//     absmax = max|x| over the block;  all-zero block -> codes 0
//     u   = uint16( (clamp(x * (1/absmax), -1, 1) + 1) * 0.5 * 65535 + 0.5 )
//     val(u) = -1 + (2*u) / 65535                    (fp32, IEEE division)
//     q   = #{ i < 255 : 0.5*(code[i] + code[i+1]) < val(u) }
// No element pays the division: val(u) is monotone in u, so every midpoint i has a threshold bin
// T_i = min{u : val(u) > mid_i} (found once per workgroup: the algebraic inverse, corrected by comparisons that evaluate val()
// with exactly the expression above) and q = #{i : T_i <= u} - an integer count. Two encoders evaluate it:
//  * quantize8_kernel (small inputs): a 1024-cell table over u (64 bins per cell) gives the count below the cell and, when the
//    cell holds exactly one threshold, its offset inside the cell: an element then costs ONE LDS read and one compare; the few
//    cells with several thresholds (the dynamic map is dense only around zero) count them in a short loop;
//  * quantize8_lut_kernel (>= 2^20 elements): the 64 K-entry byte table itself, in LDS, built by a prefix sum over the marked
//    thresholds. Same function of u, bit for bit, in both.
//
// dequantize: out[i] = T(code[A[i]] * absmax[i / blocksize])   (reference csrc/cpu_ops.cpp:436-486).
//
// Layout: a lane owns 4 consecutive elements (one 16-byte fp32 / 8-byte 16-bit load, one 4-byte store of
// codes); a wavefront covers 256 consecutive elements per step; the block max is a DPP/shuffle reduction over
// blocksize/4 lanes, or an accumulation over blocksize/256 steps of one wavefront for the larger blocks.
#include "bnb_common.h"

#include <atomic>

namespace bnb {

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bin_value(unsigned u) {
    return -1.0f + (2.0f * static_cast<float>(u)) / 65535.0f; // the reference's expression, IEEE division
}

// T(m) = min{u in [0, 65536] : bin_value(u) > m}  (65536 if none). bin_value is non-decreasing in u, so instead of 17 bisection
// steps (17 correctly rounded divisions on the critical path of every launch) start from the algebraic inverse and walk the
// few steps the roundings can be off; every comparison is made with bin_value itself, so T is the same number.
__device__ __forceinline__ unsigned first_bin_above(float m) {
    if (m < -1.0f)
        return 0u; // bin_value(0) = -1 > m
    if (!(m < 1.0f))
        return 65536u; // bin_value(65535) = 1 is not > m; NaN compares false everywhere (what the bisection returned as well)
    int c = static_cast<int>((m + 1.0f) * 32767.5f);
    c = c < 0 ? 0 : (c > 65535 ? 65535 : c);
    while (c > 0 && bin_value(static_cast<unsigned>(c - 1)) > m)
        --c;
    while (c < 65536 && !(bin_value(static_cast<unsigned>(c)) > m))
        ++c;
    return static_cast<unsigned>(c);
}

__device__ __forceinline__ unsigned bin_of(float x, float inv) {
    float v = x * inv;
    v = fminf(fmaxf(v, -1.0f), 1.0f);
    const float t = (v + 1.0f) * 0.5f;
    const float p = __fmul_rn(t, 65535.0f); // separate rounding, as the un-contracted source reads
    const float r = __fadd_rn(p, 0.5f);     // (fused and un-fused agree on every tested input)
    return static_cast<unsigned>(r) & 0xFFFFu;
}

template <typename T> __device__ __forceinline__ void load4(const T* __restrict__ A, long i, long n, bool vec_ok, float (&x)[4]) {
    if (vec_ok && i + 4 <= n) {
        if constexpr (sizeof(T) == 4) {
            const f32x4_t r = stream_load<true>(reinterpret_cast<const f32x4_t*>(A + i));
#pragma unroll
            for (int j = 0; j < 4; ++j)
                x[j] = r[j];
        } else {
            typedef T v4 __attribute__((ext_vector_type(4)));
            const v4 r = stream_load<true>(reinterpret_cast<const v4*>(A + i));
#pragma unroll
            for (int j = 0; j < 4; ++j)
                x[j] = static_cast<float>(r[j]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            x[j] = (i + j < n) ? static_cast<float>(A[i + j]) : 0.0f;
    }
}

// Sum of 256 values, one per thread of a 256-thread workgroup, result in every thread. The order is a balanced binary tree over
// the thread index (wave_sum pairs lanes l and l ^ 1, then pairs of pairs, ...; the four wavefront sums are added as (0 + 1) + (2 + 3)):
// a fixed function of the inputs, which the tests restate in numpy bit for bit.
__device__ __forceinline__ float workgroup_sum_256(float v, float* slots /* [4] in LDS */) {
    v = wave_sum(v);
    __syncthreads(); // (the slots may still be read from a previous call)
    if ((threadIdx.x & 63) == 0)
        slots[threadIdx.x >> 6] = v;
    __syncthreads();
    return (slots[0] + slots[1]) + (slots[2] + slots[3]);
}

// The encoder's two tables from the 256-entry code (256-thread workgroup; ends with a barrier). thr[i] = T_i = the first 16-bit bin
// whose value lies above the mid-point of code[i] and code[i + 1] (thr[255] = 65536); cell[c], c = bin >> 6: bits 0-7 thresholds
// below the cell, bits 8-15 thresholds inside it, bits 16-21 offset of the first one inside (what the common single-threshold case
// compares against).
__device__ __forceinline__ void build_cell_tables(const float* __restrict__ code, uint32_t* thr /* [256] LDS */, uint32_t* cell /* [1024] LDS */) {
    // below[c] = #{i < 255 : T_i < 64 c}, c = 0 .. 1024, as a prefix sum: a threshold T counts from cell (T >> 6) + 1 on, so mark it
    // there (LDS atomic: thresholds may share a cell) and scan. (A binary search per cell was 40 dependent LDS reads per thread:
    // ~1.7 us on the critical path of every launch.) cnt[0] stays 0; T = 65536 - no bin above the mid-point - lands in the spare slot.
    __shared__ __attribute__((aligned(16))) uint32_t cnt[1028];
    __shared__ uint32_t wave_tot[4];
    const int tid = threadIdx.x;
    const float lo = code[tid], hi = (tid < 255) ? code[tid + 1] : 0.0f;
#pragma unroll
    for (int c = tid; c < 1028; c += 256)
        cnt[c] = 0u;
    const unsigned t = first_bin_above((tid < 255) ? 0.5f * (lo + hi) : __builtin_inff());
    thr[tid] = t;
    __syncthreads();
    if (tid < 255)
        atomicAdd(&cnt[(t >> 6) + 1u], 1u);
    __syncthreads();
    // thread tid owns c = 4 tid + 1 .. 4 tid + 4
    const u32x4_t mine = *reinterpret_cast<const u32x4_t*>(&cnt[4 * tid]); // cnt[4 tid .. 4 tid + 3]
    const uint32_t last = cnt[4 * tid + 4];
    const unsigned own[4] = {mine[1], mine[2], mine[3], last};
    const unsigned sum = own[0] + own[1] + own[2] + own[3];
    unsigned incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = __shfl_up(incl, off, 64);
        incl += ((tid & 63) >= off) ? o : 0u;
    }
    if ((tid & 63) == 63)
        wave_tot[tid >> 6] = incl;
    __syncthreads();
    unsigned run = incl - sum; // thresholds below cell 4 tid (= below[4 tid]: cnt[0 .. 4 tid] summed)
#pragma unroll
    for (int w = 0; w < 3; ++w)
        run += (w < (tid >> 6)) ? wave_tot[w] : 0u;
    // cell c = 4 tid + j: below[c] = run + own[0 .. j - 1], inside = below[c + 1] - below[c] = own[j]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned c = 4u * tid + j;
        const unsigned below = run, inside = own[j];
        const unsigned off = (inside > 0u) ? (thr[below] - c * 64u) : 0u;
        cell[c] = below | (inside << 8) | (off << 16);
        run += inside;
    }
    __syncthreads();
}

constexpr int kQ8TableWords = 256 + 1024; // thr[256] then cell[1024]: what the partial-sum launch leaves for the nested encoder

constexpr int kSumChunk = 1024;   // elements per workgroup step of the partial-sum kernel: 4 per thread
constexpr int kSumPartials = 256; // at most this many partial sums, whatever the length

// Partial sums for the mean of the fp32 absmax vector of double quantisation (reference bitsandbytes/functional.py:938-951:
// offset = absmax.mean()). Workgroup c adds elements [c * 1024 * steps, (c + 1) * 1024 * steps): every step of 1024 elements as a
// balanced binary tree in index order (zeros past the end), the steps one after the other. partial[c] is one float; the consumer
// (quantize8_kernel<float, 256, true>) adds the <= 256 partials as one more balanced tree and divides by n.
//
// One workgroup more than there are chunks: the last one builds the 8-bit encoder's threshold / cell tables from the code and leaves
// them in `tables` (kQ8TableWords dwords) - once per call instead of once per encoder workgroup, and beside the sums, not after them.
__global__ __launch_bounds__(256) void absmax_partial_sums_kernel(const float* __restrict__ A, float* __restrict__ partial, long n, int steps,
                                                                  const float* __restrict__ code, uint32_t* __restrict__ tables) {
    if (tables != nullptr && blockIdx.x == gridDim.x - 1) {
        __shared__ uint32_t thr[256];
        __shared__ uint32_t cell[1024];
        build_cell_tables(code, thr, cell);
        tables[threadIdx.x] = thr[threadIdx.x];
#pragma unroll
        for (int c = threadIdx.x; c < 1024; c += 256)
            tables[256 + c] = cell[c];
        return;
    }
    __shared__ float slots[4];
    const long begin = static_cast<long>(blockIdx.x) * kSumChunk * steps + threadIdx.x * 4;
    const bool vec_ok = (reinterpret_cast<uintptr_t>(A) & 15u) == 0;
    float acc = 0.0f;
    for (int s = 0; s < steps; ++s) {
        const long i = begin + static_cast<long>(s) * kSumChunk;
        float x[4];
        if (vec_ok && i + 4 <= n) {
            const f32x4_t r = *reinterpret_cast<const f32x4_t*>(A + i);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                x[j] = r[j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                x[j] = (i + j < n) ? A[i + j] : 0.0f;
        }
        const float t = workgroup_sum_256(__fadd_rn(__fadd_rn(x[0], x[1]), __fadd_rn(x[2], x[3])), slots);
        acc = (s == 0) ? t : __fadd_rn(acc, t);
    }
    if (threadIdx.x == 0)
        partial[blockIdx.x] = acc;
}

// BS = quantization block size (power of two, 64 .. 4096). One wavefront step = 256 elements.
// SHIFT (double quantisation, T = float, BS = 256): the input is A - offset with offset = (sum of the partials) / n, computed by every
// workgroup in the same fixed order, written to offset_out by the first one - the reference's `absmax -= absmax.mean()` folded into
// the encoder's load (one launch and one pass over the vector instead of three).
template <typename T, int BS, bool SHIFT = false>
__global__ __launch_bounds__(256) void quantize8_kernel(const float* __restrict__ code, const T* __restrict__ A,
                                                        float* __restrict__ absmax, uint8_t* __restrict__ out, long n,
                                                        int vec_ok, const float* __restrict__ partial = nullptr, int n_partial = 0,
                                                        float* __restrict__ offset_out = nullptr, const uint32_t* __restrict__ tables = nullptr) {
    float shift = 0.0f;
    [[maybe_unused]] uint32_t t_thr = 0, t_cell[4] = {0, 0, 0, 0};
    const bool shared_tables = SHIFT && tables != nullptr; // (nullptr: tuning knob reserved0 = 9, the per-workgroup build, for A/B)
    if (shared_tables) {
        // (requested before the sum of the partials, used after it)
        t_thr = tables[threadIdx.x];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            t_cell[j] = tables[256 + j * 256 + threadIdx.x];
    }
    if constexpr (SHIFT) {
        __shared__ float slots[4];
        const float total = workgroup_sum_256(static_cast<int>(threadIdx.x) < n_partial ? partial[threadIdx.x] : 0.0f, slots);
        shift = total / static_cast<float>(n); // IEEE division
        if (blockIdx.x == 0 && threadIdx.x == 0)
            *offset_out = shift;
    }
    __shared__ uint32_t thr[256];   // T_i, ascending; thr[255] = 65536
    __shared__ uint32_t cell[1024]; // see build_cell_tables
    const int tid = threadIdx.x;
    if (shared_tables) {
        // double quantisation: the tables were built once, by the extra workgroup of the partial-sum launch
        thr[tid] = t_thr;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            cell[j * 256 + tid] = t_cell[j];
        __syncthreads();
    } else {
        build_cell_tables(code, thr, cell);
    }

    const int lane = tid & 63;
    const long wave_global = static_cast<long>(blockIdx.x) * 4 + (tid >> 6);
    const long wave_count = static_cast<long>(gridDim.x) * 4;
    constexpr int SPB = BS > 256 ? BS / 256 : 1;   // wavefront steps per block
    constexpr int GROUP = BS >= 256 ? 64 : BS / 4; // lanes sharing one block within a step
    const long units = (n + 256L * SPB - 1) / (256L * SPB); // one unit = SPB steps = max(256, BS) elements

    for (long unit = wave_global; unit < units; unit += wave_count) {
        const long base = unit * 256L * SPB + lane * 4;
        float x[SPB][4];
        float m = 0.0f;
#pragma unroll
        for (int sp = 0; sp < SPB; ++sp) {
            load4<T>(A, base + sp * 256L, n, vec_ok != 0, x[sp]);
            if constexpr (SHIFT) {
#pragma unroll
                for (int j = 0; j < 4; ++j) // (elements past the end stay zero: they must not enter the block's absmax)
                    x[sp][j] = (base + sp * 256L + j < n) ? __fsub_rn(x[sp][j], shift) : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                m = fmaxf(m, fabsf(x[sp][j]));
        }
        m = group_max<GROUP>(m);
        if ((lane % GROUP) == 0 && base < n)
            absmax[base / BS] = m;
        const float inv = 1.0f / m; // m == 0: inf; the codes are forced to 0 below (reference: all-zero block)
#pragma unroll
        for (int sp = 0; sp < SPB; ++sp) {
            uint32_t q4 = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned u = bin_of(x[sp][j], inv);
                const unsigned ce = cell[u >> 6];
                unsigned q = ce & 0xFFu;
                const unsigned cnt = (ce >> 8) & 0xFFu;
                if (cnt > 1u) { // several thresholds in this cell (the code is dense only around zero): binary search
                    unsigned lo = 0;
#pragma unroll
                    for (unsigned step = 128; step >= 1; step >>= 1)
                        lo += (lo + step - 1 < cnt && thr[q + lo + step - 1] <= u) ? step : 0u;
                    q += lo;
                } else {
                    q += (cnt == 1u && (u & 63u) >= (ce >> 16)) ? 1u : 0u;
                }
                q = (m == 0.0f) ? 0u : q;
                q4 |= q << (8 * j);
            }
            const long i = base + sp * 256L;
            if (vec_ok && i + 4 <= n) {
                stream_store<true>(q4, reinterpret_cast<uint32_t*>(out + i));
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (i + j < n)
                        out[i + j] = static_cast<uint8_t>(q4 >> (8 * j));
            }
        }
    }
}

// Large inputs: the reference's own formulation - a 65536-entry byte table (csrc/cpu_ops.cpp:501-520 builds exactly this) -
// held in LDS: one ds_read_u8 per element whatever the data (the cell table above falls into an 8-step search inside the
// cells around zero, which is where block-normalised data lives: 2.1 TB/s on randn input). 64 KiB of table per workgroup is
// only worth building when a workgroup has megabytes to encode, so small inputs (the absmax vectors of double quantisation)
// keep the cell kernel. Same function of u, bit for bit: lut[u] = #{i : T_i <= u}.
thread_local TlsKnob g_q8_variant{0}; // tuning / tests (bnb_mi355x_set_tuning reserved0): 1 = cell-table kernel, 2 = byte-table kernel, else by size
constexpr long kQ8LutMinElements = 1L << 20; // below this the 64 KiB table per workgroup costs more than it saves
constexpr int kQ8LutThreads = 512;
constexpr int kQ8LutLds = 65536 + 1024 + 1024;

template <typename T, int BS>
__global__ __launch_bounds__(kQ8LutThreads) void quantize8_lut_kernel(const float* __restrict__ code, const T* __restrict__ A,
                                                                      float* __restrict__ absmax, uint8_t* __restrict__ out, long n,
                                                                      int vec_ok) {
    extern __shared__ __attribute__((aligned(16))) unsigned char q8_smem[];
    uint8_t* const lut = q8_smem;
    uint32_t* const thr = reinterpret_cast<uint32_t*>(q8_smem + 65536); // T_i, ascending; thr[255] = 65536
    float* const mid = reinterpret_cast<float*>(q8_smem + 65536 + 1024);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    constexpr int WAVES = kQ8LutThreads / 64;
    const long wave_global = static_cast<long>(blockIdx.x) * WAVES + (tid >> 6);
    const long wave_count = static_cast<long>(gridDim.x) * WAVES;
    constexpr int SPB = BS > 256 ? BS / 256 : 1;   // wavefront steps per block
    constexpr int GROUP = BS >= 256 ? 64 : BS / 4; // lanes sharing one block within a step
    constexpr bool PREFETCH = SPB <= 4;            // the next unit's loads in flight while this one is encoded
    const long units = (n + 256L * SPB - 1) / (256L * SPB);

    // U units (1 KiB of input per wavefront each, for fp32) are loaded together and the next U are requested before these are
    // encoded: 8 KiB per wavefront in flight - one unit at a time left the kernel latency-bound at 2 TB/s
    constexpr int U = SPB >= 4 ? 1 : 4 / SPB;
    const long groups = (units + U - 1) / U;
    auto load_group = [&](float (&dst)[U][SPB][4], long grp) {
#pragma unroll
        for (int uu = 0; uu < U; ++uu)
#pragma unroll
            for (int sp = 0; sp < SPB; ++sp)
                load4<T>(A, (grp * U + uu) * 256L * SPB + lane * 4 + sp * 256L, n, vec_ok != 0, dst[uu][sp]);
    };
    // the first group's loads go out before the tables are built: the table build hides their latency
    float xn[U][SPB][4];
    if (PREFETCH && wave_global < groups)
        load_group(xn, wave_global);
    if (tid < 256)
        mid[tid] = (tid < 255) ? 0.5f * (code[tid] + code[tid + 1]) : __builtin_inff();
    __syncthreads();
    if (tid < 256)
        thr[tid] = first_bin_above(mid[tid]);
    __syncthreads();
    // lut[u] = #{i : T_i <= u} as a prefix sum: mark every threshold in a zeroed byte array (a 32-bit LDS atomic on the containing
    // dword: thresholds may coincide), then thread t owns bytes [128 t, 128 t + 128): their sum, a workgroup-wide exclusive scan
    // of the 512 sums, and a SWAR running sum over its 32 dwords. No byte can overflow: there are 255 thresholds in all.
    // (A per-u walk over the sorted thresholds cost ~5 us per launch: ~10 instructions for each of the 65536 entries.)
    {
        u32x4_t* const z = reinterpret_cast<u32x4_t*>(lut);
#pragma unroll
        for (int i = 0; i < 65536 / 16 / kQ8LutThreads; ++i)
            z[i * kQ8LutThreads + tid] = u32x4_t{0u, 0u, 0u, 0u};
    }
    __syncthreads();
    if (tid < 255) {
        const unsigned t = thr[tid];
        if (t < 65536u)
            atomicAdd(reinterpret_cast<unsigned*>(lut) + (t >> 2), 1u << (8u * (t & 3u)));
    }
    __syncthreads();
    {
        uint32_t d[32];
        const u32x4_t* const src = reinterpret_cast<const u32x4_t*>(lut + tid * 128);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const u32x4_t v = src[i];
            d[4 * i] = v[0];
            d[4 * i + 1] = v[1];
            d[4 * i + 2] = v[2];
            d[4 * i + 3] = v[3];
        }
        unsigned sum = 0;
#pragma unroll
        for (int w = 0; w < 32; ++w)
            sum = __builtin_amdgcn_sad_u8(d[w], 0u, sum); // sum of the four bytes
        // exclusive scan over the workgroup: inclusive wavefront scan by shuffles, wavefront totals through LDS
        unsigned incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned o = __shfl_up(incl, off, 64);
            incl += ((tid & 63) >= off) ? o : 0u;
        }
        uint32_t* const wave_tot = reinterpret_cast<uint32_t*>(mid); // (mid is dead: the thresholds are in thr)
        __syncthreads();
        if ((tid & 63) == 63)
            wave_tot[tid >> 6] = incl;
        __syncthreads();
        unsigned run = incl - sum;
#pragma unroll
        for (int w = 0; w < kQ8LutThreads / 64; ++w)
            run += (w < (tid >> 6)) ? wave_tot[w] : 0u;
        u32x4_t* const dst = reinterpret_cast<u32x4_t*>(lut + tid * 128);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            u32x4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint32_t x = d[4 * i + e];
                x += x << 8;
                x += x << 16; // inclusive running sum inside the dword
                x += run * 0x01010101u;
                run = x >> 24;
                o[e] = x;
            }
            dst[i] = o;
        }
    }
    __syncthreads();

    for (long grp = wave_global; grp < groups; grp += wave_count) {
        float xg[U][SPB][4];
        if constexpr (PREFETCH) {
#pragma unroll
            for (int uu = 0; uu < U; ++uu)
#pragma unroll
                for (int sp = 0; sp < SPB; ++sp)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        xg[uu][sp][j] = xn[uu][sp][j];
            if (grp + wave_count < groups)
                load_group(xn, grp + wave_count);
        } else {
            load_group(xg, grp);
        }
#pragma unroll
        for (int uu = 0; uu < U; ++uu) {
            const long base = (grp * U + uu) * 256L * SPB + lane * 4;
            float m = 0.0f;
#pragma unroll
            for (int sp = 0; sp < SPB; ++sp)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    m = fmaxf(m, fabsf(xg[uu][sp][j]));
            m = group_max<GROUP>(m);
            if ((lane % GROUP) == 0 && base < n)
                absmax[base / BS] = m;
            const float inv = 1.0f / m; // m == 0: inf; the codes are forced to 0 below (reference: all-zero block)
#pragma unroll
            for (int sp = 0; sp < SPB; ++sp) {
                uint32_t q4 = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned q = lut[bin_of(xg[uu][sp][j], inv)];
                    q = (m == 0.0f) ? 0u : q;
                    q4 |= q << (8 * j);
                }
                const long i = base + sp * 256L;
                if (vec_ok && i + 4 <= n) {
                    stream_store<true>(q4, reinterpret_cast<uint32_t*>(out + i));
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (i + j < n)
                            out[i + j] = static_cast<uint8_t>(q4 >> (8 * j));
                }
            }
        }
    }
}

// U = independent 4-element units per lane and loop iteration. The first form (U = 1) had ONE 4-byte load in flight per lane - 8 KB per
// CU with 8 resident workgroups - and ran at 51 % of the HBM peak where the 4-bit kernel, same output bytes, reaches 76 % (round 5,
// profiles/r5_stream_kernels_ab.txt): with U units the loads of an iteration are all requested before the first look-up.
template <typename T, int U>
__global__ __launch_bounds__(256) void dequantize8_kernel(const float* __restrict__ code,
                                                          const uint8_t* __restrict__ A,
                                                          const float* __restrict__ absmax, T* __restrict__ out,
                                                          int bs_shift, long n, int vec_ok) {
    __shared__ float lut[256];
    lut[threadIdx.x] = code[threadIdx.x];
    __syncthreads();
    const long stride = static_cast<long>(gridDim.x) * 256 * 4 * U;
    for (long i0 = static_cast<long>(blockIdx.x) * 256 * 4 * U + threadIdx.x * 4; i0 < n; i0 += stride) {
        // (unit u of the lane: elements i0 + u * 1024 ...: every store instruction of a wavefront stays 1 KiB contiguous)
        if (vec_ok && i0 + static_cast<long>(U - 1) * 1024 + 4 <= n) {
            uint32_t q4[U];
            float s[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long i = i0 + static_cast<long>(u) * 1024;
                q4[u] = stream_load<true>(reinterpret_cast<const uint32_t*>(A + i));
                s[u] = absmax[i >> bs_shift]; // blocksize >= 4 and i % 4 == 0: one block for the four
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long i = i0 + static_cast<long>(u) * 1024;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    v[j] = rounded_f32(lut[(q4[u] >> (8 * j)) & 0xFFu] * s[u]);
                if constexpr (sizeof(T) == 4) {
                    stream_store<true>((f32x4_t{v[0], v[1], v[2], v[3]}), reinterpret_cast<f32x4_t*>(out + i));
                } else {
                    typedef T v4 __attribute__((ext_vector_type(4)));
                    v4 r;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        r[j] = static_cast<T>(v[j]);
                    stream_store<true>(r, reinterpret_cast<v4*>(out + i));
                }
            }
        } else {
            for (int u = 0; u < U; ++u) {
                const long i = i0 + static_cast<long>(u) * 1024;
                for (long e = i; e < i + 4 && e < n; ++e)
                    out[e] = static_cast<T>(rounded_f32(lut[A[e]] * absmax[e >> bs_shift]));
            }
        }
    }
}

template <typename T>
void launch_quantize8(const float* code, const T* A, float* absmax, uint8_t* out, int blocksize, long n,
                      hipStream_t stream) {
    if (n <= 0)
        return;
    const int vec_ok = aligned_to(A, 16) && aligned_to(out, 4);
    // grid-stride over wavefront units; enough workgroups to fill the chip, few enough to amortise the
    // per-workgroup threshold / cell tables
    const long unit_elems = blocksize > 256 ? blocksize : 256;
    const long units = (n + unit_elems - 1) / unit_elems;
    const int variant = g_q8_variant.load(std::memory_order_relaxed);
    if (variant == 2 || (variant != 1 && n >= kQ8LutMinElements)) {
        // byte-table kernel: two 512-thread workgroups per CU (64 KiB of table each), persistent
        long grid = (units + 31) / 32; // (a wavefront takes up to 4 units per round)
        const long max_grid = 2L * device_cu_count_or_default();
        if (grid > max_grid)
            grid = max_grid;
#define BNB_Q8L_CASE(BS)                                                                           \
    case BS: {                                                                                     \
        auto kern = quantize8_lut_kernel<T, BS>;                                                   \
        static LdsLimit lim;                                                                       \
        ensure_dynamic_lds(lim, reinterpret_cast<const void*>(kern), kQ8LutLds);                   \
        hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(grid)), dim3(kQ8LutThreads), kQ8LutLds, stream, code, A, absmax, out, n, vec_ok); \
        break;                                                                                     \
    }
        switch (blocksize) {
            BNB_Q8L_CASE(64)
            BNB_Q8L_CASE(128)
            BNB_Q8L_CASE(256)
            BNB_Q8L_CASE(512)
            BNB_Q8L_CASE(1024)
            BNB_Q8L_CASE(2048)
            BNB_Q8L_CASE(4096)
        default:
            fprintf(stderr, "bitsandbytes_amd: quantize_blockwise: unsupported blocksize %d\n", blocksize);
            exit(1);
        }
#undef BNB_Q8L_CASE
        BNB_CHECK_LAUNCH();
        return;
    }
    long grid = (units + 3) / 4;
    if (grid > 2048)
        grid = 2048; // 8 workgroups per CU: fewer (1024) measured slower, the loads need the wavefronts
#define BNB_Q8_CASE(BS)                                                                            \
    case BS:                                                                                       \
        hipLaunchKernelGGL((quantize8_kernel<T, BS>), dim3(static_cast<unsigned>(grid)), dim3(256), 0, stream, code, A, \
                           absmax, out, n, vec_ok);                                                \
        break;
    switch (blocksize) {
        BNB_Q8_CASE(64)
        BNB_Q8_CASE(128)
        BNB_Q8_CASE(256)
        BNB_Q8_CASE(512)
        BNB_Q8_CASE(1024)
        BNB_Q8_CASE(2048)
        BNB_Q8_CASE(4096)
    default:
        fprintf(stderr, "bitsandbytes_amd: quantize_blockwise: unsupported blocksize %d\n", blocksize);
        exit(1);
    }
#undef BNB_Q8_CASE
    BNB_CHECK_LAUNCH();
}

template <typename T>
void launch_dequantize8(const float* code, const uint8_t* A, const float* absmax, T* out, int blocksize, long n,
                        hipStream_t stream) {
    if (n <= 0)
        return;
    if (!is_pow2(blocksize) || blocksize < 4) {
        fprintf(stderr, "bitsandbytes_amd: dequantize_blockwise: blocksize %d is not a power of two >= 4\n", blocksize);
        exit(1);
    }
    const int vec_ok = aligned_to(A, 4) && aligned_to(out, 16);
    // Four units per lane from 2^20 elements on: 16.7 M elements 20.2 -> 17.0 us, 67 M 65.0 -> 58.4 (profiles/r5_stream_kernels_ab.txt);
    // below that - the absmax vectors of double quantisation: 262 144 elements - a quarter as many workgroups is what costs (2.46 vs
    // 2.13 us): one unit. (Tuning knob reserved0 = 6 forces the one-unit form, 7 the four-unit form.)
    const int v8 = g_q8_variant.load(std::memory_order_relaxed);
    if (v8 == 6 || (v8 != 7 && n < (1L << 20))) {
        long grid = (n + 1023) / 1024;
        grid = grid > 8192 ? 8192 : grid;
        hipLaunchKernelGGL((dequantize8_kernel<T, 1>), dim3(static_cast<unsigned>(grid)), dim3(256), 0, stream, code, A, absmax, out, ilog2(blocksize), n, vec_ok);
    } else {
        long grid = (n + 4095) / 4096;
        grid = grid > 8192 ? 8192 : grid;
        hipLaunchKernelGGL((dequantize8_kernel<T, 4>), dim3(static_cast<unsigned>(grid)), dim3(256), 0, stream, code, A, absmax, out, ilog2(blocksize), n, vec_ok);
    }
    BNB_CHECK_LAUNCH();
}

} // namespace

// Double quantisation's statistics in two launches (reference bitsandbytes/functional.py:938-951 is mean, subtract, quantize_blockwise
// with blocksize 256: three to five launches and as many host dispatches): partial sums, then the shifted encoder. absmax: the fp32
// absmax vector of the 4-bit blocks (n values, device); partial: scratch of 256 + 1280 floats (the partial sums, then the encoder's
// tables); offset_out: 1 float; out: n codes;
// absmax2: ceil(n / 256) floats.
void quantize_absmax_nested(const float* code, const float* absmax, long n, float* partial, float* offset_out, uint8_t* out,
                            float* absmax2, hipStream_t stream) {
    if (n <= 0)
        return;
    const long per = static_cast<long>(kSumChunk) * kSumPartials;
    const int steps = static_cast<int>((n + per - 1) / per);
    const int chunks = static_cast<int>((n + static_cast<long>(kSumChunk) * steps - 1) / (static_cast<long>(kSumChunk) * steps));
    uint32_t* const tables = g_q8_variant.load(std::memory_order_relaxed) == 9 ? nullptr : reinterpret_cast<uint32_t*>(partial + kSumPartials);
    static_assert(kQ8TableWords == 1280, "include/bnb_mi355x.h: scratch of blocks + 256 + 1280 floats");
    hipLaunchKernelGGL(absmax_partial_sums_kernel, dim3(static_cast<unsigned>(chunks + (tables != nullptr ? 1 : 0))), dim3(256), 0, stream, absmax, partial, n, steps, code, tables);
    BNB_CHECK_LAUNCH();
    const int vec_ok = aligned_to(absmax, 16) && aligned_to(out, 4);
    const long units = (n + 255) / 256;
    long grid = (units + 3) / 4;
    if (grid > 2048)
        grid = 2048;
    hipLaunchKernelGGL((quantize8_kernel<float, 256, true>), dim3(static_cast<unsigned>(grid)), dim3(256), 0, stream, code, absmax, absmax2,
                       out, n, vec_ok, partial, chunks, offset_out, tables);
    BNB_CHECK_LAUNCH();
}

void quantize_8bit_set_variant(int variant) { g_q8_variant.store(variant, std::memory_order_relaxed); }

void quantize_8bit_f32(const float* code, const float* A, float* absmax, uint8_t* out, int bs, long n, hipStream_t s) {
    launch_quantize8<float>(code, A, absmax, out, bs, n, s);
}
void quantize_8bit_f16(const float* code, const void* A, float* absmax, uint8_t* out, int bs, long n, hipStream_t s) {
    launch_quantize8<f16>(code, static_cast<const f16*>(A), absmax, out, bs, n, s);
}
void quantize_8bit_bf16(const float* code, const void* A, float* absmax, uint8_t* out, int bs, long n, hipStream_t s) {
    launch_quantize8<bf16>(code, static_cast<const bf16*>(A), absmax, out, bs, n, s);
}
void dequantize_8bit_f32(const float* code, const uint8_t* A, const float* absmax, float* out, int bs, long n,
                         hipStream_t s) {
    launch_dequantize8<float>(code, A, absmax, out, bs, n, s);
}
void dequantize_8bit_f16(const float* code, const uint8_t* A, const float* absmax, void* out, int bs, long n,
                         hipStream_t s) {
    launch_dequantize8<f16>(code, A, absmax, static_cast<f16*>(out), bs, n, s);
}
void dequantize_8bit_bf16(const float* code, const uint8_t* A, const float* absmax, void* out, int bs, long n,
                          hipStream_t s) {
    launch_dequantize8<bf16>(code, A, absmax, static_cast<bf16*>(out), bs, n, s);
}

} // namespace bnb
