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
// The 64 K-entry table is never materialised and no element pays the division: val(u) is monotone in u,
// so every midpoint i has a threshold bin T_i = min{u : val(u) > mid_i} (found once per workgroup by a
// 16-step search that evaluates val() with exactly the expression above) and q = #{i : T_i <= u} - an
// integer count. A 1024-cell table over u (64 bins per cell) gives the count below the cell and, when the cell
// holds exactly one threshold, its offset inside the cell: an element then costs ONE LDS read and one compare;
// the few cells with several thresholds (the dynamic map is dense only around zero) count them in a short
// loop. Same function of u, bit for bit.
//
// dequantize: out[i] = T(code[A[i]] * absmax[i / blocksize])   (reference csrc/cpu_ops.cpp:436-486).
//
// Layout: a lane owns 4 consecutive elements (one 16-byte fp32 / 8-byte 16-bit load, one 4-byte store of
// codes); a wavefront covers 256 consecutive elements per step; the block max is a DPP/shuffle reduction over
// blocksize/4 lanes, or an accumulation over blocksize/256 steps of one wavefront for the larger blocks.
#include "bnb_common.h"

namespace bnb {

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bin_value(unsigned u) {
    return -1.0f + (2.0f * static_cast<float>(u)) / 65535.0f; // the reference's expression, IEEE division
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
            const f32x4_t r = *reinterpret_cast<const f32x4_t*>(A + i);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                x[j] = r[j];
        } else {
            typedef T v4 __attribute__((ext_vector_type(4)));
            const v4 r = *reinterpret_cast<const v4*>(A + i);
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

// BS = quantization block size (power of two, 64 .. 4096). One wavefront step = 256 elements.
template <typename T, int BS>
__global__ __launch_bounds__(256) void quantize8_kernel(const float* __restrict__ code, const T* __restrict__ A,
                                                        float* __restrict__ absmax, uint8_t* __restrict__ out, long n,
                                                        int vec_ok) {
    __shared__ float mid[256];
    __shared__ uint32_t thr[256];   // T_i, ascending; thr[255] = 65536
    __shared__ uint32_t cell[1024]; // see the table build below
    const int tid = threadIdx.x;
    mid[tid] = (tid < 255) ? 0.5f * (code[tid] + code[tid + 1]) : __builtin_inff();
    __syncthreads();
    {
        // T = min{u in [0, 65536] : bin_value(u) > mid[tid]}  (65536 if none)
        const float m = mid[tid];
        unsigned lo = 0, hi = 65536; // invariant: bin_value(u) <= m for u < lo; bin_value(u) > m for u >= hi
#pragma unroll 1
        for (int step = 0; step < 17 && lo < hi; ++step) {
            const unsigned c = (lo + hi) >> 1;
            if (bin_value(c) > m)
                hi = c;
            else
                lo = c + 1;
        }
        thr[tid] = hi;
    }
    __syncthreads();
    __shared__ uint8_t below_s[1025];
    for (int c = tid; c < 1025; c += 256) {
        // thresholds below bin 64c, by binary search over the ascending thr[0..254]
        const unsigned first = static_cast<unsigned>(c) * 64u;
        int below = 0;
#pragma unroll
        for (int step = 128; step >= 1; step >>= 1)
            below += (below + step - 1 < 255 && thr[below + step - 1] < first) ? step : 0;
        below_s[c] = static_cast<uint8_t>(below);
    }
    __syncthreads();
    for (int c = tid; c < 1024; c += 256) {
        // bits 0-7: thresholds below the cell; bits 8-15: thresholds inside the cell; bits 16-21: offset of the
        // first one inside the cell (what the common single-threshold case compares against)
        const int below = below_s[c];
        const int inside = static_cast<int>(below_s[c + 1]) - below;
        const unsigned off = (inside > 0) ? (thr[below] - static_cast<unsigned>(c) * 64u) : 0u;
        cell[c] = static_cast<uint32_t>(below) | (static_cast<uint32_t>(inside) << 8) | (off << 16);
    }
    __syncthreads();

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
                *reinterpret_cast<uint32_t*>(out + i) = q4;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (i + j < n)
                        out[i + j] = static_cast<uint8_t>(q4 >> (8 * j));
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void dequantize8_kernel(const float* __restrict__ code,
                                                          const uint8_t* __restrict__ A,
                                                          const float* __restrict__ absmax, T* __restrict__ out,
                                                          int bs_shift, long n, int vec_ok) {
    __shared__ float lut[256];
    lut[threadIdx.x] = code[threadIdx.x];
    __syncthreads();
    const long stride = static_cast<long>(gridDim.x) * 256 * 4;
    for (long i = (static_cast<long>(blockIdx.x) * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (vec_ok && i + 4 <= n) {
            const uint32_t q4 = *reinterpret_cast<const uint32_t*>(A + i);
            const float s = absmax[i >> bs_shift]; // blocksize >= 4 and i % 4 == 0: one block for the four
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                v[j] = rounded_f32(lut[(q4 >> (8 * j)) & 0xFFu] * s);
            if constexpr (sizeof(T) == 4) {
                *reinterpret_cast<f32x4_t*>(out + i) = f32x4_t{v[0], v[1], v[2], v[3]};
            } else {
                typedef T v4 __attribute__((ext_vector_type(4)));
                v4 r;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    r[j] = static_cast<T>(v[j]);
                *reinterpret_cast<v4*>(out + i) = r;
            }
        } else {
            for (long e = i; e < i + 4 && e < n; ++e)
                out[e] = static_cast<T>(rounded_f32(lut[A[e]] * absmax[e >> bs_shift]));
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
    long grid = (n + 1023) / 1024;
    if (grid > 8192)
        grid = 8192;
    hipLaunchKernelGGL((dequantize8_kernel<T>), dim3(static_cast<unsigned>(grid)), dim3(256), 0, stream, code, A,
                       absmax, out, ilog2(blocksize), n, vec_ok);
    BNB_CHECK_LAUNCH();
}

} // namespace

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
