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
//     val = -1 + (2*u) / 65535                       (fp32, IEEE division)
//     q   = #{ i < 255 : 0.5*(code[i] + code[i+1]) < val }
// The 64 K-entry table is never materialised here: q is an 8-step binary search over the 255
// midpoints held in LDS — the same function of u, bit for bit.
//
// dequantize: out[i] = T(code[A[i]] * absmax[i / blocksize])   (reference csrc/cpu_ops.cpp:436-486).
//
// These tensors are tiny (n = #4-bit blocks), so the kernels are written for exactness, not for
// bandwidth: one workgroup per 8-bit block on the quantize side.
#include "bnb_common.h"

namespace bnb {

namespace {

template <typename T>
__global__ __launch_bounds__(256) void quantize8_kernel(const float* __restrict__ code, const T* __restrict__ A,
                                                        float* __restrict__ absmax, uint8_t* __restrict__ out,
                                                        int blocksize, long n) {
    __shared__ float mid[256];
    __shared__ float wave_max[4];
    const int tid = threadIdx.x;
    if (tid < 255)
        mid[tid] = 0.5f * (code[tid] + code[tid + 1]);
    else
        mid[255] = __builtin_inff();

    const long start = static_cast<long>(blockIdx.x) * blocksize;
    const long end = (start + blocksize < n) ? start + blocksize : n;

    float m = 0.0f;
    for (long i = start + tid; i < end; i += 256)
        m = fmaxf(m, fabsf(static_cast<float>(A[i])));
    m = group_max<64>(m);
    if ((tid & 63) == 0)
        wave_max[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
    if (tid == 0)
        absmax[blockIdx.x] = m;

    if (m == 0.0f) {
        for (long i = start + tid; i < end; i += 256)
            out[i] = 0;
        return;
    }
    const float inv = 1.0f / m;
    for (long i = start + tid; i < end; i += 256) {
        float v = static_cast<float>(A[i]) * inv;
        v = fminf(fmaxf(v, -1.0f), 1.0f);
        const float t = (v + 1.0f) * 0.5f;
        const float p = __fmul_rn(t, 65535.0f);     // separate rounding, as the un-contracted source reads;
        const float r = __fadd_rn(p, 0.5f);         // (fused and un-fused agree on every tested input)
        const unsigned u = static_cast<unsigned>(r) & 0xFFFFu;
        const float val = -1.0f + (2.0f * static_cast<float>(u)) / 65535.0f;
        // count of midpoints strictly below val, over the first 255 entries (mid[255] = +inf)
        int lo = 0;
#pragma unroll
        for (int step = 128; step >= 1; step >>= 1)
            lo += (mid[lo + step - 1] < val) ? step : 0;
        out[i] = static_cast<uint8_t>(lo);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void dequantize8_kernel(const float* __restrict__ code,
                                                          const uint8_t* __restrict__ A,
                                                          const float* __restrict__ absmax, T* __restrict__ out,
                                                          int bs_shift, long n) {
    __shared__ float lut[256];
    lut[threadIdx.x] = code[threadIdx.x];
    __syncthreads();
    const long stride = static_cast<long>(gridDim.x) * 256;
    for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride)
        out[i] = static_cast<T>(rounded_f32(lut[A[i]] * absmax[i >> bs_shift]));
}

template <typename T>
void launch_quantize8(const float* code, const T* A, float* absmax, uint8_t* out, int blocksize, long n,
                      hipStream_t stream) {
    if (n <= 0)
        return;
    const long nblocks = (n + blocksize - 1) / blocksize;
    hipLaunchKernelGGL((quantize8_kernel<T>), dim3(static_cast<unsigned>(nblocks)), dim3(256), 0, stream, code, A,
                       absmax, out, blocksize, n);
    BNB_CHECK_LAUNCH();
}

template <typename T>
void launch_dequantize8(const float* code, const uint8_t* A, const float* absmax, T* out, int blocksize, long n,
                        hipStream_t stream) {
    if (n <= 0)
        return;
    if (!is_pow2(blocksize)) {
        fprintf(stderr, "bitsandbytes_amd: dequantize_blockwise: blocksize %d is not a power of two\n", blocksize);
        exit(1);
    }
    long grid = (n + 255) / 256;
    if (grid > 4096)
        grid = 4096;
    hipLaunchKernelGGL((dequantize8_kernel<T>), dim3(static_cast<unsigned>(grid)), dim3(256), 0, stream, code, A,
                       absmax, out, ilog2(blocksize), n);
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
