// dequantize4.hip — blockwise NF4/FP4 de-quantization for gfx950 (kernel behind
// cdequantize_blockwise_<T>_{nf4,fp4}).
//
//   out[i] = T( code[nibble_i] * absmax[i / blocksize] )      product in fp32, ONE rounding to T
// — reference csrc/cpu_ops.cpp:304-434 (CPU oracle) and csrc/kernels.cu:465-529 (its GPU kernel).
//
// Pure stream: reads n/2 + 4n/bs bytes, writes n*sizeof(T). The write side dominates, so the lane
// mapping is chosen for the stores: a lane owns 8 consecutive outputs (one 16-byte store for
// fp16/bf16, two for fp32) decoded from one packed dword, i.e. a wavefront writes 1 KiB contiguous
// per store instruction and reads 256 B contiguous per load; UNROLL independent dwords per lane
// keep several loads in flight. The 16-entry code table sits in LDS, one entry per bank, so a
// gather by nibble is conflict-free by construction (equal nibbles broadcast, different nibbles
// hit different banks).
#include "bnb_common.h"

namespace bnb {

namespace {

constexpr int kDqThreads = 256;
constexpr int kDqUnroll = 4;                              // packed dwords per lane
constexpr int kDqTile = kDqThreads * kDqUnroll * 8;       // outputs per workgroup (8192)

template <typename T> __device__ __forceinline__ void store8(T* __restrict__ out, long base, const float (&v)[8]) {
    if constexpr (sizeof(T) == 2) {
        using V = __attribute__((ext_vector_type(8))) T;
        V r;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            r[i] = static_cast<T>(v[i]);
        *reinterpret_cast<V*>(out + base) = r;
    } else {
        using V = __attribute__((ext_vector_type(4))) float;
        V r0, r1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            r0[i] = v[i];
            r1[i] = v[4 + i];
        }
        *reinterpret_cast<V*>(out + base) = r0;
        *reinterpret_cast<V*>(out + base + 4) = r1;
    }
}

template <typename T>
__global__ __launch_bounds__(kDqThreads) void dequantize4_kernel(const uint8_t* __restrict__ A,
                                                                 const float* __restrict__ absmax,
                                                                 T* __restrict__ out, long n, int bs_shift,
                                                                 int quant_type, int vec_ok) {
    __shared__ float code[16];
    const int tid = threadIdx.x;
    if (tid < 16)
        code[tid] = (quant_type == kNF4) ? kNF4Code[tid] : kFP4Code[tid];

    const long tile_base = static_cast<long>(blockIdx.x) * kDqTile;
    const bool full = (tile_base + kDqTile <= n) && vec_ok;

    uint32_t w[kDqUnroll];
    float s[kDqUnroll];
    if (full) {
#pragma unroll
        for (int u = 0; u < kDqUnroll; ++u) {
            const long base = tile_base + (static_cast<long>(u) * kDqThreads + tid) * 8;
            w[u] = *reinterpret_cast<const uint32_t*>(A + (base >> 1));
            s[u] = absmax[base >> bs_shift];
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kDqUnroll; ++u) {
            const long base = tile_base + (static_cast<long>(u) * kDqThreads + tid) * 8;
            float v[8];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t byte = (w[u] >> (8 * b)) & 0xFFu;
                v[2 * b] = rounded_f32(code[byte >> 4] * s[u]);
                v[2 * b + 1] = rounded_f32(code[byte & 0xF] * s[u]);
            }
            store8<T>(out, base, v);
        }
    } else {
        __syncthreads();
        // ragged tile (end of tensor or unaligned pointers): element-pair granularity
#pragma unroll 1
        for (int u = 0; u < kDqUnroll; ++u) {
            const long base = tile_base + (static_cast<long>(u) * kDqThreads + tid) * 8;
#pragma unroll 1
            for (int b = 0; b < 4; ++b) {
                const long e = base + 2 * b;
                if (e >= n)
                    break;
                const uint32_t byte = A[e >> 1];
                out[e] = static_cast<T>(rounded_f32(code[byte >> 4] * absmax[e >> bs_shift]));
                if (e + 1 < n)
                    out[e + 1] = static_cast<T>(rounded_f32(code[byte & 0xF] * absmax[(e + 1) >> bs_shift]));
            }
        }
    }
}

template <typename T>
void launch_dequantize4(const uint8_t* A, const float* absmax, T* out, int blocksize, long n, int quant_type,
                        hipStream_t stream) {
    if (n <= 0)
        return;
    if (!is_pow2(blocksize) || blocksize < 8) {
        fprintf(stderr, "bitsandbytes_amd: dequantize_4bit: unsupported blocksize %d\n", blocksize);
        exit(1);
    }
    const int vec_ok = aligned_to(A, 4) && aligned_to(out, 16);
    const long grid = (n + kDqTile - 1) / kDqTile;
    hipLaunchKernelGGL((dequantize4_kernel<T>), dim3(static_cast<unsigned>(grid)), dim3(kDqThreads), 0, stream, A,
                       absmax, out, n, ilog2(blocksize), quant_type, vec_ok);
    BNB_CHECK_LAUNCH();
}

} // namespace

void dequantize_4bit_f32(const uint8_t* A, const float* absmax, float* out, int blocksize, long n, int qt,
                         hipStream_t s) {
    launch_dequantize4<float>(A, absmax, out, blocksize, n, qt, s);
}
void dequantize_4bit_f16(const uint8_t* A, const float* absmax, void* out, int blocksize, long n, int qt,
                         hipStream_t s) {
    launch_dequantize4<f16>(A, absmax, static_cast<f16*>(out), blocksize, n, qt, s);
}
void dequantize_4bit_bf16(const uint8_t* A, const float* absmax, void* out, int blocksize, long n, int qt,
                          hipStream_t s) {
    launch_dequantize4<bf16>(A, absmax, static_cast<bf16*>(out), blocksize, n, qt, s);
}

} // namespace bnb
