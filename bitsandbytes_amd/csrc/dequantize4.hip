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

// NT: non-temporal stores (the standalone stream; the row gather keeps the default policy - its output is read next)
template <typename T, bool NT> __device__ __forceinline__ void store8(T* __restrict__ out, long base, const float (&v)[8]) {
    if constexpr (sizeof(T) == 2) {
        using V = __attribute__((ext_vector_type(8))) T;
        V r;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            r[i] = static_cast<T>(v[i]);
        stream_store<NT>(r, reinterpret_cast<V*>(out + base));
    } else {
        using V = __attribute__((ext_vector_type(4))) float;
        V r0, r1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            r0[i] = v[i];
            r1[i] = v[4 + i];
        }
        stream_store<NT>(r0, reinterpret_cast<V*>(out + base));
        stream_store<NT>(r1, reinterpret_cast<V*>(out + base + 4));
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
            w[u] = stream_load<sizeof(T) == 2>(reinterpret_cast<const uint32_t*>(A + (base >> 1)));
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
            store8<T, sizeof(T) == 2>(out, base, v);
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

// Row gather + dequantize in one pass (the Embedding4bit lookup, reference nn/modules.py:921-951, which
// runs two F.embedding gathers and a dequantize): out[t, :] = T(code[nib] * absmax) of weight row idx[t].
// Requires row_len % 8 == 0 and row_len % blocksize == 0, so a row is a whole number of packed dwords and
// of quantization blocks. grid = (ceil(row_len / 2048), rows_out); a lane owns 8 outputs exactly as above,
// so results are bit-identical to dequantize4_kernel on the gathered bytes. An index outside
// [0, num_rows) yields a row of NaN: the library has no error return and a device-side assert (what F.embedding does)
// would take the process down mid-graph, but a silent row of zeros would turn a bad token id into a plausible
// embedding - NaN propagates to the first consumer that looks.
template <typename T, typename IdxT>
__global__ __launch_bounds__(kDqThreads) void dequantize4_rows_kernel(const uint8_t* __restrict__ A,
                                                                      const float* __restrict__ absmax,
                                                                      const IdxT* __restrict__ idx,
                                                                      T* __restrict__ out, long num_rows, int row_len,
                                                                      int bs_shift, int quant_type) {
    __shared__ float code[16];
    const int tid = threadIdx.x;
    if (tid < 16)
        code[tid] = (quant_type == kNF4) ? kNF4Code[tid] : kFP4Code[tid];
    const long t = blockIdx.y;
    const long row = static_cast<long>(idx[t]);
    const int col = (static_cast<int>(blockIdx.x) * kDqThreads + tid) * 8;
    const bool live = col < row_len;
    const bool valid = row >= 0 && row < num_rows;
    uint32_t w = 0;
    float s = 0.0f;
    if (live && valid) {
        const long e = row * row_len + col;
        w = *reinterpret_cast<const uint32_t*>(A + (e >> 1));
        s = absmax[e >> bs_shift];
    }
    __syncthreads();
    if (!live)
        return;
    float v[8];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const uint32_t byte = (w >> (8 * b)) & 0xFFu;
        v[2 * b] = valid ? rounded_f32(code[byte >> 4] * s) : __builtin_nanf("");
        v[2 * b + 1] = valid ? rounded_f32(code[byte & 0xF] * s) : __builtin_nanf("");
    }
    store8<T, false>(out, t * row_len + col, v);
}

template <typename T>
void launch_dequantize4_rows(const uint8_t* A, const float* absmax, const void* idx, int index_bytes, T* out,
                             long rows_out, long num_rows, int row_len, int blocksize, int quant_type,
                             hipStream_t stream) {
    if (rows_out <= 0 || row_len <= 0)
        return;
    if (!is_pow2(blocksize) || blocksize < 8 || row_len % 8 != 0 || row_len % blocksize != 0 ||
        (index_bytes != 4 && index_bytes != 8) || !aligned_to(A, 4) || !aligned_to(out, 16) || rows_out > 0x7FFFFFFFL) {
        fprintf(stderr,
                "bitsandbytes_amd: dequantize_4bit_rows: need row_len %% 8 == 0, row_len %% blocksize == 0, "
                "int32/int64 indices, 4-byte aligned weights and 16-byte aligned output (row_len %d, blocksize %d)\n",
                row_len, blocksize);
        exit(1);
    }
    const dim3 grid(static_cast<unsigned>((row_len + kDqThreads * 8 - 1) / (kDqThreads * 8)), 1, 1);
    // grid.y is limited to 65535: launch in slabs of rows
    for (long r0 = 0; r0 < rows_out; r0 += 65535) {
        const unsigned ny = static_cast<unsigned>(rows_out - r0 < 65535 ? rows_out - r0 : 65535);
        if (index_bytes == 8)
            hipLaunchKernelGGL((dequantize4_rows_kernel<T, int64_t>), dim3(grid.x, ny), dim3(kDqThreads), 0, stream, A,
                               absmax, static_cast<const int64_t*>(idx) + r0, out + r0 * row_len, num_rows, row_len,
                               ilog2(blocksize), quant_type);
        else
            hipLaunchKernelGGL((dequantize4_rows_kernel<T, int32_t>), dim3(grid.x, ny), dim3(kDqThreads), 0, stream, A,
                               absmax, static_cast<const int32_t*>(idx) + r0, out + r0 * row_len, num_rows, row_len,
                               ilog2(blocksize), quant_type);
    }
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

void dequantize_4bit_rows(int dtype, const uint8_t* A, const float* absmax, const void* idx, int index_bytes, void* out,
                          long rows_out, long num_rows, int row_len, int blocksize, int qt, hipStream_t s) {
    if (dtype == 0)
        launch_dequantize4_rows<float>(A, absmax, idx, index_bytes, static_cast<float*>(out), rows_out, num_rows,
                                       row_len, blocksize, qt, s);
    else if (dtype == 1)
        launch_dequantize4_rows<f16>(A, absmax, idx, index_bytes, static_cast<f16*>(out), rows_out, num_rows, row_len,
                                     blocksize, qt, s);
    else
        launch_dequantize4_rows<bf16>(A, absmax, idx, index_bytes, static_cast<bf16*>(out), rows_out, num_rows,
                                      row_len, blocksize, qt, s);
}

} // namespace bnb
