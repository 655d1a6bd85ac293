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
// Packed dwords per lane = how many workgroups a tensor is (256 threads x U x 8 outputs each). Round 5, measured round-robin
// (profiles/r5_stream_kernels_ab.txt): up to ~17 M elements FEWER, larger workgroups win - U = 8: 4096^2 8.91 -> 8.35 us (64 % of the
// HBM peak; torch's fill of the same output: 7.3), 2048^2 4.47 -> 3.99 - from 33 M elements on U = 4 wins (8192^2 30.3 vs 31.4);
// U = 2 and 16 lose on one side or the other. (The reverse was expected - more rounds of workgroups to overlap loads and stores.)
constexpr long kDqBigElements = 1L << 25;
// sweeps / tests (bnb_mi355x_set_tuning reserved0 = 10 + v): v = 1, 2, 4, 8, 16: packed dwords per lane of the general kernel forced;
// v = 22 / 24 / 28: the line-contiguous fp32 kernel with 2 / 4 / 8 units per lane; v = 30: the general kernel for fp32 outputs too
// (round 4's form); 0 = built-in
thread_local TlsKnob g_dq_variant{0};

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

// Double-quantised statistics (NESTED; round 5): the scale of block b is reconstructed in the kernel,
//     absmax[b] = code2[absmax8[b]] * absmax2[b >> 8] + offset        (fp32 product, rounded, then fp32 sum - NOT one fma:
// the product passes through rounded_f32; the first build, written with __fmul_rn / __fadd_rn, compiled to v_fma_f32 and differed
// from the sequence in the last bit of rare fp16 outputs)
// which is, bit for bit, what the reference's two extra operator calls produce (dequantize_blockwise with blocksize 256, then
// `absmax += offset`: bitsandbytes/functional.py:1002-1006) - one launch instead of three and no fp32 absmax vector in HBM.
struct NestedStats {
    const uint8_t* absmax8; // one code per 4-bit block
    const float* code2;     // 256 entries (device)
    const float* offset;    // 1 float (device)
};

template <typename T, int kDqUnroll, bool NESTED = false>
__global__ __launch_bounds__(kDqThreads) void dequantize4_kernel(const uint8_t* __restrict__ A,
                                                                 const float* __restrict__ absmax,
                                                                 T* __restrict__ out, long n, int bs_shift,
                                                                 int quant_type, int vec_ok, NestedStats nested = {}) {
    constexpr int kDqTile = kDqThreads * kDqUnroll * 8; // outputs per workgroup
    __shared__ float code[16];
    __shared__ float code2[NESTED ? 256 : 1];
    const int tid = threadIdx.x;
    if (tid < 16)
        code[tid] = (quant_type == kNF4) ? kNF4Code[tid] : kFP4Code[tid];
    float offset = 0.0f;
    if constexpr (NESTED) {
        code2[tid] = nested.code2[tid];
        offset = *nested.offset;
    }
    // scale of block b; NESTED: `absmax` is the second-level vector (one float per 256 blocks)
    auto scale_of = [&](long b, uint32_t q8) -> float {
        if constexpr (NESTED)
            return rounded_f32(code2[q8] * absmax[b >> 8]) + offset; // (rounded_f32: hipcc contracts __fadd_rn(__fmul_rn()) into ONE v_fma_f32)
        else
            return absmax[b];
    };

    const long tile_base = static_cast<long>(blockIdx.x) * kDqTile;
    const bool full = (tile_base + kDqTile <= n) && vec_ok;

    uint32_t w[kDqUnroll];
    float s[kDqUnroll];
    [[maybe_unused]] uint32_t q8[kDqUnroll];
    if (full) {
#pragma unroll
        for (int u = 0; u < kDqUnroll; ++u) {
            const long base = tile_base + (static_cast<long>(u) * kDqThreads + tid) * 8;
            w[u] = stream_load<sizeof(T) == 2>(reinterpret_cast<const uint32_t*>(A + (base >> 1)));
            if constexpr (NESTED) {
                q8[u] = nested.absmax8[base >> bs_shift];
                s[u] = absmax[(base >> bs_shift) >> 8];
            } else {
                s[u] = absmax[base >> bs_shift];
            }
        }
        __syncthreads();
        if constexpr (NESTED) {
#pragma unroll
            for (int u = 0; u < kDqUnroll; ++u)
                s[u] = rounded_f32(code2[q8[u]] * s[u]) + offset;
        }
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
                const long b0 = e >> bs_shift;
                out[e] = static_cast<T>(rounded_f32(code[byte >> 4] * scale_of(b0, NESTED ? nested.absmax8[b0] : 0u)));
                if (e + 1 < n) {
                    const long b1 = (e + 1) >> bs_shift;
                    out[e + 1] = static_cast<T>(rounded_f32(code[byte & 0xF] * scale_of(b1, NESTED ? nested.absmax8[b1] : 0u)));
                }
            }
        }
    }
}

// fp32 outputs with LINE-CONTIGUOUS stores (round 5): in the kernel above a lane owns 8 outputs = two 16-byte stores at a 32-byte
// stride - every store instruction of a wavefront writes HALF of each 128-byte line it touches (the non-temporal policy measured
// +33 % there, the default one leaves the halves to the L2: 20.2 us at 4096^2 = 47 % of HBM peak). Here a lane owns 4 outputs per
// unit - one packed 16-bit piece in, ONE 16-byte store out - so a wavefront's store is 1 KiB contiguous, eight full lines; U units
// per lane keep the loads in flight. Same arithmetic, same table: bit-identical.
template <int U>
__global__ __launch_bounds__(kDqThreads) void dequantize4_f32_lines_kernel(const uint8_t* __restrict__ A, const float* __restrict__ absmax,
                                                                            float* __restrict__ out, long n, int bs_shift, int quant_type) {
    constexpr int kTile = kDqThreads * U * 4; // outputs per workgroup
    __shared__ float code[16];
    const int tid = threadIdx.x;
    if (tid < 16)
        code[tid] = (quant_type == kNF4) ? kNF4Code[tid] : kFP4Code[tid];
    const long tile_base = static_cast<long>(blockIdx.x) * kTile;
    unsigned short w[U];
    float s[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long base = tile_base + (static_cast<long>(u) * kDqThreads + tid) * 4;
        w[u] = stream_load<false>(reinterpret_cast<const unsigned short*>(A + (base >> 1)));
        s[u] = absmax[base >> bs_shift];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long base = tile_base + (static_cast<long>(u) * kDqThreads + tid) * 4;
        using V = __attribute__((ext_vector_type(4))) float;
        V r;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const uint32_t byte = (static_cast<uint32_t>(w[u]) >> (8 * b)) & 0xFFu;
            r[2 * b] = rounded_f32(code[byte >> 4] * s[u]);
            r[2 * b + 1] = rounded_f32(code[byte & 0xF] * s[u]);
        }
        stream_store<true>(r, reinterpret_cast<V*>(out + base));
    }
}

template <typename T, int U>
void launch_dequantize4_u(const uint8_t* A, const float* absmax, T* out, int blocksize, long n, int quant_type, int vec_ok, hipStream_t stream) {
    constexpr long tile = static_cast<long>(kDqThreads) * U * 8;
    const long grid = (n + tile - 1) / tile;
    hipLaunchKernelGGL((dequantize4_kernel<T, U>), dim3(static_cast<unsigned>(grid)), dim3(kDqThreads), 0, stream, A, absmax, out, n,
                       ilog2(blocksize), quant_type, vec_ok);
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
    const int variant = g_dq_variant.load(std::memory_order_relaxed);
    if constexpr (sizeof(T) == 4) {
        // the line-contiguous form (built-in: 4 units per lane - 12.5 vs 19.3 us at 4096^2, 44.7 vs 72.2 at 8192^2) needs whole tiles
        // and aligned pointers; everything else (ragged sizes) keeps the general kernel
        const int u = variant == 0 ? 4 : (variant >= 20 && variant < 30) ? variant - 20 : 0;
        if (u == 2 || u == 4 || u == 8) {
            const long tile = static_cast<long>(kDqThreads) * u * 4;
            if (vec_ok && aligned_to(A, 2) && n % tile == 0 && blocksize >= 4) {
                if (u == 2)
                    hipLaunchKernelGGL((dequantize4_f32_lines_kernel<2>), dim3(static_cast<unsigned>(n / tile)), dim3(kDqThreads), 0, stream, A, absmax, out, n, ilog2(blocksize), quant_type);
                else if (u == 4)
                    hipLaunchKernelGGL((dequantize4_f32_lines_kernel<4>), dim3(static_cast<unsigned>(n / tile)), dim3(kDqThreads), 0, stream, A, absmax, out, n, ilog2(blocksize), quant_type);
                else
                    hipLaunchKernelGGL((dequantize4_f32_lines_kernel<8>), dim3(static_cast<unsigned>(n / tile)), dim3(kDqThreads), 0, stream, A, absmax, out, n, ilog2(blocksize), quant_type);
                BNB_CHECK_LAUNCH();
                return;
            }
        }
    }
    switch (variant) {
    case 1: launch_dequantize4_u<T, 1>(A, absmax, out, blocksize, n, quant_type, vec_ok, stream); break;
    case 2: launch_dequantize4_u<T, 2>(A, absmax, out, blocksize, n, quant_type, vec_ok, stream); break;
    case 8: launch_dequantize4_u<T, 8>(A, absmax, out, blocksize, n, quant_type, vec_ok, stream); break;
    case 16: launch_dequantize4_u<T, 16>(A, absmax, out, blocksize, n, quant_type, vec_ok, stream); break;
    case 4:
    case 30: launch_dequantize4_u<T, 4>(A, absmax, out, blocksize, n, quant_type, vec_ok, stream); break;
    default:
        if (n < kDqBigElements)
            launch_dequantize4_u<T, 8>(A, absmax, out, blocksize, n, quant_type, vec_ok, stream);
        else
            launch_dequantize4_u<T, 4>(A, absmax, out, blocksize, n, quant_type, vec_ok, stream);
        break;
    }
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
    const bool valid = row >= 0 && row < num_rows;
    if constexpr (sizeof(T) == 4) {
        // fp32 outputs (round 5): two units of 4 outputs per lane - one packed 16-bit piece in, ONE 16-byte store out, a wavefront's
        // store 1 KiB contiguous - instead of 8 outputs = two 16-byte stores at a 32-byte stride (half lines per store instruction:
        // what cost the standalone kernel a third of its bandwidth, see dequantize4_f32_lines_kernel). Same arithmetic, same values.
        unsigned short w2[2] = {0, 0};
        float s2[2] = {0.0f, 0.0f};
        int col2[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            col2[u] = static_cast<int>(blockIdx.x) * kDqThreads * 8 + u * kDqThreads * 4 + tid * 4;
            if (col2[u] < row_len && valid) {
                const long e = row * row_len + col2[u];
                w2[u] = *reinterpret_cast<const unsigned short*>(A + (e >> 1));
                s2[u] = absmax[e >> bs_shift];
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (col2[u] >= row_len)
                continue;
            using V = __attribute__((ext_vector_type(4))) float;
            V r;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const uint32_t byte = (static_cast<uint32_t>(w2[u]) >> (8 * b)) & 0xFFu;
                r[2 * b] = valid ? rounded_f32(code[byte >> 4] * s2[u]) : __builtin_nanf("");
                r[2 * b + 1] = valid ? rounded_f32(code[byte & 0xF] * s2[u]) : __builtin_nanf("");
            }
            *reinterpret_cast<V*>(out + t * row_len + col2[u]) = r;
        }
        return;
    }
    const int col = (static_cast<int>(blockIdx.x) * kDqThreads + tid) * 8;
    const bool live = col < row_len;
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

template <typename T>
void launch_dequantize4_nested(const uint8_t* A, const uint8_t* absmax8, const float* absmax2, const float* code2, const float* offset,
                               T* out, int blocksize, long n, int quant_type, hipStream_t stream) {
    if (n <= 0)
        return;
    if (!is_pow2(blocksize) || blocksize < 8) {
        fprintf(stderr, "bitsandbytes_amd: dequantize_4bit (nested): unsupported blocksize %d\n", blocksize);
        exit(1);
    }
    const int vec_ok = aligned_to(A, 4) && aligned_to(out, 16);
    const NestedStats ns{absmax8, code2, offset};
    // the general kernel's two tile sizes, chosen as for plain statistics (launch_dequantize4)
    if (n < kDqBigElements) {
        constexpr long tile = static_cast<long>(kDqThreads) * 8 * 8;
        hipLaunchKernelGGL((dequantize4_kernel<T, 8, true>), dim3(static_cast<unsigned>((n + tile - 1) / tile)), dim3(kDqThreads), 0, stream, A,
                           absmax2, out, n, ilog2(blocksize), quant_type, vec_ok, ns);
    } else {
        constexpr long tile = static_cast<long>(kDqThreads) * 4 * 8;
        hipLaunchKernelGGL((dequantize4_kernel<T, 4, true>), dim3(static_cast<unsigned>((n + tile - 1) / tile)), dim3(kDqThreads), 0, stream, A,
                           absmax2, out, n, ilog2(blocksize), quant_type, vec_ok, ns);
    }
    BNB_CHECK_LAUNCH();
}

void dequantize_4bit_set_variant(int variant) { g_dq_variant.store(variant, std::memory_order_relaxed); }

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

void dequantize_4bit_nested(int dtype, const uint8_t* A, const uint8_t* absmax8, const float* absmax2, const float* code2,
                            const float* offset, void* out, int blocksize, long n, int qt, hipStream_t stream) {
    if (dtype == 0)
        launch_dequantize4_nested<float>(A, absmax8, absmax2, code2, offset, static_cast<float*>(out), blocksize, n, qt, stream);
    else if (dtype == 1)
        launch_dequantize4_nested<f16>(A, absmax8, absmax2, code2, offset, static_cast<f16*>(out), blocksize, n, qt, stream);
    else
        launch_dequantize4_nested<bf16>(A, absmax8, absmax2, code2, offset, static_cast<bf16*>(out), blocksize, n, qt, stream);
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
