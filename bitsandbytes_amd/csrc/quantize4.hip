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
// Mapping to the machine: the op is a pure HBM stream (reads n*sizeof(T), writes n/2 + 4n/bs).
// One 256-thread workgroup owns a contiguous tile of max(2048, bs) elements; every lane pulls 8
// consecutive elements with one 16-byte (16-bit types) or two 16-byte (fp32) loads, so a wavefront
// reads 1-2 KiB contiguous per instruction; the per-block max is a DPP/shuffle reduction over
// bs/8 lanes (plus one LDS hop when a block spans waves); each lane then emits one packed dword
// (a wavefront writes 256 B contiguous).
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

constexpr uint64_t make_order_word() {
    uint64_t w = 0;
    for (int i = 0; i < 16; ++i)
        w |= static_cast<uint64_t>(kFP4Order[i]) << (4 * i);
    return w;
}
constexpr uint64_t kFP4OrderWord = make_order_word();

// #{bounds < s}: a 4-level binary descent on compile-time constants (v_cmp + v_cndmask on literals).
template <int QT> __device__ __forceinline__ int encode4(float s) {
    constexpr Bounds B = (QT == kNF4) ? kNF4Bounds : kFP4Bounds;
    int pos;
    if (s > B.b[7]) {
        if (s > B.b[11]) {
            if (s > B.b[13])
                pos = (s > B.b[14]) ? 15 : 14;
            else
                pos = (s > B.b[12]) ? 13 : 12;
        } else {
            if (s > B.b[9])
                pos = (s > B.b[10]) ? 11 : 10;
            else
                pos = (s > B.b[8]) ? 9 : 8;
        }
    } else {
        if (s > B.b[3]) {
            if (s > B.b[5])
                pos = (s > B.b[6]) ? 7 : 6;
            else
                pos = (s > B.b[4]) ? 5 : 4;
        } else {
            if (s > B.b[1])
                pos = (s > B.b[2]) ? 3 : 2;
            else
                pos = (s > B.b[0]) ? 1 : 0;
        }
    }
    pos = (s != s) ? 15 : pos; // bucketize sorts NaN last
    if (QT == kNF4)
        return pos;
    return static_cast<int>((kFP4OrderWord >> (4 * pos)) & 0xF);
}

__device__ __forceinline__ float clamp_pm1(float v) {
    // NaN passes through (torch.clamp); written with compares so no fmin/fmax NaN-dropping
    v = (v < -1.0f) ? -1.0f : v;
    v = (v > 1.0f) ? 1.0f : v;
    return v;
}

template <typename T> struct Vec8 {
    float v[8];
};

template <typename T>
__device__ __forceinline__ void load8(const T* __restrict__ A, long base, long n, bool vec_ok, float (&x)[8]) {
    if (vec_ok && base + 8 <= n) {
        if constexpr (sizeof(T) == 2) {
            using V = __attribute__((ext_vector_type(8))) T;
            V r = *reinterpret_cast<const V*>(A + base);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                x[i] = static_cast<float>(r[i]);
        } else {
            using V = __attribute__((ext_vector_type(4))) float;
            V r0 = *reinterpret_cast<const V*>(A + base);
            V r1 = *reinterpret_cast<const V*>(A + base + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                x[i] = r0[i];
                x[4 + i] = r1[i];
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            x[i] = (base + i < n) ? static_cast<float>(A[base + i]) : 0.0f;
    }
}

// One workgroup = 256 threads = TILE elements, TILE = max(2048, BS); CH = TILE/2048 chunks per lane.
template <typename T, int BS, int QT>
__global__ __launch_bounds__(256) void quantize4_kernel(const T* __restrict__ A, float* __restrict__ absmax,
                                                        uint8_t* __restrict__ out, long n, int vec_ok) {
    constexpr int TILE = BS > 2048 ? BS : 2048;
    constexpr int CH = TILE / 2048;
    constexpr int GROUP = (BS < 2048 ? BS : 2048) / 8; // lanes sharing one quant block within a chunk
    __shared__ float wave_max[4];

    const int tid = threadIdx.x;
    const long tile_base = static_cast<long>(blockIdx.x) * TILE;

    float x[CH][8];
    float m = 0.0f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const long base = tile_base + c * 2048 + tid * 8;
        load8<T>(A, base, n, vec_ok != 0, x[c]);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            m = fmaxf(m, fabsf(x[c][i]));
    }

    // block-wide |x| max
    if constexpr (GROUP <= 64) {
        m = group_max<GROUP>(m);
    } else {
        m = group_max<64>(m);
        if ((tid & 63) == 0)
            wave_max[tid >> 6] = m;
        __syncthreads();
        if constexpr (GROUP == 128)
            m = fmaxf(wave_max[(tid >> 7) * 2], wave_max[(tid >> 7) * 2 + 1]);
        else
            m = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
    }

    const long first = tile_base + static_cast<long>(tid) * 8; // chunk 0 position of this lane
    if (first >= n && CH == 1)
        return;

    const long nblocks = (n + BS - 1) / BS;
    const long rem = n % BS;
    const float tiny = 1e-38f;

#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const long base = tile_base + c * 2048 + static_cast<long>(tid) * 8;
        if (base >= n)
            break;
        const long blk = base / BS;
        const bool tail = (rem != 0) && (blk == nblocks - 1);
        float am = m;
        if (tail)
            am = fmaxf(am, tiny);
        if (c == 0 && (base % BS) == 0)
            absmax[blk] = am;

        int q[8];
        if (!tail) {
            const float inv = 1.0f / fmaxf(am, tiny);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                q[i] = encode4<QT>(clamp_pm1(x[c][i] * inv));
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                q[i] = encode4<QT>(clamp_pm1(x[c][i] / am));
        }

        if (base + 8 <= n) {
            uint32_t w = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                w |= static_cast<uint32_t>((q[2 * i] << 4) | q[2 * i + 1]) << (8 * i);
            *reinterpret_cast<uint32_t*>(out + (base >> 1)) = w;
        } else {
            // ragged end: byte stores; an odd n pads the last low nibble with the code of s = 0
            const int pad = encode4<QT>(0.0f);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long e = base + 2 * i;
                if (e < n) {
                    const int lo = (e + 1 < n) ? q[2 * i + 1] : pad;
                    out[e >> 1] = static_cast<uint8_t>((q[2 * i] << 4) | lo);
                }
            }
        }
    }
}

template <typename T, int QT> void launch_quantize4(const T* A, float* absmax, uint8_t* out, int blocksize, long n,
                                                    hipStream_t stream) {
    if (n <= 0)
        return;
    const int vec_ok = aligned_to(A, 16) && aligned_to(out, 4);
#define BNB_Q4_CASE(BS)                                                                            \
    case BS: {                                                                                     \
        constexpr long TILE = BS > 2048 ? BS : 2048;                                               \
        const long grid = (n + TILE - 1) / TILE;                                                   \
        hipLaunchKernelGGL((quantize4_kernel<T, BS, QT>), dim3(static_cast<unsigned>(grid)), dim3(256), 0, stream, \
                           A, absmax, out, n, vec_ok);                                             \
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
    BNB_CHECK_LAUNCH();
}

} // namespace

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
