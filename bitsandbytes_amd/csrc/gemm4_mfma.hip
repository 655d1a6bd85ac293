// gemm4_mfma.hip — fused 4-bit dequantize + MFMA GEMM for small batches (5 <= M, typically <= 128) on gfx950.
//   out[M, N] = A[M, K] * dequant(B)[N, K]^T (+ bias)
//
// The reference has tensor-core kernels for this only on NVIDIA (csrc/gemm_4bit_sm80.cu:127-457,
// csrc/gemm_4bit_sm75.cu:97-305: mma.sync + ldmatrix + smem-staged dequantized B tiles); on ROCm it
// writes the whole dequantized weight to HBM and calls hipBLASLt
// (bitsandbytes/backends/cuda/ops.py:904-916). This is the missing MFMA kernel, designed for CDNA4
// rather than translated:
//
//  * v_mfma_f32_16x16x32_{bf16,f16}: A operand = activations (row m = lane%16), B operand = weights
//    (column n = lane%16), both hold k = 8*(lane/16) + i. Because the weight column of a lane equals
//    the accumulator column of that lane, the per-(column, block) absmax is a per-lane scalar.
//  * Weights never touch LDS. Lane (n, g) loads the 8 packed bytes holding k = k0 + 16 g .. + 16 of
//    its own row straight into registers (the four lane groups cover one 64-element quantization
//    block), turns them into two B fragments with the same bank-private byte -> bf16x2 table as the
//    gemv kernel (8 conflict-free ds_read_b32), and issues two MFMAs per M-tile. The fp32 partial
//    tile of that 64-k block is then scaled by the lane's fp32 absmax and added to the running
//    accumulator (4 v_fma per tile), so the scale is applied exactly, in fp32, after the MFMA — the
//    same idea as the reference's CDNA SIMT path (csrc/gemm_4bit_simt.cu:436-444), but on the
//    matrix pipe.
//  * Activations are the re-used operand: the workgroup's four wavefronts share one k-range and
//    one XOR-swizzled LDS image of A[MT*16, 256] (16-byte chunk index ^ (row & 15) => conflict-free
//    ds_read_b128 fragment reads), double-buffered, written by all 256 lanes with full-line loads.
//  * Work decomposition = (column group of 64*NT columns) x (K slice). K slices exist only to put
//    >= 256 workgroups on the chip when N is small. Each slice writes its fp32 partial tile to its own
//    slab of a workspace with plain stores; a small finalize kernel adds the slabs in slice order,
//    adds the bias and rounds once. No atomics: results are bit-reproducible run to run (the
//    reference's test_matmul_4bit_weight_orientation demands exact equality between calls).
#include "bnb_common.h"

#include <mutex>
#include <unordered_map>

namespace bnb {

int g_mfma_knob0 = 0; // NT override (0 = heuristic)
int g_mfma_knob1 = 0; // K-slice count override (0 = heuristic)

namespace {

using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <typename T> struct Mma;
template <> struct Mma<bf16> {
    using frag = __attribute__((ext_vector_type(8))) bf16;
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(frag, a), __builtin_bit_cast(frag, b), c, 0,
                                                       0, 0);
    }
    static __device__ __forceinline__ uint32_t pack(float lo, float hi) {
        using V = __attribute__((ext_vector_type(2))) bf16;
        V v;
        v[0] = static_cast<bf16>(lo);
        v[1] = static_cast<bf16>(hi);
        return __builtin_bit_cast(uint32_t, v);
    }
};
template <> struct Mma<f16> {
    using frag = __attribute__((ext_vector_type(8))) f16;
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(frag, a), __builtin_bit_cast(frag, b), c, 0, 0,
                                                      0);
    }
    static __device__ __forceinline__ uint32_t pack(float lo, float hi) {
        using V = __attribute__((ext_vector_type(2))) f16;
        V v;
        v[0] = static_cast<f16>(lo);
        v[1] = static_cast<f16>(hi);
        return __builtin_bit_cast(uint32_t, v);
    }
};

struct GemmArgs {
    const void* A;
    const uint8_t* B;
    const float* absmax;
    const uint8_t* absmax8;
    const float* absmax_code;
    const float* absmax_offset;
    const float* code16;
    void* out;
    const void* bias;
    float* ws; // fp32 [kslices, M, N] partial-sum slabs when kslices > 1
    int M, N, K;
    int bs_shift;
    int quant_type;
    int kslices;     // number of K slices (grid.y)
    int steps_total; // K / 256
};

constexpr int kKC = 256;       // k per pipeline stage (4 quantization blocks of 64)
constexpr int kSteps = kKC / 64;

// LDS image of the A tile for one stage: [rows][256 k] of T, 512 B per row, 16-byte chunks XOR-swizzled by row.
__device__ __forceinline__ int a_lds_off(int row, int chunk) { return row * 512 + ((chunk ^ (row & 15)) << 4); }

template <typename T, int MT, int NT, bool NESTED>
__global__ __launch_bounds__(256) void gemm4_mfma_kernel(const GemmArgs p) {
    // one LDS array: [0, 32K) pair table; then 2 x A stage buffers; then nested code table
    constexpr int kLutBytes = 256 * 32 * 4;
    constexpr int kABytes = MT * 16 * 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* lut = reinterpret_cast<uint32_t*>(smem);
    unsigned char* abuf = smem + kLutBytes;
    float* code2 = reinterpret_cast<float*>(smem + kLutBytes + 2 * kABytes);

    const int tid = threadIdx.x;
    // first vector loads of the kernel (vmcnt retires in order): this lane's two code values
    const gfloat_ptr tbl = (gfloat_ptr)(p.code16 ? p.code16 : (p.quant_type == kNF4 ? kNF4Code : kFP4Code));
    const float code_hi = tbl[tid >> 4];
    const float code_lo = tbl[tid & 15];
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int ln = lane & 15; // column within an n-tile (B operand / accumulator), row within an m-tile (A operand)
    const int lg = lane >> 4; // k group
    const int N = p.N, K = p.K, M = p.M;
    const int m_base = blockIdx.z * (MT * 16);

    // this wavefront's columns
    const int col0 = (blockIdx.x * 4 + wave) * (NT * 16);
    // this workgroup's K slice, in stages of 256 k
    const int per = (p.steps_total + p.kslices - 1) / p.kslices;
    const int st_begin = blockIdx.y * per;
    const int st_end = (st_begin + per < p.steps_total) ? st_begin + per : p.steps_total;
    const int nst = st_end - st_begin;

    const T* __restrict__ A = static_cast<const T*>(p.A);
    const uint8_t* __restrict__ B = p.B;

    long rowoff[NT]; // element offset of this lane's weight row, per n-tile
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int row = col0 + t * 16 + ln;
        row = (row < N) ? row : N - 1;
        rowoff[t] = static_cast<long>(row) * K;
    }

    struct BStage {
        u32x2 w[kSteps][NT];
        float s[kSteps][NT];
    };
    auto load_b = [&](BStage& bs, int st) {
        const int k0 = st * kKC;
#pragma unroll
        for (int u = 0; u < kSteps; ++u) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const long e = rowoff[t] + k0 + u * 64;
                bs.w[u][t] = *reinterpret_cast<const u32x2*>(B + ((e + lg * 16) >> 1));
                const long blk = e >> p.bs_shift;
                if constexpr (NESTED)
                    bs.s[u][t] = __builtin_bit_cast(float, static_cast<uint32_t>(p.absmax8[blk]));
                else
                    bs.s[u][t] = p.absmax[blk];
            }
        }
    };

    // A staging: the tile is MT*16 rows x 32 chunks of 16 B; 256 lanes move MT*2 chunks each.
    constexpr int kAChunks = MT * 2;
    struct AStage {
        u32x4 v[kAChunks];
    };
    auto load_a = [&](AStage& as, int st) {
        const int k0 = st * kKC;
#pragma unroll
        for (int i = 0; i < kAChunks; ++i) {
            const int c = tid + 256 * i;
            const int row = c >> 5, chunk = c & 31;
            const int m = m_base + row;
            const T* src = A + static_cast<long>(m < M ? m : M - 1) * K + k0 + chunk * 8;
            u32x4 v = *reinterpret_cast<const u32x4*>(src);
            as.v[i] = (m < M) ? v : u32x4{0, 0, 0, 0};
        }
    };
    auto store_a = [&](const AStage& as, int buf) {
#pragma unroll
        for (int i = 0; i < kAChunks; ++i) {
            const int c = tid + 256 * i;
            const int row = c >> 5, chunk = c & 31;
            *reinterpret_cast<u32x4*>(abuf + buf * kABytes + a_lds_off(row, chunk)) = as.v[i];
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t)
            acc[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    float offset = 0.0f;
    const uint32_t lane_slot = static_cast<uint32_t>(lane & 31);

    // ---- prologue: first stage in flight, then build the table under its latency
    AStage a_cur;
    BStage b_cur;
    if (nst > 0) {
        load_a(a_cur, st_begin);
        load_b(b_cur, st_begin);
    }
    {
        const uint32_t pr = Mma<T>::pack(code_hi, code_lo);
        const u32x4 v = {pr, pr, pr, pr};
        u32x4* dst = reinterpret_cast<u32x4*>(&lut[tid * 32]);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            dst[j] = v;
        if constexpr (NESTED) {
            code2[tid] = p.absmax_code[tid];
            offset = p.absmax_offset[0];
        }
    }
    if (nst > 0)
        store_a(a_cur, 0);
    __syncthreads();
    const int zsh = opaque_zero();

    for (int it = 0; it < nst; ++it) {
        const int st = st_begin + it;
        const int buf = it & 1;
        AStage a_nxt;
        BStage b_nxt;
        const bool more = (it + 1 < nst);
        if (more) {
            load_a(a_nxt, st + 1); // issued before the B loads so that its wait leaves them in flight
            load_b(b_nxt, st + 1);
        }

        const unsigned char* ab = abuf + buf * kABytes;
#pragma unroll
        for (int u = 0; u < kSteps; ++u) {
            // A fragments of this 64-k block: chunk = u*8 + g*2 + j
            u32x4 af[MT][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    af[mt][j] = *reinterpret_cast<const u32x4*>(ab + a_lds_off(mt * 16 + ln, u * 8 + lg * 2 + j));
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                u32x4 bf[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const uint32_t w = b_cur.w[u][t][j];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        bf[j][q] = lut[(((w >> (8 * q + zsh)) & 0xFFu) << 5) + lane_slot];
                }
                float scale;
                if constexpr (NESTED) {
                    const long blk = (rowoff[t] + static_cast<long>(st) * kKC + u * 64) >> p.bs_shift;
                    const uint32_t q8 = __builtin_bit_cast(uint32_t, b_cur.s[u][t]);
                    scale = __fadd_rn(__fmul_rn(code2[q8], p.absmax[blk >> 8]), offset);
                } else {
                    scale = b_cur.s[u][t];
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    f32x4 part = Mma<T>::run(af[mt][0], bf[0], f32x4{0.f, 0.f, 0.f, 0.f});
                    part = Mma<T>::run(af[mt][1], bf[1], part);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[mt][t][r] = fmaf(scale, part[r], acc[mt][t][r]);
                }
            }
        }

        if (more) {
            store_a(a_nxt, buf ^ 1);
            b_cur = b_nxt;
        }
        __syncthreads();
    }

    // ---- epilogue. accumulator layout: column = lane&15, row = 4*(lane>>4) + r
    T* __restrict__ out = static_cast<T*>(p.out);
    const T* __restrict__ bias = static_cast<const T*>(p.bias);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = col0 + t * 16 + ln;
        if (col >= N)
            continue;
        const float b = (bias && p.kslices == 1) ? static_cast<float>(bias[col]) : 0.0f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m_base + mt * 16 + lg * 4 + r;
                if (m >= M)
                    continue;
                const long o = static_cast<long>(m) * N + col;
                if (p.kslices == 1)
                    out[o] = static_cast<T>(acc[mt][t][r] + b);
                else
                    p.ws[static_cast<long>(blockIdx.y) * M * N + o] = acc[mt][t][r];
            }
        }
    }
}

// out = T(sum_s ws[s] + bias), slabs added in slice order (deterministic)
template <typename T>
__global__ __launch_bounds__(256) void gemm4_finalize_kernel(const float* __restrict__ ws, const T* __restrict__ bias,
                                                             T* __restrict__ out, long total, int N, int kslices) {
    const long i = (static_cast<long>(blockIdx.x) * 256 + threadIdx.x) * 4;
    if (i >= total)
        return;
    if (i + 4 <= total && (total % 4) == 0 && (N % 4) == 0) {
        f32x4 v = *reinterpret_cast<const f32x4*>(ws + i);
        for (int s = 1; s < kslices; ++s) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(ws + static_cast<long>(s) * total + i);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                v[j] += w[j];
        }
        const int col = static_cast<int>(i % N);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float b = bias ? static_cast<float>(bias[col + j]) : 0.0f;
            out[i + j] = static_cast<T>(v[j] + b);
        }
    } else {
        for (long e = i; e < i + 4 && e < total; ++e) {
            float v = ws[e];
            for (int s = 1; s < kslices; ++s)
                v += ws[static_cast<long>(s) * total + e];
            const float b = bias ? static_cast<float>(bias[e % N]) : 0.0f;
            out[e] = static_cast<T>(v + b);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Library-owned split-K workspace, used only when the caller passes none (the reference ABI has no
// workspace argument): one buffer per (device, stream) so concurrent streams never share slabs;
// created on first use; never created while the stream is being captured (hipMalloc would
// invalidate the capture) - in that case the call simply runs with a single K slice. Never freed.
// ---------------------------------------------------------------------------------------------
struct WsKey {
    int dev;
    hipStream_t s;
    bool operator==(const WsKey& o) const { return dev == o.dev && s == o.s; }
};
struct WsKeyHash {
    size_t operator()(const WsKey& k) const {
        return std::hash<const void*>()(k.s) ^ (static_cast<size_t>(k.dev) * 0x9E3779B97F4A7C15ull);
    }
};
struct WsBuf {
    float* p = nullptr;
    size_t bytes = 0;
};
std::mutex g_ws_mu;
std::unordered_map<WsKey, WsBuf, WsKeyHash> g_ws;

float* get_internal_workspace(size_t bytes, hipStream_t stream) {
    int dev = 0;
    BNB_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_ws_mu);
    WsBuf& b = g_ws[WsKey{dev, stream}];
    if (b.bytes < bytes) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
            (void)hipGetLastError();
            return nullptr;
        }
        size_t want = bytes < (size_t(16) << 20) ? (size_t(16) << 20) : bytes;
        float* np = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&np), want) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        b.p = np; // an older, smaller buffer stays allocated: kernels already enqueued may still use it
        b.bytes = want;
    }
    return b.p;
}

struct Plan {
    int mt, nt, ks;
};

// Tile shape and K-slice count for a problem: a pure function of (M, N, K) and the tuning knobs,
// shared by the launch and by the workspace-size query.
Plan make_plan(int M, int N, int K) {
    Plan pl;
    pl.mt = (M > 48) ? 4 : (M > 32) ? 3 : (M > 16) ? 2 : 1;
    int nt = g_mfma_knob0;
    if (nt == 0)
        nt = (pl.mt >= 3) ? 2 : 1;
    if (nt != 1 && nt != 2 && nt != 4)
        nt = 1;
    if ((pl.mt >= 3 && nt == 4))
        nt = 2;
    pl.nt = nt;
    const int steps = K / kKC;
    const int gx = (N + 64 * nt - 1) / (64 * nt);
    const int gz = (M + pl.mt * 16 - 1) / (pl.mt * 16);
    int ks = g_mfma_knob1;
    if (ks == 0)
        ks = (512 + gx * gz - 1) / (gx * gz); // aim for ~2 workgroups per CU
    if (ks > steps)
        ks = steps;
    if (ks < 1)
        ks = 1;
    const int per = (steps + ks - 1) / ks;
    pl.ks = (steps + per - 1) / per; // every slice non-empty
    return pl;
}

template <typename T, int MT, int NT> void launch_mfma(GemmArgs& p, hipStream_t stream) {
    const int cols_per_wg = 64 * NT;
    const int gx = (p.N + cols_per_wg - 1) / cols_per_wg;
    const int gz = (p.M + MT * 16 - 1) / (MT * 16);
    size_t smem = 256 * 32 * 4 + 2 * (MT * 16 * 512) + 1024;
    dim3 grid(gx, p.kslices, gz);
    auto kern = p.absmax8 ? gemm4_mfma_kernel<T, MT, NT, true> : gemm4_mfma_kernel<T, MT, NT, false>;
    static bool attr_set[2] = {false, false};
    if (smem > 64 * 1024 && !attr_set[p.absmax8 ? 1 : 0]) {
        BNB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(smem)));
        attr_set[p.absmax8 ? 1 : 0] = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, stream, p);
}

template <typename T> void dispatch_mfma(GemmArgs& p, float* ws, size_t ws_bytes, hipStream_t stream) {
    Plan pl = make_plan(p.M, p.N, p.K);
    const size_t slab = static_cast<size_t>(p.M) * p.N * sizeof(float);
    if (pl.ks > 1) {
        if (ws == nullptr) {
            ws = get_internal_workspace(slab * pl.ks, stream);
            ws_bytes = ws ? slab * pl.ks : 0;
        }
        if (ws_bytes < slab * pl.ks) {
            // not enough room for all slabs: use as many slices as fit (>= 2), else none
            const int fit = static_cast<int>(ws_bytes / slab);
            const int steps = p.steps_total;
            int ks = fit >= 2 ? fit : 1;
            const int per = (steps + ks - 1) / ks;
            pl.ks = (steps + per - 1) / per;
        }
    }
    p.ws = ws;
    p.kslices = pl.ks;
    const int mt = pl.mt, nt = pl.nt;

#define BNB_MFMA_CASE(MTV, NTV)                                                                    \
    if (mt == MTV && nt == NTV) {                                                                  \
        launch_mfma<T, MTV, NTV>(p, stream);                                                       \
    } else
    BNB_MFMA_CASE(1, 1) BNB_MFMA_CASE(1, 2) BNB_MFMA_CASE(1, 4) BNB_MFMA_CASE(2, 1) BNB_MFMA_CASE(2, 2)
    BNB_MFMA_CASE(2, 4) BNB_MFMA_CASE(3, 1) BNB_MFMA_CASE(3, 2) BNB_MFMA_CASE(4, 1) BNB_MFMA_CASE(4, 2) {
        launch_mfma<T, 1, 1>(p, stream);
    }
#undef BNB_MFMA_CASE
    BNB_CHECK_LAUNCH();

    if (pl.ks > 1) {
        const long total = static_cast<long>(p.M) * p.N;
        const long threads = (total + 3) / 4;
        hipLaunchKernelGGL((gemm4_finalize_kernel<T>), dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256), 0,
                           stream, p.ws, static_cast<const T*>(p.bias), static_cast<T*>(p.out), total, p.N, pl.ks);
        BNB_CHECK_LAUNCH();
    }
}

} // namespace

// Preconditions of the MFMA kernel: 16-bit activations, K a multiple of 256, blocksize >= 64
// (so that a 64-k MFMA pair stays inside one quantization block), 16-byte aligned A, 8-byte aligned B.
bool gemm_4bit_mfma_supported(int dtype, const void* A, const uint8_t* B, int M, int N, int K, int blocksize) {
    return dtype != 0 && M >= 1 && N >= 1 && (K % kKC) == 0 && blocksize >= 64 && is_pow2(blocksize) &&
           aligned_to(A, 16) && aligned_to(B, 8);
}

// Bytes of fp32 slab workspace the launch heuristics would like for this problem (0 = none needed).
size_t gemm_4bit_mfma_workspace_bytes(int M, int N, int K) {
    if (M < 1 || N < 1 || K < kKC)
        return 0;
    const Plan pl = make_plan(M, N, K);
    return pl.ks > 1 ? static_cast<size_t>(pl.ks) * M * N * sizeof(float) : 0;
}

void gemm_4bit_mfma(int dtype, const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax8,
                    const float* absmax_code, const float* absmax_offset, const float* code16, void* out,
                    const void* bias, int M, int N, int K, int blocksize, int quant_type, void* workspace,
                    size_t workspace_bytes, hipStream_t stream) {
    GemmArgs p;
    p.A = A;
    p.B = B;
    p.absmax = absmax;
    p.absmax8 = absmax8;
    p.absmax_code = absmax_code;
    p.absmax_offset = absmax_offset;
    p.code16 = code16;
    p.out = out;
    p.bias = bias;
    p.ws = nullptr;
    p.M = M;
    p.N = N;
    p.K = K;
    p.bs_shift = ilog2(blocksize);
    p.quant_type = quant_type;
    p.kslices = 1;
    p.steps_total = K / kKC;
    if (dtype == 2)
        dispatch_mfma<bf16>(p, static_cast<float*>(workspace), workspace_bytes, stream);
    else
        dispatch_mfma<f16>(p, static_cast<float*>(workspace), workspace_bytes, stream);
}

} // namespace bnb
