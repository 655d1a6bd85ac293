// c_api.hip — the extern "C" surface of libbitsandbytes_mi355x.so (declared in include/bnb_mi355x.h).
// Thin, un-mangled wrappers over the launchers in quantize4.hip, dequantize4.hip, blockwise8.hip,
// gemv4_stream.hip, gemm4_mfma.hip and gemm4_mfma_rt.hip. Mirrors the symbol set the reference exports for this path from
// csrc/pythonInterface.cpp:343-841 and csrc/gemm_4bit.cu:136-168.
#include "bnb_common.h"

#include "../../include/bnb_mi355x.h"

namespace bnb {
// quantize4.hip
void quantize_4bit_f32(const float*, float*, uint8_t*, int, long, int, hipStream_t);
void quantize_4bit_f16(const void*, float*, uint8_t*, int, long, int, hipStream_t);
void quantize_4bit_bf16(const void*, float*, uint8_t*, int, long, int, hipStream_t);
// dequantize4.hip
void dequantize_4bit_f32(const uint8_t*, const float*, float*, int, long, int, hipStream_t);
void dequantize_4bit_rows(int, const uint8_t*, const float*, const void*, int, void*, long, long, int, int, int, hipStream_t);
void dequantize_4bit_f16(const uint8_t*, const float*, void*, int, long, int, hipStream_t);
void dequantize_4bit_bf16(const uint8_t*, const float*, void*, int, long, int, hipStream_t);
// blockwise8.hip
void quantize_8bit_f32(const float*, const float*, float*, uint8_t*, int, long, hipStream_t);
void quantize_8bit_f16(const float*, const void*, float*, uint8_t*, int, long, hipStream_t);
void quantize_8bit_bf16(const float*, const void*, float*, uint8_t*, int, long, hipStream_t);
void dequantize_8bit_f32(const float*, const uint8_t*, const float*, float*, int, long, hipStream_t);
void dequantize_8bit_f16(const float*, const uint8_t*, const float*, void*, int, long, hipStream_t);
void dequantize_8bit_bf16(const float*, const uint8_t*, const float*, void*, int, long, hipStream_t);
// gemv4_stream.hip
void gemv_4bit_stream(int dtype, const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax8,
                      const float* absmax_code, const float* absmax_offset, const float* code16, void* out,
                      const void* bias, int M, int N, int K, int blocksize, int quant_type, hipStream_t stream);
bool gemv_4bit_grouped(int dtype, const void* A, int count, const uint8_t* const* B, const float* const* absmax,
                       const uint8_t* const* absmax8, const float* const* absmax_code, const float* const* absmax_offset,
                       void* const* out, const void* const* bias, const int* N, int M, int K, int blocksize, int quant_type,
                       hipStream_t stream);
void gemv_4bit_stream_tuning(int ns, int sw, int rows_per_wg, int nt, int waves);
bool gemv_4bit_peer(void* const* bufs, void* epoch_word, int world, int rank, int dtype, const void* A, const uint8_t* B, const float* absmax,
                    const uint8_t* absmax8, const float* absmax_code, const float* absmax_offset, const void* bias, void* out_local,
                    int ns, int K, int blocksize, int quant_type, int mode, long max_values, int wg_limit, uint32_t epoch_offset,
                    uint32_t spin_bound, hipStream_t stream);
void peer_chain_read(void* const* bufs, void* epoch_word, int world, int rank, int dtype, void* out, int nvalues, long max_values, uint32_t epoch_offset,
                     uint32_t spin_bound, hipStream_t stream);
bool gemv_4bit_peer_serves(int world, int ns, int K, int blocksize, int mode, long max_values, int wg_limit);
thread_local int g_last_gemm_kernel = kKernelNone;
#ifdef BNB_PROFILING
unsigned long long* g_dbg_buf = nullptr; // profiling builds only: device buffer for the kernels' s_memtime stamps
#endif
// gemm4_mfma.hip
bool gemm_4bit_mfma_supported(int dtype, const void* A, const uint8_t* B, const float* code16, int M, int N, int K, int blocksize, bool plain_absmax);
void gemm_4bit_mfma(int dtype, const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax8,
                    const float* absmax_code, const float* absmax_offset, const float* code16, void* out,
                    const void* bias, int M, int N, int K, int blocksize, int quant_type, void* workspace,
                    size_t workspace_bytes, hipStream_t stream);
size_t gemm_4bit_mfma_workspace_bytes(int M, int N, int K, int blocksize);
bool gemm_4bit_sm_routes(int dtype, int M, int N, int K, int blocksize);
bool gemm_4bit_sm_supported(int dtype, const void* A, const uint8_t* B, const float* code16, int M, int N, int K, int blocksize);
bool gemm_4bit_sm_grouped(int dtype, const void* A, int count, const uint8_t* const* B, const float* const* absmax,
                          const uint8_t* const* absmax8, const float* const* absmax_code, const float* const* absmax_offset,
                          void* const* out, const void* const* bias, const int* N, int M, int K, int blocksize, int quant_type,
                          hipStream_t stream);
extern thread_local TlsKnob g_mfma_knob0, g_mfma_knob1;
// gemm4_grad_input.hip
bool gemm_4bit_grad_input_supported(int dtype, const void* G, const uint8_t* B, int M, int N, int K, int blocksize);
size_t gemm_4bit_grad_input_workspace_bytes(int M, int N, int K);
void quantize_8bit_set_variant(int variant);
void quantize_absmax_nested(const float* code, const float* absmax, long n, float* partial, float* offset_out, uint8_t* out, float* absmax2, hipStream_t stream);
void quantize_4bit_set_variant(int variant);
void dequantize_4bit_set_variant(int variant);
void dequantize_4bit_nested(int, const uint8_t*, const uint8_t*, const float*, const float*, const float*, void*, int, long, int, hipStream_t);
void gemm_4bit_grad_input_set_slices(int ns);
void gemm_4bit_grad_input(int dtype, const void* G, const uint8_t* B, const float* absmax, const uint8_t* absmax8,
                          const float* absmax_code, const float* absmax_offset, void* out, int M, int N, int K, int blocksize,
                          int quant_type, void* workspace, size_t workspace_bytes, hipStream_t stream);

namespace {

// M at or below which the streaming dot kernel is used (its activations live in registers: 32 fp32 per row and
// lane); above it the MFMA kernels (when supported). MI355X-specific replacement for the reference's per-arch
// heuristic (bitsandbytes/backends/cuda/ops.py:814-843), calibrated on gfx950 — see DESIGN.md.
constexpr int kStreamMaxM = 4;

// plain_absmax: the call carries fp32 absmax (not the double-quantised form) - what the MFMA route needs to know at blocksize 32,
// where only the register-transposed kernel's BS32 instances (fp32 absmax) exist; the shape-only queries pass true.
bool route_to_mfma(int kernel, int dtype, const void* A, const uint8_t* B, const float* code16, int M, int N, int K, int blocksize, bool plain_absmax) {
    if (kernel == 1 || kernel == 3)
        return false;
    // the streaming MFMA kernel's own reach: 2 ... 16 rows on matrices of >= 128 rows (csrc/gemm4_mfma.hip: sm_selected has the
    // measured exceptions), any K % 64 == 0 (the other MFMA kernels need whole 256-k chunks) - round 6
    const bool sm = gemm_4bit_sm_routes(dtype, M, N, K, blocksize) && gemm_4bit_sm_supported(dtype, A, B, code16, M, N, K, blocksize);
    if (kernel == 2)
        return sm || gemm_4bit_mfma_supported(dtype, A, B, code16, M, N, K, blocksize, plain_absmax);
    // three or four rows: the streaming kernel's FMA count grows with M while the MFMA kernel's does not - on matrices that
    // fill the chip with 16-column workgroups the MFMA kernel is ahead from M = 3 (4096^2 6.5 vs 6.9 us, 4096 x 11008 10.8 vs
    // 26.7), on small ones the streaming kernel's cheaper launch still wins (1376 x 4096 4.9 vs 5.3)
    // round 6: from TWO rows on wherever the streaming MFMA kernel (gemm4_mfma_sm.hip) serves the shape - one decode for all rows,
    // activations once per CU (4096^2 M = 2: 4.44 vs 4.95 us; profiles/r6_sm_v3_ab_full.txt)
    const bool big = static_cast<long>(N) * K >= (12L << 20);
    return sm || ((M > kStreamMaxM || (M >= 3 && big)) && gemm_4bit_mfma_supported(dtype, A, B, code16, M, N, K, blocksize, plain_absmax));
}

void gemm_4bit_dispatch(int kernel, int dtype, const void* A, const uint8_t* B, const float* absmax,
                        const uint8_t* absmax8, const float* absmax_code, const float* absmax_offset,
                        const float* code16, void* out, const void* bias, int M, int N, int K, int blocksize,
                        int quant_type, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (M <= 0 || N <= 0)
        return;
    if (quant_type != kFP4 && quant_type != kNF4) {
        fprintf(stderr, "bitsandbytes_amd: gemm_4bit: quant_type must be 1 (FP4) or 2 (NF4), got %d\n", quant_type);
        exit(1);
    }
    if (route_to_mfma(kernel, dtype, A, B, code16, M, N, K, blocksize, absmax8 == nullptr && aligned_to(absmax, 16)))
        gemm_4bit_mfma(dtype, A, B, absmax, absmax8, absmax_code, absmax_offset, code16, out, bias, M, N, K, blocksize,
                       quant_type, workspace, workspace_bytes, stream);
    else
        gemv_4bit_stream(dtype, A, B, absmax, absmax8, absmax_code, absmax_offset, code16, out, bias, M, N, K, blocksize,
                         quant_type, stream);
}

} // namespace
} // namespace bnb

using namespace bnb;

static inline hipStream_t S(bnb_stream_t s) { return static_cast<hipStream_t>(s); }

extern "C" {
// the library is built with -fvisibility=hidden: the C ABI declared in include/bnb_mi355x.h is ALL it exports
#pragma GCC visibility push(default)

// ------------------------------------------------------------------ 4-bit quantize (NULL stream, like the reference)
void cquantize_blockwise_fp32_nf4(float*, float* A, float* absmax, unsigned char* out, int bs, const int n) {
    quantize_4bit_f32(A, absmax, out, bs, n, kNF4, nullptr);
}
void cquantize_blockwise_fp32_fp4(float*, float* A, float* absmax, unsigned char* out, int bs, const int n) {
    quantize_4bit_f32(A, absmax, out, bs, n, kFP4, nullptr);
}
void cquantize_blockwise_fp16_nf4(float*, void* A, float* absmax, unsigned char* out, int bs, const int n) {
    quantize_4bit_f16(A, absmax, out, bs, n, kNF4, nullptr);
}
void cquantize_blockwise_fp16_fp4(float*, void* A, float* absmax, unsigned char* out, int bs, const int n) {
    quantize_4bit_f16(A, absmax, out, bs, n, kFP4, nullptr);
}
void cquantize_blockwise_bf16_nf4(float*, void* A, float* absmax, unsigned char* out, int bs, const int n) {
    quantize_4bit_bf16(A, absmax, out, bs, n, kNF4, nullptr);
}
void cquantize_blockwise_bf16_fp4(float*, void* A, float* absmax, unsigned char* out, int bs, const int n) {
    quantize_4bit_bf16(A, absmax, out, bs, n, kFP4, nullptr);
}

// ------------------------------------------------------------------ 4-bit dequantize
void cdequantize_blockwise_fp32_nf4(float*, unsigned char* A, float* absmax, float* out, int bs, const int n,
                                    bnb_stream_t s) {
    dequantize_4bit_f32(A, absmax, out, bs, n, kNF4, S(s));
}
void cdequantize_blockwise_fp32_fp4(float*, unsigned char* A, float* absmax, float* out, int bs, const int n,
                                    bnb_stream_t s) {
    dequantize_4bit_f32(A, absmax, out, bs, n, kFP4, S(s));
}
void cdequantize_blockwise_fp16_nf4(float*, unsigned char* A, float* absmax, void* out, int bs, const int n,
                                    bnb_stream_t s) {
    dequantize_4bit_f16(A, absmax, out, bs, n, kNF4, S(s));
}
void cdequantize_blockwise_fp16_fp4(float*, unsigned char* A, float* absmax, void* out, int bs, const int n,
                                    bnb_stream_t s) {
    dequantize_4bit_f16(A, absmax, out, bs, n, kFP4, S(s));
}
void cdequantize_blockwise_bf16_nf4(float*, unsigned char* A, float* absmax, void* out, int bs, const int n,
                                    bnb_stream_t s) {
    dequantize_4bit_bf16(A, absmax, out, bs, n, kNF4, S(s));
}
void cdequantize_blockwise_bf16_fp4(float*, unsigned char* A, float* absmax, void* out, int bs, const int n,
                                    bnb_stream_t s) {
    dequantize_4bit_bf16(A, absmax, out, bs, n, kFP4, S(s));
}

// ------------------------------------------------------------------ 8-bit blockwise (double-quant helper)
void cquantize_blockwise_fp32(float* code, float* A, float* absmax, unsigned char* out, int bs, const int n) {
    quantize_8bit_f32(code, A, absmax, out, bs, n, nullptr);
}
void cquantize_blockwise_fp16(float* code, void* A, float* absmax, unsigned char* out, int bs, const int n) {
    quantize_8bit_f16(code, A, absmax, out, bs, n, nullptr);
}
void cquantize_blockwise_bf16(float* code, void* A, float* absmax, unsigned char* out, int bs, const int n) {
    quantize_8bit_bf16(code, A, absmax, out, bs, n, nullptr);
}
void cdequantize_blockwise_fp32(float* code, unsigned char* A, float* absmax, float* out, int bs, const int n,
                                bnb_stream_t s) {
    dequantize_8bit_f32(code, A, absmax, out, bs, n, S(s));
}
void cdequantize_blockwise_fp16(float* code, unsigned char* A, float* absmax, void* out, int bs, const int n,
                                bnb_stream_t s) {
    dequantize_8bit_f16(code, A, absmax, out, bs, n, S(s));
}
void cdequantize_blockwise_bf16(float* code, unsigned char* A, float* absmax, void* out, int bs, const int n,
                                bnb_stream_t s) {
    dequantize_8bit_bf16(code, A, absmax, out, bs, n, S(s));
}

// ------------------------------------------------------------------ fused dequant + GEMM
void cgemm_4bit_bf16(const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax_8bit,
                     const float* absmax_code, const float* absmax_offset, void* out, const void* bias, int M, int N,
                     int K, int blocksize, int quant_type, bnb_stream_t s) {
    gemm_4bit_dispatch(0, 2, A, B, absmax, absmax_8bit, absmax_code, absmax_offset, nullptr, out, bias, M, N, K,
                       blocksize, quant_type, nullptr, 0, S(s));
}
void cgemm_4bit_fp16(const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax_8bit,
                     const float* absmax_code, const float* absmax_offset, void* out, const void* bias, int M, int N,
                     int K, int blocksize, int quant_type, bnb_stream_t s) {
    gemm_4bit_dispatch(0, 1, A, B, absmax, absmax_8bit, absmax_code, absmax_offset, nullptr, out, bias, M, N, K,
                       blocksize, quant_type, nullptr, 0, S(s));
}
void cgemm_4bit_fp32(const float* A, const uint8_t* B, const float* absmax, const uint8_t* absmax_8bit,
                     const float* absmax_code, const float* absmax_offset, float* out, const float* bias, int M, int N,
                     int K, int blocksize, int quant_type, bnb_stream_t s) {
    gemm_4bit_dispatch(0, 0, A, B, absmax, absmax_8bit, absmax_code, absmax_offset, nullptr, out, bias, M, N, K,
                       blocksize, quant_type, nullptr, 0, S(s));
}

// ------------------------------------------------------------------ legacy gemv (m = N, n = 1, k = K)
void cgemm_4bit_inference_naive_fp16(int m, int n, int k, void* A, unsigned char* B, float* absmax, float* datatype,
                                     void* out, int, int, int, int blocksize, bnb_stream_t s) {
    (void)n;
    gemv_4bit_stream(1, A, B, absmax, nullptr, nullptr, nullptr, datatype, out, nullptr, 1, m, k, blocksize, kNF4, S(s));
}
void cgemm_4bit_inference_naive_bf16(int m, int n, int k, void* A, unsigned char* B, float* absmax, float* datatype,
                                     void* out, int, int, int, int blocksize, bnb_stream_t s) {
    (void)n;
    gemv_4bit_stream(2, A, B, absmax, nullptr, nullptr, nullptr, datatype, out, nullptr, 1, m, k, blocksize, kNF4, S(s));
}
void cgemm_4bit_inference_naive_fp32(int m, int n, int k, float* A, unsigned char* B, float* absmax, float* datatype,
                                     float* out, int, int, int, int blocksize, bnb_stream_t s) {
    (void)n;
    gemv_4bit_stream(0, A, B, absmax, nullptr, nullptr, nullptr, datatype, out, nullptr, 1, m, k, blocksize, kNF4, S(s));
}

// ------------------------------------------------------------------ loader symbols
void* get_context(void) {
    static int token = 0x4d493335; // opaque: this path needs no BLAS handle
    return &token;
}
void* cget_managed_ptr(size_t bytes) {
    void* ptr = nullptr;
    BNB_HIP_CHECK(hipMallocManaged(&ptr, bytes, hipMemAttachHost));
    return ptr;
}

// ------------------------------------------------------------------ extensions
void bnb_mi355x_quantize_4bit(const void* A, int dtype, float* absmax, unsigned char* out, int blocksize, long n,
                              int quant_type, bnb_stream_t s) {
    if (quant_type != kFP4 && quant_type != kNF4) {
        fprintf(stderr, "bitsandbytes_amd: quantize_4bit: quant_type must be 1 (FP4) or 2 (NF4)\n");
        exit(1);
    }
    if (dtype == 0)
        quantize_4bit_f32(static_cast<const float*>(A), absmax, out, blocksize, n, quant_type, S(s));
    else if (dtype == 1)
        quantize_4bit_f16(A, absmax, out, blocksize, n, quant_type, S(s));
    else
        quantize_4bit_bf16(A, absmax, out, blocksize, n, quant_type, S(s));
}
void bnb_mi355x_dequantize_4bit_rows(int dtype, const unsigned char* A, const float* absmax, const void* indices,
                                     int index_bytes, void* out, long rows_out, long num_rows, int row_len,
                                     int blocksize, int quant_type, bnb_stream_t s) {
    dequantize_4bit_rows(dtype, A, absmax, indices, index_bytes, out, rows_out, num_rows, row_len, blocksize,
                         quant_type, S(s));
}
void bnb_mi355x_dequantize_4bit_nested(int dtype, const unsigned char* A, const unsigned char* absmax_8bit, const float* absmax2,
                                       const float* absmax_code, const float* absmax_offset, void* out, int blocksize, long n,
                                       int quant_type, bnb_stream_t s) {
    if (quant_type != kFP4 && quant_type != kNF4) {
        fprintf(stderr, "bitsandbytes_amd: dequantize_4bit_nested: quant_type must be 1 (FP4) or 2 (NF4)\n");
        exit(1);
    }
    dequantize_4bit_nested(dtype, A, absmax_8bit, absmax2, absmax_code, absmax_offset, out, blocksize, n, quant_type, S(s));
}
void bnb_mi355x_quantize_8bit(const float* code, const void* A, int dtype, float* absmax, unsigned char* out,
                              int blocksize, long n, bnb_stream_t s) {
    if (dtype == 0)
        quantize_8bit_f32(code, static_cast<const float*>(A), absmax, out, blocksize, n, S(s));
    else if (dtype == 1)
        quantize_8bit_f16(code, A, absmax, out, blocksize, n, S(s));
    else
        quantize_8bit_bf16(code, A, absmax, out, blocksize, n, S(s));
}
void bnb_mi355x_quantize_4bit_nested(const void* A, int dtype, long n, int blocksize, int quant_type, unsigned char* out,
                                     float* scratch, const float* code8, unsigned char* absmax_8bit, float* absmax2,
                                     float* offset, bnb_stream_t s) {
    // quantize_4bit(compress_statistics=True) of the reference (bitsandbytes/functional.py:925-951) as ONE call: the 4-bit encoder
    // into scratch[0, blocks), then the statistics (mean, subtract, 8-bit encode in blocks of 256) in two launches
    bnb_mi355x_quantize_4bit(A, dtype, scratch, out, blocksize, n, quant_type, s);
    const long blocks = (n + blocksize - 1) / blocksize;
    quantize_absmax_nested(code8, scratch, blocks, scratch + blocks, offset, absmax_8bit, absmax2, S(s));
}
void bnb_mi355x_gemm_4bit(int kernel, int dtype, const void* A, const uint8_t* B, const float* absmax,
                          const uint8_t* absmax_8bit, const float* absmax_code, const float* absmax_offset,
                          const float* code16, void* out, const void* bias, int M, int N, int K, int blocksize,
                          int quant_type, void* workspace, size_t workspace_bytes, bnb_stream_t s) {
    gemm_4bit_dispatch(kernel, dtype, A, B, absmax, absmax_8bit, absmax_code, absmax_offset, code16, out, bias, M, N, K,
                       blocksize, quant_type, workspace, workspace_bytes, S(s));
}
// Groups of 17 ... 64 rows as ONE launch of the streaming MFMA kernel - although no member's own route at that many rows is this
// kernel: its 32-row instances multiply two 16-row blocks against every decoded weight fragment (row passes of 32 over grid.y above
// that; small groups: passes of 16 side by side on the CUs they leave idle), and the group shares one boundary, one table build and one
// activation fetch per CU. Measured (profiles/r6_grouped_ab.txt, third table; us per group, grouped launch vs the members one by one):
// 4 x 4096^2 M = 24 / 32 / 48 / 64 17.2 / 17.3 / 29.6 / 30.2 vs 32.9 / 33.4 / 39.4 / 44.4; 4096 + 2 x 1024 (x 4096) 11.0 / 11.3 / 19.7 / 20.0
// vs 19.4 / 20.2 / 21.6 / 22.8; 3 x 512 x 4096 5.9 / 5.9 / 7.9 / 7.9 vs 15.1 / 15.3 / 15.4 / 15.4; 2 x 11008 x 4096 26.8 / 27.0 vs 28.0 / 28.4
// but 47.4 / 48.4 vs 35.0 / 35.5 at 48 / 64 rows; 2 x 14336 x 4096 and 3 x 8192^2 behind (big members keep their own kernels).
// Such a group is NOT bit-identical to separate calls (another kernel family: the oracle's tolerance, like every fused call).
static bool grouped_sm_passes(int dtype, int count, const int* N, int M, int K, int blocksize) {
    if (dtype == 0 || M <= 16 || M > 64 || count < 2 || count > 8 || blocksize < 64 || (K % 64) != 0 || g_mfma_knob0.load(std::memory_order_relaxed) != 0 ||
        g_mfma_knob1.load(std::memory_order_relaxed) != 0)
        return false;
    long long weights = 0;
    for (int i = 0; i < count; ++i)
        weights += static_cast<long long>(N[i]) * K;
    return weights <= (M <= 32 ? (96LL << 20) : (72LL << 20));
}
void bnb_mi355x_gemm_4bit_grouped(int dtype, const void* A, int count, const uint8_t* const* B, const float* const* absmax,
                                  const uint8_t* const* absmax_8bit, const float* const* absmax_code,
                                  const float* const* absmax_offset, void* const* out, const void* const* bias, const int* N,
                                  int M, int K, int blocksize, int quant_type, bnb_stream_t s) {
    if (count <= 0 || M <= 0)
        return;
    if (quant_type != kFP4 && quant_type != kNF4) {
        fprintf(stderr, "bitsandbytes_amd: gemm_4bit_grouped: quant_type must be 1 (FP4) or 2 (NF4), got %d\n", quant_type);
        exit(1);
    }
    // "bit-identical to separate calls": every member runs the kernel FAMILY the single-matrix entry point would give it.
    //  * every member routed to the streaming MFMA kernel (2 ... 16 rows, csrc/gemm4_mfma.hip: sm_selected): ONE launch of that
    //    kernel over the members' rows (its summation order does not depend on the launch geometry) - round 6;
    //  * no member routed to an MFMA kernel: ONE launch of the streaming kernel (M <= 4);
    //  * anything else (mixed routes, M > 16, more than 8 matrices, odd K ...): matrix by matrix.
    bool any_mfma = false, all_sm = count <= 8;
    for (int i = 0; i < count; ++i) {
        any_mfma = any_mfma || route_to_mfma(0, dtype, A, B[i], nullptr, M, N[i], K, blocksize,
                                             (absmax_8bit == nullptr || absmax_8bit[i] == nullptr) && aligned_to(absmax[i], 16));
        all_sm = all_sm && gemm_4bit_sm_routes(dtype, M, N[i], K, blocksize);
    }
    if ((all_sm || grouped_sm_passes(dtype, count, N, M, K, blocksize)) &&
        gemm_4bit_sm_grouped(dtype, A, count, B, absmax, absmax_8bit, absmax_code, absmax_offset, out, bias, N, M, K, blocksize, quant_type, S(s)))
        return;
    if (!any_mfma && gemv_4bit_grouped(dtype, A, count, B, absmax, absmax_8bit, absmax_code, absmax_offset, out, bias, N, M, K,
                                       blocksize, quant_type, S(s)))
        return;
    // not a streaming-kernel shape (M > 4, more than 8 matrices, odd K ...): one launch per matrix, same results
    for (int i = 0; i < count; ++i)
        gemm_4bit_dispatch(0, dtype, A, B[i], absmax[i], absmax_8bit ? absmax_8bit[i] : nullptr,
                           absmax_code ? absmax_code[i] : nullptr, absmax_offset ? absmax_offset[i] : nullptr, nullptr, out[i],
                           bias ? bias[i] : nullptr, M, N[i], K, blocksize, quant_type, nullptr, 0, S(s));
}
int bnb_mi355x_gemm_4bit_grouped_route(int dtype, int count, const int* N, int M, int K, int blocksize) {
    // what bnb_mi355x_gemm_4bit_grouped does with such a group, assuming aligned pointers and members of one kind of statistics:
    // 2 = one launch of the streaming MFMA kernel, 1 = one launch of the streaming kernel, 0 = matrix by matrix
    static const int dummy_aligned[4] __attribute__((aligned(16))) = {0, 0, 0, 0};
    const void* a = dummy_aligned;
    if (count <= 0 || count > 8 || M <= 0 || blocksize <= 0 || K % blocksize != 0)
        return 0;
    bool any_mfma = false, all_sm = true;
    for (int i = 0; i < count; ++i) {
        any_mfma = any_mfma || route_to_mfma(0, dtype, a, reinterpret_cast<const uint8_t*>(a), nullptr, M, N[i], K, blocksize, true);
        all_sm = all_sm && gemm_4bit_sm_routes(dtype, M, N[i], K, blocksize) &&
                 gemm_4bit_sm_supported(dtype, a, reinterpret_cast<const uint8_t*>(a), nullptr, M, N[i], K, blocksize);
    }
    if (all_sm)
        return 2;
    if (grouped_sm_passes(dtype, count, N, M, K, blocksize)) {
        bool ok = true;
        for (int i = 0; i < count; ++i)
            ok = ok && gemm_4bit_sm_supported(dtype, a, reinterpret_cast<const uint8_t*>(a), nullptr, M, N[i], K, blocksize);
        if (ok)
            return 2;
    }
    return (!any_mfma && M <= 4) ? 1 : 0;
}
size_t bnb_mi355x_gemm_4bit_workspace_bytes(int kernel, int dtype, int M, int N, int K, int blocksize) {
    // alignment of A/B is unknown here; assume the aligned (fast) case, an unused workspace is harmless
    static const int dummy_aligned[4] __attribute__((aligned(16))) = {0, 0, 0, 0};
    const void* a = dummy_aligned;
    if (!route_to_mfma(kernel, dtype, a, reinterpret_cast<const uint8_t*>(a), nullptr, M, N, K, blocksize, true))
        return 0;
    return gemm_4bit_mfma_workspace_bytes(M, N, K, blocksize);
}
int bnb_mi355x_last_gemm_kernel(void) { return g_last_gemm_kernel; }

// ------------------------------------------------------------------ peer chain (the all-gather fused into the gemv launches)
static uint32_t peer_chain_spin_bound() {
    // a re-fetch is s_sleep 4 (256 cycles) + a system-scope load round trip: ~1 us. BNB_MI355X_PEER_WAIT_POLLS as in peer_gather.hip
    static const uint32_t bound = [] {
        const char* e = getenv("BNB_MI355X_PEER_WAIT_POLLS");
        const long long v = e ? atoll(e) : 0;
        return static_cast<uint32_t>(v > 0 && v < 4000000000LL ? v : 30000000LL);
    }();
    return bound;
}
size_t bnb_mi355x_peer_chain_buffer_bytes(long max_values) {
    // header + 64 regions (gemv4_stream.hip: kChainRegions) of max_values / 2 granules of 8 bytes; max_values counts in fours
    // (an even number of granules per region: every region starts 16-byte aligned, which the two-granule stores and fetches need)
    const size_t mv = max_values > 0 ? (static_cast<size_t>(max_values) + 3) & ~static_cast<size_t>(3) : 0;
    return 256 + 64 * 4 * mv;
}
void* bnb_mi355x_peer_chain_alloc(size_t bytes, int fine_grained) {
    // Zeroed device memory for a rank's exchange buffer; exported / mapped / freed with the bnb_mi355x_peer_* functions.
    //   fine_grained != 0: hipDeviceMallocFinegrained - what a chain whose ranks sit on DIFFERENT devices must use. Remote GPUs store
    //     into this buffer while a kernel of the owner polls it; HIP makes coarse-grained memory coherent across devices at kernel
    //     boundaries only, so a line of an ordinary allocation that the owner's L2 holds may never show the remote store.
    //   fine_grained == 0: ordinary (cacheable) hipMalloc memory - enough where every rank runs on ONE device (processes sharing a
    //     GPU, a group of one rank): one L2 hierarchy, nothing crosses a link (profiles/r4_peer_chain.txt, builds 3 and 6).
    void* p = nullptr;
    const hipError_t err = fine_grained ? hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) : hipMalloc(&p, bytes);
    if (err != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(p);
        return nullptr;
    }
    return p;
}
int bnb_mi355x_gemv_4bit_peer_serves(int world, int ns, int K, int blocksize, int mode, long max_values, int wg_limit) {
    // the launcher's own shape check (launch geometry included), without launching: 1 = bnb_mi355x_gemv_4bit_peer would accept
    // these shapes (given 16-byte aligned B / A, a valid rank and dtype). Needs the current device (CU count).
    return gemv_4bit_peer_serves(world, ns, K, blocksize, mode, max_values, wg_limit) ? 1 : 0;
}
int bnb_mi355x_gemv_4bit_peer(void* const* bufs, void* epoch_word, int world, int rank, int dtype, const void* A, const uint8_t* B, const float* absmax,
                              const uint8_t* absmax_8bit, const float* absmax_code, const float* absmax_offset, const void* bias,
                              void* out_local, int ns, int K, int blocksize, int quant_type, int mode, long max_values, int wg_limit,
                              int epoch_offset, bnb_stream_t s) {
    if (quant_type != kFP4 && quant_type != kNF4) {
        fprintf(stderr, "bitsandbytes_amd: gemv_4bit_peer: quant_type must be 1 (FP4) or 2 (NF4), got %d\n", quant_type);
        exit(1);
    }
    return gemv_4bit_peer(bufs, epoch_word, world, rank, dtype, A, B, absmax, absmax_8bit, absmax_code, absmax_offset, bias, out_local, ns, K,
                          blocksize, quant_type, mode, max_values, wg_limit, static_cast<uint32_t>(epoch_offset), peer_chain_spin_bound(), S(s))
               ? 1
               : 0;
}
void bnb_mi355x_peer_chain_read(void* const* bufs, void* epoch_word, int world, int rank, int dtype, void* out, int nvalues, long max_values, int epoch_offset,
                                bnb_stream_t s) {
    if (epoch_word == nullptr || world < 1 || world > 8 || rank < 0 || rank >= world || (dtype != 1 && dtype != 2) || nvalues < 2 || (nvalues & 1) || nvalues > max_values) {
        fprintf(stderr, "bitsandbytes_amd: peer_chain_read: bad arguments (world %d, rank %d, dtype %d, %d values of %ld)\n", world, rank, dtype,
                nvalues, max_values);
        exit(1);
    }
    peer_chain_read(bufs, epoch_word, world, rank, dtype, out, nvalues, max_values, static_cast<uint32_t>(epoch_offset), peer_chain_spin_bound(), S(s));
}
int bnb_mi355x_gemm_4bit_route(int kernel, int dtype, int M, int N, int K, int blocksize) {
    // (alignment of A / B is unknown here; the aligned - fast - case is assumed, as in the workspace query)
    static const int dummy_aligned[4] __attribute__((aligned(16))) = {0, 0, 0, 0};
    return route_to_mfma(kernel, dtype, dummy_aligned, reinterpret_cast<const uint8_t*>(dummy_aligned), nullptr, M, N, K, blocksize, true) ? 1 : 0;
}
void bnb_mi355x_gemm_4bit_grad_input(int dtype, const void* grad_out, const uint8_t* B, const float* absmax,
                                     const uint8_t* absmax_8bit, const float* absmax_code, const float* absmax_offset, void* grad_A,
                                     int M, int N, int K, int blocksize, int quant_type, void* workspace, size_t workspace_bytes,
                                     bnb_stream_t s) {
    if (quant_type != kFP4 && quant_type != kNF4) {
        fprintf(stderr, "bitsandbytes_amd: gemm_4bit_grad_input: quant_type must be 1 (FP4) or 2 (NF4), got %d\n", quant_type);
        exit(1);
    }
    gemm_4bit_grad_input(dtype, grad_out, B, absmax, absmax_8bit, absmax_code, absmax_offset, grad_A, M, N, K, blocksize,
                         quant_type, workspace, workspace_bytes, S(s));
}
size_t bnb_mi355x_gemm_4bit_grad_input_workspace_bytes(int M, int N, int K) { return gemm_4bit_grad_input_workspace_bytes(M, N, K); }
int bnb_mi355x_gemm_4bit_grad_input_supported(int dtype, int M, int N, int K, int blocksize) {
    static const int dummy_aligned[4] __attribute__((aligned(16))) = {0, 0, 0, 0};
    return gemm_4bit_grad_input_supported(dtype, dummy_aligned, reinterpret_cast<const uint8_t*>(dummy_aligned), M, N, K, blocksize) ? 1 : 0;
}
void bnb_mi355x_set_tuning(int reserved0, int reserved1, int mfma_knob0, int mfma_knob1) {
    quantize_8bit_set_variant(reserved0); // 1 / 2: force the cell-table / byte-table 8-bit encoder, anything else: by size
    // 3: the one-tile form of the 4-bit quantize kernel everywhere (A/B of the pipelined FP4 form); 4 / 5: round 4's 4 chunks / 8 chunks per
    // workgroup instead of the shipped 2 on large NF4 inputs (A/B, round 5)
    quantize_4bit_set_variant(reserved0 == 3 ? 1 : reserved0 == 4 ? 4 : reserved0 == 5 ? 3 : reserved0 == 8 ? 2 : 0); // (8: 2 chunks forced, FP4 included)
    dequantize_4bit_set_variant(reserved0 >= 10 ? reserved0 - 10 : 0); // 10 + v: tile / lane-mapping variants of dequantize4 (dequantize4.hip)
    gemm_4bit_grad_input_set_slices(reserved1); // N slices of the fused backward (sweeps), 0: built-in
    g_mfma_knob0.store(mfma_knob0, std::memory_order_relaxed);
    g_mfma_knob1.store(mfma_knob1, std::memory_order_relaxed);
}
void bnb_mi355x_set_stream_tuning(int ring_depth, int segments, int rows_per_workgroup, int nontemporal, int waves) {
    gemv_4bit_stream_tuning(ring_depth, segments, rows_per_workgroup, nontemporal, waves);
}
void bnb_mi355x_set_stamp_buffer(void* device_u64_buffer) {
#ifdef BNB_PROFILING
    g_dbg_buf = static_cast<unsigned long long*>(device_u64_buffer);
#else
    (void)device_u64_buffer; // the product library carries no stamp code: the call is accepted and ignored
#endif
}
const char* bnb_mi355x_version(void) { return "bitsandbytes_amd 0.1.0 gfx950"; }

#pragma GCC visibility pop
} // extern "C"
