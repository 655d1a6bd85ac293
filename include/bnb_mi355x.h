/*
 * bnb_mi355x.h — C ABI of libbitsandbytes_mi355x.so, the MI355X (gfx950) native backend for the
 * bitsandbytes 4-bit quantized-linear path.
 *
 * The first group of entry points is exactly what the reference's ctypes layer binds for this path
 * (reference bitsandbytes/backends/cuda/ops.py:16-66): same names, same argument order, same types,
 * `void` return, errors reported as the reference's BNB_CHECK_RETURN does (message on stderr and
 * exit(1), reference csrc/compat.cuh:78-85). A reference build pointed at this library (see
 * INTEGRATION.md) runs its quantize_4bit / dequantize_4bit / gemm_4bit / gemv_4bit ops on these
 * kernels unchanged.
 *
 * Conventions (reference SURVEY §8b):
 *   - every pointer is a raw device address owned by the caller; outputs are pre-allocated; the
 *     library allocates nothing and keeps no pointer after return;
 *   - calls only enqueue work on `stream` (no synchronisation, no allocation): legal inside
 *     hipGraph capture;
 *   - packed weights: row-major flat over [N, K]; element 2i in the HIGH nibble and 2i+1 in the
 *     low nibble of byte i; quantization block j covers flat elements [j*bs, (j+1)*bs);
 *   - quant_type: 1 = FP4, 2 = NF4 (reference csrc/common.h:3-7);
 *   - `bnb_stream_t` is a hipStream_t passed as void*.
 *
 * fp16 / bf16 buffers are declared `void*` here so the header is usable from plain C.
 */
#ifndef BNB_MI355X_H
#define BNB_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* bnb_stream_t;

/* ------------------------------------------------------------------------------------------------
 * 4-bit blockwise quantize — replaces reference csrc/pythonInterface.cpp:364-426
 * (cquantize_blockwise_<T>_{fp4,nf4}); `code` is unused (NULL). As in the reference these take NO
 * stream and run on the NULL stream. out: (n+1)/2 bytes; absmax: ceil(n/blocksize) floats.
 * blocksize in {32,64,...,4096}. Results are bit-identical to the reference CPU backend
 * (bitsandbytes/backends/default/ops.py:233-259).
 * ---------------------------------------------------------------------------------------------- */
void cquantize_blockwise_fp32_nf4(float* code, float* A, float* absmax, unsigned char* out, int blocksize, const int n);
void cquantize_blockwise_fp32_fp4(float* code, float* A, float* absmax, unsigned char* out, int blocksize, const int n);
void cquantize_blockwise_fp16_nf4(float* code, void* A, float* absmax, unsigned char* out, int blocksize, const int n);
void cquantize_blockwise_fp16_fp4(float* code, void* A, float* absmax, unsigned char* out, int blocksize, const int n);
void cquantize_blockwise_bf16_nf4(float* code, void* A, float* absmax, unsigned char* out, int blocksize, const int n);
void cquantize_blockwise_bf16_fp4(float* code, void* A, float* absmax, unsigned char* out, int blocksize, const int n);

/* 4-bit blockwise dequantize — replaces reference csrc/pythonInterface.cpp:346-444
 * (cdequantize_blockwise_<T>_{fp4,nf4}); n = number of OUTPUT elements. */
void cdequantize_blockwise_fp32_nf4(float* code, unsigned char* A, float* absmax, float* out, int blocksize, const int n, bnb_stream_t stream);
void cdequantize_blockwise_fp32_fp4(float* code, unsigned char* A, float* absmax, float* out, int blocksize, const int n, bnb_stream_t stream);
void cdequantize_blockwise_fp16_nf4(float* code, unsigned char* A, float* absmax, void* out, int blocksize, const int n, bnb_stream_t stream);
void cdequantize_blockwise_fp16_fp4(float* code, unsigned char* A, float* absmax, void* out, int blocksize, const int n, bnb_stream_t stream);
void cdequantize_blockwise_bf16_nf4(float* code, unsigned char* A, float* absmax, void* out, int blocksize, const int n, bnb_stream_t stream);
void cdequantize_blockwise_bf16_fp4(float* code, unsigned char* A, float* absmax, void* out, int blocksize, const int n, bnb_stream_t stream);

/* 8-bit blockwise pair with a 256-entry fp32 `code` (General8bit) — replaces reference
 * csrc/pythonInterface.cpp:346-444 (cquantize_blockwise_<T> / cdequantize_blockwise_<T>). On this
 * path they only (de)compress the absmax vector for double quantization. Quantize follows the
 * reference CPU kernel's 65536-bin rule (csrc/cpu_ops.cpp:501-665). Quantize: NULL stream. */
void cquantize_blockwise_fp32(float* code, float* A, float* absmax, unsigned char* out, int blocksize, const int n);
void cquantize_blockwise_fp16(float* code, void* A, float* absmax, unsigned char* out, int blocksize, const int n);
void cquantize_blockwise_bf16(float* code, void* A, float* absmax, unsigned char* out, int blocksize, const int n);
void cdequantize_blockwise_fp32(float* code, unsigned char* A, float* absmax, float* out, int blocksize, const int n, bnb_stream_t stream);
void cdequantize_blockwise_fp16(float* code, unsigned char* A, float* absmax, void* out, int blocksize, const int n, bnb_stream_t stream);
void cdequantize_blockwise_bf16(float* code, unsigned char* A, float* absmax, void* out, int blocksize, const int n, bnb_stream_t stream);

/* Fused dequantize + GEMM — replaces reference csrc/gemm_4bit.cu:138-166.
 *   out[M,N] = A[M,K] * dequant(B)[N,K]^T (+ bias[N])
 *   scale of block b = absmax[b]                                            (absmax_8bit == NULL)
 *                    = absmax_code[absmax_8bit[b]] * absmax[b >> 8] + *absmax_offset   (nested)
 * K % blocksize == 0 is guaranteed by the caller (reference backends/cuda/ops.py:956-962);
 * absmax_offset is fp32; bias has A's dtype. M is any positive value: M = 1 (and M <= 4 on matrices of fewer than 128 rows,
 * M = 2 on long rows of small matrices) runs the streaming kernel (gemv4_stream.hip: persistent workgroups, weights through a
 * register ring, any M in row passes of up to four); 2 ... 16 rows on matrices of >= 128 rows the streaming MFMA kernel
 * (gemm4_mfma_sm.hip: one persistent workgroup per CU, one decode for all rows, activations once per CU; K % 256 != 0: to 128 rows); larger M and smaller
 * matrices the other MFMA kernels (gemm4_mfma_rt.hip / gemm4_mfma.hip / gemm4_mfma_kq.hip: bf16 / fp16, K % 256 == 0, blocksize
 * >= 64, aligned pointers; any M in row tiles); shapes the MFMA kernels do not take (fp32, odd K, small blocks) run the
 * streaming kernel at any M. */
void cgemm_4bit_bf16(const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax_8bit, const float* absmax_code, const float* absmax_offset, void* out, const void* bias, int M, int N, int K, int blocksize, int quant_type, bnb_stream_t stream);
void cgemm_4bit_fp16(const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax_8bit, const float* absmax_code, const float* absmax_offset, void* out, const void* bias, int M, int N, int K, int blocksize, int quant_type, bnb_stream_t stream);
void cgemm_4bit_fp32(const float* A, const uint8_t* B, const float* absmax, const uint8_t* absmax_8bit, const float* absmax_code, const float* absmax_offset, float* out, const float* bias, int M, int N, int K, int blocksize, int quant_type, bnb_stream_t stream);

/* Legacy gemv (M = 1) — replaces reference csrc/pythonInterface.cpp:594-613.
 * m = N (output features), n = 1, k = K; `datatype` = the 16-entry fp32 code table on the device;
 * absmax already un-nested; lda/ldb/ldc are ignored exactly as the reference kernel ignores them
 * (it indexes B flat, reference csrc/kernels.cu:1452-1567). */
void cgemm_4bit_inference_naive_fp16(int m, int n, int k, void* A, unsigned char* B, float* absmax, float* datatype, void* out, int lda, int ldb, int ldc, int blocksize, bnb_stream_t stream);
void cgemm_4bit_inference_naive_bf16(int m, int n, int k, void* A, unsigned char* B, float* absmax, float* datatype, void* out, int lda, int ldb, int ldc, int blocksize, bnb_stream_t stream);
void cgemm_4bit_inference_naive_fp32(int m, int n, int k, float* A, unsigned char* B, float* absmax, float* datatype, float* out, int lda, int ldb, int ldc, int blocksize, bnb_stream_t stream);

/* Loader symbols the reference's cextension.py:109-115,371-372 probes to classify a GPU library.
 * get_context returns an opaque non-NULL token (this path needs no BLAS handle);
 * cget_managed_ptr returns hipMallocManaged memory (reference csrc/pythonInterface.cpp:522,557-563). */
void* get_context(void);
void* cget_managed_ptr(size_t bytes);

/* ------------------------------------------------------------------------------------------------
 * Extensions (not in the reference ABI). Used by bitsandbytes_amd's own host layer.
 * ---------------------------------------------------------------------------------------------- */

/* Stream-ordered 4-bit / 8-bit quantize: same kernels as the cquantize_blockwise_* family above but
 * enqueued on `stream`. dtype: 0 = fp32, 1 = fp16, 2 = bf16. */
void bnb_mi355x_quantize_4bit(const void* A, int dtype, float* absmax, unsigned char* out, int blocksize, long n, int quant_type, bnb_stream_t stream);
void bnb_mi355x_quantize_8bit(const float* code, const void* A, int dtype, float* absmax, unsigned char* out, int blocksize, long n, bnb_stream_t stream);

/* quantize_4bit(compress_statistics=True) as ONE call (reference bitsandbytes/functional.py:925-951: quantize_4bit, absmax.mean(),
 * absmax - offset, quantize_blockwise(..., blocksize=256) - four to six launches and host dispatches; here three launches behind one).
 * A: n elements of dtype; out: (n + 1) / 2 packed bytes; scratch: device buffer of ceil(n / blocksize) + 1536 floats (the fp32 absmax of
 * the 4-bit blocks, then 256 partial sums, then 1280 dwords of encoder tables built once per call; contents are unspecified afterwards); code8: the 256-entry 8-bit code (fp32, device, ascending:
 * the dynamic map); absmax_8bit: ceil(n / blocksize) codes; absmax2: ceil(ceil(n / blocksize) / 256) floats; offset: 1 float =
 * the mean of the fp32 absmax, summed in a FIXED order (a balanced binary tree over 1024-element steps, see csrc/blockwise8.hip:
 * the same bits on every launch, any device). absmax_8bit / absmax2 are bit for bit what quantize_blockwise gives on absmax - offset. */
void bnb_mi355x_quantize_4bit_nested(const void* A, int dtype, long n, int blocksize, int quant_type, unsigned char* out, float* scratch, const float* code8, unsigned char* absmax_8bit, float* absmax2, float* offset, bnb_stream_t stream);

/* dequantize_4bit for double-quantised statistics in ONE launch (reference bitsandbytes/functional.py:1002-1006 is three operator
 * calls: dequantize_blockwise(absmax, state2), `+= offset`, dequantize_4bit): the scale of 4-bit block b is reconstructed in the kernel as
 * absmax_code[absmax_8bit[b]] * absmax2[b >> 8] + *absmax_offset (fp32 product, then fp32 sum: the same two roundings). Second-level
 * blocksize 256. n = number of OUTPUT elements; dtype: 0 = fp32, 1 = fp16, 2 = bf16. */
void bnb_mi355x_dequantize_4bit_nested(int dtype, const unsigned char* A, const unsigned char* absmax_8bit, const float* absmax2, const float* absmax_code, const float* absmax_offset, void* out, int blocksize, long n, int quant_type, bnb_stream_t stream);

/* Row gather + 4-bit dequantize in one launch: out[t, 0:row_len] = dequantize(row indices[t]) for
 * t < rows_out. The fused form of the Embedding4bit lookup (reference bitsandbytes/nn/modules.py:921-951:
 * F.embedding on the packed bytes, F.embedding on absmax, dequantize_4bit). A is the packed
 * [num_rows, row_len] table, absmax its fp32 scales (un-nested); row_len % 8 == 0 and
 * row_len % blocksize == 0; indices are int32 (index_bytes 4) or int64 (8). Values are bit-identical to
 * cdequantize_blockwise_* applied to the gathered rows. An index outside [0, num_rows) gives a row of NaN. */
void bnb_mi355x_dequantize_4bit_rows(int dtype, const unsigned char* A, const float* absmax, const void* indices, int index_bytes, void* out, long rows_out, long num_rows, int row_len, int blocksize, int quant_type, bnb_stream_t stream);

/* gemm_4bit with an explicit kernel choice and a caller-owned split-K workspace:
 * kernel = 0 auto, 1 / 3 streaming dot kernel, 2 MFMA kernels. dtype as above; code16 may be NULL.
 * workspace: device buffer of bnb_mi355x_gemm_4bit_workspace_bytes(...) bytes (may be NULL / smaller:
 * the MFMA kernel then uses fewer K slices, or a library-owned per-stream buffer when NULL and the
 * stream is not being captured). Its contents are scratch; no initialisation is required. */
void bnb_mi355x_gemm_4bit(int kernel, int dtype, const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax_8bit, const float* absmax_code, const float* absmax_offset, const float* code16, void* out, const void* bias, int M, int N, int K, int blocksize, int quant_type, void* workspace, size_t workspace_bytes, bnb_stream_t stream);
size_t bnb_mi355x_gemm_4bit_workspace_bytes(int kernel, int dtype, int M, int N, int K, int blocksize);
/* Which kernel family bnb_mi355x_gemm_4bit(kernel, ...) runs for this problem (aligned pointers assumed): 0 = streaming
 * kernel, 1 = MFMA kernels. The grouped entry point below goes matrix by matrix when any member answers 1. */
int bnb_mi355x_gemm_4bit_route(int kernel, int dtype, int M, int N, int K, int blocksize);
/* Which kernel family the calling thread's LAST gemm_4bit / gemv_4bit call (any entry point) launched: 0 none yet, 1 streaming
 * kernel (gemv4_stream_kernel), 2 generic scalar kernel (odd shapes), 3 register-transposed MFMA kernel (gemm4_mfma_rt_kernel),
 * 4 producer/consumer MFMA kernel (gemm4_mfma_pc_kernel), 6 K-quarter MFMA kernel (gemm4_mfma_kq_kernel),
 * 7 streaming MFMA kernel (gemm4_mfma_sm_kernel: 2 ... 16 rows, one persistent workgroup per CU). Debug / test query:
 * a test that forces a kernel with bnb_mi355x_set_tuning asserts here that it ran (a geometry the forced kernel does not
 * serve falls back to another family by design). */
int bnb_mi355x_last_gemm_kernel(void);

/* Grouped gemm_4bit: `count` weight matrices applied to the SAME activations A[M, K] in one launch -
 *   out[i][M, N[i]] = A * dequant(B[i])^T (+ bias[i])        i = 0 .. count-1
 * (the Q/K/V projections of an attention block, the gate/up projections of an MLP: reference callers issue one
 * gemm_4bit per matrix, bitsandbytes/nn/modules.py:609-637). All matrices share K, blocksize, quant_type and
 * nested-ness (absmax_8bit is NULL, or non-NULL for every matrix). The arrays are HOST arrays of device pointers
 * / ints, read during the call. count <= 8: ONE launch - of the streaming MFMA kernel when every member's own route is that
 * kernel (2 ... 16 rows; workgroups dealt to the members by rows), of the streaming kernel over the concatenated rows when no
 * member's route is an MFMA kernel (M <= 4) - one kernel boundary, one decode-table build, one activation copy per CU;
 * otherwise the matrices are launched one by one. Results are bit-identical to `count` separate cgemm_4bit_* calls - with ONE
 * exception: groups of 17 ... 64 rows up to 96 M weights (72 M from 33 rows) are one launch of the streaming MFMA kernel - its 32-row
 * instances - although the members' own route at that many rows is another MFMA kernel (4 x 4096^2 at 32 rows: 17.3 us against 33.4
 * matrix by matrix): same values within the fused calls' tolerance, bit-reproducible, not bit-identical to separate calls.
 * bnb_mi355x_gemm_4bit_grouped_route: which of these a group takes (2 / 1 / 0), from shapes alone (aligned pointers assumed). */
int bnb_mi355x_gemm_4bit_grouped_route(int dtype, int count, const int* N, int M, int K, int blocksize);
void bnb_mi355x_gemm_4bit_grouped(int dtype, const void* A, int count, const uint8_t* const* B, const float* const* absmax, const uint8_t* const* absmax_8bit, const float* const* absmax_code, const float* const* absmax_offset, void* const* out, const void* const* bias, const int* N, int M, int K, int blocksize, int quant_type, bnb_stream_t stream);

/* Fused backward of the 4-bit linear layer (MatMul4Bit.backward, reference bitsandbytes/autograd/_functions.py:365-386, which
 * dequantizes the whole weight and calls a dense matmul):
 *   grad_A[M, K] = grad_out[M, N] * dequant(B)[N, K]      dequant = T(code * scale), exactly dequantize_4bit's arithmetic
 * dtype 1 = fp16, 2 = bf16; argument meaning of the weight side as in cgemm_4bit_*. Requires N % 64 == 0, K % 128 == 0,
 * blocksize >= 64, 16-byte aligned grad_out / B (bnb_mi355x_gemm_4bit_grad_input_supported; an unsupported call is a fatal
 * error like a failed launch). workspace: device scratch of bnb_mi355x_gemm_4bit_grad_input_workspace_bytes(M, N, K) bytes
 * (fp32 slabs of the N slices; contents need no initialisation; with less the launch uses fewer slices). */
void bnb_mi355x_gemm_4bit_grad_input(int dtype, const void* grad_out, const uint8_t* B, const float* absmax, const uint8_t* absmax_8bit, const float* absmax_code, const float* absmax_offset, void* grad_A, int M, int N, int K, int blocksize, int quant_type, void* workspace, size_t workspace_bytes, bnb_stream_t stream);
size_t bnb_mi355x_gemm_4bit_grad_input_workspace_bytes(int M, int N, int K);
int bnb_mi355x_gemm_4bit_grad_input_supported(int dtype, int M, int N, int K, int blocksize);

/* One-shot all-gather of small per-rank outputs over peer-mapped buffers (bitsandbytes_amd/csrc/peer_gather.hip): the exchange
 * step of the N-sharded 4-bit linear layer at decode sizes (2.7 KB per rank), one kernel per collective, capturable in a hipGraph.
 * Nothing in the reference to mirror (it has no collective code); bitsandbytes_amd/peer.py is the host side.
 *   buffer_bytes : size of one rank's buffer for shards of up to max_bytes in a group of `world` (<= 8) ranks
 *   alloc / free : fine-grained device memory on the current device, zeroed
 *   export       : 64-byte hipIpc handle of an allocated buffer (0 = ok); open / close: map / unmap a peer's buffer
 *   allgather    : bufs = HOST array of `world` device pointers (rank r's buffer as mapped into this process, bufs[rank] local);
 *                  src = this rank's shard (bytes <= max_bytes), out = world x bytes, rank-major (all_gather_into_tensor's layout).
 *                  Every rank of the group must call it the same number of times with the same `bytes`.
 *   status       : 0, or 1 once a wait ran into its bound (tens of seconds; BNB_MI355X_PEER_WAIT_POLLS) because a peer never arrived
 *                  (synchronises the device). */
size_t bnb_mi355x_peer_buffer_bytes(int world, size_t max_bytes);
void* bnb_mi355x_peer_alloc(size_t bytes);
void bnb_mi355x_peer_free(void* buffer);
int bnb_mi355x_peer_export(void* buffer, void* handle64);
void* bnb_mi355x_peer_open(const void* handle64);
void bnb_mi355x_peer_close(void* mapped);
void bnb_mi355x_peer_allgather(void* const* bufs, int world, int rank, const void* src, void* out, size_t bytes, size_t max_bytes, bnb_stream_t stream);
int bnb_mi355x_peer_status(const void* local_buffer);

/* Peer chain: the all-gather of the N-sharded layer FUSED into the gemv launches on either side of it (M = 1 decode; fp16 /
 * bf16). Each rank owns an exchange buffer of bnb_mi355x_peer_chain_buffer_bytes(max_values) bytes (max_values a multiple of 4) from
 * bnb_mi355x_peer_chain_alloc (zeroed; fine_grained != 0 = hipDeviceMallocFinegrained, REQUIRED whenever the ranks sit on different
 * devices: remote stores into coarse-grained memory that a running kernel of the owner polls are outside what HIP guarantees;
 * 0 = ordinary cacheable memory, enough where all ranks share one device), exported / mapped / freed like the gather buffers above;
 * bufs[r] = rank r's buffer as mapped here. bnb_mi355x_gemv_4bit_peer_serves = the launcher's own shape check without a launch. y travels as
 * 8-byte granules {two consecutive values, u32 tag}: the producing launch stores them straight into every rank's buffer, the
 * consuming launch - the next layer - fetches them behind its weight requests and re-fetches the ones whose tag is not there
 * yet. No separate collective launch, no flag, no fence (csrc/gemv4_stream.hip, PeerChain).
 *   mode bit 0: x = the current exchange (K values; A is ignored)     bit 1: y (ns values of this rank, rank-major) goes to the
 *   exchange; it is also written to out_local[ns] when that is non-NULL. A launch with bit 1 completes one exchange.
 *   mode bit 3 (with bit 1), "gated": the launch runs over a matrix whose rows INTERLEAVE this rank's gate and up rows (row 2 r =
 *   gate row r, row 2 r + 1 = up row r; ns % 4 == 0) and what goes to the exchange is a[r] = T(T(silu(g_r)) * u_r) - ns / 2 values
 *   per rank, each op in fp32 and rounded once to T: torch's arithmetic for F.silu(g) * u on 16-bit tensors. One Llama-style FFN
 *   block = this launch over [gate; up], then a plain consuming launch over the down shard (K = F = world x ns / 2).
 *   wg_limit: at most that many workgroups (0 = one per CU) - ranks that share ONE device must be co-resident.
 *   epoch_word: 4 bytes of ORDINARY device memory owned by this rank, zeroed once (exchanges completed up to the last read-out;
 *   on the device because a replayed hipGraph re-runs its launches; only the read-out advances it).
 *   epoch_offset: exchanges completed (launches with bit 1) since the last bnb_mi355x_peer_chain_read on these buffers.
 * Every rank issues the same sequence of launches; a chain ends with bnb_mi355x_peer_chain_read (epoch_offset = the number of
 * exchanges since the previous read-out, this chain's included) and holds AT LEAST TWO exchanges: an exchange lives in the region
 * of its position in the chain (epoch_offset + 1 of the producing launch, modulo 64), only its tag carries the epoch. Returns 1 when launched, 0 when the problem is outside the form's
 * preconditions (ns even, K % 32 == 0, K <= 16384 with bit 0, one phase, blocksize >= 32, 16-byte aligned B / A, world * ns and
 * K <= max_values) - nothing was launched and the caller takes the unfused path. A wait that runs into its bound
 * (BNB_MI355X_PEER_WAIT_POLLS) sets the buffer's status word (bnb_mi355x_peer_status) and yields NaN, never a hang; once the word
 * is set every later wait on the buffer gives up after a few polls (a dead peer costs the bound once, not once per round and layer).
 * bnb_mi355x_peer_chain_read: the current exchange as a plain [nvalues] tensor (the end of a chain). */
size_t bnb_mi355x_peer_chain_buffer_bytes(long max_values);
void* bnb_mi355x_peer_chain_alloc(size_t bytes, int fine_grained);
int bnb_mi355x_gemv_4bit_peer_serves(int world, int ns, int K, int blocksize, int mode, long max_values, int wg_limit);
int bnb_mi355x_gemv_4bit_peer(void* const* bufs, void* epoch_word, int world, int rank, int dtype, const void* A, const uint8_t* B, const float* absmax, const uint8_t* absmax_8bit, const float* absmax_code, const float* absmax_offset, const void* bias, void* out_local, int ns, int K, int blocksize, int quant_type, int mode, long max_values, int wg_limit, int epoch_offset, bnb_stream_t stream);
void bnb_mi355x_peer_chain_read(void* const* bufs, void* epoch_word, int world, int rank, int dtype, void* out, int nvalues, long max_values, int epoch_offset, bnb_stream_t stream);

/* Tuning overrides for sweeps and tests (0 = built-in heuristic). reserved0: encoder of the 8-bit blockwise quantize - 1 =
 * cell-table kernel, 2 = byte-table kernel, anything else = by input size; 3 = the one-tile form of the 4-bit quantize kernel
 * everywhere (A/B of its pipelined FP4 form); 4 / 5 = 4 / 8 chunks per workgroup of the 4-bit quantize kernel instead of the shipped 2
 * (large NF4 inputs); 6 = one unit in flight per lane in the 8-bit dequantize kernel (its first form); 10 + v = tile / lane-mapping
 * variants of the 4-bit dequantize kernel (csrc/dequantize4.hip); reserved1: N slices of the fused backward (> 0; the
 * workspace-size query follows it). MFMA kernels: knob0 bit 0 = round 2's form of the register-transposed kernel (measurement build only; ignored by the product library),
 * knob0 bit 1 = the built-in route without the streaming MFMA kernel (round 5's routing: A/B runs), bit 2 = its weights-first experiment,
 * knob1 = 100 * cfg + K-slice count (cfg 11-14 producer/consumer geometries, 20/21/22
 * register-transposed kernel with built-in / 8 / 16 wavefronts, 40 K-quarter kernel, 50 streaming MFMA kernel). Every setting
 * computes correct results - the knobs only choose a launch geometry. THREAD-LOCAL: a setting applies to the calls the
 * SAME host thread makes afterwards and to nothing else in the process. */
void bnb_mi355x_set_tuning(int reserved0, int reserved1, int mfma_knob0, int mfma_knob1);

/* Sweep-only overrides of the streaming kernel (0 / -1 = built-in choice): ring depth (2, 3, 6; bf16 M = 1 fp32-absmax
 * NF4 only), 2048-k segments side by side, rows per workgroup, non-temporal weight loads (0 / 1, -1 = default on),
 * wavefronts per workgroup (8; same restriction as ring depth). Every setting computes the same results. Thread-local
 * like bnb_mi355x_set_tuning. */
void bnb_mi355x_set_stream_tuning(int ring_depth, int segments, int rows_per_workgroup, int nontemporal, int waves);

/* Profiling builds only (libbitsandbytes_mi355x_prof.so, -DBNB_PROFILING): when non-NULL the kernels write s_memtime
 * stamps per wavefront (u64) into this device buffer (tools/timeline_*.py). The product library contains no stamp or
 * ablation code; there the call is accepted and ignored. */
void bnb_mi355x_set_stamp_buffer(void* device_u64_buffer);

/* Version / build identification: returns "bitsandbytes_amd <ver> gfx950". */
const char* bnb_mi355x_version(void);

#ifdef __cplusplus
}
#endif

#endif /* BNB_MI355X_H */
