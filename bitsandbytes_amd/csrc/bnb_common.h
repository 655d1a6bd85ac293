// bnb_common.h — shared device-side helpers for the MI355X (gfx950 / CDNA4) 4-bit kernels.
//
// This library is written for gfx950 only: wave64, 256 CUs, 160 KiB LDS per CU. There is no CUDA
// path, no hipify layer and no multi-arch dispatch.
#pragma once

#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

namespace bnb {

// Loads / stores of the standalone streaming kernels (quantize / dequantize: every byte is touched once). NT selects the
// non-temporal cache policy. Measured on MI355X at 16.7 M elements (profiles/r3_stream_nt_policy_ab.txt): 16-bit quantize4
// 12.4 -> 11.0 us, dequantize4 10.0 -> 9.0, FP4 quantize 15.2 -> 13.3, the 8-bit pair -4 ... -5 %; but fp32 dequantize4
// 20.1 -> 26.7 us (a lane's 32 bytes leave as two 16-byte stores at a 32-byte stride: half lines written around the L2) and
// fp32 quantize4 +2.5 % - so the 4-bit kernels use it for 16-bit tensors only.
template <bool NT, typename V> __device__ __forceinline__ V stream_load(const V* ptr) {
    if constexpr (NT)
        return __builtin_nontemporal_load(ptr);
    else
        return *ptr;
}
template <bool NT, typename V> __device__ __forceinline__ void stream_store(V val, V* ptr) {
    if constexpr (NT)
        __builtin_nontemporal_store(val, ptr);
    else
        *ptr = val;
}

constexpr int kWave = 64;

// quant_type codes on the C ABI (reference csrc/common.h:3-7)
enum QuantType : int { kGeneral8bit = 0, kFP4 = 1, kNF4 = 2 };

// ---------------------------------------------------------------------------------------------
// Error convention of the reference ABI: the C entry points return void; a failed launch prints
// and terminates the process (reference csrc/compat.cuh:78-85, used at csrc/ops.cu:74,93,450).
// ---------------------------------------------------------------------------------------------
#define BNB_HIP_CHECK(expr)                                                                        \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            fprintf(stderr, "bitsandbytes_amd: HIP error %s at %s:%d\n", hipGetErrorString(_e),    \
                    __FILE__, __LINE__);                                                           \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

#define BNB_CHECK_LAUNCH() BNB_HIP_CHECK(hipPeekAtLastError())

// ---------------------------------------------------------------------------------------------
// Dynamic LDS above 64 KiB needs hipFuncAttributeMaxDynamicSharedMemorySize on the kernel. The attribute
// belongs to (kernel, device), so it is tracked per device: a process that drives several GPUs must set it
// on each of them. Raised monotonically, never lowered. Host-side; callable from several threads: the fast
// path is one relaxed atomic load, the (rare) raise is serialised so the attribute and the record agree.
// ---------------------------------------------------------------------------------------------
struct LdsLimit {
    static constexpr int kMaxDevices = 64;
    std::atomic<size_t> bytes[kMaxDevices] = {};
    std::mutex raise;
};
inline void ensure_dynamic_lds(LdsLimit& state, const void* kernel, size_t dynamic_bytes, size_t static_bytes = 0) {
    if (dynamic_bytes + static_bytes <= 64 * 1024)
        return;
    int dev = 0;
    BNB_HIP_CHECK(hipGetDevice(&dev));
    dev = (dev >= 0 && dev < LdsLimit::kMaxDevices) ? dev : 0;
    if (dynamic_bytes <= state.bytes[dev].load(std::memory_order_acquire))
        return;
    std::lock_guard<std::mutex> lock(state.raise);
    if (dynamic_bytes > state.bytes[dev].load(std::memory_order_relaxed)) {
        BNB_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(dynamic_bytes)));
        state.bytes[dev].store(dynamic_bytes, std::memory_order_release);
    }
}

// ---------------------------------------------------------------------------------------------
// Tuning knobs (bnb_mi355x_set_tuning / bnb_mi355x_set_stream_tuning: sweeps, tests, tools - every setting computes correct
// results) are THREAD-LOCAL: a tool or test that changes one and forgets to reset it changes the launches of its own thread
// only, never the routing of another thread's calls in the same process. (load / store keep the std::atomic spelling of
// the globals they replace.)
// ---------------------------------------------------------------------------------------------
struct TlsKnob {
    int v;
    int load(std::memory_order = std::memory_order_relaxed) const { return v; }
    void store(int x, std::memory_order = std::memory_order_relaxed) { v = x; }
};

// Which kernel family the calling thread's last gemm_4bit / gemv_4bit call launched (bnb_mi355x_last_gemm_kernel: tests assert
// that the kernel they name is the kernel that ran - a forced geometry that silently falls back to another family is not
// coverage). Written by the launchers themselves, at the point of the launch.
enum GemmKernelId { kKernelNone = 0, kKernelStream = 1, kKernelGeneric = 2, kKernelRt = 3, kKernelPc = 4, kKernelPs = 5, kKernelKq = 6, kKernelSm = 7, kKernelTall = 8 };
extern thread_local int g_last_gemm_kernel;

// ---------------------------------------------------------------------------------------------
// Element types. fp16/bf16 travel as their native clang types so that conversions lower to the
// gfx950 hardware converts (v_cvt_f16_f32 / v_cvt_pk_bf16_f32, both round-to-nearest-even).
// ---------------------------------------------------------------------------------------------
using f16 = _Float16;
using bf16 = __bf16;

template <typename T> struct TypeInfo;
template <> struct TypeInfo<float> {
    static constexpr int bytes = 4;
};
template <> struct TypeInfo<f16> {
    static constexpr int bytes = 2;
};
template <> struct TypeInfo<bf16> {
    static constexpr int bytes = 2;
};

template <typename T> __device__ __forceinline__ float to_f32(T v) { return static_cast<float>(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return static_cast<T>(v); }

// ---------------------------------------------------------------------------------------------
// 4-bit code tables (reference bitsandbytes/functional.py:788-823). FP4 is the raw table divided
// by 12 in fp32, exactly as `data.div_(data.abs().max())` does (functional.py:853); entry 8 is
// +0.0 as in the Python table.
// ---------------------------------------------------------------------------------------------
#define BNB_NF4_VALUES                                                                             \
    -1.0f, -0.6961928009986877f, -0.5250730514526367f, -0.39491748809814453f,                      \
        -0.28444138169288635f, -0.18477343022823334f, -0.09105003625154495f, 0.0f,                 \
        0.07958029955625534f, 0.16093020141124725f, 0.24611230194568634f, 0.33791524171829224f,    \
        0.44070982933044434f, 0.5626170039176941f, 0.7229568362236023f, 1.0f

#define BNB_FP4_VALUES                                                                             \
    0.0f / 12.0f, 0.0625f / 12.0f, 8.0f / 12.0f, 12.0f / 12.0f, 4.0f / 12.0f, 6.0f / 12.0f,       \
        2.0f / 12.0f, 3.0f / 12.0f, 0.0f / 12.0f, -0.0625f / 12.0f, -8.0f / 12.0f, -12.0f / 12.0f, \
        -4.0f / 12.0f, -6.0f / 12.0f, -2.0f / 12.0f, -3.0f / 12.0f

// Device-resident copies. Deliberately plain __device__ (global address space), not __constant__:
// kernels select between these and a caller-supplied table pointer, and mixing address spaces would
// turn the gather into flat_load, whose completion the compiler can only await with vmcnt(0).
__device__ static const float kNF4Code[16] = {BNB_NF4_VALUES};
__device__ static const float kFP4Code[16] = {BNB_FP4_VALUES};

// explicit global-address-space view of a device pointer (forces global_load, which vmcnt can count)
typedef const float __attribute__((address_space(1))) * gfloat_ptr;

// The reference rounds `code * absmax` to fp32 first and to T second (csrc/cpu_ops.cpp:419-431).
// hipcc would otherwise fuse the multiply and the fp16 convert into v_fma_mix{lo,hi}_f16, which
// rounds once and differs from the two-step result in rare double-rounding cases. An empty asm
// makes the fp32 product an opaque register value: no instruction is emitted, the fusion is blocked.
__device__ __forceinline__ float rounded_f32(float v) {
    asm volatile("" : "+v"(v));
    return v;
}

// The fp32 scale of a double-quantised block, absmax[b] = code2[q] * absmax2 + offset, with the TWO roundings of the host-side
// sequence (dequantize_blockwise, then `+= offset`: reference bitsandbytes/functional.py:1002-1006). Written as
// __fadd_rn(__fmul_rn(c, a), offset) hipcc 7.2 emits ONE v_fma_f32 (the intrinsics are plain * and + by the time the contraction pass
// runs): one rounding, a last-bit difference in ~1/5 of the scales - enough to flip a bf16 output of a 14336-row layer every few
// calls, which is how it was found (the sharded FFN block carries un-nested statistics and is compared bit for bit with the unsharded
// block). The empty asm (not volatile: it may be scheduled freely) makes the product opaque.
__device__ __forceinline__ float nested_scale(float c, float a, float offset) {
    float prod = c * a;
    asm("" : "+v"(prod));
    return prod + offset;
}

// A zero the compiler cannot see through, materialised at the point of the call. Adding it to the
// shift amounts of the nibble/byte extraction ties the whole decode to program order *after* this
// point: without it LLVM hoists the first v_bfe_u32 of the decode above the workgroup barrier and
// into the middle of the load-issue sequence, where its s_waitcnt vmcnt stalls the wavefront for a
// full HBM round trip before the rest of its loads are even issued (seen in the ISA and as +1 us).
__device__ __forceinline__ int opaque_zero() {
    int z;
    asm volatile("s_mov_b32 %0, 0" : "=s"(z));
    return z;
}

// ---------------------------------------------------------------------------------------------
// wave64 reductions
// ---------------------------------------------------------------------------------------------
template <int WIDTH> __device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int off = 1; off < WIDTH; off <<= 1)
        v = fmaxf(v, __shfl_xor(v, off, kWave));
    return v;
}

// Sum over the 64 lanes, result valid in every lane. All cross-lane traffic stays in the VALU (DPP
// modifiers + v_readlane): no ds_bpermute round trips through the LDS pipe, which measured ~0.5 us
// for the two reductions at the end of the M = 1 gemv (s_memtime timeline, profiles/).
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    // v + (v moved by DPP control CTRL); all rows/banks enabled, bound_ctrl irrelevant (full wave active)
    const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true);
    return v + __builtin_bit_cast(float, moved);
}

__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add<0xB1>(v);  // quad_perm [1,0,3,2]  : lane ^ 1
    v = dpp_add<0x4E>(v);  // quad_perm [2,3,0,1]  : lane ^ 2
    v = dpp_add<0x141>(v); // row_half_mirror      : the other quad of each 8
    v = dpp_add<0x140>(v); // row_mirror           : the other half of each 16-lane row
    // every lane of a row now holds its row sum; combine the four rows through SGPRs
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (r0 + r1) + (r2 + r3);
}

// C independent wave sums in lock step: every DPP stage runs over all C values before the next one, so the chains
// overlap instead of paying the DPP / readlane hazards C times in a row.
template <int C> __device__ __forceinline__ void wave_sum_n(float (&v)[C]) {
#pragma unroll
    for (int c = 0; c < C; ++c)
        v[c] = dpp_add<0xB1>(v[c]);
#pragma unroll
    for (int c = 0; c < C; ++c)
        v[c] = dpp_add<0x4E>(v[c]);
#pragma unroll
    for (int c = 0; c < C; ++c)
        v[c] = dpp_add<0x141>(v[c]);
#pragma unroll
    for (int c = 0; c < C; ++c)
        v[c] = dpp_add<0x140>(v[c]);
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[c]), 0));
        const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[c]), 16));
        const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[c]), 32));
        const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[c]), 48));
        v[c] = (r0 + r1) + (r2 + r3);
    }
}

// Compute units of the current device (cached per device); 256 (MI355X) when no device can be queried - the launch-plan
// functions that use it are also reachable from pure size queries (workspace bytes) on a host without a GPU.
inline int device_cu_count_or_default() {
    static std::atomic<int> cached[LdsLimit::kMaxDevices] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return 256;
    }
    const int slot = (dev >= 0 && dev < LdsLimit::kMaxDevices) ? dev : 0;
    int v = cached[slot].load(std::memory_order_relaxed);
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) {
            (void)hipGetLastError();
            v = 256;
        }
        cached[slot].store(v, std::memory_order_relaxed);
    }
    return v;
}

static inline int ilog2(int v) {
    int s = 0;
    while ((1 << s) < v)
        ++s;
    return s;
}

static inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

static inline bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

} // namespace bnb
