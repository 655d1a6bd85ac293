// Micro-benchmark: what the instruction sequences of the fused dequantize-GEMM phases cost on one SIMD of gfx950, alone and
// beside a partner wavefront on the SAME SIMD that runs the other phase (s_memtime brackets, one workgroup of 8 wavefronts per
// CU: wavefronts w and w + 4 share a SIMD). Sequences (per "round", repeated R times inside the bracket):
//   pkmul  : 32 independent v_pk_mul_f32          cvt    : 32 v_cvt_pk_bf16_f32           mul2   : 64 v_mul_f32
//   perm   : 32 v_perm_b32                        decode : 32 x (perm, ds_read_b64), then 32 x (pk_mul, cvt_pk)  [a decode phase]
//   mfma   : 16 v_mfma_f32_32x32x16_bf16 in two dependent chains                           [an MFMA phase without operand reads]
//   mfmaA  : 16 ds_read_b128 issued up front + the 16 MFMAs consuming them                 [an MFMA phase]
// Cases: X alone on its SIMD (partner idle at a barrier), X beside X, X beside Y.
//   hipcc --offload-arch=gfx950 -O3 -o inst_cost inst_cost.hip && ./inst_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

enum Seq { IDLE = 0, PKMUL, CVT, MUL2, PERM, DECODE, MFMA, MFMAA };

template <int SEQ> __device__ __forceinline__ void run_seq(unsigned char* smem, int lane, float& sink, f32x16 (&acc)[2], uint32_t seed) {
    if constexpr (SEQ == PKMUL) {
        f32x2 v[32];
#pragma unroll
        for (int i = 0; i < 32; ++i)
            v[i] = f32x2{sink + i, sink - i};
        const f32x2 s = {sink, sink};
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < 32; ++i)
            v[i] = v[i] * s;
#pragma unroll
        for (int i = 0; i < 32; ++i)
            sink += v[i][0] + v[i][1];
    } else if constexpr (SEQ == CVT) {
        float v[64];
#pragma unroll
        for (int i = 0; i < 64; ++i)
            v[i] = sink + i;
        uint32_t o[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            using V = __attribute__((ext_vector_type(2))) __bf16;
            V t;
            t[0] = static_cast<__bf16>(v[2 * i]);
            t[1] = static_cast<__bf16>(v[2 * i + 1]);
            o[i] = __builtin_bit_cast(uint32_t, t);
        }
        uint32_t x = 0;
#pragma unroll
        for (int i = 0; i < 32; ++i)
            x ^= o[i];
        sink += __builtin_bit_cast(float, x & 0x3f800000u);
    } else if constexpr (SEQ == MUL2) {
        float v[64];
#pragma unroll
        for (int i = 0; i < 64; ++i)
            v[i] = sink + i;
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            v[i] = v[i] * sink;
            asm volatile("" : "+v"(v[i]));
        }
#pragma unroll
        for (int i = 0; i < 64; ++i)
            sink += v[i];
    } else if constexpr (SEQ == PERM) {
        uint32_t x = seed, o = 0;
#pragma unroll
        for (int i = 0; i < 32; ++i)
            o ^= __builtin_amdgcn_perm(x + i, lane * 8u, 0x0C0C0400u + ((i & 3) << 8));
        sink += __builtin_bit_cast(float, o & 0x3f800000u);
    } else if constexpr (SEQ == DECODE) {
        // 8 dwords -> 32 bytes -> 32 look-ups of (fp32, fp32) in a 64-KiB bank-private table at LDS address 0
        uint32_t w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            w[i] = seed * (2654435761u + i) + lane * 40503u;
        f32x2 pr[32];
#pragma unroll
        for (int i = 0; i < 32; ++i)
            pr[i] = *reinterpret_cast<const __attribute__((address_space(3))) f32x2*>(
                __builtin_amdgcn_perm(w[i >> 2], (lane & 31) * 8u, 0x0C0C0400u + ((i & 3) << 8)));
        const f32x2 s = {sink, sink};
        uint32_t x = 0;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const f32x2 p = pr[i] * s;
            using V = __attribute__((ext_vector_type(2))) __bf16;
            V t;
            t[0] = static_cast<__bf16>(p[0]);
            t[1] = static_cast<__bf16>(p[1]);
            x ^= __builtin_bit_cast(uint32_t, t);
        }
        sink += __builtin_bit_cast(float, x & 0x3f800000u);
    } else if constexpr (SEQ == MFMA || SEQ == MFMAA) {
        u32x4 af[16], bf[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            bf[i] = u32x4{seed + i, seed, seed, seed};
        if constexpr (SEQ == MFMAA) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                af[i] = *reinterpret_cast<const u32x4*>(smem + 65536 + ((lane & 31) * 256 + (((lane >> 5) * 8 + (i >> 1)) ^ (lane & 15)) * 16) + (i & 1) * 8192);
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                af[i] = u32x4{seed, seed + i, seed, seed};
        }
#pragma unroll
        for (int i = 0; i < 16; ++i)
            acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i]), __builtin_bit_cast(bf16x8, bf[i >> 1]), acc[i & 1], 0, 0, 0);
    }
}

// waves 0-3 run SA, waves 4-7 run SB, ROUNDS times each between two barriers; stamps: [wave][2]
template <int SA, int SB> __global__ __launch_bounds__(512) void cost_kernel(unsigned long long* stamps, float* out, int rounds, uint32_t seed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < (96 * 1024) / 4; i += 512)
        reinterpret_cast<float*>(smem)[i] = 1.0f + (i & 7);
    float sink = 1.0f + seed * 1e-9f;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i)
        acc[0][i] = acc[1][i] = 0.0f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        for (int r = 0; r < rounds; ++r)
            run_seq<SA>(smem, lane, sink, acc, seed + r);
    } else {
        for (int r = 0; r < rounds; ++r)
            run_seq<SB>(smem, lane, sink, acc, seed + r);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float a = sink;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        a += acc[0][i] + acc[1][i];
    if (a == 123.456f)
        out[0] = a;
    if (lane == 0)
        stamps[(blockIdx.x * 8 + wave)] = t1 - t0;
}

template <int SA, int SB> void run(const char* name, unsigned long long* d_st, float* d_out) {
    const int G = 256, rounds = 8;
    auto kern = cost_kernel<SA, SB>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL(kern, dim3(G), dim3(512), 96 * 1024, 0, d_st, d_out, rounds, 12345u + i);
    hipDeviceSynchronize();
    std::vector<unsigned long long> st(G * 8);
    hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> a, b;
    for (int g = 0; g < G; ++g)
        for (int w = 0; w < 8; ++w)
            (w < 4 ? a : b).push_back(static_cast<double>(st[g * 8 + w]) / rounds);
    std::sort(a.begin(), a.end());
    std::sort(b.begin(), b.end());
    printf("%-34s  first group: median %7.0f p90 %7.0f   second group: median %7.0f p90 %7.0f   ticks per round\n", name,
           a[a.size() / 2], a[a.size() * 9 / 10], b[b.size() / 2], b[b.size() * 9 / 10]);
}

int main() {
    unsigned long long* d_st;
    float* d_out;
    hipMalloc(&d_st, 256 * 8 * 8);
    hipMalloc(&d_out, 64);
    run<PKMUL, IDLE>("32 pk_mul | idle", d_st, d_out);
    run<CVT, IDLE>("32 cvt_pk_bf16 | idle", d_st, d_out);
    run<MUL2, IDLE>("64 v_mul_f32 | idle", d_st, d_out);
    run<PERM, IDLE>("32 v_perm | idle", d_st, d_out);
    run<DECODE, IDLE>("decode phase | idle", d_st, d_out);
    run<MFMA, IDLE>("16 mfma 32x32x16 | idle", d_st, d_out);
    run<MFMAA, IDLE>("16 ds_read_b128 + 16 mfma | idle", d_st, d_out);
    run<DECODE, DECODE>("decode | decode", d_st, d_out);
    run<MFMA, MFMA>("16 mfma | 16 mfma", d_st, d_out);
    run<MFMAA, MFMAA>("mfma phase | mfma phase", d_st, d_out);
    run<DECODE, MFMA>("decode | 16 mfma", d_st, d_out);
    run<DECODE, MFMAA>("decode | mfma phase", d_st, d_out);
    run<PKMUL, MFMA>("32 pk_mul | 16 mfma", d_st, d_out);
    run<CVT, MFMA>("32 cvt | 16 mfma", d_st, d_out);
    run<MUL2, MFMA>("64 v_mul | 16 mfma", d_st, d_out);
    return 0;
}
