// Micro-benchmark: what a CU's vector-memory path delivers for the two operand streams of a fused dequantize-GEMM -
//   (A) the shared activation operand: every workgroup reads the SAME addresses (L2 / Infinity-Cache hits after first touch),
//       4 rows x 256 B per wave-instruction, like the activation stage of gemm4_mfma_ps;
//   (W) the private weight stream: every workgroup reads its own HBM-resident bytes once;
// alone and mixed 2 : 1 (the M = 64, 128-column tile ratio), as a function of wavefronts per workgroup and of the loads a
// wavefront keeps in flight. Prints bytes / cycle / CU (at the measured wall clock and an assumed 2.1 GHz) and TB/s.
//   hipcc --offload-arch=gfx950 -O3 -o l2_operand l2_operand.hip && ./l2_operand
#include <hip/hip_runtime.h>
#include <cstdio>

using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

// One iteration: UA loads of the shared operand + UW loads of the private stream per wavefront, all issued before the first
// use (so UA + UW loads per wavefront in flight), then consumed. `a_span` bytes of the shared operand are walked cyclically.
template <int THREADS, int UA, int UW>
__global__ __launch_bounds__(THREADS) void operand_kernel(const unsigned char* a, const unsigned char* w, float* out, int iters,
                                                          unsigned a_span, long w_per_wg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int WAVES = THREADS / 64;
    const auto rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a), 0, 0x7FFFFFFF, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(w) + blockIdx.x * w_per_wg, 0, 0x7FFFFFFF, 0x00020000);
    u32x4 acc = {0, 0, 0, 0};
    unsigned a_off = (wave * UA) * 1024u + lane * 16u;   // a wave-instruction = 1 KiB contiguous (4 x 256 B rows adjacent)
    unsigned w_off = (wave * UW) * 1024u + lane * 16u;
    for (int it = 0; it < iters; ++it) {
        u32x4 va[UA > 0 ? UA : 1], vw[UW > 0 ? UW : 1];
#pragma unroll
        for (int i = 0; i < UA; ++i)
            va[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, (a_off + i * 1024u) & (a_span - 1u), 0, 0));
#pragma unroll
        for (int i = 0; i < UW; ++i)
            vw[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_off + i * 1024u, 0, 0));
#pragma unroll
        for (int i = 0; i < UA; ++i)
            acc ^= va[i];
#pragma unroll
        for (int i = 0; i < UW; ++i)
            acc ^= vw[i];
        a_off += WAVES * UA * 1024u;
        w_off += WAVES * UW * 1024u;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u)
        out[blockIdx.x] = 1.0f;
}

template <typename F> float time_us(F launch, int reps) {
    hipStream_t s;
    hipStreamCreate(&s);
    launch(s);
    hipStreamSynchronize(s);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int r = 0; r < reps; ++r)
        launch(s);
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipStreamDestroy(s);
    return ms * 1e3f / reps;
}

int main() {
    const int G = 256; // one workgroup per CU
    float* out;
    hipMalloc(&out, 1 << 20);
    unsigned char *a, *w;
    const unsigned a_span = 1u << 20;            // 1 MiB shared operand (config 3's activations)
    const long w_per_wg = 4l << 20;              // 4 MiB private stream per workgroup: 1 GiB in all, HBM-resident
    hipMalloc(&a, a_span);
    hipMalloc(&w, w_per_wg * G);
    hipMemset(a, 1, a_span);
    hipMemset(w, 2, w_per_wg * G);
    printf("%-44s %9s %12s %12s %10s\n", "case (256 workgroups, one per CU)", "us", "B/clk/CU@2.1", "GB/s per CU", "TB/s chip");
#define RUN(T, UA, UW, NAME)                                                                                   \
    {                                                                                                          \
        const int waves = T / 64;                                                                              \
        const long per_it = static_cast<long>(waves) * (UA + UW) * 1024;                                       \
        const long w_it = static_cast<long>(waves) * (UW) * 1024;                                              \
        int iters = static_cast<int>(UW > 0 ? w_per_wg / w_it : (16l << 20) / per_it);                         \
        float us = time_us([&](hipStream_t s) { hipLaunchKernelGGL((operand_kernel<T, UA, UW>), dim3(G), dim3(T), 0, s, a, w, out, iters, a_span, w_per_wg); }, 5); \
        const double bytes = static_cast<double>(per_it) * iters;                                              \
        printf("%-44s %9.1f %12.1f %12.1f %10.2f\n", NAME, us, bytes / (us * 2100.0), bytes / us / 1e3, bytes * G / us / 1e6); \
    }
    RUN(512, 4, 0, "A only, 8 waves, 4 loads in flight / wave")
    RUN(512, 8, 0, "A only, 8 waves, 8 in flight")
    RUN(512, 16, 0, "A only, 8 waves, 16 in flight")
    RUN(256, 8, 0, "A only, 4 waves, 8 in flight")
    RUN(256, 16, 0, "A only, 4 waves, 16 in flight")
    RUN(1024, 8, 0, "A only, 16 waves, 8 in flight")
    RUN(512, 0, 2, "W only, 8 waves, 2 in flight")
    RUN(512, 0, 4, "W only, 8 waves, 4 in flight")
    RUN(512, 0, 8, "W only, 8 waves, 8 in flight")
    RUN(1024, 0, 4, "W only, 16 waves, 4 in flight")
    RUN(512, 4, 2, "A + W 2:1, 8 waves, 6 in flight")
    RUN(512, 8, 4, "A + W 2:1, 8 waves, 12 in flight")
    RUN(512, 12, 6, "A + W 2:1, 8 waves, 18 in flight")
    RUN(512, 2, 2, "A + W 1:1, 8 waves, 4 in flight")
    RUN(512, 4, 4, "A + W 1:1, 8 waves, 8 in flight")
    RUN(512, 8, 8, "A + W 1:1, 8 waves, 16 in flight")
    RUN(1024, 4, 2, "A + W 2:1, 16 waves, 6 in flight")
    return 0;
}
