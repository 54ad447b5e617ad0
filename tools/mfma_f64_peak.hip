// mfma_f64_peak.hip — measures the sustained v_mfma_f64_16x16x4_f64 rate of the device (the
// roofline ceiling bench.py prices against).  hipcc --offload-arch=gfx950 -O3 -o mfma_f64_peak mfma_f64_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NACC> __global__ void peak(double *out, int iters, double a0, double b0) {
    v4d acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = v4d{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> void run(int waves_per_simd, int iters) {
    int threads = 256, blocks = 256 * waves_per_simd; // 4 waves per block -> 1 wave/SIMD per block/CU
    double *out;
    hipMalloc(&out, sizeof(double) * threads * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    peak<NACC><<<blocks, threads>>>(out, 10, 1.0, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    peak<NACC><<<blocks, threads>>>(out, iters, 1.000001, 0.999999);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * NACC * 2048.0;
    printf("nacc=%d waves/SIMD=%d : %.2f TFLOP/s  (%.3f ms)\n", NACC, waves_per_simd, flops / ms / 1e9, ms);
    hipFree(out);
}
int main() {
    for (int w = 1; w <= 4; w *= 2) {
        run<1>(w, 20000);
        run<4>(w, 5000);
        run<16>(w, 2000);
    }
    // sustained: the same kernel for ~0.25 s and ~1 s (power management settles within tens of ms)
    run<4>(2, 1000000);
    run<4>(2, 4000000);
    run<4>(4, 1000000);
    return 0;
}
