// gg_micro.hip — the MFMA block of gg_kernel<2,4,16> in isolation (no global loads): which part of the structure
// costs MFMA issue slots?  hipcc --offload-arch=gfx950 -O3 -o gg_micro gg_micro.hip
//   mode 0: MFMAs only (A from a register)      mode 1: + LDS reads of the A fragments
//   mode 2: + one workgroup barrier per chunk    mode 3: mode 2 with a short item (64 chunks) loop + tile store
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int MODE> __global__ __launch_bounds__(256, 2) void micro(double *out, int chunks, int items, double b0) {
    __shared__ double lds[2 * 128 * 16];
    const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    for (int i = threadIdx.x; i < 2 * 128 * 16; i += 256)
        lds[i] = 1e-9 * i;
    __syncthreads();
    for (int it = 0; it < items; it++) {
        v4d acc[8][2];
#pragma unroll
        for (int f = 0; f < 8; f++)
            acc[f][0] = acc[f][1] = v4d{0, 0, 0, 0};
        double b[2][4];
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int s = 0; s < 4; s++)
                b[q][s] = b0 + lane * 1e-9 + q + s;
        int buf = 0;
#pragma unroll 1
        for (int ch = 0; ch < chunks; ch++) {
            const double *As = lds + buf * 2048;
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int f = 0; f < 8; f++) {
                    double a = MODE == 0 ? b[0][s] + f : As[f * 256 + (4 * s + g) * 16 + c];
                    acc[f][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[0][s], acc[f][0], 0, 0, 0);
                    acc[f][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[1][s], acc[f][1], 0, 0, 0);
                }
            { // pin the issue order as gg_body does: reads run 6 fragments ahead of the MFMAs that consume them
                constexpr int MASK = MODE == 0 ? 0x002 : 0x100;
                __builtin_amdgcn_sched_group_barrier(MASK, 6, 0);
#pragma unroll
                for (int i = 0; i < 32 - 6; i++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(MASK, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
            }
            buf ^= 1;
            if (MODE >= 2)
                __syncthreads();
        }
        double *o = out + ((size_t)blockIdx.x * (MODE == 3 ? items : 1) + (MODE == 3 ? it : 0)) * 128 * 128;
        if (MODE == 3 || it == items - 1) {
#pragma unroll
            for (int q = 0; q < 2; q++)
#pragma unroll
                for (int f = 0; f < 8; f++)
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        o[(f * 16 + 4 * r + g) * 128 + (threadIdx.x >> 6) * 32 + q * 16 + c] = acc[f][q][r];
        }
    }
}
template <int MODE> void run(int chunks, int items) {
    int blocks = 512;
    double *out;
    (void)hipMalloc(&out, sizeof(double) * 128 * 128 * blocks * (MODE == 3 ? items : 1));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    micro<MODE><<<blocks, 256>>>(out, 4, 1, 1.0);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    micro<MODE><<<blocks, 256>>>(out, chunks, items, 0.999999);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * (double)chunks * items * 64 * 2048.0;
    printf("mode %d chunks %d items %d: %.2f TFLOP/s (%.2f ms)\n", MODE, chunks, items, flops / ms / 1e9, ms);
    (void)hipFree(out);
}
int main() {
    run<0>(64 * 64, 1);
    run<1>(64 * 64, 1);
    run<2>(64 * 64, 1);
    run<3>(64, 64);
    run<3>(16, 256);
    return 0;
}
