// tools/mfma_calib.hip -- calibration microbenchmark (diagnostic, not product): sustained v_mfma_f32_16x16x4_f32 rate
// alone and with V extra VALU ops (fma or 32-bit integer multiply) per MFMA, at W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V, bool IMUL>
__global__ __launch_bounds__(256) void k(float *out, int iters, int seed)
{
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f + seed, b = 1.0001f;
    int x = threadIdx.x + seed, y = 3;
    float f = a;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < V; ++v) {
                if (IMUL) x = x * y + v; else f = fmaf(f, b, 1e-9f);
            }
        }
    }
    float s = f + (float)x;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int V, bool IMUL>
void run(const char *name, int blocks_per_cu)
{
    const int nblk = 256 * blocks_per_cu, iters = 2000;
    float *out;
    hipMalloc(&out, sizeof(float) * nblk * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<V, IMUL>), dim3(nblk), dim3(256), 0, 0, out, iters, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<V, IMUL>), dim3(nblk), dim3(256), 0, 0, out, iters, 2);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)nblk * 4 * iters * 8 * 2048.0;
    printf("%-28s waves/SIMD %d : %7.1f TFLOP/s (%.3f ms)\n", name, blocks_per_cu, flops / ms / 1e9, ms);
    hipFree(out);
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<0, false>("mfma only", w);
        run<2, false>("mfma + 2 fma", w);
        run<4, false>("mfma + 4 fma", w);
        run<6, false>("mfma + 6 fma", w);
        run<8, false>("mfma + 8 fma", w);
        run<2, true>("mfma + 2 imul", w);
        run<4, true>("mfma + 4 imul", w);
    }
    return 0;
}
