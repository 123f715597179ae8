// Cost of folding per-block double partial sums into per-channel accumulators with global_atomic_add_f64 (the finalize-free
// BatchNorm statistics): G blocks each add N doubles (N addresses, optionally S slots selected by blockIdx % S) at their END, after a
// delay loop so that all blocks arrive in a burst as the persistent kernels' blocks do.
// hipcc -O3 --offload-arch=gfx950 tools/atomic_calib.hip -o /tmp/atomic_calib && /tmp/atomic_calib
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void burst(double *acc, int N, int S, int spin, int mode, double *partial)
{
    float x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = fmaf(x, 1.0001f, 0.5f);
    if (x == 12345.f) acc[0] = 1;
    __syncthreads();
    if (mode == 0) { for (int i = threadIdx.x; i < N; i += blockDim.x) atomicAdd(acc + (blockIdx.x % S) * N + i, 1.0); }
    else { for (int i = threadIdx.x; i < N; i += blockDim.x) partial[(long)i * gridDim.x + blockIdx.x] = 1.0; }
}
int main()
{
    double *acc, *partial;
    hipMalloc(&acc, 64 * 1024 * 8); hipMalloc(&partial, 1024L * 1024 * 8);
    hipMemset(acc, 0, 64 * 1024 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int Gs[] = {256, 512, 1024}, Ns[] = {64, 128, 256}, Ss[] = {1, 8};
    for (int mode = 0; mode < 2; ++mode)
    for (int G : Gs) for (int N : Ns) for (int S : Ss) {
        if (mode == 1 && S != 1) continue;
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipLaunchKernelGGL(burst, dim3(G), dim3(256), 0, 0, acc, N, S, 2000, mode, partial);
            hipEventRecord(e0);
            for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(burst, dim3(G), dim3(256), 0, 0, acc, N, S, 2000, mode, partial);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms / 20 < best) best = ms / 20;
        }
        printf("%s G=%4d N=%3d S=%d  %.2f us/launch\n", mode ? "partials" : "atomics ", G, N, S, best * 1e3f);
    }
    return 0;
}
