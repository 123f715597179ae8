// tools/dgrad_modes.hip -- DIAGNOSTIC: the PRODUCT dgrad kernel (kws_conv.h) built three ways via KWS_DGRAD_MODE
// (0 normal, 1 loads only, 2 MFMAs only) and timed on conv2's data gradient at B = 4096.
//   for m in 0 1 2; do hipcc -O3 -std=c++17 --offload-arch=gfx950 -DKWS_DGRAD_MODE=$m -I tf-keras-speech-commands_amd/csrc tools/dgrad_modes.hip -o tools/dgrad_mode$m.bin; done
#include <cstdio>
#include "kws_conv.h"
using namespace kws;
int main()
{
    const int B = 4096, H = 15, W = 10;
    float *dz, *w, *dx, *zeros;
    hipMalloc(&dz, sizeof(float) * B * H * W * 32);
    hipMalloc(&dx, sizeof(float) * B * H * W * 16);
    hipMalloc(&w, sizeof(float) * 9 * 16 * 32);
    hipMalloc(&zeros, 4096);
    hipMemset(dz, 0, sizeof(float) * B * H * W * 32);
    hipMemset(w, 0, sizeof(float) * 9 * 16 * 32);
    hipMemset(zeros, 0, 4096);
    ConvGeom g{B, H, W, H, W, 1, 1, 1, 3, 3};
    DgradClass c{0, 0, H, W};
    const long Mc = (long)B * H * W;
    const unsigned nblk = (unsigned)((Mc + 255) / 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((conv_dgrad_direct_kernel<32, 16, 4, 1>), dim3(nblk), dim3(256), 0, 0, dz, w, dx, zeros, g, c);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((conv_dgrad_direct_kernel<32, 16, 4, 1>), dim3(nblk), dim3(256), 0, 0, dz, w, dx, zeros, g, c);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("KWS_DGRAD_MODE=%d : %.4f ms per launch\n", KWS_DGRAD_MODE, ms / 10);
    return 0;
}
