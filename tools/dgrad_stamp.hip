// tools/dgrad_stamp.hip -- DIAGNOSTIC: conv2 dgrad (CR=32 -> CO=16, 15x10, stride 1) with s_memtime stamps around the
// fragment-load and the MFMA segments of the K loop.  Re-states the product kernel's loop without sched_group_barrier.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifdef NO_STAMPS
#define STAMP(var) do { var = 0; } while (0)
#else
#define STAMP(var) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#endif

template <int MODE>   // 0: normal, 1: no MFMA (loads only), 2: no loads (MFMA only on stale regs)
__global__ __launch_bounds__(256) void k(const float *__restrict__ dz, const float *__restrict__ wgt, float *__restrict__ dx, int B,
                                         unsigned long long *st)
{
    constexpr int CR = 32, CO = 16, MW = 4, JJ = 2, H = 15, W = 10;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
    const long Mc = (long)B * H * W, m0 = ((long)blockIdx.x * 4 + wave) * 64;
    if (m0 >= Mc) return;
    int rbase[MW], rmask[MW];
    for (int mt = 0; mt < MW; ++mt) {
        const long m = m0 + 16 * mt + li;
        const int rem = (int)(m % (H * W)), b = (int)(m / (H * W)), a = rem / W, c = rem % W;
        const int oh0 = a + 1, ow0 = c + 1;
        rbase[mt] = ((b * H + oh0) * W + ow0) * CR + 4 * lq;
        int msk = 0;
        for (int t = 0; t < 3; ++t) { if (oh0 - t >= 0 && oh0 - t < H) msk |= 1 << t; if (ow0 - t >= 0 && ow0 - t < W) msk |= 256 << t; }
        rmask[mt] = m < Mc ? msk : 0;
    }
    f32x4 acc[MW];
    for (int i = 0; i < MW; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    int n_th = 0, n_tw = 0, n_jj = 0;
    unsigned long long tva = 0, tvm = 0;
    auto load = [&](float4 (&af)[MW], float4 &bf) {
        unsigned long long s0, s1, s2;
        STAMP(s0);
        const int tapoff = (n_th * W + n_tw) * CR - 16 * n_jj, tapbit = (1 << n_th) | (256 << n_tw);
        const bool live = n_th < 3;
        const int tap = live ? n_th * 3 + n_tw : 0;
        const float *ap[MW];
        for (int mt = 0; mt < MW; ++mt) {
            const bool ok = live && (rmask[mt] & tapbit) == tapbit;
            ap[mt] = ok ? dz + (rbase[mt] - tapoff) : wgt + 4 * lq;   // wgt is all zeros here
        }
        const float *bp = wgt + (tap * CO + li) * CR + 16 * n_jj + 4 * lq;
        asm volatile("" :: "v"(ap[0]), "v"(ap[1]), "v"(ap[2]), "v"(ap[3]), "v"(bp));
        STAMP(s1);
        for (int mt = 0; mt < MW; ++mt) af[mt] = *reinterpret_cast<const float4 *>(ap[mt]);
        bf = *reinterpret_cast<const float4 *>(bp);
        STAMP(s2);
        tva += s1 - s0; tvm += s2 - s1;
        if (++n_jj == JJ) { n_jj = 0; if (++n_tw == 3) { n_tw = 0; ++n_th; } }
    };
    auto mma = [&](const float4 (&af)[MW], const float4 &bf) {
        for (int mt = 0; mt < MW; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt].x, bf.x, acc[mt], 0, 0, 0);
        for (int mt = 0; mt < MW; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt].y, bf.y, acc[mt], 0, 0, 0);
        for (int mt = 0; mt < MW; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt].z, bf.z, acc[mt], 0, 0, 0);
        for (int mt = 0; mt < MW; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt].w, bf.w, acc[mt], 0, 0, 0);
    };
    unsigned long long t0, t1, t2, tl = 0, tm = 0, tstart, tend;
    float4 a0[MW], b0, a1[MW], b1;
    STAMP(tstart);
    load(a0, b0);
    for (int it = 0; it < 18; it += 2) {
        STAMP(t0);
        if (MODE != 2) load(a1, b1);
        STAMP(t1);
        if (MODE != 1) mma(a0, b0);
        STAMP(t2);
        tl += t1 - t0; tm += t2 - t1;
        STAMP(t0);
        if (MODE != 2) load(a0, b0);
        STAMP(t1);
        if (MODE != 1) mma(a1, b1);
        STAMP(t2);
        tl += t1 - t0; tm += t2 - t1;
    }
    STAMP(tend);
    for (int mt = 0; mt < MW; ++mt)
        for (int r = 0; r < 4; ++r) {
            const long m = m0 + 16 * mt + 4 * lq + r;
            if (m < Mc) dx[m * CO + li] = acc[mt][r] + (MODE == 1 ? a0[mt].x + a1[mt].y + b0.x + b1.x : 0.f);
        }
    if (lane == 0) {
        const long w = (long)blockIdx.x * 4 + wave;
        st[w * 3 + 0] = tva; st[w * 3 + 1] = tvm; st[w * 3 + 2] = tm;
    }
}

template <int MODE>
void run(const char *name, const float *dz, const float *w, float *dx, int B, unsigned long long *st)
{
    const long Mc = (long)B * 150, nblk = (Mc + 255) / 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(256), 0, 0, dz, w, dx, B, st);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(256), 0, 0, dz, w, dx, B, st);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(nblk * 4 * 3);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    double tl = 0, tm = 0, tt = 0;
    for (long i = 0; i < nblk * 4; ++i) { tl += h[i * 3]; tm += h[i * 3 + 1]; tt += h[i * 3 + 2]; }
    const double n = nblk * 4.0;
    printf("%-12s %.3f ms | per wave: addr VALU %7.0f cyc, load issue %7.0f cyc, mfma %7.0f cyc (19 loads, 18 mma)\n", name, ms, tl / n, tm / n, tt / n);
}

int main()
{
    const int B = 4096;
    float *dz, *w, *dx;
    unsigned long long *st;
    hipMalloc(&dz, sizeof(float) * B * 150 * 32);
    hipMalloc(&dx, sizeof(float) * B * 150 * 16);
    hipMalloc(&w, sizeof(float) * 9 * 16 * 32);
    hipMalloc(&st, 8 * 3 * 4 * 2500);
    hipMemset(dz, 0, sizeof(float) * B * 150 * 32);
    hipMemset(w, 0, sizeof(float) * 9 * 16 * 32);
    run<0>("normal", dz, w, dx, B, st);
    run<1>("loads only", dz, w, dx, B, st);
    run<2>("mfma only", dz, w, dx, B, st);
    return 0;
}
