// csrc/kws_layer1.h -- layer 1 of simple_cnn (Conv2D(16,3,'same',use_bias=False) -> BN -> ReLU6 -> MaxPool, cnn.py:27-34)
// without ever materialising its pre-BN output.
//
// conv1 has Cin = 1: z1 costs 9 FMAs per element to recompute from the 2.4 KB/clip feature map but 38.4 KB/clip to store,
// and the layer-by-layer schedule reads or writes a z1-sized tensor seven times per step.  These kernels keep the clip's
// zero-haloed feature map in LDS and recompute z1 where it is needed:
//   l1_stats_kernel      sum z, sum z^2 per channel                    (forward, batch statistics)
//   l1_act_pool_kernel   a1 = maxpool(relu6(z*scale + shift))          (forward)
//   l1_bwd_reduce_kernel sum g, sum g*xhat from da1 (pool/ReLU6 masks)  (backward)
//   l1_bwd_wgrad_kernel  dz = k1 (g - k2 - xhat k3); dW1 += x (*) dz     (backward; dz never leaves registers)
// The same device function produces z everywhere, so the ReLU6 / arg-max decisions agree bit for bit between passes.
#pragma once
#include "kws_layers.h"

namespace kws {

template <int COUT>
struct L1Thread {
    float w[9];
    __device__ __forceinline__ void load(const float *__restrict__ wk, int c)
    {
#pragma unroll
        for (int t = 0; t < 9; ++t) w[t] = wk[t * COUT + c];
    }
    // z at output pixel (oh, ow); xs is the (H+2) x WP zero-haloed map, WP = W + 2
    __device__ __forceinline__ float z(const float *xs, int WP, int oh, int ow) const
    {
        float o = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) o = fmaf(xs[(oh + t / 3) * WP + ow + t % 3], w[t], o);
        return o;
    }
};

__device__ __forceinline__ void l1_stage(const float *__restrict__ feat, float *xs, int b, int H, int W)
{
    const int WP = W + 2, n = (H + 2) * WP;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int r = i / WP - 1, c = i % WP - 1;
        xs[i] = (r >= 0 && r < H && c >= 0 && c < W) ? feat[(long)b * H * W + r * W + c] : 0.f;
    }
}

template <int COUT>
__global__ __launch_bounds__(256) void l1_stats_kernel(const float *__restrict__ feat, const float *__restrict__ wk, int B, int H,
                                                        int W, int clips_per_block, double *__restrict__ partial)
{
    extern __shared__ float xs[];
    constexpr int R = 256 / COUT;
    const int c = threadIdx.x % COUT, r = threadIdx.x / COUT, WP = W + 2;
    L1Thread<COUT> th;
    th.load(wk, c);
    double s = 0.0, ss = 0.0;
    for (int cb = 0; cb < clips_per_block; ++cb) {
        const int b = blockIdx.x * clips_per_block + cb;
        if (b >= B) break;
        __syncthreads();
        l1_stage(feat, xs, b, H, W);
        __syncthreads();
        float fs = 0.f, fss = 0.f;
        int oh = 0, ow = r;
        while (ow >= W) { ow -= W; ++oh; }
        for (int p = r; p < H * W; p += R) {
            const float z = th.z(xs, WP, oh, ow);
            fs += z;
            fss = fmaf(z, z, fss);
            ow += R;
            while (ow >= W) { ow -= W; ++oh; }
        }
        s += (double)fs;
        ss += (double)fss;
    }
    __shared__ double sh[2][256];
    sh[0][threadIdx.x] = s;
    sh[1][threadIdx.x] = ss;
    __syncthreads();
    if (r == 0) {
        for (int j = 1; j < R; ++j) { s += sh[0][j * COUT + c]; ss += sh[1][j * COUT + c]; }
        partial[((long)0 * COUT + c) * kStatStride + blockIdx.x] = s;
        partial[((long)1 * COUT + c) * kStatStride + blockIdx.x] = ss;
    }
}

template <int COUT>
__global__ __launch_bounds__(256) void l1_act_pool_kernel(const float *__restrict__ feat, const float *__restrict__ wk,
                                                           const float *__restrict__ scale, const float *__restrict__ shift,
                                                           float *__restrict__ a1, int B, int H, int W, int clips_per_block)
{
    extern __shared__ float xs[];
    constexpr int R = 256 / COUT;
    const int c = threadIdx.x % COUT, r = threadIdx.x / COUT, WP = W + 2, Hp = H / 2, Wp = W / 2;
    L1Thread<COUT> th;
    th.load(wk, c);
    const float sc = scale[c], sh = shift[c];
    for (int cb = 0; cb < clips_per_block; ++cb) {
        const int b = blockIdx.x * clips_per_block + cb;
        if (b >= B) break;
        __syncthreads();
        l1_stage(feat, xs, b, H, W);
        __syncthreads();
        for (int q = r; q < Hp * Wp; q += R) {
            const int ph = q / Wp, pw = q % Wp;
            const float y0 = fmaf(th.z(xs, WP, 2 * ph, 2 * pw), sc, sh), y1 = fmaf(th.z(xs, WP, 2 * ph, 2 * pw + 1), sc, sh);
            const float y2 = fmaf(th.z(xs, WP, 2 * ph + 1, 2 * pw), sc, sh), y3 = fmaf(th.z(xs, WP, 2 * ph + 1, 2 * pw + 1), sc, sh);
            a1[((long)b * Hp * Wp + q) * COUT + c] = relu6f(fmaxf(fmaxf(y0, y1), fmaxf(y2, y3)));
        }
    }
}

// the four outputs of a pool window, their first arg-max and the gradient routed to it
template <int COUT>
__device__ __forceinline__ void l1_window(const L1Thread<COUT> &th, const float *xs, int WP, int ph, int pw, float sc, float sh,
                                          float da, float (&z)[4], int &arg, float &g)
{
    z[0] = th.z(xs, WP, 2 * ph, 2 * pw);
    z[1] = th.z(xs, WP, 2 * ph, 2 * pw + 1);
    z[2] = th.z(xs, WP, 2 * ph + 1, 2 * pw);
    z[3] = th.z(xs, WP, 2 * ph + 1, 2 * pw + 1);
    float y[4], best;
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = fmaf(z[j], sc, sh);
    arg = 0;
    best = relu6f(y[0]);
#pragma unroll
    for (int j = 1; j < 4; ++j) {
        const float v = relu6f(y[j]);
        if (v > best) { best = v; arg = j; }
    }
    const float ya = arg == 0 ? y[0] : arg == 1 ? y[1] : arg == 2 ? y[2] : y[3];
    g = (ya > 0.f && ya < 6.f) ? da : 0.f;
}

template <int COUT>
__global__ __launch_bounds__(256) void l1_bwd_reduce_kernel(const float *__restrict__ feat, const float *__restrict__ wk,
                                                             const float *__restrict__ da1, BnCoef k, int B, int H, int W,
                                                             int clips_per_block, double *__restrict__ partial)
{
    extern __shared__ float xs[];
    constexpr int R = 256 / COUT;
    const int c = threadIdx.x % COUT, r = threadIdx.x / COUT, WP = W + 2, Hp = H / 2, Wp = W / 2;
    L1Thread<COUT> th;
    th.load(wk, c);
    const float sc = k.scale[c], sh = k.shift[c], mean = k.mean[c], inv = k.inv[c];
    double s = 0.0, sx = 0.0;
    for (int cb = 0; cb < clips_per_block; ++cb) {
        const int b = blockIdx.x * clips_per_block + cb;
        if (b >= B) break;
        __syncthreads();
        l1_stage(feat, xs, b, H, W);
        __syncthreads();
        float fs = 0.f, fsx = 0.f;
        for (int q = r; q < Hp * Wp; q += R) {
            float z[4], g;
            int arg;
            l1_window<COUT>(th, xs, WP, q / Wp, q % Wp, sc, sh, da1[((long)b * Hp * Wp + q) * COUT + c], z, arg, g);
            const float za = arg == 0 ? z[0] : arg == 1 ? z[1] : arg == 2 ? z[2] : z[3];
            fs += g;
            fsx = fmaf(g, (za - mean) * inv, fsx);
        }
        s += (double)fs;
        sx += (double)fsx;
    }
    __shared__ double shm[2][256];
    shm[0][threadIdx.x] = s;
    shm[1][threadIdx.x] = sx;
    __syncthreads();
    if (r == 0) {
        for (int j = 1; j < R; ++j) { s += shm[0][j * COUT + c]; sx += shm[1][j * COUT + c]; }
        partial[((long)0 * COUT + c) * kStatStride + blockIdx.x] = s;
        partial[((long)1 * COUT + c) * kStatStride + blockIdx.x] = sx;
    }
}

template <int COUT>
__global__ __launch_bounds__(256) void l1_bwd_wgrad_kernel(const float *__restrict__ feat, const float *__restrict__ wk,
                                                            const float *__restrict__ da1, BnCoef k, const float *__restrict__ gamma,
                                                            float *__restrict__ dw, int B, int H, int W, int clips_per_block)
{
    extern __shared__ float xs[];
    constexpr int R = 256 / COUT;
    const int c = threadIdx.x % COUT, r = threadIdx.x / COUT, WP = W + 2, Hp = H / 2, Wp = W / 2;
    L1Thread<COUT> th;
    th.load(wk, c);
    const float sc = k.scale[c], sh = k.shift[c], mean = k.mean[c], inv = k.inv[c];
    const float k1 = gamma[c] * inv, k2 = k.k2[c], k3 = k.k3[c];
    float acc[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto accumulate = [&](int oh, int ow, float dz) {
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = fmaf(xs[(oh + t / 3) * WP + ow + t % 3], dz, acc[t]);
    };
    for (int cb = 0; cb < clips_per_block; ++cb) {
        const int b = blockIdx.x * clips_per_block + cb;
        if (b >= B) break;
        __syncthreads();
        l1_stage(feat, xs, b, H, W);
        __syncthreads();
        for (int q = r; q < Hp * Wp; q += R) {
            const int ph = q / Wp, pw = q % Wp;
            float z[4], g;
            int arg;
            l1_window<COUT>(th, xs, WP, ph, pw, sc, sh, da1[((long)b * Hp * Wp + q) * COUT + c], z, arg, g);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dz = k1 * ((j == arg ? g : 0.f) - k2 - (z[j] - mean) * inv * k3);
                accumulate(2 * ph + (j >> 1), 2 * pw + (j & 1), dz);
            }
        }
        // pixels outside every pool window (odd H or W): g = 0 but BN still back-propagates through the statistics
        const int nb = (H - 2 * Hp) * W + (W - 2 * Wp) * 2 * Hp;
        for (int q = r; q < nb; q += R) {
            int oh, ow;
            if (q < (H - 2 * Hp) * W) { oh = 2 * Hp + q / W; ow = q % W; }
            else { const int e = q - (H - 2 * Hp) * W; oh = e / (W - 2 * Wp); ow = 2 * Wp + e % (W - 2 * Wp); }
            accumulate(oh, ow, k1 * (-k2 - (th.z(xs, WP, oh, ow) - mean) * inv * k3));
        }
    }
    __shared__ float shr[9][256];
#pragma unroll
    for (int t = 0; t < 9; ++t) shr[t][threadIdx.x] = acc[t];
    __syncthreads();
    if (threadIdx.x < 9 * COUT) {
        const int t = threadIdx.x / COUT, cc = threadIdx.x % COUT;
        float s = 0.f;
        for (int j = 0; j < R; ++j) s += shr[t][j * COUT + cc];
        atomicAdd(dw + t * COUT + cc, s);
    }
}

}  // namespace kws
