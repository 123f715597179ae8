// csrc/kws_lite.h -- kernels specific to simple_cnn_lite (classifier/models/cnn.py:77-141):
// SeparableConv2D(filters, 3, strides, 'same', use_bias=True[, activation='relu']) = depthwise 3x3 (multiplier 1, no bias)
// followed by a pointwise 1x1 convolution with bias.  The pointwise part runs on the MFMA kernels of kws_conv.h
// (1x1 geometry); this file holds the depthwise kernels (HBM-bound, one element per thread, NHWC so channels coalesce)
// and the degenerate first stage (Cin = 1).
#pragma once
#include "kws_conv.h"
#include "kws_layers.h"

namespace kws {

// dw[b,oh,ow,c] = sum_tap x[b, oh*s+kh-pt, ow*s+kw-pl, c] * k[tap][c]
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const float *__restrict__ x, const float *__restrict__ k,
                                                          float *__restrict__ y, ConvGeom g, int C)
{
    const long total = (long)g.B * g.Ho * g.Wo * C, idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    long t = idx / C;
    const int ow = (int)(t % g.Wo);
    t /= g.Wo;
    const int oh = (int)(t % g.Ho), b = (int)(t / g.Ho);
    float acc = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int ih = oh * g.stride + kh - g.pt, iw = ow * g.stride + kw - g.pl;
            if (ih >= 0 && ih < g.H && iw >= 0 && iw < g.W) acc = fmaf(x[(((long)b * g.H + ih) * g.W + iw) * C + c], k[(kh * 3 + kw) * C + c], acc);
        }
    y[idx] = acc;
}

// dx[b,ih,iw,c] = sum_tap dy[b, (ih+pt-kh)/s, (iw+pl-kw)/s, c] * k[tap][c]   (only where the division is exact)
__global__ __launch_bounds__(256) void dwconv_dgrad_kernel(const float *__restrict__ dy, const float *__restrict__ k,
                                                            float *__restrict__ dx, ConvGeom g, int C)
{
    const long total = (long)g.B * g.H * g.W * C, idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    long t = idx / C;
    const int iw = (int)(t % g.W);
    t /= g.W;
    const int ih = (int)(t % g.H), b = (int)(t / g.H);
    float acc = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int ty = ih + g.pt - kh, tx = iw + g.pl - kw;
            if (ty >= 0 && tx >= 0 && ty % g.stride == 0 && tx % g.stride == 0) {
                const int oh = ty / g.stride, ow = tx / g.stride;
                if (oh < g.Ho && ow < g.Wo) acc = fmaf(dy[(((long)b * g.Ho + oh) * g.Wo + ow) * C + c], k[(kh * 3 + kw) * C + c], acc);
            }
        }
    dx[idx] = acc;
}

// dk[tap][c] = sum_m x[pix(m)+tap][c] * dy[m][c]: per-channel reduction in double, partial[(tap*C + c)*kStatStride + blk]
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ dy, ConvGeom g,
                                                            int C, int rows_per_block, double *__restrict__ partial)
{
    const int c = threadIdx.x % C, r = threadIdx.x / C, R = 256 / C;
    const long M = (long)g.B * g.Ho * g.Wo, beg = (long)blockIdx.x * rows_per_block;
    const long end = beg + rows_per_block < M ? beg + rows_per_block : M;
    float acc[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long m = beg + r; m < end; m += R) {
        const float d = dy[m * C + c];
        const int pix = (int)(m % ((long)g.Ho * g.Wo)), b = (int)(m / ((long)g.Ho * g.Wo)), oh = pix / g.Wo, ow = pix % g.Wo;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ih = oh * g.stride + t / 3 - g.pt, iw = ow * g.stride + t % 3 - g.pl;
            if (ih >= 0 && ih < g.H && iw >= 0 && iw < g.W) acc[t] = fmaf(x[(((long)b * g.H + ih) * g.W + iw) * C + c], d, acc[t]);
        }
    }
    __shared__ float sh[9][256];
#pragma unroll
    for (int t = 0; t < 9; ++t) sh[t][threadIdx.x] = acc[t];
    __syncthreads();
    for (int i = threadIdx.x; i < 9 * C; i += 256) {
        const int t = i / C, cc = i % C;
        double s = 0.0;
        for (int j = 0; j < R; ++j) s += (double)sh[t][j * C + cc];
        partial[((long)t * C + cc) * kStatStride + blockIdx.x] = s;
    }
}

// ---- first stage, Cin = 1: z[m][co] = dw[m] * pw[co] + bias[co] ---------------------------------------------------
template <int COUT>
__global__ __launch_bounds__(256) void pw1_fwd_kernel(const float *__restrict__ dw, const float *__restrict__ pw,
                                                       const float *__restrict__ bias, float *__restrict__ z, long M)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * COUT) return;
    const int co = (int)(idx % COUT);
    z[idx] = fmaf(dw[idx / COUT], pw[co], bias[co]);
}

// ddw[m] = sum_co dz[m][co] pw[co];  partial sums of dpw[co] = sum_m dw[m] dz[m][co] and db[co] = sum_m dz[m][co]
template <int COUT>
__global__ __launch_bounds__(256) void pw1_bwd_kernel(const float *__restrict__ dw, const float *__restrict__ pw,
                                                       const float *__restrict__ dz, float *__restrict__ ddw, long M,
                                                       int rows_per_block, double *__restrict__ partial)
{
    constexpr int R = 256 / COUT;
    const int co = threadIdx.x % COUT, r = threadIdx.x / COUT;
    const long beg = (long)blockIdx.x * rows_per_block;
    const long end = beg + rows_per_block < M ? beg + rows_per_block : M;
    const float p = pw[co];
    double s = 0.0, sb = 0.0;
    for (long m = beg + r; m < end; m += R) {
        const float d = dz[m * COUT + co];
        float v = d * p;                                   // reduce over the COUT lanes of this row (COUT = 16)
#pragma unroll
        for (int o = COUT / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, COUT);
        if (co == 0) ddw[m] = v;
        s += (double)dw[m] * (double)d;
        sb += (double)d;
    }
    __shared__ double sh[2][256];
    sh[0][threadIdx.x] = s;
    sh[1][threadIdx.x] = sb;
    __syncthreads();
    if (r == 0) {
        for (int j = 1; j < R; ++j) { s += sh[0][j * COUT + co]; sb += sh[1][j * COUT + co]; }
        partial[((long)0 * COUT + co) * kStatStride + blockIdx.x] = s;
        partial[((long)1 * COUT + co) * kStatStride + blockIdx.x] = sb;
    }
}

// out[c] = sum over blocks of partial[(which*C + c)*kStatStride + blk]
__global__ void partial_finalize_kernel(const double *__restrict__ partial, int nblk, int C, int which, float *__restrict__ out)
{
    __shared__ double sh[256];
    const int c = blockIdx.x;
    const double s = block_sum_partials(partial, which, C, c, nblk, sh);
    if (threadIdx.x == 0) out[c] = (float)s;
}

// ---------------------------------------------------------------------------------------------------------------------
// Inference front of simple_cnn_lite: the first two SeparableConv2D -> BatchNorm -> ReLU6 -> MaxPool stages
// (cnn.py:85-104) in ONE kernel, one wave per clip, nothing between the feature map and a2 ever touches HBM.
// (The layer-by-layer path spends six passes and 1.05 of the 2.04 ms of a B = 16384 forward on these two stages.)
//
// Per clip, in the wave's private LDS tiles:
//   xs  (H+2)(W+2)          zero-haloed features
//   d1  H*W                 depthwise 1 (one channel, 9 FMA per pixel, lane = pixel)
//   a1  (H1+2)(W1+2) x 16   pool(relu6(bn(d1 * pw1 + b1))), zero halo, lane = (window group, channel)
//   d2  (4*ceil(n2/4)*4) x 17   depthwise 2 of the pixels that a pool window of stage 2 uses, stored window-major
//                           (pixel p = 4*window + element) with a 17-float row so the MFMA A reads are conflict-free
//   pointwise 2 (16 -> 32) = 8 v_mfma_f32_16x16x4_f32 per 16-pixel tile; in the D layout a lane holds the four elements
//   of one window for one channel, so bias, BN, ReLU6 and the 2x2 max stay in registers; a2 goes to global memory.
// BatchNorm is the inference affine (scale, shift from the moving statistics, bn_infer_coef_kernel).
// Requires even H, W (every stage-1 pixel in a window), (H+2)(W+2) <= 64*12.
// ---------------------------------------------------------------------------------------------------------------------
struct LiteFrontArgs {
    const float *dwk1, *pwk1, *pwb1, *sc1, *sh1;     // [9], [16], [16], [16], [16]
    const float *dwk2, *pwk2, *pwb2, *sc2, *sh2;     // [9][16], [16][32], [32], [32], [32]
};

__host__ __device__ inline int lite_front_floats(int H, int W)
{
    const int H1 = H / 2, W1 = W / 2, n2 = (H1 / 2) * (W1 / 2), p2 = 16 * ((n2 + 3) / 4);
    const int a = (H + 2) * (W + 2) + H * W, b = p2 * 17;          // xs + d1 share their space with d2 (dead by then)
    return ((H1 + 2) * (W1 + 2) * 16 + (a > b ? a : b) + 3) & ~3;
}

// a2h != nullptr: the pooled activation is written as fp16 there instead (fp16 inference, kws_lite_f16.h)
__global__ __launch_bounds__(64) void lite_front_infer_kernel(const float *__restrict__ feat, LiteFrontArgs k, float *__restrict__ a2,
                                                               int B, int H, int W, int clips_per_wave, _Float16 *__restrict__ a2h = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) float lsm[];
    const int lane = threadIdx.x, li = lane & 15, lq = lane >> 4;
    const int WP = W + 2, H1 = H / 2, W1 = W / 2, W1P = W1 + 2, H2 = H1 / 2, W2 = W1 / 2;
    const int nxs = (H + 2) * WP, HW = H * W, n1 = H1 * W1, n2 = H2 * W2, nt2 = (n2 + 3) / 4;
    float *a1 = lsm;                                   // [(H1+2)(W1+2)][16]
    float *xs = lsm + (H1 + 2) * W1P * 16;             // [(H+2)(W+2)]
    float *d1 = xs + nxs;                              // [H*W]
    float *d2 = xs;                                    // [16*nt2][17]   (xs and d1 are dead when it is written)

    // per-lane constants
    float dw1[9], dw2[9], pb[8], pw1c, b1c, sc1c, sh1c;
#pragma unroll
    for (int t = 0; t < 9; ++t) { dw1[t] = k.dwk1[t]; dw2[t] = k.dwk2[t * 16 + li]; }
    pw1c = k.pwk1[li]; b1c = k.pwb1[li]; sc1c = k.sc1[li]; sh1c = k.sh1[li];
#pragma unroll
    for (int j = 0; j < 4; ++j) { pb[2 * j] = k.pwk2[(4 * j + lq) * 32 + li]; pb[2 * j + 1] = k.pwk2[(4 * j + lq) * 32 + 16 + li]; }
    const float b2a = k.pwb2[li], b2b = k.pwb2[16 + li], sc2a = k.sc2[li], sc2b = k.sc2[16 + li], sh2a = k.sh2[li], sh2b = k.sh2[16 + li];
    int soff[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const int i = lane + 64 * j, r = i / WP - 1, c = i % WP - 1;
        soff[j] = (i < nxs && r >= 0 && r < H && c >= 0 && c < W) ? r * W + c : -1;
    }
    for (int i = lane; i < (H1 + 2) * W1P * 16; i += 64) a1[i] = 0.f;      // the halo stays zero for every clip

    const long first = (long)blockIdx.x * clips_per_wave;
    const long left = (long)B - first;
    const int count = left <= 0 ? 0 : (left < clips_per_wave ? (int)left : clips_per_wave);
    float pre[12];
    auto fetch = [&](long b) {
#pragma unroll
        for (int j = 0; j < 12; ++j) pre[j] = soff[j] >= 0 ? feat[b * HW + soff[j]] : 0.f;
    };
    auto wsync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    if (count > 0) fetch(first);
    for (int ci = 0; ci < count; ++ci) {
        wsync();                                       // the previous clip's reads of d2 (= xs) are done
#pragma unroll
        for (int j = 0; j < 12; ++j) { const int i = lane + 64 * j; if (i < nxs) xs[i] = pre[j]; }
        wsync();
        if (ci + 1 < count) fetch(first + ci + 1);     // the next clip's features fly during this clip
        // stage 1a: depthwise 1, lane = pixel
        for (int p = lane, y = lane / W, x = lane - (lane / W) * W; p < HW; p += 64, x += 64 % W, y += 64 / W) {
            if (x >= W) { x -= W; ++y; }                // coordinates advance incrementally: no division in the loops
            const float *b = xs + y * WP + x;
            float o = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) o = fmaf(b[(t / 3) * WP + t % 3], dw1[t], o);
            d1[p] = o;
        }
        wsync();
        // stage 1b: pointwise 1 + bias, BN, ReLU6, 2x2 max; lane = (window lq + 4 i, channel li)
        for (int w = lq, ph = lq / W1, pw = lq - (lq / W1) * W1; w < n1; w += 4, pw += 4) {
            while (pw >= W1) { pw -= W1; ++ph; }
            const float *b = d1 + (2 * ph) * W + 2 * pw;
            const float y0 = fmaf(fmaf(b[0], pw1c, b1c), sc1c, sh1c), y1 = fmaf(fmaf(b[1], pw1c, b1c), sc1c, sh1c);
            const float y2 = fmaf(fmaf(b[W], pw1c, b1c), sc1c, sh1c), y3 = fmaf(fmaf(b[W + 1], pw1c, b1c), sc1c, sh1c);
            a1[((ph + 1) * W1P + pw + 1) * 16 + li] = relu6f(fmaxf(fmaxf(y0, y1), fmaxf(y2, y3)));
        }
        wsync();
        // stage 2a: depthwise 2 of the pixels stage 2's pool windows use, stored window-major; lane = (pixel lq + 4 i, channel li)
        for (int p = lq, w = 0, ph = 0, pw = 0; p < 16 * nt2; p += 4, ++w, ++pw) {     // p = 4 w + lq: window w, element lq
            const int e = lq;
            if (pw == W2) { pw = 0; ++ph; }
            float o = 0.f;
            if (w < n2) {
                const int y = 2 * ph + (e >> 1), x = 2 * pw + (e & 1);
                const float *b = a1 + (y * W1P + x) * 16 + li;
#pragma unroll
                for (int t = 0; t < 9; ++t) o = fmaf(b[((t / 3) * W1P + t % 3) * 16], dw2[t], o);
            }
            d2[p * 17 + li] = o;
        }
        wsync();
        // stage 2b: pointwise 2 on the MFMA, then bias, BN, ReLU6, pool in registers
        float *out = a2 + (first + ci) * n2 * 32;
        for (int t = 0; t < nt2; ++t) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = d2[(16 * t + li) * 17 + 4 * j + lq];
                acc0 = mfma16(a, pb[2 * j], acc0);
                acc1 = mfma16(a, pb[2 * j + 1], acc1);
            }
            const int w = 4 * t + lq;
            if (w < n2) {
                float m0 = -INFINITY, m1 = -INFINITY;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    m0 = fmaxf(m0, fmaf(acc0[r] + b2a, sc2a, sh2a));
                    m1 = fmaxf(m1, fmaf(acc1[r] + b2b, sc2b, sh2b));
                }
                if (a2h) {
                    _Float16 *outh = a2h + (first + ci) * n2 * 32;
                    outh[w * 32 + li] = (_Float16)relu6f(m0);
                    outh[w * 32 + 16 + li] = (_Float16)relu6f(m1);
                } else {
                    out[w * 32 + li] = relu6f(m0);
                    out[w * 32 + 16 + li] = relu6f(m1);
                }
            }
        }
    }
}

}  // namespace kws
