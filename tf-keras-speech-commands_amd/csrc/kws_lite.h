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

}  // namespace kws
