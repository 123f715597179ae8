// csrc/kws_layers.h -- the non-GEMM kernels of the train / inference step (gfx950).
//   BatchNormalization statistics / apply / backward, ReLU6 + 2x2 max-pool (layer 1 has fused variants in kws_layer1.h),
//   inverted dropout, the softmax head with the reference's two losses, and the Keras-form Adam update.
// Reference semantics: classifier/models/cnn.py:27-66, classifier/loss.py:21-77, common/model_utils.py:47 and the
// Keras layer defaults listed in SURVEY.md section 7.
#pragma once
#include "kws_device.h"

namespace kws {

constexpr float kBnEps = 1e-3f;        // BatchNormalization(epsilon=1e-3)
constexpr double kBnMomentum = 0.99;   // BatchNormalization(momentum=0.99)
constexpr float kCeEps = 1e-7f;        // keras.backend.epsilon()
constexpr int kStatStride = 1024;      // max blocks of a channel reduction; partial[(which*C + c)*kStatStride + blk]

// sum of partial[(which*C + c)*kStatStride + 0..nblk) by one 256-thread block, result in every thread
__device__ __forceinline__ double block_sum_partials(const double *__restrict__ partial, int which, int C, int c, int nblk,
                                                     double *sh /* [256] */)
{
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) s += partial[((long)which * C + c) * kStatStride + b];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    s = sh[0];
    __syncthreads();
    return s;
}

// the same sum by ONE wave (64-thread block), no LDS and no barriers: the finalize kernels sit between every producer and
// consumer of the main chain, often while persistent LDS-filling kernels of the side stream own the CUs -- a block that
// needs no LDS is placed at once (the 256-thread LDS form waited up to 28 us for a slot) and a wave reduces faster than
// eight block barriers.  Four independent loads per lane and trip; fixed order, so the result is deterministic.
__device__ __forceinline__ double wave_sum_partials(const double *__restrict__ partial, int which, int C, int c, int nblk)
{
    const double *p = partial + ((long)which * C + c) * kStatStride;
    const int lane = threadIdx.x & 63;
    // all kStatStride / 64 = 16 loads of a lane are issued before the first add: one memory round trip instead of four
    // dependent ones (these kernels are nothing but latency)
    double v[kStatStride / 64];
#pragma unroll
    for (int j = 0; j < kStatStride / 64; ++j) v[j] = lane + 64 * j < nblk ? p[lane + 64 * j] : 0.0;
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < kStatStride / 64; ++j) s += v[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    return s;
}

// ---- per-channel sums over the rows of an (M x C) matrix, in double -------------------------------------------
// partial[(0*C + c)*kStatStride + blk] = sum, [(1*C + c)*kStatStride + blk] = sum of squares.  C divides 256.
__global__ __launch_bounds__(256) void channel_stats_kernel(const float *__restrict__ z, long M, int C, int rows_per_block,
                                                             double *__restrict__ partial)
{
    const int c = threadIdx.x % C, r = threadIdx.x / C, R = 256 / C;
    const long beg = (long)blockIdx.x * rows_per_block;
    const long end = beg + rows_per_block < M ? beg + rows_per_block : M;
    double s = 0.0, ss = 0.0;
    for (long m = beg + r; m < end; m += 4 * R) {       // four rows per trip: their loads are issued together
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const long mm = m + (long)u * R; v[u] = z[(mm < end ? mm : end - 1) * C + c]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double x = m + (long)u * R < end ? (double)v[u] : 0.0;
            s += x;
            ss += x * x;
        }
    }
    __shared__ double sh[2][256];
    sh[0][threadIdx.x] = s;
    sh[1][threadIdx.x] = ss;
    __syncthreads();
    if (r == 0) {
        for (int j = 1; j < R; ++j) { s += sh[0][j * C + c]; ss += sh[1][j * C + c]; }
        partial[((long)0 * C + c) * kStatStride + blockIdx.x] = s;
        partial[((long)1 * C + c) * kStatStride + blockIdx.x] = ss;
    }
}

struct BnCoef {      // per-layer float arrays of C entries each, contiguous: scale, shift, mean, inv, k2, k3
    float *scale, *shift, *mean, *inv, *k2, *k3;
};

// training: batch statistics (biased variance normalises; the moving variance takes the unbiased estimate)
__global__ void bn_finalize_train_kernel(const double *__restrict__ partial, int nblk, long M, int C,
                                         const float *__restrict__ gamma, const float *__restrict__ beta,
                                         float *__restrict__ moving_mean, float *__restrict__ moving_var, BnCoef k)
{
    const int c = blockIdx.x;            // one wave per channel
    const double s = wave_sum_partials(partial, 0, C, c, nblk);
    const double ss = wave_sum_partials(partial, 1, C, c, nblk);
    if (threadIdx.x != 0) return;
    const double mean = s / (double)M;
    double var = ss / (double)M - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const double inv = 1.0 / sqrt(var + (double)kBnEps);
    const double sc = (double)gamma[c] * inv;
    k.scale[c] = (float)sc;
    k.shift[c] = (float)((double)beta[c] - mean * sc);
    k.mean[c] = (float)mean;
    k.inv[c] = (float)inv;
    const double unbiased = var * ((double)M / (double)(M > 1 ? M - 1 : 1));
    moving_mean[c] = (float)((double)moving_mean[c] * kBnMomentum + mean * (1.0 - kBnMomentum));
    moving_var[c] = (float)((double)moving_var[c] * kBnMomentum + unbiased * (1.0 - kBnMomentum));
}

// The accumulator form of the training statistics (kws_device.h: acc_add): the producer's blocks added sum and sum of squares per channel
// to `acc`; every consumer block derives scale / shift itself (threads < C, into `sc` / `sh`: LDS), with bn_finalize_train_kernel's
// arithmetic; block 0 also writes the coefficient arrays the backward pass reads and the moving statistics, and clears the set of the
// other parity.  Contains a barrier.
struct BnAccFwd { const double *acc; double *acc_clear_set; long M; const float *gamma, *beta; float *moving_mean, *moving_var; BnCoef k; };
__device__ __forceinline__ void bn_fwd_coef_prologue(const BnAccFwd &a, int C, float *sc, float *sh, float *mean_out = nullptr, float *inv_out = nullptr)
{
    const int c = threadIdx.x;
    if (c < C) {
        const double s = acc_sum(a.acc, 2 * C, c), ss = acc_sum(a.acc, 2 * C, C + c);
        const double mean = s / (double)a.M;
        double var = ss / (double)a.M - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        const double inv = 1.0 / sqrt(var + (double)kBnEps);
        const double scd = (double)a.gamma[c] * inv;
        const float scf = (float)scd, shf = (float)((double)a.beta[c] - mean * scd);
        sc[c] = scf;
        sh[c] = shf;
        if (mean_out) { mean_out[c] = (float)mean; inv_out[c] = (float)inv; }      // a kernel that also runs the layer's backward reduction
        if (blockIdx.x == 0) {
            a.k.scale[c] = scf;
            a.k.shift[c] = shf;
            a.k.mean[c] = (float)mean;
            a.k.inv[c] = (float)inv;
            const double unbiased = var * ((double)a.M / (double)(a.M > 1 ? a.M - 1 : 1));
            a.moving_mean[c] = (float)((double)a.moving_mean[c] * kBnMomentum + mean * (1.0 - kBnMomentum));
            a.moving_var[c] = (float)((double)a.moving_var[c] * kBnMomentum + unbiased * (1.0 - kBnMomentum));
        }
    }
    if (blockIdx.x == 0 && a.acc_clear_set) acc_clear(a.acc_clear_set, threadIdx.x, blockDim.x);
    __syncthreads();
}

// inference: moving statistics
__global__ void bn_infer_coef_kernel(int C, const float *__restrict__ gamma, const float *__restrict__ beta,
                                     const float *__restrict__ moving_mean, const float *__restrict__ moving_var, BnCoef k)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double inv = 1.0 / sqrt((double)moving_var[c] + (double)kBnEps);
    const double sc = (double)gamma[c] * inv;
    k.scale[c] = (float)sc;
    k.shift[c] = (float)((double)beta[c] - (double)moving_mean[c] * sc);
    k.mean[c] = moving_mean[c];
    k.inv[c] = (float)inv;
}

// the four BatchNorm layers of a CNN forward in one launch (block = layer): inference coefficients are tiny, the launches
// are what costs
struct BnInferAll { int C[4]; const float *gamma[4], *beta[4], *mm[4], *mv[4]; BnCoef k[4]; };
__global__ void bn_infer_coef_all_kernel(BnInferAll a)
{
    const int l = blockIdx.x, c = threadIdx.x;
    if (c >= a.C[l]) return;
    const double inv = 1.0 / sqrt((double)a.mv[l][c] + (double)kBnEps);
    const double sc = (double)a.gamma[l][c] * inv;
    a.k[l].scale[c] = (float)sc;
    a.k[l].shift[c] = (float)((double)a.beta[l][c] - (double)a.mm[l][c] * sc);
    a.k[l].mean[c] = a.mm[l][c];
    a.k[l].inv[c] = (float)inv;
}

// y = relu6(z*scale + shift), optional 2x2/2 'valid' max-pool, optional inverted dropout on the result
// zmax_out / arg_out (POOL, training): the pre-BatchNorm value z of the window's routed element (the FIRST maximum of relu6(y), the rule
// of the backward pass) and its index 0..3 -- what bn_bwd_reduce_routed_kernel needs instead of a second pass over the 4x larger z.
template <bool POOL>
__global__ __launch_bounds__(256) void bn_act_pool_kernel(const float *__restrict__ z, const float *__restrict__ scale,
                                                           const float *__restrict__ shift, float *__restrict__ a, int B,
                                                           int H, int W, int C, float drop_rate, uint32_t seed_lo,
                                                           uint32_t seed_hi, float *__restrict__ zmax_out = nullptr,
                                                           unsigned char *__restrict__ arg_out = nullptr)
{
    const int Hp = POOL ? H / 2 : H, Wp = POOL ? W / 2 : W;
    const long total = (long)B * Hp * Wp * C, idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    long t = idx / C;
    const int pw = (int)(t % Wp);
    t /= Wp;
    const int ph = (int)(t % Hp), b = (int)(t / Hp);
    const float sc = scale[c], sh = shift[c];
    float v;
    if (POOL) {
        const float *p = z + (((long)b * H + 2 * ph) * W + 2 * pw) * C + c;
        const float z0 = p[0], z1 = p[C], z2 = p[(long)W * C], z3 = p[(long)W * C + C];
        const float y0 = fmaf(z0, sc, sh), y1 = fmaf(z1, sc, sh), y2 = fmaf(z2, sc, sh), y3 = fmaf(z3, sc, sh);
        v = fmaxf(fmaxf(y0, y1), fmaxf(y2, y3));
        if (zmax_out) {                  // first maximum of relu6(y): the element the backward pass routes the gradient to
            int arg = 0;
            float best = relu6f(y0), zm = z0;
            const float v1 = relu6f(y1), v2 = relu6f(y2), v3 = relu6f(y3);
            if (v1 > best) { best = v1; arg = 1; zm = z1; }
            if (v2 > best) { best = v2; arg = 2; zm = z2; }
            if (v3 > best) { arg = 3; zm = z3; }
            zmax_out[idx] = zm;
            arg_out[idx] = (unsigned char)arg;
        }
    } else {
        v = fmaf(z[idx], sc, sh);
    }
    v = relu6f(v);
    if (drop_rate > 0.f) v = dropout_keep(seed_lo, seed_hi, (uint32_t)idx, drop_rate) ? v / (1.f - drop_rate) : 0.f;
    a[idx] = v;
}

// bn_act_pool_kernel<true> in the accumulator form: every block derives the layer's scale / shift itself (C <= 128), grid-stride
__global__ __launch_bounds__(256) void bn_act_pool_acc_kernel(const float *__restrict__ z, BnAccFwd in, float *__restrict__ a, int B, int H, int W,
                                                               int C, float drop_rate, uint32_t seed_lo, uint32_t seed_hi,
                                                               float *__restrict__ zmax_out, unsigned char *__restrict__ arg_out)
{
    __shared__ float cf[256];
    bn_fwd_coef_prologue(in, C, cf, cf + 128);
    const int Hp = H / 2, Wp = W / 2;
    const long total = (long)B * Hp * Wp * C;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % C);
        long t = idx / C;
        const int pw = (int)(t % Wp);
        t /= Wp;
        const int ph = (int)(t % Hp), b = (int)(t / Hp);
        const float sc = cf[c], sh = cf[128 + c];
        const float *p = z + (((long)b * H + 2 * ph) * W + 2 * pw) * C + c;
        const float z0 = p[0], z1 = p[C], z2 = p[(long)W * C], z3 = p[(long)W * C + C];
        const float y0 = fmaf(z0, sc, sh), y1 = fmaf(z1, sc, sh), y2 = fmaf(z2, sc, sh), y3 = fmaf(z3, sc, sh);
        float v = fmaxf(fmaxf(y0, y1), fmaxf(y2, y3));
        int arg = 0;                         // first maximum of relu6(y): the element the backward pass routes the gradient to
        float best = relu6f(y0), zm = z0;
        const float v1 = relu6f(y1), v2 = relu6f(y2), v3 = relu6f(y3);
        if (v1 > best) { best = v1; arg = 1; zm = z1; }
        if (v2 > best) { best = v2; arg = 2; zm = z2; }
        if (v3 > best) { arg = 3; zm = z3; }
        zmax_out[idx] = zm;
        arg_out[idx] = (unsigned char)arg;
        v = relu6f(v);
        if (drop_rate > 0.f) v = dropout_keep(seed_lo, seed_hi, (uint32_t)idx, drop_rate) ? v / (1.f - drop_rate) : 0.f;
        a[idx] = v;
    }
}

// backward through dropout / max-pool / ReLU6 to the BN output: writes g = dL/dy per z element and the per-channel
// partial sums of g and g*xhat (double).  The pool routes the gradient to the FIRST maximum of its window.
template <bool POOL>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float *__restrict__ z, const float *__restrict__ da,
                                                             BnCoef k, float *__restrict__ gz, int B, int H, int W, int C,
                                                             int rows_per_block, double *__restrict__ partial,
                                                             float drop_rate, uint32_t seed_lo, uint32_t seed_hi)
{
    const int c = threadIdx.x % C, r = threadIdx.x / C, R = 256 / C;
    const int Hp = POOL ? H / 2 : H, Wp = POOL ? W / 2 : W;
    const long M = (long)B * H * W, beg = (long)blockIdx.x * rows_per_block;
    const long end = beg + rows_per_block < M ? beg + rows_per_block : M;
    const float sc = k.scale[c], sh = k.shift[c], mean = k.mean[c], inv = k.inv[c];
    double s = 0.0, sx = 0.0;
    for (long m0 = beg + r; m0 < end; m0 += 4 * R) {    // four rows per trip: their loads are issued together
      float zv4[4], dv4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {                     // unconditional loads on clamped rows: nothing to branch around
          const long mm = m0 + (long)u * R, mc = mm < end ? mm : end - 1;
          zv4[u] = z[mc * C + c];
          dv4[u] = POOL ? 0.f : da[mc * C + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long m = m0 + (long)u * R;
        if (m >= end) continue;
        const int pix = (int)(m % ((long)H * W)), b = (int)(m / ((long)H * W)), ih = pix / W, iw = pix % W;
        const float zv = zv4[u];
        const float y = fmaf(zv, sc, sh);
        float g = 0.f;
        if (POOL) {
            const int ph = ih >> 1, pw = iw >> 1;
            if (ph < Hp && pw < Wp) {
                const float *p = z + (((long)b * H + 2 * ph) * W + 2 * pw) * C + c;
                const float v0 = relu6f(fmaf(p[0], sc, sh)), v1 = relu6f(fmaf(p[C], sc, sh));
                const float v2 = relu6f(fmaf(p[(long)W * C], sc, sh)), v3 = relu6f(fmaf(p[(long)W * C + C], sc, sh));
                int arg = 0;
                float best = v0;
                if (v1 > best) { best = v1; arg = 1; }
                if (v2 > best) { best = v2; arg = 2; }
                if (v3 > best) { best = v3; arg = 3; }
                if (arg == (ih & 1) * 2 + (iw & 1)) {
                    const long oidx = (((long)b * Hp + ph) * Wp + pw) * C + c;
                    g = da[oidx];
                    if (drop_rate > 0.f) g = dropout_keep(seed_lo, seed_hi, (uint32_t)oidx, drop_rate) ? g / (1.f - drop_rate) : 0.f;
                }
            }
        } else {
            g = dv4[u];
            if (drop_rate > 0.f) g = dropout_keep(seed_lo, seed_hi, (uint32_t)(m * C + c), drop_rate) ? g / (1.f - drop_rate) : 0.f;
        }
        g = (y > 0.f && y < 6.f) ? g : 0.f;
        gz[m * C + c] = g;
        s += (double)g;
        sx += (double)g * (double)((zv - mean) * inv);
      }
    }
    __shared__ double shm[2][256];
    shm[0][threadIdx.x] = s;
    shm[1][threadIdx.x] = sx;
    __syncthreads();
    if (r == 0) {
        for (int j = 1; j < R; ++j) { s += shm[0][j * C + c]; sx += shm[1][j * C + c]; }
        partial[((long)0 * C + c) * kStatStride + blockIdx.x] = s;
        partial[((long)1 * C + c) * kStatStride + blockIdx.x] = sx;
    }
}

// Pooled layers, one thread per (pool window, channel): the four elements of the window are read once, the gradient goes
// to the first arg-max (if its ReLU6 is live), the other three and the pixels outside every window (odd H or W) get g = 0.
// Same outputs as bn_bwd_reduce_kernel<true> (gz and the double partial sums) at a quarter of the loads.
// COMPACT: g is non-zero at ONE element per window, so instead of a z-sized gz the kernel leaves the routed, gated value
// per (window, channel) in place of da and the element index (0..3) as a byte in `arg`; the consumer (conv2's clip data
// gradient) rebuilds g while it stages.  Pixels outside every window have g = 0 by construction.
template <bool COMPACT = false>
__global__ __launch_bounds__(256) void bn_bwd_reduce_pool_kernel(const float *__restrict__ z, float *da, BnCoef k,
                                                                  float *__restrict__ gz, int B, int H, int W, int C,
                                                                  int wins_per_block, double *__restrict__ partial,
                                                                  float drop_rate, uint32_t seed_lo, uint32_t seed_hi,
                                                                  unsigned char *__restrict__ arg_out = nullptr)
{
    const int c = threadIdx.x % C, r = threadIdx.x / C, R = 256 / C;
    const int Hp = H / 2, Wp = W / 2;
    const long NW = (long)B * Hp * Wp, beg = (long)blockIdx.x * wins_per_block;
    const long end = beg + wins_per_block < NW ? beg + wins_per_block : NW;
    const float sc = k.scale[c], sh = k.shift[c], mean = k.mean[c], inv = k.inv[c];
    double s = 0.0, sx = 0.0;
    for (long q = beg + r; q < end; q += R) {
        const int win = (int)(q % ((long)Hp * Wp)), b = (int)(q / ((long)Hp * Wp)), ph = win / Wp, pw = win % Wp;
        const long o00 = (((long)b * H + 2 * ph) * W + 2 * pw) * C + c;
        const long off[4] = {o00, o00 + C, o00 + (long)W * C, o00 + (long)W * C + C};
        float zv[4], y[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { zv[j] = z[off[j]]; y[j] = fmaf(zv[j], sc, sh); }
        int arg = 0;
        float best = relu6f(y[0]);
#pragma unroll
        for (int j = 1; j < 4; ++j) { const float v = relu6f(y[j]); if (v > best) { best = v; arg = j; } }
        float g = da[q * C + c];
        if (drop_rate > 0.f) g = dropout_keep(seed_lo, seed_hi, (uint32_t)(q * C + c), drop_rate) ? g / (1.f - drop_rate) : 0.f;
        const float ya = arg == 0 ? y[0] : arg == 1 ? y[1] : arg == 2 ? y[2] : y[3];
        const float za = arg == 0 ? zv[0] : arg == 1 ? zv[1] : arg == 2 ? zv[2] : zv[3];
        g = (ya > 0.f && ya < 6.f) ? g : 0.f;
        if (COMPACT) {
            da[q * C + c] = g;
            arg_out[q * C + c] = (unsigned char)arg;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) gz[off[j]] = j == arg ? g : 0.f;
        }
        s += (double)g;
        sx += (double)g * (double)((za - mean) * inv);
    }
    // pixels outside every window: g = 0 (only when H or W is odd)
    const int nbh = H - 2 * Hp, nbw = W - 2 * Wp;
    if (!COMPACT && (nbh || nbw)) {
        const long per = (long)nbh * W + (long)nbw * 2 * Hp, NB = (long)B * per;
        for (long q = (long)blockIdx.x * (256 / C) + r; q < NB; q += (long)gridDim.x * (256 / C)) {
            const int b = (int)(q / per), e = (int)(q % per);
            int ih, iw;
            if (e < nbh * W) { ih = 2 * Hp + e / W; iw = e % W; }
            else { const int e2 = e - nbh * W; ih = e2 / nbw; iw = 2 * Wp + e2 % nbw; }
            gz[(((long)b * H + ih) * W + iw) * C + c] = 0.f;
        }
    }
    __shared__ double shm[2][256];
    shm[0][threadIdx.x] = s;
    shm[1][threadIdx.x] = sx;
    __syncthreads();
    if (r == 0) {
        for (int j = 1; j < R; ++j) { s += shm[0][j * C + c]; sx += shm[1][j * C + c]; }
        partial[((long)0 * C + c) * kStatStride + blockIdx.x] = s;
        partial[((long)1 * C + c) * kStatStride + blockIdx.x] = sx;
    }
}

// The pooled backward reduction from what the forward pass left per (pool window, channel): zmax (z of the routed element) and arg (its
// index, consumed by the data / weight gradient kernels).  Same arithmetic as bn_bwd_reduce_pool_kernel<true> on that element -- the gate
// on y = z scale + shift, g, sum g, sum g xhat -- without reading the four z values of every window again (conv2: 38 instead of 114 MB).
// da is overwritten with the routed, gated gradient (the compact form).  One thread per (window, channel), grid-stride over windows.
__global__ __launch_bounds__(256) void bn_bwd_reduce_routed_kernel(const float *__restrict__ zmax, float *da, BnCoef k, long NW, int C,
                                                                    int wins_per_block, double *__restrict__ partial)
{
    const int c = threadIdx.x % C, r = threadIdx.x / C, R = 256 / C;
    const long beg = (long)blockIdx.x * wins_per_block;
    const long end = beg + wins_per_block < NW ? beg + wins_per_block : NW;
    const float sc = k.scale[c], sh = k.shift[c], mean = k.mean[c], inv = k.inv[c];
    double s = 0.0, sx = 0.0;
    for (long q0 = beg + r; q0 < end; q0 += 4 * R) {       // four windows per trip: their loads are issued together
        float za[4], g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long q = q0 + (long)u * R, qc = q < end ? q : end - 1;
            za[u] = zmax[qc * C + c];
            g[u] = da[qc * C + c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long q = q0 + (long)u * R;
            if (q >= end) continue;
            const float ya = fmaf(za[u], sc, sh);
            const float gv = (ya > 0.f && ya < 6.f) ? g[u] : 0.f;
            da[q * C + c] = gv;
            s += (double)gv;
            sx += (double)gv * (double)((za[u] - mean) * inv);
        }
    }
    __shared__ double shm[2][256];
    shm[0][threadIdx.x] = s;
    shm[1][threadIdx.x] = sx;
    __syncthreads();
    if (r == 0) {
        for (int j = 1; j < R; ++j) { s += shm[0][j * C + c]; sx += shm[1][j * C + c]; }
        partial[((long)0 * C + c) * kStatStride + blockIdx.x] = s;
        partial[((long)1 * C + c) * kStatStride + blockIdx.x] = sx;
    }
}

// The same for a consumer that wants the full-size g (layer 4: bn_bwd_apply_planes reads gz): zmax / arg from the forward pass replace
// the four z loads of every window; dropout, gate, sums and the zero fill of the pixels outside every window as in
// bn_bwd_reduce_pool_kernel<false>.
__global__ __launch_bounds__(256) void bn_bwd_reduce_routed_full_kernel(const float *__restrict__ zmax, const unsigned char *__restrict__ argin,
                                                                         const float *__restrict__ da, BnCoef k, float *__restrict__ gz, int B,
                                                                         int H, int W, int C, int wins_per_block, double *__restrict__ partial,
                                                                         float drop_rate, uint32_t seed_lo, uint32_t seed_hi)
{
    const int c = threadIdx.x % C, r = threadIdx.x / C, R = 256 / C;
    const int Hp = H / 2, Wp = W / 2;
    const long NW = (long)B * Hp * Wp, beg = (long)blockIdx.x * wins_per_block;
    const long end = beg + wins_per_block < NW ? beg + wins_per_block : NW;
    const float sc = k.scale[c], sh = k.shift[c], mean = k.mean[c], inv = k.inv[c];
    double s = 0.0, sx = 0.0;
    for (long q = beg + r; q < end; q += R) {
        const int win = (int)(q % ((long)Hp * Wp)), b = (int)(q / ((long)Hp * Wp)), ph = win / Wp, pw = win % Wp;
        const long o00 = (((long)b * H + 2 * ph) * W + 2 * pw) * C + c;
        const long off[4] = {o00, o00 + C, o00 + (long)W * C, o00 + (long)W * C + C};
        const float za = zmax[q * C + c];
        const int arg = argin[q * C + c];
        float g = da[q * C + c];
        if (drop_rate > 0.f) g = dropout_keep(seed_lo, seed_hi, (uint32_t)(q * C + c), drop_rate) ? g / (1.f - drop_rate) : 0.f;
        const float ya = fmaf(za, sc, sh);
        g = (ya > 0.f && ya < 6.f) ? g : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) gz[off[j]] = j == arg ? g : 0.f;
        s += (double)g;
        sx += (double)g * (double)((za - mean) * inv);
    }
    const int nbh = H - 2 * Hp, nbw = W - 2 * Wp;          // pixels outside every window: g = 0 (only when H or W is odd)
    if (nbh || nbw) {
        const long per = (long)nbh * W + (long)nbw * 2 * Hp, NB = (long)B * per;
        for (long q = (long)blockIdx.x * (256 / C) + r; q < NB; q += (long)gridDim.x * (256 / C)) {
            const int b = (int)(q / per), e = (int)(q % per);
            int ih, iw;
            if (e < nbh * W) { ih = 2 * Hp + e / W; iw = e % W; }
            else { const int e2 = e - nbh * W; ih = e2 / nbw; iw = 2 * Wp + e2 % nbw; }
            gz[(((long)b * H + ih) * W + iw) * C + c] = 0.f;
        }
    }
    __shared__ double shm[2][256];
    shm[0][threadIdx.x] = s;
    shm[1][threadIdx.x] = sx;
    __syncthreads();
    if (r == 0) {
        for (int j = 1; j < R; ++j) { s += shm[0][j * C + c]; sx += shm[1][j * C + c]; }
        partial[((long)0 * C + c) * kStatStride + blockIdx.x] = s;
        partial[((long)1 * C + c) * kStatStride + blockIdx.x] = sx;
    }
}

__global__ void bn_bwd_finalize_kernel(const double *__restrict__ partial, int nblk, long M, int C,
                                       const float *__restrict__ gamma, float *__restrict__ dgamma,
                                       float *__restrict__ dbeta, BnCoef k)
{
    const int c = blockIdx.x;            // one wave per channel
    const double s = wave_sum_partials(partial, 0, C, c, nblk);
    const double sx = wave_sum_partials(partial, 1, C, c, nblk);
    if (threadIdx.x != 0) return;
    dbeta[c] = (float)s;
    dgamma[c] = (float)sx;
    k.k2[c] = (float)(s / (double)M);
    k.k3[c] = (float)(sx / (double)M);
}

// dz = gamma*inv * (g - mean(g) - xhat * mean(g*xhat)); RELU_IN additionally gates by the conv's own relu (cnn.py:55)
template <bool RELU_IN>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float *__restrict__ z, float *__restrict__ gz, BnCoef k,
                                                            const float *__restrict__ gamma, long total, int C)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const float zv = z[idx];
    const float xhat = (zv - k.mean[c]) * k.inv[c];
    float d = gamma[c] * k.inv[c] * (gz[idx] - k.k2[c] - xhat * k.k3[c]);
    if (RELU_IN) d = zv > 0.f ? d : 0.f;
    gz[idx] = d;
}

// The same for a consumer that works on the three-way bf16 split (conv4's data and weight gradients): four channels per
// thread, dz leaves as the h / m / l planes (kws_device.h: split_bf16) in the tensor's own NHWC order and the fp32 tensor
// is not written at all -- the nine taps of the data gradient and the nine tap blocks of the weight gradient then copy
// 16-byte pieces into LDS instead of each repeating the split.
struct Bf16PlanesOut { __bf16 *p[3]; };
template <bool RELU_IN>
__global__ __launch_bounds__(256) void bn_bwd_apply_planes_kernel(const float *__restrict__ z, const float *__restrict__ gz, BnCoef k,
                                                                   const float *__restrict__ gamma, long total4, int C, Bf16PlanesOut out)
{
    const long i4 = (long)blockIdx.x * 256 + threadIdx.x;
    if (i4 >= total4) return;
    const int c0 = (int)((i4 * 4) % C);
    const f32x4 zv = reinterpret_cast<const f32x4 *>(z)[i4], gv = reinterpret_cast<const f32x4 *>(gz)[i4];
    f32x4 d;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = c0 + e;
        const float xhat = (zv[e] - k.mean[c]) * k.inv[c];
        float v = gamma[c] * k.inv[c] * (gv[e] - k.k2[c] - xhat * k.k3[c]);
        if (RELU_IN) v = zv[e] > 0.f ? v : 0.f;
        d[e] = v;
    }
    bf16x4 h, m, l;
    split_bf16(d, h, m, l);
    reinterpret_cast<bf16x4 *>(out.p[0])[i4] = h;
    reinterpret_cast<bf16x4 *>(out.p[1])[i4] = m;
    reinterpret_cast<bf16x4 *>(out.p[2])[i4] = l;
}

// The accumulator form of the backward coefficients (kws_device.h: acc_add): the first 2 C threads of every block sum one value each,
// k2 = sum g / M and k3 = sum g xhat / M go to `kk` (LDS, [2 C]); block 0 also leaves what bn_bwd_finalize_kernel wrote and clears the
// set of the other parity.  Contains a barrier.
struct BnAccBwd { const double *acc; double *acc_clear_set; long M; float *dgamma, *dbeta; };
__device__ __forceinline__ void bn_bwd_k_prologue(const BnAccBwd &a, const BnCoef &k, int C, float *kk)
{
    const int i = threadIdx.x;
    if (i < 2 * C) {
        const double t = acc_sum(a.acc, 2 * C, i);
        const float kv = (float)(t / (double)a.M);
        kk[i] = kv;
        if (blockIdx.x == 0) {
            if (i < C) { a.dbeta[i] = (float)t; k.k2[i] = kv; }
            else { a.dgamma[i - C] = (float)t; k.k3[i - C] = kv; }
        }
    }
    if (blockIdx.x == 0 && a.acc_clear_set) acc_clear(a.acc_clear_set, threadIdx.x, blockDim.x);
    __syncthreads();
}

// Layer 4 (pooled, split precision) from the COMPACT gradient: gq (B, H/2, W/2, C) holds the dropped, gated gradient of every pool window
// (dense_head_fused_kernel's epilogue), arg the element it belongs to; every other element of z has g = 0.  Same arithmetic as
// bn_bwd_apply_planes_kernel on the expanded g, without the z-sized fp32 g tensor in between; grid-stride, 256 threads, C <= 128.
template <bool RELU_IN>
__global__ __launch_bounds__(256) void bn_bwd_apply_routed_planes_kernel(const float *__restrict__ z, const float *__restrict__ gq,
                                                                          const unsigned char *__restrict__ arg, BnCoef k,
                                                                          const float *__restrict__ gamma, int B, int H, int W, int C,
                                                                          BnAccBwd a, Bf16PlanesOut out)
{
    __shared__ float kk[256];
    bn_bwd_k_prologue(a, k, C, kk);
    const int Hp = H / 2, Wp = W / 2, C4 = C / 4;
    const long total4 = (long)B * H * W * C4;
    for (long i4 = (long)blockIdx.x * 256 + threadIdx.x; i4 < total4; i4 += (long)gridDim.x * 256) {
        const int c0 = (int)(i4 % C4) * 4;
        const long pix = i4 / C4;
        const int p = (int)(pix % ((long)H * W)), b = (int)(pix / ((long)H * W)), y = p / W, x = p - y * W, ph = y >> 1, pw = x >> 1;
        const f32x4 zv = reinterpret_cast<const f32x4 *>(z)[i4];
        f32x4 gv = {0.f, 0.f, 0.f, 0.f};
        if (ph < Hp && pw < Wp) {
            const long q = (((long)b * Hp + ph) * Wp + pw) * C + c0;
            const f32x4 g4 = *reinterpret_cast<const f32x4 *>(gq + q);
            const unsigned a4 = *reinterpret_cast<const unsigned *>(arg + q), e = (unsigned)((y & 1) * 2 + (x & 1));
#pragma unroll
            for (int j = 0; j < 4; ++j) gv[j] = ((a4 >> (8 * j)) & 0xffu) == e ? g4[j] : 0.f;
        }
        f32x4 d;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + j;
            const float xhat = (zv[j] - k.mean[c]) * k.inv[c];
            float v = gamma[c] * k.inv[c] * (gv[j] - kk[c] - xhat * kk[C + c]);
            if (RELU_IN) v = zv[j] > 0.f ? v : 0.f;
            d[j] = v;
        }
        bf16x4 h, m, l;
        split_bf16(d, h, m, l);
        reinterpret_cast<bf16x4 *>(out.p[0])[i4] = h;
        reinterpret_cast<bf16x4 *>(out.p[1])[i4] = m;
        reinterpret_cast<bf16x4 *>(out.p[2])[i4] = l;
    }
}

// The accumulator form of bn_bwd_apply_kernel (layers without pooling: g is full-size, in place): grid-stride, 256 threads, C <= 128
template <bool RELU_IN>
__global__ __launch_bounds__(256) void bn_bwd_apply_acc_kernel(const float *__restrict__ z, float *__restrict__ gz, BnCoef k,
                                                                const float *__restrict__ gamma, long total4, int C, BnAccBwd a)
{
    __shared__ float kk[256];
    bn_bwd_k_prologue(a, k, C, kk);
    const int C4 = C / 4;
    for (long i4 = (long)blockIdx.x * 256 + threadIdx.x; i4 < total4; i4 += (long)gridDim.x * 256) {
        const int c0 = (int)(i4 % C4) * 4;
        const f32x4 zv = reinterpret_cast<const f32x4 *>(z)[i4], gv = reinterpret_cast<const f32x4 *>(gz)[i4];
        f32x4 d;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + j;
            const float xhat = (zv[j] - k.mean[c]) * k.inv[c];
            float v = gamma[c] * k.inv[c] * (gv[j] - kk[c] - xhat * kk[C + c]);
            if (RELU_IN) v = zv[j] > 0.f ? v : 0.f;
            d[j] = v;
        }
        reinterpret_cast<f32x4 *>(gz)[i4] = d;
    }
}

// ---- head: Dense(C, softmax) 'score_predict' (classifier/model.py:37) + loss (classifier/loss.py) --------------
// block = 16 samples.  logits -> probs, per-sample loss / correct flag, dlogits = d(mean loss)/d(logits) * grad_scale
__global__ __launch_bounds__(256) void head_fwd_kernel(const float *__restrict__ d1, const float *__restrict__ w2,
                                                        const float *__restrict__ b2, const int32_t *__restrict__ labels,
                                                        const float *__restrict__ class_w, float *__restrict__ probs,
                                                        int32_t *__restrict__ argmax_out, float *__restrict__ loss_i,
                                                        float *__restrict__ correct_i, float *__restrict__ dlogits, int B,
                                                        int K, int C, float grad_scale, int ignore_index)
{
    extern __shared__ float hs[];
    float *xs = hs;              // [16][K]
    float *lg = hs + 16 * K;     // [16][C]
    const int b0 = blockIdx.x * 16;
    for (int i = threadIdx.x; i < 16 * K; i += 256) {
        const int s = i / K;
        xs[i] = (b0 + s < B) ? d1[(long)(b0 + s) * K + (i % K)] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 16 * C; i += 256) {
        const int s = i / C, c = i % C;
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc = fmaf(xs[s * K + k], w2[(long)k * C + c], acc);
        lg[i] = acc + b2[c];
    }
    __syncthreads();
    if (threadIdx.x < 16 && b0 + threadIdx.x < B) {
        const int s = threadIdx.x, b = b0 + s;
        float mx = lg[s * C];
        int am = 0;
        for (int c = 1; c < C; ++c)
            if (lg[s * C + c] > mx) { mx = lg[s * C + c]; am = c; }
        float sum = 0.f;
        for (int c = 0; c < C; ++c) { const float e = expf(lg[s * C + c] - mx); lg[s * C + c] = e; sum += e; }
        const float rs = 1.f / sum;
        if (argmax_out) argmax_out[b] = am;
        if (probs)
            for (int c = 0; c < C; ++c) probs[(long)b * C + c] = lg[s * C + c] * rs;
        if (labels) {
            const int y = labels[b];
            const float py = lg[s * C + y] * rs;
            float loss, coef;
            if (class_w) {                       // loss.py:67-71: -log(p_y) * w_y, no clipping
                loss = -logf(py) * class_w[y];
                coef = class_w[y];
            } else {                             // loss.py:36: K.categorical_crossentropy on probabilities (clipped)
                const float lo = kCeEps, hi = 1.f - kCeEps;
                loss = -logf(fminf(fmaxf(py, lo), hi));
                coef = (py >= lo && py <= hi) ? 1.f : 0.f;
            }
            if (ignore_index > 0 && y == ignore_index) { loss = 0.f; coef = 0.f; }   // loss.py:38-40,73-75
            loss_i[b] = loss;
            correct_i[b] = am == y ? 1.f : 0.f;
            if (dlogits)
                for (int c = 0; c < C; ++c)
                    dlogits[(long)b * C + c] = (lg[s * C + c] * rs - (c == y ? 1.f : 0.f)) * coef * grad_scale;
        }
    }
}

// Fast form for heads whose W2 (K x C) fits in LDS next to the 16-sample tile: W2 is staged once per block (the kernel above
// reads it from global memory inside the K loop), and a sample's classes are spread over 16 lanes, so the arg-max, the
// softmax sums and the writes are 16-wide instead of one thread per sample.  Same arithmetic order per logit (k ascending),
// same first-maximum rule, same loss formulas.
__global__ __launch_bounds__(256) void head_fwd_fast_kernel(const float *__restrict__ d1, const float *__restrict__ w2,
                                                             const float *__restrict__ b2, const int32_t *__restrict__ labels,
                                                             const float *__restrict__ class_w, float *__restrict__ probs,
                                                             int32_t *__restrict__ argmax_out, float *__restrict__ loss_i,
                                                             float *__restrict__ correct_i, float *__restrict__ dlogits, int B,
                                                             int K, int C, float grad_scale, int ignore_index)
{
    extern __shared__ float hs[];
    const int KS = K + 1;        // padded row: the 16 samples of a column read different banks
    float *xs = hs;              // [16][KS]
    float *ws = hs + 16 * KS;    // [K][C]
    float *lg = ws + K * C;      // [16][C]
    const int b0 = blockIdx.x * 16;
    for (int i = threadIdx.x; i < 16 * K; i += 256) {
        const int s = i / K, k = i - s * K;
        xs[s * KS + k] = (b0 + s < B) ? d1[(long)(b0 + s) * K + k] : 0.f;
    }
    for (int i = threadIdx.x; i < K * C; i += 256) ws[i] = w2[i];
    __syncthreads();
    const int s = threadIdx.x >> 4, j = threadIdx.x & 15, b = b0 + s;
    // logits of classes j, j + 16, ...
    for (int c = j; c < C; c += 16) {
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc = fmaf(xs[s * KS + k], ws[k * C + c], acc);
        lg[s * C + c] = acc + b2[c];
    }
    __syncthreads();
    // first maximum over the sample's classes: per lane over its classes (ascending), then across the 16 lanes (lower class wins ties)
    float mx = -INFINITY;
    int am = 0x7fffffff;
    for (int c = j; c < C; c += 16) {
        const float v = lg[s * C + c];
        if (v > mx) { mx = v; am = c; }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        const float omx = __shfl_xor(mx, o, 16);
        const int oam = __shfl_xor(am, o, 16);
        if (omx > mx || (omx == mx && oam < am)) { mx = omx; am = oam; }
    }
    float sum = 0.f;
    for (int c = j; c < C; c += 16) { const float e = expf(lg[s * C + c] - mx); lg[s * C + c] = e; sum += e; }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 16);
    const float rs = 1.f / sum;
    if (b >= B) return;
    if (argmax_out && j == 0) argmax_out[b] = am;
    if (probs)
        for (int c = j; c < C; c += 16) probs[(long)b * C + c] = lg[s * C + c] * rs;
    if (labels) {
        const int y = labels[b];
        const float py = lg[s * C + y] * rs;      // written by lane y % 16 before the shuffles above (same wave: in order)
        float loss, coef;
        if (class_w) {                       // loss.py:67-71: -log(p_y) * w_y, no clipping
            loss = -logf(py) * class_w[y];
            coef = class_w[y];
        } else {                             // loss.py:36: K.categorical_crossentropy on probabilities (clipped)
            const float lo = kCeEps, hi = 1.f - kCeEps;
            loss = -logf(fminf(fmaxf(py, lo), hi));
            coef = (py >= lo && py <= hi) ? 1.f : 0.f;
        }
        if (ignore_index > 0 && y == ignore_index) { loss = 0.f; coef = 0.f; }   // loss.py:38-40,73-75
        if (j == 0) { loss_i[b] = loss; correct_i[b] = am == y ? 1.f : 0.f; }
        if (dlogits)
            for (int c = j; c < C; c += 16) dlogits[(long)b * C + c] = (lg[s * C + c] * rs - (c == y ? 1.f : 0.f)) * coef * grad_scale;
    }
}

// deterministic sum of per-sample losses / correct flags: out[0] = sum(loss), out[1] = sum(correct)
__global__ __launch_bounds__(256) void loss_reduce_kernel(const float *__restrict__ loss_i, const float *__restrict__ correct_i,
                                                           int B, float *__restrict__ out)
{
    double s = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < B; i += 256) { s += (double)loss_i[i]; c += (double)correct_i[i]; }
    __shared__ double sh[2][256];
    sh[0][threadIdx.x] = s;
    sh[1][threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { sh[0][threadIdx.x] += sh[0][threadIdx.x + o]; sh[1][threadIdx.x] += sh[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = (float)sh[0][0]; out[1] = (float)sh[1][0]; }
}

// head backward: dx[b][k] = sum_c dlogits[b][c] W2[k][c], gated by the ReLU6 of x (x = relu6 output: 0 < x < 6);
// dW2[k][c] += sum_b x[b][k] dlogits[b][c]; db2[c] += sum_b dlogits[b][c].  block = kHeadBwdRows samples.
constexpr int kHeadBwdRows = 16;

template <bool RELU6_GATE>
__global__ __launch_bounds__(256) void head_bwd_kernel(const float *__restrict__ x, const float *__restrict__ w2,
                                                        const float *__restrict__ dlogits, float *__restrict__ dx,
                                                        float *__restrict__ dw2, float *__restrict__ db2, int B, int K, int C)
{
    extern __shared__ float hs[];
    float *xs = hs;              // [R][K]
    float *ds = hs + kHeadBwdRows * K;     // [R][C]
    const int b0 = blockIdx.x * kHeadBwdRows;
    for (int i = threadIdx.x; i < kHeadBwdRows * K; i += 256) xs[i] = (b0 + i / K < B) ? x[(long)(b0 + i / K) * K + (i % K)] : 0.f;
    for (int i = threadIdx.x; i < kHeadBwdRows * C; i += 256) ds[i] = (b0 + i / C < B) ? dlogits[(long)(b0 + i / C) * C + (i % C)] : 0.f;
    __syncthreads();
    for (int i = threadIdx.x; i < kHeadBwdRows * K; i += 256) {
        const int s = i / K, k = i % K;
        if (b0 + s >= B) continue;
        float acc = 0.f;
        for (int c = 0; c < C; ++c) acc = fmaf(ds[s * C + c], w2[(long)k * C + c], acc);
        if (RELU6_GATE) { const float xv = xs[i]; acc = (xv > 0.f && xv < 6.f) ? acc : 0.f; }
        dx[(long)(b0 + s) * K + k] = acc;
    }
    for (int i = threadIdx.x; i < K * C; i += 256) {
        const int k = i / C, c = i % C;
        float acc = 0.f;
        for (int s = 0; s < kHeadBwdRows; ++s) acc = fmaf(xs[s * K + k], ds[s * C + c], acc);
        if (dw2) atomicAdd(dw2 + i, acc);
    }
    for (int c = threadIdx.x; c < C; c += 256) {
        float acc = 0.f;
        for (int s = 0; s < kHeadBwdRows; ++s) acc += ds[s * C + c];
        if (db2) atomicAdd(db2 + c, acc);
    }
}

// Deterministic form of the head's weight / bias gradient (kws_model_set_deterministic): one thread per entry of dW2 (and of
// db2) walks the batch in order, so the sums do not depend on which block's float atomic lands first.  For parity tests.
__global__ __launch_bounds__(256) void head_wgrad_det_kernel(const float *__restrict__ x, const float *__restrict__ dlogits,
                                                              float *__restrict__ dw2, float *__restrict__ db2, int B, int K, int C)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < K * C) {
        const int k = i / C, c = i - k * C;
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc = fmaf(x[(long)b * K + k], dlogits[(long)b * C + c], acc);
        dw2[i] += acc;
    } else if (i < K * C + C) {
        const int c = i - K * C;
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc += dlogits[(long)b * C + c];
        db2[c] += acc;
    }
}

// Fast form of the head backward for K a multiple of 16 and C <= 48 (W2 and the tiles fit in LDS): the two small products
// run on the fp32 MFMA.  Block = 16 samples, 4 waves.
//   dW2[k][c] += sum_r x[r][k] dl[r][c]      M = K (K/16 tiles), N = 48 (3 tiles), reduction over the 16 samples (4 k-steps)
//   dx[r][k]   = sum_c dl[r][c] W2[k][c]      M = 16 samples, N = K (K/16 tiles), reduction over 48 padded classes (12 k-steps)
// dl and W2 are zero-padded to 48 columns in LDS, so no lane needs a mask; dW2 / db2 are added with float atomics as before.
// FWD (train step, GROUPS == 1): the kernel is the head's FORWARD as well -- logits from the staged tile and W2 (the k-ascending fmaf
// chain of head_fwd_fast_kernel), first-maximum arg-max, softmax, the per-sample loss of loss.py and dlogits, which go straight into
// the LDS tile the two products read.  One launch and the (B, C) dlogits round trip less on the step's critical chain (11 us of
// head_fwd + launch gap at B = 4096).  The per-sample losses / correct flags go to loss_i_out / correct_i_out and are summed by
// loss_reduce_kernel (on the side stream), `dlogits`, `loss_i`, `correct_i` and `stats` are then unused.
template <bool RELU6_GATE, int GROUPS, bool FWD = false>
__global__ __launch_bounds__(256) void head_bwd_mfma_kernel(const float *__restrict__ x, const float *__restrict__ w2,
                                                             const float *__restrict__ dlogits, float *__restrict__ dx,
                                                             float *__restrict__ dw2, float *__restrict__ db2, int B, int K, int C,
                                                             float *__restrict__ dx_colsum, const float *__restrict__ loss_i,
                                                             const float *__restrict__ correct_i, float *__restrict__ stats,
                                                             HeadFwdArgs fw = HeadFwdArgs{})
{
    static_assert(!FWD || GROUPS == 1, "the fused forward handles one 16-sample group per block");
    // dx_colsum (nullable): += column sums of dx, i.e. the bias gradient of the layer that produced x (float atomics).
    // stats (nullable): block 0 also writes {sum of loss_i, sum of correct_i} in a fixed order (double), which saves the
    // separate loss_reduce launch of a train step.
    // GROUPS 16-sample groups per block: dW2 / db2 accumulate in registers across them, so the float atomics (one per
    // entry and block -- the kernel's bound) shrink by that factor.  K <= 128 (at most 6 dW2 tiles per wave).
    constexpr int CP = 48, CS = 50;                 // padded classes; LDS row stride (== 18 mod 32: conflict-free column reads)
    constexpr int MAXT = 6;
    extern __shared__ float hs[];
    const int KS = K + 2;                           // x tile row stride
    float *xs = hs;                                 // [16][KS]
    float *ds = xs + 16 * KS;                       // [16][CS], columns >= C are zero
    float *ws = ds + 16 * CS;                       // [K][CS],  columns >= C are zero
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
    const int KT = K / 16;
    {
        // all of a thread's W2 loads first, then the LDS stores: one L2 round trip instead of up to 24 dependent ones (K <= 128)
        constexpr int NWV = 128 * CP / 256;
        float wv[NWV];
#pragma unroll
        for (int j = 0; j < NWV; ++j) {
            const int i = threadIdx.x + 256 * j, k = i / CP, c = i - k * CP;
            wv[j] = (i < K * CP && c < C) ? w2[(long)k * C + c] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NWV; ++j) {
            const int i = threadIdx.x + 256 * j, k = i / CP, c = i - k * CP;
            if (i < K * CP) ws[k * CS + c] = wv[j];
        }
    }
    f32x4 accw[MAXT];
#pragma unroll
    for (int q = 0; q < MAXT; ++q) accw[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float accb = 0.f;
    for (int gi = 0; gi < GROUPS; ++gi) {
        const int b0 = (blockIdx.x * GROUPS + gi) * 16;
        if (b0 >= B) break;
        __syncthreads();                            // the previous group's reads of xs / ds are done
        for (int i = threadIdx.x; i < 16 * K; i += 256) { const int r = i / K, k = i - r * K; xs[r * KS + k] = (b0 + r < B) ? x[(long)(b0 + r) * K + k] : 0.f; }
        if constexpr (!FWD) {
            for (int i = threadIdx.x; i < 16 * CP; i += 256) { const int r = i / CP, c = i - r * CP; ds[r * CS + c] = (c < C && b0 + r < B) ? dlogits[(long)(b0 + r) * C + c] : 0.f; }
        } else {
            __syncthreads();                        // ws (staged above) and xs are complete
            // ---- forward: thread (sample sm = tid / 16, lane j = tid % 16) owns classes j, j + 16, j + 32 of its sample ----
            const int sm = threadIdx.x >> 4, j = threadIdx.x & 15, b = b0 + sm;
            float *lg = ds + sm * CS;               // the sample's row of the dlogits tile doubles as its logits
            // logits[16 samples][48] = xs[16][K] . ws[K][48] on the fp32 MFMA (a k-ascending fmaf chain per element, as the vector form):
            // wave nt owns the 16-class tile nt; the padded columns of ws are zero
            if (wave < CP / 16) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                for (int kk = 0; kk < K / 4; ++kk) acc = mfma16(xs[li * KS + 4 * kk + lq], ws[(4 * kk + lq) * CS + 16 * wave + li], acc);
                const int cc = 16 * wave + li;
                const float bv = cc < C ? fw.b2[cc] : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) ds[(4 * lq + r) * CS + cc] = acc[r] + bv;
            }
            __syncthreads();
            // the 16 lanes of a sample sit in one wave: LDS operations of a wave are in order, no barrier needed between these steps
            float mx = -INFINITY;
            int am = 0x7fffffff;
            for (int c = j; c < C; c += 16) {
                const float v = lg[c];
                if (v > mx) { mx = v; am = c; }
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) {
                const float omx = __shfl_xor(mx, o, 16);
                const int oam = __shfl_xor(am, o, 16);
                if (omx > mx || (omx == mx && oam < am)) { mx = omx; am = oam; }
            }
            float sum = 0.f;
            for (int c = j; c < C; c += 16) { const float e = expf(lg[c] - mx); lg[c] = e; sum += e; }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 16);
            const float rs = 1.f / sum;
            const bool live = b < B;
            const int y = live ? fw.labels[b] : 0;
            const float py = lg[y] * rs;
            float loss, coef;
            if (fw.class_w) {                       // loss.py:67-71: -log(p_y) * w_y, no clipping
                loss = -logf(py) * fw.class_w[y];
                coef = fw.class_w[y];
            } else {                                // loss.py:36: K.categorical_crossentropy on probabilities (clipped)
                const float lo = kCeEps, hi = 1.f - kCeEps;
                loss = -logf(fminf(fmaxf(py, lo), hi));
                coef = (py >= lo && py <= hi) ? 1.f : 0.f;
            }
            if (fw.ignore_index > 0 && y == fw.ignore_index) { loss = 0.f; coef = 0.f; }   // loss.py:38-40,73-75
            if (live && j == 0) { fw.loss_i_out[b] = loss; fw.correct_i_out[b] = am == y ? 1.f : 0.f; }
            if (live && fw.probs)
                for (int c = j; c < C; c += 16) fw.probs[(long)b * C + c] = lg[c] * rs;
            for (int c = j; c < CP; c += 16)
                lg[c] = (live && c < C) ? (lg[c] * rs - (c == y ? 1.f : 0.f)) * coef * fw.grad_scale : 0.f;
        }
        __syncthreads();
        // dx tiles: one per 16 input features, dealt to the waves
        for (int nt = wave; nt < KT; nt += 4) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < CP / 4; ++j) acc = mfma16(ds[li * CS + 4 * j + lq], ws[(16 * nt + li) * CS + 4 * j + lq], acc);
            float cs = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * lq + r, k = 16 * nt + li;
                if (b0 + row < B) {
                    float v = acc[r];
                    if (RELU6_GATE) { const float xv = xs[row * KS + k]; v = (xv > 0.f && xv < 6.f) ? v : 0.f; }
                    dx[(long)(b0 + row) * K + k] = v;
                    cs += v;
                }
            }
            if (dx_colsum) {                        // lanes with equal li hold the same column: reduce over lq, one atomic per column
                cs += __shfl_xor(cs, 16, 64);
                cs += __shfl_xor(cs, 32, 64);
                if (lq == 0) atomicAdd(dx_colsum + 16 * nt + li, cs);
            }
        }
        // dW2 tiles: (K/16) x 3, dealt to the waves, accumulated over the groups
#pragma unroll
        for (int q = 0; q < MAXT; ++q) {
            const int t = wave + 4 * q;
            if (t < KT * 3) {
                const int mt = t / 3, nt = t - 3 * mt;
#pragma unroll
                for (int j = 0; j < 4; ++j) accw[q] = mfma16(xs[(4 * j + lq) * KS + 16 * mt + li], ds[(4 * j + lq) * CS + 16 * nt + li], accw[q]);
            }
        }
        if ((int)threadIdx.x < C)
            for (int r = 0; r < 16; ++r) accb += ds[r * CS + threadIdx.x];
    }
    // gather the dense (K x C) block in LDS so that the float atomics of a wave-instruction hit 64 contiguous addresses
    // (strided float atomics run ~10x slower)
    __syncthreads();                                // everyone is done with ws: reuse it as dW2[K][C]
#pragma unroll
    for (int q = 0; q < MAXT; ++q) {
        const int t = wave + 4 * q;
        if (t < KT * 3) {
            const int mt = t / 3, nt = t - 3 * mt, c = 16 * nt + li;
            if (c < C) {
#pragma unroll
                for (int r = 0; r < 4; ++r) ws[(16 * mt + 4 * lq + r) * C + c] = accw[q][r];
            }
        }
    }
    __syncthreads();
    if (dw2)
        for (int i = threadIdx.x; i < K * C; i += 256) atomicAdd(dw2 + i, ws[i]);
    if (db2 && (int)threadIdx.x < C) atomicAdd(db2 + threadIdx.x, accb);
    if (stats && blockIdx.x == 0) {                 // deterministic: fixed strided partition, tree in double
        double sl = 0.0, sc = 0.0;
        for (int i = threadIdx.x; i < B; i += 256) { sl += (double)loss_i[i]; sc += (double)correct_i[i]; }
        __syncthreads();
        double *red = reinterpret_cast<double *>(hs);
        red[threadIdx.x] = sl; red[256 + threadIdx.x] = sc;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) { red[threadIdx.x] += red[threadIdx.x + o]; red[256 + threadIdx.x] += red[256 + threadIdx.x + o]; }
            __syncthreads();
        }
        if (threadIdx.x == 0) { stats[0] = (float)red[0]; stats[1] = (float)red[256]; }
    }
}

// per-column sums of an (M x C) matrix into out[C] (bias gradients); reuses the double partial slab
__global__ void colsum_finalize_kernel(const double *__restrict__ partial, int nblk, int C, float *__restrict__ out)
{
    __shared__ double sh[256];
    const int c = blockIdx.x;
    const double s = block_sum_partials(partial, 0, C, c, nblk, sh);
    if (threadIdx.x == 0) out[c] = (float)s;
}

// ---- keras.optimizers.Adam (common/model_utils.py:47): eps OUTSIDE the bias correction -----------------------
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                    float *__restrict__ v, long n, float lr_t, float b1, float b2, float eps,
                                                    float grad_scale)
{
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        float4 pv = *reinterpret_cast<float4 *>(p + i), gv = *reinterpret_cast<const float4 *>(g + i);
        float4 mv = *reinterpret_cast<float4 *>(m + i), vv = *reinterpret_cast<float4 *>(v + i);
        float *pp = &pv.x, *gp = &gv.x, *mp = &mv.x, *vp = &vv.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ge = gp[e] * grad_scale;
            mp[e] = b1 * mp[e] + (1.f - b1) * ge;
            vp[e] = b2 * vp[e] + (1.f - b2) * ge * ge;
            pp[e] -= lr_t * mp[e] / (sqrtf(vp[e]) + eps);
        }
        *reinterpret_cast<float4 *>(p + i) = pv;
        *reinterpret_cast<float4 *>(m + i) = mv;
        *reinterpret_cast<float4 *>(v + i) = vv;
    } else {
        for (long j = i; j < n; ++j) {
            const float ge = g[j] * grad_scale;
            m[j] = b1 * m[j] + (1.f - b1) * ge;
            v[j] = b2 * v[j] + (1.f - b2) * ge * ge;
            p[j] -= lr_t * m[j] / (sqrtf(v[j]) + eps);
        }
    }
}

// classifier/loss.py __call__: per-sample losses from probabilities (or logits), one thread per sample
__global__ __launch_bounds__(256) void loss_forward_kernel(const float *__restrict__ y_pred, const int32_t *__restrict__ labels,
                                                            const float *__restrict__ class_w, int from_logits, int ignore_index,
                                                            int B, int C, float *__restrict__ losses)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float *p = y_pred + (long)b * C;
    const int y = labels[b];
    float py, sum = 0.f;
    if (from_logits) {
        float mx = p[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, p[c]);
        for (int c = 0; c < C; ++c) sum += expf(p[c] - mx);
        py = expf(p[y] - mx) / sum;
    } else {
        for (int c = 0; c < C; ++c) sum += p[c];
        py = class_w ? p[y] : p[y] / sum;     // K.categorical_crossentropy renormalises; the weighted form does not
    }
    float loss = class_w ? -logf(py) * class_w[y] : -logf(fminf(fmaxf(py, kCeEps), 1.f - kCeEps));
    if (ignore_index > 0 && y == ignore_index) loss = 0.f;
    losses[b] = loss;
}

// counts[label][pred] += 1 over a batch (eval.py:201-256 builds the same matrix with sklearn on the host)
__global__ __launch_bounds__(256) void confusion_kernel(const int32_t *__restrict__ labels, const int32_t *__restrict__ pred, int B,
                                                         int C, int32_t *__restrict__ counts)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const int y = labels[b], p = pred[b];
    if (y >= 0 && y < C && p >= 0 && p < C) atomicAdd(counts + y * C + p, 1);
}

__global__ __launch_bounds__(256) void sgd_kernel(float *__restrict__ p, const float *__restrict__ g, long n, float lr, float gs)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] -= lr * (g[i] * gs);
}

__global__ __launch_bounds__(256) void rmsprop_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ a,
                                                       long n, float lr, float rho, float eps, float gs)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float ge = g[i] * gs;
    const float av = rho * a[i] + (1.f - rho) * ge * ge;
    a[i] = av;
    p[i] -= lr * ge / (sqrtf(av) + eps);
}

}  // namespace kws
