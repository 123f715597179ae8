// csrc/kws_layer1_moments.h -- layer 1 of simple_cnn (Conv2D(16, 3x3, 1 input channel) -> BatchNormalization -> ReLU6 ->
// MaxPool, classifier/models/cnn.py:27-33) from the SECOND MOMENTS of the feature map.
//
// conv1 is linear with one input channel: z[b,p,c] = sum_t w[t,c] f[b,p+t] (t = the 9 taps, f = 0 outside the map).  So the
// batch statistics BatchNormalization needs, and every sum of the layer's backward pass that involves z, follow in closed form
// from the 10 x 10 matrix
//     Q[t][t'] = sum over clips b and pixels p of a_t a_t',   a_t = f[b,p+t] for t < 9,  a_9 = 1
// (Q[t][9] = sum of f at tap t, Q[9][9] = B*H*W), which depends on the FEATURES only, not on the weights:
//     mean_c = sum_t w_tc Q[t][9] / M            E[z_c^2] = sum_t sum_t' w_tc w_t'c Q[t][t'] / M
//     dW1[t][c] = k1 ( G[t][c] - k2 S[t] - k3 inv ( sum_t' w_t'c Q[t'][t] - mean_c S[t] ) )          (BatchNorm backward is linear
//                 in the routed gradient g:  G[t][c] = sum g f(p+t),  S[t] = Q[t][9],  k1 = gamma inv, k2 = mean(g), k3 = mean(g xhat))
// What this removes from the train step's critical chain (measured, B = 4096): the statistics pass over the features and its
// finalize launch in the forward pass (23 + 7 us: the activation kernel derives scale / shift in its prologue), and one of the two
// backward passes plus a finalize (30 + 6 us: ONE pass collects G, sum g, sum g z; no second pass that needs dz).  Q itself is 150
// MFMAs per clip (A = B = the im2col row of a pixel) and can be computed wherever the features are -- the input pipeline does it
// on its own stream behind the featurizer (kws_feature_moments), the library does it at the head of the step otherwise.
#pragma once

namespace kws {

constexpr int kMomN = 10, kMomCount = kMomN * kMomN;

// staging helper shared with kws_layer1.h: L1Mma::init reads the conv kernel; the moments kernel has none
__device__ __forceinline__ void l1_stage_init(L1Mma &t, int H, int W, float *smem)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    t.WP = W + 2; t.Wp = W / 2; t.nwin = (H / 2) * t.Wp; t.nxs = (H + 2) * t.WP; t.HW = H * W; t.ntile = (t.nwin + 3) / 4;
    t.xs = smem + wave * ((t.nxs + 3) & ~3);
#pragma unroll
    for (int j = 0; j < kL1Stage; ++j) {
        const int i = lane + 64 * j, r = i / t.WP - 1, c = i % t.WP - 1;
        t.soff[j] = (i < t.nxs && r >= 0 && r < H && c >= 0 && c < W) ? r * W + c : -1;
    }
}

// partial[m * kStatStride + blockIdx.x], m = t * 10 + t': this block's share of Q (double)
__global__ __launch_bounds__(256) void l1_moments_kernel(const float *__restrict__ feat, int B, int H, int W, int clips_per_wave,
                                                          double *__restrict__ partial)
{
    extern __shared__ float l1smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
    L1Mma t;
    l1_stage_init(t, H, W, l1smem);
    long first;
    int count;
    l1m_clips(B, clips_per_wave, first, count);
    // lane (li, lq) supplies a_t of pixel 4 step + lq with t = li: the same value is the A operand (row = tap, k = pixel) and the
    // B operand (k = pixel, col = tap) of D[t][t'] += sum_k a_t(k) a_t'(k)
    const int toff = li < 9 ? (li / 3) * t.WP + li % 3 : 0;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    if (count > 0) t.fetch(feat, first);
    for (int i = 0; i < count; ++i) {
        t.store();
        if (i + 1 < count) t.fetch(feat, first + i + 1);
        f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
        const float *px = t.xs + lq + toff;
        const float one = li == 9 ? 1.f : 0.f;
        const bool tapl = li < 9;
        if (W & 3) {
            // general width: four consecutive pixels of the row-major map per step, the lane's pixel advanced without a division
            int row = lq / W, col = lq - row * W;
            for (int q = lq; q - lq < t.HW; q += 4) {
                float a = tapl ? t.xs[row * t.WP + col + toff] : one;
                a = q < t.HW ? a : 0.f;
                d0 = mfma16(a, a, d0);
                col += 4;
                while (col >= W) { col -= W; ++row; }
                row = row < H ? row : H - 1;                    // lanes past the last pixel read a valid address and are masked
            }
        } else
        // W % 4 == 0: a row is W / 4 steps of four pixels, no per-lane index arithmetic inside the loop; two accumulators so that
        // consecutive MFMAs do not wait for each other
        for (int row = 0; row < H; ++row, px += t.WP) {
            int cg = 0;
            for (; cg + 8 <= W; cg += 8) {
                const float a0 = tapl ? px[cg] : one, a1 = tapl ? px[cg + 4] : one;
                d0 = mfma16(a0, a0, d0);
                d1 = mfma16(a1, a1, d1);
            }
            if (cg < W) {
                const float a0 = tapl ? px[cg] : one;
                d0 = mfma16(a0, a0, d0);
            }
        }
        f32x4 d = d0 + d1;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] += (double)d[r];
        __builtin_amdgcn_wave_barrier();
    }
    __shared__ double red[4][16 * 16];
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][(4 * lq + r) * 16 + li] = acc[r];
    __syncthreads();
    if (threadIdx.x < kMomCount) {
        const int tr = threadIdx.x / kMomN, tc = threadIdx.x - tr * kMomN, e = tr * 16 + tc;
        partial[(long)threadIdx.x * kStatStride + blockIdx.x] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    }
}

// Q[m] = sum over blocks, fixed order (one wave per entry)
__global__ void l1_moments_finalize_kernel(const double *__restrict__ partial, int nblk, double *__restrict__ q)
{
    const double s = wave_sum_partials(partial, 0, 1, blockIdx.x, nblk);
    if (threadIdx.x == 0) q[blockIdx.x] = s;
}

// scale / shift / mean / inv of BatchNorm 1 for channel c from Q and the conv1 kernel (all lanes may call it; double)
__device__ __forceinline__ void l1_bn_from_moments(const double *__restrict__ q, const float *__restrict__ wk, int c, float gamma, float beta,
                                                   double &mean, double &var, double &inv, double &sc, double &sh)
{
    const double M = q[kMomCount - 1];
    double w[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] = (double)wk[t * 16 + c];
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        s1 += w[t] * q[t * kMomN + 9];
        double row = 0.0;
#pragma unroll
        for (int u = 0; u < 9; ++u) row += w[u] * q[t * kMomN + u];
        s2 += w[t] * row;
    }
    mean = s1 / M;
    var = s2 / M - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    inv = 1.0 / sqrt(var + (double)kBnEps);
    sc = (double)gamma * inv;
    sh = (double)beta - mean * sc;
}

// Training forward of layer 1 with the batch statistics taken from Q: every block derives scale / shift for the 16 channels in
// its prologue; block 0 also leaves the coefficients for the backward pass and updates the moving statistics (momentum 0.99,
// unbiased variance, as bn_finalize_train_kernel).  PREP: the grid carries kPrepBlocks extra blocks behind the clip blocks that
// split the conv3 / conv4 / dense weights into bf16 planes and clear the gradient buffer (they used to ride with the statistics
// pass this kernel replaces).
struct L1PrepArgs { SplitDescs all; float *zero_buf; long zero_n; int nsplit, nzero; };
template <bool PREP>
__global__ __launch_bounds__(256) void l1m_act_pool_moments_kernel(const float *__restrict__ feat, const float *__restrict__ wk,
                                                                    const double *__restrict__ q, const float *__restrict__ gamma,
                                                                    const float *__restrict__ beta, float *__restrict__ moving_mean,
                                                                    float *__restrict__ moving_var, BnCoef k, float *__restrict__ a1, int B, int H,
                                                                    int W, int clips_per_wave, int nclip_blocks, L1PrepArgs prep)
{
    if (PREP && (int)blockIdx.x >= nclip_blocks) {
        const int e = blockIdx.x - nclip_blocks;
        if (e < 3 * prep.nsplit) { weight_split_slice(prep.all.d[e / prep.nsplit], e % prep.nsplit, prep.nsplit); return; }
        if (prep.zero_buf)
            for (long i = (long)(e - 3 * prep.nsplit) * 256 + threadIdx.x; i < prep.zero_n; i += (long)prep.nzero * 256) prep.zero_buf[i] = 0.f;
        return;
    }
    extern __shared__ float l1smem[];
    __shared__ float s_sc[16], s_sh[16];
    if (threadIdx.x < 16) {
        const int c = threadIdx.x;
        double mean, var, inv, sc, sh;
        l1_bn_from_moments(q, wk, c, gamma[c], beta[c], mean, var, inv, sc, sh);
        s_sc[c] = (float)sc; s_sh[c] = (float)sh;
        if (blockIdx.x == 0) {
            const double M = q[kMomCount - 1];
            k.scale[c] = (float)sc; k.shift[c] = (float)sh; k.mean[c] = (float)mean; k.inv[c] = (float)inv;
            const double unbiased = var * (M / (M > 1.0 ? M - 1.0 : 1.0));
            moving_mean[c] = (float)((double)moving_mean[c] * kBnMomentum + mean * (1.0 - kBnMomentum));
            moving_var[c] = (float)((double)moving_var[c] * kBnMomentum + unbiased * (1.0 - kBnMomentum));
        }
    }
    const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
    if (H == 30 && W == 20) {                     // the default map: compile-time form (kws_layer1.h: l1f_forward_clips)
        __syncthreads();
        l1f_forward_clips<30, 20>(feat, wk, s_sc[li], s_sh[li], a1, B, clips_per_wave, l1smem);
        return;
    }
    L1Mma t;
    t.init(wk, H, W, l1smem);
    long first;
    int count;
    l1m_clips(B, clips_per_wave, first, count);
    if (count > 0) t.fetch(feat, first);
    __syncthreads();
    const float sc = s_sc[li], sh = s_sh[li];
    for (int i = 0; i < count; ++i) {
        t.store();
        if (i + 1 < count) t.fetch(feat, first + i + 1);
        float *out = a1 + (first + i) * t.nwin * 16 + li;
        t.first_tile();
        for (int tile = 0; tile < t.ntile; ++tile, t.next_tile()) {
            const f32x4 z = t.z();
            const float y0 = fmaf(z[0], sc, sh), y1 = fmaf(z[1], sc, sh), y2 = fmaf(z[2], sc, sh), y3 = fmaf(z[3], sc, sh);
            const int win = 4 * tile + lq;
            if (win < t.nwin) out[win * 16] = relu6f(fmaxf(fmaxf(y0, y1), fmaxf(y2, y3)));
        }
    }
}

// Backward of layer 1 in ONE pass over (features, da1): per block, in double,
//   rows   0..143  G[t][c]  = sum g f(p+t)      (row = t * 16 + c)
//   rows 144..159  SG[c]    = sum g
//   rows 160..175  SGZ[c]   = sum g (z - m)     (z at the routed element, centred by the float mean m = k.mean[c]: the raw sum g z
//                                                cancels against mean * sum g and lost four digits in float per-clip partials)
// to partial[row * kStatStride + blockIdx.x]; g = the gradient routed through max-pool and ReLU6 (same rule as every other pass).
constexpr int kL1BwdRows = 9 * 16 + 32;
// the finalize step of the one-pass backward kernel when it runs in the accumulator form (kws_layer1_fast.h): acc = [kAccSlots][kL1BwdRows]
// doubles + the ticket counter, both zero between passes; q = the feature moments; outputs as l1_bwd_finalize_moments_kernel's
struct L1FinalizeArgs { double *acc = nullptr; unsigned *ticket = nullptr; const double *q = nullptr; const float *gamma = nullptr;
                        float *dw = nullptr, *dgamma = nullptr, *dbeta = nullptr; };
__global__ __launch_bounds__(256, 4) void l1m_bwd_onepass_kernel(const float *__restrict__ feat, const float *__restrict__ wk,
                                                               const float *__restrict__ da1, BnCoef k, int B, int H, int W,
                                                               int clips_per_wave, double *__restrict__ partial)
{
    extern __shared__ float l1smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
    L1Mma t;
    t.init(wk, H, W, l1smem);
    long first;
    int count;
    l1m_clips(B, clips_per_wave, first, count);
    const float sc = k.scale[li], sh = k.shift[li], zmean = k.mean[li];
    // B side of the G product: this lane supplies f at tap li (< 9) of the element's pixel in window 4 tile + lq
    const int tap = li < 9 ? li : 8, boff = (tap / 3) * t.WP + tap % 3;
    const float bmask = li < 9 ? 1.f : 0.f;
    f32x4 accg = {0.f, 0.f, 0.f, 0.f};                 // G[c = 4 lq + r][tap = li], this wave's clips
    double s = 0.0, sz = 0.0;
    if (count > 0) t.fetch(feat, first);
    for (int i = 0; i < count; ++i) {
        const float *dsrc = da1 + (first + i) * t.nwin * 16 + li;
        float dcur[kL1Group], dnxt[kL1Group];
        auto fetch_da = [&](int t0, float (&dv)[kL1Group]) {
#pragma unroll
            for (int j = 0; j < kL1Group; ++j) dv[j] = 4 * (t0 + j) + lq < t.nwin ? dsrc[(4 * (t0 + j) + lq) * 16] : 0.f;
        };
        fetch_da(0, dnxt);
        t.store();
        if (i + 1 < count) t.fetch(feat, first + i + 1);
        float fs = 0.f, fsz = 0.f;
        t.first_tile();
        for (int t0 = 0; t0 < t.ntile; t0 += kL1Group) {
#pragma unroll
            for (int j = 0; j < kL1Group; ++j) dcur[j] = dnxt[j];
            if (t0 + kL1Group < t.ntile) fetch_da(t0 + kL1Group, dnxt);
#pragma unroll
            for (int j = 0; j < kL1Group; ++j) {
                const int tile = t0 + j;
                if (tile < t.ntile) {
                    const int win = 4 * tile + lq;
                    const bool ok = win < t.nwin;
                    const f32x4 z = t.z();
                    int arg;
                    float g;
                    l1m_route(z, sc, sh, dcur[j], arg, g);      // da1 of a window past the clip was fetched as 0: g = 0 there
                    const float za = arg == 0 ? z[0] : arg == 1 ? z[1] : arg == 2 ? z[2] : z[3];
                    fs += g;
                    fsz = fmaf(g, za - zmean, fsz);
                    const float *xb = t.xs + t.d_window_offset(ok) + boff;
                    t.next_tile();
#pragma unroll
                    for (int r = 0; r < 4; ++r) accg = mfma16(r == arg ? g : 0.f, xb[(r >> 1) * t.WP + (r & 1)] * bmask, accg);
                }
            }
        }
        s += (double)fs;
        sz += (double)fsz;
    }
    __shared__ float shg[4][16][17];
    __shared__ double shs[4][2][16];
#pragma unroll
    for (int r = 0; r < 4; ++r) shg[wave][4 * lq + r][li] = accg[r];
    s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
    sz += __shfl_xor(sz, 16, 64); sz += __shfl_xor(sz, 32, 64);
    if (lq == 0) { shs[wave][0][li] = s; shs[wave][1][li] = sz; }
    __syncthreads();
    if (threadIdx.x < 9 * 16) {
        const int tp = threadIdx.x / 16, c = threadIdx.x % 16;
        partial[(long)threadIdx.x * kStatStride + blockIdx.x] =
            ((double)shg[0][c][tp] + (double)shg[1][c][tp]) + ((double)shg[2][c][tp] + (double)shg[3][c][tp]);
    } else if (threadIdx.x < kL1BwdRows) {
        const int e = threadIdx.x - 9 * 16, which = e >> 4, c = e & 15;
        partial[(long)threadIdx.x * kStatStride + blockIdx.x] = (shs[0][which][c] + shs[1][which][c]) + (shs[2][which][c] + shs[3][which][c]);
    }
}

// One wave per output: rows 0..143 -> dW1[t][c], rows 144..159 -> dgamma_c, dbeta_c (and k2 / k3 for completeness).
__global__ void l1_bwd_finalize_moments_kernel(const double *__restrict__ partial, int nblk, const double *__restrict__ q,
                                               const float *__restrict__ wk, const float *__restrict__ gamma, BnCoef k,
                                               float *__restrict__ dw, float *__restrict__ dgamma, float *__restrict__ dbeta)
{
    const int row = blockIdx.x;
    const int c = row < 144 ? row % 16 : row - 144;
    const double sg = wave_sum_partials(partial, 0, 1, 144 + c, nblk);
    const double sgz = wave_sum_partials(partial, 0, 1, 160 + c, nblk);
    const double gsum = row < 144 ? wave_sum_partials(partial, 0, 1, row, nblk) : 0.0;
    if (threadIdx.x != 0) return;
    const double M = q[kMomCount - 1], inv = (double)k.inv[c];
    double mean = 0.0;                                     // the batch mean of z in double (the kernels centred with its float rounding)
#pragma unroll
    for (int u = 0; u < 9; ++u) mean += (double)wk[u * 16 + c] * q[u * kMomN + 9];
    mean /= M;
    const double sgx = inv * (sgz + ((double)k.mean[c] - mean) * sg);      // sum g xhat = inv * sum g (z - mean)
    if (row >= 144) {
        dbeta[c] = (float)sg;
        dgamma[c] = (float)sgx;
        k.k2[c] = (float)(sg / M);
        k.k3[c] = (float)(sgx / M);
        return;
    }
    const int t = row / 16;
    double wq = 0.0;                                       // sum_t' w[t'][c] Q[t'][t] = sum over pixels of z f(p+t)
#pragma unroll
    for (int u = 0; u < 9; ++u) wq += (double)wk[u * 16 + c] * q[u * kMomN + t];
    const double S = q[t * kMomN + 9];
    const double k1 = (double)gamma[c] * inv, k2 = sg / M, k3 = sgx / M;
    dw[t * 16 + c] = (float)(k1 * (gsum - k2 * S - k3 * inv * (wq - mean * S)));
}

}  // namespace kws
