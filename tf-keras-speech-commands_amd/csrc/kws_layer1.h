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
// These per-thread kernels serve odd map sizes.  Even maps run on the fp32 MFMA, one wave per clip: the forward kernels further down
// (l1m_act_pool_kernel for inference, L1Mma / l1f_forward_clips shared with the training kernel) and, for training, the second-moment
// form of kws_layer1_moments.h (statistics and gradients in closed form, ONE backward pass: kws_layer1_fast.h for the default map).
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

// ---------------------------------------------------------------------------------------------------------------------
// The same four passes on the fp32 MFMA, one WAVE per clip (used when H and W are even, i.e. every pixel lies in a pool
// window, and the clip has at most 40 tiles).
//
// The per-thread kernels above spend one LDS read per FMA (9 reads per output element) and stage every clip behind two
// block barriers.  Here a 16-pixel x 16-channel tile of z is ONE implicit-GEMM step: A[pixel][tap] (3 LDS reads per lane:
// taps 4j + lq, j = 0..2, the last three zero) times B[tap][channel] (3 registers per lane for the whole kernel) = 3
// v_mfma_f32_16x16x4_f32 per 256 outputs.  The 16 pixels of a tile are 4 consecutive pool windows x their 4 elements
// (pixel p = 4*window + element), so in the D layout (row = 4*lq + r, col = li) lane (li, lq) holds the four elements r of
// window lq for channel li: pooling, arg-max and the ReLU6 gate stay in registers.  The weight gradient is a second MFMA
// per element r: A[channel][k = window] = dz_r (already in A layout), B[k = window][tap] = x at that element's tap (one
// LDS read), D = dW[channel][tap].  fp32 MFMA accumulates the taps in order, so z is the same fmaf chain in all four
// passes (identical ReLU6 / arg-max decisions between them).
//
// Latency, not arithmetic, bounded the first MFMA version (one block per clip sequence: ~10 dependent da1 loads per wave
// and clip, each a full HBM round trip): a wave now owns whole clips, keeps its zero-haloed map in a private LDS tile (no
// block barriers), issues ALL of the clip's da1 loads up front and fetches the next clip's map while it computes.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kL1MaxTiles = 40;      // tiles (4 pool windows each) per clip
constexpr int kL1Group = 10;         // da1 values fetched together, one group ahead of the tiles that use them
constexpr int kL1Stage = 12;         // staged elements per lane: (H+2)(W+2) <= 64 * 12

struct L1Mma {
    float wb[3];      // B fragments of the conv1 kernel: W[tap = 4j + lq][c = li], zero for tap >= 9
    int aoff[3];      // LDS offset of tap 4j + lq from the pixel's patch origin (clamped to tap 8 where the weight is zero)
    int WP, Wp, nwin, nxs, HW, ntile;
    int soff[kL1Stage];    // element offset of staged element (lane + 64 j) inside the clip's (H, W) map, -1 = halo / none
    float pre[kL1Stage];
    float *xs;             // this wave's (H+2) x (W+2) map

    __device__ __forceinline__ void init(const float *__restrict__ wk, int H, int W, float *smem)
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
        WP = W + 2; Wp = W / 2; nwin = (H / 2) * Wp; nxs = (H + 2) * WP; HW = H * W; ntile = (nwin + 3) / 4;
        xs = smem + wave * ((nxs + 3) & ~3);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int tap = 4 * j + lq, tc = tap < 9 ? tap : 8;
            wb[j] = tap < 9 ? wk[tap * 16 + li] : 0.f;
            aoff[j] = (tc / 3) * WP + tc % 3;
        }
#pragma unroll
        for (int j = 0; j < kL1Stage; ++j) {
            const int i = lane + 64 * j, r = i / WP - 1, c = i % WP - 1;
            soff[j] = (i < nxs && r >= 0 && r < H && c >= 0 && c < W) ? r * W + c : -1;
        }
    }
    __device__ __forceinline__ void fetch(const float *__restrict__ feat, long b)
    {
#pragma unroll
        for (int j = 0; j < kL1Stage; ++j) pre[j] = soff[j] >= 0 ? feat[b * HW + soff[j]] : 0.f;
    }
    __device__ __forceinline__ void store() const    // wave-private tile: a wave-level fence orders it against the reads
    {
        const int lane = threadIdx.x & 63;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
        for (int j = 0; j < kL1Stage; ++j) {
            const int i = lane + 64 * j;
            if (i < nxs) xs[i] = pre[j];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // The tiles of a clip are visited in order, so the pool-window coordinates of a lane advance by four windows per tile
    // without a division (an integer division by the runtime Wp per tile was a fifth of these kernels' instructions):
    // (aph, apw) is the A-side window 4 t + (li >> 2) of the z product, (dph, dpw) the D-side window 4 t + lq whose four
    // elements the lane receives.
    int aph, apw, dph, dpw;
    __device__ __forceinline__ void first_tile()
    {
        const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
        aph = (li >> 2) / Wp; apw = (li >> 2) - aph * Wp;
        dph = lq / Wp; dpw = lq - dph * Wp;
    }
    __device__ __forceinline__ void next_tile()
    {
        apw += 4; while (apw >= Wp) { apw -= Wp; ++aph; }
        dpw += 4; while (dpw >= Wp) { dpw -= Wp; ++dph; }
    }
    // LDS offset of element (0, 0) of the lane's D-side window (window 0 when the window lies past the clip)
    __device__ __forceinline__ int d_window_offset(bool ok) const { return ok ? (2 * dph) * WP + 2 * dpw : 0; }
    // z of the current tile: lane (li, lq) gets the four elements of window 4t + lq for channel li
    __device__ __forceinline__ f32x4 z() const
    {
        const int lane = threadIdx.x & 63, li = lane & 15, e = li & 3;
        const bool in = aph < (nwin / Wp);         // windows past the clip are clamped to window 0 (their results are masked by the caller)
        const float *base = xs + (in ? (2 * aph) * WP + 2 * apw : 0) + (e >> 1) * WP + (e & 1);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 3; ++j) acc = mfma16(base[aoff[j]], wb[j], acc);
        return acc;
    }
};

// clips of this wave: [first, first + count)
__device__ __forceinline__ void l1m_clips(int B, int clips_per_wave, long &first, int &count)
{
    first = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * clips_per_wave;
    const long left = (long)B - first;
    count = left <= 0 ? 0 : (left < clips_per_wave ? (int)left : clips_per_wave);
}

// first arg-max of relu6(y) over the window, and the gradient routed to it (same rule as l1_window above)
__device__ __forceinline__ void l1m_route(const f32x4 &z, float sc, float sh, float da, int &arg, float &g)
{
    float y[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = fmaf(z[j], sc, sh);
    arg = 0;
    float best = relu6f(y[0]);
#pragma unroll
    for (int j = 1; j < 4; ++j) {
        const float v = relu6f(y[j]);
        if (v > best) { best = v; arg = j; }
    }
    const float ya = arg == 0 ? y[0] : arg == 1 ? y[1] : arg == 2 ? y[2] : y[3];
    g = (ya > 0.f && ya < 6.f) ? da : 0.f;
}

// Forward body (conv1 -> BN -> ReLU6 -> 2x2 max) for a compile-time map size, shared by the training and the inference kernel.  The
// generic loop below visits the windows four at a time in index order and pays a coordinate walk with wrap loops per tile; a1 does
// not care in which order it is written, so quadrant q of every tile walks its own run of whole window ROWS: RR = ceil(Hp / 4) rows
// = RL = RR * Wp windows (30 x 20 map: 4 rows, 40 windows; quadrant 3 has only 3 real rows).  Window t of a run then sits at the
// compile-time offset (t / Wp) * 2 WP + (t % Wp) * 2 from the run's origin: inside a group of Wp tiles the LDS reads use immediate
// offsets and no lane computes an address (kws_layer1_fast.h has the backward pass in the same form).
template <int H, int W>
struct L1Runs {
    static constexpr int WP = W + 2, Wp = W / 2, Hp = H / 2, NWIN = Hp * Wp, NXS = (H + 2) * WP, HW = H * W;
    static constexpr int RR = (Hp + 3) / 4, RL = RR * Wp;            // window rows / windows per quadrant run
    static constexpr int NST = (NXS + 63) / 64;
    // LDS floats per wave: the haloed map plus the rows a run past the map's end (quadrant 3) still reads
    static constexpr int TILE = ((H + 2 + 2 * (4 * RR - Hp)) * WP + 3) & ~3;
    static_assert(H % 2 == 0 && W % 2 == 0 && NST <= kL1Stage && 4 * RL >= NWIN && 3 * RL < NWIN, "map size");
};

template <int H, int W>
__device__ __forceinline__ void l1f_forward_clips(const float *__restrict__ feat, const float *__restrict__ wk, float sc, float sh,
                                                  float *__restrict__ a1, int B, int clips_per_wave, float *smem)
{
    using R = L1Runs<H, W>;
    constexpr int WP = R::WP, Wp = R::Wp, NWIN = R::NWIN, NXS = R::NXS, HW = R::HW, NST = R::NST, RL = R::RL, RR = R::RR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
    float *xs = smem + wave * R::TILE;
    for (int q = NXS + lane; q < R::TILE; q += 64) xs[q] = 0.f;      // the rows a run past the map's end reads: finite, never stored
    float wb[3];
    int aoff[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int tap = 4 * j + lq, tc = tap < 9 ? tap : 8;
        wb[j] = tap < 9 ? wk[tap * 16 + li] : 0.f;
        aoff[j] = (tc / 3) * WP + tc % 3;
    }
    int soff[NST];
#pragma unroll
    for (int j = 0; j < NST; ++j) {
        const int i = lane + 64 * j, r = i / WP - 1, c = i % WP - 1;
        soff[j] = (i < NXS && r >= 0 && r < H && c >= 0 && c < W) ? r * W + c : -1;
    }
    long first;
    int count;
    l1m_clips(B, clips_per_wave, first, count);
    // A side: lane supplies pixel (quadrant qa = li >> 2, element e = li & 3) at tap 4j + lq; its run starts at window row RR * qa
    const int qa = li >> 2, e = li & 3;
    const float *abase[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) abase[j] = xs + 2 * (RR * qa) * WP + (e >> 1) * WP + (e & 1) + aoff[j];
    const int cnt = NWIN - RL * lq < RL ? NWIN - RL * lq : RL;    // real windows of the quadrant this lane stores
    float pre[NST];
    auto fetch = [&](long b) {
#pragma unroll
        for (int j = 0; j < NST; ++j) pre[j] = soff[j] >= 0 ? feat[b * HW + soff[j]] : 0.f;
    };
    if (count > 0) fetch(first);
    for (int i = 0; i < count; ++i) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
        for (int j = 0; j < NST; ++j) {
            const int q = lane + 64 * j;
            if (q < NXS) xs[q] = pre[j];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (i + 1 < count) fetch(first + i + 1);
        float *out = a1 + ((first + i) * NWIN + RL * lq) * 16 + li;
#pragma nounroll
        for (int row = 0; row < RR; ++row) {            // one window row of every quadrant: Wp tiles with immediate offsets
            const int ro = row * 2 * WP;
            const float *ar[3] = {abase[0] + ro, abase[1] + ro, abase[2] + ro};
            auto zprod = [&](int c) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 3; ++j) acc = mfma16(ar[j][2 * c], wb[j], acc);
                return acc;
            };
            f32x4 z = zprod(0);
#pragma unroll
            for (int c = 0; c < Wp; ++c) {
                const f32x4 zn = c + 1 < Wp ? zprod(c + 1) : z;       // the next tile's product is issued before this tile is finished
                const int t = row * Wp + c;
                const float y0 = fmaf(z[0], sc, sh), y1 = fmaf(z[1], sc, sh), y2 = fmaf(z[2], sc, sh), y3 = fmaf(z[3], sc, sh);
                if (t < cnt) out[t * 16] = relu6f(fmaxf(fmaxf(y0, y1), fmaxf(y2, y3)));
                z = zn;
            }
        }
    }
}

__global__ __launch_bounds__(256) void l1m_act_pool_kernel(const float *__restrict__ feat, const float *__restrict__ wk,
                                                            const float *__restrict__ scale, const float *__restrict__ shift,
                                                            float *__restrict__ a1, int B, int H, int W, int clips_per_wave)
{
    extern __shared__ float l1smem[];
    const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
    if (H == 30 && W == 20) {                     // the default map: compile-time form
        l1f_forward_clips<30, 20>(feat, wk, scale[li], shift[li], a1, B, clips_per_wave, l1smem);
        return;
    }
    L1Mma t;
    t.init(wk, H, W, l1smem);
    long first;
    int count;
    l1m_clips(B, clips_per_wave, first, count);
    const float sc = scale[li], sh = shift[li];
    if (count > 0) t.fetch(feat, first);
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


}  // namespace kws
