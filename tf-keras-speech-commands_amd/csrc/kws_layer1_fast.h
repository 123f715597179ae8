// csrc/kws_layer1_fast.h -- layer 1's single backward pass (kws_layer1_moments.h: l1m_bwd_onepass_kernel) for a compile-time map size.
//
// Same products, same outputs (the partial rows l1_bwd_finalize_moments_kernel reduces), different walk.  The generic kernel visits
// the pool windows of a clip four at a time in index order (tile t = windows 4t .. 4t+3); its per-tile bookkeeping -- two window
// coordinate walks with wrap loops, a clamp per operand, the arg-max select -- compiled to ~130 instructions per tile with eleven
// exec-mask branches, against 7 MFMAs of useful work, and every LDS read was waited for on the spot.  The sums this pass collects do
// not care in which order the windows are visited, so here quadrant q of every tile walks its OWN contiguous run of windows
// (window = q * NT + t): a lane's window advances by exactly one per tile, i.e. `+2 floats, or the row jump on wrap' -- one compare,
// two selects, no loop -- H, W are template constants, and the z product of tile t+1 is issued before tile t is routed (explicit software
// pipeline).  (The tile loop is unrolled by groups of ten, one group of da1 values fetched ahead as in the generic kernel: unrolled
// completely, with all 38 values of a clip requested up front, the compiler spilled 143-235 registers at 4 blocks per CU.)
// z itself is the same three-MFMA chain over the taps as in the forward kernel, so the ReLU6 / arg-max decisions agree with it bit for bit.
#pragma once

namespace kws {

template <int H, int W>
__global__ __launch_bounds__(256, 4) void l1f_bwd_onepass_kernel(const float *__restrict__ feat, const float *__restrict__ wk,
                                                               const float *__restrict__ da1, BnCoef k, int B, int clips_per_wave,
                                                               double *__restrict__ partial)
{
    constexpr int WP = W + 2, Wp = W / 2, Hp = H / 2, NWIN = Hp * Wp, NT = (NWIN + 3) / 4, NXS = (H + 2) * WP, HW = H * W;
    constexpr int NST = (NXS + 63) / 64, LAST = NWIN - 3 * NT;        // LAST: windows of quadrant 3 (the others have NT)
    constexpr int kWrap = 2 * WP - 2 * (Wp - 1);                      // offset step from the last window of a row to the first of the next
    static_assert(H % 2 == 0 && W % 2 == 0 && LAST > 0 && LAST <= NT && NST <= kL1Stage, "map size");
    extern __shared__ float l1smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
    float *xs = l1smem + wave * ((NXS + 3) & ~3);

    // conv1 kernel: B fragments W[tap = 4j + lq][c = li] (zero for tap >= 9) and the LDS offset of that tap from a pixel's patch origin
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
    const float sc = k.scale[li], sh = k.shift[li], zmean = k.mean[li];

    // A side of the z product: lane supplies pixel (quadrant qa = li >> 2, element e = li & 3) at tap 4j + lq
    const int qa = li >> 2, e = li & 3, eoff = (e >> 1) * WP + (e & 1);
    const int wa0 = qa * NT, a_ph0 = wa0 / Wp, a_pw0 = wa0 - a_ph0 * Wp;
    // D side (and B side of the G product): lane (channel li, quadrant lq) owns the four elements of window lq * NT + t
    const int wd0 = lq * NT, d_ph0 = wd0 / Wp, d_pw0 = wd0 - d_ph0 * Wp;
    // B side of the G product: x at tap li of the element's pixel; lanes li >= 9 (no such tap) read four halo zeros instead of masking
    const bool tapl = li < 9;
    const int boff = tapl ? (li / 3) * WP + li % 3 : 0;
    constexpr int kZero = WP - 1;     // xs[kZero], xs[kZero + 1], xs[kZero + WP], xs[kZero + WP + 1] are halo cells

    f32x4 accg = {0.f, 0.f, 0.f, 0.f};                 // G[c = 4 lq + r][tap = li], this wave's clips
    double s = 0.0, sz = 0.0;
    float pre[NST];
    auto fetch = [&](long b) {
#pragma unroll
        for (int j = 0; j < NST; ++j) pre[j] = soff[j] >= 0 ? feat[b * HW + soff[j]] : 0.f;
    };
    if (count > 0) fetch(first);
    for (int i = 0; i < count; ++i) {
        // all of the clip's routed-gradient inputs at once: window lq * NT + t, channel li
        const float *dsrc = da1 + ((first + i) * NWIN + wd0) * 16 + li;
        float dcur[kL1Group], dnxt[kL1Group];          // groups of kL1Group tiles, fetched one group ahead
        const int cnt = lq < 3 ? NT : LAST;            // windows of this lane's quadrant
        auto fetch_da = [&](int t0, float (&d)[kL1Group]) {
#pragma unroll
            for (int j = 0; j < kL1Group; ++j) {        // unconditional loads on clamped windows: nothing to branch around
                const int tt = t0 + j, tcl = tt < cnt ? tt : cnt - 1;
                const float v = dsrc[tcl * 16];
                d[j] = tt < cnt ? v : 0.f;
            }
        };
        fetch_da(0, dnxt);
        // this clip's map into the wave's private tile
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

        int a_pw = a_pw0, a_off = 2 * a_ph0 * WP + 2 * a_pw0 + eoff;
        int d_pw = d_pw0, d_off = 2 * d_ph0 * WP + 2 * d_pw0;
        auto zprod = [&](int t) {                       // z of tile t; advances the A-side walk
            const int ao = (t < LAST || qa < 3) ? a_off : eoff;      // past the clip: window 0 (its results are masked by g = 0)
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 3; ++j) acc = mfma16(xs[ao + aoff[j]], wb[j], acc);
            const bool wrap = ++a_pw == Wp;
            a_pw = wrap ? 0 : a_pw;
            a_off += wrap ? kWrap : 2;
            return acc;
        };
        float fs = 0.f, fsz = 0.f;
        f32x4 z = zprod(0);
#pragma nounroll
        for (int t0 = 0; t0 < NT; t0 += kL1Group) {
#pragma unroll
          for (int j = 0; j < kL1Group; ++j) dcur[j] = dnxt[j];
          if (t0 + kL1Group < NT) fetch_da(t0 + kL1Group, dnxt);
#pragma unroll
          for (int jt = 0; jt < kL1Group; ++jt) {
            const int t = t0 + jt;
            if (NT % kL1Group != 0 && t >= NT) continue;
            const f32x4 zn = zprod(t + 1);              // one product past the last tile: a valid address, the result is dropped
            // B operands of the G product for this tile, requested before the routing arithmetic
            const int bo = !tapl ? kZero : ((t < LAST || lq < 3) ? d_off + boff : boff);
            const float x0 = xs[bo], x1 = xs[bo + 1], x2 = xs[bo + WP], x3 = xs[bo + WP + 1];
            {
                const bool wrap = ++d_pw == Wp;
                d_pw = wrap ? 0 : d_pw;
                d_off += wrap ? kWrap : 2;
            }
            // first arg-max of relu6(y) over the window (the rule of every other pass), the gate, and z at the routed element
            const float y0 = fmaf(z[0], sc, sh), y1 = fmaf(z[1], sc, sh), y2 = fmaf(z[2], sc, sh), y3 = fmaf(z[3], sc, sh);
            float best = relu6f(y0), ya = y0, za = z[0];
            const float v1 = relu6f(y1), v2 = relu6f(y2), v3 = relu6f(y3);
            const bool b1 = v1 > best;
            best = b1 ? v1 : best; ya = b1 ? y1 : ya; za = b1 ? z[1] : za;
            const bool b2 = v2 > best;
            best = b2 ? v2 : best; ya = b2 ? y2 : ya; za = b2 ? z[2] : za;
            const bool b3 = v3 > best;
            ya = b3 ? y3 : ya; za = b3 ? z[3] : za;
            const float g = (ya > 0.f && ya < 6.f) ? dcur[jt] : 0.f;
            fs += g;
            fsz = fmaf(g, za - zmean, fsz);
            // one-hot placement of g over the four elements: arg = 3 if b3, else 2 if b2, else 1 if b1, else 0
            const float g3 = b3 ? g : 0.f, g2 = (b2 && !b3) ? g : 0.f, g1 = (b1 && !b2 && !b3) ? g : 0.f, g0 = (b1 || b2 || b3) ? 0.f : g;
            accg = mfma16(g0, x0, accg);
            accg = mfma16(g1, x1, accg);
            accg = mfma16(g2, x2, accg);
            accg = mfma16(g3, x3, accg);
            z = zn;
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
        const int ee = threadIdx.x - 9 * 16, which = ee >> 4, c = ee & 15;
        partial[(long)threadIdx.x * kStatStride + blockIdx.x] = (shs[0][which][c] + shs[1][which][c]) + (shs[2][which][c] + shs[3][which][c]);
    }
}

}  // namespace kws
