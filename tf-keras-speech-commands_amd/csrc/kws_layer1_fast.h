// csrc/kws_layer1_fast.h -- layer 1's single backward pass (kws_layer1_moments.h: l1m_bwd_onepass_kernel) for a compile-time map size.
//
// Same products, same outputs (the partial rows l1_bwd_finalize_moments_kernel reduces), different walk.  The generic kernel visits
// the pool windows of a clip four at a time in index order (tile t = windows 4t .. 4t+3); its per-tile bookkeeping -- two window
// coordinate walks with wrap loops, a clamp per operand, the arg-max select -- compiled to ~130 instructions per tile with eleven
// exec-mask branches, against 7 MFMAs of useful work, and every LDS read was waited for on the spot.  The sums this pass collects do
// not care in which order the windows are visited, so here quadrant q of every tile walks its OWN run of whole window rows (kws_layer1.h:
// L1Runs -- 4 rows = 40 windows per quadrant on the default map, the last quadrant has 3 real rows): inside a row every LDS address is the
// lane's row pointer plus a compile-time offset, the loop over the 10 tiles of a row unrolls completely without any coordinate
// arithmetic, and the z product of tile t+1 is issued before tile t is routed.  (An earlier form walked 38-window runs with a wrap test
// per tile: 65 instructions per tile.  Unrolled completely with all da1 values requested up front the compiler spilled 143-235
// registers at 4 blocks per CU; da1 comes one window row ahead.)
// z itself is the same three-MFMA chain over the taps as in the forward kernel, so the ReLU6 / arg-max decisions agree with it bit for bit.
#pragma once

namespace kws {

template <int H, int W>
__global__ __launch_bounds__(256, 4) void l1f_bwd_onepass_kernel(const float *__restrict__ feat, const float *__restrict__ wk,
                                                               const float *__restrict__ da1, BnCoef k, int B, int clips_per_wave,
                                                               double *__restrict__ partial, L1FinalizeArgs fin = L1FinalizeArgs{})
{
    using R = L1Runs<H, W>;                            // quadrant q of every tile walks window rows RR q .. RR q + RR - 1 (kws_layer1.h)
    constexpr int WP = R::WP, Wp = R::Wp, NWIN = R::NWIN, NXS = R::NXS, HW = R::HW, NST = R::NST, RL = R::RL, RR = R::RR;
    extern __shared__ float l1smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
    float *xs = l1smem + wave * R::TILE;
    for (int q = NXS + lane; q < R::TILE; q += 64) xs[q] = 0.f;      // the rows a run past the map's end reads: finite, masked by g = 0

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
    const int qa = li >> 2, e = li & 3;
    const float *abase[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) abase[j] = xs + 2 * (RR * qa) * WP + (e >> 1) * WP + (e & 1) + aoff[j];
    // D side: lane (channel li, quadrant lq) owns the four elements of window RL lq + t.  B side of the G product: x at tap li of the
    // element's pixel; lanes li >= 9 (no such tap) read four halo zeros instead (xs[WP - 1], [WP], [2 WP - 1], [2 WP] are halo cells)
    const bool tapl = li < 9;
    const float *bbase = tapl ? xs + 2 * (RR * lq) * WP + (li / 3) * WP + li % 3 : xs + (WP - 1);
    const int bstep = tapl ? 1 : 0;                     // lanes without a tap do not move
    const int cnt = NWIN - RL * lq < RL ? NWIN - RL * lq : RL;    // real windows of this lane's quadrant

    f32x4 accg = {0.f, 0.f, 0.f, 0.f};                 // G[c = 4 lq + r][tap = li], this wave's clips
    double s = 0.0, sz = 0.0;
    float pre[NST];
    auto fetch = [&](long b) {
#pragma unroll
        for (int j = 0; j < NST; ++j) pre[j] = soff[j] >= 0 ? feat[b * HW + soff[j]] : 0.f;
    };
    if (count > 0) fetch(first);
    for (int i = 0; i < count; ++i) {
        // the clip's routed-gradient inputs: window RL lq + t, channel li, one window row (Wp values) ahead
        const float *dsrc = da1 + ((first + i) * NWIN + RL * lq) * 16 + li;
        float dcur[Wp], dnxt[Wp];
        auto fetch_da = [&](int t0, float (&d)[Wp]) {
#pragma unroll
            for (int j = 0; j < Wp; ++j) {              // unconditional loads on clamped windows: nothing to branch around
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

        float fs = 0.f, fsz = 0.f;
#pragma nounroll
        for (int row = 0; row < RR; ++row) {            // one window row of every quadrant: Wp tiles at compile-time offsets
#pragma unroll
            for (int j = 0; j < Wp; ++j) dcur[j] = dnxt[j];
            if (row + 1 < RR) fetch_da((row + 1) * Wp, dnxt);
            const int ro = row * 2 * WP;
            const float *ar[3] = {abase[0] + ro, abase[1] + ro, abase[2] + ro};
            const float *br = bbase + bstep * ro;
            auto zprod = [&](int c) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 3; ++j) acc = mfma16(ar[j][2 * c], wb[j], acc);
                return acc;
            };
            f32x4 z = zprod(0);
#pragma unroll
            for (int c = 0; c < Wp; ++c) {
                const f32x4 zn = c + 1 < Wp ? zprod(c + 1) : z;       // the next tile's product is issued before this tile is routed
                // B operands of the G product for this tile, requested before the routing arithmetic
                const float *bp = br + bstep * 2 * c;
                const float x0 = bp[0], x1 = bp[1], x2 = bp[WP], x3 = bp[WP + 1];
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
                const float g = (ya > 0.f && ya < 6.f) ? dcur[c] : 0.f;     // windows past the clip were fetched as 0
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
    double mine = 0.0;                                   // this block's sum of row threadIdx.x (rows as l1_bwd_finalize_moments_kernel reads them)
    if (threadIdx.x < 9 * 16) {
        const int tp = threadIdx.x / 16, c = threadIdx.x % 16;
        mine = ((double)shg[0][c][tp] + (double)shg[1][c][tp]) + ((double)shg[2][c][tp] + (double)shg[3][c][tp]);
    } else if (threadIdx.x < kL1BwdRows) {
        const int ee = threadIdx.x - 9 * 16, which = ee >> 4, c = ee & 15;
        mine = (shs[0][which][c] + shs[1][which][c]) + (shs[2][which][c] + shs[3][which][c]);
    }
    if (!fin.acc) {
        if (threadIdx.x < kL1BwdRows) partial[(long)threadIdx.x * kStatStride + blockIdx.x] = mine;
        return;
    }
    // No finalize launch (the last kernel of the step's main chain, 6.6 us of pure latency): every block ADDS its rows to an accumulator
    // set (kws_device.h: acc_add; eight slots), takes a ticket, and the block that draws the last one evaluates
    // l1_bwd_finalize_moments_kernel's closed forms from the sums, clears the set and the ticket counter for the next pass.  Only atomics
    // carry data between the blocks (the sums are read back with agent-scope atomic loads).
    // No __threadfence(): an agent-scope fence writes back and INVALIDATES the XCD's L2 -- with one per block the kernel took 165 us instead
    // of 68 (every other block of the XCD lost its cached features).  Instead every wave waits for the acknowledgement of its own atomics
    // (s_waitcnt vmcnt(0): on gfx9 the counter covers atomics without return; the barrier alone does NOT wait for them) before the barrier
    // behind which thread 0 draws the ticket, and the last block reads the sums back with agent-scope atomic loads.
    if (threadIdx.x < kL1BwdRows) atomicAdd(fin.acc + (blockIdx.x & (kAccSlots - 1)) * kL1BwdRows + threadIdx.x, mine);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ unsigned last_block;
    if (threadIdx.x == 0) last_block = atomicAdd(fin.ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!last_block) return;
    __shared__ double rows[kL1BwdRows];
    if (threadIdx.x < kL1BwdRows) {
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < kAccSlots; ++j) {
            double *p = fin.acc + j * kL1BwdRows + threadIdx.x;
            t += __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *p = 0.0;                                    // consumed by the next pass's kernels only
        }
        rows[threadIdx.x] = t;
    }
    if (threadIdx.x == 0) *fin.ticket = 0u;
    __syncthreads();
    if (threadIdx.x < 144 + 16) {
        const int row = threadIdx.x, c = row < 144 ? row % 16 : row - 144;
        const double sg = rows[144 + c], sgz = rows[160 + c];
        const double M = fin.q[kMomCount - 1], inv = (double)k.inv[c];
        double mean = 0.0;                               // the batch mean of z in double (the kernels centred with its float rounding)
#pragma unroll
        for (int u = 0; u < 9; ++u) mean += (double)wk[u * 16 + c] * fin.q[u * kMomN + 9];
        mean /= M;
        const double sgx = inv * (sgz + ((double)k.mean[c] - mean) * sg);      // sum g xhat = inv * sum g (z - mean)
        if (row >= 144) {
            fin.dbeta[c] = (float)sg;
            fin.dgamma[c] = (float)sgx;
            k.k2[c] = (float)(sg / M);
            k.k3[c] = (float)(sgx / M);
        } else {
            const int t = row / 16;
            double wq = 0.0;                             // sum_t' w[t'][c] Q[t'][t] = sum over pixels of z f(p+t)
#pragma unroll
            for (int u = 0; u < 9; ++u) wq += (double)wk[u * 16 + c] * fin.q[u * kMomN + t];
            const double S = fin.q[t * kMomN + 9];
            const double k1 = (double)fin.gamma[c] * inv, k2 = sg / M, k3 = sgx / M;
            fin.dw[t * 16 + c] = (float)(k1 * (rows[row] - k2 * S - k3 * inv * (wq - mean * S)));
        }
    }
}

}  // namespace kws
