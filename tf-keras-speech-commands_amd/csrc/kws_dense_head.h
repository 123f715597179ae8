// csrc/kws_dense_head.h -- the train step of simple_cnn between the last pooling stage and BatchNorm-4's backward pass as ONE kernel:
//   Dense(128) + ReLU6 forward (cnn.py:63-66) -> Dense(C) + softmax + loss (classifier/model.py:37, classifier/loss.py) -> dlogits ->
//   head backward (dW2, db2, dd1 gated by the Dense layer's ReLU6) -> Dense data gradient da4 (and the Dense bias gradient).
// Before: conv_bf16_fwd<128,128> (19 us alone / 32 us beside the next batch's featurizer) -> head_fwd_bwd_kernel (29 / 50) ->
// conv_bf16_dgrad<128,128> (19 / 21) on the step's main chain, the three smallest products of the model as three dependent launches.
// A block owns 16 samples (the head kernel's tile, kws_layers.h: head_bwd_mfma_kernel<true, 1, true>, whose forward / loss / backward
// code this reuses line by line); the two Dense products run in the clip-group form of kws_infer_fused.h: A = the block's 16 fp32 rows in
// LDS split into h / m / l bf16 in registers, B = fragment-major weight planes straight from L2 (weight_split_slice: frag = 1 for the
// forward planes, ofrag = 1 for the data gradient's), three-way bf16 split products with fp32 accumulation like the kernels it replaces.
// The Dense weight gradient stays a kernel of its own on the side stream (it reduces over the whole batch): dd1 is still written.
#pragma once

namespace kws {

constexpr int kDhK = 128, kDhCP = 48, kDhCS = 50;          // Dense units; padded classes; LDS row stride of the class tiles
constexpr int kDhKS = kDhK + 8;                            // row stride of the 16 x 128 fp32 tiles: 544 B = 16 B x (2 mod 4)

struct DenseHeadArgs {
    const float *a4;                  // (B, flat): the pooled, dropped, flattened map
    const __bf16 *fd[3], *fo[3];      // Dense weights: forward planes (fragment-major, frag = 1), data-gradient planes (ofrag = 1)
    const float *db, *w2;             // Dense bias, head kernel (128, C)
    float *d1, *dd1, *da4;            // Dense output (kept for the deterministic paths' tests: optional), its gradient, the map's gradient
    float *dw2, *db2, *ddb;           // gradients: head kernel / bias, Dense bias (float atomics)
    int B, C, flat;
    HeadFwdArgs fw;
    // acc4 != nullptr: the epilogue is ALSO BatchNorm-4's backward reduction over the routed elements (what bn_bwd_reduce_routed_full_kernel
    // did from da4, kws_layers.h): da4 leaves as the dropped, ReLU6-gated gradient per (pool window, channel) -- the compact form
    // bn_bwd_apply_routed_planes_kernel expands -- and the block's sums of g and g xhat go to the accumulator set (kws_device.h: acc_add)
    const float *zmax4 = nullptr, *coef4 = nullptr;               // z4 at the routed element; BN4's scale | shift | mean | inv (128 each)
    double *acc4 = nullptr;
    float drop_rate = 0.f;
    uint32_t seed_lo = 0, seed_hi = 0;
    // z4 != nullptr: the kernel is ALSO layer 4's activation pass (bn_act_pool_acc_kernel's contract, kws_layers.h): BatchNorm-4's scale /
    // shift come from the accumulator set conv4's forward added to (in4), a4 = dropout(maxpool(relu6(BN4(z4)))) is formed from z4 (B, H3, W3,
    // 128) while the tile is staged and written to a4w (the Dense weight gradient reads it) with zmax4w / arg4w for the backward pass
    const float *z4 = nullptr;
    BnAccFwd in4{};
    float *a4w = nullptr, *zmax4w = nullptr;
    unsigned char *arg4w = nullptr;
    int H3 = 0, W3 = 0;
};

// kDhWaves waves per block of 16 samples: every phase below is a chain of dependent LDS / L2 round trips with little work per wave, and
// the kernel shares its CU with the next batch's featurizer (twelve vector-ALU-bound waves) -- eight waves halve the serial work of a
// wave in the Dense product, the data gradient and the dW2 tiles against four (four waves: 0.037 ms alone, 0.115 under the featurizer).
constexpr int kDhWaves = 8, kDhThreads = 64 * kDhWaves;
__global__ __launch_bounds__(kDhThreads) void dense_head_fused_kernel(DenseHeadArgs g)
{
    constexpr int NW = kDhWaves, NT = kDhThreads;
    static_assert(NW == 4 || NW == 8, "column tiles per wave below: 8 / NW Dense tiles, 24 / NW dW2 tiles");
    constexpr int DT = 8 / NW, WT = 24 / NW;                      // Dense column tiles, dW2 tiles per wave
    extern __shared__ __attribute__((aligned(16))) float dh_lds[];
    const int RS4 = g.flat + 8;                                   // row stride of the a4 tile: 16 B x (2 mod 4) for flat = 256 (multiple of 32)
    float *a4s = dh_lds;                                          // [16][RS4]
    float *xs = a4s + 16 * RS4;                                   // [16][kDhKS]  d1
    float *dd = xs + 16 * kDhKS;                                  // [16][kDhKS]  dd1
    float *ds = dd + 16 * kDhKS;                                  // [16][kDhCS]  logits -> dlogits, columns >= C zero
    float *ws = ds + 16 * kDhCS;                                  // [128][kDhCS] W2, columns >= C zero; later dW2[128][C]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lq = lane >> 4;
    const int b0 = blockIdx.x * 16, C = g.C, K = kDhK;
    const int nks = g.flat / 32;

    // the first Dense weight fragments travel while the tiles are staged
    bf16x8 bcur[DT][3], bnext[DT][3];
#pragma unroll
    for (int t = 0; t < DT; ++t) fu_load_b(g.fd, DT * wave + t, lane, bcur[t]);
    // W2 -> LDS: all of a thread's loads are issued here (one L2 round trip; a load -> LDS store loop was twelve dependent ones, most of the
    // 9 us this kernel spent in front of its first product) and stored behind the tile's staging
    constexpr int NW2 = kDhK * kDhCP / NT;
    static_assert(kDhK * kDhCP % NT == 0, "whole rounds");
    float w2v[NW2];
#pragma unroll
    for (int j = 0; j < NW2; ++j) { const int i = tid + j * NT, k = i / kDhCP, c = i - k * kDhCP; w2v[j] = c < C ? g.w2[(long)k * C + c] : 0.f; }
    __shared__ __attribute__((aligned(16))) float cf4[4 * kDhK];   // z4: BatchNorm-4's scale | shift | mean | inv, derived here
    if (g.z4) {
        bn_fwd_coef_prologue(g.in4, kDhK, cf4, cf4 + kDhK, cf4 + 2 * kDhK, cf4 + 3 * kDhK);
        const int W4 = g.W3 / 2, nf4 = g.flat / 4;                // pooled width; float4 units per clip
        constexpr int NB = 2;                                     // items per thread in flight: their eight window loads precede the first store
        for (int base = tid; base < 16 * nf4; base += NB * NT) {
            f32x4 z[NB][4];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int i = base + NT * k, r = i / nf4, u = i - r * nf4, win = u / (kDhK / 4), c0 = 4 * (u - win * (kDhK / 4));
                const int ph = win / W4, pw = win - ph * W4;
                const bool ok = i < 16 * nf4 && b0 + r < g.B;
                const float *zp = g.z4 + ((((long)(b0 + r) * g.H3 + 2 * ph) * g.W3 + 2 * pw) * kDhK + c0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    z[k][j] = ok ? *reinterpret_cast<const f32x4 *>(zp + ((j >> 1) * g.W3 + (j & 1)) * kDhK) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int i = base + NT * k, r = i / nf4, u = i - r * nf4, c0 = 4 * (u % (kDhK / 4));
                if (i >= 16 * nf4) continue;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (b0 + r < g.B) {
                    const f32x4 sc = *reinterpret_cast<const f32x4 *>(cf4 + c0), sh = *reinterpret_cast<const f32x4 *>(cf4 + kDhK + c0);
                    const long o = (long)(b0 + r) * g.flat + 4 * u;
                    f32x4 zm;
                    unsigned am = 0u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y0 = fmaf(z[k][0][e], sc[e], sh[e]), y1 = fmaf(z[k][1][e], sc[e], sh[e]);
                        const float y2 = fmaf(z[k][2][e], sc[e], sh[e]), y3 = fmaf(z[k][3][e], sc[e], sh[e]);
                        float a = relu6f(fmaxf(fmaxf(y0, y1), fmaxf(y2, y3)));
                        unsigned arg = 0u;                   // first maximum of relu6(y): the element the backward pass routes the gradient to
                        float best = relu6f(y0), zz = z[k][0][e];
                        const float v1 = relu6f(y1), v2 = relu6f(y2), v3 = relu6f(y3);
                        if (v1 > best) { best = v1; arg = 1u; zz = z[k][1][e]; }
                        if (v2 > best) { best = v2; arg = 2u; zz = z[k][2][e]; }
                        if (v3 > best) { arg = 3u; zz = z[k][3][e]; }
                        zm[e] = zz;
                        am |= arg << (8 * e);
                        if (g.drop_rate > 0.f) a = dropout_keep(g.seed_lo, g.seed_hi, (uint32_t)(o + e), g.drop_rate) ? a / (1.f - g.drop_rate) : 0.f;
                        v[e] = a;
                    }
                    *reinterpret_cast<f32x4 *>(g.a4w + o) = v;
                    *reinterpret_cast<f32x4 *>(g.zmax4w + o) = zm;
                    *reinterpret_cast<unsigned *>(g.arg4w + o) = am;
                }
                *reinterpret_cast<f32x4 *>(a4s + r * RS4 + 4 * u) = v;
            }
        }
    } else
    for (int i = tid; i < 16 * (g.flat / 4); i += NT) {
        const int r = i / (g.flat / 4), u = i - r * (g.flat / 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (b0 + r < g.B) v = *reinterpret_cast<const f32x4 *>(g.a4 + (long)(b0 + r) * g.flat + 4 * u);
        *reinterpret_cast<f32x4 *>(a4s + r * RS4 + 4 * u) = v;
    }
#pragma unroll
    for (int j = 0; j < NW2; ++j) { const int i = tid + j * NT, k = i / kDhCP, c = i - k * kDhCP; ws[k * kDhCS + c] = w2v[j]; }
    __syncthreads();

    // ---- Dense(128) + ReLU6: wave = column tiles DT wave .. DT wave + DT - 1 ----
    {
        f32x4 acc[DT];
#pragma unroll
        for (int t = 0; t < DT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < nks; ++ks) {
            if (ks + 1 < nks)
#pragma unroll
                for (int t = 0; t < DT; ++t) fu_load_b(g.fd, (long)(ks + 1) * 8 + DT * wave + t, lane, bnext[t]);
            bf16x8 a[3];
            fu_load_a(a4s + li * RS4 + 32 * ks + 4 * lq, a);
#pragma unroll
            for (int t = 0; t < DT; ++t) acc[t] = mfma_bf16x6(a, bcur[t], acc[t]);
#pragma unroll
            for (int t = 0; t < DT; ++t)
#pragma unroll
                for (int p = 0; p < 3; ++p) bcur[t][p] = bnext[t][p];
        }
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            const int ch = 16 * (DT * wave + t) + li;
            const float bias = g.db[ch];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = relu6f(acc[t][r] + bias);
                xs[(4 * lq + r) * kDhKS + ch] = v;
                if (g.d1 && b0 + 4 * lq + r < g.B) g.d1[(long)(b0 + 4 * lq + r) * K + ch] = v;
            }
        }
    }
    // the data gradient's first fragments: column tiles wave, wave + 4, ... of the flat map
    const int nct = g.flat / 16, tpw = nct / NW;                  // column tiles of da4, per wave
    __syncthreads();

    // ---- head forward: thread (sample sm = tid / 16, lane j = tid % 16) owns classes j, j + 16, j + 32 of its sample (kws_layers.h) ----
    const HeadFwdArgs &fw = g.fw;
    {
        const int sm = (tid >> 4) & 15, j = tid & 15, b = b0 + sm;            // waves 4 .. NW - 1 have no sample: they only meet the barriers
        const bool has = tid < 256;
        float *lg = ds + sm * kDhCS;
        if (wave < kDhCP / 16) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int kk = 0; kk < K / 4; ++kk) acc = mfma16(xs[li * kDhKS + 4 * kk + lq], ws[(4 * kk + lq) * kDhCS + 16 * wave + li], acc);
            const int cc = 16 * wave + li;
            const float bv = cc < C ? fw.b2[cc] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) ds[(4 * lq + r) * kDhCS + cc] = acc[r] + bv;
        }
        __syncthreads();
        if (has) {
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
        const bool live = b < g.B;
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
        for (int c = j; c < kDhCP; c += 16)
            lg[c] = (live && c < C) ? (lg[c] * rs - (c == y ? 1.f : 0.f)) * coef * fw.grad_scale : 0.f;
        }
    }
    __syncthreads();

    // ---- head backward: dd1 tiles (gated by the Dense layer's ReLU6), its column sums = the Dense bias gradient, dW2 tiles, db2 ----
    f32x4 accw[WT];
#pragma unroll
    for (int q = 0; q < WT; ++q) accw[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int nt = wave; nt < K / 16; nt += NW) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < kDhCP / 4; ++j) acc = mfma16(ds[li * kDhCS + 4 * j + lq], ws[(16 * nt + li) * kDhCS + 4 * j + lq], acc);
        float cs = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * lq + r, k = 16 * nt + li;
            const float xv = xs[row * kDhKS + k];
            float v = (xv > 0.f && xv < 6.f) ? acc[r] : 0.f;
            if (b0 + row >= g.B) v = 0.f;
            dd[row * kDhKS + k] = v;
            if (b0 + row < g.B) g.dd1[(long)(b0 + row) * K + k] = v;
            cs += v;
        }
        cs += __shfl_xor(cs, 16, 64);
        cs += __shfl_xor(cs, 32, 64);
        if (lq == 0) atomicAdd(g.ddb + 16 * nt + li, cs);
    }
#pragma unroll
    for (int q = 0; q < WT; ++q) {
        const int t = wave + NW * q, mt = t / 3, nt = t - 3 * mt;           // (K / 16) x 3 = 24 tiles, WT per wave
#pragma unroll
        for (int j = 0; j < 4; ++j) accw[q] = mfma16(xs[(4 * j + lq) * kDhKS + 16 * mt + li], ds[(4 * j + lq) * kDhCS + 16 * nt + li], accw[q]);
    }
    float accb = 0.f;
    if (tid < C)
        for (int r = 0; r < 16; ++r) accb += ds[r * kDhCS + tid];
    __syncthreads();                                // dd complete; everyone is done with ws: it becomes dW2[K][C]

    // ---- Dense data gradient: da4[16][flat] = dd1 . Wd^T, reduction over the 128 units (4 k-steps), column tile nt = wave + 4 i ----
    {
        bf16x8 a[4][3];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const float *row = dd + li * kDhKS + 32 * kk + 8 * lq;          // natural k order (ofrag planes): one unit of 8 floats
            fu_split(*reinterpret_cast<const f32x4 *>(row), *reinterpret_cast<const f32x4 *>(row + 4), a[kk]);
        }
        float za[4] = {0.f, 0.f, 0.f, 0.f}, zan[4] = {0.f, 0.f, 0.f, 0.f};     // acc4: zmax4 of this / the next tile (loaded ahead of the stores)
        if (g.acc4)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (b0 + 4 * lq + r < g.B) za[r] = g.zmax4[(long)(b0 + 4 * lq + r) * g.flat + 16 * wave + li];
        for (int i = 0; i < tpw; ++i) {
            const int nt = wave + NW * i, tap = nt / (kDhK / 16), ctl = nt - tap * (kDhK / 16);  // flat column tile -> (tap, 16-channel tile)
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            bf16x8 b[4][3];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) fu_load_b(g.fo, (long)(tap * 4 + kk) * (kDhK / 16) + ctl, lane, b[kk]);
            if (g.acc4 && i + 1 < tpw)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (b0 + 4 * lq + r < g.B) zan[r] = g.zmax4[(long)(b0 + 4 * lq + r) * g.flat + 16 * (nt + NW) + li];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc = mfma_bf16x6(a[kk], b[kk], acc);
            if (g.acc4) {
                // flat index = (pool window) * 128 + channel: dropout mask, ReLU6 gate on y = BN4(z of the routed element), sums
                const int ch = 16 * ctl + li;
                // (with z4 the global coefficient arrays are being written by block 0 of THIS launch: every block uses its own copy)
                const float *cf = g.z4 ? cf4 : g.coef4;
                const float gsc = cf[ch], gsh = cf[kDhK + ch], gmean = cf[2 * kDhK + ch], ginv = cf[3 * kDhK + ch];
                float s = 0.f, sx = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (b0 + 4 * lq + r < g.B) {
                        const long e = (long)(b0 + 4 * lq + r) * g.flat + 16 * nt + li;
                        float v = acc[r];
                        if (g.drop_rate > 0.f) v = dropout_keep(g.seed_lo, g.seed_hi, (uint32_t)e, g.drop_rate) ? v / (1.f - g.drop_rate) : 0.f;
                        const float ya = fmaf(za[r], gsc, gsh);
                        v = (ya > 0.f && ya < 6.f) ? v : 0.f;
                        g.da4[e] = v;
                        s += v; sx = fmaf(v, (za[r] - gmean) * ginv, sx);
                    }
                double sd = (double)s, sxd = (double)sx;
                sd += __shfl_xor(sd, 16, 64); sd += __shfl_xor(sd, 32, 64);
                sxd += __shfl_xor(sxd, 16, 64); sxd += __shfl_xor(sxd, 32, 64);
                if (lq == 0) { acc_add(g.acc4, 2 * kDhK, ch, sd); acc_add(g.acc4, 2 * kDhK, kDhK + ch, sxd); }
#pragma unroll
                for (int r = 0; r < 4; ++r) za[r] = zan[r];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (b0 + 4 * lq + r < g.B) g.da4[(long)(b0 + 4 * lq + r) * g.flat + 16 * nt + li] = acc[r];
            }
        }
    }
    // gather the dense (K x C) block of dW2 in LDS so that the float atomics of a wave-instruction hit 64 contiguous addresses
#pragma unroll
    for (int q = 0; q < WT; ++q) {
        const int t = wave + NW * q, mt = t / 3, nt = t - 3 * mt, c = 16 * nt + li;
        if (c < C) {
#pragma unroll
            for (int r = 0; r < 4; ++r) ws[(16 * mt + 4 * lq + r) * C + c] = accw[q][r];
        }
    }
    __syncthreads();
    for (int i = tid; i < K * C; i += NT) atomicAdd(g.dw2 + i, ws[i]);
    if (tid < C) atomicAdd(g.db2 + tid, accb);
}

}  // namespace kws
