// csrc/kws_conv_group.h -- TRAINING forward of conv3 and conv4 of simple_cnn (classifier/models/cnn.py:44-60) at the default geometry in
// the clip-group form of kws_infer_fused.h: a block owns 16 clips, the MFMA row tile of an output position is the 16 clips at that
// position (row = clip), so the (position, tap) pairs that fall into the 'same' padding are skipped as whole tiles (38 of 108), the A
// fragment of (position, tap) is the 16 clips' channel vectors at one input pixel in LDS, and the weights come fragment-major straight
// from L2 (weight_split_slice, frag forms).  Training-mode BatchNormalization needs the batch statistics of every layer before the next
// one starts, so the layers stay separate launches; each writes its pre-activation tensor and the per-block partial sums (sum, sum of
// squares per channel, double) its BatchNorm finalize kernel reduces -- the contract of conv_bf16_kernel<.., STATS>, which these replace:
// 0.051 + 0.048 ms of a forward chain that nothing overlaps (rocprofv3 timeline).
//   conv3: a2 (B, 7, 5, 32) fp32 -> z3 (B, 4, 3, 64); A = fp32 rows split into h / m / l in registers (kws_infer_fused.h: fu_split)
//   conv4: z3 -> a3 = relu6(BN3(z3)) formed and split ONCE per element while it is staged (a3 is never written to memory) -> z4 =
//          relu(conv4(a3)) (B, 4, 3, 128) (Conv2D(activation='relu'), cnn.py:55); A = three bf16 planes, one ds_read_b128 each
#pragma once

namespace kws {

constexpr int kGrThreads = 1024, kGrWaves = kGrThreads / 64;

// partial[(which * C + ch) * stride + blockIdx.x] = this block's sum / sum of squares of channel ch; `red` = 2 * G * C doubles of LDS.
// acc != nullptr: the sums are added to that accumulator set instead (kws_device.h: acc_add) and no finalize kernel follows
template <int C, int G>
__device__ __forceinline__ void group_stats(float s, float ss, int ch, int grp, int lq, double *red, double *__restrict__ partial, int stride,
                                            double *__restrict__ acc = nullptr)
{
    double a = (double)s, q = (double)ss;
    a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
    q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
    if (lq == 0) { red[(grp * 2 + 0) * C + ch] = a; red[(grp * 2 + 1) * C + ch] = q; }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += kGrThreads) {
        const int which = i / C, n = i - which * C;
        double t = 0.0;
#pragma unroll
        for (int g2 = 0; g2 < G; ++g2) t += red[(g2 * 2 + which) * C + n];          // fixed order: deterministic
        if (acc) acc_add(acc, 2 * C, i, t);
        else partial[((long)which * C + n) * stride + blockIdx.x] = t;
    }
}

// z2 != nullptr: the kernel is ALSO layer 2's activation pass -- a2 = maxpool(relu6(BN2(z2))) is formed from z2 (B, 15, 10, 32) while the tile is
// staged and written to `a2w` for the backward pass together with the routed element of every window (zmax2: its pre-BatchNorm value,
// arg2: its index 0..3, first maximum of relu6(y): the contract of bn_act_pool_kernel<true>, kws_layers.h), so that kernel's launch and
// the a2 round trip (18 MB each way) are gone.
// in.acc != nullptr (with z2): BatchNorm-2's scale / shift come from the accumulator set conv2's forward added to (kws_layers.h:
// bn_fwd_coef_prologue) instead of sc2 / sh2; acc != nullptr: this kernel's own sums go to an accumulator set instead of `partial`.
struct GroupConv3Args { const float *a2; const __bf16 *f3[3]; float *z3; double *partial; int stride, B;
                        const float *z2, *sc2, *sh2; float *a2w, *zmax2; unsigned char *arg2; BnAccFwd in; double *acc; };
constexpr int kGrH1 = 2 * kFuH2 + 1, kGrW1 = 2 * kFuW2;                         // conv2's map: 15 x 10 (the last row falls out of the 'valid' pooling)

__global__ __launch_bounds__(kGrThreads, 1) void conv3_group_fwd_kernel(GroupConv3Args g)
{
    extern __shared__ __attribute__((aligned(16))) float gr_lds[];
    float *A2 = gr_lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lq = lane >> 4;
    const int b0 = blockIdx.x * kFuClips;
    const int sw = (li >> 1) & 7;
    constexpr int NP = 12 / (kGrWaves / 4);                                     // positions per wave: 3
    const int ct = wave & 3, grp = wave >> 2;
    bf16x8 b3[3][3];
    fu_load_b(g.f3, ct, lane, b3[0]);
    fu_load_b(g.f3, 4 + ct, lane, b3[1]);
    constexpr int PER = kFuH2 * kFuW2 * kFuC2 / 4;
    if (g.z2) {
        // three items per thread in flight: the twelve window loads of a batch are issued before its first store (a global store may alias
        // a later load for the compiler, and one load -> store round trip per item was 10 us of this kernel)
        static_assert(kGrThreads % 8 == 0, "a thread's channel quad is the same for all of its items");
        const int u = tid & 7;
        __shared__ __attribute__((aligned(16))) float cf2[2 * kFuC2];
        f32x4 sc, sh;
        if (g.in.acc) {
            bn_fwd_coef_prologue(g.in, kFuC2, cf2, cf2 + kFuC2);
            sc = *reinterpret_cast<const f32x4 *>(cf2 + 4 * u); sh = *reinterpret_cast<const f32x4 *>(cf2 + kFuC2 + 4 * u);
        } else {
            sc = *reinterpret_cast<const f32x4 *>(g.sc2 + 4 * u); sh = *reinterpret_cast<const f32x4 *>(g.sh2 + 4 * u);
        }
        constexpr int NB = 3, NIT = (kFuClips * PER + kGrThreads - 1) / kGrThreads;
#pragma unroll
        for (int base = 0; base < NIT; base += NB) {
            f32x4 z[NB][4];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int i = tid + (base + k) * kGrThreads, c = i / PER, r = i - c * PER, px = r >> 3;
                const int ph = px / kFuW2, pw = px - ph * kFuW2;
                const bool ok = i < kFuClips * PER && b0 + c < g.B;
                const float *zp = g.z2 + ((((long)(b0 + c) * kGrH1 + 2 * ph) * kGrW1 + 2 * pw) * kFuC2 + 4 * u);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    z[k][j] = ok ? *reinterpret_cast<const f32x4 *>(zp + ((j >> 1) * kGrW1 + (j & 1)) * kFuC2) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int i = tid + (base + k) * kGrThreads, c = i / PER, r = i - c * PER, px = r >> 3;
                if (i >= kFuClips * PER) continue;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (b0 + c < g.B) {
                    f32x4 zm;
                    unsigned am = 0u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y0 = fmaf(z[k][0][e], sc[e], sh[e]), y1 = fmaf(z[k][1][e], sc[e], sh[e]);
                        const float y2 = fmaf(z[k][2][e], sc[e], sh[e]), y3 = fmaf(z[k][3][e], sc[e], sh[e]);
                        v[e] = relu6f(fmaxf(fmaxf(y0, y1), fmaxf(y2, y3)));
                        unsigned arg = 0u;                   // first maximum of relu6(y): the element the backward pass routes the gradient to
                        float best = relu6f(y0), zz = z[k][0][e];
                        const float v1 = relu6f(y1), v2 = relu6f(y2), v3 = relu6f(y3);
                        if (v1 > best) { best = v1; arg = 1u; zz = z[k][1][e]; }
                        if (v2 > best) { best = v2; arg = 2u; zz = z[k][2][e]; }
                        if (v3 > best) { arg = 3u; zz = z[k][3][e]; }
                        zm[e] = zz;
                        am |= arg << (8 * e);
                    }
                    const long o = ((long)(b0 + c) * PER + r) * 4;
                    *reinterpret_cast<f32x4 *>(g.a2w + o) = v;
                    *reinterpret_cast<f32x4 *>(g.zmax2 + o) = zm;
                    *reinterpret_cast<unsigned *>(g.arg2 + o) = am;
                }
                *reinterpret_cast<f32x4 *>(A2 + ((px * kFuClips + c) * 8 + (u ^ ((c >> 1) & 7))) * 4) = v;
            }
        }
    } else {
        for (int i = tid; i < kFuClips * PER; i += kGrThreads) {
            const int c = i / PER, r = i - c * PER, px = r >> 3, u = r & 7;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (b0 + c < g.B) v = *reinterpret_cast<const f32x4 *>(g.a2 + ((long)(b0 + c) * PER + r) * 4);
            *reinterpret_cast<f32x4 *>(A2 + ((px * kFuClips + c) * 8 + (u ^ ((c >> 1) & 7))) * 4) = v;
        }
    }
    __syncthreads();
    f32x4 acc[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int o0 = 4 * (lq ^ sw), o1 = 4 * ((lq + 4) ^ sw);
#pragma unroll 1
    for (int tap3 = 0; tap3 < 9; tap3 += 3) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int tap = tap3 + d;
            if (tap + 2 < 9) fu_load_b(g.f3, (tap + 2) * 4 + ct, lane, b3[(d + 2) % 3]);
            const int kh = tap3 / 3, kw = d;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int pos = NP * grp + q, oh = pos / kFuW3, ow = pos - oh * kFuW3;
                const int ih = 2 * oh + kh - 1, iw = 2 * ow + kw - 1;
                if (ih >= 0 && ih < kFuH2 && iw >= 0 && iw < kFuW2) {
                    const float *row = A2 + ((ih * kFuW2 + iw) * kFuClips + li) * kFuC2;
                    bf16x8 a[3];
                    fu_split(*reinterpret_cast<const f32x4 *>(row + o0), *reinterpret_cast<const f32x4 *>(row + o1), a);
                    acc[q] = mfma_bf16x6(a, b3[d], acc[q]);
                }
            }
        }
    }
    // z3 and the BatchNorm partial sums (rows of clips past the batch are zero: they add nothing)
    const int ch = 16 * ct + li;
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int pos = NP * grp + q;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int clip = 4 * lq + r;
            const float v = acc[q][r];
            if (b0 + clip < g.B) g.z3[((long)(b0 + clip) * (kFuH3 * kFuW3) + pos) * kFuC3 + ch] = v;
            s += v; ss = fmaf(v, v, ss);
        }
    }
    __syncthreads();                                                            // a2's region becomes the reduction scratch
    group_stats<kFuC3, kGrWaves / 4>(s, ss, ch, grp, lq, reinterpret_cast<double *>(gr_lds), g.partial, g.stride, g.acc);
}

// in.acc / acc: as in GroupConv3Args (BatchNorm-3's coefficients from conv3's accumulator set; this kernel's sums to conv4's)
struct GroupConv4Args { const float *z3, *sc3, *sh3; const __bf16 *f4[3]; float *z4; double *partial; int stride, B; BnAccFwd in; double *acc; };

__global__ __launch_bounds__(kGrThreads, 1) void conv4_group_fwd_kernel(GroupConv4Args g)
{
    extern __shared__ __attribute__((aligned(16))) float gr_lds[];
    __bf16 *A3 = reinterpret_cast<__bf16 *>(gr_lds);                            // three planes of kFuA3P, [position][clip][8 units of 8], swizzled
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lq = lane >> 4;
    const int b0 = blockIdx.x * kFuClips;
    const int sw = (li >> 1) & 7;
    constexpr int NP = 12 / (kGrWaves / 8);                                     // positions per wave: 6
    const int ct = wave & 7, grp = wave >> 3;
    bf16x8 b4[3][3];
    fu_load_b(g.f4, ct, lane, b4[0]);
    fu_load_b(g.f4, 8 + ct, lane, b4[1]);
    // a3 = relu6(BN3(z3)), split once per element, into the planes
    {
        constexpr int PER = kFuH3 * kFuW3 * kFuC3 / 4;                          // float4 per clip: 192
        static_assert(kGrThreads % 16 == 0, "a thread's channel quad is the same for all of its items");
        __shared__ __attribute__((aligned(16))) float cf3[2 * kFuC3];
        f32x4 sc, sh;
        if (g.in.acc) {
            bn_fwd_coef_prologue(g.in, kFuC3, cf3, cf3 + kFuC3);
            sc = *reinterpret_cast<const f32x4 *>(cf3 + 4 * (tid & 15)); sh = *reinterpret_cast<const f32x4 *>(cf3 + kFuC3 + 4 * (tid & 15));
        } else {
            sc = *reinterpret_cast<const f32x4 *>(g.sc3 + 4 * (tid & 15)); sh = *reinterpret_cast<const f32x4 *>(g.sh3 + 4 * (tid & 15));
        }
        for (int i = tid; i < kFuClips * PER; i += kGrThreads) {
            const int c = i / PER, r = i - c * PER, pos = r >> 4, u4 = r & 15;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (b0 + c < g.B) {
                const f32x4 z = *reinterpret_cast<const f32x4 *>(g.z3 + ((long)(b0 + c) * PER + r) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = relu6f(fmaf(z[e], sc[e], sh[e]));
            }
            bf16x4 h, m, l;
            split_bf16(v, h, m, l);
            const int e0 = ((pos * kFuClips + c) * 8 + ((u4 >> 1) ^ ((c >> 1) & 7))) * 8 + (u4 & 1) * 4;
            *reinterpret_cast<bf16x4 *>(A3 + e0) = h;
            *reinterpret_cast<bf16x4 *>(A3 + kFuA3P + e0) = m;
            *reinterpret_cast<bf16x4 *>(A3 + 2 * kFuA3P + e0) = l;
        }
    }
    __syncthreads();
    f32x4 acc[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int ks3 = 0; ks3 < 18; ks3 += 3) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int ks = ks3 + d;
            if (ks + 2 < 18) fu_load_b(g.f4, (long)(ks + 2) * 8 + ct, lane, b4[(d + 2) % 3]);
            const int tap = ks >> 1, chunk = ks & 1, kh = tap / 3, kw = tap - kh * 3;
            const int uo = ((4 * chunk + lq) ^ sw) * 8;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int pos = NP * grp + q, oh = pos / kFuW3, ow = pos - oh * kFuW3;
                const int ih = oh + kh - 1, iw = ow + kw - 1;
                if (ih >= 0 && ih < kFuH3 && iw >= 0 && iw < kFuW3) {
                    const __bf16 *row = A3 + ((ih * kFuW3 + iw) * kFuClips + li) * kFuC3 + uo;
                    bf16x8 a[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const bf16x8 *>(row + p * kFuA3P);
                    acc[q] = mfma_bf16x6(a, b4[d], acc[q]);
                }
            }
        }
    }
    // z4 = relu(conv) (the layer's own activation, in front of its BatchNormalization) and the partial sums
    const int ch = 16 * ct + li;
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int pos = NP * grp + q;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int clip = 4 * lq + r;
            const float v = fmaxf(acc[q][r], 0.f);
            if (b0 + clip < g.B) g.z4[((long)(b0 + clip) * (kFuH3 * kFuW3) + pos) * kFuC4 + ch] = v;
            s += v; ss = fmaf(v, v, ss);
        }
    }
    __syncthreads();
    group_stats<kFuC4, kGrWaves / 8>(s, ss, ch, grp, lq, reinterpret_cast<double *>(gr_lds), g.partial, g.stride, g.acc);
}

// conv4's DATA gradient in the same form: dz4 arrives as the h / m / l bf16 planes BatchNorm-4's backward wrote ((B, 4, 3, 128) each), so
// staging is a copy of 16-byte units into [position][clip][16 units], unit u of clip c at u ^ (c & 15) (256-byte rows: that swizzle makes
// the ds_read_b128 lane groups conflict-free); da3(y, x) = sum over taps of dz4(y + 1 - kh, x + 1 - kw) W[kh][kw]^T, reduced over the 128
// output channels (36 k-steps), 64 columns.  The epilogue is BatchNorm-3's backward reduction (conv3 has no pooling): g = da3 gated by
// ReLU6(BN3(z3)), stored, and the per-block sums of g and g xhat -- the contract of conv_bf16_kernel<128, 64, MODE_DGRAD, EPI_BNBWD_GATE6>.
constexpr int kGrD4P = kFuH3 * kFuW3 * kFuClips * kFuC4;                        // bf16 per plane of the staged dz4
struct GroupDgrad4Args { const __bf16 *dz[3], *fw[3]; const float *z3, *coef; float *g3; double *partial; int stride, B; double *acc; };

__global__ __launch_bounds__(kGrThreads, 1) void conv4_group_dgrad_kernel(GroupDgrad4Args g)
{
    extern __shared__ __attribute__((aligned(16))) float gr_lds[];
    __bf16 *D4 = reinterpret_cast<__bf16 *>(gr_lds);                            // three planes of kGrD4P
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lq = lane >> 4;
    const int b0 = blockIdx.x * kFuClips;
    constexpr int NP = 12 / (kGrWaves / 4);                                     // positions per wave: 3
    const int ct = wave & 3, grp = wave >> 2;
    bf16x8 bw[3][3];
    fu_load_b(g.fw, ct, lane, bw[0]);
    fu_load_b(g.fw, 4 + ct, lane, bw[1]);
    {
        constexpr int PER = kFuH3 * kFuW3 * kFuC4 / 8;                          // 16-byte units per clip and plane: 192
        for (int i = tid; i < 3 * kFuClips * PER; i += kGrThreads) {
            const int p = i / (kFuClips * PER), r0 = i - p * (kFuClips * PER), c = r0 / PER, r = r0 - c * PER, pos = r >> 4, u = r & 15;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (b0 + c < g.B) v = *reinterpret_cast<const bf16x8 *>(g.dz[p] + ((long)(b0 + c) * PER + r) * 8);
            *reinterpret_cast<bf16x8 *>(D4 + p * kGrD4P + ((pos * kFuClips + c) * 16 + (u ^ (c & 15))) * 8) = v;
        }
    }
    __syncthreads();
    f32x4 acc[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int ks3 = 0; ks3 < 36; ks3 += 3) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int ks = ks3 + d;
            if (ks + 2 < 36) fu_load_b(g.fw, (long)(ks + 2) * 4 + ct, lane, bw[(d + 2) % 3]);
            const int tap = ks >> 2, chunk = ks & 3, kh = tap / 3, kw = tap - kh * 3;
            const int uo = ((4 * chunk + lq) ^ li) * 8;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int pos = NP * grp + q, y = pos / kFuW3, x = pos - y * kFuW3;
                const int sy = y + 1 - kh, sx = x + 1 - kw;
                if (sy >= 0 && sy < kFuH3 && sx >= 0 && sx < kFuW3) {
                    const __bf16 *row = D4 + ((sy * kFuW3 + sx) * kFuClips + li) * kFuC4 + uo;
                    bf16x8 a[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const bf16x8 *>(row + p * kGrD4P);
                    acc[q] = mfma_bf16x6(a, bw[d], acc[q]);
                }
            }
        }
    }
    const int ch = 16 * ct + li;
    const float gsc = g.coef[ch], gsh = g.coef[kFuC3 + ch], gmean = g.coef[2 * kFuC3 + ch], ginv = g.coef[3 * kFuC3 + ch];
    float s = 0.f, ss = 0.f;
    float z3v[NP][4];                        // every load is issued before the first store (the stores may alias them for the compiler)
#pragma unroll
    for (int q = 0; q < NP; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int clip = 4 * lq + r;
            z3v[q][r] = b0 + clip < g.B ? g.z3[((long)(b0 + clip) * (kFuH3 * kFuW3) + NP * grp + q) * kFuC3 + ch] : 0.f;
        }
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int pos = NP * grp + q;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int clip = 4 * lq + r;
            if (b0 + clip < g.B) {
                const long e = ((long)(b0 + clip) * (kFuH3 * kFuW3) + pos) * kFuC3 + ch;
                const float zv = z3v[q][r], yv = fmaf(zv, gsc, gsh);
                const float v = (yv > 0.f && yv < 6.f) ? acc[q][r] : 0.f;
                g.g3[e] = v;
                s += v; ss = fmaf(v, (zv - gmean) * ginv, ss);
            }
        }
    }
    __syncthreads();
    group_stats<kFuC3, kGrWaves / 4>(s, ss, ch, grp, lq, reinterpret_cast<double *>(gr_lds), g.partial, g.stride, g.acc);
}

// conv3's DATA gradient (3x3, stride 2): da2(y, x) = sum over the taps with y + 1 - kh and x + 1 - kw even and inside the 4 x 3 map of
// dz3((y + 1 - kh) / 2, (x + 1 - kw) / 2) W[kh][kw]^T -- one to four taps per input pixel, none multiplied that is not needed.  dz3
// (fp32, (B, 4, 3, 64)) is split once into bf16 planes while it is staged (the layout of conv4's forward A operand); 35 output positions
// over eight position groups x two 16-channel column tiles; reduction over the 64 output channels of conv3 (two k-steps per tap).
// acc != nullptr: the epilogue is ALSO BatchNorm-2's backward reduction over the routed elements (the contract of
// bn_bwd_reduce_routed_kernel, kws_layers.h): da2 is gated by ReLU6(BN2(zmax2)) before it is stored, and the block's sums of g and
// g xhat go to the accumulator set -- that kernel's launch (37 us in the step) and the second pass over da2 / zmax2 are gone.
struct GroupDgrad3Args { const float *dz3; const __bf16 *fw[3]; float *da2; int B; const float *zmax2, *coef2; double *acc; };

__global__ __launch_bounds__(kGrThreads, 1) void conv3_group_dgrad_kernel(GroupDgrad3Args g)
{
    extern __shared__ __attribute__((aligned(16))) float gr_lds[];
    __bf16 *D3 = reinterpret_cast<__bf16 *>(gr_lds);                            // three planes of kFuA3P, [position][clip][8 units of 8], swizzled
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lq = lane >> 4;
    const int b0 = blockIdx.x * kFuClips;
    const int sw = (li >> 1) & 7;
    constexpr int NPOS = kFuH2 * kFuW2, NG = kGrWaves / 2, NP = (NPOS + NG - 1) / NG;     // 35 positions, 8 groups of 5
    const int ct = wave & 1, grp = wave >> 1;
    bf16x8 bw[3][3];
    fu_load_b(g.fw, ct, lane, bw[0]);
    fu_load_b(g.fw, 2 + ct, lane, bw[1]);
    {
        constexpr int PER = kFuH3 * kFuW3 * kFuC3 / 4;                          // float4 per clip: 192
        for (int i = tid; i < kFuClips * PER; i += kGrThreads) {
            const int c = i / PER, r = i - c * PER, pos = r >> 4, u4 = r & 15;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (b0 + c < g.B) v = *reinterpret_cast<const f32x4 *>(g.dz3 + ((long)(b0 + c) * PER + r) * 4);
            bf16x4 h, m, l;
            split_bf16(v, h, m, l);
            const int e0 = ((pos * kFuClips + c) * 8 + ((u4 >> 1) ^ ((c >> 1) & 7))) * 8 + (u4 & 1) * 4;
            *reinterpret_cast<bf16x4 *>(D3 + e0) = h;
            *reinterpret_cast<bf16x4 *>(D3 + kFuA3P + e0) = m;
            *reinterpret_cast<bf16x4 *>(D3 + 2 * kFuA3P + e0) = l;
        }
    }
    __syncthreads();
    f32x4 acc[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int ks3 = 0; ks3 < 18; ks3 += 3) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int ks = ks3 + d;
            if (ks + 2 < 18) fu_load_b(g.fw, (long)(ks + 2) * 2 + ct, lane, bw[(d + 2) % 3]);
            const int tap = ks >> 1, chunk = ks & 1, kh = tap / 3, kw = tap - kh * 3;
            const int uo = ((4 * chunk + lq) ^ sw) * 8;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int pos = NP * grp + q, y = pos / kFuW2, x = pos - y * kFuW2;
                const int sy = y + 1 - kh, sx = x + 1 - kw;
                if (pos < NPOS && sy >= 0 && sx >= 0 && !((sy | sx) & 1) && (sy >> 1) < kFuH3 && (sx >> 1) < kFuW3) {
                    const __bf16 *row = D3 + (((sy >> 1) * kFuW3 + (sx >> 1)) * kFuClips + li) * kFuC3 + uo;
                    bf16x8 a[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const bf16x8 *>(row + p * kFuA3P);
                    acc[q] = mfma_bf16x6(a, bw[d], acc[q]);
                }
            }
        }
    }
    const int ch = 16 * ct + li;
    if (g.acc) {
        const float gsc = g.coef2[ch], gsh = g.coef2[kFuC2 + ch], gmean = g.coef2[2 * kFuC2 + ch], ginv = g.coef2[3 * kFuC2 + ch];
        float s = 0.f, sx = 0.f;
        float za[NP][4];                     // every load is issued before the first store (the stores may alias them for the compiler)
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pos = NP * grp + q, clip = 4 * lq + r;
                za[q][r] = (pos < NPOS && b0 + clip < g.B) ? g.zmax2[((long)(b0 + clip) * NPOS + pos) * kFuC2 + ch] : 0.f;
            }
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int pos = NP * grp + q;
            if (pos < NPOS)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int clip = 4 * lq + r;
                    if (b0 + clip < g.B) {
                        const long e = ((long)(b0 + clip) * NPOS + pos) * kFuC2 + ch;
                        const float ya = fmaf(za[q][r], gsc, gsh);
                        const float v = (ya > 0.f && ya < 6.f) ? acc[q][r] : 0.f;
                        g.da2[e] = v;
                        s += v; sx = fmaf(v, (za[q][r] - gmean) * ginv, sx);
                    }
                }
        }
        __syncthreads();
        group_stats<kFuC2, NG>(s, sx, ch, grp, lq, reinterpret_cast<double *>(gr_lds), nullptr, 0, g.acc);
        return;
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int pos = NP * grp + q;
        if (pos < NPOS)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int clip = 4 * lq + r;
                if (b0 + clip < g.B) g.da2[((long)(b0 + clip) * NPOS + pos) * kFuC2 + ch] = acc[q][r];
            }
    }
}

}  // namespace kws
