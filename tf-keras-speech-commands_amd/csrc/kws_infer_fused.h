// csrc/kws_infer_fused.h -- inference forward of simple_cnn (classifier/models/cnn.py:44-66, classifier/model.py:37) behind the second
// pooling stage as ONE kernel: conv3 (3x3, stride 2) -> BN -> ReLU6 -> conv4 (3x3, relu) -> BN -> ReLU6 -> MaxPool -> Flatten ->
// Dense(128) + ReLU6 -> Dense(C) + softmax, for the default geometry (a2 = 7 x 5 x 32 per clip).
//
// Before: five launches (conv3, conv4, activation + pool, dense, head) of 19 + 47 + 5 + 17 + 12 us at B = 4096, each a latency-bound
// stage -> barrier -> multiply sequence with its activations making an HBM round trip, against ~25 us of matrix time.  Here a block owns
// 16 clips for the whole chain:
//   * the MFMA row tile IS the clip group: row = clip, one tile per pixel position, so the (position, tap) pairs that fall into the
//     padding are skipped as whole tiles (38 of 108 pairs in both convolutions) and the A fragment of (position, tap) is simply the 16
//     clips' channel vectors at one input pixel: activations live in LDS as [pixel][clip][channels] in fp32;
//   * a2 stays fp32 in LDS and is split into the h / m / l bf16 planes of the three-way product (kws_device.h) in registers when a fragment
//     is read (conv3 is the small product); conv3's epilogue splits a3 ONCE per element into three bf16 planes, so conv4 -- 80 % of the
//     arithmetic -- reads its A fragments with one ds_read_b128 per plane and no vector-ALU work (first version: fp32 a3 split on every
//     read, 0.061 ms; every wave re-split every fragment).  Rows are 128 B (32 floats / 64 bf16) without padding; the 16-byte unit u of
//     row (pixel, clip) sits at u ^ ((clip >> 1) & 7), which makes the ds_read_b128 lane groups {0-3,12-15,20-27}, ... conflict-free;
//   * the weights never touch LDS: infer_frag_kernel lays every (layer, k-step, column tile, plane) out as the 64 x 16 B a wave loads
//     with one coalesced instruction (768 KB in all, L2-resident), loaded two k-steps ahead;
//   * sixteen waves (four per SIMD, 94 registers): a wave owns one 16-channel column tile of conv4 for 6 of the 12 positions = one pooled
//     row (conv3: one column tile x 3 positions), so a block reads every weight fragment twice (conv3: four times) from L2 -- 1.5 MB per
//     block; with eight waves and every fragment read once the dependent read -> product chains of two waves per SIMD were the limit
//     (0.045 against 0.041 ms).
#pragma once

namespace kws {

constexpr int kFuClips = 16;                      // clips per block = MFMA rows
constexpr int kFuH2 = 7, kFuW2 = 5, kFuC2 = 32;   // a2: input of conv3
constexpr int kFuH3 = 4, kFuW3 = 3, kFuC3 = 64;   // conv3 output (stride 2, 'same': pad 1 before on both axes)
constexpr int kFuC4 = 128, kFuH4 = 2, kFuW4 = 1;  // conv4 output 4 x 3 x 128, pooled 2 x 1 x 128
constexpr int kFuFlat = kFuH4 * kFuW4 * kFuC4;    // 256
constexpr int kFuD = 128;                         // Dense units
constexpr int kFuHeadCols = 48;                   // classes padded to three column tiles
constexpr int kFuRS4 = kFuFlat + 8, kFuRSD = kFuD + 8;         // row strides in floats of the small fp32 tiles: 16 B x (2 mod 4)
constexpr int kFuA2 = kFuH2 * kFuW2 * kFuClips * kFuC2;         // floats: a2 [pixel][clip][32], swizzled units, 71 680 B
constexpr int kFuA3P = kFuH3 * kFuW3 * kFuClips * kFuC3;        // bf16 per plane of a3 [position][clip][64], swizzled units
constexpr int kFuLdsBytes = 4 * kFuA2 + 3 * 2 * kFuA3P;         // 145 408 B
constexpr int kFuThreads = 1024;                // 16 waves: 4 per SIMD (108 registers)
constexpr int kFuWaves = kFuThreads / 64, kFuPG = kFuWaves / 8;   // position groups of conv4 (conv3: twice as many)
static_assert(kFuClips * (kFuRS4 + kFuRSD + kFuHeadCols + 2) <= kFuA2, "the tail of the chain lives in a2's region");

// fragment-major weight planes: element ((ks * NCT + ct) * 64 + lane) * 8 + j of plane p = plane p of W[k(ks, lane >> 4, j)][16 ct + (lane & 15)],
// k-step ks = tap * (CI / 32) + chunk; W in Keras HWIO order [(tap * CI + ci) * CO + co].  The k order inside a step follows the A side:
// natural = 0 (A = fp32 rows: a lane's 8 values are the 4-float units lq and lq + 4): channel = 32 chunk + 4 (lane >> 4) + (j & 3) + 16 (j >> 2);
// natural = 1 (A = bf16 planes: a lane's 8 values are ONE 16-byte unit): channel = 32 chunk + 8 (lane >> 4) + j
struct FragDesc { const float *w; __bf16 *p[3]; int taps, ci, co, nct, natural; };
struct FragDescs { FragDesc d[4]; };
__global__ __launch_bounds__(256) void infer_frag_kernel(FragDescs all)
{
    const FragDesc &d = all.d[blockIdx.y];
    const int ksteps = d.taps * (d.ci / 32);
    const long total = (long)ksteps * d.nct * 512;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
        const long f = i >> 9;
        const int ct = (int)(f % d.nct), ks = (int)(f / d.nct);
        const int tap = ks / (d.ci / 32), chunk = ks - tap * (d.ci / 32);
        const int ch = 32 * chunk + (d.natural ? 8 * (lane >> 4) + j : 4 * (lane >> 4) + (j & 3) + 16 * (j >> 2)), col = 16 * ct + (lane & 15);
        const float v = col < d.co ? d.w[((long)tap * d.ci + ch) * d.co + col] : 0.f;
        const __bf16 h = (__bf16)v;
        const float r1 = v - (float)h;
        const __bf16 m = (__bf16)r1;
        d.p[0][i] = h; d.p[1][i] = m; d.p[2][i] = (__bf16)(r1 - (float)m);
    }
}

struct FusedTailArgs {
    const float *a2;                  // (B, 7, 5, 32)
    const __bf16 *f3[3], *f4[3], *fd[3], *fh[3];     // fragment-major planes of conv3 / conv4 / dense / head
    const float *sc3, *sh3, *sc4, *sh4;              // folded BatchNorm coefficients of layers 3 and 4
    const float *db, *hb;             // dense bias, head bias
    float *probs;                     // (B, C) or nullptr
    int32_t *argmax;                  // (B) or nullptr
    int B, C;
};

__device__ __forceinline__ void fu_split(const f32x4 lo, const f32x4 hi, bf16x8 (&a)[3])
{
    bf16x4 h0, m0, l0, h1, m1, l1;
    split_bf16(lo, h0, m0, l0);
    split_bf16(hi, h1, m1, l1);
    a[0] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
    a[1] = __builtin_shufflevector(m0, m1, 0, 1, 2, 3, 4, 5, 6, 7);
    a[2] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
}
// the A fragment of this lane: 8 floats of its row (clip li) at the k-step's units lq and lq + 4
__device__ __forceinline__ void fu_load_a(const float *row, bf16x8 (&a)[3])
{
    const f32x4 lo = *reinterpret_cast<const f32x4 *>(row), hi = *reinterpret_cast<const f32x4 *>(row + 16);
    fu_split(lo, hi, a);
}
__device__ __forceinline__ void fu_load_b(const __bf16 *const (&pl)[3], long frag, int lane, bf16x8 (&b)[3])
{
#pragma unroll
    for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const bf16x8 *>(pl[p] + (frag * 64 + lane) * 8);
}

__global__ __launch_bounds__(kFuThreads, 1) void infer_tail_kernel(FusedTailArgs g)
{
    extern __shared__ __attribute__((aligned(16))) float fu_lds[];
    float *A2 = fu_lds;
    __bf16 *A3 = reinterpret_cast<__bf16 *>(fu_lds + kFuA2);                    // three planes of kFuA3P
    float *A4 = fu_lds, *D1 = A4 + kFuClips * kFuRS4, *LG = D1 + kFuClips * kFuRSD, *MS = LG + kFuClips * kFuHeadCols;   // alias a2 once conv3 is done
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lq = lane >> 4;
    const int b0 = blockIdx.x * kFuClips;
    const int sw = (li >> 1) & 7;                                               // this lane's row (clip li) keeps unit u at u ^ sw

    // conv3's first weight fragments are on their way while a2 is staged
    constexpr int NP3 = 12 / (kFuWaves / 4), NP4 = 12 / kFuPG;                   // positions per wave in conv3 / conv4
    const int ct3 = wave & 3, half3 = wave >> 2;
    bf16x8 b3[3][3];
    fu_load_b(g.f3, ct3, lane, b3[0]);
    fu_load_b(g.f3, 4 + ct3, lane, b3[1]);

    // ---- a2 of the block's clips -> LDS [pixel][clip][8 units of 4 floats], unit u at u ^ ((clip >> 1) & 7) ----
    {
        constexpr int PER = kFuH2 * kFuW2 * kFuC2 / 4;                          // float4 per clip: 280
        for (int i = tid; i < kFuClips * PER; i += kFuThreads) {
            const int c = i / PER, r = i - c * PER, px = r >> 3, u = r & 7;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (b0 + c < g.B) v = *reinterpret_cast<const f32x4 *>(g.a2 + ((long)(b0 + c) * PER + r) * 4);
            *reinterpret_cast<f32x4 *>(A2 + ((px * kFuClips + c) * 8 + (u ^ ((c >> 1) & 7))) * 4) = v;
        }
    }
    __syncthreads();

    // ---- conv3: wave = (column tile ct of 4, half of the 12 positions); A = fp32 rows split in registers ----
    {
        f32x4 acc[NP3];
#pragma unroll
        for (int q = 0; q < NP3; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int o0 = 4 * (lq ^ sw), o1 = 4 * ((lq + 4) ^ sw);                  // the lane's two units of the 32-channel step
#pragma unroll 1
        for (int tap3 = 0; tap3 < 9; tap3 += 3) {                               // runtime loop: one kernel row per trip, its three taps unrolled
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int tap = tap3 + d;
                if (tap + 2 < 9) fu_load_b(g.f3, (tap + 2) * 4 + ct3, lane, b3[(d + 2) % 3]);     // two k-steps ahead
                const int kh = tap3 / 3, kw = d;
#pragma unroll
                for (int q = 0; q < NP3; ++q) {
                    const int pos = NP3 * half3 + q, oh = pos / kFuW3, ow = pos - oh * kFuW3;
                    const int ih = 2 * oh + kh - 1, iw = 2 * ow + kw - 1;
                    if (ih >= 0 && ih < kFuH2 && iw >= 0 && iw < kFuW2) {       // wave-uniform: the padding taps of this position are skipped
                        const float *row = A2 + ((ih * kFuW2 + iw) * kFuClips + li) * kFuC2;
                        bf16x8 a[3];
                        fu_split(*reinterpret_cast<const f32x4 *>(row + o0), *reinterpret_cast<const f32x4 *>(row + o1), a);
                        acc[q] = mfma_bf16x6(a, b3[d], acc[q]);
                    }
                }
            }
        }
        // conv4's first weight fragments travel under the epilogue and the barrier
        // BatchNorm (moving statistics, folded) + ReLU6 -> a3 as three bf16 planes [position][clip][8 units of 8], unit u at u ^ ((clip >> 1) & 7)
        const int ch = 16 * ct3 + li;
        const float sc = g.sc3[ch], sh = g.sh3[ch];
#pragma unroll
        for (int q = 0; q < NP3; ++q) {
            const int pos = NP3 * half3 + q;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int clip = 4 * lq + r;
                const float v = relu6f(fmaf(acc[q][r], sc, sh));
                const __bf16 h = (__bf16)v;
                const float r1 = v - (float)h;
                const __bf16 m = (__bf16)r1;
                const int e = ((pos * kFuClips + clip) * 8 + ((ch >> 3) ^ ((clip >> 1) & 7))) * 8 + (ch & 7);
                A3[e] = h; A3[kFuA3P + e] = m; A3[2 * kFuA3P + e] = (__bf16)(r1 - (float)m);
            }
        }
    }
    bf16x8 b4[3][3];                                                            // conv4's weights: a ring of three k-steps
    const int ct4 = wave & 7, pg4 = wave >> 3;
    fu_load_b(g.f4, ct4, lane, b4[0]);
    fu_load_b(g.f4, 8 + ct4, lane, b4[1]);
    __syncthreads();

    // ---- conv4 (activation='relu') -> BN -> ReLU6 -> 2 x 2 max-pool: wave = column tile (8 of 16 channels), all 12 positions; A = bf16 planes ----
    {
        const int ct = ct4;
        f32x4 acc[NP4];
#pragma unroll
        for (int q = 0; q < NP4; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int ks3 = 0; ks3 < 18; ks3 += 3) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int ks = ks3 + d;
                if (ks + 2 < 18) fu_load_b(g.f4, (long)(ks + 2) * 8 + ct, lane, b4[(d + 2) % 3]);   // two k-steps ahead
                const int tap = ks >> 1, chunk = ks & 1, kh = tap / 3, kw = tap - kh * 3;
                const int uo = ((4 * chunk + lq) ^ sw) * 8;
#pragma unroll
                for (int q = 0; q < NP4; ++q) {
                    const int pos = NP4 * pg4 + q, oh = pos / kFuW3, ow = pos - oh * kFuW3;
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
        // relu -> BN -> ReLU6, then the 2 x 2 maximum: a position group of 6 is two map rows = one pooled row
        static_assert(NP4 == 12 || NP4 == 6, "a wave holds whole pooling windows");
        const int ch = 16 * ct + li;
        const float sc = g.sc4[ch], sh = g.sh4[ch];
#pragma unroll
        for (int ph = 0; ph < NP4 / 6; ++ph)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float best = 0.f;                                                // ReLU6 outputs are >= 0
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        const int q = (2 * ph + dy) * kFuW3 + dx;                // pooled column 0 = columns 0, 1 (column 2 is dropped: 'valid')
                        best = fmaxf(best, relu6f(fmaf(fmaxf(acc[q][r], 0.f), sc, sh)));
                    }
                A4[(4 * lq + r) * kFuRS4 + ((NP4 == 6 ? pg4 : ph) * kFuW4) * kFuC4 + ch] = best;      // Flatten is (h, w, c)
            }
    }
    __syncthreads();

    // ---- Dense(128) + ReLU6: waves 0..7 = column tiles ----
    if (wave < kFuD / 16) {
        const int ct = wave;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        bf16x8 bcur[3], bnext[3];
        fu_load_b(g.fd, ct, lane, bcur);
#pragma unroll
        for (int ks = 0; ks < kFuFlat / 32; ++ks) {
            if (ks + 1 < kFuFlat / 32) fu_load_b(g.fd, (long)(ks + 1) * 8 + ct, lane, bnext);
            bf16x8 a[3];
            fu_load_a(A4 + li * kFuRS4 + 32 * ks + 4 * lq, a);
            acc = mfma_bf16x6(a, bcur, acc);
#pragma unroll
            for (int p = 0; p < 3; ++p) bcur[p] = bnext[p];
        }
        const int ch = 16 * ct + li;
        const float bias = g.db[ch];
#pragma unroll
        for (int r = 0; r < 4; ++r) D1[(4 * lq + r) * kFuRSD + ch] = relu6f(acc[r] + bias);
    }
    __syncthreads();

    // ---- Dense(C): waves 0..2 = column tiles ----
    if (wave < kFuHeadCols / 16) {
        const int ct = wave;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < kFuD / 32; ++ks) {
            bf16x8 a[3], b[3];
            fu_load_b(g.fh, (long)ks * (kFuHeadCols / 16) + ct, lane, b);
            fu_load_a(D1 + li * kFuRSD + 32 * ks + 4 * lq, a);
            acc = mfma_bf16x6(a, b, acc);
        }
        const int col = 16 * ct + li;
        const float bias = col < g.C ? g.hb[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) LG[(4 * lq + r) * kFuHeadCols + col] = acc[r] + bias;
    }
    __syncthreads();

    // ---- softmax / arg-max per clip (first maximum wins, like np.argmax) ----
    if (tid < kFuClips) {
        const float *x = LG + tid * kFuHeadCols;
        float mx = x[0];
        int am = 0;
        for (int c = 1; c < g.C; ++c)
            if (x[c] > mx) { mx = x[c]; am = c; }
        float s = 0.f;
        for (int c = 0; c < g.C; ++c) s += expf(x[c] - mx);
        MS[2 * tid] = mx;
        MS[2 * tid + 1] = 1.0f / s;
        if (g.argmax && b0 + tid < g.B) g.argmax[b0 + tid] = am;
    }
    __syncthreads();
    if (g.probs)
        for (int i = tid; i < kFuClips * g.C; i += kFuThreads) {
            const int c = i / g.C, col = i - c * g.C;
            if (b0 + c < g.B) g.probs[(long)(b0 + c) * g.C + col] = expf(LG[c * kFuHeadCols + col] - MS[2 * c]) * MS[2 * c + 1];
        }
}

}  // namespace kws
