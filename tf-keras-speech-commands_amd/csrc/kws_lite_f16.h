// csrc/kws_lite_f16.h -- simple_cnn_lite inference in fp16 (BASELINE configs[4]: streaming inference, featurize + forward
// captured in a hipGraph, fp16).
//
// Reference topology: classifier/models/cnn.py:77-141 (SimpleCNNLite) and the softmax head of classifier/model.py:37.
// The fused front kernel (kws_lite.h: both leading SeparableConv2D -> BN -> ReLU6 -> MaxPool stages, one wave per clip)
// writes its pooled activation a2 as fp16; everything behind it is ONE kernel here:
//   depthwise 3 (stride 2) -> pointwise 3 + bias + relu -> BN -> ReLU6 -> depthwise 4 -> pointwise 4 + bias + relu -> BN ->
//   ReLU6 -> 2x2 max-pool -> flatten -> Dense(128) + ReLU6 -> Dense(C) + softmax.
// "fp16" means: the activations between stages and all matrix operands are fp16; every accumulation (depthwise taps,
// MFMA, bias / BatchNorm affine, softmax) is fp32.  A block owns 16 clips: rows of the pointwise products are
// (clip, pixel) pairs (16 x 12 = 192 rows = 12 row tiles), Dense and the head are one row tile.  fp16 halves the weights to
// 36 KB (+ 64 KB for Dense, streamed from L2), so they sit in LDS beside two ping-pong activation regions and the whole
// back end of the network runs without touching HBM between the a2 read and the probability write.  Matrix products run
// on v_mfma_f32_16x16x32_f16; LDS rows are padded so that the 16-byte fragment reads of 16 consecutive rows fall into
// different banks: ds_read_b128 serves the lane groups {0-3,12-15,20-27}, ..., and reads at (row li, 16-byte piece lq) are
// conflict-free for row strides of 16 B x (2 mod 4) -- 96, 160, 288 and 544 bytes here.
#pragma once
#include <hip/hip_fp16.h>

#include "kws_layers.h"

namespace kws {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int kF16Clips = 16;                    // clips per block tile
constexpr int kF16MaxP3 = 12, kF16MaxN2 = 35;    // geometry limits of the fused back end (default geometry: 4x3 and 7x5)
constexpr int kF16HeadCols = 48;                 // head columns in LDS (C <= 48)
// fp16 weight blob (halfs), written by lite_f16_prepare_kernel: transposed [n][k] rows, padded by 8 halfs where staged in LDS
constexpr int kW3Row = 48, kW4Row = 80, kW2Row = 144, kWdRow = 256;
constexpr int kBlobW3 = 0, kBlobW4 = kBlobW3 + 64 * kW3Row, kBlobW2 = kBlobW4 + 128 * kW4Row,
              kBlobLds = kBlobW2 + kF16HeadCols * kW2Row,      // the part of the blob a block copies into LDS
              kBlobWd = kBlobLds, kBlobHalfs = kBlobWd + 128 * kWdRow;

struct LiteF16Args {
    const float *dwk3, *pwb3, *sc3, *sh3;        // depthwise kernel [9][32], pointwise bias [64], BN scale / shift [64]
    const float *dwk4, *pwb4, *sc4, *sh4;        // [9][64], [128], [128]
    const float *db, *hb;                        // Dense bias [128], head bias [C]
    const _Float16 *blob;
    int C, H2, W2, H3, W3, pt3, pl3, H4, W4;
};

// params -> fp16 blob: pointwise / dense / head kernels transposed to [n][k]
__global__ __launch_bounds__(256) void lite_f16_prepare_kernel(const float *__restrict__ pwk3, const float *__restrict__ pwk4,
                                                                const float *__restrict__ dk, const float *__restrict__ hk, int C,
                                                                int flat, _Float16 *__restrict__ blob)
{
    for (int i = blockIdx.x * 256 + threadIdx.x; i < kBlobHalfs; i += gridDim.x * 256) {
        float v = 0.f;
        if (i < kBlobW4) {
            const int n = i / kW3Row, k = i % kW3Row;
            if (k < 32) v = pwk3[k * 64 + n];
        } else if (i < kBlobW2) {
            const int j = i - kBlobW4, n = j / kW4Row, k = j % kW4Row;
            if (k < 64) v = pwk4[k * 128 + n];
        } else if (i < kBlobWd) {
            const int j = i - kBlobW2, n = j / kW2Row, k = j % kW2Row;
            if (k < 128 && n < C) v = hk[k * C + n];
        } else {
            const int j = i - kBlobWd, n = j / kWdRow, k = j % kWdRow;
            if (k < flat) v = dk[k * 128 + n];
        }
        blob[i] = (_Float16)v;
    }
}

__device__ __forceinline__ f32x4 mfma_f16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// LDS carve (bytes): two activation regions that alternate, then the weights
constexpr int kR1Bytes = 192 * 288;              // A2 (16 x 35 x 64 B) | A3 (192 x 160 B) | Z4 (192 x 288 B) | D1 + logits
constexpr int kR2Bytes = 192 * 160;              // D3 (192 x 96 B) | D4 (192 x 160 B) | A4 (16 x 544 B)
constexpr int kF16LdsBytes = kR1Bytes + kR2Bytes + 2 * kBlobLds;

__global__ __launch_bounds__(256) void lite_back_f16_kernel(const _Float16 *__restrict__ a2, LiteF16Args k, int B, float *__restrict__ probs,
                                                            int32_t *__restrict__ argmax)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
    unsigned char *R1 = fsm, *R2 = fsm + kR1Bytes;
    const _Float16 *Wl = reinterpret_cast<const _Float16 *>(fsm + kR1Bytes + kR2Bytes);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lq = lane >> 4;
    const int n2 = k.H2 * k.W2, P3 = k.H3 * k.W3, n4 = k.H4 * k.W4, flat = n4 * 128;
    const int rows = kF16Clips * P3, mtiles = rows / 16;          // 16 clips: always a whole number of row tiles

    // weights -> LDS once per block (blob rows are already padded)
    for (int i = tid; i < kBlobLds / 8; i += 256)
        reinterpret_cast<f16x8 *>(fsm + kR1Bytes + kR2Bytes)[i] = reinterpret_cast<const f16x8 *>(k.blob)[i];
    // per-thread depthwise weights: the thread's 4-channel group is the same for every item it processes
    float w3[9][4], w4[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) { w3[t][e] = k.dwk3[t * 32 + 4 * (tid & 7) + e]; w4[t][e] = k.dwk4[t * 64 + 4 * (tid & 15) + e]; }

    const int ntile = (B + kF16Clips - 1) / kF16Clips;
    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int b0 = tile * kF16Clips;
        __syncthreads();                                          // the previous tile's logits are consumed, the weights are in place
        // ---- a2 of the tile's clips -> R1 [clip][pixel][32] (same layout as in HBM; clips past B read as zero)
        {
            const int units = kF16Clips * n2 * 4;                 // 16-byte units
            const f16x8 *src = reinterpret_cast<const f16x8 *>(a2 + (long)b0 * n2 * 32);
            const long valid = (long)(B - b0 < kF16Clips ? B - b0 : kF16Clips) * n2 * 4;
            for (int i = tid; i < units; i += 256) {
                f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (i < valid) v = src[i];
                reinterpret_cast<f16x8 *>(R1)[i] = v;
            }
        }
        __syncthreads();
        // ---- depthwise 3 (3x3, stride 2, 'same'): D3[row = clip * P3 + pixel][32] in R2, row stride 96 B
        for (int i = tid; i < rows * 8; i += 256) {
            const int g = i & 7, row = i >> 3, c = row / P3, p = row - c * P3, oy = p / k.W3, ox = p - oy * k.W3;
            float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int y = oy * 2 + t / 3 - k.pt3, x = ox * 2 + t % 3 - k.pl3;
                if (y >= 0 && y < k.H2 && x >= 0 && x < k.W2) {
                    const f16x4 v = *reinterpret_cast<const f16x4 *>(R1 + ((c * n2 + y * k.W2 + x) * 32 + 4 * g) * 2);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = fmaf((float)v[e], w3[t][e], o[e]);
                }
            }
            f16x4 h = {(_Float16)o[0], (_Float16)o[1], (_Float16)o[2], (_Float16)o[3]};
            *reinterpret_cast<f16x4 *>(R2 + row * 96 + 8 * g) = h;
        }
        __syncthreads();
        // ---- pointwise 3 (32 -> 64) + bias + relu (cnn.py:113), BN, ReLU6: A3[row][64] in R1, row stride 160 B
        {
            f16x8 bf[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) bf[nt] = *reinterpret_cast<const f16x8 *>(Wl + kBlobW3 + (16 * nt + li) * kW3Row + 8 * lq);
            for (int mt = wave; mt < mtiles; mt += 4) {
                const f16x8 af = *reinterpret_cast<const f16x8 *>(R2 + (16 * mt + li) * 96 + 16 * lq);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const f32x4 acc = mfma_f16(af, bf[nt], (f32x4){0.f, 0.f, 0.f, 0.f});
                    const int col = 16 * nt + li;
                    const float bias = k.pwb3[col], sc = k.sc3[col], sh = k.sh3[col];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        *reinterpret_cast<_Float16 *>(R1 + (16 * mt + 4 * lq + r) * 160 + 2 * col) =
                            (_Float16)relu6f(fmaf(fmaxf(acc[r] + bias, 0.f), sc, sh));
                }
            }
        }
        __syncthreads();
        // ---- depthwise 4 (3x3, stride 1, 'same') over the (H3 x W3) map: D4[row][64] in R2, row stride 160 B
        for (int i = tid; i < rows * 16; i += 256) {
            const int g = i & 15, row = i >> 4, c = row / P3, p = row - c * P3, oy = p / k.W3, ox = p - oy * k.W3;
            float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int y = oy + t / 3 - 1, x = ox + t % 3 - 1;
                if (y >= 0 && y < k.H3 && x >= 0 && x < k.W3) {
                    const f16x4 v = *reinterpret_cast<const f16x4 *>(R1 + (c * P3 + y * k.W3 + x) * 160 + 8 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = fmaf((float)v[e], w4[t][e], o[e]);
                }
            }
            f16x4 h = {(_Float16)o[0], (_Float16)o[1], (_Float16)o[2], (_Float16)o[3]};
            *reinterpret_cast<f16x4 *>(R2 + row * 160 + 8 * g) = h;
        }
        __syncthreads();
        // ---- pointwise 4 (64 -> 128) + bias + relu (cnn.py:122), BN, ReLU6: Z4[row][128] in R1, row stride 288 B
        for (int mt = wave; mt < mtiles; mt += 4) {
            const f16x8 a0 = *reinterpret_cast<const f16x8 *>(R2 + (16 * mt + li) * 160 + 16 * lq);
            const f16x8 a1 = *reinterpret_cast<const f16x8 *>(R2 + (16 * mt + li) * 160 + 64 + 16 * lq);
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) {
                const f16x8 b0f = *reinterpret_cast<const f16x8 *>(Wl + kBlobW4 + (16 * nt + li) * kW4Row + 8 * lq);
                const f16x8 b1f = *reinterpret_cast<const f16x8 *>(Wl + kBlobW4 + (16 * nt + li) * kW4Row + 32 + 8 * lq);
                f32x4 acc = mfma_f16(a0, b0f, (f32x4){0.f, 0.f, 0.f, 0.f});
                acc = mfma_f16(a1, b1f, acc);
                const int col = 16 * nt + li;
                const float bias = k.pwb4[col], sc = k.sc4[col], sh = k.sh4[col];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    *reinterpret_cast<_Float16 *>(R1 + (16 * mt + 4 * lq + r) * 288 + 2 * col) =
                        (_Float16)relu6f(fmaf(fmaxf(acc[r] + bias, 0.f), sc, sh));
            }
        }
        __syncthreads();
        // ---- 2x2 max-pool + flatten (h, w, c): A4[clip][flat] in R2, row stride 544 B
        for (int i = tid; i < kF16Clips * n4 * 128; i += 256) {
            const int ch = i & 127, q = i >> 7, c = q / n4, w = q - c * n4, ph = w / k.W4, pw = w - ph * k.W4;
            const unsigned char *z = R1 + (c * P3 + 2 * ph * k.W3 + 2 * pw) * 288 + 2 * ch;
            const float m = fmaxf(fmaxf((float)*reinterpret_cast<const _Float16 *>(z), (float)*reinterpret_cast<const _Float16 *>(z + 288)),
                                  fmaxf((float)*reinterpret_cast<const _Float16 *>(z + k.W3 * 288),
                                        (float)*reinterpret_cast<const _Float16 *>(z + (k.W3 + 1) * 288)));
            *reinterpret_cast<_Float16 *>(R2 + c * 544 + 2 * (w * 128 + ch)) = (_Float16)m;
        }
        __syncthreads();
        // ---- Dense(128) + bias + ReLU6 (cnn.py:129): D1[clip][128] in R1, row stride 288 B; the kernel streams from L2
        for (int nt = 2 * wave; nt < 2 * wave + 2; ++nt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const _Float16 *wrow = k.blob + kBlobWd + (16 * nt + li) * kWdRow + 8 * lq;
            for (int ks = 0; ks < flat / 32; ++ks) {
                const f16x8 bfr = *reinterpret_cast<const f16x8 *>(wrow + 32 * ks);
                const f16x8 afr = *reinterpret_cast<const f16x8 *>(R2 + li * 544 + 64 * ks + 16 * lq);
                acc = mfma_f16(afr, bfr, acc);
            }
            const int col = 16 * nt + li;
            const float bias = k.db[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) *reinterpret_cast<_Float16 *>(R1 + (4 * lq + r) * 288 + 2 * col) = (_Float16)relu6f(acc[r] + bias);
        }
        __syncthreads();
        // ---- head: logits[clip][C] (fp32) behind D1 in R1
        float *logits = reinterpret_cast<float *>(R1 + 8192);     // [16][kF16HeadCols]
        if (wave < kF16HeadCols / 16) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f16x8 afr = *reinterpret_cast<const f16x8 *>(R1 + li * 288 + 64 * ks + 16 * lq);
                const f16x8 bfr = *reinterpret_cast<const f16x8 *>(Wl + kBlobW2 + (16 * wave + li) * kW2Row + 32 * ks + 8 * lq);
                acc = mfma_f16(afr, bfr, acc);
            }
            const int col = 16 * wave + li;
            const float bias = col < k.C ? k.hb[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) logits[(4 * lq + r) * kF16HeadCols + col] = acc[r] + bias;
        }
        __syncthreads();
        // ---- softmax: 16 lanes per clip
        {
            const int c = tid >> 4, j = tid & 15, b = b0 + c;
            float v[kF16HeadCols / 16], mx = -INFINITY;
            int am = 0;
#pragma unroll
            for (int u = 0; u < kF16HeadCols / 16; ++u) {
                const int col = j + 16 * u;
                v[u] = col < k.C ? logits[c * kF16HeadCols + col] : -INFINITY;
                if (v[u] > mx) { mx = v[u]; am = col; }
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) {                     // first maximum wins (np.argmax)
                const float om = __shfl_xor(mx, o, 16);
                const int oa = __shfl_xor(am, o, 16);
                if (om > mx || (om == mx && oa < am)) { mx = om; am = oa; }
            }
            float e[kF16HeadCols / 16], sum = 0.f;
#pragma unroll
            for (int u = 0; u < kF16HeadCols / 16; ++u) { e[u] = j + 16 * u < k.C ? __expf(v[u] - mx) : 0.f; sum += e[u]; }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 16);
            if (b < B) {
                if (probs) {
#pragma unroll
                    for (int u = 0; u < kF16HeadCols / 16; ++u)
                        if (j + 16 * u < k.C) probs[(long)b * k.C + j + 16 * u] = e[u] / sum;
                }
                if (argmax && j == 0) argmax[b] = am;
            }
        }
    }
}

}  // namespace kws
