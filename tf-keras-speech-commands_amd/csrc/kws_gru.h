// csrc/kws_gru.h -- GRU(48, activation='linear', dropout=0.2) forward and BPTT (classifier/models/rnn.py:34-35).
//
// Keras v2 GRU semantics (reset_after=True, bias (2,144), gate order z,r,h, recurrent_activation sigmoid, one input
// dropout mask per sample shared by all timesteps):
//     mx = x_t W + b0;  mh = h U + b1;  z = s(mx_z + mh_z);  r = s(mx_r + mh_r)
//     hh = mx_h + r * mh_h  (activation='linear': no tanh);   h' = z h + (1 - z) hh
//
// Mapping: block = 16 clips x 3 waves; wave w owns hidden units 16w..16w+15 for all three gates.  All of the wave's
// weight fragments (columns of W and U for its units) live in registers for the whole sequence, the hidden state is
// ping-ponged through a 2 x 16 x 48 LDS tile (one barrier per step), the clip tile's features are staged in LDS once.
// The 30-step recurrence is latency-bound: per step and wave 3*KX + 36 dependent-free v_mfma_f32_16x16x4_f32.
#pragma once
#include "kws_device.h"

namespace kws {

constexpr int kGruU = 48;                 // recurrent_units (classifier/model.py:27)
constexpr int kGruN = 3 * kGruU;          // 144 gate columns
constexpr int kGruHS = 50;                // LDS row stride of a 48-wide state tile (== 18 mod 32: conflict-free A reads)
constexpr int kGruGS = 210;               // row stride of the 192-wide gradient tile (== 18 mod 32)
constexpr int kGruSave = 5;               // saved per (clip, step): h_prev, z, r, hh, mh_h

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__host__ __device__ inline int gru_xstride(int T, int F)
{
    int s = T * F;
    while (s % 32 != 18) ++s;
    return s;
}

// stage the 16-clip feature tile (with the per-sample input dropout mask) into LDS
__device__ __forceinline__ void gru_stage_x(const float *__restrict__ feat, float *xs, int b0, int B, int T, int F, int XS,
                                            float drop_rate, uint32_t slo, uint32_t shi)
{
    const int TF = T * F;
    for (int i = threadIdx.x; i < 16 * TF; i += blockDim.x) {
        const int c = i / TF, e = i % TF, b = b0 + c;
        float v = 0.f;
        if (b < B) {
            v = feat[(long)b * TF + e];
            if (drop_rate > 0.f)
                v = dropout_keep(slo, shi, (uint32_t)(b * F + e % F), drop_rate) ? v / (1.f - drop_rate) : 0.f;
        }
        xs[c * XS + e] = v;
    }
}

template <int KX, bool SAVE>
__global__ __launch_bounds__(192) void gru_fwd_kernel(const float *__restrict__ feat, const float *__restrict__ Wk,
                                                       const float *__restrict__ Uk, const float *__restrict__ bias,
                                                       float *__restrict__ h_out, float *__restrict__ saved, int B, int T,
                                                       int F, float drop_rate, uint32_t slo, uint32_t shi)
{
    extern __shared__ float gsm[];
    const int XS = gru_xstride(T, F);
    float *xs = gsm;                       // [16][XS]
    float *hs = gsm + 16 * XS;             // [2][16][kGruHS]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
    const int b0 = blockIdx.x * 16, u = 16 * wave + li;

    float wz[KX], wr[KX], wh[KX], uz[12], ur[12], uh[12];
#pragma unroll
    for (int j = 0; j < KX; ++j) {
        const int k = 4 * j + lq;
        wz[j] = k < F ? Wk[k * kGruN + u] : 0.f;
        wr[j] = k < F ? Wk[k * kGruN + kGruU + u] : 0.f;
        wh[j] = k < F ? Wk[k * kGruN + 2 * kGruU + u] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const int k = 4 * j + lq;
        uz[j] = Uk[k * kGruN + u];
        ur[j] = Uk[k * kGruN + kGruU + u];
        uh[j] = Uk[k * kGruN + 2 * kGruU + u];
    }
    const float bz = bias[u] + bias[kGruN + u], br = bias[kGruU + u] + bias[kGruN + kGruU + u];
    const float bxh = bias[2 * kGruU + u], bhh = bias[kGruN + 2 * kGruU + u];

    gru_stage_x(feat, xs, b0, B, T, F, XS, drop_rate, slo, shi);
    for (int i = threadIdx.x; i < 2 * 16 * kGruHS; i += 192) hs[i] = 0.f;
    __syncthreads();

    int cur = 0;
    for (int t = 0; t < T; ++t) {
        f32x4 az = {0.f, 0.f, 0.f, 0.f}, ar = az, axh = az, ahh = az;
        const float *hc = hs + cur * 16 * kGruHS;
#pragma unroll
        for (int j = 0; j < KX; ++j) {
            const int k = 4 * j + lq;
            const float a = k < F ? xs[li * XS + t * F + k] : 0.f;
            az = mfma16(a, wz[j], az);
            ar = mfma16(a, wr[j], ar);
            axh = mfma16(a, wh[j], axh);
        }
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const float a = hc[li * kGruHS + 4 * j + lq];
            az = mfma16(a, uz[j], az);
            ar = mfma16(a, ur[j], ar);
            ahh = mfma16(a, uh[j], ahh);
        }
        float *hn = hs + (cur ^ 1) * 16 * kGruHS;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 4 * lq + r;
            const float z = sigmoidf_(az[r] + bz), rg = sigmoidf_(ar[r] + br);
            const float mhh = ahh[r] + bhh;
            const float hh = axh[r] + bxh + rg * mhh;
            const float hp = hc[c * kGruHS + u];
            hn[c * kGruHS + u] = z * hp + (1.f - z) * hh;
            if (SAVE && b0 + c < B) {
                float *sv = saved + (((long)(b0 + c) * T + t) * kGruSave) * kGruU + u;
                sv[0] = hp; sv[kGruU] = z; sv[2 * kGruU] = rg; sv[3 * kGruU] = hh; sv[4 * kGruU] = mhh;
            }
        }
        cur ^= 1;
        __syncthreads();
    }
    const float *hf = hs + cur * 16 * kGruHS;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = 4 * lq + r;
        if (b0 + c < B) h_out[(long)(b0 + c) * kGruU + u] = hf[c * kGruHS + u];
    }
}

// BPTT.  Per step (t = T-1 .. 0) and 16-clip tile:
//   dz = dh (h_prev - hh); dhh = dh (1 - z); dr = dhh mh_h; dmh_h = dhh r; dz' = dz z(1-z); dr' = dr r(1-r)
//   dmx = [dz' dr' dhh]; dmh = [dz' dr' dmh_h];  dh_prev = dh z + dmh U^T
//   dW += x_t^T dmx; dU += h_prev^T dmh; db0 += sum dmx; db1 += sum dmh      (block partials -> float atomics)
template <int KX>
__global__ __launch_bounds__(192) void gru_bwd_kernel(const float *__restrict__ feat, const float *__restrict__ Uk,
                                                       const float *__restrict__ saved, const float *__restrict__ dh_last,
                                                       float *__restrict__ dW, float *__restrict__ dU, float *__restrict__ db,
                                                       int B, int T, int F, float drop_rate, uint32_t slo, uint32_t shi)
{
    constexpr int MTW = (KX * 4 + 15) / 16;            // 16-row tiles covering the F input features
    constexpr int WT = (MTW * 9 + 2) / 3;              // dW tiles per wave
    extern __shared__ float gsm[];
    const int XS = gru_xstride(T, F);
    float *xs = gsm;                                   // [16][XS]
    float *G = gsm + 16 * XS;                          // [16][kGruGS]: dz' | dr' | dhh | dmh_h
    float *Hp = G + 16 * kGruGS;                       // [16][kGruHS] h_prev of the step
    float *dhs = Hp + 16 * kGruHS;                     // [2][16][kGruHS]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
    const int b0 = blockIdx.x * 16, u = 16 * wave + li;

    float ut[36];                                      // B fragments of U^T restricted to dmh's 144 columns
#pragma unroll
    for (int j = 0; j < 36; ++j) ut[j] = Uk[u * kGruN + 4 * j + lq];
    f32x4 accU[9], accW[WT];
#pragma unroll
    for (int i = 0; i < 9; ++i) accU[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < WT; ++i) accW[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float sb0 = 0.f, sb1 = 0.f;                        // bias partials of column threadIdx.x (< 144)

    gru_stage_x(feat, xs, b0, B, T, F, XS, drop_rate, slo, shi);
    for (int i = threadIdx.x; i < 16 * kGruU; i += 192) {
        const int c = i / kGruU, k = i % kGruU;
        dhs[c * kGruHS + k] = (b0 + c < B) ? dh_last[(long)(b0 + c) * kGruU + k] : 0.f;
    }
    __syncthreads();

    int cur = 0;
    for (int t = T - 1; t >= 0; --t) {
        const float *dc = dhs + cur * 16 * kGruHS;
        float dhz[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 4 * lq + r;
            float hp = 0.f, z = 0.f, rg = 0.f, hh = 0.f, mhh = 0.f;
            if (b0 + c < B) {
                const float *sv = saved + (((long)(b0 + c) * T + t) * kGruSave) * kGruU + u;
                hp = sv[0]; z = sv[kGruU]; rg = sv[2 * kGruU]; hh = sv[3 * kGruU]; mhh = sv[4 * kGruU];
            }
            const float dh = dc[c * kGruHS + u];
            const float dhh = dh * (1.f - z);
            G[c * kGruGS + u] = dh * (hp - hh) * z * (1.f - z);
            G[c * kGruGS + kGruU + u] = dhh * mhh * rg * (1.f - rg);
            G[c * kGruGS + 2 * kGruU + u] = dhh;
            G[c * kGruGS + 3 * kGruU + u] = dhh * rg;
            Hp[c * kGruHS + u] = hp;
            dhz[r] = dh * z;
        }
        __syncthreads();
        // dh_prev = dh z + dmh U^T   (dmh = columns [0,96) and [144,192) of G)
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 36; ++j) {
            const int n = 4 * j + lq;
            acc = mfma16(G[li * kGruGS + (n < 96 ? n : n + kGruU)], ut[j], acc);
        }
        float *dn = dhs + (cur ^ 1) * 16 * kGruHS;
#pragma unroll
        for (int r = 0; r < 4; ++r) dn[(4 * lq + r) * kGruHS + u] = acc[r] + dhz[r];
        // dU[16w + ..][:] += h_prev^T dmh   (reduction index = clip)
#pragma unroll
        for (int nt = 0; nt < 9; ++nt) {
            const int col = 16 * nt + li, gcol = col < 96 ? col : col + kGruU;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                accU[nt] = mfma16(Hp[(4 * j + lq) * kGruHS + u], G[(4 * j + lq) * kGruGS + gcol], accU[nt]);
        }
        // dW tiles (feature rows x 144 dmx columns = the first 144 columns of G), dealt round-robin to the waves
#pragma unroll
        for (int i = 0; i < WT; ++i) {
            const int tile = wave + 3 * i;
            if (tile < MTW * 9) {
                const int mt = tile / 9, nt = tile % 9, f = 16 * mt + li;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = f < F ? xs[(4 * j + lq) * XS + t * F + f] : 0.f;
                    accW[i] = mfma16(a, G[(4 * j + lq) * kGruGS + 16 * nt + li], accW[i]);
                }
            }
        }
        if (threadIdx.x < kGruN) {
            const int n = threadIdx.x, gcol = n < 96 ? n : n + kGruU;
            for (int c = 0; c < 16; ++c) { sb0 += G[c * kGruGS + n]; sb1 += G[c * kGruGS + gcol]; }
        }
        cur ^= 1;
        __syncthreads();
    }
    // D layout: row = 4*lq + r, col = li
#pragma unroll
    for (int nt = 0; nt < 9; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(dU + (16 * wave + 4 * lq + r) * kGruN + 16 * nt + li, accU[nt][r]);
#pragma unroll
    for (int i = 0; i < WT; ++i) {
        const int tile = wave + 3 * i;
        if (tile < MTW * 9) {
            const int mt = tile / 9, nt = tile % 9;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 16 * mt + 4 * lq + r;
                if (f < F) atomicAdd(dW + f * kGruN + 16 * nt + li, accW[i][r]);
            }
        }
    }
    if (threadIdx.x < kGruN) {
        atomicAdd(db + threadIdx.x, sb0);
        atomicAdd(db + kGruN + threadIdx.x, sb1);
    }
}

inline size_t gru_fwd_smem(int T, int F) { return sizeof(float) * (size_t)(16 * gru_xstride(T, F) + 2 * 16 * kGruHS); }
inline size_t gru_bwd_smem(int T, int F)
{
    return sizeof(float) * (size_t)(16 * gru_xstride(T, F) + 16 * kGruGS + 16 * kGruHS + 2 * 16 * kGruHS);
}

}  // namespace kws
