// csrc/kws_lstm.h -- LSTM(48, activation='tanh', dropout=0.2) forward and BPTT (classifier/models/rnn.py:46-79).
//
// Keras LSTM semantics: kernel (F, 4u), recurrent_kernel (u, 4u), bias (4u); gate order i, f, c, o; recurrent_activation
// sigmoid; one input dropout mask per sample shared by all timesteps (the four gates share it, as Keras' fused
// implementation does):
//     a = x_t W + h U + b;  i = s(a_i);  f = s(a_f);  g = tanh(a_c);  o = s(a_o);  c' = f c + i g;  h' = o tanh(c')
//
// Same mapping as the GRU (kws_gru.h): block = 16 clips x 3 waves, wave w owns hidden units 16w..16w+15 of all four gates,
// its W / U fragments stay in registers for the whole sequence, h ping-pongs through LDS (one barrier per step) and the
// cell state c lives in registers (the MFMA D layout gives every lane the same (clip, unit) elements at every step).
#pragma once
#include "kws_gru.h"

namespace kws {

constexpr int kLstmN = 4 * kGruU;          // 192 gate columns
constexpr int kLstmSave = 7;               // saved per (clip, step): h_prev, c_prev, i, f, g, o, tanh(c')

template <int KX, bool SAVE>
__global__ __launch_bounds__(192) void lstm_fwd_kernel(const float *__restrict__ feat, const float *__restrict__ Wk,
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

    float wx[4][KX], uh[4][12];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int j = 0; j < KX; ++j) {
            const int k = 4 * j + lq;
            wx[q][j] = k < F ? Wk[k * kLstmN + q * kGruU + u] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 12; ++j) uh[q][j] = Uk[(4 * j + lq) * kLstmN + q * kGruU + u];
    }
    float bq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bq[q] = bias[q * kGruU + u];

    gru_stage_x(feat, xs, b0, B, T, F, XS, drop_rate, slo, shi);
    for (int i = threadIdx.x; i < 2 * 16 * kGruHS; i += 192) hs[i] = 0.f;
    __syncthreads();

    float cst[4] = {0.f, 0.f, 0.f, 0.f};   // cell state of (clip 4 lq + r, unit u)
    int cur = 0;
    for (int t = 0; t < T; ++t) {
        f32x4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float *hc = hs + cur * 16 * kGruHS;
#pragma unroll
        for (int j = 0; j < KX; ++j) {
            const int k = 4 * j + lq;
            const float a = k < F ? xs[li * XS + t * F + k] : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = mfma16(a, wx[q][j], acc[q]);
        }
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const float a = hc[li * kGruHS + 4 * j + lq];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = mfma16(a, uh[q][j], acc[q]);
        }
        float *hn = hs + (cur ^ 1) * 16 * kGruHS;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 4 * lq + r;
            const float ig = sigmoidf_(acc[0][r] + bq[0]), fg = sigmoidf_(acc[1][r] + bq[1]);
            const float gg = tanh_fast_(acc[2][r] + bq[2]), og = sigmoidf_(acc[3][r] + bq[3]);
            const float cp = cst[r], cn = fg * cp + ig * gg, tc = tanh_fast_(cn);
            const float hp = hc[c * kGruHS + u];
            cst[r] = cn;
            hn[c * kGruHS + u] = og * tc;
            if (SAVE && b0 + c < B) {
                float *sv = saved + (((long)(b0 + c) * T + t) * kLstmSave) * kGruU + u;
                sv[0] = hp; sv[kGruU] = cp; sv[2 * kGruU] = ig; sv[3 * kGruU] = fg; sv[4 * kGruU] = gg; sv[5 * kGruU] = og;
                sv[6 * kGruU] = tc;
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

// BPTT.  Per step (t = T-1 .. 0) and 16-clip tile, with dh and dc carried backwards:
//   do = dh tanh(c');  dc += dh o (1 - tanh(c')^2)
//   da = [dc g i(1-i) | dc c_prev f(1-f) | dc i (1-g^2) | do o(1-o)]          (192 columns, tile G)
//   dW += x_t^T da;  dU += h_prev^T da;  db += sum da;  dh_prev = da U^T;  dc_prev = dc f
template <int KX>
__global__ __launch_bounds__(192) void lstm_bwd_kernel(const float *__restrict__ feat, const float *__restrict__ Uk,
                                                        const float *__restrict__ saved, const float *__restrict__ dh_last,
                                                        float *__restrict__ dW, float *__restrict__ dU, float *__restrict__ db,
                                                        int B, int T, int F, float drop_rate, uint32_t slo, uint32_t shi)
{
    constexpr int MTW = (KX * 4 + 15) / 16;            // 16-row tiles covering the F input features
    constexpr int WT = (MTW * 12 + 2) / 3;             // dW tiles per wave
    extern __shared__ float gsm[];
    const int XS = gru_xstride(T, F);
    float *xs = gsm;                                   // [16][XS]
    float *G = gsm + 16 * XS;                          // [16][kGruGS]: da_i | da_f | da_c | da_o
    float *Hp = G + 16 * kGruGS;                       // [16][kGruHS] h_prev of the step
    float *dhs = Hp + 16 * kGruHS;                     // [2][16][kGruHS]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
    const int b0 = blockIdx.x * 16, u = 16 * wave + li;

    float ut[48];                                      // B fragments of U^T: B[k = n][col = u] = U[u][n], n over the 192 columns
#pragma unroll
    for (int j = 0; j < 48; ++j) ut[j] = Uk[u * kLstmN + 4 * j + lq];
    f32x4 accU[12], accW[WT];
#pragma unroll
    for (int i = 0; i < 12; ++i) accU[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < WT; ++i) accW[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float sb = 0.f;                                    // bias partial of column threadIdx.x (192 threads = 192 columns)

    gru_stage_x(feat, xs, b0, B, T, F, XS, drop_rate, slo, shi);
    for (int i = threadIdx.x; i < 16 * kGruU; i += 192) {
        const int c = i / kGruU, k = i % kGruU;
        dhs[c * kGruHS + k] = (b0 + c < B) ? dh_last[(long)(b0 + c) * kGruU + k] : 0.f;
    }
    __syncthreads();

    float dcs[4] = {0.f, 0.f, 0.f, 0.f};               // dL/dc of (clip 4 lq + r, unit u), carried backwards
    // the step's saved forward values come from global memory: requested one step AHEAD by unconditional loads on clamped (clip, step)
    // addresses, masked afterwards (kws_gru.h: gru_bwd_kernel has the measurements)
    float svn[4][kLstmSave];
    const float *svbase[4];
    float svmask[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = 4 * lq + r;
        const bool in = b0 + c < B;
        svbase[r] = saved + ((long)(in ? b0 + c : b0) * T * kLstmSave) * kGruU + u;
        svmask[r] = in ? 1.f : 0.f;
    }
    auto fetch_saved = [&](int t) {
        const int tcl = t >= 0 ? t : 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float *sv = svbase[r] + (long)tcl * kLstmSave * kGruU;
#pragma unroll
            for (int q = 0; q < kLstmSave; ++q) svn[r][q] = sv[q * kGruU] * svmask[r];
        }
    };
    fetch_saved(T - 1);
    int cur = 0;
    for (int t = T - 1; t >= 0; --t) {
        const float *dcur = dhs + cur * 16 * kGruHS;
        float svc[4][kLstmSave];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < kLstmSave; ++q) svc[r][q] = svn[r][q];
        fetch_saved(t - 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 4 * lq + r;
            const float hp = svc[r][0], cp = svc[r][1], ig = svc[r][2], fg = svc[r][3], gg = svc[r][4], og = svc[r][5], tc = svc[r][6];
            const float dh = dcur[c * kGruHS + u];
            const float dc = dcs[r] + dh * og * (1.f - tc * tc);
            G[c * kGruGS + u] = dc * gg * ig * (1.f - ig);
            G[c * kGruGS + kGruU + u] = dc * cp * fg * (1.f - fg);
            G[c * kGruGS + 2 * kGruU + u] = dc * ig * (1.f - gg * gg);
            G[c * kGruGS + 3 * kGruU + u] = dh * tc * og * (1.f - og);
            Hp[c * kGruHS + u] = hp;
            dcs[r] = dc * fg;
        }
        __syncthreads();
        // dh_prev = da U^T
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 48; ++j) acc = mfma16(G[li * kGruGS + 4 * j + lq], ut[j], acc);
        float *dn = dhs + (cur ^ 1) * 16 * kGruHS;
#pragma unroll
        for (int r = 0; r < 4; ++r) dn[(4 * lq + r) * kGruHS + u] = acc[r];
        // dU[16w + ..][:] += h_prev^T da   (reduction index = clip)
#pragma unroll
        for (int nt = 0; nt < 12; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                accU[nt] = mfma16(Hp[(4 * j + lq) * kGruHS + u], G[(4 * j + lq) * kGruGS + 16 * nt + li], accU[nt]);
        // dW tiles (feature rows x 192 columns), dealt round-robin to the waves
#pragma unroll
        for (int i = 0; i < WT; ++i) {
            const int tile = wave + 3 * i;
            if (tile < MTW * 12) {
                const int mt = tile / 12, nt = tile % 12, f = 16 * mt + li;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = f < F ? xs[(4 * j + lq) * XS + t * F + f] : 0.f;
                    accW[i] = mfma16(a, G[(4 * j + lq) * kGruGS + 16 * nt + li], accW[i]);
                }
            }
        }
        for (int c = 0; c < 16; ++c) sb += G[c * kGruGS + threadIdx.x];
        cur ^= 1;
        __syncthreads();
    }
    // D layout: row = 4*lq + r, col = li
#pragma unroll
    for (int nt = 0; nt < 12; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(dU + (16 * wave + 4 * lq + r) * kLstmN + 16 * nt + li, accU[nt][r]);
#pragma unroll
    for (int i = 0; i < WT; ++i) {
        const int tile = wave + 3 * i;
        if (tile < MTW * 12) {
            const int mt = tile / 12, nt = tile % 12;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 16 * mt + 4 * lq + r;
                if (f < F) atomicAdd(dW + f * kLstmN + 16 * nt + li, accW[i][r]);
            }
        }
    }
    atomicAdd(db + threadIdx.x, sb);
}

}  // namespace kws
