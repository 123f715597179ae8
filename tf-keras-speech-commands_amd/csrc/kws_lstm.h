// csrc/kws_lstm.h -- LSTM(48, activation='tanh', dropout=0.2) forward and BPTT (classifier/models/rnn.py:46-79).
//
// Keras LSTM semantics: kernel (F, 4u), recurrent_kernel (u, 4u), bias (4u); gate order i, f, c, o; recurrent_activation
// sigmoid; one input dropout mask per sample shared by all timesteps (the four gates share it, as Keras' fused
// implementation does):
//     a = x_t W + h U + b;  i = s(a_i);  f = s(a_f);  g = tanh(a_c);  o = s(a_o);  c' = f c + i g;  h' = o tanh(c')
//
// Same structure as the GRU kernels (kws_gru.h), block = 16 clips:
//   forward   twelve waves, wave w owns ONE 16-column tile of the 192 gate columns (gate w / 3, hidden units 16 (w % 3) ..): 12 + KX MFMAs
//             per step in three accumulation chains, W / U fragments in registers for the whole sequence; the pre-activations go through
//             an LDS tile and after a barrier each of the 768 threads applies the gate arithmetic to ONE of the 16 x 48 outputs -- its cell
//             state stays in a register for the whole sequence; h ping-pongs through LDS.
//   BPTT      nine waves: waves 0-2 carry the recurrence (gate gradients of their 16 units -> LDS, dh_prev = da U^T: 48 MFMAs, dc in
//             registers), waves 3-5 accumulate dU and waves 6-8 dW of the SAME step from the LDS tiles between the same two barriers.
// (Round 2 ran both as three waves owning all four gates of their units: 68 / 128 dependent MFMAs per step per wave, 0.080 / 0.160 ms at
// B = 2048; this form 0.054 / 0.100 ms -- DESIGN.md section 5.)
#pragma once
#include "kws_gru.h"

namespace kws {

constexpr int kLstmN = 4 * kGruU;          // 192 gate columns
constexpr int kLstmSave = 7;               // saved per (clip, step): h_prev, c_prev, i, f, g, o, tanh(c')

constexpr int kLstmFwdThreads = 768, kLstmBwdThreads = 576;

template <int KX, bool SAVE>
__global__ __launch_bounds__(kLstmFwdThreads) void lstm_fwd_kernel(const float *__restrict__ feat, const float *__restrict__ Wk,
                                                                    const float *__restrict__ Uk, const float *__restrict__ bias,
                                                                    float *__restrict__ h_out, float *__restrict__ saved, int B, int T,
                                                                    int F, float drop_rate, uint32_t slo, uint32_t shi, float *__restrict__ zero_buf = nullptr, long zero_n = 0)
{
    // training: the gradient buffer of the backward pass that follows is cleared here (zero_n floats over the whole grid) instead of by a
    // memset node on the step's chain
    if (zero_buf)
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < zero_n; i += (long)gridDim.x * blockDim.x) zero_buf[i] = 0.f;
    extern __shared__ float gsm[];
    const int XS = gru_xstride(T, F);
    float *xs = gsm;                       // [16][XS]
    float *hs = gsm + 16 * XS;             // [2][16][kGruHS]
    float *P = hs + 2 * 16 * kGruHS;       // [4][16][kGruPS]: pre-activations of i, f, c, o (bias included)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
    const int gate = wave / 3, b0 = blockIdx.x * 16, u = 16 * (wave % 3) + li, col = gate * kGruU + u;

    float wg[KX], ug[12];
#pragma unroll
    for (int j = 0; j < KX; ++j) {
        const int k = 4 * j + lq;
        wg[j] = k < F ? Wk[k * kLstmN + col] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 12; ++j) ug[j] = Uk[(4 * j + lq) * kLstmN + col];
    const float bq = bias[col];

    gru_stage_x(feat, xs, b0, B, T, F, XS, drop_rate, slo, shi);
    for (int i = threadIdx.x; i < 2 * 16 * kGruHS; i += kLstmFwdThreads) hs[i] = 0.f;
    __syncthreads();

    const int ec = threadIdx.x / kGruU, ek = threadIdx.x - ec * kGruU;          // this thread's output (clip, unit) in the gate phase
    float cst = 0.f;                                                            // its cell state
    int cur = 0;
    for (int t = 0; t < T; ++t) {
        const float *hc = hs + cur * 16 * kGruHS;
        {
            float xa[KX], ha[12];
#pragma unroll
            for (int j = 0; j < KX; ++j) {
                const int k = 4 * j + lq;
                xa[j] = k < F ? xs[li * XS + t * F + k] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 12; ++j) ha[j] = hc[li * kGruHS + 4 * j + lq];
            f32x4 ax = {0.f, 0.f, 0.f, 0.f}, ah0 = ax, ah1 = ax;               // three short dependent chains instead of one of 12 + KX
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                if (j < KX) ax = mfma16(xa[j], wg[j], ax);
                ah0 = mfma16(ha[j], ug[j], ah0);
                ah1 = mfma16(ha[6 + j], ug[6 + j], ah1);
            }
#pragma unroll
            for (int j = 6; j < KX; ++j) ax = mfma16(xa[j], wg[j], ax);
#pragma unroll
            for (int r = 0; r < 4; ++r) P[(gate * 16 + 4 * lq + r) * kGruPS + u] = (ax[r] + (ah0[r] + ah1[r])) + bq;
        }
        __syncthreads();
        float *hn = hs + (cur ^ 1) * 16 * kGruHS;
        {
            const float ig = sigmoidf_(P[ec * kGruPS + ek]), fg = sigmoidf_(P[(16 + ec) * kGruPS + ek]);
            const float gg = tanh_fast_(P[(32 + ec) * kGruPS + ek]), og = sigmoidf_(P[(48 + ec) * kGruPS + ek]);
            const float cp = cst, cn = fg * cp + ig * gg, tc = tanh_fast_(cn);
            const float hp = hc[ec * kGruHS + ek];
            cst = cn;
            hn[ec * kGruHS + ek] = og * tc;
            if (SAVE && b0 + ec < B) {
                float *sv = saved + (((long)(b0 + ec) * T + t) * kLstmSave) * kGruU + ek;
                sv[0] = hp; sv[kGruU] = cp; sv[2 * kGruU] = ig; sv[3 * kGruU] = fg; sv[4 * kGruU] = gg; sv[5 * kGruU] = og;
                sv[6 * kGruU] = tc;
            }
        }
        cur ^= 1;
        __syncthreads();
    }
    if (b0 + ec < B) h_out[(long)(b0 + ec) * kGruU + ek] = hs[cur * 16 * kGruHS + ec * kGruHS + ek];
}

// BPTT.  Per step (t = T-1 .. 0) and 16-clip tile, with dh and dc carried backwards:
//   do = dh tanh(c');  dc += dh o (1 - tanh(c')^2)
//   da = [dc g i(1-i) | dc c_prev f(1-f) | dc i (1-g^2) | do o(1-o)]          (192 columns, tile G)
//   dW += x_t^T da;  dU += h_prev^T da;  db += sum da;  dh_prev = da U^T;  dc_prev = dc f
template <int KX>
__global__ __launch_bounds__(kLstmBwdThreads) void lstm_bwd_kernel(const float *__restrict__ feat, const float *__restrict__ Uk,
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
    const bool rec = wave < 3;                         // recurrence waves; waves 3-5 accumulate dU, waves 6-8 dW
    const bool do_u = wave >= 3 && wave < 6;
    const int w3 = wave % 3;
    const int b0 = blockIdx.x * 16, u = 16 * w3 + li;

    gru_stage_x(feat, xs, b0, B, T, F, XS, drop_rate, slo, shi);
    for (int i = threadIdx.x; i < 16 * kGruU; i += kLstmBwdThreads) {
        const int c = i / kGruU, k = i % kGruU;
        dhs[c * kGruHS + k] = (b0 + c < B) ? dh_last[(long)(b0 + c) * kGruU + k] : 0.f;
    }
    __syncthreads();

    if (rec) {
        float ut[48];                                  // B fragments of U^T: B[k = n][col = u] = U[u][n], n over the 192 columns
#pragma unroll
        for (int j = 0; j < 48; ++j) ut[j] = Uk[u * kLstmN + 4 * j + lq];
        float sbq[4] = {0.f, 0.f, 0.f, 0.f};           // bias partials of unit u over this lane's four clips, per gate
        float dcs[4] = {0.f, 0.f, 0.f, 0.f};           // dL/dc of (clip 4 lq + r, unit u), carried backwards
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
                const float gi = dc * gg * ig * (1.f - ig), gf = dc * cp * fg * (1.f - fg), gc = dc * ig * (1.f - gg * gg), go = dh * tc * og * (1.f - og);
                G[c * kGruGS + u] = gi;
                G[c * kGruGS + kGruU + u] = gf;
                G[c * kGruGS + 2 * kGruU + u] = gc;
                G[c * kGruGS + 3 * kGruU + u] = go;
                Hp[c * kGruHS + u] = hp;
                dcs[r] = dc * fg;
                sbq[0] += gi; sbq[1] += gf; sbq[2] += gc; sbq[3] += go;       // bias gradients: column sums of G (clips past B contribute zeros)
            }
            __syncthreads();                           // G, Hp of step t complete (the other waves start their products)
            // dh_prev = da U^T: the operands in one batch of LDS reads, three accumulation chains
            float ga[48];
#pragma unroll
            for (int j = 0; j < 48; ++j) ga[j] = G[li * kGruGS + 4 * j + lq];
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                acc0 = mfma16(ga[3 * j], ut[3 * j], acc0);
                acc1 = mfma16(ga[3 * j + 1], ut[3 * j + 1], acc1);
                acc2 = mfma16(ga[3 * j + 2], ut[3 * j + 2], acc2);
            }
            float *dn = dhs + (cur ^ 1) * 16 * kGruHS;
#pragma unroll
            for (int r = 0; r < 4; ++r) dn[(4 * lq + r) * kGruHS + u] = (acc0[r] + acc1[r]) + acc2[r];
            cur ^= 1;
            __syncthreads();                           // dh of step t-1 complete; G / Hp may be overwritten
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {                  // lanes with equal li hold the same unit: reduce over lq
            sbq[q] += __shfl_xor(sbq[q], 16, 64);
            sbq[q] += __shfl_xor(sbq[q], 32, 64);
            if (lq == 0) atomicAdd(db + q * kGruU + u, sbq[q]);
        }
    } else {
        f32x4 accU[12], accW[WT];
#pragma unroll
        for (int i = 0; i < 12; ++i) accU[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < WT; ++i) accW[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int t = T - 1; t >= 0; --t) {
            __syncthreads();                           // G, Hp of step t complete
            if (do_u) {
                float hv[4], gu[12][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) hv[j] = Hp[(4 * j + lq) * kGruHS + u];
#pragma unroll
                for (int nt = 0; nt < 12; ++nt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) gu[nt][j] = G[(4 * j + lq) * kGruGS + 16 * nt + li];
                // dU[16w + ..][:] += h_prev^T da (reduction index = clip); j outside: consecutive MFMAs write different accumulators
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int nt = 0; nt < 12; ++nt) accU[nt] = mfma16(hv[j], gu[nt][j], accU[nt]);
            } else {
                float gw[WT][4], xa[WT][4];
#pragma unroll
                for (int i = 0; i < WT; ++i) {
                    const int tile = w3 + 3 * i, tl = tile < MTW * 12 ? tile : 0;
                    const int mt = tl / 12, nt = tl % 12, f = 16 * mt + li;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        gw[i][j] = G[(4 * j + lq) * kGruGS + 16 * nt + li];
                        xa[i][j] = f < F ? xs[(4 * j + lq) * XS + t * F + f] : 0.f;
                    }
                }
                // dW tiles (feature rows x 192 columns), dealt round-robin to the three waves; tiles past the end multiply tile-0 operands
                // into an accumulator that is never stored
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < WT; ++i) accW[i] = mfma16(xa[i][j], gw[i][j], accW[i]);
            }
            __syncthreads();                           // the recurrence waves may overwrite G / Hp
        }
        // D layout: row = 4*lq + r, col = li
        if (do_u) {
#pragma unroll
            for (int nt = 0; nt < 12; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(dU + (16 * w3 + 4 * lq + r) * kLstmN + 16 * nt + li, accU[nt][r]);
        }
#pragma unroll
        for (int i = 0; i < WT; ++i) {
            const int tile = w3 + 3 * i;
            if (!do_u && tile < MTW * 12) {
                const int mt = tile / 12, nt = tile % 12;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = 16 * mt + 4 * lq + r;
                    if (f < F) atomicAdd(dW + f * kLstmN + 16 * nt + li, accW[i][r]);
                }
            }
        }
    }
}

}  // namespace kws
