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

// tanh(x) = (1 - e^-2|x|) / (1 + e^-2|x|) with the sign of x, on the same two hardware instructions (absolute error ~2e-7; the form
// 2 sigmoid(2x) - 1 cancels for small x)
__device__ __forceinline__ float tanh_fast_(float x)
{
    const float e = __builtin_amdgcn_exp2f(-2.88539008177792681f * fabsf(x));
    const float t = (1.f - e) * __builtin_amdgcn_rcpf(1.f + e);
    return copysignf(t, x);
}

constexpr int kGruU = 48;                 // recurrent_units (classifier/model.py:27)
constexpr int kGruN = 3 * kGruU;          // 144 gate columns
constexpr int kGruHS = 50;                // LDS row stride of a 48-wide state tile (== 18 mod 32: conflict-free A reads)
constexpr int kGruGS = 210;               // row stride of the 192-wide gradient tile (== 18 mod 32)
constexpr int kGruSave = 5;               // saved per (clip, step): h_prev, z, r, hh, mh_h

// 1 / (1 + e^-x) on the hardware exp2 / reciprocal (each within ~1 ulp; the library expf + IEEE division cost ~60 instructions per gate value,
// a third of a recurrent step that one wave per SIMD issues alone).  e^-x overflows to inf for x < -88 -> 0, underflows to 0 for x > 88 -> 1.
__device__ __forceinline__ float sigmoidf_(float x)
{
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}

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

// Nine waves: wave w owns ONE 16-column tile of the 144 gate columns (gate w / 3 of hidden units 16 (w % 3) ..), i.e. 12 + KX MFMAs per step
// instead of 36 + 3 KX in a wave that owns all three gates of its units.  The pre-activations go through an LDS tile (z | r | x-part of
// hh | h-part of hh), and after a barrier all 576 threads apply the gate arithmetic to the 16 x 48 outputs.  Two barriers per step, but the
// step's dependent chain is a third as long (65 -> see DESIGN.md for the measured time); the arithmetic per output is unchanged.
constexpr int kGruFwdThreads = 768;     // nine product waves + three more so that the gate arithmetic is one output per thread
constexpr int kGruPS = 50;                // row stride of a pre-activation plane
template <int KX, bool SAVE>
__global__ __launch_bounds__(kGruFwdThreads) void gru_fwd_kernel(const float *__restrict__ feat, const float *__restrict__ Wk,
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
    float *P = hs + 2 * 16 * kGruHS;       // [4][16][kGruPS]: z, r, x-part of hh, h-part of hh (biases included)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
    const bool mm = wave < 9;               // product waves
    const int gate = mm ? wave / 3 : 0, b0 = blockIdx.x * 16, u = 16 * (wave % 3) + li, col = gate * kGruU + u;

    float wg[KX], ug[12];
#pragma unroll
    for (int j = 0; j < KX; ++j) {
        const int k = 4 * j + lq;
        wg[j] = k < F ? Wk[k * kGruN + col] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 12; ++j) ug[j] = Uk[(4 * j + lq) * kGruN + col];
    // z, r: one pre-activation with both biases; hh keeps its x-part (bias b0) and h-part (bias b1) apart: hh = mx_h + r * mh_h
    const float bx = gate < 2 ? bias[col] + bias[kGruN + col] : bias[col], bh = gate < 2 ? 0.f : bias[kGruN + col];

    gru_stage_x(feat, xs, b0, B, T, F, XS, drop_rate, slo, shi);
    for (int i = threadIdx.x; i < 2 * 16 * kGruHS; i += kGruFwdThreads) hs[i] = 0.f;
    __syncthreads();

    int cur = 0;
    for (int t = 0; t < T; ++t) {
        const float *hc = hs + cur * 16 * kGruHS;
        if (mm) {
            float xa[KX], ha[12];
#pragma unroll
            for (int j = 0; j < KX; ++j) {
                const int k = 4 * j + lq;
                xa[j] = k < F ? xs[li * XS + t * F + k] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 12; ++j) ha[j] = hc[li * kGruHS + 4 * j + lq];
            // x-part and two halves of the h-part in separate accumulators: three short dependent chains instead of one of 12 + KX
            f32x4 ax = {0.f, 0.f, 0.f, 0.f}, ah0 = ax, ah1 = ax;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                if (j < KX) ax = mfma16(xa[j], wg[j], ax);
                ah0 = mfma16(ha[j], ug[j], ah0);
                ah1 = mfma16(ha[6 + j], ug[6 + j], ah1);
            }
#pragma unroll
            for (int j = 6; j < KX; ++j) ax = mfma16(xa[j], wg[j], ax);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 4 * lq + r;
                const float hsum = ah0[r] + ah1[r];
                if (gate < 2) P[(gate * 16 + c) * kGruPS + u] = (ax[r] + hsum) + bx;
                else {
                    P[(2 * 16 + c) * kGruPS + u] = ax[r] + bx;
                    P[(3 * 16 + c) * kGruPS + u] = hsum + bh;
                }
            }
        }
        __syncthreads();
        float *hn = hs + (cur ^ 1) * 16 * kGruHS;
        for (int e = threadIdx.x; e < 16 * kGruU; e += kGruFwdThreads) {
            const int c = e / kGruU, k = e - c * kGruU;
            const float z = sigmoidf_(P[c * kGruPS + k]), rg = sigmoidf_(P[(16 + c) * kGruPS + k]);
            const float mhh = P[(48 + c) * kGruPS + k];
            const float hh = P[(32 + c) * kGruPS + k] + rg * mhh;
            const float hp = hc[c * kGruHS + k];
            hn[c * kGruHS + k] = z * hp + (1.f - z) * hh;
            if (SAVE && b0 + c < B) {
                float *sv = saved + (((long)(b0 + c) * T + t) * kGruSave) * kGruU + k;
                sv[0] = hp; sv[kGruU] = z; sv[2 * kGruU] = rg; sv[3 * kGruU] = hh; sv[4 * kGruU] = mhh;
            }
        }
        cur ^= 1;
        __syncthreads();
    }
    const float *hf = hs + cur * 16 * kGruHS;
    for (int e = threadIdx.x; e < 16 * kGruU; e += kGruFwdThreads) {
        const int c = e / kGruU, k = e - c * kGruU;
        if (b0 + c < B) h_out[(long)(b0 + c) * kGruU + k] = hf[c * kGruHS + k];
    }
}

// BPTT.  Per step (t = T-1 .. 0) and 16-clip tile:
//   dz = dh (h_prev - hh); dhh = dh (1 - z); dr = dhh mh_h; dmh_h = dhh r; dz' = dz z(1-z); dr' = dr r(1-r)
//   dmx = [dz' dr' dhh]; dmh = [dz' dr' dmh_h];  dh_prev = dh z + dmh U^T
//   dW += x_t^T dmx; dU += h_prev^T dmh; db0 += sum dmx; db1 += sum dmh      (block partials -> float atomics)
// Nine waves: waves 0-2 carry the recurrence (gate gradients G of their 16 hidden units -> LDS, then dh_prev = dh z + dmh U^T), waves 3-5
// and 6-8 take the two weight-gradient products of the SAME step (dU: 36, dW: 24 of the step's 96 MFMAs, none of which feeds the
// recurrence) from the G / h_prev / x tiles in LDS, between the same two barriers.  The step's critical path is then 36 MFMAs instead of 96
// (three waves doing everything: 109 us at B = 2048; six waves, dU + dW together: 90 us).
constexpr int kGruBwdThreads = 576;
template <int KX>
__global__ __launch_bounds__(kGruBwdThreads) void gru_bwd_kernel(const float *__restrict__ feat, const float *__restrict__ Uk,
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
    const bool rec = wave < 3;                         // recurrence waves; waves 3-5 accumulate dU, waves 6-8 dW
    const bool do_u = wave >= 3 && wave < 6;
    const int w3 = wave % 3;
    const int b0 = blockIdx.x * 16, u = 16 * w3 + li;

    gru_stage_x(feat, xs, b0, B, T, F, XS, drop_rate, slo, shi);
    for (int i = threadIdx.x; i < 16 * kGruU; i += kGruBwdThreads) {
        const int c = i / kGruU, k = i % kGruU;
        dhs[c * kGruHS + k] = (b0 + c < B) ? dh_last[(long)(b0 + c) * kGruU + k] : 0.f;
    }
    __syncthreads();

    if (rec) {
        float ut[36];                                  // B fragments of U^T restricted to dmh's 144 columns
#pragma unroll
        for (int j = 0; j < 36; ++j) ut[j] = Uk[u * kGruN + 4 * j + lq];
        // bias partials of unit u over this lane's four clips: dz', dr', dhh, dmh_h
        float sbz = 0.f, sbr = 0.f, sbh = 0.f, sbm = 0.f;
        // the step's saved forward values (h_prev, z, r, hh, mh_h of this lane's four clips) come from global memory: they are requested
        // one step AHEAD, under the previous step's products, by unconditional loads on clamped (clip, step) addresses, masked afterwards
        float svn[4][kGruSave];
        const float *svbase[4];
        float svmask[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 4 * lq + r;
            const bool in = b0 + c < B;
            svbase[r] = saved + ((long)(in ? b0 + c : b0) * T * kGruSave) * kGruU + u;
            svmask[r] = in ? 1.f : 0.f;
        }
        auto fetch_saved = [&](int t) {
            const int tc = t >= 0 ? t : 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float *sv = svbase[r] + (long)tc * kGruSave * kGruU;
#pragma unroll
                for (int q = 0; q < kGruSave; ++q) svn[r][q] = sv[q * kGruU] * svmask[r];
            }
        };
        fetch_saved(T - 1);
        int cur = 0;
        for (int t = T - 1; t >= 0; --t) {
            const float *dc = dhs + cur * 16 * kGruHS;
            float dhz[4], svc[4][kGruSave];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q = 0; q < kGruSave; ++q) svc[r][q] = svn[r][q];
            fetch_saved(t - 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 4 * lq + r;
                const float hp = svc[r][0], z = svc[r][1], rg = svc[r][2], hh = svc[r][3], mhh = svc[r][4];
                const float dh = dc[c * kGruHS + u];
                const float dhh = dh * (1.f - z);
                const float gz = dh * (hp - hh) * z * (1.f - z), gr = dhh * mhh * rg * (1.f - rg), gm = dhh * rg;
                G[c * kGruGS + u] = gz;
                G[c * kGruGS + kGruU + u] = gr;
                G[c * kGruGS + 2 * kGruU + u] = dhh;
                G[c * kGruGS + 3 * kGruU + u] = gm;
                Hp[c * kGruHS + u] = hp;
                dhz[r] = dh * z;
                sbz += gz; sbr += gr; sbh += dhh; sbm += gm;     // bias gradients: column sums of G (clips past B contribute zeros)
            }
            __syncthreads();                           // G, Hp of step t complete (the other waves start their products)
            // dh_prev = dh z + dmh U^T   (dmh = columns [0,96) and [144,192) of G): operands in one batch, three accumulation chains
            float ga[36];
#pragma unroll
            for (int j = 0; j < 36; ++j) {
                const int n = 4 * j + lq;
                ga[j] = G[li * kGruGS + (n < 96 ? n : n + kGruU)];
            }
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0;
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                acc0 = mfma16(ga[3 * j], ut[3 * j], acc0);
                acc1 = mfma16(ga[3 * j + 1], ut[3 * j + 1], acc1);
                acc2 = mfma16(ga[3 * j + 2], ut[3 * j + 2], acc2);
            }
            float *dn = dhs + (cur ^ 1) * 16 * kGruHS;
#pragma unroll
            for (int r = 0; r < 4; ++r) dn[(4 * lq + r) * kGruHS + u] = (acc0[r] + acc1[r]) + acc2[r] + dhz[r];
            cur ^= 1;
            __syncthreads();                           // dh of step t-1 complete; G / Hp may be overwritten
        }
        // db0 = sums of dmx = [dz' dr' dhh], db1 = sums of dmh = [dz' dr' dmh_h]; lanes with equal li hold the same unit: reduce over lq
        sbz += __shfl_xor(sbz, 16, 64); sbz += __shfl_xor(sbz, 32, 64);
        sbr += __shfl_xor(sbr, 16, 64); sbr += __shfl_xor(sbr, 32, 64);
        sbh += __shfl_xor(sbh, 16, 64); sbh += __shfl_xor(sbh, 32, 64);
        sbm += __shfl_xor(sbm, 16, 64); sbm += __shfl_xor(sbm, 32, 64);
        if (lq == 0) {
            atomicAdd(db + u, sbz);
            atomicAdd(db + kGruU + u, sbr);
            atomicAdd(db + 2 * kGruU + u, sbh);
            atomicAdd(db + kGruN + u, sbz);
            atomicAdd(db + kGruN + kGruU + u, sbr);
            atomicAdd(db + kGruN + 2 * kGruU + u, sbm);
        }
    } else {
        f32x4 accU[9], accW[WT];
#pragma unroll
        for (int i = 0; i < 9; ++i) accU[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < WT; ++i) accW[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int t = T - 1; t >= 0; --t) {
            __syncthreads();                           // G, Hp of step t complete
            // every operand of the step's products in one batch of LDS reads, then the products from registers
            if (do_u) {
                float hv[4], gu[9][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) hv[j] = Hp[(4 * j + lq) * kGruHS + u];
#pragma unroll
                for (int nt = 0; nt < 9; ++nt) {
                    const int col = 16 * nt + li, gcol = col < 96 ? col : col + kGruU;
#pragma unroll
                    for (int j = 0; j < 4; ++j) gu[nt][j] = G[(4 * j + lq) * kGruGS + gcol];
                }
                // dU[16w + ..][:] += h_prev^T dmh (reduction index = clip); j outside: consecutive MFMAs write different accumulators
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int nt = 0; nt < 9; ++nt) accU[nt] = mfma16(hv[j], gu[nt][j], accU[nt]);
            } else {
                float gw[WT][4], xa[WT][4];
#pragma unroll
                for (int i = 0; i < WT; ++i) {
                    const int tile = w3 + 3 * i, tl = tile < MTW * 9 ? tile : 0;
                    const int mt = tl / 9, nt = tl % 9, f = 16 * mt + li;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        gw[i][j] = G[(4 * j + lq) * kGruGS + 16 * nt + li];
                        xa[i][j] = f < F ? xs[(4 * j + lq) * XS + t * F + f] : 0.f;
                    }
                }
                // dW tiles (feature rows x 144 dmx columns = the first 144 columns of G), dealt round-robin to the three waves; tiles past
                // the end multiply tile-0 operands into an accumulator that is never stored
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
            for (int nt = 0; nt < 9; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(dU + (16 * w3 + 4 * lq + r) * kGruN + 16 * nt + li, accU[nt][r]);
        }
#pragma unroll
        for (int i = 0; i < WT; ++i) {
            const int tile = w3 + 3 * i;
            if (!do_u && tile < MTW * 9) {
                const int mt = tile / 9, nt = tile % 9;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = 16 * mt + 4 * lq + r;
                    if (f < F) atomicAdd(dW + f * kGruN + 16 * nt + li, accW[i][r]);
                }
            }
        }
    }
}

inline size_t gru_fwd_smem(int T, int F) { return sizeof(float) * (size_t)(16 * gru_xstride(T, F) + 2 * 16 * kGruHS + 4 * 16 * kGruPS); }
inline size_t gru_bwd_smem(int T, int F)
{
    return sizeof(float) * (size_t)(16 * gru_xstride(T, F) + 16 * kGruGS + 16 * kGruHS + 2 * 16 * kGruHS);
}

}  // namespace kws
