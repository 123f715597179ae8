// csrc/kws_featurize.hip -- fused waveform -> MFCC/BFCC featurizer for gfx950 (MI355X).
//
// Replaces common/data_utils.py:73-86 (audio_to_feature -> vectorize_raw -> sonopy.mfcc_spec,
// optional add_deltas) of the reference for a whole batch in one launch.
//
// Mapping (n_fft = 1024 path):
//   block  = one clip, kWaves wavefronts of 64 lanes;  wave w owns frames w, w+kWaves, ...
//   frame  = 1024 real samples -> 512-point complex FFT done entirely inside ONE wave:
//            each lane holds 8 complex points, three radix-8 passes (8x8x8) with two
//            transposes through a per-wave 4.5 KB LDS tile (padded 72/9 so every half-wave
//            ds_read/ds_write_b64 is bank-conflict free), then the real-FFT split,
//            |X|^2/n_fft, a SPARSE band gather for the filterbank (per-lane chunks of one
//            band's non-zero span; no dense GEMM), eps-clipped log, ortho DCT-II, c0 <- log energy.
//   HBM    : every sample is requested once per frame it belongs to (2x with hop = window/2);
//            the second touch is an L1/L2 hit issued by the same block, so DRAM traffic stays at
//            the algorithmic 4 B/sample in + 4 B/feature out.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <functional>

#include "kws_common.h"
#include "kws_device.h"

#ifndef KWS_FEAT_ABLATE
#define KWS_FEAT_ABLATE 0
#endif

namespace kws {

constexpr int kWaves = 5;                 // wavefronts per clip (30 default frames -> 6 frames each)
constexpr int kThreads = kWaves * 64;
constexpr int kFftTile = 8 * 72;          // float2 elements per wave (padded 8 x 64 tile)
constexpr int kMaxChunks = 64;
constexpr float kEps = 2.220446049250313e-16f;  // np.finfo(float).eps, common/bark_feature.py:77

struct FeatDev {
    int window_eff, hop, max_samples, n_frames, n_filt, n_out, feature_size, use_delta, nchunks, nnz;
    int chp, n_filt_pad;   // chunk length padded to a multiple of 4 (zero weights); n_filt rounded up to 4 (zero DCT rows)
    int tail_batch;        // frames whose band sums / DCT one wave evaluates together (lanes = frames x bands)
    int fpw, jpc;          // frames per wave job, jobs per clip (set per launch: launch_featurize)
    const int32_t *index;  // NULL, or B device ints: clip b is ROW index[b] of wav / valid_len (kws_featurize_gather; set per launch)
    float inv_nfft;
    const float2 *tw1;   // [7][64]  W_512^(lane*k1), k1 = 1..7
    const float2 *tw2;   // [7][8]   W_64^(l2*k2a),   k2a = 1..7
    const float2 *tws;   // [257]    W_1024^k
    const int4 *chunks;  // [64 lanes] {band, first bin read, chunk id, 0}: lane placement is bank-conflict free (host matching)
    const int *bcs;      // [n_filt+1] first chunk of each band
    const float *w;      // [64 lanes][chp] bank weights of the lane's window, zero outside its chunk (nnz = 64*chp)
    const float *dct;    // [n_filt_pad][n_out], ortho scaling folded in, zero rows past n_filt
    // generic path (n_fft != 1024): radix-2 twiddles and the bank as per-band spans
    int n_fft, log2n;
    const float2 *twg;   // [n_fft/2]  W_nfft^k
    const int *bfirst;   // [n_filt] first non-zero bin of each band
    const int *bwidth;   // [n_filt] span length
    const float *bw;     // [n_filt][n_bins] dense bank rows (float)
    // tuned kernel (kws_featurize_v3.h): lane-linear LDS stores; chunks over POSITIONS of the two power planes, weights x 2^-12
    int chp3, blocks_per_cu;
    const int4 *chunks3; // [64 lanes] {band, first position read (even), chunk id, 0}
    const int *bcs3;     // [n_filt+1]
    const float *w3;     // [64 lanes][chp3]
    const float2 *tw1s;  // [7][64]  W_512^(sigma(lane) A), sigma(lane) = 8 (lane & 7) + (lane >> 3)
    const float2 *tws3;  // [4][64]  W_1024^(lane + 64 i)
};

}  // namespace kws

struct kws_featurizer {
    kws_params params;
    kws_geometry geom;
    int bank_kind;
    std::vector<float> bank;  // dense host copy (n_filt x n_bins)
    kws::FeatDev dev;
    void *dmem;               // one device allocation holding every table
    size_t smem_bytes;
    int blocks_per_cu = 2;    // persistent blocks per CU of the tuned kernel (kws_featurizer_set_cu_share)
};

namespace kws {

__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }  // a * (-i)

// forward 8-point DFT in registers, natural order in and out
__device__ __forceinline__ void dft8(float2 (&v)[8])
{
    const float h = 0.70710678118654752440f;
    float2 a0 = cadd(v[0], v[4]), a1 = csub(v[0], v[4]);
    float2 a2 = cadd(v[2], v[6]), a3 = mul_mi(csub(v[2], v[6]));
    float2 a4 = cadd(v[1], v[5]), a5 = csub(v[1], v[5]);
    float2 a6 = cadd(v[3], v[7]), a7 = mul_mi(csub(v[3], v[7]));
    float2 b0 = cadd(a0, a2), b2 = csub(a0, a2), b1 = cadd(a1, a3), b3 = csub(a1, a3);
    float2 b4 = cadd(a4, a6), b6 = csub(a4, a6), b5 = cadd(a5, a7), b7 = csub(a5, a7);
    float2 t1 = make_float2(h * (b5.x + b5.y), h * (b5.y - b5.x));    // b5 * (1 - i)/sqrt2
    float2 t2 = mul_mi(b6);                                           // b6 * (-i)
    float2 t3 = make_float2(h * (b7.y - b7.x), -h * (b7.x + b7.y));   // b7 * (-1 - i)/sqrt2
    v[0] = cadd(b0, b4); v[4] = csub(b0, b4);
    v[1] = cadd(b1, t1); v[5] = csub(b1, t1);
    v[2] = cadd(b2, t2); v[6] = csub(b2, t2);
    v[3] = cadd(b3, t3); v[7] = csub(b3, t3);
}

// all 64 lanes of a wave run in lockstep and the LDS serves one wave's DS ops in order, so a
// compiler-level fence is all an intra-wave LDS exchange needs
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(short v) { return (float)v * (1.0f / 32768.0f); }  // data_utils.py:21

template <typename WavT> struct Vec2;
template <> struct Vec2<float> { using type = float2; };
template <> struct Vec2<short> { using type = short2; };

__host__ __device__ inline int round4(int x) { return (x + 3) & ~3; }

// Complex points J0..J0+3 of the frame at sample `base`: z[n] = x[2n] + i x[2n+1], n = lane + 64 j.  A frame that lies
// wholly inside the recorded samples (wave-uniform test) takes the branch-free path: one address, immediate offsets.
template <typename WavT, int J0>
__device__ __forceinline__ void load_half(float2 (&v)[4], const WavT *__restrict__ src, int base, int pad, int window_eff,
                                          bool vec_ok, int lane)
{
    using V2 = typename Vec2<WavT>::type;
    if (vec_ok && base >= pad && window_eff >= 1024) {
        const V2 *p2 = reinterpret_cast<const V2 *>(src + (base - pad)) + lane + 64 * J0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const V2 t = p2[64 * j];
            v[j] = make_float2(to_f32(t.x), to_f32(t.y));
        }
    } else if (vec_ok) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int s0 = 2 * (lane + 64 * (J0 + j)), p0 = base + s0;
            float2 val = make_float2(0.f, 0.f);
            if (s0 < window_eff && p0 >= pad) {
                const V2 t = *reinterpret_cast<const V2 *>(src + (p0 - pad));
                val = make_float2(to_f32(t.x), to_f32(t.y));
            }
            v[j] = val;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int s0 = 2 * (lane + 64 * (J0 + j)), p0 = base + s0;
            float2 val = make_float2(0.f, 0.f);
            if (s0 < window_eff && p0 >= pad) val.x = to_f32(src[p0 - pad]);
            if (s0 + 1 < window_eff && p0 + 1 >= pad) val.y = to_f32(src[p0 + 1 - pad]);
            v[j] = val;
        }
    }
}

// One block per clip, each wave owns a run of consecutive frames.  The kernel is bound by instruction issue (about 900
// wave instructions per frame, 5 waves per SIMD), so the structure is chosen to cut instructions rather than bytes:
//   * consecutive frames overlap by half when hop = n_fft/2: the upper half of frame f is the lower half of frame f+1 and
//     stays in registers, so a frame costs 4 loads per lane, issued one frame ahead;
//   * the band sum, log and DCT run once per `tail_batch` frames with lane = (frame, band) / (frame, coefficient), which
//     keeps ~60 of 64 lanes busy instead of 20; results go straight to global memory (no staging unless use_delta);
//   * per-lane table entries (chunk start, band span) live in registers, not LDS.
// LDS per block stays under 40 KB for 4 blocks per CU.
template <typename WavT>
__global__ __launch_bounds__(kThreads, 5) void featurize_fft1024_kernel(const WavT *__restrict__ wav, int64_t stride,
                                                                      const int32_t *__restrict__ valid_len, int B,
                                                                      FeatDev c, float *__restrict__ feat)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    // the wave index is uniform: telling the compiler so keeps the frame loop, its counters and branches in scalar registers
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // A wave owns one JOB = c.fpw consecutive frames of one clip.  Without deltas the jobs of all clips are dealt to the
    // waves in order (job = blockIdx.x * kWaves + wave), so short clips -- a streaming step featurizes 2 frames per stream --
    // still fill every wave; at the default geometry (30 frames, 6 per job, 5 jobs per clip) this is block = clip.  With
    // deltas the block keeps one clip (its coefficients are staged in s_feat for the frame differences).
    const int job = c.use_delta ? (wave < c.jpc ? (int)blockIdx.x * c.jpc + wave : -1) : (int)blockIdx.x * kWaves + wave;
    const int b = job < 0 ? B : job / c.jpc;

    const int tb = c.tail_batch;
    float2 *s_fft = reinterpret_cast<float2 *>(smem) + wave * kFftTile;
    float *s_pw = reinterpret_cast<float *>(s_fft);                       // power spectrum aliases the FFT tile
    float *s_part = reinterpret_cast<float *>(smem + kWaves * kFftTile * 8) + wave * (tb * 64 + 64 + 4);   // [tb][64]
    float *s_mel = s_part + tb * 64;                                       // [tb][n_filt_pad] (<= 64)
    float *s_en = s_mel + 64;                                              // [tb] (<= 4)
    float *s_dct = reinterpret_cast<float *>(smem + kWaves * kFftTile * 8) + kWaves * (tb * 64 + 64 + 4);
    float *s_w = s_dct + round4(c.n_filt_pad * c.n_out);
    float2 *s_tw1 = reinterpret_cast<float2 *>(s_w + round4(c.nnz));      // [7][64]
    float2 *s_tw2 = s_tw1 + 7 * 64;                                        // [7][8]
    int *s_bcs = reinterpret_cast<int *>(s_tw2 + 7 * 8);                   // [n_filt + 1] first chunk of each band (<= 68 ints)
    float *s_feat = reinterpret_cast<float *>(s_bcs + 68);                 // [n_frames][n_out], only with use_delta

    for (int i = tid; i < c.n_filt_pad * c.n_out; i += kThreads) s_dct[i] = c.dct[i];
    for (int i = tid; i < c.nnz; i += kThreads) s_w[i] = c.w[i];
    for (int i = tid; i < 7 * 64; i += kThreads) s_tw1[i] = c.tw1[i];
    for (int i = tid; i < 7 * 8; i += kThreads) s_tw2[i] = c.tw2[i];
    for (int i = tid; i <= c.n_filt; i += kThreads) s_bcs[i] = c.bcs[i];
    s_mel[lane] = 0.f;                                                     // the zero padding past n_filt is read by the DCT
    __syncthreads();

    // clip geometry: keep the head, left-pad zeros (data_utils.py:77-80)
    const int bc = b < B ? b : 0;                                          // idle waves (no job) read clip 0's geometry and do nothing
    const int row = c.index ? c.index[bc] : bc;
    int len = valid_len ? valid_len[row] : (stride > c.max_samples ? c.max_samples : (int)stride);
    len = len < 0 ? 0 : len;
    if ((int64_t)len > stride) len = (int)stride;
    if (len > c.max_samples) len = c.max_samples;
    const int pad = c.max_samples - len;
    const WavT *src = wav + (int64_t)row * stride;
    const bool vec_ok = (((pad | c.hop | c.window_eff) & 1) == 0) &&
                        ((reinterpret_cast<uintptr_t>(src) & (2 * sizeof(WavT) - 1)) == 0);
    const bool reuse = 2 * c.hop == 1024 && c.window_eff == 1024;          // frame f+1 starts with frame f's upper half

    const int hi = lane >> 3, lo = lane & 7;
    // per-lane roles in the batched tail
    // gather: lane = one chunk of one band; its first bin and its chunk id (the slot of its partial sum) packed in one register
    const int chunk_pack = c.chunks[lane].y | (c.chunks[lane].z << 16);
    const float r_filt = 1.0f / (float)c.n_filt, r_out = 1.0f / (float)c.n_out;
    float *dst = feat + (int64_t)bc * c.n_frames * c.feature_size;

    const int f_beg = b < B ? (job - b * c.jpc) * c.fpw : c.n_frames;
    const int f_end = f_beg + c.fpw < c.n_frames ? f_beg + c.fpw : c.n_frames;

    float2 xl[4], xh[4];                // lower / upper half of the next frame to transform
    if (f_beg < f_end) {
        load_half<WavT, 0>(xl, src, f_beg * c.hop, pad, c.window_eff, vec_ok, lane);
        load_half<WavT, 4>(xh, src, f_beg * c.hop, pad, c.window_eff, vec_ok, lane);
    }
    int qi = 0;                          // frames waiting in this wave's tail batch
    for (int f = f_beg; f < f_end; ++f) {
        float2 v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = xl[j]; v[j + 4] = xh[j]; }
        if (f + 1 < f_end) {             // next frame's samples: their HBM latency hides under this frame's FFT
            if (reuse) {
#pragma unroll
                for (int j = 0; j < 4; ++j) xl[j] = xh[j];
            } else {
                load_half<WavT, 0>(xl, src, (f + 1) * c.hop, pad, c.window_eff, vec_ok, lane);
            }
            load_half<WavT, 4>(xh, src, (f + 1) * c.hop, pad, c.window_eff, vec_ok, lane);
        }

#if (KWS_FEAT_ABLATE & 16)
        { float acc = 0.f;
          for (int j = 0; j < 8; ++j) acc += v[j].x + v[j].y;
          if (f + 1 == f_end) dst[f * c.n_out + (lane % c.n_out)] = acc;
          continue; }
#endif
        // pass 1: DFT-8 over n1 (n = lane + 64 n1), twiddle W_512^(lane*k1)
        dft8(v);
#pragma unroll
        for (int k = 1; k < 8; ++k) v[k] = cmul(v[k], s_tw1[(k - 1) * 64 + lane]);
        wave_sync();
#pragma unroll
        for (int k = 0; k < 8; ++k) s_fft[72 * k + lane] = v[k];
        wave_sync();
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = s_fft[72 * hi + lo + 8 * j];

#if !(KWS_FEAT_ABLATE & 4)
        // pass 2: lane = (k1, l2); DFT-8 over l1, twiddle W_64^(l2*k2a)
        dft8(v);
#pragma unroll
        for (int k = 1; k < 8; ++k) v[k] = cmul(v[k], s_tw2[(k - 1) * 8 + lo]);
        wave_sync();
#pragma unroll
        for (int k = 0; k < 8; ++k) s_fft[72 * hi + 9 * k + lo] = v[k];
        wave_sync();
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = s_fft[72 * hi + 9 * lo + j];

        // pass 3: lane = (k1, k2a); DFT-8 over l2 -> Z[k1 + 8 k2a + 64 k2b]
        dft8(v);
#endif
        wave_sync();
#pragma unroll
        for (int r = 0; r < 8; ++r) {       // natural order k = hi + 8 lo + 64 r, stored at k + (k >> 3) = hi + 9 lo + 72 r: the
            s_fft[hi + 9 * lo + 72 * r] = v[r];   // plain index is a 4-way bank conflict (lo strides 16 words), the padded one 1-2 way
        }
        wave_sync();

        // real-FFT split: X[k] = E[k] + W_1024^k O[k], X[512-k] = conj(E[k] - W_1024^k O[k])
        float pk[4], pm[4], p256 = 0.f, energy = 0.f;
#if (KWS_FEAT_ABLATE & 2)
        for (int i = 0; i < 4; ++i) { pk[i] = v[i].x; pm[i] = v[i].y; energy += v[i + 4].x + v[i + 4].y; }
#else
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = lane + 64 * i;
            const int km = (512 - k) & 511;
            const float2 zk = s_fft[k + (k >> 3)], zm = s_fft[km + (km >> 3)];
            const float2 E = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
            const float2 O = make_float2(0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x));
            const float2 T = cmul(c.tws[k], O);
            const float2 xp = cadd(E, T), xm = csub(E, T);
            pk[i] = (xp.x * xp.x + xp.y * xp.y) * c.inv_nfft;   // bark_feature.py:88-89
            pm[i] = (xm.x * xm.x + xm.y * xm.y) * c.inv_nfft;
            energy += pk[i] + pm[i];
        }
        if (lane == 0) {  // bin 256 is its own partner
            const float2 z = s_fft[256 + 32];
            p256 = (z.x * z.x + z.y * z.y) * c.inv_nfft;
            energy += p256;
        }
#endif
        wave_sync();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = lane + 64 * i;
            s_pw[k] = pk[i];
            s_pw[512 - k] = pm[i];
        }
        if (lane == 0) s_pw[256] = p256;
        energy = wave_sum_dpp(energy);      // VALU-only reduction (6 ds_bpermute round trips otherwise)
        wave_sync();

        // sparse band gather: lane = one chunk (<= chp bins) of one band's non-zero span.  Fixed trip count over
        // zero-padded weights so the LDS loads of a group issue back to back.
#if (KWS_FEAT_ABLATE & 1)
        if (f + 1 == f_end) dst[f * c.n_out + (lane % c.n_out)] = energy + s_pw[lane];
        continue;
#endif
        float part = 0.f;
        {
            const float *pp = s_pw + (chunk_pack & 0xFFFF);
            const float *wp = s_w + lane * c.chp;
            for (int t = 0; t < c.chp; t += 4) {
                const float4 wv = *reinterpret_cast<const float4 *>(wp + t);
                const float p0 = pp[t], p1 = pp[t + 1], p2 = pp[t + 2], p3 = pp[t + 3];
                part = fmaf(p0, wv.x, part);
                part = fmaf(p1, wv.y, part);
                part = fmaf(p2, wv.z, part);
                part = fmaf(p3, wv.w, part);
            }
        }
        s_part[qi * 64 + (chunk_pack >> 16)] = part;
        if (lane == 0) s_en[qi] = energy;
        ++qi;
        if (qi < tb && f + 1 < f_end) continue;
#if (KWS_FEAT_ABLATE & 8)
        wave_sync();
        if (lane < qi * c.n_out) dst[(f + 1 - qi) * c.n_out + lane] = s_part[lane] + s_en[0];
        qi = 0;
        continue;
#endif

        // ---- tail of the qi frames f-qi+1 .. f: band sums -> log -> DCT, all frames of the batch at once ----
        // lane roles are recomputed per batch (twice per wave) rather than held in registers across the FFTs;
        // (lane + 0.5) / n is at least 0.5/64 away from an integer, so the reciprocal multiply floors exactly
        const int f0 = f + 1 - qi;
        int tl = lane;                     // opaque copy: keeps the role arithmetic below inside the tail (it is loop-invariant,
        asm volatile("" : "+v"(tl));     // so the compiler would otherwise hoist it out of the frame loop and spill it)
        wave_sync();
        {
            const int bq = (int)(((float)tl + 0.5f) * r_filt), bm = tl - bq * c.n_filt;       // lane = (frame in batch, band)
            if (bq < qi) {
                const int q0 = s_bcs[bm], cnt = s_bcs[bm + 1] - q0;
                const float *pq = s_part + bq * 64 + q0;
                float sum = 0.f;
                for (int g = 0; g < cnt; g += 8) {                       // reads past cnt stay inside s_part / s_mel and are masked
                    float pv[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) pv[i] = pq[g + i];
#pragma unroll
                    for (int i = 0; i < 8; ++i) sum += (g + i < cnt) ? pv[i] : 0.f;
                }
                s_mel[bq * c.n_filt_pad + bm] = logf(fmaxf(sum, kEps));   // safe_log, bark_feature.py:75-77
            }
        }
        wave_sync();
        {
            const int dq = (int)(((float)tl + 0.5f) * r_out), dn = tl - dq * c.n_out;         // lane = (frame in batch, coefficient)
            if (dq < qi) {
                const float *mq = s_mel + dq * c.n_filt_pad;
                float sum = 0.f;
                for (int n = 0; n < c.n_filt_pad; n += 4) {
                    const float4 mv = *reinterpret_cast<const float4 *>(mq + n);
                    const float d0 = s_dct[n * c.n_out + dn], d1 = s_dct[(n + 1) * c.n_out + dn];
                    const float d2 = s_dct[(n + 2) * c.n_out + dn], d3 = s_dct[(n + 3) * c.n_out + dn];
                    sum = fmaf(mv.x, d0, sum);
                    sum = fmaf(mv.y, d1, sum);
                    sum = fmaf(mv.z, d2, sum);
                    sum = fmaf(mv.w, d3, sum);
                }
                if (dn == 0) sum = logf(fmaxf(s_en[dq], kEps));           // c0 <- log energy, bark_feature.py:173
                if (c.use_delta) s_feat[f0 * c.n_out + tl] = sum;         // lane = dq * n_out + dn
                else dst[f0 * c.n_out + tl] = sum;                        // feature_size == n_out: rows f0.. are contiguous
            }
        }
        qi = 0;
        wave_sync();
    }
    if (!c.use_delta) return;
    __syncthreads();
    dst = feat + (int64_t)blockIdx.x * c.n_frames * c.feature_size;        // with deltas the block is one clip

    // add_deltas (data_utils.py:50-58): the clip's (n_features x 2 n_out) block from the staged coefficients
    const int total = c.n_frames * c.feature_size;
    for (int i = tid; i < total; i += kThreads) {
        const int fr = i / c.feature_size, col = i - fr * c.feature_size;
        float val;
        if (col < c.n_out) val = s_feat[fr * c.n_out + col];
        else val = fr == 0 ? 0.f : s_feat[fr * c.n_out + col - c.n_out] - s_feat[(fr - 1) * c.n_out + col - c.n_out];
        dst[i] = val;
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Generic fallback for n_fft != 1024 (any power of two 64..4096): same block = clip / wave = frames mapping, but a plain
// in-LDS radix-2 FFT of the zero-padded real frame and per-band loops.  Correct for every params.json the reference
// accepts; the tuned kernel above is the one the default geometry (and the benchmark) uses.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kGenWaves = 4;

template <typename WavT>
__global__ __launch_bounds__(kGenWaves * 64) void featurize_generic_kernel(const WavT *__restrict__ wav, int64_t stride,
                                                                           const int32_t *__restrict__ valid_len, int B, FeatDev c,
                                                                           float *__restrict__ feat)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    if (b >= B) return;
    const int N = c.n_fft, nb = N / 2 + 1;
    float2 *z = reinterpret_cast<float2 *>(smem) + (size_t)wave * N;                       // [N] complex per wave
    float *s_mel = reinterpret_cast<float *>(smem + (size_t)kGenWaves * N * 8) + wave * 64;
    float *s_feat = reinterpret_cast<float *>(smem + (size_t)kGenWaves * N * 8 + kGenWaves * 256);

    const int row = c.index ? c.index[b] : b;
    int len = valid_len ? valid_len[row] : (stride > c.max_samples ? c.max_samples : (int)stride);
    len = len < 0 ? 0 : len;
    if ((int64_t)len > stride) len = (int)stride;
    if (len > c.max_samples) len = c.max_samples;
    const int pad = c.max_samples - len;
    const WavT *src = wav + (int64_t)row * stride;

    for (int f = wave; f < c.n_frames; f += kGenWaves) {
        const int base = f * c.hop;
        for (int i = lane; i < N; i += 64) {                       // bit-reversed load of the (cropped / zero-padded) frame
            const int p = base + i;
            const float v = (i < c.window_eff && p >= pad) ? to_f32(src[p - pad]) : 0.f;
            z[__brev((unsigned)i) >> (32 - c.log2n)] = make_float2(v, 0.f);
        }
        wave_sync();
        for (int st = 1; st <= c.log2n; ++st) {                    // radix-2 decimation in time
            const int half = 1 << (st - 1), tstep = N >> st;
            for (int i = lane; i < N / 2; i += 64) {
                const int k = i & (half - 1), j = ((i >> (st - 1)) << st) + k;
                const float2 w = c.twg[k * tstep], u = z[j], t = cmul(w, z[j + half]);
                z[j] = cadd(u, t);
                z[j + half] = csub(u, t);
            }
            wave_sync();
        }
        float e = 0.f;
        float *pw = reinterpret_cast<float *>(z);                  // power spectrum overwrites the low half in place
        for (int k0 = 0; k0 < nb; k0 += 64) {
            const int k = k0 + lane;
            float pk = 0.f;
            if (k < nb) { const float2 x = z[k]; pk = (x.x * x.x + x.y * x.y) * c.inv_nfft; }
            wave_sync();                                           // every lane has read its z[k] of this group
            if (k < nb) pw[k] = pk;
            e += pk;
        }
        e = wave_sum(e);
        wave_sync();
        float melv = 0.f;
        if (lane < c.n_filt) {
            const float *w = c.bw + (size_t)lane * nb + c.bfirst[lane];
            const float *pp = pw + c.bfirst[lane];
            float sacc = 0.f;
            for (int t = 0; t < c.bwidth[lane]; ++t) sacc = fmaf(pp[t], w[t], sacc);
            melv = logf(fmaxf(sacc, kEps));
        }
        s_mel[lane] = melv;
        wave_sync();
        if (lane < c.n_out) {
            float sacc = 0.f;
            for (int n = 0; n < c.n_filt; ++n) sacc = fmaf(s_mel[n], c.dct[n * c.n_out + lane], sacc);
            if (lane == 0) sacc = logf(fmaxf(e, kEps));
            s_feat[f * c.n_out + lane] = sacc;
        }
        wave_sync();
    }
    __syncthreads();
    float *dst = feat + (int64_t)b * c.n_frames * c.feature_size;
    const int total = c.n_frames * c.feature_size;
    for (int i = tid; i < total; i += kGenWaves * 64) {
        const int fr = i / c.feature_size, col = i - fr * c.feature_size;
        float val;
        if (col < c.n_out) val = s_feat[fr * c.n_out + col];
        else val = fr == 0 ? 0.f : s_feat[fr * c.n_out + col - c.n_out] - s_feat[(fr - 1) * c.n_out + col - c.n_out];
        dst[i] = val;
    }
}

}  // namespace kws
#include "kws_featurize_v3.h"
namespace kws {

// ---------------------------------------------------------------------------------------------
// host side: parameter geometry and table construction (double precision, then rounded once)
// ---------------------------------------------------------------------------------------------
constexpr size_t kMaxLdsBytes = 160 * 1024;      // LDS per workgroup on gfx950

// wave jobs of the n_fft = 1024 kernel: fpw consecutive frames of one clip.  Long clips: an equal share per wave of a block;
// short ones (a streaming step has 2 frames per stream): a whole tail batch per job so that one wave serves a stream
static void feat_job_shape(FeatDev &d)
{
    d.fpw = (d.n_frames + kWaves - 1) / kWaves;
    if (!d.use_delta) d.fpw = std::max(d.fpw, std::min(d.n_frames, d.tail_batch));
    d.fpw = std::max(1, d.fpw);
    d.jpc = std::max(1, (d.n_frames + d.fpw - 1) / d.fpw);
}

static size_t generic_smem_bytes(const FeatDev &d)
{
    return (size_t)kGenWaves * d.n_fft * 8 + kGenWaves * 256 + 4 * (size_t)round4(d.n_frames * d.n_out);
}

static size_t feat_smem_bytes(const FeatDev &d)
{
    return (size_t)kWaves * (kFftTile * 8 + 4 * (d.tail_batch * 64 + 64 + 4)) +
           4 * (size_t)(round4(d.n_filt_pad * d.n_out) + round4(d.nnz)) + 8 * (7 * 64 + 7 * 8) + 272 +
           (d.use_delta ? 4 * (size_t)round4(d.n_frames * d.n_out) : 0);
}

static int derive(const kws_params *p, kws_geometry *g)
{
    if (!p || !g) return fail(KWS_ERR_INVALID, "null params");
    if (p->sample_rate <= 0 || p->window_t <= 0 || p->hop_t <= 0 || p->buffer_t <= 0)
        return fail(KWS_ERR_INVALID, "sample_rate/window_t/hop_t/buffer_t must be positive");
    // classifier/params.py:59-91
    g->window_samples = (int)(p->sample_rate * p->window_t + 0.5);
    g->hop_samples = (int)(p->sample_rate * p->hop_t + 0.5);
    g->max_samples = (int)(p->buffer_t * p->sample_rate);
    if (g->hop_samples <= 0 || g->window_samples <= 0) return fail(KWS_ERR_INVALID, "window/hop round to zero samples");
    int samples = (int)(p->sample_rate * p->buffer_t + 0.5);
    g->buffer_samples = g->hop_samples * (samples / g->hop_samples);
    g->n_features = 1 + (int)std::floor((double)(g->buffer_samples - g->window_samples) / (double)g->hop_samples);
    g->feature_size = p->use_delta ? 2 * p->n_mfcc : p->n_mfcc;
    return KWS_OK;
}

static int build_mel_bank(int sample_rate, int n_fft, int n_filt, std::vector<double> &bank)
{
    // sonopy.filterbanks as restated by inference/tflite/mfcc.h:230-264; span 0..sample_rate
    // (inference/tflite/speech_commands.h:304-307)
    const int n_bins = n_fft / 2 + 1, n = n_filt + 2;
    std::vector<int> pts(n);
    const double lo = 1127.0 * std::log(1.0 + 0.0 / 700.0), hi = 1127.0 * std::log(1.0 + sample_rate / 700.0);
    const double step = (hi - lo) / (double)(n - 1);
    for (int i = 0; i < n; ++i) {
        const double m = (i == n - 1) ? hi : lo + i * step;
        pts[i] = (int)(700.0 * (std::exp(m / 1127.0) - 1.0) * n_bins / sample_rate);
    }
    for (int i = 1; i < n; ++i)
        if (pts[i] <= pts[i - 1])
            return fail(KWS_ERR_UNSUPPORTED, "mel grid has repeated FFT bins for n_fft=%d n_filt=%d (sonopy's "
                                             "duplicate-point correction is not implemented)", n_fft, n_filt);
    if (pts[n - 1] > n_bins) return fail(KWS_ERR_INVALID, "mel grid exceeds the spectrum");
    bank.assign((size_t)n_filt * n_bins, 0.0);
    for (int i = 0; i < n_filt; ++i) {
        const int l = pts[i], m = pts[i + 1], r = pts[i + 2];
        for (int j = l; j < m; ++j) bank[(size_t)i * n_bins + j] = (double)(j - l) / (double)(m - l);
        for (int j = m; j < r; ++j) bank[(size_t)i * n_bins + j] = (double)(r - j) / (double)(r - m);
    }
    return KWS_OK;
}

static int build_bark_bank(int sample_rate, int n_fft, int n_filt, std::vector<double> &bank)
{
    // common/bark_feature.py:92-136 with scale="constant".  The bin<->Hz maps at :112/:134 are
    // called with their DEFAULT nfft=512 and sample_rate=16000 whatever the caller passes; kept.
    const int n_bins = n_fft / 2 + 1, n = n_filt + 4;
    auto hz2bark = [](double f) { return 6.0 * std::asinh(f / 600.0); };
    auto bark2hz = [](double v) { return 600.0 * std::sinh(v / 6.0); };
    auto Fm = [](double fb, double fc) {
        if (fc - 2.5 <= fb && fb <= fc - 0.5) return std::pow(10.0, 2.5 * (fb - fc + 0.5));
        if (fc - 0.5 < fb && fb < fc + 0.5) return 1.0;
        if (fc + 0.5 <= fb && fb <= fc + 1.3) return std::pow(10.0, -2.5 * (fb - fc - 0.5));
        return 0.0;
    };
    const double lo = hz2bark(0.0), hi = hz2bark(sample_rate / 2.0), step = (hi - lo) / (double)(n - 1);
    std::vector<double> pt(n);
    std::vector<int> bins(n);
    for (int i = 0; i < n; ++i) {
        pt[i] = (i == n - 1) ? hi : lo + i * step;
        bins[i] = (int)std::floor(513.0 * bark2hz(pt[i]) / 16000.0);
    }
    bank.assign((size_t)n_filt * n_bins, 0.0);
    for (int i = 0; i < n_filt; ++i)
        for (int j = bins[i]; j < bins[i + 4]; ++j) {
            if (j < 0 || j >= n_bins) return fail(KWS_ERR_INVALID, "bark bank needs bin %d but n_fft=%d has %d", j, n_fft, n_bins);
            bank[(size_t)i * n_bins + j] = std::fabs(Fm(hz2bark(j * 16000.0 / 513.0), pt[i + 2]));
        }
    return KWS_OK;
}

}  // namespace kws

using namespace kws;

// tuned kernel (kws_featurize_v3.h): default frame geometry, 20 bands, 20 coefficients, no deltas
constexpr int kTunedBands = 20, kTunedCoefs = 20;
static bool v3_applies(const FeatDev &d)
{
    return d.n_fft == 1024 && d.hop == 512 && d.window_eff == 1024 && !d.use_delta && d.n_filt == kTunedBands && d.n_out == kTunedCoefs &&
           (d.chp3 == 12 || d.chp3 == 16 || d.chp3 == 20);
}
static size_t v3_smem_bytes(int chp, int waves)
{
    constexpr int TB = 64 / kTunedBands;
    return (size_t)waves * 4 * (kV3Tile + TB * 64 + 64 + 4) + 4 * (size_t)(kTunedBands * kTunedCoefs + 64 * chp) + 8 * (size_t)(7 * 64 + 7 * 8 + 4 * 64) +
           4 * (size_t)round4(kTunedBands + 1);
}
template <typename WavT, int CHP, int WAVES>
static int launch_v3(const FeatDev &d, const WavT *wav, int B, int64_t stride, const int32_t *valid_len, float *feat, hipStream_t s, const char *name, int bpc)
{
    const int cus = device_cus();
    const size_t smem = v3_smem_bytes(CHP, WAVES);
    if ((size_t)WAVES * 4 * kV3Tile > 65536) return fail(KWS_ERR_UNSUPPORTED, "wave tiles must sit in the first 64 KiB of LDS (M0 holds 16 address bits)");
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&featurize_fft1024_v3_kernel<WavT, CHP, kTunedBands, kTunedCoefs, WAVES>), (int)smem)) return rc;
    FeatDev dd = d;
    const long waves = (long)bpc * cus * WAVES;
    {
        // cost of a candidate = rounds of jobs per wave x (frames per job + the job's fixed part: one extra half frame of loads
        // and a partly filled tail batch); candidates: the tail batch size and up, preferring divisors of the frame count
        double best = -1.0;
        for (int fpw = std::min(dd.n_frames, std::max(1, dd.tail_batch)); fpw <= dd.n_frames; ++fpw) {
            const int jpc = (dd.n_frames + fpw - 1) / fpw;
            const long jobs = (long)B * jpc, rounds = (jobs + waves - 1) / waves;
            const double cost = (double)rounds * ((double)fpw + 1.25) + ((dd.n_frames % fpw) ? 0.5 : 0.0);
            if (best < 0 || cost < best - 1e-9) { best = cost; dd.fpw = fpw; dd.jpc = jpc; }
        }
    }
    const long jobs = (long)B * dd.jpc;
    const unsigned grid = (unsigned)std::min<long>((long)bpc * cus, (jobs + WAVES - 1) / WAVES);
    KWS_LAUNCH(name, (featurize_fft1024_v3_kernel<WavT, CHP, kTunedBands, kTunedCoefs, WAVES>), dim3(grid), dim3(WAVES * 64), smem, s, wav, stride,
               valid_len, B, dd, feat);
    KWS_LAUNCH_CHECK("featurize_fft1024_v3_kernel");
    return KWS_OK;
}
#ifndef KWS_V3_ALONE_WAVES
#define KWS_V3_ALONE_WAVES 8
#endif
#ifndef KWS_V3_ALONE_BLOCKS
#define KWS_V3_ALONE_BLOCKS 2
#endif
template <typename WavT, int WAVES>
static int launch_v3_chp(const FeatDev &d, const WavT *wav, int B, int64_t stride, const int32_t *valid_len, float *feat, hipStream_t s, const char *name, int bpc)
{
    switch (d.chp3) {
    case 12: return launch_v3<WavT, 12, WAVES>(d, wav, B, stride, valid_len, feat, s, name, bpc);
    case 16: return launch_v3<WavT, 16, WAVES>(d, wav, B, stride, valid_len, feat, s, name, bpc);
    default: return launch_v3<WavT, 20, WAVES>(d, wav, B, stride, valid_len, feat, s, name, bpc);
    }
}
// The kernel holds its per-lane twiddles in registers (110 registers: 4 waves per SIMD = 16 per CU).  With the chip to itself: two blocks of 8
// waves per CU; beside a train step (kws_featurizer_set_cu_share(f, 1)): ONE block of 12 waves per CU, which leaves a quarter of the wave
// slots and half of the LDS to the step's kernels.
template <typename WavT>
static int launch_v3_any(const FeatDev &d, const WavT *wav, int B, int64_t stride, const int32_t *valid_len, float *feat, hipStream_t s, const char *name)
{
    if (d.blocks_per_cu == 1) return launch_v3_chp<WavT, kV3Waves>(d, wav, B, stride, valid_len, feat, s, name, 1);
    return launch_v3_chp<WavT, KWS_V3_ALONE_WAVES>(d, wav, B, stride, valid_len, feat, s, name, KWS_V3_ALONE_BLOCKS);
}

extern "C" {

void kws_params_default(kws_params *p)
{
    if (!p) return;
    // classifier/params.py:99-103
    p->buffer_t = 1.0; p->window_t = 0.064; p->hop_t = 0.032;
    p->sample_rate = 16000; p->sample_depth = 2; p->n_fft = 1024; p->n_filt = 20; p->n_mfcc = 20; p->use_delta = 0;
}

int kws_params_derive(const kws_params *p, kws_geometry *g) { return derive(p, g); }

int kws_featurizer_create(const kws_params *p, int bank_kind, kws_featurizer **out)
{
    if (!p || !out) return fail(KWS_ERR_INVALID, "null argument");
    *out = nullptr;
    kws_geometry g;
    int rc = derive(p, &g);
    if (rc) return rc;
    if (bank_kind != KWS_BANK_MEL && bank_kind != KWS_BANK_BARK) return fail(KWS_ERR_INVALID, "unknown bank kind %d", bank_kind);
    if (p->n_filt < 1 || p->n_filt > 64) return fail(KWS_ERR_UNSUPPORTED, "n_filt must be in 1..64, got %d", p->n_filt);
    if (p->n_mfcc < 1 || p->n_mfcc > p->n_filt)
        return fail(KWS_ERR_INVALID, "n_mfcc=%d must be in 1..n_filt=%d (the reference's feature_size would not match)", p->n_mfcc, p->n_filt);
    if (p->n_fft < 64 || p->n_fft > 4096 || (p->n_fft & (p->n_fft - 1)))
        return fail(KWS_ERR_UNSUPPORTED, "n_fft must be a power of two in 64..4096 (got %d)", p->n_fft);
    if (g.max_samples < g.window_samples) return fail(KWS_ERR_INVALID, "buffer shorter than one window");
    const int n_frames = (g.max_samples - g.window_samples) / g.hop_samples + 1;
    if (n_frames != g.n_features)
        return fail(KWS_ERR_INVALID, "params give %d frames but n_features=%d; the reference's model input would not match", n_frames, g.n_features);

    std::vector<double> bank;
    rc = bank_kind == KWS_BANK_MEL ? build_mel_bank(p->sample_rate, p->n_fft, p->n_filt, bank)
                                   : build_bark_bank(p->sample_rate, p->n_fft, p->n_filt, bank);
    if (rc) return rc;
    const int n_bins = p->n_fft / 2 + 1, n_filt = p->n_filt, n_out = p->n_mfcc;

    // sparse tables: per band the span [first non-zero, last non-zero], cut into <= 64 lane chunks
    std::vector<int> first(n_filt, 0), width(n_filt, 0);
    for (int i = 0; i < n_filt; ++i) {
        int a = -1, z = -1;
        for (int j = 0; j < n_bins; ++j)
            if (bank[(size_t)i * n_bins + j] != 0.0) { if (a < 0) a = j; z = j; }
        if (a >= 0) { first[i] = a; width[i] = z - a + 1; }
    }
    // Chunk tables for the sparse band gather.  `align` = 1: a lane reads its bins one per LDS instruction (first-generation
    // kernel); 2: chunks start on even positions so that two bins come per ds_read_b64 (kws_featurize_v3.h).
    struct ChunkTables { int ch, chp, nlog; std::vector<int4> chunks; std::vector<float> w; std::vector<int> bcs; };
    // `mirrored` (third-generation kernel): the chunks run over POSITIONS of the wave's two power planes (v3_pos_of_bin: bins 0..256 in
    // place, bins 257..512 in descending order behind them), a band's span splits into at most two runs of consecutive positions, and the
    // weights carry the factor 2^-12 the kernel leaves out of its power spectrum
    auto build_chunks = [&](int align, bool mirrored = false) {
        ChunkTables T;
        const int npos = mirrored ? kV3OffM + 256 : n_bins;
        std::vector<int> bin_of(npos, -1);
        for (int k = 0; k < n_bins; ++k) bin_of[mirrored ? v3_pos_of_bin(k) : k] = k;
        const double wscale = mirrored ? 1.0 / 4096.0 : 1.0;
        struct Run { int band, lo, hi; };                        // positions [lo, hi)
        std::vector<Run> runs;
        for (int i = 0; i < n_filt; ++i) {
            if (width[i] <= 0) continue;
            if (!mirrored) { runs.push_back(Run{i, first[i], first[i] + width[i]}); continue; }
            const int a = first[i], z = first[i] + width[i] - 1;
            if (a <= 256) runs.push_back(Run{i, a, std::min(z, 256) + 1});
            if (z >= 257) runs.push_back(Run{i, v3_pos_of_bin(z), v3_pos_of_bin(std::max(a, 257)) + 1});
        }
        int ch = align;
        for (;; ch += align) {
            int cnt = 0;
            for (const Run &r : runs) cnt += ((r.hi - r.lo) + (r.lo % align) + ch - 1) / ch;
            if (cnt <= kMaxChunks) break;
        }
        const int chp = (ch + 3) & ~3;
        // logical chunks in band order: (band, first position, positions); chunk boundaries sit on multiples of `align`
        struct Chunk { int band, a, len; };
        std::vector<Chunk> logical;
        std::vector<int> bcs(n_filt + 1, 0);
        {
            size_t ri = 0;
            for (int i = 0; i < n_filt; ++i) {
                bcs[i] = (int)logical.size();
                for (; ri < runs.size() && runs[ri].band == i; ++ri) {
                    const int s0 = runs[ri].lo - runs[ri].lo % align, end = runs[ri].hi;
                    for (int lo = s0; lo < end; lo += ch) {
                        const int a = std::max(lo, runs[ri].lo), z = std::min(lo + ch, end);
                        if (z > a) logical.push_back(Chunk{i, a, z - a});
                    }
                }
            }
        }
        bcs[n_filt] = (int)logical.size();
        const int nlog = (int)logical.size();
        // Lane placement.  Every lane reads p[start + t] for the same t, so the 32 lanes of a half-wave hit different LDS banks
        // iff their starts (in units of `align` bins) differ mod 32.  A chunk of `len` bins may start anywhere (on a multiple
        // of `align`) in [a + len - chp, a] (the extra bins get zero weights), so: bipartite matching of chunks to the 64
        // (half, residue) slots, lane = 32 * half + residue.  (Chunk starts taken as they come collide ~3-way on average:
        // 29 % of the first kernel's LDS cycles were conflicts.)
        std::vector<int> start_of(nlog, 0), owner(64, -1);
        {
            std::vector<std::vector<std::pair<int, int>>> cand(nlog);          // (slot, start)
            for (int c = 0; c < nlog; ++c)
                for (int st = logical[c].a - logical[c].a % align; st >= std::max(0, logical[c].a + logical[c].len - chp); st -= align)
                    for (int h = 0; h < 2; ++h) cand[c].push_back({32 * h + ((st / align) & 31), st});
            std::vector<int> owner_start(64, 0);
            std::function<bool(int, std::vector<char> &)> place = [&](int c, std::vector<char> &seen) -> bool {
                for (auto &e : cand[c]) {
                    if (seen[e.first]) continue;
                    seen[e.first] = 1;
                    if (owner[e.first] < 0 || place(owner[e.first], seen)) {
                        owner[e.first] = c; owner_start[e.first] = e.second;
                        return true;
                    }
                }
                return false;
            };
            bool perfect = nlog <= 64;
            for (int c = 0; c < nlog && perfect; ++c) {
                std::vector<char> seen(64, 0);
                perfect = place(c, seen);
            }
            if (perfect) {
                for (int sl = 0; sl < 64; ++sl)
                    if (owner[sl] >= 0) start_of[owner[sl]] = owner_start[sl];
            } else {                                                           // keep the natural order (correct, just slower)
                std::fill(owner.begin(), owner.end(), -1);
                for (int c = 0; c < nlog && c < 64; ++c) { start_of[c] = logical[c].a - logical[c].a % align; owner[c] = c; }
            }
        }
        // per-lane tables (always 64 lanes): {band, start, chunk id, 0} and chp weights; idle lanes take the unused chunk ids
        T.chunks.assign(64, make_int4(0, 0, 0, 0));
        T.w.assign((size_t)64 * chp, 0.f);
        int spare = nlog;
        for (int lane = 0; lane < 64; ++lane) {
            const int c = owner[lane];
            if (c < 0) { T.chunks[lane] = make_int4(0, 0, spare < 64 ? spare++ : 63, 0); continue; }
            T.chunks[lane] = make_int4(logical[c].band, start_of[c], c, 0);
            for (int t = 0; t < chp; ++t) {
                const int pos = start_of[c] + t;
                if (pos >= logical[c].a && pos < logical[c].a + logical[c].len && pos < npos && bin_of[pos] >= 0)
                    T.w[(size_t)lane * chp + t] = (float)((double)(float)bank[(size_t)logical[c].band * n_bins + bin_of[pos]] * wscale);
            }
        }
        T.ch = ch; T.chp = chp; T.nlog = nlog; T.bcs = bcs;
        return T;
    };
    const ChunkTables T1 = build_chunks(1);
    const bool v3_geom = p->n_fft == 1024;                                   // the position map is the 513-bin one
    const ChunkTables T3 = build_chunks(2, v3_geom);
    const int chp = T1.chp, nlog = T1.nlog, n_filt_pad = (n_filt + 3) & ~3;
    const std::vector<int4> &chunks = T1.chunks;
    const std::vector<float> &w = T1.w;
    const std::vector<int> &bcs = T1.bcs;

    std::vector<float2> tw1(7 * 64), tw2(7 * 8), tws(257);
    for (int k = 1; k < 8; ++k)
        for (int l = 0; l < 64; ++l) {
            const double a = -2.0 * M_PI * (double)(l * k) / 512.0;
            tw1[(k - 1) * 64 + l] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
    for (int k = 1; k < 8; ++k)
        for (int l = 0; l < 8; ++l) {
            const double a = -2.0 * M_PI * (double)(l * k) / 64.0;
            tw2[(k - 1) * 8 + l] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
    for (int k = 0; k <= 256; ++k) {
        const double a = -2.0 * M_PI * (double)k / 1024.0;
        tws[k] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    std::vector<float2> tw1s(7 * 64), tws3(4 * 64);
    for (int k = 1; k < 8; ++k)
        for (int l = 0; l < 64; ++l) {
            const double a = -2.0 * M_PI * (double)(v3_sigma(l) * k) / 512.0;
            tw1s[(k - 1) * 64 + l] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
    for (int i = 0; i < 4; ++i)
        for (int l = 0; l < 64; ++l) {
            const double a = -2.0 * M_PI * (double)(l + 64 * i) / 1024.0;
            tws3[i * 64 + l] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
    // generic-path tables (used when n_fft != 1024)
    std::vector<float2> twg((size_t)p->n_fft / 2);
    for (int k = 0; k < p->n_fft / 2; ++k) {
        const double a = -2.0 * M_PI * (double)k / (double)p->n_fft;
        twg[k] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    std::vector<float> bwf(bank.size());
    for (size_t i = 0; i < bank.size(); ++i) bwf[i] = (float)bank[i];
    std::vector<float> dct((size_t)n_filt_pad * n_out, 0.f);
    for (int n = 0; n < n_filt; ++n)
        for (int k = 0; k < n_out; ++k)  // scipy dct type II norm='ortho' (bark_feature.py:172, mfcc.h:55-67)
            dct[(size_t)n * n_out + k] = (float)(std::cos(M_PI * (n + 0.5) * k / n_filt) * (k == 0 ? std::sqrt(1.0 / n_filt) : std::sqrt(2.0 / n_filt)));

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        return fail(KWS_ERR_HIP, "no HIP device: the featurizer has no CPU fallback");
    }

    auto *f = new kws_featurizer();
    f->params = *p; f->geom = g; f->bank_kind = bank_kind;
    f->bank.resize(bank.size());
    for (size_t i = 0; i < bank.size(); ++i) f->bank[i] = (float)bank[i];

    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_tw1 = 0, o_tw2 = al(o_tw1 + tw1.size() * 8), o_tws = al(o_tw2 + tw2.size() * 8),
                 o_ch = al(o_tws + tws.size() * 8), o_bcs = al(o_ch + std::max<size_t>(1, chunks.size()) * 16),
                 o_w = al(o_bcs + bcs.size() * 4), o_dct = al(o_w + std::max<size_t>(1, w.size()) * 4),
                 o_twg = al(o_dct + dct.size() * 4), o_bf = al(o_twg + twg.size() * 8), o_bwd = al(o_bf + first.size() * 4),
                 o_bw = al(o_bwd + width.size() * 4), o_ch3 = al(o_bw + bwf.size() * 4),
                 o_bcs3 = al(o_ch3 + 64 * 16), o_w3 = al(o_bcs3 + T3.bcs.size() * 4), o_tw1s = al(o_w3 + T3.w.size() * 4),
                 o_tws3 = al(o_tw1s + tw1s.size() * 8), total = al(o_tws3 + tws3.size() * 8);
    std::vector<unsigned char> host(total, 0);
    std::memcpy(host.data() + o_tw1, tw1.data(), tw1.size() * 8);
    std::memcpy(host.data() + o_tw2, tw2.data(), tw2.size() * 8);
    std::memcpy(host.data() + o_tws, tws.data(), tws.size() * 8);
    if (!chunks.empty()) std::memcpy(host.data() + o_ch, chunks.data(), chunks.size() * 16);
    std::memcpy(host.data() + o_bcs, bcs.data(), bcs.size() * 4);
    if (!w.empty()) std::memcpy(host.data() + o_w, w.data(), w.size() * 4);
    std::memcpy(host.data() + o_dct, dct.data(), dct.size() * 4);
    std::memcpy(host.data() + o_twg, twg.data(), twg.size() * 8);
    std::memcpy(host.data() + o_bf, first.data(), first.size() * 4);
    std::memcpy(host.data() + o_bwd, width.data(), width.size() * 4);
    std::memcpy(host.data() + o_bw, bwf.data(), bwf.size() * 4);
    std::memcpy(host.data() + o_ch3, T3.chunks.data(), 64 * 16);
    std::memcpy(host.data() + o_bcs3, T3.bcs.data(), T3.bcs.size() * 4);
    std::memcpy(host.data() + o_w3, T3.w.data(), T3.w.size() * 4);
    std::memcpy(host.data() + o_tw1s, tw1s.data(), tw1s.size() * 8);
    std::memcpy(host.data() + o_tws3, tws3.data(), tws3.size() * 8);
    hipError_t e = hipMalloc(&f->dmem, total);
    if (e == hipSuccess) e = hipMemcpy(f->dmem, host.data(), total, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (f->dmem) (void)hipFree(f->dmem);
        delete f;
        return fail(KWS_ERR_HIP, "featurizer table upload failed: %s", hipGetErrorString(e));
    }
    auto *base = static_cast<unsigned char *>(f->dmem);
    FeatDev &d = f->dev;
    d.window_eff = std::min(g.window_samples, (int)p->n_fft);  // np.fft.rfft(frames, n): crop or zero-pad (bark_feature.py:87)
    d.hop = g.hop_samples; d.max_samples = g.max_samples; d.n_frames = n_frames;
    d.n_filt = n_filt; d.n_out = n_out; d.feature_size = g.feature_size; d.use_delta = p->use_delta ? 1 : 0;
    d.nchunks = nlog; d.nnz = (int)w.size(); d.inv_nfft = 1.0f / (float)p->n_fft;
    d.chp = chp; d.n_filt_pad = n_filt_pad;
    d.tail_batch = std::max(1, std::min(4, 64 / std::max(n_filt_pad, n_out)));
    d.tw1 = reinterpret_cast<const float2 *>(base + o_tw1);
    d.tw2 = reinterpret_cast<const float2 *>(base + o_tw2);
    d.tws = reinterpret_cast<const float2 *>(base + o_tws);
    d.chunks = reinterpret_cast<const int4 *>(base + o_ch);
    d.bcs = reinterpret_cast<const int *>(base + o_bcs);
    d.w = reinterpret_cast<const float *>(base + o_w);
    d.dct = reinterpret_cast<const float *>(base + o_dct);
    d.n_fft = p->n_fft;
    d.log2n = 0;
    while ((1 << d.log2n) < p->n_fft) ++d.log2n;
    d.twg = reinterpret_cast<const float2 *>(base + o_twg);
    d.bfirst = reinterpret_cast<const int *>(base + o_bf);
    d.bwidth = reinterpret_cast<const int *>(base + o_bwd);
    d.bw = reinterpret_cast<const float *>(base + o_bw);
    d.chp3 = (v3_geom && T3.nlog <= 64) ? T3.chp : 0;
    d.chunks3 = reinterpret_cast<const int4 *>(base + o_ch3);
    d.bcs3 = reinterpret_cast<const int *>(base + o_bcs3);
    d.w3 = reinterpret_cast<const float *>(base + o_w3);
    d.tw1s = reinterpret_cast<const float2 *>(base + o_tw1s);
    d.tws3 = reinterpret_cast<const float2 *>(base + o_tws3);
    // the LDS the launch will ask for (same job shape as launch_featurize / launch_generic, same 160 KiB limit)
    if (d.n_fft == 1024) {
        feat_job_shape(d);
        f->smem_bytes = feat_smem_bytes(d);
    } else {
        f->smem_bytes = generic_smem_bytes(d);
    }
    if (f->smem_bytes > kMaxLdsBytes) {
        const size_t need = f->smem_bytes;
        (void)hipFree(f->dmem);
        delete f;
        return fail(KWS_ERR_UNSUPPORTED, "featurizer needs %zu B of LDS per block (> %d KiB)", need, (int)(kMaxLdsBytes / 1024));
    }
    *out = f;
    return KWS_OK;
}

void kws_featurizer_destroy(kws_featurizer *f)
{
    if (!f) return;
    if (f->dmem) (void)hipFree(f->dmem);
    delete f;
}

int kws_featurizer_geometry(const kws_featurizer *f, kws_geometry *g)
{
    if (!f || !g) return fail(KWS_ERR_INVALID, "null argument");
    *g = f->geom;
    return KWS_OK;
}

int kws_featurizer_bank(const kws_featurizer *f, float *host_bank, size_t count)
{
    if (!f || !host_bank) return fail(KWS_ERR_INVALID, "null argument");
    if (count != f->bank.size()) return fail(KWS_ERR_INVALID, "bank has %zu floats, caller asked for %zu", f->bank.size(), count);
    std::memcpy(host_bank, f->bank.data(), count * sizeof(float));
    return KWS_OK;
}

static int launch_generic(const FeatDev &d, const void *wav, int wav_dtype, int B, int64_t stride, const int32_t *valid_len,
                          float *feat, void *stream)
{
    const size_t smem = generic_smem_bytes(d);
    if (smem > kMaxLdsBytes) return fail(KWS_ERR_UNSUPPORTED, "n_fft=%d with %d frames needs %zu B of LDS (> 160 KiB)", d.n_fft, d.n_frames, smem);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)B), block(kGenWaves * 64);
    if (wav_dtype == KWS_WAV_F32) {
        if (smem > 64 * 1024)
            KWS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&featurize_generic_kernel<float>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        KWS_LAUNCH("featurize_generic_f32", featurize_generic_kernel<float>, grid, block, smem, s, static_cast<const float *>(wav), stride,
                   valid_len, B, d, feat);
    } else if (wav_dtype == KWS_WAV_I16) {
        if (smem > 64 * 1024)
            KWS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&featurize_generic_kernel<short>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        KWS_LAUNCH("featurize_generic_i16", featurize_generic_kernel<short>, grid, block, smem, s, static_cast<const short *>(wav), stride,
                   valid_len, B, d, feat);
    } else {
        return fail(KWS_ERR_INVALID, "unknown wav dtype %d", wav_dtype);
    }
    KWS_LAUNCH_CHECK("featurize_generic_kernel");
    return KWS_OK;
}

static int launch_featurize(const FeatDev &d0, const void *wav, int wav_dtype, int B, int64_t stride,
                            const int32_t *valid_len, float *feat, void *stream)
{
    if (d0.n_fft != 1024) return launch_generic(d0, wav, wav_dtype, B, stride, valid_len, feat, stream);
    FeatDev d = d0;
    feat_job_shape(d);
    if (v3_applies(d)) {
        if (wav_dtype == KWS_WAV_F32)
            return launch_v3_any(d, static_cast<const float *>(wav), B, stride, valid_len, feat, static_cast<hipStream_t>(stream), "featurize_fft1024_f32");
        if (wav_dtype == KWS_WAV_I16)
            return launch_v3_any(d, static_cast<const short *>(wav), B, stride, valid_len, feat, static_cast<hipStream_t>(stream), "featurize_fft1024_i16");
        return fail(KWS_ERR_INVALID, "unknown wav dtype %d", wav_dtype);
    }
    const size_t smem = feat_smem_bytes(d);
    if (smem > kMaxLdsBytes) return fail(KWS_ERR_UNSUPPORTED, "%d frames need %zu B of LDS (> 160 KiB)", d.n_frames, smem);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long jobs = (long)B * d.jpc;
    const dim3 grid((unsigned)(d.use_delta ? B : (jobs + kWaves - 1) / kWaves)), block(kThreads);
    if (wav_dtype == KWS_WAV_F32) {
        if (smem > 64 * 1024)
            KWS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&featurize_fft1024_kernel<float>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        KWS_LAUNCH("featurize_fft1024_f32", featurize_fft1024_kernel<float>, grid, block, smem, s, static_cast<const float *>(wav), stride,
                           valid_len, B, d, feat);
    } else if (wav_dtype == KWS_WAV_I16) {
        if (smem > 64 * 1024)
            KWS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&featurize_fft1024_kernel<short>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        KWS_LAUNCH("featurize_fft1024_i16", featurize_fft1024_kernel<short>, grid, block, smem, s, static_cast<const short *>(wav), stride,
                           valid_len, B, d, feat);
    } else {
        return fail(KWS_ERR_INVALID, "unknown wav dtype %d", wav_dtype);
    }
    KWS_LAUNCH_CHECK("featurize_fft1024_kernel");
    return KWS_OK;
}

int kws_featurize_gather(kws_featurizer *f, const void *wav, int wav_dtype, const int32_t *index, int B, int64_t stride,
                         const int32_t *valid_len, float *feat, void *stream)
{
    if (!f || !feat || (!wav && B > 0)) return fail(KWS_ERR_INVALID, "null argument");
    if (B < 0 || stride < 0) return fail(KWS_ERR_INVALID, "negative batch or stride");
    if (B == 0) return KWS_OK;
    if (!valid_len && stride < 1) return fail(KWS_ERR_INVALID, "stride must be >= 1 when valid_len is NULL");
    FeatDev d = f->dev;
    d.blocks_per_cu = f->blocks_per_cu;
    d.index = index;
    return launch_featurize(d, wav, wav_dtype, B, stride, valid_len, feat, stream);
}

int kws_featurize(kws_featurizer *f, const void *wav, int wav_dtype, int B, int64_t stride, const int32_t *valid_len,
                  float *feat, void *stream)
{
    return kws_featurize_gather(f, wav, wav_dtype, nullptr, B, stride, valid_len, feat, stream);
}

int kws_featurizer_set_cu_share(kws_featurizer *f, int blocks_per_cu)
{
    if (!f) return fail(KWS_ERR_INVALID, "null argument");
    if (blocks_per_cu != 1 && blocks_per_cu != 2) return fail(KWS_ERR_INVALID, "blocks_per_cu must be 1 or 2");
    f->blocks_per_cu = blocks_per_cu;
    return KWS_OK;
}

int kws_featurizer_occupancy(const kws_featurizer *f, int *blocks_per_cu, size_t *lds_bytes)
{
    if (!f || !blocks_per_cu) return fail(KWS_ERR_INVALID, "null argument");
    const bool tuned = f->dev.n_fft == 1024;
    const size_t smem = f->smem_bytes;
    if (lds_bytes) *lds_bytes = smem;
    int nb = 0;
    if (tuned)
        KWS_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, featurize_fft1024_kernel<float>, kThreads, smem));
    else
        KWS_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, featurize_generic_kernel<float>, kGenWaves * 64, smem));
    *blocks_per_cu = nb;
    return KWS_OK;
}

int kws_featurize_raw_frames(const kws_featurizer *f, int32_t n_samples)
{
    if (!f || n_samples < f->geom.window_samples) return 0;  // chop_array, bark_feature.py:80-82
    return (n_samples - f->geom.window_samples) / f->geom.hop_samples + 1;
}

int kws_featurize_raw(kws_featurizer *f, const void *wav, int wav_dtype, int B, int64_t stride, int32_t n_samples,
                      float *feat, void *stream)
{
    if (!f || !feat || (!wav && B > 0)) return fail(KWS_ERR_INVALID, "null argument");
    if (B < 0 || n_samples < 0 || stride < n_samples) return fail(KWS_ERR_INVALID, "bad batch / n_samples / stride");
    FeatDev d = f->dev;
    d.n_frames = kws_featurize_raw_frames(f, n_samples);
    if (B == 0 || d.n_frames == 0) return KWS_OK;  // sonopy returns an empty matrix
    d.max_samples = n_samples;
    d.use_delta = 0;
    d.feature_size = d.n_out;
    d.blocks_per_cu = f->blocks_per_cu;
    return launch_featurize(d, wav, wav_dtype, B, stride, nullptr, feat, stream);
}

}  // extern "C"
