// csrc/kws_featurize_v2.h -- the tuned featurizer kernel for the default frame geometry (n_fft = window = 1024, hop = 512)
// with compile-time band / coefficient counts.  Included by kws_featurize.hip behind the first-generation kernel, whose
// helpers (dft8, cmul, load_half, wave_sync, FeatDev) it shares; arithmetic and results are the same (sonopy.mfcc_spec as
// restated in common/bark_feature.py:75-89,156-175 and inference/tflite/mfcc.h:214-290).
//
// What changed against featurize_fft1024_kernel, each from a measurement:
//   * no global loads inside the frame loop except the samples: the real-FFT split's twiddles W_1024^k come from LDS (they
//     were L1 hits whose in-order vmcnt wait also waited for the next frame's sample prefetch);
//   * the third LDS round trip of the FFT is gone: after pass 3 lane (k1, k2a) holds Z[ka + 64 r], ka = k1 + 8 k2a, in
//     register r, and the partner bins Z[512 - k] of the real-FFT split sit in lane (64 - ka) mod 64, register 7 - r -- a
//     FIXED lane permutation, fetched with 8 ds_bpermute_b32 (crossbar only, no LDS array write + read, no address arithmetic;
//     lane ka = 0 is its own partner with registers rotated by one, fixed up by 8 selects before the exchange);
//   * band gather on ds_read_b64: the host places every chunk on an even bin (zero weights in front), so a lane reads two
//     power-spectrum bins per LDS instruction;
//   * band count, coefficient count, chunk length and tail batch are template constants: no loop-carried trip counts,
//     address multiplies or masked tails in the log / DCT stage.
#pragma once

namespace kws {

// lane (hi = lane >> 3, lo = lane & 7) holds the FFT outputs ka + 64 r after pass 3
__host__ __device__ inline int v2_ka(int lane) { return (lane >> 3) + 8 * (lane & 7); }

// Explicit single-width LDS reads.  The compiler merges neighbouring ds_read_b64 into ds_read2_b64 / ds_read2st64_b64, which the
// LDS serves at HALF the rate (8 array cycles for two reads against 2 + 2, MI355X_MICROARCH.md section LDS), and this kernel
// is bound by LDS cycles (74 % LDS-array utilisation, 31 % of wave time waiting to ISSUE an LDS instruction).  The reads of the
// frame loop are therefore written out; lds_wait() ties the loaded registers to the s_waitcnt, so no use is scheduled above it.
// The compiler's own lgkmcnt bookkeeping does not see these reads, which only makes its waits stricter (LDS returns in order).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}
template <int OFF> __device__ __forceinline__ f32x2 lds_rd64(unsigned addr)
{
    f32x2 d;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
    return d;
}
template <int OFF> __device__ __forceinline__ f32x4 lds_rd128(unsigned addr)
{
    f32x4 d;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
    return d;
}
// PENDING = LDS operations issued after the ones waited for (they may stay in flight)
template <int PENDING = 0> __device__ __forceinline__ void lds_wait(f32x2 (&a)[8])
{
    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "n"(PENDING) : "memory");
}
__device__ __forceinline__ void lds_wait(f32x2 (&a)[7])
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6])::"memory");
}
__device__ __forceinline__ void lds_wait(f32x2 (&a)[4])
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])::"memory");
}
template <int N, int STRIDE> struct LdsRow {            // N reads of 8 bytes at addr + i * STRIDE
    template <int I> static __device__ __forceinline__ void go(f32x2 (&d)[N], unsigned addr)
    {
        if constexpr (I < N) {
            d[I] = lds_rd64<I * STRIDE>(addr);
            go<I + 1>(d, addr);
        }
    }
};

// band gather of one lane: CHP power-spectrum bins from an even bin (two per ds_read_b64) times CHP weights (four per
// ds_read_b128); all loads of the chunk are issued before the first use
template <int CHP> __device__ __forceinline__ void gather_chunk(float &part, unsigned a_pw, unsigned a_w);
#define KWS_GATHER_BODY(NQ)                                                                                         \
    f32x2 p[2 * NQ];                                                                                                \
    f32x4 w[NQ];                                                                                                    \
    gather_loads<NQ, 0>(p, w, a_pw, a_w);                                                                           \
    gather_wait<NQ>(p, w);                                                                                          \
    _Pragma("unroll") for (int t = 0; t < NQ; ++t) {                                                               \
        part = fmaf(p[2 * t].x, w[t].x, part);                                                                      \
        part = fmaf(p[2 * t].y, w[t].y, part);                                                                      \
        part = fmaf(p[2 * t + 1].x, w[t].z, part);                                                                  \
        part = fmaf(p[2 * t + 1].y, w[t].w, part);                                                                  \
    }
template <int NQ, int T> __device__ __forceinline__ void gather_loads(f32x2 (&p)[2 * NQ], f32x4 (&w)[NQ], unsigned a_pw, unsigned a_w)
{
    if constexpr (T < NQ) {
        w[T] = lds_rd128<16 * T>(a_w);
        p[2 * T] = lds_rd64<16 * T>(a_pw);
        p[2 * T + 1] = lds_rd64<16 * T + 8>(a_pw);
        gather_loads<NQ, T + 1>(p, w, a_pw, a_w);
    }
}
template <int NQ> __device__ __forceinline__ void gather_wait(f32x2 (&p)[2 * NQ], f32x4 (&w)[NQ]);
template <> __device__ __forceinline__ void gather_wait<3>(f32x2 (&p)[6], f32x4 (&w)[3])
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(w[0]), "+v"(w[1]), "+v"(w[2])::"memory");
}
template <> __device__ __forceinline__ void gather_wait<4>(f32x2 (&p)[8], f32x4 (&w)[4])
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]), "+v"(w[0]), "+v"(w[1]),
                 "+v"(w[2]), "+v"(w[3])::"memory");
}
template <> __device__ __forceinline__ void gather_wait<5>(f32x2 (&p)[10], f32x4 (&w)[5])
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]), "+v"(p[8]), "+v"(p[9]),
                 "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4])::"memory");
}
template <> __device__ __forceinline__ void gather_chunk<12>(float &part, unsigned a_pw, unsigned a_w) { KWS_GATHER_BODY(3) }
template <> __device__ __forceinline__ void gather_chunk<16>(float &part, unsigned a_pw, unsigned a_w) { KWS_GATHER_BODY(4) }
template <> __device__ __forceinline__ void gather_chunk<20>(float &part, unsigned a_pw, unsigned a_w) { KWS_GATHER_BODY(5) }
#undef KWS_GATHER_BODY

constexpr int kV2Waves = 12;          // waves per block: 2 blocks per CU = 24 waves = 6 per SIMD (<= 80 VGPRs), the filter / twiddle tables twice per CU

// WAVES: waves per block.  TWREG: the per-lane twiddles of the three passes and of the real-FFT split (7 + 7 + 4 complex values that
// never change for a lane) live in 36 registers instead of being re-read from LDS for every frame -- 18 of the ~87 LDS instructions
// per frame, in a kernel that is bound by LDS issue.  It costs occupancy (116 instead of 80 registers: 4 instead of 6 waves per SIMD),
// so which form runs is the launcher's choice (kws_featurize.hip: launch_v2).
template <typename WavT, int CHP, int NF, int NO, int WAVES = kV2Waves, bool TWREG = false>
__global__ __launch_bounds__(WAVES * 64, TWREG ? 4 : 6) void featurize_fft1024_v2_kernel(const WavT *__restrict__ wav, int64_t stride,
                                                                                  const int32_t *__restrict__ valid_len, int B,
                                                                                  FeatDev c, float *__restrict__ feat)
{
    static_assert(NF % 4 == 0 && NF <= 32 && NO <= NF && CHP % 4 == 0, "band / coefficient counts of the tuned kernel");
    constexpr int TB = 64 / NF;                       // frames per tail batch (lanes = frame x band)
    constexpr int kPerWave = TB * 64 + 64 + 4;        // floats: chunk partial sums [TB][64], band logs [TB][NF] (64), energies [TB]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int kWaves = WAVES, kThreads = WAVES * 64;

    float2 *s_fft = reinterpret_cast<float2 *>(smem) + wave * kFftTile;
    float *s_pw = reinterpret_cast<float *>(s_fft);                       // power spectrum aliases the FFT tile
    float *s_part = reinterpret_cast<float *>(smem + kWaves * kFftTile * 8) + wave * kPerWave;
    float *s_mel = s_part + TB * 64;
    float *s_en = s_mel + 64;
    float *s_dct = reinterpret_cast<float *>(smem + kWaves * kFftTile * 8) + kWaves * kPerWave;   // [NF][NO]
    float *s_w = s_dct + NF * NO;                                          // [64 lanes][CHP]
    float2 *s_tw1 = reinterpret_cast<float2 *>(s_w + 64 * CHP);           // [7][64]
    float2 *s_tw2 = s_tw1 + 7 * 64;                                        // [7][8]
    float2 *s_tws = s_tw2 + 7 * 8;                                         // [4][64]  W_1024^(ka(lane) + 64 i)
    int *s_bcs = reinterpret_cast<int *>(s_tws + 4 * 64);                  // [NF + 1]

    for (int i = tid; i < NF * NO; i += kThreads) s_dct[i] = c.dct[i];
    for (int i = tid; i < 64 * CHP; i += kThreads) s_w[i] = c.w2[i];
    for (int i = tid; i < 7 * 64; i += kThreads) s_tw1[i] = c.tw1[i];
    for (int i = tid; i < 7 * 8; i += kThreads) s_tw2[i] = c.tw2[i];
    for (int i = tid; i < 4 * 64; i += kThreads) s_tws[i] = c.tws2[i];
    for (int i = tid; i <= NF; i += kThreads) s_bcs[i] = c.bcs2[i];
    s_mel[lane] = 0.f;
    __syncthreads();

    const int hi = lane >> 3, lo = lane & 7;
    const int ka = hi + 8 * lo;                                            // this lane's FFT outputs are Z[ka + 64 r]
    const int kp = (64 - ka) & 63;                                         // partner outputs Z[512 - k] live in the lane holding kp
    const int partner = 4 * (((kp & 7) << 3) | (kp >> 3));                 // ds_bpermute byte address of that lane
    const bool lane0 = lane == 0;
    const int chunk_pack = c.chunks2[lane].y | (c.chunks2[lane].z << 16);  // first bin read (even) | slot of the partial sum

    // LDS byte addresses of this lane's reads (constant for the whole kernel)
    const unsigned a_tw1 = lds_addr(s_tw1 + lane), a_tw2 = lds_addr(s_tw2 + lo), a_tws = lds_addr(s_tws + lane);
    const unsigned a_x1 = lds_addr(s_fft + 72 * hi + lo), a_x2 = lds_addr(s_fft + 72 * hi + 9 * lo);
    const unsigned a_pw = lds_addr(s_pw + (chunk_pack & 0xFFFF)), a_w = lds_addr(s_w + lane * CHP);

    f32x2 tw1r[TWREG ? 7 : 1], tw2r[TWREG ? 7 : 1], twsr[TWREG ? 4 : 1];
    if constexpr (TWREG) {
        LdsRow<7, 64 * 8>::template go<0>(tw1r, a_tw1);
        LdsRow<7, 8 * 8>::template go<0>(tw2r, a_tw2);
        LdsRow<4, 64 * 8>::template go<0>(twsr, a_tws);
        lds_wait(tw1r);
        lds_wait(tw2r);
        lds_wait(twsr);
    }
    // Persistent waves: a JOB = c.fpw consecutive frames of one clip; the jobs of the batch are dealt round-robin to the
    // grid's waves (the tables above are loaded once per block, not once per clip, and no block waits for a launch slot).
    const int njobs = B * c.jpc;
  for (int job = (int)blockIdx.x * kWaves + wave; job < njobs; job += (int)gridDim.x * kWaves) {
    const int b = job / c.jpc;
    // clip geometry: keep the head, left-pad zeros (data_utils.py:77-80)
    const int bc = b;
    int len = valid_len ? valid_len[bc] : (stride > c.max_samples ? c.max_samples : (int)stride);
    len = len < 0 ? 0 : len;
    if ((int64_t)len > stride) len = (int)stride;
    if (len > c.max_samples) len = c.max_samples;
    const int pad = c.max_samples - len;
    const WavT *src = wav + (int64_t)bc * stride;
    const bool vec_ok = ((pad & 1) == 0) && ((reinterpret_cast<uintptr_t>(src) & (2 * sizeof(WavT) - 1)) == 0);

    float *dst = feat + (int64_t)bc * c.n_frames * NO;

    const int f_beg = (job - b * c.jpc) * c.fpw;
    const int f_end = f_beg + c.fpw < c.n_frames ? f_beg + c.fpw : c.n_frames;

    float2 xl[4], xh[4];                // lower / upper half of the next frame to transform
    if (f_beg < f_end) {
        load_half<WavT, 0>(xl, src, f_beg * 512, pad, 1024, vec_ok, lane);
        load_half<WavT, 4>(xh, src, f_beg * 512, pad, 1024, vec_ok, lane);
    }
    int qi = 0;                          // frames waiting in this wave's tail batch
    for (int f = f_beg; f < f_end; ++f) {
        float2 v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = xl[j]; v[j + 4] = xh[j]; }
        if (f + 1 < f_end) {             // next frame: its lower half is this frame's upper half; the new half loads under the FFT
#pragma unroll
            for (int j = 0; j < 4; ++j) xl[j] = xh[j];
            load_half<WavT, 4>(xh, src, (f + 1) * 512, pad, 1024, vec_ok, lane);
        }
        // pass 1: DFT-8 over n1 (n = lane + 64 n1), twiddle W_512^(lane*k1)
        {
            f32x2 tw[7];
            if constexpr (!TWREG) LdsRow<7, 64 * 8>::template go<0>(tw, a_tw1);      // issued ahead of the butterflies that hide their latency
            dft8(v);
            if constexpr (TWREG) {
#pragma unroll
                for (int k = 0; k < 7; ++k) tw[k] = tw1r[k];
            } else lds_wait(tw);
#pragma unroll
            for (int k = 1; k < 8; ++k) v[k] = cmul(v[k], make_float2(tw[k - 1].x, tw[k - 1].y));
        }
        wave_sync();
#pragma unroll
        for (int k = 0; k < 8; ++k) s_fft[72 * k + lane] = v[k];
        wave_sync();
        {
            f32x2 t[8], tw[7];
            LdsRow<8, 8 * 8>::template go<0>(t, a_x1);         // s_fft[72 hi + lo + 8 j]
            if constexpr (TWREG) lds_wait<0>(t);
            else {
                LdsRow<7, 8 * 8>::template go<0>(tw, a_tw2);   // s_tw2[(k - 1) * 8 + lo]
                lds_wait<7>(t);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = make_float2(t[j].x, t[j].y);
            // pass 2: lane = (k1, l2); DFT-8 over l1, twiddle W_64^(l2*k2a)
            dft8(v);
            if constexpr (TWREG) {
#pragma unroll
                for (int k = 0; k < 7; ++k) tw[k] = tw2r[k];
            } else lds_wait(tw);
#pragma unroll
            for (int k = 1; k < 8; ++k) v[k] = cmul(v[k], make_float2(tw[k - 1].x, tw[k - 1].y));
        }
        wave_sync();
#pragma unroll
        for (int k = 0; k < 8; ++k) s_fft[72 * hi + 9 * k + lo] = v[k];
        wave_sync();
        f32x2 tws[4];
        {
            f32x2 t[8];
            LdsRow<8, 8>::template go<0>(t, a_x2);             // s_fft[72 hi + 9 lo + j]
            if constexpr (TWREG) lds_wait<0>(t);
            else {
                LdsRow<4, 64 * 8>::template go<0>(tws, a_tws); // W_1024^(ka + 64 i), used by the split below
                lds_wait<4>(t);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = make_float2(t[j].x, t[j].y);
        }
        // pass 3: lane = (k1, k2a); DFT-8 over l2 -> register r holds Z[ka + 64 r]
        dft8(v);

        // partner bins Z[512 - (ka + 64 i)] = Z[kp + 64 (7 - i)], i = 0..3: register 7 - i of the partner lane; lane 0 (ka = 0)
        // is its own partner with Z[512 - 64 i] = Z[64 ((8 - i) & 7)]: rotate its four source registers by one
        float2 zm[4];
        {
            const float2 s7 = lane0 ? v[0] : v[7], s6 = lane0 ? v[7] : v[6], s5 = lane0 ? v[6] : v[5], s4 = lane0 ? v[5] : v[4];
            const float2 src4[4] = {s7, s6, s5, s4};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                zm[i].x = __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(src4[i].x)));
                zm[i].y = __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(src4[i].y)));
            }
        }
        // real-FFT split: X[k] = E[k] + W_1024^k O[k], X[512-k] = conj(E[k] - W_1024^k O[k])
        float pk[4], pm[4], energy = 0.f;
        if constexpr (TWREG) {
#pragma unroll
            for (int i = 0; i < 4; ++i) tws[i] = twsr[i];
        } else lds_wait(tws);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float2 zk = v[i], zq = zm[i];
            const float2 E = make_float2(0.5f * (zk.x + zq.x), 0.5f * (zk.y - zq.y));
            const float2 O = make_float2(0.5f * (zk.y + zq.y), -0.5f * (zk.x - zq.x));
            const float2 T = cmul(make_float2(tws[i].x, tws[i].y), O);
            const float2 xp = cadd(E, T), xm = csub(E, T);
            pk[i] = (xp.x * xp.x + xp.y * xp.y) * c.inv_nfft;   // bark_feature.py:88-89
            pm[i] = (xm.x * xm.x + xm.y * xm.y) * c.inv_nfft;
            energy += pk[i] + pm[i];
        }
        const float p256 = lane0 ? (v[4].x * v[4].x + v[4].y * v[4].y) * c.inv_nfft : 0.f;   // bin 256 is its own partner
        energy += p256;
        wave_sync();                        // the FFT tile is dead: the power spectrum takes its place
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s_pw[ka + 64 * i] = pk[i];
            s_pw[512 - ka - 64 * i] = pm[i];
        }
        if (lane0) s_pw[256] = p256;
        energy = wave_sum_dpp(energy);
        wave_sync();

        // sparse band gather: lane = one chunk (<= CHP bins from an even bin) of one band's non-zero span
        float part = 0.f;
        gather_chunk<CHP>(part, a_pw, a_w);
        s_part[qi * 64 + (chunk_pack >> 16)] = part;
        if (lane0) s_en[qi] = energy;
        ++qi;
        if (qi < TB && f + 1 < f_end) continue;

        // ---- tail of the qi frames f-qi+1 .. f: band sums -> log -> DCT, all frames of the batch at once ----
        const int f0 = f + 1 - qi;
        int tl = lane;                     // opaque copy: keeps the role arithmetic inside the tail (see the first-generation kernel)
        asm volatile("" : "+v"(tl));
        wave_sync();
        {
            const int bq = tl / NF, bm = tl - bq * NF;                        // lane = (frame in batch, band)
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
                s_mel[bq * NF + bm] = logf(fmaxf(sum, kEps));            // safe_log, bark_feature.py:75-77
            }
        }
        wave_sync();
        {
            const int dq = tl / NO, dn = tl - dq * NO;                        // lane = (frame in batch, coefficient)
            if (dq < qi) {
                const float4 *mq = reinterpret_cast<const float4 *>(s_mel + dq * NF);
                float sum = 0.f;
#pragma unroll
                for (int n = 0; n < NF / 4; ++n) {
                    const float4 mv = mq[n];
                    sum = fmaf(mv.x, s_dct[(4 * n) * NO + dn], sum);
                    sum = fmaf(mv.y, s_dct[(4 * n + 1) * NO + dn], sum);
                    sum = fmaf(mv.z, s_dct[(4 * n + 2) * NO + dn], sum);
                    sum = fmaf(mv.w, s_dct[(4 * n + 3) * NO + dn], sum);
                }
                if (dn == 0) sum = logf(fmaxf(s_en[dq], kEps));           // c0 <- log energy, bark_feature.py:173
                dst[f0 * NO + tl] = sum;                                  // rows f0.. are contiguous (feature_size == NO)
            }
        }
        qi = 0;
        wave_sync();
    }
  }
}

}  // namespace kws
