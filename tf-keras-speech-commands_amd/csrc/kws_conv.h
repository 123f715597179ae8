// csrc/kws_conv.h -- implicit-GEMM convolution / dense kernels on the fp32 MFMA (v_mfma_f32_16x16x4_f32).
//
// Everything GEMM-shaped in simple_cnn (Conv2D 3x3 'same', classifier/models/cnn.py:27-58; Dense 256->128, :64) is
// one of three products over NHWC activations and HWIO (Keras-order) weights:
//   FWD    z[m][n]   = sum_{tap,c} x[pix(m)+tap][c]      * W[tap][c][n]        m = output pixel, n = out channel
//   DGRAD  dx[m][n]  = sum_{tap,c} dz[pixT(m,tap)][c]    * W[tap][n][c]        m = input pixel,  n = in channel
//   WGRAD  dW[tap][c][n] = sum_m   x[pix(m)+tap][c]      * dz[m][n]
// fp32 MFMA is bit-exact fp32 (a k-ordered fmaf chain), so parity with the fp32/fp64 oracle needs no extra slack.
//
// Three kernels:
//   conv_gemm_kernel         FWD: 64 rows x all CO columns per block, A (gathered rows) and B (weights) tiles staged
//                            through LDS with the next chunk's global loads in flight during the MFMAs
//   conv_dgrad_direct_kernel DGRAD, LDS-free: float4 fragments straight from global memory, one launch per stride class
//   conv_wgrad_direct_kernel WGRAD, LDS-free: vector fragments with permuted tile maps, float atomics into grads
// LDS strides of the FWD kernel keep every half-wave ds_read_b32 of a fragment bank-conflict free
// (row stride == 2 mod 32 words for A[row][k], == 16 mod 32 for [k][col] tiles).
#pragma once
#include "kws_device.h"

namespace kws {

struct ConvGeom {
    int B, H, W, Ho, Wo;   // input and output spatial sizes (NHWC)
    int stride, pt, pl;    // TF 'SAME': pad_top / pad_left (the extra pad, if any, is bottom/right)
    int KH, KW;
    // conv_bf16_kernel only: rows in pixel-major order (row q = pixel q / B of clip q % B instead of clip-major), see there
    int pmajor = 0;
};

enum { EPI_NONE = 0, EPI_RELU = 1, EPI_BIAS_RELU6 = 2, EPI_BIAS = 3, EPI_BIAS_RELU = 4,
       EPI_BN_RELU6 = 5,     // inference: relu6(v * scale[n] + shift[n]), scale passed as `bias`, shift as `shift`
       // Data gradient whose consumer is the backward pass of the BatchNormalization -> ReLU6 IN FRONT of this convolution (no
       // pooling between them): the epilogue gates the produced gradient by that ReLU6 (y = z * scale + shift in (0, 6), z read
       // from `shift`, the layer's {scale, shift, mean, inv} rows of CO floats from `bias`), stores g, and with STATS leaves the
       // per-column partial sums of g and g * xhat -- the whole bn_bwd_reduce pass of that layer
       EPI_BNBWD_GATE6 = 6 };
enum { MODE_FWD = 0, MODE_DGRAD = 1 };

__host__ __device__ constexpr int stride16(int c) { return (c % 32 == 16) ? c : c + 16; }   // == 16 (mod 32)

// ---------------------------------------------------------------------------------------------------------------
// FWD.  CR = channels reduced per tap, CO = channels produced.
// ---------------------------------------------------------------------------------------------------------------
template <int CR, int CO, int MODE, int EPI>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const float *__restrict__ src, const float *__restrict__ wgt,
                                                         const float *__restrict__ bias, float *__restrict__ dst,
                                                         ConvGeom g)
{
    constexpr int KC = CR >= 32 ? 32 : CR;
    constexpr int SA = KC + 2;                 // A tile row stride (words): 2*row + q -> 32 distinct banks
    constexpr int SB = stride16(CO);
    constexpr int NT = CO / 16;
    constexpr int A4 = KC / 4;                 // float4 units per A row
    constexpr int NA = (64 * A4 + 255) / 256;
    constexpr int BUNITS = KC * CO / 4;
    constexpr int NB = (BUNITS + 255) / 256;
    constexpr int CPT = CR / KC;               // chunks per tap
    static_assert(CR % KC == 0 && CO % 16 == 0 && KC % 4 == 0, "channel counts must be multiples of 16");
    static_assert(MODE == MODE_FWD, "the data-gradient product has its own kernel (conv_dgrad_direct_kernel)");

    __shared__ __attribute__((aligned(16))) float As[64 * SA];
    __shared__ __attribute__((aligned(16))) float Bs[KC * SB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lq = lane >> 4;
    const int RH = g.Ho, RW = g.Wo;      // rows of this GEMM: output pixels
    const int SH = g.H, SW = g.W;        // source spatial size
    const long M = (long)g.B * RH * RW;
    const long m0 = (long)blockIdx.x * 64;
    const int ntaps = g.KH * g.KW, nchunks = ntaps * CPT;

    // per-thread A-staging coordinates (fixed over the K loop)
    int a_row[NA], a_c4[NA], a_b[NA], a_y[NA], a_x[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int u = tid + 256 * j;
        a_row[j] = u / A4;
        a_c4[j] = u % A4;
        const long m = m0 + a_row[j];
        if (u < 64 * A4 && m < M) {
            const int pix = (int)(m % ((long)RH * RW));
            a_b[j] = (int)(m / ((long)RH * RW));
            a_y[j] = pix / RW;
            a_x[j] = pix % RW;
        } else {
            a_b[j] = -1; a_y[j] = 0; a_x[j] = 0;
        }
    }

    float4 ra[NA], rb[NB];
    auto load_chunk = [&](int chunk) {
        const int tap = chunk / CPT, c0 = (chunk % CPT) * KC;
        const int kh = tap / g.KW, kw = tap % g.KW;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a_b[j] >= 0) {
                const int sy = a_y[j] * g.stride + kh - g.pt, sx = a_x[j] * g.stride + kw - g.pl;
                const bool ok = sy >= 0 && sy < SH && sx >= 0 && sx < SW;
                if (ok) v = *reinterpret_cast<const float4 *>(src + (((long)a_b[j] * SH + sy) * SW + sx) * CR + c0 + a_c4[j] * 4);
            }
            ra[j] = v;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int u = tid + 256 * j;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (u < BUNITS) {
                const int kk = u / (CO / 4), n4 = u % (CO / 4);
                v = *reinterpret_cast<const float4 *>(wgt + ((long)(tap * CR + c0 + kk)) * CO + n4 * 4);
            }
            rb[j] = v;
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int u = tid + 256 * j;
            if (u < 64 * A4) {
                float *p = &As[a_row[j] * SA + a_c4[j] * 4];      // 8-byte aligned: SA even
                *reinterpret_cast<float2 *>(p) = make_float2(ra[j].x, ra[j].y);
                *reinterpret_cast<float2 *>(p + 2) = make_float2(ra[j].z, ra[j].w);
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int u = tid + 256 * j;
            if (u < BUNITS) {
                const int kk = u / (CO / 4), n4 = u % (CO / 4);
                *reinterpret_cast<float4 *>(&Bs[kk * SB + n4 * 4]) = rb[j];
            }
        }
    };

    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    load_chunk(0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        __syncthreads();
        store_chunk();
        __syncthreads();
        if (chunk + 1 < nchunks) load_chunk(chunk + 1);
#pragma unroll
        for (int kk = 0; kk < KC; kk += 4) {
            const float a = As[(16 * wave + li) * SA + kk + lq];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = mfma16(a, Bs[(kk + lq) * SB + 16 * t + li], acc[t]);
        }
    }

#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = 16 * t + li;
        float bv = 0.f;
        if (EPI == EPI_BIAS_RELU6 || EPI == EPI_BIAS || EPI == EPI_BIAS_RELU) bv = bias[n];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long m = m0 + 16 * wave + 4 * lq + r;
            if (m < M) {
                float v = acc[t][r];
                if (EPI == EPI_RELU) v = fmaxf(v, 0.f);
                if (EPI == EPI_BIAS_RELU6) v = relu6f(v + bv);
                if (EPI == EPI_BIAS) v = v + bv;
                if (EPI == EPI_BIAS_RELU) v = fmaxf(v + bv, 0.f);
                dst[m * CO + n] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Split-precision bf16 (kws_device.h: mfma_bf16x6) form of the LDS-tiled product, for layers with >= 32 reduced channels.
//
// weight_split_kernel: once per step, every GEMM weight tensor W[tap][ci][co] (HWIO) becomes six bf16 planes
//   o[0..2] [tap][ci][co]  h / m / l in the ORIGINAL order   (k = co contiguous: the B operand of the data gradient)
//   t[0..2] [tap][co][ci]  h / m / l TRANSPOSED per tap      (k = ci contiguous: the B operand of the forward product)
// conv_bf16_kernel<CR, CO, MODE, EPI, RT>: 32*RT rows x all CO columns per block, K chunks of 32 channels (one MFMA depth).
//   The four waves form a 2 x 2 grid; a wave owns RT row tiles x CO/32 column tiles, so an A fragment read from LDS feeds
//   CO/32 and a B fragment RT tile products: the first version (one row tile x all columns per wave) re-read the whole B
//   tile in every wave and was LDS-bound (0.044 ms for conv4 with ALL global loads removed).
//   A: gathered activation rows, split into h/m/l while they are staged (once per element per block)
//   B: 16-byte copies of the prepared planes
//   LDS rows are 32 bf16 + 8 pad (80 B) (ds_read_b128 of a fragment: at most 2-way conflicts in its 16-lane groups).
//   Measured at B = 4096 (conv4 forward, RT = 3): 0.051 ms against 0.076 for the fp32 LDS-tiled kernel; phase ablation:
//   staging (split + LDS stores + 2 barriers per chunk) 0.022, MFMA phase 0.022 (= the matrix-pipe rate at 2 waves per
//   SIMD), global loads 0.007 -- the phases still run one after the other; overlapping them (double-buffered LDS or
//   direct-to-LDS loads) is the open step.
//   MODE_FWD   rows = output pixels, source (y*stride + kh - pt, x*stride + kw - pl), B = transposed planes of the layer
//   MODE_DGRAD rows = input pixels of a STRIDE-1 layer, source (y + pt - kh, x + pl - kw) in the output map, CR = the
//              layer's output channels, CO = its input channels, B = original-order planes
// ---------------------------------------------------------------------------------------------------------------
// frag: 0 = t planes transposed per tap (above); 1 / 2 = t planes FRAGMENT-MAJOR for the clip-group kernels (kws_infer_fused.h,
// kws_conv_group.h): element ((ks * (co / 16) + col / 16) * 64 + lane) * 8 + j with k-step ks = tap * (ci / 32) + c / 32, lane = col % 16 +
// 16 lq, and the channel c % 32 at (lq, j) = (unit % 4, c % 4 + 4 (unit / 4)), unit = (c % 32) / 4, for frag = 1 (A operand = fp32 rows) or
// (c % 32 / 8, c % 8) for frag = 2 (A operand = bf16 planes)
// ofrag: the o planes fragment-major for the clip-group DATA gradient (reduction over co, columns ci): element ((ks * (ci / 16) + c / 16) * 64 +
// lane) * 8 + j with ks = tap * (co / 32) + o / 32, lane = c % 16 + 16 ((o % 32) / 8), j = o % 8 for W[tap][c][o]
struct SplitDesc { const float *w; __bf16 *o[3], *t[3]; int taps, ci, co, frag, ofrag; };
struct SplitDescs { SplitDesc d[4]; };

// one x-slice (bx of nbx) of descriptor d
__device__ __forceinline__ void weight_split_slice(const SplitDesc &d, int bx, int nbx)
{
    const int per = d.ci * d.co, total = d.taps * per;
    for (int i = bx * 256 + threadIdx.x; i < total; i += nbx * 256) {
        const float v = d.w[i];
        const __bf16 h = (__bf16)v;
        const float r1 = v - (float)h;
        const __bf16 m = (__bf16)r1, l = (__bf16)(r1 - (float)m);
        const int tap = i / per, r = i - tap * per, ci = r / d.co, co = r - ci * d.co;
        int t = tap * per + co * d.ci + ci;
        if (d.frag) {
            const int ks = tap * (d.ci / 32) + ci / 32, c32 = ci & 31;
            const int lq = d.frag == 1 ? (c32 >> 2) & 3 : c32 >> 3, j = d.frag == 1 ? (c32 & 3) + 4 * (c32 >> 4) : c32 & 7;
            t = ((ks * (d.co / 16) + co / 16) * 64 + (co & 15) + 16 * lq) * 8 + j;
        }
        int io = i;
        if (d.ofrag) {
            const int ks = tap * (d.co / 32) + co / 32;
            io = ((ks * (d.ci / 16) + ci / 16) * 64 + (ci & 15) + 16 * ((co & 31) >> 3)) * 8 + (co & 7);
        }
        d.o[0][io] = h; d.o[1][io] = m; d.o[2][io] = l;
        d.t[0][t] = h; d.t[1][t] = m; d.t[2][t] = l;
    }
}

__global__ __launch_bounds__(256) void weight_split_kernel(SplitDescs all)
{
    weight_split_slice(all.d[blockIdx.y], blockIdx.x, gridDim.x);
}

struct Bf16Planes { const __bf16 *p[3]; };

// two blocks per CU (LDS: 54 KB at CO = 128): the 256-register budget keeps the 12 accumulator tiles, the 9 A and 3 B
// fragments and the staged chunk in VGPRs (at the default occupancy target the compiler spilled into the K loop)
// STATS: the epilogue also writes the BatchNorm partial sums of the block's outputs (sum and sum of squares per column, in
// double) to partial[(which*CO + n)*partial_stride + blockIdx.x], which replaces a separate pass over the conv output.
// APRE: the A operand arrives already split (planes `ap` in the layout of `src`, which is then unused): staging copies 16-byte
// pieces (one unit = 8 channels of a row) instead of splitting -- nine taps re-stage every row, so the split was done nine times.
// ABN: `src` is the PRE-activation tensor z of the BatchNormalization -> ReLU6 in front of this convolution and abn = its scale[CR] |
// shift[CR]: the activation a = relu6(z * scale + shift) is formed while the rows are staged (two vector instructions per element) and is
// never written to memory -- for a layer without pooling this replaces the activation kernel and its tensor.  Padding rows stay zero.
template <int CR, int CO, int MODE, int EPI, int RT, bool STATS = false, bool APRE = false, bool ABN = false>
__global__ __launch_bounds__(256, 2) void conv_bf16_kernel(const float *__restrict__ src, Bf16Planes wp, const float *__restrict__ bias,
                                                         float *__restrict__ dst, ConvGeom g, double *__restrict__ partial = nullptr,
                                                         int partial_stride = 0, const float *__restrict__ shift = nullptr,
                                                         Bf16Planes ap = Bf16Planes{{nullptr, nullptr, nullptr}},
                                                         const float *__restrict__ abn = nullptr)
{
    static_assert(!(ABN && APRE), "the activation is formed from fp32 rows");
    // chunk depth and LDS row stride in bf16 units.  96-byte rows: ds_read_b128 serves the lane groups {0-3,12-15,20-27}, ... (not
    // 16 consecutive lanes), and fragment reads at (row li, 16-byte piece lq) are conflict-free for strides of 16 B x (2 mod 4);
    // the 80-byte rows used before cost two LDS cycles per group (PMC: more conflict cycles than LDS instruction cycles)
    constexpr int KC = 32, SK = 48;
    constexpr int BM = 32 * RT;                // rows per block: RT row tiles per wave, 2 waves along M
    constexpr int NT = CO / 16, CT = NT / 2;   // column tiles per wave (2 waves along N)
    constexpr int CPT = CR / KC;
    constexpr int UPR = APRE ? 4 : 8;          // A units per row: 8 channels (16 B of a plane) or 4 channels (one float4)
    constexpr int NAU = BM * UPR / 256;        // A units per thread
    static_assert(BM * UPR % 256 == 0, "every thread stages the same number of A units");
    constexpr int NBU = CO * 4 / 256;          // 16-byte B units per thread and plane
    static_assert(CO * 4 % 256 == 0, "every thread stages the same number of B units");
    static_assert(CR % KC == 0 && CO % 32 == 0, "reduced channels must be a multiple of 32, produced ones of 32");
    __shared__ __attribute__((aligned(16))) __bf16 As[3][BM * SK], Bs[3][CO * SK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lq = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    // rows of this product and the map the rows are gathered from
    const int RH = MODE == MODE_FWD ? g.Ho : g.H, RW = MODE == MODE_FWD ? g.Wo : g.W;
    const int SH = MODE == MODE_FWD ? g.H : g.Ho, SW = MODE == MODE_FWD ? g.W : g.Wo;
    const long M = (long)g.B * RH * RW;
    const long m0 = (long)blockIdx.x * BM;
    const int ntaps = g.KH * g.KW;
    // Pixel-major rows (g.pmajor, small maps): the rows of a block then share ONE pixel position (two where a block straddles a
    // boundary), so the taps that fall into the padding for that position are known per block and their chunks are skipped
    // altogether -- on conv4's 3 x 2 map 48 % of all (pixel, tap) pairs are padding, on conv3's 4 x 3 output 35 %.  Clip-major rows
    // mix all positions in every block and multiply those zeros.  taps: the block's tap list, four bits each.
    const int P = RH * RW;
    const bool pm = g.pmajor != 0;
    auto row_of = [&](long q, int &b, int &pix) {      // logical row q -> (clip, pixel); M < 2^31 (checked by the launcher): 32-bit divisions
        const unsigned qq = (unsigned)q;
        if (pm) { pix = (int)(qq / (unsigned)g.B); b = (int)(qq - (unsigned)pix * (unsigned)g.B); }
        else { b = (int)(qq / (unsigned)P); pix = (int)(qq - (unsigned)b * (unsigned)P); }
    };
    unsigned long long taps = 0ull;
    int ntv = ntaps;
    if (pm) {
        const long qe = m0 + BM - 1 < M - 1 ? m0 + BM - 1 : M - 1;
        const int p_lo = (int)(m0 / g.B), p_hi = (int)(qe / g.B);
        ntv = 0;
        for (int tap = 0; tap < ntaps; ++tap) {
            const int kh = tap / g.KW, kw = tap % g.KW;
            bool any = false;
            for (int pp = p_lo; pp <= p_hi; ++pp) {
                const int y = pp / RW, x = pp % RW;
                const int sy = MODE == MODE_FWD ? y * g.stride + kh - g.pt : y + g.pt - kh;
                const int sx = MODE == MODE_FWD ? x * g.stride + kw - g.pl : x + g.pl - kw;
                any = any || (sy >= 0 && sy < SH && sx >= 0 && sx < SW);
            }
            if (any) { taps |= (unsigned long long)tap << (4 * ntv); ++ntv; }
        }
    }
    const int nchunks = ntv * CPT;

    int a_b[NAU], a_y[NAU], a_x[NAU];         // per-thread A-staging coordinates: row = u / UPR, piece = u % UPR
#pragma unroll
    for (int j = 0; j < NAU; ++j) {
        const int u = tid + 256 * j;
        const long m = m0 + u / UPR;
        if (m < M) {
            int pix;
            row_of(m, a_b[j], pix);
            a_y[j] = pix / RW;
            a_x[j] = pix % RW;
        } else {
            a_b[j] = -1; a_y[j] = 0; a_x[j] = 0;
        }
    }

    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // native vector type: HIP's uint4 struct kept these arrays in scratch
    struct Staged { f32x4 a[APRE ? 1 : NAU]; u32x4 ap[APRE ? 3 : 1][NAU]; u32x4 b[3][NBU]; unsigned ok; };
    // ABN: scale / shift of this thread's four channels in every chunk position (u % UPR = tid % UPR for all of its units)
    f32x4 abn_sc[ABN ? CPT : 1], abn_sh[ABN ? CPT : 1];
    if constexpr (ABN) {
#pragma unroll
        for (int cc = 0; cc < CPT; ++cc) {
            abn_sc[cc] = *reinterpret_cast<const f32x4 *>(abn + cc * KC + 4 * (tid % UPR));
            abn_sh[cc] = *reinterpret_cast<const f32x4 *>(abn + CR + cc * KC + 4 * (tid % UPR));
        }
    }
    auto load_chunk = [&](int chunk, Staged &st) {
        st.ok = 0u;
        const int ti = chunk / CPT, c0 = (chunk % CPT) * KC;
        const int tap = pm ? (int)((taps >> (4 * ti)) & 15ull) : ti;
        const int kh = tap / g.KW, kw = tap % g.KW;
#pragma unroll
        for (int j = 0; j < NAU; ++j) {
            const int u = tid + 256 * j;
            const int sy = MODE == MODE_FWD ? a_y[j] * g.stride + kh - g.pt : a_y[j] + g.pt - kh;
            const int sx = MODE == MODE_FWD ? a_x[j] * g.stride + kw - g.pl : a_x[j] + g.pl - kw;
            const bool ok = a_b[j] >= 0 && sy >= 0 && sy < SH && sx >= 0 && sx < SW;
            const long e = (((long)(ok ? a_b[j] : 0) * SH + (ok ? sy : 0)) * SW + (ok ? sx : 0)) * CR + c0;
            if (APRE) {
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    st.ap[p][j] = ok ? *reinterpret_cast<const u32x4 *>(ap.p[p] + e + (u % UPR) * 8) : (u32x4){0u, 0u, 0u, 0u};
            } else {
                st.a[j] = ok ? *reinterpret_cast<const f32x4 *>(src + e + (u % UPR) * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
                if (ABN && ok) st.ok |= 1u << j;
            }
        }
#pragma unroll
        for (int j = 0; j < NBU; ++j) {
            const int u = tid + 256 * j;
            const long e = ((long)(tap * CO + u / 4)) * CR + c0 + 8 * (u % 4);           // plane[tap][col][k]
#pragma unroll
            for (int p = 0; p < 3; ++p) st.b[p][j] = *reinterpret_cast<const u32x4 *>(wp.p[p] + e);
        }
    };
    auto store_chunk = [&](const Staged &st, int chunk) {
#pragma unroll
        for (int j = 0; j < NAU; ++j) {
            const int u = tid + 256 * j;
            if (APRE) {
#pragma unroll
                for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4 *>(&As[p][(u / UPR) * SK + 8 * (u % UPR)]) = st.ap[p][j];
            } else {
                f32x4 av = st.a[j];
                if constexpr (ABN) {
                    const int cc = chunk % CPT;
                    f32x4 sc = abn_sc[0], sh = abn_sh[0];
#pragma unroll
                    for (int q = 1; q < (ABN ? CPT : 1); ++q)
                        if (cc == q) { sc = abn_sc[q]; sh = abn_sh[q]; }
                    if (st.ok & (1u << j)) {
#pragma unroll
                        for (int e2 = 0; e2 < 4; ++e2) av[e2] = relu6f(fmaf(av[e2], sc[e2], sh[e2]));
                    }
                }
                bf16x4 h, m, l;
                split_bf16(av, h, m, l);
                const int o = (u / UPR) * SK + 4 * (u % UPR);
                *reinterpret_cast<bf16x4 *>(&As[0][o]) = h;
                *reinterpret_cast<bf16x4 *>(&As[1][o]) = m;
                *reinterpret_cast<bf16x4 *>(&As[2][o]) = l;
            }
        }
#pragma unroll
        for (int j = 0; j < NBU; ++j) {
            const int u = tid + 256 * j;
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4 *>(&Bs[p][(u / 4) * SK + 8 * (u % 4)]) = st.b[p][j];
        }
    };

    f32x4 acc[RT][CT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto mma_chunk = [&]() {
        bf16x8 a[RT][3];
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int p = 0; p < 3; ++p) a[r][p] = *reinterpret_cast<const bf16x8 *>(&As[p][(16 * (wm * RT + r) + li) * SK + 8 * lq]);
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            bf16x8 b[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const bf16x8 *>(&Bs[p][(16 * (wn * CT + c) + li) * SK + 8 * lq]);
#pragma unroll
            for (int r = 0; r < RT; ++r) acc[r][c] = mfma_bf16x6(a[r], b, acc[r][c]);
        }
    };

    // Two chunks of global loads are in flight at any time (two explicit register sets, loop unrolled by two): with two
    // blocks per CU one MFMA phase (600-1200 cycles) does not cover an L2 miss, two nearly do.
    Staged s0, s1;
    load_chunk(0, s0);
    if (nchunks > 1) load_chunk(1, s1);
    for (int chunk = 0; chunk < nchunks; chunk += 2) {
        __syncthreads();
        store_chunk(s0, chunk);
        __syncthreads();
        if (chunk + 2 < nchunks) load_chunk(chunk + 2, s0);
        mma_chunk();
        if (chunk + 1 < nchunks) {
            __syncthreads();
            store_chunk(s1, chunk + 1);
            __syncthreads();
            if (chunk + 3 < nchunks) load_chunk(chunk + 3, s1);
            mma_chunk();
        }
    }

    long mrow[RT][4];                                  // the NHWC row of each of this lane's accumulator rows, -1 past the end
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long q = m0 + 16 * (wm * RT + rt) + 4 * lq + r;
            long m = q < M ? q : -1;
            if (pm && q < M) {
                int b, pix;
                row_of(q, b, pix);
                m = (long)b * P + pix;
            }
            mrow[rt][r] = m;
        }
    float ssum[CT], ssq[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int n = 16 * (wn * CT + c) + li;
        float bv = 0.f, sv = 0.f;
        ssum[c] = 0.f; ssq[c] = 0.f;
        if (EPI == EPI_BIAS_RELU6 || EPI == EPI_BIAS || EPI == EPI_BIAS_RELU || EPI == EPI_BN_RELU6) bv = bias[n];
        if (EPI == EPI_BN_RELU6) sv = shift[n];
        float gsc = 0.f, gsh = 0.f, gmean = 0.f, ginv = 0.f;
        if (EPI == EPI_BNBWD_GATE6) { gsc = bias[n]; gsh = bias[CO + n]; gmean = bias[2 * CO + n]; ginv = bias[3 * CO + n]; }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long m = mrow[rt][r];
                if (m >= 0) {
                    float v = acc[rt][c][r];
                    if (EPI == EPI_RELU) v = fmaxf(v, 0.f);
                    if (EPI == EPI_BIAS_RELU6) v = relu6f(v + bv);
                    if (EPI == EPI_BIAS) v = v + bv;
                    if (EPI == EPI_BIAS_RELU) v = fmaxf(v + bv, 0.f);
                    if (EPI == EPI_BN_RELU6) v = relu6f(fmaf(v, bv, sv));
                    if (EPI == EPI_BNBWD_GATE6) {
                        const float zv = shift[m * CO + n], y = fmaf(zv, gsc, gsh);
                        v = (y > 0.f && y < 6.f) ? v : 0.f;
                        dst[m * CO + n] = v;
                        if (STATS) { ssum[c] += v; ssq[c] = fmaf(v, (zv - gmean) * ginv, ssq[c]); }
                        continue;
                    }
                    dst[m * CO + n] = v;
                    if (STATS) { ssum[c] += v; ssq[c] = fmaf(v, v, ssq[c]); }
                }
            }
    }
    if (STATS) {
        // lanes with equal li hold the same column: reduce over lq (xor 16, 32), then over the two waves along M through LDS
        __syncthreads();                                            // the K loop's reads of As are done: reuse it
        double *red = reinterpret_cast<double *>(&As[0][0]);        // [2 (wm)][2 (which)][CO]
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            double a = (double)ssum[c], q = (double)ssq[c];
            a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
            q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
            const int n = 16 * (wn * CT + c) + li;
            if (lq == 0) { red[(wm * 2 + 0) * CO + n] = a; red[(wm * 2 + 1) * CO + n] = q; }
        }
        __syncthreads();
        for (int i = tid; i < 2 * CO; i += 256) {
            const int which = i / CO, n = i % CO;
            partial[((long)which * CO + n) * partial_stride + blockIdx.x] = red[(0 * 2 + which) * CO + n] + red[(1 * 2 + which) * CO + n];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient in the split-precision form: dW[tap][ci][co] += sum_m x[pix(m) + tap][ci] * dz[m][co].
// The reduction index is the output pixel m, the STRIDED dimension of both NHWC operands.  A chunk of 32 pixels is staged
// in its natural [pixel][channel] order as three bf16 planes (one ds_write_b64 per plane and float4), and the MFMA operands
// are fetched with gfx950's transposing LDS read (ds_read_b64_tr_b16: a 16-lane group reads 4 rows x 16 channels and lane j
// receives channel j of the 4 rows).  Lane group kg takes pixels 4kg..4kg+3 and 16+4kg..16+4kg+3 of the chunk as its 8
// k-values (the same order for both operands), so a 32-lane half reads 8 consecutive rows; with a row stride of 8*odd words
// these fall into 8 different bank octets.  Block = (tap, pixel range); the 9 taps of a range are given to the same XCD so
// dz and the overlapping x rows are shared in its L2.  The block owns the whole (CIN x COUT) tile of its tap: waves split
// COUT, every wave keeps all CIN/16 row tiles; the tile is gathered in LDS at the end and added with contiguous atomics.
// ---------------------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int tr_row_words(int C) { return ((C / 2 / 8) & 1) ? C / 2 : C / 2 + 8; }   // words per [pixel] row: 8 * odd

typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s16x4 lds_read_tr16(const unsigned char *p)
{
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p));
}

// TPB = taps per block: 1, or KW (= 3: one kernel row).  With three taps per block the dz chunk is staged and split once
// for three products and the wave does 3 x MT x NW MFMA groups per chunk, which moves the kernel from issue-bound on the
// split arithmetic (one tap: ~250 vector instructions beside 48 MFMAs per wave and chunk) to matrix-bound.
// DPRE: dz arrives already split (planes `dpl`, NHWC like dzp, which is then unused): its staging is a 16-byte copy per plane.
// XBN: `x` is the pre-activation tensor z of the BatchNormalization -> ReLU6 in front of the convolution, xbn = its scale[CIN] | shift[CIN];
// the activation x = relu6(z * scale + shift) is formed while it is staged (see conv_bf16_kernel<..., ABN>).
template <int CIN, int COUT, int TPB, bool DPRE = false, bool XBN = false>
__global__ __launch_bounds__(256, 2) void conv_wgrad_bf16_kernel(const float *__restrict__ x, const float *__restrict__ dzp,
                                                                  float *__restrict__ dw, const float *__restrict__ zero_page,
                                                                  ConvGeom g, int chunks_per_block, int nranges,
                                                                  Bf16Planes dpl = Bf16Planes{{nullptr, nullptr, nullptr}},
                                                                  const float *__restrict__ xbn = nullptr)
{
    constexpr int KC = 32;                                        // pixels per chunk
    constexpr int XS = 4 * tr_row_words(CIN), DS = 4 * tr_row_words(COUT);   // row strides in bytes
    constexpr int MT = CIN / 16, NT = COUT / 16, NW = NT / 4;     // row tiles, column tiles, column tiles per wave
    constexpr int XU = KC * CIN / 4, DU = KC * COUT / 4;          // float4 units of a chunk
    constexpr int NXU = XU / 256, NDU = DPRE ? DU / 2 / 256 : DU / 256;   // dz units: 8 channels of a plane or one float4
    constexpr int XT = 3 * KC * XS;                               // bytes of one tap's x planes
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    static_assert(COUT % 64 == 0 && CIN % 32 == 0, "four waves split COUT in 16-column tiles; every thread stages whole units");
    extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
    unsigned char *Xs = wsm;                                      // [TPB][3][KC][XS]
    unsigned char *Ds = wsm + TPB * XT;                           // [3][KC][DS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lq = lane >> 4;
    const int HW = g.Ho * g.Wo;
    const long M = (long)g.B * HW;
    const int ngroups = g.KH * g.KW / TPB;                        // tap groups (TPB == KW: kernel rows)
    // block id -> (tap group, range): XCD = id % 8 holds every tap of its ranges
    const int bid = blockIdx.x, tap0 = ((bid >> 3) % ngroups) * TPB, range = (bid / (8 * ngroups)) * 8 + (bid & 7);
    const long nchunk_all = (M + KC - 1) / KC, chunk0 = (long)range * chunks_per_block;
    const int nchunks = (range >= nranges || chunk0 >= nchunk_all) ? 0
                        : (int)(chunk0 + chunks_per_block <= nchunk_all ? chunks_per_block : nchunk_all - chunk0);

    f32x4 acc[TPB][MT][NW];
#pragma unroll
    for (int t = 0; t < TPB; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nw = 0; nw < NW; ++nw) acc[t][mt][nw] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // staging units: unit u -> float4 e = u % (C/4) of pixel p = u / (C/4): coalesced global reads, contiguous LDS rows.
    // The unit's clip and pixel-in-clip advance by 32 pixels per chunk without divisions; rows past M and taps that fall
    // into the padding read the zero page.
    const float rWo = 1.0f / (float)g.Wo;
    int xb[NXU], xpix[NXU];
#pragma unroll
    for (int j = 0; j < NXU; ++j) {
        const long m = chunk0 * KC + (tid + 256 * j) / (CIN / 4);
        xb[j] = (int)(m / HW); xpix[j] = (int)(m - (long)xb[j] * HW);
    }
    f32x4 sx[TPB][NXU], sd[DPRE ? 1 : NDU];
    u32x4 sdp[DPRE ? 3 : 1][NDU];
    unsigned xok = 0u;                                            // XBN: which staged x units are real rows (bit j * TPB + t)
    f32x4 xsc = {0.f, 0.f, 0.f, 0.f}, xsh = {0.f, 0.f, 0.f, 0.f};   // XBN: this thread's four channels (the same for all of its units)
    if constexpr (XBN) {
        static_assert(256 % (CIN / 4) == 0 && NXU * TPB <= 32, "one channel group per thread");
        xsc = *reinterpret_cast<const f32x4 *>(xbn + 4 * (tid % (CIN / 4)));
        xsh = *reinterpret_cast<const f32x4 *>(xbn + CIN + 4 * (tid % (CIN / 4)));
    }
    auto load_chunk = [&](int ch) {
        const long m0 = (chunk0 + ch) * KC;
        xok = 0u;
#pragma unroll
        for (int j = 0; j < NXU; ++j) {
            const int e = (tid + 256 * j) % (CIN / 4);
            const int oy = (int)(((float)xpix[j] + 0.5f) * rWo), ox = xpix[j] - oy * g.Wo;
#pragma unroll
            for (int t = 0; t < TPB; ++t) {
                const int tap = tap0 + t, kh = tap / g.KW, kw = tap - kh * g.KW;
                const int sy = oy * g.stride + kh - g.pt, sxx = ox * g.stride + kw - g.pl;
                const bool ok = xb[j] < g.B && sy >= 0 && sy < g.H && sxx >= 0 && sxx < g.W;
                const float *p = ok ? x + (((long)xb[j] * g.H + sy) * g.W + sxx) * CIN + 4 * e : zero_page;
                sx[t][j] = *reinterpret_cast<const f32x4 *>(p);
                if (XBN && ok) xok |= 1u << (j * TPB + t);
            }
            xpix[j] += KC;
            while (xpix[j] >= HW) { xpix[j] -= HW; ++xb[j]; }
        }
#pragma unroll
        for (int j = 0; j < NDU; ++j) {
            const int u = tid + 256 * j;
            if (DPRE) {
                const long m = m0 + u / (COUT / 8);
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    sdp[p][j] = m < M ? *reinterpret_cast<const u32x4 *>(dpl.p[p] + m * COUT + 8 * (u % (COUT / 8))) : (u32x4){0u, 0u, 0u, 0u};
            } else {
                const long m = m0 + u / (COUT / 4);
                const float *p = m < M ? dzp + m * COUT + 4 * (u % (COUT / 4)) : zero_page;
                sd[j] = *reinterpret_cast<const f32x4 *>(p);
            }
        }
    };
    auto store_planes = [&](unsigned char *base, int plane_bytes, int o, f32x4 v) {
        bf16x4 h, m, l;
        split_bf16(v, h, m, l);
        *reinterpret_cast<bf16x4 *>(base + o) = h;
        *reinterpret_cast<bf16x4 *>(base + plane_bytes + o) = m;
        *reinterpret_cast<bf16x4 *>(base + 2 * plane_bytes + o) = l;
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int j = 0; j < NXU; ++j) {
            const int u = tid + 256 * j, o = (u / (CIN / 4)) * XS + 8 * (u % (CIN / 4));
#pragma unroll
            for (int t = 0; t < TPB; ++t) {
                f32x4 xv = sx[t][j];
                if constexpr (XBN) {
                    if (xok & (1u << (j * TPB + t))) {
#pragma unroll
                        for (int e2 = 0; e2 < 4; ++e2) xv[e2] = relu6f(fmaf(xv[e2], xsc[e2], xsh[e2]));
                    }
                }
                store_planes(Xs + t * XT, KC * XS, o, xv);
            }
        }
#pragma unroll
        for (int j = 0; j < NDU; ++j) {
            const int u = tid + 256 * j;
            if (DPRE) {
                const int o = (u / (COUT / 8)) * DS + 16 * (u % (COUT / 8));
#pragma unroll
                for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4 *>(Ds + p * KC * DS + o) = sdp[p][j];
            } else {
                store_planes(Ds, KC * DS, (u / (COUT / 4)) * DS + 8 * (u % (COUT / 4)), sd[j]);
            }
        }
    };
    // transposed fragment: lane 4q+pp of a group supplies row q, channels 4pp..4pp+3 of the 16-channel tile
    const int trq = li >> 2, trp = li & 3;
    const unsigned char *xfrag = Xs + (4 * lq + trq) * XS + 8 * trp;
    const unsigned char *dfrag = Ds + (4 * lq + trq) * DS + 8 * trp + 32 * (wave * NW);
    auto frag = [&](const unsigned char *base, int stride, int plane, int tile) -> bf16x8 {
        union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
        u.s.lo = lds_read_tr16(base + plane * KC * stride + 32 * tile);
        u.s.hi = lds_read_tr16(base + plane * KC * stride + 32 * tile + 16 * stride);
        return u.v;
    };

    // one chunk of global loads in flight (a second set changed nothing when it was tried with one tap per block)
    if (nchunks > 0) load_chunk(0);
    for (int ch = 0; ch < nchunks; ++ch) {
        __syncthreads();
        store_chunk();
        __syncthreads();
        if (ch + 1 < nchunks) load_chunk(ch + 1);
        bf16x8 b[NW][3];
#pragma unroll
        for (int nw = 0; nw < NW; ++nw)
#pragma unroll
            for (int p = 0; p < 3; ++p) b[nw][p] = frag(dfrag, DS, p, nw);
#pragma unroll
        for (int t = 0; t < TPB; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                bf16x8 a[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) a[p] = frag(xfrag + t * XT, XS, p, mt);
#pragma unroll
                for (int nw = 0; nw < NW; ++nw) acc[t][mt][nw] = mfma_bf16x6(a, b[nw], acc[t][mt][nw]);
            }
    }
    // gather each tap's (CIN x COUT) tile in LDS (reusing the staging space) and add it with contiguous atomics
    float *red = reinterpret_cast<float *>(wsm);                   // [CIN][COUT]
#pragma unroll
    for (int t = 0; t < TPB; ++t) {
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nw = 0; nw < NW; ++nw)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[(16 * mt + 4 * lq + r) * COUT + 16 * (wave * NW + nw) + li] = acc[t][mt][nw][r];
        __syncthreads();
        if (nchunks > 0)
            for (int i = tid; i < CIN * COUT; i += 256) atomicAdd(dw + (long)(tap0 + t) * CIN * COUT + i, red[i]);
    }
}

// N consecutive floats with the widest aligned load (N = 1, 2, 4, 8); p must be N*4-byte aligned (16 for N = 8)
template <int N>
__device__ __forceinline__ void load_vec(const float *__restrict__ p, float (&v)[N])
{
    if constexpr (N == 1) {
        v[0] = p[0];
    } else if constexpr (N == 2) {
        const float2 t = *reinterpret_cast<const float2 *>(p);
        v[0] = t.x; v[1] = t.y;
    } else {
#pragma unroll
        for (int i = 0; i < N / 4; ++i) {
            const float4 t = *reinterpret_cast<const float4 *>(p + 4 * i);
            v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// WGRAD, LDS-free: every MFMA fragment comes straight from global memory.
//   A^T fragment (row = input channel, k = pixel): lane (li, lq) reads x[pix(p0+lq) + tap][cb + 16 mt + li]
//   B   fragment (k = pixel, col = out channel):   lane (li, lq) reads dz[p0+lq][16 nt + li]
// The texture addresser spends ~16 cycles on a wave load whatever its width, so the 16x16 tiles use a PERMUTED channel
// map: M-tile e holds input channels cb + MT*row + e and N-tile e holds output channels NT*col + e; a lane then fetches
// its MT (resp. NT) fragments with ONE vector load of consecutive channels.  The GPB taps of a block re-touch the same
// lines (L1/L2 hits).  Each wave owns a
// contiguous range of 4-pixel steps and ALL GPB*MT*NT output tiles of the block (A fragments are reused across NT,
// B fragments across GPB*MT), so there is no staging, no barrier and one LDS reduction + one atomic set per block.
// ---------------------------------------------------------------------------------------------------------------
template <int CIN, int COUT, int GPB>
__global__ __launch_bounds__(256) void conv_wgrad_direct_kernel(const float *__restrict__ x, const float *__restrict__ dz,
                                                                 float *__restrict__ dw, const float *__restrict__ zeros,
                                                                 ConvGeom g, int steps_per_wave)
{
    constexpr int CB = CIN >= 64 ? 64 : CIN;
    constexpr int CBLK = CIN / CB;
    constexpr int MT = CB / 16, NT = COUT / 16;
    static_assert(CIN % CB == 0 && CB % 16 == 0 && COUT % 16 == 0, "channel counts must be multiples of 16");
    extern __shared__ __attribute__((aligned(16))) float red[];     // [4 waves][64 lanes][4] per tile round

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
    const long M = (long)g.B * g.Ho * g.Wo;
    const int HoWo = g.Ho * g.Wo;
    const int ngroups = g.KH * g.KW * CBLK, grp0 = blockIdx.y * GPB;
    const long step0 = ((long)blockIdx.x * 4 + wave) * steps_per_wave;

    // per-group constants (wave-uniform): tap coordinates and the tap's constant element offset from the pixel's base
    int kh[GPB], kw[GPB], cb[GPB], toff[GPB];
    bool gok[GPB];
#pragma unroll
    for (int gi = 0; gi < GPB; ++gi) {
        const int grp = grp0 + gi, tap = grp / CBLK;
        gok[gi] = grp < ngroups;
        kh[gi] = tap / g.KW;
        kw[gi] = tap % g.KW;
        cb[gi] = (grp % CBLK) * CB;
        toff[gi] = (kh[gi] * g.W + kw[gi]) * CIN + cb[gi];
    }

    f32x4 acc[GPB][MT][NT];
#pragma unroll
    for (int gi = 0; gi < GPB; ++gi)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[gi][mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // this lane's pixel p = 4*step + lq, decoded once and advanced by 4 per step
    long p = step0 * 4 + lq;
    int b = (int)(p / HoWo), rem = (int)(p % HoWo), oh = rem / g.Wo, ow = rem % g.Wo;
    // 32-bit element offsets (the host checks that both tensors have < 2^31 elements)
    auto load_frags = [&](float (&af)[GPB][MT], float (&bf)[NT]) {
        // out-of-range pixels / padding taps read the zero page: no op touches a loaded value before its MFMA
        const bool pok = p < M;
        load_vec<NT>(pok ? dz + ((int)p * COUT + NT * li) : zeros + NT * li, bf);
        const int y0 = oh * g.stride - g.pt, x0 = ow * g.stride - g.pl;
        const int base = ((b * g.H + y0) * g.W + x0) * CIN + MT * li;  // may point into the halo; used only when in range
#pragma unroll
        for (int gi = 0; gi < GPB; ++gi) {
            const bool ok = pok && gok[gi] && (unsigned)(y0 + kh[gi]) < (unsigned)g.H && (unsigned)(x0 + kw[gi]) < (unsigned)g.W;
            load_vec<MT>(ok ? x + (base + toff[gi]) : zeros + MT * li, af[gi]);
        }
        p += 4;
        ow += 4;
        while (ow >= g.Wo) { ow -= g.Wo; ++oh; }
        while (oh >= g.Ho) { oh -= g.Ho; ++b; }
    };
    const long left = (M + 3) / 4 - step0;
    const int nst = left <= 0 ? 0 : (left < steps_per_wave ? (int)left : steps_per_wave);
    float afn[GPB][MT], bfn[NT];
    if (nst > 0) load_frags(afn, bfn);
    for (int st = 0; st < nst; ++st) {
        float af[GPB][MT], bf[NT];
#pragma unroll
        for (int gi = 0; gi < GPB; ++gi)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[gi][mt] = afn[gi][mt];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[nt] = bfn[nt];
        if (st + 1 < nst) load_frags(afn, bfn);          // next step's fragments fly while this step's MFMAs issue
#pragma unroll
        for (int gi = 0; gi < GPB; ++gi)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[gi][mt][nt] = mfma16(af[gi][mt], bf[nt], acc[gi][mt][nt]);
    }

    // Reduce the 4 waves' partial tiles through LDS and gather each (group, M-tile) block as 16 full rows of COUT
    // contiguous output channels (the permuted N-tiles interleave), so every atomic wave-instruction adds 64 contiguous
    // floats -- strided float atomics run ~10x slower (MI355X_MICROARCH.md, Global float atomics).
    float *rowblk = red + 4 * 64 * 4;                                 // [16][COUT]
#pragma unroll
    for (int gi = 0; gi < GPB; ++gi)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                __syncthreads();
                *reinterpret_cast<f32x4 *>(&red[(wave * 64 + lane) * 4]) = acc[gi][mt][nt];
                __syncthreads();
                if (wave == 0) {
                    const f32x4 a0 = *reinterpret_cast<const f32x4 *>(&red[(0 * 64 + lane) * 4]);
                    const f32x4 a1 = *reinterpret_cast<const f32x4 *>(&red[(1 * 64 + lane) * 4]);
                    const f32x4 a2 = *reinterpret_cast<const f32x4 *>(&red[(2 * 64 + lane) * 4]);
                    const f32x4 a3 = *reinterpret_cast<const f32x4 *>(&red[(3 * 64 + lane) * 4]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) rowblk[(4 * lq + r) * COUT + NT * li + nt] = (a0[r] + a1[r]) + (a2[r] + a3[r]);
                }
            }
            __syncthreads();
            if (gok[gi]) {
                const int tap = (grp0 + gi) / CBLK;
                for (int idx = threadIdx.x; idx < 16 * COUT; idx += 256) {
                    const int row = idx / COUT, co = idx % COUT;
                    atomicAdd(dw + ((long)(tap * CIN + cb[gi] + MT * row + mt)) * COUT + co, rowblk[idx]);
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------
// DGRAD, LDS-free.  dx[m][n] = sum_{tap, c} dz[pixT(m, tap)][c] * W[tap][n][c]   (HWIO: I = n = CO, O = c = CR)
// Both operands are contiguous along c, so with the K order permuted to k = 16 jj + 4 lq + j every lane fetches ONE
// float4 per fragment and feeds four MFMA k-steps from it (element j of the A and of the B vector share the same k).
// A wave owns MW x 16 rows (input pixels) and all CO/16 column tiles; W fragments come from L1/L2 (<= 288 KB, shared
// by every wave), dz rows are 64-byte segments.  Strided convolutions are launched once per parity class
// (cy, cx): rows of a class share the set of contributing taps, so no zero tap is ever multiplied.
// ---------------------------------------------------------------------------------------------------------------
struct DgradClass { int cy, cx, ny, nx; };   // rows of the class: ih = stride*a + cy (a < ny), iw = stride*b + cx (b < nx)
struct DgradClasses { DgradClass c[4]; };     // one launch covers every class: blockIdx.y selects it (blocks past a class's rows exit)

template <int CR, int CO, int MW, int STRIDE>
__global__ __launch_bounds__(256) void conv_dgrad_direct_kernel(const float *__restrict__ dz, const float *__restrict__ wgt,
                                                                 float *__restrict__ dx, const float *__restrict__ zeros,
                                                                 ConvGeom g, DgradClasses classes)
{
    constexpr int NT = CO / 16, JJ = CR / 16;
    const DgradClass cls = classes.c[blockIdx.y];
    static_assert(CR % 16 == 0 && CO % 16 == 0, "channel counts must be multiples of 16");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
    const long Mc = (long)g.B * cls.ny * cls.nx;                      // rows of this class
    const long m0 = ((long)blockIdx.x * 4 + wave) * (16 * MW);
    if (m0 >= Mc) return;

    // taps that reach this class: kh = kh0 + th*STRIDE with kh0 == (cy + pt) (mod STRIDE); likewise kw
    const int kh0 = (cls.cy + g.pt) % STRIDE, kw0 = (cls.cx + g.pl) % STRIDE;
    const int nkh = (g.KH - kh0 + STRIDE - 1) / STRIDE, nkw = (g.KW - kw0 + STRIDE - 1) / STRIDE;
    const int nit = nkh * nkw * JJ;

    // A-fragment rows of this lane (row li of each of the MW tiles): decoded ONCE into
    //   rbase = element offset of dz[b][oh0][ow0][4*lq] with (oh0, ow0) the source pixel of tap (kh0, kw0); the source of
    //           tap (th, tw) is (oh0 - th, ow0 - tw), i.e. rbase - (th*Wo + tw)*CR  (one scalar offset per tap)
    //   rmask = bit th     : 0 <= oh0 - th < Ho      bit 8 + tw : 0 <= ow0 - tw < Wo      (0 when the row is out of range)
    // so the K loop spends one add and one mask test per row instead of re-deriving the address with multiplies.
    int rbase[MW], rmask[MW];
    {
        const long m = m0 + li;
        const long mm = m < Mc ? m : 0;
        const int per = cls.ny * cls.nx, rem = (int)(mm % per);
        int b = (int)(mm / per), a = rem / cls.nx, c = rem % cls.nx;
#pragma unroll
        for (int mt = 0; mt < MW; ++mt) {
            const int oh0 = (a * STRIDE + cls.cy + g.pt - kh0) / STRIDE, ow0 = (c * STRIDE + cls.cx + g.pl - kw0) / STRIDE;
            rbase[mt] = ((b * g.Ho + oh0) * g.Wo + ow0) * CR + 4 * lq;
            int msk = 0;
            if (m + 16 * mt < Mc) {
#pragma unroll
                for (int t = 0; t < 8; ++t) {                                 // up to 8 taps per axis (host-checked)
                    if (t < nkh && oh0 - t >= 0 && oh0 - t < g.Ho) msk |= 1 << t;
                    if (t < nkw && ow0 - t >= 0 && ow0 - t < g.Wo) msk |= 256 << t;
                }
            }
            rmask[mt] = msk;
            c += 16;
            while (c >= cls.nx) { c -= cls.nx; ++a; }
            while (a >= cls.ny) { a -= cls.ny; ++b; }
        }
    }

    f32x4 acc[MW][NT];
#pragma unroll
    for (int mt = 0; mt < MW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // iteration -> (tap row, tap col, channel block) kept incrementally in scalars: no division in the loop
    int n_th = 0, n_tw = 0, n_jj = 0;
    const int wrow = (16 * 0 + li) * CR + 4 * lq;                            // this lane's offset inside a W[tap] slab
    auto load_frags = [&](float4 (&af)[MW], float4 (&bf)[NT]) {
        const int tapoff = (n_th * g.Wo + n_tw) * CR - 16 * n_jj;            // wave-uniform
        const int tapbit = (1 << n_th) | (256 << n_tw);
        const bool live = n_th < nkh;                                         // false for the (harmless) loads past the end
        const int tap = live ? (kh0 + n_th * STRIDE) * g.KW + kw0 + n_tw * STRIDE : 0;
#pragma unroll
        for (int mt = 0; mt < MW; ++mt) {
            // padding rows read the zero page: no op touches the loaded value before its MFMA, so the wait sits there
            const bool ok = live && (rmask[mt] & tapbit) == tapbit;
#if defined(KWS_DGRAD_MODE) && KWS_DGRAD_MODE == 2      // diagnostic build: no global loads, MFMAs on constants
            af[mt] = make_float4(ok ? 1.f : 0.f, 0.5f, 0.25f, 2.f);
            continue;
#endif
            af[mt] = *reinterpret_cast<const float4 *>(ok ? dz + (rbase[mt] - tapoff) : zeros + 4 * lq);
        }
        const float *wp = wgt + (tap * CO) * CR + 16 * n_jj + wrow;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#if defined(KWS_DGRAD_MODE) && KWS_DGRAD_MODE == 2
            bf[nt] = make_float4(1.f, 2.f, 3.f, (float)tap);
            continue;
#endif
            bf[nt] = *reinterpret_cast<const float4 *>(wp + 16 * nt * CR);
        }
        if (++n_jj == JJ) { n_jj = 0; if (++n_tw == nkw) { n_tw = 0; ++n_th; } }
    };

    auto mma = [&](const float4 (&af)[MW], const float4 (&bf)[NT]) {
#if defined(KWS_DGRAD_MODE) && KWS_DGRAD_MODE == 1      // diagnostic build (tools/dgrad_modes.hip): keep the loads, drop the MFMAs
#pragma unroll
        for (int mt = 0; mt < MW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt][0] += af[mt].x + bf[nt].y;
        return;
#endif
        // k-step outermost: consecutive MFMAs hit different accumulators
#pragma unroll
        for (int mt = 0; mt < MW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma16(af[mt].x, bf[nt].x, acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < MW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma16(af[mt].y, bf[nt].y, acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < MW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma16(af[mt].z, bf[nt].z, acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < MW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma16(af[mt].w, bf[nt].w, acc[mt][nt]);
    };
    // Explicit two-set ping-pong (no loop-carried register copies): set 1 loads fly while set 0 feeds the MFMAs.
    // The waves of a SIMD run this loop in lockstep, so a body of [address math + loads][MFMA batch] leaves the matrix
    // pipe idle while everyone computes addresses and the vector ALU idle while everyone multiplies (measured: 2.5k +
    // 2.2k + wait cycles per iteration).  sched_group_barrier interleaves the two streams instruction by instruction.
    constexpr int NM = 4 * MW * NT;                        // MFMAs per fragment set
    constexpr int VPM = (24 * MW + 6 * NT + NM - 1) / NM + 1;   // VALU ops to slot behind each MFMA
    float4 a0[MW], b0[NT], a1[MW], b1[NT];
    load_frags(a0, b0);
    for (int it = 0; it < nit; it += 2) {
        load_frags(a1, b1);                                // (past the end: masked rows, clamped tap -> harmless)
        mma(a0, b0);
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);    // VALU
            if (i < MW + NT) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read
        }
        load_frags(a0, b0);
        if (it + 1 < nit) mma(a1, b1);
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
            __builtin_amdgcn_sched_group_barrier(0x002, VPM, 1);
            if (i < MW + NT) __builtin_amdgcn_sched_group_barrier(0x020, 1, 1);
        }
    }

    // D layout: row = 4*lq + r, col = li.  With stride 1 the class order is the pixel order: offset = m * CO.
#pragma unroll
    for (int mt = 0; mt < MW; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long m = m0 + 16 * mt + 4 * lq + r;
            if (m < Mc) {
                long pix = m;
                if (STRIDE != 1) {
                    const int per = cls.ny * cls.nx, rem = (int)(m % per), b = (int)(m / per);
                    pix = ((long)b * g.H + (rem / cls.nx) * STRIDE + cls.cy) * g.W + (rem % cls.nx) * STRIDE + cls.cx;
                }
                float *o = dx + pix * CO + li;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) o[16 * nt] = acc[mt][nt][r];
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------
// Clip-resident kernels for a 3x3 / stride-1 / 'same' layer with FEW channels (conv2: 16 -> 32).
// Measured on the LDS-free kernels above: loads alone cost as much as loads + MFMAs (every dz row is fetched once per
// tap through the texture path, every wave re-fetches the weights).  Here a block walks whole clips: the clip's tile is
// staged in LDS ONCE (zero halo = padding), fragments are LDS reads at tap offsets, and the wave's weight fragments
// (dgrad) or accumulators (wgrad) live in registers for the whole kernel.
// ---------------------------------------------------------------------------------------------------------------
// z[b][oh][ow][n] = sum_{tap,c} x[b][oh+kh-1][ow+kw-1][c] * W[tap][c][n],  CIN = 16, COUT = 16*NT (conv2 forward).
// K order permuted (k = 4 lq + j inside a tap) so the A fragment is ONE ds_read_b128 per tap; the wave's 9 x 4 x NT weight
// fragments stay in registers.  The BatchNormalization batch statistics are fused: every lane accumulates sum / sum of
// squares of its output column and the block writes ONE double partial per channel (the layout bn_finalize_train_kernel
// reads), so the pre-BN tensor is not re-read by a statistics pass.
// POOLED (inference): the tiles enumerate 2x2 pool windows (pixel p = 4*window + element, as in kws_layer1.h), so a lane's
// four accumulator rows are one window; the epilogue applies the BatchNorm affine, ReLU6 and the max and writes the pooled
// activation (H/2 x W/2 x COUT) instead of the conv output: no z-sized tensor, no separate activation pass.
template <int COUT, bool STATS, bool POOLED = false>
__global__ __launch_bounds__(256) void conv_fwd_clip_kernel(const float *__restrict__ x, const float *__restrict__ wgt,
                                                             float *__restrict__ z, int B, int H, int W,
                                                             double *__restrict__ partial, int partial_stride,
                                                             const float *__restrict__ scale = nullptr,
                                                             const float *__restrict__ shift = nullptr)
{
    constexpr int CIN = 16, NT = COUT / 16, XP = CIN + 4;          // padded pixel stride (words)
    extern __shared__ __attribute__((aligned(16))) float tile[];  // [(H+2)][(W+2)][XP], zero halo
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), li = lane & 15, lq = lane >> 4;
    const int HP = H + 2, WP = W + 2, HW = H * W;
    const int W2 = W / 2, n2 = (H / 2) * W2;                       // pool windows (POOLED)
    const int ntile = POOLED ? (n2 + 3) / 4 : (HW + 15) / 16;
    for (int i = threadIdx.x; i < HP * WP * XP; i += 256) tile[i] = 0.f;
    float sc[NT], sh[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { sc[nt] = POOLED ? scale[16 * nt + li] : 1.f; sh[nt] = POOLED ? shift[16 * nt + li] : 0.f; }

    // B fragments: k-step j of tap t multiplies input channel 4 lq + j; lane holds W[t][4 lq + j][16 nt + li]
    float wf[9][4][NT];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wf[t][j][nt] = wgt[(t * CIN + 4 * lq + j) * COUT + 16 * nt + li];

    constexpr int PF = 3;                                         // float4 per thread for the next clip (<= 768 per clip)
    const int nf4 = HW * (CIN / 4);
    float4 pf[PF];
    auto prefetch = [&](int b) {
        const float4 *src = reinterpret_cast<const float4 *>(x + (long)b * HW * CIN);
#pragma unroll
        for (int j = 0; j < PF; ++j) { const int i = threadIdx.x + 256 * j; pf[j] = i < nf4 ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f); }
    };
    float ssum[NT], ssq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { ssum[nt] = 0.f; ssq[nt] = 0.f; }

    if ((int)blockIdx.x < B) prefetch(blockIdx.x);
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int i = threadIdx.x + 256 * j;
            if (i < nf4) {
                const int pix = i / (CIN / 4), c4 = i % (CIN / 4), y = pix / W, xx = pix % W;
                *reinterpret_cast<float4 *>(&tile[((y + 1) * WP + xx + 1) * XP + 4 * c4]) = pf[j];
            }
        }
        if (nf4 > 256 * PF) {
            const float4 *src = reinterpret_cast<const float4 *>(x + (long)b * HW * CIN);
            for (int i = threadIdx.x + 256 * PF; i < nf4; i += 256) {
                const int pix = i / (CIN / 4), c4 = i % (CIN / 4), y = pix / W, xx = pix % W;
                *reinterpret_cast<float4 *>(&tile[((y + 1) * WP + xx + 1) * XP + 4 * c4]) = src[i];
            }
        }
        __syncthreads();
        if (b + (int)gridDim.x < B) prefetch(b + gridDim.x);
        for (int t = wave; t < ntile; t += 4) {
            int oh, ow;
            if (POOLED) {                                            // A row li = element li & 3 of window 4 t + (li >> 2)
                int wq = 4 * t + (li >> 2);
                wq = wq < n2 ? wq : 0;
                oh = 2 * (wq / W2) + ((li & 3) >> 1);
                ow = 2 * (wq % W2) + (li & 1);
            } else {
                const int p = 16 * t + li, pc = p < HW ? p : HW - 1;
                oh = pc / W; ow = pc % W;
            }
            const float *a0 = &tile[(oh * WP + ow) * XP + 4 * lq];  // tap (0,0) of this pixel in halo coordinates
            f32x4 acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float4 a = *reinterpret_cast<const float4 *>(a0 + (kh * WP + kw) * XP);
                    const int tp = kh * 3 + kw;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16(a.x, wf[tp][0][nt], acc[nt]);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16(a.y, wf[tp][1][nt], acc[nt]);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16(a.z, wf[tp][2][nt], acc[nt]);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16(a.w, wf[tp][3][nt], acc[nt]);
                }
            if (POOLED) {
                const int wd = 4 * t + lq;                          // this lane's four rows are the elements of window wd
                if (wd < n2) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const float m = fmaxf(fmaxf(fmaf(acc[nt][0], sc[nt], sh[nt]), fmaf(acc[nt][1], sc[nt], sh[nt])),
                                              fmaxf(fmaf(acc[nt][2], sc[nt], sh[nt]), fmaf(acc[nt][3], sc[nt], sh[nt])));
                        z[((long)b * n2 + wd) * COUT + 16 * nt + li] = relu6f(m);
                    }
                }
                continue;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int po = 16 * t + 4 * lq + r;
                if (po < HW) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const float v = acc[nt][r];
                        z[((long)b * HW + po) * COUT + 16 * nt + li] = v;
                        if (STATS) { ssum[nt] += v; ssq[nt] = fmaf(v, v, ssq[nt]); }
                    }
                }
            }
        }
    }
    if (STATS) {
        // lanes with equal li hold the same column: reduce over lq (xor 16, 32), then over the 4 waves through LDS
        __syncthreads();
        double *red = reinterpret_cast<double *>(tile);          // [4 waves][2][COUT]
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            double a = (double)ssum[nt], q = (double)ssq[nt];
            a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
            q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
            if (lq == 0) { red[(wave * 2 + 0) * COUT + 16 * nt + li] = a; red[(wave * 2 + 1) * COUT + 16 * nt + li] = q; }
        }
        __syncthreads();
        if (threadIdx.x < 2 * COUT) {
            const int which = threadIdx.x / COUT, c = threadIdx.x % COUT;
            const double v = (red[(0 * 2 + which) * COUT + c] + red[(1 * 2 + which) * COUT + c]) +
                             (red[(2 * 2 + which) * COUT + c] + red[(3 * 2 + which) * COUT + c]);
            partial[((long)which * COUT + c) * partial_stride + blockIdx.x] = v;
        }
    }
}

// The same forward on the bf16 matrix cores in the three-way split form (default precision, see mfma_bf16x6): K = 32 of one
// MFMA covers TWO taps x 16 input channels (k = 8 lq + j: lane group lq < 2 reads tap 2s, lq >= 2 tap 2s + 1, channels
// 8 (lq & 1) + j), five k-steps for the nine taps (the tenth half is zero weights).  The clip's haloed map is split into
// bf16 planes while it is staged, laid out [plane][channel half][halo pixel][8] so that an A fragment is one ds_read_b128
// per plane and 16 consecutive pixels fill the 64 banks.  Waves split (column tile, tile parity): each keeps the 5 x 3
// weight fragments of ITS 16 output channels in registers, and the 10 pixel tiles of a 15 x 10 map divide evenly (the fp32
// form gave its four waves 3, 3, 2, 2 tiles).  30 MFMAs of 16 cycles per tile and wave against 72 of 32.
template <bool STATS, bool POOLED = false>
__global__ __launch_bounds__(256, 2) void conv_fwd_clip_bf16_kernel(const float *__restrict__ x, const float *__restrict__ wgt,
                                                                     float *__restrict__ z, int B, int H, int W,
                                                                     double *__restrict__ partial, int partial_stride,
                                                                     const float *__restrict__ scale = nullptr,
                                                                     const float *__restrict__ shift = nullptr, double *__restrict__ acc_out = nullptr)
{
    // acc_out != nullptr (STATS): the block's sums are added to that accumulator set (kws_device.h: acc_add), no finalize kernel follows
    constexpr int CIN = 16, COUT = 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char ctile[];   // [3 planes][2 halves][(H+2)(W+2)][8 bf16], zero halo
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), li = lane & 15, lq = lane >> 4;
    const int nt = wave & 1, tpar = wave >> 1;
    // halo pixels per (plane, half), rounded to 16: the 16-byte pieces of the two channel halves then sit a multiple of 256 B
    // apart and the ds_read_b128 lane groups (which mix lq = 0/1 lanes) see 16 different bank quads
    const int HP = H + 2, WP = W + 2, HW = H * W, NPIX = (HP * WP + 15) & ~15;
    const int W2 = W / 2, n2 = (H / 2) * W2;                       // pool windows (POOLED)
    const int ntile = POOLED ? (n2 + 3) / 4 : (HW + 15) / 16;
    for (int i = threadIdx.x; i < 6 * NPIX * 4; i += 256) reinterpret_cast<unsigned *>(ctile)[i] = 0u;
    const float sc = POOLED ? scale[16 * nt + li] : 1.f, sh = POOLED ? shift[16 * nt + li] : 0.f;

    // B fragments of k-step s: lane holds W[tap][8 (lq & 1) + j][16 nt + li], tap = 2 s + (lq >> 1) (zero past tap 8)
    bf16x8 wf[5][3];
#pragma unroll
    for (int st = 0; st < 5; ++st) {
        const int tap = 2 * st + (lq >> 1);
        f32x4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = {0.f, 0.f, 0.f, 0.f};
        if (tap < 9) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                w0[j] = wgt[(tap * CIN + 8 * (lq & 1) + j) * COUT + 16 * nt + li];
                w1[j] = wgt[(tap * CIN + 8 * (lq & 1) + 4 + j) * COUT + 16 * nt + li];
            }
        }
        bf16x4 h0, m0, l0, h1, m1, l1;
        split_bf16(w0, h0, m0, l0);
        split_bf16(w1, h1, m1, l1);
        wf[st][0] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        wf[st][1] = __builtin_shufflevector(m0, m1, 0, 1, 2, 3, 4, 5, 6, 7);
        wf[st][2] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    // byte offset of this lane's A fragment of k-step s from its pixel's tap-(0,0) position: tap shift + channel half
    int aoff[5];
#pragma unroll
    for (int st = 0; st < 5; ++st) {
        const int tap = 2 * st + (lq >> 1), tc = tap < 9 ? tap : 8;
        aoff[st] = (((lq & 1) * NPIX) + (tc / 3) * WP + tc % 3) * 16;
    }

    constexpr int PF = 3;                                         // float4 per thread for the next clip (<= 768 per clip)
    const int nf4 = HW * (CIN / 4);
    f32x4 pf[PF];
    auto prefetch = [&](int b) {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(x + (long)b * HW * CIN);
#pragma unroll
        for (int j = 0; j < PF; ++j) { const int i = threadIdx.x + 256 * j; pf[j] = i < nf4 ? src[i] : (f32x4){0.f, 0.f, 0.f, 0.f}; }
    };
    auto stage = [&](int i, f32x4 v) {                            // float4 i of the clip = channels 4 c4.. of pixel i / 4
        const int pix = i >> 2, c4 = i & 3, y = pix / W, xx = pix - y * W;
        bf16x4 h, m, l;
        split_bf16(v, h, m, l);
        unsigned char *d = ctile + (((c4 >> 1) * NPIX + (y + 1) * WP + xx + 1) * 16 + (c4 & 1) * 8);
        *reinterpret_cast<bf16x4 *>(d) = h;
        *reinterpret_cast<bf16x4 *>(d + 2 * NPIX * 16) = m;
        *reinterpret_cast<bf16x4 *>(d + 4 * NPIX * 16) = l;
    };
    float ssum = 0.f, ssq = 0.f;

    if ((int)blockIdx.x < B) prefetch(blockIdx.x);
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int i = threadIdx.x + 256 * j;
            if (i < nf4) stage(i, pf[j]);
        }
        if (nf4 > 256 * PF) {
            const f32x4 *src = reinterpret_cast<const f32x4 *>(x + (long)b * HW * CIN);
            for (int i = threadIdx.x + 256 * PF; i < nf4; i += 256) stage(i, src[i]);
        }
        __syncthreads();
        if (b + (int)gridDim.x < B) prefetch(b + gridDim.x);
        for (int t = tpar; t < ntile; t += 2) {
            int oh, ow;
            if (POOLED) {                                            // A row li = element li & 3 of window 4 t + (li >> 2)
                int wq = 4 * t + (li >> 2);
                wq = wq < n2 ? wq : 0;
                oh = 2 * (wq / W2) + ((li & 3) >> 1);
                ow = 2 * (wq % W2) + (li & 1);
            } else {
                const int p = 16 * t + li, pc = p < HW ? p : HW - 1;
                oh = pc / W; ow = pc % W;
            }
            const unsigned char *a0 = ctile + (oh * WP + ow) * 16;   // tap (0,0) of this pixel in halo coordinates
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < 5; ++st) {
                bf16x8 a[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const bf16x8 *>(a0 + aoff[st] + p * 2 * NPIX * 16);
                acc = mfma_bf16x6(a, wf[st], acc);
            }
            if (POOLED) {
                const int wd = 4 * t + lq;                          // this lane's four rows are the elements of window wd
                if (wd < n2) {
                    const float m = fmaxf(fmaxf(fmaf(acc[0], sc, sh), fmaf(acc[1], sc, sh)), fmaxf(fmaf(acc[2], sc, sh), fmaf(acc[3], sc, sh)));
                    z[((long)b * n2 + wd) * COUT + 16 * nt + li] = relu6f(m);
                }
                continue;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int po = 16 * t + 4 * lq + r;
                if (po < HW) {
                    const float v = acc[r];
                    z[((long)b * HW + po) * COUT + 16 * nt + li] = v;
                    if (STATS) { ssum += v; ssq = fmaf(v, v, ssq); }
                }
            }
        }
    }
    if (STATS) {
        // lanes with equal li hold the same column: reduce over lq (xor 16, 32), then over the two waves of the column tile
        __syncthreads();
        double *red = reinterpret_cast<double *>(ctile);         // [4 waves][2][16]
        double a = (double)ssum, q = (double)ssq;
        a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
        q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
        if (lq == 0) { red[(wave * 2 + 0) * 16 + li] = a; red[(wave * 2 + 1) * 16 + li] = q; }
        __syncthreads();
        if (threadIdx.x < 2 * COUT) {
            const int which = threadIdx.x / COUT, c = threadIdx.x % COUT, n = c / 16, l = c % 16;
            const double v = red[(n * 2 + which) * 16 + l] + red[((n + 2) * 2 + which) * 16 + l];
            if (acc_out) acc_add(acc_out, 2 * COUT, threadIdx.x, v);
            else partial[((long)which * COUT + c) * partial_stride + blockIdx.x] = v;
        }
    }
}

// dx[b][ih][iw][n] = sum_{tap,c} dz[b][ih+1-kh][iw+1-kw][c] * W[tap][n][c],  CR = c range (conv Cout), CO = 16 (conv Cin)
// BN = true fuses the BatchNorm backward of the layer into the staging: the kernel then reads g (gradient w.r.t. the BN
// output, in dz) and z (the conv output), forms dz = gamma*inv * (g - k2 - xhat*k3) on the way into LDS and writes it back
// over g, where the weight-gradient kernel picks it up.
struct BnBwdArgs {
    const float *z, *gamma, *mean, *inv, *k2, *k3;
    // compact form of g (bn_bwd_reduce_pool_kernel<true>): routed value per (pool window, channel) and the element it goes to;
    // gw == nullptr: g is read full-size from the dz buffer
    const float *gw = nullptr;
    const unsigned char *arg = nullptr;
    // acc != nullptr (split-precision clip kernels): the sums of g and g xhat sit in an accumulator set (kws_device.h: acc_add) and the
    // kernel derives k2 = sum g / M, k3 = sum g xhat / M itself; the kernel that is given dgamma also writes dgamma, dbeta, k2, k3 and clears
    // acc_clear_set (block 0) -- what bn_bwd_finalize_kernel did
    const double *acc = nullptr;
    double *acc_clear_set = nullptr;
    long M = 1;
    float *dgamma = nullptr, *dbeta = nullptr, *k2w = nullptr, *k3w = nullptr;
};
// k2 / k3 of this thread's channels 4 c4 .. 4 c4 + 3 (of C <= 64) from the accumulator set: the first 2 C threads of the block sum one
// value each (eight loads), the block shares them through LDS (contains a barrier: every thread of the block calls it); `writer`: this
// block also leaves the finalize outputs
__device__ __forceinline__ void bn_bwd_k_from_acc(const BnBwdArgs &bn, int C, int c4, bool writer, float *k2, float *k3)
{
    __shared__ float kk[128];
    const int i = threadIdx.x;
    if (i < 2 * C) {
        const double t = acc_sum(bn.acc, 2 * C, i);
        const float kv = (float)(t / (double)bn.M);
        kk[i] = kv;
        if (writer) {
            if (i < C) { bn.dbeta[i] = (float)t; bn.k2w[i] = kv; }
            else { bn.dgamma[i - C] = (float)t; bn.k3w[i - C] = kv; }
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) { k2[e] = kk[4 * c4 + e]; k3[e] = kk[C + 4 * c4 + e]; }
}
template <int CR, bool BN>
__global__ __launch_bounds__(256) void conv_dgrad_clip_kernel(float *__restrict__ dz, const float *__restrict__ wgt,
                                                               float *__restrict__ dx, int B, int H, int W, BnBwdArgs bn)
{
    constexpr int CO = 16, JJ = CR / 16, CRP = CR + 4;           // padded pixel stride: spreads ds_read_b128 over the banks
    extern __shared__ __attribute__((aligned(16))) float tile[];  // [(H+2)][(W+2)][CRP], zero halo
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
    const int HP = H + 2, WP = W + 2, HW = H * W, ntile = (HW + 15) / 16;
    for (int i = threadIdx.x; i < HP * WP * CRP; i += 256) tile[i] = 0.f;

    // this wave's weight fragments for every (tap, jj): lane holds W[tap][n = li][c = 16 jj + 4 lq .. +3]
    float4 wf[9][JJ];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int jj = 0; jj < JJ; ++jj) wf[t][jj] = *reinterpret_cast<const float4 *>(wgt + ((t * CO + li) * CR + 16 * jj + 4 * lq));

    constexpr int F4 = CR / 4;                                    // float4 per pixel
    constexpr int PF = 5;                                         // float4 held per thread for the next clip (<= 1280 per clip)
    static_assert(256 % F4 == 0, "a thread keeps one channel group for all its float4");
    const int nf4 = HW * F4;
    float4 pf[PF], pz[PF];
    // BN coefficients of this thread's 4 channels (c4 = threadIdx.x % F4 for every float4 it stages)
    float gi[4], mean[4], inv[4], k2[4], k3[4];
    if (BN) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * (threadIdx.x % F4) + e;
            inv[e] = bn.inv[c]; gi[e] = bn.gamma[c] * inv[e]; mean[e] = bn.mean[c]; k2[e] = bn.k2[c]; k3[e] = bn.k3[c];
        }
    }
    auto bn_apply = [&](float4 g, float4 zv) {
        float4 d;
        d.x = gi[0] * (g.x - k2[0] - (zv.x - mean[0]) * inv[0] * k3[0]);
        d.y = gi[1] * (g.y - k2[1] - (zv.y - mean[1]) * inv[1] * k3[1]);
        d.z = gi[2] * (g.z - k2[2] - (zv.z - mean[2]) * inv[2] * k3[2]);
        d.w = gi[3] * (g.w - k2[3] - (zv.w - mean[3]) * inv[3] * k3[3]);
        return d;
    };
    auto prefetch = [&](int b) {
        const float4 *src = reinterpret_cast<const float4 *>(dz + (long)b * HW * CR);
        const float4 *zs = reinterpret_cast<const float4 *>(bn.z + (long)b * HW * CR);
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int i = threadIdx.x + 256 * j;
            pf[j] = i < nf4 ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (BN) pz[j] = i < nf4 ? zs[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    if ((int)blockIdx.x < B) prefetch(blockIdx.x);
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();                                          // previous clip's reads are done
        float4 *dst = reinterpret_cast<float4 *>(dz + (long)b * HW * CR);
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int i = threadIdx.x + 256 * j;
            if (i < nf4) {
                const int pix = i / F4, c4 = i % F4, y = pix / W, x = pix % W;
                float4 v = pf[j];
                if (BN) { v = bn_apply(v, pz[j]); dst[i] = v; }
                *reinterpret_cast<float4 *>(&tile[((y + 1) * WP + x + 1) * CRP + 4 * c4]) = v;
            }
        }
        if (nf4 > 256 * PF) {                                     // larger clips: the remainder goes straight through
            const float4 *zs = reinterpret_cast<const float4 *>(bn.z + (long)b * HW * CR);
            for (int i = threadIdx.x + 256 * PF; i < nf4; i += 256) {
                const int pix = i / F4, c4 = i % F4, y = pix / W, x = pix % W;
                float4 v = dst[i];
                if (BN) { v = bn_apply(v, zs[i]); dst[i] = v; }
                *reinterpret_cast<float4 *>(&tile[((y + 1) * WP + x + 1) * CRP + 4 * c4]) = v;
            }
        }
        __syncthreads();
        if (b + (int)gridDim.x < B) prefetch(b + gridDim.x);      // next clip's HBM latency hides under this clip's MFMAs
        for (int t = wave; t < ntile; t += 4) {
            const int p = 16 * t + li, pc = p < HW ? p : HW - 1;  // A-fragment row (clamped: extra rows are not stored)
            const int ih = pc / W, iw = pc % W;
            // source pixel of tap (kh, kw) in tile coordinates: (ih + 1 - kh + 1, iw + 1 - kw + 1)
            const float *a0 = &tile[((ih + 2) * WP + iw + 2) * CRP + 4 * lq];
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int jj = 0; jj < JJ; ++jj) {
                        const float4 a = *reinterpret_cast<const float4 *>(a0 - (kh * WP + kw) * CRP + 16 * jj);
                        const float4 w = wf[kh * 3 + kw][jj];
                        acc = mfma16(a.x, w.x, acc);
                        acc = mfma16(a.y, w.y, acc);
                        acc = mfma16(a.z, w.z, acc);
                        acc = mfma16(a.w, w.w, acc);
                    }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int po = 16 * t + 4 * lq + r;
                if (po < HW) dx[((long)b * HW + po) * CO + li] = acc[r];
            }
        }
    }
}

// The conv2 data gradient on the bf16 matrix cores in the three-way split form: K = 32 of one MFMA is one tap x the 32
// output channels (k = 8 lq + j), nine k-steps.  dz (formed from g and z on the way in when BN) is split into bf16 planes
// while it is staged, laid out [plane][channel quarter][halo pixel][8]: an A fragment is one ds_read_b128 per plane and 16
// consecutive pixels fill the 64 banks.  The 9 x 3 weight fragments W[tap][n = li][8 lq ..] stay in registers (256-register
// budget, two blocks per CU).  The ten pixel tiles of a 15 x 10 map do not divide over four waves, so the tile a wave
// starts with rotates from clip to clip and every SIMD sees the same load over time.
// COMPACT: g arrives as (routed value per pool window, element index) -- bn.gw / bn.arg -- and is rebuilt while staging
// STORE_DZ = false: dz is not written back (the weight gradient forms it itself, conv_wgrad_clip_bf16_kernel<true>)
template <bool BN, bool COMPACT = false, bool STORE_DZ = true>
__global__ __launch_bounds__(256, 2) void conv_dgrad_clip_bf16_kernel(float *__restrict__ dz, const float *__restrict__ wgt,
                                                                       float *__restrict__ dx, int B, int H, int W, BnBwdArgs bn)
{
    constexpr int CR = 32, CO = 16, F4 = CR / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char ctile[];   // [3 planes][4 quarters][(H+2)(W+2)][8 bf16], zero halo
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), li = lane & 15, lq = lane >> 4;
    const int HP = H + 2, WP = W + 2, HW = H * W, NPIX = (HP * WP + 15) & ~15, ntile = (HW + 15) / 16;   // rounded: see the forward kernel
    for (int i = threadIdx.x; i < 12 * NPIX * 4; i += 256) reinterpret_cast<unsigned *>(ctile)[i] = 0u;

    bf16x8 wf[9][3];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const f32x4 w0 = *reinterpret_cast<const f32x4 *>(wgt + (t * CO + li) * CR + 8 * lq);
        const f32x4 w1 = *reinterpret_cast<const f32x4 *>(wgt + (t * CO + li) * CR + 8 * lq + 4);
        bf16x4 h0, m0, l0, h1, m1, l1;
        split_bf16(w0, h0, m0, l0);
        split_bf16(w1, h1, m1, l1);
        wf[t][0] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        wf[t][1] = __builtin_shufflevector(m0, m1, 0, 1, 2, 3, 4, 5, 6, 7);
        wf[t][2] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
    }

    constexpr int PF = 5;                                         // float4 held per thread for the next clip (<= 1280 per clip)
    const int nf4 = HW * F4;
    f32x4 pf[PF], pz[PF];
    unsigned pa[PF];                                              // COMPACT: the four element indices of the float4's window
    float gi[4], mean[4], inv[4], k2[4], k3[4];                   // BN coefficients of this thread's 4 channels (c4 = threadIdx.x % F4)
    if (BN) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * (threadIdx.x % F4) + e;
            inv[e] = bn.inv[c]; gi[e] = bn.gamma[c] * inv[e]; mean[e] = bn.mean[c];
            if (!bn.acc) { k2[e] = bn.k2[c]; k3[e] = bn.k3[c]; }
        }
        if (bn.acc) {
            bn_bwd_k_from_acc(bn, 4 * F4, threadIdx.x % F4, bn.dgamma && blockIdx.x == 0, k2, k3);
            if (bn.acc_clear_set && blockIdx.x == 0) acc_clear(bn.acc_clear_set, threadIdx.x, 256);
        }
    }
    auto bn_apply = [&](f32x4 g, f32x4 zv) {
        f32x4 d;
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = gi[e] * (g[e] - k2[e] - (zv[e] - mean[e]) * inv[e] * k3[e]);
        return d;
    };
    // compact g: window (float4 index inside the clip's window table) and element of each float4 this thread stages; -1 =
    // the pixel lies outside every pool window (odd H or W) and its g is zero
    const int Wp = W / 2, nwin = (H / 2) * Wp;
    int cq[PF], ce[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        const int i = threadIdx.x + 256 * j, pix = i / F4, c4 = i % F4, y = pix / W, xx = pix - y * W;
        const bool in = i < nf4 && y < 2 * (H / 2) && xx < 2 * Wp;
        cq[j] = in ? ((y >> 1) * Wp + (xx >> 1)) * F4 + c4 : -1;
        ce[j] = (y & 1) * 2 + (xx & 1);
    }
    auto prefetch = [&](int b) {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(dz + (long)b * HW * CR);
        const f32x4 *zs = reinterpret_cast<const f32x4 *>(bn.z + (long)b * HW * CR);
        const f32x4 *gws = reinterpret_cast<const f32x4 *>(bn.gw) + (long)b * nwin * F4;
        const unsigned *ars = reinterpret_cast<const unsigned *>(bn.arg) + (long)b * nwin * F4;
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int i = threadIdx.x + 256 * j;
            if (COMPACT) {                                        // raw loads only: the select happens when the clip is staged
                const int q = cq[j] >= 0 ? cq[j] : 0;
                pf[j] = gws[q];
                pa[j] = ars[q];
            } else {
                pf[j] = i < nf4 ? src[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            if (BN) pz[j] = i < nf4 ? zs[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stage = [&](int i, f32x4 v) {                            // float4 i of the clip = channels 4 c4.. of pixel i / 8
        const int pix = i / F4, c4 = i % F4, y = pix / W, xx = pix - y * W;
        bf16x4 h, m, l;
        split_bf16(v, h, m, l);
        unsigned char *d = ctile + (((c4 >> 1) * NPIX + (y + 1) * WP + xx + 1) * 16 + (c4 & 1) * 8);
        *reinterpret_cast<bf16x4 *>(d) = h;
        *reinterpret_cast<bf16x4 *>(d + 4 * NPIX * 16) = m;
        *reinterpret_cast<bf16x4 *>(d + 8 * NPIX * 16) = l;
    };
    if ((int)blockIdx.x < B) prefetch(blockIdx.x);
    int rot = wave;
    for (int b = blockIdx.x; b < B; b += gridDim.x, ++rot) {
        __syncthreads();                                          // previous clip's reads are done
        f32x4 *dst = reinterpret_cast<f32x4 *>(dz + (long)b * HW * CR);
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int i = threadIdx.x + 256 * j;
            if (i < nf4) {
                f32x4 v = pf[j];
                if (COMPACT) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (cq[j] >= 0 && (int)((pa[j] >> (8 * e)) & 0xFFu) == ce[j]) ? v[e] : 0.f;
                }
                if (BN) { v = bn_apply(v, pz[j]); if (STORE_DZ) dst[i] = v; }
                stage(i, v);
            }
        }
        if (nf4 > 256 * PF) {                                     // larger clips: the remainder goes straight through
            const f32x4 *zs = reinterpret_cast<const f32x4 *>(bn.z + (long)b * HW * CR);
            for (int i = threadIdx.x + 256 * PF; i < nf4; i += 256) {
                f32x4 v = dst[i];
                if (BN) { v = bn_apply(v, zs[i]); dst[i] = v; }
                stage(i, v);
            }
        }
        __syncthreads();
        if (b + (int)gridDim.x < B) prefetch(b + gridDim.x);      // next clip's HBM latency hides under this clip's MFMAs
        for (int t = rot & 3; t < ntile; t += 4) {
            const int p = 16 * t + li, pc = p < HW ? p : HW - 1;  // A-fragment row (clamped: extra rows are not stored)
            const int ih = pc / W, iw = pc % W;
            // source pixel of tap (kh, kw) in halo coordinates: (ih + 1 - kh + 1, iw + 1 - kw + 1)
            const unsigned char *a0 = ctile + (lq * NPIX + (ih + 2) * WP + iw + 2) * 16;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    bf16x8 a[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) a[pl] = *reinterpret_cast<const bf16x8 *>(a0 - (kh * WP + kw) * 16 + pl * 4 * NPIX * 16);
                    acc = mfma_bf16x6(a, wf[kh * 3 + kw], acc);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int po = 16 * t + 4 * lq + r;
                if (po < HW) dx[((long)b * HW + po) * CO + li] = acc[r];
            }
        }
    }
}

// The conv2 weight gradient on the bf16 matrix cores in the three-way split form.  The reduction runs over the clip's
// pixels (k-steps of 32, the last one padded with a zero dz row), so both operands are transposed: the clip's haloed x map
// and its dz are staged as bf16 planes in [pixel][16 channels] rows of 32 B (dz as two 16-channel halves) and fetched with
// ds_read_b64_tr_b16; every lane supplies the address of one pixel row, so the tap shift of x is an immediate offset and
// 8 consecutive pixels per 32-lane half fall into 8 different bank octets.  Wave = (column tile nt, tap parity): it keeps
// the accumulators of its 5 or 4 taps for the whole kernel and reuses each dz fragment for all of them.
// GBN: dz is not read but formed while staging from the compact routed gradient (bn.gw / bn.arg, bn_bwd_reduce_pool_kernel<true>)
// and z (BatchNorm backward: dz = gamma inv (g - k2 - xhat k3)), exactly as conv_dgrad_clip_bf16_kernel<true, true> does -- the
// weight gradient then does not wait for the data-gradient kernel and starts beside it on the side stream.
template <bool GBN = false>
__global__ __launch_bounds__(256, 2) void conv_wgrad_clip_bf16_kernel(const float *__restrict__ x, const float *__restrict__ dz,
                                                                       float *__restrict__ dw, int B, int H, int W, BnBwdArgs bn)
{
    constexpr int CIN = 16, COUT = 32, KS = 5;                    // k-steps of 32 pixels: H * W <= 160
    extern __shared__ __attribute__((aligned(16))) unsigned char ctile[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), li = lane & 15, lq = lane >> 4;
    const int nt = wave & 1, tg = wave >> 1;
    const int HP = H + 2, WP = W + 2, HW = H * W, NPIX = HP * WP;
    const int XPL = NPIX * 32, DPL = (HW + 1) * 32;               // bytes per x plane, per dz (plane, half); dz row HW stays zero
    unsigned char *Xs = ctile, *Ds = ctile + 3 * XPL;
    for (int i = threadIdx.x; i < (3 * XPL + 6 * DPL) / 4; i += 256) reinterpret_cast<unsigned *>(ctile)[i] = 0u;

    // per-lane row addresses of the transposing reads: lane 4q + pp of a 16-lane group supplies row q, channels 4pp..4pp+3;
    // group lq takes pixels 4 lq + q (first read) and 16 + 4 lq + q (second read) of the k-step
    int xaddr[KS][2], daddr[KS][2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            const int pq = 32 * ks + 16 * rd + 4 * lq + (li >> 2);
            const bool ok = pq < HW;
            const int oh = ok ? pq / W : 0, ow = ok ? pq - oh * W : 0;   // rows past the clip: any finite x row (dz row HW is zero)
            xaddr[ks][rd] = (oh * WP + ow) * 32 + (li & 3) * 8;
            daddr[ks][rd] = (ok ? pq : HW) * 32 + (li & 3) * 8 + nt * DPL;
        }
    auto frag = [&](const unsigned char *p0, const unsigned char *p1) -> bf16x8 {
        union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
        u.s.lo = lds_read_tr16(p0);
        u.s.hi = lds_read_tr16(p1);
        return u.v;
    };

    f32x4 acc[5];                                                 // taps tg, tg + 2, ...: dW[tap][ci = 4 lq + r][co = 16 nt + li]
#pragma unroll
    for (int t = 0; t < 5; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    constexpr int PX = 3, PD = 5;                                 // float4 per thread: x (<= 768 per clip), dz (<= 1280)
    const int nx4 = HW * (CIN / 4), nd4 = HW * (COUT / 4);
    f32x4 px[PX], pd[PD], pz[GBN ? PD : 1];
    unsigned pa[GBN ? PD : 1];                                    // GBN: the four element indices of the float4's pool window
    constexpr int F4 = COUT / 4;
    float gi[4], mean[4], inv[4], k2[4], k3[4];                   // GBN: BatchNorm coefficients of this thread's 4 channels
    const int Wp = W / 2, nwin = (H / 2) * Wp;
    int cq[GBN ? PD : 1], ce[GBN ? PD : 1];
    if (GBN) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * (threadIdx.x % F4) + e;
            inv[e] = bn.inv[c]; gi[e] = bn.gamma[c] * inv[e]; mean[e] = bn.mean[c];
            if (!bn.acc) { k2[e] = bn.k2[c]; k3[e] = bn.k3[c]; }
        }
        if (bn.acc) bn_bwd_k_from_acc(bn, COUT, threadIdx.x % F4, false, k2, k3);
#pragma unroll
        for (int j = 0; j < PD; ++j) {
            const int i = threadIdx.x + 256 * j, pix = i / F4, c4 = i % F4, y = pix / W, xx = pix - y * W;
            const bool in = i < nd4 && y < 2 * (H / 2) && xx < 2 * Wp;
            cq[j] = in ? ((y >> 1) * Wp + (xx >> 1)) * F4 + c4 : -1;
            ce[j] = (y & 1) * 2 + (xx & 1);
        }
    }
    auto prefetch = [&](int b) {
        const f32x4 *xs = reinterpret_cast<const f32x4 *>(x + (long)b * HW * CIN);
#pragma unroll
        for (int j = 0; j < PX; ++j) { const int i = threadIdx.x + 256 * j; px[j] = i < nx4 ? xs[i] : (f32x4){0.f, 0.f, 0.f, 0.f}; }
        if (GBN) {
            const f32x4 *zs = reinterpret_cast<const f32x4 *>(bn.z + (long)b * HW * COUT);
            const f32x4 *gws = reinterpret_cast<const f32x4 *>(bn.gw) + (long)b * nwin * F4;
            const unsigned *ars = reinterpret_cast<const unsigned *>(bn.arg) + (long)b * nwin * F4;
#pragma unroll
            for (int j = 0; j < PD; ++j) {
                const int i = threadIdx.x + 256 * j, q = cq[j] >= 0 ? cq[j] : 0;
                pd[j] = gws[q];
                pa[j] = ars[q];
                pz[j] = i < nd4 ? zs[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        } else {
            const f32x4 *ds = reinterpret_cast<const f32x4 *>(dz + (long)b * HW * COUT);
#pragma unroll
            for (int j = 0; j < PD; ++j) { const int i = threadIdx.x + 256 * j; pd[j] = i < nd4 ? ds[i] : (f32x4){0.f, 0.f, 0.f, 0.f}; }
        }
    };
    auto form_dz = [&](int j) {                                   // GBN: g of this float4 from its window's routed value, then BatchNorm backward
        f32x4 v = pd[j];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float g = (cq[j] >= 0 && (int)((pa[j] >> (8 * e)) & 0xFFu) == ce[j]) ? v[e] : 0.f;
            v[e] = gi[e] * (g - k2[e] - (pz[j][e] - mean[e]) * inv[e] * k3[e]);
        }
        return v;
    };
    auto stage_x = [&](int i, f32x4 v) {
        const int pix = i >> 2, c4 = i & 3, y = pix / W, xx = pix - y * W;
        bf16x4 h, m, l;
        split_bf16(v, h, m, l);
        unsigned char *d = Xs + ((y + 1) * WP + xx + 1) * 32 + c4 * 8;
        *reinterpret_cast<bf16x4 *>(d) = h;
        *reinterpret_cast<bf16x4 *>(d + XPL) = m;
        *reinterpret_cast<bf16x4 *>(d + 2 * XPL) = l;
    };
    auto stage_d = [&](int i, f32x4 v) {
        const int pix = i >> 3, c4 = i & 7;
        bf16x4 h, m, l;
        split_bf16(v, h, m, l);
        unsigned char *d = Ds + (c4 >> 2) * DPL + pix * 32 + (c4 & 3) * 8;
        *reinterpret_cast<bf16x4 *>(d) = h;
        *reinterpret_cast<bf16x4 *>(d + 2 * DPL) = m;
        *reinterpret_cast<bf16x4 *>(d + 4 * DPL) = l;
    };
    if ((int)blockIdx.x < B) prefetch(blockIdx.x);
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PX; ++j) { const int i = threadIdx.x + 256 * j; if (i < nx4) stage_x(i, px[j]); }
#pragma unroll
        for (int j = 0; j < PD; ++j) { const int i = threadIdx.x + 256 * j; if (i < nd4) stage_d(i, GBN ? form_dz(j) : pd[j]); }
        if (!GBN && (nx4 > 256 * PX || nd4 > 256 * PD)) {         // larger clips: the remainder goes straight through (GBN: the host checks the size)
            const f32x4 *xs = reinterpret_cast<const f32x4 *>(x + (long)b * HW * CIN);
            for (int i = threadIdx.x + 256 * PX; i < nx4; i += 256) stage_x(i, xs[i]);
            const f32x4 *ds = reinterpret_cast<const f32x4 *>(dz + (long)b * HW * COUT);
            for (int i = threadIdx.x + 256 * PD; i < nd4; i += 256) stage_d(i, ds[i]);
        }
        __syncthreads();
        if (b + (int)gridDim.x < B) prefetch(b + gridDim.x);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 bfr[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) bfr[p] = frag(Ds + 2 * p * DPL + daddr[ks][0], Ds + 2 * p * DPL + daddr[ks][1]);
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                const int tap = tg + 2 * t;                        // wave-uniform; tg = 1 has four taps
                if (tap < 9) {
                    const int toff = ((tap / 3) * WP + tap % 3) * 32;
                    bf16x8 afr[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) afr[p] = frag(Xs + p * XPL + toff + xaddr[ks][0], Xs + p * XPL + toff + xaddr[ks][1]);
                    acc[t] = mfma_bf16x6(afr, bfr, acc[t]);
                }
            }
        }
    }
    // gather the block's 9 x 16 x 32 tile in LDS (reusing the staging space) and add it with contiguous atomics
    __syncthreads();
    float *red = reinterpret_cast<float *>(ctile);
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int tap = tg + 2 * t;
        if (tap < 9)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(tap * CIN + 4 * lq + r) * COUT + 16 * nt + li] = acc[t][r];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 9 * CIN * COUT; idx += 256) atomicAdd(dw + idx, red[idx]);
}

// dW[tap][ci][co] += sum_pixels x[b][oh+kh-1][ow+kw-1][ci] * dz[b][oh][ow][co],  CIN = 16, COUT = 32, all 9 taps per wave
template <int COUT>
__global__ __launch_bounds__(256) void conv_wgrad_clip_kernel(const float *__restrict__ x, const float *__restrict__ dz,
                                                               float *__restrict__ dw, int B, int H, int W)
{
    constexpr int CIN = 16, NT = COUT / 16, XS = 16, DS = stride16(COUT);   // LDS row strides (== 16 mod 32 words)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
    const int HP = H + 2, WP = W + 2, HW = H * W, nstep = (HW + 3) / 4;
    float *xt = smem;                                             // [(H+2)][(W+2)][16], zero halo
    float *dt = smem + HP * WP * XS;                              // [4*nstep][DS], rows >= HW are zero
    for (int i = threadIdx.x; i < HP * WP * XS + 4 * nstep * DS; i += 256) smem[i] = 0.f;

    f32x4 acc[9][NT];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    constexpr int PX = 3, PD = 5;                                 // float4 per thread: x (<= 768 per clip), dz (<= 1280)
    const int nx4 = HW * (CIN / 4), nd4 = HW * (COUT / 4);
    float4 px[PX], pd[PD];
    auto prefetch = [&](int b) {
        const float4 *xs = reinterpret_cast<const float4 *>(x + (long)b * HW * CIN);
        const float4 *ds = reinterpret_cast<const float4 *>(dz + (long)b * HW * COUT);
#pragma unroll
        for (int j = 0; j < PX; ++j) { const int i = threadIdx.x + 256 * j; px[j] = i < nx4 ? xs[i] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
        for (int j = 0; j < PD; ++j) { const int i = threadIdx.x + 256 * j; pd[j] = i < nd4 ? ds[i] : make_float4(0.f, 0.f, 0.f, 0.f); }
    };
    if ((int)blockIdx.x < B) prefetch(blockIdx.x);
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const int i = threadIdx.x + 256 * j;
            if (i < nx4) {
                const int pix = i / (CIN / 4), c4 = i % (CIN / 4), y = pix / W, xx = pix % W;
                *reinterpret_cast<float4 *>(&xt[((y + 1) * WP + xx + 1) * XS + 4 * c4]) = px[j];
            }
        }
#pragma unroll
        for (int j = 0; j < PD; ++j) {
            const int i = threadIdx.x + 256 * j;
            if (i < nd4) *reinterpret_cast<float4 *>(&dt[(i / (COUT / 4)) * DS + 4 * (i % (COUT / 4))]) = pd[j];
        }
        if (nx4 > 256 * PX || nd4 > 256 * PD) {                   // larger clips: the remainder goes straight through
            const float4 *xs = reinterpret_cast<const float4 *>(x + (long)b * HW * CIN);
            for (int i = threadIdx.x + 256 * PX; i < nx4; i += 256) {
                const int pix = i / (CIN / 4), c4 = i % (CIN / 4), y = pix / W, xx = pix % W;
                *reinterpret_cast<float4 *>(&xt[((y + 1) * WP + xx + 1) * XS + 4 * c4]) = xs[i];
            }
            const float4 *ds = reinterpret_cast<const float4 *>(dz + (long)b * HW * COUT);
            for (int i = threadIdx.x + 256 * PD; i < nd4; i += 256)
                *reinterpret_cast<float4 *>(&dt[(i / (COUT / 4)) * DS + 4 * (i % (COUT / 4))]) = ds[i];
        }
        __syncthreads();
        if (b + (int)gridDim.x < B) prefetch(b + gridDim.x);
        for (int st = wave; st < nstep; st += 4) {
            const int p = 4 * st + lq, pc = p < HW ? p : HW - 1;  // rows >= HW multiply dt rows that are zero
            const int oh = pc / W, ow = pc % W;
            float bf[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bf[nt] = dt[p * DS + 16 * nt + li];
            const float *a0 = &xt[(oh * WP + ow) * XS + li];      // tap (0,0) of this pixel in halo coordinates
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float a = a0[(kh * WP + kw) * XS];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[kh * 3 + kw][nt] = mfma16(a, bf[nt], acc[kh * 3 + kw][nt]);
                }
        }
    }
    // reduce the 4 waves through LDS (reusing the tile space) and add to global memory with contiguous atomics
    float *red = smem;                                            // [4][64][4] then [16][COUT] rows
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            __syncthreads();
            *reinterpret_cast<f32x4 *>(&red[(wave * 64 + lane) * 4]) = acc[t][nt];
            __syncthreads();
            if (wave == 0) {
                const f32x4 s0 = *reinterpret_cast<const f32x4 *>(&red[(0 * 64 + lane) * 4]);
                const f32x4 s1 = *reinterpret_cast<const f32x4 *>(&red[(1 * 64 + lane) * 4]);
                const f32x4 s2 = *reinterpret_cast<const f32x4 *>(&red[(2 * 64 + lane) * 4]);
                const f32x4 s3 = *reinterpret_cast<const f32x4 *>(&red[(3 * 64 + lane) * 4]);
#pragma unroll
                for (int r = 0; r < 4; ++r) red[1024 + (4 * lq + r) * COUT + 16 * nt + li] = (s0[r] + s1[r]) + (s2[r] + s3[r]);
            }
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < 16 * COUT; idx += 256) atomicAdd(dw + (long)t * CIN * COUT + idx, red[1024 + idx]);
    }
}

}  // namespace kws
