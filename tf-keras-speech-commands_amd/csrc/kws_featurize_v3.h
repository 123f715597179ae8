// csrc/kws_featurize_v3.h -- third-generation featurizer kernel for the default frame geometry (n_fft = window = 1024, hop = 512),
// compile-time band / coefficient counts.  Same arithmetic as its predecessors (sonopy.mfcc_spec as restated in
// common/bark_feature.py:75-89,156-175 and inference/tflite/mfcc.h:214-290); what changed is how the data moves through the LDS.
//
// Measured on MI355X (tools/valu_calib.hip): the second-generation kernel was bound by the LDS pipe, not by
// vector-ALU issue -- v_fma_f32 sustains one wave-instruction per 2 cycles per SIMD with >= 2 waves, while per CU a ds_write_b64 costs 6
// cycles, a ds_write_b32 4, a ds_bpermute_b32 6, a ds_read_b64 2.2, a ds_read_b128 4.2: ~310 LDS cycles per frame against ~250 SIMD-cycles
// of arithmetic per SIMD-share, 62 % of the kernel's time with the LDS pipe busy.  The cheapest store the hardware has is
// ds_write_addtid_b32 (address = M0 + offset + 4 * lane, no address register: 2 cycles per dword), so this kernel arranges every LDS store
// of the frame loop to be LANE-LINEAR:
//   * the two transposes of the 8 x 8 x 8 FFT write split re / im planes, plane = the digit just transformed, position = lane; the lane
//     <-> (digit, digit) maps of the three passes are chosen so that every reader then needs 8 CONSECUTIVE floats of one plane (two
//     ds_read_b128 per component): n = 64 a + 8 b + c, pass 1 lane = b + 8 c, pass 2 lane = c + 8 A, pass 3 lane = A + 8 B, output
//     k = A + 8 B + 64 C in register C.  The planes start at 64 A + 4 ((A + 2) >> 2) floats: with that padding the 16-lane groups a
//     ds_read_b128 is served in touch all 64 banks once.  An exchange costs 16 x 2 + 4 x 4.2 = 49 LDS cycles instead of 8 x 6 + 8 x 2.2 = 66;
//   * pass 3 leaves lane l with Z[l + 64 C], so the power spectrum of bins l + 64 i is lane-linear too, and the mirrored bins
//     512 - l - 64 i go to a second plane in DESCENDING bin order, which costs the band gather nothing (the host reverses the weights of the
//     chunks that read there): 9 addtid stores (18 cycles) instead of 9 ds_write_b32 with computed addresses (36);
//   * the scale 1/4 (real-FFT split) x 1/n_fft (power) = 2^-12 is folded into the band weights and into the energy sum (exact: a power
//     of two), which takes ~24 multiplies out of the split.
// Pass 1's lanes own the input points 8 (l & 7) + (l >> 3) + 64 a, i.e. a wave's sample loads are strided by 64 B between lanes inside
// the same 512 B: as many cache lines per instruction, more tag look-ups (the address path is far from its limit here).
#pragma once

// Every multiply-add of this file is written out (fmaf) and the compiler's own contraction is off: the kernel has two instantiations per
// input type (twiddles from LDS / from registers) that must give the same bits (tests/test_featurizer_gpu.py), and left to itself the
// compiler fuses the same expression differently in different surroundings.
#pragma clang fp contract(off)

namespace kws {

// Explicit single-width LDS reads.  The compiler merges neighbouring ds_read_b64 into ds_read2_b64 / ds_read2st64_b64, which the
// LDS serves at HALF the rate (8 array cycles for two reads against 2 + 2, MI355X_MICROARCH.md section LDS), and this kernel
// is bound by LDS cycles.  The reads of the
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

__device__ __forceinline__ float2 cmul3(float2 a, float2 b)
{
    return make_float2(fmaf(-a.y, b.y, a.x * b.x), fmaf(a.y, b.x, a.x * b.y));
}
// forward 8-point DFT in registers, natural order in and out (the first-generation dft8 with its multiply-adds spelled out)
__device__ __forceinline__ void dft8x(float2 (&v)[8])
{
    const float h = 0.70710678118654752440f;
    float2 a0 = cadd(v[0], v[4]), a1 = csub(v[0], v[4]);
    float2 a2 = cadd(v[2], v[6]), a3 = mul_mi(csub(v[2], v[6]));
    float2 a4 = cadd(v[1], v[5]), a5 = csub(v[1], v[5]);
    float2 a6 = cadd(v[3], v[7]), a7 = mul_mi(csub(v[3], v[7]));
    float2 b0 = cadd(a0, a2), b2 = csub(a0, a2), b1 = cadd(a1, a3), b3 = csub(a1, a3);
    float2 b4 = cadd(a4, a6), b6 = csub(a4, a6), b5 = cadd(a5, a7), b7 = csub(a5, a7);
    const float s5 = b5.x + b5.y, d5 = b5.y - b5.x;                   // b5 (1 - i) / sqrt2 = h (s5, d5)
    const float d7 = b7.y - b7.x, s7 = b7.x + b7.y;                   // b7 (-1 - i) / sqrt2 = h (d7, -s7)
    float2 t2 = mul_mi(b6);                                           // b6 * (-i)
    v[0] = cadd(b0, b4); v[4] = csub(b0, b4);
    v[1] = make_float2(fmaf(h, s5, b1.x), fmaf(h, d5, b1.y)); v[5] = make_float2(fmaf(-h, s5, b1.x), fmaf(-h, d5, b1.y));
    v[2] = cadd(b2, t2); v[6] = csub(b2, t2);
    v[3] = make_float2(fmaf(h, d7, b3.x), fmaf(-h, s7, b3.y)); v[7] = make_float2(fmaf(-h, d7, b3.x), fmaf(h, s7, b3.y));
}

constexpr int kV3Waves = 12;
constexpr int kV3Comp = 520;                 // floats per component (8 planes of 64 + padding)
constexpr int kV3Tile = 2 * kV3Comp;         // floats per wave: re planes, im planes
constexpr int kV3OffM = 320;                 // the mirrored half of the power spectrum: position kV3OffM + p holds bin 512 - p, p = 0..255
__host__ __device__ constexpr int v3_plane(int A) { return 64 * A + 4 * ((A + 2) >> 2); }
// pass-1 lane l owns the input points sigma(l) + 64 a
__host__ __device__ inline int v3_sigma(int l) { return 8 * (l & 7) + (l >> 3); }
// position of bin k in the wave's power planes
__host__ __device__ inline int v3_pos_of_bin(int k) { return k <= 256 ? k : kV3OffM + 512 - k; }

typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_wait4(f32x4v (&a)[4])
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])::"memory");
}
template <int OFF> __device__ __forceinline__ f32x4v lds_rd128v(unsigned addr)
{
    f32x4v d;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
    return d;
}

// the 16 lane-linear stores of one FFT transpose: plane A of each component at byte offset 4 * v3_plane(A) (+ 4 * kV3Comp for im)
static_assert(4 * v3_plane(1) == 256 && 4 * v3_plane(2) == 528 && 4 * v3_plane(3) == 784 && 4 * v3_plane(4) == 1040 && 4 * v3_plane(5) == 1296 &&
              4 * v3_plane(6) == 1568 && 4 * v3_plane(7) == 1824 && 4 * kV3Comp == 2080, "offsets spelled out in v3_store_planes");
__device__ __forceinline__ void v3_store_planes(const float2 (&v)[8], unsigned m0)
{
    asm volatile("s_mov_b32 m0, %8\n\ts_nop 0\n\t"
                 "ds_write_addtid_b32 %0 offset:0\n\t"
                 "ds_write_addtid_b32 %1 offset:256\n\t"
                 "ds_write_addtid_b32 %2 offset:528\n\t"
                 "ds_write_addtid_b32 %3 offset:784\n\t"
                 "ds_write_addtid_b32 %4 offset:1040\n\t"
                 "ds_write_addtid_b32 %5 offset:1296\n\t"
                 "ds_write_addtid_b32 %6 offset:1568\n\t"
                 "ds_write_addtid_b32 %7 offset:1824"
                 :: "v"(v[0].x), "v"(v[1].x), "v"(v[2].x), "v"(v[3].x), "v"(v[4].x), "v"(v[5].x), "v"(v[6].x), "v"(v[7].x), "s"(m0) : "memory");
    asm volatile("s_mov_b32 m0, %8\n\ts_nop 0\n\t"
                 "ds_write_addtid_b32 %0 offset:2080\n\t"
                 "ds_write_addtid_b32 %1 offset:2336\n\t"
                 "ds_write_addtid_b32 %2 offset:2608\n\t"
                 "ds_write_addtid_b32 %3 offset:2864\n\t"
                 "ds_write_addtid_b32 %4 offset:3120\n\t"
                 "ds_write_addtid_b32 %5 offset:3376\n\t"
                 "ds_write_addtid_b32 %6 offset:3648\n\t"
                 "ds_write_addtid_b32 %7 offset:3904"
                 :: "v"(v[0].y), "v"(v[1].y), "v"(v[2].y), "v"(v[3].y), "v"(v[4].y), "v"(v[5].y), "v"(v[6].y), "v"(v[7].y), "s"(m0) : "memory");
}
// 8 consecutive complex values of plane (lane >> 3) from position 8 (lane & 7): two ds_read_b128 per component
__device__ __forceinline__ void v3_issue_reads(f32x4v (&t)[4], unsigned a_rd)
{
    t[0] = lds_rd128v<0>(a_rd);
    t[1] = lds_rd128v<16>(a_rd);
    t[2] = lds_rd128v<4 * kV3Comp>(a_rd);
    t[3] = lds_rd128v<4 * kV3Comp + 16>(a_rd);
}
__device__ __forceinline__ void v3_unpack(float2 (&v)[8], const f32x4v (&t)[4])
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[j] = make_float2(t[0][j], t[2][j]);
        v[j + 4] = make_float2(t[1][j], t[3][j]);
    }
}
// power spectrum: pk[i] -> position lane + 64 i, bin 256 (lane 0's value; the other lanes' land in unused positions 257..319) -> 256 + lane,
// pm[i] -> position kV3OffM + lane + 64 i
__device__ __forceinline__ void v3_store_power(const float (&pk)[4], const float (&pm)[4], float p256, unsigned m0)
{
    static_assert(4 * kV3OffM == 1280, "offsets spelled out below");
    asm volatile("s_mov_b32 m0, %9\n\ts_nop 0\n\t"
                 "ds_write_addtid_b32 %0 offset:0\n\t"
                 "ds_write_addtid_b32 %1 offset:256\n\t"
                 "ds_write_addtid_b32 %2 offset:512\n\t"
                 "ds_write_addtid_b32 %3 offset:768\n\t"
                 "ds_write_addtid_b32 %4 offset:1024\n\t"
                 "ds_write_addtid_b32 %5 offset:1280\n\t"
                 "ds_write_addtid_b32 %6 offset:1536\n\t"
                 "ds_write_addtid_b32 %7 offset:1792\n\t"
                 "ds_write_addtid_b32 %8 offset:2048"
                 :: "v"(pk[0]), "v"(pk[1]), "v"(pk[2]), "v"(pk[3]), "v"(p256), "v"(pm[0]), "v"(pm[1]), "v"(pm[2]), "v"(pm[3]), "s"(m0) : "memory");
}

// WAVES: waves per block.  The per-lane twiddles (7 + 7 + 4 complex values that never change for a lane) live in 36 registers instead of
// being re-read from LDS for every frame (18 ds_read_b64 = 39 LDS cycles per frame, in a kernel bound by the LDS pipe): 110 registers,
// 4 waves per SIMD.
template <typename WavT, int CHP, int NF, int NO, int WAVES = kV3Waves>
__global__ __launch_bounds__(WAVES * 64, 4) void featurize_fft1024_v3_kernel(const WavT *__restrict__ wav, int64_t stride,
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

    float *s_tile = reinterpret_cast<float *>(smem) + wave * kV3Tile;     // FFT planes; the power spectrum aliases them
    float *s_part = reinterpret_cast<float *>(smem) + kWaves * kV3Tile + wave * kPerWave;
    float *s_mel = s_part + TB * 64;
    float *s_en = s_mel + 64;
    float *s_dct = reinterpret_cast<float *>(smem) + kWaves * kV3Tile + kWaves * kPerWave;   // [NF][NO]
    float *s_w = s_dct + NF * NO;                                          // [64 lanes][CHP]
    float2 *s_tw1 = reinterpret_cast<float2 *>(s_w + 64 * CHP);           // [7][64]  W_512^(sigma(lane) A)
    float2 *s_tw2 = s_tw1 + 7 * 64;                                        // [7][8]   W_64^(c B)
    float2 *s_tws = s_tw2 + 7 * 8;                                         // [4][64]  W_1024^(lane + 64 i)
    int *s_bcs = reinterpret_cast<int *>(s_tws + 4 * 64);                  // [NF + 1]

    for (int i = tid; i < NF * NO; i += kThreads) s_dct[i] = c.dct[i];
    for (int i = tid; i < 64 * CHP; i += kThreads) s_w[i] = c.w3[i];
    for (int i = tid; i < 7 * 64; i += kThreads) s_tw1[i] = c.tw1s[i];
    for (int i = tid; i < 7 * 8; i += kThreads) s_tw2[i] = c.tw2[i];
    for (int i = tid; i < 4 * 64; i += kThreads) s_tws[i] = c.tws3[i];
    for (int i = tid; i <= NF; i += kThreads) s_bcs[i] = c.bcs3[i];
    for (int i = lane; i < kV3Tile; i += 64) s_tile[i] = 0.f;              // gather windows may run past a chunk (zero weights): finite data everywhere
    s_mel[lane] = 0.f;
    __syncthreads();

    const int lo = lane & 7;
    const int n2 = v3_sigma(lane);                                         // pass 1: this lane's input points are n2 + 64 a
    const int partner = 4 * ((64 - lane) & 63);                            // ds_bpermute byte address of the lane that holds Z[512 - k]
    const bool lane0 = lane == 0;
    const int chunk_pack = c.chunks3[lane].y | (c.chunks3[lane].z << 16);  // first position read (even) | slot of the partial sum

    // LDS byte addresses of this lane's reads (constant for the whole kernel)
    const unsigned m0 = __builtin_amdgcn_readfirstlane(lds_addr(s_tile));
    const unsigned a_tw1 = lds_addr(s_tw1 + lane), a_tw2 = lds_addr(s_tw2 + lo), a_tws = lds_addr(s_tws + lane);
    const unsigned a_rd = lds_addr(s_tile + v3_plane(lane >> 3) + 8 * lo);
    const unsigned a_pw = lds_addr(s_tile + (chunk_pack & 0xFFFF)), a_w = lds_addr(s_w + lane * CHP);

    f32x2 tw1r[7], tw2r[7], twsr[4];
    LdsRow<7, 64 * 8>::template go<0>(tw1r, a_tw1);
    LdsRow<7, 8 * 8>::template go<0>(tw2r, a_tw2);
    LdsRow<4, 64 * 8>::template go<0>(twsr, a_tws);
    lds_wait(tw1r);
    lds_wait(tw2r);
    lds_wait(twsr);
    // Persistent waves: a JOB = c.fpw consecutive frames of one clip; the jobs of the batch are dealt round-robin to the grid's waves
    const int njobs = B * c.jpc;
  for (int job = (int)blockIdx.x * kWaves + wave; job < njobs; job += (int)gridDim.x * kWaves) {
    const int b = job / c.jpc;
    // clip geometry: keep the head, left-pad zeros (data_utils.py:77-80)
    const int bc = b;
    const int row = c.index ? c.index[bc] : bc;                            // kws_featurize_gather: clip bc is a row of the caller's dataset
    int len = valid_len ? valid_len[row] : (stride > c.max_samples ? c.max_samples : (int)stride);
    len = len < 0 ? 0 : len;
    if ((int64_t)len > stride) len = (int)stride;
    if (len > c.max_samples) len = c.max_samples;
    const int pad = c.max_samples - len;
    const WavT *src = wav + (int64_t)row * stride;
    const bool vec_ok = ((pad & 1) == 0) && ((reinterpret_cast<uintptr_t>(src) & (2 * sizeof(WavT) - 1)) == 0);

    float *dst = feat + (int64_t)bc * c.n_frames * NO;

    const int f_beg = (job - b * c.jpc) * c.fpw;
    const int f_end = f_beg + c.fpw < c.n_frames ? f_beg + c.fpw : c.n_frames;

    float2 xl[4], xh[4];                // lower / upper half of the next frame to transform
    if (f_beg < f_end) {
        load_half<WavT, 0>(xl, src, f_beg * 512, pad, 1024, vec_ok, n2);
        load_half<WavT, 4>(xh, src, f_beg * 512, pad, 1024, vec_ok, n2);
    }
    int qi = 0;                          // frames waiting in this wave's tail batch
    for (int f = f_beg; f < f_end; ++f) {
        float2 v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = xl[j]; v[j + 4] = xh[j]; }
        if (f + 1 < f_end) {             // next frame: its lower half is this frame's upper half; the new half loads under the FFT
#pragma unroll
            for (int j = 0; j < 4; ++j) xl[j] = xh[j];
            load_half<WavT, 4>(xh, src, (f + 1) * 512, pad, 1024, vec_ok, n2);
        }
        // pass 1: DFT-8 over a (n = n2 + 64 a), twiddle W_512^(n2 A)
        dft8x(v);
#pragma unroll
        for (int k = 1; k < 8; ++k) v[k] = cmul3(v[k], make_float2(tw1r[k - 1].x, tw1r[k - 1].y));
        wave_sync();
        v3_store_planes(v, m0);          // plane A, position lane (= b + 8 c)
        wave_sync();
        {
            f32x4v t[4];
            v3_issue_reads(t, a_rd);     // plane A = lane >> 3, positions 8 c + b, c = lane & 7
            lds_wait4(t);
            v3_unpack(v, t);
        }
        // pass 2: lane = c + 8 A; DFT-8 over b, twiddle W_64^(c B)
        dft8x(v);
#pragma unroll
        for (int k = 1; k < 8; ++k) v[k] = cmul3(v[k], make_float2(tw2r[k - 1].x, tw2r[k - 1].y));
        wave_sync();
        v3_store_planes(v, m0);          // plane B, position lane (= c + 8 A)
        wave_sync();
        {
            f32x4v t[4];
            v3_issue_reads(t, a_rd);     // plane B = lane >> 3, positions 8 A + c, A = lane & 7
            lds_wait4(t);
            v3_unpack(v, t);
        }
        // pass 3: lane = A + 8 B; DFT-8 over c -> register C holds Z[lane + 64 C]
        dft8x(v);

        // partner bins Z[512 - (lane + 64 i)] = Z[(64 - lane) + 64 (7 - i)], i = 0..3: register 7 - i of lane 64 - lane; lane 0 is its
        // own partner with Z[512 - 64 i] = Z[64 ((8 - i) & 7)]: rotate its four source registers by one
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
        // real-FFT split without its factors 1/2: 2 X[k] = E + W_1024^k O, 2 conj(X[512-k]) = E - W_1024^k O with E = Z[k] + conj(Z[512-k]),
        // O = -i (Z[k] - conj(Z[512-k])); the powers below are 4 n_fft = 2^12 times bark_feature.py:88-89's, which the band weights and the
        // energy sum undo exactly
        float pk[4], pm[4], energy = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float2 zk = v[i], zq = zm[i];
            const float2 E = make_float2(zk.x + zq.x, zk.y - zq.y);
            const float2 O = make_float2(zk.y + zq.y, zq.x - zk.x);
            const float2 T = cmul3(make_float2(twsr[i].x, twsr[i].y), O);
            const float2 xp = cadd(E, T), xm = csub(E, T);
            pk[i] = fmaf(xp.y, xp.y, xp.x * xp.x);
            pm[i] = fmaf(xm.y, xm.y, xm.x * xm.x);
            energy += pk[i] + pm[i];
        }
        const float p256 = 4.0f * fmaf(v[4].y, v[4].y, v[4].x * v[4].x);    // bin 256 is its own partner (lane 0's value is the bin)
        energy += lane0 ? p256 : 0.f;
        wave_sync();                        // the FFT planes are dead: the power spectrum takes their place
        v3_store_power(pk, pm, p256, m0);
        energy = wave_sum_dpp(energy) * 0x1p-12f;
        wave_sync();

        // sparse band gather: lane = one chunk (<= CHP positions from an even position) of one band's non-zero span
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
