// csrc/kws_device.h -- device-side helpers shared by the model kernels (gfx950 / wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kws {

typedef float f32x4 __attribute__((ext_vector_type(4)));


// v_mfma_f32_16x16x4_f32: exact fp32 (k-ordered fmaf chain), 64 lanes compute a 16x16 tile, K = 4.
//   A operand: lane l holds A[row = l & 15][k = l >> 4]
//   B operand: lane l holds B[k = l >> 4][col = l & 15]
//   C/D      : lane l holds D[row = 4*(l >> 4) + r][col = l & 15] in register r
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---- split-precision bf16 on the matrix cores -------------------------------------------------------------------------
// v_mfma_f32_16x16x32_bf16 (16 cycles, 16x the flops of the fp32 MFMA per cycle): lane l holds A[row l&15][k = 8(l>>4)+j]
// and B[k = 8(l>>4)+j][col l&15], j = 0..7; C/D as for the fp32 form.  An fp32 value is carried EXACTLY-to-24-bits as
// h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) (each residual is exact in fp32), and
//   a*b = ah*bh + (ah*bm + am*bh) + (ah*bl + al*bh + am*bm)        [dropped: am*bl, al*bm, al*bl < 2^-24 relative]
// accumulated in fp32, small terms first.  Six MFMAs per product = 96 cycles per 16x16x32 block against 256 for the eight
// fp32 MFMAs it replaces, at fp32-level error (the two-way split, 3 MFMAs, was measured first: 1e-5 per product, which
// the BatchNorm-backward cancellations amplified to 2e-3 on the conv1 weight gradient -- not good enough for parity).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_bf16x6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4 c)
{
    c = mfma_bf16(a[1], b[1], c);
    c = mfma_bf16(a[2], b[0], c);
    c = mfma_bf16(a[0], b[2], c);
    c = mfma_bf16(a[1], b[0], c);
    c = mfma_bf16(a[0], b[1], c);
    return mfma_bf16(a[0], b[0], c);
}
__device__ __forceinline__ void split_bf16(f32x4 v, bf16x4 &h, bf16x4 &m, bf16x4 &l)
{
    h = __builtin_convertvector(v, bf16x4);
    const f32x4 r1 = v - __builtin_convertvector(h, f32x4);
    m = __builtin_convertvector(r1, bf16x4);
    l = __builtin_convertvector(r1 - __builtin_convertvector(m, f32x4), bf16x4);
}

// Counter-based dropout keep decision, bit-identical to oracle/model_oracle.py:dropout_keep
__device__ __forceinline__ bool dropout_keep(uint32_t seed_lo, uint32_t seed_hi, uint32_t index, float rate)
{
    uint32_t h = index ^ seed_lo;
    h += seed_hi * 0x9E3779B9u;
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return (float)(h >> 8) * (1.0f / 16777216.0f) >= rate;
}

__device__ __forceinline__ float relu6f(float x) { return fminf(fmaxf(x, 0.f), 6.f); }

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Full-wave (64-lane) sum entirely in the VALU: DPP row shifts inside each 16-lane row, then row_bcast15 / row_bcast31
// across rows (the gfx9 wave64 reduction idiom).  No LDS crossbar round trips, unlike __shfl_xor (ds_bpermute).
// The total is valid in lane 63; wave_sum_dpp() returns it broadcast to every lane.
#define KWS_DPP_ADD(v, ctrl, row_mask) \
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, row_mask, 0xf, false))
__device__ __forceinline__ float wave_sum_dpp(float v)
{
    KWS_DPP_ADD(v, 0x111, 0xf);   // row_shr:1   (lanes shifted in from outside the row contribute the `old` value 0)
    KWS_DPP_ADD(v, 0x112, 0xf);   // row_shr:2
    KWS_DPP_ADD(v, 0x114, 0xf);   // row_shr:4
    KWS_DPP_ADD(v, 0x118, 0xf);   // row_shr:8   -> lane 15 of every row holds its row sum
    KWS_DPP_ADD(v, 0x142, 0xa);   // row_bcast:15 into rows 1 and 3
    KWS_DPP_ADD(v, 0x143, 0xc);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave sum
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
#undef KWS_DPP_ADD

// forward-pass arguments of the fused head kernel (kws_layers.h: head_bwd_mfma_kernel<..., FWD = true>)
struct HeadFwdArgs {
    const float *b2;
    const int32_t *labels;
    const float *class_w;
    float *probs, *loss_i_out, *correct_i_out;
    float grad_scale;
    int ignore_index;
};

// ---- BatchNorm sums without a finalize launch ------------------------------------------------------------------
// The partial-sum form (kws_layers.h: partial[(which*C + c)*kStatStride + blk] + a finalize kernel) costs a launch between every producer and consumer of the main chain (6-9 us each of pure dependency
// latency, seven per train step).  In the accumulator form the producer's blocks ADD their per-channel sums (double,
// global_atomic_add_f64) into one small set, [kAccSlots][2][C], slot = blockIdx & 7 -- with one slot the ~1000 same-address atomics of a
// persistent grid that ends in a burst serialise (+5..12 us, tools/atomic_calib.hip); with eight they cost < 1 us -- and every CONSUMER
// block derives the coefficients itself in its prologue (2 C x 8 doubles from L2, same order and arithmetic in every block, so all
// blocks and all consumer kernels see bit-identical coefficients).  One designated consumer block also writes what the finalize kernel
// wrote (the coefficient arrays, moving statistics / dgamma, dbeta) and clears the set of the OTHER parity: sets alternate between
// consecutive passes, so a set is cleared one pass before it is added to and nothing on the chain waits for a memset.  The sets live
// in the model's own device memory (ModelRes), zero at creation.  The order of the atomics is not fixed: deterministic mode keeps the
// partial-sum form.
constexpr int kAccSlots = 8, kAccDoubles = kAccSlots * 2 * 128;
__device__ __forceinline__ void acc_add(double *__restrict__ acc, int C2, int i, double v)
{
    atomicAdd(acc + (blockIdx.x & (kAccSlots - 1)) * C2 + i, v);
}
__device__ __forceinline__ double acc_sum(const double *__restrict__ acc, int C2, int i)
{
    double v[kAccSlots];
#pragma unroll
    for (int j = 0; j < kAccSlots; ++j) v[j] = acc[j * C2 + i];
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < kAccSlots; ++j) t += v[j];
    return t;
}
__device__ __forceinline__ void acc_clear(double *__restrict__ acc, int tid, int nthreads)
{
    for (int i = tid; i < kAccDoubles; i += nthreads) acc[i] = 0.0;
}

}  // namespace kws
