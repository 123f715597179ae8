// tools/valu_calib.hip -- calibration microbenchmark (diagnostic, not product): what a CU sustains of the instruction kinds the featurizer is
// made of, at W waves per SIMD, every CU busy.  Answers "what is the vector-ALU issue floor of N wave-instructions" (VERDICT r2, item 2a):
// cycles per wave-instruction per SIMD for independent v_fma_f32, for the cross-lane moves a register transpose would use
// (v_permlane32_swap, v_permlane16_swap, DPP row moves), and LDS cycles per CU for ds_write_b64 / ds_read_b64 / ds_bpermute_b32.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_calib.hip -o tools/valu_calib.bin && tools/valu_calib.bin
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kUnroll = 64;

enum Kind { FMA = 0, PERM32 = 1, PERM16 = 2, DPP_ROR8 = 3, DPP_QUAD = 4, BPERM = 5, LDS_W64 = 6, LDS_R64 = 7, LDS_W32 = 8, LDS_R128 = 9, PKFMA = 10, MIX_FMA_LDS = 11 };

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, int seed)
{
    __shared__ float2 lds[4][72 * 8 + 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float r[8];
    for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 1e-3f + i + seed;
    float2 *base = &lds[wave][lane];
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)base;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < kUnroll / 8; ++u) {
            if constexpr (KIND == FMA) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(r[(i + 1) & 7]), "v"(r[(i + 2) & 7]));
            } else if constexpr (KIND == PKFMA) {
                // 4 packed ops on register pairs = the arithmetic of 8 scalar fmas
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 *p = reinterpret_cast<f2 *>(r);
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(p[(i + 1) & 3]), "v"(p[(i + 2) & 3]));
            } else if constexpr (KIND == PERM32) {
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r[i]), "+v"(r[i + 1]));
                    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r[i + 1]), "+v"(r[i]));
                }
            } else if constexpr (KIND == PERM16) {
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(r[i]), "+v"(r[i + 1]));
                    asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(r[i + 1]), "+v"(r[i]));
                }
            } else if constexpr (KIND == DPP_ROR8) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xc" : "+v"(r[i]) : "v"(r[(i + 1) & 7]));
            } else if constexpr (KIND == DPP_QUAD) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r[i]) : "v"(r[(i + 1) & 7]));
            } else if constexpr (KIND == BPERM) {
                const int a = 4 * (lane ^ 37);
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(r[i]) : "v"(a), "v"(r[(i + 1) & 7]));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if constexpr (KIND == LDS_W64) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(*reinterpret_cast<double *>(&r[(i & 3) * 2])), "n"(i * 576) : "memory");
            } else if constexpr (KIND == LDS_W32) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(r[i]), "n"(i * 576) : "memory");
            } else if constexpr (KIND == LDS_R64) {
                double d[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d[i]) : "v"(addr), "n"(i * 576) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])::"memory");
                r[u & 7] += (float)d[u & 7];
            } else if constexpr (KIND == LDS_R128) {
                typedef float f4 __attribute__((ext_vector_type(4)));
                f4 d[4];
                const unsigned a16 = addr & ~15u;
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d[i]) : "v"(a16), "n"(i * 1152) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3])::"memory");
                r[u & 7] += d[u & 3][0];
            } else if constexpr (KIND == MIX_FMA_LDS) {
                // the featurizer's rough mix per 8 slots: 8 fma + 1 ds_write_b64 + 1 ds_read_b64
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(r[(i + 1) & 7]), "v"(r[(i + 2) & 7]));
                double d;
                asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(*reinterpret_cast<double *>(&r[0])) : "memory");
                asm volatile("ds_read_b64 %0, %1 offset:576" : "=v"(d) : "v"(addr) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d)::"memory");
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += r[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + lds[wave][lane].x;
}

// Do the vector ALU and the LDS pipe overlap when DIFFERENT waves of a SIMD use them?  Blocks alternate roles: even blocks issue only
// v_fma_f32 (n_fma per iteration), odd blocks only the LDS instruction (8 per iteration).  Time alone vs together tells max() from sum().
template <int LKIND>
__global__ __launch_bounds__(256) void k_roles(float *out, int iters, int seed, int role_mask)
{
    __shared__ float2 lds[4][72 * 8 + 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float r[8];
    for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 1e-3f + i + seed;
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)&lds[wave][lane];
    const int role = blockIdx.x & 1;
    if (!((role_mask >> role) & 1)) { out[blockIdx.x * 256 + threadIdx.x] = 0.f; return; }
    if (role == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(r[(i + 1) & 7]), "v"(r[(i + 2) & 7]));
        }
    } else {
        for (int it = 0; it < iters; ++it) {
            if constexpr (LKIND == LDS_W64) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(*reinterpret_cast<double *>(&r[(i & 3) * 2])), "n"(i * 576) : "memory");
            } else if constexpr (LKIND == LDS_R64) {
                double d[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d[i]) : "v"(addr), "n"(i * 576) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])::"memory");
            } else {
                const int a = 4 * (lane ^ 37);
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(r[i]) : "v"(a), "v"(r[(i + 1) & 7]));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += r[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + lds[wave][lane].x;
}

template <int LKIND>
void run_roles(const char *name, int blocks_per_cu, int cus, int iters_fma_scale)
{
    const int nblk = cus * blocks_per_cu, iters = 4000;
    float *out;
    hipMalloc(&out, sizeof(float) * nblk * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[4] = {0, 0, 0, 0};
    for (int mask = 1; mask <= 3; ++mask) {
        hipLaunchKernelGGL((k_roles<LKIND>), dim3(nblk), dim3(256), 0, 0, out, iters, 1, mask);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_roles<LKIND>), dim3(nblk), dim3(256), 0, 0, out, iters, 2, mask);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[mask], e0, e1);
    }
    printf("roles %-18s blocks/CU %d (half fma, half LDS): fma alone %.3f ms, LDS alone %.3f ms, together %.3f ms (max %.3f, sum %.3f)\n", name, blocks_per_cu,
           ms[1], ms[2], ms[3], ms[1] > ms[2] ? ms[1] : ms[2], ms[1] + ms[2]);
    hipFree(out);
}

template <int KIND>
void run(const char *name, int blocks_per_cu, int cus, double clock_ghz, int inst_per_8)
{
    const int nblk = cus * blocks_per_cu, iters = 4000;
    float *out;
    hipMalloc(&out, sizeof(float) * nblk * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND>), dim3(nblk), dim3(256), 0, 0, out, iters, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND>), dim3(nblk), dim3(256), 0, 0, out, iters, 2);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: blocks_per_cu waves per SIMD (a block = 4 waves, one per SIMD)
    const double inst_per_simd = (double)blocks_per_cu * iters * (kUnroll / 8) * inst_per_8;
    const double cyc = ms * 1e-3 * clock_ghz * 1e9;
    printf("%-22s waves/SIMD %d : %8.3f ms  %6.2f cycles per wave-instruction per SIMD  (%6.2f per CU)\n", name, blocks_per_cu, ms, cyc / inst_per_simd,
           cyc / inst_per_simd / 4.0);
    hipFree(out);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double ghz = p.clockRate * 1e-6;
    printf("device: %s, %d CUs, clockRate %.3f GHz (cycles below assume it)\n", p.name, cus, ghz);
    for (int w : {2, 4, 8}) {
        run_roles<LDS_W64>("ds_write_b64", w, cus, 1);
        run_roles<LDS_R64>("ds_read_b64", w, cus, 1);
        run_roles<BPERM>("ds_bpermute_b32", w, cus, 1);
    }
    for (int w : {1, 2, 4, 6, 8}) {
        run<FMA>("v_fma_f32", w, cus, ghz, 8);
        run<PKFMA>("v_pk_fma_f32", w, cus, ghz, 4);
        run<PERM32>("v_permlane32_swap", w, cus, ghz, 8);
        run<PERM16>("v_permlane16_swap", w, cus, ghz, 8);
        run<DPP_ROR8>("v_mov_dpp row_ror:8", w, cus, ghz, 8);
        run<DPP_QUAD>("v_mov_dpp quad_perm", w, cus, ghz, 8);
        run<BPERM>("ds_bpermute_b32", w, cus, ghz, 8);
        run<LDS_W64>("ds_write_b64", w, cus, ghz, 8);
        run<LDS_W32>("ds_write_b32", w, cus, ghz, 8);
        run<LDS_R64>("ds_read_b64", w, cus, ghz, 8);
        run<LDS_R128>("ds_read_b128", w, cus, ghz, 4);
        run<MIX_FMA_LDS>("8 fma + w64 + r64", w, cus, ghz, 10);
    }
    return 0;
}
