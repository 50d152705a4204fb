// Issue cost of the VALU instructions the render backward is made of, on one CU: cycles per wave-instruction per SIMD at 1, 2 and 4 waves per
// SIMD (one workgroup of 256 / 512 / 1024 threads), ten independent register (pairs) per wave, s_memtime around 2 000 x 10 instructions.
// hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 2000
typedef float v2f __attribute__((ext_vector_type(2)));
#define KERNEL(NAME, DECL, BODY, SINK)                                                               \
    __global__ void NAME(float *out, unsigned long long *cyc) {                                      \
        DECL;                                                                                        \
        __syncthreads();                                                                             \
        unsigned long long t0 = __builtin_readcyclecounter();                                        \
        for (int i = 0; i < ITERS; i++) { BODY; }                                                    \
        unsigned long long t1 = __builtin_readcyclecounter();                                        \
        out[threadIdx.x] = SINK;                                                                     \
        if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;                                \
    }
#define F10 float v0 = threadIdx.x * 1e-3f, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7, v8 = v0 + 8, v9 = v0 + 9
#define P10 v2f v0 = {threadIdx.x * 1e-3f, 1.f}, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f, v4 = v0 + 4.f, v5 = v0 + 5.f, v6 = v0 + 6.f, v7 = v0 + 7.f, v8 = v0 + 8.f, v9 = v0 + 9.f
#define OPS10 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(v8), "+v"(v9)
#define REP10(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7) I(8) I(9)
#define SUMF (v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + v8 + v9)
#define SUMP (v0.x + v1.y + v2.x + v3.y + v4.x + v5.y + v6.x + v7.y + v8.x + v9.y)
#define I_FMA(n) "v_fma_f32 %" #n ", %" #n ", %" #n ", %" #n "\n"
#define I_MUL(n) "v_mul_f32 %" #n ", %" #n ", %" #n "\n"
#define I_PKFMA(n) "v_pk_fma_f32 %" #n ", %" #n ", %" #n ", %" #n "\n"
#define I_PKMUL(n) "v_pk_mul_f32 %" #n ", %" #n ", %" #n "\n"
#define I_PKADD(n) "v_pk_add_f32 %" #n ", %" #n ", %" #n "\n"
#define I_EXP(n) "v_exp_f32 %" #n ", %" #n "\n"
#define I_RCP(n) "v_rcp_f32 %" #n ", %" #n "\n"
#define I_DPP(n) "v_add_f32_dpp %" #n ", %" #n ", %" #n " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_DPPROR(n) "v_add_f32_dpp %" #n ", %" #n ", %" #n " row_ror:8 row_mask:0xf bank_mask:0x3\n"
#define I_CND(n) "v_cndmask_b32 %" #n ", %" #n ", %" #n ", vcc\n"
#define I_CMP(n) "v_cmp_lt_f32 vcc, %" #n ", %" #n "\n"
#define I_MIN(n) "v_min_f32 %" #n ", %" #n ", %" #n "\n"
KERNEL(k_fma, F10, asm volatile(REP10(I_FMA) OPS10), SUMF)
KERNEL(k_mul, F10, asm volatile(REP10(I_MUL) OPS10), SUMF)
KERNEL(k_pkfma, P10, asm volatile(REP10(I_PKFMA) OPS10), SUMP)
KERNEL(k_pkmul, P10, asm volatile(REP10(I_PKMUL) OPS10), SUMP)
KERNEL(k_pkadd, P10, asm volatile(REP10(I_PKADD) OPS10), SUMP)
KERNEL(k_exp, F10, asm volatile(REP10(I_EXP) OPS10), SUMF)
KERNEL(k_rcp, F10, asm volatile(REP10(I_RCP) OPS10), SUMF)
KERNEL(k_dpp, F10, asm volatile(REP10(I_DPP) OPS10), SUMF)
KERNEL(k_dppror, F10, asm volatile(REP10(I_DPPROR) OPS10), SUMF)
KERNEL(k_cnd, F10, asm volatile(REP10(I_CND) OPS10 : : "vcc"), SUMF)
KERNEL(k_cmp, F10, asm volatile(REP10(I_CMP) OPS10 : : "vcc"), SUMF)
#define I_CND64(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %" #n ", %10\n"
#define I_CNDK(n) "v_cndmask_b32_e64 %" #n ", 0, %" #n ", %10\n"
#define I_CMP64(n) "v_cmp_lt_f32_e64 s[20:21], %" #n ", %" #n "\n"
#define I_MAX(n) "v_max_f32 %" #n ", %" #n ", %" #n "\n"
#define I_SUB(n) "v_sub_f32 %" #n ", %" #n ", %" #n "\n"
#define I_ADDU(n) "v_add_u32 %" #n ", %" #n ", %" #n "\n"
#define I_MOV(n) "v_mov_b32 %" #n ", %" #n "\n"
#define I_FMAC(n) "v_fmac_f32 %" #n ", %" #n ", %" #n "\n"
#define I_ADD(n) "v_add_f32 %" #n ", %" #n ", %" #n "\n"
KERNEL(k_cnd64, F10; unsigned long long m = __ballot((threadIdx.x & 1) != 0), asm volatile(REP10(I_CND64) OPS10 : "s"(m)), SUMF)
KERNEL(k_cndk, F10; unsigned long long m = __ballot((threadIdx.x & 1) != 0), asm volatile(REP10(I_CNDK) OPS10 : "s"(m)), SUMF)
KERNEL(k_cmp64, F10, asm volatile(REP10(I_CMP64) OPS10 : : "s20", "s21"), SUMF)
KERNEL(k_max, F10, asm volatile(REP10(I_MAX) OPS10), SUMF)
KERNEL(k_sub, F10, asm volatile(REP10(I_SUB) OPS10), SUMF)
KERNEL(k_add, F10, asm volatile(REP10(I_ADD) OPS10), SUMF)
KERNEL(k_addu, F10, asm volatile(REP10(I_ADDU) OPS10), SUMF)
KERNEL(k_mov, F10, asm volatile(REP10(I_MOV) OPS10), SUMF)
KERNEL(k_fmac, F10, asm volatile(REP10(I_FMAC) OPS10), SUMF)
KERNEL(k_min, F10, asm volatile(REP10(I_MIN) OPS10), SUMF)
KERNEL(k_swap32, F10, asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n v_permlane32_swap_b32 %8, %9\n"
                                   "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n v_permlane32_swap_b32 %8, %9\n" OPS10), SUMF)
KERNEL(k_swap16, F10, asm volatile("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n v_permlane16_swap_b32 %8, %9\n"
                                   "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n v_permlane16_swap_b32 %8, %9\n" OPS10), SUMF)
// dependent chains (one register): latency per instruction
#define CH10(I) I(0) I(0) I(0) I(0) I(0) I(0) I(0) I(0) I(0) I(0)
KERNEL(k_fma_chain, F10, asm volatile(CH10(I_FMA) OPS10), SUMF)
KERNEL(k_pkfma_chain, P10, asm volatile(CH10(I_PKFMA) OPS10), SUMP)
KERNEL(k_exp_chain, F10, asm volatile(CH10(I_EXP) OPS10), SUMF)
KERNEL(k_dpp_chain, F10, asm volatile("s_nop 1\n" I_DPP(0) "s_nop 1\n" I_DPP(0) "s_nop 1\n" I_DPP(0) "s_nop 1\n" I_DPP(0) "s_nop 1\n" I_DPP(0) "s_nop 1\n" I_DPP(0) "s_nop 1\n" I_DPP(0) "s_nop 1\n" I_DPP(0) "s_nop 1\n" I_DPP(0) "s_nop 1\n" I_DPP(0) OPS10), SUMF)
// LDS broadcast reads: 5 x ds_read_b128 of one address per trip (what a pair record costs)
__global__ void k_ldsb(float *out, unsigned long long *cyc) {
    __shared__ float4 s[1024];
    s[threadIdx.x] = make_float4(threadIdx.x, 1, 2, 3);
    __syncthreads();
    float acc = 0.f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; i++) {
        const int j = __builtin_amdgcn_readfirstlane((i * 5 + (int)(threadIdx.x >> 6) * 37) & 1023 & ~7);
        float4 a = s[j], b = s[j + 1], c = s[j + 2], d = s[j + 3], e = s[j + 4];
        float4 f = s[j ^ 8], g = s[(j ^ 8) + 1], h = s[(j ^ 8) + 2], k = s[(j ^ 8) + 3], l = s[(j ^ 8) + 4];
        acc += a.x + b.y + c.z + d.w + e.x + f.x + g.y + h.z + k.w + l.x;
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
template <typename K>
static void run(const char *name, K k) {
    float *out; unsigned long long *cyc, h[16];
    hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 16 * 8);
    printf("%-14s", name);
    for (int nt : {256, 512, 1024}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(nt), 0, 0, out, cyc);   // warm-up
        hipLaunchKernelGGL(k, dim3(1), dim3(nt), 0, 0, out, cyc);
        hipMemcpy(h, cyc, 16 * 8, hipMemcpyDeviceToHost);
        unsigned long long mx = 0; for (int i = 0; i < nt / 64; i++) mx = h[i] > mx ? h[i] : mx;
        // per SIMD: nt / 256 waves each issued ITERS * 10 instructions in mx ticks
        printf("  %dw/SIMD: %6.2f ticks/instr/SIMD", nt / 256, (double)mx / (ITERS * 10.0 * (nt / 256)));
    }
    printf("\n");
    hipFree(out); hipFree(cyc);
}
int main() {
    run("v_fma_f32", k_fma); run("v_mul_f32", k_mul); run("v_pk_fma_f32", k_pkfma); run("v_pk_mul_f32", k_pkmul); run("v_pk_add_f32", k_pkadd);
    run("v_exp_f32", k_exp); run("v_rcp_f32", k_rcp); run("dpp quad_perm", k_dpp); run("dpp ror bankm", k_dppror); run("v_cndmask", k_cnd); run("v_cmp", k_cmp); run("v_min_f32", k_min); run("v_max_f32", k_max); run("v_sub_f32", k_sub); run("v_add_f32", k_add); run("v_add_u32", k_addu); run("v_mov_b32", k_mov); run("v_fmac_f32", k_fmac);
    run("cndmask e64", k_cnd64); run("cndmask 0,v,s", k_cndk); run("v_cmp e64", k_cmp64);
    run("permlane32sw", k_swap32); run("permlane16sw", k_swap16);
    run("fma chain", k_fma_chain); run("pk_fma chain", k_pkfma_chain); run("exp chain", k_exp_chain); run("dpp chain+nop", k_dpp_chain);
    run("ds_read_b128x10", k_ldsb);
    return 0;
}
