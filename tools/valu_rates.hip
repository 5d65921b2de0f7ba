// valu_rates.hip -- measures the issue cost of the VALU instructions the Hull-White path kernel
// is made of, on the GPU it runs on (MI355X / gfx950).  Output: cycles per wave64 instruction per
// SIMD at full occupancy (8 waves/SIMD, every CU busy), i.e. the denominators of the VALU
// roofline in DESIGN.md.  Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rates.hip -o tools/bin/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int kChains = 8;      // independent dependency chains per thread
constexpr int kUnroll = 16;     // instructions per chain per loop iteration

#define DEFINE_KERNEL(NAME, DECL, INIT, ASM_LINE, SINK)                                   \
    __global__ void NAME(unsigned* out, int iters, unsigned seed)                        \
    {                                                                                     \
        DECL;                                                                             \
        INIT;                                                                             \
        for (int it = 0; it < iters; it++) {                                              \
            _Pragma("unroll") for (int u = 0; u < kUnroll; u++) {                         \
                _Pragma("unroll") for (int c = 0; c < kChains; c++) { ASM_LINE; }         \
            }                                                                             \
        }                                                                                 \
        unsigned acc = 0;                                                                 \
        _Pragma("unroll") for (int c = 0; c < kChains; c++) { SINK; }                     \
        if (acc == 0x12345679u) out[0] = acc;                                             \
    }

// 32-bit float / int ops on v[c]
#define DECL_F  float v[kChains]; float k1 = 1.0000001f, k2 = 1e-9f
#define INIT_F  for (int c = 0; c < kChains; c++) v[c] = 1.0f + (float)((threadIdx.x + c + seed) & 15) * 0.01f
#define SINK_F  acc += __float_as_uint(v[c])
#define DECL_U  unsigned v[kChains]; unsigned k1 = 0xD2511F53u
#define INIT_U  for (int c = 0; c < kChains; c++) v[c] = threadIdx.x * 2654435761u + c + seed
#define SINK_U  acc += v[c]

DEFINE_KERNEL(k_fma_f32, DECL_F, INIT_F, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(k1), "v"(k2)), SINK_F)
DEFINE_KERNEL(k_add_f32, DECL_F, INIT_F, asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[c]) : "v"(k2)), SINK_F)
DEFINE_KERNEL(k_mul_f32, DECL_F, INIT_F, asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[c]) : "v"(k1)), SINK_F)
DEFINE_KERNEL(k_log_f32, DECL_F, INIT_F, asm volatile("v_log_f32 %0, %0" : "+v"(v[c])), SINK_F)
DEFINE_KERNEL(k_exp_f32, DECL_F, INIT_F, asm volatile("v_exp_f32 %0, %0" : "+v"(v[c])), SINK_F)
DEFINE_KERNEL(k_sin_f32, DECL_F, INIT_F, asm volatile("v_sin_f32 %0, %0" : "+v"(v[c])), SINK_F)
DEFINE_KERNEL(k_cos_f32, DECL_F, INIT_F, asm volatile("v_cos_f32 %0, %0" : "+v"(v[c])), SINK_F)
DEFINE_KERNEL(k_sqrt_f32, DECL_F, INIT_F, asm volatile("v_sqrt_f32 %0, %0" : "+v"(v[c])), SINK_F)
DEFINE_KERNEL(k_rcp_f32, DECL_F, INIT_F, asm volatile("v_rcp_f32 %0, %0" : "+v"(v[c])), SINK_F)
DEFINE_KERNEL(k_cvt_f32_u32, DECL_U, INIT_U, asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(v[c])), SINK_U)
DEFINE_KERNEL(k_xor_b32, DECL_U, INIT_U, asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[c]) : "v"(k1)), SINK_U)
DEFINE_KERNEL(k_add_u32, DECL_U, INIT_U, asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[c]) : "v"(k1)), SINK_U)
DEFINE_KERNEL(k_mul_lo_u32, DECL_U, INIT_U, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[c]) : "v"(k1)), SINK_U)
DEFINE_KERNEL(k_mul_hi_u32, DECL_U, INIT_U, asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v[c]) : "v"(k1)), SINK_U)
DEFINE_KERNEL(k_mul_u32_u24, DECL_U, INIT_U, asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[c]) : "v"(k1)), SINK_U)
DEFINE_KERNEL(k_mul_hi_u32_u24, DECL_U, INIT_U, asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(v[c]) : "v"(k1)), SINK_U)
DEFINE_KERNEL(k_mad_u32_u24, DECL_U, INIT_U, asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(v[c]) : "v"(k1)), SINK_U)

// v_mad_u64_u32 D(64) = S0(32) * S1(32) + S2(64); the chain feeds the HIGH half back as S0
// (a sub-register pick, no extra instruction).
#define MAD64 { unsigned long long d; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d) : "v"(v[c]), "v"(k1) : "vcc"); v[c] = (unsigned)(d >> 32); }
DEFINE_KERNEL(k_mad_u64_u32, DECL_U, INIT_U, MAD64, SINK_U)
#define DECL_D double v[kChains]; double k1 = 1.0000000001, k2 = 1e-12
#define INIT_D for (int c = 0; c < kChains; c++) v[c] = 1.0 + (double)((threadIdx.x + c + seed) & 15) * 0.01
#define SINK_D acc += (unsigned)__double_as_longlong(v[c])
DEFINE_KERNEL(k_fma_f64, DECL_D, INIT_D, asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[c]) : "v"(k1), "v"(k2)), SINK_D)
DEFINE_KERNEL(k_add_f64, DECL_D, INIT_D, asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[c]) : "v"(k2)), SINK_D)
DEFINE_KERNEL(k_pk_fma_f32, DECL_D, INIT_D, asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(k1), "v"(k2)), SINK_D)
DEFINE_KERNEL(k_pk_add_f32, DECL_D, INIT_D, asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[c]) : "v"(k2)), SINK_D)

// operand-shape variants: does an SGPR / literal source, or a destination different from the
// sources, change the issue cost?
#define DECL_U2 unsigned v[kChains], t[kChains]; unsigned k1 = 0xD2511F53u, k2 = 0x9E3779B9u
#define INIT_U2 for (int c = 0; c < kChains; c++) { v[c] = threadIdx.x * 2654435761u + c + seed; t[c] = 0; }
#define XOR_3REG { asm volatile("v_xor_b32 %0, %1, %2" : "=&v"(t[c]) : "v"(v[c]), "v"(k1)); asm volatile("v_xor_b32 %0, %1, %2" : "=&v"(v[c]) : "v"(t[c]), "v"(k2)); }
DEFINE_KERNEL(k_xor_3reg_x2, DECL_U2, INIT_U2, XOR_3REG, SINK_U)
#define DECL_US unsigned v[kChains]; unsigned k1 = seed * 3u + 0xD2511F53u
DEFINE_KERNEL(k_xor_sgpr, DECL_US, INIT_U, asm volatile("v_xor_b32 %0, %1, %0" : "+v"(v[c]) : "s"(k1)), SINK_U)
DEFINE_KERNEL(k_xor_literal, DECL_U, INIT_U, asm volatile("v_xor_b32 %0, 0x9E3779B9, %0" : "+v"(v[c])), SINK_U)
// gfx950 addition: arbitrary three-input bitwise function; truth table 0x96 = a ^ b ^ c
DEFINE_KERNEL(k_bitop3_xor3, DECL_U2, INIT_U2, asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(v[c]) : "v"(k1), "v"(k2)), SINK_U)
DEFINE_KERNEL(k_bitop3_xor3_sgpr, DECL_US, INIT_U, asm volatile("v_bitop3_b32 %0, %0, %1, %0 bitop3:0x96" : "+v"(v[c]) : "s"(k1)), SINK_U)
#define MAD64S { unsigned long long d; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d) : "v"(v[c]), "s"(k1) : "vcc"); v[c] = (unsigned)(d >> 32); }
DEFINE_KERNEL(k_mad_u64_sgpr, DECL_US, INIT_U, MAD64S, SINK_U)
#define MAD64X { unsigned long long d; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d) : "v"(v[c]), "s"(k1) : "vcc"); v[c] = (unsigned)(d >> 32) ^ (unsigned)d; }
DEFINE_KERNEL(k_mad_u64_then_xor, DECL_US, INIT_U, MAD64X, SINK_U)
DEFINE_KERNEL(k_fmamk_literal, DECL_F, INIT_F, asm volatile("v_fmamk_f32 %0, %0, 0x2f800000, %1" : "+v"(v[c]) : "v"(k2)), SINK_F)
DEFINE_KERNEL(k_mul_literal, DECL_F, INIT_F, asm volatile("v_mul_f32 %0, 0x3f800001, %0" : "+v"(v[c])), SINK_F)
#define DECL_FS float v[kChains]; float k1 = 1.0000001f + (float)seed * 1e-12f
DEFINE_KERNEL(k_fmac_sgpr, DECL_FS, INIT_F, asm volatile("v_fmac_f32 %0, %1, %0" : "+v"(v[c]) : "s"(k1)), SINK_F)

typedef void (*kern_t)(unsigned*, int, unsigned);

struct Entry { const char* name; kern_t fn; };

int main(int argc, char** argv)
{
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    const double clock_hz = prop.clockRate * 1e3;
    printf("device: %s (%s)  CUs=%d  clock=%.0f MHz  wave=%d\n", prop.name, prop.gcnArchName, cus, clock_hz / 1e6, prop.warpSize);
    unsigned* d_out;
    CHECK(hipMalloc(&d_out, 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 8;
    const int block = 256, grid = cus * waves_per_simd;     // 4 waves per block -> waves_per_simd blocks per CU
    const int iters = argc > 2 ? atoi(argv[2]) : 2000;
    Entry tab[] = {
        {"v_fma_f32", k_fma_f32}, {"v_add_f32", k_add_f32}, {"v_mul_f32", k_mul_f32}, {"v_pk_fma_f32", k_pk_fma_f32},
        {"v_pk_add_f32", k_pk_add_f32}, {"v_fma_f64", k_fma_f64}, {"v_add_f64", k_add_f64},
        {"v_log_f32", k_log_f32}, {"v_exp_f32", k_exp_f32}, {"v_sin_f32", k_sin_f32}, {"v_cos_f32", k_cos_f32},
        {"v_sqrt_f32", k_sqrt_f32}, {"v_rcp_f32", k_rcp_f32}, {"v_cvt_f32_u32", k_cvt_f32_u32},
        {"v_xor_b32", k_xor_b32}, {"v_add_u32", k_add_u32}, {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_hi_u32", k_mul_hi_u32},
        {"v_mad_u64_u32", k_mad_u64_u32}, {"v_mul_u32_u24", k_mul_u32_u24}, {"v_mul_hi_u32_u24", k_mul_hi_u32_u24},
        {"v_mad_u32_u24", k_mad_u32_u24},
        {"xor d!=a!=b (x2)", k_xor_3reg_x2}, {"xor sgpr,v", k_xor_sgpr}, {"xor literal,v", k_xor_literal},
        {"bitop3 a^b^c", k_bitop3_xor3}, {"bitop3 a^sgpr^a", k_bitop3_xor3_sgpr},
        {"mad_u64 v*sgpr", k_mad_u64_sgpr}, {"mad_u64+xor(hi,lo)", k_mad_u64_then_xor},
        {"fmamk literal", k_fmamk_literal}, {"mul literal", k_mul_literal}, {"fmac sgpr", k_fmac_sgpr},
    };
    printf("grid=%d blocks x %d threads (%d waves/SIMD), %d iters x %d instr/thread\n", grid, block, waves_per_simd, iters,
           kChains * kUnroll);
    printf("%-18s %10s %14s %16s\n", "instruction", "ms(min)", "Ginstr-lanes/s", "cyc/wave-instr/SIMD");
    for (const Entry& en : tab) {
        std::vector<float> ms;
        for (int rep = 0; rep < 6; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(en.fn, dim3(grid), dim3(block), 0, 0, d_out, iters, (unsigned)rep);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float t;
            CHECK(hipEventElapsedTime(&t, e0, e1));
            if (rep > 0) ms.push_back(t);
        }
        const double tmin = *std::min_element(ms.begin(), ms.end());
        const double wave_instr = (double)grid * (block / 64) * (double)iters * kChains * kUnroll;
        const double lanes_per_s = wave_instr * 64 / (tmin * 1e-3);
        const double cyc = tmin * 1e-3 * clock_hz * (cus * 4.0) / wave_instr;
        printf("%-18s %10.3f %14.1f %16.2f\n", en.name, tmin, lanes_per_s / 1e9, cyc);
    }
    return 0;
}
