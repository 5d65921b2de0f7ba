// coissue_probe.hip -- do instruction classes of DIFFERENT waves on one SIMD overlap on gfx950?
// Every workgroup has 8 waves (2 per SIMD).  Waves 0-3 run instruction stream A, waves 4-7 stream B, so each
// SIMD holds equally many A- and B-waves.  For every pair (A, B) the kernel time is compared with the
// A-only and B-only times at the same occupancy: "sum" = no overlap, "max" = full overlap.
//   hipcc --offload-arch=gfx950 -O3 tools/coissue_probe.hip -o tools/bin/coissue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int kChains = 8, kUnroll = 4;
enum Op { NONE = 0, XOR, FMA, MAD64, CVT, SIN, LOG, SQRT, BITOP3, PKFMA, F64FMA, NOPS };
const char* kNames[] = {"none", "v_xor", "v_fma", "v_mad_u64", "v_cvt", "v_sin", "v_log", "v_sqrt", "v_bitop3", "v_pk_fma", "v_fma_f64"};

template <int OP>
__device__ __forceinline__ void stream(int iters, unsigned seed, unsigned& acc)
{
    if (OP == NONE) return;
    unsigned v[kChains];
    float f[kChains];
    double d2[kChains];
    unsigned k1 = 0xD2511F53u, k2 = 0x9E3779B9u;
    float c1 = 1.0000001f, c2 = 1e-9f;
    for (int c = 0; c < kChains; c++) { v[c] = threadIdx.x * 2654435761u + c + seed; f[c] = 0.3f + 0.01f * (float)((threadIdx.x + c + seed) & 15); d2[c] = (double)f[c]; }
    double dk1 = 1.0000000001, dk2 = 1e-12;
    asm volatile("" : "+v"(dk1), "+v"(dk2));
    asm volatile("" : "+v"(k1), "+v"(k2), "+v"(c1), "+v"(c2));
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
#pragma unroll
            for (int c = 0; c < kChains; c++) {
                if (OP == XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[c]) : "v"(k1));
                if (OP == BITOP3) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(v[c]) : "v"(k1), "v"(k2));
                if (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[c]) : "v"(c1), "v"(c2));
                if (OP == MAD64) { unsigned long long d; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d) : "v"(v[c]), "v"(k1) : "vcc"); v[c] = (unsigned)(d >> 32); }
                if (OP == CVT) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f[c]) : "v"(v[c]));
                if (OP == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d2[c]) : "v"(dk1), "v"(dk2));
                if (OP == F64FMA) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d2[c]) : "v"(dk1), "v"(dk2));
                if (OP == SIN) asm volatile("v_sin_f32 %0, %0" : "+v"(f[c]));
                if (OP == LOG) asm volatile("v_log_f32 %0, %0" : "+v"(f[c]));
                if (OP == SQRT) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[c]));
            }
        }
    }
    for (int c = 0; c < kChains; c++) acc += v[c] + __float_as_uint(f[c]) + (unsigned)__double_as_longlong(d2[c]);
}

template <int A, int B>
__global__ void __launch_bounds__(512) k_pair(unsigned* out, int iters, unsigned seed)
{
    unsigned acc = 0;
    if ((threadIdx.x >> 6) < 4) stream<A>(iters, seed, acc);      // wave-uniform branch
    else stream<B>(iters, seed, acc);
    if (acc == 0x12345678u) out[0] = acc;
}

typedef void (*Kern)(unsigned*, int, unsigned);
template <int A, int B> Kern get() { return k_pair<A, B>; }

int main(int argc, char** argv)
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double clock_hz = prop.clockRate * 1e3;
    const int wg_per_cu = argc > 1 ? atoi(argv[1]) : 4;            // 4 workgroups x 8 waves = 8 waves/SIMD
    const int iters = argc > 2 ? atoi(argv[2]) : 1000;
    unsigned* d_out;
    CHECK(hipMalloc(&d_out, 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto time = [&](Kern k) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k, dim3(cus * wg_per_cu), dim3(512), 0, 0, d_out, iters, (unsigned)rep);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float t;
            CHECK(hipEventElapsedTime(&t, e0, e1));
            if (rep) best = std::min(best, t);
        }
        return best;
    };
    struct Row { const char* a; const char* b; Kern ab, a0, b0; };
#define ROW(A, B) {kNames[A], kNames[B], get<A, B>(), get<A, NONE>(), get<NONE, B>()}
    Row rows[] = {
        ROW(XOR, XOR), ROW(XOR, FMA), ROW(XOR, MAD64), ROW(XOR, CVT), ROW(XOR, SIN), ROW(BITOP3, SIN), ROW(FMA, SIN),
        ROW(XOR, PKFMA), ROW(PKFMA, PKFMA), ROW(MAD64, PKFMA), ROW(CVT, PKFMA), ROW(SIN, PKFMA), ROW(XOR, F64FMA), ROW(CVT, CVT),
        ROW(MAD64, MAD64), ROW(MAD64, CVT), ROW(MAD64, SIN), ROW(MAD64, LOG), ROW(CVT, SIN), ROW(SIN, SIN), ROW(SIN, LOG), ROW(SIN, SQRT),
    };
    const double instr = (double)iters * kChains * kUnroll;   // per wave and stream
    printf("%d workgroups/CU x 8 waves (half run A, half run B), %d instr per wave; cycles per instruction PAIR (one A + one B) per SIMD\n",
           wg_per_cu, (int)instr);
    printf("%-10s %-10s %10s %10s %10s %10s   %s\n", "A", "B", "A alone", "B alone", "A with B", "sum", "overlap = (sum - AB) / min(A, B)");
    for (const Row& r : rows) {
        const double waves_per_simd_each = wg_per_cu * 4 / 4.0;   // A-waves per SIMD
        auto cyc = [&](float ms) { return ms * 1e-3 * clock_hz / (instr * waves_per_simd_each); };
        const double ta = cyc(time(r.a0)), tb = cyc(time(r.b0)), tab = cyc(time(r.ab));
        printf("%-10s %-10s %10.2f %10.2f %10.2f %10.2f   %5.2f\n", r.a, r.b, ta, tb, tab, ta + tb, (ta + tb - tab) / std::min(ta, tb));
    }
    return 0;
}
