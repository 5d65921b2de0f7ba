// gen_rocrand_vectors.cpp -- writes tests/golden/rocrand_vectors.json from rocRAND 4.2's OWN
// host-callable device API (/opt/rocm/include/rocrand, third-party, not the reference):
// rocrand_init(seed, subsequence, 0) + rocrand4()/rocrand_normal4() on the CPU.  The oracle and the
// GPU kernels must reproduce these words bit for bit and these normals to float rounding.
// Build+run (no GPU needed):  hipcc -O2 gen_rocrand_vectors.cpp -o /tmp/gen_rr && /tmp/gen_rr > rocrand_vectors.json
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_kernel.h>
#include <cstdio>

int main()
{
    const unsigned long long seeds[] = {0ull, 1234ull, 20240613ull, 0xdeadbeefcafef00dull};
    const unsigned long long subs[] = {0ull, 1ull, 2ull, 3ull, 1048575ull, 0x100000000ull, 0xfedcba9876543210ull};
    printf("{\n  \"source\": \"rocRAND %d host-callable rocrand4/rocrand_normal4\",\n  \"vectors\": [\n", ROCRAND_VERSION);
    bool first = true;
    for (unsigned long long seed : seeds)
        for (unsigned long long sub : subs) {
            rocrand_state_philox4x32_10 su, sn;
            rocrand_init(seed, sub, 0, &su);
            rocrand_init(seed, sub, 0, &sn);
            printf("%s    {\"seed\": %llu, \"subsequence\": %llu, \"words\": [", first ? "" : ",\n", seed, sub);
            first = false;
            for (int b = 0; b < 3; b++) {
                uint4 w = rocrand4(&su);
                printf("%s%u, %u, %u, %u", b ? ", " : "", w.x, w.y, w.z, w.w);
            }
            printf("], \"normals\": [");
            for (int b = 0; b < 3; b++) {
                float4 g = rocrand_normal4(&sn);
                printf("%s%.9g, %.9g, %.9g, %.9g", b ? ", " : "", g.x, g.y, g.z, g.w);
            }
            printf("]}");
        }
    printf("\n  ]\n}\n");
    return 0;
}
