// Probe of the v_mfma_f64_4x4x4_4b_f64 fragment maps: D lanes hit by a single A lane x a single B lane.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned long long* out)
{
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            const unsigned long long m = __ballot(d != 0.0);
            if (lane == 0) out[la * 64 + lb] = m;
        }
}
int main()
{
    unsigned long long* d; hipMalloc(&d, 4096 * 8);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    static unsigned long long h[4096];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // for every A lane: which B lanes interact, and where the product lands
    for (int la = 0; la < 64; ++la) {
        printf("A%02d:", la);
        for (int lb = 0; lb < 64; ++lb) if (h[la * 64 + lb]) {
            int ld = __builtin_ctzll(h[la * 64 + lb]);
            printf(" B%02d->D%02d%s", lb, ld, __builtin_popcountll(h[la * 64 + lb]) > 1 ? "+" : "");
        }
        printf("\n");
    }
    return 0;
}
