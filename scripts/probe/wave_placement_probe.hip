// Probe: on which SIMD do the four waves of a 256-thread workgroup run when three workgroups share a CU
// (52 KB of LDS each)?  Prints, per workgroup, the SIMD id of waves 0..3 (HW_REG_HW_ID bits 5:4) and a census.
//   hipcc --offload-arch=gfx950 -O3 -o wave_placement_probe wave_placement_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256, 3) void probe(unsigned* out, int spin)
{
    extern __shared__ double lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    double x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = fma(x, 0.999, 1e-3);      // keep the workgroups resident together
    lds[threadIdx.x] = x;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc + (lds[(threadIdx.x + 1) & 255] == 1.5 ? 1000 : 0);
    }
}
int main()
{
    const int nb = 768 * 2;
    unsigned* d; hipMalloc(&d, nb * 4 * 2 * 4);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 52 * 1024, 0, d, 200000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nb * 8);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<std::string, int> census;
    for (int b = 0; b < nb; ++b) {
        char key[64]; int s[4];
        for (int w = 0; w < 4; ++w) s[w] = (h[(b * 4 + w) * 2] >> 4) & 3;
        snprintf(key, sizeof key, "%d%d%d%d", s[0], s[1], s[2], s[3]);
        census[key]++;
        if (b < 12) {
            unsigned hw = h[b * 8];
            printf("wg %3d: simd of waves 0-3 = %s  wave slots %u %u %u %u  cu %u sh %u se %u xcc %u\n", b, key,
                   h[b * 8] & 15, h[b * 8 + 2] & 15, h[b * 8 + 4] & 15, h[b * 8 + 6] & 15,
                   (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, h[b * 8 + 1]);
        }
    }
    for (auto& kv : census) printf("pattern %s: %d workgroups\n", kv.first.c_str(), kv.second);
    return 0;
}
