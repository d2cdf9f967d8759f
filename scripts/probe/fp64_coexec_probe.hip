// Probe: do FP64 vector FMAs (v_fma_f64) and FP64 matrix instructions (v_mfma_f64_16x16x4_f64) of two
// waves on ONE SIMD run beside each other, or do they share the FP64 datapath?  gfx950.
// One 512-thread workgroup per CU (two waves per SIMD); the role of a wave follows from its number.
//   hipcc --offload-arch=gfx950 -O3 -o fp64_coexec_probe fp64_coexec_probe.hip && ./fp64_coexec_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

enum Role { IDLE = 0, MFMA = 1, VFMA = 2, VCHAIN = 3, MIX = 4, LDSRT = 5, IADD = 6, MFMAGAP = 7 };

__global__ __launch_bounds__(512) void probe(int role_lo, int role_hi, int prio_hi, int iters, unsigned long long* out, double* sink)
{
    __shared__ double line[64 * 4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int role = wave < 4 ? role_lo : role_hi;
    double x = 1.0 + lane * 1e-3, y = 0.999 + lane * 1e-6;
    if (__builtin_amdgcn_readfirstlane(wave) >= 4) {
        if (prio_hi == 1) __builtin_amdgcn_s_setprio(1);
        if (prio_hi == 2) __builtin_amdgcn_s_setprio(2);
        if (prio_hi == 3) __builtin_amdgcn_s_setprio(3);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    double res = 0.0;
    if (role == MFMA) {
        d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
            }
        }
        res = a0[0] + a1[1] + a2[2] + a3[3];
    } else if (role == VFMA) {
        double c0 = x, c1 = y, c2 = x + 1, c3 = y + 1, c4 = x + 2, c5 = y + 2, c6 = x + 3, c7 = y + 3;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 = fma(c0, y, x); c1 = fma(c1, y, x); c2 = fma(c2, y, x); c3 = fma(c3, y, x);
                c4 = fma(c4, y, x); c5 = fma(c5, y, x); c6 = fma(c6, y, x); c7 = fma(c7, y, x);
            }
        }
        res = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    } else if (role == VCHAIN) {
        double c0 = x;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 32; ++u) c0 = fma(c0, y, x);
        }
        res = c0;
    } else if (role == MIX) {
        // the instruction mix of a pivot column step: 32-bit VALU + DPP + some FP64 FMAs, dependent
        double c0 = x, c1 = y;
        int k = lane;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                k = max(k, __builtin_amdgcn_update_dpp(k, k, 0xB1, 0xF, 0xF, false));
                k = max(k, __builtin_amdgcn_update_dpp(k, k, 0x4E, 0xF, 0xF, false));
                c0 = fma(c0, y, (double)(k & 1)); c1 = fma(c1, y, c0);
                k += __double2loint(c1) & 1;
            }
        }
        res = c0 + c1 + k;
    } else if (role == IADD) {
        int k = lane;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 32; ++u) k = (k * 3) ^ u;
        }
        res = k;
    } else if (role == MFMAGAP) {
        // 8 MFMAs, then an LDS read + wait (a tile's operand fetch)
        d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        for (int i = 0; i < iters; ++i) {
            const double q = line[(lane + i) & 255];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, q, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, q, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
            }
        }
        res = a0[0] + a1[1] + a2[2] + a3[3];
    } else if (role == LDSRT) {
        // LDS round trips: one lane writes a line, every lane reads it back (the pivot-row exchange)
        double c0 = x;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (lane == (i & 63)) { line[wave * 32 + u] = c0; }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                c0 = fma(c0, 0.5, line[wave * 32 + u]);
            }
        }
        res = c0;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    if (res == 123.456) sink[0] = res;
}

int main()
{
    const int nb = 256, iters = 20000;
    unsigned long long* d; double* sink;
    hipMalloc(&d, nb * 8 * 8); hipMalloc(&sink, 8);
    std::vector<unsigned long long> h(nb * 8);
    const char* names[] = {"idle", "mfma", "vfma", "vchain", "mix", "ldsrt", "iadd", "mfmagap"};
    // operations per iteration of each role
    const double ops[] = {0, 8, 32, 32, 8 * 7, 4, 64, 8};
    const int modes[][3] = {{MFMA, IDLE, 0}, {IADD, IDLE, 0}, {MFMAGAP, IDLE, 0}, {LDSRT, IDLE, 0},
                            {MFMA, VFMA, 0}, {MFMA, VFMA, 3}, {MFMA, IADD, 0}, {MFMA, IADD, 3}, {MFMA, MIX, 0}, {MFMA, MIX, 3},
                            {MFMA, LDSRT, 0}, {MFMA, LDSRT, 3}, {MFMAGAP, VFMA, 0}, {MFMAGAP, VFMA, 3}, {MFMAGAP, MIX, 3},
                            {MFMAGAP, LDSRT, 0}, {MFMAGAP, LDSRT, 3}, {IADD, LDSRT, 0}, {VFMA, LDSRT, 0}, {IADD, IADD, 0}};
    for (auto& m : modes) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(probe, dim3(nb), dim3(512), 0, 0, m[0], m[1], m[2], iters, d, sink);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double lo = 0, hi = 0;
        for (int b = 0; b < nb; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? lo : hi) += (double)h[b * 8 + w] / (nb * 4);
        printf("waves0-3 %-7s | waves4-7 %-7s prio %d : %8.1f cycles/op (lo)", names[m[0]], names[m[1]], m[2], ops[m[0]] ? lo / (iters * ops[m[0]]) : 0.0);
        if (m[1] != IDLE) printf("   %8.1f cycles/op (hi)", hi / (iters * ops[m[1]]));
        printf("\n");
    }
    return 0;
}
