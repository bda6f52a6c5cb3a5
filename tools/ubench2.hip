// Micro-benchmarks of the field layer as the kernels use it (gl_field.hpp, ntt_common.hpp).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../xlayer_prover_amd/csrc/gl_field.hpp"
#include "../xlayer_prover_amd/csrc/ntt_common.hpp"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef uint64_t u64;
constexpr int ITER = 2048;
template <int OP>
__global__ __launch_bounds__(256) void k(u64* out, u64 seed)
{
    u64 a[16];
    u64 b = gl::canon(seed * (threadIdx.x + 1) + 0x9E3779B97F4A7C15ULL);
    for (int i = 0; i < 16; i++) a[i] = gl::canon(seed + i * 0x1234567ULL + threadIdx.x);
    for (int it = 0; it < ITER; it++) {
        if (OP == 10) { xlp::dft_regs<4>(a); }
        else if (OP == 11) { xlp::dft_regs<3>(*(u64(*)[8])a); xlp::dft_regs<3>(*(u64(*)[8])(a + 8)); }
        else {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (OP == 0) a[i] = gl::addc(a[i], b);
                else if (OP == 1) a[i] = gl::subc(a[i], b);
                else if (OP == 2) a[i] = gl::mul(a[i], b);
                else if (OP == 3) a[i] = gl::mulc(a[i], b);
                else if (OP == 4) a[i] = gl::canon(gl::mul_pow2c<12>(a[i]));
                else if (OP == 5) a[i] = gl::canon(gl::mul_pow2c<48>(a[i]));
                else if (OP == 6) a[i] = gl::canon(gl::mul_pow2c<72>(a[i]));
                else if (OP == 7) a[i] = gl::canon(a[i] + 3);
                else if (OP == 8) { u64 t = gl::addc(a[i], a[(i + 1) & 15]); a[i] = gl::subc(a[i], a[(i + 1) & 15]); a[(i + 1) & 15] = t; }
            }
        }
    }
    u64 s = 0;
    for (int i = 0; i < 16; i++) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
static const char* names[] = {"addc", "subc", "mul", "mulc", "pow2<12>+canon", "pow2<48>+canon", "pow2<72>+canon", "canon(+3)", "addc+subc pair", "", "dft16 (per dft)", "2x dft8"};
template <int OP> void run(u64* d, int per_iter)
{
    int blocks = 256 * 4, threads = 256;   // 4 waves/SIMD
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 12345ULL); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 12345ULL + r);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    double ops = (double)blocks * threads * ITER * per_iter;
    printf("%-18s %8.3f ms  %8.3f Tops/s  => %7.2f lane-cycles/op @2.4GHz\n", names[OP], ms, ops / ms * 1e-9, (256.0 * 4 * 32 * 2.4e9) / (ops / (ms * 1e-3)));
}
int main()
{
    u64* d; CK(hipMalloc(&d, 256 * 4 * 256 * 8));
    run<0>(d, 16); run<1>(d, 16); run<8>(d, 16); run<2>(d, 16); run<3>(d, 16); run<4>(d, 16); run<5>(d, 16); run<6>(d, 16); run<7>(d, 16);
    run<10>(d, 1); run<11>(d, 1);
    return 0;
}
