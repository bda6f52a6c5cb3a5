// Micro-benchmarks that price the primitives the NTT / Poseidon kernels are
// built from on gfx950: integer VALU rates, Goldilocks mul/add throughput,
// LDS exchange, and strided-segment HBM bandwidth.  Diagnostic tool only.
//   hipcc --offload-arch=gfx950 -O3 -o ubench tools/ubench.hip && ./ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../xlayer_prover_amd/csrc/gl_field.hpp"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef uint64_t u64;
typedef uint32_t u32;

constexpr int ITER = 4096;
constexpr int NACC = 8;

template <int OP>
__global__ __launch_bounds__(256) void valu_kernel(u64* out, u64 seed)
{
    u64 a[NACC];
    u64 b = seed * (threadIdx.x + 1) + 0x9E3779B97F4A7C15ULL;
    for (int i = 0; i < NACC; i++) a[i] = seed + i * 0x1234567ULL + threadIdx.x;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) {
            if (OP == 0) a[i] = (u64)(u32)a[i] * (u32)b + a[i];             // v_mad_u64_u32
            else if (OP == 1) a[i] = (u32)a[i] * (u32)b;                      // v_mul_lo_u32
            else if (OP == 2) a[i] = __umulhi((u32)a[i], (u32)b);             // v_mul_hi_u32
            else if (OP == 3) a[i] = a[i] + b;                                // 64-bit add
            else if (OP == 4) a[i] = gl::mul(a[i], b);                        // full mulmod
            else if (OP == 5) a[i] = gl::add(a[i], b);                        // lazy add
            else if (OP == 6) a[i] = gl::addc(a[i], b);                       // canonical add
            else if (OP == 7) a[i] = gl::sub(a[i], b);
            else if (OP == 8) a[i] = gl::mul_pow2(a[i], 12);
            else if (OP == 9) a[i] = gl::mul_pow2(a[i], 48);
            else if (OP == 10) a[i] = __umul24((u32)a[i], (u32)b) + (u32)a[i];  // v_mad_u32_u24
            else if (OP == 11) a[i] = __builtin_amdgcn_udot4((u32)a[i], (u32)b, (u32)a[i], false);
            else if (OP == 12) a[i] = (u32)a[i] + (u32)b;                     // v_add_u32
            else if (OP == 13) a[i] = __umul64hi(a[i], b);                    // 64x64 high
            else if (OP == 14) a[i] = gl::canon(a[i] + (a[i] >> 3));
            else if (OP == 15) a[i] = __shfl_xor((u32)a[i], 1) + (u32)a[i];   // DPP / swizzle
            else if (OP == 16) a[i] = __builtin_amdgcn_ds_bpermute((int)(b & 252), (int)a[i]) + (u32)a[i];
            else if (OP == 17) a[i] = __shfl_xor((u32)a[i], 32) + (u32)a[i];  // cross-half
        }
    }
    u64 s = 0;
    for (int i = 0; i < NACC; i++) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static const char* opname[] = {"mad_u64_u32", "mul_lo_u32", "mul_hi_u32", "add_u64", "gl::mul", "gl::add(lazy)", "gl::addc(canon)",
                               "gl::sub(lazy)", "gl::mul_pow2(12)", "gl::mul_pow2(48)", "mad_u32_u24", "udot4", "add_u32", "umul64hi",
                               "canon+shift", "shfl_xor 1", "ds_bpermute", "shfl_xor 32"};

template <int OP>
void run_valu(u64* dout)
{
    int blocks = 256 * 8, threads = 256;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(valu_kernel<OP>, dim3(blocks), dim3(threads), 0, 0, dout, 12345ULL);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(valu_kernel<OP>, dim3(blocks), dim3(threads), 0, 0, dout, 12345ULL + r);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    double ops = (double)blocks * threads * ITER * NACC;
    printf("%-20s %8.3f ms  %8.2f Tops/s (lane-ops)  => %.2f lane-cycles/op @2.4GHz full-chip\n", opname[OP], ms, ops / ms * 1e-9,
           (256.0 * 4 * 32 * 2.4e9) / (ops / (ms * 1e-3)));
}

// ---------------------------------------------------------------- strided segments
// Each "segment" is SEG bytes contiguous; consecutive segments a wave touches are
// `stride` bytes apart (like one tile column of a row-major matrix).  16 B per lane.
template <bool WRITE>
__global__ __launch_bounds__(256) void seg_kernel(uint4* buf, size_t nseg_total, int seg_bytes, size_t stride_bytes, size_t ngroups, uint4* sink)
{
    // group = a run of `run` segments at stride; groups interleave so all data is touched once
    int lanes_per_seg = seg_bytes / 16;
    size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t nthreads = (size_t)gridDim.x * blockDim.x;
    size_t total_lanes = nseg_total * lanes_per_seg;
    uint4 acc = {0, 0, 0, 0};
    size_t segs_per_row = stride_bytes / seg_bytes;   // segments that fit side by side in one stride
    for (size_t i = tid; i < total_lanes; i += nthreads) {
        size_t seg = i / lanes_per_seg, l = i % lanes_per_seg;
        // consecutive seg ids walk DOWN a column of segments (stride apart); next column after nrows
        size_t nrows = nseg_total / segs_per_row;
        size_t col = seg / nrows, row = seg % nrows;
        size_t off = row * stride_bytes + col * seg_bytes + l * 16;
        uint4* p = (uint4*)((char*)buf + off);
        if (WRITE) { uint4 v = {(u32)i, (u32)seg, 3, 4}; *p = v; }
        else { uint4 v = *p; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    }
    if (!WRITE && acc.x == 0x12345 && acc.y == 0x777) sink[tid] = acc;
}

void run_seg(uint4* buf, size_t bytes, uint4* sink)
{
    int segs[] = {16, 32, 64, 128, 256, 512, 1024, 4096};
    size_t strides[] = {4096, 65536, 1 << 20};
    for (size_t stride : strides)
        for (int seg : segs) {
            if ((size_t)seg > stride) continue;
            size_t nseg = bytes / seg;
            for (int wr = 0; wr < 2; wr++) {
                hipEvent_t e0, e1;
                CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                int blocks = 256 * 8;
                auto launch = [&]() {
                    if (wr) hipLaunchKernelGGL(seg_kernel<true>, dim3(blocks), dim3(256), 0, 0, buf, nseg, seg, stride, 0, sink);
                    else hipLaunchKernelGGL(seg_kernel<false>, dim3(blocks), dim3(256), 0, 0, buf, nseg, seg, stride, 0, sink);
                };
                launch();
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                launch(); launch();
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                ms /= 2;
                printf("seg %5d B stride %8zu  %s  %8.3f ms  %7.1f GB/s\n", seg, stride, wr ? "write" : "read ", ms, bytes / ms * 1e-6);
            }
        }
}

// ---------------------------------------------------------------- LDS exchange
__global__ __launch_bounds__(1024) void lds_kernel(u64* out, int iters)
{
    extern __shared__ u64 sm[];
    int t = threadIdx.x;
    u64 v[16];
    for (int j = 0; j < 16; j++) v[j] = t * 16 + j;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 16; j++) sm[j * 1024 + t] = v[j];       // stride-1 across lanes
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = sm[((t >> 4) * 16 + j) * 16 + (t & 15)] + it;  // transposed-ish read
        __syncthreads();
    }
    u64 s = 0;
    for (int j = 0; j < 16; j++) s ^= v[j];
    out[blockIdx.x * 1024 + t] = s;
}

int main(int argc, char** argv)
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  clock=%d MHz\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000);
    u64* dout;
    CK(hipMalloc(&dout, 256 * 8 * 1024 * sizeof(u64)));
    run_valu<12>(dout); run_valu<0>(dout); run_valu<1>(dout); run_valu<2>(dout); run_valu<3>(dout);
    run_valu<13>(dout); run_valu<4>(dout); run_valu<5>(dout); run_valu<6>(dout); run_valu<7>(dout);
    run_valu<8>(dout); run_valu<9>(dout); run_valu<10>(dout); run_valu<11>(dout); run_valu<14>(dout);
    run_valu<15>(dout); run_valu<16>(dout); run_valu<17>(dout);

    {   // LDS exchange: 128 KB per WG, 1 WG/CU
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipFuncSetAttribute((const void*)lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        int iters = 200;
        hipLaunchKernelGGL(lds_kernel, dim3(256), dim3(1024), 131072, 0, dout, iters);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(lds_kernel, dim3(256), dim3(1024), 131072, 0, dout, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        double bytes = 256.0 * iters * 131072 * 2;
        printf("LDS exchange (write+read 128KB/WG): %.3f ms  %.1f TB/s aggregate\n", ms, bytes / ms * 1e-9);
    }

    size_t bytes = (size_t)2 << 30;
    uint4 *buf, *sink;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&sink, 256 * 8 * 256 * sizeof(uint4)));
    CK(hipMemset(buf, 1, bytes));
    run_seg(buf, bytes, sink);
    return 0;
}
