// Host -> HBM upload rates for the witness (cm1_n, N x 751 row-major, prover.cpp:435 leaves it in host memory):
//   (a) one contiguous hipMemcpyAsync from pinned memory (row bands),
//   (b) hipMemcpy2DAsync of a 64-column band (512-byte pieces, pitch 751*8) -- what a per-column-chunk upload needs,
//   (c) a copy kernel that reads the mapped pinned buffer itself (same band shape),
//   (d) pageable memory through hipMemcpy (what xlp_copy_h2d does on a calloc'ed slab).
// hipcc -O3 --offload-arch=gfx950 tools/ubench_h2d.hip -o gpurun_out/ubench_h2d && gpurun_out/ubench_h2d [rows_log2]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// band copy: rows x wc words out of a row-major [rows][W] host matrix into a dense [rows][wc] device matrix
__global__ void band_copy(uint64_t* __restrict__ dst, const uint64_t* __restrict__ src, uint64_t rows, uint32_t W, uint32_t c0, uint32_t wc)
{
    const uint64_t total = rows * wc;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / wc; const uint32_t c = (uint32_t)(i % wc);
        dst[i] = __builtin_nontemporal_load(&src[r * W + c0 + c]);
    }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
    const int lg = argc > 1 ? atoi(argv[1]) : 20;
    const uint64_t rows = 1ULL << lg, W = 751, wc = 64;
    const size_t bytes = rows * W * 8;
    CK(hipSetDevice(0));
    uint64_t *h = NULL, *d = NULL, *dband = NULL;
    double t0 = now();
    CK(hipHostMalloc((void**)&h, bytes, hipHostMallocDefault));
    printf("hipHostMalloc %.2f GB: %.3f s\n", bytes / 1e9, now() - t0);
    t0 = now();
    memset(h, 1, bytes);
    printf("first touch (memset): %.3f s\n", now() - t0);
    CK(hipMalloc((void**)&d, bytes));
    CK(hipMalloc((void**)&dband, rows * wc * 8));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0, s));
        CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("(a) contiguous pinned, %.2f GB: %.2f ms = %.1f GB/s\n", bytes / 1e9, ms, bytes / ms / 1e6);
    }
    // chunked (64 MiB pieces) to see the per-call overhead
    {
        const size_t piece = 64ULL << 20;
        CK(hipEventRecord(e0, s));
        for (size_t o = 0; o < bytes; o += piece) CK(hipMemcpyAsync((char*)d + o, (char*)h + o, bytes - o < piece ? bytes - o : piece, hipMemcpyHostToDevice, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("(a') contiguous pinned in 64 MiB pieces: %.2f ms = %.1f GB/s\n", ms, bytes / ms / 1e6);
    }
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0, s));
        CK(hipMemcpy2DAsync(dband, wc * 8, h, W * 8, wc * 8, rows, hipMemcpyHostToDevice, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("(b) 2D band %llu x 512 B (pitch %llu): %.2f ms = %.1f GB/s\n", (unsigned long long)rows, (unsigned long long)(W * 8), ms, rows * wc * 8 / ms / 1e6);
    }
    for (int grid : {256, 1024, 4096}) {
        uint64_t* hd = NULL;
        CK(hipHostGetDevicePointer((void**)&hd, h, 0));
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0, s));
            band_copy<<<grid, 256, 0, s>>>(dband, hd, rows, (uint32_t)W, 64, (uint32_t)wc);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("(c) kernel band copy from mapped host, grid %d: %.2f ms = %.1f GB/s\n", grid, ms, rows * wc * 8 / ms / 1e6);
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0, s));
            band_copy<<<grid, 256, 0, s>>>(d, hd, rows, (uint32_t)W, 0, (uint32_t)W);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("(c') kernel full-width copy from mapped host, grid %d: %.2f ms = %.1f GB/s\n", grid, ms, bytes / ms / 1e6);
    }
    {
        uint64_t* p = (uint64_t*)malloc(bytes);
        memset(p, 2, bytes);
        t0 = now();
        CK(hipMemcpy(d, p, bytes, hipMemcpyHostToDevice));
        const double t = now() - t0;
        printf("(d) pageable hipMemcpy: %.2f ms = %.1f GB/s\n", t * 1e3, bytes / t / 1e9);
        t0 = now();
        CK(hipHostRegister(p, bytes, hipHostRegisterDefault));
        printf("hipHostRegister of %.2f GB: %.3f s\n", bytes / 1e9, now() - t0);
        CK(hipEventRecord(e0, s));
        CK(hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("(d') registered memory: %.2f ms = %.1f GB/s\n", ms, bytes / ms / 1e6);
        CK(hipHostUnregister(p));
        free(p);
    }
    // D2H of the same size
    CK(hipEventRecord(e0, s));
    CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s));
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("(e) D2H contiguous pinned: %.2f ms = %.1f GB/s\n", ms, bytes / ms / 1e6);
    return 0;
}
