// Microbenchmark (round 5): the in-register 16-point Goldilocks DFT of the extendPol kernels on the MFMA pipe.
//
//   VALU  the production form (csrc/ntt_common.hpp dft_regs<4>): one DFT16 per lane, 64 canonical add/sub + 17 constant shifts,
//         then one twiddle multiplication per element (gl::mulc) -- what a radix-16 round of lde_mid / lde_top_* spends per element.
//   MFMA  v_mfma_i32_32x32x32_i8: 32 DFT16 per wave.  w16 = 2^12, so every matrix entry is +-2^s: an input word is its own eight byte
//         limbs (xor 0x80 -> signed), the constant 2^(12 j k + 8 l) mod p has one or two non-zero signed byte digits at weights 2^(8 m),
//         and the 128 x 128 byte matrix (rows = (output k, digit m), columns = (input j, limb l)) sits in 64 registers per lane.
//         The eight int32 digit sums of an output (|sum| < 2^15; a bias = 103 p spread over the accumulator's initial value makes
//         every digit a non-negative 16-bit number; the xor's offset only reaches output 0 and is part of that initial value)
//         are recombined with 10 vector instructions and 2 moves; the result is any 64-bit representative (what the next
//         multiplication or the next round's limbs need).
// Both kernels keep their data in registers (no memory traffic inside the loop) and must agree with a host reference.
// build: hipcc -O3 --offload-arch=gfx950 -I xlayer_prover_amd/csrc tools/ubench_dft16_mfma.hip -o build/ubench_dft16_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "gl_field.hpp"
#include "ntt_common.hpp"
using gl::u32;
using gl::u64;
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned __int128 u128;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// ---------------------------------------------------------------- host: field, matrix, bias
static const u64 Pm = 0xFFFFFFFF00000001ULL;
static u64 hmul(u64 a, u64 b) { return (u64)((u128)a * b % Pm); }
static u64 hpow2(unsigned e) { u64 r = 1; for (unsigned i = 0; i < e % 192; i++) r = hmul(r, 2); return r; }
static void digits_of_pow(int e_total, int d[8])
{
    int e = ((e_total % 192) + 192) % 192, sg = 1;
    if (e >= 96) { e -= 96; sg = -1; }
    for (int m = 0; m < 8; m++) d[m] = 0;
    if (e < 64) d[e / 8] += sg * (1 << (e % 8));
    else { d[(e - 32) / 8] += sg * (1 << (e % 8)); d[(e - 64) / 8] -= sg * (1 << (e % 8)); }
}
static int cdigit(int k, int m, int j, int l) { int d[8]; digits_of_pow(12 * j * k + 8 * l, d); return d[m]; }

struct Tables {
    std::vector<uint8_t> A;      // [tile 4][kchunk 4][lane 64][16]
    std::vector<int32_t> C;      // [2][lane 64][16]: accumulator start of tile 0 / the other tiles
    int bound;
};
static Tables make_tables()
{
    Tables t;
    int maxabs = 0;
    for (int k = 0; k < 16; k++) for (int m = 0; m < 8; m++) {
        int s = 0;
        for (int j = 0; j < 16; j++) for (int l = 0; l < 8; l++) s += abs(cdigit(k, m, j, l));
        if (s > maxabs) maxabs = s;
    }
    t.bound = maxabs * 128;
    // bias digits b[m] >= bound with sum b[m] 2^(8m) = T = a multiple of p
    u128 base = 0; for (int m = 0; m < 8; m++) base += (u128)t.bound << (8 * m);
    u128 tt = (base + Pm - 1) / Pm + 1, T = tt * Pm, R = T - base;
    long b[8];
    for (int m = 0; m < 7; m++) { b[m] = t.bound + (long)(R & 255); R >>= 8; }
    b[7] = t.bound + (long)R;
    // the xor's offset: x = x' + B0, B0 = 128 * sum 2^(8l); DFT(B0 * ones) = 16 B0 at k = 0
    u64 B0 = 0; for (int l = 0; l < 8; l++) B0 += 128ull << (8 * l);
    u64 K0 = hmul(16, B0 % Pm);
    t.A.assign(4 * 4 * 64 * 16, 0);
    for (int tau = 0; tau < 4; tau++) for (int q = 0; q < 4; q++) for (int lane = 0; lane < 64; lane++) {
        const int rho = lane & 31, h = lane >> 5;
        const int u = rho >> 4, mhi = (rho >> 3) & 1, hh = (rho >> 2) & 1, mlo = rho & 3;
        const int k = 4 * tau + 2 * hh + u, m = 4 * mhi + mlo;
        for (int e = 0; e < 2; e++) for (int l = 0; l < 8; l++) {
            const int j = 4 * q + 2 * h + e;
            t.A[(((tau * 4 + q) * 64 + lane) * 16) + 8 * e + l] = (uint8_t)(int8_t)cdigit(k, m, j, l);
        }
    }
    t.C.assign(2 * 64 * 16, 0);
    for (int s = 0; s < 2; s++) for (int lane = 0; lane < 64; lane++) for (int i = 0; i < 16; i++) {
        const int h = lane >> 5, u = i >> 3, m = i & 7;
        const int k = (s == 0 ? 0 : 4) + 2 * h + u;      // s = 1 stands for every tile but the first (no k = 0 there)
        long v = b[m];
        if (k == 0) v += (long)((K0 >> (8 * m)) & 255);
        t.C[(s * 64 + lane) * 16 + i] = (int32_t)v;
    }
    return t;
}

// ---------------------------------------------------------------- device
// wave-uniform multipliers the compiler must not see through (it would strength-reduce the products into shift + 64-bit add sequences,
// 15 vector instructions and 6 moves per output instead of 10)
struct RecK { u32 k8, k16, k24, eps; };
__device__ __forceinline__ RecK rec_consts()
{
    RecK k{256u, 65536u, 0x1000000u, 0xFFFFFFFFu};
    asm("" : "+s"(k.k8), "+s"(k.k16), "+s"(k.k24), "+s"(k.eps));
    return k;
}
// digits d[0..7] (each < 2^16) at weights 2^(8m) -> any 64-bit representative of the sum mod p: 10 vector instructions
__device__ __forceinline__ u64 recombine(const int* d, const RecK& K)
{
    u32 a = __umul24((u32)d[1], K.k8) + (u32)d[0];
    a = __umul24((u32)d[2], K.k16) + a;                                             // d0 + d1 2^8 + d2 2^16 < 2^32
    const u64 W0 = (u64)(u32)d[3] * K.k24 + a;                                      // < 2^41
    u32 c = __umul24((u32)d[5], K.k8) + (u32)d[4];
    c = __umul24((u32)d[6], K.k16) + c;
    const u64 W1 = (u64)(u32)d[7] * K.k24 + c;
    const u64 Z = (u64)(u32)(W1 >> 32) * K.eps + W0;                                // W0 + (W1 >> 32) 2^64 (mod p), < 2^42
    u32 cy;
    const u32 hi = __builtin_addc((u32)(Z >> 32), (u32)W1, 0u, &cy);                // + (W1 mod 2^32) 2^32; a carry is worth 2^64 = EPS
    asm("" : "+v"(cy));                                                           // (hides that cy is 0 or 1: one multiply-add instead of a select and two 64-bit adds)
    return (u64)cy * K.eps + gl::pack((u32)Z, hi);
}

template <int MODE>   // 0: DFT16 + twiddle multiplication; 1: DFT16 only; 2: xor + MFMA only (digits summed, not recombined)
__global__ __launch_bounds__(512) void dft_mfma(u64* data, const v4i* Atab, const v16i* Ctab, const u64* twp, int iters)
{
    const u32 lane = threadIdx.x & 63;
    const u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const u32 n = lane & 31, h = lane >> 5;
    v4i A[4][4];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int q = 0; q < 4; q++) A[t][q] = Atab[(t * 4 + q) * 64 + lane];
    const v16i C0 = Ctab[lane], C1 = Ctab[64 + lane];
    u64* v = data + (wave * 32 + n) * 16;                 // vector n of this wave: 16 words
    const RecK RK = rec_consts();
    u64 e[8], tw[8];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int ee = 0; ee < 2; ee++) { e[2 * q + ee] = v[4 * q + 2 * h + ee]; tw[2 * q + ee] = twp[4 * q + 2 * h + ee]; }
    for (int it = 0; it < iters; it++) {
        v4i b[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            b[q].x = (int)((u32)e[2 * q] ^ 0x80808080u); b[q].y = (int)((u32)(e[2 * q] >> 32) ^ 0x80808080u);
            b[q].z = (int)((u32)e[2 * q + 1] ^ 0x80808080u); b[q].w = (int)((u32)(e[2 * q + 1] >> 32) ^ 0x80808080u);
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            v16i acc = t == 0 ? C0 : C1;
#pragma unroll
            for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[t][q], b[q], acc, 0, 0, 0);
            int d[16];
#pragma unroll
            for (int i = 0; i < 16; i++) d[i] = acc[i];
            if constexpr (MODE == 2) {
                u32 s0 = 0, s1 = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) { s0 ^= (u32)d[i]; s1 ^= (u32)d[8 + i]; }
                e[2 * t] = gl::pack(s0, s1); e[2 * t + 1] = gl::pack(s1, s0);
            } else {
                const u64 y0 = recombine(d, RK), y1 = recombine(d + 8, RK);
                if constexpr (MODE == 0) { e[2 * t] = gl::mul(y0, tw[2 * t]); e[2 * t + 1] = gl::mul(y1, tw[2 * t + 1]); }
                else { e[2 * t] = y0; e[2 * t + 1] = y1; }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int u = 0; u < 2; u++) v[4 * t + 2 * h + u] = gl::canon(e[2 * t + u]);
}

template <int MODE>
__global__ __launch_bounds__(512) void dft_valu(u64* data, const u64* twp, int iters)
{
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64* v = data + t * 16;
    u64 x[16], tw[16];
#pragma unroll
    for (int j = 0; j < 16; j++) { x[j] = gl::canon(v[j]); tw[j] = twp[j]; }
    for (int it = 0; it < iters; it++) {
        xlp::dft_regs<4>(x);
        u64 y[16];
#pragma unroll
        for (int k = 0; k < 16; k++) y[k] = x[xlp::brev<4>(k)];
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = MODE == 0 ? gl::mulc(y[k], tw[k]) : y[k];
    }
#pragma unroll
    for (int j = 0; j < 16; j++) v[j] = x[j];
}

// ---------------------------------------------------------------- host driver
static u64 htw(int k) { return (0x9E3779B97F4A7C15ull * (u64)(k + 1)) % Pm; }
static void host_ref(std::vector<u64>& d, size_t nvec, int iters, bool twiddle)
{
    u64 W[16][16];
    for (int k = 0; k < 16; k++) for (int j = 0; j < 16; j++) W[k][j] = hpow2(12 * j * k);
    for (size_t v = 0; v < nvec; v++) {
        u64* x = &d[v * 16];
        for (int j = 0; j < 16; j++) x[j] %= Pm;
        for (int it = 0; it < iters; it++) {
            u64 y[16];
            for (int k = 0; k < 16; k++) {
                u128 s = 0;
                for (int j = 0; j < 16; j++) s += (u128)hmul(x[j], W[k][j]);
                y[k] = (u64)(s % Pm);
            }
            for (int k = 0; k < 16; k++) x[k] = twiddle ? hmul(y[k], htw(k)) : y[k];
        }
    }
}

template <class F> static float time_ms(F launch, int reps)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int r = 0; r < reps; r++) launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 64;
    const size_t nvec = (size_t)256 * 4 * 512 ;           // vectors of 16 words: 2^19 vectors = 2^23 elements
    Tables T = make_tables();
    printf("accumulator bound %d (%s 2^15)\n", T.bound, T.bound < 32768 ? "<" : ">=");
    std::vector<u64> h(nvec * 16);
    srand(7);
    for (size_t i = 0; i < h.size(); i++) h[i] = ((u64)rand() << 42) ^ ((u64)rand() << 21) ^ (u64)rand();
    for (int j = 0; j < 16; j++) { h[j] = ~0ull; h[16 + j] = 0; h[32 + j] = Pm - 1; h[48 + j] = 0x8080808080808080ull; h[64 + j] = (j & 1) ? 0x7f7f7f7f7f7f7f7full : ~0ull; }
    u64 *d0, *d1; v4i* dA; v16i* dC;
    CK(hipMalloc(&d0, h.size() * 8)); CK(hipMalloc(&d1, h.size() * 8));
    CK(hipMalloc(&dA, T.A.size())); CK(hipMalloc(&dC, T.C.size() * 4));
    CK(hipMemcpy(dA, T.A.data(), T.A.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dC, T.C.data(), T.C.size() * 4, hipMemcpyHostToDevice));
    u64 htws[16], *dT; for (int k = 0; k < 16; k++) htws[k] = htw(k);
    CK(hipMalloc(&dT, sizeof htws)); CK(hipMemcpy(dT, htws, sizeof htws, hipMemcpyHostToDevice));
    // ---- correctness: 3 iterations, with and without the twiddle multiplication, against the host on the first 4096 vectors
    for (int tw = 0; tw < 2; tw++) {
        const size_t ncheck = 4096;
        std::vector<u64> ref(h.begin(), h.begin() + ncheck * 16), g0(ncheck * 16), g1(ncheck * 16);
        host_ref(ref, ncheck, 3, tw);
        CK(hipMemcpy(d0, h.data(), ncheck * 16 * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(d1, h.data(), ncheck * 16 * 8, hipMemcpyHostToDevice));
        if (tw) { dft_valu<0><<<ncheck / 256, 256>>>(d0, dT, 3); dft_mfma<0><<<ncheck / 32 * 64 / 256, 256>>>(d1, dA, dC, dT, 3); }
        else { dft_valu<1><<<ncheck / 256, 256>>>(d0, dT, 3); dft_mfma<1><<<ncheck / 32 * 64 / 256, 256>>>(d1, dA, dC, dT, 3); }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(g0.data(), d0, ncheck * 16 * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(g1.data(), d1, ncheck * 16 * 8, hipMemcpyDeviceToHost));
        size_t bad0 = 0, bad1 = 0;
        for (size_t i = 0; i < ref.size(); i++) { bad0 += g0[i] != ref[i]; bad1 += g1[i] != ref[i]; }
        printf("check (twiddle %d): VALU mismatches %zu, MFMA mismatches %zu of %zu\n", tw, bad0, bad1, ref.size());
        if (bad1) for (size_t i = 0, shown = 0; i < ref.size() && shown < 8; i++) if (g1[i] != ref[i]) { printf("  [%zu] vec %zu k %zu got %016llx want %016llx\n", i, i / 16, i % 16, (unsigned long long)g1[i], (unsigned long long)ref[i]); shown++; }
    }
    // ---- timing
    CK(hipMemcpy(d0, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d1, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    const double el = (double)nvec * 16 * iters;
    const int gv = (int)(nvec / 256), gm = (int)(nvec / 32 * 64 / 256);
    float t;
    t = time_ms([&] { dft_valu<0><<<gv, 256>>>(d0, dT, iters); }, 5); printf("VALU dft16 + mulc        %8.3f ms  %7.2f G element-rounds/s\n", t, el / t / 1e6);
    t = time_ms([&] { dft_valu<1><<<gv, 256>>>(d0, dT, iters); }, 5); printf("VALU dft16               %8.3f ms  %7.2f G element-rounds/s\n", t, el / t / 1e6);
    t = time_ms([&] { dft_mfma<0><<<gm, 256>>>(d1, dA, dC, dT, iters); }, 5); printf("MFMA dft16 + mul         %8.3f ms  %7.2f G element-rounds/s\n", t, el / t / 1e6);
    t = time_ms([&] { dft_mfma<1><<<gm, 256>>>(d1, dA, dC, dT, iters); }, 5); printf("MFMA dft16               %8.3f ms  %7.2f G element-rounds/s\n", t, el / t / 1e6);
    t = time_ms([&] { dft_mfma<2><<<gm, 256>>>(d1, dA, dC, dT, iters); }, 5); printf("MFMA only (no recombine) %8.3f ms  %7.2f G element-rounds/s\n", t, el / t / 1e6);
    printf("(extendPol 2^23 x 8, blow-up 2 = 1.208e9 element-rounds: at R G/s its arithmetic takes 1.208 / R s)\n");
    return 0;
}
