// Does gfx950 need wait states between a VALU instruction that writes a carry (VCC or an SGPR pair) and the VALU instruction that
// consumes it?  hipcc leaves two (s_nop 1); gl::reduce128 keeps them.  This probe runs the same sequence WITHOUT them, one wave per
// workgroup and one workgroup per launch (nothing else to fill the slots), against the C formula, on edge values and random ones.
//   hipcc --offload-arch=gfx950 -O2 tools/experiments/reduce128_hazard_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef uint64_t u64; typedef uint32_t u32;
static const u64 P = 0xFFFFFFFF00000001ULL;
__device__ __forceinline__ u64 pack(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }
__device__ u64 red_nonop(u64 lo, u64 hi)
{
    const u32 hh = (u32)(hi >> 32), hl = (u32)hi;
    u32 l0, l1; u64 b;
    asm volatile("v_sub_co_u32 %0, %2, %3, %4\n\tv_subbrev_co_u32 %1, %2, 0, %5, %2" : "=&v"(l0), "=&v"(l1), "=&s"(b) : "v"((u32)lo), "v"(hh), "v"((u32)(lo >> 32)));
    const u64 l = pack(l0, l1);
    u64 s, c;
    asm volatile("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=&v"(s), "=&s"(c) : "v"(hl), "v"(l));
    u32 r0, r1, t; u64 x, y;
    asm volatile("s_xor_b64 %[x], %[b], %[c]\n\ts_andn2_b64 %[y], %[b], %[c]\n\tv_cndmask_b32 %[t], 1, -1, %[c]\n\tv_cndmask_b32 %[t], 0, %[t], %[x]\n\t"
                 "v_cndmask_b32 %[r1], 0, -1, %[y]\n\tv_add_co_u32 %[r0], vcc, %[s0], %[t]\n\tv_addc_co_u32 %[r1], vcc, %[s1], %[r1], vcc"
                 : [r0] "=&v"(r0), [r1] "=&v"(r1), [t] "=&v"(t), [x] "=&s"(x), [y] "=&s"(y)
                 : [b] "s"(b), [c] "s"(c), [s0] "v"((u32)s), [s1] "v"((u32)(s >> 32)) : "vcc", "scc");
    return pack(r0, r1);
}
__global__ void probe(u64* out, const u64* lo, const u64* hi, int n)
{
    for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] = red_nonop(lo[i], hi[i]);
}
static u64 ref(u64 lo, u64 hi)
{
    unsigned __int128 v = ((unsigned __int128)hi << 64) | lo;
    return (u64)(v % P);
}
int main()
{
    std::vector<u64> lo, hi;
    const u64 edge[] = {0, 1, 0xFFFFFFFFULL, 0x100000000ULL, P - 1, P, P + 1, ~0ULL, 0xFFFFFFFF00000000ULL, 0x00000000FFFFFFFEULL, 0x8000000000000000ULL};
    for (u64 a : edge) for (u64 b : edge) { lo.push_back(a); hi.push_back(b); }
    u64 x = 0x9E3779B97F4A7C15ULL;
    for (int i = 0; i < (1 << 20); i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; lo.push_back(x); x ^= x << 13; x ^= x >> 7; x ^= x << 17; hi.push_back(i % 3 ? x : x >> 32); }
    const int n = (int)lo.size();
    u64 *dl, *dh, *dout;
    hipMalloc(&dl, n * 8); hipMalloc(&dh, n * 8); hipMalloc(&dout, n * 8);
    hipMemcpy(dl, lo.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dh, hi.data(), n * 8, hipMemcpyHostToDevice);
    std::vector<u64> out(n);
    long bad_total = 0;
    for (int waves = 1; waves <= 16; waves *= 4) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64 * waves), 0, 0, dout, dl, dh, n);
        hipMemcpy(out.data(), dout, n * 8, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int i = 0; i < n; i++) if (out[i] % P != ref(lo[i], hi[i])) bad++;
        printf("{\"waves_in_the_only_workgroup\": %d, \"values\": %d, \"mismatches\": %ld}\n", waves, n, bad);
        bad_total += bad;
    }
    return bad_total ? 1 : 0;
}
