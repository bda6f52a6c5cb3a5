// Kernels of the fflonk prover (src/rapidsnark/fflonk_prover.c.hpp, src/rapidsnark/polynomial/*.c.hpp): everything the
// reference does per coefficient / per evaluation over the BN254 scalar field, on device-resident vectors of Fr elements (4 x uint64
// Montgomery limbs, the reference's RawFr::Element).  The transforms and the multi-exponentiations are xlp_dev_fft_bn254 / xlp_dev_msm_bn254
// (bn128.hip, bn254_msm.hip); host/fflonk_prover.hpp strings the rounds together.
//
// One lane per element (or per run of a recurrence), 32-byte vector loads, nothing staged.  A Montgomery product is ~350 vector
// instructions, so the loops of the rounds (10-60 products per element) and the linear combinations run at the vector-issue rate of the
// chip (580-630 G wave-instr/s measured, profiles/r4/fflonk_2p22_valu_instructions.txt), not at HBM's; the scans (2 products per element over
// strided views) and the evaluation passes are the memory / latency-bound ones.  The sequential recurrences of the reference -- Polynomial::divByZerofier
// (one thread per residue class: a single thread when dividing by X - y), the running products of computeZ, fastEvaluate's Horner -- are
// first-order linear recurrences and run as three-pass scans (runs of SCAN_K elements: local pass, scan of the run totals, fix-up).
#include "xlp_internal.hpp"
#include "bn254_fr.hpp"

namespace xlp {
namespace ffr {

constexpr u32 TPB = 256;
static inline unsigned blocks(u64 n) { return (unsigned)((n + TPB - 1) / TPB); }

// ------------------------------------------------------------------------------------------------ element-wise
// out[i] = toMontgomery(values[map[i]]), 0 when the index is past the table (FflonkProver::getWitness, fflonk_prover.c.hpp:615-629,
// and the toMontgomery of computeWirePolynomial :702-706); values: 4 x u64 little-endian integers, NOT Montgomery (the .wtns form)
__global__ void gather_mont_kernel(Fr* __restrict__ out, const u32* __restrict__ map, const u64 nmap, const Fr* __restrict__ vals, const u64 nvals)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nmap) return;
    const u32 s = map[i];
    gst(&out[i], s < nvals ? to_mont(gld(&vals[s])) : zero());
}
__global__ void from_mont_kernel(Fr* __restrict__ out, const Fr* __restrict__ in, const u64 n)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) gst(&out[i], from_mont(gld(&in[i])));
}
__global__ void mul_kernel(Fr* __restrict__ out, const Fr* __restrict__ a, const Fr* __restrict__ b, const u64 n)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) gst(&out[i], mul(gld(&a[i]), gld(&b[i])));
}
// out[i] = sum_t coef[t] * pol_t[i] (pol_t read as 0 past its length): Polynomial::add / sub / mulScalar / subScalar chains in one pass
struct LinTerms { const Fr* p[6]; u64 len[6]; Fr k[6]; u32 n; };
__global__ void lincomb_kernel(Fr* __restrict__ out, const u64 len, const LinTerms T)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    Fr acc = zero();
    for (u32 t = 0; t < T.n; t++)
        if (i < T.len[t]) acc = add(acc, mul(gld(&T.p[t][i]), T.k[t]));
    gst(&out[i], acc);
}
// CPolynomial::getPolynomial (cpolynomial.c.hpp:39-63): out[i * k + j] = pol_j[i]
struct Interleave { const Fr* p[8]; u64 len[8]; u32 k; };
__global__ void interleave_kernel(Fr* __restrict__ out, const u64 out_len, const Interleave I)
{
    const u64 o = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= out_len) return;
    const u64 i = o / I.k;
    const u32 j = (u32)(o % I.k);
    gst(&out[o], i < I.len[j] ? gld(&I.p[j][i]) : zero());
}
// Polynomial::divZh (polynomial.c.hpp:537-569): c / (X^n - 1) on the coefficients, chunk after chunk of n
__global__ void div_zh_kernel(Fr* __restrict__ c, const u64 n, const u64 chunks)
{
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    Fr prev = neg(gld(&c[j]));
    gst(&c[j], prev);
    for (u64 i = 1; i < chunks; i++) {
        prev = sub(prev, gld(&c[i * n + j]));
        gst(&c[i * n + j], prev);
    }
}

// ------------------------------------------------------------------------------------------------ batch inverse
// out[i] = 1 / in[i] (Montgomery's trick, FflonkProver::batchInverse :1531-1555, per lane over the elements i = t, t + T, t + 2T, ...;
// one exponentiation per lane).  out != in.  A zero poisons the elements of its lane (the reference divides by zero there too).
__global__ void batch_inverse_kernel(Fr* __restrict__ out, const Fr* __restrict__ in, const u64 n, const u64 T)
{
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T || t >= n) return;
    Fr acc = one();
    for (u64 i = t; i < n; i += T) { acc = mul(acc, gld(&in[i])); gst(&out[i], acc); }     // prefix products
    Fr inv = inverse(acc);
    u64 last = t + ((n - 1 - t) / T) * T;
    for (u64 i = last; i > t; i -= T) {
        const Fr x = gld(&in[i]);
        gst(&out[i], mul(inv, gld(&out[i - T])));
        inv = mul(inv, x);
    }
    gst(&out[t], inv);
}

// ------------------------------------------------------------------------------------------------ recurrences of lag m as scans
// View: element j of a sequence lives at base[start + j * step] (step may be negative), j < count.
//   LINREC:  s_j = beta * s_(j-m) + x_j         (division by X^m - beta from the top: m interleaved Horner chains; m = 1: Horner)
//   PROD:    s_j = s_(j-1) * x_j                (running products; m = 1)
// A lane owns (run r, class c): the SCAN_K elements j = (r * SCAN_K + t) * m + c -- the m lanes of a run read neighbouring elements, so one pass
// over the array serves all m chains (until round 4 every class was a scan of its own over the same cache lines).
constexpr u64 SCAN_K = 128;
template <bool PROD>
__global__ void scan_local_kernel(Fr* __restrict__ out, const long long ostart, const long long ostep, const Fr* __restrict__ in, const long long istart, const long long istep,
                                  const u64 count, const u64 m, const u64 lanes, Fr* __restrict__ totals, const Fr beta)
{
    const u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= lanes) return;
    const u64 r = q / m, c = q % m;
    Fr s = PROD ? one() : zero();
    for (u64 t = 0; t < SCAN_K; t++) {
        const u64 j = (r * SCAN_K + t) * m + c;
        if (j >= count) break;
        const Fr x = gld(&in[istart + (long long)j * istep]);
        s = PROD ? mul(s, x) : add(mul(s, beta), x);
        gst(&out[ostart + (long long)j * ostep], s);
    }
    gst(&totals[q], s);
}
// lane (r, c), r >= 1: the true value at the end of (r - 1, c) enters its elements
template <bool PROD>
__global__ void scan_fix_kernel(Fr* __restrict__ out, const long long ostart, const long long ostep, const u64 count, const u64 m, const u64 lanes, const Fr* __restrict__ scanned, const Fr beta)
{
    const u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= lanes || q < m) return;
    const u64 r = q / m, c = q % m;
    Fr carry = gld(&scanned[q - m]);
    for (u64 t = 0; t < SCAN_K; t++) {
        const u64 j = (r * SCAN_K + t) * m + c;
        if (j >= count) break;
        Fr* o = &out[ostart + (long long)j * ostep];
        if (PROD) gst(o, mul(gld(o), carry));
        else { carry = mul(carry, beta); gst(o, add(gld(o), carry)); }
    }
}

// ------------------------------------------------------------------------------------------------ evaluation
// one level of Polynomial::fastEvaluate (polynomial.c.hpp:184-214: Horner over runs, the runs combined with powers of the point):
// out[q * runs + t] = sum_(j < K) in_q[t * K + j] * x_q^j, q < npts; in_q = in + q * in_stride (0: the same polynomial for every point)
constexpr u64 EVAL_K = 64;
struct EvalPts { Fr x[8]; u32 n; };
__global__ void eval_pass_kernel(Fr* __restrict__ out, const Fr* __restrict__ in, const u64 in_stride, const u64 len, const u64 runs, const EvalPts P)
{
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= runs) return;
    const u64 j0 = t * EVAL_K, j1 = j0 + EVAL_K < len ? j0 + EVAL_K : len;
    for (u32 q = 0; q < P.n; q++) {
        const Fr* src = in + (u64)q * in_stride;
        Fr acc = zero();
        for (u64 j = j1; j > j0; j--) acc = add(mul(acc, P.x[q]), gld(&src[j - 1]));
        gst(&out[(u64)q * runs + t], acc);
    }
}

// ------------------------------------------------------------------------------------------------ the rounds' loops
// FflonkProver::computeT0 (:727-788) on the 4n evaluations; lagrange: npub blocks of 4n evaluations, pub[j] = buffers["A"][j]
__global__ void t0_kernel(Fr* __restrict__ out, const Fr* __restrict__ a, const Fr* __restrict__ b, const Fr* __restrict__ c, const Fr* __restrict__ ql, const Fr* __restrict__ qr,
                          const Fr* __restrict__ qm, const Fr* __restrict__ qo, const Fr* __restrict__ qc, const Fr* __restrict__ lagrange, const Fr* __restrict__ pub, const u32 npub, const u64 n4)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const Fr av = gld(&a[i]), bv = gld(&b[i]), cv = gld(&c[i]);
    Fr pi = zero();
    for (u32 j = 0; j < npub; j++) pi = sub(pi, mul(gld(&lagrange[(u64)j * n4 + i]), gld(&pub[j])));
    Fr t = mul(av, gld(&ql[i]));
    t = add(t, mul(bv, gld(&qr[i])));
    t = add(t, mul(mul(av, bv), gld(&qm[i])));
    t = add(t, mul(cv, gld(&qo[i])));
    t = add(t, add(gld(&qc[i]), pi));
    gst(&out[i], t);
}
// computeZ (:871-913): num / den factors of row i; sigma evaluations are read at 4 i (the 4n-point evaluations); wpow: omega_n^(2^k)
struct ZArgs { Fr beta, gamma, k1, k2; };
__global__ void z_numden_kernel(Fr* __restrict__ num, Fr* __restrict__ den, const Fr* __restrict__ a, const Fr* __restrict__ b, const Fr* __restrict__ c, const Fr* __restrict__ s1,
                                const Fr* __restrict__ s2, const Fr* __restrict__ s3, const Fr* __restrict__ wpow, const u64 n, const ZArgs A)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fr av = gld(&a[i]), bv = gld(&b[i]), cv = gld(&c[i]);
    const Fr bw = mul(A.beta, pow_from_table(i, wpow));
    Fr n1 = add(add(av, bw), A.gamma), n2 = add(add(bv, mul(A.k1, bw)), A.gamma), n3 = add(add(cv, mul(A.k2, bw)), A.gamma);
    gst(&num[i], mul(n1, mul(n2, n3)));
    Fr d1 = add(add(av, mul(A.beta, gld(&s1[4 * i]))), A.gamma), d2 = add(add(bv, mul(A.beta, gld(&s2[4 * i]))), A.gamma), d3 = add(add(cv, mul(A.beta, gld(&s3[4 * i]))), A.gamma);
    gst(&den[i], mul(d1, mul(d2, d3)));
}
// Z[(i + 1) % n] = running product i (computeZ :915-946: numArr / denArr shifted by one, the whole product at 0)
__global__ void z_rotate_kernel(Fr* __restrict__ z, const Fr* __restrict__ run, const u64 n)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) gst(&z[(i + 1) % n], gld(&run[i]));
}
// computeT1 (:982-1011) on 2n points: z / lagrange_1 evaluations are read at 2 i; b7..b9 the blinding of Z; wpow: omega_2n^(2^k)
struct BArgs { Fr b7, b8, b9; };
__global__ void t1_kernel(Fr* __restrict__ t1, Fr* __restrict__ t1z, const Fr* __restrict__ z, const Fr* __restrict__ lagrange1, const Fr* __restrict__ wpow, const u64 n2, const BArgs B)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    const Fr om = pow_from_table(i, wpow), om2 = sqr(om);
    const Fr zp = add(add(mul(B.b7, om2), mul(B.b8, om)), B.b9);
    const Fr l1 = gld(&lagrange1[2 * i]);
    gst(&t1[i], mul(sub(gld(&z[2 * i]), one()), l1));
    gst(&t1z[i], mul(zp, l1));
}
// computeT2 (:1037-1113) on the 4n evaluations; wpow: omega_4n^(2^k), wn: omega_n
struct T2Args { Fr beta, gamma, k1, k2, wn, b7, b8, b9; };
__global__ void t2_kernel(Fr* __restrict__ t2, Fr* __restrict__ t2z, const Fr* __restrict__ a, const Fr* __restrict__ b, const Fr* __restrict__ c, const Fr* __restrict__ z,
                          const Fr* __restrict__ s1, const Fr* __restrict__ s2, const Fr* __restrict__ s3, const Fr* __restrict__ wpow, const u64 n4, const T2Args A)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const Fr om = pow_from_table(i, wpow), om2 = sqr(om), omw = mul(om, A.wn), omw2 = sqr(omw);
    const Fr av = gld(&a[i]), bv = gld(&b[i]), cv = gld(&c[i]), zv = gld(&z[i]), zw = gld(&z[(i + 4) % n4]);
    const Fr zp = add(add(mul(A.b7, om2), mul(A.b8, om)), A.b9), zwp = add(add(mul(A.b7, omw2), mul(A.b8, omw)), A.b9);
    const Fr bx = mul(A.beta, om);
    const Fr e1 = mul(mul(add(add(av, bx), A.gamma), add(add(bv, mul(bx, A.k1)), A.gamma)), add(add(cv, mul(bx, A.k2)), A.gamma));
    const Fr e2 = mul(mul(add(add(av, mul(A.beta, gld(&s1[i]))), A.gamma), add(add(bv, mul(A.beta, gld(&s2[i]))), A.gamma)), add(add(cv, mul(A.beta, gld(&s3[i]))), A.gamma));
    gst(&t2[i], sub(mul(e1, zv), mul(e2, zw)));
    gst(&t2z[i], sub(mul(e1, zp), mul(e2, zwp)));
}

static Fr load_host(const u64* w)
{
    Fr f;
    for (int i = 0; i < 4; i++) { f.v[2 * i] = (u32)w[i]; f.v[2 * i + 1] = (u32)(w[i] >> 32); }
    return f;
}

}  // namespace ffr

using ffr::Fr;

int dev_fr_gather_mont(Ctx* ctx, u64* out, const u32* map, u64 nmap, const u64* values, u64 nvalues)
{
    if (!nmap) return 0;
    hipLaunchKernelGGL(ffr::gather_mont_kernel, dim3(ffr::blocks(nmap)), dim3(ffr::TPB), 0, ctx->stream, (Fr*)out, map, nmap, (const Fr*)values, nvalues);
    XLP_HIP(hipGetLastError());
    return 0;
}
int dev_fr_from_mont(Ctx* ctx, u64* out, const u64* in, u64 n)
{
    if (!n) return 0;
    hipLaunchKernelGGL(ffr::from_mont_kernel, dim3(ffr::blocks(n)), dim3(ffr::TPB), 0, ctx->stream, (Fr*)out, (const Fr*)in, n);
    XLP_HIP(hipGetLastError());
    return 0;
}
int dev_fr_mul(Ctx* ctx, u64* out, const u64* a, const u64* b, u64 n)
{
    if (!n) return 0;
    hipLaunchKernelGGL(ffr::mul_kernel, dim3(ffr::blocks(n)), dim3(ffr::TPB), 0, ctx->stream, (Fr*)out, (const Fr*)a, (const Fr*)b, n);
    XLP_HIP(hipGetLastError());
    return 0;
}
int dev_fr_lincomb(Ctx* ctx, u64* out, u64 len, u32 nterms, const u64* const* pols, const u64* lens, const u64* coefs)
{
    if (!len) return 0;
    if (nterms > 6) return xlp_fail("fr_lincomb: at most 6 terms");
    ffr::LinTerms T;
    T.n = nterms;
    for (u32 t = 0; t < nterms; t++) { T.p[t] = (const Fr*)pols[t]; T.len[t] = lens[t]; T.k[t] = ffr::load_host(coefs + 4 * t); }
    hipLaunchKernelGGL(ffr::lincomb_kernel, dim3(ffr::blocks(len)), dim3(ffr::TPB), 0, ctx->stream, (Fr*)out, len, T);
    XLP_HIP(hipGetLastError());
    return 0;
}
int dev_fr_interleave(Ctx* ctx, u64* out, u64 out_len, u32 k, const u64* const* pols, const u64* lens)
{
    if (!out_len) return 0;
    if (k < 1 || k > 8) return xlp_fail("fr_interleave: 1..8 polynomials");
    ffr::Interleave I;
    I.k = k;
    for (u32 j = 0; j < k; j++) { I.p[j] = (const Fr*)pols[j]; I.len[j] = lens[j]; }
    hipLaunchKernelGGL(ffr::interleave_kernel, dim3(ffr::blocks(out_len)), dim3(ffr::TPB), 0, ctx->stream, (Fr*)out, out_len, I);
    XLP_HIP(hipGetLastError());
    return 0;
}
int dev_fr_div_zh(Ctx* ctx, u64* pol, u64 len, u64 n)
{
    if (!n || len % n) return xlp_fail("fr_div_zh: the length must be a multiple of the domain size");
    hipLaunchKernelGGL(ffr::div_zh_kernel, dim3(ffr::blocks(n)), dim3(ffr::TPB), 0, ctx->stream, (Fr*)pol, n, len / n);
    XLP_HIP(hipGetLastError());
    return 0;
}
int dev_fr_batch_inverse(Ctx* ctx, u64* out, const u64* in, u64 n)
{
    if (!n) return 0;
    if (out == in) return xlp_fail("fr_batch_inverse: out of place only");
    u64 T = (u64)ctx->ncu * 1024;                       // lanes: each inverts n / T elements with one exponentiation
    if (T > (n + 15) / 16) T = (n + 15) / 16;
    hipLaunchKernelGGL(ffr::batch_inverse_kernel, dim3(ffr::blocks(T)), dim3(ffr::TPB), 0, ctx->stream, (Fr*)out, (const Fr*)in, n, T);
    XLP_HIP(hipGetLastError());
    return 0;
}

void fr_host_sqr(u64* a);
// the recurrence over a view; the lanes' totals obey the same recurrence (lag m, beta^K) and are scanned recursively in scratch (slot 4)
template <bool PROD>
static int scan_view(Ctx* ctx, Fr* out, long long ostart, long long ostep, const Fr* in, long long istart, long long istep, u64 count, u64 m, const u64* beta_host, Fr* tot, u64 tot_cap)
{
    if (!count) return 0;
    const u64 runs = (count + ffr::SCAN_K * m - 1) / (ffr::SCAN_K * m), lanes = runs * m;
    if (lanes > tot_cap) return xlp_fail("fr scan: scratch too small");
    const Fr beta = ffr::load_host(beta_host);
    hipLaunchKernelGGL(ffr::scan_local_kernel<PROD>, dim3(ffr::blocks(lanes)), dim3(ffr::TPB), 0, ctx->stream, out, ostart, ostep, in, istart, istep, count, m, lanes, tot, beta);
    XLP_HIP(hipGetLastError());
    if (runs > 1) {
        u64 bk[4] = {beta_host[0], beta_host[1], beta_host[2], beta_host[3]};
        if (!PROD) for (u64 k = ffr::SCAN_K; k > 1; k >>= 1) fr_host_sqr(bk);      // beta^K (K a power of two)
        if (scan_view<PROD>(ctx, tot, 0, 1, tot, 0, 1, lanes, m, bk, tot + lanes, tot_cap - lanes)) return -1;
        hipLaunchKernelGGL(ffr::scan_fix_kernel<PROD>, dim3(ffr::blocks(lanes)), dim3(ffr::TPB), 0, ctx->stream, out, ostart, ostep, count, m, lanes, tot, beta);
        XLP_HIP(hipGetLastError());
    }
    return 0;
}

// host Montgomery square (launch constants only)
void fr_host_sqr(u64* a)
{
    typedef unsigned __int128 u128;
    static const u64 Q[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    static const u64 NP = 0xc2e1f593efffffffULL;
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { const u128 p = (u128)a[j] * a[i] + t[j] + (u64)c; t[j] = (u64)p; c = p >> 64; }
        u128 s = (u128)t[4] + (u64)c; t[4] = (u64)s; t[5] = (u64)(s >> 64);
        const u64 m = t[0] * NP;
        u128 p = (u128)m * Q[0] + t[0]; c = p >> 64;
        for (int j = 1; j < 4; j++) { p = (u128)m * Q[j] + t[j] + (u64)c; t[j - 1] = (u64)p; c = p >> 64; }
        s = (u128)t[4] + (u64)c; t[3] = (u64)s; t[4] = t[5] + (u64)(s >> 64);
    }
    bool ge = t[4] != 0;
    if (!ge) { ge = true; for (int i = 3; i >= 0; i--) if (t[i] != Q[i]) { ge = t[i] > Q[i]; break; } }
    if (ge) { u128 bw = 0; for (int i = 0; i < 4; i++) { const u128 d = (u128)t[i] - Q[i] - (u64)bw; t[i] = (u64)d; bw = (d >> 64) & 1; } }
    for (int i = 0; i < 4; i++) a[i] = t[i];
}

static int scan_scratch(Ctx* ctx, u64 count, u64 m, Fr** tot, u64* cap)
{
    u64 need = 0;
    for (u64 c = count; c > m;) { c = ((c + ffr::SCAN_K * m - 1) / (ffr::SCAN_K * m)) * m; need += c; }
    need += 8 + 2 * m;
    u64* p = nullptr;
    if (ctx_scratch(ctx, need * sizeof(Fr), 4, &p)) return -1;
    *tot = (Fr*)p; *cap = need;
    return 0;
}

// out[i] = prod_(k <= i) in[k]
int dev_fr_prefix_product(Ctx* ctx, u64* out, const u64* in, u64 n)
{
    if (!n) return 0;
    Fr* tot; u64 cap;
    if (scan_scratch(ctx, n, 1, &tot, &cap)) return -1;
    const u64 unused[4] = {0, 0, 0, 0};
    return scan_view<true>(ctx, (Fr*)out, 0, 1, (const Fr*)in, 0, 1, n, 1, unused, tot, cap);
}
// Polynomial::divByZerofier (polynomial.c.hpp:571-634): out = in / (X^m - beta) for a polynomial that IS divisible (len coefficients in,
// len out, the top m zero).  From the top: q[k] = in[k + m] + beta q[k + m]; with j counting down from the top coefficient this is
// s_j = beta s_(j-m) + x_j, x_j = in[len - 1 - j], q[len - m - 1 - j] = s_j.
int dev_fr_div_xm_minus(Ctx* ctx, u64* out, const u64* in, u64 len, u32 m, const u64* beta)
{
    if (!len) return 0;
    if (out == in) return xlp_fail("fr_div_xm_minus: out of place only");
    if (m < 1 || m > 64) return xlp_fail("fr_div_xm_minus: 1 <= m <= 64");
    if (len <= m) { XLP_HIP(hipMemsetAsync(out, 0, len * sizeof(Fr), ctx->stream)); return 0; }
    XLP_HIP(hipMemsetAsync((Fr*)out + (len - m), 0, (u64)m * sizeof(Fr), ctx->stream));
    Fr* tot; u64 cap;
    if (scan_scratch(ctx, len - m, m, &tot, &cap)) return -1;
    return scan_view<false>(ctx, (Fr*)out, (long long)(len - m - 1), -1, (const Fr*)in, (long long)(len - 1), -1, len - m, m, beta, tot, cap);
}
// values[q] = pol(points[q]), q < npoints <= 8 (HOST arrays; one pass over the coefficients per level)
int dev_fr_eval(Ctx* ctx, u64* values, const u64* pol, u64 len, const u64* points, u32 npoints)
{
    if (npoints < 1 || npoints > 8) return xlp_fail("fr_eval: 1..8 points");
    if (!len) { for (u32 q = 0; q < 4 * npoints; q++) values[q] = 0; return 0; }
    u64 need = 0;
    for (u64 c = len; c > 1;) { c = (c + ffr::EVAL_K - 1) / ffr::EVAL_K; need += c * npoints; }
    need += 8 * npoints;
    u64* sp = nullptr;
    if (ctx_scratch(ctx, need * sizeof(Fr), 4, &sp)) return -1;
    Fr* buf = (Fr*)sp;
    ffr::EvalPts P;
    P.n = npoints;
    u64 x[8][4];
    for (u32 q = 0; q < npoints; q++) for (int i = 0; i < 4; i++) x[q][i] = points[4 * q + i];
    const Fr* in = (const Fr*)pol;
    u64 stride = 0, cur = len;
    for (;;) {
        const u64 runs = (cur + ffr::EVAL_K - 1) / ffr::EVAL_K;
        for (u32 q = 0; q < npoints; q++) P.x[q] = ffr::load_host(x[q]);
        hipLaunchKernelGGL(ffr::eval_pass_kernel, dim3(ffr::blocks(runs)), dim3(ffr::TPB), 0, ctx->stream, buf, in, stride, cur, runs, P);
        XLP_HIP(hipGetLastError());
        if (runs == 1) break;
        for (u32 q = 0; q < npoints; q++) for (u64 k = ffr::EVAL_K; k > 1; k >>= 1) fr_host_sqr(x[q]);      // the runs combine at x^K
        in = buf; stride = runs; cur = runs;
        buf += runs * npoints;
    }
    XLP_HIP(hipMemcpyAsync(values, buf, (u64)npoints * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));     // stride = 1 run per point
    XLP_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

static int upload_wpow(Ctx* ctx, const u64* wpow_host, u32 count, Fr** d)
{
    if (count > 64) return xlp_fail("fflonk: at most 64 powers of the root");
    u64* p = nullptr;
    if (ctx_scratch(ctx, 64 * sizeof(Fr), 5, &p)) return -1;
    XLP_HIP(hipMemcpyAsync(p, wpow_host, (u64)count * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    XLP_HIP(hipStreamSynchronize(ctx->stream));
    *d = (Fr*)p;
    return 0;
}

int dev_fflonk_t0(Ctx* ctx, u64* out, const u64* a, const u64* b, const u64* c, const u64* ql, const u64* qr, const u64* qm, const u64* qo, const u64* qc, const u64* lagrange,
                  const u64* pub, u32 npub, u64 n4)
{
    if (!n4) return 0;
    hipLaunchKernelGGL(ffr::t0_kernel, dim3(ffr::blocks(n4)), dim3(ffr::TPB), 0, ctx->stream, (Fr*)out, (const Fr*)a, (const Fr*)b, (const Fr*)c, (const Fr*)ql, (const Fr*)qr, (const Fr*)qm,
                       (const Fr*)qo, (const Fr*)qc, (const Fr*)lagrange, (const Fr*)pub, npub, n4);
    XLP_HIP(hipGetLastError());
    return 0;
}
// scalars (HOST, Montgomery): beta, gamma, k1, k2; wpow (HOST): omega_n^(2^k), k < log2 n
int dev_fflonk_z_numden(Ctx* ctx, u64* num, u64* den, const u64* a, const u64* b, const u64* c, const u64* s1, const u64* s2, const u64* s3, u64 n, const u64* scalars, const u64* wpow, u32 nw)
{
    if (!n) return 0;
    Fr* dw;
    if (upload_wpow(ctx, wpow, nw, &dw)) return -1;
    ffr::ZArgs A = {ffr::load_host(scalars), ffr::load_host(scalars + 4), ffr::load_host(scalars + 8), ffr::load_host(scalars + 12)};
    hipLaunchKernelGGL(ffr::z_numden_kernel, dim3(ffr::blocks(n)), dim3(ffr::TPB), 0, ctx->stream, (Fr*)num, (Fr*)den, (const Fr*)a, (const Fr*)b, (const Fr*)c, (const Fr*)s1, (const Fr*)s2,
                       (const Fr*)s3, dw, n, A);
    XLP_HIP(hipGetLastError());
    return 0;
}
int dev_fflonk_z_rotate(Ctx* ctx, u64* z, const u64* run, u64 n)
{
    if (!n) return 0;
    hipLaunchKernelGGL(ffr::z_rotate_kernel, dim3(ffr::blocks(n)), dim3(ffr::TPB), 0, ctx->stream, (Fr*)z, (const Fr*)run, n);
    XLP_HIP(hipGetLastError());
    return 0;
}
// blinding (HOST): b7, b8, b9; wpow: omega_2n^(2^k)
int dev_fflonk_t1(Ctx* ctx, u64* t1, u64* t1z, const u64* z, const u64* lagrange1, u64 n2, const u64* blinding, const u64* wpow, u32 nw)
{
    if (!n2) return 0;
    Fr* dw;
    if (upload_wpow(ctx, wpow, nw, &dw)) return -1;
    ffr::BArgs B = {ffr::load_host(blinding), ffr::load_host(blinding + 4), ffr::load_host(blinding + 8)};
    hipLaunchKernelGGL(ffr::t1_kernel, dim3(ffr::blocks(n2)), dim3(ffr::TPB), 0, ctx->stream, (Fr*)t1, (Fr*)t1z, (const Fr*)z, (const Fr*)lagrange1, dw, n2, B);
    XLP_HIP(hipGetLastError());
    return 0;
}
// scalars (HOST): beta, gamma, k1, k2, omega_n, b7, b8, b9; wpow: omega_4n^(2^k)
int dev_fflonk_t2(Ctx* ctx, u64* t2, u64* t2z, const u64* a, const u64* b, const u64* c, const u64* z, const u64* s1, const u64* s2, const u64* s3, u64 n4, const u64* scalars, const u64* wpow, u32 nw)
{
    if (!n4) return 0;
    Fr* dw;
    if (upload_wpow(ctx, wpow, nw, &dw)) return -1;
    ffr::T2Args A;
    Fr* f = &A.beta;
    for (int i = 0; i < 8; i++) f[i] = ffr::load_host(scalars + 4 * i);
    hipLaunchKernelGGL(ffr::t2_kernel, dim3(ffr::blocks(n4)), dim3(ffr::TPB), 0, ctx->stream, (Fr*)t2, (Fr*)t2z, (const Fr*)a, (const Fr*)b, (const Fr*)c, (const Fr*)z, (const Fr*)s1,
                       (const Fr*)s2, (const Fr*)s3, dw, n4, A);
    XLP_HIP(hipGetLastError());
    return 0;
}

}  // namespace xlp
