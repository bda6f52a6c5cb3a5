// fflonk_bench_key.hpp -- a SYNTHETIC fflonk proving key and witness of any size, built in memory in the byte format FflonkProver::setZkey reads
// (src/rapidsnark/zkey_fflonk.cpp:18-49, fflonk_prover.c.hpp:94-330), for timing the prover at sizes a Python setup cannot reach (the
// reference's final.zkey is not in the snapshot).  The circuit: row 0 takes the public input, every other row i < n - 2 is a gate
// ql a + qr b + qm a b + qo c + qc = 0 over three signals of its own with random selectors (qc chosen so that it holds), the permutation is the
// identity.  The "powers of tau" are the points G, 2G, 3G, ...: valid curve points, NOT a structured reference string -- the proof has the cost
// of a real one and verifies against nothing.  Correctness of the prover is tests/test_fflonk.py's business (keys with a known tau).
#pragma once
#include "fflonk_prover.hpp"

namespace FflonkBench {

struct Rng { uint64_t s; uint64_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; } FrElement fr() { FrElement r = {{next(), next(), next(), next() >> 3}}; return FrH::toMontgomery(r); } };

// P_i = (i + 1) G, i < count, as affine Montgomery coordinates: `lanes` chains advance by lanes * G per step, the slopes' denominators inverted together
inline void multiplesOfG(G1PointAffine* out, uint64_t count)
{
    typedef FqH F;
    const F::Element gx = F::set(1), gy = F::set(2);
    auto addAffine = [](const G1PointAffine& p, const G1PointAffine& q) {          // p != +-q
        const F::Element l = F::mul(F::sub(q.y, p.y), F::inv(F::sub(q.x, p.x)));
        G1PointAffine r;
        r.x = F::sub(F::sub(F::square(l), p.x), q.x);
        r.y = F::sub(F::mul(l, F::sub(p.x, r.x)), p.y);
        return r;
    };
    auto dbl = [](const G1PointAffine& p) {
        const F::Element three = F::set(3), two = F::set(2);
        const F::Element l = F::mul(F::mul(three, F::square(p.x)), F::inv(F::mul(two, p.y)));
        G1PointAffine r;
        r.x = F::sub(F::square(l), F::add(p.x, p.x));
        r.y = F::sub(F::mul(l, F::sub(p.x, r.x)), p.y);
        return r;
    };
    const uint64_t lanes = std::min<uint64_t>(4096, count);
    out[0] = {gx, gy};
    if (count > 1) out[1] = dbl(out[0]);
    for (uint64_t i = 2; i < lanes; i++) out[i] = addAffine(out[i - 1], out[0]);
    if (count <= lanes) return;
    const G1PointAffine D = out[lanes - 1];                                     // lanes * G
    std::vector<F::Element> den(lanes), pre(lanes);
    for (uint64_t base = lanes; base < count; base += lanes) {
        const uint64_t k = std::min(lanes, count - base);
        F::Element acc = F::one();
        for (uint64_t j = 0; j < k; j++) { den[j] = F::sub(D.x, out[base - lanes + j].x); pre[j] = acc; acc = F::mul(acc, den[j]); }
        F::Element inv = F::inv(acc);
        for (uint64_t j = k; j-- > 0;) {
            const F::Element di = F::mul(inv, pre[j]);
            inv = F::mul(inv, den[j]);
            const G1PointAffine& p = out[base - lanes + j];
            const F::Element l = F::mul(F::sub(D.y, p.y), di);
            G1PointAffine r;
            r.x = F::sub(F::sub(F::square(l), p.x), D.x);
            r.y = F::sub(F::mul(l, F::sub(p.x, r.x)), p.y);
            out[base + j] = r;
        }
    }
}

// -> the zkey image; witness (nVars values, plain little-endian integers as in a .wtns)
inline std::vector<char> build(uint32_t power, std::vector<FrElement>& witness, uint64_t seed = 0x5EED0F10)
{
    const uint64_t n = 1ULL << power, nConstraints = n - 2, nVars = 3 * nConstraints + 2;
    Rng rng{seed};
    std::vector<FrElement> val(nVars);                                          // Montgomery values of the signals
    val[0] = FrH::zero();
    for (uint64_t i = 1; i < nVars; i++) val[i] = rng.fr();
    std::vector<uint32_t> A(nConstraints), B(nConstraints), C(nConstraints);
    std::vector<std::vector<FrElement>> q(5, std::vector<FrElement>(n, FrH::zero()));     // ql qr qm qo qc evaluations
    A[0] = 1; B[0] = 0; C[0] = 0; q[0][0] = FrH::one();
    for (uint64_t i = 1; i < nConstraints; i++) {
        A[i] = (uint32_t)(3 * i - 1); B[i] = (uint32_t)(3 * i); C[i] = (uint32_t)(3 * i + 1);
        const FrElement a = val[A[i]], b = val[B[i]], c = val[C[i]];
        for (int k = 0; k < 4; k++) q[k][i] = rng.fr();
        FrElement t = FrH::add(FrH::mul(q[0][i], a), FrH::mul(q[1][i], b));
        t = FrH::add(t, FrH::mul(q[2][i], FrH::mul(a, b)));
        t = FrH::add(t, FrH::mul(q[3][i], c));
        q[4][i] = FrH::neg(t);
    }
    witness.resize(nVars);
    for (uint64_t i = 0; i < nVars; i++) witness[i] = FrH::fromMontgomery(val[i]);
    witness[0] = FrH::fromMontgomery(FrH::one());
    auto rootOf = [](uint32_t bits) {
        uint64_t e[4] = {FrModulus::Q[0] - 1, FrModulus::Q[1], FrModulus::Q[2], FrModulus::Q[3]};
        for (int i = 0; i < 4; i++) e[i] = (e[i] >> bits) | (i < 3 && bits ? e[i + 1] << (64 - bits) : 0);
        return FrH::pow(FrH::set(5), e);
    };
    const FrElement k1 = FrH::set(2), k2 = FrH::set(3);
    // sections: coefficients (n) then the 4n evaluations; the transforms through the library (host-pointer entry point)
    auto coefAndEval = [&](std::vector<FrElement>& ev) {                        // ev: n evaluations in, 5n (coefficients | 4n evaluations) out
        XLP_CHECK(xlp_fft_bn254(ev[0].v, n, 1));
        ev.resize(5 * n, FrH::zero());
        memcpy(ev[n].v, ev[0].v, n * 32);
        XLP_CHECK(xlp_fft_bn254(ev[n].v, 4 * n, 0));
    };
    std::vector<std::vector<FrElement>> sec(9);                                 // ql qr qm qo qc s1 s2 s3 lagrange1
    for (int k = 0; k < 5; k++) { sec[k] = std::move(q[k]); coefAndEval(sec[k]); }
    for (int k = 0; k < 3; k++) {                                              // the identity permutation: S_j(X) = k_j X
        sec[5 + k].assign(5 * n, FrH::zero());
        sec[5 + k][1] = k == 0 ? FrH::one() : k == 1 ? k1 : k2;
        memcpy(sec[5 + k][n].v, sec[5 + k][0].v, n * 32);
        XLP_CHECK(xlp_fft_bn254(sec[5 + k][n].v, 4 * n, 0));
    }
    sec[8].assign(n, FrH::zero());
    sec[8][0] = FrH::one();
    coefAndEval(sec[8]);
    std::vector<FrElement> c0(8 * n, FrH::zero());
    const int order[8] = {0, 1, 3, 2, 4, 5, 6, 7};                              // QL QR QO QM QC S1 S2 S3
    for (int j = 0; j < 8; j++) for (uint64_t i = 0; i < n; i++) c0[8 * i + j] = sec[order[j]][i];
    std::vector<G1PointAffine> ptau(9 * n);
    multiplesOfG(ptau.data(), 9 * n);
    // ---- the image
    std::vector<char> img;
    auto put = [&](const void* p, size_t bytes) { const char* c = (const char*)p; img.insert(img.end(), c, c + bytes); };
    auto u32 = [&](uint32_t v) { put(&v, 4); };
    auto u64 = [&](uint64_t v) { put(&v, 8); };
    auto sectionHeader = [&](uint32_t id, uint64_t size) { u32(id); u64(size); };
    img.reserve((size_t)(9 * n * 64 + 54 * n * 32 + 12 * n + 4096));
    put("zkey", 4); u32(1); u32(17);
    sectionHeader(1, 4); u32(Zkey::FFLONK_PROTOCOL_ID);
    const FrElement w3 = [] { uint64_t e[4]; FrElement qm1 = {{FrModulus::Q[0] - 1, FrModulus::Q[1], FrModulus::Q[2], FrModulus::Q[3]}};      // 5^((r - 1) / 3)
                              unsigned __int128 rem = 0; for (int i = 3; i >= 0; i--) { const unsigned __int128 cur = (rem << 64) | qm1.v[i]; e[i] = (uint64_t)(cur / 3); rem = cur % 3; }
                              return FrH::pow(FrH::set(5), e); }();
    const uint64_t inv3 = [&] { uint64_t x = 1; while ((3 * x) % n != 1) x++; return x; }();      // wr = omega^(1/3)
    const uint64_t e3[4] = {inv3, 0, 0, 0};
    const FrElement wr = FrH::pow(rootOf(power), e3);
    sectionHeader(2, 4 + 32 + 4 + 32 + 20 + 6 * 32 + 128 + 64);
    u32(32); put(FqModulus::Q, 32); u32(32); put(FrModulus::Q, 32);
    u32((uint32_t)nVars); u32(1); u32((uint32_t)n); u32(0); u32((uint32_t)nConstraints);
    const FrElement hdr[6] = {k1, k2, w3, rootOf(2), rootOf(3), wr};
    put(hdr, sizeof hdr);
    { std::vector<char> z(128, 0); put(z.data(), 128); }
    put(&ptau[0], 64);                                                          // "C0" of the header: any point (the transcript hashes it)
    sectionHeader(3, 0);
    sectionHeader(4, nConstraints * 4); put(A.data(), nConstraints * 4);
    sectionHeader(5, nConstraints * 4); put(B.data(), nConstraints * 4);
    sectionHeader(6, nConstraints * 4); put(C.data(), nConstraints * 4);
    for (int k = 0; k < 8; k++) { sectionHeader(7 + k, 5 * n * 32); put(sec[k].data(), 5 * n * 32); std::vector<FrElement>().swap(sec[k]); }
    sectionHeader(15, 5 * n * 32); put(sec[8].data(), 5 * n * 32);
    sectionHeader(16, 9 * n * 64); put(ptau.data(), 9 * n * 64);
    sectionHeader(17, 8 * n * 32); put(c0.data(), 8 * n * 32);
    return img;
}

}  // namespace FflonkBench
