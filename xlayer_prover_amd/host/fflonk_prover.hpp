// fflonk_prover.hpp -- host-side C++ mirror of the reference's fflonk prover for the final proof, over the C ABI of
// libxlp_gl.so.  Same class / method names and argument meaning as the reference:
//
//   BinFileUtils::BinFile            src/rapidsnark/binfile_utils.{hpp,cpp}          the zkey / wtns container
//   Zkey::FflonkZkeyHeader           src/rapidsnark/zkey_fflonk.{hpp,cpp}
//   WtnsUtils::Header / loadHeader   src/rapidsnark/wtns_utils.{hpp,cpp}
//   Keccak256Transcript              src/rapidsnark/keccak_256_transcript.{hpp,c.hpp}
//   Fflonk::FflonkProver             src/rapidsnark/fflonk_prover.{hpp,c.hpp}        setZkey(), prove(): called at src/prover/prover.cpp:117-118,931
//
// Where the work happens: every vector of the prover -- the key's polynomials and evaluations, the powers of tau, the witness maps, the wire /
// quotient / combined polynomials -- lives in HBM; the loops of the five rounds are kernels (csrc/fflonk.hip: xlp_dev_fflonk_*, xlp_dev_fr_*),
// the transforms xlp_dev_fft_bn254, the five commitments xlp_dev_msm_bn254.  The host keeps the protocol: the transcript (Keccak-256), the 18
// roots, the three Lagrange interpolations on 8 / 4 / 6 points, scalars.  The reference's sequential pieces -- calculateAdditions (each addition
// may read earlier ones) stays a host loop as in the reference; divByZerofier / the running products of computeZ / fastEvaluate become scans.
//
// Differences from the reference, by design: prove() returns the two JSON documents as text (no nlohmann here); the degree checks read the
// coefficients that must vanish through one evaluation at a transcript-independent point instead of Polynomial::fixDegree scans on the host;
// setBlindingFactors() lets a test fix b_1 .. b_9 (the reference draws them with libsodium, fflonk_prover.c.hpp:645-650).
// PARITY UNPINNED against the reference (the snapshot has no zkey): checked against the CPU restatement of the prover -- same proof for the
// same blinding -- and against the fflonk verification equation (tests/test_fflonk.py).
#pragma once
#include <algorithm>
#include <array>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <vector>
#include "xlp_host.hpp"

namespace BinFileUtils {
// binfile_utils.cpp:15-56: "type" (4 bytes), u32 version, u32 sections; per section u32 id, u64 size, data
class BinFile
{
    std::vector<char> raw;
    std::map<uint32_t, std::pair<uint64_t, uint64_t>> sections;      // id -> (offset, size)
    const char* cur = NULL; const char* curEnd = NULL;
    bool pinned = false;
public:
    BinFile(const std::string& fileName, const std::string& type, uint32_t maxVersion)
    {
        FILE* f = fopen(fileName.c_str(), "rb");
        if (!f) xlp_exit_process(("BinFile: cannot open " + fileName).c_str());
        fseek(f, 0, SEEK_END); const long sz = ftell(f); fseek(f, 0, SEEK_SET);
        if (sz < 0) { fclose(f); xlp_exit_process(("BinFile: cannot size " + fileName).c_str()); }
        raw.resize((size_t)sz);
        if (sz && fread(raw.data(), 1, (size_t)sz, f) != (size_t)sz) { fclose(f); xlp_exit_process(("BinFile: short read on " + fileName).c_str()); }
        fclose(f);
        parse(type, maxVersion);
    }
    // the image of a file already in memory (binfile_utils.cpp:17-56); the vector is taken over
    BinFile(std::vector<char>&& image, const std::string& type, uint32_t maxVersion) : raw(std::move(image)) { parse(type, maxVersion); }
private:
    void parse(const std::string& type, uint32_t maxVersion)
    {
        if (raw.size() < 12 || std::string(raw.data(), 4) != type) xlp_exit_process(("Invalid file type. It should be " + type).c_str());
        uint32_t version, n;
        memcpy(&version, raw.data() + 4, 4); memcpy(&n, raw.data() + 8, 4);
        if (version > maxVersion) xlp_exit_process("BinFile: invalid version");
        uint64_t pos = 12;
        for (uint32_t i = 0; i < n; i++) {
            if (pos + 12 > raw.size()) xlp_exit_process("BinFile: truncated section table");
            uint32_t id; uint64_t size;
            memcpy(&id, raw.data() + pos, 4); memcpy(&size, raw.data() + pos + 4, 8);
            pos += 12;
            if (size > raw.size() - pos) xlp_exit_process("BinFile: section past the end of the file");
            if (!sections.count(id)) sections[id] = {pos, size};
            pos += size;
        }
    }
public:
    // page-lock the image (a prover in reserved-memory mode uploads the key's sections at every proof)
    void pin() { if (!pinned && raw.size()) pinned = xlp_host_register(raw.data(), raw.size()) == 0; }
    ~BinFile() { if (pinned) xlp_host_unregister(raw.data()); }
    BinFile(const BinFile&) = delete;
    BinFile& operator=(const BinFile&) = delete;
    void* getSectionData(uint32_t id) { if (!sections.count(id)) xlp_exit_process(("BinFile: section " + std::to_string(id) + " does not exist").c_str()); return raw.data() + sections[id].first; }
    uint64_t getSectionSize(uint32_t id) { if (!sections.count(id)) xlp_exit_process(("BinFile: section " + std::to_string(id) + " does not exist").c_str()); return sections[id].second; }
    void startReadSection(uint32_t id) { cur = (const char*)getSectionData(id); curEnd = cur + getSectionSize(id); }
    void endReadSection() { cur = curEnd = NULL; }
    void* read(uint64_t len) { if (!cur || len > (uint64_t)(curEnd - cur)) xlp_exit_process("BinFile: read past the section"); const char* p = cur; cur += len; return (void*)p; }
    uint32_t readU32LE() { uint32_t v; memcpy(&v, read(4), 4); return v; }
};
inline std::unique_ptr<BinFile> openExisting(const std::string& filename, const std::string& type, uint32_t maxVersion) { return std::unique_ptr<BinFile>(new BinFile(filename, type, maxVersion)); }
}  // namespace BinFileUtils

// 4 x 64-bit Montgomery arithmetic on the host (R = 2^256) for the handful of scalars the protocol needs; P: modulus, R2, -P^-1 mod 2^64
template <class P>
struct Mont256 {
    typedef unsigned __int128 u128;
    struct Element { uint64_t v[4]; };
    static Element mont(const Element& a, const Element& b)
    {
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; i++) {
            u128 c = 0;
            for (int j = 0; j < 4; j++) { const u128 p = (u128)a.v[j] * b.v[i] + t[j] + (uint64_t)c; t[j] = (uint64_t)p; c = p >> 64; }
            u128 s = (u128)t[4] + (uint64_t)c; t[4] = (uint64_t)s; t[5] = (uint64_t)(s >> 64);
            const uint64_t m = t[0] * P::NP;
            u128 p = (u128)m * P::Q[0] + t[0]; c = p >> 64;
            for (int j = 1; j < 4; j++) { p = (u128)m * P::Q[j] + t[j] + (uint64_t)c; t[j - 1] = (uint64_t)p; c = p >> 64; }
            s = (u128)t[4] + (uint64_t)c; t[3] = (uint64_t)s; t[4] = t[5] + (uint64_t)(s >> 64);
        }
        Element r = {{t[0], t[1], t[2], t[3]}};
        if (t[4] != 0 || geq(r)) subq(r);
        return r;
    }
    static bool geq(const Element& r) { for (int i = 3; i >= 0; i--) if (r.v[i] != P::Q[i]) return r.v[i] > P::Q[i]; return true; }
    static void subq(Element& r) { u128 bw = 0; for (int i = 0; i < 4; i++) { const u128 d = (u128)r.v[i] - P::Q[i] - (uint64_t)bw; r.v[i] = (uint64_t)d; bw = (d >> 64) & 1; } }
    static Element zero() { return {{0, 0, 0, 0}}; }
    static Element r2() { return {{P::R2[0], P::R2[1], P::R2[2], P::R2[3]}}; }
    static Element toMontgomery(const Element& a) { return mont(a, r2()); }
    static Element fromMontgomery(const Element& a) { Element o = {{1, 0, 0, 0}}; return mont(a, o); }
    static Element set(uint64_t x) { Element raw = {{x, 0, 0, 0}}; return toMontgomery(raw); }
    static Element one() { return set(1); }
    static Element mul(const Element& a, const Element& b) { return mont(a, b); }
    static Element square(const Element& a) { return mont(a, a); }
    static Element add(const Element& a, const Element& b)
    {
        Element r; u128 c = 0;
        for (int i = 0; i < 4; i++) { const u128 s = (u128)a.v[i] + b.v[i] + (uint64_t)c; r.v[i] = (uint64_t)s; c = s >> 64; }
        if (c || geq(r)) subq(r);
        return r;
    }
    static Element sub(const Element& a, const Element& b)
    {
        Element r; u128 bw = 0;
        for (int i = 0; i < 4; i++) { const u128 d = (u128)a.v[i] - b.v[i] - (uint64_t)bw; r.v[i] = (uint64_t)d; bw = (d >> 64) & 1; }
        if (bw) { u128 c = 0; for (int i = 0; i < 4; i++) { const u128 s = (u128)r.v[i] + P::Q[i] + (uint64_t)c; r.v[i] = (uint64_t)s; c = s >> 64; } }
        return r;
    }
    static Element neg(const Element& a) { return isZero(a) ? a : sub(zero(), a); }
    static bool isZero(const Element& a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; }
    static bool eq(const Element& a, const Element& b) { return memcmp(a.v, b.v, sizeof a.v) == 0; }
    static Element pow(Element base, const uint64_t e[4])
    {
        Element r = one();
        for (int i = 255; i >= 0; i--) { r = mont(r, r); if ((e[i >> 6] >> (i & 63)) & 1) r = mont(r, base); }
        return r;
    }
    static Element inv(const Element& a) { uint64_t e[4] = {P::Q[0] - 2, P::Q[1], P::Q[2], P::Q[3]}; return pow(a, e); }
    static Element div(const Element& a, const Element& b) { return mul(a, inv(b)); }
    // 32 big-endian bytes of the value (toRprBE, fr.cpp:297-312)
    static void toRprBE(const Element& a, uint8_t* out) { const Element x = fromMontgomery(a); for (int i = 0; i < 32; i++) out[i] = (uint8_t)(x.v[3 - i / 8] >> (8 * (7 - i % 8))); }
    // fromRprBE (fr.cpp:314-327, fromMpz :291-295): the 256-bit integer times R^2 through the Montgomery product, i.e. reduced mod the modulus
    static Element fromRprBE(const uint8_t* in) { Element x = zero(); for (int i = 0; i < 32; i++) x.v[3 - i / 8] |= (uint64_t)in[i] << (8 * (7 - i % 8)); return toMontgomery(x); }
    static std::string toString(const Element& a)
    {
        const Element x = fromMontgomery(a);
        uint64_t w[4] = {x.v[0], x.v[1], x.v[2], x.v[3]};
        std::string s;
        while (w[0] | w[1] | w[2] | w[3]) {
            u128 rem = 0;
            for (int i = 3; i >= 0; i--) { const u128 c = (rem << 64) | w[i]; w[i] = (uint64_t)(c / 10); rem = c % 10; }
            s.insert(s.begin(), (char)('0' + (int)rem));
        }
        return s.empty() ? "0" : s;
    }
};
struct FrModulus {
    static constexpr uint64_t Q[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    static constexpr uint64_t R2[4] = {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL};
    static constexpr uint64_t NP = 0xc2e1f593efffffffULL;
};
struct FqModulus {       // src/ffiasm/fq.asm (tests/golden/bn254_field_constants.json)
    static constexpr uint64_t Q[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    static constexpr uint64_t R2[4] = {0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL};
    static constexpr uint64_t NP = 0x87d20782e4866389ULL;
};
typedef Mont256<FrModulus> FrH;
typedef Mont256<FqModulus> FqH;
typedef FrH::Element FrElement;
struct G1PointAffine { FqH::Element x, y; };       // (0, 0): the point at infinity (the reference's zeroAffine)

// Keccak[r = 1088, c = 512] with the 0x01 suffix (keccak_wrapper.cpp:9), 32 bytes out
inline void keccak256(const uint8_t* data, size_t size, uint8_t out[32])
{
    static const uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL, 0x000000000000808BULL, 0x0000000080000001ULL,
                                    0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
                                    0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL,
                                    0x000000000000800AULL, 0x800000008000000AULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    static const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};     // [x + 5 y]
    uint64_t A[25];
    memset(A, 0, sizeof A);
    auto rol = [](uint64_t v, int n) { return n ? (v << n) | (v >> (64 - n)) : v; };
    auto permute = [&]() {
        for (int r = 0; r < 24; r++) {
            uint64_t C[5], D[5], B[25];
            for (int x = 0; x < 5; x++) C[x] = A[x] ^ A[x + 5] ^ A[x + 10] ^ A[x + 15] ^ A[x + 20];
            for (int x = 0; x < 5; x++) D[x] = C[(x + 4) % 5] ^ rol(C[(x + 1) % 5], 1);
            for (int i = 0; i < 25; i++) A[i] ^= D[i % 5];
            for (int x = 0; x < 5; x++) for (int y = 0; y < 5; y++) B[y + 5 * ((2 * x + 3 * y) % 5)] = rol(A[x + 5 * y], ROT[x + 5 * y]);
            for (int x = 0; x < 5; x++) for (int y = 0; y < 5; y++) A[x + 5 * y] = B[x + 5 * y] ^ (~B[(x + 1) % 5 + 5 * y] & B[(x + 2) % 5 + 5 * y]);
            A[0] ^= RC[r];
        }
    };
    const size_t rate = 136;
    std::vector<uint8_t> msg(data, data + size);
    msg.push_back(0x01);
    while (msg.size() % rate) msg.push_back(0);
    msg.back() |= 0x80;
    for (size_t off = 0; off < msg.size(); off += rate) {
        for (size_t i = 0; i < rate / 8; i++) { uint64_t w = 0; for (int b = 0; b < 8; b++) w |= (uint64_t)msg[off + 8 * i + b] << (8 * b); A[i] ^= w; }
        permute();
    }
    for (int i = 0; i < 4; i++) for (int b = 0; b < 8; b++) out[8 * i + b] = (uint8_t)(A[i] >> (8 * b));
}

// keccak_256_transcript.c.hpp:10-78
class Keccak256Transcript
{
    std::vector<uint8_t> data;
public:
    void reset() { data.clear(); }
    void addScalar(const FrElement& v) { uint8_t b[32]; FrH::toRprBE(v, b); data.insert(data.end(), b, b + 32); }
    void addPolCommitment(const G1PointAffine& p)
    {
        if (FqH::isZero(p.x) && FqH::isZero(p.y)) return;             // the point at infinity adds nothing (:58-61)
        uint8_t b[64];
        FqH::toRprBE(p.x, b); FqH::toRprBE(p.y, b + 32);
        data.insert(data.end(), b, b + 64);
    }
    FrElement getChallenge() { uint8_t h[32]; keccak256(data.data(), data.size(), h); return FrH::fromRprBE(h); }
};

namespace Zkey {
const int FFLONK_PROTOCOL_ID = 10;                                      // zkey.hpp:9
enum { ZKEY_HEADER_SECTION = 1, ZKEY_FF_HEADER_SECTION = 2, ZKEY_FF_ADDITIONS_SECTION = 3, ZKEY_FF_A_MAP_SECTION = 4, ZKEY_FF_B_MAP_SECTION = 5, ZKEY_FF_C_MAP_SECTION = 6,
       ZKEY_FF_QL_SECTION = 7, ZKEY_FF_QR_SECTION = 8, ZKEY_FF_QM_SECTION = 9, ZKEY_FF_QO_SECTION = 10, ZKEY_FF_QC_SECTION = 11, ZKEY_FF_SIGMA1_SECTION = 12,
       ZKEY_FF_SIGMA2_SECTION = 13, ZKEY_FF_SIGMA3_SECTION = 14, ZKEY_FF_LAGRANGE_SECTION = 15, ZKEY_FF_PTAU_SECTION = 16, ZKEY_FF_C0_SECTION = 17 };     // zkey_fflonk.hpp:9-25
inline int getProtocolIdFromZkey(BinFileUtils::BinFile* fd) { fd->startReadSection(ZKEY_HEADER_SECTION); const uint32_t id = fd->readU32LE(); fd->endReadSection(); return (int)id; }
#pragma pack(push, 1)
struct Addition { uint32_t signalId1, signalId2; FrElement factor1, factor2; };       // zkey.hpp:13-19
#pragma pack(pop)
struct FflonkZkeyHeader {                                               // zkey_fflonk.cpp:18-49
    uint32_t n8q = 0, n8r = 0, nVars = 0, nPublic = 0, domainSize = 0, nAdditions = 0, nConstraints = 0;
    uint8_t qPrime[32], rPrime[32];
    FrElement k1, k2, w3, w4, w8, wr;
    G1PointAffine C0;
    static FflonkZkeyHeader* loadFflonkZkeyHeader(BinFileUtils::BinFile* f)
    {
        FflonkZkeyHeader* h = new FflonkZkeyHeader();
        f->startReadSection(ZKEY_FF_HEADER_SECTION);
        h->n8q = f->readU32LE();
        if (h->n8q != 32) xlp_exit_process("zkey: base field of 32 bytes expected");
        memcpy(h->qPrime, f->read(32), 32);
        h->n8r = f->readU32LE();
        if (h->n8r != 32) xlp_exit_process("zkey: scalar field of 32 bytes expected");
        memcpy(h->rPrime, f->read(32), 32);
        h->nVars = f->readU32LE(); h->nPublic = f->readU32LE(); h->domainSize = f->readU32LE(); h->nAdditions = f->readU32LE(); h->nConstraints = f->readU32LE();
        FrElement* dst[6] = {&h->k1, &h->k2, &h->w3, &h->w4, &h->w8, &h->wr};
        for (FrElement* d : dst) memcpy(d, f->read(32), 32);
        f->read(128);                                                   // X2 = [tau]_2: the verifier's
        memcpy(&h->C0, f->read(64), 64);
        f->endReadSection();
        return h;
    }
};
}  // namespace Zkey

namespace WtnsUtils {
struct Header { uint32_t n8 = 0, nVars = 0; uint8_t prime[32]; };
inline std::unique_ptr<Header> loadHeader(BinFileUtils::BinFile* f)    // wtns_utils.cpp:13-26
{
    Header* h = new Header();
    f->startReadSection(1);
    h->n8 = f->readU32LE();
    if (h->n8 != 32) xlp_exit_process("wtns: field of 32 bytes expected");
    memcpy(h->prime, f->read(32), 32);
    h->nVars = f->readU32LE();
    f->endReadSection();
    return std::unique_ptr<Header>(h);
}
}  // namespace WtnsUtils

namespace Fflonk {

#define BLINDINGFACTORSLENGTH 10

class FflonkProver
{
    typedef uint64_t* DevFr;                                            // device vector of Fr elements, 4 words each
    Zkey::FflonkZkeyHeader* zkey = NULL;
    uint32_t zkeyPower = 0;
    uint64_t N = 0;
    std::vector<void*> owned;                                           // device allocations of the key / of a proof
    size_t keyAllocs = 0;
    std::map<std::string, DevFr> polPtr, evalPtr, buffers;
    uint64_t* PTau = NULL;                                              // 9 N affine points
    uint32_t* mapBuffers[3] = {NULL, NULL, NULL};
    std::vector<Zkey::Addition> additionsBuff;
    std::map<std::string, FrElement> toInverse, challenges;
    std::map<std::string, std::vector<FrElement>> roots;
    FrElement blindingFactors[BLINDINGFACTORSLENGTH];
    bool fixedBlinding = false;
    Keccak256Transcript transcript;
    std::map<std::string, G1PointAffine> polynomialCommitments;        // SnarkProof (snark_proof.c.hpp)
    std::map<std::string, FrElement> evaluationCommitments;
    std::vector<FrElement> R0, R1, R2;

    // reserved-memory mode (the reference's FflonkProver(E, reservedMemoryPtr, reservedMemorySize), prover.cpp:117: the STARKs' polynomial
    // memory doubles as the prover's): every device vector -- the key's too -- is carved from this region at the start of each proof, the
    // key uploaded again from the host image; nothing of the prover stays in HBM between final proofs (the zkEVM slab fills the card)
    uint32_t releaseMask = 0;                                           // scratch slots this proof could not be lent a region for
    char* reserved = NULL;
    uint64_t reservedBytes = 0, reservedUsed = 0;
    std::function<std::pair<void*, uint64_t>()> reservedProvider;
    BinFileUtils::BinFile* keyFile = NULL;
    FrElement* hostWitness = NULL;                                     // page-locked staging of the witness + additions
    void* rawAlloc(uint64_t bytes)
    {
        if (reserved) {
            const uint64_t at_ = (reservedUsed + 255) & ~255ULL;
            if (at_ + bytes > reservedBytes)
                xlp_exit_process(("Not enough reserved memory to generate a prove. Increase reserved memory size at least to " + std::to_string(at_ + bytes) + " bytes").c_str());
            reservedUsed = at_ + bytes;
            return reserved + at_;
        }
        void* p = NULL;
        XLP_CHECK(xlp_dev_alloc(&p, bytes ? bytes : 1));
        owned.push_back(p);
        return p;
    }
    DevFr dalloc(uint64_t elems) { return (DevFr)rawAlloc((elems ? elems : 1) * 32); }
    static DevFr at(DevFr p, uint64_t elem) { return p + 4 * elem; }
    FrElement root(uint32_t bits, uint64_t idx) const                  // FFT::root(domainPow, idx) (fft.c.hpp:52-84): w_(2^bits)^idx, w = 5^((r - 1) / 2^bits)
    {
        uint64_t e[4] = {FrModulus::Q[0] - 1, FrModulus::Q[1], FrModulus::Q[2], FrModulus::Q[3]};
        for (int i = 0; i < 4; i++) e[i] = (e[i] >> bits) | (i < 3 && bits ? e[i + 1] << (64 - bits) : 0);
        const FrElement w = FrH::pow(FrH::set(5), e);
        const uint64_t ie[4] = {idx, 0, 0, 0};
        return FrH::pow(w, ie);
    }
    std::vector<uint64_t> wpowTable(uint32_t bits) const               // w^(2^k), k < bits
    {
        std::vector<uint64_t> t(4 * (bits ? bits : 1));
        FrElement w = root(bits, 1);
        for (uint32_t k = 0; k < bits; k++) { memcpy(&t[4 * k], w.v, 32); w = FrH::square(w); }
        return t;
    }
    void fft(DevFr p, uint64_t n) { XLP_CHECK(xlp_dev_fft_bn254(p, n, 0)); }
    void ifft(DevFr p, uint64_t n) { XLP_CHECK(xlp_dev_fft_bn254(p, n, 1)); }
    void d2d(DevFr dst, const uint64_t* src, uint64_t elems) { XLP_CHECK(xlp_dev_copy_cols(dst, 4, src, 4, elems, 4)); }
    void zeroFill(DevFr p, uint64_t elems) { XLP_CHECK(xlp_dev_memset(p, 0, elems * 32)); }
    // Polynomial::fastEvaluate at up to 8 points per pass over the coefficients
    std::vector<FrElement> evaluate(DevFr pol, uint64_t len, const std::vector<FrElement>& pts)
    {
        std::vector<FrElement> out(pts.size());
        for (size_t o = 0; o < pts.size(); o += 8) {
            const uint32_t k = (uint32_t)std::min<size_t>(8, pts.size() - o);
            XLP_CHECK(xlp_dev_fr_eval(out[o].v, pol, len, pts[o].v, k));
        }
        return out;
    }
    FrElement evaluate1(DevFr pol, uint64_t len, const FrElement& x) { return evaluate(pol, len, std::vector<FrElement>(1, x))[0]; }
    // the degree checks of the reference ("... Polynomial is not well calculated"): the coefficients from `from` on must all vanish
    void requireZeroFrom(DevFr pol, uint64_t len, uint64_t from, const char* what)
    {
        if (from >= len) return;
        const FrElement v = evaluate1(at(pol, from), len - from, FrH::set(0x5EED0F0F1ULL));
        if (!FrH::isZero(v)) xlp_exit_process((std::string(what) + " Polynomial is not well calculated").c_str());
    }
    G1PointAffine multiExponentiation(DevFr pol, uint64_t count)       // FflonkProver::multiExponentiation (:1700-1715)
    {
        DevFr sc = buffers["tmp"];
        XLP_CHECK(xlp_dev_fr_from_mont(sc, pol, count));                // polynomialFromMontgomery
        uint64_t chunk = 1ULL << 26;                                    // (xlp_dev_msm_bn254 takes < 2^27 points per call)
        if (const char* e = getenv("XLP_FFLONK_MSM_CHUNK")) { const uint64_t v = strtoull(e, NULL, 10); if (v >= 16 && v < chunk) chunk = v; }     // tests: the partial sums' path at small sizes
        std::vector<G1PointAffine> parts;
        uint64_t* d_out = buffers["msm_out"];
        for (uint64_t o = 0; o < count; o += chunk) {
            const uint64_t k = std::min(chunk, count - o);
            XLP_CHECK(xlp_dev_msm_bn254(d_out, PTau + 8 * o, sc + 4 * o, k));
            G1PointAffine p;
            XLP_CHECK(xlp_copy_d2h(&p, d_out, 64));
            parts.push_back(p);
        }
        if (parts.size() == 1) return parts[0];
        std::vector<uint64_t> ones(4 * parts.size(), 0);               // the partial sums added by one more multi-exponentiation with scalars 1
        for (size_t i = 0; i < parts.size(); i++) ones[4 * i] = 1;
        XLP_CHECK(xlp_copy_h2d(buffers["msm_pts"], parts.data(), parts.size() * 64));
        XLP_CHECK(xlp_copy_h2d(sc, ones.data(), ones.size() * 8));
        XLP_CHECK(xlp_dev_msm_bn254(d_out, buffers["msm_pts"], sc, parts.size()));
        G1PointAffine p;
        XLP_CHECK(xlp_copy_d2h(&p, d_out, 64));
        return p;
    }
    static std::vector<FrElement> polyMulXSub(const std::vector<FrElement>& c, const FrElement& v)
    {
        std::vector<FrElement> o(c.size() + 1, FrH::zero());
        for (size_t i = 0; i < c.size(); i++) { o[i + 1] = FrH::add(o[i + 1], c[i]); o[i] = FrH::sub(o[i], FrH::mul(c[i], v)); }
        return o;
    }
    static FrElement hostEval(const std::vector<FrElement>& c, const FrElement& x)
    {
        FrElement r = FrH::zero();
        for (size_t i = c.size(); i > 0; i--) r = FrH::add(FrH::mul(r, x), c[i - 1]);
        return r;
    }
    // Polynomial::lagrangePolynomialInterpolation (polynomial.c.hpp:674-712)
    static std::vector<FrElement> lagrangeInterpolation(const std::vector<FrElement>& xs, const std::vector<FrElement>& ys)
    {
        std::vector<FrElement> out(xs.size(), FrH::zero());
        for (size_t i = 0; i < xs.size(); i++) {
            std::vector<FrElement> c(1, FrH::one());
            for (size_t j = 0; j < xs.size(); j++) if (j != i) c = polyMulXSub(c, xs[j]);
            const FrElement f = FrH::mul(ys[i], FrH::inv(hostEval(c, xs[i])));
            for (size_t k = 0; k < c.size(); k++) out[k] = FrH::add(out[k], FrH::mul(c[k], f));
        }
        return out;
    }
    static FrElement zerofierAt(const std::vector<FrElement>& xs, const FrElement& y)      // zerofierPolynomial(...)->fastEvaluate(y)
    {
        FrElement r = FrH::one();
        for (const FrElement& x : xs) r = FrH::mul(r, FrH::sub(y, x));
        return r;
    }
    void lincomb(DevFr out, uint64_t len, const std::vector<std::tuple<const uint64_t*, uint64_t, FrElement>>& terms)
    {
        const uint64_t* p[6]; uint64_t l[6]; uint64_t k[24];
        if (terms.size() > 6) xlp_exit_process("FflonkProver: too many terms");
        for (size_t t = 0; t < terms.size(); t++) { p[t] = std::get<0>(terms[t]); l[t] = std::get<1>(terms[t]); memcpy(&k[4 * t], std::get<2>(terms[t]).v, 32); }
        XLP_CHECK(xlp_dev_fr_lincomb(out, len, (uint32_t)terms.size(), p, l, k));
    }
    DevFr uploadSmall(const char* name, const std::vector<FrElement>& c)
    {
        XLP_CHECK(xlp_copy_h2d(buffers[name], c.data(), c.size() * 32));
        return buffers[name];
    }

public:
    std::vector<std::pair<std::string, double>> lastTimers;
    FflonkProver() { memset(blindingFactors, 0, sizeof blindingFactors); }
    // reserved-memory mode: the region is asked for at the start of every proof (the STARKs' device slab may have been re-reserved since)
    explicit FflonkProver(std::function<std::pair<void*, uint64_t>()> reservedDeviceMemory) : reservedProvider(reservedDeviceMemory)
    {
        memset(blindingFactors, 0, sizeof blindingFactors);
        reserved = (char*)1;                               // (non-null: nothing is allocated before the first proof)
    }
    ~FflonkProver() { removePrecomputedData(); }
    FflonkProver(const FflonkProver&) = delete;
    FflonkProver& operator=(const FflonkProver&) = delete;

    // test hook: b_1 .. b_9 as integers mod r (4 little-endian words each, NOT Montgomery); the reference draws them at random (:645-650)
    void setBlindingFactors(const uint64_t* b9x4)
    {
        for (int i = 1; i < BLINDINGFACTORSLENGTH; i++) { FrElement raw; memcpy(raw.v, b9x4 + 4 * (i - 1), 32); blindingFactors[i] = FrH::toMontgomery(raw); }
        fixedBlinding = true;
    }
    void removePrecomputedData()
    {
        if (hostWitness) { xlp_host_free(hostWitness); hostWitness = NULL; }
        for (void* p : owned) xlp_dev_free(p);
        owned.clear(); polPtr.clear(); evalPtr.clear(); buffers.clear();
        delete zkey; zkey = NULL;
    }

    // FflonkProver::setZkey (:94-330): everything circuit-dependent goes to the GPU once (reserved-memory mode: at every proof; the file
    // image must then outlive the prover)
    void setZkey(BinFileUtils::BinFile* fdZkey)
    {
        if (zkey) removePrecomputedData();
        XLP_CHECK(xlp_init(0));
        keyFile = fdZkey;
        if (Zkey::getProtocolIdFromZkey(fdZkey) != Zkey::FFLONK_PROTOCOL_ID) xlp_exit_process("zkey file is not fflonk");
        zkey = Zkey::FflonkZkeyHeader::loadFflonkZkeyHeader(fdZkey);
        if (memcmp(zkey->rPrime, FrModulus::Q, 32) != 0) xlp_exit_process("zkey curve not supported");
        N = zkey->domainSize;
        if (N < 8 || (N & (N - 1))) xlp_exit_process("zkey: the domain size must be a power of two");
        zkeyPower = 0;
        while ((1ULL << zkeyPower) < N) zkeyPower++;
        if (zkey->nPublic < 1) xlp_exit_process("zkey: at least one public input expected (the Lagrange section holds max(nPublic, 1) polynomials)");
        if (zkey->nConstraints > N - 2) xlp_exit_process("zkey: more constraints than rows (the last two rows are the prover's)");
        if (fdZkey->getSectionSize(Zkey::ZKEY_FF_ADDITIONS_SECTION) < (uint64_t)zkey->nAdditions * sizeof(Zkey::Addition)) xlp_exit_process("zkey: additions section too short");
        additionsBuff.resize(zkey->nAdditions);
        if (zkey->nAdditions) memcpy(additionsBuff.data(), fdZkey->getSectionData(Zkey::ZKEY_FF_ADDITIONS_SECTION), (size_t)zkey->nAdditions * sizeof(Zkey::Addition));
        void* hw = NULL;
        XLP_CHECK(xlp_host_alloc(&hw, ((uint64_t)zkey->nVars + 1) * 32));
        hostWitness = (FrElement*)hw;
        if (!reserved) deviceSetup();
    }

private:
    void deviceSetup()
    {
        BinFileUtils::BinFile* fdZkey = keyFile;
        if (reservedProvider) {
            const auto r = reservedProvider();
            if (!r.first) xlp_exit_process("FflonkProver: no reserved device memory");
            reserved = (char*)r.first; reservedBytes = r.second;
        }
        reservedUsed = 0;
        polPtr.clear(); evalPtr.clear(); buffers.clear();
        auto section = [&](int id, uint64_t elems) {
            if (fdZkey->getSectionSize(id) < elems * 32) xlp_exit_process(("zkey: section " + std::to_string(id) + " too short").c_str());
            return (const uint64_t*)fdZkey->getSectionData(id);
        };
        const char* names[8] = {"QL", "QR", "QM", "QO", "QC", "Sigma1", "Sigma2", "Sigma3"};
        const int ids[8] = {Zkey::ZKEY_FF_QL_SECTION, Zkey::ZKEY_FF_QR_SECTION, Zkey::ZKEY_FF_QM_SECTION, Zkey::ZKEY_FF_QO_SECTION, Zkey::ZKEY_FF_QC_SECTION,
                            Zkey::ZKEY_FF_SIGMA1_SECTION, Zkey::ZKEY_FF_SIGMA2_SECTION, Zkey::ZKEY_FF_SIGMA3_SECTION};
        for (int k = 0; k < 8; k++) {                                  // coefficients (N) then the 4N evaluations
            const uint64_t* s = section(ids[k], 5 * N);
            polPtr[names[k]] = dalloc(N); evalPtr[names[k]] = dalloc(4 * N);
            XLP_CHECK(xlp_copy_h2d(polPtr[names[k]], s, N * 32));
            XLP_CHECK(xlp_copy_h2d(evalPtr[names[k]], s + 4 * N, 4 * N * 32));
        }
        polPtr["C0"] = dalloc(8 * N);
        XLP_CHECK(xlp_copy_h2d(polPtr["C0"], section(Zkey::ZKEY_FF_C0_SECTION, 8 * N), 8 * N * 32));
        evalPtr["lagrange"] = dalloc(4 * N * zkey->nPublic);            // per public input: N coefficients, then 4N evaluations (:282-286)
        const uint64_t* lag = section(Zkey::ZKEY_FF_LAGRANGE_SECTION, 5 * N * zkey->nPublic);
        for (uint32_t i = 0; i < zkey->nPublic; i++) XLP_CHECK(xlp_copy_h2d(at(evalPtr["lagrange"], 4 * N * i), lag + 4 * (N + 5 * N * i), 4 * N * 32));
        if (fdZkey->getSectionSize(Zkey::ZKEY_FF_PTAU_SECTION) < 9 * N * 64) xlp_exit_process("zkey: powers of tau section too short");
        PTau = dalloc(18 * N);                                          // 9 N points of 64 bytes
        XLP_CHECK(xlp_copy_h2d(PTau, fdZkey->getSectionData(Zkey::ZKEY_FF_PTAU_SECTION), 9 * N * 64));
        const int mapIds[3] = {Zkey::ZKEY_FF_A_MAP_SECTION, Zkey::ZKEY_FF_B_MAP_SECTION, Zkey::ZKEY_FF_C_MAP_SECTION};
        for (int k = 0; k < 3; k++) {
            if (fdZkey->getSectionSize(mapIds[k]) < (uint64_t)zkey->nConstraints * 4) xlp_exit_process("zkey: map section too short");
            void* p = rawAlloc(((uint64_t)zkey->nConstraints + 1) * 4);
            mapBuffers[k] = (uint32_t*)p;
            XLP_CHECK(xlp_copy_h2d(p, fdZkey->getSectionData(mapIds[k]), (uint64_t)zkey->nConstraints * 4));
        }
        // per-proof vectors (the reference's nonPrecomputedBigBuffer, :296-371; here every polynomial has its own buffer)
        const std::pair<const char*, uint64_t> pols[] = {{"A", N}, {"B", N}, {"C", N}, {"T0", 4 * N}, {"C1", 8 * N}, {"Z", 2 * N}, {"T1", 2 * N}, {"T1z", 2 * N}, {"T2", 4 * N},
                                                         {"T2z", 4 * N}, {"C2", 16 * N}, {"F", 16 * N}, {"L", 16 * N}, {"tmp2", 16 * N}};
        for (const auto& kv : pols) polPtr[kv.first] = dalloc(kv.second);
        for (const char* e : {"A", "B", "C", "Z"}) evalPtr[e] = dalloc(4 * N);
        for (const char* b : {"A", "B", "C"}) buffers[b] = dalloc(N);
        buffers["tmp"] = dalloc(16 * N);
        buffers["numArr"] = dalloc(N); buffers["denArr"] = dalloc(N); buffers["Z"] = dalloc(N);
        buffers["witness"] = dalloc(zkey->nVars + 1);
        buffers["small"] = dalloc(64); buffers["msm_out"] = dalloc(2); buffers["msm_pts"] = dalloc(2 * 64);
        XLP_CHECK(xlp_sync());
        keyAllocs = owned.size();
    }
public:

    // FflonkProver::prove (:427-596).  Returns (proof JSON, public signals JSON) as text.
    std::pair<std::string, std::string> prove(BinFileUtils::BinFile* fdZkey, BinFileUtils::BinFile* fdWtns) { setZkey(fdZkey); return prove(fdWtns); }
    std::pair<std::string, std::string> prove(BinFileUtils::BinFile* fdWtns)
    {
        auto h = WtnsUtils::loadHeader(fdWtns);
        if (fdWtns->getSectionSize(2) < (uint64_t)h->nVars * 32) xlp_exit_process("wtns: data section too short");
        return prove((FrElement*)fdWtns->getSectionData(2), h.get());
    }
    std::pair<std::string, std::string> prove(FrElement* buffWitness, WtnsUtils::Header* wtnsHeader = NULL)
    {
        if (!zkey) xlp_exit_process("Zkey data not set");
        if (wtnsHeader) {
            if (memcmp(zkey->rPrime, wtnsHeader->prime, 32) != 0) xlp_exit_process("Curve of the witness does not match the curve of the proving key");
            if (wtnsHeader->nVars != zkey->nVars - zkey->nAdditions)
                xlp_exit_process(("Invalid witness length. Circuit: " + std::to_string(zkey->nVars) + ", witness: " + std::to_string(wtnsHeader->nVars) + ", " + std::to_string(zkey->nAdditions)).c_str());
        }
        lastTimers.clear();
        if (reserved) {
            timed("FFLONK_KEY_TO_DEVICE", [&] { deviceSetup(); });
            // what is left of the region serves as the library's scratch for the proof: the transforms' twiddle table (slot 8), scans / evaluations (4), multi-exponentiation (7)
            releaseMask = 0;
            auto lend = [&](uint32_t slot, uint64_t bytes) {
                const uint64_t at_ = (reservedUsed + 255) & ~255ULL;
                if (at_ + bytes > reservedBytes) { releaseMask |= 1u << slot; return; }      // (too small: the slot grows its own scratch and it is released below)
                XLP_CHECK(xlp_scratch_lend(slot, reserved + at_, bytes));
                reservedUsed = at_ + bytes;
            };
            lend(8, 80 * N + (1 << 12));
            lend(4, 128 * N + (1 << 20));
            const uint64_t at_ = (reservedUsed + 255) & ~255ULL;
            if (at_ < reservedBytes) XLP_CHECK(xlp_scratch_lend(7, reserved + at_, reservedBytes - at_));
        }
        transcript.reset(); polynomialCommitments.clear(); evaluationCommitments.clear(); toInverse.clear(); challenges.clear(); roots.clear();
        // the witness with the additions behind it (calculateAdditions :598-612: an addition may read earlier additions, so a host loop as in
        // the reference; values stay in the .wtns form, NOT Montgomery: factor (Montgomery) x value (plain) = plain)
        const uint64_t nw = zkey->nVars - zkey->nAdditions;
        FrElement* wit = hostWitness;                                   // page-locked: the upload below runs at PCIe speed (a pageable vector: 5x slower)
        memcpy(wit, buffWitness, nw * 32);
        wit[0] = FrH::zero();                                           // :511
        for (uint32_t i = 0; i < zkey->nAdditions; i++) {
            const Zkey::Addition& ad = additionsBuff[i];
            auto get = [&](uint32_t idx) { return idx < nw + i ? wit[idx] : FrH::zero(); };      // getWitness (:615-629); later additions are not there yet
            wit[nw + i] = FrH::add(FrH::mul(ad.factor1, get(ad.signalId1)), FrH::mul(ad.factor2, get(ad.signalId2)));
        }
        XLP_CHECK(xlp_copy_h2d(buffers["witness"], wit, (uint64_t)zkey->nVars * 32));
        timed("FFLONK_ROUND_1", [&] { round1(); });
        timed("FFLONK_ROUND_2", [&] { round2(); });
        timed("FFLONK_ROUND_3", [&] { round3(); });
        timed("FFLONK_ROUND_4", [&] { round4(); });
        timed("FFLONK_ROUND_5", [&] { round5(); });
        evaluationCommitments["inv"] = getMontgomeryBatchedInverse();
        if (reserved) {                                                 // the card goes back to the STARKs as it was
            for (uint32_t slot : {8u, 4u, 7u}) XLP_CHECK(xlp_scratch_lend(slot, NULL, 0));
            // ... and what THIS proof made the library grow goes back: the slots only this prover uses (7: multi-exponentiation, 8: twiddles -- a lent
            // region that a request outgrew falls through to the slot's own scratch) and slot 4 when its lending did not fit.  Slots 4 / 5 otherwise
            // serve the STARKs' step programs (csrc/expr.hip), which would re-allocate them inside the next batch proof's timers
            XLP_CHECK(xlp_scratch_release((1u << 7) | (1u << 8) | (releaseMask & (1u << 4))));
        }
        std::string pub = "[";
        for (uint32_t i = 1; i <= zkey->nPublic; i++) pub += std::string(i > 1 ? "," : "") + "\"" + FrH::toString(FrH::toMontgomery(wit[i])) + "\"";      // :546-553
        pub += "]";
        return {toJson(), pub};
    }

private:
    template <class F>
    void timed(const char* name, F f)
    {
        XLP_CHECK(xlp_sync());
        const auto t0 = std::chrono::steady_clock::now();
        f();
        XLP_CHECK(xlp_sync());
        lastTimers.emplace_back(name, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    // SnarkProof::toJson (snark_proof.c.hpp:62-88); keys in the order nlohmann's ordered map prints them
    std::string toJson()
    {
        std::string s = "{\"curve\":\"bn128\",\"evaluations\":{";
        bool first = true;
        for (const auto& kv : evaluationCommitments) { s += std::string(first ? "" : ",") + "\"" + kv.first + "\":\"" + FrH::toString(kv.second) + "\""; first = false; }
        s += "},\"polynomials\":{";
        first = true;
        for (const auto& kv : polynomialCommitments) {
            const bool inf = FqH::isZero(kv.second.x) && FqH::isZero(kv.second.y);
            s += std::string(first ? "" : ",") + "\"" + kv.first + "\":[\"" + (inf ? "0" : FqH::toString(kv.second.x)) + "\",\"" + (inf ? "1" : FqH::toString(kv.second.y)) + "\",\"" + (inf ? "0" : "1") + "\"]";
            first = false;
        }
        return s + "},\"protocol\":\"fflonk\"}";
    }

    // ---- ROUND 1 (:641-686)
    void round1()
    {
        if (!fixedBlinding) {                                           // 31 random bytes per factor, taken as they are (:645-650)
            FILE* f = fopen("/dev/urandom", "rb");
            for (int i = 1; i < BLINDINGFACTORSLENGTH; i++) {
                memset(blindingFactors[i].v, 0, 32);
                if (!f || fread(blindingFactors[i].v, 1, 31, f) != 31) xlp_exit_process("FflonkProver: no random bytes");
            }
            if (f) fclose(f);
        }
        const FrElement bA[2] = {blindingFactors[2], blindingFactors[1]}, bB[2] = {blindingFactors[4], blindingFactors[3]}, bC[2] = {blindingFactors[6], blindingFactors[5]};
        computeWirePolynomial("A", 0, bA); computeWirePolynomial("B", 1, bB); computeWirePolynomial("C", 2, bC);
        // computeT0 (:727-788)
        XLP_CHECK(xlp_dev_fflonk_t0(polPtr["T0"], evalPtr["A"], evalPtr["B"], evalPtr["C"], evalPtr["QL"], evalPtr["QR"], evalPtr["QM"], evalPtr["QO"], evalPtr["QC"], evalPtr["lagrange"],
                                    buffers["A"], zkey->nPublic, 4 * N));
        ifft(polPtr["T0"], 4 * N);
        XLP_CHECK(xlp_dev_fr_div_zh(polPtr["T0"], 4 * N, N));
        requireZeroFrom(polPtr["T0"], 4 * N, 2 * N - 2, "T0");
        // computeC1 (:791-811): a(X^4) + X b(X^4) + X^2 c(X^4) + X^3 T0(X^4)
        const uint64_t* ps[4] = {polPtr["A"], polPtr["B"], polPtr["C"], polPtr["T0"]};
        const uint64_t ls[4] = {N, N, N, 2 * N};
        XLP_CHECK(xlp_dev_fr_interleave(polPtr["C1"], 8 * N, 4, ps, ls));
        polynomialCommitments["C1"] = multiExponentiation(polPtr["C1"], 8 * N);
    }
    void computeWirePolynomial(const char* name, int col, const FrElement bf[2])     // :697-724
    {
        DevFr buf = buffers[name];
        zeroFill(buf, N);
        XLP_CHECK(xlp_dev_fr_gather_mont(buf, mapBuffers[col], zkey->nConstraints, buffers["witness"], zkey->nVars));
        XLP_CHECK(xlp_copy_h2d(at(buf, N - 2), bf[1].v, 32));
        XLP_CHECK(xlp_copy_h2d(at(buf, N - 1), bf[0].v, 32));
        d2d(polPtr[name], buf, N);
        ifft(polPtr[name], N);
        zeroFill(evalPtr[name], 4 * N);
        d2d(evalPtr[name], polPtr[name], N);
        fft(evalPtr[name], 4 * N);
    }

    // ---- ROUND 2 (:814-868)
    void round2()
    {
        transcript.reset();
        transcript.addPolCommitment(zkey->C0);
        std::vector<FrElement> pubs(zkey->nPublic);
        XLP_CHECK(xlp_copy_d2h(pubs.data(), buffers["A"], (uint64_t)zkey->nPublic * 32));
        for (const FrElement& v : pubs) transcript.addScalar(v);
        transcript.addPolCommitment(polynomialCommitments["C1"]);
        challenges["beta"] = transcript.getChallenge();
        transcript.reset();
        transcript.addScalar(challenges["beta"]);
        challenges["gamma"] = transcript.getChallenge();
        // computeZ (:871-964)
        {
            uint64_t sc[16];
            memcpy(sc, challenges["beta"].v, 32); memcpy(sc + 4, challenges["gamma"].v, 32); memcpy(sc + 8, zkey->k1.v, 32); memcpy(sc + 12, zkey->k2.v, 32);
            const std::vector<uint64_t> wp = wpowTable(zkeyPower);
            XLP_CHECK(xlp_dev_fflonk_z_numden(buffers["numArr"], buffers["denArr"], buffers["A"], buffers["B"], buffers["C"], evalPtr["Sigma1"], evalPtr["Sigma2"], evalPtr["Sigma3"], N, sc,
                                              wp.data(), zkeyPower));
            DevFr t = buffers["tmp"];                                   // num / den row by row, then the running product (the same values as the
            XLP_CHECK(xlp_dev_fr_batch_inverse(t, buffers["denArr"], N));       // reference's two running products divided at the end)
            XLP_CHECK(xlp_dev_fr_mul(buffers["numArr"], buffers["numArr"], t, N));
            XLP_CHECK(xlp_dev_fr_prefix_product(t, buffers["numArr"], N));
            XLP_CHECK(xlp_dev_fflonk_z_rotate(buffers["Z"], t, N));
            FrElement z0;
            XLP_CHECK(xlp_copy_d2h(&z0, buffers["Z"], 32));
            if (!FrH::eq(z0, FrH::one())) xlp_exit_process("Copy constraints does not match");
            zeroFill(polPtr["Z"], 2 * N);
            d2d(polPtr["Z"], buffers["Z"], N);
            ifft(polPtr["Z"], N);
            zeroFill(evalPtr["Z"], 4 * N);
            d2d(evalPtr["Z"], polPtr["Z"], N);
            fft(evalPtr["Z"], 4 * N);
            // blindCoefficients (polynomial.c.hpp:148-157) with (b9, b8, b7): coef[N + i] += bf[i], coef[i] -= bf[i]
            const FrElement bf[3] = {blindingFactors[9], blindingFactors[8], blindingFactors[7]};
            FrElement lo[3];
            XLP_CHECK(xlp_copy_d2h(lo, polPtr["Z"], 96));
            for (int i = 0; i < 3; i++) lo[i] = FrH::sub(lo[i], bf[i]);
            XLP_CHECK(xlp_copy_h2d(polPtr["Z"], lo, 96));
            XLP_CHECK(xlp_copy_h2d(at(polPtr["Z"], N), bf, 96));
        }
        uint64_t bl[12];
        memcpy(bl, blindingFactors[7].v, 32); memcpy(bl + 4, blindingFactors[8].v, 32); memcpy(bl + 8, blindingFactors[9].v, 32);
        // computeT1 (:982-1034)
        {
            const std::vector<uint64_t> wp = wpowTable(zkeyPower + 1);
            XLP_CHECK(xlp_dev_fflonk_t1(polPtr["T1"], polPtr["T1z"], evalPtr["Z"], evalPtr["lagrange"], 2 * N, bl, wp.data(), zkeyPower + 1));
            ifft(polPtr["T1"], 2 * N);
            XLP_CHECK(xlp_dev_fr_div_zh(polPtr["T1"], 2 * N, N));
            ifft(polPtr["T1z"], 2 * N);
            lincomb(polPtr["T1"], 2 * N, {{polPtr["T1"], 2 * N, FrH::one()}, {polPtr["T1z"], 2 * N, FrH::one()}});
            requireZeroFrom(polPtr["T1"], 2 * N, N + 2, "T1");
        }
        // computeT2 (:1037-1131)
        {
            uint64_t sc[32];
            const FrElement wn = root(zkeyPower, 1);
            const FrElement* src[8] = {&challenges["beta"], &challenges["gamma"], &zkey->k1, &zkey->k2, &wn, &blindingFactors[7], &blindingFactors[8], &blindingFactors[9]};
            for (int i = 0; i < 8; i++) memcpy(sc + 4 * i, src[i]->v, 32);
            const std::vector<uint64_t> wp = wpowTable(zkeyPower + 2);
            XLP_CHECK(xlp_dev_fflonk_t2(polPtr["T2"], polPtr["T2z"], evalPtr["A"], evalPtr["B"], evalPtr["C"], evalPtr["Z"], evalPtr["Sigma1"], evalPtr["Sigma2"], evalPtr["Sigma3"], 4 * N, sc,
                                        wp.data(), zkeyPower + 2));
            ifft(polPtr["T2"], 4 * N);
            XLP_CHECK(xlp_dev_fr_div_zh(polPtr["T2"], 4 * N, N));
            ifft(polPtr["T2z"], 4 * N);
            lincomb(polPtr["T2"], 4 * N, {{polPtr["T2"], 4 * N, FrH::one()}, {polPtr["T2z"], 4 * N, FrH::one()}});
            requireZeroFrom(polPtr["T2"], 4 * N, 3 * N, "T2");
        }
        // computeC2 (:1134-1150): z(X^3) + X T1(X^3) + X^2 T2(X^3)
        const uint64_t* ps[3] = {polPtr["Z"], polPtr["T1"], polPtr["T2"]};
        const uint64_t ls[3] = {N + 3, N + 2, 3 * N};
        XLP_CHECK(xlp_dev_fr_interleave(polPtr["C2"], 16 * N, 3, ps, ls));
        polynomialCommitments["C2"] = multiExponentiation(polPtr["C2"], 9 * N);
    }

    // ---- ROUND 3 (:1153-1241)
    void round3()
    {
        transcript.reset();
        transcript.addScalar(challenges["gamma"]);
        transcript.addPolCommitment(polynomialCommitments["C2"]);
        challenges["xiSeed"] = transcript.getChallenge();
        const FrElement seed = challenges["xiSeed"], seed2 = FrH::square(seed);
        auto powers = [&](const FrElement& w, int n) { std::vector<FrElement> r(n, FrH::one()); for (int i = 1; i < n; i++) r[i] = FrH::mul(r[i - 1], w); return r; };
        roots["w8"] = powers(zkey->w8, 8); roots["w4"] = powers(zkey->w4, 4); roots["w3"] = powers(zkey->w3, 3);
        auto times = [&](const FrElement& h, const std::vector<FrElement>& ws) { std::vector<FrElement> r(ws.size()); for (size_t i = 0; i < ws.size(); i++) r[i] = i ? FrH::mul(h, ws[i]) : h; return r; };
        const FrElement h0 = FrH::mul(seed2, seed), h1 = FrH::square(h0), h2 = FrH::mul(h1, seed2), h3 = FrH::mul(h2, zkey->wr);
        roots["S0h0"] = times(h0, roots["w8"]); roots["S1h1"] = times(h1, roots["w4"]); roots["S2h2"] = times(h2, roots["w3"]); roots["S2h3"] = times(h3, roots["w3"]);
        challenges["xi"] = FrH::mul(FrH::square(h2), h2);
        challenges["xiw"] = FrH::mul(challenges["xi"], root(zkeyPower, 1));
        const FrElement xi = challenges["xi"], xiw = challenges["xiw"];
        const std::pair<const char*, const char*> at1[] = {{"ql", "QL"}, {"qr", "QR"}, {"qm", "QM"}, {"qo", "QO"}, {"qc", "QC"}, {"s1", "Sigma1"}, {"s2", "Sigma2"}, {"s3", "Sigma3"},
                                                           {"a", "A"}, {"b", "B"}, {"c", "C"}};
        for (const auto& kv : at1) evaluationCommitments[kv.first] = evaluate1(polPtr[kv.second], N, xi);
        const std::vector<FrElement> zv = evaluate(polPtr["Z"], N + 3, {xi, xiw});
        evaluationCommitments["z"] = zv[0]; evaluationCommitments["zw"] = zv[1];
        evaluationCommitments["t1w"] = evaluate1(polPtr["T1"], N + 2, xiw);
        evaluationCommitments["t2w"] = evaluate1(polPtr["T2"], 3 * N, xiw);
    }

    // ---- ROUND 4 (:1244-1289)
    void round4()
    {
        transcript.reset();
        transcript.addScalar(challenges["xiSeed"]);
        for (const char* k : {"ql", "qr", "qm", "qo", "qc", "s1", "s2", "s3", "a", "b", "c", "z", "zw", "t1w", "t2w"}) transcript.addScalar(evaluationCommitments[k]);
        challenges["alpha"] = transcript.getChallenge();
        const FrElement alpha = challenges["alpha"], alpha2 = FrH::square(alpha), xi = challenges["xi"], xiw = challenges["xiw"];
        // computeR0 / R1 / R2 (:1292-1367): the combined polynomials at their 8 / 4 / 6 roots, interpolated
        R0 = lagrangeInterpolation(roots["S0h0"], evaluate(polPtr["C0"], 8 * N, roots["S0h0"]));
        R1 = lagrangeInterpolation(roots["S1h1"], evaluate(polPtr["C1"], 8 * N, roots["S1h1"]));
        std::vector<FrElement> s2 = roots["S2h2"];
        s2.insert(s2.end(), roots["S2h3"].begin(), roots["S2h3"].end());
        R2 = lagrangeInterpolation(s2, evaluate(polPtr["C2"], 9 * N, s2));
        // computeF (:1370-1401)
        std::vector<FrElement> small(24, FrH::zero());
        std::copy(R0.begin(), R0.end(), small.begin()); std::copy(R1.begin(), R1.end(), small.begin() + 8); std::copy(R2.begin(), R2.end(), small.begin() + 16);
        DevFr ds = uploadSmall("small", small);
        DevFr F = polPtr["F"], t = polPtr["tmp2"], u = polPtr["L"];
        const FrElement mOne = FrH::neg(FrH::one());
        lincomb(t, 16 * N, {{polPtr["C2"], 16 * N, alpha2}, {at(ds, 16), 6, FrH::neg(alpha2)}});                 // (C2 - R2) alpha^2
        XLP_CHECK(xlp_dev_fr_div_xm_minus(u, t, 16 * N, 3, xi.v));
        XLP_CHECK(xlp_dev_fr_div_xm_minus(F, u, 16 * N, 3, xiw.v));
        lincomb(t, 8 * N, {{polPtr["C1"], 8 * N, alpha}, {at(ds, 8), 4, FrH::neg(alpha)}});                      // (C1 - R1) alpha
        XLP_CHECK(xlp_dev_fr_div_xm_minus(u, t, 8 * N, 4, xi.v));
        lincomb(F, 16 * N, {{F, 16 * N, FrH::one()}, {u, 8 * N, FrH::one()}});
        lincomb(t, 8 * N, {{polPtr["C0"], 8 * N, FrH::one()}, {ds, 8, mOne}});                                   // C0 - R0
        XLP_CHECK(xlp_dev_fr_div_xm_minus(u, t, 8 * N, 8, xi.v));
        lincomb(F, 16 * N, {{F, 16 * N, FrH::one()}, {u, 8 * N, FrH::one()}});
        requireZeroFrom(F, 16 * N, 9 * N - 6, "F");
        polynomialCommitments["W1"] = multiExponentiation(F, 9 * N);
    }

    // ---- ROUND 5 (:1417-1459)
    void round5()
    {
        transcript.reset();
        transcript.addScalar(challenges["alpha"]);
        transcript.addPolCommitment(polynomialCommitments["W1"]);
        challenges["y"] = transcript.getChallenge();
        const FrElement y = challenges["y"], alpha = challenges["alpha"];
        // computeL (:1462-1528)
        std::vector<FrElement> s2 = roots["S2h2"];
        s2.insert(s2.end(), roots["S2h3"].begin(), roots["S2h3"].end());
        const FrElement mulL0 = zerofierAt(roots["S0h0"], y), mulL1 = zerofierAt(roots["S1h1"], y), mulL2 = zerofierAt(s2, y);
        const FrElement preL0 = FrH::mul(mulL1, mulL2), preL1 = FrH::mul(alpha, FrH::mul(mulL0, mulL2)), preL2 = FrH::mul(FrH::square(alpha), FrH::mul(mulL0, mulL1));
        toInverse["denH1"] = mulL1; toInverse["denH2"] = mulL2;
        const FrElement evalZTY = FrH::mul(mulL0, FrH::mul(mulL1, mulL2));      // ZT(y), ZT = the zerofier of all 18 roots
        const FrElement ZTS2Y = FrH::inv(FrH::mul(mulL1, mulL2));               // 1 / ZTS2(y)
        // L = [preL2 (C2 - R2(y)) + preL1 (C1 - R1(y)) + preL0 (C0 - R0(y)) - ZT(y) F] / ZTS2(y): one pass; the constant terms through a one-element vector
        FrElement cst = FrH::add(FrH::add(FrH::mul(preL2, hostEval(R2, y)), FrH::mul(preL1, hostEval(R1, y))), FrH::mul(preL0, hostEval(R0, y)));
        cst = FrH::neg(FrH::mul(cst, ZTS2Y));
        DevFr ds = uploadSmall("small", std::vector<FrElement>(1, FrH::one()));
        DevFr t = polPtr["tmp2"], L = polPtr["L"];
        lincomb(t, 16 * N, {{polPtr["C2"], 16 * N, FrH::mul(preL2, ZTS2Y)}, {polPtr["C1"], 8 * N, FrH::mul(preL1, ZTS2Y)}, {polPtr["C0"], 8 * N, FrH::mul(preL0, ZTS2Y)},
                            {polPtr["F"], 16 * N, FrH::neg(FrH::mul(evalZTY, ZTS2Y))}, {ds, 1, cst}});
        XLP_CHECK(xlp_dev_fr_div_xm_minus(L, t, 16 * N, 1, y.v));              // / (X - y): divByZerofier(1, y)
        requireZeroFrom(L, 16 * N, 9 * N - 1, "L");
        polynomialCommitments["W2"] = multiExponentiation(L, 9 * N);
    }

    // getMontgomeryBatchedInverse (:1557-1611) with computeLiS0 / LiS1 / LiS2 (:1613-1681)
    FrElement getMontgomeryBatchedInverse()
    {
        const FrElement xi = challenges["xi"], xiw = challenges["xiw"], y = challenges["y"];
        FrElement xiN = xi;
        for (uint32_t i = 0; i < zkeyPower; i++) xiN = FrH::square(xiN);
        toInverse["zh"] = FrH::sub(xiN, FrH::one());
        const std::vector<FrElement>&h0 = roots["S0h0"], &h1 = roots["S1h1"], &h2 = roots["S2h2"], &h3 = roots["S2h3"];
        FrElement den1 = FrH::set(8);
        for (int i = 0; i < 6; i++) den1 = FrH::mul(den1, h0[0]);
        for (int j = 0; j < 8; j++) toInverse["LiS0_" + std::to_string(j + 1) + " "] = FrH::mul(FrH::mul(den1, h0[(7 * j) % 8]), FrH::sub(y, h0[j]));
        den1 = FrH::mul(FrH::set(4), FrH::mul(h1[0], h1[0]));
        for (int j = 0; j < 4; j++) toInverse["LiS1_" + std::to_string(j + 1) + " "] = FrH::mul(FrH::mul(den1, h1[(3 * j) % 4]), FrH::sub(y, h1[j]));
        den1 = FrH::mul(FrH::mul(FrH::set(3), h2[0]), FrH::sub(xi, xiw));
        for (int j = 0; j < 3; j++) toInverse["LiS2_" + std::to_string(j + 1) + " "] = FrH::mul(FrH::mul(den1, h2[2 * j % 3]), FrH::sub(y, h2[j]));
        den1 = FrH::mul(FrH::mul(FrH::set(3), h3[0]), FrH::sub(xiw, xi));
        for (int j = 0; j < 3; j++) toInverse["LiS2_" + std::to_string(j + 1 + 3) + " "] = FrH::mul(FrH::mul(den1, h3[2 * j % 3]), FrH::sub(y, h3[j]));
        const uint32_t size = std::max<uint32_t>(1, zkey->nPublic);
        FrElement w = FrH::one();
        for (uint32_t i = 0; i < size; i++) {
            toInverse["Li_" + std::to_string(i + 1)] = FrH::mul(FrH::set(zkey->domainSize), FrH::sub(xi, w));
            w = FrH::mul(w, root(zkeyPower, 1));
        }
        FrElement acc = FrH::one();
        for (const auto& kv : toInverse) acc = FrH::mul(acc, kv.second);
        return FrH::inv(acc);
    }
};

}  // namespace Fflonk
