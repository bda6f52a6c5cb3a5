// xlp_host_bn128.hpp -- host-side C++ mirror of the BN128 classes StarkRecursiveF uses (src/starkpil/starkRecursiveF/
// starkRecursiveF.cpp:150-175), over the C ABI of libxlp_gl.so.  Same class / member names and argument meaning as the reference:
//
//   RawFr (subset)        src/ffiasm/fr.hpp: Element {v[4]} in Montgomery form, field.zero/one/toMontgomery/fromMontgomery/toString/eq
//   Poseidon_opt          src/poseidon_opt/poseidon_opt.hpp:11-31              -> xlp_poseidon_bn128
//   MerkleTreeBN128       src/starkpil/merkleTree/merkleTreeBN128.hpp:13-45    -> xlp_[dev_]merkle_bn128
//   TranscriptBN128       src/starkpil/transcript/transcriptBN128.hpp:14-38
//
// The scalar Montgomery helpers below are host plumbing for single elements (transcript inputs / outputs, printing); every
// permutation runs on the GPU.
#pragma once
#include "xlp_host.hpp"

class RawFr
{
public:
    struct Element { uint64_t v[4]; };
    typedef unsigned __int128 u128;
    static constexpr uint64_t Q[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    static constexpr uint64_t R2[4] = {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL};
    static constexpr uint64_t NP = 0xc2e1f593efffffffULL;     // -q^-1 mod 2^64

    // a * b / 2^256 mod q (CIOS, 64-bit limbs); b < q, a < 2^256
    static Element mont(const Element& a, const Element& b)
    {
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; i++) {
            u128 c = 0;
            for (int j = 0; j < 4; j++) { const u128 p = (u128)a.v[j] * b.v[i] + t[j] + (uint64_t)c; t[j] = (uint64_t)p; c = p >> 64; }
            u128 s = (u128)t[4] + (uint64_t)c; t[4] = (uint64_t)s; t[5] = (uint64_t)(s >> 64);
            const uint64_t m = t[0] * NP;
            u128 p = (u128)m * Q[0] + t[0]; c = p >> 64;
            for (int j = 1; j < 4; j++) { p = (u128)m * Q[j] + t[j] + (uint64_t)c; t[j - 1] = (uint64_t)p; c = p >> 64; }
            s = (u128)t[4] + (uint64_t)c; t[3] = (uint64_t)s; t[4] = t[5] + (uint64_t)(s >> 64);
        }
        Element r = {{t[0], t[1], t[2], t[3]}};
        bool ge = t[4] != 0;
        if (!ge) { ge = true; for (int i = 3; i >= 0; i--) if (r.v[i] != Q[i]) { ge = r.v[i] > Q[i]; break; } }
        if (ge) { u128 bw = 0; for (int i = 0; i < 4; i++) { const u128 d = (u128)r.v[i] - Q[i] - (uint64_t)bw; r.v[i] = (uint64_t)d; bw = (d >> 64) & 1; } }
        return r;
    }
    Element zero() const { return {{0, 0, 0, 0}}; }
    Element one() const { Element raw = {{1, 0, 0, 0}}; Element r2 = {{R2[0], R2[1], R2[2], R2[3]}}; return mont(raw, r2); }
    void toMontgomery(Element& r, const Element& a) const { Element r2 = {{R2[0], R2[1], R2[2], R2[3]}}; r = mont(a, r2); }
    void fromMontgomery(Element& r, const Element& a) const { Element o = {{1, 0, 0, 0}}; r = mont(a, o); }
    bool eq(const Element& a, const Element& b) const { return std::memcmp(a.v, b.v, sizeof a.v) == 0; }
    std::string toString(const Element& a, uint32_t radix = 10) const
    {
        Element x; fromMontgomery(x, a);
        if (radix != 10) xlp_exit_process("RawFr::toString: radix 10 only");
        std::string s;
        uint64_t w[4] = {x.v[0], x.v[1], x.v[2], x.v[3]};
        while (w[0] | w[1] | w[2] | w[3]) {
            u128 rem = 0;
            for (int i = 3; i >= 0; i--) { const u128 cur = (rem << 64) | w[i]; w[i] = (uint64_t)(cur / 10); rem = cur % 10; }
            s.insert(s.begin(), (char)('0' + (int)rem));
        }
        return s.empty() ? "0" : s;
    }
    static RawFr field;
};
inline RawFr RawFr::field;

class Poseidon_opt
{
public:
    typedef RawFr::Element FrElement;
    void hash(std::vector<FrElement>& state)
    {
        assert(state.size() >= 2 && state.size() < 18);
        XLP_CHECK(xlp_poseidon_bn128(&state[0].v[0], (uint32_t)state.size(), 1));
    }
    void hash(std::vector<FrElement>& state, FrElement* result) { hash(state); *result = state[0]; }
};

#define MT_BN128_ARITY 16
#define GOLDILOCKS_ELEMENTS 3

class MerkleTreeBN128
{
public:
    RawFr::Element* nodes = NULL;
    Goldilocks::Element* source = NULL;
    uint64_t numNodes = 0, height = 0, width = 0, source_width = 0;
    bool intialized = false, isSourceAllocated = false, isNodesAllocated = false;
    // GPU-resident form (like MerkleTreeGL's): source and nodes are device buffers, nothing but roots and openings crosses PCIe
    uint64_t* d_source = NULL;
    uint64_t* d_nodes = NULL;
    bool ownsDevNodes = false;
    bool emptyTreeDone = false;      // source_width == 0: the tree of all-zero leaves depends on the height only; built once per object

    MerkleTreeBN128() {}
    MerkleTreeBN128(uint64_t _height, uint64_t _width, Goldilocks::Element* _source) : source(_source), height(_height), source_width(_width)
    {
        if (source == NULL) { source = (Goldilocks::Element*)calloc((height * _width) != 0 ? height * _width : 1, sizeof(Goldilocks::Element)); isSourceAllocated = true; }
        width = _width > GOLDILOCKS_ELEMENTS + 1 ? (_width + GOLDILOCKS_ELEMENTS - 1) / GOLDILOCKS_ELEMENTS : 0;
        numNodes = getNumNodes(height);
        nodes = (RawFr::Element*)calloc(numNodes, sizeof(RawFr::Element));
        isNodesAllocated = true;
        intialized = true;
    }
    // device-resident: _d_source lives in HBM (may be set later through d_source)
    MerkleTreeBN128(uint64_t _height, uint64_t _width, uint64_t* _d_source, bool /*device*/) : height(_height), source_width(_width), d_source(_d_source)
    {
        width = _width > GOLDILOCKS_ELEMENTS + 1 ? (_width + GOLDILOCKS_ELEMENTS - 1) / GOLDILOCKS_ELEMENTS : 0;
        numNodes = getNumNodes(height);
        void* p = NULL;
        XLP_CHECK(xlp_dev_alloc(&p, numNodes * sizeof(RawFr::Element)));
        d_nodes = (uint64_t*)p;
        ownsDevNodes = true;
        intialized = true;
    }
    // file-backed layout [source_width, height, source, nodes] (MerkleTreeBN128(void*), merkleTreeBN128.cpp:42-51)
    MerkleTreeBN128(void* _source)
    {
        Goldilocks::Element* tree = (Goldilocks::Element*)_source;
        source_width = Goldilocks::toU64(tree[0]);
        height = Goldilocks::toU64(tree[1]);
        source = &tree[2];
        numNodes = getNumNodes(height);
        nodes = (RawFr::Element*)&source[source_width * height];
    }
    ~MerkleTreeBN128()
    {
        if (isNodesAllocated) free(nodes);
        if (isSourceAllocated) free(source);
        if (ownsDevNodes) xlp_dev_free(d_nodes);
    }
    MerkleTreeBN128(const MerkleTreeBN128&) = delete;
    MerkleTreeBN128& operator=(const MerkleTreeBN128&) = delete;

    static uint64_t getNumNodes(uint64_t n) { return xlp_merkle_bn128_num_nodes(n); }
    static uint64_t getMerkleProofLength(uint64_t n) { uint64_t l = 0, c = 1; while (c < n) { c *= 16; l++; } return l; }   // ceil(log16 n)
    static uint64_t getMerkleProofSize(uint64_t n) { return getMerkleProofLength(n) * 16 * sizeof(RawFr::Element); }
    RawFr::Element* address() { return nodes; }
    void initialize(Goldilocks::Element* _source) { std::memcpy(source, _source, height * source_width * sizeof(Goldilocks::Element)); merkelize(); intialized = true; }
    void merkelize()
    {
        if (d_nodes) { XLP_CHECK(xlp_dev_merkle_bn128(d_nodes, d_source, source_width, height)); return; }
        XLP_CHECK(xlp_merkle_bn128(&nodes[0].v[0], xlp_raw(source), source_width, height));
    }
    void getRoot(RawFr::Element* root)
    {
        if (d_nodes) { XLP_CHECK(xlp_copy_d2h(root, d_nodes + (numNodes - 1) * 4, sizeof(RawFr::Element))); return; }
        std::memcpy(root, &nodes[numNodes - 1], sizeof(RawFr::Element));
    }
    Goldilocks::Element getElement(uint64_t idx, uint64_t subIdx)
    {
        if (d_nodes) { Goldilocks::Element e; XLP_CHECK(xlp_copy_d2h(&e, d_source + source_width * idx + subIdx, sizeof e)); return e; }
        return source[source_width * idx + subIdx];
    }
    // res = row (source_width Goldilocks words) || per level the 16 nodes of idx's group (merkleTreeBN128.cpp:210-260)
    void getGroupProof(void* res, uint64_t idx)
    {
        assert(idx < height);
        if (d_nodes) { if (source_width) XLP_CHECK(xlp_copy_d2h(res, d_source + source_width * idx, source_width * sizeof(Goldilocks::Element))); }
        else std::memcpy(res, &source[source_width * idx], source_width * sizeof(Goldilocks::Element));
        RawFr::Element* mp = (RawFr::Element*)((uint8_t*)res + source_width * sizeof(Goldilocks::Element));
        merkle_genMerkleProof(mp, idx, 0, height);
    }
    // getGroupProof for many indices: proofs[q] = row || groups (source_width * 8 + getMerkleProofSize(height) bytes each); one kernel + one
    // copy when the tree lives on the GPU
    void getGroupProofs(std::vector<uint8_t>& proofs, const std::vector<uint64_t>& idxs)
    {
        const uint64_t per = source_width * sizeof(Goldilocks::Element) + getMerkleProofSize(height);
        proofs.assign(idxs.size() * per + 8, 0);
        if (d_nodes) { XLP_CHECK(xlp_dev_merkle_bn128_proofs((uint64_t*)proofs.data(), d_source, d_nodes, source_width, height, idxs.data(), idxs.size())); return; }
        for (size_t q = 0; q < idxs.size(); q++) getGroupProof(&proofs[q * per], idxs[q]);
    }
    void merkle_genMerkleProof(RawFr::Element* proof, uint64_t idx, uint64_t offset, uint64_t n)
    {
        while (n > 1) {
            const uint64_t si = idx & 0xFFFFFFFFFFFFFFF0ULL;
            if (d_nodes) XLP_CHECK(xlp_copy_d2h(proof, d_nodes + (offset + si) * 4, 16 * sizeof(RawFr::Element)));
            else std::memcpy(proof, &nodes[offset + si], 16 * sizeof(RawFr::Element));
            const uint64_t nextN = (n - 1) / 16 + 1;
            proof += 16; offset += nextN * 16; idx >>= 4; n = nextN;
        }
    }
};

// TranscriptBN128 (src/starkpil/transcript/transcriptBN128.hpp:12-38, .cpp:5-141) as a sponge over fixed arrays with cursors: capacity =
// one Fr element, rate = 16.  Observable behaviour kept, quirks included: absorbing an element discards the unread OUTPUT ELEMENTS but not
// the 64-bit limbs already cut from one (the reference clears `out`, not `out3`, in _add1); a squeeze on a fresh transcript returns the
// initial zero element without hashing (its `out` starts as one zero); a squeeze with a partial block hashes the block zero-padded.
class TranscriptBN128
{
    static const unsigned kRate = 16, kWidth = kRate + 1;
    RawFr::Element capacity_;                  // the reference's state[0]
    RawFr::Element block_[kRate];               // the reference's `pending`
    unsigned filled_ = 0;
    RawFr::Element squeezed_[kWidth];           // the reference's `out`: elements [head_, head_ + left_) are unread
    unsigned head_ = 0, left_ = 1;
    uint64_t limbs_[3];                        // the reference's `out3`
    unsigned limbHead_ = 0, limbsLeft_ = 0;

    // out <- Poseidon(capacity | block, zero-padded), capacity <- out[0]; 17 fresh output elements, no limbs
    void duplex()
    {
        std::vector<RawFr::Element> st(kWidth, RawFr::field.zero());
        st[0] = capacity_;
        for (unsigned i = 0; i < filled_; i++) st[1 + i] = block_[i];
        Poseidon_opt().hash(st);
        for (unsigned i = 0; i < kWidth; i++) squeezed_[i] = st[i];
        capacity_ = squeezed_[0];
        head_ = 0; left_ = kWidth;
        limbsLeft_ = 0;
        filled_ = 0;
    }
    void absorb(const RawFr::Element& e)
    {
        left_ = 0;
        block_[filled_++] = e;
        if (filled_ == kRate) duplex();
    }
    RawFr::Element nextElement()
    {
        if (left_ == 0) duplex();
        left_--;
        return squeezed_[head_++];
    }

public:
    unsigned typeSize = 2;
    TranscriptBN128()
    {
        capacity_ = RawFr::field.zero();
        squeezed_[0] = RawFr::field.zero();
    }
    // the reference's public vectors, as copies (tests look at them; the prover does not)
    std::vector<RawFr::Element> state() const { return std::vector<RawFr::Element>(1, capacity_); }
    std::vector<RawFr::Element> pending() const { return std::vector<RawFr::Element>(block_, block_ + filled_); }
    std::vector<RawFr::Element> out() const { return std::vector<RawFr::Element>(squeezed_ + head_, squeezed_ + head_ + left_); }
    std::vector<uint64_t> out3() const { return std::vector<uint64_t>(limbs_ + limbHead_, limbs_ + limbHead_ + limbsLeft_); }

    void put(Goldilocks::Element* input, uint64_t size)
    {
        for (uint64_t i = 0; i < size; i++) {
            RawFr::Element e = RawFr::field.zero();
            e.v[0] = Goldilocks::toU64(input[i]);
            RawFr::field.toMontgomery(e, e);
            absorb(e);
        }
    }
    void put(RawFr::Element* input, uint64_t size) { for (uint64_t i = 0; i < size; i++) absorb(input[i]); }
    RawFr::Element getFields253() { return nextElement(); }
    // 64-bit limbs: three per squeezed element (its low 192 bits, out of Montgomery form)
    uint64_t getFields1()
    {
        if (limbsLeft_ == 0) {
            RawFr::Element plain;
            RawFr::field.fromMontgomery(plain, nextElement());
            for (int i = 0; i < 3; i++) limbs_[i] = plain.v[i];
            limbHead_ = 0; limbsLeft_ = 3;
        }
        limbsLeft_--;
        return limbs_[limbHead_++];
    }
    void getField(uint64_t* output) { for (int i = 0; i < 3; i++) output[i] = getFields1(); }
    // query indices: n numbers of nBits bits, read LSB-first from the 253 low bits of as many squeezed elements as needed
    void getPermutations(uint64_t* res, uint64_t n, uint64_t nBits)
    {
        RawFr::Element cur;
        unsigned have = 0;                      // unread bits of `cur`
        for (uint64_t i = 0; i < n; i++) {
            uint64_t v = 0;
            for (uint64_t j = 0; j < nBits; j++) {
                if (have == 0) { RawFr::field.fromMontgomery(cur, nextElement()); have = 253; }
                const unsigned bit = 253 - have;
                v |= ((cur.v[bit >> 6] >> (bit & 63)) & 1ULL) << j;
                have--;
            }
            res[i] = v;
        }
    }
};
