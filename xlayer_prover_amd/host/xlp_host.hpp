// xlp_host.hpp -- host-side C++ mirror of the interface that src/starkpil calls
// on the (absent) goldilocks submodule and of the starkpil helper classes on the
// Starks::genProof path, implemented over the C ABI of libxlp_gl.so
// (include/xlp_gl.h).  Same class / method names, argument meaning and error
// behaviour as the reference, so starks.cpp / friProve.cpp call sites compile
// against it unchanged:
//
//   Goldilocks, Goldilocks3            field helpers used by the drivers (host scalars only)
//   NTT_Goldilocks                     starks.hpp:81-82, starks.cpp:53,134,215,262,285,326-327
//   PoseidonGoldilocks                 transcript.cpp:23,46; merkleTreeGL.cpp:39-43
//   MerklehashGoldilocks               starks.cpp:684,694
//   MerkleTreeGL                       merkleTree/merkleTreeGL.hpp:9-79, .cpp:12-44
//   Transcript                         transcript/transcript.hpp:13-37, .cpp:4-87
//   Polinomial (view subset)           polinomial.hpp:11-140
//   MerkleProof/ProofTree/Fri/Proofs/FRIProof     fri/friProof.hpp:11-235
//   FRIProve::prove                    fri/friProve.cpp:5-190
//   proof2zkinStark (as a JSON string) fri/proof2zkinStark.cpp:8-82
//   joinzkin (JSON values in, text out) fri/proof2zkinStark.cpp:84-175
//
// All field arithmetic on data buffers runs on the GPU; the scalar helpers in
// class Goldilocks exist only because the reference's drivers use them for a
// handful of per-proof constants (shift powers, challenges).
// Errors: the reference's library functions return void and end the process via
// exitProcess() (src/utils/exit_process.cpp:7); a non-zero C-ABI status does the
// same here (xlp_exit_process, overridable for tests).
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <sstream>
#include <string>
#include <vector>
#include "../../include/xlp_gl.h"

#define FIELD_EXTENSION 3
#define HASH_SIZE 4
#define SPONGE_WIDTH 12
#define RATE 8
#define CAPACITY 4

inline void xlp_exit_process(const char* what)
{
    fprintf(stderr, "xlp_gl error in %s: %s\n", what, xlp_last_error());
    xlp_comm_abort(what);                          // one rank of a row-sharded proof giving up ends the others (no-op otherwise)
    exit(-1);
}
#define XLP_CHECK(call)                         \
    do {                                        \
        if ((call) != 0) xlp_exit_process(#call); \
    } while (0)

// ------------------------------------------------------------------ Goldilocks
class Goldilocks
{
public:
    typedef struct { uint64_t fe; } Element;
    static constexpr uint64_t P = 0xFFFFFFFF00000001ULL;
    static inline Element zero() { return {0}; }
    static inline Element one() { return {1}; }
    static inline Element fromU64(uint64_t x) { return {x % P}; }
    static inline uint64_t toU64(const Element& a) { return a.fe >= P ? a.fe - P : a.fe; }
    static inline std::string toString(const Element& a) { return std::to_string(toU64(a)); }
    static inline Element add(const Element& a, const Element& b) { return {(uint64_t)(((unsigned __int128)toU64(a) + toU64(b)) % P)}; }
    static inline Element sub(const Element& a, const Element& b) { return {(uint64_t)(((unsigned __int128)toU64(a) + P - toU64(b)) % P)}; }
    static inline Element mul(const Element& a, const Element& b) { return {(uint64_t)(((unsigned __int128)a.fe * b.fe) % P)}; }
    static inline void mul(Element& r, const Element& a, const Element& b) { r = mul(a, b); }
    static inline Element square(const Element& a) { return mul(a, a); }
    static inline Element exp(Element base, uint64_t e)
    {
        Element r = one();
        while (e) { if (e & 1) r = mul(r, base); base = mul(base, base); e >>= 1; }
        return r;
    }
    static inline Element inv(const Element& a) { return exp(a, P - 2); }
    static inline bool isZero(const Element& a) { return toU64(a) == 0; }
    static inline bool equal(const Element& a, const Element& b) { return toU64(a) == toU64(b); }
    // primitive 2^n-th root of unity; W[32] pinned by the reference's golden proofs
    static inline Element w(uint64_t n)
    {
        Element r = {7277203076849721926ULL};
        for (uint64_t i = 32; i > n; i--) r = square(r);
        return r;
    }
    static inline Element shift() { return {7}; }
};
inline Goldilocks::Element operator+(const Goldilocks::Element& a, const Goldilocks::Element& b) { return Goldilocks::add(a, b); }
inline Goldilocks::Element operator-(const Goldilocks::Element& a, const Goldilocks::Element& b) { return Goldilocks::sub(a, b); }
inline Goldilocks::Element operator*(const Goldilocks::Element& a, const Goldilocks::Element& b) { return Goldilocks::mul(a, b); }
static_assert(sizeof(Goldilocks::Element) == sizeof(uint64_t), "Element must be one 64-bit word");

class Goldilocks3
{
public:
    typedef Goldilocks::Element Element[FIELD_EXTENSION];
    // Fp[x]/(x^3 - x - 1), polinomial.hpp:195-205
    static inline void mul(Element& r, const Element& a, const Element& b)
    {
        typedef Goldilocks G;
        G::Element A = (a[0] + a[1]) * (b[0] + b[1]), B = (a[0] + a[2]) * (b[0] + b[2]), C = (a[1] + a[2]) * (b[1] + b[2]);
        G::Element D = a[0] * b[0], E = a[1] * b[1], F = a[2] * b[2], Gg = D - E;
        G::Element r0 = (C + Gg) - F, r1 = ((((A + C) - E) - E) - D), r2 = B - Gg;
        r[0] = r0; r[1] = r1; r[2] = r2;
    }
    static inline void mul(Element& r, const Element& a, const Goldilocks::Element& b)
    {
        r[0] = a[0] * b; r[1] = a[1] * b; r[2] = a[2] * b;
    }
    static inline void one(Element& r) { r[0] = Goldilocks::one(); r[1] = Goldilocks::zero(); r[2] = Goldilocks::zero(); }
};

static inline uint64_t* xlp_raw(Goldilocks::Element* p) { return reinterpret_cast<uint64_t*>(p); }
static inline const uint64_t* xlp_raw(const Goldilocks::Element* p) { return reinterpret_cast<const uint64_t*>(p); }

// ------------------------------------------------------------- NTT_Goldilocks
class NTT_Goldilocks
{
    uint64_t s_;
public:
    // (size, nThreads, extension): table sizes / thread counts are the library's business here
    NTT_Goldilocks(uint64_t maxDomainSize, uint32_t /*nThreads*/ = 0, int /*extension*/ = 1) : s_(maxDomainSize) { XLP_CHECK(xlp_init(0)); }
    void NTT(Goldilocks::Element* dst, Goldilocks::Element* src, uint64_t size, uint64_t ncols = 1, Goldilocks::Element* /*buffer*/ = NULL,
             uint64_t /*nphase*/ = 0, uint64_t /*nblock*/ = 0, bool inverse = false, bool /*extend*/ = false)
    {
        XLP_CHECK(xlp_ntt(xlp_raw(dst), xlp_raw(src), size, ncols, inverse ? 1 : 0));
    }
    void INTT(Goldilocks::Element* dst, Goldilocks::Element* src, uint64_t size, uint64_t ncols = 1, Goldilocks::Element* buffer = NULL,
              uint64_t nphase = 0, uint64_t nblock = 0, bool extend = false)
    {
        NTT(dst, src, size, ncols, buffer, nphase, nblock, true, extend);
    }
    void extendPol(Goldilocks::Element* output, Goldilocks::Element* input, uint64_t N_Extended, uint64_t N, uint64_t ncols,
                   Goldilocks::Element* buffer = NULL, uint64_t /*nphase*/ = 0, uint64_t /*nblock*/ = 0)
    {
        XLP_CHECK(xlp_extend(xlp_raw(output), xlp_raw(input), N_Extended, N, ncols, xlp_raw(buffer)));
    }
};

// --------------------------------------------------------- PoseidonGoldilocks
class PoseidonGoldilocks
{
public:
    static void hash_full_result(Goldilocks::Element* state, const Goldilocks::Element* input)
    {
        XLP_CHECK(xlp_perm(xlp_raw(state), xlp_raw(input)));
    }
    // nblocks dependent hash_full_result calls on input[8b .. 8b + 8) | capacity (capacity: `state`, then words 0..3 of the previous result): what
    // Transcript::put does element by element (transcript.cpp:12-29), as ONE call
    static void hash_chain(Goldilocks::Element* out12, const Goldilocks::Element* state4, const Goldilocks::Element* input, uint64_t nblocks)
    {
        XLP_CHECK(xlp_perm_chain(xlp_raw(out12), xlp_raw(state4), xlp_raw(input), nblocks));
    }
    static void hash(Goldilocks::Element (&state)[CAPACITY], const Goldilocks::Element (&input)[SPONGE_WIDTH])
    {
        Goldilocks::Element full[SPONGE_WIDTH];
        hash_full_result(full, input);
        std::memcpy(state, full, CAPACITY * sizeof(Goldilocks::Element));
    }
    static void linear_hash(Goldilocks::Element* output, Goldilocks::Element* input, uint64_t size)
    {
        XLP_CHECK(xlp_linear_hash(xlp_raw(output), xlp_raw(input), size, 1));
    }
    static void merkletree(Goldilocks::Element* tree, Goldilocks::Element* input, uint64_t num_cols, uint64_t num_rows, int /*nThreads*/ = 0, uint64_t /*dim*/ = 1)
    {
        XLP_CHECK(xlp_merkle(xlp_raw(tree), xlp_raw(input), num_cols, num_rows));
    }
    static void merkletree_avx(Goldilocks::Element* tree, Goldilocks::Element* input, uint64_t num_cols, uint64_t num_rows, int n = 0, uint64_t dim = 1)
    {
        merkletree(tree, input, num_cols, num_rows, n, dim);
    }
    static void merkletree_avx512(Goldilocks::Element* tree, Goldilocks::Element* input, uint64_t num_cols, uint64_t num_rows, int n = 0, uint64_t dim = 1)
    {
        merkletree(tree, input, num_cols, num_rows, n, dim);
    }
};

class MerklehashGoldilocks
{
public:
    static inline uint64_t getTreeNumElements(uint64_t numRows) { return numRows * HASH_SIZE + (numRows - 1) * HASH_SIZE; }
    static inline void root(Goldilocks::Element* root, Goldilocks::Element* tree, uint64_t numElementsTree)
    {
        std::memcpy(root, &tree[numElementsTree - HASH_SIZE], HASH_SIZE * sizeof(Goldilocks::Element));
    }
};

// --------------------------------------------------------------- MerkleTreeGL
#define MERKLEHASHGL_ARITY 2
class MerkleTreeGL
{
public:
    uint64_t height;
    uint64_t width;
    Goldilocks::Element* source;
    Goldilocks::Element* nodes;
    bool isSourceAllocated = false;
    bool isNodesAllocated = false;
    // GPU-resident form (used by the GPU-resident Starks below): source / nodes stay in HBM, merkelize() runs there and
    // getRoot / getGroupProof copy only the root or one row + its sibling path to the host
    uint64_t* d_source = NULL;
    uint64_t* d_nodes = NULL;
    bool ownsDevNodes = false;
    // Row-sharded form (host/starks.hpp, sharded mode): `height` stays the GLOBAL number of rows; this rank holds the
    // sub-tree over ITS height / shardWorld rows (d_source, d_nodes) -- its root is node (level log2(height / world), rank) of the
    // reference's tree -- and every rank holds the cap: the levels over the `world` sub-roots (32 bytes per rank all-gathered).
    // getRoot / getGroupProofs answer for the GLOBAL tree on every rank (an opening is made by the rank that owns the row and shared).
    int shardRank = 0, shardWorld = 0;                         // shardWorld == 0: an ordinary tree
    // Column-major source (GPU-resident form; host/starks.hpp with a "colMajor" starkinfo): d_source = `width` columns of `height` words
    // (cmStride apart), coset-major with 2^cmExtBits cosets; leaves, nodes and openings are the same as over the row-major section
    bool colMajor = false;
    uint32_t cmExtBits = 0;
    uint64_t cmStride = 0, cmCos = 0;                          // words between the columns / between the cosets of a column (0: rows >> extBits)
    void setColMajor(uint32_t extBits, uint64_t cstride, uint64_t cosetStride = 0) { colMajor = true; cmExtBits = extBits; cmStride = cstride; cmCos = cosetStride; }
    std::vector<std::vector<uint64_t>> capLevels;              // [sub-roots (world x 4)], ..., [root (4)]
    MerkleTreeGL(uint64_t _height, uint64_t _width, uint64_t* dSource, bool /*deviceTag*/, int world = 0, int rank = 0)
        : height(_height), width(_width), source(NULL), nodes(NULL), d_source(dSource), shardRank(rank), shardWorld(world)
    {
        if (world && (_height % (uint64_t)world || rank < 0 || rank >= world)) xlp_exit_process("MerkleTreeGL: rows do not divide over the ranks");
        const uint64_t words = MerklehashGoldilocks::getTreeNumElements(localHeight());
        XLP_CHECK(xlp_dev_alloc((void**)&d_nodes, (localHeight() ? words : 1) * sizeof(uint64_t)));
        ownsDevNodes = true;
    }
    uint64_t localHeight() const { return shardWorld ? height / (uint64_t)shardWorld : height; }
    // the cap over the sub-roots: one all-gather of 4 words per rank, then world - 1 node hashes on every rank
    void assembleCap()
    {
        const uint64_t lh = localHeight(), words = MerklehashGoldilocks::getTreeNumElements(lh);
        uint64_t sub[HASH_SIZE];
        XLP_CHECK(xlp_copy_d2h(sub, d_nodes + words - HASH_SIZE, sizeof sub));
        std::vector<uint64_t> lvl((size_t)shardWorld * HASH_SIZE);
        XLP_CHECK(xlp_comm_all_gather_host(lvl.data(), sub, HASH_SIZE));
        capLevels.clear();
        capLevels.push_back(lvl);
        while (capLevels.back().size() > HASH_SIZE) {
            const std::vector<uint64_t>& cur = capLevels.back();
            std::vector<uint64_t> nxt(cur.size() / 2);
            for (size_t k = 0; k < nxt.size() / HASH_SIZE; k++) {
                uint64_t in[12], out[12];
                for (int i = 0; i < 8; i++) in[i] = cur[8 * k + i];
                in[8] = in[9] = in[10] = in[11] = 0;
                XLP_CHECK(xlp_perm(out, in));
                for (int i = 0; i < HASH_SIZE; i++) nxt[HASH_SIZE * k + i] = out[i];
            }
            capLevels.push_back(nxt);
        }
    }
    // row-sharded commits absorb the section band by band behind the exchanges (host/starks.hpp commitSharded): the leaf sponge of my rows
    // over the columns [col0, col0 + ncols) of d_source, then -- after the last band -- the levels of my sub-tree and the cap
    void absorbBand(uint64_t col0, uint64_t ncols, bool first, bool last)
    {
        if (!d_nodes || !shardWorld) xlp_exit_process("MerkleTreeGL::absorbBand: a sharded device tree only");
        if (colMajor) XLP_CHECK(xlp_dev_linear_hash_absorb_cm(d_nodes, d_source, cmStride, cmCos, col0, ncols, localHeight(), cmExtBits, first ? 1 : 0, last ? 1 : 0));
        else XLP_CHECK(xlp_dev_linear_hash_absorb(d_nodes, d_source, width, col0, ncols, localHeight(), first ? 1 : 0, last ? 1 : 0));
    }
    void finishFromLeaves() { XLP_CHECK(xlp_dev_merkle_levels(d_nodes, localHeight())); assembleCap(); }
    MerkleTreeGL() : height(0), width(0), source(NULL), nodes(NULL) {}
    // file-backed layout [width, height, source(height*width), nodes] (build_const_tree.cpp:566-603)
    MerkleTreeGL(Goldilocks::Element* tree)
    {
        width = Goldilocks::toU64(tree[0]);
        height = Goldilocks::toU64(tree[1]);
        source = &tree[2];
        nodes = &tree[2 + height * width];
    }
    MerkleTreeGL(uint64_t _height, uint64_t _width, Goldilocks::Element* _source) : height(_height), width(_width), source(_source)
    {
        if (source == NULL) {
            source = (Goldilocks::Element*)calloc((height * width) != 0 ? height * width : 1, sizeof(Goldilocks::Element));
            isSourceAllocated = true;
        }
        nodes = (Goldilocks::Element*)calloc(getTreeNumElements(), sizeof(Goldilocks::Element));
        isNodesAllocated = true;
    }
    ~MerkleTreeGL()
    {
        if (isSourceAllocated) free(source);
        if (isNodesAllocated) free(nodes);
        if (ownsDevNodes) xlp_dev_free(d_nodes);
    }
    MerkleTreeGL(const MerkleTreeGL&) = delete;
    MerkleTreeGL& operator=(const MerkleTreeGL&) = delete;
    void copySource(Goldilocks::Element* _source) { std::memcpy(source, _source, height * width * sizeof(Goldilocks::Element)); }
    void merkelize()
    {
        if (d_nodes && shardWorld && colMajor) { XLP_CHECK(xlp_dev_merkle_cm(d_nodes, d_source, cmStride, cmCos, width, localHeight(), cmExtBits)); assembleCap(); return; }
        if (d_nodes && shardWorld) { XLP_CHECK(xlp_dev_merkle(d_nodes, d_source, width, localHeight())); assembleCap(); return; }
        if (d_nodes && colMajor) { XLP_CHECK(xlp_dev_merkle_cm(d_nodes, d_source, cmStride, cmCos, width, height, cmExtBits)); return; }
        if (d_nodes) { XLP_CHECK(xlp_dev_merkle(d_nodes, d_source, width, height)); return; }
        PoseidonGoldilocks::merkletree_avx(nodes, source, width, height);
    }
    uint64_t getTreeNumElements() { return height * HASH_SIZE + (height - 1) * HASH_SIZE; }
    void getRoot(Goldilocks::Element* root)
    {
        if (shardWorld) {
            if (capLevels.empty()) xlp_exit_process("MerkleTreeGL::getRoot: sharded tree not merkelized");
            std::memcpy(root, capLevels.back().data(), HASH_SIZE * sizeof(Goldilocks::Element));
            return;
        }
        if (d_nodes) { XLP_CHECK(xlp_copy_d2h(root, d_nodes + getTreeNumElements() - HASH_SIZE, HASH_SIZE * sizeof(Goldilocks::Element))); return; }
        std::memcpy(root, &nodes[getTreeNumElements() - HASH_SIZE], HASH_SIZE * sizeof(Goldilocks::Element));
    }
    uint64_t MerkleProofSize()
    {
        uint64_t n = 0;
        while ((1ULL << n) < height) n++;
        return height > 1 ? n : 0;
    }
    // getGroupProof for many indices: proofs[q] = row || siblings (width + MerkleProofSize() * HASH_SIZE elements each);
    // one kernel + one copy when the tree lives on the GPU
    void getGroupProofs(std::vector<Goldilocks::Element>& proofs, const std::vector<uint64_t>& idxs)
    {
        const uint64_t per = width + MerkleProofSize() * HASH_SIZE;
        proofs.assign(idxs.size() * per + 1, Goldilocks::zero());
        if (shardWorld) {
            // the owner of row idx (idx / localHeight) opens it inside its sub-tree and appends the cap siblings; one all-gather shares them
            const uint64_t lh = localHeight();
            uint64_t lbits = 0;
            while ((1ULL << lbits) < lh) lbits++;
            const uint64_t perLocal = width + lbits * HASH_SIZE;
            std::vector<uint64_t> mine, pos;
            for (size_t q = 0; q < idxs.size(); q++) {
                if (idxs[q] >= height) xlp_exit_process("MerkleTreeGL::getGroupProofs: index out of range");
                if (idxs[q] / lh == (uint64_t)shardRank) { mine.push_back(idxs[q] % lh); pos.push_back(q); }
            }
            if (!mine.empty()) {
                std::vector<uint64_t> loc(mine.size() * perLocal);
                if (colMajor) XLP_CHECK(xlp_dev_merkle_proofs_cm(loc.data(), d_source, cmStride, cmCos, d_nodes, width, lh, cmExtBits, mine.data(), mine.size()));
                else XLP_CHECK(xlp_dev_merkle_proofs(loc.data(), d_source, d_nodes, width, lh, mine.data(), mine.size()));
                for (size_t k = 0; k < mine.size(); k++) {
                    uint64_t* o = xlp_raw(&proofs[pos[k] * per]);
                    std::memcpy(o, &loc[k * perLocal], perLocal * sizeof(uint64_t));
                    o += perLocal;
                    uint64_t i = (uint64_t)shardRank;
                    for (size_t l = 0; l + 1 < capLevels.size(); l++) { std::memcpy(o, &capLevels[l][(i ^ 1) * HASH_SIZE], HASH_SIZE * sizeof(uint64_t)); o += HASH_SIZE; i >>= 1; }
                }
            }
            if (shardWorld > 1) {
                const uint64_t words = idxs.size() * per;
                std::vector<uint64_t> all((size_t)shardWorld * words);
                XLP_CHECK(xlp_comm_all_gather_host(all.data(), xlp_raw(proofs.data()), words));
                for (size_t q = 0; q < idxs.size(); q++) std::memcpy(xlp_raw(&proofs[q * per]), &all[(idxs[q] / lh) * words + q * per], per * sizeof(uint64_t));
            }
            return;
        }
        if (d_nodes && colMajor) { XLP_CHECK(xlp_dev_merkle_proofs_cm(xlp_raw(proofs.data()), d_source, cmStride, cmCos, d_nodes, width, height, cmExtBits, idxs.data(), idxs.size())); return; }
        if (d_nodes) { XLP_CHECK(xlp_dev_merkle_proofs(xlp_raw(proofs.data()), d_source, d_nodes, width, height, idxs.data(), idxs.size())); return; }
        for (size_t q = 0; q < idxs.size(); q++) getGroupProof(&proofs[q * per], idxs[q]);
    }
    // proof = row || siblings bottom-up (merkleTreeGL.cpp:12-35)
    void getGroupProof(Goldilocks::Element* proof, uint64_t idx)
    {
        if (shardWorld || (d_nodes && colMajor)) {
            std::vector<Goldilocks::Element> one;
            getGroupProofs(one, std::vector<uint64_t>(1, idx));
            std::memcpy(proof, one.data(), (width + MerkleProofSize() * HASH_SIZE) * sizeof(Goldilocks::Element));
            return;
        }
        if (d_nodes) {
            if (width) XLP_CHECK(xlp_copy_d2h(proof, d_source + idx * width, width * sizeof(Goldilocks::Element)));
            Goldilocks::Element* o = &proof[width];
            uint64_t off_ = 0, cnt_ = height;
            while (cnt_ > 1) {
                XLP_CHECK(xlp_copy_d2h(o, d_nodes + off_ + (idx ^ 1) * HASH_SIZE, HASH_SIZE * sizeof(Goldilocks::Element)));
                o += HASH_SIZE; off_ += cnt_ * HASH_SIZE; cnt_ >>= 1; idx >>= 1;
            }
            return;
        }
        for (uint64_t i = 0; i < width; i++) proof[i] = source[idx * width + i];
        Goldilocks::Element* out = &proof[width];
        uint64_t off = 0, cnt = height;
        while (cnt > 1) {
            std::memcpy(out, &nodes[off + (idx ^ 1) * HASH_SIZE], HASH_SIZE * sizeof(Goldilocks::Element));
            out += HASH_SIZE;
            off += cnt * HASH_SIZE;
            cnt >>= 1;
            idx >>= 1;
        }
    }
};

// ----------------------------------------------------------------- Transcript
#define TRANSCRIPT_STATE_SIZE 4
#define TRANSCRIPT_PENDING_SIZE 8
#define TRANSCRIPT_OUT_SIZE 12
// Transcript (src/starkpil/transcript/transcript.hpp:13-37, .cpp:4-87): a duplex sponge over the Poseidon permutation, rate 8, capacity 4.
// The reference's public members keep their names and meaning (state = the capacity, pending = the rate block being filled, out = the last
// permutation's 12 words of which the last out_cursor are unread); the bodies are this file's own.
class Transcript
{
    // one permutation: out <- P(pending | state); the capacity becomes out[0..4), the block is emptied, 12 words are there to be read
    void duplex()
    {
        Goldilocks::Element in[TRANSCRIPT_OUT_SIZE];
        for (unsigned i = 0; i < TRANSCRIPT_PENDING_SIZE; i++) { in[i] = pending[i]; pending[i] = Goldilocks::zero(); }
        for (unsigned i = 0; i < TRANSCRIPT_STATE_SIZE; i++) in[TRANSCRIPT_PENDING_SIZE + i] = state[i];
        PoseidonGoldilocks::hash_full_result(out, in);
        for (unsigned i = 0; i < TRANSCRIPT_STATE_SIZE; i++) state[i] = out[i];
        pending_cursor = 0;
        out_cursor = TRANSCRIPT_OUT_SIZE;
    }
    // absorbing anything makes what is left of the old output unreadable
    void absorb(const Goldilocks::Element& e)
    {
        out_cursor = 0;
        pending[pending_cursor] = e;
        if (++pending_cursor == TRANSCRIPT_PENDING_SIZE) duplex();
    }
public:
    Goldilocks::Element state[TRANSCRIPT_STATE_SIZE];
    Goldilocks::Element pending[TRANSCRIPT_PENDING_SIZE];
    Goldilocks::Element out[TRANSCRIPT_OUT_SIZE];
    unsigned pending_cursor = 0, out_cursor = 0, state_cursor = 0;
    Transcript()
    {
        for (Goldilocks::Element& e : state) e = Goldilocks::zero();
        for (Goldilocks::Element& e : pending) e = Goldilocks::zero();
        for (Goldilocks::Element& e : out) e = Goldilocks::zero();
    }
    // transcript.cpp:12-29 absorbs element by element, one permutation per 8.  A long message (the evaluations: 3 x 1 973 words at the zkEVM)
    // is cut into head (completes the block in progress) | whole blocks | tail: the whole blocks go through ONE kernel that walks the chained
    // permutations (xlp_perm_chain) -- the same permutations in the same order, the same members afterwards
    void put(Goldilocks::Element* input, uint64_t size)
    {
        const uint64_t head = pending_cursor ? std::min<uint64_t>(size, TRANSCRIPT_PENDING_SIZE - pending_cursor) : 0;
        for (uint64_t i = 0; i < head; i++) absorb(input[i]);
        uint64_t done = head;
        const uint64_t whole = pending_cursor == 0 ? (size - done) / TRANSCRIPT_PENDING_SIZE : 0;
        if (whole >= 2) {
            PoseidonGoldilocks::hash_chain(out, state, input + done, whole);
            for (unsigned i = 0; i < TRANSCRIPT_STATE_SIZE; i++) state[i] = out[i];
            out_cursor = TRANSCRIPT_OUT_SIZE;
            done += whole * TRANSCRIPT_PENDING_SIZE;
        }
        for (; done < size; done++) absorb(input[done]);
    }
    // the unread words are the LAST out_cursor words of out, read front to back; with none left the current block is permuted first
    Goldilocks::Element getFields1()
    {
        if (out_cursor == 0) duplex();
        return out[TRANSCRIPT_OUT_SIZE - out_cursor--];
    }
    void getField(Goldilocks::Element* output) { for (int i = 0; i < 3; i++) output[i] = getFields1(); }
    // query indices (transcript.cpp:57-87): n numbers of nBits bits each, read LSB-first from the low 63 bits of as many squeezed words as needed
    void getPermutations(uint64_t* res, uint64_t n, uint64_t nBits)
    {
        uint64_t word = 0;
        unsigned have = 0;                                       // unread bits of `word`
        for (uint64_t i = 0; i < n; i++) {
            uint64_t v = 0;
            for (uint64_t j = 0; j < nBits; j++) {
                if (have == 0) { word = Goldilocks::toU64(getFields1()); have = 63; }
                v |= (word & 1ULL) << j;
                word >>= 1;
                have--;
            }
            res[i] = v;
        }
    }
};

// ----------------------------------------------------------------- Polinomial
// Strided view (address, degree, dim, offset) -- the subset FRIProve::prove needs.
class Polinomial
{
    Goldilocks::Element* _pAddress = NULL;
    uint64_t _degree = 0, _dim = 0, _offset = 0;
    bool _allocated = false;
public:
    Polinomial() {}
    Polinomial(void* pAddress, uint64_t degree, uint64_t dim, uint64_t offset = 0, std::string = "")
        : _pAddress((Goldilocks::Element*)pAddress), _degree(degree), _dim(dim), _offset(offset ? offset : dim) {}
    Polinomial(uint64_t degree, uint64_t dim, std::string = "") : _degree(degree), _dim(dim), _offset(dim), _allocated(true)
    {
        _pAddress = (Goldilocks::Element*)calloc((degree * dim) != 0 ? degree * dim : 1, sizeof(Goldilocks::Element));
    }
    ~Polinomial() { if (_allocated) free(_pAddress); }
    Polinomial(const Polinomial&) = delete;
    Polinomial& operator=(const Polinomial&) = delete;
    Goldilocks::Element* address() { return _pAddress; }
    uint64_t degree() const { return _degree; }
    uint64_t dim() const { return _dim; }
    uint64_t offset() const { return _offset; }
    Goldilocks::Element* operator[](uint64_t i) { return &_pAddress[i * _offset]; }
    std::string toString(uint64_t n = 0)
    {
        std::ostringstream os;
        uint64_t cnt = n ? n : _degree;
        for (uint64_t i = 0; i < cnt; i++) os << (i ? " " : "") << Goldilocks::toString(_pAddress[i * _offset]);
        return os.str();
    }

    // ---- row-scale helpers (polinomial.hpp:249-720), same static signatures; the strided view is packed to a
    // dense buffer, the work runs on the GPU (xlp_batch_inverse / xlp_calculate_z / xlp_calculate_h1h2)
private:
    std::vector<uint64_t> pack() const
    {
        std::vector<uint64_t> d(_degree * _dim);
        for (uint64_t i = 0; i < _degree; i++)
            for (uint64_t c = 0; c < _dim; c++) d[i * _dim + c] = _pAddress[i * _offset + c].fe;
        return d;
    }
    void unpack(const std::vector<uint64_t>& d)
    {
        for (uint64_t i = 0; i < _degree; i++)
            for (uint64_t c = 0; c < _dim; c++) _pAddress[i * _offset + c].fe = d[i * _dim + c];
    }
public:
    static void batchInverse(Polinomial& res, Polinomial& src)
    {
        if (res.dim() != src.dim() || res.degree() != src.degree()) xlp_exit_process("Polinomial::batchInverse: shape mismatch");
        std::vector<uint64_t> a = src.pack();
        XLP_CHECK(xlp_batch_inverse(a.data(), a.data(), src.degree(), (uint32_t)src.dim()));
        res.unpack(a);
    }
    static void batchInverseParallel(Polinomial& res, Polinomial& src) { batchInverse(res, src); }
    static void calculateZ(Polinomial& z, Polinomial& num, Polinomial& den)
    {
        if (z.dim() != 3 || num.dim() != 3 || den.dim() != 3 || z.degree() != num.degree() || z.degree() != den.degree())
            xlp_exit_process("Polinomial::calculateZ: three ext-3 polynomials of one degree expected");
        std::vector<uint64_t> a = num.pack(), b = den.pack(), c(a.size());
        XLP_CHECK(xlp_calculate_z(c.data(), a.data(), b.data(), z.degree()));
        z.unpack(c);
    }
    static void calculateH1H2(Polinomial& h1, Polinomial& h2, Polinomial& fPol, Polinomial& tPol)
    {
        if (h1.dim() != fPol.dim() || h2.dim() != fPol.dim() || tPol.dim() != fPol.dim() || h1.degree() != fPol.degree() ||
            h2.degree() != fPol.degree() || tPol.degree() != fPol.degree())
            xlp_exit_process("Polinomial::calculateH1H2: shape mismatch");
        std::vector<uint64_t> f = fPol.pack(), t = tPol.pack(), a(f.size()), b(f.size());
        XLP_CHECK(xlp_calculate_h1h2(a.data(), b.data(), f.data(), t.data(), fPol.degree(), (uint32_t)fPol.dim()));   // "Number not included" -> exitProcess
        h1.unpack(a); h2.unpack(b);
    }
    // the reference's scratch-buffer variants (starks.cpp:116,123): the buffer arguments are not needed here
    static void calculateH1H2_opt1(Polinomial& h1, Polinomial& h2, Polinomial& fPol, Polinomial& tPol, uint64_t, uint64_t*, uint64_t, uint64_t) { calculateH1H2(h1, h2, fPol, tPol); }
    static void calculateH1H2_opt3(Polinomial& h1, Polinomial& h2, Polinomial& fPol, Polinomial& tPol, uint64_t, uint64_t*, uint64_t, uint64_t) { calculateH1H2(h1, h2, fPol, tPol); }
};

// ------------------------------------------------------------------- FRIProof
class MerkleProof
{
public:
    std::vector<std::vector<Goldilocks::Element>> v;
    std::vector<std::vector<Goldilocks::Element>> mp;
    MerkleProof(uint64_t nLinears, uint64_t elementsTree, Goldilocks::Element* pointer)
        : v(nLinears, std::vector<Goldilocks::Element>(1, Goldilocks::zero())), mp(elementsTree, std::vector<Goldilocks::Element>(HASH_SIZE, Goldilocks::zero()))
    {
        for (uint64_t i = 0; i < nLinears; i++) v[i][0] = pointer[i];
        for (uint64_t j = 0; j < elementsTree; j++)
            std::memcpy(&mp[j][0], &pointer[nLinears + j * HASH_SIZE], HASH_SIZE * sizeof(Goldilocks::Element));
    }
};
class ProofTree
{
public:
    std::vector<Goldilocks::Element> root;
    std::vector<std::vector<MerkleProof>> polQueries;
    ProofTree() : root(HASH_SIZE) {}
    void setRoot(Goldilocks::Element* _root) { std::memcpy(&root[0], _root, HASH_SIZE * sizeof(Goldilocks::Element)); }
};
class Fri
{
public:
    std::vector<std::vector<Goldilocks::Element>> pol;
    std::vector<ProofTree> trees;
    Fri(uint64_t polN, uint64_t dim, uint64_t numSteps) : pol(polN, std::vector<Goldilocks::Element>(dim, Goldilocks::zero())), trees(numSteps) {}
    void setPol(Goldilocks::Element* pPol)
    {
        for (uint64_t i = 0; i < pol.size(); i++) std::memcpy(&pol[i][0], &pPol[i * pol[i].size()], pol[i].size() * sizeof(Goldilocks::Element));
    }
};
class Proofs
{
public:
    std::vector<Goldilocks::Element> root1, root2, root3, root4;
    Fri fri;
    std::vector<std::vector<Goldilocks::Element>> evals;
    Proofs(uint64_t polN, uint64_t dim, uint64_t numSteps, uint64_t evalSize)
        : root1(HASH_SIZE, Goldilocks::zero()), root2(HASH_SIZE, Goldilocks::zero()), root3(HASH_SIZE, Goldilocks::zero()),
          root4(HASH_SIZE, Goldilocks::zero()), fri(polN, dim, numSteps), evals(evalSize, std::vector<Goldilocks::Element>(dim, Goldilocks::zero())) {}
    void setEvals(Goldilocks::Element* _evals)
    {
        for (uint64_t i = 0; i < evals.size(); i++) std::memcpy(&evals[i][0], &_evals[i * evals[i].size()], evals[i].size() * sizeof(Goldilocks::Element));
    }
};
class FRIProof
{
public:
    Proofs proofs;
    std::vector<Goldilocks::Element> publics;
    FRIProof(uint64_t polN, uint64_t dim, uint64_t numTrees, uint64_t evalSize, uint64_t nPublics)
        : proofs(polN, dim, numTrees, evalSize), publics(nPublics) {}
};

// ------------------------------------------------------------------ StarkInfo (stark_info.hpp:19-336, stark_info.cpp:21-520)
// The contents of <pil>.starkinfo.json: STARK geometry, the slab layout (mapOffsets / mapSectionsN / varPolMap), the
// plookup / permutation / connection contexts, the evaluation map and the step programs.  Same member names and
// accessors as the reference; parsed with the small JSON reader below (the image has no nlohmann/json).
namespace xlp_json {
struct Value {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    uint64_t u = 0;                 // numbers are unsigned integers in starkinfo files
    std::string s;                  // string value, or the literal text of a number
    std::vector<Value> a;
    std::vector<std::pair<std::string, Value>> o;
    bool is_null() const { return kind == Null; }
    bool contains(const std::string& k) const { for (const auto& kv : o) if (kv.first == k) return true; return false; }
    size_t size() const { return kind == Array ? a.size() : kind == Object ? o.size() : 0; }
    const Value& operator[](const std::string& k) const
    {
        for (const auto& kv : o) if (kv.first == k) return kv.second;
        xlp_exit_process(("starkinfo: missing member \"" + k + "\"").c_str());
        return *this;
    }
    const Value& operator[](const char* k) const { return (*this)[std::string(k)]; }
    const Value& operator[](size_t i) const { if (i >= a.size()) xlp_exit_process("starkinfo: array index out of range"); return a[i]; }
    operator uint64_t() const { if (kind != Number) xlp_exit_process("starkinfo: number expected"); return u; }
    operator bool() const { if (kind != Bool) xlp_exit_process("starkinfo: boolean expected"); return b; }
    std::string str() const { if (kind != String && kind != Number) xlp_exit_process("starkinfo: string expected"); return s; }
};
class Parser {
    const std::string& t; size_t i = 0;
    void ws() { while (i < t.size() && (t[i] == ' ' || t[i] == '\n' || t[i] == '\t' || t[i] == '\r')) i++; }
    [[noreturn]] void bad(const char* m) { xlp_exit_process((std::string("starkinfo: JSON error: ") + m + " at byte " + std::to_string(i)).c_str()); abort(); }
    std::string string_()
    {
        std::string r; i++;
        while (i < t.size() && t[i] != '"') {
            if (t[i] == '\\' && i + 1 < t.size()) { i++; const char c = t[i]; r += c == 'n' ? '\n' : c == 't' ? '\t' : c; }
            else r += t[i];
            i++;
        }
        if (i >= t.size()) bad("unterminated string");
        i++;
        return r;
    }
public:
    explicit Parser(const std::string& text) : t(text) {}
    Value value()
    {
        ws();
        if (i >= t.size()) bad("unexpected end");
        Value v;
        const char c = t[i];
        if (c == '{') {
            v.kind = Value::Object; i++; ws();
            if (t[i] == '}') { i++; return v; }
            for (;;) {
                ws(); if (t[i] != '"') bad("member name expected");
                std::string k = string_(); ws();
                if (t[i] != ':') bad("':' expected");
                i++;
                v.o.emplace_back(k, value()); ws();
                if (t[i] == ',') { i++; continue; }
                if (t[i] == '}') { i++; return v; }
                bad("',' or '}' expected");
            }
        }
        if (c == '[') {
            v.kind = Value::Array; i++; ws();
            if (t[i] == ']') { i++; return v; }
            for (;;) {
                v.a.push_back(value()); ws();
                if (t[i] == ',') { i++; continue; }
                if (t[i] == ']') { i++; return v; }
                bad("',' or ']' expected");
            }
        }
        if (c == '"') { v.kind = Value::String; v.s = string_(); return v; }
        if (t.compare(i, 4, "true") == 0) { v.kind = Value::Bool; v.b = true; i += 4; return v; }
        if (t.compare(i, 5, "false") == 0) { v.kind = Value::Bool; v.b = false; i += 5; return v; }
        if (t.compare(i, 4, "null") == 0) { i += 4; return v; }
        const size_t st = i;
        while (i < t.size() && (isdigit((unsigned char)t[i]) || t[i] == '-' || t[i] == '+' || t[i] == '.' || t[i] == 'e' || t[i] == 'E')) i++;
        if (st == i) bad("value expected");
        v.kind = Value::Number; v.s = t.substr(st, i - st); v.u = strtoull(v.s.c_str(), NULL, 10);
        return v;
    }
};
}  // namespace xlp_json

struct StepStruct { uint64_t nBits; };
struct StarkStruct {
    uint64_t nBits = 0, nBitsExt = 0, nQueries = 0;
    std::string verificationHashType;
    std::vector<StepStruct> steps;
};
typedef enum { cm1_n = 0, cm1_2ns = 1, cm2_n = 2, cm2_2ns = 3, cm3_n = 4, cm3_2ns = 5, cm4_n = 6, cm4_2ns = 7, tmpExp_n = 8, q_2ns = 9, f_2ns = 10, eSectionMax = 11 } eSection;
inline eSection string2section(const std::string s)
{
    static const char* names[eSectionMax] = {"cm1_n", "cm1_2ns", "cm2_n", "cm2_2ns", "cm3_n", "cm3_2ns", "cm4_n", "cm4_2ns", "tmpExp_n", "q_2ns", "f_2ns"};
    for (int i = 0; i < eSectionMax; i++) if (s == names[i]) return (eSection)i;
    xlp_exit_process(("string2section() found invalid string=" + s).c_str());
    return eSectionMax;
}
struct PolsSections { uint64_t section[eSectionMax] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; };
struct PolsSectionsVector { std::vector<uint64_t> section[eSectionMax]; };
struct VarPolMap { eSection section; uint64_t dim; uint64_t sectionPos; };
struct PolInfo {
    VarPolMap map; uint64_t N, offset, size; Goldilocks::Element* pAddress;
    Goldilocks::Element* get(uint64_t step) { return pAddress + step * size; }
    Goldilocks::Element* get1(uint64_t step) { return pAddress + step * size; }
    Goldilocks::Element* get2(uint64_t step) { return pAddress + step * size + 1; }
    Goldilocks::Element* get3(uint64_t step) { return pAddress + step * size + 2; }
};
struct PeCtx { uint64_t tExpId, fExpId, zId, c1Id, numId, denId, c2Id; };
struct PuCtx { uint64_t tExpId, fExpId, h1Id, h2Id, zId, c1Id, numId, denId, c2Id; };
struct CiCtx { uint64_t zId, numId, denId, c1Id, c2Id; };
struct EvMap {
    typedef enum { cm = 0, _const = 1, q = 2 } eType;
    eType type; uint64_t id; bool prime;
    void setType(std::string s)
    {
        if (s == "cm") type = cm; else if (s == "const") type = _const; else if (s == "q") type = q;
        else xlp_exit_process(("EvMap::setType() found invalid type: " + s).c_str());
    }
};
struct StepType {
    typedef enum { tmp = 0, exp = 1, eval = 2, challenge = 3, tree1 = 4, tree2 = 5, tree3 = 6, tree4 = 7, number = 8, x = 9, Z = 10, _public = 11,
                   xDivXSubXi = 12, xDivXSubWXi = 13, cm = 14, _const = 15, q = 16, Zi = 17, tmpExp = 18, f = 19 } eType;
    eType type = tmp; uint64_t id = 0; bool prime = false; uint64_t p = 0; std::string value;
    void setType(std::string s)
    {
        static const char* names[20] = {"tmp", "exp", "eval", "challenge", "tree1", "tree2", "tree3", "tree4", "number", "x", "Z", "public", "xDivXSubXi",
                                        "xDivXSubWXi", "cm", "const", "q", "Zi", "tmpExp", "f"};
        for (int i = 0; i < 20; i++) if (s == names[i]) { type = (eType)i; return; }
        xlp_exit_process(("StepType::setType() found invalid type: " + s).c_str());
    }
};
struct StepOperation {
    typedef enum { add = 0, sub = 1, mul = 2, copy = 3 } eOperation;
    eOperation op; StepType dest; std::vector<StepType> src;
    void setOperation(std::string s)
    {
        if (s == "add") op = add; else if (s == "sub") op = sub; else if (s == "mul") op = mul; else if (s == "copy") op = copy;
        else xlp_exit_process(("StepOperation::setOperation() found invalid type: " + s).c_str());
    }
};
struct Step { std::vector<StepOperation> first, i, last; uint64_t tmpUsed = 0; };

class StarkInfo
{
public:
    StarkStruct starkStruct;
    uint64_t mapTotalN = 0, nConstants = 0, nPublics = 0, nCm1 = 0, nCm2 = 0, nCm3 = 0, nCm4 = 0, qDeg = 0, qDim = 0, friExpId = 0, nExps = 0;
    PolsSections mapDeg, mapOffsets, mapSectionsN, mapSectionsN1, mapSectionsN3;
    PolsSectionsVector mapSections;
    std::vector<VarPolMap> varPolMap;
    std::vector<uint64_t> qs, cm_n, cm_2ns;
    std::vector<PeCtx> peCtx;
    std::vector<PuCtx> puCtx;
    std::vector<CiCtx> ciCtx;
    std::vector<EvMap> evMap;
    Step step2prev, step3prev, step3, step42ns, step52ns;
    std::vector<uint64_t> exps_n, q_2nsVector, cm4_nVector, cm4_2nsVector, tmpExp_n;
    std::vector<std::pair<std::string, uint64_t>> exp2pol;
    // not in the reference's schema ("xlp" member, optional): the extended constants live inside the slab at this offset (one-GPU
    // aliased layout, tools/zkevm_geometry.py) instead of in a buffer of their own
    bool xlpConst2nsInSlab = false;
    uint64_t xlpConst2nsOffset = 0;
    // "xlp": {"world": G, "haloN": 1, "haloExt": 2^ext}: the sections are the per-rank WINDOWS of a proof row-sharded over G ranks
    // (tools/zkevm_geometry.py layout_window): N / G (NExtended / G) rows followed by haloN (haloExt) rows of the next rank; mapOffsets,
    // mapDeg and mapTotalN describe that window slab, starkStruct stays global.  0: the reference's whole-domain layout.
    uint64_t xlpWorld = 0, xlpHaloN = 0, xlpHaloExt = 0;
    bool xlpColMajor = false;                         // "xlp": {"colMajor": true}: the DEVICE sections are column-major (include/xlp_gl.h), same offsets and sizes

    StarkInfo() {}                                   // synthetic runs fill starkStruct by hand
    explicit StarkInfo(const std::string& file)      // StarkInfo(config, file), stark_info.cpp:7-19
    {
        FILE* f = fopen(file.c_str(), "rb");
        if (!f) xlp_exit_process(("StarkInfo: cannot open " + file).c_str());
        std::string text; char buf[65536]; size_t r;
        while ((r = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, r);
        fclose(f);
        load(xlp_json::Parser(text).value());
    }
    void load(const xlp_json::Value& j)
    {
        static const char* secs[eSectionMax] = {"cm1_n", "cm1_2ns", "cm2_n", "cm2_2ns", "cm3_n", "cm3_2ns", "cm4_n", "cm4_2ns", "tmpExp_n", "q_2ns", "f_2ns"};
        const xlp_json::Value& ss = j["starkStruct"];
        starkStruct.nBits = ss["nBits"]; starkStruct.nBitsExt = ss["nBitsExt"]; starkStruct.nQueries = ss["nQueries"];
        starkStruct.verificationHashType = ss["verificationHashType"].str();
        for (size_t i = 0; i < ss["steps"].size(); i++) { StepStruct st; st.nBits = ss["steps"][i]["nBits"]; starkStruct.steps.push_back(st); }
        mapTotalN = j["mapTotalN"]; nConstants = j["nConstants"]; nPublics = j["nPublics"];
        nCm1 = j["nCm1"]; nCm2 = j["nCm2"]; nCm3 = j["nCm3"]; nCm4 = j["nCm4"];
        friExpId = j["friExpId"]; nExps = j["nExps"]; qDim = j["qDim"]; qDeg = j["qDeg"];
        for (int s = 0; s < eSectionMax; s++) {
            mapDeg.section[s] = j["mapDeg"][secs[s]]; mapOffsets.section[s] = j["mapOffsets"][secs[s]];
            mapSectionsN.section[s] = j["mapSectionsN"][secs[s]]; mapSectionsN1.section[s] = j["mapSectionsN1"][secs[s]];
            mapSectionsN3.section[s] = j["mapSectionsN3"][secs[s]];
            const xlp_json::Value& v = j["mapSections"][secs[s]];
            for (size_t i = 0; i < v.size(); i++) mapSections.section[s].push_back(v[i]);
        }
        for (size_t i = 0; i < j["varPolMap"].size(); i++) {
            const xlp_json::Value& m = j["varPolMap"][i];
            varPolMap.push_back({string2section(m["section"].str()), (uint64_t)m["dim"], (uint64_t)m["sectionPos"]});
        }
        auto u64s = [&](const char* k, std::vector<uint64_t>& out) { for (size_t i = 0; i < j[k].size(); i++) out.push_back(j[k][i].is_null() ? 0 : (uint64_t)j[k][i]); };
        u64s("qs", qs); u64s("cm_n", cm_n); u64s("cm_2ns", cm_2ns);
        for (size_t i = 0; i < j["peCtx"].size(); i++) {
            const xlp_json::Value& c = j["peCtx"][i];
            peCtx.push_back({c["tExpId"], c["fExpId"], c["zId"], c["c1Id"], c["numId"], c["denId"], c["c2Id"]});
        }
        for (size_t i = 0; i < j["puCtx"].size(); i++) {
            const xlp_json::Value& c = j["puCtx"][i];
            puCtx.push_back({c["tExpId"], c["fExpId"], c["h1Id"], c["h2Id"], c["zId"], c["c1Id"], c["numId"], c["denId"], c["c2Id"]});
        }
        for (size_t i = 0; i < j["ciCtx"].size(); i++) {
            const xlp_json::Value& c = j["ciCtx"][i];
            ciCtx.push_back({c["zId"], c["numId"], c["denId"], c["c1Id"], c["c2Id"]});
        }
        for (size_t i = 0; i < j["evMap"].size(); i++) {
            EvMap m; m.setType(j["evMap"][i]["type"].str()); m.id = j["evMap"][i]["id"]; m.prime = j["evMap"][i]["prime"];
            evMap.push_back(m);
        }
        auto step = [&](const char* name, Step& st) {
            st.tmpUsed = j[name]["tmpUsed"];
            const xlp_json::Value& ops = j[name]["first"];
            for (size_t i = 0; i < ops.size(); i++) {
                StepOperation op;
                op.setOperation(ops[i]["op"].str());
                auto type_ = [](const xlp_json::Value& v, StepType& ty) {
                    ty.setType(v["type"].str());
                    ty.id = v.contains("id") ? (uint64_t)v["id"] : 0;
                    ty.prime = v.contains("prime") ? (bool)v["prime"] : false;
                    ty.p = v.contains("p") ? (uint64_t)v["p"] : 0;
                    if (v.contains("value")) ty.value = v["value"].str();
                };
                type_(ops[i]["dest"], op.dest);
                for (size_t k = 0; k < ops[i]["src"].size(); k++) { StepType ty; type_(ops[i]["src"][k], ty); op.src.push_back(ty); }
                st.first.push_back(op);
            }
        };
        step("step2prev", step2prev); step("step3prev", step3prev); step("step3", step3); step("step42ns", step42ns); step("step52ns", step52ns);
        u64s("exps_n", exps_n); u64s("q_2ns", q_2nsVector); u64s("cm4_n", cm4_nVector); u64s("cm4_2ns", cm4_2nsVector); u64s("tmpExp_n", tmpExp_n);
        for (const auto& kv : j["exp2pol"].o) exp2pol.emplace_back(kv.first, (uint64_t)kv.second);
        if (j.contains("xlp") && j["xlp"].contains("const2nsOffset") && !j["xlp"]["const2nsOffset"].is_null()) {
            xlpConst2nsInSlab = true;
            xlpConst2nsOffset = j["xlp"]["const2nsOffset"];
        }
        if (j.contains("xlp") && j["xlp"].contains("colMajor") && j["xlp"]["colMajor"].kind == xlp_json::Value::Bool) xlpColMajor = (bool)j["xlp"]["colMajor"];
        if (j.contains("xlp") && j["xlp"].contains("world")) {
            xlpWorld = j["xlp"]["world"]; xlpHaloN = j["xlp"]["haloN"]; xlpHaloExt = j["xlp"]["haloExt"];
            if (!xlpWorld || (xlpWorld & (xlpWorld - 1)) || xlpConst2nsInSlab) xlp_exit_process("StarkInfo: bad sharded (\"xlp\") layout");
        }
    }
    void getPol(void* pAddress, uint64_t idPol, PolInfo& polInfo)     // stark_info.cpp:456-464
    {
        polInfo.map = varPolMap[idPol];
        polInfo.N = mapDeg.section[polInfo.map.section];
        polInfo.offset = mapOffsets.section[polInfo.map.section] + polInfo.map.sectionPos;
        polInfo.size = mapSectionsN.section[polInfo.map.section];
        polInfo.pAddress = ((Goldilocks::Element*)pAddress) + polInfo.offset;
    }
    uint64_t getPolSize(uint64_t polId) { const VarPolMap p = varPolMap[polId]; return mapDeg.section[p.section] * p.dim * sizeof(Goldilocks::Element); }
    // stark_info.cpp:473-482 (returned by pointer-free value in the reference; Polinomial is not copyable here, so the view is built in place)
    void getPolinomial(Polinomial& out, Goldilocks::Element* pAddress, uint64_t idPol)
    {
        const VarPolMap p = varPolMap[idPol];
        new (&out) Polinomial(&pAddress[mapOffsets.section[p.section] + p.sectionPos], mapDeg.section[p.section], p.dim, mapSectionsN.section[p.section], std::to_string(idPol));
    }
    uint64_t getConstTreeSizeInBytes() const
    {
        const uint64_t NExtended = 1ULL << starkStruct.nBitsExt;
        return (nConstants * NExtended + NExtended * HASH_SIZE + (NExtended - 1) * HASH_SIZE + 2) * sizeof(Goldilocks::Element);
    }
};

// -------------------------------------------------------------------- FRIProve
class FRIProve
{
public:
    // friProve.cpp:5-190.  The polynomial is uploaded once; fold / transpose / Merkle run on
    // the GPU per step; only roots, the final polynomial and the opened rows come back.
    // d_friPol != NULL: the FRI polynomial already lives in HBM (GPU-resident Starks); friPol then only receives the final polynomial
    static void prove(FRIProof& fproof, MerkleTreeGL** treesGL, Transcript transcript, Polinomial& friPol, uint64_t polBits, StarkInfo starkInfo, const uint64_t* d_friPol = NULL)
    {
        const std::vector<StepStruct>& steps = starkInfo.starkStruct.steps;
        const uint64_t nSteps = steps.size();
        Goldilocks::Element polShiftInv = Goldilocks::inv(Goldilocks::shift());
        std::vector<MerkleTreeGL*> treesFRIGL(nSteps, NULL);
        std::vector<uint64_t*> friSources;

        uint64_t* d_pol = NULL; uint64_t* d_pol2 = NULL;
        const uint64_t n0 = 1ULL << polBits;
        XLP_CHECK(xlp_dev_alloc((void**)&d_pol, n0 * 3 * 8));
        XLP_CHECK(xlp_dev_alloc((void**)&d_pol2, n0 * 3 * 8));
        if (friPol.offset() != FIELD_EXTENSION) xlp_exit_process("FRIProve::prove: friPol must be dense (offset == 3)");
        if (d_friPol) XLP_CHECK(xlp_dev_copy_cols(d_pol, 3, d_friPol, 3, n0, 3));
        else XLP_CHECK(xlp_copy_h2d(d_pol, friPol.address(), n0 * 3 * 8));

        uint64_t pol2N = 0;
        for (uint64_t si = 0; si < nSteps; si++) {
            const uint64_t reductionBits = polBits - steps[si].nBits;
            pol2N = 1ULL << (polBits - reductionBits);
            Goldilocks::Element special_x[FIELD_EXTENSION];
            transcript.getField(special_x);
            XLP_CHECK(xlp_dev_fri_fold(d_pol2, d_pol, (uint32_t)polBits, (uint32_t)reductionBits, xlp_raw(special_x), Goldilocks::toU64(polShiftInv)));
            if (si < nSteps - 1) {
                const uint64_t nGroups = 1ULL << steps[si + 1].nBits;
                const uint64_t groupSize = (1ULL << steps[si].nBits) / nGroups;
                // the step's tree stays in HBM (source = the transposed groups, nodes): only its root and, at the end, the
                // queried rows + sibling paths come back
                uint64_t* d_src = NULL;
                XLP_CHECK(xlp_dev_alloc((void**)&d_src, pol2N * 3 * 8));
                friSources.push_back(d_src);
                XLP_CHECK(xlp_dev_transpose3(d_src, d_pol2, pol2N, (uint32_t)steps[si + 1].nBits));
                treesFRIGL[si + 1] = new MerkleTreeGL(nGroups, groupSize * FIELD_EXTENSION, d_src, true);
                MerkleTreeGL* t = treesFRIGL[si + 1];
                t->merkelize();
                Goldilocks::Element rootGL[HASH_SIZE];
                t->getRoot(rootGL);
                transcript.put(rootGL, HASH_SIZE);
                fproof.proofs.fri.trees[si + 1].setRoot(rootGL);
            } else {
                std::vector<Goldilocks::Element> last(pol2N * FIELD_EXTENSION);
                XLP_CHECK(xlp_copy_d2h(last.data(), d_pol2, pol2N * 3 * 8));
                for (uint64_t i = 0; i < pol2N; i++) transcript.put(&last[i * FIELD_EXTENSION], FIELD_EXTENSION);
            }
            std::swap(d_pol, d_pol2);   // friPol <- pol2
            polBits -= reductionBits;
            for (uint64_t j = 0; j < reductionBits; j++) polShiftInv = Goldilocks::square(polShiftInv);
        }
        // final polynomial back into friPol (reference copies every step; only the last is observable)
        XLP_CHECK(xlp_copy_d2h(friPol.address(), d_pol, pol2N * 3 * 8));
        fproof.proofs.fri.setPol(friPol.address());

        const uint64_t nQueries = starkInfo.starkStruct.nQueries;
        std::vector<uint64_t> ys(nQueries);
        transcript.getPermutations(ys.data(), nQueries, steps[0].nBits);
        for (uint64_t si = 0; si < nSteps; si++) {
            // friProve.cpp:155-177, batched per tree: one getGroupProofs call covers every query
            std::vector<MerkleTreeGL*> trees;
            if (si == 0) for (unsigned i = 0; i < 5; i++) trees.push_back(treesGL[i]);
            else trees.push_back(treesFRIGL[si]);
            std::vector<std::vector<Goldilocks::Element>> bufs(trees.size());
            for (size_t k = 0; k < trees.size(); k++) trees[k]->getGroupProofs(bufs[k], ys);
            for (uint64_t i = 0; i < nQueries; i++) {
                std::vector<MerkleProof> vMkProof;
                for (size_t k = 0; k < trees.size(); k++) {
                    MerkleTreeGL* t = trees[k];
                    const uint64_t per = t->width + t->MerkleProofSize() * HASH_SIZE;
                    vMkProof.push_back(MerkleProof(t->width, t->MerkleProofSize(), &bufs[k][i * per]));
                }
                fproof.proofs.fri.trees[si].polQueries.push_back(vMkProof);
            }
            if (si < nSteps - 1)
                for (uint64_t i = 0; i < nQueries; i++) ys[i] = ys[i] % (1ULL << steps[si + 1].nBits);
        }
        for (MerkleTreeGL* t : treesFRIGL) delete t;
        for (uint64_t* p_ : friSources) XLP_CHECK(xlp_dev_free(p_));
        XLP_CHECK(xlp_dev_free(d_pol)); XLP_CHECK(xlp_dev_free(d_pol2));
    }
    static void queryPol(FRIProof& fproof, MerkleTreeGL* treesGL[5], uint64_t idx, uint64_t treeIdx)
    {
        std::vector<MerkleProof> vMkProof;
        for (unsigned i = 0; i < 5; i++) {
            MerkleTreeGL* t = treesGL[i];
            std::vector<Goldilocks::Element> buff(t->width + t->MerkleProofSize() * HASH_SIZE + 1, Goldilocks::zero());
            t->getGroupProof(buff.data(), idx);
            vMkProof.push_back(MerkleProof(t->width, t->MerkleProofSize(), buff.data()));
        }
        fproof.proofs.fri.trees[treeIdx].polQueries.push_back(vMkProof);
    }
    static void queryPol(FRIProof& fproof, MerkleTreeGL* treeGL, uint64_t idx, uint64_t treeIdx)
    {
        std::vector<MerkleProof> vMkProof;
        std::vector<Goldilocks::Element> buff(treeGL->width + treeGL->MerkleProofSize() * HASH_SIZE + 1, Goldilocks::zero());
        treeGL->getGroupProof(buff.data(), idx);
        vMkProof.push_back(MerkleProof(treeGL->width, treeGL->MerkleProofSize(), buff.data()));
        fproof.proofs.fri.trees[treeIdx].polQueries.push_back(vMkProof);
    }
};

// ------------------------------------------------------------- proof -> zkin
// Emits the zkin layout of proof2zkinStark.cpp:8-82 (decimal strings) directly from a FRIProof.
namespace xlp_json {
inline void arr(std::ostringstream& os, const std::vector<Goldilocks::Element>& v)
{
    os << "[";
    for (size_t i = 0; i < v.size(); i++) os << (i ? "," : "") << "\"" << Goldilocks::toString(v[i]) << "\"";
    os << "]";
}
inline void arr2(std::ostringstream& os, const std::vector<std::vector<Goldilocks::Element>>& v)
{
    os << "[";
    for (size_t i = 0; i < v.size(); i++) { os << (i ? "," : ""); arr(os, v[i]); }
    os << "]";
}
inline void vals(std::ostringstream& os, const MerkleProof& m)   // v entries are single elements -> flat array of strings
{
    os << "[";
    for (size_t i = 0; i < m.v.size(); i++) os << (i ? "," : "") << "\"" << Goldilocks::toString(m.v[i][0]) << "\"";
    os << "]";
}
}  // namespace xlp_json

inline std::string proof2zkinStark(FRIProof& proof)
{
    using namespace xlp_json;
    std::ostringstream os;
    Proofs& p = proof.proofs;
    os << "{\"root1\":"; arr(os, p.root1);
    os << ",\"root2\":"; arr(os, p.root2);
    os << ",\"root3\":"; arr(os, p.root3);
    os << ",\"root4\":"; arr(os, p.root4);
    os << ",\"evals\":"; arr2(os, p.evals);
    const std::vector<ProofTree>& trees = p.fri.trees;
    for (size_t i = 1; i < trees.size(); i++) {
        os << ",\"s" << i << "_root\":"; arr(os, trees[i].root);
        os << ",\"s" << i << "_vals\":[";
        for (size_t q = 0; q < trees[i].polQueries.size(); q++) { os << (q ? "," : ""); vals(os, trees[i].polQueries[q][0]); }
        os << "],\"s" << i << "_siblings\":[";
        for (size_t q = 0; q < trees[i].polQueries.size(); q++) { os << (q ? "," : ""); arr2(os, trees[i].polQueries[q][0].mp); }
        os << "]";
    }
    static const char* names[5] = {"1", "2", "3", "4", "C"};
    const std::vector<std::vector<MerkleProof>>& q0 = trees[0].polQueries;
    for (int t = 0; t < 5; t++) {
        if ((t == 1 || t == 2) && (q0.empty() || q0[0][t].v.empty())) continue;   // absent sections are omitted
        os << ",\"s0_vals" << names[t] << "\":[";
        for (size_t q = 0; q < q0.size(); q++) { os << (q ? "," : ""); vals(os, q0[q][t]); }
        os << "],\"s0_siblings" << names[t] << "\":[";
        for (size_t q = 0; q < q0.size(); q++) { os << (q ? "," : ""); arr2(os, q0[q][t].mp); }
        os << "]";
    }
    os << ",\"finalPol\":"; arr2(os, p.fri.pol);
    os << "}";
    return os.str();
}

// ------------------------------------------------------------------ joinzkin
// The input of the recursive2 verifier circuit from two recursive1 / recursive2 proofs (proof2zkinStark.cpp:84-175, called by
// Prover::genAggregatedProof, prover.cpp:728): the publics are stitched (old state of the first proof, new state of the
// second), every member of the two zkin objects reappears under an "a_" / "b_" prefix and rootC is the verification key's
// constRoot as decimal strings.  Host-only; works on the JSON reader's values and returns the JSON text (member order = the
// reference's insertion order, as its ordered_json keeps it).
namespace xlp_json {
inline void dump(std::ostringstream& os, const Value& v)
{
    switch (v.kind) {
    case Value::Null: os << "null"; break;
    case Value::Bool: os << (v.b ? "true" : "false"); break;
    case Value::Number: os << v.s; break;
    case Value::String:
        os << '"';
        for (char c : v.s) { if (c == '"' || c == '\\') os << '\\'; os << c; }
        os << '"';
        break;
    case Value::Array:
        os << '[';
        for (size_t i = 0; i < v.a.size(); i++) { if (i) os << ','; dump(os, v.a[i]); }
        os << ']';
        break;
    case Value::Object:
        os << '{';
        for (size_t i = 0; i < v.o.size(); i++) { if (i) os << ','; os << '"' << v.o[i].first << "\":"; dump(os, v.o[i].second); }
        os << '}';
        break;
    }
}
}  // namespace xlp_json

inline std::string joinzkin(const xlp_json::Value& zkin1, const xlp_json::Value& zkin2, const xlp_json::Value& verKey, uint64_t steps)
{
    using xlp_json::Value;
    std::ostringstream os;
    bool first = true;
    auto member = [&](const std::string& name, const Value& v) { os << (first ? "{" : ",") << '"' << name << "\":"; xlp_json::dump(os, v); first = false; };
    // publics 0..18 (oldStateRoot, oldAccInputHash, oldBatchNum, chainId, forkId) from the first proof, 19..43 (newStateRoot,
    // newAccInputHash, newLocalExitRoot, newBatchNum) from the second
    Value pub; pub.kind = Value::Array;
    for (size_t i = 0; i < 19; i++) pub.a.push_back(zkin1["publics"][i]);
    for (size_t i = 19; i < 44; i++) pub.a.push_back(zkin2["publics"][i]);
    member("publics", pub);
    static const char* fixed[] = {"publics", "root1", "root2", "root3", "root4", "evals", "s0_vals1", "s0_vals3", "s0_vals4", "s0_valsC",
                                  "s0_siblings1", "s0_siblings3", "s0_siblings4", "s0_siblingsC"};
    const Value* zk[2] = {&zkin1, &zkin2};
    for (int w = 0; w < 2; w++) {
        const std::string pre = w == 0 ? "a_" : "b_";
        for (const char* k : fixed) member(pre + k, (*zk[w])[k]);
        for (uint64_t i = 1; i < steps; i++) {
            const std::string si = "s" + std::to_string(i);
            member(pre + si + "_root", (*zk[w])[si + "_root"]);
            member(pre + si + "_siblings", (*zk[w])[si + "_siblings"]);
            member(pre + si + "_vals", (*zk[w])[si + "_vals"]);
        }
        member(pre + "finalPol", (*zk[w])["finalPol"]);
    }
    Value rc; rc.kind = Value::Array;
    for (size_t i = 0; i < 4; i++) { Value e; e.kind = Value::String; e.s = verKey["constRoot"][i].str(); rc.a.push_back(e); }
    member("rootC", rc);
    os << "}";
    return os.str();
}
