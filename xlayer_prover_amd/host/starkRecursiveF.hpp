// GPU-resident `class StarkRecursiveF` with the reference's interface (src/starkpil/starkRecursiveF/starkRecursiveF.hpp:33-126,
// starkRecursiveF.cpp:13-650):
//     StarkRecursiveF(const Config& config, void* pAddress)
//     void genProof(FRIProofC12& proof, Goldilocks::Element publicInputs[8], Steps* steps)
// It is Starks::genProof (host/starks.hpp: StarksT<H>) with the BN128 hash policy: MerkleTreeBN128 trees over the committed
// sections, the constants and the FRI layers, TranscriptBN128 challenges, 16-ary openings; the first transcript input is the
// root of the constant tree, then the publics (starkRecursiveF.cpp:171-174).  Also here, with the reference's names:
//     MerkleProofC12 / ProofTreeC12 / FriC12 / ProofsC12 / FRIProofC12     src/starkpil/fri/friProofC12.hpp:11-215
//     FRIProveC12::prove                                                    src/starkpil/fri/friProveC12.cpp:5-190
//     proof2zkinStarkC12 (zkin JSON text: roots as one decimal string, siblings [query][level][16])   friProofC12.hpp json forms
#pragma once
#include "starks.hpp"
#include "xlp_host_bn128.hpp"

#define STARK_RECURSIVE_F_NUM_TREES 5

class MerkleProofC12
{
public:
    std::vector<std::vector<Goldilocks::Element>> v;
    std::vector<std::vector<RawFr::Element>> mp;
    MerkleProofC12(uint64_t nLinears, uint64_t elementsTree, void* pointer)
        : v(nLinears, std::vector<Goldilocks::Element>(1, Goldilocks::zero())), mp(elementsTree, std::vector<RawFr::Element>(16, RawFr::field.zero()))
    {
        for (uint64_t i = 0; i < nLinears; i++) v[i][0] = ((Goldilocks::Element*)pointer)[i];
        const RawFr::Element* mpCursor = (const RawFr::Element*)&((Goldilocks::Element*)pointer)[nLinears];
        for (uint64_t j = 0; j < elementsTree; j++) std::memcpy(&mp[j][0], &mpCursor[j * 16], 16 * sizeof(RawFr::Element));
    }
};
class ProofTreeC12
{
public:
    std::vector<RawFr::Element> root;
    std::vector<std::vector<MerkleProofC12>> polQueries;
    ProofTreeC12() : root(1, RawFr::field.zero()) {}
    void setRoot(RawFr::Element* _root) { root[0] = _root[0]; }
};
class FriC12
{
public:
    std::vector<std::vector<Goldilocks::Element>> pol;
    std::vector<ProofTreeC12> trees;
    FriC12(uint64_t polN, uint64_t dim, uint64_t numSteps) : pol(polN, std::vector<Goldilocks::Element>(dim, Goldilocks::zero())), trees(numSteps) {}
    void setPol(Goldilocks::Element* pPol)
    {
        for (uint64_t i = 0; i < pol.size(); i++) std::memcpy(&pol[i][0], &pPol[i * pol[i].size()], pol[i].size() * sizeof(Goldilocks::Element));
    }
};
class ProofsC12
{
public:
    std::vector<RawFr::Element> root1, root2, root3, root4;
    FriC12 fri;
    std::vector<std::vector<Goldilocks::Element>> evals;
    ProofsC12(uint64_t polN, uint64_t dim, uint64_t numSteps, uint64_t evalSize)
        : root1(1, RawFr::field.zero()), root2(1, RawFr::field.zero()), root3(1, RawFr::field.zero()), root4(1, RawFr::field.zero()),
          fri(polN, dim, numSteps), evals(evalSize, std::vector<Goldilocks::Element>(dim, Goldilocks::zero())) {}
    void setEvals(Goldilocks::Element* _evals)
    {
        for (uint64_t i = 0; i < evals.size(); i++) std::memcpy(&evals[i][0], &_evals[i * evals[i].size()], evals[i].size() * sizeof(Goldilocks::Element));
    }
};
class FRIProofC12
{
public:
    ProofsC12 proofs;
    std::vector<Goldilocks::Element> publics;
    FRIProofC12(uint64_t polN, uint64_t dim, uint64_t numTrees, uint64_t evalSize, uint64_t nPublics)
        : proofs(polN, dim, numTrees, evalSize), publics(nPublics) {}
};

class FRIProveC12
{
public:
    // friProveC12.cpp:5-190: FRIProve::prove with MerkleTreeBN128 layer trees.  The polynomial stays in HBM; per step fold /
    // transpose / tree run on the GPU; only roots, the final polynomial and the opened rows + 16-node groups come back.
    static void prove(FRIProofC12& fproof, MerkleTreeBN128** treesBN128, TranscriptBN128 transcript, Polinomial& friPol, uint64_t polBits, StarkInfo starkInfo,
                      const uint64_t* d_friPol = NULL)
    {
        const std::vector<StepStruct>& steps = starkInfo.starkStruct.steps;
        const uint64_t nSteps = steps.size();
        Goldilocks::Element polShiftInv = Goldilocks::inv(Goldilocks::shift());
        std::vector<MerkleTreeBN128*> treesFRI(nSteps, NULL);
        std::vector<uint64_t*> friSources;
        uint64_t *d_pol = NULL, *d_pol2 = NULL;
        const uint64_t n0 = 1ULL << polBits;
        XLP_CHECK(xlp_dev_alloc((void**)&d_pol, n0 * 3 * 8));
        XLP_CHECK(xlp_dev_alloc((void**)&d_pol2, n0 * 3 * 8));
        if (friPol.offset() != FIELD_EXTENSION) xlp_exit_process("FRIProveC12::prove: friPol must be dense (offset == 3)");
        if (d_friPol) XLP_CHECK(xlp_dev_copy_cols(d_pol, 3, d_friPol, 3, n0, 3));
        else XLP_CHECK(xlp_copy_h2d(d_pol, friPol.address(), n0 * 3 * 8));
        uint64_t pol2N = 0;
        for (uint64_t si = 0; si < nSteps; si++) {
            const uint64_t reductionBits = polBits - steps[si].nBits;
            pol2N = 1ULL << (polBits - reductionBits);
            uint64_t special_x[FIELD_EXTENSION];
            transcript.getField(special_x);
            XLP_CHECK(xlp_dev_fri_fold(d_pol2, d_pol, (uint32_t)polBits, (uint32_t)reductionBits, special_x, Goldilocks::toU64(polShiftInv)));
            if (si < nSteps - 1) {
                const uint64_t nGroups = 1ULL << steps[si + 1].nBits;
                const uint64_t groupSize = (1ULL << steps[si].nBits) / nGroups;
                uint64_t* d_src = NULL;
                XLP_CHECK(xlp_dev_alloc((void**)&d_src, pol2N * 3 * 8));
                friSources.push_back(d_src);
                XLP_CHECK(xlp_dev_transpose3(d_src, d_pol2, pol2N, (uint32_t)steps[si + 1].nBits));
                MerkleTreeBN128* t = treesFRI[si + 1] = new MerkleTreeBN128(nGroups, groupSize * FIELD_EXTENSION, d_src, true);
                t->merkelize();
                RawFr::Element root;
                t->getRoot(&root);
                transcript.put(&root, 1);
                fproof.proofs.fri.trees[si + 1].setRoot(&root);
            } else {
                std::vector<Goldilocks::Element> last(pol2N * FIELD_EXTENSION);
                XLP_CHECK(xlp_copy_d2h(last.data(), d_pol2, pol2N * 3 * 8));
                for (uint64_t i = 0; i < pol2N; i++) transcript.put(&last[i * FIELD_EXTENSION], FIELD_EXTENSION);
            }
            std::swap(d_pol, d_pol2);
            polBits -= reductionBits;
            for (uint64_t j = 0; j < reductionBits; j++) polShiftInv = Goldilocks::square(polShiftInv);
        }
        XLP_CHECK(xlp_copy_d2h(friPol.address(), d_pol, pol2N * 3 * 8));
        fproof.proofs.fri.setPol(friPol.address());

        const uint64_t nQueries = starkInfo.starkStruct.nQueries;
        std::vector<uint64_t> ys(nQueries);
        transcript.getPermutations(ys.data(), nQueries, steps[0].nBits);
        for (uint64_t si = 0; si < nSteps; si++) {
            // friProveC12.cpp:155-177, batched per tree: one getGroupProofs call covers every query
            std::vector<MerkleTreeBN128*> trees;
            if (si == 0) for (unsigned i = 0; i < 5; i++) trees.push_back(treesBN128[i]);
            else trees.push_back(treesFRI[si]);
            std::vector<std::vector<uint8_t>> bufs(trees.size());
            for (size_t k = 0; k < trees.size(); k++) trees[k]->getGroupProofs(bufs[k], ys);
            for (uint64_t i = 0; i < nQueries; i++) {
                std::vector<MerkleProofC12> vMkProof;
                for (size_t k = 0; k < trees.size(); k++) {
                    MerkleTreeBN128* t = trees[k];
                    const uint64_t per = t->source_width * sizeof(Goldilocks::Element) + MerkleTreeBN128::getMerkleProofSize(t->height);
                    vMkProof.push_back(MerkleProofC12(t->source_width, MerkleTreeBN128::getMerkleProofLength(t->height), &bufs[k][i * per]));
                }
                fproof.proofs.fri.trees[si].polQueries.push_back(vMkProof);
            }
            if (si < nSteps - 1)
                for (uint64_t i = 0; i < nQueries; i++) ys[i] = ys[i] % (1ULL << steps[si + 1].nBits);
        }
        for (MerkleTreeBN128* t : treesFRI) delete t;
        for (uint64_t* p_ : friSources) XLP_CHECK(xlp_dev_free(p_));
        XLP_CHECK(xlp_dev_free(d_pol)); XLP_CHECK(xlp_dev_free(d_pol2));
    }
    static MerkleProofC12 open(MerkleTreeBN128* t, uint64_t idx)
    {
        const uint64_t plen = MerkleTreeBN128::getMerkleProofLength(t->height);
        std::vector<uint8_t> buff(t->source_width * sizeof(Goldilocks::Element) + MerkleTreeBN128::getMerkleProofSize(t->height) + 8, 0);
        t->getGroupProof(buff.data(), idx);
        return MerkleProofC12(t->source_width, plen, buff.data());
    }
    static void queryPol(FRIProofC12& fproof, MerkleTreeBN128* trees[5], uint64_t idx, uint64_t treeIdx)
    {
        std::vector<MerkleProofC12> vMkProof;
        for (unsigned i = 0; i < 5; i++) vMkProof.push_back(open(trees[i], idx));
        fproof.proofs.fri.trees[treeIdx].polQueries.push_back(vMkProof);
    }
    static void queryPol(FRIProofC12& fproof, MerkleTreeBN128* tree, uint64_t idx, uint64_t treeIdx)
    {
        std::vector<MerkleProofC12> vMkProof;
        vMkProof.push_back(open(tree, idx));
        fproof.proofs.fri.trees[treeIdx].polQueries.push_back(vMkProof);
    }
};

struct XlpHashBN128
{
    typedef MerkleTreeBN128 Tree;
    typedef TranscriptBN128 TranscriptT;
    typedef FRIProofC12 Proof;
    typedef FRIProveC12 Prove;
    typedef RawFr::Element Root[1];
    static constexpr bool kCommitTimes = false;
    static constexpr bool kShardable = false;        // the 16-ary BN128 tree has no row-sharded form here (recursivef is 2^17 rows: one GPU)
    static constexpr bool kFileTree = true;          // recursivef.consttree as bctree (verificationHashType BN128) writes it: [width, height, source, nodes]
    static uint64_t treeWords(Tree* t) { return t->numNodes * 4; }
    static void getField(TranscriptT& t, Goldilocks::Element* out)
    {
        uint64_t f[3];
        t.getField(f);                                // three 64-bit limbs of an Fr output: raw words, as the reference writes them into the challenge (transcriptBN128.cpp:35-41)
        for (int i = 0; i < 3; i++) out[i].fe = f[i];
    }
    static void putRoot(TranscriptT& t, RawFr::Element* r) { t.put(r, 1); }
    static void putVerkey(TranscriptT& t, Goldilocks::Element* /*verkey*/, RawFr::Element* constRoot) { t.put(constRoot, 1); }   // starkRecursiveF.cpp:171-173
    static void commit(Tree* tree, uint64_t* sec2ns, const uint64_t* secn, uint64_t NExtended, uint64_t N, uint64_t W)
    {
        if (W) XLP_CHECK(xlp_dev_extend(sec2ns, secn, NExtended, N, W, NULL));
        tree->d_source = sec2ns;
        if (!W && tree->emptyTreeDone) return;        // an absent section (recursivef has no cm2): same all-zero tree in every proof (0.04 s at 2^21 leaves)
        tree->merkelize();
        if (!W) tree->emptyTreeDone = true;
    }
    static void commitFromHost(Tree* tree, uint64_t* secn, uint64_t* sec2ns, const uint64_t* h_src, uint64_t NExtended, uint64_t N, uint64_t W, double* busyMs)
    {
        if (W) XLP_CHECK(xlp_copy_h2d(secn, h_src, N * W * sizeof(uint64_t)));
        if (busyMs) *busyMs = 0.0;
        commit(tree, sec2ns, secn, NExtended, N, W);
    }
    static void storeRoots(Proof& proof, RawFr::Element* r0, RawFr::Element* r1, RawFr::Element* r2, RawFr::Element* r3)
    {
        proof.proofs.root1[0] = r0[0]; proof.proofs.root2[0] = r1[0]; proof.proofs.root3[0] = r2[0]; proof.proofs.root4[0] = r3[0];
    }
};

class StarkRecursiveF : public StarksT<XlpHashBN128>
{
public:
    // the reference reads config.recursivef{StarkInfo,ConstPols,ConstantsTree}; here the three files are given (an absent tree file: computed)
    StarkRecursiveF(const Config& config, StarkFiles starkFiles, void* _pAddress) : StarksT<XlpHashBN128>(config, starkFiles, _pAddress) {}
    void genProof(FRIProofC12& proof, Goldilocks::Element* publicInputs, Steps* steps) { StarksT<XlpHashBN128>::genProof(proof, publicInputs, NULL, steps); }
};

// zkin text of a recursivef proof: the key layout of proof2zkinStark (proof2zkinStark.cpp:8-82) with one decimal string per root and
// [query][level][16] sibling groups (friProofC12.hpp:26-60 json forms)
inline std::string proof2zkinStarkC12(FRIProofC12& proof)
{
    std::ostringstream os;
    ProofsC12& p = proof.proofs;
    auto fr = [](const RawFr::Element& e) { return "\"" + RawFr::field.toString(e, 10) + "\""; };
    auto vals = [&](const MerkleProofC12& m) {
        os << "[";
        for (size_t i = 0; i < m.v.size(); i++) os << (i ? "," : "") << "\"" << Goldilocks::toString(m.v[i][0]) << "\"";
        os << "]";
    };
    auto sib = [&](const MerkleProofC12& m) {
        os << "[";
        for (size_t l = 0; l < m.mp.size(); l++) {
            os << (l ? "," : "") << "[";
            for (size_t k = 0; k < m.mp[l].size(); k++) os << (k ? "," : "") << fr(m.mp[l][k]);
            os << "]";
        }
        os << "]";
    };
    os << "{\"root1\":" << fr(p.root1[0]) << ",\"root2\":" << fr(p.root2[0]) << ",\"root3\":" << fr(p.root3[0]) << ",\"root4\":" << fr(p.root4[0]);
    os << ",\"evals\":"; xlp_json::arr2(os, p.evals);
    const std::vector<ProofTreeC12>& trees = p.fri.trees;
    for (size_t i = 1; i < trees.size(); i++) {
        os << ",\"s" << i << "_root\":" << fr(trees[i].root[0]);
        os << ",\"s" << i << "_vals\":[";
        for (size_t q = 0; q < trees[i].polQueries.size(); q++) { os << (q ? "," : ""); vals(trees[i].polQueries[q][0]); }
        os << "],\"s" << i << "_siblings\":[";
        for (size_t q = 0; q < trees[i].polQueries.size(); q++) { os << (q ? "," : ""); sib(trees[i].polQueries[q][0]); }
        os << "]";
    }
    static const char* names[5] = {"1", "2", "3", "4", "C"};
    const std::vector<std::vector<MerkleProofC12>>& q0 = trees[0].polQueries;
    for (int t = 0; t < 5; t++) {
        if ((t == 1 || t == 2) && (q0.empty() || q0[0][t].v.empty())) continue;   // absent sections are omitted
        os << ",\"s0_vals" << names[t] << "\":[";
        for (size_t q = 0; q < q0.size(); q++) { os << (q ? "," : ""); vals(q0[q][t]); }
        os << "],\"s0_siblings" << names[t] << "\":[";
        for (size_t q = 0; q < q0.size(); q++) { os << (q ? "," : ""); sib(q0[q][t]); }
        os << "]";
    }
    os << ",\"finalPol\":"; xlp_json::arr2(os, p.fri.pol);
    os << "}";
    return os.str();
}
