// A cut-down Starks::genProof (starks.cpp:9-404) over the host mirror: the
// commit stages (extendPol / NTT -> merkelize -> transcript) and FRIProve::prove
// in the reference's order, with the constraint-evaluation steps (chelpers,
// SURVEY.md 8a12 "next" tier) replaced by seeded synthetic polynomials.
// Timer names follow starks.cpp so logs diff against the reference's.
#pragma once
#include <chrono>
#include <string>
#include <vector>
#include "xlp_host.hpp"

struct MiniStarkConfig {
    uint64_t nBits = 10, nBitsExt = 11, nQueries = 8;
    std::vector<uint64_t> steps;
    uint64_t w1 = 18, w3 = 39, w4 = 21, wC = 52, nEvals = 10, nPublics = 8;
    bool log_time = false;
};

static inline uint64_t mini_splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline void mini_fill(Goldilocks::Element* dst, uint64_t n, uint64_t seed)
{
    for (uint64_t i = 0; i < n; i++) dst[i].fe = mini_splitmix64(seed ^ i) % Goldilocks::P;
}

struct MiniTimer {
    const char* name; bool on; std::chrono::steady_clock::time_point t0;
    MiniTimer(const char* n, bool o) : name(n), on(o), t0(std::chrono::steady_clock::now()) { if (on) fprintf(stderr, "--> %s starting...\n", name); }
    ~MiniTimer() { if (on) fprintf(stderr, "<-- %s done: %.6f s\n", name, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()); }
};

inline std::string mini_stark_prove(const MiniStarkConfig& cfg)
{
    const uint64_t N = 1ULL << cfg.nBits, NExtended = 1ULL << cfg.nBitsExt;
    NTT_Goldilocks ntt(N), nttExtended(NExtended);
    StarkInfo starkInfo;
    starkInfo.starkStruct.nBits = cfg.nBits;
    starkInfo.starkStruct.nBitsExt = cfg.nBitsExt;
    starkInfo.starkStruct.nQueries = cfg.nQueries;
    for (uint64_t s : cfg.steps) starkInfo.starkStruct.steps.push_back({s});

    typedef Goldilocks::Element E;
    std::vector<E> cm1_n(N * cfg.w1), cm3_n(N * cfg.w3), const_n(N * cfg.wC), qq2(NExtended * cfg.w4);
    std::vector<E> cm1_2ns(NExtended * cfg.w1), cm3_2ns(NExtended * cfg.w3), cm4_2ns(NExtended * cfg.w4), const_2ns(NExtended * cfg.wC);
    std::vector<E> f_2ns(NExtended * FIELD_EXTENSION), evals(cfg.nEvals * FIELD_EXTENSION), publics(cfg.nPublics);
    std::vector<E> pBuffer(NExtended * cfg.wC);
    mini_fill(cm1_n.data(), cm1_n.size(), 0x5EED0101);
    mini_fill(cm3_n.data(), cm3_n.size(), 0x5EED0103);
    mini_fill(const_n.data(), const_n.size(), 0x5EED010C);
    mini_fill(qq2.data(), qq2.size(), 0x5EED0104);
    mini_fill(f_2ns.data(), f_2ns.size(), 0x5EED010F);
    mini_fill(evals.data(), evals.size(), 0x5EED010E);
    mini_fill(publics.data(), publics.size(), 0x5EED0100);

    // constant tree (bctree, build_const_tree.cpp:533-603) -> verkey
    ntt.extendPol(const_2ns.data(), const_n.data(), NExtended, N, cfg.wC, pBuffer.data());
    MerkleTreeGL* treesGL[5];
    treesGL[0] = new MerkleTreeGL(NExtended, cfg.w1, cm1_2ns.data());
    treesGL[1] = new MerkleTreeGL(NExtended, 0, NULL);                  // cm2: no columns in this STARK
    treesGL[2] = new MerkleTreeGL(NExtended, cfg.w3, cm3_2ns.data());
    treesGL[3] = new MerkleTreeGL(NExtended, cfg.w4, cm4_2ns.data());
    treesGL[4] = new MerkleTreeGL(NExtended, cfg.wC, const_2ns.data());
    treesGL[4]->merkelize();
    E verkey[HASH_SIZE];
    treesGL[4]->getRoot(verkey);

    FRIProof proof(1ULL << cfg.steps.back(), FIELD_EXTENSION, cfg.steps.size(), cfg.nEvals, cfg.nPublics);
    for (uint64_t i = 0; i < cfg.nPublics; i++) proof.publics[i] = publics[i];
    Transcript transcript;
    E root0[HASH_SIZE], root1[HASH_SIZE], root2[HASH_SIZE], root3[HASH_SIZE], challenge[FIELD_EXTENSION];
    transcript.put(verkey, HASH_SIZE);
    transcript.put(publics.data(), cfg.nPublics);
    {
        MiniTimer t("STARK_STEP_1_LDE_AND_MERKLETREE", cfg.log_time);
        { MiniTimer t1("STARK_STEP_1_LDE", cfg.log_time); ntt.extendPol(cm1_2ns.data(), cm1_n.data(), NExtended, N, cfg.w1, pBuffer.data()); }
        { MiniTimer t2("STARK_STEP_1_MERKLETREE", cfg.log_time); treesGL[0]->merkelize(); treesGL[0]->getRoot(root0); }
        transcript.put(root0, HASH_SIZE);
    }
    transcript.getField(challenge); transcript.getField(challenge);     // u, defVal
    {
        MiniTimer t("STARK_STEP_2_LDE_AND_MERKLETREE", cfg.log_time);
        treesGL[1]->merkelize(); treesGL[1]->getRoot(root1);
        transcript.put(root1, HASH_SIZE);
    }
    transcript.getField(challenge); transcript.getField(challenge);     // gamma, beta
    {
        MiniTimer t("STARK_STEP_3_LDE_AND_MERKLETREE", cfg.log_time);
        ntt.extendPol(cm3_2ns.data(), cm3_n.data(), NExtended, N, cfg.w3, pBuffer.data());
        treesGL[2]->merkelize(); treesGL[2]->getRoot(root2);
        transcript.put(root2, HASH_SIZE);
    }
    transcript.getField(challenge);                                      // challenges[4]
    {
        MiniTimer t("STARK_STEP_4_CALCULATE_EXPS_2NS_NTT", cfg.log_time);
        nttExtended.NTT(cm4_2ns.data(), qq2.data(), NExtended, cfg.w4);
        treesGL[3]->merkelize(); treesGL[3]->getRoot(root3);
        transcript.put(root3, HASH_SIZE);
    }
    transcript.getField(challenge);                                      // xi
    for (uint64_t i = 0; i < cfg.nEvals; i++) transcript.put(&evals[i * FIELD_EXTENSION], FIELD_EXTENSION);
    transcript.getField(challenge); transcript.getField(challenge);     // v1, v2
    {
        MiniTimer t("STARK_STEP_FRI", cfg.log_time);
        Polinomial friPol(f_2ns.data(), NExtended, FIELD_EXTENSION, FIELD_EXTENSION, "friPol");
        FRIProve::prove(proof, treesGL, transcript, friPol, cfg.nBitsExt, starkInfo);
    }
    proof.proofs.setEvals(evals.data());
    std::memcpy(&proof.proofs.root1[0], root0, sizeof(root0));
    std::memcpy(&proof.proofs.root2[0], root1, sizeof(root1));
    std::memcpy(&proof.proofs.root3[0], root2, sizeof(root2));
    std::memcpy(&proof.proofs.root4[0], root3, sizeof(root3));
    std::string zkin = proof2zkinStark(proof);
    // append publics + verkey for the checker
    std::ostringstream os;
    os << zkin.substr(0, zkin.size() - 1) << ",\"publics\":";
    xlp_json::arr(os, proof.publics);
    os << ",\"verkey\":[";
    for (int i = 0; i < HASH_SIZE; i++) os << (i ? "," : "") << "\"" << Goldilocks::toString(verkey[i]) << "\"";
    os << "]}";
    for (int i = 0; i < 5; i++) delete treesGL[i];
    return os.str();
}
