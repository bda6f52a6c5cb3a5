// GPU-resident mirror of `class Starks` (src/starkpil/starks.hpp:29-232) and `Starks::genProof`
// (src/starkpil/starks.cpp:9-404): the same stage order, timer names and transcript sequence, driven by a StarkInfo
// (stark_info.hpp) and the step programs of the PIL (XlpSteps: tools/chelpers_to_xlp.py / chelpers_cpp_to_xlp.py
// output), with the whole slab, the constant polynomials, the trees and every intermediate resident in HBM.  Only the
// witness (cm1_n) goes up and roots / evaluations / final polynomial / opened rows come down.
//
// What differs from the reference, by construction:
//  * transposeH1H2Columns / transposeZColumns / ...Rows (starks.cpp:406-555) become strided device copies
//    (xlp_dev_copy_cols) around xlp_dev_calculate_h1h2 / xlp_dev_calculate_z;
//  * the constant tree is computed in the constructor (the reference maps a .consttree file built by bctree;
//    host/bctree writes that file with the same kernels);
//  * nrowsStepBatch is irrelevant (one row per lane).
#pragma once
#include <chrono>
#include <string>
#include <vector>
#include "xlp_host.hpp"

#define NUM_CHALLENGES 8

class Starks
{
public:
    StarkInfo starkInfo;
    uint64_t N, NExtended, extendBits;
    Goldilocks::Element constRoot[HASH_SIZE];
    bool logTimers = false;

    // true if step k (0 step2prev .. 4 step52ns) runs in its dot-product form (tools/optimize_programs.py)
    bool stepOptimised(int k) const { return steps.optimised(k); }

private:
    XlpSteps steps;
    uint64_t *d_mem = NULL, *d_const_n = NULL, *d_const_2ns = NULL, *d_x_n = NULL, *d_x_2ns = NULL, *d_zi = NULL, *d_xdx = NULL, *d_xdwx = NULL;
    uint64_t *d_challenges = NULL, *d_publics = NULL, *d_evals = NULL, *d_work = NULL;
    MerkleTreeGL* treesGL[5] = {NULL, NULL, NULL, NULL, NULL};

    static uint64_t* dalloc(uint64_t elems)
    {
        uint64_t* p = NULL;
        XLP_CHECK(xlp_dev_alloc((void**)&p, (elems ? elems : 1) * sizeof(uint64_t)));
        return p;
    }
    uint64_t* section(eSection s) { return d_mem + starkInfo.mapOffsets.section[s]; }
    struct PolRef { uint64_t* p; uint64_t n, dim, stride; };
    PolRef pol(uint64_t id)          // StarkInfo::getPolinomial on the device slab
    {
        if (id >= starkInfo.varPolMap.size()) xlp_exit_process("Starks: polynomial id out of range");
        const VarPolMap& m = starkInfo.varPolMap[id];
        return {d_mem + starkInfo.mapOffsets.section[m.section] + m.sectionPos, starkInfo.mapDeg.section[m.section], m.dim, starkInfo.mapSectionsN.section[m.section]};
    }
    uint64_t exp2pol(uint64_t expId)
    {
        const std::string k = std::to_string(expId);
        for (const auto& kv : starkInfo.exp2pol) if (kv.first == k) return kv.second;
        xlp_exit_process(("Starks: expression " + k + " has no polynomial (exp2pol)").c_str());
        return 0;
    }
    struct Timer {
        const char* name; bool on; std::chrono::steady_clock::time_point t0;
        Timer(const char* n, bool o) : name(n), on(o), t0(std::chrono::steady_clock::now()) {}
        ~Timer() { if (on) { xlp_sync(); fprintf(stderr, "<-- %s done: %.6f s\n", name, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()); } }
    };
    xlp_steps_params params()
    {
        xlp_steps_params P;
        memset(&P, 0, sizeof P);
        P.pols = d_mem; P.pols_size = starkInfo.mapTotalN;
        P.const_pols = d_const_n; P.n_const_pols = starkInfo.nConstants; P.const_pols_size = N * starkInfo.nConstants;
        P.const_pols_2ns = d_const_2ns; P.n_const_pols_2ns = starkInfo.nConstants; P.const_pols_2ns_size = NExtended * starkInfo.nConstants;
        P.challenges = d_challenges; P.challenges_size = NUM_CHALLENGES * FIELD_EXTENSION;
        P.public_inputs = d_publics; P.public_inputs_size = starkInfo.nPublics;
        P.evals = d_evals; P.evals_size = starkInfo.evMap.size() * FIELD_EXTENSION;
        P.x_n = d_x_n; P.x_n_size = N;
        P.x_2ns = d_x_2ns; P.x_2ns_size = NExtended;
        P.zi = d_zi; P.zi_size = 1ULL << extendBits;
        P.x_div_x_sub_xi = d_xdx; P.x_div_x_sub_xi_size = NExtended * FIELD_EXTENSION;
        P.x_div_x_sub_wxi = d_xdwx; P.x_div_x_sub_wxi_size = NExtended * FIELD_EXTENSION;
        P.q_2ns = section(q_2ns); P.q_2ns_size = NExtended * starkInfo.mapSectionsN.section[q_2ns];
        P.f_2ns = section(f_2ns); P.f_2ns_size = NExtended * starkInfo.mapSectionsN.section[f_2ns];
        return P;
    }

public:
    // constPols: host, N rows x nConstants (the .const file, starks.hpp:102-115); stepsDir/<name>.<step>.xlpexpr
    Starks(const std::string& starkInfoFile, const Goldilocks::Element* constPols, const std::string& stepsDir, const std::string& name)
        : starkInfo(starkInfoFile), steps(stepsDir, name)
    {
        N = 1ULL << starkInfo.starkStruct.nBits;
        NExtended = 1ULL << starkInfo.starkStruct.nBitsExt;
        extendBits = starkInfo.starkStruct.nBitsExt - starkInfo.starkStruct.nBits;
        if (starkInfo.mapSectionsN.section[q_2ns] != starkInfo.qDim || starkInfo.mapSectionsN.section[f_2ns] != FIELD_EXTENSION)
            xlp_exit_process("Starks: q_2ns / f_2ns sections must be qDim / 3 wide");
        if (starkInfo.mapSectionsN.section[cm4_2ns] != starkInfo.qDim * starkInfo.qDeg) xlp_exit_process("Starks: cm4_2ns must hold qDeg x qDim columns");
        XLP_CHECK(xlp_init(0));
        d_mem = dalloc(starkInfo.mapTotalN);
        d_const_n = dalloc(N * starkInfo.nConstants);
        d_const_2ns = dalloc(NExtended * starkInfo.nConstants);
        XLP_CHECK(xlp_copy_h2d(d_const_n, constPols, N * starkInfo.nConstants * sizeof(uint64_t)));
        XLP_CHECK(xlp_dev_extend(d_const_2ns, d_const_n, NExtended, N, starkInfo.nConstants, NULL));
        // x_n, x_2ns (starks.hpp:149-160), ZhInv (zhInv.cpp:7-32)
        d_x_n = dalloc(N); d_x_2ns = dalloc(NExtended);
        const uint64_t one = 1, wn = Goldilocks::toU64(Goldilocks::w(starkInfo.starkStruct.nBits)), sh = Goldilocks::toU64(Goldilocks::shift()),
                       wne = Goldilocks::toU64(Goldilocks::w(starkInfo.starkStruct.nBitsExt));
        XLP_CHECK(xlp_dev_geometric(d_x_n, N, 1, &one, &wn));
        XLP_CHECK(xlp_dev_geometric(d_x_2ns, NExtended, 1, &sh, &wne));
        {
            std::vector<uint64_t> zh(1ULL << extendBits);
            Goldilocks::Element sn = Goldilocks::shift(), w = Goldilocks::one();
            for (uint64_t i = 0; i < starkInfo.starkStruct.nBits; i++) sn = Goldilocks::square(sn);
            for (uint64_t i = 0; i < zh.size(); i++) {
                zh[i] = Goldilocks::toU64(Goldilocks::inv(Goldilocks::sub(Goldilocks::mul(sn, w), Goldilocks::one())));
                w = Goldilocks::mul(w, Goldilocks::w(extendBits));
            }
            d_zi = dalloc(zh.size());
            XLP_CHECK(xlp_copy_h2d(d_zi, zh.data(), zh.size() * sizeof(uint64_t)));
        }
        d_xdx = dalloc(NExtended * FIELD_EXTENSION); d_xdwx = dalloc(NExtended * FIELD_EXTENSION);
        d_challenges = dalloc(NUM_CHALLENGES * FIELD_EXTENSION);
        d_publics = dalloc(starkInfo.nPublics);
        d_evals = dalloc(starkInfo.evMap.size() * FIELD_EXTENSION);
        d_work = dalloc(NExtended * (uint64_t)std::max<uint64_t>(4 * FIELD_EXTENSION, starkInfo.qDim * (starkInfo.qDeg + 1)));
        // trees: cm1..cm4 over the extended sections + the constant tree (starks.hpp:186-190)
        const eSection secs[4] = {cm1_2ns, cm2_2ns, cm3_2ns, cm4_2ns};
        for (int i = 0; i < 4; i++) treesGL[i] = new MerkleTreeGL(NExtended, starkInfo.mapSectionsN.section[secs[i]], section(secs[i]), true);
        treesGL[4] = new MerkleTreeGL(NExtended, starkInfo.nConstants, d_const_2ns, true);
        treesGL[4]->merkelize();
        treesGL[4]->getRoot(constRoot);
    }
    ~Starks()
    {
        for (int i = 0; i < 5; i++) delete treesGL[i];
        uint64_t* all[] = {d_mem, d_const_n, d_const_2ns, d_x_n, d_x_2ns, d_zi, d_xdx, d_xdwx, d_challenges, d_publics, d_evals, d_work};
        for (uint64_t* p : all) xlp_dev_free(p);
    }
    Starks(const Starks&) = delete;
    Starks& operator=(const Starks&) = delete;

    // cm1_n: the committed witness, N rows x mapSectionsN[cm1_n] (what the executor leaves at pAddress + mapOffsets[cm1_n])
    void genProof(FRIProof& proof, Goldilocks::Element* publicInputs, Goldilocks::Element verkey[4], const Goldilocks::Element* cm1_n_host)
    {
        const uint64_t W1 = starkInfo.mapSectionsN.section[cm1_n];
        uint64_t numCommited = starkInfo.nCm1;
        Transcript transcript;
        std::vector<Goldilocks::Element> challenges(NUM_CHALLENGES * FIELD_EXTENSION, Goldilocks::zero());
        std::vector<Goldilocks::Element> evals(starkInfo.evMap.size() * FIELD_EXTENSION, Goldilocks::zero());
        Goldilocks::Element root0[HASH_SIZE], root1[HASH_SIZE], root2[HASH_SIZE], root3[HASH_SIZE];
        auto challenge = [&](int k) {
            transcript.getField(&challenges[k * FIELD_EXTENSION]);
            XLP_CHECK(xlp_copy_h2d(d_challenges, challenges.data(), challenges.size() * sizeof(uint64_t)));
        };
        XLP_CHECK(xlp_copy_h2d(section(cm1_n), cm1_n_host, N * W1 * sizeof(uint64_t)));
        XLP_CHECK(xlp_copy_h2d(d_publics, publicInputs, starkInfo.nPublics * sizeof(uint64_t)));
        transcript.put(&verkey[0], 4);
        transcript.put(&publicInputs[0], starkInfo.nPublics);
        xlp_steps_params P = params();
        //-------------------------------- 1.- Calculate p_cm1_2ns
        {
            Timer t("STARK_STEP_1_LDE_AND_MERKLETREE", logTimers);
            XLP_CHECK(xlp_dev_extend(section(cm1_2ns), section(cm1_n), NExtended, N, W1, NULL));
            treesGL[0]->merkelize();
            treesGL[0]->getRoot(root0);
            transcript.put(root0, HASH_SIZE);
        }
        //-------------------------------- 2.- Calculate plookups h1 and h2
        challenge(0);   // u
        challenge(1);   // defVal
        {
            Timer t("STARK_STEP_2_CALCULATE_EXPS", logTimers);
            steps.step2prev_parser_first_avx(P, N, 1);
        }
        {
            Timer t("STARK_STEP_2_CALCULATEH1H2", logTimers);
            for (uint64_t i = 0; i < starkInfo.puCtx.size(); i++) {
                const PolRef f = pol(exp2pol(starkInfo.puCtx[i].fExpId)), tt = pol(exp2pol(starkInfo.puCtx[i].tExpId));
                const PolRef h1 = pol(starkInfo.cm_n[numCommited + i * 2]), h2 = pol(starkInfo.cm_n[numCommited + i * 2 + 1]);
                if (f.dim != tt.dim || h1.dim != f.dim || h2.dim != f.dim) xlp_exit_process("Starks: plookup polynomials of different dimensions");
                uint64_t *df = d_work, *dt = d_work + N * 3, *dh1 = d_work + 2 * N * 3, *dh2 = d_work + 3 * N * 3;
                XLP_CHECK(xlp_dev_copy_cols(df, f.dim, f.p, f.stride, N, (uint32_t)f.dim));          // transposeH1H2Columns
                XLP_CHECK(xlp_dev_copy_cols(dt, f.dim, tt.p, tt.stride, N, (uint32_t)f.dim));
                XLP_CHECK(xlp_dev_calculate_h1h2(dh1, dh2, df, dt, N, (uint32_t)f.dim));
                XLP_CHECK(xlp_dev_copy_cols(h1.p, h1.stride, dh1, f.dim, N, (uint32_t)f.dim));       // transposeH1H2Rows
                XLP_CHECK(xlp_dev_copy_cols(h2.p, h2.stride, dh2, f.dim, N, (uint32_t)f.dim));
            }
            numCommited += starkInfo.puCtx.size() * 2;
        }
        {
            Timer t("STARK_STEP_2_LDE_AND_MERKLETREE", logTimers);
            XLP_CHECK(xlp_dev_extend(section(cm2_2ns), section(cm2_n), NExtended, N, starkInfo.mapSectionsN.section[cm2_n], NULL));
            treesGL[1]->merkelize();
            treesGL[1]->getRoot(root1);
            transcript.put(root1, HASH_SIZE);
        }
        //-------------------------------- 3.- Compute Z polynomials
        challenge(2);   // gamma
        challenge(3);   // betta
        {
            Timer t("STARK_STEP_3_CALCULATE_EXPS", logTimers);
            steps.step3prev_parser_first_avx(P, N, 1);
        }
        {
            Timer t("STARK_STEP_3_CALCULATE_Z", logTimers);
            std::vector<std::pair<uint64_t, uint64_t>> nd;       // (numId, denId) of puCtx, peCtx, ciCtx in the reference's order
            for (const PuCtx& c : starkInfo.puCtx) nd.emplace_back(c.numId, c.denId);
            for (const PeCtx& c : starkInfo.peCtx) nd.emplace_back(c.numId, c.denId);
            for (const CiCtx& c : starkInfo.ciCtx) nd.emplace_back(c.numId, c.denId);
            for (uint64_t i = 0; i < nd.size(); i++) {
                const PolRef num = pol(exp2pol(nd[i].first)), den = pol(exp2pol(nd[i].second)), z = pol(starkInfo.cm_n[numCommited + i]);
                if (num.dim != 3 || den.dim != 3 || z.dim != 3) xlp_exit_process("Starks: Z polynomials must be ext-3");
                uint64_t *dn = d_work, *dd = d_work + N * 3, *dz = d_work + 2 * N * 3;
                XLP_CHECK(xlp_dev_copy_cols(dn, 3, num.p, num.stride, N, 3));
                XLP_CHECK(xlp_dev_copy_cols(dd, 3, den.p, den.stride, N, 3));
                XLP_CHECK(xlp_dev_calculate_z(dz, dn, dd, N));
                XLP_CHECK(xlp_dev_copy_cols(z.p, z.stride, dz, 3, N, 3));
            }
        }
        {
            Timer t("STARK_STEP_3_CALCULATE_EXPS_2", logTimers);
            steps.step3_parser_first_avx(P, N, 1);
        }
        {
            Timer t("STARK_STEP_3_LDE_AND_MERKLETREE", logTimers);
            XLP_CHECK(xlp_dev_extend(section(cm3_2ns), section(cm3_n), NExtended, N, starkInfo.mapSectionsN.section[cm3_n], NULL));
            treesGL[2]->merkelize();
            treesGL[2]->getRoot(root2);
            transcript.put(root2, HASH_SIZE);
        }
        //-------------------------------- 4. Compute C Polynomial
        challenge(4);
        {
            Timer t("STARK_STEP_4_CALCULATE_EXPS_2NS", logTimers);
            steps.step42ns_parser_first_avx(P, NExtended, 1);
        }
        {
            Timer t("STARK_STEP_4_CALCULATE_EXPS_2NS_INTT_MUL_NTT", logTimers);
            uint64_t *qq1 = d_work, *qq2 = d_work + NExtended * starkInfo.qDim;
            XLP_CHECK(xlp_dev_ntt(qq1, section(q_2ns), NExtended, starkInfo.qDim, 1));
            const Goldilocks::Element shiftIn = Goldilocks::exp(Goldilocks::inv(Goldilocks::shift()), N);
            XLP_CHECK(xlp_dev_q_split(qq2, qq1, N, NExtended, (uint32_t)starkInfo.qDeg, (uint32_t)starkInfo.qDim, Goldilocks::toU64(shiftIn)));
            XLP_CHECK(xlp_dev_ntt(section(cm4_2ns), qq2, NExtended, starkInfo.qDim * starkInfo.qDeg, 0));
        }
        {
            Timer t("STARK_STEP_4_MERKLETREE", logTimers);
            treesGL[3]->merkelize();
            treesGL[3]->getRoot(root3);
            transcript.put(root3, HASH_SIZE);
        }
        //-------------------------------- 5. Compute FRI Polynomial
        challenge(7);   // xi
        const Goldilocks::Element* xi = &challenges[7 * FIELD_EXTENSION];
        uint64_t wxi[3];
        {
            Timer t("STARK_STEP_5_LEv_LpEv", logTimers);
            const Goldilocks::Element sinv = Goldilocks::inv(Goldilocks::shift()), wN = Goldilocks::w(starkInfo.starkStruct.nBits);
            uint64_t xis[3], wxis[3];
            const uint64_t first[3] = {1, 0, 0};
            for (int c = 0; c < 3; c++) {
                xis[c] = Goldilocks::toU64(Goldilocks::mul(xi[c], sinv));
                wxi[c] = Goldilocks::toU64(Goldilocks::mul(xi[c], wN));
                wxis[c] = Goldilocks::toU64(Goldilocks::mul(Goldilocks::mul(xi[c], wN), sinv));
            }
            uint64_t *LEv = d_work, *LpEv = d_work + N * 3;
            XLP_CHECK(xlp_dev_geometric(LEv, N, 3, first, xis));
            XLP_CHECK(xlp_dev_geometric(LpEv, N, 3, first, wxis));
            XLP_CHECK(xlp_dev_ntt(LEv, LEv, N, 3, 1));
            XLP_CHECK(xlp_dev_ntt(LpEv, LpEv, N, 3, 1));
        }
        {
            Timer t("STARK_STEP_5_EVMAP", logTimers);
            std::vector<xlp_evmap_entry> ents(starkInfo.evMap.size());
            for (size_t i = 0; i < ents.size(); i++) {
                const EvMap& ev = starkInfo.evMap[i];
                if (ev.type == EvMap::_const) ents[i] = {d_const_2ns + ev.id, starkInfo.nConstants, 1, ev.prime ? 1u : 0u};
                else {
                    const PolRef p = pol(ev.type == EvMap::cm ? starkInfo.cm_2ns[ev.id] : starkInfo.qs[ev.id]);
                    ents[i] = {p.p, p.stride, (uint32_t)p.dim, ev.prime ? 1u : 0u};
                }
            }
            XLP_CHECK(xlp_dev_evmap(d_evals, d_work, d_work + N * 3, N, (uint32_t)extendBits, (uint32_t)ents.size(), ents.data()));
            XLP_CHECK(xlp_copy_d2h(evals.data(), d_evals, evals.size() * sizeof(uint64_t)));
        }
        for (uint64_t i = 0; i < starkInfo.evMap.size(); i++) transcript.put(&evals[i * FIELD_EXTENSION], 3);
        challenge(5);   // v1
        challenge(6);   // v2
        {
            Timer t("STARK_STEP_5_XDIVXSUB", logTimers);
            XLP_CHECK(xlp_dev_x_div_x_sub_xi(d_xdx, d_x_2ns, xlp_raw(xi), NExtended));
            XLP_CHECK(xlp_dev_x_div_x_sub_xi(d_xdwx, d_x_2ns, wxi, NExtended));
        }
        {
            Timer t("STARK_STEP_5_CALCULATE_EXPS", logTimers);
            steps.step52ns_parser_first_avx(P, NExtended, 1);
        }
        {
            Timer t("STARK_STEP_FRI", logTimers);
            Polinomial friPol(NExtended >> (starkInfo.starkStruct.nBitsExt - starkInfo.starkStruct.steps.back().nBits), FIELD_EXTENSION, "friPol");
            FRIProve::prove(proof, treesGL, transcript, friPol, starkInfo.starkStruct.nBitsExt, starkInfo, section(f_2ns));
        }
        proof.proofs.setEvals(evals.data());
        std::memcpy(&proof.proofs.root1[0], root0, sizeof root0);
        std::memcpy(&proof.proofs.root2[0], root1, sizeof root1);
        std::memcpy(&proof.proofs.root3[0], root2, sizeof root2);
        std::memcpy(&proof.proofs.root4[0], root3, sizeof root3);
    }
};
