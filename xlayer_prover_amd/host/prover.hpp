// `Prover::genBatchProof` (src/prover/prover.cpp:392-669) over the GPU-resident Starks: the three STARKs of a batch proof --
// zkEVM (2^23 rows), c12a, recursive1 -- chained through proof2zkinStark exactly as prover.cpp:572-651 chains them.  What the
// reference does on the CPU between and before them is injected by the caller, because it is outside this path:
//   * the executor that fills the committed polynomials at pAddress (prover.cpp:435)            -> ProverRequest::executor
//   * the circom witness calculators that turn a zkin proof into the next STARK's committed polynomials
//     (Circom::getCommitedPols prover.cpp:593, CircomRecursive1::getCommitedPols prover.cpp:630)  -> ProverRequest::c12aWitness / recursive1Witness
// Timer names are the reference's.  pAddress is page-locked host memory (xlp_host_alloc) sized for the widest witness.
// ROW-SHARDED (one process per GPU, xlp_comm_init before the Prover is built, a window bundle for the zkEVM STARK): every rank proves its rows
// of the zkEVM STARK (98 % of a batch proof's GPU time; host/starks.hpp genProofSharded) and ends with the same proof; the executor fills
// ProverRequest::firstRow .. + degree of the witness on each rank; the publics come from the first row (rank 0) and the last row (last rank);
// c12a and recursive1 (2^20 / 2^17 rows, 0.1 s together) run on rank 0 alone, which is the rank that returns the batch proof.
#pragma once
#include <functional>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include "starks.hpp"
#include "starkRecursiveF.hpp"
#include "fflonk_prover.hpp"
#include "fflonk_bench_key.hpp"

class ProverRequest
{
public:
    typedef std::function<void(Goldilocks::Element* cmPols, uint64_t degree, uint64_t nCols)> Executor;
    typedef std::function<void(const std::string& zkin, Goldilocks::Element* cmPols, uint64_t degree, uint64_t nCols)> WitnessCalculator;
    std::string uuid = "00000000-0000-0000-0000-000000000000", timestamp = "0", filePrefix;
    Executor executor;
    WitnessCalculator c12aWitness, recursive1Witness, recursive2Witness, recursivefWitness;
    std::string aggregatedProofInput1, aggregatedProofInput2, aggregatedProofOutput;     // zkin JSON texts (prover.cpp:674-783)
    std::string finalProofInput, recursivefProofOutput;                                  // prover.cpp:836-885
    // the final circuit's witness from the recursivef zkin (CircomFinal::loadJsonImpl + getBinWitness, prover.cpp:897-912: generated circom
    // code, host work): values as in a .wtns, NOT Montgomery.  With it and config.finalStarkZkey, genFinalProof ends with the fflonk proof
    std::function<void(const std::string& zkinRecursiveF, std::vector<FrElement>& witness)> finalWitness;
    std::string finalProofOutput, finalPublicsOutput;                                    // prover.cpp:931-945: the SNARK proof and its public signals (JSON)
    std::string aggregatorAddress = "0";
    int result = 1;                                  // zkresult numbering of src/config/zkresult.hpp:9-19: ZKR_SUCCESS = 1, ZKR_AGGREGATED_PROOF_INVALID_INPUT = 9
    std::string batchProofOutput;                    // zkin of the recursive1 proof + publics: the input of genAggregatedProof (prover.cpp:653)
    std::string publicsJson, proofJson;
    std::vector<std::pair<std::string, double>> timers;
    double witnessUploadMs = 0.0;
    // row-sharded zkEVM STARK: what each timed stage spent in xlp_comm_exchange (when xlp_comm_stats timing is on), and whether every rank
    // ended with the same zkEVM proof text (an all-gather of its FNV-1a hash): -1 not sharded, 0 no, 1 yes
    std::vector<Starks::StageExchange> exchanges;
    int zkevmProofSameOnAllRanks = -1;
    uint64_t firstRow = 0;                           // row-sharded: the global row the executor's `degree` rows start at (set by genBatchProof)
    std::string publicsOutputFile() const { return filePrefix + "batch_proof.public.json"; }
};

class Prover
{
    Goldilocks& fr;
    PoseidonGoldilocks& poseidon;
    const Config& config;
    void* pAddress = NULL;
    uint64_t polsSize = 0;
    int rank = 0;
    uint64_t nPublicsBatch = 0;
    struct Tm {
        ProverRequest* r; const char* name; std::chrono::steady_clock::time_point t0;
        Tm(ProverRequest* r_, const char* n) : r(r_), name(n), t0(std::chrono::steady_clock::now()) {}
        ~Tm() { r->timers.emplace_back(name, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()); }
    };
    static std::string dirOf(const std::string& f) { const size_t p = f.rfind('/'); return p == std::string::npos ? "." : f.substr(0, p); }
    static bool fileExists(const std::string& f) { struct stat st; return stat(f.c_str(), &st) == 0; }
    static void verkey(const std::string& file, Starks* stark, Goldilocks::Element out[4])
    {
        // <stark>.verkey.json {"constRoot": [..4..]} (prover.cpp:466-491); the GPU build computes the constant tree itself when the
        // file is absent, and then its root is the key
        if (fileExists(file)) {
            const xlp_json::Value j = xlp_json::Parser(Config::xlp_read_file(file)).value();
            for (int i = 0; i < 4; i++) out[i] = Goldilocks::fromU64(j["constRoot"][i]);
        } else for (int i = 0; i < 4; i++) out[i] = stark->constRoot[i];
    }
    static std::string withMembers(const std::string& zkin, const std::string& extra) { return zkin.substr(0, zkin.size() - 1) + extra + "}"; }
    static void json2file(const std::string& text, const std::string& file)
    {
        FILE* f = fopen(file.c_str(), "wb");
        if (!f) xlp_exit_process(("cannot write " + file).c_str());
        fwrite(text.data(), 1, text.size(), f); fclose(f);
    }

public:
    Starks *starkZkevm = NULL, *starksC12a = NULL, *starksRecursive1 = NULL, *starksRecursive2 = NULL;
    StarkRecursiveF* starksRecursiveF = NULL;
    Fflonk::FflonkProver* prover = NULL;             // prover.cpp:117-118
    std::vector<FrElement> syntheticFinalWitness;    // the witness that goes with a "synthetic:<power>" finalStarkZkey
    std::unique_ptr<BinFileUtils::BinFile> zkey;     // prover.cpp:52
    ZkevmSteps* zkevmSteps = NULL;
    C12aSteps* c12aSteps = NULL;
    Recursive1Steps* recursive1Steps = NULL;
    Recursive2Steps* recursive2Steps = NULL;
    StarkRecursiveFSteps* recursivefSteps = NULL;

    Prover(Goldilocks& fr, PoseidonGoldilocks& poseidon, const Config& config) : fr(fr), poseidon(poseidon), config(config)
    {
        if (!config.generateProof()) return;
        XLP_CHECK(xlp_init((int)config.xlpDevice));
        // the host slab: only the committed polynomials of the STARK being proven live on the host (prover.cpp:94-115 allocates the
        // whole 350 GB slab there; here everything past cm1_n is HBM)
        StarkInfo z(config.zkevmStarkInfo), c(config.c12aStarkInfo), r(config.recursive1StarkInfo);
        // (a window starkinfo: the rank's rows only)
        auto cm1Bytes = [](const StarkInfo& s) { return (s.mapOffsets.section[cm1_n] + ((1ULL << s.starkStruct.nBits) / (s.xlpWorld ? s.xlpWorld : 1)) * s.mapSectionsN.section[cm1_n]) * sizeof(Goldilocks::Element); };
        rank = xlp_comm_rank();
        nPublicsBatch = std::max(r.nPublics, z.nPublics);
        polsSize = std::max(cm1Bytes(z), std::max(cm1Bytes(c), cm1Bytes(r)));
        // the aggregation / final STARKs exist when their starkinfo is configured and present (prover.cpp:128-160 creates all five)
        const bool haveR2 = config.recursive2StarkInfo.size() && fileExists(config.recursive2StarkInfo);
        const bool haveRF = config.recursivefStarkInfo.size() && fileExists(config.recursivefStarkInfo);
        if (haveR2) polsSize = std::max<uint64_t>(polsSize, cm1Bytes(StarkInfo(config.recursive2StarkInfo)));
        if (haveRF) polsSize = std::max<uint64_t>(polsSize, cm1Bytes(StarkInfo(config.recursivefStarkInfo)));
        XLP_CHECK(xlp_host_alloc(&pAddress, polsSize));
        starkZkevm = new Starks(config, {config.zkevmConstPols, config.mapConstPolsFile, config.zkevmConstantsTree, config.zkevmStarkInfo}, pAddress);
        zkevmSteps = new ZkevmSteps(dirOf(config.zkevmStarkInfo));
        if (rank != 0) return;                             // the small STARKs run on rank 0 alone
        starksC12a = new Starks(config, {config.c12aConstPols, config.mapConstPolsFile, config.c12aConstantsTree, config.c12aStarkInfo}, pAddress);
        starksRecursive1 = new Starks(config, {config.recursive1ConstPols, config.mapConstPolsFile, config.recursive1ConstantsTree, config.recursive1StarkInfo}, pAddress);
        c12aSteps = new C12aSteps(dirOf(config.c12aStarkInfo));
        recursive1Steps = new Recursive1Steps(dirOf(config.recursive1StarkInfo));
        if (haveR2) {
            starksRecursive2 = new Starks(config, {config.recursive2ConstPols, config.mapConstPolsFile, config.recursive2ConstantsTree, config.recursive2StarkInfo}, pAddress);
            recursive2Steps = new Recursive2Steps(dirOf(config.recursive2StarkInfo));
        }
        if (haveRF) {
            starksRecursiveF = new StarkRecursiveF(config, {config.recursivefConstPols, config.mapConstPolsFile, config.recursivefConstantsTree, config.recursivefStarkInfo}, pAddress);
            recursivefSteps = new StarkRecursiveFSteps(dirOf(config.recursivefStarkInfo));
        }
        if (config.finalStarkZkey.size()) {               // prover.cpp:52-53,117-118: the key goes to the GPU once (fflonk only: the Groth16 branch is not built)
            // "synthetic:<power>": a key of 2^power rows built in memory (host/fflonk_bench_key.hpp; like "synthetic:<seed>" constant
            // polynomials, for runs without the reference's 75 GB archive) with its witness in syntheticFinalWitness
            if (config.finalStarkZkey.rfind("synthetic:", 0) == 0) {
                const uint32_t power = (uint32_t)strtoul(config.finalStarkZkey.c_str() + 10, NULL, 0);
                if (power < 4 || power > 26) xlp_exit_process("Prover: finalStarkZkey synthetic:<power> with 4 <= power <= 26");
                zkey.reset(new BinFileUtils::BinFile(FflonkBench::build(power, syntheticFinalWitness), "zkey", 1));
            } else
            zkey = BinFileUtils::openExisting(config.finalStarkZkey, "zkey", 1);
            if (Zkey::getProtocolIdFromZkey(zkey.get()) != Zkey::FFLONK_PROTOCOL_ID) xlp_exit_process("Prover: config.finalStarkZkey is not an fflonk key");
            // prover.cpp:117: the STARKs' polynomial memory is the prover's reserved memory -- here the DEVICE slab (the zkEVM slab fills the card:
            // nothing of the fflonk prover may stay in HBM between final proofs); the key is uploaded from the host image at every proof
            prover = new Fflonk::FflonkProver([] { return std::make_pair((void*)XlpDeviceSlab::get().address(), XlpDeviceSlab::get().capacityWords() * 8); });
            zkey->pin();
            prover->setZkey(zkey.get());
        }
    }
    ~Prover()
    {
        delete prover;
        delete zkevmSteps; delete c12aSteps; delete recursive1Steps; delete recursive2Steps; delete recursivefSteps;
        delete starkZkevm; delete starksC12a; delete starksRecursive1; delete starksRecursive2; delete starksRecursiveF;
        if (pAddress) xlp_host_free(pAddress);
        XlpDeviceSlab::get().release();
    }
    Prover(const Prover&) = delete;
    Prover& operator=(const Prover&) = delete;
    void* address() { return pAddress; }

    void genBatchProof(ProverRequest* pProverRequest)
    {
        if (!config.generateProof() || pProverRequest == NULL) xlp_exit_process("Prover::genBatchProof: not configured to generate proofs");
        ProverRequest& req = *pProverRequest;
        req.timers.clear();
        Tm total(&req, "PROVER_BATCH_PROOF");
        Goldilocks::Element* cmPols = (Goldilocks::Element*)pAddress;
        const uint64_t Nz = starkZkevm->localRows(), Wz = starkZkevm->starkInfo.mapSectionsN.section[cm1_n];      // (sharded: my rows)
        const bool shardedZ = starkZkevm->isSharded();
        const int world = shardedZ ? xlp_comm_world() : 1;
        {
            Tm t(&req, "EXECUTOR_EXECUTE_BATCH_PROOF");
            if (!req.executor) xlp_exit_process("Prover::genBatchProof: no executor");
            req.firstRow = starkZkevm->firstRow();
            req.executor(cmPols, Nz, Wz);                    // executor.execute(*pProverRequest, cmPols), prover.cpp:435
        }
        // prover.cpp:452-457: the committed polynomials as the executor left them -> zkevm.commit (row-major, pilDegree x nCommitted words);
        // a row-sharded rank writes its rows at their offset of the same file
        // (no rank truncates: the ranks' executors run concurrently, and a later "wb" open would drop rows another rank has written.  Every rank opens
        // the file O_CREAT | O_RDWR and pwrites at its offset; rank 0 cuts it to the full size -- a shorter file of an earlier run grows, a longer one shrinks)
        if (config.zkevmCmPolsAfterExecutor.size()) {
            const std::string& fn = config.zkevmCmPolsAfterExecutor;
            const int fd = ::open(fn.c_str(), O_CREAT | O_RDWR, 0644);
            bool ok = fd >= 0;
            const size_t rowBytes = Wz * sizeof(Goldilocks::Element);
            if (ok && req.firstRow == 0) ok = ::ftruncate(fd, (off_t)((uint64_t)world * Nz * rowBytes)) == 0;
            const char* src = (const char*)cmPols;
            for (size_t done = 0, total = Nz * rowBytes; ok && done < total;) {
                const ssize_t w = ::pwrite(fd, src + done, std::min<size_t>(total - done, (size_t)1 << 30), (off_t)(req.firstRow * rowBytes + done));
                if (w <= 0) ok = false; else done += (size_t)w;
            }
            if (fd >= 0 && ::close(fd)) ok = false;
            if (!ok) xlp_exit_process(("Prover::genBatchProof: cannot write " + fn).c_str());
        }
        if (req.result != ZKR_SUCCESS) return;           // prover.cpp:437: the executor reports through the request
        // ---- publics (prover.cpp:494-559): first / last row of Main's registers; recursive2's verification key behind them
        const uint64_t nPublics = nPublicsBatch;              // recursive1's count (prover.cpp:494: the publics of the batch proof; known on every rank)
        std::vector<Goldilocks::Element> publics(nPublics, Goldilocks::zero());
        Goldilocks::Element zkevmVerkey[4], c12aVerkey[4], recursive1Verkey[4], recursive2Verkey[4] = {{0}, {0}, {0}, {0}};
        std::string publicStarkJson;
        {
            Tm t(&req, "SAVE_PUBLICS_JSON_BATCH_PROOF");
            verkey(config.zkevmVerkey, starkZkevm, zkevmVerkey);
            if (starksC12a) verkey(config.c12aVerkey, starksC12a, c12aVerkey);
            if (starksRecursive1) verkey(config.recursive1Verkey, starksRecursive1, recursive1Verkey);
            if (fileExists(config.recursive2Verkey)) {
                const xlp_json::Value j = xlp_json::Parser(Config::xlp_read_file(config.recursive2Verkey)).value();
                for (int i = 0; i < 4; i++) recursive2Verkey[i] = Goldilocks::fromU64(j["constRoot"][i]);
            }
            const uint64_t lastN = Nz - 1;
            auto at = [&](uint64_t row, uint64_t col) { return cmPols[row * Wz + col]; };
            if (Wz > 617 && nPublics >= 48) {
                // column ids of the fork-9 Main state machine (src/main_sm/fork_9/pols_generated/commit_pols.hpp:1480-1540):
                // B_k = 581 - k, C_k = 589 - k, D_k = 597 - k, E_k = 605 - k, SR_k = 613 - k, CTX 614, SP 615, PC 616, GAS 617
                for (uint64_t k = 0; k < 8; k++) {
                    publics[k] = at(0, 581 - k);             // oldStateRoot      = B[0]
                    publics[8 + k] = at(0, 589 - k);         // oldAccInputHash   = C[0]
                    publics[19 + k] = at(lastN, 613 - k);    // newStateRoot      = SR[lastN]
                    publics[27 + k] = at(lastN, 597 - k);    // newAccInputHash   = D[lastN]
                    publics[35 + k] = at(lastN, 605 - k);    // localExitRoot     = E[lastN]
                }
                publics[16] = at(0, 615);                    // oldBatchNum = SP[0]
                publics[17] = at(0, 617);                    // chainId     = GAS[0]
                publics[18] = at(0, 614);                    // forkid      = CTX[0]
                publics[43] = at(lastN, 616);                // newBatchNum = PC[lastN]
                for (int i = 0; i < 4; i++) publics[44 + i] = recursive2Verkey[i];
            } else {
                for (uint64_t i = 0; i < nPublics && i < Wz; i++) publics[i] = at(0, i);      // not the zkEVM PIL: first row
            }
            for (Goldilocks::Element& p : publics) p = Goldilocks::fromU64(Goldilocks::toU64(p));
            if (shardedZ && world > 1) {
                // the first row lives on rank 0, the last one on the last rank: publics 0..18 are first-row registers, 19..43 last-row ones
                std::vector<uint64_t> mine(nPublics), all((size_t)world * nPublics);
                for (uint64_t i = 0; i < nPublics; i++) mine[i] = Goldilocks::toU64(publics[i]);
                XLP_CHECK(xlp_comm_all_gather_host(all.data(), mine.data(), nPublics));
                const bool zkevmPil = Wz > 617 && nPublics >= 48;
                for (uint64_t i = 0; i < nPublics; i++) {
                    const int from = zkevmPil && i >= 19 && i < 44 ? world - 1 : 0;
                    publics[i] = Goldilocks::fromU64(all[(size_t)from * nPublics + i]);
                }
            }
            std::ostringstream os;
            xlp_json::arr(os, publics);
            publicStarkJson = os.str();
        }
        auto newProof = [](Starks* s) {
            const StarkStruct& ss = s->starkInfo.starkStruct;
            return new FRIProof(1ULL << ss.steps.back().nBits, FIELD_EXTENSION, ss.steps.size(), s->starkInfo.evMap.size(), s->starkInfo.nPublics);
        };
        auto collect = [&](Starks* s, const char* tag) {
            for (const auto& kv : s->lastTimers) req.timers.emplace_back(std::string(tag) + "." + kv.first, kv.second);
        };
        // ---- zkEVM STARK (prover.cpp:572-579)
        std::string zkin;
        {
            FRIProof* fproof = newProof(starkZkevm);
            {
                Tm t(&req, "STARK_PROOF_BATCH_PROOF");
                starkZkevm->genProof(*fproof, publics.data(), zkevmVerkey, zkevmSteps);
            }
            req.witnessUploadMs = starkZkevm->lastUploadMs;
            collect(starkZkevm, "zkevm");
            Tm t(&req, "STARK_JSON_GENERATION_BATCH_PROOF");
            zkin = withMembers(proof2zkinStark(*fproof), ",\"publics\":" + publicStarkJson);
            delete fproof;
            req.exchanges = starkZkevm->lastExchanges;
            req.zkevmProofSameOnAllRanks = -1;
            if (starkZkevm->isSharded()) {
                uint64_t h = 0xcbf29ce484222325ULL;
                for (unsigned char c : zkin) { h ^= c; h *= 0x100000001b3ULL; }
                std::vector<uint64_t> all((size_t)xlp_comm_world());
                XLP_CHECK(xlp_comm_all_gather_host(all.data(), &h, 1));
                req.zkevmProofSameOnAllRanks = 1;
                for (uint64_t v : all) if (v != h) req.zkevmProofSameOnAllRanks = 0;
            }
        }
        if (rank != 0) {                                     // row-sharded: c12a / recursive1 are rank 0's (0.1 s); the others are done
            req.publicsJson = publicStarkJson;
            req.batchProofOutput.clear();
            return;
        }
        // ---- c12a (prover.cpp:581-612)
        std::string zkinC12a;
        {
            {
                Tm t(&req, "STARK_GEN_AND_CALC_WITNESS_C12A");
                if (!req.c12aWitness) xlp_exit_process("Prover::genBatchProof: no c12a witness calculator");
                req.c12aWitness(zkin, cmPols, starksC12a->domainSize(), starksC12a->starkInfo.nCm1);
            }
            FRIProof* fproof = newProof(starksC12a);
            {
                Tm t(&req, "STARK_C12_A_PROOF_BATCH_PROOF");
                starksC12a->genProof(*fproof, publics.data(), c12aVerkey, c12aSteps);
            }
            collect(starksC12a, "c12a");
            Tm t(&req, "STARK_JSON_GENERATION_BATCH_PROOF_C12A");
            std::ostringstream rc;
            rc << ",\"publics\":" << publicStarkJson << ",\"rootC\":[";
            for (int i = 0; i < 4; i++) rc << (i ? "," : "") << "\"" << Goldilocks::toString(recursive2Verkey[i]) << "\"";
            rc << "]";
            zkinC12a = withMembers(proof2zkinStark(*fproof), rc.str());
            delete fproof;
        }
        // ---- recursive1 (prover.cpp:629-645)
        {
            {
                Tm t(&req, "STARK_GEN_AND_CALC_WITNESS_RECURSIVE1");
                if (!req.recursive1Witness) xlp_exit_process("Prover::genBatchProof: no recursive1 witness calculator");
                req.recursive1Witness(zkinC12a, cmPols, starksRecursive1->domainSize(), starksRecursive1->starkInfo.nCm1);
            }
            FRIProof* fproof = newProof(starksRecursive1);
            {
                Tm t(&req, "STARK_RECURSIVE_1_PROOF_BATCH_PROOF");
                starksRecursive1->genProof(*fproof, publics.data(), recursive1Verkey, recursive1Steps);
            }
            collect(starksRecursive1, "recursive1");
            Tm t(&req, "SAVE_PROOF");
            req.batchProofOutput = withMembers(proof2zkinStark(*fproof), ",\"publics\":" + publicStarkJson);
            req.publicsJson = publicStarkJson;
            if (config.saveOutputToFile) json2file(req.batchProofOutput, req.filePrefix + "batch_proof.output.json");
            if (config.saveProofToFile) json2file(req.batchProofOutput, req.filePrefix + "batch_proof.proof.json");
            if (config.saveOutputToFile || config.saveProofToFile) json2file(publicStarkJson, req.publicsOutputFile());
            delete fproof;
        }
    }
    // The input checks of Prover::genAggregatedProof (prover.cpp:695-728): the second proof must continue the first -- same chainId
    // (publics[17]) and forkId ([18]); the first proof's newStateRoot ([19..26]) and newAccInputHash ([27..34]) are the second's
    // oldStateRoot ([0..7]) and oldAccInputHash ([8..15]); its newBatchNum ([43]) is the second's oldBatchNum ([16]).  Checked in the
    // reference's order, `why` names the first failure like its log line does.
    enum { ZKR_SUCCESS = 1, ZKR_AGGREGATED_PROOF_INVALID_INPUT = 9 };      // src/config/zkresult.hpp:11,19
    static bool aggregatedInputsAreConsecutive(const xlp_json::Value& in1, const xlp_json::Value& in2, std::string* why = NULL)
    {
        auto same = [&](size_t a, size_t b) { return in1["publics"][a].str() == in2["publics"][b].str(); };
        auto fail = [&](const char* what, size_t a, size_t b) {
            if (why) *why = std::string(what) + " " + in1["publics"][a].str() + "!=" + in2["publics"][b].str();
            return false;
        };
        if (!same(17, 17)) return fail("Inputs has different chainId", 17, 17);
        if (!same(18, 18)) return fail("Inputs has different forkId", 18, 18);
        for (size_t i = 0; i < 8; i++) if (!same(19 + i, i)) return fail("The newStateRoot and the oldStateRoot are not consistent", 19 + i, i);
        for (size_t i = 0; i < 8; i++) if (!same(27 + i, 8 + i)) return fail("newAccInputHash and oldAccInputHash are not consistent", 27 + i, 8 + i);
        if (!same(43, 16)) return fail("newBatchNum and oldBatchNum are not consistent", 43, 16);
        return true;
    }
    // Prover::genAggregatedProof (prover.cpp:674-783): two batch (or aggregated) proofs -> joinzkin -> recursive2 STARK
    void genAggregatedProof(ProverRequest* pProverRequest)
    {
        if (!config.generateProof() || pProverRequest == NULL || !starksRecursive2) xlp_exit_process("Prover::genAggregatedProof: recursive2 is not configured");
        ProverRequest& req = *pProverRequest;
        req.timers.clear();
        Tm total(&req, "PROVER_AGGREGATED_PROOF");
        const xlp_json::Value in1 = xlp_json::Parser(req.aggregatedProofInput1).value(), in2 = xlp_json::Parser(req.aggregatedProofInput2).value();
        std::string why;
        if (!aggregatedInputsAreConsecutive(in1, in2, &why)) {
            fprintf(stderr, "Prover::genAggregatedProof() %s\n", why.c_str());
            req.result = ZKR_AGGREGATED_PROOF_INVALID_INPUT;
            return;
        }
        Goldilocks::Element recursive2Verkey[4];
        verkey(config.recursive2Verkey, starksRecursive2, recursive2Verkey);
        std::string vk = "{\"constRoot\":[";
        for (int i = 0; i < 4; i++) vk += (i ? "," : "") + Goldilocks::toString(recursive2Verkey[i]);
        vk += "]}";
        const std::string zkinInputRecursive2 = joinzkin(in1, in2, xlp_json::Parser(vk).value(), starksRecursive2->starkInfo.starkStruct.steps.size());
        const xlp_json::Value joined = xlp_json::Parser(zkinInputRecursive2).value();
        const uint64_t nPublics = starksRecursive2->starkInfo.nPublics;
        std::vector<Goldilocks::Element> publics(nPublics, Goldilocks::zero());
        for (uint64_t i = 0; i < 44 && i < nPublics; i++) publics[i] = Goldilocks::fromU64(strtoull(joined["publics"][i].str().c_str(), NULL, 10));
        for (uint64_t i = 0; i < 4 && 44 + i < nPublics; i++) publics[44 + i] = recursive2Verkey[i];
        std::ostringstream pj;
        xlp_json::arr(pj, publics);
        Goldilocks::Element* cmPols = (Goldilocks::Element*)pAddress;
        {
            Tm t(&req, "STARK_GEN_AND_CALC_WITNESS_RECURSIVE2");
            if (!req.recursive2Witness) xlp_exit_process("Prover::genAggregatedProof: no recursive2 witness calculator");
            req.recursive2Witness(zkinInputRecursive2, cmPols, starksRecursive2->domainSize(), starksRecursive2->starkInfo.nCm1);
        }
        const StarkStruct& ss = starksRecursive2->starkInfo.starkStruct;
        FRIProof fproof(1ULL << ss.steps.back().nBits, FIELD_EXTENSION, ss.steps.size(), starksRecursive2->starkInfo.evMap.size(), nPublics);
        {
            Tm t(&req, "STARK_RECURSIVE_2_PROOF_BATCH_PROOF");
            starksRecursive2->genProof(fproof, publics.data(), recursive2Verkey, recursive2Steps);
        }
        for (const auto& kv : starksRecursive2->lastTimers) req.timers.emplace_back("recursive2." + kv.first, kv.second);
        req.aggregatedProofOutput = withMembers(proof2zkinStark(fproof), ",\"publics\":" + pj.str());
        req.publicsJson = pj.str();
        if (config.saveOutputToFile) json2file(req.aggregatedProofOutput, req.filePrefix + "aggregated_proof.output.json");
        req.result = ZKR_SUCCESS;
    }

    // Prover::genFinalProof (prover.cpp:836-960): aggregated proof -> StarkRecursiveF (BN128 trees and transcript) -> the final circuit's
    // witness -> the fflonk proof (RAPID_SNARK, prover.cpp:927-945).  The witness calculator of the final circuit is generated circom code
    // (CircomFinal, host work): it comes in as ProverRequest::finalWitness like the other witness calculators; without it, or without
    // config.finalStarkZkey, the request ends with the recursivef zkin the final circuit would consume.
    void genFinalProof(ProverRequest* pProverRequest)
    {
        if (!config.generateProof() || pProverRequest == NULL || !starksRecursiveF) xlp_exit_process("Prover::genFinalProof: recursivef is not configured");
        ProverRequest& req = *pProverRequest;
        req.timers.clear();
        Tm total(&req, "PROVER_FINAL_PROOF");
        const xlp_json::Value zkinFinal = xlp_json::Parser(req.finalProofInput).value();
        const uint64_t nPublics = starksRecursiveF->starkInfo.nPublics;
        std::vector<Goldilocks::Element> publics(nPublics, Goldilocks::zero());
        for (uint64_t i = 0; i < nPublics; i++) publics[i] = Goldilocks::fromU64(strtoull(zkinFinal["publics"][i].str().c_str(), NULL, 10));
        Goldilocks::Element* cmPols = (Goldilocks::Element*)pAddress;
        {
            Tm t(&req, "STARK_GEN_AND_CALC_WITNESS_RECURSIVEF");
            if (!req.recursivefWitness) xlp_exit_process("Prover::genFinalProof: no recursivef witness calculator");
            req.recursivefWitness(req.finalProofInput, cmPols, starksRecursiveF->domainSize(), starksRecursiveF->starkInfo.nCm1);
        }
        const StarkStruct& ss = starksRecursiveF->starkInfo.starkStruct;
        FRIProofC12 fproof(1ULL << ss.steps.back().nBits, FIELD_EXTENSION, ss.steps.size(), starksRecursiveF->starkInfo.evMap.size(), nPublics);
        {
            Tm t(&req, "STARK_RECURSIVE_F_PROOF_BATCH_PROOF");
            starksRecursiveF->genProof(fproof, publics.data(), recursivefSteps);
        }
        for (const auto& kv : starksRecursiveF->lastTimers) req.timers.emplace_back("recursivef." + kv.first, kv.second);
        std::ostringstream pj;
        xlp_json::arr(pj, publics);
        req.recursivefProofOutput = withMembers(proof2zkinStarkC12(fproof), ",\"publics\":" + pj.str() + ",\"aggregatorAddr\":\"" + req.aggregatorAddress + "\"");
        if (config.saveProofToFile) json2file(req.recursivefProofOutput, req.filePrefix + "recursivef.proof.json");
        if (prover && req.finalWitness) {
            std::vector<FrElement> witness;
            {
                Tm t(&req, "CIRCOM_GET_BIN_WITNESS_FINAL");
                req.finalWitness(req.recursivefProofOutput, witness);
            }
            Tm t(&req, "RAPID_SNARK");
            // the checks of prove(BinFile*) against the key -- field and length -- on what the witness calculator handed over (its field is BN254's
            // scalar field by construction; a --final-wtns file's prime is compared where it is read): a witness of another circuit ends with the
            // reference's "Invalid witness length" instead of an out-of-bounds read of the vector
            WtnsUtils::Header wh;
            wh.n8 = 32; wh.nVars = (uint32_t)witness.size(); memcpy(wh.prime, FrModulus::Q, 32);
            if (witness.size() > 0xffffffffULL) xlp_exit_process("Prover::genFinalProof: the witness has more than 2^32 values");
            const auto r = prover->prove(witness.data(), &wh);
            req.finalProofOutput = r.first; req.finalPublicsOutput = r.second;
            for (const auto& kv : prover->lastTimers) req.timers.emplace_back("fflonk." + kv.first, kv.second);
            if (config.saveProofToFile) json2file(req.finalProofOutput, req.filePrefix + "final_proof.proof.json");
        }
        req.result = ZKR_SUCCESS;
    }
};
