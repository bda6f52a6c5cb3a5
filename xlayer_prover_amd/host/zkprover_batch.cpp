// zkProver in file mode for a batch proof (src/main.cpp:83-112 runFileGenBatchProof -> Prover::genBatchProof) on the GPU build:
//     zkprover_batch -c config.json [--proofs K] [--seed S] [--sequential-upload] [--timers] [--e2e [--final-wtns FILE]] [--witness-file zkevm.commit]
//                    [--world G --rank r --backend rccl|shm --rendezvous <file | name> [--device d]]
// --world: this process is rank r of G (one process per GPU): the transport is set up before any other GPU work, the zkEVM
// STARK runs row-sharded (the config's zkevmStarkInfo must be the window bundle of that world, tools/make_stark_bundle.py --world G), rank 0
// runs c12a / recursive1 and prints the lines; every rank's executor stand-in fills ITS rows of the same synthetic witness.
// --e2e: BASELINE configs[4] minus the SNARK wrap -- two consecutive batch proofs, Prover::genAggregatedProof over them (joinzkin + the recursive2
// STARK) and Prover::genFinalProof up to the recursivef STARK (Poseidon-BN128 trees / transcript); needs recursive2 / recursivef in the config.
// The executor and the two circom witness calculators are CPU subsystems outside this path and their inputs
// (the 75 GB archive) are absent: they are replaced by stand-ins that leave a synthetic witness in host memory where the real
// ones would -- the executor's from --seed, the calculators' from a hash of the zkin proof they are given, so that every STARK
// depends on the previous one's proof as in prover.cpp:584-630.  Prints one JSON line per run and a summary.
#include <cstdio>
#include <cstring>
#include "prover.hpp"

static uint64_t fnv1a(const std::string& s)
{
    uint64_t h = 0xcbf29ce484222325ULL;
    for (unsigned char c : s) { h ^= c; h *= 0x100000001b3ULL; }
    return h;
}
// a synthetic committed-polynomial matrix, generated on the GPU (xlp_dev_fill) and left in host memory at cmPols
static void syntheticWitness(Goldilocks::Element* cmPols, uint64_t degree, uint64_t nCols, uint64_t seed, uint64_t firstRow = 0)
{
    const uint64_t total = degree * nCols, chunk = 1ULL << 28, start = firstRow * nCols;
    if (xlp_plan_active()) return;                     // plan mode (XLP_PLAN=1): this stand-in is not the product -- its device buffer stays out of the account, the witness stays zero
    void* d = NULL;
    XLP_CHECK(xlp_dev_alloc(&d, (total < chunk ? total : chunk) * 8));
    for (uint64_t o = 0; o < total; o += chunk) {
        const uint64_t n = total - o < chunk ? total - o : chunk;
        XLP_CHECK(xlp_dev_fill((uint64_t*)d, n, seed, start + o));
        XLP_CHECK(xlp_copy_d2h(cmPols + o, d, n * 8));
    }
    XLP_CHECK(xlp_dev_free(d));
}

int main(int argc, char** argv)
{
    std::string cfg;
    int proofs = 2;
    uint64_t seed = 0x5EED0401;
    bool sequential = false, timers = false, e2e = false;
    std::string finalWtns, witnessFile;
    int world = 0, rank = 0, device = -1;
    std::string backend, rendezvous;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "-c" && i + 1 < argc) cfg = argv[++i];
        else if (a == "--proofs" && i + 1 < argc) proofs = atoi(argv[++i]);
        else if (a == "--seed" && i + 1 < argc) seed = strtoull(argv[++i], NULL, 0);
        else if (a == "--sequential-upload") sequential = true;
        else if (a == "--timers") timers = true;
        else if (a == "--e2e") e2e = true;
        else if (a == "--final-wtns" && i + 1 < argc) finalWtns = argv[++i];
        else if (a == "--witness-file" && i + 1 < argc) witnessFile = argv[++i];
        else if (a == "--world" && i + 1 < argc) world = atoi(argv[++i]);
        else if (a == "--rank" && i + 1 < argc) rank = atoi(argv[++i]);
        else if (a == "--device" && i + 1 < argc) device = atoi(argv[++i]);
        else if (a == "--backend" && i + 1 < argc) backend = argv[++i];
        else if (a == "--rendezvous" && i + 1 < argc) rendezvous = argv[++i];
        else { fprintf(stderr, "usage: %s -c config.json [--proofs K] [--seed S] [--sequential-upload] [--timers]\n", argv[0]); return 2; }
    }
    if (cfg.empty()) { fprintf(stderr, "missing -c config.json\n"); return 2; }
    Config config;
    config.load(cfg);
    if (!config.runFileGenBatchProof) { fprintf(stderr, "config: runFileGenBatchProof is not set\n"); return 2; }
    if (device >= 0) config.xlpDevice = (uint64_t)device;
    if (world) {
        if (e2e) { fprintf(stderr, "--e2e runs on one GPU\n"); return 2; }
        XLP_CHECK(xlp_init((int)config.xlpDevice));           // one process per GPU: bind it, then the transport, before anything else
        XLP_CHECK(xlp_comm_init(rank, world, backend.empty() ? NULL : backend.c_str(), rendezvous.empty() ? NULL : rendezvous.c_str()));
    }
    Goldilocks fr;
    PoseidonGoldilocks poseidon;
    {
    const auto t0 = std::chrono::steady_clock::now();
    Prover prover(fr, poseidon, config);
    const double tSetup = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    prover.starkZkevm->pipelineWitness = !sequential;
    prover.starkZkevm->logTimers = timers;
    if (prover.starksC12a) prover.starksC12a->logTimers = prover.starksRecursive1->logTimers = timers;
    fprintf(stderr, "setup %.2f s ; steps: zkevm", tSetup);
    for (int k = 0; k < 5; k++) fprintf(stderr, " [%s]", prover.zkevmSteps->mode(k).c_str());
    fprintf(stderr, "\n");
    ProverRequest req;
    req.filePrefix = config.outputPath + "/";
    req.executor = [&](Goldilocks::Element* p, uint64_t n, uint64_t w) { syntheticWitness(p, n, w, seed, req.firstRow); };
    // --witness-file F: the executor's output comes from a zkevm.commit file (config.zkevmCmPolsAfterExecutor of an earlier run, or the
    // reference's own dump: row-major, pilDegree x nCommitted words): a rank reads its rows
    if (witnessFile.size())
        req.executor = [&](Goldilocks::Element* p, uint64_t n, uint64_t w) {
            FILE* f = fopen(witnessFile.c_str(), "rb");
            if (!f || fseek(f, (long)(req.firstRow * w * sizeof(Goldilocks::Element)), SEEK_SET) || fread(p, sizeof(Goldilocks::Element), n * w, f) != n * w)
                xlp_exit_process(("--witness-file: cannot read " + std::to_string(n) + " x " + std::to_string(w) + " words from " + witnessFile).c_str());
            fclose(f);
        };
    req.c12aWitness = [&](const std::string& zkin, Goldilocks::Element* p, uint64_t n, uint64_t w) { syntheticWitness(p, n, w, fnv1a(zkin)); };
    req.recursive1Witness = req.c12aWitness;
    req.recursive2Witness = req.c12aWitness;
    req.recursivefWitness = req.c12aWitness;
    if (e2e) {
        // two CONSECUTIVE batches: the second executor stand-in starts from the state the first one ended in (the registers the publics are
        // read from, prover.cpp:494-559), so that genAggregatedProof's continuity checks (prover.cpp:695-728) hold as they do for real batches
        auto stat = [&](const char* what, const ProverRequest& r, const std::string& out, const std::string& verkey = "") {
            std::ostringstream os;
            os << "{\"stage\":\"" << what << "\"," << (verkey.size() ? "\"verkey\":" + verkey + "," : std::string()) << "\"timers_s\":{";
            bool first = true;
            for (const auto& kv : r.timers) { os << (first ? "" : ",") << "\"" << kv.first << "\":" << kv.second; first = false; }
            os << "},\"output_bytes\":" << out.size() << ",\"output_fnv\":\"" << std::hex << fnv1a(out) << std::dec << "\"}";
            printf("%s\n", os.str().c_str()); fflush(stdout);
        };
        prover.genBatchProof(&req);
        const std::string proofA = req.batchProofOutput;
        stat("genBatchProof#1", req, proofA);
        const xlp_json::Value pa = xlp_json::Parser(req.publicsJson).value();
        auto pub = [&](size_t i) { return (uint64_t)strtoull(pa[i].str().c_str(), NULL, 10); };
        req.executor = [&](Goldilocks::Element* p, uint64_t n, uint64_t w) {
            syntheticWitness(p, n, w, seed + 1);
            if (w > 617)
                for (uint64_t k = 0; k < 8; k++) { p[581 - k] = Goldilocks::fromU64(pub(19 + k)); p[589 - k] = Goldilocks::fromU64(pub(27 + k)); }
            if (w > 617) { p[615] = Goldilocks::fromU64(pub(43)); p[617] = Goldilocks::fromU64(pub(17)); p[614] = Goldilocks::fromU64(pub(18)); }
        };
        prover.genBatchProof(&req);
        const std::string proofB = req.batchProofOutput;
        stat("genBatchProof#2", req, proofB);
        req.aggregatedProofInput1 = proofA; req.aggregatedProofInput2 = proofB;
        prover.genAggregatedProof(&req);
        if (req.result != Prover::ZKR_SUCCESS) { fprintf(stderr, "genAggregatedProof: invalid input (the two proofs are not consecutive)\n"); return 1; }
        {
            std::ostringstream vk;
            vk << "[";
            for (int i = 0; i < 4; i++) vk << (i ? "," : "") << "\"" << Goldilocks::toString(prover.starksRecursive2->constRoot[i]) << "\"";
            vk << "]";
            stat("genAggregatedProof", req, req.aggregatedProofOutput, vk.str());
        }
        req.finalProofInput = req.aggregatedProofOutput;
        // the final circuit's witness calculator (generated circom code in the reference): --final-wtns FILE reads a .wtns, a "synthetic:<power>"
        // finalStarkZkey brings its own witness; without either genFinalProof ends with the recursivef zkin
        if (finalWtns.size())
            req.finalWitness = [&](const std::string&, std::vector<FrElement>& w) {
                auto f = BinFileUtils::openExisting(finalWtns, "wtns", 2);
                auto h = WtnsUtils::loadHeader(f.get());
                if (memcmp(h->prime, FrModulus::Q, 32) != 0) xlp_exit_process("--final-wtns: the witness is not over BN254's scalar field");
                w.resize(h->nVars);
                if (f->getSectionSize(2) < (uint64_t)h->nVars * 32) xlp_exit_process("--final-wtns: data section too short");
                memcpy(w.data(), f->getSectionData(2), (size_t)h->nVars * 32);
            };
        else if (prover.syntheticFinalWitness.size())
            req.finalWitness = [&](const std::string&, std::vector<FrElement>& w) { w = prover.syntheticFinalWitness; };
        prover.genFinalProof(&req);
        stat("genFinalProof(recursivef)", req, req.recursivefProofOutput, "\"" + RawFr::field.toString(prover.starksRecursiveF->constRoot[0], 10) + "\"");
        if (req.finalProofOutput.size()) printf("{\"stage\":\"finalProof(fflonk)\",\"proof\":%s,\"publics\":%s}\n", req.finalProofOutput.c_str(), req.finalPublicsOutput.c_str());
        if (config.saveOutputToFile) {
            FILE* f = fopen((config.outputPath + "/aggregated_proof.output.json").c_str(), "wb"); if (f) { fwrite(req.aggregatedProofOutput.data(), 1, req.aggregatedProofOutput.size(), f); fclose(f); }
            f = fopen((config.outputPath + "/recursivef.zkin.json").c_str(), "wb"); if (f) { fwrite(req.recursivefProofOutput.data(), 1, req.recursivefProofOutput.size(), f); fclose(f); }
        }
        proofs = 0;
    }
    for (int it = 0; it < proofs; it++) {
        if (world) XLP_CHECK(xlp_comm_stats(NULL, NULL, NULL, 1 | (timers ? 2 : 0)));
        prover.genBatchProof(&req);
        if (xlp_plan_active()) {                       // XLP_PLAN=1: what THIS rank allocated on the device while it walked the batch proof (include/xlp_gl.h, plan mode)
            uint64_t peak = 0, live = 0, allocs = 0, big[12], slots[9];
            XLP_CHECK(xlp_plan_report(&peak, &live, &allocs, big, 12, slots));
            std::ostringstream os;
            os << "{\"memory_plan\":{\"world\":" << (world ? world : 1) << ",\"rank\":" << rank << ",\"run\":" << it << ",\"peak_bytes\":" << peak << ",\"live_bytes\":" << live
               << ",\"allocations\":" << allocs << ",\"largest_at_peak\":[";
            for (int i = 0; i < 12; i++) os << (i ? "," : "") << big[i];
            os << "],\"scratch_slot_bytes\":[";
            for (int i = 0; i < 9; i++) os << (i ? "," : "") << slots[i];
            os << "]}}";
            printf("%s\n", os.str().c_str());
            fflush(stdout);
            continue;
        }
        if (world) {
            XLP_CHECK(xlp_comm_barrier());
            if (rank != 0) continue;
        }
        double starks = 0, exec = 0, total = 0, wit = 0;
        std::ostringstream os;
        os << "{\"run\":" << it << ",\"witness_upload\":\"" << (sequential ? "sequential" : "pipelined with step 1") << "\",\"timers_s\":{";
        bool first = true;
        for (const auto& kv : req.timers) {
            os << (first ? "" : ",") << "\"" << kv.first << "\":" << kv.second; first = false;
            if (kv.first == "STARK_PROOF_BATCH_PROOF" || kv.first == "STARK_C12_A_PROOF_BATCH_PROOF" || kv.first == "STARK_RECURSIVE_1_PROOF_BATCH_PROOF") starks += kv.second;
            if (kv.first == "EXECUTOR_EXECUTE_BATCH_PROOF") exec = kv.second;
            if (kv.first == "STARK_GEN_AND_CALC_WITNESS_C12A" || kv.first == "STARK_GEN_AND_CALC_WITNESS_RECURSIVE1") wit += kv.second;
            if (kv.first == "PROVER_BATCH_PROOF") total = kv.second;
        }
        // req.timers is appended when a scope closes: PROVER_BATCH_PROOF is still open here -> take it from the parts
        if (world) {
            double sec = 0; uint64_t bytes = 0, calls = 0;
            XLP_CHECK(xlp_comm_stats(&sec, &bytes, &calls, 0));
            os << "},\"sharded\":{\"world\":" << world << ",\"ranks_seen\":" << xlp_comm_ranks_seen() << ",\"backend\":\"" << (backend.empty() ? "self" : backend)
               << "\",\"same_zkevm_proof_on_all_ranks\":" << (req.zkevmProofSameOnAllRanks == 1 ? "true" : req.zkevmProofSameOnAllRanks == 0 ? "false" : "null")
               << ",\"exchanges\":" << calls << ",\"GB_sent_to_peers_rank0\":" << bytes / 1e9 << ",\"exchange_s_rank0\":" << sec << ",\"exchange_by_stage_rank0\":{";
            for (size_t i = 0; i < req.exchanges.size(); i++) {
                const auto& e = req.exchanges[i];
                os << (i ? "," : "") << "\"" << e.stage << "\":{\"calls\":" << e.calls << ",\"GB_sent\":" << e.GB << ",\"exchange_s\":" << e.seconds
                   << ",\"GBps_sent\":" << (e.seconds > 0 ? e.GB / e.seconds : 0.0) << "}";
            }
            os << "}";
        }
        os << "},\"three_starks_s\":" << starks << ",\"executor_standin_s\":" << exec << ",\"witness_standins_s\":" << wit << ",\"zkevm_witness_upload_ms\":" << req.witnessUploadMs;
        const std::string& z = req.batchProofOutput;
        const size_t r1 = z.find("\"root1\":"), r1e = z.find("]", r1);
        os << ",\"recursive1_root1\":" << z.substr(r1 + 8, r1e - r1 - 7) << ",\"batch_proof_bytes\":" << z.size() << ",\"batch_proof_fnv\":\"" << std::hex << fnv1a(z) << std::dec << "\"}";
        (void)total;
        printf("%s\n", os.str().c_str());
        fflush(stdout);
    }
    }
    xlp_shutdown();
    return 0;
}
