// Drives the host-side mirror (xlp_host.hpp) through the commit + FRI part of
// Starks::genProof (starks.cpp:28-60,134-141,215-222,285-293,394-402) on a small
// synthetic STARK and prints the proof in zkin layout.  tests/test_host_shim.py
// rebuilds the same proof with the CPU oracle and compares every field.
//   usage: test_host_shim <nBits> <nBitsExt> <nQueries> <step nBits...>
#include <cstdio>
#include <vector>
#include <omp.h>
#include "starks_mini.hpp"
#include "starks.hpp"
#include "xlp_host_bn128.hpp"
#include "starkRecursiveF.hpp"
#include "prover.hpp"
#include "fflonk_prover.hpp"
#include "fflonk_bench_key.hpp"

// `polinomial <n> <seed>`: Polinomial::batchInverse / calculateZ / calculateH1H2 through STRIDED views of one slab
// (offset > dim, as the section views of starks.cpp:108-184 are), results as JSON for the oracle comparison.
static int polinomial_mode(uint64_t n, uint64_t seed)
{
    typedef Goldilocks::Element E;
    const uint64_t W = 11;                                  // slab width: columns 0-2 a, 3-5 b, 6-8 out3, 9 f/h1, 10 t/h2
    std::vector<E> slab(n * W), tcol(n), fcol(n), h1(n), h2(n), inv1(n);
    mini_fill(slab.data(), slab.size(), seed);
    for (uint64_t i = 0; i < n; i++) { tcol[i].fe = (i / 3) * 7 + 1; }                       // table with repeats
    for (uint64_t i = 0; i < n; i++) fcol[i] = tcol[mini_splitmix64(seed ^ (i + 99)) % n];   // every f value occurs in t
    auto dump = [](const char* name, Goldilocks::Element* p, uint64_t cnt, uint64_t dim, uint64_t off, bool last) {
        printf("\"%s\":[", name);
        for (uint64_t i = 0; i < cnt; i++)
            for (uint64_t c = 0; c < dim; c++) printf("%s\"%s\"", (i || c) ? "," : "", Goldilocks::toString(p[i * off + c]).c_str());
        printf("]%s", last ? "" : ",");
    };
    printf("{");
    dump("slab", slab.data(), n * W, 1, 1, false);
    Polinomial a(&slab[0], n, 3, W), b(&slab[3], n, 3, W), o3(&slab[6], n, 3, W);
    Polinomial::batchInverse(o3, a);
    dump("inv3", &slab[6], n, 3, W, false);
    Polinomial::calculateZ(o3, a, b);
    dump("z", &slab[6], n, 3, W, false);
    Polinomial c1(&slab[9], n, 1, W), r1(inv1.data(), n, 1);
    Polinomial::batchInverseParallel(r1, c1);
    dump("inv1", inv1.data(), n, 1, 1, false);
    Polinomial f(fcol.data(), n, 1), t(tcol.data(), n, 1), H1(h1.data(), n, 1), H2(h2.data(), n, 1);
    Polinomial::calculateH1H2_opt1(H1, H2, f, t, 0, NULL, 0, 0);
    dump("f", fcol.data(), n, 1, 1, false); dump("t", tcol.data(), n, 1, 1, false);
    dump("h1", h1.data(), n, 1, 1, false); dump("h2", h2.data(), n, 1, 1, true);
    printf("}\n");
    xlp_shutdown();
    return 0;
}

// `starkinfo <file>`: loads a starkinfo.json with the mirror of StarkInfo and prints what genProof reads from it
static int starkinfo_mode(const char* file)
{
    StarkInfo si(file);
    printf("{\"nBits\":%llu,\"nBitsExt\":%llu,\"nQueries\":%llu,\"hash\":\"%s\",\"steps\":[", (unsigned long long)si.starkStruct.nBits,
           (unsigned long long)si.starkStruct.nBitsExt, (unsigned long long)si.starkStruct.nQueries, si.starkStruct.verificationHashType.c_str());
    for (size_t i = 0; i < si.starkStruct.steps.size(); i++) printf("%s%llu", i ? "," : "", (unsigned long long)si.starkStruct.steps[i].nBits);
    printf("],\"mapTotalN\":%llu,\"nConstants\":%llu,\"qDeg\":%llu,\"constTreeBytes\":%llu,\"offsets\":[", (unsigned long long)si.mapTotalN,
           (unsigned long long)si.nConstants, (unsigned long long)si.qDeg, (unsigned long long)si.getConstTreeSizeInBytes());
    for (int s = 0; s < eSectionMax; s++) printf("%s%llu", s ? "," : "", (unsigned long long)si.mapOffsets.section[s]);
    printf("],\"pols\":[");
    Goldilocks::Element* base = (Goldilocks::Element*)0;
    for (size_t id = 0; id < si.varPolMap.size(); id++) {
        Polinomial p; si.getPolinomial(p, base, id);
        PolInfo pi; si.getPol(base, id, pi);
        printf("%s[%llu,%llu,%llu,%llu,%llu,%llu]", id ? "," : "", (unsigned long long)(p.address() - base), (unsigned long long)p.degree(),
               (unsigned long long)p.dim(), (unsigned long long)p.offset(), (unsigned long long)si.getPolSize(id), (unsigned long long)(pi.get(3) - base));
    }
    printf("],\"evMap\":[");
    for (size_t i = 0; i < si.evMap.size(); i++) printf("%s[%d,%llu,%d]", i ? "," : "", (int)si.evMap[i].type, (unsigned long long)si.evMap[i].id, si.evMap[i].prime ? 1 : 0);
    printf("],\"puCtx\":%zu,\"peCtx\":%zu,\"ciCtx\":%zu,\"step3_ops\":%zu,\"step3_first_src\":[", si.puCtx.size(), si.peCtx.size(), si.ciCtx.size(), si.step3.first.size());
    if (!si.step3.first.empty())
        for (size_t k = 0; k < si.step3.first[0].src.size(); k++) {
            const StepType& t = si.step3.first[0].src[k];
            printf("%s[%d,%llu,%d,\"%s\"]", k ? "," : "", (int)t.type, (unsigned long long)t.id, t.prime ? 1 : 0, t.value.c_str());
        }
    printf("],\"exp2pol\":%zu,\"tmpExp_n\":%zu}\n", si.exp2pol.size(), si.tmpExp_n.size());
    return 0;
}

// `genproof <dir> <name>`: the GPU-resident Starks::genProof (starks.hpp) on <dir>/<name>.starkinfo.json, <name>.const,
// <name>.commit (cm1_n), <name>.publics (u64 each) and the step programs <dir>/<name>.<step>.xlpexpr; prints the zkin proof
static std::vector<Goldilocks::Element> read_u64_file(const std::string& f)
{
    FILE* fp = fopen(f.c_str(), "rb");
    if (!fp) xlp_exit_process(("cannot open " + f).c_str());
    fseek(fp, 0, SEEK_END); const long sz = ftell(fp); fseek(fp, 0, SEEK_SET);
    std::vector<Goldilocks::Element> v(sz / 8);
    if (sz && fread(v.data(), 8, v.size(), fp) != v.size()) xlp_exit_process(("short read on " + f).c_str());
    fclose(fp);
    return v;
}
static std::string zkinWithPublics(Starks& starks, FRIProof& proof)
{
    std::string zkin = proof2zkinStark(proof);
    std::ostringstream os;
    os << zkin.substr(0, zkin.size() - 1) << ",\"publics\":";
    xlp_json::arr(os, proof.publics);
    os << ",\"verkey\":[";
    for (int i = 0; i < HASH_SIZE; i++) os << (i ? "," : "") << "\"" << Goldilocks::toString(starks.constRoot[i]) << "\"";
    os << "]}";
    return os.str();
}
static int genproof_mode(const std::string& dir, const std::string& name, bool timers)
{
    std::vector<Goldilocks::Element> cm1 = read_u64_file(dir + "/" + name + ".commit"), publics = read_u64_file(dir + "/" + name + ".publics");
    Config config;
    config.runFileGenBatchProof = true;
    // pAddress: the host slab the executor would have filled; only cm1_n (offset mapOffsets[cm1_n]) is read from it
    Starks starks(config, {dir + "/" + name + ".const", false, dir + "/" + name + ".consttree", dir + "/" + name + ".starkinfo.json"}, cm1.data());
    XlpSteps steps(dir, name);
    starks.logTimers = timers;
    static const char* stepNames[5] = {"step2prev", "step3prev", "step3", "step42ns", "step52ns"};
    for (int k = 0; k < 5; k++)
        if (steps.optimised(k)) fprintf(stderr, "%s: dot-product form\n", stepNames[k]);
    const StarkStruct& ss = starks.starkInfo.starkStruct;
    FRIProof proof(1ULL << ss.steps.back().nBits, FIELD_EXTENSION, ss.steps.size(), starks.starkInfo.evMap.size(), starks.starkInfo.nPublics);
    for (uint64_t i = 0; i < starks.starkInfo.nPublics; i++) proof.publics[i] = publics[i];
    starks.genProof(proof, publics.data(), starks.constRoot, &steps);
    const std::string out = zkinWithPublics(starks, proof);
    fwrite(out.data(), 1, out.size(), stdout);
    printf("\n");
    return 0;
}

// `bundle <dir> <name> <witness seed> <const seed> [timers] [--world G --rank r --backend rccl|shm --rendezvous <file | name>]`: one STARK from a
// bundle of tools/make_stark_bundle.py (synthesized starkinfo + relocated reference programs) on a synthetic witness left in page-locked
// host memory, synthetic constants and synthetic plookup selectors -- the inputs of tools/bench_zkevm_genproof.py, so that the two drivers
// must print the same proof.  With --world the bundle must be the per-rank window bundle of that world (make_stark_bundle.py --world G):
// this process is rank r of a row-sharded proof (host/starks.hpp), its host slab holds ITS rows of the same synthetic witness, and rank 0
// prints the proof, which must equal the one-GPU proof bit for bit.
static int bundle_mode(const std::string& dir, const std::string& name, uint64_t wseed, uint64_t cseed, bool timers, int world, int rank,
                       const std::string& backend, const std::string& rendezvous, const std::string& merkelize = "")
{
    Config config;
    config.runFileGenBatchProof = true;
    config.xlpSyntheticLookups = true;
    char cs[64];
    snprintf(cs, sizeof cs, "synthetic:%llu", (unsigned long long)cseed);
    StarkInfo si(dir + "/" + name + ".starkinfo.json");
    const uint64_t N = 1ULL << si.starkStruct.nBits, W = si.mapSectionsN.section[cm1_n];
    void* host = NULL;
    XLP_CHECK(xlp_init(0));
    if (world) XLP_CHECK(xlp_comm_init(rank, world, backend.empty() ? NULL : backend.c_str(), rendezvous.empty() ? NULL : rendezvous.c_str()));
    const uint64_t rows = world ? N / world : N, first = world ? rows * rank : 0;     // my rows of the witness
    XLP_CHECK(xlp_host_alloc(&host, (rows * W > 0 ? rows * W : 1) * 8));
    {
        void* d = NULL;
        XLP_CHECK(xlp_dev_alloc(&d, rows * W * 8));
        XLP_CHECK(xlp_dev_fill((uint64_t*)d, rows * W, wseed, first * W));
        XLP_CHECK(xlp_copy_d2h(host, d, rows * W * 8));
        XLP_CHECK(xlp_dev_free(d));
    }
    {
        Starks starks(config, {cs, false, "", dir + "/" + name + ".starkinfo.json"}, host);
        XlpSteps steps(dir, name);
        starks.logTimers = timers;
        for (int k = 0; k < 5; k++) fprintf(stderr, "step %d: %s\n", k, steps.mode(k).c_str());
        const StarkStruct& ss = starks.starkInfo.starkStruct;
        FRIProof proof(1ULL << ss.steps.back().nBits, FIELD_EXTENSION, ss.steps.size(), starks.starkInfo.evMap.size(), starks.starkInfo.nPublics);
        std::vector<Goldilocks::Element> publics(starks.starkInfo.nPublics);
        for (uint64_t i = 0; i < publics.size(); i++) publics[i] = Goldilocks::fromU64(i + 1);
        for (uint64_t i = 0; i < publics.size(); i++) proof.publics[i] = publics[i];
        if (world) XLP_CHECK(xlp_comm_stats(NULL, NULL, NULL, 1 | 2));
        starks.genProof(proof, publics.data(), starks.constRoot, &steps);
        fprintf(stderr, "witness upload busy %.3f ms\n", starks.lastUploadMs);
        if (world) {
            double sec = 0; uint64_t bytes = 0, calls = 0;
            XLP_CHECK(xlp_comm_stats(&sec, &bytes, &calls, 4));
            fprintf(stderr, "rank %d of %d: %llu exchanges, %.3f GB sent to peers in %.4f s\n", rank, world, (unsigned long long)calls, bytes / 1e9, sec);
        }
        if (!world || rank == 0) {
            const std::string out = zkinWithPublics(starks, proof);
            fwrite(out.data(), 1, out.size(), stdout);
            printf("\n");
        }
        if (merkelize.size()) {                            // Starks::merkelizeMemory after the proof: the root on stderr, the slab's words to the file (the test hashes them on the oracle)
            Goldilocks::Element r[4];
            starks.merkelizeMemory(r, false);
            fprintf(stderr, "rootDBG %llu %llu %llu %llu words %llu\n", (unsigned long long)Goldilocks::toU64(r[0]), (unsigned long long)Goldilocks::toU64(r[1]),
                    (unsigned long long)Goldilocks::toU64(r[2]), (unsigned long long)Goldilocks::toU64(r[3]), (unsigned long long)starks.slabSizeWords());
            std::vector<uint64_t> h(starks.slabSizeWords());
            XLP_CHECK(xlp_copy_d2h(h.data(), starks.slabDevice(), h.size() * 8));
            FILE* f = fopen(merkelize.c_str(), "wb");
            if (!f || fwrite(h.data(), 8, h.size(), f) != h.size()) { fprintf(stderr, "cannot write %s\n", merkelize.c_str()); return 2; }
            fclose(f);
        }
        if (world) XLP_CHECK(xlp_comm_barrier());
    }
    XlpDeviceSlab::get().release();
    XLP_CHECK(xlp_host_free(host));
    return 0;
}

// `bn128 <height> <width> <seed>`: MerkleTreeBN128 (host-pointer and GPU-resident forms), getGroupProof and TranscriptBN128 as
// StarkRecursiveF::genProof uses them (starkRecursiveF.cpp:150-210); JSON for the comparison in tests/test_bn128.py.
static int bn128_mode(uint64_t h, uint64_t w, uint64_t seed)
{
    std::vector<Goldilocks::Element> src(h * w);
    mini_fill(src.data(), src.size(), seed);
    MerkleTreeBN128 tree(h, w, src.data());
    tree.merkelize();
    RawFr::Element root;
    tree.getRoot(&root);
    // the same tree GPU-resident
    void* d = NULL;
    XLP_CHECK(xlp_dev_alloc(&d, ((h * w) != 0 ? h * w : 1) * 8));
    XLP_CHECK(xlp_copy_h2d(d, src.data(), h * w * 8));
    MerkleTreeBN128 dtree(h, w, (uint64_t*)d, true);
    dtree.merkelize();
    RawFr::Element droot;
    dtree.getRoot(&droot);
    const uint64_t idx = (seed * 7 + 3) % h, plen = MerkleTreeBN128::getMerkleProofLength(h);
    std::vector<uint8_t> pa(w * 8 + plen * 16 * 32 + 8), pb(pa.size());
    tree.getGroupProof(pa.data(), idx);
    dtree.getGroupProof(pb.data(), idx);
    printf("{\"root\":\"%s\",\"root_dev\":\"%s\",\"numNodes\":%llu,\"idx\":%llu,\"proofs_equal\":%s,\"proof\":[", RawFr::field.toString(root, 10).c_str(),
           RawFr::field.toString(droot, 10).c_str(), (unsigned long long)tree.numNodes, (unsigned long long)idx, pa == pb ? "true" : "false");
    const RawFr::Element* mp = (const RawFr::Element*)(pa.data() + w * 8);
    for (uint64_t i = 0; i < plen * 16; i++) printf("%s\"%s\"", i ? "," : "", RawFr::field.toString(mp[i], 10).c_str());
    TranscriptBN128 tr;
    tr.put(&root, 1);
    tr.put(src.data(), w < 5 ? w : 5);
    uint64_t f1[3], f2[3], ys[8];
    tr.getField(f1);
    tr.put(src.data(), 1);
    tr.getField(f2);
    tr.getPermutations(ys, 8, 10);
    printf("],\"field1\":[%llu,%llu,%llu],\"field2\":[%llu,%llu,%llu],\"ys\":[", (unsigned long long)f1[0], (unsigned long long)f1[1], (unsigned long long)f1[2],
           (unsigned long long)f2[0], (unsigned long long)f2[1], (unsigned long long)f2[2]);
    for (int i = 0; i < 8; i++) printf("%s%llu", i ? "," : "", (unsigned long long)ys[i]);
    printf("]}\n");
    XLP_CHECK(xlp_dev_free(d));
    return 0;
}

// `genproof_bn128 <dir> <name>`: StarkRecursiveF::genProof (host/starkRecursiveF.hpp: the Starks flow with MerkleTreeBN128 trees,
// TranscriptBN128 and 16-ary openings) on the files of `genproof`; prints the recursivef-style zkin + publics + the constant root
static int genproof_bn128_mode(const std::string& dir, const std::string& name)
{
    std::vector<Goldilocks::Element> cm1 = read_u64_file(dir + "/" + name + ".commit"), publics = read_u64_file(dir + "/" + name + ".publics");
    Config config;
    config.runFileGenFinalProof = true;
    StarkRecursiveF starks(config, {dir + "/" + name + ".const", false, dir + "/" + name + ".consttree_bn128", dir + "/" + name + ".starkinfo.json"}, cm1.data());   // the tree file is optional
    XlpSteps steps(dir, name);
    const StarkStruct& ss = starks.starkInfo.starkStruct;
    FRIProofC12 proof(1ULL << ss.steps.back().nBits, FIELD_EXTENSION, ss.steps.size(), starks.starkInfo.evMap.size(), starks.starkInfo.nPublics);
    for (uint64_t i = 0; i < starks.starkInfo.nPublics; i++) proof.publics[i] = publics[i];
    starks.genProof(proof, publics.data(), &steps);
    const std::string zkin = proof2zkinStarkC12(proof);
    std::ostringstream os;
    os << zkin.substr(0, zkin.size() - 1) << ",\"publics\":";
    xlp_json::arr(os, proof.publics);
    os << ",\"verkey\":\"" << RawFr::field.toString(starks.constRoot[0], 10) << "\"}";
    const std::string out = os.str();
    fwrite(out.data(), 1, out.size(), stdout);
    printf("\n");
    return 0;
}


// Re-entrancy of the host-pointer boundary (SURVEY 8b "Threading"): FRIProve::prove of the reference runs, inside
// `#pragma omp parallel` (friProve.cpp:44-108), per group `NTT_Goldilocks ntt(nX, 1); ntt.INTT(ppar_c, ppar, nX, FIELD_EXTENSION)`.
// The same here on all OpenMP threads at once: group g's input is splitmix-seeded (splitmix64(seed ^ index) mod p: the generator of SURVEY 8d, oracle orc_fill), its INTT is
// written to stdout as one JSON row.  tests/test_host_shim.py compares every row with the oracle.  Also mixes in the other small
// host-pointer calls a transcript / Merkle walk makes from a thread (hash_full_result, fri_fold) so that staging slots are shared.
static uint64_t splitmix_fe(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z % 0xFFFFFFFF00000001ULL;
}
static int omp_intt_mode(uint64_t nX, uint64_t groups, uint64_t seed)
{
    std::vector<std::vector<uint64_t>> out(groups, std::vector<uint64_t>(nX * 3)), perms(groups, std::vector<uint64_t>(12));
    int nthreads = 1;
#pragma omp parallel
    {
#pragma omp single
        nthreads = omp_get_num_threads();
#pragma omp for schedule(dynamic, 1)
        for (uint64_t g = 0; g < groups; g++) {
            Polinomial ppar(nX, FIELD_EXTENSION), ppar_c(nX, FIELD_EXTENSION);
            for (uint64_t i = 0; i < nX * 3; i++) ppar.address()[i] = Goldilocks::fromU64(splitmix_fe(seed ^ (g * nX * 3 + i)));
            NTT_Goldilocks ntt(nX, 1);
            ntt.INTT(ppar_c.address(), ppar.address(), nX, FIELD_EXTENSION);
            for (uint64_t i = 0; i < nX * 3; i++) out[g][i] = Goldilocks::toU64(ppar_c.address()[i]);
            Goldilocks::Element st[12], res[12];
            for (int i = 0; i < 12; i++) st[i] = Goldilocks::fromU64(splitmix_fe((seed + 1) ^ (g * 12 + i)));
            PoseidonGoldilocks::hash_full_result(res, st);
            for (int i = 0; i < 12; i++) perms[g][i] = Goldilocks::toU64(res[i]);
        }
    }
    printf("{\"threads\":%d,\"intt\":[", nthreads);
    for (uint64_t g = 0; g < groups; g++) {
        printf("%s[", g ? "," : "");
        for (uint64_t i = 0; i < nX * 3; i++) printf("%s%llu", i ? "," : "", (unsigned long long)out[g][i]);
        printf("]");
    }
    printf("],\"perm\":[");
    for (uint64_t g = 0; g < groups; g++) {
        printf("%s[", g ? "," : "");
        for (int i = 0; i < 12; i++) printf("%s%llu", i ? "," : "", (unsigned long long)perms[g][i]);
        printf("]");
    }
    printf("]}\n");
    xlp_shutdown();
    return 0;
}

int main(int argc, char** argv)
{
    if (argc == 4 && std::string(argv[1]) == "genproof_bn128") { const int r = genproof_bn128_mode(argv[2], argv[3]); XlpDeviceSlab::get().release(); xlp_shutdown(); return r; }
    if (argc == 5 && std::string(argv[1]) == "bn128") { const int r = bn128_mode(strtoull(argv[2], 0, 10), strtoull(argv[3], 0, 10), strtoull(argv[4], 0, 0)); xlp_shutdown(); return r; }
    if ((argc == 4 || argc == 5) && std::string(argv[1]) == "genproof") { const int r = genproof_mode(argv[2], argv[3], argc == 5); XlpDeviceSlab::get().release(); xlp_shutdown(); return r; }
    if (argc >= 6 && std::string(argv[1]) == "bundle") {
        bool timers = false;
        int world = 0, rank = 0;
        std::string backend, rendezvous, merkelize;
        for (int i = 6; i < argc; i++) {
            const std::string a = argv[i];
            if (a == "--world" && i + 1 < argc) world = atoi(argv[++i]);
            else if (a == "--merkelize" && i + 1 < argc) merkelize = argv[++i];
            else if (a == "--rank" && i + 1 < argc) rank = atoi(argv[++i]);
            else if (a == "--backend" && i + 1 < argc) backend = argv[++i];
            else if (a == "--rendezvous" && i + 1 < argc) rendezvous = argv[++i];
            else timers = true;
        }
        const int r = bundle_mode(argv[2], argv[3], strtoull(argv[4], 0, 0), strtoull(argv[5], 0, 0), timers, world, rank, backend, rendezvous, merkelize);
        xlp_shutdown();
        return r;
    }
    // fflonk <zkey> <wtns> [blinding.bin] [proofs]: Fflonk::FflonkProver::setZkey + prove (host/fflonk_prover.hpp); blinding.bin = b_1 .. b_9 as 32-byte
    // little-endian integers (a test fixes them; without the file they are random, as in the reference).  Prints {"proof": ..., "publics": ..., "timers_s": ...}
    if ((argc >= 4 && argc <= 6) && std::string(argv[1]) == "fflonk") {
        auto zk = BinFileUtils::openExisting(argv[2], "zkey", 1);
        auto wt = BinFileUtils::openExisting(argv[3], "wtns", 2);
        Fflonk::FflonkProver prover;
        if (argc >= 5 && argv[4][0]) {
            FILE* f = fopen(argv[4], "rb");
            uint64_t b[36];
            if (!f || fread(b, 8, 36, f) != 36) { fprintf(stderr, "cannot read the blinding factors\n"); return 2; }
            fclose(f);
            prover.setBlindingFactors(b);
        }
        prover.setZkey(zk.get());
        const int proofs = argc == 6 ? atoi(argv[5]) : 1;
        for (int k = 0; k < proofs; k++) {
            const auto r = prover.prove(wt.get());
            std::string t = "{";
            for (const auto& kv : prover.lastTimers) t += std::string(t.size() > 1 ? "," : "") + "\"" + kv.first + "\":" + std::to_string(kv.second);
            printf("{\"proof\":%s,\"publics\":%s,\"timers_s\":%s}}\n", r.first.c_str(), r.second.c_str(), t.c_str());
        }
        prover.removePrecomputedData();
        xlp_shutdown();
        return 0;
    }
    // keccak <hex>: host only -- Keccak-256 of the bytes and the challenge Keccak256Transcript::getChallenge would derive from them
    if (argc == 3 && std::string(argv[1]) == "keccak") {
        const std::string hex = argv[2];
        std::vector<uint8_t> data;
        for (size_t i = 0; i + 1 < hex.size(); i += 2) data.push_back((uint8_t)strtoul(hex.substr(i, 2).c_str(), NULL, 16));
        uint8_t h[32];
        keccak256(data.data(), data.size(), h);
        std::string hs;
        char b[3];
        for (int i = 0; i < 32; i++) { snprintf(b, sizeof b, "%02x", h[i]); hs += b; }
        const FrElement ch = FrH::fromRprBE(h);
        uint8_t back[32];
        FrH::toRprBE(ch, back);
        std::string bs;
        for (int i = 0; i < 32; i++) { snprintf(b, sizeof b, "%02x", back[i]); bs += b; }
        printf("{\"keccak256\":\"%s\",\"challenge\":\"%s\",\"challenge_be\":\"%s\",\"inverse_times_value\":\"%s\"}\n", hs.c_str(), FrH::toString(ch).c_str(), bs.c_str(),
               FrH::toString(FrH::mul(ch, FrH::inv(ch))).c_str());
        return 0;
    }
    // fflonk_bench <power> [proofs]: the prover on a synthetic key of 2^power rows built in memory (host/fflonk_bench_key.hpp): per-round seconds
    if ((argc == 3 || argc == 4) && std::string(argv[1]) == "fflonk_bench") {
        const uint32_t power = (uint32_t)atoi(argv[2]);
        if (power < 4 || power > 24) { fprintf(stderr, "power 4..24\n"); return 2; }
        XLP_CHECK(xlp_init(0));
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<FrElement> witness;
        BinFileUtils::BinFile zk(FflonkBench::build(power, witness), "zkey", 1);
        const double tBuild = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        Fflonk::FflonkProver prover;
        const auto t1 = std::chrono::steady_clock::now();
        prover.setZkey(&zk);
        const double tKey = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
        const int proofs = argc == 4 ? atoi(argv[3]) : 2;
        for (int k = 0; k < proofs; k++) {
            const auto t2 = std::chrono::steady_clock::now();
            const auto r = prover.prove(witness.data());
            const double tp = std::chrono::duration<double>(std::chrono::steady_clock::now() - t2).count();
            std::string t = "{";
            for (const auto& kv : prover.lastTimers) t += std::string(t.size() > 1 ? "," : "") + "\"" + kv.first + "\":" + std::to_string(kv.second);
            printf("{\"power\":%u,\"proof\":%d,\"prove_s\":%.6f,\"timers_s\":%s},\"key_build_s\":%.3f,\"set_zkey_s\":%.3f,\"proof_bytes\":%zu}\n", power, k, tp, t.c_str(), tBuild, tKey, r.first.size());
            fflush(stdout);
        }
        prover.removePrecomputedData();
        xlp_shutdown();
        return 0;
    }
    if (argc == 3 && std::string(argv[1]) == "starkinfo") return starkinfo_mode(argv[2]);
    const bool aggregate = argc == 6 && std::string(argv[1]) == "aggregate";   // aggregate a.json b.json verkey.json steps: genAggregatedProof's input checks, then joinzkin
    if (aggregate || (argc == 6 && std::string(argv[1]) == "joinzkin")) {      // joinzkin a.json b.json verkey.json steps  (host only)
        std::string t[3];
        for (int i = 0; i < 3; i++) {
            FILE* f = fopen(argv[2 + i], "rb");
            if (!f) { fprintf(stderr, "cannot open %s\n", argv[2 + i]); return 2; }
            char buf[65536]; size_t n;
            while ((n = fread(buf, 1, sizeof buf, f)) > 0) t[i].append(buf, n);
            fclose(f);
        }
        const xlp_json::Value z1 = xlp_json::Parser(t[0]).value(), z2 = xlp_json::Parser(t[1]).value(), vk = xlp_json::Parser(t[2]).value();
        std::string why;
        if (aggregate && !Prover::aggregatedInputsAreConsecutive(z1, z2, &why)) {
            printf("{\"result\":%d,\"why\":\"%s\"}\n", (int)Prover::ZKR_AGGREGATED_PROOF_INVALID_INPUT, why.c_str());
            return 0;
        }
        const std::string out = joinzkin(z1, z2, vk, strtoull(argv[5], 0, 10));
        fwrite(out.data(), 1, out.size(), stdout);
        printf("\n");
        return 0;
    }
    if (argc == 5 && std::string(argv[1]) == "omp_intt") return omp_intt_mode(strtoull(argv[2], 0, 10), strtoull(argv[3], 0, 10), strtoull(argv[4], 0, 10));
    if (argc == 4 && std::string(argv[1]) == "polinomial") return polinomial_mode(strtoull(argv[2], 0, 10), strtoull(argv[3], 0, 10));
    if (argc < 5) { fprintf(stderr, "usage: %s nBits nBitsExt nQueries step0 [step1 ...]\n", argv[0]); return 2; }
    MiniStarkConfig cfg;
    cfg.nBits = strtoull(argv[1], 0, 10);
    cfg.nBitsExt = strtoull(argv[2], 0, 10);
    cfg.nQueries = strtoull(argv[3], 0, 10);
    for (int i = 4; i < argc; i++) cfg.steps.push_back(strtoull(argv[i], 0, 10));
    std::string json = mini_stark_prove(cfg);
    fwrite(json.data(), 1, json.size(), stdout);
    printf("\n");
    xlp_shutdown();
    return 0;
}
