// GPU-resident `class Starks` with the reference's interface (src/starkpil/starks.hpp:29-232, starks.cpp:9-404):
//     Starks(const Config& config, StarkFiles starkFiles, void* pAddress)
//     void genProof(FRIProof& proof, Goldilocks::Element* publicInputs, Goldilocks::Element verkey[4], Steps* steps)
// Same stage order, timer names and transcript sequence.  What differs, by construction:
//  * pAddress is still the HOST slab the executor (or the previous STARK's witness calculator) writes the committed polynomials
//    into (prover.cpp:94-115,435,593,630); only its cm1_n section is read.  Everything else of the slab lives in HBM: one device
//    slab shared by all Starks of the process, as the reference shares pAddress (prover.cpp:123-127).  Step 1 uploads the witness
//    in column bands on a second stream while the previous band is extended and absorbed into the leaf hashes
//    (xlp_dev_commit_from_host); pin pAddress (xlp_host_alloc / xlp_host_register) for the full PCIe rate.
//  * transposeH1H2Columns / transposeZColumns / ...Rows (starks.cpp:406-555) are address arithmetic: xlp_dev_calculate_h1h2_strided /
//    xlp_dev_calculate_z_strided take the polynomials as (first word, section width) views.
//  * The constant tree comes from starkFiles.zkevmConstantsTree when that file exists (the bctree layout the reference maps,
//    starks.hpp:117-136) and is computed from the constant polynomials otherwise.  "synthetic:<seed>" as zkevmConstPols fills the
//    constants on the device (benchmarks at sizes whose 15 GB .const file does not exist on the box).
//  * A starkinfo may place the extended constants inside the slab ("xlp": {"const2nsOffset": ...}: the one-GPU aliased layout of
//    tools/zkevm_geometry.py); they are then re-extended in step 4 of every proof, only their tree stays resident.
//  * nrowsStepBatch is irrelevant (one row per lane).
//  * ROW-SHARDED MODE (BASELINE north_star: "columns shard across the 8 GPUs ... a single RCCL all-gather ... to assemble
//    Merkle caps"): one process per GPU, xlp_comm_init(rank, world, ...) before the Starks is built, and a starkinfo laid out for that
//    world ("xlp": {"world": G}: every section is the rank's WINDOW = its N / G rows + the halo rows of the next rank).  genProof then
//    runs the same stages on the rank's rows: a section commit is rows -> columns (all-to-all), per-column extendPol, columns -> rows
//    (all-to-all), sub-tree, 32-byte cap all-gather; step programs run on the window with ring halo exchanges around them; H1H2 on the
//    all-gathered table; calculateZ = local scan + 3 words per rank; the 3-column quotient / FRI polynomials are all-gathered and
//    transformed on every rank; evmap = local sums + one all-gather; openings come from the owning rank.  Every rank runs the transcript
//    and ends with the complete proof.  pAddress then holds the rank's ROWS of the witness.  See genProofSharded below.
#pragma once
#include <iostream>
#include <chrono>
#include <string>
#include <vector>
#include "xlp_host.hpp"
#include "config.hpp"
#include "steps.hpp"

#define STARK_C12_A_NUM_TREES 5
#define NUM_CHALLENGES 8

struct StarkFiles
{
    std::string zkevmConstPols;
    bool mapConstPolsFile;
    std::string zkevmConstantsTree;
    std::string zkevmStarkInfo;
};

// One device slab for all Starks of the process (the reference's shared pAddress, prover.cpp:94-127)
class XlpDeviceSlab
{
    uint64_t* p = NULL;
    uint64_t words = 0;
    XlpDeviceSlab() {}
public:
    static XlpDeviceSlab& get() { static XlpDeviceSlab s; return s; }
    uint64_t* reserve(uint64_t w)
    {
        if (w > words) {
            if (p) XLP_CHECK(xlp_dev_free(p));
            void* d = NULL;
            XLP_CHECK(xlp_dev_alloc(&d, w * sizeof(uint64_t)));
            p = (uint64_t*)d; words = w;
        }
        return p;
    }
    uint64_t* address() { return p; }
    uint64_t capacityWords() const { return words; }
    void release() { if (p) xlp_dev_free(p); p = NULL; words = 0; }
};

// What differs between Starks (Poseidon-Goldilocks trees, Transcript) and StarkRecursiveF (MerkleTreeBN128, TranscriptBN128;
// starkRecursiveF.cpp:150-650 is starks.cpp:9-404 with those types): the hash policy H of StarksT<H>.
struct XlpHashGL
{
    typedef MerkleTreeGL Tree;
    typedef Transcript TranscriptT;
    typedef FRIProof Proof;
    typedef FRIProve Prove;
    typedef Goldilocks::Element Root[HASH_SIZE];
    static constexpr bool kFileTree = true;          // <stark>.consttree (bctree's file) can be loaded instead of computed
    static constexpr bool kCommitTimes = true;       // commits go through xlp_dev_commit*, which times its LDE and Merkle halves (subTimers)
    static constexpr bool kShardable = true;         // MerkleTreeGL has a row-sharded form (binary tree: a rank's sub-root is a node of the global tree)
    static uint64_t treeWords(Tree* t) { return t->getTreeNumElements(); }
    static void getField(TranscriptT& t, Goldilocks::Element* out) { t.getField(out); }
    static void putRoot(TranscriptT& t, Goldilocks::Element* r) { t.put(r, HASH_SIZE); }
    static void putVerkey(TranscriptT& t, Goldilocks::Element* verkey, Goldilocks::Element* /*constRoot*/) { t.put(&verkey[0], 4); }   // starks.cpp:33
    static void commit(Tree* tree, uint64_t* sec2ns, const uint64_t* secn, uint64_t NExtended, uint64_t N, uint64_t W)
    {
        XLP_CHECK(xlp_dev_commit(sec2ns, tree->d_nodes, secn, NExtended, N, W));
    }
    static void commitFromHost(Tree* tree, uint64_t* secn, uint64_t* sec2ns, const uint64_t* h_src, uint64_t NExtended, uint64_t N, uint64_t W, double* busyMs)
    {
        XLP_CHECK(xlp_dev_commit_from_host(secn, sec2ns, tree->d_nodes, h_src, NExtended, N, W, busyMs));
    }
    // the same two on column-major sections (csn / cs2: words between the columns of the base / extended section)
    static void commitCm(Tree* tree, uint64_t* sec2ns, uint64_t cs2, const uint64_t* secn, uint64_t csn, uint64_t NExtended, uint64_t N, uint64_t W)
    {
        XLP_CHECK(xlp_dev_commit_cm(sec2ns, cs2, tree->d_nodes, secn, csn, NExtended, N, W));
    }
    static void commitFromHostCm(Tree* tree, uint64_t* secn, uint64_t csn, uint64_t* sec2ns, uint64_t cs2, const uint64_t* h_src, uint64_t* stage, uint64_t stageWords,
                                 bool clearStage, uint64_t NExtended, uint64_t N, uint64_t W, double* busyMs)
    {
        XLP_CHECK(xlp_dev_commit_from_host_cm(secn, csn, sec2ns, cs2, tree->d_nodes, h_src, stage, stageWords, clearStage ? 1 : 0, NExtended, N, W, busyMs));
    }
    static void storeRoots(Proof& proof, Goldilocks::Element* r0, Goldilocks::Element* r1, Goldilocks::Element* r2, Goldilocks::Element* r3)
    {
        std::memcpy(&proof.proofs.root1[0], r0, HASH_SIZE * sizeof(Goldilocks::Element));
        std::memcpy(&proof.proofs.root2[0], r1, HASH_SIZE * sizeof(Goldilocks::Element));
        std::memcpy(&proof.proofs.root3[0], r2, HASH_SIZE * sizeof(Goldilocks::Element));
        std::memcpy(&proof.proofs.root4[0], r3, HASH_SIZE * sizeof(Goldilocks::Element));
    }
};

template <class H>
class StarksT
{
public:
    const Config& config;
    StarkInfo starkInfo;
    uint64_t nrowsStepBatch;
    typename H::Root constRoot;                     // root of the constant tree (= the verification key)
    bool logTimers = false;
    double lastUploadMs = 0.0;                      // time the copy stream was busy with the witness in the last genProof
    bool pipelineWitness = true;                    // false: upload the witness in one piece first, then commit it (the measurement of what the pipeline hides)
    std::vector<std::pair<std::string, double>> lastTimers;
    struct StageExchange { std::string stage; double seconds, GB; uint64_t calls; };
    std::vector<StageExchange> lastExchanges;       // row-sharded mode with xlp_comm_stats timing on: what each timed stage spent in xlp_comm_exchange

private:
    StarkFiles starkFiles;
    uint64_t N, NExtended, extendBits;
    void* pAddress;
    uint64_t slabWords = 0;
    bool const2nsInSlab = false;
    uint64_t const2nsOffset = 0;
    // COLUMN-MAJOR device sections ("xlp": {"colMajor": true} in the starkinfo; include/xlp_gl.h): every section of the device slab is stored
    // as columns at the offsets and sizes the starkinfo names (the extended ones coset-major), the constant polynomials likewise; the step
    // programs of the bundle are converted to match.  The host slab pAddress keeps the reference's row-major form: the witness is transposed
    // band by band as it arrives, opened rows are gathered.  Row-sharded: the rank's WINDOW as columns (see commitShardedCm).
    bool colMajor = false;
    uint64_t *d_const_n = NULL, *d_const_2ns_own = NULL, *d_x_n = NULL, *d_x_2ns = NULL, *d_zi = NULL, *d_xdx = NULL, *d_xdwx = NULL;
    uint64_t *d_challenges = NULL, *d_publics = NULL, *d_evals = NULL, *d_work = NULL;
    typename H::Tree* treesGL[STARK_C12_A_NUM_TREES] = {NULL, NULL, NULL, NULL, NULL};
    // row-sharded mode
    bool sharded = false;
    int rank = 0, world = 1;
    uint64_t nLoc = 0, neLoc = 0, haloN = 0, haloExt = 0, row0 = 0, ext0 = 0;     // my rows of the base / extended domain, halo rows, my first rows
    uint64_t *d_const_n_full = NULL, *d_const_2ns_win = NULL, *d_shard = NULL;
    uint64_t shardWords = 0;
    double commSeconds = 0; uint64_t commBytes = 0, commCalls = 0;   // xlp_comm_stats at the end of the last timed stage

    static uint64_t* dalloc(uint64_t elems)
    {
        void* p = NULL;
        XLP_CHECK(xlp_dev_alloc(&p, (elems ? elems : 1) * sizeof(uint64_t)));
        return (uint64_t*)p;
    }
    uint64_t* mem() { return XlpDeviceSlab::get().address(); }
    uint64_t* section(eSection s) { return mem() + starkInfo.mapOffsets.section[s]; }
    // In the aliased layout the extended constants are recomputed in step 4 of a proof (0.083 s of the zkEVM STARK) because they lie over
    // a dead section of the slab.  From the second proof on -- when the grow-only scratch of the first one exists and the free memory is
    // what it will stay -- a RESIDENT copy is taken instead if 8 GB remain free beside it (the reference keeps constPols2ns loaded,
    // starks.hpp:138-146); `residentConstants = false` keeps the trade as it was.
    bool residentConstants = true;                  // Config::xlpResidentConstants
    uint64_t* d_const_2ns_res = NULL;
    uint64_t proofsDone = 0;
    uint64_t* constPols2ns() { return d_const_2ns_res ? d_const_2ns_res : const2nsInSlab ? mem() + const2nsOffset : d_const_2ns_own; }
    struct PolRef { uint64_t* p; uint64_t n, dim, stride, cs; };   // element i, component c at p[i * stride + c * cs]
    PolRef pol(uint64_t id)          // StarkInfo::getPolinomial on the device slab
    {
        if (id >= starkInfo.varPolMap.size()) xlp_exit_process("Starks: polynomial id out of range");
        const VarPolMap& m = starkInfo.varPolMap[id];
        const uint64_t rows = starkInfo.mapDeg.section[m.section];
        if (colMajor) return {mem() + starkInfo.mapOffsets.section[m.section] + m.sectionPos * rows, rows, m.dim, 1, rows};
        return {mem() + starkInfo.mapOffsets.section[m.section] + m.sectionPos, rows, m.dim, starkInfo.mapSectionsN.section[m.section], 1};
    }
    uint64_t colStride(eSection sec) const { return starkInfo.mapDeg.section[sec]; }   // column-major: words between the columns of a section
    // a polynomial view <-> dense elements (the operands of calculateZ / calculateH1H2 when the sections are column-major)
    void toDense(uint64_t* dst, const PolRef& r, uint64_t n) { XLP_CHECK(xlp_dev_copy_view(dst, r.dim, 1, r.p, r.stride, r.cs, n, (uint32_t)r.dim)); }
    void fromDense(const PolRef& r, const uint64_t* src, uint64_t n) { XLP_CHECK(xlp_dev_copy_view(r.p, r.stride, r.cs, src, r.dim, 1, n, (uint32_t)r.dim)); }
    uint64_t exp2pol(uint64_t expId)
    {
        const std::string k = std::to_string(expId);
        for (const auto& kv : starkInfo.exp2pol) if (kv.first == k) return kv.second;
        xlp_exit_process(("Starks: expression " + k + " has no polynomial (exp2pol)").c_str());
        return 0;
    }
    struct Timer {           // TimerStart / TimerStopAndLog (src/utils/timer.hpp:17-27)
        StarksT* s; const char* name; std::chrono::steady_clock::time_point t0;
        Timer(StarksT* s_, const char* n) : s(s_), name(n) { xlp_sync(); t0 = std::chrono::steady_clock::now(); }
        ~Timer()
        {
            xlp_sync();
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            s->lastTimers.emplace_back(name, dt);
            if (s->sharded) {
                double sec = 0; uint64_t bytes = 0, calls = 0;
                if (xlp_comm_stats(&sec, &bytes, &calls, 0) == 0 && calls > s->commCalls) {
                    s->lastExchanges.push_back({name, sec - s->commSeconds, (bytes - s->commBytes) / 1e9, calls - s->commCalls});
                    s->commSeconds = sec; s->commBytes = bytes; s->commCalls = calls;
                }
            }
            if (s->logTimers) fprintf(stderr, "<-- %s done: %.6f s\n", name, dt);
        }
    };
    // The reference logs STARK_STEP_n_LDE and STARK_STEP_n_MERKLETREE one after the other (starks.cpp:51-58,133-139,214-220); here the two are
    // interleaved band by band inside STARK_STEP_n_LDE_AND_MERKLETREE, so the two names carry the summed kernel times of their halves
    // (HIP events on the compute stream): logs diff stage by stage, and the pair adds up to the fused stage minus what the upload exposes.
    void subTimers(const char* lde, const char* merkle)
    {
        if constexpr (H::kCommitTimes) {             // (the Goldilocks commit path: xlp_dev_commit*; the BN128 trees are built by other calls)
            double l = 0, m = 0;
            XLP_CHECK(xlp_dev_commit_last_times(&l, &m));
            lastTimers.emplace_back(lde, l * 1e-3);
            lastTimers.emplace_back(merkle, m * 1e-3);
            if (logTimers) fprintf(stderr, "<-- %s done: %.6f s (kernel events)\n<-- %s done: %.6f s (kernel events)\n", lde, l * 1e-3, merkle, m * 1e-3);
        }
    }
    StepsParams params()
    {
        StepsParams P;
        memset(&P, 0, sizeof P);
        // sharded: the buffers are this rank's windows (its rows + the halo rows) and the programs run on local row numbers
        const uint64_t rowsN = sharded ? nLoc + haloN : N, rowsE = sharded ? neLoc + haloExt : NExtended;
        const uint64_t ownN = sharded ? nLoc : N, ownE = sharded ? neLoc : NExtended;
        P.pols = mem(); P.pols_size = starkInfo.mapTotalN;
        P.const_pols = d_const_n; P.n_const_pols = starkInfo.nConstants; P.const_pols_size = rowsN * starkInfo.nConstants;
        P.const_pols_2ns = sharded ? d_const_2ns_win : constPols2ns(); P.n_const_pols_2ns = starkInfo.nConstants; P.const_pols_2ns_size = rowsE * starkInfo.nConstants;
        P.challenges = d_challenges; P.challenges_size = NUM_CHALLENGES * FIELD_EXTENSION;
        P.public_inputs = d_publics; P.public_inputs_size = starkInfo.nPublics;
        P.evals = d_evals; P.evals_size = starkInfo.evMap.size() * FIELD_EXTENSION;
        P.x_n = d_x_n; P.x_n_size = ownN;
        P.x_2ns = d_x_2ns; P.x_2ns_size = ownE;
        P.zi = d_zi; P.zi_size = 1ULL << extendBits;
        P.x_div_x_sub_xi = d_xdx; P.x_div_x_sub_xi_size = ownE * FIELD_EXTENSION;
        P.x_div_x_sub_wxi = d_xdwx; P.x_div_x_sub_wxi_size = ownE * FIELD_EXTENSION;
        P.q_2ns = section(q_2ns); P.q_2ns_size = rowsE * starkInfo.mapSectionsN.section[q_2ns];
        P.f_2ns = section(f_2ns); P.f_2ns_size = rowsE * starkInfo.mapSectionsN.section[f_2ns];
        return P;
    }
    static bool fileExists(const std::string& f) { FILE* fp = fopen(f.c_str(), "rb"); if (fp) fclose(fp); return fp != NULL; }
    // copyFile() of the reference (src/utils/utils.cpp) straight into HBM, through a bounded host buffer
    static void fileToDevice(uint64_t* dst, const std::string& file, uint64_t offsetBytes, uint64_t bytes)
    {
        FILE* fp = fopen(file.c_str(), "rb");
        if (!fp) xlp_exit_process(("Starks: cannot open " + file).c_str());
        if (fseek(fp, (long)offsetBytes, SEEK_SET)) xlp_exit_process(("Starks: cannot seek in " + file).c_str());
        const uint64_t chunk = 256ULL << 20;
        std::vector<char> buf(bytes < chunk ? bytes : chunk);
        for (uint64_t o = 0; o < bytes; o += chunk) {
            const uint64_t n = bytes - o < chunk ? bytes - o : chunk;
            if (fread(buf.data(), 1, n, fp) != n) xlp_exit_process(("Starks: short read on " + file).c_str());
            XLP_CHECK(xlp_copy_h2d((char*)dst + o, buf.data(), n));
        }
        fclose(fp);
    }

    // ------------------------------------------------------------------------------------------------ row-sharded mode
    // dst (rows x width, dense) <- rows (first + i) % total of src (total x width): a rank's window of a replicated table
    static void windowRows(uint64_t* dst, const uint64_t* src, uint64_t width, uint64_t first, uint64_t rows, uint64_t total)
    {
        if (!width || !rows) return;
        const uint64_t a = first % total, head = rows < total - a ? rows : total - a;
        XLP_CHECK(xlp_dev_copy_cols(dst, width, src + a * width, width, head, (uint32_t)width));
        for (uint64_t done = head; done < rows;) {          // (the halo of the last rank wraps to row 0)
            const uint64_t k = rows - done < total ? rows - done : total;
            XLP_CHECK(xlp_dev_copy_cols(dst + done * width, width, src, width, k, (uint32_t)width));
            done += k;
        }
    }
    static bool isExtSection(eSection s) { return s == cm1_2ns || s == cm2_2ns || s == cm3_2ns || s == cm4_2ns || s == q_2ns || s == f_2ns; }
    uint64_t ownRows(eSection s) const { return isExtSection(s) ? neLoc : nLoc; }
    uint64_t haloRows(eSection s) const { return isExtSection(s) ? haloExt : haloN; }
    // contiguous, balanced column shard of a rank (the first W % G ranks own one column more)
    void colSpan(uint64_t W, int r, uint64_t& c0, uint64_t& c1) const
    {
        const uint64_t base = W / world, rem = W % world;
        c0 = r * base + ((uint64_t)r < rem ? (uint64_t)r : rem);
        c1 = c0 + base + ((uint64_t)r < rem ? 1 : 0);
    }
    uint64_t* shardScratch(uint64_t words)
    {
        if (words > shardWords) {
            if (d_shard) XLP_CHECK(xlp_dev_free(d_shard));
            d_shard = dalloc(words); shardWords = words;
        }
        return d_shard;
    }
    // ---- column-major windows: a column of a base section = nLoc + haloN rows, of an extended one = 2^extendBits coset blocks of nLoc + haloN
    // rows each (the halo row of a coset block = the next rank's first row of that coset); cosets(s) blocks, cosStride() apart
    uint64_t cosets(eSection s) const { return isExtSection(s) ? (1ULL << extendBits) : 1; }
    uint64_t cosStride() const { return nLoc + haloN; }
    // dst (cols x cosets x rows, dense) <- rows [first, first + rows) of every coset block of `cols` columns of a section (or back: toSection)
    void windowRuns(uint64_t* dense, uint64_t* sec, uint64_t cs, uint64_t cols, uint64_t ncos, uint64_t first, uint64_t rows, bool toSection)
    {
        if (!cols || !rows) return;
        if (toSection) XLP_CHECK(xlp_dev_copy_3d(sec + first, cs, cosStride(), dense, ncos * rows, rows, cols, ncos, rows));
        else XLP_CHECK(xlp_dev_copy_3d(dense, ncos * rows, rows, sec + first, cs, cosStride(), cols, ncos, rows));
    }
    void fillHaloCm(eSection s)
    {
        const uint64_t W = starkInfo.mapSectionsN.section[s];
        if (!W || !haloN) return;
        const uint64_t ncos = cosets(s), words = W * ncos * haloN;
        uint64_t *pack = shardScratch(2 * words), *recv = pack + words;
        windowRuns(pack, section(s), colStride(s), W, ncos, 0, haloN, false);              // my first rows of every (column, coset) -> the previous rank
        XLP_CHECK(xlp_comm_sendrecv(recv, (rank + 1) % world, pack, (rank + world - 1) % world, words));
        windowRuns(recv, section(s), colStride(s), W, ncos, nLoc, haloN, true);             // the next rank's first rows: my halo
    }
    void haloWritebackCm(const std::vector<NextRowStore>& stores)
    {
        if (stores.empty() || world == 1) return;
        uint64_t total = 0;
        auto reach = [&](const NextRowStore& st) { return isExtSection((eSection)st.section) ? st.next >> extendBits : st.next; };   // rows of a coset block
        for (const NextRowStore& st : stores) total += reach(st) * st.cols * cosets((eSection)st.section);
        uint64_t *pack = shardScratch(2 * total), *recv = pack + total, o = 0;
        for (const NextRowStore& st : stores) {
            const eSection sec = (eSection)st.section;
            if (reach(st) > haloN || (reach(st) << (isExtSection(sec) ? extendBits : 0)) != st.next || st.col + st.cols > starkInfo.mapSectionsN.section[sec])
                xlp_exit_process("Starks: next-row store outside the halo");
            windowRuns(pack + o, section(sec) + st.col * colStride(sec), colStride(sec), st.cols, cosets(sec), nLoc, reach(st), false);
            o += reach(st) * st.cols * cosets(sec);
        }
        XLP_CHECK(xlp_comm_sendrecv(recv, (rank + world - 1) % world, pack, (rank + 1) % world, total));
        o = 0;
        for (const NextRowStore& st : stores) {
            const eSection sec = (eSection)st.section;
            windowRuns(recv + o, section(sec) + st.col * colStride(sec), colStride(sec), st.cols, cosets(sec), 0, reach(st), true);
            o += reach(st) * st.cols * cosets(sec);
        }
    }
    // the window's halo rows <- the first rows of the next rank (ring): my first rows go to the previous rank
    void fillHalo(eSection s)
    {
        if (colMajor) { fillHaloCm(s); return; }
        const uint64_t W = starkInfo.mapSectionsN.section[s];
        if (!W) return;
        XLP_CHECK(xlp_comm_sendrecv(section(s) + ownRows(s) * W, (rank + 1) % world, section(s), (rank + world - 1) % world, haloRows(s) * W));
    }
    // after a program with next-row stores: the halo rows it wrote belong to the next rank's first rows; mine come from the previous rank
    void haloWriteback(const std::vector<NextRowStore>& stores)
    {
        // (one rank: the window is the whole domain, the operand's own modulus wraps the store of the last row to row 0 -- nothing lands in the halo)
        if (colMajor) { haloWritebackCm(stores); return; }
        if (stores.empty() || world == 1) return;
        uint64_t total = 0;
        for (const NextRowStore& st : stores) total += st.next * st.cols;
        uint64_t *pack = shardScratch(2 * total), *recv = pack + total, o = 0;
        for (const NextRowStore& st : stores) {
            const eSection sec = (eSection)st.section;
            const uint64_t W = starkInfo.mapSectionsN.section[sec];
            if (st.next > haloRows(sec) || st.col + st.cols > W) xlp_exit_process("Starks: next-row store outside the halo");
            XLP_CHECK(xlp_dev_copy_cols(pack + o, st.cols, section(sec) + ownRows(sec) * W + st.col, W, st.next, (uint32_t)st.cols));
            o += st.next * st.cols;
        }
        XLP_CHECK(xlp_comm_sendrecv(recv, (rank + world - 1) % world, pack, (rank + 1) % world, total));
        o = 0;
        for (const NextRowStore& st : stores) {
            const eSection sec = (eSection)st.section;
            const uint64_t W = starkInfo.mapSectionsN.section[sec];
            XLP_CHECK(xlp_dev_copy_cols(section(sec) + st.col, W, recv + o, st.cols, st.next, (uint32_t)st.cols));
            o += st.next * st.cols;
        }
    }
    // band k of a W-column section cut into nb bands = the columns [lo(k), lo(k + 1)), boundaries on multiples of 8 (the leaf sponge takes a
    // section band by band, eight columns per permutation); rank r extends the r-th share [a, b) of the band
    void bandSpan(uint64_t W, uint64_t nb, uint64_t k, int r, uint64_t& a, uint64_t& b) const
    {
        auto lo = [&](uint64_t j) { return j >= nb ? W : (W * j / nb) & ~7ULL; };
        const uint64_t c0 = lo(k), w = lo(k + 1) - c0;
        a = c0 + w * (uint64_t)r / (uint64_t)world; b = c0 + w * ((uint64_t)r + 1) / (uint64_t)world;
    }
    // how the sharded commit of a W-column section is cut into column bands (see commitSharded): number of bands, and per band the most
    // columns all ranks together / one rank can hold; -> words of ONE of the two staging sets it needs
    uint64_t commitBandPlan(uint64_t W, uint64_t& nb, uint64_t& bandTotalMax, uint64_t& wcbMax, uint64_t minBands = 1) const
    {
        const uint64_t wcMax = (W + world - 1) / world;
        const uint64_t regroup = colMajor ? 2 : 1;                  // column-major windows: what arrives / leaves is regrouped through a second buffer
        const uint64_t allWords = nLoc * W + regroup * (N * wcMax + NExtended * wcMax) + neLoc * W;
        uint64_t budget = 24ULL << 30;
        if (const char* e = getenv("XLP_SHARD_SCRATCH_MB")) { const uint64_t v = strtoull(e, NULL, 10); if (v) budget = v << 20; }
        budget /= 2;                                                // two sets: band k + 1 is exchanged while band k is extended
        nb = (allWords * sizeof(uint64_t) + budget - 1) / budget;
        if (nb < minBands) nb = minBands;
        if (nb > W / world && W / world) nb = W / world;            // at least one column per rank and band
        if (W < (uint64_t)world) nb = 1;
        bandTotalMax = 0; wcbMax = 0;
        for (uint64_t k = 0; k < nb; k++) {
            uint64_t tot = 0;
            for (int p_ = 0; p_ < world; p_++) {
                uint64_t a, b; bandSpan(W, nb, k, p_, a, b);
                tot += b - a; if (b - a > wcbMax) wcbMax = b - a;
            }
            if (tot > bandTotalMax) bandTotalMax = tot;
        }
        return nLoc * bandTotalMax + regroup * (N * wcbMax + NExtended * wcbMax) + neLoc * bandTotalMax;
    }
    // ntt.extendPol(cmX_2ns, cmX_n, ...) + treesGL[i]->merkelize() (starks.cpp:53-60,134-141,215-222) over the ranks: the LDE runs per
    // COLUMN (all N rows of my W / G columns), the leaf sponge per ROW (all W columns of my NExtended / G rows):
    //   rows -> columns all-to-all | extendPol of my columns | columns -> rows all-to-all | sub-tree | cap (32 bytes per rank)
    // The two exchanges and the LDE between them run in BANDS of columns (every rank's span is cut into the same number of bands) so that
    // the staging buffers stay below XLP_SHARD_SCRATCH_MB (default 24576) whatever the section: two sets of (2 + 2 * 2^ext) x rows x columns
    // words.  The exchanges and the pack / unpack copies around them are enqueued on the library's stream 1, extendPol on stream 0
    // (xlp_stream_select), ordered by marks: band k + 1 travels to its column owners while band k is extended, band k's extended columns
    // travel back to their row owners while band k + 1 is extended.  Every rank knows every rank's band widths, so the exchanges are given
    // their longest message and never wait for the host.
    // h_src != NULL (step 1): my rows of the section are still in HOST memory (the executor's output): their columns are uploaded band by band
    // on the copy stream, band k + 1 crossing PCIe while band k is exchanged and extended (at least 4 bands then)
    // COLUMN-MAJOR windows: the same bands, marks and streams.  What travels is runs of nLoc words -- my rows of one column (of one coset):
    //   front: block p = [band columns of rank p][nLoc] out of the section; what arrives from p, [my columns][nLoc], is regrouped into whole
    //          columns [my columns][N]                                                         (xlp_dev_copy_3d)
    //   xlp_dev_extend_cm: [my columns][2^ext cosets][N]
    //   back:  block p = [my columns][2^ext][rows of rank p]; what arrives from p, [its band columns][2^ext][nLoc], goes to the coset blocks
    //          of those columns of the section
    // h_src: the band's columns of my rows cross PCIe into the row-major image `stage` (nLoc x W words) and are transposed from there
    void commitShardedCm(typename H::Tree* tree, eSection sn, eSection s2, const uint64_t* h_src = NULL, uint64_t* stage = NULL)
    {
        const uint64_t W = starkInfo.mapSectionsN.section[sn], E = 1ULL << extendBits, cs = colStride(sn), cs2 = colStride(s2);
        tree->d_source = section(s2);
        if (W) {
            uint64_t nb, bandTotalMax, wcbMax;
            const uint64_t setWords = commitBandPlan(W, nb, bandTotalMax, wcbMax, h_src ? 4 : 1);
            auto band = [&](int r, uint64_t k, uint64_t& a, uint64_t& b) { bandSpan(W, nb, k, r, a, b); };
            auto widest = [&](uint64_t k) { uint64_t m = 0; for (int p_ = 0; p_ < world; p_++) { uint64_t a, b; band(p_, k, a, b); if (b - a > m) m = b - a; } return m; };
            uint64_t* base = shardScratch(2 * setWords);
            struct Set { uint64_t *pack, *arrv, *colb, *extb, *send, *rowb; } S[2];
            for (int i = 0; i < 2; i++) {
                S[i].pack = base + (uint64_t)i * setWords;
                S[i].arrv = S[i].pack + nLoc * bandTotalMax; S[i].colb = S[i].arrv + N * wcbMax; S[i].extb = S[i].colb + N * wcbMax;
                S[i].send = S[i].extb + NExtended * wcbMax; S[i].rowb = S[i].send + NExtended * wcbMax;
            }
            std::vector<uint64_t*> rp(world);
            std::vector<const uint64_t*> spc(world);
            std::vector<uint64_t> rw(world), sw(world);
            enum { M_FRONT = 0, M_EXT = 2, M_BACK = 4 };
            auto upload = [&](uint64_t k) {
                for (int p_ = 0; p_ < world; p_++) {
                    uint64_t a, b; band(p_, k, a, b);
                    if (b > a) XLP_CHECK(xlp_dev_upload_cols_async(stage, h_src, W, a, b - a, nLoc));
                }
            };
            auto front = [&](uint64_t k) {
                Set& B = S[k & 1];
                if (h_src) {
                    XLP_CHECK(xlp_dev_upload_fence());
                    if (k + 1 < nb) upload(k + 1);
                    for (int p_ = 0; p_ < world; p_++) {                 // the band's columns of the image -> columns of the section
                        uint64_t a, b; band(p_, k, a, b);
                        if (b > a) XLP_CHECK(xlp_dev_rows_to_cm(section(sn) + a * cs, cs, 0, stage, W, a, b - a, nLoc, 0));
                    }
                }
                if (k >= 2) XLP_CHECK(xlp_stream_wait_mark(1, M_EXT + (int)(k & 1)));
                uint64_t ma, mb; band(rank, k, ma, mb);
                const uint64_t wc = mb - ma;
                uint64_t o = 0;
                for (int p_ = 0; p_ < world; p_++) {
                    uint64_t a, b; band(p_, k, a, b);
                    uint64_t* blk = B.pack + o;
                    spc[p_] = blk; sw[p_] = nLoc * (b - a);
                    if (b > a) XLP_CHECK(xlp_dev_copy_3d(blk, nLoc, 0, section(sn) + a * cs, cs, 0, b - a, 1, nLoc));
                    o += sw[p_];
                    rp[p_] = B.arrv + (uint64_t)p_ * nLoc * wc; rw[p_] = nLoc * wc;
                }
                XLP_CHECK(xlp_comm_exchange(rp.data(), rw.data(), spc.data(), sw.data(), nLoc * widest(k)));
                for (int p_ = 0; p_ < world && wc; p_++)                 // rows [p nLoc, (p + 1) nLoc) of my columns
                    XLP_CHECK(xlp_dev_copy_3d(B.colb + (uint64_t)p_ * nLoc, N, 0, rp[p_], nLoc, 0, wc, 1, nLoc));
                XLP_CHECK(xlp_stream_mark(1, M_FRONT + (int)(k & 1)));
            };
            auto back = [&](uint64_t k) {
                Set& B = S[k & 1];
                XLP_CHECK(xlp_stream_wait_mark(1, M_EXT + (int)(k & 1)));
                uint64_t ma, mb; band(rank, k, ma, mb);
                const uint64_t wc = mb - ma;
                uint64_t o = 0;
                for (int p_ = 0; p_ < world; p_++) {
                    uint64_t a, b; band(p_, k, a, b);
                    uint64_t* blk = B.send + (uint64_t)p_ * neLoc * wc;
                    if (wc) XLP_CHECK(xlp_dev_copy_3d(blk, E * nLoc, nLoc, B.extb + (uint64_t)p_ * nLoc, NExtended, N, wc, E, nLoc));
                    spc[p_] = blk; sw[p_] = neLoc * wc;
                    rp[p_] = B.rowb + o; rw[p_] = neLoc * (b - a);
                    o += rw[p_];
                }
                XLP_CHECK(xlp_comm_exchange(rp.data(), rw.data(), spc.data(), sw.data(), neLoc * widest(k)));
                for (int p_ = 0; p_ < world; p_++) {
                    uint64_t a, b; band(p_, k, a, b);
                    if (b > a) XLP_CHECK(xlp_dev_copy_3d(section(s2) + a * cs2, cs2, cosStride(), rp[p_], E * nLoc, nLoc, b - a, E, nLoc));
                }
                XLP_CHECK(xlp_stream_mark(1, M_BACK + (int)(k & 1)));
            };
            // the leaf sponge of my rows takes band k (whole blocks of 8 columns, in column order) once it has come back: on stream 0, after the mark of
            // back(k) -- between the extendPol of later bands, beside their exchanges (until round 4: the whole sponge after the last exchange)
            const bool sponge = W > 8;
            uint64_t firstBand = nb, lastBand = 0;
            for (uint64_t k = 0; k < nb; k++) { uint64_t a0, b0, a1, b1; band(0, k, a0, b0); band(world - 1, k, a1, b1); if (b1 > a0) { if (firstBand == nb) firstBand = k; lastBand = k; } }
            auto absorb = [&](uint64_t k) {
                if (!sponge) return;
                uint64_t a0, b0, a1, b1; band(0, k, a0, b0); band(world - 1, k, a1, b1);
                if (b1 <= a0) return;
                XLP_CHECK(xlp_stream_wait_mark(0, M_BACK + (int)(k & 1)));
                tree->absorbBand(a0, b1 - a0, k == firstBand, k == lastBand);
            };
            XLP_CHECK(xlp_stream_wait(1, 0));
            XLP_CHECK(xlp_stream_select(1));
            if (h_src) { XLP_CHECK(xlp_dev_upload_begin()); upload(0); }
            front(0);
            for (uint64_t k = 0; k < nb; k++) {
                uint64_t ma, mb; band(rank, k, ma, mb);
                XLP_CHECK(xlp_stream_select(0));
                XLP_CHECK(xlp_stream_wait_mark(0, M_FRONT + (int)(k & 1)));
                if (k >= 2) XLP_CHECK(xlp_stream_wait_mark(0, M_BACK + (int)(k & 1)));
                if (mb > ma) XLP_CHECK(xlp_dev_extend_cm(S[k & 1].extb, NExtended, 0, S[k & 1].colb, N, NExtended, N, mb - ma));
                XLP_CHECK(xlp_stream_mark(0, M_EXT + (int)(k & 1)));
                if (k >= 1) absorb(k - 1);                                   // band k - 1 is back in my rows: its leaf sponge behind band k's extendPol
                XLP_CHECK(xlp_stream_select(1));
                if (k + 1 < nb) front(k + 1);
                back(k);
            }
            XLP_CHECK(xlp_stream_select(0));
            absorb(nb - 1);
            XLP_CHECK(xlp_stream_wait(0, 1));
            if (sponge) { tree->finishFromLeaves(); fillHalo(s2); return; }
        }
        tree->merkelize();
        fillHalo(s2);
    }
    void commitSharded(typename H::Tree* tree, eSection sn, eSection s2, const uint64_t* h_src = NULL)
    {
        if (colMajor) { commitShardedCm(tree, sn, s2, h_src, NULL); return; }
        const uint64_t W = starkInfo.mapSectionsN.section[sn];
        tree->d_source = section(s2);
        if (W) {
            uint64_t nb, bandTotalMax, wcbMax;
            const uint64_t setWords = commitBandPlan(W, nb, bandTotalMax, wcbMax, h_src ? 4 : 1);
            // band k of a rank's span [a, b): [a + (b - a) k / nb, a + (b - a) (k + 1) / nb)
            auto band = [&](int r, uint64_t k, uint64_t& a, uint64_t& b) { bandSpan(W, nb, k, r, a, b); };
            auto widest = [&](uint64_t k) { uint64_t m = 0; for (int p_ = 0; p_ < world; p_++) { uint64_t a, b; band(p_, k, a, b); if (b - a > m) m = b - a; } return m; };
            uint64_t* base = shardScratch(2 * setWords);
            struct Set { uint64_t *pack, *colb, *extb, *rowb; } S[2];
            for (int i = 0; i < 2; i++) {
                S[i].pack = base + (uint64_t)i * setWords;
                S[i].colb = S[i].pack + nLoc * bandTotalMax; S[i].extb = S[i].colb + N * wcbMax; S[i].rowb = S[i].extb + NExtended * wcbMax;
            }
            std::vector<uint64_t*> rp(world);
            std::vector<const uint64_t*> spc(world);
            std::vector<uint64_t> rw(world), sw(world);
            enum { M_FRONT = 0, M_EXT = 2, M_BACK = 4 };                 // marks (+ set): band exchanged to columns / extended / exchanged back
            auto upload = [&](uint64_t k) {                              // band k's columns (of every rank's span) of my rows: host -> section
                for (int p_ = 0; p_ < world; p_++) {
                    uint64_t a, b; band(p_, k, a, b);
                    if (b > a) XLP_CHECK(xlp_dev_upload_cols_async(section(sn), h_src, W, a, b - a, nLoc));
                }
            };
            // stream 1: rows -> columns.  Block p = my rows of rank p's band columns, packed; what arrives from p is rows [p nLoc, (p + 1) nLoc) of mine
            auto front = [&](uint64_t k) {
                Set& B = S[k & 1];
                if (h_src) {
                    XLP_CHECK(xlp_dev_upload_fence());                   // band k (and everything before it) is there ...
                    if (k + 1 < nb) upload(k + 1);                        // ... and the next one starts while this one is worked on
                }
                if (k >= 2) XLP_CHECK(xlp_stream_wait_mark(1, M_EXT + (int)(k & 1)));      // extendPol of band k - 2 has read this set's columns
                uint64_t ma, mb; band(rank, k, ma, mb);
                const uint64_t wc = mb - ma;                              // my columns of this band
                uint64_t o = 0;
                for (int p_ = 0; p_ < world; p_++) {
                    uint64_t a, b; band(p_, k, a, b);
                    uint64_t* blk = B.pack + o;
                    spc[p_] = blk; sw[p_] = nLoc * (b - a);
                    if (b > a) XLP_CHECK(xlp_dev_copy_cols(blk, b - a, section(sn) + a, W, nLoc, (uint32_t)(b - a)));
                    o += sw[p_];
                    rp[p_] = B.colb + (uint64_t)p_ * nLoc * wc; rw[p_] = nLoc * wc;
                }
                XLP_CHECK(xlp_comm_exchange(rp.data(), rw.data(), spc.data(), sw.data(), nLoc * widest(k)));
                XLP_CHECK(xlp_stream_mark(1, M_FRONT + (int)(k & 1)));
            };
            // stream 1: columns -> rows.  The rows destined to rank p are contiguous in extb; what arrives from p is its band columns of my rows
            auto back = [&](uint64_t k) {
                Set& B = S[k & 1];
                XLP_CHECK(xlp_stream_wait_mark(1, M_EXT + (int)(k & 1)));
                uint64_t ma, mb; band(rank, k, ma, mb);
                const uint64_t wc = mb - ma;
                uint64_t o = 0;
                for (int p_ = 0; p_ < world; p_++) {
                    uint64_t a, b; band(p_, k, a, b);
                    spc[p_] = B.extb + (uint64_t)p_ * neLoc * wc; sw[p_] = neLoc * wc;
                    rp[p_] = B.rowb + o; rw[p_] = neLoc * (b - a);
                    o += rw[p_];
                }
                XLP_CHECK(xlp_comm_exchange(rp.data(), rw.data(), spc.data(), sw.data(), neLoc * widest(k)));
                for (int p_ = 0; p_ < world; p_++) {
                    uint64_t a, b; band(p_, k, a, b);
                    if (b > a) XLP_CHECK(xlp_dev_copy_cols(section(s2) + a, W, rp[p_], b - a, neLoc, (uint32_t)(b - a)));
                }
                XLP_CHECK(xlp_stream_mark(1, M_BACK + (int)(k & 1)));
            };
            // the leaf sponge of my rows takes band k (whole blocks of 8 columns, in column order) once it has come back: on stream 0, after the mark of
            // back(k) -- between the extendPol of later bands, beside their exchanges (until round 4: the whole sponge after the last exchange)
            const bool sponge = W > 8;
            uint64_t firstBand = nb, lastBand = 0;
            for (uint64_t k = 0; k < nb; k++) { uint64_t a0, b0, a1, b1; band(0, k, a0, b0); band(world - 1, k, a1, b1); if (b1 > a0) { if (firstBand == nb) firstBand = k; lastBand = k; } }
            auto absorb = [&](uint64_t k) {
                if (!sponge) return;
                uint64_t a0, b0, a1, b1; band(0, k, a0, b0); band(world - 1, k, a1, b1);
                if (b1 <= a0) return;
                XLP_CHECK(xlp_stream_wait_mark(0, M_BACK + (int)(k & 1)));
                tree->absorbBand(a0, b1 - a0, k == firstBand, k == lastBand);
            };
            XLP_CHECK(xlp_stream_wait(1, 0));                             // the section is what stream 0 has made of it so far
            XLP_CHECK(xlp_stream_select(1));
            if (h_src) { XLP_CHECK(xlp_dev_upload_begin()); upload(0); }
            front(0);
            for (uint64_t k = 0; k < nb; k++) {
                uint64_t ma, mb; band(rank, k, ma, mb);
                XLP_CHECK(xlp_stream_select(0));
                XLP_CHECK(xlp_stream_wait_mark(0, M_FRONT + (int)(k & 1)));               // my columns of band k have arrived
                if (k >= 2) XLP_CHECK(xlp_stream_wait_mark(0, M_BACK + (int)(k & 1)));    // band k - 2's extended columns have left this set
                if (mb > ma) XLP_CHECK(xlp_dev_extend(S[k & 1].extb, S[k & 1].colb, NExtended, N, mb - ma, NULL));
                XLP_CHECK(xlp_stream_mark(0, M_EXT + (int)(k & 1)));
                if (k >= 1) absorb(k - 1);
                XLP_CHECK(xlp_stream_select(1));
                if (k + 1 < nb) front(k + 1);
                back(k);
            }
            XLP_CHECK(xlp_stream_select(0));
            absorb(nb - 1);
            XLP_CHECK(xlp_stream_wait(0, 1));
            if (sponge) { tree->finishFromLeaves(); fillHalo(s2); return; }
        }
        tree->merkelize();                                    // my sub-tree + the cap
        fillHalo(s2);
    }
    void runStep(Steps* steps, int k, StepsParams& P, uint64_t rows)
    {
        switch (k) {
        case 0: steps->step2prev_parser_first_avx(P, rows, nrowsStepBatch); break;
        case 1: steps->step3prev_parser_first_avx(P, rows, nrowsStepBatch); break;
        case 2: steps->step3_parser_first_avx(P, rows, nrowsStepBatch); break;
        case 3: steps->step42ns_parser_first_avx(P, rows, nrowsStepBatch); break;
        default: steps->step52ns_parser_first_avx(P, rows, nrowsStepBatch); break;
        }
        haloWriteback(steps->nextRowStores(k));
    }

    // Starks::genProof (starks.cpp:9-404) on this rank's rows; the stage order, timer names and transcript sequence are genProof's.
    void genProofSharded(typename H::Proof& proof, Goldilocks::Element* publicInputs, Goldilocks::Element verkey[4], Steps* steps)
    {
        lastTimers.clear();
        lastExchanges.clear();
        XLP_CHECK(xlp_comm_stats(&commSeconds, &commBytes, &commCalls, 0));
        if (steps->colMajor() != colMajor) xlp_exit_process("Starks: the step programs and the starkinfo disagree on the layout of the sections (colMajor)");
        if (XlpDeviceSlab::get().reserve(slabWords) == NULL) xlp_exit_process("Starks: no device slab");
        const eSection secs2ns[4] = {cm1_2ns, cm2_2ns, cm3_2ns, cm4_2ns};
        for (int i = 0; i < 4; i++) treesGL[i]->d_source = section(secs2ns[i]);
        treesGL[4]->d_source = constPols2ns();
        const uint64_t W1 = starkInfo.mapSectionsN.section[cm1_n];
        uint64_t numCommited = starkInfo.nCm1;
        typename H::TranscriptT transcript;
        std::vector<Goldilocks::Element> challenges(NUM_CHALLENGES * FIELD_EXTENSION, Goldilocks::zero());
        std::vector<Goldilocks::Element> evals(starkInfo.evMap.size() * FIELD_EXTENSION, Goldilocks::zero());
        typename H::Root root0, root1, root2, root3;
        auto challenge = [&](int k) {
            H::getField(transcript, &challenges[k * FIELD_EXTENSION]);
            XLP_CHECK(xlp_copy_h2d(d_challenges, challenges.data(), challenges.size() * sizeof(uint64_t)));
        };
        {
            const uint64_t from = starkInfo.mapOffsets.section[cm2_n], to = starkInfo.mapOffsets.section[tmpExp_n] + (nLoc + haloN) * starkInfo.mapSectionsN.section[tmpExp_n];
            if (to > from && to <= slabWords) XLP_CHECK(xlp_dev_memset(mem() + from, 0, (to - from) * sizeof(uint64_t)));
        }
        XLP_CHECK(xlp_copy_h2d(d_publics, publicInputs, starkInfo.nPublics * sizeof(uint64_t)));
        H::putVerkey(transcript, verkey, constRoot);
        transcript.put(&publicInputs[0], starkInfo.nPublics);
        StepsParams P = params();
        //-------------------------------- 1.- my rows of the witness (pAddress holds THEM), halo, commit
        {
            Timer t(this, "STARK_STEP_1_LDE_AND_MERKLETREE");
            const uint64_t* h_cm1 = (const uint64_t*)pAddress + starkInfo.mapOffsets.section[cm1_n];
            if (colMajor) {
                // my rows arrive as the executor wrote them (row-major): their image lands in the still dead base-domain sections behind
                // cm1_n (or in a buffer of its own when they are too small) and is transposed from there, band by band when pipelined
                const uint64_t from = starkInfo.mapOffsets.section[cm2_n], to = starkInfo.mapOffsets.section[tmpExp_n] + (nLoc + haloN) * starkInfo.mapSectionsN.section[tmpExp_n];
                const bool inSections = to > from && to <= slabWords && to - from >= nLoc * W1;
                uint64_t* stage = inSections ? mem() + from : dalloc(nLoc * W1);
                if (W1 && pipelineWitness) commitShardedCm(treesGL[0], cm1_n, cm1_2ns, h_cm1, stage);
                else {
                    XLP_CHECK(xlp_copy_h2d(stage, h_cm1, nLoc * W1 * sizeof(uint64_t)));
                    if (W1) XLP_CHECK(xlp_dev_rows_to_cm(section(cm1_n), colStride(cm1_n), 0, stage, W1, 0, W1, nLoc, 0));
                    commitShardedCm(treesGL[0], cm1_n, cm1_2ns);
                }
                if (inSections) XLP_CHECK(xlp_dev_memset(stage, 0, nLoc * W1 * sizeof(uint64_t)));
                else { XLP_CHECK(xlp_sync()); xlp_dev_free(stage); }
            } else if (W1 && pipelineWitness) commitSharded(treesGL[0], cm1_n, cm1_2ns, h_cm1);
            else {
                XLP_CHECK(xlp_copy_h2d(section(cm1_n), h_cm1, nLoc * W1 * sizeof(uint64_t)));
                commitSharded(treesGL[0], cm1_n, cm1_2ns);
            }
            fillHalo(cm1_n);
            treesGL[0]->getRoot(root0);
            H::putRoot(transcript, root0);
        }
        //-------------------------------- 2.- plookups
        challenge(0);
        challenge(1);
        {
            Timer t(this, "STARK_STEP_2_CALCULATE_EXPS");
            runStep(steps, 0, P, nLoc);
        }
        {
            // calculateH1H2 is a global sort-merge of f into t; both are small next to the trace (2 x N x dim words per lookup), so the
            // table is all-gathered and every rank computes the whole h1 / h2 and keeps its rows
            Timer t(this, "STARK_STEP_2_CALCULATEH1H2");
            for (uint64_t i = 0; i < starkInfo.puCtx.size(); i++) {
                const PolRef f = pol(exp2pol(starkInfo.puCtx[i].fExpId)), tt = pol(exp2pol(starkInfo.puCtx[i].tExpId));
                const PolRef h1 = pol(starkInfo.cm_n[numCommited + i * 2]), h2 = pol(starkInfo.cm_n[numCommited + i * 2 + 1]);
                if (f.dim != tt.dim || h1.dim != f.dim || h2.dim != f.dim) xlp_exit_process("Starks: plookup polynomials of different dimensions");
                const uint64_t d = f.dim;
                uint64_t *tLoc = d_work, *tFull = d_work + N * 3, *fFull = d_work + 2 * N * 3, *hh1 = d_work + 3 * N * 3, *hh2 = d_work + 4 * N * 3, *fLoc = d_work + 5 * N * 3;
                toDense(tLoc, tt, nLoc);                     // (row-major sections: a strided copy; column-major: the d columns interleaved)
                XLP_CHECK(xlp_comm_all_gather(tFull, tLoc, nLoc * d));
                if (config.xlpSyntheticLookups) XLP_CHECK(xlp_dev_gather_rows(fFull, tFull, d, N, (uint32_t)d, 0x5EED0500ULL + i));
                else {
                    toDense(fLoc, f, nLoc);
                    XLP_CHECK(xlp_comm_all_gather(fFull, fLoc, nLoc * d));
                }
                XLP_CHECK(xlp_dev_calculate_h1h2(hh1, hh2, fFull, tFull, N, (uint32_t)d));
                fromDense(h1, hh1 + row0 * d, nLoc);
                fromDense(h2, hh2 + row0 * d, nLoc);
            }
            numCommited += starkInfo.puCtx.size() * 2;
        }
        fillHalo(tmpExp_n); fillHalo(cm2_n); fillHalo(cm3_n);
        {
            Timer t(this, "STARK_STEP_2_LDE_AND_MERKLETREE");
            commitSharded(treesGL[1], cm2_n, cm2_2ns);
            treesGL[1]->getRoot(root1);
            H::putRoot(transcript, root1);
        }
        //-------------------------------- 3.- grand products
        challenge(2);
        challenge(3);
        {
            Timer t(this, "STARK_STEP_3_CALCULATE_EXPS");
            runStep(steps, 1, P, nLoc);
        }
        {
            // calculateZ (polinomial.hpp:586-607) is an exclusive prefix product over the rows: a local scan over my rows AND one more (the
            // extra output, in the halo row, is the product of all my ratios; what is read there does not enter it), one all-gather of
            // that Fp^3 element, and a scaling of my rows by the product of the ranks before me
            Timer t(this, "STARK_STEP_3_CALCULATE_Z");
            std::vector<std::pair<uint64_t, uint64_t>> nd;
            for (const PuCtx& c : starkInfo.puCtx) nd.emplace_back(c.numId, c.denId);
            for (const PeCtx& c : starkInfo.peCtx) nd.emplace_back(c.numId, c.denId);
            for (const CiCtx& c : starkInfo.ciCtx) nd.emplace_back(c.numId, c.denId);
            std::vector<uint64_t> totals(3 * nd.size()), allTotals((size_t)world * 3 * nd.size());
            for (uint64_t i = 0; i < nd.size(); i++) {
                const PolRef num = pol(exp2pol(nd[i].first)), den = pol(exp2pol(nd[i].second)), z = pol(starkInfo.cm_n[numCommited + i]);
                if (num.dim != 3 || den.dim != 3 || z.dim != 3) xlp_exit_process("Starks: Z polynomials must be ext-3");
                if (colMajor) {
                    uint64_t *dn = d_work, *dd = d_work + (nLoc + 1) * 3, *dz = d_work + 2 * (nLoc + 1) * 3;
                    toDense(dn, num, nLoc + 1); toDense(dd, den, nLoc + 1);
                    XLP_CHECK(xlp_dev_calculate_z(dz, dn, dd, nLoc + 1));
                    fromDense(z, dz, nLoc + 1);
                    XLP_CHECK(xlp_copy_d2h(&totals[3 * i], dz + nLoc * 3, 3 * sizeof(uint64_t)));
                    continue;
                }
                XLP_CHECK(xlp_dev_calculate_z_strided(z.p, z.stride, num.p, num.stride, den.p, den.stride, nLoc + 1));
                XLP_CHECK(xlp_copy_d2h(&totals[3 * i], z.p + nLoc * z.stride, 3 * sizeof(uint64_t)));
            }
            if (!nd.empty()) XLP_CHECK(xlp_comm_all_gather_host(allTotals.data(), totals.data(), totals.size()));
            for (uint64_t i = 0; i < nd.size() && rank > 0; i++) {
                Goldilocks3::Element prefix;
                Goldilocks3::one(prefix);
                for (int r = 0; r < rank; r++) {
                    Goldilocks3::Element tr_ = {Goldilocks::fromU64(allTotals[(size_t)r * totals.size() + 3 * i]), Goldilocks::fromU64(allTotals[(size_t)r * totals.size() + 3 * i + 1]),
                                                Goldilocks::fromU64(allTotals[(size_t)r * totals.size() + 3 * i + 2])};
                    Goldilocks3::mul(prefix, prefix, tr_);
                }
                const uint64_t pw[3] = {Goldilocks::toU64(prefix[0]), Goldilocks::toU64(prefix[1]), Goldilocks::toU64(prefix[2])};
                const PolRef z = pol(starkInfo.cm_n[numCommited + i]);
                if (colMajor) {
                    toDense(d_work, z, nLoc);
                    XLP_CHECK(xlp_dev_scale3_strided(d_work, 3, nLoc, pw));
                    fromDense(z, d_work, nLoc);
                } else
                XLP_CHECK(xlp_dev_scale3_strided(z.p, z.stride, nLoc, pw));
            }
        }
        fillHalo(tmpExp_n); fillHalo(cm3_n);
        {
            Timer t(this, "STARK_STEP_3_CALCULATE_EXPS_2");
            runStep(steps, 2, P, nLoc);
        }
        {
            Timer t(this, "STARK_STEP_3_LDE_AND_MERKLETREE");
            commitSharded(treesGL[2], cm3_n, cm3_2ns);
            treesGL[2]->getRoot(root2);
            H::putRoot(transcript, root2);
        }
        //-------------------------------- 4. the quotient: qDim columns, all-gathered and transformed on every rank; my rows of cm4_2ns kept
        challenge(4);
        if (colMajor) { P.coset_rows = nLoc; P.coset_stride = cosStride(); }   // the extended-domain programs run coset block by coset block
        {
            Timer t(this, "STARK_STEP_4_CALCULATE_EXPS_2NS");
            runStep(steps, 3, P, neLoc);
        }
        const uint64_t W4 = starkInfo.qDim * starkInfo.qDeg;
        {
            Timer t(this, "STARK_STEP_4_CALCULATE_EXPS_2NS_INTT_MUL_NTT");
            uint64_t *qfull = d_work, *qq2 = d_work + NExtended * starkInfo.qDim;
            XLP_CHECK(xlp_comm_all_gather(qfull, section(q_2ns), neLoc * starkInfo.qDim));
            XLP_CHECK(xlp_dev_ntt(qfull, qfull, NExtended, starkInfo.qDim, 1));
            const Goldilocks::Element shiftIn = Goldilocks::exp(Goldilocks::inv(Goldilocks::shift()), N);
            XLP_CHECK(xlp_dev_q_split(qq2, qfull, N, NExtended, (uint32_t)starkInfo.qDeg, (uint32_t)starkInfo.qDim, Goldilocks::toU64(shiftIn)));
            XLP_CHECK(xlp_dev_ntt(qq2, qq2, NExtended, W4, 0));
            if (colMajor) XLP_CHECK(xlp_dev_rows_to_cm(section(cm4_2ns), colStride(cm4_2ns), cosStride(), qq2 + ext0 * W4, W4, 0, W4, neLoc, (uint32_t)extendBits));
            else
            XLP_CHECK(xlp_dev_copy_cols(section(cm4_2ns), W4, qq2 + ext0 * W4, W4, neLoc, (uint32_t)W4));
        }
        {
            Timer t(this, "STARK_STEP_4_MERKLETREE");
            treesGL[3]->merkelize();
            fillHalo(cm4_2ns);
            treesGL[3]->getRoot(root3);
            H::putRoot(transcript, root3);
        }
        //-------------------------------- 5. evaluations and the FRI polynomial
        challenge(7);
        const Goldilocks::Element* xi = &challenges[7 * FIELD_EXTENSION];
        uint64_t wxi[3];
        {
            Timer t(this, "STARK_STEP_5_LEv_LpEv");        // replicated: N x 3 words each
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
            // evmap (starks.cpp:556-669) is a sum over the rows: my rows' partial sums, one all-gather of nev x 3 words, the field sum on the host
            Timer t(this, "STARK_STEP_5_EVMAP");
            std::vector<xlp_evmap_entry> ents(starkInfo.evMap.size());
            for (size_t i = 0; i < ents.size(); i++) {
                const EvMap& ev = starkInfo.evMap[i];
                if (ev.type == EvMap::_const) ents[i] = {d_const_2ns_win + ev.id, starkInfo.nConstants, 1, ev.prime ? 1u : 0u};
                else {
                    const PolRef p = pol(ev.type == EvMap::cm ? starkInfo.cm_2ns[ev.id] : starkInfo.qs[ev.id]);
                    ents[i] = {p.p, p.stride, (uint32_t)p.dim, ev.prime ? 1u : 0u};
                }
            }
            if (colMajor) {                                  // coset block 0 of a window column = my rows of the base domain
                for (size_t i = 0; i < ents.size(); i++) {
                    const EvMap& ev = starkInfo.evMap[i];
                    if (ev.type == EvMap::_const) ents[i] = {d_const_2ns_win + ev.id * (neLoc + haloExt), neLoc + haloExt, 1, ev.prime ? 1u : 0u};
                    else {
                        const PolRef p = pol(ev.type == EvMap::cm ? starkInfo.cm_2ns[ev.id] : starkInfo.qs[ev.id]);
                        ents[i] = {p.p, p.cs, (uint32_t)p.dim, ev.prime ? 1u : 0u};
                    }
                }
                XLP_CHECK(xlp_dev_evmap_cm(d_evals, d_work + 3 * row0, d_work + N * 3 + 3 * row0, nLoc, (uint32_t)ents.size(), ents.data()));
            } else
            XLP_CHECK(xlp_dev_evmap(d_evals, d_work + 3 * row0, d_work + N * 3 + 3 * row0, nLoc, (uint32_t)extendBits, (uint32_t)ents.size(), ents.data()));
            std::vector<uint64_t> part(evals.size()), all((size_t)world * evals.size());
            XLP_CHECK(xlp_copy_d2h(part.data(), d_evals, part.size() * sizeof(uint64_t)));
            XLP_CHECK(xlp_comm_all_gather_host(all.data(), part.data(), part.size()));
            for (size_t k = 0; k < evals.size(); k++) {
                Goldilocks::Element acc = Goldilocks::zero();
                for (int r = 0; r < world; r++) acc = Goldilocks::add(acc, Goldilocks::fromU64(all[(size_t)r * evals.size() + k]));
                evals[k] = acc;
            }
            XLP_CHECK(xlp_copy_h2d(d_evals, evals.data(), evals.size() * sizeof(uint64_t)));
        }
        transcript.put(evals.data(), starkInfo.evMap.size() * FIELD_EXTENSION);   // (starks.cpp:336-339 puts them three at a time: the same element sequence; one call lets the sponge take its full blocks in one launch)
        challenge(5);
        challenge(6);
        {
            Timer t(this, "STARK_STEP_5_XDIVXSUB");
            XLP_CHECK(xlp_dev_x_div_x_sub_xi(d_xdx, d_x_2ns, xlp_raw(xi), neLoc));
            XLP_CHECK(xlp_dev_x_div_x_sub_xi(d_xdwx, d_x_2ns, wxi, neLoc));
        }
        {
            Timer t(this, "STARK_STEP_5_CALCULATE_EXPS");
            runStep(steps, 4, P, neLoc);
        }
        {
            // FRI on the gathered polynomial (3 columns), on every rank: every rank follows the transcript and ends with the whole proof;
            // the step-0 openings of the four sharded trees come from the ranks that own the rows (MerkleTreeGL::getGroupProofs)
            Timer t(this, "STARK_STEP_FRI");
            XLP_CHECK(xlp_comm_all_gather(d_work, section(f_2ns), neLoc * FIELD_EXTENSION));
            Polinomial friPol(NExtended >> (starkInfo.starkStruct.nBitsExt - starkInfo.starkStruct.steps.back().nBits), FIELD_EXTENSION, "friPol");
            H::Prove::prove(proof, treesGL, transcript, friPol, starkInfo.starkStruct.nBitsExt, starkInfo, d_work);
        }
        proof.proofs.setEvals(evals.data());
        H::storeRoots(proof, root0, root1, root2, root3);
    }

public:
    StarksT(const Config& config, StarkFiles starkFiles, void* _pAddress)
        : config(config), starkInfo(starkFiles.zkevmStarkInfo), nrowsStepBatch(1), starkFiles(starkFiles), N(0), NExtended(0), extendBits(0), pAddress(_pAddress)
    {
        if (!config.generateProof()) return;      // starks.hpp:91-93: nothing is initialised if no proof will be generated
        residentConstants = config.xlpResidentConstants;
        N = 1ULL << starkInfo.starkStruct.nBits;
        NExtended = 1ULL << starkInfo.starkStruct.nBitsExt;
        extendBits = starkInfo.starkStruct.nBitsExt - starkInfo.starkStruct.nBits;
        if (starkInfo.mapSectionsN.section[q_2ns] != starkInfo.qDim || starkInfo.mapSectionsN.section[f_2ns] != FIELD_EXTENSION)
            xlp_exit_process("Starks: q_2ns / f_2ns sections must be qDim / 3 wide");
        if (starkInfo.mapSectionsN.section[cm4_2ns] != starkInfo.qDim * starkInfo.qDeg) xlp_exit_process("Starks: cm4_2ns must hold qDeg x qDim columns");
        XLP_CHECK(xlp_init((int)config.xlpDevice));
        const2nsInSlab = starkInfo.xlpConst2nsInSlab; const2nsOffset = starkInfo.xlpConst2nsOffset;
        colMajor = starkInfo.xlpColMajor;
        if (colMajor && !H::kCommitTimes) xlp_exit_process("Starks: column-major sections need the Goldilocks hash policy");
        slabWords = starkInfo.mapTotalN;
        XlpDeviceSlab::get().reserve(slabWords);
        if (starkInfo.xlpWorld) {                         // a window starkinfo: this Starks is one rank of a row-sharded proof
            if constexpr (!H::kShardable) xlp_exit_process("Starks: this hash policy has no row-sharded tree");
            world = xlp_comm_world(); rank = xlp_comm_rank();
            if ((uint64_t)world != starkInfo.xlpWorld) xlp_exit_process("Starks: the starkinfo is laid out for another number of ranks (xlp_comm_init first; \"xlp\".world)");
            sharded = true;
            nLoc = N / world; neLoc = NExtended / world; haloN = starkInfo.xlpHaloN; haloExt = starkInfo.xlpHaloExt;
            row0 = (uint64_t)rank * nLoc; ext0 = (uint64_t)rank * neLoc;
            if (nLoc < haloN || haloExt != (haloN << extendBits)) xlp_exit_process("Starks: shard smaller than its halo");
            // the sharded calculateZ scans one row past the rank's last and reads its total there: that row must be a halo row of the window
            if (haloN < 1 && (starkInfo.puCtx.size() || starkInfo.peCtx.size() || starkInfo.ciCtx.size())) xlp_exit_process("Starks: a window starkinfo with grand products needs a halo of at least one row");
        }
        // ---- constant polynomials (LOAD_CONST_POLS_TO_MEMORY, starks.hpp:96-115); sharded: replicated, the programs read a window of them
        if (starkFiles.zkevmConstPols.size() == 0) xlp_exit_process("Starks::Starks() received an empty config.zkevmConstPols");
        uint64_t* constFull = dalloc(N * starkInfo.nConstants);
        // column-major: the file / the synthetic fill is row-major (the same VALUES in both layouts): it lands in the still unused device slab
        // and is transposed from there
        const bool viaSlab = colMajor && slabWords >= NExtended * starkInfo.nConstants;
        uint64_t* ownLand = colMajor && !viaSlab ? dalloc(NExtended * starkInfo.nConstants) : NULL;      // (a rank's window slab can be smaller than the replicated constants)
        uint64_t* constLand = ownLand ? ownLand : colMajor ? mem() : constFull;
        if (starkFiles.zkevmConstPols.rfind("synthetic:", 0) == 0)
            XLP_CHECK(xlp_dev_fill(constLand, N * starkInfo.nConstants, strtoull(starkFiles.zkevmConstPols.c_str() + 10, NULL, 0), 0));
        else fileToDevice(constLand, starkFiles.zkevmConstPols, 0, N * starkInfo.nConstants * sizeof(uint64_t));
        if (colMajor) XLP_CHECK(xlp_dev_rows_to_cm(constFull, N, 0, constLand, starkInfo.nConstants, 0, starkInfo.nConstants, N, 0));
        if (sharded) {
            d_const_n_full = constFull;
            d_const_n = dalloc((nLoc + haloN) * starkInfo.nConstants);
            if (colMajor) {                               // my rows of every column, then the halo rows one by one (the last rank's wrap to row 0)
                XLP_CHECK(xlp_dev_copy_3d(d_const_n, cosStride(), 0, constFull + row0, N, 0, starkInfo.nConstants, 1, nLoc));
                for (uint64_t i = 0; i < haloN; i++)
                    XLP_CHECK(xlp_dev_copy_3d(d_const_n + nLoc + i, cosStride(), 0, constFull + (row0 + nLoc + i) % N, N, 0, starkInfo.nConstants, 1, 1));
            } else
            windowRows(d_const_n, constFull, starkInfo.nConstants, row0, nLoc + haloN, N);
        } else d_const_n = constFull;
        // ---- constant tree (LOAD_CONST_TREE_TO_MEMORY / LOAD_CONST_POLS_2NS_TO_MEMORY, starks.hpp:117-146)
        if (!const2nsInSlab) d_const_2ns_own = dalloc(NExtended * starkInfo.nConstants);
        treesGL[4] = new typename H::Tree(NExtended, starkInfo.nConstants, constPols2ns(), true);
        if constexpr (H::kCommitTimes) { if (colMajor) treesGL[4]->setColMajor((uint32_t)extendBits, NExtended); }
        bool treeFromFile = false;
        if constexpr (H::kFileTree) treeFromFile = starkFiles.zkevmConstantsTree.size() && fileExists(starkFiles.zkevmConstantsTree);
        if (treeFromFile) {
          if constexpr (H::kFileTree) {
            // [width, height, source (height x width), nodes] -- build_const_tree.cpp:566-603, MerkleTreeGL(Goldilocks::Element*) merkleTreeGL.hpp:24-32
            uint64_t hdr[2] = {0, 0};
            FILE* fp = fopen(starkFiles.zkevmConstantsTree.c_str(), "rb");
            if (!fp || fread(hdr, 8, 2, fp) != 2) xlp_exit_process("Starks: cannot read the constant tree header");
            fclose(fp);
            if (hdr[0] != starkInfo.nConstants || hdr[1] != NExtended) xlp_exit_process("Starks: constant tree does not match the starkinfo");
            if (colMajor && !const2nsInSlab) {            // the file's source is row-major, natural order: through the slab, then coset-major columns
                fileToDevice(constLand, starkFiles.zkevmConstantsTree, 16, NExtended * starkInfo.nConstants * 8);
                XLP_CHECK(xlp_dev_rows_to_cm(constPols2ns(), NExtended, 0, constLand, starkInfo.nConstants, 0, starkInfo.nConstants, NExtended, (uint32_t)extendBits));
            } else if (colMajor) {                        // (the extended constants live IN the slab: recomputed from the base ones instead, as in every proof)
                XLP_CHECK(xlp_dev_extend_cm(constPols2ns(), NExtended, 0, constFull, N, NExtended, N, starkInfo.nConstants));
            } else fileToDevice(constPols2ns(), starkFiles.zkevmConstantsTree, 16, NExtended * starkInfo.nConstants * 8);
            fileToDevice(treesGL[4]->d_nodes, starkFiles.zkevmConstantsTree, 16 + NExtended * starkInfo.nConstants * 8, H::treeWords(treesGL[4]) * 8);
          }
        } else {
            if (colMajor) XLP_CHECK(xlp_dev_extend_cm(constPols2ns(), NExtended, 0, constFull, N, NExtended, N, starkInfo.nConstants));
            else XLP_CHECK(xlp_dev_extend(constPols2ns(), constFull, NExtended, N, starkInfo.nConstants, NULL));      // (sharded: d_const_n is only my window)
            treesGL[4]->merkelize();
        }
        treesGL[4]->getRoot(constRoot);
        if (ownLand) { XLP_CHECK(xlp_sync()); xlp_dev_free(ownLand); }
        if (sharded) {                                    // the extended constants and their tree are replicated; the 2ns programs read my window
            d_const_2ns_win = dalloc((neLoc + haloExt) * starkInfo.nConstants);
            if (colMajor) {                               // [column][coset][my rows + halo] out of [column][coset][N]
                const uint64_t E = 1ULL << extendBits;
                XLP_CHECK(xlp_dev_copy_3d(d_const_2ns_win, neLoc + haloExt, cosStride(), constPols2ns() + row0, NExtended, N, starkInfo.nConstants, E, nLoc));
                for (uint64_t i = 0; i < haloN; i++)
                    XLP_CHECK(xlp_dev_copy_3d(d_const_2ns_win + nLoc + i, neLoc + haloExt, cosStride(), constPols2ns() + (row0 + nLoc + i) % N, NExtended, N, starkInfo.nConstants, E, 1));
            } else
            windowRows(d_const_2ns_win, constPols2ns(), starkInfo.nConstants, ext0, neLoc + haloExt, NExtended);
        }
        // ---- x_n, x_2ns (COMPUTE_X_N_AND_X_2_NS, starks.hpp:149-160), ZhInv (zhInv.cpp:7-32); sharded: my rows of them
        const uint64_t ownN = sharded ? nLoc : N, ownE = sharded ? neLoc : NExtended;
        d_x_n = dalloc(ownN); d_x_2ns = dalloc(ownE);
        const Goldilocks::Element wnE = Goldilocks::w(starkInfo.starkStruct.nBits), wneE = Goldilocks::w(starkInfo.starkStruct.nBitsExt);
        const uint64_t one = Goldilocks::toU64(Goldilocks::exp(wnE, row0)), wn = Goldilocks::toU64(wnE),
                       sh = Goldilocks::toU64(Goldilocks::mul(Goldilocks::shift(), Goldilocks::exp(wneE, ext0))), wne = Goldilocks::toU64(wneE);
        XLP_CHECK(xlp_dev_geometric(d_x_n, ownN, 1, &one, &wn));
        if (colMajor) {                                   // coset-major: x of storage row r N + q = shift w_ext^r (w_N)^q
            for (uint64_t r = 0; r < (1ULL << extendBits); r++) {      // (sharded: my rows of every coset, densely)
                const uint64_t first = Goldilocks::toU64(Goldilocks::mul(Goldilocks::mul(Goldilocks::shift(), Goldilocks::exp(wneE, r)), Goldilocks::exp(wnE, row0)));
                XLP_CHECK(xlp_dev_geometric(d_x_2ns + r * ownN, ownN, 1, &first, &wn));
            }
        } else XLP_CHECK(xlp_dev_geometric(d_x_2ns, ownE, 1, &sh, &wne));
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
        d_xdx = dalloc(ownE * FIELD_EXTENSION); d_xdwx = dalloc(ownE * FIELD_EXTENSION);
        d_challenges = dalloc(NUM_CHALLENGES * FIELD_EXTENSION);
        d_publics = dalloc(starkInfo.nPublics);
        d_evals = dalloc(starkInfo.evMap.size() * FIELD_EXTENSION);
        d_work = dalloc(NExtended * (uint64_t)std::max<uint64_t>(4 * FIELD_EXTENSION, starkInfo.qDim * (starkInfo.qDeg + 1)));
        // ---- trees over the extended sections (MERKLE_TREE_ALLOCATION, starks.hpp:185-191): sources are offsets into the shared slab
        const eSection secs[4] = {cm1_2ns, cm2_2ns, cm3_2ns, cm4_2ns};
        if (sharded) {                                    // the staging scratch of the sharded commits, once, for the section that needs most
            uint64_t need = 0, nb, bt, wm;                // (growing it between sections costs a free + malloc of tens of GB: 1 s at 2^20 rows)
            const eSection cs[3] = {cm1_n, cm2_n, cm3_n};
            for (eSection c_ : cs) if (starkInfo.mapSectionsN.section[c_]) need = std::max(need, 2 * commitBandPlan(starkInfo.mapSectionsN.section[c_], nb, bt, wm, c_ == cm1_n ? 4 : 1));
            shardScratch(need);
        }
        for (int i = 0; i < 4; i++) {
            if constexpr (H::kShardable) treesGL[i] = new typename H::Tree(NExtended, starkInfo.mapSectionsN.section[secs[i]], NULL, true, sharded ? world : 0, rank);
            else treesGL[i] = new typename H::Tree(NExtended, starkInfo.mapSectionsN.section[secs[i]], NULL, true);
            if constexpr (H::kCommitTimes) { if (colMajor) treesGL[i]->setColMajor((uint32_t)extendBits, sharded ? neLoc + haloExt : NExtended, sharded ? cosStride() : 0); }
        }
        if (colMajor) {
            for (int i = 0; i < 11; i++) {
                const eSection sec = (eSection)i;
                const uint64_t want = sharded ? (isExtSection(sec) ? neLoc + haloExt : nLoc + haloN) : isExtSection(sec) ? NExtended : N;
                if (starkInfo.mapSectionsN.section[sec] && starkInfo.mapDeg.section[sec] != want) xlp_exit_process("Starks: column-major starkinfo whose sections are not whole domains");
            }
        }
    }
    ~StarksT()
    {
        if (!config.generateProof()) return;
        for (int i = 0; i < 5; i++) delete treesGL[i];
        uint64_t* all[] = {d_const_2ns_res, d_const_n, d_const_2ns_own, d_x_n, d_x_2ns, d_zi, d_xdx, d_xdwx, d_challenges, d_publics, d_evals, d_work, d_const_n_full, d_const_2ns_win, d_shard};
        for (uint64_t* p : all) xlp_dev_free(p);
    }
    StarksT(const StarksT&) = delete;
    StarksT& operator=(const StarksT&) = delete;
    uint64_t domainSize() const { return N; }
    // row-sharded mode: this rank holds rows [firstRow(), firstRow() + localRows()) of the witness (pAddress holds exactly them)
    bool isSharded() const { return sharded; }
    uint64_t localRows() const { return sharded ? nLoc : N; }
    uint64_t firstRow() const { return row0; }

    // Starks::merkelizeMemory (starks.cpp:671-700): the debug hash of the polynomial memory -- the words of the slab as they lie, taken as a matrix of
    // 2^k rows (the largest power of two that divides the word count, at least 2; the reference's loop) and hashed with merkletree; the reference prints the
    // root.  Here the slab is THIS process's device slab (a rank's window in a sharded run; the aliased one-GPU layout has no second copy of cm3_2ns behind
    // mapTotalN, so the count is mapTotalN): a parity probe between runs, layouts being equal -- with row-major, unaliased sections the words are the
    // reference's slab and the root is the one the reference prints.  Returns the root (and prints it as the reference does when `print`).
    void merkelizeMemory(Goldilocks::Element rootDBG[4], bool print = true)
    {
        const uint64_t polsSize = slabWords;
        if (polsSize < 2 || (polsSize & 1)) xlp_exit_process("Starks::merkelizeMemory: the slab has an odd number of words");
        uint64_t nrowsDGB = 2;
        while (polsSize % (nrowsDGB * 2) == 0) nrowsDGB *= 2;
        const uint64_t ncolsDGB = polsSize / nrowsDGB;
        const uint64_t numElementsTreeDBG = MerklehashGoldilocks::getTreeNumElements(nrowsDGB);
        uint64_t* treeDBG = dalloc(numElementsTreeDBG);
        XLP_CHECK(xlp_dev_merkle(treeDBG, mem(), ncolsDGB, nrowsDGB));
        uint64_t root[4];
        XLP_CHECK(xlp_copy_d2h(root, treeDBG + numElementsTreeDBG - 4, 4 * sizeof(uint64_t)));       // MerklehashGoldilocks::root: the last four elements
        XLP_CHECK(xlp_dev_free(treeDBG));
        for (int i = 0; i < 4; i++) {
            rootDBG[i] = Goldilocks::fromU64(root[i]);
            if (print) std::cout << "rootDBG[" << i << "]: [ " << root[i] << " ]" << std::endl;
        }
    }
    uint64_t slabSizeWords() const { return slabWords; }
    const uint64_t* slabDevice() { return mem(); }

    void genProof(typename H::Proof& proof, Goldilocks::Element* publicInputs, Goldilocks::Element verkey[4], Steps* steps)
    {
        if (sharded) {
            if constexpr (H::kShardable) genProofSharded(proof, publicInputs, verkey, steps);
            return;
        }
        lastTimers.clear();
        if (steps->colMajor() != colMajor) xlp_exit_process("Starks: the step programs and the starkinfo disagree on the layout of the sections (colMajor)");
        if (XlpDeviceSlab::get().reserve(slabWords) == NULL) xlp_exit_process("Starks: no device slab");
        const eSection secs2ns[4] = {cm1_2ns, cm2_2ns, cm3_2ns, cm4_2ns};
        for (int i = 0; i < 4; i++) treesGL[i]->d_source = section(secs2ns[i]);        // the slab may have been re-reserved by a larger Starks
        treesGL[4]->d_source = constPols2ns();
        const uint64_t W1 = starkInfo.mapSectionsN.section[cm1_n];
        uint64_t numCommited = starkInfo.nCm1;
        typename H::TranscriptT transcript;
        std::vector<Goldilocks::Element> challenges(NUM_CHALLENGES * FIELD_EXTENSION, Goldilocks::zero());
        std::vector<Goldilocks::Element> evals(starkInfo.evMap.size() * FIELD_EXTENSION, Goldilocks::zero());
        typename H::Root root0, root1, root2, root3;
        auto challenge = [&](int k) {
            H::getField(transcript, &challenges[k * FIELD_EXTENSION]);
            XLP_CHECK(xlp_copy_h2d(d_challenges, challenges.data(), challenges.size() * sizeof(uint64_t)));
        };
        // the base-domain sections other than the witness start from zero (prover.cpp:426-430 clears the committed polynomials; in the
        // aliased layout the previous proof's extended sections lie there).  Column-major: AFTER step 1, whose witness bands land there first
        const uint64_t zeroFrom = starkInfo.mapOffsets.section[cm2_n], zeroTo = starkInfo.mapOffsets.section[tmpExp_n] + N * starkInfo.mapSectionsN.section[tmpExp_n];
        auto clearBaseSections = [&](uint64_t skipWords) {      // [zeroFrom + skipWords, zeroTo)
            if (zeroTo > zeroFrom + skipWords && zeroTo <= slabWords) XLP_CHECK(xlp_dev_memset(mem() + zeroFrom + skipWords, 0, (zeroTo - zeroFrom - skipWords) * sizeof(uint64_t)));
        };
        // (column-major, pipelined upload: the first N x W1 words are the staging area of step 1, zeroed there band by band once transposed)
        const bool stageInSections = colMajor && zeroTo > zeroFrom && zeroTo <= slabWords && zeroTo - zeroFrom >= N * starkInfo.mapSectionsN.section[cm1_n];
        clearBaseSections(stageInSections ? N * starkInfo.mapSectionsN.section[cm1_n] : 0);
        XLP_CHECK(xlp_copy_h2d(d_publics, publicInputs, starkInfo.nPublics * sizeof(uint64_t)));
        H::putVerkey(transcript, verkey, constRoot);
        transcript.put(&publicInputs[0], starkInfo.nPublics);
        StepsParams P = params();
        //-------------------------------- 1.- Calculate p_cm1_2ns
        {
            const uint64_t* h_cm1 = (const uint64_t*)pAddress + starkInfo.mapOffsets.section[cm1_n];
            // column-major: the bands of the row-major witness land in the dead base-domain sections behind cm1_n and are transposed from there
            uint64_t* stage = mem() + zeroFrom;
            uint64_t stageWords = zeroTo > zeroFrom && zeroTo <= slabWords ? zeroTo - zeroFrom : 0;
            uint64_t* ownStage = NULL;
            if (colMajor && stageWords < N * W1) { ownStage = dalloc(N * W1); stage = ownStage; stageWords = N * W1; }
            if (!pipelineWitness) {
                Timer t(this, "STARK_STEP_1_WITNESS_UPLOAD");
                XLP_CHECK(xlp_copy_h2d(colMajor ? stage : section(cm1_n), h_cm1, N * W1 * sizeof(uint64_t)));
            }
            Timer t(this, "STARK_STEP_1_LDE_AND_MERKLETREE");
            if constexpr (H::kCommitTimes) {
              if (colMajor) {
                if (pipelineWitness) H::commitFromHostCm(treesGL[0], section(cm1_n), N, section(cm1_2ns), NExtended, h_cm1, stage, stageWords, stageInSections, NExtended, N, W1, &lastUploadMs);
                else {
                    XLP_CHECK(xlp_dev_rows_to_cm(section(cm1_n), N, 0, stage, W1, 0, W1, N, 0));
                    if (stageInSections) XLP_CHECK(xlp_dev_memset(stage, 0, N * W1 * sizeof(uint64_t)));
                    H::commitCm(treesGL[0], section(cm1_2ns), NExtended, section(cm1_n), N, NExtended, N, W1);
                }
                if (ownStage) { XLP_CHECK(xlp_sync()); xlp_dev_free(ownStage); }
              }
            }
            if (colMajor) {}
            else if (pipelineWitness) H::commitFromHost(treesGL[0], section(cm1_n), section(cm1_2ns), h_cm1, NExtended, N, W1, &lastUploadMs);
            else H::commit(treesGL[0], section(cm1_2ns), section(cm1_n), NExtended, N, W1);
            subTimers("STARK_STEP_1_LDE", "STARK_STEP_1_MERKLETREE");
            treesGL[0]->getRoot(root0);
            H::putRoot(transcript, root0);
        }
        //-------------------------------- 2.- Calculate plookups h1 and h2
        challenge(0);   // u
        challenge(1);   // defVal
        {
            Timer t(this, "STARK_STEP_2_CALCULATE_EXPS");
            steps->step2prev_parser_first_avx(P, N, nrowsStepBatch);
        }
        {
            Timer t(this, "STARK_STEP_2_CALCULATEH1H2");
            for (uint64_t i = 0; i < starkInfo.puCtx.size(); i++) {
                const PolRef f = pol(exp2pol(starkInfo.puCtx[i].fExpId)), tt = pol(exp2pol(starkInfo.puCtx[i].tExpId));
                const PolRef h1 = pol(starkInfo.cm_n[numCommited + i * 2]), h2 = pol(starkInfo.cm_n[numCommited + i * 2 + 1]);
                if (f.dim != tt.dim || h1.dim != f.dim || h2.dim != f.dim) xlp_exit_process("Starks: plookup polynomials of different dimensions");
                // Polinomial views straight into the sections (transposeH1H2Columns / ...Rows of the reference are address arithmetic here)
                if (colMajor) {                           // dense elements for the kernel, columns in the sections: four small copies around it
                    uint64_t *df = d_work, *dt = d_work + N * 3, *dh1 = d_work + 2 * N * 3, *dh2 = d_work + 3 * N * 3;
                    toDense(dt, tt, N);
                    if (config.xlpSyntheticLookups) XLP_CHECK(xlp_dev_gather_rows(df, dt, f.dim, N, (uint32_t)f.dim, 0x5EED0500ULL + i));
                    else toDense(df, f, N);
                    XLP_CHECK(xlp_dev_calculate_h1h2(dh1, dh2, df, dt, N, (uint32_t)f.dim));
                    fromDense(h1, dh1, N); fromDense(h2, dh2, N);
                } else if (config.xlpSyntheticLookups) {
                    uint64_t *df = d_work, *dt = d_work + N * 3;
                    XLP_CHECK(xlp_dev_copy_cols(dt, f.dim, tt.p, tt.stride, N, (uint32_t)f.dim));
                    XLP_CHECK(xlp_dev_gather_rows(df, dt, f.dim, N, (uint32_t)f.dim, 0x5EED0500ULL + i));
                    XLP_CHECK(xlp_dev_calculate_h1h2_strided(h1.p, h1.stride, h2.p, h2.stride, df, f.dim, dt, f.dim, N, (uint32_t)f.dim));
                } else {
                    XLP_CHECK(xlp_dev_calculate_h1h2_strided(h1.p, h1.stride, h2.p, h2.stride, f.p, f.stride, tt.p, tt.stride, N, (uint32_t)f.dim));
                }
            }
            numCommited += starkInfo.puCtx.size() * 2;
        }
        {
            Timer t(this, "STARK_STEP_2_LDE_AND_MERKLETREE");
            if constexpr (H::kCommitTimes) { if (colMajor) H::commitCm(treesGL[1], section(cm2_2ns), NExtended, section(cm2_n), N, NExtended, N, starkInfo.mapSectionsN.section[cm2_n]); }
            if (!colMajor) H::commit(treesGL[1], section(cm2_2ns), section(cm2_n), NExtended, N, starkInfo.mapSectionsN.section[cm2_n]);
            subTimers("STARK_STEP_2_LDE", "STARK_STEP_2_MERKLETREE");
            treesGL[1]->getRoot(root1);
            H::putRoot(transcript, root1);
        }
        //-------------------------------- 3.- Compute Z polynomials
        challenge(2);   // gamma
        challenge(3);   // betta
        {
            Timer t(this, "STARK_STEP_3_CALCULATE_EXPS");
            steps->step3prev_parser_first_avx(P, N, nrowsStepBatch);
        }
        {
            Timer t(this, "STARK_STEP_3_CALCULATE_Z");
            std::vector<std::pair<uint64_t, uint64_t>> nd;       // (numId, denId) of puCtx, peCtx, ciCtx in the reference's order
            for (const PuCtx& c : starkInfo.puCtx) nd.emplace_back(c.numId, c.denId);
            for (const PeCtx& c : starkInfo.peCtx) nd.emplace_back(c.numId, c.denId);
            for (const CiCtx& c : starkInfo.ciCtx) nd.emplace_back(c.numId, c.denId);
            for (uint64_t i = 0; i < nd.size(); i++) {
                const PolRef num = pol(exp2pol(nd[i].first)), den = pol(exp2pol(nd[i].second)), z = pol(starkInfo.cm_n[numCommited + i]);
                if (num.dim != 3 || den.dim != 3 || z.dim != 3) xlp_exit_process("Starks: Z polynomials must be ext-3");
                if (colMajor) {
                    uint64_t *dn = d_work, *dd = d_work + N * 3, *dz = d_work + 2 * N * 3;
                    toDense(dn, num, N); toDense(dd, den, N);
                    XLP_CHECK(xlp_dev_calculate_z(dz, dn, dd, N));
                    fromDense(z, dz, N);
                } else
                XLP_CHECK(xlp_dev_calculate_z_strided(z.p, z.stride, num.p, num.stride, den.p, den.stride, N));   // views, no transposeZColumns copies
            }
        }
        {
            Timer t(this, "STARK_STEP_3_CALCULATE_EXPS_2");
            steps->step3_parser_first_avx(P, N, nrowsStepBatch);
        }
        {
            Timer t(this, "STARK_STEP_3_LDE_AND_MERKLETREE");
            if constexpr (H::kCommitTimes) { if (colMajor) H::commitCm(treesGL[2], section(cm3_2ns), NExtended, section(cm3_n), N, NExtended, N, starkInfo.mapSectionsN.section[cm3_n]); }
            if (!colMajor) H::commit(treesGL[2], section(cm3_2ns), section(cm3_n), NExtended, N, starkInfo.mapSectionsN.section[cm3_n]);
            subTimers("STARK_STEP_3_LDE", "STARK_STEP_3_MERKLETREE");
            treesGL[2]->getRoot(root2);
            H::putRoot(transcript, root2);
        }
        //-------------------------------- 4. Compute C Polynomial
        challenge(4);
        if (const2nsInSlab && !d_const_2ns_res) {          // the one-GPU memory trade: the extended constants lie over the now dead cm3_n
            Timer t(this, "STARK_STEP_4_CONST_LDE");
            if (residentConstants && proofsDone >= 1) {
                const size_t need = (size_t)NExtended * starkInfo.nConstants * sizeof(uint64_t);
                size_t fr = 0, tot = 0;
                void* d = NULL;
                if (xlp_dev_mem_info(&fr, &tot) == 0 && getenv("XLP_MEMINFO")) fprintf(stderr, "Starks: step 4 of proof %llu: %.1f GB free of %.1f GB, resident constants need %.1f GB\n", (unsigned long long)proofsDone, fr / 1e9, tot / 1e9, need / 1e9);
                if (fr >= need + (8ULL << 30) && xlp_dev_alloc(&d, need) == 0) {
                    d_const_2ns_res = (uint64_t*)d;
                    treesGL[4]->d_source = d_const_2ns_res;
                    P.const_pols_2ns = d_const_2ns_res;
                    if (logTimers) fprintf(stderr, "Starks: extended constants now resident (%.1f GB; %.1f GB were free)\n", need / 1e9, fr / 1e9);
                }
            }
            if (colMajor) XLP_CHECK(xlp_dev_extend_cm(constPols2ns(), NExtended, 0, d_const_n, N, NExtended, N, starkInfo.nConstants));
            else XLP_CHECK(xlp_dev_extend(constPols2ns(), d_const_n, NExtended, N, starkInfo.nConstants, NULL));
        }
        if (colMajor) { P.coset_rows = N; P.coset_stride = N; }      // the extended-domain programs run coset by coset (XLP_EXPR_COLMAJOR)
        {
            Timer t(this, "STARK_STEP_4_CALCULATE_EXPS_2NS");
            steps->step42ns_parser_first_avx(P, NExtended, nrowsStepBatch);
        }
        {
            Timer t(this, "STARK_STEP_4_CALCULATE_EXPS_2NS_INTT_MUL_NTT");
            uint64_t *qq1 = d_work, *qq2 = d_work + NExtended * starkInfo.qDim;
            XLP_CHECK(xlp_dev_ntt(qq1, section(q_2ns), NExtended, starkInfo.qDim, 1));
            const Goldilocks::Element shiftIn = Goldilocks::exp(Goldilocks::inv(Goldilocks::shift()), N);
            XLP_CHECK(xlp_dev_q_split(qq2, qq1, N, NExtended, (uint32_t)starkInfo.qDeg, (uint32_t)starkInfo.qDim, Goldilocks::toU64(shiftIn)));
            if (colMajor) {                                  // natural rows out of the transform -> coset-major columns
                XLP_CHECK(xlp_dev_ntt(qq2, qq2, NExtended, starkInfo.qDim * starkInfo.qDeg, 0));
                XLP_CHECK(xlp_dev_rows_to_cm(section(cm4_2ns), NExtended, 0, qq2, starkInfo.qDim * starkInfo.qDeg, 0, starkInfo.qDim * starkInfo.qDeg, NExtended, (uint32_t)extendBits));
            } else
            XLP_CHECK(xlp_dev_ntt(section(cm4_2ns), qq2, NExtended, starkInfo.qDim * starkInfo.qDeg, 0));
        }
        {
            Timer t(this, "STARK_STEP_4_MERKLETREE");
            treesGL[3]->merkelize();
            treesGL[3]->getRoot(root3);
            H::putRoot(transcript, root3);
        }
        //-------------------------------- 5. Compute FRI Polynomial
        challenge(7);   // xi
        const Goldilocks::Element* xi = &challenges[7 * FIELD_EXTENSION];
        uint64_t wxi[3];
        {
            Timer t(this, "STARK_STEP_5_LEv_LpEv");
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
            Timer t(this, "STARK_STEP_5_EVMAP");
            std::vector<xlp_evmap_entry> ents(starkInfo.evMap.size());
            for (size_t i = 0; i < ents.size(); i++) {
                const EvMap& ev = starkInfo.evMap[i];
                if (ev.type == EvMap::_const) ents[i] = {constPols2ns() + ev.id, starkInfo.nConstants, 1, ev.prime ? 1u : 0u};
                else {
                    const PolRef p = pol(ev.type == EvMap::cm ? starkInfo.cm_2ns[ev.id] : starkInfo.qs[ev.id]);
                    ents[i] = {p.p, p.stride, (uint32_t)p.dim, ev.prime ? 1u : 0u};
                }
            }
            if (colMajor) {                                  // coset 0 of an extended column IS the column at the rows k << extendBits
                for (size_t i = 0; i < ents.size(); i++) {
                    const EvMap& ev = starkInfo.evMap[i];
                    if (ev.type == EvMap::_const) ents[i] = {constPols2ns() + ev.id * NExtended, NExtended, 1, ev.prime ? 1u : 0u};
                    else {
                        const PolRef p = pol(ev.type == EvMap::cm ? starkInfo.cm_2ns[ev.id] : starkInfo.qs[ev.id]);
                        ents[i] = {p.p, p.cs, (uint32_t)p.dim, ev.prime ? 1u : 0u};
                    }
                }
                XLP_CHECK(xlp_dev_evmap_cm(d_evals, d_work, d_work + N * 3, N, (uint32_t)ents.size(), ents.data()));
            } else
            XLP_CHECK(xlp_dev_evmap(d_evals, d_work, d_work + N * 3, N, (uint32_t)extendBits, (uint32_t)ents.size(), ents.data()));
            XLP_CHECK(xlp_copy_d2h(evals.data(), d_evals, evals.size() * sizeof(uint64_t)));
        }
        transcript.put(evals.data(), starkInfo.evMap.size() * FIELD_EXTENSION);   // (starks.cpp:336-339 puts them three at a time: the same element sequence; one call lets the sponge take its full blocks in one launch)
        challenge(5);   // v1
        challenge(6);   // v2
        {
            Timer t(this, "STARK_STEP_5_XDIVXSUB");
            XLP_CHECK(xlp_dev_x_div_x_sub_xi(d_xdx, d_x_2ns, xlp_raw(xi), NExtended));
            XLP_CHECK(xlp_dev_x_div_x_sub_xi(d_xdwx, d_x_2ns, wxi, NExtended));
        }
        {
            Timer t(this, "STARK_STEP_5_CALCULATE_EXPS");
            steps->step52ns_parser_first_avx(P, NExtended, nrowsStepBatch);
        }
        {
            Timer t(this, "STARK_STEP_FRI");
            Polinomial friPol(NExtended >> (starkInfo.starkStruct.nBitsExt - starkInfo.starkStruct.steps.back().nBits), FIELD_EXTENSION, "friPol");
            H::Prove::prove(proof, treesGL, transcript, friPol, starkInfo.starkStruct.nBitsExt, starkInfo, section(f_2ns));
        }
        proof.proofs.setEvals(evals.data());
        H::storeRoots(proof, root0, root1, root2, root3);
        proofsDone++;
    }
};

typedef StarksT<XlpHashGL> Starks;
