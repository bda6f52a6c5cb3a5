// Constraint steps of a STARK: `Steps` is the interface Starks::genProof calls (src/starkpil/steps.hpp:21-70) and
// `XlpSteps` the GPU implementation that stands where the reference's generated ZkevmSteps / C12aSteps / Recursive1Steps /
// Recursive2Steps classes stand (src/starkpil/zkevm/chelpers/zkevmSteps.hpp:10-57): the five programs are DATA here --
// converted from the reference's generated tables by tools/chelpers_to_xlp.py, laid out for a slab by
// tools/make_stark_bundle.py -- evaluated by the expression VM of libxlp_gl.so or by its ahead-of-time plugins.
//
// StepsParams (steps.hpp:4-19) becomes its device-pointer form xlp_steps_params (include/xlp_gl.h): the sections stay in HBM.
#pragma once
#include <map>
#include "xlp_host.hpp"

typedef xlp_steps_params StepsParams;

// a program's stores to NEXT rows (zkEVM step3prev / step3 have a few): in a row-sharded run the rows written past a rank's last row belong
// to the next rank (host/starks.hpp haloWriteback).  section: eSection; columns [col, col + cols); next: the row distance
struct NextRowStore { int section; uint64_t col, cols, next; };

class Steps
{
public:
    virtual ~Steps() {}
    virtual const std::vector<NextRowStore>& nextRowStores(int /*0 step2prev .. 4 step52ns*/) const { static const std::vector<NextRowStore> none; return none; }
    virtual bool colMajor() const { return false; }          // the programs address column-major sections (must match the starkinfo's layout)
    // nrowsBatch is the reference's SIMD batch (4 / 8 rows, starks.cpp:70-80); one row per lane here, so it is ignored
    virtual void step2prev_parser_first_avx(StepsParams& params, uint64_t nrows, uint64_t nrowsBatch) = 0;
    virtual void step3prev_parser_first_avx(StepsParams& params, uint64_t nrows, uint64_t nrowsBatch) = 0;
    virtual void step3_parser_first_avx(StepsParams& params, uint64_t nrows, uint64_t nrowsBatch) = 0;
    virtual void step42ns_parser_first_avx(StepsParams& params, uint64_t nrows, uint64_t nrowsBatch) = 0;
    virtual void step52ns_parser_first_avx(StepsParams& params, uint64_t nrows, uint64_t nrowsBatch) = 0;
};

class XlpSteps : public Steps
{
    // Per step, in the order of preference the bundle's manifest (<name>.steps.json) records:
    //   * "dots": the FRI polynomial as one-row table program + xlp_dev_dot_columns over every committed column + a short row program;
    //   * "tab":  the program in dot-product form -- a row program reading powers of the challenges and evaluation constants from an
    //             extended challenge array, which a one-row table program fills once per call (xlayer_prover_amd/expr_opt.py);
    //   * the program as converted.
    // "plugin" / "tab_plugin": ahead-of-time kernels of the row / table program (xlp_expr_attach checks them against the program).
    struct DotSpec { uint64_t kind, offset, stride, dim, group, coef; };
    struct Step {
        xlp_expr_program* prog = NULL;
        xlp_expr_program* tab = NULL;
        uint64_t* ext = NULL;                // device, 3 * n_ext words: challenges 0..7, then the table
        uint64_t n_ext = 0;
        std::vector<DotSpec> dots;
        std::vector<NextRowStore> stores;    // manifest "nextRowStores" (bundles of a row-sharded proof)
        uint64_t ngroups = 0;
        uint64_t* dotOut = NULL;             // device, nrows x 3 x ngroups (allocated at the first run)
        uint64_t dotOutWords = 0;
        std::string mode = "absent";
    };
    Step step_[5];
    bool colMajor_ = false;                  // manifest "colMajor": the row programs address column-major sections (XLP_EXPR_COLMAJOR)
    static const uint64_t NCH = 8;           // challenges of a proof (starks.cpp)
    static bool exists(const std::string& path) { FILE* f = fopen(path.c_str(), "rb"); if (f) fclose(f); return f != NULL; }
    static std::vector<uint64_t> readWords(const std::string& path)
    {
        FILE* f = fopen(path.c_str(), "rb");
        if (!f) xlp_exit_process(("XlpSteps: cannot open " + path).c_str());
        fseek(f, 0, SEEK_END); const long sz = ftell(f); fseek(f, 0, SEEK_SET);
        std::vector<uint64_t> v(sz / 8);
        if (sz && fread(v.data(), 8, v.size(), f) != v.size()) { fclose(f); xlp_exit_process(("XlpSteps: short read on " + path).c_str()); }
        fclose(f);
        return v;
    }
    static xlp_expr_program* load(const std::string& path, uint64_t magic, uint64_t* extra, bool colMajor = false)
    {
        const std::vector<uint64_t> w = readWords(path);
        const size_t nh = extra ? 5 : 4;
        if (w.size() < nh || w[0] != magic || w.size() != nh + w[3]) xlp_exit_process(("XlpSteps: bad header in " + path).c_str());
        if (extra) *extra = w[4];
        xlp_expr_program* p = NULL;
        // (column-major sections of host/starks.hpp hold canonical words by construction: the promise XLP_EXPR_CANONICAL asks for)
        XLP_CHECK(xlp_expr_create_ex(&p, w.data() + nh, w[3], w[1], w[2], colMajor ? (XLP_EXPR_COLMAJOR | XLP_EXPR_CANONICAL) : 0));
        return p;
    }
    static void attach(xlp_expr_program* p, const std::string& so, std::string& mode)
    {
        if (so.empty() || !exists(so)) return;
        if (xlp_expr_attach(p, so.c_str()) != 0) { fprintf(stderr, "XlpSteps: plugin %s not attached: %s\n", so.c_str(), xlp_last_error()); return; }
        mode += ", compiled";
    }
    void run(int k, const StepsParams& params, uint64_t nrows)
    {
        Step& s = step_[k];
        if (!s.prog) return;                                     // a STARK without this step (no plookups: step2prev)
        if (!s.n_ext) { XLP_CHECK(xlp_dev_expr_run(s.prog, &params, 0, nrows)); return; }
        if (params.challenges_size < 3 * NCH) xlp_exit_process("XlpSteps: fewer than 8 challenges");
        XLP_CHECK(xlp_dev_copy_cols(s.ext, 3, params.challenges, 3, NCH, 3));
        if (s.tab) {
            StepsParams t;
            memset(&t, 0, sizeof t);
            t.pols = s.ext; t.pols_size = 3 * s.n_ext;
            t.challenges = params.challenges; t.challenges_size = params.challenges_size;
            t.evals = params.evals; t.evals_size = params.evals_size;
            XLP_CHECK(xlp_dev_expr_run(s.tab, &t, 0, 1));
        }
        StepsParams r = params;
        r.challenges = s.ext; r.challenges_size = 3 * s.n_ext;
        if (!s.dots.empty() && colMajor_) {
            // the sums as 3 * ngroups COLUMNS of nrows words (what the converted row program reads); entries: the columns' first words, constant
            // polynomials at column * rows of their buffer
            // (storage rows: a rank's window keeps halo rows behind every coset block -- coset_stride > coset_rows -- and the sums are laid out like the sections)
            const uint64_t srows = params.coset_rows && params.coset_stride > params.coset_rows ? nrows / params.coset_rows * params.coset_stride : nrows;
            const uint64_t words = srows * 3 * s.ngroups;
            if (s.dotOutWords < words) {
                if (s.dotOut) XLP_CHECK(xlp_dev_free(s.dotOut));
                void* d = NULL;
                XLP_CHECK(xlp_dev_alloc(&d, words * sizeof(uint64_t)));
                s.dotOut = (uint64_t*)d; s.dotOutWords = words;
            }
            const uint64_t crows = params.n_const_pols_2ns ? params.const_pols_2ns_size / params.n_const_pols_2ns : 0;
            std::vector<xlp_dot_entry> ents(s.dots.size());
            for (size_t i = 0; i < ents.size(); i++) {
                const DotSpec& d = s.dots[i];
                if (d.kind == 0) ents[i] = {params.pols + d.offset, d.stride, (uint32_t)d.dim, (uint32_t)d.group, d.coef};
                else ents[i] = {params.const_pols_2ns + d.offset * crows, crows, 1u, (uint32_t)d.group, d.coef};
            }
            XLP_CHECK(xlp_dev_dot_columns_cm(s.dotOut, srows, ents.data(), ents.size(), (uint32_t)s.ngroups, s.ext, 3 * s.n_ext, srows));
            r.pols = s.dotOut; r.pols_size = words;
        } else if (!s.dots.empty()) {
            const uint64_t words = nrows * 3 * s.ngroups;
            if (s.dotOutWords < words) {
                if (s.dotOut) XLP_CHECK(xlp_dev_free(s.dotOut));
                void* d = NULL;
                XLP_CHECK(xlp_dev_alloc(&d, words * sizeof(uint64_t)));
                s.dotOut = (uint64_t*)d; s.dotOutWords = words;
            }
            std::vector<xlp_dot_entry> ents(s.dots.size());
            for (size_t i = 0; i < ents.size(); i++) {
                const DotSpec& d = s.dots[i];
                if (d.kind == 0) ents[i] = {params.pols + d.offset, d.stride, (uint32_t)d.dim, (uint32_t)d.group, d.coef};
                else ents[i] = {params.const_pols_2ns + d.offset, params.n_const_pols_2ns, 1u, (uint32_t)d.group, d.coef};
            }
            XLP_CHECK(xlp_dev_dot_columns(s.dotOut, ents.data(), ents.size(), (uint32_t)s.ngroups, s.ext, 3 * s.n_ext, nrows));
            r.pols = s.dotOut; r.pols_size = words;
        }
        XLP_CHECK(xlp_dev_expr_run(s.prog, &r, 0, nrows));
    }
    // Only step2prev may be missing: a STARK without plookups has no such program (c12a, recursive1/2/f ship none; the reference's
    // step2prev_first is then an empty function).  Any other step missing means a mis-deployed bundle: a proof made without it would be
    // complete and invalid, so stop like the reference's loaders do (exitProcess).
    static void requirePresent(int k, const std::string& what)
    {
        if (k != 0) xlp_exit_process(("XlpSteps: " + what).c_str());
    }
    void allocExt(Step& s)
    {
        void* d = NULL;
        XLP_CHECK(xlp_dev_alloc(&d, 3 * s.n_ext * sizeof(uint64_t)));
        s.ext = (uint64_t*)d;
    }

public:
    // dir/<name>.steps.json (tools/make_stark_bundle.py), or -- without a manifest -- dir/<name>.<step>.xlpexpr with optional
    // .opt.xlpexpr / .tab.xlpexpr pairs (tools/optimize_programs.py)
    XlpSteps(const std::string& dir, const std::string& name, bool use_optimised = true)
    {
        static const char* steps[5] = {"step2prev", "step3prev", "step3", "step42ns", "step52ns"};
        const std::string manifest = dir + "/" + name + ".steps.json";
        if (exists(manifest)) {
            std::string text;
            FILE* f = fopen(manifest.c_str(), "rb"); char buf[65536]; size_t r;
            while ((r = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, r);
            fclose(f);
            const xlp_json::Value j = xlp_json::Parser(text).value();
            const xlp_json::Value& S = j["steps"];
            colMajor_ = j.contains("colMajor") && j["colMajor"].kind == xlp_json::Value::Bool && (bool)j["colMajor"];
            auto str = [&](const xlp_json::Value& e, const char* k) { return e.contains(k) && !e[k].is_null() ? dir + "/" + e[k].str() : std::string(); };
            for (int k = 0; k < 5; k++) {
                if (!S.contains(steps[k])) { requirePresent(k, manifest + " has no " + steps[k]); continue; }
                const xlp_json::Value& e = S[steps[k]];
                Step& s = step_[k];
                s.prog = load(str(e, "prog"), 0x3152505845504c58ULL /* "XLPEXPR1" */, NULL, colMajor_);
                s.mode = "as converted";
                if (e.contains("n_ext")) { s.n_ext = e["n_ext"]; s.mode = "dot-product form"; }
                if (e.contains("tab")) { uint64_t n = 0; s.tab = load(str(e, "tab"), 0x3130424154504c58ULL /* "XLPTAB01" */, &n); if (n != s.n_ext) xlp_exit_process("XlpSteps: table size mismatch"); }
                if (e.contains("dots")) {
                    const std::vector<uint64_t> w = readWords(str(e, "dots"));
                    if (w.size() < 3 || w[0] != 0x3153544f44504c58ULL /* "XLPDOTS1" */ || w.size() != 3 + 6 * w[2]) xlp_exit_process("XlpSteps: bad dots file");
                    s.ngroups = w[1];
                    for (uint64_t i = 0; i < w[2]; i++) s.dots.push_back({w[3 + 6 * i], w[4 + 6 * i], w[5 + 6 * i], w[6 + 6 * i], w[7 + 6 * i], w[8 + 6 * i]});
                    s.mode = "table program + dot_columns (" + std::to_string(s.dots.size()) + " entries) + row program";
                }
                if (e.contains("nextRowStores"))
                    for (const xlp_json::Value& v : e["nextRowStores"].a) {
                        NextRowStore st;
                        st.section = (int)string2section(v[(size_t)0].str()); st.col = v[(size_t)1]; st.cols = v[(size_t)2]; st.next = v[(size_t)3];
                        s.stores.push_back(st);
                    }
                if (s.n_ext) allocExt(s);
                attach(s.prog, str(e, "plugin"), s.mode);
                if (s.tab) { std::string m; attach(s.tab, str(e, "tab_plugin"), m); }
            }
            return;
        }
        for (int k = 0; k < 5; k++) {
            const std::string base = dir + "/" + name + "." + steps[k];
            Step& s = step_[k];
            if (use_optimised && exists(base + ".opt.xlpexpr") && exists(base + ".tab.xlpexpr")) {
                s.prog = load(base + ".opt.xlpexpr", 0x3152505845504c58ULL, NULL);
                s.tab = load(base + ".tab.xlpexpr", 0x3130424154504c58ULL, &s.n_ext);
                allocExt(s);
                s.mode = "dot-product form";
            } else if (exists(base + ".xlpexpr")) {
                s.prog = load(base + ".xlpexpr", 0x3152505845504c58ULL, NULL);
                s.mode = "as converted";
            } else requirePresent(k, "cannot open " + base + ".xlpexpr");
        }
    }
    ~XlpSteps()
    {
        for (int k = 0; k < 5; k++) {
            if (step_[k].prog) xlp_expr_destroy(step_[k].prog);
            if (step_[k].tab) xlp_expr_destroy(step_[k].tab);
            if (step_[k].ext) xlp_dev_free(step_[k].ext);
            if (step_[k].dotOut) xlp_dev_free(step_[k].dotOut);
        }
    }
    const std::vector<NextRowStore>& nextRowStores(int k) const override { return step_[k].stores; }
    bool optimised(int k) const { return step_[k].n_ext != 0; }
    bool colMajor() const override { return colMajor_; }
    const std::string& mode(int k) const { return step_[k].mode; }
    XlpSteps(const XlpSteps&) = delete;
    XlpSteps& operator=(const XlpSteps&) = delete;
    void step2prev_parser_first_avx(StepsParams& params, uint64_t nrows, uint64_t) { run(0, params, nrows); }
    void step3prev_parser_first_avx(StepsParams& params, uint64_t nrows, uint64_t) { run(1, params, nrows); }
    void step3_parser_first_avx(StepsParams& params, uint64_t nrows, uint64_t) { run(2, params, nrows); }
    void step42ns_parser_first_avx(StepsParams& params, uint64_t nrows, uint64_t) { run(3, params, nrows); }
    void step52ns_parser_first_avx(StepsParams& params, uint64_t nrows, uint64_t) { run(4, params, nrows); }
};

// the names Prover::genBatchProof / genAggregatedProof instantiate (prover.cpp:576,606,641,779)
struct ZkevmSteps : XlpSteps { ZkevmSteps(const std::string& dir) : XlpSteps(dir, "zkevm") {} };
struct C12aSteps : XlpSteps { C12aSteps(const std::string& dir) : XlpSteps(dir, "c12a") {} };
struct Recursive1Steps : XlpSteps { Recursive1Steps(const std::string& dir) : XlpSteps(dir, "recursive1") {} };
struct Recursive2Steps : XlpSteps { Recursive2Steps(const std::string& dir) : XlpSteps(dir, "recursive2") {} };
struct StarkRecursiveFSteps : XlpSteps { StarkRecursiveFSteps(const std::string& dir) : XlpSteps(dir, "recursivef") {} };   // starkRecursiveFSteps.hpp
