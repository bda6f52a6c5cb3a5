// The part of `class Config` (src/config/config.hpp:11-208, config.cpp:62-330) that the batch-proof path reads, loaded from the
// same JSON keys as the reference's config file (testvectors/config_runFile_BatchProof.json), each overridable by the same
// environment variable (ParseBool / ParseString of config.cpp:62-118).  Keys this path does not use are ignored.
// Added keys (GPU build only, all optional):
//   "xlpChelpersPath"        directory of the converted step programs / bundles (default: next to the library: xlayer_prover_amd/chelpers)
//   "xlpSyntheticLookups"    run plookups on a synthetic witness: f is a selection of rows of t (xlp_dev_gather_rows)
//   "xlpDevice"              HIP device index (default 0)
//   "xlpResidentConstants"   keep a resident copy of the extended constant polynomials from the second proof on when the device has
//                            room beside everything else (default true; the reference keeps constPols2ns loaded, starks.hpp:138-146)
#pragma once
#include <cstdlib>
#include <string>
#include "xlp_host.hpp"

class Config
{
public:
    bool runFileGenBatchProof = false, runFileGenAggregatedProof = false, runFileGenFinalProof = false, runAggregatorClient = false;
    bool saveOutputToFile = false, saveProofToFile = false, mapConstPolsFile = false, mapConstantsTreeFile = false;
    std::string inputFile, outputPath = "output", configPath = "config", publicsOutput = "public.json", proofFile = "proof.json";
    std::string zkevmCmPols, zkevmCmPolsAfterExecutor;
    std::string zkevmConstPols, zkevmConstantsTree, zkevmStarkInfo, zkevmVerkey;
    std::string c12aConstPols, c12aConstantsTree, c12aStarkInfo, c12aVerkey;
    std::string recursive1ConstPols, recursive1ConstantsTree, recursive1StarkInfo, recursive1Verkey;
    std::string recursive2ConstPols, recursive2ConstantsTree, recursive2StarkInfo, recursive2Verkey;
    std::string recursivefConstPols, recursivefConstantsTree, recursivefStarkInfo, recursivefVerkey;
    std::string finalStarkZkey;                      // config.cpp:268 (default <configPath>/final/final.fflonk.zkey); here: empty = no fflonk prover is set up
    std::string xlpChelpersPath;
    bool xlpSyntheticLookups = false, xlpResidentConstants = true;
    uint64_t xlpDevice = 0;

    bool generateProof(void) const { return runFileGenBatchProof || runFileGenAggregatedProof || runFileGenFinalProof || runAggregatorClient; }

    void load(const xlp_json::Value& config)
    {
        ParseBool(config, "runFileGenBatchProof", "RUN_FILE_GEN_BATCH_PROOF", runFileGenBatchProof, false);
        ParseBool(config, "runFileGenAggregatedProof", "RUN_FILE_GEN_AGGREGATED_PROOF", runFileGenAggregatedProof, false);
        ParseBool(config, "runFileGenFinalProof", "RUN_FILE_GEN_FINAL_PROOF", runFileGenFinalProof, false);
        ParseBool(config, "runAggregatorClient", "RUN_AGGREGATOR_CLIENT", runAggregatorClient, false);
        ParseBool(config, "saveOutputToFile", "SAVE_OUTPUT_TO_FILE", saveOutputToFile, false);
        ParseBool(config, "saveProofToFile", "SAVE_PROOF_TO_FILE", saveProofToFile, false);
        ParseBool(config, "mapConstPolsFile", "MAP_CONST_POLS_FILE", mapConstPolsFile, false);
        ParseBool(config, "mapConstantsTreeFile", "MAP_CONSTANTS_TREE_FILE", mapConstantsTreeFile, false);
        ParseString(config, "inputFile", "INPUT_FILE", inputFile, "");
        ParseString(config, "outputPath", "OUTPUT_PATH", outputPath, "output");
        ParseString(config, "configPath", "CONFIG_PATH", configPath, "config");
        ParseString(config, "publicsOutput", "PUBLICS_OUTPUT", publicsOutput, "public.json");
        ParseString(config, "proofFile", "PROOF_FILE", proofFile, "proof.json");
        ParseString(config, "zkevmCmPols", "ZKEVM_CM_POLS", zkevmCmPols, "");
        ParseString(config, "zkevmCmPolsAfterExecutor", "ZKEVM_CM_POLS_AFTER_EXECUTOR", zkevmCmPolsAfterExecutor, "");
        struct { const char* stark; std::string *c, *t, *s, *v; } S[5] = {{"zkevm", &zkevmConstPols, &zkevmConstantsTree, &zkevmStarkInfo, &zkevmVerkey},
                                                                          {"c12a", &c12aConstPols, &c12aConstantsTree, &c12aStarkInfo, &c12aVerkey},
                                                                          {"recursive1", &recursive1ConstPols, &recursive1ConstantsTree, &recursive1StarkInfo, &recursive1Verkey},
                                                                          {"recursive2", &recursive2ConstPols, &recursive2ConstantsTree, &recursive2StarkInfo, &recursive2Verkey},
                                                                          {"recursivef", &recursivefConstPols, &recursivefConstantsTree, &recursivefStarkInfo, &recursivefVerkey}};
        for (auto& s : S) {                                        // config.cpp:238-268: <configPath>/<stark>/<stark>.{const,consttree,starkinfo.json,verkey.json}
            const std::string n = s.stark, up = upper(n), dir = configPath + "/" + n + "/" + n;
            ParseString(config, (n + "ConstPols").c_str(), (up + "_CONST_POLS").c_str(), *s.c, dir + ".const");
            ParseString(config, (n + "ConstantsTree").c_str(), (up + "_CONSTANTS_TREE").c_str(), *s.t, dir + ".consttree");
            ParseString(config, (n + "StarkInfo").c_str(), (up + "_STARK_INFO").c_str(), *s.s, dir + ".starkinfo.json");
            ParseString(config, (n + "Verkey").c_str(), (up + "_VERKEY").c_str(), *s.v, dir + ".verkey.json");
        }
        ParseString(config, "finalStarkZkey", "FINAL_STARK_ZKEY", finalStarkZkey, "");
        ParseString(config, "xlpChelpersPath", "XLP_CHELPERS_PATH", xlpChelpersPath, "");
        ParseBool(config, "xlpSyntheticLookups", "XLP_SYNTHETIC_LOOKUPS", xlpSyntheticLookups, false);
        ParseBool(config, "xlpResidentConstants", "XLP_RESIDENT_CONSTANTS", xlpResidentConstants, true);
        if (config.contains("xlpDevice")) xlpDevice = config["xlpDevice"];
    }
    void load(const std::string& file) { load(xlp_json::Parser(xlp_read_file(file)).value()); }

    static std::string xlp_read_file(const std::string& file)
    {
        FILE* f = fopen(file.c_str(), "rb");
        if (!f) xlp_exit_process(("cannot open " + file).c_str());
        std::string text; char buf[65536]; size_t r;
        while ((r = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, r);
        fclose(f);
        return text;
    }

private:
    static std::string upper(std::string s) { for (char& c : s) c = (char)toupper((unsigned char)c); return s; }
    static void ParseBool(const xlp_json::Value& config, const char* key, const char* env, bool& v, bool dflt)
    {
        v = dflt;
        if (config.kind == xlp_json::Value::Object && config.contains(key) && config[key].kind == xlp_json::Value::Bool) v = (bool)config[key];
        if (const char* e = getenv(env)) { const std::string s = e; v = s == "true" || s == "TRUE" || s == "1"; }
    }
    static void ParseString(const xlp_json::Value& config, const char* key, const char* env, std::string& v, const std::string& dflt)
    {
        v = dflt;
        if (config.kind == xlp_json::Value::Object && config.contains(key) && config[key].kind == xlp_json::Value::String) v = config[key].str();
        if (const char* e = getenv(env)) v = e;
    }
};
