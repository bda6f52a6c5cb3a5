// GPU `bctree`: constant-polynomial tree builder.
//
// Mirrors buildConstTree(constFile, starkStructFile, constTreeFile, verKeyFile) of the reference
// (tools/starkpil/bctree/build_const_tree.cpp:533-603, GL branch): reads the row-major `.const` dump
// (n x nPols u64), extends it to nExt rows on the coset (interpolate :398-531 == extendPol), builds the
// Poseidon Merkle tree and writes
//     .consttree = [ nPols, nExt, extended source (nExt*nPols), nodes ((2*nExt-1)*4) ]   (u64 little endian)
//     .verkey    = {"constRoot":[r0,r1,r2,r3]}                                          (JSON numbers)
// which is the layout MerkleTreeGL(Goldilocks::Element *tree) parses (merkleTreeGL.hpp:24-32).
// The extension and the tree run on the GPU with the data resident in HBM; the host only does file IO.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <sys/stat.h>
#include <vector>
#include "../../include/xlp_gl.h"

namespace xlp_bctree {

// Top-level members of a JSON object, values kept as raw text (numbers / strings without quotes / nested text).
// Enough for the starkstruct file ({"nBits":..,"nBitsExt":..,"verificationHashType":"GL","steps":[...]}).
inline bool json_top_level(const std::string& s, const std::string& key, std::string& out)
{
    size_t i = 0;
    const size_t n = s.size();
    auto skip_ws = [&]() { while (i < n && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r')) i++; };
    auto read_string = [&]() -> std::string {
        std::string r;
        i++;  // opening quote
        while (i < n && s[i] != '"') { if (s[i] == '\\' && i + 1 < n) i++; r += s[i++]; }
        i++;
        return r;
    };
    skip_ws();
    if (i >= n || s[i] != '{') throw std::runtime_error("starkstruct: not a JSON object");
    i++;
    for (;;) {
        skip_ws();
        if (i >= n) throw std::runtime_error("starkstruct: truncated JSON");
        if (s[i] == '}') return false;
        if (s[i] == ',') { i++; continue; }
        if (s[i] != '"') throw std::runtime_error("starkstruct: expected a member name");
        const std::string k = read_string();
        skip_ws();
        if (i >= n || s[i] != ':') throw std::runtime_error("starkstruct: expected ':'");
        i++;
        skip_ws();
        std::string v;
        if (i < n && s[i] == '"') v = read_string();
        else {
            int depth = 0;
            bool in_str = false;
            while (i < n) {
                const char c = s[i];
                if (in_str) { if (c == '\\') { v += c; i++; } else if (c == '"') in_str = false; }
                else if (c == '"') in_str = true;
                else if (c == '[' || c == '{') depth++;
                else if (c == ']' || c == '}') { if (depth == 0) break; depth--; }
                else if (c == ',' && depth == 0) break;
                v += s[i++];
            }
            while (!v.empty() && (v.back() == ' ' || v.back() == '\n' || v.back() == '\t' || v.back() == '\r')) v.pop_back();
        }
        if (k == key) { out = v; return true; }
    }
}

inline uint64_t json_u64(const std::string& s, const std::string& key)
{
    std::string v;
    if (!json_top_level(s, key, v)) throw std::runtime_error("starkstruct: missing \"" + key + "\"");
    char* end = NULL;
    const unsigned long long r = strtoull(v.c_str(), &end, 10);
    if (end == v.c_str() || *end) throw std::runtime_error("starkstruct: \"" + key + "\" is not an unsigned integer");
    return r;
}

#define XLP_BCT(call) do { if ((call) != 0) throw std::runtime_error(std::string("bctree: ") + xlp_last_error()); } while (0)

inline void buildConstTree(const std::string constFile, const std::string starkStructFile, const std::string constTreeFile, const std::string verKeyFile)
{
    std::ifstream fs(starkStructFile);
    if (!fs) throw std::runtime_error("bctree: cannot open " + starkStructFile);
    std::stringstream ss; ss << fs.rdbuf();
    const std::string st = ss.str();
    const uint64_t nBits = json_u64(st, "nBits"), nBitsExt = json_u64(st, "nBitsExt");
    std::string hashType;
    if (!json_top_level(st, "verificationHashType", hashType)) throw std::runtime_error("starkstruct: missing \"verificationHashType\"");
    const bool bn128 = hashType == "BN128";
    if (hashType != "GL" && !bn128) throw std::runtime_error("Invalid Hash Type: " + hashType);
    if (nBitsExt < nBits || nBitsExt > 32) throw std::runtime_error("bctree: need nBits <= nBitsExt <= 32");
    const uint64_t n = 1ULL << nBits, nExt = 1ULL << nBitsExt;

    struct stat sb;
    if (stat(constFile.c_str(), &sb)) throw std::runtime_error("bctree: cannot stat " + constFile);
    const uint64_t constFileSize = (uint64_t)sb.st_size;
    const uint64_t nPols = constFileSize / (n * 8);
    if (nPols == 0 || nPols * n * 8 != constFileSize) throw std::runtime_error("bctree: const file size is not nPols * 2^nBits * 8");
    std::cout << "Pols=" << nPols << "\nnBits=" << nBits << "\nnBitsExt=" << nBitsExt << std::endl;

    // BN128 (build_const_tree.cpp:609-641, the recursivef constants): [source_width, height, source, nodes] with MerkleTreeBN128's 16-ary node
    // buffer (getNumNodes(height) Fr elements of 4 words) instead of the binary Goldilocks tree
    const uint64_t numElementsTree = bn128 ? xlp_merkle_bn128_num_nodes(nExt) * 4 : (2 * nExt - 1) * 4, header = 2;
    const uint64_t numElementsCopy = header + nPols * nExt, numElements = numElementsCopy + numElementsTree;
    std::vector<uint64_t> constTree(numElements);
    {
        FILE* f = fopen(constFile.c_str(), "rb");
        if (!f) throw std::runtime_error("bctree: cannot open " + constFile);
        // staged through the tail of the output buffer (free until the nodes come back)
        uint64_t* stage = constTree.data() + numElements - n * nPols;
        if (fread(stage, 8, n * nPols, f) != n * nPols) { fclose(f); throw std::runtime_error("bctree: short read on " + constFile); }
        fclose(f);
        XLP_BCT(xlp_init(0));
        uint64_t *d_in = NULL, *d_ext = NULL, *d_nodes = NULL;
        XLP_BCT(xlp_dev_alloc((void**)&d_in, n * nPols * 8));
        XLP_BCT(xlp_dev_alloc((void**)&d_ext, nExt * nPols * 8));
        XLP_BCT(xlp_dev_alloc((void**)&d_nodes, numElementsTree * 8));
        XLP_BCT(xlp_copy_h2d(d_in, stage, n * nPols * 8));
        XLP_BCT(xlp_dev_extend(d_ext, d_in, nExt, n, nPols, NULL));              // Interpolate
        if (bn128) XLP_BCT(xlp_dev_merkle_bn128(d_nodes, d_ext, nPols, nExt));  // MerkleTree_BN128
        else XLP_BCT(xlp_dev_merkle(d_nodes, d_ext, nPols, nExt));              // MerkleTree_GL
        XLP_BCT(xlp_sync());
        constTree[0] = nPols;
        constTree[1] = nExt;
        XLP_BCT(xlp_copy_d2h(&constTree[header], d_ext, nExt * nPols * 8));
        XLP_BCT(xlp_copy_d2h(&constTree[numElementsCopy], d_nodes, numElementsTree * 8));
        xlp_dev_free(d_in); xlp_dev_free(d_ext); xlp_dev_free(d_nodes);
    }
    if (verKeyFile != "") {
        std::ofstream fv(verKeyFile);
        if (!fv) throw std::runtime_error("bctree: cannot write " + verKeyFile);
        if (bn128) {
            // rawfr.toString(constRoot): the root out of Montgomery form, in decimal (one JSON string)
            typedef unsigned __int128 u128;
            static const uint64_t Q[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
            const uint64_t NP = 0xc2e1f593efffffffULL;
            uint64_t t[6] = {constTree[numElements - 4], constTree[numElements - 3], constTree[numElements - 2], constTree[numElements - 1], 0, 0};
            for (int i = 0; i < 4; i++) {                  // Montgomery reduction by 2^256: t <- (t + m q) / 2^64, four times
                const uint64_t m = t[0] * NP;
                u128 c = ((u128)m * Q[0] + t[0]) >> 64;
                for (int j = 1; j < 4; j++) { const u128 p_ = (u128)m * Q[j] + t[j] + (uint64_t)c; t[j - 1] = (uint64_t)p_; c = p_ >> 64; }
                const u128 s_ = (u128)t[4] + (uint64_t)c; t[3] = (uint64_t)s_; t[4] = (uint64_t)(s_ >> 64);
            }
            bool ge = t[4] != 0;
            if (!ge) { ge = true; for (int i = 3; i >= 0; i--) if (t[i] != Q[i]) { ge = t[i] > Q[i]; break; } }
            if (ge) { u128 bw = 0; for (int i = 0; i < 4; i++) { const u128 d = (u128)t[i] - Q[i] - (uint64_t)bw; t[i] = (uint64_t)d; bw = (d >> 64) & 1; } }
            std::string dec;
            uint64_t w[4] = {t[0], t[1], t[2], t[3]};
            while (w[0] | w[1] | w[2] | w[3]) {
                u128 rem = 0;
                for (int i = 3; i >= 0; i--) { const u128 cur = (rem << 64) | w[i]; w[i] = (uint64_t)(cur / 10); rem = cur % 10; }
                dec.insert(dec.begin(), (char)('0' + (int)rem));
            }
            fv << "{\"constRoot\":\"" << (dec.empty() ? "0" : dec) << "\"}" << std::endl;
        } else
        fv << "{\"constRoot\":[" << constTree[numElements - 4] << "," << constTree[numElements - 3] << "," << constTree[numElements - 2] << ","
           << constTree[numElements - 1] << "]}" << std::endl;
    }
    std::ofstream fw(constTreeFile.c_str(), std::fstream::out | std::fstream::binary);
    if (!fw) throw std::runtime_error("bctree: cannot write " + constTreeFile);
    fw.write((const char*)constTree.data(), (std::streamsize)(numElements * 8));
    fw.close();
    std::cout << "Files Generated Correctly" << std::endl;
}

}  // namespace xlp_bctree
