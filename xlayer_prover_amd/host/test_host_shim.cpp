// Drives the host-side mirror (xlp_host.hpp) through the commit + FRI part of
// Starks::genProof (starks.cpp:28-60,134-141,215-222,285-293,394-402) on a small
// synthetic STARK and prints the proof in zkin layout.  tests/test_host_shim.py
// rebuilds the same proof with the CPU oracle and compares every field.
//   usage: test_host_shim <nBits> <nBitsExt> <nQueries> <step nBits...>
#include <cstdio>
#include <vector>
#include "starks_mini.hpp"

int main(int argc, char** argv)
{
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
