// bctree -c <const_file> -s <starkstruct_file> -t <consttree_file> [-v <verkey_file>]
// Same command line, files and messages as the reference tool (tools/starkpil/bctree/main.cpp:57-107); the
// work runs on the MI355X through libxlp_gl.so (build_const_tree.hpp).
#include "build_const_tree.hpp"

static bool file_exists(const std::string& f) { struct stat s; return stat(f.c_str(), &s) == 0; }

struct Args {
    std::vector<std::string> a;
    Args(int argc, char** argv) { for (int i = 1; i < argc; i++) a.push_back(argv[i]); }
    bool has(const std::string& s, const std::string& l) const
    {
        bool found = false;
        for (const std::string& x : a)
            if (x == s || x == l) {
                if (found) throw std::runtime_error("bctree: cannot use " + s + "/" + l + " parameter twice!");
                found = true;
            }
        return found;
    }
    std::string value(const std::string& s, const std::string& l) const
    {
        for (size_t i = 0; i < a.size(); i++)
            if (a[i] == s || a[i] == l) return i + 1 < a.size() ? a[i + 1] : "";
        return "";
    }
};

int main(int argc, char** argv)
{
    const char* version = "bctree: version 0.1.0.0 (xlp_gl / gfx950)";
    Args args(argc, argv);
    try {
        std::string constFile, starkStructFile, constTreeFile, verKeyFile;
        if (!args.has("-c", "--const")) throw std::runtime_error("bctree: constants input file argument not specified <-c/--const> <const_file>");
        constFile = args.value("-c", "--const");
        if (!file_exists(constFile)) throw std::runtime_error("bctree: constants file doesn't exist (" + constFile + ")");
        if (!args.has("-s", "--stark")) throw std::runtime_error("bctree: starkstruct input file argument not specified <-s/--stark> <starkstruct_file>");
        starkStructFile = args.value("-s", "--stark");
        if (!file_exists(starkStructFile)) throw std::runtime_error("bctree: starkstruct file doesn't exist (" + starkStructFile + ")");
        if (!args.has("-t", "--tree")) throw std::runtime_error("bctree: constants tree ouput file argument not specified <-t/--tree> <consttree_file>");
        constTreeFile = args.value("-t", "--tree");
        if (constTreeFile == "") throw std::runtime_error("bctree: constants tree output file not specified");
        if (args.has("-v", "--verkey")) {
            verKeyFile = args.value("-v", "--verkey");
            if (verKeyFile == "") throw std::runtime_error("bctree: key ouput file not specified");
        }
        std::cout << version << std::endl;
        xlp_bctree::buildConstTree(constFile, starkStructFile, constTreeFile, verKeyFile);
        return EXIT_SUCCESS;
    } catch (const std::exception& e) {
        std::cerr << e.what() << std::endl;
        std::cerr << version << std::endl;
        std::cerr << "usage: bctree <-c|--const> <const_file> <-s|--stark> <starkstruct_file> <-t|--tree> <consttree_file> [<-v|--verkey> <verkey_file>]" << std::endl;
        std::cerr << "example: bctree -c zkevm.const -s zkevm.starkstruct.json -t zkevm.consttree -v zkevm.verkey" << std::endl;
        return EXIT_FAILURE;
    }
}
