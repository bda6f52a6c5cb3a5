// Runs a generated expression plugin on the CPU (tests only): stages the columns the way csrc/expr.hip does
// (gather -> segment kernels -> scatter), using the plugin's own column table.
// usage: harness <params.bin>   with the plugin source #included at build time (-DPLUGIN_SRC="...")
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include PLUGIN_SRC

// flat exchange format written by the test: header u64[8] = {row0, nrows, n_const, n_const2, 0,0,0,0}, then 13
// buffers each as {u64 count, data...} in xlp_steps_params order; the three writable ones are written back to <out>.
int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    uint64_t hdr[8];
    if (fread(hdr, 8, 8, f) != 8) return 2;
    std::vector<std::vector<uint64_t>> buf(13);
    for (auto& b : buf) {
        uint64_t n;
        if (fread(&n, 8, 1, f) != 1) return 2;
        b.resize(n);
        if (n && fread(b.data(), 8, n, f) != n) return 2;
    }
    fclose(f);
    xlp_steps_params P;
    memset(&P, 0, sizeof P);
    P.pols = buf[0].data(); P.pols_size = buf[0].size();
    P.const_pols = buf[1].data(); P.n_const_pols = hdr[2]; P.const_pols_size = buf[1].size();
    P.const_pols_2ns = buf[2].data(); P.n_const_pols_2ns = hdr[3]; P.const_pols_2ns_size = buf[2].size();
    P.challenges = buf[3].data(); P.challenges_size = buf[3].size();
    P.public_inputs = buf[4].data(); P.public_inputs_size = buf[4].size();
    P.evals = buf[5].data(); P.evals_size = buf[5].size();
    P.x_n = buf[6].data(); P.x_n_size = buf[6].size();
    P.x_2ns = buf[7].data(); P.x_2ns_size = buf[7].size();
    P.zi = buf[8].data(); P.zi_size = buf[8].size();
    P.x_div_x_sub_xi = buf[9].data(); P.x_div_x_sub_xi_size = buf[9].size();
    P.x_div_x_sub_wxi = buf[10].data(); P.x_div_x_sub_wxi_size = buf[10].size();
    P.q_2ns = buf[11].data(); P.q_2ns_size = buf[11].size();
    P.f_2ns = buf[12].data(); P.f_2ns_size = buf[12].size();
    const uint64_t row0 = hdr[0], nrows = hdr[1];
    const uint64_t ncols = xlpjit_info[0], halo = xlpjit_info[1], modulus = xlpjit_info[2], tmpw = xlpjit_info[3];
    const uint64_t threads = (nrows + 255) / 256 * 256, cap = threads + halo;
    std::vector<uint64_t> cols(ncols * cap + 1, 0xDEADBEEFDEADBEEFULL), tmp(tmpw * threads + 1, 0xDEADBEEFDEADBEEFULL);
    auto base = [&](uint64_t c) -> uint64_t* {
        const uint64_t b = xlpjit_cols[c][0];
        return (b == 0 ? P.pols : b == 1 ? (uint64_t*)P.const_pols : (uint64_t*)P.const_pols_2ns) + xlpjit_cols[c][1];
    };
    auto stride = [&](uint64_t c) { const uint64_t b = xlpjit_cols[c][0]; return b == 0 ? xlpjit_cols[c][2] : b == 1 ? P.n_const_pols : P.n_const_pols_2ns; };
    for (uint64_t c = 0; c < ncols; c++)
        for (uint64_t j = 0; j < nrows + ((xlpjit_cols[c][3] & 1) ? halo : 0); j++) {
            uint64_t ar = row0 + j;
            if (modulus) ar %= modulus;
            cols[c * cap + j] = gl::canon(base(c)[ar * stride(c)]);
        }
    XlpJitArgs A;
    A.cols = cols.data(); A.cap = cap; A.tmp = tmp.data(); A.tstride = threads; A.row0 = row0; A.nrows = nrows; A.P = P;
    if (xlpjit_launch(&A, NULL)) return 3;
    for (uint64_t c = 0; c < ncols; c++)
        if (xlpjit_cols[c][3] & 2)
            for (uint64_t j = 0; j < nrows; j++) base(c)[(row0 + j) * stride(c)] = cols[c * cap + j];
    f = fopen(argv[2], "wb");
    if (!f) return 2;
    const int w[3] = {0, 11, 12};
    for (int k : w) { const uint64_t n = buf[k].size(); fwrite(&n, 8, 1, f); fwrite(buf[k].data(), 8, n, f); }
    fclose(f);
    return 0;
}
