// Constraint-expression VM for gfx950.
//
// Replaces ZkevmSteps::step*_parser_first_avx (src/starkpil/zkevm/chelpers/zkevm.chelpers.*.parser.cpp,
// call sites starks.cpp:73,155,193,241,371): a straight-line program of Goldilocks / cubic-extension
// operations evaluated on every row of the trace.  Encoding: include/xlp_gl.h ("constraint-expression VM").
//
// Execution model: one row per lane.  The program counter, the opcode and every operand descriptor are
// wave-uniform, so decoding runs on the scalar unit (s_load + scalar branches) and only the field
// arithmetic and the operand traffic use vector instructions.  The per-row temporaries (up to ~1200 base
// and ~175 extension slots for the zkEVM's step42ns) do not fit registers or LDS and are indexed by the
// program, so they live in a slot-major HBM scratch `tmp[slot][thread]`: every access of a wave is one
// fully coalesced 512-byte segment and the hot slots stay in L2.  Row-major operand reads (pols, const
// pols) are 8-byte gathers with the row stride of the section, as in the reference; consecutive
// instructions touch neighbouring columns, which the L2 turns into whole-line reuse.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <dlfcn.h>
#include <stdlib.h>
#include <sstream>
#include <algorithm>
#include <array>
#include <map>
#include <set>
#include <string>
#include <tuple>
#include <vector>
#include "gl_field.hpp"
#include "xlp_internal.hpp"
#include "xlp_expr_jit.h"

struct xlp_expr_column {            // one staged column: element (row) = buffer[off + row * stride]
    uint32_t buf;                   // 0 pols, 1 const_pols (stride = n_const_pols), 2 const_pols_2ns
    uint64_t off, stride;
    uint64_t halo;                  // rows past the slab this column is read at (largest next-row distance of its operands)
    bool written;
};
struct xlp_expr_program {
    std::vector<uint64_t> code;   // validated user program (numbers canonical): bounds checks run on this
    uint64_t ntmp1 = 0, ntmp3 = 0;
    uint64_t ninstr = 0;
    // compiled form (see "compile" below)
    std::vector<uint64_t> dcode;
    uint64_t* d_dcode = nullptr;
    uint64_t tmp_words = 0;       // words of temporaries per row
    uint32_t hot_words = 0;       // the first hot_words of them live in LDS
    std::vector<xlp_expr_column> cols;
    std::vector<uint32_t> wcols;  // indices of the written columns
    uint64_t halo = 0, modulus = 0;
    bool serial_slabs = false;    // some row-major operand is addressed directly (not staged): slabs must run one after the other
    bool canonical = false;       // XLP_EXPR_CANONICAL: the column-major buffers hold canonical words (the generated kernels use them as loaded)
    bool colmajor = false;        // XLP_EXPR_COLMAJOR: the buffers are column-major (see include/xlp_gl.h): nothing is staged, every access is direct
    void* jit_handle = nullptr;   // dlopen'ed plugin (xlp_expr_attach)
    void* jit_launch = nullptr;
};

namespace xlp {
using gl::e3;

static const int KIND_PARAMS[XLP_K_COUNT] = {0, 1, 1, 2, 4, 1, 3, 1, 3, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
// dimension an operand kind can have: bit0 = dim 1 allowed, bit1 = dim 3 allowed
static const int KIND_DIMS[XLP_K_COUNT] = {0, 1, 2, 3, 3, 1, 1, 1, 1, 1, 2, 1, 2, 1, 1, 1, 2, 2, 2, 2};
static const char* KIND_NAME[XLP_K_COUNT] = {"none", "tmp1", "tmp3", "pols", "pols'", "const", "const'", "const2ns", "const2ns'", "number",
                                             "challenge", "public", "eval", "x_n", "x_2ns", "zi", "xDivXSubXi", "xDivXSubWXi", "q_2ns", "f_2ns"};

struct Operand { unsigned kind, dim; const uint64_t* p; };
struct Instr { unsigned op; Operand d, a, b; };

// Decode the instruction at code[pc]; returns the next pc or 0 on a structural error (msg set).
static uint64_t decode(const uint64_t* code, uint64_t ncode, uint64_t pc, Instr& in, std::string& msg)
{
    const uint64_t h = code[pc];
    in.op = (unsigned)(h & 0xff);
    in.d.kind = (unsigned)((h >> 8) & 0xff); in.a.kind = (unsigned)((h >> 16) & 0xff); in.b.kind = (unsigned)((h >> 24) & 0xff);
    in.d.dim = (unsigned)((h >> 32) & 3); in.a.dim = (unsigned)((h >> 34) & 3); in.b.dim = (unsigned)((h >> 36) & 3);
    if (h >> 38) { msg = "reserved head bits set"; return 0; }
    if (in.op > XLP_OP_MUL) { msg = "unknown op"; return 0; }
    if (in.d.kind >= XLP_K_COUNT || in.a.kind >= XLP_K_COUNT || in.b.kind >= XLP_K_COUNT) { msg = "unknown operand kind"; return 0; }
    uint64_t q = pc + 1;
    Operand* ops[3] = {&in.d, &in.a, &in.b};
    for (Operand* o : ops) {
        o->p = code + q;
        q += KIND_PARAMS[o->kind];
        if (q > ncode) { msg = "truncated instruction"; return 0; }
    }
    return q;
}

static bool dim_ok(const Operand& o) { return o.dim == 1 ? (KIND_DIMS[o.kind] & 1) : o.dim == 3 ? (KIND_DIMS[o.kind] & 2) : false; }

static int validate(const uint64_t* code, uint64_t ncode, uint64_t ntmp1, uint64_t ntmp3, uint64_t& ninstr)
{
    std::string msg;
    uint64_t pc = 0;
    ninstr = 0;
    while (pc < ncode) {
        Instr in;
        const uint64_t nx = decode(code, ncode, pc, in, msg);
        auto fail = [&](const std::string& m) { return xlp_fail(("expr program: instruction " + std::to_string(ninstr) + " (word " + std::to_string(pc) + "): " + m).c_str()); };
        if (!nx) return fail(msg);
        const unsigned dk = in.d.kind;
        if (!(dk == XLP_K_TMP1 || dk == XLP_K_TMP3 || dk == XLP_K_POLS || dk == XLP_K_POLS_PRIME || dk == XLP_K_Q_2NS || dk == XLP_K_F_2NS))
            return fail(std::string("kind ") + KIND_NAME[dk] + " cannot be a destination");
        if (in.a.kind == XLP_K_NONE) return fail("missing first source");
        if ((in.op == XLP_OP_COPY) != (in.b.kind == XLP_K_NONE)) return fail("COPY takes one source, ADD/SUB/MUL two");
        if (!dim_ok(in.d) || !dim_ok(in.a) || (in.b.kind != XLP_K_NONE && !dim_ok(in.b))) return fail("operand dimension does not fit its kind");
        const unsigned rd = in.b.kind == XLP_K_NONE ? in.a.dim : (in.a.dim > in.b.dim ? in.a.dim : in.b.dim);
        if (rd != in.d.dim) return fail("destination dimension != max(source dimensions)");
        const Operand* ops[3] = {&in.d, &in.a, &in.b};
        for (const Operand* o : ops) {
            if (o->kind == XLP_K_TMP1 && o->p[0] >= ntmp1) return fail("tmp1 index out of range");
            if (o->kind == XLP_K_TMP3 && o->p[0] >= ntmp3) return fail("tmp3 index out of range");
            if ((o->kind == XLP_K_POLS_PRIME && o->p[2] == 0) || ((o->kind == XLP_K_CONST_PRIME || o->kind == XLP_K_CONST2NS_PRIME) && o->p[2] == 0))
                return fail("prime operand with modulus 0");
        }
        pc = nx;
        ninstr++;
    }
    return 0;
}

// Largest element index an operand touches for rows [row0, row0 + nrows), and the buffer it lives in.
// Column-major programs (XLP_EXPR_COLMAJOR): one launch range = storage rows [s0, s1] of coset `coset` (base-domain programs: coset 0, the
// whole range); cbase = the coset's first storage row, dshift = storage row - dense index of the x / xDivXSubXi tables.
static int check_bounds_cm(const xlp_expr_program* prog, const xlp_steps_params& P, u64 s0, u64 s1, u64 coset, u64 cbase, u64 dshift)
{
    const uint64_t* code = prog->code.data();
    const uint64_t ncode = prog->code.size();
    typedef unsigned __int128 u128;
    auto top = [&](u64 nx, u64 md) -> u128 {               // highest storage row a next-row operand reads (the wrapping rows read lower ones)
        const u128 t = (u128)s1 + nx;
        return (t - cbase >= md) ? (u128)(cbase + md - 1 < s1 ? s1 : cbase + md - 1) : t;
    };
    const u64 crows = P.n_const_pols ? P.const_pols_size / P.n_const_pols : 0, crows2 = P.n_const_pols_2ns ? P.const_pols_2ns_size / P.n_const_pols_2ns : 0;
    std::string msg;
    uint64_t pc = 0, k = 0;
    while (pc < ncode) {
        Instr in;
        const uint64_t nx = decode(code, ncode, pc, in, msg);
        const Operand* ops[3] = {&in.d, &in.a, &in.b};
        for (const Operand* o : ops) {
            u128 hi = 0;
            uint64_t size = 0;
            const void* base = nullptr;
            const uint64_t* p = o->p;
            switch (o->kind) {
            case XLP_K_POLS: hi = (u128)p[0] + (u128)(o->dim - 1) * p[1] + s1; size = P.pols_size; base = P.pols; break;
            case XLP_K_POLS_PRIME: hi = (u128)p[0] + (u128)(o->dim - 1) * p[3] + top(p[1], p[2]); size = P.pols_size; base = P.pols; break;
            case XLP_K_CONST: case XLP_K_CONST_PRIME: {
                const bool pr = o->kind == XLP_K_CONST_PRIME;
                const u128 r = pr ? top(p[1], p[2]) : (u128)s1;
                hi = (p[0] >= P.n_const_pols || r >= crows) ? ~(u128)0 : (u128)p[0] * crows + r;
                size = P.const_pols_size; base = P.const_pols; break;
            }
            case XLP_K_CONST2NS: case XLP_K_CONST2NS_PRIME: {
                const bool pr = o->kind == XLP_K_CONST2NS_PRIME;
                const u128 r = pr ? top(p[1], p[2]) : (u128)s1;
                hi = (p[0] >= P.n_const_pols_2ns || r >= crows2) ? ~(u128)0 : (u128)p[0] * crows2 + r;
                size = P.const_pols_2ns_size; base = P.const_pols_2ns; break;
            }
            case XLP_K_CHALLENGE: hi = (u128)p[0] * 3 + 2; size = P.challenges_size; base = P.challenges; break;
            case XLP_K_EVAL: hi = (u128)p[0] * 3 + 2; size = P.evals_size; base = P.evals; break;
            case XLP_K_PUBLIC: hi = p[0]; size = P.public_inputs_size; base = P.public_inputs; break;
            case XLP_K_X_N: hi = s1 - dshift; size = P.x_n_size; base = P.x_n; break;
            case XLP_K_X_2NS: hi = s1 - dshift; size = P.x_2ns_size; base = P.x_2ns; break;
            case XLP_K_ZI: hi = coset; size = P.zi_size; base = P.zi; break;
            case XLP_K_XDIVXSUBXI: hi = (u128)(s1 - dshift) * 3 + 2; size = P.x_div_x_sub_xi_size; base = P.x_div_x_sub_xi; break;
            case XLP_K_XDIVXSUBWXI: hi = (u128)(s1 - dshift) * 3 + 2; size = P.x_div_x_sub_wxi_size; base = P.x_div_x_sub_wxi; break;
            case XLP_K_Q_2NS: hi = ((u128)(s1 - cbase) * P.zi_size + coset) * 3 + 2; size = P.q_2ns_size; base = P.q_2ns; break;
            case XLP_K_F_2NS: hi = ((u128)(s1 - cbase) * P.zi_size + coset) * 3 + 2; size = P.f_2ns_size; base = P.f_2ns; break;
            default: continue;
            }
            if (base == nullptr || hi >= size) {
                const std::string m = "expr run (column-major): instruction " + std::to_string(k) + ": operand " + KIND_NAME[o->kind] +
                                      (base == nullptr ? " has no buffer" : " reads/writes past the end of its buffer") + " for storage rows [" +
                                      std::to_string(s0) + ", " + std::to_string(s1 + 1) + ") of coset " + std::to_string(coset);
                return xlp_fail(m.c_str());
            }
        }
        pc = nx;
        k++;
    }
    return 0;
}

static int check_bounds(const xlp_expr_program* prog, const xlp_steps_params& P, uint64_t row0, uint64_t nrows)
{
    const uint64_t* code = prog->code.data();
    const uint64_t ncode = prog->code.size();
    const uint64_t last = row0 + nrows - 1;
    // highest row a next-row operand (distance nx, modulus md) reads: the whole domain if the range wraps, else last + nx.
    // A run over a row WINDOW of a larger domain (one rank's shard + its halo rows, shard.py) therefore only needs the
    // window to be resident.
    auto top_row = [&](uint64_t nx, uint64_t md) -> unsigned __int128 {
        const unsigned __int128 t = (unsigned __int128)last + nx;
        return t < md ? t : (unsigned __int128)(md - 1);
    };
    std::string msg;
    uint64_t pc = 0, k = 0;
    while (pc < ncode) {
        Instr in;
        const uint64_t nx = decode(code, ncode, pc, in, msg);
        const Operand* ops[3] = {&in.d, &in.a, &in.b};
        for (const Operand* o : ops) {
            unsigned __int128 hi = 0;   // highest element index touched
            uint64_t size = 0;
            const void* base = nullptr;
            const uint64_t* p = o->p;
            switch (o->kind) {
            case XLP_K_POLS: hi = (unsigned __int128)p[0] + (unsigned __int128)last * p[1] + o->dim - 1; size = P.pols_size; base = P.pols; break;
            case XLP_K_POLS_PRIME: hi = (unsigned __int128)p[0] + top_row(p[1], p[2]) * p[3] + o->dim - 1; size = P.pols_size; base = P.pols; break;
            case XLP_K_CONST:
                if (p[0] >= P.n_const_pols) { hi = ~(unsigned __int128)0; }
                else hi = (unsigned __int128)p[0] + (unsigned __int128)last * P.n_const_pols;
                size = P.const_pols_size; base = P.const_pols; break;
            case XLP_K_CONST_PRIME:
                if (p[0] >= P.n_const_pols) { hi = ~(unsigned __int128)0; }
                else hi = (unsigned __int128)p[0] + top_row(p[1], p[2]) * P.n_const_pols;
                size = P.const_pols_size; base = P.const_pols; break;
            case XLP_K_CONST2NS:
                if (p[0] >= P.n_const_pols_2ns) { hi = ~(unsigned __int128)0; }
                else hi = (unsigned __int128)p[0] + (unsigned __int128)last * P.n_const_pols_2ns;
                size = P.const_pols_2ns_size; base = P.const_pols_2ns; break;
            case XLP_K_CONST2NS_PRIME:
                if (p[0] >= P.n_const_pols_2ns) { hi = ~(unsigned __int128)0; }
                else hi = (unsigned __int128)p[0] + top_row(p[1], p[2]) * P.n_const_pols_2ns;
                size = P.const_pols_2ns_size; base = P.const_pols_2ns; break;
            case XLP_K_CHALLENGE: hi = (unsigned __int128)p[0] * 3 + 2; size = P.challenges_size; base = P.challenges; break;
            case XLP_K_EVAL: hi = (unsigned __int128)p[0] * 3 + 2; size = P.evals_size; base = P.evals; break;
            case XLP_K_PUBLIC: hi = p[0]; size = P.public_inputs_size; base = P.public_inputs; break;
            case XLP_K_X_N: hi = last; size = P.x_n_size; base = P.x_n; break;
            case XLP_K_X_2NS: hi = last; size = P.x_2ns_size; base = P.x_2ns; break;
            case XLP_K_ZI: hi = 0; size = P.zi_size; base = P.zi; break;
            case XLP_K_XDIVXSUBXI: hi = (unsigned __int128)last * 3 + 2; size = P.x_div_x_sub_xi_size; base = P.x_div_x_sub_xi; break;
            case XLP_K_XDIVXSUBWXI: hi = (unsigned __int128)last * 3 + 2; size = P.x_div_x_sub_wxi_size; base = P.x_div_x_sub_wxi; break;
            case XLP_K_Q_2NS: hi = (unsigned __int128)last * 3 + 2; size = P.q_2ns_size; base = P.q_2ns; break;
            case XLP_K_F_2NS: hi = (unsigned __int128)last * 3 + 2; size = P.f_2ns_size; base = P.f_2ns; break;
            default: continue;   // temporaries, numbers
            }
            if (base == nullptr || hi >= size) {
                const std::string m = "expr run: instruction " + std::to_string(k) + ": operand " + KIND_NAME[o->kind] +
                                      (base == nullptr ? " has no buffer" : " reads/writes past the end of its buffer") + " for rows [" +
                                      std::to_string(row0) + ", " + std::to_string(row0 + nrows) + ")";
                return xlp_fail(m.c_str());
            }
        }
        pc = nx;
        k++;
    }
    return 0;
}

// --------------------------------------------------------------------------------------- compile
// The user program is rewritten once, at xlp_expr_create, into a device program:
//  * temporaries: tmp1/tmp3 indices become word offsets in one per-row array, ordered by how often the
//    program touches them.  On the zkEVM programs 16 words take 90-100 % of all temporary accesses
//    (the generators keep their accumulators in a few slots), so the first `hot_words` live in LDS
//    (hot[word][thread], conflict-free ds_read_b64) and only the tail spills to the HBM scratch.
//  * row-major operands (pols / const pols): an 8-byte gather per lane with the section's row stride costs a
//    whole 128-byte line per element.  Every column the program names is therefore "staged": a gather kernel
//    transposes the chunk's rows of exactly those columns into a column-major scratch cols[c][row] (canonical
//    values, + `halo` rows for next-row operands), the VM reads and writes them fully coalesced, and a scatter
//    kernel writes the destination columns back.  Operands that cannot be staged (a destination addressed
//    with a row offset, a second modulus) keep the direct path; a column is either staged or direct, never both.
//  * instruction fetch: device instructions are fixed 8-word slots (16 for the rare ones with more than 7
//    parameter words), 8 slots = one 64-word window.  A wave holds the current window in one VGPR pair (lane i =
//    word i), loads the next one with a single coalesced global_load at the start of the current one, and
//    decodes with v_readlane -- so the fetch latency is hidden 8 instructions ahead instead of sitting, as a
//    chain of dependent scalar loads that miss the 16 KiB scalar cache, in front of every instruction.
enum { XD_TMP = 20, XD_COL = 21, XD_NOP = 7 /* op */ };

static int env_int(const char* name, int dflt)
{
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

struct ColKey {
    u32 buf; u64 stride, off;
    bool operator<(const ColKey& o) const { return std::tie(buf, stride, off) < std::tie(o.buf, o.stride, o.off); }
};

// (buffer, offset, stride, next, modulus) of a row-major operand; false if the kind is not row-major
static bool rowmajor(const Operand& o, ColKey& k, u64& next, u64& mod)
{
    next = 0; mod = 0;
    switch (o.kind) {
    case XLP_K_POLS: k = {0, o.p[1], o.p[0]}; return true;
    case XLP_K_POLS_PRIME: k = {0, o.p[3], o.p[0]}; next = o.p[1]; mod = o.p[2]; return true;
    case XLP_K_CONST: k = {1, 0, o.p[0]}; return true;
    case XLP_K_CONST_PRIME: k = {1, 0, o.p[0]}; next = o.p[1]; mod = o.p[2]; return true;
    case XLP_K_CONST2NS: k = {2, 0, o.p[0]}; return true;
    case XLP_K_CONST2NS_PRIME: k = {2, 0, o.p[0]}; next = o.p[1]; mod = o.p[2]; return true;
    default: return false;
    }
}

static void compile(xlp_expr_program* P)
{
    const u64* code = P->code.data();
    const u64 ncode = P->code.size();
    std::string msg;
    // ---- temporaries: access counts -> word offsets, densest first
    std::vector<u64> cnt1(P->ntmp1, 0), cnt3(P->ntmp3, 0);
    for (u64 pc = 0; pc < ncode;) {
        Instr in; const u64 nx = decode(code, ncode, pc, in, msg);
        const Operand* ops[3] = {&in.d, &in.a, &in.b};
        for (const Operand* o : ops) {
            if (o->kind == XLP_K_TMP1) cnt1[o->p[0]]++;
            if (o->kind == XLP_K_TMP3) cnt3[o->p[0]]++;
        }
        pc = nx;
    }
    struct Slot { u64 cnt; u32 words; u64 idx; };
    std::vector<Slot> slots;
    for (u64 i = 0; i < P->ntmp1; i++) slots.push_back({cnt1[i], 1, i});
    for (u64 i = 0; i < P->ntmp3; i++) slots.push_back({cnt3[i], 3, i});
    std::stable_sort(slots.begin(), slots.end(), [](const Slot& a, const Slot& b) { return a.cnt > b.cnt; });   // accesses per slot: a
    std::vector<u64> w1(P->ntmp1, 0), w3(P->ntmp3, 0);                                                        // slot moves as a unit
    u64 w = 0;
    for (const Slot& s : slots) { (s.words == 1 ? w1 : w3)[s.idx] = w; w += s.words; }
    P->tmp_words = w;
    u64 hot_cap = (u64)env_int("XLP_EXPR_HOT", 8);
    if (hot_cap > 32) hot_cap = 32;                               // 32 words x 256 threads x 8 B = the 64 KiB a launch may ask for
    u64 hot = 0;
    for (const Slot& s : slots) { if (hot + s.words > hot_cap) break; hot += s.words; }
    P->hot_words = (u32)hot;

    // ---- staging decisions
    const bool stage = env_int("XLP_EXPR_STAGE", 1) != 0 && !P->colmajor;   // column-major buffers: a lane's row of a column is coalesced as it is
    std::map<ColKey, std::pair<u64, bool>> colset;   // key -> (halo rows, written)
    std::set<ColKey> blocked;
    u64 modulus = 0, halo = 0;
    if (stage) {
        for (u64 pc = 0; pc < ncode;) {
            Instr in; const u64 nx = decode(code, ncode, pc, in, msg);
            const Operand* ops[3] = {&in.d, &in.a, &in.b};
            for (int s = 0; s < 3; s++) {
                ColKey k; u64 next, mod;
                if (!rowmajor(*ops[s], k, next, mod)) continue;
                bool direct = false;
                if (mod) {
                    if (s == 0) direct = true;                       // destination with a row offset
                    else if (modulus == 0) modulus = mod;
                    else if (mod != modulus) direct = true;          // a second modulus
                    if (next >= mod) direct = true;
                }
                if (direct) for (u32 c = 0; c < ops[s]->dim; c++) blocked.insert({k.buf, k.stride, k.off + c});
            }
            pc = nx;
        }
        for (bool grown = !blocked.empty(); grown;) {          // an operand that shares a column with a direct one is direct too
            grown = false;
            for (u64 pc = 0; pc < ncode;) {
                Instr in; const u64 nx = decode(code, ncode, pc, in, msg);
                const Operand* ops[3] = {&in.d, &in.a, &in.b};
                for (int s = 0; s < 3; s++) {
                    ColKey k; u64 next, mod;
                    if (!rowmajor(*ops[s], k, next, mod)) continue;
                    bool hit = false;
                    for (u32 c = 0; c < ops[s]->dim; c++) hit |= blocked.count({k.buf, k.stride, k.off + c}) != 0;
                    if (hit) for (u32 c = 0; c < ops[s]->dim; c++) grown |= blocked.insert({k.buf, k.stride, k.off + c}).second;
                }
                pc = nx;
            }
        }
        for (u64 pc = 0; pc < ncode;) {
            Instr in; const u64 nx = decode(code, ncode, pc, in, msg);
            const Operand* ops[3] = {&in.d, &in.a, &in.b};
            for (int s = 0; s < 3; s++) {
                ColKey k; u64 next, mod;
                if (!rowmajor(*ops[s], k, next, mod)) continue;
                if (blocked.count({k.buf, k.stride, k.off})) continue;
                for (u32 c = 0; c < ops[s]->dim; c++) {
                    std::pair<u64, bool>& f = colset[{k.buf, k.stride, k.off + c}];
                    if (mod && next) { if (next > f.first) f.first = next; if (next > halo) halo = next; }
                    if (s == 0) f.second = true;
                }
            }
            pc = nx;
        }
    }
    std::map<ColKey, u32> colidx;
    for (const auto& kv : colset) {
        colidx[kv.first] = (u32)P->cols.size();
        if (kv.second.second) P->wcols.push_back((u32)P->cols.size());
        P->cols.push_back({kv.first.buf, kv.first.off, kv.first.stride, kv.second.first, kv.second.second});
    }
    P->halo = halo; P->modulus = modulus;
    // Direct (unstaged) columns are touched by the program's kernels themselves.  Slabs may run side by side as long as no direct
    // column is both written and read: a row then writes its own cell (or its neighbour's, for a next-row store) exactly once and
    // nobody looks at it before the program ends.
    {
        std::set<ColKey> dwr, drd;
        for (u64 pc = 0; pc < ncode;) {
            Instr in; const u64 nx = decode(code, ncode, pc, in, msg);
            const Operand* ops[3] = {&in.d, &in.a, &in.b};
            for (int s = 0; s < 3; s++) {
                ColKey k; u64 next, mod;
                if (!rowmajor(*ops[s], k, next, mod)) continue;
                for (u32 c = 0; c < ops[s]->dim; c++)
                    if (blocked.count({k.buf, k.stride, k.off + c})) (s == 0 ? dwr : drd).insert({k.buf, k.stride, k.off + c});
            }
            pc = nx;
        }
        P->serial_slabs = false;
        for (const ColKey& k : dwr) if (drd.count(k)) P->serial_slabs = true;
    }
    if (P->colmajor) {
        // every access is direct.  A row that only touches ITS cells of a column may run in any order with the others; a column that is
        // written at one row distance and touched at another (next-row stores picked up by the next row: zkEVM step3prev, step3) orders
        // the rows through the segment boundaries, so its slabs run one after the other
        std::map<std::pair<u32, u64>, std::set<u64>> wr, any;
        for (u64 pc = 0; pc < ncode;) {
            Instr in; const u64 nx = decode(code, ncode, pc, in, msg);
            const Operand* ops[3] = {&in.d, &in.a, &in.b};
            for (int s_ = 0; s_ < 3; s_++) {
                const Operand& o = *ops[s_];
                u32 buf; u64 off, cs = 0, next = 0;
                switch (o.kind) {
                case XLP_K_POLS: buf = 0; off = o.p[0]; cs = o.p[1]; break;
                case XLP_K_POLS_PRIME: buf = 0; off = o.p[0]; cs = o.p[3]; next = o.p[1]; break;
                default: continue;
                }
                for (u32 c = 0; c < o.dim; c++) { any[{buf, off + c * cs}].insert(next); if (s_ == 0) wr[{buf, off + c * cs}].insert(next); }
            }
            pc = nx;
        }
        P->serial_slabs = false;
        for (const auto& kv : wr) if (any[kv.first].size() > 1) P->serial_slabs = true;
    }

    // ---- device program
    for (u64 pc = 0; pc < ncode;) {
        Instr in; const u64 nx = decode(code, ncode, pc, in, msg);
        const Operand* ops[3] = {&in.d, &in.a, &in.b};
        u32 kinds[3], np[3];
        std::vector<u64> prm[3];
        for (int s = 0; s < 3; s++) {
            const Operand& o = *ops[s];
            ColKey k; u64 next, mod;
            if (o.kind == XLP_K_TMP1) { kinds[s] = XD_TMP; prm[s] = {w1[o.p[0]]}; }
            else if (o.kind == XLP_K_TMP3) { kinds[s] = XD_TMP; prm[s] = {w3[o.p[0]]}; }
            else if (rowmajor(o, k, next, mod) && colidx.count(k)) { kinds[s] = XD_COL; prm[s] = {colidx[k], next}; }
            else { kinds[s] = o.kind; prm[s].assign(o.p, o.p + KIND_PARAMS[o.kind]); }
            np[s] = (u32)prm[s].size();
        }
        const u32 nprm = np[0] + np[1] + np[2];
        const bool lng = nprm > 7;
        if (lng && (P->dcode.size() / 8) % 8 == 7) { P->dcode.push_back(XD_NOP); P->dcode.resize(P->dcode.size() + 7, 0); }   // keep both halves in one window
        const u64 head = (u64)in.op | (u64)kinds[0] << 8 | (u64)kinds[1] << 16 | (u64)kinds[2] << 24 | (u64)in.d.dim << 32 | (u64)in.a.dim << 34 |
                         (u64)in.b.dim << 36 | (u64)np[0] << 40 | (u64)np[1] << 43 | (u64)np[2] << 46 | (u64)(lng ? 1 : 0) << 49;
        const size_t start = P->dcode.size();
        P->dcode.push_back(head);
        for (int s = 0; s < 3; s++) P->dcode.insert(P->dcode.end(), prm[s].begin(), prm[s].end());
        P->dcode.resize(start + (lng ? 16 : 8), 0);
        pc = nx;
    }
    while ((P->dcode.size() / 8) % 8) { P->dcode.push_back(XD_NOP); P->dcode.resize(P->dcode.size() + 7, 0); }                    // whole windows
}

// ------------------------------------------------------------------------------------------ device
struct ExprArgs {
    u32 colmajor;                             // column-major addressing (see check_bounds_cm): coset / cbase / dshift describe this launch
    u64 coset, cbase, dshift, crows, crows2;  // crows / crows2: rows per column of const_pols / const_pols_2ns
    const u64* code; u64 ncode;
    u64* tmp; u64 tstride; u32 hot_words;     // spilled temporaries tmp[word - hot_words][thread]
    const u64* cols; u64 cap;                 // staged columns cols[c][local row], cap rows each
    u64 row0, nrows;                          // absolute first row, rows of this launch
    xlp_steps_params P;
};
struct ColDesc { u64* base; u64 stride; u64 halo; };    // halo: rows past the slab to stage for this column

__device__ __forceinline__ u64 uni(u64 v)   // value is wave-uniform: keep it in SGPRs
{
    return ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)v);   // the builtin returns int
}
__device__ __forceinline__ u64 wrap(u64 r, u64 n) { return (n & (n - 1)) == 0 ? (r & (n - 1)) : r % n; }

extern __shared__ u64 xlp_expr_hot[];

__device__ __forceinline__ u64 rl(u64 win, u32 j)   // word j of the window (j wave-uniform)
{
    return ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(win >> 32), (int)j) << 32) | (u64)(u32)__builtin_amdgcn_readlane((int)(u32)win, (int)j);
}

#define PRM(i) rl(win, p + (i))
// R rows per lane: row r of a lane is virtual thread vt = gt + r * T (T = threads of the launch), so every
// per-row access below stays coalesced; decoding (scalar unit, shared by the 4 SIMDs of a CU and the bottleneck
// of a one-row-per-lane interpreter) is paid once per R rows.
template <int R>
struct Rows { e3 r[R]; };

template <int R>
__device__ __forceinline__ void load_operand(Rows<R>& out, const ExprArgs& A, u32 kind, u32 dim, u64 win, u32 p, const u64 (&ro)[R], u64 gt, u64 T)
{
#pragma unroll
    for (int r = 0; r < R; r++) out.r[r].c[0] = out.r[r].c[1] = out.r[r].c[2] = 0;
    const u64* src = nullptr;
    u64 rstride = 0;                       // element distance between consecutive rows of `src`
    if (A.colmajor && kind >= XLP_K_POLS && kind <= XLP_K_CONST2NS_PRIME) {     // element c of row i = base[off + c * cs + wrap(i + next)]
        const bool pr = kind == XLP_K_POLS_PRIME || kind == XLP_K_CONST_PRIME || kind == XLP_K_CONST2NS_PRIME;
        const bool pol = kind == XLP_K_POLS || kind == XLP_K_POLS_PRIME;
        const u64* base = pol ? A.P.pols : (kind == XLP_K_CONST || kind == XLP_K_CONST_PRIME) ? A.P.const_pols : A.P.const_pols_2ns;
        const u64 cs = pol ? (pr ? PRM(3) : PRM(1)) : 0;
        const u64 off = pol ? PRM(0) : PRM(0) * ((kind == XLP_K_CONST || kind == XLP_K_CONST_PRIME) ? A.crows : A.crows2);
        const u64 nx = pr ? PRM(1) : 0, md = pr ? PRM(2) : 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            u64 j = ro[r] + nx;
            if (pr && j - A.cbase >= md) j -= md;
            const u64* q = base + off + j;
            out.r[r].c[0] = gl::canon(q[0]);
            if (dim == 3) { out.r[r].c[1] = gl::canon(q[cs]); out.r[r].c[2] = gl::canon(q[2 * cs]); }
        }
        return;
    }
    if (A.colmajor && (kind == XLP_K_ZI || kind == XLP_K_Q_2NS || kind == XLP_K_F_2NS)) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (kind == XLP_K_ZI) { out.r[r].c[0] = gl::canon(A.P.zi[A.coset]); continue; }
            const u64* q = (kind == XLP_K_Q_2NS ? A.P.q_2ns : A.P.f_2ns) + 3 * ((ro[r] - A.cbase) * A.P.zi_size + A.coset);
            out.r[r].c[0] = gl::canon(q[0]); out.r[r].c[1] = gl::canon(q[1]); out.r[r].c[2] = gl::canon(q[2]);
        }
        return;
    }
    const u64 dsh = A.colmajor ? A.dshift : 0;   // the dense tables (x_n, x_2ns, xDivXSubXi) are indexed by storage row - dshift
    switch (kind) {
    case XD_TMP: {
        const u64 w = PRM(0);
        if (w < A.hot_words) {                                  // a slot never straddles the boundary
            const u64* q = xlp_expr_hot + w * (R * 256) + threadIdx.x;
#pragma unroll
            for (int r = 0; r < R; r++) {
                out.r[r].c[0] = q[r * 256];
                if (dim == 3) { out.r[r].c[1] = q[(R + r) * 256]; out.r[r].c[2] = q[(2 * R + r) * 256]; }
            }
        } else {
            const u64* q = A.tmp + (w - A.hot_words) * A.tstride + gt;
#pragma unroll
            for (int r = 0; r < R; r++) {
                out.r[r].c[0] = q[r * T];
                if (dim == 3) { out.r[r].c[1] = q[A.tstride + r * T]; out.r[r].c[2] = q[2 * A.tstride + r * T]; }
            }
        }
        return;
    }
    case XD_COL: {
        const u64* q = A.cols + PRM(0) * A.cap + gt + PRM(1);
#pragma unroll
        for (int r = 0; r < R; r++) {
            out.r[r].c[0] = q[r * T];
            if (dim == 3) { out.r[r].c[1] = q[A.cap + r * T]; out.r[r].c[2] = q[2 * A.cap + r * T]; }
        }
        return;                                                 // staged values are canonical already
    }
    case XLP_K_NUMBER: {
        const u64 v = PRM(0);
#pragma unroll
        for (int r = 0; r < R; r++) out.r[r].c[0] = v;
        return;
    }
    case XLP_K_POLS_PRIME: case XLP_K_CONST_PRIME: case XLP_K_CONST2NS_PRIME: {     // direct next-row operands (rare): per row
        const u64* base = kind == XLP_K_POLS_PRIME ? A.P.pols : kind == XLP_K_CONST_PRIME ? A.P.const_pols : A.P.const_pols_2ns;
        const u64 st = kind == XLP_K_POLS_PRIME ? PRM(3) : kind == XLP_K_CONST_PRIME ? A.P.n_const_pols : A.P.n_const_pols_2ns;
        const u64 off = PRM(0), nx = PRM(1), md = PRM(2);
#pragma unroll
        for (int r = 0; r < R; r++) {
            const u64* q = base + off + wrap(ro[r] + nx, md) * st;
            out.r[r].c[0] = gl::canon(q[0]);
            if (dim == 3) { out.r[r].c[1] = gl::canon(q[1]); out.r[r].c[2] = gl::canon(q[2]); }
        }
        return;
    }
    case XLP_K_ZI: {
#pragma unroll
        for (int r = 0; r < R; r++) out.r[r].c[0] = gl::canon(A.P.zi[wrap(ro[r], A.P.zi_size)]);
        return;
    }
    case XLP_K_POLS: rstride = PRM(1); src = A.P.pols + PRM(0); break;
    case XLP_K_CONST: rstride = A.P.n_const_pols; src = A.P.const_pols + PRM(0); break;
    case XLP_K_CONST2NS: rstride = A.P.n_const_pols_2ns; src = A.P.const_pols_2ns + PRM(0); break;
    case XLP_K_CHALLENGE: src = A.P.challenges + 3 * PRM(0); break;
    case XLP_K_EVAL: src = A.P.evals + 3 * PRM(0); break;
    case XLP_K_PUBLIC: src = A.P.public_inputs + PRM(0); break;
    case XLP_K_X_N: rstride = 1; src = A.P.x_n - dsh; break;
    case XLP_K_X_2NS: rstride = 1; src = A.P.x_2ns - dsh; break;
    case XLP_K_XDIVXSUBXI: rstride = 3; src = A.P.x_div_x_sub_xi - 3 * dsh; break;
    case XLP_K_XDIVXSUBWXI: rstride = 3; src = A.P.x_div_x_sub_wxi - 3 * dsh; break;
    case XLP_K_Q_2NS: rstride = 3; src = A.P.q_2ns; break;
    case XLP_K_F_2NS: rstride = 3; src = A.P.f_2ns; break;
    default: return;
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        const u64* q = src + ro[r] * rstride;
        out.r[r].c[0] = gl::canon(q[0]);
        if (dim == 3) { out.r[r].c[1] = gl::canon(q[1]); out.r[r].c[2] = gl::canon(q[2]); }
    }
}

template <int R>
__device__ __forceinline__ void store_operand(const ExprArgs& A, u32 kind, u32 dim, u64 win, u32 p, const u64 (&ro)[R], u64 gt, u64 T, u64 nrows, const Rows<R>& v)
{
    u64* dst = nullptr;
    u64 rstride = 0;
    if (A.colmajor && (kind == XLP_K_POLS || kind == XLP_K_POLS_PRIME)) {
        const bool pr = kind == XLP_K_POLS_PRIME;
        const u64 off = PRM(0), cs = pr ? PRM(3) : PRM(1), nx = pr ? PRM(1) : 0, md = pr ? PRM(2) : 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (gt + r * T >= nrows) continue;
            u64 j = ro[r] + nx;
            if (pr && j - A.cbase >= md) j -= md;
            u64* q = A.P.pols + off + j;
            q[0] = v.r[r].c[0];
            if (dim == 3) { q[cs] = v.r[r].c[1]; q[2 * cs] = v.r[r].c[2]; }
        }
        return;
    }
    if (A.colmajor && (kind == XLP_K_Q_2NS || kind == XLP_K_F_2NS)) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (gt + r * T >= nrows) continue;
            u64* q = (kind == XLP_K_Q_2NS ? A.P.q_2ns : A.P.f_2ns) + 3 * ((ro[r] - A.cbase) * A.P.zi_size + A.coset);
            q[0] = v.r[r].c[0]; q[1] = v.r[r].c[1]; q[2] = v.r[r].c[2];
        }
        return;
    }
    switch (kind) {
    case XD_TMP: {                                               // rows past the end compute garbage in their own slots: harmless
        const u64 w = PRM(0);
        if (w < A.hot_words) {
            u64* q = xlp_expr_hot + w * (R * 256) + threadIdx.x;
#pragma unroll
            for (int r = 0; r < R; r++) {
                q[r * 256] = v.r[r].c[0];
                if (dim == 3) { q[(R + r) * 256] = v.r[r].c[1]; q[(2 * R + r) * 256] = v.r[r].c[2]; }
            }
        } else {
            u64* q = A.tmp + (w - A.hot_words) * A.tstride + gt;
#pragma unroll
            for (int r = 0; r < R; r++) {
                q[r * T] = v.r[r].c[0];
                if (dim == 3) { q[A.tstride + r * T] = v.r[r].c[1]; q[2 * A.tstride + r * T] = v.r[r].c[2]; }
            }
        }
        return;
    }
    case XD_COL: {
        u64* q = const_cast<u64*>(A.cols) + PRM(0) * A.cap + gt;
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (gt + r * T >= nrows) continue;
            q[r * T] = v.r[r].c[0];
            if (dim == 3) { q[A.cap + r * T] = v.r[r].c[1]; q[2 * A.cap + r * T] = v.r[r].c[2]; }
        }
        return;
    }
    case XLP_K_POLS_PRIME: {
        const u64 off = PRM(0), nx = PRM(1), md = PRM(2), st = PRM(3);
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (gt + r * T >= nrows) continue;
            u64* q = A.P.pols + off + wrap(ro[r] + nx, md) * st;
            q[0] = v.r[r].c[0];
            if (dim == 3) { q[1] = v.r[r].c[1]; q[2] = v.r[r].c[2]; }
        }
        return;
    }
    case XLP_K_POLS: rstride = PRM(1); dst = A.P.pols + PRM(0); break;
    case XLP_K_Q_2NS: rstride = 3; dst = A.P.q_2ns; break;
    case XLP_K_F_2NS: rstride = 3; dst = A.P.f_2ns; break;
    default: return;
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        if (gt + r * T >= nrows) continue;
        u64* q = dst + ro[r] * rstride;
        q[0] = v.r[r].c[0];
        if (dim == 3) { q[1] = v.r[r].c[1]; q[2] = v.r[r].c[2]; }
    }
}

// Lane gt evaluates rows row0 + gt + r * T, r < R (T = threads of the launch >= ceil(nrows / R)).
template <int R>
__global__ __launch_bounds__(256) void expr_kernel(ExprArgs A)
{
    const u64 T = (u64)gridDim.x * blockDim.x;
    const u64 gt = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 lane = threadIdx.x & 63;
    u64 ro[R];       // absolute row of each of the lane's rows; rows past the end read the last row (in bounds), their stores are masked
#pragma unroll
    for (int r = 0; r < R; r++) { const u64 vt = gt + r * T; ro[r] = A.row0 + (vt < A.nrows ? vt : A.nrows - 1); }
    u64 win = A.code[lane];
    asm volatile("" : "+v"(win));              // wait for the first window HERE: left pending, the wait lands in the loop header
    for (u64 base = 0; base < A.ncode; base += 64) {
        const u64 nxt = base + 64 < A.ncode ? A.code[base + 64 + lane] : 0;
        for (u32 s = 0; s < 8;) {
            const u64 h = rl(win, 8 * s);
            const u32 op = (u32)(h & 0xff);
            const bool lng = (h >> 49) & 1;
            if (op == XD_NOP) { s++; continue; }
            const u32 dk = (u32)((h >> 8) & 0xff), ak = (u32)((h >> 16) & 0xff), bk = (u32)((h >> 24) & 0xff);
            const u32 dd = (u32)((h >> 32) & 3), ad = (u32)((h >> 34) & 3), bd = (u32)((h >> 36) & 3);
            const u32 nd = (u32)((h >> 40) & 7), na = (u32)((h >> 43) & 7);
            const u32 pd = 8 * s + 1, pa = pd + nd, pb = pa + na;
            s += lng ? 2 : 1;
            Rows<R> a, v;
            load_operand<R>(a, A, ak, ad, win, pa, ro, gt, T);
            if (op == XLP_OP_COPY) v = a;
            else {
                Rows<R> b;
                load_operand<R>(b, A, bk, bd, win, pb, ro, gt, T);
#pragma unroll
                for (int r = 0; r < R; r++) {
                    e3& o = v.r[r];
                    const e3& x = a.r[r];
                    const e3& y = b.r[r];
                    o.c[0] = o.c[1] = o.c[2] = 0;
                    if (op == XLP_OP_ADD) {
                        o.c[0] = gl::addc(x.c[0], y.c[0]);
                        if (dd == 3) { o.c[1] = gl::addc(x.c[1], y.c[1]); o.c[2] = gl::addc(x.c[2], y.c[2]); }
                    } else if (op == XLP_OP_SUB) {
                        o.c[0] = gl::subc(x.c[0], y.c[0]);
                        if (dd == 3) { o.c[1] = gl::subc(x.c[1], y.c[1]); o.c[2] = gl::subc(x.c[2], y.c[2]); }
                    } else {
                        if (ad == 3 && bd == 3) o = gl::mul3(x, y);
                        else if (ad == 3) o = gl::mul3s(x, y.c[0]);
                        else if (bd == 3) o = gl::mul3s(y, x.c[0]);
                        else o.c[0] = gl::mulc(x.c[0], y.c[0]);
                    }
                }
            }
            store_operand<R>(A, dk, dd, win, pd, ro, gt, T, A.nrows, v);
        }
        win = nxt;
    }
}

// cols[c][j] = canon(base_c[((row0 + j) % modulus) * stride_c]),  j < rows (+ halo for flagged columns).
// 64 columns x 64 rows per workgroup through LDS: reads run along the table (neighbouring entries are
// neighbouring columns of one section -> contiguous bytes of a row), writes along the rows.
__global__ __launch_bounds__(256) void expr_gather_kernel(u64* cols, const ColDesc* tab, u32 ncols, u64 cap, u64 row0, u64 rows, u64 halo, u64 modulus)
{
    __shared__ u64 tile[64][65];
    const u32 c0 = blockIdx.x * 64, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u64 j0 = (u64)blockIdx.y * 64;
    const u32 c = c0 + lane;
    ColDesc d = {nullptr, 0, 0};
    if (c < ncols) d = tab[c];
    for (u32 r = wv; r < 64; r += 4) {
        const u64 j = j0 + r;
        u64 v = 0;
        if (c < ncols && j < rows + d.halo) {
            u64 ar = row0 + j;
            if (modulus) ar = wrap(ar, modulus);
            v = gl::canon(d.base[ar * d.stride]);
        }
        tile[r][lane] = v;
    }
    __syncthreads();
    for (u32 k = wv; k < 64; k += 4) {
        const u64 j = j0 + lane;
        if (c0 + k < ncols && j < rows + halo) cols[(u64)(c0 + k) * cap + j] = tile[lane][k];
    }
}

// base_c[(row0 + j) * stride_c] = cols[c][j] for the written columns (list `wl` of table indices), j < rows.
__global__ __launch_bounds__(256) void expr_scatter_kernel(const u64* cols, const ColDesc* tab, const u32* wl, u32 nw, u64 cap, u64 row0, u64 rows)
{
    __shared__ u64 tile[64][65];
    const u32 k0 = blockIdx.x * 64, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u64 j0 = (u64)blockIdx.y * 64;
    for (u32 k = wv; k < 64; k += 4) {
        const u64 j = j0 + lane;
        u64 v = 0;
        if (k0 + k < nw && j < rows) v = cols[(u64)wl[k0 + k] * cap + j];
        tile[lane][k] = v;
    }
    __syncthreads();
    if (k0 + lane < nw) {
        const ColDesc d = tab[wl[k0 + lane]];
        for (u32 r = wv; r < 64; r += 4) {
            const u64 j = j0 + r;
            if (j < rows) d.base[(row0 + j) * d.stride] = tile[r][lane];
        }
    }
}

// --------------------------------------------------------------------------------------- ahead-of-time code
// xlp_expr_codegen writes the compiled program as straight-line HIP: segments (segment_cuts: a cut every XLP_EXPR_SEG instructions, or the cut points
// that move the fewest words through HBM, XLP_EXPR_CUT), one __global__ per segment, one row per lane.  What the generator decides itself, each with the
// environment variable that sets it WHEN THE TEXT IS WRITTEN (tools/build_expr_plugins.py stamps a plugin with its text): where the loads are issued
// (place_loads: XLP_EXPR_HOIST_LOADS / _AHEAD), which products go through carry-free accumulators and how many are alive (XLP_EXPR_LAZY, _LAZY_ACCS,
// _LAZY_MIN_GAIN), which values are canonicalised (REPRESENTATIONS: XLP_EXPR_LAZY_CANON; the program flag XLP_EXPR_CANONICAL), and two measured-and-rejected
// variants kept as switches (XLP_EXPR_REUSE, XLP_EXPR_EARLY_STORES).  Inside a segment every temporary is an SSA value (the compiler
// allocates registers and hoists the coalesced column loads far ahead of their uses, which the interpreter's
// decode -> load -> wait -> compute -> store loop cannot); only the temporaries that are live across a segment
// boundary go through the HBM array tmp[word][lane].  The plugin is built with
//     hipcc -O3 --offload-arch=gfx950 -shared -fPIC -I<csrc> prog.hip -o prog.so
// and attached with xlp_expr_attach, after which xlp_dev_expr_run stages the columns exactly as for the
// interpreter and launches the plugin's kernels instead of expr_kernel.
static u64 fnv(const void* p, size_t n, u64 h)
{
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001b3ULL; }
    return h;
}
static u64 program_signature(const xlp_expr_program* P)
{
    u64 h = 0xcbf29ce484222325ULL;
    h = fnv(P->dcode.data(), P->dcode.size() * 8, h);
    for (const xlp_expr_column& c : P->cols) { h = fnv(&c.buf, 4, h); h = fnv(&c.off, 8, h); h = fnv(&c.stride, 8, h); }
    h = fnv(&P->tmp_words, 8, h);
    const u64 cm = (P->colmajor ? 1 : 0) | (P->canonical ? 2 : 0);
    if (cm) h = fnv(&cm, 8, h);
    return h;
}

struct DInstr { u32 op, k[3], dim[3]; u64 prm[3][4]; };

static std::vector<DInstr> decode_dcode(const xlp_expr_program* P)
{
    std::vector<DInstr> out;
    const std::vector<u64>& c = P->dcode;
    for (size_t s = 0; s * 8 < c.size();) {
        const u64 h = c[8 * s];
        DInstr in;
        in.op = (u32)(h & 0xff);
        const bool lng = (h >> 49) & 1;
        if (in.op == XD_NOP) { s++; continue; }
        in.k[0] = (u32)((h >> 8) & 0xff); in.k[1] = (u32)((h >> 16) & 0xff); in.k[2] = (u32)((h >> 24) & 0xff);
        in.dim[0] = (u32)((h >> 32) & 3); in.dim[1] = (u32)((h >> 34) & 3); in.dim[2] = (u32)((h >> 36) & 3);
        const u32 np[3] = {(u32)((h >> 40) & 7), (u32)((h >> 43) & 7), (u32)((h >> 46) & 7)};
        size_t q = 8 * s + 1;
        for (int o = 0; o < 3; o++)
            for (u32 j = 0; j < 4; j++) in.prm[o][j] = j < np[o] ? c[q++] : 0;
        out.push_back(in);
        s += lng ? 2 : 1;
    }
    return out;
}

// LOAD PIPELINING (round 5; round 4 did this as a pass over the generated text).  hipcc issues each global load of a generated segment right before
// its first use and waits for it almost at once (tools/isa_loads_in_flight.py on zkEVM step42ns as generated: 1.4 loads per wait, the youngest load
// a median of 10 instructions before its wait), so a wavefront exposes the whole memory latency ~60 times per segment and only the 2-5 wavefronts per
// SIMD hide it: the segments ran at 2 TB/s and 30 % vector issue, neither pipe full.  The generator therefore records every load whose address no
// store of the segment can alias -- temporaries of earlier segments (T: a segment stores its temporaries at its end), columns the segment does not
// write (C / PR: the ones it writes go through CW / PW and stay where they are), constant columns (KN / KE) -- and place_loads issues the RAW loads
// in batches of XLP_EXPR_HOIST_LOADS, each batch XLP_EXPR_HOIST_AHEAD (1) batches ahead of the first use of its first value and pinned by
// __builtin_amdgcn_sched_barrier(0) (no scheduling strategy of the compiler does this by itself: tried max-ilp, max-memory-clause); the
// canonicalisation stays at the place of use.  Same values, same statements.  0 = every load at its first use.
struct SegLoad { std::string name, expr; bool canon; };
struct LoadStats { size_t segments = 0, pipelined_segments = 0, loads = 0, pipelined_loads = 0, batches = 0, tmp_loads = 0, tmp_stores = 0, longest = 0; };

static std::string place_loads(const std::string& body, const std::vector<SegLoad>& loads, size_t batch, size_t ahead, LoadStats& st)
{
    st.segments++; st.loads += loads.size();
    for (const SegLoad& l : loads) st.tmp_loads += l.expr[0] == 'T';
    const bool pipe = batch > 0 && loads.size() >= 2 * batch;
    const size_t nb = pipe ? (loads.size() + batch - 1) / batch : 0;
    if (pipe) { st.pipelined_segments++; st.pipelined_loads += loads.size(); st.batches += nb; }
    std::vector<std::vector<size_t>> issue_at(pipe ? loads.size() : 0);        // mark k -> the batches whose raw loads go in front of it
    for (size_t b = 0; b < nb; b++) issue_at[b <= ahead ? 0 : (b - ahead) * batch].push_back(b);
    std::string out;
    out.reserve(body.size() + loads.size() * 48);
    for (size_t p = 0; p < body.size();) {
        size_t q = body.find('\n', p);
        if (q == std::string::npos) q = body.size();
        if (body[p] != '\x01') { out.append(body, p, q - p + 1); p = q + 1; continue; }
        const size_t k = (size_t)strtoull(body.c_str() + p + 1, nullptr, 10);
        p = q + 1;
        const SegLoad& l = loads[k];
        if (!pipe) {
            out += "    const u64 " + l.name + " = " + (l.canon ? "gl::canon(" + l.expr + ")" : l.expr) + ";\n";
            continue;
        }
        for (size_t b : issue_at[k]) {
            for (size_t j = b * batch; j < std::min(loads.size(), (b + 1) * batch); j++) out += "    const u64 " + loads[j].name + "_raw = " + loads[j].expr + ";\n";
            out += "    __builtin_amdgcn_sched_barrier(0);\n";
        }
        out += "    const u64 " + l.name + " = " + (l.canon ? "gl::canon(" + l.name + "_raw)" : l.name + "_raw") + ";\n";
    }
    return out;
}

// Where the segments of the generated code begin: {0, ..., ins.size()}.
//   XLP_EXPR_SEG = n (default 256): a cut every n instructions.
//   XLP_EXPR_CUT = L > 0: the cut points that move the fewest words per row through HBM, by dynamic programming over the program, with segments
//   of XLP_EXPR_CUT_MIN (64) .. L instructions.  What a segment [s, e) moves: one load per distinct column word it names, one load per temporary
//   word it reads whose value was defined before s, one store per temporary word it defines that is read at or after e -- the same accounting
//   as xlayer_prover_amd/expr_opt.py segment_traffic -- plus XLP_EXPR_CUT_SEGCOST (8) words per segment for what every kernel loads anyway
//   (challenges, the launch).  zkEVM step42ns: 9 450 words per row at a cut every 256 -> 8 751 (L = 256), 7 114 (L = 512), 6 438 (L = 768).
static std::vector<size_t> segment_cuts(const std::vector<DInstr>& ins, u64 W)
{
    const size_t n = ins.size();
    std::vector<size_t> cut;
    const size_t L = (size_t)std::max(0, env_int("XLP_EXPR_CUT", 0));
    if (!L || n <= L) {
        const size_t SEG = (size_t)std::max(1, env_int("XLP_EXPR_SEG", 256));
        for (size_t i = 0; i < n; i += SEG) cut.push_back(i);
        cut.push_back(n);
        return cut;
    }
    const size_t Lmin = std::min(L, (size_t)std::max(1, env_int("XLP_EXPR_CUT_MIN", 64)));
    const double segcost = (double)env_int("XLP_EXPR_CUT_SEGCOST", 8);
    // per instruction: the column words it reads (ids), the temporary words it reads (with the instruction that defined the value), the
    // temporary words it defines (with the last instruction that reads that value)
    std::map<std::array<u64, 6>, u32> colid;
    std::vector<std::vector<u32>> cols(n);
    std::vector<std::vector<std::pair<u64, long>>> tread(n), tdef(n);
    std::vector<long> lastdef(W, -1);
    for (size_t i = 0; i < n; i++) {
        const DInstr& in = ins[i];
        for (int o = 1; o < 3; o++) {
            const u32 k = in.k[o];
            if (k == XD_TMP) {
                for (u32 c = 0; c < in.dim[o]; c++) {
                    const u64 w = in.prm[o][0] + c;
                    const long d = lastdef[w];
                    tread[i].push_back({w, d});
                    if (d >= 0) tdef[d][w - ins[d].prm[0][0]].second = (long)i;
                }
            } else if (k == XD_COL || (k >= XLP_K_POLS && k <= XLP_K_CONST2NS_PRIME)) {
                for (u32 c = 0; c < in.dim[o]; c++) {
                    const std::array<u64, 6> key = {k, in.prm[o][0], in.prm[o][1], in.prm[o][2], in.prm[o][3], c};
                    cols[i].push_back(colid.emplace(key, (u32)colid.size()).first->second);
                }
            }
        }
        if (in.k[0] == XD_TMP)
            for (u32 c = 0; c < in.dim[0]; c++) { tdef[i].push_back({in.prm[0][0] + c, -1}); lastdef[in.prm[0][0] + c] = (long)i; }
    }
    std::vector<double> f(n + 1, 1e300);
    std::vector<size_t> prev(n + 1, 0);
    std::vector<size_t> col_seen(colid.size(), 0), tmp_seen(W, 0);     // stamps: s + 1
    std::vector<u32> pend(n + 1, 0);                                   // stores that stop being needed once the segment reaches past index k
    f[0] = 0;
    for (size_t s = 0; s < n; s++) {
        if (f[s] >= 1e299) continue;
        size_t words = 0, stores = 0;
        const size_t emax = std::min(n, s + L);
        for (size_t e = s + 1; e <= emax; e++) {
            const size_t i = e - 1;
            for (u32 c : cols[i]) if (col_seen[c] != s + 1) { col_seen[c] = s + 1; words++; }
            for (const auto& r : tread[i]) if (r.second < (long)s && tmp_seen[r.first] != s + 1) { tmp_seen[r.first] = s + 1; words++; }
            for (const auto& d : tdef[i]) if (d.second >= (long)e) { stores++; pend[d.second]++; }
            stores -= pend[i]; pend[i] = 0;                            // values last read by instruction i are not live out of [s, e)
            if (e - s < Lmin && e != n) continue;
            const double c = f[s] + (double)(words + stores) + segcost;
            if (c < f[e]) { f[e] = c; prev[e] = s; }
        }
        for (size_t e = s + 1; e <= emax; e++) for (const auto& d : tdef[e - 1]) if (d.second >= 0) pend[d.second] = 0;
    }
    for (size_t e = n; e > 0; e = prev[e]) cut.push_back(e);
    cut.push_back(0);
    std::reverse(cut.begin(), cut.end());
    return cut;
}

int expr_codegen(const xlp_expr_program* P, const char* path)
{
    if (!P || !path) return xlp_fail("expr codegen: null argument");
    const std::vector<DInstr> ins = decode_dcode(P);
    const u64 W = P->tmp_words;
    // segment sg = instructions [cut[sg], cut[sg + 1])
    const std::vector<size_t> cut = segment_cuts(ins, W);
    const size_t nseg = cut.size() - 1;
    // words live at the start of each segment (backward dataflow over the temporaries)
    std::vector<std::vector<bool>> live_in(nseg + 1, std::vector<bool>(W, false));
    {
        std::vector<bool> live(W, false);
        size_t sg = nseg;
        for (size_t i = ins.size(); i-- > 0;) {
            const DInstr& in = ins[i];
            if (in.k[0] == XD_TMP) for (u32 c = 0; c < in.dim[0]; c++) live[in.prm[0][0] + c] = false;
            for (int o = 1; o < 3; o++) if (in.k[o] == XD_TMP) for (u32 c = 0; c < in.dim[o]; c++) live[in.prm[o][0] + c] = true;
            if (sg > 0 && i == cut[sg - 1]) live_in[--sg] = live;
        }
    }
    // dies[i][o]: the temporary operand o of instruction i is not read again (its words are overwritten first, or never touched)
    std::vector<std::array<bool, 3>> dies(ins.size());
    {
        std::vector<bool> live(W, false);
        for (size_t i = ins.size(); i-- > 0;) {
            const DInstr& in = ins[i];
            dies[i] = {false, false, false};
            for (int o = 1; o < 3; o++) {
                if (in.k[o] != XD_TMP) continue;
                bool d = true;
                for (u32 c = 0; c < in.dim[o]; c++) {
                    const u64 w = in.prm[o][0] + c;
                    const bool overwritten = in.k[0] == XD_TMP && w >= in.prm[0][0] && w < in.prm[0][0] + in.dim[0];
                    if (live[w] && !overwritten) d = false;
                }
                dies[i][o] = d;
            }
            if (in.k[0] == XD_TMP) for (u32 c = 0; c < in.dim[0]; c++) live[in.prm[0][0] + c] = false;
            for (int o = 1; o < 3; o++) if (in.k[o] == XD_TMP) for (u32 c = 0; c < in.dim[o]; c++) live[in.prm[o][0] + c] = true;
        }
    }
    // LAZY ACCUMULATION (round 3).  Most of a step program is "acc += challenge power x value" (the constraint sum in dot-product form:
    // 6 774 of the 15 891 multiplications of the zkEVM step42ns, 4 383 of 5 172 in step3prev).  The coefficient is wave-uniform, so the
    // product needs no reduction of its own: a MUL of a challenge / evaluation (dimension 3) by a base-field value is kept as a symbol
    // ("@P"), an ADD / SUB of such a symbol becomes gl::lazy3::mac on an accumulator ("@A", 18 vector instructions instead of ~63), and
    // the three canonical words are only computed (fold) where something else reads them or at the end of the segment.  Results are the
    // same field elements.  XLP_EXPR_LAZY=0 generates the plain form; XLP_EXPR_LAZY_ACCS bounds the accumulators alive at once
    // (42 registers each); XLP_EXPR_LAZY_MIN_GAIN: what a NEW accumulator must save to be created (below).
    const bool lazy_canon = env_int("XLP_EXPR_LAZY_CANON", 1) != 0;
    const bool lazy_on = env_int("XLP_EXPR_LAZY", 1) != 0;
    // (ONE accumulator alive at a time since round 5 -- 3 in rounds 3-4: an accumulator is 42 registers and the segments are bound by what their registers allow in
    // flight: zkEVM step42ns 0.272 -> 0.241 s, 0.249 with two, slower with four or more; without the accumulators at all 0.306 s: profiles/r5/step_program_variant_sweep.txt)
    const size_t lazy_max = (size_t)env_int("XLP_EXPR_LAZY_ACCS", 1);
    const bool early_stores = env_int("XLP_EXPR_EARLY_STORES", 0) != 0;     // (measured, not adopted: see where it is used)
    const size_t reuse_dist = (size_t)std::max(0, env_int("XLP_EXPR_REUSE", 0));
    const int lazy_min_gain = env_int("XLP_EXPR_LAZY_MIN_GAIN", 190);        // vector instructions a new accumulator must save to be created (0: always, as in rounds 3-4)
    const std::string prologue =
        "// generated by xlp_expr_codegen -- do not edit\n#include <hip/hip_runtime.h>\n#include \"gl_field.hpp\"\n#include \"xlp_expr_jit.h\"\n"
        "typedef unsigned long long u64_;\nusing gl::u64;\n"
        "static __device__ __forceinline__ u64 wrapr(u64 r, u64 n) { return (n & (n - 1)) == 0 ? (r & (n - 1)) : r % n; }\n"
        "static __device__ __forceinline__ u64 wrapc(u64 j, u64 base, u64 n) { return j - base >= n ? j - n : j; }\n";
    const bool CM = P->colmajor;
    const size_t hoist = (size_t)std::max(0, env_int("XLP_EXPR_HOIST_LOADS", 0)), hoist_ahead = (size_t)std::max(1, env_int("XLP_EXPR_HOIST_AHEAD", 1));
    LoadStats stats;
    std::vector<std::string> segsrc(nseg);
    char buf[256];
    for (size_t sg = 0; sg < nseg; sg++) {
        std::ostringstream os;
        // XLP_EXPR_MIN_WAVES (experiment; read when the text is generated): a floor on the waves per SIMD = a ceiling on the registers of a segment
        const int minWaves = env_int("XLP_EXPR_MIN_WAVES", 0);
        os << "extern \"C\" __global__ __launch_bounds__(256" << (minWaves > 0 ? ", " + std::to_string(minWaves) : std::string()) << ") void seg_" << sg << "(XlpJitArgs A)\n{\n"
              "    const u64 gt = (u64)blockIdx.x * 256 + threadIdx.x;\n    if (gt >= A.nrows) return;\n    const u64 row = A.row0 + gt;\n"
              "    u64* __restrict__ T = A.tmp + gt; const u64 TS = A.tstride;\n"
              "    const u64* __restrict__ C = A.cols + gt; u64* __restrict__ CW = A.cols + gt; const u64 CAP = A.cap;\n"
              "    (void)row; (void)T; (void)TS; (void)C; (void)CW; (void)CAP;\n";
        if (CM)
            os << "    const u64* __restrict__ PR = A.P.pols; u64* __restrict__ PW = A.P.pols; const u64* __restrict__ KN = A.P.const_pols; const u64* __restrict__ KE = A.P.const_pols_2ns;\n"
                  "    const u64 KNR = A.crows, KER = A.crows2, drow = row - A.dshift, nat3 = 3 * ((row - A.cbase) * A.P.zi_size + A.coset);\n"
                  "    (void)PR; (void)PW; (void)KN; (void)KE; (void)KNR; (void)KER; (void)drow; (void)nat3;\n";
        const size_t i0 = cut[sg], i1 = cut[sg + 1];
        std::set<u64> wcol;                              // columns this segment writes: all their accesses go through CW (restrict-clean)
        for (size_t i = i0; i < i1; i++) if (ins[i].k[0] == XD_COL) for (u32 c = 0; c < ins[i].dim[0]; c++) wcol.insert(ins[i].prm[0][0] + c);
        // column-major form: a column is named by (buffer, first word | constant column); the ones this segment writes go through PW
        typedef std::tuple<u32, u64, u64> MKey;          // (buffer 0 pols / 1 const / 2 const2ns, column, next)
        std::set<u64> wcolm;
        std::map<MKey, std::string> mval;
        if (CM)
            for (size_t i = i0; i < i1; i++)
                if (ins[i].k[0] == XLP_K_POLS || ins[i].k[0] == XLP_K_POLS_PRIME)
                    for (u32 c = 0; c < ins[i].dim[0]; c++) wcolm.insert(ins[i].prm[0][0] + c * ins[i].prm[0][ins[i].k[0] == XLP_K_POLS ? 1 : 3]);
        std::map<u64, size_t> last_def;                  // tmp word -> the last instruction of this segment that defines it
        for (size_t i = i0; i < i1; i++) if (ins[i].k[0] == XD_TMP) for (u32 c = 0; c < ins[i].dim[0]; c++) last_def[ins[i].prm[0][0] + c] = i;
        std::set<u64> stored;                            // live-out words already stored (early stores)
        std::map<u64, std::string> tval;                 // tmp word -> SSA name
        std::map<u64, bool> tdef;                        // defined (not just loaded) in this segment
        std::map<std::pair<u64, u64>, std::string> cval; // (column, next) -> SSA name
        std::map<std::string, std::string> dval;         // direct scalar expressions -> SSA name
        u64 nv = 0;
        auto fresh = [&]() { snprintf(buf, sizeof buf, "v%llu", (unsigned long long)nv++); return std::string(buf); };
        // REPRESENTATIONS (round 5).  A name is either a CANONICAL word (< p: everything loaded, every constant, every addc / subc / mul3 / fold result) or
        // an "ANY" word (< 2^64, congruent: what gl::mul / gl::add / gl::sub return, csrc/gl_field.hpp).  A product is not canonicalised (4 vector
        // instructions + two hazard wait states) when what reads it does not need that: another multiplication, the carry-free accumulators, an addition
        // or subtraction whose OTHER operand is canonical (any + canon, any - canon stay "any" at the cost of addc / subc).  cn() makes the canonical
        // word where one is required -- stores, mul3 / mac3 operands, the subtrahend, one side of any + any -- once per value.  XLP_EXPR_LAZY_CANON=0: all canonical.
        std::set<std::string> anyv;
        std::map<std::string, std::string> canon_of;
        auto cn = [&](const std::string& n) -> std::string {
            if (!anyv.count(n)) return n;
            auto it = canon_of.find(n);
            if (it != canon_of.end()) return it->second;
            const std::string c = fresh();
            os << "    const u64 " << c << " = gl::canon(" << n << ");\n";
            canon_of[n] = c;
            return c;
        };
        auto anyname = [&](const std::string& n) { if (lazy_canon) anyv.insert(n); return n; };
        const char* const MUL = lazy_canon ? "gl::mul(" : "gl::mulc(";
        // a load of a value no store of this segment can alias (a temporary of an earlier segment, a staged / section column the segment does
        // not write, a constant column): its statement is placed when the segment is assembled (place_loads), here it leaves a mark
        std::vector<SegLoad> loads;
        auto load = [&](const std::string& expr, bool canon) {
            const std::string n = fresh();
            os << '\x01' << loads.size() << "\n";
            loads.push_back({n, expr, canon});
            return n;
        };
        auto direct = [&](const std::string& expr) {
            auto it = dval.find(expr);
            if (it != dval.end()) return it->second;
            const std::string n = fresh();
            os << "    const u64 " << n << " = gl::canon(" << expr << ");\n";
            dval[expr] = n;
            return n;
        };
        auto num = [&](u64 v) { snprintf(buf, sizeof buf, "0x%llxULL", (unsigned long long)v); return std::string(buf); };
        // symbols of the lazy accumulation: "@P<id>.<c>" a product not multiplied yet, "@A<id>:<version>.<c>" a state of an accumulator
        struct Prod { std::string k[3], v, x[3], mat[3]; bool ext = false, done = false; };   // ext: the value is an Fp^3 element x (canonical words)
        struct Acc { std::string var; u32 ver = 0, count = 0; bool retired = false; size_t last = 0; std::map<u32, std::array<std::string, 3>> folded; };
        std::vector<Prod> prods;
        std::vector<Acc> accs;
        std::map<std::string, std::string> rawval;       // uniform words as loaded (no canon): coefficients of the products
        auto raw = [&](const std::string& expr) {
            auto it = rawval.find(expr);
            if (it != rawval.end()) return it->second;
            const std::string n = fresh();
            os << "    const u64 " << n << " = " << expr << ";\n";
            rawval[expr] = n;
            return n;
        };
        auto sym = [&](char kind, u32 id, u32 ver, u32 c) {
            if (kind == 'P') snprintf(buf, sizeof buf, "@P%u.%u", id, c); else snprintf(buf, sizeof buf, "@A%u:%u.%u", id, ver, c);
            return std::string(buf);
        };
        auto parse = [](const std::string& t, char& kind, u32& id, u32& ver, u32& c) {
            if (t.size() < 2 || t[0] != '@') return false;
            kind = t[1]; ver = 0;
            if (kind == 'P') return sscanf(t.c_str() + 2, "%u.%u", &id, &c) == 2;
            return sscanf(t.c_str() + 2, "%u:%u.%u", &id, &ver, &c) == 3;
        };
        // the SSA name of a value: multiplies a product / folds an accumulator state on first use
        auto real = [&](const std::string& t) -> std::string {
            char kind; u32 id, ver, c;
            if (!parse(t, kind, id, ver, c)) return t;
            if (kind == 'P') {
                Prod& p = prods[id];
                if (!p.done && p.ext) {
                    const std::string e = fresh();
                    os << "    const gl::e3 " << e << " = gl::mul3(gl::e3{{gl::canon(" << p.k[0] << "), gl::canon(" << p.k[1] << "), gl::canon(" << p.k[2] << ")}}, gl::e3{{"
                       << p.x[0] << ", " << p.x[1] << ", " << p.x[2] << "}});\n";
                    for (u32 q = 0; q < 3; q++) p.mat[q] = e + ".c[" + std::to_string(q) + "]";
                    p.done = true;
                }
                if (!p.done) {
                    for (u32 q = 0; q < 3; q++) { p.mat[q] = anyname(fresh()); os << "    const u64 " << p.mat[q] << " = " << MUL << p.k[q] << ", " << p.v << ");\n"; }
                    p.done = true;
                }
                return p.mat[c];
            }
            Acc& a = accs[id];
            auto it = a.folded.find(ver);
            if (it == a.folded.end()) {                  // only the current state can be unfolded: older ones are folded before the accumulator moves on
                if (ver != a.ver) os << "#error xlp_expr_codegen: stale accumulator state\n";
                std::array<std::string, 3> f = {fresh(), fresh(), fresh()};
                os << "    u64 " << f[0] << ", " << f[1] << ", " << f[2] << "; " << a.var << ".fold(" << f[0] << ", " << f[1] << ", " << f[2] << ");\n";
                it = a.folded.emplace(ver, f).first;
            }
            return it->second[c];
        };
        // value of component c of a source operand
        // XLP_EXPR_REUSE = D > 0 (round 5 experiment): a LOADED word (a column the segment does not write, a temporary of an earlier segment) whose last use lies more
        // than D instructions back is loaded again instead of being held in its register: bytes (which have slack: the segments are bound by vector issue and
        // registers) for registers.  0: a word is loaded once per segment.
        size_t cur_i = i0;
        std::map<u64, size_t> tlast;
        std::map<MKey, size_t> mlast;
        auto src = [&](const DInstr& in, int o, u32 c) -> std::string {
            const u64* p = in.prm[o];
            const u32 k = in.k[o];
            if (c >= in.dim[o]) return "0ULL";
            switch (k) {
            case XD_TMP: {
                const u64 w = p[0] + c;
                auto it = tval.find(w);
                if (it != tval.end() && !(reuse_dist && !tdef[w] && cur_i - tlast[w] > reuse_dist)) { tlast[w] = cur_i; return it->second; }
                const std::string n = load("T[" + std::to_string(w) + "ULL * TS]", false);
                tval[w] = n; tdef[w] = false; tlast[w] = cur_i;
                return n;
            }
            case XD_COL: {
                const std::pair<u64, u64> key(p[0] + c, p[1]);
                auto it = cval.find(key);
                if (it != cval.end()) return it->second;
                std::string n;
                if (wcol.count(key.first)) { n = fresh(); os << "    const u64 " << n << " = CW[" << key.first << "ULL * CAP + " << key.second << "ULL];\n"; }
                else n = load("C[" + std::to_string(key.first) + "ULL * CAP + " + std::to_string(key.second) + "ULL]", false);
                cval[key] = n;
                return n;
            }
            case XLP_K_NUMBER: return num(p[0]);
            case XLP_K_POLS: case XLP_K_POLS_PRIME: case XLP_K_CONST: case XLP_K_CONST_PRIME: case XLP_K_CONST2NS: case XLP_K_CONST2NS_PRIME:
                if (CM) {
                    const bool pol = k == XLP_K_POLS || k == XLP_K_POLS_PRIME, pr = k == XLP_K_POLS_PRIME || k == XLP_K_CONST_PRIME || k == XLP_K_CONST2NS_PRIME;
                    const u32 buf = pol ? 0 : (k == XLP_K_CONST || k == XLP_K_CONST_PRIME) ? 1 : 2;
                    const u64 col = pol ? p[0] + c * (pr ? p[3] : p[1]) : p[0];
                    const MKey key(buf, col, pr ? p[1] : 0);
                    auto it = mval.find(key);
                    const bool held_long = reuse_dist && !(pol && wcolm.count(col)) && it != mval.end() && cur_i - mlast[key] > reuse_dist;
                    if (it != mval.end() && !held_long) { mlast[key] = cur_i; return it->second; }
                    mlast[key] = cur_i;
                    const std::string idx = pr ? "wrapc(row + " + num(p[1]) + ", A.cbase, " + num(p[2]) + ")" : std::string("row");
                    std::string n;
                    const bool cnl = !P->canonical;            // (XLP_EXPR_CANONICAL: the sections hold canonical words)
                    if (pol && wcolm.count(col)) { n = fresh(); os << "    const u64 " << n << " = " << (cnl ? "gl::canon(" : "(") << "PW[" << num(col) << " + " << idx << "]);\n"; }
                    else if (pol) n = load("PR[" + num(col) + " + " + idx + "]", cnl);
                    else n = load(std::string(buf == 1 ? "KN[" : "KE[") + num(col) + " * " + (buf == 1 ? "KNR" : "KER") + " + " + idx + "]", cnl);
                    mval[key] = n;
                    return n;
                }
                break;
            default: break;
            }
            switch (k) {
            case XLP_K_POLS: return direct("A.P.pols[" + num(p[0] + c) + " + row * " + num(p[1]) + "]");
            case XLP_K_POLS_PRIME: return direct("A.P.pols[" + num(p[0] + c) + " + wrapr(row + " + num(p[1]) + ", " + num(p[2]) + ") * " + num(p[3]) + "]");
            case XLP_K_CONST: return direct("A.P.const_pols[" + num(p[0]) + " + row * A.P.n_const_pols]");
            case XLP_K_CONST_PRIME: return direct("A.P.const_pols[" + num(p[0]) + " + wrapr(row + " + num(p[1]) + ", " + num(p[2]) + ") * A.P.n_const_pols]");
            case XLP_K_CONST2NS: return direct("A.P.const_pols_2ns[" + num(p[0]) + " + row * A.P.n_const_pols_2ns]");
            case XLP_K_CONST2NS_PRIME: return direct("A.P.const_pols_2ns[" + num(p[0]) + " + wrapr(row + " + num(p[1]) + ", " + num(p[2]) + ") * A.P.n_const_pols_2ns]");
            case XLP_K_CHALLENGE: return direct("A.P.challenges[" + num(3 * p[0] + c) + "]");
            case XLP_K_EVAL: return direct("A.P.evals[" + num(3 * p[0] + c) + "]");
            case XLP_K_PUBLIC: return direct("A.P.public_inputs[" + num(p[0]) + "]");
            case XLP_K_X_N: return direct(CM ? "A.P.x_n[drow]" : "A.P.x_n[row]");
            case XLP_K_X_2NS: return direct(CM ? "A.P.x_2ns[drow]" : "A.P.x_2ns[row]");
            case XLP_K_ZI: return direct(CM ? "A.P.zi[A.coset]" : "A.P.zi[wrapr(row, A.P.zi_size)]");
            case XLP_K_XDIVXSUBXI: return direct(std::string("A.P.x_div_x_sub_xi[3 * ") + (CM ? "drow" : "row") + " + " + num(c) + "]");
            case XLP_K_XDIVXSUBWXI: return direct(std::string("A.P.x_div_x_sub_wxi[3 * ") + (CM ? "drow" : "row") + " + " + num(c) + "]");
            case XLP_K_Q_2NS: { const std::string n = fresh(); os << "    const u64 " << n << " = gl::canon(A.P.q_2ns[" << (CM ? "nat3" : "3 * row") << " + " << c << "]);\n"; return n; }
            case XLP_K_F_2NS: { const std::string n = fresh(); os << "    const u64 " << n << " = gl::canon(A.P.f_2ns[" << (CM ? "nat3" : "3 * row") << " + " << c << "]);\n"; return n; }
            default: return "0ULL";
            }
        };
        for (size_t i = i0; i < i1; i++) {
            cur_i = i;
            const DInstr& in = ins[i];
            const u32 dd = in.dim[0], ad = in.dim[1], bd = in.dim[2];
            std::string a[3], b[3], r[3];
            auto uniform3 = [&](int o) { return in.dim[o] == 3 && (in.k[o] == XLP_K_CHALLENGE || in.k[o] == XLP_K_EVAL); };
            const int ko = (lazy_on && in.op == XLP_OP_MUL) ? (uniform3(1) ? 1 : uniform3(2) ? 2 : 0) : 0;
            if (ko) {                                    // (uniform Fp^3) x (value of the row): a symbol, multiplied where it is read
                Prod p;
                const char* arr = in.k[ko] == XLP_K_CHALLENGE ? "A.P.challenges[" : "A.P.evals[";
                for (u32 c = 0; c < 3; c++) p.k[c] = raw(arr + num(3 * in.prm[ko][0] + c) + "]");
                const int vo = ko == 1 ? 2 : 1;
                p.ext = in.dim[vo] == 3;
                if (p.ext) for (u32 c = 0; c < 3; c++) p.x[c] = cn(real(src(in, vo, c)));     // mul3 / mac3 take canonical words
                else p.v = real(src(in, vo, 0));                                              // (any word: mac and mul take it as it is)
                prods.push_back(p);
                for (u32 c = 0; c < 3; c++) r[c] = sym('P', (u32)prods.size() - 1, 0, c);
            } else {
            for (u32 c = 0; c < 3; c++) a[c] = src(in, 1, c);
            if (in.op != XLP_OP_COPY) for (u32 c = 0; c < 3; c++) b[c] = src(in, 2, c);
            // an ADD / SUB that takes a product symbol goes to an accumulator
            bool accumulated = false;
            if (lazy_on && dd == 3 && (in.op == XLP_OP_ADD || in.op == XLP_OP_SUB)) {
                auto prod_of = [&](const std::string* x, u32 dim) -> int {
                    char kind; u32 id, ver, c;
                    if (dim != 3 || !parse(x[0], kind, id, ver, c) || kind != 'P' || c != 0) return -1;
                    return (x[1] == sym('P', id, 0, 1) && x[2] == sym('P', id, 0, 2)) ? (int)id : -1;
                };
                auto acc_of = [&](const std::string* x, u32 dim) -> int {    // the CURRENT state of an accumulator still in use
                    char kind; u32 id, ver, c;
                    if (dim != 3 || !parse(x[0], kind, id, ver, c) || kind != 'A' || c != 0) return -1;
                    if (x[1] != sym('A', id, ver, 1) || x[2] != sym('A', id, ver, 2)) return -1;
                    return (!accs[id].retired && accs[id].ver == ver) ? (int)id : -1;
                };
                int pa = in.op == XLP_OP_ADD ? prod_of(a, ad) : -1, pb = prod_of(b, bd);   // a - k v = a + (-k) v; (k v) - b stays plain
                // IS A NEW ACCUMULATOR WORTH IT (round 5)?  Folding one costs ~190 vector instructions (three components x two reduce128 + the 128-bit
                // recombination); a product that goes through it saves ~51 (base-field value: 18 multiply-adds instead of three mulc + three addc) or ~126
                // (Fp^3 value: mac3 instead of mul3 + add3).  An accumulator that takes a single product before its sum is read -- 193 of the 434 of zkEVM step42ns:
                // the steps of Horner chains the dot-product rewrite could not open, acc = c + k * (previous acc) -- costs more than the plain form.  One is only
                // created when what is added now pays for the fold, or when the NEXT reader of the sum adds another product to it (the chain goes on).
                if ((pa >= 0 || pb >= 0) && lazy_min_gain > 0) {
                    const std::string* X0 = (pa >= 0 && pb >= 0) ? nullptr : (pb >= 0 ? a : b);
                    const bool joins = X0 && acc_of(X0, X0 == a ? ad : bd) >= 0;
                    if (!joins) {
                        int gain = 0;
                        if (pa >= 0) gain += prods[pa].ext ? 126 : 51;
                        if (pb >= 0) gain += prods[pb].ext ? 126 : 51;
                        bool goes_on = false;
                        if (in.k[0] == XD_TMP) {
                            const u64 w0 = in.prm[0][0];
                            for (size_t j = i + 1; j < i1 && !goes_on; j++) {
                                const DInstr& nx = ins[j];
                                int rd = 0;
                                for (int o = 1; o < 3; o++) if (nx.k[o] == XD_TMP && nx.prm[o][0] <= w0 && w0 < nx.prm[o][0] + nx.dim[o]) rd = o;
                                if (rd) {                                       // the first reader of the sum: another accumulation step?
                                    const int oo = 3 - rd;
                                    if ((nx.op == XLP_OP_ADD || (nx.op == XLP_OP_SUB && rd == 1)) && nx.dim[0] == 3 && nx.k[oo] == XD_TMP && nx.dim[oo] == 3) {
                                        for (size_t q = j; q-- > i0;) {           // who wrote the other operand last: a product by a challenge / evaluation?
                                            const DInstr& wr = ins[q];
                                            if (wr.k[0] != XD_TMP || !(wr.prm[0][0] <= nx.prm[oo][0] && nx.prm[oo][0] < wr.prm[0][0] + wr.dim[0])) continue;
                                            auto u3 = [&](int o) { return wr.dim[o] == 3 && (wr.k[o] == XLP_K_CHALLENGE || wr.k[o] == XLP_K_EVAL); };
                                            goes_on = wr.op == XLP_OP_MUL && (u3(1) || u3(2));
                                            break;
                                        }
                                    }
                                    break;
                                }
                                if (nx.k[0] == XD_TMP && nx.prm[0][0] <= w0 && w0 < nx.prm[0][0] + nx.dim[0]) break;     // overwritten unread
                            }
                        }
                        if (!goes_on && gain < lazy_min_gain) pa = pb = -1;      // the plain form: the products are multiplied out below
                    }
                }
                if (pa >= 0 || pb >= 0) {
                    accumulated = true;
                    const std::string* X = (pa >= 0 && pb >= 0) ? nullptr : (pb >= 0 ? a : b);
                    const u32 xd = X == a ? ad : bd;
                    const int xo = X == a ? 1 : 2;
                    const size_t nprod = (pa >= 0 ? (prods[pa].ext ? 3 : 1) : 0) + (pb >= 0 ? (prods[pb].ext ? 3 : 1) : 0);   // products per accumulator
                    int aid = X ? acc_of(X, xd) : -1;
                    if (aid >= 0 && accs[aid].count + nprod > 1000) aid = -1;   // an accumulator takes 1 024 products (gl::lazy3)
                    const bool created = aid < 0;
                    if (aid >= 0) {
                        // somebody else still holds this state (a temporary that is neither this operand on its last read nor the destination):
                        // fold it now, the holders then resolve to the folded words, and the accumulator moves on
                        Acc& A = accs[aid];
                        bool others = false;
                        for (const auto& kv : tval) {
                            char kind; u32 id, ver, c;
                            if (!parse(kv.second, kind, id, ver, c) || kind != 'A' || (int)id != aid || ver != A.ver) continue;
                            const u64 w = kv.first;
                            const bool is_dst = in.k[0] == XD_TMP && w >= in.prm[0][0] && w < in.prm[0][0] + dd;
                            const bool is_x = in.k[xo] == XD_TMP && w >= in.prm[xo][0] && w < in.prm[xo][0] + xd && dies[i][xo];
                            if (!is_dst && !is_x) others = true;
                        }
                        if (others) (void)real(X[0]);
                    } else {
                        std::string base[3] = {"0ULL", "0ULL", "0ULL"};
                        if (X) for (u32 c = 0; c < xd; c++) base[c] = cn(real(X[c]));
                        size_t alive = 0;
                        for (const Acc& A : accs) alive += !A.retired;
                        while (alive >= lazy_max && alive) {                     // retire the accumulator touched longest ago
                            size_t victim = accs.size();
                            for (size_t q = 0; q < accs.size(); q++) if (!accs[q].retired && (victim == accs.size() || accs[q].last < accs[victim].last)) victim = q;
                            (void)real(sym('A', (u32)victim, accs[victim].ver, 0));
                            accs[victim].retired = true;
                            alive--;
                        }
                        Acc A;
                        A.var = "acc" + std::to_string(sg) + "_" + std::to_string(accs.size());
                        os << "    gl::lazy3 " << A.var << "; " << A.var << ".init(" << base[0] << ", " << base[1] << ", " << base[2] << ");\n";
                        accs.push_back(A);
                        aid = (int)accs.size() - 1;
                    }
                    Acc& A = accs[aid];
                    auto mac = [&](int pid, bool negate) {
                        const Prod& p = prods[pid];
                        os << "    " << A.var << (p.ext ? ".mac3(" : ".mac(");
                        for (u32 c = 0; c < 3; c++) os << (negate ? "gl::neg_any(" + p.k[c] + ")" : p.k[c]) << ", ";
                        if (p.ext) os << p.x[0] << ", " << p.x[1] << ", " << p.x[2] << ");\n";
                        else os << p.v << ");\n";
                    };
                    if (pa >= 0) mac(pa, false);
                    if (pb >= 0) mac(pb, in.op == XLP_OP_SUB);
                    if (!created) A.ver++;                                       // the state handed out before this instruction is history
                    A.count += (u32)nprod;
                    A.last = i;
                    for (u32 c = 0; c < 3; c++) r[c] = sym('A', (u32)aid, A.ver, c);
                }
            }
            if (accumulated) {}
            else if (in.op == XLP_OP_COPY) { for (u32 c = 0; c < dd; c++) r[c] = a[c]; }
            else if (in.op == XLP_OP_ADD || in.op == XLP_OP_SUB) {
                for (u32 c = 0; c < 3; c++) { a[c] = real(a[c]); b[c] = real(b[c]); }
                for (u32 c = 0; c < dd; c++) {
                    const bool ha = c < ad, hb = c < bd;
                    if (ha && hb) {
                        // any + canon / canon + any / any - canon stay "any"; any + any and canon - any canonicalise one side first
                        std::string x = a[c], y = b[c];
                        if (in.op == XLP_OP_ADD) {
                            if (anyv.count(x) && anyv.count(y)) y = cn(y);
                            if (anyv.count(y)) std::swap(x, y);                          // gl::add(any, canon)
                        } else y = cn(y);
                        const bool res_any = anyv.count(x) != 0;
                        const char* f = in.op == XLP_OP_ADD ? (res_any ? "gl::add" : "gl::addc") : (res_any ? "gl::sub" : "gl::subc");
                        r[c] = fresh(); os << "    const u64 " << r[c] << " = " << f << "(" << x << ", " << y << ");\n";
                        if (res_any) anyv.insert(r[c]);
                    }
                    else if (ha) r[c] = a[c];
                    else if (in.op == XLP_OP_ADD) r[c] = b[c];
                    else { const std::string y = cn(b[c]); r[c] = fresh(); os << "    const u64 " << r[c] << " = gl::neg(" << y << ");\n"; }
                }
            } else {
                for (u32 c = 0; c < 3; c++) { a[c] = real(a[c]); b[c] = real(b[c]); }
                if (ad == 3 && bd == 3) {
                    for (u32 c = 0; c < 3; c++) { a[c] = cn(a[c]); b[c] = cn(b[c]); }
                    const std::string e = fresh();
                    os << "    const gl::e3 " << e << " = gl::mul3(gl::e3{{" << a[0] << ", " << a[1] << ", " << a[2] << "}}, gl::e3{{" << b[0] << ", " << b[1] << ", " << b[2] << "}});\n";
                    for (u32 c = 0; c < 3; c++) r[c] = e + ".c[" + std::to_string(c) + "]";
                } else if (ad == 3 || bd == 3) {
                    const std::string* x = ad == 3 ? a : b;
                    const std::string& s = ad == 3 ? b[0] : a[0];
                    for (u32 c = 0; c < 3; c++) { r[c] = anyname(fresh()); os << "    const u64 " << r[c] << " = " << MUL << x[c] << ", " << s << ");\n"; }
                } else { r[0] = anyname(fresh()); os << "    const u64 " << r[0] << " = " << MUL << a[0] << ", " << b[0] << ");\n"; }
            }
            }
            // destination (a temporary may hold a symbol; anything stored is a word)
            if (in.k[0] != XD_TMP) for (u32 c = 0; c < dd; c++) r[c] = cn(real(r[c]));       // what is stored is a canonical word
            const u64* p = in.prm[0];
            switch (in.k[0]) {
            case XD_TMP:
                for (u32 c = 0; c < dd; c++) { tval[p[0] + c] = r[c]; tdef[p[0] + c] = true; }
                // EARLY STORES (round 5): a temporary that is live out of the segment used to be stored at the segment's END -- its value sat in a register pair until
                // then, and the compiler cannot move the store up by itself: the fences of the load pipelining (sched_barrier(0)) hold every instruction in place.
                // It is stored HERE when this is its last definition in the segment (the words of a chain that is still accumulating are not: every step redefines them).
                // MEASURED (profiles/r5/step_program_variant_sweep.txt, thirteenth call): step42ns 0.230 -> 0.241 s, the others unchanged -- the register allocation moved the
                // wrong way (four kernels above 256 registers instead of one, one with scratch).  Off by default; XLP_EXPR_EARLY_STORES=1 generates it.
                if (early_stores)
                    for (u32 c = 0; c < dd; c++) {
                        const u64 w = p[0] + c;
                        if (!live_in[sg + 1][w] || last_def[w] != i || stored.count(w)) continue;
                        char kind; u32 id, ver, cc;
                        if (parse(r[c], kind, id, ver, cc)) continue;                  // (a product / accumulator symbol: materialised where it is read or at the end)
                        const std::string v = cn(r[c]);
                        os << "    T[" << w << "ULL * TS] = " << v << ";\n";
                        stats.tmp_stores++; stored.insert(w);
                    }
                break;
            case XD_COL:
                for (u32 c = 0; c < dd; c++) {
                    os << "    CW[" << (p[0] + c) << "ULL * CAP] = " << r[c] << ";\n";
                    for (auto it = cval.begin(); it != cval.end();) { if (it->first.first == p[0] + c) it = cval.erase(it); else ++it; }
                    cval[{p[0] + c, 0}] = r[c];
                }
                break;
            case XLP_K_POLS: case XLP_K_POLS_PRIME:
                if (CM) {
                    const bool pr = in.k[0] == XLP_K_POLS_PRIME;
                    for (u32 c = 0; c < dd; c++) {
                        const u64 col = p[0] + c * (pr ? p[3] : p[1]);
                        os << "    PW[" << num(col) << " + " << (pr ? "wrapc(row + " + num(p[1]) + ", A.cbase, " + num(p[2]) + ")" : std::string("row")) << "] = " << r[c] << ";\n";
                        for (auto it = mval.begin(); it != mval.end();) { if (std::get<0>(it->first) == 0 && std::get<1>(it->first) == col) it = mval.erase(it); else ++it; }
                        mval[MKey(0, col, pr ? p[1] : 0)] = r[c];
                    }
                    break;
                }
                if (in.k[0] == XLP_K_POLS) for (u32 c = 0; c < dd; c++) os << "    A.P.pols[" << num(p[0] + c) << " + row * " << num(p[1]) << "] = " << r[c] << ";\n";
                else for (u32 c = 0; c < dd; c++) os << "    A.P.pols[" << num(p[0] + c) << " + wrapr(row + " << num(p[1]) << ", " << num(p[2]) << ") * " << num(p[3]) << "] = " << r[c] << ";\n";
                dval.clear();
                break;
            case XLP_K_Q_2NS: for (u32 c = 0; c < dd; c++) os << "    A.P.q_2ns[" << (CM ? "nat3" : "3 * row") << " + " << c << "] = " << r[c] << ";\n"; break;
            case XLP_K_F_2NS: for (u32 c = 0; c < dd; c++) os << "    A.P.f_2ns[" << (CM ? "nat3" : "3 * row") << " + " << c << "] = " << r[c] << ";\n"; break;
            default: break;
            }
        }
        for (const auto& kv : tval)
            if (tdef[kv.first] && live_in[sg + 1][kv.first] && !stored.count(kv.first)) { const std::string v = cn(real(kv.second)); os << "    T[" << kv.first << "ULL * TS] = " << v << ";\n"; stats.tmp_stores++; }
        stats.longest = std::max(stats.longest, i1 - i0);
        os << "}\n";
        segsrc[sg] = place_loads(os.str(), loads, hoist, hoist_ahead, stats);
    }
    // XLP_EXPR_PARTS > 1: the kernels go round-robin to <path>.partK.hip (compiled in parallel, linked with <path>)
    size_t parts = (size_t)env_int("XLP_EXPR_PARTS", 1);
    if (parts < 1) parts = 1;
    if (parts > nseg) parts = nseg ? nseg : 1;
    auto write_file = [](const std::string& fn, const std::string& text) {
        FILE* f = fopen(fn.c_str(), "w");
        if (!f) return false;
        const bool ok = fwrite(text.data(), 1, text.size(), f) == text.size();
        fclose(f);
        return ok;
    };
    std::ostringstream os;
    os << prologue;
    // what the kernels move per row, for the build log (tools/build_expr_plugins.py reads this line)
    os << "// xlp_expr_codegen: " << ins.size() << " instructions in " << nseg << " segments (longest " << stats.longest << "), per row " << stats.loads - stats.tmp_loads
       << " column words loaded, " << stats.tmp_loads << " temporary words loaded, " << stats.tmp_stores << " temporary words stored; " << stats.pipelined_loads << " of " << stats.loads
       << " loads pipelined in " << stats.batches << " batches of " << hoist << " in " << stats.pipelined_segments << " segments\n";
    if (parts == 1) for (size_t sg = 0; sg < nseg; sg++) os << segsrc[sg];
    else {
        for (size_t k = 0; k < parts; k++) {
            std::string text = prologue;
            for (size_t sg = k; sg < nseg; sg += parts) text += segsrc[sg];
            if (!write_file(std::string(path) + ".part" + std::to_string(k) + ".hip", text)) return xlp_fail("expr codegen: cannot write a part file");
        }
        for (size_t sg = 0; sg < nseg; sg++) os << "extern \"C\" __global__ void seg_" << sg << "(XlpJitArgs A);\n";
    }
    os << "extern \"C\" int xlpjit_launch(const XlpJitArgs* a, void* stream)\n{\n    const dim3 g((unsigned)((a->nrows + 255) / 256)), b(256);\n";
    for (size_t sg = 0; sg < nseg; sg++) os << "    hipLaunchKernelGGL(seg_" << sg << ", g, b, 0, (hipStream_t)stream, *a);\n";
    os << "    return hipGetLastError() == hipSuccess ? 0 : -1;\n}\n";
    os << "extern \"C\" u64_ xlpjit_signature(void) { return 0x" << std::hex << program_signature(P) << std::dec << "ULL; }\n";
    // self-description (used by the CPU harness of the tests): staged columns {buffer, offset, stride, halo | written << 1}
    os << "extern \"C\" { extern const u64_ xlpjit_info[4]; extern const u64_ xlpjit_cols[][4]; }\n";
    os << "const u64_ xlpjit_info[4] = {" << P->cols.size() << "ULL, " << P->halo << "ULL, " << P->modulus << "ULL, " << P->tmp_words << "ULL};\n";
    os << "const u64_ xlpjit_cols[][4] = {";
    for (const xlp_expr_column& c : P->cols) os << "{" << c.buf << ", " << c.off << "ULL, " << c.stride << "ULL, " << ((c.halo ? 1 : 0) | (c.written ? 2 : 0)) /* harness: stages the global halo for flagged columns */ << "},";
    os << "{0, 0, 0, 0}};\n";
    return write_file(path, os.str()) ? 0 : xlp_fail("expr codegen: cannot write the output file");
}

int expr_attach(xlp_expr_program* P, const char* so_path)
{
    if (!P || !so_path) return xlp_fail("expr attach: null argument");
    void* h = dlopen(so_path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return xlp_fail((std::string("expr attach: ") + dlerror()).c_str());
    xlpjit_signature_fn sig = (xlpjit_signature_fn)dlsym(h, "xlpjit_signature");
    xlpjit_launch_fn launch = (xlpjit_launch_fn)dlsym(h, "xlpjit_launch");
    if (!sig || !launch) { dlclose(h); return xlp_fail("expr attach: not an xlp expression plugin"); }
    if (sig() != program_signature(P)) { dlclose(h); return xlp_fail("expr attach: the plugin was generated from a different program"); }
    if (P->jit_handle) dlclose(P->jit_handle);
    P->jit_handle = h;
    P->jit_launch = (void*)launch;
    return 0;
}

int dev_expr_run(Ctx* ctx, const xlp_expr_program* prog, const xlp_steps_params* params, u64 row0, u64 nrows)
{
    if (!prog || !params) return xlp_fail("expr run: null program or params");
    if (!nrows || prog->code.empty()) return 0;
    if (row0 + nrows < row0) return xlp_fail("expr run: row range overflows");
    if (params->zi && params->zi_size == 0) return xlp_fail("expr run: zi_size is 0");
    // launch ranges.  Row-major programs: one, rows as they are.  Column-major programs on the extended domain (coset_rows != 0): the rows
    // [row0, row0 + nrows) are DENSE coset-major indices (coset r = index / coset_rows); coset r's rows start at storage row r * coset_stride
    struct Range { u64 s0, n, coset, cbase, dshift; };
    std::vector<Range> ranges;
    if (!prog->colmajor) {
        if (check_bounds(prog, *params, row0, nrows)) return -1;
        ranges.push_back({row0, nrows, 0, 0, 0});
    } else {
        const u64 nb = params->coset_rows, bs = params->coset_stride ? params->coset_stride : nb;
        if (!nb) ranges.push_back({row0, nrows, 0, 0, 0});
        else {
            if (bs < nb) return xlp_fail("expr run: coset_stride smaller than coset_rows");
            for (u64 r = row0 / nb; r <= (row0 + nrows - 1) / nb; r++) {
                const u64 d0 = std::max(row0, r * nb), d1 = std::min(row0 + nrows, (r + 1) * nb);
                ranges.push_back({r * bs + (d0 - r * nb), d1 - d0, r, r * bs, r * (bs - nb)});
            }
        }
        for (const Range& g : ranges) if (check_bounds_cm(prog, *params, g.s0, g.s0 + g.n - 1, g.coset, g.cbase, g.dshift)) return -1;
    }
    u64 longest = 0;
    for (const Range& g : ranges) longest = std::max(longest, g.n);
    if (!prog->d_dcode) {
        xlp_expr_program* mp = const_cast<xlp_expr_program*>(prog);
        XLP_ALLOC(plan_malloc((void**)&mp->d_dcode, prog->dcode.size() * sizeof(u64)));
        XLP_HIP(hipMemcpy(mp->d_dcode, prog->dcode.data(), prog->dcode.size() * sizeof(u64), hipMemcpyHostToDevice));
    }
    const xlp_steps_params& P = *params;
    // rows per launch: 2^19-class slabs; R rows per lane (the compiled plugin runs one row per lane)
    const bool jit = prog->jit_launch != nullptr && env_int("XLP_EXPR_JIT", 1) != 0;
    int R = env_int("XLP_EXPR_ROWS", 2);
    if (R != 1 && R != 2 && R != 4) R = 2;
    if (jit) R = 1;
    // Slabs.  A slab = gather (pure HBM, 4.6 TB/s) -> the program's kernels (integer issue + latency: 56 % of VALU issue, 3.0 TB/s) ->
    // scatter.  Consecutive slabs are independent except through the halo, so every other slab runs on a second stream with its own
    // scratch: one slab's gather and kernel tails fill the other's stalls.  The scratch budget stays the same (half-size slabs).
    // (Round 3: confining the two streams to half of the CUs each -- hipExtStreamCreateWithCUMask, by XCD halves or by halves of every XCD -- so that
    // the slabs run side by side for their whole length was measured SLOWER: zkEVM step42ns 0.444 -> 0.489 / 0.561 s, profiles/r3/expr_cumask_ab.json.)
    // A program whose kernels write a direct column that they also read (a next-row store picked up by the next row: zkEVM step3prev,
    // step3) keeps one stream: running only its gathers / scatters ahead on the second one measured no gain (0.177 vs 0.179 s).
    int S = env_int("XLP_EXPR_STREAMS", 2);
    if (S < 1 || S > 4 || prog->serial_slabs || !jit) S = 1;    // (the interpreter is slower with two half-slabs: zkEVM step42ns at 2^21 rows 0.21 -> 0.30 s)
    const bool chain = false;
    const u64 wgs = (u64)(env_int("XLP_EXPR_SLAB_WGS", 8) > 0 ? env_int("XLP_EXPR_SLAB_WGS", 8) : 8);   // workgroups per CU and slab, all streams together
    u64 chunk_max = (u64)ctx->ncu * wgs * 256;
    while (S > 1 && ((nrows <= chunk_max / S * (S - 1) && ranges.size() < 2) || wgs < (u64)S)) S--;           // every stream gets a slab
    chunk_max = chunk_max / S / 256 * 256;
    const u64 chunk = longest < chunk_max ? longest : chunk_max;
    const u64 threads_max = ((chunk + R - 1) / R + 255) / 256 * 256;     // lanes of the largest launch
    const u32 ncols = (u32)prog->cols.size(), nw = (u32)prog->wcols.size();
    const u64 cap = (u64)R * threads_max + prog->halo;
    ExprArgs A;
    A.colmajor = prog->colmajor ? 1 : 0; A.coset = A.cbase = A.dshift = 0;
    A.crows = P.n_const_pols ? P.const_pols_size / P.n_const_pols : 0; A.crows2 = P.n_const_pols_2ns ? P.const_pols_2ns_size / P.n_const_pols_2ns : 0;
    A.code = prog->d_dcode; A.ncode = prog->dcode.size();
    A.hot_words = prog->hot_words; A.tstride = (u64)R * threads_max; A.tmp = nullptr; A.cols = nullptr; A.cap = cap; A.P = P;
    const u64 spill = jit ? prog->tmp_words : prog->tmp_words - prog->hot_words;
    u64* tmp_all = nullptr;
    if (spill && ctx_scratch(ctx, (size_t)S * spill * A.tstride * sizeof(u64), 2, &tmp_all)) return -1;
    u64* cols_all = nullptr; ColDesc* d_tab = nullptr; u32* d_wl = nullptr;
    if (ncols) {
        if (ctx_scratch(ctx, (size_t)S * ncols * cap * sizeof(u64), 4, &cols_all)) return -1;
        u64* raw = nullptr;
        if (ctx_scratch(ctx, ncols * sizeof(ColDesc) + (nw ? nw : 1) * sizeof(u32), 5, &raw)) return -1;
        d_tab = (ColDesc*)raw; d_wl = (u32*)(d_tab + ncols);
        std::vector<ColDesc> tab(ncols);
        for (u32 c = 0; c < ncols; c++) {
            const xlp_expr_column& k = prog->cols[c];
            u64* base = k.buf == 0 ? P.pols : k.buf == 1 ? const_cast<u64*>(P.const_pols) : const_cast<u64*>(P.const_pols_2ns);
            const u64 stride = k.buf == 0 ? k.stride : k.buf == 1 ? P.n_const_pols : P.n_const_pols_2ns;
            tab[c] = {base + k.off, stride, k.halo};
        }
        XLP_HIP(hipMemcpyAsync(d_tab, tab.data(), ncols * sizeof(ColDesc), hipMemcpyHostToDevice, ctx->stream));
        if (nw) XLP_HIP(hipMemcpyAsync(d_wl, prog->wcols.data(), nw * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
        XLP_HIP(hipStreamSynchronize(ctx->stream));       // `tab` is a stack object
    }
    hipStream_t streams[4] = {ctx->stream, ctx->stream, ctx->stream, ctx->stream};
    u64 nslabs = 0;
    for (const Range& g : ranges) nslabs += (g.n + chunk - 1) / chunk;
    hipEvent_t computed = nullptr;                        // chain: the previous slab's kernels are done
    std::vector<hipEvent_t> gathered;                     // gathered[i]: slab i's inputs (its rows + halo) have been staged
    hipEvent_t first_written = nullptr;                   // slab 0's destination columns are back in the sections
    if (S > 1) {
        hipEvent_t start;
        XLP_HIP(hipEventCreateWithFlags(&start, hipEventDisableTiming));
        XLP_HIP(hipEventRecord(start, ctx->stream));
        for (int a = 1; a < S; a++) {
            if (!ctx->aux_stream[a - 1]) XLP_HIP(hipStreamCreateWithFlags(&ctx->aux_stream[a - 1], hipStreamNonBlocking));
            streams[a] = ctx->aux_stream[a - 1];
            XLP_HIP(hipStreamWaitEvent(streams[a], start, 0));
        }
        XLP_HIP(hipEventDestroy(start));
        gathered.resize(nslabs, nullptr);
    }
    const size_t lds = (size_t)prog->hot_words * R * 256 * sizeof(u64);
    int rc = 0;
    u64 i = 0;
    for (const Range& g : ranges)
    for (u64 r0 = 0; r0 < g.n && !rc; r0 += chunk, i++) {
        const u64 row0 = g.s0, nrows = g.n;                   // (shadow the call's range: this launch range)
        A.coset = g.coset; A.cbase = g.cbase; A.dshift = g.dshift;
        const u64 rows = nrows - r0 < chunk ? nrows - r0 : chunk;
        const int s = (int)(i % (u64)S);
        hipStream_t st = streams[s];
        u64* cols = cols_all ? cols_all + (size_t)s * ncols * cap : nullptr;
        A.tmp = tmp_all ? tmp_all + (size_t)s * spill * A.tstride : nullptr;
        A.cols = cols;
        if (S > 1 && i == nslabs - 1 && i > 0 && first_written) XLP_HIPV(hipStreamWaitEvent(st, first_written, 0));   // its halo wraps to slab 0's rows: the values written there
        if (ncols) {
            const dim3 grid((ncols + 63) / 64, (unsigned)((rows + prog->halo + 63) / 64));
            hipLaunchKernelGGL(expr_gather_kernel, grid, dim3(256), 0, st, cols, d_tab, ncols, cap, row0 + r0, rows, prog->halo, prog->modulus);
        }
        if (S > 1) {
            XLP_HIPV(hipEventCreateWithFlags(&gathered[i], hipEventDisableTiming));
            XLP_HIPV(hipEventRecord(gathered[i], st));
        }
        A.row0 = row0 + r0; A.nrows = rows;
        if (S > 1 && chain && computed) { XLP_HIPV(hipStreamWaitEvent(st, computed, 0)); XLP_HIPV(hipEventDestroy(computed)); computed = nullptr; }
        if (jit && plan_active()) {}                          // (plan mode: the plugin's kernels are not launched either)
        else if (jit) {
            XlpJitArgs J;
            J.cols = cols; J.cap = cap; J.tmp = A.tmp; J.tstride = A.tstride; J.row0 = row0 + r0; J.nrows = rows; J.P = P;
            J.coset = A.coset; J.cbase = A.cbase; J.dshift = A.dshift; J.crows = A.crows; J.crows2 = A.crows2;
            if (((xlpjit_launch_fn)prog->jit_launch)(&J, (void*)st)) rc = xlp_fail("expr run: the compiled plugin failed to launch");
        } else {
            const dim3 vgrid((unsigned)((((rows + R - 1) / R) + 255) / 256));
            if (R == 1) hipLaunchKernelGGL(expr_kernel<1>, vgrid, dim3(256), lds, st, A);
            else if (R == 2) hipLaunchKernelGGL(expr_kernel<2>, vgrid, dim3(256), lds, st, A);
            else hipLaunchKernelGGL(expr_kernel<4>, vgrid, dim3(256), lds, st, A);
        }
        if (S > 1 && chain) { XLP_HIPV(hipEventCreateWithFlags(&computed, hipEventDisableTiming)); XLP_HIPV(hipEventRecord(computed, st)); }
        if (nw && !rc) {
            if (S > 1 && i > 0) XLP_HIPV(hipStreamWaitEvent(st, gathered[i - 1], 0));    // slab i-1 staged its halo (= my first rows) before I overwrite them
            const dim3 grid((nw + 63) / 64, (unsigned)((rows + 63) / 64));
            hipLaunchKernelGGL(expr_scatter_kernel, grid, dim3(256), 0, st, cols, d_tab, d_wl, nw, cap, row0 + r0, rows);
            if (S > 1 && i == 0) { XLP_HIPV(hipEventCreateWithFlags(&first_written, hipEventDisableTiming)); XLP_HIPV(hipEventRecord(first_written, st)); }
        }
    }
    if (S > 1) {                                          // later work on the library's stream sees every slab
        for (int a = 1; a < S; a++) {
            hipEvent_t done;
            XLP_HIPV(hipEventCreateWithFlags(&done, hipEventDisableTiming));
            XLP_HIPV(hipEventRecord(done, streams[a]));
            XLP_HIPV(hipStreamWaitEvent(ctx->stream, done, 0));
            XLP_HIPV(hipEventDestroy(done));
        }
        for (hipEvent_t e : gathered) if (e) XLP_HIPV(hipEventDestroy(e));
        if (first_written) XLP_HIPV(hipEventDestroy(first_written));
        if (computed) XLP_HIPV(hipEventDestroy(computed));
    }
    if (rc) return rc;
    XLP_HIP(hipGetLastError());
    return 0;
}

int expr_create(xlp_expr_program** out, const u64* code, u64 ncode, u64 ntmp1, u64 ntmp3, u64 flags)
{
    if (flags & ~(u64)(XLP_EXPR_COLMAJOR | XLP_EXPR_CANONICAL)) return xlp_fail("expr create: unknown flag");
    if ((flags & XLP_EXPR_CANONICAL) && !(flags & XLP_EXPR_COLMAJOR)) return xlp_fail("expr create: XLP_EXPR_CANONICAL goes with XLP_EXPR_COLMAJOR");
    if (!out) return xlp_fail("expr create: null output");
    *out = nullptr;
    if (ncode && !code) return xlp_fail("expr create: null code");
    if (ntmp1 > (1u << 20) || ntmp3 > (1u << 20)) return xlp_fail("expr create: more than 2^20 temporaries");
    u64 ninstr = 0;
    if (validate(code, ncode, ntmp1, ntmp3, ninstr)) return -1;
    xlp_expr_program* p = new xlp_expr_program();
    p->code.assign(code, code + ncode);
    // numbers are stored canonical so the kernel can use them as they are
    {
        std::string msg; u64 pc = 0;
        while (pc < ncode) {
            Instr in; const u64 nx = decode(p->code.data(), ncode, pc, in, msg);
            const Operand* ops[2] = {&in.a, &in.b};
            for (const Operand* o : ops)
                if (o->kind == XLP_K_NUMBER) p->code[(size_t)(o->p - p->code.data())] = o->p[0] % hgl::P;
            pc = nx;
        }
    }
    p->ntmp1 = ntmp1; p->ntmp3 = ntmp3; p->ninstr = ninstr;
    p->colmajor = (flags & XLP_EXPR_COLMAJOR) != 0;
    p->canonical = (flags & XLP_EXPR_CANONICAL) != 0;
    compile(p);
    // the device copy is made by the first run: creating, generating code and attaching need no GPU
    *out = p;
    return 0;
}

int expr_destroy(xlp_expr_program* p)
{
    if (!p) return 0;
    if (p->d_dcode) (void)plan_free(p->d_dcode);
    if (p->jit_handle) dlclose(p->jit_handle);
    delete p;
    return 0;
}

}  // namespace xlp
