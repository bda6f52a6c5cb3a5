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
#include <string>
#include <vector>
#include "gl_field.hpp"
#include "xlp_internal.hpp"

struct xlp_expr_program {
    std::vector<uint64_t> code;   // host copy (validated), used for the per-run bounds check
    uint64_t* d_code = nullptr;
    uint64_t ntmp1 = 0, ntmp3 = 0;
    uint64_t ninstr = 0;
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
static int check_bounds(const xlp_expr_program* prog, const xlp_steps_params& P, uint64_t row0, uint64_t nrows)
{
    const uint64_t* code = prog->code.data();
    const uint64_t ncode = prog->code.size();
    const uint64_t last = row0 + nrows - 1;
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
            case XLP_K_POLS_PRIME: hi = (unsigned __int128)p[0] + (unsigned __int128)(p[2] - 1) * p[3] + o->dim - 1; size = P.pols_size; base = P.pols; break;
            case XLP_K_CONST:
                if (p[0] >= P.n_const_pols) { hi = ~(unsigned __int128)0; }
                else hi = (unsigned __int128)p[0] + (unsigned __int128)last * P.n_const_pols;
                size = P.const_pols_size; base = P.const_pols; break;
            case XLP_K_CONST_PRIME:
                if (p[0] >= P.n_const_pols) { hi = ~(unsigned __int128)0; }
                else hi = (unsigned __int128)p[0] + (unsigned __int128)(p[2] - 1) * P.n_const_pols;
                size = P.const_pols_size; base = P.const_pols; break;
            case XLP_K_CONST2NS:
                if (p[0] >= P.n_const_pols_2ns) { hi = ~(unsigned __int128)0; }
                else hi = (unsigned __int128)p[0] + (unsigned __int128)last * P.n_const_pols_2ns;
                size = P.const_pols_2ns_size; base = P.const_pols_2ns; break;
            case XLP_K_CONST2NS_PRIME:
                if (p[0] >= P.n_const_pols_2ns) { hi = ~(unsigned __int128)0; }
                else hi = (unsigned __int128)p[0] + (unsigned __int128)(p[2] - 1) * P.n_const_pols_2ns;
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

// ------------------------------------------------------------------------------------------ device
struct ExprArgs {
    const u64* code; u64 ncode;
    u64* tmp; u64 tstride; u64 ntmp1;
    u64 row0, nrows;
    xlp_steps_params P;
};

__device__ __forceinline__ u64 uni(u64 v)   // value is wave-uniform: keep it in SGPRs
{
    return ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)v);   // the builtin returns int
}
__device__ __forceinline__ u64 wrap(u64 r, u64 n) { return (n & (n - 1)) == 0 ? (r & (n - 1)) : r % n; }

__device__ __forceinline__ e3 load_operand(const ExprArgs& A, u32 kind, u32 dim, const u64* p, u64 row, const u64* t1, const u64* t3)
{
    e3 v; v.c[0] = v.c[1] = v.c[2] = 0;
    const u64* src = nullptr;
    switch (kind) {
    case XLP_K_TMP1: v.c[0] = t1[uni(p[0]) * A.tstride]; return v;
    case XLP_K_TMP3: {
        const u64* q = t3 + 3 * uni(p[0]) * A.tstride;
        v.c[0] = q[0]; v.c[1] = q[A.tstride]; v.c[2] = q[2 * A.tstride];
        return v;
    }
    case XLP_K_NUMBER: v.c[0] = uni(p[0]); return v;
    case XLP_K_POLS: src = A.P.pols + uni(p[0]) + row * uni(p[1]); break;
    case XLP_K_POLS_PRIME: src = A.P.pols + uni(p[0]) + wrap(row + uni(p[1]), uni(p[2])) * uni(p[3]); break;
    case XLP_K_CONST: src = A.P.const_pols + uni(p[0]) + row * A.P.n_const_pols; break;
    case XLP_K_CONST_PRIME: src = A.P.const_pols + uni(p[0]) + wrap(row + uni(p[1]), uni(p[2])) * A.P.n_const_pols; break;
    case XLP_K_CONST2NS: src = A.P.const_pols_2ns + uni(p[0]) + row * A.P.n_const_pols_2ns; break;
    case XLP_K_CONST2NS_PRIME: src = A.P.const_pols_2ns + uni(p[0]) + wrap(row + uni(p[1]), uni(p[2])) * A.P.n_const_pols_2ns; break;
    case XLP_K_CHALLENGE: src = A.P.challenges + 3 * uni(p[0]); break;
    case XLP_K_EVAL: src = A.P.evals + 3 * uni(p[0]); break;
    case XLP_K_PUBLIC: src = A.P.public_inputs + uni(p[0]); break;
    case XLP_K_X_N: src = A.P.x_n + row; break;
    case XLP_K_X_2NS: src = A.P.x_2ns + row; break;
    case XLP_K_ZI: src = A.P.zi + wrap(row, A.P.zi_size); break;
    case XLP_K_XDIVXSUBXI: src = A.P.x_div_x_sub_xi + 3 * row; break;
    case XLP_K_XDIVXSUBWXI: src = A.P.x_div_x_sub_wxi + 3 * row; break;
    case XLP_K_Q_2NS: src = A.P.q_2ns + 3 * row; break;
    case XLP_K_F_2NS: src = A.P.f_2ns + 3 * row; break;
    default: return v;
    }
    v.c[0] = gl::canon(src[0]);
    if (dim == 3) { v.c[1] = gl::canon(src[1]); v.c[2] = gl::canon(src[2]); }
    return v;
}

__device__ __forceinline__ void store_operand(const ExprArgs& A, u32 kind, u32 dim, const u64* p, u64 row, u64* t1, u64* t3, const e3& v)
{
    u64* dst = nullptr;
    switch (kind) {
    case XLP_K_TMP1: t1[uni(p[0]) * A.tstride] = v.c[0]; return;
    case XLP_K_TMP3: {
        u64* q = t3 + 3 * uni(p[0]) * A.tstride;
        q[0] = v.c[0]; q[A.tstride] = v.c[1]; q[2 * A.tstride] = v.c[2];
        return;
    }
    case XLP_K_POLS: dst = A.P.pols + uni(p[0]) + row * uni(p[1]); break;
    case XLP_K_POLS_PRIME: dst = A.P.pols + uni(p[0]) + wrap(row + uni(p[1]), uni(p[2])) * uni(p[3]); break;
    case XLP_K_Q_2NS: dst = A.P.q_2ns + 3 * row; break;
    case XLP_K_F_2NS: dst = A.P.f_2ns + 3 * row; break;
    default: return;
    }
    dst[0] = v.c[0];
    if (dim == 3) { dst[1] = v.c[1]; dst[2] = v.c[2]; }
}

__global__ __launch_bounds__(256) void expr_kernel(ExprArgs A)
{
    const u64 gt = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64* t1 = A.tmp + gt;
    u64* t3 = A.tmp + A.ntmp1 * A.tstride + gt;
    for (u64 r = gt; r < A.nrows; r += A.tstride) {
        const u64 row = A.row0 + r;
        u64 pc = 0;
        while (pc < A.ncode) {
            const u64 h = uni(A.code[pc]);
            const u32 op = (u32)(h & 0xff), dk = (u32)((h >> 8) & 0xff), ak = (u32)((h >> 16) & 0xff), bk = (u32)((h >> 24) & 0xff);
            const u32 dd = (u32)((h >> 32) & 3), ad = (u32)((h >> 34) & 3), bd = (u32)((h >> 36) & 3);
            // parameter words per kind, 3 bits each (KIND_PARAMS)
            constexpr u64 NP = (1ull << 3) | (1ull << 6) | (2ull << 9) | (4ull << 12) | (1ull << 15) | (3ull << 18) | (1ull << 21) | (3ull << 24) |
                               (1ull << 27) | (1ull << 30) | (1ull << 33) | (1ull << 36);
            const u32 nd = (u32)((NP >> (3 * dk)) & 7), na = (u32)((NP >> (3 * ak)) & 7), nb = (u32)((NP >> (3 * bk)) & 7);
            const u64* pd = A.code + pc + 1;
            const u64* pa = pd + nd;
            const u64* pb = pa + na;
            pc += 1 + nd + na + nb;
            const e3 a = load_operand(A, ak, ad, pa, row, t1, t3);
            e3 v;
            if (op == XLP_OP_COPY) v = a;
            else {
                const e3 b = load_operand(A, bk, bd, pb, row, t1, t3);
                if (op == XLP_OP_ADD) {
                    v.c[0] = gl::addc(a.c[0], b.c[0]);
                    if (dd == 3) { v.c[1] = gl::addc(a.c[1], b.c[1]); v.c[2] = gl::addc(a.c[2], b.c[2]); }
                } else if (op == XLP_OP_SUB) {
                    v.c[0] = gl::subc(a.c[0], b.c[0]);
                    if (dd == 3) { v.c[1] = gl::subc(a.c[1], b.c[1]); v.c[2] = gl::subc(a.c[2], b.c[2]); }
                } else {
                    if (ad == 3 && bd == 3) v = gl::mul3(a, b);
                    else if (ad == 3) v = gl::mul3s(a, b.c[0]);
                    else if (bd == 3) v = gl::mul3s(b, a.c[0]);
                    else v.c[0] = gl::mulc(a.c[0], b.c[0]);
                }
            }
            store_operand(A, dk, dd, pd, row, t1, t3, v);
        }
    }
}

int dev_expr_run(Ctx* ctx, const xlp_expr_program* prog, const xlp_steps_params* params, u64 row0, u64 nrows)
{
    if (!prog || !params) return xlp_fail("expr run: null program or params");
    if (!nrows || prog->code.empty()) return 0;
    if (row0 + nrows < row0) return xlp_fail("expr run: row range overflows");
    if (params->zi && params->zi_size == 0) return xlp_fail("expr run: zi_size is 0");
    if (check_bounds(prog, *params, row0, nrows)) return -1;
    // one lane per row, up to 8 waves per SIMD; temporaries tmp[slot][thread]
    const u64 slots = prog->ntmp1 + 3 * prog->ntmp3;
    u64 blocks = (nrows + 255) / 256;
    const u64 max_blocks = (u64)ctx->ncu * 8;
    if (blocks > max_blocks) blocks = max_blocks;
    ExprArgs A;
    A.code = prog->d_code; A.ncode = prog->code.size();
    A.tstride = blocks * 256; A.ntmp1 = prog->ntmp1;
    A.row0 = row0; A.nrows = nrows; A.P = *params;
    A.tmp = nullptr;
    if (ctx_scratch(ctx, (slots ? slots : 1) * A.tstride * sizeof(u64), 2, &A.tmp)) return -1;
    hipLaunchKernelGGL(expr_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, A);
    XLP_HIP(hipGetLastError());
    return 0;
}

int expr_create(xlp_expr_program** out, const u64* code, u64 ncode, u64 ntmp1, u64 ntmp3)
{
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
    if (ncode) {
        hipError_t e = hipMalloc((void**)&p->d_code, ncode * sizeof(u64));
        if (e == hipSuccess) e = hipMemcpy(p->d_code, p->code.data(), ncode * sizeof(u64), hipMemcpyHostToDevice);
        if (e != hipSuccess) { if (p->d_code) (void)hipFree(p->d_code); delete p; return xlp_fail_hip(e, __FILE__, __LINE__); }
    }
    *out = p;
    return 0;
}

int expr_destroy(xlp_expr_program* p)
{
    if (!p) return 0;
    if (p->d_code) (void)hipFree(p->d_code);
    delete p;
    return 0;
}

}  // namespace xlp
