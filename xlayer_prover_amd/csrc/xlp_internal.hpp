// Internal declarations shared by the HIP translation units of libxlp_gl.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <map>
#include <string>
#include <tuple>
#include <vector>
#include "../../include/xlp_gl.h"

namespace xlp {

typedef uint64_t u64;
typedef uint32_t u32;

// Host-side scalar Goldilocks helpers: used only to derive launch constants
// (roots of unity, 1/n, coset generators) -- never on the data path.
namespace hgl {
constexpr u64 P = 0xFFFFFFFF00000001ULL;
inline u64 mul(u64 a, u64 b) { return (u64)(((unsigned __int128)a * b) % P); }
inline u64 add(u64 a, u64 b) { return (u64)(((unsigned __int128)a + b) % P); }
inline u64 sub(u64 a, u64 b) { return (u64)(((unsigned __int128)a + P - (b % P)) % P); }
inline u64 pow(u64 a, u64 e)
{
    u64 r = 1;
    while (e) { if (e & 1) r = mul(r, a); a = mul(a, a); e >>= 1; }
    return r;
}
inline u64 inv(u64 a) { return pow(a, P - 2); }
}  // namespace hgl

int xlp_fail(const char* msg);             // records msg, returns -1
int xlp_fail_hip(hipError_t e, const char* file, int line);

// ---- PLAN MODE (xlp_plan_begin / XLP_PLAN=1; include/xlp_gl.h): a dry run of the host orchestration that touches no GPU.  Device allocations are
// ACCOUNTED (a fake, never dereferenced address per allocation; live and peak bytes), everything else the library would ask of the HIP runtime is
// skipped: the entry points return 0 at XLP_ENTER unless they are the ones that size memory (allocation, scratch slots, tables), and inside those every
// HIP call (XLP_HIP / XLP_HIPV), kernel launch (hipLaunchKernelGGL below) and plugin launch is a no-op.  What it is for: the memory plan of a row-sharded
// proof at sizes and rank counts no box of the build pool can hold (tests/test_memory_plan.py: 2^23 rows over 2, 4, 8 ranks), from the code that allocates.
extern bool g_plan_active;
inline bool plan_active() { return g_plan_active; }
hipError_t plan_malloc(void** p, size_t bytes);       // hipMalloc, or its account in plan mode
hipError_t plan_free(void* p);                        // hipFree, or its account in plan mode

#define XLP_HIP(expr)                                                   \
    do {                                                                \
        if (!xlp::plan_active()) {                                      \
            hipError_t _e = (expr);                                     \
            if (_e != hipSuccess) return xlp_fail_hip(_e, __FILE__, __LINE__); \
        }                                                               \
    } while (0)
// an allocation / release: always evaluated (plan mode accounts it)
#define XLP_ALLOC(expr)                                                 \
    do {                                                                \
        hipError_t _e = (expr);                                         \
        if (_e != hipSuccess) return xlp_fail_hip(_e, __FILE__, __LINE__); \
    } while (0)
// a HIP call whose result is ignored
#define XLP_HIPV(expr) do { if (!xlp::plan_active()) (void)(expr); } while (0)

struct Ctx {
    int device = 0;
    hipStream_t stream = nullptr;           // the CURRENT stream: every xlp_dev_* / xlp_comm_* call enqueues here (xlp_stream_select)
    bool own_stream = false;
    hipStream_t main_stream = nullptr;      // stream 0 of xlp_stream_select (what `stream` is unless stream 1 is selected)
    hipStream_t side_stream = nullptr;      // stream 1: exchanges and their pack / unpack copies beside the kernels of stream 0 (host/starks.hpp)
    int selected = 0;
    hipStream_t down_stream = nullptr;      // device -> host results of the pipelined host-pointer entry points
    hipStream_t aux_stream[3] = {nullptr, nullptr, nullptr};   // more compute streams: the slabs of a step program go round them (dev_expr_run)
    hipStream_t copy_stream = nullptr;      // host <-> device transfers that overlap kernels on `stream` (xlp_dev_commit_from_host)
    std::map<std::tuple<u32, u64, u64>, u64*> tables;  // (L, root, mult) -> device power tables
    void* scratch[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};      // slot 8: the BN254 transform's twiddle table, nobody else's
    size_t scratch_bytes[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    void* lent[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // caller-owned regions standing in for a slot (xlp_scratch_lend)
    size_t lent_bytes[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    // the table slot 8 holds: w^j, j < 2^(fft_tw_log - 1), w the 2^fft_tw_log-th root, at fft_tw_ptr (0 / null: none).  Kept across calls of
    // xlp_[dev_]fft_bn254; dropped when the slot is released, lent or taken back.
    u32 fft_tw_log = 0;
    const void* fft_tw_ptr = nullptr;
    bool inited = false;
    u32 ncu = 256;             // compute units of the bound device (persistent-kernel grid size)
    // optional per-kernel timing of the LDE fast path (xlp_profile_*): HIP events on ctx->stream
    bool profile = false;
    std::vector<hipEvent_t> prof_events;   // 4 events per recorded LDE chunk
};

// Grow-only device scratch slots owned by the context.
int ctx_scratch(Ctx* ctx, size_t bytes, int slot, u64** out);

u64 host_w(unsigned n);  // Goldilocks::w(n)

// ntt.hip
int ntt_init(Ctx* ctx);
int dev_ntt(Ctx* ctx, u64* dst, const u64* src, u64 n, u64 ncols, int inverse);
int dev_extend(Ctx* ctx, u64* out, const u64* in, u64 n_ext, u64 n, u64 ncols, u64* scratch);
int dev_extend_cols(Ctx* ctx, u64* out, const u64* in, u64 n_ext, u64 n, u64 ld, u64 first, u64 ncols);

// ntt_lde.hip (three-kernel fast path of extendPol)
bool lde_fast_supported(u32 L, u32 e, u64 ncols);
bool ntt_fast_supported(u32 L, u64 ncols);
int dev_ntt_fast(Ctx* ctx, u64* dst, const u64* src, u32 L, u64 ncols, int inverse);
int dev_extend_fast(Ctx* ctx, u64* out, const u64* in, u32 L, u32 e, u64 ncols);
int dev_extend_fast_cols(Ctx* ctx, u64* out, const u64* in, u32 L, u32 e, u64 ld, u64 first, u64 ncols);
int dev_lde_resident(Ctx* ctx, u64* sink, u32 L, u32 e, u64 ncols);

// colmajor.hip / ntt_lde_cm.hip (column-major device sections)
int dev_rows_to_cm(Ctx* ctx, u64* dst, u64 cstride, u64 cos, const u64* src, u64 ld, u64 col0, u64 ncols, u64 rows, u32 ext);
int dev_cm_to_rows(Ctx* ctx, u64* dst, u64 ld, u64 col0, const u64* src, u64 cstride, u64 cos, u64 ncols, u64 rows, u32 ext);
int dev_copy_view(Ctx* ctx, u64* dst, u64 drs, u64 dcs, const u64* src, u64 srs, u64 scs, u64 n, u32 dim);
int dev_copy_3d(Ctx* ctx, u64* dst, u64 ds0, u64 ds1, const u64* src, u64 ss0, u64 ss1, u64 n0, u64 n1, u64 n);
int dev_extend_cm(Ctx* ctx, u64* out, u64 out_cstride, u64 out_cos, const u64* in, u64 in_cstride, u64 n_ext, u64 n, u64 ncols);

// poseidon.hip
int poseidon_init(Ctx* ctx);
int dev_perm(Ctx* ctx, u64* out, const u64* in, u64 count);
int dev_perm_chain(Ctx* ctx, u64* out, const u64* state, const u64* elems, u64 nblocks);
int dev_perm_resident(Ctx* ctx, u64* out, u64 count, u64 iters);
int dev_linear_hash(Ctx* ctx, u64* out, const u64* src, u64 w, u64 h);
int dev_merkle(Ctx* ctx, u64* nodes, const u64* src, u64 w, u64 h);
int dev_linear_hash_absorb(Ctx* ctx, u64* state, const u64* src, u64 ld, u64 first, u64 ncols, u64 h, int is_first, int is_last);
int dev_merkle_levels(Ctx* ctx, u64* nodes, u64 h);
int dev_merkle_proofs(Ctx* ctx, u64* out, const u64* src, const u64* nodes, u64 w, u64 h, const u64* d_idxs, u64 nq);
int dev_linear_hash_absorb_cm(Ctx* ctx, u64* state, const u64* src, u64 cstride, u64 cos, u64 first, u64 ncols, u64 h, u32 ext, int is_first, int is_last);
int dev_merkle_cm(Ctx* ctx, u64* nodes, const u64* src, u64 cstride, u64 cos, u64 w, u64 h, u32 ext);
int dev_merkle_proofs_cm(Ctx* ctx, u64* out, const u64* src, u64 cstride, u64 cos, const u64* nodes, u64 w, u64 h, u32 ext, const u64* d_idxs, u64 nq);

// fri.hip
int dev_fri_fold(Ctx* ctx, u64* pol2, const u64* pol, u32 bits, u32 red, const u64 x[3], u64 shift_inv);
int dev_transpose3(Ctx* ctx, u64* dst, const u64* src, u64 n, u32 tbits);
int dev_fill(Ctx* ctx, u64* dst, u64 n, u64 seed, u64 start);
int dev_gather_rows(Ctx* ctx, u64* dst, const u64* src, u64 src_stride, u64 n, u32 dim, u64 seed);
int dev_field_op(Ctx* ctx, u64* out, const u64* a, const u64* b, u64 n, int op);

// polinomial.hip
int dev_batch_inverse(Ctx* ctx, u64* res, const u64* src, u64 n, u32 dim);
int dev_calculate_z(Ctx* ctx, u64* z, u64 zs, const u64* num, u64 ns, const u64* den, u64 ds, u64 n);
int dev_calculate_h1h2(Ctx* ctx, u64* h1, u64 h1s, u64* h2, u64 h2s, const u64* f, u64 fs, const u64* t, u64 ts, u64 n, u32 dim);
int dev_scale3(Ctx* ctx, u64* z, u64 zs, u64 n, const u64 s[3]);
int dev_copy_cols(Ctx* ctx, u64* dst, u64 dst_stride, const u64* src, u64 src_stride, u64 n, u32 dim);
int dev_q_split(Ctx* ctx, u64* qq2, const u64* qq1, u64 n, u64 n_ext, u32 qdeg, u32 qdim, u64 shift_in);
int dev_geometric(Ctx* ctx, u64* out, u64 n, u32 dim, const u64* first, const u64* ratio);
int dev_x_div_x_sub_xi(Ctx* ctx, u64* out, const u64* x, const u64* xi, u64 n);
int dev_dot_columns(Ctx* ctx, u64* out, const xlp_dot_entry* entries, u64 nentries, u32 ngroups, const u64* coefs, u64 coefs_size, u64 nrows);
int dev_evmap(Ctx* ctx, u64* evals, const u64* LEv, const u64* LpEv, u64 n, u32 eb, u32 nev, const xlp_evmap_entry* entries);
int dev_evmap_cm(Ctx* ctx, u64* evals, const u64* LEv, const u64* LpEv, u64 n, u32 nev, const xlp_evmap_entry* entries);
int dev_dot_columns_cm(Ctx* ctx, u64* out, u64 out_cstride, const xlp_dot_entry* entries, u64 nentries, u32 ngroups, const u64* coefs, u64 coefs_size, u64 nrows);


// bn128.hip (Poseidon-BN128, MerkleTreeBN128)
int bn128_init(Ctx* ctx, const char* path);
void bn128_shutdown();
u64 bn128_num_nodes(u64 n);
int dev_merkle_bn128(Ctx* ctx, u64* nodes, const u64* src, u64 w, u64 h);
int dev_poseidon_bn128(Ctx* ctx, u64* states, u32 t, u64 count);
int dev_merkle_bn128_proofs(Ctx* ctx, u64* out, const u64* src, const u64* nodes, u64 w, u64 h, const u64* d_idxs, u64 nq);
int dev_fft_bn254(Ctx* ctx, u64* data, u64 n, int inverse);
// bn254_msm.hip
int dev_msm_bn254(Ctx* ctx, u64* out, const u64* bases, const u64* scalars, u64 n);
int dev_msm_resident(Ctx* ctx, u64* out, const u64* point, u32 lanes, u32 iters);
u32 msm_bn254_window_bits(u64 n);

// fflonk.hip (vectors of BN254 Fr elements: the per-coefficient / per-evaluation work of the fflonk prover)
int dev_fr_gather_mont(Ctx* ctx, u64* out, const u32* map, u64 nmap, const u64* values, u64 nvalues);
int dev_fr_from_mont(Ctx* ctx, u64* out, const u64* in, u64 n);
int dev_fr_mul(Ctx* ctx, u64* out, const u64* a, const u64* b, u64 n);
int dev_fr_lincomb(Ctx* ctx, u64* out, u64 len, u32 nterms, const u64* const* pols, const u64* lens, const u64* coefs);
int dev_fr_interleave(Ctx* ctx, u64* out, u64 out_len, u32 k, const u64* const* pols, const u64* lens);
int dev_fr_div_zh(Ctx* ctx, u64* pol, u64 len, u64 n);
int dev_fr_batch_inverse(Ctx* ctx, u64* out, const u64* in, u64 n);
int dev_fr_prefix_product(Ctx* ctx, u64* out, const u64* in, u64 n);
int dev_fr_div_xm_minus(Ctx* ctx, u64* out, const u64* in, u64 len, u32 m, const u64* beta);
int dev_fr_eval(Ctx* ctx, u64* values, const u64* pol, u64 len, const u64* points, u32 npoints);
int dev_fflonk_t0(Ctx* ctx, u64* out, const u64* a, const u64* b, const u64* c, const u64* ql, const u64* qr, const u64* qm, const u64* qo, const u64* qc, const u64* lagrange,
                  const u64* pub, u32 npub, u64 n4);
int dev_fflonk_z_numden(Ctx* ctx, u64* num, u64* den, const u64* a, const u64* b, const u64* c, const u64* s1, const u64* s2, const u64* s3, u64 n, const u64* scalars, const u64* wpow, u32 nw);
int dev_fflonk_z_rotate(Ctx* ctx, u64* z, const u64* run, u64 n);
int dev_fflonk_t1(Ctx* ctx, u64* t1, u64* t1z, const u64* z, const u64* lagrange1, u64 n2, const u64* blinding, const u64* wpow, u32 nw);
int dev_fflonk_t2(Ctx* ctx, u64* t2, u64* t2z, const u64* a, const u64* b, const u64* c, const u64* z, const u64* s1, const u64* s2, const u64* s3, u64 n4, const u64* scalars, const u64* wpow, u32 nw);

// comm.hip (transport between the ranks of a row-sharded proof)
int comm_init(Ctx* ctx, int rank, int world, const char* backend, const char* rendezvous);
void comm_shutdown();
int comm_rank();
int comm_world();
int comm_ranks_seen();
void comm_abort(const char* why);
int comm_exchange(Ctx* ctx, u64* const* recv, const u64* recv_words, const u64* const* send, const u64* send_words, u64 all_max_words);
int comm_all_gather(Ctx* ctx, u64* out, const u64* in, u64 words);
int comm_sendrecv(Ctx* ctx, u64* recv, int from, const u64* send, int to, u64 words);
int comm_all_gather_host(Ctx* ctx, u64* out, const u64* in, u64 words);
int comm_barrier(Ctx* ctx);
int comm_stats(double* seconds, u64* bytes_sent, u64* calls, int mode);

// expr.hip
int expr_create(xlp_expr_program** out, const u64* code, u64 ncode, u64 ntmp1, u64 ntmp3, u64 flags);
int expr_destroy(xlp_expr_program* prog);
int expr_codegen(const xlp_expr_program* prog, const char* path);
int expr_attach(xlp_expr_program* prog, const char* so_path);
int dev_expr_run(Ctx* ctx, const xlp_expr_program* prog, const xlp_steps_params* params, u64 row0, u64 nrows);

}  // namespace xlp

// every kernel of the library is launched through hipLaunchKernelGGL: in plan mode (above) none is
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, ...) do { if (!xlp::plan_active()) hipLaunchKernelGGLInternal((kernelName), __VA_ARGS__); } while (0)
