"""Multi-GPU commitment of one STARK section, one process per GPU.

    1. LDE          -- rank g owns the columns col_range(W, G, g) (columns are independent:
                       NTT_Goldilocks::extendPol loops over them, starks.cpp:53)
    2. exchange     -- all-to-all: the Merkle leaf is a sponge over a WHOLE row in column order
                       (merkleTreeGL.cpp:39-43), so hashing needs a row partition; each rank keeps
                       rows row_range(h, G, g) of every column
    3. hash         -- rank g builds the sub-tree over its rows; its sub-root is node
                       (level log2(h/G), index g) of the reference's tree
    4. caps         -- all-gather of the G sub-roots (32 bytes each), top log2 G levels hashed
                       everywhere -> the root the transcript absorbs (starks.cpp:56-60)

torch.distributed is only the transport (RCCL over xGMI on GPU, gloo in the CPU tests).  The field
arithmetic is the HIP library's; `hash_pairs` / `lde` / `merkle` are injectable so that the CPU-only
tests can check the partition / exchange / cap-assembly logic with the oracle as the checker.
"""
import numpy as np
import torch
import torch.distributed as dist


FORCE_EXCHANGE = False      # tests: take the exchange paths even in a world of one rank (RCCL plumbing on one GPU)
EXCHANGE_STATS = None       # bench: set to {} to account the all-to-alls (synchronises around each one): calls, bytes sent by this rank, seconds


MAX_MESSAGE_WORDS = 1 << 25  # 256 MiB per (sender, receiver) message and round, see _exchange


def _run_p2p(ops, group):
    """one round of sends / receives.  RCCL: one coalesced group (separately enqueued sends would wait for receives that sit behind
    the peers' own sends).  gloo (CPU tests; device tensors of ranks sharing a GPU go through the host): plain isend / irecv."""
    if not ops:
        return
    if dist.get_backend(group) != "gloo":
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        return
    reqs, back = [], []
    for op in ops:
        if op.op is dist.isend:
            reqs.append(dist.isend(op.tensor.cpu().contiguous(), op.peer, group=group))
        else:
            h = torch.empty(op.tensor.shape, dtype=op.tensor.dtype)
            reqs.append(dist.irecv(h, op.peer, group=group)); back.append((op.tensor, h))
    for r in reqs:
        r.wait()
    for d, h in back:
        d.copy_(h)


def _exchange(recv, send, max_words, group):
    """recv[p] <- what rank p holds for me, send[p] -> rank p; 1-D contiguous tensors, len(send[p]) here == len(recv[me]) on p.
    max_words: the longest message of the whole exchange (every rank derives the same value from the column spans).
    The own block is a local copy; the rest goes in rounds of <= MAX_MESSAGE_WORDS per message, each round one grouped
    send / receive (`batch_isend_irecv`: direct peer exchange, all xGMI links at once) over slices of the tensors (no packing).  Why rounds: the RCCL of this image leaves the second half of a self
    message unwritten from somewhere between 1 and 1.5 GiB on (measured on one MI355X, 2^27 words fine, 751 x 2^18 and 2^28 words
    not; peer messages of that size could not be tried in this pool), and a zkEVM section at G = 8 is 1.6 GB per message; rounds
    are also what a copy / LDE / hash pipeline across bands wants."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    st = EXCHANGE_STATS
    if st is not None:
        import time
        torch.cuda.synchronize(); t0 = time.perf_counter()
    recv[rank].copy_(send[rank])
    if world > 1:
        for o in range(0, max_words, MAX_MESSAGE_WORDS):
            ops = []
            for d in range(1, world):                         # the same pairing order on every rank; empty tails are left out
                to, frm = (rank + d) % world, (rank - d) % world
                if send[to].numel() > o:
                    ops.append(dist.P2POp(dist.isend, send[to][o:o + MAX_MESSAGE_WORDS], to, group))
                if recv[frm].numel() > o:
                    ops.append(dist.P2POp(dist.irecv, recv[frm][o:o + MAX_MESSAGE_WORDS], frm, group))
            _run_p2p(ops, group)
    if st is not None:
        torch.cuda.synchronize()
        st["exchange_calls"] = st.get("exchange_calls", 0) + 1
        st["exchange_s"] = st.get("exchange_s", 0.0) + time.perf_counter() - t0
        st["bytes_sent_to_peers"] = st.get("bytes_sent_to_peers", 0) + sum(send[p].numel() for p in range(world) if p != rank) * send[rank].element_size()


def col_range(ncols, world, rank):
    """contiguous, balanced column shard of rank (first ncols % world ranks get one extra)"""
    base, rem = divmod(ncols, world)
    c0 = rank * base + min(rank, rem)
    return c0, c0 + base + (1 if rank < rem else 0)


def row_range(h, world, rank):
    assert h % world == 0, "row count must be divisible by the number of ranks"
    per = h // world
    return rank * per, (rank + 1) * per


def _band_copy(dst, dst_col, src, src_col, rows, width):
    """dst[:, dst_col : dst_col + width] = src[:, src_col : src_col + width] for 2-D row-major tensors: one strided-copy kernel of the
    library on the GPU (xlp_dev_copy_cols), a slice assignment on the CPU (gloo tests)"""
    if width == 0 or rows == 0:
        return
    if dst.is_cuda:
        from . import capi
        capi.dev_copy_cols(dst.view(-1)[dst_col:], dst.shape[1], src.view(-1)[src_col:], src.shape[1], rows, width)
    else:
        dst[:, dst_col:dst_col + width] = src[:, src_col:src_col + width]


def columns_to_rows(local, ncols, group=None):
    """local: [h, wc_local] (this rank's columns of all rows) -> [h / G, ncols] (all columns of this
    rank's rows).  One exchange (`_exchange`) -- the rows destined to peer p are already contiguous in `local`, so the send side is
    the tensor itself; direct peer exchange, every xGMI link busy (no ring) -- and one band copy per peer that drops the
    received [h / G, wc_p] block into its columns of the row-major result (the layout the leaf sponge and the step programs read)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    h, wc = local.shape
    spans = [col_range(ncols, world, p) for p in range(world)]
    assert wc == spans[rank][1] - spans[rank][0]
    if world == 1 and not FORCE_EXCHANGE:
        return local                                          # one rank owns every column of every row already
    per = h // world
    recv = [torch.empty((per, c1 - c0), dtype=local.dtype, device=local.device) for c0, c1 in spans]
    flat = local.contiguous().view(-1)
    _exchange([r.view(-1) for r in recv], [flat[p * per * wc:(p + 1) * per * wc] for p in range(world)],
              per * max(c1 - c0 for c0, c1 in spans), group)
    out = torch.empty((per, ncols), dtype=local.dtype, device=local.device)
    for (c0, c1), blk in zip(spans, recv):
        _band_copy(out, c0, blk, 0, per, c1 - c0)
    return out


def _host_staged(t, group):
    """gloo moves host memory only: under it device tensors take a round trip through the host (tests: several ranks on one GPU)"""
    return dist.get_backend(group) == "gloo" and t.is_cuda


def _all_gather(t, group=None):
    """-> list of every rank's t (same shape everywhere), on t's device"""
    world = dist.get_world_size(group)
    src = t.contiguous()
    if _host_staged(t, group):
        parts = [torch.empty(src.shape, dtype=src.dtype) for _ in range(world)]
        dist.all_gather(parts, src.cpu(), group=group)
        return [q.to(t.device) for q in parts]
    parts = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(parts, src, group=group)
    return parts


def _sendrecv(send, dst, src, group=None):
    """send `send` to rank dst, receive a tensor of the same shape from rank src"""
    if _host_staged(send, group):
        r_h = torch.empty(send.shape, dtype=send.dtype)
        reqs = [dist.isend(send.cpu(), dst=dst, group=group), dist.irecv(r_h, src=src, group=group)]
        for r in reqs:
            r.wait()
        return r_h.to(send.device)
    recv = torch.empty_like(send)
    if dist.get_backend(group) == "nccl":
        # one coalesced group: every rank sends to its ring neighbour and receives from the other one at the same time; separately
        # enqueued sends would wait for receives that sit behind the peers' own sends
        reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, send.contiguous(), dst, group), dist.P2POp(dist.irecv, recv, src, group)])
    else:
        reqs = [dist.isend(send, dst=dst, group=group), dist.irecv(recv, src=src, group=group)]
    for r in reqs:
        r.wait()
    return recv


def rows_to_columns(local, nrows, group=None):
    """inverse of columns_to_rows.  local: [h / G, ncols] (all columns of this rank's rows) -> [h, wc_local] (this rank's
    columns of all rows): the layout the per-column LDE wants.  One band copy per peer packs its columns, the
    exchange (`_exchange`) delivers straight into the result (the block of peer p IS rows row_range(p) of the result)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    per, ncols = local.shape
    assert per * world == nrows
    if world == 1 and not FORCE_EXCHANGE:
        return local
    spans = [col_range(ncols, world, p) for p in range(world)]
    c0, c1 = spans[rank]
    out = torch.empty((nrows, c1 - c0), dtype=local.dtype, device=local.device)
    send = []
    for a, b in spans:
        blk = torch.empty((per, b - a), dtype=local.dtype, device=local.device)
        _band_copy(blk, 0, local, a, per, b - a)
        send.append(blk.view(-1))
    flat = out.view(-1)
    w_me = c1 - c0
    _exchange([flat[p * per * w_me:(p + 1) * per * w_me] for p in range(world)], send, per * max(b - a for a, b in spans), group)
    return out


def cap_tree(sub_root, hash_pairs, group=None):
    """sub_root: int64 tensor [4] = root of this rank's row range.  Returns the levels of the tree over the G sub-roots,
    [caps[G,4], [G/2,4], ..., root[1,4]]; every rank holds all of them (32 bytes per rank exchanged).
    hash_pairs(nodes[2k,4]) -> [k,4] computes perm(left|right|0^4)[0..3] per pair."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return [sub_root.view(1, 4).clone()]
    levels = [torch.stack(_all_gather(sub_root, group))]                    # [G, 4] in rank order == node order
    while levels[-1].shape[0] > 1:
        levels.append(hash_pairs(levels[-1]))
    return levels


def assemble_root(sub_root, hash_pairs, group=None):
    """Returns (root[4], caps[G,4]) of cap_tree."""
    levels = cap_tree(sub_root, hash_pairs, group)
    return levels[-1][0], levels[0]


def sharded_merkle_proof(idx, rows_local, local_proof, levels, rank):
    """Sibling path of global row idx in a row-sharded tree.  The owner (idx // rows_local) returns (row, siblings):
    local_proof(idx % rows_local) -> (row, siblings inside the rank's sub-tree), followed by the siblings on the cap tree
    (cap_tree levels, replicated).  Other ranks return None.  MerkleTreeGL::getGroupProof (merkleTreeGL.cpp:12-35)."""
    owner = idx // rows_local
    if owner != rank:
        return None
    row, sib = local_proof(idx % rows_local)
    i = owner
    for lvl in levels[:-1]:
        sib = sib + [[int(v) & 0xFFFFFFFFFFFFFFFF for v in lvl[i ^ 1].tolist()]]
        i //= 2
    return row, sib


def gpu_hash_pairs(nodes):
    """product implementation of hash_pairs: the Merkle-level kernel of libxlp_gl.so"""
    from . import capi
    k = nodes.shape[0] // 2
    # a tree over k*2 leaves of width 4 (no-hash leaf rule) has exactly these pairs at level 1
    buf = torch.empty((2 * 2 * k - 1) * 4, dtype=torch.int64, device=nodes.device)
    capi.dev_merkle(buf, nodes.contiguous().view(-1), 4, 2 * k)
    return buf[2 * k * 4: 2 * k * 4 + k * 4].view(k, 4).clone()


def sharded_commit(trace_local, n, n_ext, ncols, group=None, lde=None, merkle=None, hash_pairs=None):
    """trace_local: int64 [n, wc_local] on this rank's device.  Returns (root[4], nodes_local, ext_rows)."""
    from . import capi
    world = dist.get_world_size(group)
    wc = trace_local.shape[1]
    if lde is None or merkle is None or hash_pairs is None:
        # kernels and torch's copies / collectives must be ordered: run the library on torch's current stream
        capi.use_torch_stream()
    if lde is None:
        def lde(x):
            out = torch.empty((n_ext, wc), dtype=torch.int64, device=x.device)
            capi.dev_extend(out, x.contiguous(), n_ext, n, wc)
            return out
    if merkle is None:
        def merkle(rows):
            h, w = rows.shape
            nodes = torch.empty((2 * h - 1) * 4, dtype=torch.int64, device=rows.device)
            capi.dev_merkle(nodes, rows.contiguous(), w, h)
            return nodes
    if hash_pairs is None:
        hash_pairs = gpu_hash_pairs
    ext_local = lde(trace_local)                               # [n_ext, wc]   column shard
    ext_rows = columns_to_rows(ext_local, ncols, group)        # [n_ext/G, W]  row shard
    nodes = merkle(ext_rows)
    root, _ = assemble_root(nodes[-4:].clone(), hash_pairs, group)
    return root, nodes, ext_rows


# ---------------------------------------------------------------------------------------- constraint steps across ranks
# expression evaluation is row-local except the next-row ("prime") operands -> row partition with a halo.
# Each rank keeps a WINDOW of every row-indexed buffer: its rows followed by `halo` rows of the next rank (wrapping to
# rank 0).  The step program is evaluated on local row numbers 0 .. rows_local-1 with the GLOBAL modulus in its next-row
# operands: (i + next) % N never wraps inside a window, so the reads land in the halo rows, and xlp_dev_expr_run only
# asks for the window to be resident (csrc/expr.hip check_bounds).  Shard boundaries must be multiples of the ZhInv
# period 2^extendBits (they are for power-of-two shards).  Programs that STORE to the next row (a few zkEVM step3prev
# operations) also need the reverse exchange of the halo rows they wrote; halo_writeback does that.
def halo_exchange(local, width, halo_rows, group=None):
    """local: [rows_local * width] tensor of one row-major section.  Returns [(rows_local + halo_rows) * width]: the local
    rows followed by the first halo_rows rows of rank+1 (ring).  One small point-to-point exchange per section."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rows = local.numel() // width
    out = torch.empty((rows + halo_rows) * width, dtype=local.dtype, device=local.device)
    out[: rows * width] = local
    head = local[: halo_rows * width].contiguous()
    if world == 1:
        out[rows * width:] = head
        return out
    rank = dist.get_rank(group)
    out[rows * width:] = _sendrecv(head, (rank - 1) % world, (rank + 1) % world, group)
    return out


def halo_writeback(window, width, rows_local, halo_rows, cols, group=None):
    """After a program that stores to next rows: for each (first column, column count, next distance) in `cols` the first
    `next` halo rows of `window` hold values that belong to rank+1's first rows.  Sends them forward and overwrites the same
    columns of the local first `next` rows with what rank-1 computed (rows beyond `next` are the owner's own results)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    w2 = window.view(rows_local + halo_rows, width)
    halo = torch.cat([w2[rows_local:rows_local + nx, a:a + c].reshape(-1) for a, c, nx in cols]).contiguous()
    if world == 1:
        recv = halo
    else:
        rank = dist.get_rank(group)
        recv = _sendrecv(halo, (rank + 1) % world, (rank - 1) % world, group)
    o = 0
    for a, c, nx in cols:
        w2[:nx, a:a + c] = recv[o:o + nx * c].view(nx, c); o += nx * c
    return window


# ------------------------------------------------------------------------------------------ grand products and evaluations
# calculateZ (polinomial.hpp:586-607) is an exclusive prefix product over the rows; row-sharded it is a local scan plus ONE
# all-gather of a single Fp^3 element per rank.  evmap (starks.cpp:556-669) is a sum over the rows: local partial sums plus
# one all-gather of nev x 3 words.  Field arithmetic on the handful of exchanged elements is done on the host (Python ints).
P = 0xFFFFFFFF00000001


def _mul3(a, b):
    a0, a1, a2 = a; b0, b1, b2 = b
    return [(a0 * b0 + a1 * b2 + a2 * b1) % P, (a0 * b1 + a1 * b0 + a1 * b2 + a2 * b1 + a2 * b2) % P, (a0 * b2 + a2 * b0 + a1 * b1 + a2 * b2) % P]


def _inv3(a):
    # solve a * x = 1: multiplication by a is a 3x3 matrix over Fp in the basis 1, x, x^2 (x^3 = x + 1)
    a0, a1, a2 = a
    m = [[a0, a2, a1], [a1, (a0 + a2) % P, (a1 + a2) % P], [a2, a1, (a0 + a2) % P]]
    rhs = [1, 0, 0]
    for c in range(3):                                   # Gauss-Jordan mod p
        piv = next(r for r in range(c, 3) if m[r][c] % P)
        m[c], m[piv] = m[piv], m[c]; rhs[c], rhs[piv] = rhs[piv], rhs[c]
        iv = pow(m[c][c], P - 2, P)
        m[c] = [v * iv % P for v in m[c]]; rhs[c] = rhs[c] * iv % P
        for r in range(3):
            if r != c and m[r][c]:
                f = m[r][c]
                m[r] = [(v - f * w) % P for v, w in zip(m[r], m[c])]; rhs[r] = (rhs[r] - f * rhs[c]) % P
    return rhs


def _all_gather_small(t_cpu, group=None):
    """all-gather of a few words held on the host: RCCL moves device memory only, so the words take a round trip through
    the rank's GPU under the nccl backend (gloo sends them as they are)."""
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "nccl":
        t = t_cpu.cuda()
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t, group=group)
        return [p.cpu() for p in parts]
    parts = [torch.empty_like(t_cpu) for _ in range(world)]
    dist.all_gather(parts, t_cpu.contiguous(), group=group)
    return parts


def sharded_calculate_z(num_local, den_local, calc_z, scale3, group=None):
    """num_local / den_local: this rank's rows of the Fp^3 polynomials ([rows*3] tensors).  calc_z(num, den) -> local
    exclusive product scan starting at 1 (xlp_dev_calculate_z); scale3(z, s) multiplies every element of z by the Fp^3
    constant s in place (one expression-program launch).  Returns the rank's rows of the global z."""
    z = calc_z(num_local, den_local)
    rows = z.numel() // 3
    as_int = lambda t: [int(v) & 0xFFFFFFFFFFFFFFFF for v in t.cpu().tolist()]
    last_z, last_n, last_d = as_int(z[-3:]), as_int(num_local[-3:]), as_int(den_local[-3:])
    total = _mul3(_mul3(last_z, last_n), _inv3(last_d))            # product of this rank's ratios
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return z
    mine = torch.tensor([v if v < (1 << 63) else v - (1 << 64) for v in total], dtype=torch.int64)
    allt = _all_gather_small(mine, group)
    prefix = [1, 0, 0]
    for r in range(dist.get_rank(group)):
        prefix = _mul3(prefix, [int(v) & 0xFFFFFFFFFFFFFFFF for v in allt[r].tolist()])
    if prefix != [1, 0, 0]:
        scale3(z, prefix)
    return z


def sharded_evmap(partial_evals, group=None):
    """partial_evals: [nev*3] tensor = this rank's evmap over ITS rows (xlp_dev_evmap on the local windows of LEv / LpEv and
    of the extended sections).  Returns the evaluations (list of nev [3]-lists): the field sum over ranks."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    parts = [partial_evals.cpu()]
    if world > 1:
        parts = _all_gather_small(partial_evals.cpu().contiguous(), group)
    acc = [0] * partial_evals.numel()
    for t in parts:
        for i, v in enumerate(t.tolist()):
            acc[i] = (acc[i] + (int(v) & 0xFFFFFFFFFFFFFFFF)) % P
    return [acc[i:i + 3] for i in range(0, len(acc), 3)]


def sharded_h1h2(f_local, t_local, dim, h1h2, group=None):
    """calculateH1H2 (polinomial.hpp:349-585) is a global sort-merge of f into t; its inputs are small next to the trace
    (2 x N x dim words per lookup: 400 MB for an ext-3 lookup at N = 2^23), so every rank all-gathers f and t, runs the
    full xlp_dev_calculate_h1h2 (2.5 ms at that size) and keeps its own rows of h1 / h2: two all-gathers per lookup
    instead of a distributed sort.  h1h2(f, t) -> (h1, h2) on the full vectors."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return h1h2(f_local, t_local)
    rank = dist.get_rank(group)
    h1, h2 = h1h2(torch.cat(_all_gather(f_local, group)), torch.cat(_all_gather(t_local, group)))
    rows = f_local.numel() // dim
    return h1[rank * rows * dim:(rank + 1) * rows * dim].clone(), h2[rank * rows * dim:(rank + 1) * rows * dim].clone()
