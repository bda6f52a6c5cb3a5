"""extendPol's middle pass exists twice: on the matrix pipe (csrc/ntt_lde_mid_mfma.hip, the default) and on the vector pipe
(csrc/ntt_lde_mid_kernel.inc / ntt_lde_cm_mid.hip, XLP_LDE_MFMA=0).  Both must give the same extended matrix, bit for bit, on both layouts, for every
blow-up, ragged widths and non-canonical input words; at a size the oracle finishes in seconds the matrix-pipe result is also checked against the oracle
(the other sizes are covered by tests/test_gpu_parity.py / test_colmajor_kernels.py / test_full_size.py, which run the default kernel).
The arithmetic behind the kernel is checked without a GPU in tests/test_mfma_model.py."""
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def xlp():
    import torch
    from xlayer_prover_amd import capi
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    capi.init(0)
    capi.use_torch_stream()
    yield capi
    os.environ.pop("XLP_LDE_MFMA", None)


def _extend(xlp, src, lg, e, w, mfma, colmajor):
    import torch
    os.environ["XLP_LDE_MFMA"] = "1" if mfma else "0"      # read by the library at every call
    n, n_ext = 1 << lg, 1 << (lg + e)
    out = torch.empty(n_ext * w, dtype=torch.int64, device="cuda")
    if colmajor:
        xlp.dev_extend_cm(out, n_ext, src, n, n_ext, n, w)
    else:
        xlp.dev_extend(out, src, n_ext, n, w)
    xlp.sync()
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("colmajor", [False, True])
@pytest.mark.parametrize("lg,e,w", [(17, 1, 8), (17, 0, 4), (18, 2, 5), (19, 3, 7), (20, 1, 37), (21, 1, 9), (23, 1, 8)])
def test_matrix_pipe_middle_pass_equals_the_vector_kernel(xlp, lg, e, w, colmajor):
    import torch
    n = 1 << lg
    src = torch.empty(n * w, dtype=torch.int64, device="cuda")
    xlp.dev_fill(src, 0x5EED3F3A + lg + w, 0)
    # non-canonical words and the extremes in the first rows (the row-major entry point takes any 64-bit word; a column-major section is canonical)
    if not colmajor:
        edge = np.array([0xFFFFFFFFFFFFFFFF, 0xFFFFFFFF00000000, 0xFFFFFFFF00000001, 0, 1, 0x8080808080808080, 0x7F7F7F7F7F7F7F7F, 0xFFFFFFFF00000002], np.uint64)
        src[: len(edge)] = torch.from_numpy(edge.view(np.int64)).cuda()
    a = _extend(xlp, src, lg, e, w, False, colmajor)
    b = _extend(xlp, src, lg, e, w, True, colmajor)
    assert torch.equal(a, b)


@pytest.mark.gpu
def test_matrix_pipe_middle_pass_equals_the_oracle(xlp, orc):
    lg, e, w = 17, 1, 4
    n, n_ext = 1 << lg, 1 << (lg + e)
    h = orc.fill(n * w, 0x5EED3F3B).reshape(n, w)
    out = _extend(xlp, xlp.from_np(h.reshape(-1)), lg, e, w, True, False)
    assert (xlp.to_np(out).reshape(n_ext, w) == orc.extend_pol(h, n_ext, n, w)).all()
