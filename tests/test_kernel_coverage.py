"""Every compiled encoder instantiation is reachable through the planner, and every kernel the planner can name is checked against the
oracle (round 5's review, Weak #6 / Next #2: ~85 instantiations chosen by 300 lines of dispatch, nothing proving each one is reached by a
parity test).  rec/coding/beam_search_coder.py:53-122 is ONE algorithm: every variant must be it.

  * host-only: irec_test_plan enumerated over a grid of (beams, samples, block dims, blocks per call, flags) -- the set of kernel names it
    returns equals the set of encode_* instantiations `nm` finds in libirec_hip.so.  A build nobody can reach fails the test (that is how
    round 6 found and removed encode_chunk_kernel<20,20,2>), and so does a name without a build.
  * GPU: for every name, the cheapest call of the grid that launches it runs against the oracle, bit for bit.
The dispatch of the GPU suite as a whole is traced by scripts/kernel_coverage.py (profiles/r06*/suite_kernel_coverage.txt)."""
import ctypes
import functools

import numpy as np
import pytest

import kernel_names as kn


@functools.lru_cache(maxsize=None)
def _plans(n_cu=256):
    return kn.enumerate_plans(n_cu)


@pytest.mark.both_suites
@pytest.mark.usefixtures("suite")
def test_every_compiled_encoder_is_planned_and_every_planned_kernel_is_compiled():
    compiled = {k for k in kn.compiled_kernels() if k.startswith("encode_")}
    planned = set(_plans())
    assert not compiled - planned, f"compiled, but no call of the grid launches them: {sorted(compiled - planned)}"
    assert not planned - compiled, f"named by irec_encode_plan, but not in the library: {sorted(planned - compiled)}"
    assert len(compiled) >= 60
    # the planner names the same kernels at other CU counts (nothing is reachable only on a partition, nothing only on the full device)
    for n_cu in (64, 304):
        assert set(kn.enumerate_plans(n_cu, blocks=(1, 9, 13, 16, 17, 40, 63, 64, 72, 76, 77, 126, 200, 304, 305, 342, 400, 609, 700, 2304))) == compiled, n_cu


def _names():
    try:
        return sorted(_plans())
    except Exception:        # (the library is missing: the host test above says so)
        return []


@pytest.mark.gpu
@pytest.mark.parametrize("name", _names())
def test_every_planned_kernel_matches_the_oracle(engine, oracle, name):
    import torch
    from irec import _lib
    ex = _plans(engine.plan(engine.params(3.0, 36, 20), engine.layout(1, 192, 192, 42), 8)["n_cu"])[name]
    B, S, n_blocks, dim, max_K, flags = ex["B"], ex["S"], ex["n_blocks"], ex["dim"], ex["max_K"], ex["flags"]
    margins = bool(flags & _lib.IREC_FLAG_MARGINS)
    rng = np.random.default_rng(7)
    stats = [oracle.synthetic_latent(3000 + int(rng.integers(1 << 20)), dim) for _ in range(n_blocks)]
    host = [np.stack([s[j] for s in stats]) for j in range(4)]
    ql, qs, pl, ps = (torch.from_numpy(a).cuda().contiguous() for a in host)
    lay = engine.layout(n_blocks, dim, dim, 42)                       # one block per tensor, shuffled (coder.py:62-83)
    assert lay.n_blocks == n_blocks and lay.max_dim == dim
    params = engine.params(3.0, S, B, flags & ~_lib.IREC_FLAG_MARGINS)
    plan = engine.plan(params, lay, max_K, margins=margins)
    assert kn.canonical(plan["kernel"], plan["split"]) == name, (plan, ex)
    if margins:
        K, idx, sample, _ = engine.encode_blocks_margins(params, lay, ql, qs, pl, ps, 42, max_K)
    else:
        K, idx, sample = engine.encode_blocks(params, lay, ql, qs, pl, ps, 42, max_K)
    torch.cuda.synchronize()
    Kh, ih, sh = K.cpu().numpy(), idx.cpu().numpy(), sample.cpu().numpy().reshape(n_blocks, dim)
    assert Kh.min() >= 0 and Kh.max() <= max_K, (name, int(Kh.min()), int(Kh.max()))
    ridx, rs, _ = oracle.encode_tensors_omp(*host, 42, 3.0, S, B, dim, max_K=max_K)
    for i in range(n_blocks):
        row = lay.natural[i]
        assert ih[row, :Kh[row]].tolist() == ridx[i][0], (name, i)
    assert np.array_equal(sh, rs), name
