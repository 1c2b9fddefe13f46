"""The planner at other CU counts than the one box everything was measured on (round 5's review, Weak #6 / Next #5): MI355X exposes
32 / 64 / 128 CUs in its partition modes, and grids, cooperative widths and slab counts all derive from n_cu.  Host-only: irec_test_plan
(csrc/irec_internal.h) runs the code of the launch itself (irec_host.cpp: make_plan -> shape_for_call -> call_detail) for a context of any CU
count without touching a device, and the tests assert the invariants the kernels trap on or would wait 100 ms for."""
import ctypes

import pytest

pytestmark = [pytest.mark.both_suites, pytest.mark.usefixtures("suite")]

N_CU = (32, 64, 128, 256, 304)
LDS_LIMIT = 160 * 1024
# (B, S, blocks per call, largest block, table dims, max_K): the call sizes of the BASELINE configurations and of the reference's drivers
CALLS = [
    ("configs[1] headline step", 20, 36, 589824, 1000, (1000, 192), 32),
    ("configs[2] one GPU's share of 300 images", 20, 36, 342, 1000, (1000, 192), 32),
    ("configs[2] 14 latents", 20, 36, 126, 1000, (1000, 192), 32),
    ("configs[2] one image's residual block", 20, 36, 9, 1000, (1000, 192), 32),
    ("configs[3] Kodak level 1", 10, 20, 302, 1000, (1000, 56), 32),
    ("configs[3] Kodak level 2", 10, 20, 13, 1000, (1000, 288), 32),
    ("configs[3] 1024 latents", 10, 20, 9216, 1000, (1000, 192), 32),
    ("configs[3] 72 blocks", 10, 20, 72, 1000, (1000, 192), 32),
    ("configs[4] S = 148", 30, 148, 9216, 1000, (1000, 192), 32),
    ("configs[4] S = 403", 30, 403, 9216, 1000, (1000, 192), 32),
    ("one beam", 1, 20, 2304, 1000, (1000, 192), 32),
    ("B = 50", 50, 36, 2304, 1000, (1000, 192), 32),
    ("block_size = None, one latent", 20, 36, 1, 8192, (8192,), 128),
    ("block_size = None, one image", 20, 36, 24, 8192, (8192,), 128),
    ("block_size = None, batch", 20, 36, 3072, 8192, (8192,), 128),
    ("block_size = 2048", 20, 36, 512 * 4, 2048, (2048,), 128),
    ("Kodak level 1 as one block", 10, 20, 1, 301056, (301056,), 4096),
]


def _plan(n_cu, B, S, n_blocks, max_dim, dims, max_K, flags=0):
    from irec import _lib
    from irec.engine import Engine
    lib = _lib.load()
    p = Engine.params(3.0, S, B, flags, list(dims))
    info, det = _lib.IrecPlanInfo(), _lib.IrecPlanDetail()
    st = lib.irec_test_plan(n_cu, 2400, ctypes.byref(p), n_blocks, max_dim, max_K, ctypes.byref(info), ctypes.byref(det))
    assert st == 0, lib.irec_last_error()
    return info.as_dict(), det.as_dict()


@pytest.mark.parametrize("n_cu", N_CU)
@pytest.mark.parametrize("call", CALLS, ids=[c[0] for c in CALLS])
def test_launch_invariants_at_every_cu_count(n_cu, call):
    from irec import _lib
    name, B, S, n_blocks, max_dim, dims, max_K = call
    for flags in (0, _lib.IREC_FLAG_NO_SPLIT, _lib.IREC_FLAG_TEAM, _lib.IREC_FLAG_MARGINS if max_dim <= 1024 else 0):
        info, d = _plan(n_cu, B, S, n_blocks, max_dim, dims, max_K, flags)
        ctx = (name, n_cu, flags, info, d)
        # the plan irec_encode_plan reports is the launch
        assert info["grid"] == d["grid"] >= 1 and info["split"] == d["coop_width"], ctx
        if d["kind"] in (1, 3):
            assert info["teams_per_wg"] == d["teams_per_wg"], ctx
        assert info["n_cu"] == n_cu and 0 < info["lds_bytes"] <= LDS_LIMIT, ctx
        # every scratch slab a launched team can index lies inside the workspace
        assert d["fixed_bytes"] + d["slabs_in_workspace"] * d["slab_bytes"] == info["workspace_bytes"], ctx
        if d["kind"] in (1, 3):
            assert d["grid"] * d["teams_per_wg"] <= d["slabs_in_workspace"], ctx
        elif d["kind"] != 2:
            assert d["grid"] <= d["slabs_in_workspace"], ctx
        # persistent kernels: at most one workgroup per CU (launch_bounds(.., 1): the LDS of a workgroup is the CU's)
        if d["kind"] in (1, 2, 3):
            assert d["grid"] <= n_cu, ctx
        # cooperative forms: a width of two or more, or nothing; every partner resident at once (they wait for each other every step)
        assert d["coop_width"] == 0 or d["coop_width"] >= 2, ctx
        if flags & (_lib.IREC_FLAG_NO_SPLIT | _lib.IREC_FLAG_MARGINS):
            assert d["coop_width"] == 0 and d["split_blocks"] == 0, ctx
        if d["coop_width"]:
            assert d["split_blocks"] <= d["exchange_rows"], ctx
            if d["kind"] == 3:      # shared rows: the static hand-out round deals every slot
                assert d["n_slots"] <= d["grid"] * d["teams_per_wg"] and d["n_slots"] == d["share_first"] + (n_blocks - d["share_first"]) * d["coop_width"], ctx
                assert S * (10 if B <= 10 else 20) <= d["exchange_keys"] and d["coop_width"] <= min(8, S), ctx
            elif d["kind"] == 1:    # gangs: one static slot per member
                assert d["n_slots"] == n_blocks * d["coop_width"] <= d["grid"] * d["teams_per_wg"], ctx
                assert d["gang_chunks"] >= 1 and d["coop_width"] % d["gang_chunks"] == 0, ctx
            elif d["kind"] == 4:    # split encoder: within half the CUs (room for a second such call on another stream)
                assert d["grid"] == n_blocks * d["coop_width"] <= n_cu // 2 and n_blocks <= 64, ctx
                assert not d["coop_beams"] or d["coop_width"] <= B, ctx
        else:
            assert d["n_slots"] == n_blocks, ctx
        # rows dealt by cost: the kernel ranks them assuming the static round deals EVERY slot (irec_team.hip: placed)
        if d["placed"]:
            assert d["kind"] == 3 and d["teams_per_wg"] == 2 and n_blocks <= 1024 and d["n_slots"] <= d["grid"] * d["teams_per_wg"], ctx


def test_thresholds_scale_with_the_cu_count():
    """What is "a fraction of the CUs" is an expression in n_cu (irec_host.cpp: small_call_blocks, split_width, share_all_auto,
    shape_for_call, gang_width); measured constants stay constants (DESIGN.md §8 lists which)."""
    # the team encoder takes over from the one-table / split encoders at a quarter of the CUs (at most 64, at least 8 blocks)
    for n_cu, first_team_call in ((32, 8), (64, 16), (128, 32), (256, 64), (304, 64)):
        below, _ = _plan(n_cu, 20, 36, first_team_call - 1, 1000, (1000, 192), 32)
        at, _ = _plan(n_cu, 20, 36, first_team_call, 1000, (1000, 192), 32)
        assert below["kernel"].startswith("encode_fast_kernel") and at["kernel"].startswith("encode_team_kernel"), (n_cu, below, at)
    # one to two blocks per CU take the two-team shape, beyond that three teams -- wherever "one block per CU" lies
    for n_cu in N_CU:
        two, _ = _plan(n_cu, 20, 36, n_cu + n_cu // 3, 1000, (1000, 192), 32, 16)
        three, _ = _plan(n_cu, 20, 36, 4 * n_cu, 1000, (1000, 192), 32, 16)
        assert two["kernel"] == "encode_team_kernel<20,2,1>" and three["kernel"] == "encode_team_kernel<20,3,1>", (n_cu, two, three)
    # a one-block call of 8192 dims: a gang as wide as the CUs allow (8 chunk owners x sample stripes), never wider than the team slots
    for n_cu in N_CU:
        info, d = _plan(n_cu, 20, 36, 1, 8192, (8192,), 128)
        assert info["kernel"].endswith(",gang>") and 8 <= d["coop_width"] <= min(72, n_cu * d["teams_per_wg"]), (n_cu, info, d)
    # rows dealt by cost: the two-team 20-beam build at one to two rows per CU -- wherever "a row per CU" lies; calls of at most ten beams
    # deal their rows as listed (round 6: measured both ways, profiles/r06end/placement_multi_tensor.log)
    from irec import _lib
    for n_cu in N_CU:
        at, d_at = _plan(n_cu, 20, 36, n_cu, 1000, (1000, 192), 32)
        mid, d_mid = _plan(n_cu, 20, 36, n_cu + n_cu // 5, 1000, (1000, 192), 32)
        listed, d_listed = _plan(n_cu, 20, 36, n_cu + n_cu // 5, 1000, (1000, 192), 32, _lib.IREC_FLAG_LISTED_ORDER)
        assert not d_at["placed"] and d_mid["placed"] and not d_listed["placed"], (n_cu, d_at, d_mid, d_listed)
        for n_blocks in (n_cu + 1, n_cu + n_cu // 5, 2 * n_cu - 1):
            ten, d_ten = _plan(n_cu, 10, 20, n_blocks, 1000, (1000, 192), 32)
            assert ten["kernel"] == "encode_ten_kernel<2>" and not d_ten["placed"], (n_cu, n_blocks, ten, d_ten)
