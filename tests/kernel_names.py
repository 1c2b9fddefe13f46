"""Kernel names: what irec_encode_plan says a call launches <-> the instantiations compiled into libirec_hip.so.  Shared by
tests/test_kernel_coverage.py and scripts/kernel_coverage.py (the kernel trace of the GPU suite)."""
import ctypes
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "relative-entropy-coding_amd", "csrc", "libirec_hip.so")


def compiled_kernels(lib=LIB):
    """Every __global__ function of the library, demangled and normalised: no namespace, no argument list, no spaces."""
    out = subprocess.run(["nm", "-C", lib], capture_output=True, text=True, check=True).stdout
    names = set()
    for line in out.splitlines():
        if "__device_stub__" not in line:
            continue
        n = line.split("__device_stub__", 1)[1]
        names.add(normalise(n))
    return names


def normalise(name):
    """'void irec::encode_team_kernel<20, 3, 1, false, ...>(irec::EncArgs)' / 'encode_team_kernel<20,3,1,...>' -> one spelling."""
    n = name.strip()
    n = re.sub(r"^void\s+", "", n)
    n = n.replace("irec::", "")
    depth, cut = 0, len(n)
    for i, ch in enumerate(n):        # the argument list: the first '(' outside the template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    return n[:cut].replace(" ", "")


def canonical(plan_kernel, split=0):
    """The instantiation behind a name irec_encode_plan returns (first-pass block kernel), as compiled_kernels() spells it."""
    k = plan_kernel.strip()
    if k.startswith("encode_generic_kernel"):
        return "encode_generic_kernel"
    if k == "encode_lone_kernel":
        return k
    m = re.match(r"encode_ten_kernel<(\d+)>$", k)
    if m:
        return f"encode_ten_kernel<{m.group(1)}>"
    m = re.match(r"encode_team_kernel<(\d+),(\d+),(\d+)((?:,\w+)*)>$", k)
    if m:
        tags = set(t for t in m.group(4).split(",") if t)
        b = lambda v: "true" if v else "false"
        share = split >= 2
        return (f"encode_team_kernel<{m.group(1)},{m.group(2)},{m.group(3)},{b('passes' in tags)},{b('one' in tags)},"
                f"{b(share)},{b('margins' in tags)}>")
    m = re.match(r"encode_chunk_kernel<(\d+),(\d+),(\d+)(,gang)?>$", k)
    if m:
        return f"encode_chunk_kernel<{m.group(1)},{m.group(2)},{m.group(3)},{'true' if m.group(4) else 'false'}>"
    m = re.match(r"encode_fast_kernel<(\d+),(\d+),(true|false)(?:,(\d))?>$", k)
    if m:
        return f"encode_fast_kernel<{m.group(1)},{m.group(2)},{m.group(3)},{m.group(4) or 0}>"
    raise ValueError(f"unknown kernel name {plan_kernel!r}")


# the grid the planner is enumerated over (tests/test_kernel_coverage.py): beams, samples, (largest block, max_K), blocks per call, flags
BEAMS = (1, 2, 5, 7, 10, 11, 16, 20, 21, 25, 30, 31, 32, 33, 40, 48, 50, 54, 57, 60, 61, 100)
SAMPLES = (1, 7, 20, 25, 26, 36, 38, 39, 54, 55, 80, 102, 103, 148, 244, 403, 735, 1339, 1808, 8103)
DIMS = ((192, 32), (1000, 32), (2048, 64))
BLOCKS = (1, 9, 13, 40, 63, 64, 72, 126, 200, 256, 257, 302, 342, 400, 512, 513, 700, 2304)


def flag_sets(_lib):
    sh = _lib.IREC_FLAG_SHAPE
    return (0, _lib.IREC_FLAG_NO_SPLIT, _lib.IREC_FLAG_TEAM, _lib.IREC_FLAG_ONE_TABLE, _lib.IREC_FLAG_FUSED_PHILOX, _lib.IREC_FLAG_FORCE_GENERIC,
            _lib.IREC_FLAG_MARGINS, _lib.IREC_FLAG_NO_TEN, _lib.IREC_FLAG_NO_TEN | _lib.IREC_FLAG_NO_SPLIT, sh["2"], sh["3"], sh["1x2"],
            sh["team"] | _lib.IREC_FLAG_TEAM, sh["3"] | _lib.IREC_FLAG_NO_TEN)


def enumerate_plans(n_cu=256, dims=DIMS, blocks=BLOCKS):
    """canonical kernel name -> the cheapest call of the grid that launches it: dict(B, S, n_blocks, dim, max_K, flags, plan)."""
    import sys
    for p in (ROOT, os.path.join(ROOT, "relative-entropy-coding_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from irec import _lib
    from irec.engine import Engine
    lib = _lib.load()
    info, det = _lib.IrecPlanInfo(), _lib.IrecPlanDetail()
    best = {}
    for B in BEAMS:
        for S in SAMPLES:
            if S * B >= 1 << 24:
                continue
            for dim, max_K in dims:
                for flags in flag_sets(_lib):
                    p = Engine.params(3.0, S, B, flags, [dim])
                    for nb in blocks:
                        st = lib.irec_test_plan(n_cu, 2400, ctypes.byref(p), nb, dim, max_K, ctypes.byref(info), ctypes.byref(det))
                        assert st == 0, (B, S, dim, flags, nb, lib.irec_last_error())
                        name = canonical(info.kernel.decode(), info.split)
                        cost = nb * dim * S * B
                        if name not in best or cost < best[name]["cost"]:
                            best[name] = dict(B=B, S=S, n_blocks=nb, dim=dim, max_K=max_K, flags=flags, cost=cost,
                                              kernel=info.kernel.decode(), split=int(info.split))
    return best
