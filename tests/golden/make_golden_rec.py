"""Golden vectors of the `.rec` wire format and of the arithmetic coder, produced by the REAL reference code:
/root/reference/rec/io/utils.py + data_structures.py running on the reference's own Cython coder, which oracle/ref_io.py
compiles (oracle/build_ref.sh) into a temporary directory outside the tree and removes when this script ends.  Only inputs and
expected outputs are stored (no reference source, no compiled reference).

Run (build container only):  python tests/golden/make_golden_rec.py
"""
import contextlib
import io
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_io  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        return fn(*a, **k)


def main():
    U = quiet(ref_io.load)
    rng = np.random.default_rng(2024)
    # ---- raw arithmetic-coder vectors: counts, message -> code bits ----
    cases = {}
    for ci, (nsym, mlen) in enumerate([(38, 1), (38, 9), (38, 73), (12, 9), (65, 2000), (404, 300), (3, 50)]):
        P = np.ones(nsym, dtype=np.int32)
        if ci % 2 == 0:
            P[1:] += 1000                                   # the index-stream model, utils.py:31-35
        else:
            P[1:] = rng.integers(1, 100, size=nsym - 1) + 1  # rec/io/tests/coding_test.py:13-14
        msg = np.concatenate([rng.integers(1, nsym, size=mlen), [0]]).astype(np.int64)
        ac = quiet(U.ArithmeticCoder, P, precision=32)
        code = quiet(ac.encode, msg)
        dec = quiet(ac.decode_fast, code)
        assert list(dec) == msg.tolist()
        cases[f"P{ci}"] = P.astype(np.int64)
        cases[f"msg{ci}"] = msg
        cases[f"code{ci}"] = np.frombuffer("".join(code).encode(), dtype=np.uint8)
    cases["n_cases"] = 7
    np.savez_compressed(os.path.join(HERE, "rec_ac_vectors.npz"), kind="ac", **cases)
    # ---- whole .rec files: the RVAE tensor fixture's indices as three residual blocks + a ragged synthetic set ----
    t = np.load(os.path.join(HERE, "tensor_rvae_cfg2.npz"))
    idx = [t["indices"][r, :t["K"][r]].tolist() for r in range(len(t["K"]))]
    sets = {
        "rvae": dict(seed=42, image_shape=(32, 32, 3), block_size=1000, max_index=40,
                     block_indices=[idx, idx[::-1], idx[:4]]),
        "ragged": dict(seed=7, image_shape=(512, 768, 3), block_size=1000, max_index=20,
                       block_indices=[[rng.integers(0, 20, size=k).tolist() for k in rng.integers(1, 12, size=nb)]
                                      for nb in (13, 5, 1)]),
    }
    out = {"names": np.array(list(sets))}
    for name, s in sets.items():
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "x.rec")
            quiet(U.write_compressed_code, path, s["seed"], s["image_shape"], s["block_size"], s["block_indices"], s["max_index"])
            data = open(path, "rb").read()
            back = quiet(U.read_compressed_code, path)
        assert back[3] == [[list(map(int, b)) for b in blk] for blk in s["block_indices"]]
        out[f"{name}_bytes"] = np.frombuffer(data, dtype=np.uint8)
        out[f"{name}_meta"] = np.array([s["seed"], s["block_size"], s["max_index"], *s["image_shape"]], dtype=np.int64)
        flat = [np.asarray(b, dtype=np.int64) for blk in s["block_indices"] for b in blk]
        out[f"{name}_flat"] = np.concatenate(flat)
        out[f"{name}_lens"] = np.array([len(b) for blk in s["block_indices"] for b in blk], dtype=np.int64)
        out[f"{name}_nblocks"] = np.array([len(blk) for blk in s["block_indices"]], dtype=np.int64)
        print(name, len(data), "bytes")
    np.savez_compressed(os.path.join(HERE, "rec_files.npz"), kind="rec", **out)


if __name__ == "__main__":
    main()
