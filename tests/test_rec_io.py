"""CPU tests of the .rec wire format: the C++ arithmetic coder and the container writer /
reader against golden vectors produced by the REAL reference code (tests/golden/make_golden_rec.py), and -- when the
build container has /root/reference and oracle/_ref -- against the live reference on fresh random inputs."""
import contextlib
import io
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR

pytestmark = [pytest.mark.both_suites, pytest.mark.usefixtures("suite")]   # also run (as gpu-marked items) by the driver on the GPU box



def _ac_cases():
    g = np.load(os.path.join(GOLDEN_DIR, "rec_ac_vectors.npz"))
    return [(g[f"P{i}"], g[f"msg{i}"], g[f"code{i}"].tobytes().decode()) for i in range(int(g["n_cases"]))]


@pytest.mark.parametrize("case", range(7))
def test_arithmetic_coder_matches_reference_vectors(case):
    from irec.io import ArithmeticCoder
    P, msg, code = _ac_cases()[case]
    ac = ArithmeticCoder(P, precision=32)
    assert "".join(ac.encode(msg)) == code                    # bit-identical to the reference's encoder
    assert ac.decode_fast(list(code)) == msg.tolist()
    assert ac.decode(list(code)) == msg.tolist()


def _rec_sets():
    g = np.load(os.path.join(GOLDEN_DIR, "rec_files.npz"))
    out = []
    for name in g["names"]:
        seed, bs, max_index, h, w, c = (int(v) for v in g[f"{name}_meta"])
        flat, lens, nblocks = g[f"{name}_flat"], g[f"{name}_lens"], g[f"{name}_nblocks"]
        blocks, pos, li = [], 0, 0
        for nb in nblocks:
            blk = []
            for _ in range(int(nb)):
                blk.append(flat[pos:pos + lens[li]].tolist())
                pos += int(lens[li]); li += 1
            blocks.append(blk)
        out.append((str(name), seed, (h, w, c), bs, max_index, blocks, g[f"{name}_bytes"].tobytes()))
    return out


@pytest.mark.parametrize("which", [0, 1])
def test_rec_file_bytes_match_reference(tmp_path, which):
    from irec.io import write_compressed_code, read_compressed_code
    name, seed, shape, bs, max_index, blocks, golden = _rec_sets()[which]
    path = tmp_path / f"{name}.rec"
    write_compressed_code(str(path), seed, shape, bs, blocks, max_index)
    assert path.read_bytes() == golden                        # byte-identical container
    gpath = tmp_path / "golden.rec"
    gpath.write_bytes(golden)
    rseed, rshape, rbs, rblocks = read_compressed_code(str(gpath))
    assert (rseed, rshape, rbs) == (seed, shape, bs)
    assert rblocks == blocks


def test_rec_rejects_index_beyond_max_index(tmp_path):
    # the reference overflows its 21-entry count table when S = 36 > max_index = 20; we refuse
    from irec.io import write_compressed_code
    with pytest.raises(ValueError, match="max_index"):
        write_compressed_code(str(tmp_path / "x.rec"), 42, (32, 32, 3), 1000, [[[3, 35]]], 20)
    with pytest.raises(ValueError, match="rank 3"):
        write_compressed_code(str(tmp_path / "x.rec"), 42, (32, 32), 1000, [[[3]]], 20)


def test_coder_rejects_bad_input():
    from irec.io import ArithmeticCoder
    with pytest.raises(ValueError):
        ArithmeticCoder(np.array([1, 0, 5]))
    ac = ArithmeticCoder(np.array([1, 5, 5]))
    with pytest.raises(ValueError, match="out of range"):
        ac.encode([1, 3, 0])
    try:   # garbage in: must terminate, either with a message ending in the terminator or with an error
        out = ac.decode_fast(list("1111111111111111111111111111111111111111"))
        assert out[-1] == 0
    except ValueError:
        pass


def test_empty_message_round_trip():
    from irec.io import ArithmeticCoder
    ac = ArithmeticCoder(np.array([1, 1001, 1001]))
    code = ac.encode([0])
    assert ac.decode_fast(code) == [0]


def test_against_live_reference_when_available(tmp_path):
    from oracle import ref_io
    if not ref_io.available():
        pytest.skip("no reference sources here (GPU box)")
    from irec.io import ArithmeticCoder, write_compressed_code
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        U = ref_io.load()
        rng = np.random.default_rng(99)
        for trial in range(25):
            nsym = int(rng.integers(2, 300))
            P = rng.integers(1, 2000, size=nsym).astype(np.int32)
            msg = np.concatenate([rng.integers(1, nsym, size=int(rng.integers(0, 400))), [0]]).astype(np.int64)
            ref_code = "".join(U.ArithmeticCoder(P, precision=32).encode(msg))
            mine = ArithmeticCoder(P, precision=32)
            assert "".join(mine.encode(msg)) == ref_code, trial
            assert mine.decode_fast(list(ref_code)) == msg.tolist()
        for S1, n in ((1, 3000), (2, 800)):   # near-deterministic models: thousands of symbols per bit
            P = np.array([1] + [1001] * S1, dtype=np.int32)
            msg = np.concatenate([rng.integers(1, S1 + 1, size=n), [0]]).astype(np.int64)
            ref_code = "".join(U.ArithmeticCoder(P, precision=32).encode(msg))
            mine = ArithmeticCoder(P, precision=32)
            assert "".join(mine.encode(msg)) == ref_code
            assert mine.decode_fast(list(ref_code)) == msg.tolist()
        blocks = [[rng.integers(0, 36, size=int(k)).tolist() for k in rng.integers(1, 15, size=9)] for _ in range(24)]
        ref_path, my_path = tmp_path / "ref.rec", tmp_path / "mine.rec"
        U.write_compressed_code(str(ref_path), 42, (32, 32, 3), 1000, blocks, 40)
        write_compressed_code(str(my_path), 42, (32, 32, 3), 1000, blocks, 40)
        assert ref_path.read_bytes() == my_path.read_bytes()
        assert U.read_compressed_code(str(my_path))[3] == blocks


def test_native_container_equals_the_per_stream_writer(tmp_path):
    """irec_rec_encode_file / irec_rec_decode_file (the whole container in one C++ call) against the reference-shaped
    per-stream Python writer / reader, which the golden tests above pin to the real reference: byte-identical files, same
    decoded structure -- random block structures, empty index lists (K = 0), one-block residual blocks, S up to 403."""
    from irec.io import utils as U
    rng = np.random.default_rng(5)
    for case in range(40):
        R = int(rng.integers(1, 9))
        S = int(rng.choice([4, 20, 36, 148, 403]))
        blocks = []
        for _ in range(R):
            nb = int(rng.integers(1, 12))
            blocks.append([rng.integers(0, S, int(rng.choice([0, 1, 2, 7, 9, 40]))).tolist() for _ in range(nb)])
        shape = (int(rng.integers(1, 4000)), int(rng.integers(1, 4000)), 3)
        seed, bs = int(rng.integers(0, 2 ** 32)), int(rng.choice([1000, 64, 4096]))
        a, b = tmp_path / f"a{case}.rec", tmp_path / f"b{case}.rec"
        U.write_compressed_code(str(a), seed, shape, bs, blocks, S)                      # native path
        U._write_compressed_code_py(str(b), seed, shape, bs, blocks, S)                  # reference-shaped path
        assert a.read_bytes() == b.read_bytes(), case
        assert U.read_compressed_code(str(a)) == (seed, shape, bs, blocks)               # native reader
        assert U._read_compressed_code_py(str(a)) == (seed, shape, bs, blocks)
    # long chains (a lossy image at 0.7 bpp: 300 blocks of ~200 indices) and the near-deterministic model of max_index 1, which
    # packs thousands of symbols into a few bits (the reader's first guess at the index count is then short: second call)
    for blocks, S in (([[rng.integers(0, 20, int(k)).tolist() for k in rng.integers(180, 237, 302)]], 20),
                      ([[[0] * 5000 for _ in range(3)]], 1)):
        a, b = tmp_path / f"long{S}.rec", tmp_path / f"long{S}b.rec"
        U.write_compressed_code(str(a), 7, (512, 768, 3), 1000, blocks, S)
        U._write_compressed_code_py(str(b), 7, (512, 768, 3), 1000, blocks, S)
        assert a.read_bytes() == b.read_bytes()
        assert U.read_compressed_code(str(a)) == (7, (512, 768, 3), 1000, blocks)
        assert U._read_compressed_code_py(str(a)) == (7, (512, 768, 3), 1000, blocks)
    with pytest.raises(ValueError):
        U.write_compressed_code(str(tmp_path / "bad.rec"), 1, (8, 8, 3), 1000, [[[0, 36]]], 36)   # index 36 needs max_index >= 37
    with pytest.raises(ValueError):
        U._native_decode(b"\x00" * 26 + b"\x05\x00" + b"\x00" * 12)            # R = 5 but the dynamic header is truncated


def test_residual_blocks_whose_partition_counts_are_all_zero(tmp_path):
    """A residual block whose coded blocks all have K = 0 (posterior == prior: nothing to send) is coded with the count model
    [1, 101]: 0.014 bit per block, so hundreds of blocks fit a few bytes of count stream.  The reader's damaged-header bound
    must not take such a file for a damaged one (round 3's bound assumed one bit per block); byte-identical to the
    reference-shaped writer, and through the batched calls."""
    from irec.io import utils as U
    for name, blocks in (("thousand", [[[] for _ in range(1000)]]),
                         ("mixed", [[[] for _ in range(200)], [[3, 1, 4], [], [1]] + [[] for _ in range(7)]]),
                         ("sixty_four", [[[] for _ in range(64)], [[] for _ in range(64)]])):
        a, b = tmp_path / f"{name}.rec", tmp_path / f"{name}_py.rec"
        U.write_compressed_code(str(a), 11, (32, 32, 3), 1000, blocks, 36)
        U._write_compressed_code_py(str(b), 11, (32, 32, 3), 1000, blocks, 36)
        assert a.read_bytes() == b.read_bytes()
        assert U.read_compressed_code(str(a)) == (11, (32, 32, 3), 1000, blocks)
        assert U._read_compressed_code_py(str(a)) == (11, (32, 32, 3), 1000, blocks)
    assert (tmp_path / "thousand.rec").stat().st_size < 100          # 1000 blocks in a handful of bytes
    # the packed form: one image whose second residual block is all zero, 64 blocks per residual block
    K = np.zeros((2, 2, 64), dtype=np.int32)
    K[:, 0, :] = 2
    idx = np.random.default_rng(3).integers(0, 36, (2, 2, 64, 2)).astype(np.int32)
    blob, off = U.encode_files(5, (32, 32, 3), 1000, K, idx, 36)
    hdr, K2, idx2 = U.decode_files(blob, off, 2, 64, 2)
    assert (K2 == K).all() and (idx2[:, 0] == idx[:, 0]).all() and (idx2[:, 1] == 0).all()
    # the bound still refuses a header that claims more blocks than any count stream of that length can hold
    data = bytearray((tmp_path / "thousand.rec").read_bytes())
    data[28:32] = (10 ** 6).to_bytes(4, "little")
    with pytest.raises(ValueError):
        U._native_decode(bytes(data))


def test_batched_container_calls_equal_the_per_image_writer(tmp_path):
    """irec_rec_encode_files / irec_rec_decode_files (round 3: the files of a whole batch from ONE packed read-back, on host
    threads) against the per-image writer that the golden tests above pin to the real reference: byte-identical files, for the
    golden structures re-packed and for random ones (rows of K = 0, ragged K, a single residual block), and the inverse."""
    from irec.io import utils as U
    rng = np.random.default_rng(17)
    cases = []
    for name, seed, shape, bs, max_index, blocks, golden in _rec_sets():
        if len({len(rb) for rb in blocks}) == 1:                           # (packed form: the same block count per residual block)
            R, bpt = len(blocks), len(blocks[0])
            mk = max(len(ix) for rb in blocks for ix in rb)
            K = np.array([[len(ix) for ix in rb] for rb in blocks], dtype=np.int32)[None]
            idx = np.zeros((1, R, bpt, mk), dtype=np.int32)
            for r, rb in enumerate(blocks):
                for j, ix in enumerate(rb):
                    idx[0, r, j, :len(ix)] = ix
            cases.append((seed, shape, bs, max_index, K, idx, [golden]))
    for n, R, bpt, mk, S in ((9, 5, 9, 12, 36), (3, 1, 1, 4, 20), (40, 24, 9, 16, 36)):
        K = rng.integers(0, mk + 1, (n, R, bpt)).astype(np.int32)
        idx = rng.integers(0, S, (n, R, bpt, mk)).astype(np.int32)
        cases.append((42, (32, 32, 3), 1000, S, K, idx, None))
    for seed, shape, bs, max_index, K, idx, goldens in cases:
        blob, off = U.encode_files(seed, shape, bs, K, idx, max_index)
        n, R, bpt = K.shape
        for i in range(n):
            bi = [[idx[i, r, j, :K[i, r, j]].tolist() for j in range(bpt)] for r in range(R)]
            want = goldens[i] if goldens else U._native_encode(seed, shape, bs, bi, max_index)
            assert blob[off[i]:off[i + 1]].tobytes() == want, i
        hdr, K2, idx2 = U.decode_files(blob, off, R, bpt, idx.shape[3])
        live = np.arange(idx.shape[3])[None, None, None, :] < K[..., None]
        assert (K2 == K).all() and ((idx2 == idx) | ~live).all() and (idx2[~live] == 0).all()
        assert (hdr[:, 0] == seed).all() and (hdr[:, 2] == max_index).all() and (hdr[:, 8] == R).all()
    # errors name the image: an index beyond max_index, a partition count beyond the rows, a file of another structure
    K = np.ones((3, 2, 2), dtype=np.int32); idx = np.zeros((3, 2, 2, 2), dtype=np.int32)
    bad = idx.copy(); bad[1, 0, 0, 0] = 99
    with pytest.raises(ValueError, match="image 1"):
        U.encode_files(1, (8, 8, 3), 10, K, bad, 36)
    Kbad = K.copy(); Kbad[2, 1, 1] = 3
    with pytest.raises(ValueError, match="image 2"):
        U.encode_files(1, (8, 8, 3), 10, Kbad, idx, 36)
    blob, off = U.encode_files(1, (8, 8, 3), 10, K, idx, 36)
    with pytest.raises(ValueError, match="structure"):
        U.decode_files(blob, off, 2, 3, 2)


def test_damaged_containers_are_errors_not_crashes(tmp_path):
    """The native reader on damaged files: every prefix of a valid container and 800 copies with one to three random bytes
    replaced either decode to some structure or raise ValueError with the library's message -- never an uncaught C++
    exception (a header claiming 2^32 symbols used to end the process with std::length_error), never an allocation sized by
    a damaged header.  The batched reader names the damaged member of a blob."""
    from irec.io import utils as U
    rng = np.random.default_rng(11)
    blocks = [[rng.integers(0, 36, int(k)).tolist() for k in rng.integers(0, 30, 9)] for _ in range(6)]
    path = tmp_path / "a.rec"
    U.write_compressed_code(str(path), 42, (32, 32, 3), 1000, blocks, 36)
    data = path.read_bytes()
    assert U._native_decode(data) == (42, (32, 32, 3), 1000, blocks)
    ok = err = 0
    for n in range(len(data)):
        with pytest.raises(ValueError):
            U._native_decode(data[:n])
    for _ in range(800):
        b = bytearray(data)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        try:
            U._native_decode(bytes(b)); ok += 1
        except ValueError:
            err += 1
    assert ok + err == 800 and err > 400
    # a batch whose second member has a damaged static header (max_index = 2^32 - 1)
    K = np.array([[len(ix) for ix in rb] for rb in blocks], dtype=np.int32)[None].repeat(3, axis=0)
    mk = int(K.max())
    idx = np.zeros((3, 6, 9, mk), dtype=np.int32)
    for i in range(3):
        for r, rb in enumerate(blocks):
            for j, ix in enumerate(rb):
                idx[i, r, j, :len(ix)] = ix
    blob, off = U.encode_files(42, (32, 32, 3), 1000, K, idx, 36)
    hdr, K2, idx2 = U.decode_files(blob, off, 6, 9, mk)
    assert np.array_equal(K2, K) and np.array_equal(idx2, idx)
    bad = blob.copy()
    bad[int(off[1]) + 8:int(off[1]) + 12] = 255
    with pytest.raises(ValueError, match="image 1"):
        U.decode_files(bad, off, 6, 9, mk)


@pytest.mark.timeout(120)
def test_arithmetic_coder_rejects_models_it_cannot_code():
    """irec_ac_encode / irec_ac_decode with hostile arguments return an error with text: counts below 1, counts whose total
    exceeds a quarter of the code range (a symbol could get an empty interval: the reference's coder then never terminates,
    this one used to as well), precisions outside 8..40, symbols outside the model -- and random bit strings decode or fail,
    in bounded time."""
    import ctypes
    from irec import _lib
    lib = _lib.load()
    msg = np.array([1, 2, 1, 0], dtype=np.int64)
    bits = np.zeros(256, dtype=np.uint8)
    nb = ctypes.c_int64(0)

    def enc(counts, precision=32, m=msg):
        c = np.array(counts, dtype=np.int64)
        return lib.irec_ac_encode(ctypes.c_void_p(c.ctypes.data), len(counts), ctypes.c_void_p(m.ctypes.data), len(m), precision,
                                  ctypes.c_void_p(bits.ctypes.data), bits.size, ctypes.byref(nb))
    assert enc([1, 5, 5]) == 0 and nb.value > 0
    for bad in ([0, 0, 0], [-1, 5, 5], [2 ** 62, 2 ** 62, 5], [1, 2 ** 29, 2 ** 29 + 1], [1]):
        assert enc(bad) != 0 and b"irec_ac_encode" in lib.irec_io_last_error(), bad
    assert enc([1, 2 ** 29, 2 ** 29 - 1]) == 0                   # total = 2^30 = a quarter of the 32-bit range: still fine
    for precision in (0, -5, 1, 7, 41, 63, 64, 1000):
        assert enc([1, 5, 5], precision) != 0, precision
    counts = np.array([1, 5, 5], dtype=np.int64)
    out, nm = np.zeros(64, dtype=np.int64), ctypes.c_int64(0)
    rng = np.random.default_rng(0)
    for _ in range(300):
        rb = (rng.integers(0, 2, int(rng.integers(1, 200))) + ord("0")).astype(np.uint8)
        lib.irec_ac_decode(ctypes.c_void_p(counts.ctypes.data), 3, ctypes.c_void_p(rb.ctypes.data), rb.size, 32,
                           ctypes.c_void_p(out.ctypes.data), out.size, ctypes.byref(nm))
    big = np.array([1, 2 ** 31, 5], dtype=np.int64)
    assert lib.irec_ac_decode(ctypes.c_void_p(big.ctypes.data), 3, ctypes.c_void_p(bits.ctypes.data), 8, 32,
                              ctypes.c_void_p(out.ctypes.data), out.size, ctypes.byref(nm)) != 0
    # int(np.exp(Omega * (1 + eps))) saturates instead of wrapping (beam_search_coder.py:28-29)
    assert lib.irec_n_samples(ctypes.c_double(3.0), ctypes.c_double(1.2)) == 36
    assert lib.irec_n_samples(ctypes.c_double(50.0), ctypes.c_double(2.0)) == 2 ** 31 - 1
    assert lib.irec_n_samples(ctypes.c_double(float("nan")), ctypes.c_double(1.0)) == 0
