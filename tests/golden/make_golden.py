"""Generates the golden fixtures under tests/golden/ from the CPU oracle (oracle/irec_oracle.c, canonical mode).

The reference's own tests hold no golden indices (rec/coding/tests/test_coder.py:12-21 is a round trip) and
TensorFlow 2.1 cannot run in this image, so these vectors are SELF-PINNED: they freeze the oracle's
behaviour so that (a) the oracle cannot drift silently and (b) the GPU box, which has no /root/reference and needs
no oracle build to read them, can check the HIP path against committed data.

Run:  python tests/golden/make_golden.py      (rewrites tests/golden/*.npz)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def block_case(name, mq, sq, mp, sp, seed, omega, eps1, B):
    S = O.n_samples(omega, eps1)
    idx, sample, tr = O.encode_block(mq, sq, mp, sp, seed, omega, S, B, trace=True)
    idx_lit, sample_lit = O.encode_block(mq, sq, mp, sp, seed, omega, S, B, mode=O.LITERAL)
    dec = O.decode_block(mp, sp, idx, seed, S)
    assert np.array_equal(dec, sample), name
    np.savez_compressed(os.path.join(HERE, name + ".npz"), kind="block", q_loc=mq, q_scale=sq, p_loc=mp, p_scale=sp,
                        seed=seed, kl_per_partition=omega, extra_samples=eps1, n_beams=B, n_samples=S,
                        kl=np.float32(O.block_kl(mq, sq, mp, sp)), K=len(idx), indices=np.array(idx, np.int32),
                        sample=sample, sel=tr["sel"], score_step0=tr["score"][0] if len(idx) else np.zeros(0, np.float32),
                        indices_literal=np.array(idx_lit, np.int32), codelength=O.codelength(idx, S))
    print(f"{name}: D={len(mq)} S={S} B={B} K={len(idx)} literal_equal={idx == idx_lit}")


def large_blocks():
    """Round 4: blocks of more than 1024 dims -- Coder.__init__ takes any block_size, None (the whole tensor as one block)
    included, rec/coding/coder.py:29-36,415-419: the chunked encoder's and the generic kernel's committed vectors."""
    for name, D, omega, eps1, B in (("block_D2500_large", 2500, 3.0, 1.2, 20), ("block_D1500_large_b10", 1500, 3.0, 1.0, 10)):
        mq, sq, mp, sp = O.synthetic_latent(8800 + D, D)
        block_case(name, mq, sq, mp, sp, 42, omega, eps1, B)
    # Round 5: the chunked encoder's beam passes (20 < B <= 32: 30 slots in three passes of 10, 32 in two of 16)
    for name, D, omega, eps1, B in (("block_D3200_large_b30", 3200, 3.0, 1.2, 30), ("block_D2100_large_b32", 2100, 3.0, 1.0, 32)):
        mq, sq, mp, sp = O.synthetic_latent(8800 + D, D)
        block_case(name, mq, sq, mp, sp, 42, omega, eps1, B)


def main():
    configs = [(3.0, 1.2, 20), (3.0, 1.0, 10), (5.0, 1.0, 30), (6.0, 1.0, 10)]
    for D in (1, 192, 1000):
        for ci, (omega, eps1, B) in enumerate(configs):
            mq, sq, mp, sp = O.synthetic_latent(100 * D + ci, D)
            if D == 1:  # a single dim needs a real KL to produce partitions
                mq = mp + sp * np.float32(3.0 + ci)
                sq = sp * np.float32(0.05)
            block_case(f"block_D{D}_cfg{ci}", mq, sq, mp, sp, 42, omega, eps1, B)
    # the reference's own unit-test case, rec/coding/tests/test_coder.py:12-21
    block_case("ref_test_beam_search", np.float32([5.1]), np.float32([0.001]), np.float32([0.0]), np.float32([1.0]),
               69420, 6.0, 1.0, 10)
    # ragged dims around the 256-dim group / 4-dim quad boundaries
    for D in (3, 255, 257, 1023, 1024):
        mq, sq, mp, sp = O.synthetic_latent(7000 + D, D)
        block_case(f"block_D{D}_ragged", mq, sq, mp, sp, 1234, 3.0, 1.2, 20)
    # S < B on the first steps (beam_search_coder.py:104-106 keeps only S beams)
    mq, sq, mp, sp = O.synthetic_latent(9001, 64)
    block_case("block_S_lt_B", mq, sq * np.float32(0.5), mp, sp, 5, 1.5, 1.0, 20)

    large_blocks()

    # one full RVAE-shaped latent tensor [1,16,16,32], block_size 1000 -> 8 x 1000 + 192
    omega, eps1, B, seed, bs = 3.0, 1.2, 20, 42, 1000
    S = O.n_samples(omega, eps1)
    mq, sq, mp, sp = O.synthetic_latent(0, 8192)
    shape = (1, 16, 16, 32)
    idx, sample = O.encode_tensor(mq.reshape(shape), sq.reshape(shape), mp.reshape(shape), sp.reshape(shape), seed,
                                  omega, S, B, block_size=bs)
    dec = O.decode_tensor(mp.reshape(shape), sp.reshape(shape), idx, seed, S, block_size=bs)
    assert np.array_equal(dec, sample)
    Ks = np.array([len(i) for i in idx], np.int32)
    flat = np.full((len(idx), Ks.max()), -1, np.int32)
    for r, i in enumerate(idx):
        flat[r, :len(i)] = i
    np.savez_compressed(os.path.join(HERE, "tensor_rvae_cfg2.npz"), kind="tensor", q_loc=mq.reshape(shape),
                        q_scale=sq.reshape(shape), p_loc=mp.reshape(shape), p_scale=sp.reshape(shape), seed=seed,
                        kl_per_partition=omega, extra_samples=eps1, n_beams=B, n_samples=S, block_size=bs, K=Ks,
                        indices=flat, sample=sample, perm_head=O.tf_shuffle_perm(seed, 8192)[:64],
                        codelength=sum(O.codelength(i, S) for i in idx))
    print("tensor_rvae_cfg2: K per block", Ks.tolist())


if __name__ == "__main__":
    if sys.argv[1:] == ["large"]:      # only the round-4 additions (the older fixtures are not rewritten)
        large_blocks()
    else:
        main()
