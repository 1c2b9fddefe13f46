"""Model call-surface shim: constructor surface on CPU; on the GPU the sequential res-block <->
coder hand-off, checked in situ against the oracle and by a compress -> decompress round trip."""
import inspect

import numpy as np
import pytest
import torch
import torch.nn.functional as F


def _model(device="cpu", blocks=3, det=16, sto=8):
    from irec.models import BidirectionalResNetVAE
    torch.manual_seed(0)
    m = BidirectionalResNetVAE(num_res_blocks=blocks, sampler="beam_search",
                               sampler_args={"n_beams": 20, "extra_samples": 1.2}, coder_args={"block_size": 1000},
                               deterministic_filters=det, stochastic_filters=sto, kl_per_partition=3.)
    with torch.no_grad():   # random-init weights (no checkpoints exist); keep posteriors close to priors so K stays small
        for b in m.residual_blocks:
            for head in (b.gen_posterior_loc_head, b.gen_posterior_log_scale_head, b.infer_posterior_loc_head,
                         b.infer_posterior_log_scale_head, b.prior_loc_head, b.prior_log_scale_head):
                head.weight.mul_(0.3)
        m._generative_base.normal_(0, 0.5)
    return m.to(device).eval()


def test_constructor_surface_matches_reference():
    from irec.models import BidirectionalResNetVAE, BidirectionalResidualBlock, ModelError
    blk = list(inspect.signature(BidirectionalResidualBlock.__init__).parameters)
    assert blk[1:6] == ["stochastic_filters", "deterministic_filters", "sampler", "sampler_args", "coder_args"]
    assert {"kernel_size", "is_last", "kl_per_partition", "name"} <= set(blk)
    vae = list(inspect.signature(BidirectionalResNetVAE.__init__).parameters)
    assert vae[1:5] == ["num_res_blocks", "sampler", "sampler_args", "coder_args"]
    assert {"first_kernel_size", "first_strides", "deterministic_filters", "stochastic_filters", "kl_per_partition"} <= set(vae)
    assert list(inspect.signature(BidirectionalResNetVAE.compress).parameters) == ["self", "image", "seed", "update_sampler"]
    m = _model()
    assert [b.is_last for b in m.residual_blocks] == [True, False, False]
    assert m.residual_blocks[0].coder.n_samples == 36 and m.residual_blocks[0].coder.block_size == 1000
    assert m.residual_blocks[2].coder.name == "encoder_for_resnet_block_2"
    with pytest.raises(ModelError, match="Sampler must be one of"):
        BidirectionalResNetVAE(num_res_blocks=1, sampler="nope")
    with pytest.raises(ModelError):
        BidirectionalResNetVAE(num_res_blocks=1, sampler="rejection")


@pytest.mark.gpu
def test_compress_decompress_round_trip_and_in_situ_parity(engine, oracle):
    m = _model("cuda")
    torch.manual_seed(1)
    image = (torch.rand(1, 3, 32, 32, device="cuda") - 0.5)
    # capture what each residual block hands to its coder
    seen = []
    for b in m.residual_blocks:
        orig = b.coder.encode

        def spy(target_dist, coding_dist, seed, _orig=orig, **kw):
            out = _orig(target_dist, coding_dist, seed, **kw)
            seen.append((target_dist.loc.clone(), target_dist.scale.clone(), coding_dist.loc.clone(),
                         coding_dist.scale.clone(), out))
            return out
        b.coder.encode = spy
    block_indices, recon = m.compress(image, seed=42)
    assert len(block_indices) == 3 and all(len(bi) == 3 for bi in block_indices)      # 2048 dims -> 1000 + 1000 + 48
    assert recon.shape == image.shape and torch.isfinite(recon).all()
    assert len(seen) == 3
    for ql, qs, pl, ps, (pending, sample) in seen:                                     # NHWC, batch 1, as in the reference
        assert ql.shape == (1, 16, 16, 8)
        idx = pending.to_lists()[0]                                                    # compress defers the host copy
        ridx, rs = oracle.encode_tensor(ql.cpu().numpy(), qs.cpu().numpy(), pl.cpu().numpy(), ps.cpu().numpy(), 42, 3.0,
                                        36, 20, block_size=1000)
        assert idx == ridx
        assert np.array_equal(sample.cpu().numpy(), rs)
    for b in m.residual_blocks:
        del b.coder.encode
    recon2 = m.decompress(block_indices, seed=42, image_shape=image.shape)
    assert torch.equal(recon2, recon)                                                   # decoder reproduces the encoder's pass
    # sequential dependence: a different latent in block 0 changes the prior block 1 sees
    alt = [[list(ix) for ix in bi] for bi in block_indices]
    alt[0][0][0] = (alt[0][0][0] + 1) % 36
    assert not torch.equal(m.decompress(alt, seed=42, image_shape=image.shape), recon)


@pytest.mark.gpu
def test_lossy_two_level_compress_file_decompress(engine, tmp_path):
    # BASELINE config 4 settings (B = 10, Omega = 3, eps = 0 -> S = 20, block_size 1000, max_index 20) on a 128 x 192 crop:
    # level 2 = [1, 2, 3, 128] (768 dims, 1 block), level 1 = [1, 8, 12, 196] (18 816 dims, 19 blocks)
    import irec
    from irec.models import Large2LevelVAE
    torch.manual_seed(3)
    m = Large2LevelVAE().cuda().eval()
    with torch.no_grad():
        for mod in (m.analysis_transform[-1], m.hyper_analysis_transform[-1], m.hyper_synthesis_transform[-1],
                    m._prior_loc_head, m._prior_log_scale_head, m._level_1_posterior_loc_combiner,
                    m._level_1_posterior_log_scale_combiner):
            mod.weight.mul_(0.2)
    sampler = irec.BeamSearchCoder(kl_per_partition=3., n_beams=10, extra_samples=1., block_size=1000)
    image = torch.rand(128, 192, 3, device="cuda") - 0.5
    path = str(tmp_path / "kodak_crop.rec")
    seen = []                                     # what each of the two sequential sampler.encode calls was handed, and returned
    orig = sampler.encode

    def spy(target, coder, seed, **kw):
        out = orig(target, coder, seed, **kw)
        seen.append((target.loc.clone(), target.scale.clone(), coder.loc.clone(), coder.scale.clone(), out))
        return out
    sampler.encode = spy
    recon = m.compress(path, image, seed=42, sampler=sampler, block_size=1000, max_index=20)
    del sampler.encode
    from oracle import oracle as O
    assert [tuple(t[0].shape) for t in seen] == [(1, 2, 3, 128), (1, 8, 12, 196)]       # level 2 first, then level 1 (:359,394)
    for ql, qs, pl, ps, (idx, sample) in seen:                                            # in-situ parity with the oracle
        ridx, rs = O.encode_tensor(ql.cpu().numpy(), qs.cpu().numpy(), pl.cpu().numpy(), ps.cpu().numpy(), 42, 3.0, 20, 10,
                                   block_size=1000)
        assert idx == ridx and np.array_equal(sample.cpu().numpy(), rs)
    assert recon.shape == (1, 3, 128, 192) and torch.isfinite(recon).all()
    seed, shape, bs, block_indices = irec.io.read_compressed_code(path)
    assert (seed, shape, bs) == (42, (128, 192, 3), 1000)
    assert block_indices == [seen[0][4][0], seen[1][4][0]]                               # the file holds what the coder emitted
    assert [len(b) for b in block_indices] == [1, 19]
    assert m.level_1_posterior.loc.shape == (1, 8, 12, 196) and m.level_2_posterior.loc.shape == (1, 2, 3, 128)
    recon2 = m.decompress(path, sampler)
    assert torch.equal(recon2, recon)
    import os
    assert os.path.getsize(path) < 4000          # a few hundred indices, arithmetic coded


@pytest.mark.gpu
def test_batched_compress_equals_one_by_one_and_writes_rec(engine, oracle, tmp_path):
    """N images through every residual block together == the N = 1 path image by image (indices and reconstruction),
    each block's hand-off still the oracle's; the harness writes one .rec per image, reads it back and gathers the bits
    (compression_performance.py:350-375)."""
    import irec
    from irec import harness
    m = _model("cuda", blocks=4)
    torch.manual_seed(5)
    images = torch.rand(5, 3, 32, 32, device="cuda") - 0.5
    seen = []
    blk = m.residual_blocks[2]
    orig = blk.coder.encode

    def spy(target_dist, coding_dist, seed, **kw):
        out = orig(target_dist, coding_dist, seed, **kw)
        seen.append((target_dist.loc.clone(), target_dist.scale.clone(), coding_dist.loc.clone(), coding_dist.scale.clone(), out))
        return out
    blk.coder.encode = spy
    bi_batch, recon_batch = m.compress(images, seed=42)
    del blk.coder.encode
    assert len(bi_batch) == 5 and len(bi_batch[0]) == 4 and len(bi_batch[0][0]) == 3
    ql, qs, pl, ps, (pending, sample) = seen[0]
    assert ql.shape == (5, 16, 16, 8)
    lists = pending.to_lists()
    for i in (0, 3):                                                                   # in-situ oracle parity inside the batch
        ridx, rs = oracle.encode_tensor(ql[i].cpu().numpy(), qs[i].cpu().numpy(), pl[i].cpu().numpy(), ps[i].cpu().numpy(),
                                        42, 3.0, 36, 20, block_size=1000)
        assert lists[i] == ridx and np.array_equal(sample[i].cpu().numpy(), rs)
    for i in range(5):
        bi_one, recon_one = m.compress(images[i:i + 1], seed=42)
        assert bi_one == bi_batch[i]
        assert torch.allclose(recon_one[0], recon_batch[i], atol=1e-5, rtol=0)   # convolutions at batch 1 vs 5 may pick other kernels
    assert torch.allclose(m.decompress(bi_batch, seed=42, image_shape=images.shape), recon_batch, atol=1e-5, rtol=0)
    rows, all_bits, all_nats = harness.compress_sharded(m, images.cpu(), 42, 1000, str(tmp_path), batch=3)
    assert len(rows) == 5 and all(r["indices_recovered"] for r in rows)
    assert all_bits.tolist() == [r["comp_codelength"] for r in rows] and all(b > 0 for b in all_bits.tolist())
    s, shape, bs, bi_file = irec.io.read_compressed_code(str(tmp_path / "img_00003.rec"))
    assert (s, shape, bs) == (42, (32, 32, 3), 1000) and bi_file == bi_batch[3]
    assert rows[3]["n_indices"] == sum(len(ix) for b in bi_batch[3] for ix in b)
    # the per-image form of the driver (reference's write_compressed_code / read_compressed_code on nested lists): same rows
    rows_py = harness.compress_images(m, images, [f"py_{i}" for i in range(5)], 42, 1000, str(tmp_path), batch=3, packed=False)
    for a, b in zip(rows, rows_py):
        assert (a["comp_codelength"], a["n_indices"], a["indices_recovered"]) == (b["comp_codelength"], b["n_indices"], b["indices_recovered"])
    assert (tmp_path / "py_3.rec").read_bytes() == (tmp_path / "img_00003.rec").read_bytes()
    # the packed read-back itself
    K, idx, recon_p = m.compress_packed(images, seed=42)
    assert K.shape == (5, 4, 3) and torch.equal(recon_p, recon_batch)
    for i in range(5):
        assert [[idx[i, r, j, :K[i, r, j]].tolist() for j in range(3)] for r in range(4)] == bi_batch[i]


@pytest.mark.gpu
def test_handoff_kernels_against_plain_pytorch(engine):
    """csrc/irec_shim.hip (statistics + exp in NHWC, concat + ELU, residual + ELU, each with the producing convolution's bias
    folded in) against the plain PyTorch float32 ops they replace (rec/models/resnet_vae.py:385-496); tolerance 2e-6 relative:
    HIP's expf / expm1f and PyTorch's differ in the last place, everything else is the same IEEE operation."""
    from irec.models.resnet_vae import _HandOff
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    n, s, d, h, w = 3, 8, 20, 16, 16
    y = torch.randn((n, 4 * s + d, h, w), generator=g, device="cuda")
    yi = torch.randn((n, 2 * s + d, h, w), generator=g, device="cuda")
    by, bi = torch.randn(4 * s + d, generator=g, device="cuda"), torch.randn(2 * s + d, generator=g, device="cuda")
    lat = torch.randn((n, h, w, s), generator=g, device="cuda")
    yb, yib = y + by.reshape(1, -1, 1, 1), yi + bi.reshape(1, -1, 1, 1)

    def close(a, b):
        return torch.allclose(a, b, rtol=2e-6, atol=1e-7)
    want = yb[:, :4 * s].clone()
    want[:, 2 * s:] += yib[:, :2 * s]
    want = want.view(n, 4, s, h, w).permute(1, 0, 3, 4, 2).contiguous()
    want[1::2] = want[1::2].exp()
    got = _HandOff.stats(y, yi, s, 4, by, bi)
    assert got.shape == (4, n, h, w, s) and close(got, want)
    assert torch.equal(got[0], want[0]) and torch.equal(got[2], want[2])          # the locs are exact
    assert torch.equal(_HandOff.stats(y, None, s, 2, by), got[:2])                # the decoder's prior: the encoder's bits
    assert close(_HandOff.stats(y, yi, s, 4), (lambda t: torch.cat([t[:1], t[1:2].exp(), t[2:3], t[3:4].exp()]))(
        (torch.cat([y[:, :2 * s], y[:, 2 * s:4 * s] + yi[:, :2 * s]], 1)).view(n, 4, s, h, w).permute(1, 0, 3, 4, 2).contiguous()))
    want = F.elu(torch.cat([yb[:, 4 * s:], lat.permute(0, 3, 1, 2)], dim=1))
    assert close(_HandOff.cat_elu(y, 4 * s, d, lat, by), want)
    assert close(_HandOff.cat_elu(yi, 2 * s, d, None, bi), F.elu(yib[:, 2 * s:]))
    inp, t, bt = torch.randn((n, d, h, w), generator=g, device="cuda"), torch.randn((n, d, h, w), generator=g, device="cuda"), torch.randn(d, generator=g, device="cuda")
    out, out_elu = _HandOff.residual_elu(inp, t, 0.1, bt)
    want = torch.add(inp, t + bt.reshape(1, -1, 1, 1), alpha=0.1)
    assert close(out, want) and close(out_elu, F.elu(want))
    out2, _ = _HandOff.residual_elu(inp, t, 0.1)
    assert close(out2, torch.add(inp, t, alpha=0.1))


@pytest.mark.gpu
def test_graphed_compress_replays_the_whole_pass(engine):
    """The device side of compress (convolutions + 4 sequential coder launches, no host sync) captured as one HIP graph:
    same indices and reconstruction as the eager pass, for the capture image and for others replayed through it."""
    from irec.models import GraphedCompress
    m = _model("cuda", blocks=4)
    torch.manual_seed(9)
    images = torch.rand(3, 1, 3, 32, 32, device="cuda") - 0.5
    graphed = GraphedCompress(m, (1, 3, 32, 32), seed=42)
    for k in range(3):
        idx_e, rec_e = m.compress(images[k], seed=42)
        idx_g, rec_g = graphed(images[k])
        assert idx_g == idx_e and torch.equal(rec_g, rec_e), k
    assert graphed.graph is not None
    assert torch.equal(m.decompress(idx_g, seed=42, image_shape=images[2].shape), rec_g)


@pytest.mark.gpu
@pytest.mark.parametrize("lanes", [2, 3])
def test_graphed_compress_lanes_code_a_batch_side_by_side(engine, lanes):
    """A batch cut into independent sub-batches, each its own captured graph on its own stream with its own scratch (round
    3: a mid-size coder call is latency-bound, a second lane fills the device meanwhile): same indices and reconstruction,
    image for image, as the eager batched pass -- also for a second batch replayed through the same graphs."""
    from irec.models import GraphedCompress
    m = _model("cuda", blocks=4)
    torch.manual_seed(19)
    batches = torch.rand(2, 7, 3, 32, 32, device="cuda") - 0.5
    graphed = GraphedCompress(m, (7, 3, 32, 32), seed=42, lanes=lanes)
    assert graphed.lanes == lanes
    for k in range(2):
        idx_e, rec_e = m.compress(batches[k], seed=42)
        idx_g, rec_g = graphed(batches[k])
        assert idx_g == idx_e and torch.equal(rec_g, rec_e), k


@pytest.mark.gpu
def test_config3_harness_two_ranks_share_images(tmp_path):
    """scripts/config3_harness.py under torch.distributed.run with two ranks (gloo: both on the one GPU of the test box):
    image i goes to rank i mod 2, every rank compresses its share as one batch, writes / reads back its .rec files, and the
    per-image bits of ALL images are gathered on every rank."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(IREC_DIST_BACKEND="gloo", TMPDIR=str(tmp_path))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29617", os.path.join(root, "scripts", "config3_harness.py"),
                        "--images", "10", "--blocks", "3", "--singles", "1", "--no-graph"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["images_this_rank"] == 5 and res["gathered_items"] == 10 and res["all_indices_recovered"] and res["errors"] == 0
    assert res["mean_bits_per_image"] > 0
