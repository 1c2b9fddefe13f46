"""Per-image compression driver: the loop of the reference's lossless evaluation script
(examples/lossless/compression_performance.py:305-430) around `model.compress` -- compress, write the `.rec` file
(:350-365), read it back and compare the indices (:369-375), bits / bits-per-pixel / bits-per-dimension of the code per
image (:367,380-384) -- with two changes the MI355X path is built around:

  * images are independent, so a rank compresses its images in BATCHES through `BidirectionalResNetVAE.compress`
    (one coder launch per residual block for the whole batch, one device-to-host copy per batch);
  * image i belongs to rank i mod G (irec/sharding.py); the per-image bits are gathered once at the end with the path's
    only collective (RCCL over xGMI on the GPU box, gloo in the CPU tests).

No dataset or checkpoint exists in the reference tree: the caller supplies images and a model; the shim
models have no likelihood head, so "bpd" here is the CODE's share, file bits / (pixels * channels), not the reference's
code + residual figure.
"""
import os
import time

import numpy as np
import torch

from . import sharding
from .coding import CodingError
from .io import decode_files, encode_files, read_compressed_code, write_compressed_code


def _host_leg(chunk_shape, names, seed, block_size, out_dir, S, K, idx, t_compress):
    """Write / read back / compare the .rec files of one batch from its packed read-back (compression_performance.py:350-375
    per image): the containers are built natively for all images at once (irec_rec_encode_files, host threads), written, read
    back, decoded together (irec_rec_decode_files) and compared as arrays."""
    n, _, h, w = chunk_shape
    t1 = time.perf_counter()
    blob, off = encode_files(seed, (h, w, 3), block_size, K, idx, max_index=S)      # the reference passes 20 < S = 36
    paths = [os.path.join(out_dir, f"{nm}.rec") for nm in names]
    mv = memoryview(blob)
    for i, path in enumerate(paths):
        with open(path, "wb") as fh:
            fh.write(mv[off[i]:off[i + 1]])
    back = [open(path, "rb").read() for path in paths]
    sizes = np.array([len(b) for b in back], dtype=np.int64)
    off2 = np.concatenate([[0], np.cumsum(sizes)])
    hdr, K2, idx2 = decode_files(np.frombuffer(b"".join(back), dtype=np.uint8), off2, K.shape[1], K.shape[2], idx.shape[3])
    live = np.arange(idx.shape[3])[None, None, None, :] < K[..., None]
    same = (K2 == K).all(axis=(1, 2)) & ((idx2 == idx) | ~live).all(axis=(1, 2, 3)) & \
        (hdr[:, [0, 1, 3, 4, 5]] == np.array([seed, block_size, h, w, 3], dtype=np.uint32)).all(axis=1)
    n_idx = K.sum(axis=(1, 2))
    t_host = (time.perf_counter() - t1) / n
    return [{"name": names[i], "comp_codelength": int(sizes[i]) * 8, "comp_lossy_bpp": int(sizes[i]) * 8 / (h * w),
             "comp_code_bpd": int(sizes[i]) * 8 / (h * w * 3), "code_nats": int(n_idx[i]) * float(np.log(S)),
             "n_indices": int(n_idx[i]), "indices_recovered": bool(same[i]), "comp_time": t_compress / n + t_host}
            for i in range(n)]


def compress_images(model, images, names, seed, block_size, out_dir, batch=None, packed=True):
    """images: [n, 3, H, W] in [-0.5, 0.5] on the model's device.  Returns one dict per image (reference CSV columns where
    they apply: comp_codelength, comp_lossy_bpp, comp_time) plus `indices_recovered`, `code_nats`.
    packed (default): the indices stay packed arrays from the device to the files (model.compress_packed, irec.io.encode_files
    / decode_files), and the host leg of a batch runs on a worker thread while the device codes the next batch; packed=False is
    the per-image form with the reference's write_compressed_code / read_compressed_code on nested lists."""
    os.makedirs(out_dir, exist_ok=True)
    n = images.shape[0]
    batch = n if not batch else int(batch)
    S = model.residual_blocks[0].coder.n_samples
    rows = []
    if packed and hasattr(model, "compress_packed"):
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=1) as pool:
            jobs = []
            for lo in range(0, n, batch):
                chunk = images[lo:lo + batch]
                t0 = time.perf_counter()
                try:
                    K, idx, _ = model.compress_packed(chunk, seed=seed, update_sampler=False)
                except CodingError as e:                 # compression_performance.py:375-377: log and move on
                    jobs.append([{"name": names[lo + i], "error": str(e)} for i in range(chunk.shape[0])])
                    continue
                jobs.append(pool.submit(_host_leg, tuple(chunk.shape), names[lo:lo + chunk.shape[0]], seed, block_size, out_dir,
                                        S, K, idx, time.perf_counter() - t0))
            for j in jobs:
                rows += j if isinstance(j, list) else j.result()
        return rows
    for lo in range(0, n, batch):
        chunk = images[lo:lo + batch]
        t0 = time.perf_counter()
        try:
            block_indices, _ = model.compress(chunk, seed=seed, update_sampler=False)
        except CodingError as e:                     # compression_performance.py:375-377: log and move on
            rows += [{"name": names[lo + i], "error": str(e)} for i in range(chunk.shape[0])]
            continue
        t_compress = time.perf_counter() - t0
        per_image = [block_indices] if chunk.shape[0] == 1 else block_indices
        for i, bi in enumerate(per_image):
            t1 = time.perf_counter()
            _, _, h, w = chunk.shape
            path = os.path.join(out_dir, f"{names[lo + i]}.rec")
            write_compressed_code(file_path=path, seed=seed, image_shape=(h, w, 3), block_size=block_size,
                                  block_indices=bi, max_index=S)       # the reference passes 20 < S = 36
            bits = os.path.getsize(path) * 8
            s, shape, bs, bi_ = read_compressed_code(file_path=path)
            ok = (s, tuple(shape), bs) == (seed, (h, w, 3), block_size) and bi_ == bi
            n_idx = sum(len(ix) for blk in bi for ix in blk)
            rows.append({"name": names[lo + i], "comp_codelength": bits, "comp_lossy_bpp": bits / (h * w),
                         "comp_code_bpd": bits / (h * w * 3), "code_nats": n_idx * float(np.log(S)), "n_indices": n_idx,
                         "indices_recovered": bool(ok),
                         "comp_time": t_compress / chunk.shape[0] + (time.perf_counter() - t1)})
    return rows


def compress_sharded(model, all_images, seed, block_size, out_dir, rank=0, world=1, dist=None, batch=None):
    """Config 3 (300 images over G GPUs): this rank compresses images rank, rank + G, ...; every rank gets the [n_images]
    vectors of file bits and code nats back (one all_gather each, <= 38 floats per rank: latency only)."""
    n_items = all_images.shape[0]
    mine = sharding.shard_indices(n_items, rank, world)
    names = [f"img_{int(i):05d}" for i in mine]
    dev = next(model.parameters()).device
    rows = compress_images(model, all_images[torch.as_tensor(mine)].to(dev), names, seed, block_size, out_dir, batch)
    coll_dev = dev if (dist is not None and dist.get_backend() == "nccl") else torch.device("cpu")
    bits = torch.tensor([r.get("comp_codelength", -1) for r in rows], dtype=torch.float64, device=coll_dev)
    nats = torch.tensor([r.get("code_nats", -1.0) for r in rows], dtype=torch.float64, device=coll_dev)
    all_bits = sharding.gather_per_item(bits, n_items, rank, world, dist)
    all_nats = sharding.gather_per_item(nats, n_items, rank, world, dist)
    return rows, all_bits, all_nats
