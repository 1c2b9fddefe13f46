"""PyTorch-ROCm host shim of the reference's bidirectional ResNet VAE (rec/models/resnet_vae.py) -- ONLY what the
compression path touches: the block / model constructors with the reference's keyword surface, the inference pass,
and the sequential generative pass that hands each residual block's posterior and prior to `coder.encode`
(resnet_vae.py:462-476, 803-836) or `coder.decode`.

Not a re-implementation of the model family: the convolutions are stock `torch.nn.Conv2d` (the reference's
weight-normalised `ReparameterizedConv2D` with data-dependent init, IAF posteriors, likelihoods, EMA and training are
out of scope -- no checkpoints or datasets exist in the reference tree), so weights are random-init.  What
the shim pins is the hand-off: tensors are presented to the coder in the reference's NHWC order, the residual blocks
are coded strictly in sequence (the prior of block n+1 depends on the coded latent of block n), and the keyword
arguments (`sampler`, `sampler_args`, `coder_args`, `kl_per_partition`, `encoder_args`, `decoder_args`) keep their
names and meaning.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..coding import BeamSearchCoder
from ..coding.beam_search_coder import MorePartitionsNeeded, PendingCode, SplitNotResident


class ModelError(Exception):
    """rec/models/resnet_vae.py (ModelError)."""


class deterministic_transforms:
    """Encoder and decoder must evaluate the SAME bits for every prior: a relative-entropy code is only decodable if the
    decoder's coding distribution equals the encoder's exactly.  MIOpen may pick different (or atomically accumulating)
    convolution algorithms from call to call, so compress / decompress pin the deterministic ones for their duration."""

    def __enter__(self):
        self._prev = (torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark)
        torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark = True, False
        return self

    def __exit__(self, *exc):
        torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark = self._prev
        return False


class _HandOff:
    """The elementwise hand-offs between the shim's convolutions and the coder as ONE launch each (csrc/irec_shim.hip:
    irec_shim_stats / _cat_elu / _residual_elu) instead of ~10 PyTorch launches of 3-5 us per residual block and pass.
    CUDA float32 contiguous tensors only; anything else takes the plain PyTorch ops (the coder itself has no CPU path)."""

    @staticmethod
    def ok(*tensors):
        return all(t is not None and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in tensors)

    @staticmethod
    def _eng(t):
        from ..engine import get_engine
        return get_engine(t.device)

    @staticmethod
    def _p(t):
        import ctypes
        return ctypes.c_void_p(t.data_ptr() if t is not None else 0)

    @staticmethod
    def stats(y, infer_y, s, n_stats, bias_y=None, bias_infer=None):
        """[n_stats, N, H, W, s]: prior loc, prior scale (, posterior loc, posterior scale) in the coder's NHWC order.
        bias_*: the (not yet added) biases of the convolutions behind y / infer_y."""
        from .. import _lib
        n, cy, h, w = y.shape
        out = torch.empty((n_stats, n, h, w, s), dtype=torch.float32, device=y.device)
        eng = _HandOff._eng(y)
        _lib.check(eng.lib.irec_shim_stats(eng.ctx, _HandOff._p(y), _HandOff._p(infer_y), _HandOff._p(out), n_stats, n, cy,
                                           infer_y.shape[1] if infer_y is not None else 0, s, h * w, _HandOff._p(bias_y),
                                           _HandOff._p(bias_infer), eng._stream()), "irec_shim_stats")
        return out

    @staticmethod
    def cat_elu(y, c_off, d, latent_nhwc, bias_y=None):
        """elu(cat(y[:, c_off:c_off + d] (+ bias), latent NHWC -> NCHW)); latent None: the ELU of the channel slice."""
        from .. import _lib
        n, cy, h, w = y.shape
        s = 0 if latent_nhwc is None else latent_nhwc.shape[-1]
        out = torch.empty((n, d + s, h, w), dtype=torch.float32, device=y.device)
        eng = _HandOff._eng(y)
        _lib.check(eng.lib.irec_shim_cat_elu(eng.ctx, _HandOff._p(y), _HandOff._p(latent_nhwc if s else None), _HandOff._p(out),
                                             n, cy, c_off, d, s, h * w, _HandOff._p(bias_y), eng._stream()), "irec_shim_cat_elu")
        return out

    @staticmethod
    def residual_elu(inp, t, alpha, bias_t=None):
        """(inp + alpha * (t + bias), elu of it)"""
        from .. import _lib
        out, out_elu = torch.empty_like(inp), torch.empty_like(inp)
        n, c, h, w = inp.shape
        eng = _HandOff._eng(inp)
        _lib.check(eng.lib.irec_shim_residual_elu(eng.ctx, _HandOff._p(inp), _HandOff._p(t), float(alpha), _HandOff._p(out),
                                                  _HandOff._p(out_elu), n, c, h * w, _HandOff._p(bias_t), eng._stream()),
                   "irec_shim_residual_elu")
        return out, out_elu


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def _nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


class _Normal:
    """Duck-typed distribution: the coder reads only .loc and .scale (coder.py:427-430)."""

    def __init__(self, loc, scale):
        self.loc, self.scale = loc, scale


class BidirectionalResidualBlock(nn.Module):
    """resnet_vae.py:20-497 (compression-relevant subset)."""

    def __init__(self, stochastic_filters, deterministic_filters, sampler, sampler_args={}, coder_args={},
                 distribution="gaussian", kernel_size=(3, 3), use_iaf=False, is_last=False, kl_per_partition=8.,
                 use_sig_convs=False, name="bidirectional_resnet_block", **kwargs):
        super().__init__()
        if distribution != "gaussian":
            raise ValueError("Distribution must be 'gaussian' on the beam-search path, "
                             f"but {distribution} was given!")
        if use_iaf or use_sig_convs:
            raise ModelError("IAF posteriors / SignalConv2D are outside the compression shim")
        self.name = name
        self.stochastic_filters, self.deterministic_filters = stochastic_filters, deterministic_filters
        self.is_last = is_last
        pad = (kernel_size[0] // 2, kernel_size[1] // 2)

        def conv(cin, cout):
            return nn.Conv2d(cin, cout, kernel_size, padding=pad)

        d, s = deterministic_filters, stochastic_filters
        if not is_last:
            self.infer_conv1, self.infer_conv2 = conv(d, d), conv(d, d)
        self.infer_posterior_loc_head, self.infer_posterior_log_scale_head = conv(d, s), conv(d, s)
        self.gen_conv1, self.gen_conv2 = conv(d, d), conv(d + s, d)
        self.prior_loc_head, self.prior_log_scale_head = conv(d, s), conv(d, s)
        self.gen_posterior_loc_head, self.gen_posterior_log_scale_head = conv(d, s), conv(d, s)
        self._infer_y = self._infer_bias = self._infer_heads = None   # inference pass's head convolution (and its pending bias)
        # Hand-off kernels (csrc/irec_shim.hip) or plain torch ops between the convolutions and the coder: decided ONCE per block --
        # None: by where the block lives (a float32 CUDA block takes the kernels) -- and identically in compress and decompress,
        # never per tensor: the kernels' expf / expm1f and torch.exp / F.elu differ in the last ulp, so an encoder and a decoder
        # that chose differently would rebuild different prior scales and the reconstruction would silently diverge.  Inputs
        # the kernels cannot take as they are (non-contiguous, channels_last) are made contiguous, not routed around them.
        self.use_handoff_kernels = None
        self.posterior = self.prior = None
        self._pad = pad
        self._fused = None   # (parameter versions, inference-side weight / bias, generative-side weight / bias)
        self._fuse_mods = None
        # ---- stuff for compression (resnet_vae.py:118-141) ----
        if sampler == "beam_search":
            self.coder = BeamSearchCoder(kl_per_partition=kl_per_partition, n_beams=sampler_args['n_beams'],
                                         extra_samples=sampler_args['extra_samples'], name=f"encoder_for_{self.name}",
                                         **coder_args)
        elif sampler in ("rejection", "importance"):
            raise ModelError(f"sampler '{sampler}' is outside the beam-search path of this build")
        else:
            raise ModelError("Sampler must be one of ['rejection', 'importance', 'beam_search'],"
                             f"but got {sampler}!")

    @property
    def infer_posterior_loc(self):
        """Inference-side posterior loc head (resnet_vae.py: infer_posterior_loc; 0. before an inference pass, as there).  On the
        kernel path the head convolution runs without its bias (the hand-off kernel adds it): the view is biased lazily."""
        return self._infer_head(0)

    @property
    def infer_posterior_log_scale(self):
        return self._infer_head(1)

    def _infer_head(self, k):
        if self._infer_y is None:
            return 0.
        s = self.stochastic_filters
        v = self._infer_y[:, k * s:(k + 1) * s]
        return v if self._infer_bias is None else v + self._infer_bias[k * s:(k + 1) * s].reshape(1, -1, 1, 1)

    def _handoff(self, *tensors):
        """Whether this block's passes take the hand-off kernels: the explicit flag, else float32-on-CUDA of the block ITSELF -- decided
        by the block's own parameters, identically in compress and decompress, never by the activations of one call (round 5: a
        decompress whose activations arrived in another dtype or on another device used to take the torch path silently, and
        torch.exp / F.elu differ from the kernels' expf / expm1f in the last ulp: the prior scales, and with them the reconstruction,
        diverged).  Activations that disagree with the block are an error."""
        w = self.gen_conv1.weight
        use = bool(self.use_handoff_kernels) if self.use_handoff_kernels is not None else (w.is_cuda and w.dtype == torch.float32)
        if use and not all(t is None or (t.is_cuda and t.dtype == torch.float32 and t.device == w.device) for t in tensors):
            raise ModelError("this residual block runs its hand-off kernels (float32 parameters on a HIP device): its activations must be "
                             "float32 on that device too, in compress and decompress alike")
        return use

    @property
    def posterior_loc(self):
        """resnet_vae.py: infer_posterior_loc + gen_posterior_loc (the sum is formed in place in `forward`)."""
        return _nchw(self.posterior.loc)

    @property
    def posterior_scale(self):
        return _nchw(self.posterior.scale)

    def _fused_weights(self):
        """The convolutions that read the same activation, as ONE convolution each (output channels concatenated):
        inference side = posterior heads (loc, log-scale) + infer_conv1; generative side = prior heads, posterior heads
        + gen_conv1.  Ten convolutions per residual block become four -- on a 16x16 latent grid each one is a
        launch-bound im2col + GEMM pair, and a single image spends more time in them than in the coder.  The compress and
        the decompress pass both run the SAME generative-side convolution (the decoder drops the posterior channels), so
        the prior the decoder rebuilds is bit-identical to the one the encoder coded against.  Rebuilt when a parameter
        changes in place, is replaced or moves (tensor version counters and storage addresses)."""
        mods = self._fuse_mods
        if mods is None:   # (looked up once: nn.Module attribute access is a microsecond each, and this runs 48 times per image)
            heads_i = [self.infer_posterior_loc_head, self.infer_posterior_log_scale_head] + ([] if self.is_last else [self.infer_conv1])
            heads_g = [self.prior_loc_head, self.prior_log_scale_head, self.gen_posterior_loc_head,
                       self.gen_posterior_log_scale_head, self.gen_conv1]
            mods = self._fuse_mods = (heads_i, heads_g)
        heads_i, heads_g = mods
        ver = [x for m in heads_i + heads_g for p in (m._parameters["weight"], m._parameters["bias"])
               for x in (id(p), p._version, p.data_ptr())]
        if self._fused is None or self._fused[0] != ver:
            with torch.no_grad():
                self._fused = (ver,
                               torch.cat([m.weight for m in heads_i]).contiguous(), torch.cat([m.bias for m in heads_i]).contiguous(),
                               torch.cat([m.weight for m in heads_g]).contiguous(), torch.cat([m.bias for m in heads_g]).contiguous())
        return self._fused[1:]

    def forward(self, tensor, inference_pass=True, encoder_args=None, decoder_args=None):
        """resnet_vae.py:372-497."""
        inp = tensor
        pre = getattr(tensor, "_irec_elu", None)          # the previous block's residual kernel already formed elu(tensor)
        tensor = pre if pre is not None else F.elu(tensor)
        indices = None
        s, d = self.stochastic_filters, self.deterministic_filters
        w_i, b_i, w_g, b_g = self._fused_weights()
        # On the fused path the convolutions run WITHOUT their bias: the hand-off kernel that reads a convolution's output adds
        # it first (same operation, same order, one launch fewer per convolution: 96 per image).
        fused = self._handoff(inp, tensor)                 # one decision per block: the same in compress and decompress
        if fused:
            inp, tensor = inp.contiguous(), tensor.contiguous()
        bias2 = None
        if inference_pass:
            y = F.conv2d(tensor, w_i, None if fused else b_i, padding=self._pad)   # [N, 2s (+ d), H, W]
            if fused:
                y = y.contiguous()
            self._infer_y, self._infer_bias = y, (b_i if fused else None)
            self._infer_heads = None if fused else y[:, :2 * s]     # (infer_posterior_loc / _log_scale: lazily biased views)
            if not self.is_last:
                if fused:
                    tensor = F.conv2d(_HandOff.cat_elu(y, 2 * s, d, None, b_i), self.infer_conv2.weight, None, padding=self._pad)
                    bias2 = self.infer_conv2.bias
                else:
                    tensor = self.infer_conv2(F.elu(y[:, 2 * s:]))
        else:
            if encoder_args is None and decoder_args is None:
                raise ModelError("training / sampling passes are outside the compression shim")
            if encoder_args is not None and self._infer_y is None:
                raise ModelError(f"{self.name}: a compression pass needs the statistics of an inference pass over the same input first")
            if encoder_args is not None and fused != (self._infer_bias is not None):
                raise ModelError(f"{self.name}: use_handoff_kernels changed between the inference and the generative pass")
            y = F.conv2d(tensor, w_g, None if fused else b_g, padding=self._pad)  # [N, 4s + d, H, W]
            if fused:
                y = y.contiguous()
            n, _, h, w = y.shape
            if encoder_args is not None:                                          # :462-470
                if fused:
                    st = _HandOff.stats(y, self._infer_y, s, 4, b_g, self._infer_bias)   # all four statistics, NHWC, one launch
                else:
                    y[:, 2 * s:4 * s] += self._infer_heads                                   # posterior loc / log-scale = inference + generative
                    st = y[:, :4 * s].view(n, 4, s, h, w).permute(1, 0, 3, 4, 2).contiguous()   # coder sees NHWC, as in the reference
                    st[1::2].exp_()                                               # the two scales
                self.prior, self.posterior = _Normal(st[0], st[1]), _Normal(st[2], st[3])
                indices, latent_code = self.coder.encode(self.posterior, self.prior, **encoder_args)
            else:                                                                 # :475-476
                if fused:
                    st = _HandOff.stats(y, None, s, 2, b_g)                       # the SAME kernel and exp as the encoder's prior
                else:
                    st = y[:, :2 * s].view(n, 2, s, h, w).permute(1, 0, 3, 4, 2).contiguous()
                    st[1].exp_()
                self.prior = _Normal(st[0], st[1])
                latent_code = self.coder.decode(self.prior, **decoder_args)
            if fused:
                latent_code = latent_code.to(dtype=torch.float32).contiguous()
                tensor = F.conv2d(_HandOff.cat_elu(y, 4 * s, d, latent_code, b_g), self.gen_conv2.weight, None, padding=self._pad)
                bias2 = self.gen_conv2.bias
            else:
                tensor = torch.cat([y[:, 4 * s:], latent_code.permute(0, 3, 1, 2)], dim=1)
                tensor = self.gen_conv2(F.elu(tensor, inplace=True))
        if fused:
            tensor, tensor_elu = _HandOff.residual_elu(inp, tensor.contiguous(), 0.1, bias2)
            tensor._irec_elu = tensor_elu                  # (a Python attribute: the next block, or _finish, picks it up)
        else:
            tensor = torch.add(inp, tensor, alpha=0.1)
        if encoder_args is not None:
            return indices, tensor
        return tensor


class BidirectionalResNetVAE(nn.Module):
    """resnet_vae.py:512-860 (compression-relevant subset)."""

    def __init__(self, num_res_blocks, sampler, sampler_args={}, coder_args={}, likelihood_function="discretized_logistic",
                 learn_likelihood_scale=True, first_kernel_size=(5, 5), first_strides=(2, 2), kernel_size=(3, 3),
                 strides=(1, 1), deterministic_filters=160, stochastic_filters=32, use_iaf=False, kl_per_partition=8.,
                 latent_size="variable", ema_decay=0.999, name="resnet_vae", **kwargs):
        super().__init__()
        self.sampler_name = str(sampler)
        self.num_res_blocks = num_res_blocks
        self.deterministic_filters, self.stochastic_filters = deterministic_filters, stochastic_filters
        self.kl_per_partition = kl_per_partition
        pad = (first_kernel_size[0] // 2, first_kernel_size[1] // 2)
        self.first_infer_conv = nn.Conv2d(3, deterministic_filters, first_kernel_size, stride=first_strides, padding=pad)
        self.last_gen_conv = nn.ConvTranspose2d(deterministic_filters, 3, first_kernel_size, stride=first_strides,
                                                padding=pad, output_padding=(first_strides[0] - 1, first_strides[1] - 1))
        self.residual_blocks = nn.ModuleList([
            BidirectionalResidualBlock(stochastic_filters=stochastic_filters, deterministic_filters=deterministic_filters,
                                       sampler=self.sampler_name, sampler_args=sampler_args, coder_args=coder_args,
                                       kernel_size=kernel_size, is_last=res_block_idx == 0, use_iaf=use_iaf,
                                       kl_per_partition=kl_per_partition, name=f"resnet_block_{res_block_idx}")
            for res_block_idx in range(num_res_blocks)])
        self._generative_base = nn.Parameter(torch.zeros(deterministic_filters))

    def generative_base(self, batch_size, height, width):
        """resnet_vae.py:619-623."""
        return self._generative_base.reshape(1, -1, 1, 1).expand(batch_size, -1, height // 2, width // 2).contiguous()

    def _finish(self, tensor):
        pre = getattr(tensor, "_irec_elu", None)
        reconstruction = self.last_gen_conv(pre if pre is not None else F.elu(tensor))
        return torch.clamp(reconstruction, -0.5 + 1. / 512., 0.5 - 1. / 512.)

    @torch.no_grad()
    def compress(self, image, seed, update_sampler=False):
        """resnet_vae.py:803-836.  image: [N, 3, H, W] in [-0.5, 0.5].  Returns (block_indices, reconstruction).

        N = 1 (the reference's only case): block_indices[res_block][coder_block] = list of indices, as the reference returns.
        N > 1 (extension): the N images go through every residual block together -- one coder launch per residual block
        for all of them -- and block_indices[image][res_block][coder_block].
        Nothing is copied to the host until the last residual block has been coded: each block's `coder.encode(...,
        defer=True)` leaves its K / index rows on the device and hands the merged sample straight to the next block's
        convolutions; ONE device-to-host copy then fetches all indices of all blocks and images."""
        for _attempt in range(6):
            pendings, reconstruction = self._compress_device(image, seed, update_sampler)
            try:
                per_block = PendingCode.gather(pendings)     # [res_block][image][coder_block]; the only host sync
                break
            except (MorePartitionsNeeded, SplitNotResident):
                continue   # some block's KL needs more index slots than the coders' hint (the hints are raised), or the split
                           # encoder's partner workgroups were not resident (every coder's next call goes out unshared): code again
        else:
            raise MorePartitionsNeeded(max(b.coder._max_K_hint for b in self.residual_blocks) + 1)
        return self._indices_structure(per_block, image.shape[0]), reconstruction

    @torch.no_grad()
    def compress_packed(self, image, seed, update_sampler=False):
        """`compress` for a batch, with the indices left packed: (K [N, R, bpt], idx [N, R, bpt, max_K] int32 numpy arrays,
        reconstruction) -- what irec.io.encode_files turns into N .rec files without one Python object per index.  Needs every
        residual block to share one block layout (they do: same latent shape and block_size)."""
        for _attempt in range(6):
            pendings, reconstruction = self._compress_device(image, seed, update_sampler)
            try:
                K, idx = PendingCode.gather_packed(pendings)
                return K, idx, reconstruction
            except (MorePartitionsNeeded, SplitNotResident):
                continue
        raise MorePartitionsNeeded(max(b.coder._max_K_hint for b in self.residual_blocks) + 1)

    def _compress_device(self, image, seed, update_sampler=False):
        """Everything of `compress` that runs on the device, with no host synchronisation: (PendingCode per residual
        block, reconstruction).  Capturable in a HIP graph (GraphedCompress)."""
        batch_size, _, height, width = image.shape
        with deterministic_transforms():
            tensor = self.first_infer_conv(image)
            for resnet_block in list(self.residual_blocks)[::-1]:         # inference pass, reverse order (:811-813)
                tensor = resnet_block(tensor, inference_pass=True)
            tensor = self.generative_base(batch_size=batch_size, width=width, height=height)
            pendings = []
            # every residual block codes with the same seed, S and block dims (:822-824): with ONE table window for all of them
            # the proposal tables of the first block serve the other 23 (IREC_FLAG_REUSE_TABLES)
            # (one max_K too: it bounds the window, and a window that covers max_K leaves no second pass to launch -- built
            #  once per image, a long window costs next to nothing)
            max_K = max(blk.coder._max_K_hint for blk in self.residual_blocks)
            window = max(max(blk.coder.table_window() for blk in self.residual_blocks), min(max_K, 64))
            from ..engine import get_engine
            with get_engine(tensor.device).table_session():              # twin calls back to back: no table launches after the first
                for resnet_block in self.residual_blocks:                 # strictly sequential (:821-826)
                    pending, tensor = resnet_block(tensor, inference_pass=False,
                                                   encoder_args={"seed": seed, "update_sampler": update_sampler, "batched": True,
                                                                 "defer": True, "table_steps": window, "max_K": max_K})
                    pendings.append(pending)
            return pendings, self._finish(tensor)

    def _indices_structure(self, per_block, batch_size):
        flat = [blk.coder.block_size is None for blk in self.residual_blocks]   # no block_size: one index list per tensor
        per_block = [[img[0] if flat[r] else img for img in blk] for r, blk in enumerate(per_block)]
        if batch_size == 1:
            return [blk[0] for blk in per_block]
        return [[blk[i] for blk in per_block] for i in range(batch_size)]

    @torch.no_grad()
    def decompress(self, block_indices, seed, image_shape):
        """The generative pass driven by the stored indices.  (The reference's own decompress, resnet_vae.py:844-860,
        is an unfinished stub; this is the pass its decoder_args plumbing implies.)  image_shape[0] = N > 1 takes the
        batched form of `compress`'s block_indices."""
        batch_size, _, height, width = image_shape
        with deterministic_transforms():
            tensor = self.generative_base(batch_size=batch_size, width=width, height=height)
            for r, resnet_block in enumerate(self.residual_blocks):
                if batch_size == 1:
                    args = {"seed": seed, "indices": block_indices[r]}
                else:
                    args = {"seed": seed, "indices": [block_indices[i][r] for i in range(batch_size)], "batched": True}
                tensor = resnet_block(tensor, inference_pass=False, decoder_args=args)
            return self._finish(tensor)


class GraphedCompress:
    """`model.compress` for a fixed image shape and seed as ONE HIP graph: the first call captures the whole device side
    of the pass -- every convolution, elementwise op and coder launch of the 24 strictly sequential residual blocks --
    and later calls replay it (one graph launch instead of several hundred kernel launches; the coder's entry points are
    asynchronous and allocation-free, so they capture like any other kernel).  Inputs are copied into the graph's static
    image buffer; indices are read back with the usual single device-to-host copy.  Same outputs as `model.compress`.
    A block that needs more partitions than the captured index buffers hold falls back to the eager path and re-captures.

    lanes > 1 (round 3; off by default, see below): the batch is cut into `lanes` independent sub-batches, each captured
    as its own graph on its own stream and replayed side by side.  Images are independent (compression_performance.py:305 is
    a plain loop), only the 24 residual blocks of ONE image are sequential (resnet_vae.py:821-826): while one lane waits for
    the single-wave tail of its coder call -- a mid-size call is one or two blocks per CU and lasts as long as its slowest
    block's K sequential steps -- the other lane's convolutions and coder fill the device.  Each lane has its own scratch
    (the engine keys it by stream), its own static buffers and index rows; the indices are the same, image for image."""

    def __init__(self, model, image_shape, seed, update_sampler=False, lanes=None):
        self.model, self.seed, self.update_sampler = model, seed, update_sampler
        self.device = next(model.parameters()).device
        n = int(image_shape[0])
        if lanes is None:
            lanes = 1   # (measured r03d: two lanes do not overlap -- the convolutions' 64 KB and the coder's 160 KB of LDS per workgroup cannot share a CU)
        self.lanes = max(1, min(int(lanes), n))
        cuts = [n * k // self.lanes for k in range(self.lanes + 1)]
        self.slices = [slice(cuts[k], cuts[k + 1]) for k in range(self.lanes)]
        self.static_images = [torch.zeros((sl.stop - sl.start,) + tuple(image_shape[1:]), device=self.device) for sl in self.slices]
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(self.lanes)]
        self.graphs = None

    @property
    def graph(self):                        # (kept for callers that test "has it been captured")
        return self.graphs

    @graph.setter
    def graph(self, value):
        self.graphs = value

    @torch.no_grad()
    def _capture(self):
        cur = torch.cuda.current_stream(self.device)
        self.graphs, self.pendings, self.reconstructions = [], [], []
        for lane, st in enumerate(self.streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):                                # warm-up on the lane's stream: caches, ITS scratch, MIOpen
                for _ in range(2):
                    self.model._compress_device(self.static_images[lane], self.seed, self.update_sampler)
            cur.wait_stream(st)
        torch.cuda.synchronize(self.device)
        for lane, st in enumerate(self.streams):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                pend, rec = self.model._compress_device(self.static_images[lane], self.seed, self.update_sampler)
            self.graphs.append(g); self.pendings.append(pend); self.reconstructions.append(rec)

    @torch.no_grad()
    def __call__(self, image):
        for lane, sl in enumerate(self.slices):
            self.static_images[lane].copy_(image[sl])
        if self.graphs is None:
            if any(getattr(b.coder, "_split_pause", 0) for b in self.model.residual_blocks):
                # a coder is stepping back from a give-up of its cooperative encoder: eager until its pause is over -- a graph
                # captured now would keep the pause's IREC_FLAG_NO_SPLIT for good
                return self.model.compress(image, seed=self.seed, update_sampler=self.update_sampler)
            self._capture()
        cur = torch.cuda.current_stream(self.device)
        for lane, st in enumerate(self.streams):                       # the lanes' graphs side by side, each on its own stream
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                self.graphs[lane].replay()
        for st in self.streams:
            cur.wait_stream(st)
        try:
            flat = PendingCode.gather([p for lane in self.pendings for p in lane])   # ONE device-to-host copy for all lanes
        except (MorePartitionsNeeded, SplitNotResident):
            self.graphs = None                                         # hints were raised / the coders step back from sharing: eager now, re-capture later
            return self.model.compress(image, seed=self.seed, update_sampler=self.update_sampler)
        n_res = len(self.pendings[0])
        per_block = [[img for lane in range(self.lanes) for img in flat[lane * n_res + r]] for r in range(n_res)]
        rec = self.reconstructions[0].clone() if self.lanes == 1 else torch.cat(self.reconstructions, dim=0)
        return self.model._indices_structure(per_block, image.shape[0]), rec
