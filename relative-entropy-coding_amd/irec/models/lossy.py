"""PyTorch-ROCm host shim of the reference's two-level lossy VAE call surface
(rec/models/lossy/large_2_level_vae.py:320-456): `call(tensor, sampling_fn)`, `compress(file_path, image, seed, sampler,
block_size, max_index)` and `decompress(file_path, sampler)` -- two SEQUENTIAL `sampler.encode` calls per image
(level 2, then level 1 whose prior is synthesised from the coded level-2 latent) followed by `write_compressed_code`.

As with the RVAE shim, only the hand-off is modelled: the Balle-style analysis / synthesis transforms are plain strided
(transposed) convolutions with random-init weights (GDN, SignalConv2D and the trained checkpoints are out of scope,
SURVEY.md §2 rows 12-14).  Latent shapes follow the reference: level 1 = [1, H/16, W/16, 196], level 2 = [1, H/64, W/64, 128]
(large_2_level_vae.py:313, compress_with_lossy_model.py:36-37), handed to the coder in NHWC order.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..io import read_compressed_code, write_compressed_code
from .resnet_vae import _Normal, _nchw, _nhwc, deterministic_transforms


def _down(cin, cout, n):
    layers, c = [], cin
    for i in range(n):
        layers += [nn.Conv2d(c, cout, 5, stride=2, padding=2)] + ([nn.ELU()] if i < n - 1 else [])
        c = cout
    return nn.Sequential(*layers)


def _up(cin, cmid, cout, n):
    layers, c = [], cin
    for i in range(n):
        last = i == n - 1
        layers += [nn.ConvTranspose2d(c, cout if last else cmid, 5, stride=2, padding=2, output_padding=1)] + \
                  ([] if last else [nn.ELU()])
        c = cmid
    return nn.Sequential(*layers)


class Large2LevelVAE(nn.Module):
    def __init__(self, level_1_filters=196, level_2_filters=128, name="large_level_2_vae", **kwargs):
        super().__init__()
        self.level_1_filters, self.level_2_filters = level_1_filters, level_2_filters
        f1, f2 = level_1_filters, level_2_filters
        self.analysis_transform = _down(3, 2 * f1, 4)                      # -> loc | log_scale, H/16
        self.hyper_analysis_transform = _down(f1, 2 * f2, 2)               # -> loc | log_scale, H/64
        self.hyper_synthesis_transform = _up(f2, f1, 2 * f1, 2)            # -> level-1 prior loc | log_scale
        self.synthesis_transform = _up(f1, f1, 3, 4)
        self._prior_base = nn.Parameter(torch.zeros(1, f2, 1, 1))
        self._prior_conv = nn.Conv2d(f2, f2, 3, padding=1)
        self._prior_loc_head = nn.Conv2d(f2, f2, 3, padding=1)
        self._prior_log_scale_head = nn.Conv2d(f2, f2, 3, padding=1)
        self._level_1_posterior_loc_combiner = nn.Conv2d(2 * f1, f1, 1)
        self._level_1_posterior_log_scale_combiner = nn.Conv2d(2 * f1, f1, 1)

    def prior_base(self, batch_size, height, width):
        """large_2_level_vae.py:312-313."""
        return self._prior_base.expand(batch_size, -1, height // 64, width // 64).contiguous()

    def _level_2_prior(self, batch_size, height, width):
        t = F.elu(self._prior_conv(self.prior_base(batch_size, height, width)))
        return self._prior_loc_head(t), F.softplus(self._prior_log_scale_head(t)) + 1e-7

    def _level_1_prior(self, level_2_latent):
        loc, log_scale = torch.chunk(self.hyper_synthesis_transform(level_2_latent), 2, dim=1)
        return loc, F.softplus(log_scale) + 1e-7, log_scale

    @torch.no_grad()
    def forward(self, tensor, sampling_fn=None):
        """large_2_level_vae.py:320-404.  tensor: [1, 3, H, W]; returns ([level_2_indices, level_1_indices], reconstruction)."""
        if sampling_fn is None:
            raise NotImplementedError("training / sampling passes are outside the compression shim")
        batch_size, _, height, width = tensor.shape
        l1_post_loc, l1_post_log_scale = torch.chunk(self.analysis_transform(tensor), 2, dim=1)
        l2_post_loc, l2_post_log_scale = torch.chunk(self.hyper_analysis_transform(l1_post_loc), 2, dim=1)
        self.level_2_posterior = _Normal(_nhwc(l2_post_loc), _nhwc(F.softplus(l2_post_log_scale) + 1e-7))
        l2_prior_loc, l2_prior_scale = self._level_2_prior(batch_size, height, width)
        self.level_2_prior = _Normal(_nhwc(l2_prior_loc), _nhwc(l2_prior_scale))
        level_2_indices, z = sampling_fn(target=self.level_2_posterior, coder=self.level_2_prior)            # :359-360
        l1_prior_loc, l1_prior_scale, l1_prior_log_scale = self._level_1_prior(_nchw(z))
        loc = self._level_1_posterior_loc_combiner(F.elu(torch.cat([l1_post_loc, l1_prior_loc], dim=1)))
        log_scale = self._level_1_posterior_log_scale_combiner(F.elu(torch.cat([l1_post_log_scale, l1_prior_log_scale], dim=1)))
        self.level_1_prior = _Normal(_nhwc(l1_prior_loc), _nhwc(l1_prior_scale))
        self.level_1_posterior = _Normal(_nhwc(loc), _nhwc(F.softplus(log_scale) + 1e-7))
        level_1_indices, y = sampling_fn(target=self.level_1_posterior, coder=self.level_1_prior)            # :394-395
        return [level_2_indices, level_1_indices], self.synthesis_transform(_nchw(y))

    def compress(self, file_path, image, seed, sampler, block_size, max_index):
        """large_2_level_vae.py:406-419.  image: [H, W, 3] tensor (the reference's layout)."""
        sampling_fn = lambda target, coder: sampler.encode(target, coder, seed=seed)  # noqa: E731  (:408)
        x = image.permute(2, 0, 1)[None].contiguous()
        with deterministic_transforms():
            block_indices, reconstruction = self(x, sampling_fn=sampling_fn)
        write_compressed_code(file_path=file_path, seed=seed, image_shape=tuple(image.shape), block_size=block_size,
                              block_indices=block_indices, max_index=max_index)
        return reconstruction

    @torch.no_grad()
    def decompress(self, file_path, sampler):
        """large_2_level_vae.py:421-456 (the reference unpacks image_shape as (batch, height, width); here (h, w, c))."""
        seed, image_shape, block_size, block_indices = read_compressed_code(file_path=file_path)
        height, width, _ = image_shape
        with deterministic_transforms():
            l2_prior_loc, l2_prior_scale = self._level_2_prior(1, height, width)
            self.level_2_prior = _Normal(_nhwc(l2_prior_loc), _nhwc(l2_prior_scale))
            z = sampler.decode(self.level_2_prior, seed=seed, indices=block_indices[0])                       # :441
            l1_prior_loc, l1_prior_scale, _ = self._level_1_prior(_nchw(z))
            self.level_1_prior = _Normal(_nhwc(l1_prior_loc), _nhwc(l1_prior_scale))
            y = sampler.decode(self.level_1_prior, seed=seed, indices=block_indices[1])                       # :451
            return self.synthesis_transform(_nchw(y))
