"""Reference-shaped CPU baseline.  TEST/BENCH INFRASTRUCTURE ONLY (never imported by the product).

An op-for-op PyTorch-CPU restatement of rec/coding/beam_search_coder.py:37-106 that materialises the same
[S, B, 1, D] intermediates in the same order as the TensorFlow-eager reference (quantile on every element, two
log_prob passes, reduce_sum, argsort, gather).  It stands in for the reference's TF-2.1-CPU path, which cannot be
installed here: torch eager has lower dispatch overhead than TF 2.1 eager and a
single-branch ndtri, so speed-ups quoted against it are conservative.  The int32 draws come from the C oracle's
Philox restatement; float32 rounding differs from the oracle's canonical mode (different reduction order), so this
file is a TIMING baseline, not a parity oracle.
"""
import math

import numpy as np
import torch

from . import oracle as O

AUX_RATIO_POWER_LAW = -0.7864636765648174
BIG_PRIME = 10007


def _log_prob(x, loc, scale):
    # TFP 0.9 Normal._log_prob
    return -0.5 * torch.square(x / scale - loc / scale) - (0.5 * math.log(2. * math.pi) + torch.log(scale))


def _simple_hash(matrix):
    w = torch.arange(69, 69 + matrix.shape[1], dtype=torch.int32)
    return torch.remainder(torch.sum(matrix * w, dim=1), BIG_PRIME - 1) + 1


def _pseudo_random_sample(scale, n_samples, index_matrix, seed):
    D = scale.shape[-1]
    r = torch.from_numpy(O.uniform_int(seed, n_samples * D)).reshape(n_samples, 1, D)
    hashes = _simple_hash(index_matrix)
    hashed = torch.remainder(r.unsqueeze(1) * hashes.reshape(-1, 1, 1), BIG_PRIME)
    u = hashed.to(torch.float32) / BIG_PRIME
    return torch.special.ndtri(u) * scale + 0.0          # dist.quantile


def encode_block(q_loc, q_scale, p_loc, p_scale, seed, omega, n_samples, n_beams):
    q_loc, q_scale, p_loc, p_scale = (torch.as_tensor(t, dtype=torch.float32).reshape(1, -1)
                                      for t in (q_loc, q_scale, p_loc, p_scale))
    dls = torch.log(q_scale) - torch.log(p_scale)
    total_kl = torch.sum(0.5 * torch.square(q_loc / p_scale - p_loc / p_scale) + 0.5 * torch.expm1(2. * dls) - dls)
    K = int(torch.ceil(total_kl / np.float32(omega)))
    cum = torch.zeros_like(p_scale)
    beams = beam_indices = None
    for iteration, i in enumerate(range(K - 1, -1, -1)):
        ratio = np.float32(np.power(i + 1., AUX_RATIO_POWER_LAW))
        coder_var = torch.pow(p_scale, 2)
        aux_var = ratio * (coder_var - cum)
        v = aux_var + cum
        aux_scale = torch.sqrt(aux_var)
        cum_scale = torch.sqrt(v)
        t_mean = (q_loc - p_loc) * v / coder_var
        t_var = torch.pow(q_scale, 2) * torch.pow(v, 2) / torch.pow(coder_var, 2) + v * (coder_var - v) / coder_var
        t_scale = torch.sqrt(t_var)
        if iteration > 0:
            samples = _pseudo_random_sample(aux_scale, n_samples, beam_indices, seed + iteration)
            combined = beams + samples                                             # [S, B, 1, D]
            log_probs = torch.sum(_log_prob(combined, t_mean, t_scale) - _log_prob(combined, 0. * t_mean, cum_scale),
                                  dim=(2, 3))
            flat = log_probs.reshape(-1)
            order = torch.argsort(flat, descending=True, stable=True)
            n_cur = beams.shape[0]
            best_beam = order[:n_beams] % n_cur
            best_aux = order[:n_beams] // n_cur
            beams = combined[best_aux, best_beam]
            beam_indices = torch.cat((beam_indices[best_beam, :iteration], best_aux[:, None].to(torch.int32)), dim=1)
        else:
            samples = _pseudo_random_sample(aux_scale, n_samples, torch.zeros((1, 0), dtype=torch.int32),
                                            seed + iteration)[:, 0]
            log_probs = torch.sum(_log_prob(samples, t_mean, t_scale) - _log_prob(samples, 0. * t_mean, cum_scale),
                                  dim=(1, 2))
            order = torch.argsort(log_probs, descending=True, stable=True)
            beams = samples[order[:n_beams]]
            beam_indices = order[:n_beams, None].to(torch.int32)
        cum = cum + aux_var
    if K == 0:
        return [], p_loc[0].numpy().copy()
    return [int(v) for v in beam_indices[0]], (beams[0] + p_loc)[0].numpy()


def encode_tensor(q_loc, q_scale, p_loc, p_scale, seed, omega, n_samples, n_beams, block_size):
    """GaussianCoder.encode loop (coder.py:412-457) over the reference-shaped block encoder."""
    mq, sq, mp, sp = (np.asarray(a, np.float32).reshape(-1) for a in (q_loc, q_scale, p_loc, p_scale))
    n = mq.size
    perm = O.tf_shuffle_perm(seed, n)
    out = np.zeros(n, np.float32)
    indices = []
    for lo, hi in O.split_blocks(n, block_size):
        g = perm[lo:hi]
        idx, samp = encode_block(mq[g], sq[g], mp[g], sp[g], seed, omega, n_samples, n_beams)
        indices.append(idx)
        out[g] = samp
    return indices, out
