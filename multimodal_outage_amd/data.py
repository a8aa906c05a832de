"""Synthetic feeder with the tensor contract of reference BlackMarbleDataset.__getitem__ (utils.py:67-105).

The reference data layer is broken at HEAD and its data are absent, so training inputs are
synthetic; this class reproduces only the *contract* the training step consumes:
    past, future : (H, n_counties, 1, S, S) fp32, normalised with the dataset constants (utils.py:31-38)
    time_embeds  : (n_counties, H, 64) -- one Date2Vec embedding per day, repeated for every county (utils.py:103)
Rasters are generated on the GPU (no host->device copy of images) from a per-index seed.
"""
import datetime

import torch

from .date2vec import time_embeddings

MEAN = 3.201447427712248        # utils.py:31
STD = 10.389727592468262        # utils.py:32


def normalize(radiance):
    """transforms.Normalize(mean, std) of utils.py:35-38."""
    return (radiance - MEAN) / STD


def denormalize(tensor):
    """utils.py:40-44."""
    return tensor * STD + MEAN


FILL_VALUE = 6.5535e+03         # utils.py:62: the product's no-data value, set to 0 before the transform


def prepare_rasters(raw, size=128):
    """BlackMarbleDataset's per-image transform on the device (utils.py:35-38,59-64): fill value -> 0,
    transforms.Resize((size, size)) (bilinear, antialiased: torchvision 0.18 on a float tensor), Normalize(MEAN, STD).
    raw: (..., h, w) fp32 radiance on the GPU -> (..., size, size)."""
    from . import _lib as L
    if not raw.is_cuda:
        raise RuntimeError('prepare_rasters runs on the MI355X HIP path only (no CPU fallback)')
    raw = raw.contiguous().float()
    h, w = raw.shape[-2:]
    n = raw.numel() // (h * w)
    out = torch.empty(raw.shape[:-2] + (size, size), device=raw.device, dtype=torch.float32)
    L.call('mo_raster_prepare', L.ptr(raw), n, h, w, FILL_VALUE, MEAN, STD, L.ptr(out), size, size, L.stream())
    return out


class SyntheticBlackMarble(torch.utils.data.Dataset):
    def __init__(self, d2v_model, length=64, horizon=7, n_counties=67, size=128, start=datetime.date(2018, 9, 10),
                 device='cuda', seed=0, native_size=None):
        """native_size=(h, w): rasters are generated at that size (with no-data pixels) and go through the reference's
        per-image transform on the device (prepare_rasters); None: generated at `size`, already normalised."""
        self.d2v, self.length, self.horizon, self.n_counties, self.size = d2v_model, length, horizon, n_counties, size
        self.start, self.device, self.seed = start, torch.device(device), seed
        self.native_size = native_size

    def __len__(self):
        return self.length

    def _rasters(self, idx, offset):
        g = torch.Generator(device=self.device)
        g.manual_seed(self.seed * 1000003 + (idx + offset))
        # non-negative night-light radiance with a heavy tail, then the dataset normalisation
        if self.native_size is not None:
            h, w = self.native_size
            r = torch.rand(self.horizon, self.n_counties, 1, h, w, device=self.device, generator=g)
            raw = -8.0 * torch.log1p(-r * 0.98)
            raw = torch.where(r < 0.02, torch.full_like(raw, FILL_VALUE), raw)       # cloud / no-data pixels
            return prepare_rasters(raw, self.size)
        r = torch.rand(self.horizon, self.n_counties, 1, self.size, self.size, device=self.device, generator=g)
        return normalize(-8.0 * torch.log1p(-r * 0.98))

    def __getitem__(self, idx):
        past = self._rasters(idx, 0)
        future = self._rasters(idx, self.horizon)
        days = [self.start + datetime.timedelta(days=idx + d) for d in range(self.horizon)]
        te = time_embeddings(self.d2v, [(d.year, d.month, d.day) for d in days], n_counties=self.n_counties)
        return past, future, te
