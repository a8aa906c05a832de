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


class SyntheticBlackMarble(torch.utils.data.Dataset):
    def __init__(self, d2v_model, length=64, horizon=7, n_counties=67, size=128, start=datetime.date(2018, 9, 10),
                 device='cuda', seed=0):
        self.d2v, self.length, self.horizon, self.n_counties, self.size = d2v_model, length, horizon, n_counties, size
        self.start, self.device, self.seed = start, torch.device(device), seed

    def __len__(self):
        return self.length

    def _rasters(self, idx, offset):
        g = torch.Generator(device=self.device)
        g.manual_seed(self.seed * 1000003 + (idx + offset))
        # non-negative night-light radiance with a heavy tail, then the dataset normalisation
        r = torch.rand(self.horizon, self.n_counties, 1, self.size, self.size, device=self.device, generator=g)
        return normalize(-8.0 * torch.log1p(-r * 0.98))

    def __getitem__(self, idx):
        past = self._rasters(idx, 0)
        future = self._rasters(idx, self.horizon)
        days = [self.start + datetime.timedelta(days=idx + d) for d in range(self.horizon)]
        te = time_embeddings(self.d2v, [(d.year, d.month, d.day) for d in days], n_counties=self.n_counties)
        return past, future, te
