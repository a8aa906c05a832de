"""Date2Vec time embedding -- mirror of reference date2vec.py (encode path on the HIP C-ABI).

``Date2Vec.encode`` (date2vec.py:49-53) = cat[fc1(x), sin(fc2(x))]: the 64-d date feature concatenated
to every county's feature vector (utils.py:103,124-129, unet.py:224).  The pretrained weights
(d2v_98291_17.169918439404636.pth) are not in the reference tree, so weights are whatever the caller
loads; the frozen, no-grad semantics of Date2VecConvert (date2vec.py:8-10) are kept.
"""
import torch
from torch import nn

from . import _lib as L


class Date2Vec(nn.Module):
    def __init__(self, k=32, act="sin"):
        super(Date2Vec, self).__init__()
        k1 = k // 2
        k2 = k // 2 if k % 2 == 0 else k // 2 + 1
        self.fc1 = nn.Linear(6, k1)
        self.fc2 = nn.Linear(6, k2)
        self.d2 = nn.Dropout(0.3)
        if act != 'sin':
            raise NotImplementedError("HIP encode implements act='sin' (the reference default)")
        self.fc3 = nn.Linear(k, k // 2)
        self.d3 = nn.Dropout(0.3)
        self.fc4 = nn.Linear(k // 2, 6)
        self.fc5 = torch.nn.Linear(6, 6)

    @torch.no_grad()
    def encode(self, x):
        """date2vec.py:49-53.  x: (n, 6) = [0,0,0,year,month,day] (utils.py:126)."""
        if not x.is_cuda:
            raise RuntimeError('Date2Vec.encode runs on the MI355X HIP path only')
        x = x.contiguous().float()
        n = x.shape[0]
        k1, k2 = self.fc1.out_features, self.fc2.out_features
        out = torch.empty((n, k1 + k2), device=x.device, dtype=torch.float32)
        L.call('mo_date2vec_encode', L.ptr(x), n, L.ptr(self.fc1.weight), L.ptr(self.fc1.bias), k1,
               L.ptr(self.fc2.weight), L.ptr(self.fc2.bias), k2, L.ptr(out), L.stream())
        return out


class Date2VecConvert:
    """date2vec.py:4-10: frozen model, returns a CPU tensor.  model: a Date2Vec on the GPU."""

    def __init__(self, model):
        self.model = model.eval()

    def __call__(self, x):
        dev = next(self.model.parameters()).device
        return self.model.encode(torch.as_tensor(x, dtype=torch.float32, device=dev).unsqueeze(0)).squeeze(0).cpu()


def time_embeddings(model, dates, n_counties=67):
    """utils.py:101-103: one 64-d embedding per date of the window, repeated for every county:
    dates: list of (year, month, day) of length H -> (n_counties, H, k)."""
    dev = next(model.parameters()).device
    x = torch.tensor([[0, 0, 0, y, m, d] for (y, m, d) in dates], dtype=torch.float32, device=dev)
    e = model.encode(x)                                   # (H, k)
    return e.view(1, len(dates), -1).repeat(n_counties, 1, 1)
