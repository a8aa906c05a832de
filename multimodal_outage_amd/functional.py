"""Stand-alone (unfused) HIP ops in the reference's (B,C,N,T) layout, for API parity with the
helper modules nconv / linear of graph_wavenet.py:60-74.  Forward only (inference helpers); the
training path is the fused engine."""
import torch

from . import _lib as L


def _to_nbtc(x):
    if not x.is_cuda:
        raise RuntimeError('nconv / linear / gcn run on the MI355X HIP path only (no CPU fallback); got a CPU tensor')
    B, C, N, T = x.shape
    x = x.contiguous().float()
    y = torch.empty((N * B * T, C), device=x.device, dtype=torch.float32)
    L.call('mo_nchw_to_nbtc', L.ptr(x), L.ptr(y), B, C, N, T, None, L.stream())
    return y


def _from_nbtc(y, B, C, N, T):
    x = torch.empty((B, C, N, T), device=y.device, dtype=torch.float32)
    L.call('mo_nbtc_to_nchw', L.ptr(y), L.ptr(x), B, C, N, T, None, L.stream())
    return x


@torch.no_grad()
def nconv(x, A):
    """einsum('ncvl,vw->ncwl') (graph_wavenet.py:65) as a dense node-axis product."""
    B, C, N, T = x.shape
    xi = _to_nbtc(x)
    yi = torch.empty_like(xi)
    A = A.to(x.device).contiguous().float()
    L.call('mo_adj_gemm', L.ptr(A), N, L.ptr(xi), L.ptr(yi), B * T * C, 0, L.stream())
    return _from_nbtc(yi, B, C, N, T)


@torch.no_grad()
def conv1x1(x, weight, bias=None):
    """nn.Conv2d(kernel_size=(1,1)) forward (graph_wavenet.py:71-74)."""
    B, C, N, T = x.shape
    Co = weight.shape[0]
    xi = _to_nbtc(x)
    yi = torch.empty((xi.shape[0], Co), device=x.device, dtype=torch.float32)
    w = weight.reshape(Co, C).contiguous()
    L.call('mo_conv1x1_fwd', L.ptr(xi), C, 0, 0, 0, 0, L.ptr(w), L.ptr(bias), Co, L.ptr(yi), xi.shape[0], 0, 0,
           L.stream())
    return _from_nbtc(yi, B, Co, N, T)
