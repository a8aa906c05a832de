"""LitModified_UNET -- mirror of the training surface of reference lit.py:18-72.

Same constructor ``(st_gnn, horizon, device)``, ``training_step(batch)``, ``validation_step(batch,
batch_idx)`` and ``configure_optimizers()``; ``lightning`` and ``torchmetrics`` are optional (absent in
this image): without Lightning the class derives from nn.Module and records ``self.log`` calls in
``self.logged``.  Loss and metrics (MSE, MAE, MAPE, RMSE; lit.py:33-38) come from one fused HIP
reduction (mo_mse_metrics) that also produces dL/dyhat.
"""
import torch
import torch.nn as nn
from torch.optim import lr_scheduler

from . import _lib as L
from .models.unet import Modified_UNET

try:                                     # pragma: no cover - not installed in the build image
    import lightning as _L
    _Base = _L.LightningModule
except Exception:                        # noqa: BLE001
    _Base = nn.Module


class _MseMetricsFn(torch.autograd.Function):
    """loss = mean((yhat-y)^2) with d loss/d yhat = 2 (yhat-y)/n; also MAE / MAPE / RMSE sums."""

    @staticmethod
    def forward(ctx, yhat, y):
        yhat = yhat.contiguous()
        y = y.contiguous().float()
        n = yhat.numel()
        dev = yhat.device
        sums = torch.empty(4, device=dev, dtype=torch.float32)
        grad = torch.empty_like(yhat)
        ws = torch.empty(L.load().mo_metrics_ws_floats(n), device=dev, dtype=torch.float32)
        L.call('mo_mse_metrics', L.ptr(yhat), L.ptr(y), n, L.ptr(sums), L.ptr(grad), L.ptr(ws), L.stream())
        ctx.save_for_backward(grad)
        out = sums[:3] / n
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, gl, gmae, gmape):
        (grad,) = ctx.saved_tensors
        return grad * gl, None


def mse_and_metrics(yhat, y):
    """Returns (loss, mae, mape, rmse) as 0-d tensors (lit.py:33-38; MAPE eps 1.17e-6 as torchmetrics)."""
    loss, mae, mape = _MseMetricsFn.apply(yhat, y)
    return loss, mae, mape, torch.sqrt(loss.detach())


def print_memory_usage():
    """utils.py:341-343 (works on ROCm through torch.cuda)."""
    print(f"Allocated: {torch.cuda.memory_allocated() / 1e9} GB")
    print(f"Cached: {torch.cuda.memory_reserved() / 1e9} GB")


class LitModified_UNET(_Base):
    def __init__(self, st_gnn, horizon, device, verbose=False, **model_kwargs):
        super().__init__()
        self.st_gnn = st_gnn
        self.horizon = horizon
        kw = dict(input_channels=1, output_channels=1)                  # lit.py:23; BASELINE config 3 passes 13 / 13 and
        kw.update(model_kwargs)                                         # image_dimension=256 through model_kwargs
        self.model = Modified_UNET(st_gnn=self.st_gnn, horizon=self.horizon, **kw).to(device=device)
        self._dev = torch.device(device)
        self.verbose = verbose
        self.logged = {}
        self.fused_loss = True      # False: model(x) -> yhat -> mo_mse_metrics, as two separate steps

    if _Base is nn.Module:
        @property
        def device(self):
            return self._dev

        def log(self, name, value, **kw):
            self.logged[name] = value

    def _step(self, batch, prefix):
        x, y, x_time = batch
        x, y = (tensor.to(self.device).permute(0, 2, 1, 3, 4, 5) for tensor in (x, y))     # lit.py:31
        if self.fused_loss:
            # forward + MSE / MAE / MAPE / RMSE (lit.py:32-38) with the last layer fused into the loss: yhat, which the
            # step never returns, is not materialised (Modified_UNET.forward_loss); y stays the permuted view
            return self.model.forward_loss(x, x_time.to(self.device), y)
        yhat = self.model(x, x_time.to(self.device))
        loss, mae, mape, rmse = mse_and_metrics(yhat, y)
        return loss, mae, mape, rmse

    def training_step(self, batch):
        loss, mae, mape, rmse = self._step(batch, 'train')
        if self.verbose:
            print_memory_usage()                                                            # lit.py:34
        self.log('train_loss', loss, prog_bar=True)
        self.log('train_mae', mae)
        self.log('train_mape', mape)
        self.log('train_rmse', rmse)
        return loss

    def validation_step(self, batch, batch_idx):
        with torch.no_grad():
            loss, mae, mape, rmse = self._step(batch, 'val')
        self.log('val_loss', loss, prog_bar=True)
        self.log('val_mae', mae)
        self.log('val_mape', mape)
        self.log('val_rmse', rmse)
        return loss

    def configure_optimizers(self):
        optimizer = torch.optim.Adam(self.parameters(), lr=1e-3)
        scheduler = lr_scheduler.CosineAnnealingLR(optimizer, T_max=10)
        return {'optimizer': optimizer,
                'lr_scheduler': {'scheduler': scheduler, 'monitor': 'val_loss', 'interval': 'epoch', 'frequency': 1}}
