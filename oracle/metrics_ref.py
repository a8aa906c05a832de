"""Oracle: loss + metrics of lit.py:29-43.

nn.MSELoss (lit.py:23,33) is pinned by torch itself.  MAE / MAPE / RMSE come from torchmetrics
(lit.py:11,25-27), a third-party package that is NOT in this image and is not pinned by
requirements.txt: restated from its published definitions -- "parity unpinned".
  MAE  = mean |yhat - y|
  MAPE = mean |yhat - y| / clamp(|y|, min=1.17e-06)
  RMSE = sqrt(mean (yhat - y)^2)                       (lit.py:38: torch.sqrt of the MSE metric)
"""
import torch

MAPE_EPS = 1.17e-06


def mse(yhat, y):
    return torch.mean((yhat - y) ** 2)


def metrics(yhat, y):
    d = yhat - y
    mae = d.abs().mean()
    mape = (d.abs() / torch.clamp(y.abs(), min=MAPE_EPS)).mean()
    rmse = torch.sqrt((d * d).mean())
    return mae, mape, rmse
