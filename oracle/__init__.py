"""CPU oracle (TEST INFRASTRUCTURE ONLY).

A plain-PyTorch fp32 CPU restatement of the reference algorithm for the Graph-WaveNet + UNet
training hot path.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package -- as the checker, never as the thing measured or
shipped.  The product (``multimodal_outage_amd``) never imports it and has no CPU fallback.

Pinning: the reference ships no tests/golden vectors, so the restatement is
pinned against outputs of the reference's own class bodies executed in the build container
(``tools/make_goldens.py`` -> ``tests/golden/*.npz``; checked by ``tests/test_oracle_golden.py``).
torchmetrics (MAE/MAPE/RMSE, lit.py:25-27,36-38) is a third-party dependency absent from the
image: its restatement in ``metrics_ref.py`` is "parity unpinned".
"""
