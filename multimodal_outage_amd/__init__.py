"""MI355X-native Graph-WaveNet + UNet training hot path (drop-in for aaparcedo/multimodal_outage's
models.graph_wavenet.gwnet / models.unet.Modified_UNET / lit.LitModified_UNET.training_step).

Compute runs in hand-written HIP kernels (csrc/) behind the C-ABI of include/mo_hip.h; PyTorch-ROCm
only owns device memory, streams, autograd wiring and torch.distributed.  There is no CPU fallback.
"""
__version__ = '0.1.0'
