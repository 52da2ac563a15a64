"""PixelCNN priors over the discrete VQ code grids on the HIP kernels -- drop-in for the reference's
models/pixelcnn_prior.py:25-54 (MaskedConv2d), :57-92 (GatedResBlock), :262-349 (PixelCNN) and :352-421 (HierarchicalPixelCNN):
same constructor signatures, state_dict keys (incl. the `mask` buffers) and init sequence, forward(x[B,H,W] int64) -> logits
[B, K, H, W] (a zero-copy NCHW view of the NHWC logits the kernels write).  PixelSNAIL (:95-259, :424+) adds causal
self-attention and is outside SURVEY 8f.4's row: the training stage refuses `--prior_type pixelsnail` by name.

Every convolution runs on the implicit-GEMM kernels; the embedding gather, the gated combine, the weight mask and the
cross-entropy are csrc/prior.hip.  Activations stay NHWC: nn.Embedding of a [B,H,W] grid IS the NHWC activation, and the
NHWC logits viewed as [B*H*W, K] are exactly the matrix F.cross_entropy is handed after the reference's permute + reshape."""
import torch
import torch.nn as tnn

from .. import nn as mnn
from .. import ops
from ._base import nchw_view


class MaskedConv2d(mnn.Conv2d):
    """pixelcnn_prior.py:25-54.  The mask multiplies the weight IN PLACE before each forward (so masked entries of the
    parameter are zero whenever it is used, whatever the optimizer wrote there in between)."""

    def __init__(self, mask_type, cin, cout, k, stride=1, padding=0, bias=True):
        super().__init__(cin, cout, k, stride=stride, padding=padding, bias=bias)
        assert mask_type in ("A", "B"), "mask_type must be 'A' or 'B'"
        mask = torch.zeros(cout, cin, k, k)
        mask[:, :, : k // 2, :] = 1.0
        mask[:, :, k // 2, : k // 2] = 1.0
        if mask_type == "B":
            mask[:, :, k // 2, k // 2] = 1.0
        self.register_buffer("mask", mask.contiguous(memory_format=torch.channels_last))  # the weight's memory layout
        self.mask_type = mask_type

    def forward(self, x, act=None, feeds_batchnorm=False):
        ops.mask_weight_(self.weight, self.mask)
        return super().forward(x, act, feeds_batchnorm)


class GatedResBlock(tnn.Module):
    """pixelcnn_prior.py:57-92: 1x1 -> ReLU -> masked-B 3x3 -> ReLU -> sigmoid(1x1) * tanh(1x1) + residual."""

    def __init__(self, channels, kernel_size=3):
        super().__init__()
        self.conv1 = mnn.Conv2d(channels, channels // 2, 1)
        self.conv2 = MaskedConv2d("B", channels // 2, channels // 2, kernel_size, padding=kernel_size // 2)
        self.conv_gate = mnn.Conv2d(channels // 2, channels, 1)
        self.conv_feature = mnn.Conv2d(channels // 2, channels, 1)

    def forward(self, x):
        out = self.conv2(self.conv1(x, "relu"), "relu")
        return ops.gated_residual(x, self.conv_gate(out, "sigmoid"), self.conv_feature(out, "tanh"))


class _OutHead(mnn.Stack):
    """nn.Sequential(ReLU, Conv1x1, ReLU, Conv1x1) with the reference's child indices 0..3 (pixelcnn_prior.py:300-305)."""

    def __init__(self, hidden, out):
        super().__init__(mnn.ReLU(), mnn.Conv2d(hidden, hidden, 1), mnn.ReLU(), mnn.Conv2d(hidden, out, 1))


class Embedding(tnn.Module):
    """nn.Embedding(K, D) with its default N(0, 1) init."""

    def __init__(self, K, D):
        super().__init__()
        ref = tnn.Embedding(K, D)
        self.weight = tnn.Parameter(ref.weight.detach().clone())
        self.num_embeddings, self.embedding_dim = K, D

    def forward(self, idx):
        return ops.embedding(idx, self.weight)


class PixelCNN(tnn.Module):
    """pixelcnn_prior.py:262-349."""

    def __init__(self, num_embeddings, embedding_dim=64, hidden_channels=128, num_layers=15, kernel_size=7, conditional_channels=0):
        super().__init__()
        self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
        self.embedding = Embedding(num_embeddings, embedding_dim)
        self.conv_in = MaskedConv2d("A", embedding_dim + conditional_channels, hidden_channels, kernel_size, padding=kernel_size // 2)
        self.res_blocks = tnn.ModuleList([GatedResBlock(hidden_channels) for _ in range(num_layers)])
        self.conv_out = _OutHead(hidden_channels, num_embeddings)

    def forward_nhwc(self, x, condition=None):
        """x [B, H, W] int64, condition NHWC [B, H, W, C] or None -> NHWC logits [B, H, W, K]."""
        h = self.embedding(x)
        if condition is not None:
            h = ops.concat_channels(h, condition)
        h = self.conv_in(h)
        for blk in self.res_blocks:
            h = blk(h)
        return self.conv_out(h)

    def forward(self, x, condition=None):
        if condition is not None:
            condition = ops.to_nhwc(condition)
        return nchw_view(self.forward_nhwc(x, condition))

    def loss(self, x, condition=None):
        """F.cross_entropy(logits.permute(0,2,3,1).reshape(-1, K), x.reshape(-1)) (main.py:1003-1006)."""
        logits = self.forward_nhwc(x, condition)
        return ops.cross_entropy(logits.reshape(-1, self.num_embeddings), x.reshape(-1))

    @torch.no_grad()
    def sample(self, batch_size, height, width, device, condition=None, temperature=1.0):
        """Raster-scan ancestral sampling (pixelcnn_prior.py:323-349); the categorical draw uses torch's device generator."""
        self.eval()
        samples = torch.zeros(batch_size, height, width, dtype=torch.long, device=device)
        cond = ops.to_nhwc(condition) if condition is not None else None
        for i in range(height):
            for j in range(width):
                logits = self.forward_nhwc(samples, cond)[:, i, j, :] / temperature
                probs = torch.softmax(logits, dim=1)
                samples[:, i, j] = torch.multinomial(probs, 1).squeeze(-1)
        return samples

    def total_trainable_params(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)


class HierarchicalPixelCNN(tnn.Module):
    """pixelcnn_prior.py:352-421: P(z_top) and P(z_bottom | upsampled embedding of z_top)."""

    def __init__(self, num_embeddings, embedding_dim=64, hidden_channels=128, num_layers=15):
        super().__init__()
        self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
        self.prior_top = PixelCNN(num_embeddings, embedding_dim, hidden_channels, num_layers)
        self.embedding_top = Embedding(num_embeddings, embedding_dim)
        self.upsample_top = mnn.ConvTranspose2d(embedding_dim, embedding_dim, 4, stride=2, padding=1)
        self.prior_bottom = PixelCNN(num_embeddings, embedding_dim, hidden_channels, num_layers, conditional_channels=embedding_dim)

    def _condition(self, z_top):
        return self.upsample_top(self.embedding_top(z_top))  # NHWC

    def forward_top(self, z_top):
        return self.prior_top(z_top)

    def forward_bottom(self, z_bottom, z_top):
        return nchw_view(self.prior_bottom.forward_nhwc(z_bottom, self._condition(z_top)))

    def forward(self, z_top, z_bottom):
        return {"logits_top": self.forward_top(z_top), "logits_bottom": self.forward_bottom(z_bottom, z_top)}

    def loss_function(self, z_top, z_bottom):
        loss_top = self.prior_top.loss(z_top)
        loss_bottom = self.prior_bottom.loss(z_bottom, self._condition(z_top))
        return {"loss_top": loss_top, "loss_bottom": loss_bottom, "total_loss": loss_top + loss_bottom}

    @torch.no_grad()
    def sample(self, batch_size, top_shape, bottom_shape, device, temperature=1.0):
        self.eval()
        z_top = self.prior_top.sample(batch_size, top_shape[0], top_shape[1], device, temperature=temperature)
        cond = nchw_view(self._condition(z_top))
        z_bottom = self.prior_bottom.sample(batch_size, bottom_shape[0], bottom_shape[1], device, condition=cond, temperature=temperature)
        return z_top, z_bottom

    @torch.no_grad()
    def sample_with_vqvae2(self, vqvae2_model, batch_size, device, temperature=1.0):
        t, b = vqvae2_model.latent_spatial_dim_top, vqvae2_model.latent_spatial_dim_bottom
        z_top, z_bottom = self.sample(batch_size, (t, t), (b, b), device, temperature)
        return vqvae2_model.decode_code(z_top, z_bottom)

    def total_trainable_params(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)
