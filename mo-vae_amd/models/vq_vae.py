"""VQ-VAE on the HIP kernels -- drop-in for the reference's models/vq_vae.py (VectorQuantizer
:11-124, ResidualLayer :127-145, VQVAE :148-470)."""
import os

import torch

from .. import nn as mnn
from .. import objectives as O
from .. import ops
from ._base import HotPathModel, LazyScalar, activation_module, nchw_view, resolve_lambda_weights


class VectorQuantizer(torch.nn.Module):
    """Nearest-code lookup + straight-through estimator.  Takes / returns logical NCHW tensors like
    the reference (the permutes at models/vq_vae.py:28,57 are free: the buffers are NHWC already)."""

    def __init__(self, num_embeddings: int, embedding_dim: int):
        super().__init__()
        self.K, self.D = num_embeddings, embedding_dim
        self._summary_mode = False
        self.embedding = mnn.Codebook(num_embeddings, embedding_dim)
        self.last_used_count = None

    def forward(self, latents):
        x = ops.to_nhwc(latents)
        q, commitment, embedding, idx, used = ops.vector_quantize(x, self.embedding.weight)
        self.last_used_count = used
        q = nchw_view(q)
        if self._summary_mode:
            return q
        return q, commitment, embedding, idx

    def embed_code(self, code):
        return self.embedding(code)

    def get_codebook_usage_percentage_from_indices(self, encoding_inds) -> float:
        """models/vq_vae.py:110-124.  The distinct-code count was produced by the lookup kernel, so the
        host reads one int32 instead of running torch.unique."""
        if self.last_used_count is not None:
            return LazyScalar([self.last_used_count], 100.0 / self.K)  # read on the host only when somebody needs the number
        return float(torch.unique(encoding_inds).size(0) / self.K * 100.0)

    def get_used_embeddings(self, latents):
        with torch.no_grad():
            _, _, _, idx, _ = ops.vector_quantize(ops.to_nhwc(latents), self.embedding.weight)
        return torch.unique(idx)

    def get_codebook_usage_percentage(self, latents) -> float:
        return float(self.get_used_embeddings(latents).size(0) / self.K * 100.0)


class ResidualLayer(torch.nn.Module):
    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.resblock = mnn.Stack(mnn.Conv2d(in_channels, out_channels, 3, padding=1, bias=False), mnn.ReLU(),
                                  mnn.Conv2d(out_channels, out_channels, 1, bias=False))

    def forward(self, x):
        # the identity branch's cotangent is added by the input-gradient kernel of the branch's first conv (ops.ResCarrier)
        if not isinstance(x, torch.Tensor):
            x = ops.materialize(x)
        carrier = ops.ResCarrier()
        return self.resblock(x, None, carrier, (x, carrier))  # the branch's last conv adds x in its epilogue


def _conv_lrelu(cin, cout, k, stride, padding):
    return mnn.Stack(mnn.Conv2d(cin, cout, k, stride=stride, padding=padding), mnn.LeakyReLU())


class VQVAE(HotPathModel):
    graph_safe = True  # codebook usage stays on the device (LazyScalar)

    def __init__(self, in_channels, embedding_dim, num_embeddings, hidden_dims=(128, 256), num_residual_layers=6,
                 input_size=64, layer_norm="none", recons_activation="tanh", recons_objective="mse", lambda_weights=None,
                 device=None, **kwargs):
        super().__init__()
        hidden_dims = list(hidden_dims)
        self.device = device
        self.embedding_dim, self.num_embeddings = embedding_dim, num_embeddings
        self.num_residual_layers, self.input_size, self.in_channels = num_residual_layers, input_size, in_channels
        self._summary_mode = False
        self.latent_spatial_dim = input_size // (2 ** len(hidden_dims))
        recon_obj, recons_activation = O.get_recon_obj_and_activation(recons_objective, recons_activation=recons_activation, model=self)
        self.recon_obj = recon_obj
        self.objectives = {"reconstruction_loss": recon_obj, "embedding_loss": None, "commitment_loss": None}
        self.features = ["encoding"]
        if isinstance(lambda_weights, dict):  # the reference does not validate dict keys here (models/vq_vae.py:183-194)
            self.lambda_weights = lambda_weights
        else:
            self.lambda_weights = resolve_lambda_weights(
                "VQVAE", self.objectives, lambda_weights, {"reconstruction_loss": 1.0, "embedding_loss": 1.0, "commitment_loss": 0.25})
        if layer_norm not in ("batch", "layer", "none"):
            raise ValueError(f"Layer norm {layer_norm} not supported")
        self.recons_activation = activation_module(recons_activation)

        enc, cin = [], in_channels
        for h in hidden_dims:
            enc.append(_conv_lrelu(cin, h, 4, 2, 1))
            cin = h
        enc.append(_conv_lrelu(cin, cin, 3, 1, 1))
        enc += [ResidualLayer(cin, cin) for _ in range(num_residual_layers)]
        enc.append(mnn.LeakyReLU())
        enc.append(_conv_lrelu(cin, embedding_dim, 1, 1, 0))
        self.encoder = mnn.Stack(*enc)
        self.vq_layer = VectorQuantizer(num_embeddings, embedding_dim)
        dec = [_conv_lrelu(embedding_dim, hidden_dims[-1], 3, 1, 1)]
        dec += [ResidualLayer(hidden_dims[-1], hidden_dims[-1]) for _ in range(num_residual_layers)]
        dec.append(mnn.LeakyReLU())
        rev = hidden_dims[::-1]
        for i in range(len(rev) - 1):
            dec.append(mnn.Stack(mnn.ConvTranspose2d(rev[i], rev[i + 1], 4, stride=2, padding=1), mnn.LeakyReLU()))
        dec.append(mnn.Stack(mnn.ConvTranspose2d(rev[-1], self.in_channels, 4, stride=2, padding=1), self.recons_activation))
        self.decoder = mnn.Stack(*dec)

    def encode(self, x):
        return nchw_view(self.encoder(ops.to_nhwc(x)))

    def decode(self, z):
        return nchw_view(self.decoder(ops.to_nhwc(z)))

    def forward(self, x, **kwargs):
        encoding = self.encode(x)
        vq = self.vq_layer(encoding)
        if isinstance(vq, tuple):
            quantized, commitment, embedding, inds = vq
            usage = self.vq_layer.get_codebook_usage_percentage_from_indices(inds)
        else:
            quantized, commitment, embedding, inds, usage = vq, None, None, None, 0.0
        out = {"recons": self.decode(quantized), "quantized_inputs": quantized, "encoding": encoding,
               "commitment_loss": commitment, "embedding_loss": embedding, "codebook_usage_percentage": usage,
               "encoding_inds": inds}
        return out["recons"] if self._summary_mode else out

    def loss_function(self, inputs, args: dict) -> dict:
        lw, ld = self.lambda_weights, {}
        keys = list(self.objectives)
        if keys == ["reconstruction_loss", "embedding_loss", "commitment_loss"] and os.environ.get("MOVAE_FUSE_LOSSES", "1") != "0" \
                and args["recons"].is_cuda:
            # the two weights and the total in one launch (ops.CombineLosses) instead of four
            rec = self._recon(self.objectives["reconstruction_loss"], inputs, args["recons"], lw["reconstruction_loss"], VQVAE)
            coef = [[1.0, 0, 0], [0, lw["embedding_loss"], 0], [0, 0, lw["commitment_loss"]]]
            rec, emb, com, total = ops.combine_losses([rec, args["embedding_loss"], args["commitment_loss"]], coef)
            return {"reconstruction_loss": rec, "embedding_loss": emb, "commitment_loss": com, "total_loss": total}
        for key, fn in self.objectives.items():  # order: reconstruction, embedding, commitment (models/vq_vae.py:381-389)
            if key == "embedding_loss":
                ld[key] = lw[key] * args["embedding_loss"]
            elif key == "commitment_loss":
                ld[key] = lw[key] * args["commitment_loss"]
            else:
                ld[key] = fn(inputs, args["recons"], lw[key])
        ld["total_loss"] = sum(ld.values())
        return ld

    def get_code_indices(self, x):
        self.eval()
        with torch.no_grad():
            _, _, _, idx = self.vq_layer(self.encode(x))
        return idx.view(x.size(0), self.latent_spatial_dim, self.latent_spatial_dim)

    def sample(self, num_samples=1, device=None):
        """Uniform code sampling (models/vq_vae.py:425-470): meaningful only with a trained prior."""
        self.eval()
        with torch.no_grad():
            s = self.latent_spatial_dim
            codes = torch.randint(0, self.num_embeddings, (num_samples, s, s), device=device)
            q = self.vq_layer.embed_code(codes).permute(0, 3, 1, 2)
            return self.decode(q)
