"""Hierarchical VQ-VAE-2 on the HIP kernels -- drop-in for the reference's models/vq_vae2.py
(ResBlock :12-28, Encoder :31-58, Decoder :61-103, VQVAE2 :106-390)."""
import os

import torch

from .. import nn as mnn
from .. import objectives as O
from .. import ops
from ._base import HotPathModel, LazyDict, LazyScalar, nchw_view, resolve_lambda_weights
from .vq_vae import VectorQuantizer

N_RES_CHANNEL = 32  # hard-coded at models/vq_vae2.py:190-212


class ResBlock(torch.nn.Module):
    def __init__(self, in_channel, channel):
        super().__init__()
        self.conv = mnn.Stack(mnn.ReLU(), mnn.Conv2d(in_channel, channel, 3, padding=1), mnn.ReLU(),
                              mnn.Conv2d(channel, in_channel, 1))

    def forward(self, x):
        # the identity branch's cotangent is added by the input-gradient kernel of the branch's 3x3 conv (ops.ResCarrier)
        if not isinstance(x, torch.Tensor):
            x = ops.materialize(x)
        carrier = ops.ResCarrier()
        return self.conv(x, None, carrier, (x, carrier))  # the branch's last conv adds x in its epilogue


class Encoder(torch.nn.Module):
    def __init__(self, in_channel, channel, n_res_block, n_res_channel, stride):
        super().__init__()
        if stride == 4:
            blocks = [mnn.Conv2d(in_channel, channel // 2, 4, stride=2, padding=1), mnn.ReLU(),
                      mnn.Conv2d(channel // 2, channel, 4, stride=2, padding=1), mnn.ReLU(),
                      mnn.Conv2d(channel, channel, 3, padding=1)]
        elif stride == 2:
            blocks = [mnn.Conv2d(in_channel, channel // 2, 4, stride=2, padding=1), mnn.ReLU(),
                      mnn.Conv2d(channel // 2, channel, 3, padding=1)]
        else:
            raise ValueError(f"stride {stride} not supported")
        blocks += [ResBlock(channel, n_res_channel) for _ in range(n_res_block)]
        blocks.append(mnn.ReLU())
        self.blocks = mnn.Stack(*blocks)

    def forward(self, x):
        return self.blocks(x)


class Decoder(torch.nn.Module):
    def __init__(self, in_channel, out_channel, channel, n_res_block, n_res_channel, stride, output_activation="none"):
        super().__init__()
        blocks = [mnn.Conv2d(in_channel, channel, 3, padding=1)]
        blocks += [ResBlock(channel, n_res_channel) for _ in range(n_res_block)]
        blocks.append(mnn.ReLU())
        if stride == 4:
            blocks += [mnn.ConvTranspose2d(channel, channel // 2, 4, stride=2, padding=1), mnn.ReLU(),
                       mnn.ConvTranspose2d(channel // 2, out_channel, 4, stride=2, padding=1)]
        elif stride == 2:
            blocks.append(mnn.ConvTranspose2d(channel, out_channel, 4, stride=2, padding=1))
        if output_activation == "tanh":
            blocks.append(mnn.Tanh())
        elif output_activation == "sigmoid":
            blocks.append(mnn.Sigmoid())
        elif output_activation != "none":
            raise ValueError(f"Output activation {output_activation} not supported")
        self.blocks = mnn.Stack(*blocks)

    def forward(self, x):
        return self.blocks(x)


class VQVAE2(HotPathModel):
    graph_safe = True  # codebook usage stays on the device (LazyScalar)

    def __init__(self, in_channels, embedding_dim, num_embeddings, hidden_dims=(128, 256), num_residual_layers=2,
                 input_size=64, layer_norm="none", recons_activation="tanh", recons_objective="mse", lambda_weights=None,
                 device=None, **kwargs):
        super().__init__()
        hidden_dims = list(hidden_dims)
        self.device = device
        self.embedding_dim, self.num_embeddings = embedding_dim, num_embeddings
        self.num_residual_layers, self.input_size, self.in_channels = num_residual_layers, input_size, in_channels
        self._summary_mode = False
        recon_obj, recons_activation = O.get_recon_obj_and_activation(recons_objective, recons_activation=recons_activation, model=self)
        self.recon_obj = recon_obj
        self.objectives = {"reconstruction_loss": recon_obj, "commitment_loss": None, "embedding_loss": None}
        self.features = ["encoding_top", "encoding_bottom"]
        self.lambda_weights = resolve_lambda_weights(
            "VQVAE2", self.objectives, lambda_weights, {"reconstruction_loss": 1.0, "commitment_loss": 1.0, "embedding_loss": 1.0})
        if recons_activation not in mnn.ACTIVATIONS:
            raise ValueError(f"recons_activation {recons_activation} not supported")
        self.recons_activation = mnn.ACTIVATIONS[recons_activation]()
        ch, D, n = hidden_dims[0], embedding_dim, num_residual_layers
        self.enc_b = Encoder(in_channels, ch, n, N_RES_CHANNEL, stride=4)
        self.enc_t = Encoder(ch, ch, n, N_RES_CHANNEL, stride=2)
        self.quantize_conv_t = mnn.Conv2d(ch, D, 1)
        self.quantize_t = VectorQuantizer(num_embeddings, D)
        self.dec_t = Decoder(D, D, ch, n, N_RES_CHANNEL, stride=2)
        self.quantize_conv_b = mnn.Conv2d(D + ch, D, 1)
        self.quantize_b = VectorQuantizer(num_embeddings, D)
        self.vq_top, self.vq_bottom = self.quantize_t, self.quantize_b  # aliases (models/vq_vae2.py:199-200)
        self.upsample_t = mnn.ConvTranspose2d(D, D, 4, stride=2, padding=1)
        self.dec = Decoder(D + D, in_channels, ch, n, N_RES_CHANNEL, stride=4, output_activation=recons_activation)
        self.latent_spatial_dim_bottom = input_size // 4
        self.latent_spatial_dim_top = input_size // 8

    def encode(self, x):
        # the returned NCHW views ARE the graph nodes later stages consume (mtl_backward differentiates
        # the losses w.r.t. these feature tensors), so downstream ops re-enter through the views
        enc_b = nchw_view(self.enc_b(ops.to_nhwc(x)))
        enc_t = nchw_view(self.enc_t(ops.to_nhwc(enc_b)))
        quant_t, c_t, e_t, i_t = self.quantize_t(nchw_view(self.quantize_conv_t(ops.to_nhwc(enc_t))))
        used_t = self.quantize_t.last_used_count
        dec_t = self.dec_t(ops.to_nhwc(quant_t))
        quant_b, c_b, e_b, i_b = self.quantize_b(
            nchw_view(self.quantize_conv_b(ops.concat_channels(dec_t, ops.to_nhwc(enc_b)))))
        self._used = (used_t, self.quantize_b.last_used_count)
        return enc_b, enc_t, quant_t, quant_b, c_t, c_b, e_t, e_b, i_t, i_b

    def decode(self, quant_t, quant_b):
        up = self.upsample_t(ops.to_nhwc(quant_t))
        return nchw_view(self.dec(ops.concat_channels(up, ops.to_nhwc(quant_b))))

    def decode_code(self, code_t, code_b):
        q_t = self.quantize_t.embed_code(code_t).permute(0, 3, 1, 2)
        q_b = self.quantize_b.embed_code(code_b).permute(0, 3, 1, 2)
        return self.decode(q_t, q_b)

    def forward(self, x, **kwargs):
        enc_b, enc_t, quant_t, quant_b, c_t, c_b, e_t, e_b, i_t, i_b = self.encode(x)
        recons = self.decode(quant_t, quant_b)
        K = self.num_embeddings
        used_t, used_b = self._used
        usage = LazyScalar([used_t, used_b], 100.0 / K / 2.0)  # mean of the two codebooks' usage, read lazily
        # the two sums are formed only if somebody reads them: loss_function takes the four terms (one launch for all of it)
        out = LazyDict({"recons": recons, "encoding_top": enc_t, "encoding_bottom": enc_b, "quantized_top": quant_t,
                        "quantized_bottom": quant_b, "commitment_loss": lambda: c_t + c_b, "embedding_loss": lambda: e_t + e_b,
                        "codebook_usage_percentage": usage, "encoding_inds_top": i_t, "encoding_inds_bottom": i_b,
                        "_vq_terms": (c_t, c_b, e_t, e_b)})
        return out["recons"] if self._summary_mode else out

    def get_code_indices(self, x):
        self.eval()
        with torch.no_grad():
            r = self.encode(x)
        B = x.size(0)
        return {"indices_top": r[8].view(B, self.latent_spatial_dim_top, self.latent_spatial_dim_top),
                "indices_bottom": r[9].view(B, self.latent_spatial_dim_bottom, self.latent_spatial_dim_bottom)}

    def loss_function(self, inputs, args: dict) -> dict:
        lw = self.lambda_weights
        rec = self._recon(self.recon_obj, inputs, args["recons"], lw["reconstruction_loss"], VQVAE2)
        terms = args.get("_vq_terms") if os.environ.get("MOVAE_FUSE_LOSSES", "1") != "0" and rec.is_cuda else None
        if terms is not None:
            # top + bottom, the weights and the total in one launch (ops.CombineLosses); `terms` = (c_t, c_b, e_t, e_b)
            wc, we = lw["commitment_loss"], lw["embedding_loss"]
            coef = [[1.0, 0, 0, 0, 0], [0, wc, wc, 0, 0], [0, 0, 0, we, we]]
            rec, com, emb, total = ops.combine_losses([rec, *terms], coef)
            return {"reconstruction_loss": rec, "commitment_loss": com, "embedding_loss": emb, "total_loss": total}
        com = lw["commitment_loss"] * args["commitment_loss"]
        emb = lw["embedding_loss"] * args["embedding_loss"]
        return {"reconstruction_loss": rec, "commitment_loss": com, "embedding_loss": emb, "total_loss": rec + com + emb}

    def sample(self, num_samples=1, device=None):
        self.eval()
        with torch.no_grad():
            t, b, K = self.latent_spatial_dim_top, self.latent_spatial_dim_bottom, self.num_embeddings
            return self.decode_code(torch.randint(0, K, (num_samples, t, t), device=device),
                                    torch.randint(0, K, (num_samples, b, b), device=device))
