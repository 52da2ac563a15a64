"""Convolutional VAE on the HIP kernels -- drop-in for the reference's models/vae.py:27-228
(same constructor signature, state_dict keys, forward / loss_function dictionaries)."""
import os

import torch

from .. import nn as mnn
from .. import objectives as O
from .. import ops
from ._base import HotPathModel, activation_module, nchw_view, resolve_lambda_weights


def _down(cin, cout):  # models/vae.py:119-126
    return mnn.Stack(mnn.Conv2d(cin, cout, 3, stride=2, padding=1), mnn.BatchNorm2d(cout), mnn.LeakyReLU())


def _up(cin, cout):  # models/vae.py:149-158
    return mnn.Stack(mnn.ConvTranspose2d(cin, cout, 3, stride=2, padding=1, output_padding=1), mnn.BatchNorm2d(cout),
                     mnn.LeakyReLU())


class _Flatten(torch.nn.Module):
    def forward(self, x):
        return ops.flatten_nchw(x)


class _Unflatten(torch.nn.Module):
    def __init__(self, c, h, w):
        super().__init__()
        self.chw = (c, h, w)

    def forward(self, x):
        return ops.unflatten_nchw(x, *self.chw)


class VAE(HotPathModel):
    graph_safe = True

    def __init__(self, latent_dim=2, input_size=32, in_channels=3, hidden_dims=None, layer_norm="batch",
                 recons_activation="tanh", recons_objective="mse", lambda_weights=None, device=None, **kwargs):
        super().__init__()
        self.device = device
        recon_obj, recons_activation = O.get_recon_obj_and_activation(recons_objective, recons_activation=recons_activation, model=self)
        self.recon_obj, self.kld_obj = recon_obj, O.kl_divergence
        self.objectives = {"reconstruction_loss": recon_obj, "kld_loss": O.kl_divergence}
        self.features = ["mu", "log_var"]
        self.lambda_weights = resolve_lambda_weights("VAE", self.objectives, lambda_weights,
                                                     {"reconstruction_loss": 1.0, "kld_loss": 0.00025})
        if layer_norm != "batch":
            # the factory never forwards --layer_norm (models/__init__.py:56), so "batch" is the only reachable value
            raise ValueError(f"Layer norm {layer_norm} not supported")
        hidden_dims = [32, 64, 128, 256, 512] if hidden_dims is None else list(hidden_dims)
        self.latent_dim, self.input_size, self.in_channels, self.hidden_dims = latent_dim, input_size, in_channels, hidden_dims
        sp = input_size // (2 ** len(hidden_dims))
        feat = hidden_dims[-1] * sp * sp
        final_act = activation_module(recons_activation)

        # creation order == the reference's (RNG parity): encoder, mu, log_var, decoder_input, decoder blocks, final layer
        enc, cin = [], in_channels
        for h in hidden_dims:
            enc.append(_down(cin, h))
            cin = h
        enc.append(_Flatten())
        self.encoder = mnn.Stack(*enc)
        self.mu = mnn.Linear(feat, latent_dim)
        self.log_var = mnn.Linear(feat, latent_dim)
        self.decoder_input = mnn.Linear(latent_dim, feat)
        rev = hidden_dims[::-1]
        dec = [_Unflatten(rev[0], sp, sp)] + [_up(rev[i], rev[i + 1]) for i in range(len(rev) - 1)]
        # registration order: final_layer before decoder (models/vae.py:161-175)
        self.final_layer = mnn.Stack(mnn.ConvTranspose2d(rev[-1], rev[-1], 3, stride=2, padding=1, output_padding=1),
                                     mnn.BatchNorm2d(rev[-1]), mnn.LeakyReLU(),
                                     mnn.Conv2d(rev[-1], in_channels, 3, padding=1), final_act)
        self.decoder = mnn.Stack(*dec)

    # -- reference API ------------------------------------------------------------------------
    def encode(self, x):
        h = self.encoder(ops.to_nhwc(x))
        return mnn.linear_pair(h, self.mu, self.log_var)  # (fc_mu || fc_var in one launch where the shapes allow)

    def reparameterize(self, mu, log_var):
        return self._reparameterize(mu, log_var)

    def decode(self, z):
        y = self.final_layer(self.decoder(self.decoder_input(z)))
        # the output activation's link (nn.Stack): the reconstruction loss is the one reader of `recons` in a training step
        self._recons_link, self.final_layer._out_link = self.final_layer._out_link, None
        return nchw_view(y)

    def forward(self, x):
        mu, log_var = self.encode(x)
        z = self.reparameterize(mu, log_var)
        return {"recons": self.decode(z), "mu": mu, "log_var": log_var, "z": z}

    def loss_function(self, inputs, args: dict) -> dict:
        lw = self.lambda_weights
        rec_fn, kld_fn = self.objectives["reconstruction_loss"], self.objectives["kld_loss"]
        if (type(self) is VAE and getattr(rec_fn, "kind", None) in ops.L.RECON and kld_fn is O.kl_divergence and inputs.dim() == 4
                and args["mu"].dim() == 2 and os.environ.get("MOVAE_FUSE_LOSSES", "1") != "0"):
            # both terms and their sum from one final kernel (ops.VAELosses): the same values, two launches fewer
            x, r = ops.to_nhwc(inputs), ops.to_nhwc(args["recons"])
            link = getattr(self, "_recons_link", None)
            rec, kld, total = ops.vae_losses(r, x, rec_fn.kind, lw["reconstruction_loss"], args["mu"], args["log_var"], lw["kld_loss"],
                                             act_link=link if (link is not None and link.y is not None and link.y.data_ptr() == r.data_ptr()) else None)
            return {"reconstruction_loss": rec, "kld_loss": kld, "total_loss": total}
        rec = self.objectives["reconstruction_loss"](inputs, args["recons"], lw["reconstruction_loss"])
        kld = self.objectives["kld_loss"](args["mu"], args["log_var"], lw["kld_loss"])
        return {"reconstruction_loss": rec, "kld_loss": kld, "total_loss": rec + kld}

    def sample(self, num_samples=1, device=None):
        self.eval()
        with torch.no_grad():
            z = torch.randn(num_samples, self.latent_dim).to(device)
            return self.decode(z)
