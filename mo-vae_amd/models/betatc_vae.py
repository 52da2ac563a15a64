"""Beta-TC-VAE on the HIP kernels -- drop-in for the reference's models/betatc_vae.py:12-350."""
import os

import torch

from .. import nn as mnn
from .. import objectives as O
from .. import ops
from ._base import HotPathModel, activation_module, nchw_view, resolve_lambda_weights


class BetaTCVAE(HotPathModel):
    num_iter = 0  # class-level iteration counter, like the reference (models/betatc_vae.py:13)

    def __init__(self, in_channels, latent_dim, hidden_dims=None, anneal_steps=200, input_size=32, dataset_size=None,
                 recons_objective="mse", recons_activation=None, lambda_weights=None, device=None, **kwargs):
        super().__init__()
        self.device = device
        recon_obj, recons_activation = O.get_recon_obj_and_activation(recons_objective, recons_activation=recons_activation, model=self)
        self.latent_dim, self.anneal_steps, self.input_size = latent_dim, anneal_steps, input_size
        self.in_channels, self.dataset_size = in_channels, dataset_size
        self.objectives = {"reconstruction_loss": recon_obj, "mi_loss": None, "tc_loss": None, "kld": None}
        self.lambda_weights = resolve_lambda_weights(
            "BetaTCVAE", self.objectives, lambda_weights, {"reconstruction_loss": 1.0, "mi_loss": 1.0, "tc_loss": 1.0, "kld": 1.0})
        self.features = ["mu", "log_var"]
        hidden_dims = [32, 32, 32, 32] if hidden_dims is None else list(hidden_dims)
        self.hidden_dims = hidden_dims
        final_act = activation_module(recons_activation)
        self.recons_activation = type(final_act)

        enc, cin = [], in_channels
        for h in hidden_dims:
            enc.append(mnn.Stack(mnn.Conv2d(cin, h, 4, stride=2, padding=1), mnn.LeakyReLU()))
            cin = h
        self.encoder = mnn.Stack(*enc)
        sp = input_size // (2 ** len(hidden_dims))
        self._sp = sp
        self.encoder_output_size = hidden_dims[-1] * sp * sp
        self.fc = mnn.Linear(self.encoder_output_size, 256)
        self.fc_mu = mnn.Linear(256, latent_dim)
        self.fc_var = mnn.Linear(256, latent_dim)
        self.decoder_input = mnn.Linear(latent_dim, self.encoder_output_size)
        rev = hidden_dims[::-1]
        self.decoder = mnn.Stack(*[mnn.Stack(mnn.ConvTranspose2d(rev[i], rev[i + 1], 3, stride=2, padding=1, output_padding=1),
                                             mnn.LeakyReLU()) for i in range(len(rev) - 1)])
        self.final_layer = mnn.Stack(mnn.ConvTranspose2d(rev[-1], rev[-1], 3, stride=2, padding=1, output_padding=1),
                                     mnn.LeakyReLU(), mnn.Conv2d(rev[-1], in_channels, 3, padding=1), final_act)
        self._log_iw = {}

    def encode(self, x):
        h = ops.flatten_nchw(self.encoder(ops.to_nhwc(x)))
        h = self.fc(h)  # no activation after fc (models/betatc_vae.py:180)
        return list(mnn.linear_pair(h, self.fc_mu, self.fc_var))

    def decode(self, z):
        h = ops.unflatten_nchw(self.decoder_input(z), self.hidden_dims[-1], self._sp, self._sp)
        return nchw_view(mnn.chain(h, self.decoder, self.final_layer))

    graph_safe = True
    _iter_dev = None

    def prepare_for_graph(self):
        """Moves the annealing counter (models/betatc_vae.py:13,301-305) to the device, starting from its current value."""
        super().prepare_for_graph()
        if self._iter_dev is None:
            self._iter_dev = torch.tensor(float(self.num_iter), dtype=torch.float32, device=next(self.parameters()).device)

    def reparameterize(self, mu, logvar):
        return self._reparameterize(mu, logvar)

    def forward(self, x, **kwargs):
        mu, log_var = self.encode(x)
        z = self.reparameterize(mu, log_var)
        return {"recons": self.decode(z), "input": x, "mu": mu, "log_var": log_var, "z": z}

    def _log_importance_weights(self, B, device):
        """models/betatc_vae.py:273-289; cached per batch size instead of being rebuilt on the host and
        copied to the device every step."""
        key = (B, str(device))
        if key not in self._log_iw:
            M_N = B / self.dataset_size if self.dataset_size is not None else B / 50000
            ds = (1 / M_N) * B
            strat = (ds - B + 1) / (ds * (B - 1))
            W = torch.Tensor(B, B).fill_(1 / (B - 1))
            W.view(-1)[::B] = 1 / ds
            W.view(-1)[1::B] = strat
            W[B - 2, 0] = strat
            self._log_iw[key] = W.log().to(device)
        return self._log_iw[key]

    def loss_function(self, inputs, args: dict) -> dict:
        lw = self.lambda_weights
        z = args["z"]
        rec = self._recon(self.objectives["reconstruction_loss"], inputs, args["recons"], lw["reconstruction_loss"], BetaTCVAE)
        terms = ops.tc_decomposition(z, args["mu"], args["log_var"], self._log_importance_weights(z.shape[0], z.device))
        if os.environ.get("MOVAE_FUSE_LOSSES", "1") != "0" and rec.is_cuda:
            # weights, annealing and the total in one launch (ops.CombineLosses) instead of eight; the annealing counter is
            # advanced inside the kernel in graph mode (prepare_for_graph), on the host otherwise
            anneal, host = None, 1.0
            if self.training and self._iter_dev is not None:
                anneal = (3, self._iter_dev, float(self.anneal_steps), True)
            elif self.training:
                self.num_iter += 1
                host = min(0 + 1 * self.num_iter / self.anneal_steps, 1)
            coef = [[1.0, 0, 0, 0], [0, lw["mi_loss"], 0, 0], [0, 0, lw["tc_loss"] * 1, 0], [0, 0, 0, lw["kld"] * 1 * host]]
            rec, mi, tc, kld, total = ops.combine_losses([rec, terms], coef, anneal)
            return {"reconstruction_loss": rec, "mi_loss": mi, "tc_loss": tc, "kld": kld, "total_loss": total}
        if self.training and self._iter_dev is not None:
            # graph mode (prepare_for_graph): the iteration counter lives on the device, so a replayed capture anneals
            self._iter_dev.add_(1.0)
            anneal = torch.clamp(self._iter_dev / float(self.anneal_steps), max=1.0)
        elif self.training:
            self.num_iter += 1
            anneal = min(0 + 1 * self.num_iter / self.anneal_steps, 1)
        else:
            anneal = 1.0
        mi = lw["mi_loss"] * terms[0]
        tc = lw["tc_loss"] * 1 * terms[1]
        kld = lw["kld"] * 1 * anneal * terms[2]
        return {"reconstruction_loss": rec, "mi_loss": mi, "tc_loss": tc, "kld": kld, "total_loss": rec + mi + tc + kld}

    def sample(self, num_samples, device=None, **kwargs):
        with torch.no_grad():
            return self.decode(torch.randn(num_samples, self.latent_dim).to(device))

    def generate(self, x, **kwargs):
        return self.forward(x)["recons"]
