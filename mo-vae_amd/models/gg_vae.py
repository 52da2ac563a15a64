"""Gradient-guided VAE on the HIP kernels -- drop-in for the reference's models/gg_vae.py:12-251: the VAE
plus an edge-weighted pixel loss and a Sobel edge-matching loss, K = 4 component losses.  Edge matching versions 1 (arch
`gg_vae`), 2 (max-normalised), 3 (angle) and 5 (cosine) (`gg_vae_v2 / _v3 / _v5`) run on csrc/edge.hip; version 6 cannot
run in the reference either (gg_vae.py:219 calls a function torch.nn.functional does not have) and is refused."""
import torch

from .. import objectives as O
from ._base import resolve_lambda_weights
from .vae import VAE


class GGVAE(VAE):
    def __init__(self, latent_dim=2, input_size=32, in_channels=3, hidden_dims=None, layer_norm="batch", recons_activation="tanh",
                 recons_objective="mse", lambda_weights=None, device=None, edge_matching_version=1, **kwargs):
        super().__init__(latent_dim=latent_dim, input_size=input_size, in_channels=in_channels, hidden_dims=hidden_dims,
                         layer_norm=layer_norm, recons_activation=recons_activation, recons_objective=recons_objective,
                         lambda_weights=None, device=device, **kwargs)
        if edge_matching_version == 6:
            raise NotImplementedError("edge_matching_version=6: the reference's edge_matching_loss_v6 raises AttributeError on its "
                                      "first call (gg_vae.py:219) and its thresholded edge maps carry no gradient")
        # gg_vae.py:57-63: an unknown version falls back to version 1
        mode = {1: "mag", 2: "maxnorm", 3: "angle", 5: "cosine"}.get(edge_matching_version, "mag")
        self.edge_matching_version = edge_matching_version
        # the Sobel taps are compiled into csrc/edge.hip; the buffers exist for state_dict parity (gg_vae.py:44-53)
        sx = torch.tensor([[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]])
        sy = torch.tensor([[-1.0, -2.0, -1.0], [0.0, 0.0, 0.0], [1.0, 2.0, 1.0]])
        self.register_buffer("sobel_x", sx.expand(3, 1, 3, 3).clone())
        self.register_buffer("sobel_y", sy.expand(3, 1, 3, 3).clone())
        self.objectives = {"reconstruction_loss": self.recon_obj, "kld_loss": self.kld_obj,
                           "gradient_guided_loss": O.edge_weighted_pixel_loss, "edge_matching_loss": O.make_edge_matching(mode)}
        self.lambda_weights = resolve_lambda_weights(
            "GGVAE", self.objectives, lambda_weights,
            {"reconstruction_loss": 1.0, "kld_loss": 0.00025, "gradient_guided_loss": 1.0, "edge_matching_loss": 1.0})

    def loss_function(self, inputs, args: dict) -> dict:
        lw = self.lambda_weights
        recons = args["recons"]
        rec = self.objectives["reconstruction_loss"](inputs, recons, lw["reconstruction_loss"])
        gg = self.objectives["gradient_guided_loss"](inputs, recons, lw["gradient_guided_loss"])
        em = self.objectives["edge_matching_loss"](inputs, recons, lw["edge_matching_loss"])
        kld = self.objectives["kld_loss"](args["mu"], args["log_var"], lw["kld_loss"])
        return {"reconstruction_loss": rec, "gradient_guided_loss": gg, "edge_matching_loss": em, "kld_loss": kld,
                "total_loss": rec + gg + em + kld}
