"""Gradient-guided VQ-VAE-2 on the HIP kernels -- drop-in for the reference's models/gg_vq_vae2.py:14-161:
the hierarchical VQ-VAE-2 plus the edge-weighted pixel loss (:103-116) and the Sobel-magnitude smooth-L1 edge matching
loss (`edge_matching_loss_v2`, :118-129 -- the same arithmetic as GGVAE's version 1), K = 5 component losses."""
import torch

from .. import objectives as O
from .vq_vae2 import VQVAE2

_KEYS = ("reconstruction_loss", "commitment_loss", "embedding_loss", "gradient_guided_loss", "edge_matching_loss")


class GGVQVAE2(VQVAE2):
    def __init__(self, in_channels, embedding_dim, num_embeddings, hidden_dims=(128, 256), num_residual_layers=2, input_size=64,
                 layer_norm="none", recons_activation="tanh", recons_objective="mse", lambda_weights=None, version="v3",
                 device=None, **kwargs):
        super().__init__(in_channels=in_channels, embedding_dim=embedding_dim, num_embeddings=num_embeddings, hidden_dims=hidden_dims,
                         num_residual_layers=num_residual_layers, input_size=input_size, layer_norm=layer_norm,
                         recons_activation=recons_activation, recons_objective=recons_objective, lambda_weights=None, device=device,
                         **kwargs)
        if in_channels != 3:
            raise NotImplementedError("GGVQVAE2: the Sobel kernels of csrc/edge.hip cover the 3-channel inputs of the reference's data sets")
        sx = torch.tensor([[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]])
        sy = torch.tensor([[-1.0, -2.0, -1.0], [0.0, 0.0, 0.0], [1.0, 2.0, 1.0]])
        self.register_buffer("sobel_x", sx.expand(in_channels, 1, 3, 3).clone())  # state_dict parity (gg_vq_vae2.py:51-58)
        self.register_buffer("sobel_y", sy.expand(in_channels, 1, 3, 3).clone())
        self.version = version  # stored, never branched on (the reference's class has one code path)
        self.objectives["gradient_guided_loss"] = O.edge_weighted_pixel_loss
        self.objectives["edge_matching_loss"] = O.edge_matching_loss
        if lambda_weights is None:
            lambda_weights = {"reconstruction_loss": 1.0, "commitment_loss": 1.0, "embedding_loss": 0.25,
                              "gradient_guided_loss": 1.0, "edge_matching_loss": 1.0}
        elif isinstance(lambda_weights, list):
            if len(lambda_weights) != 5:
                raise ValueError("GGVQVAE2 requires 5 lambda_weights (reconstruction, commitment, embedding, gradient_guided, "
                                 f"edge_matching), got {len(lambda_weights)}")
            lambda_weights = dict(zip(_KEYS, lambda_weights))
        elif isinstance(lambda_weights, dict):
            expected, provided = set(self.objectives.keys()), set(lambda_weights.keys())
            if expected != provided:
                msg = "lambda_weights keys must match objectives keys. "
                if expected - provided:
                    msg += f"Missing: {expected - provided}. "
                if provided - expected:
                    msg += f"Extra: {provided - expected}."
                raise ValueError(msg)
        else:
            raise TypeError(f"lambda_weights must be dict or list, got {type(lambda_weights)}")
        self.lambda_weights = lambda_weights

    def loss_function(self, inputs, args: dict) -> dict:
        lw = self.lambda_weights
        recons = args["recons"]
        rec = self.recon_obj(inputs, recons, lw["reconstruction_loss"])
        gg = O.edge_weighted_pixel_loss(inputs, recons, lw["gradient_guided_loss"])
        em = O.edge_matching_loss(inputs, recons, lw["edge_matching_loss"])
        com = lw["commitment_loss"] * args["commitment_loss"]
        emb = lw["embedding_loss"] * args["embedding_loss"]
        return {"reconstruction_loss": rec, "commitment_loss": com, "embedding_loss": emb, "gradient_guided_loss": gg,
                "edge_matching_loss": em, "total_loss": rec + com + emb + gg + em}
