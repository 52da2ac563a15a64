"""Gradient-guided VQ-VAE on the HIP kernels -- drop-in for the reference's models/gg_vq_vae.py:12-123,
the VQ-VAE plus the edge-weighted pixel loss (version v1, arch `gg_vq_vae` / `gg_vq_vae_v1`, K = 4) and, for v2..v7
(`gg_vq_vae_v2` .. `_v7`), one of the edge-matching variants of csrc/edge.hip (K = 5).  v8 compares thresholded edge maps
(gg_vq_vae.py:266-271): that loss carries no gradient, so no Jacobian row exists for it and the version is refused."""
import torch

from .. import objectives as O
from .vq_vae import VQVAE


# gg_vq_vae.py:65-88: version -> edge_matching_loss_vN -> the kernel's variant
EDGE_MODE = {"v1": None, "v2": "signed_mse", "v3": "mag", "v4": "maxnorm", "v5": "angle", "v6": "masked", "v7": "cosine"}


class GGVQVAE(VQVAE):
    def __init__(self, in_channels, embedding_dim, num_embeddings, hidden_dims=(128, 256), num_residual_layers=6, input_size=64,
                 layer_norm="none", recons_activation="tanh", recons_objective="mse", lambda_weights=None, version="v1",
                 device=None, **kwargs):
        super().__init__(in_channels=in_channels, embedding_dim=embedding_dim, num_embeddings=num_embeddings, hidden_dims=hidden_dims,
                         num_residual_layers=num_residual_layers, input_size=input_size, layer_norm="none",
                         recons_activation=recons_activation, recons_objective=recons_objective, lambda_weights=None, device=device,
                         **kwargs)
        if version == "v8":
            raise NotImplementedError("GGVQVAE version v8: edge_matching_loss_v7 (gg_vq_vae.py:266-271) is a step function of the "
                                      "reconstruction -- it has no gradient to aggregate")
        if version not in EDGE_MODE:
            raise ValueError(f"Version {version} not supported. Choose from: v1, v2, v3, v4, v5, v6, v7, v8")
        self.version = version
        sx = torch.tensor([[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]])
        sy = torch.tensor([[-1.0, -2.0, -1.0], [0.0, 0.0, 0.0], [1.0, 2.0, 1.0]])
        self.register_buffer("sobel_x", sx.expand(3, 1, 3, 3).clone())  # state_dict parity (gg_vq_vae.py:49-60)
        self.register_buffer("sobel_y", sy.expand(3, 1, 3, 3).clone())
        self.objectives = {"reconstruction_loss": self.recon_obj, "embedding_loss": None, "commitment_loss": None,
                           "gradient_guided_loss": O.edge_weighted_pixel_loss}
        if EDGE_MODE[version] is not None:
            self.objectives["edge_matching_loss"] = O.make_edge_matching(EDGE_MODE[version])
        if lambda_weights is None:
            lambda_weights = {"reconstruction_loss": 1.0, "embedding_loss": 1.0, "commitment_loss": 0.25, "gradient_guided_loss": 1.0}
            if version != "v1":
                lambda_weights["edge_matching_loss"] = 1.0
        elif isinstance(lambda_weights, list):
            if version == "v1" and len(lambda_weights) != 4:
                raise ValueError("GGVQVAE v1 requires 4 lambda_weights (reconstruction, embedding, commitment, gradient_guided), "
                                 f"got {len(lambda_weights)}")
            if version != "v1" and len(lambda_weights) != 5:
                raise ValueError("GGVQVAE v2 requires 5 lambda_weights (reconstruction, embedding, commitment, gradient_guided, "
                                 f"edge_matching), got {len(lambda_weights)}")
            lambda_weights = dict(zip(self.objectives.keys(), lambda_weights))
        self.lambda_weights = lambda_weights  # dicts are taken as given (gg_vq_vae.py:92-122 has no key validation)
