"""Model factory -- drop-in for the reference's models/__init__.py:18-211 for the four hot-path
architectures (vae, vq_vae, vq_vae2, betatc_vae / btc_vae) and the SURVEY 8f.3 widening gg_vae[_v2|_v3|_v5] / gg_vq_vae[_v1.._v7] / gg_vq_vae2."""
from .betatc_vae import BetaTCVAE
from .gg_vae import GGVAE
from .gg_vq_vae import GGVQVAE
from .gg_vq_vae2 import GGVQVAE2
from .vae import VAE
from .vq_vae import VQVAE, VectorQuantizer
from .vq_vae2 import VQVAE2

OUT_OF_SCOPE_ARCHS = {
    "recursive_kl_vae", "cycle_vae", "recursive_cyclic_vae", "rc_vae", "sphere_encoder", "sphere_encoder_vit",
}


def _recons_objective(args):
    obj = getattr(args, "recons_objective", None) or getattr(args, "recons_obj", None)
    if obj is not None:
        return obj.lower()
    # backward compatibility: recons_dist -> objective (models/__init__.py:26-37)
    return {"bernoulli": "bce", "gaussian": "mse", "laplacian": "l1"}.get(getattr(args, "recons_dist", "gaussian"), "mse")


def get_network(input_size, num_channels=3, args=None, device=None):
    arch = getattr(args, "arch", "vae").lower()
    latent_dim = getattr(args, "latent_dim", 128)
    embedding_dim = getattr(args, "embedding_dim", 64)
    num_embeddings = getattr(args, "num_embeddings", 512)
    hidden_dims = getattr(args, "hidden_dims", [32, 64, 128, 256, 512])
    num_residual_layers = getattr(args, "num_residual_layers", 2)
    recons_objective = _recons_objective(args)
    recons_activation = getattr(args, "recons_activation", None)
    lambda_weights = getattr(args, "loss_weights", None) or getattr(args, "lambda_weights", None)
    common = dict(input_size=input_size, in_channels=num_channels, recons_objective=recons_objective,
                  recons_activation=recons_activation, device=device)

    if arch == "vae":
        ratio = args.batch_size / args.dataset_size  # kld weight is forced (models/__init__.py:49-55)
        if lambda_weights is None:
            lambda_weights = {"reconstruction_loss": 1.0, "kld_loss": ratio}
        elif isinstance(lambda_weights, dict):
            lambda_weights = dict(lambda_weights, kld_loss=ratio)
        else:
            lambda_weights = [lambda_weights[0], ratio]
        return VAE(latent_dim=latent_dim, hidden_dims=hidden_dims, lambda_weights=lambda_weights, **common)
    if arch in ("gg_vae", "gg_vae_v2", "gg_vae_v3", "gg_vae_v5", "gg_vae_v6"):  # models/__init__.py:147-163
        version = 1 if arch == "gg_vae" else int(arch.split("_")[-1].replace("v", ""))
        ratio = args.batch_size / args.dataset_size
        if lambda_weights is None:
            lambda_weights = {"reconstruction_loss": 1.0, "kld_loss": ratio, "gradient_guided_loss": 1.0, "edge_matching_loss": 1.0}
        elif isinstance(lambda_weights, dict):
            lambda_weights = dict(lambda_weights, kld_loss=ratio)
        return GGVAE(latent_dim=latent_dim, hidden_dims=hidden_dims, lambda_weights=lambda_weights, edge_matching_version=version,
                     **common)
    if arch == "vq_vae":
        if lambda_weights is None:
            lambda_weights = {"reconstruction_loss": 1.0, "embedding_loss": 1.0, "commitment_loss": 0.25}
        return VQVAE(embedding_dim=embedding_dim, num_embeddings=num_embeddings, hidden_dims=hidden_dims,
                     num_residual_layers=num_residual_layers, lambda_weights=lambda_weights, **common)
    if arch in ("gg_vq_vae", "gg_vq_vae_v1"):  # models/__init__.py:169-173
        if lambda_weights is None:
            lambda_weights = {"reconstruction_loss": 1.0, "gradient_guided_loss": 1.0, "embedding_loss": 1.0, "commitment_loss": 0.25}
        return GGVQVAE(embedding_dim=embedding_dim, num_embeddings=num_embeddings, hidden_dims=hidden_dims,
                       num_residual_layers=num_residual_layers, lambda_weights=lambda_weights, version="v1", **common)
    if arch in tuple(f"gg_vq_vae_v{i}" for i in range(2, 9)):  # models/__init__.py:174-178
        if lambda_weights is None:
            lambda_weights = {"reconstruction_loss": 1.0, "gradient_guided_loss": 1.0, "embedding_loss": 1.0, "commitment_loss": 0.25,
                              "edge_matching_loss": 1.0}
        return GGVQVAE(embedding_dim=embedding_dim, num_embeddings=num_embeddings, hidden_dims=hidden_dims,
                       num_residual_layers=num_residual_layers, lambda_weights=lambda_weights, version=arch.replace("gg_vq_vae_", ""),
                       **common)
    if arch == "vq_vae2":
        if lambda_weights is None:
            lambda_weights = {"reconstruction_loss": 1.0, "commitment_loss": 1.0, "embedding_loss": 0.25}
        return VQVAE2(embedding_dim=embedding_dim, num_embeddings=num_embeddings, hidden_dims=hidden_dims,
                      num_residual_layers=num_residual_layers, lambda_weights=lambda_weights, **common)
    if arch == "gg_vq_vae2":  # models/__init__.py:184-188
        if lambda_weights is None:
            lambda_weights = {"reconstruction_loss": 1.0, "commitment_loss": 1.0, "embedding_loss": 0.25,
                              "gradient_guided_loss": 1.0, "edge_matching_loss": 1.0}
        return GGVQVAE2(embedding_dim=embedding_dim, num_embeddings=num_embeddings, hidden_dims=hidden_dims,
                        num_residual_layers=num_residual_layers, lambda_weights=lambda_weights, version="v3", **common)
    if arch in ("betatc_vae", "btc_vae"):
        ratio = args.batch_size / args.dataset_size
        if lambda_weights is None:
            lambda_weights = {"reconstruction_loss": 1.0, "mi_loss": 1.0, "tc_loss": 1.0, "kld": ratio}
        elif isinstance(lambda_weights, dict):
            lambda_weights = dict(lambda_weights, kld=ratio)
        else:
            lambda_weights = [lambda_weights[0], lambda_weights[1], lambda_weights[2], ratio]
        return BetaTCVAE(latent_dim=latent_dim, hidden_dims=hidden_dims, anneal_steps=getattr(args, "anneal_steps", 200),
                         dataset_size=getattr(args, "dataset_size", 50000), lambda_weights=lambda_weights, **common)
    if arch in OUT_OF_SCOPE_ARCHS:
        raise NotImplementedError(
            f"Network architecture {arch} exists in the reference but is outside this build's hot-path scope "
            "(vae, vq_vae, vq_vae2, betatc_vae); see DESIGN.md")
    raise ValueError(f"Network architecture {arch} not supported")


__all__ = ["VAE", "VQVAE", "VQVAE2", "BetaTCVAE", "GGVAE", "GGVQVAE", "GGVQVAE2", "VectorQuantizer", "get_network"]
