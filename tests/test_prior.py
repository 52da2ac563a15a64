"""PixelCNN prior: the oracle against the golden vectors produced by the reference's own
models/pixelcnn_prior.py (CPU), and the HIP models against the same vectors and against one step of the prior loop (GPU)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, meta_of

CFG_KEYS = ("num_embeddings", "embedding_dim", "hidden_channels", "num_layers")


def _cfg(fx, hier):
    m = meta_of(fx)
    c = {k: int(m[k]) for k in CFG_KEYS}
    c["hierarchical"] = hier
    return c, int(m["seed"]), float(m["lr"])


def _codes(fx, hier):
    if hier:
        return torch.from_numpy(fx["z_top"]), torch.from_numpy(fx["z_bottom"])
    return torch.from_numpy(fx["z"]), None


@pytest.mark.parametrize("tag", ["flat", "hier"])
def test_oracle_prior_matches_reference_vectors(tag):
    from oracle import prior as OP

    fx = load_golden("pixelcnn_tiny")
    hier = tag == "hier"
    cfg, seed, lr = _cfg(fx, hier)
    tr = OP.PriorTrainer(cfg, seed, lr=lr)
    keys = [k[len(tag) + 5:] for k in fx.files if k.startswith(f"{tag}.sd0.")]
    assert list(tr.sd.keys()) == keys  # state_dict keys and order incl. the mask buffers
    for k in keys:
        assert np.array_equal(tr.sd[k].detach().numpy(), fx[f"{tag}.sd0.{k}"]), f"init replay {k}"
    zt, zb = _codes(fx, hier)
    ld, out, g = tr.grads(zt, zb)
    for k, v in out.items():
        np.testing.assert_allclose(v.detach().numpy(), fx[f"{tag}.{k}"], rtol=1e-5, atol=1e-6, err_msg=k)
    for k, v in ld.items():
        np.testing.assert_allclose(float(v.detach()), fx[f"{tag}.loss.{k}"], rtol=1e-6, err_msg=k)
    for n, v in g.items():
        np.testing.assert_allclose(v.numpy(), fx[f"{tag}.g.{n}"], rtol=1e-4, atol=1e-7, err_msg="grad " + n)
    tr2 = OP.PriorTrainer(cfg, seed, lr=lr)
    tr2.step(zt, zb)
    ld2, _ = OP.losses(tr2.sd, cfg, zt, zb)  # the second forward re-masks the weights, as the fixture's did
    for k, v in ld2.items():
        np.testing.assert_allclose(float(v.detach()), fx[f"{tag}.loss2.{k}"], rtol=2e-6, err_msg="loss2 " + k)
    for k in keys:
        np.testing.assert_allclose(tr2.sd[k].detach().numpy(), fx[f"{tag}.sd1.{k}"], rtol=1e-5, atol=2e-7, err_msg="sd1 " + k)


def _build_hip(cfg, seed, device):
    import movae_amd  # noqa: F401
    from movae_amd.models.pixelcnn_prior import HierarchicalPixelCNN, PixelCNN

    torch.manual_seed(seed)
    a = (cfg["num_embeddings"], cfg["embedding_dim"], cfg["hidden_channels"], cfg["num_layers"])
    return (HierarchicalPixelCNN(*a) if cfg["hierarchical"] else PixelCNN(*a)).to(device).train()


@pytest.mark.parametrize("tag", ["flat", "hier"])
def test_state_dict_surface_matches_reference(tag):
    """Constructor signature, state_dict keys / shapes / order and the init RNG sequence (CPU: no kernel runs)."""
    fx = load_golden("pixelcnn_tiny")
    cfg, seed, _ = _cfg(fx, tag == "hier")
    net = _build_hip(cfg, seed, "cpu")
    keys = [k[len(tag) + 5:] for k in fx.files if k.startswith(f"{tag}.sd0.")]
    sd = net.state_dict()
    assert list(sd.keys()) == keys
    for k in keys:
        assert tuple(sd[k].shape) == fx[f"{tag}.sd0.{k}"].shape and np.array_equal(sd[k].numpy(), fx[f"{tag}.sd0.{k}"]), k
    net.load_state_dict({k: torch.from_numpy(fx[f"{tag}.sd1.{k}"]) for k in keys})  # a reference checkpoint loads


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["flat", "hier"])
def test_hip_prior_forward_backward_and_step(tag, gpu_device):
    from movae_amd.optim import FusedAdam, clip_grad_norm_

    fx = load_golden("pixelcnn_tiny")
    hier = tag == "hier"
    cfg, seed, lr = _cfg(fx, hier)
    net = _build_hip(cfg, seed, gpu_device)
    zt, zb = _codes(fx, hier)
    zt = zt.to(gpu_device)
    zb = zb.to(gpu_device) if zb is not None else None
    K = cfg["num_embeddings"]
    if hier:
        o = net(zt, zb)
        ld = net.loss_function(zt, zb)
        for k in ("logits_top", "logits_bottom"):
            assert tuple(o[k].shape) == fx[f"{tag}.{k}"].shape
            np.testing.assert_allclose(o[k].detach().cpu().numpy(), fx[f"{tag}.{k}"], rtol=2e-4, atol=2e-5, err_msg=k)
    else:
        logits = net(zt)
        assert tuple(logits.shape) == fx[f"{tag}.logits"].shape  # [B, K, H, W]
        np.testing.assert_allclose(logits.detach().cpu().numpy(), fx[f"{tag}.logits"], rtol=2e-4, atol=2e-5)
        # the reference's own expression on the NCHW view must agree with the fused kernel
        ref_expr = torch.nn.functional.cross_entropy(logits.permute(0, 2, 3, 1).reshape(-1, K).detach(), zt.reshape(-1))
        ld = {"total_loss": net.loss(zt)}
        np.testing.assert_allclose(ld["total_loss"].item(), ref_expr.item(), rtol=1e-6)
    assert list(ld.keys()) == [k[len(tag) + 6:] for k in fx.files if k.startswith(f"{tag}.loss.")]
    for k, v in ld.items():
        np.testing.assert_allclose(v.item(), fx[f"{tag}.loss.{k}"], rtol=2e-5, err_msg=k)
    opt = FusedAdam(net.parameters(), lr=lr, weight_decay=0.0)
    opt.zero_grad()
    ld["total_loss"].backward()
    for n, p in net.named_parameters():
        want = fx[f"{tag}.g.{n}"]
        got = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-5 * max(1e-3, float(np.abs(want).max())), err_msg="grad " + n)
    gn = clip_grad_norm_(net.parameters(), max_norm=1.0)
    np.testing.assert_allclose(float(gn), float(fx[f"{tag}.gnorm"]), rtol=1e-4)
    opt.step()
    ld2 = net.loss_function(zt, zb) if hier else {"total_loss": net.loss(zt)}
    for k, v in ld2.items():
        np.testing.assert_allclose(v.item(), fx[f"{tag}.loss2.{k}"], rtol=5e-5, err_msg="loss2 " + k)
    sd1 = net.state_dict()
    for k in [k[len(tag) + 5:] for k in fx.files if k.startswith(f"{tag}.sd1.")]:
        np.testing.assert_allclose(sd1[k].detach().cpu().numpy(), fx[f"{tag}.sd1.{k}"], rtol=2e-4, atol=2e-6, err_msg="sd1 " + k)


@pytest.mark.gpu
def test_prior_training_stage_on_vq_codes(gpu_device, tmp_path):
    """main.py:890-1085 on the HIP path: codes extracted once from a (frozen) VQ-VAE into the in-memory stand-in of the LMDB
    cache, a PixelCNN trained on them for a few epochs (loss must fall), ancestral sampling decodes to images; `--prior_type
    pixelsnail` is refused by name."""
    import movae_amd  # noqa: F401
    from movae_amd import prior as P
    from movae_amd import train

    argv = ["--dataset", "synthetic_cifar10", "--arch", "vq_vae", "--embedding_dim", "8", "--num_embeddings", "16", "--hidden_dims", "16", "32",
            "--batch_size", "32", "--max_items", "128", "--epochs", "1", "--pixelcnn_epochs", "6", "--pixelcnn_hidden_channels", "16",
            "--pixelcnn_num_layers", "2", "--pixelcnn_lr", "3e-3", "--save_path", str(tmp_path), "--seed", "1", "--device", "cuda:0",
            "--eval_freq", "0"]
    args = train.parse_args(argv)
    train.set_seed(args.seed)
    hist = train.main(args)
    assert len(hist) == 1
    rec = P.LAST_RUN
    assert rec["use_cache"] and rec["n_codes"] == 128 and len(rec["epoch_losses"]) == 6
    assert rec["epoch_losses"][-1] < rec["epoch_losses"][0] < np.log(16) * 1.2
    ck = list(tmp_path.rglob("final_prior.pth"))
    assert len(ck) == 1 and list(tmp_path.rglob("best_prior.pth"))
    imgs = rec["samples"]
    assert tuple(imgs.shape) == (4, 3, 32, 32) and torch.isfinite(imgs).all()
    with pytest.raises(NotImplementedError):
        P.build_prior(rec["net"], train.parse_args(argv + ["--prior_type", "pixelsnail"]), gpu_device)
