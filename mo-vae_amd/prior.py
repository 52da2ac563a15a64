"""Prior stage that follows VQ-VAE training -- drop-in for the reference's main.py:890-1085 (train_pixelcnn_prior,
generate_samples_vq_with_prior) on the HIP kernels.

The reference extracts every training image's code grid once into an LMDB file (utils/vq_codes_lmdb.py:20-103) and trains the
prior from a Dataset over it (:106-172).  `lmdb` is not importable here (and its on-disk format is a pickle per sample), so the
cache is kept IN MEMORY: `VQCodeDataset` has the reference Dataset's `__len__` / `__getitem__` contract (a long tensor `z`, or a
pair `(z_top, z_bottom)` for the hierarchical models) -- parity unpinned as a storage format, the tensors it yields are the
golden-pinned models' own outputs.  `--no_prior_lmdb_codes` re-extracts the codes on the fly each batch, like the reference.
"""
import os

import torch

from .models.pixelcnn_prior import HierarchicalPixelCNN, PixelCNN
from .optim import FusedAdam, clip_grad_norm_

#: what the last train_pixelcnn_prior call did (tests and callers that want the numbers; the reference only prints them)
LAST_RUN = {}


class VQCodeDataset(torch.utils.data.Dataset):
    """utils/vq_codes_lmdb.py:106-172 without the file: item i is the code grid(s) of training sample i."""

    def __init__(self, codes, is_hierarchical, transform_index=None):
        self.is_hierarchical, self.transform_index = is_hierarchical, transform_index
        if is_hierarchical:
            self.z_top, self.z_bottom = codes
            assert self.z_top.size(0) == self.z_bottom.size(0)
            self.meta = {"is_hierarchical": True, "z_top_shape": tuple(self.z_top.shape[1:]),
                         "z_bottom_shape": tuple(self.z_bottom.shape[1:]), "dtype": "int64"}
        else:
            (self.z,) = codes
            self.meta = {"is_hierarchical": False, "z_shape": tuple(self.z.shape[1:]), "dtype": "int64"}

    def __len__(self):
        return (self.z_top if self.is_hierarchical else self.z).size(0)

    def __getitem__(self, idx):
        if idx < 0 or idx >= len(self):
            raise IndexError(f"Index {idx} not found in the code cache")
        t = self.transform_index
        if self.is_hierarchical:
            a, b = self.z_top[idx].long(), self.z_bottom[idx].long()
            return (t(a), t(b)) if t is not None else (a, b)
        z = self.z[idx].long()
        return t(z) if t is not None else z


@torch.no_grad()
def extract_codes(net, train_loader, device, is_hierarchical):
    """One pass of the frozen VQ model over the training set (utils/vq_codes_lmdb.py:47-93): code grids on the host."""
    net.eval()
    tops, bots = [], []
    # held as int16 / int32 on the host (a code is < num_embeddings): an int64 grid per ImageNet-scale VQ-VAE-2 sample would be
    # ~50 GB where the reference spills to LMDB; VQCodeDataset.__getitem__ hands out .long() like the reference's Dataset
    store = torch.int16 if int(getattr(net, "num_embeddings", 1 << 20)) <= (1 << 15) else torch.int32
    for images, _ in train_loader:
        images = images.to(device)
        if is_hierarchical:
            cd = net.get_code_indices(images)
            tops.append(cd["indices_top"].to(store).cpu())
            bots.append(cd["indices_bottom"].to(store).cpu())
        else:
            tops.append(net.get_code_indices(images).to(store).cpu())
    if is_hierarchical:
        return VQCodeDataset((torch.cat(tops), torch.cat(bots)), True)
    return VQCodeDataset((torch.cat(tops),), False)


def is_hierarchical_arch(arch):
    return (arch or "").lower() in ("vq_vae2", "gg_vq_vae2")


def build_prior(net, args, device):
    """main.py:917-957."""
    if getattr(args, "prior_type", "pixelcnn").lower() == "pixelsnail":
        raise NotImplementedError("--prior_type pixelsnail (causal self-attention, models/pixelcnn_prior.py:95-259) is outside the "
                                  "MI355X hot-path scope (SURVEY 8f.4 names the PixelCNN prior); use --prior_type pixelcnn")
    kw = dict(num_embeddings=net.num_embeddings, embedding_dim=net.embedding_dim,
              hidden_channels=getattr(args, "pixelcnn_hidden_channels", 128), num_layers=getattr(args, "pixelcnn_num_layers", 15))
    cls = HierarchicalPixelCNN if is_hierarchical_arch(getattr(args, "arch", "vae")) else PixelCNN
    return cls(**kw).to(device)


def train_pixelcnn_prior(net, train_loader, device, args, save_root):
    """main.py:890-1043.  Returns the trained prior (eval mode)."""
    hier = is_hierarchical_arch(getattr(args, "arch", "vae"))
    epochs = getattr(args, "pixelcnn_epochs", 100)
    lr = getattr(args, "pixelcnn_lr", 3e-4)
    net.eval()
    for p in net.parameters():
        p.requires_grad = False
    prior_dir = os.path.join(save_root, "pixelcnn_prior")
    os.makedirs(os.path.join(prior_dir, "checkpoints"), exist_ok=True)
    prior = build_prior(net, args, device)
    opt = FusedAdam(prior.parameters(), lr=lr, weight_decay=0.0)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=epochs, eta_min=1e-6)
    use_cache = getattr(args, "prior_use_lmdb_codes", True)
    codes_loader, n_codes = None, 0
    if use_cache:
        ds = extract_codes(net, train_loader, device, hier)
        n_codes = len(ds)
        codes_loader = torch.utils.data.DataLoader(ds, batch_size=args.batch_size, shuffle=True, num_workers=0, drop_last=False)
    print(f"Training PixelCNN prior for {epochs} epochs..." + (" (codes extracted once, held in memory)" if use_cache
                                                             else " (extracting codes on-the-fly)"))
    K = net.num_embeddings
    best, epoch_losses = float("inf"), []

    def one_step(z_top, z_bottom=None):
        opt.zero_grad()
        if hier:
            loss = prior.loss_function(z_top, z_bottom)["total_loss"]
        else:
            loss = prior.loss(z_top)  # == F.cross_entropy(logits.permute(0,2,3,1).reshape(-1,K), z.reshape(-1)), main.py:1003
        loss.backward()
        clip_grad_norm_(prior.parameters(), max_norm=1.0)
        opt.step()
        return loss.detach()

    for epoch in range(1, epochs + 1):
        prior.train()
        losses = []  # device scalars: read once per epoch, not once per step (the reference calls .item() every step)
        if codes_loader is not None:
            for batch in codes_loader:
                if hier:
                    losses.append(one_step(batch[0].to(device), batch[1].to(device)))
                else:
                    z = batch if isinstance(batch, torch.Tensor) else batch[0]
                    losses.append(one_step(z.to(device)))
        else:
            for images, _ in train_loader:
                images = images.to(device)
                with torch.no_grad():
                    if hier:
                        cd = net.get_code_indices(images)
                        zt, zb = cd["indices_top"], cd["indices_bottom"]
                    else:
                        zt, zb = net.get_code_indices(images), None
                losses.append(one_step(zt, zb))
        sched.step()
        if hasattr(opt, "sync_hyper"):
            opt.sync_hyper()
        avg = float(torch.stack(losses).mean().item()) if losses else 0.0
        epoch_losses.append(avg)
        if avg < best:
            best = avg
            torch.save({"epoch": epoch, "model_state_dict": {k: v.contiguous() for k, v in prior.state_dict().items()}, "loss": best},
                       os.path.join(prior_dir, "checkpoints", "best_prior.pth"))
    torch.save({"model_state_dict": {k: v.contiguous() for k, v in prior.state_dict().items()}, "epoch": epochs},
               os.path.join(prior_dir, "checkpoints", "final_prior.pth"))
    print(f"PixelCNN prior training complete. Best loss: {best:.4f}. Saved to {prior_dir}")
    prior.eval()
    samples = generate_samples_vq_with_prior(net, prior, getattr(args, "num_vis_samples", 4), device,
                                             getattr(args, "pixelcnn_temperature", 1.0))
    LAST_RUN.clear()
    LAST_RUN.update(net=net, prior=prior, epoch_losses=epoch_losses, use_cache=use_cache, n_codes=n_codes, samples=samples.detach().cpu(),
                    num_embeddings=K)
    return prior


@torch.no_grad()
def generate_samples_vq_with_prior(net, prior, num_samples, device, temperature=1.0):
    """main.py:1046-1085."""
    net.eval()
    prior.eval()
    if hasattr(prior, "sample_with_vqvae2"):
        return prior.sample_with_vqvae2(vqvae2_model=net, batch_size=num_samples, device=device, temperature=temperature)
    s = net.latent_spatial_dim
    z = prior.sample(batch_size=num_samples, height=s, width=s, device=device, temperature=temperature)
    q = net.vq_layer.embed_code(z).permute(0, 3, 1, 2)
    return net.decode(q)
