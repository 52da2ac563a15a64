"""Shared plumbing of the four hot-path models: loss-weight validation, summaries, noise source."""
import os

import torch
import torch.nn as tnn

from .. import nn as mnn
from .. import ops


def resolve_lambda_weights(owner, objectives, lambda_weights, defaults, list_order=None):
    """dict | list | None -> dict keyed like `objectives` (models/vae.py:53-81 and siblings):
    list length / dict keys are validated with the reference's exception types."""
    keys = list(objectives.keys())
    order = list_order or keys
    if lambda_weights is None:
        return dict(defaults)
    if isinstance(lambda_weights, list):
        if len(lambda_weights) != len(keys):
            raise ValueError(f"{owner} requires {len(keys)} lambda_weights ({', '.join(order)}), got {len(lambda_weights)}")
        return {k: lambda_weights[i] for i, k in enumerate(order)}
    if isinstance(lambda_weights, dict):
        expected, provided = set(keys), set(lambda_weights.keys())
        if expected != provided:
            missing, extra = expected - provided, provided - expected
            msg = "lambda_weights keys must match objectives keys. "
            if missing:
                msg += f"Missing: {missing}. "
            if extra:
                msg += f"Extra: {extra}."
            raise ValueError(msg)
        return lambda_weights
    raise TypeError(f"lambda_weights must be dict or list, got {type(lambda_weights)}")


class LazyDict(dict):
    """A forward() result whose function-valued entries are computed on first read.  VQ-VAE-2's `commitment_loss` /
    `embedding_loss` are sums of the two codebooks' terms (models/vq_vae2.py:291-292): loss_function consumes the four terms
    directly, so in a training step nobody reads the sums and their launches are never made; any other reader (`out[key]`,
    .get, .items(), .values()) sees ordinary tensors."""

    def _resolve(self, key, v):
        if callable(v) and not isinstance(v, torch.Tensor) and getattr(v, "__name__", "") == "<lambda>":
            v = v()
            dict.__setitem__(self, key, v)
        return v

    def __getitem__(self, key):
        return self._resolve(key, dict.__getitem__(self, key))

    def get(self, key, default=None):
        return self[key] if key in self else default

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]


class LazyScalar:
    """A number that lives on the device until somebody needs it on the host.

    The reference's VQ models return `codebook_usage_percentage` as a Python float computed with `.item()` inside
    forward (models/vq_vae.py:110-124,353), which parks the host in the middle of every step until the encoder has
    finished; the backward launches can only be queued afterwards.  The models here return this object instead: it
    converts (one device read) when it is formatted, compared or used in arithmetic -- e.g. by the meter update at the
    end of the step (main.py:220) -- and it is what lets the whole step be captured into a hipGraph."""

    __slots__ = ("_terms", "_scale")

    def __init__(self, tensors, scale):
        self._terms, self._scale = list(tensors), float(scale)

    def __float__(self):
        return float(sum(float(t.item()) for t in self._terms) * self._scale)

    def item(self):
        return float(self)

    def __format__(self, spec):
        return format(float(self), spec)

    def __repr__(self):
        return repr(float(self))

    def __mul__(self, o):
        return float(self) * o

    __rmul__ = __mul__

    def __add__(self, o):
        return float(self) + o

    __radd__ = __add__

    def __sub__(self, o):
        return float(self) - o

    def __rsub__(self, o):
        return o - float(self)

    def __truediv__(self, o):
        return float(self) / o

    def __rtruediv__(self, o):
        return o / float(self)

    def __eq__(self, o):
        return float(self) == o

    def __lt__(self, o):
        return float(self) < o

    def __le__(self, o):
        return float(self) <= o

    def __gt__(self, o):
        return float(self) > o

    def __ge__(self, o):
        return float(self) >= o

    __hash__ = None


def activation_module(name):
    if name not in mnn.ACTIVATIONS:
        raise ValueError(f"recons_activation {name} not supported")
    return mnn.ACTIVATIONS[name]()


class HotPathModel(tnn.Module):
    """Protocol consumed by the training loop (main.py:148,159-160,168,180-181): `.objectives`,
    `.features`, `.lambda_weights`, forward -> dict, loss_function -> dict ending in total_loss."""

    #: True when forward / loss_function neither read device values on the host nor bake per-step Python scalars into
    #: their launches, i.e. the step can be captured into a hipGraph (train.GraphedTrainStep)
    graph_safe = False

    def prepare_for_graph(self):
        """Hook called once before capture; models with per-step host state move it to the device here.  The reparameterisation
        noise is drawn inside the reparameterisation kernel from here on (ops.ReparameterizeRNG): torch.randn_like under capture
        costs three extra launches per replay.  MOVAE_DEVICE_RNG=0 keeps torch's generator."""
        if os.environ.get("MOVAE_DEVICE_RNG", "1") != "0":
            self.noise_on_device = True
            params = list(self.parameters())
            if params and params[0].is_cuda:
                self._noise_state(params[0].device)  # (exists before the capture's state snapshot: train._state_tensors)

    #: draw the reparameterisation noise in the kernel (set by prepare_for_graph); `_noise_state_t`: device int64 {seed, draws}
    noise_on_device = False
    _noise_state_t = None

    def _recon(self, fn, inputs, recons, weight, plain_cls):
        """The reconstruction objective; in the PLAIN model `plain_cls` (not a subclass that adds losses on `recons`: those would be
        further readers) it is told that it is the one reader of the decoder's output activation (objectives._make: out_act)."""
        if type(self) is plain_cls and getattr(fn, "kind", None) in ops.L.RECON:
            return fn(inputs, recons, weight, out_act=True)
        return fn(inputs, recons, weight)

    def _noise_state(self, device):
        st = self._noise_state_t
        if st is None or st.device != device:
            seed = torch.initial_seed()  # the run's torch.manual_seed (read, not drawn: the host generator's stream is untouched)
            if torch.distributed.is_available() and torch.distributed.is_initialized():
                seed ^= (torch.distributed.get_rank() + 1) * 0x9E3779B97F4A7C15  # ranks draw different noise
            seed &= (1 << 63) - 1
            st = self._noise_state_t = torch.tensor([seed, 0], dtype=torch.int64, device=device)
        return st

    def _reparameterize(self, mu, log_var):
        """z = mu + exp(0.5 log_var) * eps (models/vae.py:196-209): eps from eps_override (tests), from the kernel's own
        generator (graph mode), or from torch.randn_like like the reference."""
        if self.eps_override is None and self.noise_on_device and mu.is_cuda:
            return ops.reparameterize_rng(mu, log_var, self._noise_state(mu.device))
        return ops.reparameterize(mu, log_var, self._noise_like(mu))

    #: set to a tensor to replace torch.randn_like in reparameterize (seed-parity tests: the CPU and
    #: HIP generators draw different streams, SURVEY section 7 "hard parts")
    eps_override = None

    def _noise_like(self, t):
        if self.eps_override is not None:
            return self.eps_override.to(device=t.device, dtype=t.dtype)
        return torch.randn_like(t)

    def total_trainable_params(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    def print_model_summary(self):
        """Layer table in place of torchsummary (absent offline); like the reference it runs one
        eval-mode forward on a 2-image batch and returns None."""
        was_training = self.training
        try:
            self.train(False)
            dev = next(self.parameters()).device
            x = torch.zeros(2, self.in_channels, self.input_size, self.input_size, device=dev)
            with torch.no_grad():
                self(x)
            print(f"{'parameter':<44}{'shape':<24}{'count':>10}")
            for n, p in self.named_parameters():
                print(f"{n:<44}{str(tuple(p.shape)):<24}{p.numel():>10}")
            print(f"Total trainable params: {self.total_trainable_params():,}")
        except Exception as e:  # noqa: BLE001 -- the reference swallows summary errors too (models/vae.py:279-281)
            print(f"Error printing model summary: {e}")
        finally:
            self.train(was_training)
        return None


def nchw_view(x_nhwc):
    """NHWC buffer -> logical NCHW tensor (what callers of the reference models expect), zero copy."""
    return x_nhwc.permute(0, 3, 1, 2)
