"""nn.Module shells whose parameters carry the reference's names / shapes / init sequence and whose
forward passes run on the HIP ops (ops.py).  All modules consume and produce NHWC tensors
([N, H, W, C]); the models convert at their NCHW boundary.

Initialisation replays torch.nn's reset_parameters on a temporary contiguous tensor in the order
the reference constructors create their layers, so a given seed yields bit-identical parameters
(models/vae.py:117-173 etc.), then stores conv weights channels_last.
"""
import os

import torch
import torch.nn as tnn

from . import ops

#: MOVAE_FUSE_BN=0: every training-mode BatchNorm runs as its own statistics + apply passes (the pre-fusion path)
FUSE_BN = os.environ.get("MOVAE_FUSE_BN", "1") != "0"

LRELU_SLOPE = 0.01  # nn.LeakyReLU() default used everywhere in the reference


def _channels_last_param(t):
    return tnn.Parameter(t.detach().clone().contiguous(memory_format=torch.channels_last))


def _take_lazy(x, fusion):
    """(fusion, raw tensor) for a conv whose input is an unmaterialised BatchNorm / activation output (ops.LazyBN / ops.LazyAct): the
    operand transform -- and a LazyAct's ActLink -- join whatever the caller already asked of this conv."""
    if fusion is None:
        fusion = x.fusion()
    elif fusion.in_scale is None:
        fusion.in_scale, fusion.in_shift, fusion.in_slope, fusion.link = x.scale, x.shift, x.slope, x.link
        if getattr(x, "act_link", None) is not None:
            fusion.act_in = x.act_link
    return fusion, x.y


class Conv2d(tnn.Module):
    def __init__(self, cin, cout, k, stride=1, padding=0, bias=True):
        super().__init__()
        ref = tnn.Conv2d(cin, cout, k, stride=stride, padding=padding, bias=bias)  # draws the init RNG sequence
        self.weight = _channels_last_param(ref.weight)
        self.bias = tnn.Parameter(ref.bias.detach().clone()) if bias else None
        self.stride, self.padding, self.kernel_size = stride, padding, k
        self.in_channels, self.out_channels = cin, cout

    def forward(self, x, act=None, feeds_batchnorm=False, fusion=None):
        if isinstance(x, ops.LazyBN):  # the producer's BatchNorm + activation is applied while this conv loads its input
            fusion, x = _take_lazy(x, fusion)
        return ops.conv2d(x, self.weight, self.bias, self.stride, self.padding, act, LRELU_SLOPE, feeds_batchnorm, fusion)

    def extra_repr(self):
        return f"{self.in_channels}, {self.out_channels}, k={self.kernel_size}, s={self.stride}, p={self.padding}"


class ConvTranspose2d(tnn.Module):
    def __init__(self, cin, cout, k, stride=1, padding=0, output_padding=0):
        super().__init__()
        ref = tnn.ConvTranspose2d(cin, cout, k, stride=stride, padding=padding, output_padding=output_padding)
        self.weight = _channels_last_param(ref.weight)
        self.bias = tnn.Parameter(ref.bias.detach().clone())
        self.stride, self.padding, self.output_padding, self.kernel_size = stride, padding, output_padding, k
        self.in_channels, self.out_channels = cin, cout

    def forward(self, x, act=None, feeds_batchnorm=False, fusion=None):
        if isinstance(x, ops.LazyBN):
            fusion, x = _take_lazy(x, fusion)
        return ops.conv_transpose2d(x, self.weight, self.bias, self.stride, self.padding, self.output_padding, act, LRELU_SLOPE,
                                    feeds_batchnorm, fusion)

    def extra_repr(self):
        return (f"{self.in_channels}, {self.out_channels}, k={self.kernel_size}, s={self.stride}, p={self.padding}, "
                f"op={self.output_padding}")


class Linear(tnn.Module):
    def __init__(self, fin, fout):
        super().__init__()
        ref = tnn.Linear(fin, fout)
        self.weight = tnn.Parameter(ref.weight.detach().clone())
        self.bias = tnn.Parameter(ref.bias.detach().clone())
        self.in_features, self.out_features = fin, fout

    def forward(self, x):
        return ops.linear(x, self.weight, self.bias)

    def extra_repr(self):
        return f"{self.in_features}, {self.out_features}"


def linear_pair(x, lin1, lin2):
    """(lin1(x), lin2(x)) for two Linear modules that read the same input (fc_mu, fc_var): one launch forward, two backward."""
    return ops.linear_pair(x, lin1.weight, lin1.bias, lin2.weight, lin2.bias)


class BatchNorm2d(tnn.Module):
    """nn.BatchNorm2d(affine, track_running_stats) semantics: eps 1e-5, momentum 0.1."""

    def __init__(self, c, eps=1e-5, momentum=0.1):
        super().__init__()
        self.weight = tnn.Parameter(torch.ones(c))
        self.bias = tnn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.eps, self.momentum, self.num_features = eps, momentum, c

    def forward(self, y, act=None):
        # num_batches_tracked += 1 happens inside the statistics kernel (training mode only)
        return ops.batch_norm_act(y, self.weight, self.bias, self.running_mean, self.running_var, self.training, self.eps,
                                  self.momentum, act, LRELU_SLOPE, self.num_batches_tracked)

    def forward_lazy(self, y, act, fusion):
        """Training mode, fused: statistics from the producer conv's epilogue (fusion), output left unmaterialised (ops.LazyBN)."""
        return ops.batch_norm_lazy(y, self.weight, self.bias, self.running_mean, self.running_var, self.num_batches_tracked, self.eps,
                                   self.momentum, act, LRELU_SLOPE, fusion)

    def extra_repr(self):
        return f"{self.num_features}"


class _Act(tnn.Module):
    kind = None

    def forward(self, x):
        return ops.activation(x, self.kind, LRELU_SLOPE)


class LeakyReLU(_Act):
    kind = "lrelu"


class ReLU(_Act):
    kind = "relu"


class Tanh(_Act):
    kind = "tanh"


class Sigmoid(_Act):
    kind = "sigmoid"


class Identity(_Act):
    kind = None


ACTIVATIONS = {"tanh": Tanh, "sigmoid": Sigmoid, "none": Identity}


def _starts_with_conv(m):
    """Is the first op of `m` a conv (m itself, or the first module of a Stack, recursively)?  Such a module takes an ActLink."""
    while isinstance(m, Stack) and len(m) > 0:
        m = m[0]
    return isinstance(m, (Conv2d, ConvTranspose2d))


class Stack(tnn.Sequential):
    """nn.Sequential (same child names, hence same state_dict keys) whose forward fuses
    conv -> [batchnorm] -> activation runs into the conv / batch-norm kernels' epilogues."""

    def accepts_res_in(self):
        """Can this Stack's first op take the identity cotangent of a residual block (ops.ResCarrier)?  A conv can, and so can a
        stand-alone ReLU / LeakyReLU that the next conv is linked to."""
        mods = list(self)
        if not mods:
            return False
        if isinstance(mods[0], Stack):
            return mods[0].accepts_res_in()
        if isinstance(mods[0], (Conv2d, ConvTranspose2d)):
            return True
        return (isinstance(mods[0], _Act) and mods[0].kind in ("lrelu", "relu") and len(mods) > 1 and
                isinstance(mods[1], (Conv2d, ConvTranspose2d)))

    def accepts_res_out(self):
        """Does this Stack end with a plain conv (no BatchNorm, no activation after it) that can add a residual block's input?"""
        mods = list(self)
        return bool(mods) and isinstance(mods[-1], (Conv2d, ConvTranspose2d))

    def forward(self, x, act_in=None, res_in=None, res_out=None):
        """x: an NHWC tensor or an ops.LazyBN (the unmaterialised output of a fused BatchNorm); may return either -- a LazyBN
        leaves a Stack only when its last module is a fused BatchNorm (+ activation), and is handed on to Stacks and convs
        as it is; every other module receives the materialised tensor.

        Activation links (ops.ActLink): the output of a conv + activation pair that is handed to exactly one next conv (the next
        module of this Stack, or the first conv of the Stack that follows in an enclosing Stack: `act_in` / `self._out_link`)
        lets that conv's input-gradient pass apply the activation derivative, so the pair's backward has no pass of its own."""
        mods = list(self)
        i, n = 0, len(mods)
        link = act_in  # the link whose activation output `x` currently is (None: x is something else)
        self._out_link = None
        if res_in is not None and (isinstance(x, ops.LazyBN) or not self.accepts_res_in()):
            raise ValueError("res_in: the first op of this Stack cannot take a residual block's identity cotangent")
        if res_out is not None and not self.accepts_res_out():
            raise ValueError("res_out: this Stack does not end with a plain conv")
        while i < n:
            m = mods[i]
            rin = res_in if i == 0 else None  # (only the op applied to the block's input)
            if isinstance(m, Stack):
                x = m(x, link, rin)
                link = m._out_link
                m._out_link = None
                i += 1
            elif isinstance(m, (Conv2d, ConvTranspose2d)):
                nxt = mods[i + 1] if i + 1 < n else None
                if isinstance(nxt, BatchNorm2d):
                    act = mods[i + 2] if i + 2 < n and isinstance(mods[i + 2], _Act) else None
                    kind = act.kind if act is not None else None
                    if FUSE_BN and nxt.training and kind in (None, "lrelu", "relu") and m.out_channels % 4 == 0:
                        fusion = ops.ConvFusion(want_stats=True, act_in=link, res_in=rin)
                        # (a producer that can finishes this BatchNorm inside its own launch: ops.ConvFusion.bn_fin)
                        fusion.bn_fin = (nxt.weight, nxt.bias, nxt.running_mean, nxt.running_var, nxt.num_batches_tracked, nxt.eps, nxt.momentum)
                        if isinstance(x, ops.LazyBN):
                            fusion, x = _take_lazy(x, fusion)
                        x = nxt.forward_lazy(m(x, None, True, fusion), kind, fusion)
                    else:
                        x = nxt(m(x, None, nxt.training, ops.ConvFusion(act_in=link, res_in=rin) if (link is not None or rin is not None) else None), kind)
                    link = None
                    i += 3 if act is not None else 2
                elif isinstance(nxt, _Act):
                    # (a tanh / sigmoid pair that ENDS the Stack gets a link too: the decoder's output activation, whose one reader
                    # is the reconstruction loss -- ops.VAELosses applies the derivative in its backward kernel)
                    ends = i + 2 == n and nxt.kind in ("tanh", "sigmoid")
                    out = ops.ActLink() if (nxt.kind in ("lrelu", "relu") or ends) else None
                    # (a LazyBN / LazyAct input adds its operand transform to this request: _take_lazy)
                    x = m(x, nxt.kind, False, ops.ConvFusion(act_in=link, act_out=out, res_in=rin)
                          if (link is not None or out is not None or rin is not None) else None)
                    link = out
                    i += 2
                else:
                    rout = res_out if i == n - 1 else None  # (the residual branch's last conv adds the block's input)
                    x = m(x, None, False, ops.ConvFusion(act_in=link, res_in=rin, res_out=rout)
                          if (link is not None or rin is not None or rout is not None) else None)
                    link = None
                    i += 1
            elif (isinstance(m, _Act) and m.kind in ("lrelu", "relu") and i + 1 < n and _starts_with_conv(mods[i + 1])
                  and not isinstance(x, ops.LazyBN)):
                # a stand-alone activation whose output only the next conv reads: that conv's input gradient applies its derivative
                link = ops.ActLink()
                nxt = mods[i + 1]
                if ops.LAZY_ACT and isinstance(nxt, (Conv2d, ConvTranspose2d)) and x.shape[-1] % 4 == 0:
                    # not written at all: the conv applies it while loading (ops.LazyAct); the link travels inside the LazyAct
                    x = ops.LazyAct(x, m.kind, LRELU_SLOPE, link, rin)
                    link = None
                else:
                    x = ops.activation(x, m.kind, LRELU_SLOPE, link, rin)
                i += 1
            else:
                x = m(ops.materialize(x))
                link = None
                i += 1
        self._out_link = link
        return x


def chain(x, *stacks):
    """stacks[-1](... stacks[0](x)) for Stacks that a model keeps as separate attributes (decoder -> final_layer): the activation
    link of each Stack's output is handed to the next one, as an enclosing Stack would.  Each intermediate result must have no other
    reader (ops.ActLink)."""
    link = None
    for s in stacks:
        x = s(x, link)
        link, s._out_link = s._out_link, None
    return x


class Codebook(tnn.Module):
    """nn.Embedding(K, D) drawn like models/vq_vae.py:24-25 (normal_ first, then uniform(-1/K, 1/K))."""

    def __init__(self, K, D):
        super().__init__()
        ref = tnn.Embedding(K, D)
        ref.weight.data.uniform_(-1.0 / K, 1.0 / K)
        self.weight = tnn.Parameter(ref.weight.detach().clone())
        self.num_embeddings, self.embedding_dim = K, D

    def forward(self, code):
        return self.weight[code]  # index gather for decode_code / sampling (outside the training path)
