"""CPU-side checks: the C-ABI library loads and exports every symbol include/movae.h declares,
host logic (factories, validation, init replay) behaves like the reference, and the product path
refuses to run without a GPU instead of falling back."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, cfg_from_meta, load_golden, meta_of


@pytest.fixture(scope="module")
def pkg():
    import movae_amd

    return movae_amd


def test_library_exports_every_declared_symbol(pkg):
    hdr = open(os.path.join(ROOT, "include", "movae.h")).read()
    names = set(re.findall(r"\b(movae_[a-z0-9_]+)\s*\(", hdr, flags=re.I))
    names -= {"movae_stream_t"}
    assert len(names) >= 40
    lib = pkg.load_library()
    raw = ctypes.CDLL(os.path.join(ROOT, "mo-vae_amd", "libmovae_hip.so"))
    for n in sorted(names):
        assert hasattr(raw, n), f"{n} declared in movae.h but not exported"
    from movae_amd import _lib

    assert set(_lib.SIGNATURES) == names, set(_lib.SIGNATURES) ^ names
    assert lib.movae_version() >= 100
    assert lib.movae_bn_ws_bytes(1024, 64) > 0 and lib.movae_gram_ws_bytes(4, 1 << 20) > 0


def test_deferred_reduce_policy_names_and_switches(pkg):
    """Host logic of the deferred weight-gradient reduces (no GPU): every call the policy lets pass while a reduce may be parked is
    an entry point that exists and is one of the backward's own ops -- never an aggregation / optimizer / forward-statistics call,
    which must flush first -- and the switch nests and restores."""
    from movae_amd import _lib as L
    from movae_amd import ops

    assert L.DEFER_PASS <= set(L.SIGNATURES), L.DEFER_PASS - set(L.SIGNATURES)
    readers = {n for n in L.SIGNATURES if n.startswith(("movae_gram", "movae_combine", "movae_weights", "movae_adam", "movae_clip", "movae_sumsq",
                                                          "movae_scale_by", "movae_gd_"))}
    assert readers and not (readers - {"movae_combine_losses_bwd"}) & L.DEFER_PASS
    assert not any(n.endswith(("_fwd", "_fwd_f", "_fwd_mse")) for n in L.DEFER_PASS)
    assert L.DEFER_ON[0] is False
    with ops.deferred_reduces(enabled=False):
        assert L.DEFER_ON[0] is False
    assert L.DEFER_ON[0] is False


def test_no_cpu_fallback(pkg):
    from movae_amd import aggregation, ops

    with pytest.raises(RuntimeError, match="no CPU"):
        ops.conv2d(torch.zeros(1, 4, 4, 3), torch.zeros(8, 3, 3, 3))
    with pytest.raises(RuntimeError, match="no CPU"):
        aggregation.UPGrad()(torch.zeros(2, 8))
    # nothing under the package imports the oracle
    for dp, _, fs in os.walk(os.path.join(ROOT, "mo-vae_amd")):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


class Args:
    def __init__(self, **kw):
        self.__dict__.update(kw)


@pytest.mark.parametrize("tag", ["vae_tiny", "vae_tiny_bce", "vae_1x1", "vq_vae_tiny", "vq_vae2_tiny", "betatc_vae_tiny"])
def test_constructors_replay_reference_init_and_state_dict_layout(pkg, tag):
    from movae_amd.models import get_network

    fx = load_golden(tag)
    m = meta_of(fx)
    c = cfg_from_meta(m)
    args = Args(arch=c["arch"], batch_size=c["batch_size"], dataset_size=c["dataset_size"], recons_objective=c["recons_objective"],
                recons_activation=None, loss_weights=None,
                **{k: v for k, v in c.items() if k in ("latent_dim", "hidden_dims", "embedding_dim", "num_embeddings",
                                                        "num_residual_layers", "anneal_steps")})
    torch.manual_seed(int(m["seed"]))
    net = get_network(c["input_size"], 3, args, None)
    sd = net.state_dict()
    keys = [k[4:] for k in fx.files if k.startswith("sd0.")]
    assert list(sd.keys()) == keys
    for k in keys:
        assert tuple(sd[k].shape) == fx["sd0." + k].shape, k
        assert np.array_equal(sd[k].numpy(), fx["sd0." + k]), k
    assert list(net.features) == [str(s) for s in fx["features"]]
    assert list(net.objectives.keys()) == [str(s) for s in fx["objectives"]]
    want_lw = dict(str(s).split("=") for s in fx["lambda_weights"])
    assert {k: float(v) for k, v in want_lw.items()} == {k: float(v) for k, v in net.lambda_weights.items()}
    # reference checkpoints (contiguous tensors) load into the channels_last parameters
    net.load_state_dict({k: torch.from_numpy(fx["sd0." + k]) for k in keys})


def test_factory_errors_and_quirks(pkg):
    from movae_amd.models import VAE, VQVAE2, BetaTCVAE, get_network

    with pytest.raises(ValueError):
        get_network(32, 3, Args(arch="nope", batch_size=1, dataset_size=1), None)
    with pytest.raises(NotImplementedError):
        get_network(32, 3, Args(arch="gg_vq_vae_v8", batch_size=1, dataset_size=1), None)  # step-function loss: no gradient
    with pytest.raises(NotImplementedError):
        get_network(32, 3, Args(arch="gg_vae_v6", batch_size=1, dataset_size=1), None)  # broken in the reference itself
    with pytest.raises(ValueError):
        get_network(32, 3, Args(arch="gg_vae_v4", batch_size=1, dataset_size=1), None)  # not in the reference's factory
    with pytest.raises(ValueError):
        VAE(latent_dim=4, hidden_dims=[4], input_size=8, lambda_weights=[1.0])
    with pytest.raises(ValueError):
        VAE(latent_dim=4, hidden_dims=[4], input_size=8, lambda_weights={"reconstruction_loss": 1.0})
    with pytest.raises(TypeError):
        VAE(latent_dim=4, hidden_dims=[4], input_size=8, lambda_weights=3.0)
    with pytest.raises(ValueError):
        VAE(latent_dim=4, hidden_dims=[4], input_size=8, recons_objective="foo")
    with pytest.raises(ValueError):
        VQVAE2(3, 8, 16, hidden_dims=[16, 32], lambda_weights=[1.0, 2.0])
    with pytest.raises(ValueError):
        BetaTCVAE(3, 4, hidden_dims=[4], input_size=8, lambda_weights=[1.0])
    # bce forces sigmoid whatever activation was asked (utils/objectives.py:26-27)
    v = VAE(latent_dim=4, hidden_dims=[4], input_size=8, recons_objective="bce", recons_activation="tanh")
    assert type(v.final_layer[4]).__name__ == "Sigmoid"
    # kld weight is forced to batch_size / dataset_size by the factory (models/__init__.py:49-55)
    n = get_network(8, 3, Args(arch="vae", batch_size=10, dataset_size=100, latent_dim=4, hidden_dims=[4],
                               loss_weights={"reconstruction_loss": 2.0, "kld_loss": 9.0}), None)
    assert n.lambda_weights == {"reconstruction_loss": 2.0, "kld_loss": 0.1}


def test_aggregator_factory_names(pkg):
    from movae_amd import aggregation as A

    def mk(name):
        return A.make_aggregator(Args(aggregator=name, agg_norm_eps=1e-4, agg_reg_eps=1e-4, mgda_epsilon=1e-5,
                                      mgda_max_iters=250, pref_weights=None))

    assert mk(None) is None and mk("sum") == "sum"
    assert isinstance(mk("upgrad"), A.UPGrad) and isinstance(mk("jd_sum"), A.Sum) and isinstance(mk("mean"), A.Mean)
    for n, nt in [("mgda", "none"), ("mgda_ln", "l2"), ("mgda_gn", "loss"), ("mgda_lgn", "loss+")]:
        assert mk(n).mgda_weighting.norm_type == nt
    for n, sm in [("aligned_mtl", "min"), ("amtl", "min"), ("aligned_mtl_median", "median"), ("aligned_mtl_rmse", "rmse")]:
        assert mk(n)._scale_mode == sm
    assert isinstance(mk("nupgrad"), A.NUPGrad) and mk("nupgrad").gramian_weighting.norm == "min_l2"
    assert isinstance(mk("pnupgrad"), A.PNUPGrad) and mk("pnupgrad").gramian_weighting.prob == 0.5
    c = mk("comfort")
    assert isinstance(c, A.COMFORT) and c.weighting is c._mgda.weighting and c._get_beta() == 1.0  # total_epochs <= 1 -> u
    c.set_epoch(1, 10)
    assert abs(c._get_beta() - 0.01) < 1e-12
    c.set_epoch(10, 10)
    assert abs(c._get_beta() - 1.0) < 1e-12
    assert isinstance(mk("pcgrad"), A.PCGrad) and isinstance(mk("imtlg"), A.IMTLG) and isinstance(mk("dualproj"), A.DualProj)
    assert isinstance(mk("cagrad"), A.CAGrad) and mk("cagrad")._c == 1.0
    with pytest.raises(NotImplementedError):
        mk("nashmtl")
    with pytest.raises(ValueError):
        A.CAGrad(c=-1.0)
    with pytest.raises(ValueError):
        mk("bogus")
    with pytest.raises(ValueError):
        A.MGDA(norm_type="l3")
    a = Args(aggregator=None)
    A.make_aggregator(a)
    assert a.aggregator == "sum"  # main.py:1245-1246


#: configs/celeba-hq/vq_vae2/mgda_ln/bce/config_1.yaml of the reference, as data (keys and values, in file order)
REF_YAML_VQVAE2 = {
    "dataset": "celeba-hq", "data_dir": "../data", "normalize_inputs": False, "arch": "vq_vae2", "embedding_dim": 64,
    "num_embeddings": 512, "hidden_dims": [128, 256],
    "loss_weights": {"reconstruction_loss": 1.0, "embedding_loss": 1.0, "commitment_loss": 0.25}, "recons_objective": "bce",
    "recons_activation": "sigmoid", "epochs": 400, "batch_size": 128, "optimizer": "adam", "lr": "1e-4", "scheduler": "cosine",
    "scheduler_lr_min": "1e-6", "wd": 0.0, "aggregator": "mgda_ln", "seed": 42, "save_path": "logs/", "save_freq": 50,
    "eval_freq": 50, "num_vis_samples": 4,
    "hv_ref": {"reconstruction_loss": 1.1, "commitment_loss": 1.1, "embedding_loss": 1.1}, "use_wandb": True,
    "wandb_project": "mo-vae", "wandb_entity": "rasa_research", "wandb_name": "celeba_hq-vq_vae2-512k-64d-bce-mgda_ln-seed42",
    "wandb_group": "celeba_hq-vq_vae2-512k-64d-bce-mgda_ln",
}


def _yaml_to_argv(config):
    """The conversion runner.py:32-85 applies to a YAML dict (aliases, bool -> flag, dict -> JSON, list -> tokens)."""
    import json

    aliases = {"agg": "aggregator", "wd": "weight_decay", "normalize": "normalize_inputs", "num_samples": "num_vis_samples",
               "norm_eps": "agg_norm_eps", "reg_eps": "agg_reg_eps"}
    argv = []
    for key, value in config.items():
        if key in ("device", "num_workers") or value is None:
            continue
        name = "--" + aliases.get(key, key)
        if isinstance(value, bool):
            if value:
                argv.append(name)
        elif isinstance(value, dict):
            argv += [name, json.dumps(value)]
        elif isinstance(value, list):
            argv += [name] + [str(v) for v in value]
        else:
            argv += [name, str(value)]
    return argv


def test_reference_yaml_argv_parses(pkg, capsys):
    """Any reference YAML -> runner.py argv must parse: the prior / PixelSNAIL / sphere / ViT flags of main.py:1603-1651 are
    accepted with the reference's defaults; the ones only out-of-scope architectures read are ignored with one warning."""
    from movae_amd import train

    a = train.parse_args(_yaml_to_argv(REF_YAML_VQVAE2))
    assert (a.arch, a.aggregator, a.embedding_dim, a.num_embeddings, a.hidden_dims) == ("vq_vae2", "mgda_ln", 64, 512, [128, 256])
    assert a.loss_weights == {"reconstruction_loss": 1.0, "embedding_loss": 1.0, "commitment_loss": 0.25}
    assert a.hv_ref == {"reconstruction_loss": 1.1, "commitment_loss": 1.1, "embedding_loss": 1.1}
    assert (a.lr, a.scheduler_lr_min, a.wd, a.recons_objective, a.recons_activation) == (1e-4, 1e-6, 0.0, "bce", "sigmoid")
    assert a.use_wandb and not a.normalize_inputs and a.seed == 42
    # defaults of the prior stage (main.py:1625-1651)
    assert (a.prior_type, a.pixelcnn_epochs, a.pixelcnn_hidden_channels, a.pixelcnn_num_layers, a.pixelcnn_lr, a.pixelcnn_temperature,
            a.prior_use_lmdb_codes, a.skip_pixelcnn) == ("pixelcnn", 100, 128, 15, 3e-4, 1.0, True, False)
    assert "ignoring" not in capsys.readouterr().out
    b = train.parse_args(["--arch", "vae", "--vit_depth", "12", "--sigma_mix_prob", "0.1", "--patch_size", "2", "--num_classes", "10",
                          "--recursive_kld_anneal_steps", "5", "--pixelsnail_dropout", "0.2", "--no_prior_lmdb_codes",
                          "--lambda_pix_con", "0.3", "--prior_lmdb_map_size_gb", "1", "--prior_force_extract_codes"])
    out = capsys.readouterr().out
    assert "ignoring options of architectures outside the MI355X hot path" in out and "--vit_depth" in out and "--pixelsnail_dropout" in out
    assert b.vit_depth == 12 and b.prior_use_lmdb_codes is False and b.prior_force_extract_codes
    # every option string of the reference's parser is known here
    ref_flags = """--seed --device --data_dir --save_path --epochs --dataset --normalize_inputs --batch_size --num_workers --aggregator --agg
        --agg_norm_eps --agg-norm-eps --norm_eps --norm-eps --agg_reg_eps --agg-reg-eps --reg_eps --reg-eps --mgda_epsilon --mgda-epsilon
        --mgda_max_iters --mgda-max-iters --mgda_min_eigenvalue_eps --mgda-min-eigenvalue-eps --comfort_mgda_norm_type
        --comfort-mgda-norm-type --comfort_mgda_stable --comfort-mgda-stable --comfort_beta_k --comfort_beta_a --comfort_beta_l
        --comfort_beta_u --arch --layer_norm --latent_dim --hidden_dims --num_residual_layers --recons_objective --recons_activation
        --loss_weights --pref_weights --optimizer --momentum --max_grad_norm --lr --wd --weight_decay --scheduler --scheduler_lr_min
        --scheduler_gamma --scheduler_milestones --embedding_dim --num_embeddings --anneal_steps --recursive_kld_anneal_steps
        --sigma_max_angle_deg --sigma_mix_prob --sigma_mix_angle_min_deg --sigma_mix_angle_max_deg --lambda_pix_recon --lambda_pix_con
        --lambda_lat_con --patch_size --vit_embed_dim --vit_depth --vit_num_heads --vit_mixer_depth --num_classes --hv_ref
        --num_vis_samples --save_freq --eval_freq --use_wandb --wandb_project --wandb_entity --wandb_name --wandb_group --wandb_tags
        --max_fid_samples --max_gen_metrics_samples --prior_type --skip_pixelcnn --pixelcnn_epochs --pixelcnn_hidden_channels
        --pixelcnn_num_layers --pixelcnn_lr --pixelcnn_temperature --pixelsnail_num_blocks --pixelsnail_num_res_blocks
        --pixelsnail_num_heads --pixelsnail_dropout --prior_use_lmdb_codes --no_prior_lmdb_codes --prior_force_extract_codes
        --prior_lmdb_map_size_gb""".split()
    known = {s for act in train.build_parser()._actions for s in act.option_strings}
    assert len(ref_flags) == 96 and not [f for f in ref_flags if f not in known]


def test_fuse_struct_layout_matches_the_header(pkg, tmp_path):
    """movae_fuse_t crosses the C ABI by address: the ctypes mirror (_lib.MovaeFuse) must have the header's field offsets and size.
    include/movae.h is plain C: compile a probe with the host compiler and compare."""
    import ctypes as C
    import shutil
    import subprocess

    import movae_amd._lib as L

    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no host C compiler")
    names = [n for n, _ in L.MovaeFuse._fields_]
    src = tmp_path / "probe.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "movae.h"\nint main(void) {\n'
                   '  printf("%zu\\n", sizeof(movae_fuse_t));\n' +
                   "".join(f'  printf("{n} %zu\\n", offsetof(movae_fuse_t, {n}));\n' for n in names) + "  return 0;\n}\n")
    exe = tmp_path / "probe"
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.run([cc, "-I", inc, str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    assert int(out[0]) == C.sizeof(L.MovaeFuse)
    want = {ln.split()[0]: int(ln.split()[1]) for ln in out[1:] if ln.strip()}
    assert want == {n: getattr(L.MovaeFuse, n).offset for n in names}


def test_concurrent_rebuilds_are_serialised(pkg, tmp_path):
    """Every rank of a torchrun launch imports the package and consults build.build(): with a stale csrc/ they must not compile the
    same objects / relink the same .so at once.  Four processes rebuild a scratch tree through a stand-in compiler that logs its
    invocations: each object is compiled once, the library is linked once, nothing partial is left behind."""
    import stat
    import subprocess
    import sys
    import textwrap

    csrc, bdir = tmp_path / "csrc", tmp_path / "_build"
    csrc.mkdir()
    for s in ("a.hip", "b.hip"):
        (csrc / s).write_text("// " + s)
    (tmp_path / "include").mkdir()
    log = tmp_path / "cc.log"
    cc = tmp_path / "fakecc"
    cc.write_text(textwrap.dedent(f"""\
        #!/bin/sh
        out=""
        while [ $# -gt 0 ]; do if [ "$1" = "-o" ]; then out="$2"; fi; shift; done
        printf 'partial' > "$out"; sleep 0.3; printf 'complete-object' > "$out"
        echo "$out" >> {log}
        """))
    cc.chmod(cc.stat().st_mode | stat.S_IEXEC)
    child = textwrap.dedent(f"""\
        import os, sys
        sys.path.insert(0, {str(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))!r})
        import movae_amd  # noqa: F401
        from movae_amd import build as B
        B.CSRC, B.BUILD, B.LIB, B.HERE = {str(csrc)!r}, {str(bdir)!r}, {str(tmp_path / 'lib.so')!r}, {str(tmp_path / 'pkg')!r}
        B.SOURCES = ["a.hip", "b.hip"]
        os.makedirs({str(tmp_path / 'pkg')!r}, exist_ok=True)
        open({str(tmp_path / 'include' / 'movae.h')!r}, "a").close()
        B.build(verbose=False)
        assert open(B.LIB).read() == "complete-object"
        """)
    env = dict(os.environ, HIPCC=str(cc), MOVAE_NO_REBUILD="1")
    procs = [subprocess.Popen([sys.executable, "-c", child], env=env, stderr=subprocess.PIPE) for _ in range(4)]
    errs = [p.communicate()[1].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), errs
    lines = log.read_text().split()
    assert len(lines) == 3, lines  # two objects + one link, not 4 x 3
    left = [f for f in os.listdir(bdir) if ".tmp" in f] + [f for f in os.listdir(tmp_path) if ".tmp" in f]
    assert not left, left
