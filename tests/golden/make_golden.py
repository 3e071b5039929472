#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden.py

What it does: imports the reference's ``src.models`` from /root/reference without
modifying it (two in-process shims: ``typing.Self`` for Python 3.10 and
``sys.dont_write_bytecode`` so nothing is written into the mount), loads the
deterministic synthetic weights of ``adafortitran_amd.synth`` into the reference
modules via ``load_state_dict``, runs the reference forward on seeded synthetic
inputs and stores INPUTS + OUTPUTS + selected intermediates as compressed ``.npz``.
Weights are not stored: tests regenerate them from the recorded generator arguments
and verify the recorded checksum.  No reference source text is stored.
"""
import json
import os
import sys
import typing

sys.dont_write_bytecode = True
import typing_extensions  # noqa: E402

if not hasattr(typing, "Self"):
    typing.Self = typing_extensions.Self  # reference needs py>=3.11

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("AFT_REFERENCE", "/root/reference")
sys.path.insert(0, REF)          # reference's `src` package wins over the repo's shim
sys.path.append(REPO)            # adafortitran_amd (the reference has no such package)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from src.config.schemas import ModelConfig, SystemConfig  # noqa: E402  (reference)
from src.models import AdaFortiTranEstimator, FortiTranEstimator, LinearEstimator  # noqa: E402
import src as _ref_src  # noqa: E402

assert os.path.realpath(_ref_src.__file__).startswith(os.path.realpath(REF)), _ref_src.__file__

from adafortitran_amd import synth  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)


def build_reference(spec):
    sysc = SystemConfig(ofdm=dict(num_scs=spec["ofdm"][0], num_symbols=spec["ofdm"][1]),
                        pilot=dict(num_scs=spec["pilot"][0], num_symbols=spec["pilot"][1]))
    kw = dict(model_type="adafortitran" if spec.get("adaptive_hidden") else "fortitran",
              patch_size=tuple(spec["patch"]), num_layers=spec["num_layers"], model_dim=spec["model_dim"],
              num_head=spec["num_head"], activation=spec.get("activation", "gelu"),
              max_seq_len=spec.get("max_seq_len", 512),
              pos_encoding_type=spec.get("pos_encoding_type", "learnable"), device="cpu")
    if "dropout" in spec:
        kw["dropout"] = spec["dropout"]
    if spec.get("adaptive_hidden"):
        kw.update(channel_adaptivity_hidden_sizes=list(spec["adaptive_hidden"]), adaptive_token_length=6)
    mc = ModelConfig(**kw)
    cls = AdaFortiTranEstimator if spec.get("adaptive_hidden") else FortiTranEstimator
    model = cls(sysc, mc)
    sd = synth.make_state_dict(**synth_args(spec))
    missing = model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    model.eval()
    return model, sd


def synth_args(spec):
    return dict(ofdm=tuple(spec["ofdm"]), pilot=tuple(spec["pilot"]), patch=tuple(spec["patch"]),
                num_layers=spec["num_layers"], model_dim=spec["model_dim"], num_head=spec["num_head"],
                max_seq_len=spec.get("max_seq_len", 512),
                adaptive_hidden=tuple(spec["adaptive_hidden"]) if spec.get("adaptive_hidden") else None,
                pos_encoding_type=spec.get("pos_encoding_type", "learnable"), seed=spec["seed"],
                attn_gain=spec.get("attn_gain", 1.0), ffn_gain=spec.get("ffn_gain", 1.0),
                head_gain=spec.get("head_gain", 1.0))


def planes(pair):
    """[real-pass tensor, imag-pass tensor] each [B,...] -> [2B,...] with plane n = 2*b + part."""
    re, im = (t.detach().cpu().numpy() for t in pair)
    return np.stack([re, im], axis=1).reshape((-1,) + re.shape[1:]).astype(np.float32)


def run_set(name, spec, batch, keep):
    model, sd = build_reference(spec)
    inp = synth.make_inputs(batch, ofdm=tuple(spec["ofdm"]), pilot=tuple(spec["pilot"]), seed=spec["seed"] + 1)
    caps = {}

    def grab(key):
        def hook(_m, _i, o):
            caps.setdefault(key, []).append(o)
        return hook

    hooks = [model.pilot_upsampler.register_forward_hook(grab("upsampled")),
             model.initial_enhancer.register_forward_hook(grab("conv_enhanced")),
             model.transformer_encoder.linear_1.register_forward_hook(grab("lin1")),
             model.transformer_encoder.positional_encoding.register_forward_hook(grab("x0")),
             model.transformer_encoder.register_forward_hook(grab("enc_out")),
             model.final_refiner.register_forward_hook(grab("final"))]
    hooks.append(model.transformer_encoder.register_forward_pre_hook(
        lambda _m, i: caps.setdefault("embed_in", []).append(i[0])))
    hooks.append(model.final_refiner.register_forward_pre_hook(
        lambda _m, i: caps.setdefault("residual", []).append(i[0])))
    if spec.get("adaptive_hidden"):
        hooks.append(model.channel_adapter.register_forward_hook(grab("tokens6")))
    for li, layer in enumerate(model.transformer_encoder.transformer.layers):
        hooks.append(layer.register_forward_hook(grab(f"layer{li}")))

    pil = torch.from_numpy(inp["pilots"])
    meta = synth.meta_tuple(inp) if spec.get("adaptive_hidden") else None
    with torch.no_grad():
        out = model(pil, meta) if meta is not None else model(pil)
    for h in hooks:
        h.remove()

    S, T = spec["ofdm"]
    arrays = {"pilots": inp["pilots"], "target": inp["target"], "out": out.numpy().astype(np.complex64)}
    if spec.get("adaptive_hidden"):
        arrays.update(snr=inp["snr"], ds=inp["ds"], dop=inp["dop"])
    avail = {
        "upsampled": lambda: planes(caps["upsampled"]).reshape(-1, S, T),
        "conv_enhanced": lambda: planes(caps["conv_enhanced"]).reshape(-1, S, T),
        "embed_in": lambda: planes(caps["embed_in"]),
        "x0": lambda: planes(caps["x0"]),
        "enc_out": lambda: planes(caps["enc_out"]),
        "residual": lambda: planes(caps["residual"]).reshape(-1, S, T),
        "tokens6": lambda: caps["tokens6"][0].numpy().astype(np.float32),
        "layer_out": lambda: np.stack([planes(caps[f"layer{li}"]) for li in range(spec["num_layers"])]),
        # size-bounded captures for the default-size sets: frame 0 only / plane 0, first+last layer
        "x0_f0": lambda: planes(caps["x0"])[:2],
        "layer_first_last_p0": lambda: np.stack([planes(caps[f"layer{li}"])[0]
                                                 for li in (0, spec["num_layers"] - 1)]),
    }
    for k in keep:
        arrays[k] = avail[k]()
    pe_key = "transformer_encoder.positional_encoding.pe"
    if pe_key in sd:  # libm-dependent table: ship the rows the forward actually reads
        ntok = (S // spec["patch"][0]) * (T // spec["patch"][1])
        arrays["pe_rows"] = sd[pe_key][0, :ntok].copy()
    # metric fixture on the same (estimate, target) pair: 2*MSELoss(cat(Re,Im;dim=1)) and dB
    # (reference src/utils.py:164-180,233-245; src/main/trainer.py:338-347)
    tgt = torch.from_numpy(inp["target"])
    cat = lambda z: torch.cat((torch.real(z), torch.imag(z)), dim=1)  # noqa: E731
    loss = torch.nn.MSELoss()(cat(out), cat(tgt)).item()
    meta_json = dict(spec=spec, batch=batch, torch=torch.__version__, weights_crc=synth.state_dict_checksum(sd),
                     n_params=int(sum(v.size for k, v in sd.items() if not k.endswith(".pe"))),
                     ref_params=int(sum(p.numel() for p in model.parameters())),
                     metric_2xmse=2.0 * loss, metric_db=float(10 * np.log10(2.0 * loss)))
    arrays["meta_json"] = np.frombuffer(json.dumps(meta_json).encode(), dtype=np.uint8)
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: B={batch} |out|max={np.abs(arrays['out']).max():.4f} "
          f"2xMSE={2 * loss:.6f} params={meta_json['ref_params']} -> {os.path.getsize(path) / 1024:.0f} KiB")


GRAD_SAMPLE_STRIDE, GRAD_SAMPLE_MAX = 7, 4096


def run_grad(name, spec, batch, store_inputs=True):
    """One training step of the reference, as TrainingLoop.train_epoch runs it (reference
    src/main/trainer.py:195-233): model.train(), loss = MSELoss(cat(Re,Im)) of trainer._compute_loss
    (:173-176 via utils.concat_complex_channel), loss.backward().  dropout = 0 so the step is
    deterministic.  Kept per parameter: L2 norm, max |g| and every 7th element (<= 4096 of them)."""
    model, sd = build_reference(spec)
    model.train()
    inp = synth.make_inputs(batch, ofdm=tuple(spec["ofdm"]), pilot=tuple(spec["pilot"]), seed=spec["seed"] + 1)
    pil, tgt = torch.from_numpy(inp["pilots"]), torch.from_numpy(inp["target"])
    meta = synth.meta_tuple(inp) if spec.get("adaptive_hidden") else None
    out = model(pil, meta) if meta is not None else model(pil)
    cat = lambda z: torch.cat((torch.real(z), torch.imag(z)), dim=1)  # noqa: E731
    loss = torch.nn.MSELoss()(cat(out), cat(tgt))
    loss.backward()
    arrays = {"loss": np.float64(loss.item())}
    if store_inputs:
        arrays.update(pilots=inp["pilots"], target=inp["target"], out=out.detach().numpy().astype(np.complex64))
        if spec.get("adaptive_hidden"):
            arrays.update(snr=inp["snr"], ds=inp["ds"], dop=inp["dop"])
    else:   # full-size sets: the inputs are regenerated bit-exactly by synth.make_inputs(batch, seed = spec.seed + 1)
        arrays["out_sample"] = out.detach().numpy().astype(np.complex64).reshape(-1)[::997]
    names = []
    for n, p in model.named_parameters():
        g = p.grad.detach().reshape(-1).numpy()
        names.append(n)
        arrays[f"gnorm__{n}"] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
        arrays[f"gmax__{n}"] = np.float64(np.abs(g).max())
        arrays[f"gsample__{n}"] = g[::GRAD_SAMPLE_STRIDE][:GRAD_SAMPLE_MAX].astype(np.float32)
    arrays["names"] = np.asarray(names)
    meta_json = dict(spec=spec, batch=batch, torch=torch.__version__, weights_crc=synth.state_dict_checksum(sd))
    arrays["meta_json"] = np.frombuffer(json.dumps(meta_json).encode(), dtype=np.uint8)
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: B={batch} loss={loss.item():.6f} params={len(names)} -> {os.path.getsize(path) / 1024:.0f} KiB")


def _grad64(spec, batch, perturb=0.0):
    model, sd = build_reference(spec)
    model.double().train()
    if perturb:
        rng = np.random.default_rng(2024)
        with torch.no_grad():
            for p in model.parameters():
                p.mul_(1 + perturb * torch.from_numpy(rng.uniform(-1, 1, tuple(p.shape))))
    inp = synth.make_inputs(batch, ofdm=tuple(spec["ofdm"]), pilot=tuple(spec["pilot"]), seed=spec["seed"] + 1)
    pil, tgt = torch.from_numpy(inp["pilots"]).to(torch.complex128), torch.from_numpy(inp["target"]).to(torch.complex128)
    meta = None
    if spec.get("adaptive_hidden"):
        meta = tuple(t.double() if torch.is_tensor(t) and t.is_floating_point() else t for t in synth.meta_tuple(inp))
    out = model(pil, meta) if meta is not None else model(pil)
    assert out.dtype == torch.complex128
    cat = lambda z: torch.cat((torch.real(z), torch.imag(z)), dim=1)  # noqa: E731
    loss = torch.nn.MSELoss()(cat(out), cat(tgt))
    loss.backward()
    return {n: p.grad.detach().reshape(-1).numpy().copy() for n, p in model.named_parameters()}, float(loss), sd


def run_grad64(name, spec, batch):
    """The same training step as run_grad, computed by the reference in FLOAT64 (model.double(), complex128 inputs): the
    "true" gradient that both fp32 computations -- the reference's own (fixture G_grad_*) and the HIP path's -- are
    measured against (tests/test_train_golden.py).  Kept per parameter: the same every-7th-element sample, L2 norm and
    max|g| in float64 -- and `gcond`, the CONDITIONING of that tensor's gradient at fp32 resolution: the max-relative change
    of the float64 gradient when every parameter is multiplied by (1 + 6e-8 u), u uniform in [-1, 1] (half an fp32 ulp).
    A coherent half-ulp perturbation bounds what ANY fp32 evaluation can promise for that tensor (its intermediate
    roundings are perturbations of that size): 2e-5 for the encoder's weights, 2e-3 for pilot_upsampler.weight at full
    depth -- the far end of six layers of backward."""
    grads, loss, sd = _grad64(spec, batch)
    pert, _, _ = _grad64(spec, batch, perturb=6e-8)
    arrays = {"loss": np.float64(loss)}
    names = []
    for n, g in grads.items():
        assert g.dtype == np.float64
        names.append(n)
        arrays[f"gnorm__{n}"] = np.float64(np.sqrt((g ** 2).sum()))
        arrays[f"gmax__{n}"] = np.float64(np.abs(g).max())
        arrays[f"gsample__{n}"] = g[::GRAD_SAMPLE_STRIDE][:GRAD_SAMPLE_MAX].copy()
        arrays[f"gcond__{n}"] = np.float64(np.abs(pert[n] - g).max() / np.abs(g).max())
        arrays[f"gcondnorm__{n}"] = np.float64(abs(np.sqrt((pert[n] ** 2).sum()) - arrays[f"gnorm__{n}"]) / arrays[f"gnorm__{n}"])
    arrays["names"] = np.asarray(names)
    meta_json = dict(spec=spec, batch=batch, torch=torch.__version__, weights_crc=synth.state_dict_checksum(sd), dtype="float64",
                     cond_perturbation=6e-8)
    arrays["meta_json"] = np.frombuffer(json.dumps(meta_json).encode(), dtype=np.uint8)
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **arrays)
    worst = max(names, key=lambda n: arrays[f"gcond__{n}"])
    print(f"{name}: B={batch} fp64 loss={loss:.9f} params={len(names)} worst conditioning {arrays['gcond__' + worst]:.1e} ({worst}) "
          f"-> {os.path.getsize(path) / 1024:.0f} KiB")


def run_linear(name, batch, seed):
    """Config #1: the reference LinearEstimator driven plane-wise (it raises on complex input,
    SURVEY.md B5): its real nn.Linear applied to .real and .imag, recombined."""
    sysc = SystemConfig(ofdm=dict(num_scs=120, num_symbols=14), pilot=dict(num_scs=12, num_symbols=2))
    mc = ModelConfig(model_type="linear", patch_size=(3, 2), num_layers=1, model_dim=8, num_head=1, device="cpu")
    model = LinearEstimator(sysc, mc).eval()
    w = synth.uniform_pm(seed, "linear.weight", (1680, 24), 1 / np.sqrt(24))
    b = synth.uniform_pm(seed, "linear.bias", (1680,), 1 / np.sqrt(24))
    model.load_state_dict({"linear.weight": torch.from_numpy(w), "linear.bias": torch.from_numpy(b)})
    inp = synth.make_inputs(batch, seed=seed + 1)
    pil = torch.from_numpy(inp["pilots"])
    with torch.no_grad():
        out = torch.complex(model(pil.real.contiguous()), model(pil.imag.contiguous()))
        try:
            model(pil)
            complex_raises = ""
        except RuntimeError as exc:
            complex_raises = str(exc)[:120]
    meta_json = dict(seed=seed, batch=batch, torch=torch.__version__, complex_input_error=complex_raises)
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), pilots=inp["pilots"], out=out.numpy(),
                        meta_json=np.frombuffer(json.dumps(meta_json).encode(), dtype=np.uint8))
    print(f"{name}: B={batch} complex input raises: {bool(complex_raises)}")


def run_ingest(name, seed):
    """f2/f4 fixtures: synthetic .mat trees run through the REFERENCE's MatDataset / extract_values /
    get_ls_mse_per_folder (prettytable is stubbed in-process: it is only needed by an unrelated
    pretty-printer of src/utils.py)."""
    import shutil
    import tempfile
    import types
    import scipy.io as sio
    sys.modules.setdefault("prettytable", types.SimpleNamespace(PrettyTable=object))
    from src.data.dataset import MatDataset          # reference
    from src.utils import extract_values, get_ls_mse_per_folder
    from src.config.schemas import PilotParams
    rng = np.random.default_rng(seed)
    root = tempfile.mkdtemp(prefix="aft_ingest_")
    arrays = {}
    try:
        folders = {"SNR_10": 3, "SNR_0": 2}
        names, pilots_ref, ideal_ref, meta_ref = [], [], [], []
        for folder, count in folders.items():
            os.makedirs(os.path.join(root, folder))
            for i in range(count):
                H = np.zeros((120, 14, 3), np.complex128)
                H[:, :, 0] = rng.standard_normal((120, 14)) + 1j * rng.standard_normal((120, 14))
                H[:, :, 2] = H[:, :, 0] + 0.1 * (rng.standard_normal((120, 14)) + 1j * rng.standard_normal((120, 14)))
                sc = np.arange(2, 120, 10)[:12]      # 12 pilot subcarriers, symbols 3 and 10
                for s_ in sc:
                    for t_ in (3, 10):
                        H[s_, t_, 1] = H[s_, t_, 2]
                snr = int(folder.split("_")[1])
                fname = f"{i + 1}_SNR-{snr}_DS-{50 * (i + 1)}_DOP-{200 * (i + 1)}_N-3_TDL-A.mat"
                sio.savemat(os.path.join(root, folder, fname), {"H": H})
                arrays[f"H__{folder}__{fname}"] = H.astype(np.complex64)
        for folder in folders:
            ds = MatDataset(os.path.join(root, folder), PilotParams(num_scs=12, num_symbols=2))
            order = sorted(range(len(ds)), key=lambda k: ds.file_list[k].name)
            for k in order:
                hp, hi, meta = ds[k]
                names.append(f"{folder}/{ds.file_list[k].name}")
                pilots_ref.append(hp.numpy()); ideal_ref.append(hi.numpy())
                meta_ref.append([float(t.item()) for t in meta[:5]])
                assert extract_values(ds.file_list[k].name)[5] == meta[5]
        ls = get_ls_mse_per_folder(root)
        arrays.update(names=np.asarray(names), pilots=np.stack(pilots_ref), ideal=np.stack(ideal_ref),
                      meta=np.asarray(meta_ref, np.float32), ls_keys=np.asarray(list(ls.keys())),
                      ls_vals=np.asarray(list(ls.values()), np.float64))
        arrays["meta_json"] = np.frombuffer(json.dumps(dict(seed=seed, torch=torch.__version__)).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **arrays)
        print(f"{name}: {len(names)} files, LS dB {dict(ls)}")
    finally:
        shutil.rmtree(root)


DEFAULT = dict(ofdm=[120, 14], pilot=[12, 2], patch=[3, 2], num_layers=6, model_dim=128, num_head=4, seed=20251114)
SETS = {
    # tiny: every stage dumped
    "T_tiny_ada": (dict(ofdm=[12, 4], pilot=[4, 2], patch=[3, 2], num_layers=2, model_dim=16, num_head=2,
                        adaptive_hidden=[3, 5, 16], max_seq_len=16, seed=7, attn_gain=3.0, head_gain=2.0), 3,
                   ["upsampled", "conv_enhanced", "tokens6", "embed_in", "x0", "layer_out", "enc_out", "residual"]),
    "T_tiny_forti": (dict(ofdm=[12, 4], pilot=[4, 2], patch=[3, 2], num_layers=2, model_dim=16, num_head=2,
                          max_seq_len=8, seed=8, activation="relu", pos_encoding_type="sinusoidal"), 3,
                     ["upsampled", "conv_enhanced", "embed_in", "x0", "layer_out", "enc_out", "residual"]),
    "D_forti": (dict(DEFAULT), 8, ["conv_enhanced", "enc_out"]),
    "A_ada": (dict(DEFAULT, adaptive_hidden=[7, 42, 560]), 8, ["conv_enhanced", "tokens6", "enc_out"]),
    # softmax regimes (probed against an fp64 run of the reference; all well-conditioned, <=3e-6):
    #  D_forti  : logits ~0, softmax ~uniform 1/280      A_ada: raw meta scalars -> logits ~500, near one-hot
    #  DH       : FortiTran, logits ~+-70, mean max-prob 0.22    AH: AdaFortiTran, logits ~+-30, mean max-prob 0.28
    "DH_forti_hot": (dict(DEFAULT, seed=98, attn_gain=32.0, ffn_gain=2.0, head_gain=4.0), 4,
                     ["conv_enhanced", "x0_f0", "layer_first_last_p0", "enc_out", "residual"]),
    "AH_ada_mid": (dict(DEFAULT, adaptive_hidden=[7, 42, 560], seed=99, attn_gain=0.25, ffn_gain=1.0, head_gain=2.0),
                   4, ["conv_enhanced", "x0_f0", "layer_first_last_p0", "enc_out", "residual"]),
    "AS_ada_sin_relu": (dict(DEFAULT, adaptive_hidden=[7, 42, 560], seed=5, activation="relu",
                             pos_encoding_type="sinusoidal", attn_gain=0.5, head_gain=4.0), 4, ["enc_out"]),
    # head dimensions other than 32 (nn.MultiheadAttention takes any num_head | model_dim, reference encoders.py:44-51) and a grid
    # with fewer than 32 tokens (one masked key tile; a 32-row tile of the row-local chain spans several planes)
    "H16_ada_heads8": (dict(DEFAULT, num_layers=2, num_head=8, adaptive_hidden=[7, 42, 560], seed=161, attn_gain=0.25, head_gain=2.0), 4,
                       ["enc_out"]),
    "H64_forti_heads2": (dict(DEFAULT, num_layers=2, num_head=2, seed=641, attn_gain=24.0, ffn_gain=2.0, head_gain=4.0), 4, ["enc_out"]),
    # heads that do not line up with the kernels' 32-feature blocks (late round 5): 4 heads of 24 at model_dim 96, 4 heads of 48 at 192
    "H24_ada_d96_heads4": (dict(DEFAULT, num_layers=2, model_dim=96, num_head=4, adaptive_hidden=[7, 42, 560], seed=241, attn_gain=0.25,
                                head_gain=2.0), 4, ["enc_out"]),
    "H48_forti_d192_heads4": (dict(DEFAULT, num_layers=2, model_dim=192, num_head=4, seed=481, attn_gain=24.0, ffn_gain=2.0, head_gain=4.0), 3,
                              ["enc_out"]),
    "S28_ada_tokens28": (dict(ofdm=[12, 14], pilot=[4, 2], patch=[3, 2], num_layers=2, model_dim=64, num_head=2,
                              adaptive_hidden=[7, 42, 56], max_seq_len=32, seed=281, attn_gain=0.5, head_gain=2.0), 5,
                         ["conv_enhanced", "enc_out"]),
    # round 6: what the reference builds and the packed engine does not take -- the general engine's shapes.
    # model_dim 512 (8 heads of 64), heads of 128 / 56 features, a model_dim off the multiples of 32 with heads of 25 features,
    # a 24-element patch, and 40 layers (more than one window of the by-value layer table; packed engine)
    "W512_ada_d512_heads8": (dict(DEFAULT, num_layers=2, model_dim=512, num_head=8, adaptive_hidden=[7, 42, 560], seed=5121,
                                  attn_gain=0.25, head_gain=2.0), 3, ["enc_out"]),
    "H128_forti_d256_heads2": (dict(DEFAULT, num_layers=2, model_dim=256, num_head=2, seed=1281, attn_gain=24.0, ffn_gain=2.0,
                                    head_gain=4.0), 3, ["enc_out"]),
    "H56_ada_d224_heads4": (dict(DEFAULT, num_layers=2, model_dim=224, num_head=4, adaptive_hidden=[7, 42, 560], seed=561,
                                 attn_gain=0.25, head_gain=2.0), 3, ["enc_out"]),
    "D200_forti_d200_heads8": (dict(DEFAULT, num_layers=2, model_dim=200, num_head=8, seed=2001, attn_gain=16.0, head_gain=2.0), 3,
                               ["enc_out"]),
    "P24_ada_patch12x2": (dict(ofdm=[96, 14], pilot=[12, 2], patch=[12, 2], num_layers=2, model_dim=128, num_head=4,
                               adaptive_hidden=[7, 42, 112], max_seq_len=64, seed=241224, attn_gain=0.5, head_gain=2.0), 4,
                          ["conv_enhanced", "enc_out"]),
    "L40_forti_layers40": (dict(ofdm=[12, 14], pilot=[4, 2], patch=[3, 2], num_layers=40, model_dim=32, num_head=1,
                                max_seq_len=32, seed=4001, attn_gain=4.0), 3, ["enc_out"]),
    "C5_ada_large": (dict(ofdm=[240, 28], pilot=[24, 4], patch=[3, 2], num_layers=12, model_dim=256, num_head=8,
                          adaptive_hidden=[7, 42, 2240], max_seq_len=1120, seed=55, attn_gain=0.5, head_gain=2.0),
                     1, []),
}

if __name__ == "__main__":
    only = set(sys.argv[1:])
    for nm, (spec, batch, keep) in SETS.items():
        if only and nm not in only:
            continue
        run_set(nm, spec, batch, keep)
    if not only or "L_linear" in only:
        run_linear("L_linear", 32, 31)
    if not only or "I_ingest" in only:
        run_ingest("I_ingest", 77)
    if not only or "G_grad_ada" in only:
        run_grad("G_grad_ada", dict(DEFAULT, num_layers=2, adaptive_hidden=[7, 42, 560], dropout=0.0, seed=777), 3)
    if not only or "G_grad_forti" in only:
        run_grad("G_grad_forti", dict(DEFAULT, num_layers=2, dropout=0.0, activation="relu", seed=778), 2)
    # full depth at the benchmark's batch: 6 layers, B = 128, inputs regenerated from the seed
    if not only or "G_grad_forti_full" in only:
        run_grad("G_grad_forti_full", dict(DEFAULT, dropout=0.0, seed=779), 128, store_inputs=False)
    if not only or "G_grad_ada_full" in only:
        run_grad("G_grad_ada_full", dict(DEFAULT, adaptive_hidden=[7, 42, 560], dropout=0.0, seed=780), 128, store_inputs=False)
    # round 5: the training kernels' new shapes pinned on the reference's own gradients -- head dim 16 (8 heads at
    # model_dim 128, as `num_head: 8` in the YAML) and a 28-token grid (12 x 14, patch 3 x 2: one masked key tile)
    H16 = dict(DEFAULT, num_layers=2, num_head=8, dropout=0.0, seed=781, attn_gain=8.0)
    S28 = dict(DEFAULT, ofdm=(12, 14), pilot=(4, 2), num_layers=2, dropout=0.0, seed=782, attn_gain=8.0)
    H64 = dict(DEFAULT, num_layers=2, num_head=2, dropout=0.0, seed=783, attn_gain=8.0)      # head dim 64 (`num_head: 2`)
    # head dim 24: heads straddle blocks.  (Seed: round 5's 784 has a ReLU pre-activation of the final refiner within fp32 rounding of
    # zero -- the step's gradients move by 4e-3 with the summation order of the conv kernel in FRONT of it; of twenty other seeds none
    # has one (tools/debug/h24_seed_search.py: the 16x16x4 and the 32x32x2 training conv kernels and PyTorch-ROCm autograd all within
    # 5e-5 of float64), 7853 is the cleanest.)
    H24 = dict(DEFAULT, num_layers=2, model_dim=96, num_head=4, dropout=0.0, seed=7853, attn_gain=8.0)
    # round 6: the general engine's training shapes -- heads of 128 features (four 32-feature blocks), model_dim 512
    H128 = dict(DEFAULT, num_layers=2, model_dim=256, num_head=2, dropout=0.0, seed=785, attn_gain=8.0)
    D512 = dict(DEFAULT, num_layers=2, model_dim=512, num_head=8, dropout=0.0, seed=786, attn_gain=8.0)
    if not only or "G_grad_forti_h128" in only:
        run_grad("G_grad_forti_h128", H128, 3)
    if not only or "G_grad_forti_d512" in only:
        run_grad("G_grad_forti_d512", D512, 2)
    if not only or "G_grad_forti_h24" in only:
        run_grad("G_grad_forti_h24", H24, 3)
    if not only or "G_grad_forti_h64" in only:
        run_grad("G_grad_forti_h64", H64, 3)
    if not only or "G_grad_forti_h16" in only:
        run_grad("G_grad_forti_h16", H16, 3)
    if not only or "G_grad_forti_s28" in only:
        run_grad("G_grad_forti_s28", S28, 5)
    # float64 runs of the reference on the same steps: the yardstick for the fp32 gradient tolerances
    GRAD64 = {"G_grad64_forti": ("G_grad_forti", dict(DEFAULT, num_layers=2, dropout=0.0, activation="relu", seed=778), 2),
              "G_grad64_ada": ("G_grad_ada", dict(DEFAULT, num_layers=2, adaptive_hidden=[7, 42, 560], dropout=0.0, seed=777), 3),
              "G_grad64_forti_full": ("G_grad_forti_full", dict(DEFAULT, dropout=0.0, seed=779), 128),
              "G_grad64_ada_full": ("G_grad_ada_full", dict(DEFAULT, adaptive_hidden=[7, 42, 560], dropout=0.0, seed=780), 128),
              "G_grad64_forti_h16": ("G_grad_forti_h16", H16, 3), "G_grad64_forti_s28": ("G_grad_forti_s28", S28, 5),
              "G_grad64_forti_h64": ("G_grad_forti_h64", H64, 3), "G_grad64_forti_h24": ("G_grad_forti_h24", H24, 3),
              "G_grad64_forti_h128": ("G_grad_forti_h128", H128, 3), "G_grad64_forti_d512": ("G_grad_forti_d512", D512, 2)}
    for nm, (_f32, spec, batch) in GRAD64.items():
        if not only or nm in only:
            run_grad64(nm, spec, batch)
    leftovers = [os.path.join(r, f) for r, _d, fs in os.walk(REF) for f in fs if f.endswith(".pyc") and "cpython-310" in f]
    assert not leftovers, f"bytecode leaked into the reference mount: {leftovers}"
