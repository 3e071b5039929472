"""One training step against gradients produced by the reference itself (fixtures G_grad_*, generated
by tests/golden/make_golden.py::run_grad from /root/reference: model.train(), dropout 0, the trainer's
MSELoss(cat(Re,Im)), loss.backward()).  Per parameter the fixture keeps the L2 norm, max|g| and every
7th element.

CPU: the package's differentiable composite must reproduce them (pins the training semantics of the
module surface).  GPU: the same step with the encoder on the HIP training kernels."""
import numpy as np
import pytest
import torch

import adafortitran_amd as A
from adafortitran_amd import synth
from helpers import Golden

STRIDE, MAXN = 7, 4096
# ---- Tolerances (relative to each tensor's max|g|, and to its L2 norm) -----------------------------------------------
# Every gradient fixture now has a FLOAT64 twin (G_grad64_*: the reference run in double) that also records, per tensor,
# the CONDITIONING of the gradient at fp32 resolution: `gcond` = max-relative change of the float64 gradient when every
# parameter moves by a random half fp32 ulp (x (1 + 6e-8 u)).  No fp32 evaluation can promise more than that for a tensor,
# whatever its summation order -- its intermediate roundings ARE perturbations of that size.  Measured (make_golden.py):
#   2-layer sets            gcond <= 2e-7 (FortiTran), <= 2e-5 (AdaFortiTran: raw Doppler / delay-spread scalars)
#   full depth, B = 128     2e-5 for the encoder's and the final refiner's weights; 1e-4 .. 5e-4 for initial_enhancer;
#                           1.9e-3 position_embeddings, 2.1e-3 pilot_upsampler.bias, 4.1e-3 pilot_upsampler.weight
#                           (|g|max 5e-7: the far end of six layers of backward)
# and what fp32 implementations actually deliver on the full-depth FortiTran step against float64 (tools/debug/
# grad_vs_fp64.py, grad_flow_fp64.py, MI355X box): pilot_upsampler.weight 1.9e-3 (this library), 2.9e-3 (PyTorch-ROCm
# autograd), 4.1e-3 (torch CPU on the box's EPYC 9575F); encoder weights 1.5e-5 / 2.7e-5 / 1.1e-5.  (The build container's
# Xeon happens to land at 3e-7 on this one step -- that is what the fp32 fixture holds -- which is why round 3's "noise"
# looked like a regression of the HIP path; it is the conditioning of the quantity.)
# So the HIP path is held, PER TENSOR, to  base + 2 x gcond  against the float64 gradient: 1.4e-4 for 84 of the 95 tensors
# (round 2 allowed 3e-3 everywhere, round 3 6e-3), and as loose as the conditioning demands only where it demands it.
# The fp32 fixtures (the reference's own fp32 step) stay the pin of the CPU composite, at round 2's tolerances.
TOL = {"G_grad_forti": (2e-5, 5e-4), "G_grad_ada": (2e-3, 2e-3),
       # round 5: head dims 16 (`num_head: 8`) and 64 (`num_head: 2`) and a 28-token grid, the shapes the training kernels gained
       "G_grad_forti_h16": (2e-5, 5e-4), "G_grad_forti_s28": (2e-5, 5e-4), "G_grad_forti_h64": (2e-5, 5e-4),
       # late round 5: 4 heads of 24 at model_dim 96 -- heads that straddle the kernels' 32-feature blocks
       "G_grad_forti_h24": (2e-5, 5e-4),
       # round 6: the general engine's shapes -- 2 heads of 128 at model_dim 256, model_dim 512 (8 heads of 64)
       "G_grad_forti_h128": (2e-5, 5e-4), "G_grad_forti_d512": (2e-5, 5e-4),
       # full depth at the benchmark's batch (6 layers, B = 128; inputs regenerated bit-exactly from the fixture's seed)
       "G_grad_forti_full": (2e-4, None), "G_grad_ada_full": (4e-3, None)}
# HIP vs float64 at full depth: (element base, norm base); + COND_FACTOR x the tensor's measured conditioning.
# base = fp32 summation noise of sums over 71 680 token rows in another order than ATen's (2e-5 observed), x 5.
# AdaFortiTran's adapter feeds raw conditions (Doppler 1400, delay spread 350) through three MLPs: its fp32 noise is not a
# parameter perturbation (activations of magnitude 1e3 .. 1e5 meet ones of 1e-1): the reference's OWN fp32 step is 5.1e-3 from its
# float64 step on channel_adapter.dop_encoder.4.weight at full depth (fixture against fixture, the CPU test below), so its base is
# round 2's CPU tolerance for that set, 4e-3 (2-layer set: 1e-3).
BASE64 = {"G_grad_forti_full": (1e-4, 5e-5), "G_grad_ada_full": (4e-3, 2e-3), "G_grad_forti": (5e-6, 5e-6), "G_grad_ada": (1e-3, 5e-4),
          # (attention 8 x sharper than the default initialisation: the reference's OWN fp32 step is 1.0e-5 from its float64 step on
          #  position_embeddings there, fixture against fixture -- base = twice that)
          "G_grad_forti_h16": (2e-5, 2e-5), "G_grad_forti_s28": (2e-5, 2e-5),
          "G_grad_forti_h64": (4e-5, 4e-5),     # (the reference's own fp32 step: 2.1e-5 from float64 on position_embeddings)
          "G_grad_forti_h24": (4e-5, 4e-5),     # (... 1.6e-5 on the last conv bias, 1.3e-5 on position_embeddings)
          "G_grad_forti_h128": (4e-5, 4e-5), "G_grad_forti_d512": (4e-5, 4e-5)}
COND_FACTOR = 2.0


def _step(name, device):
    g = Golden(name)
    s = g.spec
    sc = A.SystemConfig(ofdm=dict(num_scs=s["ofdm"][0], num_symbols=s["ofdm"][1]),
                        pilot=dict(num_scs=s["pilot"][0], num_symbols=s["pilot"][1]))
    kw = dict(model_type="adafortitran" if g.adaptive else "fortitran", patch_size=tuple(s["patch"]),
              num_layers=s["num_layers"], model_dim=s["model_dim"], num_head=s["num_head"],
              activation=s.get("activation", "gelu"), max_seq_len=512, pos_encoding_type="learnable",
              device=device, dropout=s["dropout"])
    if g.adaptive:
        kw.update(channel_adaptivity_hidden_sizes=list(s["adaptive_hidden"]), adaptive_token_length=6)
    model = (A.AdaFortiTranEstimator if g.adaptive else A.FortiTranEstimator)(sc, A.ModelConfig(**kw))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in g.state_dict().items()}, strict=True)
    model.train()
    if "pilots" in g:
        inp = {k: g[k] for k in ("pilots", "target")}
        if g.adaptive:
            inp.update({k: g[k] for k in ("snr", "ds", "dop")})
    else:   # full-size sets keep no inputs: synth regenerates them bit-exactly (make_golden.py::run_grad)
        inp = synth.make_inputs(g.meta["batch"], ofdm=tuple(s["ofdm"]), pilot=tuple(s["pilot"]), seed=s["seed"] + 1)
    pil, tgt = torch.from_numpy(inp["pilots"]), torch.from_numpy(inp["target"]).to(device)
    meta = synth.meta_tuple(inp) if g.adaptive else None
    out = model(pil, meta) if meta is not None else model(pil)
    cat = lambda z: torch.cat((torch.real(z), torch.imag(z)), dim=1)  # noqa: E731
    loss = torch.nn.MSELoss()(cat(out), cat(tgt))      # reference trainer._compute_loss
    loss.backward()
    return g, model, float(loss.detach())


def _check(g, model, loss, tol, norm_tol=None):
    norm_tol = tol if norm_tol is None else norm_tol
    assert abs(loss - float(g["loss"])) <= 2e-6 * abs(float(g["loss"])) + 1e-9
    names = [str(n) for n in g["names"]]
    params = dict(model.named_parameters())
    assert sorted(names) == sorted(params)              # same parameter set as the reference module
    for n in names:
        got = params[n].grad.detach().reshape(-1).cpu().numpy()
        gmax = float(g[f"gmax__{n}"])
        assert np.abs(got[::STRIDE][:MAXN] - g[f"gsample__{n}"]).max() <= tol * gmax + 1e-12, n
        norm = float(np.sqrt((got.astype(np.float64) ** 2).sum()))
        assert abs(norm - float(g[f"gnorm__{n}"])) <= norm_tol * float(g[f"gnorm__{n}"]) + 1e-12, n


def _errors64(g64, grads):
    """Per tensor: (max|g - g64| / |g64|max on the sample, | ||g|| - ||g64|| | / ||g64||)."""
    out = {}
    for n in [str(x) for x in g64["names"]]:
        got = grads[n]
        e = float(np.abs(got[::STRIDE][:MAXN].astype(np.float64) - g64[f"gsample__{n}"]).max() / float(g64[f"gmax__{n}"]))
        norm = float(np.sqrt((got.astype(np.float64) ** 2).sum()))
        out[n] = (e, abs(norm - float(g64[f"gnorm__{n}"])) / float(g64[f"gnorm__{n}"]))
    return out


def _check64(name, model):
    """HIP (or any fp32) gradients against the reference's FLOAT64 gradients, per tensor: base + COND_FACTOR x the
    tensor's measured fp32 conditioning (see the tolerance block above)."""
    g64 = Golden(name.replace("G_grad_", "G_grad64_"))
    base_e, base_n = BASE64[name]
    grads = {n: p.grad.detach().reshape(-1).cpu().numpy() for n, p in model.named_parameters()}
    errs = _errors64(g64, grads)
    bad = []
    for n, (e, en) in errs.items():
        tol_e = base_e + COND_FACTOR * float(g64[f"gcond__{n}"])
        tol_n = base_n + COND_FACTOR * float(g64[f"gcondnorm__{n}"])
        if e > tol_e or en > tol_n:
            bad.append(f"{n}: elem {e:.2e} (tol {tol_e:.2e}) norm {en:.2e} (tol {tol_n:.2e})")
    assert not bad, "\n".join(bad)
    return errs


def test_full_depth_composite_matches_reference_gradients_cpu():
    """6 layers at the benchmark's batch of 128 on the CPU composite (one forward + backward, ~10 s)."""
    g, model, loss = _step("G_grad_forti_full", "cpu")
    _check(g, model, loss, TOL["G_grad_forti_full"][0])


def test_fp32_fixtures_sit_inside_the_float64_conditioning_cpu():
    """The reference's own fp32 gradients (G_grad_*) against its float64 gradients (G_grad64_*), fixture against fixture:
    pins the yardstick (same step, same parameter set) and shows that torch's fp32 obeys the same per-tensor bound the HIP
    path is held to."""
    for name in ("G_grad_forti", "G_grad_ada", "G_grad_forti_full", "G_grad_ada_full", "G_grad_forti_h16", "G_grad_forti_s28", "G_grad_forti_h64", "G_grad_forti_h24"):
        g32, g64 = Golden(name), Golden(name.replace("G_grad_", "G_grad64_"))
        assert [str(n) for n in g32["names"]] == [str(n) for n in g64["names"]]
        assert abs(float(g32["loss"]) - float(g64["loss"])) <= 1e-6 * float(g64["loss"])
        base_e, _ = BASE64[name]
        for n in [str(x) for x in g64["names"]]:
            e = np.abs(g32[f"gsample__{n}"].astype(np.float64) - g64[f"gsample__{n}"]).max() / float(g64[f"gmax__{n}"])
            assert e <= base_e + COND_FACTOR * float(g64[f"gcond__{n}"]), (name, n, e)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["G_grad_forti_full", "G_grad_ada_full"])
def test_hip_full_depth_training_step_matches_float64_reference_gradients(name):
    """The whole training step of the default models (6 layers, B = 128 = 71 680 token rows through every training
    kernel at its benchmark size) against the reference's FLOAT64 gradients, every tensor within
    base + 2 x its measured fp32 conditioning -- 1.4e-4 of |g|max for 84 of FortiTran's 95 tensors."""
    g, model, loss = _step(name, "cuda")
    assert abs(loss - float(g["loss"])) <= 2e-6 * abs(float(g["loss"])) + 1e-9
    errs = _check64(name, model)
    tight = sum(1 for n in errs if BASE64[name][0] + COND_FACTOR * float(Golden(name.replace("G_grad_", "G_grad64_"))[f"gcond__{n}"]) <= 2e-4)
    if name == "G_grad_forti_full":
        assert tight >= 80, tight                            # the bound is tight where the problem is well-conditioned


def _fresh_step(spec, adaptive, device, dtype, inp, sd, hip, maps=None):
    """One dropout-free training step of a model built from `spec` (float64 = the yardstick, CPU).  `maps` (a dict) receives the
    gradients leaving the two conv stacks (dL/d input of final_refiner / initial_enhancer), where a differing ReLU decision shows as
    an isolated 5 x 5 patch."""
    from adafortitran_amd import blocks, training
    saved = (blocks.TransformerEncoderForChannels.hip_training, blocks.ConvEnhancer.hip_training,
             blocks.ChannelAdapter.hip_training, training.HipLinear.default_hip_training)
    try:
        blocks.TransformerEncoderForChannels.hip_training = blocks.ConvEnhancer.hip_training = hip
        blocks.ChannelAdapter.hip_training = training.HipLinear.default_hip_training = hip
        sc = A.SystemConfig(ofdm=dict(num_scs=spec["ofdm"][0], num_symbols=spec["ofdm"][1]),
                            pilot=dict(num_scs=spec["pilot"][0], num_symbols=spec["pilot"][1]))
        kw = dict(model_type="adafortitran" if adaptive else "fortitran", patch_size=tuple(spec["patch"]),
                  num_layers=spec["num_layers"], model_dim=spec["model_dim"], num_head=spec["num_head"], max_seq_len=512,
                  device=device, dropout=0.0)
        if adaptive:
            kw.update(channel_adaptivity_hidden_sizes=[7, 42, 560], adaptive_token_length=6)
        model = (A.AdaFortiTranEstimator if adaptive else A.FortiTranEstimator)(sc, A.ModelConfig(**kw))
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        if dtype == torch.float64:
            model.double()
        model.train()
        if maps is not None:
            for name, mod in (("final", model.final_refiner), ("initial", model.initial_enhancer)):
                mod.register_full_backward_hook(lambda _m, gin, _go, name=name: maps.setdefault(name, []).append(gin[0].detach().double().cpu().numpy()))
        cdt = torch.complex128 if dtype == torch.float64 else torch.complex64
        pil, tgt = torch.from_numpy(inp["pilots"]).to(cdt), torch.from_numpy(inp["target"]).to(cdt).to(device)
        meta = None
        if adaptive:
            meta = tuple(t.to(dtype) if torch.is_tensor(t) and t.is_floating_point() else t for t in synth.meta_tuple(inp))
        out = model(pil, meta) if adaptive else model(pil)
        cat = lambda z: torch.cat((torch.real(z), torch.imag(z)), dim=1)  # noqa: E731
        torch.nn.MSELoss()(cat(out), cat(tgt)).backward()
        return {n: p.grad.detach().double().reshape(-1).cpu().numpy() for n, p in model.named_parameters()}
    finally:
        (blocks.TransformerEncoderForChannels.hip_training, blocks.ConvEnhancer.hip_training,
         blocks.ChannelAdapter.hip_training, training.HipLinear.default_hip_training) = saved


@pytest.mark.gpu
@pytest.mark.parametrize("adaptive", [False, True])
def test_hip_gradients_are_as_close_to_float64_as_pytorch_fp32(adaptive):
    """is the HIP gradient as close to the TRUE gradient as torch's fp32 one?  Reduced config (2 layers,
    B = 16, gelu): the gradient in float64 (CPU composite = the reference's arithmetic in double), in fp32 by
    PyTorch-ROCm autograd on the same GPU and by torch on this host's CPU, and in fp32 by the hand-written kernels.
    FortiTran, per tensor:   err(HIP, fp64) <= 2 x err(PyTorch-ROCm fp32, fp64) + floor
    (floor = 1e-6 of |g|max = 8 fp32 ulps: below that both are rounding and a ratio of two roundings means nothing).
    AdaFortiTran feeds raw conditions (Doppler 1400 Hz, delay spread 350 ns) through the adapter: every fp32 evaluation of
    its gradients carries ~1e-4 of CHAOTIC noise -- differentiating an unrelated block through another backend moves single
    adapter tensors by 0.3x .. 3x in either direction (tools/debug/grad_vs_fp64.py ada --small: reference CPU 1.5e-4,
    PyTorch-ROCm 1.7e-4, HIP 2.2e-4 at worst) -- so there the bound is on the worst tensor and on the typical one:
        max_n err_hip <= 2 x max_n err_torch,   median_n (err_hip / err_torch) <= 1.5,
        per tensor err_hip <= 4 x max(err_rocm, err_cpu) + floor.
    ReLU DECISIONS.  The two conv stacks hold 5 M ReLUs per step at this size; a pre-activation within fp32 rounding of zero is
    decided one way by one fp32 evaluation and the other way by the next (about one such element per pair of evaluations: found
    when the streaming conv training kernel landed -- its one differing decision happened to be the one torch's CPU kernels made
    too, tools/debug/conv_model_ab2.py).  One flipped decision is an isolated 5 x 5 patch in the gradient that leaves the stack
    and moves the small upstream tensors (pilot_upsampler: |g|max 1e-6) by 1e-3 of their max -- not an accuracy property of either
    implementation.  The test therefore looks at those gradient maps first: HIP and PyTorch-ROCm must agree on them to 2e-6 of the
    map's max (AdaFortiTran: 5e-4, above its noise) EVERYWHERE (then the per-tensor bound is checked on that input seed), or differ in isolated patches only (< 0.5 % of the
    pixels: a differing ReLU decision -- next seed); a dense difference fails at once.  Two clean seeds are required."""
    from helpers import DEFAULT_SPEC
    spec = dict(DEFAULT_SPEC, num_layers=2)
    sd = synth.make_state_dict(**spec, adaptive_hidden=(7, 42, 560) if adaptive else None, seed=4321)
    clean = 0
    for seed in range(4322, 4334):
        inp = synth.make_inputs(16, seed=seed)
        m_rocm, m_hip = {}, {}
        g_rocm = _fresh_step(spec, adaptive, "cuda", torch.float32, inp, sd, hip=False, maps=m_rocm)
        g_hip = _fresh_step(spec, adaptive, "cuda", torch.float32, inp, sd, hip=True, maps=m_hip)
        flipped = False
        for name in ("final", "initial"):     # backward order: a flip in final_refiner spreads through the encoder into everything upstream
            a, b = m_rocm[name][0], m_hip[name][0]
            dev = np.abs(a - b) > (5e-4 if adaptive else 2e-6) * np.abs(a).max()   # (adaptive: above its ~1e-4 of chaotic noise)
            assert dev.mean() < 5e-3, f"{name}: HIP and PyTorch-ROCm gradient maps differ in {dev.mean():.1%} of the pixels"
            if dev.any():
                flipped = True
                break
        if flipped:
            continue          # a differing ReLU decision somewhere in a conv stack: not an accuracy statement, next seed
        g64 = _fresh_step(spec, adaptive, "cpu", torch.float64, inp, sd, hip=False)
        g_cpu = _fresh_step(spec, adaptive, "cpu", torch.float32, inp, sd, hip=False)
        err = lambda g, n: float(np.abs(g[n] - g64[n]).max() / np.abs(g64[n]).max())  # noqa: E731
        bad, ratios = [], []
        for n in g64:
            e_hip, e_rocm, e_cpu = err(g_hip, n), err(g_rocm, n), err(g_cpu, n)
            ratios.append(e_hip / max(e_rocm, 1e-12))
            limit = 2.0 * e_rocm + 1e-6 if not adaptive else 4.0 * max(e_rocm, e_cpu) + 1e-6
            if e_hip > limit:
                bad.append(f"{n}: hip {e_hip:.2e} rocm {e_rocm:.2e} cpu {e_cpu:.2e}")
        assert not bad, f"seed {seed}\n" + "\n".join(bad)
        worst = lambda g: max(err(g, n) for n in g64)  # noqa: E731
        assert worst(g_hip) <= 2.0 * max(worst(g_rocm), worst(g_cpu)) + 1e-6, (seed, worst(g_hip), worst(g_rocm), worst(g_cpu))
        assert float(np.median(ratios)) <= 1.5, (seed, float(np.median(ratios)))     # typically no further from the truth than torch
        clean += 1
        if clean == 2:
            break
    assert clean == 2, f"only {clean} input seeds without a differing ReLU decision between HIP and PyTorch-ROCm in 12 tries"


@pytest.mark.parametrize("name", ["G_grad_ada", "G_grad_forti", "G_grad_forti_h16", "G_grad_forti_s28", "G_grad_forti_h64", "G_grad_forti_h24",
                                  "G_grad_forti_h128", "G_grad_forti_d512"])
def test_composite_training_step_matches_reference_gradients_cpu(name):
    g, model, loss = _step(name, "cpu")
    _check(g, model, loss, TOL[name][0])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["G_grad_ada", "G_grad_forti", "G_grad_forti_h16", "G_grad_forti_s28", "G_grad_forti_h64", "G_grad_forti_h24",
                                  "G_grad_forti_h128", "G_grad_forti_d512"])
def test_hip_training_step_matches_reference_gradients(name):
    g, model, loss = _step(name, "cuda")
    assert model.transformer_encoder._hip_train_eligible(torch.empty(2, 280, 128, device="cuda"))
    assert all(v is None for v in model.training_backends().values())      # every block on the library's training kernels
    _check(g, model, loss, TOL[name][1])
    _check64(name, model)
