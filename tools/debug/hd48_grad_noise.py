"""Is the 2.3e-3 difference on pilot_upsampler.weight (d = 192, 4 heads of 48) fp32 noise?  HIP and PyTorch-ROCm fp32 against the same module in float64."""
import copy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import adafortitran_amd as A
from adafortitran_amd import synth, training

def rel(a, b): return float((a.double() - b.double()).abs().max() / b.double().abs().max())

for d, hd in ((192, 48), (192, 32), (96, 24), (128, 32)):
    sc = A.SystemConfig(ofdm=dict(num_scs=120, num_symbols=14), pilot=dict(num_scs=12, num_symbols=2))
    kw = dict(model_type="adafortitran", patch_size=(3, 2), num_layers=2, model_dim=d, num_head=d // hd, activation="gelu", max_seq_len=512,
              pos_encoding_type="learnable", device="cuda", dropout=0.0, channel_adaptivity_hidden_sizes=[5, 9, 560], adaptive_token_length=6)
    torch.manual_seed(0)
    model = A.AdaFortiTranEstimator(sc, A.ModelConfig(**kw)).train()
    inp = synth.make_inputs(2, ofdm=(120, 14), pilot=(12, 2), seed=9)
    pil, tgt = torch.from_numpy(inp["pilots"]).cuda(), torch.from_numpy(inp["target"]).cuda()
    meta = synth.meta_tuple(inp)

    def step(m, hip, p, t, mt):
        m.transformer_encoder.hip_training = hip
        m.initial_enhancer.hip_training = m.final_refiner.hip_training = hip
        training.HipLinear.default_hip_training = hip
        m.channel_adapter.hip_training = hip
        m.zero_grad()
        out = m(p, mt)
        loss = torch.nn.functional.mse_loss(torch.view_as_real(out), torch.view_as_real(t))
        loss.backward()
        return {n: q.grad.clone() for n, q in m.named_parameters()}

    g32 = step(model, False, pil, tgt, meta)
    ghip = step(model, True, pil, tgt, meta)
    training.HipLinear.default_hip_training = False
    m64 = A.AdaFortiTranEstimator(sc, A.ModelConfig(**kw)).train()
    m64.load_state_dict(model.state_dict())
    m64 = m64.double()
    meta64 = tuple(x.double() if torch.is_tensor(x) and x.is_floating_point() else x for x in meta)
    g64 = step(m64, False, pil.to(torch.complex128), tgt.to(torch.complex128), meta64)
    training.HipLinear.default_hip_training = True
    worst = sorted(((rel(ghip[n], g64[n]), rel(g32[n], g64[n]), rel(ghip[n], g32[n]), n) for n in g64 if not n.startswith("channel_adapter")), reverse=True)[:4]
    print(f"d={d} hd={hd}")
    for w in worst:
        print("   hip-vs-f64 %.2e  torch32-vs-f64 %.2e  hip-vs-torch32 %.2e  %s" % w)
