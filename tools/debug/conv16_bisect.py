"""Round 6: which conv launch of the G_grad_forti_h24 step makes the gradients differ: the 16x16x4 kernel in the forward only, in the
backward only, in both, in neither (switch AFT_CONV_MFMA32 flipped between forward and backward)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np, torch
import adafortitran_amd as A
from adafortitran_amd import _lib, synth
from helpers import Golden
import test_train_golden as T

name = sys.argv[1] if len(sys.argv) > 1 else "G_grad_forti_h24"
G = Golden(name); g64 = Golden(name.replace("G_grad_", "G_grad64_"))
s = G.spec
def run(fwd32, bwd32):
    sc = A.SystemConfig(ofdm=dict(num_scs=s["ofdm"][0], num_symbols=s["ofdm"][1]), pilot=dict(num_scs=s["pilot"][0], num_symbols=s["pilot"][1]))
    kw = dict(model_type="fortitran", patch_size=tuple(s["patch"]), num_layers=s["num_layers"], model_dim=s["model_dim"], num_head=s["num_head"],
              activation=s.get("activation", "gelu"), max_seq_len=512, pos_encoding_type="learnable", device="cuda", dropout=s["dropout"])
    model = A.FortiTranEstimator(sc, A.ModelConfig(**kw))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in G.state_dict().items()}, strict=True)
    model.train()
    pil, tgt = torch.from_numpy(G["pilots"]), torch.from_numpy(G["target"]).cuda()
    _lib.set_switch("AFT_CONV_MFMA32", "1" if fwd32 else None)
    out = model(pil)
    cat = lambda z: torch.cat((torch.real(z), torch.imag(z)), dim=1)
    loss = torch.nn.MSELoss()(cat(out), cat(tgt))
    torch.cuda.synchronize()
    _lib.set_switch("AFT_CONV_MFMA32", "1" if bwd32 else None)
    loss.backward()
    torch.cuda.synchronize()
    _lib.set_switch("AFT_CONV_MFMA32", None)
    errs = []
    for n, p in model.named_parameters():
        got = p.grad.detach().reshape(-1).cpu().numpy()[::T.STRIDE][:T.MAXN]
        errs.append((float(np.abs(got.astype(np.float64) - g64[f"gsample__{n}"]).max() / float(g64[f"gmax__{n}"])), n))
    errs.sort(reverse=True)
    return errs[:3]
for fwd32 in (False, True):
    for bwd32 in (False, True):
        print("fwd", "32" if fwd32 else "16", "bwd", "32" if bwd32 else "16", [(f"{e:.1e}", n[-40:]) for e, n in run(fwd32, bwd32)])
