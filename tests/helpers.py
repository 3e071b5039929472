"""Shared test helpers: golden-set loading, weight regeneration, tolerances."""
import json
import os

import numpy as np

from adafortitran_amd import _abi, synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# SURVEY.md 8(d) "Tolerances (stated fp32)"
TOL_ORACLE_OUT = 1e-5      # CPU restatement vs reference fixtures: max|d| <= 1e-5*max(1,|y|max)
TOL_ORACLE_STAGE = 2e-5    # ... and 2e-5 relative (to the tensor's max) on every dumped intermediate
TOL_HIP_OUT = 5e-5         # HIP fp32 vs CPU restatement / fixtures: max|d| <= 5e-5*|y|max
TOL_HIP_MSE = 1e-4         # |dMSE|/MSE


class Golden:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.arrays = {k: z[k] for k in z.files if k != "meta_json"}
        self.meta = json.loads(bytes(z["meta_json"]).decode())
        self.spec = self.meta.get("spec")
        self.name = name

    def __getitem__(self, k):
        return self.arrays[k]

    def __contains__(self, k):
        return k in self.arrays

    @property
    def adaptive(self):
        return bool(self.spec.get("adaptive_hidden"))

    def synth_args(self):
        s = self.spec
        return dict(ofdm=tuple(s["ofdm"]), pilot=tuple(s["pilot"]), patch=tuple(s["patch"]),
                    num_layers=s["num_layers"], model_dim=s["model_dim"], num_head=s["num_head"],
                    max_seq_len=s.get("max_seq_len", 512),
                    adaptive_hidden=tuple(s["adaptive_hidden"]) if s.get("adaptive_hidden") else None,
                    pos_encoding_type=s.get("pos_encoding_type", "learnable"), seed=s["seed"],
                    attn_gain=s.get("attn_gain", 1.0), ffn_gain=s.get("ffn_gain", 1.0),
                    head_gain=s.get("head_gain", 1.0))

    def state_dict(self):
        sd = synth.make_state_dict(**self.synth_args())
        assert synth.state_dict_checksum(sd) == self.meta["weights_crc"], \
            "deterministic weight generator no longer reproduces the fixture's weights"
        if "pe_rows" in self.arrays:  # sinusoid table as the reference host computed it
            pe = sd["transformer_encoder.positional_encoding.pe"].copy()
            pe[0, : self["pe_rows"].shape[0]] = self["pe_rows"]
            sd["transformer_encoder.positional_encoding.pe"] = pe
        return sd

    def abi_config(self):
        s = self.spec
        return _abi.make_config(ofdm=s["ofdm"], pilot=s["pilot"], patch=s["patch"], num_layers=s["num_layers"],
                                model_dim=s["model_dim"], num_head=s["num_head"],
                                activation=s.get("activation", "gelu"),
                                adaptive_hidden=s.get("adaptive_hidden"))

    def meta_arrays(self):
        if not self.adaptive:
            return None, None, None
        return self["snr"], self["ds"], self["dop"]


def max_rel(a, b, floor=0.0):
    """max|a-b| / max(floor, max|b|)"""
    a, b = np.asarray(a), np.asarray(b)
    return float(np.abs(a - b).max() / max(floor, np.abs(b).max()))


DEFAULT_SPEC = dict(ofdm=(120, 14), pilot=(12, 2), patch=(3, 2), num_layers=6, model_dim=128, num_head=4)
