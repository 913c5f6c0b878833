"""ScaledTanh marker module (0.5*tanh(x)+0.5; /root/reference/SOccDPT/model/scaled_tanh.py:4-10).
On the MI355X path the activation is fused into the seg-head tail kernel; this module only
keeps the module tree (and so the state-dict indices seg_head.0/.1/.4) identical."""
import torch.nn as nn


class ScaledTanh(nn.Module):
    def forward(self, x):  # pragma: no cover - never on the HIP path
        raise RuntimeError("ScaledTanh is fused into libsoccdpt_hip's seg-head kernel; call the model, not the module")
