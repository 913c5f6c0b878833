"""CPU: the encoder restatements of oracle/soccdpt_ref.py against the independent HF `transformers` ports carrying the same synthetic
weights (oracle/hf_crosscheck.py).  The encoders stay "parity unpinned" at the timm boundary (timm 0.6.12 is not installable and the
reference has no fixture); this keeps the only independent check alive in the committed suite, for both
Swin-V2 models -- base_384's log-CPB normalisation by `pretrained_window_sizes` (12, 12, 12, 6) included -- and for the ViT-hybrid."""
import pytest
import torch

from oracle import soccdpt_ref as R
from soccdpt_amd.utils.synth import synth_input, synth_state_dict

transformers = pytest.importorskip("transformers")


@pytest.mark.parametrize("backbone,img,tol", [("swin2t16_256", 256, 2e-5), ("swin2b24_384", 384, 2e-5)])
def test_swin_encoder_matches_hf_swinv2(backbone, img, tol):
    from oracle.hf_crosscheck import swinv2_hf_features
    torch.set_num_threads(8)
    sd = synth_state_dict(backbone)
    x = synth_input(1, size=img, seed0=5)
    with torch.no_grad():
        mine = R.swin_encoder(sd, x, R.ARCHS[backbone])
    theirs = swinv2_hf_features(sd, x, backbone)
    assert len(mine) == len(theirs) == 4
    for s, (a, b) in enumerate(zip(theirs, mine)):
        assert a.shape == b.shape
        err = float((a - b).abs().max() / b.abs().max())
        assert err < tol, (backbone, s, err)


def test_hybrid_encoder_matches_hf_dpt_hybrid():
    """vitb_rn50_384 (BASELINE configs[2]): ResNetV2 stem / stages (weight-standardised SAME convs, GroupNorm, -inf padded max-pool),
    ViT-B blocks over 577 tokens, 'project' readout and the reassemble convs against HF's DPT-hybrid port, same weights."""
    from oracle.hf_crosscheck import hybrid_hf_features
    torch.set_num_threads(8)
    sd = synth_state_dict("vitb_rn50_384")
    x = synth_input(1, size=384, seed0=6)
    with torch.no_grad():
        mine = R.hybrid_encoder(sd, x)
    theirs = hybrid_hf_features(sd, x)
    assert [tuple(t.shape) for t in mine] == [(1, 256, 96, 96), (1, 512, 48, 48), (1, 768, 24, 24), (1, 768, 12, 12)]
    for s, (a, b) in enumerate(zip(theirs, mine)):
        assert a.shape == b.shape
        err = float((a - b).abs().max() / b.abs().max())
        assert err < 2e-5, (s, err)
