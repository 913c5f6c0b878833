"""GPU: the XCD-local persistent stage kernel (csrc/stage_xcd.hip) against the launch chain it replaces.

Swin-V2 stages 2-3 of dpt_swin2_tiny_256 -- 58 launches of the chain -- run as ONE launch when the batch is a multiple of 8 (one frame per XCD).  Every
phase body is the device function of the stand-alone kernel, so the contract is bit-identity: hooked feature maps 2 and 3, the residual stream and all
four outputs of the forward equal the chain's exactly, in every arithmetic mode the path supports; other batch sizes keep the chain."""
import os
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(prec, xcd, dev):
    from soccdpt_amd.model.SOccDPT import SOccDPT_V3
    from soccdpt_amd.utils.synth import synth_state_dict, write_synth_calib
    calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
    m = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True, precision=prec)
    m.load_state_dict(synth_state_dict(alias_pretrained=True), strict=False)
    m = m.eval().to(dev)
    if xcd:
        m._engine(dev).set_stage_xcd(True)
    return m


def _same(a, b):
    return torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0))


@pytest.mark.parametrize("precision", ["bf16", "f16", "mixed"])
@pytest.mark.parametrize("B", [8, 16])
def test_persistent_stages_bit_identical_to_the_launch_chain(gpu_device, precision, B):
    from soccdpt_amd.lib import PREC_BF16, PREC_F16, PREC_MIXED
    from soccdpt_amd.utils.synth import synth_input
    prec = {"bf16": PREC_BF16, "f16": PREC_F16, "mixed": PREC_MIXED}[precision]
    x = synth_input(B, seed0=60).to(gpu_device)
    chain, pers = _build(prec, False, gpu_device), _build(prec, True, gpu_device)
    out_c = [t.clone() for t in chain(x)]
    ec = chain._engine(gpu_device)
    feats_c = {k: ec.workspace_tensor(B, k).clone() for k in ("feat2", "feat3", "xf")}
    n_chain = ec.launch_count()
    for rep in range(3):            # repeated launches reuse the self-resetting synchronisation words
        out_p = pers(x)
    ep = pers._engine(gpu_device)
    torch.cuda.synchronize()
    assert ep.stage_xcd_status() == 0
    assert ep.launch_count() == n_chain - 57, (ep.launch_count(), n_chain)      # 58 launches became one
    for k, v in feats_c.items():
        assert _same(ep.workspace_tensor(B, k), v), k
    for u, v in zip(out_p, out_c):
        assert _same(u, v)


def test_other_batch_sizes_keep_the_launch_chain(gpu_device):
    from soccdpt_amd.lib import PREC_F16
    from soccdpt_amd.utils.synth import synth_input
    x = synth_input(3, seed0=61).to(gpu_device)
    chain, pers = _build(PREC_F16, False, gpu_device), _build(PREC_F16, True, gpu_device)
    a, b = chain(x), pers(x)
    torch.cuda.synchronize()
    assert pers._engine(gpu_device).launch_count() == chain._engine(gpu_device).launch_count()
    assert pers._engine(gpu_device).stage_xcd_status() == -1      # the persistent kernel never ran
    for u, v in zip(a, b):
        assert _same(u, v)
