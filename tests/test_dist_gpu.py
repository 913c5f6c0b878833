"""GPU: the RCCL exchange path (all-gather of packed grids + occ_or + expand) under torchrun with one rank
(SOCCDPT_FORCE_DIST=1; more ranks need more GPUs than a test box has): the union of one rank's grid with itself must equal
the single-process result bit for bit.  The N-rank logic is covered on CPU by tests/test_dist_cpu.py (gloo, world_size 2)."""
import hashlib
import os
import subprocess
import sys
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_rccl_exchange_single_rank_matches_plain_forward(gpu_device):
    from soccdpt_amd.model.SOccDPT import SOccDPT_V3
    from soccdpt_amd.utils.synth import synth_input, synth_state_dict, write_synth_calib
    calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
    net = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True)
    net.load_state_dict(synth_state_dict(alias_pretrained=True), strict=False)
    net = net.eval().to(gpu_device)
    inv_up, _, _, occ = net(synth_input(2, seed0=40).to(gpu_device))
    torch.cuda.synchronize()
    h = lambda t: hashlib.sha1(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()
    want = (h(net.last_occ_bits), h(inv_up), str(int((occ[0] > 0).sum())))
    env = dict(os.environ, SOCCDPT_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    worker = os.path.join(os.path.dirname(__file__), "dist_worker_gpu.py")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29547", worker], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1].split()
    assert tuple(line[1:4]) == want
