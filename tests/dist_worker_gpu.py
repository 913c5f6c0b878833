"""Worker of tests/test_dist_gpu.py: one torchrun rank on cuda:0 with the RCCL exchange path forced (SOCCDPT_FORCE_DIST=1):
forward of 2 frames, prints the SHA-1 of the union occupancy bits and of inv_up."""
import hashlib
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from soccdpt_amd import dist as sdist  # noqa: E402
from soccdpt_amd.model.SOccDPT import SOccDPT_V3  # noqa: E402
from soccdpt_amd.utils.synth import synth_input, synth_state_dict, write_synth_calib  # noqa: E402

rank, local, world = sdist.init_from_env("nccl")
dev = torch.device("cuda", local)
calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
net = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True)
net.load_state_dict(synth_state_dict(alias_pretrained=True), strict=False)
net = sdist.attach(net.eval().to(dev))
assert net.occ_exchange is not None, "exchange not attached"
x = synth_input(2, seed0=40).to(dev)
inv_up, seg_up, pts, occ = net(x)
occ2 = net(x)[3]                                  # second call: the reused gather buffer must give the same grid
torch.cuda.synchronize()
assert torch.equal(occ, occ2)
h = lambda t: hashlib.sha1(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()
print("RESULT", h(net.last_occ_bits), h(inv_up), int((occ[0] > 0).sum()), flush=True)
torch.distributed.destroy_process_group()
