"""CPU: the committed PMC counters describe the kernels in the tree.  bench.py fills roofline.traffic from the newest
profiles/r*_pmc_traffic.json and prints null when that file was collected from other forward-path sources; this test makes such a
commit fail instead of shipping a null: whoever edits a FORWARD_SOURCES file re-runs tools/collect_profiles.sh +
tools/install_profiles.py as the last step."""
import glob
import json
import os

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OTHER = ("_base384", "_hybrid384")


def _newest(suffix):
    c = sorted(glob.glob(os.path.join(REPO, "profiles", f"r*{suffix}_pmc_traffic.json")))
    if not suffix:
        c = [f for f in c if not any(t in os.path.basename(f) for t in OTHER)]
    return c[-1] if c else None


@pytest.mark.parametrize("suffix", ["", "_base384", "_hybrid384"])
def test_newest_pmc_traffic_matches_the_tree(suffix):
    from soccdpt_amd.lib import csrc_sha
    f = _newest(suffix)
    assert f is not None, f"no profiles/r*{suffix}_pmc_traffic.json"
    pj = json.load(open(f))
    assert pj.get("csrc_sha") == csrc_sha(), (f"{os.path.basename(f)} was collected from other forward-path kernel sources "
                                              f"({pj.get('csrc_sha')} != {csrc_sha()}): re-run tools/collect_profiles.sh / install_profiles.py")
    assert pj["kernels"], "empty PMC summary"


def test_newest_kernel_stats_is_beside_the_pmc_file():
    """The rocprofv3 --kernel-trace --stats summary the roofline's average launch time must agree with is from the same collection."""
    f = _newest("")
    tag = os.path.basename(f)[: -len("_pmc_traffic.json")]
    for name in ("_kernel_stats.csv", "_bench.json", "_bench_under_rocprof.json"):
        assert os.path.exists(os.path.join(REPO, "profiles", tag + name)), tag + name
