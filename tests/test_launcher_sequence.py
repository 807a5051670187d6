"""The drop-in boundary as the launcher exercises it (VERDICT r1 missing #4): tests/launcher_worker.py replays
augment_lip_sync.py:187-213 + the loop body of core/function.py:72-107 with npp_amd installed under the reference's module
names."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _run(mode, **env):
    return subprocess.run([sys.executable, os.path.join(HERE, "launcher_worker.py"), mode], capture_output=True, text=True,
                          timeout=900, env=dict(os.environ, **env))


def test_launcher_setup_sequence_cpu():
    r = _run("cpu")
    assert r.returncode == 0 and "LAUNCHER_CPU_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


import bg_children      # noqa: E402  (the two GPU runs are background children, see tests/bg_children.py)

bg_children.register("launcher-gpu", [sys.executable, os.path.join(HERE, "launcher_worker.py"), "gpu"])
bg_children.register("launcher-auto", [sys.executable, os.path.join(HERE, "launcher_worker.py"), "gpu"], dict(NPP_AUTO_GRAPH="1"))


@pytest.mark.gpu
def test_launcher_sequence_with_ddp_and_one_step_gpu():
    r = bg_children.result("launcher-gpu")
    assert r.returncode == 0 and "LAUNCHER_GPU_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.gpu
def test_launcher_sequence_with_auto_graph_gpu():
    """The same unchanged loop with NPP_AUTO_GRAPH=1: Network.forward + backward replayed as hipGraphs from the third call on
    (npp_amd/auto_graph.py), under DistributedDataParallel(find_unused_parameters=True), torch.optim.Adam and MultiStepLR."""
    r = bg_children.result("launcher-auto")
    assert r.returncode == 0 and "LAUNCHER_GPU_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
