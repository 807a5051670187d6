"""Worker of test_train_step_gpu.test_capture_failure_falls_back_to_eager: NPP_TEST_FAIL_CAPTURE makes TrainStep's capture
fail half way (an illegal call inside it); the step must keep training eagerly in the same process."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import test_train_step_gpu as T      # noqa: E402

dev = torch.device("cuda:0")
net, opt, step = T._make(dev, graph=True)
im, lpar, lpose, w = T._batch(2, 64, 3, dev)
losses = [float(step(im, lpar, lpose).detach()) for _ in range(5)]
assert not step.graphed and not step.use_graph, "the forced failure should have switched the step to eager"
assert all(l == l for l in losses) and losses[-1] < losses[0], losses
assert opt.device_step_count() == 5, opt.device_step_count()
print("FALLBACK_OK", losses)
