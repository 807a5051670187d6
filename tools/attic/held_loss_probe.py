"""Does TrainStep's capture survive a caller that keeps the previous (eager) loss alive across the capturing call?"""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import test_train_step_gpu as T
dev = torch.device("cuda:0")
net, opt, step = T._make(dev, graph=True)
im, lpar, lpose, w = T._batch(2, 64, 3, dev)
loss = None
for it in range(5):
    loss = step(im, lpar, lpose)          # the previous loss (and its autograd graph) stays referenced during this call
    torch.cuda.synchronize()
    print("step", it, float(loss.detach()), "graphed" if step.graphed else "eager", flush=True)
print("HELD_OK")
