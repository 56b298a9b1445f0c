"""kNN kernel timing on SURVEY.md section 8(d) shapes: distance evaluations per second (torch events on the current stream)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from slide_amd import _ext as hip
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
for B, n2, n1, K in ((256, 256, 128, 32), (256, 1024, 256, 32), (256, 2048, 1024, 32), (256, 8192, 2048, 32), (2048, 2048, 1024, 32),
                     (256, 2048, 2048, 8), (256, 2048, 2048, 16), (256, 2048, 1024, 64), (256, 16, 16, 16), (256, 2048, 16, 12)):
    p1 = torch.rand(B, n1, 3, generator=g).to(dev); p2 = torch.rand(B, n2, 3, generator=g).to(dev)
    hip.knn_points(p1, p2, K, None); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        hip.knn_points(p1, p2, K, None)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print("B%d n2 %d n1 %d K%d: %.1f us  %.2f T dist/s" % (B, n2, n1, K, us, B * n1 * n2 / us / 1e6))
