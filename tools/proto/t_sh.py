import sys, numpy as np
sys.path.insert(0, '/root/repo/retrieval-scaling_amd'); sys.path.insert(0, '/root/repo')
import rsx
mode = sys.argv[1]
print("gpus", rsx.get_num_gpus())
x = rsx.synth_vectors(64, 4, 1, 2, 0.5, 0, 1000)
if mode == "single":
    ix = rsx.IndexFlatIP(64); ix.add(x); print(ix.search(x[:2], 3)[1])
elif mode == "create_only":
    ix = rsx.IndexFlatIP(64, devices=[0, 0])
elif mode == "add":
    ix = rsx.IndexFlatIP(64, devices=[0, 0]); ix.add(x)
elif mode == "search":
    ix = rsx.IndexFlatIP(64, devices=[0, 0]); ix.add(x); print(ix.search(x[:2], 3)[1])
import torch
try:
    t = torch.zeros(4).cuda(); print(mode, "torch ok", t.device)
except Exception as e:
    print(mode, "torch FAILED:", e)
