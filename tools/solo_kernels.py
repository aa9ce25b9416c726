import os, sys, json
sys.path.insert(0, "/root/repo")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch, bench
from gpujpeg_amd import libgpujpeg as G
lib = G.Library(sys.argv[1]); assert lib.L.gpujpeg_init_device(0, 0) == 0
dev = torch.device("cuda", 0)
sp = bench.Spec(lib, sys.argv[2] if len(sys.argv) > 2 else "8k", "natural", 75, dev, 12345)
L = bench.Lanes(lib, sp, dev, 1); L.warm(3)
print(sys.argv[1].split("/")[-1], [round(x * 1e3, 1) for x in L.solo_kernel_ms(20)])
