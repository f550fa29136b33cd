"""Runs the C4-shard property test after poisoning the caching allocator's memory: a kernel that reads a workspace region it (or an
earlier launch) never wrote would then see NaN / huge values instead of the zeros a fresh process hands out."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch
pat = sys.argv[1] if len(sys.argv) > 1 else "nan"
x = torch.empty(6 << 30, dtype=torch.uint8, device="cuda").view(torch.float32)
x.fill_(float("nan") if pat == "nan" else 3.0e38 if pat == "big" else 1.0)
torch.cuda.synchronize(); del x
import test_gpu_train as T
T.test_C4_shard_shape_forward_and_properties()
print("C4 shard test passed on", pat, "poisoned memory")
import test_gpu_e2e as E
for name in dir(E):
    pass
