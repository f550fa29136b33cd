"""Feed-forward time of the two CBHG stages (scans skipped) for several start delays of the second K half of k_cbhg_front
(csrc/taco_front.h), and with the fused front off (bank and proj_1 as two launches).  python tools/time_front.py [B T_in T_mel]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, taco_amd
B, T_in, T_mel = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 128, 512)
hp = taco_amd.hparams.copy(max_iters=T_mel // 4)
m = taco_amd.create_model(hp); m.initialize(None, None, 1, None)
L = m._lib
ids = torch.randint(2, 80, (B, T_in), dtype=torch.int32, device="cuda"); ids[:, -1] = 1
lens = torch.full((B,), T_in - 1, dtype=torch.int32, device="cuda")
mel = torch.rand(B, T_mel, hp.num_mels, device="cuda")
L.taco_debug_set_skip_scans(m._handle, 1)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
try:
    for d, pr in (("off", 0), (0, 0), (6000, 0), (0, 1), (0, 2), (0, 3), (3000, 1), (6000, 1), (6000, 3), (12000, 1)):
        if d == "off":
            L.taco_debug_set_bf3(m._handle, 9, 0)
        else:
            L.taco_debug_set_bf3(m._handle, 1, 0); L.taco_debug_set_front(m._handle, int(d), int(pr))
        print("front delay %-6s prio %d  post-net feed-forward %7.1f us   encoder feed-forward %7.1f us" % (d, pr, t(lambda: m.postnet(mel)), t(lambda: m.encoder(ids, lens))))
finally:
    L.taco_debug_set_skip_scans(m._handle, 0)
