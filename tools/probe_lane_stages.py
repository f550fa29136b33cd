"""Which stage of the forward stops scaling when 4 forwards are in flight?  Per-stage hipGraphs (captured around the
stage-level C ABI calls) replayed on 1 and on 4 PlanPool-selected streams."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, taco_amd
from taco_amd.tacotron import _concurrent_streams
hp = taco_amd.hparams.copy(max_iters=128)
m = taco_amd.create_model(hp); m.initialize(None, None, 1, None)
L = m._lib
B, T_in, n = 32, 128, 128
P = lambda t: C.c_void_p(t.data_ptr() if t is not None else None)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
chk = taco_amd._lib.check
streams = _concurrent_streams(m.device, 4)
def make(stage):
    """returns a list of 4 (graph) objects, one per stream, each with private buffers"""
    gs = []
    for s in streams:
        with torch.cuda.stream(s):
            ws_n = int(L.taco_stage_workspace_bytes(m._handle, B, 512)); ws = torch.empty(ws_n, dtype=torch.uint8, device="cuda")
            ids = torch.randint(2, 80, (B, T_in), dtype=torch.int32, device="cuda"); lens = torch.full((B,), T_in, dtype=torch.int32, device="cuda")
            enc = torch.randn(B, T_in, 256, device="cuda") * 0.3
            mel = torch.randn(B, 512, 80, device="cuda") * 0.3; lin = torch.empty(B, 512, 1025, device="cuda")
            melo = torch.empty(B, 512, 80, device="cuda"); al = torch.empty(B, T_in, n, device="cuda"); stop = torch.zeros(1, dtype=torch.int32, device="cuda")
            x256 = torch.randn(B, 512, 256, device="cuda") * 0.3; o512 = torch.empty(B, 512, 512, device="cuda")
            if stage == "encoder": fn = lambda: chk(L.taco_encoder_forward(m._handle, st(), P(ids), P(lens), P(None), B, T_in, P(enc), P(ws), ws_n))
            elif stage == "decoder": fn = lambda: chk(L.taco_decoder_forward(m._handle, st(), P(enc), P(None), B, T_in, n, P(None), P(None), P(melo), P(al), P(stop), P(None), P(ws), ws_n))
            elif stage == "postnet": fn = lambda: chk(L.taco_postnet_forward(m._handle, st(), P(mel), P(None), B, 512, P(lin), P(None), P(ws), ws_n))
            elif stage == "post_scan": fn = lambda: chk(L.taco_bigru_f32(m._handle, st(), b"post_cbhg", P(x256), P(None), P(None), B, 512, P(o512), P(ws), ws_n))
            elif stage == "post_proj1": fn = lambda: chk(L.taco_conv1d_bn_f32(m._handle, st(), b"post_cbhg/proj_1", P(torch.empty(0)) if False else P(bank), B, 512, 1, 2, P(p1)))
            if stage == "post_proj1":
                bank = torch.randn(B, 512, 2048, device="cuda") * 0.3; p1 = torch.empty(B, 512, 256, device="cuda")
            keep = (ws, ids, lens, enc, mel, lin, melo, al, stop, x256, o512)
            fn(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                fn()
            g.replay(); torch.cuda.synchronize()
            gs.append((g, keep, locals().get("bank"), locals().get("p1")))
    return gs
def run(gs, lanes, reps=6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        for i in range(lanes):
            with torch.cuda.stream(streams[i]): gs[i][0].replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for stage in ("encoder", "decoder", "postnet", "post_scan", "post_proj1"):
    gs = make(stage)
    run(gs, 1); t1 = run(gs, 1); run(gs, 4); t4 = run(gs, 4); t2 = run(gs, 2)
    print("%-11s 1 lane %7.3f ms | 2 lanes %7.3f ms (x%.2f) | 4 lanes %7.3f ms (x%.2f of one lane's time)" % (stage, t1, t2, t2 / t1, t4, t4 / t1), flush=True)
    del gs
