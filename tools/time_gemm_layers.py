"""Times the large feed-forward layers of C2 under each k_gemm tile configuration (HIP events)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, taco_amd
hp = taco_amd.hparams.copy(max_iters=128)
m = taco_amd.create_model(hp); m.initialize(None, None, 1, None)
L = m._lib
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
cases = [  # (kind, layer, B, T, Cin, Cout, kw, mpw, act)
    ("conv", "post_cbhg/proj_1", 32, 512, 2048, 256, 3, 2, 1),
    ("conv", "post_cbhg/conv_bank/conv1d_8", 32, 512, 80, 256, 8, 1, 1),
    ("conv", "post_cbhg/proj_2", 32, 512, 256, 80, 3, 1, 0),
    ("dense", "post_cbhg/dense", 32, 512, 80, 256, 1, 1, 0),
    ("conv", "encoder_cbhg/proj_1", 32, 128, 2048, 128, 3, 2, 1),
    ("conv", "encoder_cbhg/proj_2", 32, 128, 128, 128, 3, 1, 0),
    ("hw", "encoder_cbhg/highway_1", 32, 128, 128, 128, 1, 1, 0),
    ("conv", "encoder_cbhg/conv_bank/conv1d_16", 32, 128, 128, 128, 16, 1, 1),
    ("dense", "linear", 32, 512, 512, 1025, 1, 1, 0),
    ("hw", "post_cbhg/highway_1", 32, 512, 256, 256, 1, 1, 0),
]
for kind, layer, B, T, Cin, Cout, kw, mpw, act in cases:
    x = torch.randn(B, T, Cin, device="cuda"); out = torch.empty(B, T, Cout, device="cuda")
    gf = 2.0 * B * T * Cin * Cout * kw * (2 if kind == "hw" else 1) / 1e9
    row = []
    for cfg in (1, 2):
        L.taco_debug_set_bf3(m._handle, 0, 0)
        L.taco_debug_force_gemm_config(m._handle, cfg)
        if kind == "conv":
            fn = lambda: taco_amd._lib.check(L.taco_conv1d_bn_f32(m._handle, st(), layer.encode(), C.c_void_p(x.data_ptr()), B, T, act, mpw, C.c_void_p(out.data_ptr())))
        elif kind == "dense":
            fn = lambda: taco_amd._lib.check(L.taco_dense_f32(m._handle, st(), layer.encode(), C.c_void_p(x.data_ptr()), B * T, act, C.c_void_p(out.data_ptr())))
        else:
            fn = lambda: taco_amd._lib.check(L.taco_highway_f32(m._handle, st(), layer.encode(), C.c_void_p(x.data_ptr()), B * T, C.c_void_p(out.data_ptr())))
        us = timeit(fn)
        row.append("cfg%d %7.1f us %5.1f TF" % (cfg, us, gf / us * 1e-3 * 1e3 / 1e3 * 1e3 / 1e3 if False else gf / (us * 1e-6) / 1e3))
    L.taco_debug_force_gemm_config(m._handle, -1)
    if True:
        for tn in (1, 3, 4, 5, 7, 9, 10):
            L.taco_debug_set_bf3(m._handle, 1, tn)
            us = timeit(fn)
            row.append("bf3t%d %7.1f us %5.1f TFeq" % (tn, us, gf / (us * 1e-6) / 1e3))
        L.taco_debug_set_bf3(m._handle, 1, 0)
    print("%-36s %6.1f GFLOP | " % (layer, gf) + " | ".join(row))
