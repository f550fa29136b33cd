"""Runs one feed-forward layer in a loop (for rocprofv3 --pmc): python tools/prof_one_layer.py <tile> <mpw>"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, taco_amd
tile, mpw = int(sys.argv[1]), int(sys.argv[2])
hp = taco_amd.hparams.copy(max_iters=128)
m = taco_amd.create_model(hp); m.initialize(None, None, 1, None)
L = m._lib
B, T, Cin, Cout = 32, 512, 2048, 256
x = torch.randn(B, T, Cin, device="cuda"); out = torch.empty(B, T, Cout, device="cuda")
if tile > 0: L.taco_debug_set_bf3(m._handle, 1, tile)
else: L.taco_debug_set_bf3(m._handle, 0, 0)
for _ in range(5):
    taco_amd._lib.check(L.taco_conv1d_bn_f32(m._handle, C.c_void_p(torch.cuda.current_stream().cuda_stream), b"post_cbhg/proj_1",
                        C.c_void_p(x.data_ptr()), B, T, 1, mpw, C.c_void_p(out.data_ptr())))
torch.cuda.synchronize()
