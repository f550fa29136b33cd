"""Does feed-forward work of another request fill the CUs while a post-net scan runs?  postnet(A) on one stream, encoder(B) on another."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, taco_amd
hp = taco_amd.hparams.copy(max_iters=128)
m = taco_amd.create_model(hp); m.load_weights(taco_amd.weights.random_weights(hp, 1, seed=1)); m.initialize(None, None, 1, None)
B, T_in, n = 32, 128, 128
rs = np.random.RandomState(1)
ids = rs.randint(2, 80, size=(B, T_in)).astype(np.int32); ids[:, -1] = 1
L = taco_amd.input_lengths_from_tokens(ids)
enc = m.encoder(ids, L, None); mel = m.decoder(enc, n, None)[0]
m2 = taco_amd.create_model(hp); m2.load_weights(taco_amd.weights.random_weights(hp, 1, seed=1)); m2.initialize(None, None, 1, None)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(a, b, reps=6):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
    for _ in range(reps):
        if a:
            with torch.cuda.stream(s1): m.postnet(mel)
        if b:
            with torch.cuda.stream(s2): m2.encoder(ids, L, None)
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for _ in range(2):
    run(True, True)
print("postnet alone %.3f ms, encoder alone %.3f ms, both concurrently %.3f ms per pair" % (run(True, False), run(False, True), run(True, True)))
m.check_device_errors(); m2.check_device_errors()
print(m.decoder_engine_info())
