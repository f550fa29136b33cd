import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, taco_amd
hp = taco_amd.hparams.copy(max_iters=128)
m = taco_amd.create_model(hp); m.load_weights(taco_amd.weights.random_weights(hp, 1, seed=1)); m.initialize(None, None, 1, None)
B, T_in, n = 32, 128, 128
rs = np.random.RandomState(1)
ids = rs.randint(2, 80, size=(B, T_in)).astype(np.int32); ids[:, -1] = 1
L = taco_amd.input_lengths_from_tokens(ids)
for _ in range(3):
    enc = m.encoder(ids, L, None); torch.cuda.synchronize()
mel = m.decoder(enc, n, None)[0]; torch.cuda.synchronize()
for _ in range(3):
    m.postnet(mel); torch.cuda.synchronize()
