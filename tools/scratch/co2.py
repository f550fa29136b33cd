import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch, taco_amd
hp = taco_amd.hparams.copy(max_iters=128)
m = taco_amd.create_model(hp); m.load_weights(taco_amd.weights.random_weights(hp, 1, seed=1)); m.initialize(None, None, 1, None)
main_pool = m.plan_pool(32, 128, 128, lanes=1)
main_pool.launch(0); torch.cuda.synchronize()
import sys as _s
mode = _s.argv[1] if len(_s.argv) > 1 else "none"
if mode == "stages":
    ids0 = np.random.RandomState(1).randint(2, 80, size=(32, 128)).astype(np.int32); ids0[:, -1] = 1
    enc = m.encoder(ids0, taco_amd.input_lengths_from_tokens(ids0)); mel0 = m.decoder(enc, 128)[0]; m.postnet(mel0); torch.cuda.synchronize()
if mode in ("bf3", "both"):
    m._lib.taco_debug_set_bf3(m._handle, 0, 0); q = m.plan_pool(32, 128, 128, lanes=1); q.launch(0); torch.cuda.synchronize(); q.close(); m._lib.taco_debug_set_bf3(m._handle, 1, 0)
if mode in ("engine", "both"):
    m.set_decoder_engine(0); q = m.plan_pool(32, 128, 128, lanes=4); q.launch(0); torch.cuda.synchronize(); q.close(); m.set_decoder_engine(1)
if mode == "poison":
    x = torch.full((1 << 30,), 0x7F, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize(); del x
p = m.plan_pool(32, 128, 128, lanes=1, coalesce=2)
ids = np.random.RandomState(0).randint(2, 80, size=(64, 128)).astype(np.int32); ids[:, -1] = 1
p.plans[0].inputs.copy_(torch.from_numpy(ids)); p.plans[0].lengths.copy_(torch.from_numpy(taco_amd.input_lengths_from_tokens(ids)))
torch.cuda.synchronize()
print(mode, m.decoder_engine_info())
for rep in range(2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st = p.streams[0]
    main = torch.cuda.current_stream()
    e0.record(main); st.wait_event(e0)
    for i in range(8): p.launch(0)
    main.wait_stream(st); e1.record(main); torch.cuda.synchronize()
    print("rep", rep, e0.elapsed_time(e1) / 8, "ms per pass")
