"""The HIP path under N = 2 ranks (the CPU sharding test computes with the oracle, not the product).
Two processes share the one GPU of a test box; what is tested is the data-parallel contract of inference: every rank runs the
product on its own rows, nothing is exchanged on the data path, and the gathered result IS the single-process result."""
import os
import subprocess
import sys

import numpy as np
import pytest

import taco_oracle as O
from util import build_model, maxabs, argmax_match

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model_type,ns", [("single", 1), ("deepvoice", 3)])
def test_two_ranks_on_shards_equal_one_process_on_the_batch(tmp_path, model_type, ns):
    import torch
    B, T_in, n, seed = 7, 21, 6, 811                       # 7 rows: shards of 4 and 3
    ohp = O.OracleHParams(max_iters=n, model_type=model_type)
    w = O.init_weights(ohp, ns, seed)
    ids, L = O.synthetic_inputs(B, T_in, seed + 1, ragged=True)
    spk = (np.arange(B) % ns).astype(np.int32)
    ref = O.forward(w, ohp, ids, L, speaker_id=spk if ns > 1 else None, num_speakers=ns)
    m = build_model(ohp, w, num_speakers=ns)
    lin1, ali1 = m.run(inputs=ids, input_lengths=L, speaker_id=spk if ns > 1 else None)
    torch.cuda.synchronize()
    lin1, ali1, mel1 = lin1.cpu().numpy(), ali1.cpu().numpy(), m.mel_outputs.cpu().numpy()
    m.close()
    np.savez(os.path.join(tmp_path, "case.npz"), ids=ids, L=L, spk=spk, n=n, seed=seed, model_type=model_type, num_speakers=ns)
    port = str(29900 + os.getpid() % 90)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_infer_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", port, str(tmp_path)], env=env) for r in range(2)]
    for pr in procs:
        assert pr.wait(timeout=600) == 0
    res = np.load(os.path.join(tmp_path, "result.npz"))
    assert res["linear"].shape == lin1.shape and float(res["wall"]) == 2.0
    # rows never interact: the sharded run reproduces the single-process run (other rows-per-group / tile positions: rounding only)
    assert maxabs(res["mel"], mel1) < 2e-6 and maxabs(res["linear"], lin1) < 2e-6 and maxabs(res["alignments"], ali1) < 2e-6
    # and both are the reference function
    k = ref["mel"].shape[1]
    assert maxabs(res["mel"][:, :k], ref["mel"]) < 2e-4 and maxabs(res["linear"][:, :k], ref["linear"]) < 2e-4
    nchk, bad = argmax_match(res["alignments"][:, :, :ref["alignments"].shape[2]], ref["alignments"])
    assert bad == 0 and nchk > 0
