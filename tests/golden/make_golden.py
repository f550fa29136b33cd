"""Generates tests/golden/tiny_forward.npz from the oracle (the reference cannot run here: TF1 is
not installable -- see oracle/taco_oracle.py header; fixtures are therefore oracle-made and the
parity claim is 'unpinned').  Run: python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
import taco_oracle as O  # noqa: E402


def fixture_config():
    hp = O.OracleHParams.scaled(8, num_mels=8, num_freq=36, enc_bank_size=5, post_bank_size=4, max_iters=6,
                                reduction_factor=3, model_type="deepvoice")
    ns = 3
    w = O.init_weights(hp, ns, seed=4321)
    ids, L = O.synthetic_inputs(3, 11, seed=77, ragged=True)
    spk = np.array([2, 0, 1], np.int32)
    return hp, w, ids, L, spk, ns


def main():
    hp, w, ids, L, spk, ns = fixture_config()
    out = O.forward(w, hp, ids, L, speaker_id=spk, num_speakers=ns)
    np.savez_compressed(os.path.join(HERE, "tiny_forward.npz"), inputs=ids, input_lengths=L, speaker_id=spk,
                        mel=out["mel"], linear=out["linear"], alignments=out["alignments"],
                        **{"w:" + k: v for k, v in w.items()})
    print("wrote", os.path.join(HERE, "tiny_forward.npz"))


if __name__ == "__main__":
    main()
