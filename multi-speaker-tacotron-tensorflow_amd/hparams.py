"""Hyper-parameters of the hot path: the model keys of the reference's hparams.py:33-69 and
max_iters (hparams.py:141), with the effective defaults after its override chain
(hparams.py:83-94: dropout_prob 0.8, post_rnn_size 256, reduction_factor 4).

`HParams` mimics the slice of tf.contrib.training.HParams the reference uses (attribute access,
values(), set_hparam, parse_json via utils/__init__.py:110-126 load_hparams)."""
import copy
import json
import os

SCALE_FACTOR = 1  # hparams.py:3-6


def f(num):
    return num // SCALE_FACTOR


basic_params = {
    # audio keys that shape the model (hparams.py:16-17)
    'num_mels': 80,
    'num_freq': 1025,
    # model (hparams.py:33-69, 83-94)
    'model_type': 'single',  # [single, simple, deepvoice]
    'speaker_embedding_size': f(16),
    'embedding_size': f(256),
    'dropout_prob': 0.8,
    'enc_prenet_sizes': [f(256), f(128)],
    'enc_bank_size': 16,
    'enc_bank_channel_size': f(128),
    'enc_maxpool_width': 2,
    'enc_highway_depth': 4,
    'enc_rnn_size': f(128),
    'enc_proj_sizes': [f(128), f(128)],
    'enc_proj_width': 3,
    'attention_type': 'bah_mon',
    'attention_size': f(256),
    'attention_state_size': f(256),
    'dec_layer_num': 2,
    'dec_rnn_size': f(256),
    'dec_prenet_sizes': [f(256), f(128)],
    'post_bank_size': 8,
    'post_bank_channel_size': f(256),
    'post_maxpool_width': 2,
    'post_highway_depth': 4,
    'post_rnn_size': f(256),
    'post_proj_sizes': [f(256), 80],
    'post_proj_width': 3,
    'reduction_factor': 4,
    # eval (hparams.py:139-141)
    'min_tokens': 50,
    'min_iters': 30,
    'max_iters': 200,
    # training (hparams.py:28,123-133)
    'sample_rate': 24000,
    'adam_beta1': 0.9,
    'adam_beta2': 0.999,
    'initial_learning_rate': 0.002,
    'decay_learning_rate_mode': 0,
    'prioritize_loss': False,
    # audio keys of the spectrogram -> waveform step (hparams.py:16-23,144-145)
    'frame_length_ms': 50,
    'frame_shift_ms': 12.5,
    'preemphasis': 0.97,
    'min_level_db': -100,
    'ref_level_db': 20,
    'griffin_lim_iters': 60,
    'power': 1.5,
}

MODEL_TYPES = {'single': 0, 'simple': 1, 'deepvoice': 2}
ATTENTION_TYPES = {'bah': 0, 'bah_norm': 1, 'bah_mon': 2}
NUM_SYMBOLS = 80  # len(text.symbols.symbols): PAD '_' (0), EOS '~' (1), 78 jamo/punctuation (text/korean.py:11-21)
PAD_ID, EOS_ID = 0, 1


class HParams(object):
    def __init__(self, **kw):
        self._keys = []
        for k, v in kw.items():
            self.add_hparam(k, v)

    def add_hparam(self, name, value):
        if name not in self._keys:
            self._keys.append(name)
        setattr(self, name, copy.deepcopy(value))

    def set_hparam(self, name, value):
        if name not in self._keys:
            raise ValueError('Unknown hyperparameter: %s' % name)
        setattr(self, name, copy.deepcopy(value))

    def values(self):
        return {k: getattr(self, k) for k in self._keys}

    def parse_json(self, text):
        for k, v in (json.loads(text) if isinstance(text, str) else text).items():
            if k in self._keys:
                self.set_hparam(k, v)
        return self

    def copy(self, **overrides):
        hp = HParams(**self.values())
        for k, v in overrides.items():
            hp.add_hparam(k, v)
        return hp


hparams = HParams(**basic_params)


def hparams_debug_string():
    values = hparams.values()
    hp = ['    %s: %s' % (name, values[name]) for name in sorted(values)]
    return 'Hyperparameters:\n' + '\n'.join(hp)


def load_hparams(hp, load_path, skip_list=()):
    """utils/__init__.py:110-126: overwrite hparams from <load_path>/params.json."""
    path = os.path.join(load_path, "params.json")
    with open(path) as fh:
        new = json.load(fh)
    for key, value in new.items():
        if key in skip_list or key not in hp.values():
            continue
        hp.set_hparam(key, value)
    return hp


def save_hparams(model_dir, hp):
    """utils/__init__.py:100-108."""
    os.makedirs(model_dir, exist_ok=True)
    with open(os.path.join(model_dir, "params.json"), "w") as fh:
        json.dump(hp.values(), fh, indent=4, sort_keys=True)
