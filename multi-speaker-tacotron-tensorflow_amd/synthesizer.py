"""`Synthesizer` -- the inference driver of the reference (synthesizer.py:23-207), re-hosted.

Kept: load() / synthesize() / close() names and arguments; token -> input_lengths rule
(synthesizer.py:120); default speaker id zeros (:43-44); the (linear_outputs, alignments) fetch pair
(:122-126,166-167); manual-attention second pass for modes 1 and 3 (:171-205; mode 2 is broken in the
reference: np.pow does not exist); text -> ids with the Korean normaliser (text.py, korean.py); the attention trim
(:242-262, device kernel) and, with vocode=True, the spectrogram -> waveform step on the GPU (audio.py).
Out of scope: plots, wav / npy file output, sentence concatenation -- `synthesize` returns the model
outputs as numpy arrays (and keeps `spec_end_idx` / `wavs` as attributes) instead."""
import glob
import os
import re

import numpy as np

from .hparams import hparams, load_hparams, EOS_ID
from .tacotron import create_model
from .weights import load_weights


def get_most_recent_checkpoint(checkpoint_dir, checkpoint_step=None):
    """synthesizer.py:289-299: `model.ckpt-<step>.safetensors` packs written here, or the reference's own TensorFlow
    checkpoints (`model.ckpt-<step>.index` + `.data-*`, read by tf_checkpoint.py without TensorFlow)."""
    if checkpoint_step is None:
        paths = glob.glob(os.path.join(checkpoint_dir, "*.safetensors")) + glob.glob(os.path.join(checkpoint_dir, "model.ckpt-*.index"))
        if not paths:
            raise Exception(" [!] No checkpoint found in {}".format(checkpoint_dir))
        def step_of(p):
            m = re.search(r"ckpt-(\d+)", os.path.basename(p))
            return int(m.group(1)) if m else -1
        return max(paths, key=step_of)
    st = os.path.join(checkpoint_dir, "model.ckpt-{}.safetensors".format(checkpoint_step))
    return st if os.path.exists(st) or not os.path.exists(st[:-len("safetensors")] + "index") else st[:-len("safetensors")] + "index"


def manual_alignments_of(alignments, mode):
    """The alignments the reference feeds to its second pass (synthesizer.py:171-205) from the first pass's `alignments` [N, T_in, T_dec]:
    [N, T_dec, T_in] (the layout of the `manual_alignments` placeholder, rnn_wrappers.py:313-317).
      mode 1 ("argmax one hot", :173-179): zeros, and for every ENCODER position e a one at the decoder step where e was attended most --
             the reference takes `alignments[idx].argmax(1)`, the argmax over decoder steps, and writes `new[(argmax, range(E))] = 1`; a
             decoder step can therefore end up with no one at all, or with several.  Reproduced as it stands.
      mode 3 ("prunning", :189-195): the same ones written into a copy of the transposed alignments instead of into zeros.
      mode 2 calls np.pow, which does not exist (:181-188): the reference raises AttributeError there; so does this (as an Exception).
    Pinned on the reference's own code: tests/golden/manual_vectors.npz (tools/make_reference_vectors.py)."""
    alignments = np.asarray(alignments)
    if mode == 2:
        raise Exception("manual_attention_mode 2 is broken in the reference (np.pow, synthesizer.py:181-188)")
    alignments_T = np.transpose(alignments, [0, 2, 1])                                   # [N, D, E]
    new_alignments = np.zeros_like(alignments_T) if mode == 1 else alignments_T.copy()
    for idx in range(len(alignments)):
        argmax = alignments[idx].argmax(1)                                               # [E]: decoder step of every encoder position
        new_alignments[idx][(argmax, np.arange(len(argmax)))] = 1
    return new_alignments


class Synthesizer(object):
    # text -> ids ending in EOS (text/__init__.py:23-58): jamo tokeniser of text.py; the reference's Korean number / abbreviation
    # normaliser is not part of it -- assign a callable that includes one if the input needs it
    text_to_sequence = None

    def close(self):
        if getattr(self, "model", None) is not None:
            self.model.close()
            self.model = None

    def load(self, checkpoint_path, num_speakers=2, checkpoint_step=None, model_name='tacotron', device=None):
        self.num_speakers = num_speakers
        if os.path.isdir(checkpoint_path):
            load_path = checkpoint_path
            checkpoint_path = get_most_recent_checkpoint(checkpoint_path, checkpoint_step)
        else:
            load_path = os.path.dirname(checkpoint_path)
        self.hparams = hparams.copy()
        load_hparams(self.hparams, load_path)
        self.model = create_model(self.hparams)
        if checkpoint_path.endswith(".index") or os.path.exists(checkpoint_path + ".index"):      # a TensorFlow checkpoint of the reference
            from .tf_checkpoint import import_tf_checkpoint
            prefix = checkpoint_path[:-len(".index")] if checkpoint_path.endswith(".index") else checkpoint_path
            self.model.load_weights(import_tf_checkpoint(prefix, self.hparams, num_speakers))
        else:
            self.model.load_weights(load_weights(checkpoint_path))
        self.model.initialize(None, None, self.num_speakers, None, device=device)   # placeholders (:39-52)
        return self

    def synthesize(self, texts=None, tokens=None, base_path=None, paths=None, speaker_ids=None,
                   start_of_sentence=None, end_of_sentence=True, pre_word_num=0, post_word_num=0,
                   pre_surplus_idx=0, post_surplus_idx=1, use_short_concat=False,
                   manual_attention_mode=0, base_alignment_path=None, librosa_trim=False,
                   attention_trim=True, manual_alignments=None, vocode=False):
        if type(texts) == str:
            texts = [texts]
        if texts is not None and tokens is None:
            if self.text_to_sequence is None:
                # text/__init__.py:23-58 with the korean cleaner: normalise (numbers, units, Latin letters), decompose, append EOS
                from .text import text_to_sequence as _t2s
                from .korean import KoreanNormalizer
                if getattr(self, "normalizer", None) is None:
                    self.normalizer = KoreanNormalizer()
                sequences = [_t2s(text, normalizer=self.normalizer) for text in texts]
            else:
                sequences = [self.text_to_sequence(text) for text in texts]
            if len(set(len(x) for x in sequences)) > 1:            # the reference needs equal lengths here (App. C); pad like the feeder
                from .text import pad_token_rows
                sequences = pad_token_rows(sequences)
        elif tokens is not None:
            sequences = tokens
        else:
            raise Exception("either texts or tokens is required")
        sequences = np.asarray(sequences)
        if sequences.ndim != 2:
            raise Exception("token rows must have equal length (pre-pad with 0 as eval.py / train.py:27-40 do)")
        input_lengths = np.argmax(sequences == EOS_ID, 1).astype(np.int32)             # synthesizer.py:120
        if type(speaker_ids) == dict:
            raise Exception("dict-valued speaker_ids is broken in the reference (synthesizer.py:153-164) and not supported")
        if manual_alignments is None and base_alignment_path is not None:               # :134-150
            alignment_path = os.path.join(base_alignment_path, os.path.basename(base_path))
            loaded = [np.load("{}.{}.npy".format(alignment_path, idx)) for idx in range(len(sequences))]
            manual_alignments = np.transpose(loaded, [0, 2, 1])
        linear, alignments = self.model.run(
            inputs=sequences.astype(np.int32), input_lengths=input_lengths, speaker_id=speaker_ids,
            manual_alignments=manual_alignments, is_manual_attention=manual_alignments is not None)
        linear, alignments = linear.cpu().numpy(), alignments.cpu().numpy()
        if manual_attention_mode > 0:                                                    # :171-205
            new_alignments = manual_alignments_of(alignments, manual_attention_mode)
            linear, alignments = self.model.run(
                inputs=sequences.astype(np.int32), input_lengths=input_lengths, speaker_id=speaker_ids,
                manual_alignments=new_alignments, is_manual_attention=True)
            linear, alignments = linear.cpu().numpy(), alignments.cpu().numpy()
        # attention_trim (:242-262): frames to keep per utterance, from the argmax walk over the alignments (device kernel)
        self.spec_end_idx = None
        if attention_trim and end_of_sentence:
            self.spec_end_idx = self.attention_trim_end(alignments, [len(seq) for seq in sequences])
        # plot_graph_and_save_audio (:264): wav = wav[:spec_end_idx]; audio_out = inv_spectrogram(wav.T) -- on the GPU, whole batch at once
        self.wavs = None
        if vocode:
            self.wavs = self.inv_spectrogram(linear, self.spec_end_idx)
        return linear, alignments

    def inv_spectrogram(self, linear, spec_end_idx=None):
        """audio/__init__.py:54-56 for a batch [N, T, num_freq]; returns a list of 1-D float32 arrays (each cut to the samples its
        own frames produce when spec_end_idx is given: Griffin-Lim runs on the padded batch, frames past the end are silence-level)."""
        from .audio import GriffinLim
        if getattr(self, "_gl", None) is None:
            self._gl = GriffinLim(self.hparams, device=str(self.model.device))
        x = np.array(linear, np.float32, copy=True)
        if spec_end_idx is not None:
            for i, e in enumerate(spec_end_idx):
                x[i, int(e):] = 0.0                       # normalised 0 = min_level_db: the reference would not have synthesised these frames
        wav = self._gl.inv_spectrogram(x).cpu().numpy()
        hop = self._gl.num_samples(2)
        if spec_end_idx is None:
            return [w for w in wav]
        return [w[:hop * max(int(e) - 1, 1)] for w, e in zip(wav, spec_end_idx)]

    def attention_trim_end(self, alignments, sequence_lengths):
        """spec_end_idx = reduction_factor * j + 3 per utterance (synthesizer.py:242-262); alignments [N, T_in, T_dec]."""
        import ctypes as C
        import torch
        from . import _lib
        m = self.model
        al = m._as_dev(alignments, torch.float32)
        sl = m._as_dev(np.asarray(sequence_lengths, np.int32), torch.int32)
        N, T_in, n = al.shape
        out = torch.zeros((N,), dtype=torch.int32, device=al.device)
        p = lambda t: C.c_void_p(t.data_ptr())
        with torch.cuda.device(al.device):
            _lib.check(m._lib.taco_attention_trim(C.c_void_p(torch.cuda.current_stream().cuda_stream), p(al), p(sl), N, T_in, n,
                                                  self.hparams.reduction_factor, p(out)))
        return out.cpu().numpy()
