"""Training-side batcher: the contract the reference's DataFeeder hands to `train.py` (datasets/datafeeder.py:210-243,289-328),
as plain host code with its own shape.

* `Example`   one utterance: token ids (EOS included), loss coefficient, mel [T, num_mels], linear [T, num_freq], speaker id.
* `collate`   a list of examples -> one `Batch` of dense arrays.  Inputs are zero-padded to the longest token row; targets are
              zero-padded to the next multiple of the reduction factor ABOVE the longest target (at least one padding frame, so
              the model always sees an end of utterance); `input_lengths` counts the tokens including the EOS -- the feeder's
              convention (the synthesizer instead uses the index of the EOS, synthesizer.py:120).
* `bucket`    length bucketing of one GROUP of examples (batch_size x batches_per_group of them): sort by target length, cut into
              consecutive batches (each batch then pads little), shuffle the ORDER of the batches, and -- for training data -- the
              rows inside each batch.
* `GroupFeeder`  iterator over batches that draws group after group from one or several example sources, with the reference's
              per-dataset draw ratios."""
import collections

import numpy as np

Example = collections.namedtuple("Example", "tokens loss_coeff mel linear speaker_id")
Example.__new__.__defaults__ = (None,)
Batch = collections.namedtuple("Batch", "inputs input_lengths loss_coeff mel_targets linear_targets speaker_id")


def padded_length(longest, reduction_factor):
    """Frames a target batch is padded to: the smallest multiple of r that is > longest... unless longest + 1 already is one."""
    return -(-(longest + 1) // reduction_factor) * reduction_factor


def _stack_rows(rows, length, dtype):
    first = np.asarray(rows[0])
    out = np.zeros((len(rows), length) + first.shape[1:], dtype)
    for i, r in enumerate(rows):
        r = np.asarray(r)
        out[i, :len(r)] = r
    return out


def collate(examples, reduction_factor):
    ex = [e if isinstance(e, Example) else Example(*e) for e in examples]
    t_in = max(len(e.tokens) for e in ex)
    t_out = padded_length(max(len(e.mel) for e in ex), reduction_factor)
    spk = None
    if ex[0].speaker_id is not None:
        spk = np.asarray([e.speaker_id for e in ex], np.int32)
    return Batch(_stack_rows([e.tokens for e in ex], t_in, np.int32),
                 np.asarray([len(e.tokens) for e in ex], np.int32),
                 np.asarray([e.loss_coeff for e in ex], np.float32),
                 _stack_rows([e.mel for e in ex], t_out, np.float32),
                 _stack_rows([e.linear for e in ex], t_out, np.float32), spk)


def bucket(examples, batch_size, rng, shuffle_rows=True):
    """One group of examples -> list of lists (batches), bucketed by target length."""
    order = sorted(range(len(examples)), key=lambda i: len(examples[i].mel if isinstance(examples[i], Example) else examples[i][2]))
    batches = [[examples[i] for i in order[k:k + batch_size]] for k in range(0, len(order), batch_size)]
    rng.shuffle(batches)
    if shuffle_rows:
        for b in batches:
            rng.shuffle(b)
    return batches


class GroupFeeder(object):
    """sources: {name: callable returning the next Example}; ratios: {name: share of a group} (equal shares when None)."""

    def __init__(self, sources, batch_size, reduction_factor, batches_per_group=32, ratios=None, seed=123, training=True):
        self.sources = dict(sources)
        self.batch_size, self.r, self.bpg, self.training = batch_size, reduction_factor, batches_per_group, training
        n = len(self.sources)
        self.ratios = {k: (1.0 / n if ratios is None else ratios[k]) for k in self.sources}
        self.rng = np.random.RandomState(seed)
        self._pending = []
        self.step = 0

    def next_group(self):
        group = []
        for name, draw in self.sources.items():
            for _ in range(int(self.batch_size * self.bpg * self.ratios[name])):
                group.append(draw())
        return bucket(group, self.batch_size, self.rng, shuffle_rows=self.training)

    def __iter__(self):
        return self

    def __next__(self):
        if not self._pending:
            self._pending = self.next_group()
            if not self._pending:
                raise StopIteration
        self.step += 1
        return collate(self._pending.pop(0), self.r)
