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
              per-dataset draw ratios.
* `NpzSource`  one data directory of the reference's own training examples -- `*.npz` with `tokens`, `mel`, `linear` and optionally
              `loss_coeff`, the files its preprocessing writes -- drawn as DataFeeder._get_next_example draws them
              (datafeeder.py:245-287); `frame_limits` / `filter_items` / `split_paths` / `data_ratios` are the path bookkeeping of
              get_path_dict (:26-76) and DataFeeder.__init__ (:104-121)."""
import os
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
    """sources: {name: callable returning the next Example}; ratios: {name: share of a group} (equal shares when None).

    The reference's two phases (datafeeder.py:219-231): while `step` (batches handed out so far) is below `initial_phase_step` every
    source contributes batch_size * batches_per_group // len(sources) examples -- and with `initial_data_greedy` all of them come from
    the first source whose name contains "krbook", if there is one; from then on source d contributes
    int(batch_size * batches_per_group * ratios[d])."""

    def __init__(self, sources, batch_size, reduction_factor, batches_per_group=32, ratios=None, seed=123, training=True,
                 initial_phase_step=0, initial_data_greedy=False, step=0):
        self.sources = dict(sources)
        self.batch_size, self.r, self.bpg, self.training = batch_size, reduction_factor, batches_per_group, training
        n = len(self.sources)
        self.ratios = {k: (1.0 / n if ratios is None else ratios[k]) for k in self.sources}
        self.initial_phase_step, self.initial_data_greedy = initial_phase_step, initial_data_greedy
        self.rng = np.random.RandomState(seed)
        self._pending = []
        self.step = step

    def next_group(self):
        group = []
        names = list(self.sources)
        initial = self.step < self.initial_phase_step
        for name in names:
            draw_from = name
            if self.initial_data_greedy and initial and any("krbook" in d for d in names):
                draw_from = [d for d in names if "krbook" in d][0]
            count = int(self.batch_size * self.bpg // len(names)) if initial else int(self.batch_size * self.bpg * self.ratios[name])
            for _ in range(count):
                group.append(self.sources[draw_from]())
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


# ---- the reference's on-disk examples (datasets/datafeeder.py:20-76,104-121,245-287) ----
def frame_limits(reduction_factor, min_iters, max_iters):
    """(min_n_frame, max_n_frame) of datafeeder.py:44-45 / :96-97: r * min_iters .. r * max_iters - r target frames."""
    return reduction_factor * min_iters, reduction_factor * max_iters - reduction_factor


def filter_items(items, min_n_frame, max_n_frame, min_tokens):
    """get_path_dict's filter (:47-48) over (path, n_frame, n_token) triples: frames within the limits and AT LEAST min_tokens tokens
    (the per-example filter of _get_next_example asks for MORE than min_tokens, :273 -- both kept as they are).  The reference's
    blacklist for the 'son' / 'yuinna' directories (:50-53) keeps an item if ANY of three substrings is absent from its path, which
    every path satisfies: a no-op, not reproduced.  Order: the reference collects the triples with Pool.imap_unordered (utils
    parallel_run), so the order of its filtered list -- and with it the train / test split -- is not deterministic; here the input
    order is kept."""
    return [p for p, n, nt in items if min_n_frame <= n <= max_n_frame and nt >= min_tokens]


def split_paths(paths, data_type, n_test):
    """datafeeder.py:66-71: the last n_test (= batch_size) paths of a directory are its test set."""
    if data_type == "train":
        return paths[:-n_test]
    if data_type == "test":
        return paths[-n_test:]
    raise Exception(" [!] Unkown data_type: {}".format(data_type))


def data_ratios(data_dirs, main_data=("",), main_data_greedy_factor=0):
    """DataFeeder.__init__ (:104-121): weight 1 per directory, + main_data_greedy_factor for every entry of hparams.main_data that
    occurs in the directory's name, normalised -- the draw ratios of a group once step >= initial_phase_step (:216-222)."""
    weight = {d: 1.0 for d in data_dirs}
    if main_data_greedy_factor > 0 and any(md in d for d in data_dirs for md in main_data):
        for md in main_data:
            for d in data_dirs:
                if md in d:
                    weight[d] += main_data_greedy_factor
    z = sum(weight.values())
    return {d: w / z for d, w in weight.items()}


class NpzSource(object):
    """Callable example source over the `.npz` files of one data directory, for GroupFeeder.

    The draw sequence of DataFeeder._get_next_example (datafeeder.py:245-287): the cursor starts at the THIRD path
    (`defaultdict(lambda: 2)`, :88); at the end of the list it wraps to 0 and, for training data, reshuffles the list with the
    FEEDER's generator (pass the GroupFeeder's `rng`); a path that no longer exists is skipped; with skip_path_filter (train.py:291,
    the lists were NOT pre-filtered) an example is taken only if min_n_frame <= frames <= max_n_frame and len(tokens) > min_tokens,
    without it every loadable file is taken.  `loss_coeff` defaults to 1 (:279-282).  One deliberate difference: a file that fails
    to load is skipped, not deleted (the reference calls remove_file on it, :267)."""

    def __init__(self, paths, speaker_id, rng, training=True, skip_path_filter=False, min_n_frame=0, max_n_frame=1 << 30, min_tokens=0):
        self.paths, self.speaker_id, self.rng, self.training = list(paths), speaker_id, rng, training
        self.skip_path_filter, self.min_n_frame, self.max_n_frame, self.min_tokens = skip_path_filter, min_n_frame, max_n_frame, min_tokens
        self.offset = 2
        self.skipped = []

    def __call__(self):
        while True:
            if self.offset >= len(self.paths):
                self.offset = 0
                if self.training:
                    self.rng.shuffle(self.paths)
            path = self.paths[self.offset]
            self.offset += 1
            if not os.path.exists(path):
                continue
            try:
                data = np.load(path)
                tokens, mel, linear = data["tokens"], data["mel"], data["linear"]
            except Exception:
                self.skipped.append(path)
                continue
            if not self.skip_path_filter:
                break
            if self.min_n_frame <= linear.shape[0] <= self.max_n_frame and len(tokens) > self.min_tokens:
                break
        coeff = data["loss_coeff"] if "loss_coeff" in data else 1
        return Example(tokens, coeff, mel, linear, self.speaker_id)


def open_data_dirs(data_dirs, batch_size, hparams, data_type="train", batches_per_group=32, seed=123, skip_path_filter=False, step=0):
    """A GroupFeeder over the reference's data directories, wired the way DataFeeder.__init__ wires itself (datafeeder.py:78-121): one
    generator (config.random_seed) shuffles every directory's path list once (training data), filters it by frames / tokens unless
    skip_path_filter, keeps all but the last `batch_size` paths for training (those are the test set), and is then shared by the example
    sources (reshuffles) and the batcher (bucketing); speaker id = position of the directory; the two draw phases of GroupFeeder from
    hparams.initial_phase_step / initial_data_greedy / main_data / main_data_greedy_factor."""
    import glob
    g = lambda k, d: getattr(hparams, k, d)
    r = hparams.reduction_factor
    lo, hi = frame_limits(r, g("min_iters", 30), hparams.max_iters)
    rng = np.random.RandomState(seed)
    sources = {}
    for idx, d in enumerate(data_dirs):
        paths = glob.glob("{}/*.npz".format(d))
        if data_type == "train":
            rng.shuffle(paths)
        if not skip_path_filter:
            items = []
            for p in paths:
                z = np.load(p)
                items.append((p, z["linear"].shape[0], len(z["tokens"])))
            paths = filter_items(items, lo, hi, g("min_tokens", 50))
        paths = split_paths(paths, data_type, batch_size)
        sources[d] = NpzSource(paths, idx if len(data_dirs) > 1 else None, rng, data_type == "train", skip_path_filter, lo, hi, g("min_tokens", 50))
    ratios = data_ratios(list(data_dirs), g("main_data", [""]), g("main_data_greedy_factor", 0))
    f = GroupFeeder(sources, batch_size, r, batches_per_group, ratios, seed, training=data_type == "train",
                    initial_phase_step=g("initial_phase_step", 8000), initial_data_greedy=g("initial_data_greedy", True), step=step)
    f.rng = rng
    return f
