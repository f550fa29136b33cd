"""`Tacotron` -- the reference's model object (models/tacotron.py:16-343) re-hosted on libtaco_hip.

The reference builds a TF1 graph in `initialize()` and evaluates it with `sess.run`.  Here
`initialize()` creates the device weight pack and (when concrete inputs are given) runs the forward
eagerly; `run()` is the `sess.run([linear_outputs, alignments], feed_dict)` of
synthesizer.py:166-167.  Attribute names after `initialize()` are the reference's
(tacotron.py:242-251; is_manual_attention / manual_alignments :120-125).

PyTorch is used only for device memory and streams; all compute is in the HIP library."""
import ctypes as C
import time

import numpy as np
import torch

from . import _lib
from .hparams import hparams as default_hparams, EOS_ID
from .weights import random_weights


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _Plan(object):
    """Static buffers + one captured hipGraph for a (B, T_in, n_steps, manual?) shape."""

    def __init__(self, model, B, T_in, n, manual):
        hp, dev = model._hparams, model.device
        lib = model._lib
        r, M, F = hp.reduction_factor, hp.num_mels, hp.num_freq
        self.B, self.T_in, self.n = B, T_in, n
        self.inputs = torch.zeros((B, T_in), dtype=torch.int32, device=dev)
        self.lengths = torch.zeros((B,), dtype=torch.int32, device=dev)
        self.speaker_id = torch.zeros((B,), dtype=torch.int32, device=dev)
        self.manual = torch.zeros((B, n, T_in), dtype=torch.float32, device=dev) if manual else None
        self.mel = torch.empty((B, n * r, M), dtype=torch.float32, device=dev)
        self.linear = torch.empty((B, n * r, F), dtype=torch.float32, device=dev)
        self.align = torch.empty((B, T_in, n), dtype=torch.float32, device=dev)
        self.stop = torch.zeros((1,), dtype=torch.int32, device=dev)
        nbytes = int(lib.taco_workspace_bytes(model._handle, B, T_in, n))
        self.ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        self.ws_bytes = nbytes
        self.handle = C.c_void_p()
        self._lib = lib
        spk = self.speaker_id if model.num_speakers > 1 else None
        _lib.check(lib.taco_plan_create(
            model._handle, _ptr(self.inputs), _ptr(self.lengths), _ptr(spk), B, T_in, n, _ptr(self.manual),
            _ptr(self.mel), _ptr(self.linear), _ptr(self.align), _ptr(self.stop), _ptr(self.ws), nbytes,
            C.byref(self.handle)))
        self.num_nodes = lib.taco_plan_num_nodes(self.handle)

    def launch(self):
        _lib.check(self._lib.taco_plan_launch(self.handle, _stream()))

    def __del__(self):
        try:
            if self.handle:
                self._lib.taco_plan_destroy(self.handle)
        except Exception:
            pass


def _concurrent_streams(device, want, max_candidates=24):
    """`want` HIP streams that run concurrently with each other.

    HIP binds every stream to one of the GPU's hardware queues (4 by default, GPU_MAX_HW_QUEUES) and two streams
    on the same queue serialise.  The binding is static but not round-robin (tools/probe_lane_queues.py: eight
    consecutive streams landed as {0,3,7} {1,2,6} {4} {5}), and HIP has no call to query or set it -- so probe
    it: a short spin kernel on two streams takes 1x its duration if they are on different queues, 2x if they
    share one.  Greedy selection over up to `max_candidates` streams; if fewer than `want` distinct queues
    exist the remaining lanes share queues (still correct, just not concurrent)."""
    if want == 1:
        return [torch.cuda.Stream(device=device)]
    cyc = 400_000                                   # ~0.17 ms

    def spin(streams):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for st in streams:
            with torch.cuda.stream(st):
                torch.cuda._sleep(cyc)
        torch.cuda.synchronize(device)
        return time.perf_counter() - t0

    cands = []
    for _ in range(max_candidates):
        st = torch.cuda.Stream(device=device)
        if all(st.cuda_stream != c.cuda_stream for c in cands):
            cands.append(st)
    spin(cands[:1])
    alone = min(spin(cands[:1]) for _ in range(3))
    chosen = [cands[0]]
    for st in cands[1:]:
        if len(chosen) == want:
            break
        if all(min(spin([st, c]), spin([st, c])) < 1.5 * alone for c in chosen):
            chosen.append(st)
    rest = [c for c in cands if all(c is not x for x in chosen)]
    while len(chosen) < want:                       # fewer hardware queues than lanes
        chosen.append(rest.pop(0) if rest else torch.cuda.Stream(device=device))
    return chosen


class PlanPool(object):
    """`lanes` independent forwards of one shape in flight, each with its own hipGraph plan, buffers and
    HIP stream.

    Two engines, two ways to use the chip.  The persistent engine (one whole-chip launch for the decoder loop, one
    for the post-net scan) finishes a C2 forward in 2.7 ms but owns every CU while those two kernels run: the library keeps the
    whole-chip kernels of a process in a total order per device (`taco_plan_whole_chip`, an event wait in front of a replay and an
    event record behind it), so forwards of different lanes on this engine take turns -- correct from any number of streams and
    threads, but with nothing gained over one lane.  The launch-per-stage engine leaves most CUs idle per forward (8.6 ms alone) and
    is the one several lanes can fill (4 lanes: 2.9 ms per forward).  `engine="auto"` therefore captures the plans of a one-lane pool
    with the model's current engine and the plans of a multi-lane pool with the launch-per-stage engine; "persistent" / "launch"
    force one.  The model object is read-only during a forward, so lanes share it.

    The reference serves one `sess.run` at a time (synthesizer.py:166-167); this is the same call with
    several requests outstanding.  submit() enqueues and returns immediately; result() waits for that lane."""

    def __init__(self, model, B, T_in, n_steps=None, lanes=1, coalesce=1, engine="auto"):
        """coalesce > 1: every lane's plan serves `coalesce` requests of B rows at once (one forward over coalesce*B rows: rows are
        independent at inference, and the decoder stages and scans cost almost the same for twice the rows).  submit() then
        returns a ticket (lane, slot), the lane is launched when its last slot is filled (or by flush()), and result(ticket)
        returns that request's rows -- the same function of the request as serving it alone (rows never interact; a layer may pick a
        different tile shape for the larger row count, so equality is to rounding, ~1e-6, not bitwise)."""
        if model._handle is None:
            raise RuntimeError("initialize() must be called first")
        if lanes < 1 or coalesce < 1:
            raise ValueError("lanes and coalesce must be >= 1")
        if engine not in ("auto", "persistent", "launch"):
            raise ValueError("engine must be 'auto', 'persistent' or 'launch'")
        current = getattr(model, "_decoder_engine", (1, 0))
        if engine == "launch" or (engine == "auto" and lanes > 1):
            capture_with = (0, 0)
        elif engine == "persistent" and current[0] == 0:
            capture_with = (1, 0)
        else:
            capture_with = current
        self.engine = "launch-per-stage" if capture_with[0] == 0 else "persistent"
        n = model._hparams.max_iters if n_steps is None else n_steps
        self.model, self.B, self.T_in, self.n, self.lanes, self.coalesce = model, B, T_in, n, lanes, coalesce
        self.streams, self.plans, self.done, self.pending = [], [], [], []
        self.filled = [0] * lanes            # slots of the lane that hold a submitted request
        self.taken = [0] * lanes             # of those, how many results were collected
        self.slot_stop = []
        with torch.cuda.device(model.device):
            self.streams = _concurrent_streams(model.device, lanes)
            if capture_with != current:       # a plan keeps the engine it was captured with
                _lib.check(model._lib.taco_debug_set_decoder_persist(model._handle, *capture_with))
            try:
                for st in self.streams:
                    with torch.cuda.stream(st):
                        self.plans.append(_Plan(model, B * coalesce, T_in, n, False))
                        self.plans[-1].launch()      # first replay uploads the graph: keep that out of the serving path
                    self.done.append(torch.cuda.Event())
                    self.pending.append(False)
                    self.slot_stop.append(torch.zeros((coalesce,), dtype=torch.int32, device=model.device))
                torch.cuda.synchronize()
            finally:
                if capture_with != current:
                    _lib.check(model._lib.taco_debug_set_decoder_persist(model._handle, *current))
        self._next = 0
        self._filling = None

    def next_lane(self):
        lane = self._next
        self._next = (self._next + 1) % self.lanes
        return lane

    def launch(self, lane):
        """Replay lane's plan on its stream over whatever its input buffers hold."""
        hp = self.model._hparams
        with torch.cuda.stream(self.streams[lane]):
            p = self.plans[lane]
            p.launch()
            if self.coalesce > 1:        # per-request stop steps (the plan's own covers all its rows)
                _lib.check(self.model._lib.taco_stop_steps(_stream(), _ptr(p.mel), p.B, self.n, hp.reduction_factor * hp.num_mels,
                                                           self.B, _ptr(self.slot_stop[lane])))
            self.done[lane].record()
        self.pending[lane] = True
        if self.filled[lane] == 0:
            self.filled[lane] = self.coalesce       # launched directly (bench): every slot counts
        self.taken[lane] = 0

    def submit(self, inputs, input_lengths, speaker_id=None, lane=None):
        """Enqueue one request of B rows.  Returns what result() takes: the lane (coalesce == 1) or the ticket (lane, slot).
        Inputs may be host or device arrays."""
        if self.coalesce > 1:
            if lane is not None:
                raise ValueError("with coalesce > 1 the pool picks the lane")
            if self._filling is None:
                self._filling = self.next_lane()
            lane = self._filling
        else:
            lane = self.next_lane() if lane is None else lane
        if self.pending[lane]:
            raise RuntimeError("lane %d still holds an uncollected result; call result() for it first" % lane)
        m, plan = self.model, self.plans[lane]
        ids = m._as_dev(inputs, torch.int32)
        lens = m._as_dev(input_lengths, torch.int32)
        if tuple(ids.shape) != (self.B, self.T_in):
            raise Exception("inputs must be [%d, %d], got shape %s" % (self.B, self.T_in, tuple(ids.shape)))
        slot = self.filled[lane] if self.coalesce > 1 else 0
        rows = slice(slot * self.B, (slot + 1) * self.B)
        cur = torch.cuda.current_stream(m.device)
        with torch.cuda.stream(self.streams[lane]):
            self.streams[lane].wait_stream(cur)      # ids/lens may have been produced on the caller's stream
            plan.inputs[rows].copy_(ids, non_blocking=True)
            plan.lengths[rows].copy_(lens, non_blocking=True)
            # the temporaries were allocated on the caller's stream and are read here on the lane's: tell the caching allocator,
            # or it may hand their blocks out again while these copies are still pending
            ids.record_stream(self.streams[lane])
            lens.record_stream(self.streams[lane])
            if m.num_speakers > 1:
                if speaker_id is None:
                    plan.speaker_id[rows].zero_()
                else:
                    spk = m._as_dev(speaker_id, torch.int32)
                    plan.speaker_id[rows].copy_(spk, non_blocking=True)
                    spk.record_stream(self.streams[lane])
        if self.coalesce == 1:
            self.filled[lane] = 1
            self.launch(lane)
            return lane
        self.filled[lane] += 1
        if self.filled[lane] == self.coalesce:
            self.flush()
        return (lane, slot)

    def flush(self):
        """coalesce > 1: launch the lane that is being filled even if some of its slots are empty (they recompute stale rows)."""
        if self._filling is not None and self.filled[self._filling] > 0:
            lane, self._filling = self._filling, None
            self.launch(lane)

    def result(self, ticket, copy=True):
        """Waits for the forward that holds the request.  Returns dict(linear, mel, alignments, stop_step); with copy=False the
        tensors are views of the lane's own buffers and are overwritten by the lane's next forward."""
        lane, slot = (ticket, 0) if self.coalesce == 1 else ticket
        if lane == self._filling:
            self.flush()
        if not self.pending[lane] or slot >= self.filled[lane]:
            raise RuntimeError("lane %d slot %d has no forward in flight" % (lane, slot))
        self.done[lane].synchronize()
        p = self.plans[lane]
        rows = slice(slot * self.B, (slot + 1) * self.B)
        f = (lambda t: t.clone()) if copy else (lambda t: t)
        with torch.cuda.device(self.model.device):
            plan_stop = int(p.stop.item())
            if plan_stop < 0:
                self._failed(lane)
            stop = plan_stop if self.coalesce == 1 else int(self.slot_stop[lane][slot].item())
            out = {"linear": f(p.linear[rows]), "mel": f(p.mel[rows]), "alignments": f(p.align[rows]), "stop_step": stop}
        self.taken[lane] += 1
        if self.taken[lane] >= self.filled[lane]:
            self.pending[lane] = False
            self.filled[lane] = 0
        return out

    def _failed(self, lane):
        """The lane's forward reported a device error through its stop word (taco_abi.h: negative = a persistent kernel gave up; the
        error word is sticky, so every forward enqueued before this acknowledgement reports it too).  Acknowledge (clears the word, so
        that forwards enqueued from now on start clean), drop the lane's result and raise."""
        self.pending[lane] = False
        self.filled[lane] = 0
        self.taken[lane] = 0
        try:
            self.model.check_device_errors()
        except _lib.TacoError:
            pass
        raise _lib.TacoError(_lib.TACO_ERR_HIP, "a persistent kernel timed out waiting for a peer workgroup during the forward of lane %d; "
                             "its outputs are invalid (the error is acknowledged: resubmit, or capture the pool with engine='launch')" % lane)

    def wait_all(self):
        """Waits for every forward in flight; raises if one of them reported a device error."""
        self.flush()
        bad = None
        for lane in range(self.lanes):
            if self.pending[lane]:
                self.done[lane].synchronize()
                if bad is None and int(self.plans[lane].stop.item()) < 0:
                    bad = lane
        if bad is not None:
            self._failed(bad)

    def close(self):
        self.wait_all()
        self.plans = []


class Tacotron(object):
    MAX_PLANS = 16     # least-recently-used bound of the per-shape plan cache (plan_for)

    def __init__(self, hparams=None):
        self._hparams = hparams if hparams is not None else default_hparams
        self._lib = None
        self._handle = None
        self._weights = None
        self._plans = {}
        self.device = None
        self.num_speakers = 1
        self.is_manual_attention = False
        self.manual_alignments = None

    # ---- weights (tf variables + Saver.restore in the reference) ----
    def load_weights(self, weights):
        """dict name -> float32 array (canonical names, weights.py).  Must precede initialize()."""
        if self._handle is not None:
            raise RuntimeError("load_weights() must be called before initialize()")
        self._weights = {k: np.array(v, dtype=np.float32, order='C') for k, v in weights.items()}   # keeps rank 0

    def _build(self, num_speakers, device, seed=0):
        if not torch.cuda.is_available():
            raise RuntimeError("taco_amd needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
        self._lib = _lib.load_library()
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        chp = _lib.to_c_hparams(self._hparams, num_speakers)
        h = C.c_void_p()
        _lib.check(self._lib.taco_model_create(C.byref(chp), self.device.index or 0, C.byref(h)))
        self._handle = h
        if self._weights is None:   # variables keep their initial values (synthesizer.py:65 / train.py:187)
            self._weights = random_weights(self._hparams, num_speakers, seed)
        for name, arr in self._weights.items():
            shp = (C.c_int64 * max(arr.ndim, 1))(*arr.shape)
            _lib.check(self._lib.taco_model_set_weight(h, name.encode(), arr.ctypes.data_as(C.c_void_p), shp, arr.ndim))
        _lib.check(self._lib.taco_model_finalize(h))
        self._weights = None

    def initialize(self, inputs, input_lengths, num_speakers, speaker_id,
                   mel_targets=None, linear_targets=None, loss_coeff=None,
                   rnn_decoder_test_mode=False, is_randomly_initialized=False, device=None):
        """models/tacotron.py:21-25.  `inputs` int32 [B,T_in] (PAD=0, EOS=1) and `input_lengths`
        int32 [B] may be None ("placeholders", synthesizer.py:39-44): then only the model is built
        and `run()` supplies the feed later."""
        is_training = linear_targets is not None                          # tacotron.py:26
        if is_training:
            return self._initialize_training(inputs, input_lengths, num_speakers, speaker_id, mel_targets, linear_targets,
                                             loss_coeff, rnn_decoder_test_mode, is_randomly_initialized, device)
        self.is_randomly_initialized = is_randomly_initialized
        self.num_speakers = num_speakers
        if self._handle is None:
            self._build(num_speakers, device)
        self.inputs, self.speaker_id, self.input_lengths = inputs, speaker_id, input_lengths
        self.loss_coeff, self.mel_targets, self.linear_targets = loss_coeff, mel_targets, linear_targets
        self.mel_outputs = self.linear_outputs = self.alignments = None
        self.final_decoder_state = None
        if inputs is not None:
            self.run()
        return self

    # ---- training graph (tacotron.py:26,199-202,274-336; train.py:145-166,215-219) ----
    def _initialize_training(self, inputs, input_lengths, num_speakers, speaker_id, mel_targets, linear_targets, loss_coeff,
                             rnn_decoder_test_mode, is_randomly_initialized, device):
        """`linear_targets` given => is_training (tacotron.py:26): batch-statistics BatchNorm, TacoTrainingHelper teacher forcing
        (or, with rnn_decoder_test_mode, the decoder's own outputs fed back: the test model of train.py:158-166), decoder
        steps = T_out / r.  The parameters move into a Trainer (one flat device buffer); add_loss() / add_optimizer() expose the
        reference's attributes; train_step() is the sess.run of train.py:217-219."""
        from .trainer import Trainer
        self.is_randomly_initialized = is_randomly_initialized
        self.num_speakers = num_speakers
        self.rnn_decoder_test_mode = bool(rnn_decoder_test_mode)
        if getattr(self, "_trainer", None) is None:
            w = self._weights if self._weights is not None else random_weights(self._hparams, num_speakers, seed=0)
            dev = device or ("cuda:%d" % torch.cuda.current_device())
            self._trainer = Trainer(self._hparams, w, device=dev, is_randomly_initialized=is_randomly_initialized, num_speakers=num_speakers)
            self.device = self._trainer.device
        self.inputs, self.speaker_id, self.input_lengths = inputs, speaker_id, input_lengths
        self.loss_coeff, self.mel_targets, self.linear_targets = loss_coeff, mel_targets, linear_targets
        self.mel_outputs = self.linear_outputs = self.alignments = None
        self.loss = self.mel_loss = self.linear_loss = self.loss_without_coeff = None
        if inputs is not None:
            self._train_forward()
        return self

    def share_variables_with(self, other):
        """train.py:158-159 builds the test model under reuse=True: same variables as the training model."""
        self._trainer = other._trainer
        self.device = other.device
        return self

    def _train_forward(self):
        tr = self._trainer
        losses = tr.forward_backward(self.inputs, self.input_lengths, self.mel_targets, self.linear_targets, self.loss_coeff,
                                     backward=False, keep_outputs=True, rnn_decoder_test_mode=self.rnn_decoder_test_mode,
                                     speaker_id=self.speaker_id)
        self.mel_outputs, self.linear_outputs, self.alignments = tr.mel_outputs, tr.linear_outputs, tr.alignments
        self._losses = losses.clone()
        return self._losses

    def add_loss(self):
        """tacotron.py:274-302: sets loss, mel_loss, linear_loss, loss_without_coeff (device scalars) for the current feed."""
        if getattr(self, "_trainer", None) is None:
            raise RuntimeError("initialize(..., mel_targets, linear_targets) must be called first")
        if self.inputs is not None:
            l = self._losses if getattr(self, "_losses", None) is not None else self._train_forward()
            self.loss, self.mel_loss, self.linear_loss, self.loss_without_coeff = l[0], l[1], l[2], l[3]
        return self

    def add_optimizer(self, global_step=0):
        """tacotron.py:305-336: learning-rate schedule, Adam(beta1, beta2), clip_by_global_norm(1.0); `optimize` applies one update
        to the current feed (with the BatchNorm moving-average updates it depends on, :334)."""
        tr = self._trainer
        tr.adam.global_step = int(global_step)
        self.learning_rate = tr.learning_rate
        self.optimize = lambda: self.train_step()
        self.gradients = tr.grads
        return self

    def train_step(self, inputs=None, input_lengths=None, mel_targets=None, linear_targets=None, loss_coeff=None, speaker_id=None):
        """sess.run([global_step, loss_without_coeff, optimize]) of train.py:217-219 -> (global_step, loss_without_coeff)."""
        if self.rnn_decoder_test_mode:
            raise _lib.TacoError(_lib.TACO_ERR_UNSUPPORTED, "the rnn_decoder_test_mode model is forward-only (train.py:158-166 never optimises it)")
        if inputs is not None:
            self.inputs, self.input_lengths, self.mel_targets, self.linear_targets, self.loss_coeff = \
                inputs, input_lengths, mel_targets, linear_targets, loss_coeff
            if speaker_id is not None:
                self.speaker_id = speaker_id
        tr = self._trainer
        step, lwc = tr.train_step(self.inputs, self.input_lengths, self.mel_targets, self.linear_targets, self.loss_coeff,
                                  speaker_id=self.speaker_id)
        l = tr.losses
        self.loss, self.mel_loss, self.linear_loss, self.loss_without_coeff = l[0], l[1], l[2], l[3]
        self.learning_rate = tr.learning_rate
        self._losses = None
        return step, lwc

    def trained_weights(self):
        """Current parameters as a dict (tf.train.Saver.save of train.py:242-244 -> weights.save_weights)."""
        return self._trainer.get_weights()

    def get_dummy_feed_dict(self):
        """tacotron.py:338-343."""
        return {"is_manual_attention": False, "manual_alignments": np.zeros([1, 1, 1])}

    # ---- sess.run([linear_outputs, alignments], feed_dict) ----
    def _as_dev(self, x, dtype):
        if x is None:
            return None
        if not torch.is_tensor(x):
            x = torch.as_tensor(np.asarray(x))
        return x.to(device=self.device, dtype=dtype).contiguous()

    def plan_for(self, B, T_in, n_steps=None, manual=False):
        n = self._hparams.max_iters if n_steps is None else n_steps
        key = (B, T_in, n, bool(manual))
        plan = self._plans.pop(key, None)
        if plan is None:
            # every distinct shape owns a captured graph plus output / workspace buffers (tens of MB): keep the most recently used
            # MAX_PLANS of them (a server that synthesises arbitrary text lengths would otherwise grow without bound)
            while len(self._plans) >= self.MAX_PLANS:
                self._plans.pop(next(iter(self._plans)))
            with torch.cuda.device(self.device):
                plan = _Plan(self, B, T_in, n, manual)
        self._plans[key] = plan          # (re)inserted last = most recently used
        return plan

    def plan_pool(self, B, T_in, n_steps=None, lanes=1, coalesce=1, engine="auto"):
        """`lanes` forwards of this shape in flight at once, each serving `coalesce` requests of B rows (PlanPool)."""
        return PlanPool(self, B, T_in, n_steps, lanes, coalesce, engine)

    def run(self, inputs=None, input_lengths=None, speaker_id=None, manual_alignments=None,
            is_manual_attention=None, n_steps=None, honor_stop=True):
        """One forward.  Returns (linear_outputs, alignments) as device tensors and refreshes the
        public attributes.  `manual_alignments` [B,T_dec,T_in] + `is_manual_attention`
        (rnn_wrappers.py:313-317).  With honor_stop the outputs are cut where the reference's stop rule
        (helpers.py:29 + dynamic_decode) would have ended the loop (one host sync)."""
        if self._handle is None:
            raise RuntimeError("initialize() must be called first")
        inputs = self.inputs if inputs is None else inputs
        input_lengths = self.input_lengths if input_lengths is None else input_lengths
        speaker_id = self.speaker_id if speaker_id is None else speaker_id
        if is_manual_attention is None:
            is_manual_attention = self.is_manual_attention
        if manual_alignments is None:
            manual_alignments = self.manual_alignments
        ids = self._as_dev(inputs, torch.int32)
        if ids.dim() != 2:
            raise Exception("inputs must be [batch, time], got shape %s" % (tuple(ids.shape),))
        B, T_in = ids.shape
        lens = self._as_dev(input_lengths, torch.int32)
        manual = bool(is_manual_attention)
        plan = self.plan_for(B, T_in, n_steps, manual)
        with torch.cuda.device(self.device):
            plan.inputs.copy_(ids)
            plan.lengths.copy_(lens)
            if self.num_speakers > 1:
                if speaker_id is None:      # placeholder_with_default(zeros) (synthesizer.py:43-44)
                    plan.speaker_id.zero_()
                else:
                    plan.speaker_id.copy_(self._as_dev(speaker_id, torch.int32))
            if manual:
                ma = self._as_dev(manual_alignments, torch.float32)
                if tuple(ma.shape) != tuple(plan.manual.shape):
                    raise Exception("manual_alignments must be [B, T_dec, T_in] = %s, got %s"
                                    % (tuple(plan.manual.shape), tuple(ma.shape)))
                plan.manual.copy_(ma)
            plan.launch()
            mel, linear, align = plan.mel, plan.linear, plan.align
            self.stop_step = plan.n
            # the stop word doubles as the forward's error latch (negative: a persistent kernel gave up, taco_abi.h); reading it is
            # the one host sync of a run(), with or without honor_stop
            stop = int(plan.stop.item())
            if stop < 0:
                self.check_device_errors()
                raise _lib.TacoError(_lib.TACO_ERR_HIP, "a persistent kernel reported error %d; outputs are invalid" % -stop)
            if honor_stop:
                self.stop_step = stop
                if stop < plan.n:   # dynamic_decode ended early: re-run the post-net on the frames that exist
                    r = self._hparams.reduction_factor
                    mel = mel.view(B, plan.n, -1)[:, :stop].reshape(B, stop * r, self._hparams.num_mels).contiguous()
                    align = align[:, :, :stop].contiguous()
                    linear = self.postnet(mel, speaker_id=plan.speaker_id if self.num_speakers > 1 else None)
        self.mel_outputs, self.linear_outputs, self.alignments = mel, linear, align
        return linear, align

    # ---- stage / op level access (used by the parity tests) ----
    def _stage_ws(self, B, T):
        n = int(self._lib.taco_stage_workspace_bytes(self._handle, B, T))
        return torch.empty((n,), dtype=torch.uint8, device=self.device), n

    def encoder(self, inputs, input_lengths, speaker_id=None):
        ids, lens = self._as_dev(inputs, torch.int32), self._as_dev(input_lengths, torch.int32)
        spk = self._as_dev(speaker_id, torch.int32)
        B, T = ids.shape
        out = torch.empty((B, T, 2 * self._hparams.enc_rnn_size), dtype=torch.float32, device=self.device)
        ws, n = self._stage_ws(B, T)
        _lib.check(self._lib.taco_encoder_forward(self._handle, _stream(), _ptr(ids), _ptr(lens), _ptr(spk), B, T,
                                                  _ptr(out), _ptr(ws), n))
        return out

    def decoder(self, encoder_out, n_steps, speaker_id=None, manual_alignments=None, teacher_frames=None, debug=False):
        hp = self._hparams
        enc = self._as_dev(encoder_out, torch.float32)
        B, T_in, _ = enc.shape
        spk = self._as_dev(speaker_id, torch.int32)
        man = self._as_dev(manual_alignments, torch.float32)
        tf_ = self._as_dev(teacher_frames, torch.float32)
        r, M = hp.reduction_factor, hp.num_mels
        mel = torch.empty((B, n_steps * r, M), dtype=torch.float32, device=self.device)
        align = torch.empty((B, T_in, n_steps), dtype=torch.float32, device=self.device)
        stop = torch.zeros((1,), dtype=torch.int32, device=self.device)
        dbgw = hp.attention_state_size + 2 * hp.enc_rnn_size + hp.dec_layer_num * hp.dec_rnn_size
        dbg = torch.empty((n_steps, B, dbgw), dtype=torch.float32, device=self.device) if debug else None
        ws, n = self._stage_ws(B, max(T_in, n_steps))
        _lib.check(self._lib.taco_decoder_forward(self._handle, _stream(), _ptr(enc), _ptr(spk), B, T_in, n_steps,
                                                  _ptr(man), _ptr(tf_), _ptr(mel), _ptr(align), _ptr(stop), _ptr(dbg),
                                                  _ptr(ws), n))
        return mel, align, stop, dbg

    def postnet(self, mel, return_post=False, speaker_id=None):
        hp = self._hparams
        mel = self._as_dev(mel, torch.float32)
        spk = self._as_dev(speaker_id, torch.int32)
        B, T, _ = mel.shape
        lin = torch.empty((B, T, hp.num_freq), dtype=torch.float32, device=self.device)
        post = torch.empty((B, T, 2 * hp.post_rnn_size), dtype=torch.float32, device=self.device) if return_post else None
        ws, n = self._stage_ws(B, T)
        _lib.check(self._lib.taco_postnet_forward(self._handle, _stream(), _ptr(mel), _ptr(spk), B, T, _ptr(lin), _ptr(post),
                                                  _ptr(ws), n))
        return (lin, post) if return_post else lin

    def set_batch_invariant(self, on=True):
        """taco_model_set_batch_invariant: outputs bitwise independent of the batch position / shard a row lands in, at any batch size (the fused
        point-wise kernel then starts every tile's contraction at step 0; default off: equal under permutation to fp32 rounding only).
        Cached plans are dropped: a plan keeps the setting it was captured with."""
        _lib.check(self._lib.taco_model_set_batch_invariant(self._handle, 1 if on else 0))
        self._plans.clear()

    def set_decoder_engine(self, mode=1, rows_per_group=0):
        """Decoder loop engine: 1 = one persistent weight-stationary launch when the configuration fits (default), 0 = one launch per
        stage, 2 = persistent with write-through exchanges.  Cached plans are dropped (they captured the previous engine)."""
        _lib.check(self._lib.taco_debug_set_decoder_persist(self._handle, int(mode), int(rows_per_group)))
        self._decoder_engine = (int(mode), int(rows_per_group))
        self._plans.clear()

    def decoder_engine_info(self):
        """After a forward (synchronises): {'protocol': 0 none / 1 XCD-local / 2 write-through, 'per_xcd': [...], 'has_pack': bool,
        'compute_units': CUs of the device -- the whole-chip persistent kernels run only on 256 (an unpartitioned MI355X)}."""
        torch.cuda.synchronize(self.device)
        v = (C.c_int * 16)()
        _lib.check(self._lib.taco_debug_decoder_info(self._handle, v))
        return {"protocol": int(v[0]), "per_xcd": [int(x) for x in v[1:9]], "has_pack": bool(v[15]), "compute_units": int(v[14])}

    def engine_plan(self, batch, t_in, t_mel=None, manual=False):
        """One line saying which engine a forward of this shape would run on and, if not the persistent whole-chip kernels, why not
        (widths, rows per launch, LDS, compute units of the device, debug switches).  Nothing is launched; PlanPool(lanes > 1) additionally
        switches its lanes to the launch-per-stage engine (two whole-chip kernels cannot share the chip)."""
        buf = C.create_string_buffer(1024)
        t_mel = t_mel if t_mel is not None else self._hparams.max_iters * self._hparams.reduction_factor
        _lib.check(self._lib.taco_model_engine_plan(self._handle, int(batch), int(t_in), int(t_mel), 1 if manual else 0, buf, 1024))
        return buf.value.decode()

    def decoder_trace(self, enable=True, read=False, scan=False):
        """Phase stamps (shader clocks) of group 0 / member 0 of the persistent decoder (scan=True: of the post-net scan, which has
        its own half of the buffer), 8 steps x 16 slots.  Cached plans are dropped whenever the setting changes: a plan captured
        with tracing on keeps stamping on every replay, one captured with it off never does."""
        out = (C.c_longlong * 128)() if read else None
        if read:
            torch.cuda.synchronize(self.device)
        if bool(enable) != getattr(self, "_trace_on", False):
            self._plans.clear()
            self._trace_on = bool(enable)
        _lib.check(self._lib.taco_debug_decoder_trace(self._handle, (1 if enable else 0) | (2 if scan else 0), out))
        return np.array(out[:], np.int64).reshape(8, 16) if read else None

    def raise_device_error_for_test(self, value=2):
        """Test hook: sets the sticky device error word, as a persistent kernel whose bounded spin expired does."""
        _lib.check(self._lib.taco_debug_raise_device_error(self._handle, int(value)))

    def check_device_errors(self):
        """Synchronises and raises if a persistent kernel's bounded spin expired (outputs invalid)."""
        v = C.c_int(0)
        _lib.check(self._lib.taco_model_device_errors(self._handle, C.byref(v)))
        if v.value:
            raise _lib.TacoError(_lib.TACO_ERR_HIP, "a persistent kernel timed out waiting for a peer workgroup; outputs are invalid")

    def close(self):
        self._plans.clear()
        if self._handle is not None and self._lib is not None:
            self._lib.taco_model_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def create_model(hparams):
    """models/__init__.py:6-7."""
    return Tacotron(hparams)


def input_lengths_from_tokens(sequences):
    """synthesizer.py:120: index of the first EOS (the EOS itself is outside the length)."""
    return np.argmax(np.asarray(sequences) == EOS_ID, 1).astype(np.int32)
