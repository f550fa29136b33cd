"""`Trainer` -- the training half of the reference's model object: `initialize(..., mel_targets, linear_targets)`
+ `add_loss()` + `add_optimizer(global_step)` (models/tacotron.py:21-336) and the step of train.py:215-219
(`sess.run([global_step, loss_without_coeff, optimize])`), on libtaco_hip's training path.

All parameters live in ONE flat fp32 device tensor (`params`, layout = taco_model_weight_name order) and all
gradients in a second one (`grads`) -- the single RCCL all-reduce bucket of the data-parallel step.
Per step: taco_train_forward_backward (teacher-forced forward with batch-statistics BatchNorm, L1 losses, full
backward) -> all-reduce of `grads` / world size -> clip_by_global_norm + Adam (taco_adam_step_f32) ->
taco_train_refresh (weight packs regenerated from the flat parameters).  PyTorch is device memory, streams and
torch.distributed only."""
import ctypes as C
import hashlib
import os

import numpy as np
import torch

from . import _lib
from .train_ops import FlatAdam, allreduce_gradients, train_state_paths, list_train_checkpoints, prune_train_checkpoints


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Trainer(object):
    def __init__(self, hparams, weights, device="cuda:0", is_randomly_initialized=True, num_speakers=1):
        """`weights`: dict canonical name -> array (weights.random_weights / load_weights).  num_speakers > 1: model_type 'deepvoice' or 'simple' (pass speaker_id to every step)."""
        self.hp = hparams
        self.num_speakers = int(num_speakers)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.TacoError(_lib.TACO_ERR_ARG, "the training path runs on a GPU (got device %s); there is no CPU fallback" % device)
        self._lib = _lib.load_library()
        chp = _lib.to_c_hparams(hparams, self.num_speakers)
        self._h = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.taco_train_create(C.byref(chp), idx, C.byref(self._h)))
        mh = C.c_void_p(self._lib.taco_train_model(self._h))
        self.spec, self.offsets = [], {}
        buf = C.create_string_buffer(256)
        shp = (C.c_int64 * 4)()
        nd = C.c_int()
        off = C.c_size_t()
        for i in range(self._lib.taco_model_num_weights(mh)):
            _lib.check(self._lib.taco_model_weight_name(mh, i, buf, 256, shp, C.byref(nd)))
            name, shape = buf.value.decode(), tuple(int(shp[d]) for d in range(nd.value))
            _lib.check(self._lib.taco_train_param_offset(self._h, name.encode(), C.byref(off)))
            self.spec.append((name, shape))
            self.offsets[name] = (int(off.value), int(np.prod(shape)) if shape else 1)
        self.num_params = int(self._lib.taco_train_num_params(self._h))
        self.params = torch.zeros(self.num_params, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros_like(self.params)
        self.losses = torch.zeros(4, dtype=torch.float32, device=self.device)
        self.set_weights(weights)
        g = lambda k, d: getattr(hparams, k, d)
        self.adam = FlatAdam(self.params, g("initial_learning_rate", 0.002), g("adam_beta1", 0.9), g("adam_beta2", 0.999), 1e-8,
                             g("decay_learning_rate_mode", 0), is_randomly_initialized, 1.0)
        self._ws = None          # workspace of the (captured or only) step
        self._ws_eager = None    # workspace of eager calls made while a captured step exists
        self._ws_live = None
        self._graph = None
        self._capturing = False
        self._sync_cb = None
        self._sync_err = None
        self.sync_exchanges = 0
        self.mel_outputs = self.linear_outputs = self.alignments = None

    # ---- data-parallel SyncBN ----
    def enable_sync_bn(self, on=True, group=None):
        """BatchNorm statistics over the GLOBAL batch of the data-parallel group: one all-reduce per BatchNorm layer in the forward
        pass (rank means, centred sums and squared means in one vector; the widths of a conv bank are one layer here) and one per
        layer / per conv bank in the backward pass (sum dy, sum dy*xhat), over `group` (RCCL on GPUs) inside the step, so a step over W shards equals the
        reference's single-device step over the whole batch (modules.py:131, train.py:145-166) -- up to summation order.  Without it
        (default) each rank normalises with its own rows.  Returns True when synchronisation is active (world size > 1)."""
        import torch.distributed as dist
        world = dist.get_world_size(group) if (on and dist.is_available() and dist.is_initialized()) else 1
        if world <= 1:
            _lib.check(self._lib.taco_train_set_sync_bn(self._h, None, None, 1))
            self._sync_cb = None
            return False
        if getattr(self, "_graph", None) is not None:
            raise _lib.TacoError(_lib.TACO_ERR_STATE, "SyncBN calls back into the host: it cannot be combined with a captured step")

        self.sync_exchanges = 0       # all-reduces issued by SyncBN since it was enabled (12 per forward+backward at the default depth)

        def _sum(user, ptr, n):
            self.sync_exchanges += 1
            try:            # the vector lives in this step's workspace: view it as a tensor and sum it over the ranks, stream-ordered
                wsb = self._ws_live
                off = int(ptr) - wsb.data_ptr()
                if off < 0 or off + 4 * n > wsb.numel():
                    raise RuntimeError("statistics vector outside the workspace")
                dist.all_reduce(wsb[off:off + 4 * n].view(torch.float32), op=dist.ReduceOp.SUM, group=group)
            except Exception as e:     # an exception must not unwind through the C frames: it is re-raised after the call returns
                self._sync_err = e
        self._sync_cb = _lib.SYNC_SUM_FN(_sum)
        self._sync_group = group
        _lib.check(self._lib.taco_train_set_sync_bn(self._h, C.cast(self._sync_cb, C.c_void_p), None, world))
        return True

    def _check_sync_shapes(self, B, T_in, T_out):
        """The merge of the per-rank statistics (k_bn_sync_combine) weights every rank equally: it is exact only when every rank
        normalises the same number of rows.  Checked on EVERY step with one 4-element MAX all-reduce (a per-rank cache of verified
        shapes would let one rank skip the collective while its peer issues it -- exactly when the shapes differ -- and pair the
        peer's shape exchange with this rank's first SyncBN exchange); unequal shards raise on every rank instead of silently
        producing wrong global statistics."""
        import torch.distributed as dist
        v = torch.tensor([B * T_in, B * T_out, -B * T_in, -B * T_out], dtype=torch.int64, device=self.device)
        dist.all_reduce(v, op=dist.ReduceOp.MAX, group=self._sync_group)
        hi_in, hi_out, lo_in, lo_out = (int(x) for x in v.tolist())
        if hi_in != -lo_in or hi_out != -lo_out:
            raise _lib.TacoError(_lib.TACO_ERR_SHAPE, "SyncBN needs the same number of rows on every rank: this rank has B*T_in = %d, "
                                 "B*T_out = %d, the group spans %d..%d and %d..%d" % (B * T_in, B * T_out, -lo_in, hi_in, -lo_out, hi_out))

    def set_deterministic(self, on=True):
        """Run-to-run reproducible steps (the reference's single-device step is): every row sum that normally leaves its workgroup
        through fp32 atomics -- weight gradients, bias / BatchNorm sums, embedding gradients -- becomes a two-stage sum in a fixed
        order.  The DEFAULT since round 4; set_deterministic(False) selects the atomics (14.87 instead of 15.85 ms per step at the C4
        shard, gradients equal up to summation order, ~3e-4 of the gradient scale between two runs).  192 MB of workspace."""
        _lib.check(self._lib.taco_train_set_deterministic(self._h, 1 if on else 0))
        self._ws = self._ws_eager = None       # the workspace size changes
        if getattr(self, "_graph", None) is not None:
            self._graph = None

    def set_exact_wgrad(self, on=True):
        """Weight gradients on the exact-fp32 MFMA (round 1's k_wgrad) instead of the default split-bf16 matrix-core kernel
        (k_wgrad_bf3: operands split three ways, six products, fp32-grade).  Process-wide A/B and test hook."""
        _lib.check(self._lib.taco_train_set_exact_wgrad(self._h, 1 if on else 0))
        if getattr(self, "_graph", None) is not None:
            self._graph = None

    def set_wgrad_planes(self, mode=1):
        """Split-bf16 weight gradients from pre-split operands (csrc/taco_wgrad_planes.h): 1 (default) = the large problems (a whole conv
        bank, proj_1, the linear head) convert their operands once into bf16 planes and multiply them with a kernel that converts nothing;
        2 = every eligible problem (test hook); 0 = all on k_wgrad_bf3 (the step before round 6's second half)."""
        _lib.check(self._lib.taco_train_set_wgrad_planes(self._h, int(mode)))
        self._ws = self._ws_eager = None       # the workspace size changes
        if getattr(self, "_graph", None) is not None:
            self._graph = None

    def planes_problems(self):
        """Weight gradients of the last backward pass that were computed from pre-split planes (a conv bank counts once)."""
        return int(self._lib.taco_train_planes_problems(self._h))

    # ---- engines ----
    def set_decoder_engine(self, mode=1):
        """1 (default): at the reference widths the teacher-forced decoder loop and the post-net scan of the forward run as the
        persistent whole-chip kernels of inference with tape outputs (csrc/taco_decoder_xcd.h, taco_bigru_xcd.h); 0: one launch per
        stage everywhere (round 1's engine; also what other widths and more than 64 rows use).  rnn_decoder_test_mode is a mode of the
        persistent kernel since round 4 (the step's own last frame is exchanged in place of the teacher's)."""
        mh = C.c_void_p(self._lib.taco_train_model(self._h))
        _lib.check(self._lib.taco_debug_set_decoder_persist(mh, int(mode), 0))
        if getattr(self, "_graph", None) is not None:
            self._graph = None          # a captured step keeps the engine it was captured with

    def set_exact_gemm(self, on=True):
        """Engines of the feed-forward GEMMs and their data gradients (include/taco_abi.h, taco_train_set_exact_gemm).  4 (default): see below.  3 (the default of rounds 2-3):
        forward on the exact-fp32 MFMA, data gradients on the split-bf16 kernels of inference (4e-6 of the gradient norm against the
        all-exact step).  True / 1: everything exact.  False / 0: everything split-bf16 (fastest; ~1e-3 of the gradient norm, near-ties of
        the forward may resolve differently).  2: forward split-bf16, data gradients exact (A/B hook).  4: forward on the six-product
        split (operands split three ways, fp32-grade products on the bf16 matrix cores), data gradients as in 3."""
        _lib.check(self._lib.taco_train_set_exact_gemm(self._h, int(on)))
        self.refresh()                 # the split-bf16 planes are (re)generated only while an engine that needs them is selected
        if getattr(self, "_graph", None) is not None:
            self._graph = None

    def set_bptt_engine(self, persistent=True):
        """True (default): back-propagation through the decoder loop is ONE whole-chip launch (csrc/taco_decoder_bwd_xcd.h) whenever the
        forward ran on the persistent decoder; False: the chain of per-stage launches (round 1's engine; A/B and test hook)."""
        _lib.check(self._lib.taco_train_set_bptt_engine(self._h, 1 if persistent else 0))
        if getattr(self, "_graph", None) is not None:
            self._graph = None

    def decoder_engine_info(self):
        """After a forward (synchronises): {'protocol': 0 launch per stage / 1 XCD-local / 2 write-through, 'per_xcd': [...]} of the last
        persistent decoder launch of the training forward."""
        torch.cuda.synchronize(self.device)
        mh = C.c_void_p(self._lib.taco_train_model(self._h))
        v = (C.c_int * 16)()
        _lib.check(self._lib.taco_debug_decoder_info(mh, v))
        return {"protocol": int(v[0]), "per_xcd": [int(x) for x in v[1:9]], "has_pack": bool(v[15]), "compute_units": int(v[14]),
                "bptt_protocol": int(v[9])}       # 0: the last decoder backward was the chain of per-stage launches

    def raise_device_error_for_test(self, value=2):
        """Test hook: sets the sticky device error word, as a persistent kernel whose bounded spin expired does."""
        mh = C.c_void_p(self._lib.taco_train_model(self._h))
        _lib.check(self._lib.taco_debug_raise_device_error(mh, int(value)))

    def check_device_errors(self):
        """Synchronises and raises if a persistent kernel of the training forward gave up (outputs and gradients invalid)."""
        mh = C.c_void_p(self._lib.taco_train_model(self._h))
        v = C.c_int(0)
        _lib.check(self._lib.taco_model_device_errors(mh, C.byref(v)))
        if v.value:
            raise _lib.TacoError(_lib.TACO_ERR_HIP, "a persistent kernel of the training forward timed out waiting for a peer workgroup")

    # ---- parameters ----
    def set_weights(self, weights):
        host = np.zeros(self.num_params, np.float32)
        for name, shape in self.spec:
            if name not in weights:
                raise _lib.TacoError(_lib.TACO_ERR_STATE, "weight '%s' missing" % name)
            v = np.asarray(weights[name], np.float32)
            if tuple(v.shape) != shape:
                raise _lib.TacoError(_lib.TACO_ERR_SHAPE, "weight '%s' has shape %s, expected %s" % (name, v.shape, shape))
            o, c = self.offsets[name]
            host[o:o + c] = v.reshape(-1)
        self.params.copy_(torch.from_numpy(host))
        self.refresh()

    def get_weights(self):
        host = self.params.detach().cpu().numpy()
        return {name: host[o:o + c].reshape(shape).copy() for (name, shape) in self.spec for (o, c) in [self.offsets[name]]}

    def grad_dict(self):
        host = self.grads.detach().cpu().numpy()
        return {name: host[o:o + c].reshape(shape).copy() for (name, shape) in self.spec for (o, c) in [self.offsets[name]]}

    def refresh(self):
        with torch.cuda.device(self.device):
            _lib.check(self._lib.taco_train_refresh(self._h, _st(), _p(self.params)))

    # ---- train-state checkpoints (train.py:175 `tf.train.Saver`, :189-203 restore, :242-244 save) ----
    def _layout_id(self):
        h = hashlib.sha256()
        for name, shape in self.spec:
            h.update(("%s:%s:%d;" % (name, "x".join(map(str, shape)), self.offsets[name][0])).encode())
        return np.frombuffer(h.digest()[:8], np.uint8).copy()

    def save_checkpoint(self, log_dir, max_to_keep=5, keep_every_n_hours=2.0):
        """`saver.save(sess, checkpoint_path, global_step=step)` (train.py:242-244): everything a resumed run needs, i.e. every variable
        the reference's Saver holds -- the parameters INCLUDING the BatchNorm moving statistics (they live in the flat parameter
        buffer), Adam's m and v, the step counters.  Two safetensors files (train_ops.train_state_paths); the weights file is a weight
        pack Synthesizer.load reads as it stands.  Older checkpoints are pruned like Saver(max_to_keep=5,
        keep_checkpoint_every_n_hours=2).  Returns the weights file."""
        from safetensors.numpy import save_file
        from .weights import save_weights
        torch.cuda.synchronize(self.device)
        wpath, opath = train_state_paths(log_dir, self.adam.global_step)
        save_weights(wpath, self.get_weights())
        save_file({"adam_m": self.adam.m.detach().cpu().numpy(), "adam_v": self.adam.v.detach().cpu().numpy(),
                   "global_step": np.asarray([self.adam.global_step], np.int64), "adam_t": np.asarray([self.adam.adam_t], np.int64),
                   "layout": self._layout_id()}, opath)
        prune_train_checkpoints(log_dir, max_to_keep, keep_every_n_hours)
        return wpath

    def restore_checkpoint(self, path_or_dir, reset_global_step=False):
        """`saver.restore(sess, get_most_recent_checkpoint(dir))` (train.py:189-193): parameters, BatchNorm moving statistics, Adam
        slots and counters come back, and the next train_step continues the run to the bit.  reset_global_step: the
        `--initialize_path` branch (train.py:194-203) -- every variable restored, then `global_step` assigned 0: the learning-rate
        schedule starts over while the Adam moments and the beta-power accumulators carry on.  Returns the global step."""
        from safetensors.numpy import load_file
        from .weights import load_weights
        if os.path.isdir(path_or_dir):
            ck = list_train_checkpoints(path_or_dir)
            if not ck:
                raise Exception(" [!] No checkpoint found in {}".format(path_or_dir))
            wpath = ck[-1][1]
        else:
            wpath = path_or_dir
        opath = wpath[:-len(".safetensors")] + ".optim.safetensors"
        if not os.path.exists(opath):
            raise _lib.TacoError(_lib.TACO_ERR_STATE, "%s has no optimizer state beside it (%s): a weight pack, not a train-state checkpoint"
                                 % (wpath, os.path.basename(opath)))
        st = load_file(opath)
        if st["adam_m"].shape != (self.num_params,) or not np.array_equal(st["layout"], self._layout_id()):
            raise _lib.TacoError(_lib.TACO_ERR_SHAPE, "the optimizer state of %s was written for another parameter layout (other hyper-parameters)" % wpath)
        self.set_weights(load_weights(wpath))
        self.adam.m.copy_(torch.from_numpy(st["adam_m"]))
        self.adam.v.copy_(torch.from_numpy(st["adam_v"]))
        self.adam.adam_t = int(st["adam_t"][0])
        self.adam.global_step = 0 if reset_global_step else int(st["global_step"][0])
        return self.adam.global_step

    # ---- one forward (+ backward) ----
    def forward_backward(self, inputs, input_lengths, mel_targets, linear_targets, loss_coeff=None, backward=True, keep_outputs=False,
                         rnn_decoder_test_mode=False, speaker_id=None, freeze_moving_averages=False):
        """Fills self.grads (when backward) and returns the device tensor [4] = loss, mel_loss, linear_loss, loss_without_coeff.
        The BatchNorm moving averages (in self.params) follow a forward+backward pass only (tacotron.py:334), and not even that with
        freeze_moving_averages (a warm-up whose result is thrown away)."""
        dev = self.device
        ids = torch.as_tensor(np.asarray(inputs) if not torch.is_tensor(inputs) else inputs).to(dev, torch.int32).contiguous()
        lens = torch.as_tensor(np.asarray(input_lengths) if not torch.is_tensor(input_lengths) else input_lengths).to(dev, torch.int32).contiguous()
        mt = torch.as_tensor(np.asarray(mel_targets) if not torch.is_tensor(mel_targets) else mel_targets).to(dev, torch.float32).contiguous()
        lt = torch.as_tensor(np.asarray(linear_targets) if not torch.is_tensor(linear_targets) else linear_targets).to(dev, torch.float32).contiguous()
        co = None if loss_coeff is None else torch.as_tensor(np.asarray(loss_coeff) if not torch.is_tensor(loss_coeff) else loss_coeff).to(dev, torch.float32).contiguous()
        B, T_in = ids.shape
        T_out = mt.shape[1]
        hp = self.hp
        spk = None
        if self.num_speakers > 1:
            spk = torch.zeros(B, dtype=torch.int32, device=dev) if speaker_id is None else \
                torch.as_tensor(np.asarray(speaker_id) if not torch.is_tensor(speaker_id) else speaker_id).to(dev, torch.int32).contiguous()
        if mt.shape != (B, T_out, hp.num_mels) or lt.shape != (B, T_out, hp.num_freq):
            raise Exception("targets must be [B, T_out, num_mels] / [B, T_out, num_freq], got %s / %s" % (tuple(mt.shape), tuple(lt.shape)))
        if self._sync_cb is not None:
            self._check_sync_shapes(B, T_in, T_out)
        nb = int(self._lib.taco_train_workspace_bytes(self._h, B, T_in, T_out))
        # A captured step has the address of ITS workspace baked into every kernel node: that buffer is never reallocated while the
        # graph is alive.  Eager calls beside the graph (another shape, a loss fetch) use a workspace of their own.
        wsattr = "_ws" if (getattr(self, "_graph", None) is None or getattr(self, "_capturing", False)) else "_ws_eager"
        ws = getattr(self, wsattr, None)
        if ws is None or ws.numel() < nb:
            if wsattr == "_ws" and getattr(self, "_capturing", False) and ws is not None:
                raise _lib.TacoError(_lib.TACO_ERR_STATE, "workspace grew during graph capture")
            ws = torch.empty(nb, dtype=torch.uint8, device=dev)
            setattr(self, wsattr, ws)
        mel = lin = ali = None
        if keep_outputs:
            mel = torch.empty((B, T_out, hp.num_mels), dtype=torch.float32, device=dev)
            lin = torch.empty((B, T_out, hp.num_freq), dtype=torch.float32, device=dev)
            ali = torch.empty((B, T_in, T_out // hp.reduction_factor), dtype=torch.float32, device=dev)
        self._ws_live = ws      # the workspace of the call in flight (the SyncBN callback reduces slices of it)
        with torch.cuda.device(dev):
            _lib.check(self._lib.taco_train_forward_backward(
                self._h, _st(), _p(self.params), _p(self.grads if backward else None), _p(ids), _p(lens), _p(spk), _p(mt), _p(lt), _p(co),
                B, T_in, T_out, int(bool(getattr(hp, "prioritize_loss", False))), int(getattr(hp, "sample_rate", 24000)), _p(self.losses),
                _p(mel), _p(lin), _p(ali), int(bool(rnn_decoder_test_mode)) | (2 if freeze_moving_averages else 0), _p(ws), ws.numel()))
        if self._sync_err is not None:
            e, self._sync_err = self._sync_err, None
            raise e
        self.mel_outputs, self.linear_outputs, self.alignments = mel, lin, ali
        return self.losses

    def capture(self, inputs, input_lengths, mel_targets, linear_targets, loss_coeff=None, speaker_id=None):
        """Record forward+backward for these shapes as ONE hipGraph over static input buffers (the ~6000 kernel launches
        of a step become one graph launch).  Later train_step() calls with the same shapes copy into the static buffers and
        replay.  Returns self."""
        dev = self.device
        if self._sync_cb is not None:
            raise _lib.TacoError(_lib.TACO_ERR_STATE, "SyncBN calls back into the host: it cannot be combined with a captured step")
        cv = lambda x, dt: (x if torch.is_tensor(x) else torch.as_tensor(np.asarray(x))).to(dev, dt).contiguous().clone()
        self._g_in = [cv(inputs, torch.int32), cv(input_lengths, torch.int32), cv(mel_targets, torch.float32),
                      cv(linear_targets, torch.float32), None if loss_coeff is None else cv(loss_coeff, torch.float32)]
        self._g_spk = None if speaker_id is None else cv(speaker_id, torch.int32)
        self._graph = None
        with torch.cuda.device(dev):
            # sizes the workspace outside the capture; its BatchNorm update is discarded (the step itself has not happened yet)
            self.forward_backward(*self._g_in, speaker_id=self._g_spk, freeze_moving_averages=True)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            self._capturing = True
            try:
                with torch.cuda.graph(graph):
                    self.forward_backward(*self._g_in, speaker_id=self._g_spk)
            finally:
                self._capturing = False
            self._graph = graph
            return self

    def _replay(self, inputs, input_lengths, mel_targets, linear_targets, loss_coeff, speaker_id=None):
        src = [inputs, input_lengths, mel_targets, linear_targets, loss_coeff, speaker_id]
        for dst, x in zip(self._g_in + [self._g_spk], src):
            if dst is None or x is None:
                if (dst is None) != (x is None):
                    return False
                continue
            x = x if torch.is_tensor(x) else torch.as_tensor(np.asarray(x))
            if tuple(x.shape) != tuple(dst.shape):
                return False
            if x.data_ptr() != dst.data_ptr():
                dst.copy_(x)
        # (replays mixed with eager steps of other shapes are safe: every stream-ordered clear in the library is a kernel launch --
        # hipMemsetAsync NODES of a captured graph were seen to fill with stale data once other work had run in between, see zero_async
        # in csrc/taco_lib.hip)
        self._graph.replay()
        return True

    def train_step(self, inputs, input_lengths, mel_targets, linear_targets, loss_coeff=None, speaker_id=None):
        """train.py:217-219: one fwd+bwd+update; returns (global_step, loss_without_coeff) like the reference's fetch."""
        if not (getattr(self, "_graph", None) is not None and self._replay(inputs, input_lengths, mel_targets, linear_targets, loss_coeff, speaker_id)):
            self.forward_backward(inputs, input_lengths, mel_targets, linear_targets, loss_coeff, backward=True, speaker_id=speaker_id)
        allreduce_gradients(self.grads)          # no-op on one process; RCCL all-reduce of the flat bucket otherwise
        self.adam.step(self.grads)
        self.refresh()
        return self.adam.global_step, self.losses[3]

    @property
    def global_step(self):
        return self.adam.global_step

    @property
    def learning_rate(self):
        return self.adam.learning_rate

    def close(self):
        if getattr(self, "_h", None):
            self._lib.taco_train_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
