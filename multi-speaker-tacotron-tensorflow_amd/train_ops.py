"""Training-side operators around the forward/backward of trainer.py:

* `l1_losses`      -- Tacotron.add_loss (models/tacotron.py:274-302) on device tensors (HIP reduction kernels)
* `FlatAdam`       -- Tacotron.add_optimizer's update (tacotron.py:305-336): LR schedule, clip_by_global_norm(1.0),
                      tf.train.AdamOptimizer in TF form, fused over ONE flat fp32 parameter buffer
* `allreduce_gradients` -- the single collective of the data-parallel train step: one all-reduce
                      (sum) of the flat gradient bucket over RCCL (xGMI), then division by the world size because
                      the loss is a mean over the global batch.

PyTorch provides device memory, the stream and torch.distributed; the arithmetic is in libtaco_hip.so."""
import ctypes as C
import glob
import os
import re

import torch

from . import _lib


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def l1_losses(mel_outputs, mel_targets, linear_outputs, linear_targets, loss_coeff=None, prioritize_loss=False,
              sample_rate=24000):
    """Returns a device tensor [4] = (loss, mel_loss, linear_loss, loss_without_coeff)."""
    lib = _lib.load_library()
    B, T, M = mel_outputs.shape
    F = linear_outputs.shape[-1]
    assert mel_targets.shape == mel_outputs.shape and linear_targets.shape == linear_outputs.shape
    out = torch.empty(4, dtype=torch.float32, device=mel_outputs.device)
    ws = torch.empty(1 << 16, dtype=torch.uint8, device=mel_outputs.device)
    tens = [t.contiguous().float() for t in (mel_outputs, mel_targets, linear_outputs, linear_targets)]
    coeff = loss_coeff.contiguous().float() if loss_coeff is not None else None
    _lib.check(lib.taco_loss_f32(_st(), _p(tens[0]), _p(tens[1]), _p(tens[2]), _p(tens[3]), _p(coeff), B, T, M, F,
                                 int(bool(prioritize_loss)), int(sample_rate), _p(out), _p(ws), ws.numel()))
    return out


class FlatAdam(object):
    """One flat fp32 buffer for all parameters (37.3 MB at the default hparams), Adam slots beside it."""

    def __init__(self, flat_params, initial_learning_rate=0.002, beta1=0.9, beta2=0.999, epsilon=1e-8,
                 decay_learning_rate_mode=0, is_randomly_initialized=True, clip_norm=1.0):
        assert flat_params.dtype == torch.float32 and flat_params.is_contiguous() and flat_params.dim() == 1
        self.params = flat_params
        self.m = torch.zeros_like(flat_params)
        self.v = torch.zeros_like(flat_params)
        self.global_step = 0          # what the learning-rate schedule reads (tacotron.py:312-325); train.py:196-197 resets it to 0 for --initialize_path
        self.adam_t = 0               # Adam updates applied so far = the exponent of TF's beta1_power / beta2_power accumulators, which a restore keeps
        self.hyper = (initial_learning_rate, beta1, beta2, epsilon, decay_learning_rate_mode, is_randomly_initialized, clip_norm)
        self._ws = torch.empty(1 << 14, dtype=torch.uint8, device=flat_params.device)
        self.gnorm = torch.zeros(1, dtype=torch.float32, device=flat_params.device)
        self._lib = _lib.load_library()

    @property
    def learning_rate(self):
        lr0, _, _, _, mode, rnd, _ = self.hyper
        return float(self._lib.taco_learning_rate(self.global_step, lr0, mode, int(rnd)))

    def step(self, flat_grads):
        lr0, b1, b2, eps, mode, rnd, clip = self.hyper
        assert flat_grads.shape == self.params.shape and flat_grads.dtype == torch.float32
        _lib.check(self._lib.taco_adam_step_f32(_st(), _p(self.params), _p(flat_grads.contiguous()), _p(self.m), _p(self.v),
                                                self.params.numel(), self.adam_t, self.learning_rate, b1, b2, eps, clip,
                                                _p(self.gnorm), _p(self._ws), self._ws.numel()))
        self.global_step += 1
        self.adam_t += 1


# ---- train-state checkpoints: what tf.train.Saver writes and restores for the training run (train.py:175,189-203,242-244) ----
def train_state_paths(log_dir, step):
    """(weights file, optimizer file) of a checkpoint: `model.ckpt-<step>.safetensors` (the parameters incl. the BatchNorm moving statistics:
    what Synthesizer.load reads) and `model.ckpt-<step>.optim.safetensors` (Adam m / v, global_step, the number of Adam updates)."""
    base = os.path.join(log_dir, "model.ckpt-%d" % int(step))
    return base + ".safetensors", base + ".optim.safetensors"


def list_train_checkpoints(log_dir):
    """[(step, weights file)] in ascending step order (a weights file without its optimizer file is a weight pack, not a resumable state)."""
    out = []
    for p in glob.glob(os.path.join(log_dir, "model.ckpt-*.safetensors")):
        m = re.match(r"model\.ckpt-(\d+)\.safetensors$", os.path.basename(p))
        if m and os.path.exists(p[:-len(".safetensors")] + ".optim.safetensors"):
            out.append((int(m.group(1)), p))
    return sorted(out)


def prune_train_checkpoints(log_dir, max_to_keep=5, keep_every_n_hours=2.0):
    """tf.train.Saver(max_to_keep=5, keep_checkpoint_every_n_hours=2) (train.py:175): the newest `max_to_keep` stay; of the older ones, one per
    `keep_every_n_hours` of file time stays for good.  Returns the paths removed."""
    ck = list_train_checkpoints(log_dir)
    removed = []
    if max_to_keep is None or len(ck) <= max_to_keep:
        return removed
    last_kept = None
    for step, path in ck[:-max_to_keep]:
        t = os.path.getmtime(path)
        if keep_every_n_hours and (last_kept is None or t - last_kept >= keep_every_n_hours * 3600.0):
            last_kept = t
            continue
        for q in (path, path[:-len(".safetensors")] + ".optim.safetensors"):
            if os.path.exists(q):
                os.remove(q)
                removed.append(q)
    return removed


def allreduce_gradients(flat_grads):
    """In place: sum over ranks (RCCL all-reduce of ONE bucket), divided by the world size."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM)
        flat_grads.div_(dist.get_world_size())
    return flat_grads


def raise_if_loss_exploded(loss, step):
    """The guard of the reference's training loop, right after the fetch of the step (train.py:228-230): `loss > 100 or isnan(loss)`
    raises Exception('Loss Exploded').  `loss` is the step's loss_without_coeff (a device scalar or a float).  A device fault of a
    persistent kernel poisons the step's losses with NaN (taco_abi.h), so this is also where a caller's loop learns of one."""
    import math
    v = float(loss.item() if hasattr(loss, "item") else loss)
    if v > 100 or math.isnan(v):
        raise Exception("Loss Exploded")
    return v
