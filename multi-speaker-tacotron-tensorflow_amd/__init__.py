"""taco_amd -- MI355X-native Tacotron hot path (CBHG encoder -> attention decoder -> post-net CBHG ->
linear spectrogram) behind the reference's Tacotron / Synthesizer surface.  Compute lives in
csrc/libtaco_hip.so (hand-written gfx950 HIP); this package is the thin host mirror."""
from .hparams import hparams, HParams, basic_params, load_hparams, save_hparams   # noqa: F401
from .tacotron import Tacotron, create_model, input_lengths_from_tokens            # noqa: F401
from .synthesizer import Synthesizer                                               # noqa: F401
from .audio import GriffinLim                                                     # noqa: F401
from .trainer import Trainer                                                       # noqa: F401
from . import weights, dist, _lib, train_ops, tf_checkpoint, text, korean, feeder  # noqa: F401
