"""Import alias: the product package lives in the directory `multi-speaker-tacotron-tensorflow_amd/`,
whose name is not a valid Python identifier.  `import taco_amd` loads that directory as the package
`taco_amd` (sub-modules: taco_amd.tacotron, taco_amd.synthesizer, taco_amd.hparams, ...)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "multi-speaker-tacotron-tensorflow_amd")
_spec = importlib.util.spec_from_file_location(
    "taco_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["taco_amd"] = _mod
_spec.loader.exec_module(_mod)
