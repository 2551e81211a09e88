"""Drop-in module name of the reference (cama/dataset_reader.py): re-exports the MI355X implementation."""
from cama_amd.dataset_reader import *  # noqa: F401,F403
from cama_amd import dataset_reader as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
