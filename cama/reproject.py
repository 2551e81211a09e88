"""Drop-in module name of the reference (cama/reproject.py): re-exports the MI355X implementation."""
from cama_amd.reproject import *  # noqa: F401,F403
from cama_amd import reproject as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})

# the north_star's name for the path (BASELINE.json: "Keep the Reprojector/PoseTransformer class surface"): a facade over
# ClipManager, see cama_amd/reprojector.py
from cama_amd.reprojector import Reprojector, load_configs  # noqa: E402,F401
