"""d-vae_amd: the D-VAE DAG encode / decode / ELBO hot path on MI355X (gfx950).

Import name is ``dvae_amd`` (see dvae_amd.py at the repository root; the
directory name is not a Python identifier).
"""
from .metrics import reconstruction_accuracy  # noqa: F401
from .model import Decoder, Dvae, Encoder  # noqa: F401
from .svae import Svae  # noqa: F401
from . import optim, svae, synth  # noqa: F401
