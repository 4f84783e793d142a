"""lvae -- MI355X-native QARV / QRes-VAE inference codec (drop-in for the reference's `lvae` package on the
encode/decode path: get_model / compress_mode / compress / decompress / compress_file / decompress_file)."""
from .paths import known_datasets
from .models.registry import get_model
from . import models
from .engine import NonFiniteError
