"""moldiff_amd -- MI355X (gfx950) native implementation of MolDiff's denoising hot path.

Public surface mirrors the reference for the path in scope:
    MolDiff, BondPredictor, NodeEdgeNet, make_data_placeholder, load_config, seed_all
Importing this package never imports torch-side fallbacks or the CPU oracle; the HIP library
(moldiff_amd/libmoldiff_hip.so) is loaded on first use and its absence is a hard error.
"""
from .common import AttrDict  # noqa: F401
from .graph import NodeEdgeNet  # noqa: F401
from .model import MolDiff  # noqa: F401
from .bond_predictor import BondPredictor  # noqa: F401
from .harness import make_data_placeholder, load_config, seed_all, recipe_state_dict, is_frozen_key  # noqa: F401

__all__ = ['MolDiff', 'BondPredictor', 'NodeEdgeNet', 'make_data_placeholder', 'load_config', 'seed_all',
           'recipe_state_dict', 'is_frozen_key', 'AttrDict']
