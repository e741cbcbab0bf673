"""hands_amd -- MI355X-native forward path of the WildHands hand-mesh regressor.

Host side mirrors the reference's module API (``HandsLight``, ``xdict``); compute is
``libhands_hip.so`` (hand-written gfx950 HIP, see include/hands_hip.h).
"""
from .xdict import xdict, prefix_dict  # noqa: F401
from .hands_light import HandsLight, DEFAULT_ARGS, ManoHeadsPlan  # noqa: F401
from .hamer import HAMER, HAMER_DEFAULT_ARGS  # noqa: F401
from .handoccnet import HandOccNet, HANDOCC_DEFAULT_ARGS  # noqa: F401
from .mano import ManoAsset, synthetic_mano_asset, build_mano_asset  # noqa: F401
from .weights import apply_recipe, synthetic_inputs  # noqa: F401
from .frontend import HandsFrontEnd  # noqa: F401
from .graph import GraphedForward  # noqa: F401
from .wrapper import HandsWrapper, HaMeRWrapper, HandOccNetWrapper  # noqa: F401

__all__ = ["HandsLight", "DEFAULT_ARGS", "ManoHeadsPlan", "HAMER", "HAMER_DEFAULT_ARGS", "HandOccNet", "HANDOCC_DEFAULT_ARGS", "xdict", "prefix_dict", "ManoAsset", "synthetic_mano_asset",
           "build_mano_asset", "apply_recipe", "synthetic_inputs", "HandsFrontEnd", "GraphedForward", "HandsWrapper",
           "HaMeRWrapper", "HandOccNetWrapper"]
