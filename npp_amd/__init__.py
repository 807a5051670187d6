"""npp_amd -- MI355X (gfx950) native hot path for NPPNet (GuHuangAI/NPP).

    from npp_amd.model_augment import Network          # == models.model_augment.Network
    from npp_amd.operations import OPS                 # == models.operations.OPS
    from npp_amd.criterion import Criterion_pose, Criterion_par   # == core.criterion

`install_as_reference_modules()` registers these under the reference's own module names (`models`,
`models.operations`, `models.genotypes`, `models.model_augment`, `models.model_search_interact`,
`core.criterion`) so that `augment_lip_sync.py` / `search_lip_sync.py` import them unchanged.
"""
import sys
import types

__version__ = "0.1.0"


def install_as_reference_modules(auto_graph=None):
    """auto_graph: True / False switches npp_amd.auto_graph (hipGraph replay of Network.forward + backward for an unchanged
    launcher) on or off; None leaves it to the NPP_AUTO_GRAPH environment variable."""
    if auto_graph is not None:
        from . import auto_graph as _ag
        _ag.ENABLED = bool(auto_graph)
    from . import criterion, genotypes, model_augment, model_search_interact, operations
    models = types.ModuleType("models")
    models.__path__ = []
    models.operations, models.genotypes, models.model_augment = operations, genotypes, model_augment
    sys.modules["models"] = models
    sys.modules["models.operations"] = operations
    sys.modules["models.genotypes"] = genotypes
    sys.modules["models.model_augment"] = model_augment
    models.model_search_interact = model_search_interact
    sys.modules["models.model_search_interact"] = model_search_interact
    core = sys.modules.get("core")
    if core is None:
        core = types.ModuleType("core")
        core.__path__ = []
        sys.modules["core"] = core
    core.criterion = criterion
    sys.modules["core.criterion"] = criterion
