"""Import alias: ``import eae_amd`` loads the package that lives in the (non-identifier) directory
``hybrid-autoencoder-mlp-pipeline-for-satellite-image-classification_amd/``."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                    "hybrid-autoencoder-mlp-pipeline-for-satellite-image-classification_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_DIR, "__init__.py"),
                                               submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
