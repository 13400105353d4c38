import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tiny-cuda-nn_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): compiled on demand from oracle/tcnn_oracle.cpp."""
    import oracle as orc

    orc.build()
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def tcnn():
    """The product's Python surface. Fails loudly if libtcnn_amd.so is missing -- there is no fallback."""
    lib = os.path.join(ROOT, "tiny-cuda-nn_amd", "libtcnn_amd.so")
    if not os.path.exists(lib):
        sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
        import build as _b

        _b.build()
    import tinycudann

    return tinycudann


# configurations shared by the tests
CONFIG_C3A = {  # README variant: T = 2^19, scale 2.0
    "loss": {"otype": "RelativeL2"},
    "optimizer": {"otype": "Adam", "learning_rate": 1e-2, "beta1": 0.9, "beta2": 0.99, "epsilon": 1e-15, "l2_reg": 1e-6},
    "encoding": {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16, "per_level_scale": 2.0},
    "network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 2},
}
CONFIG_C3B = {  # shipped data/config_hash.json: T = 2^15, scale 1.5
    "loss": {"otype": "RelativeL2"},
    "optimizer": {"otype": "Adam", "learning_rate": 1e-2, "beta1": 0.9, "beta2": 0.99, "epsilon": 1e-15, "l2_reg": 1e-6},
    "encoding": {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 15, "base_resolution": 16, "per_level_scale": 1.5},
    "network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 2},
}
CONFIG_C2 = {  # data/config_oneblob.json with the bench override 64 x 2
    "loss": {"otype": "RelativeL2"},
    "optimizer": {"otype": "Adam", "learning_rate": 1e-2, "beta1": 0.9, "beta2": 0.99, "epsilon": 1e-8, "l2_reg": 1e-8},
    "encoding": {"otype": "OneBlob", "n_bins": 64},
    "network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 2},
}
CONFIG_C1 = {  # Identity -> MLP 32 x 1 (CutlassMLP semantics), SURVEY 8d C1
    "loss": {"otype": "RelativeL2"},
    "optimizer": {"otype": "Adam", "learning_rate": 1e-2, "beta1": 0.9, "beta2": 0.99, "epsilon": 1e-8, "l2_reg": 1e-8},
    "encoding": {"otype": "Identity", "scale": 1.0, "offset": 0.0},
    "network": {"otype": "CutlassMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 32, "n_hidden_layers": 1},
}
CONFIG_C5 = {  # BASELINE config 5 at its real size: HashGrid L16 F4 T=2^22, 3-D -> 128 x 2; 210 937 856 parameters
    "loss": {"otype": "RelativeL2"},
    "optimizer": {"otype": "Adam", "learning_rate": 1e-2, "beta1": 0.9, "beta2": 0.99, "epsilon": 1e-15, "l2_reg": 1e-6},
    "encoding": {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 4, "log2_hashmap_size": 22, "base_resolution": 16, "per_level_scale": 2.0},
    "network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 128, "n_hidden_layers": 2},
}
CONFIG_C5_SMALL = {  # C5's shape (F = 4, 3-D, 128-wide) with a small table so the oracle finishes in seconds
    "loss": {"otype": "L2"},
    "optimizer": {"otype": "Adam", "learning_rate": 1e-2, "beta1": 0.9, "beta2": 0.99, "epsilon": 1e-15, "l2_reg": 1e-6},
    "encoding": {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 4, "log2_hashmap_size": 14, "base_resolution": 16, "per_level_scale": 2.0},
    "network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 128, "n_hidden_layers": 2},
}
