"""`spconv` — the symbols the reference imports from the external spconv v1.x package, on liblidar_hip.so.

    import spconv                                   (pcdet/models/backbones_3d/spconv_backbone.py:3)
    spconv.SparseConvTensor / SubMConv3d / SparseConv3d / SparseInverseConv3d / SparseSequential / SparseModule
    from spconv.utils import VoxelGeneratorV2       (pcdet/datasets/processor/data_processor.py:51)

Semantics follow SURVEY.md Appendix A (spconv v1.2): weight parameter (kD, kH, kW, Cin, Cout), SubM forces stride 1 /
padding k//2 and keeps the input sites, regular conv emits a site wherever >= 1 active input falls in the receptive
field, inverse conv reuses the paired conv's rulebook by `indice_key`, rulebooks are cached in `indice_dict`.
spconv itself is absent from /root/reference and from this image: parity is pinned against dense
torch.nn.functional.conv3d and a brute-force rulebook (tests/test_gpu_spconv.py), i.e. "parity unpinned" w.r.t. upstream.
"""
from .conv import SparseConv3d, SparseConvolution, SparseInverseConv3d, SubMConv3d
from .modules import SparseModule, SparseSequential, prebuild_rulebooks, run_stages_pipelined
from .tensor import SparseConvTensor
from .graphed import GraphedStages
from . import ops, utils

__all__ = ["SparseConvTensor", "SparseModule", "SparseSequential", "SparseConvolution", "SubMConv3d", "SparseConv3d",
           "SparseInverseConv3d", "prebuild_rulebooks", "run_stages_pipelined", "GraphedStages", "ops", "utils"]
