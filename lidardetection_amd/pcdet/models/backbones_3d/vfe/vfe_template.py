import torch.nn as nn


class VFETemplate(nn.Module):
    """pcdet/models/backbones_3d/vfe/vfe_template.py: interface of the voxel feature encoders."""

    def __init__(self, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg

    def get_output_feature_dim(self):
        raise NotImplementedError

    def forward(self, **kwargs):
        raise NotImplementedError
