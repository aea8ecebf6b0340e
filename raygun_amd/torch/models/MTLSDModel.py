"""U-Net + affinity / LSD heads — drop-in for raygun.torch.models.MTLSDModel (reference: models/MTLSDModel.py:12-63)."""
import torch

from ..networks import UNet
from ..networks.UNet import ConvPass
from ... import hipnn


class MTLSDModel(torch.nn.Module):
    def __init__(self, unet_kwargs={"input_nc": 1, "ngf": 12, "fmap_inc_factor": 6,
                                    "downsample_factors": [(2, 2, 2), (2, 2, 2), (2, 2, 2)], "constant_upsample": True},
                 num_affs=3):
        super().__init__()
        self.unet = UNet(**unet_kwargs)
        self.aff_head = ConvPass(unet_kwargs["ngf"], num_affs, [[1, 1, 1]], activation="Sigmoid")
        self.lsd_head = ConvPass(unet_kwargs["ngf"], 10, [[1, 1, 1]], activation="Sigmoid")
        self.output_arrays = ["pred_affs", "pred_lsds"]
        self.data_dict = {}

    def add_log(self, writer, step):
        """MTLSDModel.py:39-55 — mid-slice image of the stored raw input."""
        for name, data in self.data_dict.items():
            img = data[0].squeeze() if len(data.shape) > 3 else data.squeeze()
            if len(img.shape) == 3:
                img = img[img.shape[0] // 2]
            if (img.min() < 0) and (img.min() >= -1.0) and (img.max() <= 1.0):
                img = (img * 0.5) + 0.5
            writer.add_image(name, img, global_step=step, dataformats="HW")

    @hipnn.scoped
    def forward(self, raw):
        self.data_dict.update({"raw": raw.detach()})
        z = self.unet(raw)
        affs = self.aff_head(z)
        lsds = self.lsd_head(z)
        return affs, lsds
